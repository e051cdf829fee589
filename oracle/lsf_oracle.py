"""CPU oracle for the per-voxel warp-field gradient-descent path.   *** TEST INFRASTRUCTURE ***

This file is a vectorised, dimension-generic (D = 2, 3) numpy restatement of the numpy/Python hot path of
Algomorph/LevelSetFusion-Python.  It is the *checker* for the HIP path: only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it.  The shipped package never does (and fails loudly when its
HIP library is missing) -- there is no CPU fallback in the product.

Parity status
  * 2-D: PINNED.  Checked against (i) the reference's own golden literals re-captured as data under
    tests/golden/ (16x16 hierarchical pair + warp, 4x4 Slavcheva goldens, field_warping cases, convolution
    known answers, pyramid known answers, convergence-report known answer) and (ii) outputs of the reference
    itself imported in the build container (tests/golden/make_golden.py).  See tests/test_oracle_golden.py.
  * 3-D optimizers: "parity unpinned" by any reference test (the reference's only 3-D optimizer lives in an
    un-vendored C++ submodule, github.com/Algomorph/LevelSetFusion-CPP @ master, absent from the checkout).
    The 3-D rules here are the dimension generalisation written down in DESIGN.md section 3 and are
    pinned indirectly by 2-D-embedding tests (a z-constant volume must reproduce the 2-D result on interior
    slices, bit for bit).  The 3-D *convolution* and *linear resampling* pieces ARE pinned by reference tests.

Conventions
  scalar fields  float32 [y,x] or [z,y,x];  vector fields float32 [...,c], c = 0:x(u) 1:y(v) 2:z(w)
  (layout of math_utils/convolution.py:92-106).  Component c displaces along spatial axis  D-1-c.

Arithmetic is float32 with separately rounded multiply and add (numpy semantics under NEP 50), in the
exact operation order written below; the HIP kernels are compiled with -ffp-contract=off and follow the
same order so that everything except sum-reductions matches this file bit for bit.
"""
import math

import numpy as np

F32 = np.float32
SCALING_FACTOR = F32(10.0)  # data_term.py:183, level_set_term.py:54
SNAP_EPS = F32(1e-6)        # field_warping.py:138
LEVEL_SET_EPS = F32(1e-5)   # level_set_term.py:28


# =====================================================================================================
#  a1/a2  D-linear resampling under a warp         (utils/sampling.py:139-175,222-263; field_warping.py:67-109)
# =====================================================================================================
def _axis_offsets(axis0_offset, d):
    """per-axis global offsets of a slab's index 0: an int = a z-slab (axis 0), a tuple = one offset per axis (slabs cut
    along y carry theirs in position 1)"""
    if isinstance(axis0_offset, (tuple, list)):
        return [int(v) for v in axis0_offset] + [0] * (d - len(axis0_offset))
    return [int(axis0_offset)] + [0] * (d - 1)


def _grid_positions(warp, axis0_offset=0):
    """float32 sample positions p + warp[p], one array per spatial axis (axis order), as
    field_warping.py:82  Point2d(x, y) + Point2d(coordinates=warp[y, x])  evaluates them under numpy>=2.
    axis0_offset: the array is a z-slab whose slice 0 is slice `axis0_offset` of the whole volume -- positions along
    axis 0 are formed from the GLOBAL coordinate (float32 rounding of coordinate + displacement depends on the
    coordinate's magnitude), sample_linear takes the same offset back out of the tap indices.
    A tuple gives one offset per axis (slabs cut along y)."""
    shape = warp.shape[:-1]
    d = len(shape)
    offs = _axis_offsets(axis0_offset, d)
    pos = []
    for axis in range(d):
        c = d - 1 - axis
        idx_shape = [1] * d
        idx_shape[axis] = shape[axis]
        coord = np.arange(shape[axis], dtype=F32).reshape(idx_shape)
        if offs[axis]:
            coord = (coord + F32(offs[axis])).astype(F32)
        pos.append((coord + warp[..., c]).astype(F32))
    return pos


def sample_linear(field, pos, oob, axis0_offset=0):
    """D-linear interpolation of `field` at float32 positions `pos` (list, axis order).  Every tap that
    falls outside the array reads `oob` (scalar or per-output array).  Lerp order: slowest axis first
    (2-D: along y then x, sampling.py:170-172; 3-D adds z in front -- DESIGN.md section 3)."""
    d = field.ndim
    shape = field.shape
    base = [np.floor(p) for p in pos]
    ratio = [(p - b).astype(F32) for p, b in zip(pos, base)]
    inv = [(F32(1.0) - r).astype(F32) for r in ratio]
    offs = _axis_offsets(axis0_offset, d)
    base_i = [np.clip(b - offs[a], -2, shape[a] + 1).astype(np.int64) for a, b in enumerate(base)]
    oob_arr = np.broadcast_to(np.asarray(oob, dtype=F32), pos[0].shape)

    def tap(offsets):
        idx = []
        inside = np.ones(pos[0].shape, dtype=bool)
        for a in range(d):
            i = base_i[a] + offsets[a]
            inside &= (i >= 0) & (i < shape[a])
            idx.append(np.clip(i, 0, shape[a] - 1))
        return np.where(inside, field[tuple(idx)], oob_arr).astype(F32)

    # corner values, then reduce axis 0 first (slowest), last axis last
    corners = {}
    for code in range(1 << d):
        offs = tuple((code >> (d - 1 - a)) & 1 for a in range(d))
        corners[offs] = tap(offs)
    for a in range(d):
        nxt = {}
        for offs, v in corners.items():
            if offs[0] == 0:
                rest = offs[1:]
                v1 = corners[(1,) + rest]
                nxt[rest] = (v * inv[a] + v1 * ratio[a]).astype(F32)
        corners = nxt
    return corners[()]


def warp_field(field, warp):
    """field_warping.py:67-85 -- OOB taps read 1."""
    return sample_linear(field, _grid_positions(warp), F32(1.0))


def warp_field_replacement(field, warp, replacement):
    """field_warping.py:88-109 -- OOB taps read `replacement`."""
    return sample_linear(field, _grid_positions(warp), F32(replacement))


def is_truncated(field):
    """utils/tsdf_set_routines.py:42-43 -- truncated <=> value is exactly +-1."""
    return np.abs(field) == F32(1.0)


def warp_field_advanced(canonical, live, warp, gradient=None, band_union_only=False, known_values_only=False,
                        substitute_original=False, axis0_offset=0):
    """field_warping.py:112-151.  Returns the new live field; zeroes warp[p] (and gradient[p]) in place where
    the resampled value snaps to +-1.  axis0_offset: see _grid_positions (z-slab tests only)."""
    pos = _grid_positions(warp, axis0_offset)
    oob = live if substitute_original else F32(1.0)
    new = sample_linear(live, pos, oob, axis0_offset)
    snap = (F32(1.0) - np.abs(new)) < SNAP_EPS
    new = np.where(snap, np.sign(new), new).astype(F32)
    skip = np.zeros(live.shape, dtype=bool)
    if band_union_only:
        skip |= is_truncated(live) & is_truncated(canonical)
    if known_values_only:
        skip |= (live == F32(1.0))
    new = np.where(skip, live, new).astype(F32)
    snap &= ~skip
    warp[snap] = 0.0
    if gradient is not None:
        gradient[snap] = 0.0
    return new


# =====================================================================================================
#  a4  np.gradient  (interior central, one-sided first order at borders), returned per COMPONENT (x,y[,z])
# =====================================================================================================
def gradient(field):
    d = field.ndim
    out = []
    for c in range(d):
        axis = d - 1 - c
        f = np.moveaxis(field, axis, 0)
        g = np.empty_like(f)
        if f.shape[0] == 1:
            g[...] = 0
        else:
            g[1:-1] = (f[2:] - f[:-2]) * F32(0.5)
            g[0] = f[1] - f[0]
            g[-1] = f[-1] - f[-2]
        out.append(np.ascontiguousarray(np.moveaxis(g, 0, axis)))
    return out


# =====================================================================================================
#  a5/a6  pyramid restrict (2^D mean) and prolong (repeat)          (hierarchical/pyramid.py:28-56)
# =====================================================================================================
def restrict_mean(field):
    """2x2(x2) block mean.  2-D order ((a00+a01)+a10)+a11 then /4 (numpy's float32 reduction of the 4-vector,
    pyramid.py:50-53); 3-D = (s(z0)+s(z1))/8 with s the 2-D block sums, so z-constant volumes embed exactly."""
    if field.ndim == 2:
        s = ((field[0::2, 0::2] + field[0::2, 1::2]) + field[1::2, 0::2]) + field[1::2, 1::2]
        return (s * F32(0.25)).astype(F32)
    s0 = ((field[0::2, 0::2, 0::2] + field[0::2, 0::2, 1::2]) + field[0::2, 1::2, 0::2]) + field[0::2, 1::2, 1::2]
    s1 = ((field[1::2, 0::2, 0::2] + field[1::2, 0::2, 1::2]) + field[1::2, 1::2, 0::2]) + field[1::2, 1::2, 1::2]
    return ((s0 + s1) * F32(0.125)).astype(F32)


def is_power_of_two(n):
    return n > 0 and (n & (n - 1)) == 0


def pyramid_level_count(shape, maximum_chunk_size):
    """pyramid.py:31-45 incl. its ValueErrors."""
    if not all(is_power_of_two(int(s)) for s in shape):
        raise ValueError("The argument 'field' must be an array where each dimension is a power of two.")
    if not is_power_of_two(int(maximum_chunk_size)):
        raise ValueError("The argument 'maximum_chunk_size' must be an integer power of 2, i.e. 4, 8, 16, etc.")
    p = int(math.log2(maximum_chunk_size))
    if min(int(math.log2(s)) for s in shape) <= p:
        raise ValueError("maximum chunk size {:d} is too large for a field of size {:s}"
                         .format(maximum_chunk_size, str(tuple(shape))))
    return p + 1


def pyramid(field, maximum_chunk_size=8, linear=False):
    """levels, coarsest first (pyramid.py:45-56); linear=True: ResamplingStrategy.LINEAR restriction (3-D)."""
    n = pyramid_level_count(field.shape, maximum_chunk_size)
    levels = [np.array(field, dtype=F32, copy=True)]
    for _ in range(1, n):
        levels.append(downsample2x_linear(levels[-1]).astype(F32) if linear else restrict_mean(levels[-1]))
    levels.reverse()
    return levels


def prolong_repeat(warp):
    """hierarchical_optimizer2d.py:155-156 -- nearest 2x upsample, vectors NOT rescaled."""
    d = warp.ndim - 1
    for axis in range(d):
        warp = warp.repeat(2, axis=axis)
    return np.ascontiguousarray(warp)


# =====================================================================================================
#  Laplacian with edge replication  (scipy.ndimage.laplace, modes 'nearest' / 'reflect' coincide for 3 taps)
# =====================================================================================================
def laplace_replicate(a):
    """scipy evaluates each axis' [1,-2,1] correlation in double and stores float32, then sums the axes in
    float32, axis 0 first (ndimage generic_laplace / correlate1d symmetric branch:  -2*a0 + (ap + am))."""
    out = None
    for axis in range(a.ndim):
        f = np.moveaxis(a, axis, 0).astype(np.float64)
        am = np.concatenate((f[:1], f[:-1]), axis=0)
        ap = np.concatenate((f[1:], f[-1:]), axis=0)
        d2 = np.moveaxis((-2.0 * f + (ap + am)).astype(F32), 0, axis)
        out = d2 if out is None else (out + d2).astype(F32)
    return np.ascontiguousarray(out)


# =====================================================================================================
#  a9/a10  separable convolution of a vector field            (math_utils/convolution.py:70-132)
# =====================================================================================================
def _convolve_axis(vf, kernel64, axis):
    """np.convolve(line, k, 'same') along `axis`, zero padded: out[i] = sum_j k[j]*line[i + c - j], c = (len-1)//2
    ('same' cuts the full convolution at (N - 1) // 2; for the odd kernels the reference uses that is len // 2).
    Accumulated in float64 in tap order j = 0..n-1 and rounded to float32 on store (convolution.py:78-83:
    float64 kernel => float64 accumulation; with a float32 kernel the reference differs by <= 1 ulp)."""
    n = len(kernel64)
    c = (n - 1) // 2
    f = np.moveaxis(vf, axis, 0).astype(np.float64)
    length = f.shape[0]
    acc = np.zeros_like(f)
    for j in range(n):
        s = c - j  # source offset
        lo = max(0, -s)
        hi = min(length, length - s)
        if hi > lo:
            acc[lo:hi] = acc[lo:hi] + kernel64[j] * f[lo + s:hi + s]
    return np.ascontiguousarray(np.moveaxis(acc.astype(F32), 0, axis))


def _conv_axis_order(d):
    # 2-D: y then x (convolution.py:77-83); 3-D: x, y, z (convolution.py:94-105)
    return [0, 1] if d == 2 else [2, 1, 0]


def convolve_with_kernel(vf, kernel):
    """in place, returns vf"""
    k = np.asarray(kernel, dtype=np.float64)
    d = vf.ndim - 1
    cur = vf
    for axis in _conv_axis_order(d):
        cur = _convolve_axis(cur, k, axis)
    np.copyto(vf, cur)
    return vf


def convolve_with_kernel_preserve_zeros(vf, kernel):
    """convolution.py:114-132: mask |v|<1e-6 per component from the INPUT, re-zeroed after every pass.
    (2-D in the reference; the 3-D version applies the same rule to the x,y,z pass order.)"""
    k = np.asarray(kernel, dtype=np.float64)
    d = vf.ndim - 1
    zero = np.abs(vf) < 1e-6
    cur = vf
    for axis in _conv_axis_order(d):
        cur = _convolve_axis(cur, k, axis)
        cur[zero] = 0.0
    np.copyto(vf, cur)
    return vf


def vector_norm(vf):
    """np.linalg.norm(vf, axis=-1) for float32: sqrt(((v0^2 + v1^2) [+ v2^2]))."""
    s = vf[..., 0] * vf[..., 0]
    for c in range(1, vf.shape[-1]):
        s = (s + vf[..., c] * vf[..., c]).astype(F32)
    return np.sqrt(s).astype(F32)


def first_argmax(a):
    flat = int(np.argmax(a))
    return flat, np.unravel_index(flat, a.shape)


# =====================================================================================================
#  a19  Sobolev kernel generation (host side)                  (slavcheva/sobolev_filter.py:107-252)
# =====================================================================================================
def generate_1d_sobolev_kernel(size=7, strength=0.1, precision=np.float32):
    s3 = size ** 3
    lap = np.zeros((s3, s3), precision)
    for i in range(s3):
        lap[i, i] = -6.0
        for off in (-1, 1, size, -size, -size * size, size * size):
            j = i + off
            if 0 <= j < s3:  # flat-index clipping only (sobolev_filter.py:154-158)
                lap[i, j] = 1.0
    rhs = np.zeros((s3, 1), precision)
    rhs[s3 // 2] = 1.0
    flat = np.linalg.solve(np.identity(s3, precision) - strength * lap, rhs)
    cube = flat.reshape((size, size, size))
    # mode-1 unfolding (tucker.py:99-108 with tenmat): rows = index along axis 1
    m = np.moveaxis(cube, 1, 0).reshape(size, -1)
    u, _, _ = np.linalg.svd(m)
    k = -u[:, 0]
    if k[size // 2] < 0:  # LAPACK sign convention guard: the filter's centre tap is positive
        k = -k
    return k.astype(precision)


# =====================================================================================================
#  Hierarchical optimizer                                  (hierarchical/hierarchical_optimizer2d.py:123-246)
# =====================================================================================================
class HierarchicalOracle:
    def __init__(self, tikhonov_term_enabled=True, gradient_kernel_enabled=True, maximum_chunk_size=8, rate=0.1,
                 maximum_iteration_count=100, maximum_warp_update_threshold=0.001, data_term_amplifier=1.0,
                 tikhonov_strength=0.2, kernel=None, linear_resampling=False):
        self.linear_resampling = linear_resampling
        self.maximum_chunk_size = maximum_chunk_size
        self.rate = rate
        self.data_term_amplifier = data_term_amplifier
        self.tikhonov_strength = tikhonov_strength if tikhonov_term_enabled else 0.0
        self.tikhonov_term_enabled = bool(tikhonov_term_enabled and tikhonov_strength != 0.0)
        self.gradient_kernel = kernel if gradient_kernel_enabled else None
        self.gradient_kernel_enabled = bool(gradient_kernel_enabled and kernel is not None)
        self.maximum_warp_update_threshold = maximum_warp_update_threshold
        self.maximum_iteration_count = maximum_iteration_count
        self.per_level_iteration_counts = []
        self.per_level_max_updates = []
        # float64 sums behind the reference's optional printouts (hierarchical_optimizer2d.py:204-210,233-238):
        # normalised data energy = 1e6 * sum / N, normalised tikhonov energy = 1e6 * 0.5 * sum / N
        self.per_level_data_energy_sums = []
        self.per_level_tikhonov_energy_sums = []
        self.iteration_hook = None  # f(level, iteration, warp, gradient, max_update)

    def iteration(self, canonical, live, live_grads, warp, g_prev):
        """one pass of hierarchical_optimizer2d.py:186-225; returns (new gradient, max update length).
        `warp` is updated in place."""
        resampled = warp_field(live, warp)
        diff = (resampled - canonical).astype(F32)
        comps = [(diff * warp_field_replacement(g, warp, 0.0)).astype(F32) for g in live_grads]
        data_gradient = np.stack(comps, axis=-1)
        self._tikhonov_energy_sum = 0.0
        if self.tikhonov_term_enabled:
            for c in range(g_prev.shape[-1]):  # np.gradient of every component of the previous gradient, squared
                for d_axis in gradient(np.ascontiguousarray(g_prev[..., c])):
                    self._tikhonov_energy_sum += float((d_axis.astype(F32) ** 2).astype(np.float64).sum())
            tik = np.stack([laplace_replicate(g_prev[..., c]) for c in range(g_prev.shape[-1])], axis=-1)
            g = (F32(self.data_term_amplifier) * data_gradient - F32(self.tikhonov_strength) * tik).astype(F32)
        else:
            g = (F32(self.data_term_amplifier) * data_gradient).astype(F32)
        if self.gradient_kernel_enabled:
            convolve_with_kernel(g, self.gradient_kernel)
        warp -= (F32(self.rate) * g).astype(F32)
        lengths = vector_norm(g)
        _, at = first_argmax(lengths)
        return g, float(lengths[at]), diff

    def optimize_level(self, level, canonical, live, live_grads, warp):
        g = np.zeros_like(warp)
        max_update = float(np.finfo(np.float32).max)
        it = 0
        maxes, data_sums, tik_sums = [], [], []
        while not (max_update < self.maximum_warp_update_threshold or it >= self.maximum_iteration_count):
            g, max_update, diff = self.iteration(canonical, live, live_grads, warp, g)
            maxes.append(max_update)
            data_sums.append(float((diff.astype(np.float64) ** 2).sum()))
            tik_sums.append(self._tikhonov_energy_sum)
            if self.iteration_hook is not None:
                self.iteration_hook(level, it, warp, g, max_update)
            it += 1
        self.per_level_iteration_counts.append(it)
        self.per_level_max_updates.append(maxes)
        self.per_level_data_energy_sums.append(data_sums)
        self.per_level_tikhonov_energy_sums.append(tik_sums)
        return warp

    def optimize(self, canonical_field, live_field):
        canonical_field = np.asarray(canonical_field, dtype=F32)
        live_field = np.asarray(live_field, dtype=F32)
        d = live_field.ndim
        grads = gradient(live_field)  # at full resolution, THEN averaged (hierarchical_optimizer2d.py:126-131)
        lin = self.linear_resampling
        canonical_pyr = pyramid(canonical_field, self.maximum_chunk_size, lin)
        live_pyr = pyramid(live_field, self.maximum_chunk_size, lin)
        grad_pyrs = [pyramid(g, self.maximum_chunk_size, lin) for g in grads]
        self.per_level_iteration_counts = []
        self.per_level_max_updates = []
        self.per_level_data_energy_sums = []
        self.per_level_tikhonov_energy_sums = []
        warp = None
        n_levels = len(canonical_pyr)
        for level in range(n_levels):
            if level == 0:
                warp = np.zeros(canonical_pyr[0].shape + (d,), dtype=F32)
            warp = self.optimize_level(level, canonical_pyr[level], live_pyr[level],
                                       [gp[level] for gp in grad_pyrs], warp)
            if level != n_levels - 1:
                if lin:
                    warp = np.stack([upsample2x_linear(np.ascontiguousarray(warp[..., c])) for c in range(d)], axis=-1)
                else:
                    warp = prolong_repeat(warp)
        return warp


# =====================================================================================================
#  Slavcheva (KillingFusion / SobolevFusion style) optimizer   (slavcheva/slavcheva_optimizer2d.py:163-408)
# =====================================================================================================
DIRECT, VECTORIZED = 0, 1
BASIC, THRESHOLDED_FDM = 0, 2
TIKHONOV, KILLING = 0, 1


def _shift(a, axis, step, oob):
    """a sampled at index+step along `axis`; `oob`: float constant, or 'centre' (own value), spatial axes only"""
    n = a.shape[axis]
    out = np.empty_like(a)
    src = [slice(None)] * a.ndim
    dst = [slice(None)] * a.ndim
    fill = [slice(None)] * a.ndim
    if step > 0:
        src[axis], dst[axis], fill[axis] = slice(step, None), slice(0, n - step), slice(n - step, None)
    else:
        src[axis], dst[axis], fill[axis] = slice(0, n + step), slice(-step, None), slice(0, -step)
    out[tuple(dst)] = a[tuple(src)]
    out[tuple(fill)] = a[tuple(fill)] if isinstance(oob, str) else oob
    return out


def _shift2(a, axis_a, step_a, axis_b, step_b, oob):
    """diagonal neighbour.  OOB semantics are per final coordinate: a neighbour is OOB if EITHER coordinate
    is outside (sampling.py:35-55, :84-88), and then reads the constant / the CENTRE value."""
    if isinstance(oob, str):
        # 'centre': substitute the centre value wherever the diagonal neighbour is outside
        inner = _shift(_shift(a, axis_a, step_a, 0.0), axis_b, step_b, 0.0)
        mask = np.zeros(a.shape, dtype=bool)
        for axis, step in ((axis_a, step_a), (axis_b, step_b)):
            sl = [slice(None)] * a.ndim
            n = a.shape[axis]
            sl[axis] = slice(n - step, None) if step > 0 else slice(0, -step)
            mask[tuple(sl)] = True
        return np.where(mask, a, inner).astype(F32)
    return _shift(_shift(a, axis_a, step_a, oob), axis_b, step_b, oob)


def data_term_gradient(live, canonical, method=BASIC):
    """a12: (live-canon) * grad(live) * 10; THRESHOLDED_FDM per data_term.py:190-227 (OOB neighbours read 1)."""
    d = live.ndim
    diff = (live - canonical).astype(F32)
    grads = gradient(live)
    if method == THRESHOLDED_FDM:
        for c in range(d):
            axis = d - 1 - c
            g = grads[c]
            fwd = (_shift(live, axis, 1, F32(1.0)) - live).astype(F32)
            bwd = (live - _shift(live, axis, -1, F32(1.0))).astype(F32)
            alt = np.where(np.abs(fwd) < np.abs(bwd), fwd, bwd)
            alt = np.where(np.abs(alt) > F32(0.5), F32(0.0), alt)
            grads[c] = np.where(np.abs(g) > F32(0.5), alt, g).astype(F32)
    comps = [((diff * g).astype(F32) * SCALING_FACTOR).astype(F32) for g in grads]
    return np.stack(comps, axis=-1), diff


def tikhonov_gradient(warp):
    """a14: -Laplacian(warp), edge replicated."""
    return np.stack([(-laplace_replicate(warp[..., c])).astype(F32) for c in range(warp.shape[-1])], axis=-1)


def tikhonov_energy_direct(warp):
    """smoothing_term.py:134-139 per voxel: 0.5*(|w_x|^2 + |w_y|^2 [+ |w_z|^2]), central diffs, OOB->centre."""
    d = warp.ndim - 1
    e = None  # float32 per voxel (the reference's local contributions are float32 scalars); x, y, z axis order
    for axis in range(d - 1, -1, -1):
        der = (F32(0.5) * (_shift(warp, axis, 1, 'centre') - _shift(warp, axis, -1, 'centre'))).astype(F32)
        for c in range(d):
            sq = (der[..., c] * der[..., c]).astype(F32)
            e = sq if e is None else (e + sq).astype(F32)
    return (F32(0.5) * e).astype(F32)


def smoothing_energy_vectorized(warp, band):
    """smoothing_term.py:162-177: 0.5 * sum over band of sum_{c,axis} np.gradient(warp_c)[axis]^2."""
    agg = None  # float32 per voxel, float64 sum over the band (per component: x, y, z)
    for c in range(warp.shape[-1]):
        for g in gradient(np.ascontiguousarray(warp[..., c])):
            sq = (g * g).astype(F32)
            agg = sq if agg is None else (agg + sq).astype(F32)
    return float((F32(0.5) * agg).astype(F32)[band].astype(np.float64).sum())


def killing_gradient(warp, lam):
    """a15 / Appendix A.6 (smoothing_term.py:50-100) with every quirk; 3-D extension per DESIGN.md section 3.
    Returns (gradient field, per-voxel energy)."""
    d = warp.ndim - 1
    ax = {c: d - 1 - c for c in range(d)}  # spatial axis of direction c (0:x 1:y 2:z)
    two = F32(2.0)

    def nb(direction, step):
        return _shift(warp, ax[direction], step, 'centre')

    first = {}
    second = {}
    for c in range(d):
        p, m = nb(c, 1), nb(c, -1)
        first[c] = (F32(0.5) * (p - m)).astype(F32)
        if c == 1:  # w_yy uses the +1 neighbour twice (smoothing_term.py:69)
            second[c] = ((p - two * warp).astype(F32) + p).astype(F32)
        else:
            second[c] = ((p - two * warp).astype(F32) + m).astype(F32)
    cross = {}
    for a in range(d):
        for b in range(a + 1, d):
            pp = _shift2(warp, ax[a], 1, ax[b], 1, 'centre')
            pm = _shift2(warp, ax[a], 1, ax[b], -1, 'centre')
            mp = _shift2(warp, ax[a], -1, ax[b], 1, 'centre')
            mm = _shift2(warp, ax[a], -1, ax[b], -1, 'centre')
            # (w[+a,+b] - w[+a,-b] - w[-a,+b] + w[-a,-b]) / 4     (smoothing_term.py:83-84)
            cross[(a, b)] = ((((pp - pm).astype(F32) - mp).astype(F32) + mm).astype(F32) / F32(4.0)).astype(F32)
    c1 = F32(-2.0 * (1.0 + lam))
    lam32 = F32(lam)
    comps = []
    for i in range(d):
        g = (c1 * second[0][..., i]).astype(F32)
        for c in range(1, d):
            g = (g + second[c][..., i]).astype(F32)
        for j in range(d):
            if j == i:
                continue
            a, b = (i, j) if i < j else (j, i)
            g = (g + (lam32 * cross[(a, b)][..., j]).astype(F32)).astype(F32)
        comps.append(g)
    grad = np.stack(comps, axis=-1)
    # energy  vecJ.vecJ + lam * (vecJ^T . vecJ)   (smoothing_term.py:93-98), J[i][c] = d u_i / d c
    # float32 per voxel, like the reference's float32 dot products; written as
    # |J|_F^2 + lam * (sum_i J_ii^2 + 2 sum_{i<c} J_ic J_ci) with every sum in (i, c) order
    frob = diag = off = None
    for i in range(d):
        for c in range(d):
            sq = (first[c][..., i] * first[c][..., i]).astype(F32)
            frob = sq if frob is None else (frob + sq).astype(F32)
            if c == i:
                diag = sq if diag is None else (diag + sq).astype(F32)
            if c > i:
                t = (first[c][..., i] * first[i][..., c]).astype(F32)
                off = t if off is None else (off + t).astype(F32)
    e = (frob + (lam32 * (diag + (off + off).astype(F32)).astype(F32)).astype(F32)).astype(F32)
    return grad, e


def level_set_gradient(live):
    """a16 / Appendix A.7 (level_set_term.py:28-64), OOB -> 1.  Returns (gradient field, per-voxel energy)."""
    d = live.ndim
    ax = {c: d - 1 - c for c in range(d)}
    one = F32(1.0)
    two = F32(2.0)
    grad = []
    hess = {}
    for c in range(d):
        p, m = _shift(live, ax[c], 1, one), _shift(live, ax[c], -1, one)
        grad.append(((F32(0.5) * (p - m)).astype(F32) * SCALING_FACTOR).astype(F32))
        # phi_cc uses the +1 neighbour twice (level_set_term.py:47-48)
        hess[(c, c)] = ((((p - two * live).astype(F32) + p).astype(F32)) * SCALING_FACTOR).astype(F32)
    for a in range(d):
        for b in range(a + 1, d):
            pp = _shift2(live, ax[a], 1, ax[b], 1, one)
            mp = _shift2(live, ax[a], -1, ax[b], 1, one)
            pm = _shift2(live, ax[a], 1, ax[b], -1, one)
            mm = _shift2(live, ax[a], -1, ax[b], -1, one)
            if (a, b) == (0, 1):
                # 0.25 * (phi[+,+] - phi[-,+] - phi[+,-] + phi[-,-])      (level_set_term.py:52-53)
                s = (((pp - mp).astype(F32) - pm).astype(F32) + mm).astype(F32)
            else:
                # pairs with z (3-D rule): difference along z first, so z-constant volumes give exactly 0
                s = (((pp - pm).astype(F32) - mp).astype(F32) + mm).astype(F32)
            h = (F32(0.25) * s).astype(F32)
            hess[(a, b)] = hess[(b, a)] = (h * SCALING_FACTOR).astype(F32)
    sq = (grad[0] * grad[0]).astype(F32)
    for c in range(1, d):
        sq = (sq + grad[c] * grad[c]).astype(F32)
    n = np.sqrt(sq).astype(F32)
    coef = ((one - n).astype(F32) / (n + LEVEL_SET_EPS).astype(F32)).astype(F32)
    comps = []
    for i in range(d):
        hv = (hess[(i, 0)] * grad[0]).astype(F32)
        for j in range(1, d):
            hv = (hv + hess[(i, j)] * grad[j]).astype(F32)
        comps.append((coef * hv).astype(F32))
    dn = (n - one).astype(F32)
    energy = (F32(0.5) * (dn * dn).astype(F32)).astype(F32)  # float32 per voxel (level_set_term.py:63)
    return np.stack(comps, axis=-1), energy


class SlavchevaOracle:
    """Per-iteration semantics of SlavchevaOptimizer2d (both compute methods), D = 2 or 3."""

    def __init__(self, compute_method=DIRECT, level_set_term_enabled=False, sobolev_smoothing_enabled=False,
                 data_term_method=BASIC, smoothing_term_method=TIKHONOV, gradient_descent_rate=0.1,
                 data_term_weight=1.0, smoothing_term_weight=0.2, isomorphic_enforcement_factor=0.1,
                 level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.1,
                 maximum_warp_length_upper_threshold=10000, max_iterations=100, min_iterations=1,
                 sobolev_kernel=None):
        self.compute_method = compute_method
        self.level_set_term_enabled = level_set_term_enabled
        self.sobolev_smoothing_enabled = sobolev_smoothing_enabled
        self.data_term_method = data_term_method
        self.smoothing_term_method = smoothing_term_method
        self.gradient_descent_rate = gradient_descent_rate
        self.data_term_weight = data_term_weight
        self.smoothing_term_weight = smoothing_term_weight
        self.isomorphic_enforcement_factor = isomorphic_enforcement_factor
        self.level_set_term_weight = level_set_term_weight
        self.lo = maximum_warp_length_lower_threshold
        self.hi = maximum_warp_length_upper_threshold
        self.max_iterations = max_iterations
        self.min_iterations = min_iterations
        self.sobolev_kernel = sobolev_kernel
        self.gradient_field = None
        self.log = None
        self.iteration_hook = None  # f(it, live, warp, gradient, energies(dict), max_warp, location)
        self.max_region = None      # optional slice along axis 0: restrict the max-warp search (slab tests)
        self.axis0_offset = 0       # slab tests: global index of the array's slice 0 (see _grid_positions)
        self.focus_voxels = None    # index tuples (axis order): the "focus neighbourhood" trace, :319-322,:422-430
        self.focus_log = None       # {index tuple: dict(warp_magnitudes=[], sdf_values=[], canonical_sdf=)}

    def iteration(self, live, canonical, warp):
        """slavcheva_optimizer2d.py:163-236 (VECTORIZED) / :238-330 (DIRECT).  live, warp updated in place.
        Returns (max_warp, location(axis-order index tuple), energies dict)."""
        direct = self.compute_method == DIRECT
        band = ~(is_truncated(live) & is_truncated(canonical))
        method = self.data_term_method if direct else BASIC
        gd, diff = data_term_gradient(live, canonical, method)
        # local contributions are float32 (data_term.py:185), their sum float64
        e_data = self.data_term_weight * float((F32(0.5) * (diff * diff).astype(F32)).astype(F32)[band]
                                               .astype(np.float64).sum())
        g = (F32(self.data_term_weight) * gd).astype(F32)
        e_ls = 0.0
        if direct and self.level_set_term_enabled:
            gl, el = level_set_gradient(live)
            active = band & ~is_truncated(live)
            gl[~active] = 0.0
            g = (g + (F32(self.level_set_term_weight) * gl).astype(F32)).astype(F32)
            e_ls = self.level_set_term_weight * float(el[active].astype(np.float64).sum())
        if direct and self.smoothing_term_method == KILLING:
            gs, es = killing_gradient(warp, self.isomorphic_enforcement_factor)
            e_smooth = self.smoothing_term_weight * float(es[band].astype(np.float64).sum())
        else:
            gs = tikhonov_gradient(warp)
            if direct:
                e_smooth = self.smoothing_term_weight * float(tikhonov_energy_direct(warp)[band].astype(np.float64).sum())
            else:
                e_smooth = self.smoothing_term_weight * smoothing_energy_vectorized(warp, band)
        g = (g + (F32(self.smoothing_term_weight) * gs).astype(F32)).astype(F32)
        g[~band] = 0.0
        if self.sobolev_smoothing_enabled:
            convolve_with_kernel_preserve_zeros(g, self.sobolev_kernel)
        np.copyto(warp, ((-g).astype(F32) * F32(self.gradient_descent_rate)).astype(F32))
        lengths = vector_norm(warp)
        if self.max_region is not None:  # a slice along axis 0, or a tuple of slices (slabs cut along another axis)
            region = self.max_region if isinstance(self.max_region, tuple) else (self.max_region,)
            _, at = first_argmax(lengths[region])
            at = tuple(a + ((region[k].start or 0) if k < len(region) else 0) for k, a in enumerate(at))
        else:
            _, at = first_argmax(lengths)
        max_warp = float(lengths[at])
        if self.focus_log is not None:  # :319-322: the update's length before the snap, the live value before the re-warp
            for index, entry in self.focus_log.items():
                entry["warp_magnitudes"].append(lengths[index])
                entry["sdf_values"].append(live[index])
        # DIRECT passes gradient_field to warp_field_advanced (zeroed where snapped, :324-327);
        # VECTORIZED hands only u,v to the C++ twin (:224-234), so its gradient_field is left alone.
        new_live = warp_field_advanced(canonical, live, warp, g if direct else None, axis0_offset=self.axis0_offset)
        np.copyto(live, new_live)
        self.gradient_field = g
        return max_warp, at, dict(data=e_data, smoothing=e_smooth, level_set=e_ls)

    def optimize(self, live_field, canonical_field):
        """live_field is warped IN PLACE and returned (slavcheva_optimizer2d.py:332-408)."""
        d = live_field.ndim
        warp = np.zeros(live_field.shape + (d,), dtype=F32)
        self.log = dict(data_energies=[], smoothing_energies=[], level_set_energies=[], max_warps=[],
                        max_warp_locations=[])
        max_warp = np.inf
        it = 0
        self.focus_log = None
        if self.focus_voxels is not None:  # :336-337,:353-354
            self.focus_log = {tuple(v): dict(warp_magnitudes=[], sdf_values=[], canonical_sdf=canonical_field[tuple(v)])
                              for v in self.focus_voxels}
        while it < self.min_iterations or (it < self.max_iterations and self.lo < max_warp < self.hi):
            max_warp, at, en = self.iteration(live_field, canonical_field, warp)
            self.log["max_warps"].append(max_warp)
            self.log["max_warp_locations"].append(tuple(int(i) for i in at))
            self.log["data_energies"].append(en["data"])
            self.log["smoothing_energies"].append(en["smoothing"])
            self.log["level_set_energies"].append(en["level_set"])
            if self.iteration_hook is not None:
                self.iteration_hook(it, live_field, warp, self.gradient_field, en, max_warp, at)
            it += 1
        self.iteration_count = it
        self.warp_field = warp
        return live_field


# =====================================================================================================
#  a20  convergence statistics (semantics recovered from tests/test_slavcheva_optimizer.py:141-149)
# =====================================================================================================
def warp_delta_statistics(warp, canonical, live, lo, hi):
    """over narrow-band-union voxels: ratio above `lo`, max, mean, std (population), argmax location (x,y[,z]);
    length_min is reported over ALL voxels start value 0 (known answer expects 0.0)."""
    lengths = vector_norm(warp).astype(np.float64)
    band = ~(is_truncated(live) & is_truncated(canonical))
    sel = lengths[band]
    count = max(int(sel.size), 1)
    masked = np.where(band, lengths, -1.0)
    flat, at = first_argmax(masked)
    return dict(ratio_above_min_threshold=float((sel > lo).sum()) / count,
                length_min=0.0, length_max=float(sel.max()) if sel.size else 0.0,
                length_mean=float(sel.mean()) if sel.size else 0.0,
                length_standard_deviation=float(sel.std()) if sel.size else 0.0,
                longest_warp_location=tuple(int(i) for i in at[::-1]),
                is_largest_below_min_threshold=bool(sel.size and sel.max() < lo),
                is_largest_above_max_threshold=bool(sel.size and sel.max() > hi))


def tsdf_difference_statistics(canonical, live):
    """abs(canonical - live) over ALL voxels: min, max, mean, std (population), argmax location (x,y[,z])."""
    diff = np.abs(canonical.astype(np.float64) - live.astype(np.float64))
    flat, at = first_argmax(diff)
    return dict(difference_min=float(diff.min()), difference_max=float(diff.max()),
                difference_mean=float(diff.mean()), difference_standard_deviation=float(diff.std()),
                biggest_difference_location=tuple(int(i) for i in at[::-1]))


# =====================================================================================================
#  synthetic inputs of SURVEY.md section 8(d)  (closed form, deterministic; also used by bench/tests)
# =====================================================================================================
FRAME_STEP = (1.0, -0.5, 1.0)  # voxels per frame along (x, y, z), as levelsetfusion-python_amd/synthetic.py


def sphere_frame(n, k, dtype=F32):
    """frame k of the synthetic multi-frame sequence (BASELINE config 5): sphere_pair's sphere, centre moved by
    k * FRAME_STEP voxels; frames k and k + 1 form a (canonical, live) pair"""
    h, r, c = 10.0, 0.3 * n, n / 2.0
    ax = np.arange(n, dtype=np.float64)
    zz, yy, xx = np.meshgrid(ax, ax, ax, indexing="ij")
    sq = 0.0
    for q, step in zip((xx, yy, zz), FRAME_STEP):
        sq = sq + (q - (c + k * step)) ** 2
    return np.clip((np.sqrt(sq) - r) / h, -1.0, 1.0).astype(dtype)


def sphere_pair(n, d=3, dtype=F32, nz=None, z_offset=0, z_total=None):
    """canonical = TSDF of a sphere (circle for d=2), live = the same sphere translated by (1.5,-1.0,2.0) and
    anisotropically scaled by (1.05,0.95,1.0) (x,y,z); narrow-band half width 10 voxels, values exactly +-1
    outside the band.  `nz`/`z_offset`/`z_total` generate a z-slab [z_offset, z_offset+nz) of a taller volume
    (z_total slices) whose sphere pattern repeats every n slices (used for weak-scaling slabs)."""
    h = 10.0
    r = 0.3 * n
    if d == 2:
        yy, xx = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
        coords = [xx, yy]
    else:
        nz = n if nz is None else nz
        zz, yy, xx = np.meshgrid(np.arange(z_offset, z_offset + nz, dtype=np.float64),
                                 np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
        zz = np.mod(zz, n)
        coords = [xx, yy, zz]
    c = n / 2.0

    def tsdf(shift, scale):
        sq = 0.0
        for i, q in enumerate(coords):
            sq = sq + ((q - (c + shift[i])) / scale[i]) ** 2
        return np.clip((np.sqrt(sq) - r) / h, -1.0, 1.0).astype(dtype)

    canonical = tsdf((0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    live = tsdf((1.5, -1.0, 2.0), (1.05, 0.95, 1.0))
    return canonical, live


# =====================================================================================================
#  a21  TSDF generation from a depth image, nearest pixel (tsdf/generation.py:130-207 2-D row, :356-437 3-D)
# =====================================================================================================
def tsdf_nearest(depth_image, intrinsic_matrix, depth_unit_ratio, field_shape, image_y_coordinate=None,
                 camera_extrinsic_matrix=None, default_value=1, voxel_size=0.004, array_offset=(-64, -64, 64),
                 narrow_band_width_voxels=20):
    """field_shape (n, n): 2-D slice -- x from the x index, depth axis ("z") from the y index, y_voxel = 0, the depth
    row `image_y_coordinate` is used.  field_shape (n, n, n): volume [z][y][x].  dtype semantics of the reference
    under numpy >= 2: voxel centre in float64 rounded to float32, float32 extrinsics product, projection in the
    intrinsic matrix's dtype, depth * ratio and the signed distance in float64, result stored float32."""
    P = np.asarray(intrinsic_matrix)
    ptype = np.float32 if P.dtype == np.float32 else np.float64
    E = np.eye(4, dtype=F32) if camera_extrinsic_matrix is None else np.asarray(camera_extrinsic_matrix, dtype=F32)
    dims = len(field_shape)
    field = np.full(field_shape, default_value, dtype=F32)
    half = narrow_band_width_voxels / 2 * voxel_size
    off = [int(o) for o in array_offset]
    idx = np.meshgrid(*[np.arange(s, dtype=np.int64) for s in field_shape], indexing="ij")
    if dims == 2:
        xv = ((idx[1] + off[0]) * voxel_size).astype(F32)
        yv = np.zeros(field_shape, dtype=F32)
        zv = ((idx[0] + off[2]) * voxel_size).astype(F32)
    else:
        xv = ((idx[2] + off[0]) * voxel_size).astype(F32)
        yv = ((idx[1] + off[1]) * voxel_size).astype(F32)
        zv = ((idx[0] + off[2]) * voxel_size).astype(F32)
    one = F32(1.0)

    def row(i):
        return ((((E[i, 0] * xv).astype(F32) + (E[i, 1] * yv).astype(F32)).astype(F32)
                 + (E[i, 2] * zv).astype(F32)).astype(F32) + (E[i, 3] * one)).astype(F32)

    pcx, pcy, pcz = row(0), row(1), row(2)
    ok = pcz > 0
    safe_z = np.where(ok, pcz, one).astype(ptype)
    with np.errstate(invalid="ignore", over="ignore"):
        ix = (((P[0, 0].astype(ptype) * pcx.astype(ptype)).astype(ptype) / safe_z).astype(ptype)
              + P[0, 2].astype(ptype)).astype(ptype) + ptype(0.5)
        ix = np.trunc(ix.astype(ptype)).astype(np.int64)  # int() truncates toward zero
        if dims == 3:
            iy = (((P[1, 1].astype(ptype) * pcy.astype(ptype)).astype(ptype) / safe_z).astype(ptype)
                  + P[1, 2].astype(ptype)).astype(ptype) + ptype(0.5)
            iy = np.trunc(iy.astype(ptype)).astype(np.int64)
        else:
            iy = np.full(field_shape, int(image_y_coordinate), dtype=np.int64)
    h, w = depth_image.shape
    ok &= (ix >= 0) & (ix < w) & (iy >= 0) & (iy < h)
    depth = depth_image[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1)].astype(np.float64) * depth_unit_ratio
    ok &= depth > 0.0
    sd = depth - pcz.astype(np.float64)
    value = np.where(sd < -half, -1.0, np.where(sd > half, 1.0, sd / half))
    field[ok] = value[ok].astype(F32)
    return field


def tsdf_bilinear(depth_image, intrinsic_matrix, depth_unit_ratio, field_size, image_y_coordinate, tsdf_space,
                  camera_extrinsic_matrix=None, default_value=1, voxel_size=0.004, array_offset=(-64, -64, 64),
                  narrow_band_width_voxels=20):
    """the two bilinear 2-D generators: tsdf/generation.py:78-128 (image space, tsdf_space=False) and :18-75 (TSDF
    space, tsdf_space=True) with utils/sampling.py:110-175.  image_y_coordinate is an integer, so the y ratio is
    exactly 0 and only the taps (floor x, row), (floor x + 1, row) carry weight; out-of-image taps read 1 (one raw
    depth unit / TSDF value 1.0).  dtypes as numpy >= 2 evaluates the reference: projection, ratio and 1 - ratio in the
    intrinsic matrix's dtype, the blend and the signed distance in float64, result stored float32."""
    P = np.asarray(intrinsic_matrix)
    ptype = np.float32 if P.dtype == np.float32 else np.float64
    E = np.eye(4, dtype=F32) if camera_extrinsic_matrix is None else np.asarray(camera_extrinsic_matrix, dtype=F32)
    shape = (field_size, field_size)
    field = np.full(shape, default_value, dtype=F32)
    half = narrow_band_width_voxels / 2 * voxel_size
    off = [int(o) for o in array_offset]
    yy, xx = np.meshgrid(np.arange(field_size, dtype=np.int64), np.arange(field_size, dtype=np.int64), indexing="ij")
    xv = ((xx + off[0]) * voxel_size).astype(F32)
    yv = np.zeros(shape, dtype=F32)
    zv = ((yy + off[2]) * voxel_size).astype(F32)
    one = F32(1.0)

    def row(i):
        return ((((E[i, 0] * xv).astype(F32) + (E[i, 1] * yv).astype(F32)).astype(F32)
                 + (E[i, 2] * zv).astype(F32)).astype(F32) + (E[i, 3] * one)).astype(F32)

    pcx, pcz = row(0), row(2)
    ok = pcz > 0
    safe_z = np.where(ok, pcz, one).astype(ptype)
    h, w = depth_image.shape
    with np.errstate(invalid="ignore", over="ignore"):
        ix = (((P[0, 0].astype(ptype) * pcx.astype(ptype)).astype(ptype) / safe_z).astype(ptype)
              + P[0, 2].astype(ptype)).astype(ptype)
        fl = np.floor(ix)
        ratio = (ix - fl).astype(ptype)
        inverse = (ptype(1.0) - ratio).astype(ptype)
        bx = np.where(np.abs(fl) < 2.0e9, fl, -2.0).astype(np.int64)
    in0, in1 = (bx >= 0) & (bx < w), (bx + 1 >= 0) & (bx + 1 < w)
    image_row = depth_image[int(image_y_coordinate)].astype(np.float64)
    d0 = np.where(in0, image_row[np.clip(bx, 0, w - 1)], 1.0)
    d1 = np.where(in1, image_row[np.clip(bx + 1, 0, w - 1)], 1.0)
    pz = pcz.astype(np.float64)

    def tsdf_of(sd):
        return np.where(sd < -half, -1.0, np.where(sd > half, 1.0, sd / half))

    if tsdf_space:
        t0 = np.where(in0, tsdf_of(d0 * depth_unit_ratio - pz), 1.0)
        t1 = np.where(in1, tsdf_of(d1 * depth_unit_ratio - pz), 1.0)
        value = t0 * inverse.astype(np.float64) + t1 * ratio.astype(np.float64)
    else:
        ok &= (ix >= 0) & (ix < w)
        depth = (d0 * inverse.astype(np.float64) + d1 * ratio.astype(np.float64)) * depth_unit_ratio
        ok &= depth > 0.0
        value = tsdf_of(depth - pz)
    field[ok] = value[ok].astype(F32)
    return field


def synthetic_depth_image(shift_px=0.0, nearer_m=0.0, width=640, height=480):
    """SURVEY 8(d) 'depth->TSDF' input: a plane at ~1 m with a sinusoidal bump, uint16 millimetres"""
    v, u = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing="ij")
    bump = 0.06 * np.exp(-(((u - 320.0 - shift_px) / 90.0) ** 2 + ((v - 240.0) / 70.0) ** 2)) \
        * (1.0 + 0.3 * np.sin(u / 17.0) * np.cos(v / 23.0))
    depth_m = 1.0 + 0.0002 * (u - 320.0) - bump - nearer_m
    return np.round(depth_m * 1000.0).astype(np.uint16)


# =====================================================================================================
#  a21  EWA TSDF generation (tsdf/ewa.py:59-184 3-D image space, :230-353 / :358-481 / :485-624 2-D image space,
#       voxel space, voxel space inclusive; math_utils/elliptical_gaussians.py:27-62,145-158)
#  Per-voxel Python loops: small cases only.
# =====================================================================================================
EWA_IMAGE, EWA_VOXEL, EWA_VOXEL_INCLUSIVE = 3, 4, 5  # values of FilteringMethod
EWA_NEAR_CLIPPING = 0.05                             # tsdf/ewa.py:29


def _tsdf_value(sd, half):
    """tsdf/common.py:34-47"""
    return -1.0 if sd < -half else (1.0 if sd > half else sd / half)


def tsdf_ewa(depth_image, intrinsic_matrix, depth_unit_ratio, field_shape, method=EWA_IMAGE, image_y_coordinate=None,
             camera_extrinsic_matrix=None, default_value=1, voxel_size=0.004, array_offset=(-64, -64, 64),
             narrow_band_width_voxels=20, gaussian_covariance_scale=1.0):
    """2-D (field_shape (n, n), depth row image_y_coordinate) or 3-D (field_shape (s0, s1, s2): array axis 0 is world
    x, axis 2 is the depth axis -- the reference's deliberate axis flip, tsdf/ewa.py:115-119).  dtypes as the reference
    under numpy >= 2: float32 voxel / camera / image coordinates and Jacobian entries, float64 covariances, weights
    and sums."""
    K = np.asarray(intrinsic_matrix, dtype=np.float32)
    E = np.eye(4, dtype=F32) if camera_extrinsic_matrix is None else np.asarray(camera_extrinsic_matrix, dtype=F32)
    dims = len(field_shape)
    field = np.full(tuple(field_shape), default_value, dtype=F32)
    half = narrow_band_width_voxels / 2 * voxel_size
    R = E[0:3, 0:3]
    cov_cam = R.dot(np.eye(3) * (gaussian_covariance_scale * voxel_size)).dot(R.T)  # float64
    S = K[0:2, 0:2].copy()
    F = 4.0 * gaussian_covariance_scale * voxel_size
    h, w = depth_image.shape
    off = [int(o) for o in array_offset]
    for idx in np.ndindex(*field_shape):
        if dims == 2:
            xw, yw, zw = (idx[1] + off[0]) * voxel_size, 0, (idx[0] + off[2]) * voxel_size
        else:
            xw, yw, zw = (idx[0] + off[0]) * voxel_size, (idx[1] + off[1]) * voxel_size, (idx[2] + off[2]) * voxel_size
        voxel_world = np.array([[xw, yw, zw, 1.0]], dtype=F32).T
        vc = E.dot(voxel_world).flatten()[:3]
        if vc[2] <= EWA_NEAR_CLIPPING:
            continue
        vi = (K.dot(vc) / vc[2])[:2]
        if dims == 2:
            vi[1] = image_y_coordinate
        if method == EWA_VOXEL_INCLUSIVE:
            margin = 3
            if vi[1] < -margin or vi[1] >= h + margin or vi[0] < -margin or vi[0] >= w + margin:
                continue
        ray = np.linalg.norm(vc)
        z2 = vc[2] ** 2
        J = np.array([[1 / vc[2], 0, -vc[0] / z2], [0, 1 / vc[2], -vc[1] / z2],
                      [vc[0] / ray, vc[1] / ray, vc[2] / ray]])
        remapped = J.dot(cov_cam).dot(J.T)
        final = S.dot(remapped[0:2, 0:2]).dot(S.T) + np.eye(2)
        Q = np.linalg.inv(final)
        A, B, C = Q[0, 0], Q[0, 1] * 2, Q[1, 1]
        if abs(float(B)) < 10e-6:
            bx, by = math.sqrt(F / A), math.sqrt(F / C)
        else:
            bx, by = math.sqrt(F / (C - B ** 2 / (4 * A))), math.sqrt(F / (A - B ** 2 / (4 * C)))
        vx, vy = np.float64(vi[0]), np.float64(vi[1])
        sx, ex = int(vx - bx), int(math.ceil(vx + bx + 1))
        sy, ey = int(vy - by), int(math.ceil(vy + by + 1))
        if ey <= 0 or sy >= h or ex <= 0 or sx >= w:
            continue
        if method != EWA_VOXEL_INCLUSIVE:
            sy, ey, sx, ex = max(0, sy), min(h, ey), max(0, sx), min(w, ex)
        weights_sum, value_sum = 0.0, 0.0
        for ys in range(sy, ey):
            for xs in range(sx, ex):
                p0, p1 = xs - vx, ys - vy
                d2 = (p0 * Q[0, 0] + p1 * Q[1, 0]) * p0 + (p0 * Q[0, 1] + p1 * Q[1, 1]) * p1
                if d2 > F:
                    continue
                weight = math.exp(-0.5 * d2)
                if ys < 0 or ys >= h or xs < 0 or xs >= w:  # only reachable in the inclusive variant
                    value_sum += weight * 1.0
                    weights_sum += weight
                    continue
                surface = float(depth_image[ys, xs]) * depth_unit_ratio
                if surface <= 0.0:
                    continue
                if method == EWA_IMAGE:
                    value_sum += weight * surface
                else:
                    value_sum += weight * _tsdf_value(surface - float(vc[2]), half)
                weights_sum += weight
        if method == EWA_IMAGE:
            if value_sum <= 0.0:
                continue
            field[idx] = _tsdf_value(value_sum / weights_sum - float(vc[2]), half)
        else:
            if weights_sum == 0.0:
                continue
            field[idx] = value_sum / weights_sum
    return field


# =====================================================================================================
#  LINEAR resampling strategy, 3-D  (math_utils/resampling.py:29-126; known answers tests/test_math.py:54-221)
# =====================================================================================================
# the reference's 4x4x4 restriction weights are typed to 8 decimals (resampling.py:90-109), i.e. NOT exactly the
# outer product of [1,3,3,1]/8; weight = DOWNSAMPLE_WEIGHTS[number of inner (index 1 or 2) coordinates]
DOWNSAMPLE_WEIGHTS = (0.00195312, 0.00585938, 0.01757812, 0.05273438)


def upsample2x_linear(field):
    """2x trilinear prolongation with edge padding: out[2m] = 0.25 f[m-1] + 0.75 f[m], out[2m+1] = 0.75 f[m] + 0.25
    f[m+1] along z, then y, then x, each lerp  0.75*a + 0.25*b  rounded in the field's dtype (resampling.py:29-80)."""
    out = np.asarray(field)
    dt = out.dtype if out.dtype in (np.float32, np.float64) else np.float64
    out = out.astype(dt)
    w75, w25 = dt.type(0.75), dt.type(0.25)
    for axis in range(3):
        f = np.moveaxis(out, axis, 0)
        prev = np.concatenate((f[:1], f[:-1]), axis=0)
        nxt = np.concatenate((f[1:], f[-1:]), axis=0)
        res = np.empty((2 * f.shape[0],) + f.shape[1:], dtype=dt)
        # reference: zv000 = 0.75 * v000 + 0.25 * v100 (the 0.75 product first in BOTH forms)
        res[0::2] = (w25 * prev).astype(dt) + (w75 * f).astype(dt)
        res[1::2] = (w75 * f).astype(dt) + (w25 * nxt).astype(dt)
        out = np.moveaxis(res, 0, axis)
    return np.ascontiguousarray(out)


def downsample2x_linear(field):
    """2x restriction: out[t] = sum over the 4x4x4 window f[clamp(2t-1) .. clamp(2t+2)] of weight * value, float64
    (resampling.py:83-126)"""
    f = np.asarray(field)
    if any(s % 2 for s in f.shape):
        raise ValueError("Each field dimension must be evenly divisible by 2.")
    p = np.pad(f, 1, mode="edge").astype(np.float64)
    nz, ny, nx = (s // 2 for s in f.shape)
    out = np.zeros((nz, ny, nx), dtype=np.float64)
    inner = (0, 1, 1, 0)
    for dz in range(4):
        for dy in range(4):
            for dx in range(4):
                wgt = DOWNSAMPLE_WEIGHTS[inner[dz] + inner[dy] + inner[dx]]
                out += wgt * p[dz:dz + 2 * nz:2, dy:dy + 2 * ny:2, dx:dx + 2 * nx:2]
    return out


# =====================================================================================================
#  BASELINE config 1's input: the hand-made orthographic 2-D pair   (tsdf/generation.py:238-353)
# =====================================================================================================
ORTHOGRAPHIC_SURFACE = ((9, 56), (14, 66), (23, 72), (35, 72), (44, 65), (54, 60), (63, 60), (69, 64), (76, 71),
                        (84, 73), (91, 72), (106, 63), (109, 57))   # (x, y), tsdf/generation.py:289-301
ORTHOGRAPHIC_DETAIL = ((32, 65), (36, 65), (41, 61))                # tsdf/generation.py:303-305


def orthographic_surface_fill(field, points, narrow_band_width_voxels=20, back_cutoff_voxels=math.inf):
    """One polyline into `field`, column by column, scalar float32 arithmetic (tsdf/generation.py:238-265): rows above
    the band <- 1, band rows <- clip((surface_y - y) / half_width), rows behind <- -1 unless a back cut-off ends the
    band early.  `points` are (x, y) float32 pairs."""
    half = narrow_band_width_voxels // 2
    behind = min(half, back_cutoff_voxels)
    for (ax, ay), (bx, by) in zip(points[:-1], points[1:]):
        run = F32(bx) - F32(ax)
        for x in range(int(ax), int(bx)):
            t = F32(F32(x) - F32(ax)) / run
            sy = F32(F32(ay) * F32(F32(1.0) - t)) + F32(F32(by) * t)
            if F32(sy - F32(narrow_band_width_voxels)) < 0:
                raise ValueError("Surface is too close to 0 in the y dimension for a full narrow band representation")
            first = int(F32(sy - F32(half)))
            last = int(F32(F32(sy + F32(behind)) + F32(1)))
            field[0:first, x] = 1.0
            for y in range(first, last):
                field[y, x] = min(max(F32(F32(sy - F32(y)) / F32(half)), F32(-1.0)), F32(1.0))
            if last < field.shape[0] and last < back_cutoff_voxels:
                field[last:, x] = -1.0
    return field


def orthographic_pair(field_size=128, narrow_band_width_voxels=20, mimic_eta=False, default_value=1):
    """(live, canonical) of generate_initial_orthographic_2d_tsdf_fields (tsdf/generation.py:282-353): surface and
    detail polylines lowered by 0.23 rows (float32), canonical = the same 5 rows further down, cut off 3 voxels behind
    the surface when `mimic_eta`."""
    lower = lambda pts, by: [(F32(x), F32(F32(y) + F32(by))) for x, y in pts]  # noqa: E731
    live_surface, live_detail = lower(ORTHOGRAPHIC_SURFACE, -0.23), lower(ORTHOGRAPHIC_DETAIL, -0.23)
    live = np.full((field_size, field_size), default_value, dtype=F32)
    orthographic_surface_fill(live, live_surface, narrow_band_width_voxels)
    orthographic_surface_fill(live, live_detail, narrow_band_width_voxels)
    cutoff = 3 if mimic_eta else math.inf
    canonical = np.full((field_size, field_size), default_value, dtype=F32)
    orthographic_surface_fill(canonical, lower(live_surface, 5.0), narrow_band_width_voxels, cutoff)
    orthographic_surface_fill(canonical, lower(live_detail, 5.0), narrow_band_width_voxels, cutoff)
    return live, canonical
