// Field utilities of liblsf_hip.so: layout conversion, resampling under a warp (a1-a3), np.gradient packing
// (a4), pyramid restrict / prolong (a5, a6), separable convolution passes (a9, a10), convergence statistics
// (a20).  Reference citations are in include/lsf_hip.h next to each entry point.
#include <cstring>

#include "lsf_device.h"

using namespace lsf;

// =====================================================================================================
//  layout helpers
// =====================================================================================================
__global__ __launch_bounds__(kBlock) void deinterleave_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              long long n, int channels) {
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < channels; ++c) out[(long long)c * n + i] = in[i * channels + c];
}

__global__ __launch_bounds__(kBlock) void interleave_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long long n, int channels) {
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < channels; ++c) out[i * channels + c] = in[(long long)c * n + i];
}

extern "C" int lsf_deinterleave(const float* interleaved, float* planar, int64_t n_voxels, int32_t channels,
                                void* stream) {
    (void)hipGetLastError();
    if (!interleaved || !planar || n_voxels < 0 || channels < 1 || channels > 4) return LSF_ERR_BAD_ARGUMENT;
    if (n_voxels == 0) return 0;
    unsigned blocks = (unsigned)((n_voxels + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(deinterleave_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), interleaved, planar,
                       (long long)n_voxels, channels);
    return launch_status();
}

extern "C" int lsf_interleave(const float* planar, float* interleaved, int64_t n_voxels, int32_t channels,
                              void* stream) {
    (void)hipGetLastError();
    if (!interleaved || !planar || n_voxels < 0 || channels < 1 || channels > 4) return LSF_ERR_BAD_ARGUMENT;
    if (n_voxels == 0) return 0;
    unsigned blocks = (unsigned)((n_voxels + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(interleave_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), planar, interleaved,
                       (long long)n_voxels, channels);
    return launch_status();
}

// =====================================================================================================
//  slab halo staging: pack the boundary slices of a scalar field + a planar vector field into one contiguous message
//  per neighbour (and unpack received ones into the halo slices) -- one launch per direction instead of 4-8 copies
// =====================================================================================================
// message layout per side: [channel][h][ny][nx], channel 0 = scalar field, 1..planes = vector planes.
// side 0 = lower neighbour, side 1 = upper neighbour; z0[side] = first slice to copy from / into.
__global__ __launch_bounds__(kBlock) void halo_copy_kernel(float* __restrict__ scalar, float* __restrict__ planar,
                                                           float* __restrict__ msg_lo, float* __restrict__ msg_hi,
                                                           long long plane, int slice, int h, int planes, int z_lo,
                                                           int z_hi, int unpack) {
    const int side = blockIdx.z;
    float* msg = side == 0 ? msg_lo : msg_hi;
    if (!msg) return;
    const int z0 = side == 0 ? z_lo : z_hi;
    const int channel = blockIdx.y;  // 0 .. planes
    const long long per_channel = (long long)h * slice;
    float* field = channel == 0 ? scalar : planar + (long long)(channel - 1) * plane;
    if (!field) return;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < per_channel;
         i += (long long)gridDim.x * kBlock) {
        const long long f = (long long)z0 * slice + i;  // h consecutive slices are contiguous
        if (unpack) field[f] = msg[channel * per_channel + i];
        else msg[channel * per_channel + i] = field[f];
    }
}

extern "C" int lsf_halo_copy(float* scalar, float* planar, float* msg_lo, float* msg_hi, const lsf_grid* grid,
                             int32_t planes, int32_t halo, int32_t z_lo, int32_t z_hi, int32_t unpack, void* stream) {
    if (int e = check_grid(grid)) return e;
    if ((!scalar && !planar) || (!msg_lo && !msg_hi) || halo < 1 || planes < 0 || planes > 3)
        return LSF_ERR_BAD_ARGUMENT;
    if ((msg_lo && (z_lo < 0 || z_lo + halo > grid->nz)) || (msg_hi && (z_hi < 0 || z_hi + halo > grid->nz)))
        return LSF_ERR_BAD_ARGUMENT;
    const int slice = grid->ny * grid->nx;
    const long long per_channel = (long long)halo * slice;
    unsigned bx = (unsigned)((per_channel + kBlock - 1) / kBlock);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(halo_copy_kernel, dim3(bx, planes + 1, 2), dim3(kBlock), 0, as_stream(stream), scalar, planar,
                       msg_lo, msg_hi, (long long)grid->nz * slice, slice, halo, planes, z_lo, z_hi, unpack);
    return launch_status();
}

// =====================================================================================================
//  a1/a2  warp_field / warp_field_replacement
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(kBlock) void warp_field_kernel(const float* __restrict__ field,
                                                            const float* __restrict__ warp, float* __restrict__ out,
                                                            Grid g, float oob) {
    for_each_voxel(g, [&](int x, int y, int z) {
    long long i = vidx(g, x, y, z);
    float px = (float)x + warp[i];
    float py = (float)y + warp[g.plane + i];
    float pz = D == 3 ? (float)(z + g.z_global_offset) + warp[2 * g.plane + i] : 0.0f;
    out[i] = sample_linear<D>(field, g, px, py, pz, oob);
    });
}

extern "C" int lsf_warp_field(const float* field, const float* warp_planar, float* out, const lsf_grid* grid,
                              float oob_value, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!field || !warp_planar || !out) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    if (grid->dims == 2)
        hipLaunchKernelGGL(warp_field_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), field,
                           warp_planar, out, g, oob_value);
    else
        hipLaunchKernelGGL(warp_field_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), field,
                           warp_planar, out, g, oob_value);
    return launch_status();
}

// =====================================================================================================
//  a3  warp_field_advanced
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(kBlock) void warp_field_advanced_kernel(const float* __restrict__ canonical,
                                                                     const float* __restrict__ live,
                                                                     float* __restrict__ warp,
                                                                     float* __restrict__ gradient,
                                                                     float* __restrict__ new_live, Grid g, int flags) {
    for_each_voxel(g, [&](int x, int y, int z) {
    long long i = vidx(g, x, y, z);
    float original = live[i];
    bool skip = false;
    if (flags & 1) skip = skip || (fabsf(original) == 1.0f && fabsf(canonical[i]) == 1.0f);
    if (flags & 2) skip = skip || (original == 1.0f);
    if (skip) {
        new_live[i] = original;
        return;
    }
    float px = (float)x + warp[i];
    float py = (float)y + warp[g.plane + i];
    float pz = D == 3 ? (float)(z + g.z_global_offset) + warp[2 * g.plane + i] : 0.0f;
    float v = sample_linear<D>(live, g, px, py, pz, (flags & 4) ? original : 1.0f);
    if (1.0f - fabsf(v) < 1e-6f) {
        v = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v);
#pragma unroll
        for (int c = 0; c < D; ++c) {
            warp[c * g.plane + i] = 0.0f;
            if (gradient) gradient[c * g.plane + i] = 0.0f;
        }
    }
    new_live[i] = v;
    });
}

extern "C" int lsf_warp_field_advanced(const float* canonical, const float* live, float* warp_planar,
                                       float* gradient_planar, float* new_live, const lsf_grid* grid,
                                       int32_t flags, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!canonical || !live || !warp_planar || !new_live || live == new_live) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    if (grid->dims == 2)
        hipLaunchKernelGGL(warp_field_advanced_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream),
                           canonical, live, warp_planar, gradient_planar, new_live, g, flags);
    else
        hipLaunchKernelGGL(warp_field_advanced_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream),
                           canonical, live, warp_planar, gradient_planar, new_live, g, flags);
    return launch_status();
}

// =====================================================================================================
//  a4  np.gradient of live, packed as float4 (live, gx, gy, gz)
// =====================================================================================================
__device__ inline float np_gradient_axis(const float* __restrict__ f, long long i, int coord, int n, long long stride) {
    if (n == 1) return 0.0f;
    if (coord == 0) return f[i + stride] - f[i];
    if (coord == n - 1) return f[i] - f[i - stride];
    return (f[i + stride] - f[i - stride]) * 0.5f;
}

template <int D>
__global__ __launch_bounds__(kBlock) void pack_live_gradient_kernel(const float* __restrict__ live,
                                                                    float4* __restrict__ packed, Grid g) {
    for_each_voxel(g, [&](int x, int y, int z) {
    long long i = vidx(g, x, y, z);
    float4 r;
    r.x = live[i];
    r.y = np_gradient_axis(live, i, x, g.nx, 1);
    r.z = np_gradient_axis(live, i, y, g.ny, g.nx);
    r.w = D == 3 ? np_gradient_axis(live, i, z, g.nz, (long long)g.ny * g.nx) : 0.0f;
    packed[i] = r;
    });
}

extern "C" int lsf_pack_live_gradient(const float* live, float* packed4, const lsf_grid* grid, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!live || !packed4) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    if (grid->dims == 2)
        hipLaunchKernelGGL(pack_live_gradient_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), live,
                           reinterpret_cast<float4*>(packed4), g);
    else
        hipLaunchKernelGGL(pack_live_gradient_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream), live,
                           reinterpret_cast<float4*>(packed4), g);
    return launch_status();
}

// =====================================================================================================
//  a5  restrict (2^D block mean per channel)     a6  prolong (repeat)
// =====================================================================================================
// sum order: 2-D ((a00 + a01) + a10) + a11 (numpy's float32 reduction of the reshaped 4-vector);
// 3-D (s(z0) + s(z1)) * 0.125 with s the 2-D block sums (oracle.restrict_mean).
template <int D, int C>
__global__ __launch_bounds__(kBlock) void restrict_mean_kernel(const float* __restrict__ fine,
                                                               float* __restrict__ coarse, Grid gc, int fnx, int fny) {
    for_each_voxel(gc, [&](int x, int y, int z) {
    long long o = vidx(gc, x, y, z);
    const long long row = fnx, slice = (long long)fnx * fny;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        long long b = (((long long)(D == 3 ? 2 * z : 0) * fny + 2 * y) * fnx + 2 * x);
        auto at = [&](long long j) { return fine[j * C + c]; };
        float s0 = ((at(b) + at(b + 1)) + at(b + row)) + at(b + row + 1);
        float r;
        if (D == 2) {
            r = s0 * 0.25f;
        } else {
            long long b1 = b + slice;
            float s1 = ((at(b1) + at(b1 + 1)) + at(b1 + row)) + at(b1 + row + 1);
            r = (s0 + s1) * 0.125f;
        }
        coarse[o * C + c] = r;
    }
    });
}

extern "C" int lsf_restrict_mean(const float* fine, float* coarse, const lsf_grid* fine_grid, int32_t channels,
                                 void* stream) {
    if (int e = check_grid(fine_grid)) return e;
    if (!fine || !coarse || (channels != 1 && channels != 4)) return LSF_ERR_BAD_ARGUMENT;
    if ((fine_grid->nx & 1) || (fine_grid->ny & 1) || (fine_grid->dims == 3 && (fine_grid->nz & 1)))
        return LSF_ERR_BAD_DIMS;
    lsf_grid cg = *fine_grid;
    cg.nx /= 2;
    cg.ny /= 2;
    if (cg.dims == 3) cg.nz /= 2;
    cg.z_begin = 0;
    cg.z_end = cg.nz;
    Grid gc = make_grid(&cg);
    Tiling t = make_tiling(gc);
    if (t.total == 0) return 0;
    hipStream_t s = as_stream(stream);
#define LSF_LAUNCH_RESTRICT(D, C)                                                                             \
    hipLaunchKernelGGL((restrict_mean_kernel<D, C>), dim3(launch_blocks(t.total)), dim3(kBlock), 0, s, fine, coarse, gc, \
                       fine_grid->nx, fine_grid->ny)
    if (fine_grid->dims == 2) {
        if (channels == 1) LSF_LAUNCH_RESTRICT(2, 1); else LSF_LAUNCH_RESTRICT(2, 4);
    } else {
        if (channels == 1) LSF_LAUNCH_RESTRICT(3, 1); else LSF_LAUNCH_RESTRICT(3, 4);
    }
#undef LSF_LAUNCH_RESTRICT
    return launch_status();
}

template <int D>
__global__ __launch_bounds__(kBlock) void prolong_repeat_kernel(const float* __restrict__ coarse,
                                                                float* __restrict__ fine, Grid gf) {
    for_each_voxel(gf, [&](int x, int y, int z) {
    const int cnx = gf.nx / 2, cny = gf.ny / 2, cnz = D == 3 ? gf.nz / 2 : 1;
    const long long cplane = (long long)cnx * cny * cnz;
    long long ci = ((long long)(D == 3 ? z / 2 : 0) * cny + y / 2) * cnx + x / 2;
    long long fi = vidx(gf, x, y, z);
#pragma unroll
    for (int c = 0; c < D; ++c) fine[c * gf.plane + fi] = coarse[c * cplane + ci];
    });
}

extern "C" int lsf_prolong_repeat(const float* coarse_planar, float* fine_planar, const lsf_grid* fine_grid,
                                  void* stream) {
    if (int e = check_grid(fine_grid)) return e;
    if (!coarse_planar || !fine_planar) return LSF_ERR_BAD_ARGUMENT;
    if ((fine_grid->nx & 1) || (fine_grid->ny & 1) || (fine_grid->dims == 3 && (fine_grid->nz & 1)))
        return LSF_ERR_BAD_DIMS;
    Grid g = make_grid(fine_grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    if (fine_grid->dims == 2)
        hipLaunchKernelGGL(prolong_repeat_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream),
                           coarse_planar, fine_planar, g);
    else
        hipLaunchKernelGGL(prolong_repeat_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, as_stream(stream),
                           coarse_planar, fine_planar, g);
    return launch_status();
}

// =====================================================================================================
//  LINEAR resampling strategy (3-D): math_utils/resampling.py:29-126
// =====================================================================================================
// prolongation: out[2m] = 0.25 f[m-1] + 0.75 f[m], out[2m+1] = 0.75 f[m] + 0.25 f[m+1] (edge clamped), applied along
// z, then y, then x with a float32 rounding after every lerp (oracle.upsample2x_linear)
struct UpTap {
    int a, b;      // clamped source indices
    float wa, wb;  // weights
};

__device__ inline UpTap up_tap(int o, int n) {
    UpTap t;
    const int m = o >> 1;
    if (o & 1) { t.a = m; t.b = min(m + 1, n - 1); t.wa = 0.75f; t.wb = 0.25f; }
    else       { t.a = max(m - 1, 0); t.b = m; t.wa = 0.25f; t.wb = 0.75f; }
    return t;
}

template <int C>
__global__ __launch_bounds__(kBlock) void upsample2x_linear_kernel(const float* __restrict__ coarse,
                                                                   float* __restrict__ fine, Grid gf) {
    const int cnx = gf.nx / 2, cny = gf.ny / 2, cnz = gf.nz / 2;
    for_each_voxel(gf, [&](int x, int y, int z) {
        const UpTap tz = up_tap(z, cnz), ty = up_tap(y, cny), tx = up_tap(x, cnx);
        const long long o = vidx(gf, x, y, z);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            auto at = [&](int zz, int yy, int xx) { return coarse[(((long long)zz * cny + yy) * cnx + xx) * C + c]; };
            // z lerp for the four (y, x) source combinations
            const float v00 = tz.wa * at(tz.a, ty.a, tx.a) + tz.wb * at(tz.b, ty.a, tx.a);
            const float v01 = tz.wa * at(tz.a, ty.a, tx.b) + tz.wb * at(tz.b, ty.a, tx.b);
            const float v10 = tz.wa * at(tz.a, ty.b, tx.a) + tz.wb * at(tz.b, ty.b, tx.a);
            const float v11 = tz.wa * at(tz.a, ty.b, tx.b) + tz.wb * at(tz.b, ty.b, tx.b);
            const float u0 = ty.wa * v00 + ty.wb * v10;
            const float u1 = ty.wa * v01 + ty.wb * v11;
            fine[o * C + c] = tx.wa * u0 + tx.wb * u1;
        }
    });
}

// restriction: 4x4x4 window f[clamp(2t-1) .. clamp(2t+2)], weight by the number of inner coordinates; the reference's
// weights are its 8-decimal literals (resampling.py:90-109), accumulated in float64
template <int C>
__global__ __launch_bounds__(kBlock) void downsample2x_linear_kernel(const float* __restrict__ fine,
                                                                     float* __restrict__ coarse, Grid gc, int fnx,
                                                                     int fny, int fnz) {
    const double w[4] = {0.00195312, 0.00585938, 0.01757812, 0.05273438};
    for_each_voxel(gc, [&](int x, int y, int z) {
        const long long o = vidx(gc, x, y, z);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            double acc = 0.0;
            for (int dz = 0; dz < 4; ++dz) {
                const int zz = min(max(2 * z - 1 + dz, 0), fnz - 1);
                for (int dy = 0; dy < 4; ++dy) {
                    const int yy = min(max(2 * y - 1 + dy, 0), fny - 1);
                    const long long row = ((long long)zz * fny + yy) * fnx;
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) {
                        const int xx = min(max(2 * x - 1 + dx, 0), fnx - 1);
                        const int inner = ((dz == 1 || dz == 2) ? 1 : 0) + ((dy == 1 || dy == 2) ? 1 : 0) +
                                          ((dx == 1 || dx == 2) ? 1 : 0);
                        acc += w[inner] * (double)fine[(row + xx) * C + c];
                    }
                }
            }
            coarse[o * C + c] = (float)acc;
        }
    });
}

extern "C" int lsf_upsample2x_linear(const float* coarse, float* fine, const lsf_grid* fine_grid, int32_t channels,
                                     void* stream) {
    if (int e = check_grid(fine_grid)) return e;
    if (!coarse || !fine || (channels != 1 && channels != 4)) return LSF_ERR_BAD_ARGUMENT;
    if (fine_grid->dims != 3 || (fine_grid->nx & 1) || (fine_grid->ny & 1) || (fine_grid->nz & 1)) return LSF_ERR_BAD_DIMS;
    Grid g = make_grid(fine_grid);
    Tiling t = make_tiling(g);
    if (t.total == 0) return 0;
    if (channels == 1)
        hipLaunchKernelGGL(upsample2x_linear_kernel<1>, dim3(launch_blocks(t.total)), dim3(kBlock), 0,
                           as_stream(stream), coarse, fine, g);
    else
        hipLaunchKernelGGL(upsample2x_linear_kernel<4>, dim3(launch_blocks(t.total)), dim3(kBlock), 0,
                           as_stream(stream), coarse, fine, g);
    return launch_status();
}

extern "C" int lsf_downsample2x_linear(const float* fine, float* coarse, const lsf_grid* fine_grid, int32_t channels,
                                       void* stream) {
    if (int e = check_grid(fine_grid)) return e;
    if (!coarse || !fine || (channels != 1 && channels != 4)) return LSF_ERR_BAD_ARGUMENT;
    if (fine_grid->dims != 3 || (fine_grid->nx & 1) || (fine_grid->ny & 1) || (fine_grid->nz & 1)) return LSF_ERR_BAD_DIMS;
    lsf_grid cg = *fine_grid;
    cg.nx /= 2; cg.ny /= 2; cg.nz /= 2;
    cg.z_begin = 0; cg.z_end = cg.nz;
    Grid gc = make_grid(&cg);
    Tiling t = make_tiling(gc);
    if (t.total == 0) return 0;
    if (channels == 1)
        hipLaunchKernelGGL(downsample2x_linear_kernel<1>, dim3(launch_blocks(t.total)), dim3(kBlock), 0,
                           as_stream(stream), fine, coarse, gc, fine_grid->nx, fine_grid->ny, fine_grid->nz);
    else
        hipLaunchKernelGGL(downsample2x_linear_kernel<4>, dim3(launch_blocks(t.total)), dim3(kBlock), 0,
                           as_stream(stream), fine, coarse, gc, fine_grid->nx, fine_grid->ny, fine_grid->nz);
    return launch_status();
}

// =====================================================================================================
//  a9/a10  one separable-convolution pass
// =====================================================================================================
struct Taps {
    double k[LSF_MAX_KERNEL_TAPS];
    int n;
};

// One pass = one launch.  Every block stages its tile PLUS the taps' reach along the filter axis in LDS (zero outside
// the array = np.convolve's zero padding), so each input is fetched from global memory once per block instead of once
// per tap.  The tile is stored as float64: the conversion happens once per input, not once per tap, and the tap loop
// is one ds_read_b64 + v_mul_f64 + v_add_f64 (the pass is otherwise FP64-issue bound next to its HBM time).
// Accumulation in float64 in tap order, one rounding to float32 per pass, exactly as the oracle / the reference's
// float64-kernel path.
constexpr int kConvRun = 32;   // outputs along the filter axis per block (strided axes): 8 per thread
constexpr int kConvRows = 16;  // rows per block for the x pass: 4 outputs per thread

// filter along y (AXIS 1) or z (AXIS 2): tile = 64 x-lanes x kConvRun positions along the axis, fixed other coordinate
template <int AXIS>
__global__ __launch_bounds__(kBlock) void convolve_strided_kernel(const float* __restrict__ in,
                                                                  float* __restrict__ out,
                                                                  const float* __restrict__ mask_src, Grid g,
                                                                  Taps taps, lsf_gate gate) {
    if (gate_closed(gate)) return;
    // dynamic LDS sized to the ACTUAL reach: (kConvRun + n - 1) x 64 doubles (19 KiB for 7 taps -> 8 blocks per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double (*tile)[kTileX] = reinterpret_cast<double (*)[kTileX]>(lds_raw);
    const int lx = threadIdx.x & (kTileX - 1), lr = threadIdx.x / kTileX;  // lr in 0..3
    const int x = blockIdx.x * kTileX + lx;
    const int len = AXIS == 1 ? g.ny : g.nz;
    const int stride = AXIS == 1 ? g.nx : g.nx * g.ny;
    // blockIdx.y enumerates (run along the axis, other coordinate); blockIdx.z = plane
    const int runs = AXIS == 1 ? (g.ny + kConvRun - 1) / kConvRun : (g.z_end - g.z_begin + kConvRun - 1) / kConvRun;
    const int run = blockIdx.y % runs, other = blockIdx.y / runs;
    const int a0 = (AXIS == 1 ? 0 : g.z_begin) + run * kConvRun;
    const int a_end = AXIS == 1 ? g.ny : g.z_end;
    const int count = min(kConvRun, a_end - a0);
    const int y_or_z = AXIS == 1 ? g.z_begin + other : other;  // AXIS 1: other = z ; AXIS 2: other = y
    const long long base = (long long)blockIdx.z * g.plane;
    const int fixed = AXIS == 1 ? y_or_z * g.nx * g.ny : y_or_z * g.nx;
    const int c = (taps.n - 1) / 2, lo = taps.n - 1 - c;  // np.convolve(..., 'same') starts at (n - 1) / 2
    const int rows = count + taps.n - 1;
    if (x < g.nx) {
        for (int r = lr; r < rows; r += kBlock / kTileX) {
            const int a = a0 - lo + r;
            const int ac = min(max(a, 0), len - 1);
            const float v = in[base + fixed + ac * stride + x];
            tile[r][lx] = (a >= 0 && a < len) ? (double)v : 0.0;
        }
    }
    __syncthreads();
    if (x >= g.nx) return;
    for (int m = lr; m < count; m += kBlock / kTileX) {
        double acc = 0.0;
        for (int j = 0; j < taps.n; ++j) acc = acc + taps.k[j] * tile[m + taps.n - 1 - j][lx];
        float r = (float)acc;
        const long long o = base + fixed + (a0 + m) * stride + x;
        if (mask_src && fabsf(mask_src[o]) < 1e-6f) r = 0.0f;
        out[o] = r;
    }
}

// filter along x: tile = (64 + reach) x kConvRows rows of one z-slice
__global__ __launch_bounds__(kBlock) void convolve_x_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const float* __restrict__ mask_src, Grid g, Taps taps,
                                                            lsf_gate gate) {
    if (gate_closed(gate)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int pitch = kTileX + taps.n - 1;
    double* tile = reinterpret_cast<double*>(lds_raw);  // [kConvRows][pitch]
    const int lx = threadIdx.x & (kTileX - 1), ly = threadIdx.x / kTileX;
    const int tiles_y = (g.ny + kConvRows - 1) / kConvRows;
    const int x0 = blockIdx.x * kTileX;
    const int y0 = (blockIdx.y % tiles_y) * kConvRows;
    const int z = g.z_begin + blockIdx.y / tiles_y;
    const long long base = (long long)blockIdx.z * g.plane;
    const int c = (taps.n - 1) / 2, lo = taps.n - 1 - c;  // np.convolve(..., 'same') starts at (n - 1) / 2
    const int cols = kTileX + taps.n - 1;
    for (int r = ly; r < kConvRows; r += kBlock / kTileX) {
        const int y = y0 + r;
        if (y >= g.ny) break;
        const int row = vidx(g, 0, y, z);
        for (int q = lx; q < cols; q += kTileX) {
            const int xx = x0 - lo + q;
            const int xc = min(max(xx, 0), g.nx - 1);
            const float v = in[base + row + xc];
            tile[r * pitch + q] = (xx >= 0 && xx < g.nx) ? (double)v : 0.0;
        }
    }
    __syncthreads();
    const int x = x0 + lx;
    if (x >= g.nx) return;
    for (int r = ly; r < kConvRows; r += kBlock / kTileX) {
        const int y = y0 + r;
        if (y >= g.ny) break;
        double acc = 0.0;
        for (int j = 0; j < taps.n; ++j) acc = acc + taps.k[j] * tile[r * pitch + lx + taps.n - 1 - j];
        float res = (float)acc;
        const long long o = base + vidx(g, x, y, z);
        if (mask_src && fabsf(mask_src[o]) < 1e-6f) res = 0.0f;
        out[o] = res;
    }
}

// ---- register-window variants for the common tap counts (3, 5, 7, 9): no LDS, no barrier, compile-time taps --------
// The tiled kernels above reach ~2 TB/s at 256^3 (load phase, barrier, compute phase in small blocks; taps indexed
// dynamically).  Here every thread owns its outputs outright:
//   y / z pass: a thread marches along the filter axis over kMarch outputs of one (x, other) column with a rolling
//     window of NT doubles; a wave's loads and stores are 256-byte rows (x fastest); the NT - 1 halo rows of a run are
//     re-read by the neighbouring run (x 1.19 at kMarch = 32, from L2);
//   x pass: a thread computes 4 consecutive outputs from three aligned float4 loads (previous, own, next quad).
// Same arithmetic as the tiled kernels: float64 products and sums in tap order, one rounding per pass.
constexpr int kMarch = 32;

template <int AXIS, int NT, bool MASK, bool FMA, bool UPDATE = false>
__device__ inline void convolve_march(const float* __restrict__ in, float* __restrict__ out,
                                      const float* __restrict__ mask_src, const Grid& g, const TapsN<NT>& taps,
                                      const lsf_gate& gate, float* __restrict__ warp = nullptr, float rate = 0.0f) {
    if (gate_closed(gate)) return;
    const int lx = threadIdx.x & (kTileX - 1), w = threadIdx.x / kTileX;
    const int x = blockIdx.x * kTileX + lx;
    const int len = AXIS == 1 ? g.ny : g.nz;
    const int stride = AXIS == 1 ? g.nx : g.nx * g.ny;
    const int n_other = AXIS == 1 ? g.z_end - g.z_begin : g.ny;
    const int runs = AXIS == 1 ? (g.ny + kMarch - 1) / kMarch : (g.z_end - g.z_begin + kMarch - 1) / kMarch;
    const int run = blockIdx.y % runs, other = (blockIdx.y / runs) * (kBlock / kTileX) + w;
    if (x >= g.nx || other >= n_other) return;
    const int a0 = (AXIS == 1 ? 0 : g.z_begin) + run * kMarch;
    const int a_end = AXIS == 1 ? g.ny : g.z_end;
    const int count = min(kMarch, a_end - a0);
    const int fixed = AXIS == 1 ? (g.z_begin + other) * g.nx * g.ny : other * g.nx;
    const float* __restrict__ src = in + (long long)blockIdx.z * g.plane + fixed + x;
    float* __restrict__ dst = out + (long long)blockIdx.z * g.plane + fixed + x;
    const float* __restrict__ msk = MASK ? mask_src + (long long)blockIdx.z * g.plane + fixed + x : nullptr;
    float* __restrict__ moved = UPDATE ? warp + (long long)blockIdx.z * g.plane + fixed + x : nullptr;
    constexpr int c = NT / 2, lo = NT - 1 - c;
    float v[kMarch + NT - 1];  // v[q] = in[a0 - lo + q], zero outside the array (np.convolve's zero padding)
#pragma unroll
    for (int q = 0; q < kMarch + NT - 1; ++q) {
        const int a = a0 - lo + q;
        const float t = src[min(max(a, 0), len - 1) * stride];  // < plane < 2^31 (check_grid)
        v[q] = (a >= 0 && a < len && q < count + NT - 1) ? t : 0.0f;
    }
    float wv[UPDATE ? kMarch : 1];  // the run's warp values, fetched with the inputs (a load inside the loop below would
    if (UPDATE) {                    // put a memory round trip in front of every store)
#pragma unroll
        for (int m = 0; m < kMarch; ++m) wv[m] = moved[(a0 + min(m, count - 1)) * stride];
    }
    double win[NT];
#pragma unroll
    for (int t = 0; t < NT - 1; ++t) win[t] = (double)v[t];
#pragma unroll
    for (int m = 0; m < kMarch; ++m) {
        win[(m + NT - 1) % NT] = (double)v[m + NT - 1];
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], win[(m + NT - 1 - j) % NT]);  // = in[a0 + m + c - j]
        if (m < count) {
            float r = (float)acc;
            const int o = (a0 + m) * stride;
            if (MASK && fabsf(msk[o]) < 1e-6f) r = 0.0f;
            dst[o] = r;
            if (UPDATE) moved[o] = wv[m] - rate * r;  // lsf_hier_update's component-wise half (a11), as lsf_convolve_xyz
        }
    }
}

typedef float cvf4 __attribute__((ext_vector_type(4)));

// x pass, nx % 4 == 0, NT <= 9 (reach <= 4): one thread = one aligned quad of outputs
template <int NT, bool MASK, bool FMA>
__global__ __launch_bounds__(kBlock) void convolve_x4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const float* __restrict__ mask_src, Grid g,
                                                             TapsN<NT> taps, long long first_quad, long long n_quads,
                                                             lsf_gate gate) {
    if (gate_closed(gate)) return;
    const long long k = blockIdx.x * (long long)kBlock + threadIdx.x;
    if (k >= n_quads) return;
    const long long q = first_quad + k;
    const unsigned quads_x = (unsigned)g.nx / 4u;
    const unsigned qx = (unsigned)(q % quads_x);
    const long long base = (long long)blockIdx.y * g.plane;
    const cvf4* __restrict__ src = reinterpret_cast<const cvf4*>(in + base);
    const cvf4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const cvf4 mid = src[q];
    const cvf4 left = qx > 0 ? src[q - 1] : zero;
    const cvf4 right = qx + 1 < quads_x ? src[q + 1] : zero;
    const float v[12] = {left.x, left.y, left.z, left.w, mid.x, mid.y, mid.z, mid.w, right.x, right.y, right.z, right.w};
    double d[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) d[t] = (double)v[t];
    constexpr int c = NT / 2;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], d[4 + i + c - j]);
        r[i] = (float)acc;
    }
    if (MASK) {
        const cvf4 m = reinterpret_cast<const cvf4*>(mask_src + base)[q];
        if (fabsf(m.x) < 1e-6f) r[0] = 0.0f;
        if (fabsf(m.y) < 1e-6f) r[1] = 0.0f;
        if (fabsf(m.z) < 1e-6f) r[2] = 0.0f;
        if (fabsf(m.w) < 1e-6f) r[3] = 0.0f;
    }
    cvf4 o;
    o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
    reinterpret_cast<cvf4*>(out + base)[q] = o;
}

// one zero-preserving pass at LISTED voxels only (SobolevFusion on a band list): the gradient is zero outside the
// narrow band, a masked pass leaves zeros where its mask source is zero (math_utils/convolution.py:118-127), so only
// band voxels can hold non-zero output -- everything else stays at the zeros the caller initialised.  One thread per
// listed voxel (all planes); NT taps read along the axis from the planar field (zero padding outside the array).
template <int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void convolve_list_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               const float* __restrict__ mask_src, Grid g,
                                                               TapsN<NT> taps, int axis, int planes,
                                                               const int* __restrict__ list, unsigned count,
                                                               lsf_gate gate) {
    if (gate_closed(gate)) return;
    const unsigned k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= count) return;
    const unsigned i = (unsigned)list[k];
    const unsigned zy = fast_div(i, g.div_nx);
    const int x = (int)(i - zy * (unsigned)g.nx);
    const int z = (int)fast_div(zy, g.div_ny);
    const int y = (int)zy - z * g.ny;
    const int a = axis == 0 ? x : (axis == 1 ? y : z);
    const int len = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
    const int stride = axis == 0 ? 1 : (axis == 1 ? g.nx : g.nx * g.ny);
    constexpr int c = NT / 2;
    // every plane of the voxel in one thread, mask and taps fetched together (one round trip after the list entry; the
    // taps of a masked component are then simply not used): these passes are bound by their chains of dependent loads
#pragma unroll
    for (int plane = 0; plane < 4; ++plane) {  // planes <= 4 (checked by the entry point)
        if (plane >= planes) break;
        const long long base = (long long)plane * g.plane;
        const float* __restrict__ src = in + base + i;
        const float m = mask_src[base + i];
        float v[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {  // out[a] = sum_j k[j] * in[a + c - j], zero outside [0, len)
            const int d = c - j, q = a + d;
            const bool inside = q >= 0 && q < len;
            const float t = src[inside ? d * stride : 0];
            v[j] = inside ? t : 0.0f;
        }
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], (double)v[j]);
        out[base + i] = fabsf(m) < 1e-6f ? 0.0f : (float)acc;
    }
}

template <int NT>
static void launch_list_pass(const float* in, float* out, const float* mask, const Grid& g, int planes, int axis,
                             const double* taps_host, const int* list, unsigned count, const lsf_gate& gt,
                             hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    if (taps_are_float32(taps_host, NT))
        hipLaunchKernelGGL((convolve_list_kernel<NT, true>), dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, in,
                           out, mask, g, taps, axis, planes, list, count, gt);
    else
        hipLaunchKernelGGL((convolve_list_kernel<NT, false>), dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, in,
                           out, mask, g, taps, axis, planes, list, count, gt);
}

template <int NT, bool MASK, bool FMA>
__global__ __launch_bounds__(kBlock) void march_y_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         const float* __restrict__ mask_src, Grid g, TapsN<NT> taps,
                                                         lsf_gate gate) {
    convolve_march<1, NT, MASK, FMA>(in, out, mask_src, g, taps, gate);
}

template <int NT, bool MASK, bool FMA>
__global__ __launch_bounds__(kBlock) void march_z_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         const float* __restrict__ mask_src, Grid g, TapsN<NT> taps,
                                                         lsf_gate gate) {
    convolve_march<2, NT, MASK, FMA>(in, out, mask_src, g, taps, gate);
}

// the LAST pass of a hierarchical iteration's filter (y in 2-D, z in 3-D): as it writes the filtered gradient it also
// moves the warp by it, component by component (hierarchical_optimizer2d.py:220-222) -- what lsf_convolve_xyz does on
// large levels; the maximum, which needs all components of a voxel, stays lsf_hier_update's (or the next iteration's)
template <int AXIS, int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void march_update_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              float* __restrict__ warp, float rate, Grid g,
                                                              TapsN<NT> taps, lsf_gate gate) {
    convolve_march<AXIS, NT, false, FMA, true>(in, out, nullptr, g, taps, gate, warp, rate);
}

template <int NT>
static bool launch_update_pass(const float* in, float* out, float* warp, float rate, const Grid& g, int planes, int axis,
                               const double* taps_host, const lsf_gate& gt, hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    const int slices = g.z_end - g.z_begin;
    const unsigned tiles_x = (unsigned)(g.nx + kTileX - 1) / kTileX, waves = kBlock / kTileX;
    const bool fma = taps_are_float32(taps_host, NT);
    const unsigned runs = (unsigned)((axis == 1 ? g.ny : slices) + kMarch - 1) / kMarch;
    const unsigned others = (unsigned)(axis == 1 ? slices : g.ny);
    const dim3 grid(tiles_x, runs * ((others + waves - 1) / waves), (unsigned)planes);
    if (grid.y > 65535u) return false;
    if (axis == 1) {
        if (fma) hipLaunchKernelGGL((march_update_kernel<1, NT, true>), grid, dim3(kBlock), 0, s, in, out, warp, rate, g, taps, gt);
        else hipLaunchKernelGGL((march_update_kernel<1, NT, false>), grid, dim3(kBlock), 0, s, in, out, warp, rate, g, taps, gt);
    } else {
        if (fma) hipLaunchKernelGGL((march_update_kernel<2, NT, true>), grid, dim3(kBlock), 0, s, in, out, warp, rate, g, taps, gt);
        else hipLaunchKernelGGL((march_update_kernel<2, NT, false>), grid, dim3(kBlock), 0, s, in, out, warp, rate, g, taps, gt);
    }
    return true;
}

template <int NT>
static bool launch_window_pass(const float* in, float* out, const float* mask, const Grid& g, int planes, int axis,
                               const double* taps_host, const lsf_gate& gt, hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    const int slices = g.z_end - g.z_begin;
    const unsigned tiles_x = (unsigned)(g.nx + kTileX - 1) / kTileX, waves = kBlock / kTileX;
    const bool fma = taps_are_float32(taps_host, NT);
#define LSF_LAUNCH_PASS(KERNEL, ...)                                                                                 \
    do {                                                                                                             \
        if (mask && fma) hipLaunchKernelGGL((KERNEL<NT, true, true>), grid, dim3(kBlock), 0, s, __VA_ARGS__);          \
        else if (mask) hipLaunchKernelGGL((KERNEL<NT, true, false>), grid, dim3(kBlock), 0, s, __VA_ARGS__);           \
        else if (fma) hipLaunchKernelGGL((KERNEL<NT, false, true>), grid, dim3(kBlock), 0, s, __VA_ARGS__);            \
        else hipLaunchKernelGGL((KERNEL<NT, false, false>), grid, dim3(kBlock), 0, s, __VA_ARGS__);                    \
    } while (0)
    if (axis == 0) {
        if (g.nx % 4 != 0) return false;
        const long long quads_per_slice = (long long)g.ny * (g.nx / 4);
        const long long first = quads_per_slice * g.z_begin, n = quads_per_slice * slices;
        const dim3 grid((unsigned)((n + kBlock - 1) / kBlock), (unsigned)planes);
        LSF_LAUNCH_PASS(convolve_x4_kernel, in, out, mask, g, taps, first, n, gt);
    } else if (axis == 1) {
        const unsigned runs = (unsigned)(g.ny + kMarch - 1) / kMarch;
        const dim3 grid(tiles_x, runs * (((unsigned)slices + waves - 1) / waves), (unsigned)planes);
        if (grid.y > 65535u) return false;
        LSF_LAUNCH_PASS(march_y_kernel, in, out, mask, g, taps, gt);
    } else {
        const unsigned runs = (unsigned)(slices + kMarch - 1) / kMarch;
        const dim3 grid(tiles_x, runs * (((unsigned)g.ny + waves - 1) / waves), (unsigned)planes);
        if (grid.y > 65535u) return false;
        LSF_LAUNCH_PASS(march_z_kernel, in, out, mask, g, taps, gt);
    }
#undef LSF_LAUNCH_PASS
    return true;
}

// ---- all three passes of a 3-D separable filter in ONE launch (x, then y, then z: math_utils/convolution.py:94-105) ----
// A block owns a column of kXyzRows rows x 64 x-positions and marches through kXyzChunk output slices (+ NT - 1 slices of
// run-in).  Per slice: the raw tile with its x / y reach goes through LDS, the x pass writes its float32-rounded
// results back to LDS, the y pass reads them into registers -- and the z pass never leaves the
// registers: each thread keeps the last NT (x, y)-filtered values of its four columns.  One read of the input and one
// write of the output instead of three each; the run-in slices are re-read by the neighbouring chunk (x 1.19 at 32).
// Same arithmetic as the single passes: float64 products and sums in tap order, one float32 rounding per pass, zeros
// outside the array.
constexpr int kXyzRows = 16;
constexpr int kXyzChunk = 32;
constexpr int kXyzCols = kTileX + 8;  // the tile's columns plus one aligned quad on either side (reach <= 4)

template <int NT, bool FMA>
__global__ __launch_bounds__(kBlock, 5) void convolve_xyz_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              float* __restrict__ warp, float rate, Grid g,
                                                              TapsN<NT> taps, unsigned chunks_z, int chunk,
                                                              lsf_gate gate) {
    if (gate_closed(gate)) return;
    constexpr int R = NT / 2;                   // odd NT: reach R on either side, out[o] = sum_j k[j] * in[o + R - j]
    constexpr int kStaged = kXyzRows + 2 * R;   // rows of the raw tile and of the x pass
    constexpr int kQuads = kStaged * (kXyzCols / 4);
    constexpr int kLoads = (kQuads + kBlock - 1) / kBlock;
    __shared__ __attribute__((aligned(16))) float raw[kStaged][kXyzCols];  // float32 in LDS: its bandwidth, not the conversions, is the scarcer one
    __shared__ float xs[kStaged][kTileX];
    const int t = threadIdx.x, lx = t & (kTileX - 1), wy = t / kTileX;
    const int x0 = blockIdx.x * kTileX;
    const int y0 = (int)(blockIdx.y / chunks_z) * kXyzRows;
    const int z0 = g.z_begin + (int)(blockIdx.y % chunks_z) * chunk;  // output slices: the grid's z-range; the run-in
    const int count = min(chunk, g.z_end - z0);                       // reads whatever slices of the array it needs
    const float* __restrict__ src = in + (long long)blockIdx.z * g.plane;
    float* __restrict__ dst = out + (long long)blockIdx.z * g.plane;
    float* __restrict__ moved = warp ? warp + (long long)blockIdx.z * g.plane : nullptr;

    // this thread's share of a slice's raw tile: quads q = t + 256 m -> (row, quad column)
    int q_row[kLoads], q_col[kLoads], q_off[kLoads];  // offsets inside a slice: < 2^31 (check_grid)
    bool q_ok[kLoads];
#pragma unroll
    for (int m = 0; m < kLoads; ++m) {
        const int q = t + m * kBlock;
        q_row[m] = q / (kXyzCols / 4);
        q_col[m] = (q % (kXyzCols / 4)) * 4;
        const int gx = x0 - 4 + q_col[m], gy = y0 - R + q_row[m];
        q_ok[m] = q < kQuads && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny;  // nx % 4 == 0: quads are in or out whole
        q_off[m] = gy * g.nx + gx;
    }
    const cvf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const long long slice = (long long)g.nx * g.ny;
    auto fetch = [&](int p, cvf4 (&v)[kLoads]) {
#pragma unroll
        for (int m = 0; m < kLoads; ++m)
            v[m] = (q_ok[m] && p >= 0 && p < g.nz) ? *reinterpret_cast<const cvf4*>(src + p * slice + q_off[m]) : zero4;
    };

    float win[4][NT];  // win[k][i]: (x, y)-filtered value of column k at slice p - (NT - 1) + i
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < NT; ++i) win[k][i] = 0.0f;

    const int p_first = z0 - R, p_last = z0 + count - 1 + R;
    cvf4 next[kLoads];
    fetch(p_first, next);
    for (int p = p_first; p <= p_last; ++p) {
        const bool inside = p >= 0 && p < g.nz;  // uniform
        float value[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (inside) {
#pragma unroll
            for (int m = 0; m < kLoads; ++m) {
                if (t + m * kBlock < kQuads) {
                    *reinterpret_cast<cvf4*>(&raw[q_row[m]][q_col[m]]) = next[m];
                }
            }
        }
        __syncthreads();  // raw complete; every wave is done with the previous slice's xs
        fetch(p + 1 <= p_last ? p + 1 : -1, next);  // in flight while this slice is filtered
        if (inside) {
#pragma unroll
            for (int m = 0; m < (kStaged + 3) / 4; ++m) {  // x pass: rows wy, wy + 4, ...
                const int r = wy + 4 * m;
                if (r < kStaged) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], (double)raw[r][lx + 4 + R - j]);
                    xs[r][lx] = (float)acc;
                }
            }
        }
        __syncthreads();  // xs complete; every wave is done with raw
        if (inside) {
            double d[4 + 2 * R];  // y pass: four consecutive rows share their taps
#pragma unroll
            for (int i = 0; i < 4 + 2 * R; ++i) d[i] = (double)xs[4 * wy + i][lx];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], d[k + 2 * R - j]);
                value[k] = (float)acc;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i + 1 < NT; ++i) win[k][i] = win[k][i + 1];
            win[k][NT - 1] = value[k];
        }
        const int po = p - R;  // the output slice whose NT taps are now in the window
        if (po >= z0 && x0 + lx < g.nx) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int y = y0 + 4 * wy + k;
                if (y < g.ny) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], (double)win[k][2 * R - j]);
                    const long long o = ((long long)po * g.ny + y) * g.nx + x0 + lx;
                    const float r = (float)acc;
                    dst[o] = r;
                    if (moved) moved[o] = moved[o] - rate * r;  // lsf_hier_update's component-wise half (a11)
                }
            }
        }
    }
}

// ---- the x and the y pass of a 3-D filter in ONE launch (levels below the fused kernel's size: its 64 x 16-column blocks
// march through 32 slices, which leaves most of the chip idle at 128^3 and below; there the passes are launch-bound and one
// launch less per iteration is what counts).  A block owns a 64 x 16 tile of ONE slice: the raw tile with its x / y reach goes
// through LDS, the x pass writes its float32-rounded results back to LDS, the y pass reads them.  convolve_xyz_kernel's first
// two stages, slice by slice: the same arithmetic as the single passes.
template <int NT, bool FMA>
__global__ __launch_bounds__(kBlock) void convolve_xy_kernel(const float* __restrict__ in, float* __restrict__ out, Grid g,
                                                             TapsN<NT> taps, unsigned tiles_y, lsf_gate gate) {
    if (gate_closed(gate)) return;
    constexpr int R = NT / 2;
    constexpr int kStaged = kXyzRows + 2 * R;
    constexpr int kQuads = kStaged * (kXyzCols / 4);
    constexpr int kLoads = (kQuads + kBlock - 1) / kBlock;
    __shared__ __attribute__((aligned(16))) float raw[kStaged][kXyzCols];
    __shared__ float xs[kStaged][kTileX];
    const int t = threadIdx.x, lx = t & (kTileX - 1), wy = t / kTileX;
    const int x0 = blockIdx.x * kTileX;
    const int y0 = (int)(blockIdx.y % tiles_y) * kXyzRows;
    const int p = g.z_begin + (int)(blockIdx.y / tiles_y);  // the slice
    const float* __restrict__ src = in + (long long)blockIdx.z * g.plane + (long long)p * g.nx * g.ny;
    float* __restrict__ dst = out + (long long)blockIdx.z * g.plane + (long long)p * g.nx * g.ny;
    const cvf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < kLoads; ++m) {
        const int q = t + m * kBlock;
        const int row = q / (kXyzCols / 4), col = (q % (kXyzCols / 4)) * 4;
        const int gx = x0 - 4 + col, gy = y0 - R + row;
        const bool ok = q < kQuads && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny;  // nx % 4 == 0: quads are in or out whole
        const cvf4 v = ok ? *reinterpret_cast<const cvf4*>(src + gy * g.nx + gx) : zero4;
        if (q < kQuads) *reinterpret_cast<cvf4*>(&raw[row][col]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < (kStaged + 3) / 4; ++m) {  // x pass: rows wy, wy + 4, ...
        const int r = wy + 4 * m;
        if (r < kStaged) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], (double)raw[r][lx + 4 + R - j]);
            xs[r][lx] = (float)acc;
        }
    }
    __syncthreads();
    double d[4 + 2 * R];  // y pass: four consecutive rows share their taps
#pragma unroll
    for (int i = 0; i < 4 + 2 * R; ++i) d[i] = (double)xs[4 * wy + i][lx];
    if (x0 + lx < g.nx) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = y0 + 4 * wy + k;
            if (y < g.ny) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NT; ++j) acc = mac<FMA>(acc, taps.k[j], d[k + 2 * R - j]);
                dst[y * g.nx + x0 + lx] = (float)acc;
            }
        }
    }
}

template <int NT>
static bool launch_xy(const float* in, float* out, const Grid& g, int planes, const double* taps_host, const lsf_gate& gt,
                      hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    const unsigned tiles_x = (unsigned)(g.nx + kTileX - 1) / kTileX, tiles_y = (unsigned)(g.ny + kXyzRows - 1) / kXyzRows;
    const unsigned long long blocks_y = (unsigned long long)tiles_y * (unsigned)(g.z_end - g.z_begin);
    if (blocks_y > 65535ull) return false;
    const dim3 grid(tiles_x, (unsigned)blocks_y, (unsigned)planes);
    if (taps_are_float32(taps_host, NT))
        hipLaunchKernelGGL((convolve_xy_kernel<NT, true>), grid, dim3(kBlock), 0, s, in, out, g, taps, tiles_y, gt);
    else
        hipLaunchKernelGGL((convolve_xy_kernel<NT, false>), grid, dim3(kBlock), 0, s, in, out, g, taps, tiles_y, gt);
    return true;
}

template <int NT>
static void launch_xyz(const float* in, float* out, float* warp, float rate, const Grid& g, int planes,
                       const double* taps_host, const lsf_gate& gt, hipStream_t s) {
    TapsN<NT> taps;
    for (int j = 0; j < NT; ++j) taps.k[j] = taps_host[j];
    const unsigned tiles_x = (unsigned)(g.nx + kTileX - 1) / kTileX, tiles_y = (unsigned)(g.ny + kXyzRows - 1) / kXyzRows;
    const int chunk = kXyzChunk;
    const unsigned chunks_z = (unsigned)(g.z_end - g.z_begin + chunk - 1) / chunk;
    if (taps_are_float32(taps_host, NT))
        hipLaunchKernelGGL((convolve_xyz_kernel<NT, true>), dim3(tiles_x, tiles_y * chunks_z, (unsigned)planes),
                           dim3(kBlock), 0, s, in, out, warp, rate, g, taps, chunks_z, chunk, gt);
    else
        hipLaunchKernelGGL((convolve_xyz_kernel<NT, false>), dim3(tiles_x, tiles_y * chunks_z, (unsigned)planes),
                           dim3(kBlock), 0, s, in, out, warp, rate, g, taps, chunks_z, chunk, gt);
}

extern "C" int lsf_convolve_xyz(const float* in_planar, float* out_planar, float* warp_planar, float rate,
                                const lsf_grid* grid, int32_t planes, const double* taps_host, int32_t n_taps,
                                const lsf_gate* gate, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !out_planar || in_planar == out_planar || !taps_host || planes < 1 || planes > 4 ||
        warp_planar == in_planar || (warp_planar && warp_planar == out_planar))
        return LSF_ERR_BAD_ARGUMENT;
    if (grid->dims != 3 || grid->nx % 4 != 0) return LSF_ERR_BAD_DIMS;
    if (grid->z_end == grid->z_begin) return 0;
    if (n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_KERNEL_TOO_LONG;
    const Grid g = make_grid(grid);
    const unsigned long long blocks_y = (unsigned long long)((g.ny + kXyzRows - 1) / kXyzRows) *
                                        (unsigned long long)((g.z_end - g.z_begin + kXyzChunk - 1) / kXyzChunk);
    if (blocks_y > 65535ull) return LSF_ERR_BAD_DIMS;
    const lsf_gate gt = gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
    hipStream_t s = as_stream(stream);
    switch (n_taps) {
        case 3: launch_xyz<3>(in_planar, out_planar, warp_planar, rate, g, planes, taps_host, gt, s); break;
        case 5: launch_xyz<5>(in_planar, out_planar, warp_planar, rate, g, planes, taps_host, gt, s); break;
        case 7: launch_xyz<7>(in_planar, out_planar, warp_planar, rate, g, planes, taps_host, gt, s); break;
        default: launch_xyz<9>(in_planar, out_planar, warp_planar, rate, g, planes, taps_host, gt, s); break;
    }
    return launch_status();
}

extern "C" int lsf_convolve_axis_listed(const float* in_planar, float* out_planar, const float* zero_mask_source,
                                        const lsf_grid* grid, int32_t planes, int32_t axis, const double* taps_host,
                                        int32_t n_taps, const lsf_gate* gate, const int32_t* band_list,
                                        int64_t band_count, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !out_planar || in_planar == out_planar || !taps_host || !zero_mask_source || !band_list ||
        band_count < 0 || band_count > 0x7fffffffll)
        return LSF_ERR_BAD_ARGUMENT;
    if (n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_KERNEL_TOO_LONG;
    if (axis < 0 || axis >= grid->dims || planes < 1 || planes > 4) return LSF_ERR_BAD_ARGUMENT;
    if (band_count == 0) return 0;
    const Grid g = make_grid(grid);
    const lsf_gate gt = gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
    hipStream_t s = as_stream(stream);
    const unsigned n = (unsigned)band_count;
    switch (n_taps) {
        case 3: launch_list_pass<3>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, band_list, n, gt, s); break;
        case 5: launch_list_pass<5>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, band_list, n, gt, s); break;
        case 7: launch_list_pass<7>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, band_list, n, gt, s); break;
        default: launch_list_pass<9>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, band_list, n, gt, s); break;
    }
    return launch_status();
}

extern "C" int lsf_convolve_axis(const float* in_planar, float* out_planar, const float* zero_mask_source,
                                 const lsf_grid* grid, int32_t planes, int32_t axis, const double* taps_host,
                                 int32_t n_taps, const lsf_gate* gate, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !out_planar || in_planar == out_planar || !taps_host) return LSF_ERR_BAD_ARGUMENT;
    if (n_taps < 1 || n_taps > LSF_MAX_KERNEL_TAPS) return LSF_ERR_KERNEL_TOO_LONG;
    if (axis < 0 || axis >= grid->dims || planes < 1 || planes > 4) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    const int slices = g.z_end - g.z_begin;
    if (slices == 0) return 0;
    Taps taps;
    taps.n = n_taps;
    for (int j = 0; j < LSF_MAX_KERNEL_TAPS; ++j) taps.k[j] = j < n_taps ? taps_host[j] : 0.0;
    lsf_gate gt = gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
    const unsigned tiles_x = (unsigned)(g.nx + kTileX - 1) / kTileX;
    hipStream_t s = as_stream(stream);
    bool done = false;
    switch (n_taps) {  // register-window kernels for the common tap counts; anything else: the tiled kernels
        case 3: done = launch_window_pass<3>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, gt, s); break;
        case 5: done = launch_window_pass<5>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, gt, s); break;
        case 7: done = launch_window_pass<7>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, gt, s); break;
        case 9: done = launch_window_pass<9>(in_planar, out_planar, zero_mask_source, g, planes, axis, taps_host, gt, s); break;
        default: break;
    }
    if (done) return launch_status();
    if (axis == 0) {
        const unsigned tiles_y = (unsigned)(g.ny + kConvRows - 1) / kConvRows;
        hipLaunchKernelGGL(convolve_x_kernel, dim3(tiles_x, tiles_y * slices, planes), dim3(kBlock),
                           sizeof(double) * kConvRows * (kTileX + n_taps - 1), s, in_planar,
                           out_planar, zero_mask_source, g, taps, gt);
    } else if (axis == 1) {
        const unsigned runs = (unsigned)(g.ny + kConvRun - 1) / kConvRun;
        hipLaunchKernelGGL(convolve_strided_kernel<1>, dim3(tiles_x, runs * slices, planes), dim3(kBlock),
                           sizeof(double) * kTileX * (kConvRun + n_taps - 1), s,
                           in_planar, out_planar, zero_mask_source, g, taps, gt);
    } else {
        const unsigned runs = (unsigned)(slices + kConvRun - 1) / kConvRun;
        hipLaunchKernelGGL(convolve_strided_kernel<2>, dim3(tiles_x, runs * g.ny, planes), dim3(kBlock),
                           sizeof(double) * kTileX * (kConvRun + n_taps - 1), s,
                           in_planar, out_planar, zero_mask_source, g, taps, gt);
    }
    return launch_status();
}

extern "C" int lsf_convolve_xy(const float* in_planar, float* out_planar, const lsf_grid* grid, int32_t planes,
                               const double* taps_host, int32_t n_taps, const lsf_gate* gate, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !out_planar || in_planar == out_planar || !taps_host || planes < 1 || planes > 4)
        return LSF_ERR_BAD_ARGUMENT;
    if (grid->dims != 3 || grid->nx % 4 != 0) return LSF_ERR_BAD_DIMS;
    if (grid->z_end == grid->z_begin) return 0;
    if (n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_KERNEL_TOO_LONG;
    const Grid g = make_grid(grid);
    const lsf_gate gt = gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
    hipStream_t s = as_stream(stream);
    bool done = false;
    switch (n_taps) {
        case 3: done = launch_xy<3>(in_planar, out_planar, g, planes, taps_host, gt, s); break;
        case 5: done = launch_xy<5>(in_planar, out_planar, g, planes, taps_host, gt, s); break;
        case 7: done = launch_xy<7>(in_planar, out_planar, g, planes, taps_host, gt, s); break;
        default: done = launch_xy<9>(in_planar, out_planar, g, planes, taps_host, gt, s); break;
    }
    return done ? launch_status() : LSF_ERR_BAD_DIMS;
}

extern "C" int lsf_convolve_axis_update(const float* in_planar, float* out_planar, float* warp_planar, float rate,
                                        const lsf_grid* grid, int32_t planes, int32_t axis, const double* taps_host,
                                        int32_t n_taps, const lsf_gate* gate, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!in_planar || !out_planar || in_planar == out_planar || !warp_planar || warp_planar == out_planar ||
        warp_planar == in_planar || !taps_host)
        return LSF_ERR_BAD_ARGUMENT;
    if (n_taps < 1 || n_taps > LSF_MAX_KERNEL_TAPS) return LSF_ERR_KERNEL_TOO_LONG;
    if (axis < 1 || axis >= grid->dims || planes < 1 || planes > 4) return LSF_ERR_BAD_ARGUMENT;
    if (n_taps != 3 && n_taps != 5 && n_taps != 7 && n_taps != 9) return LSF_ERR_BAD_DIMS;
    Grid g = make_grid(grid);
    if (g.z_end - g.z_begin == 0) return 0;
    lsf_gate gt = gate ? *gate : lsf_gate{nullptr, 0, 0.0f, 0.0f};
    hipStream_t s = as_stream(stream);
    bool done = false;
    switch (n_taps) {
        case 3: done = launch_update_pass<3>(in_planar, out_planar, warp_planar, rate, g, planes, axis, taps_host, gt, s); break;
        case 5: done = launch_update_pass<5>(in_planar, out_planar, warp_planar, rate, g, planes, axis, taps_host, gt, s); break;
        case 7: done = launch_update_pass<7>(in_planar, out_planar, warp_planar, rate, g, planes, axis, taps_host, gt, s); break;
        default: done = launch_update_pass<9>(in_planar, out_planar, warp_planar, rate, g, planes, axis, taps_host, gt, s); break;
    }
    return done ? launch_status() : LSF_ERR_BAD_DIMS;
}

// =====================================================================================================
//  a20  convergence statistics
// =====================================================================================================
__global__ void stats_init_kernel(double* out8, int is_tsdf) {
    if (threadIdx.x < 8) out8[threadIdx.x] = 0.0;
    if (threadIdx.x == 0 && is_tsdf) out8[1] = __longlong_as_double(0x7ff0000000000000ll);  // +inf (min)
}

__device__ inline void atomic_min_f64_nonneg(double* addr, double v) {
    // non-negative doubles order like their bit patterns
    atomicMin(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__device__ inline void atomic_max_f64_nonneg(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

template <int D>
__global__ __launch_bounds__(kBlock) void warp_statistics_kernel(const float* __restrict__ warp,
                                                                 const float* __restrict__ canonical,
                                                                 const float* __restrict__ live, Grid g, float lo,
                                                                 double* out8, unsigned long long* packed_out) {
    unsigned long long packed = 0ull;
    double sums[4] = {0.0, 0.0, 0.0, 0.0};
    for_each_voxel(g, [&](int x, int y, int z) {
        long long i = vidx(g, x, y, z);
        bool band = !(fabsf(live[i]) == 1.0f && fabsf(canonical[i]) == 1.0f);
        if (band) {
            float v[3] = {warp[i], warp[g.plane + i], D == 3 ? warp[2 * g.plane + i] : 0.0f};
            float len = vec_length<D>(v);
            unsigned long long p = pack_max(len, linear_index(g, x, y, z));
            packed = p > packed ? p : packed;
            sums[0] += 1.0;
            sums[1] += len > lo ? 1.0 : 0.0;
            sums[2] += (double)len;
            sums[3] += (double)len * (double)len;
        }
    });
    double* dst[4] = {out8 + 0, out8 + 1, out8 + 3, out8 + 4};
    block_reduce_commit<4>(packed, sums, packed_out, dst);
}

template <int D>
__global__ __launch_bounds__(kBlock) void tsdf_statistics_kernel(const float* __restrict__ canonical,
                                                                 const float* __restrict__ live, Grid g, double* out8,
                                                                 unsigned long long* packed_out) {
    unsigned long long packed = 0ull;
    double sums[3] = {0.0, 0.0, 0.0};
    double mn = __longlong_as_double(0x7ff0000000000000ll);
    for_each_voxel(g, [&](int x, int y, int z) {
        long long i = vidx(g, x, y, z);
        double d = fabs((double)canonical[i] - (double)live[i]);
        unsigned long long p = pack_max((float)d, linear_index(g, x, y, z));
        packed = p > packed ? p : packed;
        sums[0] += 1.0;
        sums[1] += d;
        sums[2] += d * d;
        mn = fmin(mn, d);
    });
    // min: wave shuffle, one atomic per wave (at most 4 * kMaxBlocks atomics per launch)
    for (int dl = kWave / 2; dl > 0; dl >>= 1) mn = fmin(mn, shfl_down_f64(mn, dl));
    if ((threadIdx.x & (kWave - 1)) == 0) atomic_min_f64_nonneg(out8 + 1, mn);
    double* dst[3] = {out8 + 0, out8 + 3, out8 + 4};
    block_reduce_commit<3>(packed, sums, packed_out, dst);
}

__global__ void stats_finish_kernel(double* out8, const unsigned long long* packed, int is_tsdf) {
    if (threadIdx.x != 0) return;
    unsigned long long p = *packed;
    out8[2] = p ? (double)unpack_max_value(p) : 0.0;
    out8[5] = p ? (double)(~(unsigned)p) : -1.0;
    (void)is_tsdf;
}

extern "C" int lsf_warp_statistics(const float* warp_planar, const float* canonical, const float* live,
                                   const lsf_grid* grid, float lower_threshold, double* out8, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!warp_planar || !canonical || !live || !out8) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    hipStream_t s = as_stream(stream);
    unsigned long long* packed = reinterpret_cast<unsigned long long*>(out8 + 7);
    hipLaunchKernelGGL(stats_init_kernel, dim3(1), dim3(64), 0, s, out8, 0);
    if (t.total) {
        if (grid->dims == 2)
            hipLaunchKernelGGL(warp_statistics_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, s, warp_planar, canonical,
                               live, g, lower_threshold, out8, packed);
        else
            hipLaunchKernelGGL(warp_statistics_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, s, warp_planar, canonical,
                               live, g, lower_threshold, out8, packed);
    }
    hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, s, out8, packed, 0);
    return launch_status();
}

extern "C" int lsf_tsdf_difference_statistics(const float* canonical, const float* live, const lsf_grid* grid,
                                              double* out8, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (!canonical || !live || !out8) return LSF_ERR_BAD_ARGUMENT;
    Grid g = make_grid(grid);
    Tiling t = make_tiling(g);
    hipStream_t s = as_stream(stream);
    unsigned long long* packed = reinterpret_cast<unsigned long long*>(out8 + 7);
    hipLaunchKernelGGL(stats_init_kernel, dim3(1), dim3(64), 0, s, out8, 1);
    if (t.total) {
        if (grid->dims == 2)
            hipLaunchKernelGGL(tsdf_statistics_kernel<2>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, s, canonical, live, g, out8,
                               packed);
        else
            hipLaunchKernelGGL(tsdf_statistics_kernel<3>, dim3(launch_blocks(t.total)), dim3(kBlock), 0, s, canonical, live, g, out8,
                               packed);
    }
    hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, s, out8, packed, 1);
    return launch_status();
}

extern "C" int lsf_abi_version(void) { return LSF_ABI_VERSION; }
extern "C" const char* lsf_target_arch(void) { return "gfx950"; }
#ifndef LSF_BUILD_ID
#define LSF_BUILD_ID "unknown"
#endif
extern "C" const char* lsf_build_id(void) { return LSF_BUILD_ID; }
#ifndef LSF_ABI_HASH
#define LSF_ABI_HASH "unknown"
#endif
extern "C" const char* lsf_abi_hash(void) { return LSF_ABI_HASH; }

// records on the HOST -> their values (max over the slots' packed maxima, slot-ordered sum of the energies): what
// device.decode_records did with a dozen numpy calls (~30 us at the end of every optimize() call)
extern "C" int lsf_records_decode(const int64_t* slots, int32_t n, int32_t n_slots, int32_t slot_words, float* max_value,
                                  int64_t* argmax, double* energies3, uint8_t* executed) {
    if (n < 0 || n_slots < 1 || slot_words < 4 || (n > 0 && (!slots || !max_value || !argmax || !energies3 || !executed)))
        return LSF_ERR_BAD_ARGUMENT;
    for (int32_t r = 0; r < n; ++r) {
        const int64_t* rec = slots + (size_t)r * n_slots * slot_words;
        uint64_t packed = 0;
        double e[3] = {0.0, 0.0, 0.0};
        for (int32_t k = 0; k < n_slots; ++k) {
            const int64_t* w = rec + (size_t)k * slot_words;
            const uint64_t p = (uint64_t)w[0];
            packed = p > packed ? p : packed;
            for (int c = 0; c < 3; ++c) {
                double v;
                memcpy(&v, &w[1 + c], sizeof(v));
                e[c] += v;
            }
        }
        const uint32_t bits = (uint32_t)(packed >> 32), inverted = (uint32_t)packed;
        memcpy(&max_value[r], &bits, sizeof(float));
        argmax[r] = (int64_t)(uint32_t)~inverted;
        energies3[3 * r + 0] = e[0];
        energies3[3 * r + 1] = e[1];
        energies3[3 * r + 2] = e[2];
        executed[r] = packed != 0;
    }
    return 0;
}
