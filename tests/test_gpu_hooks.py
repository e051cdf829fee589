"""Opt-in per-iteration call-backs on the drop-in optimizers: the place where the reference calls its visualiser inside
both hot loops (hierarchical_optimizer2d.py:242-245, slavcheva_optimizer2d.py:387-388).  A hook sees every executed
iteration with the fields the oracle's own hook sees; without one nothing changes (results and iteration counts)."""
import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu


def exact(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max()) == 0.0


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


@pytest.mark.parametrize("d", [2, 3])
def test_hierarchical_iteration_hook(lsf, d):
    n = 32
    canonical, live = O.sphere_pair(n, d=d)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=4, rate=0.1,
              maximum_iteration_count=5, maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    cls = lsf.HierarchicalOptimizer2d if d == 2 else lsf.HierarchicalOptimizer3d
    seen, want = [], []
    opt = cls(**kw)
    assert opt.iteration_hook is None
    opt.iteration_hook = lambda level, it, warp, g, m: seen.append((level, it, warp.cpu().numpy(), g.cpu().numpy(), m))
    warp = opt.optimize(canonical, live)
    o = O.HierarchicalOracle(**kw)
    o.iteration_hook = lambda level, it, warp, g, m: want.append((level, it, warp.copy(), g.copy(), m))
    warp_ref = o.optimize(canonical, live)
    assert exact(warp, warp_ref)
    assert [(s[0], s[1]) for s in seen] == [(w[0], w[1]) for w in want] and len(seen) == 3 * 5
    for s, w in zip(seen, want):
        assert exact(s[2], w[2]) and exact(s[3], w[3]) and np.float32(s[4]) == np.float32(w[4])
    # unhooked: the same result (graph replay, batched checks), nothing called
    del seen[:]
    opt.iteration_hook = None
    assert exact(opt.optimize(canonical, live), warp_ref) and not seen


@pytest.mark.parametrize("sobolev", [False, True])
def test_slavcheva_iteration_hook(lsf, sobolev):
    n = 32
    canonical, live0 = O.sphere_pair(n, d=3)
    common = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=4, min_iterations=4)
    k3 = lsf.generate_1d_sobolev_kernel(3, 0.1)
    gpu = dict(level_set_term_enabled=not sobolev, smoothing_term_method=lsf.SmoothingTermMethod.TIKHONOV if sobolev
               else lsf.SmoothingTermMethod.KILLING, sobolev_smoothing_enabled=sobolev,
               sobolev_kernel=k3 if sobolev else None)
    cpu = dict(level_set_term_enabled=not sobolev, smoothing_term_method=O.TIKHONOV if sobolev else O.KILLING,
               sobolev_smoothing_enabled=sobolev, sobolev_kernel=k3 if sobolev else None)
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, **common, **gpu)
    seen, want = [], []
    opt.iteration_hook = lambda level, it, warp, g, m: seen.append((level, it, warp.cpu().numpy(), g.cpu().numpy(), m))
    live = live0.copy()
    opt.optimize(live, canonical)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, **common, **cpu)
    o.iteration_hook = lambda it, lv, warp, g, en, m, at: want.append((it, warp.copy(), g.copy(), m))
    live_ref = live0.copy()
    o.optimize(live_ref, canonical)
    assert exact(live, live_ref) and [s[1] for s in seen] == [w[0] for w in want] == [0, 1, 2, 3]
    for s, w in zip(seen, want):
        assert s[0] == 0 and exact(s[2], w[1]) and exact(s[3], w[2]) and np.float32(s[4]) == np.float32(w[3])
    # the hook of the oracle sees the warp AFTER warp_field_advanced zeroed the snapped voxels, and so does ours
    opt.iteration_hook = None
    live2 = live0.copy()
    opt.optimize(live2, canonical)
    assert exact(live2, live_ref)


def test_hierarchical_settings_write_through(lsf):
    """the reference reads rate, thresholds, ... from `self` in every iteration (hierarchical_optimizer2d.py:186-225):
    assigning to them after construction must take effect -- also on levels that replay a captured HIP graph"""
    canonical, live = O.sphere_pair(64, d=2)
    kw = dict(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8, maximum_iteration_count=8,
              maximum_warp_update_threshold=0.0, tikhonov_strength=0.05)
    opt = lsf.HierarchicalOptimizer2d(rate=0.1, **kw)
    first = opt.optimize(canonical, live)  # captures the graphs of the small levels with rate 0.1
    opt.rate = 0.25
    opt.maximum_iteration_count = 6
    assert opt.rate == 0.25 and opt.engine.rate == 0.25
    changed = opt.optimize(canonical, live)
    fresh = lsf.HierarchicalOptimizer2d(rate=0.25, **dict(kw, maximum_iteration_count=6)).optimize(canonical, live)
    assert exact(changed, fresh) and not exact(changed, first)
    assert opt.get_per_level_iteration_counts() == [6] * len(opt.get_per_level_iteration_counts())
