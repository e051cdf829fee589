"""The chain kernel (lsf_slavcheva_state_chain: K fused iterations of an INTERIOR band list per launch, workgroups that
wait for their neighbouring list chunks only) against K launches of the per-iteration kernel: the same bits in every
word of the state, every iteration's maximum and arg-max, energies to 1e-12 (float64 sums in launch-dependent order).

A stale read across workgroups (an L1 line another CU has rewritten, a progress word seen too early) would show up as a
different voxel somewhere, so the comparisons cover several list lengths -- fewer workgroups than CUs, one per CU, several
stages --, many iterations, and every word.  Reference loop: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330,
:360-362."""
import os

import numpy as np
import pytest
import torch

from oracle import lsf_oracle as O

pytestmark = pytest.mark.gpu

BENCH = dict(level_set_term_enabled=True, gradient_descent_rate=0.1, data_term_weight=1.0, smoothing_term_weight=0.2,
             isomorphic_enforcement_factor=0.1, level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.0)


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


class _Env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(lsf, canonical, live0, iterations, chain, stages=None, check_interval=None, **kw):
    n = canonical.shape[-1]
    args = dict(BENCH, smoothing_term_method=lsf.SmoothingTermMethod.KILLING)
    args.update(kw)
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                   max_iterations=iterations, min_iterations=iterations,
                                   check_interval=check_interval or iterations, **args)
    opt._run_checks = lambda *a: None
    live = live0.clone()
    env = dict(LSF_CHAIN="1" if chain else "0")
    if stages:
        env["LSF_CHAIN_STAGES"] = str(stages)
    with _Env(**env):
        opt.optimize(live, canonical)
    return opt, live


def _same(a, b):
    (oa, la), (ob, lb) = a, b
    assert torch.equal(la, lb), "live field"
    assert torch.equal(oa.warp_field, ob.warp_field), "warp field"
    assert np.array_equal(np.float32(oa.log.max_warps), np.float32(ob.log.max_warps))
    assert oa.log.max_warp_locations == ob.log.max_warp_locations
    for x, y in ((oa.log.data_energies, ob.log.data_energies), (oa.log.smoothing_energies, ob.log.smoothing_energies),
                 (oa.log.level_set_energies, ob.log.level_set_energies)):
        assert np.allclose(x, y, rtol=1e-12, atol=0.0)


@pytest.mark.parametrize("n,iterations", [(80, 40), (96, 50), (128, 60), (256, 50)])
def test_chain_equals_per_iteration_launches(lsf, n, iterations):
    """80^3 / 96^3: fewer workgroups than CUs (below 72^3 the sphere pair's band touches the array's faces and the call
    takes per-iteration launches); 128^3: one per CU with short chunks; 256^3: the bench's launch"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live0 = sphere_pair(n, 3, "cuda")
    a = _run(lsf, canonical, live0, iterations, chain=True)
    assert a[0]._engine._chain_used
    b = _run(lsf, canonical, live0, iterations, chain=False)
    _same(a, b)
    assert 0.0 < max(a[0].log.max_warps) < 1.0


def test_chain_is_what_runs_and_splits_into_batches(lsf):
    """the fixed-count path takes the chain (one launch per check_interval batch), and batches of odd length keep the
    ping-pong parity right"""
    from levelsetfusion_python_amd import device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live0 = sphere_pair(80, 3, "cuda")
    launches = []
    original = dev.StateChain.launch

    def counting(self, first, count):
        launches.append((first, count))
        return original(self, first, count)
    dev.StateChain.launch = counting
    try:
        a = _run(lsf, canonical, live0, 23, chain=True, check_interval=7)
    finally:
        dev.StateChain.launch = original
    assert launches == [(0, 7), (7, 7), (14, 7), (21, 2)]
    b = _run(lsf, canonical, live0, 23, chain=False, check_interval=7)
    _same(a, b)


def test_chain_with_other_terms_and_the_oracle(lsf):
    """Tikhonov smoothing + thresholded-FDM data term, 30 iterations at 80^3, against the numpy oracle"""
    canonical, live0 = O.sphere_pair(80, d=3)
    kw = dict(data_term_method=lsf.DataTermMethod.THRESHOLDED_FDM)
    c, l0 = torch.from_numpy(canonical).cuda(), torch.from_numpy(live0).cuda()
    opt, live = _run(lsf, c, l0, 30, chain=True, smoothing_term_method=lsf.SmoothingTermMethod.TIKHONOV, **kw)
    o = O.SlavchevaOracle(compute_method=O.DIRECT, smoothing_term_method=O.TIKHONOV, data_term_method=O.THRESHOLDED_FDM,
                          max_iterations=30, min_iterations=30, **BENCH)
    live_ref = live0.copy()
    o.optimize(live_ref, canonical)
    assert opt._engine._chain_used
    assert np.array_equal(live.cpu().numpy(), live_ref)
    assert np.array_equal(opt.warp_field.cpu().numpy(), o.warp_field)
    assert np.array_equal(np.float32(opt.log.max_warps), np.float32(o.log["max_warps"]))


def test_stages_follow_each_other_through_the_list(lsf):
    """S = 4 stages (what 512^3 runs with): 384^3 is the smallest cube whose list is long enough for the library to deal
    chunks round-robin; identical to S = 1 and to per-iteration launches"""
    from levelsetfusion_python_amd import _lib
    from levelsetfusion_python_amd.synthetic import sphere_pair
    import ctypes
    canonical, live0 = sphere_pair(384, 3, "cuda")
    a = _run(lsf, canonical, live0, 14, chain=True, stages=4)
    shape = (ctypes.c_int32 * 4)()
    count = a[0]._engine._fast.bands[0].count
    assert _lib.chain_lib().lsf_state_chain_shape(count, 4, shape) == 0
    assert shape[1] == 4 and shape[2] > shape[0], "this list is meant to be long enough for stages: %r" % list(shape)
    b = _run(lsf, canonical, live0, 14, chain=False)
    _same(a, b)
    c = _run(lsf, canonical, live0, 14, chain=True, stages=1)
    _same(c, b)


def test_updates_beyond_the_windows_fall_back(lsf):
    """a descent rate of 4 (40 x the default) moves the sphere pair by several voxels per iteration: the chain launch flags it,
    the finalize pass behind it leaves the caller's live field alone, the engine repeats the call with per-iteration
    launches -- the result is the per-iteration one, and the optimizer stays on that path"""
    from levelsetfusion_python_amd import device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live0 = sphere_pair(80, 3, "cuda")  # its band touches no face of the array: one INTERIOR list
    kw = dict(level_set_term_enabled=True, maximum_warp_length_lower_threshold=0.0, gradient_descent_rate=4.0)
    args = dict(field_size=80, compute_method=lsf.ComputeMethod.DIRECT,
                smoothing_term_method=lsf.SmoothingTermMethod.KILLING, max_iterations=5, min_iterations=5,
                check_interval=5, **kw)
    with _Env(LSF_CHAIN="0"):
        ref = lsf.SlavchevaOptimizer3d(**args)
        live_ref = live0.clone()
        ref.optimize(live_ref, canonical)
    assert max(ref.log.max_warps) > 2.0
    seen = []
    original = dev.StateChain.launch

    def spying(self, first, count):
        ok = original(self, first, count)
        seen.append(self)
        return ok
    dev.StateChain.launch = spying
    try:
        with _Env(LSF_CHAIN="1"):
            opt = lsf.SlavchevaOptimizer3d(**args)
            live = live0.clone()
            opt.optimize(live, canonical)
            assert len(seen) == 1 and int(seen[0].scratch[1].item()) == 1 and int(seen[0].scratch[0].item()) == 0
            assert opt._engine._chain_disabled
            live2 = live0.clone()
            opt.optimize(live2, canonical)  # no second attempt
            assert len(seen) == 1
    finally:
        dev.StateChain.launch = original
    for got in (live, live2):
        assert torch.equal(got, live_ref)
    assert torch.equal(opt.warp_field, ref.warp_field)
    assert np.array_equal(np.float32(opt.log.max_warps), np.float32(ref.log.max_warps))
