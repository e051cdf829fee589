"""lsf_sobolev_state_update_boxes (round 5): the y pass, the z pass, the update and the re-warp of the SobolevFusion iteration
in ONE launch, box by box through LDS, must leave exactly what lsf_convolve_axis_listed4(axis 1) followed by
lsf_sobolev_state_update(axis 2) over one list of the whole band leave -- every bit of the new state and of the final
gradient at every voxel, and the same record -- on a sphere band, on a band that touches the volume's faces (footprints and
shells that stick out of the array), on bands with holes (listed voxels around unlisted ones: an unlisted voxel's y-filtered
value is zero, whatever its neighbours hold), with updates beyond one voxel (the re-warp falls back to the gather), for
3 / 5 / 7 / 9 taps, float32-valued and arbitrary float64 taps.  Through the C ABI, as the engine calls it; then through the
optimizer class (boxes against lists, 64^3 and at the bench's full size).
Reference arithmetic: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:208-236, math_utils/convolution.py:94-132,
nonrigid_opt/field_warping.py:112-151."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def _fields(kind, n):
    from levelsetfusion_python_amd.synthetic import depth_pair, sphere_pair
    g = torch.Generator(device="cuda").manual_seed(11)
    if kind == "sphere":
        canonical, live = sphere_pair(n, 3, "cuda")
    elif kind == "depth":  # the band is a sheet that reaches the x and y faces
        canonical, live = depth_pair(n, "cuda")
    else:  # "holes": a truncated random field -- isolated band voxels, short runs, voxels on every face
        canonical = torch.rand((n, n, n), device="cuda", generator=g) * 4.0 - 2.0
        live = torch.rand((n, n, n), device="cuda", generator=g) * 4.0 - 2.0
        canonical, live = canonical.clamp(-1.0, 1.0), live.clamp(-1.0, 1.0)
    warp = (torch.rand((3, n, n, n), device="cuda", generator=g) - 0.5) * 0.6
    return canonical.contiguous(), live.contiguous(), warp.contiguous()


def _taps(n_taps, float32_valued):
    k = np.exp(-0.5 * (np.arange(n_taps) - n_taps // 2) ** 2 / 1.7) * (1.0 + 0.07 * np.arange(n_taps))  # not symmetric
    k = k / k.sum()
    return np.ascontiguousarray(k.astype(np.float32).astype(np.float64) if float32_valued else k + 1e-11)


@pytest.mark.parametrize("kind,n,rate", [("sphere", 64, 0.1), ("depth", 64, 0.1), ("holes", 40, 0.1), ("sphere", 48, 40.0),
                                         ("holes", 24, 25.0)])
@pytest.mark.parametrize("n_taps,float32_valued", [(7, True), (7, False), (3, True), (5, False), (9, True)])
def test_boxes_equal_y_pass_then_update(lsf, kind, n, rate, n_taps, float32_valued):
    from levelsetfusion_python_amd import _lib, device as dev
    canonical, live, warp = _fields(kind, n)
    grid = dev.make_grid(live.shape)
    whole_grid = dev.full_range(grid)
    state = dev.state_pack(live, warp, grid, copies=1)[0]
    prepared = dev.StatePrepare(live, canonical, whole_grid)
    bands, _ = prepared.collect()
    bands = [b for b in bands if b.count]
    if kind != "sphere":
        assert len(bands) == 2, "this pair is meant to have band voxels on the volume's faces"
    whole = torch.sort(torch.cat([b.indices[:b.count] for b in bands])).values.contiguous()
    boxes, n_boxes = dev.band_boxes(prepared, _lib.BAND_ALL)
    # the boxes hold exactly the listed voxels
    torch.cuda.synchronize()
    bx = boxes[:n_boxes].cpu().numpy()
    origins, masks = bx[:, 0] & 0xffffffff, bx[:, 1].astype(np.uint64)
    lanes = np.arange(64)
    offs = (lanes >> 4) * n * n + ((lanes >> 2) & 3) * n + (lanes & 3)
    member = ((masks[:, None] >> lanes[None, :].astype(np.uint64)) & np.uint64(1)).astype(bool)
    assert np.array_equal(np.sort((origins[:, None] + offs[None, :])[member]), whole.cpu().numpy())
    taps = _taps(n_taps, float32_valued)
    p_taps = taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                                   sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1), level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, gradient_descent_rate=rate)
    params = ctypes.byref(opt.engine.params)
    records = dev.new_records(3, live.device)
    rec = [ctypes.c_void_p(records.data_ptr() + i * _lib.RECORD_BYTES) for i in range(3)]
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    stream = dev.stream_ptr()
    gx, gxb, gy, g_ref, g_box = (torch.zeros((n, n, n, 4), dtype=torch.float32, device="cuda") for _ in range(5))
    out_ref, out_box = state.clone(), state.clone()
    G = ctypes.byref(grid)
    _lib.check(_lib.lib.lsf_sobolev_state_gradient_x(ptr(state), ptr(canonical), ptr(gx), G, params, p_taps, n_taps, None,
                                                     rec[0], ptr(whole), whole.numel(), 0, stream), "gradient + x")
    _lib.check(_lib.lib.lsf_sobolev_state_gradient_x(ptr(state), ptr(canonical), ptr(gxb), G, params, p_taps, n_taps, None,
                                                     rec[0], ptr(whole), whole.numel(), 1, stream), "gradient + x, bricks")
    # the same values brick by brick: [bz][by][bx][lz][ly][lx]
    m = n // 4
    bricked = gx.view(m, 4, m, 4, m, 4, 4).permute(0, 2, 4, 1, 3, 5, 6).contiguous().view(n, n, n, 4)
    assert torch.equal(gxb.view(torch.int32), bricked.view(torch.int32)), "brick layout"
    _lib.check(_lib.lib.lsf_convolve_axis_listed4(ptr(gx), ptr(gy), None, G, 1, p_taps, n_taps, None, ptr(whole),
                                                  whole.numel(), stream), "y pass")
    _lib.check(_lib.lib.lsf_sobolev_state_update(ptr(gy), None, ptr(state), ptr(out_ref), ptr(g_ref), G, params, 2, p_taps,
                                                 n_taps, None, rec[1], ptr(whole), whole.numel(), 1, stream), "z pass + update")
    _lib.check(_lib.lib.lsf_sobolev_state_update_boxes(ptr(gxb), ptr(state), ptr(out_box), ptr(g_box), G, params, p_taps,
                                                       n_taps, None, rec[2], ptr(boxes), n_boxes, stream), "boxes")
    torch.cuda.synchronize()
    assert float(g_ref[..., :3].abs().max()) > 1e-4
    assert torch.equal(out_box.view(torch.int32), out_ref.view(torch.int32)), "state"
    assert torch.equal(g_box.view(torch.int32), g_ref.view(torch.int32)), "final gradient"
    dec = dev.decode_records(dev.records_to_host(records))
    assert dec["max_value"][1] == dec["max_value"][2] and dec["argmax"][1] == dec["argmax"][2]
    if rate > 1.0:
        assert dec["max_value"][2] > 1.0, "this case is meant to take the gather beyond the voxel's own neighbourhood"
    # a NULL gradient output: the same state, nothing else written
    out_box2 = state.clone()
    _lib.check(_lib.lib.lsf_sobolev_state_update_boxes(ptr(gxb), ptr(state), ptr(out_box2), None, G, params, p_taps, n_taps,
                                                       None, rec[2], ptr(boxes), n_boxes, stream), "boxes, no gradient")
    torch.cuda.synchronize()
    assert torch.equal(out_box2.view(torch.int32), out_ref.view(torch.int32))


def test_box_entry_rejects_what_it_cannot_do(lsf):
    from levelsetfusion_python_amd import _lib, device as dev
    opt = lsf.SlavchevaOptimizer3d(field_size=16, sobolev_smoothing_enabled=True,
                                   sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1))
    params = ctypes.byref(opt.engine.params)
    taps = np.ones(11, dtype=np.float64)
    p_taps = taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    one, two, three = ctypes.c_void_p(16), ctypes.c_void_p(32), ctypes.c_void_p(48)
    call = _lib.lib.lsf_sobolev_state_update_boxes
    g3 = ctypes.byref(dev.make_grid((16, 16, 16)))
    assert call(one, two, three, None, ctypes.byref(dev.make_grid((16, 16))), params, p_taps, 7, None, one, one, 1, None) \
        == -2
    assert call(one, two, three, None, ctypes.byref(dev.make_grid((16, 18, 16))), params, p_taps, 7, None, one, one, 1, None) \
        == -2
    slab = dev.make_grid((16, 16, 16))
    slab.z_begin = 4
    assert call(one, two, three, None, ctypes.byref(slab), params, p_taps, 7, None, one, one, 1, None) == -2
    assert call(one, two, three, None, g3, params, p_taps, 11, None, one, one, 1, None) == -3
    assert call(one, two, two, None, g3, params, p_taps, 7, None, one, one, 1, None) == -1   # in place
    assert call(one, two, three, one, g3, params, p_taps, 7, None, one, one, 1, None) == -1  # g_out == in
    assert call(one, two, three, None, g3, params, p_taps, 7, None, one, None, 1, None) == -1
    count = _lib.lib.lsf_band_boxes_count
    assert count(g3, _lib.BAND_BOUNDARY, one, one, one, None) == -1  # INTERIOR or ALL


def _run(lsf, canonical, live0, iterations, boxes, kind="sphere"):
    n = live0.shape[-1]
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, sobolev_smoothing_enabled=True,
                                   sobolev_kernel=lsf.generate_1d_sobolev_kernel(7, 0.1), level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING, max_iterations=iterations,
                                   min_iterations=iterations, maximum_warp_length_lower_threshold=0.0,
                                   engine_options=dict(sobolev_boxes=boxes))
    live = live0.clone()
    opt.optimize(live, canonical)
    assert opt.engine.last_call.sobolev_boxes == boxes
    return opt, live


def _same_runs(a, b):
    (oa, la), (ob, lb) = a, b
    assert torch.equal(la, lb), "live field"
    assert np.array_equal(np.float32(oa.log.max_warps), np.float32(ob.log.max_warps))
    assert oa.log.max_warp_locations == ob.log.max_warp_locations
    assert np.allclose(oa.log.data_energies, ob.log.data_energies, rtol=1e-12, atol=0.0)
    assert torch.equal(torch.as_tensor(oa.warp_field), torch.as_tensor(ob.warp_field)), "warp field"
    assert np.array_equal(oa.gradient_field, ob.gradient_field), "gradient field"
    assert float(np.abs(oa.gradient_field).max()) > 0.0


@pytest.mark.parametrize("kind,n,iterations", [("sphere", 64, 8), ("depth", 64, 6), ("sphere", 256, 3)])
def test_optimizer_on_boxes_equals_optimizer_on_lists(lsf, kind, n, iterations):
    """the whole SobolevFusion call, the box path against the list path (256^3: the size bench.py times, where the list path's
    z pass walks a strip-major list)"""
    from levelsetfusion_python_amd.synthetic import depth_pair, sphere_pair
    canonical, live0 = sphere_pair(n, 3, "cuda") if kind == "sphere" else depth_pair(n, "cuda")
    _same_runs(_run(lsf, canonical, live0, iterations, True), _run(lsf, canonical, live0, iterations, False))


def test_optimizer_on_boxes_equals_the_oracle(lsf):
    from oracle import lsf_oracle as O
    canonical, live0 = O.sphere_pair(40, d=3)
    opt, live = _run(lsf, torch.from_numpy(canonical).cuda(), torch.from_numpy(live0).cuda(), 5, True)
    ref = O.SlavchevaOracle(compute_method=O.DIRECT, smoothing_term_method=O.KILLING, level_set_term_enabled=True,
                            sobolev_smoothing_enabled=True, sobolev_kernel=O.generate_1d_sobolev_kernel(7, 0.1),
                            max_iterations=5, min_iterations=5, maximum_warp_length_lower_threshold=0.0)
    live_ref = live0.copy()
    ref.optimize(live_ref, canonical)
    assert np.array_equal(live.cpu().numpy(), live_ref)
    assert np.array_equal(opt.warp_field.cpu().numpy(), ref.warp_field)
    assert np.array_equal(np.float32(opt.log.max_warps), np.float32(ref.log["max_warps"]))
