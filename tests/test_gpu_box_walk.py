"""The fused iteration over BOXES (lsf_slavcheva_state_iteration_boxes: 4 x 4 x 4 boxes of INTERIOR band voxels, their
neighbourhoods staged through wave-private LDS by LDS-DMA; round 5) against the list walk of the same kernel body
(lsf_slavcheva_state_iteration over the INTERIOR list): every word of both ping-pong states, every iteration's maximum and
arg-max bit for bit, energies to 1e-12 (float64 atomic sums) -- on full and on sparsely initialised states (idle lanes of a
box read voxels nobody initialised), with band voxels on the faces of the array (they keep their BOUNDARY list), every term
configuration, and through the library-enqueued call with the box walk forced on.  The boxes themselves against the list:
same voxels, ascending origins.  Reference loop: nonrigid_opt/slavcheva/slavcheva_optimizer2d.py:238-330."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KILLING = dict(level_set_term_enabled=True, gradient_descent_rate=0.1, data_term_weight=1.0, smoothing_term_weight=0.2,
               isomorphic_enforcement_factor=0.1, level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.0)


@pytest.fixture(scope="module")
def lsf():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import levelsetfusion_python_amd as pkg
    return pkg


def _engine(lsf, n, **kw):
    args = dict(KILLING, smoothing_term_method=lsf.SmoothingTermMethod.KILLING)
    args.update(kw)
    return lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, **args).engine


def _walks(lsf, canonical, live, iterations, params, sparse_reach=0):
    """(states, decoded records) after `iterations` of the list walk and of the box walk on the same prepared states"""
    from levelsetfusion_python_amd import _lib, device as dev
    grid = dev.make_grid(tuple(live.shape))
    prepared = dev.StatePrepare(live, canonical, grid, sparse_reach=sparse_reach)
    bands, _ = prepared.collect()
    boxes, n_boxes = dev.band_boxes(prepared)
    canonical_boxed = dev.band_boxes_canonical(canonical, grid, boxes, n_boxes)
    interior = [b for b in bands if b.subset == _lib.BAND_INTERIOR and b.count]
    others = [b for b in bands if b.subset != _lib.BAND_INTERIOR]
    out = []
    for boxed in (False, True):
        states = [t.clone() for t in prepared.states]
        records = dev.new_records(iterations, live.device)
        for i in range(iterations):
            if boxed and n_boxes:
                dev.slavcheva_state_iteration_boxes(states[i % 2], canonical_boxed, states[(i + 1) % 2], grid, params, None,
                                                    records, i, boxes, n_boxes)
            else:
                for b in interior:
                    dev.slavcheva_state_iteration(states[i % 2], canonical, states[(i + 1) % 2], grid, params, None,
                                                  records, i, b)
            for b in others:  # voxels on the faces of the array: the general list walk, in both runs
                dev.slavcheva_state_iteration(states[i % 2], canonical, states[(i + 1) % 2], grid, params, None, records,
                                              i, b)
        out.append((states, dev.decode_records(dev.records_to_host(records))))
    return out, bands, boxes[:n_boxes], prepared


def _same(runs, bands):
    (sa, ra), (sb, rb) = runs
    listed = torch.cat([b.indices[:b.count].long() for b in bands if b.count])
    for a, b in zip(sa, sb):
        # (sparse states: words nobody initialised are not compared -- the listed voxels are what an iteration writes)
        assert torch.equal(a.view(-1, 4)[listed], b.view(-1, 4)[listed])
    assert np.array_equal(ra["max_value"], rb["max_value"]) and np.array_equal(ra["argmax"], rb["argmax"])
    for key in ("data_energy", "smoothing_energy", "level_set_energy"):
        assert np.allclose(ra[key], rb[key], rtol=1e-12, atol=0.0), key


def test_boxes_hold_the_interior_list(lsf):
    from levelsetfusion_python_amd import _lib, device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n = 64
    canonical, live = sphere_pair(n, 3, "cuda")
    live[:, :, 0] = 0.25  # band voxels on a face: BOUNDARY, in no box
    grid = dev.make_grid((n, n, n))
    prepared = dev.StatePrepare(live, canonical, grid)
    bands, _ = prepared.collect()
    boxes, count = dev.band_boxes(prepared)
    # ... and their canonical values, gathered box by box: [box][lz][ly][lx]
    boxed = dev.band_boxes_canonical(canonical, grid, boxes, count).view(-1, 4, 4, 4)
    for b in (0, count // 2, count - 1):
        o = int(boxes[b, 0]) & 0xffffffff
        x0, y0, z0 = o % n, (o // n) % n, o // (n * n)
        assert torch.equal(boxed[b], canonical[z0:z0 + 4, y0:y0 + 4, x0:x0 + 4])
    interior = [b for b in bands if b.subset == _lib.BAND_INTERIOR][0]
    origin, mask = boxes[:count, 0] & 0xffffffff, boxes[:count, 1]
    assert bool((origin[1:] > origin[:-1]).all()), "ascending origins"
    x0, y0, z0 = origin % n, (origin // n) % n, origin // (n * n)
    assert bool(((x0 % 4 == 0) & (y0 % 4 == 0) & (z0 % 4 == 0)).all()) and bool((mask != 0).all())
    voxels = []
    for lane in range(64):
        has = ((mask >> lane) & 1).bool()
        lx, ly, lz = lane & 3, (lane >> 2) & 3, lane >> 4
        voxels.append((origin + (lz * n + ly) * n + lx)[has])
    voxels = torch.sort(torch.cat(voxels)).values
    assert torch.equal(voxels, interior.indices[:interior.count].long())


@pytest.mark.parametrize("n,iterations,reach", [(32, 6, 0), (64, 12, 0), (96, 8, 2)])
def test_box_walk_equals_list_walk(lsf, n, iterations, reach):
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(n, 3, "cuda")
    runs, bands, boxes, prepared = _walks(lsf, canonical, live, iterations, _engine(lsf, n).params, sparse_reach=reach)
    assert len(boxes) > 0
    if reach:
        assert prepared.needed_fraction() < 0.9, "this case is meant to leave part of the states uninitialised"
    _same(runs, bands)


@pytest.mark.parametrize("config", ["tikhonov", "no_level_set", "thresholded_fdm", "no_energies"])
def test_box_walk_equals_list_walk_in_every_configuration(lsf, config):
    from levelsetfusion_python_amd import _lib
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n = 48
    canonical, live = sphere_pair(n, 3, "cuda")
    kw = {"tikhonov": dict(smoothing_term_method=lsf.SmoothingTermMethod.TIKHONOV),
          "no_level_set": dict(level_set_term_enabled=False),
          "thresholded_fdm": dict(data_term_method=lsf.DataTermMethod.THRESHOLDED_FDM),
          "no_energies": {}}[config]
    params = _engine(lsf, n, **kw).params
    if config == "no_energies":
        params = _lib.SlavchevaParams.from_buffer_copy(params)
        params.energy_mode = _lib.ENERGY_NONE
    runs, bands, boxes, _ = _walks(lsf, canonical, live, 5, params)
    _same(runs, bands)


def test_box_walk_with_band_voxels_on_the_faces_and_long_updates(lsf, ref_slavcheva):
    """the reference's orthographic pair swept through z: its band runs into the faces of the array (those voxels stay on
    their BOUNDARY list) and its updates are several voxels long -- every wave takes the general gather for the re-warp"""
    c2, l2 = ref_slavcheva["ortho64.canonical"], ref_slavcheva["ortho64.live"]
    canonical = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(c2, (16, 64, 64)))).cuda()
    live = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(l2, (16, 64, 64)))).cuda()
    runs, bands, boxes, _ = _walks(lsf, canonical, live, 4, _engine(lsf, 64).params)
    assert len([b for b in bands if b.count]) == 2 and len(boxes) > 0
    assert runs[0][1]["max_value"].max() > 2.0
    _same(runs, bands)


def test_the_library_enqueued_call_on_boxes(lsf):
    """optimize() with the box walk forced on (it is chosen by band size otherwise: 512^3 sphere pairs) against the same
    call on lists: live field, records, report, warp and gradient fields"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n = 64
    canonical, live0 = sphere_pair(n, 3, "cuda")
    results = []
    for boxed in (True, False):
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, max_iterations=10,
                                       min_iterations=10, smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       engine_options=dict(box_walk=boxed), **KILLING)
        live = live0.clone()
        opt.optimize(live, canonical)
        assert opt.engine.last_call.box_walk == boxed
        results.append((opt, live))
    (oa, la), (ob, lb) = results
    assert torch.equal(la, lb)
    assert np.array_equal(np.float32(oa.log.max_warps), np.float32(ob.log.max_warps))
    assert oa.log.max_warp_locations == ob.log.max_warp_locations
    assert np.allclose(oa.log.data_energies, ob.log.data_energies, rtol=1e-12, atol=0.0)
    assert vars(oa.get_convergence_report().warp_delta_statistics) == vars(ob.get_convergence_report().warp_delta_statistics)
    assert torch.equal(oa.warp_field, ob.warp_field) and np.array_equal(oa.gradient_field, ob.gradient_field)


def test_box_walk_refuses_what_it_cannot_walk(lsf):
    from levelsetfusion_python_amd import _lib, device as dev
    L = _lib.lib
    params = _engine(lsf, 32).params
    rec = dev.new_records(1, "cuda")
    for shape in ((30, 32, 32), (32, 32, 34)):  # extents that are not multiples of 4
        grid = dev.make_grid(shape)
        assert not dev.boxes_ok(grid)
        assert L.lsf_band_boxes_scratch_elements(ctypes.byref(grid)) == 0
        assert L.lsf_slavcheva_state_iteration_boxes(1, 1, 2, ctypes.byref(grid), ctypes.byref(params), None,
                                                     rec.data_ptr(), 1, 1, None) == -2  # LSF_ERR_BAD_DIMS
    grid = dev.make_grid((32, 32, 32))
    assert L.lsf_slavcheva_state_iteration_boxes(1, 1, 1, ctypes.byref(grid), ctypes.byref(params), None, rec.data_ptr(), 1,
                                                 1, None) == -1  # in place
    assert L.lsf_slavcheva_state_iteration_boxes(1, 1, 2, ctypes.byref(dev.make_grid((32, 32))), ctypes.byref(params), None,
                                                 rec.data_ptr(), 1, 1, None) == -2  # 2-D


def test_full_size_box_walk_equals_list_walk_512(lsf):
    """BASELINE's 512^3 sphere pair -- the size at which the engine picks the box walk by itself (7.0 M band voxels: 224 MB of
    listed state) -- against the same call on lists: live field, records, report (the bench's configuration, 8 iterations)"""
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n = 512
    canonical, live0 = sphere_pair(n, 3, "cuda")
    results = []
    for boxed in (None, False):
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, max_iterations=8,
                                       min_iterations=8, smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       engine_options=dict(box_walk=boxed), **KILLING)
        live = live0.clone()
        opt.optimize(live, canonical)
        assert opt.engine.last_call.box_walk == (boxed is None), "512^3: the box walk is the engine's own choice"
        results.append((opt.log, opt.get_convergence_report(), live))
        del opt
        torch.cuda.empty_cache()
    (la, ra, a), (lb, rb, b) = results
    assert torch.equal(a, b)
    assert np.array_equal(np.float32(la.max_warps), np.float32(lb.max_warps)) and la.max_warp_locations == lb.max_warp_locations
    assert np.allclose(la.data_energies, lb.data_energies, rtol=1e-12, atol=0.0)
    assert np.allclose(la.smoothing_energies, lb.smoothing_energies, rtol=1e-12, atol=0.0)
    assert vars(ra.warp_delta_statistics) == vars(rb.warp_delta_statistics)
    assert vars(ra.tsdf_difference_statistics) == vars(rb.tsdf_difference_statistics)
