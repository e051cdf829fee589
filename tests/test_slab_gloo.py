"""Multi-rank path on CPU: world_size 2, 3, 4 and 8 -- the node size the slab path is built for -- (middle ranks with two
neighbours), gloo backend.  The product's z-slab communicator (SlabComm: halo exchange
+ iteration-record reduction, the same code that runs over RCCL/xGMI on the GPUs) drives the ORACLE's per-iteration
step on each rank's slab (+ halo); the stitched result must equal the oracle run on the whole volume.

Bit-equal: like the HIP kernels, the oracle forms its gather positions z + w from the GLOBAL z on a slab
(SlavchevaOracle.axis0_offset; the float32 rounding of that sum depends on z's magnitude -- formed from the slab-local z the
stitched result drifts by 2e-6 with two slabs and by 1.7e-5 with eight)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, nz, halo, iterations, sobolev, out_dir, axis=0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)  # up to 8 ranks on the container's 8 cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from levelsetfusion_python_amd.slab import RECORD_SLOTS, SLOT_WORDS, SlabComm, SlabLayout
    from oracle import lsf_oracle as O
    canonical, live = O.sphere_pair(n, d=3, nz=nz)
    layout = SlabLayout(nz if axis == 0 else n, rank, world, halo, axis=axis)
    comm = SlabComm(layout)
    sl = layout.local_slice() if axis == 0 else (slice(None), layout.local_slice())
    live_l = torch.from_numpy(np.ascontiguousarray(live[sl]))
    canon_l = np.ascontiguousarray(canonical[sl])
    warp_l = torch.zeros(live_l.shape + (3,), dtype=torch.float32)
    kernel = O.generate_1d_sobolev_kernel(3, 0.1) if sobolev else None
    opt = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=not sobolev,
                            smoothing_term_method=O.TIKHONOV if sobolev else O.KILLING,
                            sobolev_smoothing_enabled=sobolev, sobolev_kernel=kernel)
    records = torch.zeros((iterations, RECORD_SLOTS * SLOT_WORDS), dtype=torch.int64)
    partial = torch.zeros_like(records)  # this rank's records before any reduction
    own = layout.owned_local() if axis == 0 else (slice(None), layout.owned_local())
    opt.max_region = own
    # global index of local slice 0 (z-slabs) / of local row 0 (slabs cut along y)
    opt.axis0_offset = layout.z_global_offset if axis == 0 else (0, layout.global_offset)
    for it in range(iterations):
        lv, wp = live_l.numpy(), warp_l.numpy()
        # the local step: everything the oracle computes within `halo` of a fake (interior) array edge is wrong and
        # is overwritten by the exchange below; owned voxels only read up to one slice into the halo (+ the gather)
        max_warp, at, en = opt.iteration(lv, canon_l, wp)
        m = np.float32(max_warp)
        at_global = (at[0] - layout.z_begin + layout.z0, at[1], at[2]) if axis == 0 else \
            (at[0], at[1] - layout.begin + layout.g0, at[2])
        flat = int(np.ravel_multi_index(at_global, (nz, n, n)))
        packed = (np.uint64(m.view(np.uint32)) << np.uint64(32)) | np.uint64((~np.uint32(flat)) & 0xFFFFFFFF)
        records[it, 0] = int(np.array([packed], np.uint64).view(np.int64)[0])
        band = ~(O.is_truncated(lv) & O.is_truncated(canon_l))
        records[it, 1] = int(np.array([float(band[own].sum())]).view(np.int64)[0])  # any per-rank partial sum
        warp_planar = warp_l.permute(3, 0, 1, 2).contiguous()
        comm.exchange_halos([live_l, warp_planar])
        warp_l.copy_(warp_planar.permute(1, 2, 3, 0))
        partial[it] = records[it]
        comm.reduce_records(records, it, it + 1)
    # the fixed-count path: one gather of every rank's slots instead of the reductions -- decodes to the same values
    from levelsetfusion_python_amd import device as dev
    gathered = comm.gather_records(partial, 0, iterations)
    assert gathered.shape == (iterations, world * RECORD_SLOTS, 4)
    a, b = dev.decode_records(gathered), dev.decode_records(dev.slot_view(records.numpy())[:, :, :4])
    for key in ("max_value", "argmax", "executed"):
        assert np.array_equal(a[key], b[key]), key
    assert np.allclose(a["data_energy"], b["data_energy"], rtol=1e-15)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), live=live_l.numpy()[own], warp=warp_l.numpy()[own],
             records=records.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nz", [(2, 16), (3, 24), (4, 32), (8, 64)])
def test_slab_run_matches_whole_volume(tmp_path, world, nz):
    from oracle import lsf_oracle as O
    n, halo, iterations, sobolev = 24, 3, 3, False
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, nz, halo, iterations, sobolev, str(tmp_path)), nprocs=world, join=True)
    canonical, live = O.sphere_pair(n, d=3, nz=nz)
    opt = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                            maximum_warp_length_lower_threshold=0.0, max_iterations=iterations,
                            min_iterations=iterations)
    opt.optimize(live, canonical)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    live_cat = np.concatenate([p["live"] for p in parts], axis=0)
    warp_cat = np.concatenate([p["warp"] for p in parts], axis=0)
    assert live_cat.shape == live.shape
    assert np.array_equal(live_cat, live)
    assert np.array_equal(warp_cat, opt.warp_field)
    # reduced records are identical on every rank and carry the global max / arg-max and the summed partials
    for p in parts[1:]:
        assert np.array_equal(parts[0]["records"], p["records"])
    rec = parts[0]["records"]
    packed = rec[:, 0].view(np.uint64)
    got_max = (packed >> np.uint64(32)).astype(np.uint32).view(np.float32)
    assert np.array_equal(got_max, np.float32(opt.log["max_warps"]))
    got_idx = (~packed.astype(np.uint32)).astype(np.int64)
    want_idx = [np.ravel_multi_index(at, live.shape) for at in opt.log["max_warp_locations"]]
    assert list(got_idx) == [int(i) for i in want_idx]
    assert np.all(rec[:, 1].view(np.float64) > 0)


@pytest.mark.parametrize("world", [2, 4])
def test_y_slab_run_matches_whole_volume(tmp_path, world):
    """the same with slabs cut along Y (SlabLayout(axis=1): strided faces through the packed staging buffers, gather
    positions formed from the GLOBAL row)"""
    from oracle import lsf_oracle as O
    n, nz, halo, iterations = 24, 16, 3, 3
    mp.spawn(_worker, args=(world, _free_port(), n, nz, halo, iterations, False, str(tmp_path), 1), nprocs=world, join=True)
    canonical, live = O.sphere_pair(n, d=3, nz=nz)
    opt = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                            maximum_warp_length_lower_threshold=0.0, max_iterations=iterations,
                            min_iterations=iterations)
    opt.optimize(live, canonical)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate([p["live"] for p in parts], axis=1), live)
    assert np.array_equal(np.concatenate([p["warp"] for p in parts], axis=1), opt.warp_field)
    packed = parts[0]["records"][:, 0].view(np.uint64)
    assert np.array_equal((packed >> np.uint64(32)).astype(np.uint32).view(np.float32), np.float32(opt.log["max_warps"]))
    want_idx = [int(np.ravel_multi_index(at, live.shape)) for at in opt.log["max_warp_locations"]]
    assert list((~packed.astype(np.uint32)).astype(np.int64)) == want_idx


def test_y_slab_layout():
    from levelsetfusion_python_amd.slab import SlabLayout
    L = SlabLayout(64, 1, 4, 2, axis=1)
    assert (L.axis, L.n_global, L.g0, L.g1, L.begin, L.end, L.n_local, L.global_offset) == (1, 64, 16, 32, 2, 18, 20, 14)
    v = torch.arange(4 * 64 * 3, dtype=torch.float32).reshape(4, 64, 3)
    assert torch.equal(L.cut(v), v[:, 14:34]) and torch.equal(L.owned_of(L.cut(v)), v[:, 16:32])
    with pytest.raises(ValueError):
        SlabLayout(64, 0, 4, 2, axis=2)


def _gather_worker(rank, world, port, nz, halo, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    layout = SlabLayout(nz, rank, world, halo)
    comm = SlabComm(layout)
    sl = layout.local_slice()
    whole = torch.arange(nz * 3 * 5 * 4, dtype=torch.float32).reshape(nz, 3, 5, 4)  # a "packed" field [z, y, x, 4]
    local = whole[sl].clone()
    local[:layout.z_begin] = -1.0  # halo contents must not travel
    local[layout.z_end:] = -1.0
    got = comm.all_gather_owned(local)
    assert torch.equal(got, whole), "rank %d" % rank
    open(os.path.join(out_dir, "ok%d" % rank), "w").close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_owned_rebuilds_the_whole_level(tmp_path, world):
    """SlabComm.all_gather_owned: every rank's OWNED slices, in rank order = the whole level on every rank (the static
    gather operand of a hierarchical slab run whose warp outgrows the halo)"""
    mp.spawn(_gather_worker, args=(world, _free_port(), 6 * world, 2, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))
