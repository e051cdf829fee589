"""Helper of test_gpu_slab_two_ranks.py::test_native_rccl_transport_equals_torch_transport (run as a script in its own
process): a torch.distributed "nccl" world of ONE rank plays the middle slab of three and exchanges both halo faces with
ITSELF (RCCL allows self send / recv inside a group).  The physics is meaningless (a slab that is its own neighbour); the
point is that the library's RCCL transport (lsf_slab_state_iteration) and the torch.distributed transport move exactly
the same bytes at exactly the same points of the iteration, so the two runs must agree bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    from levelsetfusion_python_amd.synthetic import sphere_pair

    class SelfComm(SlabComm):
        def native_identity(self):
            return 0, 1, 0, 0


    n, halo = 48, 2
    layout = SlabLayout(3 * n, 1, 3, halo)
    sl = layout.local_slice()
    canonical, live0 = sphere_pair(n, 3, "cuda", (sl.start, sl.stop))
    kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                  smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0,
                  max_iterations=12, min_iterations=12, check_interval=5)
    results = {}
    # ..._sparse: the ping-pong states initialised near the band only (by default from 2^21 voxels on; here at 48^2 x 52)
    for tag, transport, faces in (("rccl", "rccl", "compact"), ("rccl_full", "rccl", "full"), ("torch", "torch", "full"),
                                  ("rccl_sparse", "rccl", "compact"), ("rccl_full_sparse", "rccl", "full")):
        os.environ["LSF_SLAB_TRANSPORT"] = transport
        os.environ["LSF_SLAB_FACES"] = faces
        options = dict(sparse_min_voxels=0) if tag.endswith("_sparse") else {}
        comm = SelfComm(layout)
        used = "rccl" if comm.native() is not None else "torch"
        if tag == "rccl":
            # a call of exactly one exchange group plans its faces (and starts the count collective) but never exchanges:
            # the next call on the same communicator must not trip over the collective left in flight
            short = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, engine_options=options,
                                             **dict(kwargs, max_iterations=halo, min_iterations=halo))
            short.optimize(live0.clone(), canonical)
            assert len(short.log.max_warps) == halo
        opt = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, engine_options=options, **kwargs)
        live = live0.clone()
        opt.optimize(live, canonical)
        if tag == "rccl":
            used += ":compact" if opt.engine._fast.faces_ref is not None else ":full"
        own = layout.owned_local()
        results[tag] = dict(used=used, live=live[own].cpu().numpy(), warp=opt.warp_field[own].cpu().numpy(),
                            max_warps=np.float32(opt.log.max_warps), data=np.float64(opt.log.data_energies))
        comm.close()
    dist.destroy_process_group()
    # the physics: sphere_pair is periodic in z with period n, so the stack of three slabs is three copies of this slab;
    # a rank that is its own neighbour reproduces the MIDDLE slab of the whole 3n-slice volume as long as the outer
    # domain boundaries (n slices away, influence spreads <= 2 slices per iteration) have not reached it -- up to the
    # float32 rounding of "global z + displacement", which differs between the periods, hence a
    # tolerance here and not bit-equality
    whole_c, whole_l = sphere_pair(n, 3, "cuda", (0, 3 * n))
    ref = lsf.SlavchevaOptimizer3d(field_size=n, **kwargs)
    ref._run_checks = lambda *a: None  # the stacked volume is not a cube
    ref.optimize(whole_l, whole_c)
    ref_live, ref_warp = whole_l[n:2 * n].cpu().numpy(), ref.warp_field[n:2 * n].cpu().numpy()
    a, b, c = results["rccl"], results["rccl_full"], results["torch"]
    sparse_differs = [t + ":" + k for t in ("rccl_sparse", "rccl_full_sparse") for k in ("live", "warp", "max_warps")
                      if not np.array_equal(a[k], results[t][k])]
    sparse_differs += [t + ":data" for t in ("rccl_sparse", "rccl_full_sparse")  # float64 sums by atomics: order varies
                       if not np.allclose(a["data"], results[t]["data"], rtol=1e-10)]
    np.savez(out_path, used_rccl=a["used"], used_torch=c["used"], sparse_differs=",".join(sparse_differs),
             live_equal=np.array_equal(a["live"], c["live"]) and np.array_equal(a["live"], b["live"]),
             warp_equal=np.array_equal(a["warp"], c["warp"]) and np.array_equal(a["warp"], b["warp"]),
             max_equal=np.array_equal(a["max_warps"], c["max_warps"]) and np.array_equal(a["max_warps"], b["max_warps"]),
             data_close=np.allclose(a["data"], c["data"], rtol=1e-10) and np.allclose(a["data"], b["data"], rtol=1e-10),
             whole_live_diff=float(np.abs(a["live"] - ref_live).max()),
             whole_warp_diff=float(np.abs(a["warp"] - ref_warp).max()),
             moved=float(np.abs(a["live"] - live0[layout.owned_local()].cpu().numpy()).max()))


def main_library_run(out_path, port):
    """the call of a z-slab rank enqueued by the LIBRARY (lsf_slab_run_begin / _finish, engine option library_run) against the
    same call made iteration by iteration from Python (lsf_slab_state_iteration), both on the native RCCL transport in the
    one-GPU loop-back: exchange groups with and without a remainder, calls too short to exchange, an exchange per iteration,
    end ranks (one neighbour), band voxels on the x faces (BOUNDARY lists, merged face lists), whole and compact faces,
    sparse states; and once against the torch.distributed transport.  Everything must agree bit for bit (energies: 1e-10)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    from levelsetfusion_python_amd.synthetic import sphere_pair

    def comm_class(identity):
        class SelfComm(SlabComm):
            def native_identity(self):
                return identity
        return SelfComm

    n = 64  # slices of 4096 voxels: a multiple of the counting pass's 1024-voxel chunks
    kwargs = dict(compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                  smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0)
    # name: (halo, iterations, rank of the layout among three slabs, (rank, world, lower, upper) in the communicator, roll x)
    cases = {"groups": (2, 12, 1, (0, 1, 0, 0), False), "remainder": (4, 13, 1, (0, 1, 0, 0), False),
             "too_short_to_exchange": (4, 3, 1, (0, 1, 0, 0), False), "one_exchange": (4, 5, 1, (0, 1, 0, 0), False),
             "lower_end_rank": (2, 9, 0, (0, 1, -1, 0), False), "upper_end_rank": (2, 9, 2, (0, 1, 0, -1), False),
             "band_on_x_faces": (2, 8, 1, (0, 1, 0, 0), True), "exchange_every_iteration": (1, 6, 1, (0, 1, 0, 0), False)}
    # (with the level-set term a band voxel ON an array face moves more than a voxel per iteration -- its out-of-bounds
    # neighbour reads 1, level_set_term.py:28-64 -- and the call would be re-run on a wider slab, which a rank that is its own
    # neighbour cannot do: that case runs the data and Killing terms only)
    case_kwargs = {"band_on_x_faces": dict(level_set_term_enabled=False)}
    problems, taken = [], []
    for name, (halo, iterations, rank, identity, roll) in cases.items():
        layout = SlabLayout(3 * n, rank, 3, halo)
        sl = layout.local_slice()
        # spheres centred ON the slab faces, so that the faces carry band voxels -- except for the end ranks: a rank that is
        # its own only neighbour receives its own boundary slices as its halo, which is consistent data only where nothing
        # moves (spheres inside the slab: the compact faces are empty, whole faces carry the truncated values)
        canonical, live0 = sphere_pair(n, 3, "cuda", (sl.start, sl.stop), n // 2 if rank == 1 else 0)
        if roll:  # the band crosses x = 0 and x = n - 1: BOUNDARY list entries in every slice, faces merged from two lists
            canonical, live0 = (torch.roll(t, n // 2, dims=2).contiguous() for t in (canonical, live0))
        own = layout.owned_local()
        runs = {}
        variants = [("python", "rccl", "compact", dict(library_run=False)), ("library", "rccl", "compact", {}),
                    ("library_whole_faces", "rccl", "full", {}), ("library_sparse", "rccl", "compact", dict(sparse_min_voxels=0))]
        if name in ("groups", "band_on_x_faces"):
            variants.append(("torch", "torch", "full", {}))
        for tag, transport, faces, options in variants:
            os.environ["LSF_SLAB_TRANSPORT"] = transport
            os.environ["LSF_SLAB_FACES"] = faces
            comm = comm_class(identity)(layout)
            opt = lsf.SlavchevaOptimizer3d(field_size=n, comm=comm, engine_options=options, max_iterations=iterations,
                                           min_iterations=iterations, **dict(kwargs, **case_kwargs.get(name, {})))
            live = live0.clone()
            try:
                opt.optimize(live, canonical)
            except Exception as exc:  # noqa: BLE001 -- say which case it was
                raise RuntimeError("case %s / %s failed" % (name, tag)) from exc
            if tag.startswith("library"):
                taken.append(bool(opt.engine.last_call.library_run))
                # (a rank that is its own ONLY neighbour pairs its lower boundary with its lower halo: the cross-check, made for
                # a periodic stack -- lower boundary against upper halo --, refuses the counts and whole faces travel, from
                # Python and from the library alike)
                want = -1 if iterations <= halo else (0 if faces == "full" or rank != 1 else 1)
                if opt.engine._fast.compact_faces != want:
                    problems.append("%s/%s: compact_faces %d" % (name, tag, opt.engine._fast.compact_faces))
                if tag == "library_sparse" and halo >= 2 and not opt.engine.last_call.sparse_states:
                    problems.append("%s/%s: did not run on sparse states" % (name, tag))
            elif opt.engine.last_call.library_run:
                problems.append("%s/%s: took the library run" % (name, tag))
            runs[tag] = dict(live=live[own].cpu().numpy(), warp=opt.warp_field[own].cpu().numpy(),
                             gradient=np.asarray(opt.gradient_field)[own.start:own.stop],
                             max_warps=np.float32(opt.log.max_warps), where=np.int64(opt.log.max_warp_locations),
                             data=np.float64(opt.log.data_energies), smoothing=np.float64(opt.log.smoothing_energies),
                             level_set=np.float64(opt.log.level_set_energies))
            comm.close()
        base = runs["python"]
        if not (len(base["max_warps"]) == iterations and float(base["max_warps"].max()) < 1.0
                and float(base["max_warps"].min()) > 0.0):
            problems.append("%s: maxima %r" % (name, base["max_warps"]))
        if roll and not np.any(np.abs(base["warp"][:, :, 0]) > 0):
            problems.append("%s: nothing moves on the x face" % name)
        for tag, r in runs.items():
            for key in ("live", "warp", "gradient", "max_warps", "where"):
                if not np.array_equal(r[key], base[key]):
                    problems.append("%s/%s: %s differs" % (name, tag, key))
            for key in ("data", "smoothing", "level_set"):
                if not np.allclose(r[key], base[key], rtol=1e-10, atol=0.0):
                    problems.append("%s/%s: %s energies differ" % (name, tag, key))
    dist.destroy_process_group()
    np.savez(out_path, problems="; ".join(problems), taken=np.array(taken))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "library_run":
        main_library_run(sys.argv[1], int(sys.argv[2]))
    else:
        main(sys.argv[1], int(sys.argv[2]))
