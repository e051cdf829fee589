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


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
