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

        def exchange_state(self, state):  # the torch.distributed transport with both neighbours = this rank
            L = self.layout
            h = L.halo
            ops = [dist.P2POp(dist.isend, state[L.z_begin:L.z_begin + h], 0, self.group),
                   dist.P2POp(dist.irecv, state[L.z_begin - h:L.z_begin], 0, self.group),
                   dist.P2POp(dist.isend, state[L.z_end - h:L.z_end], 0, self.group),
                   dist.P2POp(dist.irecv, state[L.z_end:L.z_end + h], 0, self.group)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    n, halo = 48, 2
    layout = SlabLayout(3 * n, 1, 3, halo)
    sl = layout.local_slice()
    canonical, live0 = sphere_pair(n, 3, "cuda", (sl.start, sl.stop))
    results = {}
    for transport in ("rccl", "torch"):
        os.environ["LSF_SLAB_TRANSPORT"] = transport
        comm = SelfComm(layout)
        used = "rccl" if comm.native() is not None else "torch"
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                       level_set_term_enabled=True,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       maximum_warp_length_lower_threshold=0.0, max_iterations=12, min_iterations=12,
                                       check_interval=5, comm=comm)
        live = live0.clone()
        opt.optimize(live, canonical)
        results[transport] = dict(used=used, live=live.cpu().numpy(), warp=opt.warp_field.cpu().numpy(),
                                  max_warps=np.float32(opt.log.max_warps),
                                  data=np.float64(opt.log.data_energies))
        comm.close()
    dist.destroy_process_group()
    a, b = results["rccl"], results["torch"]
    np.savez(out_path, used_rccl=a["used"], used_torch=b["used"],
             live_equal=np.array_equal(a["live"], b["live"]), warp_equal=np.array_equal(a["warp"], b["warp"]),
             max_equal=np.array_equal(a["max_warps"], b["max_warps"]),
             data_close=np.allclose(a["data"], b["data"], rtol=1e-10),
             moved=float(np.abs(a["live"] - live0.cpu().numpy()).max()))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
