"""Host-side cost per iteration of the N > 1 enqueue path (boundary launches, event hand-off to the comm stream, packed
halo copies, interior launch) with the wire replaced by a loop-back: run on a tiny volume so that GPU time is
negligible and wall time per iteration = host time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
from levelsetfusion_python_amd.synthetic import sphere_pair


class LoopbackComm(SlabComm):
    """middle rank of 3: two neighbours; sends are dropped, receives keep whatever the staging buffer holds"""
    def __init__(self, layout):
        super().__init__(layout)
        self.stage_through_host = False

    def exchange_halos(self, tensors, width=None):
        L = self.layout
        h = L.halo if width is None else width
        channels = [1 if t.dim() == 3 else t.shape[0] for t in tensors]
        shape = (sum(channels), h) + tuple(tensors[0].shape[-2:])
        for tag, (sa, sb), (ra, rb) in (("lo", (L.z_begin, L.z_begin + h), (L.z_begin - h, L.z_begin)),
                                        ("hi", (L.z_end - h, L.z_end), (L.z_end, L.z_end + h))):
            send, recv = self._staging((tag, shape, tensors[0].dtype), shape, tensors[0])
            k = 0
            for t, c in zip(tensors, channels):
                src = self._z_view(t, sa, sb)
                send[k:k + c].copy_(src if t.dim() == 4 else src.unsqueeze(0))
                k += c
            recv.copy_(send)  # stands in for the wire
            k = 0
            for t, c in zip(tensors, channels):
                dst = self._z_view(t, ra, rb)
                dst.copy_(recv[k:k + c] if t.dim() == 4 else recv[k])
                k += c

    def exchange_live_and_warp(self, live, warp):
        from levelsetfusion_python_amd import device as dev
        L = self.layout
        h = L.halo
        shape = (1 + warp.shape[0], h) + tuple(live.shape[-2:])
        lo = self._staging(("lo", shape, live.dtype), shape, live)
        hi = self._staging(("hi", shape, live.dtype), shape, live)
        dev.halo_copy(live, warp, lo[0], hi[0], h, L.z_begin, L.z_end - h, unpack=False)
        lo[1].copy_(lo[0]); hi[1].copy_(hi[0])  # stands in for the wire
        dev.halo_copy(live, warp, lo[1], hi[1], h, L.z_begin - h, L.z_end, unpack=True)

    def reduce_max(self, records, index):
        pass

    def reduce_records(self, records, first, last, energies=True):
        pass


n, iters = 32, 200
layout = SlabLayout(3 * n, 1, 3, 2)
sl = layout.local_slice()
canonical, live0 = sphere_pair(n, 3, "cuda", (sl.start, sl.stop))
for name, comm in (("single GPU path", None), ("slab path (loop-back)", LoopbackComm(layout))):
    c, l = (canonical, live0) if comm is not None else (canonical[2:-2].contiguous(), live0[2:-2].contiguous())
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters,
                                   check_interval=iters, comm=comm)
    live = l.clone()
    opt.optimize(live, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        live.copy_(l)
        opt.optimize(live, c)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-24s %.1f us per iteration (host-bound at %d^3)" % (name, dt / iters * 1e6, n))
