"""The fused iteration over BOXES (lsf_slavcheva_state_iteration_boxes) against the list walk (lsf_slavcheva_state_iteration)
on the same pair: `iters` ping-pong iterations each, states and records compared, HIP-event time per launch (best of 4).
usage: box_kernel_ab.py [size] [iterations] [sphere|depth]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
data = sys.argv[3] if len(sys.argv) > 3 else "sphere"
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
if os.environ.get("ENERGY", "1") == "0":  # measurements: the kernels without the energy sums
    eng.params.energy_mode = _lib.ENERGY_NONE
if data == "depth":
    from levelsetfusion_python_amd.synthetic import depth_pair
    c, l = depth_pair(n, "cuda")
else:
    c, l = sphere_pair(n, 3, "cuda")
grid = dev.make_grid((n, n, n))
prep = dev.StatePrepare(l, c, grid)
bands, _ = prep.collect()
band = [b for b in bands if b.subset == _lib.BAND_INTERIOR][0]
L = _lib.lib
box_scratch = torch.empty(int(L.lsf_band_boxes_scratch_elements(ctypes.byref(grid))), dtype=torch.int32, device="cuda")
count = torch.zeros(1, dtype=torch.int64, device="cuda")
_lib.check(L.lsf_band_boxes_count(ctypes.byref(grid), _lib.BAND_INTERIOR, prep._scratch.data_ptr(), box_scratch.data_ptr(), count.data_ptr(),
                                  dev.stream_ptr()), "lsf_band_boxes_count")
n_boxes = int(count.item())
boxes = torch.empty((n_boxes, 2), dtype=torch.int64, device="cuda")
_lib.check(L.lsf_band_boxes_fill(ctypes.byref(grid), _lib.BAND_INTERIOR, prep._scratch.data_ptr(), box_scratch.data_ptr(), boxes.data_ptr(),
                                 dev.stream_ptr()), "lsf_band_boxes_fill")
torch.cuda.synchronize()
# the boxes hold exactly the list's voxels
masks = boxes[:, 1]
bits = sum(((masks >> k) & 1) for k in range(64))
print("%d^3 %s: %d INTERIOR band voxels in %d boxes (fill %.1f %%); voxels in the boxes: %d" %
      (n, data, band.count, n_boxes, 100.0 * band.count / (64.0 * n_boxes), int(bits.sum().item())))
assert int(bits.sum().item()) == band.count
st0 = dev.state_pack(l, None, grid, copies=2)
c_boxed = dev.band_boxes_canonical(c, grid, boxes, n_boxes)


def run_list(st, rec):
    for i in range(iters):
        dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, i, band)


def run_boxes(st, rec):
    for i in range(iters):
        _lib.check(L.lsf_slavcheva_state_iteration_boxes(st[i % 2].data_ptr(), c_boxed.data_ptr(), st[(i + 1) % 2].data_ptr(),
                                                         ctypes.byref(grid), ctypes.byref(eng.params), None,
                                                         rec.data_ptr() + i * _lib.RECORD_BYTES, boxes.data_ptr(), n_boxes,
                                                         dev.stream_ptr()), "lsf_slavcheva_state_iteration_boxes")


out = {}
for name, run in (("list walk", run_list), ("box walk", run_boxes)):
    st = [t.clone() for t in st0]
    rec = dev.new_records(iters, "cuda")
    run(st, rec)
    torch.cuda.synchronize()
    out[name] = (st, dev.decode_records(dev.records_to_host(rec)))
    best = None
    for _ in range(4):
        rec2 = dev.new_records(iters, "cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(st, rec2)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / iters
        best = t if best is None else min(best, t)
    print("%-10s %.2f us per launch (%.3f of the HBM roofline at 52 B per band voxel)" %
          (name, best, 52.0 * band.count / (best * 1e-6) / 8e12))
(sa, ra), (sb, rb) = out["list walk"], out["box walk"]
print("states equal:", all(torch.equal(a, b) for a, b in zip(sa, sb)),
      "| maxima equal:", bool((ra["max_value"] == rb["max_value"]).all()), "| arg-max equal:",
      bool((ra["argmax"] == rb["argmax"]).all()), "| energies within 1e-12:",
      all(bool(abs(ra[k] - rb[k]).max() <= 1e-12 * abs(ra[k]).max()) for k in ("data_energy", "smoothing_energy", "level_set_energy")))
