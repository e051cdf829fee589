"""the INTERIOR list kernel on the same band voxels in different LIST ORDERS: ascending (what lsf_state_prepare builds) and
patches of PZ slices x PY rows (patches in z-major order, ascending inside); per order the kernel's time for several
group sizes is measured with variant builds (tools/build_variant.sh g16 - -DLSF_LIST_GROUP=16, then LSF_HIP_LIBRARY=...
variants/g16.so: the group size is a compile-time constant since round 5).  States must come out equal.
usage: list_order_probe.py [size] [PZ] [PY]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, device as dev
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PZ = int(sys.argv[2]) if len(sys.argv) > 2 else 4
PY = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
grid = dev.make_grid((n, n, n))
c, l = sphere_pair(n, 3, "cuda")
prep = dev.StatePrepare(l, c, grid)
bands, _ = prep.collect()
st = prep.states
band = [b for b in bands if b.subset == _lib.BAND_INTERIOR][0]
idx = band.indices[:band.count].to(torch.int64)
z, y, x = idx // (n * n), (idx // n) % n, idx % n
key = (((z // PZ) * (n // PY) + y // PY) * PZ + z % PZ) * PY + y % PY
order = torch.argsort(key * n + x)
patched = dev.BandList(idx[order].to(torch.int32).contiguous(), band.count, band.subset)
rec = dev.new_records(2, "cuda")


def run(b, s):
    for i in range(50):
        dev.slavcheva_state_iteration(s[i % 2], c, s[(i + 1) % 2], grid, eng.params, None, rec, 0, b)


outs = []
for name, b in (("ascending", band), ("patches %d x %d" % (PZ, PY), patched)):
    s = [t.clone() for t in st]
    run(b, s)
    torch.cuda.synchronize()
    outs.append(s)
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        run(b, s)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / 50
        best = t if best is None else min(best, t)
    print("library %s, %-14s: %.2f us per launch" % (os.path.basename(os.environ.get("LSF_HIP_LIBRARY", "default")), name, best))
print("states equal:", all(torch.equal(a, b) for a, b in zip(*outs)))
