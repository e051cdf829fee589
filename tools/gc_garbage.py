"""What cyclic garbage does one optimize() call leave for Python's collector?  Ten calls under gc.DEBUG_SAVEALL (the
collector keeps what it would have freed in gc.garbage), objects counted by type -- VERDICT round 4, item 7.
usage: gc_garbage.py [size] [killing|hier]"""
import collections
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.synthetic import sphere_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "killing"
c, l0 = sphere_pair(n, 3, "cuda")
if kind == "killing":
    opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                   smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                   maximum_warp_length_lower_threshold=0.0, max_iterations=50, min_iterations=50,
                                   check_interval=50)
    call = lambda: opt.optimize(l0.clone(), c)  # noqa: E731
else:
    opt = lsf.HierarchicalOptimizer3d(tikhonov_term_enabled=True, gradient_kernel_enabled=False, maximum_chunk_size=8,
                                      maximum_iteration_count=20, maximum_warp_update_threshold=0.0,
                                      tikhonov_strength=0.05, check_interval=20)
    call = lambda: opt.optimize(c, l0)  # noqa: E731
for _ in range(3):
    call()
torch.cuda.synchronize()
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
gc.disable()
before = gc.get_count()
for _ in range(10):
    call()
torch.cuda.synchronize()
allocated = gc.get_count()
found = gc.collect()
types = collections.Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage)
print("%s %d^3: collector counters %s -> %s over 10 calls; unreachable objects found by a full pass: %d" %
      (kind, n, before, allocated, found))
for name, count in types.most_common(20):
    print("  %6d  %s" % (count, name))
# who holds the cycles: for the most common container types, one example with its referents' types
shown = set()
for o in gc.garbage:
    t = type(o).__qualname__
    if t in shown or len(shown) >= 6 or t in ("tuple", "cell", "dict", "list"):
        continue
    shown.add(t)
    print("  example %s: %.200r" % (t, o))
gc.set_debug(0)
gc.garbage.clear()
