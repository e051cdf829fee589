import sys, time
sys.path.insert(0, "/root/repo")
import torch, gc
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd import _lib, engine
from levelsetfusion_python_amd.synthetic import sphere_pair
n = 256
canonical, live0 = sphere_pair(n, 3, "cuda")
opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                               smoothing_term_method=lsf.SmoothingTermMethod.KILLING, maximum_warp_length_lower_threshold=0.0,
                               max_iterations=50, min_iterations=50, check_interval=50)
marks = {}
class Lib:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in ("lsf_state_run_begin", "lsf_state_run_finish"):
            def inner(*a):
                t0 = time.perf_counter(); r = fn(*a); marks[name] = (t0, time.perf_counter()); return r
            return inner
        return fn
engine._lib.lib = Lib(_lib.lib)
stamps = {}
def stamp_init(cls, name):
    orig = cls.__init__
    def init(self, *a, **k):
        stamps.setdefault(name + " in", time.perf_counter())
        orig(self, *a, **k)
        stamps[name + " out"] = time.perf_counter()
    cls.__init__ = init
from levelsetfusion_python_amd import device as dev
stamp_init(dev.BandList, "BandList")
stamp_init(engine._Counted, "_Counted")
stamp_init(engine._RunOutcome, "_RunOutcome")
stamp_init(engine._RunLog, "_RunLog")
orig_opt = engine.SlavchevaEngine.optimize
def eng_opt(self, *a, **k):
    r = orig_opt(self, *a, **k); stamps["engine.optimize out"] = time.perf_counter(); return r
engine.SlavchevaEngine.optimize = eng_opt
detail = {}
live = torch.empty_like(live0)
gc.collect(); gc.freeze(); gc.disable()
acc = [0.0] * 5
N = 200
for k in range(N + 10):
    t0 = time.perf_counter()
    live.copy_(live0)
    t1 = time.perf_counter()
    opt.optimize(live, canonical)
    t2 = time.perf_counter()
    if k >= 10:
        b0, b1 = marks["lsf_state_run_begin"]; f0, f1 = marks["lsf_state_run_finish"]
        acc[0] += t1 - t0; acc[1] += b0 - t1; acc[2] += f0 - b1; acc[3] += t2 - f1; acc[4] += t2 - t0
        prev = f1
        for key in ("BandList in", "BandList out", "_Counted out", "_RunLog out", "_RunOutcome in", "_RunOutcome out", "engine.optimize out"):
            if key in stamps:
                detail[key] = detail.get(key, 0.0) + stamps[key] - prev
                prev = stamps[key]
        detail["optimize() out"] = detail.get("optimize() out", 0.0) + t2 - prev
    stamps.clear()
print("per step, us: copy_ call %.1f | optimize() entry -> run_begin %.1f | between begin and finish (host) %.1f | after run_finish returned -> optimize() returned %.1f | whole step %.1f"
      % tuple(1e6 * a / N for a in acc))
print("after run_finish returned, us between: " + " | ".join("%s %.1f" % (k, 1e6 * v / N) for k, v in detail.items()))
