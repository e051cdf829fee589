"""Host-side cost per iteration of the N > 1 enqueue path over the REAL transport API (torch.distributed "nccl" = RCCL)
on ONE GPU: a world of size 1 whose rank plays the middle slab of three and sends both halo faces to ITSELF (RCCL
allows self send / recv inside a group).  The received data are meaningless; what is exercised and timed is the exact
call sequence of the N > 1 path on device tensors.
Usage: [HALO=8] [ITERS=50] [FIXED_ONLY=1] [LB_TIMELINE=1] [LB_PROFILE=1] [PATTERN=faces|centered] slab_nccl_loopback.py [n]
n = edge of the slab (default 32: a tiny volume, wall time per iteration = host time; 256 = the bench's slab).
PATTERN (round 4): faces (default) = the bench's weak-scaling input, spheres centred ON the slab faces, so that the compact
faces carry their band voxels (~15 % of a face); centered = round 3's input, one sphere inside the slab, whose compact
faces are EMPTY (what profiles/r01g / r01h / r03_slab_rccl_loopback.txt measured: sending nothing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import levelsetfusion_python_amd as lsf
from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
from levelsetfusion_python_amd.synthetic import sphere_pair


class SelfComm(SlabComm):
    """middle rank of three; both neighbours are this very rank"""
    def native_identity(self):
        return 0, 1, 0, 0


n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(os.environ.get("ITERS", "200"))
layout = SlabLayout(3 * n, 1, 3, int(os.environ.get("HALO", "4")))
comm = SelfComm(layout)
assert comm.active and not comm.stage_through_host
sl = layout.local_slice()
pattern = os.environ.get("PATTERN", "faces")
canonical, live0 = sphere_pair(n, 3, "cuda", (sl.start, sl.stop), n // 2 if pattern == "faces" else 0)
_band = ~((live0.abs() == 1.0) & (canonical.abs() == 1.0))
_h = layout.halo
print("pattern %s: band voxels in the %d boundary slices of the lower / upper face: %d / %d of %d (%.1f %% / %.1f %%)"
      % (pattern, _h, int(_band[layout.z_begin:layout.z_begin + _h].sum()), int(_band[layout.z_end - _h:layout.z_end].sum()),
         _h * n * n, 100.0 * float(_band[layout.z_begin:layout.z_begin + _h].float().mean()),
         100.0 * float(_band[layout.z_end - _h:layout.z_end].float().mean())), flush=True)
print("transport:", "native RCCL (lsf_slab_state_iteration)" if comm.native() is not None else "torch.distributed",
      flush=True)
from levelsetfusion_python_amd.engine import SlavchevaEngine
_orig = SlavchevaEngine._enqueue_state_iteration
_host = [0.0, 0]
def _timed(self, *a, **k):
    t = time.perf_counter()
    _orig(self, *a, **k)
    _host[0] += time.perf_counter() - t
    _host[1] += 1
SlavchevaEngine._enqueue_state_iteration = _timed
_plan = {}
def _wrap_plan(name):
    fn = getattr(SlavchevaEngine, name)
    def inner(self, *a, **k):
        t = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            _plan[name] = _plan.get(name, 0.0) + time.perf_counter() - t
    setattr(SlavchevaEngine, name, inner)
for _name in ("_plan_slab", "_plan_compact_faces"):  # the second runs inside the first
    _wrap_plan(_name)
_marks = []
if os.environ.get("LB_TIMELINE") == "1":  # start / end of the host-side phases of every optimize() call
    from levelsetfusion_python_amd import device as _dev, engine as _eng, slab as _slab
    def _mark(owner, name):
        fn = getattr(owner, name)
        def inner(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                _marks.append((getattr(owner, "__name__", str(owner)) + "." + name, t, time.perf_counter()))
        setattr(owner, name, inner)
    for _o, _n in ((_dev.StatePrepare, "__init__"), (_dev.StatePrepare, "collect"), (_dev, "new_records"),
                   (_dev.IterationLauncher, "__init__"), (SlavchevaEngine, "_plan_slab"),
                   (SlavchevaEngine, "_plan_compact_faces"), (_eng.SlavchevaOutcome, "enqueue_finalize"),
                   (_slab.SlabComm, "gather_records"), (_dev, "decode_records"), (_eng.SlavchevaOutcome, "finalize")):
        _mark(_o, _n)
def make(fixed, library_run):
    kw = dict(maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters if fixed else 1)
    return lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                    smoothing_term_method=lsf.SmoothingTermMethod.KILLING, check_interval=50,
                                    comm=comm, engine_options=dict(library_run=library_run), **kw)


def timed_call(opt, live_in, canon):
    live = live_in.clone()
    torch.cuda.synchronize()
    del _marks[:]
    t0 = time.perf_counter()
    opt.optimize(live, canon)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, t0


# ROUND 6: the whole slab call enqueued by the library (lsf_slab_run_begin / _finish, the default) against the call
# enqueued iteration by iteration from Python (library_run=False: rounds 3-5) and the single-GPU call, ALTERNATING on one box
c1, l1 = sphere_pair(n, 3, "cuda")
single = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                  smoothing_term_method=lsf.SmoothingTermMethod.KILLING, check_interval=50,
                                  maximum_warp_length_lower_threshold=0.0, max_iterations=iters, min_iterations=iters)
contenders = [("slab, library-enqueued call (lsf_slab_run_*)", make(True, True), live0, canonical),
              ("slab, one call per iteration (lsf_slab_state_iteration)", make(True, False), live0, canonical),
              ("single GPU (lsf_state_run_*)", single, l1, c1)]
times = {name: [] for name, *_ in contenders}
for rep in range(int(os.environ.get("REPS", "6"))):
    for name, opt, live_in, canon in contenders:
        dt, t0 = timed_call(opt, live_in, canon)
        if rep:
            times[name].append(dt * 1e3)
        if rep == 1:
            for _label, _a, _b in _marks:
                print("      %-44s %8.1f -> %8.1f us" % (_label, (_a - t0) * 1e6, (_b - t0) * 1e6))
        if rep == 0 and opt is not single:
            _f = opt.engine._fast
            print("    %s: library_run %s, exchange interval %d, compact faces %s" % (
                name, opt.engine.last_call.library_run, _f.exchange_interval,
                getattr(_f, "compact_faces", getattr(_f, "faces_ref", None) is not None)), flush=True)
            if getattr(_f, "faces", None) is not None:
                print("    compact faces: send %s / recv %s band voxels = %s / %s bytes per exchange (whole faces: 2 x %d bytes)"
                      % (list(_f.faces.send_count), list(_f.faces.recv_count), [16 * int(c) for c in _f.faces.send_count],
                         [16 * int(c) for c in _f.faces.recv_count], 16 * _h * n * n), flush=True)
for name, ts in times.items():
    print("%-58s %d iterations: whole optimize() %s ms (median %.3f)" % (name, iters, " ".join("%.3f" % t for t in ts),
                                                                         sorted(ts)[len(ts) // 2]), flush=True)
med = {name: sorted(ts)[len(ts) // 2] for name, ts in times.items()}
names = [name for name, *_ in contenders]
print("single GPU / slab, library-enqueued: %.3f   single GPU / slab, call per iteration: %.3f"
      % (med[names[2]] / med[names[0]], med[names[2]] / med[names[1]]), flush=True)
print("host time inside the per-iteration enqueue calls: %.1f us per iteration; launch plan %.0f us per call, of which the "
      "compact-face plan %.0f us" % (_host[0] / max(_host[1], 1) * 1e6, _plan.get("_plan_slab", 0.0) * 1e6,
                                    _plan.get("_plan_compact_faces", 0.0) * 1e6))
if os.environ.get("FIXED_ONLY") != "1":
    gated = make(False, True)
    for rep in range(3):
        dt, _ = timed_call(gated, live0, canonical)
        print("gated (MAX all-reduce per iteration, Python path): %d iterations, %.1f us per iteration (whole optimize %.2f ms)"
              % (len(gated.log.max_warps), dt / max(len(gated.log.max_warps), 1) * 1e6, dt * 1e3), flush=True)
if os.environ.get("LB_PROFILE") == "1":
    import cProfile, pstats
    live = live0.clone()
    pr = cProfile.Profile()
    pr.enable()
    contenders[0][1].optimize(live, canonical)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
comm.close()
dist.destroy_process_group()
