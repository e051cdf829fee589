#!/usr/bin/env python3
"""bench.py -- voxel-warp-updates/s and HBM-roofline fraction of the fused per-voxel warp-update path.

Workload (BASELINE.json config 4, the configuration the metric is quoted on): 3-D 256^3 KillingFusion-style
optimizer -- SlavchevaOptimizer3d, DIRECT, Killing regulariser (lambda 0.1, weight 0.2) + level-set term
(weight 0.2), no Sobolev filter, rate 0.1, FIXED 50 iterations -- on the synthetic "sphere pair" of SURVEY.md
section 8(d), fp32.  One *step* = one optimize(live, canonical) call = 50 iterations of the fused warp-update kernel
over one 256^3 pair: one launch of lsf_slavcheva_state_iteration per iteration and band list, all of them enqueued by the
library (lsf_state_run_begin / lsf_state_run_finish) -- plus the prepare / finalize passes and the convergence-statistics
reductions the reference also runs per call.

N > 1 (launched by torch.distributed.run, one rank per GPU; `python bench.py --gpus N` starts it itself):
  --scaling weak (default): every rank owns a 256^3 z-slab of a 256 x 256 x (256 N) volume whose spheres are centred ON
    the slab faces (--pattern faces), so that every interior face cuts a narrow band through its equator (~15 % of the
    face) and the halo exchange carries data; the band voxels per rank equal the single-GPU sphere pair's.
  --scaling strong: BASELINE config 4 as written -- ONE 256^3 sphere pair z-slabbed over the N ranks (256 / N slices each).
With an h-slice halo the band voxels of the h boundary slices of the state (live field + warp) travel to the z-neighbours
every h-th iteration (RCCL send / recv over xGMI, issued by the library: lsf_slab_state_iteration) while the interior --
and the halo-independent part of the next iteration -- is computed; the iterations in between recompute the neighbours'
slices they still have valid inputs for; the iteration records are gathered once per step.  The N > 1 line carries
`halo_band_voxels_per_face` and `halo_bytes_per_exchange` (what actually travels).

`secondary` (single GPU, default run only): short measurements of the other configurations of BASELINE.json on the
same box, so that every figure the documentation quotes is on the driver's line.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = {"killing": 52, "sobolev": 76}  # algorithmic bytes per voxel-update, fp32, 3-D
HBM_PEAK_GBS = 8000.0                    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 100 for the millisecond-scale steps -- killing, sobolev, hier2d --, else 20)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps in front (default: 10 resp. 5)")
    ap.add_argument("--size", type=int, default=None,
                    help="edge of the per-GPU volume (default 256 = BASELINE config 4; multiframe: 512 = config 5)")
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--halo", type=int, default=None,
                    help="halo slices per interior slab face = iterations per exchange group: the faces travel every "
                         "--halo iterations, the iterations in between recompute the neighbours' slices (exact while "
                         "every warp update stays below one voxel; the engine re-runs wider otherwise).  Default: 8 for "
                         "weak scaling; strong scaling picks it from the slab height (slices per rank / 8, 1..8), so that "
                         "the recomputed slices stay below ~10 %% of a slab")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank a --size^3 slab of a --size x --size x (--size N) volume (default); "
                         "strong = ONE --size^3 pair z-slabbed over the N ranks (BASELINE config 4 as written)")
    ap.add_argument("--pattern", default=None, choices=["faces", "centered"],
                    help="weak scaling input: faces = spheres centred on the slab faces, every interior face cuts a band "
                         "(default for N > 1); centered = one sphere inside every slab (no band voxel near a face: the "
                         "compact halo exchange then carries nothing)")
    ap.add_argument("--secondary-divisor", type=int, default=1,
                    help="test rigs only: run the secondary measurements at 1/divisor of their edge lengths")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short secondary measurements (512^3, hierarchical, multi-frame, SobolevFusion, 2-D)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo stages halos through the host -- only for "
                         "exercising the N > 1 flow on a box with fewer GPUs than ranks (see --share-device)")
    ap.add_argument("--share-device", action="store_true",
                    help="all ranks use cuda:0 (test rigs only; requires --backend gloo)")
    ap.add_argument("--workload", default="killing",
                    choices=["killing", "sobolev", "hier-tik", "hier-full", "multiframe", "hier2d"],
                    help="killing = BASELINE config 4 (default, the metric's configuration); multiframe = BASELINE "
                         "config 5 (--frames synthetic 512^3 frames, consecutive frames as pairs, hierarchical "
                         "Tikhonov + 7-tap kernel); the others are extra single-GPU measurements: SobolevFusion-style "
                         "Slavcheva, hierarchical Tikhonov-only, hierarchical Tikhonov + 7-tap kernel; hier2d = BASELINE "
                         "config 2 (2-D 512^2 HierarchicalOptimizer2d, 3 levels, Tikhonov, 100 iterations per level)")
    ap.add_argument("--frames", type=int, default=8, help="multiframe: frames of the synthetic sequence")
    ap.add_argument("--parallelism", default="replicas", choices=["replicas", "slab"],
                    help="multiframe on N > 1 GPUs: every rank its own sequence (pairs are independent) or every pair "
                         "z-slabbed over the ranks")
    ap.add_argument("--data", default="sphere", choices=["sphere", "depth"],
                    help="sphere: SURVEY 8(d)'s sphere pair (BASELINE's configuration, default); depth: two synthetic depth "
                         "frames -> TSDF volumes through the package's own generator (single GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-size", type=int, default=128)
    ap.add_argument("--cpu-sample-iterations", type=int, default=24)
    args = ap.parse_args()
    if args.size is None:
        args.size = 512 if args.workload in ("multiframe", "hier2d") else 256
    # a step of a few milliseconds: 20 of them are over before the card's clocks have settled (20 timed steps 1.74-1.75 ms
    # each at 256^3, 200: 1.71-1.72) -- the default run times 100 (0.2 s at 256^3, 0.8 s at 512^3)
    short_steps = args.workload in ("killing", "sobolev", "hier2d") and args.size <= 512
    if args.steps is None:
        args.steps = 100 if short_steps else 20
    if args.warmup is None:
        args.warmup = 10 if short_steps else 5
    return args


def cpu_baseline(size, iterations):
    """the numpy oracle (vectorised restatement of the reference, 1 thread) on a bounded sample of the SAME
    workload: a size^3 sphere pair, same optimizer configuration, `iterations` fixed iterations"""
    from oracle import lsf_oracle as O
    canonical, live = O.sphere_pair(size, d=3)
    opt = O.SlavchevaOracle(compute_method=O.DIRECT, level_set_term_enabled=True, smoothing_term_method=O.KILLING,
                            maximum_warp_length_lower_threshold=0.0, max_iterations=iterations,
                            min_iterations=iterations)
    t0 = time.perf_counter()
    opt.optimize(live, canonical)
    dt = time.perf_counter() - t0
    return dict(value=size ** 3 * iterations / dt, unit="voxel-warp-updates/s", cores=1, kind="port",
                sample="%d^3 sphere pair, %d fixed iterations of the same Killing+level-set configuration, "
                       "oracle/lsf_oracle.py (numpy, 1 thread), %.1f s" % (size, iterations, dt),
                host_cpus=os.cpu_count(), host_affinity=len(os.sched_getaffinity(0)))


def cpu_baseline_hierarchical(size, iterations, full):
    """the numpy oracle's HierarchicalOracle on one size^3 pair of the multi-frame sequence, same configuration"""
    from oracle import lsf_oracle as O
    canonical, live = O.sphere_frame(size, 0), O.sphere_frame(size, 1)
    opt = O.HierarchicalOracle(tikhonov_term_enabled=True, gradient_kernel_enabled=full, maximum_chunk_size=8, rate=0.1,
                               maximum_iteration_count=iterations, maximum_warp_update_threshold=0.0,
                               tikhonov_strength=0.05, kernel=O.generate_1d_sobolev_kernel(7, 0.1) if full else None)
    t0 = time.perf_counter()
    opt.optimize(canonical, live)
    dt = time.perf_counter() - t0
    updates = iterations * sum((size >> k) ** 3 for k in range(4))
    return dict(value=updates / dt, unit="voxel-warp-updates/s", cores=1, kind="port",
                sample="one %d^3 pair of the sphere sequence, 4 levels x %d fixed iterations, Tikhonov (strength 0.05)%s, "
                       "oracle/lsf_oracle.py (numpy, 1 thread), %.1f s" % (size, iterations,
                                                                          " + 7-tap kernel" if full else "", dt),
                host_cpus=os.cpu_count(), host_affinity=len(os.sched_getaffinity(0)))


def timed_steps(step, args, fence, max_over_ranks=None):
    """W untimed + K timed steps between fences (barrier + synchronize on both sides), cyclic GC parked as in main();
    returns (sum of what step() returns over the timed steps, seconds)"""
    late = min(3, args.warmup)  # the last warm-up steps run BEHIND the collection (see main())
    for _ in range(args.warmup - late):
        step()
    gc.collect()
    gc.freeze()  # keep torch's objects out of the cyclic collector's full passes
    gc.disable()
    for _ in range(late):
        step()
    fence()
    t0 = time.perf_counter()
    total = None
    for _ in range(args.steps):
        r = step()
        total = r if total is None else tuple(a + b for a, b in zip(total, r))
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if max_over_ranks is not None:
        elapsed = max_over_ranks(elapsed)
    return total, elapsed


def hier_finest_level_kernels(n, full, taps, device, reps=20):
    """HIP events around the launches ONE finest-level iteration of the hierarchical optimizer makes, on buffers of the
    level's size (the engine's own launches carry the same kernels plus the record bookkeeping): hier_iteration_kernel
    (Tikhonov; with the in-kernel update when no filter follows) and, with a gradient kernel, convolve_xyz_kernel (x, y, z
    passes + warp update in one launch).  Loop body: hierarchical_optimizer2d.py:184-225.  Returns per-launch milliseconds."""
    from levelsetfusion_python_amd import _lib, device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    canonical, live = sphere_pair(n, 3, device)
    packed = dev.pack_live_gradient(live)
    grid = dev.make_grid((n, n, n))
    warp = torch.zeros((3, n, n, n), device=device)
    F = [torch.zeros_like(warp) for _ in range(2)]
    raw = torch.zeros_like(warp) if full else None
    rec = dev.new_records(2, device)
    params = _lib.HierParams(1.0, 0.05, 0.1, 1, 0 if full else 1, 0)

    def iteration(k):
        dev.hier_iteration(packed, canonical, warp, F[k % 2], raw if full else F[(k + 1) % 2], grid, params, None, rec, 0)

    def filter_pass(k):
        dev.convolve_xyz(raw, F[(k + 1) % 2], grid, taps, None, warp, 0.1)

    def timed(*launchers):
        for k in range(3):
            for f in launchers:
                f(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(reps):
            for f in launchers:
                f(k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {"hier_iteration_kernel<3,TIK%s>" % ("" if full else ",UPDATE"): timed(iteration)}
    if full:
        out["convolve_xyz_kernel<7 taps, warp update>"] = timed(filter_pass)
        out["both, back to back"] = timed(iteration, filter_pass)
    return out


def extra_workload(args, device, world, rank, dist):
    """the other iteration kernels: SobolevFusion-style Slavcheva, hierarchical Tikhonov (+ 7-tap kernel), and
    BASELINE config 5 -- the multi-frame sequence (`multiframe`): F synthetic frames, frame k against k + 1 as
    (canonical, live) (experiment/multiframe_experiment.py:186-233; pair loop run_hierarchical_optimizer3d_multipair.py:
    403-432) through HierarchicalOptimizer3d with Tikhonov term + 7-tap gradient kernel, fixed iterations per level.
    N > 1: `replicas` -- every rank optimizes the F - 1 pairs of a sequence of its own (pairs are independent; weak
    scaling, no data-path collective) -- or `slab` -- ONE sequence, every pair z-slabbed over the ranks (halo exchange
    per iteration over RCCL)."""
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.synthetic import sphere_frame, sphere_pair
    n, iters = args.size, args.iterations
    if args.halo is None:
        args.halo = 8
    k7 = lsf.generate_1d_sobolev_kernel(7, 0.1)
    extra = {}
    comm = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if args.backend == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.workload == "hier2d":
        # BASELINE config 2: 2-D n^2 (512^2) circle pair, maximum_chunk_size 4 => 3 levels n/4, n/2, n (pyramid.py:45),
        # Tikhonov only, rate 0.1, `iters` FIXED iterations per level (threshold 0).  tikhonov_strength 0.05: with the
        # class default 0.2 the reference's recurrence diverges (4 D s >= 1) -- same kernels, same cost
        canonical, live0 = sphere_pair(n, 2, device)
        opt = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False,
                                          maximum_chunk_size=4, rate=0.1, maximum_iteration_count=iters,
                                          maximum_warp_update_threshold=0.0, tikhonov_strength=0.05, check_interval=iters)
        per_pair = iters * sum((n >> k) ** 2 for k in range(3))

        def step():
            opt.optimize(canonical, live0)
            return per_pair, per_pair
        b_alg = 48
        name = "2D %d^2 HierarchicalOptimizer2d, 3 levels (maximum_chunk_size 4), Tikhonov (tikhonov_strength 0.05), %d " \
               "fixed iterations per level" % (n, iters)
        note = "LAUNCH-BOUND, not a roofline point: a %d^2 level is %d KiB per field -- 1024 voxels per CU; since round 6 a " \
               "launch advances 32 x 32 (16 x 16 on small levels) tiles EIGHT iterations inside LDS (lsf_hier_level_run_2d: " \
               "temporal blocking, the rings around a tile recomputed), 13 launches per 100 iterations instead of 100; " \
               "read us_per_iteration (round 5: 7.5 from a HIP graph).  frac = whole-step rate x B_alg (48 B) for " \
               "completeness" % (n, n * n * 4 // 1024)
        extra["us_per_iteration_of"] = "3 levels x %d iterations" % iters
    elif args.workload == "sobolev":
        canonical, live0 = sphere_pair(n, 3, device)
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                       sobolev_smoothing_enabled=True, sobolev_kernel=k7,
                                       maximum_warp_length_lower_threshold=0.0, max_iterations=iters,
                                       min_iterations=iters, check_interval=iters)
        live = torch.empty_like(live0)

        def step():
            live.copy_(live0)
            opt.optimize(live, canonical)
            band = opt.engine._sobolev_band
            return iters * n ** 3, iters * (band.count if band is not None else n ** 3)
        b_alg = 76
        name = "3D %d^3 SobolevFusion-style SlavchevaOptimizer3d (Tikhonov + 7-tap Sobolev), %d iterations" % (n, iters)
        note = "rate over the voxels the launches VISIT (the band list; the gradient is zero elsewhere and the " \
               "zero-preserving filter keeps it there) x B_alg (76 B), both kernels of the iteration together " \
               "(sobolev_state_gradient_x_kernel: gradient + x pass over the band list; sobolev_state_box_kernel: y pass, z " \
               "pass, update and re-warp box by box through LDS)"
    else:
        full = args.workload in ("hier-full", "multiframe")
        # tikhonov_strength 0.05: the reference's recurrence diverges for strength >= 1/12 in 3-D
        slab = args.workload == "multiframe" and args.parallelism == "slab" and world > 1
        if slab:
            from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
            layout = SlabLayout(n, rank, world, args.halo)
            comm = SlabComm(layout)
        opt = lsf.HierarchicalOptimizer3d(tikhonov_term_enabled=True, gradient_kernel_enabled=full,
                                          maximum_chunk_size=8, rate=0.1, maximum_iteration_count=iters,
                                          maximum_warp_update_threshold=0.0, tikhonov_strength=0.05,
                                          kernel=k7 if full else None, check_interval=iters,
                                          **(dict(comm=comm) if comm is not None else {}))
        per_pair = iters * sum((n >> k) ** 3 for k in range(4))
        if args.workload == "multiframe":
            first = 0 if slab else rank * (args.frames - 1)  # replicas: a sequence of its own per rank
            if slab:
                sl = layout.local_slice()
                frames = [sphere_frame(n, first + k, device)[sl].contiguous() for k in range(args.frames)]
            else:
                frames = [sphere_frame(n, first + k, device) for k in range(args.frames)]

            def step():
                for k in range(args.frames - 1):  # live_frame_index = canonical_frame_index + 1
                    opt.optimize(frames[k], frames[k + 1])
                done = (args.frames - 1) * per_pair
                return (done // world, done // world) if slab else (done, done)
            name = "3D %d^3 multi-frame sequence, %d frames (%d pairs per step), HierarchicalOptimizer3d: 4 levels, " \
                   "Tikhonov (tikhonov_strength 0.05: the reference's recurrence diverges from 1/12 up in 3-D) + 7-tap " \
                   "kernel, %d fixed iterations per level" % (n, args.frames, args.frames - 1, iters)
            extra["parallelism"] = "single GPU" if world == 1 else (
                "z-slab x%d of every pair, halo %d" % (world, args.halo) if slab else
                "replicas x%d: %d independent pairs per rank" % (world, args.frames - 1))
        else:
            canonical, live0 = sphere_pair(n, 3, device)

            def step():
                opt.optimize(canonical, live0)
                return per_pair, per_pair
            name = "3D %d^3 HierarchicalOptimizer3d, 4 levels, Tikhonov (tikhonov_strength 0.05)%s, %d iterations per " \
                   "level" % (n, " + 7-tap kernel" if full else "", iters)
        b_alg = 104 if full else 68
        note = "whole-step rate x B_alg (%d B/voxel-update), all kernels of the iteration together" % b_alg
    (updates, visited), elapsed = timed_steps(step, args, fence, max_over_ranks if world > 1 else None)
    updates, visited = updates * world, visited * world
    value = updates / elapsed
    achieved = visited / elapsed * b_alg / 1e9 / world  # per GPU
    out = dict(metric="voxel-warp-updates/sec", value=value, unit="voxel-warp-updates/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
               higher_is_better=True, scaling="weak" if args.parallelism == "replicas" or world == 1 else "strong",
               vs_baseline=None, dtype="f32", data="synthetic",
               visited_voxel_updates_per_s=visited / elapsed, config=dict(workload=name, **extra),
               roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=achieved / HBM_PEAK_GBS, traffic=None, note=note))
    if visited != updates:
        out["roofline"]["dense_equivalent_gbs"] = value * b_alg / 1e9 / world
    if args.workload == "hier2d":
        out["us_per_iteration"] = elapsed / args.steps / (3 * iters) * 1e6
        # the same pyramid with the reference's DEFAULT threshold (hierarchical_optimizer2d.py:69: 0.001; on this pair no
        # level gets there within its 100 iterations, so the work is the same): the stop test is looked at launch by launch
        # on the card, one record read-back per level (round 6) -- next to the path that threshold took before (a HIP graph
        # per check_interval iterations)
        for key, blocked in (("us_per_iteration_default_threshold", True),
                             ("us_per_iteration_default_threshold_graph_path", False)):
            armed = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=False,
                                                maximum_chunk_size=4, rate=0.1, maximum_iteration_count=iters,
                                                maximum_warp_update_threshold=0.001, tikhonov_strength=0.05,
                                                engine_options=dict(blocked_levels=blocked))
            for k in range(args.warmup + args.steps):
                if k == args.warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                armed.optimize(canonical, live0)
            torch.cuda.synchronize()
            counts = armed.get_per_level_iteration_counts()
            out[key] = (time.perf_counter() - t0) / args.steps / sum(counts) * 1e6
            out["iterations_default_threshold"] = counts
        # ... and with the gradient kernel as well: the reference's default constructor (hierarchical_optimizer2d.py:63-73:
        # Tikhonov term + kernel, threshold 0.001).  Blocked levels run the filter's two passes and the update inside the
        # launch (two iterations per launch for seven taps); the graph path takes four launches per iteration
        for key, blocked in (("us_per_iteration_default_constructor", True),
                             ("us_per_iteration_default_constructor_graph_path", False)):
            full = lsf.HierarchicalOptimizer2d(tikhonov_term_enabled=True, gradient_kernel_enabled=True, kernel=k7,
                                               maximum_chunk_size=4, rate=0.1, maximum_iteration_count=iters,
                                               maximum_warp_update_threshold=0.001, tikhonov_strength=0.05,
                                               engine_options=dict(blocked_levels=blocked))
            for k in range(args.warmup + args.steps):
                if k == args.warmup:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                full.optimize(canonical, live0)
            torch.cuda.synchronize()
            out[key] = (time.perf_counter() - t0) / args.steps / sum(full.get_per_level_iteration_counts()) * 1e6
    if args.workload in ("hier-tik", "hier-full", "multiframe") and comm is None:
        # the dominant kernels of this workload: the finest level's launches alone (87.5 % of a step's voxel-updates),
        # HIP events on their stream; kernel_ms = one finest-level iteration, frac = B_alg x n^3 / kernel_ms
        opt = frames = canonical = live0 = None  # noqa: F841  (the level's buffers: 4 vector fields at n^3)
        gc.collect()
        torch.cuda.empty_cache()
        full = args.workload != "hier-tik"
        kernels = hier_finest_level_kernels(n, full, k7, device)
        per_iteration = kernels["both, back to back"] if full else next(iter(kernels.values()))
        out["roofline"].update(kernel=" + ".join(k for k in kernels if not k.startswith("both")), kernel_ms=per_iteration,
                               kernels_ms=kernels, finest_level_frac=b_alg * n ** 3 / (per_iteration * 1e-3) / 1e9 / HBM_PEAK_GBS)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload not in ("sobolev", "hier2d"):
        out["cpu_baseline"] = cpu_baseline_hierarchical(64, iters, args.workload != "hier-tik")
    if comm is not None:
        comm.close()
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a child (rendezvous on 127.0.0.1, a free port), pass its output through -- rank 0 prints the one
    JSON line -- and return its exit status.  The pair loop this scales: run_hierarchical_optimizer3d_multipair.py:403-432."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this image
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU), as a CHILD process --
            # nothing in this process has touched the GPU yet, and it never will
            raise SystemExit(self_launch(args.gpus))
        args.gpus = world
    if args.share_device:
        if args.backend != "gloo":
            raise SystemExit("--share-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    if args.workload != "killing":
        if world > 1 and args.workload != "multiframe":
            raise SystemExit("--workload %s is a single-GPU measurement" % args.workload)
        out = extra_workload(args, device, world, rank, dist if world > 1 else None)
    else:
        out = killing_workload(args, device, world, rank, local_rank, dist if world > 1 else None)
        if world == 1 and not args.no_secondary and args.data == "sphere" and \
                (args.size == 256 or args.secondary_divisor > 1):
            out["secondary"] = secondary_measurements(args, device)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


SECONDARY = (  # (workload, size, steps, warmup, iterations, extra arguments): short runs of BASELINE's other configurations
    ("killing", 512, 20, 4, 50, {}),
    ("killing-pairs", 256, 60, 4, 50, {}),
    ("killing-default", 256, 20, 4, 100, {}),
    ("hier-tik", 256, 3, 1, 50, {}),
    ("hier-full", 256, 3, 1, 50, {}),
    ("config3", 128, 10, 2, 50, {"workload": "hier-full"}),  # BASELINE config 3 at its own size (fits the Infinity Cache)
    ("multiframe", 512, 1, 1, 50, {}),
    ("sobolev", 256, 20, 4, 50, {}),
    ("hier2d", 512, 20, 4, 100, {}),
)


def secondary_measurements(args, device):
    """every other figure the documentation quotes, measured in the SAME run on the SAME box (single GPU, a few steps
    each, ~15 s together): KillingFusion at 512^3 (the other half of north_star's target), the hierarchical optimizer with
    and without the 7-tap kernel at 256^3 (BASELINE configs 3 / 2's 3-D counterparts), the 8-frame 512^3 sequence
    (config 5), the SobolevFusion iteration, and config 2 itself (2-D 512^2; launch-bound: microseconds per iteration).
    Caller shape: run_hierarchical_optimizer3d_multipair.py:403-432.  A failure of one entry is recorded, not raised:
    the primary line must still print."""
    import copy
    rows = []
    for workload, size, steps, warmup, iterations, more in SECONDARY:
        a = copy.copy(args)
        size = size // max(1, args.secondary_divisor)
        if workload == "config3":
            size = max(size, 64)  # its coarsest level (size / 8) must hold the 7-tap kernel
        a.workload, a.size, a.steps, a.warmup, a.iterations = workload, size, steps, warmup, iterations
        a.no_cpu_baseline, a.frames, a.parallelism, a.halo = True, 8, "replicas", None
        for k, v in more.items():
            setattr(a, k, v)
        row = dict(workload=workload, size=size, steps=steps, iterations=iterations)
        try:
            t0 = time.perf_counter()
            if workload == "killing-pairs":
                row.update(pairs_in_flight(a, device))
                row["wall_s"] = time.perf_counter() - t0
                rows.append(row)
                continue
            if workload == "killing-default":
                row.update(default_loop_call(a, device))
                row["wall_s"] = time.perf_counter() - t0
                rows.append(row)
                continue
            workload = a.workload  # (an entry may run another entry's workload under its own name: config3)
            if workload == "killing":
                d = killing_workload(a, device, 1, 0, device.index or 0, None, dense_walk=False)
            else:
                d = extra_workload(a, device, 1, 0, None)
            r = d["roofline"]
            row.update(config=d["config"]["workload"], ms_per_step=d["ms_per_step"], value=d["value"],
                       visited_voxel_updates_per_s=d.get("visited_voxel_updates_per_s", d["value"]),
                       frac=r["frac"], achieved_gbs=r["achieved"], kernel_ms=r.get("kernel_ms"),
                       kernel=r.get("kernel"), wall_s=None)
            for key in ("kernels_ms", "finest_level_frac"):
                if key in r:
                    row[key] = r[key]
            if "us_per_iteration" in d:
                row["us_per_iteration"] = d["us_per_iteration"]
                row["note"] = r.get("note")
                for key in ("us_per_iteration_default_threshold", "us_per_iteration_default_threshold_graph_path",
                            "iterations_default_threshold", "us_per_iteration_default_constructor",
                            "us_per_iteration_default_constructor_graph_path"):
                    row[key] = d.get(key)
            row["wall_s"] = time.perf_counter() - t0
        except Exception as exc:  # noqa: BLE001 -- recorded on the line
            row["error"] = "%s: %s" % (type(exc).__name__, exc)
        rows.append(row)
        gc.enable()
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def default_loop_call(args, device):
    """BASELINE config 4's terms under the reference's DEFAULT loop condition (slavcheva_optimizer2d.py:74-102,360-362:
    min_iterations 1, max_iterations 100, lower threshold 0.1 -- what every reference caller constructs): the call ends
    when the longest update falls to 0.1 voxels.  The library enqueues check_interval (32) gated launches at a time and
    reads the records in between (lsf_state_run_finish with an lsf_run_loop): milliseconds per CALL and the iterations the
    call executed, next to the same call enqueued launch by launch from Python."""
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.hostloop import parked_collector
    from levelsetfusion_python_amd.synthetic import sphere_pair
    from levelsetfusion_python_amd.synthetic import depth_pair
    n = args.size
    out = {}
    # two inputs: the bench's sphere pair (at 256^3 its first update is already shorter than 0.1 voxels: the default loop ends
    # after ONE iteration) and SURVEY 8(d)'s depth-frame pair (updates of several voxels: the loop runs on)
    for tag, (canonical, live0) in (("", sphere_pair(n, 3, device)), ("_depth_pair", depth_pair(n, device))):
        for key, library_run in (("ms_per_call", True), ("ms_per_call_launch_by_launch", False)):
            opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                           smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                           engine_options=dict(library_run=library_run))
            live = torch.empty_like(live0)
            with parked_collector():
                for k in range(args.warmup + args.steps):
                    if k == args.warmup:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    live.copy_(live0)
                    opt.optimize(live, canonical)
                torch.cuda.synchronize()
                out[key + tag] = (time.perf_counter() - t0) / args.steps * 1e3
            executed = out.setdefault("iterations_executed" + tag, len(opt.log.max_warps))
            assert executed == len(opt.log.max_warps)
    return dict(config="3D %d^3 KillingFusion under the reference's default loop condition (min_iterations 1, max_iterations "
                       "100, lower threshold 0.1): sphere-pair TSDF, and (_depth_pair) the TSDF pair of two synthetic depth "
                       "frames" % n, ms_per_step=out["ms_per_call"],
                value=n ** 3 * out["iterations_executed"] / (out["ms_per_call"] * 1e-3), **out)


def pairs_in_flight(args, device, lanes=2):
    """The caller loop of the reference's multi-pair scripts (run_hierarchical_optimizer3d_multipair.py:403-432: independent
    pairs, one optimize() each) with TWO pairs in flight: two optimizers, a host thread and a HIP stream each -- what
    experiment/multipair.run_pairs does with a sequence of optimizers.  A whole call is two foreign calls that run without
    the interpreter lock (lsf_state_run_begin / _finish), so one pair's launches fill the other's ramp, drain and host gaps.
    BASELINE config 4 per pair; reports milliseconds per PAIR (wall time / pairs) next to the one-pair-at-a-time figure."""
    import threading
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd.hostloop import parked_collector
    from levelsetfusion_python_amd.synthetic import sphere_pair
    n, iters, steps = args.size, args.iterations, args.steps
    canonical, live0 = sphere_pair(n, 3, device)

    def optimizer():
        return lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                        smoothing_term_method=lsf.SmoothingTermMethod.KILLING, gradient_descent_rate=0.1,
                                        data_term_weight=1.0, smoothing_term_weight=0.2, isomorphic_enforcement_factor=0.1,
                                        level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.0,
                                        max_iterations=iters, min_iterations=iters, check_interval=iters)

    def run(p):
        streams = [torch.cuda.Stream(device=device) for _ in range(p)]
        barrier = threading.Barrier(p + 1)
        sums = [None] * p

        def lane(k):
            torch.cuda.set_device(device)
            with torch.cuda.stream(streams[k]):
                opt, live = optimizer(), torch.empty_like(live0)
                for _ in range(args.warmup):
                    live.copy_(live0)
                    opt.optimize(live, canonical)
                streams[k].synchronize()
                barrier.wait()
                for _ in range(steps):
                    live.copy_(live0)
                    opt.optimize(live, canonical)
                streams[k].synchronize()
                sums[k] = float(live.double().sum().item())
        threads = [threading.Thread(target=lane, args=(k,)) for k in range(p)]
        for t in threads:
            t.start()
        with parked_collector():
            barrier.wait()
            t0 = time.perf_counter()
            for t in threads:
                t.join()
            dt = time.perf_counter() - t0
        return dt / (p * steps) * 1e3, sums
    one, s1 = run(1)
    two, s2 = run(lanes)
    return dict(config="3D %d^3 KillingFusion, %d fixed iterations per pair, %d independent pairs in flight (a host thread and "
                       "a HIP stream each)" % (n, iters, lanes),
                ms_per_pair=two, ms_per_pair_one_in_flight=one, ms_per_step=two, value=n ** 3 * iters / (two * 1e-3),
                results_equal=len(set(s1 + s2)) == 1, pairs_in_flight=lanes)


def killing_workload(args, device, world, rank, local_rank, dist, dense_walk=True):
    """BASELINE config 4 (module docstring); returns the JSON line as a dict"""
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd import _lib, device as dev
    from levelsetfusion_python_amd.slab import SlabComm, SlabLayout
    from levelsetfusion_python_amd.synthetic import sphere_pair

    n, iters = args.size, args.iterations
    strong = world > 1 and args.scaling == "strong"
    if strong:
        # BASELINE config 4 as written: ONE n^3 pair, z-slabbed over the ranks.  The exchange-group depth follows the slab
        # height: iteration j of a group recomputes h - 1 - j neighbour slices per interior side, (h - 1) / slices-per-rank
        # of a slab on average (h = 8 on 32-slice slabs would recompute 22 %; slices / 8 keeps it below 10 %)
        if n % world:
            raise SystemExit("--scaling strong: --size %d is not divisible by %d ranks" % (n, world))
        halo = args.halo if args.halo is not None else max(1, min(8, (n // world) // 8))
        layout = SlabLayout(n, rank, world, halo)
        pattern, z_shift = "one centred sphere pair, cut into slabs", 0
    else:
        halo = (8 if args.halo is None else args.halo) if world > 1 else 0
        layout = SlabLayout(n * world, rank, world, halo)
        faces = world > 1 and (args.pattern or "faces") == "faces"
        pattern = "spheres centred on the slab faces" if faces else "one sphere centred in every slab"
        z_shift = n // 2 if faces else 0
    if args.data == "depth" and world > 1:
        # north_star's "synthetic depth->TSDF volumes ... at 1, 2, 4 and 8 GPUs": ONE n^3 pair generated from two synthetic
        # depth frames, cut into slabs along Y -- the camera looks along +z (tsdf/generation.py:356-437), so the narrow band
        # is a sheet across z that z-slabs would leave to one or two ranks; y-slabs cut it into equal strips
        if n % world:
            raise SystemExit("--data depth: --size %d is not divisible by %d ranks" % (n, world))
        strong = True
        halo = args.halo if args.halo is not None else max(1, min(8, (n // world) // 8))
        layout = SlabLayout(n, rank, world, halo, axis=1)
        pattern = "one depth-frame pair, cut into slabs along y"
    args.halo = halo
    sl = layout.local_slice()
    if args.data == "depth":
        from levelsetfusion_python_amd.synthetic import depth_pair
        canonical, live0 = depth_pair(n, device)
        if world > 1:
            canonical, live0 = layout.cut(canonical), layout.cut(live0)
    else:
        canonical, live0 = sphere_pair(n, 3, device, (sl.start, sl.stop), z_shift)

    def make_optimizer():
        comm = SlabComm(layout) if world > 1 else None
        return comm, lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                              level_set_term_enabled=True,
                                              smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                              gradient_descent_rate=0.1, data_term_weight=1.0,
                                              smoothing_term_weight=0.2, isomorphic_enforcement_factor=0.1,
                                              level_set_term_weight=0.2, maximum_warp_length_lower_threshold=0.0,
                                              max_iterations=iters, min_iterations=iters, check_interval=iters,
                                              comm=comm)

    comm, opt = make_optimizer()
    live = torch.empty_like(live0)
    if world > 1 and comm.native() is not None:
        # One call through the library's RCCL transport before anything is timed.  Whether the library transport is
        # used at all was decided collectively while it was set up (SlabComm.native(): every rank falls back to
        # torch.distributed together if any rank cannot bind RCCL).  A failure AFTER that is fatal: a rank whose call
        # raises leaves its neighbours inside a device-side ncclRecv, so there is no collective left to agree on a
        # fallback with -- the exception ends this process with a non-zero status and the launcher tears the job down.
        live.copy_(live0)
        opt.optimize(live, canonical)
        torch.cuda.synchronize()

    def step():
        live.copy_(live0)  # a fresh pair every step (optimize() warps live in place)
        opt.optimize(live, canonical)
        return len(opt.log.max_warps)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if args.backend == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    executed = 0
    late = min(3, args.warmup)  # warm-up steps that run behind the collection below
    for _ in range(args.warmup - late):
        executed = step()
    # Python's cyclic collector walks every object torch has created (~40 ms per full collection, measured with
    # tools/step_times.py: one 2.5 ms step in ~25 took 40 ms): park the existing objects in the permanent generation,
    # as a serving loop would, and keep the cyclic collector out of the timed steps altogether (with few warm-up steps the
    # objects of the first timed steps would otherwise be walked in a full pass: one 60 ms pause seen in a 5-step
    # 512^3 run); reference counting still frees every tensor of a step as the step ends.
    # The collection takes ~40 ms during which the card sits idle and drops its clocks: timed straight behind it, the
    # first steps ran slow (20 timed steps: 1.78-1.80 ms each, 200: 1.72-1.73, 400: 1.71 -- a one-off of ~1.4 ms whatever
    # K), so the last warm-up steps run between the collection and the timed region.  Still W untimed steps in all.
    gc.collect()
    gc.freeze()
    gc.disable()
    for _ in range(late):
        executed = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        executed = step()
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        elapsed = max_over_ranks(elapsed)
    assert executed == iters, "expected %d fixed iterations, the gate closed after %d" % (iters, executed)
    # The same K steps under the conditions of the PRODUCT's pair loop (experiment/multipair.run_pairs ->
    # hostloop.parked_collector(): torch's objects parked, the cyclic collector left ON; caller shape
    # run_hierarchical_optimizer3d_multipair.py:403-432) -- reported next to ms_per_step as ms_per_step_gc_on
    from levelsetfusion_python_amd.hostloop import parked_collector
    gc.unfreeze()
    with parked_collector():
        for _ in range(late):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed_gc_on = time.perf_counter() - t0
    if world > 1:
        elapsed_gc_on = max_over_ranks(elapsed_gc_on)
    voxels_per_rank = n ** 3 // world if strong else n ** 3
    updates = voxels_per_rank * world * iters * args.steps
    value = updates / elapsed

    # ---- what the halo exchange carries (N > 1): the band voxels of the h boundary slices of every interior face, and
    # the bytes one exchange moves per rank and direction (compact faces: 16 B per band voxel; whole faces otherwise)
    halo_info = None
    if world > 1:
        L, h = layout, layout.halo
        band = ~((live0.abs() == 1.0) & (canonical.abs() == 1.0))
        mine = []
        if L.rank > 0:
            mine.append(int(band.narrow(L.axis, L.begin, h).sum().item()))
        if L.rank < world - 1:
            mine.append(int(band.narrow(L.axis, L.end - h, h).sum().item()))
        fast = getattr(opt.engine, "_fast", None)
        # (the library-enqueued slab call reports what travelled: lsf_state_run_result::compact_faces)
        compact = fast is not None and (getattr(fast, "compact_faces", None) == 1 or (
            getattr(fast, "native", None) is not None and getattr(fast, "faces_ref", None) is not None))
        face_voxels = h * live0.numel() // live0.shape[L.axis]
        face_bytes = [16 * v for v in mine] if compact else [16 * face_voxels] * len(mine)
        interval = getattr(fast, "exchange_interval", 1) if fast is not None else 1
        rows = [None] * world
        dist.all_gather_object(rows, dict(band_voxels=mine, bytes=face_bytes,
                                          band=int(layout.owned_of(band).sum().item())))
        per_face = [v for r in rows for v in r["band_voxels"]]
        halo_info = dict(halo_slices=h, iterations_per_exchange=interval,
                         exchanges_per_step=len([i for i in range(iters) if i % interval == interval - 1 and i + 1 < iters]),
                         faces="compact (band voxels only)" if compact else "whole slices",
                         band_voxels_per_face=per_face, face_voxels=face_voxels,
                         band_voxels_per_rank=[r["band"] for r in rows],
                         bytes_sent_per_exchange_per_rank=[sum(r["bytes"]) for r in rows])

    # ---- roofline of the dominant kernel: the fused warp-update kernel alone, HIP events on its stream, over exactly
    # the launch sequence of one step (band lists of the initial pair, `iters` ping-pong launches on the float4 state)
    eng = opt.engine
    grid = eng._grid(live0)
    rec = dev.new_records(2, device)

    boxes = None  # (tensor, count) when the timed steps walked the INTERIOR band voxels box by box (engine.last_call.box_walk)

    def launches(bands):
        states = dev.state_pack(live0, None, grid, copies=2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(iters):
            for band in bands:  # interior + (usually empty, then absent) boundary band voxels
                if boxes is not None and band is not None and band.subset == _lib.BAND_INTERIOR:
                    dev.slavcheva_state_iteration_boxes(states[i % 2], boxes[2], states[(i + 1) % 2], grid, eng.params,
                                                        None, rec, 0, boxes[0], boxes[1])
                else:
                    dev.slavcheva_state_iteration(states[i % 2], canonical, states[(i + 1) % 2], grid, eng.params, None,
                                                  rec, 0, band)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def roofline_of(bands, units, kernel_name, traffic):
        launches(bands)
        # the median of five measurements of the step's launch sequence: one 1.4 ms sample can catch a clock ramp or
        # a neighbour's burst (seen once: 34.3 us on a box whose other runs gave 28.2)
        kernel_ms = sorted(launches(bands) for _ in range(5))[2]
        alg_bytes = B_ALG["killing"] * units
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                   traffic=traffic, kernel=kernel_name, kernel_ms=kernel_ms, units_per_launch=units,
                   algorithmic_bytes_per_launch=alg_bytes, voxels_per_launch=voxels_per_rank)
        return out

    build_id = _lib.lib.lsf_build_id().decode()
    traffic_source = dict(loaded_build_id=build_id)

    def committed_traffic(key):
        """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md) -- only for the default
        size AND only while the loaded library is the build the counters were collected on (lsf_build_id()): a kernel
        change without a re-profile reports null, not stale bytes"""
        try:  # one file per profiled size: traffic.json (256^3, the default workload), traffic_512.json
            with open(os.path.join(ROOT, "profiles", "traffic.json" if n == 256 else "traffic_%d.json" % n)) as f:
                t = json.load(f)
        except (OSError, ValueError):
            return None
        traffic_source.update(tag=t.get("tag"), profiled_build_id=t.get("build_id"), loaded_build_id=build_id)
        if t.get("size") == n and t.get("workload") == "killing" and args.data == "sphere" and \
                t.get("build_id") == build_id:
            return t.get(key)
        return None

    # Units one launch processes = the voxels it visits: the band list (every other voxel is provably unchanged and
    # is not touched).  52 B per visited voxel-update is SURVEY 8(d)'s figure.  Next to it: the
    # SAME kernel walking every voxel (use_band_list=False: streams the whole state every iteration) -- the data point
    # that is bound by HBM rather than by the per-band-voxel arithmetic.
    name = "slavcheva_state_kernel<3,KILLING,LEVELSET,BASIC,DIRECT,%s>"
    if eng.use_band_list:
        if world > 1 and layout.axis == 1:
            # slabs cut along y: the lists of the whole local array, reduced to the owned rows (a row range is not a
            # contiguous run of a sorted list)
            bands = []
            for b in dev.band_lists(live0, canonical, dev.make_grid(live0.shape)):
                idx = b.indices[:b.count]
                rows = (idx // live0.shape[2]) % live0.shape[1]
                own = idx[(rows >= layout.begin) & (rows < layout.end)].contiguous()
                if own.numel():
                    bands.append(dev.BandList(own, own.numel(), b.subset))
        else:
            bands = dev.band_lists(live0, canonical, grid)
        # what the timed steps launched: one launch per iteration and list -- over the INTERIOR voxels' BOXES when the
        # engine walked those (its choice by band size: a 512^3 sphere pair)
        walk = "LIST"
        if world == 1 and eng.last_call.box_walk:
            boxes = dev.band_boxes(dev.StatePrepare(live0, canonical, grid))
            boxes = boxes + (dev.band_boxes_canonical(canonical, grid, *boxes),)
            walk = "BOXES of 4x4x4 through LDS"
            name = "slavcheva_state_box_kernel<KILLING,LEVELSET,BASIC,DIRECT> (%s)"
        roofline = roofline_of(bands, sum(b.count for b in bands), name % walk, committed_traffic("hbm_bytes_per_launch"))
        name = "slavcheva_state_kernel<3,KILLING,LEVELSET,BASIC,DIRECT,%s>"
        if boxes is not None:
            roofline["boxes_per_launch"] = boxes[1]
            boxes = None  # (the dense walk below is the list kernel's)
        roofline["dense_equivalent_gbs"] = B_ALG["killing"] * voxels_per_rank / (roofline["kernel_ms"] * 1e-3) / 1e9
    else:
        roofline = roofline_of([None], voxels_per_rank, name % "DENSE", committed_traffic("dense_hbm_bytes_per_launch"))
    roofline_dense = roofline_of([None], voxels_per_rank, name % "DENSE",
                                 committed_traffic("dense_hbm_bytes_per_launch")) \
        if eng.use_band_list and dense_walk and not (world > 1 and layout.axis == 1) else None  # y-cut grids: lists only

    out = dict(metric="voxel-warp-updates/sec", value=value, unit="voxel-warp-updates/s", n_gpus=world,
               steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
               ms_per_step_gc_on=elapsed_gc_on / args.steps * 1e3, higher_is_better=True,
               scaling="strong" if strong else "weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload="3D %d^3 KillingFusion (Killing + level-set) SlavchevaOptimizer3d, %d fixed "
                                    "iterations per step, %s" % (n, iters, "sphere-pair TSDF" if args.data == "sphere"
                                                                 else "TSDF pair from two synthetic depth frames"),
                           voxels_per_gpu=voxels_per_rank, iterations_per_step=iters,
                           parallelism=("%s-slab x%d (%s: %s), halo %d, %s" % (
                               "zy"[layout.axis], world, "strong scaling" if strong else "weak scaling", pattern, args.halo,
                               "RCCL send/recv from the library (lsf_slab_state_iteration)"
                               if comm.native() is not None else "torch.distributed " + args.backend))
                           if world > 1 else "single GPU"),
               roofline=roofline)
    if world > 1:
        # the transport as RCCL itself reports it (ncclCommUserRank / ncclCommCount of the library's communicator) and how
        # the timed calls were enqueued
        info = comm.native_info()
        fast = getattr(eng, "_fast", None)
        out["rccl"] = dict(transport="native RCCL (lsf_slab_run_begin / _finish: the whole call enqueued by the library)"
                           if eng.last_call.library_run else
                           ("native RCCL (lsf_slab_state_iteration, one call per iteration)" if info is not None
                            else "torch.distributed " + args.backend),
                           nranks=info[1] if info is not None else None, rank=info[0] if info is not None else None,
                           library_run=bool(eng.last_call.library_run),
                           compact_faces=getattr(fast, "compact_faces", None))
    if halo_info is not None:
        out["halo_exchange"] = halo_info
        # every interior face of the run carries at least this many band voxels / the busiest rank sends this much
        out["halo_band_voxels_per_face"] = min(halo_info["band_voxels_per_face"])
        out["halo_bytes_per_exchange"] = max(halo_info["bytes_sent_per_exchange_per_rank"])
    roofline["traffic_source"] = traffic_source
    # rocprofv3's own per-dispatch average of the same kernel (profiles/<tag>_bench_kernel_stats.csv, committed with the
    # traffic figure and valid for the same build only): per-dispatch timing reads ~6 % above back-to-back HIP events
    avg_us = committed_traffic("list_kernel_avg_us" if eng.use_band_list else "dense_kernel_avg_us")
    if avg_us:
        roofline["rocprofv3"] = dict(kernel_us=avg_us, frac=roofline["algorithmic_bytes_per_launch"] / (avg_us * 1e-6) / 1e9
                                     / HBM_PEAK_GBS, source=traffic_source.get("tag"))
    # the rate over the voxels the launches actually visit (band lists): comparable across inputs and rounds, where
    # `value` (field voxels, as the reference counts its work) grows with the share of the volume outside the band
    out["visited_voxel_updates_per_s"] = roofline["units_per_launch"] * world * iters * args.steps / elapsed
    if roofline_dense is not None:
        out["roofline_dense_walk"] = roofline_dense
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_size, args.cpu_sample_iterations)
    if world > 1:
        comm.close()  # the library's own RCCL communicator, before torch tears its process group down
    return out


if __name__ == "__main__":
    main()
