"""Multi-pair driver: N independent (canonical, live) TSDF pairs -> one optimizer -> per-level convergence reports
-> table + analysis files.  Mirrors the caller loop and report post-processing of the reference's
run_hierarchical_optimizer3d_multipair.py:85-211,330-331,403-441 (same .npz pair format {canonical, live}, same
DataFrame columns -- 2 + 17 per level --, same analysis.txt / bad_cases.csv / all_cases.csv), not its CLI.

Pairs are independent (no state is carried from pair to pair), so with torch.distributed initialised they are dealt
round-robin to the ranks ("replicas only") and the report sets are gathered on rank 0.
"""
import glob
import os
import re

import numpy as np
import pandas as pd

from ..hostloop import parked_collector

LEVEL_COLUMNS = ["iter_count", "iter_lim_reached", "warp_delta_amt_ratio", "warp_delta_min", "warp_delta_max",
                 "warp_delta_mean", "warp_delta_std", "warp_delta_max_x", "warp_delta_max_y",
                 "warps_below_min_thresh", "warps_above_max_thresh", "diff_delta_min", "diff_delta_max",
                 "diff_delta_mean", "diff_delta_std", "diff_max_x", "diff_max_y"]


def save_pair(folder, frame_number, pixel_row, canonical, live):
    """data_{frame}_{row}.npz with arrays `canonical`, `live` (…3d_multipair.py:330-331)"""
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "data_{:d}_{:d}.npz".format(int(frame_number), int(pixel_row)))
    np.savez(path, canonical=np.asarray(canonical, dtype=np.float32), live=np.asarray(live, dtype=np.float32))
    return path


def load_pairs(folder):
    """[(frame_number, pixel_row, canonical, live)] sorted by (frame, row) (…3d_multipair.py:346-355)"""
    out = []
    for path in glob.glob(os.path.join(folder, "data_*_*.npz")):
        m = re.match(r"data_(\d+)_(\d+)\.npz$", os.path.basename(path))
        if not m:
            continue
        with np.load(path) as archive:
            out.append((int(m.group(1)), int(m.group(2)), archive["canonical"], archive["live"]))
    out.sort(key=lambda t: (t[0], t[1]))
    return out


def run_pairs(optimizer, pairs, progress=None):
    """for (canonical, live) in pairs: optimizer.optimize(canonical, live); collect the per-level report sets.
    `optimizer` must have been built with LoggingParameters(collect_per_level_convergence_reports=True).
    Returns (report_sets, frame_numbers_and_rows) -- on rank 0 for ALL pairs when torch.distributed is up.

    `optimizer` may also be a SEQUENCE of P equally configured optimizers: P pairs are then in flight at once on this
    rank's GPU, each optimizer in a host thread and on a HIP stream of its own.  Pairs are independent, so the table is
    the same; what changes is the throughput -- one pair leaves the card idle while its launch-bound coarse levels run and
    at every launch boundary of the fine ones, and another pair's kernels move into those gaps (tools/pairs_in_flight.py
    at 256^3: the Tikhonov-only hierarchical optimizer 21.1 -> 18.0 ms per pair with two in flight; with the 7-tap kernel
    and for the Slavcheva optimizer the gain stays inside the run-to-run spread -- one host thread at a time holds the
    interpreter).

    The loop runs under `hostloop.parked_collector()`: a full pass of Python's cyclic collector over torch's objects
    costs as much as twenty KillingFusion calls at 256^3, and bench.py's timed steps run without it too."""
    with parked_collector():
        return _run_pairs(optimizer, pairs, progress)


def _run_pairs(optimizer, pairs, progress):
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lanes = list(optimizer) if isinstance(optimizer, (list, tuple)) else [optimizer]
    if not lanes:
        raise ValueError("no optimizer given")
    todo = [(k, item) for k, item in enumerate(pairs) if k % world == rank]

    def one(opt, k, item):
        frame, row, canonical, live = item
        opt.optimize(canonical, live)
        reports = opt.get_per_level_convergence_reports()
        if not reports:
            raise ValueError("the optimizer collects no per-level convergence reports: construct it with "
                             "logging_parameters=LoggingParameters(collect_per_level_convergence_reports=True)")
        return (k, (frame, row), reports)

    mine = []
    if len(lanes) == 1:
        for k, item in todo:
            mine.append(one(lanes[0], k, item))
            if progress is not None:
                progress(k, len(pairs))
    else:
        import threading
        import torch
        device = torch.cuda.current_device()
        lock = threading.Lock()
        failures = []
        # every optimizer runs its first pair alone: what it sets up once per field shape (HIP graphs of the launch-bound
        # levels) is captured before anything runs beside it -- HIP refuses ordinary calls of other threads while a
        # capture is in progress; later shapes without a graph run eagerly (same results)
        for opt, (k, item) in zip(lanes, todo):
            mine.append(one(opt, k, item))
            if progress is not None:
                progress(k, len(pairs))
        cursor = [min(len(lanes), len(todo))]
        engines = [getattr(opt, "_engine", None) for opt in lanes]
        for e in engines:
            if hasattr(e, "allow_graph_capture"):
                e.allow_graph_capture = False

        def lane(opt):
            try:
                torch.cuda.set_device(device)
                stream = torch.cuda.Stream(device=device)
                with torch.cuda.stream(stream):
                    while not failures:
                        with lock:
                            if cursor[0] >= len(todo):
                                break
                            k, item = todo[cursor[0]]
                            cursor[0] += 1
                        result = one(opt, k, item)
                        with lock:
                            mine.append(result)
                            if progress is not None:
                                progress(k, len(pairs))
                    stream.synchronize()
            except BaseException as exc:  # noqa: BLE001 -- handed to the caller's thread below
                failures.append(exc)

        threads = [threading.Thread(target=lane, args=(opt,)) for opt in lanes]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in engines:
            if hasattr(e, "allow_graph_capture"):
                e.allow_graph_capture = True
        if failures:
            raise failures[0]
        mine.sort(key=lambda t: t[0])
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank != 0:
            return [], []
        mine = sorted((item for part in gathered for item in part), key=lambda t: t[0])
    return [m[2] for m in mine], [m[1] for m in mine]


def post_process_convergence_report_sets(convergence_report_sets, frame_numbers_and_rows):
    """the reference's table: canonical_frame, pixel_row, then l{i}_<17 columns> per level (…multipair.py:85-133)"""
    data = {"canonical_frame": [f for f, _ in frame_numbers_and_rows],
            "pixel_row": [r for _, r in frame_numbers_and_rows]}
    n_levels = len(convergence_report_sets[0]) if convergence_report_sets else 0
    for i in range(n_levels):
        for col in LEVEL_COLUMNS:
            data["l%d_%s" % (i, col)] = []
    for report_set in convergence_report_sets:
        for i, report in enumerate(report_set):
            w, t = report.warp_delta_statistics, report.tsdf_difference_statistics
            values = [report.iteration_count, report.iteration_limit_reached, w.ratio_above_min_threshold,
                      w.length_min, w.length_max, w.length_mean, w.length_standard_deviation,
                      w.longest_warp_location.x, w.longest_warp_location.y, w.is_largest_below_min_threshold,
                      w.is_largest_above_max_threshold, t.difference_min, t.difference_max, t.difference_mean,
                      t.difference_standard_deviation, t.biggest_difference_location.x,
                      t.biggest_difference_location.y]
            for col, v in zip(LEVEL_COLUMNS, values):
                data["l%d_%s" % (i, col)].append(v)
    return pd.DataFrame.from_dict(data)


def infer_level_count(data_frame):
    return (len(data_frame.columns) - 2) // len(LEVEL_COLUMNS)


def get_converged_ratio_for_level(data_frame, i_level):
    reached = data_frame["l{:d}_iter_lim_reached".format(i_level)]
    return float((~reached.astype(bool)).sum()) / len(data_frame) if len(data_frame) else 0.0


def get_mean_iteration_count_for_level(data_frame, i_level):
    return data_frame["l{:d}_iter_count".format(i_level)].mean()


def analyze_convergence_data(data_frame, out_path):
    """analysis.txt: per-level convergence ratios and mean iteration counts (…multipair.py:166-191)"""
    os.makedirs(out_path, exist_ok=True)
    n = infer_level_count(data_frame)
    lines = ["Per-level convergence ratios:",
             "".join("  level {:d}: {:.2%}".format(i, get_converged_ratio_for_level(data_frame, i)) for i in range(n)),
             "Per-level mean iteration counts:",
             "".join("  level {:d}: {:.2f}".format(i, get_mean_iteration_count_for_level(data_frame, i))
                     for i in range(n))]
    text = "\n".join(lines) + "\n"
    with open(os.path.join(out_path, "analysis.txt"), "w") as f:
        f.write(text)
    return text


def save_bad_cases(data_frame, out_path):
    n = infer_level_count(data_frame)
    cols = ["canonical_frame", "pixel_row", "l{:d}_warp_delta_max_x".format(n - 1), "l{:d}_warp_delta_max_y".format(n - 1)]
    bad = data_frame[cols][data_frame["l{:d}_iter_lim_reached".format(n - 1)].astype(bool)]
    bad.to_csv(os.path.join(out_path, "bad_cases.csv"), header=False, index=False)
    return bad


def save_all_cases(data_frame, out_path):
    n = infer_level_count(data_frame)
    cols = ["canonical_frame", "pixel_row", "l{:d}_warp_delta_max_x".format(n - 1), "l{:d}_warp_delta_max_y".format(n - 1)]
    data_frame[cols].to_csv(os.path.join(out_path, "all_cases.csv"), header=False, index=False)


def run_experiment(optimizer, pairs, out_path):
    """the whole driver: optimize every pair, write convergence_reports.csv/.pkl, analysis.txt, bad/all cases"""
    report_sets, ids = run_pairs(optimizer, pairs)
    if not report_sets:  # non-zero ranks of a distributed run
        return None
    os.makedirs(out_path, exist_ok=True)
    df = post_process_convergence_report_sets(report_sets, ids)
    df.to_csv(os.path.join(out_path, "convergence_reports.csv"), index=False)
    df.to_pickle(os.path.join(out_path, "convergence_reports.pkl"))
    analyze_convergence_data(df, out_path)
    save_bad_cases(df, out_path)
    save_all_cases(df, out_path)
    return df
