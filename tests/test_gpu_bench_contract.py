"""bench.py's one-line JSON contract on a small volume (GPU): the keys the driver reads, the roofline and cpu_baseline
objects, the --data / --workload variants."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout  # exactly ONE line on stdout
    return json.loads(lines[0])


def test_default_workload_line():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--size", "64", "--iterations", "4", "--cpu-sample-size", "16", "--cpu-sample-iterations", "2",
             "--secondary-divisor", "4")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "voxel-warp-updates/sec" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and abs(d["value"] - 64 ** 3 * 4 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_ms"] > 0
    assert abs(r["achieved"] - 52 * r["units_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    # the rate over the voxels the launches visit, next to the field-voxel rate; committed PMC traffic only for the size
    # and the build it was measured on (64^3 is neither): null, with the provenance it was checked against
    assert 0 < d["visited_voxel_updates_per_s"] <= d["value"]
    assert abs(d["visited_voxel_updates_per_s"] - r["units_per_launch"] * 4 * 2 / (d["ms_per_step"] * 2e-3)) \
        < 1e-6 * d["value"]
    assert r["traffic"] is None and r["traffic_source"]["loaded_build_id"]
    # the other configurations of BASELINE.json ride on the same line (here at 1/4 of their edge lengths)
    sec = {row["workload"]: row for row in d["secondary"]}
    assert sorted(sec) == ["config3", "hier-full", "hier-tik", "hier2d", "killing", "killing-default", "killing-pairs",
                           "multiframe", "sobolev"]
    default = sec.pop("killing-default")  # the reference's default loop condition: ms per call, iterations executed
    assert "error" not in default, default
    assert 1 <= default["iterations_executed"] <= 100 and default["ms_per_call"] > 0
    assert default["ms_per_call_launch_by_launch"] > 0 and "default loop condition" in default["config"]
    pairs = sec.pop("killing-pairs")  # two independent pairs in flight: milliseconds per pair, same results as one by one
    assert "error" not in pairs, pairs
    assert pairs["pairs_in_flight"] == 2 and pairs["results_equal"] and pairs["ms_per_pair"] > 0
    assert pairs["ms_per_pair_one_in_flight"] > 0
    for name, row in sec.items():
        assert "error" not in row, row
        assert row["ms_per_step"] > 0 and row["visited_voxel_updates_per_s"] > 0 and row["frac"] > 0 and row["config"]
    # the hierarchical rows name their dominant kernels and time them alone (HIP events): one finest-level iteration
    assert sec["config3"]["size"] == 64 and "7-tap" in sec["config3"]["config"]  # BASELINE config 3 (here 64^3)
    for name in ("hier-tik", "hier-full", "multiframe"):
        row = sec[name]
        assert row["kernel"] and "hier_iteration_kernel" in row["kernel"] and row["kernel_ms"] > 0
        assert row["finest_level_frac"] > 0 and all(v > 0 for v in row["kernels_ms"].values())
    assert "convolve_xyz_kernel" in sec["hier-full"]["kernel"] and "convolve_xyz_kernel" not in sec["hier-tik"]["kernel"]
    assert sec["killing"]["size"] == 128 and sec["killing"]["kernel_ms"] > 0
    assert sec["hier2d"]["size"] == 128 and sec["hier2d"]["us_per_iteration"] > 0 and "LAUNCH-BOUND" in sec["hier2d"]["note"]
    assert abs(sec["hier2d"]["us_per_iteration"] - sec["hier2d"]["ms_per_step"] * 1e3 / 300) < 1e-6
    for key in ("us_per_iteration_default_threshold", "us_per_iteration_default_constructor",
                "us_per_iteration_default_constructor_graph_path"):  # the stop test armed; + the gradient kernel
        assert sec["hier2d"][key] > 0, key


def test_hier2d_workload_line():
    """BASELINE config 2 as a bench mode: 2-D, 3 levels, microseconds per iteration next to the rate"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--workload", "hier2d", "--size", "128", "--iterations", "10", "--no-cpu-baseline")
    assert "2D 128^2 HierarchicalOptimizer2d" in d["config"]["workload"] and "secondary" not in d
    per_pair = 10 * (128 ** 2 + 64 ** 2 + 32 ** 2)
    assert abs(d["value"] - per_pair * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert abs(d["us_per_iteration"] - d["ms_per_step"] * 1e3 / 30) < 1e-6


def test_depth_data_and_other_workloads():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--size", "64", "--iterations", "4", "--data", "depth", "--no-cpu-baseline")
    assert "depth" in d["config"]["workload"] and d["roofline"]["traffic"] is None and "cpu_baseline" not in d
    for workload in ("hier-tik", "hier-full", "sobolev"):
        d = _run("--size", "64", "--iterations", "3", "--workload", workload, "--no-cpu-baseline")
        assert d["value"] > 0 and d["n_gpus"] == 1 and d["config"]["workload"]
        r = d["roofline"]
        b_alg = {"hier-tik": 68, "hier-full": 104, "sobolev": 76}[workload]
        # the fraction is over the voxels the launches VISIT; a dense-equivalent figure is labelled as such
        assert abs(r["achieved"] - d["visited_voxel_updates_per_s"] * b_alg / 1e9) < 1e-6 * r["achieved"]
        if workload == "sobolev":
            assert d["visited_voxel_updates_per_s"] < d["value"] and r["dense_equivalent_gbs"] > r["achieved"]
        else:
            assert d["visited_voxel_updates_per_s"] == d["value"] and "dense_equivalent_gbs" not in r


def test_multiframe_workload_line():
    """BASELINE config 5 as a bench mode (small: 4 frames of 64^3, 2 iterations per level)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run("--workload", "multiframe", "--size", "64", "--frames", "4", "--iterations", "2", "--no-cpu-baseline")
    per_pair = 2 * sum((64 >> k) ** 3 for k in range(4))
    assert "multi-frame" in d["config"]["workload"] and "4 frames" in d["config"]["workload"]
    assert abs(d["value"] - 3 * per_pair * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0


def _run_ranks(world, *flags):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank) -- here with gloo and
    all ranks on the one GPU of the test box (the only difference to the 8-GPU node is the transport)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", str(world), "--steps", "2", "--warmup", "1", "--backend", "gloo", "--share-device",
                          "--no-cpu-baseline", *flags], capture_output=True, text=True, timeout=900, cwd=ROOT)
    # the ranks' own tracebacks, not the launcher's summary of them
    assert out.returncode == 0, "\n".join(ln for ln in out.stderr.splitlines() if ln.startswith("[rank"))[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 alone prints the JSON line
    return json.loads(lines[0])


def test_two_rank_lines():
    """N = 2: the z-slabbed default workload, and config 5 both ways (replicas: pairs dealt to ranks; slab: every pair cut)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = _run_ranks(2, "--size", "64", "--iterations", "6", "--halo", "2")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "z-slab x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 64 ** 3 * 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    # the DEFAULT N > 1 input puts narrow bands across every slab face: the halo exchange carries data
    h = d["halo_exchange"]
    assert "faces" in d["config"]["parallelism"] and h["halo_slices"] == 2 and h["iterations_per_exchange"] == 2
    assert h["exchanges_per_step"] == 2 and len(h["band_voxels_per_face"]) == 2      # after iterations 1 and 3 (not the last)
    assert d["halo_band_voxels_per_face"] > 0.05 * h["face_voxels"] and d["halo_bytes_per_exchange"] > 0
    # ... where round 3's input (one sphere inside every slab) had nothing to send
    d = _run_ranks(2, "--size", "96", "--iterations", "4", "--halo", "2", "--pattern", "centered")
    assert d["halo_band_voxels_per_face"] == 0
    # strong scaling: BASELINE config 4 as written, ONE pair cut over the ranks; the exchange-group depth follows the slab
    d = _run_ranks(2, "--size", "64", "--iterations", "6", "--scaling", "strong")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "strong scaling" in d["config"]["parallelism"]
    assert d["config"]["voxels_per_gpu"] == 64 ** 3 // 2 and d["halo_exchange"]["halo_slices"] == 4
    assert abs(d["value"] - 64 ** 3 * 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert d["halo_band_voxels_per_face"] > 0 and d["halo_bytes_per_exchange"] > 0
    # depth-derived volumes on N > 1: ONE pair from two synthetic depth frames, slabs cut along y (the band is a sheet
    # across z); band voxels per rank balanced
    d = _run_ranks(2, "--size", "64", "--iterations", "6", "--data", "depth")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "y-slab x2" in d["config"]["parallelism"]
    assert "depth" in d["config"]["workload"] and d["halo_band_voxels_per_face"] > 0
    per_rank = d["halo_exchange"]["band_voxels_per_rank"]
    assert len(per_rank) == 2 and abs(per_rank[0] - per_rank[1]) <= 0.2 * (per_rank[0] + per_rank[1]) / 2
    assert abs(d["value"] - 64 ** 3 * 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    per_pair = 3 * sum((64 >> k) ** 3 for k in range(4))
    d = _run_ranks(2, "--workload", "multiframe", "--size", "64", "--frames", "3", "--iterations", "3")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "replicas x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 2 * per_pair * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    d = _run_ranks(2, "--workload", "multiframe", "--parallelism", "slab", "--halo", "4", "--size", "64", "--frames", "3",
                   "--iterations", "3")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "z-slab x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * per_pair * 2 / (d["ms_per_step"] * 2e-3)) < 2e-6 * d["value"]


def test_plain_form_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher in front (the shape of the driver's N = 1 command): bench.py starts
    torch.distributed.run itself as a child process and relays rank 0's one JSON line and the exit status"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                           "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo", "--share-device", "--no-cpu-baseline", "--size", "64", "--iterations", "6",
                          "--halo", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "z-slab x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 64 ** 3 * 6 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
