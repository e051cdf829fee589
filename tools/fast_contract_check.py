"""What bit-exactness costs: the same sources built with -ffp-contract=fast (the compiler may fuse
a * b + c into one rounding), default configuration of the bench (KillingFusion, sphere pair), 50 iterations: speed of the
fused list kernel and the deviation from the default build's result (== the oracle, tests/test_gpu_bench_config.py).
usage: fast_contract_check.py  (runs itself twice, once per library)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = os.path.join(ROOT, "levelsetfusion-python_amd/lib/variants/fastcontract.so")


def child(sizes, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd import device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    res = {}
    for n in sizes:
        c, l0 = sphere_pair(n, 3, "cuda")
        opt = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT, level_set_term_enabled=True,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING,
                                       maximum_warp_length_lower_threshold=0.0, max_iterations=50, min_iterations=50,
                                       check_interval=50)
        l = l0.clone()
        opt.optimize(l, c)
        np.savez(out % n, live=l.cpu().numpy(), warp=opt.warp_field.cpu().numpy(), max_warps=np.float32(opt.log.max_warps))
        eng = opt.engine
        grid = dev.make_grid((n, n, n))
        bands = dev.band_lists(l0, c, grid)
        rec = dev.new_records(1, "cuda")
        best = None
        for _ in range(5):
            st = dev.state_pack(l0, None, grid, copies=2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(50):
                for b in bands:
                    dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec, 0, b)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / 50
            best = t if best is None else min(best, t)
        res[n] = dict(us=best, band=sum(b.count for b in bands))
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child([int(v) for v in sys.argv[2].split(",")], sys.argv[3])
        sys.exit(0)
    import numpy as np
    sizes = "64,128,256"
    out = {}
    for tag, lib in (("exact", None), ("fast", FAST)):
        env = dict(os.environ)
        if lib:
            env["LSF_HIP_LIBRARY"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", sizes, "/tmp/fc_%s_%%d.npz" % tag], env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for n in (64, 128, 256):
        a, b = np.load("/tmp/fc_exact_%d.npz" % n), np.load("/tmp/fc_fast_%d.npz" % n)
        dw = (a["warp"].astype(np.float64) - b["warp"]).ravel()
        dl = (a["live"].astype(np.float64) - b["live"]).ravel()
        e, f = out["exact"][str(n)], out["fast"][str(n)]
        print("%3d^3: exact %.2f us (%.3f of the roofline), contract=fast %.2f us (%.3f); after 50 iterations warp RMSE %.3e "
              "max %.3e, live RMSE %.3e max %.3e, max-warp trajectory max diff %.3e" % (
                  n, e["us"], 52.0 * e["band"] / (e["us"] * 1e-6) / 8e12, f["us"], 52.0 * f["band"] / (f["us"] * 1e-6) / 8e12,
                  np.sqrt((dw ** 2).mean()), np.abs(dw).max(), np.sqrt((dl ** 2).mean()), np.abs(dl).max(),
                  np.abs(a["max_warps"] - b["max_warps"]).max()))
