"""A/B of builds of liblsf_hip.so on the fused Slavcheva kernel (band lists, sphere pair): per build the time per
launch at the given sizes, a hash of the state after `iters` iterations from a fresh pair (bit-equality across builds)
and the last record.  usage: ab_state_kernel.py [--sizes 256,512] [--iters 50] name=path.so [name=path.so ...]
("name=" alone = the in-tree library).  Every build runs in its own process (LSF_HIP_LIBRARY)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(sizes, iters):
    sys.path.insert(0, ROOT)
    import torch
    import levelsetfusion_python_amd as lsf
    from levelsetfusion_python_amd import _lib, device as dev
    from levelsetfusion_python_amd.synthetic import sphere_pair
    out = {}
    for n in sizes:
        eng = lsf.SlavchevaOptimizer3d(field_size=n, compute_method=lsf.ComputeMethod.DIRECT,
                                       level_set_term_enabled=True,
                                       smoothing_term_method=lsf.SmoothingTermMethod.KILLING).engine
        if os.environ.get("ENERGY", "1") == "0":
            eng.params.energy_mode = _lib.ENERGY_NONE
        grid = dev.make_grid((n, n, n))
        c, l = sphere_pair(n, 3, "cuda")
        bands = dev.band_lists(l, c, grid)
        rec = dev.new_records(iters, "cuda")

        def run(records_index=lambda i: 0):
            st = dev.state_pack(l, None, grid, copies=2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(iters):
                for b in bands:
                    dev.slavcheva_state_iteration(st[i % 2], c, st[(i + 1) % 2], grid, eng.params, None, rec,
                                                  records_index(i), b)
            e1.record()
            torch.cuda.synchronize()
            return st[iters % 2], e0.elapsed_time(e1) / iters

        rec.zero_()
        final, _ = run(lambda i: i)
        h = hashlib.sha1(final.cpu().numpy().tobytes()).hexdigest()[:16]
        host = dev.decode_records(dev.records_to_host(rec))
        times = sorted(run()[1] for _ in range(5))
        out[str(n)] = dict(us=times[0] * 1e3, us_median=times[2] * 1e3, hash=h, units=sum(b.count for b in bands),
                           last_max=float(host["max_value"][-1]), last_energies=[float(host[k][-1]) for k in
                                                                           ("data_energy", "smoothing_energy",
                                                                            "level_set_energy")])
    print("RESULT " + json.dumps(out))


def main():
    args = sys.argv[1:]
    sizes, iters = [256], 50
    builds = []
    while args:
        a = args.pop(0)
        if a == "--child":
            return child([int(s) for s in args[0].split(",")], int(args[1]))
        if a == "--sizes":
            sizes = [int(s) for s in args.pop(0).split(",")]
        elif a == "--iters":
            iters = int(args.pop(0))
        else:
            name, _, path = a.partition("=")
            builds.append((name, path))
    results = {}
    for name, path in builds:
        env = dict(os.environ)
        if path:
            env["LSF_HIP_LIBRARY"] = os.path.join(ROOT, path) if not os.path.isabs(path) else path
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", ",".join(map(str, sizes)), str(iters)],
                           env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("%s FAILED\n%s\n%s" % (name, p.stdout[-2000:], p.stderr[-2000:]))
            continue
        results[name] = json.loads(line[0][7:])
    for n in sizes:
        base = None
        for name, _ in builds:
            r = results.get(name, {}).get(str(n))
            if not r:
                continue
            base = base or r
            frac = 52.0 * r["units"] / (r["us"] * 1e-6) / 8e12
            print("%4d^3 %-14s %8.2f us (median %8.2f)  frac(52 B) %.3f  state %s %s  max %.9g  energies %r" % (
                n, name, r["us"], r["us_median"], frac, r["hash"], "==" if r["hash"] == base["hash"] else "!=",
                r["last_max"], r["last_energies"]))


if __name__ == "__main__":
    main()
