"""Builds liblsf_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.  No torch extension machinery:
the library has a plain C ABI (include/lsf_hip.h) and is loaded with ctypes."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liblsf_hip.so")
SOURCES = ["lsf_fields.hip", "lsf_hierarchical.hip", "lsf_slavcheva.hip", "lsf_slavcheva_state.hip",
           "lsf_slavcheva_chain.hip", "lsf_sobolev_state.hip", "lsf_slab.hip", "lsf_tsdf.hip"]
HEADERS = ["lsf_device.h", "lsf_slavcheva_terms.h", "lsf_slavcheva_state_taps.h",
           os.path.join("..", "..", "include", "lsf_hip.h")]
# -ffp-contract=off: multiply and add stay separately rounded so that results are bit-identical to the numpy
# oracle (numpy never fuses); the path is HBM/L1-bound, the lost FMAs do not show.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17",
               "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def source_id():
    """first 16 hex digits of the SHA-256 over the library's sources (names and contents, sorted): lsf_build_id()"""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(sources() + [os.path.join(CSRC, x) for x in HEADERS]):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


ID_PATH = os.path.join(LIB_DIR, "build_id.txt")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    try:  # a snapshot copy may have fresh mtimes: the recorded id of the sources the library was built from decides
        with open(ID_PATH) as f:
            return f.read().strip() != source_id()
    except OSError:
        pass
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """compile every .hip source for gfx950 and link liblsf_hip.so; returns the library path"""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = find_hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    build_id = source_id()
    jobs = []
    for src in sources():  # one hipcc per source, all at once (7 files on 8 cores: ~25 s instead of ~60 s)
        obj = os.path.join(LIB_DIR, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + ['-DLSF_BUILD_ID="%s"' % build_id, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [cmd for cmd, job in jobs if job.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(ID_PATH, "w") as f:
        f.write(build_id + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True))
