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
           "lsf_slavcheva_run.hip", "lsf_slavcheva_box.hip", "lsf_sobolev_state.hip", "lsf_sobolev_box.hip", "lsf_slab.hip", "lsf_tsdf.hip"]
HEADERS = ["lsf_device.h", "lsf_slavcheva_terms.h", "lsf_slavcheva_state_taps.h",
           os.path.join("..", "..", "include", "lsf_hip.h")]
ABI_HEADER = os.path.join(PKG_DIR, "..", "include", "lsf_hip.h")
# -ffp-contract=off: multiply and add stay separately rounded so that results are bit-identical to the numpy
# oracle (numpy never fuses); the path is HBM/L1-bound, the lost FMAs do not show.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17",
               "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def sources(names=None):
    return [os.path.join(CSRC, s) for s in (SOURCES if names is None else names)
            if os.path.exists(os.path.join(CSRC, s))]


def source_id(names=None, headers=None):
    """first 16 hex digits of the SHA-256 over the library's sources (names and contents, sorted): lsf_build_id()"""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(sources(names) + [os.path.join(CSRC, x) for x in (HEADERS if headers is None else headers)]):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def abi_hash(header=None):
    """first 16 hex digits of the SHA-256 over the NORMALISED text of include/lsf_hip.h: comments stripped, every run of
    white space collapsed to one blank -- what is left is the structs, the constants and the prototypes, i.e. the ABI.
    Compiled into the library (-DLSF_ABI_HASH, lsf_abi_hash()) and compared by _lib.py with the hash of the header the
    binding was written against: a struct that grew or a prototype that changed is refused at load time whether or
    not LSF_ABI_VERSION was bumped."""
    import hashlib
    import re
    with open(header or ABI_HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"\s+", " ", text).strip()
    return hashlib.sha256(text.encode()).hexdigest()[:16]


ID_PATH = os.path.join(LIB_DIR, "build_id.txt")


def is_stale(lib_path=LIB_PATH, id_path=ID_PATH, names=None, headers=None):
    if not os.path.exists(lib_path):
        return True
    try:  # a snapshot copy may have fresh mtimes: the recorded id of the sources the library was built from decides
        with open(id_path) as f:
            return f.read().strip() != source_id(names, headers)
    except OSError:
        pass
    t = os.path.getmtime(lib_path)
    deps = sources(names) + [os.path.join(CSRC, h) for h in (HEADERS if headers is None else headers)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_and_link(names, headers, lib_path, id_path, verbose, extra=()):
    hipcc = find_hipcc()
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    build_id = source_id(names, headers)
    jobs = []
    for src in sources(names):  # one hipcc per source, all at once (7 files on 8 cores: ~25 s instead of ~60 s)
        obj = os.path.join(LIB_DIR, os.path.basename(src).replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + ['-DLSF_BUILD_ID="%s"' % build_id, '-DLSF_ABI_HASH="%s"' % abi_hash()] + \
            list(extra) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [cmd for cmd, job in jobs if job.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(id_path, "w") as f:
        f.write(build_id + "\n")
    return lib_path


def build(force=False, verbose=True):
    """compile every .hip source of the product library for gfx950 and link liblsf_hip.so; returns the library path"""
    if not force and not is_stale():
        return LIB_PATH
    return _compile_and_link(None, None, LIB_PATH, ID_PATH, verbose)


if __name__ == "__main__":
    print(build(force=True))
