"""Builds libvitmi_hip.so (gfx950) in-tree with hipcc.  `python visiontransformer_amd/build.py`.

The library is plain HIP + C ABI (no torch headers), so a build is a handful of hipcc
invocations; objects are cached under csrc/build/ and rebuilt when their source or a header
is newer.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libvitmi_hip.so")
ARCH = "gfx950"
SOURCES = ["runtime.hip", "elementwise.hip", "gemm.hip", "gemm_fast.hip", "attention.hip", "rollout.hip", "blocks.hip", "persample.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "vitmi.h")]
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libvitmi_hip.so")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    hdrs = HEADERS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if _stale(obj, [path] + hdrs):
        cmd = [_hipcc()] + FLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(r for _, r in results)
    if rebuilt or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB}")
    elif verbose:
        print(f"{LIB} up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
