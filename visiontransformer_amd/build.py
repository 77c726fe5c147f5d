"""Builds libvitmi_hip.so (gfx950) in-tree with hipcc.  `python visiontransformer_amd/build.py`.

The library is plain HIP + C ABI (no torch headers), so a build is a handful of hipcc
invocations; objects are cached under csrc/build/ and rebuilt when their source or a header
is newer.

`python visiontransformer_amd/build.py --host-sanitize` builds libvitmi_hip_hostsan.so beside it: the same sources with
the HOST side of every translation unit (argument checks, workspace arithmetic, launch geometry) under
-fsanitize=undefined in trap mode (no sanitizer runtime to load, a finding stops the process with SIGILL); device code
is unchanged (GPU sanitizers are not available on the target pool).  Load it with VITMI_LIB=<path>.
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
SOURCES = ["runtime.hip", "elementwise.hip", "gemm.hip", "gemm_fast.hip", "attention.hip", "rollout.hip", "blocks.hip", "persample.hip", "small.hip"]
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


HOSTSAN_FLAGS = ["-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fsanitize-trap=undefined", "-Xarch_host",
                 "-fno-sanitize=vptr,function"]
HOSTSAN_LIB = os.path.join(HERE, "libvitmi_hip_hostsan.so")


def _compile(src, obj_dir=None, extra=()):
    obj = os.path.join(obj_dir or OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    hdrs = HEADERS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if _stale(obj, [path] + hdrs):
        cmd = [_hipcc()] + FLAGS + list(extra) + ["-c", path, "-o", obj]
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


def build_host_sanitized(verbose=True):
    obj_dir = os.path.join(CSRC, "build_hostsan")
    os.makedirs(obj_dir, exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        results = list(ex.map(lambda src: _compile(src, obj_dir, HOSTSAN_FLAGS), SOURCES))
    objs = [o for o, _ in results]
    if any(r for _, r in results) or _stale(HOSTSAN_LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", HOSTSAN_LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {HOSTSAN_LIB}")
    return HOSTSAN_LIB


if __name__ == "__main__":
    if "--host-sanitize" in sys.argv:
        build_host_sanitized()
    else:
        build(force="--force" in sys.argv)
