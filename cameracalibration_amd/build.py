"""Builds libmdisp.so (HIP, gfx950) in-tree with hipcc.  No JIT, no torch extension machinery:
the library is a plain C-ABI shared object loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmdisp.so")
SOURCES = ["api.hip", "census.hip", "sgm.hip", "sgm2.hip", "sgm3.hip", "sgbm_bt.hip", "bm.hip", "fgs.hip", "legacy.hip", "ncc.hip", "stereo.hip", "filters.hip", "gif_mc.hip", "host.hip", "multi.hip"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "mdisp.h")]


def _obj(src):
    return os.path.join(HERE, "lib", "obj", src.replace(".hip", ".o"))


def _outdated(src):
    """A source is recompiled when its object is missing or older than the source or any header."""
    obj = _obj(src)
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + _headers())


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_outdated(s) or os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs = []
    os.makedirs(os.path.join(HERE, "lib", "obj"), exist_ok=True)
    for src in SOURCES:
        if not force and not _outdated(src):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            if os.path.exists(_obj(src)):
                os.remove(_obj(src))          # never link an object of an earlier version of a source that no longer compiles
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES]
    subprocess.check_call(cmd)
    return LIB


def build_host_test():
    """Compile the C++ host-layer exerciser (tests/cpp/host_mirror.cpp) against libmdisp.so with g++."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", "host_mirror.cpp")
    out = os.path.join(root, "tests", "cpp", "_build", "host_mirror")
    deps = [src, os.path.join(root, "include", "mdisp_host.hpp"), os.path.join(root, "include", "mdisp.h"), LIB]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "include"), src, "-o", out,
                           "-L" + os.path.dirname(LIB), "-lmdisp", "-Wl,-rpath,$ORIGIN/../../../cameracalibration_amd/lib"])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host_test())
