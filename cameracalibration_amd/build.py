"""Builds libmdisp.so (HIP, gfx950) in-tree with hipcc.  No JIT, no torch extension machinery:
the library is a plain C-ABI shared object loaded through ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmdisp.so")
SOURCES = ["api.hip", "census.hip", "sgm.hip", "sgm2.hip", "sgm3.hip", "sgbm_bt.hip", "bm.hip", "fgs.hip", "legacy.hip", "ncc.hip", "stereo.hip", "filters.hip", "gif_mc.hip", "host.hip", "multi.hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mdisp.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
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
