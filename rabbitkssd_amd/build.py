"""Builds librabbitkssd.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

    python -m rabbitkssd_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librabbitkssd.so")
TOOL = os.path.join(HERE, "rabbit_kssd")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(ROOT, "include", "rabbitkssd.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    # host tool (C++ above the C ABI): rabbit_kssd with the reference's command line
    host_src = os.path.join(HERE, "host", "rabbit_kssd.cpp")
    host_deps = [host_src, os.path.join(HERE, "host", "formats.hpp"), os.path.join(ROOT, "include", "rabbitkssd.h"), LIB]
    if force or _newer(TOOL, host_deps):
        # plain g++: the host side sees only the C ABI (no HIP headers, no device pass)
        run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), host_src, "-o", TOOL,
             "-L" + HERE, "-lrabbitkssd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
