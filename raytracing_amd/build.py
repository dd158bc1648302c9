"""Builds the native libraries in-tree.

    python -m raytracing_amd.build            # HIP backend + host layer + rt executable
    python -m raytracing_amd.build --oracle   # also the CPU oracle (test infrastructure)

hipcc cross-compiles for gfx950 without a GPU present.  Artefacts:
    raytracing_amd/librtx_hip.so   kernels + C ABI (include/rtx.h) + host layer (include/rt_host.h)
    raytracing_amd/rt              command-line renderer (same flags as the reference's rt)
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing_amd")
CSRC = os.path.join(PKG, "csrc")
HOST_SRC = [os.path.join(CSRC, "host", f) for f in ("material.cpp", "scene.cpp", "host_api.cpp", "jpeg_decoder.cpp")]
HIP_SRC = os.path.join(CSRC, "rtx_hip.hip")
LIB = os.path.join(PKG, "librtx_hip.so")
EXE = os.path.join(PKG, "rt")

# -ffp-contract=off: pixel arithmetic must not be fused into FMA (the reference is an SSE2 build)
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall",
            "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _all_sources():
    out = [HIP_SRC] + HOST_SRC
    for d in (CSRC, os.path.join(CSRC, "host"), os.path.join(ROOT, "include")):
        out += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hpp", ".h"))]
    return out


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP backend cannot be built")
    return exe


def build_hip(force=False, verbose=True):
    srcs = _all_sources()
    if not force and _newer(LIB, srcs) and _newer(EXE, srcs):
        return LIB
    cmd = [hipcc()] + HIPFLAGS + ["-shared", "-Wl,-Bsymbolic", HIP_SRC] + HOST_SRC + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    cmd = [hipcc(), "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "host", "rt_main.cpp"),
           "-L" + PKG, "-lrtx_hip", "-Wl,-rpath,$ORIGIN", "-o", EXE]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_oracle(verbose=True):
    """CPU oracle + (when /root/reference exists) the reference build.  Test infrastructure."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"], stdout=out)


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
    if "--oracle" in sys.argv:
        build_oracle()
