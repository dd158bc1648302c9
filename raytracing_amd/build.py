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
GROUP_SRC = os.path.join(CSRC, "rtx_group.cpp")
LIB = os.path.join(PKG, "librtx_hip.so")
LIB_FAST = os.path.join(PKG, "librtx_hip_fast.so")
EXE = os.path.join(PKG, "rt")

# -ffp-contract=off: pixel arithmetic must not be fused into FMA (the reference is an SSE2 build)
# -fno-slp-vectorize: the SLP vectoriser pairs scalar float operations into packed ones (v_pk_mul_f32, v_pk_add_f32: same
# IEEE results), and in these kernels the moves that line the operands up in register pairs cost more than the pairs save:
# 4-6 % of the frame time, measured build against build
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
          "-I" + os.path.join(ROOT, "include")]
HIPFLAGS = COMMON + ["-ffp-contract=off"]
# The tolerance build of the same sources (librtx_hip_fast.so): fused multiply-adds in the shading arithmetic, powf on the
# exp / log units.  The geometry (rtx_device.hpp and what the sources mark) stays as in the bit-faithful build: a mirror
# ball magnifies an ulp in a direction into another texel of the environment map.  Not bit-faithful: within the stated
# tolerance of the path (PSNR >= 50 dB, >= 99.9 % of pixels within 1/255), which tests/test_gpu_fast_math.py holds it to.
FASTFLAGS = COMMON + ["-ffp-contract=fast-honor-pragmas", "-DRTX_FAST_MATH=1"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _all_sources():
    out = [HIP_SRC, GROUP_SRC, os.path.abspath(__file__)] + HOST_SRC  # (this file holds the compiler flags)
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
    if not force and _newer(LIB, srcs) and _newer(EXE, srcs) and _newer(LIB_FAST, srcs):
        return LIB
    procs = []
    for flags, lib in ((HIPFLAGS, LIB), (FASTFLAGS, LIB_FAST)):  # the two builds side by side
        cmd = [hipcc()] + flags + ["-shared", "-Wl,-Bsymbolic", HIP_SRC, GROUP_SRC] + HOST_SRC + ["-ldl", "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
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
