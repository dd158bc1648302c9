"""Builds a variant of the exact library for A/B measurements: python scripts/build_variant.py NAME [-DFLAG=VALUE ...]
-> raytracing_amd/librtx_hip_NAME.so (same sources and flags as librtx_hip.so, plus the given ones)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import build as b
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(b.PKG, "librtx_hip_%s.so" % name)
cmd = [b.hipcc()] + b.HIPFLAGS + extra + ["-shared", "-Wl,-Bsymbolic", b.HIP_SRC, b.GROUP_SRC] + b.HOST_SRC + ["-ldl", "-o", out]
subprocess.check_call(cmd)
print(out)
