"""Registers, spills and scratch memory of every kernel in a built library, read from the code object inside it
(no GPU needed).  The compiler's register allocation in these kernels moves with small changes of the sources; a
kernel that suddenly keeps its arguments in scratch memory or spills inside a tree walk shows here before it shows
in a frame time.  usage: python scripts/kernel_resources.py [library]"""
import os
import re
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def short_name(mangled):
    """k_render<false, true> from _ZN12_GLOBAL__N_18k_renderILb0ELb1EEEv...; None for kernels that are not this library's own"""
    m = re.match(r"_ZN(?:12_GLOBAL__N_1|3rtx)\d+(k_[a-z_0-9]+?)(I(?:Lb[01]E)+E|I[A-Za-z0-9_]+E)?E", mangled)
    if not m:
        return None
    flags = re.findall(r"Lb([01])E", m.group(2) or "")
    return m.group(1) + ("<" + ", ".join("true" if f == "1" else "false" for f in flags) + ">" if flags else "")


def kernel_resources(lib):
    """{kernel name: {field: int}} for the gfx950 code object in `lib`"""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    out = {}
    for entry in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", entry)
        if not name:
            continue
        vals = {}
        for f in FIELDS:
            m = re.search(r"\.%s:\s+(\d+)" % f, entry)
            vals[f] = int(m.group(1)) if m else 0
        if short_name(name.group(1)):
            out[short_name(name.group(1))] = vals
    return out


def flat_instructions(lib):
    """{kernel name: number of flat_* memory instructions} for the gfx950 code object in `lib`.  A flat access is what the
    compiler emits when it no longer knows whether a pointer means LDS or device memory -- typically after merging an LDS
    read and a global read into one load of a selected address; it is slower than either and waits for both memory pipes."""
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
        asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)
    out, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = short_name(m.group(1))
            if name:
                out[name] = 0
        elif name and re.match(r"\s+flat_", line):
            out[name] += 1
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "raytracing_amd", "librtx_hip.so")
    res = kernel_resources(lib)
    print("%-34s %5s %5s %11s %11s %8s %6s" % ("kernel", "VGPR", "SGPR", "VGPR spills", "SGPR spills", "scratch", "LDS"))
    for k in sorted(res):
        v = res[k]
        print("%-34s %5d %5d %11d %11d %8d %6d" % (k, v["vgpr_count"], v["sgpr_count"], v["vgpr_spill_count"], v["sgpr_spill_count"],
                                                   v["private_segment_fixed_size"], v["group_segment_fixed_size"]))
    flat = {k: n for k, n in flat_instructions(lib).items() if n}
    print("flat memory instructions:", flat if flat else "none")
