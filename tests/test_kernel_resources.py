"""The hot kernels of the built library stay within their register budget: no scene description in scratch memory, no
spills inside the tree walks.  (Read from the code object inside librtx_hip.so -- no GPU needed.  The compiler's register
allocation in these kernels moves with small changes of the sources; once, folding four normalizations into one made the
general build of the level-by-level trace kernel keep its arguments in scratch memory: +39 % memory traffic, silently.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

#           kernel                        most scratch bytes per lane, most spilled VGPRs, most VGPRs
LIMITS = {"k_render<false, false, false>": (512, 16, 168),  # three waves per SIMD; the frames of the recursion beyond the LDS slots live in scratch
          "k_render<false, false, true>": (640, 48, 168),   # the quad-mode launch beside it: must fit the same slots (its look-ahead hit record spills)
          "k_wf_trace<false, false>": (0, 0, 128),   # the pipeline's walking kernels: four waves per SIMD ...
          "k_wf_shadow<false, false>": (0, 0, 128),
          "k_wf_trace<false, true>": (112, 32, 96),   # ... and five in the general build (a stress-scene frame is faster with these
          "k_wf_shadow<false, true>": (0, 0, 96),     # spills and five waves than without them and four; each spilled register is
          "k_wf_shadow_pool<false>": (64, 16, 96),    # HBM-side traffic: at 40 and 34 spills the two hot ones moved a third of a frame's bytes)
          "k_wf_shade<false, false>": (0, 0, 128),
          "k_wf_shade<false, true>": (0, 0, 128)}


@pytest.fixture(scope="module")
def resources():
    import kernel_resources
    lib = os.path.join(ROOT, "raytracing_amd", "librtx_hip.so")
    if not os.path.exists(lib) or not os.path.exists(kernel_resources.LLVM + "/llvm-readelf"):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    return kernel_resources.kernel_resources(lib)


def test_every_hot_kernel_is_in_the_library(resources):
    assert set(LIMITS) <= set(resources), sorted(resources)


@pytest.mark.parametrize("kernel", sorted(LIMITS))
def test_hot_kernels_keep_within_their_registers(resources, kernel):
    scratch, spills, vgprs = LIMITS[kernel]
    r = resources[kernel]
    print(kernel, r)
    assert r["private_segment_fixed_size"] <= scratch and r["vgpr_spill_count"] <= spills, (kernel, r)
    assert r["vgpr_count"] <= vgprs


def test_no_flat_memory_instructions_in_the_render_kernels():
    """Every memory access of the kernels that trace rays says which memory it means (LDS, device memory, scratch).  A
    flat_* instruction appears when the compiler merges an LDS read and a device-memory read into one load of a selected
    address (the staged list entries and their tails in device memory, the frame heads in LDS and in scratch): it is
    slower than either and makes the wave wait for both memory pipes.  It came and went with unrelated edits -- 2-4 % of a
    3840x2160 frame -- until the staged reads were typed as LDS reads."""
    import kernel_resources
    lib = os.path.join(ROOT, "raytracing_amd", "librtx_hip.so")
    if not os.path.exists(lib) or not os.path.exists(kernel_resources.LLVM + "/llvm-objdump"):
        pytest.skip("needs the built library and the ROCm LLVM tools")
    flat = kernel_resources.flat_instructions(lib)
    render = {k: n for k, n in flat.items() if k.startswith(("k_render", "k_wf_", "k_trace_rays", "k_intersect_rays", "k_tile_probe"))}
    assert render, sorted(flat)
    assert not {k: n for k, n in render.items() if n}, render
