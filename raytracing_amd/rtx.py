"""ctypes bindings for include/rtx.h and include/rt_host.h.

`Backend(path)` wraps one shared library implementing the two headers.  The product library is
`HIP_LIB_PATH`; `hip_backend()` loads it and raises if it has not been built -- nothing here falls
back to another implementation.  (The test suite instantiates `Backend` a second time on the CPU
oracle, which implements the same headers.)
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# (RTX_HIP_LIB: another build of the same library, for A/B measurements)
HIP_LIB_PATH = os.environ.get("RTX_HIP_LIB") or os.path.join(_PKG, "librtx_hip.so")
HIP_FAST_LIB_PATH = os.path.join(_PKG, "librtx_hip_fast.so")  # the tolerance build of the same sources (see build.py)
ASSETS_DIR = os.path.join(os.path.dirname(_PKG), "assets")

MASK_SPHERES, MASK_CUBES, MASK_TRIANGLES, MASK_MODEL, MASK_ISLANDS, MASK_ALL = 1, 2, 4, 8, 16, 31
OPT_STATS, OPT_PRUNE, OPT_FAR_CLIP, OPT_SURFACE_BIAS, OPT_MESH_GRID, OPT_BVH_BUILDER, OPT_SCHEDULE_FEEDBACK, OPT_QUAD, OPT_SPP, OPT_PIPELINE, OPT_PIPELINE_BUDGET_MB, OPT_PIN_OUTPUT, OPT_TILE_PROBE, OPT_QUAD_KERNEL, OPT_RAY_POOL = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15


class RtxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("rtx error %d: %s" % (code, text))
        self.code = code


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("basis", C.c_float * 9), ("tan_half_fov", C.c_float), ("add_origin", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("depth", C.c_int), ("row_first", C.c_int), ("row_step", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("sky_lookups", C.c_uint64), ("mesh_box_hits", C.c_uint64), ("bvh_node_tests", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("triangle_full", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("loose_tri_tests", C.c_uint64), ("box_tests", C.c_uint64), ("plane_tests", C.c_uint64),
                ("island_tests", C.c_uint64), ("lit_lights", C.c_uint64), ("node_wave_trips", C.c_uint64), ("tri_wave_trips", C.c_uint64),
                ("kernel_ms", C.c_double)]

    @property
    def rays(self):
        """sceneIntersect-equivalent queries traced: primary + secondary + shadow"""
        return self.primary_rays + self.secondary_rays + self.shadow_rays

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


# every symbol include/rtx.h and include/rt_host.h declare; the test suite checks the libraries export them all
RTX_SYMBOLS = ["rtx_last_error", "rtx_create", "rtx_destroy", "rtx_set_option", "rtx_set_materials", "rtx_set_lights",
               "rtx_set_spheres", "rtx_set_boxes", "rtx_set_triangles", "rtx_set_mesh", "rtx_set_islands", "rtx_set_water",
               "rtx_set_envmap", "rtx_commit", "rtx_render", "rtx_render_device", "rtx_render_device_noalloc", "rtx_get_stats", "rtx_frame_rows",
               "rtx_deinterleave_rows_device", "rtx_trace_rays", "rtx_intersect_rays", "rtx_group_create", "rtx_group_destroy",
               "rtx_group_size", "rtx_group_ctx", "rtx_group_commit", "rtx_group_render", "rtx_group_last_error",
               "rtx_group_set_option", "rtx_group_render_device", "rtx_group_root_stream", "rtx_group_wait", "rtx_host_alloc", "rtx_host_free"]
HOST_SYMBOLS = ["rt_default_materials", "rt_material_index", "rt_default_camera", "rt_load_model", "rt_free",
                "rt_decode_jpeg", "rt_scene_load_default", "rt_scene_load_default_tex", "rt_save_bmp", "rt_main",
                "rt_host_last_error"]

_vp, _i, _f, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
_SIGNATURES = {
    "rtx_last_error": (C.c_char_p, []),
    "rtx_create": (_i, [C.POINTER(_vp), _i]),
    "rtx_destroy": (None, [_vp]),
    "rtx_set_option": (_i, [_vp, _i, _u32]),
    "rtx_set_materials": (_i, [_vp, _vp, _i]),
    "rtx_set_lights": (_i, [_vp, _vp, _i]),
    "rtx_set_spheres": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "rtx_set_boxes": (_i, [_vp, _vp, _vp, _vp, _i]),
    "rtx_set_triangles": (_i, [_vp, _vp, _vp, _i]),
    "rtx_set_mesh": (_i, [_vp, _vp, _i, _vp, _i, _i]),
    "rtx_set_islands": (_i, [_vp, _vp, _i, _f, _i, _vp, _vp]),
    "rtx_set_water": (_i, [_vp, _vp, _i, _i, _f, _f, _i]),
    "rtx_set_envmap": (_i, [_vp, _vp, _i, _i, _i]),
    "rtx_commit": (_i, [_vp]),
    "rtx_render": (_i, [_vp, C.POINTER(Camera), C.POINTER(Frame), _vp, C.POINTER(Stats)]),
    "rtx_render_device": (_i, [_vp, C.POINTER(Camera), C.POINTER(Frame), _vp, _vp]),
    "rtx_render_device_noalloc": (_i, [_vp, C.POINTER(Camera), C.POINTER(Frame), _vp, _vp]),
    "rtx_get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "rtx_frame_rows": (_i, [C.POINTER(Frame)]),
    "rtx_deinterleave_rows_device": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "rtx_trace_rays": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "rtx_intersect_rays": (_i, [_vp, _vp, _vp, _i, _vp]),
    "rtx_group_create": (_i, [C.POINTER(_vp), _vp, _i]),
    "rtx_group_destroy": (None, [_vp]),
    "rtx_group_size": (_i, [_vp]),
    "rtx_group_ctx": (_vp, [_vp, _i]),
    "rtx_group_commit": (_i, [_vp]),
    "rtx_group_render": (_i, [_vp, C.POINTER(Camera), _i, _i, _i, _vp, C.POINTER(Stats)]),
    "rtx_group_last_error": (C.c_char_p, []),
    "rtx_group_set_option": (_i, [_vp, _i, _u32]),
    "rtx_group_render_device": (_i, [_vp, C.POINTER(Camera), _i, _i, _i, C.POINTER(_vp)]),
    "rtx_group_root_stream": (_vp, [_vp]),
    "rtx_group_wait": (_i, [_vp]),
    "rtx_host_alloc": (_vp, [C.c_size_t]),
    "rtx_host_free": (None, [_vp]),
    "rt_default_materials": (_i, [_vp]),
    "rt_material_index": (_i, [C.c_char_p]),
    "rt_default_camera": (_i, [C.POINTER(Camera)]),
    "rt_load_model": (_i, [C.c_char_p, _f, _vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_i)]),
    "rt_free": (None, [_vp]),
    "rt_decode_jpeg": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "rt_scene_load_default": (_i, [_vp, C.c_char_p, C.c_char_p, C.c_uint]),
    "rt_scene_load_default_tex": (_i, [_vp, C.c_char_p, C.c_uint, _vp, _i, _i, _vp, _i, _i]),
    "rt_save_bmp": (_i, [_vp, _i, _i, C.c_char_p]),
    "rt_main": (_i, [_i, C.POINTER(C.c_char_p)]),
    "rt_host_last_error": (C.c_char_p, []),
}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class Backend:
    """One loaded implementation of rtx.h + rt_host.h."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError("%s is missing: build it first (python -m raytracing_amd.build)" % path)
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args

    # ---- error handling ----
    def check(self, code):
        if code != 0:
            text = self.lib.rtx_last_error() or self.lib.rt_host_last_error() or b""
            raise RtxError(code, text.decode("utf-8", "replace"))

    # ---- host layer ----
    def default_camera(self):
        cam = Camera()
        self.check(self.lib.rt_default_camera(C.byref(cam)))
        return cam

    def default_materials(self):
        m = np.zeros((16, 9), np.float32)
        n = self.lib.rt_default_materials(m.ctypes.data)
        return m[:n].copy()

    def load_model(self, path, scale, offset):
        verts, idx, nv, nt = _vp(), _vp(), _i(), _i()
        off = _f32(offset)
        self.check(self.lib.rt_load_model(os.fsencode(path), scale, off.ctypes.data, C.byref(verts), C.byref(nv),
                                          C.byref(idx), C.byref(nt)))
        v = np.ctypeslib.as_array(C.cast(verts, C.POINTER(C.c_float)), shape=(max(nv.value, 1), 3))[:nv.value].copy()
        f = np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_int)), shape=(max(nt.value, 1), 3))[:nt.value].copy()
        self.lib.rt_free(verts)
        self.lib.rt_free(idx)
        return v, f

    def decode_jpeg(self, path):
        p, w, h, n = _vp(), _i(), _i(), _i()
        code = self.lib.rt_decode_jpeg(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h), C.byref(n))
        if code != 0:
            raise RtxError(code, (self.lib.rt_host_last_error() or b"").decode())
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, n.value)).copy()
        self.lib.rt_free(p)
        return a

    def save_bmp(self, bgr, path):
        bgr = _u8(bgr)
        h, w = bgr.shape[0], bgr.shape[1]
        self.check(self.lib.rt_save_bmp(bgr.ctypes.data, w, h, os.fsencode(path)))

    def host_array(self, shape):
        """A uint8 array in page-locked host memory (rtx_host_alloc): rtx_render copies a frame into it at the speed of
        the link.  Freed when the array (and every view of it) is gone."""
        n = int(np.prod(shape))
        p = self.lib.rtx_host_alloc(n)
        if not p:
            raise MemoryError("rtx_host_alloc(%d)" % n)
        buf = (C.c_ubyte * n).from_address(p)
        lib = self.lib
        import weakref
        arr = np.frombuffer(buf, np.uint8).reshape(shape)
        weakref.finalize(buf, lib.rtx_host_free, p)
        return arr

    # ---- contexts ----
    def create(self, device=0):
        return Context(self, device)


class Context:
    """One rtx_ctx.  Scene setters take numpy arrays; render returns a (rows, W, 3) uint8 BGR array."""

    def __init__(self, backend, device=0, handle=None):
        self.backend = backend
        self.lib = backend.lib
        self.owned = handle is None  # contexts of a Group belong to the group
        self.handle = _vp() if handle is None else _vp(handle)
        if handle is None:
            backend.check(self.lib.rtx_create(C.byref(self.handle), device))

    def close(self):
        if self.handle and self.owned:
            self.lib.rtx_destroy(self.handle)
        self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, code):
        self.backend.check(code)

    def set_option(self, option, value):
        if isinstance(value, float):
            value = int(np.float32(value).view(np.uint32))
        self._ck(self.lib.rtx_set_option(self.handle, option, int(value)))

    def set_materials(self, m9):
        m9 = _f32(m9).reshape(-1, 9)
        self._ck(self.lib.rtx_set_materials(self.handle, m9.ctypes.data, len(m9)))

    def set_lights(self, xyzi):
        xyzi = _f32(xyzi).reshape(-1, 4)
        self._ck(self.lib.rtx_set_lights(self.handle, xyzi.ctypes.data, len(xyzi)))

    def set_spheres(self, centers, radii, mat):
        c = _f32(centers).reshape(-1, 3)
        cx, cy, cz = _f32(c[:, 0]), _f32(c[:, 1]), _f32(c[:, 2])
        r, m = _f32(radii).reshape(-1), _i32(mat).reshape(-1)
        self._ck(self.lib.rtx_set_spheres(self.handle, cx.ctypes.data, cy.ctypes.data, cz.ctypes.data, r.ctypes.data,
                                          m.ctypes.data, len(r)))

    def set_boxes(self, corner_a, corner_b, mat):
        a, b, m = _f32(corner_a).reshape(-1, 3), _f32(corner_b).reshape(-1, 3), _i32(mat).reshape(-1)
        self._ck(self.lib.rtx_set_boxes(self.handle, a.ctypes.data, b.ctypes.data, m.ctypes.data, len(m)))

    def set_triangles(self, v9, mat):
        v, m = _f32(v9).reshape(-1, 9), _i32(mat).reshape(-1)
        self._ck(self.lib.rtx_set_triangles(self.handle, v.ctypes.data, m.ctypes.data, len(m)))

    def set_mesh(self, verts, idx, mat):
        v, f = _f32(verts).reshape(-1, 3), _i32(idx).reshape(-1, 3)
        self._ck(self.lib.rtx_set_mesh(self.handle, v.ctypes.data, len(v), f.ctypes.data, len(f), int(mat)))

    def set_islands(self, cx_cz_a_b, plane_y, mat, light_rgb, dark_rgb):
        p, l, d = _f32(cx_cz_a_b).reshape(-1, 4), _f32(light_rgb), _f32(dark_rgb)
        self._ck(self.lib.rtx_set_islands(self.handle, p.ctypes.data, len(p), plane_y, int(mat), l.ctypes.data, d.ctypes.data))

    def set_water(self, rgb, plane_y=-4.5, scale=10.0, mat=5):
        rgb = _u8(rgb)
        self._ck(self.lib.rtx_set_water(self.handle, rgb.ctypes.data, rgb.shape[1], rgb.shape[0], plane_y, scale, int(mat)))

    def set_envmap(self, rgb, row_shift=100):
        rgb = _u8(rgb)
        self._ck(self.lib.rtx_set_envmap(self.handle, rgb.ctypes.data, rgb.shape[1], rgb.shape[0], row_shift))

    def load_default_scene(self, water_rgb, sky_rgb, mask=MASK_ALL, bunny_path=None):
        """Scene::render's scene through the C++ host layer, with decoded textures."""
        bunny_path = bunny_path or os.path.join(ASSETS_DIR, "bunny.obj")
        sky = _u8(sky_rgb)
        if water_rgb is None:
            wp, ww, wh = None, 0, 0
        else:
            water = _u8(water_rgb)
            wp, ww, wh = water.ctypes.data, water.shape[1], water.shape[0]
        self._ck(self.lib.rt_scene_load_default_tex(self.handle, os.fsencode(bunny_path), mask, wp, ww, wh,
                                                    sky.ctypes.data, sky.shape[1], sky.shape[0]))

    def load_default_scene_files(self, resources_dir=None, envmap_path=None, mask=MASK_ALL):
        """Same, decoding water.jpg and the panorama with the host layer's JPEG decoder."""
        resources_dir = resources_dir or ASSETS_DIR
        envmap_path = envmap_path or os.path.join(resources_dir, "envmap.jpg")
        self._ck(self.lib.rt_scene_load_default(self.handle, os.fsencode(resources_dir), os.fsencode(envmap_path), mask))

    def commit(self):
        self._ck(self.lib.rtx_commit(self.handle))

    def frame_rows(self, frame):
        return self.lib.rtx_frame_rows(C.byref(frame))

    def render(self, cam, width, height, depth=4, row_first=0, row_step=1, want_stats=False, out=None):
        """`out`: a (rows, width, 3) uint8 array to render into (e.g. Backend.host_array: page-locked memory)."""
        fr = Frame(width, height, depth, row_first, row_step)
        rows = self.frame_rows(fr)
        if out is None:
            out = np.zeros((rows, width, 3), np.uint8)
        assert out.shape == (rows, width, 3) and out.dtype == np.uint8 and out.flags.c_contiguous
        st = Stats()
        self._ck(self.lib.rtx_render(self.handle, C.byref(cam), C.byref(fr), out.ctypes.data, C.byref(st)))
        return (out, st) if want_stats else out

    def render_device(self, cam, frame, device_ptr, stream_ptr=None):
        self._ck(self.lib.rtx_render_device(self.handle, C.byref(cam), C.byref(frame), _vp(device_ptr), _vp(stream_ptr or 0)))

    def render_device_noalloc(self, cam, frame, device_ptr, stream_ptr=None):
        """rtx_render_device without any device allocation inside the call (static scheduling memory)."""
        self._ck(self.lib.rtx_render_device_noalloc(self.handle, C.byref(cam), C.byref(frame), _vp(device_ptr), _vp(stream_ptr or 0)))

    def deinterleave_device(self, gathered_ptr, n_ranks, rows_per_rank, width, height, frame_ptr, stream_ptr=None):
        self._ck(self.lib.rtx_deinterleave_rows_device(self.handle, _vp(gathered_ptr), n_ranks, rows_per_rank, width,
                                                       height, _vp(frame_ptr), _vp(stream_ptr or 0)))

    def stats(self):
        st = Stats()
        self._ck(self.lib.rtx_get_stats(self.handle, C.byref(st)))
        return st

    def trace_rays(self, orig, direction, depth=4):
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        out = np.zeros_like(o)
        self._ck(self.lib.rtx_trace_rays(self.handle, o.ctypes.data, d.ctypes.data, len(o), depth, out.ctypes.data))
        return out

    def intersect_rays(self, orig, direction):
        o, d = _f32(orig).reshape(-1, 3), _f32(direction).reshape(-1, 3)
        out = np.zeros((len(o), 16), np.float32)
        self._ck(self.lib.rtx_intersect_rays(self.handle, o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data))
        return out


class Group:
    """One rtx_group: several ranks (one GPU each; ids may repeat) rendering row-interleaved shares of a frame,
    gathered on the first one.  `contexts` are the ranks' Context objects: set the same scene on each."""

    def __init__(self, backend, device_ids):
        self.backend, self.lib = backend, backend.lib
        ids = (C.c_int * len(device_ids))(*device_ids)
        self.handle = _vp()
        self._ck(self.lib.rtx_group_create(C.byref(self.handle), ids, len(device_ids)))
        self.contexts = [Context(backend, handle=self.lib.rtx_group_ctx(self.handle, r)) for r in range(len(device_ids))]

    def _ck(self, code):
        if code != 0:
            raise RtxError(code, (self.lib.rtx_group_last_error() or b"").decode("utf-8", "replace"))

    def commit(self):
        self._ck(self.lib.rtx_group_commit(self.handle))

    def set_option(self, option, value):
        self._ck(self.lib.rtx_group_set_option(self.handle, option, int(value)))

    def render_device(self, cam, width, height, depth=4):
        """Enqueues a frame and returns the device address (on the root's GPU) where it will be; see rtx.h."""
        p = _vp()
        self._ck(self.lib.rtx_group_render_device(self.handle, C.byref(cam), width, height, depth, C.byref(p)))
        return p.value

    def root_stream(self):
        return self.lib.rtx_group_root_stream(self.handle)

    def wait(self):
        self._ck(self.lib.rtx_group_wait(self.handle))

    def render(self, cam, width, height, depth=4, want_stats=False, out=None):
        if out is None:
            out = np.zeros((height, width, 3), np.uint8)
        st = Stats()
        self._ck(self.lib.rtx_group_render(self.handle, C.byref(cam), width, height, depth, out.ctypes.data, C.byref(st)))
        return (out, st) if want_stats else out

    def close(self):
        if self.handle:
            for c in self.contexts:
                c.handle = _vp()
            self.lib.rtx_group_destroy(self.handle)
            self.handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_hip = None
_hip_fast = None


def hip_backend():
    """The product backend.  Raises if librtx_hip.so has not been built: there is no fallback."""
    global _hip
    if _hip is None:
        _load_torch_first()
        _hip = Backend(HIP_LIB_PATH)
    return _hip


def hip_fast_backend():
    """The tolerance build (fused multiply-add, fast division / square root / powf): same sources, same C ABI,
    frames within PSNR >= 50 dB / 99.9 % of pixels within 1/255 of the bit-faithful build's instead of equal."""
    global _hip_fast
    if _hip_fast is None:
        _load_torch_first()
        _hip_fast = Backend(HIP_FAST_LIB_PATH)
    return _hip_fast


def _load_torch_first():
    """PyTorch wheels bundle their own copy of the HIP runtime.  A process must not end up with two
    runtimes (the second one sees no GPUs), so when torch is installed it is imported before
    librtx_hip.so, whose libamdhip64 dependency then resolves to the runtime torch already loaded."""
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
