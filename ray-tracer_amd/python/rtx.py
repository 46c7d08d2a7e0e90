"""ctypes binding of librtx_hip.so: the C ABI in include/rtx.h plus the host mirror in
include/rtx_host.h.  There is no CPU fallback: importing works without a GPU (so the
symbol table can be checked), but every render needs a HIP device and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

import rtapi
from rtapi import Camera, Material, d3  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(_HERE)
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, os.environ.get("RTX_LIB", "librtx_hip.so"))

OK, E_INVALID, E_DEVICE, E_CAPACITY, E_UNSUPPORTED = 0, -1, -2, -3, -4
F_COUNT_EVENTS, F_NO_SHADOW_EARLY_OUT, F_BANDS_COMPACT = 1, 2, 4
TILE = 8
MAX_DEPTH = 8


class RtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtx error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_secondary", C.c_uint64),
                ("ev", C.c_uint64 * 24), ("overflow_pixels", C.c_uint64), ("kernel_ms", C.c_double),
                ("rays_shadow_skipped", C.c_uint64)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary

    def events(self):
        return {n: int(self.ev[i]) for i, n in enumerate(rtapi.EV_NAMES)}


class Bands(C.Structure):
    _fields_ = [("first", C.c_int32), ("stride", C.c_int32)]


class RtxCamera(C.Structure):
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("pixel_size", C.c_double),
                ("half_width", C.c_double), ("half_height", C.c_double), ("inverse", rtapi.D16)]


class RtxMaterial(C.Structure):
    _fields_ = [("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("color_a", rtapi.D3), ("color_b", rtapi.D3),
                ("pattern_inverse", rtapi.D16), ("pattern_kind", C.c_int32), ("_pad", C.c_int32)]


class SceneDesc(C.Structure):
    _pi, _pd = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    _fields_ = [("n_nodes", C.c_int32), ("node_kind", _pi), ("node_inverse", _pd), ("node_bounds", _pd),
                ("node_first_child", _pi), ("node_child_count", _pi), ("node_csg_op", _pi),
                ("node_shape_index", _pi), ("node_material", _pi), ("n_children", C.c_int32),
                ("children", _pi), ("n_roots", C.c_int32), ("roots", _pi), ("n_materials", C.c_int32),
                ("materials", C.POINTER(RtxMaterial)), ("n_lights", C.c_int32), ("light_position", _pd),
                ("light_intensity", _pd)]


# every symbol include/rtx.h and include/rtx_host.h declare
RTX_SYMBOLS = ["rtx_init", "rtx_shutdown", "rtx_scene_create", "rtx_scene_destroy", "rtx_render",
               "rtx_render_device", "rtx_render_device_noalloc", "rtx_ctx_poll", "rtx_set_option", "rtx_debug_counter",
               "rtx_comm_unique_id", "rtx_comm_init_rank", "rtx_comm_init_all", "rtx_comm_init_loopback", "rtx_comm_concurrency", "rtx_comm_destroy", "rtx_comm_rank",
               "rtx_comm_size", "rtx_render_gathered", "rtx_comm_wait", "rtx_render_multi", "rtx_band_rows_padded",
               "rtx_band_deinterleave", "rtx_render_batch", "rtx_band_rows", "rtx_debug_math", "rtx_strerror", "rtx_last_error"]
RTXH_SYMBOLS = ["rtxh_identity", "rtxh_translation", "rtxh_scaling", "rtxh_rotation_x", "rtxh_rotation_y",
                "rtxh_rotation_z", "rtxh_view_transform", "rtxh_mat_mul", "rtxh_inverse",
                "rtxh_default_material", "rtxh_world_new", "rtxh_world_free", "rtxh_add_shape",
                "rtxh_add_group", "rtxh_set_group_of", "rtxh_add_csg", "rtxh_add_object_to_scene",
                "rtxh_add_light", "rtxh_build", "rtxh_set_material_of", "rtxh_get_material_of",
                "rtxh_set_light", "rtxh_get_bounds", "rtxh_camera_new", "rtxh_world_serialize", "rtxh_scene_create",
                "rtxh_camera_flatten", "rtxh_render", "rtxh_load_yaml", "rtxh_last_error",
                "rtxh_write_png", "rtxh_render_file", "rtxh_render_file_all", "rtxh_render_file_gpus"]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("librtx_hip.so is not built (run `make -C ray-tracer_amd` or "
                          "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i, u32 = C.c_void_p, C.c_int, C.c_uint32
    pu8, pi32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    sig = {
        "rtx_init": (i, [i, C.POINTER(vp)]), "rtx_shutdown": (None, [vp]),
        "rtx_scene_create": (i, [vp, C.POINTER(SceneDesc), C.POINTER(vp)]),
        "rtx_scene_destroy": (None, [vp]),
        "rtx_render": (i, [vp, vp, C.POINTER(RtxCamera), i, Bands, u32, pu8, pi32, C.POINTER(Stats)]),
        "rtx_render_device": (i, [vp, vp, C.POINTER(RtxCamera), i, Bands, u32, vp, vp, vp, C.POINTER(Stats)]),
        "rtx_render_device_noalloc": (i, [vp, vp, C.POINTER(RtxCamera), i, Bands, u32, vp, vp, vp]),
        "rtx_ctx_poll": (i, [vp, i]), "rtx_set_option": (i, [vp, C.c_char_p, i]),
        "rtx_render_batch": (i, [vp, vp, C.POINTER(RtxCamera), i, i, u32, C.POINTER(pu8), C.POINTER(pi32), C.POINTER(Stats)]),
        "rtx_band_rows": (i, [i, Bands]),
        "rtx_comm_unique_id": (i, [vp]), "rtx_comm_init_rank": (i, [vp, vp, i, i, C.POINTER(vp)]),
        "rtx_comm_init_all": (i, [C.POINTER(vp), i, C.POINTER(vp)]), "rtx_comm_init_loopback": (i, [vp, i, C.POINTER(vp)]), "rtx_comm_concurrency": (i, [vp]), "rtx_debug_counter": (C.c_int64, [vp, C.c_char_p]), "rtx_comm_destroy": (None, [vp]),
        "rtx_comm_rank": (i, [vp]), "rtx_comm_size": (i, [vp]),
        "rtx_render_gathered": (i, [vp, vp, C.POINTER(RtxCamera), i, u32, vp, vp, vp, C.POINTER(Stats)]),
        "rtx_comm_wait": (i, [vp, vp]),
        "rtx_render_multi": (i, [C.POINTER(vp), C.POINTER(vp), i, C.POINTER(RtxCamera), i, u32, pu8, pi32, C.POINTER(Stats)]),
        "rtx_band_rows_padded": (i, [i, i]), "rtx_band_deinterleave": (i, [pu8, i, i, i, i, pu8]),
        "rtxh_render_file_gpus": (i, [C.c_char_p, C.c_char_p, i]),
        "rtx_debug_math": (i, [vp, i, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rtx_strerror": (C.c_char_p, [i]), "rtx_last_error": (C.c_char_p, [vp]),
        "rtxh_scene_create": (i, [vp, vp, C.POINTER(vp)]),
        "rtxh_world_serialize": (C.c_int64, [vp, pu8, C.c_int64]),
        "rtxh_camera_flatten": (None, [C.POINTER(Camera), C.POINTER(RtxCamera)]),
        "rtxh_render": (i, [vp, vp, C.POINTER(Camera), i, u32, pu8, pi32, C.POINTER(Stats)]),
        "rtxh_load_yaml": (i, [C.c_char_p, C.POINTER(vp), C.POINTER(Camera), i, C.POINTER(i)]),
        "rtxh_last_error": (C.c_char_p, []),
        "rtxh_write_png": (i, [C.c_char_p, pu8, i, i]),
        "rtxh_render_file": (i, [C.c_char_p, C.c_char_p, i]),
        "rtxh_render_file_all": (i, [C.c_char_p, C.c_char_p, i, C.POINTER(i)]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # an OLDER library named through RTX_LIB for an A/B measurement (tools/ms.py) may lack the newest entry points; the
            # product library must export every one (tests/test_host_cpu.py, __graft_entry__.build)
            if os.environ.get("RTX_LIB") and os.path.basename(LIB_PATH) != "librtx_hip.so":
                continue
            raise
        fn.restype, fn.argtypes = res, args
    return lib


LIB = _load()
API = rtapi.HostApi(LIB, "rtxh_")


class Context:
    """rtx_ctx: one per process per GPU."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = LIB.rtx_init(device, C.byref(h))
        if rc != OK:
            raise RtxError(rc, "rtx_init(%d): %s (no CPU fallback exists)" % (device, LIB.rtx_strerror(rc).decode()))
        self.h = h

    def close(self):
        if self.h:
            LIB.rtx_shutdown(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != OK:
            raise RtxError(rc, "%s: %s" % (LIB.rtx_strerror(rc).decode(), LIB.rtx_last_error(self.h).decode()))

    def scene(self, world):
        return Scene(self, world)

    def set_option(self, name, value):
        """rtx_set_option: scheduling knobs ("lpt", "split", "split_class", "tile_mul", "probe_depth", "probe_ahead", ...); -1 = default."""
        self._check(LIB.rtx_set_option(self.h, name.encode(), int(value)))

    def options(self, **kv):
        for k, v in kv.items():
            self.set_option(k, v)

    def poll(self, wait=True):
        """rtx_ctx_poll: raises RtxError(E_CAPACITY) if an asynchronous render overflowed; clears the condition."""
        self._check(LIB.rtx_ctx_poll(self.h, 1 if wait else 0))

    def debug_math(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float64)
        out = np.zeros_like(a)
        pd = C.POINTER(C.c_double)
        self._check(LIB.rtx_debug_math(self.h, op, a.size, a.ctypes.data_as(pd), b.ctypes.data_as(pd), out.ctypes.data_as(pd)))
        return out

    def render(self, world, cam, max_depth=5, flags=0, want_hit=True):
        """Camera::render (camera.rs:39-54) through rtxh_render: flatten, upload, render, free."""
        rgb = np.zeros((cam.vsize, cam.hsize, 3), np.uint8)
        hit = np.zeros((cam.vsize, cam.hsize), np.int32) if want_hit else None
        st = Stats()
        rc = LIB.rtxh_render(self.h, world.h, C.byref(cam), max_depth, flags,
                             rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                             hit.ctypes.data_as(C.POINTER(C.c_int32)) if want_hit else None, C.byref(st))
        self._check(rc)
        return dict(rgb=rgb, hit=hit, stats=st)


class Scene:
    """rtx_scene: a flattened World resident on the device."""

    def __init__(self, ctx, world=None, desc=None):
        self.ctx = ctx
        h = C.c_void_p()
        if desc is not None:
            rc = LIB.rtx_scene_create(ctx.h, C.byref(desc), C.byref(h))
        else:
            rc = LIB.rtxh_scene_create(ctx.h, world.h, C.byref(h))
        ctx._check(rc)
        self.h = h

    def close(self):
        if self.h:
            LIB.rtx_scene_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def flat_camera(cam):
        rc = RtxCamera(); LIB.rtxh_camera_flatten(C.byref(cam), C.byref(rc)); return rc

    def render(self, cam, max_depth=5, bands=(0, 1), flags=0, want_hit=True, out=None):
        """rtx_render into host arrays.  With F_BANDS_COMPACT the arrays hold only this rank's rows."""
        fc = self.flat_camera(cam)
        b = Bands(*bands)
        rows = LIB.rtx_band_rows(cam.vsize, b) if flags & F_BANDS_COMPACT else cam.vsize
        if out is None:
            rgb = np.zeros((rows, cam.hsize, 3), np.uint8)
            hit = np.full((rows, cam.hsize), -2, np.int32) if want_hit else None
        else:
            rgb, hit = out
        st = Stats()
        rc = LIB.rtx_render(self.ctx.h, self.h, C.byref(fc), max_depth, b, flags,
                            rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                            hit.ctypes.data_as(C.POINTER(C.c_int32)) if hit is not None else None, C.byref(st))
        self.ctx._check(rc)
        return dict(rgb=rgb, hit=hit, stats=st)

    def render_batch(self, cams, max_depth=5, flags=0, want_hit=True, want_stats=True):
        """rtx_render_batch: every camera against the resident scene, one wait for all frames."""
        n = len(cams)
        flat = (RtxCamera * n)(*[self.flat_camera(c) for c in cams])
        rgb = [np.zeros((c.vsize, c.hsize, 3), np.uint8) for c in cams]
        hit = [np.full((c.vsize, c.hsize), -2, np.int32) for c in cams] if want_hit else None
        pu8, pi32 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        prgb = (pu8 * n)(*[a.ctypes.data_as(pu8) for a in rgb])
        phit = (pi32 * n)(*[a.ctypes.data_as(pi32) for a in hit]) if want_hit else None
        st = (Stats * n)() if want_stats else None
        self.ctx._check(LIB.rtx_render_batch(self.ctx.h, self.h, flat, n, max_depth, flags, prgb, phit, st))
        return [dict(rgb=rgb[k], hit=hit[k] if want_hit else None, stats=st[k] if want_stats else None) for k in range(n)]

    def render_device_noalloc(self, cam, d_rgb, d_hit=0, max_depth=5, bands=(0, 1), flags=0, stream=0):
        """rtx_render_device_noalloc: same output, counters and tile lists in the library's static scratch array."""
        fc = self.flat_camera(cam)
        rc = LIB.rtx_render_device_noalloc(self.ctx.h, self.h, C.byref(fc), max_depth, Bands(*bands), flags,
                                           C.c_void_p(d_rgb), C.c_void_p(d_hit) if d_hit else None,
                                           C.c_void_p(stream) if stream else None)
        self.ctx._check(rc)

    def render_device(self, cam, d_rgb, d_hit=0, max_depth=5, bands=(0, 1), flags=0, stream=0, stats=False):
        """rtx_render_device: outputs are device pointers (ints), e.g. torch tensor .data_ptr()."""
        fc = self.flat_camera(cam)
        st = Stats() if stats else None
        rc = LIB.rtx_render_device(self.ctx.h, self.h, C.byref(fc), max_depth, Bands(*bands), flags,
                                   C.c_void_p(d_rgb), C.c_void_p(d_hit) if d_hit else None,
                                   C.c_void_p(stream) if stream else None, C.byref(st) if stats else None)
        self.ctx._check(rc)
        return st


COMM_ID_BYTES = 128


def comm_unique_id():
    """rtx_comm_unique_id: 128 bytes rank 0 creates and the host hands to the other ranks."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = LIB.rtx_comm_unique_id(buf)
    if rc != OK:
        raise RtxError(rc, "rtx_comm_unique_id: RCCL is not available")
    return buf.raw


class Comm:
    """rtx_comm of one rank (one process per GPU): bands {rank, world} + RCCL all-gather + de-interleave."""

    def __init__(self, ctx, unique_id, rank, world, handle=None):
        self.ctx = ctx
        if handle is not None:
            self.h = handle
            return
        h = C.c_void_p()
        ctx._check(LIB.rtx_comm_init_rank(ctx.h, C.c_char_p(unique_id), rank, world, C.byref(h)))
        self.h = h

    @staticmethod
    def init_all(ctxs):
        """rtx_comm_init_all: one process, one context per GPU -> [Comm]"""
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c.h for c in ctxs]); out = (C.c_void_p * n)()
        ctxs[0]._check(LIB.rtx_comm_init_all(hs, n, out))
        return [Comm(ctxs[k], None, k, n, handle=C.c_void_p(out[k])) for k in range(n)]

    @staticmethod
    def loopback(ctx, world):
        """rtx_comm_init_loopback: all `world` ranks on this one GPU, no RCCL (band-mapping tests)"""
        h = C.c_void_p()
        ctx._check(LIB.rtx_comm_init_loopback(ctx.h, world, C.byref(h)))
        return Comm(ctx, None, 0, world, handle=h)

    def close(self):
        if self.h:
            LIB.rtx_comm_destroy(self.h); self.h = None

    def render_gathered(self, scene, cam, d_frame, d_hit_frame=0, max_depth=5, flags=0, stream=0, stats=False):
        fc = Scene.flat_camera(cam)
        st = Stats() if stats else None
        rc = LIB.rtx_render_gathered(self.h, scene.h, C.byref(fc), max_depth, flags, C.c_void_p(d_frame),
                                     C.c_void_p(d_hit_frame) if d_hit_frame else None,
                                     C.c_void_p(stream) if stream else None, C.byref(st) if stats else None)
        self.ctx._check(rc)
        return st

    def wait(self, stream=0):
        self.ctx._check(LIB.rtx_comm_wait(self.h, C.c_void_p(stream) if stream else None))


def render_multi(comms, scenes, cam, max_depth=5, flags=0, want_hit=True, want_stats=False):
    """rtx_render_multi: one host thread, N GPUs -> host image."""
    n = len(comms)
    fc = Scene.flat_camera(cam)
    ch = (C.c_void_p * n)(*[c.h for c in comms]); sh = (C.c_void_p * n)(*[s.h for s in scenes])
    rgb = np.zeros((cam.vsize, cam.hsize, 3), np.uint8)
    hit = np.full((cam.vsize, cam.hsize), -2, np.int32) if want_hit else None
    st = Stats() if want_stats else None
    rc = LIB.rtx_render_multi(ch, sh, n, C.byref(fc), max_depth, flags, rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                              hit.ctypes.data_as(C.POINTER(C.c_int32)) if want_hit else None, C.byref(st) if want_stats else None)
    comms[0].ctx._check(rc)
    return dict(rgb=rgb, hit=hit, stats=st)


def band_deinterleave(gathered, n_ranks, width, height, bpp=3):
    """rtx_band_deinterleave (host): gathered compact band buffers of all ranks -> row-major frame."""
    g = np.ascontiguousarray(gathered, np.uint8).ravel()
    assert g.size == n_ranks * LIB.rtx_band_rows_padded(height, n_ranks) * width * bpp
    out = np.zeros(height * width * bpp, np.uint8)
    rc = LIB.rtx_band_deinterleave(g.ctypes.data_as(C.POINTER(C.c_uint8)), n_ranks, width, height, bpp, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc != OK:
        raise RtxError(rc, "rtx_band_deinterleave")
    return out.reshape(height, width, bpp)


def load_yaml(text, max_cameras=8):
    """load_world_and_cameras_from_str (yaml.rs:22-73) -> (World, [Camera])."""
    wh = C.c_void_p(); cams = (Camera * max_cameras)(); n = C.c_int()
    rc = LIB.rtxh_load_yaml(text.encode(), C.byref(wh), cams, max_cameras, C.byref(n))
    if rc != OK:
        raise RtxError(rc, LIB.rtxh_last_error().decode())
    w = rtapi.World.__new__(rtapi.World)
    w.api, w.h = API, wh
    out = []
    for k in range(min(n.value, max_cameras)):
        c = Camera(); C.memmove(C.byref(c), C.byref(cams[k]), C.sizeof(Camera)); out.append(c)
    return w, out


def serialize(world):
    n = LIB.rtxh_world_serialize(world.h, None, 0)
    buf = np.zeros(n, np.uint8)
    LIB.rtxh_world_serialize(world.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), n)
    return buf.tobytes()


def write_png(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    rc = LIB.rtxh_write_png(path.encode(), rgb.ctypes.data_as(C.POINTER(C.c_uint8)), rgb.shape[1], rgb.shape[0])
    if rc != OK:
        raise RtxError(rc, LIB.rtxh_last_error().decode())
