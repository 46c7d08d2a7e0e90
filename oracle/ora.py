"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product never imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, os.path.join(_ROOT, "ray-tracer_amd", "python"))
import rtapi  # noqa: E402
from rtapi import D3, D6, D16, Camera, Material, Stats, d3  # noqa: E402,F401


class Comps(C.Structure):
    _fields_ = [("t", C.c_double), ("shape_index", C.c_int32), ("inside", C.c_int32),
                ("point", D3), ("over_point", D3), ("under_point", D3), ("eyev", D3),
                ("normalv", D3), ("reflectv", D3), ("n1", C.c_double), ("n2", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "librt_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _load():
    lib = C.CDLL(build())
    vp, i, d = C.c_void_p, C.c_int, C.c_double
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    pu8 = C.POINTER(C.c_uint8)
    sig = {
        "mul_point": (None, [pd, pd, pd]), "mul_vector": (None, [pd, pd, pd]),
        "reflect": (None, [pd, pd, pd]),
        "set_bounds_of": (None, [vp, i, i, pd]),
        "parent_of": (i, [vp, i, i, C.POINTER(C.c_int)]),
        "local_intersect": (i, [i, pd, pd, pd]), "local_normal_at": (None, [i, pd, pd]),
        "bounds_intersects": (i, [pd, pd, pd]),
        "intersection_allowed": (i, [i, i, i, i]),
        "filter_intersections": (i, [i, pi, i, pd, pi, i, pd, pi]),
        "object_intersect": (i, [vp, i, i, pd, pd, pd, pi, i]),
        "world_intersect": (i, [vp, pd, pd, pd, pi, i]),
        "hit": (i, [pd, i]),
        "color_at": (None, [vp, pd, pd, i, pd]),
        "is_shadowed": (i, [vp, pd, i]),
        "prepare_computations": (None, [vp, i, pd, pd, pd, pi, i, C.POINTER(Comps)]),
        "shade_hit": (None, [vp, C.POINTER(Comps), i, pd]),
        "reflected_color": (None, [vp, C.POINTER(Comps), i, pd]),
        "refracted_color": (None, [vp, C.POINTER(Comps), i, pd]),
        "schlick": (d, [C.POINTER(Comps)]),
        "world_to_object": (None, [vp, i, i, pd, pd]),
        "normal_to_world": (None, [vp, i, i, pd, pd]),
        "normal_at": (None, [vp, i, pd, pd]),
        "pattern_color_at": (None, [C.POINTER(Material), pd, pd]),
        "object_color_at": (None, [vp, i, pd, pd]),
        "shader_lighting": (None, [C.POINTER(Material), pd, pd, pd, pd, pd, pd, i, pd]),
        "object_lighting": (None, [vp, i, i, pd, pd, pd, i, pd]),
        "color_to_rgb": (None, [pd, pu8]),
        "ray_for_pixel": (None, [C.POINTER(Camera), i, i, pd, pd]),
        "render": (i, [vp, C.POINTER(Camera), i, pi, pi, C.c_int64, pu8, pi, pd, pu8,
                       C.POINTER(Stats), i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, "ora_" + name)
        fn.restype, fn.argtypes = res, args
    return lib


LIB = _load()
API = rtapi.HostApi(LIB, "ora_")
CAP = 256


def _xs(fn, *args):
    t = (C.c_double * CAP)(); s = (C.c_int32 * CAP)()
    n = fn(*args, t, s, CAP)
    return [(t[k], s[k]) for k in range(n)]


def world_intersect(w, o, d):
    return _xs(LIB.ora_world_intersect, w.h, d3(o), d3(d))


def object_intersect(w, obj, o, d):
    return _xs(LIB.ora_object_intersect, w.h, obj[0], obj[1], d3(o), d3(d))


def color_at(w, o, d, remaining):
    out = D3(); LIB.ora_color_at(w.h, d3(o), d3(d), remaining, out); return list(out)


def is_shadowed(w, p, light=0):
    return bool(LIB.ora_is_shadowed(w.h, d3(p), light))


def prepare(w, hit, o, d, xs):
    n = len(xs)
    t = (C.c_double * n)(*[x[0] for x in xs]); s = (C.c_int32 * n)(*[x[1] for x in xs])
    c = Comps(); LIB.ora_prepare_computations(w.h, hit, d3(o), d3(d), t, s, n, C.byref(c)); return c


def shade_hit(w, comps, remaining):
    out = D3(); LIB.ora_shade_hit(w.h, C.byref(comps), remaining, out); return list(out)


def reflected_color(w, comps, remaining):
    out = D3(); LIB.ora_reflected_color(w.h, C.byref(comps), remaining, out); return list(out)


def refracted_color(w, comps, remaining):
    out = D3(); LIB.ora_refracted_color(w.h, C.byref(comps), remaining, out); return list(out)


def schlick(comps):
    return LIB.ora_schlick(C.byref(comps))


def local_intersect(shape_type, o, d):
    out = (C.c_double * 2)(); n = LIB.ora_local_intersect(shape_type, d3(o), d3(d), out)
    return [out[k] for k in range(n)]


def local_normal_at(shape_type, p):
    out = D3(); LIB.ora_local_normal_at(shape_type, d3(p), out); return list(out)


def vec3(fn, *args):
    out = D3(); fn(*args, out); return list(out)


def render(w, cam, max_depth=5, pixels=None, threads=1, want_color=False):
    """Camera::render (camera.rs:39-54) over the whole image or a pixel list.
    Returns dict(rgb, hit, tie, color?, stats)."""
    if pixels is None:
        n = cam.hsize * cam.vsize; xs = ys = None
    else:
        px = np.ascontiguousarray(pixels[0], dtype=np.int32)
        py = np.ascontiguousarray(pixels[1], dtype=np.int32)
        n = len(px)
        xs = px.ctypes.data_as(C.POINTER(C.c_int32)); ys = py.ctypes.data_as(C.POINTER(C.c_int32))
    rgb = np.zeros((n, 3), np.uint8); hit = np.zeros(n, np.int32); tie = np.zeros(n, np.uint8)
    col = np.zeros((n, 3), np.float64) if want_color else None
    st = Stats()
    LIB.ora_render(w.h, C.byref(cam), max_depth, xs, ys, n,
                   rgb.ctypes.data_as(C.POINTER(C.c_uint8)), hit.ctypes.data_as(C.POINTER(C.c_int32)),
                   col.ctypes.data_as(C.POINTER(C.c_double)) if want_color else None,
                   tie.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(st), threads)
    out = dict(rgb=rgb, hit=hit, tie=tie, stats=st)
    if want_color:
        out["color"] = col
    if pixels is None:
        out["rgb"] = rgb.reshape(cam.vsize, cam.hsize, 3); out["hit"] = hit.reshape(cam.vsize, cam.hsize)
        out["tie"] = tie.reshape(cam.vsize, cam.hsize)
        if want_color:
            out["color"] = col.reshape(cam.vsize, cam.hsize, 3)
    return out
