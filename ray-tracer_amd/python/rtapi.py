"""ctypes mirror of the reference's host-side API (WorldBuilder / Camera / geometry).

The same C function set is exported by the product host library
(`librtx_hip.so`, prefix ``rtxh_``) and — separately implemented — by the CPU
oracle (prefix ``ora_``, loaded only by tests/bench/smoke through
``oracle/ora.py``).  This module holds only the prefix-generic binding; it does
not know about the oracle.

Mirrors: world.rs:15-79 (WorldBuilder), camera.rs:10-37 (Camera),
geometry.rs:12-67 (transform constructors), material.rs:11-51 (Material).
Matrices are 16 doubles, column-major (nalgebra storage).
"""
import ctypes as C

OBJ_GROUP, OBJ_CSG, OBJ_SHAPE = 0, 1, 2          # object_store.rs:17-21
SPHERE, PLANE, CUBE = 0, 1, 2                    # shapes/mod.rs:84-89
UNION, INTERSECTION, DIFFERENCE = 0, 1, 2        # csg.rs:22-26
PAT_PLAIN, PAT_STRIPES, PAT_RINGS, PAT_CHECKERS, PAT_GRADIENT, PAT_TEST = range(6)

D3 = C.c_double * 3
D6 = C.c_double * 6
D16 = C.c_double * 16
EV_COUNT = 16          # event slots of the host-API stats struct (the device stats in rtx.py hold 24)
EV_NAMES = ["xform", "sphere", "plane", "cube", "bounds", "csg_entries", "hit", "levels",
            "light", "pow", "refr", "schlick", "lane_slots", "op_slots", "cull", "x_xform", "x_sphere", "x_box", "backface"]


REF_EVENTS = EV_NAMES[:12]          # reference-defined events (the oracle counts exactly these)


class Material(C.Structure):
    _fields_ = [("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double),
                ("transparency", C.c_double), ("refractive_index", C.c_double),
                ("color_a", D3), ("color_b", D3), ("pattern_transform", D16),
                ("pattern_kind", C.c_int32), ("_pad", C.c_int32)]

    def copy(self):
        m = Material()
        C.memmove(C.byref(m), C.byref(self), C.sizeof(Material))
        return m


class Camera(C.Structure):
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("pixel_size", C.c_double),
                ("half_width", C.c_double), ("half_height", C.c_double), ("transform", D16)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_secondary", C.c_uint64), ("ev", C.c_uint64 * EV_COUNT),
                ("tie_pixels", C.c_uint64), ("ms", C.c_double)]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary

    def events(self):
        return {n: int(self.ev[i]) for i, n in enumerate(EV_NAMES) if i < len(self.ev)}


def d3(v):
    return D3(*[float(x) for x in v])


def mat(v):
    return D16(*[float(x) for x in v])


class HostApi:
    """Prefix-generic binding of the WorldBuilder/Camera/geometry C functions."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix
        f = self._f
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        pd = C.POINTER(C.c_double)
        f("identity", None, [pd])
        f("translation", None, [d, d, d, pd])
        f("scaling", None, [d, d, d, pd])
        for ax in "xyz":
            f("rotation_" + ax, None, [d, pd])
        f("view_transform", None, [pd, pd, pd, pd])
        f("mat_mul", None, [pd, pd, pd])
        f("inverse", None, [pd, pd])
        f("default_material", None, [C.POINTER(Material)])
        f("world_new", vp, [])
        f("world_free", None, [vp])
        f("add_shape", i, [vp, i, C.POINTER(Material), pd])
        f("add_group", i, [vp, pd])
        f("set_group_of", None, [vp, i, i, i])
        f("add_csg", i, [vp, i, i, i, i, i, pd])
        f("add_object_to_scene", None, [vp, i, i])
        f("add_light", i, [vp, pd, pd])
        f("build", None, [vp])
        f("set_material_of", None, [vp, i, C.POINTER(Material)])
        f("get_material_of", None, [vp, i, C.POINTER(Material)])
        f("set_light", None, [vp, i, pd, pd])
        f("get_bounds", None, [vp, i, i, pd])
        f("camera_new", None, [i, i, d, C.POINTER(Camera)])

    def _f(self, name, res, args):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype, fn.argtypes = res, args
        setattr(self, "_" + name, fn)
        return fn

    # ---- geometry.rs
    def identity(self):
        o = D16(); self._identity(o); return o

    def translation(self, x, y, z):
        o = D16(); self._translation(x, y, z, o); return o

    def scaling(self, x, y, z):
        o = D16(); self._scaling(x, y, z, o); return o

    def rotation_x(self, a):
        o = D16(); self._rotation_x(a, o); return o

    def rotation_y(self, a):
        o = D16(); self._rotation_y(a, o); return o

    def rotation_z(self, a):
        o = D16(); self._rotation_z(a, o); return o

    def view_transform(self, frm, to, up):
        o = D16(); self._view_transform(d3(frm), d3(to), d3(up), o); return o

    def mat_mul(self, a, b):
        o = D16(); self._mat_mul(a, b, o); return o

    def inverse(self, a):
        o = D16(); self._inverse(a, o); return o

    def compose(self, ops):
        """yaml.rs:246-279: each op LEFT-multiplies the running transform."""
        t = self.identity()
        for name, arg in ops:
            if name == "scale":
                m = self.scaling(*arg)
            elif name == "translate":
                m = self.translation(*arg)
            elif name == "rotate-x":
                m = self.rotation_x(arg)
            elif name == "rotate-y":
                m = self.rotation_y(arg)
            elif name == "rotate-z":
                m = self.rotation_z(arg)
            else:
                raise ValueError(name)
            t = self.mat_mul(m, t)
        return t

    # ---- material.rs
    def material(self, **kw):
        m = Material(); self._default_material(C.byref(m))
        for k, v in kw.items():
            if k in ("color_a", "color_b"):
                setattr(m, k, d3(v))
            elif k == "color":
                m.color_a = d3(v); m.pattern_kind = PAT_PLAIN
            elif k == "pattern_transform":
                m.pattern_transform = v
            else:
                setattr(m, k, v)
        return m

    # ---- WorldBuilder
    def world_new(self):
        return World(self)

    def camera(self, hsize, vsize, fov, transform=None):
        c = Camera(); self._camera_new(hsize, vsize, fov, C.byref(c))
        if transform is not None:
            c.transform = transform
        return c


class World:
    def __init__(self, api):
        self.api = api
        self.h = C.c_void_p(api._world_new())

    def __del__(self):
        try:
            if self.h:
                self.api._world_free(self.h); self.h = None
        except Exception:
            pass

    def add_shape(self, shape_type, material=None, transform=None):
        t = transform if transform is not None else self.api.identity()
        m = C.byref(material) if material is not None else None
        return self.api._add_shape(self.h, shape_type, m, t)

    def add_group(self, transform=None):
        return self.api._add_group(self.h, transform if transform is not None else self.api.identity())

    def set_group_of(self, obj, group):
        self.api._set_group_of(self.h, obj[0], obj[1], group)

    def add_csg(self, op, left, right, transform=None):
        t = transform if transform is not None else self.api.identity()
        return self.api._add_csg(self.h, op, left[0], left[1], right[0], right[1], t)

    def add_object_to_scene(self, obj):
        self.api._add_object_to_scene(self.h, obj[0], obj[1])

    def add_light(self, pos, intensity):
        return self.api._add_light(self.h, d3(pos), d3(intensity))

    def build(self):
        self.api._build(self.h); return self

    def set_material_of(self, shape, m):
        self.api._set_material_of(self.h, shape, C.byref(m))

    def get_material_of(self, shape):
        m = Material(); self.api._get_material_of(self.h, shape, C.byref(m)); return m

    def set_light(self, i, pos, intensity):
        self.api._set_light(self.h, i, d3(pos), d3(intensity))

    def get_bounds(self, obj):
        o = D6(); self.api._get_bounds(self.h, obj[0], obj[1], o); return list(o)


def shape(i):
    return (OBJ_SHAPE, i)


def group(i):
    return (OBJ_GROUP, i)


def csg(i):
    return (OBJ_CSG, i)
