"""Independent Python restatement of the reference's scene loader (yaml.rs:22-509).

Drives any `rtapi.HostApi` (the product's host mirror or, in tests, the oracle's
builder), so the product's C++ loader (ray-tracer_amd/host/yaml_loader.cpp) can be
cross-checked: both must produce the same builder calls for the same file.

YAML is read with PyYAML's BaseLoader (every scalar a string) and typed here with
yaml-rust 0.4.5's rules (Yaml::from_str), so PyYAML's own YAML-1.1 typing does not
leak in.  Quoted scalars are not distinguished from plain ones (scene files never
quote numbers).
"""
import math
import re

import yaml

from rtapi import (CUBE, DIFFERENCE, INTERSECTION, OBJ_CSG, OBJ_GROUP, OBJ_SHAPE, PAT_CHECKERS,
                   PAT_GRADIENT, PAT_PLAIN, PAT_RINGS, PAT_STRIPES, PLANE, SPHERE, UNION, d3)

_F64 = re.compile(r"^[+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf|infinity|nan)$", re.I)
_I64 = re.compile(r"^[+-]?\d+$")


class SceneError(ValueError):
    pass


def rust_f64(s):
    if not _F64.match(s):
        return None
    return float(s)


def classify(s):
    """-> ('int', v) | ('real', v) | ('null',) | ('bool',) | ('str', s)"""
    try:
        if s.startswith("0x"):
            return ("int", int(s[2:], 16))
        if s.startswith("0o"):
            return ("int", int(s[2:], 8))
    except ValueError:
        pass
    if s.startswith("+") and _I64.match(s[1:]) and not s[1:].startswith(("+", "-")):
        return ("int", int(s[1:]))
    if s in ("~", "null"):
        return ("null",)
    if s in ("true", "false"):
        return ("bool",)
    if _I64.match(s) and -2 ** 63 <= int(s) < 2 ** 63:
        return ("int", int(s))
    if s in (".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF"):
        return ("real", math.inf)
    if s in ("-.inf", "-.Inf", "-.INF"):
        return ("real", -math.inf)
    if s in (".nan", "NaN", ".NAN"):
        return ("real", math.nan)
    f = rust_f64(s)
    if f is not None:
        return ("real", f)
    return ("str", s)


def f64_const(w):                                   # yaml.rs:497-509
    f = rust_f64(w)
    if f is not None:
        return f
    if w == "pi":
        return math.pi
    if w == "-pi":
        return -math.pi
    return math.nan


def _div(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def f64_expr(e):                                    # yaml.rs:466-495
    if not isinstance(e, str):
        raise SceneError("expected a number")
    c = classify(e)
    if c[0] in ("real", "int"):
        return float(c[1])
    if c[0] != "str":
        raise SceneError("expected a number")
    words = e.split()
    res = f64_const(words[0])
    op = None
    for w in words[1:]:
        if op is None:
            op = w
        else:
            r = f64_const(w)
            if op == "+":
                res += r
            elif op == "-":
                res -= r
            elif op == "*":
                res *= r
            elif op == "/":
                res = _div(res, r)
    return res


def f64_tuple(e):
    if not isinstance(e, list) or len(e) != 3:
        raise SceneError("expected 3 numbers")
    return [f64_expr(x) for x in e]


def _str(e):
    if not isinstance(e, str) or classify(e)[0] != "str":
        raise SceneError("expected a string, got %r" % (e,))
    return e


class Loader:
    def __init__(self, api, world_offset=None, wrap=None):
        """world_offset (dx, dy, dz): every root object, light and camera is moved by it (test scenes far from the origin);
        wrap (scale, (dx, dy, dz)): all root objects go into ONE extra root group with transform translate . scale, lights and
        camera are mapped the same way (test scenes inside a strongly scaled / far translated group).  Not part of the DSL."""
        self.api = api
        self.world_offset = world_offset
        self.wrap = wrap
        self.wrap_group = None
        self.world = api.world_new()
        self.cameras = []
        self.materials = {}
        self.groups = {}

    def transform(self, e):                         # yaml.rs:246-279
        ops = []
        if not isinstance(e, list):
            raise SceneError("transform must be a list")
        for t in e:
            if not isinstance(t, dict) or len(t) != 1:
                raise SceneError("bad transform entry")
            (k, v), = t.items()
            if k in ("scale", "translate"):
                ops.append((k, f64_tuple(v)))
            elif k in ("rotate-x", "rotate-y", "rotate-z"):
                ops.append((k, f64_expr(v)))
        return self.api.compose(ops)

    def map_point(self, p):
        p = list(p)
        if self.wrap is not None:
            sc, off = self.wrap
            p = [p[k] * sc + off[k] for k in range(3)]
        if self.world_offset is not None:
            p = [p[k] + self.world_offset[k] for k in range(3)]
        return p

    def root_transform(self, t):
        if self.world_offset is not None:
            t = self.api.mat_mul(self.api.translation(*self.world_offset), t)
        return t

    def add_root(self, obj):
        if self.wrap is None:
            self.world.add_object_to_scene(obj)
            return
        if self.wrap_group is None:
            sc, off = self.wrap
            self.wrap_group = self.world.add_group(self.api.compose([("scale", [sc, sc, sc]), ("translate", list(off))]))
            self.world.add_object_to_scene((OBJ_GROUP, self.wrap_group))
        self.world.set_group_of(obj, self.wrap_group)

    def camera(self, e):                            # yaml.rs:75-87
        w, h = classify(e["width"]), classify(e["height"])
        if w[0] != "int" or h[0] != "int":
            raise SceneError("camera size must be integer")
        cam = self.api.camera(w[1], h[1], f64_expr(e["field-of-view"]))
        if "up" in e:
            cam.transform = self.api.view_transform(self.map_point(f64_tuple(e["from"])), self.map_point(f64_tuple(e["to"])), f64_tuple(e["up"]))
        self.cameras.append(cam)

    def pattern(self, e, m):                        # yaml.rs:427-464
        ty = _str(e["type"])
        m.pattern_transform = self.api.identity()
        m.color_a = d3([0, 0, 0]); m.color_b = d3([0, 0, 0])
        kinds = {"stripes": PAT_STRIPES, "rings": PAT_RINGS, "checkers": PAT_CHECKERS, "gradient": PAT_GRADIENT}
        if ty in ("plain", "uniform"):
            m.pattern_kind = PAT_PLAIN; m.color_a = d3(f64_tuple(e["color"]))
        elif ty in kinds:
            m.pattern_kind = kinds[ty]
            m.color_a = d3(f64_tuple(e["color1"])); m.color_b = d3(f64_tuple(e["color2"]))
        else:
            m.pattern_kind = PAT_PLAIN
        if "transform" in e:
            m.pattern_transform = self.transform(e["transform"])

    def material(self, e):                          # yaml.rs:357-416
        m = self.api.material()
        name = None
        for k, v in e.items():
            if k == "name":
                name = v if isinstance(v, str) and classify(v)[0] == "str" else None
            elif k == "extends":
                m = self.materials[_str(v)].copy()
            elif k == "color":
                m.pattern_kind = PAT_PLAIN; m.pattern_transform = self.api.identity()
                m.color_a = d3(f64_tuple(v)); m.color_b = d3([0, 0, 0])
            elif k == "pattern":
                self.pattern(v, m)
            elif k in ("ambient", "diffuse", "specular", "shininess", "reflective", "transparency"):
                setattr(m, k, f64_expr(v))
            elif k == "refractive-index":
                m.refractive_index = f64_expr(v)
        if name is not None:
            self.materials[name] = m

    def group(self, e):                             # yaml.rs:294-333
        name, parent, t = None, None, self.api.identity()
        for k, v in e.items():
            if k == "name":
                name = v if isinstance(v, str) and classify(v)[0] == "str" else None
            elif k == "parent":
                parent = self.groups[_str(v)]
            elif k == "transform":
                t = self.transform(v)
        g = self.world.add_group(t if parent is not None else self.root_transform(t))
        if parent is not None:
            self.world.set_group_of((OBJ_GROUP, g), parent)
        else:
            self.add_root((OBJ_GROUP, g))
        if name is None:
            raise SceneError("Group without name.")
        self.groups[name] = g

    def shape_properties(self, kind, e, root=False):            # yaml.rs:108-137
        m, t, grp = self.api.material(), self.api.identity(), None
        for k, v in e.items():
            if k == "material":
                m = self.materials[_str(v)]
            elif k == "transform":
                t = self.transform(v)
            elif k == "group":
                grp = v if isinstance(v, str) and classify(v)[0] == "str" else None
        if root and grp is None:
            t = self.root_transform(t)
        return self.world.add_shape(kind, m, t), grp

    def toplevel_shape(self, kind, e):              # yaml.rs:95-106
        s, grp = self.shape_properties(kind, e, root=True)
        if grp is not None:
            self.world.set_group_of((OBJ_SHAPE, s), self.groups[grp])
        else:
            self.add_root((OBJ_SHAPE, s))

    def obj(self, e):                               # yaml.rs:139-171
        what = _str(e["add"])
        kinds = {"plane": PLANE, "sphere": SPHERE, "cube": CUBE}
        if what in kinds:
            s, grp = self.shape_properties(kinds[what], e)
            if grp is not None:
                raise SceneError("CSG operand names a group")
            return (OBJ_SHAPE, s)
        if what == "csg":
            return (OBJ_CSG, self.csg(e))
        raise SceneError("bad CSG operand %r" % what)

    def csg(self, e, root=False):                   # yaml.rs:173-216
        t, op, operands = self.api.identity(), None, None
        for k, v in e.items():
            if k == "operator":
                op = {"union": UNION, "intersection": INTERSECTION, "difference": DIFFERENCE}[_str(v)]
            elif k == "transform":
                t = self.transform(v)
            elif k == "operands":
                if len(v) != 2:
                    raise SceneError("CSG needs two operands")
                l = self.obj(v[0]); r = self.obj(v[1])
                operands = (l, r)
        if op is None or operands is None:
            raise SceneError("CSG needs operator and operands")
        return self.world.add_csg(op, operands[0], operands[1], self.root_transform(t) if root else t)

    def load(self, text):                           # yaml.rs:22-73
        doc = yaml.load(text, Loader=yaml.BaseLoader)
        shapes = {"plane": PLANE, "sphere": SPHERE, "cube": CUBE}
        for e in doc:
            if "add" in e:
                what = _str(e["add"])
                if what == "camera":
                    self.camera(e)
                elif what == "light":
                    self.world.add_light(self.map_point(f64_tuple(e["at"])), f64_tuple(e["intensity"]))
                elif what in shapes:
                    self.toplevel_shape(shapes[what], e)
                elif what == "csg":
                    self.add_root((OBJ_CSG, self.csg(e, root=True)))
            elif "define" in e:
                what = _str(e["define"])
                if what == "material":
                    self.material(e)
                elif what == "group":
                    self.group(e)
        self.world.build()
        return self.world, self.cameras


def load(text, api, world_offset=None, wrap=None):
    return Loader(api, world_offset, wrap).load(text)
