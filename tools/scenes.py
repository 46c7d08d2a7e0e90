"""Seeded synthetic scenes S2..S5 (SURVEY.md §8d / BASELINE.json configs 2..5) as YAML text in
the reference's scene DSL, plus S1 = a restatement of the reference's example scene layout.

All numbers are written with repr() (shortest round-trip) so every loader parses identical
doubles.  PRNG = PCG32 (XSH-RR, 64-bit state) so other languages can mirror the streams.
`width`/`height` can be overridden to render the same scene at test sizes.
"""
import math

SEED = 20261111


class PCG32:
    def __init__(self, seed, seq=54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, a=0.0, b=1.0):
        return a + (b - a) * (self.next() / 4294967296.0)


def _v(xs):
    return "[" + ", ".join(repr(float(x)) for x in xs) + "]"


def _camera(w, h, frm, to):
    return ("- add: camera\n  width: %d\n  height: %d\n  field-of-view: pi / 3.\n  from: %s\n  to: %s\n  up: [0, 1, 0]\n\n"
            % (w, h, _v(frm), _v(to)))


def _light(at, inten=(1, 1, 1)):
    return "- add: light\n  at: %s\n  intensity: %s\n\n" % (_v(at), _v(inten))


def _material(name, **kw):
    s = "- define: material\n  name: %s\n" % name
    for k, v in kw.items():
        key = k.replace("_", "-")
        if k == "color":
            s += "  color: %s\n" % _v(v)
        elif k == "pattern":
            s += "  pattern:\n    type: %s\n    color1: %s\n    color2: %s\n" % (v["type"], _v(v["c1"]), _v(v["c2"]))
            if v.get("transform"):
                s += "    transform:\n" + _xf(v["transform"], 6)
        else:
            s += "  %s: %r\n" % (key, float(v))
    return s + "\n"


def _xf(ops, indent=4):
    s = ""
    for k, v in ops:
        s += " " * indent + "- %s: %s\n" % (k, _v(v) if isinstance(v, (list, tuple)) else repr(float(v)))
    return s


def _shape(kind, material=None, transform=None, group=None, indent=0):
    p = " " * indent
    s = p + "- add: %s\n" % kind
    if material:
        s += p + "  material: %s\n" % material
    if group:
        s += p + "  group: %s\n" % group
    if transform:
        s += p + "  transform:\n" + _xf(transform, indent + 4)
    return s


def s1_example(width=400, height=200):
    """The layout of the reference's example scene (test.yaml:5-106): checker floor, two striped
    walls, glass sphere, cube, small sphere.  Restated from SURVEY.md §9.7 (values, not text)."""
    s = _camera(width, height, (0, 1.5, -5), (0, 1, 0)) + _light((-10, 10, -10))
    s += _material("floor", pattern=dict(type="checkers", c1=(1, 1, 1), c2=(0.1, 0.1, 0.1),
                                         transform=[("translate", (0, 0.5, 0))]), specular=0, reflective=0.5)
    s += _material("wall", pattern=dict(type="stripes", c1=(1, 0.9, 0.9), c2=(0.8, 0.4, 0.4)), specular=0)
    s += _material("ball", diffuse=0.7, specular=0.3)
    s += "- define: material\n  name: right cube\n  extends: ball\n  color: [0.5, 1, 0.1]\n\n"
    s += "- define: material\n  name: left sphere\n  extends: ball\n  color: [1, 0.8, 0.1]\n\n"
    s += ("- define: material\n  name: middle sphere\n  extends: ball\n  pattern:\n    type: stripes\n"
          "    color1: [0.06, 0.6, 0.3]\n    color2: [0.02, 0.2, 0.1]\n    transform:\n"
          "      - scale: [0.2, 0.2, 0.2]\n      - rotate-y: pi / 8\n  ambient: 0.1\n  diffuse: 0.3\n"
          "  specular: 1\n  shininess: 300\n  reflective: 0.9\n  transparency: 0.9\n  refractive-index: 1.52\n\n")
    s += "- add: plane\n  material: floor\n\n"
    s += "- add: plane\n  material: wall\n  transform:\n    - rotate-x: pi / 2\n    - rotate-y: -pi / 4\n    - translate: [0, 0, 5]\n\n"
    s += "- add: plane\n  material: wall\n  transform:\n    - rotate-x: pi / 2\n    - rotate-y: pi / 4\n    - translate: [0, 0, 5]\n\n"
    s += "- add: sphere\n  material: middle sphere\n  transform:\n    - translate: [-0.5, 1, 0.5]\n\n"
    s += ("- add: cube\n  material: right cube\n  transform:\n    - scale: [0.5, 0.5, 0.5]\n    - rotate-y: pi / 4\n"
          "    - translate: [1.5, 0.5, -0.5]\n\n")
    s += "- add: sphere\n  material: left sphere\n  transform:\n    - scale: [0.33, 0.33, 0.33]\n    - translate: [-1.5, 0.33, -0.75]\n"
    return s


def s2_spheres_planes(width=1920, height=1080, seed=SEED):
    """config 2: 256 diffuse spheres on a jittered 16x16 grid + floor + back wall, 1 light, shadows only."""
    r = PCG32(seed, 2)
    s = _camera(width, height, (0, 3, -12), (0, 1, 0)) + _light((-10, 12, -10))
    s += _material("floor", color=(0.9, 0.9, 0.9), specular=0)
    s += _shape("plane", "floor") + "\n"
    s += _shape("plane", "floor", [("rotate-x", math.pi / 2), ("translate", (0, 0, 8))]) + "\n"
    for i in range(16):
        for j in range(16):
            rad = r.uniform(0.2, 0.45)
            x = -8 + (i + 0.5) + r.uniform(-0.25, 0.25)
            z = -4 + (j + 0.5) * 10 / 16 + r.uniform(-0.1, 0.1)
            name = "m%d_%d" % (i, j)
            s += _material(name, color=(r.uniform(0.1, 1), r.uniform(0.1, 1), r.uniform(0.1, 1)))
            s += _shape("sphere", name, [("scale", (rad, rad, rad)), ("translate", (x, rad, z))]) + "\n"
    return s


def _s3_materials():
    s = ""
    cols = [(0.9, 0.3, 0.2), (0.2, 0.7, 0.9), (0.95, 0.85, 0.3)]
    for k, c in enumerate(cols):
        s += _material("u%d" % k, color=c)
    for k in range(3):
        s += _material("c%d" % k, pattern=dict(type="checkers", c1=cols[k], c2=(0.1, 0.1, 0.1),
                                               transform=[("scale", (0.25, 0.25, 0.25)), ("translate", (0.013, 0.017, 0.019))]))
    for k in range(2):
        s += _material("s%d" % k, pattern=dict(type="stripes", c1=cols[k], c2=(1, 1, 1),
                                               transform=[("scale", (0.2, 0.2, 0.2)), ("rotate-y", math.pi / 8)]))
    return s, ["u0", "u1", "u2", "c0", "c1", "c2", "s0", "s1"]


def s3_nested_groups(width=3840, height=2160, seed=SEED, per_leaf=20, levels=3):
    """config 3: 8^levels leaf groups x per_leaf spheres (10 240 by default) in nested groups
    (octants, each a pure translate relative to its parent), Phong + checker/stripe patterns."""
    r = PCG32(seed, 3)
    s = _camera(width, height, (0, 6, -16), (0, 4, 0)) + _light((-10, 14, -10))
    ms, names = _s3_materials()
    s += ms
    s += _material("floor", pattern=dict(type="checkers", c1=(0.8, 0.8, 0.8), c2=(0.3, 0.3, 0.3),
                                         transform=[("translate", (0, 0.5, 0))]), specular=0)
    s += _shape("plane", "floor") + "\n"
    s += "- define: group\n  name: root\n  transform:\n    - translate: [0, 4.2, 0]\n\n"
    count = [0]

    def rec(parent, level, edge):
        # 8 children of half the edge, centred at +-edge/4 relative to the parent
        for o in range(8):
            off = [(edge / 4) * (1 if (o >> a) & 1 else -1) for a in range(3)]
            name = "%s_%d" % (parent, o)
            nonlocal_s.append("- define: group\n  name: %s\n  parent: %s\n  transform:\n    - translate: %s\n\n"
                              % (name, parent, _v(off)))
            if level + 1 < levels:
                rec(name, level + 1, edge / 2)
            else:
                half = edge / 4
                for _ in range(per_leaf):
                    rad = r.uniform(0.05, 0.12)
                    c = [r.uniform(-(half - rad), half - rad) for _ in range(3)]
                    m = names[count[0] % len(names)]
                    count[0] += 1
                    nonlocal_s.append(_shape("sphere", m, [("scale", (rad, rad, rad)), ("translate", c)], group=name) + "\n")

    nonlocal_s = []
    rec("root", 0, 8.0)
    return s + "".join(nonlocal_s)


def s4_glass_mirror(width=3840, height=2160, seed=SEED):
    """config 4: reflective checker floor, mirror wall, 8 mirror + 8 glass + 4 glass-in-glass pairs
    + 4 diffuse spheres, 2 glass cubes; reflection + refraction to the reference's depth 5."""
    r = PCG32(seed, 4)
    s = _camera(width, height, (0, 3.2, -11), (0, 1.2, 0)) + _light((-8, 12, -10))
    s += _material("floor", pattern=dict(type="checkers", c1=(0.9, 0.9, 0.9), c2=(0.15, 0.15, 0.15),
                                         transform=[("translate", (0.011, 0.5, 0.007))]), specular=0, reflective=0.5)
    s += _material("mirrorwall", color=(0.1, 0.1, 0.12), diffuse=0.2, reflective=0.9)
    s += _material("mirror", color=(0.9, 0.9, 0.95), diffuse=0.1, specular=1, shininess=300, reflective=0.9)
    s += _material("glass", color=(0.05, 0.1, 0.05), ambient=0.1, diffuse=0.3, specular=1, shininess=300,
                   reflective=0.9, transparency=0.9, refractive_index=1.52)
    s += _material("glass15", color=(0.1, 0.05, 0.05), diffuse=0.2, specular=1, shininess=300,
                   reflective=0.9, transparency=0.9, refractive_index=1.5)
    s += _material("bubble", color=(1, 1, 1), ambient=0, diffuse=0, specular=0.5, reflective=0.5,
                   transparency=1, refractive_index=1.0)
    s += _shape("plane", "floor") + "\n"
    s += _shape("plane", "mirrorwall", [("rotate-x", math.pi / 2), ("translate", (0, 0, 9))]) + "\n"
    kinds = ["mirror"] * 8 + ["glass"] * 8 + ["pair"] * 4 + ["diffuse"] * 4
    k = 0
    for i in range(6):
        for j in range(4):
            kind = kinds[k]; k += 1
            rad = r.uniform(0.45, 0.8)
            x = -7.5 + i * 3.0 + r.uniform(-0.5, 0.5)
            z = -3.0 + j * 2.8 + r.uniform(-0.4, 0.4)
            if kind == "diffuse":
                name = "d%d" % k
                s += _material(name, color=(r.uniform(0.2, 1), r.uniform(0.2, 1), r.uniform(0.2, 1)))
                s += _shape("sphere", name, [("scale", (rad, rad, rad)), ("translate", (x, rad, z))]) + "\n"
            elif kind == "pair":
                s += _shape("sphere", "glass15", [("scale", (rad, rad, rad)), ("translate", (x, rad, z))]) + "\n"
                ri = rad * 0.5
                s += _shape("sphere", "bubble", [("scale", (ri, ri, ri)), ("translate", (x, rad, z))]) + "\n"
            else:
                s += _shape("sphere", kind, [("scale", (rad, rad, rad)), ("translate", (x, rad, z))]) + "\n"
    for sx in (-2.2, 2.4):
        s += _shape("cube", "glass", [("scale", (0.6, 0.6, 0.6)), ("rotate-y", 0.6), ("translate", (sx, 0.6001, -5.0))]) + "\n"
    return s


def s5_csg(width=7680, height=4320, seed=SEED):
    """config 5: 6x4 CSG objects (4 kinds incl. a nested one) + 24 patterned cubes + reflective
    checker floor, 2 lights."""
    r = PCG32(seed, 5)
    s = _camera(width, height, (0, 7, -15), (0, 1, 0)) + _light((-10, 12, -10), (0.7, 0.7, 0.7)) + _light((8, 10, -6), (0.4, 0.4, 0.45))
    s += _material("floor", pattern=dict(type="checkers", c1=(0.85, 0.85, 0.85), c2=(0.2, 0.2, 0.25),
                                         transform=[("translate", (0.009, 0.5, 0.013))]), specular=0, reflective=0.3)
    s += _shape("plane", "floor") + "\n"
    pats = ["stripes", "checkers", "rings", "gradient"]
    for k, p in enumerate(pats):
        s += _material("p%d" % k, pattern=dict(type=p, c1=(0.9, 0.4 + 0.1 * k, 0.2), c2=(0.2, 0.3, 0.9 - 0.1 * k),
                                               transform=[("scale", (0.31, 0.31, 0.31)), ("rotate-z", 0.3 + 0.2 * k),
                                                          ("translate", (0.017, 0.023, 0.011))]))
    s += _material("red", color=(0.9, 0.2, 0.2))
    s += _material("yellow", color=(0.95, 0.8, 0.2))
    s += _material("blue", color=(0.2, 0.3, 0.9), reflective=0.2)

    def csg(kind, xf):
        t = "  transform:\n" + _xf(xf, 4)
        if kind == 0:
            return ("- add: csg\n  operator: difference\n" + t + "  operands:\n" + _shape("cube", "yellow", indent=4)
                    + _shape("sphere", "red", [("scale", (1.35, 1.35, 1.35))], indent=4))
        if kind == 1:
            return ("- add: csg\n  operator: intersection\n" + t + "  operands:\n"
                    + _shape("sphere", "blue", [("scale", (1.3, 1.3, 1.3))], indent=4) + _shape("cube", "p0", indent=4))
        if kind == 2:
            return ("- add: csg\n  operator: union\n" + t + "  operands:\n" + _shape("sphere", "red", indent=4)
                    + _shape("sphere", "p1", [("translate", (0.5, 0.3, 0.1))], indent=4))
        inner1 = ("    - add: csg\n      operator: intersection\n      operands:\n" + _shape("cube", "yellow", indent=8)
                  + _shape("sphere", "blue", [("scale", (1.35, 1.35, 1.35))], indent=8))
        inner2 = ("    - add: csg\n      operator: union\n      operands:\n"
                  + _shape("cube", "red", [("scale", (0.5, 2, 0.5))], indent=8)
                  + _shape("cube", "red", [("scale", (2, 0.5, 0.5))], indent=8))
        return "- add: csg\n  operator: difference\n" + t + "  operands:\n" + inner1 + inner2

    k = 0
    for i in range(6):
        for j in range(4):
            ang = r.uniform(0.05, math.pi / 2 - 0.05)
            x = -10 + i * 4.0 + r.uniform(-0.3, 0.3)
            z = -2 + j * 4.2 + r.uniform(-0.3, 0.3)
            s += csg(k % 4, [("rotate-y", ang), ("translate", (x, 1.0007, z))]) + "\n"
            k += 1
    for i in range(24):
        sc = r.uniform(0.3, 0.55)
        x = -11 + (i % 12) * 2.0 + r.uniform(-0.2, 0.2)
        z = -6.5 + (i // 12) * 19.5 + r.uniform(-0.3, 0.3)
        s += _shape("cube", "p%d" % (i % 4), [("scale", (sc, sc, sc)), ("rotate-y", r.uniform(0.1, 1.4)),
                                              ("translate", (x, sc + 0.0003, z))]) + "\n"
    return s


SCENES = {"S1": s1_example, "S2": s2_spheres_planes, "S3": s3_nested_groups, "S4": s4_glass_mirror, "S5": s5_csg}


def make(name, **kw):
    return SCENES[name](**kw)


if __name__ == "__main__":
    import sys
    sys.stdout.write(make(sys.argv[1]))
