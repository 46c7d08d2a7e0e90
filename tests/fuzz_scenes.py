"""Seeded generator of well-formed scene files that roam the scene DSL (yaml.rs): nested groups with
transforms, CSG trees up to two levels, all pattern kinds with pattern transforms, mirror / glass / diffuse
materials, one or two lights, a camera looking at the origin.  Values stay where the reference renders without
panicking (no zero scales, no NaN expressions); used by the randomized GPU-vs-oracle parity test."""
import math


def _v(rng, lo, hi):
    return float(rng.uniform(lo, hi))


def _tuple(vals):
    return "[" + ", ".join("%.17g" % v for v in vals) + "]"


def _transform(rng, ind, spread=3.0, smin=0.4, smax=1.6):
    out = []
    for k in rng.permutation(["scale", "rotate-x", "rotate-y", "rotate-z", "translate"]):
        if rng.random() < 0.45:
            continue
        if k == "scale":
            s = [_v(rng, smin, smax) * (1 if rng.random() < 0.9 else -1) for _ in range(3)]
            if rng.random() < 0.5:
                s = [s[0]] * 3
            out.append("%s- scale: %s" % (ind, _tuple(s)))
        elif k == "translate":
            out.append("%s- translate: %s" % (ind, _tuple([_v(rng, -spread, spread), _v(rng, -1, spread), _v(rng, -spread, spread)])))
        else:
            out.append("%s- %s: %.17g" % (ind, k, _v(rng, -math.pi, math.pi)))
    return out


def _material(rng, name):
    lines = ["- define: material", "  name: " + name]
    kind = rng.choice(["plain", "plain", "stripes", "rings", "checkers", "gradient"])
    if kind == "plain":
        lines.append("  color: " + _tuple([_v(rng, 0.05, 1) for _ in range(3)]))
    else:
        lines += ["  pattern:", "    type: " + kind,
                  "    color1: " + _tuple([_v(rng, 0, 1) for _ in range(3)]),
                  "    color2: " + _tuple([_v(rng, 0, 1) for _ in range(3)])]
        tr = _transform(rng, "      ", spread=1.0, smin=0.15, smax=0.8)
        if tr:
            lines += ["    transform:"] + tr
    look = rng.choice(["diffuse", "diffuse", "mirror", "glass", "tinted"])
    lines += ["  ambient: %.17g" % _v(rng, 0.05, 0.3), "  diffuse: %.17g" % _v(rng, 0.2, 0.9),
              "  specular: %.17g" % _v(rng, 0, 1), "  shininess: %.17g" % rng.choice([10.0, 50.0, 200.0, _v(rng, 5, 300)])]
    if look == "mirror":
        lines.append("  reflective: %.17g" % _v(rng, 0.2, 1))
    elif look == "glass":
        lines += ["  reflective: %.17g" % _v(rng, 0.1, 1), "  transparency: %.17g" % _v(rng, 0.3, 1),
                  "  refractive-index: %.17g" % _v(rng, 1.0, 2.5)]
    elif look == "tinted":
        lines += ["  transparency: %.17g" % _v(rng, 0.2, 0.9), "  refractive-index: %.17g" % _v(rng, 1.0, 1.8)]
    return lines


def _shape(rng, mats, group=None, ind="", operand=False):
    kind = rng.choice(["sphere", "sphere", "cube", "plane"] if not operand else ["sphere", "cube", "cube"])
    lines = ["add: " + kind, "material: " + mats[int(rng.integers(0, len(mats)))]]
    tr = _transform(rng, "  ")
    if kind == "plane":
        # (a second plane in the floor's own plane would tie with it in every pixel: tilt and move it)
        tr = ["  - rotate-x: %.17g" % _v(rng, 0.3, 2.8), "  - rotate-y: %.17g" % _v(rng, -3, 3),
              "  - translate: %s" % _tuple([_v(rng, -3, 3), _v(rng, 0.5, 3), _v(rng, 2, 9)])]
    if tr:
        lines += ["transform:"] + tr
    if group:
        lines.append("group: " + group)
    return [ind + l for l in lines]


def _csg(rng, mats, depth=0):
    lines = ["add: csg", "operator: " + rng.choice(["union", "intersection", "difference"])]
    tr = _transform(rng, "  ")
    if tr:
        lines += ["transform:"] + tr
    lines.append("operands:")
    for _ in range(2):
        sub = _csg(rng, mats, depth + 1) if depth < 1 and rng.random() < 0.35 else _shape(rng, mats, operand=True)
        lines += ["  - " + sub[0]] + ["    " + l for l in sub[1:]]
    return lines


def make(rng, width=64, height=48):
    out = []

    def item(lines):
        out.append("- " + lines[0]); out.extend("  " + l for l in lines[1:])

    for _ in range(int(rng.integers(1, 3))):
        item(["add: light", "at: " + _tuple([_v(rng, -10, 10), _v(rng, 4, 12), _v(rng, -10, -2)]),
              "intensity: " + _tuple([_v(rng, 0.4, 1)] * 3)])
    mats = []
    for i in range(int(rng.integers(2, 7))):
        out += _material(rng, "m%d" % i); mats.append("m%d" % i)
    groups = []
    for i in range(int(rng.integers(0, 5))):
        g = ["define: group", "name: g%d" % i]
        if groups and rng.random() < 0.6:
            g.append("parent: " + groups[int(rng.integers(0, len(groups)))])
        tr = _transform(rng, "  ", spread=2.0, smin=0.6, smax=1.4)
        if tr:
            g += ["transform:"] + tr
        item(g); groups.append("g%d" % i)
    if rng.random() < 0.8:
        item(["add: plane", "material: " + mats[0]])                     # a floor
    for _ in range(int(rng.integers(2, 12))):
        if rng.random() < 0.25:
            item(_csg(rng, mats))
        else:
            grp = groups[int(rng.integers(0, len(groups)))] if groups and rng.random() < 0.5 else None
            item(_shape(rng, mats, group=grp))
    a = _v(rng, 0, 2 * math.pi)
    r = _v(rng, 6, 14)
    item(["add: camera", "width: %d" % width, "height: %d" % height, "field-of-view: %.17g" % _v(rng, 0.6, 1.4),
          "from: " + _tuple([r * math.sin(a), _v(rng, 1, 7), -r * math.cos(a)]), "to: " + _tuple([0, _v(rng, 0, 2), 0]),
          "up: [0, 1, 0]"])
    return "\n".join(out) + "\n"
