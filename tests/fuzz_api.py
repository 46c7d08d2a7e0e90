"""Seeded random scene graphs built through the WorldBuilder API (world.rs:15-79), beyond what the scene
DSL can express: groups as CSG operands, CSG inside groups, empty groups, lists of more than 32 children,
deep nesting, planes anywhere.  `plan(rng)` draws a builder-independent plan, `build(api, plan)` issues the
same calls against any rtapi.HostApi (the product's host mirror or the oracle's builder)."""
import math

import rtapi


def _ops(rng, spread=3.0, smin=0.4, smax=1.5):
    ops = []
    for k in rng.permutation(["scale", "rotate-x", "rotate-y", "rotate-z", "translate"]):
        if rng.random() < 0.5:
            continue
        if k == "scale":
            s = [float(rng.uniform(smin, smax)) * (1 if rng.random() < 0.9 else -1) for _ in range(3)]
            if rng.random() < 0.5:
                s = [s[0]] * 3
            ops.append(("scale", s))
        elif k == "translate":
            ops.append(("translate", [float(rng.uniform(-spread, spread)), float(rng.uniform(-0.5, spread)), float(rng.uniform(-spread, spread))]))
        else:
            ops.append((k, float(rng.uniform(-math.pi, math.pi))))
    return ops


def _material(rng):
    m = dict(ambient=float(rng.uniform(0.05, 0.3)), diffuse=float(rng.uniform(0.2, 0.9)), specular=float(rng.uniform(0, 1)),
             shininess=float(rng.choice([10.0, 50.0, 200.0, 33.3])))
    kind = int(rng.integers(0, 6))
    col = [float(x) for x in rng.uniform(0.05, 1, 3)]
    if kind <= 1:
        m["color"] = col
    else:
        m.update(pattern_kind=[rtapi.PAT_STRIPES, rtapi.PAT_RINGS, rtapi.PAT_CHECKERS, rtapi.PAT_GRADIENT][kind - 2],
                 color_a=col, color_b=[float(x) for x in rng.uniform(0, 1, 3)], pattern_ops=_ops(rng, 1.0, 0.15, 0.8))
    look = int(rng.integers(0, 5))
    if look == 2:
        m["reflective"] = float(rng.uniform(0.2, 1))
    elif look == 3:
        m.update(reflective=float(rng.uniform(0.1, 1)), transparency=float(rng.uniform(0.3, 1)), refractive_index=float(rng.uniform(1, 2.5)))
    elif look == 4:
        m.update(transparency=float(rng.uniform(0.2, 0.9)), refractive_index=float(rng.uniform(1, 1.8)))
    return m


def _shape(rng, n_mats, small=False, planes=True):
    kinds = [rtapi.SPHERE, rtapi.SPHERE, rtapi.CUBE] + ([rtapi.PLANE] if planes and rng.random() < 0.5 else [])
    kind = kinds[int(rng.integers(0, len(kinds)))]
    ops = _ops(rng, 3.0, 0.1, 0.3) if small else _ops(rng)
    if kind == rtapi.PLANE:
        ops = [("rotate-x", float(rng.uniform(0.3, 2.8))), ("translate", [0.0, float(rng.uniform(0.5, 3)), float(rng.uniform(2, 9))])]
    return ("shape", kind, int(rng.integers(0, n_mats)), ops)


def _csg(rng, n_mats, budget, depth, max_depth=3, p_nest=0.2):
    """budget = [leaf shapes left, containers left] under the CSG root (limits of the fast kernels: 8 / 8; plan(big_csg=True) asks
    for roots beyond them, which render with the general CSG kernels)."""
    def operand():
        r = rng.random()
        if r < p_nest and budget[0] >= 2 and budget[1] >= 1 and depth < max_depth:
            budget[1] -= 1
            return _csg(rng, n_mats, budget, depth + 1, max_depth, p_nest)
        if r < 0.4 and budget[1] >= 1:
            budget[1] -= 1
            kids = []
            for _ in range(int(rng.integers(0, 3))):
                if budget[0] >= 1:
                    budget[0] -= 1
                    kids.append(_shape(rng, n_mats, planes=False))
            return ("group", _ops(rng, 1.0, 0.7, 1.2), kids)
        budget[0] -= 1
        return _shape(rng, n_mats, planes=False)
    op = [rtapi.UNION, rtapi.INTERSECTION, rtapi.DIFFERENCE][int(rng.integers(0, 3))]
    left = operand()
    right = operand()
    return ("csg", op, _ops(rng), left, right)


def _group(rng, n_mats, depth):
    kids = []
    n = int(rng.integers(0, 6)) if rng.random() < 0.9 else int(rng.integers(33, 70))
    for _ in range(n):
        r = rng.random()
        if n > 8:
            kids.append(_shape(rng, n_mats, small=True, planes=False))
        elif r < 0.2 and depth < 6:
            kids.append(_group(rng, n_mats, depth + 1))
        elif r < 0.35:
            kids.append(_csg(rng, n_mats, [6, 6], 0))
        else:
            kids.append(_shape(rng, n_mats))
    return ("group", _ops(rng, 2.0, 0.6, 1.4), kids)


def plan(rng, width=64, height=48, big_csg=False):
    n_mats = int(rng.integers(2, 7))
    roots = []
    if rng.random() < 0.8:
        roots.append(("shape", rtapi.PLANE, 0, []))
    for _ in range(int(rng.integers(1, 9))):
        r = rng.random()
        roots.append(_group(rng, n_mats, 0) if r < 0.35 else _csg(rng, n_mats, [7, 7], 0) if r < 0.55 else _shape(rng, n_mats))
    if big_csg:
        # CSG roots beyond the fast kernels' bounds: up to 40 leaf shapes, containers nested up to 14 deep under a root
        for _ in range(int(rng.integers(1, 3))):
            roots.append(_csg(rng, n_mats, [int(rng.integers(9, 41)), 40], 0, max_depth=14, p_nest=0.75))
    a, rad = float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(6, 14))
    return dict(materials=[_material(rng) for _ in range(n_mats)],
                lights=[([float(rng.uniform(-10, 10)), float(rng.uniform(4, 12)), float(rng.uniform(-10, -2))], [float(rng.uniform(0.3, 1))] * 3)
                        for _ in range(int(rng.integers(0, 4)))],
                roots=roots, width=width, height=height, fov=float(rng.uniform(0.6, 1.4)),
                cam_from=[rad * math.sin(a), float(rng.uniform(1, 7)), -rad * math.cos(a)], cam_to=[0.0, float(rng.uniform(0, 2)), 0.0])


def build(api, p):
    w = api.world_new()
    mats = []
    for m in p["materials"]:
        kw = {k: v for k, v in m.items() if k != "pattern_ops"}
        if "pattern_ops" in m:
            kw["pattern_transform"] = api.compose(m["pattern_ops"])
        mats.append(api.material(**kw))

    def add(node):
        if node[0] == "shape":
            return rtapi.shape(w.add_shape(node[1], mats[node[2]], api.compose(node[3])))
        if node[0] == "group":
            g = w.add_group(api.compose(node[1]))
            for kid in node[2]:
                w.set_group_of(add(kid), g)
            return rtapi.group(g)
        left = add(node[3]); right = add(node[4])
        return rtapi.csg(w.add_csg(node[1], left, right, api.compose(node[2])))

    for node in p["roots"]:
        w.add_object_to_scene(add(node))
    for pos, inten in p["lights"]:
        w.add_light(pos, inten)
    w.build()
    cam = api.camera(p["width"], p["height"], p["fov"], api.view_transform(p["cam_from"], p["cam_to"], [0, 1, 0]))
    return w, cam
