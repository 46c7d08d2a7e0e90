"""API-built scenes that the YAML DSL cannot express (world.rs WorldBuilder allows them): CSGs inside
groups, groups as CSG operands, empty groups, groups of more than 32 children (list chunking), planes
inside groups, deep nesting, non-uniform scales, two lights, glass.  `build(api)` issues the same
builder calls against any rtapi.HostApi (the product's host mirror or the oracle's builder)."""
import math

import rtapi
import scenes


def build(api, width=160, height=96, seed=7):
    r = scenes.PCG32(seed, 9)
    w = api.world_new()
    T, S, RY, RX = api.translation, api.scaling, api.rotation_y, api.rotation_x
    mul = api.mat_mul

    def xf(*ms):                       # applied left to right (like the DSL's transform lists)
        t = api.identity()
        for m in ms:
            t = mul(m, t)
        return t

    def mat(**kw):
        return api.material(**kw)

    floor = mat(pattern_kind=rtapi.PAT_CHECKERS, color_a=(0.9, 0.9, 0.9), color_b=(0.2, 0.2, 0.3), specular=0.0, reflective=0.3,
                pattern_transform=api.translation(0.013, 0.5, 0.007))
    glass = mat(color=(0.05, 0.1, 0.05), diffuse=0.3, specular=1.0, shininess=300.0, reflective=0.9, transparency=0.9, refractive_index=1.5)
    mirror = mat(color=(0.9, 0.9, 0.95), diffuse=0.1, reflective=0.8)

    def rnd_mat():
        k = r.next() % 5
        col = (r.uniform(0.2, 1), r.uniform(0.2, 1), r.uniform(0.2, 1))
        if k == 0:
            return mat(pattern_kind=rtapi.PAT_STRIPES, color_a=col, color_b=(0.1, 0.1, 0.1),
                       pattern_transform=mul(api.scaling(0.23, 0.23, 0.23), api.rotation_z(0.4)))
        if k == 1:
            return mat(pattern_kind=rtapi.PAT_RINGS, color_a=col, color_b=(0.8, 0.8, 0.1), pattern_transform=api.scaling(0.17, 0.17, 0.17))
        if k == 2:
            return mat(pattern_kind=rtapi.PAT_GRADIENT, color_a=col, color_b=(0.1, 0.2, 0.9))
        return mat(color=col, shininess=float(10 + r.next() % 200))

    w.add_object_to_scene(rtapi.shape(w.add_shape(rtapi.PLANE, floor)))
    # a group holding a tilted plane (plane inside a group -> infinite bounds) and a sphere
    g0 = w.add_group(xf(T(0, 0, 12)))
    w.add_object_to_scene(rtapi.group(g0))
    w.set_group_of(rtapi.shape(w.add_shape(rtapi.PLANE, mirror, xf(RX(math.pi / 2)))), g0)
    w.set_group_of(rtapi.shape(w.add_shape(rtapi.SPHERE, rnd_mat(), xf(T(-6, 1, -2)))), g0)
    # an empty group, and a group whose only child is an empty group
    w.add_object_to_scene(rtapi.group(w.add_group(xf(T(1, 1, 1)))))
    ge = w.add_group(xf(S(2, 2, 2)))
    w.add_object_to_scene(rtapi.group(ge))
    w.set_group_of(rtapi.group(w.add_group()), ge)
    # a wide group: 70 small spheres and cubes (three list chunks), non-uniform scales and rotations
    gw = w.add_group(xf(RY(0.3), T(-1, 0, 2)))
    w.add_object_to_scene(rtapi.group(gw))
    for i in range(70):
        x, z = -5 + (i % 10) * 1.1 + r.uniform(-0.2, 0.2), -2 + (i // 10) * 1.1 + r.uniform(-0.2, 0.2)
        sx, sy, sz = r.uniform(0.15, 0.4), r.uniform(0.15, 0.4), r.uniform(0.15, 0.4)
        kind = rtapi.CUBE if i % 3 == 0 else rtapi.SPHERE
        m = glass if i % 11 == 0 else rnd_mat()
        w.set_group_of(rtapi.shape(w.add_shape(kind, m, xf(S(sx, sy, sz), RY(r.uniform(0, 1.5)), T(x, sy + 0.001, z)))), gw)
    # nesting 5 deep with a CSG at the bottom (CSG inside groups)
    parent = w.add_group(xf(T(4.5, 0, -1)))
    w.add_object_to_scene(rtapi.group(parent))
    for d in range(4):
        g = w.add_group(xf(RY(0.2), S(0.9, 0.9, 0.9), T(0.1, 0.15, 0.1)))
        w.set_group_of(rtapi.group(g), parent)
        w.set_group_of(rtapi.shape(w.add_shape(rtapi.SPHERE, rnd_mat(), xf(S(0.2, 0.2, 0.2), T(-0.9, 0.3 + 0.3 * d, -0.5)))), g)
        parent = g
    c_in = w.add_csg(rtapi.DIFFERENCE, rtapi.shape(w.add_shape(rtapi.CUBE, rnd_mat())),
                     rtapi.shape(w.add_shape(rtapi.SPHERE, mirror, xf(S(1.3, 1.3, 1.3)))), xf(S(0.6, 0.6, 0.6), T(0, 0.7, 0)))
    w.set_group_of(rtapi.csg(c_in), parent)
    # a CSG whose operands are a group (two spheres) and a nested CSG; top level
    go = w.add_group(xf(T(0.2, 0, 0)))
    w.set_group_of(rtapi.shape(w.add_shape(rtapi.SPHERE, rnd_mat(), xf(T(-0.4, 0, 0)))), go)
    w.set_group_of(rtapi.shape(w.add_shape(rtapi.SPHERE, glass, xf(S(0.8, 0.8, 0.8), T(0.5, 0.1, 0)))), go)
    inner = w.add_csg(rtapi.INTERSECTION, rtapi.shape(w.add_shape(rtapi.CUBE, rnd_mat(), xf(S(0.9, 0.9, 0.9)))),
                      rtapi.shape(w.add_shape(rtapi.SPHERE, rnd_mat(), xf(S(1.2, 1.2, 1.2)))))
    top = w.add_csg(rtapi.UNION, rtapi.group(go), rtapi.csg(inner), xf(RY(0.5), T(-4.5, 1.2, -2.5)))
    w.add_object_to_scene(rtapi.csg(top))
    # 40 more top-level shapes so the world's own list has several chunks with containers in between
    for i in range(40):
        sc = r.uniform(0.1, 0.3)
        w.add_object_to_scene(rtapi.shape(w.add_shape(rtapi.SPHERE if i % 2 else rtapi.CUBE, mirror if i % 7 == 0 else rnd_mat(),
                                                      xf(S(sc, sc, sc), T(-7 + (i % 20) * 0.7, sc, -5.5 + (i // 20) * 0.8)))))
        if i == 17:
            gm = w.add_group(xf(T(2.5, 0.0, -3.5)))
            w.add_object_to_scene(rtapi.group(gm))
            w.set_group_of(rtapi.shape(w.add_shape(rtapi.CUBE, glass, xf(S(0.5, 0.5, 0.5), RY(0.7), T(0, 0.5003, 0)))), gm)
    w.add_light((-8, 12, -10), (0.7, 0.7, 0.7))
    w.add_light((9, 7, -8), (0.4, 0.4, 0.45))
    w.build()
    cam = api.camera(width, height, math.pi / 3, api.view_transform((0, 5.5, -11), (0, 0.8, 0), (0, 1, 0)))
    return w, cam
