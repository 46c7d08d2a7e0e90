"""Pins the CPU oracle against the reference's own known-answer tests.

Every hot-path #[test] of felgru/ray-tracer is re-asserted
here against oracle/librt_oracle.so with the reference's tolerances
(approx::assert_relative_eq!, default max_relative = f64::EPSILON).  Each test
names the reference test it restates (file:line).
"""
import math

import pytest

import ora
from ora import API as A
from rtapi import (CUBE, DIFFERENCE, INTERSECTION, OBJ_CSG, OBJ_GROUP, OBJ_SHAPE, PAT_CHECKERS,
                   PAT_GRADIENT, PAT_RINGS, PAT_STRIPES, PAT_TEST, PLANE, SPHERE, UNION, csg, d3,
                   group, shape)
from util import assert_rel

PI = math.pi
SQRT2 = math.sqrt(2.0)
WHITE, BLACK = [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]


def default_world(extra=None):
    """world.rs:326-354"""
    w = A.world_new()
    w.add_light([-10, 10, -10], WHITE)
    m = A.material(color=[0.8, 1.0, 0.6], diffuse=0.7, specular=0.2)
    s1 = w.add_shape(SPHERE, m, A.identity())
    s2 = w.add_shape(SPHERE, None, A.scaling(0.5, 0.5, 0.5))
    w.add_object_to_scene(shape(s1)); w.add_object_to_scene(shape(s2))
    if extra:
        extra(w)
    return w.build()


# ------------------------------------------------------------------ geometry.rs:73-129
def test_reflect():
    assert_rel(ora.vec3(ora.LIB.ora_reflect, d3([1, -1, 0]), d3([0, 1, 0])), [1, 1, 0])
    x = 1 / SQRT2
    assert_rel(ora.vec3(ora.LIB.ora_reflect, d3([0, -1, 0]), d3([x, x, 0])), [1, 0, 0])


def test_view_transforms():
    assert_rel(A.view_transform([0, 0, 0], [0, 0, -1], [0, 1, 0]), A.identity())
    assert_rel(A.view_transform([0, 0, 0], [0, 0, 1], [0, 1, 0]), A.scaling(-1, 1, -1))
    assert_rel(A.view_transform([0, 0, 8], [0, 0, 0], [0, 1, 0]), A.translation(0, 0, -8))
    t = A.view_transform([1, 3, 2], [4, -2, 8], [1, 1, 0])
    rows = [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
            [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]]
    for r in range(4):
        for c in range(4):
            assert abs(t[c * 4 + r] - rows[r][c]) <= 1e-4 * max(abs(rows[r][c]), 1e-12) + 1e-5


# ------------------------------------------------------------------ rays.rs:40-66
def test_ray_transform():
    o = ora.vec3(ora.LIB.ora_mul_point, A.translation(3, 4, 5), d3([1, 2, 3]))
    d = ora.vec3(ora.LIB.ora_mul_vector, A.translation(3, 4, 5), d3([0, 1, 0]))
    assert_rel(o, [4, 6, 8]); assert_rel(d, [0, 1, 0])
    o = ora.vec3(ora.LIB.ora_mul_point, A.scaling(2, 3, 4), d3([1, 2, 3]))
    d = ora.vec3(ora.LIB.ora_mul_vector, A.scaling(2, 3, 4), d3([0, 1, 0]))
    assert_rel(o, [2, 6, 12]); assert_rel(d, [0, 3, 0])


# ------------------------------------------------------------------ sphere.rs:48-113
@pytest.mark.parametrize("o,exp", [([0, 0, -5], [4, 6]), ([0, 1, -5], [5, 5]), ([0, 2, -5], []),
                                   ([0, 0, 0], [-1, 1]), ([0, 0, 5], [-6, -4])])
def test_sphere_intersect(o, exp):
    assert_rel(ora.local_intersect(SPHERE, o, [0, 0, 1]), exp)


def test_sphere_normals():
    for p in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1 / math.sqrt(3)] * 3):
        assert_rel(ora.local_normal_at(SPHERE, p), p)


# ------------------------------------------------------------------ plane.rs:41-78
def test_plane():
    for p in ([0, 0, 0], [10, 0, -10], [-5, 0, 150]):
        assert ora.local_normal_at(PLANE, p) == [0, 1, 0]
    assert ora.local_intersect(PLANE, [0, 10, 0], [0, 0, 1]) == []
    assert ora.local_intersect(PLANE, [0, 0, 0], [0, 0, 1]) == []
    assert_rel(ora.local_intersect(PLANE, [0, 1, 0], [0, -1, 0]), [1])
    assert_rel(ora.local_intersect(PLANE, [0, -1, 0], [0, 1, 0]), [1])


# ------------------------------------------------------------------ cube.rs:78-141
@pytest.mark.parametrize("o,d,t1,t2", [
    ([5, 0.5, 0], [-1, 0, 0], 4, 6), ([-5, 0.5, 0], [1, 0, 0], 4, 6), ([0.5, 5, 0], [0, -1, 0], 4, 6),
    ([0.5, -5, 0], [0, 1, 0], 4, 6), ([0.5, 0, 5], [0, 0, -1], 4, 6), ([0.5, 0, -5], [0, 0, 1], 4, 6),
    ([0, 0.5, 0], [0, 0, 1], -1, 1)])
def test_cube_hits(o, d, t1, t2):
    assert_rel(ora.local_intersect(CUBE, o, d), [t1, t2])


@pytest.mark.parametrize("o,d", [
    ([-2, 0, 0], [0.2673, 0.5345, 0.8018]), ([0, -2, 0], [0.8018, 0.2673, 0.5345]),
    ([0, 0, -2], [0.5345, 0.8018, 0.2673]), ([2, 0, 2], [0, 0, -1]), ([0, 2, 2], [0, -1, 0]),
    ([2, 2, 0], [-1, 0, 0])])
def test_cube_misses(o, d):
    assert ora.local_intersect(CUBE, o, d) == []


@pytest.mark.parametrize("p,n", [
    ([1, 0.5, -0.8], [1, 0, 0]), ([-1, -0.2, 0.9], [-1, 0, 0]), ([-0.4, 1, -0.1], [0, 1, 0]),
    ([0.3, -1, -0.7], [0, -1, 0]), ([-0.6, 0.3, 1], [0, 0, 1]), ([0.4, 0.4, -1], [0, 0, -1]),
    ([1, 1, 1], [1, 0, 0]), ([-1, -1, -1], [-1, 0, 0])])
def test_cube_normals(p, n):
    assert_rel(ora.local_normal_at(CUBE, p), n)


# ------------------------------------------------------------------ bounds.rs:128-141
def test_empty_bounds_never_intersect():
    nan = float("nan")
    b = (ora.C.c_double * 6)(*[nan] * 6)
    assert ora.LIB.ora_bounds_intersects(b, d3([0, 0, 0]), d3([0, 0, 1])) == 0


def test_adding_bounds_to_empty_bounds():
    w = A.world_new()
    g = w.add_group(A.identity())
    s = w.add_shape(SPHERE, None, A.identity())
    w.set_group_of(shape(s), g)
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    assert_rel(w.get_bounds(group(g)), [-1, -1, -1, 1, 1, 1])


# ------------------------------------------------------------------ intersections.rs:103-150
def test_hit_selection():
    def hit(ts):
        ts = sorted(ts)
        arr = (ora.C.c_double * len(ts))(*ts)
        i = ora.LIB.ora_hit(arr, len(ts))
        return None if i < 0 else ts[i]
    assert hit([1, 2]) == 1 and hit([2, 1]) == 1
    assert hit([1, -1]) == 1
    assert hit([-1, -2]) is None
    assert hit([5, 7, -3, 2]) == 2


# ------------------------------------------------------------------ shapes/mod.rs:133-153
def test_scaled_and_translated_sphere():
    w = A.world_new()
    s = w.add_shape(SPHERE, None, A.scaling(2, 2, 2))
    xs = ora.object_intersect(w, shape(s), [0, 0, -5], [0, 0, 1])
    assert_rel([x[0] for x in xs], [3, 7])
    s2 = w.add_shape(SPHERE, None, A.translation(5, 0, 0))
    assert ora.object_intersect(w, shape(s2), [0, 0, -5], [0, 0, 1]) == []


# ------------------------------------------------------------------ group.rs:137-207
def test_empty_group():
    w = A.world_new()
    g = w.add_group(A.identity())
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    assert ora.object_intersect(w, group(g), [0, 0, 0], [0, 0, 1]) == []


def test_nonempty_group_order():
    w = A.world_new()
    g = w.add_group(A.identity())
    s0 = w.add_shape(SPHERE, None, A.identity()); w.set_group_of(shape(s0), g)
    s1 = w.add_shape(SPHERE, None, A.translation(0, 0, -3)); w.set_group_of(shape(s1), g)
    s2 = w.add_shape(SPHERE, None, A.translation(5, 0, 0)); w.set_group_of(shape(s2), g)
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    xs = ora.object_intersect(w, group(g), [0, 0, -5], [0, 0, 1])
    assert [x[1] for x in xs] == [s1, s1, s0, s0]


def test_transformed_group():
    w = A.world_new()
    g = w.add_group(A.scaling(2, 2, 2))
    s = w.add_shape(SPHERE, None, A.translation(5, 0, 0)); w.set_group_of(shape(s), g)
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    assert len(ora.object_intersect(w, group(g), [10, 0, -10], [0, 0, 1])) == 2


def test_group_bounding_boxes():
    w = A.world_new()
    g = w.add_group(A.scaling(2, 2, 2))
    s = w.add_shape(SPHERE, None, A.identity()); w.set_group_of(shape(s), g)
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    assert_rel(w.get_bounds(group(g)), [-1, -1, -1, 1, 1, 1])
    s = w.add_shape(SPHERE, None, A.translation(5, 0, 0)); w.set_group_of(shape(s), g)
    ora.LIB.ora_set_bounds_of(w.h, OBJ_GROUP, g, None)
    assert_rel(w.get_bounds(group(g)), [-1, -1, -1, 6, 1, 1])


# ------------------------------------------------------------------ csg.rs:160-260
def test_csg_truth_table():
    T, F = True, False
    table = {
        UNION: [(T, T, T, F), (T, T, F, T), (T, F, T, F), (T, F, F, T), (F, T, T, F), (F, T, F, F),
                (F, F, T, T), (F, F, F, T)],
        INTERSECTION: [(T, T, T, T), (T, T, F, F), (T, F, T, T), (T, F, F, F), (F, T, T, T),
                       (F, T, F, T), (F, F, T, F), (F, F, F, F)],
        DIFFERENCE: [(T, T, T, F), (T, T, F, T), (T, F, T, F), (T, F, F, T), (F, T, T, T),
                     (F, T, F, T), (F, F, T, F), (F, F, F, F)],
    }
    for op, rows in table.items():
        for lhit, inl, inr, res in rows:
            assert bool(ora.LIB.ora_intersection_allowed(op, lhit, inl, inr)) == res


@pytest.mark.parametrize("op,x0,x1", [(UNION, 0, 3), (INTERSECTION, 1, 2), (DIFFERENCE, 0, 1)])
def test_csg_filter(op, x0, x1):
    C = ora.C
    t = (C.c_double * 4)(1, 2, 3, 4); s = (C.c_int32 * 4)(0, 1, 0, 1)
    left = (C.c_int32 * 1)(0)
    ot = (C.c_double * 4)(); os_ = (C.c_int32 * 4)()
    n = ora.LIB.ora_filter_intersections(op, left, 1, t, s, 4, ot, os_)
    assert n == 2
    assert (ot[0], os_[0]) == (t[x0], s[x0]) and (ot[1], os_[1]) == (t[x1], s[x1])


def test_csg_miss_and_hit():
    w = A.world_new()
    l = w.add_shape(SPHERE, None, A.identity()); r = w.add_shape(CUBE, None, A.identity())
    c = w.add_csg(UNION, shape(l), shape(r), A.identity())
    assert ora.LIB.ora_parent_of(w.h, OBJ_SHAPE, l, None) == OBJ_CSG
    # csg.rs:235-243: bounds never set -> NaN bounds -> no intersections
    assert ora.object_intersect(w, csg(c), [0, 2, -5], [0, 0, 1]) == []
    w = A.world_new()
    l = w.add_shape(SPHERE, None, A.identity()); r = w.add_shape(SPHERE, None, A.translation(0, 0, 0.5))
    c = w.add_csg(UNION, shape(l), shape(r), A.identity())
    ora.LIB.ora_set_bounds_of(w.h, OBJ_CSG, c, None)
    xs = ora.object_intersect(w, csg(c), [0, 0, -5], [0, 0, 1])
    assert len(xs) == 2
    assert_rel(xs[0][0], 4.0); assert xs[0][1] == l
    assert_rel(xs[1][0], 6.5); assert xs[1][1] == r


# ------------------------------------------------------------------ material.rs:108-191
def _lighting(eyev, normalv, light_pos, in_shadow=False):
    m = A.material()
    return ora.vec3(ora.LIB.ora_shader_lighting, ora.C.byref(m), d3(WHITE), d3(light_pos), d3(WHITE),
                    d3([0, 0, 0]), d3(eyev), d3(normalv), int(in_shadow))


def test_phong_cases():
    x = 1 / SQRT2
    assert_rel(_lighting([0, 0, -1], [0, 0, -1], [0, 0, -10]), [1.9] * 3)
    assert_rel(_lighting([0, x, -x], [0, 0, -1], [0, 0, -10]), [1.0] * 3)
    assert_rel(_lighting([0, 0, -1], [0, 0, -1], [0, 10, -10]), [0.1 + 0.9 * 1.0 / SQRT2 + 0.0] * 3)
    assert_rel(_lighting([0, -x, -x], [0, 0, -1], [0, 10, -10]), [0.1 + 0.9 * x + 0.9] * 3, 1e-12)
    assert_rel(_lighting([0, 0, -1], [0, 0, -1], [0, 0, 10]), [0.1] * 3)
    assert_rel(_lighting([0, 0, -1], [0, 0, -1], [0, 0, -10], True), [0.1] * 3)


# ------------------------------------------------------------------ patterns.rs:173-253
def _pat(kind, p):
    m = A.material(color_a=WHITE, color_b=BLACK, pattern_kind=kind)
    return ora.vec3(ora.LIB.ora_pattern_color_at, ora.C.byref(m), d3(p))


def test_patterns():
    for p in ([0, 0, 0], [0, 1, 0], [0, 2, 0], [0, 0, 1], [0, 0, 2], [0.9, 0, 0]):
        assert _pat(PAT_STRIPES, p) == WHITE
    assert _pat(PAT_STRIPES, [1, 0, 0]) == BLACK
    assert _pat(PAT_STRIPES, [-0.1, 0, 0]) == BLACK
    assert _pat(PAT_STRIPES, [-1, 0, 0]) == BLACK
    assert _pat(PAT_STRIPES, [-1.1, 0, 0]) == WHITE
    assert _pat(PAT_RINGS, [0, 0, 0]) == WHITE
    assert _pat(PAT_RINGS, [1, 0, 0]) == BLACK
    assert _pat(PAT_RINGS, [0, 0, 1]) == BLACK
    assert _pat(PAT_RINGS, [0.708, 0, 0.708]) == BLACK
    for ax in range(3):
        p = [0, 0, 0]
        assert _pat(PAT_CHECKERS, p) == WHITE
        p[ax] = 0.99; assert _pat(PAT_CHECKERS, p) == WHITE
        p[ax] = 1.01; assert _pat(PAT_CHECKERS, p) == BLACK
    assert_rel(_pat(PAT_GRADIENT, [0, 0, 0]), WHITE)
    assert_rel(_pat(PAT_GRADIENT, [0.25, 0, 0]), [0.75] * 3)
    assert_rel(_pat(PAT_GRADIENT, [0.5, 0, 0]), [0.5] * 3)
    assert_rel(_pat(PAT_GRADIENT, [0.75, 0, 0]), [0.25] * 3)


# ------------------------------------------------------------------ object_store.rs:421-507
def test_object_store_normals():
    w = A.world_new()
    s = w.add_shape(SPHERE, None, A.translation(0, 1, 0))
    assert_rel(ora.vec3(ora.LIB.ora_normal_at, w.h, s, d3([0, 1.70711, -0.70711])), [0, 0.70711, -0.70711], 1e-5)
    t = A.mat_mul(A.scaling(1, 0.5, 1), A.rotation_z(PI / 5))
    s = w.add_shape(SPHERE, None, t)
    x = 1 / SQRT2
    n = ora.vec3(ora.LIB.ora_normal_at, w.h, s, d3([0, x, -x]))
    assert abs(n[0]) < 1e-12
    assert_rel(n[1:], [0.97014, -0.24254], 1e-4)


def test_object_store_stripes():
    w = A.world_new()
    m = A.material(color_a=WHITE, color_b=BLACK, pattern_kind=PAT_STRIPES)
    s = w.add_shape(SPHERE, m, A.scaling(2, 2, 2))
    assert ora.vec3(ora.LIB.ora_object_color_at, w.h, s, d3([1.5, 0, 0])) == WHITE
    m2 = m.copy(); m2.pattern_transform = A.scaling(2, 2, 2)
    s = w.add_shape(SPHERE, m2, A.identity())
    assert ora.vec3(ora.LIB.ora_object_color_at, w.h, s, d3([1.5, 0, 0])) == WHITE
    m3 = m.copy(); m3.pattern_transform = A.translation(0.5, 0, 0)
    s = w.add_shape(SPHERE, m3, A.scaling(2, 2, 2))
    assert ora.vec3(ora.LIB.ora_object_color_at, w.h, s, d3([2.5, 0, 0])) == WHITE


def test_lighting_with_stripe_pattern():
    w = A.world_new()
    m = A.material(color_a=WHITE, color_b=BLACK, pattern_kind=PAT_STRIPES, ambient=1.0, diffuse=0.0, specular=0.0)
    s = w.add_shape(SPHERE, m, A.identity())
    w.add_light([0, 0, -10], WHITE)
    c1 = ora.vec3(ora.LIB.ora_object_lighting, w.h, s, 0, d3([0.9, 0, 0]), d3([0, 0, -1]), d3([0, 0, -1]), 0)
    c2 = ora.vec3(ora.LIB.ora_object_lighting, w.h, s, 0, d3([1.1, 0, 0]), d3([0, 0, -1]), d3([0, 0, -1]), 0)
    assert_rel(c1, WHITE); assert_rel(c2, BLACK)


# ------------------------------------------------------------------ camera.rs:92-153
def test_camera():
    assert_rel(A.camera(200, 125, PI / 2).pixel_size, 0.01)
    assert_rel(A.camera(125, 200, PI / 2).pixel_size, 0.01)
    c = A.camera(201, 101, PI / 2)
    o, d = ora.D3(), ora.D3()
    ora.LIB.ora_ray_for_pixel(ora.C.byref(c), 100, 50, o, d)
    assert_rel(o, [0, 0, 0])
    assert abs(d[0]) < 1e-15 and abs(d[1]) < 1e-15 and d[2] == -1.0
    ora.LIB.ora_ray_for_pixel(ora.C.byref(c), 0, 0, o, d)
    assert_rel(o, [0, 0, 0]); assert_rel(d, [0.66519, 0.33259, -0.66851], 1e-4)
    c.transform = A.mat_mul(A.rotation_y(PI / 4), A.translation(0, -2, 5))
    ora.LIB.ora_ray_for_pixel(ora.C.byref(c), 100, 50, o, d)
    assert_rel(o, [0, 2, -5], 1e-14)
    x = 1 / SQRT2
    assert_rel([d[0], d[2]], [x, -x], 1e-12); assert abs(d[1]) < 1e-15


def test_render_default_world():
    w = default_world()
    c = A.camera(11, 11, PI / 2, A.view_transform([0, 0, -5], [0, 0, 0], [0, 1, 0]))
    img = ora.render(w, c, 5, want_color=True)
    assert_rel(img["color"][5, 5], [0.38066, 0.47583, 0.2855], 1e-4)
    assert img["hit"][5, 5] == 0 and img["hit"][0, 0] == -1


# ------------------------------------------------------------------ world.rs:389-1030
RAY = ([0, 0, -5], [0, 0, 1])
EXPECTED = [0.38066119308103435, 0.47582649135129296, 0.28549589481077575]


def test_intersect_world():
    xs = ora.world_intersect(default_world(), *RAY)
    assert [x[0] for x in xs] == [4.0, 4.5, 5.5, 6.0]


def test_precompute_outside_and_inside():
    def extra(w):
        s = w.add_shape(SPHERE, A.material(color=[1, 0, 0]), A.identity()); w.add_object_to_scene(shape(s))
    w = default_world(extra)
    c = ora.prepare(w, 0, *RAY, [(4.0, 2)])
    assert c.t == 4.0 and c.shape_index == 2 and c.inside == 0
    assert_rel(c.point, [0, 0, -1]); assert_rel(c.eyev, [0, 0, -1]); assert_rel(c.normalv, [0, 0, -1])
    c = ora.prepare(w, 0, [0, 0, 0], [0, 0, 1], [(1.0, 2)])
    assert c.inside == 1
    assert_rel(c.point, [0, 0, 1]); assert_rel(c.eyev, [0, 0, -1]); assert_rel(c.normalv, [0, 0, -1])


def test_shading_an_intersection():
    w = default_world()
    c = ora.prepare(w, 0, *RAY, [(4.0, 0)])
    assert_rel(ora.shade_hit(w, c, 4), EXPECTED)


def test_shading_from_the_inside():
    w = default_world()
    w.set_light(0, [0, 0.25, 0], WHITE)
    c = ora.prepare(w, 0, [0, 0, 0], [0, 0, 1], [(0.5, 1)])
    assert_rel(ora.shade_hit(w, c, 4), [0.9049844720832575] * 3)


def test_color_at_miss_hit_behind():
    w = default_world()
    assert ora.color_at(w, [0, 0, -5], [0, 1, 0], 4) == BLACK
    assert_rel(ora.color_at(w, *RAY, 4), EXPECTED)
    for i in (0, 1):
        m = w.get_material_of(i); m.ambient = 1.0; w.set_material_of(i, m)
    assert_rel(ora.color_at(w, [0, 0, 0.75], [0, 0, -1], 4), WHITE)


def test_is_shadowed():
    w = default_world()
    assert not ora.is_shadowed(w, [0, 10, 0])
    assert ora.is_shadowed(w, [10, -10, 10])
    assert not ora.is_shadowed(w, [-20, 20, -20])
    assert not ora.is_shadowed(w, [-2, 2, -2])


def test_shade_hit_in_shadow():
    w = A.world_new()
    w.add_light([0, 0, -10], WHITE)
    s1 = w.add_shape(SPHERE, None, A.identity()); s2 = w.add_shape(SPHERE, None, A.translation(0, 0, 10))
    w.add_object_to_scene(shape(s1)); w.add_object_to_scene(shape(s2)); w.build()
    c = ora.prepare(w, 0, [0, 0, 5], [0, 0, 1], [(4.0, 1)])
    assert_rel(ora.shade_hit(w, c, 4), [0.1] * 3)


def test_over_under_point():
    w = A.world_new()
    s = w.add_shape(SPHERE, None, A.translation(0, 0, 1)); w.add_object_to_scene(shape(s)); w.build()
    c = ora.prepare(w, 0, *RAY, [(5.0, 0)])
    eps = 1e-10
    assert c.over_point[2] < -eps / 2 and c.point[2] > c.over_point[2]
    assert c.under_point[2] > eps / 2 and c.point[2] < c.under_point[2]


def _plane_world(reflective=0.5):
    def extra(w):
        s = w.add_shape(PLANE, A.material(reflective=reflective), A.translation(0, -1, 0))
        w.add_object_to_scene(shape(s))
    return default_world(extra)


DIAG = ([0, 0, -3], [0, -1 / SQRT2, 1 / SQRT2])


def test_reflection():
    w = A.world_new()
    s = w.add_shape(PLANE, None, A.identity()); w.add_object_to_scene(shape(s)); w.build()
    c = ora.prepare(w, 0, [0, 1, -1], [0, -1 / SQRT2, 1 / SQRT2], [(SQRT2, s)])
    assert_rel(c.reflectv, [0, 1 / SQRT2, 1 / SQRT2])
    w = default_world()
    m = w.get_material_of(1); m.ambient = 1.0; w.set_material_of(1, m)
    c = ora.prepare(w, 0, [0, 0, 0], [0, 0, 1], [(1.0, 1)])
    assert ora.reflected_color(w, c, 4) == BLACK
    w = _plane_world()
    c = ora.prepare(w, 0, *DIAG, [(SQRT2, 2)])
    assert_rel(ora.reflected_color(w, c, 4), [0.19032, 0.2379, 0.14274], 1e-4)
    assert ora.reflected_color(w, c, 0) == BLACK
    assert_rel(ora.shade_hit(w, c, 4), [0.87677, 0.92436, 0.82918], 1e-4)


def test_infinite_reflection_terminates():
    w = A.world_new()
    w.add_light([0, 0, 0], WHITE)
    m = A.material(reflective=1.0)
    lo = w.add_shape(PLANE, m, A.translation(0, -1, 0)); up = w.add_shape(PLANE, m, A.translation(0, 1, 0))
    w.add_object_to_scene(shape(lo)); w.add_object_to_scene(shape(up)); w.build()
    ora.color_at(w, [0, 0, 0], [0, 1, 0], 4)


@pytest.mark.parametrize("index,n1,n2", [(0, 1.0, 1.5), (1, 1.5, 2.0), (2, 2.0, 2.5), (3, 2.5, 2.5),
                                         (4, 2.5, 1.5), (5, 1.5, 1.0)])
def test_refraction_indices(index, n1, n2):
    w = A.world_new()
    for ri, t in ((1.5, A.scaling(2, 2, 2)), (2.0, A.translation(0, 0, -0.25)), (2.5, A.translation(0, 0, 0.25))):
        s = w.add_shape(SPHERE, A.material(transparency=1.0, refractive_index=ri), t)
        w.add_object_to_scene(shape(s))
    w.build()
    xs = ora.world_intersect(w, [0, 0, -4], [0, 0, 1])
    c = ora.prepare(w, index, [0, 0, -4], [0, 0, 1], xs)
    assert (c.n1, c.n2) == (n1, n2)


def _glassy_default():
    w = default_world()
    m = w.get_material_of(0); m.transparency = 1.0; m.refractive_index = 1.5; w.set_material_of(0, m)
    return w


def test_refracted_color_cases():
    w = default_world()
    c = ora.prepare(w, 0, *RAY, [(4.0, 0), (6.0, 0)])
    assert ora.refracted_color(w, c, 4) == BLACK
    w = _glassy_default()
    c = ora.prepare(w, 0, *RAY, [(4.0, 0), (6.0, 0)])
    assert ora.refracted_color(w, c, 0) == BLACK
    x = 1 / SQRT2
    c = ora.prepare(w, 1, [0, 0, x], [0, 1, 0], [(-x, 0), (x, 0)])
    assert ora.refracted_color(w, c, 4) == BLACK
    w = default_world()
    m = w.get_material_of(0); m.ambient = 1.0; m.pattern_kind = PAT_TEST; w.set_material_of(0, m)
    m = w.get_material_of(1); m.transparency = 1.0; m.refractive_index = 1.5; w.set_material_of(1, m)
    xs = [(-0.9899, 0), (-0.4899, 1), (0.4899, 1), (0.9899, 0)]
    c = ora.prepare(w, 2, [0, 0, 0.1], [0, 1, 0], xs)
    col = ora.refracted_color(w, c, 5)
    assert abs(col[0]) < 1e-12
    assert_rel(col[1:], [0.99888, 0.04722], 1e-4)


def _transparency_world(reflective=0.0):
    def extra(w):
        f = w.add_shape(PLANE, A.material(transparency=0.5, refractive_index=1.5, reflective=reflective),
                        A.translation(0, -1, 0))
        w.add_object_to_scene(shape(f))
        b = w.add_shape(SPHERE, A.material(color=[1, 0, 0], ambient=0.5), A.translation(0, -3.5, -0.5))
        w.add_object_to_scene(shape(b))
    return default_world(extra)


def test_shade_hit_transparent_and_schlick_mix():
    w = _transparency_world()
    c = ora.prepare(w, 0, *DIAG, [(SQRT2, 2)])
    assert_rel(ora.shade_hit(w, c, 5), [0.93642, 0.68642, 0.68642], 1e-4)
    w = _transparency_world(0.5)
    c = ora.prepare(w, 0, *DIAG, [(SQRT2, 2)])
    assert_rel(ora.shade_hit(w, c, 5), [0.93391, 0.69643, 0.69243], 1e-4)


def test_schlick():
    w = A.world_new()
    s = w.add_shape(SPHERE, A.material(transparency=1.0, refractive_index=1.5), A.identity())
    w.add_object_to_scene(shape(s)); w.build()
    x = 1 / SQRT2
    c = ora.prepare(w, 1, [0, 0, x], [0, 1, 0], [(-x, 0), (x, 0)])
    assert_rel(ora.schlick(c), 1.0)
    c = ora.prepare(w, 1, [0, 0, 0], [0, 1, 0], [(-1.0, 0), (1.0, 0)])
    assert_rel(ora.schlick(c), 0.04, 1e-15)
    c = ora.prepare(w, 0, [0, 0.99, -2], [0, 0, 1], [(1.8589, 0)])
    assert_rel(ora.schlick(c), 0.48873, 1e-4)


def _hierarchy(g2_scale):
    w = A.world_new()
    g1 = w.add_group(A.rotation_y(PI / 2)); g2 = w.add_group(A.scaling(*g2_scale))
    s = w.add_shape(SPHERE, None, A.translation(5, 0, 0))
    w.add_object_to_scene(group(g1)); w.set_group_of(group(g2), g1); w.set_group_of(shape(s), g2)
    return w.build(), g1, g2, s


def test_group_hierarchy_conversions():
    w, g1, g2, s = _hierarchy((2, 2, 2))
    assert ora.LIB.ora_parent_of(w.h, OBJ_SHAPE, s, None) == OBJ_GROUP
    assert ora.LIB.ora_parent_of(w.h, OBJ_GROUP, g1, None) == -1
    p = ora.vec3(ora.LIB.ora_world_to_object, w.h, OBJ_SHAPE, s, d3([-2, 0, -10]))
    assert abs(p[0]) < 1e-14 and abs(p[1]) < 1e-14 and abs(p[2] + 1) < 1e-14
    w, g1, g2, s = _hierarchy((1, 2, 3))
    x = math.sqrt(3) / 3
    n = ora.vec3(ora.LIB.ora_normal_to_world, w.h, OBJ_SHAPE, s, d3([x, x, x]))
    assert_rel(n, [0.2857, 0.4286, -0.8571], 1e-4)
    s3 = math.sqrt(3)
    n = ora.vec3(ora.LIB.ora_normal_at, w.h, s, d3([s3, 2 / 3 * s3, -5 - s3 / 3]))
    assert_rel(n, [0.2857, 0.4286, -0.8571], 1e-4)


def test_group_translation_culls_and_nested_bounds():
    w = A.world_new()
    g = w.add_group(A.translation(0, 2, 0))
    s = w.add_shape(SPHERE, None, A.identity()); w.set_group_of(shape(s), g)
    w.add_object_to_scene(group(g)); w.build()
    assert ora.world_intersect(w, [0, 0, 0], [0, 0, 1]) == []
    w = A.world_new()
    g1 = w.add_group(A.translation(0, 1, 0)); g2 = w.add_group(A.translation(0, 1, 0))
    s = w.add_shape(CUBE, None, A.translation(0, 1, 0))
    w.set_group_of(shape(s), g2); w.set_group_of(group(g2), g1); w.add_object_to_scene(group(g1)); w.build()
    assert len(ora.world_intersect(w, [0, 3, -2], [0, 0, 1])) == 2


# ------------------------------------------------------------------ color.rs:40-48 (no reference test; literal restatement)
def test_quantisation():
    out = (ora.C.c_uint8 * 3)()
    ora.LIB.ora_color_to_rgb(d3([0.0, 0.5, 1.0]), out); assert list(out) == [0, 128, 255]
    ora.LIB.ora_color_to_rgb(d3([-1.0, 2.0, float("nan")]), out); assert list(out) == [0, 255, 0]
    ora.LIB.ora_color_to_rgb(d3([0.5 / 255, 1.5 / 255, 0.499 / 255]), out); assert list(out) == [1, 2, 0]
