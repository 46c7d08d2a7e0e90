// rt_world.cpp — host World/WorldBuilder/Camera mirror, bounds pre-pass, flattener and
// the rtxh_* C entry points.  Host arithmetic follows nalgebra 0.30.1's operation order
// (the reference's only math dependency, Cargo.lock:368-381) because the bits of every
// inverse and bound decide hit ids on the device.  Compile with -ffp-contract=off.
#include "rt_world.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace rth {

static const double kInf = std::numeric_limits<double>::infinity();
static const double kNaN = std::numeric_limits<double>::quiet_NaN();

// ------------------------------------------------------------------ matrices
Mat4 mat_identity() {
    Mat4 m;
    for (int i = 0; i < 16; i++) m.e[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return m;
}
Mat4 mat_translation(double x, double y, double z) {
    Mat4 m = mat_identity(); m(0, 3) = x; m(1, 3) = y; m(2, 3) = z; return m;
}
Mat4 mat_scaling(double x, double y, double z) {
    Mat4 m = mat_identity(); m(0, 0) = x; m(1, 1) = y; m(2, 2) = z; return m;
}
// f64::sin_cos() = (sin(), cos()): two libm calls.  glibc's sincos() differs from cos() in the last
// ulp for some arguments, so the pair must not be merged into one sincos() call.
__attribute__((noinline)) static double sin_alone(double x) { return std::sin(x); }
__attribute__((noinline)) static double cos_alone(double x) { return std::cos(x); }

// Rotation3::from_axis_angle for the unit axis e_axis, embedded in a 4x4 (na::convert)
Mat4 mat_rotation(int axis, double angle) {
    Mat4 m = mat_identity();
    if (angle == 0.0) return m;                 // simd_ne(zero).if_else(.., identity)
    const double u[3] = {axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0};
    const double sn = sin_alone(angle), cs = cos_alone(angle), k = 1.0 - cs;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            if (r == c) {
                const double sq = u[r] * u[r];
                m(r, c) = sq + (1.0 - sq) * cs;
            } else {
                // off-diagonal: u_r*u_c*k -/+ u_w*sin, w = the third axis, sign from the cyclic order
                const int w = 3 - r - c;
                const bool plus = ((r + 1) % 3) != c;     // (0,2)+ (1,0)+ (2,1)+ ; (0,1)- (1,2)- (2,0)-
                const double base = (r < c ? u[r] * u[c] : u[c] * u[r]) * k;
                m(r, c) = plus ? base + u[w] * sn : base - u[w] * sn;
            }
        }
    return m;
}
// gemm as nalgebra does it for small fixed sizes: per output column, accumulate
// column k of `a` scaled by b(k,j), k ascending.
Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; j++) {
        double col[4];
        for (int i = 0; i < 4; i++) col[i] = a(i, 0) * b(0, j);
        for (int k = 1; k < 4; k++)
            for (int i = 0; i < 4; i++) col[i] = a(i, k) * b(k, j) + col[i];
        for (int i = 0; i < 4; i++) r(i, j) = col[i];
    }
    return r;
}
// do_inverse4: adjugate by cofactor expansion on the column-major slice.  Table rows:
// output slot, then six (sign, a, b, c) products evaluated left to right.
Mat4 mat_inverse(const Mat4& a) {
    static const signed char T[16][6][4] = {
        {{+1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {+1, 9, 7, 14}, {+1, 13, 6, 11}, {-1, 13, 7, 10}},
        {{-1, 1, 10, 15}, {+1, 1, 11, 14}, {+1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {+1, 13, 3, 10}},
        {{+1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {+1, 5, 3, 14}, {+1, 13, 2, 7}, {-1, 13, 3, 6}},
        {{-1, 1, 6, 11}, {+1, 1, 7, 10}, {+1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {+1, 9, 3, 6}},
        {{-1, 4, 10, 15}, {+1, 4, 11, 14}, {+1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {+1, 12, 7, 10}},
        {{+1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {+1, 8, 3, 14}, {+1, 12, 2, 11}, {-1, 12, 3, 10}},
        {{-1, 0, 6, 15}, {+1, 0, 7, 14}, {+1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {+1, 12, 3, 6}},
        {{+1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {+1, 4, 3, 10}, {+1, 8, 2, 7}, {-1, 8, 3, 6}},
        {{+1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {+1, 8, 7, 13}, {+1, 12, 5, 11}, {-1, 12, 7, 9}},
        {{-1, 0, 9, 15}, {+1, 0, 11, 13}, {+1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {+1, 12, 3, 9}},
        {{+1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {+1, 4, 3, 13}, {+1, 12, 1, 7}, {-1, 12, 3, 5}},
        {{-1, 0, 5, 11}, {+1, 0, 7, 9}, {+1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {+1, 8, 3, 5}},
        {{-1, 4, 9, 14}, {+1, 4, 10, 13}, {+1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {+1, 12, 6, 9}},
        {{+1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {+1, 8, 2, 13}, {+1, 12, 1, 10}, {-1, 12, 2, 9}},
        {{-1, 0, 5, 14}, {+1, 0, 6, 13}, {+1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {+1, 12, 2, 5}},
        {{+1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {+1, 4, 2, 9}, {+1, 8, 1, 6}, {-1, 8, 2, 5}},
    };
    const double* m = a.e;
    double adj[16];
    for (int s = 0; s < 16; s++) {
        double acc = 0.0;
        for (int t = 0; t < 6; t++) {
            const signed char* q = T[s][t];
            const double term = m[q[1]] * m[q[2]] * m[q[3]];
            if (t == 0) acc = q[0] > 0 ? term : -term;
            else acc = q[0] > 0 ? acc + term : acc - term;
        }
        adj[s] = acc;
    }
    const double det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    Mat4 r;
    if (det != 0.0) {
        const double inv_det = 1.0 / det;
        for (int i = 0; i < 16; i++) r.e[i] = adj[i] * inv_det;
    } else {
        for (int i = 0; i < 16; i++) r.e[i] = adj[i];   // try_inverse_mut's bool is ignored by Transform::inverse
    }
    return r;
}
// Projective3 * Point3: (M3x3 p + t), divided by (bottom row . p + m33) when that is non-zero
void mat_point(const Mat4& m, const double* p, double* out) {
    double r[3];
    for (int i = 0; i < 3; i++) {
        double y = m(i, 0) * p[0];
        y = m(i, 1) * p[1] + y;
        y = m(i, 2) * p[2] + y;
        r[i] = y + m(i, 3);
    }
    double n = 0.0;
    n += m(3, 0) * p[0]; n += m(3, 1) * p[1]; n += m(3, 2) * p[2];
    n = n + m(3, 3);
    if (n != 0.0) for (int i = 0; i < 3; i++) r[i] = r[i] / n;
    for (int i = 0; i < 3; i++) out[i] = r[i];
}

static void vsub(const double* a, const double* b, double* o) { for (int i = 0; i < 3; i++) o[i] = a[i] - b[i]; }
static void vnormalize(double* v) {
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int i = 0; i < 3; i++) v[i] = v[i] / n;
}
static void vcross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
Mat4 mat_view(const double* from, const double* to, const double* up) {
    double fwd[3], upn[3] = {up[0], up[1], up[2]}, left[3], tup[3];
    vsub(to, from, fwd); vnormalize(fwd); vnormalize(upn);
    vcross(fwd, upn, left); vcross(left, fwd, tup);
    Mat4 o = mat_identity();
    for (int c = 0; c < 3; c++) { o(0, c) = left[c]; o(1, c) = tup[c]; o(2, c) = -fwd[c]; }
    return mat_mul(o, mat_translation(0.0 - from[0], 0.0 - from[1], 0.0 - from[2]));
}

// ------------------------------------------------------------------ material / camera
void default_material(rtxh_material* m) {
    std::memset(m, 0, sizeof *m);
    m->ambient = 0.1; m->diffuse = 0.9; m->specular = 0.9; m->shininess = 200.0;
    m->reflective = 0.0; m->transparency = 0.0; m->refractive_index = 1.0;
    for (int i = 0; i < 3; i++) m->color_a[i] = 1.0;     // Pattern::uniform(white)
    m->pattern_kind = RTX_PAT_PLAIN;
    Mat4 id = mat_identity(); std::memcpy(m->pattern_transform, id.e, sizeof id.e);
}
void camera_new(int hsize, int vsize, double fov, rtxh_camera* c) {
    const double half_view = std::tan(fov / 2.0);
    const double aspect = (double)hsize / (double)vsize;
    c->hsize = hsize; c->vsize = vsize;
    if (aspect >= 1.0) { c->half_width = half_view; c->half_height = half_view / aspect; }
    else { c->half_width = half_view * aspect; c->half_height = half_view; }
    c->pixel_size = (c->half_width * 2.0) / (double)hsize;
    Mat4 id = mat_identity(); std::memcpy(c->transform, id.e, sizeof id.e);
}

// ------------------------------------------------------------------ World
int World::node_of(int kind, int index) const {
    const std::vector<int>& v = kind == OBJ_GROUP ? group_nodes : kind == OBJ_CSG ? csg_nodes : shape_nodes;
    if (index < 0 || (size_t)index >= v.size()) return -1;
    return v[index];
}
static void empty_bounds(double* b) { for (int i = 0; i < 6; i++) b[i] = kNaN; }

int World::add_shape(int type, const rtxh_material* m, const Mat4& t) {
    Node n; n.kind = type; n.transform = t; n.shape_index = (int)shape_nodes.size();
    if (m) n.material = *m; else default_material(&n.material);
    empty_bounds(n.bounds);
    nodes.push_back(n); shape_nodes.push_back((int)nodes.size() - 1);
    return n.shape_index;
}
int World::add_group(const Mat4& t) {
    Node n; n.kind = RTX_NODE_GROUP; n.transform = t; default_material(&n.material); empty_bounds(n.bounds);
    nodes.push_back(n); group_nodes.push_back((int)nodes.size() - 1);
    return (int)group_nodes.size() - 1;
}
void World::set_group_of(int kind, int index, int group) {
    const int c = node_of(kind, index), g = node_of(OBJ_GROUP, group);
    if (c < 0 || g < 0) return;
    nodes[g].children.push_back(c); nodes[c].parent = g;
}
int World::add_csg(int op, int lk, int li, int rk, int ri, const Mat4& t) {
    const int l = node_of(lk, li), r = node_of(rk, ri);
    Node n; n.kind = RTX_NODE_CSG; n.transform = t; n.csg_op = op; default_material(&n.material); empty_bounds(n.bounds);
    n.children = {l, r};
    nodes.push_back(n);
    const int me = (int)nodes.size() - 1;
    csg_nodes.push_back(me);
    if (l >= 0) nodes[l].parent = me;
    if (r >= 0) nodes[r].parent = me;
    return (int)csg_nodes.size() - 1;
}
void World::add_object_to_scene(int kind, int index) { const int n = node_of(kind, index); if (n >= 0) scene.push_back(n); }
int World::add_light(const double* pos, const double* intensity) {
    Light l; for (int i = 0; i < 3; i++) { l.pos[i] = pos[i]; l.intensity[i] = intensity[i]; }
    lights.push_back(l); return (int)lights.size() - 1;
}

// Bounds::to_global + from_points (bounds.rs:29-40,63-74) merged into `acc` (merge_bounds :56-61).
// from_points seeds min with corner 0 and max with corner 1 and folds corners 1..7, so corner 0
// never reaches max — reproduced on purpose (it decides which rays a group culls).
static void merge_child_bounds(double* acc, const double* child, const Mat4& t) {
    double mn[3], mx[3];
    int idx = 0;
    for (int ix = 0; ix < 2; ix++) for (int iy = 0; iy < 2; iy++) for (int iz = 0; iz < 2; iz++, idx++) {
        const double p[3] = {child[ix * 3 + 0], child[iy * 3 + 1], child[iz * 3 + 2]};
        double q[3]; mat_point(t, p, q);
        if (idx == 0) { for (int a = 0; a < 3; a++) mn[a] = q[a]; continue; }
        if (idx == 1) for (int a = 0; a < 3; a++) mx[a] = q[a];
        for (int a = 0; a < 3; a++) { mn[a] = std::fmin(mn[a], q[a]); mx[a] = std::fmax(mx[a], q[a]); }
    }
    for (int a = 0; a < 3; a++) { acc[a] = std::fmin(acc[a], mn[a]); acc[3 + a] = std::fmax(acc[3 + a], mx[a]); }
}
void World::set_bounds_of(int node, double* out6) {
    Node& n = nodes[node];
    if (n.kind == RTX_NODE_GROUP || n.kind == RTX_NODE_CSG) {
        double acc[6]; empty_bounds(acc);
        for (size_t k = 0; k < n.children.size(); k++) {
            const int c = nodes[node].children[k];
            if (c < 0) continue;
            double cb[6]; set_bounds_of(c, cb);
            merge_child_bounds(acc, cb, nodes[c].transform);
        }
        std::memcpy(nodes[node].bounds, acc, sizeof acc);
        std::memcpy(out6, acc, sizeof acc);
    } else if (n.kind == RTX_NODE_PLANE) {                 // plane.rs:29-33
        const double b[6] = {-kInf, 0.0, -kInf, kInf, 0.0, kInf}; std::memcpy(out6, b, sizeof b);
    } else {                                               // sphere.rs:38-40, cube.rs:49-51
        const double b[6] = {-1.0, -1.0, -1.0, 1.0, 1.0, 1.0}; std::memcpy(out6, b, sizeof b);
    }
}
void World::build() { double tmp[6]; for (int r : scene) set_bounds_of(r, tmp); }

// ------------------------------------------------------------------ flattener (-> rtx_scene_desc)
void flatten(const World& w, Flattened& f) {
    const int n = (int)w.nodes.size();
    f.kind.resize(n); f.first_child.resize(n); f.child_count.resize(n); f.csg_op.resize(n);
    f.shape_index.resize(n); f.material.resize(n); f.inverse.resize((size_t)n * 16); f.bounds.resize((size_t)n * 6);
    f.children.clear(); f.materials.clear();
    for (int i = 0; i < n; i++) {
        const Node& nd = w.nodes[i];
        f.kind[i] = nd.kind;
        const Mat4 inv = mat_inverse(nd.transform);        // object_store.rs:245, once instead of per ray
        std::memcpy(&f.inverse[(size_t)i * 16], inv.e, sizeof inv.e);
        std::memcpy(&f.bounds[(size_t)i * 6], nd.bounds, sizeof nd.bounds);
        f.first_child[i] = (int)f.children.size();
        f.child_count[i] = (int)nd.children.size();
        for (int c : nd.children) f.children.push_back(c);
        f.csg_op[i] = nd.csg_op;
        f.shape_index[i] = nd.shape_index;
        f.material[i] = -1;
        if (nd.kind <= RTX_NODE_CUBE) {
            rtx_material m; std::memset(&m, 0, sizeof m);
            const rtxh_material& s = nd.material;
            m.ambient = s.ambient; m.diffuse = s.diffuse; m.specular = s.specular; m.shininess = s.shininess;
            m.reflective = s.reflective; m.transparency = s.transparency; m.refractive_index = s.refractive_index;
            for (int k = 0; k < 3; k++) { m.color_a[k] = s.color_a[k]; m.color_b[k] = s.color_b[k]; }
            Mat4 pt; std::memcpy(pt.e, s.pattern_transform, sizeof pt.e);
            const Mat4 pinv = mat_inverse(pt);             // object_store.rs:297
            std::memcpy(m.pattern_inverse, pinv.e, sizeof pinv.e);
            m.pattern_kind = s.pattern_kind;
            f.material[i] = (int)f.materials.size();
            f.materials.push_back(m);
        }
    }
    f.roots.assign(w.scene.begin(), w.scene.end());
    f.light_pos.clear(); f.light_int.clear();
    for (const Light& l : w.lights) for (int k = 0; k < 3; k++) { f.light_pos.push_back(l.pos[k]); f.light_int.push_back(l.intensity[k]); }
    rtx_scene_desc& d = f.desc; std::memset(&d, 0, sizeof d);
    d.n_nodes = n; d.node_kind = f.kind.data(); d.node_inverse = f.inverse.data(); d.node_bounds = f.bounds.data();
    d.node_first_child = f.first_child.data(); d.node_child_count = f.child_count.data(); d.node_csg_op = f.csg_op.data();
    d.node_shape_index = f.shape_index.data(); d.node_material = f.material.data();
    d.n_children = (int)f.children.size(); d.children = f.children.data();
    d.n_roots = (int)f.roots.size(); d.roots = f.roots.data();
    d.n_materials = (int)f.materials.size(); d.materials = f.materials.data();
    d.n_lights = (int)w.lights.size(); d.light_position = f.light_pos.data(); d.light_intensity = f.light_int.data();
}

}  // namespace rth

// ------------------------------------------------------------------ C entry points
using namespace rth;
static Mat4 m4(const double* p) { Mat4 m; std::memcpy(m.e, p, sizeof m.e); return m; }
static void put(double* o, const Mat4& m) { std::memcpy(o, m.e, sizeof m.e); }

extern "C" {

void rtxh_identity(double* o) { put(o, mat_identity()); }
void rtxh_translation(double x, double y, double z, double* o) { put(o, mat_translation(x, y, z)); }
void rtxh_scaling(double x, double y, double z, double* o) { put(o, mat_scaling(x, y, z)); }
void rtxh_rotation_x(double a, double* o) { put(o, mat_rotation(0, a)); }
void rtxh_rotation_y(double a, double* o) { put(o, mat_rotation(1, a)); }
void rtxh_rotation_z(double a, double* o) { put(o, mat_rotation(2, a)); }
void rtxh_view_transform(const double* f, const double* t, const double* u, double* o) { put(o, mat_view(f, t, u)); }
void rtxh_mat_mul(const double* a, const double* b, double* o) { put(o, mat_mul(m4(a), m4(b))); }
void rtxh_inverse(const double* m, double* o) { put(o, mat_inverse(m4(m))); }
void rtxh_default_material(rtxh_material* m) { default_material(m); }

rtxh_world* rtxh_world_new(void) { return new rtxh_world(); }
void rtxh_world_free(rtxh_world* w) { delete w; }
int rtxh_add_shape(rtxh_world* w, int type, const rtxh_material* m, const double* t) { return w->w.add_shape(type, m, m4(t)); }
int rtxh_add_group(rtxh_world* w, const double* t) { return w->w.add_group(m4(t)); }
void rtxh_set_group_of(rtxh_world* w, int k, int i, int g) { w->w.set_group_of(k, i, g); }
int rtxh_add_csg(rtxh_world* w, int op, int lk, int li, int rk, int ri, const double* t) { return w->w.add_csg(op, lk, li, rk, ri, m4(t)); }
void rtxh_add_object_to_scene(rtxh_world* w, int k, int i) { w->w.add_object_to_scene(k, i); }
int rtxh_add_light(rtxh_world* w, const double* p, const double* c) { return w->w.add_light(p, c); }
void rtxh_build(rtxh_world* w) { w->w.build(); }
void rtxh_set_material_of(rtxh_world* w, int s, const rtxh_material* m) { int n = w->w.node_of(OBJ_SHAPE, s); if (n >= 0) w->w.nodes[n].material = *m; }
void rtxh_get_material_of(rtxh_world* w, int s, rtxh_material* m) { int n = w->w.node_of(OBJ_SHAPE, s); if (n >= 0) *m = w->w.nodes[n].material; }
void rtxh_set_light(rtxh_world* w, int i, const double* p, const double* c) {
    if (i < 0 || (size_t)i >= w->w.lights.size()) return;
    for (int k = 0; k < 3; k++) { w->w.lights[i].pos[k] = p[k]; w->w.lights[i].intensity[k] = c[k]; }
}
void rtxh_get_bounds(rtxh_world* w, int k, int i, double* o) {
    int n = w->w.node_of(k, i);
    if (n < 0) return;
    const Node& nd = w->w.nodes[n];
    if (nd.kind == RTX_NODE_GROUP || nd.kind == RTX_NODE_CSG) std::memcpy(o, nd.bounds, sizeof nd.bounds);
    else { double b[6]; w->w.set_bounds_of(n, b); std::memcpy(o, b, sizeof b); }
}
void rtxh_camera_new(int h, int v, double fov, rtxh_camera* c) { camera_new(h, v, fov, c); }

int64_t rtxh_world_serialize(const rtxh_world* w, uint8_t* buf, int64_t cap) {
    std::vector<uint8_t> o;
    auto put_i = [&](int32_t v) { const uint8_t* p = (const uint8_t*)&v; o.insert(o.end(), p, p + 4); };
    auto put_d = [&](const double* v, int n) { const uint8_t* p = (const uint8_t*)v; o.insert(o.end(), p, p + 8 * n); };
    put_i((int32_t)w->w.nodes.size());
    for (const Node& n : w->w.nodes) {
        put_i(n.kind); put_i(n.parent); put_d(n.transform.e, 16);
        put_i((int32_t)n.children.size()); for (int c : n.children) put_i(c);
        put_i(n.csg_op); put_i(n.shape_index);
        const rtxh_material& m = n.material;
        put_d(&m.ambient, 7); put_d(m.color_a, 3); put_d(m.color_b, 3); put_d(m.pattern_transform, 16); put_i(m.pattern_kind);
        put_d(n.bounds, 6);
    }
    put_i((int32_t)w->w.scene.size()); for (int r : w->w.scene) put_i(r);
    put_i((int32_t)w->w.lights.size());
    for (const Light& l : w->w.lights) { put_d(l.pos, 3); put_d(l.intensity, 3); }
    if (buf && cap >= (int64_t)o.size()) std::memcpy(buf, o.data(), o.size());
    return (int64_t)o.size();
}

void rtxh_camera_flatten(const rtxh_camera* c, rtx_camera* o) {
    o->hsize = c->hsize; o->vsize = c->vsize;
    o->pixel_size = c->pixel_size; o->half_width = c->half_width; o->half_height = c->half_height;
    put(o->inverse, mat_inverse(m4(c->transform)));
}
int rtxh_scene_create(rtx_ctx* ctx, rtxh_world* w, rtx_scene** out) {
    if (!ctx || !w || !out) return RTX_E_INVALID;
    Flattened f; flatten(w->w, f);
    return rtx_scene_create(ctx, &f.desc, out);
}
int rtxh_render(rtx_ctx* ctx, rtxh_world* w, const rtxh_camera* cam, int max_depth, uint32_t flags,
                uint8_t* rgb, int32_t* hit, rtx_stats* stats) {
    if (!ctx || !w || !cam || !rgb) return RTX_E_INVALID;
    rtx_scene* s = nullptr;
    int rc = rtxh_scene_create(ctx, w, &s);
    if (rc != RTX_OK) return rc;
    rtx_camera rc_cam; rtxh_camera_flatten(cam, &rc_cam);
    rtx_bands all = {0, 1};
    rc = rtx_render(ctx, s, &rc_cam, max_depth, all, flags & ~(uint32_t)RTX_F_BANDS_COMPACT, rgb, hit, stats);
    rtx_scene_destroy(s);
    return rc;
}

}  // extern "C"
