// rt_oracle.cpp — CPU ORACLE: literal restatement of felgru/ray-tracer's
// trace-and-shade path.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
//
// Structure deliberately mirrors the reference ("faithful-cost"): a fresh 4x4
// inverse per object per ray (object_store.rs:245), a heap vector per
// primitive test (sphere.rs:26-30), a stable re-sort after every append
// (intersections.rs:53,58) and true recursion (world.rs:100-193).  Every
// function cites the reference lines it follows.  Build with
// -O2 -ffp-contract=off: Rust never contracts a*b+c.
#include "rt_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

const double INF = std::numeric_limits<double>::infinity();
const double NaN = std::numeric_limits<double>::quiet_NaN();
// EPSILON-tie thresholds: far above f64 rounding noise (1e-16 relative), far below the
// reference's own 1e-10 surface offset (world.rs:298-300) so that by-design near-zero
// self-intersections of shadow/secondary rays are not flagged.
const double TIE_EPS = 1e-11;     // relative gap between competing t's / pattern cell borders
const double TIE_ZERO = 1e-12;    // |t| this close to 0 could flip the t >= 0 test

// ---------------------------------------------------------------- nalgebra
struct V3 { double x, y, z; };            // Point3 / Vector3 (geometry.rs:7-8)
struct M4 { double m[16]; };              // column-major Matrix4 (geometry.rs:10)

inline double& at(M4& a, int r, int c) { return a.m[c * 4 + r]; }
inline double at(const M4& a, int r, int c) { return a.m[c * 4 + r]; }

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }

// nalgebra dot for 3x1: (a0*b0 + a1*b1) + a2*b2
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// norm = sqrt(0 + dot); normalize = unscale(norm) (component-wise division)
inline double norm(V3 a) { double r = 0.0; r += dot(a, a); return std::sqrt(r); }
inline V3 normalize(V3 a) { return a / norm(a); }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

M4 identity() {
    M4 r; for (int i = 0; i < 16; i++) r.m[i] = 0.0;
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r;
}

// Matrix * Matrix: per output column, gemv = column-axpy accumulation in
// increasing column order: ((a_i0*b_0j + a_i1*b_1j) + a_i2*b_2j) + a_i3*b_3j
M4 mat_mul(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            double y = at(a, i, 0) * at(b, 0, j);
            for (int k = 1; k < 4; k++) y = at(a, i, k) * at(b, k, j) + y;
            at(r, i, j) = y;
        }
    return r;
}

// 3x3 block times vector (gemv, same accumulation order)
inline V3 m33_mul(const M4& m, V3 v) {
    V3 r;
    r.x = at(m, 0, 1) * v.y + at(m, 0, 0) * v.x; r.x = at(m, 0, 2) * v.z + r.x;
    r.y = at(m, 1, 1) * v.y + at(m, 1, 0) * v.x; r.y = at(m, 1, 2) * v.z + r.y;
    r.z = at(m, 2, 1) * v.y + at(m, 2, 0) * v.x; r.z = at(m, 2, 2) * v.z + r.z;
    return r;
}

// Projective3 * Point3 (nalgebra transform_ops: normalizer.tr_dot + m33, divide if != 0)
V3 mul_point(const M4& m, V3 p) {
    V3 t = {at(m, 0, 3), at(m, 1, 3), at(m, 2, 3)};
    double n = 0.0;
    n += at(m, 3, 0) * p.x; n += at(m, 3, 1) * p.y; n += at(m, 3, 2) * p.z;
    n = n + at(m, 3, 3);
    V3 r = m33_mul(m, p) + t;
    if (n != 0.0) return r / n;   // !n.is_zero(): NaN != 0 is true
    return r;
}

// Projective3 * Vector3
V3 mul_vector(const M4& m, V3 v) {
    double n = 0.0;
    n += at(m, 3, 0) * v.x; n += at(m, 3, 1) * v.y; n += at(m, 3, 2) * v.z;
    if (n != 0.0) return m33_mul(m, v / n);
    return m33_mul(m, v);
}

// Matrix4::try_inverse_mut -> do_inverse4 (closed-form adjugate on the column-major slice)
M4 inverse(const M4& a) {
    const double* m = a.m;
    double o[16];
    o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15]
         + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15]
         - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15]
         + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11]
         - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15]
         - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15]
         + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15]
         - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11]
         + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15]
         + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15]
         - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15]
          + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11]
          - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14]
          - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14]
          + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14]
          - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10]
          + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
    M4 r;
    if (det != 0.0) {
        double inv_det = 1.0 / det;
        for (int i = 0; i < 16; i++) r.m[i] = o[i] * inv_det;
    } else {
        for (int i = 0; i < 16; i++) r.m[i] = o[i];   // unchecked: reference ignores the bool
    }
    return r;
}

// ---------------------------------------------------------------- geometry.rs
M4 translation(V3 v) {                       // geometry.rs:16-23
    M4 r = identity(); at(r, 0, 3) = v.x; at(r, 1, 3) = v.y; at(r, 2, 3) = v.z; return r;
}
M4 scaling(double x, double y, double z) {   // geometry.rs:25-31
    M4 r = identity(); at(r, 0, 0) = x; at(r, 1, 1) = y; at(r, 2, 2) = z; return r;
}
// f64::sin_cos() is (sin(), cos()): two separate libm calls.  glibc's sincos() can differ from cos()
// in the last ulp, so keep the compiler from merging the pair (GCC does at -O2).
__attribute__((noinline)) double rust_sin(double x) { return std::sin(x); }
__attribute__((noinline)) double rust_cos(double x) { return std::cos(x); }

// geometry.rs:45-47 -> Rotation3::from_axis_angle (nalgebra) -> to_homogeneous
M4 rotation_around_axis(V3 u, double angle) {
    if (!(angle != 0.0)) return identity();
    double ux = u.x, uy = u.y, uz = u.z;
    double sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    double s = rust_sin(angle), c = rust_cos(angle);
    double k = 1.0 - c;
    M4 r = identity();
    at(r, 0, 0) = sqx + (1.0 - sqx) * c;
    at(r, 0, 1) = ux * uy * k - uz * s;
    at(r, 0, 2) = ux * uz * k + uy * s;
    at(r, 1, 0) = ux * uy * k + uz * s;
    at(r, 1, 1) = sqy + (1.0 - sqy) * c;
    at(r, 1, 2) = uy * uz * k - ux * s;
    at(r, 2, 0) = ux * uz * k - uy * s;
    at(r, 2, 1) = uy * uz * k + ux * s;
    at(r, 2, 2) = sqz + (1.0 - sqz) * c;
    return r;
}
V3 reflect(V3 v, V3 n) { return v - n * 2.0 * dot(v, n); }   // geometry.rs:49-51
M4 view_transform(V3 from, V3 to, V3 up) {                   // geometry.rs:53-67
    V3 forward = normalize(to - from);
    V3 upn = normalize(up);
    V3 left = cross(forward, upn);
    V3 true_up = cross(left, forward);
    M4 o = identity();
    at(o, 0, 0) = left.x;     at(o, 0, 1) = left.y;     at(o, 0, 2) = left.z;
    at(o, 1, 0) = true_up.x;  at(o, 1, 1) = true_up.y;  at(o, 1, 2) = true_up.z;
    at(o, 2, 0) = -forward.x; at(o, 2, 1) = -forward.y; at(o, 2, 2) = -forward.z;
    V3 zero = {0.0, 0.0, 0.0};
    return mat_mul(o, translation(zero - from));
}

// ---------------------------------------------------------------- color.rs
struct Color { double r, g, b; };
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator*(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }
const Color BLACK = {0.0, 0.0, 0.0};

inline double rust_clamp01(double x) {        // f64::clamp(0.,1.): NaN stays NaN
    if (x < 0.0) return 0.0;
    if (x > 1.0) return 1.0;
    return x;
}
inline uint8_t discretize(double x) {         // color.rs:41-43: round() then saturating `as u8`
    double v = std::round(x * 255.0);         // round half away from zero
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

// ---------------------------------------------------------------- rays.rs
struct Ray { V3 origin, direction; };
inline V3 position(const Ray& r, double t) { return r.origin + t * r.direction; }   // rays.rs:17-19
inline Ray transform(const Ray& r, const M4& m) {                                   // rays.rs:21-23
    return {mul_point(m, r.origin), mul_vector(m, r.direction)};
}

// ---------------------------------------------------------------- bounds.rs
struct Bounds { V3 min, max; };
inline double rmin(double a, double b) { return std::fmin(a, b); }   // f64::min: NaN-ignoring
inline double rmax(double a, double b) { return std::fmax(a, b); }
Bounds bounds_empty() { return {{NaN, NaN, NaN}, {NaN, NaN, NaN}}; }   // bounds.rs:23-27

// bounds.rs:29-40 — including the seed quirk: min=points[0], max=points[1], loop over points[1..]
Bounds bounds_from_points(const std::vector<V3>& p) {
    V3 mn = p[0], mx = p[1];
    for (size_t i = 1; i < p.size(); i++) {
        mn.x = rmin(mn.x, p[i].x); mx.x = rmax(mx.x, p[i].x);
        mn.y = rmin(mn.y, p[i].y); mx.y = rmax(mx.y, p[i].y);
        mn.z = rmin(mn.z, p[i].z); mx.z = rmax(mx.z, p[i].z);
    }
    return {mn, mx};
}
// bounds.rs:77-94 (and cube.rs:54-71 with min=-1,max=1)
inline void check_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    double tmin_numerator = mn - origin;
    double tmax_numerator = mx - origin;
    if (std::fabs(direction) >= 1e-12) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = tmin_numerator * INF;
        tmax = tmax_numerator * INF;
    }
    if (tmin > tmax) std::swap(tmin, tmax);
}
bool bounds_intersects(const Bounds& b, const Ray& r) {          // bounds.rs:42-54
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(r.origin.x, r.direction.x, b.min.x, b.max.x, xtmin, xtmax);
    check_axis(r.origin.y, r.direction.y, b.min.y, b.max.y, ytmin, ytmax);
    check_axis(r.origin.z, r.direction.z, b.min.z, b.max.z, ztmin, ztmax);
    double tmin = rmax(rmax(xtmin, ytmin), ztmin);
    double tmax = rmin(rmin(xtmax, ytmax), ztmax);
    return tmin <= tmax;
}
void bounds_merge(Bounds& a, const Bounds& o) {                  // bounds.rs:56-61
    a.min.x = rmin(a.min.x, o.min.x); a.max.x = rmax(a.max.x, o.max.x);
    a.min.y = rmin(a.min.y, o.min.y); a.max.y = rmax(a.max.y, o.max.y);
    a.min.z = rmin(a.min.z, o.min.z); a.max.z = rmax(a.max.z, o.max.z);
}
Bounds bounds_to_global(const Bounds& b, const M4& t) {          // bounds.rs:63-74
    std::vector<V3> corners;
    double xs[2] = {b.min.x, b.max.x}, ys[2] = {b.min.y, b.max.y}, zs[2] = {b.min.z, b.max.z};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int k = 0; k < 2; k++)
        corners.push_back(mul_point(t, V3{xs[i], ys[j], zs[k]}));
    return bounds_from_points(corners);
}

// ---------------------------------------------------------------- shapes
struct Counters { uint64_t ev[ORA_EV_COUNT]; uint64_t rays_radiance, rays_shadow; bool tie, tie_id; };
thread_local Counters* g_cnt = nullptr;
inline void count(int e, uint64_t n = 1) { if (g_cnt) g_cnt->ev[e] += n; }

std::vector<double> sphere_intersect(const Ray& ray) {           // sphere.rs:16-32
    count(ORA_EV_SPHERE);
    V3 zero = {0.0, 0.0, 0.0};
    V3 s2r = ray.origin - zero;
    double a = dot(ray.direction, ray.direction);
    double b = 2.0 * dot(ray.direction, s2r);
    double c = dot(s2r, s2r) - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return {};
    double t1 = (-b - std::sqrt(disc)) / (2.0 * a);
    double t2 = (-b + std::sqrt(disc)) / (2.0 * a);
    return {t1, t2};
}
std::vector<double> plane_intersect(const Ray& ray) {            // plane.rs:16-23
    count(ORA_EV_PLANE);
    if (std::fabs(ray.direction.y) < 1e-12) return {};
    return {-ray.origin.y / ray.direction.y};
}
std::vector<double> cube_intersect(const Ray& ray) {             // cube.rs:18-31
    count(ORA_EV_CUBE);
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(ray.origin.x, ray.direction.x, -1.0, 1.0, xtmin, xtmax);
    check_axis(ray.origin.y, ray.direction.y, -1.0, 1.0, ytmin, ytmax);
    check_axis(ray.origin.z, ray.direction.z, -1.0, 1.0, ztmin, ztmax);
    double tmin = rmax(rmax(xtmin, ytmin), ztmin);
    double tmax = rmin(rmin(xtmax, ytmax), ztmax);
    if (tmin <= tmax) return {tmin, tmax};
    return {};
}
std::vector<double> local_intersect(int type, const Ray& r) {    // shapes/mod.rs:48-51,63-69
    switch (type) {
        case ORA_SPHERE: return sphere_intersect(r);
        case ORA_PLANE:  return plane_intersect(r);
        default:         return cube_intersect(r);
    }
}
V3 local_normal_at(int type, V3 p) {
    switch (type) {
        case ORA_SPHERE: { V3 z = {0.0, 0.0, 0.0}; return p - z; }    // sphere.rs:34-36
        case ORA_PLANE:  return {0.0, 1.0, 0.0};                      // plane.rs:25-27
        default: {                                                    // cube.rs:33-47
            double maxc = rmax(rmax(std::fabs(p.x), std::fabs(p.y)), std::fabs(p.z));
            if (maxc == std::fabs(p.x)) return {p.x, 0.0, 0.0};
            if (maxc == std::fabs(p.y)) return {0.0, p.y, 0.0};
            return {0.0, 0.0, p.z};
        }
    }
}
Bounds local_bounds(int type) {
    if (type == ORA_PLANE) return {{-INF, 0.0, -INF}, {INF, 0.0, INF}};   // plane.rs:29-33
    return {{-1.0, -1.0, -1.0}, {1.0, 1.0, 1.0}};                        // sphere.rs:38-40, cube.rs:49-51
}

// ---------------------------------------------------------------- patterns.rs / material.rs
inline int64_t rust_f64_as_i64(double x) {       // saturating `as i64`, NaN -> 0
    if (!(x == x)) return 0;
    if (x >= 9223372036854775808.0) return INT64_MAX;
    if (x <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)x;
}
Color pattern_local_color_at(const ora_material& m, V3 p) {      // patterns.rs:84-160
    Color a = {m.color_a[0], m.color_a[1], m.color_a[2]};
    Color b = {m.color_b[0], m.color_b[1], m.color_b[2]};
    switch (m.pattern_kind) {
        case ORA_PAT_PLAIN: return a;
        case ORA_PAT_STRIPES:
            return (rust_f64_as_i64(std::floor(p.x)) % 2 == 0) ? a : b;
        case ORA_PAT_RINGS:
            return (rust_f64_as_i64(std::floor(std::sqrt(p.x * p.x + p.z * p.z))) % 2 == 0) ? a : b;
        case ORA_PAT_CHECKERS: {
            uint64_t s = 0;   // i64 sum (wrapping, as a release build does)
            s += (uint64_t)rust_f64_as_i64(std::floor(p.x));
            s += (uint64_t)rust_f64_as_i64(std::floor(p.y));
            s += (uint64_t)rust_f64_as_i64(std::floor(p.z));
            return ((int64_t)s % 2 == 0) ? a : b;
        }
        case ORA_PAT_GRADIENT: {
            Color distance = b - a;
            double fraction = p.x - std::floor(p.x);
            return a + distance * fraction;
        }
        default: return {p.x, p.y, p.z};                          // TestPattern
    }
}
inline bool near_int(double v) { return std::fabs(v - std::round(v)) < TIE_EPS; }
void pattern_tie_check(const ora_material& m, V3 p) {
    if (!g_cnt) return;
    bool t = false;
    switch (m.pattern_kind) {
        case ORA_PAT_STRIPES: t = near_int(p.x); break;
        case ORA_PAT_RINGS: t = near_int(std::sqrt(p.x * p.x + p.z * p.z)); break;
        case ORA_PAT_CHECKERS: t = near_int(p.x) || near_int(p.y) || near_int(p.z); break;
        default: break;
    }
    if (t) g_cnt->tie = true;
}

struct PointLight { V3 position; Color intensity; };             // light.rs:8-11

Color shader_lighting(const ora_material& s, Color color, const PointLight& light,
                      V3 pos, V3 eyev, V3 normalv, bool in_shadow) {   // material.rs:54-81
    count(ORA_EV_LIGHT);
    Color effective = color * light.intensity;
    V3 lightv = normalize(light.position - pos);
    Color ambient = effective * s.ambient;
    Color diffuse, specular;
    double ldn = dot(lightv, normalv);
    if (in_shadow || ldn < 0.0) {
        diffuse = BLACK; specular = BLACK;
    } else {
        diffuse = effective * s.diffuse * ldn;
        V3 reflectv = reflect(neg(lightv), normalv);
        double rde = dot(reflectv, eyev);
        if (rde <= 0.0) {
            specular = BLACK;
        } else {
            count(ORA_EV_POW);
            double factor = std::pow(rde, s.shininess);
            specular = light.intensity * s.specular * factor;
        }
    }
    return ambient + diffuse + specular;
}

// ---------------------------------------------------------------- intersections.rs
struct Intersection { double t; int shape; };
inline bool same(const Intersection& a, const Intersection& b) { return a.t == b.t && a.shape == b.shape; }
typedef std::vector<Intersection> Intersections;

void sort_xs(Intersections& xs) {    // sort_by(partial_cmp().unwrap()): stable; NaN would panic
    for (auto& i : xs) if (!(i.t == i.t)) { if (g_cnt) g_cnt->tie = true; }
    std::stable_sort(xs.begin(), xs.end(),
                     [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
}
int hit_index(const Intersections& xs) {                        // intersections.rs:37-39
    for (size_t i = 0; i < xs.size(); i++) if (xs[i].t >= 0.0) return (int)i;
    return -1;
}
// intersections.rs:41-47 — slice::binary_search_by as in Rust std 1.52..1.81
void add_intersection(Intersections& xs, Intersection x) {
    size_t size = xs.size(), left = 0, right = size;
    size_t pos = 0; bool found = false;
    while (left < right) {
        size_t mid = left + size / 2;
        if (xs[mid].t < x.t) left = mid + 1;
        else if (xs[mid].t > x.t) right = mid;
        else { pos = mid; found = true; break; }
        size = right - left;
    }
    if (!found) pos = left;
    else if (g_cnt && xs[pos].shape != x.shape) g_cnt->tie = true;  // equal-t order is std-version specific
    xs.insert(xs.begin() + pos, x);
}
void add_intersections(Intersections& xs, int shape, const std::vector<double>& ts) {   // :49-54
    for (double t : ts) xs.push_back({t, shape});
    sort_xs(xs);
}
void add_intersections_from(Intersections& xs, const Intersections& o) {                // :56-59
    xs.insert(xs.end(), o.begin(), o.end());
    sort_xs(xs);
}

// ---------------------------------------------------------------- csg.rs
bool intersection_allowed(int op, bool lhit, bool inl, bool inr) {   // csg.rs:123-136
    switch (op) {
        case ORA_UNION:        return (lhit && !inr) || (!lhit && !inl);
        case ORA_INTERSECTION: return (lhit && inr) || (!lhit && inl);
        default:               return (lhit && !inr) || (!lhit && inl);
    }
}
Intersections filter_intersections(int op, const std::vector<int>& left_indices,
                                   const Intersections& xs) {         // csg.rs:94-121
    bool inl = false, inr = false;
    Intersections result;
    for (const Intersection& i : xs) {
        count(ORA_EV_CSG_ENTRIES);
        bool lhit = std::binary_search(left_indices.begin(), left_indices.end(), i.shape);
        if (intersection_allowed(op, lhit, inl, inr)) add_intersection(result, i);
        if (lhit) inl = !inl; else inr = !inr;
    }
    return result;
}

}  // namespace

// ---------------------------------------------------------------- object_store.rs / world.rs
struct ObjRef { int kind, index; };          // ObjectIndex (object_store.rs:17-21)
struct Parent { int kind, index; };          // kind -1 = None (object_store.rs:56-60)

struct ora_world {
    // ObjectStore (object_store.rs:185-192)
    std::vector<int> shape_type;
    std::vector<ora_material> shape_material;
    std::vector<std::vector<ObjRef>> group_children;    // group.rs:17
    std::vector<Bounds> group_bounds;                   // group.rs:18
    std::vector<int> csg_op;                            // csg.rs:30
    std::vector<std::pair<ObjRef, ObjRef>> csg_children;
    std::vector<Bounds> csg_bounds;
    std::vector<Parent> group_parent, csg_parent, shape_parent;   // object_store.rs:74-78
    std::vector<M4> group_transform, csg_transform, shape_transform;   // :125-129
    // World (world.rs:81-85)
    std::vector<ObjRef> scene;
    std::vector<PointLight> lights;

    const M4& transform_of(ObjRef o) const {
        switch (o.kind) {
            case ORA_OBJ_GROUP: return group_transform[o.index];
            case ORA_OBJ_CSG:   return csg_transform[o.index];
            default:            return shape_transform[o.index];
        }
    }
    Parent parent_of(ObjRef o) const {
        switch (o.kind) {
            case ORA_OBJ_GROUP: return group_parent[o.index];
            case ORA_OBJ_CSG:   return csg_parent[o.index];
            default:            return shape_parent[o.index];
        }
    }
    void set_parent(ObjRef o, Parent p) {
        switch (o.kind) {
            case ORA_OBJ_GROUP: group_parent[o.index] = p; break;
            case ORA_OBJ_CSG:   csg_parent[o.index] = p; break;
            default:            shape_parent[o.index] = p; break;
        }
    }

    // object_store.rs:243-260
    Intersections intersect(ObjRef i, const Ray& ray) const {
        count(ORA_EV_XFORM);
        Ray local_ray = transform(ray, inverse(transform_of(i)));
        switch (i.kind) {
            case ORA_OBJ_SHAPE: {
                Intersections xs;
                add_intersections(xs, i.index, local_intersect(shape_type[i.index], local_ray));
                return xs;
            }
            case ORA_OBJ_GROUP: return group_local_intersect(i.index, local_ray);
            default:            return csg_local_intersect(i.index, local_ray);
        }
    }
    // group.rs:64-75
    Intersections group_local_intersect(int g, const Ray& ray) const {
        Intersections xs;
        count(ORA_EV_BOUNDS);
        if (!bounds_intersects(group_bounds[g], ray)) return xs;
        for (const ObjRef& child : group_children[g]) add_intersections_from(xs, intersect(child, ray));
        return xs;
    }
    // csg.rs:73-91
    Intersections csg_local_intersect(int c, const Ray& ray) const {
        count(ORA_EV_BOUNDS);
        if (!bounds_intersects(csg_bounds[c], ray)) return {};
        ObjRef left = csg_children[c].first, right = csg_children[c].second;
        Intersections xs = intersect(left, ray);
        Intersections rxs = intersect(right, ray);
        std::vector<int> left_shapes;
        for (const auto& i : xs) left_shapes.push_back(i.shape);
        std::sort(left_shapes.begin(), left_shapes.end());
        add_intersections_from(xs, rxs);
        return filter_intersections(csg_op[c], left_shapes, xs);
    }
    // object_store.rs:309-320
    V3 world_to_object(ObjRef obj, V3 p) const {
        Parent par = parent_of(obj);
        if (par.kind >= 0) p = world_to_object(ObjRef{par.kind, par.index}, p);
        return mul_point(inverse(transform_of(obj)), p);
    }
    // object_store.rs:331-349
    V3 normal_to_world(ObjRef obj, V3 n) const {
        M4 inv = inverse(transform_of(obj));
        M4 it = identity();                      // fixed_slice<3,3>.transpose().to_homogeneous()
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) at(it, r, c) = at(inv, c, r);
        V3 wn = normalize(mul_vector(it, n));
        Parent par = parent_of(obj);
        if (par.kind >= 0) return normal_to_world(ObjRef{par.kind, par.index}, wn);
        return wn;
    }
    // object_store.rs:322-329
    V3 normal_at(int shp, V3 world_point) const {
        ObjRef obj = {ORA_OBJ_SHAPE, shp};
        if (g_cnt) { int lv = 1; Parent p = parent_of(obj); while (p.kind >= 0) { lv++; p = parent_of(ObjRef{p.kind, p.index}); } count(ORA_EV_LEVELS, lv); }
        V3 lp = world_to_object(obj, world_point);
        V3 ln = local_normal_at(shape_type[shp], lp);
        return normal_to_world(obj, ln);
    }
    // object_store.rs:293-299
    Color color_at_object(int shp, V3 world_point) const {
        const ora_material& m = shape_material[shp];
        V3 op = world_to_object(ObjRef{ORA_OBJ_SHAPE, shp}, world_point);
        M4 pt; std::memcpy(pt.m, m.pattern_transform, sizeof pt.m);
        V3 pp = mul_point(inverse(pt), op);
        pattern_tie_check(m, pp);
        return pattern_local_color_at(m, pp);
    }
    // object_store.rs:301-307
    Color lighting(int shp, const PointLight& l, V3 pos, V3 eyev, V3 normalv, bool in_shadow) const {
        Color c = color_at_object(shp, pos);
        return shader_lighting(shape_material[shp], c, l, pos, eyev, normalv, in_shadow);
    }
    // object_store.rs:360-389
    Bounds set_bounds_of(ObjRef obj) {
        switch (obj.kind) {
            case ORA_OBJ_GROUP: {
                Bounds b = bounds_empty();
                for (size_t k = 0; k < group_children[obj.index].size(); k++) {
                    ObjRef child = group_children[obj.index][k];
                    Bounds cb = bounds_to_global(set_bounds_of(child), transform_of(child));
                    bounds_merge(b, cb);
                }
                group_bounds[obj.index] = b;
                return b;
            }
            case ORA_OBJ_CSG: {
                Bounds b = bounds_empty();
                ObjRef l = csg_children[obj.index].first, r = csg_children[obj.index].second;
                Bounds lb = bounds_to_global(set_bounds_of(l), transform_of(l));
                bounds_merge(b, lb);
                Bounds rb = bounds_to_global(set_bounds_of(r), transform_of(r));
                bounds_merge(b, rb);
                csg_bounds[obj.index] = b;
                return b;
            }
            default: return local_bounds(shape_type[obj.index]);
        }
    }

    // world.rs:111-118
    Intersections world_intersect(const Ray& ray) const {
        Intersections xs;
        for (const ObjRef& i : scene) add_intersections_from(xs, intersect(i, ray));
        return xs;
    }
    struct Comps {                                // world.rs:210-222
        double t; int shape; V3 point, over_point, under_point, eyev, normalv, reflectv;
        double n1, n2; bool inside;
    };
    // world.rs:258-296
    void refractive_indices_at(const Intersection& hit, const Intersections& xs, double& n1, double& n2) const {
        std::vector<int> visited;
        for (const Intersection& i : xs) {
            if (same(i, hit)) break;              // take_while(|&i| i != hit)
            auto pos = std::find(visited.begin(), visited.end(), i.shape);
            if (pos != visited.end()) visited.erase(pos); else visited.push_back(i.shape);
        }
        auto ri = [&](int s) { return shape_material[s].refractive_index; };
        if (!visited.empty()) {
            n1 = ri(visited.back());
            auto pos = std::find(visited.begin(), visited.end(), hit.shape);
            if (pos != visited.end()) {
                visited.erase(pos);
                n2 = visited.empty() ? 1.0 : ri(visited.back());
            } else {
                n2 = ri(hit.shape);
            }
        } else {
            n1 = 1.0;
            n2 = ri(hit.shape);
        }
    }
    // world.rs:225-256
    Comps prepare(const Intersection& hit, const Ray& ray, const Intersections& xs) const {
        Comps c;
        c.point = position(ray, hit.t);
        c.shape = hit.shape;
        c.eyev = neg(ray.direction);
        c.normalv = normal_at(hit.shape, c.point);
        if (dot(c.normalv, c.eyev) < 0.0) { c.inside = true; c.normalv = neg(c.normalv); }
        else c.inside = false;
        const double eps = 1e-10;                 // world.rs:298-300
        c.over_point = c.point + c.normalv * eps;
        c.under_point = c.point - c.normalv * eps;
        refractive_indices_at(hit, xs, c.n1, c.n2);
        c.t = hit.t;
        c.reflectv = reflect(ray.direction, c.normalv);
        return c;
    }
    static double schlick(const Comps& c) {       // world.rs:304-323
        count(ORA_EV_SCHLICK);
        double cosv = dot(c.eyev, c.normalv);
        if (c.n1 > c.n2) {
            double n = c.n1 / c.n2;
            double sin2_t = n * n * (1.0 - cosv * cosv);
            if (sin2_t > 1.0) return 1.0;
            cosv = std::sqrt(1.0 - sin2_t);
        }
        double q = (c.n1 - c.n2) / (c.n1 + c.n2);
        double r0 = q * q;                        // powi(2)
        double x = 1.0 - cosv;
        double x2 = x * x;                        // powi(5) = llvm.powi: x * (x^2)^2
        double x5 = x * (x2 * x2);
        return r0 + (1.0 - r0) * x5;
    }
    // world.rs:195-207
    bool is_shadowed(V3 point, const PointLight& light) const {
        if (g_cnt) g_cnt->rays_shadow++;
        V3 v = light.position - point;
        double distance = norm(v);
        V3 direction = normalize(v);
        Ray ray = {point, direction};
        Intersections xs = world_intersect(ray);
        int h = hit_index(xs);
        if (h < 0) return false;
        if (g_cnt && std::fabs(xs[h].t - distance) <= TIE_EPS * std::fmax(1.0, distance)) g_cnt->tie = true;
        tie_check_hit(xs, h, false);
        return xs[h].t < distance;
    }
    void tie_check_hit(const Intersections& xs, int h, bool primary) const {
        if (!g_cnt) return;
        bool t = false;
        double th = xs[h].t;
        if (th <= TIE_ZERO) t = true;                                      // could flip sign
        if (h > 0 && xs[h - 1].t >= -TIE_ZERO) t = true;
        for (size_t k = h + 1; k < xs.size(); k++) {
            if (xs[k].t - th > TIE_EPS * std::fmax(1.0, std::fabs(th))) break;
            if (xs[k].shape != xs[h].shape) { t = true; break; }
        }
        if (t) { g_cnt->tie = true; if (primary) g_cnt->tie_id = true; }
    }
    // world.rs:151-163
    Color reflected_color(const Comps& c, int remaining) const {
        double reflectivity = shape_material[c.shape].reflective;
        if (remaining == 0 || reflectivity == 0.0) return BLACK;
        Ray rr = {c.over_point, c.reflectv};
        Color col = color_at(rr, remaining - 1, nullptr);
        return col * reflectivity;
    }
    // world.rs:165-193
    Color refracted_color(const Comps& c, int remaining) const {
        if (remaining == 0 || shape_material[c.shape].transparency == 0.0) return BLACK;
        count(ORA_EV_REFR);
        double n_ratio = c.n1 / c.n2;
        double cos_i = dot(c.eyev, c.normalv);
        double sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
        if (g_cnt && std::fabs(sin2_t - 1.0) < TIE_EPS) g_cnt->tie = true;
        if (sin2_t > 1.0) return BLACK;
        double cos_t = std::sqrt(1.0 - sin2_t);
        V3 direction = c.normalv * (n_ratio * cos_i - cos_t) - c.eyev * n_ratio;
        Ray rr = {c.under_point, direction};
        return color_at(rr, remaining - 1, nullptr) * shape_material[c.shape].transparency;
    }
    // world.rs:127-149
    Color shade_hit(const Comps& c, int remaining) const {
        count(ORA_EV_HIT);
        Color surface = BLACK;                    // Sum: fold(black, +)
        for (const PointLight& l : lights) {
            bool sh = is_shadowed(c.over_point, l);
            surface = surface + lighting(c.shape, l, c.point, c.eyev, c.normalv, sh);
        }
        Color reflected = reflected_color(c, remaining);
        Color refracted = refracted_color(c, remaining);
        const ora_material& s = shape_material[c.shape];
        if (s.reflective > 0.0 && s.transparency > 0.0) {
            double reflectance = schlick(c);
            return surface + reflected * reflectance + refracted * (1.0 - reflectance);
        }
        return surface + reflected + refracted;
    }
    // world.rs:100-109
    Color color_at(const Ray& ray, int remaining, int* hit_shape) const {
        if (g_cnt) g_cnt->rays_radiance++;
        Intersections xs = world_intersect(ray);
        int h = hit_index(xs);
        if (hit_shape) *hit_shape = (h >= 0) ? xs[h].shape : -1;
        if (h < 0) {
            if (g_cnt && !xs.empty() && xs.back().t >= -TIE_ZERO) { g_cnt->tie = true; if (hit_shape) g_cnt->tie_id = true; }
            return BLACK;
        }
        tie_check_hit(xs, h, hit_shape != nullptr);
        Comps c = prepare(xs[h], ray, xs);
        return shade_hit(c, remaining);
    }
};

namespace {
inline V3 v3(const double* p) { return {p[0], p[1], p[2]}; }
inline void put(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
inline void putc(double* o, Color c) { o[0] = c.r; o[1] = c.g; o[2] = c.b; }
inline M4 m4(const double* p) { M4 r; std::memcpy(r.m, p, sizeof r.m); return r; }
inline void putm(double* o, const M4& m) { std::memcpy(o, m.m, sizeof m.m); }
inline void putb(double* o, const Bounds& b) {
    o[0] = b.min.x; o[1] = b.min.y; o[2] = b.min.z; o[3] = b.max.x; o[4] = b.max.y; o[5] = b.max.z;
}
ora_world::Comps from_c(const ora_comps* c) {
    ora_world::Comps r;
    r.t = c->t; r.shape = c->shape_index; r.inside = c->inside != 0;
    r.point = v3(c->point); r.over_point = v3(c->over_point); r.under_point = v3(c->under_point);
    r.eyev = v3(c->eyev); r.normalv = v3(c->normalv); r.reflectv = v3(c->reflectv);
    r.n1 = c->n1; r.n2 = c->n2;
    return r;
}

// camera.rs:56-69
Ray ray_for_pixel(const ora_camera& cam, int x, int y) {
    double xoffset = ((double)x + 0.5) * cam.pixel_size;
    double yoffset = ((double)y + 0.5) * cam.pixel_size;
    double world_x = cam.half_width - xoffset;
    double world_y = cam.half_height - yoffset;
    M4 t = m4(cam.transform);
    V3 pixel = mul_point(inverse(t), V3{world_x, world_y, -1.0});
    V3 origin = mul_point(inverse(t), V3{0.0, 0.0, 0.0});
    V3 direction = normalize(pixel - origin);
    return {origin, direction};
}
}  // namespace

extern "C" {

void ora_identity(double* o) { putm(o, identity()); }
void ora_translation(double x, double y, double z, double* o) { putm(o, translation(V3{x, y, z})); }
void ora_scaling(double x, double y, double z, double* o) { putm(o, scaling(x, y, z)); }
void ora_rotation_x(double a, double* o) { putm(o, rotation_around_axis(V3{1.0, 0.0, 0.0}, a)); }   // geometry.rs:33-35
void ora_rotation_y(double a, double* o) { putm(o, rotation_around_axis(V3{0.0, 1.0, 0.0}, a)); }
void ora_rotation_z(double a, double* o) { putm(o, rotation_around_axis(V3{0.0, 0.0, 1.0}, a)); }
void ora_view_transform(const double* f, const double* t, const double* u, double* o) {
    putm(o, view_transform(v3(f), v3(t), v3(u)));
}
void ora_mat_mul(const double* a, const double* b, double* o) { putm(o, mat_mul(m4(a), m4(b))); }
void ora_inverse(const double* m, double* o) { putm(o, inverse(m4(m))); }
void ora_mul_point(const double* m, const double* p, double* o) { put(o, mul_point(m4(m), v3(p))); }
void ora_mul_vector(const double* m, const double* p, double* o) { put(o, mul_vector(m4(m), v3(p))); }
void ora_reflect(const double* v, const double* n, double* o) { put(o, reflect(v3(v), v3(n))); }

void ora_default_material(ora_material* m) {          // material.rs:17-22,41-51
    std::memset(m, 0, sizeof *m);
    m->ambient = 0.1; m->diffuse = 0.9; m->specular = 0.9; m->shininess = 200.0;
    m->reflective = 0.0; m->transparency = 0.0; m->refractive_index = 1.0;
    m->color_a[0] = m->color_a[1] = m->color_a[2] = 1.0;
    m->pattern_kind = ORA_PAT_PLAIN;
    putm(m->pattern_transform, identity());
}

ora_world* ora_world_new(void) { return new ora_world(); }
void ora_world_free(ora_world* w) { delete w; }
int ora_add_shape(ora_world* w, int type, const ora_material* m, const double* t) {   // object_store.rs:211-217
    ora_material mm; if (m) mm = *m; else ora_default_material(&mm);
    w->shape_parent.push_back({-1, 0});
    w->shape_transform.push_back(m4(t));
    w->shape_type.push_back(type);
    w->shape_material.push_back(mm);
    return (int)w->shape_type.size() - 1;
}
int ora_add_group(ora_world* w, const double* t) {                                    // :219-223, group.rs:37-42
    w->group_parent.push_back({-1, 0});
    w->group_transform.push_back(m4(t));
    w->group_children.push_back({});
    w->group_bounds.push_back(bounds_empty());
    return (int)w->group_bounds.size() - 1;
}
void ora_set_group_of(ora_world* w, int k, int i, int g) {                            // :238-241
    w->group_children[g].push_back({k, i});
    w->set_parent({k, i}, {ORA_OBJ_GROUP, g});
}
int ora_add_csg(ora_world* w, int op, int lk, int li, int rk, int ri, const double* t) {   // :225-236, csg.rs:52-59
    w->csg_parent.push_back({-1, 0});
    w->csg_transform.push_back(m4(t));
    w->csg_op.push_back(op);
    w->csg_children.push_back({{lk, li}, {rk, ri}});
    w->csg_bounds.push_back(bounds_empty());
    int c = (int)w->csg_bounds.size() - 1;
    w->set_parent({lk, li}, {ORA_OBJ_CSG, c});
    w->set_parent({rk, ri}, {ORA_OBJ_CSG, c});
    return c;
}
void ora_add_object_to_scene(ora_world* w, int k, int i) { w->scene.push_back({k, i}); }   // world.rs:67-69
int ora_add_light(ora_world* w, const double* p, const double* c) {                       // world.rs:71-74
    w->lights.push_back({v3(p), {c[0], c[1], c[2]}});
    return (int)w->lights.size() - 1;
}
void ora_build(ora_world* w) { for (size_t i = 0; i < w->scene.size(); i++) w->set_bounds_of(w->scene[i]); }   // world.rs:39-43
void ora_set_material_of(ora_world* w, int s, const ora_material* m) { w->shape_material[s] = *m; }
void ora_get_material_of(ora_world* w, int s, ora_material* m) { *m = w->shape_material[s]; }
void ora_set_light(ora_world* w, int i, const double* p, const double* c) { w->lights[i] = {v3(p), {c[0], c[1], c[2]}}; }
void ora_set_bounds_of(ora_world* w, int k, int i, double* o) { Bounds b = w->set_bounds_of({k, i}); if (o) putb(o, b); }
void ora_get_bounds(ora_world* w, int k, int i, double* o) {
    if (k == ORA_OBJ_GROUP) putb(o, w->group_bounds[i]);
    else if (k == ORA_OBJ_CSG) putb(o, w->csg_bounds[i]);
    else putb(o, local_bounds(w->shape_type[i]));
}
int ora_parent_of(ora_world* w, int k, int i, int* pi) { Parent p = w->parent_of({k, i}); if (pi) *pi = p.index; return p.kind; }

int ora_local_intersect(int type, const double* o, const double* d, double* out) {
    std::vector<double> ts = local_intersect(type, Ray{v3(o), v3(d)});
    for (size_t i = 0; i < ts.size(); i++) out[i] = ts[i];
    return (int)ts.size();
}
void ora_local_normal_at(int type, const double* p, double* o) { put(o, local_normal_at(type, v3(p))); }
int ora_bounds_intersects(const double* b, const double* o, const double* d) {
    Bounds bb = {{b[0], b[1], b[2]}, {b[3], b[4], b[5]}};
    return bounds_intersects(bb, Ray{v3(o), v3(d)}) ? 1 : 0;
}
int ora_intersection_allowed(int op, int lhit, int inl, int inr) { return intersection_allowed(op, lhit, inl, inr) ? 1 : 0; }
int ora_filter_intersections(int op, const int32_t* left, int nleft, const double* t, const int32_t* s, int n,
                             double* ot, int32_t* os) {
    std::vector<int> l(left, left + nleft);
    Intersections xs; for (int i = 0; i < n; i++) xs.push_back({t[i], s[i]});
    sort_xs(xs);                                   // Intersections::from_vec
    Intersections r = filter_intersections(op, l, xs);
    for (size_t i = 0; i < r.size(); i++) { ot[i] = r[i].t; os[i] = r[i].shape; }
    return (int)r.size();
}
static int copy_xs(const Intersections& xs, double* ot, int32_t* os, int cap) {
    int n = (int)xs.size();
    for (int i = 0; i < n && i < cap; i++) { if (ot) ot[i] = xs[i].t; if (os) os[i] = xs[i].shape; }
    return n;
}
int ora_object_intersect(ora_world* w, int k, int i, const double* o, const double* d, double* ot, int32_t* os, int cap) {
    return copy_xs(w->intersect({k, i}, Ray{v3(o), v3(d)}), ot, os, cap);
}
int ora_world_intersect(ora_world* w, const double* o, const double* d, double* ot, int32_t* os, int cap) {
    return copy_xs(w->world_intersect(Ray{v3(o), v3(d)}), ot, os, cap);
}
int ora_hit(const double* t, int n) { for (int i = 0; i < n; i++) if (t[i] >= 0.0) return i; return -1; }
void ora_color_at(ora_world* w, const double* o, const double* d, int remaining, double* out) {
    putc(out, w->color_at(Ray{v3(o), v3(d)}, remaining, nullptr));
}
int ora_is_shadowed(ora_world* w, const double* p, int light) { return w->is_shadowed(v3(p), w->lights[light]) ? 1 : 0; }
void ora_prepare_computations(ora_world* w, int hit, const double* o, const double* d,
                              const double* t, const int32_t* s, int n, ora_comps* out) {
    Intersections xs; for (int i = 0; i < n; i++) xs.push_back({t[i], s[i]});
    ora_world::Comps c = w->prepare(xs[hit], Ray{v3(o), v3(d)}, xs);
    out->t = c.t; out->shape_index = c.shape; out->inside = c.inside ? 1 : 0;
    put(out->point, c.point); put(out->over_point, c.over_point); put(out->under_point, c.under_point);
    put(out->eyev, c.eyev); put(out->normalv, c.normalv); put(out->reflectv, c.reflectv);
    out->n1 = c.n1; out->n2 = c.n2;
}
void ora_shade_hit(ora_world* w, const ora_comps* c, int remaining, double* out) { putc(out, w->shade_hit(from_c(c), remaining)); }
void ora_reflected_color(ora_world* w, const ora_comps* c, int remaining, double* out) { putc(out, w->reflected_color(from_c(c), remaining)); }
void ora_refracted_color(ora_world* w, const ora_comps* c, int remaining, double* out) { putc(out, w->refracted_color(from_c(c), remaining)); }
double ora_schlick(const ora_comps* c) { return ora_world::schlick(from_c(c)); }
void ora_world_to_object(ora_world* w, int k, int i, const double* p, double* o) { put(o, w->world_to_object({k, i}, v3(p))); }
void ora_normal_to_world(ora_world* w, int k, int i, const double* n, double* o) { put(o, w->normal_to_world({k, i}, v3(n))); }
void ora_normal_at(ora_world* w, int s, const double* p, double* o) { put(o, w->normal_at(s, v3(p))); }
void ora_pattern_color_at(const ora_material* m, const double* p, double* o) { putc(o, pattern_local_color_at(*m, v3(p))); }
void ora_object_color_at(ora_world* w, int s, const double* p, double* o) { putc(o, w->color_at_object(s, v3(p))); }
void ora_shader_lighting(const ora_material* m, const double* c, const double* lp, const double* li,
                         const double* pos, const double* eyev, const double* nv, int sh, double* out) {
    PointLight l = {v3(lp), {li[0], li[1], li[2]}};
    putc(out, shader_lighting(*m, Color{c[0], c[1], c[2]}, l, v3(pos), v3(eyev), v3(nv), sh != 0));
}
void ora_object_lighting(ora_world* w, int s, int light, const double* pos, const double* eyev,
                         const double* nv, int sh, double* out) {
    putc(out, w->lighting(s, w->lights[light], v3(pos), v3(eyev), v3(nv), sh != 0));
}
void ora_color_to_rgb(const double* c, uint8_t* o) {            // color.rs:40-48
    o[0] = discretize(rust_clamp01(c[0])); o[1] = discretize(rust_clamp01(c[1])); o[2] = discretize(rust_clamp01(c[2]));
}

void ora_camera_new(int hsize, int vsize, double fov, ora_camera* c) {   // camera.rs:20-37
    double half_view = std::tan(fov / 2.0);
    double aspect = (double)hsize / (double)vsize;
    double hw, hh;
    if (aspect >= 1.0) { hw = half_view; hh = half_view / aspect; }
    else { hw = half_view * aspect; hh = half_view; }
    c->hsize = hsize; c->vsize = vsize;
    c->pixel_size = (hw * 2.0) / (double)hsize;
    c->half_width = hw; c->half_height = hh;
    putm(c->transform, identity());
}
void ora_ray_for_pixel(const ora_camera* c, int x, int y, double* o, double* d) {
    Ray r = ray_for_pixel(*c, x, y); put(o, r.origin); put(d, r.direction);
}

int ora_render(ora_world* w, const ora_camera* cam, int max_depth,
               const int32_t* xs, const int32_t* ys, int64_t n,
               uint8_t* rgb, int32_t* hit_id, double* color, uint8_t* tie,
               ora_stats* stats, int threads) {
    const bool whole = (xs == nullptr);
    const int64_t total = whole ? (int64_t)cam->hsize * cam->vsize : n;
    if (threads < 1) threads = 1;
    std::vector<Counters> cnts(threads);
    for (auto& c : cnts) std::memset(&c, 0, sizeof c);
    std::vector<uint64_t> ties(threads, 0);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int tid) {
        Counters& c = cnts[tid];
        g_cnt = &c;
        // camera.rs:45-51: for y { for x { ... } }  (threads>1 interleaves pixels; results are per-pixel independent)
        for (int64_t i = tid; i < total; i += threads) {
            int x = whole ? (int)(i % cam->hsize) : xs[i];
            int y = whole ? (int)(i / cam->hsize) : ys[i];
            c.tie = false; c.tie_id = false;
            Ray ray = ray_for_pixel(*cam, x, y);
            int hs = -1;
            Color col = w->color_at(ray, max_depth, &hs);
            if (color) { color[3 * i] = col.r; color[3 * i + 1] = col.g; color[3 * i + 2] = col.b; }
            if (rgb) { double cc[3] = {col.r, col.g, col.b}; ora_color_to_rgb(cc, rgb + 3 * i); }
            if (hit_id) hit_id[i] = hs;
            if (tie) tie[i] = (uint8_t)((c.tie ? 1 : 0) | (c.tie_id ? 2 : 0));
            if (c.tie) ties[tid]++;
        }
        g_cnt = nullptr;
    };
    if (threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        uint64_t rad = 0;
        for (int t = 0; t < threads; t++) {
            for (int e = 0; e < ORA_EV_COUNT; e++) stats->ev[e] += cnts[t].ev[e];
            rad += cnts[t].rays_radiance; stats->rays_shadow += cnts[t].rays_shadow;
            stats->tie_pixels += ties[t];
        }
        stats->rays_primary = (uint64_t)total;
        stats->rays_secondary = rad - (uint64_t)total;
        stats->ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    }
    return 0;
}

}  // extern "C"
