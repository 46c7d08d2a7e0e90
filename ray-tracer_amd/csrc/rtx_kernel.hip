// rtx_kernel.hip — trace_shade: the MI355X (gfx950, wave64) kernel behind rtx_render().
//
// One thread per pixel, one wavefront per 8x8 tile (block = 64 threads, no barriers).
// Replaces the body of the pixel loop of Camera::render (camera.rs:45-51):
//   ray_for_pixel (camera.rs:56-69) -> World::color_at (world.rs:100-109)
//   -> intersect / hit (world.rs:111-118, intersections.rs:37-39)
//   -> PreparedComputations (world.rs:225-296) -> shade_hit (world.rs:127-149)
//   -> is_shadowed (:195-207), lighting (material.rs:54-81), patterns (patterns.rs:84-160),
//      reflected/refracted_color (:151-193), schlick (:304-323) -> Color::to_rgb (color.rs:40-48).
//
// Design:
//  * all geometry and shading arithmetic f64 in the reference's operation order; built with
//    -ffp-contract=off (Rust never fuses a*b+c), IEEE division and sqrt (one exception: rtx_pow.h).
//  * outside CSG the wave walks the scene's child lists together (rtx_device.h: ListChunk/ContRec):
//    per chunk one packed-f32 bounding-sphere sweep for all lanes, then per-lane exact f64 tests of
//    each lane's own candidates, then only the containers some lane may hit.  Under a CSG root an op
//    interpreter walks the ops in program order (hit lists need the reference's order).
//  * no hit lists outside CSG subtrees: closest hit / shadow / refractive containers are O(1)
//    running state that is provably equal to the reference's sorted Vec.
//    Under a CSG root a bounded per-lane list (RTX_MAX_CSG_ENTRIES, in LDS) is built, stably merged
//    and filtered exactly like csg.rs:73-136.
//  * recursion (world.rs:160,190) becomes a per-lane state machine with an explicit post-order
//    frame stack of max_depth frames; every lane always traces *some* ray (radiance or shadow)
//    in the shared traversal, so reflection/refraction divergence does not idle lanes.
//  * the ray being traced and the local rays of the enclosing containers live in an LDS stack
//    (n_levels x 6 x 64 doubles); the shading context that must survive a traversal is held to ~30
//    registers and the frames in scratch carry only what a parent needs when a child returns.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtx.h"
#include "rtx_device.h"
#include "rtx_pow.h"

#define DI __device__ __forceinline__
#define AS4 __attribute__((address_space(4)))

namespace {

struct V3 { double x, y, z; };
DI V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DI V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DI V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
DI V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
DI V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
// nalgebra 3-vector dot: (a0*b0 + a1*b1) + a2*b2
DI double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DI double norm3(V3 a) { return __builtin_sqrt(dot3(a, a)); }
DI V3 normalize3(V3 a) { return a / norm3(a); }            // unscale: component-wise division
DI V3 reflect3(V3 v, V3 n) { return v - n * 2.0 * dot3(v, n); }   // geometry.rs:49-51

#define INF __builtin_inf()
#define LDS_HALF_DOUBLES 192           // one origin OR one direction per lane: 3 x 64 doubles
#define CULL_KAPPA 1.0e-5f             // relative slack on |oc|^2 covering f32 rounding of the pre-test

// Projective3 * Point3 with an affine matrix: (M3x3 p + t) / w33   (rows m[r*4+c])
template <typename P>
DI V3 xf_point(P m, double w33, bool div, V3 p) {
    V3 r;
    r.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    r.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    r.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    if (div) r = r / w33;
    return r;
}
template <typename P>
DI V3 xf_vector(P m, V3 v) {
    V3 r;
    r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z;
    r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z;
    r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z;
    return r;
}
// transpose(inverse 3x3) * n   (object_store.rs:332-337)
template <typename P>
DI V3 xf_normal(P m, V3 n) {
    V3 r;
    r.x = m[0] * n.x + m[4] * n.y + m[8] * n.z;
    r.y = m[1] * n.x + m[5] * n.y + m[9] * n.z;
    r.z = m[2] * n.x + m[6] * n.y + m[10] * n.z;
    return r;
}

// bounds.rs:77-94 / cube.rs:54-71
DI void check_axis(double origin, double direction, double mn, double mx, double& tmin, double& tmax) {
    double tmin_n = mn - origin;
    double tmax_n = mx - origin;
    double a, b;
    if (__builtin_fabs(direction) >= 1e-12) {
        a = tmin_n / direction;
        b = tmax_n / direction;
    } else {
        a = tmin_n * INF;
        b = tmax_n * INF;
    }
    if (a > b) { double t = a; a = b; b = t; }
    tmin = a; tmax = b;
}
// slab test shared by Bounds::intersects (bounds.rs:42-54) and the cube (cube.rs:18-31)
DI bool slabs(V3 o, V3 d, double x0, double y0, double z0, double x1, double y1, double z1,
              double& tmin, double& tmax) {
    double xa, xb, ya, yb, za, zb;
    check_axis(o.x, d.x, x0, x1, xa, xb);
    check_axis(o.y, d.y, y0, y1, ya, yb);
    check_axis(o.z, d.z, z0, z1, za, zb);
    tmin = __builtin_fmax(__builtin_fmax(xa, ya), za);     // f64::max ignores NaN == fmax
    tmax = __builtin_fmin(__builtin_fmin(xb, yb), zb);
    return tmin <= tmax;
}

// Conservative interval filter for the DECISION of the slab test above (no counterpart in the reference; it
// cannot change a result).  The six distances are formed with the raw hardware reciprocal (v_rcp_f64,
// measured relative error 2^-24.4 on gfx950, assumed < 2^-21 here) instead of six IEEE divisions; L = max of
// the near distances and H = min of the far ones then carry a relative error below EPS = 2^-20 of
// M = max(|L|, |H|) (distances of larger magnitude than M cannot decide the comparison).  If L and H are
// further apart than that the exact test must agree: +1 = certainly passes, 0 = certainly fails,
// -1 = undecided (the caller then runs the exact test).  Rays with a direction component below the
// reference's 1e-12 threshold, or operands beyond 1e100, are always undecided; infinite bounds are exact.
DI int slab_filter(V3 o, V3 d, double x0, double y0, double z0, double x1, double y1, double z1) {
    const double ax = __builtin_fabs(d.x), ay = __builtin_fabs(d.y), az = __builtin_fabs(d.z);
    const double amin = __builtin_fmin(__builtin_fmin(ax, ay), az);
    const double big = __builtin_fmax(__builtin_fmax(__builtin_fmax(ax, ay), az),
                                      __builtin_fmax(__builtin_fmax(__builtin_fabs(o.x), __builtin_fabs(o.y)), __builtin_fabs(o.z)));
    const bool dom = amin >= 1e-12 && big <= 1e100;                     // false on NaN
    const double rx = __builtin_amdgcn_rcp(d.x), ry = __builtin_amdgcn_rcp(d.y), rz = __builtin_amdgcn_rcp(d.z);
    const double xa = (x0 - o.x) * rx, xb = (x1 - o.x) * rx;
    const double ya = (y0 - o.y) * ry, yb = (y1 - o.y) * ry;
    const double za = (z0 - o.z) * rz, zb = (z1 - o.z) * rz;
    const double L = __builtin_fmax(__builtin_fmax(__builtin_fmin(xa, xb), __builtin_fmin(ya, yb)), __builtin_fmin(za, zb));
    const double H = __builtin_fmin(__builtin_fmin(__builtin_fmax(xa, xb), __builtin_fmax(ya, yb)), __builtin_fmax(za, zb));
    const double aL = __builtin_fabs(L), aH = __builtin_fabs(H);
    const double M = __builtin_fmax(aL < INF ? aL : 0.0, aH < INF ? aH : 0.0);
    const double e = M * 0x1p-20 + 1e-290;
    if (!dom) return -1;
    if (L + e < H - e) return 1;
    if (L - e > H + e) return 0;
    return -1;
}

// low bit of `floor(v) as i64` (saturating cast, NaN -> 0): patterns.rs:98,115,130
DI int floor_parity(double v) {
    double f = __builtin_floor(v);
    if (!(f == f)) return 0;
    if (f >= 9223372036854775808.0) return 1;     // i64::MAX is odd
    if (f <= -9223372036854775808.0) return 0;    // i64::MIN is even
    double h = f * 0.5;
    return (h == __builtin_floor(h)) ? 0 : 1;
}

DI uint8_t quantise(double x) {                   // color.rs:32-43
    if (x < 0.0) x = 0.0; else if (x > 1.0) x = 1.0;      // f64::clamp, NaN stays
    double v = __builtin_round(x * 255.0);        // half away from zero
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)(int)v;
}

DI int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
DI unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One shade_hit (world.rs:127-149) waiting for its children.  Frames live in scratch (indexed by the
// per-lane stack pointer) and hold only what the parent needs when a child returns; the context of the
// hit being shaded right now (point, normal, eye vector) stays in registers instead.
struct Frame {           // 40 bytes
    double s[3];         // surface; once the reflected child is back: surface + reflected [* reflectance]
    double R;            // Schlick reflectance (world.rs:304-323), computed before descending
    int32_t flags;       // FR_*
    int32_t rem_mat;     // the `remaining` this shade_hit was called with (4 bits) | material of the hit << 4: the
                         //   coefficients `reflective` / `transparency` are read from the material when a child returns
};
// the refraction ray (world.rs:186-188) of a frame while its reflected child is traced: written only by hits that have BOTH
// children (kept apart from the frames so that the other hits do not pay for it)
struct PendRay { double o[3], d[3]; };
enum {
    FR_REFL = 1,         // the first child is the reflected ray (else: the only child is the refracted ray)
    FR_REFR_PENDING = 2, // a refraction ray waits in o2/d2
    FR_SECOND = 4,       // the child now being traced is that refraction ray
    FR_BOTH = 8,         // reflective > 0 && transparency > 0: Schlick mix (world.rs:142-145)
    // split tiles (SPLIT kernels): the refracted child of this hit is traced by another lane of the pixel's group
    FR_REMOTE = 16,      // ... whose lane index is in bits 8..13; its colour arrives in that lane's mailbox
    FR_HELPER_SHIFT = 8
};

// One op head = 8 dwords fetched with a single scalar s_load_dwordx8 (constant address space) and
// kept as a uniform vector: [0] info, [1] skip, [2] shape, [4..7] cull sphere (cx, cy, cz, r^2).
typedef uint32_t Head __attribute__((ext_vector_type(8)));
DI Head load_head(const AS4 OpHead* p) { return *(const AS4 Head*)p; }
#define H_INFO(h) ((h)[0])
#define H_SKIP(h) ((int)(h)[1])
#define H_SHAPE(h) ((int)(h)[2])
#define H_RUN(h) ((int)(h)[3])
// (bit_cast straight on a vector element reads element 0: go through a scalar temporary)
DI float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
#define H_CULL(h, i) as_f32((h)[4 + (i)])

// conservative bounding-sphere pre-test (no counterpart in the reference; it cannot change a result):
// the node lies inside the sphere (centre c, radius^2 r2) of its parent's space.  A line whose
// distance to c exceeds the radius by the margins folded into r2 and CULL_KAPPA has no intersection
// with the node at ANY t, so the reference's exact test would return an empty list as well.
// od_min: -inf = test against the whole line, 0 = against the half-line t >= 0 (see cull_hit2)
DI bool cull_miss(Head h, float fox, float foy, float foz, float fdx, float fdy, float fdz, float fdd, float od_min) {
    const float ocx = H_CULL(h, 0) - fox, ocy = H_CULL(h, 1) - foy, ocz = H_CULL(h, 2) - foz;
    const float od = __builtin_fmaxf(__builtin_fmaf(ocz, fdz, __builtin_fmaf(ocy, fdy, ocx * fdx)), od_min);
    const float oo = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
    const float lhs = __builtin_fmaf(oo, fdd, -(od * od));            // dist^2 * |d|^2
    const float rhs = __builtin_fmaf(CULL_KAPPA, oo, H_CULL(h, 3)) * fdd;
    return lhs > rhs;                                                 // false on NaN
}

// world_to_object (object_store.rs:309-320): the ancestors root first, then the shape's own transform
DI V3 to_object(const SceneDev& sc, const HitRec& hr, int hop, V3 p) {
    for (int i = 0; i < hr.path_len; i++) {
        const OpBody* xo = sc.bodies + sc.path_ops[hr.path_begin + i];
        p = xf_point(xo->inv, xo->w33, xo->div != 0, p);
    }
    const OpBody* xo = sc.bodies + hop;
    return xf_point(xo->inv, xo->w33, xo->div != 0, p);
}

// the same pre-test for a PAIR of ops in packed f32 (v_pk_*_f32: two ops per instruction); p = {cxA,cxB,
// cyA,cyB, czA,czB, r2A,r2B}.  Returns bit 0 / bit 1 = op A / op B may be hit (NaN counts as "may be hit").
typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F8 __attribute__((ext_vector_type(8)));
// `od_min` (0 or -inf, wave-uniform): with 0 the test is against the HALF-line t >= 0 instead of the whole line — where the
// centre lies behind the origin plane (oc . d < 0) the closest point of the half-line is the origin itself, so the projection
// is clamped at 0 and a child whose whole bounding sphere lies behind the ray origin is dropped: all its intersections have
// t < 0.  For a sphere or a cube (0 or 2 entries) such entries change nothing: not the closest hit, not a shadow answer, and
// a shape with BOTH entries negative is no refractive container (world.rs:258-296 keeps the shapes with exactly one).  A
// CSG's filtered list can keep an odd number of a shape's entries, so containers are only dropped this way in scenes
// without transparency (n1 / n2 are never consumed there); the caller decides.  One packed max per pair of children.
DI unsigned cull_hit2(F8 p, float fox, float foy, float foz, float fdx, float fdy, float fdz, float fdd, float od_min) {
    const F2 ocx = F2{p[0], p[1]} - F2{fox, fox}, ocy = F2{p[2], p[3]} - F2{foy, foy}, ocz = F2{p[4], p[5]} - F2{foz, foz};
    const F2 dx = {fdx, fdx}, dy = {fdy, fdy}, dz = {fdz, fdz}, dd = {fdd, fdd};
    const F2 od = __builtin_elementwise_max(__builtin_elementwise_fma(ocz, dz, __builtin_elementwise_fma(ocy, dy, ocx * dx)), F2{od_min, od_min});
    const F2 oo = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, ocx * ocx));
    const F2 lhs = __builtin_elementwise_fma(oo, dd, -(od * od));              // dist^2 * |d|^2
    const F2 rhs = __builtin_elementwise_fma(F2{CULL_KAPPA, CULL_KAPPA}, oo, F2{p[6], p[7]}) * dd;
    return (lhs[0] > rhs[0] ? 0u : 1u) | (lhs[1] > rhs[1] ? 0u : 2u);
}
typedef float F16 __attribute__((ext_vector_type(16)));
DI unsigned cull_hit4(F16 p, float fox, float foy, float foz, float fdx, float fdy, float fdz, float fdd, float behind) {
    const F8 lo = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
    const F8 hi = {p[8], p[9], p[10], p[11], p[12], p[13], p[14], p[15]};
    return cull_hit2(lo, fox, foy, foz, fdx, fdy, fdz, fdd, behind) | (cull_hit2(hi, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << 2);
}

// Tile-cost feedback, the end of a frame wave (rtx_api.hip: enqueue_render): leaves the tile's cost for the next frame's tile
// order — the traced rays of the tile's busiest pixel, or (uniform scenes: low bit of the word set) how long the wave ran, in
// 10 ns ticks; the first word of the wave's LDS kept the start time.  The array's address waits in a spare word of the counter block.  Out of
// line on purpose: inlined, the few registers it needs shift the allocation of the whole kernel (measured +1 to +2 % on the CSG
// workload although it runs once per wave).
__device__ __attribute__((noinline)) void tile_cost_tail(const unsigned long long* counters_base, int traced, unsigned t_start, int tile_index) {
    const unsigned long long word = __hip_atomic_load(&counters_base[RTX_CNT_COST_PTR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!word) return;
    int32_t* const tile_cost = (int32_t*)(word & ~1ull);
    // (the word may already name the array of the NEXT render queued on another stream of the context: never write past the
    // capacity the array itself states)
    if (tile_index >= __builtin_nontemporal_load(tile_cost - RTX_COST_HEADER)) return;
    int c;
    if (word & 1ull) {
        c = (int)((unsigned)__builtin_amdgcn_s_memrealtime() - (unsigned)__builtin_amdgcn_readfirstlane((int)t_start));
        c = c < 1 ? 1 : c;
    } else {
        c = traced;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(c, o, 64); c = w > c ? w : c; }
    }
    if (threadIdx.x == 0) tile_cost[tile_index] = c;
}

enum { MODE_DONE = 0, MODE_RADIANCE = 1, MODE_SHADOW = 2 };
const uint32_t LBIT = 0x80000000u;

}  // namespace

// occupancy target (waves per SIMD): the kernel waits on memory and on instruction issue, so resident
// waves matter, but every register past the budget becomes scratch traffic that misses L2 (the scratch
// of 4096 resident waves is far larger than the caches).  The shading state that has to survive a
// traversal is therefore kept small: see the comments at the state variables below.
#ifndef RTX_MIN_WAVES
#define RTX_MIN_WAVES 4
#endif
// scenes with CSG: the hit lists (LDS) and the op interpreter on top of the list walk need more of both
// LDS and registers per wave, so that variant is built for 3 waves per SIMD (168 VGPRs; 2, 3 and 4 were measured)
#ifndef RTX_MIN_WAVES_CSG
#define RTX_MIN_WAVES_CSG 3
#endif

// PROBE: the tile-cost probe launch (rtx_api.hip) — same code, but lanes map to tile centres, nothing is written
// to the image or the counters, and the rays needed file the tile into a cost class
// SPLIT: an expensive tile is rendered by 2^RTX_SPLIT_BITS waves; each takes 64 >> RTX_SPLIT_BITS of its pixels
// on its first lanes, the other lanes start idle.  Whenever a hit has two children (reflection AND refraction)
// and some lane of the wave is idle — a spare lane, or one whose pixel is finished — the refracted child is
// handed to it: its ray goes straight into the helper's LDS ray slot, the helper traces the whole subtree (handing
// out work itself) and posts the colour color_at would have returned into its mailbox; the owner meanwhile traces
// the reflected child and, back at the frame, takes the posted colour as if its own second trace had just returned
// it.  Every ray is traced exactly once and every sum is evaluated by the frame's owner in the reference's order,
// so image and counts are those of the unsplit kernel; only the wave's critical path changes: a glass pixel at
// depth 5 is a spine of two-children hits with short side chains (up to ~110 rays one after the other), and the
// side chains now run beside the spine.  Donors and idle lanes are paired wave-wide with two ballots per
// iteration (k-th donor <-> k-th idle lane), no atomics.  Used for short dispatches (rtx_api.hip), where the
// longest wave, not the throughput, bounds the frame: many GPUs per frame, small frames.
#ifndef RTX_SPLIT_BITS
#define RTX_SPLIT_BITS 2
#endif
// HCTX: the hit context (point, eye vector, normal of the hit being shaded: 9 doubles per lane) waits in LDS instead of registers
// while the hit's shadow rays are traced.  In registers it is what the compiler spills (the CSG kernels keep all of it in
// scratch, the others part of it: written per hit, re-read per light, 4.6 GB of the 4K glass frame's fabric writes); in LDS it
// costs 4.6 KB per wave, which scenes with several ray-stack slots cannot afford (S3 +22 %, S5 +21 % from the lost occupancy).
// The launcher therefore picks it per launch: only where the wave's LDS still fits the kernel's occupancy target with it.
// CSG: 0 = the scene has no CSG; 1 = CSG roots within the bounded structures (per-lane hit lists and the root's program in LDS, at
// most RTX_MAX_CSG_ENTRIES entries and RTX_MAX_CSG_NEST nested containers per root: the fast kernels); 2 = GENERAL: any number of
// entries and any nesting under a root — the lists and the per-level marks live in a slot of a global arena that the wave claims
// for its lifetime (rtx_device.h: arena), the program is read from memory.  Same merge / filter / binary-search insertion
// (csg.rs:73-136, intersections.rs:41-59); the scene picks the kernel (rtx_api.hip).
template <int CSG, bool COUNT, bool PROBE = false, bool SPLIT = false, bool HCTX = false>
__global__ __launch_bounds__(64, CSG ? RTX_MIN_WAVES_CSG : RTX_MIN_WAVES) void trace_shade_kernel(SceneDev sc, CameraDev cam, RenderParams rp,
                                                         uint8_t* __restrict__ rgb,
                                                         int32_t* __restrict__ hit_out,
                                                         unsigned long long* __restrict__ counters_base) {
    unsigned long long* counters = counters_base;
#ifdef RTX_WAVE_TRACE
    // diagnostic build (tools/wave_trace.py): start / end of every wave on the 100 MHz clock go to the hit-id
    // plane (which then holds no hit ids): [3b] start, [3b+1] end, [3b+2] rays of the slowest lane
    const unsigned wt_start = (unsigned)__builtin_amdgcn_s_memrealtime();
    int wt_class = 0;
#endif
    // LDS ray stack: origin slots [n_org][3][64] then direction slots [n_dir][3][64].  Origin / direction slot 0 = the ray being
    // traced; a container's children get a NEW origin slot unless its transform is the identity, and a NEW direction slot only
    // if the transform has a linear part (rtx_device.h: XF_*): a pure translation leaves the direction as it is and the identity
    // leaves the whole ray, so those levels share their parent's slots (nested translated groups: 1.5 KB per level, not 3 KB).
    // slotmap: lane L holds (origin slot | direction slot << 8) of list level L (a VGPR used as a wave-uniform array)
    // (the first 8 bytes of the wave's LDS keep its start time for the tile-cost feedback: a fixed address, so neither the store
    // here nor the load at the end keeps a register live through the kernel)
    extern __shared__ double lds_all[];
    double* const lds_rays = lds_all + 1;
    double* const org_base = lds_rays + threadIdx.x;
    double* const dir_base = lds_rays + sc.n_org_slots * LDS_HALF_DOUBLES + threadIdx.x;
    double* const lds_after = lds_rays + (sc.n_org_slots + sc.n_dir_slots) * LDS_HALF_DOUBLES;
    int slotmap = 0;
    const int lane = threadIdx.x;
    if (!PROBE && !SPLIT && lane == 0) *(volatile unsigned*)lds_all = (unsigned)__builtin_amdgcn_s_memrealtime();
    int tile = (int)(((unsigned long long)blockIdx.x * (unsigned)rp.tile_mul) % (unsigned)(rp.tiles_x * rp.n_bands_local));
#ifdef RTX_NO_LOCKSTEP
    constexpr bool LOCKSTEP = false;                // (A/B builds: tools/lights_probe.py)
#else
    constexpr bool LOCKSTEP = !SPLIT && !PROBE;     // one kind of ray per iteration (see the top of the main loop)
#endif
    constexpr int S = SPLIT ? RTX_SPLIT_BITS : 0;   // log2 waves per tile
    int sub = 0;                          // which part of the tile this wave renders
    if (!PROBE && rp.use_lists) {
        // block b -> b-th tile of the class lists, most expensive class first (the probe launch filled them)
        typedef int32_t I8 __attribute__((ext_vector_type(RTX_TILE_CLASSES)));
        const I8 cnt = *(const AS4 I8*)(uintptr_t)rp.tile_counts;
        int rest = (int)blockIdx.x, k = RTX_TILE_CLASSES - 1;
        if (rp.split_cap > 0) {
            // the first n_split tiles of the lists (classes >= split_min_class) belong to the SPLIT launch, which runs
            // beside this one on a second stream; both launches are sized for split_cap and drop their surplus blocks
            int n_heavy = 0;
#pragma unroll
            for (int q = RTX_TILE_CLASSES - 1; q > 0; q--) n_heavy += q >= rp.split_min_class ? cnt[q] : 0;
            const int n_split = n_heavy < rp.split_cap ? n_heavy : rp.split_cap;
            if (SPLIT) {
                sub = rest & ((1 << RTX_SPLIT_BITS) - 1);
                rest >>= RTX_SPLIT_BITS;
                if (rest >= n_split) return;
            } else {
                rest += n_split;
                if (rest >= rp.tiles_x * rp.n_bands_local) return;
            }
        }
#pragma unroll
        for (int q = RTX_TILE_CLASSES - 1; q > 0; q--)
            if (k == q && rest >= cnt[q]) { rest -= cnt[q]; k = q - 1; }
        tile = ((const AS4 int32_t*)(uintptr_t)rp.tile_lists)[(size_t)k * (rp.tiles_x * rp.n_bands_local) + rest];
#ifdef RTX_WAVE_TRACE
        wt_class = k;
#endif
    }
    // pixel of the tile this lane works for; in a split wave only the first 64 >> S lanes have one
    const int pl = SPLIT ? (((sub << (6 - S)) | lane) & 63) : lane;
    const bool spare = SPLIT && lane >= (64 >> S);
    bool waiting = false;                 // the lane is back at a frame whose refracted child a helper traces, and waits for the colour
    bool task_root = false;               // the lane's stack root is a handed-over child: post the colour instead of writing a pixel
    int wait_h = 0;                       // ... from this helper lane
    unsigned pend = 0;                    // bit i: frames[i] has a refracted child that is neither started nor handed over yet
    const int tx = tile % rp.tiles_x, bl = tile / rp.tiles_x;
    const int band = rp.band_first + bl * rp.band_stride;

    const AS4 OpBody* bodies = (const AS4 OpBody*)(uintptr_t)sc.bodies;
    const int n_lights = sc.n_lights;

    // ---- per-lane counters
    unsigned n_radiance = 0, n_shadow = 0, overflow = 0, n_skipped = 0;
    unsigned ev[RTX_EV_USED];             // reference-defined events + executed-work and occupancy diagnostics
    if (COUNT) {
#pragma unroll
        for (int i = 0; i < RTX_EV_USED; i++) ev[i] = 0;
    }
#ifdef RTX_DIAG_PHASE
#define EV(e, n) do { if (COUNT && (e) != RTX_EV_POW && (e) != RTX_EV_SCHLICK && (e) != RTX_EV_REFR && (e) != RTX_EV_BACKFACE) ev[e] += (n); } while (0)
#elif defined(RTX_DIAG_DIV)
// diagnostic build (tools/divergence.py): the event slots carry lane-occupancy figures of the divergent parts of an iteration
// instead of the reference-defined events.  DV(i, n): this lane adds n to slot i; DVW(i, pred): 64 x (wave has a lane with pred)
#define EV(e, n) do { } while (0)
#define DV(i, n) do { if (COUNT) ev[i] += (n); } while (0)
#define DVW(i, pred) do { if (COUNT && __any(pred)) { if (lane == (int)__builtin_ctzll(__ballot(1))) ev[i] += 64; } } while (0)
#else
#define EV(e, n) do { if (COUNT) ev[e] += (n); } while (0)
#endif
#ifndef DV
#define DV(i, n) do { } while (0)
#define DVW(i, pred) do { } while (0)
#endif
#ifdef RTX_TIMING
    unsigned long long tm_ffwd = 0, tm_body = 0, tm_shade = 0, tm_other = 0, tm_prep = 0, tm_light = 0, tm_mark = __builtin_amdgcn_s_memtime();
#define TM(acc) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tm_mark; tm_mark = now_; } while (0)
#else
#define TM(acc) do { } while (0)
#endif
    // RTX_TIMING == 2: the traversal is one bucket ("other") and the cull/exact slots time the shadow-ray
    // set-up and the children decision instead; "shade" is then the unwind + loop turn-around
// RTX_TIMING == 3: the CSG interpreter in detail — cull = its shape ops, exact = its BEGIN ops, shade = its MID / END list work,
// prep = the program copy into LDS, other = the rest of the traversal, light = all shading
#if defined(RTX_TIMING) && RTX_TIMING == 3
#define TMI(acc) TM(acc)
#else
#define TMI(acc) do { } while (0)
#endif
#if defined(RTX_TIMING) && RTX_TIMING == 3
#define TMT(acc) do { } while (0)
#define TMS(acc) TM(tm_light)
#define TML(acc) TM(tm_light)
#elif defined(RTX_TIMING) && RTX_TIMING == 2
#define TMT(acc) TM(tm_other)
#define TMS(acc) TM(acc)
#else
#define TMT(acc) TM(acc)
#define TMS(acc) do { } while (0)
#endif
#ifndef TML
#define TML(acc) TM(acc)
#endif

    int mode;
#ifdef RTX_WAVE_TRACE
    unsigned wt_iters = 0;
#endif
    // ---- Camera::ray_for_pixel (camera.rs:56-69); the ray goes to LDS slot 0
    {
        int px = tx * RTX_TILE + (pl & 7);
        int py = band * RTX_TILE + (pl >> 3);
        mode = (px < cam.hsize && py < cam.vsize) ? MODE_RADIANCE : MODE_DONE;
        if (spare) mode = MODE_DONE;                              // the spare lanes of a split wave start idle
        if (PROBE) {
            // probe launch: one wave per 8x8 block of TILES (so its 64 rays stay coherent), lane -> centre pixel of a tile
            const int sx_n = (rp.tiles_x + 7) >> 3;
            const int sb = (int)(((unsigned long long)blockIdx.x * (unsigned)rp.super_mul) % (unsigned)(sx_n * ((rp.n_bands_local + 7) >> 3)));
            const int ptx = (sb % sx_n) * 8 + (lane & 7), pbl = (sb / sx_n) * 8 + (lane >> 3);
            px = ptx * RTX_TILE + 4;
            py = (rp.band_first + pbl * rp.band_stride) * RTX_TILE + 4;
            mode = (ptx < rp.tiles_x && pbl < rp.n_bands_local) ? MODE_RADIANCE : MODE_DONE;
            px = px < cam.hsize ? px : cam.hsize - 1;
            py = py < cam.vsize ? py : cam.vsize - 1;
        }
        double xoffset = ((double)px + 0.5) * cam.pixel_size;
        double yoffset = ((double)py + 0.5) * cam.pixel_size;
        double world_x = cam.half_width - xoffset;
        double world_y = cam.half_height - yoffset;
        V3 pixel = xf_point(cam.inv, cam.w33, cam.div != 0, V3{world_x, world_y, -1.0});
        V3 ro = V3{cam.origin[0], cam.origin[1], cam.origin[2]};
        V3 rd = normalize3(pixel - ro);
        org_base[0] = ro.x; org_base[64] = ro.y; org_base[128] = ro.z;
        dir_base[0] = rd.x; dir_base[64] = rd.y; dir_base[128] = rd.z;
    }
    // Canvas::write_pixel + Color::to_rgb (canvas.rs:31-33, color.rs:40-48), the moment a lane's colour is final
    auto write_pixel = [&](double r, double g, double b) {
        const int px = tx * RTX_TILE + (pl & 7);
        const int row = rp.compact ? (bl * RTX_TILE + (pl >> 3)) : (band * RTX_TILE + (pl >> 3));
        const size_t idx = (size_t)row * cam.hsize + px;
        rgb[idx * 3 + 0] = quantise(r);
        rgb[idx * 3 + 1] = quantise(g);
        rgb[idx * 3 + 2] = quantise(b);
    };

    // ---- state that survives a traversal (everything else is rebuilt or re-read):
    int remaining = rp.max_depth;         // budget of the color_at call being traced
    int sp = 0;                           // frames below sp wait for children
    PendRay prays[RTX_MAX_DEPTH + 1];
    Frame frames[RTX_MAX_DEPTH + 1];      // + 1: the hit at budget 0 (sp == max_depth) still parks its partial light sum in frames[sp]
    bool primary = !spare;                // (the spare lanes of a split wave never trace a primary ray)
    // the hit being shaded while its shadow rays are traced (PreparedComputations, world.rs:225-256)
    V3 hp_r = {0, 0, 0}, eyev_r = {0, 0, 0}, nv_r = {0, 0, 0};   // (HCTX: unused — the context waits in LDS)
    int hop = 0;                          // op index of the hit shape
    int hmat = 0;                         // its material
    int n1op = -1, n2op = -1;             // ops whose refractive indices are n1 / n2 (-1: vacuum, 1.0)
    int light_k = 0;
    double shadow_dist = 0.0;

    // CSG hit list (per lane)
    // (in LDS, after the ray slots: entry e of this lane at [e * 64 + lane]; sized by the scene's largest CSG
    // root.  In scratch, every step of the merge loops was a dependent round trip to memory.)
    double* lt_ = lds_after + lane;
    uint32_t* ls_ = (uint32_t*)(lds_after + sc.csg_lds_entries * 64) + lane;
    uint32_t* marks_ = nullptr;           // CSG == 2: list length at BEGIN / MID per container level: [level][2][64]
    int arena_slot = -1;
    if constexpr (CSG == 2) {
        // ---- claim a slot of the arena (one bit per slot; the arena has more slots than waves can be resident, so a free one
        // exists; the search is bounded all the same: a wave that finds none reports an overflow and ends, it never spins for ever)
        if (lane == 0) {
            unsigned* bm = sc.arena_bitmap;
            const unsigned W = (unsigned)sc.arena_words;
            unsigned w = (blockIdx.x * 2654435761u) % W;
            for (unsigned tries = 0; tries < 64u * W && arena_slot < 0; tries++) {
                unsigned cur = __hip_atomic_load(&bm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (~cur != 0u) {
                    const unsigned bit = (unsigned)__builtin_ctz(~cur);
                    const unsigned old = atomicOr(&bm[w], 1u << bit);
                    if (!(old & (1u << bit))) { arena_slot = (int)(w * 32u + bit); break; }
                    cur = old | (1u << bit);
                }
                if (arena_slot < 0) { w = w + 1u == W ? 0u : w + 1u; if ((tries & 63u) == 63u) __builtin_amdgcn_s_sleep(32); }
            }
        }
        arena_slot = __builtin_amdgcn_readfirstlane(arena_slot);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (arena_slot < 0) {
            if (lane == 0 && !PROBE) {
                atomicAdd(&counters_base[(blockIdx.x % RTX_CNT_SLOTS) * RTX_CNT_STRIDE + RTX_CNT_OVERFLOW], 64ull);
                if (rp.sticky) __hip_atomic_store(rp.sticky, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        unsigned char* slot_base = sc.arena + (size_t)arena_slot * (size_t)sc.arena_slot_bytes;
        lt_ = (double*)slot_base + lane;
        ls_ = (uint32_t*)(slot_base + (size_t)sc.csg_entries * 64 * 8) + lane;
        marks_ = (uint32_t*)(slot_base + (size_t)sc.csg_entries * 64 * 12) + lane;
    }
    auto arena_release = [&]() {
        if constexpr (CSG == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) atomicAnd(&sc.arena_bitmap[arena_slot >> 5], ~(1u << (arena_slot & 31)));
        }
    };
#define lt(i) lt_[(i) * 64]
#define ls(i) ls_[(i) * 64]
    // split mailboxes (after the CSG lists): colour [3][64] indexed by the SENDER's lane + a posted flag per lane.
    // A lane posts at most once (then it retires), so one box per lane is enough for all levels.
    // mb_state: 0 free, 1 a task is running on this lane, 2 its colour is posted and not yet taken
    // the program of the CSG root being walked (after the lists; 16-byte aligned), then the split mailboxes
    uint32_t* const prog_lds = (uint32_t*)(lds_after + ((sc.csg_lds_entries * 64 * 12 + 15) / 16) * 2);
    volatile double* const mb_ = (volatile double*)(prog_lds + sc.csg_prog_lds);
    volatile int* const mb_state = (volatile int*)(mb_ + 3 * 64);
    volatile int* const mb_rem = mb_state + 64;
    if (SPLIT) mb_state[lane] = 0;
    // HCTX: the hit context's LDS area, after everything else: [9][64] doubles
    // (its offset is computed from the LDS base with integers, so the pointer provably stays an LDS pointer: ds_* instead of flat_*)
    const unsigned hctx_off = ((unsigned)(sc.n_org_slots + sc.n_dir_slots) * LDS_HALF_DOUBLES * 8u + (((unsigned)sc.csg_lds_entries * 64u * 12u + 15u) & ~15u) +
                               (unsigned)sc.csg_prog_lds * 4u + (SPLIT ? 64u * 32u : 0u) + 7u) & ~7u;
    double* const hctx = (double*)((char*)lds_rays + hctx_off) + lane;

    bool phaseR = false;                  // the kind of ray the last iteration traced (see the top of the loop)
    bool guard_hit = false;
    int guard = 0;                        // SPLIT: a wave whose lanes wait for each other must still end (it cannot happen by
                                          // construction — helpers never wait for their owners — but a hung wave hangs the GPU)
    while (__any(mode != MODE_DONE || (SPLIT && waiting))) {
        if (SPLIT && ++guard > rp.split_guard) { guard_hit = true; break; }
        // ---- one KIND of ray per iteration.  A lane alternates radiance and shadow traces, and the two kinds end in different
        // shading blocks (PreparedComputations / lighting + children), ~450 vector instructions each.  Lanes fall out of step
        // whenever a radiance ray misses or a shadow ray is skipped, and a wave whose lanes are at both kinds runs BOTH blocks
        // in one iteration (27 % of S4's iterations did).  So when both kinds wait, only the SHADOW rays are traced and the
        // radiance lanes sit this iteration out (as MODE_DONE: they keep their state and their ray in slot 0), which puts them
        // back in step with the rest: S4 4.80 -> 4.55 ms.  Shadow rays first, because a hit is followed by one shadow ray
        // per light: plain alternation gave the shadow lanes every other turn and lost 7 % on a three-light variant of S4
        // (tools/lights_probe.py: 3.39 ms without any of this, 3.67 alternating, 3.30 shadow-first; 1.60 / 1.56 / 1.53 with
        // one light).  Letting the LARGER group go first was measured too: no gain — the lanes that sit out repeatedly
        // become the tail.  Not in SPLIT waves: their idle lanes are the helpers.
        unsigned long long hold_mask = 0;     // (a wave mask in scalar registers, not a per-lane flag: nothing more to keep live per lane)
        if (LOCKSTEP) {
            phaseR = !__any(mode == MODE_SHADOW);
            const bool hold = mode == MODE_RADIANCE && !phaseR;
            hold_mask = __ballot(hold);
            if (hold) mode = MODE_DONE;
        }
        TML(tm_shade);
        // ================================================================= TRACE
        // World::intersect (world.rs:111-118) for every lane's current ray.
#ifdef RTX_WAVE_TRACE
        wt_iters += (mode != MODE_DONE) ? 1 : 0;
#endif
        if (SPLIT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // a donor lane may have written this lane's ray slot
        if (mode == MODE_RADIANCE) n_radiance++;
        else if (mode == MODE_SHADOW) n_shadow++;
        EV(RTX_EV_LANE_SLOTS, 1);         // every lane, busy or not: 64 per wave trace
        DV(0, 1); DV(1, mode != MODE_DONE ? 1 : 0);                       // slot 0: lane-slots of all iterations, 1: lanes tracing a ray

        // running state instead of a sorted hit list; shapes are identified by op index
        double best_t = 0.0; int best_op = -1; bool best_inside = false;
        double c1t = -INF, c2t = -INF; int c1s = -1, c2s = -1;     // refractive containers
        int resume_at = 0x7fffffff;       // CSG interpreter: first program offset the lane takes part in again
        int alive = (mode != MODE_DONE) ? 0 : -1;   // deepest list level the lane takes part in
        bool decided = false;             // shadow ray already answered (world.rs:203-206)
        int cnt = 0;                      // CSG list length
        uint64_t starts = 0, mids = 0;    // 8-bit fields per clevel (CSG == 2: in the arena instead, any number of levels)
        auto set_start = [&](int cl, int v) { if constexpr (CSG == 2) marks_[cl * 128] = (uint32_t)v; else starts = (starts & ~(0xffull << (cl * 8))) | ((uint64_t)v << (cl * 8)); };
        auto set_mid = [&](int cl, int v) { if constexpr (CSG == 2) marks_[cl * 128 + 64] = (uint32_t)v; else mids = (mids & ~(0xffull << (cl * 8))) | ((uint64_t)v << (cl * 8)); };
        auto get_start = [&](int cl) -> int { if constexpr (CSG == 2) return (int)marks_[cl * 128]; else return (int)((starts >> (cl * 8)) & 0xff); };
        auto get_mid = [&](int cl) -> int { if constexpr (CSG == 2) return (int)marks_[cl * 128 + 64]; else return (int)((mids >> (cl * 8)) & 0xff); };
        const int list_cap = CSG == 2 ? sc.csg_entries : RTX_MAX_CSG_ENTRIES;
        // (wave-uniform where the kinds take turns: the early-out tests and the container bookkeeping below become scalar branches)
        const bool is_shadow = LOCKSTEP ? !phaseR : mode == MODE_SHADOW;

        // The running state is ORDER-INDEPENDENT: ties in t resolve by op index (= the reference's visiting
        // order, which its stable sorts preserve), so shapes of one run may be fed in any order.
        // closest hit: smallest (t, op); innermost open containers: largest (t, op)
        auto closer = [&](double t, int op) { return best_op < 0 || t < best_t || (t == best_t && op < best_op); };
        auto later = [&](double t, int op, double ct, int cs) { return t > ct || (t == ct && op > cs); };
        // (written with selects on the VALUES: as two branches storing into c1 / c2 the compiler turned the four variables
        // into an array in scratch addressed through a selected pointer — a memory round trip per negative-t entry)
        auto enter = [&](double t, int op) {
            const bool a = later(t, op, c1t, c1s);
            const bool b = !a && later(t, op, c2t, c2s);
            c2t = a ? c1t : (b ? t : c2t); c2s = a ? c1s : (b ? op : c2s);
            c1t = a ? t : c1t; c1s = a ? op : c1s;
        };
        // feed the hits of one shape (sorted, k <= 2) into the running state
        auto feed = [&](int k, double t0, double t1, int shape) {
            if (k == 0) return;
            const bool nn0 = t0 >= 0.0;
            if (nn0) {
                if (closer(t0, shape)) { best_t = t0; best_op = shape; best_inside = false; }
            } else {
                if (k == 2 && t1 >= 0.0) {
                    if (closer(t1, shape)) { best_t = t1; best_op = shape; best_inside = true; }
                }
                // exactly one negative entry -> the shape is an open container behind the origin (only a radiance ray's
                // hit asks for its containers: world.rs:258-296 runs inside PreparedComputations)
                if (!is_shadow && (k == 1 || t1 >= 0.0)) enter(t0, shape);
            }
        };

        // parent-space ray of the level being walked: f64 for the exact tests, an f32 copy for the
        // conservative bounding-sphere pre-test; (re)read from the LDS stack when the walk changes level
        V3 po, pd;
        float fox, foy, foz, fdx, fdy, fdz, fdd;
        int cached_level = -1;
        bool reg_ray = false;             // inside a CSG root whose local ray lives in registers only (OPH_REGRAY): po / pd are it, at every level
        auto sync_level = [&](int level) {
            if (level == cached_level) return;
            if (reg_ray) { cached_level = level; return; }
            const int sm = __builtin_amdgcn_readlane(slotmap, level);
            const double* so = org_base + (sm & 0xff) * LDS_HALF_DOUBLES;
            const double* sd = dir_base + (sm >> 8) * LDS_HALF_DOUBLES;
            po = V3{so[0], so[64], so[128]}; pd = V3{sd[0], sd[64], sd[128]};
            fox = (float)po.x; foy = (float)po.y; foz = (float)po.z;
            fdx = (float)pd.x; fdy = (float)pd.y; fdz = (float)pd.z;
            fdd = __builtin_fmaf(fdz, fdz, __builtin_fmaf(fdy, fdy, fdx * fdx));
            cached_level = level;
        };
        sync_level(0);
        // the lanes that pass a container's bounds test enter it: the children (level + 1) walk in the container's space.
        // xf: XF_GENERAL / XF_TRANSLATE / XF_IDENTITY (host-classified from the inverse); the caller has put the local
        // ray into po / pd (only the parts that change: pd is untouched unless XF_GENERAL, po unless not XF_IDENTITY)
        auto enter_level = [&](int level, int xf, bool pass) {
            const int sm = __builtin_amdgcn_readlane(slotmap, level);
            const int os = (sm & 0xff) + (xf != XF_IDENTITY ? 1 : 0), ds = (sm >> 8) + (xf == XF_GENERAL ? 1 : 0);
            if (lane == level + 1) slotmap = os | (ds << 8);
            if (xf != XF_IDENTITY) {
                fox = (float)po.x; foy = (float)po.y; foz = (float)po.z;
                if (pass) { double* s = org_base + os * LDS_HALF_DOUBLES; s[0] = po.x; s[64] = po.y; s[128] = po.z; }
            }
            if (xf == XF_GENERAL) {
                fdx = (float)pd.x; fdy = (float)pd.y; fdz = (float)pd.z;
                fdd = __builtin_fmaf(fdz, fdz, __builtin_fmaf(fdy, fdy, fdx * fdx));
                if (pass) { double* s = dir_base + ds * LDS_HALF_DOUBLES; s[0] = pd.x; s[64] = pd.y; s[128] = pd.z; }
            }
            cached_level = level + 1;
        };
        // Ray into a container's space (object_store.rs:244-246, rays.rs:21-23), by transform class.  A translation's inverse has
        // the identity as its linear part: the direction is unchanged and the origin is p + t (the products by 1 and the sums
        // with products by 0 of the general formula change nothing but the sign of a zero, which no later operation can
        // observe: zeros of ray components are only ever added, multiplied and compared, never divided by); the identity
        // changes nothing.

        // ---- the CSG interpreter: walks the PROGRAM of ONE CSG root in program order (hit lists need the reference's order;
        // everything outside CSG goes through the child lists below).  The program (rtx_device.h: CSG program records —
        // heads, container transforms and bounds, the shapes' inverses) is copied into LDS by the whole wave with one
        // vector load per lane, and every record is then read from LDS: walking a root costs ONE trip to memory instead
        // of two or three dependent scalar round trips per op.  `cur` / resume_at are dword offsets into the program.
        auto interpret = [&](const int poff, const int plen) {
        TMI(tm_other);
        const uint32_t* const prog = CSG == 2 ? sc.csg_prog + poff : prog_lds;
        if constexpr (CSG != 2) {
            typedef uint32_t U4 __attribute__((ext_vector_type(4)));
            const U4* src = (const U4*)(sc.csg_prog + poff);
            U4* dst = (U4*)prog_lds;
            for (int i = lane; i < (plen >> 2); i += 64) dst[i] = src[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        TMI(tm_prep);
        int cur = 0;
        while (cur < plen) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            cached_level = __builtin_amdgcn_readfirstlane(cached_level);
            const bool active = cur >= resume_at;
            if (!__any(active)) {
                cur = __builtin_amdgcn_readfirstlane(wave_min(resume_at));   // keep the offset scalar
                continue;
            }
            const uint32_t* rec = prog + cur;
            const uint32_t info = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[0]);
            const int kind = OPH_KIND(info);
            const int level = OPH_LEVEL(info);
            const int flags = OPH_FLAGS(info);
            const int aux = __builtin_amdgcn_readfirstlane((int)rec[2]);
            int next = cur + __builtin_amdgcn_readfirstlane((int)rec[3]);
            const float cu0 = as_f32(rec[4]), cu1 = as_f32(rec[5]), cu2 = as_f32(rec[6]), cu3 = as_f32(rec[7]);
            // conservative bounding-sphere pre-test (see cull_miss)
            auto sphere_miss = [&]() {
                const float ocx = cu0 - fox, ocy = cu1 - foy, ocz = cu2 - foz;
                const float od = __builtin_fmaf(ocz, fdz, __builtin_fmaf(ocy, fdy, ocx * fdx));
                const float oo = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
                const float lhs = __builtin_fmaf(oo, fdd, -(od * od));
                const float rhs = __builtin_fmaf(CULL_KAPPA, oo, cu3) * fdd;
                return lhs > rhs;                                             // false on NaN
            };

            if (kind <= OPK_CUBE) {
                // ================= one shape: ObjectStore::intersect (object_store.rs:243-260) for every lane still in
                TM(tm_other);
                sync_level(level);
                const int idx = aux;                                          // the shape's op index (its identity)
                if (active) {
                    EV(RTX_EV_XFORM, 1); EV(RTX_EV_CULL, 1);
                    if (kind == OPK_SPHERE) EV(RTX_EV_SPHERE, 1); else if (kind == OPK_PLANE) EV(RTX_EV_PLANE, 1); else EV(RTX_EV_CUBE, 1);
                }
                EV(RTX_EV_OP_SLOTS, 1);
                const bool cand = active && !sphere_miss();
                TMT(tm_ffwd);
                if (__any(cand)) {
                    const double* inv = (const double*)(rec + 8);
                    V3 lo = xf_point(inv, inv[12], (flags & OPF_DIV) != 0, po);
                    V3 ld = xf_vector(inv, pd);
                    if (cand) {
                        EV(RTX_EV_X_XFORM, 1);
                        int k = 0; double t0 = 0.0, t1 = 0.0;
                        if (kind == OPK_SPHERE) {                 // sphere.rs:16-32
                            EV(RTX_EV_X_SPHERE, 1);
                            double a = dot3(ld, ld);
                            double b = 2.0 * dot3(ld, lo);
                            double c = dot3(lo, lo) - 1.0;
                            double disc = b * b - 4.0 * a * c;
                            if (!(disc < 0.0)) {
                                double sq = __builtin_sqrt(disc);
                                t0 = (-b - sq) / (2.0 * a);
                                t1 = (-b + sq) / (2.0 * a);
                                k = 2;
                            }
                        } else if (kind == OPK_PLANE) {           // plane.rs:16-23
                            if (!(__builtin_fabs(ld.y) < 1e-12)) { t0 = -lo.y / ld.y; k = 1; }
                        } else {                                  // cube.rs:18-31
                            EV(RTX_EV_X_BOX, 1);
                            if (slabs(lo, ld, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, t0, t1)) k = 2;
                        }
                        // (every op under a CSG root goes to the per-lane list, in program order)
                        if (cnt + k > list_cap) { overflow = 1; k = 0; }
                        if (k >= 1) { lt(cnt) = t0; ls(cnt) = (uint32_t)idx; cnt++; }
                        if (k == 2) { lt(cnt) = t1; ls(cnt) = (uint32_t)idx; cnt++; }
                    }
                }
                TMT(tm_body);
                TMI(tm_ffwd);
            } else if (kind == OPK_GROUP_BEGIN || kind == OPK_CSG_BEGIN) {
                sync_level(level);
                const int skip = __builtin_amdgcn_readfirstlane((int)rec[1]);
                if (active && cur != 0) { EV(RTX_EV_XFORM, 1); EV(RTX_EV_BOUNDS, 1); }   // (the root was counted with its list)
                EV(RTX_EV_OP_SLOTS, 1);
                bool test = active;
                if (cu3 < __builtin_inff()) {
                    const bool miss = sphere_miss();
                    if (active) EV(RTX_EV_CULL, 1);
                    test = active && !miss;
                    if (active && miss) resume_at = skip;
                }
                if (!__any(test)) {
                    next = skip;
                } else {
                    // ---- Groups::local_intersect (group.rs:64-75) / CSGs::local_intersect (csg.rs:73-78)
                    const double* bnd = (const double*)(rec + 8);             // local bounds, then the transform data
                    const double* xd = bnd + 6;
                    const int xf = OPH_XF(info);
                    V3 lo, ld;
                    if (xf == XF_GENERAL) { lo = xf_point(xd, xd[12], (flags & OPF_DIV) != 0, po); ld = xf_vector(xd, pd); }
                    else if (xf == XF_TRANSLATE) { lo = V3{po.x + xd[0], po.y + xd[1], po.z + xd[2]}; ld = pd; }
                    else { lo = po; ld = pd; }
                    if (test) EV(RTX_EV_X_XFORM, 1);
                    // decision by the interval filter; the exact slab test (six divisions) only where a lane is undecided
                    const int fdec = OPH_NOFILTER(info) ? -1 : slab_filter(lo, ld, bnd[0], bnd[1], bnd[2], bnd[3], bnd[4], bnd[5]);
                    bool pass = test && fdec == 1;
                    const bool undecided = test && fdec < 0;
                    if (__any(undecided)) {
                        if (undecided) EV(RTX_EV_X_BOX, 1);
                        double tmin, tmax;
                        const bool ex = slabs(lo, ld, bnd[0], bnd[1], bnd[2], bnd[3], bnd[4], bnd[5], tmin, tmax);
                        pass = pass || (undecided && ex);
                    }
                    if (test && !pass) resume_at = skip;
                    if (!__any(pass)) {
                        next = skip;
                    } else {
                        // the children (level + 1) walk in this node's space
                        po = lo; pd = ld;
                        if (OPH_REGRAY(info)) {
                            // every container below has the identity transform: this ray serves the whole program from registers
                            const int sm = __builtin_amdgcn_readlane(slotmap, level);
                            if (lane == level + 1) slotmap = sm;
                            fox = (float)po.x; foy = (float)po.y; foz = (float)po.z;
                            fdx = (float)pd.x; fdy = (float)pd.y; fdz = (float)pd.z;
                            fdd = __builtin_fmaf(fdz, fdz, __builtin_fmaf(fdy, fdy, fdx * fdx));
                            cached_level = level + 1;
                            reg_ray = true;
                        } else enter_level(level, xf, pass);
                        if (pass) set_start(OPH_CLEVEL(info), cnt);
                    }
                }
                TMI(tm_body);
            } else {
                // ================= CSG_MID / CSG_END / GROUP_END: list work of the lanes that are in
                const int cl = OPH_CLEVEL(info);
                if (kind == OPK_CSG_MID) {
                    if (active) set_mid(cl, cnt);
                } else {
                    // The merge / filter of csg.rs:83-121 on the lanes' LDS lists is a chain of ~60 dependent LDS round trips
                    // per END op when done in place.  FAST PATH: segments of at most four entries (two children with two hits
                    // each: the common case) are read into registers at once, marked, sorted, filtered and handed on there.
                    // It is taken when EVERY participating lane's segment is that short and no lane has two equal t after
                    // the sort (the reference re-inserts survivors by binary search, intersections.rs:41-47, whose result
                    // differs from appending only among equal keys); otherwise all lanes take the in-place code below.
                    int s0w;
                    if constexpr (CSG == 2) s0w = active ? get_start(cl) : 0; else s0w = get_start(cl);
                    const int lenw = cnt - s0w;
                    bool fast = !__any(active && lenw > 4);
                    double ft0 = INF, ft1 = INF, ft2 = INF, ft3 = INF; uint32_t fi0 = 0, fi1 = 0, fi2 = 0, fi3 = 0;
                    if (fast) {
                        if (active) {
                            if (lenw > 0) { ft0 = lt(s0w); fi0 = ls(s0w); }
                            if (lenw > 1) { ft1 = lt(s0w + 1); fi1 = ls(s0w + 1); }
                            if (lenw > 2) { ft2 = lt(s0w + 2); fi2 = ls(s0w + 2); }
                            if (lenw > 3) { ft3 = lt(s0w + 3); fi3 = ls(s0w + 3); }
                            if (kind == OPK_CSG_END) {
                                // csg.rs:83-90: left_shapes membership == "came from the left child"
                                const int ml = get_mid(cl) - s0w;      // entries of the left child
                                fi0 = 0 < ml ? (fi0 | LBIT) : (fi0 & ~LBIT); fi1 = 1 < ml ? (fi1 | LBIT) : (fi1 & ~LBIT);
                                fi2 = 2 < ml ? (fi2 | LBIT) : (fi2 & ~LBIT); fi3 = 3 < ml ? (fi3 | LBIT) : (fi3 & ~LBIT);
                            }
                            // stable sort (intersections.rs:56-59): insertion sort as adjacent exchanges on strict >, the
                            // comparisons the in-place loop makes (absent entries hold +inf and stay behind)
#define FSWAP(a, b) do { const bool sw_ = ft##a > ft##b; const double tt_ = sw_ ? ft##b : ft##a; ft##b = sw_ ? ft##a : ft##b; ft##a = tt_; \
                         const uint32_t ii_ = sw_ ? fi##b : fi##a; fi##b = sw_ ? fi##a : fi##b; fi##a = ii_; } while (0)
                            FSWAP(0, 1); FSWAP(1, 2); FSWAP(0, 1); FSWAP(2, 3); FSWAP(1, 2); FSWAP(0, 1);
#undef FSWAP
                        }
                        const bool tie = active && kind == OPK_CSG_END &&
                                         ((lenw > 1 && ft0 == ft1) || (lenw > 2 && ft1 == ft2) || (lenw > 3 && ft2 == ft3));
                        if (__any(tie)) fast = false;
                    }
                    if (fast) {
                        if (active) {
                            int n_out = lenw;
                            if (kind == OPK_CSG_END) {
                                // filter_intersections (csg.rs:94-121); keys are distinct here, so re-insertion = appending
                                const int cop = aux;
                                bool inl = false, inr = false;
                                double o0 = INF, o1 = INF, o2 = INF, o3 = INF; uint32_t j0 = 0, j1 = 0, j2 = 0, j3 = 0;
                                n_out = 0;
#define FSTEP(k) do { if (lenw > k) { EV(RTX_EV_CSG_ENTRIES, 1); const bool lhit = (fi##k & LBIT) != 0; bool allowed; \
                          if (cop == RTX_CSG_UNION) allowed = (lhit && !inr) || (!lhit && !inl); \
                          else if (cop == RTX_CSG_INTERSECTION) allowed = (lhit && inr) || (!lhit && inl); \
                          else allowed = (lhit && !inr) || (!lhit && inl); \
                          if (allowed) { \
                              o0 = n_out == 0 ? ft##k : o0; j0 = n_out == 0 ? fi##k : j0; o1 = n_out == 1 ? ft##k : o1; j1 = n_out == 1 ? fi##k : j1; \
                              o2 = n_out == 2 ? ft##k : o2; j2 = n_out == 2 ? fi##k : j2; o3 = n_out == 3 ? ft##k : o3; j3 = n_out == 3 ? fi##k : j3; \
                              n_out++; } \
                          if (lhit) inl = !inl; else inr = !inr; } } while (0)
                                FSTEP(0); FSTEP(1); FSTEP(2); FSTEP(3);
#undef FSTEP
                                ft0 = o0; ft1 = o1; ft2 = o2; ft3 = o3; fi0 = j0; fi1 = j1; fi2 = j2; fi3 = j3;
                            }
                            cnt = s0w + n_out;
                            if (kind == OPK_CSG_END && (flags & OPF_ROOT)) {
                                // hand the filtered list of this CSG root to the running state (entries ascending in t)
                                const int h0 = (int)(fi0 & ~LBIT), h1 = (int)(fi1 & ~LBIT), h2 = (int)(fi2 & ~LBIT), h3 = (int)(fi3 & ~LBIT);
                                const bool g0 = n_out > 0 && !(ft0 >= 0.0), g1 = n_out > 1 && !(ft1 >= 0.0), g2 = n_out > 2 && !(ft2 >= 0.0), g3 = n_out > 3 && !(ft3 >= 0.0);
                                bool open = true;      // no entry with t >= 0 met yet
#define FHAND(k) do { if (open && n_out > k) { const int shape = h##k; \
                          const int negs = (g0 && h0 == shape) + (g1 && h1 == shape) + (g2 && h2 == shape) + (g3 && h3 == shape); \
                          if (ft##k >= 0.0) { if (closer(ft##k, shape)) { best_t = ft##k; best_op = shape; best_inside = (negs == 1); } open = false; } \
                          else if (negs == 1) enter(ft##k, shape); } } while (0)
                                FHAND(0); FHAND(1); FHAND(2); FHAND(3);
#undef FHAND
                                cnt = 0;
                                if (is_shadow && !rp.no_early_out && best_op >= 0 && best_t < shadow_dist) { resume_at = 0x7fffffff; decided = true; }
                            } else {
                                // the merged (and filtered) segment goes back for the enclosing CSG
                                if (n_out > 0) { lt(s0w) = ft0; ls(s0w) = fi0; }
                                if (n_out > 1) { lt(s0w + 1) = ft1; ls(s0w + 1) = fi1; }
                                if (n_out > 2) { lt(s0w + 2) = ft2; ls(s0w + 2) = fi2; }
                                if (n_out > 3) { lt(s0w + 3) = ft3; ls(s0w + 3) = fi3; }
                            }
                        }
                    } else if (active) {
                    const int s0 = get_start(cl);
                    if (kind == OPK_CSG_END) {
                        // csg.rs:83-90: left_shapes membership == "came from the left child"
                        const int m0 = get_mid(cl);
                        for (int i = s0; i < cnt; i++) ls(i) = (i < m0) ? (ls(i) | LBIT) : (ls(i) & ~LBIT);
                    }
                    // stable sort of the concatenated child lists (intersections.rs:56-59)
                    for (int i = s0 + 1; i < cnt; i++) {
                        double t = lt(i); uint32_t id = ls(i);
                        int j = i - 1;
                        while (j >= s0 && lt(j) > t) { lt(j + 1) = lt(j); ls(j + 1) = ls(j); j--; }
                        lt(j + 1) = t; ls(j + 1) = id;
                    }
                    if (kind == OPK_CSG_END) {
                        // filter_intersections (csg.rs:94-121)
                        const int cop = aux;
                        bool inl = false, inr = false;
                        int out = s0;
                        for (int i = s0; i < cnt; i++) {
                            EV(RTX_EV_CSG_ENTRIES, 1);
                            const double t = lt(i); const uint32_t id = ls(i);
                            const bool lhit = (id & LBIT) != 0;
                            bool allowed;                     // csg.rs:123-136
                            if (cop == RTX_CSG_UNION) allowed = (lhit && !inr) || (!lhit && !inl);
                            else if (cop == RTX_CSG_INTERSECTION) allowed = (lhit && inr) || (!lhit && inl);
                            else allowed = (lhit && !inr) || (!lhit && inl);
                            if (allowed) {
                                // Intersections::add_intersection (intersections.rs:41-47): binary search insert
                                int size = out - s0, left = 0, right = size, pos = -1;
                                while (left < right) {
                                    int mid = left + size / 2;
                                    double tm = lt(s0 + mid);
                                    if (tm < t) left = mid + 1;
                                    else if (tm > t) right = mid;
                                    else { pos = mid; break; }
                                    size = right - left;
                                }
                                if (pos < 0) pos = left;
                                for (int j = out; j > s0 + pos; j--) { lt(j) = lt(j - 1); ls(j) = ls(j - 1); }
                                lt(s0 + pos) = t; ls(s0 + pos) = id;
                                out++;
                            }
                            if (lhit) inl = !inl; else inr = !inr;
                        }
                        cnt = out;
                        if (flags & OPF_ROOT) {
                            // hand the filtered list of this CSG root to the running state
                            for (int i = 0; i < cnt; i++) {
                                const double t = lt(i); const int shape = (int)(ls(i) & ~LBIT);
                                int negs = 0;
                                for (int j = 0; j < cnt; j++)
                                    if ((int)(ls(j) & ~LBIT) == shape && !(lt(j) >= 0.0)) negs++;
                                if (t >= 0.0) {
                                    if (closer(t, shape)) { best_t = t; best_op = shape; best_inside = (negs == 1); }
                                    break;      // later entries are larger (or tie with a later op)
                                } else if (negs == 1) {
                                    enter(t, shape);
                                }
                            }
                            cnt = 0;
                            if (is_shadow && !rp.no_early_out && best_op >= 0 && best_t < shadow_dist) { resume_at = 0x7fffffff; decided = true; }
                        }
                    }
                    }
                }
            }
            TMI(tm_shade);
            cur = next;
        }
        if (reg_ray) { reg_ray = false; cached_level = -1; }      // (the list walk re-reads its level's ray from LDS)
        };

        // ================= list-by-list traversal (everything outside CSG)
        // Wave-uniform walk over child lists (rtx_device.h: ListChunk).  Per chunk: one packed-f32 sweep over
        // the children's bounding spheres -> per-lane candidate bits; shapes are tested per lane; of the
        // containers only those that SOME lane may hit are visited (one scalar round trip each: ContRec).
        // A lane takes part in the list at `level` iff alive >= level (a lane that fails a group's test,
        // group.rs:67, keeps alive at the parent's level).  The per-level (chunk, remaining containers) pairs
        // live in two VGPRs used as lane-indexed uniform arrays.
        {
            const AS4 ListChunk* chunks = (const AS4 ListChunk*)(uintptr_t)sc.chunks;
            const AS4 ContRec* conts = (const AS4 ContRec*)(uintptr_t)sc.cont_recs;
            // (the spheres of CSG roots are the tight ones unless the reference's node-visit counts are to be reproduced: a lane that
            // misses the tight sphere gets an empty list from the CSG, but the reference would still have visited its children)
            const AS4 F16* lq = (const AS4 F16*)(uintptr_t)(rp.no_early_out ? sc.list_cull : sc.list_cull_tight);
            const AS4 int32_t* lops = (const AS4 int32_t*)(uintptr_t)sc.list_ops;
            int level = 0, chunk = 0;
            unsigned U = 0;
            bool fresh = true;
            int stk_chunk = 0, stk_U = 0, stk_cm = 0, stk_next = 0, stk_cb = 0;
            bool have_hdr = false;                                 // hdr already holds the fresh chunk's header (it came with the ContRec)
            Head hdr = load_head((const AS4 OpHead*)chunks);       // ListChunk has the size of a head: 8 dwords
            for (;;) {
                level = __builtin_amdgcn_readfirstlane(level);
                chunk = __builtin_amdgcn_readfirstlane(chunk);
                U = __builtin_amdgcn_readfirstlane(U);
                cached_level = __builtin_amdgcn_readfirstlane(cached_level);
                const bool act = alive >= level && !decided;
                if (fresh) {
                    fresh = false;
                    U = 0;
                    if (!have_hdr) hdr = load_head((const AS4 OpHead*)(chunks + chunk));
                    have_hdr = false;
                    if (__any(act)) {
                        TM(tm_other);
                        const int n = (int)(hdr[0] & 0xffu), child_base = (int)hdr[1], op0 = (int)hdr[6];
                        const unsigned shape_mask = hdr[2], plane_mask = hdr[3], cont_mask = hdr[4];
                        sync_level(level);
                        EV(RTX_EV_OP_SLOTS, n);
                        if (COUNT && act) {
                            const ChunkCounts cn = sc.chunk_counts[chunk];
                            EV(RTX_EV_XFORM, cn.spheres + cn.planes + cn.cubes + cn.containers);
                            EV(RTX_EV_SPHERE, cn.spheres); EV(RTX_EV_PLANE, cn.planes); EV(RTX_EV_CUBE, cn.cubes);
                            EV(RTX_EV_BOUNDS, cn.containers); EV(RTX_EV_CULL, n);
                        }
                        // ---- sweep: conservative bounding-sphere pre-test of every child (cull_hit4: no counterpart
                        // in the reference, cannot change a result).  Scalar loads return out of order (only
                        // lgkmcnt(0) can be waited for), so the loop is a ping-pong over QUADS of children: two
                        // s_load_dwordx16 are in flight per wait.  The pointers are opaque so the rotation stays.
                        unsigned cand = 0;
                        // (children wholly behind the ray origin are dropped too — see cull_hit2 — unless the chunk has containers
                        // and the scene has transparent materials)
                        // (... and, in the reference-faithful counting mode, the children a dropped container would have counted)
                        const float behind = (cont_mask == 0 || (!sc.has_transparency && !rp.no_early_out)) ? 0.f : -__builtin_inff();
                        if (hdr[0] & CHUNK_TWO_LEVEL) {
                            // ---- long lists (more than one chunk): the eight quads of the chunk have bounding spheres of their own
                            // (two loads); only the quads SOME lane's ray may touch are swept, two per wait.  A coherent wave —
                            // primary rays of one tile, their shadow rays — touches one or two quads of a chunk.
                            const int q0 = child_base >> 2, nq = (n + 3) >> 2;
                            const AS4 F16* sq = (const AS4 F16*)(uintptr_t)(rp.no_early_out ? sc.quad_cull : sc.quad_cull_tight) + 2 * chunk;
                            F16 sa = sq[0];
                            const AS4 F16* psb = sq + 1;
                            asm volatile("" : "+s"(psb));
                            const F16 sb = *psb;
                            unsigned ql = cull_hit4(sa, fox, foy, foz, fdx, fdy, fdz, fdd, behind) | (cull_hit4(sb, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << 4);
                            if (!act) ql = 0;
                            unsigned qmask = 0;
#pragma unroll
                            for (int b = 0; b < 8; b++) if (__any((ql >> b) & 1u)) qmask |= 1u << b;
                            qmask &= (1u << nq) - 1u;
                            unsigned acc = 0;
                            while (qmask != 0) {
                                const int ia = __builtin_ctz(qmask);
                                qmask &= qmask - 1;
                                const int ib = qmask != 0 ? __builtin_ctz(qmask) : ia;
                                qmask &= qmask - 1;                                  // (0 & -1 = 0)
                                const AS4 F16* pa = lq + q0 + ia;
                                const AS4 F16* pb = lq + q0 + ib;
                                asm volatile("" : "+s"(pa), "+s"(pb));
                                const F16 qa = *pa, qb = *pb;
                                acc |= cull_hit4(qa, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << (4 * ia);
                                acc |= cull_hit4(qb, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << (4 * ib);
                            }
                            cand = acc & (n >= 32 ? ~0u : ((1u << n) - 1u));
                        } else {
                            const int q0 = child_base >> 2, nq = (n + 3) >> 2;
                            unsigned acc = 0;
                            F16 qa = lq[q0];
                            for (int q = 0; q < nq; q += 2) {
                                const AS4 F16* pb = lq + q0 + q + 1;
                                asm volatile("" : "+s"(pb));
                                const F16 qb = *pb;                              // (array is padded)
                                acc |= cull_hit4(qa, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << (4 * q);
                                const AS4 F16* pa = lq + q0 + q + 2;
                                asm volatile("" : "+s"(pa));
                                qa = *pa;
                                // (branch-free so both loads stay ahead of the math: past the end the bits are masked off)
                                acc |= cull_hit4(qb, fox, foy, foz, fdx, fdy, fdz, fdd, behind) << ((4 * (q + 1)) & 31);
                            }
                            cand = acc & (n >= 32 ? ~0u : ((1u << n) - 1u));
                        }
                        if (!act) cand = 0;
                        TMT(tm_ffwd);
                        // ---- planes (OPF_INLINE): never culled, every lane tests them -> the exact test (plane.rs:16-23,
                        // only the y row of the local ray) runs wave-uniformly with the row in SGPRs
                        for (unsigned pm = plane_mask; pm != 0; pm &= pm - 1) {
                            const int k = __builtin_ctz(pm);
                            const int op = op0 >= 0 ? op0 + k : lops[child_base + k];
                            typedef double Row __attribute__((ext_vector_type(4)));
                            const Row row = *(const AS4 Row*)&(bodies + op)->inv[4];
                            const double loy = row[0] * po.x + row[1] * po.y + row[2] * po.z + row[3];
                            const double ldy = row[0] * pd.x + row[1] * pd.y + row[2] * pd.z;
                            if (alive >= level && !decided) {
                                EV(RTX_EV_X_XFORM, 1);
                                if (!(__builtin_fabs(ldy) < 1e-12)) {
                                    feed(1, -loy / ldy, 0.0, op);
                                    if (is_shadow && !rp.no_early_out && best_op >= 0 && best_t < shadow_dist) decided = true;
                                }
                            }
                        }
                        // ---- shapes: every lane walks ITS OWN surviving children; node records come through per-lane
                        // vector loads.  ObjectStore::intersect (object_store.rs:243-260) for one shape per lane.
                        unsigned sc_cand = decided ? 0u : (cand & shape_mask);
#ifdef RTX_DIAG_DIV
                        if (COUNT) {
                            // slot 2: (lane, candidate) pairs of this chunk; 3: 64 x the longest lane's candidates (the rounds the loop
                            // below runs if nobody stops early); 4: 64 x rounds really run; 5: lanes busy in those rounds
                            int pc = __builtin_popcount(sc_cand), mx = pc;
                            for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(mx, o, 64); mx = w > mx ? w : mx; }
                            ev[2] += pc; ev[3] += mx;             // (every lane adds the maximum: 64 x max per wave)
                        }
#endif
                        while (__any(sc_cand != 0)) {
                            DV(4, 1); DV(5, sc_cand != 0 ? 1 : 0);
                            if (sc_cand != 0) {
                                const int i = __builtin_ctz(sc_cand);
                                sc_cand &= sc_cand - 1;
                                const int idx = op0 >= 0 ? op0 + i : sc.list_ops[child_base + i];
                                const OpBody* gb = sc.bodies + idx;
                                const int k1 = gb->kind;
                                V3 lo = xf_point(gb->inv, gb->w33, gb->div != 0, po);
                                V3 ld = xf_vector(gb->inv, pd);
                                EV(RTX_EV_X_XFORM, 1);
                                int k = 0; double t0 = 0.0, t1 = 0.0;
                                if (k1 == OPK_SPHERE) {                   // sphere.rs:16-32
                                    EV(RTX_EV_X_SPHERE, 1);
                                    double a = dot3(ld, ld);
                                    double b = 2.0 * dot3(ld, lo);
                                    double c = dot3(lo, lo) - 1.0;
                                    double disc = b * b - 4.0 * a * c;
                                    if (!(disc < 0.0)) {
                                        double sq = __builtin_sqrt(disc);
                                        t0 = (-b - sq) / (2.0 * a);
                                        t1 = (-b + sq) / (2.0 * a);
                                        k = 2;
                                    }
                                } else if (k1 == OPK_PLANE) {             // plane.rs:16-23
                                    if (!(__builtin_fabs(ld.y) < 1e-12)) { t0 = -lo.y / ld.y; k = 1; }
                                } else {                                  // cube.rs:18-31
                                    EV(RTX_EV_X_BOX, 1);
                                    if (slabs(lo, ld, -1.0, -1.0, -1.0, 1.0, 1.0, 1.0, t0, t1)) k = 2;
                                }
                                feed(k, t0, t1, idx);
                                // is_shadowed (world.rs:203-206): any 0 <= t < distance decides
                                if (is_shadow && !rp.no_early_out && best_op >= 0 && best_t < shadow_dist) { decided = true; sc_cand = 0; }
                            }
                        }
                        TMT(tm_body);
                        // ---- containers that some lane may hit
                        const unsigned cc = decided ? 0u : (cand & cont_mask);
                        for (unsigned cm = cont_mask; cm != 0; cm &= cm - 1) {
                            const unsigned bit = cm & (0u - cm);
                            if (__any((cc & bit) != 0)) U |= bit;
                        }
                    }
                }
                if (U != 0) {
                    const int c = __builtin_ctz(U);
                    U &= U - 1;
                    // one container child: the first half of its record (ContRec: everything but a general transform's matrix)
                    // arrives with ONE batch of scalar loads — field-by-field access made the compiler wait five times in a row
                    typedef uint32_t U16 __attribute__((ext_vector_type(16)));
                    const AS4 U16* crp = (const AS4 U16*)(conts + (int)hdr[7] + __builtin_popcount(hdr[4] & ((1u << c) - 1u)));
                    U16 ra = crp[0], rb;
                    if (!CSG) {
                        // (both loads must be in flight before the first use: tie one element of each to an opaque point)
                        rb = crp[1];
                        uint32_t e0 = ra[5], e1 = rb[8];
                        asm volatile("" : "+s"(e0), "+s"(e1));
                        ra[5] = e0; rb[8] = e1;
                    }
                    // (scenes with CSG: a CSG root needs the first 64 bytes only, and the 16 scalar registers of the second load
                    // would be live across the interpreter — there the second half is fetched when a GROUP is visited)
                    auto f64_of = [](uint32_t lo_, uint32_t hi_) { return __builtin_bit_cast(double, (uint64_t)lo_ | ((uint64_t)hi_ << 32)); };
                    const int cflags = (int)ra[5];
                    // this lane's own pre-test of the child (the sweep's, recomputed: the candidate bits of a chunk do not survive
                    // a descent); a lane takes part only if ITS ray may hit, whichever lane made the wave come here
                    auto own_miss = [&]() {
                        sync_level(level);
                        Head ch = {};
                        ch[4] = ra[0]; ch[5] = ra[1]; ch[6] = ra[2]; ch[7] = ra[3];
                        if (rp.no_early_out) {                       // counting mode: the sphere around the bounds
                            typedef uint32_t U4 __attribute__((ext_vector_type(4)));
                            const U4 lc = *(const AS4 U4*)((const AS4 uint32_t*)crp + 60);
                            ch[4] = lc[0]; ch[5] = lc[1]; ch[6] = lc[2]; ch[7] = lc[3];
                        }
                        return cull_miss(ch, fox, foy, foz, fdx, fdy, fdz, fdd, (!sc.has_transparency && !rp.no_early_out) ? 0.f : -__builtin_inff());
                    };
                    if (CSG && (cflags & CONT_CSG)) {
                        if constexpr (CSG) {
                            const bool cmiss = own_miss();
                            // (a CSG root's record: `list` = offset of its program in csg_prog, first word of `t` = its length, in dwords)
                            const int poff = (int)ra[4], plen = (int)ra[6];
                            resume_at = (act && !cmiss) ? 0 : 0x7fffffff;
                            cnt = 0; starts = 0; mids = 0;
                            interpret(poff, plen);
                        }
                        continue;
                    }
                    // ---- Groups::local_intersect (group.rs:64-75): pre-test, ray into the group's space, bounds
                    if (CSG) rb = crp[1];
                    EV(RTX_EV_OP_SLOTS, 1);
                    const bool test = act && !own_miss();
                    DV(16, 1);                                  // (diag build) slot 16: 64 x group visits; 18: 64 x visits that run the bounds filter
                    if (__any(test)) DV(18, 1);
                    if (__any(test)) {
                        const int xf = CONT_XF(cflags);
                        V3 lo, ld;
                        if (xf == XF_GENERAL) {
                            const AS4 double* gi = (const AS4 double*)(crp + 2);      // the second half: inverse rows + w33
                            lo = xf_point(gi, gi[12], (cflags & CONT_DIV) != 0, po); ld = xf_vector(gi, pd);
                        } else if (xf == XF_TRANSLATE) {
                            lo = V3{po.x + f64_of(ra[6], ra[7]), po.y + f64_of(ra[8], ra[9]), po.z + f64_of(ra[10], ra[11])}; ld = pd;
                        } else { lo = po; ld = pd; }
                        if (test) EV(RTX_EV_X_XFORM, 1);
                        const double b0 = f64_of(ra[12], ra[13]), b1 = f64_of(ra[14], ra[15]), b2 = f64_of(rb[0], rb[1]),
                                     b3 = f64_of(rb[2], rb[3]), b4 = f64_of(rb[4], rb[5]), b5 = f64_of(rb[6], rb[7]);
                        // decision by the interval filter; the exact slab test (six divisions) only where a lane is undecided
                        const int fdec = (cflags & CONT_NOFILTER) ? -1 : slab_filter(lo, ld, b0, b1, b2, b3, b4, b5);
                        bool pass = test && fdec == 1;
                        const bool undecided = test && fdec < 0;
                        if (__any(undecided)) {
                            if (undecided) EV(RTX_EV_X_BOX, 1);
                            double tmin, tmax;
                            const bool ex = slabs(lo, ld, b0, b1, b2, b3, b4, b5, tmin, tmax);
                            pass = pass || (undecided && ex);
                        }
                        if (__any(pass)) {
                            // the children (level + 1) walk in this group's space
                            po = lo; pd = ld;
                            enter_level(level, xf, pass);
                            if (pass) alive = level + 1;
                            // lane-indexed uniform arrays: what the walk needs of this chunk when it comes back (no header reload)
                            if (lane == level) { stk_chunk = chunk; stk_U = (int)U; stk_cm = (int)hdr[4]; stk_next = (int)hdr[5]; stk_cb = (int)hdr[7]; }
                            level = level + 1;
                            chunk = (int)ra[4];
                            // the first chunk of the child list travels with the record
                            hdr = Head{rb[8], rb[9], rb[10], rb[11], rb[12], rb[13], rb[14], rb[15]};
                            have_hdr = true;
                            fresh = true;
                        }
                    }
                    continue;
                }
                // this chunk is done: next chunk of the list, or back to the parent list
                if ((int)hdr[5] >= 0) { chunk = (int)hdr[5]; fresh = true; continue; }
                if (level == 0) break;
                if (alive == level) alive = level - 1;
                level = level - 1;
                chunk = __builtin_amdgcn_readlane(stk_chunk, level);
                U = (unsigned)__builtin_amdgcn_readlane(stk_U, level);
                hdr[4] = (uint32_t)__builtin_amdgcn_readlane(stk_cm, level);
                hdr[5] = (uint32_t)__builtin_amdgcn_readlane(stk_next, level);
                hdr[7] = (uint32_t)__builtin_amdgcn_readlane(stk_cb, level);
            }
        }

        TM(tm_other);
#ifdef RTX_DIAG_PHASE
        // diagnostic: how often a wave's lanes are at BOTH kinds of shading in one iteration (ev pow = iterations with a
        // radiance lane, ev schlick = with a shadow lane, ev refr = with both, ev backface = all iterations)
        if (COUNT) {
            const bool anyR = __any(mode == MODE_RADIANCE), anyS = __any(mode == MODE_SHADOW);
            if (lane == 0) { ev[RTX_EV_POW] += anyR; ev[RTX_EV_SCHLICK] += anyS; ev[RTX_EV_REFR] += anyR && anyS; ev[RTX_EV_BACKFACE] += 1; }
        }
#endif
        // ================================================================= SHADE
        // the hit being shaded: this iteration's view of it.  HCTX: read back from LDS by the lanes whose shadow ray has just been
        // traced (a radiance lane computes a new one below), so nothing of it is live across the traversal; else the registers
        V3 hp = hp_r, eyev = eyev_r, nv = nv_r;
        if (HCTX && mode == MODE_SHADOW) {
            hp = V3{hctx[0], hctx[64], hctx[128]}; eyev = V3{hctx[192], hctx[256], hctx[320]}; nv = V3{hctx[384], hctx[448], hctx[512]};
        }
        bool returning = false;                 // lane has a finished colour to hand to its parent frame
        double rr = 0.0, rg = 0.0, rb = 0.0;
        bool lights_done = false;
        bool shading = false;                   // the lane is at a hit whose lights are being evaluated
        double sr = 0.0, sg = 0.0, sb = 0.0;    // surface colour of the hit being shaded (this iteration's view)
        bool newray = false;                    // lane launches (nro, nrd) next
        V3 nro = {0, 0, 0}, nrd = {0, 0, 0};

        DVW(6, mode == MODE_RADIANCE); DV(7, mode == MODE_RADIANCE ? 1 : 0);          // 6: 64 x iterations with a radiance lane, 7: those lanes
        DVW(8, mode == MODE_RADIANCE && best_op >= 0); DV(9, (mode == MODE_RADIANCE && best_op >= 0) ? 1 : 0);   // PreparedComputations: waves x 64 / lanes
        DVW(10, mode == MODE_SHADOW); DV(11, mode == MODE_SHADOW ? 1 : 0);            // lighting block: waves x 64 / lanes
        if (mode == MODE_RADIANCE) {
            if (primary) {
                primary = false;
#ifdef RTX_WAVE_TRACE
                if (false) {
#else
                if (hit_out && !PROBE) {
#endif
                    const int px = tx * RTX_TILE + (pl & 7);
                    const int row = rp.compact ? (bl * RTX_TILE + (pl >> 3)) : (band * RTX_TILE + (pl >> 3));
                    hit_out[(size_t)row * cam.hsize + px] = best_op >= 0 ? sc.hits[best_op].shape : -1;
                }
            }
            if (best_op < 0) {                  // world.rs:107
                returning = true;
            } else {
                // ---- PreparedComputations::new (world.rs:225-256)
                EV(RTX_EV_HIT, 1);
                hop = best_op;
                const HitRec hr = sc.hits[hop];
                hmat = hr.material;
                EV(RTX_EV_LEVELS, hr.path_len + 1);
                {
                    // the ray just traced
                    const V3 wo = {org_base[0], org_base[64], org_base[128]}, wd = {dir_base[0], dir_base[64], dir_base[128]};
                    hp = wo + wd * best_t;      // rays.rs:17-19 (t * direction is commutative)
                    eyev = neg(wd);
                }
                const V3 objp = to_object(sc, hr, hop, hp);
                V3 ln;
                if (hr.type == OPK_SPHERE) ln = objp;                            // sphere.rs:34-36
                else if (hr.type == OPK_PLANE) ln = V3{0.0, 1.0, 0.0};            // plane.rs:25-27
                else {                                                            // cube.rs:33-47
                    double ax = __builtin_fabs(objp.x), ay = __builtin_fabs(objp.y), az = __builtin_fabs(objp.z);
                    double maxc = __builtin_fmax(__builtin_fmax(ax, ay), az);
                    if (maxc == ax) ln = V3{objp.x, 0.0, 0.0};
                    else if (maxc == ay) ln = V3{0.0, objp.y, 0.0};
                    else ln = V3{0.0, 0.0, objp.z};
                }
                // normal_to_world (object_store.rs:331-349): shape -> root, normalised per level
                ln = normalize3(xf_normal((sc.bodies + hop)->inv, ln));
                for (int i = hr.path_len - 1; i >= 0; i--) {
                    const OpBody* xo = sc.bodies + sc.path_ops[hr.path_begin + i];
                    ln = normalize3(xf_normal(xo->inv, ln));
                }
                nv = ln;
                if (dot3(nv, eyev) < 0.0) nv = neg(nv);                           // world.rs:232-237
                if (HCTX) {
                    hctx[0] = hp.x; hctx[64] = hp.y; hctx[128] = hp.z; hctx[192] = eyev.x; hctx[256] = eyev.y; hctx[320] = eyev.z;
                    hctx[384] = nv.x; hctx[448] = nv.y; hctx[512] = nv.z;
                } else { hp_r = hp; eyev_r = eyev; nv_r = nv; }
                // get_refractive_indices_at (world.rs:258-296) from the running state: remember
                // WHICH shapes supply n1 / n2; the indices themselves are read when the children are decided
                n1op = c1s;
                n2op = (c1s >= 0 && best_inside) ? ((hop == c1s) ? c2s : c1s) : hop;
                light_k = 0;
                shading = true;
            }
            TML(tm_prep);
        } else if (mode == MODE_SHADOW) {
            // ---- lighting for light_k (material.rs:54-81) with the shadow answer of this trace
            const bool in_shadow = best_op >= 0 && best_t < shadow_dist;
            EV(RTX_EV_LIGHT, 1);
            const LightRec* L = sc.lights + light_k;
            const MaterialRec* m = sc.materials + hmat;
            // pattern colour at the hit (object_store.rs:293-299), evaluated per light like the reference
            double pcr, pcg, pcb;
            {
                const int pk = m->kind;
                V3 a = {m->a[0], m->a[1], m->a[2]};
                if (pk == RTX_PAT_PLAIN) { pcr = a.x; pcg = a.y; pcb = a.z; }
                else {
                    const HitRec hr = sc.hits[hop];
                    const V3 objp = to_object(sc, hr, hop, hp);
                    V3 pp = xf_point(m->pinv, m->pw33, m->pdiv != 0, objp);
                    V3 b = {m->b[0], m->b[1], m->b[2]};
                    if (pk == RTX_PAT_TEST) { pcr = pp.x; pcg = pp.y; pcb = pp.z; }
                    else if (pk == RTX_PAT_GRADIENT) {                         // patterns.rs:144-151
                        double fr = pp.x - __builtin_floor(pp.x);
                        pcr = a.x + (b.x - a.x) * fr; pcg = a.y + (b.y - a.y) * fr; pcb = a.z + (b.z - a.z) * fr;
                    } else {
                        int odd;
                        if (pk == RTX_PAT_STRIPES) odd = floor_parity(pp.x);                                    // :96-104
                        else if (pk == RTX_PAT_RINGS) odd = floor_parity(__builtin_sqrt(pp.x * pp.x + pp.z * pp.z));   // :112-121
                        else odd = floor_parity(pp.x) ^ floor_parity(pp.y) ^ floor_parity(pp.z);                // :128-136
                        V3 c = odd ? b : a;
                        pcr = c.x; pcg = c.y; pcb = c.z;
                    }
                }
            }
            DVW(12, m->kind != RTX_PAT_PLAIN); DV(13, m->kind != RTX_PAT_PLAIN ? 1 : 0);   // pattern look-up with its transforms: waves x 64 / lanes
            const double ir = L->intensity[0], ig = L->intensity[1], ib = L->intensity[2];
            const double er = pcr * ir, eg = pcg * ig, eb = pcb * ib;             // effective_color
            V3 lightv = normalize3(V3{L->pos[0], L->pos[1], L->pos[2]} - hp);
            const double amb = m->ambient;
            double cr = er * amb, cg = eg * amb, cb = eb * amb;                   // ambient
            double dr = 0.0, dg = 0.0, db = 0.0, spr = 0.0, spg = 0.0, spb = 0.0;
            const double ldn = dot3(lightv, nv);
            if (ldn < 0.0) EV(RTX_EV_BACKFACE, 1);
            DVW(14, !(in_shadow || ldn < 0.0)); DV(15, !(in_shadow || ldn < 0.0) ? 1 : 0);   // lit: diffuse + specular: waves x 64 / lanes
            if (!(in_shadow || ldn < 0.0)) {
                const double dif = m->diffuse;
                dr = er * dif * ldn; dg = eg * dif * ldn; db = eb * dif * ldn;
                V3 rv = reflect3(neg(lightv), nv);
                const double rde = dot3(rv, eyev);
                if (!(rde <= 0.0)) {
                    EV(RTX_EV_POW, 1);
                    const double factor = rtx_spec_pow(rde, m->shininess);
                    const double spec = m->specular;
                    spr = ir * spec * factor; spg = ig * spec * factor; spb = ib * spec * factor;
                }
            }
            cr = cr + dr + spr; cg = cg + dg + spg; cb = cb + db + spb;           // ambient + diffuse + specular
            // Sum::sum folds from black (color.rs:69-73); the partial sum waits in the frame slot being built
            if (light_k > 0) { const Frame& f = frames[sp]; sr = f.s[0]; sg = f.s[1]; sb = f.s[2]; }
            sr = sr + cr; sg = sg + cg; sb = sb + cb;
            light_k++;
            shading = true;
            TM(tm_light);
        }

        if (shading) {
            // ---- lights whose is_shadowed answer cannot matter: material.rs:67 ignores `in_shadow` when light_dot_normal < 0,
            // so a light that is CERTAINLY behind the surface contributes its ambient term whatever the trace would say.
            // Done at once for plain-coloured materials (the term is then three products); the is_shadowed call stays a
            // counted ray (reference-defined work) but is not traced.  Sign test without the normalisation: (pos - point) .
            // normal below a margin 1000x the rounding error of the exact expression => the exact ldn is negative too.
            const MaterialRec* m = sc.materials + hmat;
            if (!rp.no_early_out && m->kind == RTX_PAT_PLAIN) {
                while (light_k < n_lights) {
                    const LightRec* L = sc.lights + light_k;
                    const V3 lv = V3{L->pos[0], L->pos[1], L->pos[2]} - hp;
                    const double px_ = lv.x * nv.x, py_ = lv.y * nv.y, pz_ = lv.z * nv.z;
                    if (!((px_ + py_ + pz_) < -1e-12 * (__builtin_fabs(px_) + __builtin_fabs(py_) + __builtin_fabs(pz_)))) break;
                    n_shadow++; n_skipped++;
                    EV(RTX_EV_LIGHT, 1); EV(RTX_EV_BACKFACE, 1);
                    const double amb = m->ambient;
                    double cr = m->a[0] * L->intensity[0] * amb, cg = m->a[1] * L->intensity[1] * amb, cb = m->a[2] * L->intensity[2] * amb;
                    cr = cr + 0.0 + 0.0; cg = cg + 0.0 + 0.0; cb = cb + 0.0 + 0.0;      // ambient + diffuse + specular
                    sr = sr + cr; sg = sg + cg; sb = sb + cb;
                    light_k++;
                }
            }
            if (PROBE && light_k < n_lights) {
                // the probe only estimates how many rays a tile's pixels need: a shadow ray spawns nothing, so it is counted
                // and not traced (the probe is bound by the length of its longest chain of rays, which this halves)
                n_shadow += (unsigned)(n_lights - light_k);
                light_k = n_lights;
            }
            if (light_k >= n_lights) lights_done = true;
        }
        if (shading && !lights_done) {
            // ---- is_shadowed (world.rs:195-207): next shadow ray, from over_point
            if (light_k > 0) { Frame& f = frames[sp]; f.s[0] = sr; f.s[1] = sg; f.s[2] = sb; }   // the partial sum waits in the frame slot
            const LightRec* L = sc.lights + light_k;
            V3 over = hp + nv * 1e-10;                                            // world.rs:238-239
            V3 v = V3{L->pos[0], L->pos[1], L->pos[2]} - over;
            shadow_dist = norm3(v);
            nro = over; nrd = v / shadow_dist;                                    // normalize() recomputes the same norm
            newray = true;
            mode = MODE_SHADOW;
        }
        TMS(tm_ffwd);

        DV(17, lights_done ? 1 : 0);                                                  // children decision: lanes
        if (lights_done) {
            // ---- rest of shade_hit (world.rs:138-148): decide the children
            const MaterialRec* m = sc.materials + hmat;
            const double refl = m->reflective, transp = m->transparency;
            const bool do_refl = remaining != 0 && !(refl == 0.0);                // world.rs:155
            const bool want_refr = remaining != 0 && !(transp == 0.0);            // world.rs:167-170
            const bool both = refl > 0.0 && transp > 0.0;
            bool do_refr = false;
            V3 fo = {0, 0, 0}, fd = {0, 0, 0};
            double R = 0.0;
            if (want_refr || both) {
                // refracted_color (world.rs:174-188) and schlick (world.rs:304-323) evaluate the same
                // n1/n2, cos_i, sin2_t and cos_t expressions: computed once, same bits
                const double n1 = (n1op >= 0) ? sc.hits[n1op].ri : 1.0;
                const double n2 = (n2op >= 0) ? sc.hits[n2op].ri : 1.0;
                const double cos_i = dot3(eyev, nv);
                double sin2_t = 0.0, cos_t = 0.0, n_ratio = 0.0;
                if (want_refr || n1 > n2) {
                    n_ratio = n1 / n2;
                    sin2_t = n_ratio * n_ratio * (1.0 - cos_i * cos_i);
                    if (!(sin2_t > 1.0)) cos_t = __builtin_sqrt(1.0 - sin2_t);
                }
                if (want_refr) {
                    EV(RTX_EV_REFR, 1);
                    if (!(sin2_t > 1.0)) {
                        fd = nv * (n_ratio * cos_i - cos_t) - eyev * n_ratio;
                        fo = hp - nv * 1e-10;                                     // under_point
                        do_refr = true;
                    }
                }
                if (both) {
                    EV(RTX_EV_SCHLICK, 1);
                    double cosv = cos_i;
                    bool tir = false;
                    if (n1 > n2) {
                        if (sin2_t > 1.0) tir = true;
                        else cosv = cos_t;
                    }
                    if (tir) R = 1.0;
                    else {
                        const double q = (n1 - n2) / (n1 + n2);
                        const double r0 = q * q;
                        const double x = 1.0 - cosv;
                        const double x2 = x * x;
                        R = r0 + (1.0 - r0) * (x * (x2 * x2));
                    }
                }
            }
            if (!do_refl && !do_refr) {
                if (both) { rr = sr + 0.0 * R + 0.0 * (1.0 - R); rg = sg + 0.0 * R + 0.0 * (1.0 - R); rb = sb + 0.0 * R + 0.0 * (1.0 - R); }
                else { rr = sr + 0.0 + 0.0; rg = sg + 0.0 + 0.0; rb = sb + 0.0 + 0.0; }
                returning = true;
            } else {
                Frame& f = frames[sp];
                f.s[0] = sr; f.s[1] = sg; f.s[2] = sb;
                f.R = R;
                f.rem_mat = remaining | (hmat << 4);
                if (do_refl) {
                    if (SPLIT && do_refr) pend |= 1u << sp;
                    f.flags = FR_REFL | (do_refr ? FR_REFR_PENDING : 0) | (both ? FR_BOTH : 0);
                    if (do_refr) { PendRay& pr = prays[sp]; pr.o[0] = fo.x; pr.o[1] = fo.y; pr.o[2] = fo.z; pr.d[0] = fd.x; pr.d[1] = fd.y; pr.d[2] = fd.z; }
                    nro = hp + nv * 1e-10;                                        // over_point
                    nrd = reflect3(neg(eyev), nv);                                // reflectv (world.rs:251)
                } else {
                    f.flags = both ? FR_BOTH : 0;
                    nro = fo; nrd = fd;
                }
                newray = true;
                sp++;
                remaining = remaining - 1;
                mode = MODE_RADIANCE;
            }
        }

        TMS(tm_body);
        // ---- unwind: hand finished colours to the waiting frames (post-order)
        if (SPLIT) {
            // ---- hand refracted children to idle lanes of the same pixel group (uniform control flow: two ballots)
            const bool idle = mode == MODE_DONE && !waiting && !returning && mb_state[lane] == 0;
            // a lane offers the refracted child of its SHALLOWEST frame that still has one (the largest subtree)
            // ... and only when it is about to trace a radiance ray itself: lanes alternate radiance and shadow
            // traces, and a wave whose lanes do so in step runs one shading block per iteration instead of both
            const bool want = pend != 0 && mode == MODE_RADIANCE;
            const unsigned long long Dm = __ballot(want), Im = __ballot(idle);
            if (Dm != 0 && Im != 0) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const int n_don = (int)__builtin_popcountll(Dm), n_idle = (int)__builtin_popcountll(Im);
                if (want) {
                    // the k-th donor of the wave gets the k-th idle lane
                    int r = (int)__builtin_popcountll(Dm & below);
                    if (r < n_idle) {
                        int target = 0;
#pragma unroll
                        for (int w = 32; w > 0; w >>= 1) {
                            const int c = (int)__builtin_popcountll(Im & (((1ull << w) - 1ull) << target));
                            if (r >= c) { r -= c; target += w; }
                        }
                        const int level = __builtin_ctz(pend);
                        pend &= pend - 1u;
                        Frame& f = frames[level];
                        double* hs = lds_rays + target;            // the helper's ray slot 0
                        double* hd = lds_rays + sc.n_org_slots * LDS_HALF_DOUBLES + target;
                        const PendRay& pr = prays[level];
                        hs[0] = pr.o[0]; hs[64] = pr.o[1]; hs[128] = pr.o[2]; hd[0] = pr.d[0]; hd[64] = pr.d[1]; hd[128] = pr.d[2];
                        mb_rem[target] = (f.rem_mat & 15) - 1;
                        // (the helper reads its ray slot with plain loads: order the stores above before the flag, and see
                        // the acquire fence at the top of the trace)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        mb_state[target] = 1;
                        f.flags = (f.flags & ~FR_REFR_PENDING) | FR_REMOTE | (target << FR_HELPER_SHIFT);
                    }
                }
                if (idle && (int)__builtin_popcountll(Im & below) < n_don) {
                    // this idle lane was paired: start the subtree as a color_at call of its own
                    mode = MODE_RADIANCE;
                    remaining = mb_rem[lane];
                    sp = 0;
                    task_root = true;
                    primary = false;      // (a lane whose own pixel lies outside the image has never shaded)
                }
            }
            if (waiting) {
                // back at a frame with a remote child: has the helper posted the colour?
                const int h = wait_h;
                if (mb_state[h] == 2) {
                    rr = mb_[h]; rg = mb_[64 + h]; rb = mb_[128 + h];
                    mb_state[h] = 0;
                    waiting = false;
                    returning = true;     // the frame carries FR_SECOND: "refracted child finished" below
                }
            }
        }
        while (returning) {
            if (sp == 0) {
                if (SPLIT && task_root) {
                    // a handed-over child is done: post what color_at returned; the frame's owner takes it from here
                    mb_[lane] = rr; mb_[64 + lane] = rg; mb_[128 + lane] = rb;
                    mb_state[lane] = 2;
                    task_root = false;
                } else if (!PROBE) write_pixel(rr, rg, rb);
                mode = MODE_DONE;
                returning = false;
            } else {
                Frame& f = frames[sp - 1];
                const int fl = f.flags;
                const double R = f.R;
                const MaterialRec* fm = sc.materials + (f.rem_mat >> 4);
                double s0 = f.s[0], s1 = f.s[1], s2 = f.s[2];
                if (fl & FR_SECOND) {
                    // refracted child finished (world.rs:190-192); f.s already holds surface + reflected [* R]
                    const double transp = fm->transparency;
                    rr = rr * transp; rg = rg * transp; rb = rb * transp;
                    if (fl & FR_BOTH) { rr = s0 + rr * (1.0 - R); rg = s1 + rg * (1.0 - R); rb = s2 + rb * (1.0 - R); }
                    else { rr = s0 + rr; rg = s1 + rg; rb = s2 + rb; }
                } else if (fl & FR_REFL) {
                    // reflected child finished (world.rs:160-162): surface + reflected [* reflectance] ...
                    const double refl = fm->reflective;
                    double ar = rr * refl, ag = rg * refl, ab = rb * refl;
                    if (fl & FR_BOTH) { s0 = s0 + ar * R; s1 = s1 + ag * R; s2 = s2 + ab * R; }
                    else { s0 = s0 + ar; s1 = s1 + ag; s2 = s2 + ab; }
                    if (fl & FR_REFR_PENDING) {
                        if (SPLIT) pend &= ~(1u << (sp - 1));
                        f.s[0] = s0; f.s[1] = s1; f.s[2] = s2;
                        f.flags = fl | FR_SECOND;
                        { const PendRay& pr = prays[sp - 1]; nro = V3{pr.o[0], pr.o[1], pr.o[2]}; nrd = V3{pr.d[0], pr.d[1], pr.d[2]}; }
                        newray = true;
                        remaining = (f.rem_mat & 15) - 1;
                        mode = MODE_RADIANCE;
                        returning = false;
                        continue;
                    }
                    if (SPLIT && (fl & FR_REMOTE)) {
                        // the refracted child runs on a helper lane: park here until its colour is posted
                        f.s[0] = s0; f.s[1] = s1; f.s[2] = s2;
                        f.flags = fl | FR_SECOND;
                        waiting = true;
                        wait_h = (fl >> FR_HELPER_SHIFT) & 63;
                        mode = MODE_DONE;
                        returning = false;
                        continue;
                    }
                    // ... + refracted, which is black here
                    if (fl & FR_BOTH) { rr = s0 + 0.0 * (1.0 - R); rg = s1 + 0.0 * (1.0 - R); rb = s2 + 0.0 * (1.0 - R); }
                    else { rr = s0 + 0.0; rg = s1 + 0.0; rb = s2 + 0.0; }
                } else {
                    // the only child was the refracted ray; reflected is black
                    const double transp = fm->transparency;
                    rr = rr * transp; rg = rg * transp; rb = rb * transp;
                    if (fl & FR_BOTH) { rr = s0 + 0.0 * R + rr * (1.0 - R); rg = s1 + 0.0 * R + rg * (1.0 - R); rb = s2 + 0.0 * R + rb * (1.0 - R); }
                    else { rr = s0 + 0.0 + rr; rg = s1 + 0.0 + rg; rb = s2 + 0.0 + rb; }
                }
                sp--;
            }
        }

        // the next ray goes to LDS slot 0, where the traversal and the next shade read it
        if (newray) {
            org_base[0] = nro.x; org_base[64] = nro.y; org_base[128] = nro.z;
            dir_base[0] = nrd.x; dir_base[64] = nrd.y; dir_base[128] = nrd.z;
        }
        if ((hold_mask >> lane) & 1ull) mode = phaseR ? MODE_SHADOW : MODE_RADIANCE;     // the kind that sat out
    }

#ifdef RTX_TIMING
    if (lane == 0) {
        unsigned long long* counters = counters_base + (blockIdx.x % RTX_CNT_SLOTS) * RTX_CNT_STRIDE;
        atomicAdd(&counters[RTX_CNT_EV0 + 18], tm_ffwd);
        atomicAdd(&counters[RTX_CNT_EV0 + 19], tm_body);
        atomicAdd(&counters[RTX_CNT_EV0 + 20], tm_shade);
        atomicAdd(&counters[RTX_CNT_EV0 + 21], tm_other);
        atomicAdd(&counters[RTX_CNT_EV0 + 22], tm_prep);
        atomicAdd(&counters[RTX_CNT_EV0 + 23], tm_light);
    }
#endif
    arena_release();
    if (PROBE) {
        // probe launch: the rays this tile's centre pixel needed at the reduced depth -> cost class of the tile;
        // one list append per wave and class (wave-aggregated atomic)
        const int n_tiles = rp.tiles_x * rp.n_bands_local;
        const int sx_n = (rp.tiles_x + 7) >> 3;
        const int sb = (int)(((unsigned long long)blockIdx.x * (unsigned)rp.super_mul) % (unsigned)(sx_n * ((rp.n_bands_local + 7) >> 3)));
        const int ptx = (sb % sx_n) * 8 + (lane & 7), pbl = (sb / sx_n) * 8 + (lane >> 3);
        const bool pvalid = ptx < rp.tiles_x && pbl < rp.n_bands_local;
        // a tile next to an expensive one (the silhouette of a glass object) is probably expensive in a corner
        // its own centre pixel does not see: take the maximum over the 3x3 neighbourhood inside this wave's block
        int c = pvalid ? (int)(n_radiance + n_shadow) : 0;
        {
            const int l = __shfl(c, lane - 1, 64), r = __shfl(c, lane + 1, 64);
            int h = c;
            if ((lane & 7) > 0 && l > h) h = l;
            if ((lane & 7) < 7 && r > h) h = r;
            const int u = __shfl(h, lane - 8, 64), d = __shfl(h, lane + 8, 64);
            c = h;
            if (lane >= 8 && u > c) c = u;
            if (lane < 56 && d > c) c = d;
        }
        int cls = RTX_TILE_CLASSES == 8 ? (c >> 1) : c;
        cls = cls > RTX_TILE_CLASSES - 1 ? RTX_TILE_CLASSES - 1 : cls;
        if (!pvalid) cls = -1;
        const int t = pbl * rp.tiles_x + ptx;
        if (pvalid && rp.tile_cost) rp.tile_cost[t] = RTX_COST_OF_CLASS(cls);     // an estimate, until a frame launch measures the tile
        for (int k = 0; k < RTX_TILE_CLASSES; k++) {
            const unsigned long long m = __ballot(cls == k);
            if (m == 0) continue;
            const int leader = (int)__builtin_ctzll(m);
            int base = 0;
            if (lane == leader) base = atomicAdd(&rp.tile_counts[k], (int)__builtin_popcountll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            if (cls == k) rp.tile_lists[(size_t)k * n_tiles + base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = t;
        }
        return;
    }
#ifdef RTX_WAVE_TRACE
    if (hit_out) {
        unsigned it = wt_iters;
        for (int o = 32; o > 0; o >>= 1) { unsigned v = __shfl_xor(it, o, 64); it = v > it ? v : it; }
        if (lane == 0) {
            const size_t wt_i = blockIdx.x + (SPLIT ? (size_t)rp.tiles_x * rp.n_bands_local : 0);
            hit_out[3 * wt_i] = (int)wt_start;
            hit_out[3 * wt_i + 1] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
            hit_out[3 * wt_i + 2] = (int)(it | (S ? 0x10000u : 0u) | ((unsigned)wt_class << 17));
        }
    }
#endif
#ifdef RTX_PIXEL_COST
    // diagnostic build (tools/pixel_cost.py): the hit-id plane receives every pixel's traced rays instead of its hit id
    if (hit_out && !SPLIT) {
        const int px = tx * RTX_TILE + (pl & 7), py = band * RTX_TILE + (pl >> 3);
        const int row = rp.compact ? (bl * RTX_TILE + (pl >> 3)) : py;
        if (px < cam.hsize && py < cam.vsize) hit_out[(size_t)row * cam.hsize + px] = (int)(n_radiance + n_shadow - n_skipped);
    }
#endif
    // ---- tile-cost feedback: the traced rays of the tile's busiest pixel, for the next frame's tile order (split tiles keep
    // their last figure: their pixels' rays are spread over helper lanes).  The array's address waits in a spare counter word.
    // ---- ray / event counters: one atomic per wave per counter
    unsigned t0 = wave_sum(n_radiance), t1 = wave_sum(n_shadow), t2 = wave_sum(overflow), t3 = wave_sum(n_skipped);
    counters += (blockIdx.x % RTX_CNT_SLOTS) * RTX_CNT_STRIDE;      // this wave's replica (rtx_device.h)
    if (!SPLIT) tile_cost_tail(counters_base, (int)(n_radiance + n_shadow - n_skipped), *(volatile unsigned*)lds_all, bl * rp.tiles_x + tx);
    if (lane == 0) {
        atomicAdd(&counters[RTX_CNT_RADIANCE], (unsigned long long)t0);
        atomicAdd(&counters[RTX_CNT_SHADOW], (unsigned long long)t1);
        if (t3) atomicAdd(&counters[RTX_CNT_SKIPPED], (unsigned long long)t3);
        if (t2 || guard_hit) {
            // never silently truncated: the count goes to the counters (synchronous renders read them) and the
            // context's sticky host word is raised (asynchronous renders: checked by the next call / rtx_ctx_poll)
            atomicAdd(&counters[RTX_CNT_OVERFLOW], (unsigned long long)t2 + (guard_hit ? RTX_GUARD_UNIT : 0ull));
            if (rp.sticky) __hip_atomic_store(rp.sticky, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (COUNT) {
#pragma unroll
        for (int i = 0; i < RTX_EV_USED; i++) {
            unsigned s = wave_sum(ev[i]);
            if (lane == 0 && s) atomicAdd(&counters[RTX_CNT_EV0 + i], (unsigned long long)s);
        }
    }
#undef EV
#undef lt
#undef ls
}

// ---------------------------------------------------------------- launcher (called by rtx_api.hip)
extern "C" hipError_t rtx_launch_trace_shade(const SceneDev* sc, const CameraDev* cam, const RenderParams* rp,
                                             uint8_t* rgb, int32_t* hit, unsigned long long* counters,
                                             int count_events, hipStream_t stream) {
    int blocks = rp->probe ? ((rp->tiles_x + 7) / 8) * ((rp->n_bands_local + 7) / 8) : rp->tiles_x * rp->n_bands_local;
    if (blocks <= 0) return hipSuccess;
    const bool split = !rp->probe && rp->use_lists && rp->split_cap > 0 && rp->split_launch;
    if (split) blocks = rp->split_cap << RTX_SPLIT_BITS;
    size_t lds = 8 + (size_t)(sc->n_org_slots + sc->n_dir_slots) * LDS_HALF_DOUBLES * sizeof(double) + (((size_t)sc->csg_lds_entries * 64 * 12 + 15) & ~(size_t)15) +
                 (size_t)sc->csg_prog_lds * sizeof(uint32_t);
    if (split) lds += 64 * (3 * sizeof(double) + 2 * sizeof(int));
    // hit context in LDS (template flag HCTX): only if the wave's LDS still fits the occupancy the kernel is built for
    // (4 waves per SIMD = 16 per CU of 160 KB without CSG, 3 = 12 with; 256 bytes of slack per wave for the allocation granule)
    const size_t hctx_bytes = 8 + 9 * 64 * sizeof(double);
    const size_t waves_cu = 4 * (size_t)(sc->has_csg ? RTX_MIN_WAVES_CSG : RTX_MIN_WAVES);
    const bool general = sc->has_csg == 2;       // (the general CSG kernels are built without SPLIT and HCTX variants)
    if (general && split) return hipErrorInvalidValue;
    const bool hctx = !general && (rp->hctx >= 0 ? rp->hctx != 0 : (lds + hctx_bytes + 256) * waves_cu <= (size_t)160 * 1024);
    if (hctx) lds += hctx_bytes;
    dim3 grid(blocks), block(64);
    const bool csg = sc->has_csg != 0;
    const bool count = count_events != 0;
#define RTX_LAUNCH(C, N, P, S, H) hipLaunchKernelGGL((trace_shade_kernel<C, N, P, S, H>), grid, block, lds, stream, *sc, *cam, *rp, rgb, hit, counters)
#define RTX_PICK_H(C, N, P, S) do { if (hctx) RTX_LAUNCH(C, N, P, S, true); else RTX_LAUNCH(C, N, P, S, false); } while (0)
#define RTX_PICK_C(N, P, S) do { if (csg) RTX_PICK_H(1, N, P, S); else RTX_PICK_H(0, N, P, S); } while (0)
    if (general) {
        if (rp->probe) RTX_LAUNCH(2, false, true, false, false);
        else if (count) RTX_LAUNCH(2, true, false, false, false);
        else RTX_LAUNCH(2, false, false, false, false);
    } else
    if (rp->probe) RTX_PICK_C(false, true, false);
    else if (split) { if (count) RTX_PICK_C(true, false, true); else RTX_PICK_C(false, false, true); }
    else { if (count) RTX_PICK_C(true, false, false); else RTX_PICK_C(false, false, false); }
#undef RTX_PICK_C
#undef RTX_PICK_H
#undef RTX_LAUNCH
    return hipGetLastError();
}

// ---------------------------------------------------------------- tile-cost feedback: classify launch (called by rtx_api.hip)
// Files every tile of the launch under the class of its cost (rtx_device.h: RTX_CLASS_OF_COST) — what the probe launch does
// from its own estimate, here from the costs the previous frame of the same view measured.  One thread per tile, one
// wave-aggregated append per wave and class.
// time mode (uniform scenes: the cost is the wave's duration): classes by the ratio to the launch's MEAN cost, which
// cost_sum_kernel leaves in *sum first
__global__ __launch_bounds__(256) void cost_sum_kernel(const int32_t* __restrict__ cost, int n_tiles, unsigned long long* __restrict__ sum) {
    __shared__ unsigned long long part[4];
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long v = t < n_tiles ? (unsigned long long)(unsigned)cost[t] : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long a = 0; for (int i = 0; i < 4; i++) a += part[i]; atomicAdd(sum, a); }
}

__global__ __launch_bounds__(256) void classify_tiles_kernel(const int32_t* __restrict__ cost, int n_tiles, int32_t* __restrict__ lists, int32_t* __restrict__ counts,
                                                              const unsigned long long* __restrict__ sum) {
    // (all waves of a launch appending to the same eight counters serialise at ~12 ns per atomic — 60 us for a 4K frame; a block
    // of 256 tiles first counts in LDS, then reserves its ranges with ONE global atomic per class.  Blocks of four waves on
    // purpose: the launch runs beside frame launches that keep every wave slot busy, and a 1024-thread block would have to wait
    // until ONE compute unit has sixteen free slots at the same moment — measured 0.8 ms on average in a streamed run)
    __shared__ int cnt[RTX_TILE_CLASSES], base[RTX_TILE_CLASSES];
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int lane = (int)(threadIdx.x & 63);
    if (threadIdx.x < RTX_TILE_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    int cls = -1;
    if (t < n_tiles) {
        if (sum) {
            // 16 x cost / mean -> class: < 0.6, < 0.85, < 1.1, < 1.4, < 1.8, < 2.4, < 3.5, more
            const unsigned long long mean16 = (*sum + (unsigned long long)n_tiles - 1) / (unsigned long long)n_tiles;
            const unsigned long long r = mean16 ? ((unsigned long long)(unsigned)cost[t] * 160ull) / mean16 : 0ull;     // 10 x ratio x 16
            cls = r < 96 ? 0 : r < 136 ? 1 : r < 176 ? 2 : r < 224 ? 3 : r < 288 ? 4 : r < 384 ? 5 : r < 560 ? 6 : 7;
        } else cls = RTX_CLASS_OF_COST(cost[t]);
    }
    int off = 0;                                   // this tile's position among the block's tiles of its class
    for (int k = 0; k < RTX_TILE_CLASSES; k++) {
        const unsigned long long m = __ballot(cls == k);
        if (m == 0) continue;
        const int leader = (int)__builtin_ctzll(m);
        int wbase = 0;
        if (lane == leader) wbase = atomicAdd(&cnt[k], (int)__builtin_popcountll(m));
        wbase = __builtin_amdgcn_readlane(wbase, leader);
        if (cls == k) off = wbase + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < RTX_TILE_CLASSES) base[threadIdx.x] = cnt[threadIdx.x] > 0 ? atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]) : 0;
    __syncthreads();
    if (cls >= 0) lists[(size_t)cls * n_tiles + base[cls] + off] = t;
}

extern "C" hipError_t rtx_launch_classify_tiles(const int32_t* cost, int n_tiles, int32_t* lists, int32_t* counts, unsigned long long* time_sum,
                                                hipStream_t stream) {
    static_assert(RTX_TILE_CLASSES == 8, "RTX_CLASS_OF_COST files into 8 classes");
    if (n_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_tiles + 255) / 256)), block(256);
    if (time_sum) {
        hipError_t e = hipMemsetAsync(time_sum, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(cost_sum_kernel, grid, block, 0, stream, cost, n_tiles, time_sum);
    }
    hipLaunchKernelGGL(classify_tiles_kernel, grid, block, 0, stream, cost, n_tiles, lists, counts, (const unsigned long long*)time_sum);
    return hipGetLastError();
}

// zeroes one replicated counter block and leaves the tile-cost array's address (or 0) in its spare word (rtx_device.h)
__global__ void reset_counters_kernel(unsigned long long* counters, unsigned long long cost_ptr) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < RTX_CNT_SLOTS * RTX_CNT_STRIDE) counters[i] = i == RTX_CNT_COST_PTR ? cost_ptr : 0ull;
}

extern "C" hipError_t rtx_launch_reset_counters(unsigned long long* counters, const int32_t* tile_cost, hipStream_t stream) {
    hipLaunchKernelGGL(reset_counters_kernel, dim3((RTX_CNT_SLOTS * RTX_CNT_STRIDE + 255) / 256), dim3(256), 0, stream, counters,
                       (unsigned long long)(uintptr_t)tile_cost);
    return hipGetLastError();
}
