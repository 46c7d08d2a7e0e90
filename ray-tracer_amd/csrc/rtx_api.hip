// rtx_api.hip — implementation of the C ABI in include/rtx.h: scene compiler
// (rtx_scene_desc -> traversal program), device memory management, launch, stats.
//
// There is NO CPU fallback: every entry point fails with RTX_E_DEVICE when HIP
// cannot provide a device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_device.h"
#include "rtx_internal.h"
#include "rtx_pow.h"

extern "C" hipError_t rtx_launch_trace_shade(const SceneDev* sc, const CameraDev* cam, const RenderParams* rp,
                                             uint8_t* rgb, int32_t* hit, unsigned long long* counters,
                                             int count_events, hipStream_t stream);
extern "C" hipError_t rtx_launch_reset_counters(unsigned long long* counters, const int32_t* tile_cost, hipStream_t stream);
extern "C" hipError_t rtx_launch_classify_tiles(const int32_t* cost, int n_tiles, int32_t* lists, int32_t* counts, unsigned long long* time_sum,
                                                hipStream_t stream);

__global__ void debug_math_kernel(int op, long long n, const double* a, const double* b, double* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = a[i], y = b[i], r;
    switch (op) {
        case 0: r = __builtin_sqrt(x); break;
        case 1: r = x / y; break;
        case 2: r = pow(x, y); break;
        case 3: r = __builtin_floor(x); break;
        case 4: r = __builtin_round(x); break;
        case 5: r = __builtin_fmax(x, y); break;
        case 7: r = rtx_spec_pow(x, y); break;
        case 8: r = __builtin_amdgcn_rcp(x); break;                           // raw v_rcp_f64 (accuracy probe for the slab filter)
        default: r = __builtin_fmin(x, y); break;
    }
    out[i] = r;
}

// ---- the library's no-alloc scratch (rtx_render_device_noalloc): counters + tile-order lists of ONE render in a
// file-scope device array, so that entry point never calls hipMalloc.  Every GPU the library runs on holds its own
// copy at its own address: rtx_init() asks for it with hipGetSymbolAddress and keeps the pointer in the context
// (host code must never pass the array's NAME to a kernel or a memset: on the host it is only a stand-in).
#define RTX_NOALLOC_COUNTER_BYTES ((size_t)RTX_CNT_SLOTS * RTX_CNT_STRIDE * sizeof(unsigned long long))   /* 64 KiB */
#define RTX_NOALLOC_COUNTS_BYTES  256                                                                    /* tile_counts */
#define RTX_NOALLOC_LISTS_BYTES   ((size_t)RTX_TILE_CLASSES * RTX_NOALLOC_MAX_TILES * sizeof(int32_t))    /* 4 MiB */
#define RTX_NOALLOC_SCRATCH_BYTES (RTX_NOALLOC_COUNTER_BYTES + RTX_NOALLOC_COUNTS_BYTES + RTX_NOALLOC_LISTS_BYTES)
static __device__ __align__(256) unsigned char scratch[RTX_NOALLOC_SCRATCH_BYTES];

int rtx_earlier_overflow(rtx_ctx* ctx);

// The host's share of rtx_render's band-set pipeline: a landed set's bands go from pinned staging to their rows of the caller's
// (pageable) image.  One thread moves ~10 GB/s, which made the host the bottleneck of an 8K frame (99.5 MB) and left the last
// set's copy exposed on every frame; a few persistent workers share the copies of a set (the caller's thread takes part).
struct rtx_copy_pool {
    struct Job { const uint8_t* src; uint8_t* dst; size_t n; };
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable wake, done;
    const std::vector<Job>* jobs = nullptr;
    std::atomic<size_t> next{0};
    size_t acks = 0;                         // workers that have finished with the current generation's job list
    unsigned long long generation = 0;
    bool stop = false;

    explicit rtx_copy_pool(int n) {
        for (int i = 0; i < n; i++) workers.emplace_back([this] { loop(); });
    }
    ~rtx_copy_pool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        wake.notify_all();
        for (std::thread& t : workers) t.join();
    }
    void drain(const std::vector<Job>& js) {
        for (size_t i; (i = next.fetch_add(1)) < js.size();) std::memcpy(js[i].dst, js[i].src, js[i].n);
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            const std::vector<Job>* js;
            {
                std::unique_lock<std::mutex> g(m);
                wake.wait(g, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation; js = jobs;
            }
            drain(*js);
            { std::lock_guard<std::mutex> g(m); acks++; }
            done.notify_one();
        }
    }
    // copies every job (they must not overlap); returns when all are done AND every worker has let go of the list
    void run(const std::vector<Job>& js) {
        if (js.empty()) return;
        { std::lock_guard<std::mutex> g(m); jobs = &js; next = 0; acks = 0; generation++; }
        wake.notify_all();
        drain(js);
        std::unique_lock<std::mutex> g(m);
        done.wait(g, [&] { return acks == workers.size(); });
        jobs = nullptr;
    }
};

namespace {

int fail(rtx_ctx* c, int code, const std::string& msg) { return rtx_fail(c, code, msg); }

// column-major 4x4 -> rows 0..2 (row-major 3x4) + w33; false if the bottom row is not (0,0,0,*)
bool affine_rows(const double* m, double* rows12, double* w33) {
    for (int c = 0; c < 3; c++) if (!(m[c * 4 + 3] == 0.0)) return false;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) rows12[r * 4 + c] = m[c * 4 + r];
    *w33 = m[15];
    return true;
}

// general 4x4 inverse (Gauss-Jordan, partial pivoting) of a column-major matrix; used only to
// recover forward transforms for the conservative bounding spheres, never for rendering arithmetic
bool invert_general(const double* m, double* out) {
    double a[4][8];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int col = 0; col < 4; col++) {
        int piv = col;
        for (int r = col + 1; r < 4; r++) if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (!(std::fabs(a[piv][col]) > 0.0) || !std::isfinite(a[piv][col])) return false;
        if (piv != col) for (int c = 0; c < 8; c++) std::swap(a[piv][c], a[col][c]);
        const double d = a[col][col];
        for (int c = 0; c < 8; c++) a[col][c] /= d;
        for (int r = 0; r < 4; r++) if (r != col) { const double f = a[r][col]; if (f != 0.0) for (int c = 0; c < 8; c++) a[r][c] -= f * a[col][c]; }
    }
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[c * 4 + r] = a[r][4 + c];
    return true;
}

// Bounding sphere (parent space) of a node whose local extent is the sphere |p - lc| <= lr:
// centre = T*lc, radius = sigma_max(M) * lr (upper bound), inflated for the f32 pre-test.
void cull_sphere(const double* inverse16, const double* lc, double lr, float* out4) {
    out4[0] = out4[1] = out4[2] = 0.f; out4[3] = INFINITY;
    double fwd[16];
    if (!std::isfinite(lr) || !invert_general(inverse16, fwd)) return;
    const double w = fwd[15];
    if (!(w != 0.0)) return;
    double c[3], m[3][3];
    for (int r = 0; r < 3; r++) {
        c[r] = (fwd[0 * 4 + r] * lc[0] + fwd[1 * 4 + r] * lc[1] + fwd[2 * 4 + r] * lc[2] + fwd[3 * 4 + r]) / w;
        for (int k = 0; k < 3; k++) m[r][k] = fwd[k * 4 + r] / w;
    }
    // upper bound of the largest singular value: Gershgorin on M^T M and on M M^T (same eigenvalues;
    // one of them is diagonal for scale-then-rotate or rotate-then-scale transforms, so the bound is tight)
    double ga = 0.0, gb = 0.0;
    for (int i = 0; i < 3; i++) {
        double ra = 0.0, rb = 0.0;
        for (int j = 0; j < 3; j++) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 3; k++) { a += m[k][i] * m[k][j]; b += m[i][k] * m[j][k]; }
            ra += std::fabs(a); rb += std::fabs(b);
        }
        ga = std::fmax(ga, ra); gb = std::fmax(gb, rb);
    }
    const double fro = std::fmin(ga, gb);     // sigma_max^2 <= fro
    const double cn = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    // 0.1 % on the radius, plus the f32 rounding of centre and ray origin (6e-8 relative to their size)
    const double R = std::sqrt(fro) * lr * 1.001 + 1e-6 * cn + 1e-30;
    const double r2 = R * R;
    if (!std::isfinite(r2) || !std::isfinite(cn) || r2 > 1e30) return;
    for (int k = 0; k < 3; k++) out4[k] = (float)c[k];
    out4[3] = std::nextafterf((float)r2, INFINITY);
}

// compile-time view of one op (head + body); split into the two device arrays at upload
struct OpRec {
    int32_t kind, level, skip, flags, shape, clevel, xf, regray;
    float cull[4];       // bounding sphere from the node's own extent / its local bounds (what the reference's bounds test implies)
    float cull_t[4];     // the product's sphere: shapes = cull; containers: around the children themselves (geom_sphere), CSG: around
                         // where SURVIVING hits can lie (extent_sphere) when that is tighter
    double inv[12], w33, bounds[6];
};

struct Compiler {
    const rtx_scene_desc* d;
    std::vector<OpRec> ops;
    std::vector<ShapeRec> shapes;
    std::vector<int32_t> paths;
    std::vector<int32_t> stack;      // op indices of the ancestors (root first)
    std::vector<char> seen;
    int max_container_level = -1;
    int org_depth = 0, dir_depth = 0, max_org = 0, max_dir = 0;   // LDS slots along the current path / maxima
    bool has_csg = false;
    bool general = false;            // some CSG root exceeds the bounded (LDS) structures: the scene renders with the general CSG kernels
    int max_csg_entries = 0;
    int max_clevel = 0;              // deepest container level inside a CSG root
    int err = RTX_OK;
    std::string msg;

    bool bad(int code, const std::string& m) { if (err == RTX_OK) { err = code; msg = m; } return false; }

    OpRec make(int node, int kind, int level, int flags, int clevel) {
        OpRec o; std::memset(&o, 0, sizeof o);
        o.kind = kind; o.level = level; o.flags = flags; o.clevel = clevel;
        if (node >= 0) {
            if (!affine_rows(d->node_inverse + (size_t)node * 16, o.inv, &o.w33))
                bad(RTX_E_UNSUPPORTED, "node " + std::to_string(node) + ": transform is not affine");
            if (!(o.w33 == 1.0)) o.flags |= OPF_DIV;
            // XF_* class from the exact bits of the inverse (rows 0..2 + w33)
            bool lin_id = o.w33 == 1.0;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) lin_id = lin_id && (o.inv[r * 4 + c] == (r == c ? 1.0 : 0.0));
            const bool t0 = o.inv[3] == 0.0 && o.inv[7] == 0.0 && o.inv[11] == 0.0;
            o.xf = !lin_id ? XF_GENERAL : (t0 ? XF_IDENTITY : XF_TRANSLATE);
            if (d->node_bounds) std::memcpy(o.bounds, d->node_bounds + (size_t)node * 6, sizeof o.bounds);
        }
        o.cull[0] = o.cull[1] = o.cull[2] = 0.f; o.cull[3] = INFINITY;
        if (node >= 0 && err == RTX_OK) {
            const double* inv16 = d->node_inverse + (size_t)node * 16;
            const double zero[3] = {0.0, 0.0, 0.0};
            if (kind == OPK_SPHERE) cull_sphere(inv16, zero, 1.0, o.cull);
            else if (kind == OPK_CUBE) cull_sphere(inv16, zero, std::sqrt(3.0), o.cull);
            else if (kind == OPK_GROUP_BEGIN || kind == OPK_CSG_BEGIN) {
                const double* b = o.bounds;
                bool fin = true;
                for (int k = 0; k < 6; k++) fin = fin && std::isfinite(b[k]);
                if (fin) {
                    const double lc[3] = {(b[0] + b[3]) * 0.5, (b[1] + b[4]) * 0.5, (b[2] + b[5]) * 0.5};
                    const double hx = (b[3] - b[0]) * 0.5, hy = (b[4] - b[1]) * 0.5, hz = (b[5] - b[2]) * 0.5;
                    cull_sphere(inv16, lc, std::sqrt(hx * hx + hy * hy + hz * hz), o.cull);
                }
            }
        }
        std::memcpy(o.cull_t, o.cull, sizeof o.cull);
        return o;
    }

    // The product's pre-test sphere of a container (OpRec::cull_t) must bound its GEOMETRY, not the reference's bounds: the
    // kernel tests it against the HALF-line t >= 0 too (children wholly behind the ray origin), and the reference's group
    // bounds need not contain the children — Bounds::from_points (bounds.rs:29-40) seeds `max` with the second corner and
    // never looks at the first, and a plane's infinite box turns into NaN / garbage under a rotation — while its bounds test
    // (bounds.rs:42-54) is one against the whole line.  A ray that misses the sphere around the bounds gets an empty list
    // from the reference at any t (that sphere, OpRec::cull, serves the whole-line test of the counting mode); a HALF-line
    // that misses a sphere around the children themselves cannot hit a child at t >= 0, whatever the bounds say.  Found by
    // the randomized sweep (tests/sweep_fuzz.py): 24 of 1 500 scenes, all with rotated groups, differed in a few pixels.
    // Children = the ops between b + 1 and the end of `ops`, hopping over nested containers (whose spheres are final).
    void geom_sphere(int node, int b) {
        float* out = ops[b].cull_t;
        out[0] = out[1] = out[2] = 0.f; out[3] = INFINITY;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::vector<const float*> ch;
        for (size_t j = (size_t)b + 1; j < ops.size();) {
            const OpRec& o = ops[j];
            const bool cont = o.kind == OPK_GROUP_BEGIN || o.kind == OPK_CSG_BEGIN;
            if (o.kind <= OPK_CUBE || cont) {
                if (!std::isfinite(o.cull_t[3])) return;                 // a plane (or a container with one) below: unbounded
                ch.push_back(o.cull_t);
                const double r = std::sqrt((double)o.cull_t[3]);
                for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], o.cull_t[k] - r); hi[k] = std::fmax(hi[k], o.cull_t[k] + r); }
            }
            j = cont ? (size_t)o.skip : j + 1;
        }
        if (ch.empty()) return;                                          // (an empty group has no hits; it is never visited for long)
        // enclosing sphere of the children's spheres: start at the centre of their joint box, then walk the centre towards the
        // farthest child with shrinking steps (Badoiu-Clarkson), keeping the best centre seen
        auto radius_at = [&](const double* c, int* far) {
            double R = 0.0;
            for (size_t i = 0; i < ch.size(); i++) {
                const double dx = ch[i][0] - c[0], dy = ch[i][1] - c[1], dz = ch[i][2] - c[2];
                const double ri = std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)ch[i][3]);
                if (ri > R) { R = ri; if (far) *far = (int)i; }
            }
            return R;
        };
        double c[3] = {(lo[0] + hi[0]) * 0.5, (lo[1] + hi[1]) * 0.5, (lo[2] + hi[2]) * 0.5};
        int far = 0;
        double R = radius_at(c, &far), w[3] = {c[0], c[1], c[2]};
        for (int it = 1; it <= 48 && ch.size() > 1; it++) {
            // towards the farthest POINT of the farthest child: its centre pushed out by its radius
            double dx = ch[far][0] - w[0], dy = ch[far][1] - w[1], dz = ch[far][2] - w[2];
            const double dist = std::sqrt(dx * dx + dy * dy + dz * dz), rf = std::sqrt((double)ch[far][3]);
            const double f = dist > 0.0 ? (dist + rf) / dist : 0.0;
            w[0] += dx * f / (it + 1); w[1] += dy * f / (it + 1); w[2] += dz * f / (it + 1);
            const double Rw = radius_at(w, &far);
            if (Rw < R) { R = Rw; c[0] = w[0]; c[1] = w[1]; c[2] = w[2]; }
        }
        cull_sphere(d->node_inverse + (size_t)node * 16, c, R * 1.0001, out);
    }

    // Where can hits that SURVIVE a CSG's filter (csg.rs:94-136) lie?  Union: on either child; intersection: left hits inside
    // the right child and right hits inside the left one, i.e. inside BOTH; difference: left hits outside the right child and
    // right hits inside the left one, i.e. inside the LEFT child.  A ray that misses a bounding sphere of that region gets an
    // empty list from the CSG (for the intersection and the difference: it has no hit on that child, so the in/out toggle the
    // other child's hits depend on never flips).  The children's spheres (in the CSG's space) give a sphere of the region,
    // often far tighter than the one around the CSG's bounds, which merge BOTH children's boxes whatever the operator
    // (object_store.rs:373-384).  Used for the pre-tests only where reference-faithful node-visit counts are not asked for.
    void extent_sphere(int node, int begin, int left, int right, int cop) {
        const float* a = ops[left].cull_t; const float* b = ops[right].cull_t;
        const bool fa = std::isfinite(a[3]), fb = std::isfinite(b[3]);
        double c[3], r;
        auto take = [&](const float* s) { c[0] = s[0]; c[1] = s[1]; c[2] = s[2]; r = std::sqrt((double)s[3]); };
        if (cop == RTX_CSG_DIFFERENCE) { if (!fa) return; take(a); }
        else if (cop == RTX_CSG_INTERSECTION) { if (!fa && !fb) return; take((fa && (!fb || a[3] <= b[3])) ? a : b); }
        else {
            if (!fa || !fb) return;
            const double ra = std::sqrt((double)a[3]), rb = std::sqrt((double)b[3]);
            const double dx = (double)b[0] - a[0], dy = (double)b[1] - a[1], dz = (double)b[2] - a[2];
            const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (dist + rb <= ra) take(a);
            else if (dist + ra <= rb) take(b);
            else {
                r = (dist + ra + rb) * 0.5;
                const double f = (r - ra) / dist;
                c[0] = a[0] + dx * f; c[1] = a[1] + dy * f; c[2] = a[2] + dz * f;
            }
        }
        float t4[4];
        cull_sphere(d->node_inverse + (size_t)node * 16, c, r * 1.0001, t4);
        if (std::isfinite(t4[3]) && (!std::isfinite(ops[begin].cull_t[3]) || t4[3] < ops[begin].cull_t[3])) std::memcpy(ops[begin].cull_t, t4, sizeof t4);
    }

    // true if every container BELOW `node` has exactly the identity as its inverse transform (the class XF_IDENTITY of make())
    bool containers_below_are_identity(int node, int depth = 0) {
        if (node < 0 || node >= d->n_nodes || depth > 300) return false;
        const int kind = d->node_kind[node];
        if (kind != RTX_NODE_GROUP && kind != RTX_NODE_CSG) return true;
        const int fc = d->node_first_child[node], nc = d->node_child_count[node];
        if (nc < 0 || fc < 0 || fc + nc > d->n_children) return false;
        for (int i = 0; i < nc; i++) {
            const int ch = d->children[fc + i];
            if (ch < 0 || ch >= d->n_nodes) return false;
            const int ck = d->node_kind[ch];
            if (ck == RTX_NODE_GROUP || ck == RTX_NODE_CSG) {
                const double* m = d->node_inverse + (size_t)ch * 16;
                for (int k = 0; k < 16; k++) if (!(m[k] == ((k % 5 == 0) ? 1.0 : 0.0))) return false;
                if (!containers_below_are_identity(ch, depth + 1)) return false;
            }
        }
        return true;
    }

    // returns the number of list entries the subtree can produce (for the CSG capacity check)
    int emit(int node, int level, bool list, int clevel) {
        if (err != RTX_OK) return 0;
        if (node < 0 || node >= d->n_nodes) { bad(RTX_E_INVALID, "node id out of range"); return 0; }
        if (seen[node]) { bad(RTX_E_INVALID, "node " + std::to_string(node) + " reachable twice (not a tree)"); return 0; }
        seen[node] = 1;
        const int kind = d->node_kind[node];
        const int lf = list ? OPF_LIST : 0;
        if (kind == RTX_NODE_SPHERE || kind == RTX_NODE_PLANE || kind == RTX_NODE_CUBE) {
            OpRec o = make(node, kind, level, lf, clevel);
            const int si = d->node_shape_index[node];
            if (si < 0) { bad(RTX_E_INVALID, "negative shape index"); return 0; }
            const int mi = d->node_material[node];
            if (mi < 0 || mi >= d->n_materials) { bad(RTX_E_INVALID, "material index out of range"); return 0; }
            o.shape = si;
            const int me = (int)ops.size();
            ops.push_back(o);
            if ((size_t)si >= shapes.size()) shapes.resize(si + 1, ShapeRec{-1, 0, 0, 0});
            if (shapes[si].type != -1) { bad(RTX_E_INVALID, "duplicate shape index"); return 0; }
            ShapeRec s; s.type = kind; s.material = mi; s.path_begin = (int)paths.size();
            for (int a : stack) paths.push_back(a);          // ancestors only, root first
            s.path_len = (int)stack.size();
            shapes[si] = s;
            (void)me;
            return kind == RTX_NODE_PLANE ? 1 : 2;
        }
        if (kind != RTX_NODE_GROUP && kind != RTX_NODE_CSG) { bad(RTX_E_INVALID, "unknown node kind"); return 0; }
        if (level >= RTX_MAX_LEVELS) { bad(RTX_E_CAPACITY, "container nesting deeper than RTX_MAX_LEVELS"); return 0; }
        const int fc = d->node_first_child[node], nc = d->node_child_count[node];
        if (nc < 0 || fc < 0 || fc + nc > d->n_children) { bad(RTX_E_INVALID, "children range out of bounds"); return 0; }
        if (level > max_container_level) max_container_level = level;
        int entries = 0;
        if (kind == RTX_NODE_GROUP) {
            if (list && clevel >= RTX_MAX_CSG_NEST) general = true;
            if (list && clevel > max_clevel) max_clevel = clevel;
            if (clevel > 255) { bad(RTX_E_CAPACITY, "containers nested deeper than 255 inside a CSG"); return 0; }
            const int b = (int)ops.size();
            ops.push_back(make(node, OPK_GROUP_BEGIN, level, lf, clevel));
            stack.push_back(b);
            const int xf = ops[b].xf;
            org_depth += xf != XF_IDENTITY; dir_depth += xf == XF_GENERAL;
            max_org = std::max(max_org, org_depth); max_dir = std::max(max_dir, dir_depth);
            for (int i = 0; i < nc; i++) entries += emit(d->children[fc + i], level + 1, list, clevel + 1);
            if (err == RTX_OK) geom_sphere(node, b);
            org_depth -= xf != XF_IDENTITY; dir_depth -= xf == XF_GENERAL;
            stack.pop_back();
            if (list) ops.push_back(make(-1, OPK_GROUP_END, level, lf, clevel));   // outside CSG an END has nothing to do
            ops[b].skip = (int)ops.size();
        } else {
            if (nc != 2) { bad(RTX_E_INVALID, "CSG node needs exactly 2 children"); return 0; }
            const int cop = d->node_csg_op[node];
            if (cop < 0 || cop > 2) { bad(RTX_E_INVALID, "unknown CSG operator"); return 0; }
            has_csg = true;
            const bool root = !list;
            const int cl = root ? 0 : clevel;
            if (cl >= RTX_MAX_CSG_NEST) general = true;
            if (cl > max_clevel) max_clevel = cl;
            if (cl > 255) { bad(RTX_E_CAPACITY, "containers nested deeper than 255 inside a CSG"); return 0; }
            const int f = OPF_LIST | (root ? OPF_ROOT : 0);
            const int b = (int)ops.size();
            OpRec ob = make(node, OPK_CSG_BEGIN, level, f, cl); ob.shape = cop;
            // a CSG root whose inner containers all have the identity transform (operands without a transform of their own: the
            // usual case) needs ONE local ray for its whole program: it stays in registers, no LDS slot (rtx_device.h: OPH_REGRAY)
            ob.regray = (root && ob.xf != XF_IDENTITY && err == RTX_OK && containers_below_are_identity(node)) ? 1 : 0;
            ops.push_back(ob);
            stack.push_back(b);
            const int xf = ob.regray ? XF_IDENTITY : ob.xf;      // (for the slot accounting only)
            org_depth += xf != XF_IDENTITY; dir_depth += xf == XF_GENERAL;
            max_org = std::max(max_org, org_depth); max_dir = std::max(max_dir, dir_depth);
            const int left_op = (int)ops.size();
            entries += emit(d->children[fc], level + 1, true, cl + 1);
            OpRec om = make(-1, OPK_CSG_MID, level, f, cl); om.shape = cop; ops.push_back(om);
            const int right_op = (int)ops.size();
            entries += emit(d->children[fc + 1], level + 1, true, cl + 1);
            if (err == RTX_OK) { geom_sphere(node, b); extent_sphere(node, b, left_op, right_op, cop); }
            org_depth -= xf != XF_IDENTITY; dir_depth -= xf == XF_GENERAL;
            stack.pop_back();
            OpRec oe = make(-1, OPK_CSG_END, level, f, cl); oe.shape = cop; ops.push_back(oe);
            ops[b].skip = (int)ops.size();
            if (root && entries > max_csg_entries) max_csg_entries = entries;
            if (root && entries > RTX_MAX_CSG_ENTRIES) general = true;
        }
        return entries;
    }
};

// Child lists for the list-by-list traversal outside CSG (rtx_device.h: ListChunk / ContRec)
struct ListBuilder {
    const std::vector<OpRec>& ops;
    std::vector<ListChunk> chunks;
    std::vector<ChunkCounts> counts;
    std::vector<float> cull;          // 4 floats per child slot, pair-interleaved per quad
    std::vector<float> quad;          // per chunk 8 spheres (one per quad of children), same layout: 32 floats
    std::vector<float> cull_t, quad_t;   // the same from the tighter CSG spheres (OpRec::cull_t)
    std::vector<int32_t> lops;
    std::vector<ContRec> conts;
    explicit ListBuilder(const std::vector<OpRec>& o) : ops(o) {}

    static void put_cull(std::vector<float>& cull, size_t slot, const float* c4) {
        const size_t q = slot / 4, p = slot % 4;
        if (cull.size() < (q + 1) * 16) cull.resize((q + 1) * 16, INFINITY);
        for (int k = 0; k < 4; k++) cull[q * 16 + (p / 2) * 8 + k * 2 + (p & 1)] = c4[k];
    }
    // children of the op range [begin, end); returns the index of the list's first chunk
    int build(int begin, int end) {
        std::vector<int> kids;
        for (int i = begin; i < end;) {
            kids.push_back(i);
            const int k = ops[i].kind;
            i = (k == OPK_GROUP_BEGIN || k == OPK_CSG_BEGIN) ? ops[i].skip : i + 1;
        }
        const int nchunks = kids.empty() ? 1 : (int)((kids.size() + 31) / 32);
        const int first = (int)chunks.size();
        chunks.resize(first + nchunks); counts.resize(first + nchunks);
        std::vector<std::pair<int, int>> groups;      // (cont_recs index, op) to recurse into afterwards
        for (int c = 0; c < nchunks; c++) {
            ListChunk ch; std::memset(&ch, 0, sizeof ch);
            ChunkCounts cn; std::memset(&cn, 0, sizeof cn);
            const int k0 = c * 32, n = std::min<int>(32, (int)kids.size() - k0);
            ch.n = n < 0 ? 0 : n;
            while (lops.size() % 4) lops.push_back(-1);
            ch.child_base = (int)lops.size();
            ch.next = c + 1 < nchunks ? first + c + 1 : -1;
            ch.cont_base = (int)conts.size();
            bool consec = ch.n > 0;
            for (int k = 0; k < ch.n; k++) {
                const int op = kids[k0 + k];
                const OpRec& o = ops[op];
                if (op != kids[k0] + k) consec = false;
                lops.push_back(op);
                put_cull(cull, (size_t)ch.child_base + k, o.cull);
                put_cull(cull_t, (size_t)ch.child_base + k, o.cull_t);
                if (o.kind == OPK_GROUP_BEGIN || o.kind == OPK_CSG_BEGIN) {
                    ch.cont_mask |= 1u << k; cn.containers++;
                    ContRec r; std::memset(&r, 0, sizeof r);
                    std::memcpy(r.inv, o.inv, sizeof r.inv); r.w33 = o.w33; std::memcpy(r.bounds, o.bounds, sizeof r.bounds);
                    r.t[0] = o.inv[3]; r.t[1] = o.inv[7]; r.t[2] = o.inv[11];
                    std::memcpy(r.cull, o.cull_t, sizeof r.cull);
                    std::memcpy(r.cull_loose, o.cull, sizeof r.cull_loose);
                    r.op = op; r.skip = o.skip; r.list = -1;
                    bool ok = true;
                    for (int b = 0; b < 6; b++) ok = ok && (std::isinf(o.bounds[b]) || std::fabs(o.bounds[b]) <= 1e100);
                    r.flags = ((o.flags & OPF_DIV) ? CONT_DIV : 0) | (ok ? 0 : CONT_NOFILTER) | (o.kind == OPK_CSG_BEGIN ? CONT_CSG : 0) |
                              (o.xf << CONT_XF_SHIFT);
                    if (o.kind == OPK_GROUP_BEGIN) groups.push_back({(int)conts.size(), op});
                    conts.push_back(r);
                } else if (o.flags & OPF_INLINE) {
                    ch.plane_mask |= 1u << k; cn.planes++;
                } else {
                    ch.shape_mask |= 1u << k;
                    if (o.kind == OPK_SPHERE) cn.spheres++; else if (o.kind == OPK_PLANE) cn.planes++; else cn.cubes++;
                }
            }
            ch.op0 = consec ? kids[k0] : -1;
            // bounding spheres of the chunk's quads; the two-level sweep pays on long lists only (a short list's quads are
            // all touched by some lane of an incoherent wave anyway, and the extra test is then pure overhead)
            for (int pass = 0; pass < 2; pass++) {
            std::vector<float>& quad = pass == 0 ? this->quad : this->quad_t;
            if (quad.size() < (size_t)(first + c + 1) * 32) quad.resize((size_t)(first + c + 1) * 32, INFINITY);
            for (int q = 0; q < 8; q++) {
                float s4[4] = {0.f, 0.f, 0.f, INFINITY};
                double cx = 0, cy = 0, cz = 0; int m = 0; bool fin = true;
                for (int k = q * 4; k < q * 4 + 4 && k < ch.n; k++) {
                    const float* cu = pass == 0 ? ops[kids[k0 + k]].cull : ops[kids[k0 + k]].cull_t;
                    fin = fin && std::isfinite(cu[3]);
                    cx += cu[0]; cy += cu[1]; cz += cu[2]; m++;
                }
                if (m > 0 && fin) {
                    cx /= m; cy /= m; cz /= m;
                    double R = 0.0;
                    for (int k = q * 4; k < q * 4 + 4 && k < ch.n; k++) {
                        const float* cu = pass == 0 ? ops[kids[k0 + k]].cull : ops[kids[k0 + k]].cull_t;
                        const double dx = cu[0] - cx, dy = cu[1] - cy, dz = cu[2] - cz;
                        R = std::fmax(R, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)cu[3]));
                    }
                    // (the children's radii already carry their f32 margins; the quad's centre is rounded to f32 once more)
                    R = R * 1.001 + 1e-6 * std::sqrt(cx * cx + cy * cy + cz * cz) + 1e-30;
                    const double r2 = R * R;
                    if (std::isfinite(r2) && r2 < 1e30) { s4[0] = (float)cx; s4[1] = (float)cy; s4[2] = (float)cz; s4[3] = std::nextafterf((float)r2, INFINITY); }
                }
                for (int k = 0; k < 4; k++) quad[(size_t)(first + c) * 32 + (q / 2) * 8 + k * 2 + (q & 1)] = s4[k];
            }
            }
            if (nchunks > 1) ch.n |= (int32_t)CHUNK_TWO_LEVEL;
            chunks[first + c] = ch; counts[first + c] = cn;
        }
        for (auto& g : groups) {
            const int li = build(g.second + 1, ops[g.second].skip);     // (no END op is emitted outside CSG)
            conts[g.first].list = li;
            static_assert(sizeof(ListChunk) == sizeof conts[0].child_hdr, "ContRec carries one ListChunk");
            static_assert(sizeof(ContRec) == 256 && offsetof(ContRec, inv) == 128 && offsetof(ContRec, cull_loose) == 240, "the kernel reads ContRec by dword offsets");
            std::memcpy(conts[g.first].child_hdr, &chunks[li], sizeof(ListChunk));
        }
        return first;
    }
    void finish() {
        while (lops.size() % 4) lops.push_back(-1);
        cull.resize((lops.size() / 4 + 3) * 16, INFINITY);              // whole quads + padding for the look-ahead loads
        cull_t.resize(cull.size(), INFINITY);
        quad.resize(chunks.size() * 32, INFINITY); quad_t.resize(chunks.size() * 32, INFINITY);
        if (conts.empty()) { ContRec r; std::memset(&r, 0, sizeof r); conts.push_back(r); }
    }
};

// CSG programs (rtx_device.h): the ops [begin, end) of one CSG root as self-contained records the kernel copies into LDS
// and walks there.  Returns the program's offset in `prog` (dwords); *len_out = its length.
int build_csg_program(const std::vector<OpRec>& ops, int begin, int end, std::vector<uint32_t>& prog, int* len_out) {
    auto rec_size = [&](const OpRec& o) {
        if (o.kind <= OPK_CUBE) return CSGP_SHAPE_DWORDS;
        if (o.kind == OPK_GROUP_BEGIN || o.kind == OPK_CSG_BEGIN)
            return o.xf == XF_GENERAL ? CSGP_BEGIN_GENERAL_DWORDS : o.xf == XF_TRANSLATE ? CSGP_BEGIN_TRANSLATE_DWORDS : CSGP_BEGIN_IDENTITY_DWORDS;
        return CSGP_HEAD_DWORDS;
    };
    std::vector<int> off(end - begin + 1, 0);
    for (int i = begin; i < end; i++) off[i - begin + 1] = off[i - begin] + rec_size(ops[i]);
    const int base = (int)prog.size();
    prog.resize(base + off[end - begin], 0u);
    auto put_d = [&](uint32_t* at, const double* d, int n) { std::memcpy(at, d, (size_t)n * sizeof(double)); };
    for (int i = begin; i < end; i++) {
        const OpRec& o = ops[i];
        uint32_t* r = prog.data() + base + off[i - begin];
        const bool is_begin = o.kind == OPK_GROUP_BEGIN || o.kind == OPK_CSG_BEGIN;
        r[0] = (uint32_t)o.kind | ((uint32_t)o.flags << 4) | ((uint32_t)o.level << 8) | ((uint32_t)o.clevel << 16);
        if (is_begin) {
            bool ok = true;
            for (int k = 0; k < 6; k++) ok = ok && (std::isinf(o.bounds[k]) || std::fabs(o.bounds[k]) <= 1e100);   // NaN fails both
            if (!ok) r[0] |= 1u << 24;
            r[0] |= (uint32_t)o.xf << 25;
            if (o.regray) r[0] |= 1u << 27;
            r[1] = (uint32_t)off[o.skip - begin];
        }
        r[2] = o.kind <= OPK_CUBE ? (uint32_t)i : (uint32_t)o.shape;     // shapes: their op index; CSG ops: the operator
        r[3] = (uint32_t)rec_size(o);
        std::memcpy(r + 4, o.cull, sizeof o.cull);
        if (o.kind <= OPK_CUBE) { put_d(r + 8, o.inv, 12); put_d(r + 32, &o.w33, 1); }
        else if (is_begin) {
            put_d(r + 8, o.bounds, 6);
            if (o.xf == XF_GENERAL) { put_d(r + 20, o.inv, 12); put_d(r + 44, &o.w33, 1); }
            else if (o.xf == XF_TRANSLATE) { const double t[3] = {o.inv[3], o.inv[7], o.inv[11]}; put_d(r + 20, t, 3); }
        }
    }
    *len_out = off[end - begin];
    return base;
}

template <typename T>
int upload(rtx_ctx* ctx, const std::vector<T>& v, void** out) {
    *out = nullptr;
    size_t bytes = v.size() * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIPCHK(ctx, hipMalloc(out, bytes));
    if (!v.empty()) HIPCHK(ctx, hipMemcpy(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return RTX_OK;
}

int local_bands(int vsize, rtx_bands b) {
    const int total = (vsize + RTX_TILE - 1) / RTX_TILE;
    if (b.first >= total) return 0;
    return (total - b.first + b.stride - 1) / b.stride;
}

}  // namespace

extern "C" {

const char* rtx_strerror(int code) {
    switch (code) {
        case RTX_OK: return "ok";
        case RTX_E_INVALID: return "invalid argument or malformed scene";
        case RTX_E_DEVICE: return "HIP device error";
        case RTX_E_CAPACITY: return "scene exceeds a bounded-list limit of the kernel";
        case RTX_E_UNSUPPORTED: return "unsupported scene feature";
        default: return "unknown error";
    }
}

const char* rtx_last_error(const rtx_ctx* c) { return c ? c->err.c_str() : ""; }

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

static void release_scene_memory(rtx_scene* s) {
    (void)hipSetDevice(s->device);
    for (void** p : {&s->d_ops, &s->d_bodies, &s->d_shapes, &s->d_paths, &s->d_materials, &s->d_lights, &s->d_pairs,
                     &s->d_chunks, &s->d_counts, &s->d_lcull, &s->d_lops, &s->d_conts, &s->d_quad, &s->d_lcull_t, &s->d_quad_t,
                     &s->d_arena, &s->d_arena_bm}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (rtx_cost_entry& e : s->cost) { if (e.d_cost) (void)hipFree(e.d_cost - RTX_COST_HEADER); e = rtx_cost_entry(); }
}

int rtx_init(int device, rtx_ctx** out) {
    if (!out) return RTX_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RTX_E_DEVICE;
    rtx_ctx* c = new rtx_ctx();
    c->device = device;
    // the scheduling knobs are read here, once; rtx_set_option() changes them afterwards
    c->opt.lpt = env_int("RTX_LPT", -1);
    c->opt.tile_mul = env_int("RTX_TILE_MUL", 7919);
    c->opt.probe_depth = env_int("RTX_PROBE_DEPTH", 2);
    c->opt.split = env_int("RTX_SPLIT", -1);
    c->opt.split_class = env_int("RTX_SPLIT_CLASS", 7);
    c->opt.debug = env_int("RTX_DEBUG", 0);
    c->opt.pipeline = env_int("RTX_PIPELINE", 3);
    c->opt.host_sets = env_int("RTX_HOST_SETS", 0);
    c->opt.stream_lpt = env_int("RTX_STREAM_LPT", -1);
    c->opt.probe_ahead = env_int("RTX_PROBE_AHEAD", -1);
    c->opt.hctx = env_int("RTX_HCTX", -1);
    c->opt.cost_feedback = env_int("RTX_COST_FEEDBACK", -1);
    c->opt.time_lpt = env_int("RTX_TIME_LPT", -1);
    rtx_ctx::Lane& l0 = c->lanes[0];
    void* sticky_dev = nullptr;
    void* scratch_dev = nullptr;
    const bool ok =
        hipSetDevice(device) == hipSuccess && hipStreamCreate(&c->stream) == hipSuccess &&
        hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
        hipMalloc((void**)&c->d_counters, RTX_CNT_SLOTS * RTX_CNT_STRIDE * sizeof(unsigned long long)) == hipSuccess &&
        hipMalloc((void**)&l0.d_tile_counts, 64 * sizeof(int32_t)) == hipSuccess &&
        // everything a queued render may need later is created here, so that rendering itself creates nothing
        hipStreamCreateWithFlags(&l0.side, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&l0.ev_fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&l0.ev_join, hipEventDisableTiming) == hipSuccess &&
        hipHostMalloc((void**)&c->h_counters, RTX_CNT_SLOTS * RTX_CNT_STRIDE * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
        hipHostMalloc((void**)&c->h_sticky, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer(&sticky_dev, (void*)c->h_sticky, 0) == hipSuccess &&
        // this device's copy of the no-alloc scratch array (each GPU has its own, at its own address)
        hipGetSymbolAddress(&scratch_dev, HIP_SYMBOL(scratch)) == hipSuccess && scratch_dev != nullptr;
    if (!ok) {
        rtx_shutdown(c);
        return RTX_E_DEVICE;
    }
    *c->h_sticky = 0;
    c->d_sticky = (unsigned long long*)sticky_dev;
    c->d_scratch = (unsigned char*)scratch_dev;
    l0.stream = c->stream;
    // the no-alloc lane: counts and lists live inside the scratch array; the side stream is lanes[0]'s
    c->noalloc_lane.d_tile_counts = (int32_t*)(c->d_scratch + RTX_NOALLOC_COUNTER_BYTES);
    c->noalloc_lane.d_lists = (int32_t*)(c->d_scratch + RTX_NOALLOC_COUNTER_BYTES + RTX_NOALLOC_COUNTS_BYTES);
    c->noalloc_lane.tile_cap = RTX_NOALLOC_MAX_TILES;
    c->noalloc_lane.side = l0.side; c->noalloc_lane.ev_fork = l0.ev_fork; c->noalloc_lane.ev_join = l0.ev_join;
    if (rtx_lane_events(c, &l0) != RTX_OK || rtx_lane_events(c, &c->noalloc_lane) != RTX_OK) {
        rtx_shutdown(c);
        return RTX_E_DEVICE;
    }
    *out = c;
    return RTX_OK;
}

void rtx_shutdown(rtx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();                 // renders may still be in flight on any of the streams used so far
    // scenes the caller has not destroyed yet: their device memory goes now, the handles stay valid for
    // rtx_scene_destroy (which then only frees the host struct)
    for (rtx_scene* s : c->scenes) { release_scene_memory(s); s->ctx = nullptr; }
    c->scenes.clear();
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    if (c->d_hit) (void)hipFree(c->d_hit);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_batch_counters) (void)hipFree(c->d_batch_counters);
    for (int l = 0; l < rtx_ctx::kLanes; l++) {
        if (c->lanes[l].d_lists) (void)hipFree(c->lanes[l].d_lists);
        if (c->lanes[l].d_tile_counts) (void)hipFree(c->lanes[l].d_tile_counts);
        c->lanes[l].release_ahead();
        if (l == 0) c->noalloc_lane.release_ahead();
        if (l > 0 && c->lanes[l].stream) (void)hipStreamDestroy(c->lanes[l].stream);
        if (c->lanes[l].side) (void)hipStreamDestroy(c->lanes[l].side);
        if (c->lanes[l].ev_fork) (void)hipEventDestroy(c->lanes[l].ev_fork);
        if (c->lanes[l].ev_join) (void)hipEventDestroy(c->lanes[l].ev_join);
    }
    delete c->copy_pool; c->copy_pool = nullptr;
    for (hipEvent_t e : c->set_events) if (e) (void)hipEventDestroy(e);
    c->set_events.clear();
    if (c->h_batch_counters) (void)hipHostFree(c->h_batch_counters);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_sticky) (void)hipHostFree((void*)c->h_sticky);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rtx_set_option(rtx_ctx* c, const char* name, int value) {
    if (!c || !name) return RTX_E_INVALID;
    const std::string n(name);
    if (n == "lpt") c->opt.lpt = value;
    else if (n == "tile_mul") c->opt.tile_mul = value;
    else if (n == "probe_depth") c->opt.probe_depth = value;
    else if (n == "split") c->opt.split = value;
    else if (n == "split_class") c->opt.split_class = value;
    else if (n == "debug") c->opt.debug = value;
    else if (n == "pipeline") c->opt.pipeline = value;
    else if (n == "host_sets") c->opt.host_sets = value;
    else if (n == "stream_lpt") c->opt.stream_lpt = value;
    else if (n == "probe_ahead") c->opt.probe_ahead = value;
    else if (n == "hctx") c->opt.hctx = value;
    else if (n == "cost_feedback") c->opt.cost_feedback = value;
    else if (n == "time_lpt") c->opt.time_lpt = value;
    else return fail(c, RTX_E_INVALID, "unknown option '" + n + "'");
    return RTX_OK;
}

int64_t rtx_debug_counter(const rtx_ctx* c, const char* name) {
    if (!c || !name) return -1;
    const std::string n(name);
    if (n == "probe_launches") return c->n_probe_launches;
    if (n == "classify_launches") return c->n_classify_launches;
    return -1;
}

int rtx_ctx_poll(rtx_ctx* c, int wait) {
    if (!c) return RTX_E_INVALID;
    if (wait) {
        // renders may be in flight on several streams (caller streams used in turn, the render lanes of a communicator, band
        // sets): wait for the device, not for the stream of the last call
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
    }
    const int rc = rtx_check_sticky(c);
    *c->h_sticky = 0;
    return rc;
}

int rtx_scene_create(rtx_ctx* ctx, const rtx_scene_desc* d, rtx_scene** out) {
    if (!ctx || !d || !out) return RTX_E_INVALID;
    *out = nullptr;
    if (d->n_nodes < 0 || d->n_roots < 0 || d->n_materials < 0 || d->n_lights < 0 || d->n_children < 0)
        return fail(ctx, RTX_E_INVALID, "negative count");
    if (d->n_nodes > 0 && (!d->node_kind || !d->node_inverse || !d->node_first_child || !d->node_child_count ||
                           !d->node_csg_op || !d->node_shape_index || !d->node_material || !d->node_bounds))
        return fail(ctx, RTX_E_INVALID, "NULL node array");
    if ((d->n_roots > 0 && !d->roots) || (d->n_children > 0 && !d->children) ||
        (d->n_materials > 0 && !d->materials) || (d->n_lights > 0 && (!d->light_position || !d->light_intensity)))
        return fail(ctx, RTX_E_INVALID, "NULL array");

    Compiler cc; cc.d = d; cc.seen.assign(d->n_nodes, 0);
    for (int i = 0; i < d->n_roots; i++) cc.emit(d->roots[i], 0, false, 0);
    if (cc.err != RTX_OK) return fail(ctx, cc.err, cc.msg);
    for (auto& s : cc.shapes) if (s.type == -1) { s.type = OPK_SPHERE; s.path_len = 0; }   // unreachable shapes

    std::vector<MaterialRec> mats(d->n_materials);
    bool transp = false, secondary = false;
    for (int i = 0; i < d->n_materials; i++) {
        const rtx_material& m = d->materials[i];
        MaterialRec& r = mats[i]; std::memset(&r, 0, sizeof r);
        r.ambient = m.ambient; r.diffuse = m.diffuse; r.specular = m.specular; r.shininess = m.shininess;
        r.reflective = m.reflective; r.transparency = m.transparency; r.refractive_index = m.refractive_index;
        for (int k = 0; k < 3; k++) { r.a[k] = m.color_a[k]; r.b[k] = m.color_b[k]; }
        if (m.pattern_kind < 0 || m.pattern_kind > RTX_PAT_TEST) return fail(ctx, RTX_E_INVALID, "unknown pattern kind");
        if (!affine_rows(m.pattern_inverse, r.pinv, &r.pw33)) return fail(ctx, RTX_E_UNSUPPORTED, "pattern transform is not affine");
        r.kind = m.pattern_kind; r.pdiv = !(r.pw33 == 1.0);
        if (!(m.transparency == 0.0)) transp = true;
        if (!(m.transparency == 0.0) && !(m.reflective == 0.0)) secondary = true;
    }
    std::vector<LightRec> lights(d->n_lights);
    for (int i = 0; i < d->n_lights; i++)
        for (int k = 0; k < 3; k++) { lights[i].pos[k] = d->light_position[i * 3 + k]; lights[i].intensity[k] = d->light_intensity[i * 3 + k]; }

    HIPCHK(ctx, hipSetDevice(ctx->device));
    rtx_scene* s = new rtx_scene(); s->ctx = ctx; s->device = ctx->device;
    ctx->scenes.push_back(s);
    int rc;
    std::vector<OpHead> heads(cc.ops.size() + 2);
    std::vector<OpBody> bodies(cc.ops.size() + 1);
    std::memset(heads.data(), 0, heads.size() * sizeof(OpHead));
    std::memset(bodies.data(), 0, bodies.size() * sizeof(OpBody));
    std::vector<HitRec> hits(cc.ops.size() + 1);
    std::memset(hits.data(), 0, hits.size() * sizeof(HitRec));
    for (auto& o : cc.ops)
        if (o.kind == OPK_PLANE && !(o.flags & (OPF_LIST | OPF_DIV))) o.flags |= OPF_INLINE;
    for (size_t i = 0; i < cc.ops.size(); i++) {
        const OpRec& o = cc.ops[i];
        OpHead& h = heads[i]; OpBody& b = bodies[i];
        h.info = (uint32_t)o.kind | ((uint32_t)o.flags << 4) | ((uint32_t)o.level << 8) | ((uint32_t)o.clevel << 16);
        if (o.kind == OPK_GROUP_BEGIN || o.kind == OPK_CSG_BEGIN) {
            bool ok = true;
            for (int k = 0; k < 6; k++) ok = ok && (std::isinf(o.bounds[k]) || std::fabs(o.bounds[k]) <= 1e100);   // NaN fails both
            if (!ok) h.info |= 1u << 24;
            h.info |= (uint32_t)o.xf << 25;
        }
        h.skip = o.skip; h.shape = o.shape;
        std::memcpy(h.cull, o.cull, sizeof h.cull);
        std::memcpy(b.inv, o.inv, sizeof b.inv); b.w33 = o.w33;
        b.div = (o.flags & OPF_DIV) ? 1 : 0; b.kind = o.kind;
        if (o.kind <= OPK_CUBE) {
            const ShapeRec& sr = cc.shapes[o.shape];
            HitRec& hr = hits[i];
            hr.shape = o.shape; hr.material = sr.material; hr.path_begin = sr.path_begin; hr.path_len = sr.path_len;
            hr.ri = d->materials[sr.material].refractive_index; hr.type = sr.type;
        }
    }
    // runs of consecutive shapes with the same level, list flag and inline flag, at most 60 long (one candidate bit each, plus up
    // to three for the quad partners of an unaligned first op)
    for (size_t i = cc.ops.size(); i-- > 0;) {
        const OpRec& o = cc.ops[i];
        if (o.kind > OPK_CUBE) continue;
        int run = 1;
        if (i + 1 < cc.ops.size()) {
            const OpRec& nx = cc.ops[i + 1];
            if (nx.kind <= OPK_CUBE && nx.level == o.level && ((nx.flags ^ o.flags) & (OPF_LIST | OPF_INLINE)) == 0 && heads[i + 1].run < 60)
                run = heads[i + 1].run + 1;
        }
        heads[i].run = run;
    }
    for (size_t i = cc.ops.size(); i < heads.size(); i++) { heads[i].info = OPK_GROUP_END; heads[i].cull[3] = INFINITY; }   // padding, never executed
    if (ctx->opt.debug == 1)
        for (size_t i = 0; i < heads.size(); i++)
            fprintf(stderr, "head %zu: info=%08x skip=%d shape=%d cull=%g %g %g %g\n", i, heads[i].info, heads[i].skip,
                    heads[i].shape, heads[i].cull[0], heads[i].cull[1], heads[i].cull[2], heads[i].cull[3]);
    ListBuilder lb(cc.ops);
    lb.build(0, (int)cc.ops.size());
    lb.finish();
    // one program per CSG root (a CSG that is a child of the world or of a group): the record points at it
    std::vector<uint32_t> prog;
    int prog_max = 0;
    for (ContRec& r : lb.conts)
        if (r.flags & CONT_CSG) {
            int len = 0;
            r.list = build_csg_program(cc.ops, r.op, r.skip, prog, &len);
            std::memset(r.t, 0, sizeof r.t);
            std::memcpy(r.t, &len, sizeof len);          // (a CSG root's `t` is unused by the list walk: it carries the length)
            prog_max = std::max(prog_max, len);
        }
    if (prog.empty()) prog.resize(8, 0u);
    if ((rc = upload(ctx, lb.chunks, &s->d_chunks)) || (rc = upload(ctx, lb.counts, &s->d_counts)) || (rc = upload(ctx, lb.cull, &s->d_lcull)) || (rc = upload(ctx, lb.quad, &s->d_quad)) ||
        (rc = upload(ctx, lb.cull_t, &s->d_lcull_t)) || (rc = upload(ctx, lb.quad_t, &s->d_quad_t)) ||
        (rc = upload(ctx, lb.lops, &s->d_lops)) || (rc = upload(ctx, lb.conts, &s->d_conts)) ||
        (rc = upload(ctx, prog, &s->d_pairs)) || (rc = upload(ctx, bodies, &s->d_bodies)) || (rc = upload(ctx, hits, &s->d_shapes)) ||
        (rc = upload(ctx, cc.paths, &s->d_paths)) || (rc = upload(ctx, mats, &s->d_materials)) ||
        (rc = upload(ctx, lights, &s->d_lights))) {
        rtx_scene_destroy(s);
        return rc;
    }
    s->dev.bodies = (const OpBody*)s->d_bodies; s->dev.csg_prog = (const uint32_t*)s->d_pairs; s->dev.csg_prog_max = prog_max; s->dev.hits = (const HitRec*)s->d_shapes;
    s->dev.chunks = (const ListChunk*)s->d_chunks; s->dev.chunk_counts = (const ChunkCounts*)s->d_counts; s->dev.list_cull = (const float*)s->d_lcull; s->dev.quad_cull = (const float*)s->d_quad;
    s->dev.list_cull_tight = (const float*)s->d_lcull_t; s->dev.quad_cull_tight = (const float*)s->d_quad_t;
    s->dev.list_ops = (const int32_t*)s->d_lops; s->dev.cont_recs = (const ContRec*)s->d_conts;
    s->dev.path_ops = (const int32_t*)s->d_paths; s->dev.materials = (const MaterialRec*)s->d_materials;
    s->dev.lights = (const LightRec*)s->d_lights;
    s->dev.n_ops = (int)cc.ops.size(); s->dev.n_shapes = (int)cc.shapes.size(); s->dev.n_lights = d->n_lights;
    s->dev.n_levels = cc.max_container_level + 2;
    s->dev.n_org_slots = cc.max_org + 1; s->dev.n_dir_slots = cc.max_dir + 1;   // slot 0 = the ray being traced
    s->dev.csg_entries = cc.max_csg_entries; s->dev.has_csg = cc.has_csg ? (cc.general ? 2 : 1) : 0; s->dev.has_transparency = transp ? 1 : 0; s->dev.has_branching = secondary ? 1 : 0;
    s->dev.csg_lds_entries = s->dev.has_csg == 1 ? cc.max_csg_entries : 0;
    s->dev.csg_prog_lds = s->dev.has_csg == 1 ? prog_max : 0;
    {
        // the wave's LDS (ray stack + bounded CSG lists + program + split mailboxes) must fit one workgroup's allocation
        const size_t lds = (size_t)(s->dev.n_org_slots + s->dev.n_dir_slots) * 192 * sizeof(double) + (((size_t)s->dev.csg_lds_entries * 64 * 12 + 15) & ~(size_t)15) +
                           (size_t)s->dev.csg_prog_lds * sizeof(uint32_t) + 64 * (3 * sizeof(double) + 2 * sizeof(int));
        if (lds > RTX_MAX_LDS_BYTES) {
            rtx_scene_destroy(s);
            return fail(ctx, RTX_E_CAPACITY, "nested transforms need " + std::to_string(lds) + " bytes of LDS ray stack per wave > " + std::to_string(RTX_MAX_LDS_BYTES));
        }
    }
    if (s->dev.has_csg == 2) {
        // GENERAL CSG: hit lists of any length and any nesting under a root (the reference's Vec, intersections.rs:41-59) live in
        // a slot of this arena, claimed by a wave for its lifetime.  More slots than waves can be resident on the chip.
        const size_t slot = (((size_t)cc.max_csg_entries * 64 * 12 + (size_t)(cc.max_clevel + 1) * 2 * 64 * 4) + 255) & ~(size_t)255;
        const int words = RTX_ARENA_SLOTS / 32;
        if (slot > 0x7fffffffull || slot * RTX_ARENA_SLOTS > ((size_t)48 << 30)) {
            rtx_scene_destroy(s);
            return fail(ctx, RTX_E_CAPACITY, "CSG root can produce " + std::to_string(cc.max_csg_entries) + " hits: its list arena would exceed 48 GiB");
        }
        if (hipMalloc(&s->d_arena, slot * RTX_ARENA_SLOTS) != hipSuccess || hipMalloc(&s->d_arena_bm, words * sizeof(unsigned)) != hipSuccess ||
            hipMemset(s->d_arena_bm, 0, words * sizeof(unsigned)) != hipSuccess) {
            rtx_scene_destroy(s);
            return fail(ctx, RTX_E_DEVICE, "cannot allocate the CSG list arena (" + std::to_string(slot * RTX_ARENA_SLOTS) + " bytes)");
        }
        s->dev.arena = (unsigned char*)s->d_arena; s->dev.arena_bitmap = (unsigned*)s->d_arena_bm;
        s->dev.arena_words = words; s->dev.arena_slot_bytes = (int32_t)slot;
    }
    *out = s;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* s) {
    if (!s) return;
    if (s->ctx) {
        // (renders queued asynchronously may still read the scene: wait for the last one before freeing)
        (void)hipSetDevice(s->device);
        (void)hipDeviceSynchronize();
        auto& v = s->ctx->scenes;
        v.erase(std::remove(v.begin(), v.end(), s), v.end());
        release_scene_memory(s);
    }
    delete s;
}

int rtx_debug_math(rtx_ctx* ctx, int op, int64_t n, const double* a, const double* b, double* out) {
    if (!ctx || !a || !b || !out || n <= 0 || op < 0 || op > 8) return RTX_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    double* d = nullptr;                         // one allocation: a | b | out
    const size_t bytes = (size_t)n * sizeof(double);
    HIPCHK(ctx, hipMalloc((void**)&d, 3 * bytes));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, (long long)n, d, d + n, d + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, RTX_E_DEVICE, std::string("rtx_debug_math: ") + hipGetErrorString(e));
    return RTX_OK;
}

int rtx_band_rows(int vsize, rtx_bands b) {
    if (vsize <= 0 || b.first < 0 || b.stride < 1) return 0;
    return local_bands(vsize, b) * RTX_TILE;
}

}  // extern "C"

namespace {

constexpr size_t kCounterWords = RTX_COUNTER_WORDS;

// Queues one render (counter reset, optional tile-order probe, frame launch) on `st`; nothing here waits.
int enqueue_render(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
                   uint32_t flags, void* d_rgb, void* d_hit, hipStream_t st, rtx_ctx::Lane* lane,
                   unsigned long long* d_counters, hipEvent_t ev_begin, hipEvent_t ev_end, bool may_allocate = true) {
    if (!ctx || !scene || !cam || !d_rgb) return RTX_E_INVALID;
    if (scene->ctx != ctx) return fail(ctx, RTX_E_INVALID, "scene belongs to another context");
    if (cam->hsize <= 0 || cam->vsize <= 0) return fail(ctx, RTX_E_INVALID, "empty camera");
    if (max_depth < 0 || max_depth > RTX_MAX_DEPTH) return fail(ctx, RTX_E_CAPACITY, "max_depth exceeds RTX_MAX_DEPTH");
    if (bands.first < 0 || bands.stride < 1) return fail(ctx, RTX_E_INVALID, "bad band selection");

    CameraDev cd; std::memset(&cd, 0, sizeof cd);
    cd.hsize = cam->hsize; cd.vsize = cam->vsize;
    cd.pixel_size = cam->pixel_size; cd.half_width = cam->half_width; cd.half_height = cam->half_height;
    if (!affine_rows(cam->inverse, cd.inv, &cd.w33)) return fail(ctx, RTX_E_UNSUPPORTED, "camera transform is not affine");
    cd.div = !(cd.w33 == 1.0);
    // origin = inverse * Point(0,0,0)  (camera.rs:66), same arithmetic as the per-pixel transform
    for (int r = 0; r < 3; r++) {
        const double* m = cd.inv + r * 4;
        double v = m[0] * 0.0 + m[1] * 0.0 + m[2] * 0.0 + m[3];
        cd.origin[r] = cd.div ? v / cd.w33 : v;
    }
    RenderParams rp; std::memset(&rp, 0, sizeof rp);
    rp.max_depth = max_depth; rp.band_first = bands.first; rp.band_stride = bands.stride;
    rp.n_bands_local = local_bands(cam->vsize, bands);
    rp.tiles_x = (cam->hsize + RTX_TILE - 1) / RTX_TILE;
    {
        // a multiplier coprime with the tile count (default 7919; RTX_TILE_MUL overrides for experiments)
        const long long n_tiles = (long long)rp.tiles_x * rp.n_bands_local;
        long long m = ctx->opt.tile_mul;
        auto gcd = [](long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; };
        if (m < 1) m = 1;
        while (n_tiles > 1 && gcd(m % n_tiles ? m % n_tiles : n_tiles, n_tiles) != 1) m++;
        rp.tile_mul = (int)(n_tiles > 0 ? m % (n_tiles > 1 ? n_tiles : 2) : 1);
        if (rp.tile_mul == 0) rp.tile_mul = 1;
    }
    rp.sticky = ctx->d_sticky;
    rp.hctx = ctx->opt.hctx;
    rp.compact = (flags & RTX_F_BANDS_COMPACT) ? 1 : 0;
    rp.no_early_out = (flags & RTX_F_NO_SHADOW_EARLY_OUT) ? 1 : 0;

    int ahead_set = -1;                           // list set this frame's launches read: its "launch done" event is recorded at the end
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ev_begin) HIPCHK(ctx, hipEventRecord(ev_begin, st));
    // ---- expensive tiles first: a probe launch (the centre pixel of every tile, depth budget 2, no image, no
    // counters) files every tile under a cost class; the frame launch walks the classes most expensive first.
    // A wave runs until its slowest pixel is done (up to ~126 rays for a glass pixel at depth 5); in an arbitrary
    // order such waves start late and drain at low occupancy (with exact costs the order was worth 16 % on the
    // glass/mirror workload).  Only used when some material branches (reflective AND transparent: per-pixel cost
    // then varies by two orders of magnitude) and the dispatch is long; RTX_LPT=0/1 overrides.
    {
        const int n_tiles = rp.tiles_x * rp.n_bands_local;
        // UNIFORM scenes (no material branches: a pixel traces 1-3 rays) have no use for a ray-count probe, but their tiles differ
        // in how much TRAVERSAL a ray costs (a tile over a dense cluster takes several times the median) and the launch ends in a
        // thinning tail of such waves.  With the tile-cost feedback they get a tile order too, from the second frame of a view on:
        // every wave leaves its DURATION as the tile's cost (time mode), the next frame files the tiles by cost / mean.  Measured
        // (tools/ms.py, same box): S3 4K 2.05 -> 1.98 ms, S5 8K 10.44 -> 9.97 ms; the three small launches it costs (zero, sum,
        // classify: ~15 us) outweigh the gain on short dispatches (S2 1080p 0.36 -> 0.38 ms), hence the tile threshold.
        const bool time_lpt = !(flags & RTX_F_INTERNAL_NO_LPT) && ctx->opt.lpt < 0 && ctx->opt.cost_feedback != 0 && ctx->opt.time_lpt != 0 && may_allocate &&
                              !scene->dev.has_branching && n_tiles >= (ctx->opt.time_lpt > 0 ? ctx->opt.time_lpt : RTX_TIME_LPT_MIN_TILES);
        const bool want = time_lpt || (!(flags & RTX_F_INTERNAL_NO_LPT) && (ctx->opt.lpt >= 0 ? ctx->opt.lpt != 0 : (max_depth >= 1 && n_tiles >= 2048 && scene->dev.has_branching)));
        if (want && n_tiles > 0) {
            if (lane->tile_cap < (size_t)n_tiles) {
                if (!may_allocate) return fail(ctx, RTX_E_CAPACITY, "frame has more tiles than the no-alloc scratch holds");
                if (lane->d_lists) (void)hipFree(lane->d_lists);
                lane->d_lists = nullptr; lane->tile_cap = 0;
                HIPCHK(ctx, hipMalloc((void**)&lane->d_lists, (size_t)RTX_TILE_CLASSES * n_tiles * sizeof(int32_t)));
                lane->tile_cap = n_tiles;
            }
#ifdef RTX_EXP_ORDER
            // experiment build only (tools/order_probe.py): the tile lists come from a file instead of the probe launch
            static bool exp_loaded = false;
            const char* exp_file = std::getenv("RTX_EXP_ORDER_FILE");
            if (exp_file && !exp_loaded) {
                std::vector<int32_t> data((size_t)RTX_TILE_CLASSES + (size_t)RTX_TILE_CLASSES * n_tiles, 0);
                FILE* f = std::fopen(exp_file, "rb");
                if (!f || std::fread(data.data(), sizeof(int32_t), data.size(), f) != data.size()) return fail(ctx, RTX_E_INVALID, "RTX_EXP_ORDER_FILE unreadable");
                std::fclose(f);
                HIPCHK(ctx, hipStreamSynchronize(st));
                HIPCHK(ctx, hipMemcpy(lane->d_tile_counts, data.data(), RTX_TILE_CLASSES * sizeof(int32_t), hipMemcpyHostToDevice));
                HIPCHK(ctx, hipMemcpy(lane->d_lists, data.data() + RTX_TILE_CLASSES, (size_t)RTX_TILE_CLASSES * n_tiles * sizeof(int32_t), hipMemcpyHostToDevice));
                exp_loaded = true;
            }
            if (exp_file) { rp.use_lists = 1; rp.tile_lists = lane->d_lists; rp.tile_counts = lane->d_tile_counts; }
            else {
#endif
            // short dispatches are bounded by their longest wave (a glass pixel traces up to ~110 rays one after the
            // other), not by throughput: there the expensive tiles are rendered by 4 waves each, 4 lanes per pixel
            // sharing the pixel's ray tree (SPLIT kernels, rtx_kernel.hip).  RTX_SPLIT=0/1 and RTX_SPLIT_CLASS override.
            const bool split = !time_lpt && !(flags & RTX_F_INTERNAL_NO_SPLIT) && (ctx->opt.split >= 0 ? ctx->opt.split != 0 : n_tiles <= RTX_SPLIT_MAX_TILES) && lane->side &&
                               scene->dev.has_csg != 2;      // (the general CSG kernels have no SPLIT variant)
            // PROBE AHEAD.  The probe is a chain of dependent rays on a few thousand waves: ~0.1 ms during which the chip is
            // nearly empty (2.5 % of a 4K frame).  When the caller's stream still holds work — frames are being queued one
            // behind the other — the probe goes to the lane's side stream and into the list set the frame before it is NOT
            // reading, so it runs beside that frame and this frame's launch finds its lists ready.  Stream order of everything
            // the caller can see (image, counters) is unchanged; if the two streams share a hardware queue the probe simply
            // runs where it used to.  A frame queued on an idle stream keeps the in-stream probe (no event hop in its latency).
            // (Callers that overlap whole frames themselves — the render lanes of rtx_render_gathered and rtx_render's band sets,
            // which pass RTX_F_INTERNAL_NO_SPLIT, and the lanes of rtx_render_batch — gain nothing from it: measured 0 to -4 %.)
            // TILE-COST FEEDBACK.  Every frame launch leaves per tile the traced rays of its busiest pixel (rp.tile_cost).  A frame of
            // the same VIEW of the scene (same image size, bands and depth budget, camera equal or close) files its tiles by those
            // measured costs with a classify launch (one thread per tile, microseconds) instead of tracing probe rays: the order is
            // the one exact costs give (2-3 % better than the probe's estimate) and the probe launch — a chain of
            // dependent rays on a nearly empty chip — is gone.  First frame of a view, a camera jump, or every 32nd frame while
            // tiles are split (a split tile cannot measure its own unsplit cost): the probe, which also seeds the costs.
            rtx_cost_entry* ce = nullptr;
            bool reuse = false;
            if (may_allocate && ctx->opt.cost_feedback != 0) {
                rtx_cost_entry* lru = &scene->cost[0];
                for (rtx_cost_entry& e : scene->cost) {
                    if (e.d_cost && e.hsize == cam->hsize && e.vsize == cam->vsize && e.band_first == bands.first && e.band_stride == bands.stride &&
                        e.max_depth == max_depth) { ce = &e; break; }
                    if (e.used < lru->used) lru = &e;
                }
                if (!ce) {
                    ce = lru;
                    if (ce->cap < (size_t)n_tiles) {
                        // The array carries its own capacity in a 16-byte header in FRONT of the costs (d_cost points behind it).  A
                        // wave finds the array through a word of the counter block it adds into; renders queued on DIFFERENT caller
                        // streams of one context share that block, so the tail of one frame can pick up the array of the frame
                        // queued after it, of another size: the wave checks its tile index against the array it really holds.
                        if (ce->d_cost) (void)hipFree(ce->d_cost - RTX_COST_HEADER);         // (hipFree waits for the device: no launch still uses it)
                        ce->d_cost = nullptr; ce->cap = 0;
                        int32_t* base = nullptr;
                        HIPCHK(ctx, hipMalloc((void**)&base, ((size_t)n_tiles + RTX_COST_HEADER + 4) * sizeof(int32_t)));      // (+ the time mode's 8-byte sum)
                        const int32_t header[RTX_COST_HEADER] = {n_tiles, 0, 0, 0};
                        if (hipMemcpy(base, header, sizeof header, hipMemcpyHostToDevice) != hipSuccess) {
                            (void)hipFree(base);
                            return fail(ctx, RTX_E_DEVICE, "tile-cost array: header upload failed");
                        }
                        ce->d_cost = base + RTX_COST_HEADER;
                        ce->cap = n_tiles;
                    }
                    ce->hsize = cam->hsize; ce->vsize = cam->vsize; ce->band_first = bands.first; ce->band_stride = bands.stride; ce->max_depth = max_depth;
                    ce->valid = false; ce->age = 0;
                }
                ce->used = ++scene->cost_clock;
                bool close = ce->valid;
                for (int k = 0; k < 16 && close; k++) close = std::fabs(ce->cam_inverse[k] - cam->inverse[k]) <= 0.05;    // (false on NaN)
                reuse = close && !(split && ce->age >= 32);
                std::memcpy(ce->cam_inverse, cam->inverse, sizeof ce->cam_inverse);
                ce->age = reuse ? ce->age + 1 : 0;
                ce->valid = true;
                rp.tile_cost = ce->d_cost;
            }
            if (time_lpt && !reuse) {
                // first frame of a uniform scene's view: nothing to order by yet — plain permuted order, the waves measure
                rp.tile_cost = (int32_t*)((uintptr_t)ce->d_cost | 1u);
            } else {
            bool ahead = !split && !reuse && !(flags & (RTX_F_INTERNAL_NO_SPLIT | RTX_F_INTERNAL_NO_AHEAD)) && may_allocate && lane->side && ctx->opt.probe_ahead != 0;
            if (ahead && ctx->opt.probe_ahead < 0) ahead = hipStreamQuery(st) == hipErrorNotReady;
            (void)hipGetLastError();
            int32_t* lists = lane->d_lists; int32_t* counts = lane->d_tile_counts;
            hipStream_t pst = st;
            int set = 0;
            if (!lane->ev_main[0] || !lane->ev_main[1] || !lane->ev_probe[0] || !lane->ev_probe[1])
                return fail(ctx, RTX_E_INVALID, "render lane without its list-set events (rtx_lane_events)");
            if (ahead) {
                set = lane->flip; lane->flip ^= 1;
                if (set == 1) {
                    if (lane->tile_cap_b < (size_t)n_tiles) {
                        if (lane->d_lists_b) (void)hipFree(lane->d_lists_b);
                        lane->d_lists_b = nullptr; lane->tile_cap_b = 0;
                        HIPCHK(ctx, hipMalloc((void**)&lane->d_lists_b, (size_t)RTX_TILE_CLASSES * n_tiles * sizeof(int32_t)));
                        lane->tile_cap_b = n_tiles;
                    }
                    lists = lane->d_lists_b; counts = lane->d_tile_counts + 32;
                }
                pst = lane->side;
            }
            // The probe REFILLS list set `set` (counts zeroed, lists rewritten): it must come after the last frame launch that
            // READ the set, whichever stream that launch was queued on and whichever kind of frame it belonged to (in-stream
            // probe, split, probe ahead).  Every such launch records ev_main[set] (below); a set nobody has read yet has none.
            if (lane->main_recorded[set]) HIPCHK(ctx, hipStreamWaitEvent(pst, lane->ev_main[set], 0));
            HIPCHK(ctx, hipMemsetAsync(counts, 0, RTX_TILE_CLASSES * sizeof(int32_t), pst));
            RenderParams pp = rp;
            {
                const long long n_super = (long long)((rp.tiles_x + 7) / 8) * ((rp.n_bands_local + 7) / 8);
                auto gcd = [](long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; };
                long long m = 211;
                while (n_super > 1 && gcd(m % n_super ? m % n_super : n_super, n_super) != 1) m++;
                pp.super_mul = (int)(n_super > 1 ? m % n_super : 1);
                if (pp.super_mul == 0) pp.super_mul = 1;
            }
            pp.probe = 1; pp.max_depth = ctx->opt.probe_depth;   // (1, 2, 3 measured: 2 is best)
            if (pp.max_depth < 0) pp.max_depth = 0;
            if (pp.max_depth > max_depth) pp.max_depth = max_depth; pp.tile_lists = lists; pp.tile_counts = counts;
            if (reuse) {
                unsigned long long* tsum = time_lpt ? (unsigned long long*)(ce->d_cost + ((ce->cap + 1) & ~(size_t)1)) : nullptr;
                HIPCHK(ctx, rtx_launch_classify_tiles(ce->d_cost, n_tiles, lists, counts, tsum, pst));
                if (time_lpt) rp.tile_cost = (int32_t*)((uintptr_t)ce->d_cost | 1u);
                ctx->n_classify_launches++;
            } else {
                HIPCHK(ctx, rtx_launch_trace_shade(&scene->dev, &cd, &pp, (uint8_t*)d_rgb, nullptr, d_counters, 0, pst));
                ctx->n_probe_launches++;
            }
            rp.use_lists = 1; rp.tile_lists = lists; rp.tile_counts = counts;
            if (ahead) {
                HIPCHK(ctx, hipEventRecord(lane->ev_probe[set], pst));
                HIPCHK(ctx, hipStreamWaitEvent(st, lane->ev_probe[set], 0));
            }
            ahead_set = set;
            if (split) {
                rp.split_cap = n_tiles / 4 > 0 ? n_tiles / 4 : 1;
                rp.split_min_class = ctx->opt.split_class;
                if (rp.split_min_class < 1) rp.split_min_class = 1;
                // a wave of the SPLIT kernel whose lanes wait for each other must still end; one pixel's own chain is at most
                // (2^(d+1) - 1) * (1 + n_lights) traces, the bound leaves 4x room on top of 16 pixels per wave
                const long long g = 4ll * 16 * (2ll << max_depth) * (1 + scene->dev.n_lights);
                rp.split_guard = (int32_t)(g > 0x3fffffff ? 0x3fffffff : g);
                if (ctx->opt.debug == 2) rp.split_guard = 1;        // test hook: every split wave gives up (sticky-error test)
            }
            }   // (!time_lpt || reuse)
#ifdef RTX_EXP_ORDER
            }
#endif
        }
    }
    // counters to zero (and the address of the tile-cost array into the block's spare word): one small launch on `st`
    HIPCHK(ctx, rtx_launch_reset_counters(d_counters, rp.tile_cost, st));
    if (rp.split_cap > 0) {
        // fork: the SPLIT launch (expensive tiles, several waves each) goes first, on the lane's side stream; the
        // main launch renders the other tiles beside it and waits for it at the end
        HIPCHK(ctx, hipEventRecord(lane->ev_fork, st));
        HIPCHK(ctx, hipStreamWaitEvent(lane->side, lane->ev_fork, 0));
        RenderParams sp = rp; sp.split_launch = 1;
        HIPCHK(ctx, rtx_launch_trace_shade(&scene->dev, &cd, &sp, (uint8_t*)d_rgb, (int32_t*)d_hit, d_counters,
                                           (flags & RTX_F_COUNT_EVENTS) ? 1 : 0, lane->side));
        HIPCHK(ctx, hipEventRecord(lane->ev_join, lane->side));
    }
    HIPCHK(ctx, rtx_launch_trace_shade(&scene->dev, &cd, &rp, (uint8_t*)d_rgb, (int32_t*)d_hit, d_counters,
                                       (flags & RTX_F_COUNT_EVENTS) ? 1 : 0, st));
    if (rp.split_cap > 0) HIPCHK(ctx, hipStreamWaitEvent(st, lane->ev_join, 0));
    if (ahead_set >= 0) { HIPCHK(ctx, hipEventRecord(lane->ev_main[ahead_set], st)); lane->main_recorded[ahead_set] = true; }
    if (ev_end) HIPCHK(ctx, hipEventRecord(ev_end, st));
    ctx->last_stream = st;
    return RTX_OK;
}

// Streams, side streams, events and tile counts of lanes[0 .. n): everything a queued render on that lane needs.  The streams
// of lanes 1.. are PICKED so that they really run beside the lanes before them (rtx_internal.h: hardware queues).
int ensure_lanes(rtx_ctx* ctx, int n) {
    if (n > rtx_ctx::kLanes) n = rtx_ctx::kLanes;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bool need = false;
    for (int l = 1; l < n; l++) need = need || !ctx->lanes[l].stream;
    if (need) {
        std::vector<hipStream_t> pool(2 * rtx_ctx::kLanes, nullptr), kept;
        for (hipStream_t& p : pool) HIPCHK(ctx, hipStreamCreateWithFlags(&p, hipStreamNonBlocking));
        for (int l = 0; l < n; l++) if (ctx->lanes[l].stream) kept.push_back(ctx->lanes[l].stream);
        bool ok = true;
        (void)rtx_spin_together({pool[0]}, &ok);                     // warm-up: code object load
        for (int l = 1; l < n; l++) {
            if (ctx->lanes[l].stream) continue;
            size_t pick = pool.size();
            for (size_t i = 0; i < pool.size() && pick == pool.size(); i++) {
                if (!pool[i]) continue;
                std::vector<hipStream_t> trial = kept; trial.push_back(pool[i]);
                if (rtx_streams_overlap(trial)) pick = i;
            }
            for (size_t i = 0; i < pool.size() && pick == pool.size(); i++) if (pool[i]) pick = i;
            ctx->lanes[l].stream = pool[pick]; kept.push_back(pool[pick]); pool[pick] = nullptr;
        }
        for (hipStream_t p : pool) if (p) (void)hipStreamDestroy(p);
    }
    for (int l = 0; l < n; l++) {
        rtx_ctx::Lane& ln = ctx->lanes[l];
        if (!ln.d_tile_counts) HIPCHK(ctx, hipMalloc((void**)&ln.d_tile_counts, 64 * sizeof(int32_t)));
        if (!ln.side) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ln.side, hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ln.ev_fork, hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
        }
        const int erc = rtx_lane_events(ctx, &ln);
        if (erc != RTX_OK) return erc;
    }
    return RTX_OK;
}

// Sums the replicated counter block `rep` (already on the host) into *stats.
int fill_stats(rtx_ctx* ctx, const rtx_camera* cam, rtx_bands bands, const unsigned long long* rep, float ms, rtx_stats* stats) {
    unsigned long long h[RTX_CNT_TOTAL] = {0};
    for (int sl = 0; sl < RTX_CNT_SLOTS; sl++)
        for (int k = 0; k < RTX_CNT_TOTAL; k++) h[k] += rep[(size_t)sl * RTX_CNT_STRIDE + k];
    std::memset(stats, 0, sizeof *stats);
    // valid pixels of the rendered bands = primary rays
    uint64_t prim = 0;
    const int nb = local_bands(cam->vsize, bands);
    for (int b = 0; b < nb; b++) {
        int y0 = (bands.first + b * bands.stride) * RTX_TILE;
        int rows = cam->vsize - y0; if (rows > RTX_TILE) rows = RTX_TILE;
        if (rows > 0) prim += (uint64_t)rows * cam->hsize;
    }
    stats->rays_primary = prim;
    stats->rays_secondary = h[RTX_CNT_RADIANCE] - prim;
    stats->rays_shadow = h[RTX_CNT_SHADOW];
    stats->overflow_pixels = h[RTX_CNT_OVERFLOW] % RTX_GUARD_UNIT;
    for (int i = 0; i < RTX_EV_COUNT; i++) stats->ev[i] = h[RTX_CNT_EV0 + i];
    stats->kernel_ms = ms;
    stats->rays_shadow_skipped = h[RTX_CNT_SKIPPED];
    if (h[RTX_CNT_OVERFLOW] >= RTX_GUARD_UNIT)
        return fail(ctx, RTX_E_CAPACITY, "a split-tile wave exceeded its iteration guard (pixels of that tile are unwritten)");
    if (stats->overflow_pixels) return fail(ctx, RTX_E_CAPACITY, "bounded CSG list overflowed on the device");
    return RTX_OK;
}

// rtx_render of a large frame: the frame is cut into C interleaved band sets (set i = bands i, i + C, ...: every set is a
// sample of the whole image, so the sets cost the same), rendered one after the other on two alternating lanes, each followed
// on its lane by the copy of its compact buffer into pinned staging; the host moves a set's bands into the caller's
// (pageable) image while the GPU renders the next sets.  One launch + one hipMemcpy left the GPU idle during the copy and the
// copy engine idle during the render: S5 8K 20.7 -> see DESIGN.md 7.
int render_chunked(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, uint32_t flags, int C,
                   uint8_t* rgb, int32_t* hit, rtx_stats* stats) {
    const size_t W = cam->hsize;
    const int nb_total = (cam->vsize + RTX_TILE - 1) / RTX_TILE;
    const size_t chunk_px = (size_t)((nb_total + C - 1) / C) * RTX_TILE * W;
    int rc = rtx_check_sticky(ctx);
    if (rc != RTX_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (C * chunk_px * 3 > ctx->rgb_cap) {
        if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
        ctx->d_rgb = nullptr; ctx->rgb_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_rgb, C * chunk_px * 3));
        ctx->rgb_cap = C * chunk_px * 3;
    }
    if (hit && C * chunk_px * 4 > ctx->hit_cap) {
        if (ctx->d_hit) (void)hipFree(ctx->d_hit);
        ctx->d_hit = nullptr; ctx->hit_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_hit, C * chunk_px * 4));
        ctx->hit_cap = C * chunk_px * 4;
    }
    const size_t stage_bytes = C * chunk_px * (hit ? 7 : 3);
    if (stage_bytes > ctx->stage_cap) {
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr; ctx->stage_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stage, stage_bytes, hipHostMallocDefault));
        ctx->stage_cap = stage_bytes;
    }
    if ((size_t)C > ctx->batch_cap) {
        if (ctx->d_batch_counters) (void)hipFree(ctx->d_batch_counters);
        ctx->d_batch_counters = nullptr; ctx->batch_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_batch_counters, (size_t)C * kCounterWords * sizeof(unsigned long long)));
        ctx->batch_cap = C;
    }
    rc = ensure_lanes(ctx, 2);
    if (rc != RTX_OK) return rc;
    const size_t set_tiles = (size_t)((W + RTX_TILE - 1) / RTX_TILE) * ((nb_total + C - 1) / C);
    for (int l = 0; l < 2; l++) {
        rtx_ctx::Lane& ln = ctx->lanes[l];
        if (scene->dev.has_branching && ln.tile_cap < set_tiles) {           // grow now: hipFree would stall the queue later
            if (ln.d_lists) (void)hipFree(ln.d_lists);
            ln.d_lists = nullptr; ln.tile_cap = 0;
            HIPCHK(ctx, hipMalloc((void**)&ln.d_lists, (size_t)RTX_TILE_CLASSES * set_tiles * sizeof(int32_t)));
            ln.tile_cap = set_tiles;
        }
    }
    // per set: begin, end (timed), landed in staging.  The events and the pinned copy of the counter blocks live with the
    // context: creating 3 C events, destroying them and a synchronous counter copy were ~0.3 ms of every frame
    std::vector<hipEvent_t>& ev = ctx->set_events;
    while (ev.size() < (size_t)3 * C) {
        hipEvent_t e = nullptr;
        if ((ev.size() % 3 == 2 ? hipEventCreateWithFlags(&e, hipEventDisableTiming) : hipEventCreate(&e)) != hipSuccess)
            return fail(ctx, RTX_E_DEVICE, "hipEventCreate failed");
        ev.push_back(e);
    }
    auto drop_events = [&]() { };
    if ((size_t)C > ctx->h_batch_cap) {
        if (ctx->h_batch_counters) (void)hipHostFree(ctx->h_batch_counters);
        ctx->h_batch_counters = nullptr; ctx->h_batch_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_batch_counters, (size_t)C * kCounterWords * sizeof(unsigned long long), hipHostMallocDefault));
        ctx->h_batch_cap = C;
    }
    uint8_t* const stage_hit = ctx->h_stage + C * chunk_px * 3;
    if (!ctx->copy_pool) ctx->copy_pool = new rtx_copy_pool(env_int("RTX_COPY_THREADS", 4) - 1);     // (+ the caller's thread)
    std::vector<rtx_copy_pool::Job> copy_jobs;
    for (int i = 0; i < C && rc == RTX_OK; i++) {
        rtx_ctx::Lane& ln = ctx->lanes[i & 1];
        rc = enqueue_render(ctx, scene, cam, max_depth, rtx_bands{i, C}, flags | RTX_F_BANDS_COMPACT | RTX_F_INTERNAL_NO_SPLIT,
                            ctx->d_rgb + (size_t)i * chunk_px * 3, hit ? (void*)(ctx->d_hit + (size_t)i * chunk_px) : nullptr, ln.stream, &ln,
                            ctx->d_batch_counters + (size_t)i * kCounterWords, ev[3 * i], ev[3 * i + 1]);
        if (rc != RTX_OK) break;
        if (hipMemcpyAsync(ctx->h_stage + (size_t)i * chunk_px * 3, ctx->d_rgb + (size_t)i * chunk_px * 3, chunk_px * 3, hipMemcpyDeviceToHost, ln.stream) != hipSuccess ||
            (hit && hipMemcpyAsync(stage_hit + (size_t)i * chunk_px * 4, ctx->d_hit + (size_t)i * chunk_px, chunk_px * 4, hipMemcpyDeviceToHost, ln.stream) != hipSuccess) ||
            hipEventRecord(ev[3 * i + 2], ln.stream) != hipSuccess ||
            hipMemcpyAsync(ctx->h_batch_counters + (size_t)i * kCounterWords, ctx->d_batch_counters + (size_t)i * kCounterWords,
                           kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ln.stream) != hipSuccess)
            rc = fail(ctx, RTX_E_DEVICE, "device to host copy failed");
    }
    // the host's share: a set's bands go to their rows as soon as the set has landed
    for (int i = 0; i < C && rc == RTX_OK; i++) {
        if (hipEventSynchronize(ev[3 * i + 2]) != hipSuccess) { rc = fail(ctx, RTX_E_DEVICE, "hipEventSynchronize failed"); break; }
        copy_jobs.clear();
        for (int k = 0; i + k * C < nb_total; k++) {
            const size_t y0 = (size_t)(i + k * C) * RTX_TILE;
            size_t rows = cam->vsize - y0; if (rows > RTX_TILE) rows = RTX_TILE;
            const size_t src_px = (size_t)i * chunk_px + (size_t)k * RTX_TILE * W;
            copy_jobs.push_back({ctx->h_stage + src_px * 3, rgb + y0 * W * 3, rows * W * 3});
            if (hit) copy_jobs.push_back({stage_hit + src_px * 4, (uint8_t*)(hit + y0 * W), rows * W * 4});
        }
        ctx->copy_pool->run(copy_jobs);
    }
    // nothing of this call may still be running when it returns
    for (int l = 0; l < 2; l++)
        if (hipStreamSynchronize(ctx->lanes[l].stream) != hipSuccess && rc == RTX_OK) rc = fail(ctx, RTX_E_DEVICE, "hipStreamSynchronize failed");
    if (rc == RTX_OK) {
        const unsigned long long* rep = ctx->h_batch_counters;           // (landed: both lanes were waited for above)
        rtx_stats sum; std::memset(&sum, 0, sizeof sum);
        for (int i = 0; i < C && rc == RTX_OK; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[0], ev[3 * i + 1]) != hipSuccess) { rc = fail(ctx, RTX_E_DEVICE, "hipEventElapsedTime failed"); break; }
            rtx_stats one;
            rc = fill_stats(ctx, cam, rtx_bands{i, C}, rep + (size_t)i * kCounterWords, ms, &one);
            sum.rays_primary += one.rays_primary; sum.rays_shadow += one.rays_shadow; sum.rays_secondary += one.rays_secondary;
            for (int k = 0; k < RTX_EV_COUNT; k++) sum.ev[k] += one.ev[k];
            sum.overflow_pixels += one.overflow_pixels; sum.rays_shadow_skipped += one.rays_shadow_skipped;
            if (one.kernel_ms > sum.kernel_ms) sum.kernel_ms = one.kernel_ms;      // first launch's begin to the last launch's end
        }
        if (stats) *stats = sum;
    }
    drop_events();
    // this call reports its own overflow through the return code; a raised word with clean counters is an earlier render's
    const bool raised = *ctx->h_sticky != 0;
    *ctx->h_sticky = 0;
    if (rc == RTX_OK && raised) rc = rtx_earlier_overflow(ctx);
    return rc;
}

}  // namespace

// ---- which streams really run side by side (rtx_internal.h)
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();                 // 100 MHz constant clock: the loop ends whatever happens
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

double rtx_spin_together(const std::vector<hipStream_t>& ss, bool* ok) {
    for (hipStream_t s : ss) if (hipStreamSynchronize(s) != hipSuccess) *ok = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (hipStream_t s : ss) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 10000ull);
    for (hipStream_t s : ss) if (hipStreamSynchronize(s) != hipSuccess) *ok = false;
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

bool rtx_streams_overlap(const std::vector<hipStream_t>& ss) {
    bool ok = true;
    double best = 1e30;
    for (int k = 0; k < 2; k++) { const double t = rtx_spin_together(ss, &ok); if (t < best) best = t; }
    return ok && best < 160.0;
}

int rtx_lane_events(rtx_ctx* ctx, rtx_ctx::Lane* lane) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < 2; i++) {
        if (!lane->ev_probe[i]) HIPCHK(ctx, hipEventCreateWithFlags(&lane->ev_probe[i], hipEventDisableTiming));
        if (!lane->ev_main[i]) HIPCHK(ctx, hipEventCreateWithFlags(&lane->ev_main[i], hipEventDisableTiming));
    }
    return RTX_OK;
}

int rtx_earlier_overflow(rtx_ctx* ctx) {
    return rtx_fail(ctx, RTX_E_CAPACITY, "an earlier asynchronous render overflowed a bounded list or tripped the split guard "
                                         "(its image is incomplete); rtx_ctx_poll() clears the condition");
}

int rtx_check_sticky(rtx_ctx* ctx) {
    if (ctx->h_sticky && *ctx->h_sticky != 0) return rtx_earlier_overflow(ctx);
    return RTX_OK;
}

int rtx_enqueue_render(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
                       uint32_t flags, void* d_rgb, void* d_hit, hipStream_t st, rtx_ctx::Lane* lane,
                       unsigned long long* d_counters, hipEvent_t ev_begin, hipEvent_t ev_end, bool may_allocate) {
    return enqueue_render(ctx, scene, cam, max_depth, bands, flags, d_rgb, d_hit, st, lane, d_counters, ev_begin, ev_end, may_allocate);
}

extern "C" {

int rtx_render_device(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
                      uint32_t flags, void* d_rgb, void* d_hit, void* stream, rtx_stats* stats) {
    if (!ctx) return RTX_E_INVALID;
    int rc = rtx_check_sticky(ctx);               // an earlier asynchronous render overflowed and nobody collected it
    if (rc != RTX_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    rc = enqueue_render(ctx, scene, cam, max_depth, bands, flags, d_rgb, d_hit, st, &ctx->lanes[0], ctx->d_counters,
                        stats ? ctx->ev0 : nullptr, stats ? ctx->ev1 : nullptr);
    if (rc != RTX_OK || !stats) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, kCounterWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    // This call reports its OWN overflow through the return code (fill_stats reads this call's counters).  A sticky word that
    // is raised although this call's counters are clean belongs to an EARLIER asynchronous render that was still running when
    // the check at the top looked: it is reported now, never wiped.
    const bool raised = *ctx->h_sticky != 0;
    *ctx->h_sticky = 0;
    rc = fill_stats(ctx, cam, bands, ctx->h_counters, ms, stats);
    if (rc == RTX_OK && raised) rc = rtx_earlier_overflow(ctx);
    return rc;
}

int rtx_render_device_noalloc(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
                              uint32_t flags, void* d_rgb, void* d_hit, void* stream) {
    if (!ctx || !cam) return RTX_E_INVALID;
    int rc = rtx_check_sticky(ctx);
    if (rc != RTX_OK) return rc;
    if (cam->hsize <= 0 || cam->vsize <= 0) return fail(ctx, RTX_E_INVALID, "empty camera");
    if (bands.first < 0 || bands.stride < 1) return fail(ctx, RTX_E_INVALID, "bad band selection");
    const long long n_tiles = (long long)((cam->hsize + RTX_TILE - 1) / RTX_TILE) * local_bands(cam->vsize, bands);
    if (n_tiles > RTX_NOALLOC_MAX_TILES)
        return fail(ctx, RTX_E_CAPACITY, "rtx_render_device_noalloc: " + std::to_string(n_tiles) + " tiles in this launch > RTX_NOALLOC_MAX_TILES (" +
                                         std::to_string(RTX_NOALLOC_MAX_TILES) + "): render fewer bands per call or use rtx_render_device");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // counters, tile counts and tile lists all live in this device's copy of `scratch` (ctx->d_scratch, from
    // hipGetSymbolAddress at rtx_init); enqueue_render clears what must start at zero on `st` before the launches
    return enqueue_render(ctx, scene, cam, max_depth, bands, flags, d_rgb, d_hit, st, &ctx->noalloc_lane,
                          (unsigned long long*)ctx->d_scratch, nullptr, nullptr, /*may_allocate=*/false);
}

int rtx_render(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
               uint32_t flags, uint8_t* rgb, int32_t* hit, rtx_stats* stats) {
    if (!ctx || !scene || !cam || !rgb) return RTX_E_INVALID;
    if (cam->hsize <= 0 || cam->vsize <= 0) return fail(ctx, RTX_E_INVALID, "empty camera");
    if (bands.first < 0 || bands.stride < 1) return fail(ctx, RTX_E_INVALID, "bad band selection");
    const bool compact = (flags & RTX_F_BANDS_COMPACT) != 0;
    const int nb = local_bands(cam->vsize, bands);
    if (!compact && bands.first == 0 && bands.stride == 1) {
        // whole frames of many tiles: render in interleaved band sets and copy each out while the next ones render
        const long long n_tiles = (long long)((cam->hsize + RTX_TILE - 1) / RTX_TILE) * nb;
        // how many sets: more sets hide more of the copy behind the render (what is left over is the last set's), but each set
        // is a shorter dispatch with a tail and a probe of its own; scenes whose pixel costs span orders of magnitude
        // (has_branching: the tail-bound ones) want few.  Measured with tools/host_latency.py
        // The FIRST whole-frame render of a context takes the plain path below: the band sets need pinned staging for the frame
        // and two calibrated render lanes (S5 8K: +22 ms, S3 4K: +11 ms of set-up against 9 / 2 ms saved per frame), which a
        // process that renders one frame (the CLI) would pay for nothing; a second render says that more are coming.
        const bool cold = ctx->host_renders++ == 0;
        long long C = 1;
        if (ctx->opt.host_sets > 0) C = ctx->opt.host_sets;
        else if (n_tiles >= 8000 && !cold) {
            C = scene->dev.has_branching ? n_tiles / 32000 : n_tiles / 8000;
            const long long hi = scene->dev.has_branching ? 4 : 8;
            C = C < 2 ? 2 : (C > hi ? hi : C);
        }
        if (C > 8) C = 8;
        if (C > nb) C = nb;
        if (C >= 2) return render_chunked(ctx, scene, cam, max_depth, flags, (int)C, rgb, hit, stats);
    }
    const size_t rows = compact ? (size_t)nb * RTX_TILE : (size_t)cam->vsize;
    const size_t npx = rows * cam->hsize;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (npx * 3 > ctx->rgb_cap) {
        if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
        ctx->d_rgb = nullptr; ctx->rgb_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_rgb, npx * 3));
        ctx->rgb_cap = npx * 3;
    }
    if (hit && npx * 4 > ctx->hit_cap) {
        if (ctx->d_hit) (void)hipFree(ctx->d_hit);
        ctx->d_hit = nullptr; ctx->hit_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_hit, npx * 4));
        ctx->hit_cap = npx * 4;
    }
    rtx_stats local;
    int rc = rtx_render_device(ctx, scene, cam, max_depth, bands, flags, ctx->d_rgb, hit ? ctx->d_hit : nullptr,
                               nullptr, stats ? stats : &local);
    if (rc != RTX_OK) return rc;
    const size_t W = cam->hsize;
    if (compact || bands.stride == 1) {
        // contiguous block of rows
        size_t r0 = compact ? 0 : (size_t)bands.first * RTX_TILE;
        size_t r1 = compact ? (size_t)nb * RTX_TILE : (size_t)cam->vsize;
        if (compact) {
            // the last band may be partial: copy only rows that exist
            int last_y0 = (bands.first + (nb - 1) * bands.stride) * RTX_TILE;
            int last_rows = nb > 0 ? cam->vsize - last_y0 : 0; if (last_rows > RTX_TILE) last_rows = RTX_TILE;
            r1 = nb > 0 ? (size_t)(nb - 1) * RTX_TILE + last_rows : 0;
        }
        if (r1 > r0) {
            HIPCHK(ctx, hipMemcpy(rgb + r0 * W * 3, ctx->d_rgb + r0 * W * 3, (r1 - r0) * W * 3, hipMemcpyDeviceToHost));
            if (hit) HIPCHK(ctx, hipMemcpy(hit + r0 * W, ctx->d_hit + r0 * W, (r1 - r0) * W * 4, hipMemcpyDeviceToHost));
        }
    } else {
        for (int b = 0; b < nb; b++) {
            size_t y0 = (size_t)(bands.first + b * bands.stride) * RTX_TILE;
            size_t y1 = y0 + RTX_TILE; if (y1 > (size_t)cam->vsize) y1 = cam->vsize;
            HIPCHK(ctx, hipMemcpy(rgb + y0 * W * 3, ctx->d_rgb + y0 * W * 3, (y1 - y0) * W * 3, hipMemcpyDeviceToHost));
            if (hit) HIPCHK(ctx, hipMemcpy(hit + y0 * W, ctx->d_hit + y0 * W, (y1 - y0) * W * 4, hipMemcpyDeviceToHost));
        }
    }
    return RTX_OK;
}

int rtx_render_batch(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cams, int n_cams, int max_depth,
                     uint32_t flags, uint8_t* const* rgb, int32_t* const* hit, rtx_stats* stats) {
    if (!ctx || !scene || !cams || !rgb || n_cams < 0) return RTX_E_INVALID;
    if (flags & RTX_F_BANDS_COMPACT) return fail(ctx, RTX_E_INVALID, "a batch renders whole frames");
    if (n_cams == 0) return RTX_OK;
    // one device image per frame, carved out of the context's (grown on demand) output buffers
    std::vector<size_t> off_rgb(n_cams + 1, 0), off_hit(n_cams + 1, 0);
    for (int i = 0; i < n_cams; i++) {
        if (cams[i].hsize <= 0 || cams[i].vsize <= 0 || !rgb[i]) return fail(ctx, RTX_E_INVALID, "empty camera or NULL image in batch");
        const size_t rows = (size_t)rtx_band_rows(cams[i].vsize, rtx_bands{0, 1});      // padded to whole tiles
        const size_t npx = rows * cams[i].hsize;
        off_rgb[i + 1] = off_rgb[i] + ((npx * 3 + 255) & ~(size_t)255);
        off_hit[i + 1] = off_hit[i] + ((hit && hit[i]) ? ((npx * 4 + 255) & ~(size_t)255) : 0);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (off_rgb[n_cams] > ctx->rgb_cap) {
        if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
        ctx->d_rgb = nullptr; ctx->rgb_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_rgb, off_rgb[n_cams]));
        ctx->rgb_cap = off_rgb[n_cams];
    }
    if (off_hit[n_cams] > ctx->hit_cap) {
        if (ctx->d_hit) (void)hipFree(ctx->d_hit);
        ctx->d_hit = nullptr; ctx->hit_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_hit, off_hit[n_cams]));
        ctx->hit_cap = off_hit[n_cams];
    }
    if ((size_t)n_cams > ctx->batch_cap) {
        if (ctx->d_batch_counters) (void)hipFree(ctx->d_batch_counters);
        ctx->d_batch_counters = nullptr; ctx->batch_cap = 0;
        HIPCHK(ctx, hipMalloc((void**)&ctx->d_batch_counters, (size_t)n_cams * kCounterWords * sizeof(unsigned long long)));
        ctx->batch_cap = n_cams;
    }
    // frames go round-robin over kLanes streams, so small frames overlap and fill the chip together; each
    // frame's image is copied to pinned staging on its own stream (the copy overlaps the other lanes' kernels)
    const int n_lanes = n_cams < rtx_ctx::kLanes ? n_cams : rtx_ctx::kLanes;
    size_t max_tiles = 0;
    for (int i = 0; i < n_cams; i++) {
        const size_t t = (size_t)((cams[i].hsize + RTX_TILE - 1) / RTX_TILE) * ((cams[i].vsize + RTX_TILE - 1) / RTX_TILE);
        if (t > max_tiles) max_tiles = t;
    }
    { const int lrc = ensure_lanes(ctx, n_lanes); if (lrc != RTX_OK) return lrc; }
    for (int l = 0; l < n_lanes; l++) {
        rtx_ctx::Lane& ln = ctx->lanes[l];
        if (scene->dev.has_branching && ln.tile_cap < max_tiles) {           // grow now: hipFree would stall the queue later
            if (ln.d_lists) (void)hipFree(ln.d_lists);
            ln.d_lists = nullptr; ln.tile_cap = 0;
            HIPCHK(ctx, hipMalloc((void**)&ln.d_lists, (size_t)RTX_TILE_CLASSES * max_tiles * sizeof(int32_t)));
            ln.tile_cap = max_tiles;
        }
    }
    const size_t stage_bytes = off_rgb[n_cams] + off_hit[n_cams];
    if (stage_bytes > ctx->stage_cap) {
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr; ctx->stage_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stage, stage_bytes, hipHostMallocDefault));
        ctx->stage_cap = stage_bytes;
    }
    std::vector<hipEvent_t> ev(stats ? (size_t)n_cams * 2 : 0, nullptr);
    auto drop_events = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    for (hipEvent_t& e : ev) if (hipEventCreate(&e) != hipSuccess) { drop_events(); return fail(ctx, RTX_E_DEVICE, "hipEventCreate failed"); }
    int rc = RTX_OK;
    for (int i = 0; i < n_cams && rc == RTX_OK; i++) {
        rtx_ctx::Lane& ln = ctx->lanes[i % n_lanes];
        const size_t npx = (size_t)cams[i].hsize * cams[i].vsize;
        const bool want_hit = hit && hit[i];
        rc = enqueue_render(ctx, scene, &cams[i], max_depth, rtx_bands{0, 1}, flags | RTX_F_INTERNAL_NO_AHEAD, ctx->d_rgb + off_rgb[i],
                            want_hit ? (void*)((uint8_t*)ctx->d_hit + off_hit[i]) : nullptr, ln.stream, &ln,
                            ctx->d_batch_counters + (size_t)i * kCounterWords,
                            stats ? ev[2 * i] : nullptr, stats ? ev[2 * i + 1] : nullptr);
        if (rc != RTX_OK) break;
        if (hipMemcpyAsync(ctx->h_stage + off_rgb[i], ctx->d_rgb + off_rgb[i], npx * 3, hipMemcpyDeviceToHost, ln.stream) != hipSuccess ||
            (want_hit && hipMemcpyAsync(ctx->h_stage + off_rgb[n_cams] + off_hit[i], (uint8_t*)ctx->d_hit + off_hit[i], npx * 4,
                                        hipMemcpyDeviceToHost, ln.stream) != hipSuccess))
            rc = fail(ctx, RTX_E_DEVICE, "device to host copy failed");
    }
    // every lane is drained even after an error: nothing of this call may still be running when it returns
    for (int l = 0; l < n_lanes; l++)
        if (hipStreamSynchronize(ctx->lanes[l].stream) != hipSuccess && rc == RTX_OK) rc = fail(ctx, RTX_E_DEVICE, "hipStreamSynchronize failed");
    for (int i = 0; i < n_cams && rc == RTX_OK; i++) {
        const size_t npx = (size_t)cams[i].hsize * cams[i].vsize;
        std::memcpy(rgb[i], ctx->h_stage + off_rgb[i], npx * 3);
        if (hit && hit[i]) std::memcpy(hit[i], ctx->h_stage + off_rgb[n_cams] + off_hit[i], npx * 4);
    }
    if (rc == RTX_OK && stats) {
        std::vector<unsigned long long> rep((size_t)n_cams * kCounterWords);
        if (hipMemcpy(rep.data(), ctx->d_batch_counters, rep.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(ctx, RTX_E_DEVICE, "counter copy failed");
        for (int i = 0; i < n_cams && rc == RTX_OK; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) != hipSuccess) { rc = fail(ctx, RTX_E_DEVICE, "hipEventElapsedTime failed"); break; }
            rc = fill_stats(ctx, &cams[i], rtx_bands{0, 1}, rep.data() + (size_t)i * kCounterWords, ms, &stats[i]);
        }
    }
    drop_events();
    // (frames without stats: an overflow shows up in the sticky word the launches set; every lane is drained here)
    if (rc == RTX_OK) rc = rtx_check_sticky(ctx);
    *ctx->h_sticky = 0;
    return rc;
}

}  // extern "C"
