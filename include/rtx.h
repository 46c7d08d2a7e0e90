/* rtx.h — C ABI of the MI355X trace-and-shade library (librtx_hip.so).
 *
 * This is the drop-in boundary for the body of the reference's
 *   Camera::render(&World) -> Canvas            (/root/reference/src/camera.rs:39-54)
 * i.e. everything inside the pixel loop camera.rs:45-51 (ray_for_pixel :56-69,
 * World::color_at world.rs:100-323) plus the 8-bit quantisation the Canvas is
 * only ever consumed through (canvas.rs:45-53, color.rs:40-48).
 *
 * The reference has no FFI/plugin interface of its own; these
 * entry points are what a Rust `extern "C"` block inside the crate would bind
 * (INTEGRATION.md shows that binding).  The host flattens its World
 * (object_store.rs:185-192, group.rs:16-19, csg.rs:29-33) into the
 * structure-of-arrays `rtx_scene_desc`: inverse transforms, primitive records,
 * group/CSG bounds, CSG trees, pattern parameters.  Plain pointers and sizes
 * only; the caller owns every array it passes; the library copies what it
 * needs at rtx_scene_create().
 *
 * All matrices: 16 doubles, COLUMN-major (nalgebra storage: m[c*4+r]).
 * All arithmetic on the device is f64 with the reference's operation order.
 * Not thread-safe per context (the reference is single-threaded, main.rs:11-28).
 * Renders queued on DIFFERENT streams of one context may overlap on the GPU: their
 * images and hit ids are unaffected, but they share the context's counter block, so
 * the rtx_stats of a render are meaningful only while no other render of that context
 * is in flight.  Use one context per concurrent stream where the counts matter.
 */
#ifndef RTX_H
#define RTX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* error codes (the reference panics / unwrap()s instead: main.rs:21,27) */
enum {
    RTX_OK = 0,
    RTX_E_INVALID = -1,      /* NULL / out-of-range argument, malformed scene graph */
    RTX_E_DEVICE = -2,       /* HIP runtime error (rtx_last_error() has the text) */
    RTX_E_CAPACITY = -3,     /* scene exceeds a bounded-list / nesting limit of the kernel */
    RTX_E_UNSUPPORTED = -4   /* e.g. non-affine (projective) transform */
};

/* node kinds: ShapeTypeMarker (shapes/mod.rs:84-89) + Group (group.rs) + CSG (csg.rs) */
enum { RTX_NODE_SPHERE = 0, RTX_NODE_PLANE = 1, RTX_NODE_CUBE = 2, RTX_NODE_GROUP = 3, RTX_NODE_CSG = 4 };
/* CsgOperator (csg.rs:22-26) */
enum { RTX_CSG_UNION = 0, RTX_CSG_INTERSECTION = 1, RTX_CSG_DIFFERENCE = 2 };
/* PatternData (patterns.rs:66-73) */
enum { RTX_PAT_PLAIN = 0, RTX_PAT_STRIPES = 1, RTX_PAT_RINGS = 2, RTX_PAT_CHECKERS = 3,
       RTX_PAT_GRADIENT = 4, RTX_PAT_TEST = 5 };

/* limits of the bounded per-thread structures (validated by rtx_scene_create) */
#define RTX_MAX_DEPTH        8    /* reflection/refraction budget; the reference hard-codes 5 (camera.rs:42) */
#define RTX_MAX_LEVELS      62    /* container nesting depth (groups/CSGs); the nested NON-IDENTITY transforms along one path must
                                     also fit the wave's LDS ray stack (64 KiB: about 20 general transforms, 40 translations) */
/* CSG roots within the next two limits render with the fast kernels (hit lists and the root's program in LDS); a scene with a
 * root beyond either renders with the GENERAL CSG kernels: lists of any length and any nesting (what csg.rs:73-91 and
 * intersections.rs:41-59 build with a Vec), held in a device arena sized by the scene's largest root at rtx_scene_create. */
#define RTX_MAX_CSG_ENTRIES 16    /* hits collected under one CSG root = 2 x leaf shapes */
#define RTX_MAX_CSG_NEST     8    /* containers nested inside one CSG root */
#define RTX_TILE             8    /* 8x8 pixel tile = one wavefront; band = RTX_TILE rows */

/* Material (material.rs:11-14) = Shader (material.rs:30-38) + Pattern (patterns.rs:9-12) */
typedef struct rtx_material {
    double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
    double color_a[3], color_b[3];       /* plain: a; two-colour patterns: a, b */
    double pattern_inverse[16];          /* inverse(pattern.transform) (object_store.rs:297) */
    int32_t pattern_kind;                /* RTX_PAT_* */
    int32_t _pad;
} rtx_material;

/* Flattened World.  One entry per node (shape, group or CSG) in any order. */
typedef struct rtx_scene_desc {
    int32_t n_nodes;
    const int32_t* node_kind;            /* RTX_NODE_* */
    const double*  node_inverse;         /* n_nodes*16: inverse(transform) (object_store.rs:244-245) */
    const double*  node_bounds;          /* n_nodes*6 {min xyz, max xyz}: local bounds of groups/CSGs
                                            as computed by set_bounds_of (object_store.rs:360-389);
                                            NaN = empty axis (bounds.rs:23-27); ignored for shapes */
    const int32_t* node_first_child;     /* groups: range into children[] in insertion order     */
    const int32_t* node_child_count;     /*   (group.rs:17,70); CSG: exactly 2 = {left, right}   */
    const int32_t* node_csg_op;          /* CSG: RTX_CSG_*; else ignored */
    const int32_t* node_shape_index;     /* shapes: reference ShapeIndex (reported as hit id) */
    const int32_t* node_material;        /* shapes: index into materials[] */
    int32_t n_children;
    const int32_t* children;             /* node ids */
    int32_t n_roots;
    const int32_t* roots;                /* World.scene in order (world.rs:82,113) */
    int32_t n_materials;
    const rtx_material* materials;
    int32_t n_lights;
    const double* light_position;        /* n_lights*3 (light.rs:8-11) */
    const double* light_intensity;       /* n_lights*3 */
} rtx_scene_desc;

/* Camera (camera.rs:10-17) with the inverse precomputed (camera.rs:65-66) */
typedef struct rtx_camera {
    int32_t hsize, vsize;
    double pixel_size, half_width, half_height;
    double inverse[16];                  /* inverse(camera.transform) */
} rtx_camera;

/* event counters: same meaning as the oracle's (oracle/rt_oracle.h) */
enum {
    RTX_EV_XFORM = 0, RTX_EV_SPHERE, RTX_EV_PLANE, RTX_EV_CUBE, RTX_EV_BOUNDS, RTX_EV_CSG_ENTRIES,
    RTX_EV_HIT, RTX_EV_LEVELS, RTX_EV_LIGHT, RTX_EV_POW, RTX_EV_REFR, RTX_EV_SCHLICK,
    RTX_EV_LANE_SLOTS,   /* diagnostic: 64 x wave-level traces (rays / this = lane packing) */
    RTX_EV_OP_SLOTS,     /* diagnostic: 64 x traversal ops the waves executed */
    /* work the kernel actually executes: the conservative bounding-sphere pre-test removes most exact
       node tests; the reference-defined counts above are unaffected by it */
    RTX_EV_CULL,         /* pre-tests evaluated (14 f32 flops each) */
    RTX_EV_X_XFORM,      /* exact ray-into-node transforms executed */
    RTX_EV_X_SPHERE,     /* exact sphere tests executed */
    RTX_EV_X_BOX,        /* exact cube / bounds slab tests executed */
    RTX_EV_BACKFACE,     /* light evaluations with light_dot_normal < 0 (the shadow answer cannot matter) */
    RTX_EV_USED,
    RTX_EV_COUNT = 24
};

typedef struct rtx_stats {
    uint64_t rays_primary, rays_shadow, rays_secondary;   /* 1 per color_at (world.rs:100) / is_shadowed (world.rs:195) */
    uint64_t ev[RTX_EV_COUNT];           /* filled only with RTX_F_COUNT_EVENTS */
    uint64_t overflow_pixels;            /* pixels that hit a bounded-list limit (then RTX_E_CAPACITY) */
    double   kernel_ms;                  /* hipEvent time of the launches of this render (tile-order probe, when used, + frame) */
    uint64_t rays_shadow_skipped;        /* of rays_shadow: is_shadowed calls whose answer material.rs:67 ignores (the light is
                                            behind the surface) — counted as the reference-defined rays they are, not traced.
                                            rays traced = primary + secondary + shadow - this.  0 with RTX_F_NO_SHADOW_EARLY_OUT */
} rtx_stats;

/* render flags */
enum {
    RTX_F_COUNT_EVENTS = 1,              /* run the event-counting kernel variant */
    RTX_F_NO_SHADOW_EARLY_OUT = 2,       /* traverse the whole scene for shadow rays (as the reference does) */
    RTX_F_BANDS_COMPACT = 4              /* write this rank's bands back-to-back instead of at image rows */
};

/* Which rows to render: 8-row bands b = first, first+stride, ... (row-interleaved tile bands,
 * one residue class per GPU).  {0,1} = whole image. */
typedef struct rtx_bands { int32_t first, stride; } rtx_bands;

typedef struct rtx_ctx rtx_ctx;
typedef struct rtx_scene rtx_scene;

/* Binds a context to one HIP device (one process per GPU).  Replaces nothing in the
 * reference (it has no device); called once before the first Camera::render (main.rs:24). */
int  rtx_init(int device, rtx_ctx** out);
void rtx_shutdown(rtx_ctx*);

/* Uploads a flattened World: what `&World` is to Camera::render (camera.rs:39).
 * Compiles the graph into the kernel's traversal program and validates the bounded-list limits. */
int  rtx_scene_create(rtx_ctx*, const rtx_scene_desc*, rtx_scene** out);
void rtx_scene_destroy(rtx_scene*);

/* Camera::render (camera.rs:39-54) + Canvas::to_image (canvas.rs:45-53): renders the selected
 * bands and copies RGB8 (and optional primary-hit ShapeIndex per pixel, -1 = miss) to host
 * buffers of hsize*vsize*3 bytes / hsize*vsize int32 (rows outside the bands are untouched;
 * with RTX_F_BANDS_COMPACT the buffers hold rtx_band_rows() rows).  hit may be NULL.  A whole frame of many tiles
 * is rendered in interleaved band sets whose copies to the host overlap the render of the next sets (same result;
 * stats.kernel_ms then spans the first launch's start to the last launch's end; option "host_sets" / RTX_HOST_SETS). */
int  rtx_render(rtx_ctx*, const rtx_scene*, const rtx_camera*, int max_depth, rtx_bands bands,
                uint32_t flags, uint8_t* rgb, int32_t* hit, rtx_stats* stats);

/* Same, but the outputs are DEVICE pointers on the context's device and the launch is queued on
 * `stream` (a hipStream_t, NULL = the context's own stream) without a host copy; used when the
 * caller gathers bands itself.  Synchronises only if stats != NULL; without stats a device-side overflow is
 * reported through the sticky error of rtx_ctx_poll() below.  Everything the caller can observe (image, ids, counters)
 * is ordered on `stream`; a frame queued while `stream` is still busy may have its tile-order probe (which reads only the
 * scene and the camera) run on an internal stream beside the work before it (option "probe_ahead"). */
int  rtx_render_device(rtx_ctx*, const rtx_scene*, const rtx_camera*, int max_depth, rtx_bands bands,
                       uint32_t flags, void* d_rgb, void* d_hit, void* stream, rtx_stats* stats);

/* Asynchronous renders (rtx_render_device without stats, rtx_render_device_noalloc) cannot return a device-side
 * RTX_E_CAPACITY themselves.  The condition is STICKY instead: the launch raises a flag in pinned host memory, the next
 * rtx_render_device* call on the context refuses to queue more work and returns RTX_E_CAPACITY, and
 * rtx_ctx_poll(ctx, wait) returns it (after waiting for the last queued render if wait != 0) and clears it.
 * The reference panics instead (intersections.rs:31, main.rs:27); nothing is ever silently truncated. */
int  rtx_ctx_poll(rtx_ctx*, int wait);

/* Scheduling knobs of a context (they never change an image).  Defaults come from the environment
 * ONCE, at rtx_init (RTX_LPT, RTX_TILE_MUL, RTX_PROBE_DEPTH, RTX_SPLIT, RTX_SPLIT_CLASS, RTX_PROBE_AHEAD, RTX_STREAM_LPT,
 * RTX_HOST_SETS, RTX_PIPELINE, RTX_HCTX, RTX_COST_FEEDBACK, RTX_TIME_LPT); names here: "lpt", "tile_mul", "probe_depth", "split", "split_class",
 * "probe_ahead", "stream_lpt", "host_sets", "pipeline", "hctx", "cost_feedback", "time_lpt"; value -1 = the default for "lpt",
 * "split", "probe_ahead", "stream_lpt", "hctx", "cost_feedback" and "time_lpt".
 * No counterpart in the reference (it has no scheduler). */
int  rtx_set_option(rtx_ctx*, const char* name, int value);
/* Diagnostic, no counterpart in the reference: launches of the scheduler so far — "probe_launches" (tile-order probes) and
 * "classify_launches" (tile orders taken from the costs the previous frame of the same view measured; option "cost_feedback",
 * RTX_COST_FEEDBACK: 0 = always probe).  -1 for an unknown name. */
int64_t rtx_debug_counter(const rtx_ctx*, const char* name);

/* rtx_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  No counterpart in the reference.  Same kernels and same output as rtx_render_device, but
 * the ray counters and the tile-order lists live in a file-scope device array of the library (one copy per GPU,
 * RTX_NOALLOC_SCRATCH: 64 KiB counters + 8 x RTX_NOALLOC_MAX_TILES x 4 B lists, about 4.1 MiB, held until the process
 * ends) instead of hipMalloc'ed buffers, and the call allocates nothing (no hipMalloc / hipFree / stream or event
 * creation: those happen in rtx_init).  Limit: launches of more than RTX_NOALLOC_MAX_TILES 8x8 tiles (a 4K frame is
 * 129 600) are rejected with RTX_E_CAPACITY.  No stats; errors are sticky (rtx_ctx_poll).  All calls a process makes
 * on one GPU share that GPU's array, which keeps the last call's contents (the call clears what must start at zero):
 * calls that overlap on one GPU - two streams, two contexts - race on it; queue them on one stream. */
#define RTX_NOALLOC_MAX_TILES 131072
int  rtx_render_device_noalloc(rtx_ctx*, const rtx_scene*, const rtx_camera*, int max_depth, rtx_bands bands,
                               uint32_t flags, void* d_rgb, void* d_hit, void* stream);

/* SURVEY.md §8f N4 — several frames of ONE resident scene: all cameras of a scene file (yaml.rs:34 parses
 * every `add: camera`, main.rs:24 renders only cameras[0]) or an animation sweep.  Every frame's launches are
 * queued back to back on the context's stream, the host waits once, then copies each image out: no per-frame
 * synchronisation, allocation or upload.  Frames may differ in size.  rgb: n_cams host pointers
 * (hsize*vsize*3 bytes each); hit: NULL or n_cams pointers, each NULL or hsize*vsize int32; stats: NULL or
 * n_cams records (kernel_ms is each frame's own launches).  Each frame equals what rtx_render returns for
 * that camera.  Multi-GPU use is frame-parallel: rank r renders cameras r, r+N, ... (no collective). */
int  rtx_render_batch(rtx_ctx*, const rtx_scene*, const rtx_camera* cams, int n_cams, int max_depth,
                      uint32_t flags, uint8_t* const* rgb, int32_t* const* hit, rtx_stats* stats);

/* ---- One frame on N GPUs (BASELINE.json north_star): the image is cut into row-interleaved 8-row tile
 * bands, band b belongs to rank b % N, every rank renders its bands into a compact buffer, ONE RCCL all-gather over xGMI
 * (ncclAllGather of rows_padded * W * 3 bytes per rank, ncclUint8) gives every rank all bands, and a copy kernel
 * de-interleaves them into the row-major frame.  The reference has no counterpart (one thread, camera.rs:45-51); the call
 * site is still main.rs:24.  librccl.so.1 is opened on the first rtx_comm_* call (not a load-time dependency).
 *
 * One process per GPU: rank 0 calls rtx_comm_unique_id and hands the RTX_COMM_ID_BYTES to the other ranks over the
 * host's own channel; every rank calls rtx_comm_init_rank with its own context, then rtx_render_gathered per frame.
 * One process, N GPUs: N contexts (rtx_init(i)), rtx_comm_init_all, one scene per context, rtx_render_multi. */
typedef struct rtx_comm rtx_comm;
#define RTX_COMM_ID_BYTES 128
int  rtx_comm_unique_id(void* id /* RTX_COMM_ID_BYTES */);
int  rtx_comm_init_rank(rtx_ctx*, const void* id, int rank, int world_size, rtx_comm** out);
int  rtx_comm_init_all(rtx_ctx* const* ctxs, int n, rtx_comm** out /* n handles */);
/* No counterpart in the reference and no RCCL: a communicator whose `world_size` ranks are all THIS GPU.  Every
 * rtx_render_gathered renders the N band sets one after the other where the all-gather would have put them and runs the
 * same de-interleave.  For checking the band mapping of N > 1 on a single GPU (tests), not for speed. */
int  rtx_comm_init_loopback(rtx_ctx*, int world_size, rtx_comm** out);
void rtx_comm_destroy(rtx_comm*);
int  rtx_comm_rank(const rtx_comm*);
int  rtx_comm_size(const rtx_comm*);
/* Diagnostic: how many of the communicator's streams (render lanes + gather stream) were found to run side by side on this
 * machine's hardware queues when it was created (wanted: frames in flight + 1; 1 = everything serialises). */
int  rtx_comm_concurrency(const rtx_comm*);

/* COLLECTIVE: every rank of `comm` must make the same sequence of rtx_render_gathered calls.  Once a rank's buffers exist the
 * all-gather is queued whatever happens to that rank's own render (an earlier frame's overflow, a failed launch), and the
 * rank-local error is returned AFTER it, so the other ranks never wait for a rank that gave up; only invalid arguments (the
 * same on every rank) or an allocation failure return without the collective.
 * Camera::render (camera.rs:39-54) on all ranks of `comm`: queues this rank's bands, then the all-gather and the
 * de-interleave on the communicator's own stream.  Consecutive calls render on alternating internal render lanes (three
 * frames in flight by default; option "pipeline" / RTX_PIPELINE = 1..4, read when the communicator is created), each
 * behind the work `stream` (NULL = the context's) holds at the time of the call: the start of frame k+1 fills the wave
 * slots the tail of frame k leaves idle, and its render overlaps frame k's gather.  Later work on `stream` does NOT wait
 * for the frame: rtx_comm_wait does.  d_frame: hsize*vsize*3 device bytes, row-major, complete on EVERY rank (a buffer
 * per frame in flight); d_hit_frame: NULL or hsize*vsize int32.  Returns once everything is queued.
 * stats != NULL renders on `stream` itself, waits for this rank's render and returns ITS share's counts.  scene must
 * belong to comm's context. */
int  rtx_render_gathered(rtx_comm*, const rtx_scene*, const rtx_camera*, int max_depth, uint32_t flags,
                         void* d_frame, void* d_hit_frame, void* stream, rtx_stats* stats);
/* Makes `stream` wait (on the device) for the last rtx_render_gathered's frame; stream == NULL: the HOST waits, and the
 * sticky error of the context (rtx_ctx_poll) is returned. */
int  rtx_comm_wait(rtx_comm*, void* stream);

/* The same frame from ONE host thread that owns all N GPUs: comms[i] / scenes[i] belong to rank i (rtx_comm_init_all;
 * the same scene description uploaded to every context).  Renders, gathers (one grouped all-gather), assembles on rank
 * 0's GPU and copies to host buffers (rgb: hsize*vsize*3, hit: NULL or hsize*vsize int32).  stats: NULL or the whole
 * frame's counts (sum over ranks).  NOTE: with stats != NULL every rank's render is waited for before the next rank's is
 * queued (each rank's counters are read back synchronously), so the ranks render ONE AFTER THE OTHER and stats->kernel_ms is
 * the slowest rank's own time, not a concurrent measurement; pass NULL when timing the frame.  With n == 1 the same calls
 * run with a one-rank communicator. */
int  rtx_render_multi(rtx_comm* const* comms, const rtx_scene* const* scenes, int n, const rtx_camera*, int max_depth,
                      uint32_t flags, uint8_t* rgb, int32_t* hit, rtx_stats* stats);

/* Host-side de-interleave for callers that gather compact band buffers (RTX_F_BANDS_COMPACT) with a transport of their
 * own: gathered = n_ranks buffers of rtx_band_rows_padded() rows back to back, frame = height rows.  Row y of the image is
 * row (y/8/N)*8 + y%8 of rank (y/8) % N. */
int  rtx_band_rows_padded(int vsize, int n_ranks);
int  rtx_band_deinterleave(const uint8_t* gathered, int n_ranks, int width, int height, int bytes_per_pixel, uint8_t* frame);

/* rows (padded to whole bands) a rank holds under `bands`, i.e. the compact buffer height */
int  rtx_band_rows(int vsize, rtx_bands bands);

/* Diagnostic with no counterpart in the reference: evaluates one f64 library operation per element
 * on the device so tests can compare the GPU's sqrt / division / pow / floor / round bits with the
 * host's.  op: 0 sqrt(a), 1 a/b, 2 pow(a,b), 3 floor(a), 4 round(a), 5 fmax(a,b), 6 fmin(a,b).
 * a, b, out are HOST arrays of n doubles. */
int  rtx_debug_math(rtx_ctx*, int op, int64_t n, const double* a, const double* b, double* out);

const char* rtx_strerror(int code);
const char* rtx_last_error(const rtx_ctx*);

#ifdef __cplusplus
}
#endif
#endif
