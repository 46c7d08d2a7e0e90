// rtx_internal.h — host-side state behind the opaque handles of include/rtx.h, shared by
// rtx_api.hip (scene compiler, single-GPU entry points) and rtx_multi.hip (RCCL band gather).
#ifndef RTX_INTERNAL_H
#define RTX_INTERNAL_H
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_device.h"

// scheduling knobs: read from the environment ONCE, at rtx_init; rtx_set_option() changes them per context
struct rtx_opts {
    int lpt = -1;            // expensive-tiles-first order: -1 = per-scene default, 0 / 1 = forced
    int tile_mul = 7919;     // multiplier of the tile permutation
    int probe_depth = 2;     // depth budget of the tile-cost probe launch
    int split = -1;          // split tiles: -1 = default (short dispatches), 0 / 1 = forced
    int split_class = 7;     // tiles of this cost class and above are split
    int debug = 0;
    int probe_ahead = -1;    // the probe of a frame queued behind a running one runs beside it, on the lane's side stream: -1 = when the
                             // caller's stream is busy, 0 = never, 1 = always
    int stream_lpt = -1;     // streaming callers of rtx_render_gathered keep the tile-order probe: -1 = by dispatch length, 0 / 1 = forced
    int host_sets = 0;       // rtx_render (host buffers): band sets a whole frame is rendered and copied out in; 0 = by size, 1 = one launch
    int cost_feedback = -1;  // tile order from the costs the previous frame of the same view measured, instead of a probe launch:
                             // -1 / 1 = on, 0 = off (always probe)
    int time_lpt = -1;       // uniform scenes (no branching material): tile order from the wave durations the previous frame measured: -1 = launches of at least
                             // RTX_TIME_LPT_MIN_TILES tiles, 0 = off, n > 0 = launches of at least n tiles
    int hctx = -1;           // hit context in LDS instead of registers: -1 = where the wave's LDS budget allows (rtx_kernel.hip), 0 / 1 = forced
    int pipeline = 3;        // rtx_render_gathered: frames in flight, each on a render lane of its own (read at rtx_comm_init_*;
                             // 2..4; below 2 = every frame on the caller's stream)
};

constexpr size_t RTX_COUNTER_WORDS = (size_t)RTX_CNT_SLOTS * RTX_CNT_STRIDE;   // one replicated counter block

struct rtx_scene;
struct rtx_copy_pool;        // rtx_api.hip: a few host threads that move landed band sets from pinned staging into the caller's image

struct rtx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_counters = nullptr;
    // scheduling state of one stream: tile class lists (grown on demand) and their fill counts.  lanes[0] runs
    // on `stream`; lanes 1.. have streams of their own and are created by the first rtx_render_batch.
    // side/ev_fork/ev_join: the SPLIT launch of a frame runs on `side`, beside the main launch
    // probe ahead (rtx_api.hip, enqueue_render): the tile-order probe of a frame runs on `side` into one of TWO list sets
    // (d_lists / d_lists_b, counts at d_tile_counts + 0 / + 32), so it can run beside the frame before it; ev_probe[s] = the
    // probe that filled set s is done, ev_main[s] = the frame launch that read set s is done (the set may be refilled)
    struct Lane { hipStream_t stream = nullptr; int32_t* d_lists = nullptr; int32_t* d_tile_counts = nullptr; size_t tile_cap = 0;
                  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
                  int32_t* d_lists_b = nullptr; size_t tile_cap_b = 0; hipEvent_t ev_probe[2] = {nullptr, nullptr}, ev_main[2] = {nullptr, nullptr};
                  bool main_recorded[2] = {false, false}; int flip = 0;
                  void release_ahead() {
                      if (d_lists_b) (void)hipFree(d_lists_b);
                      for (int i = 0; i < 2; i++) { if (ev_probe[i]) (void)hipEventDestroy(ev_probe[i]); if (ev_main[i]) (void)hipEventDestroy(ev_main[i]); }
                      d_lists_b = nullptr; tile_cap_b = 0; ev_probe[0] = ev_probe[1] = ev_main[0] = ev_main[1] = nullptr;
                      main_recorded[0] = main_recorded[1] = false;
                  } };
    static constexpr int kLanes = 4;
    Lane lanes[kLanes];
    uint8_t* h_stage = nullptr; size_t stage_cap = 0;                        // pinned staging of a batch's images
    rtx_copy_pool* copy_pool = nullptr;                                      // created by the first band-set render (rtx_render)
    std::vector<hipEvent_t> set_events;                                      // band-set renders: 3 events per set, kept between calls
    unsigned long long* h_batch_counters = nullptr; size_t h_batch_cap = 0;  // pinned copy of the sets' counter blocks
    uint8_t* d_rgb = nullptr; size_t rgb_cap = 0;
    int32_t* d_hit = nullptr; size_t hit_cap = 0;
    unsigned long long* d_batch_counters = nullptr; size_t batch_cap = 0;   // one counter block per frame of a batch
    unsigned long long* h_counters = nullptr;                               // pinned copy of one counter block (stats path)
    // sticky error word of asynchronous renders: pinned host memory the device can write (rtx_kernel.hip stores the
    // overflow count's "non-zero" there); read by the next render call / rtx_ctx_poll without touching the stream
    volatile unsigned long long* h_sticky = nullptr;
    unsigned long long* d_sticky = nullptr;                                  // the same word as the device addresses it
    long long host_renders = 0;                                              // whole-frame rtx_render calls so far (the first one sets nothing up: rtx_api.hip)
    hipStream_t last_stream = nullptr;                                       // stream of the last queued render (rtx_ctx_poll waits on it)
    // the library's no-alloc scratch (rtx_api.hip: `scratch`): this device's copy of the file-scope array
    unsigned char* d_scratch = nullptr;
    Lane noalloc_lane;                                                       // lists / counts inside d_scratch, side stream shared with lanes[0]
    rtx_opts opt;
    long long n_probe_launches = 0, n_classify_launches = 0;                 // diagnostics (rtx_debug_counter)
    std::vector<rtx_scene*> scenes;                                          // live scenes (rtx_shutdown releases their device memory)
    std::string err;
};

// Tile-cost feedback (rtx_api.hip, enqueue_render): what the last frame of one VIEW of a scene (image size, band selection, depth
// budget, camera) measured per tile; the next frame of the same view orders its tiles by it instead of probing.
struct rtx_cost_entry {
    int hsize = 0, vsize = 0, band_first = 0, band_stride = 0, max_depth = 0;
    double cam_inverse[16] = {};
    int32_t* d_cost = nullptr; size_t cap = 0;   // costs[cap] (+ the time mode's 8-byte sum); the allocation starts RTX_COST_HEADER words earlier: [0] = cap,
                                                 // which the writing wave checks its tile index against (rtx_kernel.hip: tile_cost_tail)
    bool valid = false;          // d_cost holds the costs of a frame of this view (queued or done)
    int age = 0;                 // frames since the last probe launch of this view
    unsigned long long used = 0; // for eviction: context-wide use counter at the last use
};

struct rtx_scene {
    static constexpr int kCostEntries = 16;
    mutable rtx_cost_entry cost[kCostEntries];
    mutable unsigned long long cost_clock = 0;
    rtx_ctx* ctx = nullptr;      // nullptr once the context is gone (rtx_shutdown released the device memory)
    int device = 0;
    SceneDev dev{};
    void* d_ops = nullptr; void* d_bodies = nullptr; void* d_shapes = nullptr; void* d_paths = nullptr;
    void* d_materials = nullptr; void* d_lights = nullptr; void* d_pairs = nullptr;
    void* d_arena = nullptr; void* d_arena_bm = nullptr;     // general CSG mode: list arena + slot bitmap (rtx_device.h)
    void* d_chunks = nullptr; void* d_counts = nullptr; void* d_lcull = nullptr; void* d_lops = nullptr; void* d_conts = nullptr; void* d_quad = nullptr; void* d_lcull_t = nullptr; void* d_quad_t = nullptr;
};

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return RTX_E_DEVICE;                                                       \
        }                                                                              \
    } while (0)

static inline int rtx_fail(rtx_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

// rtx_api.hip: queues one render (counter reset, optional tile-order probe, frame launch) on `st`; nothing waits.
int rtx_enqueue_render(rtx_ctx* ctx, const rtx_scene* scene, const rtx_camera* cam, int max_depth, rtx_bands bands,
                       uint32_t flags, void* d_rgb, void* d_hit, hipStream_t st, rtx_ctx::Lane* lane,
                       unsigned long long* d_counters, hipEvent_t ev_begin, hipEvent_t ev_end, bool may_allocate);
// Which streams really run side by side: HIP maps streams onto a few hardware queues by a policy of its own, and launches on
// streams that share a queue run one after the other.  rtx_spin_together: wall time (us) of one 100 us spin kernel on each of
// the given (idle) streams, started together; rtx_streams_overlap: true if that is well under the serial time.
double rtx_spin_together(const std::vector<hipStream_t>& streams, bool* ok);
bool rtx_streams_overlap(const std::vector<hipStream_t>& streams);
// internal flag bit of enqueue_render's `flags`: never use the SPLIT launch (the caller overlaps launches itself)
constexpr uint32_t RTX_F_INTERNAL_NO_SPLIT = 1u << 16;
// ... and: no tile-order probe either (plain permuted order)
constexpr uint32_t RTX_F_INTERNAL_NO_LPT = 1u << 17;
// ... and: the probe stays on the render's own stream (callers that put whole frames side by side themselves)
constexpr uint32_t RTX_F_INTERNAL_NO_AHEAD = 1u << 18;
// RTX_E_CAPACITY (and the message) if an earlier asynchronous render reported an overflow that nobody collected yet
int rtx_check_sticky(rtx_ctx* ctx);
// The four list-set events of a lane (ev_probe / ev_main).  They are created WITH the lane (rtx_init, ensure_lanes, the render
// lanes of a communicator), never lazily by a render: every frame launch that reads a list set records ev_main[set], and every
// refill of a set waits for it, whichever kind of frame (in-stream probe, split, probe ahead) used the set last.
int rtx_lane_events(rtx_ctx* ctx, rtx_ctx::Lane* lane);

#endif
