// rtx_multi.hip — one frame on N GPUs: row-interleaved 8-row tile bands, one band set per GPU, gathered with an
// RCCL all-gather over xGMI and de-interleaved into the row-major frame (BASELINE.json north_star).
//
// The reference has no counterpart (single thread, camera.rs:45-51); the seam is still main.rs:24: ONE call renders
// cameras[0], on however many GPUs the host was given.  Two ways to own the GPUs:
//   * one process per GPU  — rtx_comm_unique_id on rank 0, the host hands the 128 bytes to the other ranks over its
//     own channel, every rank calls rtx_comm_init_rank and then rtx_render_gathered per frame;
//   * one process, N GPUs  — rtx_comm_init_all over N contexts, rtx_render_multi per frame (what `rtx --gpus N` does).
// RCCL (librccl.so.1, 570 MB) is opened on the first rtx_comm_* call, not at load time: single-GPU users never map it,
// and in a process that already carries an RCCL (PyTorch-ROCm) the same copy is used.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_internal.h"

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    if (r.so || !r.err.empty()) return &r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.so) break;
    }
    if (!r.so) { r.err = std::string("cannot open librccl.so.1: ") + dlerror(); return &r; }
    auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.err.empty()) { dlclose(r.so); r.so = nullptr; }
    return &r;
}

// gathered: [rank][rows_pad][row_bytes] (what the all-gather leaves); frame: [height][row_bytes].
// Image row y lies in band y / 8, which rank (y / 8) % n rendered as its ((y / 8) / n)-th band.
template <typename T>
__global__ void deinterleave_kernel(const T* __restrict__ gathered, T* __restrict__ frame, int height, long long row_elems,
                                    int n, long long rows_pad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)height * row_elems) return;
    const int y = (int)(i / row_elems);
    const long long x = i - (long long)y * row_elems;
    const int band = y / RTX_TILE;
    const long long src_row = (long long)(band % n) * rows_pad + (long long)(band / n) * RTX_TILE + (y % RTX_TILE);
    frame[i] = gathered[src_row * row_elems + x];
}

int rows_padded(int vsize, int n) { return ((vsize + RTX_TILE - 1) / RTX_TILE + n - 1) / n * RTX_TILE; }

}  // namespace

struct rtx_comm {
    rtx_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;
    bool loopback = false;                       // all `world` ranks are this GPU, no RCCL (rtx_comm_init_loopback)
    hipStream_t gstream = nullptr;               // the gather runs here, beside the next frame's render
    // n_sets buffer sets (2..kSets): frame k renders into set k % n_sets while the sets of the frames before it are still
    // being rendered / gathered
    static constexpr int kSets = 4;
    int n_sets = 2;
    int n_concurrent = 1;                        // mutually concurrent streams found (lanes + gather stream)
    uint8_t* d_band[kSets] = {}; uint8_t* d_all[kSets] = {}; size_t band_cap = 0;
    int32_t* d_band_hit[kSets] = {}; int32_t* d_all_hit[kSets] = {}; size_t hit_cap = 0;
    hipEvent_t ev_render[kSets] = {}, ev_gather[kSets] = {};
    // one render lane per set (stream, side stream, tile lists, counters): frame k renders on lane k % n_sets, so the START of frame
    // k + 1 (its probe launch, its expensive tiles) fills the wave slots that the TAIL of frame k leaves empty
    rtx_ctx::Lane lane[kSets]; unsigned long long* d_counters[kSets] = {}; hipEvent_t ev_call[kSets] = {};
    bool used[kSets] = {};
    unsigned seq = 0;
    // what the last rtx_comm_render_bands queued, for the gather step
    int cur = 0; int W = 0, H = 0; bool want_hit = false;
    // the buffers of set `cur` exist and cur / W / H / want_hit describe the frame being queued: from here on a rank-local
    // failure must not keep this rank out of the all-gather (the other ranks enqueue theirs and would wait for ever)
    bool can_gather = false;
};

namespace {

int nccl_fail(rtx_ctx* ctx, const char* what, ncclResult_t r) {
    return rtx_fail(ctx, RTX_E_DEVICE, std::string(what) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(r) : "RCCL error"));
}
#define NCCLCHK(ctx, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return nccl_fail(ctx, #call, r_); } while (0)

// greedy: keep a candidate if it overlaps with everything kept so far
int pick_streams(rtx_ctx* ctx, const std::vector<hipStream_t>& pool, int want, std::vector<hipStream_t>* chosen) {
    bool ok = true;
    (void)rtx_spin_together({pool[0]}, &ok);                            // warm-up: code object load
    chosen->assign(1, pool[0]);
    for (size_t i = 1; i < pool.size() && (int)chosen->size() < want; i++) {
        std::vector<hipStream_t> trial = *chosen; trial.push_back(pool[i]);
        double best = 1e30;
        for (int k = 0; k < 2; k++) { const double t = rtx_spin_together(trial, &ok); if (t < best) best = t; }
        if (best < 160.0) chosen->push_back(pool[i]);
    }
    if (!ok) return rtx_fail(ctx, RTX_E_DEVICE, "stream calibration failed");
    return RTX_OK;
}

int comm_finish_init(rtx_comm* c) {
    rtx_ctx* ctx = c->ctx;
    c->n_sets = ctx->opt.pipeline < 2 ? 2 : (ctx->opt.pipeline > rtx_comm::kSets ? rtx_comm::kSets : ctx->opt.pipeline);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // HIP deals its few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) to streams by a policy of its own, and launches
    // on streams that share a hardware queue run one after the other.  The frames in flight must really overlap, so
    // the streams are PICKED: from a pool of candidates, by running two short timed spin kernels side by side
    // (pick_streams: ~1 ms per communicator).  Wanted: the lane streams and the gather stream mutually concurrent; the
    // side stream of a lane (SPLIT launch) concurrent with its own lane at least.
    {
        std::vector<hipStream_t> pool(2 * rtx_comm::kSets + 2, nullptr);
        for (hipStream_t& p : pool) HIPCHK(ctx, hipStreamCreateWithFlags(&p, hipStreamNonBlocking));
        std::vector<hipStream_t> chosen;
        int rc = pick_streams(ctx, pool, c->n_sets + 1, &chosen);
        if (rc != RTX_OK) { for (hipStream_t p : pool) if (p) (void)hipStreamDestroy(p); return rc; }
        // the mutually concurrent streams first (at least one was found), then the others: lanes, gather stream, and
        // what is left over as candidates for the side streams
        c->n_concurrent = (int)chosen.size();
        auto taken = [&](hipStream_t p) { for (hipStream_t q : chosen) if (q == p) return true; return false; };
        std::vector<hipStream_t> order = chosen;
        for (hipStream_t p : pool) if (!taken(p)) order.push_back(p);
        for (int b = 0; b < c->n_sets; b++) c->lane[b].stream = order[b];
        c->gstream = order[c->n_sets];
        std::vector<hipStream_t> rest(order.begin() + c->n_sets + 1, order.end());
        for (int b = 0; b < c->n_sets; b++) {
            // a leftover stream that runs beside ALL the chosen ones (a hardware queue of its own) if there is one, else
            // one that runs beside this lane's own stream at least
            size_t best = rest.size();
            for (size_t i = 0; i < rest.size() && best == rest.size(); i++) {
                if (!rest[i]) continue;
                std::vector<hipStream_t> trial = chosen; trial.push_back(rest[i]);
                bool ok = true; double t = rtx_spin_together(trial, &ok); const double t2 = rtx_spin_together(trial, &ok); if (t2 < t) t = t2;
                if (ok && t < 160.0) best = i;
            }
            for (size_t i = 0; i < rest.size() && best == rest.size(); i++)
                if (rest[i] && rtx_streams_overlap({c->lane[b].stream, rest[i]})) best = i;
            for (size_t i = 0; i < rest.size() && best == rest.size(); i++) if (rest[i]) best = i;
            c->lane[b].side = rest[best]; rest[best] = nullptr;
        }
        for (hipStream_t p : rest) if (p) (void)hipStreamDestroy(p);
    }
    for (int b = 0; b < c->n_sets; b++) {
        HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_render[b], hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_gather[b], hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&c->ev_call[b], hipEventDisableTiming));
        rtx_ctx::Lane& ln = c->lane[b];
        HIPCHK(ctx, hipEventCreateWithFlags(&ln.ev_fork, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
        HIPCHK(ctx, hipMalloc((void**)&ln.d_tile_counts, 64 * sizeof(int32_t)));
        { const int erc = rtx_lane_events(ctx, &ln); if (erc != RTX_OK) return erc; }
        HIPCHK(ctx, hipMalloc((void**)&c->d_counters[b], RTX_COUNTER_WORDS * sizeof(unsigned long long)));
    }
    return RTX_OK;
}

int comm_reserve(rtx_comm* c, int W, int H, bool want_hit) {
    rtx_ctx* ctx = c->ctx;
    const size_t band_px = (size_t)rows_padded(H, c->world) * W;
    if (band_px * 3 > c->band_cap) {
        HIPCHK(ctx, hipStreamSynchronize(c->gstream));
        for (int b = 0; b < c->n_sets; b++) {
            if (c->d_band[b]) (void)hipFree(c->d_band[b]);
            if (c->d_all[b]) (void)hipFree(c->d_all[b]);
            c->d_band[b] = c->d_all[b] = nullptr;
        }
        c->band_cap = 0;
        for (int b = 0; b < c->n_sets; b++) {
            HIPCHK(ctx, hipMalloc((void**)&c->d_band[b], band_px * 3));
            HIPCHK(ctx, hipMalloc((void**)&c->d_all[b], band_px * 3 * c->world));
            HIPCHK(ctx, hipMemset(c->d_band[b], 0, band_px * 3));              // (the padding rows of the last band are sent too)
        }
        c->band_cap = band_px * 3;
    }
    if (want_hit && band_px * 4 > c->hit_cap) {
        HIPCHK(ctx, hipStreamSynchronize(c->gstream));
        for (int b = 0; b < c->n_sets; b++) {
            if (c->d_band_hit[b]) (void)hipFree(c->d_band_hit[b]);
            if (c->d_all_hit[b]) (void)hipFree(c->d_all_hit[b]);
            c->d_band_hit[b] = c->d_all_hit[b] = nullptr;
        }
        c->hit_cap = 0;
        for (int b = 0; b < c->n_sets; b++) {
            HIPCHK(ctx, hipMalloc((void**)&c->d_band_hit[b], band_px * 4));
            HIPCHK(ctx, hipMalloc((void**)&c->d_all_hit[b], band_px * 4 * c->world));
            HIPCHK(ctx, hipMemset(c->d_band_hit[b], 0xff, band_px * 4));
        }
        c->hit_cap = band_px * 4;
    }
    return RTX_OK;
}

// step 1: this rank's bands {rank, world}, compact, into the current buffer set, on `st`
int comm_render_bands(rtx_comm* c, const rtx_scene* scene, const rtx_camera* cam, int max_depth, uint32_t flags,
                      bool want_hit, hipStream_t st, rtx_stats* stats) {
    rtx_ctx* ctx = c->ctx;
    if (!scene || !cam) return RTX_E_INVALID;
    if (cam->hsize <= 0 || cam->vsize <= 0) return rtx_fail(ctx, RTX_E_INVALID, "empty camera");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c->can_gather = false;
    int rc = comm_reserve(c, cam->hsize, cam->vsize, want_hit);
    if (rc != RTX_OK) return rc;                 // (no buffers: this rank cannot take part; an allocation failure is fatal for the job)
    const int b = (int)(c->seq % (unsigned)c->n_sets);
    c->cur = b; c->W = cam->hsize; c->H = cam->vsize; c->want_hit = want_hit;
    c->can_gather = true;
    int deferred = RTX_OK;                       // a rank-local error that is reported AFTER the collective has been queued
    if (c->used[b] && (stats || ctx->opt.pipeline < 2)) HIPCHK(ctx, hipStreamWaitEvent(st, c->ev_gather[b], 0));   // the set's previous gather has read it
    // asynchronous calls render on the set's own lane, behind whatever the caller's stream holds now; a call that wants
    // its counts back is waited for anyway and stays on the caller's stream with the context's counters
    hipStream_t rs = st;
    rtx_ctx::Lane* lane = nullptr;
    if (!stats && ctx->opt.pipeline >= 2) {
        lane = &c->lane[b];
        rs = lane->stream;
        // behind whatever the caller's stream holds NOW — asked first, because an event recorded on an idle stream is not
        // free: its marker lands on that stream's hardware queue, which may be the queue of the frame in flight, and the
        // new frame would then wait for the old one to finish
        if (hipStreamQuery(st) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(ctx, hipEventRecord(c->ev_call[b], st));
            HIPCHK(ctx, hipStreamWaitEvent(rs, c->ev_call[b], 0));
        }
        if (c->used[b]) HIPCHK(ctx, hipStreamWaitEvent(rs, c->ev_gather[b], 0));
        // an earlier frame of THIS rank overflowed: the frame is still rendered and gathered (the other ranks cannot know and
        // are queueing their all-gather right now), the error is returned once everything is queued
        deferred = rtx_check_sticky(ctx);
    }
    // a caller that streams frames (the previous frame is still rendering when this one is queued) is after throughput: the
    // tail of a frame is filled by the start of the next, so the measures that shorten ONE frame's critical path at the
    // price of extra work (the 4-way SPLIT launch; for short dispatches also the tile-order probe) are left out.  A caller
    // that waits for every frame gets them as before.
    uint32_t sched = 0;
    if (lane && c->seq > 0) {
        const int prev = (int)((c->seq - 1u) % (unsigned)c->n_sets);
        if (hipEventQuery(c->ev_render[prev]) == hipErrorNotReady) {
            // (the probe still pays for long dispatches: measured on S4 with three frames in flight, tools/pipeline_probe.py,
            // with / without it: 4.14 / 4.36 ms at N = 1, 2.14 / 2.18 at N = 2, 1.17 / 1.15 at N = 4, 0.67 / 0.63 at N = 8)
            const long long share_tiles = (long long)((cam->hsize + RTX_TILE - 1) / RTX_TILE) * (rows_padded(cam->vsize, c->world) / RTX_TILE);
            // (with the tile-cost feedback the lists cost a classify launch of microseconds from the second frame of a view on,
            // not a probe launch: they are kept for short dispatches too)
            const bool keep_probe = ctx->opt.stream_lpt >= 0 ? ctx->opt.stream_lpt != 0 : (share_tiles >= 40000 || ctx->opt.cost_feedback != 0);
            sched = RTX_F_INTERNAL_NO_SPLIT | (keep_probe ? 0u : RTX_F_INTERNAL_NO_LPT);
        }
        (void)hipGetLastError();
    }
    auto render = [&](rtx_bands bands, void* d_rgb, void* d_hit, rtx_stats* out) {
        return lane ? rtx_enqueue_render(ctx, scene, cam, max_depth, bands, flags | RTX_F_BANDS_COMPACT | sched, d_rgb, d_hit, rs, lane,
                                         c->d_counters[b], nullptr, nullptr, true)
                    : rtx_render_device(ctx, scene, cam, max_depth, bands, flags | RTX_F_BANDS_COMPACT, d_rgb, d_hit, rs, out);
    };
    if (c->loopback) {
        // every rank's share in turn, each where the all-gather would have put it
        const size_t band_px = (size_t)rows_padded(cam->vsize, c->world) * cam->hsize;
        rtx_stats sum{}, one{};
        const int n_shares = ctx->opt.debug == 3 ? 1 : c->world;    // (debug 3: rank 0's share only — tools/pipeline_probe.py)
        for (int r = 0; r < n_shares; r++) {
            rc = render(rtx_bands{r, c->world}, c->d_all[b] + (size_t)r * band_px * 3,
                        want_hit ? c->d_all_hit[b] + (size_t)r * band_px : nullptr, stats ? &one : nullptr);
            if (rc != RTX_OK) break;
            if (stats) {
                sum.rays_primary += one.rays_primary; sum.rays_shadow += one.rays_shadow; sum.rays_secondary += one.rays_secondary;
                for (int k = 0; k < RTX_EV_COUNT; k++) sum.ev[k] += one.ev[k];
                sum.overflow_pixels += one.overflow_pixels; sum.rays_shadow_skipped += one.rays_shadow_skipped;
                sum.kernel_ms += one.kernel_ms;
            }
        }
        if (stats) *stats = sum;
    } else {
        rc = render(rtx_bands{c->rank, c->world}, c->d_band[b], want_hit ? c->d_band_hit[b] : nullptr, stats);
    }
    st = rs;
    // (also after a failed render: the gather must be ordered behind whatever part of it was queued)
    const hipError_t he = hipEventRecord(c->ev_render[b], st);
    if (rc != RTX_OK) return rc;
    if (he != hipSuccess) return rtx_fail(ctx, RTX_E_DEVICE, std::string("hipEventRecord: ") + hipGetErrorString(he));
    return deferred;
}

// step 2 (inside a group when one thread drives several ranks): the all-gather of the current set, on gstream
int comm_all_gather(rtx_comm* c) {
    rtx_ctx* ctx = c->ctx;
    const int b = c->cur;
    const size_t band_px = (size_t)rows_padded(c->H, c->world) * c->W;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamWaitEvent(c->gstream, c->ev_render[b], 0));
    if (c->loopback) return RTX_OK;              // the shares were rendered into place
    NCCLCHK(ctx, rccl()->AllGather(c->d_band[b], c->d_all[b], band_px * 3, ncclUint8, c->nccl, c->gstream));
    if (c->want_hit) NCCLCHK(ctx, rccl()->AllGather(c->d_band_hit[b], c->d_all_hit[b], band_px, ncclInt32, c->nccl, c->gstream));
    return RTX_OK;
}

// step 3: de-interleave into the caller's row-major frame (gstream), mark the set reusable
int comm_deinterleave(rtx_comm* c, void* d_frame, void* d_hit_frame) {
    rtx_ctx* ctx = c->ctx;
    const int b = c->cur;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long long rows_pad = rows_padded(c->H, c->world);
    if (d_frame) {
        const long long row_bytes = (long long)c->W * 3;
        if (row_bytes % 16 == 0 && ((uintptr_t)d_frame % 16) == 0) {
            const long long n = (long long)c->H * (row_bytes / 16);
            hipLaunchKernelGGL(deinterleave_kernel<uint4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->gstream,
                               (const uint4*)c->d_all[b], (uint4*)d_frame, c->H, row_bytes / 16, c->world, rows_pad);
        } else {
            const long long n = (long long)c->H * row_bytes;
            hipLaunchKernelGGL(deinterleave_kernel<uint8_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->gstream,
                               (const uint8_t*)c->d_all[b], (uint8_t*)d_frame, c->H, row_bytes, c->world, rows_pad);
        }
        HIPCHK(ctx, hipGetLastError());
    }
    if (d_hit_frame && c->want_hit) {
        const long long n = (long long)c->H * c->W;
        hipLaunchKernelGGL(deinterleave_kernel<int32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->gstream,
                           (const int32_t*)c->d_all_hit[b], (int32_t*)d_hit_frame, c->H, (long long)c->W, c->world, rows_pad);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(c->ev_gather[b], c->gstream));
    c->used[b] = true;
    c->seq++;
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_comm_unique_id(void* id) {
    if (!id) return RTX_E_INVALID;
    Rccl* r = rccl();
    if (!r->so) return RTX_E_DEVICE;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return RTX_E_DEVICE;
    static_assert(sizeof(u) == RTX_COMM_ID_BYTES, "RTX_COMM_ID_BYTES must be RCCL's unique id size");
    std::memcpy(id, &u, sizeof u);
    return RTX_OK;
}

int rtx_comm_init_rank(rtx_ctx* ctx, const void* id, int rank, int world_size, rtx_comm** out) {
    if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return RTX_E_INVALID;
    *out = nullptr;
    Rccl* r = rccl();
    if (!r->so) return rtx_fail(ctx, RTX_E_DEVICE, r->err);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rtx_comm* c = new rtx_comm();
    c->ctx = ctx; c->rank = rank; c->world = world_size;
    ncclUniqueId u; std::memcpy(&u, id, sizeof u);
    ncclResult_t nr = r->CommInitRank(&c->nccl, world_size, u, rank);
    if (nr != ncclSuccess) { delete c; return nccl_fail(ctx, "ncclCommInitRank", nr); }
    int rc = comm_finish_init(c);
    if (rc != RTX_OK) { rtx_comm_destroy(c); return rc; }
    *out = c;
    return RTX_OK;
}

int rtx_comm_init_all(rtx_ctx* const* ctxs, int n, rtx_comm** out) {
    if (!ctxs || !out || n < 1) return RTX_E_INVALID;
    for (int i = 0; i < n; i++) { out[i] = nullptr; if (!ctxs[i]) return RTX_E_INVALID; }
    Rccl* r = rccl();
    if (!r->so) return rtx_fail(ctxs[0], RTX_E_DEVICE, r->err);
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; j++) if (devs[j] == devs[i]) return rtx_fail(ctxs[0], RTX_E_INVALID, "rtx_comm_init_all: two contexts on one device");
    }
    std::vector<ncclComm_t> nc(n, nullptr);
    ncclResult_t nr = r->CommInitAll(nc.data(), n, devs.data());
    if (nr != ncclSuccess) return nccl_fail(ctxs[0], "ncclCommInitAll", nr);
    int rc = RTX_OK;
    for (int i = 0; i < n; i++) {
        rtx_comm* c = new rtx_comm();
        c->ctx = ctxs[i]; c->rank = i; c->world = n; c->nccl = nc[i];
        out[i] = c;
        if (rc == RTX_OK) rc = comm_finish_init(c);
    }
    if (rc != RTX_OK) for (int i = 0; i < n; i++) { rtx_comm_destroy(out[i]); out[i] = nullptr; }
    return rc;
}

int rtx_comm_init_loopback(rtx_ctx* ctx, int world_size, rtx_comm** out) {
    if (!ctx || !out || world_size < 1) return RTX_E_INVALID;
    *out = nullptr;
    rtx_comm* c = new rtx_comm();
    c->ctx = ctx; c->rank = 0; c->world = world_size; c->loopback = true;
    int rc = comm_finish_init(c);
    if (rc != RTX_OK) { rtx_comm_destroy(c); return rc; }
    *out = c;
    return RTX_OK;
}

void rtx_comm_destroy(rtx_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    if (c->gstream) (void)hipStreamSynchronize(c->gstream);
    if (c->nccl && rccl()->so) (void)rccl()->CommDestroy(c->nccl);
    for (int b = 0; b < c->n_sets; b++) {
        for (void* p : {(void*)c->d_band[b], (void*)c->d_all[b], (void*)c->d_band_hit[b], (void*)c->d_all_hit[b]}) if (p) (void)hipFree(p);
        if (c->ev_render[b]) (void)hipEventDestroy(c->ev_render[b]);
        if (c->ev_gather[b]) (void)hipEventDestroy(c->ev_gather[b]);
        if (c->ev_call[b]) (void)hipEventDestroy(c->ev_call[b]);
        rtx_ctx::Lane& ln = c->lane[b];
        if (ln.stream) {
            (void)hipStreamSynchronize(ln.stream);
            if (c->ctx->last_stream == ln.stream) c->ctx->last_stream = nullptr;
            (void)hipStreamDestroy(ln.stream);
        }
        if (ln.side) (void)hipStreamDestroy(ln.side);
        if (ln.ev_fork) (void)hipEventDestroy(ln.ev_fork);
        if (ln.ev_join) (void)hipEventDestroy(ln.ev_join);
        if (ln.d_lists) (void)hipFree(ln.d_lists);
        if (ln.d_tile_counts) (void)hipFree(ln.d_tile_counts);
        ln.release_ahead();
        if (c->d_counters[b]) (void)hipFree(c->d_counters[b]);
    }
    if (c->gstream) (void)hipStreamDestroy(c->gstream);
    delete c;
}

int rtx_comm_rank(const rtx_comm* c) { return c ? c->rank : -1; }
int rtx_comm_size(const rtx_comm* c) { return c ? c->world : 0; }
int rtx_comm_concurrency(const rtx_comm* c) { return c ? c->n_concurrent : 0; }

int rtx_render_gathered(rtx_comm* c, const rtx_scene* scene, const rtx_camera* cam, int max_depth, uint32_t flags,
                        void* d_frame, void* d_hit_frame, void* stream, rtx_stats* stats) {
    if (!c || !d_frame) return RTX_E_INVALID;
    hipStream_t st = stream ? (hipStream_t)stream : c->ctx->stream;
    // The all-gather is a COLLECTIVE: once this rank's buffers exist it is queued whatever happened to the rank's own render
    // (an earlier frame's overflow, a failed launch: its band buffer then holds the previous or a zero-filled frame), and the
    // rank-local error is returned afterwards.  A rank that skipped it would leave the others waiting for ever on their gather
    // streams.  Only a failure before the buffers exist (invalid arguments - the same on every rank - or out of memory) returns
    // without it.
    const int rc_local = comm_render_bands(c, scene, cam, max_depth, flags, d_hit_frame != nullptr, st, stats);
    if (rc_local != RTX_OK && !c->can_gather) return rc_local;
    const std::string local_err = rc_local != RTX_OK ? c->ctx->err : std::string();
    int rc = comm_all_gather(c);
    if (rc == RTX_OK) rc = comm_deinterleave(c, d_frame, d_hit_frame);
    if (rc_local != RTX_OK) { c->ctx->err = local_err; return rc_local; }
    return rc;
}

int rtx_comm_wait(rtx_comm* c, void* stream) {
    if (!c) return RTX_E_INVALID;
    rtx_ctx* ctx = c->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!stream) { HIPCHK(ctx, hipStreamSynchronize(c->gstream)); return rtx_ctx_poll(ctx, 0); }
    const int last = (int)((c->seq + (unsigned)c->n_sets - 1u) % (unsigned)c->n_sets);                    // the set of the most recent frame
    if (c->used[last]) HIPCHK(ctx, hipStreamWaitEvent((hipStream_t)stream, c->ev_gather[last], 0));
    return RTX_OK;
}

int rtx_render_multi(rtx_comm* const* comms, const rtx_scene* const* scenes, int n, const rtx_camera* cam, int max_depth,
                     uint32_t flags, uint8_t* rgb, int32_t* hit, rtx_stats* stats) {
    if (!comms || !scenes || n < 1 || !cam || !rgb) return RTX_E_INVALID;
    for (int i = 0; i < n; i++) if (!comms[i] || !scenes[i] || comms[i]->world != n || comms[i]->rank != i) return RTX_E_INVALID;
    rtx_ctx* ctx0 = comms[0]->ctx;
    if (cam->hsize <= 0 || cam->vsize <= 0) return rtx_fail(ctx0, RTX_E_INVALID, "empty camera");
    const size_t npx = (size_t)cam->hsize * cam->vsize;
    // the whole frame lands on rank 0's GPU (every rank receives all bands; only rank 0 de-interleaves here)
    HIPCHK(ctx0, hipSetDevice(ctx0->device));
    if (npx * 3 > ctx0->rgb_cap) {
        if (ctx0->d_rgb) (void)hipFree(ctx0->d_rgb);
        ctx0->d_rgb = nullptr; ctx0->rgb_cap = 0;
        HIPCHK(ctx0, hipMalloc((void**)&ctx0->d_rgb, npx * 3));
        ctx0->rgb_cap = npx * 3;
    }
    if (hit && npx * 4 > ctx0->hit_cap) {
        if (ctx0->d_hit) (void)hipFree(ctx0->d_hit);
        ctx0->d_hit = nullptr; ctx0->hit_cap = 0;
        HIPCHK(ctx0, hipMalloc((void**)&ctx0->d_hit, npx * 4));
        ctx0->hit_cap = npx * 4;
    }
    // One thread drives all ranks, so the ranks agree on errors by construction — provided every check that can fail on ONE
    // rank only is made before anything collective is queued: the sticky words and the buffers of all ranks first ...
    int rc = RTX_OK;
    for (int i = 0; i < n && rc == RTX_OK; i++) {
        rc = rtx_check_sticky(comms[i]->ctx);
        if (rc == RTX_OK && hipSetDevice(comms[i]->ctx->device) != hipSuccess) rc = rtx_fail(comms[i]->ctx, RTX_E_DEVICE, "hipSetDevice failed");
        if (rc == RTX_OK) rc = comm_reserve(comms[i], cam->hsize, cam->vsize, hit != nullptr);
        if (rc != RTX_OK && i > 0) ctx0->err = comms[i]->ctx->err;
    }
    if (rc != RTX_OK) return rc;
    // ... then every GPU renders its bands (queued, nothing waits; a failure here ends the call before any rank gathers) ...
    std::vector<rtx_stats> per(n);
    for (int i = 0; i < n && rc == RTX_OK; i++) {
        rc = comm_render_bands(comms[i], scenes[i], cam, max_depth, flags, hit != nullptr, comms[i]->ctx->stream, stats ? &per[i] : nullptr);
        if (rc != RTX_OK && i > 0) ctx0->err = comms[i]->ctx->err;
    }
    if (rc != RTX_OK) return rc;
    // ... one all-gather over all ranks (one thread drives them: a group).  The HIP calls of the step (device switch, the wait
    // for the render) are made for ALL ranks before the group opens, so that only ncclAllGather itself runs inside it.
    for (int i = 0; i < n; i++) {
        rtx_comm* c = comms[i];
        if (hipSetDevice(c->ctx->device) != hipSuccess || hipStreamWaitEvent(c->gstream, c->ev_render[c->cur], 0) != hipSuccess)
            return rtx_fail(ctx0, RTX_E_DEVICE, "rtx_render_multi: cannot order the gather behind the render");
    }
    NCCLCHK(ctx0, rccl()->GroupStart());
    for (int i = 0; i < n && rc == RTX_OK; i++) rc = comm_all_gather(comms[i]);
    { const ncclResult_t ge = rccl()->GroupEnd(); if (ge != ncclSuccess && rc == RTX_OK) rc = nccl_fail(ctx0, "ncclGroupEnd", ge); }
    // ... rank 0 assembles the frame and hands it to the host
    for (int i = 0; i < n && rc == RTX_OK; i++) rc = comm_deinterleave(comms[i], i == 0 ? ctx0->d_rgb : nullptr, i == 0 && hit ? ctx0->d_hit : nullptr);
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(comms[i]->ctx->device);
        if (hipStreamSynchronize(comms[i]->gstream) != hipSuccess && rc == RTX_OK) rc = rtx_fail(ctx0, RTX_E_DEVICE, "hipStreamSynchronize failed");
        if (rc == RTX_OK) { rc = rtx_ctx_poll(comms[i]->ctx, 0); if (rc != RTX_OK && i > 0) ctx0->err = comms[i]->ctx->err; }
    }
    if (rc != RTX_OK) return rc;
    HIPCHK(ctx0, hipSetDevice(ctx0->device));
    HIPCHK(ctx0, hipMemcpy(rgb, ctx0->d_rgb, npx * 3, hipMemcpyDeviceToHost));
    if (hit) HIPCHK(ctx0, hipMemcpy(hit, ctx0->d_hit, npx * 4, hipMemcpyDeviceToHost));
    if (stats) {
        // the frame's counts are the sums of the ranks' shares (with stats every rank's launch was waited for in turn,
        // so kernel_ms is the slowest rank's, not a concurrent measurement)
        std::memset(stats, 0, sizeof *stats);
        for (int i = 0; i < n; i++) {
            stats->rays_primary += per[i].rays_primary; stats->rays_shadow += per[i].rays_shadow; stats->rays_secondary += per[i].rays_secondary;
            for (int k = 0; k < RTX_EV_COUNT; k++) stats->ev[k] += per[i].ev[k];
            stats->overflow_pixels += per[i].overflow_pixels; stats->rays_shadow_skipped += per[i].rays_shadow_skipped;
            if (per[i].kernel_ms > stats->kernel_ms) stats->kernel_ms = per[i].kernel_ms;
        }
    }
    return RTX_OK;
}

int rtx_band_rows_padded(int vsize, int n_ranks) { return (vsize > 0 && n_ranks > 0) ? rows_padded(vsize, n_ranks) : 0; }

int rtx_band_deinterleave(const uint8_t* gathered, int n_ranks, int width, int height, int bytes_per_pixel, uint8_t* frame) {
    if (!gathered || !frame || n_ranks < 1 || width <= 0 || height <= 0 || bytes_per_pixel <= 0) return RTX_E_INVALID;
    const size_t row = (size_t)width * bytes_per_pixel;
    const size_t rows_pad = (size_t)rows_padded(height, n_ranks);
    for (int y = 0; y < height; y++) {
        const int band = y / RTX_TILE;
        const size_t src = (size_t)(band % n_ranks) * rows_pad + (size_t)(band / n_ranks) * RTX_TILE + (y % RTX_TILE);
        std::memcpy(frame + (size_t)y * row, gathered + src * row, row);
    }
    return RTX_OK;
}

}  // extern "C"
