#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + shadow + secondary) of the MI355X trace-and-shade path.

A "step" is one frame: every rank renders its row-interleaved 8-row tile bands of the SAME frame
(rank r gets bands r, r+N, ...) with the scene resident in HBM, and the bands are all-gathered and
de-interleaved INSIDE the product (rtx_render_gathered in include/rtx.h: one ncclAllGather over
RCCL/xGMI on the library's own stream, overlapped with the next frame's render).  EVERY N takes that
entry point, N = 1 included (a one-rank RCCL communicator: the same calls, the gather degenerates to a
copy), so the 1 -> 8 curve compares like with like: `value` / `ms_per_step` are the streamed rate (the
K frames are queued back to back, up to three in flight on the library's render lanes) at every N, and
`ms_per_frame_one_at_a_time` is the latency of one frame (queue, wait, repeat) at every N.  Total work
per step is fixed as N grows -> "strong" scaling; value = whole-job rays per second.

Launch: `python bench.py --gpus N ...`.  For N > 1 without a torch.distributed environment the script
starts its N ranks itself (a child `python -m torch.distributed.run`, before anything touches the
GPU) and forwards the exit code; under torch.distributed.run it is one of the ranks.

Extra objects on the JSON line: `roofline` (FP64 vector ALU bound: algorithmic flops of the frame from
the kernel's own event counters / the kernel's launch time measured with HIP events on the launch
stream; `valu_issue_frac`, `resident_waves` and `traffic` come from the rocprofv3 PMC passes recorded in
profiles/<tag>_counters.json together with the git revision they were taken at), `cpu_baseline` (the
CPU oracle — a C++ restatement of the single-threaded reference, `kind: "port"` — timed on a bounded
tile sample of the same frame on this box's host cores, rank 0, N = 1 only) and `secondary` / `secondary_c5` (the other
3840x2160 configuration, S3, and the 7680x4320 CSG configuration, S5, measured the same way in the same run).  At N = 1 the line also carries
`kernel_region` (the same K frames through rtx_render_device, one HIP event pair per launch on the
launch stream: the average launch duration the roofline is priced with) and `host_render_ms`
(rtx_render of the same frame into pageable host memory: what Camera::render, camera.rs:39-54, returns).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOADS = {
    # name: (scene, width, height, description)
    "S2": ("S2", 1920, 1080, "synthetic 1920x1080 spheres+planes, 1 light, shadows only"),
    "S3": ("S3", 3840, 2160, "synthetic 3840x2160, 10240 spheres in nested groups, Phong + patterns"),
    "S4": ("S4", 3840, 2160, "synthetic 3840x2160 glass/mirror scene, reflection+refraction depth 5"),
    "S5": ("S5", 7680, 4320, "synthetic 7680x4320 CSG + cubes + patterns"),
}


def cpu_baseline(text, width, height, budget_s=15.0):
    """Times the CPU oracle (oracle/: test infrastructure, here only as the reported baseline) on a
    seeded sample of 8x8 tiles of the same frame, single-threaded like the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import ora
    import yaml_scene
    w, cams = yaml_scene.load(text, ora.API)
    cam = cams[0]
    tx, ty = (width + 7) // 8, (height + 7) // 8
    rng = np.random.default_rng(20261111)
    order = rng.permutation(tx * ty)

    def pixels(tiles):
        xs, ys = [], []
        for t in tiles:
            x0, y0 = (t % tx) * 8, (t // tx) * 8
            gx, gy = np.meshgrid(np.arange(x0, min(x0 + 8, width)), np.arange(y0, min(y0 + 8, height)))
            xs.append(gx.ravel()); ys.append(gy.ravel())
        return np.concatenate(xs), np.concatenate(ys)

    probe = order[:16]
    r = ora.render(w, cam, 5, pixels=pixels(probe), threads=1)
    per_tile = max(r["stats"].ms / 1000.0 / len(probe), 1e-6)
    n = int(min(len(order) - 16, max(16, budget_s / per_tile)))
    tiles = order[16:16 + n]
    r = ora.render(w, cam, 5, pixels=pixels(tiles), threads=1)
    st = r["stats"]
    secs = st.ms / 1000.0
    return {"value": st.rays / secs / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": "%d seeded 8x8 tiles (%d px, %d rays) of the same frame in %.1f s, 1 thread; "
                      "C++ restatement of the single-threaded reference (no Rust toolchain on the box)"
                      % (len(tiles), st.rays_primary, st.rays, secs),
            "ms_per_frame_extrapolated": secs * (width * height) / st.rays_primary * 1000.0}


def self_launch(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks (one process per GPU) as a child
    `python -m torch.distributed.run` and return its exit code.  Nothing in THIS process has touched the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def len_lights(text):
    return text.count("- add: light")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="S4", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--kernel-only", action="store_true",
                    help="N = 1: only the kernel region (K whole-frame launches through rtx_render_device) — what the rocprofv3 PMC "
                         "passes profile, so that every dispatch of the frame kernel they average over is one whole frame")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if world_size != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU, or let bench.py start them: "
                         "run it without a torch.distributed environment)" % (args.gpus, world_size))

    # ONE JSON line on stdout: libraries below print banners of their own through C stdio (RCCL's version block at the
    # first communicator), so file descriptor 1 is pointed at stderr for the run and the line goes to the saved one
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import flops
    import rtx
    import scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (2 ranks, gloo, one GPU):
    # RTX_BENCH_BACKEND=gloo maps every rank onto the GPUs that exist and gathers through host memory + the C ABI's host
    # de-interleave, because RCCL refuses two ranks on one device.  The driver's runs never set it.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    N = world_size
    dist = None
    # rehearsal of the communicator path on one GPU: RTX_BENCH_FORCE_COMM=1 runs N = 1 through torch.distributed (nccl, one
    # rank) and rtx_render_gathered exactly as N > 1 does.  The driver's runs never set it.
    force_comm = N == 1 and os.environ.get("RTX_BENCH_FORCE_COMM") == "1"
    if force_comm:
        os.environ.setdefault("MASTER_PORT", "29533")
    if N > 1 or force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=N)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace-and-shade path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    def all_reduce(t, op=None):
        if backend == "nccl":
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
        else:
            c = t.cpu(); dist.all_reduce(c, op=op or dist.ReduceOp.SUM); t.copy_(c)

    ctx = rtx.Context(local_rank)
    # a real (non-null) stream for everything below: the kernel launches and the timing events are then one queue
    compute = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(compute)
    stream = compute.cuda_stream

    # the product's communicator for the band gather (RCCL inside librtx_hip.so); torch.distributed only carries the 128-byte id,
    # the barrier and the max-over-ranks of the timings.  Should the product's communicator not come up (all ranks decide
    # together), the run falls back to torch.distributed's all-gather of the same compact bands and says so in the line,
    # so that a scaling run still yields a labelled number; RTX_BENCH_NO_PRODUCT_COMM=1 rehearses that path.
    comm, comm_error = None, None
    comm_kind = "rccl"
    torch_gather = False
    if N == 1 and not force_comm and not args.kernel_only:
        # N = 1: the same entry point over a one-rank RCCL communicator (no torch.distributed needed: the id stays here).  If
        # RCCL does not come up, the library's loopback communicator (same render lanes, same de-interleave, no ncclAllGather)
        # takes its place and the line says so.
        try:
            comm = rtx.Comm(ctx, rtx.comm_unique_id(), 0, 1)
        except rtx.RtxError as e:
            comm_error = str(e); comm_kind = "loopback"
            comm = rtx.Comm.loopback(ctx, 1)
            print("bench.py: one-rank RCCL communicator did not come up (%s); using the loopback communicator" % comm_error,
                  file=sys.stderr, flush=True)
    if (N > 1 or force_comm) and backend == "nccl":
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        idt = torch.zeros(rtx.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                if os.environ.get("RTX_BENCH_NO_PRODUCT_COMM") == "1":
                    raise rtx.RtxError(-1, "RTX_BENCH_NO_PRODUCT_COMM=1")
                idt.copy_(torch.frombuffer(bytearray(rtx.comm_unique_id()), dtype=torch.uint8))
            except rtx.RtxError as e:
                ok.zero_(); comm_error = str(e)
        dist.broadcast(idt, 0)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) == 1:
            try:
                comm = rtx.Comm(ctx, bytes(idt.cpu().numpy().tobytes()), rank, N)
            except rtx.RtxError as e:
                ok.zero_(); comm_error = str(e)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok) == 0 and comm is not None:
                comm.close(); comm = None
        if comm is None:
            torch_gather = True
            print("bench.py: the product's communicator did not come up (%s); gathering through torch.distributed" % comm_error,
                  file=sys.stderr, flush=True)

    def measure(workload, width, height, steps, warmup):
        """-> dict of this workload's numbers (on every rank; only rank 0 prints)"""
        scene_name, w0, h0, desc = WORKLOADS[workload]
        width, height = width or w0, height or h0
        text = scenes.make(scene_name, width=width, height=height)
        world, cams = rtx.load_yaml(text)
        cam = cams[0]
        scene = ctx.scene(world)                      # scene resident in HBM from here on
        bands = (rank, N)
        rows_pad = rtx.LIB.rtx_band_rows_padded(height, N)
        local = torch.zeros(rows_pad * width * 3, dtype=torch.uint8, device=dev)
        frames = [torch.zeros(height * width * 3, dtype=torch.uint8, device=dev) for _ in range(4)] if comm is not None else None
        gathered_host = torch.zeros((N, rows_pad * width * 3), dtype=torch.uint8) if (N > 1 and comm is None and not torch_gather) else None
        if torch_gather:     # fallback gather: frame row y is row (y // 8 // N) * 8 + y % 8 of rank (y // 8) % N's compact buffer
            y = torch.arange(height, device=dev)
            src_row = ((y // 8) % N) * rows_pad + (y // 8 // N) * 8 + y % 8
            gathered_dev = torch.zeros((N * rows_pad, width * 3), dtype=torch.uint8, device=dev)
            frame_dev = torch.zeros((height, width * 3), dtype=torch.uint8, device=dev)

        # exact ray and event counts of this rank's share (untimed); a device-side overflow would raise here.  Two passes:
        # the reference-faithful one (no shadow early-out, no skipped shadow rays, loose CSG spheres: its event counts are the
        # ones the GPU tests prove equal to the oracle's) prices the ALGORITHMIC work, the product-mode one says what the
        # timed kernel really walks and executes
        ref = scene.render_device(cam, local.data_ptr(), 0, 5, bands,
                                  rtx.F_BANDS_COMPACT | rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT, stream, True)
        st = scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT | rtx.F_COUNT_EVENTS, stream, True)
        assert st.overflow_pixels == 0 and ref.overflow_pixels == 0
        assert (ref.rays_primary, ref.rays_shadow, ref.rays_secondary) == (st.rays_primary, st.rays_shadow, st.rays_secondary)
        evs, revs = st.events(), ref.events()
        nl = len_lights(text)
        counts = torch.tensor([st.rays_primary, st.rays_shadow, st.rays_secondary,
                               flops.frame_flops(revs, st.rays_primary, st.rays_shadow, nl),
                               flops.executed_flops(evs, st.rays_primary, st.rays_shadow, nl),
                               evs["cull"], evs["x_xform"], revs["xform"], st.rays_shadow_skipped, evs["xform"]], dtype=torch.float64, device=dev)
        if N > 1:
            all_reduce(counts)
        rays_frame = int(counts[0] + counts[1] + counts[2])

        def step(k):
            if comm is not None:
                comm.render_gathered(scene, cam, frames[k & 3].data_ptr(), 0, 5, 0, stream)
            elif torch_gather:
                scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
                dist.all_gather_into_tensor(gathered_dev.view(-1), local)
                torch.index_select(gathered_dev, 0, src_row, out=frame_dev)
            else:
                scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
                if N > 1:      # gloo rehearsal: host gather + the C ABI's host de-interleave
                    torch.cuda.current_stream().synchronize()
                    parts = [torch.empty_like(gathered_host[0]) for _ in range(N)]
                    dist.all_gather(parts, local.cpu())
                    rtx.band_deinterleave(torch.stack(parts).numpy(), N, width, height)

        for k in range(warmup):
            step(k)
        if comm is not None:
            comm.wait()
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        if comm is not None:
            comm.wait()                               # host waits for the last gather + de-interleave (and polls the sticky error)
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()
        t1 = time.perf_counter()
        ctx.poll(wait=True)                           # an overflow in any timed frame is an error, not a number
        if os.environ.get("RTX_BENCH_CHECK") == "1" and (comm is not None or torch_gather):
            # rehearsal aid: the gathered frame of the last timed step must equal this GPU's own render of the whole frame
            whole = torch.zeros(height * width * 3, dtype=torch.uint8, device=dev)
            scene.render_device(cam, whole.data_ptr(), 0, 5, (0, 1), 0, stream, True)
            got = frame_dev.view(-1) if torch_gather else frames[(steps - 1) & 3]
            assert torch.equal(got, whole), "gathered frame differs from the single-GPU frame"
            print("bench.py: gathered frame == single-GPU frame (rank %d)" % rank, file=sys.stderr, flush=True)
        kernel_region = None
        if N == 1:
            # the roofline's kernel time: the same K frames through rtx_render_device, queued back to back on `stream`, one HIP
            # event pair per launch on that stream (inside the region above, consecutive frames run on alternating render lanes
            # and overlap, so a launch's own duration cannot be read off it)
            for k in range(warmup):
                scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            k0 = time.perf_counter()
            for k in range(steps):
                ev[k][0].record()
                scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
                ev[k][1].record()
            torch.cuda.synchronize()
            k1 = time.perf_counter()
            ctx.poll(wait=True)
            kern = sum(a.elapsed_time(b) for a, b in ev) / steps
            kernel_region = {"entry_point": "rtx_render_device", "steps": steps, "ms_per_step": (k1 - k0) / steps * 1000.0,
                             "kernel_ms_avg": kern, "mrays_per_s": rays_frame * steps / (k1 - k0) / 1e6}
        else:
            # per-rank kernel time of its share, from the library's own HIP events (3 synchronous renders after the timed region:
            # inside it consecutive frames run on alternating render lanes and overlap each other and the gathers)
            kern = min(scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, True).kernel_ms for _ in range(3))
        # one frame at a time through the same entry point (render + gather + de-interleave, the host waits for each): the
        # "ms/frame" a caller sees who needs every frame before queuing the next; the timed region above streams frames
        one = 0.0
        if comm is not None:
            lat = []
            for k in range(3):
                if N > 1:
                    dist.barrier()
                a = time.perf_counter()
                comm.render_gathered(scene, cam, frames[0].data_ptr(), 0, 5, 0, stream)
                comm.wait()
                lat.append(time.perf_counter() - a)
            one = min(lat) * 1000.0
        host_ms = None
        if N == 1 and not args.kernel_only:
            # what Camera::render (camera.rs:39-54) returns: the frame in HOST memory (pageable, as a caller's Vec would be).
            # The first whole-frame rtx_render of a context takes the one-launch path, the later ones the band sets.
            import numpy as np
            img = np.zeros((height, width, 3), np.uint8)
            lat = []
            for k in range(4):
                a = time.perf_counter()
                scene.render(cam, 5, want_hit=False, out=(img, None))
                lat.append((time.perf_counter() - a) * 1000.0)
            host_ms = {"first_call": lat[0], "steady": min(lat[1:]), "note": "rtx_render into pageable host memory, wall clock of the call"}
        t = torch.tensor([t1 - t0, kern, one], dtype=torch.float64, device=dev)
        if N > 1:
            all_reduce(t, dist.ReduceOp.MAX)
        elapsed, kern_ms, one_ms = float(t[0]), float(t[1]), float(t[2])
        scene.close()
        full_size = (width, height) == (w0, h0)
        pmc = flops.pmc_counters(workload) if (N == 1 and full_size) else None
        achieved = float(counts[3]) / N / (kern_ms / 1000.0) / 1e12
        executed = float(counts[4]) / N / (kern_ms / 1000.0) / 1e12
        roof = {"bound": "valu_fp64", "achieved": achieved, "peak": flops.FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / flops.FP64_VECTOR_PEAK_TFLOPS,
                "traffic": pmc["traffic_bytes"] if pmc else None,
                "traffic_note": (("L2<->fabric bytes per frame (FETCH_SIZE + WRITE_SIZE, KiB units) from the rocprofv3 --pmc passes of "
                                  "profiles/%s (git %s), not collected by this run; algorithmic bytes are the %d-byte image: the rest is the "
                                  "per-lane frame stack and register spills in scratch") % (pmc["file"], pmc["git_rev"], width * height * 3))
                if pmc else "no PMC record for this configuration",
                "valu_issue_frac": pmc["valu_issue_frac"] if pmc else None,
                "valu_busy_frac": pmc["valu_busy_frac"] if pmc else None,
                "resident_waves": pmc["resident_waves"] if pmc else None,
                "lane_util_pmc": pmc["lane_util"] if pmc else None,
                "hbm_gbps": pmc["hbm_gbps"] if pmc else None,
                "spilled_vgprs": pmc["spilled_vgprs"] if pmc else None, "spilled_sgprs": pmc["spilled_sgprs"] if pmc else None,
                "pmc_source": ({"file": "profiles/" + pmc["file"], "git_rev": pmc["git_rev"], "kernel_ms_in_pass": pmc["kernel_ms"]} if pmc else None),
                "kernel": "trace_shade_kernel", "kernel_ms": kern_ms,
                "algorithmic_flops_per_frame": float(counts[3]), "executed_f64_flops_per_frame": float(counts[4]),
                "node_visits": {"reference_defined": int(counts[7]), "walked_by_the_timed_kernel": int(counts[9]),
                                "f32_pre_tests": int(counts[5]), "exact_f64_tests": int(counts[6])},
                "executed": {"f64_tflops": executed, "frac_of_peak": executed / flops.FP64_VECTOR_PEAK_TFLOPS,
                             "frac_of_unfused_ceiling": executed / flops.FP64_UNFUSED_PEAK_TFLOPS}}
        rays_traced = rays_frame - int(counts[8])
        return {"text": text, "width": width, "height": height, "desc": desc, "rays_frame": rays_frame, "elapsed": elapsed,
                "value": rays_frame * steps / elapsed / 1e6, "ms_per_step": elapsed / steps * 1000.0,
                "traced_mrays_per_s": rays_traced * steps / elapsed / 1e6, "rays_traced": rays_traced,
                "rays": {"primary": int(counts[0]), "shadow": int(counts[1]), "secondary": int(counts[2])}, "roofline": roof,
                "one_ms": one_ms, "kernel_region": kernel_region, "host_ms": host_ms}

    m = measure(args.workload, args.width, args.height, args.steps, args.warmup)
    out = None
    if rank == 0:
        gather = ("row-interleaved 8-row bands x%d + one ncclAllGather per frame inside the product (rtx_render_gathered, the SAME entry "
                  "point at every N; %s), K frames queued back to back with up to three in flight on alternating render lanes: value / "
                  "ms_per_step = that streamed rate at every N, ms_per_frame_one_at_a_time = one frame queued, waited for, repeated"
                  % (N, "RCCL communicator" if comm_kind == "rccl" else "LOOPBACK communicator, RCCL did not come up: %s" % comm_error)) if comm is not None else \
                 ("row-interleaved 8-row bands x%d, FALLBACK gather through torch.distributed (the product's communicator did not "
                  "come up: %s)" % (N, comm_error) if torch_gather else
                  "row-interleaved 8-row bands x%d, gloo REHEARSAL gather through host memory" % N if N > 1 else "1 GPU")
        m["roofline"]["note"] = ("achieved/frac are ALGORITHMIC (SURVEY.md 8d): reference-defined node visits and shading events (counted by a "
                                 "pass in the reference-faithful mode: is_shadowed visits everything, as world.rs:196-206 does) x the survey's "
                                 "flop constants / kernel time; the conservative pre-tests answer most node visits without the f64 arithmetic, so "
                                 "this figure is not bounded by the ALU peak (it is the flop-equivalent of Mrays/s).  `executed` is the f64 work "
                                 "really issued; valu_issue_frac = SQ_INSTS_VALU / (kernel cycles x 1024 SIMDs / 4) is the ALU-issue utilisation.  "
                                 "No dense contraction, so no MFMA; kernel_ms = slowest rank's launch time from HIP events on the launch stream "
                                 "(includes the tile-order probe launch when used)")
        out = {"metric": "Mrays/sec (primary+shadow+secondary)", "value": m["value"], "unit": "Mrays/s", "n_gpus": N,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s: %s" % (args.workload, m["desc"]), "width": m["width"], "height": m["height"], "max_depth": 5,
                          "rays_per_frame": m["rays_frame"], "rays": m["rays"], "parallelism": gather},
               "traced_rays": {"rays_per_frame": m["rays_traced"], "mrays_per_s": m["traced_mrays_per_s"],
                               "note": "rays the kernel really traces: shadow rays whose light is behind the surface (material.rs:67: the answer is "
                                       "unused) are counted as the reference-defined rays they are but not traced (rtx_stats.rays_shadow_skipped)"},
               "roofline": m["roofline"]}
        if comm is not None:
            out["ms_per_frame_one_at_a_time"] = m["one_ms"]      # ms_per_step is the streamed rate (up to 3 frames in flight)
        if m["kernel_region"] is not None:
            out["kernel_region"] = m["kernel_region"]
        if m["host_ms"] is not None:
            out["host_render_ms"] = m["host_ms"]
    if N == 1 and not args.no_secondary and args.workload == "S4" and not args.width and not args.height:
        s3 = measure("S3", 0, 0, args.steps, args.warmup)
        if rank == 0:
            out["secondary"] = {"config": {"workload": "S3: %s" % s3["desc"], "width": s3["width"], "height": s3["height"],
                                           "rays_per_frame": s3["rays_frame"], "rays": s3["rays"]},
                                "value": s3["value"], "unit": "Mrays/s", "ms_per_step": s3["ms_per_step"], "steps": args.steps,
                                "ms_per_frame_one_at_a_time": s3["one_ms"], "kernel_region": s3["kernel_region"],
                                "host_render_ms": s3["host_ms"], "roofline": s3["roofline"]}
    if N == 1 and not args.no_secondary and args.workload == "S4" and not args.width and not args.height and not args.kernel_only:
        # C5 of BASELINE.json (the 8K CSG scene, defined as the 8-GPU configuration) on this one GPU, same way, fewer steps
        s5 = measure("S5", 0, 0, max(4, args.steps // 2), 2)
        if rank == 0:
            out["secondary_c5"] = {"config": {"workload": "S5: %s" % s5["desc"], "width": s5["width"], "height": s5["height"],
                                              "rays_per_frame": s5["rays_frame"], "rays": s5["rays"]},
                                   "value": s5["value"], "unit": "Mrays/s", "ms_per_step": s5["ms_per_step"], "steps": max(4, args.steps // 2),
                                   "ms_per_frame_one_at_a_time": s5["one_ms"], "kernel_region": s5["kernel_region"],
                                   "host_render_ms": s5["host_ms"], "roofline": s5["roofline"]}
    if rank == 0:
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m["text"], m["width"], m["height"])
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if comm is not None:
        comm.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
