#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + shadow + secondary) of the MI355X trace-and-shade path.

A "step" is one frame: every rank renders its row-interleaved 8-row tile bands of the SAME frame
(rank r gets bands r, r+N, ...) with the scene resident in HBM, then the bands are all-gathered
over RCCL (N > 1).  Total work per step is fixed as N grows -> "strong" scaling; value = whole-job
rays per second.  Launched by the driver as `python bench.py --gpus 1 ...` or, for N > 1, through
torch.distributed.run (one process per GPU).

Extra objects on the JSON line: `roofline` (FP64 vector ALU bound: algorithmic flops of the frame
from the kernel's own event counters / the kernel's measured launch time) and `cpu_baseline` (the
CPU oracle — a C++ restatement of the single-threaded reference, `kind: "port"` — timed on a
bounded tile sample of the same frame on this box's host cores, rank 0, N = 1 only).
"""
import argparse
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="S4", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import bands as bandlib
    import flops
    import rtx
    import scenes

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (e.g. 2 ranks, gloo, one GPU):
    # RTX_BENCH_BACKEND=gloo maps every rank onto the GPUs that exist.  The driver's runs never set it.
    backend = os.environ.get("RTX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    N = args.gpus
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
        N = world_size
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace-and-shade path has no CPU fallback")

    # collectives: RCCL on device tensors; the gloo rehearsal stages through host memory
    def all_reduce(t, op=None):
        if backend == "nccl":
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
        else:
            c = t.cpu(); dist.all_reduce(c, op=op or dist.ReduceOp.SUM); t.copy_(c)

    def all_gather(out, inp):
        if backend == "nccl":
            dist.all_gather_into_tensor(out, inp)
        else:
            parts = [torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(world_size)]
            dist.all_gather(parts, inp.cpu()); out.copy_(torch.cat(parts))

    class _Done:                     # stand-in for a finished collective (the gloo rehearsal gathers synchronously)
        def wait(self):
            return True

    def all_gather_async(out, inp):
        if backend == "nccl":
            return dist.all_gather_into_tensor(out, inp, async_op=True)
        all_gather(out, inp)
        return _Done()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene_name, width, height, desc = WORKLOADS[args.workload]
    width, height = args.width or width, args.height or height
    text = scenes.make(scene_name, width=width, height=height)
    world, cams = rtx.load_yaml(text)
    cam = cams[0]
    ctx = rtx.Context(local_rank)
    scene = ctx.scene(world)                      # scene resident in HBM from here on

    bands = (rank, N)
    rows_local = rtx.LIB.rtx_band_rows(height, rtx.Bands(*bands))
    rows_pad = bandlib.rows_padded(height, N)                  # equal send counts for the all-gather
    assert rows_local <= rows_pad
    local = torch.zeros(rows_pad * width * 3, dtype=torch.uint8, device=dev)
    gathered = torch.zeros(N * rows_pad * width * 3, dtype=torch.uint8, device=dev) if N > 1 else None
    # a real (non-null) stream for everything below: the kernel launches, the timing events and the streams RCCL
    # synchronises with are then one and the same queue (the null stream would hand the C ABI its own stream)
    compute = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(compute)
    stream = compute.cuda_stream

    def step(stats=False):
        st = scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, stats)
        if N > 1:
            all_gather(gathered, local)
        return st

    # exact ray and event counts of this rank's share (untimed)
    st = scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT | rtx.F_COUNT_EVENTS, stream, True)
    rays_local = st.rays
    evs = st.events()
    fl_local = flops.frame_flops(evs, st.rays_primary, st.rays_shadow, len_lights(text))
    fl_exec = flops.executed_flops(evs, st.rays_primary, st.rays_shadow, len_lights(text))
    counts = torch.tensor([st.rays_primary, st.rays_shadow, st.rays_secondary, fl_local, fl_exec,
                           evs["cull"], evs["x_xform"], evs["xform"]], dtype=torch.float64, device=dev)
    if N > 1:
        all_reduce(counts)
    rays_frame = int(counts[0] + counts[1] + counts[2])

    for _ in range(args.warmup):
        step()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # N > 1 over RCCL: frame k's all-gather runs on RCCL's stream while frame k+1 renders on the compute stream
    # (two band buffers, two frame buffers); a buffer is reused only after its own gather has been waited for.
    overlap = N > 1 and os.environ.get("RTX_BENCH_OVERLAP", "1") != "0"
    if overlap:
        locals_ = [local, torch.zeros_like(local)]
        frames_ = [gathered, torch.zeros_like(gathered)]
        pending = [None, None]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if overlap:
            b = k & 1
            if pending[b] is not None:
                pending[b].wait()                 # stream-level: the compute stream waits for that gather
            ev[k][0].record()
            scene.render_device(cam, locals_[b].data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
            ev[k][1].record()
            pending[b] = all_gather_async(frames_[b], locals_[b])
            continue
        ev[k][0].record()
        scene.render_device(cam, local.data_ptr(), 0, 5, bands, rtx.F_BANDS_COMPACT, stream, False)
        ev[k][1].record()
        if N > 1:
            all_gather(gathered, local)
    if overlap:
        for w in pending:
            if w is not None:
                w.wait()
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    kern_ms = torch.tensor([sum(a.elapsed_time(b) for a, b in ev) / args.steps], dtype=torch.float64, device=dev)
    if N > 1:
        all_reduce(elapsed, dist.ReduceOp.MAX)
        all_reduce(kern_ms, dist.ReduceOp.MAX)
    elapsed = float(elapsed[0]); kern_ms = float(kern_ms[0])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1000.0
        # per launch (= per GPU): each rank's launch does 1/N of the frame's flops in kern_ms (slowest rank)
        achieved = float(counts[3]) / N / (kern_ms / 1000.0) / 1e12
        executed = float(counts[4]) / N / (kern_ms / 1000.0) / 1e12
        traffic = flops.measured_traffic_bytes(args.workload) if (N == 1 and not args.width and not args.height) else None
        out = {
            "metric": "Mrays/sec (primary+shadow+secondary)", "value": rays_frame * args.steps / elapsed / 1e6,
            "unit": "Mrays/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "width": width, "height": height,
                       "max_depth": 5, "rays_per_frame": rays_frame,
                       "rays": {"primary": int(counts[0]), "shadow": int(counts[1]), "secondary": int(counts[2])},
                       "parallelism": ("row-interleaved 8-row bands x%d + RCCL all-gather%s" % (N, " overlapped with the next frame" if overlap else "")) if N > 1 else "1 GPU"},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": flops.FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / flops.FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_note": ("L2<->fabric bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB units) from the committed "
                                          "rocprofv3 --pmc passes profiles/r1f_S4_pmc.txt / r1e_*_pmc.txt of this workload, not collected by this run; "
                                          "upper bound with FETCH doubled (gfx950 64-B tally): %d; algorithmic bytes are the %d-byte "
                                          "image: the rest is the per-lane frame stack and register spills in scratch"
                                          % (traffic[1], width * height * 3)) if traffic else "not measured for this configuration",
                         "kernel": "trace_shade_kernel", "kernel_ms": kern_ms,
                         "algorithmic_flops_per_frame": float(counts[3]),
                         "executed_f64_flops_per_frame": float(counts[4]),
                         "node_visits": {"reference_defined": int(counts[7]), "f32_pre_tests": int(counts[5]),
                                         "exact_f64_tests": int(counts[6])},
                         "valu_busy_pmc": flops.MEASURED_VALU_BUSY.get(args.workload) if traffic else None,
                         "executed": {"f64_tflops": executed, "frac_of_peak": executed / flops.FP64_VECTOR_PEAK_TFLOPS,
                                      "frac_of_unfused_ceiling": executed / flops.FP64_UNFUSED_PEAK_TFLOPS},
                         "note": "achieved/frac are ALGORITHMIC (SURVEY.md 8d): reference-defined node visits and shading events x "
                                 "the survey's flop constants / kernel time; the conservative pre-tests (f32 bounding spheres, "
                                 "interval filter) answer most node visits without the f64 arithmetic, so this figure is not bounded "
                                 "by the ALU peak (it is the flop-equivalent of Mrays/s).  `executed` is the f64 work really issued "
                                 "against the same peak.  No dense contraction, so no MFMA; kernel_ms = slowest rank's average launch "
                                 "time from HIP events on the launch stream (includes the tile-order probe launch when used)"},
        }
        if N == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(text, width, height)
        print(json.dumps(out), flush=True)
    scene.close(); ctx.close()
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


def len_lights(text):
    return text.count("- add: light")


if __name__ == "__main__":
    main()
