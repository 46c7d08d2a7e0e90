#!/usr/bin/env python3
"""Back-to-back frames of one rank's share (bands 0, N, 2N, ...) without per-frame synchronisation: wall time per
frame against the HIP-event kernel time of one frame — what the host-side launch sequence (counter reset, probe,
fork/join of the split launch, frame launch) costs when frames are short."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import torch
import rtx, scenes
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make("S4", width=3840, height=2160))
sc = ctx.scene(world)
cam = cams[0]
for N in (1, 2, 4, 8):
    rows = rtx.LIB.rtx_band_rows(cam.vsize, rtx.Bands(0, N))
    buf = torch.empty(rows * cam.hsize * 3, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            k = sc.render_device(cam, buf.data_ptr(), bands=(0, N), flags=rtx.F_BANDS_COMPACT, stream=st.cuda_stream, stats=True)
        torch.cuda.synchronize()
        K = 50
        t0 = time.perf_counter()
        for _ in range(K):
            sc.render_device(cam, buf.data_ptr(), bands=(0, N), flags=rtx.F_BANDS_COMPACT, stream=st.cuda_stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("N=%d: kernel %.3f ms, wall/frame %.3f ms (host enqueue %.3f ms/frame)" % (N, k.kernel_ms, (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3), flush=True)
