#!/usr/bin/env python3
"""Kernel time of one rank's share (bands r, r+N, ...) of the S4 frame for N = 1, 2, 4, 8 — what each GPU of an
N-GPU run computes (the all-gather is not included)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make("S4", width=3840, height=2160))
sc = ctx.scene(world)
for N in (1, 2, 4, 8):
    ms = []
    for r in range(min(N, 2)):
        for _ in range(3):
            st = sc.render(cams[0], 5, bands=(r, N), flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"]
        ms.append(round(st.kernel_ms, 3))
    print("N=%d rank0/1 kernel ms:" % N, ms, flush=True)
