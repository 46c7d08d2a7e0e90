#!/usr/bin/env python3
"""Kernel time of one rank's share (bands r, r+N, ...) of a frame for N = 1, 2, 4, 8 — what each GPU of an N-GPU run
computes (the all-gather is not included).  usage: band_probe.py [S3 S4 S5 ...]; min of 5 renders for ranks 0 and 1."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx, scenes
from bench import WORKLOADS
ctx = rtx.Context(0)
for name in sys.argv[1:] or ["S4"]:
    sn, w, h, _ = WORKLOADS[name]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    base = None
    for N in (1, 2, 4, 8):
        ms = []
        for r in range(min(N, 2)):
            ms.append(min(sc.render(cams[0], 5, bands=(r, N), flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"].kernel_ms for _ in range(5)))
        base = base or ms[0]
        print("%s %dx%d N=%d rank0/1 kernel ms: %s  speed-up of the slower rank vs N=1: %.2fx" %
              (name, w, h, N, [round(m, 3) for m in ms], base / max(ms)), flush=True)
    sc.close()
