#!/usr/bin/env python3
"""Timeline of one launch from the RTX_WAVE_TRACE build (make -C ray-tracer_amd librtx_xA.so
XFLAGS_A=-DRTX_WAVE_TRACE; RTX_LIB=librtx_xA.so): every wave's start / end on the 100 MHz clock.
usage: wave_trace.py SCENE W H N [split]   (renders rank 0's bands of an N-way split)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import numpy as np
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx, scenes
name, w, h, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["RTX_SPLIT"] = sys.argv[5] if len(sys.argv) > 5 else "0"
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make(name, width=w, height=h))
sc = ctx.scene(world)
for _ in range(3):
    r = sc.render(cams[0], 5, bands=(0, N), flags=rtx.F_BANDS_COMPACT)
hit = r["hit"].ravel()
nb = hit.size // 3
t = hit[:3 * nb].reshape(nb, 3).astype(np.int64)
t = t[(t[:, 0] != 0) & (t[:, 1] != 0) & (t[:, 0] != -2)]
start = (t[:, 0] & 0xffffffff); end = (t[:, 1] & 0xffffffff)
t0 = start.min()
start = (start - t0) / 100.0; end = (end - t0) / 100.0      # microseconds
dur = end - start
iters = t[:, 2] & 0xffff; split = (t[:, 2] >> 16) & 1; cls = (t[:, 2] >> 17) & 7
print("%s %dx%d N=%d split=%s: kernel_ms %.3f, %d waves (%d split), last end %.0f us" %
      (name, w, h, N, os.environ["RTX_SPLIT"], r["stats"].kernel_ms, len(dur), int(split.sum()), end.max()))
print("wave duration us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; iterations of slowest lane: mean %.1f max %d; us/iter %.1f" %
      (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), iters.mean(), iters.max(), dur.sum() / max(iters.sum(), 1)))
for k in range(7, -1, -1):
    m = cls == k
    if m.any():
        print("  class %d: %6d waves, iters mean %.1f p90 %.0f max %d, dur mean %.0f max %.0f us, share of wave-time %.1f %%" %
              (k, int(m.sum()), iters[m].mean(), np.percentile(iters[m], 90), iters[m].max(), dur[m].mean(), dur[m].max(), 100 * dur[m].sum() / dur.sum()))
order = np.argsort(-dur)[:5]
for i in order:
    print("  slow wave: start %.0f end %.0f dur %.0f iters %d split %d class %d" % (start[i], end[i], dur[i], iters[i], split[i], cls[i]))
edges = np.linspace(0, end.max(), 21)
occ = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("waves alive per 5 % slice of the launch:", occ)
print("wave-time integral %.0f slot-us = %.1f %% of 4096 slots x launch" % (dur.sum(), 100 * dur.sum() / (4096 * end.max())))
