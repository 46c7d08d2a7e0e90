#!/usr/bin/env python3
"""Renders one rank's share (bands r, r+N, ...) of the S4 4K frame repeatedly: the program to put under
rocprofv3 --kernel-trace to see the probe, SPLIT and main launches of a short dispatch.
usage: rank_share.py [N] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make("S4", width=3840, height=2160))
sc = ctx.scene(world)
ms = [sc.render(cams[0], 5, bands=(0, N), flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"].kernel_ms for _ in range(frames)]
print("N=%d rank 0: kernel ms min %.3f median %.3f" % (N, min(ms), sorted(ms)[len(ms) // 2]))
