#!/usr/bin/env python3
"""Wall clock of rtx_scene_create (flattened world -> compiled traversal program + upload) and of the first render after it,
per workload: what a caller pays once per scene.  usage: tools/create_latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
from bench import WORKLOADS
ctx = rtx.Context(0)
for name in ("S1", "S2", "S3", "S4", "S5"):
    sn, w, h = (name, 400, 200) if name == "S1" else WORKLOADS[name][:3]
    t0 = time.perf_counter(); world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h)); t1 = time.perf_counter()
    best = 1e9
    for _ in range(3):
        a = time.perf_counter(); sc = ctx.scene(world); b = time.perf_counter(); best = min(best, b - a)
        sc.close()
    print("%s: load_yaml %.1f ms, rtx_scene_create %.2f ms (best of 3)" % (name, (t1 - t0) * 1e3, best * 1e3), flush=True)
