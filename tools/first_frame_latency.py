#!/usr/bin/env python3
"""What a process that renders ONE frame pays, step by step (wall clock).  usage: tools/first_frame_latency.py [S3|S4|S5]"""
import os, sys, time
t00 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
import numpy as np
t0 = time.perf_counter()
ctx = rtx.Context(0)
t1 = time.perf_counter()
name = sys.argv[1] if len(sys.argv) > 1 else "S4"
from bench import WORKLOADS
text = scenes.make(WORKLOADS[name][0], width=WORKLOADS[name][1], height=WORKLOADS[name][2])
t2 = time.perf_counter()
world, cams = rtx.load_yaml(text)
t3 = time.perf_counter()
sc = ctx.scene(world)
t4 = time.perf_counter()
r = sc.render(cams[0], 5, want_hit=False)
t5 = time.perf_counter()
r = sc.render(cams[0], 5, want_hit=False)
t6 = time.perf_counter()
rtx.write_png("/tmp/x.png", r["rgb"])
t7 = time.perf_counter()
print(name, "import %.1f  rtx_init %.1f  load_yaml %.1f  scene_create %.1f  first render %.1f  second render %.1f  png %.1f ms" %
      ((t0-t00)*1e3, (t1-t0)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t6-t5)*1e3, (t7-t6)*1e3))
