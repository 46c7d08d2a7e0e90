#!/usr/bin/env python3
"""Renders a workload at depth budgets 0, 1, 2, 5 and prints time, rays and (diagnostic builds) the section timers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx, scenes
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "S4"
sn, w, h, _ = WORKLOADS[name]
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
sc = ctx.scene(world)
for depth in (0, 1, 2, 5):
    for _ in range(3):
        r = sc.render(cams[0], depth, want_hit=False)
    c = sc.render(cams[0], depth, want_hit=False, flags=rtx.F_COUNT_EVENTS)["stats"].events()
    st = r["stats"]
    t = [int(st.ev[i]) for i in range(18, 24)]
    wi = c["lane_slots"] / 64
    print(depth, st.rays, round(st.kernel_ms, 3), "wave-iters %d" % wi, "packing %.3f" % (st.rays / c["lane_slots"]),
          "cycles/wave-iter:", dict(zip(["cullORsetup", "exactORchildren", "shadeORunwind", "otherORtrav", "prep", "light"], [round(x / wi) for x in t])), flush=True)
