#!/usr/bin/env python3
"""Lane utilisation of the frame launch: traced rays / (64 x wave-iterations), per workload (product mode, counting build).
usage: tools/lane_util.py [S3 S4 S5 ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")
import rtx, scenes
from bench import WORKLOADS
ctx = rtx.Context(0)
for name in (sys.argv[1:] or ["S2", "S3", "S4", "S5"]):
    sn, w, h = WORKLOADS[name][:3]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    st = sc.render(cams[0], 5, flags=rtx.F_COUNT_EVENTS, want_hit=False)["stats"]
    ev = st.events()
    traced = st.rays - st.rays_shadow_skipped
    print("%s: rays %d (traced %d), wave-iterations %d (%.1f per wave), lane utilisation %.1f %%" %
          (name, st.rays, traced, ev["lane_slots"] // 64, ev["lane_slots"] / 64 / (((w + 7) // 8) * ((h + 7) // 8)), 100.0 * traced / ev["lane_slots"]), flush=True)
    sc.close()
