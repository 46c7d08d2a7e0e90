#!/usr/bin/env python3
"""Renders S4 at depth budgets 0, 1, 5 (3 launches each) so PMC passes can attribute traffic to the frame stack."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make("S4", width=3840, height=2160))
sc = ctx.scene(world)
for depth in (0, 1, 5):
    for _ in range(3):
        r = sc.render(cams[0], depth, want_hit=False)
    print(depth, r["stats"].rays, round(r["stats"].kernel_ms, 3), flush=True)
