#!/usr/bin/env python3
"""Per-pixel traced-ray counts at several depth budgets from the RTX_PIXEL_COST build (make -C ray-tracer_amd librtx_xA.so
XFLAGS_A=-DRTX_PIXEL_COST; RTX_LIB=librtx_xA.so), saved as gpurun_out/order/cost_<scene>_d<k>.npy (uint8) for offline studies
of tile-cost estimators (tools/order_sim.py).  usage: tools/pixel_cost.py [SCENE [W H]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")
os.environ["RTX_LPT"] = "0"; os.environ["RTX_SPLIT"] = "0"
import numpy as np
import rtx, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "S4"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
OUT = os.path.join(ROOT, "gpurun_out", "order"); os.makedirs(OUT, exist_ok=True)
ctx = rtx.Context(0)
world, cams = rtx.load_yaml(scenes.make(name, width=W, height=H))
sc = ctx.scene(world)
for d in (5, 1, 2, 3):
    r = sc.render(cams[0], d)
    c = r["hit"].reshape(H, W)
    print("depth %d: rays per pixel mean %.2f max %d" % (d, c.mean(), c.max()), flush=True)
    np.save(os.path.join(OUT, "cost_%s_d%d.npy" % (name, d)), np.minimum(c, 255).astype(np.uint8))
