#!/usr/bin/env python3
"""Kernel ms of one rank's share (N = 1, 8) against the depth budget of the tile-order probe (option "probe_depth") and without
the probe.  usage: tools/probe_depth.py [S4]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")
import rtx, scenes
ctx = rtx.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "S4"
world, cams = rtx.load_yaml(scenes.make(name, width=3840, height=2160))
sc = ctx.scene(world)


def best(bands, reps=6):
    return min(sc.render(cams[0], 5, bands=bands, flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"].kernel_ms for _ in range(reps))


for N in (1, 4, 8):
    out = []
    for pd in (1, 2, 3, 4):
        ctx.set_option("probe_depth", pd)
        out.append("depth %d: %.3f" % (pd, best((0, N))))
    ctx.set_option("probe_depth", 2)
    ctx.set_option("lpt", 0); out.append("no probe: %.3f" % best((0, N))); ctx.set_option("lpt", -1)
    print("%s 4K N=%d kernel ms  " % (name, N) + "  ".join(out), flush=True)
