#!/usr/bin/env python3
"""Wall clock of the host-buffer entry point rtx_render (what a drop-in caller pays per frame: launches + D2H copy of the
image into the caller's pageable buffer) beside the kernel time of the same call.  usage: tools/host_latency.py [S4 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
from bench import WORKLOADS

ctx = rtx.Context(0)
for name in sys.argv[1:] or ["S2", "S3", "S4", "S5"]:
    sn, w, h, _ = WORKLOADS[name]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    for want_hit, sets in [(False, int(x)) for x in os.environ.get('SETS', '1,0').split(',')] + [(True, 1), (True, 0)]:
        ctx.set_option("host_sets", sets)
        best = (1e9, 0)
        for _ in range(6):
            t0 = time.perf_counter()
            r = sc.render(cams[0], 5, want_hit=want_hit)
            dt = (time.perf_counter() - t0) * 1e3
            if dt < best[0]:
                best = (dt, r["stats"].kernel_ms)
        print("%s %dx%d rtx_render%s host_sets=%d: wall %.2f ms, kernel %.2f ms, image %.1f MB" % (
            name, w, h, " + hit ids" if want_hit else "", sets, best[0], best[1], w * h * (7 if want_hit else 3) / 1e6), flush=True)
    sc.close()
