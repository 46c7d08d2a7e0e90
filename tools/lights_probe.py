#!/usr/bin/env python3
"""Kernel ms of the glass / mirror scene (S4, 1920x1080) with 1, 2 and 3 lights: how the scheduling of ray kinds behaves
when a hit is followed by several shadow rays.  usage: [RTX_LIB=...] tools/lights_probe.py"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")
import rtx, scenes
ctx = rtx.Context(0)
base = scenes.make("S4", width=1920, height=1080)
extra = ["- add: light\n  at: [%.1f, %.1f, %.1f]\n  intensity: [0.3, 0.3, 0.3]\n\n" % p for p in ((9, 11, -8), (0, 14, 3), (-5, 9, 6), (6, 8, 7), (-9, 6, -3), (3, 13, -11), (1, 5, -14))]
for n in [int(x) for x in os.environ.get('LIGHTS', '1,2,3').split(',')]:
    text = base.replace("- add: light", "".join(extra[:n - 1]) + "- add: light", 1)
    assert text.count("- add: light") == n
    world, cams = rtx.load_yaml(text)
    sc = ctx.scene(world)
    ms = []
    for _ in range(7):
        r = sc.render(cams[0], 5, want_hit=False)
        ms.append(r["stats"].kernel_ms)
    print("%-16s S4 1080p %d light(s): min %.3f ms  rays %d  sha %s" % (os.environ.get("RTX_LIB", "librtx_hip.so"), n, min(ms), r["stats"].rays,
          hashlib.sha1(r["rgb"].tobytes()).hexdigest()[:10]), flush=True)
    sc.close()
