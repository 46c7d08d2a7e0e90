"""Wall-clock of n frames of one resident scene: n rtx_render calls (one wait + copy each) against one
rtx_render_batch call.  Cameras orbit the scene (an animation sweep).
usage: python tools/batch_probe.py [S1|S4] [width height frames]"""
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "ray-tracer_amd", "python")]
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx      # noqa: E402
import scenes   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S1"
w, h, n = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (400, 200, 32)
world, cams = rtx.load_yaml(scenes.make(name, width=w, height=h))
base = cams[0]
sweep = []
for k in range(n):
    a = 2 * math.pi * k / n * 0.25
    c = rtx.API.camera(w, h, math.pi / 3, rtx.API.view_transform([12 * math.sin(a), 3, -12 * math.cos(a)], [0, 1, 0], [0, 1, 0]))
    sweep.append(c)
ctx = rtx.Context(0)
sc = ctx.scene(world)
sc.render_batch(sweep, 5, want_hit=False, want_stats=False)           # warm-up (allocations, code load)
for c in sweep[:2]:
    sc.render(c, 5, want_hit=False)
t0 = time.perf_counter()
for c in sweep:
    sc.render(c, 5, want_hit=False)
t1 = time.perf_counter()
sc.render_batch(sweep, 5, want_hit=False, want_stats=False)
t2 = time.perf_counter()
r = sc.render_batch(sweep, 5, want_hit=False, want_stats=True)
kern = sum(x["stats"].kernel_ms for x in r)
print("%s %dx%d x %d frames: single calls %.2f ms, one batch %.2f ms (kernels %.2f ms) -> %.2fx"
      % (name, w, h, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, kern, (t1 - t0) / (t2 - t1)))
