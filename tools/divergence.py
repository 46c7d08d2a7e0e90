#!/usr/bin/env python3
"""Lane occupancy of the divergent parts of a wave-iteration: needs the diagnostic build
`make -C ray-tracer_amd librtx_hip_div.so` (-DRTX_DIAG_DIV: the event slots then carry these figures instead of the
reference-defined events).  usage: RTX_LIB=librtx_hip_div.so python3 tools/divergence.py [S3 S4 S5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")
os.environ.setdefault("RTX_LIB", "librtx_hip_div.so")
import rtx, scenes
from bench import WORKLOADS

ctx = rtx.Context(0)
for name in sys.argv[1:] or ["S4"]:
    sn, w, h, _ = WORKLOADS[name]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    st = sc.render(cams[0], 5, flags=rtx.F_COUNT_EVENTS, want_hit=False)["stats"]
    e = [int(x) for x in st.ev]
    def frac(a, b):
        return "%6.1f %%" % (100.0 * a / b) if b else "   n/a"
    iters = e[0] / 64.0
    print("%s %dx%d: %d rays, %.0f wave-iterations" % (name, w, h, st.rays, iters))
    print("  lanes tracing a ray per iteration (lane packing)         %s" % frac(e[1], e[0]))
    print("  exact shape tests, per chunk: (lane, candidate) pairs %d, 64 x longest lane %d" % (e[2], e[3]))
    print("    lanes busy if the loop ran to its longest lane          %s" % frac(e[2], e[3]))
    print("    lanes busy in the rounds really run (early-outs)        %s   (%.2f rounds per iteration, %.2f pairs per traced ray)"
          % (frac(e[5], e[4]), e[4] / 64.0 / iters, e[2] / max(e[1], 1)))
    print("    rounds if the pairs were spread evenly over the lanes   %.2f per iteration (lower bound: pairs / 64)" % (e[2] / 64.0 / iters))
    print("  radiance iterations %5.1f %% of all; lanes at a radiance ray in them   %s" % (100.0 * e[6] / e[0], frac(e[7], e[6])))
    print("    PreparedComputations block: lanes with a hit             %s" % frac(e[9], e[8]))
    print("  lighting block entered in %5.1f %% of iterations; lanes in it           %s" % (100.0 * e[10] / e[0], frac(e[11], e[10])))
    print("    pattern look-up (non-plain material)                     %s   (entered in %.1f %% of lighting blocks)" % (frac(e[13], e[12]), 100.0 * e[12] / max(e[10], 1)))
    print("    lit branch (diffuse + specular + pow)                    %s   (entered in %.1f %%)" % (frac(e[15], e[14]), 100.0 * e[14] / max(e[10], 1)))
    print("  group visits per wave-iteration %.2f, of which run the transform + bounds filter %.2f" % (e[16] / 64.0 / iters, e[18] / 64.0 / iters))
    sc.close()
