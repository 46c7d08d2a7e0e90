#!/usr/bin/env python3
"""Split tiles (context options "split" / "split_class"): kernel ms of one rank's share of the 4K frame
for N = 1, 2, 4, 8 and of smaller whole frames, unsplit against splitting classes >= 7, 6, 5, 4."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import rtx, scenes
ctx = rtx.Context(0)
ctx.set_option("host_sets", 1)          # one launch per render: this probe is about the launch itself
CLS = os.environ.get("PROBE_CLASSES", "7,6,5,4").split(",")     # split classes >= c


def best(sc, cam, bands, reps=5):
    return min(sc.render(cam, 5, bands=bands, flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"].kernel_ms for _ in range(reps))


def row(sc, cam, bands):
    ctx.set_option("split", 0)
    out = ["unsplit %.3f" % best(sc, cam, bands)]
    ctx.set_option("split", 1)
    for c in CLS:
        ctx.set_option("split_class", int(c))
        out.append("%s %.3f" % (c, best(sc, cam, bands)))
    return "  ".join(out)


name = sys.argv[1] if len(sys.argv) > 1 else "S4"
world, cams = rtx.load_yaml(scenes.make(name, width=3840, height=2160))
sc = ctx.scene(world)
for N in (2, 4, 8):
    print("%s 4K N=%d rank0 kernel ms: %s" % (name, N, row(sc, cams[0], (0, N))), flush=True)
sc.close()
for w, h in ((1920, 1080), (1280, 720)):
    world, cams = rtx.load_yaml(scenes.make(name, width=w, height=h))
    sc = ctx.scene(world)
    print("%s %dx%d kernel ms: %s" % (name, w, h, row(sc, cams[0], (0, 1))), flush=True)
    sc.close()
