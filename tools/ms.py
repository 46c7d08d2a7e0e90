#!/usr/bin/env python3
"""Kernel ms (min / median of 9 renders) and an image checksum per workload for the library named by RTX_LIB;
A/B runs compare libraries inside ONE gpurun call (boxes differ by +-5 %).  usage: ms.py [S3 S4 S5 ...] [N=<ranks>]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import numpy as np
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx, scenes
from bench import WORKLOADS
ctx = rtx.Context(0)
N = 1
names = []
for a in sys.argv[1:]:
    if a.startswith("N="):
        N = int(a[2:])
    else:
        names.append(a)
for name in names or ["S3", "S4", "S5"]:
    sn, w, h, _ = WORKLOADS[name]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    ms = []
    for _ in range(9):
        r = sc.render(cams[0], 5, bands=(0, N), flags=rtx.F_BANDS_COMPACT if N > 1 else 0, want_hit=False)
        ms.append(r["stats"].kernel_ms)
    print("%-24s %s N=%d  min %.3f  med %.3f ms  rays %d  sha %s" % (os.environ.get("RTX_LIB", "librtx_hip.so"), name, N, min(ms), float(np.median(ms)),
          r["stats"].rays, hashlib.sha1(r["rgb"].tobytes()).hexdigest()[:10]), flush=True)
    sc.close()
