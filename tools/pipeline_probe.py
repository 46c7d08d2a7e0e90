#!/usr/bin/env python3
"""Throughput of ONE rank's share of a frame at N ranks, frame after frame, with and without the alternating render
lanes of rtx_render_gathered (rtx_set_option "pipeline"): a loopback communicator restricted to rank 0's share
(option debug = 3), K frames queued back to back, wall clock between two host waits.
usage: tools/pipeline_probe.py [S4 S3 S5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import ctypes as C
import numpy as np
import rtx, scenes
from bench import WORKLOADS

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
ctx = rtx.Context(0)
K = 40
NS = [int(x) for x in os.environ.get("PROBE_N", "1,2,4,8").split(",")]
PIPES = [int(x) for x in os.environ.get("PROBE_PIPE", "1,2,3,4").split(",")]
for name in sys.argv[1:] or ["S4", "S3", "S5"]:
    sn, w, h, _ = WORKLOADS[name]
    world, cams = rtx.load_yaml(scenes.make(sn, width=w, height=h))
    sc = ctx.scene(world)
    frame = C.c_void_p(); assert hip.hipMalloc(C.byref(frame), w * h * 3) == 0
    for n in NS:
        line = "%s %dx%d one rank's share at N=%d, ms per frame over %d frames:" % (name, w, h, n, K)
        for pipe in PIPES:
            ctx.set_option("pipeline", pipe)
            ctx.set_option("debug", 3)
            comm = rtx.Comm.loopback(ctx, n)
            for _ in range(4):
                comm.render_gathered(sc, cams[0], frame.value)
            comm.wait()
            t0 = time.perf_counter()
            for _ in range(K):
                comm.render_gathered(sc, cams[0], frame.value)
            comm.wait()
            dt = (time.perf_counter() - t0) / K * 1e3
            conc = rtx.LIB.rtx_comm_concurrency(comm.h)
            comm.close()
            line += "  %d in flight %.3f" % (pipe, dt) + " (%d concurrent)" % conc
        ctx.set_option("debug", 0)
        print(line, flush=True)
    sc.close()
