#!/usr/bin/env python3
"""Prints ray/event counts and lane-packing diagnostics of a workload on the GPU."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTX_HOST_SETS", "1")     # one launch per render: these tools time the launch itself
import rtx, scenes, flops
from bench import WORKLOADS

ctx = rtx.Context(0)
for name in sys.argv[1:] or ["S2", "S3", "S4", "S5"]:
    sn, w, h, _ = WORKLOADS[name]
    text = scenes.make(sn, width=w, height=h)
    world, cams = rtx.load_yaml(text)
    sc = ctx.scene(world)
    import numpy as np
    r = sc.render(cams[0], 5, flags=rtx.F_COUNT_EVENTS, want_hit=False)
    st = r["stats"]; ev = st.events()
    fast = sc.render(cams[0], 5, want_hit=False)["stats"]
    out = dict(workload=name, rays=st.rays, primary=st.rays_primary, shadow=st.rays_shadow, secondary=st.rays_secondary,
               events=ev, lane_packing=st.rays / max(ev["lane_slots"], 1),
               op_packing=ev["xform"] / max(ev["op_slots"], 1),
               flops=flops.frame_flops(ev, st.rays_primary, st.rays_shadow, text.count("- add: light")),
               count_ms=st.kernel_ms, fast_ms=fast.kernel_ms,
               timing=dict(cull=int(fast.ev[18]), exact=int(fast.ev[19]), shade=int(fast.ev[20]), other=int(fast.ev[21]), prep=int(fast.ev[22]), light=int(fast.ev[23])))
    print(json.dumps(out))
    sc.close()
