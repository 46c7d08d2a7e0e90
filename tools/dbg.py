import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import rtx, ora, parity
ctx = rtx.Context(0)
text = parity.scene_text(sys.argv[1] if len(sys.argv) > 1 else "S1", 160, 80)
(pw, pc), (ow, oc) = parity.build_both(text)
cpu = ora.render(ow, oc, 5, threads=8)
for fl in (0, rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT):
    g = ctx.render(pw, pc, 5, flags=fl)
    st = g["stats"]
    print("flags", fl, "rays", st.rays_primary, st.rays_shadow, st.rays_secondary, "cpu", cpu["stats"].rays_shadow, cpu["stats"].rays_secondary)
    print(" hit hist gpu", np.unique(g["hit"], return_counts=True))
    print(" hit hist cpu", np.unique(cpu["hit"], return_counts=True))
    print(" ", parity.compare(g, cpu))
    print(" ev", st.events())
