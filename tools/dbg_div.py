import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "ray-tracer_amd", "python")]
import numpy as np, rtx
ctx = rtx.Context(0)
rng = np.random.default_rng(1)
xa = np.concatenate([rng.uniform(0, 1e6, 500000), 10.0 ** rng.uniform(-320, 308, 300000) * rng.choice([-1.0, 1.0], 300000)])
xb = np.concatenate([rng.uniform(-1e3, 1e3, 500000), 10.0 ** rng.uniform(-320, 308, 300000) * rng.choice([-1.0, 1.0], 300000)])
with np.errstate(all="ignore"):
    want = xa / xb
for op in (1, 7, 8, 9):
    got = ctx.debug_math(op, xa, xb)
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    print(op, int(bad.sum()), "of", len(want))
    for i in np.nonzero(bad)[0][:8]:
        print("   x=%r y=%r got=%r want=%r" % (xa[i], xb[i], got[i], want[i]))
