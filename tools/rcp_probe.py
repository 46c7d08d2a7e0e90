import os, sys
sys.path[:0] = [os.path.join(os.getcwd(), "ray-tracer_amd", "python")]
import numpy as np, rtx
ctx = rtx.Context(0)
rng = np.random.default_rng(3)
x = np.concatenate([rng.uniform(1, 2, 2000000), 10.0 ** rng.uniform(-300, 300, 1000000) * rng.choice([-1.0, 1.0], 1000000)])
r = ctx.debug_math(8, x)
rel = np.abs(r * x - 1.0)
print("max rel err", rel.max(), "log2", np.log2(rel.max()))
