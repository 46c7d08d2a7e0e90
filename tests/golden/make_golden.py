#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle (oracle/, pinned by the
reference's own #[test] values in tests/test_oracle_kat.py).  The reference itself cannot be run here
(Rust, no toolchain) and ships no rendered image, so these are oracle outputs, not reference outputs.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Each fixture: scene text (yaml), rgb (H,W,3) u8, hit (H,W) i32 primary ShapeIndex or -1, tie (H,W) u8
EPSILON-tie flags, rays = [primary, shadow, secondary].
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

CASES = [("S1", 100, 50, {}), ("S3", 100, 56, dict(levels=2, per_leaf=6)), ("S4", 96, 54, {}), ("S5", 96, 54, {})]


def main():
    import ora
    import scenes
    import yaml_scene
    for name, w, h, kw in CASES:
        text = scenes.make(name, width=w, height=h, **kw)
        world, cams = yaml_scene.load(text, ora.API)
        r = ora.render(world, cams[0], 5, threads=8)
        st = r["stats"]
        np.savez_compressed(os.path.join(HERE, "%s_%dx%d.npz" % (name, w, h)), text=np.array(text), rgb=r["rgb"], hit=r["hit"],
                            tie=r["tie"], rays=np.array([st.rays_primary, st.rays_shadow, st.rays_secondary], dtype=np.int64))
        print(name, w, h, "rays", st.rays, "ties", int((r["tie"] != 0).sum()))


if __name__ == "__main__":
    main()
