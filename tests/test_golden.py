"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle still reproduces them bit for bit (guards the checker against drift).
GPU: the HIP path, through the C ABI, against the committed fixtures — no oracle involved at run time."""
import glob
import os

import numpy as np
import pytest

import parity

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return str(z["text"]), dict(rgb=z["rgb"], hit=z["hit"], tie=z["tie"]), [int(v) for v in z["rays"]]


def test_fixtures_present():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    import ora
    import yaml_scene
    text, want, rays = _load(path)
    world, cams = yaml_scene.load(text, ora.API)
    r = ora.render(world, cams[0], 5, threads=8)
    assert np.array_equal(r["rgb"], want["rgb"]) and np.array_equal(r["hit"], want["hit"])
    st = r["stats"]
    assert [st.rays_primary, st.rays_shadow, st.rays_secondary] == rays


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_gpu_matches_golden(path):
    import rtx
    text, want, rays = _load(path)
    ctx = rtx.Context(0)
    try:
        world, cams = rtx.load_yaml(text)
        r = ctx.render(world, cams[0], 5)
        parity.assert_parity(r, want, os.path.basename(path))
        st = r["stats"]
        assert [st.rays_primary, st.rays_shadow, st.rays_secondary] == rays
    finally:
        ctx.close()
