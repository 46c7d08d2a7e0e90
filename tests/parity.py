"""Shared helpers for the parity tests: build the same scene for the product (C++ host mirror ->
HIP kernel) and for the CPU oracle (independent Python loader -> oracle builder), render both,
compare.  Acceptance (BASELINE.json north_star): primary-hit ids bit-exact away from
EPSILON-tie pixels; 8-bit channels within +-1 LSB on >= 99.9 % of pixels."""
import numpy as np

import scenes
import yaml_scene


def build_both(text):
    import ora
    import rtx
    pw, pc = rtx.load_yaml(text)
    ow, oc = yaml_scene.load(text, ora.API)
    return (pw, pc[0]), (ow, oc[0])


def compare(gpu, cpu):
    """gpu/cpu: dicts with rgb (H,W,3) u8, hit (H,W) i32; cpu also has tie (H,W) u8."""
    tie = cpu["tie"] != 0
    id_mismatch = (gpu["hit"] != cpu["hit"]) & ~tie
    diff = np.abs(gpu["rgb"].astype(np.int16) - cpu["rgb"].astype(np.int16)).max(axis=2)
    n = diff.size
    return dict(
        n=n,
        id_mismatch=int(id_mismatch.sum()),
        id_mismatch_incl_ties=int((gpu["hit"] != cpu["hit"]).sum()),
        tie_pixels=int(tie.sum()),
        within1=float((diff <= 1).sum()) / n,
        exact=float((diff == 0).sum()) / n,
        max_diff=int(diff.max()),
    )


def assert_parity(gpu, cpu, label="", max_tie_frac=0.005):
    """The acceptance bar.  Pixels the oracle flags as EPSILON ties (a decision within 1e-11 relative) are exempt from
    the id check, so the exemption itself is bounded: at most max_tie_frac of the pixels (and never fewer than 8 allowed,
    for tiny images) may be flagged — a regression that inflated the tie flags would otherwise hide id mismatches.  The
    committed scenes S1-S5 are checked with max_tie_frac = 0.  `id_mismatch_incl_ties` is reported for the logs."""
    r = compare(gpu, cpu)
    assert r["id_mismatch"] == 0, (label, r)
    assert r["within1"] >= 0.999, (label, r)
    limit = max(8, int(max_tie_frac * r["n"])) if max_tie_frac > 0 else 0
    if max_tie_frac > 0 and r["tie_pixels"] > limit:
        # a scene full of coincident surfaces (generated scenes do that: a plane through a CSG face): no exemption at all then
        assert r["id_mismatch_incl_ties"] == 0, (label, "too many tie pixels for the exemption, and ids differ inside them", r)
    else:
        assert r["tie_pixels"] <= limit, (label, "too many tie pixels", r)
    return r


def scene_text(name, w, h, **kw):
    return scenes.make(name, width=w, height=h, **kw)
