"""GPU parity tests proper: the HIP path (through the C ABI in include/rtx.h) against the CPU
oracle on the same seeded scenes, at sizes the oracle finishes in seconds.  Bars: hit ids
bit-exact away from EPSILON-tie pixels, RGB8 within +-1 LSB on >= 99.9 % of pixels
(BASELINE.json north_star), ray counts and event counts exactly equal."""
import os

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import rtx
    c = rtx.Context(0)
    yield c
    c.close()


def test_device_math_bits(ctx):
    """f64 sqrt and division on gfx950 must be correctly rounded (== host) or hit ids can differ."""
    rng = np.random.default_rng(20261111)
    a = np.concatenate([rng.uniform(0, 1e6, 500000), 10.0 ** rng.uniform(-300, 300, 500000)])
    b = np.concatenate([rng.uniform(-1e3, 1e3, 500000), 10.0 ** rng.uniform(-150, 150, 500000)])
    assert np.array_equal(ctx.debug_math(0, a), np.sqrt(a))
    assert np.array_equal(ctx.debug_math(1, a, b), a / b)
    assert np.array_equal(ctx.debug_math(3, b), np.floor(b))
    x = rng.uniform(0, 1, 200000); y = rng.uniform(1, 400, 200000)
    p = ctx.debug_math(2, x, y); q = np.power(x, y)
    ok = np.abs(p - q) <= 4 * np.spacing(np.maximum(np.abs(q), 1e-300))
    assert ok.all(), "pow differs from glibc by more than 4 ulp"
    # specular power (csrc/rtx_pow.h): integer exponents by binary powering, relative error <= n * 2^-53 (+ margin);
    # anything else is the library pow
    yi = rng.integers(0, 400, 200000).astype(np.float64)
    p = ctx.debug_math(7, x, yi); q = np.power(x, yi)
    assert (np.abs(p - q) <= (yi + 4) * 2.0 ** -52 * np.abs(q) + 1e-320).all(), "integer power outside n * 2^-52"
    p = ctx.debug_math(7, x, y); q = np.power(x, y)
    assert (np.abs(p - q) <= 4 * np.spacing(np.maximum(np.abs(q), 1e-300))).all()
    assert ctx.debug_math(7, np.array([0.5, 0.5, 2.0]), np.array([0.0, 1.0, 4096.0])).tolist() == [1.0, 0.5, np.inf]
    nan = float("nan")
    assert ctx.debug_math(5, np.array([nan, 1.0, nan]), np.array([2.0, nan, nan]))[:2].tolist() == [2.0, 1.0]


CASES = [
    ("S1", 400, 200, {}),                      # config 1: the reference's example scene at its own size
    ("S2", 240, 136, {}),
    ("S3", 200, 112, dict(levels=2, per_leaf=6)),
    ("S4", 240, 136, {}),
    ("S5", 240, 136, {}),
    ("S1", 101, 53, {}),                       # ragged: neither dimension a multiple of the 8x8 tile
    ("S4", 7, 3, {}),                          # smaller than one tile
]


@pytest.mark.parametrize("name,w,h,kw", CASES, ids=["%s-%dx%d" % c[:3] for c in CASES])
def test_scene_parity(ctx, name, w, h, kw):
    import ora
    import rtx
    text = parity.scene_text(name, w, h, **kw)
    (pw, pc), (ow, oc) = parity.build_both(text)
    gpu = ctx.render(pw, pc, 5, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
    cpu = ora.render(ow, oc, 5, threads=8)
    r = parity.assert_parity(gpu, cpu, name)                        # tie pixels bounded at 0.5 % ...
    assert r["id_mismatch_incl_ties"] == 0                          # ... and on the committed scenes not even they differ
    gs, cs = gpu["stats"], cpu["stats"]
    assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), r
    ge, ce = gs.events(), cs.events()
    # (pow is called iff rde > 0: an ulp in rde may flip it, so it is not compared)
    import rtapi
    ge = {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"}    # reference-defined events only
    ce = {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}
    assert ge == ce
    # default flags (shadow early-out on, no counting) must give the same image
    fast = ctx.render(pw, pc, 5)
    assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"])
    fs = fast["stats"]
    assert (fs.rays_primary, fs.rays_shadow, fs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary)


def test_bands_equal_full_frame(ctx):
    """Row-interleaved 8-row bands (one residue class per GPU) reassemble to the 1-GPU image."""
    import rtx
    text = parity.scene_text("S4", 200, 100, )
    pw, pc = rtx.load_yaml(text)
    sc = ctx.scene(pw)
    full = sc.render(pc[0], 5)
    for G in (2, 4, 8):
        rgb = np.zeros_like(full["rgb"]); hit = np.full_like(full["hit"], -7)
        rays = 0
        for g in range(G):
            r = sc.render(pc[0], 5, bands=(g, G), out=(rgb, hit)); rays += r["stats"].rays
        assert np.array_equal(rgb, full["rgb"]) and np.array_equal(hit, full["hit"])
        assert rays == full["stats"].rays
        # compact layout: rank g holds its bands back to back
        for g in range(G):
            c = sc.render(pc[0], 5, bands=(g, G), flags=rtx.F_BANDS_COMPACT)
            rows = np.concatenate([np.arange(b * 8, min(b * 8 + 8, 100)) for b in range(g, 13, G)])
            assert np.array_equal(c["rgb"][:len(rows)], full["rgb"][rows])
    sc.close()


def test_errors(ctx):
    import rtx
    text = parity.scene_text("S1", 64, 32)
    pw, pc = rtx.load_yaml(text)
    with pytest.raises(rtx.RtxError) as e:
        ctx.render(pw, pc[0], 9)
    assert e.value.code == rtx.E_CAPACITY


def test_torture_scene_parity(ctx):
    """Structures the YAML DSL cannot express (tests/torture.py): CSG inside groups, a group as a CSG operand,
    empty groups, > 32 children per list, a plane inside a group, 5-deep nesting, two lights, glass."""
    import ora
    import rtapi
    import rtx
    import torture
    pw, pc = torture.build(rtx.API)
    ow, oc = torture.build(ora.API)
    gpu = ctx.render(pw, pc, 5, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
    cpu = ora.render(ow, oc, 5, threads=8)
    r = parity.assert_parity(gpu, cpu, "torture")
    gs, cs = gpu["stats"], cpu["stats"]
    assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), r
    ge, ce = gs.events(), cs.events()
    assert {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"} == {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}
    fast = ctx.render(pw, pc, 5)
    assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"])
    # an empty world renders black and counts one ray per pixel
    ew = rtx.API.world_new(); ew.build()
    e = ctx.render(ew, pc, 5)
    assert not e["rgb"].any() and (e["hit"] == -1).all() and e["stats"].rays == pc.hsize * pc.vsize


def test_cli_renders_scene_file_to_png(ctx, tmp_path):
    """main.rs:20-28 end to end: the `rtx` CLI loads a scene file, renders cameras[0] on the GPU and writes
    <stem>.png into the working directory; the decoded PNG must be the image rtx_render returns."""
    import os
    import subprocess
    from PIL import Image
    import rtx
    text = parity.scene_text("S1", 96, 48)
    scene = tmp_path / "example.yaml"
    scene.write_text(text)
    exe = os.path.join(os.path.dirname(rtx.LIB_PATH), "rtx")
    subprocess.run([exe, str(scene)], check=True, timeout=100, cwd=str(tmp_path))
    img = np.asarray(Image.open(str(tmp_path / "example.png")))
    assert img.shape == (48, 96, 3) and img.dtype == np.uint8            # 8-bit RGB (canvas.rs:45-53)
    pw, pc = rtx.load_yaml(text)
    want = ctx.render(pw, pc[0], 5)["rgb"]
    assert np.array_equal(img, want)


FULL = [("S2", 1920, 1080), ("S3", 3840, 2160), ("S4", 3840, 2160), ("S5", 7680, 4320)]


@pytest.mark.parametrize("name,w,h", FULL, ids=[c[0] for c in FULL])
def test_full_size_properties_and_sampled_parity(ctx, name, w, h):
    """BASELINE.json's full sizes: size-independent properties of the frame (determinism, counting
    identities, same image with and without the shadow early-out) plus parity with the oracle on a seeded
    sample of 8x8 tiles of the full frame (the oracle traces exactly those pixels)."""
    import ora
    import rtx
    text = parity.scene_text(name, w, h)
    (pw, pc), (ow, oc) = parity.build_both(text)
    sc = ctx.scene(pw)
    try:
        a = sc.render(pc, 5)
        b = sc.render(pc, 5)
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["hit"], b["hit"])          # deterministic
        # (frames of this size go out in interleaved band sets on two lanes, rtx_render's default: the same as ONE launch)
        ctx.set_option("host_sets", 1)
        try:
            one = sc.render(pc, 5)
        finally:
            ctx.set_option("host_sets", 0)
        assert np.array_equal(a["rgb"], one["rgb"]) and np.array_equal(a["hit"], one["hit"]) and a["stats"].rays == one["stats"].rays
        c = sc.render(pc, 5, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        assert np.array_equal(a["rgb"], c["rgb"]) and np.array_equal(a["hit"], c["hit"])
        st, ev = c["stats"], c["stats"].events()
        n_lights = text.count("- add: light")
        assert st.rays_primary == w * h
        assert st.rays_shadow == ev["light"] == n_lights * ev["hit"]                              # world.rs:129-136
        assert (a["stats"].rays_primary, a["stats"].rays_shadow, a["stats"].rays_secondary) == \
            (st.rays_primary, st.rays_shadow, st.rays_secondary)
        assert int((a["hit"] >= 0).sum()) <= ev["hit"]
        rng = np.random.default_rng(20261111)
        tx, ty = (w + 7) // 8, (h + 7) // 8
        tiles = rng.choice(tx * ty, 48, replace=False)
        xs, ys = [], []
        for t in tiles:
            gx, gy = np.meshgrid(np.arange((t % tx) * 8, min((t % tx) * 8 + 8, w)), np.arange((t // tx) * 8, min((t // tx) * 8 + 8, h)))
            xs.append(gx.ravel()); ys.append(gy.ravel())
        xs, ys = np.concatenate(xs), np.concatenate(ys)
        cpu = ora.render(ow, oc, 5, pixels=(xs, ys), threads=8)
        gpu = dict(rgb=a["rgb"][ys, xs][None], hit=a["hit"][ys, xs][None])
        parity.assert_parity(gpu, dict(rgb=cpu["rgb"][None], hit=cpu["hit"][None], tie=cpu["tie"][None]), name + " full-size sample")
    finally:
        sc.close()


def test_tile_order_does_not_change_the_image(ctx, monkeypatch):
    """The expensive-tiles-first order (probe launch + class lists, rtx_api.hip) is scheduling only: forced on and
    forced off must give the same image, hit ids and ray counts, also for ragged sizes and band subsets."""
    import rtx
    text = parity.scene_text("S4", 333, 171)
    pw, pc = rtx.load_yaml(text)
    sc = ctx.scene(pw)
    try:
        ctx.set_option("lpt", 0)
        a = sc.render(pc[0], 5)
        ctx.set_option("lpt", 1)
        b = sc.render(pc[0], 5)
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["hit"], b["hit"])
        assert a["stats"].rays == b["stats"].rays
        rgb = np.zeros_like(a["rgb"]); hit = np.full_like(a["hit"], -7)
        for g in range(3):
            sc.render(pc[0], 5, bands=(g, 3), out=(rgb, hit))
        assert np.array_equal(rgb, a["rgb"]) and np.array_equal(hit, a["hit"])
    finally:
        ctx.set_option("lpt", -1)
        sc.close()


def _with_extra_cameras(text, extra):
    return text + "".join("- add: camera\n  width: %d\n  height: %d\n  field-of-view: %s\n  from: %s\n  to: %s\n  up: [0, 1, 0]\n" % c
                          for c in extra)


def test_batch_of_cameras_equals_single_renders(ctx, monkeypatch):
    """SURVEY.md §8f N4 (rtx_render_batch): every `add: camera` of a file (yaml.rs:34; the reference renders
    only cameras[0], main.rs:24) against ONE resident scene, queued back to back with one wait.  Each frame
    (different sizes, ragged, one large enough for the tile-order probe) must be byte-identical to
    rtx_render's, with its own ray and event counts; cameras[0] is additionally checked against the oracle."""
    import ora
    import rtapi
    import rtx
    text = _with_extra_cameras(parity.scene_text("S4", 160, 96),
                               [(101, 53, "pi / 3", "[3, 2, -9]", "[0, 1, 0]"), (512, 320, "0.9", "[-4, 5, -10]", "[0, 0.5, 0]"),
                                (8, 8, "pi / 2", "[0, 1, -5]", "[0, 1, 0]")])
    pw, pcs = rtx.load_yaml(text)
    assert len(pcs) == 4
    sc = ctx.scene(pw)
    try:
        ctx.set_option("lpt", 1)                                 # the probe and its lists are shared by the queued frames
        batch = sc.render_batch(pcs, 5, flags=rtx.F_COUNT_EVENTS)
        for k, cam in enumerate(pcs):
            one = sc.render(cam, 5, flags=rtx.F_COUNT_EVENTS)
            assert np.array_equal(batch[k]["rgb"], one["rgb"]) and np.array_equal(batch[k]["hit"], one["hit"]), k
            bs, os_ = batch[k]["stats"], one["stats"]
            assert (bs.rays_primary, bs.rays_shadow, bs.rays_secondary) == (os_.rays_primary, os_.rays_shadow, os_.rays_secondary)
            # (lane_slots / op_slots count wave iterations: they depend on which tiles the launch happened to split)
            be, oe = bs.events(), os_.events()
            assert {k: v for k, v in be.items() if k not in ("lane_slots", "op_slots")} == \
                {k: v for k, v in oe.items() if k not in ("lane_slots", "op_slots")} and bs.kernel_ms > 0
        ctx.set_option("lpt", -1)
        plain = sc.render_batch(pcs, 5, want_hit=False, want_stats=False)
        assert all(np.array_equal(plain[k]["rgb"], batch[k]["rgb"]) for k in range(4))
        assert sc.render_batch([], 5) == []
        with pytest.raises(rtx.RtxError):
            sc.render_batch(pcs, 5, flags=rtx.F_BANDS_COMPACT)
    finally:
        ctx.set_option("lpt", -1)
        sc.close()
    (_, _), (ow, oc) = parity.build_both(parity.scene_text("S4", 160, 96))
    cpu = ora.render(ow, oc, 5, threads=8)
    parity.assert_parity(batch[0], cpu, "batch frame 0")
    assert batch[0]["stats"].rays == cpu["stats"].rays


def test_cli_all_cameras(ctx, tmp_path):
    """`rtx <scene> --all-cameras`: cameras[0] -> <stem>.png exactly as main.rs:25-26, the others -> <stem>.cam<i>.png."""
    import os
    import subprocess
    from PIL import Image
    import rtx
    text = _with_extra_cameras(parity.scene_text("S1", 96, 48), [(40, 30, "pi / 2", "[0, 3, -6]", "[0, 1, 0]")])
    scene = tmp_path / "two.yaml"
    scene.write_text(text)
    exe = os.path.join(os.path.dirname(rtx.LIB_PATH), "rtx")
    subprocess.run([exe, str(scene), "--all-cameras"], check=True, timeout=100, cwd=str(tmp_path))
    pw, pcs = rtx.load_yaml(text)
    for k, name in enumerate(["two.png", "two.cam1.png"]):
        img = np.asarray(Image.open(str(tmp_path / name)))
        assert np.array_equal(img, ctx.render(pw, pcs[k], 5)["rgb"]), name


@pytest.mark.parametrize("name,w,h", [("S4", 333, 171), ("S4", 640, 360), ("S5", 320, 200), ("S1", 200, 100)])
def test_split_tiles_do_not_change_the_image(ctx, monkeypatch, name, w, h):
    """Split tiles (SPLIT launch: the most expensive tiles rendered by 4 waves of 16 pixels each, refracted
    children handed to idle lanes through LDS mailboxes, every sum still evaluated by the frame's owner in the
    reference's order) are scheduling only: image, hit ids, ray counts and every reference-defined event count
    equal the unsplit render's."""
    import rtapi
    import rtx
    text = parity.scene_text(name, w, h)
    pw, pc = rtx.load_yaml(text)
    sc = ctx.scene(pw)
    try:
        ctx.options(lpt=1, split=0)
        a = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)
        ctx.options(split=1, split_class=4)
        b = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)
        ctx.set_option("split_class", 6)
        c = sc.render(pc[0], 5)
        for r in (b, c):
            assert np.array_equal(a["rgb"], r["rgb"]) and np.array_equal(a["hit"], r["hit"])
            assert (a["stats"].rays_primary, a["stats"].rays_shadow, a["stats"].rays_secondary) == \
                (r["stats"].rays_primary, r["stats"].rays_shadow, r["stats"].rays_secondary)
        ea, eb = a["stats"].events(), b["stats"].events()
        assert {k: ea[k] for k in rtapi.REF_EVENTS} == {k: eb[k] for k in rtapi.REF_EVENTS}
        rgb = np.zeros_like(a["rgb"]); hit = np.full_like(a["hit"], -7)
        for g in range(3):
            sc.render(pc[0], 5, bands=(g, 3), out=(rgb, hit))
        assert np.array_equal(rgb, a["rgb"]) and np.array_equal(hit, a["hit"])
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7)
        sc.close()


def test_split_tiles_on_the_torture_scene(ctx, monkeypatch):
    """The same invariance on the API-built torture scene (glass inside groups and CSG, two lights): the CSG
    instantiation of the SPLIT kernel hands refracted children between lanes there."""
    import rtapi
    import rtx
    import torture
    pw, pc = torture.build(rtx.API)
    try:
        ctx.options(lpt=1, split=0)
        a = ctx.render(pw, pc, 5, flags=rtx.F_COUNT_EVENTS)
        ctx.set_option("split", 1)
        for cls in (1, 4, 6):
            ctx.set_option("split_class", cls)
            b = ctx.render(pw, pc, 5, flags=rtx.F_COUNT_EVENTS)
            assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["hit"], b["hit"]), cls
            assert a["stats"].rays == b["stats"].rays
            ea, eb = a["stats"].events(), b["stats"].events()
            assert {k: ea[k] for k in rtapi.REF_EVENTS} == {k: eb[k] for k in rtapi.REF_EVENTS}, cls
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7)


def test_randomized_scene_files_parity(ctx):
    """Randomized coverage of the scene DSL (tests/fuzz_scenes.py: nested groups, CSG trees, every pattern kind with
    pattern transforms, mirror / glass / tinted materials, planes inside groups, negative scales, 1-2 lights): every
    generated file is rendered through the C ABI and by the oracle; ids, colours (the bar), ray counts and
    reference-defined event counts must agree.  Scenes beyond a documented kernel limit (RTX_E_CAPACITY /
    RTX_E_UNSUPPORTED) are skipped and counted.  The depth budget cycles through 5 (the reference's, camera.rs:42), 0, 1,
    3 and 8 (= RTX_MAX_DEPTH: the whole frame stack)."""
    import fuzz_scenes
    import ora
    import rtapi
    import rtx
    import yaml_scene
    rng = np.random.default_rng(20261111)
    done = skipped = ties = px = tie_mis = 0
    for i in range(120):
        text = fuzz_scenes.make(rng)
        depth = (5, 0, 1, 3, 8)[i % 5]
        pw, pcs = rtx.load_yaml(text)
        try:
            gpu = ctx.render(pw, pcs[0], depth, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        except rtx.RtxError as e:
            assert e.code in (rtx.E_CAPACITY, rtx.E_UNSUPPORTED), (i, str(e))
            skipped += 1
            continue
        ow, ocs = yaml_scene.load(text, ora.API)
        cpu = ora.render(ow, ocs[0], depth, threads=8)
        # (random scenes do produce coincident surfaces — a plane through a CSG face, touching operands: per scene the tie
        # exemption is bounded at 25 %, over the whole run at 3 %, and mismatches INSIDE tie pixels are counted below)
        r = parity.assert_parity(gpu, cpu, "generated scene %d depth %d\n%s" % (i, depth, text), max_tie_frac=0.25)
        px += r["n"]; tie_mis += r["id_mismatch_incl_ties"]
        gs, cs = gpu["stats"], cpu["stats"]
        if r["tie_pixels"] == 0:
            # (a tie pixel may legitimately follow another branch of the ray tree, so its rays differ)
            assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), (i, r)
            ge, ce = gs.events(), cs.events()
            assert {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"} == {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}, i
        fast = ctx.render(pw, pcs[0], depth)
        assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]), i
        # (a scene over the per-scene bound had to match without any exemption: its flags do not count as exempted pixels)
        ties += r["tie_pixels"] if r["tie_pixels"] <= 0.25 * r["n"] else 0; done += 1
    assert done >= 90, (done, skipped)
    print("randomized scene files: %d scenes, %d pixels, %d tie pixels, %d id mismatches inside them" % (done, px, ties, tie_mis))
    assert ties <= 0.03 * px and tie_mis <= 0.02 * max(ties, 1) + 2, (ties, px, tie_mis)


def test_children_outside_the_reference_bounds_of_their_group(ctx):
    """Regression (found by tests/sweep_fuzz.py, 24 of 1 500 scenes): the reference's group bounds need not contain the
    children (Bounds::from_points never looks at the first corner for `max`, bounds.rs:29-40; a rotated plane's infinite box
    turns into garbage), and its bounds test is one against the whole line.  The product also drops children wholly BEHIND
    the ray origin, which is only sound with a sphere around the children themselves (Compiler::geom_sphere), not around
    those bounds.  Two minimal scenes (rotated groups with a plane inside): the product render, the reference-faithful
    counting render and the oracle must agree."""
    import glob
    import ora
    import rtx
    import yaml_scene
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress_behind_*.yaml")))
    assert len(files) >= 2
    for f in files:
        text = open(f).read()
        depth = int(f.rsplit("depth", 1)[1].split(".")[0])
        pw, pcs = rtx.load_yaml(text)
        ow, ocs = yaml_scene.load(text, ora.API)
        for dd in sorted({depth, 5}):
            faithful = ctx.render(pw, pcs[0], dd, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
            cpu = ora.render(ow, ocs[0], dd, threads=8)
            parity.assert_parity(faithful, cpu, "%s depth %d" % (f, dd), max_tie_frac=0.25)
            fast = ctx.render(pw, pcs[0], dd)
            assert np.array_equal(fast["rgb"], faithful["rgb"]) and np.array_equal(fast["hit"], faithful["hit"]), (f, dd)
            assert fast["stats"].rays == faithful["stats"].rays, (f, dd)


def test_randomized_api_scene_graphs_parity(ctx):
    """Randomized scene GRAPHS through the WorldBuilder API (tests/fuzz_api.py): groups as CSG operands, CSG inside
    groups, empty groups, lists of more than 32 children, nesting up to 7, planes anywhere, 0-3 lights — structures
    the scene DSL cannot express.  Same checks as for the generated scene files, depth budgets 0..8."""
    import fuzz_api
    import ora
    import rtapi
    import rtx
    rng = np.random.default_rng(20261112)
    rng_big = np.random.default_rng(20261113)
    done = px = ties = tie_mis = 0
    for i in range(56):
        # the last 16: with CSG roots beyond the fast kernels' bounds (up to 40 leaves, nesting up to 14: general CSG kernels)
        p = fuzz_api.plan(rng) if i < 40 else fuzz_api.plan(rng_big, big_csg=True)
        depth = (5, 0, 2, 8)[i % 4]
        pw, pc = fuzz_api.build(rtx.API, p)
        try:
            gpu = ctx.render(pw, pc, depth, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        except rtx.RtxError as e:
            assert e.code in (rtx.E_CAPACITY, rtx.E_UNSUPPORTED), (i, str(e))
            continue
        ow, oc = fuzz_api.build(ora.API, p)
        cpu = ora.render(ow, oc, depth, threads=8)
        # (random graphs contain coincident surfaces — the same plane twice, touching CSG operands — whose every pixel is a
        # tie: a scene with more than 25 % tie pixels must then match WITHOUT the exemption; the others are bounded per
        # scene and, below, over the whole run)
        r = parity.assert_parity(gpu, cpu, "generated graph %d depth %d" % (i, depth), max_tie_frac=1.0)
        if r["tie_pixels"] > 0.25 * r["n"]:
            assert r["id_mismatch_incl_ties"] == 0, (i, r)
        else:
            px += r["n"]; ties += r["tie_pixels"]; tie_mis += r["id_mismatch_incl_ties"]
        if r["tie_pixels"] == 0:
            gs, cs = gpu["stats"], cpu["stats"]
            assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), (i, r)
            ge, ce = gs.events(), cs.events()
            assert {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"} == {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}, i
        fast = ctx.render(pw, pc, depth)
        assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]), i
        done += 1
    assert done >= 46
    print("randomized scene graphs: %d scenes, %d pixels, %d tie pixels, %d id mismatches inside them" % (done, px, ties, tie_mis))
    assert ties <= 0.03 * px and tie_mis <= 0.02 * max(ties, 1) + 2, (ties, px, tie_mis)


def test_kernel_limits_are_reported_not_truncated(ctx):
    """The bounded structures of the FAST kernels (include/rtx.h: RTX_MAX_CSG_ENTRIES, RTX_MAX_CSG_NEST) no longer bound the
    scenes: a CSG root beyond them renders with the general CSG kernels (lists of any length and nesting in a device arena —
    what csg.rs:73-91 / intersections.rs:41-59 build with a Vec) and must match the oracle like any other scene: ids, colours,
    ray and event counts.  What remains limited (container nesting of more than RTX_MAX_LEVELS, nested transforms beyond the
    wave's LDS ray stack, non-affine transforms) is an error code at rtx_scene_create, never a silently wrong image."""
    import ora
    import rtapi
    import rtx

    def chain(api, n_leaves):
        """union(union(...union(s0, s1)..., s_{n-2}), s_{n-1}): n_leaves shapes under one CSG root, n_leaves - 1 nested CSGs."""
        w = api.world_new()
        node = rtapi.shape(w.add_shape(rtapi.SPHERE, None, api.translation(-2.0, 0.5, 0)))
        for k in range(1, n_leaves):
            s = rtapi.shape(w.add_shape(rtapi.SPHERE if k % 2 else rtapi.CUBE, None,
                                        api.compose([("scale", [0.5, 0.5, 0.5]), ("translate", [-2.0 + 4.4 * k / n_leaves, 0.5, 0.8 * k / n_leaves])])))
            node = rtapi.csg(w.add_csg(rtapi.UNION, node, s))
        w.add_object_to_scene(node)
        w.add_light([-5, 8, -8], [1, 1, 1])
        w.build()
        return w, api.camera(96, 48, 1.0, api.view_transform([0, 2, -9], [0, 0.5, 0], [0, 1, 0]))

    def nested_csg(api, depth):
        """`depth` CSGs inside each other, every one with a transform of its own (a new ray-stack level each), the operators
        cycling union / difference / intersection, operands alternating sides; glass and mirror materials so that the refractive
        containers and the secondary rays go through the general lists too."""
        w = api.world_new()
        glass = api.material(color=[0.2, 0.3, 0.9], transparency=0.7, reflective=0.3, refractive_index=1.4, diffuse=0.4)
        node = rtapi.shape(w.add_shape(rtapi.SPHERE, glass, api.scaling(1.2, 1.2, 1.2)))
        ops = [rtapi.UNION, rtapi.DIFFERENCE, rtapi.UNION, rtapi.INTERSECTION, rtapi.UNION]
        for k in range(depth):
            big = ops[k % 5] == rtapi.INTERSECTION
            s = rtapi.shape(w.add_shape(rtapi.CUBE if k % 2 else rtapi.SPHERE, glass if k % 3 == 0 else None,
                                        api.compose([("scale", [2.5, 2.5, 2.5] if big else [0.45, 0.6, 0.45]),
                                                     ("translate", [0, 0, 0] if big else [1.2 * ((k % 4) - 1.5), 0.3 * (k % 3), 0.4 * ((k % 2) - 0.5)])])))
            xf = api.compose([("rotate-y", 0.05 * (k + 1)), ("translate", [0.02 * k, 0.01, 0])])
            op = ops[k % 5]
            node = rtapi.csg(w.add_csg(op, node, s, xf) if (op == rtapi.DIFFERENCE or k % 2) else w.add_csg(op, s, node, xf))
        w.add_object_to_scene(node)
        w.add_object_to_scene(rtapi.shape(w.add_shape(rtapi.PLANE, api.material(color=[0.8, 0.8, 0.7], reflective=0.2), api.translation(0, -1.5, 0))))
        w.add_light([-5, 8, -8], [1, 1, 1])
        w.add_light([6, 5, -7], [0.4, 0.4, 0.4])
        w.build()
        return w, api.camera(96, 64, 1.0, api.view_transform([0, 2, -9], [0, 0.3, 0], [0, 1, 0]))

    def nested_groups(api, depth):
        w = api.world_new()
        g = w.add_group(api.translation(0, 0.1, 0))
        w.add_object_to_scene(rtapi.group(g))
        for _ in range(depth - 1):
            h = w.add_group(api.compose([("rotate-y", 0.1), ("translate", [0.05, 0.02, 0])]))
            w.set_group_of(rtapi.group(h), g)
            g = h
        w.set_group_of(rtapi.shape(w.add_shape(rtapi.SPHERE)), g)
        w.add_light([-5, 8, -8], [1, 1, 1])
        w.build()
        return w, api.camera(64, 48, 1.0, api.view_transform([0, 1, -6], [0, 0.5, 0], [0, 1, 0]))

    # at the limits of the fast kernels (8 leaf shapes = 16 list entries under one root; the old nesting limit), past them
    # (general CSG kernels: 9 and 32 leaves = 8 and 31 nested CSGs, 12 CSGs with transforms inside each other) and deeper groups
    for builder, arg in ((chain, 8), (nested_groups, 16), (chain, 9), (chain, 32), (nested_csg, 12), (nested_csg, 3), (nested_groups, 17)):
        pw, pc = builder(rtx.API, arg)
        ow, oc = builder(ora.API, arg)
        gpu = ctx.render(pw, pc, 5, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        cpu = ora.render(ow, oc, 5, threads=8)
        what = "%s(%d)" % (builder.__name__, arg)
        parity.assert_parity(gpu, cpu, what)
        gs, cs = gpu["stats"], cpu["stats"]
        assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), what
        assert (gpu["hit"] >= 0).any(), what
        ge, ce = gs.events(), cs.events()
        assert {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"} == {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}, what
        # the product mode (early-outs, tight spheres, tile lists forced on) renders the same bytes
        ctx.options(lpt=1)
        fast = ctx.render(pw, pc, 5)
        ctx.options(lpt=-1)
        assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]), what
    # what is still limited is reported: nesting beyond RTX_MAX_LEVELS, nested general transforms beyond the LDS ray stack
    for builder, arg in ((nested_groups, 63), (nested_groups, 30)):
        pw, pc = builder(rtx.API, arg)
        with pytest.raises(rtx.RtxError) as e:
            ctx.render(pw, pc, 5)
        assert e.value.code == rtx.E_CAPACITY, (builder.__name__, arg, str(e.value))
    # a projective (non-affine) transform cannot come out of the scene DSL; the API rejects it
    w = rtx.API.world_new()
    m = rtx.API.identity(); m[3] = 0.25                       # bottom row (0.25, 0, 0, 1)
    w.add_object_to_scene(rtapi.shape(w.add_shape(rtapi.SPHERE, None, m)))
    w.build()
    with pytest.raises(rtx.RtxError) as e:
        ctx.render(w, rtx.API.camera(16, 16, 1.0), 5)
    assert e.value.code == rtx.E_UNSUPPORTED


class _Hip:
    """Minimal ctypes view of the HIP runtime librtx_hip.so itself is linked against (device buffers and a stream
    for the rtx_render_device test; torch is deliberately not involved: it brings a HIP runtime of its own)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.lib.hipStreamDestroy.argtypes = [C.c_void_p]

    def malloc(self, n, fill=0):
        p = self.C.c_void_p()
        assert self.lib.hipMalloc(self.C.byref(p), n) == 0
        assert self.lib.hipMemset(p, fill, n) == 0
        return p

    def to_host(self, p, n, dtype):
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        assert self.lib.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, n, 2) == 0      # hipMemcpyDeviceToHost
        return out


def test_every_way_of_rendering_agrees_on_random_scenes(ctx):
    """tests/sweep_entry_points.py on 36 seeded random scenes (files and API-built graphs, a sixth with general-CSG roots, sizes up
    to 400 x 240, depth budgets 0-8): tile lists from the probe and from measured costs, split tiles, time-ordered tiles, band sets,
    frames queued on the caller's stream, the no-alloc entry point, a batch, frames in flight over a loopback communicator of 2-8
    ranks - all with random option settings - leave image, hit ids and ray counts exactly as the plain render has them.  The
    sweep itself runs by hand over thousands of scenes (profiles/r3j_entry_points_15000_tail.txt)."""
    import sweep_entry_points
    msgs = []
    n, bad = sweep_entry_points.sweep(ctx, _Hip(), 1, 20261115, 400, 240, per_seed=36, log=msgs.append)
    assert n >= 30 and bad == 0, msgs


def test_general_csg_scenes_through_every_entry_point(ctx):
    """Scenes whose CSG roots exceed the fast kernels' bounds (general CSG kernels: lists in the device arena, one slot claimed per
    wave) through every way of rendering: rtx_render (one launch and band sets on two lanes, i.e. two launches claiming slots at
    the same time), tile lists and split tiles forced on (the general kernels have no SPLIT variant: the request is ignored),
    rtx_render_device on a caller's stream, the no-alloc entry point (the arena is allocated with the scene), a batch of
    cameras on four lanes, and the 3-rank band path of a loopback communicator.  All must equal the plain render."""
    import ctypes as C
    import fuzz_api
    import rtx
    hip = _Hip()
    rng = np.random.default_rng(20261114)
    bufs = []
    try:
        for i in range(3):
            p = fuzz_api.plan(rng, 160, 96, big_csg=True)
            pw, pc = fuzz_api.build(rtx.API, p)
            sc = ctx.scene(pw)
            try:
                ctx.set_option("host_sets", 1)
                want = sc.render(pc, 5)
                assert (want["hit"] >= 0).any()
                H, W = pc.vsize, pc.hsize
                for sets in (2, 3):
                    ctx.set_option("host_sets", sets)
                    got = sc.render(pc, 5)
                    assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["hit"], want["hit"]), (i, sets)
                ctx.set_option("host_sets", 1)
                ctx.options(lpt=1, split=1, split_class=1)
                got = sc.render(pc, 5)
                assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["hit"], want["hit"]) and got["stats"].rays == want["stats"].rays, i
                ctx.options(lpt=-1, split=-1, split_class=7)
                rgb = hip.malloc(H * W * 3, 0x11); ids = hip.malloc(H * W * 4, 0x22); rgb2 = hip.malloc(H * W * 3, 0x11)
                bufs += [rgb, ids, rgb2]
                st = sc.render_device(pc, rgb.value, ids.value, 5, stats=True)
                assert st.rays == want["stats"].rays
                sc.render_device_noalloc(pc, rgb2.value, 0, 5)
                ctx.poll(wait=True)
                for b in (rgb, rgb2):
                    assert np.array_equal(hip.to_host(b, H * W * 3, np.uint8).reshape(H, W, 3), want["rgb"]), i
                assert np.array_equal(hip.to_host(ids, H * W * 4, np.int32).reshape(H, W), want["hit"]), i
                for one in sc.render_batch([pc] * 5, 5):
                    assert np.array_equal(one["rgb"], want["rgb"]) and np.array_equal(one["hit"], want["hit"]), i
                comm = rtx.Comm.loopback(ctx, 3)
                try:
                    frame = hip.malloc(H * W * 3, 0x33); bufs.append(frame)
                    for rep in range(3):
                        comm.render_gathered(sc, pc, frame.value, 0, 5)
                    comm.wait()
                    assert np.array_equal(hip.to_host(frame, H * W * 3, np.uint8).reshape(H, W, 3), want["rgb"]), i
                finally:
                    comm.close()
            finally:
                ctx.options(lpt=-1, split=-1, split_class=7)
                ctx.set_option("host_sets", 0)
                sc.close()
    finally:
        for b in bufs:
            hip.lib.hipFree(b)



def test_render_device_writes_device_buffers_on_the_callers_stream(ctx):
    """rtx_render_device (the entry point bench.py and a multi-GPU host use): outputs are device pointers, the work
    is queued on the caller's stream and nothing is copied.  Full frame with hit ids and stats, then asynchronous
    compact band sets of 3 ranks reassembled on the host, must equal rtx_render's image."""
    import ctypes as C
    import rtx
    hip = _Hip()
    text = parity.scene_text("S4", 333, 171)
    pw, pc = rtx.load_yaml(text)
    sc = ctx.scene(pw)
    st = C.c_void_p()
    assert hip.lib.hipStreamCreate(C.byref(st)) == 0
    bufs = []
    try:
        want = sc.render(pc[0], 5)
        H, W = pc[0].vsize, pc[0].hsize
        rgb = hip.malloc(H * W * 3); hit = hip.malloc(H * W * 4, 0xee)
        bufs += [rgb, hit]
        stats = sc.render_device(pc[0], rgb.value, hit.value, 5, stream=st.value, stats=True)
        assert stats.rays == want["stats"].rays and stats.kernel_ms > 0
        assert np.array_equal(hip.to_host(rgb, H * W * 3, np.uint8).reshape(H, W, 3), want["rgb"])
        assert np.array_equal(hip.to_host(hit, H * W * 4, np.int32).reshape(H, W), want["hit"])
        out = np.zeros_like(want["rgb"])
        parts = []
        for g in range(3):
            rows = rtx.LIB.rtx_band_rows(H, rtx.Bands(g, 3))
            b = hip.malloc(rows * W * 3)
            bufs.append(b)
            assert sc.render_device(pc[0], b.value, 0, 5, bands=(g, 3), flags=rtx.F_BANDS_COMPACT, stream=st.value) is None
            parts.append((g, rows, b))
        assert hip.lib.hipStreamSynchronize(st) == 0                 # the three launches were only queued
        for g, rows, b in parts:
            img = hip.to_host(b, rows * W * 3, np.uint8).reshape(rows, W, 3)
            ys = np.concatenate([np.arange(k * 8, min(k * 8 + 8, H)) for k in range(g, (H + 7) // 8, 3)])
            out[ys] = img[:len(ys)]
        assert np.array_equal(out, want["rgb"])
    finally:
        sc.close()
        for b in bufs:
            hip.lib.hipFree(b)
        hip.lib.hipStreamDestroy(st)


def test_probe_ahead_of_queued_frames_does_not_change_them(ctx):
    """A frame queued behind a running one gets its tile-order probe on the lane's side stream, into the list set the running
    frame is not reading (rtx_api.hip: probe ahead).  Eight frames of alternating cameras and SIZES queued back to back on
    one stream (so a list set that belonged to the wrong frame would leave tiles unrendered or render them twice), forced
    on (option probe_ahead = 1) and in the automatic mode, must equal the frames rendered one at a time; ray counts through
    the asynchronous counters are checked by the last, synchronous frame."""
    import ctypes as C
    import rtx
    hip = _Hip()
    cams, wants, worlds = [], [], []
    st = C.c_void_p()
    assert hip.lib.hipStreamCreate(C.byref(st)) == 0
    bufs = []
    ctx.options(lpt=1, split=0)
    try:
        pw, pc = rtx.load_yaml(parity.scene_text("S4", 333, 171))
        sc = ctx.scene(pw)
        for w, h in ((333, 171), (200, 301), (640, 96)):
            _, c = rtx.load_yaml(parity.scene_text("S4", w, h))
            cams.append(c[0]); wants.append(sc.render(c[0], 5))
        for mode in (1, -1):
            ctx.set_option("probe_ahead", mode)
            outs = []
            for k in range(8):
                cam = cams[k % 3]
                b = hip.malloc(cam.vsize * cam.hsize * 3, 0x55); bufs.append(b)
                assert sc.render_device(cam, b.value, 0, 5, stream=st.value) is None
                outs.append((k % 3, b))
            stats = sc.render_device(cams[0], outs[0][1].value, 0, 5, stream=st.value, stats=True)
            assert stats.rays == wants[0]["stats"].rays
            assert hip.lib.hipStreamSynchronize(st) == 0
            for i, b in outs:
                cam = cams[i]
                got = hip.to_host(b, cam.vsize * cam.hsize * 3, np.uint8).reshape(cam.vsize, cam.hsize, 3)
                assert np.array_equal(got, wants[i]["rgb"]), (mode, i)
        sc.close()
    finally:
        ctx.options(lpt=-1, split=-1)
        ctx.set_option("probe_ahead", -1)
        for b in bufs:
            hip.lib.hipFree(b)
        hip.lib.hipStreamDestroy(st)


def test_in_stream_probe_then_probe_ahead_on_a_fresh_context():
    """on a FRESH context in the default mode (probe_ahead = -1) the first frame of a burst is queued on an
    idle stream and takes the in-stream probe into list set 0; the next frames find the stream busy and put their probes on the
    side stream, alternating the sets - the second one refills set 0.  Every frame launch that reads a set now records its
    event (the events are created with the lane, not by the first ahead-frame), so the refill waits for frame 1.  Frames of
    alternating sizes (a set that belonged to another frame leaves tiles unrendered or renders them twice) queued back to back
    on one stream, then on two streams in turn, must equal the frames rendered one at a time.  Large enough that the stream is
    still busy when the next frame is queued.
    The two-stream leg is also the case the tile-cost arrays guard against: frames of one context on different streams share the
    context's counter block, whose spare word names the cost array of the render queued LAST; the tail waves of a 9 090-tile
    frame can therefore find the array of the 117-tile frame queued behind them - they check their tile index against the
    capacity the array itself carries (rtx_device.h: RTX_COST_HEADER) and never write past it."""
    import ctypes as C
    import rtx
    hip = _Hip()
    bufs = []
    streams = []
    # the expected images come from a context of their own: the context under test must not have rendered anything before
    ref = rtx.Context(0)
    cams, wants = [], []
    try:
        pw, _ = rtx.load_yaml(parity.scene_text("S4", 960, 544))
        rsc = ref.scene(pw)
        for w, h in ((960, 544), (720, 808), (1280, 296), (104, 72)):
            _, c = rtx.load_yaml(parity.scene_text("S4", w, h))
            cams.append(c[0]); wants.append(rsc.render(c[0], 5))
        rsc.close()
    finally:
        ref.close()
    for n_streams in (1, 2):
        fresh = rtx.Context(0)
        try:
            fresh.options(lpt=1, split=0)                   # tile lists on, no split launch; probe_ahead stays at its default
            sc = fresh.scene(pw)
            for _ in range(n_streams):
                st = C.c_void_p()
                assert hip.lib.hipStreamCreate(C.byref(st)) == 0
                streams.append(st)
            sts = streams[-n_streams:]
            outs = []
            order = (0, 1, 2, 0, 1, 3, 1, 3, 2, 1, 3)
            for k, i in enumerate(order):
                cam = cams[i]
                b = hip.malloc(cam.vsize * cam.hsize * 3, 0x55); bufs.append(b)
                assert sc.render_device(cam, b.value, 0, 5, stream=sts[k % n_streams].value) is None
                outs.append((i, b))
            for st in sts:
                assert hip.lib.hipStreamSynchronize(st) == 0
            fresh.poll(wait=True)
            for k, (i, b) in enumerate(outs):
                cam = cams[i]
                got = hip.to_host(b, cam.vsize * cam.hsize * 3, np.uint8).reshape(cam.vsize, cam.hsize, 3)
                assert np.array_equal(got, wants[i]["rgb"]), (n_streams, k, int((got != wants[i]["rgb"]).any(axis=2).sum()))
            sc.close()
        finally:
            fresh.close()
    for b in bufs:
        hip.lib.hipFree(b)
    for st in streams:
        hip.lib.hipStreamDestroy(st)


def test_stats_render_does_not_wipe_an_earlier_async_overflow(ctx):
    """a synchronous render (stats) that follows an overflowing ASYNCHRONOUS one on the same stream used to
    clear the sticky word after its own sync - the earlier frame's overflow was raised during that very sync and lost.  Now a
    raised word with clean own counters is reported as the earlier render's RTX_E_CAPACITY.  Same for the band-set path of
    rtx_render."""
    import rtx
    hip = _Hip()
    pw, pc = rtx.load_yaml(parity.scene_text("S4", 960, 544))
    sc = ctx.scene(pw)
    buf = hip.malloc(544 * 960 * 3, 0)
    try:
        for path in ("device", "host_sets"):
            ctx.options(lpt=1, split=1, split_class=1, debug=2)     # debug=2: split guard of one iteration
            assert sc.render_device(pc[0], buf.value, 0, 5) is None  # queued; trips the guard while the next call is made
            ctx.options(lpt=-1, split=0, split_class=7, debug=0)
            if path == "host_sets":
                ctx.set_option("host_sets", 2)
            with pytest.raises(rtx.RtxError) as e:
                if path == "device":
                    sc.render_device(pc[0], buf.value, 0, 5, stats=True)
                else:
                    sc.render(pc[0], 5)
            assert e.value.code == rtx.E_CAPACITY and "earlier" in str(e.value), (path, str(e.value))
            ctx.set_option("host_sets", 0)
            # whichever check caught it (the one before queueing or the one after the sync), poll clears it and work goes on
            rtx.LIB.rtx_ctx_poll(ctx.h, 1)
            ctx.poll(wait=True)
            assert sc.render_device(pc[0], buf.value, 0, 5, stats=True).rays > 0
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7, debug=0)
        ctx.set_option("host_sets", 0)
        rtx.LIB.rtx_ctx_poll(ctx.h, 1)
        sc.close()
        hip.lib.hipFree(buf)



@pytest.mark.parametrize("name,w,h", [("S4", 333, 171), ("S5", 320, 200), ("S3", 200, 120), ("S1", 200, 100)])
def test_hit_context_in_lds_does_not_change_the_image(ctx, name, w, h):
    """The hit context (point, eye vector, normal) waits either in registers or in LDS while a hit's shadow rays are traced
    (template flag HCTX of the kernel, picked per launch from the wave's LDS budget; option "hctx" forces it).  Placement only:
    image, hit ids, ray and event counts must be identical, also with tile lists and split tiles on."""
    import rtx
    pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
    sc = ctx.scene(pw)
    try:
        got = {}
        for hctx in (0, 1):
            ctx.set_option("hctx", hctx)
            got[hctx] = [sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)]
            ctx.options(lpt=1, split=1, split_class=1)
            got[hctx].append(sc.render(pc[0], 5))
            ctx.options(lpt=-1, split=-1, split_class=7)
        for a, b in zip(got[0], got[1]):
            assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["hit"], b["hit"]), name
            assert a["stats"].rays == b["stats"].rays
        ea, eb = got[0][0]["stats"].events(), got[1][0]["stats"].events()
        assert ea == eb, name
        assert np.array_equal(got[0][0]["rgb"], got[0][1]["rgb"])
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7)
        ctx.set_option("hctx", -1)
        sc.close()



def test_tile_cost_feedback_orders_tiles_without_changing_the_image(ctx):
    """Tile-cost feedback (rtx_api.hip): every frame launch leaves per tile the traced rays of its busiest pixel, and the next
    frame of the same view (size, bands, depth budget; camera equal or close) files its tiles by those costs with a classify
    launch instead of a probe launch.  Scheduling only: a sequence of frames — same camera, a slightly moved camera, a camera
    jump, another size, band sets, split tiles forced on — must equal the renders with the feedback off byte for byte (ids
    and ray counts too), and the launch counters must show probes only where a view is new, jumps, or is due for a refresh."""
    import rtx
    def counters():
        return (rtx.LIB.rtx_debug_counter(ctx.h, b"probe_launches"), rtx.LIB.rtx_debug_counter(ctx.h, b"classify_launches"))
    pw, pc = rtx.load_yaml(parity.scene_text("S4", 333, 171))
    near = rtx.API.camera(333, 171, 1.0471975511965976, rtx.API.view_transform([0.02, 3.21, -10.99], [0, 1.2, 0], [0, 1, 0]))
    far = rtx.API.camera(333, 171, 1.0471975511965976, rtx.API.view_transform([6, 2, -9], [0, 1, 0], [0, 1, 0]))
    _, other = rtx.load_yaml(parity.scene_text("S4", 200, 301))
    sc = ctx.scene(pw)
    try:
        ctx.options(lpt=1, split=0, cost_feedback=0)
        want = {k: sc.render(c, 5) for k, c in (("a", pc[0]), ("near", near), ("far", far), ("other", other[0]))}
        want_bands = [sc.render(pc[0], 5, bands=(g, 3), flags=rtx.F_BANDS_COMPACT) for g in range(3)]
        ctx.options(cost_feedback=-1)
        def same(got, w):
            assert np.array_equal(got["rgb"], w["rgb"]) and np.array_equal(got["hit"], w["hit"]) and got["stats"].rays == w["stats"].rays
        p0, c0 = counters()
        for k in range(3):
            same(sc.render(pc[0], 5), want["a"])
        assert counters() == (p0 + 1, c0 + 2)                     # first frame of the view probes, the next two classify
        same(sc.render(near, 5), want["near"])                    # a camera close to the last one: the costs still order the tiles
        assert counters() == (p0 + 1, c0 + 3)
        same(sc.render(far, 5), want["far"])                      # a jump: probe
        assert counters() == (p0 + 2, c0 + 3)
        same(sc.render(other[0], 5), want["other"])               # another image size: a view of its own
        same(sc.render(other[0], 5), want["other"])
        assert counters() == (p0 + 3, c0 + 4)
        for rep in range(2):                                      # band sets are views of their own (rank r of 3)
            for g in range(3):
                same(sc.render(pc[0], 5, bands=(g, 3), flags=rtx.F_BANDS_COMPACT), want_bands[g])
        assert counters() == (p0 + 6, c0 + 7)
        # split tiles: the top class comes from the measured costs; a split tile keeps its last figure, refreshed by a probe every 32 frames
        ctx.options(split=1, split_class=5)
        p1, c1 = counters()
        for k in range(36):
            got = sc.render(far, 5)
            if k in (0, 1, 35):
                same(got, want["far"])
        p2, c2 = counters()
        assert (p2 - p1, c2 - c1) == (1, 35), (p2 - p1, c2 - c1)
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7, cost_feedback=-1)
        sc.close()



@pytest.mark.parametrize("name,w,h", [("S3", 333, 171), ("S5", 320, 200), ("S2", 200, 120)])
def test_time_ordered_tiles_of_uniform_scenes_do_not_change_the_image(ctx, name, w, h):
    """Uniform scenes (no branching material) get their tile order from the wave DURATIONS the previous frame of the view measured
    (time mode of the tile-cost feedback; by default only for launches of >= RTX_TIME_LPT_MIN_TILES tiles, forced here for
    small frames with option time_lpt = 1).  Scheduling only: frames 2.. of a view (ordered) must equal frame 1 (plain
    permuted order) and the renders with the feature off, ids and counts included; band sets too."""
    import rtx
    pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
    sc = ctx.scene(pw)
    try:
        ctx.set_option("time_lpt", 0)
        want = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)
        want_b = [sc.render(pc[0], 5, bands=(g, 2), flags=rtx.F_BANDS_COMPACT) for g in range(2)]
        ctx.set_option("time_lpt", 1)
        c0 = rtx.LIB.rtx_debug_counter(ctx.h, b"classify_launches")
        p0 = rtx.LIB.rtx_debug_counter(ctx.h, b"probe_launches")
        for k in range(4):
            got = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)
            assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["hit"], want["hit"]), (name, k)
            assert got["stats"].rays == want["stats"].rays
            ge, we = got["stats"].events(), want["stats"].events()
            assert {k2: ge[k2] for k2 in ge if k2 not in ("lane_slots", "op_slots")} == {k2: we[k2] for k2 in we if k2 not in ("lane_slots", "op_slots")}, (name, k)
        for rep in range(2):
            for g in range(2):
                got = sc.render(pc[0], 5, bands=(g, 2), flags=rtx.F_BANDS_COMPACT)
                assert np.array_equal(got["rgb"], want_b[g]["rgb"]) and np.array_equal(got["hit"], want_b[g]["hit"]), (name, g)
        # the first frame of each of the three views only measures; the others are ordered by a classify launch; never a probe
        assert rtx.LIB.rtx_debug_counter(ctx.h, b"classify_launches") - c0 == 3 + 2
        assert rtx.LIB.rtx_debug_counter(ctx.h, b"probe_launches") == p0
    finally:
        ctx.set_option("time_lpt", -1)
        sc.close()



def test_host_render_in_band_sets_equals_one_launch(ctx):
    """rtx_render of a whole frame into host buffers renders large frames as 2..8 interleaved band sets on two lanes and copies
    each set out while the next ones render (option host_sets; 0 = by frame size).  Forced here on small ragged frames:
    image, hit ids, ray counts and event counts must equal the one-launch path's."""
    import rtx
    try:
        for name, w, h in (("S4", 333, 171), ("S5", 200, 97), ("S1", 64, 9), ("S3", 160, 120)):
            pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
            sc = ctx.scene(pw)
            try:
                ctx.set_option("host_sets", 1)
                want = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS)
                for sets in (2, 3, 8):
                    ctx.set_option("host_sets", sets)
                    for want_hit in (True, False):
                        got = sc.render(pc[0], 5, flags=rtx.F_COUNT_EVENTS, want_hit=want_hit)
                        assert np.array_equal(got["rgb"], want["rgb"]), (name, sets)
                        if want_hit:
                            assert np.array_equal(got["hit"], want["hit"]), (name, sets)
                        a, b = got["stats"], want["stats"]
                        assert (a.rays_primary, a.rays_shadow, a.rays_secondary, a.rays_shadow_skipped) == \
                               (b.rays_primary, b.rays_shadow, b.rays_secondary, b.rays_shadow_skipped), (name, sets)
                        ea, eb = a.events(), b.events()
                        for k in eb:
                            if k not in ("lane_slots", "op_slots"):
                                assert ea[k] == eb[k], (name, sets, k)
            finally:
                sc.close()
    finally:
        ctx.set_option("host_sets", 0)


def test_band_shares_of_2_3_8_ranks_assemble_to_the_full_frame_on_the_device(ctx):
    """The N > 1 data path without RCCL: a loopback communicator (rtx_comm_init_loopback) renders the compact band sets of
    all N ranks on this one GPU, each where the all-gather would have delivered it, and the product's de-interleave kernel
    assembles the frame.  Must equal rtx_render byte for byte (ids and ray counts too), for ragged heights (last band
    partial, ranks with unequal numbers of bands, more ranks than bands) and row lengths that are and are not multiples of 16 bytes."""
    import rtx
    hip = _Hip()
    bufs = []
    try:
        for name, w, h in (("S4", 333, 171), ("S5", 320, 180), ("S1", 64, 9), ("S2", 128, 70)):
            pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
            sc = ctx.scene(pw)
            try:
                want = sc.render(pc[0], 5)
                for n in (2, 3, 8):
                    comm = rtx.Comm.loopback(ctx, n)
                    try:
                        assert rtx.LIB.rtx_comm_size(comm.h) == n
                        for rep in range(2):                        # both internal buffer sets
                            frame = hip.malloc(h * w * 3, 0x33); ids = hip.malloc(h * w * 4, 0x44)
                            bufs += [frame, ids]
                            stats = comm.render_gathered(sc, pc[0], frame.value, ids.value, 5, stats=(rep == 0))
                            comm.wait()
                            assert np.array_equal(hip.to_host(frame, h * w * 3, np.uint8).reshape(h, w, 3), want["rgb"]), (name, n)
                            assert np.array_equal(hip.to_host(ids, h * w * 4, np.int32).reshape(h, w), want["hit"]), (name, n)
                            if stats is not None:
                                assert (stats.rays_primary, stats.rays_shadow, stats.rays_secondary) == \
                                       (want["stats"].rays_primary, want["stats"].rays_shadow, want["stats"].rays_secondary), (name, n)
                    finally:
                        comm.close()
            finally:
                sc.close()
        # frames in flight: six frames of two cameras queued back to back without a wait in between (the communicator renders
        # consecutive frames on alternating lanes, so they overlap on the GPU), each into a buffer of its own
        pw, pc = rtx.load_yaml(parity.scene_text("S4", 333, 171))
        sc = ctx.scene(pw)
        try:
            cam2 = rtx.load_yaml(parity.scene_text("S4", 200, 120))[1][0]
            wants = [sc.render(pc[0], 5), sc.render(cam2, 5)]
            for n_ranks, in_flight in ((3, 2), (2, 3), (1, 4), (2, 1)):
                ctx.set_option("pipeline", in_flight)
                comm = rtx.Comm.loopback(ctx, n_ranks)
                try:
                    assert 1 <= rtx.LIB.rtx_comm_concurrency(comm.h) <= in_flight + 1 or in_flight == 1
                    frames = []
                    for k in range(6):
                        cam = (pc[0], cam2)[k & 1]
                        f = hip.malloc(cam.vsize * cam.hsize * 3, 0x33); i = hip.malloc(cam.vsize * cam.hsize * 4, 0x44)
                        bufs += [f, i]; frames.append((f, i, cam, wants[k & 1]))
                        comm.render_gathered(sc, cam, f.value, i.value, 5)
                    comm.wait()
                    for f, i, cam, want in frames:
                        assert np.array_equal(hip.to_host(f, cam.vsize * cam.hsize * 3, np.uint8).reshape(cam.vsize, cam.hsize, 3), want["rgb"]), (n_ranks, in_flight)
                        assert np.array_equal(hip.to_host(i, cam.vsize * cam.hsize * 4, np.int32).reshape(cam.vsize, cam.hsize), want["hit"]), (n_ranks, in_flight)
                finally:
                    comm.close()
                    ctx.set_option("pipeline", 3)
            ctx.poll(wait=True)
        finally:
            sc.close()
    finally:
        for b in bufs:
            hip.lib.hipFree(b)


def test_band_path_at_full_size_s5_8k_over_8_ranks(ctx):
    """C5 (BASELINE.json: 7680x4320 on 8 GPUs) through the product's N = 8 data path AT SIZE on the one GPU there is: a loopback
    communicator of 8 ranks renders every rank's compact band set (540 bands -> 68 per rank, 67.5 rounded up: the last rank's
    set ends in a padding band; 12.5 MB per rank, 99.5 MB gathered) where the all-gather would deliver it and the product's
    de-interleave kernel (long long indexing) assembles frame and hit ids.  Must equal rtx_render byte for byte."""
    import rtx
    hip = _Hip()
    w, h = 7680, 4320
    pw, pc = rtx.load_yaml(parity.scene_text("S5", w, h))
    sc = ctx.scene(pw)
    bufs = []
    comm = rtx.Comm.loopback(ctx, 8)
    try:
        assert rtx.LIB.rtx_band_rows_padded(h, 8) == 544
        ctx.set_option("host_sets", 1)
        want = sc.render(pc[0], 5)
        frame = hip.malloc(h * w * 3, 0x33); ids = hip.malloc(h * w * 4, 0x44)
        bufs += [frame, ids]
        for rep in range(2):                                          # asynchronous (render lanes), then with stats
            stats = comm.render_gathered(sc, pc[0], frame.value, ids.value, 5, stats=(rep == 1))
            comm.wait()
            got = hip.to_host(frame, h * w * 3, np.uint8).reshape(h, w, 3)
            assert np.array_equal(got, want["rgb"]), (rep, int((got != want["rgb"]).any(axis=2).sum()))
            assert np.array_equal(hip.to_host(ids, h * w * 4, np.int32).reshape(h, w), want["hit"]), rep
            if stats is not None:
                assert (stats.rays_primary, stats.rays_shadow, stats.rays_secondary) == \
                       (want["stats"].rays_primary, want["stats"].rays_shadow, want["stats"].rays_secondary)
            assert hip.lib.hipMemset(frame, 0x33, h * w * 3) == 0 and hip.lib.hipMemset(ids, 0x44, h * w * 4) == 0
        ctx.poll(wait=True)
    finally:
        ctx.set_option("host_sets", 0)
        comm.close()
        sc.close()
        for b in bufs:
            hip.lib.hipFree(b)



def test_multi_gpu_entry_points_with_a_one_rank_communicator(ctx, tmp_path):
    """The C-ABI multi-GPU render (rtx_comm_* / rtx_render_gathered / rtx_render_multi, `rtx --gpus N`): bands of rank r of N,
    ONE ncclAllGather (RCCL, opened by the library itself), de-interleave kernel.  The box has one GPU, so the communicator
    has one rank: the same calls, the gather degenerates to RCCL's own copy.  Three frames in a row (the two internal
    buffer sets alternate; sizes change, so the buffers regrow) must equal rtx_render byte for byte, ids included."""
    import ctypes as C
    import os
    import subprocess
    from PIL import Image
    import rtx
    hip = _Hip()
    comm = rtx.Comm(ctx, rtx.comm_unique_id(), 0, 1)
    assert rtx.LIB.rtx_comm_rank(comm.h) == 0 and rtx.LIB.rtx_comm_size(comm.h) == 1
    st = C.c_void_p()
    assert hip.lib.hipStreamCreate(C.byref(st)) == 0
    bufs = []
    try:
        for name, w, h in (("S4", 333, 171), ("S1", 64, 32), ("S5", 640, 360)):
            pw, pc = rtx.load_yaml(parity.scene_text(name, w, h))
            sc = ctx.scene(pw)
            try:
                want = sc.render(pc[0], 5)
                frame = hip.malloc(h * w * 3, 0x33); ids = hip.malloc(h * w * 4, 0x44)
                bufs += [frame, ids]
                stats = comm.render_gathered(sc, pc[0], frame.value, ids.value, 5, stream=st.value, stats=(name == "S4"))
                comm.wait()                                         # host waits for gather + de-interleave
                assert np.array_equal(hip.to_host(frame, h * w * 3, np.uint8).reshape(h, w, 3), want["rgb"]), name
                assert np.array_equal(hip.to_host(ids, h * w * 4, np.int32).reshape(h, w), want["hit"]), name
                if stats is not None:
                    assert stats.rays == want["stats"].rays
                # stream-level wait: a copy queued on the caller's stream behind rtx_comm_wait sees the finished frame
                frame2 = hip.malloc(h * w * 3, 0x55); bufs.append(frame2)
                comm.render_gathered(sc, pc[0], frame2.value, 0, 5, stream=st.value)
                comm.wait(st.value)
                assert hip.lib.hipStreamSynchronize(st) == 0
                assert np.array_equal(hip.to_host(frame2, h * w * 3, np.uint8).reshape(h, w, 3), want["rgb"]), name
            finally:
                sc.close()
    finally:
        comm.close()
        for b in bufs:
            hip.lib.hipFree(b)
        hip.lib.hipStreamDestroy(st)
    # one host thread, "all" GPUs of the box (ncclCommInitAll + grouped all-gather)
    comms = rtx.Comm.init_all([ctx])
    try:
        pw, pc = rtx.load_yaml(parity.scene_text("S4", 200, 100))
        sc = ctx.scene(pw)
        want = sc.render(pc[0], 5)
        got = rtx.render_multi(comms, [sc], pc[0], 5, want_stats=True)
        assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["hit"], want["hit"])
        assert got["stats"].rays == want["stats"].rays
        sc.close()
    finally:
        for c in comms:
            c.close()
    # the CLI: rtx <scene> --gpus 1 writes the same PNG as the single-GPU path
    text = parity.scene_text("S1", 96, 48)
    scene = tmp_path / "multi.yaml"
    scene.write_text(text)
    exe = os.path.join(os.path.dirname(rtx.LIB_PATH), "rtx")
    subprocess.run([exe, str(scene), "--gpus", "1"], check=True, timeout=200, cwd=str(tmp_path))
    pw, pc = rtx.load_yaml(text)
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "multi.png"))), ctx.render(pw, pc[0], 5)["rgb"])


def test_async_overflow_is_sticky_not_silent(ctx):
    """include/rtx.h: an asynchronous render cannot return a device-side RTX_E_CAPACITY itself; the condition must be
    sticky: refused by the next render call, returned and cleared by rtx_ctx_poll.  Provoked with the split guard: at
    max_depth 8 with split tiles forced and a guard of 1 wave iteration (debug option) every split wave gives up."""
    import ctypes as C
    import rtx
    hip = _Hip()
    pw, pc = rtx.load_yaml(parity.scene_text("S4", 320, 200))
    sc = ctx.scene(pw)
    buf = hip.malloc(200 * 320 * 3, 0)
    try:
        ctx.options(lpt=1, split=1, split_class=1, debug=2)         # debug=2: split guard of one iteration
        assert sc.render_device(pc[0], buf.value, 0, 5) is None     # queued; the launch trips the guard
        with pytest.raises(rtx.RtxError) as e:
            ctx.poll(wait=True)
        assert e.value.code == rtx.E_CAPACITY
        ctx.poll(wait=True)                                         # cleared
        sc.render_device(pc[0], buf.value, 0, 5)
        assert rtx.LIB.rtx_ctx_poll(ctx.h, 1) == rtx.E_CAPACITY     # still reported when nobody asked in between ...
        sc.render_device(pc[0], buf.value, 0, 5)
        hip.lib.hipDeviceSynchronize()
        with pytest.raises(rtx.RtxError):                           # ... and the next call refuses to queue more work
            sc.render_device(pc[0], buf.value, 0, 5)
        assert rtx.LIB.rtx_ctx_poll(ctx.h, 1) == rtx.E_CAPACITY
        with pytest.raises(rtx.RtxError) as e:                      # the synchronous path names the cause
            sc.render(pc[0], 5)
        assert e.value.code == rtx.E_CAPACITY and "guard" in str(e.value)
        ctx.options(debug=0)
        ok = sc.render(pc[0], 5)                                    # with the real guard the same frame renders
        ctx.options(lpt=-1, split=-1, split_class=7)
        assert np.array_equal(ok["rgb"], sc.render(pc[0], 5)["rgb"])
    finally:
        ctx.options(lpt=-1, split=-1, split_class=7, debug=0)
        rtx.LIB.rtx_ctx_poll(ctx.h, 1)
        sc.close()
        hip.lib.hipFree(buf)


def test_gathered_render_inside_a_pytorch_process(tmp_path):
    """bench.py's N > 1 configuration: PyTorch-ROCm initialises the GPU first (its own HIP runtime and RCCL are then the
    ones in the process), the library opens RCCL itself (the copy already loaded) and gathers into a torch tensor.  One
    rank here (one GPU); run in a child process so torch's runtime does not leak into the other tests."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
import torch
torch.cuda.init(); dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
import parity, rtx
ctx = rtx.Context(0)
comm = rtx.Comm(ctx, rtx.comm_unique_id(), 0, 1)
pw, pc = rtx.load_yaml(parity.scene_text("S4", 333, 171))
sc = ctx.scene(pw)
want = sc.render(pc[0], 5)
s = torch.cuda.Stream(device=dev)
frames = [torch.zeros(171 * 333 * 3, dtype=torch.uint8, device=dev) for _ in range(2)]
for k in range(4):
    comm.render_gathered(sc, pc[0], frames[k & 1].data_ptr(), 0, 5, 0, s.cuda_stream)
comm.wait()
for f in frames:
    assert np.array_equal(f.cpu().numpy().reshape(171, 333, 3), want["rgb"])
sc.close(); comm.close(); ctx.close()
print("ok")
'''
    import os
    env = dict(os.environ)
    here = os.path.dirname(os.path.abspath(__file__))
    env["PYTHONPATH"] = os.pathsep.join([here, os.path.join(here, "..", "ray-tracer_amd", "python"), os.path.join(here, "..", "tools")])
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


FAR = [("S4", 240, 136, {}), ("S3", 200, 112, dict(levels=2, per_leaf=6)), ("S5", 240, 136, {})]
FAR_VARIANTS = [dict(world_offset=(1.0e4, -2.5e3, 7.0e3)), dict(world_offset=(1.0e6, 1.0e6, -1.0e6)),
                dict(wrap=(1.0e-3, (0.0, 0.0, 0.0))), dict(wrap=(1.0e3, (5.0e3, -2.0e3, 7.0e3)))]


@pytest.mark.parametrize("name,w,h,kw", FAR, ids=[c[0] for c in FAR])
def test_scenes_far_from_the_origin_and_in_scaled_groups(ctx, name, w, h, kw):
    """The conservative f32 pre-tests (bounding spheres in rtx_api.hip: cull_sphere, CULL_KAPPA; the behind-the-origin test)
    must stay conservative when coordinates are large against the distances that matter: the whole scene, lights and
    camera moved by 1e4 and 1e6 (every root object's transform gets the translation), and the whole scene inside one root
    group scaled by 1e-3 / by 1e3 and moved.  Same checks as the scene parity test, against the oracle on the same worlds."""
    import ora
    import rtapi
    import rtx
    import yaml_scene
    text = parity.scene_text(name, w, h, **kw)
    for var in FAR_VARIANTS:
        pw, pcs = yaml_scene.load(text, rtx.API, **var)
        ow, ocs = yaml_scene.load(text, ora.API, **var)
        gpu = ctx.render(pw, pcs[0], 5, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        cpu = ora.render(ow, ocs[0], 5, threads=8)
        # (1e6 away from the origin one ulp of a coordinate is 1.2e-10, the size of the reference's EPSILON offsets
        # (world.rs:298-300): surface acne, i.e. tie pixels, is then expected — the exemption is bounded at 30 % there)
        far = max(abs(v) for v in var.get("world_offset", (0,))) >= 1e6
        r = parity.assert_parity(gpu, cpu, "%s %s" % (name, var), max_tie_frac=0.3 if far else 0.005)
        print(name, var, "tie pixels", r["tie_pixels"], "id mismatches incl. ties", r["id_mismatch_incl_ties"])
        assert (gpu["hit"] >= 0).mean() > 0.2, (name, var)          # the scene is in view (a plane inside the wrap group is clipped
                                                                    # to the group's box: bounds.rs:29-40 gives it no bounds)
        if r["tie_pixels"] == 0:
            gs, cs = gpu["stats"], cpu["stats"]
            assert (gs.rays_primary, gs.rays_shadow, gs.rays_secondary) == (cs.rays_primary, cs.rays_shadow, cs.rays_secondary), (name, var, r)
            ge, ce = gs.events(), cs.events()
            assert {k: ge[k] for k in rtapi.REF_EVENTS if k != "pow"} == {k: ce[k] for k in rtapi.REF_EVENTS if k != "pow"}, (name, var)
        fast = ctx.render(pw, pcs[0], 5)
        assert np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]), (name, var)


def test_scene_description_in_the_rust_shims_node_order(ctx):
    """N1: integration/rust/src/flatten.rs numbers the nodes shapes-first, then groups, then CSGs, with one material per
    shape; the C++ host mirror numbers them in creation order with the same materials.  The C ABI must not care: the
    description rebuilt in flatten.rs's order (tests/rust_order.py, from the world's canonical dump) through raw
    rtx_scene_create renders the same image, ids and counts as rtxh_render, on a scene file with groups and CSG and on
    the API-built torture scene (CSG inside groups, a group as a CSG operand, empty groups)."""
    import rtx
    import rust_order
    import torture
    worlds = []
    pw, pcs = rtx.load_yaml(parity.scene_text("S5", 200, 120))
    worlds.append((pw, pcs[0]))
    pw, pcs = rtx.load_yaml(parity.scene_text("S3", 160, 96, levels=2, per_leaf=5))
    worlds.append((pw, pcs[0]))
    worlds.append(torture.build(rtx.API))
    for k, (w, cam) in enumerate(worlds):
        want = ctx.render(w, cam, 5, flags=rtx.F_COUNT_EVENTS)
        desc, keep = rust_order.desc_in_rust_order(w)
        sc = rtx.Scene(ctx, desc=desc)
        try:
            got = sc.render(cam, 5, flags=rtx.F_COUNT_EVENTS)
        finally:
            sc.close()
        assert np.array_equal(got["rgb"], want["rgb"]) and np.array_equal(got["hit"], want["hit"]), k
        assert got["stats"].rays == want["stats"].rays and got["stats"].events() == want["stats"].events(), k
