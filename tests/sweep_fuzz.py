"""One-off randomized parity sweep (not collected by pytest): python tests/sweep_fuzz.py [n_seeds [first_seed]] on a GPU box.
Test infrastructure: compares the HIP path with the oracle on generated scene files (tests/fuzz_scenes.py)."""
import sys, os
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "ray-tracer_amd/python"), os.path.join(ROOT, "oracle")]
import numpy as np, fuzz_scenes, rtx, yaml_scene, ora, parity, rtapi
ctx = rtx.Context(0)
bad = skipped = ties = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng((int(sys.argv[2]) if len(sys.argv) > 2 else 1000) + seed)
    for i in range(150):
        w, h = int(rng.integers(8, 120)), int(rng.integers(8, 90))
        text = fuzz_scenes.make(rng, w, h)
        depth = (5, 0, 1, 2, 3, 8, 7, 4)[i % 8]
        pw, pcs = rtx.load_yaml(text)
        try:
            gpu = ctx.render(pw, pcs[0], depth, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        except rtx.RtxError as e:
            skipped += 1; continue
        ow, ocs = yaml_scene.load(text, ora.API)
        cpu = ora.render(ow, ocs[0], depth, threads=16)
        r = parity.compare(gpu, cpu)
        gs, cs = gpu["stats"], cpu["stats"]
        # (the images are tiny, so one pixel is already > 0.1 %: colours are compared away from the oracle's tie pixels,
        # where a different branch of the ray tree is legitimate)
        away = cpu["tie"] == 0
        diff = np.abs(gpu["rgb"].astype(np.int16) - cpu["rgb"].astype(np.int16)).max(axis=2)
        ok = r["id_mismatch"] == 0 and not (diff[away] > 1).any()
        if r["tie_pixels"] == 0:
            ok = ok and gs.rays == cs.rays
            ge, ce = gs.events(), cs.events()
            ok = ok and all(ge[k] == ce[k] for k in rtapi.REF_EVENTS if k != "pow")
        ctx.options(lpt=1, split=1, split_class=2)
        fast = ctx.render(pw, pcs[0], depth)
        ctx.options(lpt=-1, split=-1, split_class=7)
        ok = ok and np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]) and fast["stats"].rays == gs.rays
        ties += r["tie_pixels"]; tie_scenes = globals().get("tie_scenes", 0) + (1 if r["tie_pixels"] else 0); tie_mis = globals().get("tie_mis", 0) + r["id_mismatch_incl_ties"]
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "scene", i, "depth", depth, r, gs.rays, cs.rays, flush=True)
            open("gpurun_out/fuzz_bad_%d_%d.yaml" % (seed, i), "w").write(text)
    print("seed", seed, "done; bad", bad, "skipped", skipped, "tie pixels", ties, flush=True)
print("TOTAL bad", bad, "skipped", skipped, "scenes with ties", tie_scenes, "id mismatches inside tie pixels", tie_mis)
