"""One-off randomized parity sweep over API-built scene graphs (not collected by pytest):
python tests/sweep_fuzz_api.py [seeds] on a GPU box.  Test infrastructure (uses the oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "ray-tracer_amd/python"), os.path.join(ROOT, "oracle")]
import numpy as np, fuzz_api, rtx, ora, parity, rtapi
ctx = rtx.Context(0)
bad = skipped = ties = tie_scenes = tie_mis = n = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    rng = np.random.default_rng((int(sys.argv[2]) if len(sys.argv) > 2 else 5000) + seed)
    for i in range(150):
        p = fuzz_api.plan(rng, int(rng.integers(8, 120)), int(rng.integers(8, 90)), big_csg=(i % 3 == 2))   # every third: general CSG kernels
        depth = (5, 0, 1, 2, 3, 8, 7, 4)[i % 8]
        pw, pc = fuzz_api.build(rtx.API, p)
        try:
            gpu = ctx.render(pw, pc, depth, flags=rtx.F_COUNT_EVENTS | rtx.F_NO_SHADOW_EARLY_OUT)
        except rtx.RtxError as e:
            if e.code not in (rtx.E_CAPACITY, rtx.E_UNSUPPORTED):
                print("ERROR seed", seed, "scene", i, e, flush=True); bad += 1
            skipped += 1; continue
        ow, oc = fuzz_api.build(ora.API, p)
        cpu = ora.render(ow, oc, depth, threads=16)
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
        fast = ctx.render(pw, pc, depth)
        ctx.options(lpt=-1, split=-1, split_class=7)
        ok = ok and np.array_equal(fast["rgb"], gpu["rgb"]) and np.array_equal(fast["hit"], gpu["hit"]) and fast["stats"].rays == gs.rays
        if i % 5 == 0:
            # band subsets (row-interleaved, plain and compact) and a batch of two cameras against single renders
            sc = ctx.scene(pw)
            rgb = np.zeros_like(gpu["rgb"]); hit = np.full_like(gpu["hit"], -7)
            G = 2 + i % 3
            for g in range(G):
                sc.render(pc, depth, bands=(g, G), out=(rgb, hit))
            ok = ok and np.array_equal(rgb, gpu["rgb"]) and np.array_equal(hit, gpu["hit"])
            nb = (pc.vsize + 7) // 8
            for g in range(G):
                c = sc.render(pc, depth, bands=(g, G), flags=rtx.F_BANDS_COMPACT)
                rows = np.concatenate([np.arange(b * 8, min(b * 8 + 8, pc.vsize)) for b in range(g, nb, G)] or [np.zeros(0, int)])
                ok = ok and np.array_equal(c["rgb"][:len(rows)], gpu["rgb"][rows])
            pc2 = rtx.API.camera(max(8, pc.hsize // 2), pc.vsize + 3, 0.9, rtx.API.view_transform([3, 4, -9], [0, 1, 0], [0, 1, 0]))
            batch = sc.render_batch([pc, pc2, pc], depth)
            one2 = sc.render(pc2, depth)
            ok = ok and np.array_equal(batch[0]["rgb"], gpu["rgb"]) and np.array_equal(batch[2]["hit"], gpu["hit"])
            ok = ok and np.array_equal(batch[1]["rgb"], one2["rgb"]) and np.array_equal(batch[1]["hit"], one2["hit"])
            ok = ok and batch[1]["stats"].rays == one2["stats"].rays
            sc.close()
        ties += r["tie_pixels"]; tie_scenes += 1 if r["tie_pixels"] else 0; tie_mis += r["id_mismatch_incl_ties"]; n += 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "scene", i, "depth", depth, r, gs.rays, cs.rays, flush=True)
    print("seed", seed, "done; scenes", n, "bad", bad, "skipped", skipped, "tie pixels", ties, flush=True)
print("TOTAL scenes", n, "bad", bad, "skipped", skipped, "scenes with ties", tie_scenes, "id mismatches inside tie pixels", tie_mis)
