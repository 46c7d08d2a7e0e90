"""One-off randomized sweep over the WAYS of rendering (not collected by pytest): python tests/sweep_entry_points.py [seeds [first [max_width max_height]]]
on a GPU box.  Every scheduling feature of the library — tile lists from the probe or from measured costs, time-ordered tiles,
split tiles, band sets on two lanes, the caller's stream, the no-alloc entry point, batches on four lanes, the band path of a
loopback communicator with 2..8 ranks and frames in flight — must leave image, hit ids and ray counts exactly as the plain
render (one launch, no tile lists) has them.  GPU against GPU, byte for byte; the oracle is not involved."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "ray-tracer_amd/python")]
import numpy as np, fuzz_api, fuzz_scenes, rtx

DEFAULTS = dict(lpt=-1, split=-1, split_class=7, host_sets=0, probe_ahead=-1, cost_feedback=-1, time_lpt=-1, hctx=-1, pipeline=3)


def sweep(ctx, hip, n_seeds, first_seed, max_w=640, max_h=400, per_seed=60, log=print):
    """-> (scenes rendered, mismatches); `hip` = the ctypes view of the HIP runtime of tests/test_gpu_parity.py (_Hip)"""
    bad = n = 0
    d_rgb = hip.malloc(max_w * max_h * 3); d_hit = hip.malloc(max_w * max_h * 4)

    def same(a, want, what, tag):
        nonlocal bad
        ok = np.array_equal(a["rgb"], want["rgb"]) and (a.get("hit") is None or np.array_equal(a["hit"], want["hit"]))
        if ok and a.get("stats") is not None:
            s, w = a["stats"], want["stats"]
            ok = (s.rays_primary, s.rays_shadow, s.rays_secondary) == (w.rays_primary, w.rays_shadow, w.rays_secondary)
        if not ok:
            bad += 1
            log("MISMATCH %s %s" % (tag, what))
        return ok

    def dev_image(W, H):
        return dict(rgb=hip.to_host(d_rgb, H * W * 3, np.uint8).reshape(H, W, 3), hit=hip.to_host(d_hit, H * W * 4, np.int32).reshape(H, W))

    try:
        for seed in range(n_seeds):
            rng = np.random.default_rng(first_seed + seed)
            for i in range(per_seed):
                W, H = int(rng.integers(9, max_w)), int(rng.integers(9, max_h))
                depth = (5, 0, 1, 2, 3, 8, 7, 4)[i % 8]
                if i % 2:
                    p = fuzz_api.plan(rng, W, H, big_csg=(i % 6 == 5))
                    pw, cam = fuzz_api.build(rtx.API, p)
                else:
                    pw, cams = rtx.load_yaml(fuzz_scenes.make(rng, W, H)); cam = cams[0]
                tag = "seed %d scene %d %dx%d depth %d" % (first_seed + seed, i, W, H, depth)
                try:
                    sc = ctx.scene(pw)
                except rtx.RtxError as e:
                    log("SKIP %s %s" % (tag, e)); continue
                n += 1
                try:
                    ctx.options(**DEFAULTS); ctx.options(lpt=0, host_sets=1, cost_feedback=0, time_lpt=0)
                    want = sc.render(cam, depth)
                    # tile lists: probe first, then the measured costs (three frames of one view), with and without split tiles
                    ctx.options(**DEFAULTS); ctx.options(lpt=1, host_sets=1, split=int(rng.integers(0, 2)), split_class=int(rng.integers(1, 8)),
                                                          probe_ahead=int(rng.integers(-1, 2)), hctx=int(rng.integers(-1, 2)))
                    for k in range(3):
                        same(sc.render(cam, depth), want, "lists frame %d" % k, tag)
                    # time-ordered tiles (scenes without branching materials take it, the others ignore the option)
                    ctx.options(**DEFAULTS); ctx.options(host_sets=1, time_lpt=1)
                    for k in range(3):
                        same(sc.render(cam, depth), want, "time order frame %d" % k, tag)
                    # band sets on two lanes
                    ctx.options(**DEFAULTS); ctx.options(host_sets=int(rng.integers(2, 7)), lpt=int(rng.integers(-1, 2)), time_lpt=int(rng.integers(-1, 2)))
                    for k in range(2):
                        same(sc.render(cam, depth), want, "band sets frame %d" % k, tag)
                    ctx.options(**DEFAULTS)
                    # the caller's buffers: default knobs, frames queued back to back, then the no-alloc entry point
                    for k in range(3):
                        sc.render_device(cam, d_rgb.value, d_hit.value, depth)
                    ctx.poll(wait=True)
                    same(dev_image(W, H), want, "render_device x3", tag)
                    hip.lib.hipMemset(d_rgb, 0x5a, W * H * 3)
                    if ((W + 7) // 8) * ((H + 7) // 8) <= 131072:
                        sc.render_device_noalloc(cam, d_rgb.value, d_hit.value, depth)
                        ctx.poll(wait=True)
                        same(dev_image(W, H), want, "no-alloc", tag)
                    for one in sc.render_batch([cam] * int(rng.integers(1, 6)), depth):
                        same(one, want, "batch", tag)
                    # the band path: 2..8 ranks on this GPU, frames in flight
                    ranks = int(rng.integers(2, 9))
                    ctx.options(pipeline=int(rng.integers(1, 4)), lpt=int(rng.integers(-1, 2)))
                    comm = rtx.Comm.loopback(ctx, ranks)
                    try:
                        for k in range(4):
                            comm.render_gathered(sc, cam, d_rgb.value, d_hit.value, depth)
                        comm.wait()
                        same(dev_image(W, H), want, "gathered over %d ranks" % ranks, tag)
                    finally:
                        comm.close()
                except rtx.RtxError as e:
                    bad += 1
                    log("ERROR %s %s" % (tag, e))
                finally:
                    ctx.options(**DEFAULTS)
                    sc.close()
            log("seed %d done; scenes %d mismatches %d" % (first_seed + seed, n, bad))
    finally:
        hip.lib.hipFree(d_rgb); hip.lib.hipFree(d_hit)
    return n, bad


if __name__ == "__main__":
    from test_gpu_parity import _Hip
    a = sys.argv
    n, bad = sweep(rtx.Context(0), _Hip(), int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 8000,
                   int(a[3]) if len(a) > 3 else 640, int(a[4]) if len(a) > 4 else 400, log=lambda m: print(m, flush=True))
    print("TOTAL scenes %d mismatches %d" % (n, bad))
    sys.exit(1 if bad else 0)
