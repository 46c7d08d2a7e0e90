#!/usr/bin/env python3
"""How much is a better tile order worth?  Renders one frame with the RTX_WAVE_TRACE build to get every tile's exact cost
(iterations of its slowest lane), then times the frame launch with tile lists made from those costs (experiment build
-DRTX_EXP_ORDER, lists from a file, no probe launch) against the product's probe order and against no order at all.
  make -C ray-tracer_amd librtx_xA.so XFLAGS_A=-DRTX_WAVE_TRACE librtx_xB.so XFLAGS_B=-DRTX_EXP_ORDER
usage: tools/order_probe.py [SCENE [W H]]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import numpy as np

name = sys.argv[1] if len(sys.argv) > 1 else "S4"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
OUT = os.path.join(ROOT, "gpurun_out", "order")
os.makedirs(OUT, exist_ok=True)
CLASSES = 8


def child(role):
    os.environ.setdefault("RTX_HOST_SETS", "1")
    import rtx, scenes
    ctx = rtx.Context(0)
    world, cams = rtx.load_yaml(scenes.make(name, width=W, height=H))
    sc = ctx.scene(world)
    if role == "trace":
        for _ in range(2):
            r = sc.render(cams[0], 5, flags=rtx.F_BANDS_COMPACT)
        hit = r["hit"].ravel()
        n = ((W + 7) // 8) * ((H + 7) // 8)
        t = hit[:3 * n].reshape(n, 3).astype(np.int64)
        np.save(os.path.join(OUT, "iters.npy"), (t[:, 2] & 0xffff).astype(np.int32))
        np.save(os.path.join(OUT, "dur.npy"), ((t[:, 1] & 0xffffffff) - (t[:, 0] & 0xffffffff)).astype(np.int64))
        return
    ms = [sc.render(cams[0], 5, flags=rtx.F_BANDS_COMPACT, want_hit=False)["stats"].kernel_ms for _ in range(12)]
    print("%-44s %.3f ms (min of 12, median %.3f)" % (role, min(ms), float(np.median(ms))), flush=True)


def run(role, lib, **env):
    e = dict(os.environ, RTX_LIB=lib, ORDER_PROBE_CHILD=role, **{k: str(v) for k, v in env.items()})
    subprocess.run([sys.executable, os.path.abspath(__file__), name, str(W), str(H)], env=e, check=True)


def write_lists(path, classes):
    """classes: list of CLASSES arrays of tile ids (class 7 = rendered first)"""
    n = sum(len(c) for c in classes)
    data = np.zeros(CLASSES + CLASSES * n, np.int32)
    for k, c in enumerate(classes):
        data[k] = len(c)
        data[CLASSES + k * n: CLASSES + k * n + len(c)] = c
    data.tofile(path)


if __name__ == "__main__":
    role = os.environ.get("ORDER_PROBE_CHILD")
    if role:
        child(role); sys.exit(0)
    n = ((W + 7) // 8) * ((H + 7) // 8)
    run("trace", "librtx_xA.so", RTX_LPT=0)
    iters = np.load(os.path.join(OUT, "iters.npy")); dur = np.load(os.path.join(OUT, "dur.npy"))
    mul = 7919
    while np.gcd(mul % n or n, n) != 1:
        mul += 1
    tile_of_block = (np.arange(n, dtype=np.int64) * (mul % n)) % n
    cost = np.zeros(n, np.int64); cost[tile_of_block] = iters
    print("%s %dx%d: %d tiles, iterations of the slowest lane: mean %.1f p50 %d p90 %d p99 %d max %d" %
          (name, W, H, n, cost.mean(), *np.percentile(cost, [50, 90, 99]).astype(int), cost.max()), flush=True)
    run("product: no order (RTX_LPT=0)", "librtx_hip.so", RTX_LPT=0)
    run("product: probe order (default)", "librtx_hip.so")
    empty = [np.zeros(0, np.int32)] * (CLASSES - 1)
    exact = np.argsort(-cost, kind="stable").astype(np.int32)
    f = os.path.join(OUT, "lists.bin")
    write_lists(f, empty + [exact])
    run("exact costs, fully sorted, no probe launch", "librtx_xB.so", RTX_EXP_ORDER_FILE=f)
    # the same information quantised to 8 classes (what the probe's lists can express), order inside a class as the probe leaves
    # it: by tile index
    edges = np.unique(np.percentile(cost, [50, 75, 87, 93, 96, 98, 99]))
    cls = np.searchsorted(edges, cost, side="right")
    write_lists(f, [np.flatnonzero(cls == k).astype(np.int32) for k in range(CLASSES)])
    run("exact costs in 8 quantile classes, no probe launch", "librtx_xB.so", RTX_EXP_ORDER_FILE=f)
    if os.environ.get("ORDER_PROBE_TRACE") == "1":      # needs librtx_xA.so built with BOTH -DRTX_WAVE_TRACE and -DRTX_EXP_ORDER
        for label, env in (("probe order", {}), ("exact 8 classes", {"RTX_EXP_ORDER_FILE": f})):
            print("---- wave trace, %s" % label, flush=True)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wave_trace.py"), name, str(W), str(H), "1"],
                           env=dict(os.environ, RTX_LIB="librtx_xA.so", **env), check=True)
    # exact costs, but only the top 10 % ordered; the rest in the product's permuted order
    top = exact[: n // 10]
    rest_mask = np.ones(n, bool); rest_mask[top] = False
    rest = tile_of_block[rest_mask[tile_of_block]].astype(np.int32)
    write_lists(f, empty[:-1] + [rest, top])
    run("top 10 % sorted first, rest permuted, no probe launch", "librtx_xB.so", RTX_EXP_ORDER_FILE=f)
