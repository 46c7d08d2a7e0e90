#!/usr/bin/env python3
"""Candidate tile-cost estimators, timed on the GPU: tile lists built offline from the per-pixel cost maps (tmp_order/, from
tools/pixel_cost.py) and fed to the -DRTX_EXP_ORDER build (librtx_xB.so, no probe launch).  usage: tools/order_probe3.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import numpy as np
import order_probe as op
name, W, H = "S4", 3840, 2160
TY, TX = H // 8, W // 8
n = TY * TX


def nmax(a, k=1):
    p = np.pad(a, k, mode="edge"); out = a.copy()
    for dy in range(2 * k + 1):
        for dx in range(2 * k + 1):
            out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def qcls(est, qs=(50, 75, 87, 93, 96, 98, 99)):
    return np.searchsorted(np.unique(np.percentile(est, qs)), est, side="right")


if __name__ == "__main__":
    if os.environ.get("ORDER_PROBE_CHILD"):
        op.child(os.environ["ORDER_PROBE_CHILD"]); sys.exit(0)
    d = {k: np.load(os.path.join(ROOT, "tmp_order", "cost_S4_d%d.npy" % k)).astype(np.int32) for k in (1, 2, 3, 5)}
    cost = d[5].reshape(TY, 8, TX, 8).max(axis=(1, 3)).ravel()
    c = {k: d[k][4::8, 4::8] for k in d}
    f = os.path.join(op.OUT, "lists.bin")

    def go(label, cls):
        cls = np.minimum(np.asarray(cls).ravel(), 7)
        op.write_lists(f, [np.flatnonzero(cls == k).astype(np.int32) for k in range(8)])
        # rank correlation with the exact cost, for the record
        e = dict(os.environ, RTX_LIB="librtx_xB.so", ORDER_PROBE_CHILD=label, RTX_EXP_ORDER_FILE=f)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=True)

    def go_sorted(label, est):
        est = np.asarray(est).ravel()
        order = np.argsort(-est, kind="stable").astype(np.int32)
        op.write_lists(f, [np.zeros(0, np.int32)] * 7 + [order])
        e = dict(os.environ, RTX_LIB="librtx_xB.so", ORDER_PROBE_CHILD=label, RTX_EXP_ORDER_FILE=f)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=True)

    def samples(k, offs):
        return np.maximum.reduce([d[k][oy::8, ox::8] for oy in offs for ox in offs])

    m2 = nmax(c[2])
    s4_2 = samples(2, (2, 6)); s16_2 = samples(2, (1, 3, 5, 7)); all_2 = d[2].reshape(TY, 8, TX, 8).max(axis=(1, 3))
    all_3 = d[3].reshape(TY, 8, TX, 8).max(axis=(1, 3))
    for rep in range(2):
        go("exact, quantile classes", qcls(cost))
        go("d2 centre, 3x3 max, c>>1 (the product's)", m2 >> 1)
        go("d2 4 samples, 3x3 max, c>>1", nmax(s4_2) >> 1)
        go("d2 4 samples, own tile, c>>1", s4_2 >> 1)
        go("d2 16 samples, own tile, c>>1", s16_2 >> 1)
        go("d2 all 64 pixels, own tile, c>>1", all_2 >> 1)
        go("d2 all 64 pixels, 3x3 max, c>>1", nmax(all_2) >> 1)
        go("d3 all 64 pixels, own tile, c>>2", all_3 >> 2)
    # where the product's estimate is wrong: exact cost of the tiles it files under its cheap classes
    for k in range(1, 4):
        sel = (np.minimum(m2 >> 1, 7) == k).ravel()
        print("product class %d: %d tiles, exact cost p50 %d p99 %d max %d; tiles with cost >= 20: %d" %
              (k, sel.sum(), *np.percentile(cost[sel], [50, 99]).astype(int), cost[sel].max(), int((cost[sel] >= 20).sum())))
