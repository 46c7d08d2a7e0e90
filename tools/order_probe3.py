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

    m2, m3 = nmax(c[2]), nmax(c[3])
    for rep in range(3):
        go("exact, quantile classes", qcls(cost))
        go("d2 3x3 max, c>>1 (the product's)", m2 >> 1)
        go("d2 3x3 max, c-1 capped", m2 - 1)
        go("d2 3x3 max, thresholds 3,4,6,7,10,12,13", np.searchsorted([3, 4, 6, 7, 10, 12, 13], m2.ravel(), side="right"))
        go("d2 3x3 max, thresholds 3,4,6,7,8,10,12", np.searchsorted([3, 4, 6, 7, 8, 10, 12], m2.ravel(), side="right"))
        go_sorted("d2 3x3 max, one class per value (14)", m2)
        go_sorted("d3 3x3 max, one class per value (27)", m3)
        go("d3 3x3 max, c>>2", m3 >> 2)
