#!/usr/bin/env python3
"""Which property of a tile order pays: exact costs or spatial coherence inside a class?  Times the frame launch of the
-DRTX_EXP_ORDER build (librtx_xB.so) with tile lists built offline from the per-pixel cost maps of tools/pixel_cost.py
(copied to tmp_order/).  usage: tools/order_probe2.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ray-tracer_amd", "python"), os.path.join(ROOT, "tools")]
import numpy as np
import order_probe as op
name, W, H = "S4", 3840, 2160
TY, TX = H // 8, W // 8
n = TY * TX
if __name__ == "__main__":
    if os.environ.get("ORDER_PROBE_CHILD"):
        op.child(os.environ["ORDER_PROBE_CHILD"]); sys.exit(0)
    d5 = np.load(os.path.join(ROOT, "tmp_order", "cost_S4_d5.npy")).astype(np.int32)
    d2 = np.load(os.path.join(ROOT, "tmp_order", "cost_S4_d2.npy")).astype(np.int32)
    cost = d5.reshape(TY, 8, TX, 8).max(axis=(1, 3)).ravel()
    c2 = d2[4::8, 4::8]
    p = np.pad(c2, 1, mode="edge"); m3 = c2.copy()
    for dy in range(3):
        for dx in range(3):
            m3 = np.maximum(m3, p[dy:dy + TY, dx:dx + TX])
    probe_cls = np.minimum(m3 >> 1, 7).ravel()
    edges = np.unique(np.percentile(cost, [50, 75, 87, 93, 96, 98, 99]))
    exact_cls = np.searchsorted(edges, cost, side="right")
    perm = (np.arange(n, dtype=np.int64) * 7919) % n
    rank_of = np.empty(n, np.int64); rank_of[perm] = np.arange(n)      # position of a tile in the permuted order
    # Morton-like coherent order: 8x8-tile superblocks in row-major order, row-major inside
    ty, tx = np.divmod(np.arange(n), TX)
    sb_key = ((ty // 8) * ((TX + 7) // 8) + tx // 8) * 64 + (ty % 8) * 8 + tx % 8
    f = os.path.join(op.OUT, "lists.bin")

    def lists(cls, within):
        return [np.array(sorted(np.flatnonzero(cls == k), key=None), np.int32)[np.argsort(within[np.flatnonzero(cls == k)], kind="stable")] for k in range(8)]

    def go(label, classes):
        op.write_lists(f, classes)
        e = dict(os.environ, RTX_LIB="librtx_xB.so", ORDER_PROBE_CHILD=label, RTX_EXP_ORDER_FILE=f)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=True)

    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RTX_LIB="librtx_hip.so", ORDER_PROBE_CHILD="product: probe order (default)"), check=True)
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, RTX_LIB="librtx_hip.so", RTX_LPT="0", ORDER_PROBE_CHILD="product: no order (permuted)"), check=True)
    one = np.zeros(n, np.int64)
    go("one class, row-major", lists(one, np.arange(n)))
    go("one class, permuted", lists(one, rank_of))
    go("one class, superblock order", lists(one, sb_key))
    go("exact classes, row-major inside", lists(exact_cls, np.arange(n)))
    go("exact classes, permuted inside", lists(exact_cls, rank_of))
    go("exact classes, superblock order inside", lists(exact_cls, sb_key))
    go("probe classes (offline), row-major inside", lists(probe_cls, np.arange(n)))
    go("probe classes (offline), permuted inside", lists(probe_cls, rank_of))
    go("probe classes (offline), superblock inside", lists(probe_cls, sb_key))
    print("class sizes: exact", np.bincount(exact_cls, minlength=8), "probe", np.bincount(probe_cls, minlength=8))
