#!/usr/bin/env python3
"""Offline study of tile-cost estimators: list-scheduling simulation of one frame launch (4096 wave slots, waves slow each
other down when the chip is full) on the per-pixel cost maps of tools/pixel_cost.py.  usage: tools/order_sim.py [SCENE]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "order")
name = sys.argv[1] if len(sys.argv) > 1 else "S4"
maps = {d: np.load(os.path.join(OUT, "cost_%s_d%d.npy" % (name, d))).astype(np.int32) for d in (5, 1, 2, 3)}
H, W = maps[5].shape
TY, TX = (H + 7) // 8, (W + 7) // 8
n = TY * TX


def tiles(m, f):
    p = np.zeros((TY * 8, TX * 8), m.dtype); p[:H, :W] = m
    return f(p.reshape(TY, 8, TX, 8).transpose(0, 2, 1, 3).reshape(TY, TX, 64), axis=2)


true_cost = tiles(maps[5], np.max).ravel()
SLOTS = 4096
R4 = 1 / 10.6          # iterations per us of a wave when all 4096 slots are taken (tools/wave_trace.py)
R1 = float(os.environ.get("SIM_R1", "2.2")) * R4   # ... of a wave with a SIMD to itself


def sim(order, cost=true_cost, dt=4.0):
    q = cost[order].astype(np.float64)
    left = np.zeros(0); nxt = 0; t = 0.0
    while nxt < len(q) or len(left):
        free = SLOTS - len(left)
        if free > 0 and nxt < len(q):
            left = np.concatenate([left, q[nxt:nxt + free]]); nxt += free
        # SIMD sharing: 1024 SIMDs x 4 slots; the per-wave rate falls from R1 (alone) to R4 (four per SIMD)
        per_simd = max(len(left) / 1024.0, 1.0)
        rate = R1 / (1 + (per_simd - 1) * (R1 / R4 - 1) / 3)
        left = left - rate * dt
        left = left[left > 0]
        t += dt
    return t / 1000.0


def classes_to_order(cls, within=None):
    """most expensive class first; inside a class by `within` (default: tile index, as the probe's appends roughly are)"""
    key = -cls.astype(np.int64) * (n + 1) + (np.arange(n) if within is None else within)
    return np.argsort(key, kind="stable")


def max3(a, k=1):
    p = np.pad(a, k, mode="edge"); out = a.copy()
    for dy in range(2 * k + 1):
        for dx in range(2 * k + 1):
            out = np.maximum(out, p[dy:dy + a.shape[0], dx:dx + a.shape[1]])
    return out


def centre(m, oy=4, ox=4):
    p = np.zeros((TY * 8, TX * 8), m.dtype); p[:H, :W] = m
    return p[oy::8, ox::8]


def quantile_classes(est, qs=(50, 75, 87, 93, 96, 98, 99)):
    edges = np.unique(np.percentile(est, qs))
    return np.searchsorted(edges, est, side="right")


perm = (np.arange(n, dtype=np.int64) * 7919) % n
rows = []
rows.append(("no order (permuted)", sim(perm)))
rows.append(("exact, sorted", sim(np.argsort(-true_cost, kind="stable"))))
rows.append(("exact, 8 quantile classes", sim(classes_to_order(quantile_classes(true_cost)))))
c2 = centre(maps[2])
probe = np.minimum(max3(c2) >> 1, 7).ravel()
rows.append(("product probe: centre d2, 3x3 max, c>>1", sim(classes_to_order(probe))))
rows.append(("centre d2, 3x3 max, quantile classes", sim(classes_to_order(quantile_classes(max3(c2).ravel())))))
rows.append(("centre d2, no neighbourhood, c>>1", sim(classes_to_order(np.minimum(c2 >> 1, 7).ravel()))))
rows.append(("centre d2, 5x5 max, c>>1", sim(classes_to_order(np.minimum(max3(c2, 2) >> 1, 7).ravel()))))
for d in (1, 3):
    cd = centre(maps[d])
    rows.append(("centre d%d, 3x3 max, quantile classes" % d, sim(classes_to_order(quantile_classes(max3(cd).ravel())))))
# 4 samples per tile (quadrant centres), max
for d in (2, 3):
    s4 = np.maximum.reduce([centre(maps[d], oy, ox) for oy in (2, 6) for ox in (2, 6)])
    rows.append(("4 samples d%d, own tile only, quantile classes" % d, sim(classes_to_order(quantile_classes(s4.ravel())))))
    rows.append(("4 samples d%d, 3x3 max, quantile classes" % d, sim(classes_to_order(quantile_classes(max3(s4).ravel())))))
# all 64 pixels at reduced depth (upper bound of what a reduced-depth probe can know)
for d in (1, 2, 3):
    full = tiles(maps[d], np.max).ravel()
    rows.append(("all 64 pixels d%d (bound), quantile classes" % d, sim(classes_to_order(quantile_classes(full)))))
# the full-depth cost of the centre pixel only
c5 = centre(maps[5])
rows.append(("centre d5, 3x3 max, quantile classes", sim(classes_to_order(quantile_classes(max3(c5).ravel())))))
rows.append(("centre d5, own tile, quantile classes", sim(classes_to_order(quantile_classes(c5.ravel())))))
s45 = np.maximum.reduce([centre(maps[5], oy, ox) for oy in (2, 6) for ox in (2, 6)])
rows.append(("4 samples d5, own tile, quantile classes", sim(classes_to_order(quantile_classes(s45.ravel())))))
rows.append(("4 samples d5, 3x3 max, quantile classes", sim(classes_to_order(quantile_classes(max3(s45).ravel())))))
print("%s %dx%d, %d tiles; total iterations %d -> %.3f ms at full occupancy; R1 = %.1f x R4" %
      (name, W, H, n, true_cost.sum(), true_cost.sum() / (SLOTS * R4) / 1000, R1 / R4))
for k, v in rows:
    print("  %-52s %.3f ms" % (k, v))
