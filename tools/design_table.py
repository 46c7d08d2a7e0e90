#!/usr/bin/env python3
"""usage: tools/design_table.py TAG  ->  the markdown table of DESIGN.md section 5 from profiles/<TAG>_bench_S*.json and
profiles/<TAG>_counters.json (the bench lines and the PMC passes of one tools/profile_round.sh run)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cnt = json.load(open(os.path.join(ROOT, "profiles", tag + "_counters.json")))
print("`profiles/%s_*`, library at git %s:\n" % (tag, cnt["git_rev"]))
print("| workload | rays/frame | kernel ms | Mrays/s | algorithmic TFLOP/s (frac of 78.6) | executed f64 TFLOP/s | VALU issue | resident waves | fabric read / written per frame |")
print("|---|---|---|---|---|---|---|---|---|")
for wl in ("S2", "S3", "S4", "S5"):
    b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, wl))))
    r = b["roofline"]
    c = cnt["workloads"].get(wl, {}).get("frame")
    if c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        issue = "%.2f" % (c["SQ_INSTS_VALU"] / (cyc * 1024 / 4))
        res = "%d" % round(c["SQ_WAVE_CYCLES"] * 4 / cyc)
        traffic = "%.2f / %.2f GB" % (c["FETCH_SIZE"] * 1024 / 1e9, c["WRITE_SIZE"] * 1024 / 1e9)
    else:
        issue = res = traffic = "-"
    print("| %s | %.2f M | %.2f | %d | %.1f (%.2f) | %.1f | %s | %s | %s |" % (
        b["config"]["workload"].split(",")[0].replace("synthetic ", ""), b["config"]["rays_per_frame"] / 1e6, r["kernel_ms"],
        round(b["value"]), r["achieved"], r["frac"], r["executed"]["f64_tflops"], issue, res, traffic))
