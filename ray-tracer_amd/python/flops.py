"""Algorithmic f64 flops of one frame from event counts (SURVEY.md §8d constants).

One flop per f64 add/sub/mul/div/sqrt/compare/min/max; precomputed inverses assumed.
The same formula is applied to the oracle's counters (reference-defined work) and to the
kernel's own counters (work actually executed, fewer shadow-ray visits with early-out).
"""
X_RAY = 36        # ray into node space: point 21 + vector 15
SPH = 30
PLN = 4
BOX = 23          # cube test or bounds test
CSGF = 4          # per merged CSG entry
HITPREP = 38      # position, flip, over/under, reflect
NRM = 45          # per hierarchy level: point xform 21 + normal xform 15 + normalize 9
PAT_LEVEL = 21    # per light and level: world_to_object for the pattern
PAT_FIXED = 26    # per light: pattern xform 21 + lookup 5
PHONG = 75
POW = 1
SHADOWSETUP = 17
REFR = 30
SCHLICK = 20
RAYGEN = 40
QUANT = 12


def frame_flops(ev, rays_primary, rays_shadow, n_lights):
    """ev: dict from Stats.events(); returns algorithmic flops."""
    f = 0
    f += X_RAY * ev["xform"]
    f += SPH * ev["sphere"] + PLN * ev["plane"] + BOX * (ev["cube"] + ev["bounds"])
    f += CSGF * ev["csg_entries"]
    f += HITPREP * ev["hit"] + NRM * ev["levels"]
    f += n_lights * (PAT_LEVEL * ev["levels"] + PAT_FIXED * ev["hit"])
    f += PHONG * ev["light"] + POW * ev["pow"]
    f += SHADOWSETUP * rays_shadow
    f += REFR * ev["refr"] + SCHLICK * ev["schlick"]
    f += (RAYGEN + QUANT) * rays_primary
    return f


def executed_flops(ev, rays_primary, rays_shadow, n_lights):
    """f64 flops the kernel actually issued: the conservative f32 bounding-sphere pre-test (rtx_kernel.hip
    cull_miss, not counted here: it is f32) removes most exact node tests, so this is <= frame_flops()."""
    f = frame_flops(ev, rays_primary, rays_shadow, n_lights)
    f -= X_RAY * ev["xform"] + SPH * ev["sphere"] + BOX * (ev["cube"] + ev["bounds"])
    f += X_RAY * ev["x_xform"] + SPH * ev["x_sphere"] + BOX * ev["x_box"]
    return f


FP64_VECTOR_PEAK_TFLOPS = 78.6     # 256 CU x 64 lanes x 2 (FMA) x 2.4 GHz (MI355X datasheet FP64 vector)
FP64_UNFUSED_PEAK_TFLOPS = 39.3    # one flop per lane-issue: the ceiling for parity code (no a*b+c fusion)


# ---- PMC record of the profiled revision (tools/profile_round.sh -> tools/pmc_counters.py -> profiles/<tag>_counters.json).
# bench.py cannot collect counters itself; it quotes the newest record, with the git revision and the kernel time the
# passes were taken at, and only for the exact workload and size they were measured on.  Units: FETCH_SIZE / WRITE_SIZE
# report KiB (checked against a 24 883 200-byte fill in the same trace: WRITE_SIZE 24300.0).  FETCH_SIZE on gfx950
# tallies 128-B requests at 64 B for wide coalesced reads and is uncalibrated for other widths (MI355X_MICROARCH.md, HBM);
# this kernel's fabric reads are scratch re-loads (4-16 B per lane), so FETCH is given as counted.
import glob
import json
import os

_PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles")
N_SIMD = 1024          # 256 CUs x 4
N_XCD = 8


def newest_counter_file():
    files = sorted(glob.glob(os.path.join(_PROFILES, "*_counters.json")))
    return files[-1] if files else None


def pmc_counters(workload, path=None):
    """-> derived figures of the newest PMC record for `workload`, or None.
    valu_issue_frac = SQ_INSTS_VALU / (kernel cycles x 1024 SIMDs / 4 cycles per wave64 instruction)
    valu_busy_frac  = SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024)
    resident_waves  = SQ_WAVE_CYCLES x 4 / kernel cycles          (kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs)"""
    path = path or newest_counter_file()
    if not path:
        return None
    rec = json.load(open(path))
    w = rec.get("workloads", {}).get(workload)
    if not w:
        return None
    c = w["frame"]                                   # per frame: frame launch (+ probe / split launches when used)
    cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
    # lane utilisation INSIDE issued vector instructions (exec-mask): both counters from the one "sq3" pass (older records:
    # the two counters sat in different passes, whose kernel times differ by a few per cent)
    if "sq3:SQ_THREAD_CYCLES_VALU" in c:
        lane_util = c["sq3:SQ_THREAD_CYCLES_VALU"] / (64.0 * c["sq3:SQ_ACTIVE_INST_VALU"])
    elif "SQ_THREAD_CYCLES_VALU" in c:
        lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    else:
        lane_util = None
    by_pass = w.get("kernel_ms_by_pass", {})
    ms_f, ms_w = by_pass.get("fetchsz", w.get("kernel_ms")), by_pass.get("write", w.get("kernel_ms"))
    hbm_gbps = (c["FETCH_SIZE"] * 1024 / (ms_f * 1e-3) + c["WRITE_SIZE"] * 1024 / (ms_w * 1e-3)) / 1e9 if ms_f and ms_w else None
    # the instantiation the timed launch of this workload runs (S5: bounded-CSG kernel; S2 / S4: hit context in LDS, S3 / S5: in registers)
    key = "CSG=%d COUNT=0 PROBE=0 SPLIT=0" % (1 if workload == "S5" else 0)
    kres = rec.get("kres", {}).get(key + " HCTX=%d" % (1 if workload in ("S2", "S4") else 0), rec.get("kres", {}).get(key, {}))
    return {"lane_util": lane_util, "hbm_gbps": hbm_gbps,
            "spilled_vgprs": kres.get("VGPRs Spill"), "spilled_sgprs": kres.get("SGPRs Spill"),"file": os.path.basename(path), "git_rev": rec.get("git_rev"), "kernel_ms": w.get("kernel_ms"),
            "traffic_bytes": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "fetch_bytes": int(c["FETCH_SIZE"] * 1024), "write_bytes": int(c["WRITE_SIZE"] * 1024),
            "valu_issue_frac": c["SQ_INSTS_VALU"] / (cyc * N_SIMD / 4.0),
            "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD),
            "resident_waves": c["SQ_WAVE_CYCLES"] * 4.0 / cyc}
