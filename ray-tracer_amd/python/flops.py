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


# L2 <-> fabric traffic of one trace_shade_kernel launch at the bench's default size on ONE GPU, from separate
# rocprofv3 --pmc passes (tools/pmc_passes.sh; summaries in profiles/r1e_*_pmc.txt, r1f for S4).  Units: the counters
# report KiB (checked against a 24 883 200-byte fill in the same trace: WRITE_SIZE 24300.0).  FETCH_SIZE on
# gfx950 tallies 128-B requests at 64 B for wide coalesced reads and is uncalibrated for other widths
# (MI355X_MICROARCH.md, HBM); this kernel's fabric reads are scratch re-loads (4-16 B per lane), so FETCH is
# given as counted and the doubled figure as the upper bound.  bench.py cannot collect counters itself; it
# quotes these numbers only for the exact workload they were measured on.
MEASURED_TRAFFIC_KIB = {          # workload: (FETCH_SIZE, WRITE_SIZE) per launch
    "S3": (63663, 688858),
    "S4": (2670007, 7179379),      # frame launch + tile-order probe launch (profiles/r1f_S4_pmc.txt)
    "S5": (427135, 8269739),
}


def measured_traffic_bytes(workload):
    """-> (bytes as counted, upper bound with FETCH doubled) or None"""
    t = MEASURED_TRAFFIC_KIB.get(workload)
    if t is None:
        return None
    return (t[0] + t[1]) * 1024, (2 * t[0] + t[1]) * 1024


# VALU pipe utilisation of the frame launch from the same committed PMC passes:
# SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
MEASURED_VALU_BUSY = {"S4": 0.71}
