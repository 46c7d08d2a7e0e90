#!/usr/bin/env python3
"""Collects the PMC passes of tools/profile_round.sh into profiles-style JSON:
usage: pmc_counters.py TAG GIT_REV WORKLOAD... > <tag>_counters.json
Reads gpurun_out/pmc_<TAG>_<WL>_<set>/**/*.db (rocprofv3 rocpd).  Per workload: "frame" = per-frame sums over the
launches one frame consists of (frame launch + tile-order probe + split launch when used), "main" = the frame launch
alone; counters are averages per dispatch of the non-counting kernel instantiations."""
import glob
import json
import sqlite3
import sys

SETS = ["sq", "sq2", "sq3", "tcc", "fetchsz", "write"]
# counters of the "sq3" pass (lane utilisation: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), both from ONE pass) are stored
# under "sq3:<name>": some of them ride in the "sq" pass as well and the counters of a ratio must come from one pass
PREFIXED = {"sq3"}


def kernels_of(db):
    cur = sqlite3.connect(db).cursor()
    times = {n: (c, a) for n, c, a in cur.execute("select name, count(*), avg(end-start) from kernels group by name")}
    ctr = {}
    try:
        for k, c, avg in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by 1, 2"):
            ctr.setdefault(k, {})[c] = avg
    except sqlite3.Error:
        pass
    return times, ctr


def targs(name):
    """template arguments of trace_shade_kernel<...> -> [CSG, COUNT, PROBE, SPLIT] booleans"""
    a = name[name.index("<") + 1:name.index(">")].split(",")
    a = [x.strip() == "true" for x in a]
    return a + [False] * (4 - len(a))


def main():
    tag, rev, wls = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = {"tag": tag, "git_rev": rev, "workloads": {}}
    for wl in wls:
        frame, main_, kms, names, kms_by_set = {}, {}, None, {}, {}
        for s in SETS:
            dbs = glob.glob("gpurun_out/pmc_%s_%s_%s/**/*.db" % (tag, wl, s), recursive=True)
            if not dbs:
                continue
            times, ctr = kernels_of(dbs[0])
            ts = {n: v for n, v in times.items() if "trace_shade_kernel" in n and not targs(n)[1]}
            if not ts:
                continue
            mains = [n for n in ts if not targs(n)[2] and not targs(n)[3]]
            main_name = max(mains, key=lambda n: ts[n][0])
            n_frames = ts[main_name][0]
            kms_by_set[s] = sum(c * a for n, (c, a) in ts.items()) / n_frames / 1e6
            if s == "sq":
                kms = kms_by_set[s]
                names = {n[:60]: c for n, (c, a) in ts.items()}
            for n in ts:
                for c, avg in ctr.get(n, {}).items():
                    if s in PREFIXED:
                        c = s + ":" + c
                    per_frame = avg * ts[n][0] / n_frames          # launches of this instantiation per frame
                    frame[c] = frame.get(c, 0.0) + per_frame
                    if n == main_name:
                        main_[c] = avg
        if frame:
            out["workloads"][wl] = {"kernel_ms": kms, "kernel_ms_by_pass": kms_by_set, "launches": names, "frame": frame, "main": main_}
    # registers / spills of the kernel instantiations at this revision: tools/kres.sh > profiles/<tag>_kres.txt (run where hipcc
    # cross-compiles, before the profiling call), one line per instantiation
    import os
    import re
    kres = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_kres.txt" % tag)
    if os.path.exists(kres):
        out["kres"] = {}
        for line in open(kres):
            m = re.match(r"(CSG=\d COUNT=\d PROBE=\d SPLIT=\d(?: HCTX=\d)?) (.*)", line.strip())
            if m:
                out["kres"][m.group(1)] = {k.strip().rsplit(" ", 1)[0]: int(k.strip().rsplit(" ", 1)[1]) for k in m.group(2).split("|")}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
