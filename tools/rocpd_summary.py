#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel time stats and, if present, PMC counters
per kernel (averaged over dispatches).  Usage: rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("== %s" % path)
        rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        print("%-60s %8s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
        for name, n, tot, avg, mn, mx in rows:
            print("%-60s %8d %14d %14.0f %14d %14d %6.2f%%" % (name[:60], n, tot, avg, mn, mx, 100.0 * tot / total))
        # the same per launch SIZE (a whole frame, a band set and one rank's share are the same kernel), with the number of
        # dispatches that ran beside another dispatch of the table (frames in flight on several streams: such a dispatch's
        # duration includes its wait for wave slots, so only the rows with "beside 0" are per-launch kernel times)
        try:
            ds = cur.execute("select name, grid_x, start, end from kernels").fetchall()
            groups = {}
            for name, gx, st, en in ds:
                beside = sum(1 for n2, g2, s2, e2 in ds if (n2, g2, s2, e2) != (name, gx, st, en) and s2 < en and e2 > st and e2 - s2 > 20000)
                g = groups.setdefault((name, gx), [0, 0, 0, 1 << 62, 0])
                g[0] += 1; g[1] += en - st; g[2] += 1 if beside else 0; g[3] = min(g[3], en - st); g[4] = max(g[4], en - st)
            print("%-60s %10s %8s %14s %14s %14s %8s" % ("kernel, by launch size", "grid_x", "calls", "avg_ns", "min_ns", "max_ns", "beside"))
            for (name, gx), g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
                if "trace_shade" in name or "classify" in name:
                    print("%-60s %10d %8d %14.0f %14d %14d %8d" % (name[:60], gx, g[0], g[1] / g[0], g[3], g[4], g[2]))
        except sqlite3.Error as e:
            print("(no per-size table: %s)" % e)
        try:
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if cols:
                rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                                   "group by kernel_name, counter_name order by 1, 2").fetchall()
                if rows:
                    print("%-60s %-24s %8s %20s %22s" % ("kernel", "counter", "n", "avg/dispatch", "sum"))
                for k, c, n, avg, sm in rows:
                    print("%-60s %-24s %8d %20.1f %22.1f" % (k[:60], c, n, avg, sm))
        except sqlite3.Error as e:
            print("(no counters: %s)" % e)
        try:
            r = cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                            "workgroup_size_x, grid_size_x from kernels group by name").fetchall()
            for x in r:
                print("resources:", x)
        except sqlite3.Error as e:
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            print("(kernels columns: %s)" % cols)


if __name__ == "__main__":
    main()
