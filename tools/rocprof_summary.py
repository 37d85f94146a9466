#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into a small text/JSON report.

usage: rocprof_summary.py <dir with trace/ fetch/ write/ sq/ sub-runs> <out prefix> [source label [workload]]
The JSON records the sha256 of the tile-kernel sources (bench.kernel_source_hash): bench.py refuses counters taken on
other sources.
  trace/  rocprofv3 --kernel-trace --stats      -> per-kernel calls / average duration
  fetch/  rocprofv3 --pmc FETCH_SIZE             (own pass: TCC slot budget, MI355X_MICROARCH.md)
  write/  rocprofv3 --pmc WRITE_SIZE             (own pass)
  sq/     rocprofv3 --pmc SQ_* counters
HBM bytes per launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024: on gfx950 FETCH_SIZE reports half the bytes
of a coalesced streaming read (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is taken at face value
and checks out against a known byte count here (every launch writes exactly n*8 bytes).
"""
import glob
import json
import os
import sqlite3
import sys


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def main():
    root, out = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rep = {"kernel_source_hash": bench.kernel_source_hash(), "source": sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out) + ".txt",
           "kernels": [], "counters": {}}
    wl = sys.argv[4] if len(sys.argv) > 4 else None
    if wl in bench.SOURCES_BY_WORKLOAD:      # a workload with its own kernel sources (the scans): validated against those
        rep["workload_source_hash"] = bench.kernel_source_hash(bench.SOURCES_BY_WORKLOAD[wl])
        rep["workload_source_note"] = "sha256 of " + ", ".join(bench.SOURCES_BY_WORKLOAD[wl])
    db = db_of(os.path.join(root, "trace"))
    if db:
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            rep["kernels"].append({"name": name, "calls": calls, "total_us": total, "avg_us": avg, "pct": pct})
    for sub in ("fetch", "write", "sq"):
        db = db_of(os.path.join(root, sub))
        if not db:
            continue
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
             "group by kernel_name, counter_name")
        for k, c, n, a, lo, hi in db.execute(q):
            rep["counters"].setdefault(k, {})[c] = {"launches": n, "avg": a, "min": lo, "max": hi}
    avg_us = {k["name"]: k["avg_us"] for k in rep["kernels"]}
    for k, c in rep["counters"].items():
        if k in avg_us:
            c["_avg_us"] = avg_us[k]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd = c["FETCH_SIZE"]["avg"] * 1024 * 2
            wr = c["WRITE_SIZE"]["avg"] * 1024
            c["_hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
    with open(out + ".json", "w") as f:
        json.dump(rep, f, indent=1)
    with open(out + ".txt", "w") as f:
        f.write("rocprofv3 summary of %s\n\n== kernel trace (--kernel-trace --stats)\n" % root)
        for k in rep["kernels"]:
            f.write("%-70s calls %6d  avg %9.3f us  total %11.1f us  %5.1f%%\n" % (k["name"][:70], k["calls"], k["avg_us"], k["total_us"], k["pct"]))
        f.write("\n== PMC counters (per launch: avg [min .. max])\n")
        for k, c in rep["counters"].items():
            f.write("%s\n" % k)
            for name, v in sorted(c.items()):
                if name == "_avg_us":
                    continue
                if name.startswith("_"):
                    f.write("    HBM bytes/launch: read %.0f (FETCH_SIZE KB x1024 x2 gfx950 correction) + write %.0f = %.0f\n"
                            % (v["read_corrected_x2"], v["write"], v["total"]))
                else:
                    f.write("    %-24s %16.1f [%14.1f .. %14.1f]  (%d launches)\n" % (name, v["avg"], v["min"], v["max"], v["launches"]))
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
