#!/usr/bin/env python3
"""Per-pass view of a rocprofv3 --pmc run of `bench.py --streams 1`: the two launches of a transform alternate, so
dispatch parity separates pass 1 from pass 2.  usage: pmc_by_pass.py <dir with *_results.db> <COUNTER>"""
import glob, os, sqlite3, sys
db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0])
rows = db.execute("select dispatch_id, value from counters_collection where counter_name=? and kernel_name like '%ntt_tile_kernel<11%' "
                  "order by dispatch_id", (sys.argv[2],)).fetchall()
ev = [v for i, (d, v) in enumerate(rows) if i % 2 == 0]; od = [v for i, (d, v) in enumerate(rows) if i % 2 == 1]
print(sys.argv[2], "launches", len(rows), "first-of-pair avg %.1f" % (sum(ev) / len(ev)), "second-of-pair avg %.1f" % (sum(od) / len(od)))
