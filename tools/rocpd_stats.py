"""kernel statistics from a rocprofv3 `*_results.db` (rocpd sqlite): name, calls, average and total duration"""
import glob
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from %s d join %s s "
         "on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (kd, ks))
    return list(cur.execute(q))


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for f in glob.glob(path, recursive=True):
            print("#", f)
            rows = stats(f)
            tot = sum(r[3] for r in rows)
            print("%-72s %6s %12s %12s %6s" % ("kernel", "calls", "avg_us", "total_us", "%"))
            for r in rows:
                print("%-72s %6d %12.1f %12.1f %6.1f" % (r[0][:72], r[1], r[2], r[3], 100.0 * r[3] / tot))
