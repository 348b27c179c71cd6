#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) into the text we commit under profiles/:
per-kernel launch count / total / average / min / max duration (the `--stats` view) and, when a
--pmc pass was recorded, the per-kernel average of every counter."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    q = ("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
         "group by %s order by sum(end-start) desc" % (name_col, name_col))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("# kernel stats from %s" % path)
    print("%-90s %8s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for r in rows:
        print("%-90s %8d %14d %12.0f %12d %12d %6.2f" % (r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    try:
        pcols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if pcols:
            print("\n# counters_collection columns: %s" % pcols)
            kn = "kernel_name" if "kernel_name" in pcols else ("name" if "name" in pcols else None)
            cn = "counter_name" if "counter_name" in pcols else None
            vn = "value" if "value" in pcols else ("counter_value" if "counter_value" in pcols else None)
            if kn and cn and vn:
                rows = list(cur.execute("select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s "
                                        "order by sum(%s) desc" % (kn, cn, vn, vn, kn, cn, vn)))
                if rows:
                    print("\n# PMC counters (per-dispatch average, summed over dimensions as recorded)")
                    print("%-90s %-16s %8s %18s %20s" % ("kernel", "counter", "rows", "avg_value", "sum_value"))
                    for r in rows:
                        print("%-90s %-16s %8d %18.1f %20.1f" % (r[0][:90], r[1], r[2], r[3], r[4]))
            else:
                print("\n# counters_collection columns:", pcols)
    except sqlite3.Error as e:
        print("# no counters:", e)


if __name__ == "__main__":
    main(sys.argv[1])
