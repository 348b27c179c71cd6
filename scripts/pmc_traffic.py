#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md 'HBM') of one bench
command into the per-launch traffic record bench.py reports as roofline.traffic.
usage: pmc_traffic.py <workload> <kernel substring> <fetch_results.db> <write_results.db> <out.json>"""
import json
import sqlite3
import sys


def avg_counter(db, kernel_sub, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    vn = "value" if "value" in cols else "counter_value"
    rows = list(cur.execute("select %s, avg(%s), count(*) from counters_collection where counter_name = ? group by %s" % (kn, vn, kn),
                            (counter,)))
    rows = [r for r in rows if kernel_sub in r[0]]
    if not rows:
        return None, 0
    r = max(rows, key=lambda x: x[1] * x[2])
    return float(r[1]), int(r[2])


def main():
    workload, ksub, fdb, wdb, out = sys.argv[1:6]
    f, nf = avg_counter(fdb, ksub, "FETCH_SIZE")
    w, nw = avg_counter(wdb, ksub, "WRITE_SIZE")
    try:
        rec = json.load(open(out))
    except Exception:
        rec = {}
    rec[workload] = {
        "kernel": ksub,
        "fetch_bytes_per_launch": None if f is None else f * 1024.0,
        "write_bytes_per_launch": None if w is None else w * 1024.0,
        "launches_sampled": [nf, nw],
        "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the bench command; raw "
               "counter x 1 KiB, per-dispatch average.  The guide's x2 FETCH_SIZE correction applies to 16 B/lane "
               "streaming reads; this kernel gathers with dword loads / fp32 atomics, a width the guide lists as "
               "uncalibrated, so the raw value is reported",
    }
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec[workload]))


if __name__ == "__main__":
    main()
