#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md 'HBM') of one bench
command into the per-launch traffic record bench.py reports as roofline.traffic.

A bench command launches its dominant kernel at several sizes (warm-ups on a few thousand queries, pilot sweeps over a
sample of the rows, then the timed full-size launches).  The record describes the FULL-SIZE launch: per dispatch the
counter is summed over its dimensions, the largest dispatch is taken as the reference and the record is the mean over
the dispatches within 10 % of it (the timed launches); how many there were and how many smaller ones were left out is
recorded next to it.

usage: pmc_traffic.py <workload> <kernel substring> <fetch_results.db> <write_results.db> <out.json> [wide|narrow]
  wide   = the kernel streams with 16-byte-per-lane loads: FETCH_SIZE x 2 (the guide's gfx950 correction), raw value kept
  narrow = dword gathers / fp32 atomics (default): the guide lists that width as uncalibrated, the raw value is reported"""
import json
import os
import sqlite3
import sys


def per_dispatch(db, kernel_sub, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    vn = "value" if "value" in cols else "counter_value"
    did = next((c for c in ("dispatch_id", "dispatch_index", "correlation_id", "id") if c in cols), None)
    if did is None:
        raise SystemExit("counters_collection has no per-dispatch key: %s" % cols)
    rows = list(cur.execute("select %s, %s, sum(%s) from counters_collection where counter_name = ? group by %s, %s"
                            % (kn, did, vn, kn, did), (counter,)))
    return [float(r[2]) for r in rows if kernel_sub in r[0]]


def full_size(values):
    if not values:
        return None, 0, 0
    top = max(values)
    full = [v for v in values if v >= 0.9 * top]
    return sum(full) / len(full), len(full), len(values) - len(full)


def main():
    workload, ksub, fdb, wdb, out = sys.argv[1:6]
    wide = len(sys.argv) > 6 and sys.argv[6] == "wide"
    f, nf, sf = full_size(per_dispatch(fdb, ksub, "FETCH_SIZE"))
    w, nw, sw = full_size(per_dispatch(wdb, ksub, "WRITE_SIZE"))
    try:
        rec = json.load(open(out))
    except Exception:
        rec = {}
    fetch = None if f is None else f * 1024.0 * (2.0 if wide else 1.0)
    rec[workload] = {
        "kernel": ksub,
        "session": os.environ.get("GORSE_PMC_SESSION", ""),  # bench.py reports a record only if this names the current round
        "fetch_bytes_per_launch": fetch,
        "fetch_bytes_raw": None if f is None else f * 1024.0,
        "write_bytes_per_launch": None if w is None else w * 1024.0,
        "full_size_launches": [nf, nw],
        "smaller_launches_left_out": [sf, sw],
        "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the bench command; counter x 1 KiB "
               "summed per dispatch, mean over the full-size dispatches (within 10 % of the largest); "
               + ("FETCH_SIZE x 2: 16-byte-per-lane streaming loads (MI355X_MICROARCH.md, HBM)" if wide else
                  "dword gathers / fp32 atomics: a width the guide lists as uncalibrated, raw value"),
    }
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec[workload]))


if __name__ == "__main__":
    main()
