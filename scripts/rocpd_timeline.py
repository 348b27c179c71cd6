#!/usr/bin/env python3
"""The kernels of a rocprofv3 rocpd database in time order: start (us from the first kernel), duration, queue / stream, name -- the
last N of them (default 60).  What overlaps what is read off the start / end columns.  usage: rocpd_timeline.py results.db [N] [substr]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sub = sys.argv[3] if len(sys.argv) > 3 else ""
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
extra = [c for c in ("queue_id", "stream_id", "tid") if c in cols]
rows = list(cur.execute("select start, end, %s%s from kernels order by start" % (name_col, "".join(", " + c for c in extra))))
rows = [r for r in rows if sub in r[2]] if sub else rows
t0 = rows[0][0]
print("# columns: start_us end_us dur_us gap_to_prev_end_us %s name" % " ".join(extra))
prev_end = None
for r in rows[-n:]:
    gap = (r[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%12.1f %12.1f %9.1f %9.1f %s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, gap,
                                              " ".join(str(x) for x in r[3:]), r[2][:70]))
    prev_end = r[1] if prev_end is None else max(prev_end, r[1])
