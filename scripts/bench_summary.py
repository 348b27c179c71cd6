#!/usr/bin/env python3
"""One line per object of a bench.py JSON line: value, ms per step, roofline fraction (and traffic where recorded)."""
import json
import sys


def show(name, d):
    if not isinstance(d, dict) or d.get("value") is None or "ms_per_step" not in d:
        print(name, json.dumps(d)[:3000] if isinstance(d, dict) else d)
        return
    r = d.get("roofline", {})
    extra = ""
    if d.get("config", {}).get("sync_entry_point_ms_per_step") is not None:
        extra = " sync-call %.3f ms" % d["config"]["sync_entry_point_ms_per_step"]
    if d.get("cpu_baseline", {}).get("value"):
        extra += " cpu %.3g %s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])
    print("%-10s %.4g %s  %.3f ms/step  %s frac %.3f (kernel %.3f ms x %s)%s" % (
        name, d["value"], d["unit"], d["ms_per_step"], r.get("bound"), r.get("frac", float("nan")), r.get("avg_launch_ms", float("nan")),
        r.get("launches", "-") if r.get("launches") is not None else "-", extra))


for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        show("main", d)
        for k, v in d.items():
            if isinstance(v, dict) and ("metric" in v):
                show(k, v)
