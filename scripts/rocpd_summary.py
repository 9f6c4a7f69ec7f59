#!/usr/bin/env python3
"""Summarise rocprofv3 result databases (rocpd sqlite): per kernel, the number of dispatches, the average duration and
the average of every collected counter.  Usage: rocpd_summary.py <dir or .db> [name substring] -> JSON on stdout."""
import glob
import json
import os
import sqlite3
import sys


def summarise(path, want=None):
    out = {}
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for dbf in dbs:
        db = sqlite3.connect(dbf)
        cur = db.cursor()
        try:
            rows = cur.execute("select name, count(*), avg(duration) from kernels group by name").fetchall()
        except sqlite3.Error:
            continue
        for name, n, avg in rows:
            if want and want not in name:
                continue
            k = out.setdefault(name, {"calls": 0, "avg_ms": None, "counters": {}})
            k["calls"] = max(k["calls"], n)
            k["avg_ms"] = avg / 1e6
            durs = sorted(d[0] / 1e6 for d in cur.execute("select duration from kernels where name = ?", (name,)).fetchall())
            k["median_ms"] = durs[len(durs) // 2]
            k["max_ms"] = durs[-1]
            if len(durs) <= 16:      # (a bench_configs pass = a few full-size launches + one small validation launch)
                k["durations_ms"] = [round(d, 6) for d in durs]
        try:
            rows = cur.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
                               "group by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            rows = []
        for name, cname, nd, total in rows:
            if want and want not in name:
                continue
            k = out.setdefault(name, {"calls": nd, "avg_ms": None, "counters": {}})
            k["counters"][cname] = total / max(nd, 1)     # per dispatch
    return out


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None), indent=1))
