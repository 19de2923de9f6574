#!/usr/bin/env python3
"""Per-dispatch means of the PMC counters of one kernel from a rocprofv3 output directory (rocpd sqlite):
    python tools/pmc_kernel.py <dir> [kernel substring = render_kernel]   -> one JSON object on stdout
{"dispatches": n, "avg_ms": kernel-trace mean duration, "<counter>": mean over dispatches of the per-dispatch sum}"""
import glob
import json
import os
import sqlite3
import sys


def read(root, needle="render_kernel"):
    out = {}
    for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? group by counter_name, dispatch_id",
                               (f"%{needle}%",)).fetchall()
        except sqlite3.OperationalError:
            rows = []
        per = {}
        for c, _, v in rows:
            per.setdefault(c, []).append(v)
        for c, vs in per.items():
            out[c] = sum(vs) / len(vs)
            out["dispatches"] = len(vs)
        try:
            r = cur.execute("select total_calls, average from top_kernels where name like ?", (f"%{needle}%",)).fetchall()
            if r:
                out["calls"] = sum(x[0] for x in r)
                out["avg_ms"] = round(sum(x[0] * x[1] for x in r) / sum(x[0] for x in r) / 1e3, 4)  # (top_kernels.average is in microseconds: profiles/summarize.py)
        except sqlite3.OperationalError:
            pass
    return out


if __name__ == "__main__":
    print(json.dumps(read(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "render_kernel")))
