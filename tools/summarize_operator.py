#!/usr/bin/env python3
"""rocpd databases of tools/prof_operator.sh -> one markdown table per case (kernel time, counters per dispatch).  Cases are told apart by
kernel name and dispatch order (tools/op_driver.py)."""
import glob
import json
import os
import sqlite3
import sys

REPS = 4
CASES = [  # (label, kernel substring, slice of that kernel's dispatches in launch order, samples per dispatch)
    ("inference, 2^22 random samples", "network_kernel<0", slice(0, REPS), 1 << 22),
    ("inference, 2^22 ray-ordered samples", "network_kernel<0", slice(REPS, 2 * REPS), 1 << 22),
    ("density, 2^22 random samples", "network_kernel<1", slice(0, REPS), 1 << 22),
    ("refresh, aabb 1 (2^21 samples, cage operator)", "grid_refresh_kernel", slice(1, 1 + REPS), 1 << 21),
    ("refresh, aabb 16 (5 * 2^21 samples, cage operator)", "grid_refresh_kernel", slice(2 + REPS, 2 + 2 * REPS), 5 << 21),
]


def variants(sub):
    """demangled and mangled spellings of a kernel (template argument list: <1, ...> <-> ILi1E...)"""
    if "<" not in sub:
        return [sub]
    base, arg = sub.split("<")
    return [sub, f"{base}ILi{arg}E"]


def dispatches(cur, sub):
    for v in variants(sub):
        d = dispatches1(cur, v)
        if d:
            return d
    return []


def dispatches1(cur, sub):
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view', 'table')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select d.id, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id where s.display_name like ? or s.kernel_name like ? order by d.start"
    try:
        return cur.execute(q, (f"%{sub}%", f"%{sub}%")).fetchall()
    except sqlite3.OperationalError:
        q = f"select d.id, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id where s.kernel_name like ? order by d.start"
        return cur.execute(q, (f"%{sub}%",)).fetchall()


def counters(cur, sub):
    for v in variants(sub):
        c = counters1(cur, v)
        if c:
            return c
    return {}


def counters1(cur, sub):
    """{counter: [value per dispatch in launch order]}"""
    try:
        rows = cur.execute("select counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? group by counter_name, dispatch_id "
                           "order by dispatch_id", (f"%{sub}%",)).fetchall()
    except sqlite3.OperationalError:
        return {}
    out = {}
    for c, _, v in rows:
        out.setdefault(c, []).append(v)
    return out


def main():
    root, out = sys.argv[1], sys.argv[2]
    lines = ["# rocprofv3 summary: operator kernels (tools/prof_operator.sh, tools/op_driver.py)", ""]
    try:
        lines += ["host-timed (no profiler): `" + open(os.path.join(root, "plain.log")).read().strip().splitlines()[-1] + "`", ""]
    except Exception:
        pass
    dbs = {os.path.basename(os.path.dirname(p)): p for p in glob.glob(os.path.join(root, "*", "*_results.db"))}
    per_case = {c[0]: {} for c in CASES}
    for tag, db in sorted(dbs.items()):
        cur = sqlite3.connect(db).cursor()
        for label, sub, sl, n in CASES:
            if tag == "trace":
                d = dispatches(cur, sub)
                dur = [t for _, t in d][sl]
                if dur:
                    per_case[label]["kernel_ms"] = sum(dur) / len(dur) / 1e6
            else:
                for c, vals in counters(cur, sub).items():
                    v = vals[sl]
                    if v:
                        per_case[label][c] = sum(v) / len(v)
    for label, sub, sl, n in CASES:
        r = per_case[label]
        lines += [f"## {label}", ""]
        ms = r.get("kernel_ms")
        if ms:
            alg = 512 * n
            lines += [f"* kernel {ms:.3f} ms per dispatch -> {n / ms / 1e6:.2f} Gsamples/s; algorithmic gather bytes 512 B x {n} = {alg / 1e9:.2f} GB -> "
                      f"{alg / ms / 1e9:.2f} TB/s ({alg / ms / 1e9 / 8.0:.2f} of the 8 TB/s HBM roof); MFMA: {n * (6144 if 'inference' not in label else 20480) / ms / 1e9:.0f} TFLOP/s"]
        if "FETCH_SIZE" in r:
            lines += [f"* FETCH_SIZE {r['FETCH_SIZE']:.6g} KB (x2 on gfx950 = {2 * r['FETCH_SIZE'] * 1024 / 1e9:.2f} GB), WRITE_SIZE {r.get('WRITE_SIZE', float('nan')):.6g} KB"]
        if "TCC_HIT_sum" in r:
            h, m = r["TCC_HIT_sum"], r["TCC_MISS_sum"]
            lines += [f"* L2: requests {r.get('TCC_REQ_sum', 0):.4g}, hit {h:.4g}, miss {m:.4g} (hit rate {h / max(h + m, 1):.2f}); miss x 128 B = {m * 128 / 1e9:.2f} GB"
                      + (f" -> {m * 128 / ms / 1e9:.2f} TB/s" if ms else "")]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in r:
            lines += [f"* L1: accesses {r['TCP_TOTAL_CACHE_ACCESSES_sum']:.4g}, read requests to L2 {r.get('TCP_TCC_READ_REQ_sum', 0):.4g}, TA_BUSY_avr {r.get('TA_BUSY_avr', 0):.4g}"]
        if "SQ_BUSY_CYCLES" in r:
            lines += ["* SQ: " + ", ".join(f"{k} {r[k]:.4g}" for k in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                                                                      "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE") if k in r)]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in r and "GRBM_GUI_ACTIVE" in r:
                cyc = r["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
                lines += [f"* MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles {cyc:.4g} x 1024 SIMDs) = {r['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.4f}"]
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
