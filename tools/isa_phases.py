#!/usr/bin/env python3
"""Instruction counts per basic block of one kernel in a hipcc -S listing built with -DNRS_MEASURE=9 (nrs_kernels.hip: NRS_MARK comments at
the phase boundaries of render_kernel).  usage: isa_phases.py listing.s <mangled kernel substring>"""
import re
import sys


def main():
    path, sub = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and sub in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    phase, block = "pre", "entry"
    rows, cur = [], None

    def flush():
        nonlocal cur
        if cur and sum(cur[2].values()):
            rows.append(cur)
        cur = [phase, block, {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "mfma": 0, "smem": 0, "branch": 0, "wait": 0, "other": 0}]

    flush()
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t:
            continue
        m = re.match(r"; NRS_MARK (\d+)", t)
        if m:
            phase = m.group(1)
            flush()
            continue
        if re.match(r"\.LBB\d+_\d+:", t):
            block = t.split(":")[0]
            flush()
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        c = cur[2]
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
            c["wait"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc"):
            c["branch"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["other"] += 1
    flush()
    print(f"{'phase':>5} {'block':>12} " + " ".join(f"{k:>6}" for k in rows[0][2]))
    tot = {}
    for ph, b, c in rows:
        print(f"{ph:>5} {b:>12} " + " ".join(f"{v:>6}" for v in c.values()))
        t = tot.setdefault(ph, dict.fromkeys(c, 0))
        for k, v in c.items():
            t[k] += v
    print("static totals per phase:")
    for ph, c in tot.items():
        print(f"{ph:>5} {'':>12} " + " ".join(f"{v:>6}" for v in c.values()))


if __name__ == "__main__":
    main()
