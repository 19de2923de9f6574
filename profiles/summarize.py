#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (gpurun_out/prof/<pass>/bench_results.db) into the small text summaries that are
committed under profiles/:  python profiles/summarize.py gpurun_out/prof profiles/r01_<tag>.md "<title>" """
import glob
import os
import sqlite3
import sys


def main():
    root, out, title = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    lines = [f"# rocprofv3 summary: {title}", ""]
    for db in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(db))
        cur = sqlite3.connect(db).cursor()
        lines += [f"## pass `{tag}`", ""]
        rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 8").fetchall()
        if rows:
            lines += ["| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
            for n, c, tot, avg, pct in rows:
                lines.append(f"| `{n[:90]}` | {c} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {pct:.2f} |")
            lines.append("")
        try:
            rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), max(vgpr_count), max(accum_vgpr_count), "
                               "max(sgpr_count), max(lds_block_size), max(grid_size), max(workgroup_size) from counters_collection "
                               "group by kernel_name, counter_name order by kernel_name").fetchall()
        except sqlite3.OperationalError:
            rows = []
        rows = [r for r in rows if "render_kernel" in r[0] or "network_kernel" in r[0]]
        if rows:
            lines += ["| kernel | counter | dispatches | avg / dispatch | min | max |", "|---|---|---|---|---|---|"]
            for k, cn, n, avg, mn, mx, vg, ag, sg, lds, grid, wg in rows:
                lines.append(f"| `{k[:60]}` | {cn} | {n} | {avg:.6g} | {mn:.6g} | {mx:.6g} |")
            k, cn, n, avg, mn, mx, vg, ag, sg, lds, grid, wg = rows[0]
            # (rocprofv3's vgpr_count column is not the compiler's NumVgprs on gfx950 -- it printed 64 for the 125-register render kernel of round 3 --
            # so it is labelled as what it is; the allocation is in `make -C nerfshop_amd/csrc resource-usage | python tools/resource_usage.py`)
            lines += ["", f"dispatch shape: grid {grid} x wg {wg}, lds {lds} B, sgpr {sg}; rocprofv3 `vgpr_count` field {vg} (NOT the allocation: see tools/resource_usage.py), agpr {ag}", ""]
    open(out, "w").write("\n".join(lines) + "\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
