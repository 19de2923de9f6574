#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the PMC passes of tools/prof.sh:  python tools/make_traffic_json.py <out.json> <round tag> <workload>=<gpurun_out/prof_tag dir> ...
(the first workload is the headline: its fields sit at the top level, the others under their own key -- the layout bench.measured_traffic reads)"""
import json
import os
import sys



def entry(workload, d):
    c = json.load(open(os.path.join(d, "counters.json")))  # written by tools/prof.sh: {pass: per-dispatch means of the render kernel's counters}
    f = c["pmc_fetch"].get("FETCH_SIZE")
    w = c["pmc_write"].get("WRITE_SIZE")
    t = c["pmc_tcc"]
    n = t.get("dispatches")
    e = {"source": f"tools/prof.sh {workload} {os.path.basename(d).replace('prof_', '')} -> rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, gfx950 correction "
                   f"2 x FETCH_SIZE + WRITE_SIZE; mean over {n} dispatches; the round's final build",
         "fetch_size_kb": f, "write_size_kb": w, "traffic_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
    if "TCC_MISS_sum" in t:
        e["tcc_miss"] = t["TCC_MISS_sum"]
        e["tcc_miss_x_128B"] = int(t["TCC_MISS_sum"] * 128)
        e["tcc_req"] = t.get("TCC_REQ_sum")
    return e


def main():
    out = sys.argv[1]
    specs = [a.split("=", 1) for a in sys.argv[2:]]
    j = entry(*specs[0])
    for wl, d in specs[1:]:
        j[wl] = entry(wl, d)
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j)[:400])


if __name__ == "__main__":
    main()
