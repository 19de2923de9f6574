"""Where a rank's share of the frame spends its time (VERDICT r4 next #4a): one rank's tiles of the 1080p bench frame (N = 1, 2, 4, 8; view 0) rendered by the
PROFILING instantiation (NRS_DEBUG=4: s_memtime per phase, the per-wave log), next to the production kernel's time for the same launch.

    python tools/share_phase_probe.py [workload] > gpurun_out/share_phases.md        (runs on the GPU box; spawns itself once per N with NRS_DEBUG=4)

Per N the report gives, from the wave log (wall_clock64, 100 MHz, common to all XCDs): when the last wave started, when the queue ran dry (the first wave to see it),
when half of the waves had finished, when the last one had -- i.e. the launch's critical path cut into
   start   launch begins .. every wave runs (dispatch + LDS staging of the weights)
   fill    .. the frame's queue is dry: packets claimed, primary rays set up, first hits found (rounds of earlier generations overlap here)
   rounds  .. half of the waves have finished
   drain   .. the last wave has finished (the long rays; lane teams, re-teaming and the hand-over work here)
and, from the phase counters, how the waves' own time divides (fill / refill / set-up + warp / gather / SH + MLP / composite + march)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(workload, N, log_path):
    import numpy as np  # noqa: F401
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth, tiles
    ctx = rt.Context(0)
    scene = bench.build_scene(workload, rt, synth, ctx, torch)
    tb = scene["tb"]
    W, H = 1920, 1080
    sh = tiles.TileSharder(W, H, bench.TILE, 0, N, "cuda:0")
    p = synth.render_params(W, H, bench.camera_for(0, synth, scene["aabb_scale"]), aabb_scale=scene["aabb_scale"])
    sh.fill(p)
    for _ in range(40):
        sh.clear()
        tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(16):
        sh.clear()
        e0.record()
        tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    st = tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None, want_stats=True)  # the launch the wave log is taken from
    print(f"PROBE n={N} kernel_ms_median={ms[len(ms) // 2]:.4f} kernel_ms_min={ms[0]:.4f} samples={int(st.n_samples)} rays={int(st.n_rays_alive)}", flush=True)


def report(N, log_path, prof_line, prod_line, stderr_text):
    import numpy as np
    raw = np.fromfile(log_path, dtype=np.uint64).reshape(-1, 4)
    raw = raw[raw[:, 0] != 0]
    rounds = (raw[:, 1] & np.uint64(0xffff)).astype(np.int64)
    tq = (raw[:, 1] >> np.uint64(32)).astype(np.float64) / 100.0       # us after the wave's start (0: the wave never saw the queue dry before it ended)
    wall = (raw[:, 3] & np.uint64(0xffffffff)).astype(np.float64) / 100.0
    start = (raw[:, 3] >> np.uint64(32)).astype(np.float64)
    start = (start - start.min()) / 100.0
    end = start + wall
    t_start = start.max()
    t_dry = (start + tq)[tq > 0].min() if (tq > 0).any() else float("nan")  # the FIRST wave to find the queue dry: from here on no new rays enter
    t_half = np.median(end)
    t_end = end.max()
    phases = dict(re.findall(r" ([a-z+]+)=([0-9.]+)%", re.search(r"\[nrs phases\].*", stderr_text).group(0)))
    life = re.search(r"mean wave lifetime = ([0-9.]+)% of the longest", stderr_text).group(1)
    walk = re.search(r"\[nrs walk\].*", stderr_text)
    g = lambda line, key: float(re.search(key + r"=([0-9.]+)", line).group(1))
    return dict(N=N, prod_ms=g(prod_line, "kernel_ms_median"), prof_ms=g(prof_line, "kernel_ms_median"), samples=int(g(prof_line, "samples")), rays=int(g(prof_line, "rays")),
                waves=len(end), t_start=t_start, t_dry=t_dry, t_half=t_half, t_end=t_end, rounds_mean=rounds.mean(), rounds_max=int(rounds.max()), phases=phases, life=life,
                walk=walk.group(0) if walk else "")


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        return
    workload = sys.argv[1] if len(sys.argv) > 1 else "lego_cage"
    rows = []
    for N in tuple(int(v) for v in os.environ.get("NRS_PROBE_N", "1,2,4,8").split(",")):
        log_path = f"/tmp/nrs_wave_{N}.bin"
        cmd = [sys.executable, os.path.abspath(__file__), "--child", workload, str(N), log_path]
        prod = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, NRS_DEV_KNOBS="1", NRS_DEBUG="0"))
        prof = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, NRS_DEV_KNOBS="1", NRS_DEBUG="4", NRS_WAVE_LOG_FILE=log_path))
        pl = [l for l in prod.stdout.splitlines() if l.startswith("PROBE")]
        fl = [l for l in prof.stdout.splitlines() if l.startswith("PROBE")]
        if not pl or not fl:
            print(f"N={N}: probe failed\n{prod.stderr[-800:]}\n{prof.stderr[-800:]}")
            continue
        rows.append(report(N, log_path, fl[0], pl[0], prof.stderr))
    print(f"# A rank's share of the 1080p frame, phase by phase (`{workload}`, view 0, one frame at a time)\n")
    print("Production kernel = what bench.py runs; the breakdown comes from the profiling instantiation of the same body (`NRS_DEBUG=4`: `s_memtime` stamps per phase and a per-wave log, which cost it "
          "some speed: both times are given).  Critical path from the wave log, in microseconds after the first wave's start.\n")
    print("| ranks N | rays | samples | production ms | profiling ms | waves | all waves started | queue dry | half of the waves done | last wave done | rounds per wave mean / max | mean wave life / longest |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['N']} | {r['rays']} | {r['samples']} | {r['prod_ms']:.3f} | {r['prof_ms']:.3f} | {r['waves']} | {r['t_start']:.0f} | {r['t_dry']:.0f} | {r['t_half']:.0f} | {r['t_end']:.0f} | "
              f"{r['rounds_mean']:.1f} / {r['rounds_max']} | {r['life']} % |")
    print("\nThe same as shares of the launch (profiling build; `launch + start` = its HIP-event time minus the span of the wave log, i.e. dispatch, LDS staging and the kernel's end):\n")
    print("| ranks N | launch + start | fill (.. queue dry) | rounds (.. half done) | drain (.. last wave) |")
    print("|---|---|---|---|---|")
    for r in rows:
        tot = r["prof_ms"] * 1e3
        pre = max(tot - r["t_end"], 0.0) + r["t_start"]
        print(f"| {r['N']} | {pre:.0f} us ({100 * pre / tot:.0f} %) | {max(r['t_dry'] - r['t_start'], 0):.0f} us ({100 * max(r['t_dry'] - r['t_start'], 0) / tot:.0f} %) | "
              f"{max(r['t_half'] - r['t_dry'], 0):.0f} us ({100 * max(r['t_half'] - r['t_dry'], 0) / tot:.0f} %) | {r['t_end'] - r['t_half']:.0f} us ({100 * (r['t_end'] - r['t_half']) / tot:.0f} %) |")
    print("\nHow the waves' own time divides (sum over waves of the `s_memtime` phase stamps):\n")
    names = ["fill", "refill", "setup+warp", "gather", "sh+mlp", "composite+march+shade"]
    print("| ranks N | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for r in rows:
        print(f"| {r['N']} | " + " | ".join(f"{r['phases'].get(n, '0')} %" for n in names) + " |")
    print("\nVoxel-walk counters of the same launches (a `trip` = one pass of the wave through the walk's loop body; a round's march costs as many trips as its slowest lane needs):\n")
    for r in rows:
        print(f"* N = {r['N']}: `{r['walk']}`")


if __name__ == "__main__":
    main()
