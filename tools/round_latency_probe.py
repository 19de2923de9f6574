"""The closed-form floor of a rank's 1/N share: launch + fill + ceil(max samples per ray / lanes per ray) x round latency.

Measures on one MI355X, on the bench's lego + cage scene (view 0, 1920x1080, the rank's tiles of an N-GPU job):
  * the ROUND LATENCY of the share's launch: the same launch with every ray cut off after K samples (`max_march_steps` = MARCH_ITER, testbed_nerf.cu:56 --
    a ray that reaches it ends there, tn:957-960), K = 4, 8, 12, ... : the slope of kernel time over K is the time one more round of the longest rays costs
    while the rest of the launch is in flight around it; the intercept is launch + LDS staging + fill + first hits + the kernel's end;
  * the longest ray of the share (per-pixel step counts of the uncut frame) and the lanes per ray the automatic schedule ends on (4: re-teaming, hand-over);
  * the uncut launch.
    python tools/round_latency_probe.py > gpurun_out/round_latency.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth, tiles
    ctx = rt.Context(0)
    scene = bench.build_scene(os.environ.get("NRS_PROBE_SCENE", "lego_cage"), rt, synth, ctx, torch)
    tb = scene["tb"]
    W, H, T = 1920, 1080, bench.TILE

    def time_launch(p, sh, reps=24):
        for _ in range(6):
            sh.clear()
            tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
        torch.cuda.synchronize()
        ms = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            sh.clear()
            e0.record()
            tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        return ms[len(ms) // 2], ms[0]

    # warm the clocks
    wp = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
    wf = torch.zeros((H, W, 4), device="cuda:0"); wd = torch.zeros((H, W), device="cuda:0")
    for _ in range(100):
        tb.render_with_params(tb.nerf_network, wp, wf, wd, None, None)
    torch.cuda.synchronize()

    print("# The floor of a rank's 1/N share: launch + fill + rounds x round latency (lego + cage, view 0, 1920x1080, one frame at a time)\n")
    print("`K` = every ray cut off after K samples (max_march_steps); slope = ms per additional round of the longest rays; intercept = everything that is not a round.\n")
    rows = []
    for N in (1, 2, 4, 8):
        sh = tiles.TileSharder(W, H, T, 0, N, "cuda:0")
        p = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
        sh.fill(p)
        # the uncut launch + its longest ray
        steps = torch.zeros((sh.padded, T, T), dtype=torch.int32, device="cuda:0")
        sh.clear()
        st = tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, steps, None, want_stats=True)
        torch.cuda.synchronize()
        s = steps.cpu().numpy().reshape(-1)
        s_hit = s[s > 0]
        full_med, full_min = time_launch(p, sh)
        ks, ts = [], []
        for K in (4, 8, 12, 16, 20, 24, 28, 32):
            pk = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
            sh.fill(pk)
            pk.max_march_steps = K
            med, mn = time_launch(pk, sh, reps=16)
            ks.append(K); ts.append(med)
        ks_a, ts_a = np.array(ks, np.float64), np.array(ts, np.float64)
        # least squares over the K where every live ray is still running (K <= the 25th percentile of the ray lengths: nearly all rays reach the cut)
        lim = max(8, int(np.percentile(s_hit, 25)))
        sel = ks_a <= lim
        A = np.vstack([ks_a[sel], np.ones(sel.sum())]).T
        slope, icpt = np.linalg.lstsq(A, ts_a[sel], rcond=None)[0]
        rows.append((N, int(st.n_rays_alive), int(st.n_samples), int(s_hit.max()), float(np.percentile(s_hit, 99)), float(s_hit.mean()), full_med, full_min, slope, icpt, ks, ts, lim))
    print("| ranks N | rays | samples | longest ray (samples) | p99 | mean | uncut launch ms (median / min) | slope: ms per round (all rays running) | intercept ms | K used for the fit |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for N, rays, samples, smax, p99, mean, fm, fmin, slope, icpt, ks, ts, lim in rows:
        print(f"| {N} | {rays} | {samples} | {smax} | {p99:.0f} | {mean:.1f} | {fm:.3f} / {fmin:.3f} | {slope * 1e3:.1f} us | {icpt:.3f} | <= {lim} |")
    print("\nCut-off launches (median ms):\n")
    print("| ranks N | " + " | ".join(f"K = {k}" for k in rows[0][10]) + " |")
    print("|---|" + "---|" * len(rows[0][10]))
    for r in rows:
        print(f"| {r[0]} | " + " | ".join(f"{t:.3f}" for t in r[11]) + " |")
    print("\n## The model\n")
    print("floor(N) = intercept(N) + ceil(longest ray / lanes per ray) x L, with L the latency of ONE round of a wave on a GPU that holds only the launch's last rays (the drain): "
          "the slope at N = 8 is an upper estimate of it (all of the share's rays still run), the `K = 4 .. 8` step of the smallest share the closest this probe gets.\n")
    print("| ranks N | measured | intercept + longest / 1 x slope | / 2 | / 4 | 0.9 bar (N = 1 time / N / 0.9) |")
    print("|---|---|---|---|---|---|")
    t1 = rows[0][6]
    for N, rays, samples, smax, p99, mean, fm, fmin, slope, icpt, ks, ts, lim in rows:
        f = [icpt + -(-smax // L) * slope for L in (1, 2, 4)]
        print(f"| {N} | {fm:.3f} | {f[0]:.3f} | {f[1]:.3f} | {f[2]:.3f} | {t1 / N / 0.9:.3f} |")


if __name__ == "__main__":
    main()
