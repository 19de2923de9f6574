"""Single-GPU probe of the multi-GPU compute leg (rounds 3-5; was tools/scale_probe_r03.py): (1) one rank's share of the 1080p bench frame (tiles r, r + N, ...) for N = 1, 2, 4, 8 with
1 / 2 / 4 frames in flight -- what a rank of an N-GPU job does between gathers; (2) the load balance of the round-robin deal of 32x32 tiles: samples per rank
for the 8 bench views at N = 2 / 4 / 8, from the per-pixel step counts of whole-frame renders.  Writes markdown to stdout.
    python tools/scale_probe.py > gpurun_out/scaling.md   (NRS_PROBE_QUICK=1: 1 and 2 frames in flight only, no load-balance table)"""
import os
import sys
import time

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
    warm = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
    wf = torch.zeros((H, W, 4), device="cuda:0"); wd = torch.zeros((H, W), device="cuda:0")
    for _ in range(150):
        tb.render_with_params(tb.nerf_network, warm, wf, wd, None, None)
    torch.cuda.synchronize()
    print("# Strong-scaling compute leg on ONE MI355X\n")
    print("One rank's share of the 1080p lego + cage frame (32x32 tiles dealt round-robin on the odd-pitch tile index), 32 frames over the 8 bench views, automatic lane-team choice.")
    print("`retained` = this GPU's samples/s relative to the whole frame rendered one at a time (N = 1, 1 in flight): the quantity north_star's 0.9 target is about.\n")
    print("| ranks N | share | frames in flight | ms per share-frame | Msamples/s on this GPU | retained |")
    print("|---|---|---|---|---|---|")
    base = None
    for N in tuple(int(v) for v in os.environ.get('NRS_PROBE_N', '1,2,4,8').split(',')):
        for F in ((1, 2) if os.environ.get('NRS_PROBE_QUICK') else (1, 2, 4)):
            shs = [tiles.TileSharder(W, H, T, 0, N, "cuda:0") for _ in range(F)]
            streams = [torch.cuda.Stream() for _ in range(F)]
            samples = 0
            for step in range(8):
                p = synth.render_params(W, H, bench.camera_for(step, synth, 1), aabb_scale=1)
                shs[0].fill(p)
                samples += tb.render_with_params(tb.nerf_network, p, shs[0].local_frame, shs[0].local_depth, None, None, want_stats=True).n_samples
            for rep in range(2):
                torch.cuda.synchronize()
                K = 32
                t0 = time.perf_counter()
                for step in range(K):
                    b = step % F
                    p = synth.render_params(W, H, bench.camera_for(step % 8, synth, 1), aabb_scale=1)
                    shs[b].fill(p)
                    with torch.cuda.stream(streams[b]):
                        shs[b].clear()
                        tb.render_with_params(tb.nerf_network, p, shs[b].local_frame, shs[b].local_depth, None, streams[b])
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / K
            rate = samples * (K / 8) / (ms * K) / 1e3
            if base is None:
                base = rate
            print(f"| {N} | 1/{N} | {F} | {ms:.3f} | {rate:.0f} | {rate / base:.2f} |")
    if os.environ.get("NRS_PROBE_QUICK"):
        return
    # ---- load balance of the round-robin deal
    print("\n## Load balance of the round-robin deal (odd-pitch tile index)\n")
    print("Samples per rank (from the per-pixel step counts of whole-frame renders), max / mean over the ranks: the slowest rank bounds a frame.\n")
    print("| view | N = 2 max/mean | N = 4 max/mean | N = 8 max/mean | N = 8 min/mean |")
    print("|---|---|---|---|---|")
    frame = torch.zeros((H, W, 4), device="cuda:0"); depth = torch.zeros((H, W), device="cuda:0"); steps = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    tx, ty = (W + T - 1) // T, (H + T - 1) // T
    worst = {2: 0.0, 4: 0.0, 8: 0.0}
    for view in range(8):
        p = synth.render_params(W, H, bench.camera_for(view, synth, 1), aabb_scale=1)
        frame.zero_()
        tb.render_with_params(tb.nerf_network, p, frame, depth, steps, None)
        torch.cuda.synchronize()
        s = steps.cpu().numpy().astype(np.int64)
        pad = np.zeros((ty * T, tx * T), np.int64); pad[:H, :W] = s
        per_tile = pad.reshape(ty, T, tx, T).sum(axis=(1, 3))
        pitch = tiles.tile_pitch(W, T)          # the tile index runs over an ODD row pitch (include/nrs.h): indices beyond the last column are virtual
        grid = np.zeros((ty, pitch), np.int64); grid[:, :tx] = per_tile
        per_tile = grid.reshape(-1)
        row = [f"{view}"]
        for N in (2, 4, 8):
            per_rank = np.array([per_tile[r::N].sum() for r in range(N)], np.float64)
            worst[N] = max(worst[N], per_rank.max() / per_rank.mean())
            row.append(f"{per_rank.max() / per_rank.mean():.3f}")
        row.append(f"{per_rank.min() / per_rank.mean():.3f}")
        print("| " + " | ".join(row) + " |")
    print(f"\nworst max/mean over the 8 views: N = 2: {worst[2]:.3f}, N = 4: {worst[4]:.3f}, N = 8: {worst[8]:.3f} (1.000 = perfect balance)")


if __name__ == "__main__":
    main()
