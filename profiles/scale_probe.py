"""Single-GPU probe of the strong-scaling compute leg: one rank's share of a 1080p frame (tiles r, r+N, ...) rendered on
one MI355X, for N = 1, 2, 4, 8.  The RCCL gather is not part of it (no second GPU on the dev box); this isolates how the
per-frame ray latency limits per-GPU time when the work per GPU shrinks.   python profiles/scale_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth, tiles
    ctx = rt.Context(0)
    scene = bench.build_scene("lego_cage", rt, synth, ctx, torch)
    tb = scene["tb"]
    W, H = int(os.environ.get("PROBE_W", 1920)), int(os.environ.get("PROBE_H", 1080))
    ranks = [int(v) for v in os.environ.get("PROBE_RANKS", "1,2,4,8").split(",")]
    flights = [int(v) for v in os.environ.get("PROBE_FLIGHTS", "1,2").split(",")]
    import time
    base = {}
    warm = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
    wf = torch.zeros((H, W, 4), device="cuda:0"); wd = torch.zeros((H, W), device="cuda:0")
    for _ in range(150):   # clocks settle over the first ~100 frames of a process: without this the first rows read 5-10 % slow
        tb.render_with_params(tb.nerf_network, warm, wf, wd, None, None)
    torch.cuda.synchronize()
    teams = [int(v) for v in os.environ.get("PROBE_TEAMS", "0").split(",")]   # 0 = automatic
    for team in teams:
      ctx.set_lane_teams(team)
      for F in flights:   # frames in flight (bench.py uses 2 for N > 1)
          for N in ranks:
              shs = [tiles.TileSharder(W, H, bench.TILE, 0, N, "cuda:0") for _ in range(F)]
              streams = [torch.cuda.Stream() for _ in range(F)]
              samples = 0
              for step in range(8):   # sample counts (untimed)
                  p = synth.render_params(W, H, bench.camera_for(step, synth, 1), aabb_scale=1)
                  shs[0].fill(p)
                  samples += tb.render_with_params(tb.nerf_network, p, shs[0].local_frame, shs[0].local_depth, None, None, want_stats=True).n_samples
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
              base.setdefault(F, rate); base.setdefault(1, rate)
              print(json.dumps({"lanes_per_ray": team or "auto", "frames_in_flight": F, "ranks": N, "share_of_frame": f"1/{N}", "ms_per_frame": round(ms, 3),
                                "msamples_per_s_this_gpu": round(rate, 1), "per_gpu_throughput_retained_vs_1_gpu_sequential": round(rate / base[1], 3)}))


if __name__ == "__main__":
    main()
