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
    W, H = 1920, 1080
    base = None
    for N in (1, 2, 4, 8):
        sh = tiles.TileSharder(W, H, bench.TILE, 0, N, "cuda:0")
        times, samples = [], 0
        for step in range(-3, 16):
            p = synth.render_params(W, H, bench.camera_for(step % 8, synth, 1), aabb_scale=1)
            sh.fill(p)
            sh.clear()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st = tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None, want_stats=True)
            e1.record()
            torch.cuda.synchronize()
            if step >= 0:
                times.append(e0.elapsed_time(e1))
                samples += st.n_samples
        ms = sum(times) / len(times)
        rate = samples / sum(times) / 1e3
        base = base or rate
        print(json.dumps({"ranks": N, "share_of_frame": f"1/{N}", "render_ms_per_frame": round(ms, 3), "msamples_per_s_this_gpu": round(rate, 1),
                          "per_gpu_throughput_retained": round(rate / base, 3)}))


if __name__ == "__main__":
    main()
