"""Where does a small launch spend its time?  One rank's 1/N share of the 1080p bench frame under NRS_DEBUG=4 (profiling instantiation: per-phase
s_memtime shares, voxel-walk counters, per-wave end times), plus plain timings of the same share.  usage: NRS_DEV_KNOBS=1 NRS_DEBUG=4 python tools/small_launch_probe.py [N]  (the library ignores NRS_DEBUG without NRS_DEV_KNOBS=1)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NRS_DEV_KNOBS"] = "1"  # (before libnrs is loaded: its measurement knobs are ignored without it)


def main():
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth, tiles
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = rt.Context(0)
    scene = bench.build_scene(os.environ.get("NRS_PROBE_SCENE", "lego_cage"), rt, synth, ctx, torch)
    tb = scene["tb"]
    W, H, T = 1920, 1080, bench.TILE
    sh = tiles.TileSharder(W, H, T, 0, N, "cuda:0")
    for step in range(3):
        p = synth.render_params(W, H, bench.camera_for(step, synth, 1), aabb_scale=1)
        sh.fill(p)
        sh.clear()
        st = tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None, want_stats=True)
        print(f"view {step}: samples {st.n_samples} rays {st.n_rays_alive}", file=sys.stderr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 32
    for step in range(K):
        p = synth.render_params(W, H, bench.camera_for(step % 8, synth, 1), aabb_scale=1)
        sh.fill(p)
        sh.clear()
        tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
    torch.cuda.synchronize()
    print(f"1/{N} share: {(time.perf_counter() - t0) * 1e3 / K:.3f} ms per share-frame", file=sys.stderr)


if __name__ == "__main__":
    main()
