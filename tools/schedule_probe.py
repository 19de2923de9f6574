"""Which schedule for which launch size?  Whole-image renders of the bench scene at several resolutions, and one rank's tiles of the 1080p frame for
N = 1, 2, 4, 8, each with the schedule forced through nrs_ctx_set_lane_teams (0 automatic, -1 hybrid, -2 / -3 / -4 small-launch with 4x4 / 8x4 / 8x8-pixel packets, 1 / 2 / 4 lanes per ray):
ms per frame over the 8 bench views, one frame at a time.  Markdown to stdout.
    python tools/schedule_probe.py [whole|tiles|all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TEAMS = (0, -1, -2, -3, -4, 1, 2, 4)


def main():
    import torch
    import bench
    from nerfshop_amd import runtime as rt, synth, tiles
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ctx = rt.Context(0)
    scene = bench.build_scene(os.environ.get("NRS_PROBE_SCENE", "lego_cage"), rt, synth, ctx, torch)
    tb = scene["tb"]
    warm = synth.render_params(1920, 1080, bench.camera_for(0, synth, 1), aabb_scale=1)
    wf = torch.zeros((1080, 1920, 4), device="cuda:0"); wd = torch.zeros((1080, 1920), device="cuda:0")
    for _ in range(100):
        tb.render_with_params(tb.nerf_network, warm, wf, wd, None, None)
    torch.cuda.synchronize()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for step in range(16):
                fn(step % 8)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3 / 16)
        return best

    if what in ("whole", "all"):
        print("## whole images (ms per frame)\n")
        print("| resolution | " + " | ".join(f"teams {t}" for t in TEAMS) + " |")
        print("|---|" + "---|" * len(TEAMS))
        for W, H in ((480, 270), (640, 360), (960, 540), (1280, 720), (1600, 900), (1920, 1080), (2560, 1440)):
            frame = torch.zeros((H, W, 4), device="cuda:0"); depth = torch.zeros((H, W), device="cuda:0")
            row = []
            for team in TEAMS:
                ctx.set_lane_teams(team)

                def one(view):
                    p = synth.render_params(W, H, bench.camera_for(view, synth, 1), aabb_scale=1)
                    frame.zero_()
                    tb.render_with_params(tb.nerf_network, p, frame, depth, None, None)
                for v in range(8):
                    one(v)
                row.append(f"{timed(one):.3f}")
            print(f"| {W}x{H} | " + " | ".join(row) + " |")
    if what in ("tiles", "all"):
        print("\n## one rank's tiles of the 1080p frame (ms per share-frame)\n")
        print("| ranks | " + " | ".join(f"teams {t}" for t in TEAMS) + " |")
        print("|---|" + "---|" * len(TEAMS))
        W, H = 1920, 1080
        for N in (1, 2, 4, 8):
            sh = tiles.TileSharder(W, H, bench.TILE, 0, N, "cuda:0")
            row = []
            for team in TEAMS:
                ctx.set_lane_teams(team)

                def one(view):
                    p = synth.render_params(W, H, bench.camera_for(view, synth, 1), aabb_scale=1)
                    sh.fill(p)
                    sh.clear()
                    tb.render_with_params(tb.nerf_network, p, sh.local_frame, sh.local_depth, None, None)
                for v in range(8):
                    one(v)
                row.append(f"{timed(one):.3f}")
            print(f"| {N} | " + " | ".join(row) + " |")
    ctx.set_lane_teams(0)


if __name__ == "__main__":
    main()
