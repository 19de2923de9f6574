"""Reads the raw per-wave log of the profiling instantiation (NRS_DEV_KNOBS=1 NRS_DEBUG=4 NRS_WAVE_LOG_FILE=path: written by nrs_render_nerf after a launch with statistics)
and prints how the launch's time is distributed over waves, SIMDs and CUs.  Record per wave (4 x u64): [0] lifetime in shader cycles | rays << 48,
[1] rounds | rounds before the queue was found dry << 16 | time of that << 32, [2] packets | HW_ID[15:0] << 16 | fill cycles >> 8 << 32 | xcc << 56,
[3] wall-clock lifetime (10 ns ticks) | start tick << 32.
    python tools/wave_log_report.py gpurun_out/wave.bin"""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
    raw = raw[raw[:, 0] != 0]
    life = (raw[:, 0] & np.uint64((1 << 48) - 1)).astype(np.float64)
    rays = (raw[:, 0] >> np.uint64(48)).astype(np.int64)
    rounds = (raw[:, 1] & np.uint64(0xffff)).astype(np.int64)
    packets = (raw[:, 2] & np.uint64(0xffff)).astype(np.int64)
    hw = ((raw[:, 2] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64)
    fill = (((raw[:, 2] >> np.uint64(32)) & np.uint64(0xffffff)).astype(np.float64)) * 256.0
    xcc = (raw[:, 2] >> np.uint64(56)).astype(np.int64)
    wall = (raw[:, 3] & np.uint64(0xffffffff)).astype(np.float64) / 100.0        # us
    start = (raw[:, 3] >> np.uint64(32)).astype(np.float64)
    start = (start - start.min()) / 100.0
    end = start + wall
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    simd_key = cu_key * 4 + simd
    n = len(end)
    print(f"waves {n}, last end {end.max():.1f} us, mean end {end.mean():.1f} us ({100 * end.mean() / end.max():.1f} %), start spread {start.max():.1f} us")
    print(f"rounds per wave: mean {rounds.mean():.1f}, max {rounds.max()}; packets: mean {packets.mean():.1f}, max {packets.max()}")
    busy = rounds > 0
    us_per_round = (wall[busy] - fill[busy] / life[busy] * wall[busy]) / rounds[busy]
    print(f"fill share of a wave's life: mean {100 * (fill[busy] / life[busy]).mean():.1f} %; us per round (life minus fill): "
          f"p5 {np.percentile(us_per_round, 5):.1f}, p50 {np.percentile(us_per_round, 50):.1f}, p95 {np.percentile(us_per_round, 95):.1f}")

    def group(key, name):
        keys = np.unique(key)
        ends = np.array([end[key == k].max() for k in keys])
        tot_rounds = np.array([rounds[key == k].sum() for k in keys])
        tot_rays = np.array([rays[key == k].sum() for k in keys])
        tot_packets = np.array([packets[key == k].sum() for k in keys])
        nw = np.array([(key == k).sum() for k in keys])
        print(f"\n{name}: {len(keys)} units, waves per unit {nw.min()}..{nw.max()}; last end per unit: min {ends.min():.1f}, mean {ends.mean():.1f}, max {ends.max():.1f} us "
              f"(mean / max = {100 * ends.mean() / ends.max():.1f} %)")
        for label, v in (("rounds", tot_rounds), ("packets", tot_packets)):
            c = np.corrcoef(v, ends)[0, 1] if v.std() > 0 else float("nan")
            print(f"   {label} per unit: min {v.min()}, mean {v.mean():.1f}, max {v.max()}; correlation with the unit's end {c:+.2f}")
        order = np.argsort(ends)
        for tag, i in (("earliest", order[0]), ("median", order[len(order) // 2]), ("latest", order[-1])):
            sel = key == keys[i]
            if sel.sum() <= 16:
                print(f"   {tag} unit: end {ends[i]:.1f} us, waves {sel.sum()}, rounds {rounds[sel].tolist()}, packets {packets[sel].tolist()}")
            else:
                print(f"   {tag} unit: end {ends[i]:.1f} us, waves {sel.sum()}, rounds in all {rounds[sel].sum()}, packets in all {packets[sel].sum()}")

    group(simd_key, "SIMDs")
    group(cu_key, "CUs")
    group(xcc, "XCDs")


if __name__ == "__main__":
    main()
