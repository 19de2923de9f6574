"""What the exchange step of a sharded frame costs next to the frame (VERDICT r3 next #1c), on ONE MI355X:
  (1) the host side and the device side of nrs_gather_tiles at the root with one rank (the root's own copy + nrs_detile of frame and depth);
  (2) RCCL's own point-to-point path -- ncclGroupStart / (ncclSend + ncclRecv) x k / ncclGroupEnd -- for k = 1, 3, 7 transfers of one rank's 1/8-share
      buffer (5.2 MB) from the rank to itself (nrs_comm_probe_self_p2p): host time to enqueue and time to completion.  RCCL refuses two ranks on one
      device, so the wire is not measured here (xGMI: 5.2 MB at ~50 GB/s effective per link ~ 0.1 ms, the seven links of the root in parallel).
  (3) one rank's 1/8 share-frame for comparison.
usage: python tools/gather_probe.py > gpurun_out/gather_probe.md"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from nerfshop_amd import _abi, runtime as rt, synth, tiles
    lib = _abi.load()
    ctx = rt.Context(0)
    W, H, T, N = 1920, 1080, bench.TILE, 8
    dev = torch.device("cuda", 0)
    # a one-rank communicator on the real RCCL
    buf = (C.c_uint8 * 128)()
    _abi.check(lib.nrs_comm_unique_id(buf))
    comm = C.c_void_p()
    _abi.check(lib.nrs_comm_create(0, 0, 1, bytes(buf), C.byref(comm)))
    info = [C.c_int(), C.c_int(), C.c_int()]
    path = C.create_string_buffer(256)
    _abi.check(lib.nrs_comm_info(comm, C.byref(info[0]), C.byref(info[1]), C.byref(info[2]), path, 256))
    print("# The exchange step next to a share-frame, one MI355X (round 4)\n")
    print(f"RCCL {info[2].value} ({path.value.decode()}); 1920x1080, {T}-pixel tiles, N = {N}: one rank's buffer = frame block + depth block.\n")
    sh8 = tiles.TileSharder(W, H, T, 0, N, dev)
    n_floats = sh8.local.numel()
    print(f"one rank's buffer: {n_floats * 4 / 1e6:.2f} MB\n")
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        host = (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / reps
        return host * 1e6, total * 1e6

    print("| step | host time to enqueue, us | time per call incl. completion (back to back), us |")
    print("|---|---|---|")
    # (2) RCCL self send / recv
    src = torch.zeros(7 * n_floats, dtype=torch.float32, device=dev)
    dst = torch.zeros(7 * n_floats, dtype=torch.float32, device=dev)
    for k in (1, 3, 7):
        h, t = timed(lambda: _abi.check(lib.nrs_comm_probe_self_p2p(comm, src.data_ptr(), dst.data_ptr(), n_floats, k, sp)))
        print(f"| RCCL group of {k} send/recv pair(s) of {n_floats * 4 / 1e6:.1f} MB (rank to itself) | {h:.1f} | {t:.1f} |")
    # (1) nrs_gather_tiles with one rank: the root's copy + de-tile of a whole frame (N = 1) -- the de-tile cost of the full image
    sh1 = tiles.TileSharder(W, H, T, 0, 1, dev)
    p = synth.render_params(W, H, bench.camera_for(0, synth, 1), aabb_scale=1)
    sh1.fill(p)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    depth = torch.zeros((H, W), dtype=torch.float32, device=dev)
    h, t = timed(lambda: _abi.check(lib.nrs_gather_tiles(ctx.h, comm, 0, C.byref(p), sh1.padded, sh1.local.data_ptr(), sh1.all.data_ptr(), frame.data_ptr(), depth.data_ptr(), sp)))
    print(f"| nrs_gather_tiles at the root, one rank: device-to-device copy of its block ({sh1.local.numel() * 4 / 1e6:.0f} MB) + nrs_detile of frame and depth (whole 1080p image) | {h:.1f} | {t:.1f} |")
    h, t = timed(lambda: (_abi.check(lib.nrs_detile(ctx.h, sp, C.byref(p), 1, sh1.padded, sh1.all.data_ptr(), 4, sh1.local.numel(), frame.data_ptr())),
                          _abi.check(lib.nrs_detile(ctx.h, sp, C.byref(p), 1, sh1.padded, sh1.all.data_ptr() + sh1.padded * T * T * 16, 1, sh1.local.numel(), depth.data_ptr()))))
    print(f"| nrs_detile x 2 alone (frame + depth, whole image) | {h:.1f} | {t:.1f} |")
    # (3) the share-frame itself
    scene = bench.build_scene("lego_cage", rt, synth, ctx, torch)
    tb = scene["tb"]
    ps = [synth.render_params(W, H, bench.camera_for(k, synth, 1), aabb_scale=1) for k in range(8)]
    for q in ps:
        sh8.fill(q)
    state = {"k": 0}

    def share():
        q = ps[state["k"] % 8]
        state["k"] += 1
        with torch.cuda.stream(s):
            sh8.clear()
            tb.render_with_params(tb.nerf_network, q, sh8.local_frame, sh8.local_depth, None, s)
    h, t = timed(share, reps=64)
    print(f"| one rank's 1/8 share-frame: clear + nrs_render_nerf (one frame at a time) | {h:.1f} | {t:.1f} |")
    lib.nrs_comm_destroy(comm)


if __name__ == "__main__":
    main()
