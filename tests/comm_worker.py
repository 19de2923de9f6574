"""One rank of tests/test_gpu_comm_multiproc.py: renders its tiles of a frame on cuda:0 and takes part in nrs_gather_tiles -- with NRS_RCCL_LIB set to
tests/fake_rccl/libfake_rccl.so, so that N ranks can share the one GPU of the box.  usage: comm_worker.py <rank> <world> <port> <result file>"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    from conftest import GpuRig, Scene
    from nerfshop_amd import _abi, tiles
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # the id's channel only (gloo carries the 128 bytes; the tiles go through nrs_gather_tiles)
    lib = _abi.load()
    rig = GpuRig(Scene(aabb_scale=1, with_edit=True, lattice_n=6))
    rig.use_edit(True)
    res = {"rank": rank}
    W, H, T = 200, 120, 64          # 4 x 2 = 8 tiles over 3 ranks: 3 / 3 / 2 (ragged), not a multiple of the tile size
    p = rig.scene.params_for(W, H, 60.0)
    whole, whole_depth, _, _ = rig.render(p)
    sh = tiles.TileSharder(W, H, T, rank, world, "cuda:0")
    assert sh.comm is not None, "no communicator"
    info = [C.c_int(), C.c_int(), C.c_int()]
    path = C.create_string_buffer(256)
    _abi.check(lib.nrs_comm_info(sh.comm, C.byref(info[0]), C.byref(info[1]), C.byref(info[2]), path, 256))
    res["comm"] = {"rank": info[0].value, "n_ranks": info[1].value, "version": info[2].value, "lib": path.value.decode()}
    assert res["comm"]["rank"] == rank and res["comm"]["n_ranks"] == world and "fake_rccl" in res["comm"]["lib"]
    tiles._VERIFIED[(0, world)] = True  # (the first-frame cross-check against torch.distributed.gather needs a backend that gathers CUDA tensors; the check here is the whole image)
    sh.fill(p)
    sh.clear()
    rig.testbed.render_with_params(rig.net, p, sh.local_frame, sh.local_depth, None, None)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    # 1) TileSharder.gather -> nrs_gather_tiles, root 0: ncclSend on ranks 1.., ncclRecv x (N - 1) at rank-major offsets on rank 0, de-tile
    for rep in range(2):  # twice: the second exchange reuses the channels (message matching per pair)
        sh.gather(rig.ctx, p, frame, depth)
        torch.cuda.synchronize()
    assert "nrs_gather_tiles" in sh.gather_impl
    hit = whole[..., 3] > 0
    if rank == 0:
        res["root0_frame_equal"] = bool(np.array_equal(frame.cpu().numpy().view(np.uint32), whole.view(np.uint32)))
        res["root0_depth_equal"] = bool(np.array_equal(depth.cpu().numpy()[hit], whole_depth[hit]))
        res["pixels_hit"] = int(hit.sum())
    # 2) the C-ABI directly with root = world - 1 (root != 0: the root's own block is copied, its peers are received)
    root = world - 1
    n_px = sh.padded * T * T
    recv = torch.zeros((world, n_px * 5), dtype=torch.float32, device="cuda:0") if rank == root else None
    f2 = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    d2 = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    _abi.check(lib.nrs_gather_tiles(rig.ctx.h, sh.comm, root, C.byref(p), sh.padded, sh.local.data_ptr(), recv.data_ptr() if recv is not None else None,
                                    f2.data_ptr() if rank == root else None, d2.data_ptr() if rank == root else None, None))
    torch.cuda.synchronize()
    if rank == root:
        res["rootN_frame_equal"] = bool(np.array_equal(f2.cpu().numpy().view(np.uint32), whole.view(np.uint32)))
        res["rootN_depth_equal"] = bool(np.array_equal(d2.cpu().numpy()[hit], whole_depth[hit]))
        # the received blocks are the peers' buffers verbatim (rank-major): compare this rank's own slot with what it rendered
        res["rootN_own_block"] = bool(torch.equal(recv[rank], sh.local))
    dist.barrier()
    json.dump(res, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
