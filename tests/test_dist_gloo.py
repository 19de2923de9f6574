"""N > 1 path on CPU: world_size-2 and world_size-8 gloo processes exercise the tile sharding + gather + de-tile host logic of
nerfshop_amd.tiles exactly as bench.py drives it on GPUs (there the backend is RCCL and the renderer the HIP kernel;
here the backend is gloo and each rank's tiles come from the CPU oracle, which honours the same nrs_render_params
tile fields)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pack_compact(full, sharder):
    """full-resolution per-rank image (zeros outside the rank's tiles) -> the compact [padded, tile, tile, C] layout the HIP
    kernel writes directly."""
    t = sharder.tile
    out = torch.zeros_like(sharder.local_frame if full.dim() == 3 else sharder.local_depth)
    k = 0
    for T in range(sharder.rank, sharder.total, sharder.world):
        ty, tx = divmod(T, sharder.tiles_x)
        blk = full[ty * t:(ty + 1) * t, tx * t:(tx + 1) * t]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
        k += 1
    return out


def _worker(rank, world, port, W, H, tile, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerfshop_amd import synth, tiles
        from oracle import oracle as orc
        desc = synth.model_desc(1)
        params = synth.make_params(desc)
        bitfield = synth.grid_to_bitfield(synth.density_grid(1))
        model = orc.Model(desc, params, bitfield)
        sharder = tiles.TileSharder(W, H, tile, rank, world, "cpu")
        p = synth.render_params(W, H, synth.orbit_camera(30.0))
        sharder.fill(p)
        f, d, s, st = model.render(p, n_threads=2)           # this rank's tiles only
        sharder.local_frame.copy_(_pack_compact(torch.from_numpy(f), sharder))
        sharder.local_depth.copy_(_pack_compact(torch.from_numpy(d), sharder))
        frame = torch.zeros((H, W, 4))
        depth = torch.zeros((H, W))
        sharder.gather(None, p, frame, depth)
        total = torch.tensor([float(st.composited)], dtype=torch.float64)
        dist.all_reduce(total)                                # whole-job sample count, as bench.py aggregates it
        if rank == 0:
            p0 = synth.render_params(W, H, synth.orbit_camera(30.0))
            f0, d0, _, st0 = model.render(p0, n_threads=2)
            ok = bool(np.array_equal(frame.numpy(), f0)) and bool(np.array_equal(depth.numpy()[f0[..., 3] > 0], d0[f0[..., 3] > 0]))
            q.put((ok, int(total.item()) == int(st0.composited), sharder.per_rank, sharder.padded))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,tile", [(2, 96, 56, 16), (2, 100, 60, 24), (8, 120, 72, 16)])
def test_tile_shard_gather(built, world, W, H, tile):
    """world 8 = the configuration BASELINE configs[4] names; (120, 72, 16) gives 8 x 5 tiles on a pitch of 9: 45 indices, 5-6 per rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, tile, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    ok, samples_ok, per_rank, padded = q.get(timeout=10)
    assert ok and samples_ok
    # every index of the (odd-pitch) tile grid is owned exactly once: ceil(W / tile) | 1 columns, of which those beyond the image are virtual
    assert sum(per_rank) == (((W + tile - 1) // tile) | 1) * ((H + tile - 1) // tile) and padded == max(per_rank)


def test_detile_index_is_a_bijection_on_owned_pixels():
    from nerfshop_amd import tiles
    for W, H, t, n in [(64, 64, 16, 4), (100, 60, 24, 3), (1920, 1080, 64, 8)]:
        tx, ty, total, per = tiles.tile_counts(W, H, t, n)
        idx = tiles.detile_index(W, H, t, n, max(per))
        assert idx.numel() == W * H and idx.unique().numel() == W * H
        assert int(idx.max()) < n * max(per) * t * t
        assert sum(per) == total and max(per) - min(per) <= 1
