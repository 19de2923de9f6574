"""Image-tile sharding of one frame across the GPUs of a node (the reference is single-GPU, README.md:423).

Rays are independent, so the only exchange step is the final gather of the rendered tiles.  The image is cut into
tile x tile pixel tiles indexed t = Ty * pitch + Tx with an ODD row pitch (tile_pitch) and dealt round-robin: rank r owns tiles r, r + N, r + 2N, ...
(lego rays are spatially clustered; an interleave balances the load where contiguous strips would not, and the odd pitch makes it a diagonal one).  Each rank
renders its tiles into a COMPACT buffer [tiles_per_rank, tile, tile, C] (nrs_render_params.tile_*); ONE exchange step
collects them on rank 0 and nrs_detile scatters them back into the W x H image.  On GPUs the exchange is libnrs's own
nrs_gather_tiles (host C++: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL over xGMI, include/nrs.h) on a
communicator bootstrapped through one torch.distributed broadcast of the 128-byte id; on CPU (the gloo tests) it is a
torch.distributed gather.  The model is replicated (24-27 MB of parameters plus the optional cell records: 9.2 GB by default, up to 68 GB with the sparse brick records of an aabb-16 scene); nothing else is exchanged per frame.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi


def tile_pitch(width, tile):
    """row pitch of the tile index (include/nrs.h): odd, so that `index mod ranks` deals diagonals rather than columns; indices beyond the image's
    last tile column are virtual (they own no pixels)"""
    return ((width + tile - 1) // tile) | 1


def tile_counts(width, height, tile, world):
    tiles_x = tile_pitch(width, tile)
    tiles_y = (height + tile - 1) // tile
    total = tiles_x * tiles_y
    per_rank = [(total - r + world - 1) // world if r < total else 0 for r in range(world)]
    return tiles_x, tiles_y, total, per_rank


def detile_index(width, height, tile, world, padded):
    """For every pixel of the full image, the flat pixel index inside the rank-major gathered buffer
    [world, padded, tile, tile].  (CPU twin of the nrs_detile kernel, used by the gloo tests.)"""
    tiles_x = tile_pitch(width, tile)
    y = torch.arange(height).view(-1, 1).expand(height, width)
    x = torch.arange(width).view(1, -1).expand(height, width)
    T = (y // tile) * tiles_x + (x // tile)
    r, k = T % world, T // world
    return (((r * padded + k) * tile + (y % tile)) * tile + (x % tile)).reshape(-1)


_COMM = {}  # (device index, world) -> nrs_comm handle: one RCCL communicator per process, shared by every TileSharder


def _nrs_comm(device, rank, world):
    """libnrs's RCCL communicator for this process (created on first use: rank 0 makes the id, torch.distributed broadcasts it)."""
    key = (device.index or 0, world)
    if key not in _COMM:
        lib = _abi.load()
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _abi.check(lib.nrs_comm_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        ident = ident.to(device)
        dist.broadcast(ident, src=0)
        raw = bytes(ident.cpu().tolist())
        h = C.c_void_p()
        ok = 1
        try:
            _abi.check(lib.nrs_comm_create(device.index or 0, rank, world, raw, C.byref(h)))
        except _abi.NrsError as e:
            import sys
            print(f"[nerfshop_amd.tiles] rank {rank}: nrs_comm_create failed ({e})", file=sys.stderr)
            ok = 0
        # every rank must take the same path: the C-ABI gather is used only if every rank has a communicator
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if ok:
                lib.nrs_comm_destroy(h)
            h = None
        _COMM[key] = h
    if _COMM[key] is None:
        raise _abi.NrsError("no RCCL communicator on some rank")
    return _COMM[key]


_VERIFIED = {}  # (device index, world) -> True once the C-ABI gather has reproduced torch.distributed.gather on this process's first frame


class TileSharder:
    def __init__(self, width, height, tile, rank, world, device):
        if tile % 8:
            raise ValueError("tile size must be a multiple of 8 (ray packets are 8x8 pixels)")
        self.width, self.height, self.tile, self.rank, self.world = width, height, tile, rank, world
        self.device = torch.device(device)
        self.tiles_x, self.tiles_y, self.total, self.per_rank = tile_counts(width, height, tile, world)
        self.padded = max(self.per_rank)  # equal-sized contributions: one collective, no ragged sends
        # frame and depth of a rank live in ONE buffer [frame block | depth block] so that a single gather moves both
        n_px = self.padded * tile * tile
        self.local = torch.zeros(n_px * 5, dtype=torch.float32, device=self.device)
        self.local_frame = self.local[: n_px * 4].view(self.padded, tile, tile, 4)
        self.local_depth = self.local[n_px * 4:].view(self.padded, tile, tile)
        if rank == 0:
            self.all = torch.zeros((world, n_px * 5), dtype=torch.float32, device=self.device)
        else:
            self.all = None
        self._index = None
        # GPUs: the C-ABI gather (RCCL inside libnrs).  NRS_GATHER=torch keeps torch.distributed.gather (also the fallback if RCCL cannot be loaded).
        import os
        self.gather_impl = "torch.distributed"
        self.comm = None
        if self.device.type == "cuda" and world > 1 and dist.is_initialized() and os.environ.get("NRS_GATHER", "nrs") != "torch":
            try:
                self.comm = _nrs_comm(self.device, rank, world)
                self.gather_impl = "nrs_gather_tiles (RCCL send/recv, C-ABI)"
            except _abi.NrsError as e:
                import sys
                print(f"[nerfshop_amd.tiles] nrs_comm_create failed ({e}); falling back to torch.distributed.gather", file=sys.stderr)

    def _c_gather(self, ctx, p, frame, depth):
        lib = _abi.load()
        s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        root = self.rank == 0
        _abi.check(lib.nrs_gather_tiles(ctx.h, self.comm, 0, C.byref(p), self.padded, self.local.data_ptr(), self.all.data_ptr() if root else None,
                                        frame.data_ptr() if root else None, depth.data_ptr() if root else None, s))

    def _verify_c_gather(self, ctx, p):
        """First frame of the process: the C-ABI gather must reproduce torch.distributed.gather + nrs_detile bit for bit on the root, and every
        rank must get through it; otherwise all ranks fall back together (the verdict is agreed with one all_reduce)."""
        ok = 1
        if self.rank == 0:
            f1 = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=self.device)
            d1 = torch.zeros((self.height, self.width), dtype=torch.float32, device=self.device)
            f2, d2 = torch.zeros_like(f1), torch.zeros_like(d1)
        else:
            f1 = d1 = f2 = d2 = None
        try:
            self._c_gather(ctx, p, f1, d1)
            self._wait_bounded(torch.cuda.current_stream(self.device), "the first nrs_gather_tiles")
        except _abi.NrsError as e:
            import sys
            print(f"[nerfshop_amd.tiles] rank {self.rank}: nrs_gather_tiles failed ({e})", file=sys.stderr)
            ok = 0
        comm, self.comm = self.comm, None  # every rank takes part in the torch.distributed gather, whatever happened above
        try:
            self.gather(ctx, p, f2, d2)
        finally:
            self.comm = comm
        if self.rank == 0 and ok:
            torch.cuda.current_stream(self.device).synchronize()
            ok = int(torch.equal(f1, f2) and torch.equal(d1, d2))
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    @staticmethod
    def _wait_bounded(stream, what, seconds=120.0):
        """stream.synchronize() with a deadline: if a peer failed before it enqueued its half of a send / receive pair, the matching operation here
        never completes -- fail loudly instead of hanging the job (the RCCL operation itself cannot be cancelled: the process must exit)."""
        import time
        t0 = time.monotonic()
        while not stream.query():
            if time.monotonic() - t0 > seconds:
                raise RuntimeError(f"nerfshop_amd.tiles: {what} did not complete within {seconds:.0f} s (a peer rank failed?)")
            time.sleep(0.0005)

    def fill(self, p):
        """Write the sharding fields of an nrs_render_params."""
        p.tile_size, p.tile_first, p.tile_stride = self.tile, self.rank, self.world
        return p

    def clear(self):
        self.local.zero_()

    def gather(self, ctx, p, frame, depth):
        """One gather to rank 0, then de-tile there.  `frame` [H, W, 4] / `depth` [H, W] are written on rank 0."""
        if self.comm is not None:
            key = (self.device.index or 0, self.world)
            if key not in _VERIFIED:
                _VERIFIED[key] = self._verify_c_gather(ctx, p)
            if _VERIFIED[key]:
                self._c_gather(ctx, p, frame, depth)
                return
            self.gather_impl = "torch.distributed (the C-ABI gather failed its first-frame check)"
        if self.world > 1:
            if self.device.type == "cuda" and dist.get_backend() == "gloo":
                # gloo gathers host tensors only: stage through the host (bench.py's one-GPU mode NRS_BENCH_DIST=gloo and its first-frame cross-check of the
                # C-ABI gather; never the timed path of a real multi-GPU job, whose backend is nccl)
                loc = self.local.cpu()
                parts = [torch.empty_like(loc) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(loc, parts, dst=0)
                if self.rank == 0:
                    self.all.copy_(torch.stack(parts))
            else:
                dist.gather(self.local, list(self.all.unbind(0)) if self.rank == 0 else None, dst=0)
        elif self.rank == 0:
            self.all[0].copy_(self.local)
        if self.rank != 0:
            return
        n_px = self.padded * self.tile * self.tile
        if self.device.type == "cuda":
            # the gathered buffer is [rank][frame block | depth block]: de-tile each block with its rank stride
            lib = _abi.load()
            s = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _abi.check(lib.nrs_detile(ctx.h, s, C.byref(p), self.world, self.padded, self.all.data_ptr(), 4, n_px * 5, frame.data_ptr()))
            _abi.check(lib.nrs_detile(ctx.h, s, C.byref(p), self.world, self.padded, self.all.data_ptr() + n_px * 16, 1, n_px * 5, depth.data_ptr()))
        else:
            if self._index is None:
                self._index = detile_index(self.width, self.height, self.tile, self.world, self.padded)
            all_frame = self.all[:, : n_px * 4].reshape(-1, 4)
            all_depth = self.all[:, n_px * 4:].reshape(-1)
            frame.view(-1, 4).copy_(all_frame[self._index])
            depth.view(-1).copy_(all_depth[self._index])
