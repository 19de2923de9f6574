"""nrs_comm_* / nrs_gather_tiles (host C++, RCCL dlopen'ed by libnrs): what can be exercised on ONE GPU -- the bootstrap (unique id, communicator of
one rank), the root's own leg of the gather and the de-tile of both blocks -- against the whole-image render, bit for bit.  The N > 1 legs
(ncclSend / ncclRecv between ranks) need N GPUs: the driver's multi-GPU bench is their test; tests/test_dist_gloo.py covers the sharding logic."""
import ctypes as C

import numpy as np
import pytest

from nerfshop_amd import _abi, tiles

pytestmark = pytest.mark.gpu


def test_gather_tiles_one_rank_matches_whole_image(rig):
    torch = rig.torch
    lib = _abi.load()
    ident = (C.c_uint8 * 128)()
    _abi.check(lib.nrs_comm_unique_id(ident))
    assert any(ident)
    comm = C.c_void_p()
    _abi.check(lib.nrs_comm_create(0, 0, 1, bytes(ident), C.byref(comm)))
    rig.use_edit(True)
    try:
        W, H = 200, 120   # not a multiple of the 64-pixel tile
        p = rig.scene.params_for(W, H, 60.0)
        whole, whole_depth, _, _ = rig.render(p)
        sh = tiles.TileSharder(W, H, 64, 0, 1, "cuda:0")
        sh.fill(p)
        sh.clear()
        rig.testbed.render_with_params(rig.net, p, sh.local_frame, sh.local_depth, None, None)
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
        _abi.check(lib.nrs_gather_tiles(rig.ctx.h, comm, 0, C.byref(p), sh.padded, sh.local.data_ptr(), sh.all.data_ptr(), frame.data_ptr(), depth.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), whole.view(np.uint32))
        hit = whole[..., 3] > 0
        assert np.array_equal(depth.cpu().numpy()[hit], whole_depth[hit])
        # argument checking
        with pytest.raises(_abi.NrsError):
            _abi.check(lib.nrs_gather_tiles(rig.ctx.h, comm, 3, C.byref(p), sh.padded, sh.local.data_ptr(), sh.all.data_ptr(), None, None, None))
        with pytest.raises(_abi.NrsError):
            _abi.check(lib.nrs_comm_create(0, 2, 2, None, C.byref(C.c_void_p())))
    finally:
        lib.nrs_comm_destroy(comm)
        rig.use_edit(False)
