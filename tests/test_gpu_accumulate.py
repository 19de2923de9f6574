"""nrs_accumulate = CudaRenderBuffer::accumulate (src/render_buffer.cu:540-560, accumulate_kernel :217-254) on an MI355X against the oracle, which
tests/test_ref_pin.py pins bit for bit to the reference's compiled accumulate_kernel: Linear and VisPosNeg are plain fp32 in the reference's order -- bit-exact; SRGB goes
through powf (device library here, glibc in the oracle, CUDA's in the reference): 2e-6 absolute.  Then the offline configuration (BASELINE configs[1], scripts/run.py's
8 spp): 8 frames with the Sobol pixel offsets of spp_index 0..7 accumulated on the GPU against the same 8 oracle frames accumulated by the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("color_space,exact", [(0, True), (1, False), (2, True)])
def test_accumulate_against_the_oracle(rig, color_space, exact):
    import ctypes as C
    torch = rig.torch
    from nerfshop_amd import _abi
    from oracle import oracle as orc
    lib = _abi.load()
    H, W = 53, 77   # ragged against the 256-thread workgroups
    rng = np.random.default_rng(color_space)
    acc = torch.full((H, W, 4), 3.0, dtype=torch.float32, device="cuda:0")   # garbage that frame 0 must overwrite
    ref = np.full((H, W, 4), 9.0, np.float32)
    for k in range(6):
        f = rng.uniform(0.0, 1.5, (H, W, 4)).astype(np.float32)
        f[::3, ::5, :3] *= np.float32(1e-3)
        ft = torch.from_numpy(f).cuda()
        _abi.check(lib.nrs_accumulate(rig.ctx.h, None, W, H, ft.data_ptr(), acc.data_ptr(), k, color_space))
        torch.cuda.synchronize()
        orc.accumulate(f, ref, k, color_space)
        got = acc.cpu().numpy()
        if exact:
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (k, np.abs(got - ref).max())
        else:
            assert np.abs(got - ref).max() < 2e-6, (k, np.abs(got - ref).max())
            assert np.array_equal(got[..., 3].view(np.uint32), ref[..., 3].view(np.uint32))   # alpha takes no sRGB curve
    with pytest.raises(_abi.NrsError):
        _abi.check(lib.nrs_accumulate(rig.ctx.h, None, W, H, acc.data_ptr(), acc.data_ptr(), 0, 3))


def test_eight_spp_offline_frame(rig):
    """8 spp as scripts/run.py renders its test images: snap_to_pixel_centers off... the reference's run.py snaps AND accumulates 8 identical offsets; the viewer's
    accumulation uses the Sobol offsets -- both are `spp_index = k` frames joined by accumulate(): done here with the Sobol offsets (the harder case)."""
    from nerfshop_amd import runtime
    from oracle import oracle as orc
    rig.use_edit(True)
    try:
        W, H = 160, 90
        buf = runtime.RenderBuffer(W, H)
        ref_acc = np.zeros((H, W, 4), np.float32)
        for k in range(8):
            p = rig.scene.params_for(W, H, 60.0, snap=False, spp_index=k)
            buf.clear_frame()
            rig.testbed.render_with_params(rig.net, p, buf.frame_buffer(), buf.depth_buffer(), None, None)
            assert buf.spp() == k
            acc = buf.accumulate(rig.ctx)
            ref_frame = rig.scene.oracle_model.render(p, [rig.scene.oracle_edit])[0]
            orc.accumulate(ref_frame, ref_acc, k, 0)
        rig.torch.cuda.synchronize()
        got = acc.cpu().numpy()
        d = np.abs(got - ref_acc)
        assert d.max() < 6e-3 and d.mean() < 2e-4, (d.max(), d.mean())   # the Shade bar: a mean of 8 frames that each meet it
        one = rig.render(rig.scene.params_for(W, H, 60.0, snap=False, spp_index=0))[0]
        assert np.abs(got - one).max() > 1e-3   # anti-aliasing did something: 8 offsets are not 1
    finally:
        rig.use_edit(False)
