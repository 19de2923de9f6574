"""The marching accelerator (occupied box + 32^3 look-ahead mask, nrs_internal.h OccAccel) is built on the device from the bitfield
(launch_occ_accel).  It has no reference counterpart (the reference walks cell by cell, testbed_nerf.cu:1100-1131); what it must be is
CONSERVATIVE, and that is checked here against a numpy restatement of its definition: every occupied 2x2x2 Morton block of every
cascade, inflated by 1/16 cell, lies inside the box and inside blocks whose mask bit is set.  The box is also exact (min / max are
order-independent): bit-equal to numpy's.  Result parity of the walks that use it is test_gpu_parity.py's business."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G, VOL, COARSE = 128, 128 ** 3, 32


def _compact3(x):
    x = x & 0x49249249
    x = (x | (x >> 2)) & 0xc30c30c3
    x = (x | (x >> 4)) & 0x0f00f00f
    x = (x | (x >> 8)) & 0xff0000ff
    x = (x | (x >> 16)) & 0x0000ffff
    return x


def _blocks(bitfield, exact):
    """lo, hi [n, 3] float32 world bounds of the relevant occupied blocks (inflated)."""
    f32 = np.float32
    los, his = [], []
    for level in range(5):
        b = bitfield[level * VOL // 8:(level + 1) * VOL // 8]
        byte = np.nonzero(b)[0].astype(np.uint32)
        m = byte * np.uint32(8)
        c = np.stack([_compact3(m), _compact3(m >> 1), _compact3(m >> 2)], 1).astype(f32)
        s = f32(2.0 ** level)
        margin = f32(f32(s / f32(G)) / f32(16))
        a = (c / f32(G) - f32(0.5)) * s
        bb = ((c + f32(2)) / f32(G) - f32(0.5)) * s
        if exact and level > 0:
            far = np.maximum(np.abs(a - margin), np.abs(bb + margin)).max(1)
            keep = far >= f32(2.0 ** (level - 2))
            a, bb = a[keep], bb[keep]
        los.append(a + f32(0.5) - margin)
        his.append(bb + f32(0.5) + margin)
    return np.concatenate(los), np.concatenate(his)


def _check(net, bitfield):
    for which in (0, 1):
        box, bits = net.get_march_accelerator(which)
        lo, hi = _blocks(bitfield, exact=bool(which))
        if lo.shape[0] == 0:
            assert np.all(np.isinf(box[:6]))
            assert not bits.any()
            continue
        assert box[:3].tobytes() == lo.min(0).tobytes() and box[3:6].tobytes() == hi.max(0).tobytes()
        np.testing.assert_allclose(box[6:9], (box[3:6] - box[:3]) / COARSE, rtol=1e-6)
        np.testing.assert_allclose(box[9:12], 1.0 / box[6:9], rtol=1e-6)
        # the device's own index arithmetic, in float32: (w - min) * inv_cell, floor, clamp
        ilo = np.clip(np.floor((lo - box[:3]) * box[9:12]).astype(np.int64), 0, COARSE - 1)
        ihi = np.clip(np.floor((hi - box[:3]) * box[9:12]).astype(np.int64), 0, COARSE - 1)
        want = np.zeros((COARSE,) * 3, bool)
        uniq = np.unique(np.concatenate([ilo, ihi], 1), axis=0)
        for x0, y0, z0, x1, y1, z1 in uniq:
            want[z0:z1 + 1, y0:y1 + 1, x0:x1 + 1] = True
        assert np.array_equal(bits, want), f"flavour {which}: {np.count_nonzero(bits != want)} mask bits differ"


@pytest.mark.parametrize("fix", ["rig", "rig16"])
def test_accelerator_covers_every_occupied_block(request, fix):
    rig = request.getfixturevalue(fix)
    rig.net.set_density_bitfield(rig.scene.bitfield)
    _check(rig.net, rig.scene.bitfield)


def test_accelerator_edge_cases(rig):
    try:
        empty = np.zeros(5 * VOL // 8, np.uint8)
        rig.net.set_density_bitfield(empty)
        _check(rig.net, empty)
        one = empty.copy()
        one[12345] = 0x10  # a single cell: the box is one block, every mask bit inside it is set
        rig.net.set_density_bitfield(one)
        _check(rig.net, one)
        full = np.full(5 * VOL // 8, 0xff, np.uint8)
        rig.net.set_density_bitfield(full)
        _check(rig.net, full)
        rng = np.random.default_rng(5)
        sparse = (rng.random(5 * VOL // 8) < 1e-4).astype(np.uint8) * rng.integers(1, 256, 5 * VOL // 8).astype(np.uint8)
        rig.net.set_density_bitfield(sparse)
        _check(rig.net, sparse)
    finally:
        rig.net.set_density_bitfield(rig.scene.bitfield)


def test_accelerator_after_refresh(rig_shaped):
    """nrs_model_update_density_grid derives bitfield and accelerator in one stream, without a host pass."""
    rig = rig_shaped
    tb = rig.testbed
    saved = tb.edit_operators
    tb.edit_operators = []
    try:
        u = tb.new_grid_update(max_cascade=0, seed=7)
        u.reset_grid = 1
        tb.update_density_grid_nerf_operator(u)
        _check(rig.net, rig.net.get_density_bitfield())
    finally:
        tb.edit_operators = saved
        rig.net.set_density_bitfield(rig.scene.bitfield)
