"""The cell-record cache (nrs_model_set_cell_cache) is a layout of the same numbers: everything that reads the hash grid
must return the same bits with the records, without them, and with any number of cached levels -- inside the unit cube,
on its faces and outside it (where the kernels gather the native way)."""
import ctypes as C

import numpy as np
import pytest

from nerfshop_amd import _abi

pytestmark = pytest.mark.gpu


def _level_cells(desc):
    res = (C.c_uint32 * 16)()
    _abi.check(_abi.load().nrs_model_level_table(C.byref(desc), None, res, None, None, None))
    return [int(r) ** 3 for r in res]


def _encode(rig, coords):
    torch = rig.torch
    out = torch.zeros((coords.shape[0], 32), dtype=torch.float16, device="cuda:0")
    rig.net.hashgrid_encode(None, torch.from_numpy(coords).cuda(), out)
    return out.cpu().numpy().view(np.uint16)


def _coords(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    c = rng.uniform(lo, hi, size=(n, 7)).astype(np.float32)
    c[:, 3:] = 0.5
    return c


def test_plan_follows_the_budget(rig):
    cells = _level_cells(rig.scene.desc)
    default_bytes, default_levels = rig.net.cell_cache()
    try:
        assert default_levels == 12 and default_bytes == 32 * sum(cells[:12])   # 9.3 GB: levels 0..11 of base.json's table
        for budget, want in [(0, 0), (31, 0), (32 * sum(cells[:2]) - 1, 0), (32 * sum(cells[:2]), 2), (32 * sum(cells[:5]), 4),
                             (32 * sum(cells[:8]) + 5, 8)]:
            rig.net.set_cell_cache(budget)
            assert rig.net.cell_cache() == (32 * sum(cells[:want]), want), budget
    finally:
        rig.net.set_cell_cache(10 << 30)
    assert rig.net.cell_cache() == (default_bytes, default_levels)


def test_features_identical_with_and_without_records(rig):
    inside = _coords(50000, 5, 0.0, 1.0)
    inside[:8, :3] = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)   # corners of the cube
    inside[8:2008, :3] = np.round(inside[8:2008, :3] * 64) / 64                                          # cell boundaries of many levels
    around = _coords(50000, 6, -0.75, 1.75)        # most samples outside the cube: waves fall back to the native gathers
    around[::7, :3] = _coords(50000, 8, 0.0, 1.0)[::7, :3]
    far = _coords(4096, 7, -300.0, 300.0)
    ref_in = rig.scene.oracle_model.hashgrid_encode(inside[:3000])
    ref_out = rig.scene.oracle_model.hashgrid_encode(around[:3000])
    try:
        got = {}
        for budget in (10 << 30, 0, 300 << 20, 3 << 30):
            rig.net.set_cell_cache(budget)
            got[budget] = [_encode(rig, c) for c in (inside, around, far)]
            assert np.array_equal(got[budget][0][:3000], ref_in) and np.array_equal(got[budget][1][:3000], ref_out), rig.net.cell_cache()
        for budget, arrays in got.items():
            for a, b in zip(arrays, got[0]):
                assert np.array_equal(a, b), f"features differ between a {budget}-byte cache and none"
    finally:
        rig.net.set_cell_cache(10 << 30)


def test_frames_identical_with_and_without_records(rig):
    rig.use_edit(True)
    try:
        p = rig.scene.params_for(512, 288, 60.0)
        frames = {}
        for budget in (10 << 30, 0, 1 << 30):
            rig.net.set_cell_cache(budget)
            frames[budget] = rig.render(p)
        ref = frames[0]
        assert ref[3].n_samples > 100000
        for budget, (frame, depth, steps, stats) in frames.items():
            assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)), budget
            assert np.array_equal(depth.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(steps, ref[2])
            assert stats.n_samples == ref[3].n_samples
    finally:
        rig.net.set_cell_cache(10 << 30)
        rig.use_edit(False)


def test_records_follow_set_params(rig):
    """nrs_model_set_params rebuilds the records: new parameters must show up through the cached levels at once."""
    c = _coords(20000, 9, 0.0, 1.0)
    before = _encode(rig, c)
    rng = np.random.default_rng(10)
    other = rig.scene.params.copy()
    n_net = other.size - 2 * sum(_level_entries(rig.scene.desc))
    grid = rng.uniform(-0.3, 0.3, size=other.size - n_net).astype(np.float16)
    other[n_net:] = grid.view(np.uint16)
    try:
        rig.net.set_params(other)
        with_records = _encode(rig, c)
        rig.net.set_cell_cache(0)
        without = _encode(rig, c)
        assert np.array_equal(with_records, without)
        assert (with_records != before).mean() > 0.5
    finally:
        rig.net.set_cell_cache(10 << 30)
        rig.net.set_params(rig.scene.params)
    assert np.array_equal(_encode(rig, c), before)


def _level_entries(desc):
    cnt = (C.c_uint32 * 16)()
    _abi.check(_abi.load().nrs_model_level_table(C.byref(desc), None, None, None, cnt, None))
    return [int(v) for v in cnt]


def test_occupancy_refresh_identical_with_and_without_records(rig_shaped):
    """The density-grid refresh evaluates the network through the same gathers."""
    rig = rig_shaped
    grids = []
    try:
        for budget in (10 << 30, 0):
            rig.net.set_cell_cache(budget)
            rig.net.set_density_grid(rig.scene.grid)
            u = rig.testbed.new_grid_update(max_cascade=0, seed=1337)
            rig.testbed.update_density_grid_nerf_render(2, True, u)
            rig.torch.cuda.synchronize()
            grids.append(rig.net.get_density_grid().copy())
        assert np.array_equal(grids[0].view(np.uint32), grids[1].view(np.uint32))
    finally:
        rig.net.set_cell_cache(10 << 30)
        rig.net.set_density_bitfield(rig.scene.bitfield)


def _samples_in_mask(scene, mask, n_per_cascade, seed):
    """warped positions inside marked cells of every cascade of `mask` (density-bitfield layout)"""
    from nerfshop_amd import synth
    bits = np.unpackbits(mask, bitorder="little").reshape(5, -1)
    rng = np.random.default_rng(seed)
    x, y, z = synth._cell_coords()
    mn = np.array(scene.desc.aabb_min[:])
    ext = np.array(scene.desc.aabb_max[:]) - mn
    out = []
    for lvl in range(5):
        idx = np.nonzero(bits[lvl])[0]
        if idx.size == 0:
            continue
        pick = rng.choice(idx, n_per_cascade)
        p = (np.stack([x[pick], y[pick], z[pick]], 1) + rng.uniform(0, 1, (n_per_cascade, 3))) / 128.0
        p = (p - 0.5) * 2.0 ** lvl + 0.5
        w = (p - mn) / ext
        out.append(w[((w >= 0) & (w <= 1)).all(1)])
    return np.concatenate(out).astype(np.float32)


def test_sparse_brick_records_are_a_layout_of_the_same_numbers(rig16):
    """nrs_model_set_sparse_cell_cache (aabb-16 scenes: the levels after the dense records get occupancy-sparse 8^3-cell bricks of records):
    features, frames, depth, steps and statistics are identical bit for bit with the sparse records, without them, with a partial budget, and
    with a mask that does NOT cover the samples (those lanes fall back to the hashed gathers)."""
    rig, scene = rig16, rig16.scene
    mask = scene.bitfield | scene.edited_bitfield
    c = _coords(150000, 21, 0.0, 1.0)
    inside = _samples_in_mask(scene, mask, 20000, 3)
    c[:inside.shape[0], :3] = inside
    c[-8:, :3] = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    rig.use_edit(True)
    try:
        rig.net.set_sparse_cell_cache(None, 0)
        assert rig.net.sparse_cell_cache() == (0, 0, 0)
        base = _encode(rig, c)
        assert np.array_equal(base[:3000], scene.oracle_model.hashgrid_encode(c[:3000]))
        p = scene.params_for(384, 216, 60.0)
        ref = rig.render(p)
        assert ref[3].n_samples > 100000
        dense_levels = rig.net.cell_cache()[1]
        wrong_mask = np.zeros_like(mask)
        wrong_mask[:1000] = 0xff  # a few cells near one corner of cascade 0: almost no sample has records
        for m, budget, want_levels in ((mask, 4 << 30, 2), (mask, 80 << 30, 4), (wrong_mask, 4 << 30, None), (mask, 1 << 20, 0)):
            rig.net.set_sparse_cell_cache(m, budget)
            nbytes, first, n = rig.net.sparse_cell_cache()
            if want_levels is not None:
                assert n == want_levels and (first == dense_levels or n == 0) and nbytes <= budget, (nbytes, first, n)
            assert np.array_equal(_encode(rig, c), base), (budget, n)
            frame, depth, steps, stats = rig.render(p)
            assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(depth.view(np.uint32), ref[1].view(np.uint32))
            assert np.array_equal(steps, ref[2]) and stats.n_samples == ref[3].n_samples
        # every launch above ran the GATE instantiation (cone stepping, 2 / 3 / 4 trailing hashed level pairs behind records: two, three or four L2 phases); without any
        # records six pairs are hashed and nrs_render_nerf takes the default kernel: the ungated picture is the same picture, bit for bit
        rig.net.set_cell_cache(0)
        assert rig.net.cell_cache()[1] == 0 and rig.net.sparse_cell_cache() == (0, 0, 0)
        frame, depth, steps, stats = rig.render(p)
        assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(depth.view(np.uint32), ref[1].view(np.uint32))
        assert np.array_equal(steps, ref[2]) and stats.n_samples == ref[3].n_samples
        rig.net.set_cell_cache(10 << 30)
        # the records follow the parameters, and the dense budget drops the sparse levels (they start where the dense ones end)
        rig.net.set_sparse_cell_cache(mask, 4 << 30)
        other = scene.params.copy()
        n_net = other.size - 2 * sum(_level_entries(scene.desc))
        other[n_net:] = np.random.default_rng(4).uniform(-0.3, 0.3, size=other.size - n_net).astype(np.float16).view(np.uint16)
        rig.net.set_params(other)
        with_sparse = _encode(rig, c)
        rig.net.set_cell_cache(0)
        assert rig.net.sparse_cell_cache() == (0, 0, 0)
        assert np.array_equal(_encode(rig, c), with_sparse) and (with_sparse != base).mean() > 0.5
    finally:
        rig.net.set_cell_cache(10 << 30)
        rig.net.set_params(scene.params)
        rig.use_edit(False)


def test_sparse_records_on_dense_levels_with_a_mask_that_misses(rig16):
    """with no dense record cache (set_cell_cache(0)) the sparse levels start at level 0, i.e. they include DENSE levels; a lane whose
    brick has no records must then gather with the level's own (dense) index function, not the hashed one.  Arbitrary positions (the network
    operators evaluate anywhere), a mask that allocates almost nothing, and one that covers half the scene."""
    rig, scene = rig16, rig16.scene
    c = _coords(120000, 33, 0.0, 1.0)
    c[-8:, :3] = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    rig.use_edit(False)
    try:
        rig.net.set_sparse_cell_cache(None, 0)
        rig.net.set_cell_cache(0)
        assert rig.net.cell_cache()[1] == 0
        base = _encode(rig, c)
        assert np.array_equal(base[:4000], scene.oracle_model.hashgrid_encode(c[:4000]))
        wrong_mask = np.zeros_like(scene.bitfield)
        wrong_mask[:1000] = 0xff
        half_mask = scene.bitfield.copy()
        half_mask[half_mask.size // 10:] = 0
        for m in (wrong_mask, half_mask, scene.bitfield):
            rig.net.set_sparse_cell_cache(m, 2 << 30)
            nbytes, first, n = rig.net.sparse_cell_cache()
            assert first == 0 and n >= 2, (nbytes, first, n)   # dense levels (0..) carry sparse records now
            got = _encode(rig, c)
            assert np.array_equal(got, base), f"{(got != base).any(1).sum()} of {c.shape[0]} samples differ"
        p = scene.params_for(256, 144, 60.0)
        with_sparse = rig.render(p)
        rig.net.set_sparse_cell_cache(None, 0)
        plain = rig.render(p)
        assert np.array_equal(with_sparse[0].view(np.uint32), plain[0].view(np.uint32)) and np.array_equal(with_sparse[2], plain[2])
    finally:
        rig.net.set_sparse_cell_cache(None, 0)
        rig.net.set_cell_cache(10 << 30)
        rig.use_edit(False)


def test_set_params_from_a_device_pointer(rig):
    """nrs_model_set_params_device (NerfNetworkFull::set_params takes device pointers, nerf_network_full.h:316-349): the same model state as the host
    entry point -- features, network outputs and the cell records rebuilt from it."""
    torch = rig.torch
    c = _coords(30000, 12, 0.0, 1.0)
    before = _encode(rig, c)
    other = rig.scene.params.copy()
    rng = np.random.default_rng(13)
    other[:] = rng.uniform(-0.4, 0.4, size=other.size).astype(np.float16).view(np.uint16)
    try:
        rig.net.set_params(other)
        want_feat = _encode(rig, c)
        want = torch.zeros((16, c.shape[0]), dtype=torch.float16, device="cuda:0")
        rig.net.inference_mixed_precision(None, torch.from_numpy(c).cuda(), want)
        rig.net.set_params(rig.scene.params)
        assert np.array_equal(_encode(rig, c), before)
        d = torch.from_numpy(other.view(np.int16)).cuda()
        rig.net.set_params_device(d)
        got = torch.zeros_like(want)
        rig.net.inference_mixed_precision(None, torch.from_numpy(c).cuda(), got)
        assert np.array_equal(_encode(rig, c), want_feat) and torch.equal(got.view(torch.int16), want.view(torch.int16))
        # asynchronous on a side stream: nothing waits between the hand-over and the launch that uses it; the blob is scribbled over right behind it
        side = torch.cuda.Stream()
        rig.net.set_params(rig.scene.params)
        torch.cuda.synchronize()
        d2 = d.clone()
        cin = torch.from_numpy(c).cuda()
        got2 = torch.zeros_like(want)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            rig.net.set_params_device(d2, stream=side)
            rig.net.inference_mixed_precision(side, cin, got2)
            d2.zero_()
        side.synchronize()
        assert torch.equal(got2.view(torch.int16), want.view(torch.int16))
        with pytest.raises(_abi.NrsError):
            rig.net.set_params_device(d[:-2])
    finally:
        rig.net.set_params(rig.scene.params)
