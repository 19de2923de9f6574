"""the selection tool's ray shooting (GrowingSelection::project_selection_pixels,
growing_selection.cu:1832-2035) through nrs_project_selection_pixels, against the oracle's three-step restatement.
The product stops a ray at its answer and evaluates one sample per ray and round; the oracle collects all samples, runs
the network on them and composites, as the reference does."""
import numpy as np
import pytest

from nerfshop_amd import _abi, synth

pytestmark = pytest.mark.gpu


def _scribble(w, h, n, seed):
    rng = np.random.default_rng(seed)
    # a few strokes across the object plus stray pixels (background, image border)
    t = rng.uniform(0, 1, size=n)
    x = (0.5 + 0.28 * np.cos(7 * t) * t) * w
    y = (0.5 + 0.28 * np.sin(5 * t) * t) * h
    px = np.stack([x, y], 1).astype(np.int32)
    px[:8] = [[0, 0], [w - 1, h - 1], [w // 2, h // 2], [w // 2, 0], [0, h // 2], [w - 1, 0], [3, h - 2], [w // 2 + 1, h // 2]]
    return px


@pytest.mark.parametrize("which,threshold", [("rig", 0.1), ("rig", 1e-3), ("rig16", 0.1)])
def test_projection_matches_oracle(request, which, threshold):
    rig = request.getfixturevalue(which)
    rig.use_edit(False)
    w, h = 640, 360
    p = rig.scene.params_for(w, h, 50.0)
    px = _scribble(w, h, 3000, 3)
    (pos, cells, found), _ = rig.testbed.project_selection_pixels(p, px, threshold)
    ref_pos, ref_cells, ref_found = rig.scene.oracle_model.project_selection_pixels(p, px, threshold)
    assert 0.2 < ref_found.mean() < 1.0          # strokes on the object, strays off it
    # the density MLP runs on MFMA (fp32 accumulation in its own order, fast exp): a transmittance within rounding of the
    # threshold may cross it one sample earlier or later.  Everything else must be the same bits.
    same = (found == ref_found) & (cells == ref_cells) & (pos.view(np.uint32) == ref_pos.view(np.uint32)).all(1)
    assert same.mean() > 0.995, f"{(~same).sum()} of {same.size} rays differ"
    off = ~same & (found == 1) & (ref_found == 1)
    if off.any():   # a neighbouring sample of the same ray: at most two steps away
        step = np.sqrt(3.0) / 1024.0 * (16.0 if which == "rig16" else 1.0)
        d = np.linalg.norm(pos[off] - ref_pos[off], axis=1)
        assert (d < 2.5 * step * np.linalg.norm([1.0, 1.0, 1.0])).all()
    # not found -> the marker outside the box, exactly
    mn = np.asarray(rig.scene.desc.aabb_min, np.float32)
    assert np.array_equal(pos[found == 0], np.broadcast_to(mn - np.float32(1.0), pos[found == 0].shape))


def test_projected_points_lie_on_the_surface(rig):
    """domain property: a projected point is inside an occupied cell of the cascade it reports, on the ray of its pixel"""
    rig.use_edit(False)
    w, h = 640, 360
    p = rig.scene.params_for(w, h, 20.0)
    px = _scribble(w, h, 2000, 5)
    (pos, cells, found), (sel_cells, sel_pos, level) = rig.testbed.project_selection_pixels(p, px)
    f = found == 1
    assert f.sum() > 300
    vol = 128 ** 3
    bits = np.unpackbits(rig.scene.bitfield, bitorder="little")
    assert bits[cells[f]].all()
    mip = cells[f] // vol
    assert (mip == 0).all()                                   # an aabb-scale-1 scene lives in cascade 0
    cell_xyz = np.floor(pos[f] * 128).astype(np.int64).clip(0, 127)
    assert np.array_equal(synth.morton3d(cell_xyz[:, 0], cell_xyz[:, 1], cell_xyz[:, 2]).astype(np.uint32), cells[f] % vol)
    # bookkeeping: unique cells, in pixel order, each represented by the first pixel that reached it
    assert level == 0 and len(set(sel_cells.tolist())) == len(sel_cells) == len(set(cells[f].tolist()))
    first = {}
    for i in np.nonzero(f)[0]:
        first.setdefault(int(cells[i]), i)
    assert sel_cells.tolist() == list(first.keys())
    assert np.array_equal(sel_pos, pos[list(first.values())])
