"""The lattice jump of the voxel walk (nrs_device.cuh: lattice_jump) under cameras chosen to stress its guard: many directions, views along the axes --
rays whose direction has a zero or tiny component (1 / d infinite or huge: the band B then refuses the jump and the cell-by-cell walk must take over) -- and both
places it runs: the fill's first hit (the (t, dt) stream of nrs_trace_samples: bit for bit against the oracle's cell-by-cell walk) and the per-round walk of the render
kernel (per-pixel sample counts and depth against the oracle).  tests/test_lattice.py checks the integer arithmetic on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _axis_camera(synth, axis, sign, jitter=0.0):
    """A camera on a coordinate axis of the scene looking at its centre: the central ray is parallel to that axis, its neighbours nearly so."""
    import math
    az = {0: 0.0, 1: 90.0, 2: 0.0}[axis] + (180.0 if sign < 0 else 0.0) + jitter
    el = 89.999 * sign if axis == 2 else jitter
    return synth.orbit_camera(az, el, scale=0.33)


def _views(synth):
    rng = np.random.default_rng(99)
    cams = [synth.orbit_camera(float(a), float(e), scale=0.33) for a, e in zip(rng.uniform(0, 360, 14), rng.uniform(-60, 75, 14))]
    for axis in range(3):
        for sign in (1, -1):
            cams.append(_axis_camera(synth, axis, sign))
            cams.append(_axis_camera(synth, axis, sign, jitter=0.02))   # a hundredth of a degree off the axis: 1 / d around 5000
    return cams


def test_first_hit_streams_over_many_cameras(rig):
    from test_gpu_parity import _trace_equal
    scene = rig.scene
    rig.use_edit(False)
    W, H = 192, 108
    hit = 0
    for k, cam in enumerate(_views(scene.synth)):
        p = scene.synth.render_params(W, H, cam, snap=(k % 2 == 0), spp_index=k)
        c = _trace_equal(rig, scene, p, W * H, 6)
        hit += int((c > 0).sum())
    assert hit > 50000


@pytest.mark.parametrize("edit", [False, True])
def test_frames_over_many_cameras(rig, edit):
    scene = rig.scene
    rig.use_edit(edit)
    try:
        W, H = 160, 90
        worst = 1.0
        for k, cam in enumerate(_views(scene.synth)[::2]):
            p = scene.synth.render_params(W, H, cam, snap=True)
            frame, depth, steps, stats = rig.render(p)
            ref_frame, ref_depth, ref_steps, ref_stats = scene.oracle_model.render(p, [scene.oracle_edit] if edit else [])
            assert stats.n_rays_alive == ref_stats.n_alive0, k
            ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
            assert ds.max() <= 1, k                      # (one sample apart only where alpha sits on the saturation threshold)
            worst = min(worst, float((ds == 0).mean()))
            same = (ds == 0) & (ref_frame[..., 3] > 0.2) & (frame[..., 3] > 0.2)
            assert np.allclose(depth[same], ref_depth[same], rtol=0, atol=2e-3), k
            assert np.abs(frame - ref_frame).max() < 1.5e-2, k
        assert worst >= 0.998
    finally:
        rig.use_edit(False)
