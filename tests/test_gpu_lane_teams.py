"""Lane teams (render_kernel's TEAM; nrs_ctx_set_lane_teams): 2 or 4 lanes of a wavefront share one ray, evaluate
consecutive samples in the same round and composite them in marching order.  Nothing a caller can observe may depend on it:
frame, depth, step counts and statistics must be the bits of the one-lane-per-ray kernel, with and without edit
operators, for whole images and tiles, for constant and cone stepping."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(rig, w, h, az, **fields):
    p = rig.scene.params_for(w, h, az)
    for k, v in fields.items():
        setattr(p, k, v)
    return p


def _render_all(rig, p):
    out = {}
    try:
        for team in (1, 2, 4, -1, -2, -3, -4, 0):
            rig.ctx.set_lane_teams(team)
            out[team] = rig.render(p)
    finally:
        rig.ctx.set_lane_teams(0)
    return out


def _assert_same(out, what):
    ref = out[1]
    assert ref[3].n_samples > 0
    for team, (frame, depth, steps, stats) in out.items():
        assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)), f"{what}: frame differs with {team} lanes per ray"
        assert np.array_equal(depth.view(np.uint32), ref[1].view(np.uint32)), f"{what}: depth differs with {team} lanes per ray"
        assert np.array_equal(steps, ref[2]), f"{what}: step counts differ with {team} lanes per ray"
        assert (stats.n_samples, stats.n_rays_alive, stats.n_rays_hit) == (ref[3].n_samples, ref[3].n_rays_alive, ref[3].n_rays_hit), (what, team)


@pytest.mark.parametrize("edit", [False, True])
@pytest.mark.parametrize("size", [(256, 144), (200, 120), (640, 360)])
def test_teams_do_not_change_the_picture(rig, edit, size):
    rig.use_edit(edit)
    try:
        p = rig.scene.params_for(size[0], size[1], 60.0)
        _assert_same(_render_all(rig, p), f"{size} edit={edit}")
    finally:
        rig.use_edit(False)


def test_teams_with_low_opacity_and_step_cap(rig):
    """rays that never saturate (every sample of the team is consumed) and rays cut by max_march_steps inside a team"""
    p = _params(rig, 256, 144, 20.0, min_transmittance=1e-6)
    _assert_same(_render_all(rig, p), "min_transmittance 1e-6")
    for cap in (1, 2, 3, 5, 6, 7):
        p = _params(rig, 192, 108, 20.0, max_march_steps=cap)
        out = _render_all(rig, p)
        ref = out[1]
        for team, (frame, depth, steps, stats) in out.items():
            assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(steps, ref[2]), (cap, team)
            assert stats.n_samples == ref[3].n_samples, (cap, team)


def test_teams_in_cost_mode_and_with_jittered_pixels(rig):
    """ERenderMode::Cost (step counts as colour) and a non-zero spp index (Sobol pixel offsets, accumulation into the frame)"""
    from nerfshop_amd import _abi
    p = _params(rig, 256, 144, 70.0, render_mode=_abi.RENDER_COST)
    _assert_same(_render_all(rig, p), "cost mode")
    p = _params(rig, 256, 144, 70.0, spp_index=5, snap_to_pixel_centers=0)
    _assert_same(_render_all(rig, p), "spp 5, jittered")


def test_teams_on_tiles(rig):
    """one rank's tiles of a sharded frame (the case lane teams exist for), against the same tiles with one lane per ray"""
    rig.use_edit(True)
    try:
        for first, stride in ((0, 8), (3, 8), (1, 4)):
            p = _params(rig, 640, 360, 45.0, tile_size=64, tile_first=first, tile_stride=stride)  # (32-pixel tiles, the bench's: test_teams_on_small_tiles)
            torch = rig.torch
            got = {}
            try:
                for team in (1, 2, 4, -2, -3, -4, 0):
                    rig.ctx.set_lane_teams(team)
                    from nerfshop_amd import _abi
                    import ctypes as C
                    owned = _abi.load().nrs_render_owned_tiles(C.byref(p))
                    frame = torch.zeros((owned, 64, 64, 4), dtype=torch.float32, device="cuda:0")
                    depth = torch.zeros((owned, 64, 64), dtype=torch.float32, device="cuda:0")
                    st = rig.testbed.render_with_params(rig.net, p, frame, depth, None, None, want_stats=True)
                    torch.cuda.synchronize()
                    got[team] = (frame.cpu().numpy(), depth.cpu().numpy(), st.n_samples)
            finally:
                rig.ctx.set_lane_teams(0)
            assert got[1][2] > 0
            for team in (2, 4, -2, -3, -4, 0):
                assert np.array_equal(got[team][0].view(np.uint32), got[1][0].view(np.uint32)), (first, stride, team)
                assert np.array_equal(got[team][1].view(np.uint32), got[1][1].view(np.uint32)) and got[team][2] == got[1][2]
    finally:
        rig.use_edit(False)


def test_teams_aabb16(rig16):
    rig16.use_edit(True)
    try:
        p = rig16.scene.params_for(256, 144, 40.0)
        _assert_same(_render_all(rig16, p), "aabb 16, cone stepping")
    finally:
        rig16.use_edit(False)


def test_hybrid_at_1080p(rig):
    """the schedule bench.py runs: full generations, lane teams for the tail rows of the queue (every 8th packet row)"""
    rig.use_edit(True)
    try:
        p = rig.scene.params_for(1920, 1080, 75.0)
        out = {}
        try:
            for team in (1, -1, 0):
                rig.ctx.set_lane_teams(team)
                out[team] = rig.render(p)
        finally:
            rig.ctx.set_lane_teams(0)
        _assert_same(out, "1080p hybrid")
        assert out[1][3].n_samples > 10_000_000
    finally:
        rig.use_edit(False)


@pytest.mark.parametrize("size", [(8, 8), (9, 25), (64, 24), (173, 131), (7, 3), (331, 47)])
def test_odd_resolutions(rig, size):
    """packet geometry of every schedule at sizes that are not multiples of the packet shapes (8x8, 8x4, 4x4; tail rows of the hybrid)"""
    rig.use_edit(False)
    p = rig.scene.params_for(size[0], size[1], 35.0)
    out = _render_all(rig, p)
    ref = out[1]
    for team, (frame, depth, steps, stats) in out.items():
        assert np.array_equal(frame.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(depth.view(np.uint32), ref[1].view(np.uint32)), (size, team)
        assert np.array_equal(steps, ref[2]) and stats.n_samples == ref[3].n_samples and stats.n_rays_alive == ref[3].n_rays_alive, (size, team)


def test_teams_on_small_tiles(rig):
    """32-pixel tiles (bench.py's tile size) of a 1/8 and a 1/2 share, every schedule against one lane per ray"""
    import ctypes as C
    from nerfshop_amd import _abi
    torch = rig.torch
    for first, stride in ((5, 8), (1, 2)):
        p = _params(rig, 800, 450, 100.0, tile_size=32, tile_first=first, tile_stride=stride)
        owned = _abi.load().nrs_render_owned_tiles(C.byref(p))
        got = {}
        try:
            for team in (1, 2, 4, -2, -3, -4, 0):
                rig.ctx.set_lane_teams(team)
                frame = torch.zeros((owned, 32, 32, 4), dtype=torch.float32, device="cuda:0")
                depth = torch.zeros((owned, 32, 32), dtype=torch.float32, device="cuda:0")
                st = rig.testbed.render_with_params(rig.net, p, frame, depth, None, None, want_stats=True)
                torch.cuda.synchronize()
                got[team] = (frame.cpu().numpy(), depth.cpu().numpy(), st.n_samples)
        finally:
            rig.ctx.set_lane_teams(0)
        assert got[1][2] > 0
        for team in (2, 4, -2, -3, -4, 0):
            assert np.array_equal(got[team][0].view(np.uint32), got[1][0].view(np.uint32)), (first, stride, team)
            assert np.array_equal(got[team][1].view(np.uint32), got[1][1].view(np.uint32)) and got[team][2] == got[1][2]


def test_bad_team_size_is_refused(rig):
    from nerfshop_amd import _abi
    with pytest.raises(_abi.NrsError):
        rig.ctx.set_lane_teams(3)


@pytest.mark.parametrize("edit", [False, True])
def test_ray_handover_does_not_change_the_picture(rig, edit):
    """Waves that run out of work take rays from a sibling wave of their workgroup (pending ring entries, or half of the rays held in lanes):
    the hybrid whole-image schedule and the small-launch schedule, hand-over on and off -- identical bits, and the hand-over did happen."""
    rig.use_edit(edit)
    try:
        for size, team in (((960, 540), -1), ((640, 360), -2), ((1920, 1080), 0)):
            p = rig.scene.params_for(size[0], size[1], 60.0)
            out = {}
            moved = {}
            for on in (0, 1):
                rig.ctx.set_ray_handover(on)
                rig.ctx.set_lane_teams(team)
                out[on] = rig.render(p)
                moved[on] = rig.ctx.ray_handovers()
            assert moved[0] == (0, 0)
            assert moved[1][0] > 0 and moved[1][1] > 0, (size, team, moved)
            for i in range(3):
                assert np.array_equal(np.asarray(out[0][i]).view(np.uint32), np.asarray(out[1][i]).view(np.uint32)), (size, team, i)
            assert (out[0][3].n_samples, out[0][3].n_rays_alive, out[0][3].n_rays_hit) == (out[1][3].n_samples, out[1][3].n_rays_alive, out[1][3].n_rays_hit)
    finally:
        rig.ctx.set_ray_handover(1)
        rig.ctx.set_lane_teams(0)
        rig.use_edit(False)
