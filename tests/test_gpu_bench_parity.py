"""Full-size parity of the workloads bench.py times: the bench's OWN scenes (bench.build_scene: cage lattice 10, the
occupancy refreshed through the HIP operator; density noise 1.5 for the varied-opacity scene) on the bench's own cameras, rendered with the automatic
schedule, against the oracle.  Bars as in tests/test_gpu_numerics.py::test_1080p_bench_view_against_the_oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class BenchScene:
    """bench.build_scene(workload) on cuda:0 + the oracle's view of the same scene."""

    def __init__(self, workload):
        import torch
        import bench
        from nerfshop_amd import runtime as rt, synth
        from oracle import oracle as orc
        self.torch, self.bench, self.synth = torch, bench, synth
        self.ctx = rt.Context(0)
        sc = bench.build_scene(workload, rt, synth, self.ctx, torch)
        self.sc, self.tb = sc, sc["tb"]
        bitfield = synth.grid_to_bitfield(sc["grid"])
        # the occupancy the device derived from the float grid (threshold + pooling on the GPU) is the one the oracle marches through: same bits
        assert np.array_equal(self.tb.nerf_network.get_density_bitfield(), bitfield)
        self.model = orc.Model(sc["desc"], sc["params"], bitfield)
        self.edits = [orc.Edit(sc["desc"], sc["edit"].tet_mesh_struct(), keepalive=sc["edit"])] if sc["edit"] is not None else []

    def params(self, step, w=1920, h=1080, apply_operators=True):
        return self.synth.render_params(w, h, self.bench.camera_for(step, self.synth, self.sc["aabb_scale"]), aabb_scale=self.sc["aabb_scale"], apply_operators=apply_operators)

    def render(self, p):
        torch = self.torch
        W, H = p.resolution[0], p.resolution[1]
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
        steps = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        stats = self.tb.render_with_params(self.tb.nerf_network, p, frame, depth, steps, None, want_stats=True)
        torch.cuda.synchronize()
        return frame.cpu().numpy(), depth.cpu().numpy(), steps.cpu().numpy(), stats


def check_against_oracle(bs, p, edits, min_samples, depth_scale=1.0):
    frame, depth, steps, stats = bs.render(p)
    ref_frame, ref_depth, ref_steps, ref_stats = bs.model.render(p, edits)
    assert ref_stats.composited > min_samples
    assert stats.n_rays_alive == ref_stats.n_alive0
    d = np.abs(frame - ref_frame).max(axis=-1)
    n_px = d.size
    # at most 1e-5 of the pixels (and never fewer than 3 allowed) may sit on the other side of the alpha normalisation (|d| <= 1.01e-2), the rest within 6e-3
    assert d.max() < 1.5e-2 and (d > 6e-3).sum() <= max(3, 1e-5 * n_px) and float(np.abs(frame - ref_frame).mean()) < 2e-6, (d.max(), (d > 6e-3).sum(), np.abs(frame - ref_frame).mean())
    ds = np.abs(steps.astype(np.int64) - ref_steps.astype(np.int64))
    assert ds.max() <= 1 and (ds == 0).mean() >= 0.9998, (ds.max(), (ds == 0).mean())
    hit = (ref_frame[..., 3] > 0.2) & (frame[..., 3] > 0.2) & (ds == 0)
    assert np.allclose(depth[hit], ref_depth[hit], rtol=0, atol=2e-3 * depth_scale)
    assert abs(int(stats.n_samples) - int(ref_stats.composited)) <= 0.0003 * ref_stats.composited
    return stats


@pytest.fixture(scope="module")
def lego_cage(built):
    return BenchScene("lego_cage")


@pytest.mark.parametrize("step", [0, 3])
def test_bench_scene_1080p_views_against_the_oracle(lego_cage, step):
    """`value`'s workload itself: BASELINE configs[2], bench scene (lattice 10), bench views 0 and 3 at 1920x1080."""
    check_against_oracle(lego_cage, lego_cage.params(step), lego_cage.edits, 10_000_000)


def test_bench_scene_noedit_1080p_against_the_oracle(lego_cage):
    """The bench's `noedit` key (BASELINE configs[1]): the same scene and occupancy with apply_operators off, view 1, 1920x1080."""
    check_against_oracle(lego_cage, lego_cage.params(1, apply_operators=False), [], 10_000_000)


def test_varied_opacity_scene_against_the_oracle(built):
    """`lego_cage_varied` (geometry in the network, density noise 1.5): ray lengths vary widely, which is where re-teaming, thinning generations and the
    ray hand-over fire most -- 960x540 against the oracle with the automatic schedule, hand-over on, and the hand-over must have happened."""
    bs = BenchScene("lego_cage_varied")
    check_against_oracle(bs, bs.params(0, 960, 540), bs.edits, 3_000_000)
    rays, handovers = bs.ctx.ray_handovers()
    assert rays > 0 and handovers > 0, (rays, handovers)
    # and the same picture with the hand-over off (the schedule must not show in the bits)
    f1 = bs.render(bs.params(2, 960, 540))
    bs.ctx.set_ray_handover(False)
    try:
        f0 = bs.render(bs.params(2, 960, 540))
    finally:
        bs.ctx.set_ray_handover(True)
    assert np.array_equal(f1[0].view(np.uint32), f0[0].view(np.uint32)) and np.array_equal(f1[2], f0[2])
