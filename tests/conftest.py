import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """CPU-side prerequisites: the oracle .so (g++) and libnrs.so (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from nerfshop_amd import _abi
    from oracle import oracle as orc
    if not os.path.exists(_abi.LIB_PATH) or not os.path.exists(orc.LIB_PATH):
        g.build()


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return True


class Scene:
    """Synthetic lego-like scene (nerfshop_amd.synth) + the CPU oracle's view of it."""

    def __init__(self, aabb_scale=1, with_edit=True, lattice_n=6, shaped=False, **desc_kw):
        from nerfshop_amd import synth
        from oracle import oracle as orc
        self.synth, self.orc = synth, orc
        self.aabb_scale = aabb_scale
        self.desc = synth.model_desc(aabb_scale, **desc_kw)  # (desc_kw: the other members of configs/nerf/'s family, tests/test_gpu_architectures.py)
        self.params = synth.make_params(self.desc, sigma_raw=synth.default_sigma_raw(aabb_scale), shaped=shaped, aabb_scale=aabb_scale)
        self.grid = synth.density_grid(aabb_scale)
        self.bitfield = synth.grid_to_bitfield(self.grid)
        self.oracle_model = orc.Model(self.desc, self.params, self.bitfield)
        self.edit = None
        self.oracle_edit = None
        self.edited_bitfield = None
        if with_edit:
            scale = 1.0 if aabb_scale == 1 else 6.0
            self.edit = synth.make_cage_edit(lattice_n=lattice_n, scene_scale=scale)
            self.oracle_edit = orc.Edit(self.desc, self.edit.tet_mesh_struct(), keepalive=self.edit)
            grid2 = synth.deformed_density_grid(self.grid, self.desc, self.oracle_edit.map_positions, aabb_scale)
            self.edited_grid = grid2
            self.edited_bitfield = synth.grid_to_bitfield(grid2)

    def camera(self, azimuth=30.0, elevation=30.0):
        scale = 0.33 if self.aabb_scale == 1 else 0.33 * 6.0
        return self.synth.orbit_camera(azimuth, elevation, scale=scale)

    def params_for(self, w, h, azimuth=30.0, **kw):
        return self.synth.render_params(w, h, self.camera(azimuth), aabb_scale=self.aabb_scale, **kw)


@pytest.fixture(scope="session")
def scene(built):
    return Scene(aabb_scale=1, with_edit=True, lattice_n=6)


@pytest.fixture(scope="session")
def scene16(built):
    return Scene(aabb_scale=16, with_edit=True, lattice_n=5)


@pytest.fixture(scope="session")
def scene_shaped(built):
    """Geometry inside the network (synth.make_params(shaped=True)): what the occupancy refresh needs."""
    return Scene(aabb_scale=1, with_edit=True, lattice_n=6, shaped=True)


@pytest.fixture(scope="session")
def scene16_shaped(built):
    return Scene(aabb_scale=16, with_edit=True, lattice_n=5, shaped=True)


class GpuRig:
    """libnrs objects for a Scene on cuda:0."""

    def __init__(self, scene):
        import torch
        from nerfshop_amd import runtime
        assert torch.cuda.is_available(), "gpu tests need a GPU"
        self.torch, self.rt, self.scene = torch, runtime, scene
        self.ctx = runtime.Context(0)
        self.testbed = runtime.Testbed(self.ctx, scene.desc, scene.aabb_scale)
        self.net = self.testbed.nerf_network
        self.net.set_params(scene.params)
        self.net.set_density_bitfield(scene.bitfield)
        self.op = runtime.CageDeformation(self.ctx, scene.desc, scene.edit) if scene.edit is not None else None

    def use_edit(self, on):
        self.testbed.edit_operators = [self.op] if (on and self.op is not None) else []
        self.net.set_density_bitfield(self.scene.edited_bitfield if on else self.scene.bitfield)
        self.scene.oracle_model.set_bitfield(self.scene.edited_bitfield if on else self.scene.bitfield)

    def render(self, p, want_steps=True):
        torch = self.torch
        W, H = p.resolution[0], p.resolution[1]
        frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
        depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
        steps = torch.zeros((H, W), dtype=torch.int32, device="cuda:0") if want_steps else None
        stats = self.testbed.render_with_params(self.net, p, frame, depth, steps, None, want_stats=True)
        torch.cuda.synchronize()
        return frame.cpu().numpy(), depth.cpu().numpy(), (steps.cpu().numpy() if steps is not None else None), stats


@pytest.fixture(scope="session")
def rig(scene):
    return GpuRig(scene)


@pytest.fixture(scope="session")
def rig16(scene16):
    return GpuRig(scene16)


@pytest.fixture(scope="session")
def rig_shaped(scene_shaped):
    return GpuRig(scene_shaped)


@pytest.fixture(scope="session")
def rig16_shaped(scene16_shaped):
    return GpuRig(scene16_shaped)
