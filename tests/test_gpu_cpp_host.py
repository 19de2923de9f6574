"""A C++ host (examples/render_from_files.cpp: nrs_compat.hpp + the HIP runtime's hipMalloc, no Python and no PyTorch in its process) renders a scene from files in
the reference's formats; its frame equals the Python host's bit for bit.  The drop-in boundary is the C-ABI, whoever calls it."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "render_from_files")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    return EXE


def test_cpp_host_fails_loudly_without_inputs(exe, tmp_path):
    r = subprocess.run([exe, str(tmp_path / "missing.ingp"), "-", "64", "36", "0.69", str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert r.returncode == 1 and "render_from_files:" in r.stderr and not (tmp_path / "o.raw").exists()


@pytest.mark.gpu
def test_cpp_host_matches_python_host_bit_for_bit(exe, rig, tmp_path):
    from nerfshop_amd import formats, runtime, synth
    scene = rig.scene
    W, H = 256, 144
    camera = scene.camera(60.0)
    formats.save_snapshot(tmp_path / "scene.ingp", scene.desc, 1, scene.params, scene.edited_grid, camera=camera)
    formats.save_edits(tmp_path / "edits.json", [scene.edit])
    r = subprocess.run([exe, str(tmp_path / "scene.ingp"), str(tmp_path / "edits.json"), str(W), str(H), repr(synth.CAMERA_ANGLE_X), str(tmp_path / "o.raw")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    import json
    stats = json.loads(r.stdout)
    assert stats["operators"] == 1 and stats["n_rays_hit"] > 1000
    raw = np.fromfile(tmp_path / "o.raw", np.float32)
    frame, depth = raw[:W * H * 4].reshape(H, W, 4), raw[W * H * 4:].reshape(H, W)

    # the Python host on the same files, same Testbed defaults as nrs::compat::Testbed (m_poisson_target true)
    snap = formats.load_snapshot(tmp_path / "scene.ingp")
    tb = runtime.Testbed(rig.ctx, snap.desc, snap.aabb_scale)
    tb.nerf_network.set_params(snap.params)
    tb.nerf_network.set_density_grid(snap.density_grid)
    tb.add_edit_operator(runtime.CageDeformation(rig.ctx, snap.desc, formats.load_edits(tmp_path / "edits.json")[0], device_authoring=True))
    p = synth.render_params(W, H, snap.camera)
    p.poisson_target = 1
    torch = rig.torch
    py_frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    py_depth = torch.zeros((H, W), dtype=torch.float32, device="cuda:0")
    tb.render_with_params(tb.nerf_network, p, py_frame, py_depth, None, None, want_stats=True)
    torch.cuda.synchronize()
    assert stats["n_samples"] == tb.last_stats.n_samples and stats["n_rays_hit"] == tb.last_stats.n_rays_hit
    assert np.array_equal(frame, py_frame.cpu().numpy())
    assert np.array_equal(depth, py_depth.cpu().numpy())


@pytest.mark.gpu
def test_cpp_host_reads_the_reference_written_edits_file(exe, rig, tmp_path):
    """Same program, fed the edits file the reference's own Testbed::save_edits wrote (tests/golden/ref_edits_golden.json.gz): two cage deformations, an affine duplication and a cage without interpolation mesh."""
    import gzip
    import json
    from nerfshop_amd import formats
    scene = rig.scene
    formats.save_snapshot(tmp_path / "scene.ingp", scene.desc, 1, scene.params, scene.edited_grid, camera=scene.camera(60.0))
    (tmp_path / "ref_edits.json").write_bytes(gzip.open(os.path.join(ROOT, "tests", "golden", "ref_edits_golden.json.gz"), "rb").read())
    r = subprocess.run([exe, str(tmp_path / "scene.ingp"), str(tmp_path / "ref_edits.json"), "128", "72", "0.69", str(tmp_path / "o.raw")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    stats = json.loads(r.stdout)
    assert stats["operators"] == 3 and stats["n_rays_hit"] > 300   # 4 in the file; the one saved before its cage was tetrahedralised has nothing to apply
    assert np.isfinite(np.fromfile(tmp_path / "o.raw", np.float32)).all()
