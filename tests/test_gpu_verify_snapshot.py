"""tools/verify_snapshot.py on the synthetic files nerfshop_amd.formats writes -- the reference's snapshot (.ingp) and edits (.json)
schemas plus a NeRF-synthetic style transforms file with ground-truth images: the one command that somebody with a real lego snapshot would run.
The "ground truth" here is the HIP renderer's own picture written as an 8-bit PNG, so the PSNR leg must come out high (quantisation only)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_verify_snapshot_on_synthetic_files(rig, tmp_path):
    from PIL import Image
    from nerfshop_amd import formats, synth
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import verify_snapshot as vs
    scene = rig.scene
    formats.save_snapshot(tmp_path / "scene.ingp", scene.desc, 1, scene.params, scene.edited_grid, camera=scene.camera(60.0))
    formats.save_edits(tmp_path / "edits.json", [scene.edit])
    # a transforms file whose camera is the bench orbit's (the inverse of nerf_matrix_to_ngp) and whose image is what the renderer shows from there
    W = H = 160
    frames = []
    rig.use_edit(True)
    try:
        for k, az in enumerate((30.0, 200.0)):
            cam = synth.orbit_camera(az, 30.0)                        # NGP 3x4, column-major flat
            ngp = np.asarray(cam, np.float64).reshape(4, 3).T         # 3x4
            m = ngp[[2, 0, 1], :].copy()                              # undo the axis cycle
            m[:, 3] = (m[:, 3] - 0.5) / 0.33
            m[:, 1] *= -1
            m[:, 2] *= -1
            c2w = np.vstack([m, [0, 0, 0, 1]])
            assert np.allclose(np.ascontiguousarray(synth.nerf_matrix_to_ngp(c2w[:3], 0.33).T.reshape(-1), np.float32), cam, atol=1e-6)
            p = synth.render_params(W, H, cam)
            p.min_transmittance = 1e-4
            frame = rig.render(p)[0].astype(np.float64)               # premultiplied linear RGBA on black
            rgb = np.divide(frame[..., :3], frame[..., 3:4], out=np.zeros_like(frame[..., :3]), where=frame[..., 3:4] > 0)
            png = np.concatenate([np.clip(vs.linear_to_srgb(rgb), 0, 1), frame[..., 3:4]], axis=2)
            os.makedirs(tmp_path / "test", exist_ok=True)
            Image.fromarray((png * 255.0 + 0.5).astype(np.uint8), "RGBA").save(tmp_path / "test" / f"r_{k}.png")
            frames.append({"file_path": f"./test/r_{k}", "transform_matrix": c2w.tolist()})
    finally:
        rig.use_edit(False)
    json.dump({"camera_angle_x": synth.CAMERA_ANGLE_X, "frames": frames}, open(tmp_path / "transforms_test.json", "w"))
    out = tmp_path / "report.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_snapshot.py"), str(tmp_path / "scene.ingp"), "--edits", str(tmp_path / "edits.json"),
                        "--transforms", str(tmp_path / "transforms_test.json"), "--max-views", "2", "--out", str(out)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.load(open(out))
    assert rep["parity_all_views"] and len(rep["views"]) == 2 and rep["edits"]["operators"] == 1
    for v in rep["views"]:
        for mode in ("fp32_fp32 (default)", "network_fp16 (tiny-cuda-nn as recalled)"):
            rec = v[mode]
            assert rec["parity"] and rec["samples"] > 10000 and rec["max_sample_count_difference"] <= 1
        assert v["fp32_fp32 (default)"]["psnr"] > 38.0            # the picture it was made from, through 8-bit sRGB
        assert v["network_fp16 (tiny-cuda-nn as recalled)"]["psnr"] > 30.0


@pytest.mark.parametrize("kw", [dict(no_dir=True), dict(rgb_hidden_layers=1), dict(rgb_hidden_layers=3)], ids=["base_nodir", "base_1layer", "base_3layer"])
def test_verify_snapshot_on_the_network_family(built, tmp_path, kw):
    """The same command on snapshots of base.json's relatives (the architecture is read from the file: nrs_snapshot_open)."""
    from conftest import Scene
    from nerfshop_amd import formats
    scene = Scene(aabb_scale=1, with_edit=False, shaped=True, **kw)
    formats.save_snapshot(tmp_path / "scene.msgpack", scene.desc, 1, scene.params, scene.grid, camera=scene.camera(60.0))
    out = tmp_path / "report.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_snapshot.py"), str(tmp_path / "scene.msgpack"), "--res", "160x90", "--max-views", "1", "--out", str(out)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.load(open(out))
    assert rep["parity_all_views"] and len(rep["views"]) == 1
    for mode in ("fp32_fp32 (default)", "network_fp16 (tiny-cuda-nn as recalled)"):
        assert rep["views"][0][mode]["parity"] and rep["views"][0][mode]["samples"] > 5000
