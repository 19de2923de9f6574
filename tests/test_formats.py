"""the snapshot (.msgpack / .ingp) and edits (.json) readers of libnrs against files written in the
reference's schemas (Testbed::save_snapshot / export_snapshot / save_edits, src/testbed.cu:3089-3203) by the harness.
Host-only: no GPU needed.  Arrays must come back bit-identical (fp16 grids: identical to the fp16 rounding)."""
import ctypes as C
import json

import numpy as np
import pytest

from nerfshop_amd import _abi, formats, synth


@pytest.mark.parametrize("aabb_scale,ext", [(1, "msgpack"), (1, "ingp"), (16, "ingp"), (16, "msgpack")])
def test_snapshot_round_trip(built, tmp_path, aabb_scale, ext):
    desc = synth.model_desc(aabb_scale)
    rng = np.random.default_rng(3)
    n = _abi.load().nrs_model_n_params(C.byref(desc))
    params = rng.integers(0, 0x7BFF, size=n, dtype=np.uint16)       # arbitrary finite fp16 bit patterns
    grid = np.zeros(5 * 128 ** 3, np.float32)
    used = (int(np.log2(aabb_scale)) + 1) * 128 ** 3
    grid[:used] = rng.uniform(0, 0.2, size=used).astype(np.float32)
    grid[:1000] = -1.0                                               # untrained cells
    cam = synth.orbit_camera(30.0, 30.0, scale=0.33)
    path = tmp_path / f"scene.{ext}"
    formats.save_snapshot(path, desc, aabb_scale, params, grid, camera=cam)
    s = formats.load_snapshot(path)
    assert s.aabb_scale == aabb_scale
    for f, _ in _abi.ModelDesc._fields_:
        a, b = getattr(s.desc, f), getattr(desc, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), f
    assert np.array_equal(s.params, params)
    if ext == "ingp":   # export_snapshot stores fp16 of the used cascades
        assert np.array_equal(s.density_grid[:used], grid[:used].astype(np.float16).astype(np.float32))
        assert (s.density_grid[used:] == 0).all()
    else:
        assert np.array_equal(s.density_grid, grid)
    assert np.array_equal(s.camera, np.asarray(cam, np.float32).reshape(-1))


@pytest.mark.parametrize("kw", [dict(rgb_hidden_layers=0), dict(rgb_hidden_layers=1), dict(rgb_hidden_layers=3), dict(no_dir=True), dict(log2_hashmap_size=15),
                                dict(log2_hashmap_size=21), dict(rgb_hidden_layers=0, density_hidden_layers=0)],
                         ids=["base_0layer", "base_1layer", "base_3layer", "base_nodir", "small", "big", "linear"])
def test_snapshot_round_trip_of_the_family(built, tmp_path, kw):
    """configs/nerf/base.json's relatives: the rgb network's depth is read from the file, a file with neither dir_encoding nor rgb_network describes a
    NerfNetworkNoDir (testbed.cu:2314: sh_degree 0 in nrs_model_desc), and the parameter count follows."""
    desc = synth.model_desc(1, **kw)
    n = _abi.load().nrs_model_n_params(C.byref(desc))
    assert n > 0
    params = np.random.default_rng(4).integers(0, 0x7BFF, size=n, dtype=np.uint16)
    grid = np.zeros(5 * 128 ** 3, np.float32)
    path = tmp_path / "scene.msgpack"
    formats.save_snapshot(path, desc, 1, params, grid)
    s = formats.load_snapshot(path)
    for f, _ in _abi.ModelDesc._fields_:
        a, b = getattr(s.desc, f), getattr(desc, f)
        assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), f
    assert np.array_equal(s.params, params)
    # a file whose blob was written for another depth is refused by its size
    import msgpack
    cfg = msgpack.unpackb(path.read_bytes(), raw=False)
    if "rgb_network" in cfg:
        cfg["rgb_network"]["n_hidden_layers"] = (int(cfg["rgb_network"]["n_hidden_layers"]) + 1) % 4
        (tmp_path / "other.msgpack").write_bytes(msgpack.packb(cfg, use_bin_type=True))
        with pytest.raises(_abi.NrsError) as ei:
            formats.load_snapshot(tmp_path / "other.msgpack")
        assert "wrong size" in str(ei.value)


def test_snapshot_float_params_and_errors(built, tmp_path):
    import msgpack
    desc = synth.model_desc(1)
    n = _abi.load().nrs_model_n_params(C.byref(desc))
    rng = np.random.default_rng(5)
    pf = rng.normal(0, 0.3, size=n).astype(np.float32)
    cfg = formats.network_config(desc, explicit_per_level_scale=True)
    cfg["snapshot"] = {"density_grid_size": 128, "params_type": "float", "params_binary": pf.tobytes(), "n_params": int(n),
                       "density_grid_binary": np.zeros(5 * 128 ** 3, np.float32).tobytes(), "nerf": {"aabb_scale": 1}}
    p = tmp_path / "f.msgpack"
    p.write_bytes(msgpack.packb(cfg, use_bin_type=True))
    s = formats.load_snapshot(p)
    assert np.array_equal(s.params, pf.astype(np.float16).view(np.uint16))   # Trainer::deserialize converts float -> half

    # errors, with the reference's messages where it has them
    def expect(mutate, fragment):
        c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
        c["snapshot"] = dict(cfg["snapshot"])
        c["snapshot"]["nerf"] = dict(cfg["snapshot"]["nerf"])
        mutate(c)
        q = tmp_path / "bad.msgpack"
        q.write_bytes(msgpack.packb(c, use_bin_type=True))
        with pytest.raises(_abi.NrsError) as ei:
            formats.load_snapshot(q)
        assert fragment in str(ei.value), str(ei.value)

    expect(lambda c: c.pop("snapshot"), "does not contain a snapshot")
    expect(lambda c: c["snapshot"].__setitem__("density_grid_size", 64), "Incompatible grid size")
    expect(lambda c: c["snapshot"].__setitem__("params_binary", b"123"), "wrong size")
    expect(lambda c: c["snapshot"]["nerf"].__setitem__("aabb_scale", 3), "power of two")
    expect(lambda c: c["encoding"].__setitem__("n_levels", 8), "architecture")
    # what the path does not render is REFUSED with NRS_ERR_UNSUPPORTED (-2), never mis-rendered: other encodings (configs/nerf/{frequency,densegrid,...}.json),
    # other activations, light directions (dataset.has_light_dirs -> n_extra_dims = 3, testbed.cu:2318, nerf.h:73-93) -- by flag and by the parameter blob's size
    expect(lambda c: c["encoding"].__setitem__("otype", "Frequency"), "nrs error -2")
    expect(lambda c: c["encoding"].__setitem__("otype", "DenseGrid"), "only HashGrid")
    expect(lambda c: c["encoding"].__setitem__("type", "Tiled"), "nrs error -2")
    expect(lambda c: c["encoding"].__setitem__("interpolation", "Smoothstep"), "nrs error -2")
    expect(lambda c: c["network"].__setitem__("activation", "Sine"), "nrs error -2")
    expect(lambda c: c["rgb_network"].__setitem__("output_activation", "Sigmoid"), "nrs error -2")
    expect(lambda c: c["dir_encoding"].__setitem__("otype", "Frequency"), "nrs error -2")
    expect(lambda c: c["dir_encoding"].__setitem__("nested", [{"otype": "OneBlob", "n_dims_to_encode": 3}]), "SphericalHarmonics")
    expect(lambda c: c["snapshot"]["nerf"].__setitem__("dataset", {"aabb_scale": 1, "has_light_dirs": True}), "light directions")
    expect(lambda c: c["snapshot"]["nerf"].__setitem__("n_extra_dims", 3), "light directions")
    expect(lambda c: c["snapshot"].__setitem__("params_binary", np.zeros(n + 64 * 16, np.float32).tobytes()), "3 extra input dimensions")
    # (otype is compared case-insensitively, as tiny-cuda-nn's create_encoding does)
    c_ok = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    c_ok["encoding"]["otype"] = "hashgrid"
    ok = tmp_path / "lower.msgpack"
    ok.write_bytes(msgpack.packb(c_ok, use_bin_type=True))
    assert np.array_equal(formats.load_snapshot(ok).params, s.params)
    trunc = tmp_path / "trunc.msgpack"
    trunc.write_bytes(p.read_bytes()[:1000])
    with pytest.raises(_abi.NrsError):
        formats.load_snapshot(trunc)
    with pytest.raises(_abi.NrsError):
        formats.load_snapshot(tmp_path / "missing.msgpack")


def test_edits_round_trip(scene, tmp_path):
    e = scene.edit
    path = tmp_path / "edits.json"
    formats.save_edits(path, [e])
    ops = formats.load_edits(path)
    assert len(ops) == 1
    c = ops[0]
    assert np.array_equal(c.vertices, e.vertices) and np.array_equal(c.original_vertices, e.original_vertices)
    assert np.array_equal(c.tets, e.tets)
    assert np.array_equal(c.mvc_weights, e.mvc_weights)
    assert np.array_equal(c.cage_deformed, e.cage_deformed) and np.array_equal(c.cage_vertices, e.cage_vertices)
    assert np.array_equal(c.cage_triangles.reshape(-1), e.cage_triangles.reshape(-1))
    # operators the reference's load_edits accepts but this path does not execute are reported by type
    doc = json.loads(path.read_text())
    doc["edit_operators"].insert(0, {"type": "twist"})
    path.write_text(json.dumps(doc))
    ops = formats.load_edits(path)
    assert ops[0] == "twist" and np.array_equal(ops[1].tets, e.tets)
    doc["edit_operators"].append({"type": "bogus"})
    path.write_text(json.dumps(doc))
    with pytest.raises(_abi.NrsError) as ei:
        formats.load_edits(path)
    assert "Invalid edit operator!" in str(ei.value)
    path.write_text('{"edit_operators": [{"type": "cage_deformation", "proxy_cage": {"vertices": [[0, 0]]}}]}')
    with pytest.raises(_abi.NrsError):
        formats.load_edits(path)


def test_json_parser_corner_cases(built, tmp_path):
    """escapes, exponents, nesting, whitespace -- through the edits reader (the only JSON entry point)."""
    p = tmp_path / "e.json"
    p.write_text(' {\n "edit_\\u006fperators" : [ ] , "x": [1e-3, -2.5E+2, true, false, null, "a\\"b\\\\c\\n"] }\n')
    assert formats.load_edits(p) == []
    p.write_text('{"edit_operators": [}')
    with pytest.raises(_abi.NrsError):
        formats.load_edits(p)
    p.write_text("[" * 100 + "]" * 100)
    with pytest.raises(_abi.NrsError):
        formats.load_edits(p)


def test_affine_edit_round_trip(built, tmp_path):
    op = synth.make_affine_edit(hide_original=True)
    path = tmp_path / "affine.json"
    formats.save_edits(path, [op])
    got = formats.load_edits(path)
    assert len(got) == 1
    assert bytes(got[0]) == bytes(op)
    doc = json.loads(path.read_text())
    del doc["edit_operators"][0]["translation"]
    path.write_text(json.dumps(doc))
    with pytest.raises(_abi.NrsError):
        formats.load_edits(path)


def test_readers_survive_corrupt_input(built, tmp_path):
    """The readers parse files from disk: truncations and random byte flips must end in NrsError (or a clean load), never in a crash."""
    import msgpack
    import zlib
    desc = synth.model_desc(1)
    cfg = formats.network_config(desc)
    cfg["snapshot"] = {"density_grid_size": 128, "params_type": "__half", "params_binary": b"\x00" * 64, "n_params": 32,
                       "density_grid_binary": b"\x00" * 128, "nerf": {"aabb_scale": 1, "rgb": {"rays_per_batch": 4096}},
                       "camera": {"matrix": [[1.0, 0.0, 0.0, 0.5], [0.0, 1.0, 0.0, 0.5], [0.0, 0.0, 1.0, 0.5]]}}
    blob = msgpack.packb(cfg, use_bin_type=True)
    rng = np.random.default_rng(11)
    path = tmp_path / "fuzz.msgpack"
    n_err = 0
    variants = [blob[:k] for k in range(0, len(blob), 7)]
    for _ in range(300):
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        variants.append(bytes(b))
    variants.append(b"\xdd\xff\xff\xff\xff")                 # array32 announcing 4 G elements
    variants.append(b"\xc6\xff\xff\xff\xff" + b"x" * 10)      # bin32 longer than the file
    variants.append(b"\x91" * 200)                               # nesting bomb
    variants.append(zlib.compress(blob)[:-5])                     # truncated zlib stream
    variants.append(b"\x78\x9c" + b"\x00" * 50)                 # zlib header, garbage body
    # hyper-parameters that would divide by zero / shift out of range if used unchecked: must be refused, not crash the process
    for key, val in (("n_features_per_level", 0), ("log2_hashmap_size", 200), ("log2_hashmap_size", -3), ("base_resolution", 1e12), ("n_features", 1e300),
                     ("per_level_scale", float("nan"))):
        bad = json.loads(json.dumps(formats.network_config(desc)))
        bad["encoding"][key] = val
        if key == "n_features_per_level":
            bad["encoding"]["n_features"] = 32
        bad["snapshot"] = cfg["snapshot"]
        path.write_bytes(msgpack.packb(bad, use_bin_type=True))
        with pytest.raises(_abi.NrsError):
            formats.load_snapshot(path)
    for v in variants:
        path.write_bytes(v)
        try:
            formats.load_snapshot(path)
        except _abi.NrsError:
            n_err += 1
    assert n_err >= len(variants) - 5      # (the params blob is deliberately the wrong size: essentially everything must be refused)
    jpath = tmp_path / "fuzz.json"
    good = json.dumps({"edit_operators": [{"type": "affine_duplication", "selection_box": {"center": [0, 0, 0], "scale": [1, 1, 1],
                       "rot_matrix": [[1, 0, 0], [0, 1, 0], [0, 0, 1]]}, "translation": [0, 0, 0], "scale": [1, 1, 1],
                       "rotation_matrix": [[1, 0, 0], [0, 1, 0], [0, 0, 1]], "hide_original": False, "correct_dir": True}]})
    jpath.write_text(good)
    assert len(formats.load_edits(jpath)) == 1
    for k in range(0, len(good), 3):
        jpath.write_text(good[:k])
        with pytest.raises(_abi.NrsError):
            formats.load_edits(jpath)
    for _ in range(200):
        b = bytearray(good.encode())
        b[int(rng.integers(0, len(b)))] = int(rng.integers(32, 127))
        jpath.write_bytes(bytes(b))
        try:
            formats.load_edits(jpath)
        except _abi.NrsError:
            pass
