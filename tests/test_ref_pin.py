"""The oracle pinned to the REFERENCE'S OWN CODE (VERDICT r1 "next round" #1).

oracle/_ref/libref_render.so is the reference's render path compiled as host code: its headers on the path (common_device.cuh,
bounding_box.cuh, triangle.cuh, random_val.cuh, nerf.h, editing/tools/selection_utils.h, mvc.h, svd3.h) and src/common_nerf.cu whole, and
the kernels cut out of src/testbed_nerf.cu (:557-606, :637-696, :698-979, :2448-2616 ...), src/editing/cage_deformation.cu (:136-269, :341-541),
src/editing/datastructures/tet_mesh.cu (:49-70, :407-468, :585-641), src/editing/affine_duplication.cu (:69-118) by oracle/ref_extract.py --
against a stand-in for the EMPTY Eigen / tiny-cuda-nn submodules (oracle/ref_stubs; the stand-in's evaluation-order model is stated there).

Two layers:
  * `*_golden`: the oracle against tests/golden/ref_pin_golden.npz (made from that library by tests/golden/make_ref_pin_golden.py).
    Runs everywhere, also on the GPU box where /root/reference does not exist.
  * `*_live`: the oracle against the library itself, array by array with counts of differing elements (only where it is built).
Bar: bit-exact, 10^5 seeded inputs per function, whole frames (RGBA, depth, per-pixel sample counts) and whole sample streams for the
lego-like and the aabb-16 scene.  What stays unpinned is tiny-cuda-nn (hash grid, MLPs, SH encoding, pcg32): the frames use the oracle's
network on both sides.
"""
import os

import numpy as np
import pytest

import ref_pin_cases as cases
from oracle import ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_pin_golden.npz")
live = pytest.mark.skipif(not ref.available(), reason="oracle/_ref/libref_render.so needs /root/reference (build: make -C oracle)")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def scenes(built):
    return cases.Scenes()


def _check_hashed(golden, prefix, results):
    bad = []
    for name, arrays in results.items():
        if not np.array_equal(cases.sha(*arrays), golden[f"{prefix}/{name}/sha"]):
            heads = [int((np.ascontiguousarray(a)[:64].view(np.uint8) != golden[f"{prefix}/{name}/head{k}"].view(np.uint8)).sum()) for k, a in enumerate(arrays)]
            bad.append((name, heads))
    assert not bad, f"oracle differs from the reference's code (name, differing bytes in the first 64 rows of each output): {bad}"


def _check_live(a, b):
    bad = {}
    for name in a:
        for k, (x, y) in enumerate(zip(a[name], b[name])):
            x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
            assert x.shape == y.shape and x.dtype == y.dtype, (name, k, x.shape, y.shape)
            n = int((x.reshape(x.shape[0], -1).view(np.uint8) != y.reshape(y.shape[0], -1).view(np.uint8)).any(axis=1).sum())
            if n:
                bad[f"{name}[{k}]"] = n
    assert not bad, f"rows that differ between the reference's code and the oracle: {bad}"


# ---- header / common_nerf.cu functions, 10^5 inputs each ---------------------------------------------------------------------
def test_functions_golden(built, golden):
    _check_hashed(golden, "probe", cases.run_probes("orc"))


def test_pixel_to_ray_golden(scenes, golden):
    _check_hashed(golden, "probe", {"pixel_to_ray": cases.pixel_to_ray_case(scenes.get("lego"), "orc")})


@live
def test_functions_live(built):
    _check_live(cases.run_probes("ref"), cases.run_probes("orc"))


# ---- edit operators on caller batches: interpolate_tet(_pos), compute_residual_poisson_kernel, translate_in_box(_pos) ---------------
def test_operators_golden(scenes, golden):
    _check_hashed(golden, "op", cases.operator_cases(scenes, "orc"))


@live
def test_operators_live(scenes):
    a, b = cases.operator_cases(scenes, "ref"), cases.operator_cases(scenes, "orc")
    _check_live(a, b)
    # the cases must exercise the branches: samples mapped back, samples masked, membrane terms present
    for key in ("lego", "aabb16"):
        c = cases.edit_coords(scenes.get(key), cases.N, 11)
        moved = (a[f"{key}_map_rays_copy0"][0][:, :3] != c[:, :3]).any(axis=1)
        assert moved.sum() > 5000 and a[f"{key}_map_rays_copy0"][1].sum() > 500 and a[f"{key}_map_rays_copy1"][1].sum() == 0
        assert (a[f"{key}_poisson_residuals"][1] > 1e-9).sum() > 2000


# ---- build_tet_grid, update_local_rotations, compute_mvc / interpolate_with_mvc, grid_to_bitfield + bitfield_max_pool -------------
def test_authoring_golden(scenes, golden):
    _check_hashed(golden, "authoring", cases.authoring_cases(scenes, "orc"))
    _check_hashed(golden, "bitfield", {"grids": cases.bitfield_case(scenes, "orc")})


@live
def test_authoring_live(scenes):
    _check_live(cases.authoring_cases(scenes, "ref"), cases.authoring_cases(scenes, "orc"))
    _check_live({"grids": cases.bitfield_case(scenes, "ref")}, {"grids": cases.bitfield_case(scenes, "orc")})


# ---- deformed-space occupancy refresh: the reference's sample generator, map_positions, residual, splat and decayed-maximum kernels --------
def test_refresh_golden(scenes, golden):
    got = cases.refresh_cases(scenes, "orc")
    _check_hashed(golden, "refresh", got)
    g = got["lego_membrane"]
    assert (g[0][:128 ** 3] > 0.01).sum() > 20000 and not g[0][128 ** 3:].any()  # the shape was found in cascade 0; the other cascades stay untouched


@live
def test_refresh_live(scenes):
    _check_live(cases.refresh_cases(scenes, "ref"), cases.refresh_cases(scenes, "orc"))


# ---- the selection tool's ray shooting: shoot_selection_rays_kernel -> density() -> composite_shot_rays; get_upper_cell_idx -------------
def test_selection_golden(scenes, golden):
    got = cases.selection_cases(scenes, "orc")
    _check_hashed(golden, "selection", got)
    assert got["lego_shaped_thr0.1"][2].sum() > 1000 and got["aabb16_thr0.5"][2].sum() > 3000  # rays did find a surface


@live
def test_selection_live(scenes):
    _check_live(cases.selection_cases(scenes, "ref"), cases.selection_cases(scenes, "orc"))


# ---- get_density_on_grid / get_rgba_on_grid: generate_grid_samples_nerf_uniform(_dir), grid_samples_half_to_float, compute_nerf_density -----
def test_grid_eval_golden(scenes, golden):
    got = cases.grid_eval_cases(scenes, "orc")
    _check_hashed(golden, "grid_eval", got)
    assert (got["lego_shaped"][0] == -10000).sum() > 1000 and got["lego_shaped"][2][:, 3].max() > 0.3


@live
def test_grid_eval_live(scenes):
    _check_live(cases.grid_eval_cases(scenes, "ref"), cases.grid_eval_cases(scenes, "orc"))


# ---- compute_poisson_boundary: the reference's sampling loop, activate_network_output, filter_empty, density pick and project_sh9 fit loop ----
def test_poisson_boundary_golden(scenes, golden):
    got = cases.poisson_boundary_cases(scenes, "orc")
    _check_hashed(golden, "poisson_boundary", got)
    assert (got["outside"][0] > 0).any() and 0 < (got["inside"][0] == 0).sum() < got["inside"][0].size  # filter_empty took both branches


@live
def test_poisson_boundary_live(scenes):
    _check_live(cases.poisson_boundary_cases(scenes, "ref"), cases.poisson_boundary_cases(scenes, "orc"))


# ---- whole frames: init_rays -> advance_pos -> [compact -> generate inputs -> residuals -> map_rays -> network -> composite]* -> shade ----
@pytest.mark.parametrize("case", cases.FRAME_CASES, ids=[c[0] for c in cases.FRAME_CASES])
def test_frame_golden(scenes, golden, case):
    f, d, s, st = cases.render_case(scenes, case, "orc")
    g = lambda k: golden[f"frame/{case[0]}/{k}"]
    assert np.array_equal(st, g("stats")), (st, g("stats"))
    assert np.array_equal(s, g("steps").astype(np.uint32))
    assert np.array_equal(f.view(np.uint32), g("frame").view(np.uint32)), int((f.view(np.uint32) != g("frame").view(np.uint32)).sum())
    assert np.array_equal(d.view(np.uint32), g("depth").view(np.uint32))
    assert (st[0] > 500 or case[3].get("render_mode") == 7) and (f[..., 3] > 0).sum() > 500  # not an empty picture (render mode Distortion traces nothing: n_hit = 0)


@live
@pytest.mark.parametrize("case", [cases.FRAME_CASES[1], cases.FRAME_CASES[5], cases.FRAME_CASES[8]], ids=lambda c: c[0])
def test_frame_live_larger(scenes, case):
    """the same pipeline at 160x90 (not stored): frame, depth, per-pixel sample counts and the trace() statistics bit for bit"""
    big = (case[0], case[1], (160, 90, case[2][2] + 17.0), case[3], case[4])
    fr, dr, sr, str_ = cases.render_case(scenes, big, "ref")
    fo, do, so, sto = cases.render_case(scenes, big, "orc")
    assert np.array_equal(str_, sto)
    assert np.array_equal(sr, so)
    assert np.array_equal(fr.view(np.uint32), fo.view(np.uint32)) and np.array_equal(dr.view(np.uint32), do.view(np.uint32))


# ---- whole sample streams of every pixel of a view (lego and aabb 16; snapped, jittered, min_mip) ---------------------------------
@pytest.mark.parametrize("k", range(len(cases.STREAM_CASES)))
def test_stream_golden(scenes, golden, k):
    coords, t_after, cnt, odt = cases.stream_case(scenes, cases.STREAM_CASES[k], "orc")
    assert np.array_equal(cnt, golden[f"stream/{k}/count"].astype(np.uint32))
    assert np.array_equal(cases.sha(coords, t_after, cnt, odt), golden[f"stream/{k}/sha"])
    assert cnt.sum() > 100000 and cnt.max() > 60


@live
@pytest.mark.parametrize("k", [0, 3])
def test_stream_live(scenes, k):
    a, b = cases.stream_case(scenes, cases.STREAM_CASES[k], "ref"), cases.stream_case(scenes, cases.STREAM_CASES[k], "orc")
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


# ---- CudaRenderBuffer::accumulate: the reference's accumulate_kernel (render_buffer.cu:217-254), all three colour spaces ----
def test_accumulate_golden(golden):
    _check_hashed(golden, "accumulate", cases.accumulate_cases("orc"))


@live
def test_accumulate_live():
    _check_live(cases.accumulate_cases("ref"), cases.accumulate_cases("orc"))


# ---- on-disk edits format: nrs_edits_open against files written by the REFERENCE's own writer ---------------------------------
# oracle/_ref/libref_json.so = Testbed::save_edits + the to_json / from_json it reaches, compiled from /root/reference (oracle/ref_json.cpp).
# Golden: tests/golden/ref_edits_golden.json.gz (+ .npz: the arrays that went into the reference's writer), tests/golden/make_ref_edits_golden.py.
def _check_edits_against(ops, a):
    from nerfshop_amd import _abi
    assert len(ops) == 4
    c0, ad, c2, c3 = ops
    for c, k in ((c0, "op0"), (c2, "op2")):
        for got, name in ((c.vertices, "vertices"), (c.original_vertices, "original_vertices"), (c.tets, "tets"), (c.cage_deformed, "cage_vertices"),
                          (c.cage_vertices, "cage_original_vertices"), (c.cage_triangles, "cage_triangles")):
            want = a[f"{k}/{name}"]
            assert got.dtype == want.dtype and np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{k}/{name}"  # bit for bit (-0, denormals, FLT_MAX included)
    assert np.array_equal(c0.mvc_weights.view(np.uint32), a["op0/mvc"].view(np.uint32))
    assert c2.mvc_weights is None  # an empty mvc_coordinates vector: the reference writes null
    assert isinstance(ad, _abi.AffineDuplicationOp)
    for name in ("selection_center", "selection_scale", "selection_rot", "translation", "scale", "rotation"):
        assert np.array_equal(np.array(list(getattr(ad, name)), np.float32).view(np.uint32), a["op1/" + name].reshape(-1).view(np.uint32)), name
    assert int(ad.hide_original) == 1 and int(ad.correct_dir) == 0
    # an operator without interpolation mesh: the proxy cage is there, the mesh is empty
    assert c3.tets.shape == (0, 4) and c3.vertices.shape == (0, 3)
    assert np.array_equal(c3.cage_deformed, a["op3/cage_vertices"]) and np.array_equal(c3.cage_vertices, a["op3/cage_original_vertices"]) and np.array_equal(c3.cage_triangles, a["op3/cage_triangles"])


def test_edits_reader_golden(built, tmp_path):
    """nrs_edits_open on a file written by the reference's Testbed::save_edits returns the arrays that went into the writer, bit for bit."""
    import gzip
    from nerfshop_amd import formats
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    path = tmp_path / "ref_edits.json"
    path.write_bytes(gzip.open(os.path.join(gdir, "ref_edits_golden.json.gz"), "rb").read())
    text = path.read_text()
    # what makes this file the reference's and not the harness's: untouched values are null, keys are sorted, the selection bookkeeping is all there
    assert '"all_indices":null' in text and '"mvc_coordinates":null' in text and '"region_growing":{"density_grid_host":[]' in text and text.endswith("\n")
    _check_edits_against(formats.load_edits(path), np.load(os.path.join(gdir, "ref_edits_golden.npz")))


def test_edits_reader_golden_in_nlohmann_text(built, tmp_path):
    """The same value tree as the REAL nlohmann/json prints it (tests/golden/ref_edits_golden_nlohmann.json.gz: the file above read and written back by this image's
    nlohmann/json 3.1.1 the way Testbed::save_edits / load_edits do, oracle/ref_json_redump.cpp) -- shortest round-trip floats where the stand-in the reference's
    to_json code was compiled against prints 17 digits: nrs_edits_open returns the same arrays, bit for bit."""
    import gzip
    import json
    from nerfshop_amd import formats
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    raw = gzip.open(os.path.join(gdir, "ref_edits_golden_nlohmann.json.gz"), "rb").read()
    standin = gzip.open(os.path.join(gdir, "ref_edits_golden.json.gz"), "rb").read()
    assert raw != standin and len(raw) < len(standin) and json.loads(raw) == json.loads(standin)  # another text, the same doubles
    assert b"0.712745189666748," in raw and b"0.71274518966674805," in standin  # (the first float of the file, both ways)
    path = tmp_path / "ref_edits_nlohmann.json"
    path.write_bytes(raw)
    _check_edits_against(formats.load_edits(path), np.load(os.path.join(gdir, "ref_edits_golden.npz")))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "json_redump")),
                    reason="oracle/_ref/json_redump is needed (make -C oracle where /opt/conda/include/json.hpp exists)")
def test_nlohmann_golden_is_what_the_library_writes_live():
    """The committed nlohmann-text golden is what the real library writes NOW from the stand-in's golden, and a second pass through the library is a fixed point."""
    import gzip
    import importlib.util
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_ref_edits_golden", os.path.join(gdir, "make_ref_edits_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    raw = mk.redump_with_nlohmann(gzip.open(os.path.join(gdir, "ref_edits_golden.json.gz"), "rb").read())
    assert raw == gzip.open(os.path.join(gdir, "ref_edits_golden_nlohmann.json.gz"), "rb").read()
    assert mk.redump_with_nlohmann(raw) == raw


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "json_redump")),
                    reason="oracle/_ref/json_redump is needed (make -C oracle where /opt/conda/include/json.hpp exists)")
def test_harness_writer_is_parsed_by_nlohmann_live(built, tmp_path):
    """The other direction: the real library's parser accepts a file of the harness's own writer (nerfshop_amd/formats.py), and its own print of what it parsed is
    read by nrs_edits_open as the same arrays."""
    import importlib.util
    from nerfshop_amd import formats, synth
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_ref_edits_golden", os.path.join(gdir, "make_ref_edits_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    edit = synth.make_cage_edit(lattice_n=3)
    mine, back = tmp_path / "mine.json", tmp_path / "through_nlohmann.json"
    formats.save_edits(mine, [edit])
    back.write_bytes(mk.redump_with_nlohmann(mine.read_bytes()))
    a, b = formats.load_edits(mine), formats.load_edits(back)
    for name in ("vertices", "original_vertices", "tets", "mvc_weights", "cage_deformed", "cage_vertices", "cage_triangles"):
        x, y = getattr(a[0], name), getattr(b[0], name)
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)) and np.array_equal(x, getattr(edit, name)), name


# (the library travels with the repo snapshot to boxes where /root/reference does not exist; the live tests also read the reference's configs/nerf/ there)
ref_json_live = pytest.mark.skipif(not (os.path.exists(os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "_ref", "libref_json.so")) and
                                        os.path.isdir("/root/reference/configs/nerf")),
                                   reason="oracle/_ref/libref_json.so and /root/reference are needed (build: make -C oracle)")


@ref_json_live
def test_edits_reader_live(built, tmp_path):
    """The same with the file written NOW by the reference's code, and the committed golden file is what that code writes today (byte for byte)."""
    import gzip
    import importlib.util
    from nerfshop_amd import formats
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_ref_edits_golden", os.path.join(gdir, "make_ref_edits_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    a = mk.inputs()
    path = tmp_path / "live.json"
    mk.write_with_reference(str(path), a)
    _check_edits_against(formats.load_edits(path), a)
    assert path.read_bytes() == gzip.open(os.path.join(gdir, "ref_edits_golden.json.gz"), "rb").read()
    g = np.load(os.path.join(gdir, "ref_edits_golden.npz"))
    assert sorted(g.files) == sorted(a) and all(np.array_equal(g[k], a[k]) for k in a)


@ref_json_live
def test_harness_writer_is_read_by_the_reference_live(scene, tmp_path):
    """The other direction: a file of the harness's own writer (nerfshop_amd/formats.py, which the round-trip tests and the synthetic scenes use) goes through the
    REFERENCE's readers -- Testbed::load_edits' dispatch, from_json of Cage / TetMesh / Mesh / AffineBoundingBox (every `at(key)` they demand must be there) -- and what
    they loaded, written back by the reference's writers, is read by nrs_edits_open as the same arrays."""
    from nerfshop_amd import _abi, formats
    from oracle import ref_json
    ad = _abi.AffineDuplicationOp()
    ad.selection_center, ad.selection_scale = (C3 := type(ad.selection_center))(0.5, 0.5, 0.5), C3(0.1, 0.2, 0.3)
    ad.selection_rot = type(ad.selection_rot)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    ad.translation, ad.scale = C3(0.25, 0.0, -0.125), C3(1, 1, 1)
    ad.rotation = type(ad.rotation)(0, 1, 0, -1, 0, 0, 0, 0, 1)
    ad.hide_original, ad.correct_dir = 0, 1
    mine = tmp_path / "mine.json"
    formats.save_edits(mine, [scene.edit, ad])
    back = tmp_path / "through_the_reference.json"
    assert ref_json.reload_edits(mine, back) == 2
    a, b = formats.load_edits(mine), formats.load_edits(back)
    for name in ("vertices", "original_vertices", "tets", "mvc_weights", "cage_deformed", "cage_vertices", "cage_triangles"):
        x, y = getattr(a[0], name), getattr(b[0], name)
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)) and np.array_equal(x, getattr(scene.edit, name)), name
    assert bytes(a[1]) == bytes(b[1]) == bytes(ad)


# ---- snapshot container and keys: nrs_snapshot_open against a file written by the REFERENCE's Testbed::save_snapshot over its own configs/nerf/*.json ---------------
# (oracle/ref_json.cpp: save_snapshot, merge_parent_network_config and to_json(NerfDataset) are the reference's code; Trainer::serialize's three keys and the binary
# conversions are tiny-cuda-nn's, restated there -- those stay unpinned.)  Golden: tests/golden/ref_snapshot_golden.msgpack.gz + .npz, make_ref_snapshot_golden.py.
def _load_snapshot_golden_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_snapshot_golden", os.path.join(os.path.dirname(__file__), "golden", "make_ref_snapshot_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk


def _check_snapshot(s, params, idx, val, per_level_scale, log2_t, aabb_scale):
    d = s.desc
    assert (d.n_levels, d.n_features_per_level, d.log2_hashmap_size, d.base_resolution) == (16, 2, log2_t, 16)          # configs/nerf/base.json:23-29
    assert (d.n_neurons, d.density_hidden_layers, d.density_output_dims, d.rgb_hidden_layers, d.sh_degree) == (64, 1, 16, 2, 4)
    assert np.float32(d.per_level_scale) == np.float32(per_level_scale)                                                 # derived on load (testbed.cu:2280-2292)
    assert s.aabb_scale == aabb_scale                                                                                   # snapshot.nerf.dataset.aabb_scale (json_binding.h:154)
    assert np.array_equal(s.params, params)
    grid = np.zeros(5 * 128 ** 3, np.float32)
    grid[idx] = val
    assert np.array_equal(s.density_grid.view(np.uint32), grid.view(np.uint32))
    assert s.camera is None                                                                                             # save_snapshot stores no camera (export_snapshot does)


def test_snapshot_reader_golden(built, tmp_path):
    import gzip
    from nerfshop_amd import formats
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    path = tmp_path / "ref_snapshot.msgpack"
    path.write_bytes(gzip.open(os.path.join(gdir, "ref_snapshot_golden.msgpack.gz"), "rb").read())
    g = np.load(os.path.join(gdir, "ref_snapshot_golden.npz"))
    _check_snapshot(formats.load_snapshot(path), g["params"], g["grid_idx"], g["grid_val"], g["per_level_scale"], 12, 4)


@ref_json_live
def test_snapshot_reader_live(built, tmp_path):
    """The golden file is what the reference's code writes today, byte for byte; and EVERY config of the reference's configs/nerf/ -- parsed where it lies, through the
    reference's own parent chain -- is either read back as the network it describes or refused as outside the path (NRS_ERR_UNSUPPORTED), never misread."""
    import ctypes as C
    import gzip
    from nerfshop_amd import _abi, formats, synth
    from oracle import ref_json
    mk = _load_snapshot_golden_module()
    desc, params, idx, val = mk.inputs()
    path = tmp_path / "live.msgpack"
    ref_json.save_snapshot(path, mk.CONFIG, params, mk.grid_from(idx, val), mk.AABB_SCALE, log2_hashmap_size=mk.LOG2_T, training_step=12345, loss=0.0123)
    assert path.read_bytes() == gzip.open(os.path.join(os.path.dirname(__file__), "golden", "ref_snapshot_golden.msgpack.gz"), "rb").read()
    _check_snapshot(formats.load_snapshot(path), params, idx, val, desc.per_level_scale, mk.LOG2_T, mk.AABB_SCALE)

    cfg_dir = "/root/reference/configs/nerf"
    # (density hidden layers, rgb hidden layers, sh degree) of the members of base.json's family; None = an encoding / network the path does not render
    expect = {"base.json": (1, 2, 4), "hashgrid.json": (1, 2, 4), "small.json": (1, 2, 4), "base_14.json": (1, 2, 4), "big.json": (1, 2, 4), "base_2layer.json": (1, 2, 4),
              "base_0layer.json": (1, 0, 4), "base_1layer.json": (1, 1, 4), "base_3layer.json": (1, 3, 4), "linear.json": (0, 0, 4), "base_nodir.json": (1, 0, 0),
              "frequency.json": None, "densegrid.json": None, "densegrid_1res.json": None, "tensor.json": None, "none.json": None}
    assert sorted(os.listdir(cfg_dir)) == sorted(expect), "a config this test does not know"
    grid = np.zeros(5 * 128 ** 3, np.float32)
    lib = _abi.load()
    for name, arch in expect.items():
        out = tmp_path / (name + ".msgpack")
        if arch is None:
            ref_json.save_snapshot(out, os.path.join(cfg_dir, name), np.zeros(64, np.uint16), grid, 1)
            with pytest.raises(_abi.NrsError) as ei:
                formats.load_snapshot(out)
            assert "nrs error -2" in str(ei.value), (name, str(ei.value))
            continue
        d = synth.model_desc(1, log2_hashmap_size=12, density_hidden_layers=arch[0], rgb_hidden_layers=arch[1], no_dir=arch[2] == 0)
        n = lib.nrs_model_n_params(C.byref(d))
        p = (np.arange(n) % 0x7bff).astype(np.uint16)
        ref_json.save_snapshot(out, os.path.join(cfg_dir, name), p, grid, 1, log2_hashmap_size=12)
        s = formats.load_snapshot(out)
        assert (s.desc.density_hidden_layers, s.desc.rgb_hidden_layers, s.desc.sh_degree, s.desc.log2_hashmap_size) == (arch[0], arch[1], arch[2], 12), name
        assert np.array_equal(s.params, p), name
