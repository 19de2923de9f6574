"""Deterministic synthetic inputs for the render path (SURVEY.md 8d "Synthetic inputs").

No lego / garden snapshot exists offline, so tests and bench.py render a procedural stand-in that exercises
exactly the same code: a base.json-shaped network with seeded fp16 parameters, a lego-like solid baked into the
occupancy grid (shape comes from the bitfield, as in the reference where unoccupied cells are never sampled), the
nerf_synthetic orbit cameras, and one cage edit (box cage -> MVC -> Kuhn tet lattice -> translate + twist).

This module is numpy + the HOST-side authoring entry points of libnrs.so (LUT builder, MVC, rotations: product
code, CPU like the reference).  It never touches oracle/.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._abi import BITFIELD_BYTES, GRID_CASCADES, GRID_SIZE, GRID_VOLUME, N_LUT_CELLS, ModelDesc, RenderParams, TetMesh

SEED = 1337  # Testbed::m_seed, testbed.h:507
N_DENSITY_W = 64 * 32 + 16 * 64
N_RGB_W = 64 * 32 + 64 * 64 + 16 * 64
SQRT3 = np.float32(1.73205080757)
MIN_STEP = SQRT3 / np.float32(1024)


# ----------------------------------------------------------------------------------------------------------------
# Morton helpers (tcnn morton3D, x lowest bit)
# ----------------------------------------------------------------------------------------------------------------
def _expand_bits(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(x, y, z):
    return _expand_bits(x) | (_expand_bits(y) << np.uint32(1)) | (_expand_bits(z) << np.uint32(2))


def morton3d_invert(v):
    x = v.astype(np.uint32) & np.uint32(0x49249249)
    x = (x | (x >> np.uint32(2))) & np.uint32(0xC30C30C3)
    x = (x | (x >> np.uint32(4))) & np.uint32(0x0F00F00F)
    x = (x | (x >> np.uint32(8))) & np.uint32(0xFF0000FF)
    x = (x | (x >> np.uint32(16))) & np.uint32(0x0000FFFF)
    return x


_MORTON_CACHE = {}


def _cell_coords():
    """x, y, z (each [128^3] uint32) of every Morton index 0..128^3-1."""
    if "xyz" not in _MORTON_CACHE:
        idx = np.arange(GRID_VOLUME, dtype=np.uint32)
        _MORTON_CACHE["xyz"] = (morton3d_invert(idx), morton3d_invert(idx >> np.uint32(1)), morton3d_invert(idx >> np.uint32(2)))
    return _MORTON_CACHE["xyz"]


def cell_centres(level):
    """World-space centres of the 128^3 cells of cascade `level`, Morton order (get_cell_pos, selection_utils.cu:65)."""
    x, y, z = _cell_coords()
    s = np.float32(2.0 ** level)
    f = lambda c: ((c.astype(np.float32) + np.float32(0.5)) / np.float32(GRID_SIZE) - np.float32(0.5)) * s + np.float32(0.5)
    return np.stack([f(x), f(y), f(z)], axis=1)


# ----------------------------------------------------------------------------------------------------------------
# Model
# ----------------------------------------------------------------------------------------------------------------
def per_level_scale(aabb_scale, base_resolution=16, n_levels=16):
    """testbed.cu:2288-2292: std::exp(std::log(2048.f * aabb_scale / base_resolution) / (n_levels - 1)), every step in float -- through the C library's logf / expf, as
    the reference (and nrs_snapshot_open) evaluate it: numpy's float32 log / exp are its own SIMD kernels and differ from libm by an ulp for some arguments (aabb_scale 4:
    found by the snapshot pin of round 5, tests/test_ref_pin.py::test_snapshot_reader_golden)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    libm.logf.restype = libm.expf.restype = ctypes.c_float
    libm.logf.argtypes = libm.expf.argtypes = [ctypes.c_float]
    v = np.float32(2048.0) * np.float32(aabb_scale) / np.float32(base_resolution)
    x = np.float32(libm.logf(ctypes.c_float(float(v)))) / np.float32(n_levels - 1)
    return float(np.float32(libm.expf(ctypes.c_float(float(x)))))


def scene_aabb(aabb_scale):
    """Testbed::m_aabb = BoundingBox(Vector3f::Constant(0.5), aabb_scale) (testbed_nerf.cu:3410-3425)."""
    h = 0.5 * aabb_scale
    return (0.5 - h,) * 3, (0.5 + h,) * 3


def model_desc(aabb_scale=1, rgb_hidden_layers=2, no_dir=False, log2_hashmap_size=19, density_hidden_layers=1):
    """configs/nerf/base.json by default; its relatives: rgb_hidden_layers 0 / 1 / 3 (base_{0,1,3}layer.json), no_dir (base_nodir.json -> NerfNetworkNoDir:
    sh_degree 0, no rgb network), log2_hashmap_size 14 / 15 / 21 (base_14 / small / big.json), density_hidden_layers 0 (linear.json, with rgb_hidden_layers 0)."""
    d = ModelDesc()
    d.n_levels, d.n_features_per_level, d.log2_hashmap_size, d.base_resolution = 16, 2, log2_hashmap_size, 16
    d.per_level_scale = per_level_scale(aabb_scale)
    d.n_neurons, d.density_hidden_layers, d.density_output_dims, d.rgb_hidden_layers, d.sh_degree = 64, density_hidden_layers, 16, (0 if no_dir else rgb_hidden_layers), (0 if no_dir else 4)
    d.rgb_activation, d.density_activation = _abi.ACT_LOGISTIC, _abi.ACT_EXPONENTIAL
    mn, mx = scene_aabb(aabb_scale)
    d.aabb_min[:] = mn
    d.aabb_max[:] = mx
    return d


def level_table(desc):
    lib = _abi.load()
    scale = np.zeros(16, np.float32)
    res, off, cnt, hashed = (np.zeros(16, np.uint32) for _ in range(4))
    _abi.check(lib.nrs_model_level_table(C.byref(desc), scale.ctypes.data, res.ctypes.data, off.ctypes.data, cnt.ctypes.data, hashed.ctypes.data))
    return {"scale": scale, "resolution": res, "offset": off, "count": cnt, "hashed": hashed}


def default_sigma_raw(aabb_scale=1):
    """Raw density that gives alpha ~ 0.14 per sample at the scene's typical step: dt_min for aabb_scale 1; for the
    x6 "garden-style" scene the orbit camera sits ~8 units away, where cone stepping makes dt ~ t/256 ~ 0.031."""
    dt = 1.73205080757 / 1024.0 if aabb_scale == 1 else 0.031
    return math.log(0.15 / dt)


def make_params(desc, seed=SEED, sigma_raw=None, density_noise=0.25, shaped=False, aabb_scale=1, shape_gain=12.0):
    """fp16 parameter blob (uint16 bits) in tiny-cuda-nn order: density MLP | rgb MLP | hash grid.

    Draw order from numpy.random.Generator(PCG64(seed)): density W1 [64x32], density W2 [16x64], rgb W1 [64x32],
    rgb W2 [64x64], rgb W3 [16x64] (Xavier-uniform each), then the hash table ~ U(-0.5, 0.5).
    Opacity is controlled without biases (tcnn MLPs have none) through one constant-feature channel: feature 0 of
    every level-0 entry is 1.0, hidden unit 0 of the density MLP sees only that input (W1[0,0] = 1), and
    W2[0,0] = sigma_raw, so density_raw = sigma_raw + density_noise * (Xavier mix of the other hidden units).
    Default sigma_raw makes exp(sigma_raw) * dt_min = 0.15, i.e. alpha ~ 0.14 per sample.

    shaped=True additionally puts the scene's geometry INTO the network, as a trained snapshot has it (needed by the
    occupancy refresh, which derives the grid from density()): feature 0 of the finest dense level holds the solid's
    indicator at the level's vertices, hidden unit 1 reads only that feature, and W2[0,1] = shape_gain while the constant
    channel drops to sigma_raw - shape_gain: density_raw ~ sigma_raw inside the solid, sigma_raw - shape_gain outside.
    """
    lib = _abi.load()
    n = lib.nrs_model_n_params(C.byref(desc))
    if n == 0:
        raise ValueError("unsupported model description")
    if sigma_raw is None:
        sigma_raw = default_sigma_raw(1)
    rng = np.random.Generator(np.random.PCG64(seed))

    def xavier(n_out, n_in):
        lim = math.sqrt(6.0 / (n_in + n_out))
        return rng.uniform(-lim, lim, size=(n_out, n_in)).astype(np.float32)

    linear_density = desc.density_hidden_layers == 0  # configs/nerf/linear.json: the density network is one [16 x 32] matrix
    dw1, dw2 = (xavier(16, 32), np.zeros((0, 0), np.float32)) if linear_density else (xavier(64, 32), xavier(16, 64))
    # rgb network of the description (tiny-cuda-nn's layouts as recalled): none (NerfNetworkNoDir), one [8 x 32] matrix (CutlassMLP without hidden layer),
    # or [64 x 32] + (L - 1) [64 x 64] + [16 x 64]; base.json (L = 2) draws rgb W1, W2, W3 as it always did
    if desc.sh_degree == 0:
        rgb = []
    elif desc.rgb_hidden_layers == 0:
        rgb = [xavier(8, 32)]
    else:
        rgb = [xavier(64, 32)] + [xavier(64, 64) for _ in range(desc.rgb_hidden_layers - 1)] + [xavier(16, 64)]
    n_rgb = sum(w.size for w in rgb)
    grid = rng.uniform(-0.5, 0.5, size=n - dw1.size - dw2.size - n_rgb).astype(np.float32)
    lt = level_table(desc)
    # constant feature: level 0, feature 0
    o0, c0 = int(lt["offset"][0]), int(lt["count"][0])
    grid[2 * o0: 2 * (o0 + c0): 2] = 1.0
    if linear_density:       # density_raw = sigma_raw * const + density_noise * (mix of the other features)
        dw1[0, :] *= density_noise
        dw1[0, 0] = sigma_raw
    else:
        dw1[0, :] = 0.0
        dw1[0, 0] = 1.0          # hidden unit 0 = relu(1 * const) = 1
        dw2[0, :] *= density_noise
        dw2[0, 0] = sigma_raw
    if rgb:
        rgb[0][:, 0] = 0.0   # keep the large density channel out of the colour network
    if shaped:
        ls = int(np.nonzero(lt["hashed"] == 0)[0].max())          # finest dense level
        res, sc, off = int(lt["resolution"][ls]), np.float32(lt["scale"][ls]), int(lt["offset"][ls])
        ax = (np.arange(res, dtype=np.float32) - np.float32(0.5)) / sc   # warped coordinate of vertex i (pos * scale + 0.5 = i)
        zz, yy, xx = np.meshgrid(ax, ax, ax, indexing="ij")            # entry = x + y * res + z * res^2
        mn, mx = np.asarray(desc.aabb_min, np.float32), np.asarray(desc.aabb_max, np.float32)
        pts = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], axis=1) * (mx - mn) + mn
        ind = scene_indicator(pts, aabb_scale)
        grid[2 * off: 2 * (off + res ** 3): 2] = ind
        if linear_density:
            dw1[0, 2 * ls] = shape_gain
            dw1[0, 0] = sigma_raw - shape_gain
        else:
            dw1[1, :] = 0.0
            dw1[1, 2 * ls] = 1.0
            dw2[0, 1] = shape_gain
            dw2[0, 0] = sigma_raw - shape_gain
    blob = np.concatenate([dw1.ravel(), dw2.ravel()] + [w.ravel() for w in rgb] + [grid]).astype(np.float16)
    assert blob.size == n
    return blob.view(np.uint16)


# ----------------------------------------------------------------------------------------------------------------
# Occupancy: a lego-like solid (union of bricks + studs); NGP space, y is up (nerf_matrix_to_ngp cycles the axes)
# ----------------------------------------------------------------------------------------------------------------
_BRICKS = [  # (min xyz, max xyz) in the unit cube of the aabb_scale-1 scene
    ((0.20, 0.22, 0.26), (0.80, 0.27, 0.74)),  # base plate
    ((0.28, 0.27, 0.32), (0.72, 0.42, 0.68)),  # chassis
    ((0.36, 0.42, 0.38), (0.62, 0.58, 0.62)),  # cabin
    ((0.62, 0.42, 0.44), (0.76, 0.50, 0.56)),  # hood
    ((0.24, 0.27, 0.44), (0.30, 0.70, 0.56)),  # mast
    ((0.24, 0.64, 0.44), (0.56, 0.70, 0.56)),  # arm
    ((0.50, 0.50, 0.47), (0.56, 0.64, 0.53)),  # bucket link
]
_STUDS = [  # (centre x, z, y0, y1, radius): vertical cylinders
    (0.42, 0.44, 0.58, 0.61, 0.025), (0.42, 0.56, 0.58, 0.61, 0.025), (0.56, 0.44, 0.58, 0.61, 0.025), (0.56, 0.56, 0.58, 0.61, 0.025),
    (0.68, 0.50, 0.50, 0.53, 0.025), (0.34, 0.50, 0.70, 0.73, 0.025), (0.46, 0.50, 0.70, 0.73, 0.025),
]


def solid_indicator(points, solid_scale=1.0):
    """1.0 where a world-space point is inside the solid.  solid_scale > 1 scales the solid about the centre (0.5)."""
    p = (np.asarray(points, np.float32) - np.float32(0.5)) / np.float32(solid_scale) + np.float32(0.5)
    inside = np.zeros(p.shape[0], bool)
    for mn, mx in _BRICKS:
        inside |= np.all((p >= np.float32(mn)) & (p <= np.float32(mx)), axis=1)
    for cx, cz, y0, y1, r in _STUDS:
        inside |= ((p[:, 0] - cx) ** 2 + (p[:, 2] - cz) ** 2 <= r * r) & (p[:, 1] >= y0) & (p[:, 1] <= y1)
    return inside.astype(np.float32)


def scene_indicator(points, aabb_scale=1):
    """The synthetic scene's solid: the lego-like model for aabb_scale 1; for larger boxes ("garden-style") the model
    scaled x6 about the centre plus a ground slab."""
    if aabb_scale == 1:
        return solid_indicator(points, 1.0)
    scale = 6.0
    c = np.asarray(points, np.float32)
    v = solid_indicator(c, scale)
    y0 = 0.5 + (0.22 - 0.5) * scale
    slab = (c[:, 1] <= y0) & (c[:, 1] >= y0 - 0.35) & (np.abs(c[:, 0] - 0.5) <= 6.0) & (np.abs(c[:, 2] - 0.5) <= 6.0)
    return np.maximum(v, slab.astype(np.float32))


def n_scene_cascades(aabb_scale):
    """Cascades the reference trains for a scene box: max_cascade + 1 (testbed_nerf.cu:3410-3425)."""
    return 1 if aabb_scale == 1 else min(GRID_CASCADES, int(math.ceil(math.log2(aabb_scale))) + 1)


def density_grid(aabb_scale=1):
    """Float density grid [5 * 128^3], Morton order per cascade: 1.0 inside the solid, 0 outside, sampled at cell
    centres of every cascade the scene box covers (coarser mips come from bitfield_max_pool)."""
    grid = np.zeros(GRID_CASCADES * GRID_VOLUME, np.float32)
    for level in range(n_scene_cascades(aabb_scale)):
        grid[level * GRID_VOLUME:(level + 1) * GRID_VOLUME] = scene_indicator(cell_centres(level), aabb_scale)
    return grid


def grid_to_bitfield(grid):
    """numpy restatement of update_density_grid_mean_and_bitfield (testbed_nerf.cu:3642-3657) for CPU-only callers.
    (libnrs's nrs_model_set_density_grid is the device version; tests compare the two.)"""
    grid = np.asarray(grid, np.float32)
    mean = float(np.sum(np.maximum(grid[:GRID_VOLUME], 0.0).astype(np.float64) / GRID_VOLUME))
    thresh = np.float32(min(0.01, mean))
    bits = (grid > thresh).reshape(-1, 8)
    bitfield = np.zeros(BITFIELD_BYTES, np.uint8)
    for j in range(8):
        bitfield |= (bits[:, j].astype(np.uint8) << np.uint8(j))
    i = np.arange(GRID_VOLUME // 64, dtype=np.uint32)
    x = morton3d_invert(i) + np.uint32(GRID_SIZE // 8)
    y = morton3d_invert(i >> np.uint32(1)) + np.uint32(GRID_SIZE // 8)
    z = morton3d_invert(i >> np.uint32(2)) + np.uint32(GRID_SIZE // 8)
    dst = morton3d(x, y, z)
    lvl_bytes = GRID_VOLUME // 8
    for level in range(1, GRID_CASCADES):
        prev = bitfield[(level - 1) * lvl_bytes: level * lvl_bytes].reshape(-1, 8)
        pooled = np.zeros(GRID_VOLUME // 64, np.uint8)
        for j in range(8):
            pooled |= ((prev[:, j] > 0).astype(np.uint8) << np.uint8(j))
        nxt = bitfield[level * lvl_bytes:(level + 1) * lvl_bytes]
        nxt[dst] |= pooled
    return bitfield


# ----------------------------------------------------------------------------------------------------------------
# Cameras: nerf_synthetic orbit -> NGP space (nerf_loader.h:74-92, nerf_loader.cu:60)
# ----------------------------------------------------------------------------------------------------------------
CAMERA_ANGLE_X = 0.6911112070083618
ORBIT_RADIUS = 4.0311


def nerf_matrix_to_ngp(c2w, scale=0.33, offset=(0.5, 0.5, 0.5)):
    m = np.array(c2w, np.float32)[:3, :4].copy()
    m[:, 1] *= -1
    m[:, 2] *= -1
    m[:, 3] = m[:, 3] * np.float32(scale) + np.float32(offset)
    return m[[1, 2, 0], :]  # cycle axes xyz <- yzx


def orbit_camera(azimuth_deg, elevation_deg=30.0, radius=ORBIT_RADIUS, scale=0.33):
    """Blender-convention look-at camera on a sphere (z up), converted to the NGP 3x4 matrix (column-major flat)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    pos = np.array([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    back = pos / np.linalg.norm(pos)  # camera +Z points away from the target
    right = np.cross([0.0, 0.0, 1.0], back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    c2w = np.stack([right, up, back, pos], axis=1)
    ngp = nerf_matrix_to_ngp(c2w, scale)
    return np.ascontiguousarray(ngp.T.reshape(-1), np.float32)  # columns: 3 axes + origin


def render_params(width, height, camera, aabb_scale=1, spp_index=0, snap=True, apply_operators=True, camera_angle_x=CAMERA_ANGLE_X):
    p = RenderParams()
    p.resolution[:] = (width, height)
    focal = 0.5 * width / math.tan(0.5 * camera_angle_x)
    p.focal_length[:] = (focal, focal)
    p.camera_matrix0[:] = list(camera)
    p.camera_matrix1[:] = list(camera)
    p.rolling_shutter[:] = (0.0, 0.0, 0.0, 0.0)
    p.screen_center[:] = (0.5, 0.5)
    mn, mx = scene_aabb(aabb_scale)
    p.render_aabb_min[:] = mn
    p.render_aabb_max[:] = mx
    p.spp_index = spp_index
    p.snap_to_pixel_centers = 1 if snap else 0
    p.min_transmittance = 0.01
    p.cone_angle_constant = 0.0 if aabb_scale <= 1 else 1.0 / 256.0  # testbed_nerf.cu:3410-3425
    p.render_mode = _abi.RENDER_SHADE
    p.linear_colors = 0
    p.apply_operators = 1 if apply_operators else 0
    p.poisson_target = 0
    p.min_mip = 0
    p.max_march_steps = 0
    p.tile_size = p.tile_first = p.tile_stride = 0
    p.dof = 0.0            # Testbed::m_dof (testbed.h)
    p.slice_plane_z = 1.0  # m_slice_plane_z (0) + m_scale (1): testbed_nerf.cu:3067
    p.depth_scale = 1.0    # 1 / dataset.scale
    p.show_accel = 0       # m_nerf.show_accel = -1
    return p


# ----------------------------------------------------------------------------------------------------------------
# Cage edit
# ----------------------------------------------------------------------------------------------------------------
def box_cage(mn, mx, n=2):
    """Closed triangulated box surface with an n x n grid per face; outward-facing CCW triangles."""
    mn, mx = np.asarray(mn, np.float64), np.asarray(mx, np.float64)
    verts, index = [], {}

    def vid(i, j, k):
        key = (i, j, k)
        if key not in index:
            index[key] = len(verts)
            verts.append(mn + (mx - mn) * np.array(key, np.float64) / n)
        return index[key]

    tris = []
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (0, n):
            for a in range(n):
                for b in range(n):
                    def corner(da, db):
                        c = [0, 0, 0]
                        c[axis], c[u], c[v] = side, a + da, b + db
                        return vid(*c)
                    q = [corner(0, 0), corner(1, 0), corner(1, 1), corner(0, 1)]
                    if side == 0:
                        q = q[::-1]  # flip so the normal points outwards (-axis)
                    tris += [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
    return np.array(verts, np.float32), np.array(tris, np.uint32)


def kuhn_lattice(mn, mx, n):
    """(n+1)^3 lattice vertices and 6 n^3 Kuhn tetrahedra filling the box (conforming across cubes)."""
    mn, mx = np.asarray(mn, np.float64), np.asarray(mx, np.float64)
    g = np.arange(n + 1)
    I, J, K = np.meshgrid(g, g, g, indexing="ij")
    verts = (mn + (mx - mn) * np.stack([I, J, K], -1).reshape(-1, 3) / n).astype(np.float32)
    vid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    tets = []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                for perm in perms:
                    c = [i, j, k]
                    t = [vid(*c)]
                    for ax in perm:
                        c[ax] += 1
                        t.append(vid(*c))
                    tets.append(t)
    return verts, np.array(tets, np.uint32)


class CageEdit:
    """Host arrays of one cage-deformation operator (what GrowingSelection::update_tet_mesh leaves on the TetMesh)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def tet_mesh_struct(self, device_authoring=False):
        """device_authoring=True hands over only vertices + tets: libnrs builds the LUT, the canonical bitfield and the
        local rotations on the device (nrs.h, "next" row f1)."""
        m = TetMesh()
        m.n_vertices, m.n_tets = self.vertices.shape[0], self.tets.shape[0]
        m.h_vertices = self.vertices.ctypes.data
        m.h_original_vertices = self.original_vertices.ctypes.data
        m.h_tets = self.tets.ctypes.data
        if device_authoring:
            m.h_lut_offsets = m.h_lut_idx = m.h_original_bitfield = m.h_local_rotations = None
            m.correct_direction = 1 if self.local_rotations is not None else 0
        else:
            m.h_lut_offsets = self.lut_offsets.ctypes.data
            m.h_lut_idx = self.lut_idx.ctypes.data
            m.h_original_bitfield = self.original_bitfield.ctypes.data
            m.h_local_rotations = self.local_rotations.ctypes.data if self.local_rotations is not None else None
        m.copy = 1 if self.copy else 0
        shs = getattr(self, "boundary_shs", None)
        if shs is not None:  # membrane ("Poisson") correction, cage_deformation.h:163
            m.apply_poisson = 1
            m.residual_amplitude = float(getattr(self, "residual_amplitude", 1.0))
            m.h_boundary_shs = self.boundary_shs.ctypes.data
            m.h_boundary_outside_density = self.boundary_outside_density.ctypes.data
            m.h_boundary_residual_density = self.boundary_residual_density.ctypes.data
        else:
            m.apply_poisson = 0
            m.residual_amplitude = 1.0
        return m

    def with_membrane(self, seed=7, residual_amplitude=1.0):
        """A copy of this edit with synthetic per-vertex membrane terms (what GrowingSelection::interpolate_poisson_boundary
        leaves on the tet mesh, growing_selection.cu:2350): SH9RGB boundary colour, outside density, residual density.
        A third of the vertices get zero outside density so that both branches of composite_kernel_nerf are exercised."""
        import copy as _copy
        e = _copy.copy(self)
        rng = np.random.default_rng(seed)
        V = self.vertices.shape[0]
        shs = rng.normal(0.0, 0.15, size=(V, 27)).astype(np.float32)
        shs[:, 0::9] += np.float32(0.5 / 0.2820947917738781)  # DC term: mid-grey
        dens = rng.uniform(5.0, 60.0, size=V).astype(np.float32)
        dens[self.original_vertices[:, 0] < np.quantile(self.original_vertices[:, 0], 0.33)] = 0.0
        e.boundary_shs = np.ascontiguousarray(shs)
        e.boundary_outside_density = np.ascontiguousarray(dens)
        e.boundary_residual_density = np.ascontiguousarray(rng.uniform(-10.0, 40.0, size=V).astype(np.float32))
        e.residual_amplitude = residual_amplitude
        return e


def build_tet_lut(vertices, tets, n_threads=0):
    """libnrs host LUT builder -> (offsets [5*128^3+1], idx, touched-cell bitfield, max tets per cell)."""
    lib = _abi.load()
    vertices = np.ascontiguousarray(vertices, np.float32)
    tets = np.ascontiguousarray(tets, np.uint32)
    h = C.c_void_p()
    _abi.check(lib.nrs_tet_lut_build(vertices.ctypes.data, vertices.shape[0], tets.ctypes.data, tets.shape[0], n_threads, C.byref(h)))
    try:
        n_idx = lib.nrs_tet_lut_n_idx(h)
        offsets = np.ctypeslib.as_array(C.cast(lib.nrs_tet_lut_offsets(h), C.POINTER(C.c_uint32)), (N_LUT_CELLS + 1,)).copy()
        idx = np.ctypeslib.as_array(C.cast(lib.nrs_tet_lut_idx(h), C.POINTER(C.c_uint32)), (max(n_idx, 1),)).copy()[:n_idx]
        bitfield = np.ctypeslib.as_array(C.cast(lib.nrs_tet_lut_bitfield(h), C.POINTER(C.c_uint8)), (BITFIELD_BYTES,)).copy()
        max_per_cell = lib.nrs_tet_lut_max_per_cell(h)
    finally:
        lib.nrs_tet_lut_destroy(h)
    return offsets, idx, bitfield, max_per_cell


def mvc_weights(cage_vertices, cage_triangles, points):
    lib = _abi.load()
    cv = np.ascontiguousarray(cage_vertices, np.float32)
    tr = np.ascontiguousarray(cage_triangles, np.uint32)
    pts = np.ascontiguousarray(points, np.float32)
    w = np.zeros((pts.shape[0], cv.shape[0]), np.float32)
    labels = np.zeros(pts.shape[0], np.uint8)
    _abi.check(lib.nrs_mvc_compute(cv.ctypes.data, cv.shape[0], tr.ctypes.data, tr.shape[0], pts.ctypes.data, pts.shape[0], w.ctypes.data, labels.ctypes.data))
    return w, labels


def mvc_apply(weights, cage_vertices):
    lib = _abi.load()
    w = np.ascontiguousarray(weights, np.float32)
    cv = np.ascontiguousarray(cage_vertices, np.float32)
    out = np.zeros((w.shape[0], 3), np.float32)
    _abi.check(lib.nrs_mvc_apply(w.ctypes.data, cv.ctypes.data, cv.shape[0], w.shape[0], out.ctypes.data))
    return out


def local_rotations(vertices, original_vertices, tets):
    lib = _abi.load()
    v = np.ascontiguousarray(vertices, np.float32)
    o = np.ascontiguousarray(original_vertices, np.float32)
    t = np.ascontiguousarray(tets, np.uint32)
    out = np.zeros((t.shape[0], 9), np.float32)
    _abi.check(lib.nrs_tet_local_rotations(v.ctypes.data, o.ctypes.data, t.ctypes.data, t.shape[0], out.ctypes.data))
    return out


def deform_cage(cage_vertices, translate=(0.10, 0.05, 0.0), twist_deg=20.0):
    """Translate the top half of the cage and twist it about the vertical (y) axis through the cage centre."""
    cv = np.array(cage_vertices, np.float64)
    lo, hi = cv.min(0), cv.max(0)
    ctr = 0.5 * (lo + hi)
    top = cv[:, 1] > ctr[1] + 1e-9
    frac = np.clip((cv[:, 1] - ctr[1]) / (hi[1] - ctr[1]), 0.0, 1.0)
    ang = np.radians(twist_deg) * frac
    dx, dz = cv[:, 0] - ctr[0], cv[:, 2] - ctr[2]
    out = cv.copy()
    out[:, 0] = ctr[0] + np.cos(ang) * dx - np.sin(ang) * dz
    out[:, 2] = ctr[2] + np.sin(ang) * dx + np.cos(ang) * dz
    out[top] += np.array(translate, np.float64) * frac[top, None]
    return out.astype(np.float32)


def make_cage_edit(lattice_n=10, box=((0.22, 0.60, 0.40), (0.60, 0.76, 0.60)), inflate=0.05, translate=(0.10, 0.05, 0.0), twist_deg=20.0,
                   scene_scale=1.0, copy=False, correct_direction=True, n_threads=0):
    """One cage edit around the arm of the solid: V = (n+1)^3, T = 6 n^3 (n = 10: V = 1331, T = 6000).

    The cage is the lattice box inflated by `inflate` (so no tet vertex lies on a cage face, where the float MVC
    formula degenerates).  scene_scale scales the box about the scene centre for the aabb-16 variant.
    """
    mn, mx = np.array(box[0], np.float64), np.array(box[1], np.float64)
    mn, mx = (mn - 0.5) * scene_scale + 0.5, (mx - 0.5) * scene_scale + 0.5
    orig_verts, tets = kuhn_lattice(mn, mx, lattice_n)
    ext = (mx - mn) * inflate
    cage_v, cage_t = box_cage(mn - ext, mx + ext, 2)
    weights, labels = mvc_weights(cage_v, cage_t, orig_verts)
    cage_def = deform_cage(cage_v, tuple(t * scene_scale for t in translate), twist_deg)
    verts = mvc_apply(weights, cage_def)
    lut_off, lut_idx, _, max_per_cell = build_tet_lut(verts, tets, n_threads)
    _, _, orig_bits, _ = build_tet_lut(orig_verts, tets, n_threads)
    rot = local_rotations(verts, orig_verts, tets) if correct_direction else None
    return CageEdit(vertices=np.ascontiguousarray(verts, np.float32), original_vertices=np.ascontiguousarray(orig_verts, np.float32),
                    tets=np.ascontiguousarray(tets, np.uint32), lut_offsets=lut_off, lut_idx=lut_idx if lut_idx.size else np.zeros(1, np.uint32),
                    original_bitfield=orig_bits, local_rotations=rot, copy=copy, cage_vertices=cage_v, cage_triangles=cage_t,
                    cage_deformed=cage_def, mvc_weights=weights, mvc_labels=labels, max_per_cell=max_per_cell)


def make_affine_edit(box=((0.22, 0.60, 0.40), (0.60, 0.76, 0.60)), translation=(0.0, 0.12, 0.0), scale=(0.8, 0.9, 1.1), yaw_deg=25.0, box_yaw_deg=10.0,
                     scene_scale=1.0, hide_original=False, correct_dir=True):
    """An AffineDuplication of the solid's arm: an oriented selection box (rotated by box_yaw about y) shown again translated,
    anisotropically scaled and rotated by yaw_deg about y.  -> _abi.AffineDuplicationOp"""
    from ._abi import AffineDuplicationOp
    mn, mx = np.array(box[0], np.float64), np.array(box[1], np.float64)
    mn, mx = (mn - 0.5) * scene_scale + 0.5, (mx - 0.5) * scene_scale + 0.5

    def rot_y(deg):
        a = np.radians(deg)
        return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)

    op = AffineDuplicationOp()
    op.selection_center[:] = [float(v) for v in (0.5 * (mn + mx)).astype(np.float32)]
    op.selection_scale[:] = [float(v) for v in (mx - mn).astype(np.float32)]
    op.selection_rot[:] = [float(v) for v in rot_y(box_yaw_deg).T.reshape(-1)]   # column-major
    op.translation[:] = [float(np.float32(t * scene_scale)) for t in translation]
    op.scale[:] = [float(np.float32(v)) for v in scale]
    op.rotation[:] = [float(v) for v in rot_y(yaw_deg).T.reshape(-1)]
    op.hide_original = 1 if hide_original else 0
    op.correct_dir = 1 if correct_dir else 0
    return op


def deformed_density_grid(grid, desc, map_positions, aabb_scale=1):
    """One-shot, noise-free update_density_grid_nerf_operator (testbed_nerf.cu:3533-3640): occupancy of the EDITED scene.

    Every cell centre is pushed through the operator's map_positions (deformed -> canonical); the cell takes the
    canonical solid's value there, or 0 where the operator reports vacated space.  `map_positions(warped_pos[n,3])`
    -> (mapped[n,3], empty[n]) is supplied by the caller: libnrs on the GPU (bench) or the CPU oracle (tests).
    """
    out = np.array(grid, np.float32, copy=True)
    mn = np.array(desc.aabb_min[:], np.float32)
    diag = np.array(desc.aabb_max[:], np.float32) - mn
    for level in range(n_scene_cascades(aabb_scale)):
        c = cell_centres(level)
        inside = np.all((c >= mn) & (c <= mn + diag), axis=1)
        warped = ((c[inside] - mn) / diag).astype(np.float32)
        mapped, empty = map_positions(warped)
        moved = np.any(mapped != warped, axis=1) | (empty != 0)
        world = (mn + mapped[moved] * diag).astype(np.float32)
        vals = scene_indicator(world, aabb_scale)
        vals[empty[moved] != 0] = 0.0
        sel = np.flatnonzero(inside)[moved]
        out[level * GRID_VOLUME + sel] = vals
    return out
