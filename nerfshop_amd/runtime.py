"""Host-side mirror of the reference's operator surface for the render path, over the C-ABI of libnrs.so.

Names, argument meaning and error behaviour follow the reference classes so call sites read the same:

    NerfNetwork            <- ngp::NerfNetwork<T> / NerfNetworkFull<T>   include/neural-graphics-primitives/nerf_network.h:86
    CageDeformation        <- ngp::CageDeformation : EditOperator        include/.../editing/edit_operator.h:25, cage_deformation.cu:547
    RenderBuffer           <- ngp::CudaRenderBuffer (frame_buffer / depth_buffer / spp / clear_frame)   render_buffer.h:164
    Testbed.render_nerf    <- Testbed::render_nerf                        src/testbed_nerf.cu:3066

PyTorch is plumbing only: device memory (torch tensors), streams.  All arithmetic happens in the HIP kernels;
there is no CPU path -- a missing library or GPU raises NrsError.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _abi
from ._abi import NrsError, RenderParams, RenderStats, check


def _stream_handle(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    if isinstance(stream, torch.cuda.Stream):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


def _require_cuda(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NrsError(f"{name} must be a CUDA (HIP) tensor")
    if t.dtype != dtype:
        raise NrsError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise NrsError(f"{name} must be contiguous")


class Context:
    def __init__(self, device=0):
        self.lib = _abi.load()
        self.h = C.c_void_p()
        check(self.lib.nrs_ctx_create(int(device), C.byref(self.h)))
        self.device = int(device)
        name = C.create_string_buffer(256)
        ncu, hbm = C.c_int(), C.c_size_t()
        check(self.lib.nrs_ctx_device_info(self.h, name, 256, C.byref(ncu), C.byref(hbm)))
        self.device_name, self.n_cus, self.hbm_bytes = name.value.decode(), ncu.value, hbm.value

    def set_lane_teams(self, lanes_per_ray):
        """0 = automatic (default), 1 / 2 / 4 = lanes of a wavefront per ray in render launches, -1 = hybrid, -2 / -3 / -4 = small-launch schedule with
        16- / 32- / 64-pixel packets (nrs_ctx_set_lane_teams)."""
        check(self.lib.nrs_ctx_set_lane_teams(self.h, int(lanes_per_ray)))

    def set_ray_handover(self, enabled):
        """Waves that run out of work take rays from a sibling wave of their workgroup (on by default; nrs_ctx_set_ray_handover)."""
        check(self.lib.nrs_ctx_set_ray_handover(self.h, int(bool(enabled))))

    def ray_handovers(self):
        """(rays moved, hand-overs) of the last render launch that returned statistics."""
        n, k = C.c_uint64(), C.c_uint64()
        check(self.lib.nrs_ctx_ray_handovers(self.h, C.byref(n), C.byref(k)))
        return n.value, k.value

    def close(self):
        if self.h:
            self.lib.nrs_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


DEFAULT_CELL_CACHE_BYTES = 10 << 30  # levels 0..11 of base.json's table (9.2 GB)


class NerfNetwork:
    """pos-encoding (HashGrid) -> density MLP -> (SH dir-encoding | density features) -> RGB MLP -> extract_density."""

    def __init__(self, ctx, desc, cell_cache_bytes=DEFAULT_CELL_CACHE_BYTES):
        """cell_cache_bytes: budget of the cell-record cache this HARNESS opts into (nrs_model_set_cell_cache: opt-in at the C-ABI since round 6; never more than a
        quarter of the free HBM); 0 = none."""
        self.ctx, self.lib, self.desc = ctx, ctx.lib, desc
        self.h = C.c_void_p()
        check(self.lib.nrs_model_create(ctx.h, C.byref(desc), C.byref(self.h)))
        if cell_cache_bytes and "NRS_CELL_CACHE_GB" not in os.environ:
            free_b = torch.cuda.mem_get_info(ctx.device)[0]
            try:
                self.set_cell_cache(min(int(cell_cache_bytes), free_b // 4))
            except NrsError:
                self.set_cell_cache(0)  # an optimisation: render without it

    # -- NerfNetwork<T> accessors (nerf_network.h:97-120)
    def padded_output_width(self):
        return 16

    def input_width(self):
        return 7

    def n_extra_dims(self):
        return 0

    def n_params(self):
        return int(self.lib.nrs_model_n_params(C.byref(self.desc)))

    def set_params(self, params_fp16):
        """fp16 parameter blob in tiny-cuda-nn order (density | rgb | grid), a host array (numpy uint16/float16)."""
        p = np.ascontiguousarray(params_fp16)
        if p.dtype == np.float16:
            p = p.view(np.uint16)
        if p.dtype != np.uint16:
            raise NrsError("set_params expects fp16 parameters (numpy float16 or their uint16 bits)")
        check(self.lib.nrs_model_set_params(self.h, p.ctypes.data, p.size))

    def set_params_device(self, params_fp16_cuda, stream=None):
        """NerfNetwork::set_params with device pointers: a CUDA tensor of fp16 (or the same bits as int16/uint16) in tcnn order."""
        t = params_fp16_cuda
        if not t.is_cuda or t.element_size() != 2 or not t.is_contiguous():
            raise NrsError("set_params_device expects a contiguous 2-byte CUDA tensor")
        check(self.lib.nrs_model_set_params_device(self.h, t.data_ptr(), t.numel(), _stream_handle(stream)))

    def set_numerics(self, grid_acc=0, mlp_acc=0):
        """tiny-cuda-nn's two unpinned roundings (nrs_model_set_numerics): grid_acc 0 = fp32 sum rounded once, 1 = per-corner fp16 accumulation;
        mlp_acc 0 = fp32 accumulators, 1 = fp16 rounding of the running sum every 16-wide k step."""
        check(self.lib.nrs_model_set_numerics(self.h, int(grid_acc), int(mlp_acc)))

    def set_cell_cache(self, max_bytes):
        """Budget of the cell-record cache (nrs_model_set_cell_cache): 0 drops it; results do not depend on it."""
        check(self.lib.nrs_model_set_cell_cache(self.h, int(max_bytes)))

    def cell_cache(self):
        """(bytes held by the cell records, number of levels they cover)"""
        n = C.c_uint32()
        b = self.lib.nrs_model_cell_cache_bytes(self.h, C.byref(n))
        return int(b), int(n.value)

    def set_sparse_cell_cache(self, mask_bitfield_u8, max_bytes):
        """Sparse brick records for the levels after the dense ones (nrs_model_set_sparse_cell_cache); mask = density-bitfield layout."""
        if mask_bitfield_u8 is None:
            check(self.lib.nrs_model_set_sparse_cell_cache(self.h, None, 0))
            return
        b = np.ascontiguousarray(mask_bitfield_u8, np.uint8)
        assert b.size == _abi.BITFIELD_BYTES
        check(self.lib.nrs_model_set_sparse_cell_cache(self.h, b.ctypes.data, int(max_bytes)))

    def sparse_cell_cache(self):
        """(bytes held by brick tables + records, first sparse level, number of sparse levels)"""
        f, n = C.c_uint32(), C.c_uint32()
        b = self.lib.nrs_model_sparse_cell_cache_bytes(self.h, C.byref(f), C.byref(n))
        return int(b), int(f.value), int(n.value)

    def set_density_bitfield(self, bitfield_u8):
        b = np.ascontiguousarray(bitfield_u8, np.uint8)
        check(self.lib.nrs_model_set_density_bitfield(self.h, b.ctypes.data, b.size))

    def set_density_grid(self, grid_f32):
        g = np.ascontiguousarray(grid_f32, np.float32)
        check(self.lib.nrs_model_set_density_grid(self.h, g.ctypes.data, g.size))

    def get_density_bitfield(self):
        out = np.zeros(_abi.BITFIELD_BYTES, np.uint8)
        check(self.lib.nrs_model_get_density_bitfield(self.h, out.ctypes.data, out.size))
        return out

    def get_march_accelerator(self, which):
        """Test hook: (box12 = min, max, cell, 1/cell; mask bits [32, 32, 32] as bool, z-major) of the marching accelerator."""
        box = np.zeros(12, np.float32)
        mask = np.zeros(1024, np.uint32)
        check(self.lib.nrs_model_get_march_accelerator(self.h, int(which), box.ctypes.data, mask.ctypes.data))
        bits = np.unpackbits(mask.view(np.uint8), bitorder="little").reshape(32, 32, 32).astype(bool)
        return box, bits

    def get_density_grid(self):
        out = np.zeros(_abi.GRID_VOLUME * _abi.GRID_CASCADES, np.float32)
        check(self.lib.nrs_model_get_density_grid(self.h, out.ctypes.data, out.size))
        return out

    def inference_mixed_precision(self, stream, input, output):
        """input: [n, 7] f32 (tcnn: column-major 7 x n).  output: fp16, [16, n_el] (row-major planes, n_el >= n)
        or [n, 16] (column-major / interleaved), as GPUMatrixDynamic's layout selects in the reference."""
        _require_cuda(input, torch.float32, "input")
        _require_cuda(output, torch.float16, "output")
        if input.dim() != 2 or input.shape[1] != 7:
            raise NrsError("NerfNetwork::inference_mixed_precision input must be [n, 7]")
        n = input.shape[0]
        layout, ld = self._out_layout(output, n)
        check(self.lib.nrs_network_inference(self.h, _stream_handle(stream), n, input.data_ptr(), output.data_ptr(), ld, layout))

    def density(self, stream, input, output):
        """input: [n, ld] f32 with ld in 3..7 (only the position is read); output as above, the density MLP's 16 outputs."""
        _require_cuda(input, torch.float32, "input")
        _require_cuda(output, torch.float16, "output")
        if input.dim() != 2 or not (3 <= input.shape[1] <= 7):
            raise NrsError("NerfNetwork::density input must be in column major format ([n, 3..7] here).")
        n = input.shape[0]
        layout, ld = self._out_layout(output, n)
        check(self.lib.nrs_network_density(self.h, _stream_handle(stream), n, input.data_ptr(), input.shape[1], output.data_ptr(), ld, layout))

    def input_gradient(self, stream, input, output):
        """NerfNetwork::input_gradient(stream, 3, input, output): d density_raw / d position, input [n, >= 3] f32, output [n, 3] f32 (cuda tensors)."""
        _require_cuda(input, torch.float32, "input")
        _require_cuda(output, torch.float32, "output")
        n = input.shape[0]
        assert output.shape == (n, 3) and output.is_contiguous() and input.is_contiguous()
        check(self.lib.nrs_network_input_gradient(self.h, _stream_handle(stream), n, input.data_ptr(), input.shape[1], output.data_ptr()))

    def visualize_activation(self, stream, layer, dimension, input, output):
        """Network::visualize_activation: unit `dimension` of forward_activations(layer), input [n, 7] f32, output [n] f32 (cuda tensors)."""
        _require_cuda(input, torch.float32, "input")
        _require_cuda(output, torch.float32, "output")
        n = input.shape[0]
        assert input.shape[1] == 7 and output.shape == (n,) and output.is_contiguous() and input.is_contiguous()
        check(self.lib.nrs_network_visualize_activation(self.h, _stream_handle(stream), int(layer), int(dimension), n, input.data_ptr(), output.data_ptr()))

    def hashgrid_encode(self, stream, input, output):
        _require_cuda(input, torch.float32, "input")
        _require_cuda(output, torch.float16, "output")
        n = input.shape[0]
        if tuple(output.shape) != (n, 32):
            raise NrsError("hashgrid_encode output must be [n, 32]")
        check(self.lib.nrs_hashgrid_encode(self.h, _stream_handle(stream), n, input.data_ptr(), input.shape[1], output.data_ptr()))

    @staticmethod
    def _out_layout(output, n):
        if output.dim() == 2 and output.shape[0] == 16 and output.shape[1] >= n:
            return _abi.LAYOUT_PLANES, int(output.shape[1])
        if output.dim() == 2 and output.shape[1] == 16 and output.shape[0] >= n:
            return _abi.LAYOUT_INTERLEAVED, 16
        raise NrsError("output must be [16, n_el] (planes) or [n, 16] (interleaved) fp16")

    def close(self):
        if self.h:
            self.lib.nrs_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CageDeformation:
    """One cage-deformation edit operator; owns its GPU tables (tet_mesh.h:80-94)."""

    def __init__(self, ctx, desc, cage_edit, device_authoring=False):
        self.ctx, self.lib = ctx, ctx.lib
        self.host = cage_edit  # keeps the numpy arrays alive
        self.h = C.c_void_p()
        mesh = cage_edit.tet_mesh_struct(device_authoring) if device_authoring else cage_edit.tet_mesh_struct()
        self.n_vertices, self.n_tets = int(mesh.n_vertices), int(mesh.n_tets)
        check(self.lib.nrs_edit_create(ctx.h, C.byref(desc), C.byref(mesh), C.byref(self.h)))

    # ---- the per-gizmo-move chain on the device (nrs.h "next" row f1) ----
    def set_mvc(self, weights):
        """Cage::compute_mvc's [V, n_cage_vertices] weights, uploaded once."""
        w = np.ascontiguousarray(weights, np.float32)
        if w.ndim != 2 or w.shape[0] != self.n_vertices:
            raise NrsError("set_mvc expects weights [n_vertices, n_cage_vertices]")
        check(self.lib.nrs_edit_set_mvc(self.h, w.ctypes.data, w.shape[1]))

    def update_cage(self, stream, cage_vertices):
        """Cage::interpolate_with_mvc + post_update_vertices + build_tet_grid + update_local_rotations for a moved cage."""
        c = np.ascontiguousarray(cage_vertices, np.float32)
        check(self.lib.nrs_edit_update_cage(self.h, _stream_handle(stream), c.ctypes.data, c.shape[0]))

    def update_vertices(self, stream, vertices):
        v = np.ascontiguousarray(vertices, np.float32)
        check(self.lib.nrs_edit_update_vertices(self.h, _stream_handle(stream), v.ctypes.data, v.shape[0]))

    def poisson_interpolate(self, stream, inside_density, outside_density, inside_shs, outside_shs, residual_amplitude=1.0, gamma=None):
        """GrowingSelection::interpolate_poisson_boundary: per-cage-vertex membrane terms -> the operator's per-tet-vertex ones (on the device)."""
        i_d, o_d = np.ascontiguousarray(inside_density, np.float32), np.ascontiguousarray(outside_density, np.float32)
        i_s, o_s = np.ascontiguousarray(inside_shs, np.float32).reshape(-1, 27), np.ascontiguousarray(outside_shs, np.float32).reshape(-1, 27)
        g = np.ascontiguousarray(gamma, np.float32) if gamma is not None else None
        check(self.lib.nrs_edit_poisson_interpolate(self.h, _stream_handle(stream), g.ctypes.data if g is not None else None, i_d.size, i_d.ctypes.data, o_d.ctypes.data,
                                                    i_s.ctypes.data, o_s.ctypes.data, float(residual_amplitude)))

    def download_poisson(self, n_vertices):
        sh, od, rd = np.zeros((n_vertices, 27), np.float32), np.zeros(n_vertices, np.float32), np.zeros(n_vertices, np.float32)
        check(self.lib.nrs_edit_download_poisson(self.h, sh.ctypes.data, od.ctypes.data, rd.ctypes.data))
        return sh, od, rd

    def lut_size(self):
        n, m = C.c_uint32(), C.c_uint32()
        check(self.lib.nrs_edit_lut_size(self.h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def download(self, rotations=True):
        """-> dict(vertices, lut_offsets, lut_idx, rotations | None, original_bitfield, bbox)"""
        n_idx, _ = self.lut_size()
        out = dict(vertices=np.zeros((self.n_vertices, 3), np.float32), lut_offsets=np.zeros(_abi.N_LUT_CELLS + 1, np.uint32),
                   lut_idx=np.zeros(max(n_idx, 1), np.uint32), rotations=np.zeros((self.n_tets, 9), np.float32) if rotations else None,
                   original_bitfield=np.zeros(_abi.BITFIELD_BYTES, np.uint8), bbox=np.zeros(6, np.float32))
        check(self.lib.nrs_edit_download(self.h, out["vertices"].ctypes.data, out["lut_offsets"].ctypes.data, out["lut_idx"].ctypes.data,
                                         out["rotations"].ctypes.data if rotations else None, out["original_bitfield"].ctypes.data,
                                         out["bbox"].ctypes.data))
        out["lut_idx"] = out["lut_idx"][:n_idx]
        return out

    def map_rays(self, stream, nerf_coords, empty_mask):
        _require_cuda(nerf_coords, torch.float32, "nerf_coords")
        _require_cuda(empty_mask, torch.uint8, "empty_mask")
        if nerf_coords.dim() != 2 or nerf_coords.shape[1] != 7 or empty_mask.numel() < nerf_coords.shape[0]:
            raise NrsError("map_rays expects coords [n, 7] and an empty mask of n bytes")
        check(self.lib.nrs_edit_map_rays(self.h, _stream_handle(stream), nerf_coords.shape[0], nerf_coords.data_ptr(), empty_mask.data_ptr()))

    def map_positions(self, stream, nerf_pos, empty_mask):
        _require_cuda(nerf_pos, torch.float32, "nerf_pos")
        _require_cuda(empty_mask, torch.uint8, "empty_mask")
        if nerf_pos.dim() != 2 or nerf_pos.shape[1] < 3 or empty_mask.numel() < nerf_pos.shape[0]:
            raise NrsError("map_positions expects positions [n, >=3] and an empty mask of n bytes")
        check(self.lib.nrs_edit_map_positions(self.h, _stream_handle(stream), nerf_pos.shape[0], nerf_pos.data_ptr(), nerf_pos.shape[1],
                                              empty_mask.data_ptr()))

    def close(self):
        if self.h:
            self.lib.nrs_edit_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AffineDuplication(CageDeformation):
    """AffineDuplication edit operator (editing/affine_duplication.h): shares map_rays / map_positions with the cage operator."""

    def __init__(self, ctx, desc, op):
        self.ctx, self.lib = ctx, ctx.lib
        self.host = op
        self.h = C.c_void_p()
        self.n_vertices = self.n_tets = 0
        check(self.lib.nrs_edit_create_affine(ctx.h, C.byref(desc), C.byref(op), C.byref(self.h)))


class RenderBuffer:
    """frame_buffer(): f32x4 premultiplied linear RGBA [H, W, 4]; depth_buffer(): f32 [H, W]; spp(): Sobol sample index."""

    def __init__(self, width, height, device="cuda:0", with_steps=False):
        self.width, self.height = int(width), int(height)
        self._frame = torch.zeros((self.height, self.width, 4), dtype=torch.float32, device=device)
        self._depth = torch.zeros((self.height, self.width), dtype=torch.float32, device=device)
        self._steps = torch.zeros((self.height, self.width), dtype=torch.int32, device=device) if with_steps else None
        self._spp = 0

    def in_resolution(self):
        return (self.width, self.height)

    def frame_buffer(self):
        return self._frame

    def depth_buffer(self):
        return self._depth

    def steps_buffer(self):
        return self._steps

    def spp(self):
        return self._spp

    def set_spp(self, v):
        self._spp = int(v)

    def accumulate(self, ctx, stream=None, color_space=0):
        """CudaRenderBuffer::accumulate (render_buffer.cu:540): the frame buffer joins the running mean of this view's spp frames; spp() counts them."""
        if getattr(self, "_accumulate", None) is None:
            self._accumulate = torch.zeros_like(self._frame)
        check(_abi.load().nrs_accumulate(ctx.h, _stream_handle(stream), self.width, self.height, self._frame.data_ptr(), self._accumulate.data_ptr(), int(self._spp), int(color_space)))
        self._spp += 1
        return self._accumulate

    def clear_frame(self, stream=None):
        self._frame.zero_()
        self._depth.zero_()


def set_camera_extras(p, render_distortion=None, distortion_map=None, envmap=None):
    """Camera model and background of an nrs_render_params (init_rays_from_camera's arguments, testbed_nerf.cu:3078-3100): lens distortion (mode, 7 params),
    the distortion map [H, W, 2] and the environment map [H, W, 4] as float32 CUDA tensors (the struct keeps raw device pointers: keep the tensors alive)."""
    if render_distortion is not None:
        p.distortion_mode = int(render_distortion[0])
        p.distortion_params[:] = [float(v) for v in render_distortion[1]]
    if distortion_map is not None:
        _require_cuda(distortion_map, torch.float32, "distortion_map")
        p.d_distortion_map = distortion_map.data_ptr()
        p.distortion_resolution[:] = (distortion_map.shape[1], distortion_map.shape[0])
    if envmap is not None:
        _require_cuda(envmap, torch.float32, "envmap")
        p.d_envmap = envmap.data_ptr()
        p.envmap_resolution[:] = (envmap.shape[1], envmap.shape[0])
    return p


class Testbed:
    """The slice of ngp::Testbed the render path reads: network, occupancy, edit operators and the render knobs."""

    def __init__(self, ctx, desc, aabb_scale=1):
        self.ctx, self.lib, self.desc = ctx, ctx.lib, desc
        self.nerf_network = NerfNetwork(ctx, desc)
        self.edit_operators = []          # NerfTracer::m_edit_operators, applied last-to-first
        self.enable_edits = True          # m_enable_edits
        self.snap_to_pixel_centers = True
        self.rendering_min_transmittance = 0.01
        self.cone_angle_constant = 0.0 if aabb_scale <= 1 else 1.0 / 256.0
        self.render_mode = _abi.RENDER_SHADE
        self.linear_colors = False
        self.show_accel = -1              # m_nerf.show_accel
        self.dof = 0.0                    # m_dof
        self.slice_plane_z, self.scale = 0.0, 1.0   # m_slice_plane_z, m_scale
        self.dataset_scale = 1.0          # m_nerf.training.dataset.scale
        self.render_distortion = (0, (0.0,) * 7)   # m_nerf.render_distortion (mode, params) when render_with_camera_distortion
        self.distortion_map = None        # m_distortion.map: float32 CUDA tensor [H, W, 2] or None
        self.envmap = None                # m_envmap.envmap: float32 CUDA tensor [H, W, 4] or None
        self.glow_mode, self.glow_y_cutoff = 0, 0.0   # m_nerf.m_glow_mode / m_glow_y_cutoff
        mn, mx = list(desc.aabb_min), list(desc.aabb_max)
        self.render_aabb = (mn, mx)       # m_render_aabb
        self.last_stats = None

    def add_edit_operator(self, op):
        self.edit_operators.append(op)

    def make_params(self, render_buffer, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center, apply_operators):
        p = RenderParams()
        p.resolution[:] = render_buffer.in_resolution()
        p.focal_length[:] = list(focal_length)
        p.camera_matrix0[:] = [float(v) for v in np.asarray(camera_matrix0, np.float32).reshape(-1)]
        p.camera_matrix1[:] = [float(v) for v in np.asarray(camera_matrix1, np.float32).reshape(-1)]
        p.rolling_shutter[:] = list(rolling_shutter)
        p.screen_center[:] = list(screen_center)
        p.render_aabb_min[:] = self.render_aabb[0]
        p.render_aabb_max[:] = self.render_aabb[1]
        p.spp_index = render_buffer.spp()
        p.snap_to_pixel_centers = 1 if self.snap_to_pixel_centers else 0
        p.min_transmittance = self.rendering_min_transmittance
        p.cone_angle_constant = self.cone_angle_constant
        p.render_mode = self.render_mode
        p.linear_colors = 1 if self.linear_colors else 0
        p.apply_operators = 1 if apply_operators else 0
        p.min_mip = self.show_accel if self.show_accel >= 0 else 0
        p.show_accel = 1 if self.show_accel >= 0 else 0
        p.dof = self.dof
        p.slice_plane_z = self.slice_plane_z + self.scale  # testbed_nerf.cu:3067
        p.depth_scale = 1.0 / self.dataset_scale           # :3113
        p.glow_mode, p.glow_y_cutoff = self.glow_mode, self.glow_y_cutoff
        set_camera_extras(p, self.render_distortion, self.distortion_map, self.envmap)
        return p

    def render_nerf(self, network, render_buffer, max_res, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center,
                    apply_operators, stream=None, want_stats=False):
        """Testbed::render_nerf(network, render_buffer, max_res, focal_length, camera_matrix0, camera_matrix1, rolling_shutter,
        screen_center, apply_operators, stream).  The frame buffer must have been cleared by the caller (render_frame does)."""
        p = self.make_params(render_buffer, focal_length, camera_matrix0, camera_matrix1, rolling_shutter, screen_center,
                             apply_operators and self.enable_edits)
        return self.render_with_params(network, p, render_buffer.frame_buffer(), render_buffer.depth_buffer(), render_buffer.steps_buffer(),
                                       stream, want_stats)

    def render_with_params(self, network, p, frame, depth, steps=None, stream=None, want_stats=False):
        _require_cuda(frame, torch.float32, "frame_buffer")
        _require_cuda(depth, torch.float32, "depth_buffer")
        n = len(self.edit_operators)
        arr = (C.c_void_p * max(n, 1))(*[op.h for op in self.edit_operators])
        stats = RenderStats() if want_stats else None
        check(self.lib.nrs_render_nerf(network.h, C.byref(p), arr, n, frame.data_ptr(), depth.data_ptr(),
                                       steps.data_ptr() if steps is not None else None, _stream_handle(stream),
                                       C.byref(stats) if stats is not None else None))
        self.last_stats = stats
        return stats

    def get_density_on_grid(self, res3d, aabb_min, aabb_max, mask_with_density_grid=True, stream=None):
        """Testbed::get_density_on_grid(res3d, aabb) (testbed_nerf.cu:4538) -> float32 CUDA tensor [rz, ry, rx]"""
        res = (C.c_uint32 * 3)(*[int(v) for v in res3d])
        mn, mx = (C.c_float * 3)(*aabb_min), (C.c_float * 3)(*aabb_max)
        out = torch.zeros((int(res3d[2]), int(res3d[1]), int(res3d[0])), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        check(self.lib.nrs_density_on_grid(self.nerf_network.h, _stream_handle(stream), C.byref(res), C.byref(mn), C.byref(mx),
                                           1 if mask_with_density_grid else 0, out.data_ptr()))
        return out

    def project_selection_pixels(self, params, pixels_xy, transmittance_threshold=0.1, automatic_max_level=True, growing_level=0, stream=None):
        """GrowingSelection::project_selection_pixels (growing_selection.cu:1832): scribbled pixels -> surface points and the
        occupancy cells they fall into.  Returns (positions [n, 3], cells [n], found [n]) per pixel plus the de-duplicated
        (cells, positions, growing_level) the region growing starts from."""
        px = torch.as_tensor(np.ascontiguousarray(pixels_xy, np.int32).reshape(-1, 2), device=f"cuda:{self.ctx.device}")
        n = px.shape[0]
        pos = torch.zeros((n, 3), dtype=torch.float32, device=px.device)
        cells = torch.zeros((n,), dtype=torch.int32, device=px.device)
        found = torch.zeros((n,), dtype=torch.uint8, device=px.device)
        check(self.lib.nrs_project_selection_pixels(self.nerf_network.h, _stream_handle(stream), C.byref(params), px.data_ptr(), n,
                                                    float(transmittance_threshold), pos.data_ptr(), cells.data_ptr(), found.data_ptr()))
        if stream is not None:
            stream.synchronize()
        else:
            torch.cuda.synchronize()
        h_pos, h_cells, h_found = pos.cpu().numpy(), cells.cpu().numpy().view(np.uint32), found.cpu().numpy()
        level = C.c_uint32(int(growing_level))
        out_cells, out_pos, n_out = np.zeros(n, np.uint32), np.zeros((n, 3), np.float32), C.c_uint32()
        check(self.lib.nrs_selection_cells(h_pos.ctypes.data, h_cells.ctypes.data, h_found.ctypes.data, n, 1 if automatic_max_level else 0,
                                           C.byref(level), out_cells.ctypes.data, out_pos.ctypes.data, C.byref(n_out)))
        return (h_pos, h_cells, h_found), (out_cells[: n_out.value], out_pos[: n_out.value], level.value)

    def compute_poisson_boundary(self, vertices, is_inside, jitter, sh_sampling_width=10, hemisphere_width=10):
        """GrowingSelection::compute_poisson_boundary (growing_selection.cu:2220): per cage vertex the density and the SH9 fit of
        the colours around it.  jitter: [n_verts * w * w, 2] in [0, 1] (the reference's std::rand() draws).  -> (density [n], sh [n, 27])"""
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        jt = np.ascontiguousarray(jitter, np.float32)
        n = v.shape[0]
        if jt.size != n * sh_sampling_width * sh_sampling_width * 2:
            raise NrsError("compute_poisson_boundary: jitter must hold two draws per sample")
        density, sh = np.zeros(n, np.float32), np.zeros((n, 27), np.float32)
        check(self.lib.nrs_poisson_boundary(self.nerf_network.h, v.ctypes.data, n, int(sh_sampling_width), int(hemisphere_width), jt.ctypes.data,
                                            1 if is_inside else 0, density.ctypes.data, sh.ctypes.data))
        return density, sh

    def get_rgba_on_grid(self, res3d, ray_dir, stream=None):
        """Testbed::get_rgba_on_grid(res3d, ray_dir) (testbed_nerf.cu:4588) over m_render_aabb -> float32 CUDA tensor [rz, ry, rx, 4]"""
        res = (C.c_uint32 * 3)(*[int(v) for v in res3d])
        mn, mx, rd = (C.c_float * 3)(*self.render_aabb[0]), (C.c_float * 3)(*self.render_aabb[1]), (C.c_float * 3)(*ray_dir)
        out = torch.zeros((int(res3d[2]), int(res3d[1]), int(res3d[0]), 4), dtype=torch.float32, device=f"cuda:{self.ctx.device}")
        check(self.lib.nrs_rgba_on_grid(self.nerf_network.h, _stream_handle(stream), C.byref(res), C.byref(mn), C.byref(mx), C.byref(rd), out.data_ptr()))
        return out

    def new_grid_update(self, max_cascade=0, seed=1337, decay=0.95):
        """The Testbed members update_density_grid_nerf_operator reads: m_rng = default_rng_t{m_seed} (testbed.cu:2220),
        density_grid_ema_step = 0, density_grid_decay = 0.95 (testbed.h:604), sized as update_density_grid_nerf_render does."""
        u = _abi.GridUpdate()
        u.n_uniform_samples = _abi.GRID_VOLUME * (max_cascade + 1)
        u.n_nonuniform_samples = 0
        u.reset_grid = 0
        u.max_cascade = max_cascade
        u.decay = decay
        u.ema_step = 0
        st, inc = C.c_uint64(), C.c_uint64()
        self.lib.nrs_rng_seed(seed, C.byref(st), C.byref(inc))
        u.rng_state, u.rng_inc = st.value, inc.value
        return u

    def update_density_grid_nerf_operator(self, update, stream=None, apply_operators=True):
        """Testbed::update_density_grid_nerf_operator (testbed_nerf.cu:3533): one refresh of the occupancy in deformed space."""
        ops = self.edit_operators if (apply_operators and self.enable_edits) else []
        arr = (C.c_void_p * max(len(ops), 1))(*[op.h for op in ops])
        check(self.lib.nrs_model_update_density_grid(self.nerf_network.h, arr, len(ops), C.byref(update), _stream_handle(stream)))

    def update_density_grid_nerf_render(self, n_iterations, reset_grid, update, stream=None):
        """Testbed::update_density_grid_nerf_render (testbed_nerf.cu:3514)."""
        for i in range(n_iterations):
            update.reset_grid = 1 if (reset_grid and i == 0) else 0
            self.update_density_grid_nerf_operator(update, stream)
        update.reset_grid = 0

    def trace_samples(self, p, pixel_idx, max_samples, stream=None):
        """Test hook: (t, dt) stream per listed pixel -> (t [n, max], dt [n, max], count [n]) as CUDA tensors."""
        _require_cuda(pixel_idx, torch.int32, "pixel_idx")
        n = pixel_idx.numel()
        dev = pixel_idx.device
        t = torch.zeros((n, max_samples), dtype=torch.float32, device=dev)
        dt = torch.zeros((n, max_samples), dtype=torch.float32, device=dev)
        cnt = torch.zeros(n, dtype=torch.int32, device=dev)
        check(self.lib.nrs_trace_samples(self.nerf_network.h, C.byref(p), _stream_handle(stream), n, pixel_idx.data_ptr(), max_samples,
                                         t.data_ptr(), dt.data_ptr(), cnt.data_ptr()))
        return t, dt, cnt
