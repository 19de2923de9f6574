#!/usr/bin/env python3
"""Build helper of oracle/_ref (TEST INFRASTRUCTURE): cuts the render-path functions out of the reference's .cu files.

The reference's headers on the path compile whole against oracle/ref_stubs (a stand-in for the empty Eigen / tiny-cuda-nn /
tinylogger submodules plus a CUDA-keyword shim).  Its .cu translation units do not -- they pull in the GUI, the trainer and
tcnn's network classes -- so the kernels on the path are lifted out of them, each from its first line to the closing brace
in column 0, and #included by oracle/ref_render.cpp.  Every fragment starts with a `#line` directive naming the reference file
and line it came from, so compiler diagnostics and debuggers point into /root/reference.

Fragments are written to a scratch directory handed in by oracle/Makefile and deleted after the compile: no reference source is
ever stored in this repository (oracle/_ref holds only the resulting .so, and is git-ignored).

usage: ref_extract.py <reference root> <output dir>
"""
import os
import re
import sys

# (file, fragment name, regex of the first line, how the fragment ends)
#   "fn"    : up to the first following line that is exactly "}"
#   "block:<owner>:<n>" : brace-balanced block that starts on the n-th matching line inside member function <owner>
#   "until:<regex>" : up to the first following line matching <regex>; "before:<regex>" : up to the line before it
FRAGMENTS = [
    ("src/render_buffer.cu", "accumulate_kernel", r"^__global__ void accumulate_kernel\(", "fn"),
    ("src/testbed_nerf.cu", "march_constants", r"^static constexpr uint32_t MARCH_ITER", r"until:^static constexpr uint32_t MAX_STEPS_INBETWEEN_COMPACTION"),
    ("include/neural-graphics-primitives/editing/datastructures/tet_mesh.h", "corner_offsets", r"^static const std::vector<Eigen::Vector3f> corner_offsets = \{", r"until:^\};"),
    ("src/testbed_nerf.cu", "network_to_rgb_derivative", r"^__device__ float network_to_rgb_derivative\(", "fn"),
    ("src/testbed_nerf.cu", "network_to_density_derivative", r"^__device__ float network_to_density_derivative\(", "fn"),
    ("src/testbed_nerf.cu", "splat_already_activated", r"^__global__ void splat_grid_samples_nerf_max_nearest_neighbor_already_activated\(", "fn"),
    ("src/testbed_nerf.cu", "ema_grid_samples_nerf", r"^__global__ void ema_grid_samples_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "grid_to_bitfield", r"^__global__ void grid_to_bitfield\(", "fn"),
    ("src/testbed_nerf.cu", "bitfield_max_pool", r"^__global__ void bitfield_max_pool\(", "fn"),
    ("src/testbed_nerf.cu", "advance_pos_nerf", r"^__global__ void advance_pos_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "generate_next_nerf_network_inputs", r"^__global__ void generate_next_nerf_network_inputs\(", "fn"),
    ("src/testbed_nerf.cu", "composite_kernel_nerf", r"^__global__ void composite_kernel_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "shade_kernel_nerf", r"^__global__ void shade_kernel_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "compact_kernel_nerf", r"^__global__ void compact_kernel_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "init_rays_with_payload_kernel_nerf", r"^__global__ void init_rays_with_payload_kernel_nerf\(", "fn"),
    ("src/testbed_nerf.cu", "activate_network_density", r"^__global__ void activate_network_density\(", "fn"),
    ("src/testbed_nerf.cu", "generate_grid_samples_nerf_uniform", r"^__global__ void generate_grid_samples_nerf_uniform\(", "fn"),
    ("src/testbed_nerf.cu", "generate_grid_samples_nerf_uniform_dir", r"^__global__ void generate_grid_samples_nerf_uniform_dir\(", "fn"),
    ("src/testbed_nerf.cu", "grid_samples_half_to_float", r"^__global__ void grid_samples_half_to_float\(", "fn"),
    ("src/testbed_nerf.cu", "compute_nerf_density", r"^__global__ void compute_nerf_density\(", "fn"),
    ("src/testbed_nerf.cu", "generate_nerf_network_inputs_at_current_position", r"^__global__ void generate_nerf_network_inputs_at_current_position\(", "fn"),
    # (the "template <typename T>" line above it is supplied by the including file)
    ("src/testbed_nerf.cu", "clear_empty_space", r"^__global__ void clear_empty_space\(", "fn"),
    ("src/editing/cage_deformation.cu", "interpolate_tet_pos", r"^__global__ void interpolate_tet_pos\(", "fn"),
    ("src/editing/cage_deformation.cu", "interpolate_tet", r"^__global__ void interpolate_tet\(", "fn"),
    ("src/editing/cage_deformation.cu", "compute_poisson_residual_density_kernel", r"^__global__ void compute_poisson_residual_density_kernel\(", "fn"),
    ("src/editing/cage_deformation.cu", "compute_residual_poisson_kernel", r"^__global__ void compute_residual_poisson_kernel\(", "fn"),
    ("include/neural-graphics-primitives/editing/tools/affine_bounding_box.cuh", "affine_bounding_box_struct", r"^struct AffineBoundingBox \{", r"before:^    nlohmann::json to_json\(\) const \{"),
    ("include/neural-graphics-primitives/editing/affine_duplication.h", "update_destination", r"^    void update_destination\(\) \{", "block::0"),
    ("src/editing/affine_duplication.cu", "warp_direction_ad", r"^__device__ Vector3f warp_direction_ad\(", "fn"),
    ("src/editing/affine_duplication.cu", "unwarp_direction_ad", r"^__device__ Vector3f unwarp_direction_ad\(", "fn"),
    ("src/editing/affine_duplication.cu", "translate_in_box_pos", r"^__global__ void translate_in_box_pos\(", "fn"),
    ("src/editing/affine_duplication.cu", "translate_in_box", r"^__global__ void translate_in_box\(", "fn"),
    ("src/editing/tools/selection_utils.cu", "get_upper_cell_idx", r"^uint32_t get_upper_cell_idx\(", "fn"),
    ("src/editing/tools/selection_utils.cu", "get_cell_pos", r"^Eigen::Vector3f get_cell_pos\(", "fn"),
    ("src/editing/tools/selection_utils.cu", "get_cell_at_pos", r"^Eigen::Vector3i get_cell_at_pos\(", "fn"),
    ("src/editing/tools/growing_selection.cu", "shoot_selection_rays_kernel", r"^__global__ void shoot_selection_rays_kernel\(", "fn"),
    ("src/editing/tools/growing_selection.cu", "composite_shot_rays", r"^__global__ void composite_shot_rays\(", "fn"),
    ("src/editing/tools/growing_selection.cu", "activate_network_output", r"^__global__ void activate_network_output\(", "fn"),
    ("src/editing/tools/growing_selection.cu", "filter_empty", r"^__global__ void filter_empty\(", "fn"),
    # GrowingSelection::compute_poisson_boundary: the direction sampling loop (std::rand jitter), the density pick and the SH9 fit loop
    ("src/editing/tools/growing_selection.cu", "poisson_boundary_sampling_loop", r"^\tfor \(uint32_t k = 0; k < n_verts; k\+\+\) \{", "block:compute_poisson_boundary:0"),
    ("src/editing/tools/growing_selection.cu", "poisson_boundary_density_loop", r"^\tfor \(int k = 0; k < n_verts; k\+\+\) \{", "block:compute_poisson_boundary:0"),
    ("src/editing/tools/growing_selection.cu", "poisson_boundary_fit_loop", r"^\tfor \(int k = 0; k < n_verts; k\+\+\) \{", "block:compute_poisson_boundary:1"),
    ("src/editing/tools/growing_selection.cu", "activate_network_output", r"^__global__ void activate_network_output\(", "fn"),
    ("src/editing/tools/growing_selection.cu", "filter_empty", r"^__global__ void filter_empty\(", "fn"),
    # GrowingSelection::interpolate_poisson_boundary: the per-tet-vertex loop (MVC-weighted transfer of the cage's membrane terms)
    ("src/editing/tools/growing_selection.cu", "interpolate_poisson_boundary_loop", r"^\tfor \(int i = 0; i < n_tet_vertices; i\+\+\) \{", "block:interpolate_poisson_boundary:0"),
    # TetMesh::update_local_rotations: the per-tet loop (centroids, correlation matrix, svd_eigen, R = U V^T)
    ("src/editing/datastructures/tet_mesh.cu", "update_local_rotations_loop", r"^\tfor \(int i = 0; i < n_tets; i\+\+\) \{", "block:update_local_rotations:0"),
    # Cage::interpolate_with_mvc(weights, points): the accumulation loop
    ("src/editing/datastructures/cage.cu", "interpolate_with_mvc_loop", r"^\tfor \(int i = 0; i < n_points; i\+\+\) \{", "block:interpolate_with_mvc:0"),
    # TetMesh::build_tet_grid: the per-tet marking loop of the FIRST pass (deformed mesh, fills up_ids) and of the THIRD
    # pass (canonical mesh, sets original_bitfield): the bodies of the two std::async lambdas
    ("src/editing/datastructures/tet_mesh.cu", "build_tet_grid_mark_deformed", r"^\t\t\tfor \(int i = beginn; i < endingg; i\+\+\) \{", "block:build_tet_grid:0"),
    ("src/editing/datastructures/tet_mesh.cu", "build_tet_grid_mark_canonical", r"^\t\t\tfor \(int i = beginn; i < endingg; i\+\+\) \{", "block:build_tet_grid:1"),
    # oracle/ref_json.cpp: the to_json members of the edit operators and Testbed::save_edits (the on-disk edits format)
    ("src/editing/tools/region_growing.cu", "region_growing_to_json", r"^nlohmann::json RegionGrowing::to_json\(\) \{", r"until:^\s+\}\s*$"),
    ("src/editing/tools/growing_selection.cu", "growing_selection_to_json", r"^void GrowingSelection::to_json\(nlohmann::json& j\) \{", "fn"),
    ("src/editing/cage_deformation.cu", "cage_deformation_to_json", r"^nlohmann::json CageDeformation::to_json\(\) \{", "fn"),
    ("src/editing/affine_duplication.cu", "affine_duplication_to_json", r"^nlohmann::json AffineDuplication::to_json\(\) \{", "fn"),
    ("src/testbed.cu", "testbed_save_edits", r"^void Testbed::save_edits\(const std::string& filepath_string\) \{", "fn"),
    ("src/testbed.cu", "merge_parent_network_config", r"^json merge_parent_network_config\(const json& child, const fs::path& child_filename\) \{", "fn"),
    ("src/testbed.cu", "testbed_save_snapshot", r"^void Testbed::save_snapshot\(const std::string& filepath_string, bool include_optimizer_state\) \{", "fn"),
]


def cut(lines, first_re, how):
    rx = re.compile(first_re)
    if how == "fn":
        starts = [i for i, l in enumerate(lines) if rx.search(l)]
        if len(starts) != 1:
            raise SystemExit(f"ref_extract: {first_re!r} matched {len(starts)} lines, expected 1")
        s = starts[0]
        e = next(i for i in range(s, len(lines)) if lines[i].rstrip("\r\n") == "}")
        return s, e
    if how.startswith("until:"):
        s = next(i for i, l in enumerate(lines) if rx.search(l))
        end = re.compile(how[len("until:"):])
        e = next(i for i in range(s, len(lines)) if end.search(lines[i]))
        return s, e
    if how.startswith("before:"):
        s = next(i for i, l in enumerate(lines) if rx.search(l))
        end = re.compile(how[len("before:"):])
        e = next(i for i in range(s, len(lines)) if end.search(lines[i]))
        return s, e - 1
    _, owner, nth = how.split(":")
    owner_line = 0 if not owner else next(i for i, l in enumerate(lines) if re.search(r"::" + owner + r"\(", l) and not l.lstrip().startswith("//"))
    starts = [i for i in range(owner_line, len(lines)) if rx.search(lines[i])]
    s = starts[int(nth)]
    depth = 0
    for i in range(s, len(lines)):
        code = lines[i].split("//")[0]
        depth += code.count("{") - code.count("}")
        if depth == 0:
            return s, i
    raise SystemExit("ref_extract: unbalanced block")


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    cache = {}
    for rel, name, first_re, how in FRAGMENTS:
        path = os.path.join(ref, rel)
        if path not in cache:
            with open(path, encoding="utf-8", errors="replace") as f:
                cache[path] = f.readlines()
        lines = cache[path]
        s, e = cut(lines, first_re, how)
        with open(os.path.join(out, name + ".inc"), "w") as f:
            f.write(f'#line {s + 1} "{path}"\n')
            f.writelines(lines[s:e + 1])
        print(f"  {rel}:{s + 1}-{e + 1} -> {name}.inc")


if __name__ == "__main__":
    main()
