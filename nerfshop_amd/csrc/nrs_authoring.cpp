// nrs_authoring.cpp -- host-side edit authoring behind include/nrs.h: cell->tet LUT builder, MVC weights, per-tet
// rotations.  SURVEY 8(f) row 1: the steps that run on the CPU in the reference too, immediately BEFORE the render path
// on every gizmo move.  They are here so a cage edit can be produced without the reference's GUI; a device version of
// the LUT builder runs in nrs_cage.hip.  No HIP calls: usable without a GPU.
//
//   nrs_tet_lut_build        TetMesh::build_tet_grid / build_original_tet_grid   src/editing/datastructures/tet_mesh.cu:368 / :76
//   nrs_mvc_compute / apply  Cage::compute_mvc / interpolate_with_mvc            src/editing/datastructures/cage.cu:6 / :38,
//                            MVC3D::computeCoordinatesCustomCode                 include/.../editing/tools/mvc.h:125-188
//   nrs_tet_local_rotations  TetMesh::update_local_rotations (svd3.h numerics)    tet_mesh.cu:37-74, nrs_svd3.h
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "nrs_internal.h"
#include "nrs_svd3.h"

using namespace nrs;

namespace {

struct P3 { float x, y, z; };
inline P3 sub(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 add(P3 a, P3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 mul(P3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// Eigen's 3-term reduction order (Eigen/src/Core/Redux.h, complete unrolling: x0 + (x1 + x2)); see nrs_device.cuh
inline float dotp(P3 a, P3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline P3 crossp(P3 a, P3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float at(const P3& p, int i) { return i == 0 ? p.x : (i == 1 ? p.y : p.z); }

inline uint32_t spread3(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}
inline uint32_t morton(uint32_t x, uint32_t y, uint32_t z) { return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2); }

inline bool same_side(P3 v1, P3 v2, P3 v3, P3 v4, P3 p) { // selection_utils.h:33-39
	P3 n = crossp(sub(v2, v1), sub(v3, v1));
	return std::signbit(dotp(n, sub(v4, v1))) == std::signbit(dotp(n, sub(p, v1)));
}
inline bool in_tet(const P3 t[4], P3 p) { // selection_utils.h:41-47
	return same_side(t[0], t[1], t[2], t[3], p) && same_side(t[1], t[2], t[3], t[0], p) && same_side(t[2], t[3], t[0], t[1], p) &&
	       same_side(t[3], t[0], t[1], t[2], p);
}

inline void span(const P3* pts, int n, P3 axis, float& lo, float& hi) {
	lo = std::numeric_limits<float>::infinity();
	hi = -lo;
	for (int i = 0; i < n; ++i) {
		float v = dotp(axis, pts[i]);
		if (v < lo) lo = v;
		if (v > hi) hi = v;
	}
}
// BoundingBox::intersects(Triangle), bounding_box.cuh:126-178 (separating axes: 3 box normals, the triangle normal,
// 9 edge x axis products)
bool cube_hits_triangle(P3 bmin, P3 bmax, P3 a, P3 b, P3 c) {
	const P3 axes[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	P3 tri[3] = {a, b, c};
	float tlo, thi, blo, bhi;
	for (int i = 0; i < 3; ++i) {
		span(tri, 3, axes[i], tlo, thi);
		if (thi < at(bmin, i) || tlo > at(bmax, i)) return false;
	}
	P3 n = crossp(sub(b, a), sub(c, a));
	const float len = std::sqrt(dotp(n, n));
	n = {n.x / len, n.y / len, n.z / len};
	const P3 corners[8] = {{bmin.x, bmin.y, bmin.z}, {bmin.x, bmin.y, bmax.z}, {bmin.x, bmax.y, bmin.z}, {bmin.x, bmax.y, bmax.z},
	                       {bmax.x, bmin.y, bmin.z}, {bmax.x, bmin.y, bmax.z}, {bmax.x, bmax.y, bmin.z}, {bmax.x, bmax.y, bmax.z}};
	const float off = dotp(n, a);
	span(corners, 8, n, blo, bhi);
	if (bhi < off || blo > off) return false;
	const P3 edges[3] = {sub(a, b), sub(a, c), sub(b, c)};
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			P3 ax = crossp(edges[i], axes[j]);
			span(corners, 8, ax, blo, bhi);
			span(tri, 3, ax, tlo, thi);
			if (bhi < tlo || blo > thi) return false;
		}
	return true;
}

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline void cell_of(P3 p, uint32_t level, int out[3]) { // get_cell_at_pos, selection_utils.cu:70-83
	const float s = std::scalbn(1.0f, -(int)level);
	p = sub(p, {0.5f, 0.5f, 0.5f});
	p = mul(p, s);
	p = add(p, {0.5f, 0.5f, 0.5f});
	out[0] = clampi((int)(p.x * (float)kGrid), 0, kGrid - 1);
	out[1] = clampi((int)(p.y * (float)kGrid), 0, kGrid - 1);
	out[2] = clampi((int)(p.z * (float)kGrid), 0, kGrid - 1);
}
inline P3 cell_centre(uint32_t x, uint32_t y, uint32_t z, uint32_t level) { // get_cell_pos, selection_utils.cu:65-68
	const float s = std::scalbn(1.0f, (int)level);
	return {(((float)x + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f, (((float)y + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f,
	        (((float)z + 0.5f) / (float)kGrid - 0.5f) * s + 0.5f};
}
const P3 kCorners[8] = {{-0.5f, -0.5f, -0.5f}, {-0.5f, -0.5f, 0.5f}, {-0.5f, 0.5f, -0.5f}, {0.5f, -0.5f, -0.5f},
                        {0.5f, 0.5f, -0.5f}, {-0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, 0.5f}}; // tet_mesh.h:34-43

struct Mark { uint32_t cell, tet; };

// first pass of build_tet_grid for tets [t0, t1): which (cell, tet) pairs exist
void mark_range(const P3* verts, const uint32_t* tets, uint32_t t0, uint32_t t1, std::vector<Mark>& marks) {
	for (uint32_t i = t0; i < t1; ++i) {
		const P3 tv[4] = {verts[tets[4 * i]], verts[tets[4 * i + 1]], verts[tets[4 * i + 2]], verts[tets[4 * i + 3]]};
		const float inf = std::numeric_limits<float>::infinity();
		P3 lo = {inf, inf, inf}, hi = {-inf, -inf, -inf};
		for (int j = 0; j < 4; ++j) {
			lo = {std::fmin(lo.x, tv[j].x), std::fmin(lo.y, tv[j].y), std::fmin(lo.z, tv[j].z)};
			hi = {std::fmax(hi.x, tv[j].x), std::fmax(hi.y, tv[j].y), std::fmax(hi.z, tv[j].z)};
		}
		for (uint32_t level = 0; level < kCascades; ++level) {
			const float cell = std::scalbn(1.0f, (int)level) * (1.0f / (float)kGrid); // cell edge at this cascade
			int c0[3], c1[3];
			cell_of(lo, level, c0);
			cell_of(hi, level, c1);
			for (int x = c0[0]; x <= c1[0]; ++x)
				for (int y = c0[1]; y <= c1[1]; ++y)
					for (int z = c0[2]; z <= c1[2]; ++z) {
						const P3 ctr = cell_centre((uint32_t)x, (uint32_t)y, (uint32_t)z, level);
						bool hit = false;
						for (int k = 0; k < 8 && !hit; ++k) hit = in_tet(tv, add(ctr, mul(kCorners[k], cell)));
						if (!hit) {
							const P3 h = mul({0.5f, 0.5f, 0.5f}, cell);
							const P3 a = sub(ctr, h), b = add(ctr, h);
							const P3 bmin = {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
							const P3 bmax = {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
							for (int j = 0; j < 4 && !hit; ++j) hit = cube_hits_triangle(bmin, bmax, tv[j], tv[(j + 1) % 4], tv[(j + 2) % 4]);
						}
						if (hit) marks.push_back({level * kGridVol + morton((uint32_t)x, (uint32_t)y, (uint32_t)z), i});
					}
		}
	}
}

} // namespace

struct nrs_tet_lut {
	std::vector<uint32_t> offsets, idx;
	std::vector<uint8_t> bitfield;
	uint32_t max_per_cell = 0;
};

static thread_local std::string g_auth_err;

// hardware threads this process may really use: hardware_concurrency() cut to the cgroup's CPU quota (a container that shows 256 CPUs under a 16-core
// quota is throttled when 256 busy threads start)
static uint32_t usable_threads() {
	uint32_t n = std::max(1u, std::thread::hardware_concurrency());
	double quota = -1.0, period = 100000.0;
	if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[64] = {0};
		if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
		fclose(f);
	} else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
		if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
		fclose(g);
		if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &period) != 1) period = 100000.0; fclose(h); }
	}
	if (quota > 0.0 && period > 0.0) n = std::min<uint32_t>(n, (uint32_t)std::max(1.0, std::ceil(quota / period)));
	return n;
}

extern "C" {

int nrs_tet_lut_build(const float* h_vertices, uint32_t n_vertices, const uint32_t* h_tets, uint32_t n_tets, int n_threads, nrs_tet_lut** out) {
	if (!h_vertices || !h_tets || !out || n_tets == 0) return NRS_ERR_INVALID_ARG;
	for (size_t i = 0; i < 4 * (size_t)n_tets; ++i)
		if (h_tets[i] >= n_vertices) return NRS_ERR_INVALID_ARG;
	nrs_tet_lut* lut = new (std::nothrow) nrs_tet_lut();
	if (!lut) return NRS_ERR_STATE;
	const P3* verts = reinterpret_cast<const P3*>(h_vertices);
	uint32_t nt = n_threads > 0 ? (uint32_t)n_threads : usable_threads();
	nt = std::min(nt, n_tets);
	// contiguous tet ranges per thread, merged in thread order: within a cell the LUT lists tets in ascending index
	// (the order the reference's thread-ordered second pass produces, tet_mesh.cu:496-512)
	std::vector<std::vector<Mark>> marks(nt);
	std::vector<std::thread> pool;
	const uint32_t chunk = n_tets / nt;
	for (uint32_t t = 0; t < nt; ++t) {
		const uint32_t t0 = chunk * t, t1 = (t == nt - 1) ? n_tets : chunk * (t + 1);
		pool.emplace_back([&, t, t0, t1]() { mark_range(verts, h_tets, t0, t1, marks[t]); });
	}
	for (auto& th : pool) th.join();

	const uint32_t n_cells = kGridVol * kCascades;
	lut->offsets.assign((size_t)n_cells + 1, 0);
	lut->bitfield.assign(n_cells / 8, 0);
	size_t total = 0;
	for (auto& v : marks) {
		total += v.size();
		for (const Mark& mk : v) lut->offsets[(size_t)mk.cell + 1]++;
	}
	for (uint32_t c = 0; c < n_cells; ++c) {
		const uint32_t n_in_cell = lut->offsets[(size_t)c + 1];
		lut->max_per_cell = std::max(lut->max_per_cell, n_in_cell);
		if (n_in_cell) lut->bitfield[c / 8] |= (uint8_t)(1u << (c % 8)); // byte (level*128^3 + morton)/8 == morton/8 + level*128^3/8
		lut->offsets[(size_t)c + 1] = lut->offsets[c] + n_in_cell;
	}
	lut->idx.assign(total, 0);
	std::vector<uint32_t> cursor(lut->offsets.begin(), lut->offsets.end() - 1);
	for (auto& v : marks)
		for (const Mark& mk : v) lut->idx[cursor[mk.cell]++] = mk.tet;
	*out = lut;
	return NRS_OK;
}
uint32_t nrs_tet_lut_n_idx(const nrs_tet_lut* l) { return l ? (uint32_t)l->idx.size() : 0; }
uint32_t nrs_tet_lut_max_per_cell(const nrs_tet_lut* l) { return l ? l->max_per_cell : 0; }
const uint32_t* nrs_tet_lut_offsets(const nrs_tet_lut* l) { return l ? l->offsets.data() : nullptr; }
const uint32_t* nrs_tet_lut_idx(const nrs_tet_lut* l) { return l ? l->idx.data() : nullptr; }
const uint8_t* nrs_tet_lut_bitfield(const nrs_tet_lut* l) { return l ? l->bitfield.data() : nullptr; }
void nrs_tet_lut_destroy(nrs_tet_lut* l) { delete l; }

// mean value coordinates, Ju/Schaefer/Warren 2005 as coded in mvc.h:125-188 (float_t = float, growing_selection.h:89)
int nrs_mvc_compute(const float* h_cage_vertices, uint32_t n_cv, const uint32_t* h_cage_triangles, uint32_t n_tris, const float* h_points,
                    uint32_t n_points, float* h_weights_out, uint8_t* h_labels_out) {
	if (!h_cage_vertices || !h_cage_triangles || !h_points || !h_weights_out || n_cv == 0) return NRS_ERR_INVALID_ARG;
	for (size_t i = 0; i < 3 * (size_t)n_tris; ++i)
		if (h_cage_triangles[i] >= n_cv) return NRS_ERR_INVALID_ARG;
	const P3* cv = reinterpret_cast<const P3*>(h_cage_vertices);
	const float eps = 0.00000001f;
	std::vector<float> dist(n_cv), acc(n_cv);
	std::vector<P3> unit(n_cv);
	for (uint32_t pi = 0; pi < n_points; ++pi) {
		const P3 eta = {h_points[3 * pi], h_points[3 * pi + 1], h_points[3 * pi + 2]};
		float* w_out = h_weights_out + (size_t)pi * n_cv;
		std::fill(w_out, w_out + n_cv, 0.f);
		bool early = false;
		for (uint32_t v = 0; v < n_cv && !early; ++v) {
			const P3 e = sub(eta, cv[v]);
			dist[v] = std::sqrt(dotp(e, e));
			if (dist[v] < eps) { w_out[v] = 1.0f; early = true; break; }
			const P3 q = sub(cv[v], eta);
			unit[v] = {q.x / dist[v], q.y / dist[v], q.z / dist[v]};
		}
		if (!early) {
			std::fill(acc.begin(), acc.end(), 0.f);
			float sum = 0.f;
			for (uint32_t t = 0; t < n_tris && !early; ++t) {
				const uint32_t id[3] = {h_cage_triangles[3 * t], h_cage_triangles[3 * t + 1], h_cage_triangles[3 * t + 2]};
				float len[3], theta[3], w[3], c[3], s[3];
				for (int i = 0; i < 3; ++i) {
					const P3 q = sub(unit[id[(i + 1) % 3]], unit[id[(i + 2) % 3]]);
					len[i] = std::sqrt(dotp(q, q));
					theta[i] = (float)(2.0 * std::asin((double)len[i] / 2.0));
				}
				// Overloads as the reference's translation units resolve them: unqualified sin(float) / fabs(float) are the FLOAT functions
				// (<math.h> and CUDA's global float overloads are in scope), asin(l / 2.0) and sqrt(std::max<double>(..)) take doubles;
				// pinned by mvc.h compiled with Eigen::Vector3f points (tests/golden/ref_mvc_golden.npz).
				const float h = (float)((double)((theta[0] + theta[1]) + theta[2]) / 2.0);
				if (M_PI - (double)h < (double)eps) { // eta lies on the triangle: 2-D barycentric
					for (int i = 0; i < 3; ++i) w[i] = (sinf(theta[i]) * len[(i + 2) % 3]) * len[(i + 1) % 3];
					const float sw = (w[0] + w[1]) + w[2];
					std::fill(w_out, w_out + n_cv, 0.f);
					for (int i = 0; i < 3; ++i) w_out[id[i]] = w[i] / sw;
					early = true;
					break;
				}
				for (int i = 0; i < 3; ++i)
					c[i] = (float)((((2.0 * (double)sinf(h)) * (double)sinf(h - theta[i])) / (double)(sinf(theta[(i + 1) % 3]) * sinf(theta[(i + 2) % 3]))) - 1.0);
				const float sgn = ((double)dotp(crossp(unit[id[0]], unit[id[1]]), unit[id[2]]) < 0.0) ? -1.f : 1.f;
				for (int i = 0; i < 3; ++i) s[i] = (float)((double)sgn * std::sqrt(std::max(0.0, 1.0 - (double)(c[i] * c[i]))));
				if (fabsf(s[0]) < eps || fabsf(s[1]) < eps || fabsf(s[2]) < eps) continue; // coplanar, outside the triangle
				for (int i = 0; i < 3; ++i)
					w[i] = (float)(((double)((theta[i] - c[(i + 1) % 3] * theta[(i + 2) % 3]) - c[(i + 2) % 3] * theta[(i + 1) % 3])) /
					               (((2.0 * (double)dist[id[i]]) * (double)sinf(theta[(i + 1) % 3])) * (double)s[(i + 2) % 3]));
				sum += (w[0] + w[1] + w[2]);
				acc[id[0]] += w[0]; acc[id[1]] += w[1]; acc[id[2]] += w[2];
			}
			if (!early)
				for (uint32_t v = 0; v < n_cv; ++v) w_out[v] = acc[v] / sum;
		}
		// Cage::compute_mvc sets labels[i] = 1 when the routine returns false, i.e. on the regular path (cage.cu:19-21)
		if (h_labels_out) h_labels_out[pi] = early ? 0 : 1;
	}
	return NRS_OK;
}

int nrs_mvc_apply(const float* h_weights, const float* h_cage_vertices, uint32_t n_cv, uint32_t n_points, float* h_points_out) {
	if (!h_weights || !h_cage_vertices || !h_points_out) return NRS_ERR_INVALID_ARG;
	for (uint32_t i = 0; i < n_points; ++i) { // cage.cu:38-49: points[i] += weights[i][v] * vertices[v], v ascending
		P3 p = {0.f, 0.f, 0.f};
		for (uint32_t v = 0; v < n_cv; ++v) {
			const float w = h_weights[(size_t)i * n_cv + v];
			p = add(p, {w * h_cage_vertices[3 * v], w * h_cage_vertices[3 * v + 1], w * h_cage_vertices[3 * v + 2]});
		}
		h_points_out[3 * i] = p.x; h_points_out[3 * i + 1] = p.y; h_points_out[3 * i + 2] = p.z;
	}
	return NRS_OK;
}

// TetMesh::update_local_rotations (tet_mesh.cu:37-74) with the reference's approximate SVD, see nrs_svd3.h
int nrs_tet_local_rotations(const float* h_vertices, const float* h_original_vertices, const uint32_t* h_tets, uint32_t n_tets, float* h_out) {
	if (!h_vertices || !h_original_vertices || !h_tets || !h_out) return NRS_ERR_INVALID_ARG;
	for (uint32_t i = 0; i < n_tets; ++i) {
		float o[4][3], d[4][3];
		for (int j = 0; j < 4; ++j)
			for (int k = 0; k < 3; ++k) { o[j][k] = h_original_vertices[3 * h_tets[4 * i + j] + k]; d[j][k] = h_vertices[3 * h_tets[4 * i + j] + k]; }
		svd3::tet_rotation(o, d, h_out + 9 * (size_t)i);
	}
	return NRS_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Selection bookkeeping after nrs_project_selection_pixels (host; growing_selection.cu:1964-2021, selection_utils.cu:36-48)
static uint32_t morton_part(uint32_t x) { // tcnn morton3D's expand_bits
	x = (x * 0x00010001u) & 0xFF0000FFu;
	x = (x * 0x00000101u) & 0x0F00F00Fu;
	x = (x * 0x00000011u) & 0xC30C30C3u;
	x = (x * 0x00000005u) & 0x49249249u;
	return x;
}
static uint32_t morton_compact(uint32_t x) { // tcnn morton3D_invert
	x = x & 0x49249249u;
	x = (x | (x >> 2)) & 0xc30c30c3u;
	x = (x | (x >> 4)) & 0x0f00f00fu;
	x = (x | (x >> 8)) & 0xff0000ffu;
	x = (x | (x >> 16)) & 0x0000ffffu;
	return x;
}
extern "C" uint32_t nrs_upper_cell_idx(uint32_t cell_idx, uint32_t target_level) {
	const uint32_t vol = 128u * 128u * 128u;
	const uint32_t level = cell_idx / vol, pos = cell_idx % vol;
	uint32_t x = morton_compact(pos), y = morton_compact(pos >> 1), z = morton_compact(pos >> 2);
	for (uint32_t i = level; i < target_level; ++i) { x = x / 2 + 32; y = y / 2 + 32; z = z / 2 + 32; }
	return target_level * vol + (morton_part(x) | (morton_part(y) << 1) | (morton_part(z) << 2));
}
extern "C" int nrs_selection_cells(const float* h_positions, const uint32_t* h_cells, const uint8_t* h_found, uint32_t n, int automatic_max_level,
                                   uint32_t* growing_level, uint32_t* out_cells, float* out_positions, uint32_t* n_out) {
	if (!growing_level || !n_out || (n && (!h_positions || !h_cells || !h_found || !out_cells || !out_positions))) return NRS_ERR_INVALID_ARG;
	const uint32_t vol = 128u * 128u * 128u;
	if (automatic_max_level) { // :1964-1976
		*growing_level = 0;
		for (uint32_t i = 0; i < n; ++i)
			if (h_found[i] && h_cells[i] / vol > *growing_level) *growing_level = h_cells[i] / vol;
	}
	std::unordered_set<uint32_t> seen;
	uint32_t k = 0;
	for (uint32_t i = 0; i < n; ++i) { // :1983-2021, in pixel order (the reference iterates in the order its atomics compacted the rays)
		if (!h_found[i]) continue;
		uint32_t cell = h_cells[i];
		const uint32_t level = cell / vol;
		if (level > *growing_level) continue;
		if (level < *growing_level) cell = nrs_upper_cell_idx(cell, *growing_level);
		if (!seen.insert(cell).second) continue;
		out_cells[k] = cell;
		for (int c = 0; c < 3; ++c) out_positions[3 * k + c] = h_positions[3 * i + c];
		++k;
	}
	*n_out = k;
	return NRS_OK;
}
