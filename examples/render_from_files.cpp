// render_from_files -- a C++ host with no Python and no PyTorch in the process: a snapshot (.ingp / .msgpack) and an edits file (.json) in the reference's
// formats go in, one frame comes out.  It is the call sequence a maintainer of the reference would write around the drop-in:
//
//   Testbed::load_snapshot            src/testbed.cu:3054   -> nrs_snapshot_open + NerfNetwork::set_params / set_density_grid
//   Testbed::load_edits               src/testbed.cu:3205   -> nrs_edits_open + CageDeformation / AffineDuplication
//   Testbed::render_nerf              src/testbed_nerf.cu:3066 -> nrs::compat::Testbed::render_nerf
//
//   usage: render_from_files <snapshot> <edits.json | -> <width> <height> <camera_angle_x radians> <out.raw>
//
// out.raw = float32 RGBA [H][W][4] followed by float32 depth [H][W].  Device memory comes straight from the HIP runtime (hipMalloc); the library takes the
// pointers as they are.  tests/test_gpu_cpp_host.py builds this file, runs it on the GPU and compares out.raw bit for bit with the Python host's frame.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <nrs_compat.hpp>

namespace {

void hip_check(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

struct Snapshot {
	nrs_snapshot* s = nullptr;
	explicit Snapshot(const char* path) { nrs::compat::check(nrs_snapshot_open(path, &s), "nrs_snapshot_open"); }
	~Snapshot() { nrs_snapshot_close(s); }
};

struct Edits {
	nrs_edits* e = nullptr;
	explicit Edits(const char* path) { nrs::compat::check(nrs_edits_open(path, &e), "nrs_edits_open"); }
	~Edits() { nrs_edits_close(e); }
};

struct DeviceBuffer {
	void* p = nullptr;
	size_t bytes;
	explicit DeviceBuffer(size_t n) : bytes(n) {
		hip_check(hipMalloc(&p, n), "hipMalloc");
		hip_check(hipMemset(p, 0, n), "hipMemset");
	}
	~DeviceBuffer() { (void)hipFree(p); }
};

int run(int argc, char** argv) {
	if (argc != 7) {
		std::fprintf(stderr, "usage: %s <snapshot.ingp|.msgpack> <edits.json|-> <width> <height> <camera_angle_x> <out.raw>\n", argv[0]);
		return 2;
	}
	const int width = std::atoi(argv[3]), height = std::atoi(argv[4]);
	const double angle_x = std::atof(argv[5]);
	if (width <= 0 || height <= 0 || !(angle_x > 0)) throw std::runtime_error("width, height and camera_angle_x must be positive");

	// ---- load_snapshot: the network, its parameters, the occupancy grid, the saved camera
	Snapshot snap(argv[1]);
	nrs_model_desc desc;
	uint32_t aabb_scale = 1;
	nrs::compat::check(nrs_snapshot_model_desc(snap.s, &desc, &aabb_scale), "nrs_snapshot_model_desc");
	nrs::compat::Context ctx(0);
	nrs::compat::NerfNetwork network(ctx, desc);
	size_t n_params = 0, n_grid = 0;
	const void* params = nrs_snapshot_params_fp16(snap.s, &n_params);
	network.set_params(params, n_params);
	const float* grid = nrs_snapshot_density_grid(snap.s, &n_grid);
	network.set_density_grid(grid, n_grid);
	float camera[12];
	nrs::compat::check(nrs_snapshot_camera(snap.s, camera), "nrs_snapshot_camera");

	// ---- load_edits: every operator of the file, tables built on the device as the reference's JSON constructors rebuild them
	nrs::compat::Testbed testbed;
	std::vector<std::unique_ptr<nrs::compat::EditOperator>> operators;
	std::unique_ptr<Edits> edits;
	if (std::strcmp(argv[2], "-") != 0) {
		edits.reset(new Edits(argv[2]));
		for (uint32_t i = 0; i < nrs_edits_count(edits->e); ++i) {
			const std::string type = nrs_edits_type(edits->e, i);
			if (type == "cage_deformation") {
				nrs_tet_mesh mesh;
				const float *mvc = nullptr, *cage = nullptr, *cage_rest = nullptr;
				const uint32_t* cage_triangles = nullptr;
				uint32_t n_cage = 0, n_cage_triangles = 0;
				nrs::compat::check(nrs_edits_cage(edits->e, i, &mesh, &mvc, &cage, &cage_rest, &cage_triangles, &n_cage, &n_cage_triangles), "nrs_edits_cage");
				if (mesh.n_tets == 0) continue; // saved before its cage was tetrahedralised: nothing to apply (growing_selection.cu:2477)
				auto* op = new nrs::compat::CageDeformation(ctx, desc, mesh);
				operators.emplace_back(op);
				if (mvc) op->set_mvc(mvc, n_cage); // ready for update_cage() when the host moves the cage
			} else if (type == "affine_duplication") {
				nrs_affine_duplication a;
				nrs::compat::check(nrs_edits_affine(edits->e, i, &a), "nrs_edits_affine");
				operators.emplace_back(new nrs::compat::AffineDuplication(ctx, desc, a));
			} else {
				throw std::runtime_error("edit operator '" + type + "' is outside the render path's scope");
			}
			testbed.m_edit_operators.push_back(operators.back().get());
		}
	}

	// ---- the Testbed members render_nerf reads
	testbed.m_nerf.cone_angle_constant = aabb_scale <= 1 ? 0.f : 1.f / 256.f; // testbed_nerf.cu:3410-3425
	const float half = 0.5f * (float)aabb_scale;                              // m_render_aabb = the dataset's box: centred on 0.5, aabb_scale wide
	for (int i = 0; i < 3; ++i) {
		testbed.m_render_aabb_min[i] = 0.5f - half;
		testbed.m_render_aabb_max[i] = 0.5f + half;
	}
	const float focal = (float)(0.5 * width / std::tan(0.5 * angle_x));
	const float focal_length[2] = {focal, focal}, screen_center[2] = {0.5f, 0.5f}, rolling_shutter[4] = {0, 0, 0, 0};
	const int max_res[2] = {width, height};

	// ---- render_nerf into buffers this program owns
	const size_t n_pixels = (size_t)width * height;
	DeviceBuffer frame(n_pixels * 4 * sizeof(float)), depth(n_pixels * sizeof(float));
	nrs::compat::RenderBuffer buffer{(float*)frame.p, (float*)depth.p, width, height, 0};
	nrs_render_stats stats{};
	testbed.render_nerf(network, buffer, max_res, focal_length, camera, camera, rolling_shutter, screen_center, true, nullptr, &stats);

	std::vector<float> host(n_pixels * 5);
	hip_check(hipMemcpy(host.data(), frame.p, frame.bytes, hipMemcpyDeviceToHost), "hipMemcpy(frame)");
	hip_check(hipMemcpy(host.data() + n_pixels * 4, depth.p, depth.bytes, hipMemcpyDeviceToHost), "hipMemcpy(depth)");
	FILE* f = std::fopen(argv[6], "wb");
	if (!f || std::fwrite(host.data(), sizeof(float), host.size(), f) != host.size()) throw std::runtime_error(std::string("cannot write ") + argv[6]);
	std::fclose(f);
	std::printf("{\"width\": %d, \"height\": %d, \"operators\": %zu, \"n_samples\": %llu, \"n_rays_alive\": %u, \"n_rays_hit\": %u}\n", width, height,
	            operators.size(), (unsigned long long)stats.n_samples, stats.n_rays_alive, stats.n_rays_hit);
	return 0;
}

} // namespace

int main(int argc, char** argv) {
	try {
		return run(argc, argv);
	} catch (const std::exception& e) {
		std::fprintf(stderr, "render_from_files: %s\n", e.what());
		return 1;
	}
}
