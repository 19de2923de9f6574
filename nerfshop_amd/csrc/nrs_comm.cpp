// nrs_comm.cpp -- the one exchange step of the multi-GPU path in host C++ (north_star: "a final RCCL gather over xGMI"):
// every rank renders its image tiles into a compact [frame block | depth block] buffer, nrs_gather_tiles sends it to the root with
// ncclGroupStart / ncclSend x 1 per non-root / ncclRecv x (N - 1) at the root / ncclGroupEnd on the caller's stream (point-to-point over
// xGMI: 5.2 MB per rank for a 1080p frame on 8 GPUs, no ring), and nrs_detile scatters both blocks into the image there.
// The reference is single-GPU (README.md:423-425): no counterpart.
//
// RCCL is loaded with dlopen ("librccl.so.1", then "librccl.so": the copy the process already holds -- PyTorch's -- is reused), so libnrs.so
// has no link-time dependency on it and single-GPU users never touch it.  The communicator is libnrs's own: rank 0 calls
// nrs_comm_unique_id, the application hands the 128 bytes to every rank through whatever channel it has (bench.py: one
// torch.distributed broadcast), every rank calls nrs_comm_create.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "nrs_internal.h"

namespace {

// the slice of rccl.h this file uses (ABI of NCCL 2.x / RCCL: rccl.h:40-43, :187, :220, :260, :339, :466, :700, :722, :923)
struct ncclUniqueId_ { char internal[128]; };
typedef struct ncclComm* ncclComm_t_;
constexpr int kNcclFloat32 = 7;

struct Rccl {
	void* lib = nullptr;
	int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
	int (*CommInitRank)(ncclComm_t_*, int, ncclUniqueId_, int) = nullptr;
	int (*CommDestroy)(ncclComm_t_) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, ncclComm_t_, hipStream_t) = nullptr;
	int (*GetVersion)(int*) = nullptr; // optional
	std::string error, path;
};

Rccl& rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, []() {
		// NRS_RCCL_LIB: load THIS library instead (tests/fake_rccl: several ranks on one GPU, so that the N > 1 legs below run on a one-GPU box).
		// A local load, so that it never shadows the RCCL the process may already hold.
		if (const char* over = getenv("NRS_RCCL_LIB")) {
			r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
			if (!r.lib) { r.error = std::string("NRS_RCCL_LIB: ") + dlerror(); return; }
			r.path = over;
			// (an override of the transport must not be silent -- a stale variable would otherwise only show in nrs_comm_info)
			fprintf(stderr, "[nrs comm] NRS_RCCL_LIB is set: every collective of this process goes through %s instead of RCCL\n", over);
		}
		for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			if (r.lib) break;
			r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.lib) r.path = name;
		}
		if (!r.lib) { r.error = std::string("librccl.so not found: ") + dlerror(); return; }
		auto sym = [&](const char* n) -> void* {
			void* p = dlsym(r.lib, n);
			if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
			return p;
		};
		r.GetUniqueId = (int (*)(ncclUniqueId_*))sym("ncclGetUniqueId");
		r.CommInitRank = (int (*)(ncclComm_t_*, int, ncclUniqueId_, int))sym("ncclCommInitRank");
		r.CommDestroy = (int (*)(ncclComm_t_))sym("ncclCommDestroy");
		r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
		r.GroupStart = (int (*)())sym("ncclGroupStart");
		r.GroupEnd = (int (*)())sym("ncclGroupEnd");
		r.Send = (int (*)(const void*, size_t, int, int, ncclComm_t_, hipStream_t))sym("ncclSend");
		r.Recv = (int (*)(void*, size_t, int, int, ncclComm_t_, hipStream_t))sym("ncclRecv");
		r.GetVersion = (int (*)(int*))dlsym(r.lib, "ncclGetVersion");
	});
	return r;
}

int fail(int code, const std::string& msg) {
	nrs::set_last_error(msg.c_str());
	return code;
}
int ccl_fail(int rc, const char* what) {
	const Rccl& r = rccl();
	return fail(NRS_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

} // namespace

struct nrs_comm {
	ncclComm_t_ comm = nullptr;
	int rank = 0, n_ranks = 1, device = 0;
};

extern "C" {

int nrs_comm_unique_id(uint8_t* out128) {
	if (!out128) return fail(NRS_ERR_INVALID_ARG, "nrs_comm_unique_id: NULL argument");
	Rccl& r = rccl();
	if (!r.error.empty()) return fail(NRS_ERR_UNSUPPORTED, "nrs_comm_unique_id: " + r.error);
	ncclUniqueId_ id;
	const int rc = r.GetUniqueId(&id);
	if (rc != 0) return ccl_fail(rc, "ncclGetUniqueId");
	memcpy(out128, id.internal, 128);
	return NRS_OK;
}

int nrs_comm_create(int device, int rank, int n_ranks, const uint8_t* unique_id128, nrs_comm** out) {
	if (!out || !unique_id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(NRS_ERR_INVALID_ARG, "nrs_comm_create: bad argument");
	Rccl& r = rccl();
	if (!r.error.empty()) return fail(NRS_ERR_UNSUPPORTED, "nrs_comm_create: " + r.error);
	if (hipSetDevice(device) != hipSuccess) return fail(NRS_ERR_HIP, "nrs_comm_create: hipSetDevice failed");
	nrs_comm* c = new (std::nothrow) nrs_comm();
	if (!c) return fail(NRS_ERR_STATE, "out of host memory");
	c->rank = rank; c->n_ranks = n_ranks; c->device = device;
	ncclUniqueId_ id;
	memcpy(id.internal, unique_id128, 128);
	const int rc = r.CommInitRank(&c->comm, n_ranks, id, rank);
	if (rc != 0) { delete c; return ccl_fail(rc, "ncclCommInitRank"); }
	*out = c;
	return NRS_OK;
}

int nrs_comm_info(const nrs_comm* c, int* rank_out, int* n_ranks_out, int* rccl_version_out, char* lib_out, size_t lib_len) {
	if (!c) return fail(NRS_ERR_INVALID_ARG, "nrs_comm_info: NULL communicator");
	Rccl& r = rccl();
	if (rank_out) *rank_out = c->rank;
	if (n_ranks_out) *n_ranks_out = c->n_ranks;
	if (rccl_version_out) { int v = 0; if (!r.GetVersion || r.GetVersion(&v) != 0) v = 0; *rccl_version_out = v; }
	if (lib_out && lib_len) snprintf(lib_out, lib_len, "%s", r.path.c_str());
	return NRS_OK;
}

// Diagnostic: `pairs` point-to-point transfers of n_floats floats from this rank TO ITSELF in one ncclGroupStart / ncclGroupEnd on `stream` -- the enqueue
// path nrs_gather_tiles takes at the root (N - 1 receives) with the sends of the peers folded in, executable with ONE rank: the only way to put RCCL's own
// host-side cost per frame next to a share-frame's render time on a one-GPU box (RCCL refuses two ranks on a device).  tools/gather_probe.py times it.
int nrs_comm_probe_self_p2p(nrs_comm* c, const float* d_src, float* d_dst, size_t n_floats, int pairs, void* stream) {
	if (!c || !d_src || !d_dst || pairs < 1) return fail(NRS_ERR_INVALID_ARG, "nrs_comm_probe_self_p2p: bad argument");
	Rccl& r = rccl();
	if (hipSetDevice(c->device) != hipSuccess) return fail(NRS_ERR_HIP, "nrs_comm_probe_self_p2p: hipSetDevice failed");
	hipStream_t s = (hipStream_t)stream;
	int rc = r.GroupStart();
	if (rc != 0) return ccl_fail(rc, "ncclGroupStart");
	for (int k = 0; k < pairs && rc == 0; ++k) {
		rc = r.Send(d_src + (size_t)k * n_floats, n_floats, kNcclFloat32, c->rank, c->comm, s);
		if (rc == 0) rc = r.Recv(d_dst + (size_t)k * n_floats, n_floats, kNcclFloat32, c->rank, c->comm, s);
	}
	const int rc_end = r.GroupEnd();
	if (rc != 0) return ccl_fail(rc, "ncclSend / ncclRecv");
	if (rc_end != 0) return ccl_fail(rc_end, "ncclGroupEnd");
	return NRS_OK;
}

void nrs_comm_destroy(nrs_comm* c) {
	if (!c) return;
	if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
	delete c;
}

// d_local: this rank's compact tile buffer (local_floats floats).  Root only: d_recv holds n_ranks such buffers, rank-major (the root's own
// contribution is copied device-to-device); d_image / d_depth receive the de-tiled frame (4 channels) and depth (1 channel) when not NULL;
// the buffer layout of a rank is [tiles_padded * tile^2 * 4 floats of frame | tiles_padded * tile^2 floats of depth].
int nrs_gather_tiles(nrs_ctx* ctx, nrs_comm* c, int root, const nrs_render_params* p, uint32_t tiles_per_rank_padded, const float* d_local, float* d_recv,
                     float* d_image, float* d_depth, void* stream) {
	if (!ctx || !c || !p || !d_local) return fail(NRS_ERR_INVALID_ARG, "nrs_gather_tiles: NULL argument");
	if (p->struct_size != (uint32_t)sizeof(nrs_render_params)) return fail(NRS_ERR_INVALID_ARG, "nrs_gather_tiles: nrs_render_params.struct_size does not match this library (NRS_RENDER_PARAMS_INIT)");
	if (root < 0 || root >= c->n_ranks) return fail(NRS_ERR_INVALID_ARG, "nrs_gather_tiles: root out of range");
	if (p->tile_size == 0 || p->tile_size % 8) return fail(NRS_ERR_INVALID_ARG, "nrs_gather_tiles: bad tiling");
	if (c->rank == root && !d_recv) return fail(NRS_ERR_INVALID_ARG, "nrs_gather_tiles: the root needs a receive buffer");
	Rccl& r = rccl();
	if (hipSetDevice(c->device) != hipSuccess) return fail(NRS_ERR_HIP, "nrs_gather_tiles: hipSetDevice failed");
	hipStream_t s = (hipStream_t)stream;
	const size_t n_px = (size_t)tiles_per_rank_padded * p->tile_size * p->tile_size, local_floats = n_px * 5;
	int rc = 0;
	if (c->n_ranks > 1) {
		if ((rc = r.GroupStart()) != 0) return ccl_fail(rc, "ncclGroupStart");
		if (c->rank == root) {
			for (int peer = 0; peer < c->n_ranks && rc == 0; ++peer)
				if (peer != root) rc = r.Recv(d_recv + (size_t)peer * local_floats, local_floats, kNcclFloat32, peer, c->comm, s);
		} else {
			rc = r.Send(d_local, local_floats, kNcclFloat32, root, c->comm, s);
		}
		const int rc_end = r.GroupEnd();
		if (rc != 0) return ccl_fail(rc, "ncclSend / ncclRecv");
		if (rc_end != 0) return ccl_fail(rc_end, "ncclGroupEnd");
	}
	if (c->rank != root) return NRS_OK;
	if (hipMemcpyAsync(d_recv + (size_t)root * local_floats, d_local, local_floats * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
		return fail(NRS_ERR_HIP, "nrs_gather_tiles: copy of the root's own tiles failed");
	if (d_image) {
		const int st = nrs_detile(ctx, stream, p, (uint32_t)c->n_ranks, tiles_per_rank_padded, d_recv, 4, local_floats, d_image);
		if (st != NRS_OK) return st;
	}
	if (d_depth) {
		const int st = nrs_detile(ctx, stream, p, (uint32_t)c->n_ranks, tiles_per_rank_padded, d_recv + n_px * 4, 1, local_floats, d_depth);
		if (st != NRS_OK) return st;
	}
	return NRS_OK;
}

} // extern "C"
