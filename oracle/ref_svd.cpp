// ref_svd.cpp -- thin driver around the REFERENCE's own 3x3 SVD, compiled from where it lies:
//   /root/reference/include/neural-graphics-primitives/editing/tools/svd3.h   (svd(), McAdams et al. / E. Jang)
// and the loop of TetMesh::update_local_rotations (src/editing/datastructures/tet_mesh.cu:37-74) restated around it:
// centroids, correlation matrix sum (orig - c0)(def - c1)^T, R = U V^T.  <Eigen/Core> comes from oracle/ref_stubs (the
// reference's Eigen submodule is empty); svd() itself touches no Eigen type.  Test infrastructure only.
#include <cmath>
#include <cstdint>

// svd3.h calls rsqrt(), which nvcc's host math headers provide (1 / sqrt in host code); g++ has no such function
inline float rsqrt(float x) { return 1.0f / std::sqrt(x); }

#include <neural-graphics-primitives/editing/tools/svd3.h>

extern "C" void ref_local_rotations(const float* def, const float* org, const uint32_t* tets, uint32_t n_tets, float* out9_colmajor) {
	for (uint32_t i = 0; i < n_tets; ++i) {
		float c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
		for (int j = 0; j < 4; ++j)
			for (int k = 0; k < 3; ++k) { c0[k] += org[3 * tets[4 * i + j] + k]; c1[k] += def[3 * tets[4 * i + j] + k]; }
		for (int k = 0; k < 3; ++k) { c0[k] /= 4.f; c1[k] /= 4.f; }
		float A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
		for (int j = 0; j < 4; ++j)
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c) A[r][c] += (org[3 * tets[4 * i + j] + r] - c0[r]) * (def[3 * tets[4 * i + j] + c] - c1[c]);
		float U[3][3], S[3][3], V[3][3];
		svd(A[0][0], A[0][1], A[0][2], A[1][0], A[1][1], A[1][2], A[2][0], A[2][1], A[2][2],
		    U[0][0], U[0][1], U[0][2], U[1][0], U[1][1], U[1][2], U[2][0], U[2][1], U[2][2],
		    S[0][0], S[0][1], S[0][2], S[1][0], S[1][1], S[1][2], S[2][0], S[2][1], S[2][2],
		    V[0][0], V[0][1], V[0][2], V[1][0], V[1][1], V[1][2], V[2][0], V[2][1], V[2][2]);
		for (int r = 0; r < 3; ++r)
			for (int c = 0; c < 3; ++c) { // R = U V^T, stored column-major like Eigen::Matrix3f
				float s = 0.f;
				for (int k = 0; k < 3; ++k) s += U[r][k] * V[c][k];
				out9_colmajor[9 * (size_t)i + 3 * c + r] = s;
			}
	}
}
