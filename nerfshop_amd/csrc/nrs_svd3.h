// nrs_svd3.h -- the per-tet rotation of TetMesh::update_local_rotations (src/editing/datastructures/tet_mesh.cu:37-74) with the
// reference's OWN numerics: it decomposes the 3x3 correlation matrix with the approximate SVD of McAdams, Selle, Tamstorf,
// Teran, Sifakis (TR1690, 2011) as implemented in include/neural-graphics-primitives/editing/tools/svd3.h -- four cyclic
// Jacobi sweeps with approximate Givens rotations on A^T A, column sort, Givens QR -- all in fp32.  Its R = U V^T deviates
// from the exact polar factor by up to 1.3e-2 per entry on ordinary tets (median 1e-6), so "the same picture as the
// reference" means restating that procedure step by step, not computing a better rotation.  Host and device share this
// header; the reference header itself, compiled from /root/reference by oracle/Makefile into oracle/_ref/libref_render.so (ref_local_rotations in
// oracle/ref_render.cpp), is what both must match bit for bit (tests/test_ref_pin.py authoring cases; its outputs travel in
// tests/golden/ref_pin_golden.npz; -ffp-contract=off, same operation order).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NRS_HD __host__ __device__ inline
#else
#define NRS_HD inline
#endif

namespace nrs {
namespace svd3 {

struct M3 { float a[3][3]; }; // a[row][col]

NRS_HD float rsqrt_exact(float x) { return 1.0f / sqrtf(x); } // what nvcc's host math gives for rsqrt(float)
NRS_HD float rsqrt_fast2(float x) {                          // svd3.h:54-63 (bit trick + two Newton steps)
	const float xhalf = 0.5f * x;
	int32_t i;
	memcpy(&i, &x, 4);
	i = 0x5f37599e - (i >> 1);
	memcpy(&x, &i, 4);
	x = x * (1.5f - xhalf * x * x);
	x = x * (1.5f - xhalf * x * x);
	return x;
}
NRS_HD M3 mul(const M3& A, const M3& B) { // svd3.h:87-102
	M3 M;
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) M.a[r][c] = A.a[r][0] * B.a[0][c] + A.a[r][1] * B.a[1][c] + A.a[r][2] * B.a[2][c];
	return M;
}
NRS_HD M3 mul_at_b(const M3& A, const M3& B) { // svd3.h:105-119
	M3 M;
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) M.a[r][c] = A.a[0][r] * B.a[0][c] + A.a[1][r] * B.a[1][c] + A.a[2][r] * B.a[2][c];
	return M;
}
// one approximate-Givens conjugation of the symmetric matrix held as (s11, s21, s22, s31, s32, s33); q accumulates the
// rotation as a quaternion (x, y, z, w).  (x, y, z) = the cyclic index triple of this step.  svd3.h:147-213
NRS_HD void jacobi_step(int x, int y, int z, float s[6], float q[4]) {
	float ch = 2 * (s[0] - s[2]), sh = s[1];
	const bool small_angle = 5.828427124 * sh * sh < ch * ch; // _gamma is a double literal: the left side is evaluated in double
	const float w = rsqrt_exact(ch * ch + sh * sh);
	ch = small_angle ? w * ch : (float)0.923879532;
	sh = small_angle ? w * sh : (float)0.3826834323;
	const float scale = ch * ch + sh * sh;
	const float a = (ch * ch - sh * sh) / scale, b = (2 * sh * ch) / scale;
	const float t11 = s[0], t21 = s[1], t22 = s[2], t31 = s[3], t32 = s[4], t33 = s[5];
	const float n11 = a * (a * t11 + b * t21) + b * (a * t21 + b * t22);
	const float n21 = a * (-b * t11 + a * t21) + b * (-b * t21 + a * t22);
	const float n22 = -b * (-b * t11 + a * t21) + a * (-b * t21 + a * t22);
	const float n31 = a * t31 + b * t32, n32 = -b * t31 + a * t32, n33 = t33;
	float tmp[3] = {q[0] * sh, q[1] * sh, q[2] * sh};
	sh *= q[3];
	q[0] *= ch; q[1] *= ch; q[2] *= ch; q[3] *= ch;
	q[z] += sh;
	q[3] -= tmp[z];
	q[x] += tmp[y];
	q[y] -= tmp[x];
	// cyclic re-arrangement for the next pivot
	s[0] = n22; s[1] = n32; s[2] = n33; s[3] = n21; s[4] = n31; s[5] = n11;
}
NRS_HD M3 quat_to_mat(const float q[4]) { // svd3.h:121-145
	const float w = q[3], x = q[0], y = q[1], z = q[2];
	const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
	M3 m;
	m.a[0][0] = 1 - 2 * (qyy + qzz); m.a[0][1] = 2 * (qxy - qwz); m.a[0][2] = 2 * (qxz + qwy);
	m.a[1][0] = 2 * (qxy + qwz); m.a[1][1] = 1 - 2 * (qxx + qzz); m.a[1][2] = 2 * (qyz - qwx);
	m.a[2][0] = 2 * (qxz - qwy); m.a[2][1] = 2 * (qyz + qwx); m.a[2][2] = 1 - 2 * (qxx + qyy);
	return m;
}
NRS_HD void neg_swap_cols(bool c, M3& m, int i, int j) { // condNegSwap on a column pair, svd3.h:78-84
	for (int r = 0; r < 3; ++r) {
		const float zneg = -m.a[r][i];
		m.a[r][i] = c ? m.a[r][j] : m.a[r][i];
		m.a[r][j] = c ? zneg : m.a[r][j];
	}
}
NRS_HD void qr_givens(float a1, float a2, float& ch, float& sh) { // svd3.h:270-285
	const float epsilon = (float)1e-6;
	const float rho = (a1 * a1 + a2 * a2) * rsqrt_fast2(a1 * a1 + a2 * a2); // accurateSqrt
	sh = rho > epsilon ? a2 : 0;
	ch = fabsf(a1) + fmaxf(rho, epsilon);
	if (a1 < 0) { const float t = sh; sh = ch; ch = t; }
	const float w = rsqrt_exact(ch * ch + sh * sh);
	ch *= w;
	sh *= w;
}

// U and V of the reference's svd(A) (svd3.h:355-403); S is not needed for the rotation
NRS_HD void svd_uv(const M3& A, M3& U, M3& V) {
	const M3 ata = mul_at_b(A, A);
	float s[6] = {ata.a[0][0], ata.a[1][0], ata.a[1][1], ata.a[2][0], ata.a[2][1], ata.a[2][2]};
	float q[4] = {0, 0, 0, 1};
	for (int sweep = 0; sweep < 4; ++sweep) {
		jacobi_step(0, 1, 2, s, q);
		jacobi_step(1, 2, 0, s, q);
		jacobi_step(2, 0, 1, s, q);
	}
	V = quat_to_mat(q);
	M3 B = mul(A, V);
	// sort the singular values (column norms of B), permuting V alongside: svd3.h:240-267
	float rho[3];
	for (int c = 0; c < 3; ++c) rho[c] = B.a[0][c] * B.a[0][c] + B.a[1][c] * B.a[1][c] + B.a[2][c] * B.a[2][c];
	bool c = rho[0] < rho[1];
	neg_swap_cols(c, B, 0, 1); neg_swap_cols(c, V, 0, 1);
	if (c) { const float t = rho[0]; rho[0] = rho[1]; rho[1] = t; }
	c = rho[0] < rho[2];
	neg_swap_cols(c, B, 0, 2); neg_swap_cols(c, V, 0, 2);
	if (c) { const float t = rho[0]; rho[0] = rho[2]; rho[2] = t; }
	c = rho[1] < rho[2];
	neg_swap_cols(c, B, 1, 2); neg_swap_cols(c, V, 1, 2);
	// QR of B by three Givens rotations; only Q (= U) is kept: svd3.h:288-352
	float ch1, sh1, ch2, sh2, ch3, sh3;
	M3 R;
	qr_givens(B.a[0][0], B.a[1][0], ch1, sh1);
	float a = 1 - 2 * sh1 * sh1, b = 2 * ch1 * sh1;
	for (int k = 0; k < 3; ++k) { R.a[0][k] = a * B.a[0][k] + b * B.a[1][k]; R.a[1][k] = -b * B.a[0][k] + a * B.a[1][k]; R.a[2][k] = B.a[2][k]; }
	qr_givens(R.a[0][0], R.a[2][0], ch2, sh2);
	a = 1 - 2 * sh2 * sh2; b = 2 * ch2 * sh2;
	for (int k = 0; k < 3; ++k) { B.a[0][k] = a * R.a[0][k] + b * R.a[2][k]; B.a[1][k] = R.a[1][k]; B.a[2][k] = -b * R.a[0][k] + a * R.a[2][k]; }
	qr_givens(B.a[1][1], B.a[2][1], ch3, sh3);
	const float sh12 = sh1 * sh1, sh22 = sh2 * sh2, sh32 = sh3 * sh3;
	U.a[0][0] = (-1 + 2 * sh12) * (-1 + 2 * sh22);
	U.a[0][1] = 4 * ch2 * ch3 * (-1 + 2 * sh12) * sh2 * sh3 + 2 * ch1 * sh1 * (-1 + 2 * sh32);
	U.a[0][2] = 4 * ch1 * ch3 * sh1 * sh3 - 2 * ch2 * (-1 + 2 * sh12) * sh2 * (-1 + 2 * sh32);
	U.a[1][0] = 2 * ch1 * sh1 * (1 - 2 * sh22);
	U.a[1][1] = -8 * ch1 * ch2 * ch3 * sh1 * sh2 * sh3 + (-1 + 2 * sh12) * (-1 + 2 * sh32);
	U.a[1][2] = -2 * ch3 * sh3 + 4 * sh1 * (ch3 * sh1 * sh3 + ch1 * ch2 * sh2 * (-1 + 2 * sh32));
	U.a[2][0] = 2 * ch2 * sh2;
	U.a[2][1] = 2 * ch3 * (1 - 2 * sh22) * sh3;
	U.a[2][2] = (-1 + 2 * sh22) * (-1 + 2 * sh32);
}

// R = U V^T of the correlation matrix sum_j (orig_j - c0)(def_j - c1)^T of one tet; out column-major (Eigen::Matrix3f)
NRS_HD void tet_rotation(const float org[4][3], const float def[4][3], float out9[9]) {
	float c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
	for (int j = 0; j < 4; ++j)
		for (int k = 0; k < 3; ++k) { c0[k] += org[j][k]; c1[k] += def[j][k]; }
	for (int k = 0; k < 3; ++k) { c0[k] /= 4.f; c1[k] /= 4.f; }
	M3 A;
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) A.a[r][c] = 0.f;
	for (int j = 0; j < 4; ++j)
		for (int r = 0; r < 3; ++r)
			for (int c = 0; c < 3; ++c) A.a[r][c] += (org[j][r] - c0[r]) * (def[j][c] - c1[c]);
	M3 U, V;
	svd_uv(A, U, V);
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 3; ++c) {
			out9[3 * c + r] = U.a[r][0] * V.a[c][0] + (U.a[r][1] * V.a[c][1] + U.a[r][2] * V.a[c][2]); // Eigen's 3-term reduction: x0 + (x1 + x2)
		}
}

} // namespace svd3
} // namespace nrs
