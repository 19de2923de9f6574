// ref_mvc.cpp -- thin driver around the REFERENCE's own mean-value-coordinate code, compiled from where it lies:
//   /root/reference/include/neural-graphics-primitives/editing/tools/mvc.h   (MVC3D::computeCoordinatesCustomCode, :125-188)
// mvc.h is self-contained (<vector>, <cmath>, <cassert>); it is templated on the point type, which the reference instantiates
// with Eigen::Vector3f (growing_selection.h:90).  Eigen is an un-vendored submodule (dependencies/eigen is empty), so the
// point type below is a 3-float stand-in with the members mvc.h uses (operator-, operator/, norm, cross, dot); its
// summation order (x*x + y*y) + z*z may differ from Eigen's reduction order by an ulp, which is why the pin is a 1e-6
// tolerance on the weights and exact equality on the labels, not bit equality.
// Built by oracle/Makefile into oracle/_ref/libref_mvc.so ONLY where /root/reference exists (this container); the GPU box
// tests against tests/golden/ref_mvc_golden.npz generated from it (tests/golden/make_ref_mvc_golden.py).
// Test infrastructure: nothing of the product links or loads this.
#include <cmath>
#include <cstdint>
#include <vector>

#include <neural-graphics-primitives/editing/tools/mvc.h>

namespace {
struct P3 {
	typedef float type_t;
	float x, y, z;
	P3() : x(0), y(0), z(0) {}
	P3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
	P3 operator-(const P3& o) const { return P3(x - o.x, y - o.y, z - o.z); }
	P3 operator/(float s) const { return P3(x / s, y / s, z / s); }
	float norm() const { return std::sqrt((x * x + y * y) + z * z); }
	P3 cross(const P3& o) const { return P3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
	float dot(const P3& o) const { return (x * o.x + y * o.y) + z * o.z; }
};
} // namespace

extern "C" void ref_mvc_compute(const float* cage_vertices, uint32_t n_cv, const uint32_t* cage_triangles, uint32_t n_tris, const float* points,
                                uint32_t n_points, float* weights_out, uint8_t* labels_out) {
	std::vector<P3> cv(n_cv), normals;
	for (uint32_t i = 0; i < n_cv; ++i) cv[i] = P3(cage_vertices[3 * i], cage_vertices[3 * i + 1], cage_vertices[3 * i + 2]);
	std::vector<uint32_t> tris(cage_triangles, cage_triangles + 3 * (size_t)n_tris);
	std::vector<float> w, ww;
	for (uint32_t p = 0; p < n_points; ++p) { // Cage::compute_mvc, src/editing/datastructures/cage.cu:6-22
		const bool success = MVC3D::computeCoordinatesCustomCode<uint32_t, float, P3>(P3(points[3 * p], points[3 * p + 1], points[3 * p + 2]), tris, cv, normals, w, ww);
		for (uint32_t v = 0; v < n_cv; ++v) weights_out[(size_t)p * n_cv + v] = w[v];
		labels_out[p] = success ? 0 : 1;
	}
}
