// Drop-in for src/super4pcs/shared4pcs.h: Point3D (:61-111), Quadrilateral (:116-138),
// Match4PCSOptions (:148-190) -- same names, members and defaults.
// With Eigen on the include path the vector types are Eigen's (source compatible with the PCL /
// Meshlab wrappers); without it a minimal 3-float stand-in with the accessors the API needs is used,
// which is what the in-repo tests compile against (Eigen is not installed in the build image).
#ifndef S4P_FACADE_SHARED4PCS_H_
#define S4P_FACADE_SHARED4PCS_H_

#include <array>
#include <cmath>
#include <cstddef>
#include <random>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(S4P_NO_EIGEN)
#include <Eigen/Core>
#define S4P_HAVE_EIGEN 1
#endif
#endif

namespace GlobalRegistration {

#ifndef S4P_HAVE_EIGEN
namespace compat {
struct Vector3f {
  float v[3];
  Vector3f() : v{0.f, 0.f, 0.f} {}
  Vector3f(float x, float y, float z) : v{x, y, z} {}
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
  float& coeffRef(int i) { return v[i]; }
  float coeff(int i) const { return v[i]; }
  float squaredNorm() const { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }
  static Vector3f Zero() { return Vector3f(); }
};
struct Matrix4f {
  float m[16];   // column-major like Eigen's default
  Matrix4f() { for (int i = 0; i < 16; ++i) m[i] = 0.f; }
  float& operator()(int r, int c) { return m[c * 4 + r]; }
  float operator()(int r, int c) const { return m[c * 4 + r]; }
  static Matrix4f Identity() { Matrix4f a; a(0, 0) = a(1, 1) = a(2, 2) = a(3, 3) = 1.f; return a; }
};
}  // namespace compat
#endif

class Point3D {
 public:
  using Scalar = float;
#ifdef S4P_HAVE_EIGEN
  using VectorType = Eigen::Matrix<Scalar, 3, 1>;
#else
  using VectorType = compat::Vector3f;
#endif
  inline Point3D(Scalar x, Scalar y, Scalar z) : pos_(x, y, z) {}
  inline Point3D() {}
  inline VectorType& pos() { return pos_; }
  inline const VectorType& pos() const { return pos_; }
  inline const VectorType& rgb() const { return rgb_; }
  inline const VectorType& normal() const { return normal_; }
  inline void set_rgb(const VectorType& rgb) { rgb_ = rgb; }
  inline void set_normal(const VectorType& n) {            // normal.normalized(), shared4pcs.h:85-87
    normal_ = n;
    normalize();
  }
  inline void normalize() {
    const Scalar z = normal_.squaredNorm();
    if (z > Scalar(0)) { const Scalar s = std::sqrt(z); for (int k = 0; k < 3; ++k) normal_(k) = normal_(k) / s; }
  }
  inline bool hasColor() const { return rgb_.squaredNorm() > Scalar(0.001); }
  Scalar& x() { return pos_.coeffRef(0); }
  Scalar& y() { return pos_.coeffRef(1); }
  Scalar& z() { return pos_.coeffRef(2); }
  Scalar x() const { return pos_.coeff(0); }
  Scalar y() const { return pos_.coeff(1); }
  Scalar z() const { return pos_.coeff(2); }

 private:
  VectorType pos_{0.0f, 0.0f, 0.0f};
  VectorType normal_{0.0f, 0.0f, 0.0f};
  VectorType rgb_{-1.0f, -1.0f, -1.0f};
};

struct Quadrilateral {
  std::array<int, 4> vertices;
  inline Quadrilateral(int v0, int v1, int v2, int v3) { vertices = {v0, v1, v2, v3}; }
  inline bool operator<(const Quadrilateral& rhs) const { return vertices < rhs.vertices; }   // lexicographic, :122-127
  inline bool operator==(const Quadrilateral& rhs) const { return vertices == rhs.vertices; }
  int operator[](int idx) const { return vertices[idx]; }
  int& operator[](int idx) { return vertices[idx]; }
};

struct Match4PCSOptions {
  using Scalar = typename Point3D::Scalar;
  Match4PCSOptions() {}
  Scalar delta = 5.0;
  Scalar max_normal_difference = -1;
  Scalar max_translation_distance = -1;
  Scalar max_angle = -1;
  Scalar max_color_distance = -1;
  size_t sample_size = 200;
  int max_time_seconds = 60;
  unsigned int randomSeed = std::mt19937::default_seed;

  inline bool configureOverlap(Scalar overlap_, Scalar terminate_threshold_ = Scalar(1)) {
    if (terminate_threshold_ < overlap_) return false;
    overlap_estimation = overlap_;
    terminate_threshold = terminate_threshold_;
    return true;
  }
  inline Scalar getTerminateThreshold() const { return terminate_threshold; }
  inline Scalar getOverlapEstimation() const { return overlap_estimation; }

 private:
  Scalar terminate_threshold = 1.0;
  Scalar overlap_estimation = 0.2;
};

}  // namespace GlobalRegistration
#endif
