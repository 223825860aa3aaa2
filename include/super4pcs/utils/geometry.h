// Drop-in for the part of src/super4pcs/utils/geometry.h the CLI uses: Utils::CleanInvalidNormals (:56-82).
#ifndef S4P_FACADE_GEOMETRY_H_
#define S4P_FACADE_GEOMETRY_H_
#include <iostream>

namespace GlobalRegistration {
namespace Utils {

// Point sets only (no faces): a vertex whose normal has squared length < 0.01 loses it (zero vector in both
// containers); every other normal is normalised in both containers.
template <typename PointContainer, typename VecContainer>
static inline void CleanInvalidNormals(PointContainer& v, VecContainer& normals) {
  using Vector = typename VecContainer::value_type;
  if (v.size() != normals.size()) return;
  auto itV = v.begin();
  auto itN = normals.begin();
  unsigned int nb = 0;
  for (; itV != v.end(); itV++, itN++) {
    if ((*itV).normal().squaredNorm() < 0.01) {
      Vector zero;
      zero(0) = 0.f; zero(1) = 0.f; zero(2) = 0.f;
      (*itN) = zero;
      (*itV).set_normal(zero);
      nb++;
    } else {
      Vector& n = *itN;                                   // (*itN).normalize()
      const float z = n(0) * n(0) + (n(1) * n(1) + n(2) * n(2));
      if (z > 0.f) { const float s = std::sqrt(z); n(0) = n(0) / s; n(1) = n(1) / s; n(2) = n(2) / s; }
      (*itV).normalize();
    }
  }
  if (nb != 0) std::cout << "Found " << nb << " vertices with invalid normals" << std::endl;
}

}  // namespace Utils
}  // namespace GlobalRegistration
#endif
