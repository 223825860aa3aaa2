// External application against the installed package, the check the reference pins with tests/externalAppTest/main.cpp:
// the three public headers, IOManager / Utils symbols, a MatchSuper4PCS built from options + logger, ComputeTransformation
// with explicit sampler / visitor types on empty clouds (returns kLargeNumber, match4pcsBase.hpp:69-70), WriteMatrix.
// Without a gfx950 device the matcher's constructor throws (no CPU fallback): reported and counted as success of the
// PACKAGING check, which is what this program is for.
#include <cstdio>
#include <exception>
#include <string>
#include <vector>

#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/io/io.h"
#include "super4pcs/utils/geometry.h"

int main() {
  using namespace GlobalRegistration;
  std::vector<Point3D> cloud_a, cloud_b;
  std::vector<IOManager::TexCoord> tex_a;
  std::vector<Point3D::VectorType> normals_a;
  std::vector<tripple> faces_a;
  std::vector<std::string> materials_a;

  IOManager io;
  const bool read_ok = io.ReadObject("", cloud_a, tex_a, normals_a, faces_a, materials_a);   // no such file: false, nothing thrown
  if (faces_a.empty()) Utils::CleanInvalidNormals(cloud_a, normals_a);

  Match4PCSOptions options;
  options.configureOverlap(1.0);
  Match4PCSBase::MatrixType mat;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) mat(r, c) = r == c ? 1.f : 0.f;
  Point3D::Scalar score = 0;
  constexpr Utils::LogLevel level = Utils::Verbose;
  Utils::Logger logger(level);
  using Visitor = Match4PCSBase::DummyTransformVisitor;
  using Sampler = Match4PCSBase::DefaultSampler;
  try {
    MatchSuper4PCS matcher(options, logger);
    score = matcher.ComputeTransformation<Sampler, Visitor>(cloud_a, &cloud_b, mat);
    std::printf("external_app: read=%d score=%g (kLargeNumber=%g)\n", int(read_ok), double(score), double(Match4PCSBase::kLargeNumber));
    if (score != Match4PCSBase::kLargeNumber) return 2;
  } catch (const std::exception& e) {
    std::printf("external_app: matcher refused to start: %s\n", e.what());
  }
  const bool wrote = io.WriteMatrix("external_app_output.map", mat.cast<double>(), IOManager::POLYWORKS);
  std::printf("external_app: matrix written=%d\n", int(wrote));
  return wrote ? 0 : 3;
}
