// Super4PCS, the command-line program, on the MI355X path.  Drop-in for the reference's binary
// (demos/Super4PCS/super4pcs_test.cc; usage in doc/Usage.md and scripts/run-example.sh:68): same flags, same files in
// and out (io/io.h), same exit statuses -- 254 (-2) usage / runtime error, 1 help or unknown flag, 253 (-3) bad
// options, 255 (-1) unreadable input.
//   Super4PCS -i P.obj Q.obj [-o overlap] [-d delta] [-n samples] [-t seconds] [-a normal_deg] [-c colour]
//             [-r registered_geometry] [-m polyworks_matrix] [--sampled1 file] [--sampled2 file]
// -x (the legacy 4PCS matcher, algorithms/4pcs.cc) is outside this library and is refused.
#include <cstdio>
#include <exception>
#include <string>
#include <vector>

#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/io/io.h"
#include "super4pcs/utils/geometry.h"

#include "../cli_options.h"

namespace {

using namespace GlobalRegistration;

struct Mesh {                                    // everything IOManager returns for one file
  std::vector<Point3D> points;
  std::vector<IOManager::TexCoord> tex;
  std::vector<Point3D::VectorType> normals;
  std::vector<tripple> faces;
  std::vector<std::string> materials;

  bool load(IOManager& io, const std::string& path) {
    if (!io.ReadObject(path.c_str(), points, tex, normals, faces, materials)) return false;
    if (faces.empty()) Utils::CleanInvalidNormals(points, normals);   // point sets only: faces index the vertex list
    return true;
  }
  bool save(IOManager& io, const std::string& path) const {
    return io.WriteObject(path.c_str(), points, tex, normals, faces, materials);
  }
};

bool save_points(IOManager& io, const std::string& path, const std::vector<Point3D>& pts) {
  Mesh m;
  m.points = pts;
  return m.save(io, path);
}

// progress line while the matcher runs: one call per trial with the fraction done; per-candidate calls carry -1
struct Progress {
  inline void operator()(float fraction, float best_lcp, Match4PCSBase::MatrixRef) const {
    if (fraction < 0) return;
    std::printf("done: %d%c best: %f                  \r", int(fraction * 100), '%', best_lcp);
    std::fflush(stdout);
  }
  constexpr bool needsGlobalTransformation() const { return false; }
};

int run(const s4p_cli::Options& opt, const Utils::Logger& log) {
  Match4PCSOptions mopt;
  if (!s4p_cli::to_matcher_options(opt, mopt)) {
    log.Log<Utils::ErrorReport>("Invalid overlap configuration. ABORT");
    return -3;
  }
  if (opt.legacy_4pcs) {
    log.Log<Utils::ErrorReport>("-x: the legacy 4PCS matcher is not part of this library (Super4PCS only)");
    return -3;
  }
  IOManager io;
  Mesh P, Q;
  if (!P.load(io, opt.first)) { log.Log<Utils::ErrorReport>("Can't read input set1"); return -1; }
  if (!Q.load(io, opt.second)) { log.Log<Utils::ErrorReport>("Can't read input set2"); return -1; }

  Match4PCSBase::MatrixType mat = Match4PCSBase::MatrixType::Identity();
  float score = 0.f;
  try {
    MatchSuper4PCS matcher(mopt, log);
    log.Log<Utils::Verbose>("Use Super4PCS");
    score = matcher.ComputeTransformation(P.points, &Q.points, mat, Sampling::UniformDistSampler(), Progress());
    const std::vector<Point3D>* sampled[2] = {&matcher.getFirstSampled(), &matcher.getSecondSampled()};
    for (int k = 0; k < 2; ++k) {
      if (opt.sampled[k].empty()) continue;
      log.Log<Utils::Verbose>("Exporting Sampled cloud ", k + 1, " to ", opt.sampled[k].c_str(), " ...");
      save_points(io, opt.sampled[k], *sampled[k]);
    }
  } catch (const std::exception& e) {
    log.Log<Utils::ErrorReport>("[Error]: ", e.what());
    log.Log<Utils::ErrorReport>("Aborting with code -2 ...");
    return -2;
  } catch (...) {
    log.Log<Utils::ErrorReport>("[Unknown Error]: Aborting with code -3 ...");
    return -3;
  }

  log.Log<Utils::Verbose>("Score: ", score);
  log.Log<Utils::Verbose>("(Homogeneous) Transformation from ", opt.second.c_str(), " to ", opt.first.c_str(), ":");
  for (int r = 0; r < 4; ++r) std::printf("%12.6g %12.6g %12.6g %12.6g\n", mat(r, 0), mat(r, 1), mat(r, 2), mat(r, 3));

  if (!opt.matrix.empty()) {
    log.Log<Utils::Verbose>("Exporting Matrix to ", opt.matrix.c_str(), "...");
#ifdef S4P_HAVE_EIGEN
    io.WriteMatrix(opt.matrix, mat.cast<double>(), IOManager::POLYWORKS);
#else
    io.WriteMatrix(opt.matrix, compat::cast_double(mat), IOManager::POLYWORKS);
#endif
  }
  if (!opt.registered.empty()) {
    log.Log<Utils::Verbose>("Exporting Registered geometry to ", opt.registered.c_str(), "...");
    Q.save(io, opt.registered);
  }
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  s4p_cli::Options opt;
  if (argc < 4) {
    s4p_cli::usage(opt, argv[0], false);
    return -2;
  }
  if (s4p_cli::parse(opt, argc, argv) != s4p_cli::Parse::Run) {
    s4p_cli::usage(opt, argv[0], true);
    return 1;            // the reference leaves with 1 for -h and for an unknown flag alike (super4pcs_test.cc:71-76)
  }
  return run(opt, GlobalRegistration::Utils::Logger(GlobalRegistration::Utils::Verbose));
}
