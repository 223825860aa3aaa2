// The Super4PCS command-line program on the MI355X path: same flags, same files in and out, same exit codes as the
// reference's demos/Super4PCS/super4pcs_test.cc (usage: doc/Usage.md, scripts/run-example.sh:68).
//   Super4PCS -i P.obj Q.obj [-o overlap] [-d delta] [-n samples] [-t seconds] [-a normal_deg] [-c colour]
//             [-r registered_geometry] [-m polyworks_matrix] [--sampled1 file] [--sampled2 file]
// -x (legacy 4PCS matcher, algorithms/4pcs.cc) is outside the scope of this library and is refused.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <iostream>
#include <string>
#include <type_traits>
#include <vector>

#include "super4pcs/algorithms/super4pcs.h"
#include "super4pcs/io/io.h"
#include "super4pcs/utils/geometry.h"

#include "../demo-utils.h"

using namespace GlobalRegistration;

static void printS4PCSParameterList(const Demo::Args& a) {
  std::fprintf(stderr, "\t[ -r result_file_name (%s) ]\n", a.output.c_str());
  std::fprintf(stderr, "\t[ -m output matrix file (%s) ]\n", a.outputMat.c_str());
  std::fprintf(stderr, "\t[ -x (use 4pcs: not available in this build) ]\n");
  std::fprintf(stderr, "\t[ --sampled1 (output sampled cloud 1) ]\n");
  std::fprintf(stderr, "\t[ --sampled2 (output sampled cloud 2) ]\n");
}

// progress line of the reference (super4pcs_test.cc:29-42); per-candidate calls (fraction < 0) print nothing
struct TransformVisitor {
  inline void operator()(float fraction, float best_LCP, Match4PCSBase::MatrixRef) const {
    if (fraction >= 0) {
      std::printf("done: %d%c best: %f                  \r", static_cast<int>(fraction * 100), '%', best_LCP);
      std::fflush(stdout);
    }
  }
  constexpr bool needsGlobalTransformation() const { return false; }
};

static void printMatrix(const Match4PCSBase::MatrixType& m) {
  for (int r = 0; r < 4; ++r) std::printf("%12.6g %12.6g %12.6g %12.6g\n", m(r, 0), m(r, 1), m(r, 2), m(r, 3));
}

int main(int argc, char** argv) {
  std::vector<Point3D> set1, set2;
  std::vector<IOManager::TexCoord> tex_coords1, tex_coords2;
  std::vector<Point3D::VectorType> normals1, normals2;
  std::vector<tripple> tris1, tris2;
  std::vector<std::string> mtls1, mtls2;
  Point3D::Scalar score = 0;

  constexpr Utils::LogLevel loglvl = Utils::Verbose;
  using SamplerType = Sampling::UniformDistSampler;
  SamplerType sampler;
  TransformVisitor visitor;
  Utils::Logger logger(loglvl);
  Demo::Args args;

  if (argc < 4) {
    Demo::printUsage(args, argv);
    std::exit(-2);
  }
  if (const int c = Demo::getArgs(args, argc, argv)) {
    Demo::printUsage(args, argv);
    printS4PCSParameterList(args);
    std::exit(1);      // both -h and an unknown flag: `int c = getArgs(...) != 0` makes the reference exit(1) for either (super4pcs_test.cc:71-76)
  }
  Match4PCSOptions options;
  Match4PCSBase::MatrixType mat = Match4PCSBase::MatrixType::Identity();
  if (!Demo::setOptionsFromArgs(args, options, logger)) std::exit(-3);
  if (!args.use_super4pcs) {
    logger.Log<Utils::ErrorReport>("-x: the legacy 4PCS matcher is not part of this library (Super4PCS only)");
    std::exit(-3);
  }

  IOManager iomanager;
  if (!iomanager.ReadObject(args.input1.c_str(), set1, tex_coords1, normals1, tris1, mtls1)) {
    logger.Log<Utils::ErrorReport>("Can't read input set1");
    std::exit(-1);
  }
  if (!iomanager.ReadObject(args.input2.c_str(), set2, tex_coords2, normals2, tris2, mtls2)) {
    logger.Log<Utils::ErrorReport>("Can't read input set2");
    std::exit(-1);
  }
  // clean only point sets, to keep the face -> vertex indexation of meshes
  if (tris1.size() == 0) Utils::CleanInvalidNormals(set1, normals1);
  if (tris2.size() == 0) Utils::CleanInvalidNormals(set2, normals2);

  try {
    MatchSuper4PCS matcher(options, logger);
    logger.Log<Utils::Verbose>("Use Super4PCS");
    score = matcher.ComputeTransformation(set1, &set2, mat, sampler, visitor);
    const std::vector<IOManager::TexCoord> no_tex;
    const std::vector<Point3D::VectorType> no_normals;
    const std::vector<tripple> no_tris;
    const std::vector<std::string> no_mtls;
    if (!args.outputSampled1.empty()) {
      logger.Log<Utils::Verbose>("Exporting Sampled cloud 1 to ", args.outputSampled1.c_str(), " ...");
      iomanager.WriteObject(args.outputSampled1.c_str(), matcher.getFirstSampled(), no_tex, no_normals, no_tris, no_mtls);
      logger.Log<Utils::Verbose>("Export DONE");
    }
    if (!args.outputSampled2.empty()) {
      logger.Log<Utils::Verbose>("Exporting Sampled cloud 2 to ", args.outputSampled2.c_str(), " ...");
      iomanager.WriteObject(args.outputSampled2.c_str(), matcher.getSecondSampled(), no_tex, no_normals, no_tris, no_mtls);
      logger.Log<Utils::Verbose>("Export DONE");
    }
  } catch (const std::exception& e) {
    logger.Log<Utils::ErrorReport>("[Error]: ", e.what());
    logger.Log<Utils::ErrorReport>("Aborting with code -2 ...");
    return -2;
  } catch (...) {
    logger.Log<Utils::ErrorReport>("[Unknown Error]: Aborting with code -3 ...");
    return -3;
  }

  logger.Log<Utils::Verbose>("Score: ", score);
  logger.Log<Utils::Verbose>("(Homogeneous) Transformation from ", args.input2.c_str(), " to ", args.input1.c_str(), ": ");
  printMatrix(mat);

  if (!args.outputMat.empty()) {
    logger.Log<Utils::Verbose>("Exporting Matrix to ", args.outputMat.c_str(), "...");
#ifdef S4P_HAVE_EIGEN
    iomanager.WriteMatrix(args.outputMat, mat.cast<double>(), IOManager::POLYWORKS);
#else
    iomanager.WriteMatrix(args.outputMat, compat::cast_double(mat), IOManager::POLYWORKS);
#endif
    logger.Log<Utils::Verbose>("Export DONE");
  }
  if (!args.output.empty()) {
    logger.Log<Utils::Verbose>("Exporting Registered geometry to ", args.output.c_str(), "...");
    iomanager.WriteObject(args.output.c_str(), set2, tex_coords2, normals2, tris2, mtls2);
    logger.Log<Utils::Verbose>("Export DONE");
  }
  return 0;
}
