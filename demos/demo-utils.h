// Command-line options shared by the demo programs: the flag set, defaults and exit conventions of the reference's
// demos/demo-utils.h:57-180 (namespace GlobalRegistration::Demo), so scripts written for its Super4PCS binary
// (scripts/run-example.sh:68: `-i a.obj b.obj -o 0.7 -d 0.01 -t 1000 -n 200 -r out.obj -m mat.txt`) run unchanged.
#ifndef S4P_DEMO_UTILS_H_
#define S4P_DEMO_UTILS_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include <super4pcs/shared4pcs.h>
#include <super4pcs/utils/logger.h>

namespace GlobalRegistration {
namespace Demo {

struct Args {
  std::string input1 = "input1.obj";        // -i first second
  std::string input2 = "input2.obj";
  std::string output = "";                  // -r  transformed second input
  std::string defaultObjOutput = "output.obj";
  std::string outputMat = "";               // -m  Polyworks matrix file
  std::string outputSampled1 = "";          // --sampled1 / --sampled2
  std::string outputSampled2 = "";
  double delta = 5.0;                       // -d
  double overlap = 0.2;                     // -o
  double thr = 1.0;                         // terminate threshold (no flag in the reference either)
  double max_color = -1;                    // -c
  int n_points = 200;                       // -n
  double norm_diff = -1;                    // -a
  int max_time_seconds = 10;                // -t
  bool use_super4pcs = true;                // -x selects the legacy 4PCS matcher
};

static inline void printParameterList(const Args& a) {
  std::fprintf(stderr, "Parameter list:\n");
  std::fprintf(stderr, "\t[ -o overlap (%2.2f) ]\n", a.overlap);
  std::fprintf(stderr, "\t[ -d delta (%2.2f) ]\n", a.delta);
  std::fprintf(stderr, "\t[ -n n_points (%d) ]\n", a.n_points);
  std::fprintf(stderr, "\t[ -a norm_diff (%f) ]\n", a.norm_diff);
  std::fprintf(stderr, "\t[ -c max_color_diff (%f) ]\n", a.max_color);
  std::fprintf(stderr, "\t[ -t max_time_seconds (%d) ]\n", a.max_time_seconds);
}

static inline void printUsage(const Args& a, char** argv) {
  std::fprintf(stderr, "\nUsage: %s -i input1 input2\n", argv[0]);
  printParameterList(a);
}

// 0: go on, 1: help requested, -1: unknown flag / missing value (the reference reads past argv in that case)
static inline int getArgs(Args& a, int argc, char** argv) {
  int i = 1;
  auto value = [&](int n) { return i + n < argc; };
  while (i < argc) {
    const char* f = argv[i];
    if (!std::strcmp(f, "-i")) { if (!value(2)) return -1; a.input1 = argv[++i]; a.input2 = argv[++i]; }
    else if (!std::strcmp(f, "-o")) { if (!value(1)) return -1; a.overlap = std::atof(argv[++i]); }
    else if (!std::strcmp(f, "-d")) { if (!value(1)) return -1; a.delta = std::atof(argv[++i]); }
    else if (!std::strcmp(f, "-c")) { if (!value(1)) return -1; a.max_color = std::atof(argv[++i]); }
    else if (!std::strcmp(f, "-t")) { if (!value(1)) return -1; a.max_time_seconds = std::atoi(argv[++i]); }
    else if (!std::strcmp(f, "-a")) { if (!value(1)) return -1; a.norm_diff = std::atof(argv[++i]); }
    else if (!std::strcmp(f, "-n")) { if (!value(1)) return -1; a.n_points = std::atoi(argv[++i]); }
    else if (!std::strcmp(f, "-r")) { if (!value(1)) return -1; a.output = argv[++i]; }
    else if (!std::strcmp(f, "-m")) { if (!value(1)) return -1; a.outputMat = argv[++i]; }
    else if (!std::strcmp(f, "-x")) { a.use_super4pcs = false; }
    else if (!std::strcmp(f, "--sampled1")) { if (!value(1)) return -1; a.outputSampled1 = argv[++i]; }
    else if (!std::strcmp(f, "--sampled2")) { if (!value(1)) return -1; a.outputSampled2 = argv[++i]; }
    else if (!std::strcmp(f, "-h")) { return 1; }
    else if (f[0] == '-') { std::cerr << "Unknown flag\n"; return -1; }
    i++;
  }
  // no output file (geometry / matrix) requested: write the registered geometry
  if (a.output.empty() && a.outputMat.empty()) a.output = a.defaultObjOutput;
  return 0;
}

static inline bool setOptionsFromArgs(const Args& a, Match4PCSOptions& options, const Utils::Logger& logger = Utils::Logger()) {
  if (!options.configureOverlap(Match4PCSOptions::Scalar(a.overlap))) {
    logger.Log<Utils::ErrorReport>("Invalid overlap configuration. ABORT");
    return false;
  }
  options.sample_size = size_t(a.n_points);
  options.max_normal_difference = Match4PCSOptions::Scalar(a.norm_diff);
  options.max_color_distance = Match4PCSOptions::Scalar(a.max_color);
  options.max_time_seconds = a.max_time_seconds;
  options.delta = Match4PCSOptions::Scalar(a.delta);
  return true;
}

}  // namespace Demo
}  // namespace GlobalRegistration
#endif
