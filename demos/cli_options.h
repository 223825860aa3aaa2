// Command-line surface of the Super4PCS program: a table of the flags the reference's binary understands
// (demos/demo-utils.h:119-162: -i -o -d -c -t -a -n -r -m -x --sampled1 --sampled2 -h), with its defaults
// (demo-utils.h:57-101), so that scripts written for it (scripts/run-example.sh:68) run unchanged.
#ifndef S4P_CLI_OPTIONS_H_
#define S4P_CLI_OPTIONS_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <super4pcs/shared4pcs.h>

namespace s4p_cli {

struct Options {
  std::string first = "input1.obj", second = "input2.obj";   // -i P Q
  std::string registered;                                    // -r  second input after registration
  std::string matrix;                                        // -m  Polyworks matrix file
  std::string sampled[2];                                    // --sampled1 / --sampled2
  double overlap = 0.2, delta = 5.0, colour = -1, normal_deg = -1;
  int samples = 200, seconds = 10;
  bool legacy_4pcs = false;                                  // -x
};

enum class Parse { Run, Help, Bad };

// One row per flag: how many values follow and where they go.
struct Flag {
  const char* name;
  int values;
  void (*store)(Options&, char** v);
};

inline const Flag* flag_table(size_t* n) {
  static const Flag table[] = {
      {"-i", 2, [](Options& o, char** v) { o.first = v[0]; o.second = v[1]; }},
      {"-o", 1, [](Options& o, char** v) { o.overlap = std::atof(v[0]); }},
      {"-d", 1, [](Options& o, char** v) { o.delta = std::atof(v[0]); }},
      {"-c", 1, [](Options& o, char** v) { o.colour = std::atof(v[0]); }},
      {"-t", 1, [](Options& o, char** v) { o.seconds = std::atoi(v[0]); }},
      {"-a", 1, [](Options& o, char** v) { o.normal_deg = std::atof(v[0]); }},
      {"-n", 1, [](Options& o, char** v) { o.samples = std::atoi(v[0]); }},
      {"-r", 1, [](Options& o, char** v) { o.registered = v[0]; }},
      {"-m", 1, [](Options& o, char** v) { o.matrix = v[0]; }},
      {"-x", 0, [](Options& o, char**) { o.legacy_4pcs = true; }},
      {"--sampled1", 1, [](Options& o, char** v) { o.sampled[0] = v[0]; }},
      {"--sampled2", 1, [](Options& o, char** v) { o.sampled[1] = v[0]; }},
  };
  *n = sizeof(table) / sizeof(table[0]);
  return table;
}

inline Parse parse(Options& o, int argc, char** argv) {
  size_t nflags = 0;
  const Flag* table = flag_table(&nflags);
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-h")) return Parse::Help;
    const Flag* hit = nullptr;
    for (size_t k = 0; k < nflags && !hit; ++k)
      if (!std::strcmp(argv[i], table[k].name)) hit = &table[k];
    if (!hit) {
      if (argv[i][0] == '-') { std::fputs("Unknown flag\n", stderr); return Parse::Bad; }
      continue;                                   // stray words are ignored, as the reference does
    }
    if (i + hit->values > argc - 1) return Parse::Bad;      // value(s) missing (the reference reads past argv here)
    hit->store(o, argv + i + 1);
    i += hit->values;
  }
  // neither geometry nor matrix requested: write the registered geometry under the reference's default name
  if (o.registered.empty() && o.matrix.empty()) o.registered = "output.obj";
  return Parse::Run;
}

inline void usage(const Options& o, const char* prog, bool all) {
  std::fprintf(stderr, "\nUsage: %s -i input1 input2\n", prog);
  std::fprintf(stderr, "Parameter list:\n");
  std::fprintf(stderr, "\t[ -o overlap (%2.2f) ]\n\t[ -d delta (%2.2f) ]\n\t[ -n n_points (%d) ]\n", o.overlap, o.delta, o.samples);
  std::fprintf(stderr, "\t[ -a norm_diff (%f) ]\n\t[ -c max_color_diff (%f) ]\n\t[ -t max_time_seconds (%d) ]\n", o.normal_deg, o.colour, o.seconds);
  if (!all) return;
  std::fprintf(stderr, "\t[ -r result_file_name (%s) ]\n\t[ -m output matrix file (%s) ]\n", o.registered.c_str(), o.matrix.c_str());
  std::fprintf(stderr, "\t[ -x (legacy 4PCS: not available in this build) ]\n");
  std::fprintf(stderr, "\t[ --sampled1 file ] [ --sampled2 file ]  (sampled clouds)\n");
}

// false: the overlap / terminate-threshold pair is inconsistent (Match4PCSOptions::configureOverlap)
inline bool to_matcher_options(const Options& o, GlobalRegistration::Match4PCSOptions& m) {
  using Scalar = GlobalRegistration::Match4PCSOptions::Scalar;
  if (!m.configureOverlap(Scalar(o.overlap))) return false;
  m.delta = Scalar(o.delta);
  m.sample_size = size_t(o.samples);
  m.max_time_seconds = o.seconds;
  m.max_normal_difference = Scalar(o.normal_deg);
  m.max_color_distance = Scalar(o.colour);
  return true;
}

}  // namespace s4p_cli
#endif
