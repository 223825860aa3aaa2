// Meshlab filter "Global registration": aligns the target mesh to the reference mesh with Super4PCS running on the
// MI355X (facade headers include/super4pcs/**).  Counterpart of the reference's
// demos/MeshlabPlugin/filter_globalregistration/globalregistration.cpp: same parameters (names, defaults, help texts refer
// to the same command-line options), same result (the target's Tr matrix), other code.
#include "globalregistration.h"

#include <QtScript>

#include <stdexcept>
#include <vector>

#include "super4pcs/algorithms/4pcs.h"
#include "super4pcs/algorithms/super4pcs.h"

namespace {

namespace GR = GlobalRegistration;

// one row per dialog entry: created in initParameterSet, read back in applyFilter
struct FloatParam { const char* name; float value; const char* label; const char* help; };
struct IntParam { const char* name; int value; const char* label; const char* help; };
const FloatParam kFloatParams[] = {
    {"delta", 0.1f, "Registration tolerance",
     "Tolerance value for the congruent set exploration and LCP computation (command line option: -d)"},
    {"norm_diff", -1.f, "Filter: difference of normal (degrees)",
     "Allowed difference of normals allowed between corresponding pairs of points(command line option: -a)"},
    {"color_diff", -1.f, "Filter: difference color",
     "Allowed difference of colors allowed between corresponding pairs of points(command line option: -c)"},
};
const IntParam kIntParams[] = {
    {"nbSamples", 200, "Number of samples", "Number of samples used in each mesh (command line option: -n)"},
    {"max_time_seconds", 10000, "Max. Computation time, in seconds",
     "Stop the computation before the end of the exploration (command line option: -t)"},
};

std::vector<GR::Point3D> vertices_of(const CMeshO& mesh) {
  std::vector<GR::Point3D> pts;
  pts.reserve(mesh.vert.size());
  for (std::size_t i = 0; i < mesh.vert.size(); ++i) {
    GR::Point3D p;
    mesh.vert[i].P().ToEigenVector(p.pos());
    pts.push_back(p);
  }
  return pts;
}

// reports every trial's best LCP in the Meshlab log; the transformation is only applied at the end
struct LogVisitor {
  GlobalRegistrationPlugin* plugin;
  template <class Matrix>
  void operator()(float /*fraction*/, float best_lcp, Matrix&&) const { plugin->Log("Found new configuration. LCP = %f", best_lcp); }
  constexpr bool needsGlobalTransformation() const { return false; }
};

}  // namespace

GlobalRegistrationPlugin::GlobalRegistrationPlugin() {
  typeList << FP_GLOBAL_REGISTRATION;
  foreach (FilterIDType tt, types()) actionList << new QAction(filterName(tt), this);
}

QString GlobalRegistrationPlugin::filterName(FilterIDType filterId) const {
  return filterId == FP_GLOBAL_REGISTRATION ? QString("Global registration") : QString();
}

QString GlobalRegistrationPlugin::filterInfo(FilterIDType filterId) const {
  return filterId == FP_GLOBAL_REGISTRATION ? QString("Compute the rigid transforation aligning two 3d objets.") : QString("Unknown Filter");
}

GlobalRegistrationPlugin::FilterClass GlobalRegistrationPlugin::getClass(QAction* a) {
  return ID(a) == FP_GLOBAL_REGISTRATION ? MeshFilterInterface::PointSet : MeshFilterInterface::Generic;
}

void GlobalRegistrationPlugin::initParameterSet(QAction* action, MeshDocument& md, RichParameterSet& parlst) {
  if (ID(action) != FP_GLOBAL_REGISTRATION) return;
  parlst.addParam(new RichMesh("refMesh", md.mm(), &md, "Reference Mesh", "Reference point-cloud or mesh"));
  parlst.addParam(new RichMesh("targetMesh", md.mm(), &md, "Target Mesh", "Point-cloud or mesh to be aligned to the reference"));
  parlst.addParam(new RichAbsPerc("overlap", 50, 0, 100, "Overlap Ratio", "Overlap ratio between the two clouds (command line option: -o)"));
  for (const FloatParam& p : kFloatParams) parlst.addParam(new RichFloat(p.name, p.value, p.label, p.help));
  for (const IntParam& p : kIntParams) parlst.addParam(new RichInt(p.name, p.value, p.label, p.help));
  parlst.addParam(new RichBool("useSuper4PCS", true, "Use Super4PCS", "When disable, use 4PCS algorithm (command line option: -x"));
}

bool GlobalRegistrationPlugin::applyFilter(QAction* /*filter*/, MeshDocument& /*md*/, RichParameterSet& par, vcg::CallBackPos* /*cb*/) {
  CMeshO& reference = par.getMesh("refMesh")->cm;
  CMeshO& target = par.getMesh("targetMesh")->cm;

  GR::Match4PCSOptions opt;
  opt.configureOverlap(par.getAbsPerc("overlap") / 100.f);
  opt.delta = par.getFloat("delta");
  opt.sample_size = par.getInt("nbSamples");
  opt.max_normal_difference = par.getFloat("norm_diff");
  opt.max_color_distance = par.getFloat("color_diff");
  opt.max_time_seconds = par.getInt("max_time_seconds");

  GR::Utils::Logger logger(GR::Utils::LogLevel::NoLog);
  GR::Match4PCSBase::MatrixType mat;
  std::vector<GR::Point3D> set1 = vertices_of(reference), set2 = vertices_of(target);
  GR::Sampling::UniformDistSampler sampler;
  LogVisitor visitor{this};
  try {
    // legacy 4PCS is not part of the MI355X build: its constructor throws, which is reported in the log below
    std::unique_ptr<GR::Match4PCSBase> matcher;
    if (par.getBool("useSuper4PCS")) matcher.reset(new GR::MatchSuper4PCS(opt, logger));
    else matcher.reset(new GR::Match4PCS(opt, logger));
    const float score = matcher->ComputeTransformation(set1, &set2, mat, sampler, visitor);
    Log("Final LCP = %f", score);
  } catch (const std::exception& e) {
    Log("Global registration failed: %s", e.what());
    return false;
  }
  target.Tr.FromEigenMatrix(mat);
  return true;
}

MESHLAB_PLUGIN_NAME_EXPORTER(GlobalRegistrationPlugin)
