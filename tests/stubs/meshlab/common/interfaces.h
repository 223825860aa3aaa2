// TEST STUB (neither Qt nor Meshlab): the subset of Qt + Meshlab's common/interfaces.h that
// demos/MeshlabPlugin/filter_globalregistration touches, so the plugin source can be syntax-checked without them.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <Eigen/Core>
struct QString { std::string s; QString() {} QString(const char* c) : s(c) {} };
struct QObject { virtual ~QObject() {} };
struct QAction { QAction(const QString&, QObject*) {} int id = 0; };
template <class T> struct QList : std::vector<T> { QList& operator<<(const T& v) { this->push_back(v); return *this; } };
#define foreach(decl, container) for (decl : container)
#define Q_OBJECT
#define Q_INTERFACES(x)
#define MESHLAB_PLUGIN_IID_EXPORTER(x)
#define MESH_FILTER_INTERFACE_IID "stub"
#define MESHLAB_PLUGIN_NAME_EXPORTER(x)
namespace vcg { typedef bool CallBackPos(const int, const char*); }
struct StubPoint3 { float v[3]; template <class V> void ToEigenVector(V& out) const { out(0) = v[0]; out(1) = v[1]; out(2) = v[2]; } };
struct StubVertex { StubPoint3 p; const StubPoint3& P() const { return p; } };
struct StubMatrix44 { float m[16]; template <class M> void FromEigenMatrix(const M& e) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m[4 * r + c] = e(r, c); } };
struct CMeshO { std::vector<StubVertex> vert; StubMatrix44 Tr; };
struct MeshModel { enum { MM_VERTCOORD = 1 }; CMeshO cm; };
struct MeshDocument { MeshModel model; MeshModel* mm() { return &model; } };
struct RichParameter { virtual ~RichParameter() {} };
struct RichMesh : RichParameter { RichMesh(const char*, MeshModel*, MeshDocument*, const char*, const char*) {} };
struct RichAbsPerc : RichParameter { RichAbsPerc(const char*, float, float, float, const char*, const char*) {} };
struct RichFloat : RichParameter { RichFloat(const char*, float, const char*, const char*) {} };
struct RichInt : RichParameter { RichInt(const char*, int, const char*, const char*) {} };
struct RichBool : RichParameter { RichBool(const char*, bool, const char*, const char*) {} };
struct RichParameterSet {
  std::vector<std::unique_ptr<RichParameter>> params; MeshModel a, b;
  void addParam(RichParameter* p) { params.emplace_back(p); }
  MeshModel* getMesh(const char* n) { return std::string(n) == "refMesh" ? &a : &b; }
  float getAbsPerc(const char*) const { return 50.f; }
  float getFloat(const char*) const { return 0.1f; }
  int getInt(const char*) const { return 200; }
  bool getBool(const char*) const { return true; }
};
class MeshFilterInterface {
 public:
  typedef int FilterIDType;
  enum FilterClass { Generic = 0, PointSet = 1 };
  enum FILTER_ARITY { SINGLE_MESH = 0 };
  virtual ~MeshFilterInterface() {}
  QList<FilterIDType> types() const { return typeList; }
  FilterIDType ID(QAction* a) const { return a->id; }
  void Log(const char* fmt, ...) { va_list ap; va_start(ap, fmt); std::vfprintf(stderr, fmt, ap); va_end(ap); }
 protected:
  QList<FilterIDType> typeList;
  QList<QAction*> actionList;
};
