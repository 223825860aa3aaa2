// Test harness for the IO row: ONE source, compiled twice --
//   * against include/ (the product's IOManager, -DS4P_NO_EIGEN), and
//   * against the reference's own io.cc / io.h / geometry.h with oracle/eigen_shim (`make -C oracle ref`),
// so tests/test_io.py can compare what the two produce from the same files.
//   read  <file> <dump>          ReadObject, then dump everything it returned (hex floats)
//   clean <file> <dump>          ReadObject + Utils::CleanInvalidNormals for point sets (as the CLI does), then dump
//   write <file> <out>           ReadObject, then WriteObject(out, ...) with what was read
//   matrix <out> m00 m01 ... m33 WriteMatrix(POLYWORKS), row-major arguments
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "super4pcs/io/io.h"
#include "super4pcs/utils/geometry.h"

using GlobalRegistration::Point3D;

#ifdef S4P_IO_HARNESS_REFERENCE
typedef Eigen::Matrix2f TexCoord;
#else
typedef IOManager::TexCoord TexCoord;
#endif

static void dump(const char* path, bool ok, const std::vector<Point3D>& v, const std::vector<TexCoord>& tex,
                 const std::vector<Point3D::VectorType>& normals, const std::vector<tripple>& tris, const std::vector<std::string>& mtls) {
  FILE* f = std::fopen(path, "w");
  std::fprintf(f, "ok %d v %zu tex %zu normals %zu tris %zu mtls %zu\n", int(ok), v.size(), tex.size(), normals.size(), tris.size(), mtls.size());
  for (const Point3D& p : v)
    std::fprintf(f, "v %a %a %a n %a %a %a c %a %a %a\n", double(p.x()), double(p.y()), double(p.z()), double(p.normal()(0)),
                 double(p.normal()(1)), double(p.normal()(2)), double(p.rgb()(0)), double(p.rgb()(1)), double(p.rgb()(2)));
  for (const auto& n : normals) std::fprintf(f, "vn %a %a %a\n", double(n(0)), double(n(1)), double(n(2)));
  for (const auto& t : tex) std::fprintf(f, "vt %a %a\n", double(t.coeffRef(0)), double(t.coeffRef(1)));
  for (const tripple& t : tris) std::fprintf(f, "f %d %d %d\n", t.a, t.b, t.c);
  for (const std::string& m : mtls) std::fprintf(f, "mtl %s\n", m.c_str());
  std::fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  IOManager io;
  std::vector<Point3D> v;
  std::vector<TexCoord> tex;
  std::vector<Point3D::VectorType> normals;
  std::vector<tripple> tris;
  std::vector<std::string> mtls;
  const std::string cmd = argv[1];
  if (cmd == "matrix") {
    if (argc != 19) return 2;
#ifdef S4P_IO_HARNESS_REFERENCE
    Eigen::Matrix<double, 4, 4> m;
#else
    IOManager::Mat4dArg m;
#endif
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m(r, c) = std::atof(argv[3 + 4 * r + c]);
    return io.WriteMatrix(argv[2], m, IOManager::POLYWORKS) ? 0 : 1;
  }
  if (argc != 4) return 2;
  const bool ok = io.ReadObject(argv[2], v, tex, normals, tris, mtls);
  if (cmd == "read") { dump(argv[3], ok, v, tex, normals, tris, mtls); return 0; }
  if (cmd == "clean") {
    if (tris.size() == 0) GlobalRegistration::Utils::CleanInvalidNormals(v, normals);
    dump(argv[3], ok, v, tex, normals, tris, mtls);
    return 0;
  }
  if (cmd == "write") return ok && io.WriteObject(argv[3], v, tex, normals, tris, mtls) ? 0 : 1;
  return 2;
}
