// Drop-in for src/super4pcs/io/io.h + io.cc + io_ply.h (class IOManager, struct tripple): same public interface
// (io.h:20-58), same file formats, same parsed values and -- for the writers -- the same bytes.  Host-only code (no
// device work on this side of the path), header-only so that a wrapper or the CLI needs nothing but -Iinclude.
//
//   ReadObject   dispatch on the last three characters of the name: "ply", "obj", "ptx"      io.cc:20-43
//     OBJ        v / vt / vn / f (four index layouts) / mtllib                                 io.cc:138-268
//     PLY        ascii 1.0, binary little/big endian 1.0; 3, 6 (normals or colours), 7, 9, 10 properties
//                                                                                              io_ply.h:20-361
//     PTX        cols, rows, eight header lines, then "x y z intensity r g b" per line         io.cc:83-136
//   WriteObject  no faces -> binary little-endian PLY, faces -> OBJ (extension replaced)      io.cc:275-303, 330-457
//   WriteMatrix  Polyworks text matrix                                                         io.cc:305-328, 461-482
//
// Differences from the reference, all in places where its behaviour is undefined or unusable:
//   * its OBJ loop tests an uninitialised token buffer on empty lines and on the read after the last newline
//     (io.cc:151-153: `char ch[128]` is not cleared and `sscanf` of an empty line writes nothing), which in practice
//     repeats the previous line's action -- e.g. a duplicated last face; here such lines are skipped;
//   * an OBJ line longer than 1022 characters puts its stream into a fail state it never leaves (endless loop,
//     io.cc:148-149); here long lines are read whole;
//   * textures (map_Kd) need OpenCV in the reference (io.cc:216-262) and are skipped with the same message here;
//   * `vt` fills two of the four coefficients of a 2x2 (io.cc:158-161); the other two are zero here.
#ifndef S4P_FACADE_IO_H_
#define S4P_FACADE_IO_H_

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <locale>
#include <sstream>
#include <string>
#include <vector>

#include "super4pcs/shared4pcs.h"

#ifdef S4P_HAVE_EIGEN
#include <Eigen/Core>
#endif

#ifndef S4P_HAVE_EIGEN
namespace GlobalRegistration {
namespace compat {
struct Matrix2f {
  float m[4];
  Matrix2f() : m{0.f, 0.f, 0.f, 0.f} {}
  float& coeffRef(int i) { return m[i]; }
  const float& coeffRef(int i) const { return m[i]; }
};
struct Matrix4d {
  double m[16];   // column-major like Eigen's default
  Matrix4d() { for (int i = 0; i < 16; ++i) m[i] = 0.0; }
  double& operator()(int r, int c) { return m[c * 4 + r]; }
  double operator()(int r, int c) const { return m[c * 4 + r]; }
};
inline Matrix4d cast_double(const Matrix4f& a) {
  Matrix4d r;
  for (int i = 0; i < 16; ++i) r.m[i] = double(a.m[i]);
  return r;
}
}  // namespace compat
}  // namespace GlobalRegistration
#endif

struct tripple {          // io.h:20-32 (spelling as in the reference)
  int a;
  int b;
  int c;
  int n1;
  int n2;
  int n3;
  int t1;
  int t2;
  int t3;
  tripple() : a(0), b(0), c(0), n1(0), n2(0), n3(0), t1(0), t2(0), t3(0) {}
  tripple(int _a, int _b, int _c) : a(_a), b(_b), c(_c), n1(0), n2(0), n3(0), t1(0), t2(0), t3(0) {}
};

class IOManager {
 public:
  enum MATRIX_MODE { POLYWORKS };
  using Point3D = GlobalRegistration::Point3D;
  using Vec3 = typename Point3D::VectorType;
#ifdef S4P_HAVE_EIGEN
  using TexCoord = Eigen::Matrix2f;
  using Mat4dArg = Eigen::Ref<const Eigen::Matrix<double, 4, 4> >;
#else
  using TexCoord = GlobalRegistration::compat::Matrix2f;
  using Mat4dArg = GlobalRegistration::compat::Matrix4d;
#endif

  inline bool ReadObject(const char* name, std::vector<Point3D>& v, std::vector<TexCoord>& tex_coords,
                         std::vector<Vec3>& normals, std::vector<tripple>& tris, std::vector<std::string>& mtls) {
    const std::string filename(name);
    if (filename.length() < 4) return false;
    const std::string ext = filename.substr(filename.size() - 3);
    if (ext == "ply") return ReadPly(name, v, normals);
    if (ext == "obj") return ReadObj(name, v, tex_coords, normals, tris, mtls);
    if (ext == "ptx") return ReadPtx(name, v);
    std::cerr << "Unsupported file format" << std::endl;
    return false;
  }

  inline bool WriteObject(const char* name, const std::vector<Point3D>& v, const std::vector<TexCoord>& tex_coords,
                          const std::vector<Vec3>& normals, const std::vector<tripple>& tris,
                          const std::vector<std::string>& mtls) {
    std::string filename(name);
    if (filename.size() < 4) return false;
    const bool haveExt = filename.at(filename.size() - 4) == '.';
    if (tris.size() == 0)
      return WritePly(haveExt ? filename.substr(0, filename.size() - 3).append("ply") : filename.append(".ply"), v, normals);
    return WriteObj(haveExt ? filename.substr(0, filename.size() - 3).append("obj") : filename.append(".obj"), v, tex_coords,
                    normals, tris, mtls);
  }

  // Polyworks text matrix: four rows of four fixed-point numbers (six decimals, what std::to_string prints), each padded
  // with a leading blank when it has no minus sign so that the columns line up; two blanks between columns.
  inline bool WriteMatrix(const std::string& name, const Mat4dArg& mat, MATRIX_MODE mode) {
    std::ofstream out(name, std::ofstream::out | std::ofstream::trunc);
    if (mode != POLYWORKS) return false;
    std::string text = "VERSION\t=\t1\nMATRIX\t=\n";
    for (int row = 0; row < 4; ++row) {
      for (int col = 0; col < 4; ++col) {
        const double value = mat(row, col);
        if (col) text += "  ";
        if (!(value < 0.)) text += ' ';             // (NaN prints without a sign, so it gets the pad as well)
        text += std::to_string(value);
      }
      text += '\n';
    }
    out << text;
    return true;
  }

 private:
  static inline Vec3 vec3(float x, float y, float z) { Vec3 r; r(0) = x; r(1) = y; r(2) = z; return r; }

  // ---------------------------------------------------------------- text scanning helper
  // A cursor over one line with the conversions of C's scanf family ("%f", "%d", a literal character), each returning
  // false -- and leaving its output untouched -- at the first mismatch, after which every later conversion fails too.
  // The readers below are written against it; the values it yields are those the reference obtains from sscanf.
  struct LineCursor {
    const char* p;
    bool ok = true;
    explicit LineCursor(const char* s) : p(s) {}
    void skip_blank() { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r') ++p; }
    // first blank-delimited word (empty if the line is blank); stays in front of the rest of the line
    std::string word() {
      skip_blank();
      const char* b = p;
      while (*p && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) ++p;
      return std::string(b, p);
    }
    bool real(float& out) {
      if (!ok) return false;
      skip_blank();
      char* end = nullptr;
      const float v = std::strtof(p, &end);
      if (end == p) return ok = false;
      p = end; out = v;
      return true;
    }
    bool integer(int& out) {
      if (!ok) return false;
      skip_blank();
      char* end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) return ok = false;
      p = end; out = int(v);
      return true;
    }
    bool literal(char c) {                       // no blank skipping: "1//2" matches "%d//%d", "1 //2" does not
      if (!ok) return false;
      if (*p != c) return ok = false;
      ++p;
      return true;
    }
  };

  // ---------------------------------------------------------------- OBJ (io.cc:138-268)
  // Statements: "v x y z", "vt u v", "vn x y z", "f ..." (index layout decided by which attribute lists are non-empty
  // when the face is met), "mtllib name"; everything else is ignored.
  inline bool ReadObj(const char* filename, std::vector<Point3D>& v, std::vector<TexCoord>& tex_coords,
                      std::vector<Vec3>& normals, std::vector<tripple>& tris, std::vector<std::string>& mtls) {
    std::ifstream file(filename, std::ios::in);
    if (!file) return false;
    v.clear();
    tris.clear();
    float xyz[3] = {0.f, 0.f, 0.f};              // survives from line to line: a short "v"/"vn" line reuses the previous values
    auto in_range = [](int i, size_t n) { return i >= 1 && size_t(i) <= n; };   // (the reference indexes unchecked)
    for (std::string line; std::getline(file, line);) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      LineCursor cur(line.c_str());
      const std::string kind = cur.word();
      if (kind == "v" || kind == "vn") {
        cur.real(xyz[0]) && cur.real(xyz[1]) && cur.real(xyz[2]);
        if (kind == "v") {
          v.emplace_back(xyz[0], xyz[1], xyz[2]);
          v.back().set_rgb(vec3(0.f, 0.f, 0.f));
        } else {
          normals.push_back(vec3(xyz[0], xyz[1], xyz[2]));
        }
      } else if (kind == "vt") {
        TexCoord tc;
        for (int k = 0; k < 4; ++k) tc.coeffRef(k) = 0.f;
        cur.real(tc.coeffRef(0)) && cur.real(tc.coeffRef(1));
        tex_coords.push_back(tc);
      } else if (kind == "f") {
        tripple t;
        int* vert[3] = {&t.a, &t.b, &t.c};
        int* tex[3] = {&t.t1, &t.t2, &t.t3};
        int* nrm[3] = {&t.n1, &t.n2, &t.n3};
        const bool with_n = !normals.empty(), with_t = !tex_coords.empty();
        for (int corner = 0; corner < 3 && cur.ok; ++corner) {
          cur.integer(*vert[corner]);
          if (with_n && with_t) { cur.literal('/') && cur.integer(*tex[corner]) && cur.literal('/') && cur.integer(*nrm[corner]); }
          else if (with_n) { cur.literal('/') && cur.literal('/') && cur.integer(*nrm[corner]); }
          else if (with_t) { cur.literal('/') && cur.integer(*tex[corner]); }
        }
        tris.push_back(t);
        if (with_n)
          for (int corner = 0; corner < 3; ++corner)
            if (in_range(*vert[corner], v.size()) && in_range(*nrm[corner], normals.size()))
              v[size_t(*vert[corner] - 1)].set_normal(normals[size_t(*nrm[corner] - 1)]);
      } else if (kind == "mtllib") {
        mtls.push_back(line.size() > 7 ? line.substr(7) : std::string());
      }
    }

    if (tris.empty()) {
      // point set: the i-th normal belongs to the i-th vertex when the two lists have the same length
      if (v.size() == normals.size())
        for (size_t i = 0; i < v.size(); ++i) v[i].set_normal(normals[i]);
    } else if (!normals.empty()) {
      // mesh: normals were attached through the faces; hand them back one per vertex
      normals.clear();
      normals.reserve(v.size());
      for (const Point3D& pt : v) normals.push_back(pt.normal());
    }

    if (!mtls.empty()) {                         // textures need OpenCV in the reference; they are named and skipped
      std::ifstream material(mtls[0].c_str(), std::ios::in);
      for (std::string token; material >> token;)
        if (token == "map_Kd") {
          std::string image;
          material >> image;
          std::cerr << "OpenCV is required to load material textures. Skipping " << image.c_str() << std::endl;
        }
    }
    return !v.empty();
  }

  // ---------------------------------------------------------------- PTX (io.cc:83-136)
  // "columns", "rows", eight lines of scanner pose, then one "x y z intensity r g b" record per line.
  // Numbers are taken with the stream extraction the reference uses (a record that stops early leaves the remaining
  // fields at their previous values and the point is kept), hence a persistent record and istringstream per line.
  inline bool ReadPtx(const char* filename, std::vector<Point3D>& vertex) {
    std::ifstream file(filename, std::ios::in);
    if (!file) {
      std::cerr << "(PTX) error opening file" << std::endl;
      return false;
    }
    std::string line;
    auto header_int = [&]() { int value = 0; std::getline(file, line); std::istringstream(line) >> value; return value; };
    const int columns = header_int();
    const int rows = header_int();
    const long long expected = (long long)columns * rows;
    for (int skipped = 0; skipped < 8; ++skipped) std::getline(file, line);
    vertex.clear();
    if (expected > 0) vertex.reserve(size_t(expected));
    struct { float pos[3] = {0.f, 0.f, 0.f}; float intensity = 0.f; float rgb[3] = {0.f, 0.f, 0.f}; } rec;
    for (long long count = 0; count < expected && !file.eof(); ++count) {
      std::getline(file, line);
      std::istringstream fields(line);
      fields >> rec.pos[0] >> rec.pos[1] >> rec.pos[2] >> rec.intensity >> rec.rgb[0] >> rec.rgb[1] >> rec.rgb[2];
      Point3D pt(rec.pos[0], rec.pos[1], rec.pos[2]);
      pt.set_rgb(vec3(rec.rgb[0], rec.rgb[1], rec.rgb[2]));
      vertex.push_back(pt);
    }
    return (long long)vertex.size() == expected;
  }

  // ---------------------------------------------------------------- PLY (io_ply.h)
  enum PLYFormat { BINARY_BIG_ENDIAN_1, BINARY_LITTLE_ENDIAN_1, ASCII_1 };

  // PLY header (io_ply.h:20-124) as a keyword-driven scan over blank-separated words.  Returns the offset of the first
  // body byte (0 on error).  What counts: "format <kind> 1.0"; "element vertex|face <n>" (other elements switch property
  // counting off); per "property": float/double add one vertex property, uchar adds one and flags colour, list is skipped;
  // "comment" / "*obj_info*" lines are skipped to their end.
  static inline unsigned int readPlyHeader(const char* filename, unsigned int& numOfVertices, unsigned int& numOfFaces,
                                           PLYFormat& format, unsigned int& numOfVertexProperties, bool& haveColor) {
    std::ifstream in(filename, std::ios_base::in | std::ios_base::binary);
    if (!in) {
      std::cerr << "(PLY) error opening file" << std::endl;
      return 0;
    }
    numOfVertexProperties = 0; numOfVertices = 0; numOfFaces = 0; haveColor = false;
    format = ASCII_1;
    auto fail = [](const char* why) { std::cerr << "(PLY) " << why << std::endl; return 0u; };
    std::string word;
    if (!(in >> word) || word != "ply") return fail("not a PLY file");
    bool counting = false;                         // inside an element whose properties are counted (vertex or face)
    while (true) {
      if (!(in >> word)) return fail("error parsing header (no end_header)");   // (the reference never returns here)
      if (word == "end_header") break;
      if (word == "format") {
        std::string kind, version;
        in >> kind >> version;
        if (kind == "ascii") format = ASCII_1;
        else if (kind == "binary_little_endian") format = BINARY_LITTLE_ENDIAN_1;
        else if (kind == "binary_big_endian") format = BINARY_BIG_ENDIAN_1;
        else return fail("error parsing header (format)");
        if (version != "1.0") return fail("error parsing header - bad version");
      } else if (word == "element") {
        std::string what;
        in >> what;
        counting = what == "vertex" || what == "face";
        if (what == "vertex") in >> numOfVertices;
        else if (what == "face") in >> numOfFaces;
        else std::cerr << "(PLY) ignoring unknown element " << what << std::endl;
      } else if (word == "property" && counting) {
        std::string type, skipped;
        in >> type;
        if (type == "float" || type == "double") { ++numOfVertexProperties; in >> skipped; }
        else if (type == "uchar") { ++numOfVertexProperties; haveColor = true; in >> skipped; }
        else if (type == "list") { in >> skipped >> skipped >> skipped; }
        else return fail("error parsing header (property)");
      } else if (word == "comment" || word.find("obj_info") != std::string::npos) {
        std::getline(in, word);
      }
    }
    return (unsigned int)in.tellg() + 1u;          // the byte after the end-of-line that follows "end_header"
  }

  static inline void swap4(void* p, unsigned int count) {   // bigLittleEndianSwap, io_ply.h:127-141
    char* b = static_cast<char*>(p);
    for (unsigned int j = 0; j < count; ++j) {
      char* q = b + 4 * j;
      char c = q[0]; q[0] = q[3]; q[3] = c;
      c = q[1]; q[1] = q[2]; q[2] = c;
    }
  }

  // vertex record -> Point3D (+ normal list), shared by the ascii and binary bodies (io_ply.h:209-231, 312-335)
  static inline void pushPlyVertex(const float* f, const unsigned int* rgb_buff, unsigned int nprop, bool haveColor,
                                   std::vector<Point3D>& vertex, std::vector<Vec3>& normal) {
    vertex.emplace_back(f[0], f[1], f[2]);
    if (nprop == 6) {
      if (haveColor) {
        vertex.back().set_rgb(vec3(float(rgb_buff[0]), float(rgb_buff[1]), float(rgb_buff[2])));
      } else {
        const Vec3 n = vec3(f[3], f[4], f[5]);
        normal.push_back(n);
        vertex.back().set_normal(n);
      }
    } else if (nprop == 7) {
      vertex.back().set_rgb(vec3(float(rgb_buff[0]), float(rgb_buff[1]), float(rgb_buff[2])));
    } else if (nprop == 9 || nprop == 10) {
      const Vec3 n = vec3(f[3], f[4], f[5]);
      normal.push_back(n);
      vertex.back().set_normal(n);
      vertex.back().set_rgb(vec3(float(rgb_buff[0]), float(rgb_buff[1]), float(rgb_buff[2])));
    }
  }

  inline bool ReadPly(const char* filename, std::vector<Point3D>& v, std::vector<Vec3>& normals) {
    std::vector<tripple> face;
    unsigned int nprop = 0, nvert = 0, nface = 0;
    PLYFormat format = ASCII_1;
    bool haveColor = false;
    const unsigned int headerSize = readPlyHeader(filename, nvert, nface, format, nprop, haveColor);
    if (haveColor) std::cout << "haveColor" << std::endl;
    if (headerSize == 0) return false;
    FILE* in = std::fopen(filename, "rb");
    if (!in) {
      std::cerr << "(PLY) error opening file" << std::endl;
      return false;
    }
    std::fseek(in, long(headerSize), SEEK_SET);
    // how many leading floats and trailing colour bytes a vertex record has (io_ply.h:190-206, 282-310)
    unsigned int nfloat = nprop, ncol = 0;
    if (nprop == 10) { nfloat = 6; ncol = 4; }
    else if (nprop == 9) { nfloat = 6; ncol = 3; }
    else if (nprop == 6 && haveColor) { nfloat = 3; ncol = 3; }
    else if (nprop == 7) { nfloat = 3; ncol = 4; }
    std::vector<float> rec(nprop > 3 ? nprop : 3, 0.f);
    unsigned int rgb_buff[4] = {0, 0, 0, 0};
    bool ok = true;
    if (format == ASCII_1) {
      for (unsigned int i = 0; i < nvert && !std::feof(in); i++) {
        for (unsigned int j = 0; j < nfloat; j++) if (std::fscanf(in, "%f", &rec[j]) != 1) break;
        for (unsigned int j = 0; j < ncol; j++) if (std::fscanf(in, "%i", &rgb_buff[j]) != 1) break;
        pushPlyVertex(rec.data(), rgb_buff, nprop, haveColor, v, normals);
      }
      if (nface != 0) {
        if (std::feof(in)) { std::cerr << "(PLY) incomplete file" << std::endl; ok = false; }
        for (unsigned int i = 0; ok && i < nface && !std::feof(in); i++) {
          int f[3] = {0, 0, 0}, polygonSize = 0;
          if (std::fscanf(in, "%d %d %d %d", &polygonSize, &f[0], &f[1], &f[2]) != 4) break;
          face.emplace_back(f[0], f[1], f[2]);
        }
      }
    } else {
      const bool bigEndian = format == BINARY_BIG_ENDIAN_1;
      for (unsigned int i = 0; i < nvert && !std::feof(in); i++) {
        unsigned char cb[4] = {0, 0, 0, 0};
        if (std::fread(rec.data(), 4, nfloat, in) != nfloat) break;   // (a "double" property is read as 4 bytes, as in the reference)
        if (ncol && std::fread(cb, 1, ncol, in) != ncol) break;
        if (bigEndian) swap4(rec.data(), nprop);
        for (int k = 0; k < 4; ++k) rgb_buff[k] = cb[k];
        pushPlyVertex(rec.data(), rgb_buff, nprop, haveColor, v, normals);
      }
      if (nface != 0) {
        if (std::feof(in)) { std::cerr << "(PLY) incomplete file" << std::endl; ok = false; }
        for (unsigned int i = 0; ok && i < nface && !std::feof(in); i++) {
          unsigned int f[3] = {0, 0, 0};
          char polygonSize = 0;
          if (std::fread(&polygonSize, 1, 1, in) != 1) break;
          if (std::fread(f, 4, 3, in) != 3) break;
          if (bigEndian) swap4(f, 3);
          face.emplace_back(int(f[0]), int(f[1]), int(f[2]));
        }
      }
    }
    std::fclose(in);
    return ok;
  }

  // ---------------------------------------------------------------- writers (io.cc:330-457)
  inline bool WritePly(std::string filename, const std::vector<Point3D>& v, const std::vector<Vec3>& normals) {
    std::ofstream plyFile;
    plyFile.open(filename.c_str(), std::ios::out | std::ios::trunc | std::ios::binary);
    if (!plyFile.is_open()) {
      std::cerr << "Cannot open file to write!" << std::endl;
      return false;
    }
    const bool useNormals = normals.size() == v.size();
    bool useColors = false;
    for (size_t i = 0; i != v.size(); i++)
      if (v[i].hasColor()) { useColors = true; break; }
    plyFile.imbue(std::locale::classic());
    plyFile << "ply" << std::endl;
    plyFile << "format binary_little_endian 1.0" << std::endl;
    plyFile << "comment Super4PCS output file" << std::endl;
    plyFile << "element vertex " << v.size() << std::endl;
    plyFile << "property float x" << std::endl;
    plyFile << "property float y" << std::endl;
    plyFile << "property float z" << std::endl;
    if (useNormals) {
      plyFile << "property float nx" << std::endl;
      plyFile << "property float ny" << std::endl;
      plyFile << "property float nz" << std::endl;
    }
    if (useColors) {
      plyFile << "property uchar red" << std::endl;
      plyFile << "property uchar green" << std::endl;
      plyFile << "property uchar blue" << std::endl;
    }
    plyFile << "end_header" << std::endl;
    for (size_t i = 0; i != v.size(); i++) {
      const float xyz[3] = {v[i].x(), v[i].y(), v[i].z()};
      plyFile.write(reinterpret_cast<const char*>(xyz), 3 * sizeof(float));
      if (useNormals) {
        const float n[3] = {normals[i](0), normals[i](1), normals[i](2)};
        plyFile.write(reinterpret_cast<const char*>(n), 3 * sizeof(float));
      }
      if (useColors) {
        for (int k = 0; k < 3; ++k) {
          const char c = char(v[i].rgb()[k]);      // float -> char, as the reference
          plyFile.write(&c, 1);
        }
      }
    }
    plyFile.close();
    return true;
  }

  inline bool WriteObj(std::string filename, const std::vector<Point3D>& v, const std::vector<TexCoord>& tex_coords,
                       const std::vector<Vec3>& normals, const std::vector<tripple>& tris, const std::vector<std::string>& mtls) {
    std::ofstream f(filename.c_str(), std::ios::out);
    if (!f || f.fail()) return false;
    for (size_t i = 0; i < mtls.size(); ++i) f << "mtllib " << mtls[i] << std::endl;
    for (size_t i = 0; i < v.size(); ++i) {
      f << "v " << v[i].x() << " " << v[i].y() << " " << v[i].z() << " ";
      if (v[i].rgb()[0] != 0) f << v[i].rgb()[0] << " " << v[i].rgb()[1] << " " << v[i].rgb()[2];
      f << std::endl;
    }
    for (size_t i = 0; i < normals.size(); ++i) f << "vn " << normals[i](0) << " " << normals[i](1) << " " << normals[i](2) << std::endl;
    for (size_t i = 0; i < tex_coords.size(); ++i) f << "vt " << tex_coords[i].coeffRef(0) << " " << tex_coords[i].coeffRef(1) << std::endl;
    for (size_t i = 0; i < tris.size(); ++i) {
      if (!normals.size() && !tex_coords.size())
        f << "f " << tris[i].a << " " << tris[i].b << " " << tris[i].c << std::endl;
      else if (tex_coords.size())
        f << "f " << tris[i].a << "/" << tris[i].t1 << " " << tris[i].b << "/" << tris[i].t2 << " " << tris[i].c << "/" << tris[i].t3 << std::endl;
      else
        f << "f " << tris[i].a << "/" << tris[i].n1 << " " << tris[i].b << "/" << tris[i].n2 << " " << tris[i].c << "/" << tris[i].n3 << std::endl;
    }
    f.close();
    return true;
  }
};

#endif
