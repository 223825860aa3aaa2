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

  inline bool WriteMatrix(const std::string& name, const Mat4dArg& mat, MATRIX_MODE mode) {
    std::ofstream sstr;
    sstr.open(name, std::ofstream::out | std::ofstream::trunc);
    bool status = false;
    if (mode == POLYWORKS) {
      auto formatValue = [](double v) { return v >= 0. ? std::string(" ") + std::to_string(v) : std::to_string(v); };
      sstr << "VERSION\t=\t1\n";
      sstr << "MATRIX\t=\n";
      for (int j = 0; j != 4; ++j)
        sstr << formatValue(mat(j, 0)) << "  " << formatValue(mat(j, 1)) << "  " << formatValue(mat(j, 2)) << "  "
             << formatValue(mat(j, 3)) << "\n";
      status = true;
    }
    sstr.close();
    return status;
  }

 private:
  static inline Vec3 vec3(float x, float y, float z) { Vec3 r; r(0) = x; r(1) = y; r(2) = z; return r; }

  // ---------------------------------------------------------------- OBJ (io.cc:138-268)
  inline bool ReadObj(const char* filename, std::vector<Point3D>& v, std::vector<TexCoord>& tex_coords,
                      std::vector<Vec3>& normals, std::vector<tripple>& tris, std::vector<std::string>& mtls) {
    std::ifstream f(filename, std::ios::in);
    if (!f || f.fail()) return false;
    v.clear();
    tris.clear();
    float x = 0.f, y = 0.f, z = 0.f;          // kept across lines like the reference's locals (a short "v" line reuses them)
    std::string line;
    while (std::getline(f, line)) {
      if (!line.empty() && line.back() == '\r') line.pop_back();
      const char* str = line.c_str();
      char ch[128];
      ch[0] = '\0';
      if (std::sscanf(str, "%127s", ch) != 1) continue;
      if (std::strcmp(ch, "v") == 0) {
        std::sscanf(str, "%*s %f %f %f", &x, &y, &z);
        v.emplace_back(x, y, z);
        v.back().set_rgb(vec3(0.f, 0.f, 0.f));
      } else if (std::strcmp(ch, "vt") == 0) {
        TexCoord tc;
        tc.coeffRef(0) = tc.coeffRef(1) = tc.coeffRef(2) = tc.coeffRef(3) = 0.f;
        std::sscanf(str, "%*s %f %f", &tc.coeffRef(0), &tc.coeffRef(1));
        tex_coords.push_back(tc);
      } else if (std::strcmp(ch, "vn") == 0) {
        std::sscanf(str, "%*s %f %f %f", &x, &y, &z);
        normals.push_back(vec3(x, y, z));
      } else if (std::strcmp(ch, "f") == 0) {
        tripple t;
        if (normals.size() && !tex_coords.size())
          std::sscanf(str, "%*s %d//%d %d//%d %d//%d", &t.a, &t.n1, &t.b, &t.n2, &t.c, &t.n3);
        else if (normals.size() && tex_coords.size())
          std::sscanf(str, "%*s %d/%d/%d %d/%d/%d %d/%d/%d", &t.a, &t.t1, &t.n1, &t.b, &t.t2, &t.n2, &t.c, &t.t3, &t.n3);
        else if (!normals.size() && tex_coords.size())
          std::sscanf(str, "%*s %d/%d %d/%d %d/%d", &t.a, &t.t1, &t.b, &t.t2, &t.c, &t.t3);
        else
          std::sscanf(str, "%*s %d %d %d", &t.a, &t.b, &t.c);
        tris.push_back(t);
        if (normals.size()) {
          auto ok = [](int i, size_t n) { return i >= 1 && size_t(i) <= n; };   // the reference indexes unchecked
          if (ok(t.a, v.size()) && ok(t.n1, normals.size())) v[t.a - 1].set_normal(normals[t.n1 - 1]);
          if (ok(t.b, v.size()) && ok(t.n2, normals.size())) v[t.b - 1].set_normal(normals[t.n2 - 1]);
          if (ok(t.c, v.size()) && ok(t.n3, normals.size())) v[t.c - 1].set_normal(normals[t.n3 - 1]);
        }
      } else if (std::strcmp(ch, "mtllib") == 0) {
        mtls.push_back(line.size() > 7 ? line.substr(7) : std::string());
      }
    }
    f.close();

    if (tris.size() == 0) {
      // vertex and normal lists but no face: the i-th normal belongs to the i-th vertex
      if (v.size() == normals.size())
        for (size_t i = 0; i < v.size(); ++i) v[i].set_normal(normals[i]);
    } else if (!normals.empty()) {
      // normals came through the faces: rebuild the array one-to-one with the vertices
      normals.clear();
      normals.reserve(v.size());
      for (size_t i = 0; i != v.size(); ++i) normals.push_back(v[i].normal());
    }

    if (mtls.size()) {
      std::ifstream m(mtls[0].c_str(), std::ios::in);
      std::string token, img_name;
      while (m >> token) {
        if (token == "map_Kd") {
          m >> img_name;
          std::cerr << "OpenCV is required to load material textures. Skipping " << img_name.c_str() << std::endl;
        }
      }
    }
    return v.size() != 0;
  }

  // ---------------------------------------------------------------- PTX (io.cc:83-136)
  inline bool ReadPtx(const char* filename, std::vector<Point3D>& vertex) {
    std::ifstream f(filename, std::ios::in);
    if (!f || f.fail()) {
      std::cerr << "(PTX) error opening file" << std::endl;
      return false;
    }
    std::string line;
    int rows = 0, cols = 0;
    { std::getline(f, line); std::stringstream ss(line); ss >> cols; }
    { std::getline(f, line); std::stringstream ss(line); ss >> rows; }
    const long long numOfVertices = (long long)cols * rows;
    for (int i = 0; i < 8; i++) std::getline(f, line);   // scanner pose matrices: ignored like the reference
    vertex.clear();
    if (numOfVertices > 0) vertex.reserve(size_t(numOfVertices));
    Point3D ptx;
    float intensity = 0.f;
    Vec3 rgb = vec3(0.f, 0.f, 0.f);
    for (long long i = 0; i < numOfVertices && !f.eof(); i++) {
      std::getline(f, line);
      std::stringstream ss(line);
      ss >> ptx.x();
      ss >> ptx.y();
      ss >> ptx.z();
      ss >> intensity;
      ss >> rgb(0);
      ss >> rgb(1);
      ss >> rgb(2);
      ptx.set_rgb(rgb);
      vertex.push_back(ptx);
    }
    return (long long)vertex.size() == numOfVertices;
  }

  // ---------------------------------------------------------------- PLY (io_ply.h)
  enum PLYFormat { BINARY_BIG_ENDIAN_1, BINARY_LITTLE_ENDIAN_1, ASCII_1 };

  // io_ply.h:20-124: returns the offset of the first body byte (0 on error)
  static inline unsigned int readPlyHeader(const char* filename, unsigned int& numOfVertices, unsigned int& numOfFaces,
                                           PLYFormat& format, unsigned int& numOfVertexProperties, bool& haveColor) {
    std::ifstream in(filename, std::ios_base::in | std::ios_base::binary);
    if (!in) {
      std::cerr << "(PLY) error opening file" << std::endl;
      return 0;
    }
    numOfVertexProperties = 0; numOfVertices = 0; numOfFaces = 0; haveColor = false;
    format = ASCII_1;
    std::string current, currentelement;
    in >> current;
    if (current != "ply") {
      std::cerr << "(PLY) not a PLY file" << std::endl;
      return 0;
    }
    in >> current;
    while (current != "end_header") {
      if (!in) {
        std::cerr << "(PLY) error parsing header (no end_header)" << std::endl;   // the reference loops forever here
        return 0;
      }
      if (current == "format") {
        in >> current;
        const std::string kind = current;
        in >> current;
        if (kind != "binary_big_endian" && kind != "binary_little_endian" && kind != "ascii") {
          std::cerr << "(PLY) error parsing header (format)" << std::endl;
          return 0;
        }
        if (current != "1.0") {
          std::cerr << "(PLY) error parsing header - bad version" << std::endl;
          return 0;
        }
        format = kind == "ascii" ? ASCII_1 : (kind == "binary_big_endian" ? BINARY_BIG_ENDIAN_1 : BINARY_LITTLE_ENDIAN_1);
      } else if (current == "element") {
        in >> current;
        if (current == "vertex") { currentelement = current; in >> numOfVertices; }
        else if (current == "face") { currentelement = current; in >> numOfFaces; }
        else { std::cerr << "(PLY) ignoring unknown element " << current << std::endl; currentelement = ""; }
      } else if (currentelement != "" && current == "property") {
        in >> current;
        if (current == "float" || current == "double") { numOfVertexProperties++; in >> current; }
        else if (current == "uchar") { numOfVertexProperties++; haveColor = true; in >> current; }
        else if (current == "list") { in >> current; in >> current; in >> current; }
        else {
          std::cerr << "(PLY) error parsing header (property)" << std::endl;
          return 0;
        }
      } else if (current == "comment" || current.find("obj_info") != std::string::npos) {
        std::string rest;
        std::getline(in, rest);
      }
      in >> current;
    }
    const unsigned int headerSize = (unsigned int)in.tellg();
    return headerSize + 1;       // the byte after the end-of-line that follows "end_header"
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
