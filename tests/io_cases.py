"""Deterministic input files for the IO parity tests (tests/test_io.py, tests/golden/make_io_golden.py).

Every OBJ ends with a comment line: on the read after the last newline the reference tests an uninitialised token
buffer (io.cc:151-153) and in practice repeats the previous line's action, which is not something to pin."""
import struct

import numpy as np


def _pts(n, seed):
    r = np.random.default_rng(seed)
    p = r.normal(size=(n, 3)).astype(np.float32)
    nrm = r.normal(size=(n, 3)).astype(np.float32)
    nrm[::7] *= 0.01                                   # some invalid (short) normals for CleanInvalidNormals
    rgb = r.integers(0, 256, size=(n, 4)).astype(np.int64)
    return p, nrm, rgb


def _faces(n, m, seed):
    r = np.random.default_rng(seed)
    return r.integers(1, n + 1, size=(m, 3))


def write_cases(d):
    """Writes all cases into directory d; returns [(name, filename)]."""
    out = []
    p, nrm, rgb = _pts(57, 1)
    f = _faces(57, 31, 2)
    tex = np.random.default_rng(3).uniform(0, 1, size=(57, 2)).astype(np.float32)

    def obj(name, lines):
        path = "%s/%s.obj" % (d, name)
        with open(path, "w") as fh:
            fh.write("# generated\n" + "\n".join(lines) + "\n# End of File\n")
        out.append((name, path))

    V = ["v %.7g %.7g %.7g" % tuple(x) for x in p]
    VN = ["vn %.7g %.7g %.7g" % tuple(x) for x in nrm]
    VT = ["vt %.7g %.7g" % tuple(x) for x in tex]
    obj("obj_points", V)
    obj("obj_points_normals", V + VN)
    obj("obj_mesh_plain", V + ["f %d %d %d" % tuple(t) for t in f])
    obj("obj_mesh_normals", V + VN + ["f %d//%d %d//%d %d//%d" % (t[0], t[0], t[1], t[1], t[2], t[2]) for t in f])
    obj("obj_mesh_tex", V + VT + ["f %d/%d %d/%d %d/%d" % (t[0], t[1], t[1], t[2], t[2], t[0]) for t in f])
    obj("obj_mesh_tex_normals", ["mtllib missing_material.mtl"] + V + VT + VN +
        ["f %d/%d/%d %d/%d/%d %d/%d/%d" % (t[0], t[1], t[2], t[1], t[2], t[0], t[2], t[0], t[1]) for t in f])
    obj("obj_crlf_and_blank_free", [l + "\r" for l in V[:9]])

    def ply(name, props, fmt, with_faces):
        # props: "xyz", "xyzn", "xyzc3" (6 + colour), "xyzc4" (7), "xyznc3" (9), "xyznc4" (10)
        path = "%s/%s.ply" % (d, name)
        hdr = ["ply", "format %s 1.0" % fmt, "comment made for the parity tests", "element vertex %d" % len(p),
               "property float x", "property float y", "property float z"]
        if "n" in props:
            hdr += ["property float nx", "property float ny", "property float nz"]
        ncol = 3 if props.endswith("c3") else (4 if props.endswith("c4") else 0)
        hdr += ["property uchar %s" % c for c in ("red", "green", "blue", "alpha")[:ncol]]
        if with_faces:
            hdr += ["element face %d" % len(f), "property list uchar int vertex_indices"]
        hdr += ["end_header"]
        with open(path, "wb") as fh:
            fh.write(("\n".join(hdr) + "\n").encode())
            e = ">" if fmt == "binary_big_endian" else "<"
            for i in range(len(p)):
                vals = list(p[i]) + (list(nrm[i]) if "n" in props else [])
                if fmt == "ascii":
                    fh.write((" ".join("%.7g" % x for x in vals) + "".join(" %d" % c for c in rgb[i][:ncol]) + "\n").encode())
                else:
                    fh.write(struct.pack(e + "%df" % len(vals), *vals) + bytes(int(c) for c in rgb[i][:ncol]))
            if with_faces:
                for t in f:
                    if fmt == "ascii":
                        fh.write(("3 %d %d %d\n" % (t[0] - 1, t[1] - 1, t[2] - 1)).encode())
                    else:
                        fh.write(struct.pack(e + "B3i", 3, int(t[0] - 1), int(t[1] - 1), int(t[2] - 1)))
        out.append((name, path))

    for fmt, tag in (("ascii", "a"), ("binary_little_endian", "le"), ("binary_big_endian", "be")):
        for props in ("xyz", "xyzn", "xyzc3", "xyzc4", "xyznc3", "xyznc4"):
            ply("ply_%s_%s" % (tag, props), props, fmt, with_faces=False)
        ply("ply_%s_xyzn_faces" % tag, "xyzn", fmt, with_faces=True)

    path = "%s/scan.ptx" % d
    with open(path, "w") as fh:
        fh.write("19\n3\n0 0 0\n1 0 0\n0 1 0\n0 0 1\n1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
        for i in range(57):
            fh.write("%.7g %.7g %.7g %.3f %d %d %d\n" % (p[i][0], p[i][1], p[i][2], 0.5, rgb[i][0], rgb[i][1], rgb[i][2]))
    out.append(("ptx_scan", path))
    return out


MATRICES = [
    [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1],
    [0.25881904, -0.96592583, 0, 1.5, 0.96592583, 0.25881904, 0, -2.25, 0, 0, 1, 1e-7, 0, 0, 0, 1],
    [-123456.789, 1e9, -1e-9, 0.1234567891, 3, -0.0, 7.5, -7.5, 1e-5, 2e-5, -3e-5, 4e15, 0, 0, 0, 1],
]
