"""Independent, minimal Python restatement of the reference's scene ingest (main.cpp:28-58)
used by the tests as the checker for the C++ host loader. Test infrastructure only.

Semantics restated (tinyobjloader is not vendored in /root/reference):
 * `v x y z`, `f a b c d ...` with 1-based or negative (relative) indices, `a/b/c` forms;
 * polygons are fan-triangulated (0,1,2),(0,2,3),...; for this OBJ the shorter-diagonal
   rule of newer tinyobjloader gives the same surface, only other prim ids;
 * `usemtl` selects the material id of following faces; `mtllib` + `newmtl`/`Kd`/`Ke`;
 * per mesh index: push (x, -y, z) and a running index (main.cpp:40-45);
 * per face: push {Kd, Ke} of its material (main.cpp:47-56).
"""
import os

import numpy as np


def load_mtl(path):
    mats, names, cur = [], {}, None
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].split()
            if not line:
                continue
            if line[0] == "newmtl":
                cur = {"Kd": [0.0, 0.0, 0.0], "Ke": [0.0, 0.0, 0.0]}  # tinyobj InitMaterial: all zero
                names[line[1]] = len(mats)
                mats.append(cur)
            elif line[0] in ("Kd", "Ke") and cur is not None:
                cur[line[0]] = ([float(x) for x in line[1:4]] + [0.0, 0.0, 0.0])[:3]  # parseReal3: missing -> 0
    return mats, names


def load_obj(path):
    base = os.path.dirname(path)
    verts, tris, tri_mat = [], [], []
    mats, names, mat = [], {}, -1
    with open(path) as f:
        for line in f:
            tok = line.split("#", 1)[0].split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == "mtllib":
                mats, names = load_mtl(os.path.join(base, tok[1]))
            elif tok[0] == "usemtl":
                mat = names.get(tok[1], -1)
            elif tok[0] == "f":
                idx = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append((idx[0], idx[k], idx[k + 1]))
                    tri_mat.append(mat)
    v = np.array(verts, dtype=np.float32)
    out_v = np.zeros((3 * len(tris), 3), dtype=np.float32)
    for t, tri in enumerate(tris):
        for c in range(3):
            x, y, z = v[tri[c]]
            out_v[3 * t + c] = (x, -y, z)
    out_i = np.arange(3 * len(tris), dtype=np.uint32)
    out_f = np.zeros((len(tris), 6), dtype=np.float32)
    for t, m in enumerate(tri_mat):
        if m >= 0:
            out_f[t, :3] = mats[m]["Kd"]
            out_f[t, 3:] = mats[m]["Ke"]
        else:
            out_f[t, :3] = (0.6, 0.6, 0.6)
    return out_v.reshape(-1), out_i, out_f.reshape(-1)


class ObjError(Exception):
    pass


def load_obj_strict(path, quad_shorter_diagonal=False):
    """The host loader's full contract, line by line (one reader, file order): several `mtllib` lines ADD materials and a later
    `newmtl` of a known name overrides it for the `usemtl` lines after that library; errors carry the loader's text and line
    number, checked token by token.  What test_loader_chunked_text_equals_one_reader holds the multi-threaded loader against."""
    base = os.path.dirname(path)
    verts, tris, tri_mat = [], [], []
    mats, names, mat = [], {}, -1
    with open(path, newline="") as f:
        text = f.read()
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for no, line in enumerate(lines, 1):
        tok = line.split("#", 1)[0].split()
        if not tok:
            continue
        if tok[0] == "v":
            try:
                verts.append([np.float32(float(x)) for x in (tok[1:4] + [None, None, None])[:3]])
            except (TypeError, ValueError):
                raise ObjError(f"bad vertex at line {no}")
        elif tok[0] == "mtllib":
            try:
                m2, n2 = load_mtl(os.path.join(base, tok[1] if len(tok) > 1 else ""))
            except OSError:
                continue
            for k, v in n2.items():
                names[k] = len(mats) + v
            mats += m2
        elif tok[0] == "usemtl":
            mat = names.get(tok[1] if len(tok) > 1 else "", -1)
        elif tok[0] == "f":
            idx = []
            for t in tok[1:]:
                try:
                    i = int(t.split("/")[0])
                except ValueError:
                    raise ObjError(f"bad face at line {no}")
                vi = i - 1 if i > 0 else len(verts) + i
                if i == 0 or vi < 0 or vi >= len(verts):
                    raise ObjError(f"face index out of range at line {no}")
                idx.append(vi)
            if len(idx) < 3:
                raise ObjError(f"face with < 3 vertices at line {no}")
            if len(idx) == 4 and quad_shorter_diagonal:
                def d2(a, b):
                    s2 = np.float32(0)
                    for c in range(3):
                        d = np.float32(verts[a][c] - verts[b][c])
                        s2 = np.float32(s2 + np.float32(d * d))
                    return s2
                order = (0, 1, 2, 0, 2, 3) if d2(idx[0], idx[2]) < d2(idx[1], idx[3]) else (0, 1, 3, 1, 2, 3)
                tris += [tuple(idx[k] for k in order[:3]), tuple(idx[k] for k in order[3:])]
                tri_mat += [mat, mat]
                continue
            for k in range(1, len(idx) - 1):
                tris.append((idx[0], idx[k], idx[k + 1]))
                tri_mat.append(mat)
    if not tris:
        raise ObjError("no faces in OBJ")
    v = np.array(verts, dtype=np.float32).reshape(-1, 3)
    t = np.array(tris, dtype=np.int64).reshape(-1)
    out_v = v[t] * np.float32([1, -1, 1])
    out_f = np.zeros((len(tris), 6), dtype=np.float32)
    for k, m in enumerate(tri_mat):
        if m >= 0:
            out_f[k, :3] = mats[m]["Kd"]
            out_f[k, 3:] = mats[m]["Ke"]
        else:
            out_f[k, :3] = (0.6, 0.6, 0.6)
    return out_v.astype(np.float32).reshape(-1), np.arange(3 * len(tris), dtype=np.uint32), out_f.reshape(-1)
