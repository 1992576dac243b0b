"""Independent numpy PLY reader/writer used by the tests as a cross-check of the product's C++
reader (pt_ply_load) and to produce the binary model fixtures under tests/golden/models.

Handles what the reference's inputs use (Models/CBbunny.ply:1-17): vertex properties
x,y,z,nx,ny,nz (float) and red,green,blue,alpha (uchar), faces `list uchar uint vertex_indices`,
ASCII or binary_little_endian.  Polygons are fan-triangulated (v0,v1,v2),(v0,v2,v3),...
"""
import numpy as np

_NP = {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
       "short": "<i2", "ushort": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def _parse_header(f):
    assert f.readline().strip() == b"ply"
    fmt = None
    elements = []
    while True:
        line = f.readline().decode("ascii").strip()
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
            else:
                elements[-1]["props"].append((tok[1], tok[2]))
        elif tok[0] == "end_header":
            break
    return fmt, elements


def read_ply(path):
    """Returns dict(positions [V,4] f32 (w=1), normals [V,3] f32, colors [V,4] u8, indices [T,3] u32)."""
    with open(path, "rb") as f:
        fmt, elements = _parse_header(f)
        verts = faces = None
        for el in elements:
            if el["name"] == "vertex":
                names = [p[1] for p in el["props"]]
                if fmt == "ascii":
                    rows = [f.readline().split() for _ in range(el["count"])]
                    arr = np.array(rows, dtype=np.float64)
                    verts = {n: arr[:, i] for i, n in enumerate(names)}
                else:
                    dt = np.dtype([(p[1], _NP[p[0]]) for p in el["props"]])
                    arr = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt)
                    verts = {n: arr[n] for n in names}
            elif el["name"] == "face":
                tris = []
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        t = f.readline().split()
                        n = int(t[0]); v = [int(x) for x in t[1:1 + n]]
                        for k in range(1, n - 1):
                            tris.append((v[0], v[k], v[k + 1]))
                else:
                    _, ct, it, _ = el["props"][0]
                    cdt, idt = np.dtype(_NP[ct]), np.dtype(_NP[it])
                    for _ in range(el["count"]):
                        n = int(np.frombuffer(f.read(cdt.itemsize), dtype=cdt)[0])
                        v = np.frombuffer(f.read(idt.itemsize * n), dtype=idt)
                        for k in range(1, n - 1):
                            tris.append((int(v[0]), int(v[k]), int(v[k + 1])))
                faces = np.array(tris, dtype=np.uint32).reshape(-1, 3)
    V = len(verts["x"])
    pos = np.ones((V, 4), dtype=np.float32)
    pos[:, 0], pos[:, 1], pos[:, 2] = verts["x"], verts["y"], verts["z"]
    nrm = np.zeros((V, 3), dtype=np.float32)
    if "nx" in verts:
        nrm[:, 0], nrm[:, 1], nrm[:, 2] = verts["nx"], verts["ny"], verts["nz"]
    col = np.full((V, 4), 255, dtype=np.uint8)
    for i, n in enumerate(("red", "green", "blue", "alpha")):
        if n in verts:
            col[:, i] = verts[n].astype(np.uint8)
    return {"positions": pos, "normals": nrm, "colors": col, "indices": faces}


def write_ply_binary(path, positions, normals, colors, faces):
    """faces: list of index lists (polygons are kept as they are)."""
    V = positions.shape[0]
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0", "comment binary fixture", f"element vertex {V}"]
        hdr += [f"property float {n}" for n in ("x", "y", "z", "nx", "ny", "nz")]
        hdr += [f"property uchar {n}" for n in ("red", "green", "blue", "alpha")]
        hdr += [f"element face {len(faces)}", "property list uchar uint vertex_indices", "end_header"]
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        dt = np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("c", "u1", 4)])
        arr = np.zeros(V, dtype=dt)
        arr["p"], arr["n"], arr["c"] = positions[:, :3], normals, colors
        f.write(arr.tobytes())
        for face in faces:
            f.write(np.uint8(len(face)).tobytes())
            f.write(np.asarray(face, dtype="<u4").tobytes())


def pack_aux(normals, colors):
    """VertexAux records (16 B: normal f32x3, RGBA8) as a [V] structured array."""
    dt = np.dtype([("normal", "<f4", 3), ("color", "u1", 4)])
    aux = np.zeros(normals.shape[0], dtype=dt)
    aux["normal"], aux["color"] = normals, colors
    return aux
