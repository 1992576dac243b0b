"""Regenerates tests/golden/models/*.ply from the reference's model files.

The two real models of the reference (Models/CBbunny.ply, Models/CBmonkey.ply; the others are
Git-LFS pointer stubs) are data, not source: they are re-encoded as binary_little_endian PLY with
the same vertices, attributes and polygons (quads are kept as quads), which also makes them
fixtures for the binary branch of the PLY reader.  Run from the repository root in a container
that has /root/reference:  python tests/golden/make_models.py
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import plyio  # noqa: E402

REF = "/root/reference/Models"
OUT = os.path.join(os.path.dirname(__file__), "models")
EXPECT = {"CBbunny.ply": "4aa16720ba3888c2", "CBmonkey.ply": "1aa92bdd6c5de5b8"}


def convert(name):
    src = os.path.join(REF, name)
    digest = hashlib.sha256(open(src, "rb").read()).hexdigest()
    assert digest.startswith(EXPECT[name]), (name, digest)
    with open(src, "rb") as f:
        fmt, elements = plyio._parse_header(f)
        assert fmt == "ascii"
        nv = elements[0]["count"]; nf = elements[1]["count"]
        rows = np.array([f.readline().split() for _ in range(nv)], dtype=np.float64)
        faces = []
        for _ in range(nf):
            t = f.readline().split()
            faces.append([int(x) for x in t[1:1 + int(t[0])]])
    pos = rows[:, 0:3].astype(np.float32)
    nrm = rows[:, 3:6].astype(np.float32)
    col = rows[:, 6:10].astype(np.uint8)
    plyio.write_ply_binary(os.path.join(OUT, name), pos, nrm, col, faces)
    print(name, nv, "vertices", nf, "faces ->", os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in EXPECT:
        convert(n)
