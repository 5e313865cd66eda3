"""Minimal PLY reader / writer: the subset of include/pcp/io/ply.hpp the hot path's inputs use
(SURVEY.md section 8f-1).  `element vertex N` with float/double x,y,z; optional separate
`element normal M` block with nx,ny,nz (ply.hpp:239-250); ascii, binary_little_endian and
binary_big_endian.  Like the reference's binary reader (ply.hpp:741-764) vertex records are read as
3 x 4-byte floats.  Failures return empty arrays instead of raising (ply.hpp:111-123).
"""
import numpy as np


def read_ply(path):
    empty = (np.empty((0, 3), np.float32), np.empty((0, 3), np.float32))
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return empty
    end = data.find(b"end_header")
    if end < 0 or not data.startswith(b"ply"):
        return empty
    nl = data.find(b"\n", end)
    header = data[:end].decode("ascii", "replace").splitlines()
    body = data[nl + 1:]
    fmt = None
    counts = []
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format" and len(tok) >= 2:
            fmt = tok[1]
        elif tok[0] == "element" and len(tok) == 3:
            counts.append((tok[1], int(tok[2])))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        return empty
    nv = dict(counts).get("vertex", 0)
    nn = dict(counts).get("normal", 0)
    if fmt == "ascii":
        vals = np.array(body.split(), dtype=np.float32)
        if len(vals) < 3 * (nv + nn):
            return empty
        pts = vals[: 3 * nv].reshape(nv, 3)
        nrm = vals[3 * nv: 3 * (nv + nn)].reshape(nn, 3)
        return pts.copy(), nrm.copy()
    dt = np.dtype("<f4") if fmt == "binary_little_endian" else np.dtype(">f4")
    if len(body) < 12 * (nv + nn):
        return empty
    arr = np.frombuffer(body, dtype=dt, count=3 * (nv + nn)).astype(np.float32)
    return arr[: 3 * nv].reshape(nv, 3).copy(), arr[3 * nv:].reshape(nn, 3).copy()


def write_ply(path, points, normals=None, fmt="binary_little_endian"):
    points = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    hdr = ["ply", "format %s 1.0" % fmt, "element vertex %d" % len(points), "property float x", "property float y",
           "property float z"]
    if normals is not None:
        hdr += ["element normal %d" % len(normals), "property float nx", "property float ny", "property float nz"]
    hdr.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        blocks = [points] + ([normals] if normals is not None else [])
        for b in blocks:
            if fmt == "ascii":
                for row in b:
                    f.write(("%.9g %.9g %.9g\n" % tuple(row)).encode("ascii"))
            else:
                f.write(b.astype("<f4" if fmt == "binary_little_endian" else ">f4").tobytes())
