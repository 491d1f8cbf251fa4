"""TEST INFRASTRUCTURE: CPU restatement of the pose-graph file formats of include/karto_hip.h (SURVEY.md section
8f-3).  The reference's own on-disk form is a Boost binary archive of the whole Mapper (Mapper.cpp:2635-2651,
serialization.hpp:38-82), unreadable without Boost -- so there is no reference golden file for this row; what is
pinned is (a) g2o's published SE2 record layout (VERTEX_SE2 id x y theta / EDGE_SE2 a b dx dy dtheta + the
upper triangle of the 3x3 information, row-major) and (b) how the plugin turns a LinkInfo covariance into that
information matrix (ceres_solver.cpp:364-375 via oracle/spa.py).  Only tests/ may import this module."""
import struct

import numpy as np

from . import spa

MAGIC = b"KHPG\x01\x00\x00\x00"


def information_upper(cov):
    """Upper triangle (00 01 02 11 12 22) of the symmetrised inverse covariance (ceres_solver.cpp:364-375)."""
    p = spa.matrix3_inverse(np.asarray(cov, dtype=np.float64).reshape(3, 3))
    return np.array([p[0, 0], p[0, 1], p[0, 2], p[1, 1], p[1, 2], p[2, 2]])


def write_text(path, ids, poses, edges, z, info, fix=True):
    with open(path, "w") as f:
        f.write("# kartohip pose graph: g2o SE2 records, nodes in AddNode order, the first one is the gauge\n")
        for i, p in zip(ids, poses):
            f.write("VERTEX_SE2 %d %s %s %s\n" % (i, *[repr(float(v)) for v in p]))
        if fix and len(ids):
            f.write("FIX %d\n" % ids[0])
        for (a, b), zz, w in zip(edges, z, info):
            f.write("EDGE_SE2 %d %d %s\n" % (a, b, " ".join(repr(float(v)) for v in list(zz) + list(w))))


def read_text(path):
    ids, poses, edges, z, info, fix = [], [], [], [], [], None
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "VERTEX_SE2":
                assert len(t) == 5
                ids.append(int(t[1])); poses.append([float(v) for v in t[2:5]])
            elif t[0] == "EDGE_SE2":
                assert len(t) == 12
                edges.append([int(t[1]), int(t[2])]); z.append([float(v) for v in t[3:6]]); info.append([float(v) for v in t[6:12]])
            elif t[0] == "FIX":
                fix = int(t[1])
            else:
                raise ValueError("unsupported record " + t[0])
    return dict(ids=np.asarray(ids, dtype=np.int32), poses=np.asarray(poses).reshape(-1, 3),
                edges=np.asarray(edges, dtype=np.int32).reshape(-1, 2), z=np.asarray(z).reshape(-1, 3),
                info=np.asarray(info).reshape(-1, 6), fix=fix)


def write_binary(path, ids, poses, edges, z, info):
    edges = np.asarray(edges, dtype="<i4").reshape(-1, 2)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<qq", len(ids), len(edges)))
        f.write(np.asarray(ids, dtype="<i4").tobytes())
        f.write(np.asarray(poses, dtype="<f8").reshape(-1, 3).tobytes())
        f.write(np.ascontiguousarray(edges[:, 0]).tobytes())
        f.write(np.ascontiguousarray(edges[:, 1]).tobytes())
        f.write(np.asarray(z, dtype="<f8").reshape(-1, 3).tobytes())
        f.write(np.asarray(info, dtype="<f8").reshape(-1, 6).tobytes())


def read_binary(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == MAGIC
    n, m = struct.unpack("<qq", data[8:24])
    off = 24
    ids = np.frombuffer(data, "<i4", n, off); off += 4 * n
    poses = np.frombuffer(data, "<f8", 3 * n, off).reshape(n, 3); off += 24 * n
    a = np.frombuffer(data, "<i4", m, off); off += 4 * m
    b = np.frombuffer(data, "<i4", m, off); off += 4 * m
    z = np.frombuffer(data, "<f8", 3 * m, off).reshape(m, 3); off += 24 * m
    info = np.frombuffer(data, "<f8", 6 * m, off).reshape(m, 6); off += 48 * m
    assert off == len(data)
    return dict(ids=ids, poses=poses, edges=np.stack([a, b], axis=1), z=z, info=info, fix=None)
