"""Offline model compiler: MJCF subset + STL meshes -> flat `.ksm` model blob.

Replaces what `mujoco_py.load_model_from_path` does for the reference at
kinova_gripper_env.py:62,102,878,1002 (reference: gym-kinova-gripper/gym_kinova_gripper/envs/).
Only the MJCF features used by `kinova_description/j2s7s300_end_effector_v1_*.xml`
are understood: compiler angle=radian, option timestep/impratio, default geom margin,
default joint damping/armature, nested bodies (pos/quat/euler), inertial, slide/hinge/free
joints, mesh geoms, plane geom, sites, explicit contact pairs, fixed tendons + tendon
equalities, velocity/motor actuators, jointpos/rangefinder sensors.

The topology is fixed (Kinova j2s7s300 end effector: 3 slides on link_7, 3 fingers x
(proximal, distal) hinges, one free object, ground plane); every numeric parameter is read
from the XML / STL.  Compiler-derived quantities (not in the XML):
per-mesh convex hull, centroid, principal frame, AABB half extents, bounding radius,
hull face planes, object inertia from mesh at the given mass, dof/body/tendon inverse
weights at qpos0.

This module runs in the authoring container (it needs the MJCF/STL assets); the compiled
blobs are committed under kinovagrasping_amd/assets/ and are what travels to the GPU box.
"""
from __future__ import annotations

import os
import re
import struct
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

MAGIC = b"KSMB"
VERSION = 5

# fixed topology ------------------------------------------------------------------------------
NQ, NV, NU = 16, 15, 9
NBODY, NGEOM, NSITE = 10, 9, 17
NGEOM_MAX, NPAIR_MAX_MG = 17, 96     # multi-geom objects: up to 8 welded pieces beside `object` (RoundBowl), 30 + 8 * 8 pairs
BODY_NAMES = ["world", "root", "j2s7s300_link_7",
              "j2s7s300_link_finger_1", "j2s7s300_link_finger_tip_1",
              "j2s7s300_link_finger_2", "j2s7s300_link_finger_tip_2",
              "j2s7s300_link_finger_3", "j2s7s300_link_finger_tip_3", "object"]
GEOM_NAMES = ["ground", "palm", "f1_prox", "f1_dist", "f2_prox", "f2_dist", "f3_prox", "f3_dist", "object"]
GEOM_BODY = [0, 2, 3, 4, 5, 6, 7, 8, 9]
GEOM_MESH = [-1, 0, 1, 2, 1, 2, 1, 2, 3]   # mesh slot: 0 palm, 1 proximal, 2 distal, 3 object
# the two debug bar sites on `root` (XML:55-56, absent from some siblings) are never read -> dropped
SITE_NAMES = ["palm", "palm_1", "palm_2", "palm_3", "palm_4",
              "f1_prox", "f1_prox_1", "f1_dist", "f1_dist_1",
              "f2_prox", "f2_prox_1", "f2_dist", "f2_dist_1",
              "f3_prox", "f3_prox_1", "f3_dist", "f3_dist_1"]


# ---------------------------------------------------------------------------------------------
# small math helpers
def quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                     w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def quat_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mat_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def euler_to_quat(e):
    """MJCF euler, default eulerseq 'xyz' (intrinsic): R = Rx(a) Ry(b) Rz(c)  (SURVEY App. B.10)."""
    a, b, c = e
    qx = np.array([np.cos(a / 2), np.sin(a / 2), 0, 0])
    qy = np.array([np.cos(b / 2), 0, np.sin(b / 2), 0])
    qz = np.array([np.cos(c / 2), 0, 0, np.sin(c / 2)])
    return quat_mul(quat_mul(qx, qy), qz)


def truncated_euler(values):
    """The reference patches hand eulers into the XML as str(v)[:5] (kinova_gripper_env.py:870-874).

    SURVEY note N4: ordinary values are truncated to a 5-character decimal string; for
    |v| < 1e-4 numpy prints scientific notation and the slice is nonsense, so we map those to 0.
    """
    out = []
    for v in values:
        v = float(v)
        if abs(v) < 1e-4:
            out.append(0.0)
            continue
        s = repr(v)
        if "e" in s or "E" in s:
            s = f"{v:.10f}"
        out.append(float(s[:5]))
    return np.array(out)


# ---------------------------------------------------------------------------------------------
# STL + mesh processing
def load_stl(path: Path) -> np.ndarray:
    """Returns triangles [n,3,3] float64 (vertices are float32 in binary STL)."""
    raw = path.read_bytes()
    n = struct.unpack("<I", raw[80:84])[0] if len(raw) >= 84 else -1
    if n >= 0 and len(raw) == 84 + 50 * n:
        rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
        return np.frombuffer(raw[84:], dtype=rec)["v"].astype(np.float64)
    verts = []
    for line in raw.decode("latin1").splitlines():
        s = line.split()
        if s and s[0] == "vertex":
            verts.append([float(x) for x in s[1:4]])
    return np.array(verts, dtype=np.float64).reshape(-1, 3, 3)


def mesh_mass_properties(tri: np.ndarray):
    """Exact signed-volume integration over the triangle soup (unit density).

    Returns (volume, com[3], inertia_about_com[3,3]).  Inward-wound meshes (negative signed
    volume: CylinderS.stl, Vase1S.stl, ... SURVEY note N3) are flipped.
    """
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    d6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    vol = d6.sum() / 6.0
    sign = 1.0 if vol >= 0 else -1.0
    d6 = d6 * sign
    vol = abs(vol)
    com = ((a + b + c) * d6[:, None]).sum(0) / 24.0 / vol
    # second moments of each tetra (origin, a, b, c): integral x_i x_j dV =
    #   d6/120 * (sum_k v_k,i v_k,j + s_i s_j), s = a+b+c
    s = a + b + c
    C = (np.einsum("n,ni,nj->ij", d6, a, a) + np.einsum("n,ni,nj->ij", d6, b, b)
         + np.einsum("n,ni,nj->ij", d6, c, c) + np.einsum("n,ni,nj->ij", d6, s, s)) / 120.0
    C = C - vol * np.outer(com, com)          # covariance about com
    inertia = np.trace(C) * np.eye(3) - C
    return vol, com, inertia


def mesh_mass_properties_legacy(tri: np.ndarray):
    """MuJoCo <= 2.1's mesh inertia (the algorithm of the reference's MuJoCo 1.50; later kept as inertia="legacy"):
    pass 1 - centre of mass from pyramids whose apex is the AREA-WEIGHTED CENTRE OF THE FACES, every pyramid's volume taken
    ABSOLUTE (a non-convex mesh is over-counted: hand_3finger.STL 5.703e-4 m^3 against the exact 5.531e-4); pass 2 - second
    moments from pyramids whose apex is that centre of mass, volumes absolute again.  Winding does not matter.
    Pinned by real MuJoCo 1.50 output: row 0 of the reference's Old Code/Pose_file_2.csv holds geom_xpos of the palm and the
    six finger links at qpos0 (tests/test_mujoco_recorded.py); the exact integration misses the palm by 1.25 mm.
    Returns (volume, com[3], inertia_about_com[3,3]), unit density."""
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    n = np.cross(b - a, c - a)
    area = 0.5 * np.linalg.norm(n, axis=1)
    ok = area > 0
    nrm = np.zeros_like(n)
    nrm[ok] = n[ok] / (2 * area[ok, None])
    cen = (a + b + c) / 3.0
    facecen = (area[:, None] * cen).sum(0) / area.sum()
    vol = np.abs(np.einsum("ij,ij->i", cen - facecen, nrm) * area / 3.0)
    com = (vol[:, None] * (0.75 * cen + 0.25 * facecen)).sum(0) / vol.sum()
    D, E, F = a - com, b - com, c - com
    vol = np.abs(np.einsum("ij,ij->i", (D + E + F) / 3.0, nrm) * area / 3.0)
    P = (np.einsum("n,ni,nj->ij", vol, D, D) + np.einsum("n,ni,nj->ij", vol, E, E) + np.einsum("n,ni,nj->ij", vol, F, F)) * 2.0
    for X, Y in ((D, E), (D, F), (E, F)):
        P += np.einsum("n,ni,nj->ij", vol, X, Y) + np.einsum("n,ni,nj->ij", vol, Y, X)
    P /= 20.0
    inertia = np.trace(P) * np.eye(3) - P
    return float(vol.sum()), com, inertia


def principal_frame(I: np.ndarray):
    """Jacobi diagonalisation started from identity, then eigenvalues sorted descending by
    90-degree axis swaps (right-handed).  Mirrors the convention described for MuJoCo's
    mju_eig3 in SURVEY hard-part 6: a nearly diagonal, already-descending tensor (the palm)
    yields the nearby small rotation, a degenerate one (square prisms) yields identity.
    """
    A = I.copy()
    R = np.eye(3)
    scale = np.abs(np.diag(A)).max()
    for _ in range(100):
        off = [(abs(A[0, 1]), 0, 1), (abs(A[0, 2]), 0, 2), (abs(A[1, 2]), 1, 2)]
        m, p, q = max(off)
        if m < 1e-12 * scale:
            break
        theta = 0.5 * np.arctan2(2 * A[p, q], A[q, q] - A[p, p])
        # choose the small rotation |theta| <= pi/4
        if theta > np.pi / 4:
            theta -= np.pi / 2
        if theta < -np.pi / 4:
            theta += np.pi / 2
        G = np.eye(3)
        cth, sth = np.cos(theta), np.sin(theta)
        G[p, p], G[q, q], G[p, q], G[q, p] = cth, cth, sth, -sth
        A = G.T @ A @ G
        R = R @ G
    ev = np.diag(A).copy()
    # sort descending with right-handed 90 degree swaps: swapping axes i,j -> new_i = old_j, new_j = -old_i
    for _ in range(3):
        for i, j in ((0, 1), (1, 2)):
            if ev[i] < ev[j] * (1 - 1e-9):
                ev[i], ev[j] = ev[j], ev[i]
                ci, cj = R[:, i].copy(), R[:, j].copy()
                R[:, i], R[:, j] = cj, -ci
    assert np.linalg.det(R) > 0.999
    return ev, R


def build_bvh(tri: np.ndarray, leaf: int = 4):
    """Median-split BVH over triangles [n,3,3] (float32).  Returns (tri_reordered [n,9], box [nnode,6],
    lr [nnode,2] int32): internal node -> (left, right) child ids; leaf -> (first triangle, -count)."""
    cen = tri.mean(1)
    order = np.arange(len(tri))
    boxes, lr = [], []

    def rec(lo, hi):
        idx = order[lo:hi]
        pts = tri[idx].reshape(-1, 3)
        node = len(boxes)
        boxes.append(np.concatenate([pts.min(0), pts.max(0)]))
        lr.append([0, 0])
        if hi - lo <= leaf:
            lr[node] = [lo, -(hi - lo)]
            return node
        ext = cen[idx].max(0) - cen[idx].min(0)
        ax = int(np.argmax(ext))
        sub = idx[np.argsort(cen[idx, ax], kind="stable")]
        order[lo:hi] = sub
        mid = (lo + hi) // 2
        left = rec(lo, mid)
        right = rec(mid, hi)
        lr[node] = [left, right]
        return node

    import sys
    sys.setrecursionlimit(10000)
    rec(0, len(tri))
    return tri[order].reshape(-1, 9).astype(np.float32), np.array(boxes, dtype=np.float32), np.array(lr, dtype=np.int32)


def box_triangles(hx, hy, hz):
    """the 12 triangles of a box with half extents (hx, hy, hz), outward winding"""
    c = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])   # index = 4 ix + 2 iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]         # -x +x -y +y -z +z
    return np.array([[c[a], c[b], c[d]] for a, b, d, e in quads] + [[c[a], c[d], c[e]] for a, b, d, e in quads])


def cylinder_triangles(r, h, n=64):
    """an n-gon prism (radius r to the vertices, half height h, axis z): side quads + cap fans, outward winding"""
    ang = 2 * np.pi * np.arange(n) / n
    ring = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
    tri = []
    for i in range(n):
        j = (i + 1) % n
        a, b = ring[i], ring[j]
        tri += [[[a[0], a[1], -h], [b[0], b[1], -h], [b[0], b[1], h]], [[a[0], a[1], -h], [b[0], b[1], h], [a[0], a[1], h]],
                [[0, 0, h], [a[0], a[1], h], [b[0], b[1], h]], [[0, 0, -h], [b[0], b[1], -h], [a[0], a[1], -h]]]
    return np.array(tri, dtype=np.float64)


class CompiledMesh:
    def __init__(self, tri: np.ndarray, name: str, keep_frame: bool = False, mesh_inertia: str = "legacy"):
        """mesh_inertia: "legacy" = MuJoCo 1.50's pyramid sums (the reference's engine; default), "exact" = signed-volume integration.
        keep_frame: the triangles already are in the geom frame (primitive geoms: MuJoCo keeps the user frame of a box /
        cylinder, only meshes are re-centred on their inertial frame)"""
        from scipy.spatial import ConvexHull
        self.name = name
        self.ntri = len(tri)
        assert mesh_inertia in ("legacy", "exact")
        self.volume, self.com, inertia = (mesh_mass_properties_legacy if mesh_inertia == "legacy" and not keep_frame else mesh_mass_properties)(tri)
        self.principal, self.R = principal_frame(inertia)        # unit-density moments
        if keep_frame:
            self.com, self.R, self.principal = np.zeros(3), np.eye(3), np.diag(inertia).copy()
        self.quat = mat_to_quat(self.R)
        pts = np.unique(tri.reshape(-1, 3), axis=0)
        # points in the geom frame (origin = com, axes = principal frame).  MuJoCo keeps a mesh's vertices as float32 (mjModel.mesh_vert),
        # re-centred on the mesh's inertial frame by its compiler: the geom-frame vertices every convex query and plane scan sees are float32
        # numbers, and the hull (its qhull graph) is the hull of THOSE.  Measured on the recorded MuJoCo 1.50 trajectory
        # (tests/test_mujoco_recorded.py): rows 5-40 agree to 1.05e-10 - the recording's resolution - with float32 vertices, 1.9e-10
        # without.  Primitive geoms (keep_frame) stay analytic.
        allv = (pts - self.com) @ self.R
        if not keep_frame:
            allv = allv.astype(np.float32).astype(np.float64)                    # (vertex order kept: ties go to the lowest index)
        self.verts = allv[np.sort(ConvexHull(allv).vertices)]
        h2 = ConvexHull(self.verts)
        while len(h2.vertices) < len(self.verts):           # (points qhull merged into a facet of the rounded set: not vertices of its graph)
            self.verts = self.verts[np.sort(h2.vertices)]
            h2 = ConvexHull(self.verts)
        self.size = np.maximum(np.abs(allv.min(0)), np.abs(allv.max(0)))
        self.rbound = float(np.linalg.norm(self.verts, axis=1).max())
        # hull faces in geom frame: merge coplanar simplices into unique planes n.x <= d
        eq = h2.equations                           # n.x + off <= 0 inside
        planes = np.concatenate([eq[:, :3], -eq[:, 3:4]], axis=1)
        key = np.round(planes / 1e-7).astype(np.int64)
        _, idx = np.unique(key, axis=0, return_index=True)
        self.planes = planes[np.sort(idx)]
        self.nsimplex = len(h2.simplices)
        # rangefinder geometry: the ORIGINAL triangles (MuJoCo ray-casts mesh faces, not the hull) in the geom
        # frame, reordered so that the leaves of a median-split bounding-volume hierarchy are contiguous
        tri_g = ((tri - self.com) @ self.R).astype(np.float32)
        area = np.linalg.norm(np.cross(tri_g[:, 1] - tri_g[:, 0], tri_g[:, 2] - tri_g[:, 0]), axis=1)
        tri_g = tri_g[area > 0]                      # zero-area triangles cannot be hit (Vase1S has 246)
        self.tri, self.bvh_box, self.bvh_lr = build_bvh(tri_g)
        # vertex adjacency of the (triangulated) hull in CSR form, neighbours in ascending index order:
        # lets the GPU do hill-climbing support queries instead of scanning every vertex
        nbr = [set() for _ in range(len(self.verts))]
        for tri_ in h2.simplices:
            for a_ in tri_:
                for b_ in tri_:
                    if a_ != b_:
                        nbr[a_].add(int(b_))
        assert all(len(x) >= 3 for x in nbr)
        self.adj_off = np.cumsum([0] + [len(x) for x in nbr]).astype(np.int32)
        self.adj = np.concatenate([sorted(x) for x in nbr]).astype(np.int32)


# ---------------------------------------------------------------------------------------------
def _floats(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None:
        assert len(v) == n, (s, n)
    return v


def _frame_of(elem):
    pos = _floats(elem.get("pos", "0 0 0"), 3)
    if elem.get("quat") is not None:
        q = _floats(elem.get("quat"), 4)
        q = q / np.linalg.norm(q)
    elif elem.get("euler") is not None:
        q = euler_to_quat(_floats(elem.get("euler"), 3))
    else:
        q = np.array([1.0, 0, 0, 0])
    return pos, q


def compile_model(xml_path: Path, mesh_inertia: str = "legacy") -> dict:
    """mesh_inertia: how mesh geoms get their centre / principal frame / (object) inertia: "legacy" reproduces MuJoCo 1.50
    (default: parity with the reference's engine), "exact" is the signed-volume integration of MuJoCo >= 2.2's default."""
    xml_path = Path(xml_path)
    text = xml_path.read_text()
    end = text.find("</mujoco>")
    # (one of the reference's files, ..._v1_mhg.xml, carries a stray "oco>" behind its closing tag: everything behind the root element is dropped)
    root = ET.fromstring(text[:end + len("</mujoco>")] if end >= 0 else text)
    comp = root.find("compiler")
    assert comp.get("angle") == "radian"
    meshdir = xml_path.parent / comp.get("meshdir", "")
    opt = root.find("option")
    dt = float(opt.get("timestep"))
    impratio = float(opt.get("impratio", "1"))
    dflt = root.find("default")
    margin = float(dflt.find("geom").get("margin", "0"))
    dj = dflt.find("joint")
    damping = float(dj.get("damping", "0"))
    armature = float(dj.get("armature", "0"))
    assert dj.get("limited", "false") == "false"

    mesh_assets = {m.get("name"): m for m in root.find("asset").findall("mesh")}

    def get_mesh(name):
        m = mesh_assets[name]
        tri = load_stl(meshdir / m.get("file"))
        sc = _floats(m.get("scale", "1 1 1"), 3)
        return CompiledMesh(tri * sc, name, mesh_inertia=mesh_inertia)

    wb = root.find("worldbody")
    bodies = {b.get("name"): b for b in wb.iter("body")}
    parent = {c: p for p in wb.iter() for c in p}

    M = {}
    M["opt"] = np.array([dt, impratio, -9.81, margin, 0.02, 1.0, 0.9, 0.95, 0.001, 1e-6, 50.0])
    # opt layout: dt, impratio, gravity_z, margin, solref[2], solimp[3], mpr_tolerance, mpr_iterations

    # bodies ----------------------------------------------------------------------------------
    body_pos = np.zeros((NBODY, 3))
    body_quat = np.tile(np.array([1.0, 0, 0, 0]), (NBODY, 1))
    body_mass = np.zeros(NBODY)
    body_ipos = np.zeros((NBODY, 3))
    body_iquat = np.tile(np.array([1.0, 0, 0, 0]), (NBODY, 1))
    body_inertia = np.zeros((NBODY, 3))
    for i, name in enumerate(BODY_NAMES):
        if i == 0:
            continue
        b = bodies[name]
        body_pos[i], body_quat[i] = _frame_of(b)
        ine = b.find("inertial")
        if ine is not None:
            body_mass[i] = float(ine.get("mass"))
            body_ipos[i] = _floats(ine.get("pos"), 3)
            body_inertia[i] = _floats(ine.get("diaginertia"), 3)
            assert ine.get("quat") is None
    assert np.allclose(body_pos[1], 0) and np.allclose(body_quat[1], [1, 0, 0, 0])
    assert parent[bodies["object"]] is wb

    # joints ----------------------------------------------------------------------------------
    l7 = bodies[BODY_NAMES[2]]
    slides = l7.findall("joint")
    assert [j.get("type") for j in slides] == ["slide"] * 3
    slide_axis = np.array([_floats(j.get("axis"), 3) for j in slides])
    slide_range = np.array([_floats(j.get("range"), 2) for j in slides])
    assert all(j.get("limited") == "true" for j in slides)
    hinge_range = np.zeros((6, 2))
    hinge_limited = np.zeros(6)
    for k in range(6):
        j = bodies[BODY_NAMES[3 + k]].find("joint")
        assert j.get("type", "hinge") == "hinge" and np.allclose(_floats(j.get("axis"), 3), [0, 0, 1])
        assert np.allclose(_floats(j.get("pos", "0 0 0"), 3), 0)
        hinge_range[k] = _floats(j.get("range"), 2)
        hinge_limited[k] = 1.0 if j.get("limited", "false") == "true" else 0.0
    assert bodies["object"].find("joint").get("type") == "free"
    M["slide_axis"], M["slide_range"] = slide_axis, slide_range
    M["hinge_range"], M["hinge_limited"] = hinge_range, hinge_limited
    M["dof_damping"] = np.full(NV, damping)
    M["dof_armature"] = np.full(NV, armature)

    # geoms + meshes ---------------------------------------------------------------------------
    # Multi-geom objects (Bottle / TBottle / Bowl / RBowl, e.g. ..._sbottle.xml:158-186): the `object` body carries jointless
    # child bodies with one mesh geom each.  MuJoCo welds them to the object (one rigid body); their geoms collide dynamically
    # (default contype / conaffinity 1) with the ground and the hand.  Compiled as: extra geoms 9.. of body 9 with a mesh slot
    # each, the object's inertial = the composite of the pieces, the dynamic pairs appended, and the per-geom inverse weight of
    # the MuJoCo body that owns the geom (`geom_invweight0`, see _invweights).
    pieces = [c for c in bodies["object"] if c.tag == "body"]
    for c in pieces:
        assert c.find("joint") is None and len(c.findall("geom")) == 1 and c.find("body") is None, f"{xml_path.name}: object child {c.get('name')}"
        assert np.allclose(_frame_of(c)[0], 0) and np.allclose(_frame_of(c)[1], [1, 0, 0, 0]), "welded object pieces sit in the object's frame"
    geom_names = GEOM_NAMES + [c.find("geom").get("name") for c in pieces]
    ngeom = len(geom_names)
    geom_mesh_slot = GEOM_MESH + [4 + k for k in range(len(pieces))]
    assert ngeom <= NGEOM_MAX
    meshes = [None] * (4 + len(pieces))
    geom_pos = np.zeros((ngeom, 3))
    geom_quat = np.tile(np.array([1.0, 0, 0, 0]), (ngeom, 1))
    geom_size = np.zeros((ngeom, 3))
    geom_rbound = np.zeros(ngeom)
    geom_elems = {g.get("name"): g for g in wb.iter("geom")}
    assert len(geom_elems) == ngeom, f"{xml_path.name}: geoms {sorted(geom_elems)} "
    for gi, gname in enumerate(geom_names):
        g = geom_elems[gname]
        if gi == 0:
            assert g.get("class") == "ground"
            geom_size[gi] = _floats(g.get("size"), 3)
            continue
        slot = geom_mesh_slot[gi]
        gtype = g.get("type", "sphere")
        if gtype in ("box", "cylinder"):
            # Primitive object geoms (the env's default model is ..._mbox.xml, ENV:62; bbox / scyl / mcyl / bcyl likewise):
            # compiled to their convex polytope in the geom's own frame - a box exactly (8 vertices), a cylinder as a
            # 64-gon prism (radial deviation <= r (1 - cos(pi/64)) = 1.2e-3 r; the README's CylinderS/B meshes are 67-gons) -
            # so that one narrow phase (GJK / MPR on hulls, plane-hull) and one ray caster serve every object.
            # Inertia and the sizes the observation reports are the primitive's own (analytic), not the polytope's.
            assert gi == 8 and not pieces, "only a single-geom object may be a primitive"
            raw = _floats(g.get("size"))
            if gtype == "box":
                assert len(raw) == 3
                prim = dict(kind="box", raw=raw, half=raw.copy(), rbound=float(np.linalg.norm(raw)),
                            inertia=np.array([raw[1] ** 2 + raw[2] ** 2, raw[0] ** 2 + raw[2] ** 2, raw[0] ** 2 + raw[1] ** 2]) / 3.0)
                meshes[slot] = CompiledMesh(box_triangles(*raw), "box", keep_frame=True)
            else:
                assert len(raw) == 2
                r_, h_ = raw
                prim = dict(kind="cylinder", raw=np.array([r_, h_, 0.0]), half=np.array([r_, r_, h_]), rbound=float(np.hypot(r_, h_)),
                            inertia=np.array([r_ ** 2 / 4 + h_ ** 2 / 3, r_ ** 2 / 4 + h_ ** 2 / 3, r_ ** 2 / 2]))
                meshes[slot] = CompiledMesh(cylinder_triangles(r_, h_), "cylinder", keep_frame=True)
        else:
            assert gtype == "mesh", f"{xml_path.name}: geom type {gtype} is not compiled"
            prim = None
            if meshes[slot] is None:
                meshes[slot] = get_mesh(g.get("mesh"))
        cm = meshes[slot]
        gp, gq = _frame_of(g)                       # user frame of the geom in the body (identity here)
        Rg = quat_to_mat(gq)
        geom_pos[gi] = gp + Rg @ cm.com
        geom_quat[gi] = mat_to_quat(Rg @ cm.R)
        geom_size[gi] = cm.size if prim is None else prim["half"]
        geom_rbound[gi] = cm.rbound if prim is None else prim["rbound"]
    # object inertial inferred from its geom (mass on the geom, XML:153)
    og = geom_elems["object"]
    omass = float(og.get("mass"))
    cm = meshes[3]
    body_mass[9] = omass
    body_ipos[9] = geom_pos[8]
    body_iquat[9] = geom_quat[8]
    body_inertia[9] = cm.principal * (omass / cm.volume) if prim is None else prim["inertia"] * omass
    if pieces:
        # the welded pieces: MuJoCo keeps them as bodies 10.. with the inertial of their geom; one rigid body with the composite
        # inertial (parallel axes about the composite centre of mass, then its principal frame) has the same dynamics
        pm = [omass] + [float(geom_elems[n].get("mass")) for n in geom_names[9:]]
        pc = [geom_pos[8 + k] for k in range(len(pm))]                       # centres of mass of the pieces, object frame
        pI = []
        for k in range(len(pm)):
            c_ = meshes[3 + k]
            Rk = quat_to_mat(geom_quat[8 + k])
            pI.append(Rk @ np.diag(c_.principal * (pm[k] / c_.volume)) @ Rk.T)
        mtot = float(np.sum(pm))
        com = np.sum([m_ * c_ for m_, c_ in zip(pm, pc)], axis=0) / mtot
        Ic = np.zeros((3, 3))
        for m_, c_, I_ in zip(pm, pc, pI):
            r_ = c_ - com
            Ic += I_ + m_ * (r_ @ r_ * np.eye(3) - np.outer(r_, r_))
        pr_, Rc = principal_frame(Ic)
        body_mass[9], body_ipos[9], body_iquat[9], body_inertia[9] = mtot, com, mat_to_quat(Rc), pr_
        M["piece_mass"] = np.array(pm)
    M["body_pos"], M["body_quat"], M["body_mass"] = body_pos, body_quat, body_mass
    M["body_ipos"], M["body_iquat"], M["body_inertia"] = body_ipos, body_iquat, body_inertia
    M["geom_pos"], M["geom_quat"], M["geom_size"], M["geom_rbound"] = geom_pos, geom_quat, geom_size, geom_rbound
    M["geom_body"] = np.array(GEOM_BODY + [9] * len(pieces), dtype=np.int32)
    M["geom_mesh"] = np.array(geom_mesh_slot, dtype=np.int32)
    for s, cm in enumerate(meshes):
        M[f"mesh{s}_vert"] = cm.verts
        M[f"mesh{s}_tri"] = cm.tri
        M[f"mesh{s}_bvh_box"] = cm.bvh_box
        M[f"mesh{s}_bvh_lr"] = cm.bvh_lr
        M[f"mesh{s}_adj_off"] = cm.adj_off
        M[f"mesh{s}_adj"] = cm.adj
    M["mesh_info"] = np.array([[cm.volume, len(cm.verts), len(cm.planes), cm.ntri, cm.nsimplex] for cm in meshes], dtype=np.float64)

    # sites -------------------------------------------------------------------------------------
    site_pos = np.zeros((NSITE, 3))
    site_quat = np.zeros((NSITE, 4))
    site_body = np.zeros(NSITE, dtype=np.int32)
    site_elems = {s.get("name"): s for s in wb.iter("site")}
    for si, sname in enumerate(SITE_NAMES):
        s = site_elems[sname]
        site_pos[si], site_quat[si] = _frame_of(s)
        site_body[si] = BODY_NAMES.index(parent[s].get("name"))
    M["site_pos"], M["site_quat"], M["site_body"] = site_pos, site_quat, site_body

    # contact pairs -------------------------------------------------------------------------------
    # explicit pairs first (XML order), then the dynamic candidates of SURVEY App. A
    # MuJoCo gives an explicit <pair> the PAIR defaults, not its geoms' attributes: margin = the pair's own attribute, else
    # <default><pair margin>, else 0 - the geoms' margin 0.001 (XML:40) only reaches the dynamically generated pairs.
    # Pinned by real MuJoCo 1.50 output (Old Code/Pose_file_2.csv: a box released 5 mm inside the floor recovers along
    # 0.052898 / 0.054325 / 0.054774; with margin 0.001 on the object-ground pair it would be 0.053477 / ...).
    pairs = []
    seen = set()
    dpair = dflt.find("pair")
    pair_margin_default = float(dpair.get("margin", "0")) if dpair is not None else 0.0
    for p in root.find("contact").findall("pair"):
        g1, g2 = geom_names.index(p.get("geom1")), geom_names.index(p.get("geom2"))
        fr = _floats(p.get("friction"), 5)
        assert p.get("condim") == "3"
        a, b = min(g1, g2), max(g1, g2)
        pairs.append([a, b, fr[0], fr[1], float(p.get("margin", pair_margin_default))])
        seen.add((a, b))
    hand = list(range(1, 8))
    for a in range(0, 8):
        for b in range(a + 1, 8):
            if (a, b) in seen:
                continue
            if a == 0:
                pass                                   # ground x hand geom (conaffinity 1 vs contype 1)
            else:
                ba, bb = GEOM_BODY[a], GEOM_BODY[b]
                # parent-child filter: palm(2)-prox(3,5,7); prox(i)-dist(i+1)
                if (ba == 2 and bb in (3, 5, 7)) or (bb == ba + 1 and ba in (3, 5, 7)):
                    continue
            pairs.append([a, b, 1.0, 1.0, margin])     # geom default friction 1 0.005 0.0001, condim 3
    assert len(pairs) == 30, len(pairs)
    # the welded pieces of a multi-geom object: dynamic candidates with the ground and the seven hand geoms (geom order); not with each
    # other nor with `object` (bodies welded together are never tested, MuJoCo's body_weldid filter)
    for b in range(9, ngeom):
        for a in range(0, 8):
            pairs.append([a, b, 1.0, 1.0, margin])
    M["pairs"] = np.array(pairs)

    # tendons / equality / actuators -----------------------------------------------------------------
    tend = root.find("tendon").findall("fixed")
    coefs = []
    for t in tend:
        js = t.findall("joint")
        coefs.append([float(js[0].get("coef")), float(js[1].get("coef"))])
    M["tendon_coef"] = np.array(coefs)
    acts = list(root.find("actuator"))
    assert [a.tag for a in acts] == ["velocity", "motor"] * 3 + ["velocity"] * 3
    M["actuator"] = np.array([float(acts[0].get("kv")), float(acts[1].get("gear")),
                              _floats(acts[0].get("ctrlrange"), 2)[1],
                              float(acts[6].get("kv")), _floats(acts[6].get("ctrlrange"), 2)[1]])
    # actuator layout: kv_slide, gear_motor, ctrlrange_slide, kv_finger, ctrlrange_finger

    # inverse weights at qpos0 ------------------------------------------------------------------------
    M.update(_invweights(M))
    # the observation's object size comes from MuJoCo's geom_size: AABB half extents of a mesh, the size attribute of a primitive
    M["obj_size_obs"] = object_size_obs(geom_size[8] if prim is None else prim["raw"], xml_path.name, geom_size[9:])
    return M


def object_size_obs(size, filename, piece_sizes=()):
    """Restatement of KinovaGripper_Env._get_obj_size (kinova_gripper_env.py:706-746); the observation stores
    [s0, s1, 2*s2] (kinova_gripper_env.py:529).  `size`: geom_size of `object`; `piece_sizes`: geom_size of the welded pieces of a
    multi-geom object in geom order (the reference walks the object's geoms from the LAST one back to `object`: widths by maximum,
    heights summed; the bowls get constants scaled by the env's size letter - 'm' on the drivers' path, see below)."""
    final = np.zeros(3)
    # the size letter the bowls' constants are scaled by is `self.obj_size`, which only __init__ ('m': the default model, ENV:62) and the
    # obj_params hook (obj_shape_generator, ENV:1057-1147) ever set: an object that comes from the object schedule - every training and
    # evaluation episode of main_DDPGfD.py (select_object -> get_object, ENV:1171-1172, 986-1005) - is seen with 'm', whatever its file says:
    # compiled with 'm' = x 0.85 (the obj_params hook's own letters, 's' x 0.7 / 'b' x 1, are pinned as data in tests/golden/reset_helpers.npz)
    BOWL_SCALE_M = 0.85
    for size in [np.array(p_, dtype=np.float64) for p_ in list(piece_sizes)[::-1]] + [np.array(size, dtype=np.float64)]:
        size = size.copy()
        if size[2] == 0:
            size[2] = size[1]
            size[1] = size[0]
        diffs = [abs(size[0] - size[1]), abs(size[1] - size[2]), abs(size[0] - size[2])]
        if ("lemon" in filename) or (int(np.argmin(diffs)) != 0):
            size[0], size[2] = size[2], size[0]
        if "Bowl" in filename:
            final[:] = [0.17, 0.17, 0.075] if "Rect" in filename else [0.175, 0.175, 0.07]
            final *= BOWL_SCALE_M      # (the reference multiplies component by component: same products)
        else:
            final[0] = max(size[0], final[0])
            final[1] = max(size[1], final[1])
            final[2] += size[2]
    return np.array([final[0], final[1], final[2] * 2.0])


def _invweights(M):
    """dof_invweight0 / body_invweight0 / tendon_invweight0 at qpos0 with the hand in its
    link frame (the result is invariant to the per-episode hand orientation)."""
    nb = NBODY
    # FK at qpos0
    R = [np.eye(3)] * nb
    p = [np.zeros(3)] * nb
    par = [0, 0, 1, 2, 3, 2, 5, 2, 7, 0]
    for b in range(1, nb):
        R[b] = R[par[b]] @ quat_to_mat(M["body_quat"][b])
        p[b] = p[par[b]] + R[par[b]] @ M["body_pos"][b]
    # jacobians at COM per body
    def jac(b, x):
        Jp = np.zeros((3, NV))
        Jr = np.zeros((3, NV))
        chain = []
        c = b
        while c != 0:
            chain.append(c)
            c = par[c]
        if 2 in chain:
            for k in range(3):
                Jp[:, k] = R[2] @ M["slide_axis"][k]
        for k in range(6):
            hb = 3 + k
            if hb in chain:
                z = R[hb][:, 2]
                Jr[:, 3 + k] = z
                Jp[:, 3 + k] = np.cross(z, x - p[hb])
        if b == 9:
            Jp[:, 9:12] = np.eye(3)
            for k in range(3):
                Jr[:, 12 + k] = R[9][:, k]
                Jp[:, 12 + k] = np.cross(R[9][:, k], x - p[9])
        return Jp, Jr
    Mm = np.diag(M["dof_armature"]).astype(np.float64)
    coms = []
    for b in range(2, nb):
        x = p[b] + R[b] @ M["body_ipos"][b]
        coms.append(x)
        Jp, Jr = jac(b, x)
        Ri = R[b] @ quat_to_mat(M["body_iquat"][b])
        Iw = Ri @ np.diag(M["body_inertia"][b]) @ Ri.T
        Mm += M["body_mass"][b] * Jp.T @ Jp + Jr.T @ Iw @ Jr
    Minv = np.linalg.inv(Mm)
    body_inv = np.zeros((nb, 2))
    for b in range(2, nb):
        Jp, Jr = jac(b, coms[b - 2])
        body_inv[b, 0] = np.trace(Jp @ Minv @ Jp.T) / 3
        body_inv[b, 1] = np.trace(Jr @ Minv @ Jr.T) / 3
    dof_inv = np.diag(Minv).copy()
    dof_inv[9:12] = dof_inv[9:12].mean()
    dof_inv[12:15] = dof_inv[12:15].mean()
    tinv = np.zeros(3)
    for t in range(3):
        Jt = np.zeros(NV)
        Jt[3 + 2 * t] = M["tendon_coef"][t, 0]
        Jt[4 + 2 * t] = M["tendon_coef"][t, 1]
        tinv[t] = Jt @ Minv @ Jt
    out = {"dof_invweight0": dof_inv, "body_invweight0": body_inv, "tendon_invweight0": tinv, "M0": Mm}
    if len(M["geom_body"]) > NGEOM:
        # Multi-geom object: MuJoCo keeps every welded piece as a body of its own, and a contact's regularisation takes the inverse
        # weight of the bodies that own its two geoms - for a piece: the translational inverse weight at THAT piece's centre of mass
        # (mj_setConst: J at the body's xipos).  Hand geoms: their body's value.
        ginv = np.array([body_inv[b, 0] for b in M["geom_body"]])
        for g in range(8, len(ginv)):
            Jp, _ = jac(9, p[9] + R[9] @ M["geom_pos"][g])
            ginv[g] = np.trace(Jp @ Minv @ Jp.T) / 3
        out["geom_invweight0"] = ginv
    return out


# ---------------------------------------------------------------------------------------------
def blob_bytes(M: dict) -> bytes:
    """dtype codes: 0 = float64, 1 = int32, 2 = float32"""
    recs = []
    for name, arr in M.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype.kind in "iu":
            arr, code = arr.astype("<i4"), 1
        elif arr.dtype == np.float32:
            arr, code = arr.astype("<f4"), 2
        else:
            arr, code = arr.astype("<f8"), 0
        nb = name.encode()
        assert len(nb) < 24
        shape = list(arr.shape) + [0] * (4 - arr.ndim)
        hdr = nb.ljust(24, b"\0") + struct.pack("<II4I", code, arr.size, *shape)
        data = arr.tobytes()
        recs.append(hdr + data + b"\0" * ((-len(data)) % 8))
    return MAGIC + struct.pack("<II", VERSION, len(recs)) + b"\0" * 4 + b"".join(recs)


def write_blob(M: dict, path: Path):
    Path(path).write_bytes(blob_bytes(M))


def read_blob(path_or_bytes) -> dict:
    raw = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else Path(path_or_bytes).read_bytes()
    assert raw[:4] == MAGIC, "not a KSMB model blob"
    ver, n = struct.unpack("<II", raw[4:12])
    assert ver == VERSION, f"model blob version {ver} != {VERSION}"
    off = 16
    out = {}
    for _ in range(n):
        name = raw[off:off + 24].rstrip(b"\0").decode()
        code, size, s0, s1, s2, s3 = struct.unpack("<II4I", raw[off + 24:off + 48])
        off += 48
        dt = {0: "<f8", 1: "<i4", 2: "<f4"}[code]
        isz = 8 if code == 0 else 4
        arr = np.frombuffer(raw[off:off + size * isz], dtype=dt).copy()
        shape = [s for s in (s0, s1, s2, s3) if s > 0]
        out[name] = arr.reshape(shape) if shape else arr.reshape(())
        off += size * isz + ((-size * isz) % 8)
    return out


def blob_record_shape(raw: bytes, name: str):
    """shape of one record of a model blob without decoding the arrays (None if absent)"""
    assert raw[:4] == MAGIC, "not a KSMB model blob"
    _, n = struct.unpack("<II", raw[4:12])
    off = 16
    for _ in range(n):
        nm = raw[off:off + 24].rstrip(b"\0").decode()
        code, size, s0, s1, s2, s3 = struct.unpack("<II4I", raw[off + 24:off + 48])
        if nm == name:
            return tuple(x for x in (s0, s1, s2, s3) if x > 0)
        isz = 8 if code == 0 else 4
        off += 48 + size * isz + ((-size * isz) % 8)
    return None


HAND_RAY_KEYS = [f"mesh{s}_{k}" for s in range(3) for k in ("tri", "bvh_box", "bvh_lr")]


def load_model_blob(shape: str, assets: Path) -> bytes:
    """Model blob as the loaders want it: <shape>.ksm plus the rangefinder triangle / BVH tables of the three
    hand meshes, which are identical for every object and therefore stored once in hand_raymesh.kst."""
    M = read_blob(Path(assets) / f"{shape}.ksm")
    M.update(read_blob(Path(assets) / "hand_raymesh.kst"))
    return blob_bytes(M)


def load_coords_table(path: Path) -> np.ndarray:
    """Object start coordinates as the reference reads them (kinova_gripper_env.py:1008-1028):
    the first line is consumed by the delimiter sniffer, the rest are rows of x,y,z[,rx,ry,rz]."""
    lines = Path(path).read_text().splitlines()
    delim = "," if "," in lines[0] else " "
    rows = []
    for ln in lines[1:]:
        parts = [p for p in re.split(delim, ln.strip()) if p != ""]
        if len(parts) >= 3:
            rows.append([float(x) for x in parts[:6]])
    width = max(len(r) for r in rows)
    return np.array([r + [0.0] * (width - len(r)) for r in rows])
