"""Seeded synthetic stand-ins for the reference's stripped Artec scans.

python_scripts/Bus.obj, Car-body.ply and Motorbike_ply.ply are listed in the
reference's .MISSING_LARGE_BLOBS; BASELINE.md §4 replaces them by procedural
meshes of comparable size and extent.  Everything here is deterministic
(numpy default_rng(seed), float32 output) so the meshes can be regenerated on
the GPU box instead of being shipped as blobs; tests/test_meshgen.py pins
their SHA-256.
"""
import hashlib

import numpy as np

f32 = np.float32


def _grid(nu, nv):
    """(nu+1)*(nv+1) lattice in [0,1]^2 and its 2*nu*nv triangles."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return uu.reshape(-1), vv.reshape(-1), tris.astype(np.uint32)


def stairs(num_steps):
    """src/librender/tests/mesh_generation.py:29-63 (unit-cube stairs)."""
    size_step = 1.0 / num_steps
    v = np.zeros((4 * num_steps, 3))
    f = np.zeros((4 * num_steps - 2, 3), dtype=np.uint32)
    for i in range(num_steps):
        h = i * size_step
        s1 = i * size_step
        s2 = (i + 1) * size_step
        k = 4 * i
        v[k + 0] = [0.0, s1, h]
        v[k + 1] = [1.0, s1, h]
        v[k + 2] = [0.0, s2, h]
        v[k + 3] = [1.0, s2, h]
        f[k] = [k, k + 1, k + 2]
        f[k + 1] = [k + 1, k + 3, k + 2]
        if i < num_steps - 1:
            f[k + 2] = [k + 2, k + 3, k + 5]
            f[k + 3] = [k + 5, k + 4, k + 2]
    return v.astype(f32), f


def rectangle_obj():
    """Two-triangle rectangle matching the known answers of
    src/librender/tests/test_mesh.py:257-298 (resources/data is absent):
    verts (+-1,+-1,0); prim 0 holds (-.3,-.3) with prim_uv (.35,.3), prim 1
    holds (.3,.3) with prim_uv (.3,.35)."""
    a, b, c, d = [-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]
    v = np.array([a, b, c, d], dtype=f32)
    f = np.array([[1, 3, 0], [1, 2, 3]], dtype=np.uint32)
    return v, f


def triangle_soup(n, seed, extent=1.0, size=0.1):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    v = (c + rng.uniform(-size, size, (n, 3, 3))).reshape(-1, 3).astype(f32)
    f = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return v, f


def _transform(v, yaw_deg, translate, scale=1.0):
    a = np.radians(yaw_deg)
    r = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    return ((v * scale) @ r.T + np.asarray(translate)).astype(f32)


def bus(n_tris=200_000, seed=1):
    """Closed box-with-wheels, panels displaced by ribs + seeded noise.

    bbox ~ 12 x 2.5 x 3 m, centred at the origin, long axis x, up z.
    """
    rng = np.random.default_rng(seed)
    L, W, H = 12.0, 2.5, 3.0
    n_wheels = 6
    wheel_frac = 0.12
    quads_body = int(n_tris * (1 - wheel_frac) / 2)
    # distribute quads over the six faces proportionally to area
    areas = np.array([L * H, L * H, W * H, W * H, L * W, L * W])
    dens = np.sqrt(quads_body / areas.sum())
    verts, faces, off = [], [], 0
    phase = rng.uniform(0, 2 * np.pi, 8)

    def add(v, t):
        nonlocal off
        verts.append(v)
        faces.append(t + off)
        off += v.shape[0]

    def face(origin, eu, ev, normal, lu, lv):
        nu, nv = max(2, int(round(lu * dens))), max(2, int(round(lv * dens)))
        u, v, t = _grid(nu, nv)
        p = origin + u[:, None] * eu + v[:, None] * ev
        # window ribs along the long axis + low-amplitude seeded waviness,
        # faded to zero at the face border so the box stays closed
        fade = np.minimum(np.minimum(u, 1 - u), np.minimum(v, 1 - v))
        fade = np.clip(fade * 12.0, 0.0, 1.0)
        rib = 0.03 * np.sin(2 * np.pi * 14 * u + phase[0]) * (np.sin(2 * np.pi * 2 * v + phase[1]) > 0.2)
        wav = 0.01 * np.sin(2 * np.pi * 37 * u + phase[2]) * np.sin(2 * np.pi * 29 * v + phase[3])
        noise = 0.004 * rng.standard_normal(u.shape[0])
        p = p + ((rib + wav + noise) * fade)[:, None] * normal
        add(p, t)

    x0, y0, z0 = -L / 2, -W / 2, -H / 2 + 0.35
    ex, ey, ez = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0, 0, 1.0])
    face(np.array([x0, y0, z0]), ex * L, ez * H, -ey, L, H)                 # -y side
    face(np.array([x0, -y0, z0]), ez * H, ex * L, ey, H, L)                 # +y side
    face(np.array([x0, y0, z0]), ez * H, ey * W, -ex, H, W)                 # back
    face(np.array([-x0, y0, z0]), ey * W, ez * H, ex, W, H)                 # front
    face(np.array([x0, y0, z0 + H]), ex * L, ey * W, ez, L, W)              # roof
    face(np.array([x0, y0, z0]), ey * W, ex * L, -ez, W, L)                 # floor
    # wheels: closed cylinders, axis y
    quads_wheel = int(n_tris * wheel_frac / 2 / n_wheels)
    nr = max(8, int(np.sqrt(quads_wheel * 2.0)))
    nw = max(2, quads_wheel // nr // 2)
    R, Wd = 0.5, 0.3
    for k in range(n_wheels):
        cx = x0 + L * (0.15 + 0.7 * (k // 2) / max(1, n_wheels // 2 - 1))
        cy = (y0 + Wd / 2 + 0.05) if k % 2 == 0 else (-y0 - Wd / 2 - 0.05)
        cz = z0 - 0.05
        u, v, t = _grid(nr, nw)
        ang = 2 * np.pi * u
        p = np.stack([cx + R * np.cos(ang), cy + (v - 0.5) * Wd, cz + R * np.sin(ang)], -1)
        add(p, t)
        for side in (-0.5, 0.5):      # caps as polar grids
            u, v, t = _grid(nr, max(2, nw // 2))
            ang = 2 * np.pi * u
            p = np.stack([cx + R * v * np.cos(ang), np.full_like(u, cy + side * Wd), cz + R * v * np.sin(ang)], -1)
            add(p, t)
    v = np.concatenate(verts).astype(f32)
    f = np.concatenate(faces).astype(np.uint32)
    # drop degenerate (zero-area) triangles of the polar caps
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1)
    f = f[area > 1e-12]
    return v, f


def car_body(n_tris=1_000_000, seed=2, with_normals=True):
    """Superellipsoid shell ~ 4.5 x 1.8 x 1.4 m with vertex normals."""
    rng = np.random.default_rng(seed)
    nq = n_tris // 2
    nu = int(np.sqrt(nq * 2.0))
    nv = max(4, nq // nu)
    u, v, t = _grid(nu, nv)
    th = 2 * np.pi * u                 # longitude
    ph = np.pi * (v - 0.5) * 0.999     # latitude (poles pinched, not collapsed)
    e1, e2 = 0.6, 0.8

    def sp(x, e):
        return np.sign(x) * np.abs(x) ** e

    a, b, c = 2.25, 0.9, 0.7
    x = a * sp(np.cos(ph), e1) * sp(np.cos(th), e2)
    y = b * sp(np.cos(ph), e1) * sp(np.sin(th), e2)
    z = c * sp(np.sin(ph), e1)
    bump = 0.01 * np.sin(9 * th + rng.uniform(0, 6.28)) * np.cos(7 * ph + rng.uniform(0, 6.28))
    p = np.stack([x * (1 + bump), y * (1 + bump), z * (1 + bump)], -1)
    verts = p.astype(f32)
    faces = t
    normals = None
    if with_normals:
        # area-weighted vertex normals (mesh.cpp:201-278 recompute style)
        p0, p1, p2 = verts[faces[:, 0]].astype(np.float64), verts[faces[:, 1]].astype(np.float64), verts[faces[:, 2]].astype(np.float64)
        fn = np.cross(p1 - p0, p2 - p0)
        n = np.zeros((verts.shape[0], 3))
        for k in range(3):
            np.add.at(n, faces[:, k], fn)
        ln = np.linalg.norm(n, axis=1, keepdims=True)
        ln[ln == 0] = 1.0
        normals = (n / ln).astype(f32)
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = np.linalg.norm(np.cross(p1 - p0, p2 - p0), axis=1)
    faces = faces[area > 1e-14]
    return verts, faces, normals


def motorbike(n_tris=300_000, seed=5):
    """Two torus wheels + frame tubes; ~2.1 x 0.4 x 1.1 m."""
    rng = np.random.default_rng(seed)
    verts, faces, off = [], [], 0

    def add(v, t):
        nonlocal off
        verts.append(v)
        faces.append(t + off)
        off += v.shape[0]

    n_parts = 2 + 5
    per = n_tris // n_parts // 2
    nu = int(np.sqrt(per * 3.0))
    nv = max(4, per // nu)
    for cx in (-0.7, 0.7):     # torus wheels in the xz plane
        u, v, t = _grid(nu, nv)
        a, b = 2 * np.pi * u, 2 * np.pi * v
        R, r = 0.3, 0.06
        p = np.stack([cx + (R + r * np.cos(b)) * np.cos(a), r * np.sin(b) * 1.2, 0.36 + (R + r * np.cos(b)) * np.sin(a)], -1)
        add(p, t)
    ends = [((-0.7, 0, 0.36), (0.0, 0, 0.55)), ((0.7, 0, 0.36), (0.45, 0, 0.95)), ((0.0, 0, 0.55), (0.45, 0, 0.95)),
            ((0.0, 0, 0.55), (-0.35, 0, 0.85)), ((-0.35, 0, 0.85), (0.45, 0, 0.95))]
    for (p0, p1) in ends:
        p0, p1 = np.array(p0), np.array(p1)
        d = p1 - p0
        ln = np.linalg.norm(d)
        d = d / ln
        s = np.cross(d, [0, 1.0, 0])
        s /= np.linalg.norm(s)
        tt = np.cross(d, s)
        u, v, t = _grid(nu, nv)
        ang = 2 * np.pi * v
        rad = 0.04 * (1 + 0.1 * np.sin(6 * np.pi * u + rng.uniform(0, 6.28)))
        p = p0 + (u * ln)[:, None] * d + (rad * np.cos(ang))[:, None] * s + (rad * np.sin(ang))[:, None] * tt
        add(p, t)
    v = np.concatenate(verts).astype(f32)
    f = np.concatenate(faces).astype(np.uint32)
    return v, f


def vertex_normals(verts, faces):
    """Angle-weighted vertex normals (Thuermer & Wuethrich) — what Mesh::recompute_vertex_normals
    (mesh.cpp:201-249) gives an OBJ / PLY file that carries none, e.g. a raw 3-D scan."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    p = [v[f[:, k]] for k in range(3)]
    fn = np.cross(p[1] - p[0], p[2] - p[0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    ok = ln[:, 0] > 0
    fn[ok] /= ln[ok]
    n = np.zeros_like(v)
    for j in range(3):
        a = p[(j + 1) % 3] - p[j]
        b = p[(j + 2) % 3] - p[j]
        c = np.einsum("ij,ij->i", a, b) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)
        ang = np.arccos(np.clip(c, -1.0, 1.0))
        ang[~ok] = 0.0
        np.add.at(n, f[:, j], fn * ang[:, None])
    l = np.linalg.norm(n, axis=1, keepdims=True)
    bad = l[:, 0] == 0
    l[bad] = 1.0
    n /= l
    n[bad] = (1.0, 0.0, 0.0)
    return n.astype(f32)


def place(v, yaw_deg=0.0, translate=(0, 0, 0), scale=1.0):
    return _transform(v.astype(np.float64), yaw_deg, translate, scale)


def sha256(v, f):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(v, dtype=f32).tobytes())
    h.update(np.ascontiguousarray(f, dtype=np.uint32).tobytes())
    return h.hexdigest()
