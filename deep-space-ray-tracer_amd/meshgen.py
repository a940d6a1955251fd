"""Procedural "ISS-like" OBJ + MTL generator (deterministic, no randomness).

Why it exists: the mesh every reference number is quoted on -- ../../iss_model/ISS_stationary.obj, src/main.cpp:238 -- is
outside the reference repository and not in this container, and there is no network.  This stand-in has the
station's overall proportions in metres in the model frame the reference renders in (camera on +Z looking at the origin,
src/main.cpp:399): a ~109 m lattice truss along X, eight solar-array wings of ~35 x 11.6 m in the X-Y plane, pressurised
modules as cylinders along Z and X, radiators, and a handful of small parts, written in a fixed generic attitude (see _attitude).  The MTL exercises every branch of the
reference's material heuristics (inc/triangle_mesh.h:75-112): Kd-only -> lambertian, |Ks| > 0.05 -> metal with
fuzz = 100/(Ns+100), d < 0.999 -> dielectric.  Triangle count is a parameter; every figure we report names it.
A real ISS OBJ can be used instead wherever a mesh path is accepted.
"""
import math
import os

import numpy as np

VERSION = "2-attitude"        # bump when the generated geometry changes: profiles and bench lines name it

MATERIALS = """# materials for the procedural ISS-like mesh
newmtl module_white
Kd 0.85 0.85 0.82
Ks 0.0 0.0 0.0
Ns 10
d 1.0

newmtl truss_metal
Kd 0.5 0.5 0.5
Ks 0.6 0.6 0.62
Ns 50
d 1.0

newmtl solar_blue
Kd 0.05 0.07 0.20
Ks 0.02 0.02 0.02
Ns 100
d 1.0

newmtl solar_back
Kd 0.70 0.55 0.30
Ks 0.0 0.0 0.0
d 1.0

newmtl radiator_white
Kd 0.92 0.92 0.92
Ks 0.0 0.0 0.0
d 1.0

newmtl foil_gold
Kd 0.6 0.5 0.2
Ks 0.8 0.6 0.2
Ns 200
d 1.0

newmtl window_glass
Kd 0.9 0.9 0.9
Ks 0.0 0.0 0.0
d 0.3
Ni 1.5

newmtl dark_panel
Kd 0.10 0.10 0.11
Ks 0.0 0.0 0.0
d 1.0
"""


class _Mesh:
    def __init__(self):
        self.verts = []     # list of (n,3) float32 arrays
        self.faces = []     # list of (material, (m,3) int64 arrays, 1-based global indices)
        self.nv = 0

    def add(self, material, verts, faces):
        verts = np.asarray(verts, np.float32).reshape(-1, 3)
        faces = np.asarray(faces, np.int64).reshape(-1, 3)
        self.verts.append(verts)
        self.faces.append((material, faces + self.nv + 1))
        self.nv += len(verts)

    def count(self):
        return sum(len(f) for _, f in self.faces)


def _grid_faces(nu, nv, flip=False):
    """Two triangles per cell of a (nu+1) x (nv+1) vertex grid, row-major in u."""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    c = b + 1
    d = a + 1
    t1 = np.stack([a, b, c], 1)
    t2 = np.stack([a, c, d], 1)
    f = np.concatenate([t1, t2], 0)
    return f[:, ::-1] if flip else f


def _panel(mesh, material, origin, eu, ev, nu, nv, flip=False):
    """Flat rectangle origin + s*eu + t*ev, s,t in [0,1], tessellated nu x nv."""
    nu, nv = max(1, int(nu)), max(1, int(nv))
    s, t = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    p = np.asarray(origin, np.float64)[None, :] + s.reshape(-1, 1) * np.asarray(eu, np.float64)[None, :] + t.reshape(-1, 1) * np.asarray(ev, np.float64)[None, :]
    mesh.add(material, p, _grid_faces(nu, nv, flip))


def _box(mesh, material, lo, hi, n):
    """Axis-aligned box with outward faces; n = (nx, ny, nz) cells per edge."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    nx, ny, nz = n
    _panel(mesh, material, lo, ey, ex, ny, nx)                                  # z = lo, normal -z
    _panel(mesh, material, lo + ez, ex, ey, nx, ny)                             # z = hi, normal +z
    _panel(mesh, material, lo, ex, ez, nx, nz)                                  # y = lo, normal -y
    _panel(mesh, material, lo + ey, ez, ex, nz, nx)                             # y = hi, normal +y
    _panel(mesh, material, lo, ez, ey, nz, ny)                                  # x = lo, normal -x
    _panel(mesh, material, lo + ex, ey, ez, ny, nz)                             # x = hi, normal +x


def _strut(mesh, material, a, b, width, nseg):
    """Thin square-section beam from a to b."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    axis = b - a
    L = np.linalg.norm(axis)
    w = axis / L
    helper = np.array([0.0, 1.0, 0.0]) if abs(w[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(w, helper); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    h = width * 0.5
    corners = [(-h, -h), (h, -h), (h, h), (-h, h)]
    nseg = max(1, int(nseg))
    for k in range(4):
        c0, c1 = corners[k], corners[(k + 1) % 4]
        p0 = a + c0[0] * u + c0[1] * v
        p1 = a + c1[0] * u + c1[1] * v
        _panel(mesh, material, p0, p1 - p0, axis, 1, nseg)


def _cylinder(mesh, material, center, axis, radius, length, nseg, nlen, caps=True, cap_material=None):
    center, axis = np.asarray(center, np.float64), np.asarray(axis, np.float64)
    w = axis / np.linalg.norm(axis)
    helper = np.array([0.0, 1.0, 0.0]) if abs(w[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(w, helper); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    nseg, nlen = max(6, int(nseg)), max(1, int(nlen))
    th = np.linspace(0.0, 2.0 * math.pi, nseg + 1)
    zz = np.linspace(-0.5 * length, 0.5 * length, nlen + 1)
    T, Z = np.meshgrid(th, zz, indexing="ij")
    p = center[None, :] + radius * (np.cos(T).reshape(-1, 1) * u[None, :] + np.sin(T).reshape(-1, 1) * v[None, :]) + Z.reshape(-1, 1) * w[None, :]
    mesh.add(material, p, _grid_faces(nseg, nlen, flip=True))
    if caps:
        for sign in (-1.0, 1.0):
            c = center + sign * 0.5 * length * w
            ring = c[None, :] + radius * (np.cos(th[:-1])[:, None] * u[None, :] + np.sin(th[:-1])[:, None] * v[None, :])
            verts = np.concatenate([c[None, :], ring], 0)
            k = np.arange(nseg)
            f = np.stack([np.zeros(nseg, np.int64), 1 + k, 1 + (k + 1) % nseg], 1)
            mesh.add(cap_material or material, verts, f if sign > 0 else f[:, ::-1])


def _uv_sphere(mesh, material, center, radius, nlat, nlon):
    nlat, nlon = max(4, int(nlat)), max(6, int(nlon))
    lat = np.linspace(0.0, math.pi, nlat + 1)
    lon = np.linspace(0.0, 2.0 * math.pi, nlon + 1)
    A, B = np.meshgrid(lat, lon, indexing="ij")
    p = np.stack([np.sin(A) * np.cos(B), np.cos(A), np.sin(A) * np.sin(B)], -1).reshape(-1, 3) * radius + np.asarray(center)[None, :]
    mesh.add(material, p, _grid_faces(nlat, nlon, flip=True))


def _build(r):
    """Assemble the station at resolution multiplier r (1.0 ~ 60k triangles)."""
    m = _Mesh()
    q = lambda x: max(1, int(round(x * r)))

    # ---- integrated truss: four longerons + battens + diagonals, bays of 4.5 m along X ----
    half, bay, sec = 54.5, 4.5, 2.2
    nb = int(2 * half / bay)
    xs = np.linspace(-half, half, nb + 1)
    for (sy, sz) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        _strut(m, "truss_metal", (-half, sy * sec, sz * sec), (half, sy * sec, sz * sec), 0.18, q(nb * 2))
    for i, x in enumerate(xs):
        c = [(x, -sec, -sec), (x, sec, -sec), (x, sec, sec), (x, -sec, sec)]
        for k in range(4):
            _strut(m, "truss_metal", c[k], c[(k + 1) % 4], 0.10, q(3))
        if i < nb:
            x2 = xs[i + 1]
            s = 1 if i % 2 == 0 else -1
            _strut(m, "truss_metal", (x, -sec * s, -sec), (x2, sec * s, -sec), 0.08, q(4))
            _strut(m, "truss_metal", (x, -sec * s, sec), (x2, sec * s, sec), 0.08, q(4))
            _strut(m, "truss_metal", (x, sec, -sec * s), (x2, sec, sec * s), 0.08, q(4))
            _strut(m, "truss_metal", (x, -sec, -sec * s), (x2, -sec, sec * s), 0.08, q(4))
    # equipment boxes along the truss
    for i in range(2, nb - 1, 3):
        xc = 0.5 * (xs[i] + xs[i + 1])
        _box(m, "foil_gold" if i % 2 else "dark_panel", (xc - 1.4, -1.2, -1.6), (xc + 1.4, 1.2, 1.6), (q(3), q(3), q(3)))

    # ---- eight solar array wings: two blankets each, in the X-Y plane, thin in Z ----
    for x0 in (-48.0, -34.0, 34.0, 48.0):
        for sy in (-1.0, 1.0):
            y0, y1 = sy * 4.0, sy * 39.0
            ylo, yhi = min(y0, y1), max(y0, y1)
            for dx in (-3.1, 3.1):
                lo = (x0 + dx - 2.7, ylo, -0.03)
                hi = (x0 + dx + 2.7, yhi, 0.03)
                ex, ey = np.array([hi[0] - lo[0], 0, 0]), np.array([0, hi[1] - lo[1], 0])
                _panel(m, "solar_blue", (lo[0], lo[1], hi[2]), ex, ey, q(8), q(48))             # sun side (+z)
                _panel(m, "solar_back", (lo[0], lo[1], lo[2]), ey, ex, q(48), q(8))             # back side (-z)
            _strut(m, "truss_metal", (x0, ylo, 0.0), (x0, yhi, 0.0), 0.25, q(24))               # mast
            _cylinder(m, "foil_gold", (x0, sy * 3.0, 0.0), (0, 1, 0), 0.6, 2.0, q(14), q(2))   # beta gimbal

    # ---- pressurised modules: main stack along Z, cross modules along X ----
    _cylinder(m, "module_white", (0.0, -3.2, 4.0), (0, 0, 1), 2.2, 44.0, q(40), q(60))
    _cylinder(m, "module_white", (0.0, -3.2, -24.0), (0, 0, 1), 1.6, 10.0, q(32), q(16))
    _cylinder(m, "module_white", (-8.5, -3.2, 18.0), (1, 0, 0), 2.2, 11.0, q(40), q(18))
    _cylinder(m, "module_white", (8.5, -3.2, 18.0), (1, 0, 0), 2.2, 11.0, q(40), q(18))
    _cylinder(m, "module_white", (7.0, -3.2, 6.0), (1, 0, 0), 2.0, 8.0, q(36), q(14))
    _cylinder(m, "foil_gold", (0.0, -3.2, 27.5), (0, 0, 1), 1.1, 3.0, q(24), q(4))              # docking adapter
    _cylinder(m, "dark_panel", (0.0, -7.0, 10.0), (0, 1, 0), 1.0, 3.0, q(24), q(4))            # nadir port
    _uv_sphere(m, "window_glass", (0.0, -6.2, 14.0), 1.2, q(12), q(20))                         # cupola
    for z in (-6.0, 0.0, 6.0, 12.0):
        _box(m, "window_glass", (2.15, -3.5, z - 0.3), (2.3, -2.9, z + 0.3), (1, q(2), q(2)))

    # ---- radiators: three panels each side hanging below the truss ----
    for x0 in (-14.0, 14.0):
        for k in range(3):
            xc = x0 + (k - 1) * 3.6
            _panel(m, "radiator_white", (xc - 1.6, -2.5, -1.0), (3.2, 0, 0), (0, -21.0, 0.0), q(5), q(30), flip=True)
            _panel(m, "radiator_white", (xc - 1.6, -2.5, -1.06), (0, -21.0, 0.0), (3.2, 0, 0), q(30), q(5), flip=True)

    # ---- small parts ----
    _cylinder(m, "truss_metal", (20.0, 4.5, 0.0), (0, 1, 0), 0.9, 0.2, q(24), 1)               # antenna dish
    _strut(m, "truss_metal", (20.0, 2.2, 0.0), (20.0, 4.4, 0.0), 0.12, q(4))
    _cylinder(m, "truss_metal", (-20.0, 4.5, 1.0), (0, 1, 0), 0.9, 0.2, q(24), 1)
    _strut(m, "truss_metal", (-20.0, 2.2, 1.0), (-20.0, 4.4, 1.0), 0.12, q(4))
    _strut(m, "foil_gold", (3.0, -1.0, 2.3), (12.0, 3.0, 9.0), 0.35, q(20))                    # robotic arm
    return m


def build_station(target_triangles):
    """Find the resolution whose triangle count is closest to the target (monotone in r), return the mesh."""
    lo, hi = 0.05, 1.0
    while _build(hi).count() < target_triangles and hi < 64:
        hi *= 2.0
    for _ in range(18):
        mid = 0.5 * (lo + hi)
        if _build(mid).count() < target_triangles:
            lo = mid
        else:
            hi = mid
    return _build(hi)


def _attitude():
    """Fixed, generic attitude of the station in its model frame: yaw 11.3, pitch 7.1, roll 4.7 degrees.

    Not cosmetic.  Built axis-aligned, half of the triangles end up in BVH leaves whose box has zero thickness (a few coplanar
    panel triangles), and the reference's slab test rejects such a box outright (`t_max <= t_min` with equality,
    src/gpu_render.cu:312): the reference -- and therefore this renderer -- would not see them at all, and the benchmark would
    trace rays through an invisible truss.  A real model is never exactly aligned with the axes everywhere; neither is this one."""
    ay, ax, az = (math.radians(v) for v in (11.3, 7.1, 4.7))
    ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    rz = np.array([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1]])
    return rz @ rx @ ry


def write_obj(mesh, obj_path, mtl_name=None):
    obj_path = str(obj_path)
    mtl_name = mtl_name or (os.path.splitext(os.path.basename(obj_path))[0] + ".mtl")
    with open(os.path.join(os.path.dirname(obj_path) or ".", mtl_name), "w") as f:
        f.write(MATERIALS)
    with open(obj_path, "w") as f:
        f.write("# procedural ISS-like mesh (deep-space-ray-tracer_amd/meshgen.py)\n")
        f.write(f"mtllib {mtl_name}\n")
        rot = _attitude()
        for v in mesh.verts:
            v = (np.asarray(v, np.float64) @ rot.T).astype(np.float32)
            f.write("".join("v %.9g %.9g %.9g\n" % (float(a), float(b), float(c)) for a, b, c in v))
        for material, faces in mesh.faces:
            f.write(f"usemtl {material}\n")
            f.write("".join("f %d %d %d\n" % (a, b, c) for a, b, c in faces.tolist()))
    return mesh.count()


def generate(obj_path, target_triangles):
    """Write <obj_path> (+ .mtl next to it); returns the exact triangle count."""
    return write_obj(build_station(int(target_triangles)), obj_path)
