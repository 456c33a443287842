"""Synthetic RGB-D scene of SURVEY.md section 8(d): a closed 6 x 3 x 4 m box room with procedurally
textured walls, a pinhole camera (TUM fr1 intrinsics, rgbd_benchmark/fr1_cam.yaml:1-4) on an
orbit, analytic ray-plane depth, optional holes / Kinect-like depth noise (the sigma(z) model the
reference itself uses, core/src/motion_detection_kernels.cu:98), and a seeded supersurfel model
for the large-N configurations.  Pure numpy, seeded (default 1234).

Camera frame: x right, y down, z forward.  pose = (R, t), camera-to-map: p_map = R p_cam + t,
row-major 12 floats as Transform3 (matrix_types.h:38-42)."""
import numpy as np

ROOM = np.array([3.0, 1.5, 2.0], np.float64)  # half extents: x +-3, y +-1.5 (y down: floor at +1.5), z +-2

# Axis-aligned boxes standing on the floor (lo, hi): with the pitched camera every view contains
# three plane orientations, so the point-to-plane system of the ICP stage is well conditioned.
BOXES = [((1.8, 0.3, 0.9), (2.6, 1.5, 1.7)), ((-2.6, 0.0, 1.0), (-1.9, 1.5, 1.8)),
         ((-2.5, 0.5, -1.8), (-1.7, 1.5, -1.0)), ((1.7, -0.2, -1.7), (2.5, 1.5, -0.9)),
         ((-0.4, 0.2, 1.5), (0.4, 1.5, 2.0)), ((-0.5, 0.4, -2.0), (0.3, 1.5, -1.4)),
         ((2.5, 0.1, -0.4), (3.0, 1.5, 0.4)), ((-3.0, 0.3, -0.5), (-2.4, 1.5, 0.3))]
BASE = np.array([[200, 90, 70], [70, 160, 200], [120, 120, 120], [230, 230, 210], [90, 190, 100],
                 [210, 180, 60], [170, 80, 190], [60, 200, 190], [240, 140, 40], [130, 210, 60]], np.float64)


def _faces():
    """(axis, coord, lo3, hi3, normal_sign, texture id) of every rectangle in the scene."""
    f = []
    for axis in range(3):
        for sign in (+1, -1):  # room walls face inwards
            f.append((axis, sign * ROOM[axis], -ROOM, ROOM.copy(), -sign, len(f)))
    for b, (lo, hi) in enumerate(BOXES):
        lo, hi = np.array(lo, np.float64), np.array(hi, np.float64)
        for axis in range(3):
            for sign in (+1, -1):  # box faces face outwards
                coord = hi[axis] if sign > 0 else lo[axis]
                if axis == 1 and sign > 0:
                    continue  # bottom face sits on the floor
                f.append((axis, coord, lo, hi, sign, 6 + (b * 3 + axis) % 4))
    return f


FACES = _faces()
WALLS = [(a, int(np.sign(c))) for a, c, _, _, _, _ in FACES[:6]]


def intrinsics(width=640, height=480):
    s = width / 640.0
    return dict(width=width, height=height, fx=525.0 * s, fy=525.0 * s,
                cx=(319.5 + 0.5) * s - 0.5, cy=(239.5 + 0.5) * s - 0.5)


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def orbit_pose(k, radius=0.5, deg_per_frame=1.0, yaw0_deg=35.0, pitch_deg=18.0):
    """Ground-truth camera-to-map pose of frame k: camera on a circle of `radius` around the room
    centre, yaw advancing deg_per_frame per frame from yaw0, pitched down by pitch_deg (y is down,
    so looking down is a negative rotation about x)."""
    a = np.deg2rad(yaw0_deg + k * deg_per_frame)
    R = rot_y(a) @ rot_x(-np.deg2rad(pitch_deg))
    t = np.array([radius * np.sin(a), -0.2, radius * np.cos(a)]) * 0.6
    return R, t


def pose12(R, t):
    return np.concatenate([np.asarray(R, np.float64).reshape(9), np.asarray(t, np.float64)]).astype(np.float32)


def _face_uv(axis, p):
    others = [a for a in range(3) if a != axis]
    return p[..., others[0]], p[..., others[1]]


def texture(tex, axis, p):
    """sRGB (0..255 float) at world points p (...,3) of a face with texture id `tex` normal to
    `axis`: base colour, 0.5 m checker with strong contrast (adjacent superpixels differ by > 20 Lab
    units across checker edges) and a low-frequency sinusoid."""
    a, b = _face_uv(axis, p)
    chk = (np.floor(a / 0.5) + np.floor(b / 0.5)) % 2
    base = BASE[tex % len(BASE)]
    col = base[None, :] * (0.55 + 0.45 * chk[..., None])
    wave = 18.0 * np.sin(1.3 * a + 0.7 * tex)[..., None] * np.array([1.0, 0.6, -0.8]) \
        + 12.0 * np.cos(0.9 * b - 0.4 * tex)[..., None] * np.array([-0.5, 1.0, 0.7])
    return np.clip(col + wave, 0.0, 255.0)


def render(R, t, width=640, height=480, noise=False, holes=0.0, rng=None):
    """Return (rgb uint8 HxWx3, depth float32 HxW metres, face id HxW)."""
    K = intrinsics(width, height)
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_cam = np.stack([(u - K["cx"]) / K["fx"], (v - K["cy"]) / K["fy"], np.ones_like(u)], -1)
    d = d_cam @ np.asarray(R, np.float64).T
    o = np.asarray(t, np.float64)
    best = np.full(u.shape, np.inf)
    face_id = np.zeros(u.shape, np.int32)
    for fi, (axis, coord, lo, hi, _, _) in enumerate(FACES):
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (coord - o[axis]) / d[..., axis]
        ok = (s > 1e-6) & (s < best)
        for a in range(3):
            if a != axis:
                pa = o[a] + s * d[..., a]
                ok &= (pa >= lo[a] - 1e-9) & (pa <= hi[a] + 1e-9)
        best = np.where(ok, s, best)
        face_id = np.where(ok, fi, face_id)
    pts = o + np.where(np.isfinite(best), best, 0.0)[..., None] * d
    rgb = np.zeros(u.shape + (3,), np.float64)
    for fi, (axis, _, _, _, _, tex) in enumerate(FACES):
        m = face_id == fi
        if m.any():
            rgb[m] = texture(tex, axis, pts[m])
    depth = best.copy()  # d_cam.z == 1 -> range along the optical axis
    if rng is None:
        rng = np.random.default_rng(1234)
    if noise:
        sigma = 0.0012 + 0.0019 * (depth - 0.4) ** 2
        depth = depth + rng.standard_normal(depth.shape) * sigma
    if holes > 0:
        depth = np.where(rng.random(depth.shape) < holes, 0.0, depth)
    depth = np.where(np.isfinite(depth), depth, 0.0)
    return np.round(rgb).astype(np.uint8), depth.astype(np.float32), face_id


def seed_model(n, stamp=1, seed=1234, conf=3000.0, sigma_plane=0.02, sigma_normal=0.002):
    """n supersurfels sampled uniformly (seeded, area weighted) on the scene's faces, in the
    reference SoA layout (supersurfels.hpp:34-40)."""
    rng = np.random.default_rng(seed)
    areas = []
    for axis, _, lo, hi, _, _ in FACES:
        o = [a for a in range(3) if a != axis]
        areas.append((hi[o[0]] - lo[o[0]]) * (hi[o[1]] - lo[o[1]]))
    areas = np.array(areas)
    face = rng.choice(len(FACES), size=n, p=areas / areas.sum())
    pos = np.zeros((n, 3))
    ori = np.zeros((n, 3, 3))
    col = np.zeros((n, 3))
    for fi, (axis, coord, lo, hi, nsign, tex) in enumerate(FACES):
        m = face == fi
        k = int(m.sum())
        if k == 0:
            continue
        p = np.zeros((k, 3))
        others = [a for a in range(3) if a != axis]
        for a in others:
            p[:, a] = rng.uniform(lo[a], hi[a], k)
        p[:, axis] = coord
        pos[m] = p
        nrm = np.zeros(3); nrm[axis] = nsign
        e0 = np.zeros(3); e0[others[0]] = 1.0
        e1 = np.cross(nrm, e0)
        ori[m] = np.stack([e0, e1, nrm])[None]
        col[m] = texture(tex, axis, p)
    lam = np.array([sigma_plane ** 2 * 1.5, sigma_plane ** 2, sigma_normal ** 2])
    shape_full = np.einsum("nki,k,nkj->nij", ori, lam, ori)
    shapes = np.stack([shape_full[:, 0, 0], shape_full[:, 0, 1], shape_full[:, 0, 2], shape_full[:, 1, 1],
                       shape_full[:, 1, 2], shape_full[:, 2, 2]], 1)
    return dict(positions=pos.astype(np.float32), colors=col.astype(np.float32),
                stamps=np.tile(np.array([[0, max(stamp - 1, 0)]], np.int32), (n, 1)),
                orientations=ori.reshape(n, 9).astype(np.float32), shapes=shapes.astype(np.float32),
                dims=np.tile(lam[None, :2], (n, 1)).astype(np.float32),
                confidences=np.full(n, conf, np.float32))


def visible_first(model, R, t, width=640, height=480, zmin=0.2, zmax=5.0):
    """Reorder a seeded model so that the surfels inside the view frustum of (R,t) come first
    (the order filterModel + the stable partition would produce); returns (model, n_visible)."""
    K = intrinsics(width, height)
    p = (model["positions"].astype(np.float64) - np.asarray(t)) @ np.asarray(R)
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K["fx"] * p[:, 0] / p[:, 2] + K["cx"]
        v = K["fy"] * p[:, 1] / p[:, 2] + K["cy"]
    vis = (p[:, 2] > zmin) & (p[:, 2] < zmax) & (u >= 0) & (u < width) & (v >= 0) & (v < height)
    order = np.concatenate([np.nonzero(vis)[0], np.nonzero(~vis)[0]])
    return {k: a[order] for k, a in model.items()}, int(vis.sum())


def relative_pose(k, **kw):
    """Ground-truth pose of frame k expressed in the frame of camera 0 (the reference starts at
    the identity pose and its first frame copies camera-frame surfels into the map,
    supersurfel_fusion.cu:133-136,477-483)."""
    R0, t0 = orbit_pose(0, **kw)
    Rk, tk = orbit_pose(k, **kw)
    return R0.T @ Rk, R0.T @ (tk - t0)


def seed_model_cam0(n, width=640, height=480, stamp=30, seed=1234):
    """seed_model() expressed in the frame of camera 0 (= the map frame of a run that starts at the
    identity pose), visible rows first.  Returns (model arrays, n_visible)."""
    R0, t0 = orbit_pose(0)
    model = seed_model(n, stamp=stamp, seed=seed)
    model["positions"] = ((model["positions"].astype(np.float64) - t0) @ R0).astype(np.float32)
    O = model["orientations"].reshape(-1, 3, 3).astype(np.float64) @ R0
    model["orientations"] = O.reshape(-1, 9).astype(np.float32)
    Rm = R0.T
    s = model["shapes"].astype(np.float64)
    Cf = np.zeros((n, 3, 3))
    Cf[:, 0, 0], Cf[:, 0, 1], Cf[:, 0, 2], Cf[:, 1, 1], Cf[:, 1, 2], Cf[:, 2, 2] = s.T
    Cf[:, 1, 0], Cf[:, 2, 0], Cf[:, 2, 1] = Cf[:, 0, 1], Cf[:, 0, 2], Cf[:, 1, 2]
    Cf = Rm @ Cf @ Rm.T
    model["shapes"] = np.stack([Cf[:, 0, 0], Cf[:, 0, 1], Cf[:, 0, 2], Cf[:, 1, 1], Cf[:, 1, 2], Cf[:, 2, 2]], 1).astype(np.float32)
    return visible_first(model, np.eye(3), np.zeros(3), width, height)


def seed_model_cam0_visible(n, width=640, height=480, stamp=30, seed=1234, chunk=1000000):
    """n seeded supersurfels that are ALL inside the view frustum of camera 0 (BASELINE config 3: "~1M
    surfels, all visible"): seed_model_cam0 chunks (seeds seed, seed+1, ..) filtered to their visible rows."""
    parts, have, k = [], 0, 0
    while have < n:
        m, nv = seed_model_cam0(chunk, width, height, stamp=stamp, seed=seed + k)
        parts.append({key: v[:nv] for key, v in m.items()})
        have += nv
        k += 1
    model = {key: np.concatenate([p_[key] for p_ in parts])[:n] for key in parts[0]}
    return model, n


def tile_owner(positions, nranks, tile=0.5):
    """numpy twin of the library's spatial-tile owner hash (ssf_stage_fuse / shard_owner)."""
    if nranks <= 1:
        return np.zeros(len(positions), np.int64)
    t = np.float32(tile)
    ijk = np.floor(positions.astype(np.float32) / t).astype(np.int32).astype(np.uint32)
    h = (ijk[:, 0] * np.uint32(73856093)) ^ (ijk[:, 1] * np.uint32(19349663)) ^ (ijk[:, 2] * np.uint32(83492791))
    return (h % np.uint32(nranks)).astype(np.int64)
