"""Deterministic synthetic TempleRing-style sequences (there is no dataset in the image and no network).

A cloud of Gaussian blobs on a bumpy shell (radius 0.07-0.10) around the world origin is observed by a
pinhole camera that travels on a ring of radius ``dist`` and looks at the origin, ``deg_per_frame`` per
frame -- the geometry of Middlebury TempleRing.  The generator writes exactly the
on-disk layout the reference CLI ingests (T:1678-1711):

    <root>/templeRing/templeR_par.txt   n, then "name k11..k33 r11..r33 t1 t2 t3"  (world->camera R,t)
    <root>/templeRing/templeR_ang.txt   "lat lon name"
    <root>/templeRing_pgm/<stem>.pgm    binary P5, maxval 255

Only numpy is used; every random draw comes from ``numpy.random.default_rng(seed)``.
"""
from __future__ import annotations

import os

import numpy as np

K_TEMPLE = np.array([[1520.4, 0.0, 302.32], [0.0, 1525.9, 246.87], [0.0, 0.0, 1.0]])


def make_scene(n_blobs: int = 20000, seed: int = 7, shell_scale: float = 1.0, coarse_frac: float = 0.0):
    """shell_scale > 1 inflates the shell (x3.5 fills a 640x480 frame with texture: the 5k-track config C3);
    coarse_frac of the blobs are drawn 5x wider so that the upper pyramid levels keep structure when the fine blobs
    are dense enough to merge there."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_blobs, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # bumpy, non-planar shell: radius depends smoothly on direction plus jitter
    bump = 0.085 + 0.012 * np.sin(7.0 * d[:, 0]) * np.cos(5.0 * d[:, 1]) + 0.003 * rng.normal(size=n_blobs)
    r = np.clip(bump, 0.07, 0.10) * shell_scale
    pts = d * r[:, None]
    sigma = rng.uniform(1.2, 2.2, size=n_blobs)
    amp = rng.uniform(60.0, 210.0, size=n_blobs)
    if coarse_frac > 0.0:  # drawn after everything else: the default scene (coarse_frac = 0) is unchanged
        coarse = rng.random(n_blobs) < coarse_frac
        sigma = np.where(coarse, sigma * 5.0, sigma)
        amp = np.where(coarse, amp * 0.5, amp)
    return dict(pts=pts, normals=d, sigma=sigma, amp=amp)


def ring_pose(angle_deg: float, dist: float = 0.65):
    """World->camera (R, t) for a camera on the ring at ``angle_deg`` looking at the origin."""
    a = np.deg2rad(angle_deg)
    C = np.array([dist * np.sin(a), 0.0, -dist * np.cos(a)])
    z = -C / np.linalg.norm(C)
    up = np.array([0.0, -1.0, 0.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    t = -R @ C
    return R, t


def render(scene, R, t, K, w: int, h: int, rng: np.random.Generator | None, background: float = 20.0,
           scale: float = 1.0) -> np.ndarray:
    """Splat every front-facing blob as a Gaussian; returns u8 [h, w]."""
    pts, nrm = scene["pts"], scene["normals"]
    Xc = pts @ R.T + t
    cam_dir = -(R.T @ t)
    facing = (nrm @ (cam_dir / np.linalg.norm(cam_dir))) > 0.15
    z = Xc[:, 2]
    u = K[0, 0] * Xc[:, 0] / z + K[0, 2]
    v = K[1, 1] * Xc[:, 1] / z + K[1, 2]
    img = np.full((h, w), background, np.float64)
    sig_all = scene["sigma"] * scale
    for rad, pick in ((6, sig_all <= 3.0 * scale), (30, sig_all > 3.0 * scale)):  # fine blobs, then the wide ones
        ok = pick & facing & (z > 1e-3) & (u > -rad) & (u < w + rad) & (v > -rad) & (v < h + rad)
        if not ok.any():
            continue
        uu, vv = u[ok], v[ok]
        sig = sig_all[ok]
        amp = scene["amp"][ok]
        u0 = np.floor(uu).astype(np.int64)
        v0 = np.floor(vv).astype(np.int64)
        offs = np.arange(-rad, rad + 1)
        dx, dy = np.meshgrid(offs, offs)
        dx, dy = dx.ravel(), dy.ravel()
        X = u0[:, None] + dx[None, :]
        Y = v0[:, None] + dy[None, :]
        val = amp[:, None] * np.exp(-((X - uu[:, None]) ** 2 + (Y - vv[:, None]) ** 2) / (2.0 * sig[:, None] ** 2))
        inside = (X >= 0) & (X < w) & (Y >= 0) & (Y < h)
        np.add.at(img, (Y[inside], X[inside]), val[inside])
    if rng is not None:
        img += rng.normal(size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_sequence(n_frames: int, w: int = 640, h: int = 480, deg_per_frame: float = 0.3, n_blobs: int = 20000,
                  seed: int = 7, noise: bool = True, K: np.ndarray | None = None, dist: float = 0.65,
                  start_deg: float = 0.0, angles=None, shell_scale: float = 1.0, coarse_frac: float = 0.0):
    """Returns dict(images [F,h,w] u8, K, R [F,3,3], t [F,3], names, lat, lon).
    angles (optional): explicit ring angle in degrees per frame (e.g. out-and-back paths that revisit a view)."""
    if K is None:
        K = K_TEMPLE.copy()
        K[0, :] *= w / 640.0
        K[1, :] *= h / 480.0
    scene = make_scene(n_blobs, seed, shell_scale, coarse_frac)
    rng = np.random.default_rng(seed + 1000) if noise else None
    imgs = np.zeros((n_frames, h, w), np.uint8)
    Rs = np.zeros((n_frames, 3, 3))
    ts = np.zeros((n_frames, 3))
    blob_scale = max(w / 640.0, 0.6)
    if angles is None:
        angles = [start_deg + f * deg_per_frame for f in range(n_frames)]
    assert len(angles) == n_frames
    for f in range(n_frames):
        R, t = ring_pose(float(angles[f]), dist)
        Rs[f], ts[f] = R, t
        imgs[f] = render(scene, R, t, K, w, h, rng, scale=blob_scale)
    names = [f"templeR{f + 1:04d}.png" for f in range(n_frames)]
    lat = np.zeros(n_frames)
    lon = np.array([float(a) for a in angles])
    return dict(images=imgs, K=K, R=Rs, t=ts, names=names, lat=lat, lon=lon)


def write_pgm(path: str, img: np.ndarray) -> None:
    h, w = img.shape
    with open(path, "wb") as f:
        f.write(f"P5\n{w} {h}\n255\n".encode())
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def write_par_ang(root: str, seq) -> None:
    """<root>/templeRing/templeR_par.txt and templeR_ang.txt (the ground truth the evaluators read)."""
    os.makedirs(os.path.join(root, "templeRing"), exist_ok=True)
    n = len(seq["names"])
    with open(os.path.join(root, "templeRing", "templeR_par.txt"), "w") as f:
        f.write(f"{n}\n")
        for i in range(n):
            vals = list(seq["K"].ravel()) + list(seq["R"][i].ravel()) + list(seq["t"][i].ravel())
            f.write(seq["names"][i] + " " + " ".join(repr(float(v)) for v in vals) + "\n")
    with open(os.path.join(root, "templeRing", "templeR_ang.txt"), "w") as f:
        for i in range(n):
            f.write(f"{float(seq['lat'][i])!r} {float(seq['lon'][i])!r} {seq['names'][i]}\n")


def write_dataset(root: str, seq) -> None:
    """Write the reference CLI's input layout under ``root``."""
    write_par_ang(root, seq)
    os.makedirs(os.path.join(root, "templeRing_pgm"), exist_ok=True)
    for i in range(len(seq["names"])):
        stem = os.path.splitext(seq["names"][i])[0]
        write_pgm(os.path.join(root, "templeRing_pgm", stem + ".pgm"), seq["images"][i])
