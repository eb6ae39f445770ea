#!/usr/bin/env python3
"""Fixtures for the sparse-mesh export (tests/test_host_math.py::test_sparse_mesh_matches_reference): inputs and the
vertices / faces the REAL reference function build_mesh_from_sparse_points (T:1384-1461) returns, called through
oracle/_ref/libsfmref.so (make -C oracle ref).  Only runs where /root/reference exists; tests/golden/mesh.npz is committed."""
import importlib, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
synth = importlib.import_module(H.PKG_NAME + ".synth")


def run(lib, prefix, K, pose12, X, w, h, max_points, grid_px, max_edge_px):
    cap = max(len(X), 1)
    v = np.zeros((cap, 3))
    f = np.zeros((cap * 4, 3), np.int32)
    nf = np.zeros(1, np.int32)
    nv = lib.call(prefix + "sparse_mesh", int, H.f64(K), H.f64(pose12), H.f64(X), len(X), w, h, max_points, grid_px, float(max_edge_px), v, cap,
                  f, cap * 4, nf)
    return v[:nv].copy(), f[:int(nf[0])].copy()


def cases():
    rng = np.random.default_rng(5)
    K = synth.K_TEMPLE
    R, t = synth.ring_pose(3.0)
    pose = np.concatenate([R.T.ravel(), -R.T @ t])  # camera->world (R, centre), as Keyframe::pose stores it
    scene = synth.make_scene(4000, 3)
    yield "shell", K, pose, scene["pts"], 640, 480, 2500, 4, 80.0
    yield "capped", K, pose, scene["pts"], 640, 480, 300, 4, 80.0
    yield "coarse_grid", K, pose, scene["pts"][:1500], 640, 480, 2500, 16, 40.0
    # coincident projections and collinear runs: exactly cocircular / degenerate configurations
    grid = np.stack(np.meshgrid(np.linspace(-0.06, 0.06, 25), np.linspace(-0.05, 0.05, 21)), -1).reshape(-1, 2)
    Xg = np.concatenate([grid, np.zeros((len(grid), 1))], 1)
    yield "lattice", K, pose, Xg, 640, 480, 2500, 1, 200.0
    yield "sixty", K, pose, scene["pts"][:60], 640, 480, 2500, 4, 80.0
    yield "too_few", K, pose, scene["pts"][:40], 640, 480, 2500, 4, 80.0
    yield "thinned_below_50", K, pose, scene["pts"][:400], 640, 480, 2500, 400, 80.0   # vertices kept, no faces
    yield "behind", K, pose, scene["pts"] + np.array([0, 0, -5.0]), 640, 480, 2500, 4, 80.0


def main():
    r = H.ref()
    assert r is not None, "oracle/_ref/libsfmref.so missing: run `make -C oracle ref` in the build container"
    out = {}
    names = []
    for name, K, pose, X, w, h, mp, gp, me in cases():
        v, f = run(r, "ref_", K, pose, np.ascontiguousarray(X), w, h, mp, gp, me)
        print(name, "points", len(X), "-> vertices", len(v), "faces", len(f))
        names.append(name)
        out[name + "_K"] = K; out[name + "_pose"] = pose; out[name + "_X"] = np.ascontiguousarray(X)
        out[name + "_cfg"] = np.array([w, h, mp, gp, me], np.float64)
        out[name + "_v"] = v; out[name + "_f"] = f
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "mesh.npz"), **out)


if __name__ == "__main__":
    main()
