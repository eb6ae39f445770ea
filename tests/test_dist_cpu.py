"""CPU suite: the N>1 path with gloo, world size 2.  The sharding / reduction logic of dist.py is exercised with the
oracle as the per-shard compute (the HIP kernels need a GPU); the GPU box runs the same logic over RCCL."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

D = importlib.import_module(H.PKG_NAME + ".dist")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O = H.oracle()
        g = np.load(os.path.join(H.GOLDEN, "hotpath.npz"))
        # ---- BA: point-sharded partial sums + all-reduce == unsharded build (to rounding)
        W, P = 6, 80
        K = g["ba_K_6_80"]
        rng = np.random.default_rng(3)
        synth = importlib.import_module(H.PKG_NAME + ".synth")
        pw = np.zeros((W, 12))
        for k in range(W):
            R, t = synth.ring_pose(2.0 * k)
            pw[k, :9], pw[k, 9:] = R.ravel(), t
        X = rng.normal(size=(P, 3)) * 0.05
        ptr = np.arange(0, (P + 1) * W, W, dtype=np.int32)
        li = np.tile(np.arange(W, dtype=np.int32), P)
        uv = np.zeros((P * W, 2))
        for p in range(P):
            for k in range(W):
                Xc = pw[k, :9].reshape(3, 3) @ X[p] + pw[k, 9:]
                uv[p * W + k] = [K[0, 0] * Xc[0] / Xc[2] + K[0, 2] + rng.normal(), K[1, 1] * Xc[1] / Xc[2] + K[1, 2] + rng.normal()]

        def build(Xs, ps, ls, us, damp):
            S, b = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
            O.call("orc_ba_build", None, H.f64(pw), W, H.f64(Xs), len(Xs), H.i32(ps), H.i32(ls), H.f64(us), float(K[0, 0]), float(K[1, 1]),
                   float(K[0, 2]), float(K[1, 2]), 3.0, 1e-3, int(damp), S, b)
            return S, b
        Xs, ps, ls, us = D.shard_ba_points(X, ptr, li, uv, rank, world)
        lo, hi = D.shard_range(P, rank, world)
        assert len(Xs) == hi - lo and ps[0] == 0 and ps[-1] == len(ls)
        S, b = build(Xs, ps, ls, us, False)
        St, bt = torch.from_numpy(S), torch.from_numpy(b)
        D.allreduce_normal_equations(St, bt)
        D.damp_and_gauge(St, bt, 1e-3)
        Sf, bf = build(X, ptr, li, uv, True)
        assert np.allclose(St.numpy(), Sf, rtol=1e-9, atol=1e-9 * np.abs(Sf).max())
        assert np.allclose(bt.numpy(), bf, rtol=1e-9, atol=1e-9 * np.abs(bf).max())
        assert np.array_equal(bt.numpy()[:6], np.zeros(6))
        # ---- RANSAC: hypothesis-sharded counts + all-reduce(max) == global first maximum
        xi, xj, E = g["tv_xi"], g["tv_xj"], g["tv_E"]
        Hn = len(E)
        cnt = np.zeros(Hn, np.int32)
        O.call("orc_ransac_counts", None, H.f64(xi), H.f64(xj), len(xi), H.f64(E), Hn, 1e-3, cnt)
        cnt[7] = cnt.max()  # force a tie between two iterations on different ranks
        cnt[70] = cnt.max()
        lo, hi = D.shard_range(Hn, rank, world)
        local = cnt[lo:hi]
        bi = int(np.argmax(local))
        bc, bit = D.allreduce_best_hypothesis(int(local[bi]), lo + bi)
        assert (bc, bit) == (int(cnt.max()), int(np.argmax(cnt)))
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 600, 50001):
        for world in (1, 2, 3, 8):
            r = [D.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert D.unpack_best(D.pack_best(123, 45)) == (123, 45)
    assert D.pack_best(10, 3) > D.pack_best(10, 4) > D.pack_best(9, 0)


def test_two_rank_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_native_shard_range_equals_python():
    """sfmx_shard_range (libsfmx: what the native sharded BA / RANSAC use) == dist.shard_range (what the gloo test covers)"""
    capi = importlib.import_module(H.PKG_NAME + ".capi")
    for n in (0, 1, 7, 600, 2500, 50001):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert capi.shard_range(n, rank, world) == D.shard_range(n, rank, world)
