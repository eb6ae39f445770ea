"""GPU suite, multi-GPU part: runs only where more than one MI355X is visible (the 1-GPU test box skips it; the
arithmetic of the sharded modes is covered on one GPU in test_gpu_kernels.py and over gloo in test_dist_cpu.py).
One process per GPU, native RCCL communicators (csrc/hip/comm.hip), torch.distributed only carries the unique ids."""
import importlib
import json
import os
import socket

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        capi = importlib.import_module(H.PKG_NAME + ".capi")
        pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
        D = importlib.import_module(H.PKG_NAME + ".dist")
        from test_gpu_kernels import _c4_like_problem
        ctx = capi.Context(rank)
        comm_ba, comm_r = D.make_comms(2, rank)
        # ---- point-sharded BA iteration at a C4-like size: every rank gets the same dx, within 1e-9 of one GPU
        W, P = 10, 20000
        pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 5)
        lo, hi = capi.shard_range(P, rank, world)
        o0, o1 = int(ptr[lo]), int(ptr[hi])
        shard = ctx.ba_problem(W, X[lo:hi], ptr[lo:hi + 1] - o0, li[o0:o1], uv[o0:o1])
        rc, dx = shard.step_sharded(comm_ba, pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
        assert rc == 0
        full = ctx.ba_problem(W, X, ptr, li, uv)
        rc1, dx1 = full.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
        assert rc1 == 0 and np.allclose(dx, dx1, rtol=0, atol=1e-7 * np.abs(dx1).max()), np.abs(dx - dx1).max()
        t = torch.from_numpy(dx).to(f"cuda:{rank}")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(e.view(torch.int64), every[0].view(torch.int64)) for e in every), "dx differs between ranks"
        # ---- one sequence on all ranks, BA points and RANSAC hypotheses sharded: ranks agree byte for byte; everything that
        # depends on RANSAC only (stdout, edges) is the single-GPU output because hypothesis sharding is exact
        g = np.load(os.path.join(H.GOLDEN, "e2e_keyframes.npz"))
        cfg = H.pipe_cfg_from_json(json.loads(str(g["config"])))
        names = [str(s) for s in g["names"]]
        out = os.path.join(out_dir, f"sharded{rank}")
        r = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, out, comms=(comm_ba, comm_r))
        single = os.path.join(out_dir, f"single{rank}")
        r1 = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, single)
        assert r["log"].replace(out, "X") == r1["log"].replace(single, "X")
        assert open(os.path.join(out, "posegraph_edges.csv")).read() == open(os.path.join(single, "posegraph_edges.csv")).read()
        c, c1 = r["centres"], r1["centres"]
        assert c.shape == c1.shape and np.allclose(c, c1, rtol=1e-6, atol=1e-6 * max(1.0, np.abs(c1).max()))
        t = torch.from_numpy(np.ascontiguousarray(c)).to(f"cuda:{rank}")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(e.view(torch.int64), every[0].view(torch.int64)) for e in every), "keyframe centres differ between ranks"
        for m in (comm_ba, comm_r):
            m.close()
        ctx.close()
        with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs at least two GPUs")
def test_two_rank_rccl_sharded_ba_and_pipeline(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_pipeline_with_world_size_one_communicators_is_the_plain_pipeline(tmp_path):
    """The sharded code path with one rank (no RCCL needed) writes the bytes of the unsharded run."""
    capi = importlib.import_module(H.PKG_NAME + ".capi")
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    ctx = capi.Context(0)
    g = np.load(os.path.join(H.GOLDEN, "e2e_loop.npz"))
    cfg = H.pipe_cfg_from_json(json.loads(str(g["config"])))
    names = [str(s) for s in g["names"]]
    a = str(tmp_path / "a")
    r1 = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, a)
    for k in range(2):
        comms = tuple(capi.Comm(0, None, 0, 1) for _ in range(2))
        b = str(tmp_path / f"b{k}")
        r2 = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, b, comms=comms)
        assert r1["log"].replace(a, "X") == r2["log"].replace(b, "X")
        for fn in ("keyframes_camera_centers.csv", "posegraph_edges.csv", "templeRing_sparse_points.ply"):
            assert open(os.path.join(a, fn)).read() == open(os.path.join(b, fn)).read(), fn
        for m in comms:
            m.close()
    ctx.close()
