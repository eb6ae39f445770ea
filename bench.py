#!/usr/bin/env python3
"""bench.py — keyframes/sec of the SfM hot path (KLT + RANSAC + local BA) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole per-frame loop (reference main(), T:1708-1871) over one synthetic
TempleRing-style sequence: 47 frames, 640x480 u8, ring camera, the reference's default config.json.
The frames are resident in HBM before the timed region starts.  With N ranks every rank processes its
own sequence (independent sequences are the unit the path shards on: SURVEY.md §8e) -- weak scaling,
no data-path collective; only the barrier / max-time reduction use RCCL.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (by accumulated GPU time,
measured with HIP events on the kernel's own stream inside the library); `cpu_baseline` is the real
reference binary (oracle/_ref, kind "reference") or, when that is absent, the oracle restatement
(kind "port") timed on a bounded sample of the same workload on this host's cores.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

# One pipeline keeps eight contexts busy.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default
# 4) in creation order; with 8 the BA lane, the tracker lane and the prefetch lane each get a queue of their own
# (measured: +15-20 % keyframes/s).  Must be in the environment before the HIP runtime initialises, i.e. before torch.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "structure-from-motion-3d-reconstruction_amd"

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # SURVEY.md §8d (public spec figure for vector FP64)
LK_STEP_FLOP = 12.7e3       # 121 px x 105 flop + solve
SAMPSON_FLOP = 45.0


def cpu_baseline(seq, cfg, sample_frames: int):
    """Reference CPU path on a bounded sample (first `sample_frames` frames) of the same sequence."""
    synth = importlib.import_module(PKG + ".synth")
    ref_cli = os.path.join(ROOT, "oracle", "_ref", "templering_sfm_ref")
    sub = {k: (v[:sample_frames] if k in ("images", "R", "t", "names", "lat", "lon") else v) for k, v in seq.items()}
    cores = 1  # the reference is single-threaded
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    host = f"{model}, {os.cpu_count()} logical CPUs visible, 1 used"
    if os.path.exists(ref_cli):
        with tempfile.TemporaryDirectory() as td:
            synth.write_dataset(td, sub)
            t0 = time.perf_counter()
            p = subprocess.run([ref_cli, td, os.path.join(td, "out"), str(sample_frames)], capture_output=True, text=True, cwd=td)
            dt = time.perf_counter() - t0
            if p.returncode == 0:
                kf = int(p.stdout.split("Keyframes:")[1].split()[0])
                return dict(value=kf / dt, unit="keyframes/s", cores=cores, kind="reference", host_cpu=host,
                            sample=f"first {sample_frames} frames of the bench sequence, reference CLI wall time {dt:.2f} s, {kf} keyframes",
                            frames_per_s=sample_frames / dt)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        rc, log, nk, npnt = H.orc_pipeline_run(sub["images"], sub["names"], sub["K"], sub["lat"], sub["lon"],
                                               dict(cfg, frames=sample_frames), td)
    dt = time.perf_counter() - t0
    return dict(value=nk / dt, unit="keyframes/s", cores=cores, kind="port", host_cpu=host,
                sample=f"first {sample_frames} frames of the bench sequence, oracle restatement wall time {dt:.2f} s, {nk} keyframes",
                frames_per_s=sample_frames / dt)


def run_ba_sharded(capi, synth, rank, local_rank, world, steps, warmup, ctx=None, comm=None, shard="elements"):
    """BASELINE config C4: local BA on W=10 poses, P=50 000 points, every point seen by every pose (500 k residuals), sharded
    over the ranks; a step = one BA iteration = build of this rank's share of S|b, RCCL all-reduce(sum) of D*D + D = 3 660
    doubles in HBM, damping + gauge, dense solve on the device, dx to the host.
    shard = "elements" (sfmx_ba_step_sharded_elements, the parity mode): every rank holds all points, computes the per-point
    records (replicated) and reduces a disjoint slice of the elements of S|b, the all-reduce adds zeros -- bit-identical to one GPU;
    shard = "points" (sfmx_ba_step_sharded, tolerance mode): every rank holds a contiguous range of the points, the all-reduce
    regroups the addends of every element.
    torch.distributed must be up when world > 1 (it carries the unique id and the max-time reduction).  Returns the result
    dict (meaningful on rank 0)."""
    import torch
    import torch.distributed as dist
    D = importlib.import_module(PKG + ".dist")
    W, P = 10, 50000
    rng = np.random.default_rng(1)
    pw = np.zeros((W, 12))
    for k in range(W):
        R, t = synth.ring_pose(2.0 * k)
        pw[k, :9], pw[k, 9:] = R.ravel(), t
    K = synth.K_TEMPLE
    X = rng.normal(size=(P, 3)) * 0.08
    ptr = np.arange(0, (P + 1) * W, W, dtype=np.int32)
    li = np.tile(np.arange(W, dtype=np.int32), P)
    Xc = np.einsum("kij,pj->pki", pw[:, :9].reshape(W, 3, 3), X) + pw[None, :, 9:]
    uv = np.stack([K[0, 0] * Xc[..., 0] / Xc[..., 2] + K[0, 2], K[1, 1] * Xc[..., 1] / Xc[..., 2] + K[1, 2]], -1)
    uv = np.ascontiguousarray((uv + rng.normal(size=uv.shape) * 0.5).reshape(P * W, 2))
    own_ctx, own_comm = ctx is None, comm is None
    if own_ctx:
        ctx = capi.Context(local_rank)
    if own_comm:
        comm = D.make_comms(1, local_rank)[0]
    lo, hi = capi.shard_range(P, rank, world) if shard == "points" else (0, P)
    o0, o1 = int(ptr[lo]), int(ptr[hi])
    prob = ctx.ba_problem(W, X[lo:hi], ptr[lo:hi + 1] - o0, li[o0:o1], uv[o0:o1])
    a = (pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    step = prob.step_sharded if shard == "points" else prob.step_sharded_elements

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, warmup)):
        rc, dx = step(comm, *a)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        rc, dx = step(comm, *a)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ctx.set_timing(True)
    for _ in range(3):
        step(comm, *a)
    prof = {k: v for k, v in ctx.kernel_profile().items() if v[1] > 0}
    ctx.set_timing(False)
    Pl, R = (hi - lo, (hi - lo) * W) if shard == "points" else ((P + world - 1) // world, (P * W + world - 1) // world)
    # SURVEY.md 8(d) for THIS rank's share (points: its point range; elements: 1 / world of the system), per BA iteration: 535 flop per residual + (126 n + 180 n^2 + 40) per point (n = W poses
    # see the point); bytes 20 R + 24 P + 96 W read, 8 (D^2 + D) written -- the figures the roofline is priced on.  What the
    # kernels move on top of that (the contribution rows) shows up in `kernel_us_per_step`, not in `achieved`.
    Dd = 6 * W
    alg_flop = 535.0 * R + Pl * (126.0 * W + 180.0 * W * W + 40.0)
    alg_bytes = 20.0 * R + 24.0 * Pl + 96.0 * W + 8.0 * (Dd * Dd + Dd)
    step_s = dt / steps
    out = {"metric": f"BA iterations/sec, BASELINE config C4 (W=10 poses, P=50 000 points, 500 k residuals), {shard} of the system sharded over the ranks",
           "value": round(steps / dt, 3), "unit": "iterations/s", "n_gpus": world, "rccl_world": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(step_s * 1e3, 4), "higher_is_better": True, "scaling": "strong", "mode": "ba-sharded", "shard": shard,
           "bit_identical_to_one_gpu": shard == "elements" or world == 1, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "local BA S|b build + reduce + dense solve, W=10, P=50000, every point in every pose, N(0,0.5 px) noise",
                      "points_per_rank": hi - lo, "allreduce_doubles": Dd * Dd + Dd, "allreduce_bytes_per_step": 8 * (Dd * Dd + Dd),
                      "parallelism": f"BA {shard} x{world}, RCCL all-reduce(sum) of S|b in HBM"},
           "roofline": {"bound": "valu_fp64", "achieved": round(alg_flop / step_s / 1e12, 4), "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(alg_flop / step_s / 1e12 / FP64_VALU_PEAK_TF, 5), "traffic": None,
                        "algorithmic_flop_per_step": int(alg_flop), "algorithmic_bytes_per_step": int(alg_bytes),
                        "hbm": {"achieved": round(alg_bytes / step_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(alg_bytes / step_s / 1e9 / HBM_PEAK_GBS, 6)},
                        "note": "whole iteration (all kernels + host round trip) against SURVEY.md 8(d)'s figures for this rank's shard; "
                                "`traffic` (FETCH_SIZE + WRITE_SIZE per iteration) is in profiles/ when a PMC pass was taken",
                        "kernel_us_per_step": {k: round(v[0] / v[1], 2) for k, v in prof.items()}},
           "dx_head": [float(v) for v in dx[6:9]]}
    prob.close()
    if own_comm:
        comm.close()
    if own_ctx:
        ctx.close()
    return out


def bench_ba_sharded(args, capi, synth, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    out = run_ba_sharded(capi, synth, rank, local_rank, world, args.steps, args.warmup, shard=args.ba_shard)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=47)
    ap.add_argument("--deg-per-frame", type=float, default=0.3)
    ap.add_argument("--max-tracks", type=int, default=2200)
    ap.add_argument("--cpu-sample-frames", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sequences-per-gpu", type=int, default=1,
                    help="independent sequences in flight on each GPU (a batch of S sequences per step); the headline uses 1")
    ap.add_argument("--batched-probe", type=int, default=3,
                    help="after the headline measurement (1 sequence per GPU) also time this many sequences in flight and "
                         "report it as `batched` (single-GPU runs only; 0 = skip)")
    ap.add_argument("--sharded-probe", type=int, default=1,
                    help="default mode: after the headline also time the sharded BA step (C4) and one sequence on all ranks and "
                         "report them as `sharded_ba` / `sharded_sequence` (0 = skip)")
    ap.add_argument("--ba-shard", choices=("elements", "points"), default="elements",
                    help="--mode ba-sharded: elements (parity mode, sfmx_ba_step_sharded_elements) or points (tolerance mode)")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="watchdog of the sharded sub-benchmarks, seconds")
    ap.add_argument("--mode", choices=("sequences", "ba-sharded", "sharded-sequence"), default="sequences",
                    help="sequences (headline): one independent sequence per rank, weak scaling, no data-path collective; "
                         "ba-sharded: BASELINE config C4 (W=10, P=50 000, 500 k residuals) with the points sharded over the ranks and one "
                         "RCCL all-reduce of S|b per iteration (strong scaling); sharded-sequence: ONE sequence on all ranks, BA points "
                         "and RANSAC hypotheses sharded (strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)

    capi = importlib.import_module(PKG + ".capi")
    pipe = importlib.import_module(PKG + ".pipeline")
    synth = importlib.import_module(PKG + ".synth")
    if args.mode == "ba-sharded":
        return bench_ba_sharded(args, capi, synth, rank, local_rank, world)

    # --- synthetic TempleRing-47 stand-in: S independent sequences per rank, each with its own context and host thread
    # (one sequence alone cannot fill the device: its kernels are short dependent chains)
    import threading
    S = max(1, args.sequences_per_gpu)
    if args.mode == "sharded-sequence":
        S = 1  # one job on all ranks: a communicator serves ONE pipeline (its collectives must be issued in one order)
    cfg = dict(pipe.DEFAULTS, frames=args.frames, max_tracks=args.max_tracks, min_tracks=min(900, args.max_tracks * 9 // 22),
               export_pointcloud=0)
    seqs, ctxs, devs = [], [], []

    def ensure_sequences(n):  # contexts are created only when used: idle streams still take part in the HW-queue mapping
        while len(seqs) < n:
            q = len(seqs)
            seed = 7 + 101 * q + (0 if args.mode == "sharded-sequence" else rank)  # sharded-sequence: every rank holds THE sequence
            seqs.append(synth.make_sequence(args.frames, 640, 480, args.deg_per_frame, n_blobs=20000, seed=seed))
            ctxs.append(capi.Context(local_rank))
            devs.append(torch.from_numpy(np.ascontiguousarray(seqs[q]["images"])).to(f"cuda:{local_rank}"))  # resident in HBM
        torch.cuda.synchronize()

    ensure_sequences(S)
    seq, ctx, frames_dev = seqs[0], ctxs[0], devs[0]
    shape = tuple(frames_dev.shape)

    comms = None  # sharded-sequence: two native RCCL communicators (BA lane; RANSAC merges of the geometry thread)

    def one_pass(timing=False, q=0):
        return pipe.run(ctxs[q], None, seqs[q]["names"], seqs[q]["K"], seqs[q]["lat"], seqs[q]["lon"], cfg, None,
                        images_dev=devs[q].data_ptr(), shape=shape, timing=timing, comms=comms)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The pipeline's contexts (and with them their streams) are created by the first pass.  RCCL comes up only after that,
    # so that a rank's streams get the same hardware queues as in a single-process run (csrc/hip/ctx.hip).
    one_pass()
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.mode == "sharded-sequence":
        comms = tuple(importlib.import_module(PKG + ".dist").make_comms(2, local_rank))

    def measure(n_seq, n_warm, n_steps):
        """n_seq sequences in flight (one host thread + context each): wall time of n_steps passes of every sequence.
        Every pass of a sequence must agree bit for bit: the host lanes reorder WHEN work happens, never WHAT comes out."""
        ensure_sequences(n_seq)
        result = [dict(kf=0, identical=True, last=None, error=None) for _ in range(n_seq)]

        def worker(q, n_pass, count):
            try:
                for _ in range(n_pass):
                    cur = one_pass(q=q)
                    r = result[q]
                    if count:
                        if r["last"] is not None:
                            r["identical"] = r["identical"] and cur["log"] == r["last"]["log"] and np.array_equal(
                                cur["centres"].view(np.uint64), r["last"]["centres"].view(np.uint64))
                        r["kf"] += cur["stats"]["n_keyframes"]
                    r["last"] = cur
            except Exception as e:  # surfaced after the join
                result[q]["error"] = e

        def run_all(n_pass, count):
            if n_seq == 1:
                worker(0, n_pass, count)
            else:
                th = [threading.Thread(target=worker, args=(q, n_pass, count)) for q in range(n_seq)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
            for r in result:
                if r["error"] is not None:
                    raise r["error"]

        run_all(n_warm, False)
        barrier()
        t_start = time.perf_counter()
        run_all(n_steps, True)
        for c in ctxs[:n_seq]:
            c.sync()
        barrier()
        wall = time.perf_counter() - t_start
        return wall, sum(r["kf"] for r in result), all(r["identical"] for r in result), result[0]["last"]

    dt, kf_total, identical, last = measure(S, args.warmup, args.steps)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if args.mode != "sharded-sequence":  # there every rank reports the keyframes of the same one job
            k = torch.tensor([kf_total], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(k, op=dist.ReduceOp.SUM)
            kf_total = int(k.item())

    # --- one extra, untimed pass with per-kernel HIP-event timing for the roofline object
    prof = one_pass(timing=True)["stats"]

    # --- the north_star's split, measured by the SAME command with the same process group (default mode only): BA points
    # sharded with an RCCL all-reduce of S | b per iteration (BASELINE config C4), and ONE sequence on all ranks with BA points
    # and RANSAC hypotheses sharded.  With one rank the same code runs with world-size-1 communicators (no RCCL): the schema
    # and the code path are exercised by every 1-GPU run.  Both run under a watchdog: a collective that never completes must
    # not cost the headline line (the sub-object then says so and the process leaves without the RCCL teardown).
    sharded = {"sharded_ba": None, "sharded_sequence": None}
    sub_failed = []
    if args.mode == "sequences" and args.sharded_probe:
        def sub_benchmarks():
            Dm = importlib.import_module(PKG + ".dist")
            # C4 through sfmx_ba_step_sharded
            try:
                sharded["sharded_ba"] = run_ba_sharded(capi, synth, rank, local_rank, world, 10, 2, shard="elements")
                if world > 1:  # the tolerance mode next to it (with one rank both are the plain step)
                    pts = run_ba_sharded(capi, synth, rank, local_rank, world, 10, 2, shard="points")
                    sharded["sharded_ba"]["tolerance_mode_points"] = {k: pts[k] for k in ("value", "unit", "ms_per_step", "shard", "bit_identical_to_one_gpu")}
            except Exception as e:
                sharded["sharded_ba"] = {"error": repr(e), "rccl_world": world}
            # one sequence on all ranks: every rank holds rank 0's sequence
            try:
                if rank == 0:
                    sq, dv = seqs[0], devs[0]
                else:
                    sq = synth.make_sequence(args.frames, 640, 480, args.deg_per_frame, n_blobs=20000, seed=7)
                    dv = torch.from_numpy(np.ascontiguousarray(sq["images"])).to(f"cuda:{local_rank}")
                    torch.cuda.synchronize()
                cm = tuple(Dm.make_comms(2, local_rank))

                def sh_pass():
                    return pipe.run(ctx, None, sq["names"], sq["K"], sq["lat"], sq["lon"], cfg, None, images_dev=dv.data_ptr(),
                                    shape=tuple(dv.shape), comms=cm)
                first = sh_pass()
                barrier()
                t0 = time.perf_counter()
                n_sh = max(2, min(args.steps, 5))
                same = True
                for _ in range(n_sh):
                    cur = sh_pass()
                    same = same and cur["log"] == first["log"] and np.array_equal(cur["centres"].view(np.uint64), first["centres"].view(np.uint64))
                ctx.sync()
                barrier()
                sdt = time.perf_counter() - t0
                agree = True
                if world > 1:
                    tt = torch.tensor([sdt], dtype=torch.float64, device=f"cuda:{local_rank}")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    sdt = float(tt.item())
                    # every rank must hold the same keyframe centres, bit for bit (the collectives return the same bytes everywhere)
                    cc = torch.from_numpy(np.ascontiguousarray(first["centres"]).view(np.int64).copy()).to(f"cuda:{local_rank}")
                    lo_, hi_ = cc.clone(), cc.clone()
                    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
                    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
                    agree = bool(torch.equal(lo_, hi_))
                stt = first["stats"]
                ba_it = stt["ba_iters"]
                sharded["sharded_sequence"] = {
                    "value": round(stt["n_keyframes"] * n_sh / sdt, 3), "unit": "keyframes/s", "rccl_world": world, "steps": n_sh,
                    "ms_per_step": round(sdt / n_sh * 1e3, 3), "scaling": "strong", "passes_bit_identical": bool(same),
                    "ranks_bit_identical": agree, "keyframes_per_step": stt["n_keyframes"],
                    "collectives_per_step": {"ba_allreduce_sum": int(ba_it), "ba_allreduce_bytes_each": 8 * (36 * 36 + 36),
                                             "ransac_merges": int(stt["ransac_calls"]), "ransac_merge_bytes_each": 8 + 72},
                    "same_result_as_unsharded": bool(rank != 0 or (first["log"] == last["log"] and np.allclose(
                        first["centres"], last["centres"], rtol=1e-6, atol=1e-6 * max(1.0, float(np.abs(last["centres"]).max()))))),
                    "parallelism": f"one sequence on {world} rank(s): BA points + RANSAC hypotheses sharded; comm_ba on lane B, comm_ransac on the geometry thread"}
                for m in cm:
                    m.close()
            except Exception as e:
                sharded["sharded_sequence"] = {"error": repr(e), "rccl_world": world}

        th = threading.Thread(target=sub_benchmarks, daemon=True)
        th.start()
        th.join(args.sharded_timeout)
        if th.is_alive():
            for k in sharded:
                if sharded[k] is None:
                    sharded[k] = {"error": f"no result within {args.sharded_timeout} s (watchdog)", "rccl_world": world}
                    sub_failed.append(k)
    batched = None
    if world == 1 and S == 1 and args.batched_probe > 1:  # how much more the device takes with several sequences in flight
        b_steps = max(2, min(args.steps, 8))
        b_dt, b_kf, b_same, _ = measure(args.batched_probe, 1, b_steps)
        batched = {"sequences_per_gpu": args.batched_probe, "value": round(b_kf / b_dt, 3), "unit": "keyframes/s", "steps": b_steps,
                   "ms_per_step": round(b_dt / b_steps * 1e3, 3), "passes_bit_identical": bool(b_same)}
    if rank == 0:
        st = last["stats"]
        # --- roofline of the DOMINANT kernel of this run: the kernel with the largest accumulated GPU time in the
        # per-kernel profile of the extra timed pass (HIP events recorded inside libsfmx on each context's own stream
        # around every launch, summed over the contexts of the pipeline).  Nothing here is read from a previous
        # round's files; `traffic` comes from the newest committed PMC summary only if it was taken at this launch size.
        kern = {k: v for k, v in prof["kernels"].items() if v[1] > 0}
        w, h = 640, 480
        n_klt = max(1, prof["klt_calls"])
        tracks = prof["tracks_in"] / n_klt
        npts = prof["ransac_points"] / max(1, prof["ransac_calls"])
        # algorithmic (flop, bytes) per launch -- SURVEY.md 8(d) figures x the units one launch processes
        alg = {
            "k_klt_track": (LK_STEP_FLOP * prof["lk_steps"] / n_klt, 2 * 1.3125 * w * h + 49.0 * tracks),
            "k_hypotheses": (25e3 * 2500, 256.0 * 2500 + 80.0 * 2500),
            "k_score": (SAMPSON_FLOP * 2500 * npts, 32.0 * npts + 84.0 * 2500),
            "k_ba_points": (535.0 * 3600, 20.0 * 3600 + 24.0 * 600 + 96 * 6),
            "k_ba_reduce": (1.0 * 600 * (36 * 36 + 36), 8.0 * 600 * (36 * 36 + 36 + 48)),
            "k_shi_score": (260.0 * w * h, 9.0 * w * h),
            "k_downsample2": (4.0 * 0.3125 * w * h, 1.3125 * w * h),
            "solve": (2.0 / 3.0 * 36 ** 3, 8.0 * (36 * 36 + 2 * 36)),
        }
        # "shi fixpoint" is a GROUP of ~20 launches timed as one interval (and, with timing on, issued without the hipGraph that
        # normally replays them): it is reported, but the roofline object is about a single kernel
        single = {k: v for k, v in kern.items() if not k.startswith("shi fixpoint")}
        dom = max(single, key=lambda k: single[k][0]) if single else "k_klt_track"
        dom_us, dom_calls = kern.get(dom, (0.0, 0))
        avg_us = dom_us / dom_calls if dom_calls else 0.0
        alg_flop, alg_bytes = alg.get(dom.split(" ")[0], (None, None))
        ach_tf = alg_flop / (avg_us * 1e-6) / 1e12 if (alg_flop and avg_us > 0) else None
        ach_gbs = alg_bytes / (avg_us * 1e-6) / 1e9 if (alg_bytes and avg_us > 0) else None
        traffic, valu_issue = None, None
        try:  # newest profiles/rNN_pmc_summary.json + its meta (launch sizes of tools/prof_kernels.py)
            import glob
            metas = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_meta.json")))
            if metas:
                meta = json.load(open(metas[-1]))
                pmc = json.load(open(metas[-1].replace("_pmc_meta.json", "_pmc_summary.json")))
                same_size = dom != "k_klt_track" or abs(meta.get("klt_tracks", 0) - tracks) <= 0.1 * tracks
                for row in pmc:
                    if row["kernel"].replace("void ", "").startswith(dom.split(" ")[0]) and same_size:
                        # FETCH_SIZE / WRITE_SIZE are KB; byte-granular loads are uncalibrated on gfx950 (MI355X_MICROARCH.md,
                        # HBM), so the raw counter sum is what is reported
                        traffic = int((row.get("FETCH_SIZE_avg", 0.0) + row.get("WRITE_SIZE_avg", 0.0)) * 1024)
                        if "SQ_INSTS_VALU_avg" in row and dom == "k_klt_track" and meta.get("klt_lk_steps"):
                            valu_issue = dict(source=os.path.basename(metas[-1]).replace("_pmc_meta.json", "_pmc_summary.json"),
                                              valu_wave_insts_per_lk_step=round(row["SQ_INSTS_VALU_avg"] / meta["klt_lk_steps"], 1),
                                              waves=int(row.get("SQ_WAVES_avg", 0)),
                                              # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES both count quad-cycles summed over the waves:
                                              # the share of a wave's lifetime in which it is issuing vector instructions
                                              valu_active_over_wave_cycles=(round(row["SQ_ACTIVE_INST_VALU_avg"] / row["SQ_WAVE_CYCLES_avg"], 4)
                                                                            if "SQ_ACTIVE_INST_VALU_avg" in row and row.get("SQ_WAVE_CYCLES_avg") else None),
                                              issue_stalled_over_wave_cycles=(round(row["SQ_WAIT_INST_ANY_avg"] / row["SQ_WAVE_CYCLES_avg"], 4)
                                                                              if "SQ_WAIT_INST_ANY_avg" in row and row.get("SQ_WAVE_CYCLES_avg") else None))
        except Exception:
            traffic, valu_issue = None, None
        roofline = dict(bound="valu_fp64", achieved=None if ach_tf is None else round(ach_tf, 4), peak=FP64_VALU_PEAK_TF, unit="TFLOP/s",
                        frac=None if ach_tf is None else round(ach_tf / FP64_VALU_PEAK_TF, 5), traffic=traffic,
                        kernel=dom, avg_launch_us=round(avg_us, 2), launches_per_pass=int(dom_calls),
                        algorithmic_flop_per_launch=None if alg_flop is None else int(alg_flop),
                        algorithmic_bytes_per_launch=None if alg_bytes is None else int(alg_bytes),
                        hbm=dict(achieved=None if ach_gbs is None else round(ach_gbs, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=None if ach_gbs is None else round(ach_gbs / HBM_PEAK_GBS, 6)),
                        valu_issue=valu_issue,
                        note="ordered FP64 sums make this path VALU-issue / dependent-latency bound, not HBM bound (SURVEY.md 8d): frac is "
                             "algorithmic FP64 flop/s over the vector FP64 peak, `hbm` the same launch against the HBM roof",
                        kernel_us_per_pass={k: round(v[0], 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])},
                        kernel_launches_per_pass={k: v[1] for k, v in kern.items()})
        # the same kernel with the device to itself (the pipeline runs 2-3 kernels side by side, which stretches each of them):
        # KLT of the first frame pair at this run's launch size, HIP events inside the library
        if dom == "k_klt_track":
            try:
                trk = pipe.Tracker(ctx, w, h, max_tracks=args.max_tracks)
                trk.step(seq["images"][0])
                pts, _ = trk.tracks()
                trk.close()
                pts = np.ascontiguousarray(pts[:max(1, int(round(tracks)))])
                pa, pb = ctx.pyramid(seq["images"][0], 3), ctx.pyramid(seq["images"][1], 3)
                ctx.set_timing(True)
                us, steps_alone = [], 0
                for _ in range(7):
                    _, _, _, steps_alone = ctx.klt_track(pa, pb, pts)
                    us.append(ctx.last_kernel_us())
                ctx.set_timing(False)
                us_med = float(np.median(us))
                tf = LK_STEP_FLOP * steps_alone / (us_med * 1e-6) / 1e12
                roofline["alone"] = dict(tracks=int(len(pts)), lk_steps=int(steps_alone), avg_launch_us=round(us_med, 2), achieved=round(tf, 4),
                                         frac=round(tf / FP64_VALU_PEAK_TF, 5))
            except Exception as e:  # the line must still be printed
                roofline["alone"] = {"error": repr(e)}
        out = {
            "metric": "keyframes/sec (KLT + RANSAC + local BA per-frame loop), synthetic TempleRing-47 stand-in",
            "value": round(kf_total / dt, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.mode == "sharded-sequence" else "weak", "mode": args.mode, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic TempleRing-47 stand-in: {args.frames} frames 640x480 u8, ring camera {args.deg_per_frame} deg/frame, "
                                   f"reference default config (max_tracks={args.max_tracks}, RANSAC 2500 iters, BA window 6 / 600 pts / 5 iters); "
                                   f"{S} independent sequence(s) in flight per GPU", "frames_per_step": args.frames * S, "sequences_per_gpu": S,
                       "parallelism": (f"one sequence on {world} ranks: BA points + RANSAC hypotheses sharded, RCCL all-reduce"
                                       if args.mode == "sharded-sequence" else f"sequences x{world * S}")},
            "frames_per_s": round(args.frames * args.steps * (1 if args.mode == "sharded-sequence" else world) * S / dt, 2),
            "keyframes_per_step": int(round(kf_total / max(1, args.steps) / (1 if args.mode == "sharded-sequence" else world))), "map_points": st["n_points"], "passes_bit_identical": bool(identical),
            "host_seconds_per_step": {k: round(st[k], 4) for k in ("sec_wall", "sec_total", "sec_setup", "sec_klt", "sec_shi", "sec_shi_wait", "sec_shi_gpu", "sec_shi_replay", "sec_ransac", "sec_ba", "sec_upload", "sec_host", "sec_desc", "sec_bookkeeping", "sec_r_pre", "sec_r_gpu", "sec_r_verify", "sec_r_decomp", "sec_tri_iter", "sec_tri_solve", "sec_tri_insert", "sec_pf_busy", "sec_pf_gpu", "sec_pf_replay", "sec_lane_a_busy", "sec_lane_b_busy", "sec_lane_c_busy", "sec_lane_e_busy", "sec_join_wait", "sec_ba_gather", "sec_m_step", "sec_m_ransac", "sec_m_kf", "sec_feed_wait")},
            "counters_per_step": {k: int(st[k]) for k in ("klt_calls", "tracks_in", "lk_steps", "ransac_calls", "ransac_points", "ransac_verified", "ransac_cert_misses", "ba_calls", "ba_iters", "shi_calls", "shi_memo_hits", "shi_prefetched", "shi_fallbacks")},
            "roofline": roofline,
        }
        # ATE-RMSE of the keyframe centres against the synthetic ground truth, stated by the build's own evaluator
        # (structure-from-motion-3d-reconstruction_amd/_build/ate_keyframes; its digits are pinned to the reference tool's
        # in tests/test_tools.py) on the CSV of one extra, untimed pass
        out["batched"] = batched
        out.update(sharded)
        out["ate_rmse_sim3_vs_gt"] = None
        try:
            import subprocess, tempfile
            with tempfile.TemporaryDirectory() as td:
                pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, os.path.join(td, "out"),
                         images_dev=frames_dev.data_ptr(), shape=shape)
                synth.write_par_ang(td, seq)
                tool = os.path.join(ROOT, PKG, "_build", "ate_keyframes")
                r = subprocess.run([tool, "--par", os.path.join(td, "templeRing", "templeR_par.txt"), "--keyframes",
                                    os.path.join(td, "out", "keyframes_camera_centers.csv"), "--count", str(st["n_keyframes"])],
                                   capture_output=True, text=True)
                for line in r.stdout.splitlines():
                    if line.strip().startswith("ATE_RMSE:"):
                        out["ate_rmse_sim3_vs_gt"] = float(line.split(":")[1])
                    if line.strip().startswith("scale (s):"):
                        out["ate_sim3_scale"] = float(line.split(":")[1])
        except Exception as e:  # the evaluator is reporting only
            out["ate_error"] = str(e)
        if not args.no_cpu_baseline:
            cb = cpu_baseline(seq, cfg, min(args.cpu_sample_frames, args.frames))
            out["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in cb.items()}
            out["speedup_vs_cpu_baseline"] = round(out["value"] / cb["value"], 1) if cb["value"] > 0 else None
        print(json.dumps(out))
    sys.stdout.flush()
    if sub_failed:  # a collective is stuck somewhere: no barrier, no RCCL teardown -- the line is out, leave
        os._exit(0)
    if world > 1:  # everything is measured and printed: a rank that cannot finish the teardown (a peer left early) just leaves
        def teardown():
            dist.barrier()
            dist.destroy_process_group()
        th = threading.Thread(target=teardown, daemon=True)
        th.start()
        th.join(60.0)
        if th.is_alive():
            os._exit(0)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
