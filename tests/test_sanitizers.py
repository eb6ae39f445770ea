"""CPU suite: the HOST runtime (csrc/host: tracker lane, prefetch workers, lanes A/B/C/E, thread pool, arena, polled joins) under
ThreadSanitizer and AddressSanitizer + UBSan, and the multi-process sharded mode with REAL processes.

The kernels need a GPU, so the sanitizer builds link tests/fake_sfmx/fake_sfmx.cpp -- a CPU stand-in for libsfmx.so whose
arithmetic is the oracle's and whose sfmx_comm_* is an all-reduce through POSIX shared memory (test infrastructure; the product
never loads it).  What is under test is everything ABOVE the C ABI, compiled from the product's own sources with -fsanitize:
 * all lanes on, e2e_loop (loop closures, pose graph, second BA): no data race / lock-order report, output = the reference CLI's;
 * two and three ranks (SFMX_DIST_WORLD): BA elements and RANSAC hypotheses sharded, two communicators -- every collective is
   issued in the same order on every rank, the run ends, and rank 0 writes the bytes of the one-rank run;
 * point-sharded BA (SFMX_BA_SHARD=points, tolerance mode) still agrees on everything RANSAC decides (stdout, edges)."""
import importlib
import json
import os
import subprocess
import time

import numpy as np
import pytest

import helpers as H
from test_oracle_golden import check_e2e_against_reference

FAKE = os.path.join(H.ROOT, "tests", "fake_sfmx")
synth = importlib.import_module(H.PKG_NAME + ".synth")
FILES = ("keyframes_camera_centers.csv", "posegraph_edges.csv", "templeRing_sparse_points.ply")


@pytest.fixture(scope="module")
def builds():
    procs = {san: subprocess.Popen(["make", "-C", FAKE, f"SAN={san}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for san in ("thread", "address,undefined")}
    out = {}
    for san, p in procs.items():
        log, _ = p.communicate(timeout=900)
        assert p.returncode == 0, log[-3000:]
        out[san] = os.path.join(FAKE, "_build", san.replace(",", "_"), "templering_sfm")
    return out


def _dataset(tmp_path, name):
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    root = str(tmp_path / name)
    names = [str(s) for s in g["names"]]
    synth.write_dataset(root, dict(images=g["images"], K=g["K"], R=g["R"], t=g["t"], names=names, lat=g["lat"], lon=g["lon"]))
    with open(os.path.join(root, "cfg.json"), "w") as f:
        f.write(str(g["config"]))
    return g, root


def _env(**extra):
    return {**os.environ, "TSAN_OPTIONS": "halt_on_error=0 exitcode=66", "ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1",
            "SFMX_HOST_THREADS": "4", **extra}


def _clean(stderr):
    for needle in ("ThreadSanitizer", "AddressSanitizer", "LeakSanitizer", "runtime error"):
        assert needle not in stderr, stderr[:4000]


def _run_ranks(exe, root, world, tag, **extra):
    idf = os.path.join(root, f"ids_{tag}.bin")
    procs = []
    run_id = f"{tag}-{time.time()}"  # one id per launch: a rank ignores id files written for another one
    for r in range(world):
        env = _env(SFMX_DIST_WORLD=str(world), SFMX_DIST_RANK=str(r), SFMX_DIST_ID_FILE=idf, SFMX_DIST_RUN_ID=run_id, **extra)
        procs.append(subprocess.Popen([exe, root, os.path.join(root, f"out_{tag}_r{r}"), "--config", os.path.join(root, "cfg.json")],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root, env=env))
    res = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:  # a collective that never completes: the deadlock this design excludes
            for q in procs:
                q.kill()
            pytest.fail(f"{world}-rank run did not finish")
        assert p.returncode == 0, e[:4000]
        _clean(e)
        res.append(o)
    return res, os.path.join(root, f"out_{tag}_r0")


def test_host_runtime_under_tsan_all_lanes(builds, tmp_path):
    g, root = _dataset(tmp_path, "e2e_loop")
    out = os.path.join(root, "out")
    p = subprocess.run([builds["thread"], root, out, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True, cwd=root, env=_env())
    assert p.returncode == 0, p.stderr[:4000]
    _clean(p.stderr)
    check_e2e_against_reference(g, p.stdout, out)
    # the serial schedule gives the same bytes (what the GPU suite checks on the device, here for the instrumented build)
    out2 = os.path.join(root, "out_serial")
    q = subprocess.run([builds["thread"], root, out2, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True, cwd=root,
                       env=_env(SFMX_NO_ASYNC="1", SFMX_NO_PREFETCH="1"))
    assert q.returncode == 0 and p.stdout.replace(out, "X") == q.stdout.replace(out2, "X")
    for fn in FILES:
        assert open(os.path.join(out, fn)).read() == open(os.path.join(out2, fn)).read(), fn


def test_host_runtime_under_asan_ubsan(builds, tmp_path):
    for name in ("e2e_small", "e2e_loop"):
        g, root = _dataset(tmp_path, name)
        out = os.path.join(root, "out")
        p = subprocess.run([builds["address,undefined"], root, out, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True,
                           cwd=root, env=_env())
        assert p.returncode == 0, p.stderr[:4000]
        _clean(p.stderr)
        check_e2e_against_reference(g, p.stdout, out)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_with_real_processes_under_tsan(builds, tmp_path, world):
    g, root = _dataset(tmp_path, "e2e_loop")
    one = os.path.join(root, "out_one")
    p = subprocess.run([builds["thread"], root, one, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True, cwd=root, env=_env())
    assert p.returncode == 0
    outs, r0 = _run_ranks(builds["thread"], root, world, f"w{world}")
    assert outs[0].replace(r0, "X") == p.stdout.replace(one, "X")   # rank 0 reports; its bytes are the one-rank run's
    assert all(o == "" for o in outs[1:])
    for fn in FILES:
        assert open(os.path.join(r0, fn)).read() == open(os.path.join(one, fn)).read(), fn
    # tolerance mode: point-sharded BA regroups the sums (centres move), everything RANSAC decides stays exact
    outs, rp = _run_ranks(builds["thread"], root, world, f"p{world}", SFMX_BA_SHARD="points")
    assert outs[0].replace(rp, "X") == p.stdout.replace(one, "X")
    assert open(os.path.join(rp, "posegraph_edges.csv")).read() == open(os.path.join(one, "posegraph_edges.csv")).read()
