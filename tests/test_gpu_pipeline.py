"""GPU suite: the whole per-frame loop (C++ host pipeline on the sfmx C ABI) and the drop-in CLI.

Two comparators:
 * the golden output of the REAL reference CLI for everything the reference computes with defined
   behaviour (stdout, posegraph_edges.csv, keyframe selection) -- see check_e2e_against_reference;
 * the oracle's pipeline (same Q12 reading) for every byte of every output file, including the map
   points and BA-refined camera centres."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from test_oracle_golden import check_e2e_against_reference

pytestmark = pytest.mark.gpu
capi = importlib.import_module(H.PKG_NAME + ".capi")
pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
synth = importlib.import_module(H.PKG_NAME + ".synth")
FILES = ("keyframes_camera_centers.csv", "posegraph_edges.csv", "templeRing_sparse_points.ply")


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _same_files(a, b):
    for fn in FILES:
        ta, tb = open(os.path.join(a, fn)).read(), open(os.path.join(b, fn)).read()
        assert ta == tb, f"{fn} differs:\n{ta[:400]}\n---\n{tb[:400]}"


@pytest.mark.parametrize("name", ["e2e_small", "e2e_keyframes", "e2e_loop"])
def test_pipeline_matches_reference_and_oracle(ctx, name, tmp_path):
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    cfg = H.pipe_cfg_from_json(json.loads(str(g["config"])))
    names = [str(s) for s in g["names"]]
    out_g, out_o = str(tmp_path / "gpu"), str(tmp_path / "orc")
    r = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, out_g)
    check_e2e_against_reference(g, r["log"], out_g)
    rc, olog, nk, npnt = H.orc_pipeline_run(g["images"], names, g["K"], g["lat"], g["lon"], cfg, out_o)
    assert rc == 0
    assert r["log"].replace(out_g, "X") == olog.replace(out_o, "X")
    _same_files(out_g, out_o)
    assert r["stats"]["n_keyframes"] == nk and r["stats"]["n_points"] == npnt


def test_pipeline_640x480_vs_oracle(ctx, tmp_path):
    """Full-size frames, default reference config, BA + keyframes firing; every output byte vs the oracle."""
    seq = synth.make_sequence(6, 640, 480, 0.3, n_blobs=20000, seed=7)
    cfg = dict(H.PIPE_DEFAULTS, frames=6)
    out_g, out_o = str(tmp_path / "gpu"), str(tmp_path / "orc")
    r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_g)
    rc, olog, nk, npnt = H.orc_pipeline_run(seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_o)
    assert rc == 0 and nk >= 3 and npnt > 100
    assert r["log"].replace(out_g, "X") == olog.replace(out_o, "X")
    _same_files(out_g, out_o)


def test_bench_workload_47_frames_vs_oracle(ctx, tmp_path):
    """BASELINE config 2 == the workload bench.py times: 47 frames 640x480, reference default config (synthetic
    TempleRing-47 stand-in, same generator call and seed as bench.py rank 0).  stdout and all three output files must be
    byte-equal to the oracle's run: the BA window slides past 6 keyframes, lane joins happen at full image size."""
    seq = synth.make_sequence(47, 640, 480, 0.3, n_blobs=20000, seed=7)
    cfg = dict(H.PIPE_DEFAULTS, frames=47)
    out_g, out_o = str(tmp_path / "gpu"), str(tmp_path / "orc")
    r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_g)
    rc, olog, nk, npnt = H.orc_pipeline_run(seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_o)
    assert rc == 0 and nk >= 30 and npnt > 1000, (nk, npnt)
    assert r["stats"]["ransac_cert_misses"] == 0
    assert r["log"].replace(out_g, "X") == olog.replace(out_o, "X")
    _same_files(out_g, out_o)
    assert r["stats"]["n_keyframes"] == nk and r["stats"]["n_points"] == npnt


def test_two_keyframe_pair_ate_two_frames(ctx, tmp_path):
    """BASELINE config 1: a 2-keyframe 640x480 pair (KLT + RANSAC + triangulate) checked with ate_two_frames
    (cpp/tools/ate_two_frames.cpp; the build's tool is pinned to the reference tool digit for digit in
    tests/test_tools.py): the tool's text on the GPU output equals its text on the oracle's output."""
    seq = synth.make_sequence(2, 640, 480, 2.0, n_blobs=20000, seed=7)
    cfg = dict(H.PIPE_DEFAULTS, frames=2)
    outs = {}
    for tag in ("gpu", "orc"):
        out = str(tmp_path / tag)
        if tag == "gpu":
            r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out)
            assert r["stats"]["n_keyframes"] == 2
        else:
            rc, _, nk, _ = H.orc_pipeline_run(seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out)
            assert rc == 0 and nk == 2
        outs[tag] = out
    _same_files(outs["gpu"], outs["orc"])
    synth.write_par_ang(str(tmp_path), seq)
    tool = os.path.join(H.ROOT, H.PKG_NAME, "_build", "ate_two_frames")
    texts = []
    for tag in ("gpu", "orc"):
        for extra in ([], ["--se3"]):
            p = subprocess.run([tool, "--par", str(tmp_path / "templeRing" / "templeR_par.txt"), "--keyframes",
                                os.path.join(outs[tag], "keyframes_camera_centers.csv")] + extra, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            texts.append(p.stdout)
    assert texts[0] == texts[2] and texts[1] == texts[3]
    assert "ATE_RMSE:" in texts[0] and "templeR0001.png  ->  [1] templeR0002.png" in texts[0]


def test_pipeline_c3_5000_tracks_vs_oracle(ctx, tmp_path):
    """BASELINE config C3 on a prefix the oracle finishes in seconds: 640x480, frame-filling texture,
    klt.max_tracks=5000 / min_tracks=2045 / min_distance=4 -- every output byte vs the oracle, and >= 4500 live tracks
    actually reach the KLT kernel."""
    # 0.01 deg/frame: the reference's LK adds ~iters x the true flow per level (it samples I0 and I1 at the same moved
    # coordinates, lk_step T:424-460), so on this close, frame-filling shell only sub-pixel flow keeps the tracks alive
    seq = synth.make_sequence(4, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
    cfg = dict(H.PIPE_DEFAULTS, frames=4, max_tracks=5000, min_tracks=2045, min_distance=4, kf_parallax_px=1.0)
    out_g, out_o = str(tmp_path / "gpu"), str(tmp_path / "orc")
    r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_g)
    st = r["stats"]
    assert st["tracks_in"] / max(1, st["klt_calls"]) >= 4500, st
    rc, olog, nk, npnt = H.orc_pipeline_run(seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_o)
    assert rc == 0 and nk >= 3 and npnt >= 4000
    assert r["log"].replace(out_g, "X") == olog.replace(out_o, "X")
    _same_files(out_g, out_o)


def test_c5_end_to_end_1080p_loop_closure_posegraph(ctx, tmp_path, monkeypatch):
    """BASELINE config C5 in ONE run of the per-frame loop (T:1708-1871): 1920x1080 frames on an out-and-back ring path, so
    that the descriptor search finds old keyframes again, the LK + RANSAC verification accepts them (T:1822-1858) and every
    accepted loop closure runs the pose graph and the second BA (T:1859-1863).  With the pose graph on the reference's own
    dense system (SFMX_POSEGRAPH_SOLVER=dense) stdout and all three files are byte-equal to the oracle's; with the
    structured FP64-MFMA solver (the product path above 6 400 unknowns, forced here) the same keyframes / edges come out
    and the centres agree to 1e-9 of their magnitude."""
    ang = [0.1 * a for a in (0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0, 1)]
    seq = synth.make_sequence(len(ang), 1920, 1080, 0.1, n_blobs=20000, seed=13, angles=ang)
    cfg = dict(H.PIPE_DEFAULTS, frames=len(ang), kf_min_inliers=100, kf_parallax_px=1.0)
    out_o = str(tmp_path / "orc")
    rc, olog, nk, npnt = H.orc_pipeline_run(seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_o)
    edges_o = open(os.path.join(out_o, "posegraph_edges.csv")).read().splitlines()
    n_loops = sum(1 for l in edges_o[1:] if l.endswith(",1"))
    assert rc == 0 and nk == len(ang) and n_loops >= 3, (rc, nk, n_loops)
    monkeypatch.setenv("SFMX_POSEGRAPH_SOLVER", "dense")
    out_d = str(tmp_path / "dense")
    rd = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_d)
    assert rd["log"].replace(out_d, "X") == olog.replace(out_o, "X")
    _same_files(out_d, out_o)
    assert rd["stats"]["n_keyframes"] == nk and rd["stats"]["n_points"] == npnt and rd["stats"]["ransac_cert_misses"] == 0
    monkeypatch.setenv("SFMX_POSEGRAPH_SOLVER", "structured")
    out_s = str(tmp_path / "structured")
    rs = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out_s)
    assert rs["log"].replace(out_s, "X") == olog.replace(out_o, "X")
    assert open(os.path.join(out_s, "posegraph_edges.csv")).read() == open(os.path.join(out_o, "posegraph_edges.csv")).read()
    cd, cs = rd["centres"], rs["centres"]
    assert cd.shape == cs.shape == (nk, 3)
    assert np.allclose(cs, cd, rtol=0, atol=1e-9 * np.abs(cd).max()), np.abs(cs - cd).max() / np.abs(cd).max()


def test_device_resident_frames_give_identical_results(ctx, tmp_path):
    import torch
    g = np.load(os.path.join(H.GOLDEN, "e2e_keyframes.npz"))
    cfg = H.pipe_cfg_from_json(json.loads(str(g["config"])))
    names = [str(s) for s in g["names"]]
    a, b = str(tmp_path / "host"), str(tmp_path / "dev")
    r1 = pipe.run(ctx, g["images"], names, g["K"], g["lat"], g["lon"], cfg, a)
    dev = torch.from_numpy(np.ascontiguousarray(g["images"])).to("cuda:0")
    torch.cuda.synchronize()
    r2 = pipe.run(ctx, None, names, g["K"], g["lat"], g["lon"], cfg, b, images_dev=dev.data_ptr(), shape=tuple(dev.shape))
    assert r1["log"].replace(a, "X") == r2["log"].replace(b, "X")
    _same_files(a, b)


def test_cli_drop_in(tmp_path):
    """The templering_sfm binary on a dataset directory: same stdout / files as the oracle, same exit codes."""
    g = np.load(os.path.join(H.GOLDEN, "e2e_small.npz"))
    cfgj = json.loads(str(g["config"]))
    names = [str(s) for s in g["names"]]
    root = str(tmp_path / "data")
    seq = dict(images=g["images"], K=g["K"], R=g["R"], t=g["t"], names=names, lat=g["lat"], lon=g["lon"])
    synth.write_dataset(root, seq)
    with open(os.path.join(root, "cfg.json"), "w") as f:
        json.dump(cfgj, f)
    out = os.path.join(root, "out")
    p = subprocess.run([pipe.CLI_PATH, root, out, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True, cwd=root)
    assert p.returncode == 0, p.stderr
    check_e2e_against_reference(g, p.stdout.replace(out, "<OUT>").replace("<OUT>", out), out)
    out_o = str(tmp_path / "orc")
    rc, olog, _, _ = H.orc_pipeline_run(g["images"], names, g["K"], g["lat"], g["lon"], H.pipe_cfg_from_json(cfgj), out_o)
    assert p.stdout.replace(out, "X") == olog.replace(out_o, "X")
    _same_files(out, out_o)
    # exit codes / messages of the reference CLI (T:1520-1536, 1605-1611, 1913-1916)
    assert subprocess.run([pipe.CLI_PATH], capture_output=True).returncode == 2
    q = subprocess.run([pipe.CLI_PATH, root, out, "--help"], capture_output=True, text=True)
    assert q.returncode == 0 and "Run without args to see usage." in q.stderr
    q = subprocess.run([pipe.CLI_PATH, root, out, "--bogus"], capture_output=True, text=True)
    assert q.returncode == 1 and q.stderr.strip() == "ERROR: Unknown option: --bogus"
    q = subprocess.run([pipe.CLI_PATH, str(tmp_path / "nope"), out], capture_output=True, text=True, cwd=str(tmp_path))
    assert q.returncode == 1 and q.stderr.startswith("ERROR: Failed to open: ")
    q = subprocess.run([pipe.CLI_PATH, root, out, "3", "--export-geometry", "none"], capture_output=True, text=True, cwd=str(tmp_path))
    assert q.returncode == 0 and q.stdout.count("frame ") == 3


def test_cli_mesh_export(tmp_path):
    """--export-geometry both on a 640x480 run: the mesh PLY is well formed, its vertices are map points (the PLY writers
    print 6 significant digits), its faces index them, and the file does not depend on the lane schedule.  The mesh
    builder itself is pinned to the reference function bit for bit in tests/test_host_math.py."""
    seq = synth.make_sequence(5, 640, 480, 0.3, n_blobs=20000, seed=7)
    root = str(tmp_path / "data")
    synth.write_dataset(root, seq)
    outs = []
    for k, extra in enumerate(({}, {"SFMX_NO_ASYNC": "1", "SFMX_NO_PREFETCH": "1"})):
        out = os.path.join(root, f"out{k}")
        p = subprocess.run([pipe.CLI_PATH, root, out, "5", "--export-geometry", "both", "--mesh-kf", "2", "--mesh-max-points", "400"],
                           capture_output=True, text=True, cwd=str(tmp_path), env={**os.environ, **extra})
        assert p.returncode == 0 and "WARN" not in p.stderr, p.stderr
        outs.append(out)
    mesh = open(os.path.join(outs[0], "templeRing_mesh_sparse_kf2.ply")).read()
    assert mesh == open(os.path.join(outs[1], "templeRing_mesh_sparse_kf2.ply")).read()
    lines = mesh.splitlines()
    nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
    nf = int([l for l in lines if l.startswith("element face")][0].split()[-1])
    body = lines[lines.index("end_header") + 1:]
    assert 50 <= nv <= 400 and nf > nv and len(body) == nv + nf
    cloud = open(os.path.join(outs[0], "templeRing_sparse_points.ply")).read().splitlines()
    cloud_pts = set(cloud[cloud.index("end_header") + 1:])
    assert all(v in cloud_pts for v in body[:nv])
    faces = np.array([[int(x) for x in l.split()] for l in body[nv:]])
    assert np.all(faces[:, 0] == 3) and faces[:, 1:].min() >= 0 and faces[:, 1:].max() < nv
    # too few points in view: the reference's warning, no file, exit 0
    out = os.path.join(root, "out_few")
    p = subprocess.run([pipe.CLI_PATH, root, out, "1", "--export-geometry", "mesh"], capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0 and p.stderr.strip().endswith("WARN: mesh export skipped (insufficient projected points or no valid triangles).")
    assert not os.path.exists(os.path.join(out, "templeRing_mesh_sparse_kf0.ply")) and not os.path.exists(os.path.join(out, "templeRing_sparse_points.ply"))


def test_long_sequence_lanes_equal_serial(ctx, tmp_path, monkeypatch):
    """120 frames (every one a keyframe, loop closures being verified all along): the five-lane schedule, run twice, and
    the fully serial one must write the same bytes -- the short CLI sequences above hardly fill the tracker lane's ring."""
    seq = synth.make_sequence(120, 320, 240, 0.3, n_blobs=12000, seed=5)
    cfg = dict(H.PIPE_DEFAULTS, frames=120, max_tracks=900, min_tracks=400, kf_min_inliers=100000)  # inliers < kf_min_inliers: keyframe (T:1700-1704)
    outs = []
    for k, serial in enumerate((False, True, False)):
        if serial:
            monkeypatch.setenv("SFMX_NO_ASYNC", "1")
            monkeypatch.setenv("SFMX_NO_PREFETCH", "1")
        else:
            monkeypatch.delenv("SFMX_NO_ASYNC", raising=False)
            monkeypatch.delenv("SFMX_NO_PREFETCH", raising=False)
        out = str(tmp_path / f"o{k}")
        r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out)
        outs.append((r["log"].replace(out, "X"), out, r["stats"]))
    assert outs[0][2]["n_keyframes"] >= 100 and outs[0][2]["n_points"] > 1000
    for k in (1, 2):
        assert outs[0][0] == outs[k][0]
        _same_files(outs[0][1], outs[k][1])


def test_shi_fast_path_equals_full_sort(tmp_path):
    """The prefix-select fast path of the corner pick must equal the full std::sort path (run in a
    child process with SFMX_SHI_FULL_SORT=1) byte for byte, on a noisy and on a noise-free (tie-prone) scene."""
    import sys
    for noise in (True, False):
        seq = synth.make_sequence(4, 320, 240, 0.3, n_blobs=6000, seed=11, noise=noise)
        root = str(tmp_path / f"d{int(noise)}")
        synth.write_dataset(root, seq)
        outs = []
        for env_extra in ({}, {"SFMX_SHI_FULL_SORT": "1"}):
            out = os.path.join(root, "out" + str(len(outs)))
            p = subprocess.run([pipe.CLI_PATH, root, out, "4"], capture_output=True, text=True, cwd=str(tmp_path),
                               env={**os.environ, **env_extra})
            assert p.returncode == 0, p.stderr
            outs.append((p.stdout.replace(out, "X"), out))
        assert outs[0][0] == outs[1][0]
        _same_files(outs[0][1], outs[1][1])


def test_async_lanes_do_not_change_results(tmp_path):
    """The tracker lane, the corner prefetch worker and lanes B / C (async BA, kf->kf RANSAC, loop-closure verification)
    only reorder WHEN work happens: outputs must be byte-identical to the fully serial schedule, whichever subset is on."""
    g = np.load(os.path.join(H.GOLDEN, "e2e_loop.npz"))
    cfgj = json.loads(str(g["config"]))
    names = [str(s) for s in g["names"]]
    root = str(tmp_path / "data")
    synth.write_dataset(root, dict(images=g["images"], K=g["K"], R=g["R"], t=g["t"], names=names, lat=g["lat"], lon=g["lon"]))
    with open(os.path.join(root, "cfg.json"), "w") as f:
        json.dump(cfgj, f)
    outs = []
    for extra in ({}, {"SFMX_NO_PREFETCH": "1", "SFMX_NO_ASYNC": "1", "SFMX_NO_GRAPH": "1"}, {"SFMX_NO_ASYNC": "1"}, {"SFMX_NO_PREFETCH": "1"},
                  {"SFMX_NO_TRACK_LANE": "1"}, {"SFMX_NO_TRACK_LANE": "1", "SFMX_NO_PREFETCH": "1", "SFMX_NO_CTX_POOL": "1"},
                  {"SFMX_NO_EDGE_LANE": "1"}, {"SFMX_NO_RANSAC_LANE": "1", "SFMX_PREFETCH_WORKERS": "1"},
                  {"SFMX_BA_NO_FUSE": "1", "SFMX_BA_NO_WAVE_PRIO": "1"}, {"SFMX_BA_EXPAND": "split", "SFMX_BA_NO_POLL": "1"},
                  {"SFMX_BA_SOLVE": "device"}, {"SFMX_BA_RESIDENT": "1"}, {"SFMX_SPIN_US": "200"}, {"SFMX_NO_PRELOAD": "1"}, {"SFMX_RANSAC_LANES": "2"}, {"SFMX_KLT_K": "0"}, {"SFMX_KLT_K": "2"}, {"SFMX_KLT_K": "4"}, {"SFMX_RANSAC_HYP": "legacy"}, {"SFMX_JOIN_C_EARLY": "1"}, {"SFMX_VERIFY_LATE": "1"}, {"SFMX_VERIFY_LATE": "1", "SFMX_JOIN_C_EARLY": "1", "SFMX_RANSAC_LANES": "1"}, {"SFMX_BA_POINTS": "global", "SFMX_BA_PUBLISH": "last", "SFMX_BA_TILE": "64"}, {"SFMX_BA_TILE": "16", "SFMX_BA_PTS": "1"}, {"SFMX_BA_PTS": "4"}, {"SFMX_KLT_PIPE": "1"}, {"SFMX_KLT_PIPE": "0"}, {"SFMX_KLT_SUMS": "valu"}, {"SFMX_KLT_SUMS": "valu", "SFMX_KLT_PIPE": "1"},
                  {"SFMX_SHI_MODE": "tile"}, {"SFMX_SHI_MODE": "tile,2"}, {"SFMX_SHI_MODE": "sweeps"}, {"SFMX_SHI_SWEEPS": "5,8,40", "SFMX_SHI_INNER": "3"}, {"SFMX_SHI_SWEEPS": "2,1,0"}):
        out = os.path.join(root, f"out{len(outs)}")
        p = subprocess.run([pipe.CLI_PATH, root, out, "--config", os.path.join(root, "cfg.json")], capture_output=True, text=True,
                           cwd=root, env={**os.environ, **extra})
        assert p.returncode == 0, p.stderr
        outs.append((p.stdout.replace(out, "X"), out))
    for k in range(1, len(outs)):
        assert outs[0][0] == outs[k][0]
        _same_files(outs[0][1], outs[k][1])
    check_e2e_against_reference(g, outs[0][0].replace("X", outs[0][1]), outs[0][1])
