#!/usr/bin/env python3
"""N-rank sharded BA on ONE GPU (SFMX_VIRTUAL_WORLD, csrc/hip/ba.hip): how far do the keyframe centres of a whole run move when
S | b is formed the way N ranks form it?  shard = elements (sfmx_ba_step_sharded_elements: disjoint element slices, zeros
elsewhere, summed -- exact by construction) or points (sfmx_ba_step_sharded: N shard sums combined in all-reduce order rank /
reverse / ring / tree -- the addends of every element regrouped; relay = each shard continues the sums of the one before).  Workloads: the 47-frame bench sequence and the e2e_loop fixture.  Prints one JSON line per
(workload, N, order): max |dC| / max |C|, the ATE-RMSE (ate_keyframes, Sim(3)) of both runs and their difference."""
import importlib, json, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
pipe = importlib.import_module(I.PKG + ".pipeline")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, I.PKG, "_build", "ate_keyframes")


def ate(td, seq, out, n_kf):
    I.synth.write_par_ang(td, seq)
    r = subprocess.run([TOOL, "--par", os.path.join(td, "templeRing", "templeR_par.txt"), "--keyframes",
                        os.path.join(out, "keyframes_camera_centers.csv"), "--count", str(n_kf)], capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.strip().startswith("ATE_RMSE:"):
            return float(line.split(":")[1])
    return None


def workloads():
    seq = I.synth.make_sequence(47, 640, 480, 0.3, n_blobs=20000, seed=7)
    yield "bench47", seq, dict(pipe.DEFAULTS, frames=47)
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_loop.npz"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    seq = dict(images=g["images"], K=g["K"], R=g["R"], t=g["t"], names=[str(s) for s in g["names"]], lat=g["lat"], lon=g["lon"])
    yield "e2e_loop", seq, H.pipe_cfg_from_json(json.loads(str(g["config"])))


ctx = I.capi.Context(0)
for name, seq, cfg in workloads():
    with tempfile.TemporaryDirectory() as td:
        def run(tag, env):
            for k in ("SFMX_VIRTUAL_WORLD", "SFMX_VIRTUAL_WORLD_ORDER", "SFMX_BA_SHARD"):
                os.environ.pop(k, None)
            os.environ.update(env)
            out = os.path.join(td, tag)
            r = pipe.run(ctx, seq["images"], seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out)
            return r, out
        base, base_out = run("one", {})
        c0 = base["centres"]
        ate0 = ate(td, seq, base_out, len(c0))
        for n in (2, 8):
            for shard, order in (("elements", "rank"), ("elements", "ring"), ("points", "rank"), ("points", "reverse"), ("points", "ring"),
                                 ("points", "tree"), ("points", "relay")):
                r, out = run(f"w{n}{shard}{order}", {"SFMX_VIRTUAL_WORLD": str(n), "SFMX_VIRTUAL_WORLD_ORDER": order, "SFMX_BA_SHARD": shard})
                c = r["centres"]
                same_kf = c.shape == c0.shape
                d = float(np.nanmax(np.abs(c - c0))) if same_kf else float("nan")
                a1 = ate(td, seq, out, len(c))
                print(json.dumps({"workload": name, "virtual_world": n, "shard": shard, "order": order, "keyframes": int(len(c)), "same_keyframes": bool(same_kf),
                                  "same_log": r["log"].replace(out, "X") == base["log"].replace(base_out, "X"),
                                  "bit_identical": bool(same_kf and np.array_equal(c.view(np.uint64), c0.view(np.uint64))),
                                  "max_abs_centre": float(np.nanmax(np.abs(c0))), "max_abs_dcentre": d,
                                  "rel": d / float(np.nanmax(np.abs(c0))) if same_kf else None,
                                  "ate_one_rank": ate0, "ate_virtual": a1, "ate_delta": None if (a1 is None or ate0 is None) else abs(a1 - ate0)}), flush=True)
ctx.close()
