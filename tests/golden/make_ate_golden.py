#!/usr/bin/env python3
"""Fixtures for the ATE evaluator (tests/test_tools.py): inputs (a par file, keyframe CSVs) and the stdout / stderr /
exit code the REAL reference tool gives on them.  Needs oracle/_ref/ate_keyframes_ref (make -C oracle ref), i.e. it
only runs where /root/reference exists; the resulting tests/golden/ate_keyframes.json is committed."""
import importlib, json, os, subprocess, sys, tempfile
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
synth = importlib.import_module(H.PKG_NAME + ".synth")
REF = os.path.join(ROOT, "oracle", "_ref", "ate_keyframes_ref")
REF2 = os.path.join(ROOT, "oracle", "_ref", "ate_two_frames_ref")
REF3 = os.path.join(ROOT, "oracle", "_ref", "gt_keyframe_edge_ref")


def par_text(seq):
    lines = [str(len(seq["names"]))]
    for i, n in enumerate(seq["names"]):
        vals = list(seq["K"].ravel()) + list(seq["R"][i].ravel()) + list(seq["t"][i].ravel())
        lines.append(n + " " + " ".join(repr(float(v)) for v in vals))
    return "\n".join(lines) + "\n"


def main():
    rng = np.random.default_rng(11)
    seq = synth.make_sequence(16, 64, 48, 7.0, n_blobs=10, seed=3, noise=False)  # only the poses matter
    C = np.array([-seq["R"][i].T @ seq["t"][i] for i in range(16)])
    # an "estimated" trajectory: ground truth under an unknown similarity + noise (monocular gauge), 6 significant digits
    th = 0.7
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    est = 3.7 * (C @ Rz.T) + np.array([0.4, -1.2, 2.5]) + rng.normal(size=C.shape) * 0.02
    def csv(points, names, header="kf_id,frame_idx,image,x,y,z,lat,lon", extra_rows=()):
        rows = [header]
        for k, (p, n) in enumerate(zip(points, names)):
            rows.append(f"{k},{k},{n},{p[0]:.6g},{p[1]:.6g},{p[2]:.6g},0,{7.0 * k:.6g}")
        rows.extend(extra_rows)
        return "\n".join(rows) + "\n"
    files = {
        "par.txt": par_text(seq),
        "kf.csv": csv(est, seq["names"]),
        "kf_exact.csv": csv(2.0 * C, seq["names"]),                                    # zero residual up to rounding
        "kf_mirror.csv": csv(est * np.array([1, 1, -1.0]), seq["names"]),               # reflection branch (D(2,2) = -1)
        "kf_line.csv": csv(np.stack([np.linspace(0, 1, 16), np.zeros(16), np.zeros(16)], 1), seq["names"]),  # rank-1 covariance
        "kf_quoted.csv": csv(est, ['"' + n + '"' for n in seq["names"]], extra_rows=["bad,row", "9,9,x.png,a,b,c,0,0", ""]),
        "kf_nocol.csv": csv(est, seq["names"], header="kf_id,frame_idx,image,px,y,z,lat,lon"),
        "kf_unknown.csv": csv(est, ["nope.png"] + list(seq["names"][1:])),
    }
    cases = [
        ["--par", "par.txt", "--keyframes", "kf.csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--count", "16"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--start", "3", "--count", "9", "--se3"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--count", "16", "--se3", "--sim3"],
        ["--par", "par.txt", "--keyframes", "kf_exact.csv", "--count", "12"],
        ["--par", "par.txt", "--keyframes", "kf_mirror.csv", "--count", "16"],
        ["--par", "par.txt", "--keyframes", "kf_line.csv", "--count", "8"],
        ["--par", "par.txt", "--keyframes", "kf_quoted.csv", "--count", "16"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--count", "2"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--count", "abc", "--start", "1x"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--start", "10", "--count", "9"],   # range error
        ["--par", "par.txt", "--keyframes", "kf_nocol.csv"],                              # missing column
        ["--par", "par.txt", "--keyframes", "kf_unknown.csv"],                            # name not in par
        ["--par", "missing.txt", "--keyframes", "kf.csv"],                                # unreadable par
        ["--par", "par.txt", "--keyframes", "kf.csv", "--count", "1"],                    # usage
        ["--keyframes", "kf.csv"],                                                        # usage
    ]
    out = {"files": files, "cases": []}
    with tempfile.TemporaryDirectory() as d:
        for n, txt in files.items():
            open(os.path.join(d, n), "w").write(txt)
        for args in cases:
            r = subprocess.run([REF] + args, cwd=d, capture_output=True, text=True)
            out["cases"].append({"args": args, "rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr})
            print(args, "->", r.returncode, r.stdout.count("\n"), "lines")
    json.dump(out, open(os.path.join(HERE, "ate_keyframes.json"), "w"), indent=1)

    # ---- ate_two_frames: closed-form two-point alignment
    gt0, gt1 = C[0], C[1]
    d = gt1 - gt0
    files2 = {
        "par.txt": files["par.txt"],
        "par_short.txt": files["par.txt"] + "broken.png 1 2 3\n",                       # a short record spoils the file
        "kf.csv": files["kf.csv"],
        "kf_same_dir.csv": csv([gt0 * 1.0, gt0 + 2.5 * d] + list(est[2:]), seq["names"]),   # parallel baselines: identity rotation
        "kf_opposite.csv": csv([gt0 * 1.0, gt0 - 0.7 * d] + list(est[2:]), seq["names"]),   # anti-parallel: half turn
        "kf_zero.csv": csv([est[0], est[0]] + list(est[2:]), seq["names"]),                # zero estimated baseline
        "kf_ragged.csv": csv(est, seq["names"], extra_rows=["1,2", "7,7,templeR0003.png,1,2,3", "8,8,templeR0004.png,1,2,3,0,0,extra"]),
        "kf_unknown.csv": files["kf_unknown.csv"],
        "kf_nocol.csv": files["kf_nocol.csv"],
    }
    cases2 = [
        ["--par", "par.txt", "--keyframes", "kf.csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "3", "--j", "11"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "9", "--j", "2", "--se3"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--se3", "--sim3", "--i", "x", "--j", "5"],
        ["--par", "par.txt", "--keyframes", "kf_same_dir.csv"],
        ["--par", "par.txt", "--keyframes", "kf_opposite.csv"],
        ["--par", "par.txt", "--keyframes", "kf_zero.csv"],
        ["--par", "par.txt", "--keyframes", "kf_zero.csv", "--se3"],
        ["--par", "par.txt", "--keyframes", "kf_ragged.csv", "--i", "16", "--j", "17"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "2", "--j", "2"],            # invalid indices
        ["--par", "par.txt", "--keyframes", "kf.csv", "--j", "40"],                        # out of range
        ["--par", "par.txt", "--keyframes", "kf_unknown.csv"],                             # name missing in par
        ["--par", "par_short.txt", "--keyframes", "kf.csv"],                               # unreadable par
        ["--par", "par.txt", "--keyframes", "kf_nocol.csv"],                               # unreadable CSV
        ["--par", "par.txt", "--keyframes"],                                               # flag without value -> usage
        ["--keyframes", "kf.csv", "--i", "0"],                                             # usage
    ]
    out2 = {"files": files2, "cases": []}
    with tempfile.TemporaryDirectory() as d2:
        for n, txt in files2.items():
            open(os.path.join(d2, n), "w").write(txt)
        for args in cases2:
            r = subprocess.run([REF2] + args, cwd=d2, capture_output=True, text=True)
            out2["cases"].append({"args": args, "rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr})
            print("two_frames", args, "->", r.returncode, r.stdout.count("\n"), "lines")
    json.dump(out2, open(os.path.join(HERE, "ate_two_frames.json"), "w"), indent=1)

    # ---- gt_keyframe_edge: ground-truth relative pose and the error of an estimated edge
    def rodrigues_of(i, j, noise):
        Rij = seq["R"][j] @ seq["R"][i].T
        th = np.arccos(np.clip((np.trace(Rij) - 1) / 2, -1, 1))
        w = th / (2 * np.sin(th)) * np.array([Rij[2, 1] - Rij[1, 2], Rij[0, 2] - Rij[2, 0], Rij[1, 0] - Rij[0, 1]])
        tij = seq["t"][j] - Rij @ seq["t"][i]
        return w + noise * rng.normal(size=3), tij / np.linalg.norm(tij) * 2.5 + noise * rng.normal(size=3)
    erows = ["i,j,kind,rvec_x,rvec_y,rvec_z,t_x,t_y,t_z"]
    for (i, j, kind) in ((0, 1, "seq"), (1, 2, "seq"), (3, 9, "loop"), (5, 4, "seq")):
        w, tt = rodrigues_of(i, j, 0.01)
        erows.append(f"{i},{j},{kind}," + ",".join(f"{v:.6g}" for v in list(w) + list(tt)))
    erows += ["7,8,seq,0,0,0,0,0,0", "bad,row", ' 2 , 6 , "quoted" , 0.1,0.2,0.3, -1,0,0 ']
    pipeline_edges = "i,j,rvec_x,rvec_y,rvec_z,t_x,t_y,t_z,inliers,is_loop\n0,1,0.01,0.02,0.03,0.1,0.2,0.3,500,0\n"
    shuffled = ["kf_id,frame_idx,image,x,y,z,lat,lon"] + [f"{k},{k},{seq['names'][k]},0,0,0,0,0" for k in (2, 0, 3, 1, 9)]
    files3 = {
        "par.txt": files["par.txt"],
        "kf.csv": files["kf.csv"],
        "kf_shuffled.csv": "\n".join(shuffled) + "\n",         # ids not 0..n-1 in order: remapped, id 9 dropped, id 4 a hole
        "kf_unknown.csv": files["kf_unknown.csv"],
        "kf_noid.csv": files["kf.csv"].replace("kf_id", "id", 1),
        "edges.csv": "\n".join(erows) + "\n",
        "edges_pipeline.csv": pipeline_edges,                  # the pipeline's own schema has no `kind` column
    }
    cases3 = [
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "1"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "3", "--j", "9", "--emit-csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "1", "--edges", "edges.csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "3", "--j", "9", "--edges", "edges.csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "5", "--j", "4", "--edges", "edges.csv"],
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "7", "--j", "8", "--edges", "edges.csv"],     # zero rotation / direction
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "2", "--j", "6", "--edges", "edges.csv"],     # padded, quoted row
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "4", "--j", "4"],                              # identity edge
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "1", "--j", "0", "--edges", "edges.csv"],     # edge missing
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "1", "--edges", "edges_pipeline.csv"],  # schema mismatch
        ["--par", "par.txt", "--keyframes", "kf_shuffled.csv", "--i", "0", "--j", "3"],
        ["--par", "par.txt", "--keyframes", "kf_shuffled.csv", "--i", "0", "--j", "4"],                     # hole: empty image name
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "99"],                             # out of range
        ["--par", "par.txt", "--keyframes", "kf_unknown.csv", "--i", "0", "--j", "1"],                      # not in par
        ["--par", "par.txt", "--keyframes", "kf_noid.csv", "--i", "0", "--j", "1"],                         # unreadable CSV
        ["--par", "par.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "x"],                              # usage
        ["--par", "missing.txt", "--keyframes", "kf.csv", "--i", "0", "--j", "1"],                          # unreadable par
    ]
    out3 = {"files": files3, "cases": []}
    with tempfile.TemporaryDirectory() as d3:
        for n, txt in files3.items():
            open(os.path.join(d3, n), "w").write(txt)
        for args in cases3:
            r = subprocess.run([REF3] + args, cwd=d3, capture_output=True, text=True)
            out3["cases"].append({"args": args, "rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr})
            print("edge", args[4:], "->", r.returncode, r.stdout.count("\n"), "lines")
    json.dump(out3, open(os.path.join(HERE, "gt_keyframe_edge.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
