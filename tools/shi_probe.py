#!/usr/bin/env python3
"""Corner fixpoint on the device (sfmx_shi_tomasi_candidates_pruned) under the schedule the environment selects (SFMX_SHI_MODE /
SFMX_SHI_SWEEPS): survivors / undecided pixels that travel to the host, wall time per call (hipGraph replay) and the GPU time
of the fixpoint group (HIP events, launched kernel by kernel), for a VGA bench frame, a C3-like frame (min_distance 4) and a
1920x1080 frame.  A hash of the final corner pick (tracker seam) lets two runs with different schedules be compared."""
import hashlib, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
pipe = importlib.import_module(I.PKG + ".pipeline")
ctx = I.capi.Context(0)
mode = os.environ.get("SFMX_SHI_MODE", "sweeps:" + os.environ.get("SFMX_SHI_SWEEPS", "default"))


def probe(tag, img, md, max_tracks):
    h, w = img.shape
    pyr = ctx.pyramid(img, 1)
    for _ in range(3):
        r = ctx.shi_candidates_pruned(pyr, 0.01, md)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.shi_candidates_pruned(pyr, 0.01, md)
    wall = (time.perf_counter() - t0) / 20 * 1e6
    ctx.set_timing(True)
    for _ in range(5):
        ctx.shi_candidates_pruned(pyr, 0.01, md)
    prof = ctx.kernel_profile()
    ctx.set_timing(False)
    fix = [v for k, v in prof.items() if k.startswith("shi fixpoint")][0]
    sc = [v for k, v in prof.items() if k.startswith("k_shi_score")][0]
    trk = pipe.Tracker(ctx, w, h, max_tracks=max_tracks, min_tracks=max_tracks - 1, min_distance=md)
    trk.step(img)
    xy, ids = trk.tracks()
    trk.close()
    print(f"{mode:>18s} | {tag}: survivors {r[4]:6d} of {r[5]:7d} candidates, undecided {int((~r[2]).sum()):6d} | call wall {wall:7.1f} us | "
          f"fixpoint group {fix[0] / max(1, fix[1]):7.1f} us, score {sc[0] / max(1, sc[1]):6.1f} us | corners {len(xy)} hash {hashlib.sha1(xy.tobytes()).hexdigest()[:12]}",
          flush=True)


seq = I.synth.make_sequence(1, 640, 480, 0.3, n_blobs=20000, seed=7)
probe("vga md=8 ", seq["images"][0], 8, 2200)
seq3 = I.synth.make_sequence(1, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
probe("c3  md=4 ", seq3["images"][0], 4, 5000)
seq5 = I.synth.make_sequence(1, 1920, 1080, 0.1, n_blobs=20000, seed=13)
probe("1080p md=8", seq5["images"][0], 8, 2200)
flat = I.synth.make_sequence(1, 320, 240, 0.3, n_blobs=6000, seed=11, noise=False)
probe("ties md=8 ", flat["images"][0], 8, 2200)
probe("ties md=16", flat["images"][0], 16, 2200)
probe("ties md=2 ", flat["images"][0], 2, 2200)
