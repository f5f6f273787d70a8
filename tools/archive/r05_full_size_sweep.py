#!/usr/bin/env python3
"""GPU box, one-off: the full-size comparison of tests/test_gpu_full_size.py (one planned batch of the bench configuration against the oracle on ~1 000
probe pixels, bit for bit) repeated for many DIFFERENT sample ranges (first_sample = k * S) — the suite checks samples 0..S-1 only."""
import json, os, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params
import oracle_lib, test_gpu_full_size as T, export_gltf

r = Renderer(device=0)
out = []
for wl, S, reps in (("c3", 128, int(os.environ.get("REPS_C3", "12"))), ("c2", 128, int(os.environ.get("REPS_C2", "12"))), ("c5", 46, int(os.environ.get("REPS_C5", "4")))):
    factory, W, H, _spp, B = scenes.CONFIGS[wl]
    sc = export_gltf.atrium_through_ingestion(tempfile.mkdtemp()) if wl == "c5" else factory()
    r.startRender(sc, (W, H), 1, max_bounces=B)
    ids = r.tracePrimary(0)["instance"]
    bad_total = 0
    t0 = time.time()
    for k in range(1, reps + 1):
        first = k * S
        r.startRender(sc, (W, H), S, max_bounces=B, first_sample=first, samples_in_flight=S)
        r.render(S); r.wait()
        acc = r.readbackAccumulator()
        xy = T._probe_pixels(W, H, ids, seed=100 + k)
        o = oracle_lib.OracleScene(sc, make_params(W, H, S, B, first_sample=first))
        ref = o.render_pixels(xy, first, S)
        o.close()
        got = acc[xy[:, 1], xy[:, 0]]
        bad = ~((got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))).all(axis=1)
        bad_total += int(bad.sum())
        if bad.any():
            print(wl, "first_sample", first, "MISMATCH at", xy[bad][:5].tolist(), flush=True)
    line = {"workload": wl, "batches": reps, "samples_each": S, "probe_pixels_each": int(len(xy)), "mismatching_pixels": bad_total, "seconds": round(time.time() - t0, 1)}
    print(json.dumps(line), flush=True)
    out.append(line)
