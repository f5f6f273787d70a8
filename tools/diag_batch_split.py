"""GPU box: C5 (ingested atrium, 4K, 12 bounces) — where does the image of ONE planned batch differ from the same samples in batches of 16,
and which of the two equals the oracle?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params
import oracle_lib, export_gltf

W, H, B = 3840, 2160, 12
sc = export_gltf.atrium_through_ingestion(tempfile.mkdtemp())
r = Renderer(device=0)
imgs = {}
for sif in (0, 16, 8, 23):
    r.startRender(sc, (W, H), 46, max_bounces=B, samples_in_flight=sif)
    r.render(0); r.wait()
    st = r.stats()
    imgs[sif] = r.readbackAccumulator()
    print("sif", sif, "->", st.samples_in_flight, "batches", st.batches, "nonfinite", st.nonfinite_samples, "rays", st.closest_rays, st.shadow_rays, st.shaded_hits, flush=True)
a = imgs[0]
for sif in (16, 8, 23):
    b = imgs[sif]
    diff = ~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(axis=2)
    ys, xs = np.nonzero(diff)
    print("sif", sif, "differing pixels:", len(xs), list(zip(xs[:10].tolist(), ys[:10].tolist())))
b = imgs[16]
diff = ~((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all(axis=2)
ys, xs = np.nonzero(diff)
if len(xs):
    xy = np.stack([xs, ys], 1).astype(np.uint32)[:64]
    o = oracle_lib.OracleScene(sc, make_params(W, H, 46, B))
    ref = o.render_pixels(xy, 0, 46)
    for i, (x, y) in enumerate(xy):
        print((int(x), int(y)), "one batch", a[y, x, :3], "16s", b[y, x, :3], "oracle", ref[i, :3],
              "one==oracle", np.array_equal(a[y, x].view(np.uint32), ref[i].view(np.uint32)), "16==oracle", np.array_equal(b[y, x].view(np.uint32), ref[i].view(np.uint32)))
        # which sample?  running means after k samples
        if i < 4:
            for k in range(1, 47):
                rk = o.render_pixels(xy[i:i + 1], 0, k)
                print("   k", k, rk[0, :3], np.isfinite(rk[0]).all())
