"""GPU box: per-sample radiance of the two C5 pixels under one 46-sample batch vs batches of 16 vs the oracle."""
import os, sys, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params
import oracle_lib, export_gltf

W, H, B = 3840, 2160, 12
sc = export_gltf.atrium_through_ingestion(tempfile.mkdtemp())
r = Renderer(device=0)
o = oracle_lib.OracleScene(sc, make_params(W, H, 46, B))
for (x, y) in ((2544, 532), (3062, 895)):
    os.environ["PTAMD_DEBUG_PIXEL"] = "%d,%d" % (x, y)
    for sif in (0, 16):
        print("=== pixel", x, y, "sif", sif, flush=True); sys.stderr.flush()
        r.startRender(sc, (W, H), 46, max_bounces=B, samples_in_flight=sif)
        while r.renderProgress()[0] < 46:
            r.render(sif or 46); r.wait()
        sys.stderr.flush()
    print("=== oracle", x, y, flush=True)
    xy = np.array([[x, y]], np.uint32)
    for s in range(46):
        v = o.render_pixels(xy, s, 1)[0]
        b = v.view(np.uint32)
        print("oracle pixel %d,%d sample %d: %08x %08x %08x  %.9g %.9g %.9g" % (x, y, s, b[0], b[1], b[2], v[0], v[1], v[2]), flush=True)
