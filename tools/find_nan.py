"""Dev tool (GPU box): find (sample, pixel) pairs whose per-sample radiance is non-finite, compare with the oracle."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from platinum_amd import Renderer, scenes, abi
from platinum_amd.renderer import make_params
import oracle_lib

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
factory, W, H, spp, B = scenes.CONFIGS[wl]
sc = factory()
r = Renderer()
r.startRender(sc, (W, H), nsamp, max_bounces=B)
found = []
for s in range(nsamp):
    rad, hits = r.debugSample(s)
    bad = ~np.isfinite(rad[..., :3]).all(-1)
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in zip(ys, xs):
            found.append({"s": s, "x": int(x), "y": int(y), "rad": [float(v) for v in rad[y, x, :3]],
                          "hits": hits[:, y, x, :].tolist()})
    if len(found) > 20:
        break
print(json.dumps({"n": len(found), "found": found[:8]}))
if found:
    o = oracle_lib.OracleScene(sc, make_params(W, H, 1, B))
    f = found[0]
    rad_o, hits_o = o.debug_sample(f["s"])
    print("oracle at same sample/pixel:", rad_o[f["y"], f["x"]].tolist(), hits_o[:, f["y"], f["x"], :].tolist())
