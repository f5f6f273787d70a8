import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from platinum_amd import Renderer, scenes, abi
factory, W, H, spp, B = scenes.CONFIGS["c2"]
sc = factory()
r = Renderer()
def report(tag, a):
    bad = ~np.isfinite(a[..., :3]).all(-1)
    ys, xs = np.nonzero(bad)
    print(tag, "nonfinite pixels:", int(bad.sum()), list(zip(xs[:5].tolist(), ys[:5].tolist())), "mean finite", float(a[..., :3][~bad].mean()))
for S in (1, 8):
    r.startRender(sc, (W, H), 64, max_bounces=B, samples_in_flight=S)
    r.render(0)
    report(f"internal S={S}", r.readbackAccumulator())
acc = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
r.startRender(sc, (W, H), 64, max_bounces=B, samples_in_flight=8, external_accumulator=acc.data_ptr())
for _ in range(8): r.render(8)
r.wait()
report("external S=8", acc.cpu().numpy())
r.setProfiling(True)
r.startRender(sc, (W, H), 256, max_bounces=B, samples_in_flight=8)
for _ in range(32): r.render(8)
a = r.readbackAccumulator()
report("internal S=8 256spp profiling", a)
bad = ~np.isfinite(a[..., :3]).all(-1)
if bad.any():
    ys, xs = np.nonzero(bad)
    x, y = int(xs[0]), int(ys[0])
    r.startRender(sc, (W, H), 256, max_bounces=B, samples_in_flight=1)
    for s in range(256):
        rad, hits = r.debugSample(s)
        if not np.isfinite(rad[y, x, :3]).all():
            print("pixel", x, y, "sample", s, rad[y, x].tolist(), hits[:, y, x].tolist()); break
