#!/usr/bin/env python3
"""Diagnose one fuzz seed (tests/fuzz_parity_sweep.py's parameters): where does the accumulator differ from the oracle's?  usage: diag_seed.py SEED"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params

seed = int(sys.argv[1])
sc = scenes.random_scene(seed)
integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
w, h = (96, 54) if seed % 3 else (71, 45)
B = 3 + seed % 7
spp = 2 + seed % 2
sif = 1 + seed % 3
if len(sys.argv) > 2: sif = int(sys.argv[2])
r = Renderer(device=0)
r.selectKernel(integ)
p = make_params(w, h, spp, B, flags=abi.FLAG_MULTISCATTER_GGX, integrator=integ)
o = oracle_lib.OracleScene(sc, p)
r.startRender(sc, (w, h), spp, flags=abi.FLAG_MULTISCATTER_GGX, max_bounces=B, samples_in_flight=sif)
print("seed", seed, "size", w, h, "B", B, "spp", spp, "sif", sif, "tris", r.stats().triangles, "slots", r.stats().leaf_slots)
for s in range(spp):
    rg, hg = r.debugSample(s)
    rc, hc = o.debug_sample(s)
    d = (rg.view(np.uint32) != rc.view(np.uint32)) & ~(np.isnan(rg) & np.isnan(rc))
    print(" sample", s, "hits equal", np.array_equal(hg, hc), "radiance differing values", int(d.sum()), np.argwhere(d.any(-1))[:4].tolist())
r.startRender(sc, (w, h), spp, flags=abi.FLAG_MULTISCATTER_GGX, max_bounces=B, samples_in_flight=sif)
r.render(0)
a = r.readbackAccumulator()
ref = o.render(0, spp)
d = (a.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(a) & np.isnan(ref))
px = np.argwhere(d.any(-1))
print(" accumulator differing pixels", len(px), px[:6].tolist())
for (y, x) in px[:4]:
    print("   pixel", y, x, "gpu", a[y, x], "oracle", ref[y, x])
    for s in range(spp):
        print("     sample", s, "gpu", r.debugSample(s)[0][y, x], "oracle", o.debug_sample(s)[0][y, x])
# step by step
r.startRender(sc, (w, h), spp, flags=abi.FLAG_MULTISCATTER_GGX, max_bounces=B, samples_in_flight=sif)
done = 0
while done < spp:
    r.render(1); r.wait(); done += 1
a2 = r.readbackAccumulator()
print(" one sample per render(): equal to oracle", a2.tobytes() == ref.tobytes(), "equal to batched", a2.tobytes() == a.tobytes())
