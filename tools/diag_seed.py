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
extras = len(sys.argv) > 2 and sys.argv[2] == "extras"
sc = scenes.random_scene(seed, extras=extras)
integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
w, h = (96, 54) if seed % 3 else (71, 45)
B = 3 + seed % 7
spp = 2 + seed % 2
accel = abi.ACCEL_TWO_LEVEL if seed % 11 == 0 else abi.ACCEL_AUTO
FLAGS, space, first, sif = abi.FLAG_MULTISCATTER_GGX, scenes.BT2020, 0, 1 + seed % 3
if extras:
    FLAGS = abi.FLAG_MULTISCATTER_GGX if seed % 5 else 0
    space = scenes.BT2020 if seed % 7 else scenes.BT709
    first, spp, sif = (seed % 13) * 3, 2 + seed % 4, 1 + seed % 5
if seed % 6 == 5:
    spp = 33 + seed % 41
    sif = spp if seed % 12 == 5 else 1 + (spp // 2)
    w, h = (40, 27) if seed % 3 else (33, 18)
r = Renderer(device=0)
r.selectKernel(integ)
p = make_params(w, h, spp, B, flags=FLAGS, integrator=integ, working_space=space, first_sample=first)
o = oracle_lib.OracleScene(sc, p)
ob = oracle_lib.OracleScene(sc, p, use_bvh=False)
def START():
    r.startRender(sc, (w, h), spp, workingSpace=space, flags=FLAGS, max_bounces=B, first_sample=first, accel_structure=accel, samples_in_flight=sif)
START()
print("seed", seed, "size", w, h, "B", B, "spp", spp, "sif", sif, "tris", r.stats().triangles, "slots", r.stats().leaf_slots)
for s in range(first, first + spp):
    rg, hg = r.debugSample(s)
    rc, hc = o.debug_sample(s)
    d = (rg.view(np.uint32) != rc.view(np.uint32)) & ~(np.isnan(rg) & np.isnan(rc))
    print(" sample", s, "hits equal", np.array_equal(hg, hc), "radiance differing values", int(d.sum()), np.argwhere(d.any(-1))[:4].tolist())
START()
r.render(0)
a = r.readbackAccumulator()
ref = o.render(first, spp)
refb = ob.render(first, spp)
print(' gpu == brute force:', a.tobytes() == refb.tobytes(), ' oracle bvh == brute force:', ref.tobytes() == refb.tobytes())
d = (a.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(a) & np.isnan(ref))
px = np.argwhere(d.any(-1))
print(" accumulator differing pixels", len(px), px[:6].tolist())
for (y, x) in px[:4]:
    print("   pixel", y, x, "gpu", a[y, x], "oracle", ref[y, x])
    for s in range(first, first + spp):
        print("     sample", s, "gpu", r.debugSample(s)[0][y, x], "oracle", o.debug_sample(s)[0][y, x], "brute", ob.debug_sample(s)[0][y, x], "hits", r.debugSample(s)[1][:, y, x].tolist())
# step by step
START()
done = 0
while done < spp:
    r.render(1); r.wait(); done += 1
a2 = r.readbackAccumulator()
print(" one sample per render(): equal to oracle", a2.tobytes() == ref.tobytes(), "equal to batched", a2.tobytes() == a.tobytes())
