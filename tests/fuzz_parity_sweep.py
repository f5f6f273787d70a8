#!/usr/bin/env python3
"""One-off extended fuzz (not collected by pytest): `python tests/fuzz_parity_sweep.py FIRST COUNT [extras]` runs test_random_scene_fuzz_parity's
checks — constants, primary hits, per-bounce hit ids, per-sample radiance and the accumulated image of the HIP path against the oracle,
bit for bit — on scenes.random_scene(seed) for seed in [FIRST, FIRST + COUNT), with bounce counts and image sizes varied by seed, and
prints one JSON line.  With `extras`: scenes.random_scene(seed, extras=True) (every texture slot, degenerate material corners) and the
render parameters varied too (multiscatter flag, working space, first sample, samples in flight up to 5).  Test infrastructure (it uses the oracle as the checker)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from platinum_amd import Renderer, abi, scenes
from platinum_amd.renderer import make_params


def same(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    return np.array_equal(an, bn) and np.array_equal(np.where(an, 0, a).view(np.uint32), np.where(bn, 0, b).view(np.uint32))


def main():
    seed0, count = int(sys.argv[1]), int(sys.argv[2])
    extras = len(sys.argv) > 3 and sys.argv[3] == "extras"
    r = Renderer(device=0)
    bad, grazing, t0 = [], [], time.time()
    for seed in range(seed0, seed0 + count):
        sc = scenes.random_scene(seed, extras=extras)
        integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
        w, h = (96, 54) if seed % 3 else (71, 45)            # (odd sizes: partial edge tiles)
        B = 3 + seed % 7
        spp = 2 + seed % 2
        accel = abi.ACCEL_TWO_LEVEL if seed % 11 == 0 else abi.ACCEL_AUTO
        flags, space, first, sif = abi.FLAG_MULTISCATTER_GGX, scenes.BT2020, 0, 1 + seed % 3
        if extras:
            flags = abi.FLAG_MULTISCATTER_GGX if seed % 5 else 0
            space = scenes.BT2020 if seed % 7 else scenes.BT709
            first, spp, sif = (seed % 13) * 3, 2 + seed % 4, 1 + seed % 5
        if seed % 6 == 5:   # r4: big odd batches — a chunk of camera rays spans several pixels' samples (k_raygen, pixel-major), the accumulate tiles are ragged
            spp = 33 + seed % 41
            sif = spp if seed % 12 == 5 else 1 + (spp // 2)
            w, h = (40, 27) if seed % 3 else (33, 18)
        r.selectKernel(integ)
        r.startRender(sc, (w, h), spp, workingSpace=space, flags=flags, max_bounces=B, first_sample=first, accel_structure=accel, samples_in_flight=sif)
        p = make_params(w, h, spp, B, flags=flags, integrator=integ, working_space=space, first_sample=first)
        o = oracle_lib.OracleScene(sc, p)
        checks = [bytes(r.constants()) == bytes(o.constants()), r.tracePrimary(1).tobytes() == o.trace_primary(1).tobytes()]
        rg, hg = r.debugSample(first)
        rc, hc = o.debug_sample(first)
        checks += [bool(np.array_equal(hg, hc)), bool(same(rg, rc))]
        r.render(0)
        checks.append(bool(same(r.readbackAccumulator(), o.render(first, spp))))
        if not all(checks):
            # The contract defines a hit over ALL triangles; the oracle's BVH — like any BVH — may skip a triangle whose
            # Moeller-Trumbore test accepts a near-parallel ray with a large error in t (the hit point lies outside the triangle's inflated box).
            # Such a seed is re-checked against the oracle's BRUTE-FORCE traversal, the definition itself: equal there = the GPU follows the
            # contract and the oracle's tree was the one that skipped (r4: seed 20341, a shadow ray leaving one half of a quad and grazing the other).
            ob = oracle_lib.OracleScene(sc, p, use_bvh=False)
            if checks[0] and checks[1] and bool(same(r.readbackAccumulator(), ob.render(first, spp))):
                grazing.append(seed)
            else:
                bad.append([seed, checks])   # [constants, primary hits, per-bounce hit ids, per-sample radiance, accumulator]
    r.close()
    print(json.dumps({"first": seed0, "extras": extras, "count": count, "mismatching_seeds": bad,
                      "equal_to_brute_force_but_not_to_the_oracles_bvh": grazing, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
