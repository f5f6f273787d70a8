#!/usr/bin/env python3
"""One-off extended fuzz (not collected by pytest): `python tests/fuzz_parity_sweep.py FIRST COUNT` runs test_random_scene_fuzz_parity's
checks — constants, primary hits, per-bounce hit ids, per-sample radiance and the accumulated image of the HIP path against the oracle,
bit for bit — on scenes.random_scene(seed) for seed in [FIRST, FIRST + COUNT), with bounce counts and image sizes varied by seed, and
prints one JSON line.  Test infrastructure (it uses the oracle as the checker)."""
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
    first, count = int(sys.argv[1]), int(sys.argv[2])
    r = Renderer(device=0)
    bad, t0 = [], time.time()
    for seed in range(first, first + count):
        sc = scenes.random_scene(seed)
        integ = abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE
        w, h = (96, 54) if seed % 3 else (71, 45)            # (odd sizes: partial edge tiles)
        B = 3 + seed % 7
        spp = 2 + seed % 2
        accel = abi.ACCEL_TWO_LEVEL if seed % 11 == 0 else abi.ACCEL_AUTO
        r.selectKernel(integ)
        r.startRender(sc, (w, h), spp, max_bounces=B, accel_structure=accel, samples_in_flight=1 + seed % 3)
        p = make_params(w, h, spp, B, integrator=integ)
        o = oracle_lib.OracleScene(sc, p)
        checks = [bytes(r.constants()) == bytes(o.constants()), r.tracePrimary(1).tobytes() == o.trace_primary(1).tobytes()]
        rg, hg = r.debugSample(0)
        rc, hc = o.debug_sample(0)
        checks += [bool(np.array_equal(hg, hc)), bool(same(rg, rc))]
        r.render(0)
        checks.append(bool(same(r.readbackAccumulator(), o.render(0, spp))))
        if not all(checks):
            bad.append([seed, checks])   # [constants, primary hits, per-bounce hit ids, per-sample radiance, accumulator]
    r.close()
    print(json.dumps({"first": first, "count": count, "mismatching_seeds": bad, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
