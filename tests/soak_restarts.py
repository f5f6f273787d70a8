#!/usr/bin/env python3
"""Soak (not collected by pytest): `python tests/soak_restarts.py SECONDS` keeps ONE renderer busy for that long the way an editing session
does — start, render a few samples, restart with another scene / size / integrator / accel structure / GMoN setting / batch size, now and then
a large frame — and checks (a) every small render against the oracle, bit for bit; (b) that free device memory stops shrinking once the
largest arrays have been allocated (kept allocations must not leak); (c) that nothing hangs (the caller runs it under `timeout`).  Prints one
JSON line.  Test infrastructure (it uses the oracle as the checker)."""
import ctypes as C
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


def free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    import torch
    torch.cuda.init()
    r = Renderer(device=0)
    rng = np.random.default_rng(20261004)
    small = [scenes.cornell_scene(), scenes.cornell_sphere_scene(), scenes.field_scene(4), scenes.textured_scene()] + [scenes.random_scene(s, extras=s % 2 == 0) for s in range(40, 52)]
    big = [scenes.field_scene(32), scenes.cornell_sphere_scene()]
    t0 = time.time()
    n = checked = big_frames = 0
    bad = []
    free_after_big = None
    min_free = free_bytes()
    while time.time() - t0 < budget:
        n += 1
        if n % 25 == 0:   # a large frame: full-size queues, the 1 M-triangle build
            sc = big[(n // 25) % 2]
            r.selectKernel(abi.INTEGRATOR_MIS)
            r.startRender(sc, (1920, 1080), 8, max_bounces=8, nonfinite_policy=abi.NONFINITE_ZERO)
            r.render(0); r.wait()
            a = r.readbackAccumulator()
            if not np.isfinite(a).all() or not (a[..., 3] == 1).all():
                bad.append(["big", n])
            big_frames += 1
            f = free_bytes()
            if big_frames == 2:
                free_after_big = f
            elif big_frames > 2 and f < free_after_big - (64 << 20):
                bad.append(["leak", n, int(free_after_big - f)])
            min_free = min(min_free, f)
            continue
        sc = small[int(rng.integers(0, len(small)))]
        w, h = [(96, 54), (71, 45), (128, 72), (33, 17)][int(rng.integers(0, 4))]
        B = int(rng.integers(1, 9))
        spp = int(rng.integers(1, 5))
        integ = abi.INTEGRATOR_MIS if rng.random() < 0.7 else abi.INTEGRATOR_SIMPLE
        accel = abi.ACCEL_TWO_LEVEL if rng.random() < 0.15 else abi.ACCEL_AUTO
        gmon = rng.random() < 0.15
        sif = int(rng.integers(0, 4))
        kw = dict(max_bounces=B, accel_structure=accel, samples_in_flight=sif)
        if gmon:
            kw.update(gmonBuckets=3, flags=abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON)
        r.selectKernel(integ)
        r.startRender(sc, (w, h), spp, **kw)
        calls = 0
        while r.renderProgress()[0] < spp and calls < 64:   # the frontend's loop: one sample per call, merged by the library
            r.render(1); calls += 1
        r.wait()
        if gmon:
            continue   # (GMoN images are checked by the parity tests; here the mode only has to coexist with the others)
        p = make_params(w, h, spp, B, integrator=integ)
        o = oracle_lib.OracleScene(sc, p)
        if not same(r.readbackAccumulator(), o.render(0, spp)):
            bad.append(["parity", n, sc.name, w, h, B, spp, int(integ), int(accel), sif])
        checked += 1
    r.close()
    print(json.dumps({"seconds": round(time.time() - t0, 1), "renders": n, "checked_against_oracle": checked, "large_frames": big_frames,
                      "failures": bad, "min_free_GiB": round(min_free / 2**30, 1)}))


if __name__ == "__main__":
    main()
