#!/usr/bin/env python3
"""Pin A7-A9 against reference-held data: re-integrate texels of the eight GGX energy tables with the ORACLE's BSDF pieces,
following the reference's generator (ms_lut_gen.metal:337-743, restated in oracle/pt_oracle.cpp `lutgen`), and compare with the
values the reference committed (resource/lut/*.exr -> platinum_amd/data/ggx_luts.bin).

    python tools/lut_pin.py [--texels 400] [--samples 16384] [--lambda-mode 0|1] [--threads 8]

Prints, per table, the mean / 95th percentile / max absolute deviation over a seeded random set of texels.  The CPU test
tests/test_lut_pin.py runs a smaller set with the tolerances derived from this tool's output."""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["E", "E_avg", "E_ms", "E_ms_avg", "E_trans_in", "E_trans_out", "E_trans_in_avg", "E_trans_out_avg"]
# What the committed data files correspond to (found with this tool): E_ms / E_ms_avg hold the integrand WITHOUT its multiscatter term
# (mode bit 1), E does not carry the 0.961 "funny hack" on its low-roughness grazing corner (mode bit 2).
MS_TABLES_MODE = {0: 4, 2: 2, 3: 2}
SHAPES = [(128, 128, 1), (128, 1, 1), (32, 32, 32), (32, 32, 1), (32, 32, 32), (32, 32, 32), (32, 32, 1), (32, 32, 1)]


def texel_set(which, n, seed=1234):
    w, h, d = SHAPES[which]
    rng = np.random.default_rng(seed + which)
    total = w * h * d
    idx = rng.choice(total, size=min(n, total), replace=False)
    return [(int(i % w), int(i // w % h), int(i // (w * h))) for i in idx]


def deviations(o, which, texels, samples, lambda_mode=0, threads=8):
    def one(t):
        x, y, z = t
        return o.L.orc_lut_regen_texel(o.h, which, x, y, z, samples, 7, lambda_mode) - o.L.orc_lut_texel(o.h, which, x, y, z)
    with ThreadPoolExecutor(threads) as ex:  # (ctypes releases the GIL)
        return np.array(list(ex.map(one, texels)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--texels", type=int, default=400)
    ap.add_argument("--samples", type=int, default=16384)
    ap.add_argument("--lambda-mode", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--ms-term-as-written", action="store_true",
                    help="integrate exactly as ms_lut_gen.metal is written today (E_ms / E_ms_avg with fresnel_ms * brdf_ms, :252-282; E with "
                         "the 0.961 corner factor, :371-374); the committed tables match the integrals WITHOUT them")
    a = ap.parse_args()
    import oracle_lib
    from platinum_amd import scenes
    from platinum_amd.renderer import make_params
    o = oracle_lib.OracleScene(scenes.cornell_scene("bench"), make_params(8, 8, 1, 1))
    print("table             texels   mean|d|     p95|d|     max|d|     mean d")
    for which, name in enumerate(NAMES):
        mode = a.lambda_mode | (0 if a.ms_term_as_written else MS_TABLES_MODE.get(which, 0))
        d = deviations(o, which, texel_set(which, a.texels), a.samples, mode, a.threads)
        print(f"{name:16s} {len(d):7d}  {np.abs(d).mean():9.2e}  {np.percentile(np.abs(d), 95):9.2e}  {np.abs(d).max():9.2e}  {d.mean():+9.2e}")


if __name__ == "__main__":
    main()
