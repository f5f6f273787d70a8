"""The committed golden fixtures (tests/golden/, minted by tools/make_golden.py from the CPU oracle) fed to the HIP PATH.

These tests do not load the oracle: the expected values are the committed bytes.  SURVEY §8(c)(4) / BASELINE.md §5: the
parity gate is C1 (BASELINE.json configs[0]: Cornell box, 512x512, 4 bounces) at 1, 4 and the full 64 spp — hit ids
bit-exact, accumulator bit-identical (stated fallback tolerance 1e-5 relative per sample)."""
import hashlib
import os

import numpy as np
import pytest

from platinum_amd import abi, scenes

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
C = slice(224, 288)


def test_c1_full_config_against_the_committed_golden(gpu_renderer):
    g = np.load(os.path.join(G, "c1_cornell_golden.npz"))
    sc = scenes.cornell_scene("bench")
    r = gpu_renderer
    r.selectKernel(abi.INTEGRATOR_MIS)
    r.startRender(sc, (512, 512), 64, max_bounces=4)      # C1 as BASELINE.json states it
    # primary rays: ids by checksum over the whole image, (t, u, v) on the crop
    prim = r.tracePrimary(0)
    ids = np.stack([prim["instance"], prim["primitive"]], -1)
    assert sha(ids) == str(g["prim_ids_sha"]) and np.array_equal(ids[C, C], g["prim_ids_crop"])
    assert np.stack([prim["t"], prim["u"], prim["v"]], -1)[C, C].tobytes() == g["prim_tuv_crop"].tobytes()
    # the accumulator after 1, 4 and 64 samples (progressive: 1 + 3 + 60)
    for upto, step in ((1, 1), (4, 3), (64, 60)):
        r.render(step)
        acc = r.readbackAccumulator()
        assert r.renderProgress() == (upto, 64)
        np.testing.assert_allclose(acc[C, C], g["acc%d_crop" % upto], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(acc[..., :3].mean((0, 1)), g["acc%d_mean" % upto], rtol=1e-6)
        assert acc[C, C].tobytes() == g["acc%d_crop" % upto].tobytes()
        assert sha(acc) == str(g["acc%d_sha" % upto]), "accumulator at %d spp is not bit-identical to the golden image" % upto
    assert r.status() == abi.STATUS_READY | abi.STATUS_DONE


def test_c2_small_against_the_committed_golden(gpu_renderer):
    g = np.load(os.path.join(G, "c2_small_golden.npz"))
    r = gpu_renderer
    r.selectKernel(abi.INTEGRATOR_MIS)
    r.startRender(scenes.cornell_sphere_scene(), (160, 90), 2, max_bounces=8)
    rad0, hits0 = r.debugSample(0)
    assert np.array_equal(hits0, g["hits0"].astype(np.int32))          # (instance, primitive) at every bounce of every path
    np.testing.assert_allclose(rad0, g["rad0"], rtol=1e-5, atol=1e-7)
    assert rad0.tobytes() == g["rad0"].tobytes()
    r.render(0)
    assert r.readbackAccumulator().tobytes() == g["acc2"].tobytes()
