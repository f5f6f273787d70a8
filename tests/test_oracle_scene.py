"""CPU: oracle scene-level checks — BVH vs brute force (the intersection contract is BVH-independent), the golden
fixtures minted by tools/make_golden.py, host-side constants."""
import os

import numpy as np
import pytest

import oracle_lib
from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("factory", [lambda: scenes.cornell_scene("bench"), lambda: scenes.cornell_sphere_scene(), lambda: scenes.field_scene(2)])
def test_oracle_bvh_equals_brute_force(factory):
    sc = factory()
    p = make_params(40, 24, 1, 5)
    a = oracle_lib.OracleScene(sc, p, use_bvh=True)
    b = oracle_lib.OracleScene(sc, p, use_bvh=False)
    pa, pb = a.trace_primary(0), b.trace_primary(0)
    assert pa.tobytes() == pb.tobytes()
    ra, ha = a.debug_sample(1)
    rb, hb = b.debug_sample(1)
    assert np.array_equal(ha, hb) and ra.tobytes() == rb.tobytes()


def test_c1_golden():
    g = np.load(os.path.join(G, "c1_cornell_golden.npz"))
    sc = scenes.cornell_scene("bench")
    o = oracle_lib.OracleScene(sc, make_params(512, 512, 4, 4))
    prim = o.trace_primary(0)
    c = slice(224, 288)
    assert np.array_equal(np.stack([prim["instance"], prim["primitive"]], -1)[c, c], g["prim_ids_crop"])
    assert np.stack([prim["t"], prim["u"], prim["v"]], -1)[c, c].tobytes() == g["prim_tuv_crop"].tobytes()
    acc1 = o.render(0, 1)
    assert acc1[c, c].tobytes() == g["acc1_crop"].tobytes()
    np.testing.assert_allclose(acc1[..., :3].mean((0, 1)), g["acc1_mean"], rtol=1e-6)


def test_c2_small_golden():
    g = np.load(os.path.join(G, "c2_small_golden.npz"))
    o = oracle_lib.OracleScene(scenes.cornell_sphere_scene(), make_params(160, 90, 2, 8))
    acc = o.render(0, 2)
    assert acc.tobytes() == g["acc2"].tobytes()
    rad0, hits0 = o.debug_sample(0)
    assert np.array_equal(hits0, g["hits0"].astype(np.int32)) and rad0.tobytes() == g["rad0"].tobytes()


def test_running_mean_is_batch_independent():
    o = oracle_lib.OracleScene(scenes.cornell_scene(), make_params(32, 32, 6, 4))
    a = o.render(0, 6)
    b = o.render(0, 2)
    b = o.render(2, 4, acc=b, acc_n0=2)
    assert a.tobytes() == b.tobytes()


def test_constants_cornell_default_camera():
    """updateConstants (renderer_pt.cpp:965-1021) on the reference's default camera node (scene_explorer.cpp:84-90)."""
    o = oracle_lib.OracleScene(scenes.cornell_scene("default"), make_params(640, 480, 1, 4))
    c = o.constants()
    assert (c.camera.position.x, c.camera.position.y, c.camera.position.z) == (-5.0, 5.0, 5.0)
    assert c.lightCount == 2 and c.totalLightPower == pytest.approx(2 * 50 * 2.0 * np.pi, rel=1e-6)  # 2 tris x (Le.g * area * pi)
    assert c.camera.apertureRadius == 0.0 and c.gmonBuckets == 1 and c.lutSizeE == 128 and c.lutSizeEavg == 128
    # BT709 -> BT2020 (colorspace.cpp): the textbook matrix
    idt = np.array([[v.x, v.y, v.z] for v in c.idt]).T
    np.testing.assert_allclose(idt, [[0.6274, 0.3293, 0.0433], [0.0691, 0.9195, 0.0114], [0.0164, 0.0880, 0.8956]], atol=2e-4)
    # pixel deltas: vw = focus * 36/28 (sensor aspect 1.5 > 4:3 => cropped height = 36/1.5... uses max(sensorAspect, aspect))
    du = np.array([c.camera.pixelDeltaU.x, c.camera.pixelDeltaU.y, c.camera.pixelDeltaU.z])
    vh = 1.0 * (36.0 / 1.5) / 28.0
    assert np.linalg.norm(du) * 640 == pytest.approx(vh * 640 / 480, rel=1e-5)


def test_gmon_resolve_properties():
    """gmon.metal: identical buckets -> that value (G = 0); one outlier bucket is trimmed when the Gini coefficient is high."""
    o = oracle_lib.OracleScene(scenes.cornell_scene(), make_params(4, 2, 15, 4, flags=abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON, gmon_buckets=15))
    b = np.zeros((15, 2, 4, 4), np.float32)
    b[..., :3] = 0.25
    b[..., 3] = 1
    out = o.gmon_resolve(b, 15)
    np.testing.assert_allclose(out[..., :3], 0.25, rtol=1e-6)
    b2 = b.copy()
    b2[7, 0, 0, :3] = 1000.0                      # a firefly in one bucket of one pixel
    out2 = o.gmon_resolve(b2, 15)
    assert out2[0, 0, 0] < 0.3                   # trimmed away (plain mean would be ~66.9)
    assert abs(out2[1, 3, 0] - 0.25) < 1e-6
    # cap = 0 -> plain mean of the buckets
    out3 = o.gmon_resolve(b2, 15, cap=0.0)
    assert out3[0, 0, 0] == pytest.approx((14 * 0.25 + 1000.0) / 15, rel=1e-5)


def test_gmon_bucket_assignment_quirk():
    """With GMoN the running-mean weight is frameIdx / gmonBuckets (kernel.metal:675), not the index inside the bucket:
    with spp = 8, 4 buckets (2 samples per bucket) every sample has weight index < 1 until frame 4, so buckets 0 and 1 end
    up holding just their LAST sample. Reproduced, not fixed."""
    sc = scenes.cornell_scene()
    p = make_params(24, 16, 8, 3, flags=abi.FLAG_MULTISCATTER_GGX | abi.FLAG_GMON, gmon_buckets=4)
    o = oracle_lib.OracleScene(sc, p)
    buckets, resolved = o.render_gmon(8)
    singles = [oracle_lib.OracleScene(sc, make_params(24, 16, 8, 3)).debug_sample(s)[0] for s in range(8)]
    assert buckets[0].tobytes() == singles[1].tobytes()        # frame 1 overwrote frame 0 (n = 1 // 4 = 0)
    assert buckets[1].tobytes() == singles[3].tobytes()
    # frames 4, 5 -> bucket 2 with n = 1 both times: ((L5 + ((L4 + 0 * 1) / 2) * 1) / 2): the first sample is halved twice
    f32 = np.float32
    b2 = ((singles[5][..., :3] + ((singles[4][..., :3] + f32(0)) / f32(2)) * f32(1)) / f32(2)).astype(np.float32)
    assert buckets[2][..., :3].tobytes() == b2.tobytes()
    assert resolved.shape == (16, 24, 4) and (resolved[..., 3] == 1).all()
