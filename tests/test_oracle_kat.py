"""CPU: the oracle against every pin available for this path: the integer sampler known-answers, the LUT
data files (blob sha + spot values from the tinyexr decode), closed-form BSDF values, and a white-furnace check that
pins the LUT axis convention against the oracle's own GGX."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib
from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params

G = os.path.join(os.path.dirname(__file__), "golden")
L = oracle_lib.lib()


@pytest.fixture(scope="module")
def osc():
    return oracle_lib.OracleScene(scenes.cornell_scene(), make_params(8, 8, 1, 4))


def test_pcg4d_offsets_match_survey_kats():
    kat = json.load(open(os.path.join(G, "sampler_kat.json")))
    for x, y, s, want in kat["offsets"]:
        assert L.orc_halton_offset(x, y, s) == want


def test_halton_exact_values_and_primes():
    kat = json.load(open(os.path.join(G, "sampler_kat.json")))
    for i, d, want in kat["halton_exact"]:
        assert L.orc_halton(i, d) == np.float32(want)
    assert L.orc_prime(0) == 2 and L.orc_prime(619) == 4583 and L.orc_prime(620) == 0  # defs.metal:115-194
    for i, d, bits in kat["halton_bits"]:
        assert int(np.float32(L.orc_halton(i, d)).view(np.uint32)) == bits
    # clamp to 1 - eps (defs.metal:22) and range
    vals = [L.orc_halton(0xFFFFFFFF, d) for d in range(0, 620, 37)]
    assert all(0.0 <= v < 1.0 for v in vals)


def test_halton_matches_independent_python_radical_inverse():
    def radical_inverse(i, b):
        f, r, inv = np.float32(1), np.float32(0), np.float32(1) / np.float32(b)
        while i > 0:
            f = np.float32(f * inv)
            r = np.float32(r + np.float32(f * np.float32(i % b)))
            i //= b
        return min(r, np.float32(np.nextafter(np.float32(1), np.float32(0))))
    rng = np.random.default_rng(7)
    for _ in range(200):
        i = int(rng.integers(0, 2**32)); d = int(rng.integers(0, 620))
        assert L.orc_halton(i, d) == radical_inverse(i, L.orc_prime(d))


def test_lut_blob_is_the_reference_data():
    blob = open(abi.LUT_PATH, "rb").read()
    assert hashlib.sha256(blob).hexdigest() == "be57e4067611c1421a0b9bf1f9c772f9fc102b51fb9dfa060ddb29380a07280e"
    assert blob[:8] == b"PTLUT01\0"
    hdr = np.frombuffer(blob, dtype="<u4", count=33, offset=8)
    assert hdr[0] == 8
    dims = hdr[1:].reshape(8, 4)
    assert dims[:, :3].tolist() == [[128, 128, 1], [128, 1, 1], [32, 32, 32], [32, 32, 1], [32, 32, 32], [32, 32, 32], [32, 32, 1], [32, 32, 1]]
    data = np.frombuffer(blob, dtype="<f4", offset=12 + 128)
    E = data[dims[0, 3]:dims[0, 3] + 128 * 128]
    # spot values recorded by the surveyor from the tinyexr decode
    assert abs(E.min() - 0.3116) < 1e-4 and abs(E.max() - 1.00003) < 1e-5 and abs(E.mean() - 0.8300) < 1e-4
    Eavg = data[dims[1, 3]:dims[1, 3] + 128]
    assert abs(Eavg.min() - 0.4127) < 1e-4
    assert abs(data[dims[2, 3]] - 0.701358) < 1e-6          # ggx_ms_E_0 first texel
    assert abs(data[dims[6, 3]:dims[6, 3] + 1024].min() - 0.5461) < 1e-4   # ggx_E_trans_in_avg


def test_lut_sampling_texel_centres_and_clamp(osc):
    blob = open(abi.LUT_PATH, "rb").read()
    hdr = np.frombuffer(blob, dtype="<u4", count=32, offset=12).reshape(8, 4)
    data = np.frombuffer(blob, dtype="<f4", offset=12 + 128)
    E = data[hdr[0, 3]:hdr[0, 3] + 128 * 128].reshape(128, 128)
    # exact texel centres (ms_lut_gen.metal:348-349: (i + 0.5) / N) return the texel
    for (x, y) in [(0, 0), (5, 17), (127, 127), (64, 3)]:
        assert L.orc_lut_sample(osc.h, 0, (x + 0.5) / 128, (y + 0.5) / 128, 0) == pytest.approx(E[y, x], abs=1e-6)
    # clamp to edge
    assert L.orc_lut_sample(osc.h, 0, -1.0, 0.5 / 128, 0) == E[0, 0]
    assert L.orc_lut_sample(osc.h, 0, 2.0, 2.0, 0) == E[127, 127]
    # 3-D: EMs[z][y][x]
    EMs = data[hdr[2, 3]:hdr[2, 3] + 32 ** 3].reshape(32, 32, 32)
    assert L.orc_lut_sample(osc.h, 2, 3.5 / 32, 7.5 / 32, 20.5 / 32) == pytest.approx(EMs[20, 7, 3], abs=1e-6)
    # midpoint between two texels along x is their mean
    assert L.orc_lut_sample(osc.h, 0, 11.0 / 128, 9.5 / 128, 0) == pytest.approx(0.5 * (E[9, 10] + E[9, 11]), abs=1e-6)


def test_closed_forms():
    assert L.orc_fresnel(1.0, 1.5) == pytest.approx(0.04, abs=1e-6)          # ((1.5-1)/(1.5+1))^2
    assert L.orc_fresnel(0.0, 1.5) == pytest.approx(1.0, abs=1e-6)
    assert L.orc_avg_dielectric_fresnel_fit(1.5) == pytest.approx(0.08950, abs=2e-5)
    assert L.orc_fresnel(0.2, 1.0 / 1.5) == 1.0                                # total internal reflection


def test_deterministic_transcendentals_accuracy():
    xs = np.linspace(0, 2 * np.pi, 4001, dtype=np.float32)
    s, c = C.c_float(), C.c_float()
    err = 0.0
    for x in xs:
        L.orc_sincos(float(x), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert err < 3e-7
    for v in (0.01, 0.3, 0.9999, 1.0, 2.5, 37.0):
        assert L.orc_log2(v) == pytest.approx(np.log2(v), abs=3e-6, rel=3e-6)
    for v in (-5.25, -0.5, 0.0, 0.3, 1.0, 4.75):
        assert L.orc_exp2(v) == pytest.approx(2.0 ** v, rel=3e-6)


def test_warps():
    out3 = (C.c_float * 3)()
    for u in [(0.1, 0.2), (0.9, 0.5), (0.5, 0.999)]:
        L.orc_sample_cosine_hemisphere(u[0], u[1], out3)
        v = np.array(out3[:])
        assert abs(np.linalg.norm(v) - 1) < 1e-6 and v[2] >= 0 and abs(v[2] - np.sqrt(1 - u[1])) < 1e-6
    out2 = (C.c_float * 2)()
    rng = np.random.default_rng(0)
    for _ in range(100):
        a, b = rng.random(2)
        L.orc_sample_tri_uniform(a, b, out2)
        assert out2[0] >= 0 and out2[1] >= 0 and out2[0] + out2[1] <= 1 + 1e-6


def test_white_furnace_pins_lut_axes(osc):
    """E(cos, rough) of the LUT must equal the directional albedo of the oracle's own single-scatter GGX (white metal,
    no multiscatter flag would need another scene; instead: f*cos/pdf of sampleMetallic minus the multiscatter term is
    not separable, so compare WITH multiscatter: a white metal must then conserve energy, albedo ~ 1)."""
    mat = scenes.Material(base_color=(1, 1, 1, 1), roughness=0.6, metallic=1.0).to_gpu()
    rng = np.random.default_rng(3)
    for cos in (0.9, 0.5):
        wo = (C.c_float * 3)(float(np.sqrt(1 - cos * cos)), 0.0, cos)
        acc, n = 0.0, 20000
        out = (C.c_float * 11)()
        for _ in range(n):
            r = (C.c_float * 4)(*rng.random(4).astype(np.float32)); rc = (C.c_float * 2)(0.5, 0.5)
            r[3] = 0.5
            L.orc_bsdf_sample(osc.h, C.byref(mat), wo, r, rc, out)
            if out[9] > 0 and int(out[10]) & 2:
                acc += out[3] * abs(out[2]) / out[9]
        albedo = acc / n
        assert 0.93 < albedo < 1.07, (cos, albedo)  # Kulla-Conty compensated white metal ~ energy conserving


def test_bsdf_sample_pdf_matches_eval_pdf(osc):
    """For the default opaque-dielectric material the pdf returned by sample() for a glossy/diffuse direction must be
    the lobe's share of what eval() reports for the same pair (mixture pdf = sum of both lobes)."""
    mat = scenes.Material(base_color=(0.7, 0.6, 0.5, 1), roughness=0.5).to_gpu()
    wo = (C.c_float * 3)(0.3, 0.1, float(np.sqrt(1 - 0.1)))
    rng = np.random.default_rng(5)
    out = (C.c_float * 11)(); ev = (C.c_float * 4)()
    n_ok = 0
    for _ in range(300):
        r = (C.c_float * 4)(*rng.random(4).astype(np.float32)); rc = (C.c_float * 2)(0.3, 0.6)
        L.orc_bsdf_sample(osc.h, C.byref(mat), wo, r, rc, out)
        wi = (C.c_float * 3)(out[0], out[1], out[2])
        if out[2] < 2e-3 or out[9] <= 0:
            continue
        L.orc_bsdf_eval(osc.h, C.byref(mat), wo, wi, ev)
        assert ev[3] >= out[9] * (1 - 1e-4)     # mixture pdf >= the sampled lobe's pdf
        assert np.isfinite(ev[3]) and all(np.isfinite(ev[k]) and ev[k] >= 0 for k in range(3))
        n_ok += 1
    assert n_ok > 200
