"""CPU: the product's per-path stage functions (pt_sampler.h, pt_bsdf.h, pt_bvh.h, pt_shade.h, host_scene.h) compiled
for the host by the tests/emu harness must agree bit-for-bit with the oracle.  This is a debugging net for the
authoring container (no GPU); the real parity tests are the `-m gpu` ones that go through libptamd.so."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params


@pytest.mark.parametrize("factory,integrator", [
    (lambda: scenes.cornell_scene("bench"), abi.INTEGRATOR_MIS),
    (lambda: scenes.cornell_scene("default"), abi.INTEGRATOR_SIMPLE),
    (lambda: scenes.cornell_sphere_scene(), abi.INTEGRATOR_MIS),
    (lambda: scenes.field_scene(3), abi.INTEGRATOR_MIS),
])
def test_stage_functions_bit_exact_vs_oracle(factory, integrator):
    sc = factory()
    p = make_params(72, 40, 2, 6, integrator=integrator)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert bytes(o.constants()) == bytes(e.constants())
    assert [bytes(a) for a in o.lights()] == [bytes(b) for b in e.lights()]
    assert o.trace_primary(1).tobytes() == e.trace_primary(1).tobytes()
    for s in (0, 1):
        ro, ho = o.debug_sample(s)
        re_, he = e.debug_sample(s)
        assert np.array_equal(ho, he)
        assert ro.tobytes() == re_.tobytes()


def test_magic_division_halton_equals_oracle():
    e = emu_lib.EmuScene(scenes.cornell_scene(), make_params(8, 8, 1, 2))
    L = oracle_lib.lib()
    rng = np.random.default_rng(11)
    for i in [0, 1, 2, 0xFFFFFFFF, 0x80000000, 65535, 65536, 59049, 59048, 14641 * 14641 - 1] + [int(v) for v in rng.integers(0, 2**32, 1500)]:
        for d in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 52, 53, 54, 55, 100, 256, 257, 302, 303, 619):  # every digits-per-chunk class
            assert e.halton(i, d) == L.orc_halton(i, d), (i, d)


def test_thin_lens_camera_and_clearcoat_material():
    sc = scenes.cornell_sphere_scene(transmission=0.0)
    sc.nodes[1].materials[0].clearcoat = 0.8
    sc.nodes[1].materials[0].metallic = 0.5
    sc.nodes[1].materials[0].anisotropy = 0.4
    sc.camera.aperture = 2.8; sc.camera.focus_distance = 12.0; sc.camera.roundness = 0.3; sc.camera.bokeh_power = 0.5
    p = make_params(64, 36, 1, 5)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert o.constants().camera.apertureRadius > 0
    ro, ho = o.debug_sample(0)
    re_, he = e.debug_sample(0)
    assert np.array_equal(ho, he) and ro.tobytes() == re_.tobytes()


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_scene_fuzz_stage_functions(seed):
    """scenes.random_scene: arbitrary TRS (mirrored / non-uniform scale), the whole material parameter space with the exact
    0 / 1 corners over-represented, thin-lens cameras, textures, environment, both integrators."""
    sc = scenes.random_scene(seed)
    p = make_params(48, 27, 1, 6, integrator=abi.INTEGRATOR_MIS if seed % 4 else abi.INTEGRATOR_SIMPLE)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert bytes(o.constants()) == bytes(e.constants())
    assert [bytes(a) for a in o.lights()] == [bytes(b) for b in e.lights()]
    ro, ho = o.debug_sample(0)
    re_, he = e.debug_sample(0)
    nan = np.isnan(ro)
    assert np.array_equal(ho, he) and np.array_equal(nan, np.isnan(re_))
    assert np.array_equal(ro.view(np.uint32)[~nan], re_.view(np.uint32)[~nan])


@pytest.mark.parametrize("ulp", [-1, 1, 2])
@pytest.mark.parametrize("name", ["field3", "random3", "random7", "random12", "textured"])
def test_no_answer_depends_on_the_last_bit_of_the_rays_reciprocal_direction(name, ulp, monkeypatch):
    """on the device the reciprocal direction that feeds the slab tests comes from v_rcp_f32 (within 1 ulp of 1 / d), on the host from
    an IEEE division; the slab comparison's slack (pt_bvh.h kSlabSlack: the error budget is written there) has to absorb that.  Here the host build
    of the product's 6-wide / pair-slot traversal moves every component of the reciprocal by one ulp — down, up, or alternating — and has to find
    the same hits and the same radiance as the oracle, bit for bit."""
    monkeypatch.setenv("EMU_WIDE6", "1"); monkeypatch.setenv("EMU_PAIRS", "1"); monkeypatch.setenv("EMU_MORTON", "1"); monkeypatch.setenv("EMU_PLOC", "8")
    monkeypatch.setenv("EMU_RCP_ULP", str(ulp))
    sc = {"field3": lambda: scenes.field_scene(3), "textured": scenes.textured_scene}.get(name) or (lambda: scenes.random_scene(int(name[6:])))
    sc = sc()
    p = make_params(64, 36, 2, 6)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    monkeypatch.delenv("EMU_RCP_ULP")
    assert o.trace_primary(1).tobytes() == e.trace_primary(1).tobytes()
    for s_ in (0, 1):
        ro, ho = o.debug_sample(s_)
        re_, he = e.debug_sample(s_)
        nan = np.isnan(ro)
        assert np.array_equal(ho, he) and np.array_equal(nan, np.isnan(re_))
        assert np.array_equal(ro.view(np.uint32)[~nan], re_.view(np.uint32)[~nan])
    emu_lib.EmuScene(sc, p)   # (the switch is process-wide in the harness: the next scene created without the variable turns it off again)


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("name", ["cornell_sphere", "field3", "random5", "random11", "textured"])
def test_six_wide_nodes_give_the_same_hits_and_radiance(name, pairs, monkeypatch):
    """r03: the one-BVH structure's 6-wide node form (pt_device.h BvhNode6: quantize_node6 / trav_node6 / the range entries of the leaf
    queue in pt_bvh.h) through the host harness, which builds it with the two properties the GPU builder gives it — a node's internal
    children are consecutive records, its leaf children consecutive triangles — against the oracle: the hits and the radiance do not
    depend on the structure that was walked (the GPU side of this is every `-m gpu` parity test: the device build is 6-wide by default)."""
    monkeypatch.setenv("EMU_WIDE6", "1")
    if pairs:   # r4: leaf slots that hold two edge-sharing triangles (pt_device.h TriRec, host_scene.h pair_mesh_triangles)
        monkeypatch.setenv("EMU_PAIRS", "1")
    sc = {"cornell_sphere": scenes.cornell_sphere_scene, "field3": lambda: scenes.field_scene(3), "random5": lambda: scenes.random_scene(5),
          "random11": lambda: scenes.random_scene(11), "textured": lambda: scenes.textured_scene()}[name]()
    p = make_params(64, 36, 1, 6)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert o.trace_primary(1).tobytes() == e.trace_primary(1).tobytes()
    ro, ho = o.debug_sample(0)
    re_, he = e.debug_sample(0)
    nan = np.isnan(ro)
    assert np.array_equal(ho, he) and np.array_equal(nan, np.isnan(re_))
    assert np.array_equal(ro.view(np.uint32)[~nan], re_.view(np.uint32)[~nan])


@pytest.mark.parametrize("name", ["cornell", "random7", "textured"])
def test_threaded_host_render_of_the_shared_kernels_equals_the_oracle(name, monkeypatch):
    """bench.py's cpu_baseline (kind "same-kernels-host"): emu_render runs the product's stage functions over 16x16
    tiles on several std::thread workers and folds samples like k_accumulate.  The running mean it leaves must be the oracle's, bit for
    bit, whatever the thread count and however the samples are split over calls."""
    for k, v in (("EMU_MORTON", "1"), ("EMU_PLOC", "8"), ("EMU_WIDE6", "1")):   # the tree form bench.py asks for
        monkeypatch.setenv(k, v)
    sc = {"cornell": lambda: scenes.cornell_scene("bench"), "random7": lambda: scenes.random_scene(7), "textured": scenes.textured_scene}[name]()
    p = make_params(72, 40, 5, 5)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    want = o.render(0, 5)
    a = e.render(0, 2, threads=3)
    a = e.render(2, 3, acc=a, acc_n0=2, threads=5)
    nan = np.isnan(want)
    assert np.array_equal(nan, np.isnan(a)) and np.array_equal(want.view(np.uint32)[~nan], a.view(np.uint32)[~nan])
    assert np.array_equal(e.render(0, 5, threads=1).view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def test_leaf_slot_pairing_edge_cases_on_the_host(monkeypatch):
    """scenes.pairing_edge_cases_scene: every branch of the pairing (plain quad, fan, zero-area partner, degenerate first triangle, duplicates, one
    shared vertex, two materials in one slot, a mirrored instance) through the product's traversal compiled for the host, against the oracle."""
    for k, v in (("EMU_MORTON", "1"), ("EMU_PLOC", "8"), ("EMU_WIDE6", "1"), ("EMU_PAIRS", "1")):
        monkeypatch.setenv(k, v)
    sc = scenes.pairing_edge_cases_scene()
    p = make_params(96, 64, 2, 5)
    o, e = oracle_lib.OracleScene(sc, p), emu_lib.EmuScene(sc, p)
    assert o.trace_primary(0).tobytes() == e.trace_primary(0).tobytes()
    assert (o.trace_primary(0)["instance"] >= 0).mean() > 0.2
    for s in (0, 1):
        ro, ho = o.debug_sample(s)
        re_, he = e.debug_sample(s)
        nan = np.isnan(ro)
        assert np.array_equal(ho, he) and np.array_equal(nan, np.isnan(re_)) and np.array_equal(ro.view(np.uint32)[~nan], re_.view(np.uint32)[~nan])


def test_halton_fp32_division_boundaries_equal_oracle():
    """The strength-reduced radical inverse of pt_sampler.h at the multiples of every dimension's chunk (+-1), around 2^21 and at
    the top of the 32-bit range.  (An fp32 division of the small quotients was tried on top of it: bit-exact, but slower.)"""
    e = emu_lib.EmuScene(scenes.cornell_scene(), make_params(8, 8, 1, 2))
    L = oracle_lib.lib()
    for d in list(range(0, 64)) + list(range(64, 620, 7)) + [37, 53, 302, 303, 619]:  # (37..53: primes 163..251, two divisions)
        p = L.orc_prime(d)
        chunk = p
        while chunk * p < (1 << 22):   # pt_sampler.h make_halton_entry: the largest power below 2^22
            chunk *= p
        # (chunk - 1, chunk, chunk + 1: the quotient that is / is not its own last remainder — pt_sampler.h stops dividing at q < chunk)
        ks = [1, 2, 3, max(1, (1 << 21) // chunk - 1), max(1, (1 << 21) // chunk), (1 << 21) // chunk + 1, chunk - 1, chunk, chunk + 1,
              chunk * chunk - 1, chunk * chunk, chunk * chunk + 1, (1 << 32) // chunk - 1, (1 << 32) // chunk]
        # remainders whose digit splits sit on the fp32 rounding boundary: multiples of the prime (+-1) just below the chunk
        rems = [chunk - 1, chunk - p, chunk - p - 1, chunk - p + 1, (chunk // p) * p - 1, p * (p - 1), p * p - 1 if p * p <= chunk else chunk - 1]
        for r in rems:
            for q in (0, 1, (1 << 32) // chunk - 1):
                i = q * chunk + r
                if 0 <= i < (1 << 32):
                    assert e.halton(i, d) == L.orc_halton(i, d), (i, d)
        for k in ks:
            for i in (k * chunk - 1, k * chunk, k * chunk + 1):
                if 0 <= i < (1 << 32):
                    assert e.halton(i, d) == L.orc_halton(i, d), (i, d)
        for i in ((1 << 21) - 1, 1 << 21, (1 << 21) + 1, (1 << 32) - 1):
            assert e.halton(i, d) == L.orc_halton(i, d), (i, d)
