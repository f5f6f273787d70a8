"""Physical anchors for the integrator (GPU, statistical).  The reference ships no golden image and its Metal kernels cannot be run
here, so the oracle's `trace_path` is pinned only in pieces.  These tests add anchors that do NOT depend on the restatement:
properties any unbiased path tracer with an energy-conserving BSDF must have, checked on the HIP path at sample counts only the GPU
affords (the HIP path is bit-identical to the oracle on every small case of test_gpu_parity.py, so what holds for one holds for the other).

* the two integrators of the reference (kernel.metal:289-356 SIMPLE = BSDF sampling only; :505-670 MIS = + next-event estimation with
  the balance heuristic) are two estimators of the same integral: on scenes lit by (untextured) area lights their images must agree
  within Monte-Carlo error — a wrong light pdf, MIS weight, NEE gate or shadow-ray epsilon shows up as a bias;
* white furnace: a white object inside a constant environment of radiance 1 must look like the environment (radiance 1) — BSDF
  sampling weights f*cos/pdf, the energy-compensation tables, Russian roulette and the throughput bookkeeping all enter.

Tolerances are statistical and stated per test.  Measured deviations that are the REFERENCE's (kept as they are):
the environment's NEE pdf has no sin(theta) Jacobian, so the MIS integrator loses 1.5-5 % in a furnace; at roughness 1 the energy
tables over-compensate (+11 % for a white metal); an emission TEXTURE modulates the light a path hits but not the light NEE samples
(bsdf.metal:28-29 vs kernel.metal:428), so the integrators differ on emission-textured lights."""
import numpy as np
import pytest

from platinum_amd import abi, scenes
from platinum_amd.scenes import Camera, Material, Scene, Transform, cube, sphere

pytestmark = pytest.mark.gpu


def _render(r, sc, w, h, spp, bounces, integrator, flags=abi.FLAG_MULTISCATTER_GGX):
    r.selectKernel(integrator)
    r.startRender(sc, (w, h), spp, max_bounces=bounces, flags=flags, nonfinite_policy=abi.NONFINITE_ZERO)
    r.render(0)
    return r.readbackAccumulator().astype(np.float64)[..., :3]


@pytest.mark.parametrize("name,mean_tol", [("cornell", 0.005), ("cornell_sphere", 0.02)])
def test_mis_and_simple_integrators_estimate_the_same_image(gpu_renderer, name, mean_tol):
    """96x96 x 4096 spp x 8 bounces each (75 M paths per image).  Measured r03: image means differ by 0.09 % (Cornell box) and 0.9 %
    (with the rough glass sphere, whose caustic paths only BSDF sampling finds: fireflies in both images); medians of 12x12-pixel block means
    within 1.5 / 2.0 %.  A bias in the NEE estimator (a light pdf off by a constant, a missing cosine) would move the mean by tens of percent."""
    sc = scenes.cornell_scene() if name == "cornell" else scenes.cornell_sphere_scene()
    mis = _render(gpu_renderer, sc, 96, 96, 4096, 8, abi.INTEGRATOR_MIS)
    simple = _render(gpu_renderer, sc, 96, 96, 4096, 8, abi.INTEGRATOR_SIMPLE)
    assert abs(mis.mean() - simple.mean()) / simple.mean() < mean_tol
    bm = mis.reshape(8, 12, 8, 12, 3).mean(axis=(1, 3))
    bs = simple.reshape(8, 12, 8, 12, 3).mean(axis=(1, 3))
    assert np.median(np.abs(bm - bs) / np.maximum(bs, 1e-3)) < 0.04


def _panel_scene(rho, ior, Le, height, side, px):
    """A 20 x 20 floor slab (top face at y = 0) under a side x side panel light at `height`, both cut from the reference's cube primitive (its
    PLANE primitive carries tangent.w = 0 — primitives.cpp:16-22 — i.e. a shading frame without a bitangent: BSDF samples leave such a
    surface in one plane only; kept as it is in scenes.plane, avoided here).  The camera looks at the floor point (px, 0, 0) through a long lens."""
    sc = Scene(name="panel")
    q = sc.add_mesh(cube(1.0))
    sc.add_instance(q, Transform(translation=(0, -0.5, 0), scale=(20, 1, 20)), [Material(base_color=(rho, rho, rho, 1), roughness=1.0, ior=ior)])
    sc.add_instance(q, Transform(translation=(0, height + 0.005, 0), scale=(side, 0.01, side)),
                    [Material(base_color=(0, 0, 0, 1), emission=(1, 1, 1), emission_strength=Le)])
    sc.set_camera(Camera.with_focal_length(200.0), Transform(translation=(px + 4.0, 3.0, 6.0), target=(px, 0, 0), track=True))
    return sc


def _form_factor(px, height, side, n=2000):
    """integral over the panel of cos(theta) cos(theta') / d^2 dA seen from the floor point (px, 0, 0): midpoint rule, 4 M cells"""
    u = (np.arange(n) + 0.5) / n * side - side / 2
    X, Z = np.meshgrid(u, u)
    d2 = (X - px) ** 2 + Z ** 2 + height * height
    return float((height * height / (d2 * d2)).sum() * (side / n) ** 2)


@pytest.mark.parametrize("rho,ior,Le,height,side,px", [(0.5, 1.0, 5.0, 3.0, 2.0, 0.0), (0.8, 1.0, 2.0, 1.5, 3.0, 1.0), (0.5, 1.5, 5.0, 3.0, 2.0, 0.0)])
def test_diffuse_floor_under_a_panel_light_matches_the_analytic_irradiance(gpu_renderer, rho, ior, Le, height, side, px):
    """Absolute radiometry: with one bounce of light transport (max_bounces 2) a rough dielectric floor (ior 1: no specular lobe at all, i.e.
    Lambert; ior 1.5 at roughness 1: Lambert to within the energy tables) shows L = rho / pi * Le * form factor.  Pins the emission scale
    (emission x strength through the BT.709 -> working-space matrix), the light-selection pdf and area pdf of NEE, both cosines, the 1 / pi of the
    diffuse lobe, the shadow ray and the running mean — in BOTH integrators.  Measured r03: MIS 0.998 / 0.997 / 1.001 of the prediction,
    SIMPLE 0.988 / 0.997 / 1.001 (12 % of its samples find the light: noisier)."""
    pred = rho / np.pi * Le * _form_factor(px, height, side)
    sc = _panel_scene(rho, ior, Le, height, side, px)
    mis = _render(gpu_renderer, sc, 64, 64, 4096, 2, abi.INTEGRATOR_MIS)[28:36, 28:36].mean()
    simple = _render(gpu_renderer, sc, 64, 64, 4096, 2, abi.INTEGRATOR_SIMPLE)[28:36, 28:36].mean()
    assert abs(mis / pred - 1.0) < 0.01
    assert abs(simple / pred - 1.0) < 0.025


def _furnace(roughness, metallic, transmission=0.0, clearcoat=0.0):
    sc = Scene(name="furnace")
    sc.add_instance(sc.add_mesh(sphere(1.0, 48, 64)), Transform(),
                    [Material(base_color=(1, 1, 1, 1), roughness=roughness, metallic=metallic, transmission=transmission, ior=1.5, clearcoat=clearcoat)])
    sc.env_texture = sc.add_texture(np.ones((8, 16, 4), dtype=np.float32), abi.TEX_RGBA32F)
    sc.set_camera(Camera.with_focal_length(50.0), Transform(translation=(0, 0, 6), target=(0, 0, 0), track=True))
    return sc


@pytest.mark.parametrize("roughness,metallic,transmission,clearcoat,tol", [
    (0.0, 0.0, 0.0, 0.0, 0.004), (0.2, 0.0, 0.0, 0.0, 0.004), (0.5, 0.0, 0.0, 0.0, 0.004),   # opaque dielectric: specular + diffuse, compensated
    (0.0, 1.0, 0.0, 0.0, 0.001), (0.2, 1.0, 0.0, 0.0, 0.004), (0.5, 1.0, 0.0, 0.0, 0.02),    # white metal (0.5: the tables lose 1.1 %)
    (0.3, 0.0, 1.0, 0.0, 0.02),                                                               # rough glass (measured 0.9925)
    (0.3, 0.0, 0.0, 1.0, 0.004),                                                              # clearcoat over a dielectric
])
def test_white_furnace_simple_integrator(gpu_renderer, roughness, metallic, transmission, clearcoat, tol):
    """A white sphere in a constant environment of radiance 1, SIMPLE integrator (the environment is only ever found by BSDF sampling, so the
    reference's environment pdf does not enter): the centre of the sphere must show radiance 1.  64x64 x 1024 spp x 16 bounces; the 16x16-pixel
    centre averages 262 k paths (noise ~0.1 %)."""
    a = _render(gpu_renderer, _furnace(roughness, metallic, transmission, clearcoat), 64, 64, 1024, 16, abi.INTEGRATOR_SIMPLE)
    assert abs(a[24:40, 24:40].mean() - 1.0) < tol
    assert abs(a[:8, :8].mean() - 1.0) < 1e-6   # the corner sees the environment itself


def test_reference_quirks_show_up_where_expected(gpu_renderer):
    """The deviations the reference itself has (kept, not fixed), measured so that a change in them is noticed: (1) MIS in the furnace loses
    energy (the alias-table pdf `importance / 4 pi` has no sin(theta) Jacobian, kernel.metal:440-467); (2) roughness 1 over-compensates."""
    mis = _render(gpu_renderer, _furnace(0.5, 0.0), 64, 64, 1024, 16, abi.INTEGRATOR_MIS)[24:40, 24:40].mean()
    assert 0.92 < mis < 0.97          # measured 0.944
    rough_metal = _render(gpu_renderer, _furnace(1.0, 1.0), 64, 64, 1024, 16, abi.INTEGRATOR_SIMPLE)[24:40, 24:40].mean()
    assert 1.08 < rough_metal < 1.14  # measured 1.1125
