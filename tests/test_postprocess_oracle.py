"""CPU: the oracle's restatement of the post-process chain + tonemap — closed-form checks."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from platinum_amd import abi, scenes
from platinum_amd.renderer import make_params


def _defaults():
    lib = abi.load_library()
    po, to = abi.PostOptions(), abi.TonemapOptions()
    lib.pt_default_post_options(C.byref(po))
    lib.pt_default_tonemap_options(C.byref(to))
    return po, to


def _srgb(c):
    c = np.asarray(c, dtype=np.float64)
    return np.where(c < 0.0031308, 12.92 * c, 1.055 * np.power(np.maximum(c, 1e-30), 1 / 2.4) - 0.055)


@pytest.fixture(scope="module")
def osc():
    return oracle_lib.OracleScene(scenes.cornell_scene(), make_params(16, 8, 1, 2))


def test_defaults_match_reference_structs():
    po, to = _defaults()
    assert (po.ca_green_shift, po.vig_feather, po.vig_power, po.vig_roundness) == (70.0, 50.0, 20.0, 100.0)   # postprocessing.hpp:181-193
    assert to.tonemapper == abi.TONEMAP_AGX and list(to.agx_slope) == [1, 1, 1] and to.agx_saturation == 1.0
    assert (to.khr_compression_start, to.khr_desaturation) == (pytest.approx(0.8), pytest.approx(0.15))
    assert to.flim_pre_exposure == pytest.approx(4.3) and to.flim_print_density == pytest.approx(27.5) and to.flim_auto_black_point == 1
    assert list(to.output_space.r) == [pytest.approx(0.680), pytest.approx(0.320)]                             # Display P3


def test_no_tonemap_is_odt_then_srgb(osc):
    po, to = _defaults()
    to.tonemapper = abi.TONEMAP_NONE
    to.output_space = scenes.colorspace(scenes.BT2020)   # odt = identity (working space is BT2020)
    acc = np.zeros((8, 16, 4), np.float32)
    acc[..., 0] = np.linspace(0, 1, 16)[None, :]
    acc[..., 1] = 0.25
    acc[..., 2] = np.linspace(0, 0.5, 8)[:, None]
    acc[..., 3] = 1
    out, fl = osc.postprocess(acc, po, to, want_float=True)
    np.testing.assert_allclose(fl, _srgb(acc[..., :3]), atol=2e-5)
    assert (out[..., 3] == 255).all()
    np.testing.assert_array_equal(out[..., :3], np.floor(np.clip(fl, 0, 1) * 255 + 0.5).astype(np.uint8))


def test_exposure_and_khronos_linear_segment(osc):
    po, to = _defaults()
    to.tonemapper = abi.TONEMAP_KHRONOS_PBR
    to.output_space = scenes.colorspace(scenes.BT2020)
    po.exposure = 1.0                                     # x2
    acc = np.full((8, 16, 4), 0.1, np.float32)
    _, fl = osc.postprocess(acc, po, to, want_float=True)
    x = 0.2                                               # min channel after exposure
    offset = 0.04                                         # x >= 0.08 (postprocess.metal:160)
    np.testing.assert_allclose(fl, np.broadcast_to(_srgb(np.full(3, x - offset)), fl.shape), atol=2e-4)   # peak < compressionStart: unchanged


def test_agx_and_flim_are_monotonic_and_bounded(osc):
    po, to = _defaults()
    ramp = np.zeros((8, 16, 4), np.float32)
    ramp[..., :3] = (2.0 ** np.linspace(-8, 4, 16))[None, :, None]
    ramp[..., 3] = 1
    for tm in (abi.TONEMAP_AGX, abi.TONEMAP_FLIM):
        to.tonemapper = tm
        out, fl = osc.postprocess(ramp, po, to, want_float=True)
        assert np.isfinite(fl).all() and fl.min() >= 0 and fl.max() <= 1.0 + 1e-5
        row = fl[0, :, 1]
        assert (np.diff(row) >= -1e-4).all() and row[-1] > row[0] + 0.5


def test_vignette_and_chromatic_aberration_are_spatial(osc):
    po, to = _defaults()
    to.tonemapper = abi.TONEMAP_NONE
    to.output_space = scenes.colorspace(scenes.BT2020)   # odt = identity, so the channels stay separate
    acc = np.zeros((8, 16, 4), np.float32)
    acc[..., :3] = 0.5
    acc[..., 0] = np.linspace(0.1, 0.9, 16)[None, :]       # a horizontal ramp in red
    acc[..., 3] = 1
    base, fb = osc.postprocess(acc, po, to, want_float=True)
    po.vig_amount = -2.0
    _, fv = osc.postprocess(acc, po, to, want_float=True)
    assert fv[0, 0, 1] < fb[0, 0, 1] - 0.05 and abs(fv[4, 8, 1] - fb[4, 8, 1]) < 0.02    # corners darken, centre stays
    po.vig_amount = 0.0
    po.ca_amount = 100.0
    _, fc = osc.postprocess(acc, po, to, want_float=True)
    # red is sampled further out, blue further in (postprocess.metal:541-547); the flat green channel cannot change
    assert (fc[:, 12:15, 0] > fb[:, 12:15, 0]).all() and (fc[:, 1:4, 0] < fb[:, 1:4, 0]).all()   # border texels clamp
    assert np.allclose(fc[..., 1], fb[..., 1], atol=1e-6)
