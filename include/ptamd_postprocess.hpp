/* ptamd_postprocess.hpp — the option types the reference's UI edits, under their own names, for the untouched frontend.
 *
 * `Renderer::postProcessOptions()` returns `std::vector<postprocess::PostProcessPass::Options>` — tagged unions of POINTERS into the passes'
 * own option structs (/root/reference/src/core/postprocessing.hpp:241-252) that `RenderViewport` walks with a switch on `options.type`
 * and edits in place (frontend/windows/pt_viewport.cpp:259-335) — and `Renderer::tonemapOptions()` a `TonemapOptions*` with nested
 * `agxOptions.look`, `khrOptions`, `flimOptions`, `postTonemap` (postprocessing.hpp:216-226, edited at pt_viewport.cpp:337-535, presets
 * assigned whole: `look = agx::looks::golden`, `options = flim::presets::silver`).  This header declares those shapes — member names, enumerator
 * order, defaults and presets are the reference's data — so that code compiles against ptamd::renderer_pt::Renderer unchanged;
 * ptamd_renderer.hpp flattens them into the C ABI's pt_post_options / pt_tonemap_options whenever an image is asked for.
 * No Metal types: the passes themselves (PostProcessPass::apply, postprocessing.hpp:254-315) are ONE HIP kernel inside the library.
 *
 * Vector members: the reference uses <simd/simd.h> float2 / float3 (16-byte float3).  A port that brings its own replacement defines
 * PTAMD_SIMD_TYPES before including this header and gets `using ::float2; using ::float3;`; otherwise the minimal aggregates below
 * (same size and alignment, `(float*)&v` reaches x, y, z as the UI's widgets do). */
#ifndef PTAMD_POSTPROCESS_HPP
#define PTAMD_POSTPROCESS_HPP

#include "ptamd.h"

namespace ptamd::postprocess {

#ifdef PTAMD_SIMD_TYPES
using ::float2;
using ::float3;
#else
struct alignas(8) float2 { float x, y; };
struct alignas(16) float3 { float x, y, z; };
#endif

namespace agx {
struct Look { float3 offset, slope, power; float saturation; };               // postprocessing.hpp:31-34
namespace looks {                                                                // :36-59
inline constexpr Look none{{0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}, 1.0f};
inline constexpr Look golden{{0.0f, 0.0f, 0.0f}, {1.0f, 0.9f, 0.5f}, {0.8f, 0.8f, 0.8f}, 0.8f};
inline constexpr Look punchy{{0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}, {1.35f, 1.35f, 1.35f}, 1.4f};
}  // namespace looks
struct Options { Look look = looks::none; };                                    // :61-63
}  // namespace agx

namespace khronos_pbr {
struct Options { float compressionStart = 0.8f; float desaturation = 0.15f; };  // :69-72
}

namespace flim {
struct Options {                                                                 // :78-105
  float preExposure; float3 preFormationFilter; float preFormationFilterStrength;
  float3 extendedGamutScale, extendedGamutRotation, extendedGamutMul;
  float sigmoidLog2Min, sigmoidLog2Max; float2 sigmoidToe, sigmoidShoulder;
  float negativeExposure, negativeDensity;
  float3 printBacklight; float printExposure, printDensity;
  float blackPoint; bool autoBlackPoint; float3 postFormationFilter; float postFormationFilterStrength;
  float midtoneSaturation;
};
namespace presets {                                                              // :107-163
inline constexpr Options flim{4.3f, {1.0f, 1.0f, 1.0f}, 0.0f, {1.05f, 1.12f, 1.045f}, {0.5f, 2.0f, 0.1f}, {1.0f, 1.0f, 1.0f}, -10.0f, 22.0f,
                              {0.440f, 0.280f}, {0.591f, 0.779f}, 6.0f, 5.0f, {1.0f, 1.0f, 1.0f}, 6.0f, 27.5f, 0.0f, true, {1.0f, 1.0f, 1.0f}, 0.0f, 1.02f};
inline constexpr Options silver{3.9f, {0.0f, 0.5f, 1.0f}, 0.05f, {1.05f, 1.12f, 1.045f}, {0.5f, 2.0f, 0.1f}, {1.0f, 1.0f, 1.06f}, -10.0f, 22.0f,
                                {0.440f, 0.280f}, {0.591f, 0.779f}, 4.7f, 7.0f, {0.9992f, 0.99f, 1.0f}, 4.7f, 30.0f, 0.5f, false, {1.0f, 1.0f, 0.0f}, 0.04f, 1.0f};
}  // namespace presets
}  // namespace flim

enum class Tonemapper { None, AgX, KhronosPBR, flim };                           // :169-174 (== PT_TONEMAP_*)

struct ExposureOptions { float exposure = 0.0f; };                               // :176-178
struct ToneCurveOptions { float k = 1.0f, blacks = 0.0f, shadows = 0.0f, highlights = 0.0f, whites = 0.0f; };            // :180-186 (k: "debug option", unused by the pass)
struct VignetteOptions { float amount = 0.0f, midpoint = 0.0f, feather = 50.0f, power = 20.0f, roundness = 100.0f; };    // :188-194
struct ChromaticAberrationOptions { float amount = 0.0f, greenShift = 70.0f; };  // :196-199
struct ContrastSaturationOptions { float contrast = 0.0f, saturation = 0.0f; };  // :201-204
struct LiftGammaGain {                                                           // :206-214
  float3 shadowColor{0.5f, 0.5f, 0.5f}, midtoneColor{0.5f, 0.5f, 0.5f}, highlightColor{0.5f, 0.5f, 0.5f};
  float shadowOffset = 0.0f, midtoneOffset = 0.0f, highlightOffset = 0.0f;
};
struct TonemapOptions {                                                          // :216-226
  Tonemapper tonemapper = Tonemapper::AgX;
  agx::Options agxOptions;
  khronos_pbr::Options khrOptions;
  flim::Options flimOptions = flim::presets::flim;
  LiftGammaGain postTonemap;
  float odt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // working -> display space: the reference's Renderer overwrites it before every tonemap pass
                                               // (renderer_pt.cpp:190-192); here the library derives it from outputColorspace() at the same moment
};

struct PostProcessPass {                                                         // :230-252 (the option half of the class)
  enum class Type { Exposure, ToneCurve, Vignette, ChromaticAberration, ContrastSaturation, Tonemap };
  struct Options {
    Type type = Type::Exposure;
    union {
      ExposureOptions* exposure = nullptr;
      ToneCurveOptions* toneCurve;
      VignetteOptions* vignette;
      ChromaticAberrationOptions* chromaticAberration;
      ContrastSaturationOptions* contrastSaturation;
      TonemapOptions* tonemap;
    };
  };
};
struct Tonemap { using Options = TonemapOptions; };                              // `postprocess::Tonemap::Options*` (renderer_pt.hpp:67)

// ---- flattening into the C ABI (what ptamd_renderer.hpp pushes before every image) -------------------------------------------------
inline void flatten(const ExposureOptions& e, const ChromaticAberrationOptions& ca, const ContrastSaturationOptions& cs, const ToneCurveOptions& tc,
                    const VignetteOptions& v, pt_post_options* o) {
  o->exposure = e.exposure;
  o->ca_amount = ca.amount; o->ca_green_shift = ca.greenShift;
  o->contrast = cs.contrast; o->saturation = cs.saturation;
  o->blacks = tc.blacks; o->shadows = tc.shadows; o->highlights = tc.highlights; o->whites = tc.whites;
  o->vig_amount = v.amount; o->vig_midpoint = v.midpoint; o->vig_feather = v.feather; o->vig_power = v.power; o->vig_roundness = v.roundness;
}
inline void put3(float* d, const float3& s) { d[0] = s.x; d[1] = s.y; d[2] = s.z; }
inline void flatten(const TonemapOptions& t, const pt_colorspace& outputSpace, pt_tonemap_options* o) {
  o->tonemapper = (uint32_t)t.tonemapper;
  const agx::Look& l = t.agxOptions.look;
  put3(o->agx_offset, l.offset); put3(o->agx_slope, l.slope); put3(o->agx_power, l.power); o->agx_saturation = l.saturation;
  o->khr_compression_start = t.khrOptions.compressionStart; o->khr_desaturation = t.khrOptions.desaturation;
  const flim::Options& f = t.flimOptions;
  o->flim_pre_exposure = f.preExposure; put3(o->flim_pre_formation_filter, f.preFormationFilter);
  o->flim_pre_formation_filter_strength = f.preFormationFilterStrength;
  put3(o->flim_extended_gamut_scale, f.extendedGamutScale); put3(o->flim_extended_gamut_rotation, f.extendedGamutRotation);
  put3(o->flim_extended_gamut_mul, f.extendedGamutMul);
  o->flim_sigmoid_log2_min = f.sigmoidLog2Min; o->flim_sigmoid_log2_max = f.sigmoidLog2Max;
  o->flim_sigmoid_toe[0] = f.sigmoidToe.x; o->flim_sigmoid_toe[1] = f.sigmoidToe.y;
  o->flim_sigmoid_shoulder[0] = f.sigmoidShoulder.x; o->flim_sigmoid_shoulder[1] = f.sigmoidShoulder.y;
  o->flim_negative_exposure = f.negativeExposure; o->flim_negative_density = f.negativeDensity;
  put3(o->flim_print_backlight, f.printBacklight); o->flim_print_exposure = f.printExposure; o->flim_print_density = f.printDensity;
  o->flim_black_point = f.blackPoint; o->flim_auto_black_point = f.autoBlackPoint ? 1u : 0u;
  put3(o->flim_post_formation_filter, f.postFormationFilter); o->flim_post_formation_filter_strength = f.postFormationFilterStrength;
  o->flim_midtone_saturation = f.midtoneSaturation;
  const LiftGammaGain& g = t.postTonemap;
  put3(o->shadow_color, g.shadowColor); put3(o->midtone_color, g.midtoneColor); put3(o->highlight_color, g.highlightColor);
  o->shadow_offset = g.shadowOffset; o->midtone_offset = g.midtoneOffset; o->highlight_offset = g.highlightOffset;
  o->output_space = outputSpace;
}

}  // namespace ptamd::postprocess

#endif
