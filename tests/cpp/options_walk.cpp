// TEST PROGRAM (tests/): the reference UI's way of editing the renderer's options, restated against include/ptamd_renderer.hpp.
//
// RenderViewport::renderPostprocessSettings (frontend/windows/pt_viewport.cpp:259-535) iterates `m_renderer->postProcessOptions()`, switches
// on `options.type`, dereferences the matching union member and hands the address of each field to a widget; then it takes
// `*m_renderer->tonemapOptions()`, selects `tonemapOptions.tonemapper`, edits `agxOptions.look` / `khrOptions` / `flimOptions` in place
// (assigning whole presets: `look = agx::looks::golden`, `options = flim::presets::silver`) and the lift / gamma / gain of `postTonemap`.
// Here a `Widgets` object plays the widgets: every "drag" writes the next value of a script into the float it is given.  Four frames with
// four scripts; after each the image is read back (readbackRenderTarget) and written to <out_prefix>_<frame>.rgba.  tests/test_cpp_shim.py
// sets the same values in the flat C structs for the oracle and the Python host and compares bytes.
//   options_walk scene.json W H spp bounces out_prefix
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "ptamd_renderer.hpp"
#include "ptamd_scene.h"

namespace postprocess = ptamd::postprocess;
using ptamd::renderer_pt::Renderer;

struct Widgets {  // stands in for widgets::dragFloat / dragVec3 / color / ImGui::Checkbox: "the user" types the script's next value
  std::vector<float> script;
  size_t next = 0;
  float value() { const float v = script[next % script.size()]; next++; return v; }
  void dragFloat(const char*, float* v) { *v = value(); }
  void dragVec2(const char*, float* v) { v[0] = value(); v[1] = value(); }
  void dragVec3(const char*, float* v) { v[0] = value(); v[1] = value(); v[2] = value(); }
  void color(const char* l, float* v) { dragVec3(l, v); }
  void checkbox(const char*, bool* v) { *v = value() > 0.5f; }
};

// pt_viewport.cpp:259-335, the same traversal
static void editPostProcess(Renderer& renderer, Widgets& widgets) {
  for (const auto& options : renderer.postProcessOptions()) {
    switch (options.type) {
      case postprocess::PostProcessPass::Type::Exposure: {
        auto& exposureOptions = *options.exposure;
        widgets.dragFloat("Exposure", &exposureOptions.exposure);
        break;
      }
      case postprocess::PostProcessPass::Type::ContrastSaturation: {
        auto& csOptions = *options.contrastSaturation;
        widgets.dragFloat("Contrast", &csOptions.contrast);
        widgets.dragFloat("Saturation", &csOptions.saturation);
        break;
      }
      case postprocess::PostProcessPass::Type::ToneCurve: {
        auto& toneCurveOptions = *options.toneCurve;
        widgets.dragFloat("Blacks", &toneCurveOptions.blacks);
        widgets.dragFloat("Shadows", &toneCurveOptions.shadows);
        widgets.dragFloat("Highlights", &toneCurveOptions.highlights);
        widgets.dragFloat("Whites", &toneCurveOptions.whites);
        break;
      }
      case postprocess::PostProcessPass::Type::Vignette: {
        auto& vignetteOptions = *options.vignette;
        widgets.dragFloat("Amount", &vignetteOptions.amount);
        widgets.dragFloat("Midpoint", &vignetteOptions.midpoint);
        widgets.dragFloat("Feather", &vignetteOptions.feather);
        widgets.dragFloat("Power", &vignetteOptions.power);
        widgets.dragFloat("Roundness", &vignetteOptions.roundness);
        break;
      }
      case postprocess::PostProcessPass::Type::ChromaticAberration: {
        auto& caOptions = *options.chromaticAberration;
        widgets.dragFloat("Amount", &caOptions.amount);
        widgets.dragFloat("Green Shift", &caOptions.greenShift);
        break;
      }
      default: break;
    }
  }
}

// pt_viewport.cpp:337-535: tonemapper select, per-tonemapper options (preset buttons first, then the fields), final grading
static void editTonemap(Renderer& renderer, Widgets& widgets, postprocess::Tonemapper pick, int preset) {
  auto& tonemapOptions = *renderer.tonemapOptions();
  tonemapOptions.tonemapper = pick;
  switch (tonemapOptions.tonemapper) {
    case postprocess::Tonemapper::AgX: {
      auto& look = tonemapOptions.agxOptions.look;
      if (preset == 1) look = postprocess::agx::looks::golden;
      else if (preset == 2) look = postprocess::agx::looks::punchy;
      else if (preset == 0) look = postprocess::agx::looks::none;
      else {
        widgets.dragVec3("Offset", (float*)&look.offset);
        widgets.dragVec3("Slope", (float*)&look.slope);
        widgets.dragVec3("Power", (float*)&look.power);
        widgets.dragFloat("Saturation", &look.saturation);
      }
      break;
    }
    case postprocess::Tonemapper::KhronosPBR: {
      auto& options = tonemapOptions.khrOptions;
      widgets.dragFloat("Threshold", &options.compressionStart);
      widgets.dragFloat("Desaturation", &options.desaturation);
      break;
    }
    case postprocess::Tonemapper::flim: {
      auto& options = tonemapOptions.flimOptions;
      if (preset == 1) options = postprocess::flim::presets::silver; else options = postprocess::flim::presets::flim;
      if (preset == 3) {
        widgets.dragFloat("Pre-exposure", &options.preExposure);
        widgets.dragVec2("Toe", (float*)&options.sigmoidToe);
        widgets.color("Pre filter", (float*)&options.preFormationFilter);
        widgets.dragFloat("Pre strength", &options.preFormationFilterStrength);
        widgets.checkbox("Auto black point", &options.autoBlackPoint);
        widgets.dragFloat("Black point", &options.blackPoint);
      }
      break;
    }
    default: break;
  }
  widgets.color("Shadows", (float*)&tonemapOptions.postTonemap.shadowColor);
  widgets.color("Midtones", (float*)&tonemapOptions.postTonemap.midtoneColor);
  widgets.color("Highlights", (float*)&tonemapOptions.postTonemap.highlightColor);
  widgets.dragFloat("Shadows", &tonemapOptions.postTonemap.shadowOffset);
  widgets.dragFloat("Midtones", &tonemapOptions.postTonemap.midtoneOffset);
  widgets.dragFloat("Highlights", &tonemapOptions.postTonemap.highlightOffset);
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: options_walk scene.json W H spp bounces out_prefix\n"); return 2; }
  const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), spp = (uint32_t)atoi(argv[4]), bounces = (uint32_t)atoi(argv[5]);
  const std::string out = argv[6];
  Renderer renderer(0);
  if (!renderer.ok()) return 3;
  pt_scene* scene = nullptr;
  uint64_t camera = 0;
  const pt_scene_snapshot* snap = nullptr;
  if (pt_scene_load_json(argv[1], &scene) != PT_OK || pt_scene_get_camera(scene, 0, &camera, nullptr, 0) != PT_OK ||
      pt_scene_build_snapshot(scene, camera, &snap) != PT_OK) { fprintf(stderr, "scene: %s\n", pt_last_error()); return 4; }
  renderer.setMaxBounces(bounces);
  const pt_colorspace bt2020 = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}, {0.3127f, 0.3290f}};
  renderer.startRender(*snap, {(float)W, (float)H}, spp, 0, bt2020, PT_FLAG_MULTISCATTER_GGX);
  pt_scene_destroy(scene);
  while (renderer.ok() && !(renderer.status() & Renderer::Status_Done)) renderer.render();
  if (!renderer.ok()) return 4;

  // the defaults the UI shows before anything is touched (postprocessing.hpp:176-226)
  {
    const auto opts = renderer.postProcessOptions();
    if (opts.size() != 5 || opts[0].type != postprocess::PostProcessPass::Type::Exposure || opts[4].vignette->feather != 50.0f ||
        opts[1].chromaticAberration->greenShift != 70.0f || renderer.tonemapOptions()->tonemapper != postprocess::Tonemapper::AgX ||
        renderer.tonemapOptions()->flimOptions.printDensity != 27.5f || renderer.tonemapOptions()->khrOptions.compressionStart != 0.8f) return 5;
  }
  struct Frame { postprocess::Tonemapper pick; int preset; std::vector<float> post, tone; };
  const Frame frames[4] = {
      {postprocess::Tonemapper::AgX, 1, {0.5f, 10.0f, 60.0f, 12.0f, -8.0f, 5.0f, -10.0f, 15.0f, -5.0f, 0.8f, 10.0f, 40.0f, 25.0f, 90.0f},
       {0.52f, 0.5f, 0.48f, 0.5f, 0.51f, 0.5f, 0.49f, 0.5f, 0.53f, 3.0f, -2.0f, 4.0f}},
      {postprocess::Tonemapper::KhronosPBR, 0, {-0.4f, 0.0f, 70.0f, -15.0f, 20.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f, -20.0f, 60.0f, 30.0f, 70.0f},
       {0.7f, 0.2f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f}},
      {postprocess::Tonemapper::flim, 1, {0.0f, 25.0f, 50.0f, 0.0f, 0.0f, 10.0f, 10.0f, -10.0f, -10.0f, 0.0f, 0.0f, 50.0f, 20.0f, 100.0f},
       {0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f, 0.0f}},
      {postprocess::Tonemapper::flim, 3, {1.0f, -30.0f, 80.0f, 30.0f, -30.0f, -20.0f, 20.0f, 20.0f, -20.0f, 1.5f, 30.0f, 20.0f, 10.0f, 50.0f},
       {4.0f, 0.4f, 0.3f, 0.9f, 0.8f, 0.7f, 0.1f, 0.0f, 0.02f, 0.45f, 0.5f, 0.55f, 0.5f, 0.5f, 0.5f, 0.55f, 0.5f, 0.45f, -3.0f, 2.0f, 1.0f}},
  };
  for (int f = 0; f < 4; f++) {
    Widgets post{frames[f].post}, tone{frames[f].tone};
    editPostProcess(renderer, post);
    editTonemap(renderer, tone, frames[f].pick, frames[f].preset);
    ptamd::renderer_pt::uint2 size{};
    const std::vector<uint8_t> rgba = renderer.readbackRenderTarget(&size);
    if (rgba.size() != (size_t)W * H * 4) return 4;
    const std::string path = out + "_" + std::to_string(f) + ".rgba";
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp || fwrite(rgba.data(), 1, rgba.size(), fp) != rgba.size()) return 4;
    fclose(fp);
  }
  printf("options_walk: 4 frames\n");
  return 0;
}
