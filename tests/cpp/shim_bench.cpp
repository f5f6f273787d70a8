// MEASUREMENT PROGRAM (tests/): throughput of libptamd.so driven from a C++ host the way the reference's frontend drives
// pt::renderer_pt::Renderer — no Python, no torch in the process, so the HIP runtime in play is the system's (/opt/rocm), not the copy a
// PyTorch wheel bundles.
//   shim_bench scene.json W H bounces steps warmup [samples_per_step]
// One step = samples_per_step (default: the library's own batch for the image) render() calls of ONE sample each, which the library merges into
// batches (renderer_pt.cpp:131-153).  Prints one JSON line: Msamples/s = W * H * spp * bounces / t / 1e6 over `steps` steps,
// the runtime pt_get_runtime_info reports, and the library's counters.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "ptamd_renderer.hpp"
#include "ptamd_scene.h"

using ptamd::renderer_pt::Renderer;

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: shim_bench scene.json W H bounces steps warmup [samples_per_step]\n"); return 2; }
  const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), bounces = (uint32_t)atoi(argv[4]), steps = (uint32_t)atoi(argv[5]),
                 warmup = (uint32_t)atoi(argv[6]);
  uint32_t S = argc > 7 ? (uint32_t)atoi(argv[7]) : 0;
  Renderer renderer(0);
  if (!renderer.ok()) return 3;
  pt_scene* scene = nullptr;
  if (pt_scene_load_json(argv[1], &scene) != PT_OK) { fprintf(stderr, "scene: %s\n", pt_last_error()); return 4; }
  uint64_t camera = 0;
  const pt_scene_snapshot* snap = nullptr;
  if (pt_scene_get_camera(scene, 0, &camera, nullptr, 0) != PT_OK || pt_scene_build_snapshot(scene, camera, &snap) != PT_OK) {
    fprintf(stderr, "scene: %s\n", pt_last_error());
    return 4;
  }
  renderer.selectKernel(uint32_t(Renderer::Integrators::MIS));
  renderer.setMaxBounces(bounces);
  renderer.setNonfinitePolicy(PT_NONFINITE_ZERO);   // as bench.py
  const pt_colorspace bt2020 = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}, {0.3127f, 0.3290f}};
  pt_stats st{};
  if (S == 0) {  // the batch the library plans for this image on this device
    renderer.startRender(*snap, {(float)W, (float)H}, 1u << 16, 0, bt2020, PT_FLAG_MULTISCATTER_GGX);  // (nothing is rendered: the plan is made at start)
    if (!renderer.ok() || pt_get_stats(renderer.handle(), &st) != PT_OK) return 4;
    S = st.samples_in_flight;
  }
  auto run = [&](uint32_t nsteps) -> double {   // nsteps * S render() calls of one sample, then wait; seconds
    renderer.setSamplesInFlight(S);
    renderer.startRender(*snap, {(float)W, (float)H}, nsteps * S, 0, bt2020, PT_FLAG_MULTISCATTER_GGX);
    if (!renderer.ok()) return -1.0;
    renderer.wait();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t k = 0; k < nsteps * S; k++) renderer.render();
    renderer.wait();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  if (warmup && run(warmup) < 0) return 4;
  const double sec = run(steps);
  if (sec < 0 || pt_get_stats(renderer.handle(), &st) != PT_OK) return 4;
  pt_runtime_info ri{};
  pt_get_runtime_info(&ri);
  printf("{\"host\": \"C++ (include/ptamd_renderer.hpp), render() = one sample per call\", \"value\": %.2f, \"unit\": \"Msamples/s\", \"width\": %u, \"height\": %u, "
         "\"max_bounces\": %u, \"steps\": %u, \"spp_per_step\": %u, \"samples_in_flight\": %u, \"batches\": %u, \"ms_per_step\": %.3f, \"triangles\": %llu, "
         "\"closest_rays\": %llu, \"shadow_rays\": %llu, \"shaded_hits\": %llu, \"nonfinite_samples\": %llu, \"hip_runtime_path\": \"%s\", \"hip_runtime_version\": %d, "
         "\"hip_runtimes_mapped\": %u}\n",
         (double)W * H * steps * S * bounces / sec / 1e6, W, H, bounces, steps, S, st.samples_in_flight, st.batches, sec / steps * 1e3,
         (unsigned long long)st.triangles, (unsigned long long)st.closest_rays, (unsigned long long)st.shadow_rays, (unsigned long long)st.shaded_hits,
         (unsigned long long)st.nonfinite_samples, ri.hip_runtime_path, ri.hip_runtime_version, ri.hip_runtimes_mapped);
  pt_scene_destroy(scene);
  return 0;
}
