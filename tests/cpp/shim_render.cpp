// TEST PROGRAM (tests/): a C++ host driving libptamd.so the way the reference's frontend drives pt::renderer_pt::Renderer
// (frontend/windows/pt_viewport.cpp:539-548 startRender, frontend.cpp:207-210 render() once per frame, status() polled, :711 present,
// readback for export) — through include/ptamd_renderer.hpp, whose members carry the reference's names.
//   shim_render scene.json W H spp bounces out_prefix [gmon_buckets [devices]]     devices: e.g. 0,0 = a device group (two logical shards on GPU 0)
// loads the scene file with the library's own reader (pt_scene_load_json, the reference's scene.json + _data.bin format), renders
// one sample per render() call until Status_Done, and writes <out_prefix>.acc (W*H*4 float), <out_prefix>.rgba (W*H*4 bytes:
// readbackRenderTarget) and <out_prefix>.present (the same image copied back from the device pointer of presentRenderTarget).
// Exit codes: 0 ok, 2 usage, 3 the renderer could not be created (no GPU), 4 a call failed, 5 the progress protocol misbehaved.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "ptamd_renderer.hpp"
#include "ptamd_scene.h"

using ptamd::renderer_pt::Renderer;

static bool write_file(const std::string& path, const void* data, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(data, 1, bytes, f) == bytes;
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: shim_render scene.json W H spp bounces out_prefix [gmon_buckets [devices]]\n"); return 2; }
  const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), spp = (uint32_t)atoi(argv[4]), bounces = (uint32_t)atoi(argv[5]);
  const std::string out = argv[6];
  const uint32_t buckets = argc > 7 ? (uint32_t)atoi(argv[7]) : 0;

  std::vector<int> devices;
  if (argc > 8) for (const char* p = argv[8]; *p; p++) if (*p >= '0' && *p <= '9') devices.push_back(*p - '0');
  // never throws: a failure is printed and leaves the object blocked
  const std::unique_ptr<Renderer> holder = devices.empty() ? std::make_unique<Renderer>(0) : std::make_unique<Renderer>(devices);
  Renderer& renderer = *holder;
  if (!renderer.ok()) return renderer.status() == Renderer::Status_Blocked ? 3 : 5;

  pt_scene* scene = nullptr;
  if (pt_scene_load_json(argv[1], &scene) != PT_OK) { fprintf(stderr, "scene: %s\n", pt_last_error()); return 4; }
  uint64_t camera = 0;
  const pt_scene_snapshot* snap = nullptr;
  if (pt_scene_get_camera(scene, 0, &camera, nullptr, 0) != PT_OK || pt_scene_build_snapshot(scene, camera, &snap) != PT_OK) {
    fprintf(stderr, "scene: %s\n", pt_last_error());
    return 4;
  }

  if (renderer.status() != Renderer::Status_Blocked) return 5;          // nothing started yet
  renderer.selectKernel(uint32_t(Renderer::Integrators::MIS));
  renderer.setMaxBounces(bounces);
  const pt_colorspace bt2020 = {{0.708f, 0.292f}, {0.170f, 0.797f}, {0.131f, 0.046f}, {0.3127f, 0.3290f}};  // pt_viewport.hpp:95
  const int flags = PT_FLAG_MULTISCATTER_GGX | (buckets ? PT_FLAG_GMON : 0);
  renderer.startRender(*snap, {(float)W, (float)H}, spp, buckets, bt2020, flags);
  if (!renderer.ok()) return 4;
  pt_scene_destroy(scene);                                // the snapshot was only read during startRender

  // the frontend's loop: one render() per UI frame while the status says busy (frontend.cpp:207-210)
  uint32_t frames = 0;
  while (!(renderer.status() & Renderer::Status_Done)) {
    if (!(renderer.status() & Renderer::Status_Busy)) return 5;
    renderer.render();
    if (!renderer.ok()) return 4;
    if (++frames > spp) return 5;
  }
  // exactly one sample per call and device: a group of N devices advances N samples per render()
  const uint32_t members = devices.empty() ? 1u : (uint32_t)devices.size();
  const auto progress = renderer.renderProgress();
  if (progress.first != spp || progress.second != spp || frames != (spp + members - 1) / members) return 5;
  if (!(renderer.status() & Renderer::Status_Ready)) return 5;

  const std::vector<float> acc = renderer.readbackAccumulator();
  ptamd::renderer_pt::uint2 size{};
  const std::vector<uint8_t> rgba = renderer.readbackRenderTarget(&size);
  if (acc.size() != (size_t)W * H * 4 || rgba.size() != (size_t)W * H * 4 || size.x != W || size.y != H) return 4;

  // presentRenderTarget: the same image, left on the device; copy it back with the HIP runtime the library itself uses
  const void* dev = renderer.presentRenderTarget();
  std::vector<uint8_t> presented((size_t)W * H * 4);
  using memcpy_fn = int (*)(void*, const void*, size_t, int);
  using sync_fn = int (*)(void*);
  void* self = dlopen(nullptr, RTLD_NOW);
  auto hipMemcpy_ = (memcpy_fn)dlsym(self, "hipMemcpy");
  auto hipStreamSynchronize_ = (sync_fn)dlsym(self, "hipStreamSynchronize");
  if (!dev || !hipMemcpy_ || !hipStreamSynchronize_) return 4;
  if (hipStreamSynchronize_(renderer.presentStream()) != 0 || hipMemcpy_(presented.data(), dev, presented.size(), 2 /* DeviceToHost */) != 0) return 4;

  if (!write_file(out + ".acc", acc.data(), acc.size() * sizeof(float)) || !write_file(out + ".rgba", rgba.data(), rgba.size()) ||
      !write_file(out + ".present", presented.data(), presented.size())) return 4;
  printf("shim_render: %ux%u, %u spp in %u render() calls, %zu ms\n", W, H, spp, frames, renderer.renderTime());
  return 0;
}
