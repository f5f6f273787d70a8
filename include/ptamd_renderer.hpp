/* ptamd_renderer.hpp — the C++ face of libptamd.so: a header-only class with the public members of the reference's
 * pt::renderer_pt::Renderer (/root/reference/src/renderer_pt/renderer_pt.hpp:14-73), implemented over the C ABI of
 * ptamd.h.  It is what a Linux/ROCm build of the reference's frontend would compile instead of renderer_pt.cpp
 * (INTEGRATION.md shows the Store-walking half of that file); here it is also the C++ host side the parity tests drive
 * (tests/cpp/shim_render.cpp).
 *
 *   reference member (renderer_pt.hpp)                          here
 *   Renderer(MTL::Device*, MTL::CommandQueue*, Store&) noexcept  Renderer(int device) / Renderer(std::vector<int> devices) noexcept
 *   ~Renderer()                                            :34   ~Renderer()
 *   void render()                                          :36   void render()                 one sample per call (merged into batches by the
 *                                                                                               library while the GPU is busy), returns at once
 *   void startRender(cameraNodeId, viewportSize, sampleCount,
 *                    gmonBuckets, workingSpace, flags = 0) :38   void startRender(scene, viewportSize, sampleCount, gmonBuckets,
 *                                                                                 workingSpace, flags = 0)
 *        (the Store& of the constructor and the camera node become the flat pt_scene_snapshot, which names its camera)
 *   selectedKernel() / selectKernel(uint32_t)              :47   the same; enum class Integrators { Simple, MIS }   :16-19
 *   const MTL::Texture* presentRenderTarget() const        :55   const void* presentRenderTarget() const   RGBA8 in device memory
 *   NS::SharedPtr<MTL::Buffer> readbackRenderTarget(uint2*) :57  std::vector<uint8_t> readbackRenderTarget(uint2*) const   blocks
 *   int status() const                                     :59   the same bits: Status_Blocked/Ready/Busy/Done      :21-26
 *   std::pair<size_t, size_t> renderProgress() const       :61   the same
 *   size_t renderTime() const                              :63   the same (milliseconds)
 *   std::vector<postprocess::PostProcessPass::Options>
 *     postProcessOptions()                              :65      the same: one tagged pointer per pass (Exposure, ChromaticAberration,
 *                                                                ContrastSaturation, ToneCurve, Vignette — the order of renderer_pt.cpp:343-352)
 *                                                                into option structs this object owns (ptamd_postprocess.hpp)
 *   postprocess::Tonemap::Options* tonemapOptions()     :67      the same: TonemapOptions with agxOptions.look / khrOptions / flimOptions / postTonemap
 *   shaders_pt::GmonOptions& gmonOptions()              :71      pt_gmon_options& (the one member, `cap`)
 *   color::Colorspace& outputColorspace()               :73      pt_colorspace& (the four chromaticity pairs a color::Colorspace is built from)
 *        the caller edits them in place; they are flattened and handed to the library whenever an image is asked for (the reference's
 *        passes read theirs when they are encoded)
 * Error behaviour as the reference's: nothing throws; a failing call prints "renderer_pt: <message>" to stderr (the
 * reference prints and asserts, renderer_pt.cpp:402, 1044) and leaves the object in Status_Blocked; lastError() keeps the text.
 * Threading as the reference's: one caller thread per Renderer.
 */
#ifndef PTAMD_RENDERER_HPP
#define PTAMD_RENDERER_HPP

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "ptamd.h"
#include "ptamd_postprocess.hpp"

namespace ptamd::renderer_pt {

struct float2 { float x, y; };
struct uint2 { uint32_t x, y; };

class Renderer {
public:
  enum class Integrators { Simple = 0, MIS };  // renderer_pt.hpp:16-19
  enum Status {                                // renderer_pt.hpp:21-26
    Status_Blocked = 0,
    Status_Ready = 1 << 0,
    Status_Busy = 1 << 2,
    Status_Done = 1 << 3,
  };

  // `device`: HIP ordinal (the reference takes the window's MTL::Device).  lutPath: the GGX energy tables
  // (loadGgxLutTextures, renderer_pt.cpp:385-446); nullptr = $PTAMD_LUT_PATH.
  explicit Renderer(int device = 0, const char* lutPath = nullptr) noexcept { create(&device, 0, device, lutPath); }
  // A device group: every render is shared by these devices (sample ranges, one RCCL reduce; ptamd.h pt_create_info).
  explicit Renderer(const std::vector<int>& devices, const char* lutPath = nullptr) noexcept {
    create(devices.data(), (uint32_t)devices.size(), devices.empty() ? 0 : devices[0], lutPath);
  }
  Renderer(const Renderer&) = delete;
  Renderer& operator=(const Renderer&) = delete;
  ~Renderer() { if (m_pt) pt_destroy(m_pt); }

  // renderer_pt.cpp:113-197 steady state: one more sample of every pixel is accepted (on EVERY device of a group: N samples);
  // does not wait.  Calls that arrive while the GPU is busy are merged into batches by the library (ptamd.h pt_render_step), so
  // a tight `while (status() & Status_Busy) render();` loop runs at the full batch rate.  A frontend that calls render() once per
  // vsync is limited to refresh-rate samples per second by that loop itself (the reference's structural ceiling, BASELINE.md §1:
  // 60 spp/s where an MI355X traces ~950 on C3); setSamplesPerRender(n) lets such a loop accept n samples per call.
  void render() {
    if (!m_pt || !m_started) return;
    check(pt_render_step(m_pt, m_samplesPerRender));
  }

  // renderer_pt.cpp:199-217 + the rebuild half of the first render() (:72-111): the scene is copied to the device, light table,
  // constants and acceleration structure are built; progress restarts at 0.  `scene` is only read during the call.
  void startRender(const pt_scene_snapshot& scene, float2 viewportSize, uint32_t sampleCount, uint32_t gmonBuckets,
                   const pt_colorspace& workingSpace, int flags = 0) {
    m_started = false;
    if (!m_pt) return;
    pt_render_params p{};
    p.width = (uint32_t)viewportSize.x;
    p.height = (uint32_t)viewportSize.y;
    p.spp = sampleCount;
    p.gmon_buckets = gmonBuckets;
    p.flags = flags;
    p.integrator = m_selectedPipeline;
    p.working_space = workingSpace;
    p.max_bounces = m_maxBounces;
    p.first_sample = m_firstSample;
    p.samples_in_flight = m_samplesInFlight;
    p.nonfinite_policy = m_nonfinitePolicy;
    p.accel_structure = m_accelStructure;
    if (!check(pt_start_render(m_pt, &scene, &p))) return;
    m_size = uint2{p.width, p.height};
    m_started = true;
  }

  [[nodiscard]] constexpr uint32_t selectedKernel() const { return m_selectedPipeline; }
  constexpr void selectKernel(uint32_t kernel) { m_selectedPipeline = kernel; }

  // The post-processed RGBA8 image in DEVICE memory (W*H*4 bytes; valid until the next startRender); the work is enqueued on
  // presentStream() (a hipStream_t) — order a blit after it.  nullptr on failure.
  [[nodiscard]] const void* presentRenderTarget() const {
    if (!m_pt || !m_started || !pushOptions()) return nullptr;
    void* img = nullptr;
    if (!check(pt_present_render_target(m_pt, &img, &m_presentStream))) return nullptr;
    return img;
  }
  [[nodiscard]] void* presentStream() const { return m_presentStream; }

  // renderer_pt.cpp:1039-1059: blocks until the enqueued samples are done, returns W*H*4 bytes (row-major, top-left origin).
  [[nodiscard]] std::vector<uint8_t> readbackRenderTarget(uint2* size) const {
    std::vector<uint8_t> out;
    if (size) *size = m_size;
    if (!m_pt || !m_started || !pushOptions()) return out;
    out.resize((size_t)m_size.x * m_size.y * 4);
    if (!check(pt_read_render_target(m_pt, out.data()))) out.clear();
    return out;
  }

  [[nodiscard]] int status() const { return m_pt && m_started ? pt_status(m_pt) : (int)Status_Blocked; }  // renderer_pt.cpp:1023-1031
  [[nodiscard]] std::pair<size_t, size_t> renderProgress() const {                                        // :1033-1035
    uint64_t done = 0, total = 0;
    if (m_pt && m_started) pt_progress(m_pt, &done, &total);
    return {(size_t)done, (size_t)total};
  }
  [[nodiscard]] size_t renderTime() const { return m_pt && m_started ? (size_t)pt_render_time_ms(m_pt) : 0; }  // :1037

  // Option structs the UI edits every frame (renderer_pt.hpp:65-73; types: ptamd_postprocess.hpp = core/postprocessing.hpp's shapes).
  // One entry per post-process pass, in the order the reference runs them (renderer_pt.cpp:343-352); the UI switches on `type` and edits
  // through the pointer (pt_viewport.cpp:259-335).
  [[nodiscard]] std::vector<postprocess::PostProcessPass::Options> postProcessOptions() {
    using P = postprocess::PostProcessPass;
    std::vector<P::Options> options(5);
    options[0].type = P::Type::Exposure; options[0].exposure = &m_exposure;
    options[1].type = P::Type::ChromaticAberration; options[1].chromaticAberration = &m_chromaticAberration;
    options[2].type = P::Type::ContrastSaturation; options[2].contrastSaturation = &m_contrastSaturation;
    options[3].type = P::Type::ToneCurve; options[3].toneCurve = &m_toneCurve;
    options[4].type = P::Type::Vignette; options[4].vignette = &m_vignette;
    return options;
  }
  [[nodiscard]] constexpr postprocess::Tonemap::Options* tonemapOptions() { return &m_tonemap; }
  [[nodiscard]] constexpr pt_gmon_options& gmonOptions() { return m_gmonOptions; }
  pt_colorspace& outputColorspace() { return m_outputSpace; }

  // ---- what the reference fixes at compile time or does not have (ptamd.h "NEW") ----
  void setMaxBounces(uint32_t b) { m_maxBounces = b; }              // kernel.metal:5 MAX_BOUNCES = 50 (the default here too)
  void setFirstSample(uint32_t s) { m_firstSample = s; }            // frameIdx of the first sample: sample-range shards
  void setSamplesInFlight(uint32_t s) { m_samplesInFlight = s; }    // 0 = automatic
  void setSamplesPerRender(uint32_t n) { m_samplesPerRender = n ? n : 1; }  // samples one render() call accepts; 1 = the reference
  void setNonfinitePolicy(uint32_t p) { m_nonfinitePolicy = p; }    // PT_NONFINITE_*
  void setAccelStructure(uint32_t a) { m_accelStructure = a; }      // PT_ACCEL_*
  // The float accumulator (renderer_pt.cpp:812-821): W*H RGBA32F running mean, alpha 1 — the parity surface.  Blocks.
  [[nodiscard]] std::vector<float> readbackAccumulator() const {
    std::vector<float> out;
    if (!m_pt || !m_started || !check(pt_set_gmon_options(m_pt, &m_gmonOptions))) return out;  // (the resolve reads GmonOptions.cap)
    out.resize((size_t)m_size.x * m_size.y * 4);
    if (!check(pt_read_accumulator(m_pt, out.data()))) out.clear();
    return out;
  }
  void wait() const { if (m_pt && m_started) check(pt_wait(m_pt)); }
  [[nodiscard]] bool ok() const { return m_pt != nullptr && m_lastError.empty(); }
  [[nodiscard]] const std::string& lastError() const { return m_lastError; }
  [[nodiscard]] pt_renderer* handle() const { return m_pt; }

private:
  void create(const int* devices, uint32_t count, int first, const char* lutPath) noexcept {
    pt_tonemap_options defaults;
    pt_default_tonemap_options(&defaults);
    m_outputSpace = defaults.output_space;   // Display P3 (renderer_pt.hpp:182)
    std::vector<int32_t> ord(devices, devices + count);
    pt_create_info ci{};
    ci.abi_version = PT_ABI_VERSION;
    ci.device_ordinal = first;
    ci.lut_path = lutPath;
    ci.device_ordinals = count ? ord.data() : nullptr;
    ci.device_count = count;
    if (!check(pt_create(&ci, &m_pt))) m_pt = nullptr;
  }
  bool check(int rc) const {
    if (rc == PT_OK) { m_lastError.clear(); return true; }
    m_lastError = pt_last_error();
    std::fprintf(stderr, "renderer_pt: %s\n", m_lastError.c_str());
    return false;
  }
  // the option structs are pushed when an image is asked for (the reference's passes read them when they are encoded)
  bool pushOptions() const {
    pt_post_options post;
    pt_tonemap_options tonemap;
    postprocess::flatten(m_exposure, m_chromaticAberration, m_contrastSaturation, m_toneCurve, m_vignette, &post);
    postprocess::flatten(m_tonemap, m_outputSpace, &tonemap);
    return check(pt_set_gmon_options(m_pt, &m_gmonOptions)) && check(pt_set_post_options(m_pt, &post)) && check(pt_set_tonemap_options(m_pt, &tonemap));
  }

  pt_renderer* m_pt = nullptr;
  bool m_started = false;
  uint2 m_size{1, 1};
  uint32_t m_selectedPipeline = uint32_t(Integrators::MIS);  // renderer_pt.hpp:98
  uint32_t m_maxBounces = 50, m_firstSample = 0, m_samplesInFlight = 0, m_nonfinitePolicy = 0, m_accelStructure = PT_ACCEL_AUTO;
  uint32_t m_samplesPerRender = 1;
  postprocess::ExposureOptions m_exposure;                        // the passes' option structs (BasicPostProcessPass::m_options, postprocessing.hpp:304)
  postprocess::ChromaticAberrationOptions m_chromaticAberration;
  postprocess::ContrastSaturationOptions m_contrastSaturation;
  postprocess::ToneCurveOptions m_toneCurve;
  postprocess::VignetteOptions m_vignette;
  postprocess::TonemapOptions m_tonemap;
  pt_colorspace m_outputSpace{};
  pt_gmon_options m_gmonOptions{1.0f};
  mutable void* m_presentStream = nullptr;
  mutable std::string m_lastError;
};

}  // namespace ptamd::renderer_pt

#endif
