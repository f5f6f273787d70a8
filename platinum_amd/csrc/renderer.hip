// renderer.hip — host driver behind the C ABI (include/ptamd.h): the MI355X counterpart of
// `pt::renderer_pt::Renderer` (src/renderer_pt/renderer_pt.{hpp,cpp}).
//
//   pt_create          Renderer::Renderer + loadGgxLutTextures            renderer_pt.cpp:18-60, 385-446
//   pt_start_render    startRender + the rebuild* half of render()         :72-111, 199-217, 448-651, 838-1021
//   pt_render_step     render() steady state                               :113-197
//   pt_status/...      status / renderProgress / renderTime                :1023-1037
//   pt_read_accumulator  (the reference only reads back RGBA8, :1039-1059; the float mean is the parity surface)
//
// No CPU path exists in this library: without a HIP device pt_create fails.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "queue_plan.h"
#include "runtime_identity.h"
#include "renderer_state.h"

using namespace pt;

namespace {
thread_local std::string g_last_error;
}
int pt_fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const std::string& pt_last_error_string() { return g_last_error; }

namespace {

struct ScopedTimer {  // records a pair of events around one launch when profiling is on
  pt_renderer* r;
  int cls;
  hipEvent_t start = nullptr, stop = nullptr;
  ScopedTimer(pt_renderer* r_, int cls_) : r(r_), cls(cls_) {
    if (!r->profiling) return;
    if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) { start = stop = nullptr; return; }
    (void)hipEventRecord(start, r->stream);
  }
  ~ScopedTimer() {
    if (!start) return;
    (void)hipEventRecord(stop, r->stream);
    r->timed.push_back({cls, start, stop});
  }
};

enum BatchMode { BATCH_RENDER, BATCH_DEBUG, BATCH_MEASURE };

// One batch: `ns` samples of every pixel, first sample index `first`, `n0` samples already in the accumulator.
int enqueue_batch(pt_renderer* r, uint32_t first, uint32_t ns, uint32_t n0, BatchMode mode, int32_t* hitlog) {
  const DeviceScene& S = r->S;
  hipStream_t s = r->stream;
  BatchCounters* ctr = r->ctr.p;
  // the per-bounce hit log is indexed by pixel through pixel_of_pid_1spp (kernels.hip): it only exists for one-sample batches
  if (hitlog && ns != 1) return fail(PT_ERR_INVALID_ARGUMENT, "the hit log is kept for one-sample batches only");
  if (ns == 0 || ns > r->samples_in_flight) return fail(PT_ERR_INVALID_ARGUMENT, "batch larger than the queues");
  PT_HIP(hipMemsetAsync(ctr, 0, sizeof(BatchCounters), s));
  const bool count = mode == BATCH_MEASURE;
  Segments seg = r->segments();
  seg.nsamples = ns;
#ifdef PT_DEBUG_PID
  if (const char* e = getenv("PTAMD_DEBUG_RAY")) {  // debug build only: "x,y,sample,bounce" -> the closest-hit kernel prints that ray's traversal
    uint32_t x = 0, y = 0, sm = 0, b = 0;
    if (sscanf(e, "%u,%u,%u,%u", &x, &y, &sm, &b) == 4 && sm >= first && sm < first + ns) {
      const uint32_t tilesX = (S.width + 7) / 8;
      const uint32_t v[2] = {(((y >> 3) * tilesX + (x >> 3)) * 64u + (y & 7) * 8u + (x & 7)) * ns + (sm - first), b + 1u};
      PT_HIP(hipMemcpyAsync(&ctr->_pad[0], v, 8, hipMemcpyHostToDevice, s));
    }
  }
#endif
  {
    ScopedTimer t(r, K_RAYGEN);
    launch_raygen(s, r->grid, S, r->path_state(0), r->Lbuf.p, seg, ctr, first, ns);
    launch_chunk_tables(s, seg, 0, ctr, 0, 0, false);
  }
  int cur = 0;
  const bool mis = S.integrator == PT_INTEGRATOR_MIS;
  for (uint32_t b = 0; b < S.max_bounces; b++) {
    {
      ScopedTimer t(r, K_CLOSEST);
      launch_trace_closest(s, r->closest_grid, S, r->path_state(cur), r->hit.p, seg, (uint32_t)cur, ctr, b, r->spill.p, hitlog, S.width * S.height, count);
    }
    {
      ScopedTimer t(r, K_SHADE);
      launch_shade(s, r->shade_grid, r->scene_d.p, r->path_state(cur), r->path_state(cur ^ 1), r->hit.p, r->shadow_queue(), r->Lbuf.p, seg, (uint32_t)cur, ctr, b);
      // closest-hit list of bounce b + 1 (written to chunks_closest[b + 1]) and shadow list of bounce b
      launch_chunk_tables(s, seg, (uint32_t)(cur ^ 1), ctr, b + 1, b, mis);
    }
    if (mis) {
      ScopedTimer t(r, K_SHADOW);
      launch_trace_shadow(s, r->shadow_grid, S, r->shadow_queue(), r->Lbuf.p, seg, ctr, b, r->spill.p, count);
    }
    cur ^= 1;
  }
  if (mode == BATCH_RENDER) {
    ScopedTimer t(r, K_ACCUM);
    const uint32_t npix = S.width * S.height;
    if (r->params.flags & PT_FLAG_GMON) {
      const uint32_t buckets = r->params.gmon_buckets;
      const uint32_t total_spp = r->gmon_total_spp ? r->gmon_total_spp : r->params.spp;
      const uint32_t spb = (total_spp + buckets - 1) / buckets;  // renderer_pt.cpp:124-125
      const uint32_t f0 = n0 + r->gmon_sample_base;              // (a member of a device group: index within the whole render)
      launch_accumulate_gmon(s, r->gmon_buckets_d.p, r->Lbuf.p, npix, S.width, ns, f0, spb, buckets, r->gmon_bucket_base, r->params.nonfinite_policy, ctr);
      // the reference resolves after every frame with fullBuckets = gmonIdx + 1 (renderer_pt.cpp:164-179); only the last
      // resolve of a batch is observable
      launch_gmon(s, r->acc, r->gmon_buckets_d.p, npix, (f0 + ns - 1) / spb + 1 - r->gmon_bucket_base, r->gmon_cap);
    } else {
      launch_accumulate(s, r->acc, r->Lbuf.p, npix, S.width, ns, n0, r->params.nonfinite_policy, ctr);
    }
  }
  // BATCH_DEBUG still folds (to clear the per-wave statistics) but into a scratch Totals slot
  launch_fold_counters(s, ctr, mode == BATCH_DEBUG ? r->totals.p + 1 : r->totals.p, seg, count);
  PT_HIP(hipGetLastError());
  r->last_batch_ns = ns; r->last_batch_first = first;
  return PT_OK;
}

void collect_timings(pt_renderer* r) {
  for (auto& t : r->timed) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      r->ms_class[t.cls] += ms;
      r->launches[t.cls]++;
    }
  }
  r->drop_timed();
}

int load_luts(pt_renderer* r, const uint8_t* b, size_t size) {
  if (size < 12 + 16 * 8 || memcmp(b, "PTLUT01\0", 8) != 0) return fail(PT_ERR_BAD_LUT, "LUT blob: bad magic (expected PTLUT01)");
  uint32_t count;
  memcpy(&count, b + 8, 4);
  if (count != 8) return fail(PT_ERR_BAD_LUT, "LUT blob: expected 8 tables (renderer_pt.hpp:154-165)");
  uint32_t hdr[32];
  memcpy(hdr, b + 12, sizeof(hdr));
  const size_t data_off = 12 + 16 * 8;
  const size_t nfloats = (size - data_off) / 4;
  // expected shapes (renderer_pt.hpp:154-165 + resource/lut): E 128x128, Eavg 128, 3-D 32^3, 2-D avg 32^2
  const uint32_t expect[8][3] = {{128, 128, 1}, {128, 1, 1}, {32, 32, 32}, {32, 32, 1}, {32, 32, 32}, {32, 32, 32}, {32, 32, 1}, {32, 32, 1}};
  for (int i = 0; i < 8; i++) {
    const uint32_t w = hdr[4 * i], h = hdr[4 * i + 1], d = hdr[4 * i + 2], off = hdr[4 * i + 3];
    if (w != expect[i][0] || h != expect[i][1] || d != expect[i][2] || (size_t)off + (size_t)w * h * d > nfloats)
      return fail(PT_ERR_BAD_LUT, "LUT blob: unexpected table shape");
  }
  std::vector<float> data(nfloats);
  memcpy(data.data(), b + data_off, nfloats * 4);
  PT_HIP(r->lut_data.upload(data));
  Lut* ls[6] = {&r->luts.E, &r->luts.Eavg, &r->luts.EMs, &r->luts.EavgMs, &r->luts.ETransIn, &r->luts.ETransOut};
  for (int i = 0; i < 6; i++) {
    ls[i]->w = (int)hdr[4 * i]; ls[i]->h = (int)hdr[4 * i + 1]; ls[i]->depth = (int)hdr[4 * i + 2];
    ls[i]->d = r->lut_data.p + hdr[4 * i + 3];
  }
  r->lut_w_E = hdr[0];
  r->lut_w_Eavg = hdr[4];
  return PT_OK;
}

int build_halton_table(pt_renderer* r) {
  std::vector<HaltonEntry> tab;
  tab.reserve(kHaltonDims);
  for (uint32_t c = 2; (int)tab.size() < kHaltonDims; c++) {  // defs.metal:115-194: the first 620 primes
    bool prime = true;
    for (uint32_t d = 2; d * d <= c; d++)
      if (c % d == 0) { prime = false; break; }
    if (!prime) continue;
    tab.push_back(make_halton_entry(c));
  }
  PT_HIP(r->halton.upload(tab));
  return PT_OK;
}

}  // namespace

extern "C" const char* pt_last_error(void) { return g_last_error.c_str(); }

extern "C" int pt_plan_queues(uint32_t width, uint32_t height, uint32_t spp, uint32_t samples_in_flight, uint64_t free_hbm_bytes,
                              uint32_t tiles_per_seg_override, uint32_t seg_bands, pt_queue_plan* out) {
  if (!out) return fail(PT_ERR_INVALID_ARGUMENT, "pt_plan_queues: null argument");
  const char* why = "";
  const int rc = plan_queues(width, height, spp, samples_in_flight, free_hbm_bytes, tiles_per_seg_override, seg_bands, out, &why);
  return rc == PT_OK ? PT_OK : fail(rc, std::string("pt_plan_queues: ") + why);
}

std::string pt_rccl_bound_path();  // multi_device.hip: "" until RCCL has been loaded

extern "C" int pt_get_runtime_info(pt_runtime_info* out) {
  if (!out) return fail(PT_ERR_INVALID_ARGUMENT, "pt_get_runtime_info: null argument");
  memset(out, 0, sizeof(*out));
  const RuntimeObjects o = mapped_runtime_objects();
  auto put = [](char* dst, size_t cap, const std::string& s) { snprintf(dst, cap, "%s", s.c_str()); };
  put(out->hip_runtime_path, sizeof(out->hip_runtime_path), object_of((const void*)&hipRuntimeGetVersion));
  put(out->hsa_runtime_path, sizeof(out->hsa_runtime_path), o.hsa.empty() ? std::string() : o.hsa[0]);
  put(out->rccl_path, sizeof(out->rccl_path), pt_rccl_bound_path());
  out->hip_runtimes_mapped = (uint32_t)o.hip.size();
  out->hsa_runtimes_mapped = (uint32_t)o.hsa.size();
  out->rccl_mapped = (uint32_t)o.rccl.size();
  int v = 0;
  out->hip_runtime_version = hipRuntimeGetVersion(&v) == hipSuccess ? v : 0;
  put(out->all_mapped, sizeof(out->all_mapped), join_paths(o.hip) + " | " + join_paths(o.hsa) + " | " + join_paths(o.rccl));
  return PT_OK;
}

int dev_create(const pt_create_info* info, int device_ordinal, pt_renderer** out) {
  if (!info || !out) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: null argument");
  *out = nullptr;
  if (info->abi_version != PT_ABI_VERSION) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: ABI version mismatch");
  {  // one HIP runtime over one HSA runtime per process, or whoever initialises second sees no GPU (runtime_identity.h)
    const std::string conflict = runtime_conflict();
    if (!conflict.empty()) return fail(PT_ERR_RUNTIME_CONFLICT, "pt_create: " + conflict);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PT_ERR_NO_DEVICE, "pt_create: no HIP device available (this library has no CPU fallback)");
  if (device_ordinal < 0 || device_ordinal >= ndev) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: bad device ordinal");
  PT_HIP(hipSetDevice(device_ordinal));
  auto* r = new pt_renderer();
  pt_default_post_options(&r->post);
  pt_default_tonemap_options(&r->tonemap);
  r->device = device_ordinal;
  if (const char* e = getenv("PTAMD_REFILL")) r->refill_threshold = (uint32_t)atoi(e);
  if (const char* e = getenv("PTAMD_TILES_PER_SEG")) r->tiles_per_seg_override = (uint32_t)std::max(0, atoi(e));  // tuning knobs
  if (const char* e = getenv("PTAMD_SEG_BANDS")) r->seg_bands = (uint32_t)std::max(1, std::min(64, atoi(e)));
  if (const char* e = getenv("PTAMD_TWO_LEVEL")) r->two_level_override = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("PTAMD_BLOCKS_PER_CU")) r->blocks_per_cu = (uint32_t)std::max(1, std::min(8, atoi(e)));
  if (const char* e = getenv("PTAMD_CLOSEST_BLOCKS_PER_CU")) r->closest_blocks_per_cu = (uint32_t)std::max(1, std::min(8, atoi(e)));
  if (const char* e = getenv("PTAMD_SHADOW_BLOCKS_PER_CU")) r->shadow_blocks_per_cu = (uint32_t)std::max(1, std::min(8, atoi(e)));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, r->device) == hipSuccess) r->num_cu = prop.multiProcessorCount;
  int rc = PT_OK;
  do {
    if (hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(PT_ERR_HIP, "hipStreamCreate failed"); break; }
    r->stream = r->own_stream;
    if (hipEventCreateWithFlags(&r->batch_done, hipEventDisableTiming) != hipSuccess) { rc = fail(PT_ERR_HIP, "hipEventCreate failed"); break; }
    std::vector<uint8_t> file;
    const uint8_t* blob = (const uint8_t*)info->lut_blob;
    size_t size = (size_t)info->lut_blob_size;
    if (!blob) {
      const char* path = info->lut_path ? info->lut_path : getenv("PTAMD_LUT_PATH");
      if (!path) { rc = fail(PT_ERR_BAD_LUT, "pt_create: no LUT blob, lut_path or $PTAMD_LUT_PATH"); break; }
      FILE* f = fopen(path, "rb");
      if (!f) { rc = fail(PT_ERR_BAD_LUT, std::string("pt_create: cannot open LUT file ") + path); break; }
      fseek(f, 0, SEEK_END);
      long n = ftell(f);
      fseek(f, 0, SEEK_SET);
      file.resize((size_t)n);
      if (fread(file.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); rc = fail(PT_ERR_BAD_LUT, "pt_create: short read on LUT file"); break; }
      fclose(f);
      blob = file.data();
      size = file.size();
    }
    if ((rc = load_luts(r, blob, size)) != PT_OK) break;
    if ((rc = build_halton_table(r)) != PT_OK) break;
    if (r->ctr.alloc(1) != hipSuccess || r->totals.alloc(2) != hipSuccess) { rc = fail(PT_ERR_OUT_OF_MEMORY, "counter allocation failed"); break; }
  } while (0);
  if (rc != PT_OK) { dev_destroy(r); return rc; }
  *out = r;
  return PT_OK;
}

void dev_destroy(pt_renderer* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->stream) (void)hipStreamSynchronize(r->stream);
  r->drop_timed();
  r->release_all();  // the scene arrays, the queues AND the BVH builder's kept scratch (LbvhScratch has no destructor of its own)
  if (r->batch_done) (void)hipEventDestroy(r->batch_done);
  if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
  delete r;
}

int dev_start_render(pt_renderer* r, const pt_scene_snapshot* scene, const pt_render_params* p) {
  if (!r || !scene || !p) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: null argument");
  if (p->width == 0 || p->height == 0 || p->spp == 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: empty size or spp");
  if (p->max_bounces < 1 || p->max_bounces > 50)
    return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: max_bounces must be 1..50 (620 Halton dimensions, kernel.metal:5)");
  if ((uint64_t)p->width * p->height > (1ull << 28)) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: image too large");
  if (p->integrator != PT_INTEGRATOR_SIMPLE && p->integrator != PT_INTEGRATOR_MIS) return fail(PT_ERR_INVALID_ARGUMENT, "bad integrator");
  if (p->nonfinite_policy > PT_NONFINITE_ZERO) return fail(PT_ERR_INVALID_ARGUMENT, "bad nonfinite_policy");
  if (p->accel_structure > PT_ACCEL_TWO_LEVEL) return fail(PT_ERR_INVALID_ARGUMENT, "bad accel_structure");
  if ((p->flags & PT_FLAG_GMON) && (p->gmon_buckets < 1 || p->gmon_buckets > 32))
    return fail(PT_ERR_INVALID_ARGUMENT, "gmon_buckets must be 1..32 (gmon.metal:12 maxBuckets)");
  if (scene->instance_count && (!scene->instances || !scene->instance_materials || !scene->meshes))
    return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: null scene arrays");
  PT_HIP(hipSetDevice(r->device));
  if (r->stream) PT_HIP(hipStreamSynchronize(r->stream));
  r->drop_timed();
  r->free_scene();
  r->params = *p;
  r->stream = p->stream ? (hipStream_t)p->stream : r->own_stream;
  const auto t_up0 = std::chrono::steady_clock::now();
  const bool phase_log = getenv("PTAMD_START_PHASES") != nullptr;   // analysis aid: where a (re)start spends its wall time
  auto phase = [&, t_last = std::chrono::steady_clock::now()](const char* what) mutable {
    if (!phase_log) return;
    (void)hipDeviceSynchronize();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "ptamd start: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };

  // ---- flatten the snapshot, derive constants and the light table (host_scene.h) ----
  HostScene hs;
  {
    std::string err;
    const int rc = build_host_scene(scene, p, r->lut_w_E, r->lut_w_Eavg, &hs, &err);
    if (rc != PT_OK) return fail(rc, err);
  }
  if (hs.tri_count >= (1u << 27)) return fail(PT_ERR_UNSUPPORTED, "more than 2^27 flattened triangles (the hit record keeps 28 bits for 2 * leaf slot + half)");
  if (hs.instances.size() >= (1u << kSlotInstBits)) return fail(PT_ERR_UNSUPPORTED, "more than 2^26 instances (a leaf slot keeps 26 bits for the instance)");
  phase("host flatten + light table");
  r->instance_count = (uint32_t)hs.instances.size();
  r->tri_count = hs.tri_count;
  r->constants = hs.constants;
  r->lights = hs.lights;
  const pt_constants& C = r->constants;
  const Mat3 idt = hs.idt;

  // ---- upload ----
  PT_HIP(r->positions.upload(hs.positions));
  PT_HIP(r->vdata.upload(hs.vdata));
  PT_HIP(r->indices.upload(hs.indices));
  PT_HIP(r->slots.upload(hs.slots));
  PT_HIP(r->meshes.upload(hs.meshes));
  PT_HIP(r->instances.upload(hs.instances));
  PT_HIP(r->materials.upload(hs.materials));
  PT_HIP(r->lights_d.upload(r->lights));
  PT_HIP(r->tex_data.upload(hs.tex_data));
  PT_HIP(r->tex_decode.upload(hs.tex_decode));
  PT_HIP(r->textures.upload(hs.textures));
  PT_HIP(r->env_alias_d.upload(hs.env_alias));
  r->env_alias = std::move(hs.env_alias);

  DeviceScene& S = r->S;
  memset(&S, 0, sizeof(S));
  S.positions = r->positions.p; S.vdata = r->vdata.p; S.indices = r->indices.p; S.slots = r->slots.p;
  S.meshes = r->meshes.p; S.instances = r->instances.p; S.materials = r->materials.p; S.lights = r->lights_d.p;
  S.halton = r->halton.p;
  S.luts = r->luts;
  S.camera = C.camera;
  S.idt = idt;
  S.width = p->width; S.height = p->height;
  S.tex_data = r->tex_data.p; S.tex_decode = r->tex_decode.p; S.textures = r->textures.p; S.tex_native = hs.tex_native; S.env_alias = r->env_alias_d.p;
  S.env_texture = hs.env_texture;
  S.envLightCount = C.envLightCount;
  S.has_alpha = hs.has_alpha ? 1u : 0u;
  S.lightCount = C.lightCount;
  S.totalLightPower = C.totalLightPower;
  S.flags = p->flags;
  S.integrator = p->integrator;
  S.max_bounces = p->max_bounces;
  S.tri_count = r->tri_count;
  S.root_ref = kInvalidRef;
  phase("upload");
  PT_HIP(hipDeviceSynchronize());
  r->upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count();

  // ---- acceleration structure (replaces rebuildAccelerationStructures, renderer_pt.cpp:653-749) ----
  {
    struct EventPair {  // destroyed on every path out of this block
      hipEvent_t e0 = nullptr, e1 = nullptr;
      ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    PT_HIP(hipEventCreate(&ev.e0));
    PT_HIP(hipEventCreate(&ev.e1));
    PT_HIP(hipEventRecord(ev.e0, r->stream));
    // One BVH over the flattened triangles (fewest node visits per ray) unless the scene repeats its meshes so often that the
    // two-level structure — TLAS over instances + one BLAS per mesh, small enough to sit in LDS — is the better trade
    // (renderer_pt.cpp:653-749 always builds the two-level one).  Every instance must be invertible for that.
    uint64_t unique_tris = 0;
    for (const MeshInfo& m : hs.meshes) unique_tris += m.tri_count;
    std::vector<InstanceTrav> itrav(hs.instances.size());
    bool invertible = true;
    for (size_t i = 0; i < hs.instances.size(); i++) {
      const InstanceInfo& in = hs.instances[i];
      M3d m{};
      for (int k = 0; k < 3; k++) { m.m[0][k] = in.c0[k]; m.m[1][k] = in.c1[k]; m.m[2][k] = in.c2[k]; }
      const double det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[2][1] * m.m[1][2]) - m.m[1][0] * (m.m[0][1] * m.m[2][2] - m.m[2][1] * m.m[0][2]) +
                         m.m[2][0] * (m.m[0][1] * m.m[1][2] - m.m[1][1] * m.m[0][2]);
      double scale = 0;
      for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) scale = std::max(scale, std::fabs(m.m[c][k]));
      if (!(std::fabs(det) > 1e-12 * scale * scale * scale) || !std::isfinite(det)) { invertible = false; break; }
      const M3d inv = m3_inv(m);
      InstanceTrav& t = itrav[i];
      for (int k = 0; k < 3; k++) { t.ic0[k] = (float)inv.m[0][k]; t.ic1[k] = (float)inv.m[1][k]; t.ic2[k] = (float)inv.m[2][k]; t.c[k] = in.c3[k]; }
      t.tri_base = in.tri_global_base; t.mesh = in.mesh; t._pad[0] = t._pad[1] = 0;
    }
    // Measured on C3 (1024 instances of a 1012-triangle mesh, MI355X): the two-level walk costs 1.75x the time of the one-BVH
    // walk per ray (same 17.7 node visits, plus the instance entries) while its structure is 65 KB instead of 82 MB and reads
    // a fraction of the bytes.  Rays per second is what this renderer is for, so one BVH is the default and the two-level
    // structure is chosen when flattening would not leave room for the path queues (or on request, $PTAMD_TWO_LEVEL=1).
    {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
      // TriRec + nodes + ShadeRec + the builder's scratch at its peak.  Worst case: no triangle finds a partner (a soup, $PTAMD_NO_PAIRS) — one 64-byte
      // slot and TWO 32-byte shade records (entry 2 * slot + half; the B entries unused) per triangle, 32 B more than a paired mesh
      const uint64_t flat_bytes = (uint64_t)r->tri_count * 292ull;
      r->two_level = invertible && (uint64_t)r->tri_count >= 8 * unique_tris && free_b != 0 && flat_bytes > free_b / 2;
      // The switch is not silent: the two structures agree bit for bit except where fp32 Moeller-Trumbore reports a FALSE hit that
      // lies outside an instance's tighter object-space boxes (fuzz seed 310601: one pixel-sample in ~1 500 scenes).
      if (r->two_level && p->accel_structure == PT_ACCEL_AUTO && r->two_level_override < 0 && !getenv("PTAMD_QUIET"))
        fprintf(stderr, "ptamd: PT_ACCEL_AUTO chose the TWO-LEVEL structure (flattening %u triangles would take %.1f GB of the %.1f GB free): 1.75x the "
                        "traversal time of one BVH, and closest hits can differ from it where fp32 Moeller-Trumbore accepts a triangle the ray passes "
                        "outside of (the seed-310601 class: ~1 pixel-sample per 1 500 scenes); PT_ACCEL_ONE_BVH / PT_ACCEL_TWO_LEVEL pin the choice\n",
                r->tri_count, (double)flat_bytes / 1e9, (double)free_b / 1e9);
    }
    if (p->accel_structure == PT_ACCEL_ONE_BVH) r->two_level = false;
    else if (p->accel_structure == PT_ACCEL_TWO_LEVEL) r->two_level = invertible;
    else if (r->two_level_override >= 0) r->two_level = invertible && r->two_level_override != 0;
    hipError_t be;
    if (r->two_level) {
      PT_HIP(r->inst_trav.upload(itrav));
      be = build_two_level(r->stream, S, hs.meshes.data(), (uint32_t)hs.meshes.size(), r->instance_count, r->tri_count,
                           (uint32_t)(kLdsStack + kSpillStack), &r->bvh_scratch, &r->bvh);
    } else {
      // leaf slots: two consecutive triangles of a mesh that share an edge go into ONE slot ($PTAMD_NO_PAIRS: one triangle per slot, as r3 had it)
      build_primitives(&hs, getenv("PTAMD_NO_PAIRS") == nullptr);
      PT_HIP(r->prim_tri_d.upload(hs.prim_tri));
      PT_HIP(r->mesh_prim_base_d.upload(hs.mesh_prim_base));
      PT_HIP(r->inst_prim_base_d.upload(hs.inst_prim_base));
      PrimTables prims;
      prims.prim_tri = r->prim_tri_d.p; prims.mesh_prim_base = r->mesh_prim_base_d.p; prims.inst_prim_base = r->inst_prim_base_d.p;
      prims.slot_count = hs.prim_count;
      be = build_lbvh(r->stream, S, prims, r->instance_count, (uint32_t)(kLdsStack + kSpillStack), &r->bvh_scratch, &r->bvh);
    }
    if (be != hipSuccess)
      return fail(be == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP, std::string("LBVH build failed: ") + hipGetErrorString(be));
    PT_HIP(hipEventRecord(ev.e1, r->stream));
    PT_HIP(hipEventSynchronize(ev.e1));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev.e0, ev.e1);
    r->bvh_ms = ms;
    // traversal stack: <= 3 pushes per 4-wide level; 4-wide depth = ceil(binary depth / 2)
    if (r->bvh.depth4 * (r->bvh.wide6 ? 5u : 3u) + (r->two_level ? 1 : 0) > (uint32_t)(kLdsStack + kSpillStack))
      return fail(PT_ERR_UNSUPPORTED, "BVH too deep for the traversal stack (degenerate geometry: thousands of coincident triangles?)");
    S.nodes = r->bvh.nodes;
    S.tris = r->bvh.tris;
    S.slot_count = r->slot_count = r->bvh.slot_count;
    S.root_ref = r->bvh.root_ref;
    S.node_count = r->bvh.node_count;
    S.two_level = r->two_level ? 1u : 0u;
    S.wide6 = r->bvh.wide6 ? 1u : 0u;
    S.inst_trav = r->inst_trav.p;
    S.mesh_trav = r->bvh.mesh_trav;
  }
  phase("acceleration structure");
  PT_HIP(r->shade_recs.alloc(std::max<size_t>(1, 2 * (size_t)r->slot_count)));  // entry 2 * slot + half
  S.shade_recs = r->shade_recs.p;
  launch_shade_records(r->stream, S, r->shade_recs.p);
  PT_HIP(r->light_recs.alloc(std::max<size_t>(1, r->lights.size())));
  PT_HIP(r->light_cdf.alloc(std::max<size_t>(1, r->lights.size())));
  S.light_recs = r->light_recs.p;
  S.light_cdf = r->light_cdf.p;
  launch_light_records(r->stream, S, r->light_recs.p, r->light_cdf.p);
  PT_HIP(r->scene_d.upload(std::vector<DeviceScene>(1, S)));  // k_shade reads the table from memory (scalar loads)

  // ---- wavefront buffers ----
  phase("shade / light records");
  const uint64_t npix = (uint64_t)p->width * p->height;
  // Queue segments (kernels.hip): one per 8x8 tile (a few tiles each once the image has more than 32640 of them), each with
  // room for its tiles under all samples in flight; queue_plan.h holds the sizing and every index-width limit.  The producers
  // (raygen, shade) are persistent grids whose waves take segments round-robin; the trace kernels claim chunks from a table,
  // so every grid is sized for its own kernel's occupancy.
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
    else free_b += r->queue_bytes_held();  // what the previous render's queues occupy is reused, i.e. available
    pt_queue_plan plan{};
    const char* why = "";
    const int rc = plan_queues(p->width, p->height, p->spp, p->samples_in_flight, free_b, r->tiles_per_seg_override, r->seg_bands, &plan, &why);
    if (rc != PT_OK) return fail(rc, std::string("pt_start_render: ") + why);
    r->samples_in_flight = plan.samples_in_flight;
    r->tiles_per_seg = plan.tiles_per_seg;
    r->nseg = plan.nseg;
    r->seg_cap = plan.seg_cap;
    r->capacity = (size_t)plan.capacity;
  }
  const uint32_t sif = r->samples_in_flight;
  r->grid = (uint32_t)r->num_cu * r->blocks_per_cu;                 // raygen, hit records: 256-thread blocks
  r->shade_grid = (uint32_t)r->num_cu * shade_blocks_per_cu();     // as many blocks as k_shade's registers / LDS keep resident
  r->closest_grid = (uint32_t)r->num_cu * (r->two_level ? trace_blocks_per_cu_two_level() : r->closest_blocks_per_cu);
  r->shadow_grid = (uint32_t)r->num_cu * (r->two_level ? trace_blocks_per_cu_two_level() : r->shadow_blocks_per_cu);
  r->nstats = std::max(r->grid * (kBlock / 64), r->shade_grid * (shade_block_threads() / 64));
  for (int k = 0; k < 2; k++) {
    PT_HIP(r->st_rayO[k].alloc(r->capacity)); PT_HIP(r->st_rayD[k].alloc(r->capacity));
    PT_HIP(r->st_att[k].alloc(r->capacity));
  }
  PT_HIP(r->hit.alloc(r->capacity));
  PT_HIP(r->sq_o.alloc(r->capacity)); PT_HIP(r->sq_d.alloc(r->capacity)); PT_HIP(r->sq_c.alloc(r->capacity));
  PT_HIP(r->Lbuf.alloc((size_t)((p->width + 7) / 8) * ((p->height + 7) / 8) * 64 * sif));  // tile-major, whole tiles (kernels.hip lbuf_index)
  for (int k = 0; k < 2; k++) PT_HIP(r->seg_active[k].alloc(r->nseg));
  PT_HIP(r->seg_shadow.alloc(r->nseg));
  PT_HIP(r->seg_poison.alloc(r->nseg));
  PT_HIP(r->wave_stats.alloc(r->nstats));
  for (int k = 0; k < 2; k++) PT_HIP(r->chunk_table[k].alloc(r->capacity / 64 + 8));   // + 8: a trace wave loads the entries of a whole run of up to 8 chunks, wherever the list ends
  PT_HIP(r->shade_order.alloc(r->nseg));
  // (max_bounces + 1 rows: the table pass after the LAST bounce's k_shade still sorts for a bounce nobody runs, and reads that row)
  PT_HIP(r->shade_cost.alloc((size_t)r->nseg * (r->S.max_bounces + 1u)));
  PT_HIP(hipMemsetAsync(r->shade_cost.p, 0, sizeof(uint32_t) * r->shade_cost.n, r->stream));   // no batch of this render has been shaded yet
  PT_HIP(r->spill.alloc((size_t)std::max(r->closest_grid, r->shadow_grid) * trace_block_threads(r->two_level) * kSpillStack));  // per-thread HBM stack slab behind the LDS stack
  PT_HIP(hipMemsetAsync(r->wave_stats.p, 0, sizeof(WaveStats) * r->nstats, r->stream));
  if (p->external_accumulator) {
    r->acc = (vec4*)p->external_accumulator;
  } else {
    PT_HIP(r->acc_own.alloc(npix));
    r->acc = r->acc_own.p;
  }
  if (p->flags & PT_FLAG_GMON) {
    const uint32_t own = r->gmon_own_buckets ? r->gmon_own_buckets : p->gmon_buckets;
    PT_HIP(r->gmon_buckets_d.alloc((size_t)npix * own));
    PT_HIP(hipMemsetAsync(r->gmon_buckets_d.p, 0, sizeof(vec4) * npix * own, r->stream));
  }
  PT_HIP(hipMemsetAsync(r->acc, 0, sizeof(vec4) * npix, r->stream));
  PT_HIP(hipMemsetAsync(r->totals.p, 0, sizeof(Totals), r->stream));
  PT_HIP(hipStreamSynchronize(r->stream));

  phase("queues + accumulator");
  for (int k = 0; k < K_CLASSES; k++) { r->ms_class[k] = 0; r->launches[k] = 0; }
  r->accumulated = 0;
  r->launched = 0;
  r->batches = 0;
  r->last_batch_ns = 0;   // Lbuf holds nothing of THIS render yet
  r->batch_done_valid = false;
  r->total = p->spp;
  r->started = true;
  r->render_start = std::chrono::steady_clock::now();
  r->timer_ms = 0;
  return PT_OK;
}

// Enqueues accepted-but-pending samples.  The reference's frontend calls render() once per UI frame for exactly ONE sample
// (renderer_pt.cpp:131-153, frontend.cpp:209-210); a one-sample batch leaves the deep bounces of the wavefront too narrow for the
// chip (C3: 1 in flight runs at about a third of the 64-in-flight rate).  So a step that arrives while the previous batch is
// still executing is not enqueued on its own: it waits until the GPU has drained (then everything pending goes out as ONE batch)
// or until a full batch of `samples_in_flight` has gathered (enqueued behind the running one, so the GPU never idles).  A caller
// slower than the GPU gets exactly the old behaviour (one batch per call); a tight render() loop converges to full batches.
// The image does not depend on the batching: k_accumulate folds samples in index order.  `all`: enqueue everything now.
int flush_pending(pt_renderer* r, bool all) {
  while (r->launched < r->accumulated) {
    const uint64_t pending = r->accumulated - r->launched;
    bool idle = true;
    if (r->batch_done_valid) {
      const hipError_t q = hipEventQuery(r->batch_done);
      if (q == hipErrorNotReady) {
        idle = false;
        (void)hipGetLastError();  // hipErrorNotReady is an answer, not a failure: do not leave it for the next hipGetLastError() check
      } else if (q != hipSuccess) {  // a fault or a lost context is NOT "still busy": report it now, not at some later wait
        (void)hipGetLastError();
        return fail(PT_ERR_HIP, std::string("pt_render_step: the previous batch failed: ") + hipGetErrorString(q));
      }
    }
    if (!all && !idle && pending < r->samples_in_flight) break;
    const uint32_t ns = (uint32_t)std::min<uint64_t>(pending, r->samples_in_flight);
    const int rc = enqueue_batch(r, r->params.first_sample + (uint32_t)r->launched, ns, (uint32_t)r->launched, BATCH_RENDER, nullptr);
    if (rc != PT_OK) return rc;
    r->launched += ns;
    r->batches++;
    if (r->batch_done && hipEventRecord(r->batch_done, r->stream) == hipSuccess) r->batch_done_valid = true;
  }
  return PT_OK;
}

int dev_render_step(pt_renderer* r, uint32_t max_spp) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null renderer");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "pt_render_step before pt_start_render");
  PT_HIP(hipSetDevice(r->device));
  const uint64_t remaining = r->total - r->accumulated;
  r->accumulated += max_spp == 0 ? remaining : std::min<uint64_t>(max_spp, remaining);
  // the last samples of the render go out at once: nothing is ever pending while status() reports Done
  const int rc = flush_pending(r, r->accumulated == r->total);
  r->timer_ms = (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r->render_start).count();
  return rc;
}

int dev_wait(pt_renderer* r) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null renderer");
  PT_HIP(hipSetDevice(r->device));
  if (r->started) { const int rc = flush_pending(r, true); if (rc != PT_OK) return rc; }
  PT_HIP(hipStreamSynchronize(r->stream));
  collect_timings(r);
  if (r->started && getenv("PTAMD_DUMP_CHUNKS")) {  // analysis aid: 64-ray chunks per bounce of the LAST batch (the counters are per batch)
    BatchCounters h{};
    if (hipMemcpy(&h, r->ctr.p, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "ptamd chunks closest:");
      for (uint32_t b = 0; b < r->S.max_bounces; b++) fprintf(stderr, " %u", h.chunks_closest[b]);
      fprintf(stderr, "\nptamd chunks shadow:");
      for (uint32_t b = 0; b < r->S.max_bounces; b++) fprintf(stderr, " %u", h.chunks_shadow[b]);
      fprintf(stderr, "\n");
    }
  }
#ifdef PT_TAIL_PROBE   // analysis build only: per launch of the LAST batch, how long the chip waited for its last wave after the average wave had run out of work
  if (r->started) {
    BatchCounters h{};
    if (hipMemcpy(&h, r->ctr.p, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      const char* names[3] = {"closest", "shade", "shadow"};
      for (int k = 0; k < 3; k++) {
        double tot = 0, tail = 0;
        fprintf(stderr, "ptamd tail %s:", names[k]);
        for (uint32_t b = 0; b < r->S.max_bounces && b < 16; b++) {
          if (!h.tail_waves[k][b]) continue;
          const double t0 = (double)(~h.tail_start_inv[k][b]), t1 = (double)h.tail_end_max[k][b], mean = (double)h.tail_end_sum[k][b] / (double)h.tail_waves[k][b];
          fprintf(stderr, " b%u %.2f ms, idle tail %.1f %%;", b, (t1 - t0) / 1e5, 100.0 * (t1 - mean) / (t1 - t0));
          tot += t1 - t0; tail += t1 - mean;
        }
        fprintf(stderr, "  all: %.2f ms, %.1f %% idle tail", tot / 1e5, tot > 0 ? 100.0 * tail / tot : 0.0);
        if (k != 1) {
          double take = 0, setup = 0, busy = 0;
          for (uint32_t b = 0; b < r->S.max_bounces && b < 16; b++) { take += (double)h.tail_take[k][b]; setup += (double)h.tail_setup[k][b]; busy += (double)h.tail_busy[k][b]; }
          if (busy > 0) fprintf(stderr, "; wave time: %.1f %% taking chunks, %.1f %% loading rays + set-up", 100.0 * take / busy, 100.0 * setup / busy);
        }
        fprintf(stderr, "\n");
      }
    } else (void)hipGetLastError();
  }
#endif
#ifdef PT_DEBUG_PID   // debug build only (tools/build_variant.sh dbg -DPT_DEBUG_PID), like $PTAMD_DEBUG_RAY: nothing of it ships in libptamd.so
  if (r->started && r->last_batch_ns) {
    if (const char* e = getenv("PTAMD_DEBUG_PIXEL")) {  // analysis aid: "x,y" -> the per-sample radiance of that pixel in the LAST batch
      uint32_t x = 0, y = 0;
      if (sscanf(e, "%u,%u", &x, &y) == 2 && x < r->S.width && y < r->S.height) {
        const uint32_t tilesX = (r->S.width + 7) / 8, ns = r->last_batch_ns;
        const uint32_t tile = (y >> 3) * tilesX + (x >> 3), pl = (y & 7) * 8 + (x & 7);
        const size_t at0 = lbuf_index_host(tile, 0, ns, pl), stride = lbuf_sample_stride_host();   // the layout kernels.hip was compiled with
        std::vector<vec4> v(ns);
        bool ok = at0 + (size_t)(ns - 1) * stride < r->Lbuf.n;   // (a restart with another size / batch before any new batch: last_batch_ns is reset then, this is the belt)
        for (uint32_t k = 0; ok && k < ns; k++) ok = hipMemcpy(&v[k], r->Lbuf.p + at0 + (size_t)k * stride, sizeof(vec4), hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) (void)hipGetLastError();   // an analysis aid must not leave an error for the next batch's hipGetLastError()
        for (uint32_t k = 0; ok && k < ns; k++) {
          uint32_t b[3]; memcpy(b, &v[k], 12);
          fprintf(stderr, "ptamd pixel %u,%u sample %u: %08x %08x %08x  %.9g %.9g %.9g\n", x, y, r->last_batch_first + k, b[0], b[1], b[2], v[k].x, v[k].y, v[k].z);
        }
      }
    }
  }
#endif
  if (r->started)
    r->timer_ms = (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r->render_start).count();
  return PT_OK;
}

int dev_status(const pt_renderer* r) {  // renderer_pt.cpp:1023-1031
  if (!r) return PT_STATUS_BLOCKED;
  if (r->started && r->accumulated < r->total) return PT_STATUS_BUSY;
  int st = PT_STATUS_READY;
  if (r->started) st |= PT_STATUS_DONE;
  return st;
}

int dev_progress(const pt_renderer* r, uint64_t* accumulated, uint64_t* total) {
  if (!r || !accumulated || !total) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  *accumulated = r->accumulated;
  *total = r->total;
  return PT_OK;
}

uint64_t dev_render_time_ms(const pt_renderer* r) { return r ? r->timer_ms : 0; }

int dev_read_accumulator(pt_renderer* r, float* rgba_out) {
  if (!r || !rgba_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "pt_read_accumulator before pt_start_render");
  int rc = dev_wait(r);
  if (rc != PT_OK) return rc;
  PT_HIP(hipMemcpy(rgba_out, r->acc, sizeof(vec4) * (size_t)r->S.width * r->S.height, hipMemcpyDeviceToHost));
  return PT_OK;
}

extern "C" void pt_default_post_options(pt_post_options* o) {  // core/postprocessing.hpp:168-198
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->ca_green_shift = 70.0f;
  o->vig_feather = 50.0f; o->vig_power = 20.0f; o->vig_roundness = 100.0f;
}

extern "C" void pt_default_tonemap_options(pt_tonemap_options* o) {  // postprocessing.hpp:29-160, 209-226
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->tonemapper = PT_TONEMAP_AGX;
  for (int k = 0; k < 3; k++) { o->agx_slope[k] = 1.0f; o->agx_power[k] = 1.0f; }
  o->agx_saturation = 1.0f;
  o->khr_compression_start = 0.8f; o->khr_desaturation = 0.15f;
  o->flim_pre_exposure = 4.3f;
  for (int k = 0; k < 3; k++) { o->flim_pre_formation_filter[k] = 1.0f; o->flim_extended_gamut_mul[k] = 1.0f; o->flim_print_backlight[k] = 1.0f; o->flim_post_formation_filter[k] = 1.0f; }
  o->flim_extended_gamut_scale[0] = 1.05f; o->flim_extended_gamut_scale[1] = 1.12f; o->flim_extended_gamut_scale[2] = 1.045f;
  o->flim_extended_gamut_rotation[0] = 0.5f; o->flim_extended_gamut_rotation[1] = 2.0f; o->flim_extended_gamut_rotation[2] = 0.1f;
  o->flim_sigmoid_log2_min = -10.0f; o->flim_sigmoid_log2_max = 22.0f;
  o->flim_sigmoid_toe[0] = 0.440f; o->flim_sigmoid_toe[1] = 0.280f;
  o->flim_sigmoid_shoulder[0] = 0.591f; o->flim_sigmoid_shoulder[1] = 0.779f;
  o->flim_negative_exposure = 6.0f; o->flim_negative_density = 5.0f;
  o->flim_print_exposure = 6.0f; o->flim_print_density = 27.5f;
  o->flim_black_point = 0.0f; o->flim_auto_black_point = 1;
  o->flim_midtone_saturation = 1.02f;
  for (int k = 0; k < 3; k++) { o->shadow_color[k] = 0.5f; o->midtone_color[k] = 0.5f; o->highlight_color[k] = 0.5f; }
  const float p3[4][2] = {{0.680f, 0.320f}, {0.265f, 0.690f}, {0.150f, 0.060f}, {0.3127f, 0.3290f}};  // colorspace.cpp:6
  for (int k = 0; k < 2; k++) { o->output_space.r[k] = p3[0][k]; o->output_space.g[k] = p3[1][k]; o->output_space.b[k] = p3[2][k]; o->output_space.w[k] = p3[3][k]; }
}

int dev_set_post_options(pt_renderer* r, const pt_post_options* o) {
  if (!r || !o) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  r->post = *o;
  return PT_OK;
}

int dev_set_tonemap_options(pt_renderer* r, const pt_tonemap_options* o) {
  if (!r || !o) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (o->tonemapper > PT_TONEMAP_FLIM) return fail(PT_ERR_INVALID_ARGUMENT, "bad tonemapper");
  r->tonemap = *o;
  return PT_OK;
}

int dev_postprocess_to_host(pt_renderer* r, const vec4* acc_device, uint8_t* rgba8_out) {
  PT_HIP(hipSetDevice(r->device));
  const size_t npix = (size_t)r->S.width * r->S.height;
  if (r->render_target.n != npix) PT_HIP(r->render_target.alloc(npix));
  PostConstants pc;
  pc.post = r->post;
  pc.tm = r->tonemap;
  const Mat3 odt = compute_transform(r->params.working_space, r->tonemap.output_space);  // renderer_pt.cpp:190-191
  pc.odt = PPMat3{odt.c0, odt.c1, odt.c2};
  launch_postprocess(r->stream, acc_device, r->render_target.p, r->S.width, r->S.height, pc);
  PT_HIP(hipGetLastError());
  PT_HIP(hipStreamSynchronize(r->stream));
  PT_HIP(hipMemcpy(rgba8_out, r->render_target.p, npix * 4, hipMemcpyDeviceToHost));
  return PT_OK;
}

// the device half of dev_postprocess_to_host: enqueue only
int dev_present(pt_renderer* r, const vec4* acc_device, void** device_rgba8_out, void** stream_out) {
  if (!r || !device_rgba8_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "pt_present_render_target before pt_start_render");
  PT_HIP(hipSetDevice(r->device));
  { const int rc = flush_pending(r, true); if (rc != PT_OK) return rc; }  // the image shows every sample accepted so far
  const size_t npix = (size_t)r->S.width * r->S.height;
  if (r->render_target.n != npix) PT_HIP(r->render_target.alloc(npix));
  PostConstants pc;
  pc.post = r->post;
  pc.tm = r->tonemap;
  const Mat3 odt = compute_transform(r->params.working_space, r->tonemap.output_space);  // renderer_pt.cpp:190-191
  pc.odt = PPMat3{odt.c0, odt.c1, odt.c2};
  launch_postprocess(r->stream, acc_device ? acc_device : r->acc, r->render_target.p, r->S.width, r->S.height, pc);
  PT_HIP(hipGetLastError());
  *device_rgba8_out = r->render_target.p;
  if (stream_out) *stream_out = (void*)r->stream;
  return PT_OK;
}

int dev_read_render_target(pt_renderer* r, uint8_t* rgba8_out) {
  if (!r || !rgba8_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "pt_read_render_target before pt_start_render");
  int rc = dev_wait(r);
  if (rc != PT_OK) return rc;
  return dev_postprocess_to_host(r, r->acc, rgba8_out);
}

int dev_set_gmon_options(pt_renderer* r, const pt_gmon_options* o) {
  if (!r || !o) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  r->gmon_cap = o->cap;
  return PT_OK;
}

int dev_read_gmon_bucket(pt_renderer* r, uint32_t bucket, float* rgba_out) {
  if (!r || !rgba_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  const uint32_t own = r->gmon_own_buckets ? r->gmon_own_buckets : r->params.gmon_buckets;
  if (!r->started || !(r->params.flags & PT_FLAG_GMON) || bucket < r->gmon_bucket_base || bucket - r->gmon_bucket_base >= own)
    return fail(PT_ERR_BAD_STATE, "no such GMoN bucket");
  bucket -= r->gmon_bucket_base;
  int rc = dev_wait(r);
  if (rc != PT_OK) return rc;
  const size_t npix = (size_t)r->S.width * r->S.height;
  PT_HIP(hipMemcpy(rgba_out, r->gmon_buckets_d.p + npix * bucket, sizeof(vec4) * npix, hipMemcpyDeviceToHost));
  return PT_OK;
}

void* dev_accumulator_device_ptr(pt_renderer* r) { return (r && r->started) ? (void*)r->acc : nullptr; }

int dev_get_constants(const pt_renderer* r, pt_constants* out) {
  if (!r || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  *out = r->constants;
  return PT_OK;
}

int dev_get_lights(const pt_renderer* r, pt_area_light* out, uint32_t capacity, uint32_t* count) {
  if (!r || !count) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  *count = (uint32_t)r->lights.size();
  if (out)
    for (uint32_t i = 0; i < std::min<uint32_t>(capacity, *count); i++) out[i] = r->lights[i];
  return PT_OK;
}

int dev_get_env_alias(const pt_renderer* r, pt_alias_entry* out, uint64_t capacity, uint64_t* count) {
  if (!r || !count) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  *count = r->env_alias.size();
  if (out)
    for (uint64_t i = 0; i < std::min<uint64_t>(capacity, *count); i++) out[i] = r->env_alias[i];
  return PT_OK;
}

int dev_trace_primary(pt_renderer* r, uint32_t sample_idx, pt_hit_record* out) {
  if (!r || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  PT_HIP(hipSetDevice(r->device));
  { const int rc = flush_pending(r, true); if (rc != PT_OK) return rc; }
  const uint32_t npix = r->S.width * r->S.height;
  DevBuf<pt_hit_record> rec;
  PT_HIP(rec.alloc(npix));
  hipStream_t s = r->stream;
  PT_HIP(hipMemsetAsync(r->ctr.p, 0, sizeof(BatchCounters), s));
  Segments seg = r->segments();
  seg.nsamples = 1;
  launch_raygen(s, r->grid, r->S, r->path_state(0), r->Lbuf.p, seg, r->ctr.p, sample_idx, 1);
  launch_chunk_tables(s, seg, 0, r->ctr.p, 0, 0, false);
  launch_trace_closest(s, r->closest_grid, r->S, r->path_state(0), r->hit.p, seg, 0, r->ctr.p, 0, r->spill.p, nullptr, npix, false);
  launch_hit_records(s, r->grid, r->S, r->path_state(0), r->hit.p, seg, rec.p);
  launch_fold_counters(s, r->ctr.p, r->totals.p + 1, seg, false);  // clears the per-wave statistics (scratch slot)
  PT_HIP(hipGetLastError());
  PT_HIP(hipStreamSynchronize(s));
  PT_HIP(hipMemcpy(out, rec.p, sizeof(pt_hit_record) * npix, hipMemcpyDeviceToHost));
  return PT_OK;
}

int dev_debug_sample(pt_renderer* r, uint32_t sample_idx, float* radiance_out, int32_t* hits_out) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  PT_HIP(hipSetDevice(r->device));
  { const int rc = flush_pending(r, true); if (rc != PT_OK) return rc; }
  const size_t npix = (size_t)r->S.width * r->S.height;
  DevBuf<int32_t> log;
  if (hits_out) {
    PT_HIP(log.alloc(npix * 2 * r->S.max_bounces));
    PT_HIP(hipMemsetAsync(log.p, 0xff, sizeof(int32_t) * log.n, r->stream));
  }
  int rc = enqueue_batch(r, sample_idx, 1, 0, BATCH_DEBUG, log.p);
  if (rc != PT_OK) return rc;
  PT_HIP(hipStreamSynchronize(r->stream));
  r->drop_timed();
  if (radiance_out) {  // Lbuf of a one-sample batch is [tile][lane]: bring it back and put it in image order
    const uint32_t W = r->S.width, H = r->S.height, tilesX = (W + 7) / 8, tilesY = (H + 7) / 8;
    std::vector<vec4> tiled((size_t)tilesX * tilesY * 64);
    PT_HIP(hipMemcpy(tiled.data(), r->Lbuf.p, sizeof(vec4) * tiled.size(), hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < H; y++)
      for (uint32_t x = 0; x < W; x++) {
        const vec4& v = tiled[((size_t)(y >> 3) * tilesX + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7)];
        float* o = radiance_out + ((size_t)y * W + x) * 4;
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
      }
  }
  if (hits_out) PT_HIP(hipMemcpy(hits_out, log.p, sizeof(int32_t) * log.n, hipMemcpyDeviceToHost));
  return PT_OK;
}

int dev_measure_traversal(pt_renderer* r, uint32_t sample_idx) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  PT_HIP(hipSetDevice(r->device));
  { const int rc = flush_pending(r, true); if (rc != PT_OK) return rc; }
  const bool prof = r->profiling;
  r->profiling = false;
  int rc = enqueue_batch(r, sample_idx, 1, 0, BATCH_MEASURE, nullptr);
  r->profiling = prof;
  if (rc != PT_OK) return rc;
  PT_HIP(hipStreamSynchronize(r->stream));
  return PT_OK;
}

int dev_set_profiling(pt_renderer* r, int enabled) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  r->profiling = enabled != 0;
  return PT_OK;
}

int dev_get_stats(pt_renderer* r, pt_stats* out) {
  if (!r || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  if (!r->started) return fail(PT_ERR_BAD_STATE, "no render started");
  int rc = dev_wait(r);
  if (rc != PT_OK) return rc;
  Totals t{};
  PT_HIP(hipMemcpy(&t, r->totals.p, sizeof(Totals), hipMemcpyDeviceToHost));
  memset(out, 0, sizeof(*out));
  out->triangles = r->tri_count;
  out->leaf_slots = r->slot_count;
  out->bvh_nodes = r->bvh.node_count;
  out->bvh_max_depth = r->bvh.depth4;
  out->samples_in_flight = r->samples_in_flight;
  out->upload_ms = r->upload_ms;
  out->bvh_build_ms = r->bvh_ms;
  out->closest_rays = t.closest_rays;
  out->shadow_rays = t.shadow_rays;
  out->shaded_hits = t.shaded_hits;
  out->paths = t.paths;
  out->nonfinite_samples = t.nonfinite;
  out->ms_raygen = r->ms_class[K_RAYGEN]; out->ms_closest = r->ms_class[K_CLOSEST]; out->ms_shade = r->ms_class[K_SHADE];
  out->ms_shadow = r->ms_class[K_SHADOW]; out->ms_accumulate = r->ms_class[K_ACCUM];
  out->accel_two_level = r->two_level ? 1u : 0u;
  out->batches = r->batches;
  out->launches_closest = r->launches[K_CLOSEST];
  out->launches_shadow = r->launches[K_SHADOW];
  if (t.counted_closest) {
    out->nodes_per_closest_ray = (double)t.nodes_closest / (double)t.counted_closest;
    out->tris_per_closest_ray = (double)t.tris_closest / (double)t.counted_closest;
  }
  if (t.counted_shadow) {
    out->nodes_per_shadow_ray = (double)t.nodes_shadow / (double)t.counted_shadow;
    out->tris_per_shadow_ray = (double)t.tris_shadow / (double)t.counted_shadow;
  }
  return PT_OK;
}

