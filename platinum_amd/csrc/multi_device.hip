// multi_device.hip — the extern "C" entry points of include/ptamd.h and the DEVICE GROUP behind them.
//
// independent-sample parallelism.  pt_create with a device list builds one single-device renderer
// (renderer.hip) per entry — scene, BVH and queues replicated per device — each driven by its own host thread on its own
// stream.  pt_start_render deals the render's sample indices [first, first + spp) to the members in contiguous ranges
// (`frameIdx` is the only seed of the reference's sampler, samplers.metal:154-156, so the union over the members is exactly
// the sample set of one big render); nothing is exchanged while rendering.  The merge happens when the image is asked for
// (pt_wait / pt_read_*):
//   plain accumulation   every member scales its running mean by (its samples / all samples) into a scratch image, ONE
//                        ncclAllReduce(sum, float32, 4*W*H) over RCCL/xGMI combines them, device 0 keeps the result
//                        (in the caller's external accumulator when one was given).
//   GMoN                 the sample ranges follow bucket boundaries (bucket b = samples [b*ceil(spp/B), ...),
//                        renderer_pt.cpp:124-126), so every bucket image lives on exactly one device; the bucket means are
//                        gathered to device 0 (hipMemcpyPeer) and k_gmon runs there — bit-identical to the single-device result.
// Members that share a physical device (a device listed twice: logical shards, used by the 1-GPU tests) are summed by a kernel
// on that device; RCCL is only entered with distinct devices, and is loaded (dlopen librccl.so) only then.
// The reference has no counterpart: it renders on the one MTL::Device of its window (frontend.cpp:138).
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: librccl.so itself is loaded with dlopen on first use

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "renderer_state.h"
#include "runtime_identity.h"

namespace {

// ---- one host thread per member ----------------------------------------------------------------------------------------
struct Worker {
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has = false, stop = false;
  int rc = PT_OK;
  std::string err;
  std::thread th;  // declared LAST and started in the constructor body: run() must find every other member constructed
  Worker() { th = std::thread([this] { run(); }); }
  ~Worker() {
    { std::lock_guard<std::mutex> l(m); stop = true; }
    cv.notify_all();
    th.join();
  }
  void run() {
    std::unique_lock<std::mutex> l(m);
    for (;;) {
      cv.wait(l, [this] { return has || stop; });
      if (stop) return;
      std::function<int()> j = std::move(job);
      l.unlock();
      const int r = j();
      std::string e = r != PT_OK ? pt_last_error_string() : std::string();  // (pt_last_error is thread-local: carry it over)
      l.lock();
      rc = r; err = std::move(e); has = false;
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [this] { return !has; });
    job = std::move(j); has = true;
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [this] { return !has; });
    return rc;
  }
};

// ---- RCCL, loaded on first use -------------------------------------------------------------------------------------------
// The library is dlopen'ed (a one-GPU host never needs it), but the entry points are typed from the image's own <rccl/rccl.h>,
// so a changed signature or enumerator is a compile error here instead of a wrong call on an 8-GPU node.
struct Rccl {
  typedef ncclComm_t comm_t;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  void* lib = nullptr;
  static constexpr ncclDataType_t kFloat32 = ncclFloat32;
  static constexpr ncclRedOp_t kSum = ncclSum;
  std::string path;  // the shared object the entry points were bound from
  // Prefers a librccl that is ALREADY mapped (in a process that holds PyTorch that is torch's bundled copy, built against the HIP runtime
  // the process runs on): loading a second RCCL beside it would be a second set of collectives over the same runtime at best, and one
  // linked against the other HIP runtime at worst.  Only when none is mapped is one searched for (this library's RUNPATH: /opt/rocm/lib).
  // The resolved path is logged once and reported by pt_get_runtime_info.
  bool load(std::string* why) {
    if (lib && CommInitAll) return true;
    if (!lib) {
      const RuntimeObjects mapped = mapped_runtime_objects();
      for (const std::string& p : mapped.rccl) {
        lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (lib) break;
      }
      if (!lib)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
          lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
          if (lib) break;
        }
    }
    if (!lib) { const char* e = dlerror(); *why = std::string("cannot load librccl.so: ") + (e ? e : "?"); return false; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { *why = std::string("librccl.so lacks ") + n; ok = false; } return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (!ok) { CommInitAll = nullptr; return false; }
    path = object_of((const void*)CommInitAll);
    // RCCL brings its own DT_NEEDED libamdhip64: if that mapped a second HIP runtime, its collectives would run on a runtime that
    // does not own this library's buffers and streams
    const std::string conflict = runtime_conflict();
    if (!conflict.empty()) { *why = "after loading " + path + ": " + conflict; CommInitAll = nullptr; return false; }
    if (!getenv("PTAMD_QUIET"))
      fprintf(stderr, "ptamd: RCCL bound from %s (HIP runtime %s)\n", path.c_str(), object_of((const void*)&hipRuntimeGetVersion).c_str());
    return true;
  }
};
static_assert(ncclFloat32 == 7 && ncclSum == 0, "rccl.h enumerators this file was written against");
Rccl g_rccl;
std::mutex g_rccl_mutex;

}  // namespace

std::string pt_rccl_bound_path() {
  std::lock_guard<std::mutex> l(g_rccl_mutex);
  return g_rccl.CommInitAll ? g_rccl.path : std::string();
}

// A physical device of the group: the members on it, the scratch image they are summed into, its RCCL rank.
struct PhysDevice {
  int ordinal = 0;
  std::vector<size_t> members;
  vec4* scratch = nullptr;
  hipStream_t stream = nullptr;  // the first member's stream
  Rccl::comm_t comm = nullptr;
};

struct DeviceGroup {
  std::vector<pt_renderer*> shards;
  std::vector<std::unique_ptr<Worker>> workers;
  std::vector<uint64_t> first, count;       // sample range of each member in the current render
  std::vector<uint32_t> bucket0, bucket1;   // GMoN: global bucket range of each member
  std::vector<PhysDevice> phys;
  bool comms_ready = false;
  pt_render_params params{};
  bool started = false, dirty = false;
  vec4* merged = nullptr;                    // on shards[0]'s device: the caller's external accumulator or merged_own
  vec4* merged_own = nullptr;
  vec4* gmon_gather = nullptr;               // [gmon_buckets][pixel] on shards[0]'s device
  std::chrono::steady_clock::time_point render_start;
  uint64_t timer_ms = 0;

  template <class F>
  int for_all(F f) {  // f(member index) on every member's own thread; first error wins
    for (size_t g = 0; g < shards.size(); g++) workers[g]->post([f, g] { return f(g); });
    int rc = PT_OK;
    for (size_t g = 0; g < shards.size(); g++) {
      const int r = workers[g]->wait();
      if (r != PT_OK && rc == PT_OK) rc = fail(r, workers[g]->err);
    }
    return rc;
  }
  void release_images() {
    if (shards.empty()) return;
    (void)hipSetDevice(shards[0]->device);
    if (merged_own) (void)hipFree(merged_own);
    if (gmon_gather) (void)hipFree(gmon_gather);
    merged_own = gmon_gather = nullptr;
    merged = nullptr;
    for (auto& pd : phys) {
      (void)hipSetDevice(pd.ordinal);
      if (pd.scratch) (void)hipFree(pd.scratch);
      pd.scratch = nullptr;
    }
  }
};

namespace {

bool is_group(const pt_renderer* r) { return r && r->group; }

int group_create(const pt_create_info* info, pt_renderer** out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PT_ERR_NO_DEVICE, "pt_create: no HIP device available (this library has no CPU fallback)");
  if (info->device_count > 64) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: more than 64 devices");
  auto* front = new pt_renderer();
  auto* grp = new DeviceGroup();
  front->group = grp;
  front->device = info->device_ordinals[0];
  int rc = PT_OK;
  for (uint32_t g = 0; g < info->device_count && rc == PT_OK; g++) {
    pt_renderer* m = nullptr;
    rc = dev_create(info, info->device_ordinals[g], &m);
    if (rc != PT_OK) break;
    grp->shards.push_back(m);
    grp->workers.emplace_back(new Worker());
    auto it = std::find_if(grp->phys.begin(), grp->phys.end(), [&](const PhysDevice& p) { return p.ordinal == m->device; });
    if (it == grp->phys.end()) { grp->phys.push_back(PhysDevice{}); it = grp->phys.end() - 1; it->ordinal = m->device; it->stream = m->stream; }
    it->members.push_back(g);
  }
  if (rc != PT_OK) {
    for (auto* m : grp->shards) dev_destroy(m);
    delete grp;
    delete front;
    return rc;
  }
  pt_default_post_options(&front->post);
  pt_default_tonemap_options(&front->tonemap);
  *out = front;
  return PT_OK;
}

void group_destroy(pt_renderer* front) {
  DeviceGroup* grp = front->group;
  for (auto* m : grp->shards) { (void)hipSetDevice(m->device); if (m->stream) (void)hipStreamSynchronize(m->stream); }
  grp->release_images();
  if (grp->comms_ready)
    for (auto& pd : grp->phys) if (pd.comm) (void)g_rccl.CommDestroy(pd.comm);
  grp->workers.clear();  // joins the threads
  for (auto* m : grp->shards) dev_destroy(m);
  delete grp;
  delete front;
}

// what dev_start_render would reject, checked BEFORE the group's previous render is torn down (renderer.hip dev_start_render)
int group_validate_params(const pt_scene_snapshot* scene, const pt_render_params* p) {
  if (!scene || !p) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: null argument");
  if (p->spp == 0 || p->width == 0 || p->height == 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: empty size or spp");
  if (p->stream) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: a caller stream cannot drive a device group (every member owns its stream)");
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
  return PT_OK;
}

// a failed (re)start leaves the group with NO render: nothing a later pt_wait / pt_read_* could merge or present
int group_abandon(DeviceGroup* grp, int rc) {
  const std::string why = pt_last_error_string();  // (release_images makes HIP calls; keep the first error's text)
  grp->started = false;
  grp->dirty = false;
  for (auto* m : grp->shards) m->started = false;
  grp->release_images();
  return fail(rc, why);
}

int group_start_render(pt_renderer* front, const pt_scene_snapshot* scene, const pt_render_params* p) {
  DeviceGroup* grp = front->group;
  {
    const int rc = group_validate_params(scene, p);
    if (rc != PT_OK) return rc;  // the previous render (if any) stays as it was
  }
  const size_t N = grp->shards.size();
  const bool gmon = (p->flags & PT_FLAG_GMON) != 0;
  std::vector<uint64_t> first(N, 0), count(N, 0);
  std::vector<uint32_t> bucket0(N, 0), bucket1(N, 0);
  {
    const int rc = pt_group_partition(p->spp, (uint32_t)N, p->flags, p->gmon_buckets, first.data(), count.data(), bucket0.data(), bucket1.data());
    if (rc != PT_OK) return rc;
  }
  // from here on the previous render is gone: no member is "started" until the new one is completely set up
  for (auto* m : grp->shards) { (void)hipSetDevice(m->device); if (m->stream) (void)hipStreamSynchronize(m->stream); m->started = false; }
  grp->started = false;
  grp->dirty = false;
  grp->release_images();
  grp->params = *p;
  grp->first = first; grp->count = count; grp->bucket0 = bucket0; grp->bucket1 = bucket1;
  const size_t npix = (size_t)p->width * p->height;
  // every member renders its range into an accumulator of its own
  int rc = grp->for_all([grp, scene, p](size_t g) {
    pt_renderer* m = grp->shards[g];
    if (grp->count[g] == 0) { m->started = false; return (int)PT_OK; }
    pt_render_params q = *p;
    q.spp = (uint32_t)grp->count[g];
    q.first_sample = p->first_sample + (uint32_t)grp->first[g];
    q.external_accumulator = nullptr;
    q.stream = nullptr;
    m->gmon_sample_base = (uint32_t)grp->first[g];
    m->gmon_total_spp = p->spp;
    m->gmon_bucket_base = grp->bucket0[g];
    m->gmon_own_buckets = (p->flags & PT_FLAG_GMON) ? std::max(1u, grp->bucket1[g] - grp->bucket0[g]) : 0;
    return dev_start_render(m, scene, &q);
  });
  if (rc != PT_OK) return group_abandon(grp, rc);
  // images of the merge
  pt_renderer* m0 = grp->shards[0];
#define PT_GRP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); return group_abandon(grp, e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP); } } while (0)
  PT_GRP(hipSetDevice(m0->device));
  if (p->external_accumulator) grp->merged = (vec4*)p->external_accumulator;
  else { PT_GRP(hipMalloc((void**)&grp->merged_own, sizeof(vec4) * npix)); grp->merged = grp->merged_own; }
  PT_GRP(hipMemset(grp->merged, 0, sizeof(vec4) * npix));
  if (gmon) PT_GRP(hipMalloc((void**)&grp->gmon_gather, sizeof(vec4) * npix * p->gmon_buckets));
  else
    for (auto& pd : grp->phys) {
      PT_GRP(hipSetDevice(pd.ordinal));
      PT_GRP(hipMalloc((void**)&pd.scratch, sizeof(vec4) * npix));
    }
#undef PT_GRP
  // RCCL communicator over the distinct devices (once per group)
  if (!gmon && grp->phys.size() > 1 && !grp->comms_ready) {
    std::lock_guard<std::mutex> l(g_rccl_mutex);
    std::string why;
    if (!g_rccl.load(&why)) { fail(PT_ERR_UNSUPPORTED, "device group: " + why); return group_abandon(grp, PT_ERR_UNSUPPORTED); }
    std::vector<int> devs;
    for (auto& pd : grp->phys) devs.push_back(pd.ordinal);
    std::vector<Rccl::comm_t> comms(devs.size(), nullptr);
    const ncclResult_t e = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) { fail(PT_ERR_HIP, std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(e)); return group_abandon(grp, PT_ERR_HIP); }
    for (size_t d = 0; d < devs.size(); d++) grp->phys[d].comm = comms[d];
    grp->comms_ready = true;
  }
  front->params = *p;
  front->S.width = p->width; front->S.height = p->height;
  grp->started = true;
  grp->dirty = true;
  grp->render_start = std::chrono::steady_clock::now();
  grp->timer_ms = 0;
  return PT_OK;
}

int group_render_step(pt_renderer* front, uint32_t max_spp) {
  DeviceGroup* grp = front->group;
  if (!grp->started) return fail(PT_ERR_BAD_STATE, "pt_render_step before pt_start_render");
  const size_t N = grp->shards.size();
  const uint32_t share = max_spp == 0 ? 0 : (uint32_t)((max_spp + N - 1) / N);  // 0 = all remaining
  grp->dirty = true;
  const int rc = grp->for_all([grp, share](size_t g) {
    pt_renderer* m = grp->shards[g];
    if (!m->started) return (int)PT_OK;
    return dev_render_step(m, share);
  });
  grp->timer_ms = (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - grp->render_start).count();
  return rc;
}

// The merge.  Called with every member idle (after dev_wait).
int group_merge(pt_renderer* front) {
  DeviceGroup* grp = front->group;
  const size_t N = grp->shards.size();
  const pt_render_params& p = grp->params;
  const uint32_t npix = p.width * p.height;
  pt_renderer* m0 = grp->shards[0];
  uint64_t done = 0;
  for (size_t g = 0; g < N; g++) if (grp->shards[g]->started) done += grp->shards[g]->accumulated;
  if (done == 0) return PT_OK;
  if (!grp->merged) return fail(PT_ERR_BAD_STATE, "device group: no render in progress");
  if (p.flags & PT_FLAG_GMON) {
    // gather the bucket means every member has touched so far, in bucket order, then resolve on device 0 (gmon.metal:14-55)
    const uint32_t B = p.gmon_buckets, spb = (p.spp + B - 1) / B;
    uint32_t nb = 0;
    PT_HIP(hipSetDevice(m0->device));
    for (size_t g = 0; g < N; g++) {
      pt_renderer* m = grp->shards[g];
      if (!m->started || m->accumulated == 0) continue;
      const uint32_t touched = (uint32_t)((m->accumulated + spb - 1) / spb);
      PT_HIP(hipMemcpyPeerAsync(grp->gmon_gather + (size_t)nb * npix, m0->device, m->gmon_buckets_d.p, m->device, sizeof(vec4) * (size_t)npix * touched, m0->stream));
      nb += touched;
    }
    launch_gmon(m0->stream, grp->merged, grp->gmon_gather, npix, nb, m0->gmon_cap);
    PT_HIP(hipGetLastError());
    PT_HIP(hipStreamSynchronize(m0->stream));
    return PT_OK;
  }
  // per physical device: scratch = sum over its members of (samples_g / samples) * acc_g
  const bool equal = [&] { for (size_t g = 0; g < N; g++) if (grp->shards[g]->accumulated != grp->shards[0]->accumulated) return false; return true; }();
  const bool single = grp->phys.size() == 1;
  for (auto& pd : grp->phys) {
    PT_HIP(hipSetDevice(pd.ordinal));
    bool first = true;
    for (size_t k = 0; k < pd.members.size(); k++) {
      pt_renderer* m = grp->shards[pd.members[k]];
      if (!m->started || m->accumulated == 0) continue;
      // equal shares: plain sum now, one multiplication by 1/N at the end (all-reduce(sum) + 1/N)
      const float w = equal ? 1.0f : (float)((double)m->accumulated / (double)done);
      launch_weighted_add(pd.stream, pd.scratch, m->acc, w, npix, first, false);
      first = false;
    }
    if (first) PT_HIP(hipMemsetAsync(pd.scratch, 0, sizeof(vec4) * npix, pd.stream));
    PT_HIP(hipGetLastError());
  }
  if (!single) {
    for (auto& pd : grp->phys) { PT_HIP(hipSetDevice(pd.ordinal)); PT_HIP(hipStreamSynchronize(pd.stream)); }
    // the single RCCL reduction of the float accumulation buffer over xGMI
    ncclResult_t e = g_rccl.GroupStart();
    for (auto& pd : grp->phys) {
      if (e != ncclSuccess) break;
      PT_HIP(hipSetDevice(pd.ordinal));
      e = g_rccl.AllReduce(pd.scratch, pd.scratch, (size_t)npix * 4, Rccl::kFloat32, Rccl::kSum, pd.comm, pd.stream);
    }
    const ncclResult_t e2 = g_rccl.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) return fail(PT_ERR_HIP, std::string("ncclAllReduce failed: ") + g_rccl.GetErrorString(e != ncclSuccess ? e : e2));
  }
  PT_HIP(hipSetDevice(m0->device));
  uint32_t active = 0;
  for (size_t g = 0; g < N; g++) if (grp->shards[g]->started && grp->shards[g]->accumulated) active++;
  launch_weighted_add(grp->phys[0].stream, grp->merged, grp->phys[0].scratch, equal ? 1.0f / (float)active : 1.0f, npix, true, true);
  PT_HIP(hipGetLastError());
  for (auto& pd : grp->phys) { PT_HIP(hipSetDevice(pd.ordinal)); PT_HIP(hipStreamSynchronize(pd.stream)); }
  PT_HIP(hipSetDevice(m0->device));
  return PT_OK;
}

int group_wait(pt_renderer* front) {
  DeviceGroup* grp = front->group;
  int rc = grp->for_all([grp](size_t g) { return grp->shards[g]->started ? dev_wait(grp->shards[g]) : (int)PT_OK; });
  if (rc != PT_OK) return rc;
  if (grp->started && grp->dirty) {
    rc = group_merge(front);
    if (rc != PT_OK) return rc;
    grp->dirty = false;
  }
  if (grp->started)
    grp->timer_ms = (uint64_t)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - grp->render_start).count();
  return PT_OK;
}

int group_get_stats(pt_renderer* front, pt_stats* out) {
  DeviceGroup* grp = front->group;
  if (!grp->started) return fail(PT_ERR_BAD_STATE, "no render started");
  int rc = group_wait(front);
  if (rc != PT_OK) return rc;
  memset(out, 0, sizeof(*out));
  bool have = false;
  for (auto* m : grp->shards) {
    if (!m->started) continue;
    pt_stats s;
    rc = dev_get_stats(m, &s);
    if (rc != PT_OK) return rc;
    if (!have) { *out = s; have = true; continue; }
    out->closest_rays += s.closest_rays; out->shadow_rays += s.shadow_rays; out->shaded_hits += s.shaded_hits; out->paths += s.paths;
    out->nonfinite_samples += s.nonfinite_samples;
    // device time per kernel class: the slowest member (they run concurrently)
    out->ms_raygen = std::max(out->ms_raygen, s.ms_raygen); out->ms_closest = std::max(out->ms_closest, s.ms_closest);
    out->ms_shade = std::max(out->ms_shade, s.ms_shade); out->ms_shadow = std::max(out->ms_shadow, s.ms_shadow);
    out->ms_accumulate = std::max(out->ms_accumulate, s.ms_accumulate);
    out->launches_closest += s.launches_closest; out->launches_shadow += s.launches_shadow;
    out->upload_ms = std::max(out->upload_ms, s.upload_ms); out->bvh_build_ms = std::max(out->bvh_build_ms, s.bvh_build_ms);
  }
  return PT_OK;
}

pt_renderer* first_started(DeviceGroup* grp) {
  for (auto* m : grp->shards) if (m->started) return m;
  return nullptr;
}

}  // namespace

// ---- the C ABI (include/ptamd.h) ---------------------------------------------------------------------------------------
extern "C" {

int pt_group_partition(uint32_t spp, uint32_t members, int32_t flags, uint32_t gmon_buckets, uint64_t* first, uint64_t* count,
                       uint32_t* bucket0, uint32_t* bucket1) {
  if (!first || !count || members == 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_group_partition: bad argument");
  if (flags & PT_FLAG_GMON) {
    // whole buckets per member: bucket b holds the samples [b * spb, (b + 1) * spb) (renderer_pt.cpp:124-126)
    if (gmon_buckets < 1 || gmon_buckets > 32) return fail(PT_ERR_INVALID_ARGUMENT, "gmon_buckets must be 1..32 (gmon.metal:12 maxBuckets)");
    const uint32_t B = gmon_buckets, spb = (spp + B - 1) / B;
    // B / members buckets each, the remainder to the LOWEST-numbered members: member 0 — whose device resolves, post-processes and
    // presents the image — owns a bucket whenever anybody does (5..7 buckets on 8 devices: members 0..B-1 get one each)
    const uint32_t per = B / members, extra = B % members;
    for (uint32_t g = 0; g < members; g++) {
      const uint32_t b0 = g * per + std::min(g, extra), b1 = b0 + per + (g < extra ? 1u : 0u);
      const uint64_t s0 = std::min<uint64_t>((uint64_t)b0 * spb, spp), s1 = std::min<uint64_t>((uint64_t)b1 * spb, spp);
      if (bucket0) bucket0[g] = b0;
      if (bucket1) bucket1[g] = b1;
      first[g] = s0; count[g] = s1 - s0;
    }
  } else {
    uint64_t at = 0;
    for (uint32_t g = 0; g < members; g++) {
      const uint64_t c = spp / members + (g < spp % members ? 1 : 0);
      first[g] = at; count[g] = c;
      if (bucket0) bucket0[g] = 0;
      if (bucket1) bucket1[g] = 0;
      at += c;
    }
  }
  return PT_OK;
}

int pt_rccl_probe(void) {
  std::lock_guard<std::mutex> l(g_rccl_mutex);
  std::string why;
  if (!g_rccl.load(&why)) return fail(PT_ERR_UNSUPPORTED, why);
  return PT_OK;
}

// The merge's RCCL calls on ONE device: communicator of one rank, a grouped in-place all-reduce(sum, f32) of a known pattern, result
// checked, communicator destroyed.  What a one-GPU box can show of the distinct-device path (the calls, their argument types and
// enumerators, stream ordering); the exchange over xGMI itself needs two GPUs.
int pt_rccl_selftest(int32_t device_ordinal) {
  std::lock_guard<std::mutex> l(g_rccl_mutex);
  std::string why;
  if (!g_rccl.load(&why)) return fail(PT_ERR_UNSUPPORTED, why);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_ordinal < 0 || device_ordinal >= ndev) return fail(PT_ERR_NO_DEVICE, "pt_rccl_selftest: no such HIP device");
  PT_HIP(hipSetDevice(device_ordinal));
  const size_t n = 4 * 1920 * 8;  // a few rows of a 1080p accumulator
  std::vector<float> h(n);
  for (size_t i = 0; i < n; i++) h[i] = (float)(i % 977) * 0.25f;
  float* d = nullptr;
  hipStream_t s = nullptr;
  Rccl::comm_t comm = nullptr;
  int rc = PT_OK;
  do {
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc = fail(PT_ERR_HIP, "pt_rccl_selftest: allocation failed"); break; }
    if (hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(PT_ERR_HIP, "pt_rccl_selftest: upload failed"); break; }
    const int dev = device_ordinal;
    ncclResult_t e = g_rccl.CommInitAll(&comm, 1, &dev);
    if (e != ncclSuccess) { rc = fail(PT_ERR_HIP, std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(e)); break; }
    e = g_rccl.GroupStart();
    if (e == ncclSuccess) e = g_rccl.AllReduce(d, d, n, Rccl::kFloat32, Rccl::kSum, comm, s);
    const ncclResult_t e2 = g_rccl.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) { rc = fail(PT_ERR_HIP, std::string("ncclAllReduce failed: ") + g_rccl.GetErrorString(e != ncclSuccess ? e : e2)); break; }
    if (hipStreamSynchronize(s) != hipSuccess) { rc = fail(PT_ERR_HIP, "pt_rccl_selftest: stream synchronisation failed"); break; }
    std::vector<float> back(n);
    if (hipMemcpy(back.data(), d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(PT_ERR_HIP, "pt_rccl_selftest: readback failed"); break; }
    for (size_t i = 0; i < n; i++) if (back[i] != h[i]) { rc = fail(PT_ERR_HIP, "pt_rccl_selftest: a one-rank sum changed the data"); break; }
  } while (0);
  if (comm) (void)g_rccl.CommDestroy(comm);
  if (s) (void)hipStreamDestroy(s);
  if (d) (void)hipFree(d);
  return rc;
}

int pt_create(const pt_create_info* info, pt_renderer** out) {
  if (!info || !out) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: null argument");
  *out = nullptr;
  if (info->abi_version != PT_ABI_VERSION) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: ABI version mismatch");
  if (info->device_count == 0) return dev_create(info, info->device_ordinal, out);
  if (!info->device_ordinals) return fail(PT_ERR_INVALID_ARGUMENT, "pt_create: device_count without device_ordinals");
  if (info->device_count == 1) return dev_create(info, info->device_ordinals[0], out);
  return group_create(info, out);
}

void pt_destroy(pt_renderer* r) {
  if (!r) return;
  if (is_group(r)) group_destroy(r); else dev_destroy(r);
}

int pt_start_render(pt_renderer* r, const pt_scene_snapshot* scene, const pt_render_params* p) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "pt_start_render: null argument");
  return is_group(r) ? group_start_render(r, scene, p) : dev_start_render(r, scene, p);
}

int pt_render_step(pt_renderer* r, uint32_t max_spp) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null renderer");
  return is_group(r) ? group_render_step(r, max_spp) : dev_render_step(r, max_spp);
}

int pt_wait(pt_renderer* r) {
  if (!r) return fail(PT_ERR_INVALID_ARGUMENT, "null renderer");
  return is_group(r) ? group_wait(r) : dev_wait(r);
}

int pt_status(const pt_renderer* r) {  // renderer_pt.cpp:1023-1031
  if (!is_group(r)) return dev_status(r);
  const DeviceGroup* grp = r->group;
  if (!grp->started) return PT_STATUS_READY;
  for (auto* m : grp->shards) if (m->started && m->accumulated < m->total) return PT_STATUS_BUSY;
  return PT_STATUS_READY | PT_STATUS_DONE;
}

int pt_progress(const pt_renderer* r, uint64_t* accumulated, uint64_t* total) {
  if (!is_group(r)) return dev_progress(r, accumulated, total);
  if (!accumulated || !total) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  *accumulated = 0;
  for (auto* m : r->group->shards) if (m->started) *accumulated += m->accumulated;
  *total = r->group->started ? r->group->params.spp : 0;
  return PT_OK;
}

uint64_t pt_render_time_ms(const pt_renderer* r) { return is_group(r) ? r->group->timer_ms : dev_render_time_ms(r); }

int pt_read_accumulator(pt_renderer* r, float* rgba_out) {
  if (!is_group(r)) return dev_read_accumulator(r, rgba_out);
  if (!rgba_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGroup* grp = r->group;
  if (!grp->started) return fail(PT_ERR_BAD_STATE, "pt_read_accumulator before pt_start_render");
  int rc = group_wait(r);
  if (rc != PT_OK) return rc;
  PT_HIP(hipSetDevice(grp->shards[0]->device));
  PT_HIP(hipMemcpy(rgba_out, grp->merged, sizeof(vec4) * (size_t)grp->params.width * grp->params.height, hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_set_post_options(pt_renderer* r, const pt_post_options* o) {
  if (!is_group(r)) return dev_set_post_options(r, o);
  for (auto* m : r->group->shards) { int rc = dev_set_post_options(m, o); if (rc != PT_OK) return rc; }
  return PT_OK;
}

int pt_set_tonemap_options(pt_renderer* r, const pt_tonemap_options* o) {
  if (!is_group(r)) return dev_set_tonemap_options(r, o);
  for (auto* m : r->group->shards) { int rc = dev_set_tonemap_options(m, o); if (rc != PT_OK) return rc; }
  return PT_OK;
}

int pt_read_render_target(pt_renderer* r, uint8_t* rgba8_out) {
  if (!is_group(r)) return dev_read_render_target(r, rgba8_out);
  if (!rgba8_out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  DeviceGroup* grp = r->group;
  if (!grp->started) return fail(PT_ERR_BAD_STATE, "pt_read_render_target before pt_start_render");
  int rc = group_wait(r);
  if (rc != PT_OK) return rc;
  pt_renderer* m = first_started(grp);
  if (!m || m->device != grp->shards[0]->device) return fail(PT_ERR_BAD_STATE, "device group: no member on the first device has samples");
  return dev_postprocess_to_host(m, grp->merged, rgba8_out);
}

int pt_present_render_target(pt_renderer* r, void** device_rgba8_out, void** stream_out) {
  if (!is_group(r)) return dev_present(r, nullptr, device_rgba8_out, stream_out);
  DeviceGroup* grp = r->group;
  if (!grp->started) return fail(PT_ERR_BAD_STATE, "pt_present_render_target before pt_start_render");
  int rc = group_wait(r);  // merges the members' accumulators
  if (rc != PT_OK) return rc;
  pt_renderer* m = first_started(grp);
  if (!m || m->device != grp->shards[0]->device) return fail(PT_ERR_BAD_STATE, "device group: no member on the first device has samples");
  return dev_present(m, grp->merged, device_rgba8_out, stream_out);
}

int pt_set_gmon_options(pt_renderer* r, const pt_gmon_options* o) {
  if (!is_group(r)) return dev_set_gmon_options(r, o);
  for (auto* m : r->group->shards) { int rc = dev_set_gmon_options(m, o); if (rc != PT_OK) return rc; }
  r->group->dirty = true;
  return PT_OK;
}

int pt_read_gmon_bucket(pt_renderer* r, uint32_t bucket, float* rgba_out) {
  if (!is_group(r)) return dev_read_gmon_bucket(r, bucket, rgba_out);
  DeviceGroup* grp = r->group;
  if (!grp->started || !(grp->params.flags & PT_FLAG_GMON)) return fail(PT_ERR_BAD_STATE, "no such GMoN bucket");
  for (size_t g = 0; g < grp->shards.size(); g++)
    if (grp->shards[g]->started && bucket >= grp->bucket0[g] && bucket < grp->bucket1[g]) return dev_read_gmon_bucket(grp->shards[g], bucket, rgba_out);
  return fail(PT_ERR_BAD_STATE, "no such GMoN bucket");
}

void* pt_accumulator_device_ptr(pt_renderer* r) {
  if (!is_group(r)) return dev_accumulator_device_ptr(r);
  return r->group->started ? (void*)r->group->merged : nullptr;
}

int pt_get_constants(const pt_renderer* r, pt_constants* out) {
  if (!is_group(r)) return dev_get_constants(r, out);
  pt_renderer* m = first_started(r->group);
  if (!m) return fail(PT_ERR_BAD_STATE, "no render started");
  int rc = dev_get_constants(m, out);
  if (rc == PT_OK) out->spp = r->group->params.spp;
  return rc;
}

int pt_get_lights(const pt_renderer* r, pt_area_light* out, uint32_t capacity, uint32_t* count) {
  if (!is_group(r)) return dev_get_lights(r, out, capacity, count);
  pt_renderer* m = first_started(r->group);
  return m ? dev_get_lights(m, out, capacity, count) : fail(PT_ERR_BAD_STATE, "no render started");
}

int pt_get_env_alias(const pt_renderer* r, pt_alias_entry* out, uint64_t capacity, uint64_t* count) {
  if (!is_group(r)) return dev_get_env_alias(r, out, capacity, count);
  pt_renderer* m = first_started(r->group);
  return m ? dev_get_env_alias(m, out, capacity, count) : fail(PT_ERR_BAD_STATE, "no render started");
}

int pt_trace_primary(pt_renderer* r, uint32_t sample_idx, pt_hit_record* out) {
  if (!is_group(r)) return dev_trace_primary(r, sample_idx, out);
  pt_renderer* m = first_started(r->group);
  return m ? dev_trace_primary(m, sample_idx, out) : fail(PT_ERR_BAD_STATE, "no render started");
}

int pt_debug_sample(pt_renderer* r, uint32_t sample_idx, float* radiance_out, int32_t* hits_out) {
  if (!is_group(r)) return dev_debug_sample(r, sample_idx, radiance_out, hits_out);
  pt_renderer* m = first_started(r->group);
  return m ? dev_debug_sample(m, sample_idx, radiance_out, hits_out) : fail(PT_ERR_BAD_STATE, "no render started");
}

int pt_measure_traversal(pt_renderer* r, uint32_t sample_idx) {
  if (!is_group(r)) return dev_measure_traversal(r, sample_idx);
  pt_renderer* m = first_started(r->group);
  return m ? dev_measure_traversal(m, sample_idx) : fail(PT_ERR_BAD_STATE, "no render started");
}

int pt_set_profiling(pt_renderer* r, int enabled) {
  if (!is_group(r)) return dev_set_profiling(r, enabled);
  for (auto* m : r->group->shards) { int rc = dev_set_profiling(m, enabled); if (rc != PT_OK) return rc; }
  return PT_OK;
}

int pt_get_stats(pt_renderer* r, pt_stats* out) {
  if (!r || !out) return fail(PT_ERR_INVALID_ARGUMENT, "null argument");
  return is_group(r) ? group_get_stats(r, out) : dev_get_stats(r, out);
}

}  // extern "C"
