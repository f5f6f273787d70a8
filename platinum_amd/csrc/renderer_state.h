// renderer_state.h — private state of one `pt_renderer` (shared by renderer.hip: the single-device driver behind the C ABI,
// and multi_device.hip: the device group that shards samples over several of them).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "host_scene.h"
#include "kernels.h"
#include "pt_bvh.h"

using namespace pt;

int pt_fail(int code, const std::string& msg);          // records the message for pt_last_error() (thread-local), returns code
const std::string& pt_last_error_string();
inline int fail(int code, const std::string& msg) { return pt_fail(code, msg); }

namespace {

#define PT_HIP(call)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) {                                                                                \
      char buf_[512];                                                                                      \
      snprintf(buf_, sizeof(buf_), "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return fail(e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP, buf_);                     \
    }                                                                                                      \
  } while (0)

// A device array that KEEPS its allocation across pt_start_render calls: the frontend restarts the render on every camera or scene
// edit (renderer_pt.cpp:199-217 makes startRender cheap), and releasing + re-allocating the ~53 GB of path queues took 1.3-1.5 s per
// restart on MI355X (hipFree of multi-GB buffers unmaps them; measured r03) against 12 ms for everything else.  alloc() only goes back
// to the driver when the array has to GROW; the memory is returned in pt_destroy.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;    // elements in use
  size_t cap = 0;  // elements allocated
  hipError_t alloc(size_t count) {
    if (count <= cap) { n = count; return hipSuccess; }
    release();
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * count);
    if (e != hipSuccess) { p = nullptr; return e; }
    n = cap = count;
    return hipSuccess;
  }
  size_t bytes_held() const { return cap * sizeof(T); }
  hipError_t upload(const std::vector<T>& v) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = cap = 0;
  }
  ~DevBuf() { release(); }
};

enum KernelClass { K_RAYGEN = 0, K_CLOSEST, K_SHADE, K_SHADOW, K_ACCUM, K_CLASSES };

struct TimedLaunch { int cls; hipEvent_t start, stop; };

}  // namespace

struct pt_renderer {
  int device = 0;
  int num_cu = 256;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;

  // create-time tables
  DevBuf<float> lut_data;
  LutSet luts{};
  uint32_t lut_w_E = 0, lut_w_Eavg = 0;
  DevBuf<HaltonEntry> halton;

  // scene (valid after pt_start_render)
  bool started = false;
  DevBuf<pt_float3> positions;
  DevBuf<pt_vertex_data> vdata;
  DevBuf<uint32_t> indices, slots;
  DevBuf<MeshInfo> meshes;
  DevBuf<InstanceInfo> instances;
  DevBuf<pt_material_gpu> materials;
  DevBuf<pt_area_light> lights_d;
  std::vector<pt_area_light> lights;
  DevBuf<DeviceScene> scene_d;
  DevBuf<ShadeRec> shade_recs;
  DevBuf<LightRec> light_recs;
  DevBuf<float> light_cdf;
  DevBuf<uint32_t> prim_tri_d, mesh_prim_base_d, inst_prim_base_d;  // leaf-slot grouping of the one-BVH structure (host_scene.h build_primitives)
  DevBuf<InstanceTrav> inst_trav;   // two-level structure only
  bool two_level = false;
  int two_level_override = -1;      // $PTAMD_TWO_LEVEL: 0 / 1 force the choice, -1 = by instancing factor
  DevBuf<uint8_t> tex_data;           // all textures in their own formats (host_scene.h decode_textures)
  DevBuf<float> tex_decode;           // the 8-bit decode tables
  DevBuf<TexInfo> textures;
  DevBuf<pt_alias_entry> env_alias_d;
  std::vector<pt_alias_entry> env_alias;
  LbvhResult bvh{};
  LbvhScratch bvh_scratch{};  // the builder's temporaries, kept between builds (release_all gives them back)
  DeviceScene S{};
  pt_render_params params{};
  pt_constants constants{};
  uint32_t instance_count = 0, tri_count = 0, slot_count = 0;

  // wavefront buffers
  uint32_t samples_in_flight = 0;
  size_t capacity = 0;  // path slots
  DevBuf<vec4> st_rayO[2], st_rayD[2], st_att[2], hit, sq_o, sq_d, sq_c, Lbuf, acc_own;
  DevBuf<uint32_t> spill, seg_active[2], seg_shadow, seg_poison;
  DevBuf<WaveStats> wave_stats;
  DevBuf<uint32_t> chunk_table[2];
  DevBuf<uint32_t> shade_order, shade_cost;
  DevBuf<vec4> gmon_buckets_d;  // [bucket][pixel] with PT_FLAG_GMON (renderer_pt.cpp:824-830)
  float gmon_cap = 1.0f;        // GmonOptions.cap (pt_shader_defs.hpp:164-166)
  pt_post_options post{};
  pt_tonemap_options tonemap{};
  DevBuf<uint32_t> render_target;  // RGBA8 (renderer_pt.cpp:832-835)
  uint32_t closest_grid = 0, shadow_grid = 0, closest_blocks_per_cu = PT_CLOSEST_WAVES, shadow_blocks_per_cu = PT_SHADOW_WAVES;  // persistent trace grids, each sized for its kernel's occupancy
  uint32_t last_batch_ns = 0, last_batch_first = 0;  // the batch Lbuf holds ($PTAMD_DEBUG_PIXEL)
  uint32_t nseg = 0, tiles_per_seg = 1, seg_bands = 4, tiles_per_seg_override = 0, nstats = 0, seg_cap = 0, blocks_per_cu = 6, shade_grid = 0, refill_threshold = 48;
  DevBuf<BatchCounters> ctr;
  DevBuf<Totals> totals;
  vec4* acc = nullptr;
  uint32_t grid = 0;

  // ---- member of a device group (multi_device.hip): where this renderer's samples sit in the whole render ----
  uint32_t gmon_sample_base = 0;   // index of its first sample relative to the render's first sample (GMoN bucket + weight use the global index)
  uint32_t gmon_total_spp = 0;     // spp of the whole render (bucket size = ceil(spp / buckets)); 0 = params.spp
  uint32_t gmon_bucket_base = 0;   // first bucket it owns; its bucket array starts there
  uint32_t gmon_own_buckets = 0;   // buckets it owns; 0 = params.gmon_buckets
  struct DeviceGroup* group = nullptr;  // non-null on the front object of a device group

  // progress (renderer_pt.hpp:168-171).  `accumulated` counts the samples pt_render_step has ACCEPTED (what the reference's
  // m_accumulatedFrames counts: encoded, not finished); `launched` of them are enqueued on the stream, the rest are pending:
  // render() calls that arrive while the GPU is still busy are merged into one batch (renderer.hip flush_pending).
  uint64_t accumulated = 0, total = 0, launched = 0;
  uint32_t batches = 0;                 // batches enqueued since pt_start_render
  hipEvent_t batch_done = nullptr;      // recorded behind the newest batch
  bool batch_done_valid = false;
  std::chrono::steady_clock::time_point render_start;
  uint64_t timer_ms = 0;

  // measurement
  bool profiling = false;
  std::vector<TimedLaunch> timed;
  double ms_class[K_CLASSES] = {0, 0, 0, 0, 0};
  uint64_t launches[K_CLASSES] = {0, 0, 0, 0, 0};
  double upload_ms = 0, bvh_ms = 0;

  PathState path_state(int k) { return PathState{st_rayO[k].p, st_rayD[k].p, st_att[k].p}; }
  ShadowQueue shadow_queue() { return ShadowQueue{sq_o.p, sq_d.p, sq_c.p}; }
  Segments segments() { return Segments{{seg_active[0].p, seg_active[1].p}, seg_shadow.p, seg_poison.p, wave_stats.p, chunk_table[0].p, chunk_table[1].p, shade_order.p, shade_cost.p, seg_cap, nseg, tiles_per_seg, /*nsamples: set per batch*/ 0u, seg_bands, nstats, refill_threshold}; }

  // A restart keeps every device array (they are re-filled, and only re-allocated when they must grow); what it drops is the
  // acceleration structure of the previous scene and the "started" state.  release_all() returns the memory (pt_destroy).
  void free_scene() {
    if (bvh.nodes) (void)hipFree(bvh.nodes);
    if (bvh.tris) (void)hipFree(bvh.tris);
    if (bvh.mesh_trav) (void)hipFree(bvh.mesh_trav);
    bvh = LbvhResult{};
    acc = nullptr;
    started = false;
    last_batch_ns = 0;
  }
  // bytes of path-queue memory this renderer already holds (reused by the next render: they count as free when the batch is sized)
  size_t queue_bytes_held() const {
    size_t b = hit.bytes_held() + sq_o.bytes_held() + sq_d.bytes_held() + sq_c.bytes_held() + Lbuf.bytes_held() + chunk_table[0].bytes_held() + chunk_table[1].bytes_held();
    for (int k = 0; k < 2; k++) b += st_rayO[k].bytes_held() + st_rayD[k].bytes_held() + st_att[k].bytes_held();
    return b;
  }
  void release_all() {
    free_scene();
    positions.release(); vdata.release(); indices.release(); slots.release(); meshes.release(); instances.release();
    materials.release(); lights_d.release(); tex_data.release(); tex_decode.release(); textures.release(); env_alias_d.release(); scene_d.release(); shade_recs.release(); light_recs.release(); light_cdf.release();
    inst_trav.release(); prim_tri_d.release(); mesh_prim_base_d.release(); inst_prim_base_d.release();
    bvh_scratch.release();
    for (int k = 0; k < 2; k++) { st_rayO[k].release(); st_rayD[k].release(); st_att[k].release(); }
    seg_active[0].release(); seg_active[1].release(); seg_shadow.release(); seg_poison.release(); wave_stats.release(); chunk_table[0].release(); chunk_table[1].release(); shade_order.release(); shade_cost.release(); gmon_buckets_d.release(); render_target.release();
    hit.release(); sq_o.release(); sq_d.release(); sq_c.release(); Lbuf.release(); acc_own.release(); spill.release();
  }
  void drop_timed() {
    for (auto& t : timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    timed.clear();
  }
};


// ---- the single-device driver (renderer.hip); the extern "C" entry points (multi_device.hip) dispatch here or to the group ----
int dev_create(const pt_create_info* info, int device_ordinal, pt_renderer** out);
void dev_destroy(pt_renderer* r);
int dev_start_render(pt_renderer* r, const pt_scene_snapshot* scene, const pt_render_params* p);
int dev_render_step(pt_renderer* r, uint32_t max_spp);
int dev_wait(pt_renderer* r);
int dev_status(const pt_renderer* r);
int dev_progress(const pt_renderer* r, uint64_t* accumulated, uint64_t* total);
uint64_t dev_render_time_ms(const pt_renderer* r);
int dev_read_accumulator(pt_renderer* r, float* rgba_out);
int dev_set_post_options(pt_renderer* r, const pt_post_options* o);
int dev_set_tonemap_options(pt_renderer* r, const pt_tonemap_options* o);
int dev_read_render_target(pt_renderer* r, uint8_t* rgba8_out);
int dev_postprocess_to_host(pt_renderer* r, const vec4* acc_device, uint8_t* rgba8_out);  // the tail of read_render_target on a given image
int dev_present(pt_renderer* r, const vec4* acc_device, void** device_rgba8_out, void** stream_out);  // enqueue only; acc_device NULL = own accumulator
int dev_set_gmon_options(pt_renderer* r, const pt_gmon_options* o);
int dev_read_gmon_bucket(pt_renderer* r, uint32_t bucket, float* rgba_out);
void* dev_accumulator_device_ptr(pt_renderer* r);
int dev_get_constants(const pt_renderer* r, pt_constants* out);
int dev_get_lights(const pt_renderer* r, pt_area_light* out, uint32_t capacity, uint32_t* count);
int dev_get_env_alias(const pt_renderer* r, pt_alias_entry* out, uint64_t capacity, uint64_t* count);
int dev_trace_primary(pt_renderer* r, uint32_t sample_idx, pt_hit_record* out);
int dev_debug_sample(pt_renderer* r, uint32_t sample_idx, float* radiance_out, int32_t* hits_out);
int dev_measure_traversal(pt_renderer* r, uint32_t sample_idx);
int dev_set_profiling(pt_renderer* r, int enabled);
int dev_get_stats(pt_renderer* r, pt_stats* out);
