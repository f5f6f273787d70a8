// kernels.h — host-visible launchers of the wavefront kernels (kernels.hip) and the LBVH builder (lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_device.h"
#include "pt_post.h"

namespace pt {

constexpr int kBlock = 256;  // 4 waves; traversal kernels keep a 16 KiB LDS stack slab per block

struct WaveStats { unsigned long long closest, shadow, shaded, paths; };  // per physical wave, owner-updated, reduced by k_fold_counters

// Queue segments (kernels.hip): segment s = the queue share of tiles [s * tiles_per_seg, (s + 1) * tiles_per_seg) under all
// samples of a batch; it owns `seg_cap` slots of every queue array (interleaved in groups of 16 chunks) and one count per queue.
struct Segments {
  uint32_t* active[2];  // [state buffer][segment] live paths in the segment
  uint32_t* shadow;     // [segment] shadow rays in the segment
  uint32_t* poison;     // [segment] != 0: a path of the segment carries a throughput that is not finite (k_shade's scan of the misses)
  WaveStats* stats;     // [nstats] per physical wave of the producer kernels
  uint32_t* table_closest;  // dense lists of non-empty chunks, (k << 16) | segment, rebuilt by k_chunk_tables
  uint32_t* table_shadow;
  uint32_t* shade_order;    // [nseg] the segments by falling size of the closest-hit queue: the order in which k_shade's waves claim them (k_chunk_tables)
  uint32_t* shade_cost;     // [bounce][nseg] 100 MHz ticks k_shade spent on the segment at that bounce in the PREVIOUS batch of this render (0: none yet)
  uint32_t seg_cap;     // slots per segment (a multiple of 64) = tiles_per_seg * samples_in_flight * 64
  uint32_t nseg;        // segments (<= 32768: k_chunk_tables packs the id into 16 bits and scans <= 1024 per block)
  uint32_t tiles_per_seg;
  uint32_t nsamples;      // samples of the current batch (the per-sample radiance buffer holds nsamples * 64 entries per tile)
  uint32_t bands;         // consecutive segments cycle over this many horizontal bands of the image (nseg % bands == 0)
  uint32_t nstats;      // WaveStats slots (>= waves of the largest producer grid)
  uint32_t refill_threshold; // trace kernels: refill a wave's idle lanes when fewer than this many still hold a ray (0 = only when all idle)
};

void launch_raygen(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* Lbuf, Segments seg, BatchCounters* ctr,
                   uint32_t first_sample, uint32_t nsamples);
// After a producer: list the non-empty chunks of the closest-hit queue (state buffer `cur`, consumed at bounce
// `bounce_closest`) and, if do_shadow, of the shadow queue consumed at `bounce_shadow`.
void launch_chunk_tables(hipStream_t s, Segments seg, uint32_t cur, BatchCounters* ctr, uint32_t bounce_closest,
                         uint32_t bounce_shadow, bool do_shadow);
uint32_t seg_group_chunks();
size_t lbuf_index_host(uint32_t tile, uint32_t s, uint32_t nsamples, uint32_t lane);   // kernels.hip lbuf_index, for the host (debug read-back)
size_t lbuf_sample_stride_host();
uint32_t trace_block_threads(bool two_level);
uint32_t trace_blocks_per_cu_two_level();  // 256 (7 blocks per CU) for one BVH, 1024 (one block per CU) for the two-level structure
void launch_trace_closest(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* hit, Segments seg, uint32_t cur,
                          BatchCounters* ctr, uint32_t bounce, uint32_t* spill, int32_t* hitlog, uint32_t log_stride, bool count);
// `grid` blocks of shade_block_threads() threads (a persistent grid: shade_blocks_per_cu() per CU keeps it resident)
uint32_t shade_block_threads();
uint32_t shade_blocks_per_cu();
void launch_shade(hipStream_t s, uint32_t grid, const DeviceScene* S_device, PathState sin, PathState sout, const vec4* hit,
                  ShadowQueue sq, vec4* Lbuf, Segments seg, uint32_t cur, BatchCounters* ctr, uint32_t bounce);
void launch_trace_shadow(hipStream_t s, uint32_t grid, const DeviceScene& S, ShadowQueue sq, vec4* Lbuf, Segments seg,
                         BatchCounters* ctr, uint32_t bounce, uint32_t* spill, bool count);
void launch_accumulate(hipStream_t s, vec4* acc, const vec4* Lbuf, uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                       uint32_t nonfinite_policy, BatchCounters* ctr);
void launch_accumulate_gmon(hipStream_t s, vec4* buckets, const vec4* Lbuf, uint32_t npixels, uint32_t width, uint32_t nsamples, uint32_t n0,
                            uint32_t samples_per_bucket, uint32_t gmon_buckets, uint32_t bucket_base, uint32_t nonfinite_policy, BatchCounters* ctr);
// out[p] = (first ? 0 : out[p]) + w * in[p]  — merging the accumulators of a device group's members (alpha is set to 1 by the last call)
void launch_weighted_add(hipStream_t s, vec4* out, const vec4* in, float w, uint32_t npixels, bool first, bool last);
void launch_gmon(hipStream_t s, vec4* acc, const vec4* buckets, uint32_t npixels, uint32_t nBuckets, float cap);
void launch_postprocess(hipStream_t s, const vec4* acc, uint32_t* rgba8, uint32_t W, uint32_t H, const PostConstants& pc);
void launch_fold_counters(hipStream_t s, const BatchCounters* ctr, Totals* tot, Segments seg, bool counted);
void launch_shade_records(hipStream_t s, const DeviceScene& S, ShadeRec* out);
void launch_light_records(hipStream_t s, const DeviceScene& S, LightRec* out, float* cdf);
void launch_hit_records(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, const vec4* hit, Segments seg,
                        pt_hit_record* out);

// ---- LBVH (lbvh.hip) ----
struct LbvhResult {
  BvhNode* nodes = nullptr;   // node_count records, the root first
  TriRec* tris = nullptr;     // slot_count leaf slots: in leaf order (one BVH: one or two triangles each), in flattening order (two-level: one each)
  uint32_t slot_count = 0;
  MeshTrav* mesh_trav = nullptr;  // two-level only: one record per mesh
  uint32_t root_ref = kInvalidRef;
  uint32_t node_count = 0;    // 4-wide nodes emitted
  uint32_t max_depth = 0;     // of the binary tree
  uint32_t depth4 = 0;        // levels of the tree (the traversal stack needs <= 3 entries per level; 5 when wide6)
  bool wide6 = false;         // nodes[] holds BvhNode6 records (pt_device.h)
};
// The builder's temporaries: ONE device allocation that is kept between builds and only ever grows (a hipFree is a device synchronisation
// and took ~0.2 ms of a 2.2 ms build; C3 needs ~0.4 GB).  Owned by the renderer; release() gives the memory back.
struct LbvhScratch {
  char* base = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes);
  void release();
};
// How the flattened triangles are grouped into leaf slots (host_scene.h build_primitives; device copies).  prim_tri == nullptr: one triangle
// per slot, slot = flattened triangle index.
struct PrimTables { const uint32_t* prim_tri = nullptr; const uint32_t* mesh_prim_base = nullptr; const uint32_t* inst_prim_base = nullptr; uint32_t slot_count = 0; };
// Flattens the instanced scene to world-space triangles and builds the BVH entirely on the device.
// `S` needs positions / indices / meshes / instances filled in. Returns hipSuccess or the failing HIP error.
hipError_t build_lbvh(hipStream_t s, const DeviceScene& S, const PrimTables& prims, uint32_t instance_count, uint32_t stack_capacity,
                      LbvhScratch* scratch, LbvhResult* out);
// The two-level structure: a TLAS over the instances' world boxes + one object-space BLAS per mesh, in one node array
// (depth4 = TLAS levels + deepest BLAS levels; the traversal stack also holds one exit marker).  `meshes` is the host copy
// of S.meshes (mesh_count entries).
hipError_t build_two_level(hipStream_t s, const DeviceScene& S, const MeshInfo* meshes, uint32_t mesh_count, uint32_t instance_count,
                           uint32_t tri_count, uint32_t stack_capacity, LbvhScratch* scratch, LbvhResult* out);

}  // namespace pt
