// kernels.h — host-visible launchers of the wavefront kernels (kernels.hip) and the LBVH builder (lbvh.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pt_device.h"

namespace pt {

constexpr int kBlock = 256;  // 4 waves; traversal kernels keep a 24 KiB LDS stack slab per block

void launch_raygen(hipStream_t s, const DeviceScene& S, PathState st, vec4* Lbuf, BatchCounters* ctr, uint32_t first_sample,
                   uint32_t nsamples);
void launch_trace_closest(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState st, vec4* hit, BatchCounters* ctr,
                          uint32_t bounce, uint32_t* spill, int32_t* hitlog, uint32_t log_stride, bool count, uint32_t refill);
void launch_shade(hipStream_t s, uint32_t grid, const DeviceScene& S, PathState sin, PathState sout, const vec4* hit,
                  ShadowQueue sq, vec4* Lbuf, BatchCounters* ctr, uint32_t bounce);
void launch_trace_shadow(hipStream_t s, uint32_t grid, const DeviceScene& S, ShadowQueue sq, vec4* Lbuf, BatchCounters* ctr,
                         uint32_t bounce, uint32_t* spill, bool count, uint32_t refill);
void launch_accumulate(hipStream_t s, vec4* acc, const vec4* Lbuf, uint32_t npixels, uint32_t nsamples, uint32_t n0,
                       uint32_t nonfinite_policy, BatchCounters* ctr);
void launch_fold_counters(hipStream_t s, const BatchCounters* ctr, Totals* tot, uint32_t max_bounces, bool counted);
void launch_hit_records(hipStream_t s, const DeviceScene& S, PathState st, const vec4* hit, const BatchCounters* ctr,
                        pt_hit_record* out, uint32_t npixels);

// ---- LBVH (lbvh.hip) ----
struct LbvhResult {
  BvhNode* nodes = nullptr;   // indexed like the binary radix tree (tri_count - 1 slots, even-depth ones used)
  TriRec* tris = nullptr;     // tri_count records in leaf (Morton) order
  uint32_t root_ref = kInvalidRef;
  uint32_t node_count = 0;    // 4-wide nodes emitted
  uint32_t max_depth = 0;     // of the binary tree
};
// Flattens the instanced scene to world-space triangles and builds the BVH entirely on the device.
// `S` needs positions / indices / meshes / instances filled in. Returns hipSuccess or the failing HIP error.
hipError_t build_lbvh(hipStream_t s, const DeviceScene& S, uint32_t instance_count, uint32_t tri_count, LbvhResult* out);

}  // namespace pt
