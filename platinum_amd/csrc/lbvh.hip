// lbvh.hip — GPU LBVH build: replaces Metal's closed acceleration-structure build
// (renderer_pt.cpp:244-294 makeAccelStruct, :653-749 rebuildAccelerationStructures).
//
//   k_flatten   one leaf slot per thread: the world-space corners of one triangle, or of two consecutive triangles of a mesh that share an
//               edge (host_scene.h pair_mesh_triangles), + the slot's AABB (fp32 transformPoint, the intersection contract)
//   k_bounds    scene centroid bounds (wave reduce + ordered-int atomics)
//   k_morton    63-bit Morton code of the AABB centre (21 bits / axis, cubic cells)
//   radix sort  rocPRIM (hipcub::DeviceRadixSort::SortPairs, 64-bit keys)
//   k_karras    Karras 2012 radix tree: one thread per internal node, duplicate codes split by position
//   k_refit     bottom-up AABB union, second arrival at a node proceeds (agent-scope fences around the counter)
//   k_emit      collapse to 4-wide 64-byte nodes (8-bit quantised, inflated child boxes) + triangles in leaf order
//
// Not on the per-sample hot path: runs once per pt_start_render; timed separately (pt_stats.bvh_build_ms).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>


#include "kernels.h"
#include "pt_shade.h"

namespace pt {

namespace {

struct alignas(16) Box { float lo[3]; float hi[3]; float _pad[2]; };

__device__ __forceinline__ int float_to_ordered(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// One thread per leaf slot: the slot's one or two triangles in world space (fp32 transformPoint of the mesh vertices: the intersection
// contract) and the slot's box.  P.prim_tri == nullptr: one triangle per slot, slot = flattened triangle index (two-level structure).
__global__ void __launch_bounds__(256) k_flatten(DeviceScene S, PrimTables P, uint32_t instance_count, TriRec* __restrict__ tris, Box* __restrict__ boxes) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= P.slot_count) return;
  // instance = last i whose first slot is <= g
  uint32_t lo = 0, hi = instance_count;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((P.prim_tri ? P.inst_prim_base[mid] : S.instances[mid].tri_global_base) <= g) lo = mid; else hi = mid;
  }
  const InstanceInfo inst = S.instances[lo];
  const MeshInfo mesh = S.meshes[inst.mesh];
  uint32_t prim = g - inst.tri_global_base;
  bool pair = false;
  if (P.prim_tri) {
    const uint32_t e = P.prim_tri[P.mesh_prim_base[inst.mesh] + (g - P.inst_prim_base[lo])];
    prim = e & 0x7fffffffu;
    pair = (e >> 31) != 0;
  }
  const uint32_t* idx = &S.indices[3 * (size_t)(mesh.tri_base + prim)];
  const Xform X = load_xform(inst);
  const vec3 v0 = transformPoint(ld3(S.positions[mesh.vertex_base + idx[0]]), X);
  const vec3 v1 = transformPoint(ld3(S.positions[mesh.vertex_base + idx[1]]), X);
  const vec3 v2 = transformPoint(ld3(S.positions[mesh.vertex_base + idx[2]]), X);
  vec3 v3_ = v0;
  TriRec t;
  uint32_t code = 0u;
  t._pad = 0u;
  t.gid_a = ((inst.tri_global_base + prim) << 2) | material_class(S.materials[inst.material_base + S.slots[mesh.tri_base + prim]]);
  t.gid_b = kInvalidRef;
  if (pair) {  // triangle prim + 1 shares two vertex indices with this one (pair_mesh_triangles): its corners by index comparison
    for (int k = 0; k < 3; k++) {
      const uint32_t ib = idx[3 + k];
      uint32_t c = 3u;
      if (ib == idx[0]) c = 0u; else if (ib == idx[1]) c = 1u; else if (ib == idx[2]) c = 2u;
      else v3_ = transformPoint(ld3(S.positions[mesh.vertex_base + ib]), X);
      code |= c << (2 * k);
    }
    t.gid_b = ((inst.tri_global_base + prim + 1u) << 2) | material_class(S.materials[inst.material_base + S.slots[mesh.tri_base + prim + 1u]]);
  }
  t.inst_code = lo | (code << kSlotInstBits);
  t.q0[0] = v0.x; t.q0[1] = v0.y; t.q0[2] = v0.z;
  t.q1[0] = v1.x; t.q1[1] = v1.y; t.q1[2] = v1.z;
  t.q2[0] = v2.x; t.q2[1] = v2.y; t.q2[2] = v2.z;
  t.q3[0] = v3_.x; t.q3[1] = v3_.y; t.q3[2] = v3_.z;
  tris[g] = t;
  Box b;
  b.lo[0] = fminf(fminf(v0.x, v3_.x), fminf(v1.x, v2.x)); b.hi[0] = fmaxf(fmaxf(v0.x, v3_.x), fmaxf(v1.x, v2.x));
  b.lo[1] = fminf(fminf(v0.y, v3_.y), fminf(v1.y, v2.y)); b.hi[1] = fmaxf(fmaxf(v0.y, v3_.y), fmaxf(v1.y, v2.y));
  b.lo[2] = fminf(fminf(v0.z, v3_.z), fminf(v1.z, v2.z)); b.hi[2] = fmaxf(fmaxf(v0.z, v3_.z), fmaxf(v1.z, v2.z));
  b._pad[0] = b._pad[1] = 0.0f;
  boxes[g] = b;
}

__global__ void k_init_bounds(int* bounds) {
  if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7fffffff;
  else if (threadIdx.x < 6) bounds[threadIdx.x] = (int)0x80000000;
}

// (grid <= 256 blocks, one set of atomics per BLOCK: with one per wave of 1 024 blocks the 24 k same-address atomics took 280 us of a 2.5 ms build)
__global__ void __launch_bounds__(256) k_bounds(const Box* __restrict__ boxes, uint32_t n, int* __restrict__ bounds) {
  __shared__ float red[6][4];
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
    const Box b = boxes[g];
    for (int k = 0; k < 3; k++) {
      const float c = 0.5f * (b.lo[k] + b.hi[k]);
      lo[k] = fminf(lo[k], c);
      hi[k] = fmaxf(hi[k], c);
    }
  }
  for (int k = 0; k < 3; k++)
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 3; k++) { red[k][threadIdx.x >> 6] = lo[k]; red[3 + k][threadIdx.x >> 6] = hi[k]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    atomicMin(&bounds[k], float_to_ordered(fminf(fminf(red[k][0], red[k][1]), fminf(red[k][2], red[k][3]))));
  } else if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    atomicMax(&bounds[k], float_to_ordered(fmaxf(fmaxf(red[k][0], red[k][1]), fmaxf(red[k][2], red[k][3]))));
  }
}

__device__ __forceinline__ uint64_t expand21(uint64_t v) {  // spread 21 bits to every third bit
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

__global__ void __launch_bounds__(256) k_morton(const Box* __restrict__ boxes, uint32_t n, const int* __restrict__ bounds,
                                                 uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  float lo[3], ext = 0.0f;
  for (int k = 0; k < 3; k++) {
    lo[k] = ordered_to_float(bounds[k]);
    ext = fmaxf(ext, ordered_to_float(bounds[3 + k]) - lo[k]);
  }
  const float scale = ext > 0.0f ? 2097152.0f / ext : 0.0f;  // 2^21 cells along the longest axis
  const Box b = boxes[g];
  uint64_t code = 0;
  for (int k = 0; k < 3; k++) {
    const float c = 0.5f * (b.lo[k] + b.hi[k]);
    float q = (c - lo[k]) * scale;
    q = fminf(fmaxf(q, 0.0f), 2097151.0f);
    code |= expand21((uint64_t)q) << (2 - k);  // x is the most significant of each triple
  }
  keys[g] = code;
  vals[g] = g;
}

__device__ __forceinline__ int karras_delta(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  if (x == 0) return 64 + __clz((uint32_t)(i ^ j));
  return __clzll((long long)x);
}

// Internal node i covers a range of sorted leaves; children are encoded as refs (kLeafBit | leaf) or node index.
__global__ void __launch_bounds__(256) k_karras(const uint64_t* __restrict__ keys, int n, uint2* __restrict__ children,
                                                 uint32_t* __restrict__ parent_int, uint32_t* __restrict__ parent_leaf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (karras_delta(keys, n, i, i + 1) - karras_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = karras_delta(keys, n, i, i - d);
  int lmax = 2;
  while (karras_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (karras_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = karras_delta(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (karras_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  uint32_t left, right;
  if (lo == gamma) { left = kLeafBit | (uint32_t)gamma; parent_leaf[gamma] = (uint32_t)i; }
  else { left = (uint32_t)gamma; parent_int[gamma] = (uint32_t)i; }
  if (hi == gamma + 1) { right = kLeafBit | (uint32_t)(gamma + 1); parent_leaf[gamma + 1] = (uint32_t)i; }
  else { right = (uint32_t)(gamma + 1); parent_int[gamma + 1] = (uint32_t)i; }
  children[i] = make_uint2(left, right);
  if (i == 0) parent_int[0] = kInvalidRef;
}

__device__ __forceinline__ Box load_child_box(uint32_t ref, const Box* __restrict__ leaf_boxes, const uint32_t* __restrict__ order,
                                              const Box* node_boxes) {
  if (ref & kLeafBit) return leaf_boxes[order[ref & ~kLeafBit]];
  // written by another workgroup during this kernel: bypass the (non-coherent) vector L1
  Box b;
  const float* p = reinterpret_cast<const float*>(&node_boxes[ref]);
  for (int k = 0; k < 3; k++) {
    b.lo[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.hi[k] = __hip_atomic_load(p + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return b;
}

__global__ void __launch_bounds__(256) k_refit(int n, const uint2* __restrict__ children, const uint32_t* __restrict__ parent_int,
                                                const uint32_t* __restrict__ parent_leaf, const Box* __restrict__ leaf_boxes,
                                                const uint32_t* __restrict__ order, Box* node_boxes, uint32_t* flags,
                                                uint32_t* __restrict__ max_depth) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  // depth of this leaf (stats only)
  {
    uint32_t depth = 1, p = parent_leaf[j];
    while (p != 0) { p = parent_int[p]; depth++; }
    atomicMax(max_depth, depth);
  }
  uint32_t cur = parent_leaf[j];
  for (;;) {
    __threadfence();  // release: this thread's node_boxes store (if any) is visible before the arrival count
    const uint32_t old = atomicAdd(&flags[cur], 1u);
    if (old == 0) return;  // first arrival: the sibling subtree is not finished yet
    __threadfence();       // acquire side of the hand-off
    const uint2 ch = children[cur];
    const Box a = load_child_box(ch.x, leaf_boxes, order, node_boxes);
    const Box b = load_child_box(ch.y, leaf_boxes, order, node_boxes);
    float* p = reinterpret_cast<float*>(&node_boxes[cur]);
    for (int k = 0; k < 3; k++) {
      __hip_atomic_store(p + k, fminf(a.lo[k], b.lo[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 3 + k, fmaxf(a.hi[k], b.hi[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (cur == 0) return;
    cur = parent_int[cur];
  }
}

// Collapse the binary radix tree to 4-wide nodes: every binary node at EVEN depth becomes a BvhNode whose children are
// its grandchildren (or a child itself where that child is a leaf).  Odd-depth nodes are absorbed.  Nodes keep their
// binary index, so refs need no remapping.
__global__ void __launch_bounds__(256) k_emit(int n, const uint2* __restrict__ children, const uint32_t* __restrict__ parent_int,
                                               const Box* __restrict__ leaf_boxes, const uint32_t* __restrict__ order,
                                               const Box* __restrict__ node_boxes, BvhNode* __restrict__ nodes, uint32_t ref_base,
                                               uint32_t leaf_tag, uint32_t remap, uint32_t* __restrict__ emitted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n - 1) return;
  uint32_t depth = 0;
  for (uint32_t p = (uint32_t)i; p != 0; p = parent_int[p]) depth++;
  if (depth & 1u) return;
  uint32_t refs[4];
  Box3 boxes[4];
  int count = 0;
  auto add = [&](uint32_t ref) {
    const Box b = (ref & kLeafBit) ? leaf_boxes[order[ref & ~kLeafBit]] : node_boxes[ref];
    Box3 e;
    for (int a = 0; a < 3; a++) { e.lo[a] = b.lo[a]; e.hi[a] = b.hi[a]; }
    boxes[count] = inflate_box(e);
    if (ref & kLeafBit) { const uint32_t pos = ref & ~kLeafBit; refs[count] = leaf_tag | (remap ? order[pos] : pos); }
    else refs[count] = ref_base + ref;
    count++;
  };
  const uint2 ch = children[i];
  const uint32_t c[2] = {ch.x, ch.y};
  for (int k = 0; k < 2; k++) {
    if (c[k] & kLeafBit) {
      add(c[k]);
    } else {
      const uint2 g = children[c[k]];
      add(g.x);
      add(g.y);
    }
  }
  nodes[i] = quantize_node4(boxes, refs, count);
  atomicAdd(emitted, 1u);
}

// Surface-area-guided collapse, one tree level per launch: the 4-wide node rooted at binary node i starts from i's two
// children and keeps opening the internal child with the largest surface area until four slots are used (the even-depth
// rule above opens both children blindly).  On Morton trees this visits 5-13 % fewer nodes per ray (measured with the host
// build of the same traversal: C2 7.73 -> 7.37, a 259 k triangle field 11.3 -> 10.5, the atrium 17.0 -> 14.8).
// Nodes are written DENSELY, level by level: the level's queue holds (binary node, dense index) pairs; an internal child gets
// the dense index next_base + its position in the next level's queue.  The 4-wide tree therefore occupies node_count
// consecutive 64-byte records with the top levels first (C3: 506 175 nodes = 32 MB instead of the 66 MB of n - 1 slots, and a
// small tree can be copied into LDS as it stands).  Leaf refs: leaf_tag | (remap ? order[sorted position] : sorted position).
__device__ __forceinline__ void emit_sah_node(uint32_t t, const uint2* __restrict__ q_in, uint2* __restrict__ q_out,
                                              uint32_t* __restrict__ n_out, uint32_t next_base, uint32_t ref_base, uint32_t leaf_tag,
                                              uint32_t remap, const uint2* __restrict__ children, const Box* __restrict__ leaf_boxes,
                                              const uint32_t* __restrict__ order, const Box* __restrict__ node_boxes,
                                              BvhNode* __restrict__ nodes) {
  const uint32_t i = q_in[t].x, dense = q_in[t].y;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? leaf_boxes[order[ref & ~kLeafBit]] : node_boxes[ref]; };
  auto half_area = [](const Box& b) {
    const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
  };
  const uint2 ch = children[i];
  uint32_t refs[4] = {ch.x, ch.y, kInvalidRef, kInvalidRef};
  Box bx[4];
  bx[0] = box_of(refs[0]); bx[1] = box_of(refs[1]);
  int count = 2;
  while (count < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < count; k++)
      if (!(refs[k] & kLeafBit)) { const float a = half_area(bx[k]); if (a > best_area) { best_area = a; best = k; } }
    if (best < 0) break;
    const uint2 g = children[refs[best]];
    refs[best] = g.x; bx[best] = box_of(g.x);
    refs[count] = g.y; bx[count] = box_of(g.y);
    count++;
  }
  Box3 boxes[4];
  // the internal children of one node get CONSECUTIVE records (one reservation): siblings share 128-byte lines, and a ray that
  // visits two children of a node finds the second one in the line the first one brought in
  uint32_t n_internal = 0;
  for (int k = 0; k < count; k++) n_internal += (refs[k] & kLeafBit) ? 0u : 1u;
  uint32_t p = n_internal ? atomicAdd(n_out, n_internal) : 0u;
  for (int k = 0; k < count; k++) {
    Box3 e;
    for (int a = 0; a < 3; a++) { e.lo[a] = bx[k].lo[a]; e.hi[a] = bx[k].hi[a]; }
    boxes[k] = inflate_box(e);
    if (refs[k] & kLeafBit) {
      const uint32_t pos = refs[k] & ~kLeafBit;
      refs[k] = leaf_tag | (remap ? order[pos] : pos);
    } else {
      q_out[p] = make_uint2(refs[k], next_base + p);
      refs[k] = ref_base + next_base + p;
      p++;
    }
  }
  nodes[dense] = quantize_node4(boxes, refs, count);
}
// The 6-wide form (BvhNode6; one-BVH structure, device-driven build): open the internal child with the largest surface area until six
// slots are used; internal children first, then the leaves.  One reservation of consecutive node records AND one of consecutive triangle
// slots per node: the leaf children of a node become base_leaf + 0, 1, ... (tri_perm[slot] = the triangle's position in the Morton
// order; k_reorder_tris6 places the triangles accordingly).
__device__ __forceinline__ void emit_sah_node6(uint32_t t, const uint2* __restrict__ q_in, uint2* __restrict__ q_out, uint32_t* n_out, uint32_t* leaf_out,
                                               uint32_t next_base, const uint2* __restrict__ children, const Box* __restrict__ leaf_boxes,
                                               const uint32_t* __restrict__ order, const Box* __restrict__ node_boxes, uint32_t* __restrict__ tri_perm,
                                               BvhNode* __restrict__ nodes) {
  const uint32_t i = q_in[t].x, dense = q_in[t].y;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? leaf_boxes[order[ref & ~kLeafBit]] : node_boxes[ref]; };
  auto half_area = [](const Box& b) {
    const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
  };
  const uint2 ch = children[i];
  uint32_t refs[6] = {ch.x, ch.y, kInvalidRef, kInvalidRef, kInvalidRef, kInvalidRef};
  Box bx[6];
  bx[0] = box_of(refs[0]); bx[1] = box_of(refs[1]);
  int count = 2;
  while (count < 6) {
    int best = -1;
    float best_area = -1.0f;
    for (int k = 0; k < count; k++)
      if (!(refs[k] & kLeafBit)) { const float a = half_area(bx[k]); if (a > best_area) { best_area = a; best = k; } }
    if (best < 0) break;
    const uint2 g = children[refs[best]];
    refs[best] = g.x; bx[best] = box_of(g.x);
    refs[count] = g.y; bx[count] = box_of(g.y);
    count++;
  }
  uint32_t n_int = 0;
  for (int k = 0; k < count; k++) n_int += (refs[k] & kLeafBit) ? 0u : 1u;
  const uint32_t n_leaf = (uint32_t)count - n_int;
  const uint32_t p_node = n_int ? atomicAdd(n_out, n_int) : 0u;
  const uint32_t p_leaf = n_leaf ? atomicAdd(leaf_out, n_leaf) : 0u;
  Box3 boxes[6];
  uint32_t a_int = 0, a_leaf = 0;
  for (int k = 0; k < count; k++) {
    Box3 e;
    for (int a = 0; a < 3; a++) { e.lo[a] = bx[k].lo[a]; e.hi[a] = bx[k].hi[a]; }
    if (refs[k] & kLeafBit) {
      boxes[n_int + a_leaf] = inflate_box(e);
      tri_perm[p_leaf + a_leaf] = refs[k] & ~kLeafBit;
      a_leaf++;
    } else {
      boxes[a_int] = inflate_box(e);
      q_out[p_node + a_int] = make_uint2(refs[k], next_base + p_node + a_int);
      a_int++;
    }
  }
  reinterpret_cast<BvhNode6*>(nodes)[dense] = quantize_node6(boxes, (int)n_int, (int)n_leaf, next_base + p_node, p_leaf);
}
__global__ void __launch_bounds__(256) k_emit_sah(uint32_t n_in, const uint2* __restrict__ q_in, uint2* __restrict__ q_out,
                                                   uint32_t* __restrict__ n_out, uint32_t next_base, uint32_t ref_base, uint32_t leaf_tag,
                                                   uint32_t remap, const uint2* __restrict__ children, const Box* __restrict__ leaf_boxes,
                                                   const uint32_t* __restrict__ order, const Box* __restrict__ node_boxes,
                                                   BvhNode* __restrict__ nodes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_in) return;
  emit_sah_node(t, q_in, q_out, n_out, next_base, ref_base, leaf_tag, remap, children, leaf_boxes, order, node_boxes, nodes);
}

// ---- the builder's device-resident state: PLOC passes and collapse levels are launched back to back, each kernel reads what the one
// before it left here, and the host looks in only every few passes / levels (it used to wait for a read-back after EVERY pass and level:
// 46 round trips of ~25 us on C3, a third of the build).  Grids are sized from the last count the host saw (counts only shrink during
// PLOC; a collapse level is at most 4x the one before); threads beyond the real count leave at once.
// (Measured and dropped, r3: the same loops as two COOPERATIVE kernels with grid barriers — correct, and 3.5x slower: a grid-wide
// barrier costs ~100 us on this part, it has to write back and invalidate eight XCD-private L2s; tools/experiments/r03_bvh_cooperative_build.patch.)
constexpr uint32_t kMaxLevels = 128;  // 4-wide levels (the traversal stack bounds them far lower: 3 entries per level)
struct PlocSlot { uint32_t cur, base, buf, ok; };  // clusters left, binary nodes created, which cluster buffer holds them, 0 = gave up
struct BuildState {
  PlocSlot ploc[2];                  // pass p reads slot p & 1 and writes slot (p + 1) & 1
  uint32_t level_count[kMaxLevels];  // collapse: nodes queued for level l + 1 by level l (zeroed before the first level)
  uint32_t leaf_count;               // 6-wide collapse: triangle slots handed out so far
};

// The first kHeadLevels collapse levels (level l holds at most 4^l nodes: <= 1 024 up to level 5) in one single-block launch, block barriers
// between the levels: a level is a chain of ~6 dependent loads whatever its size, so a launch per tiny level cost ~27 us each.
constexpr uint32_t kHeadLevels = 6;
template <bool W6>
__global__ void __launch_bounds__(1024) k_emit_sah_head(BuildState* st, uint32_t ploc_slot, uint32_t levels, uint2* q0, uint2* q1, uint32_t* level_count,
                                                         uint32_t ref_base, uint32_t leaf_tag, uint32_t remap, const uint2* __restrict__ children,
                                                         const Box* __restrict__ leaf_boxes, const uint32_t* __restrict__ order,
                                                         const Box* __restrict__ node_boxes, uint32_t* __restrict__ tri_perm, BvhNode* __restrict__ nodes) {
  if (st->ploc[ploc_slot].ok == 0) return;
  uint32_t n_in = 1, before = 0;
  for (uint32_t level = 0; level < levels; level++) {
    if (threadIdx.x < n_in) {
      if (W6) emit_sah_node6(threadIdx.x, (level & 1u) ? q1 : q0, (level & 1u) ? q0 : q1, &level_count[level], &st->leaf_count, before + n_in, children, leaf_boxes,
                             order, node_boxes, tri_perm, nodes);
      else emit_sah_node(threadIdx.x, (level & 1u) ? q1 : q0, (level & 1u) ? q0 : q1, &level_count[level], before + n_in, ref_base, leaf_tag, remap, children,
                         leaf_boxes, order, node_boxes, nodes);
    }
    __threadfence_block();
    __syncthreads();
    before += n_in;
    n_in = __hip_atomic_load(&level_count[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (n_in > 1024u) n_in = 0;  // (cannot happen: the host keeps width^level <= 1 024)
  }
}

// One collapse level, its size and numbering base read from the counts the levels before it left (level 0: the root alone).
template <bool W6>
__global__ void __launch_bounds__(256) k_emit_sah_dev(BuildState* st, uint32_t ploc_slot, uint32_t level, const uint2* __restrict__ q_in,
                                                       uint2* __restrict__ q_out, uint32_t* level_count, uint32_t ref_base, uint32_t leaf_tag,
                                                       uint32_t remap, const uint2* __restrict__ children, const Box* __restrict__ leaf_boxes,
                                                       const uint32_t* __restrict__ order, const Box* __restrict__ node_boxes, uint32_t* __restrict__ tri_perm,
                                                       BvhNode* __restrict__ nodes) {
  if (st->ploc[ploc_slot].ok == 0) return;  // PLOC gave up: the host falls back to the radix tree
  uint32_t n_in = 1, before = 0;            // nodes of this level, nodes of all levels before it
  for (uint32_t l = 0; l < level; l++) { before += n_in; n_in = st->level_count[l]; }
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_in) return;
  if (W6) emit_sah_node6(t, q_in, q_out, &level_count[level], &st->leaf_count, before + n_in, children, leaf_boxes, order, node_boxes, tri_perm, nodes);
  else emit_sah_node(t, q_in, q_out, &level_count[level], before + n_in, ref_base, leaf_tag, remap, children, leaf_boxes, order, node_boxes, nodes);
}
// triangles in the order the 6-wide collapse numbered them: slot -> position in the Morton order -> flattening index
__global__ void __launch_bounds__(256) k_reorder_tris6(int n, const uint32_t* __restrict__ tri_perm, const uint32_t* __restrict__ order,
                                                        const TriRec* __restrict__ tris_in, TriRec* __restrict__ tris_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) tris_out[i] = tris_in[order[tri_perm[i]]];
}
__global__ void __launch_bounds__(256) k_reorder_tris(int n, const uint32_t* __restrict__ order, const TriRec* __restrict__ tris_in,
                                                       TriRec* __restrict__ tris_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) tris_out[i] = tris_in[order[i]];
}
__global__ void k_seed_queue(uint2* q, uint32_t* counters, uint32_t root) { q[0] = make_uint2(root, 0u); counters[0] = 0u; counters[1] = 0u; }

// ---- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) over the Morton order -----------------------------
// Every cluster looks +-kPlocRadius positions around itself for the neighbour whose union with it has the smallest surface
// area; mutual nearest neighbours merge into a new binary node, the cluster array is compacted (order preserved) and the
// search repeats until one cluster is left.  Node indices and positions come from prefix sums, so the tree is the same on
// every run.  Against the Karras radix tree over the same order: 11-17 % fewer node visits and 7-21 % fewer triangle tests
// per ray (host build of the same traversal, tests/emu EMU_PLOC).
#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 8
#endif
constexpr int kPlocRadius = PT_PLOC_RADIUS;
constexpr uint32_t kPlocTail = 1024;  // clusters the single-block tail takes over at (k_ploc_tail)
__device__ __forceinline__ float merged_half_area(const Box& a, const Box& b) {
  const float x = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
  const float y = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
  const float z = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
  return x * y + y * z + z * x;
}
__global__ void __launch_bounds__(256) k_ploc_init(uint32_t n, const Box* __restrict__ leaf_boxes, const uint32_t* __restrict__ order,
                                                    uint32_t* __restrict__ cl_ref, Box* __restrict__ cl_box) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cl_ref[i] = kLeafBit | i;
  cl_box[i] = leaf_boxes[order[i]];
}
__global__ void __launch_bounds__(256) k_ploc_nn(uint32_t n, const Box* __restrict__ cl_box, uint32_t* __restrict__ nn) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= (int)n) return;
  const Box me = cl_box[i];
  float best = kInf;
  int bj = i;
  // candidate order: the pairing partner i ^ 1 first, then by distance (left before right); a strict `<` keeps the first
  // of equal candidates, so a run of identical boxes pairs up as (0,1)(2,3)... instead of one merge per pass
  auto consider = [&](int j) {
    if (j < 0 || j >= (int)n || j == i) return;
    const float a = merged_half_area(me, cl_box[j]);
    if (a < best) { best = a; bj = j; }
  };
  consider(i ^ 1);
  for (int d = 1; d <= kPlocRadius; d++) { consider(i - d); consider(i + d); }
  nn[i] = (uint32_t)bj;
}
__global__ void __launch_bounds__(256) k_ploc_flags(uint32_t n, const uint32_t* __restrict__ nn, uint32_t* __restrict__ merge_flag,
                                                     uint32_t* __restrict__ keep_flag) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = nn[i];
  const bool mutual = j != i && nn[j] == i;
  merge_flag[i] = (mutual && i < j) ? 1u : 0u;
  keep_flag[i] = (mutual && i > j) ? 0u : 1u;
}
__global__ void __launch_bounds__(256) k_ploc_apply(uint32_t n, const uint32_t* __restrict__ nn, const uint32_t* __restrict__ merge_flag,
                                                     const uint32_t* __restrict__ keep_flag, const uint32_t* __restrict__ node_off,
                                                     const uint32_t* __restrict__ pos, uint32_t base, const uint32_t* __restrict__ cl_ref,
                                                     const Box* __restrict__ cl_box, uint32_t* __restrict__ out_ref, Box* __restrict__ out_box,
                                                     uint2* __restrict__ children, Box* __restrict__ node_boxes, uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1) { counts[0] = node_off[i] + merge_flag[i]; counts[1] = pos[i] + keep_flag[i]; }
  if (!keep_flag[i]) return;
  const uint32_t p = pos[i];
  if (merge_flag[i]) {
    const uint32_t j = nn[i], idx = base + node_off[i];
    const Box a = cl_box[i], b = cl_box[j];
    Box m;
    for (int k = 0; k < 3; k++) { m.lo[k] = fminf(a.lo[k], b.lo[k]); m.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    children[idx] = make_uint2(cl_ref[i], cl_ref[j]);
    node_boxes[idx] = m;
    out_ref[p] = idx;
    out_box[p] = m;
  } else {
    out_ref[p] = cl_ref[i];
    out_box[p] = cl_box[i];
  }
}

// One PLOC pass as three launches that take the cluster count from device memory: (A) nearest-neighbour search; (B) every block counts the
// merges / survivors of its contiguous tile of the cluster array; (C) every block sums the counts of the blocks before it (its output
// offsets), recomputes its tile's flags, numbers them with ballots, writes the merged / kept clusters, and block 0 leaves the next pass its
// state.  Same candidate order, same mutual-pair rule, same numbering (positions and node indices in cluster order) as k_ploc_nn / _flags /
// two device-wide prefix sums / _apply: the tree is the same; a pass is 3 launches instead of 7 + a read-back.  Passes launched after the
// cluster count has fallen to kPlocTail (the host's count is a few passes old) do nothing but hand the state on.
__device__ __forceinline__ void ploc_flags(uint32_t i, uint32_t cur, const uint32_t* __restrict__ nn, uint32_t& j, bool& mf, bool& kf) {
  j = i; mf = false; kf = false;
  if (i < cur) {
    j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    mf = mutual && i < j;
    kf = !(mutual && i > j);
  }
}
__device__ __forceinline__ bool ploc_active(const PlocSlot& ps) { return ps.ok != 0 && ps.cur > kPlocTail; }
__global__ void k_ploc_state_init(BuildState* st, uint32_t n) {
  st->ploc[0] = PlocSlot{n, 0u, 0u, 1u};
  st->ploc[1] = PlocSlot{n, 0u, 0u, 1u};
}
__global__ void __launch_bounds__(256) k_ploc_nn_dev(const BuildState* __restrict__ st, uint32_t pass, const Box* __restrict__ box0, const Box* __restrict__ box1,
                                                      uint32_t* __restrict__ nn) {
  const PlocSlot ps = st->ploc[pass & 1u];
  if (!ploc_active(ps)) return;
  const Box* cl_box = ps.buf ? box1 : box0;
  const uint32_t cur = ps.cur;
  for (uint32_t ii = blockIdx.x * 256 + threadIdx.x; ii < cur; ii += gridDim.x * 256) {
    const int i = (int)ii;
    const Box me = cl_box[i];
    float best = kInf;
    int bj = i;
    auto consider = [&](int j) {
      if (j < 0 || j >= (int)cur || j == i) return;
      const float ar = merged_half_area(me, cl_box[j]);
      if (ar < best) { best = ar; bj = j; }
    };
    consider(i ^ 1);
    for (int d = 1; d <= kPlocRadius; d++) { consider(i - d); consider(i + d); }
    nn[i] = (uint32_t)bj;
  }
}
// this block's tile [t0, t1) of the cluster array: a multiple of the block size per block, in cluster order
__device__ __forceinline__ void ploc_tile(uint32_t cur, uint32_t& t0, uint32_t& t1) {
  const uint32_t tile = ((cur + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
  t0 = min(cur, blockIdx.x * tile);
  t1 = min(cur, t0 + tile);
}
__global__ void __launch_bounds__(256) k_ploc_count_dev(const BuildState* __restrict__ st, uint32_t pass, const uint32_t* __restrict__ nn,
                                                         uint2* __restrict__ block_counts) {
  __shared__ uint32_t s_m[4], s_k[4];
  const PlocSlot ps = st->ploc[pass & 1u];
  if (!ploc_active(ps)) return;
  uint32_t t0, t1;
  ploc_tile(ps.cur, t0, t1);
  uint32_t cm = 0, ck = 0;
  for (uint32_t c = t0; c < t1; c += 256) {
    uint32_t j; bool mf, kf;
    ploc_flags(c + threadIdx.x, ps.cur, nn, j, mf, kf);
    cm += (uint32_t)__popcll(__ballot(mf)); ck += (uint32_t)__popcll(__ballot(kf));
  }
  if ((threadIdx.x & 63u) == 0) { s_m[threadIdx.x >> 6] = cm; s_k[threadIdx.x >> 6] = ck; }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = make_uint2(s_m[0] + s_m[1] + s_m[2] + s_m[3], s_k[0] + s_k[1] + s_k[2] + s_k[3]);
}
__global__ void __launch_bounds__(256) k_ploc_apply_dev(BuildState* __restrict__ st, uint32_t pass, uint32_t n, const uint32_t* __restrict__ nn,
                                                         const uint2* __restrict__ block_counts, uint32_t* __restrict__ ref0, uint32_t* __restrict__ ref1,
                                                         Box* __restrict__ box0, Box* __restrict__ box1, uint2* __restrict__ children,
                                                         Box* __restrict__ node_boxes) {
  __shared__ uint32_t s_m[4], s_k[4];
  __shared__ uint32_t s_red[4][256];
  const PlocSlot ps = st->ploc[pass & 1u];
  if (!ploc_active(ps)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->ploc[(pass + 1u) & 1u] = ps;
    return;
  }
  const uint32_t cur = ps.cur, base = ps.base;
  const uint32_t* cl_ref = ps.buf ? ref1 : ref0; const Box* cl_box = ps.buf ? box1 : box0;
  uint32_t* out_ref = ps.buf ? ref0 : ref1; Box* out_box = ps.buf ? box0 : box1;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // offsets of this block = counts of the blocks before it; totals = counts of all blocks
  uint32_t off_m, off_k, tot_m, tot_k;
  {
    uint32_t pm = 0, pk = 0, am = 0, ak = 0;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += 256) {
      const uint2 v = block_counts[b];
      am += v.x; ak += v.y;
      if (b < blockIdx.x) { pm += v.x; pk += v.y; }
    }
    s_red[0][threadIdx.x] = pm; s_red[1][threadIdx.x] = pk; s_red[2][threadIdx.x] = am; s_red[3][threadIdx.x] = ak;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o)
        for (int q = 0; q < 4; q++) s_red[q][threadIdx.x] += s_red[q][threadIdx.x + o];
      __syncthreads();
    }
    off_m = s_red[0][0]; off_k = s_red[1][0]; tot_m = s_red[2][0]; tot_k = s_red[3][0];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // no progress / inconsistent counts: give up (the host falls back to the radix tree)
    const bool good = tot_m != 0 && tot_k == cur - tot_m && base + tot_m <= n - 1;
    st->ploc[(pass + 1u) & 1u] = PlocSlot{tot_k, base + tot_m, ps.buf ^ 1u, good ? 1u : 0u};
  }
  uint32_t t0, t1;
  ploc_tile(cur, t0, t1);
  for (uint32_t c = t0; c < t1; c += 256) {
    const uint32_t i = c + threadIdx.x;
    uint32_t j; bool mf, kf;
    ploc_flags(i, cur, nn, j, mf, kf);
    const unsigned long long bm = __ballot(mf), bk = __ballot(kf);
    if (lane == 0) { s_m[wave] = (uint32_t)__popcll(bm); s_k[wave] = (uint32_t)__popcll(bk); }
    __syncthreads();
    uint32_t wm = 0, wk = 0;  // counts of the waves before this one in the chunk
    for (uint32_t w = 0; w < wave; w++) { wm += s_m[w]; wk += s_k[w]; }
    const uint32_t cm = s_m[0] + s_m[1] + s_m[2] + s_m[3], ck = s_k[0] + s_k[1] + s_k[2] + s_k[3];
    const unsigned long long below = (1ull << lane) - 1ull;
    if (kf) {
      const uint32_t pos = off_k + wk + (uint32_t)__popcll(bk & below);
      if (mf) {
        const uint32_t idx = base + off_m + wm + (uint32_t)__popcll(bm & below);
        const Box x = cl_box[i], y = cl_box[j];
        Box m;
        for (int k = 0; k < 3; k++) { m.lo[k] = fminf(x.lo[k], y.lo[k]); m.hi[k] = fmaxf(x.hi[k], y.hi[k]); }
        m._pad[0] = m._pad[1] = 0.0f;
        if (idx < n - 1) {  // (always, when the counts are consistent; an inconsistent pass is abandoned above)
          children[idx] = make_uint2(cl_ref[i], cl_ref[j]);
          node_boxes[idx] = m;
        }
        out_ref[pos] = idx;
        out_box[pos] = m;
      } else {
        out_ref[pos] = cl_ref[i];
        out_box[pos] = cl_box[i];
      }
    }
    off_m += cm; off_k += ck;
    __syncthreads();
  }
}

// The LAST passes of PLOC in one launch: once at most kPlocTail clusters are left, one 1024-thread block keeps them in LDS and
// runs nearest-neighbour search, mutual-pair test, the two prefix sums and the merge for every remaining pass between block
// barriers.  Same arithmetic, same candidate order, same node numbering as the multi-kernel passes (k_ploc_nn / _flags / _apply):
// the tree does not change; what goes away is ~25 of the ~40 passes' worth of tiny dependent launches and host round trips
// (C3: BVH build 4.4 -> 3.x ms).  counts[0] = nodes created in all (base), counts[1] = clusters left (1 on success).
__device__ __forceinline__ void ploc_tail_body(uint32_t n0, uint32_t base0, const uint32_t* __restrict__ ref_in, const Box* __restrict__ box_in,
                                               uint2* __restrict__ children, Box* __restrict__ node_boxes, uint32_t* __restrict__ counts) {
  __shared__ Box s_box[2][kPlocTail];
  __shared__ uint32_t s_ref[2][kPlocTail];
  __shared__ uint32_t s_nn[kPlocTail];
  __shared__ uint32_t s_merge[kPlocTail], s_keep[kPlocTail];   // inclusive prefix sums of the pass's flags
  const uint32_t i = threadIdx.x;
  if (i < n0) { s_box[0][i] = box_in[i]; s_ref[0][i] = ref_in[i]; }
  __syncthreads();
  uint32_t cur = n0, base = base0;
  int a = 0;
  for (uint32_t pass = 0; cur > 1 && pass < 4 * kPlocTail; pass++) {
    // nearest neighbour within +-kPlocRadius (k_ploc_nn)
    if (i < cur) {
      const Box me = s_box[a][i];
      float best = kInf;
      int bj = (int)i;
      auto consider = [&](int j) {
        if (j < 0 || j >= (int)cur || j == (int)i) return;
        const float ar = merged_half_area(me, s_box[a][j]);
        if (ar < best) { best = ar; bj = j; }
      };
      consider((int)i ^ 1);
      for (int d = 1; d <= kPlocRadius; d++) { consider((int)i - d); consider((int)i + d); }
      s_nn[i] = (uint32_t)bj;
    }
    __syncthreads();
    // mutual pairs (k_ploc_flags) and their inclusive prefix sums (Hillis-Steele over the block)
    uint32_t j = 0, mf = 0, kf = 0;
    if (i < cur) {
      j = s_nn[i];
      const bool mutual = j != i && s_nn[j] == i;
      mf = (mutual && i < j) ? 1u : 0u;
      kf = (mutual && i > j) ? 0u : 1u;
    }
    s_merge[i] = mf; s_keep[i] = kf;
    __syncthreads();
    for (uint32_t off = 1; off < kPlocTail; off <<= 1) {
      const uint32_t vm = i >= off ? s_merge[i - off] : 0u, vk = i >= off ? s_keep[i - off] : 0u;
      __syncthreads();
      s_merge[i] += vm; s_keep[i] += vk;
      __syncthreads();
    }
    const uint32_t n_merge = s_merge[kPlocTail - 1], n_keep = s_keep[kPlocTail - 1];
    // merge / copy into the other buffer (k_ploc_apply)
    if (i < cur && kf) {
      const uint32_t pos = s_keep[i] - 1u;
      if (mf) {
        const uint32_t idx = base + (s_merge[i] - 1u);
        const Box x = s_box[a][i], y = s_box[a][j];
        Box m;
        for (int k = 0; k < 3; k++) { m.lo[k] = fminf(x.lo[k], y.lo[k]); m.hi[k] = fmaxf(x.hi[k], y.hi[k]); }
        m._pad[0] = m._pad[1] = 0.0f;
        children[idx] = make_uint2(s_ref[a][i], s_ref[a][j]);
        node_boxes[idx] = m;
        s_ref[a ^ 1][pos] = idx;
        s_box[a ^ 1][pos] = m;
      } else {
        s_ref[a ^ 1][pos] = s_ref[a][i];
        s_box[a ^ 1][pos] = s_box[a][i];
      }
    }
    __syncthreads();
    if (n_merge == 0) break;  // (cannot happen: the globally closest pair is always mutual)
    base += n_merge;
    cur = n_keep;
    a ^= 1;
  }
  if (i == 0) { counts[0] = base; counts[1] = cur; }
}
__global__ void __launch_bounds__(1024) k_ploc_tail(uint32_t n0, uint32_t base0, const uint32_t* __restrict__ ref_in, const Box* __restrict__ box_in,
                                                     uint2* __restrict__ children, Box* __restrict__ node_boxes, uint32_t* __restrict__ counts) {
  ploc_tail_body(n0, base0, ref_in, box_in, children, node_boxes, counts);
}
// ... the same, entered from the device-driven passes: cluster count, node base and buffer come from the state slot; it leaves its
// verdict there (ok = the root is the last node created, n - 2, and one cluster is left)
__global__ void __launch_bounds__(1024) k_ploc_tail_dev(BuildState* __restrict__ st, uint32_t slot, uint32_t n, const uint32_t* __restrict__ ref0,
                                                         const uint32_t* __restrict__ ref1, const Box* __restrict__ box0, const Box* __restrict__ box1,
                                                         uint2* __restrict__ children, Box* __restrict__ node_boxes, uint32_t* __restrict__ counts) {
  const PlocSlot ps = st->ploc[slot];
  if (ps.ok == 0) return;
  if (ps.cur > kPlocTail) { if (threadIdx.x == 0) st->ploc[slot].ok = 0; return; }
  ploc_tail_body(ps.cur, ps.base, ps.buf ? ref1 : ref0, ps.buf ? box1 : box0, children, node_boxes, counts);
  __syncthreads();
  if (threadIdx.x == 0) st->ploc[slot] = PlocSlot{counts[1], counts[0], ps.buf, (counts[0] == n - 1 && counts[1] == 1) ? 1u : 0u};
}

#define LB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

// One device allocation per tree build, carved up front: the builder used to make ~30 hipMalloc / hipFree pairs per build (each a
// driver round trip, every hipFree a device synchronisation) around 2.5 ms of kernels.
struct Arena {
  char* base = nullptr;
  size_t off = 0, cap = 0;
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  template <class T> T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += pad(sizeof(T) * count);
    return off <= cap ? p : nullptr;
  }
};

// Builds the binary tree by PLOC over the Morton order `order` (n >= 2).  Fills children[] / node_boxes[] (n - 1 slots, the
// root is the last node created) and *root.  *ok = false when the pass limit is hit (the caller falls back to the radix tree).
static size_t ploc_scratch_bytes(uint32_t n, size_t* scan_bytes_out) {
  size_t scan_bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0);
  *scan_bytes_out = scan_bytes;
  return 2 * (Arena::pad(sizeof(uint32_t) * (size_t)n) + Arena::pad(sizeof(Box) * (size_t)n)) + 5 * Arena::pad(sizeof(uint32_t) * (size_t)n) +
         Arena::pad(2 * sizeof(uint32_t)) + Arena::pad(scan_bytes);
}
static hipError_t ploc_build(hipStream_t s, uint32_t n, const Box* leaf_boxes, const uint32_t* order, uint2* children, Box* node_boxes,
                             uint32_t* root, bool* ok, Arena& arena, size_t scan_bytes) {
  hipError_t err = hipSuccess;
  *ok = false;
  uint32_t* cl_ref[2] = {nullptr, nullptr}; Box* cl_box[2] = {nullptr, nullptr};
  uint32_t *nn = nullptr, *merge_flag = nullptr, *keep_flag = nullptr, *node_off = nullptr, *pos = nullptr, *counts = nullptr;
  void* scan_tmp = nullptr;
  uint32_t cur = n, base = 0, passes = 0;
  int a = 0;
  const size_t mark = arena.off;  // (the radix-tree fallback reuses the arena: everything taken here is handed back on return)
  for (int k = 0; k < 2; k++) { cl_ref[k] = arena.take<uint32_t>(n); cl_box[k] = arena.take<Box>(n); }
  nn = arena.take<uint32_t>(n); merge_flag = arena.take<uint32_t>(n); keep_flag = arena.take<uint32_t>(n); node_off = arena.take<uint32_t>(n);
  pos = arena.take<uint32_t>(n); counts = arena.take<uint32_t>(2);
  scan_tmp = arena.take<char>(scan_bytes);
  if (!scan_tmp && scan_bytes) { arena.off = mark; return hipErrorOutOfMemory; }
  hipLaunchKernelGGL(k_ploc_init, dim3((n + 255) / 256), dim3(256), 0, s, n, leaf_boxes, order, cl_ref[0], cl_box[0]);
  while (cur > 1) {
    if (cur <= kPlocTail) {  // the rest in one launch
      hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(1024), 0, s, cur, base, cl_ref[a], cl_box[a], children, node_boxes, counts);
      uint32_t h[2];
      LB_CHECK(hipMemcpyAsync(h, counts, sizeof(h), hipMemcpyDeviceToHost, s));
      LB_CHECK(hipStreamSynchronize(s));
      if (h[1] != 1 || h[0] > n - 1) goto done;
      base = h[0];
      cur = 1;
      break;
    }
    if (++passes > 256) goto done;  // >= 1 merge per pass is guaranteed, ~30 % per pass is typical: this is a degenerate input
    const uint32_t blocks = (cur + 255) / 256;
    hipLaunchKernelGGL(k_ploc_nn, dim3(blocks), dim3(256), 0, s, cur, cl_box[a], nn);
    hipLaunchKernelGGL(k_ploc_flags, dim3(blocks), dim3(256), 0, s, cur, nn, merge_flag, keep_flag);
    LB_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, merge_flag, node_off, (int)cur, s));
    LB_CHECK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, keep_flag, pos, (int)cur, s));
    hipLaunchKernelGGL(k_ploc_apply, dim3(blocks), dim3(256), 0, s, cur, nn, merge_flag, keep_flag, node_off, pos, base, cl_ref[a], cl_box[a],
                       cl_ref[a ^ 1], cl_box[a ^ 1], children, node_boxes, counts);
    uint32_t h[2];
    LB_CHECK(hipMemcpyAsync(h, counts, sizeof(h), hipMemcpyDeviceToHost, s));
    LB_CHECK(hipStreamSynchronize(s));
    if (h[0] == 0 || h[1] != cur - h[0] || base + h[0] > n - 1) goto done;  // no progress / inconsistent: fall back
    base += h[0];
    cur = h[1];
    a ^= 1;
  }
  if (base == n - 1) {
    *root = n - 2;  // the last node created
    *ok = true;
  }
done:
  arena.off = mark;
  return err;
}


// The device-driven build: PLOC passes, PLOC tail and collapse levels launched back to back (kernels above); the host reads the state
// after each batch of passes / levels (2 + 1-2 read-backs on C3 where there were 46).  PTAMD_BVH_LEGACY=1 selects the per-pass / per-level launches (the fallback, kept for A/B).
constexpr uint32_t kPlocBlocks = 1024;
static bool device_driven_build() { static const bool on = getenv("PTAMD_BVH_LEGACY") == nullptr; return on; }
static size_t dev_scratch_bytes(uint32_t n) {
  return 2 * (Arena::pad(sizeof(uint32_t) * (size_t)n) + Arena::pad(sizeof(Box) * (size_t)n)) + Arena::pad(sizeof(uint32_t) * (size_t)n) +
         Arena::pad(sizeof(uint2) * (size_t)kPlocBlocks) + Arena::pad(2 * sizeof(uint32_t)) + Arena::pad(sizeof(BuildState));
}
// children[] / node_boxes[] / nodes_out are filled on success (*ok): *emitted 4-wide nodes in *levels levels, root = dense node 0.
static hipError_t build_dev(hipStream_t s, uint32_t n, const Box* leaf_boxes, const uint32_t* order, uint32_t stack_capacity, uint2* children,
                            Box* node_boxes, uint2* q0, uint2* q1, uint32_t ref_base, uint32_t leaf_tag, uint32_t remap, BvhNode* nodes_out,
                            bool wide6, uint32_t* tri_perm, Arena& arena, bool* ok, uint32_t* emitted, uint32_t* levels) {
  hipError_t err = hipSuccess;
  *ok = false;
  const size_t mark = arena.off;
  uint32_t* ref0 = arena.take<uint32_t>(n); Box* box0 = arena.take<Box>(n);
  uint32_t* ref1 = arena.take<uint32_t>(n); Box* box1 = arena.take<Box>(n);
  uint32_t* nn = arena.take<uint32_t>(n);
  uint2* block_counts = arena.take<uint2>(kPlocBlocks);
  uint32_t* counts = arena.take<uint32_t>(2);
  BuildState* st = arena.take<BuildState>(1);
  BuildState h;
  uint32_t pass = 0, bound = n, level = 0, level_bound = 1, done_levels = 0, total = 0;
  bool finished = false;
  if (!st) { arena.off = mark; return hipErrorOutOfMemory; }
  LB_CHECK(hipMemsetAsync(st, 0, sizeof(BuildState), s));
  hipLaunchKernelGGL(k_ploc_state_init, dim3(1), dim3(1), 0, s, st, n);
  hipLaunchKernelGGL(k_ploc_init, dim3((n + 255) / 256), dim3(256), 0, s, n, leaf_boxes, order, ref0, box0);
  // ---- PLOC passes down to kPlocTail clusters: as many as the count the host last saw should need (a pass keeps <~ 80 % of the clusters);
  //      passes that turn out to be one too many do nothing ----
  while (bound > kPlocTail) {
    if (pass > 256) goto done;  // >= 1 merge per pass is guaranteed, ~25 % per pass is typical: a degenerate input (-> radix tree)
    const uint32_t blocks = (bound + 255) / 256;
    const uint32_t tiles = blocks < kPlocBlocks ? blocks : kPlocBlocks;
    uint32_t batch = (uint32_t)std::ceil(std::log((double)bound / (double)kPlocTail) / 0.2231);
    batch = batch < 2 ? 2 : batch > 48 ? 48 : batch;
    for (uint32_t k = 0; k < batch; k++, pass++) {
      hipLaunchKernelGGL(k_ploc_nn_dev, dim3(blocks < 4096 ? blocks : 4096), dim3(256), 0, s, st, pass, box0, box1, nn);
      hipLaunchKernelGGL(k_ploc_count_dev, dim3(tiles), dim3(256), 0, s, st, pass, nn, block_counts);
      hipLaunchKernelGGL(k_ploc_apply_dev, dim3(tiles), dim3(256), 0, s, st, pass, n, nn, block_counts, ref0, ref1, box0, box1, children, node_boxes);
    }
    LB_CHECK(hipMemcpyAsync(&h.ploc[0], &st->ploc[pass & 1u], sizeof(PlocSlot), hipMemcpyDeviceToHost, s));
    LB_CHECK(hipStreamSynchronize(s));
    if (!h.ploc[0].ok || h.ploc[0].cur > bound) goto done;
    bound = h.ploc[0].cur;
  }
  hipLaunchKernelGGL(k_ploc_tail_dev, dim3(1), dim3(1024), 0, s, st, pass & 1u, n, ref0, ref1, box0, box1, children, node_boxes, counts);
  // ---- level-synchronous SAH collapse; the root is the last binary node created.  First batch: the levels a balanced tree has and
  //      eight more; further batches of six while the last level still queued something ----
  hipLaunchKernelGGL(k_seed_queue, dim3(1), dim3(1), 0, s, q0, counts, n - 2);
  {
    const uint32_t width = wide6 ? 6u : 4u;
    uint32_t allowed = std::min<uint32_t>(stack_capacity / (width - 1), kMaxLevels - 1);  // levels the traversal stack can hold (width - 1 pushes per level)
    if (wide6) if (const char* e = getenv("PTAMD_TEST_W6_LEVELS")) allowed = std::min<uint32_t>(allowed, (uint32_t)std::max(1, atoi(e)));  // test hook: makes the 6-wide
                                                                                         // form "too deep" so that the retry in the 4-wide form runs (tests/test_gpu_parity.py)
    const uint32_t head = std::min(wide6 ? 4u : kHeadLevels, allowed);                          // width^(head - 1) <= 1 024
    auto grow = [&](uint32_t b) { return b > n / width ? n : b * width; };
    if (head) {
      if (wide6) hipLaunchKernelGGL(k_emit_sah_head<true>, dim3(1), dim3(1024), 0, s, st, pass & 1u, head, q0, q1, st->level_count, ref_base, leaf_tag, remap, children,
                                    leaf_boxes, order, node_boxes, tri_perm, nodes_out);
      else hipLaunchKernelGGL(k_emit_sah_head<false>, dim3(1), dim3(1024), 0, s, st, pass & 1u, head, q0, q1, st->level_count, ref_base, leaf_tag, remap, children,
                              leaf_boxes, order, node_boxes, tri_perm, nodes_out);
    }
    level = head;
    for (uint32_t l = 0; l < head; l++) level_bound = grow(level_bound);
    uint32_t batch = (uint32_t)std::ceil(std::log((double)n) / std::log((double)width)) + 8;
    batch = batch > head ? batch - head : 1;
    while (!finished) {
      for (uint32_t k = 0; k < batch && level < allowed; k++, level++) {
        if (wide6) hipLaunchKernelGGL(k_emit_sah_dev<true>, dim3((level_bound + 255) / 256), dim3(256), 0, s, st, pass & 1u, level, (level & 1u) ? q1 : q0,
                                      (level & 1u) ? q0 : q1, st->level_count, ref_base, leaf_tag, remap, children, leaf_boxes, order, node_boxes, tri_perm, nodes_out);
        else hipLaunchKernelGGL(k_emit_sah_dev<false>, dim3((level_bound + 255) / 256), dim3(256), 0, s, st, pass & 1u, level, (level & 1u) ? q1 : q0,
                                (level & 1u) ? q0 : q1, st->level_count, ref_base, leaf_tag, remap, children, leaf_boxes, order, node_boxes, tri_perm, nodes_out);
        level_bound = grow(level_bound);
      }
      batch = 6;
      LB_CHECK(hipMemcpyAsync(&h, st, sizeof(BuildState), hipMemcpyDeviceToHost, s));
      LB_CHECK(hipStreamSynchronize(s));
      if (!h.ploc[pass & 1u].ok) goto done;
      // levels done so far: level l exists when l == 0 or level l - 1 queued something for it
      total = 0; done_levels = 0;
      for (uint32_t l = 0, n_in = 1; l < level && n_in > 0; l++) { total += n_in; done_levels = l + 1; n_in = h.level_count[l]; finished = n_in == 0; }
      if (!finished) {
        if (level >= allowed) goto done;  // deeper than the traversal stack: the caller falls back
        level_bound = std::min<uint64_t>((uint64_t)h.level_count[level - 1], (uint64_t)n);  // the next level's real size
      }
    }
    if (wide6 && h.leaf_count != n) goto done;  // (every triangle gets exactly one slot)
  }
  if (total >= 1 && total <= n - 1) {
    *ok = true;
    *emitted = total;
    *levels = done_levels;
  }
done:
  arena.off = mark;
  return err;
}

// ---- one tree ------------------------------------------------------------------------------------------------------------
struct TreeInfo { uint32_t root_ref = kInvalidRef, node_span = 0, depth4 = 0; bool wide6 = false; };

// A 4-wide quantised BVH over n >= 1 leaf boxes (device memory), written at nodes_out[0 .. node_span): Morton order (rocPRIM
// radix sort) -> PLOC binary tree (Karras radix tree as the fallback) -> SAH-guided 4-wide collapse, dense in BFS order.
// Internal child refs are `ref_base + index`, leaf refs `leaf_tag | id` with id = the leaf's index in leaf_boxes[] when
// `remap`, its position in the Morton order otherwise (the caller then reorders its leaf array by order_out[], n entries,
// sorted position -> index).  nodes_out needs room for n - 1 records.
static hipError_t tree_arena_bytes(uint32_t n, size_t* sort_bytes_out, size_t* scan_bytes_out, size_t* bytes) {
  size_t sort_bytes = 0, scan_bytes = 0;
  *bytes = 0;
  if (n >= 2) {
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63, 0);
    if (e != hipSuccess) return e;
    const size_t N = n;
    *bytes = Arena::pad(sizeof(int) * 8) + Arena::pad(4 * sizeof(uint32_t)) + 2 * Arena::pad(sizeof(uint64_t) * N) + 2 * Arena::pad(sizeof(uint32_t) * N) +
             Arena::pad(sizeof(uint2) * (N - 1)) + Arena::pad(sizeof(uint32_t) * (N - 1)) + Arena::pad(sizeof(uint32_t) * N) + Arena::pad(sizeof(uint32_t) * (N - 1)) +
             Arena::pad(sizeof(Box) * (N - 1)) + 2 * Arena::pad(sizeof(uint2) * N) + Arena::pad(sort_bytes) +
             std::max(ploc_scratch_bytes(n, &scan_bytes), dev_scratch_bytes(n));
  }
  if (sort_bytes_out) *sort_bytes_out = sort_bytes;
  if (scan_bytes_out) *scan_bytes_out = scan_bytes;
  return hipSuccess;
}
// (`scratch`: tree_arena_bytes(n) bytes of device memory, the caller's)
// `tri_perm_out` != nullptr asks for the 6-wide form (BvhNode6: ref_base 0, no remap): info->wide6 says whether it was built (only the
// device-driven PLOC path builds it; the fallbacks give the 4-wide form) — tri_perm_out[slot] = position in the Morton order.
static hipError_t build_tree(hipStream_t s, uint32_t n, const Box* leaf_boxes, uint32_t stack_capacity, bool use_ploc, BvhNode* nodes_out,
                             uint32_t ref_base, uint32_t leaf_tag, bool remap, uint32_t* order_out, uint32_t* tri_perm_out, char* scratch,
                             size_t scratch_bytes, TreeInfo* info) {
  hipError_t err = hipSuccess;
  *info = TreeInfo{};
  const uint32_t blocks = (n + 255) / 256;
  Box* node_boxes = nullptr;
  uint64_t *keys_a = nullptr, *keys_b = nullptr; uint32_t *vals_a = nullptr, *vals_b = nullptr;
  uint2 *children = nullptr, *queue[2] = {nullptr, nullptr}; uint32_t *parent_int = nullptr, *parent_leaf = nullptr, *flags = nullptr;
  int* bounds = nullptr; uint32_t* counters = nullptr; void* sort_tmp = nullptr; size_t sort_bytes = 0, scan_bytes = 0;
  uint32_t depth_h[2] = {0, 0};
  uint32_t root = 0, binary_depth = 0;
  Arena arena;

  if (n == 1) {
    const uint32_t zero = 0;
    if (order_out) LB_CHECK(hipMemcpyAsync(order_out, &zero, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    LB_CHECK(hipStreamSynchronize(s));
    info->root_ref = leaf_tag | 0u;
    return hipSuccess;
  }
  LB_CHECK(tree_arena_bytes(n, &sort_bytes, &scan_bytes, &arena.cap));
  if (arena.cap > scratch_bytes) return hipErrorOutOfMemory;  // (the callers size the scratch with tree_arena_bytes)
  arena.base = scratch;
  bounds = arena.take<int>(8);
  counters = arena.take<uint32_t>(4);  // [0] max binary depth, [1] emitted (fallback), [2] next level size, [3] unused
  LB_CHECK(hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), s));
  keys_a = arena.take<uint64_t>(n); keys_b = arena.take<uint64_t>(n);
  vals_a = arena.take<uint32_t>(n); vals_b = arena.take<uint32_t>(n);
  children = arena.take<uint2>(n - 1);
  parent_int = arena.take<uint32_t>(n - 1);
  parent_leaf = arena.take<uint32_t>(n);
  flags = arena.take<uint32_t>(n - 1);
  node_boxes = arena.take<Box>(n - 1);
  for (int k = 0; k < 2; k++) queue[k] = arena.take<uint2>(n);
  sort_tmp = arena.take<char>(sort_bytes);

  hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, s, bounds);
  hipLaunchKernelGGL(k_bounds, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, s, leaf_boxes, n, bounds);
  hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, s, leaf_boxes, n, bounds, keys_a, vals_a);
  LB_CHECK(hipcub::DeviceRadixSort::SortPairs(sort_tmp, sort_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 63, s));

  for (;;) {
    if (use_ploc && device_driven_build()) {
      // PLOC + collapse without host round trips; anything it cannot finish (pass limit, a tree deeper than the stack) goes the old way
      bool ok = false;
      uint32_t emitted = 0, levels = 0;
      const bool wide6 = tri_perm_out != nullptr && n >= 2 && n < kMaxWide6Triangles && getenv("PTAMD_BVH4") == nullptr;
      LB_CHECK(build_dev(s, n, leaf_boxes, vals_b, stack_capacity, children, node_boxes, queue[0], queue[1], ref_base, leaf_tag, remap ? 1u : 0u,
                          nodes_out, wide6, tri_perm_out, arena, &ok, &emitted, &levels));
      if (!ok && wide6) {  // (e.g. deeper than the stack at five pushes per level: the 4-wide form needs three)
        LB_CHECK(build_dev(s, n, leaf_boxes, vals_b, stack_capacity, children, node_boxes, queue[0], queue[1], ref_base, leaf_tag, remap ? 1u : 0u,
                            nodes_out, false, nullptr, arena, &ok, &emitted, &levels));
      } else if (ok) info->wide6 = wide6;
      if (ok) {
        info->root_ref = ref_base + 0u;
        if (order_out) LB_CHECK(hipMemcpyAsync(order_out, vals_b, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
        LB_CHECK(hipGetLastError());
        LB_CHECK(hipStreamSynchronize(s));
        info->node_span = emitted;
        info->depth4 = levels;
        break;
      }
      use_ploc = false;
      continue;
    }
    if (use_ploc) {
      bool ok = false;
      LB_CHECK(ploc_build(s, n, leaf_boxes, vals_b, children, node_boxes, &root, &ok, arena, scan_bytes));
      if (!ok) { use_ploc = false; continue; }
    } else {
      LB_CHECK(hipMemsetAsync(flags, 0, sizeof(uint32_t) * (size_t)(n - 1), s));
      LB_CHECK(hipMemsetAsync(counters, 0, 4 * sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_karras, dim3(blocks), dim3(256), 0, s, keys_b, (int)n, children, parent_int, parent_leaf);
      hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(256), 0, s, (int)n, children, parent_int, parent_leaf, leaf_boxes, vals_b,
                         node_boxes, flags, counters);
      LB_CHECK(hipMemcpyAsync(depth_h, counters, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      LB_CHECK(hipStreamSynchronize(s));
      binary_depth = depth_h[0];
      root = 0;
    }
    // level-synchronous SAH collapse, dense BFS numbering
    uint2 *q_in = queue[0], *q_out = queue[1];
    hipLaunchKernelGGL(k_seed_queue, dim3(1), dim3(1), 0, s, q_in, counters + 2, root);
    uint32_t n_in = 1, levels = 0, emitted = 0;
    bool too_deep = false;
    while (n_in > 0) {
      if ((levels + 1) * 3 > stack_capacity) { too_deep = true; break; }
      hipLaunchKernelGGL(k_emit_sah, dim3((n_in + 255) / 256), dim3(256), 0, s, n_in, q_in, q_out, counters + 2, emitted + n_in, ref_base,
                         leaf_tag, remap ? 1u : 0u, children, leaf_boxes, vals_b, node_boxes, nodes_out);
      uint32_t h = 0;
      LB_CHECK(hipMemcpyAsync(&h, counters + 2, sizeof(h), hipMemcpyDeviceToHost, s));
      LB_CHECK(hipMemsetAsync(counters + 2, 0, sizeof(uint32_t), s));
      LB_CHECK(hipStreamSynchronize(s));
      emitted += n_in;
      n_in = h;
      std::swap(q_in, q_out);
      levels++;
    }
    if (too_deep && use_ploc) { use_ploc = false; continue; }  // a pathological cluster tree: try the radix tree
    if (too_deep) {
      // pathological radix tree: the even-depth collapse bounds the 4-wide depth by half the binary depth (nodes keep their
      // binary index: the tree spans n - 1 slots)
      LB_CHECK(hipMemsetAsync(counters + 1, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(256), 0, s, (int)n, children, parent_int, leaf_boxes, vals_b, node_boxes, nodes_out, ref_base,
                         leaf_tag, remap ? 1u : 0u, counters + 1);
      emitted = n - 1;
      levels = (binary_depth + 1) / 2;
      info->root_ref = ref_base + 0u;
    } else {
      info->root_ref = ref_base + 0u;  // the root is dense node 0
    }
    if (order_out) LB_CHECK(hipMemcpyAsync(order_out, vals_b, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
    LB_CHECK(hipGetLastError());
    LB_CHECK(hipStreamSynchronize(s));
    info->node_span = emitted;
    info->depth4 = levels;
    break;
  }

done:
  return err;
}

}  // namespace

hipError_t LbvhScratch::ensure(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  release();
  hipError_t e = hipMalloc((void**)&base, bytes);
  if (e != hipSuccess) { base = nullptr; return e; }
  cap = bytes;
  return hipSuccess;
}
void LbvhScratch::release() {
  if (base) (void)hipFree(base);
  base = nullptr;
  cap = 0;
}

// Flattens the instanced scene to world-space triangles and builds ONE BVH over them (the default: fewest node visits per ray).
hipError_t build_lbvh(hipStream_t s, const DeviceScene& S, const PrimTables& prims, uint32_t instance_count, uint32_t stack_capacity,
                      LbvhScratch* scratch, LbvhResult* out) {
  *out = LbvhResult{};
  if (prims.slot_count == 0) return hipSuccess;
  hipError_t err = hipSuccess;
  const uint32_t n = prims.slot_count;  // the builder's primitives are the leaf slots (one triangle, or two that share an edge)
  const uint32_t blocks = (n + 255) / 256;
  TriRec* tris_tmp = nullptr; Box* leaf_boxes = nullptr; BvhNode* nodes_tmp = nullptr; uint32_t *order = nullptr, *tri_perm = nullptr;
  TreeInfo info;
  const bool use_ploc = getenv("PTAMD_RADIX_TREE") == nullptr;  // PLOC by default; the Karras radix tree is the fallback

  Arena tmp;  // the flattening's temporaries and, behind them, the tree builder's: one kept allocation (LbvhScratch)
  size_t tree_bytes = 0;
  char* tree_scratch = nullptr;
  LB_CHECK(tree_arena_bytes(n, nullptr, nullptr, &tree_bytes));
  tmp.cap = Arena::pad(sizeof(TriRec) * (size_t)n) + Arena::pad(sizeof(Box) * (size_t)n) + 2 * Arena::pad(sizeof(uint32_t) * (size_t)n) +
            Arena::pad(sizeof(BvhNode) * (size_t)(n > 1 ? n - 1 : 1)) + Arena::pad(tree_bytes);
  LB_CHECK(scratch->ensure(tmp.cap));
  tmp.base = scratch->base;
  tris_tmp = tmp.take<TriRec>(n);
  leaf_boxes = tmp.take<Box>(n);
  order = tmp.take<uint32_t>(n);
  tri_perm = tmp.take<uint32_t>(n);
  nodes_tmp = tmp.take<BvhNode>(n > 1 ? n - 1 : 1);
  tree_scratch = tmp.take<char>(tree_bytes);
  LB_CHECK(hipMalloc(&out->tris, sizeof(TriRec) * (size_t)n));
  hipLaunchKernelGGL(k_flatten, dim3(blocks), dim3(256), 0, s, S, prims, instance_count, tris_tmp, leaf_boxes);
  LB_CHECK(build_tree(s, n, leaf_boxes, stack_capacity, use_ploc, nodes_tmp, 0u, kLeafBit, false, order, tri_perm, tree_scratch, tree_bytes, &info));
  if (info.wide6) hipLaunchKernelGGL(k_reorder_tris6, dim3(blocks), dim3(256), 0, s, (int)n, tri_perm, order, tris_tmp, out->tris);  // triangles in the collapse's leaf numbering
  else hipLaunchKernelGGL(k_reorder_tris, dim3(blocks), dim3(256), 0, s, (int)n, order, tris_tmp, out->tris);  // triangles in leaf (Morton) order
  if (info.node_span) {  // keep exactly the records the tree uses
    LB_CHECK(hipMalloc(&out->nodes, sizeof(BvhNode) * (size_t)info.node_span));
    LB_CHECK(hipMemcpyAsync(out->nodes, nodes_tmp, sizeof(BvhNode) * (size_t)info.node_span, hipMemcpyDeviceToDevice, s));
  }
  LB_CHECK(hipGetLastError());
  LB_CHECK(hipStreamSynchronize(s));
  out->root_ref = info.root_ref;
  out->node_count = info.node_span;
  out->depth4 = info.depth4;
  out->wide6 = info.wide6;
  out->slot_count = n;

done:
  if (err != hipSuccess) {
    (void)hipFree(out->nodes); (void)hipFree(out->tris);
    *out = LbvhResult{};
  }
  return err;
}


namespace {
// world box of every instance = union of its flattened triangles' boxes (one block per instance)
__global__ void __launch_bounds__(256) k_instance_boxes(DeviceScene S, const Box* __restrict__ tri_boxes, Box* __restrict__ inst_boxes) {
  const InstanceInfo inst = S.instances[blockIdx.x];
  const uint32_t n = S.meshes[inst.mesh].tri_count;
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (uint32_t t = threadIdx.x; t < n; t += 256) {
    const Box b = tri_boxes[inst.tri_global_base + t];
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], b.lo[k]); hi[k] = fmaxf(hi[k], b.hi[k]); }
  }
  __shared__ float red[6][256];
  for (int k = 0; k < 3; k++) { red[k][threadIdx.x] = lo[k]; red[3 + k][threadIdx.x] = hi[k]; }
  __syncthreads();
  for (uint32_t off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off)
      for (int k = 0; k < 3; k++) {
        red[k][threadIdx.x] = fminf(red[k][threadIdx.x], red[k][threadIdx.x + off]);
        red[3 + k][threadIdx.x] = fmaxf(red[3 + k][threadIdx.x], red[3 + k][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Box b;
    for (int k = 0; k < 3; k++) { b.lo[k] = red[k][0]; b.hi[k] = red[3 + k][0]; }
    b._pad[0] = b._pad[1] = 0.0f;
    inst_boxes[blockIdx.x] = b;
  }
}
// object-space boxes of one mesh's triangles
__global__ void __launch_bounds__(256) k_mesh_boxes(DeviceScene S, MeshInfo mesh, Box* __restrict__ boxes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= mesh.tri_count) return;
  const uint32_t* idx = &S.indices[3 * (size_t)(mesh.tri_base + t)];
  const vec3 v0 = ld3(S.positions[mesh.vertex_base + idx[0]]), v1 = ld3(S.positions[mesh.vertex_base + idx[1]]),
             v2 = ld3(S.positions[mesh.vertex_base + idx[2]]);
  Box b;
  b.lo[0] = fminf(v0.x, fminf(v1.x, v2.x)); b.hi[0] = fmaxf(v0.x, fmaxf(v1.x, v2.x));
  b.lo[1] = fminf(v0.y, fminf(v1.y, v2.y)); b.hi[1] = fmaxf(v0.y, fmaxf(v1.y, v2.y));
  b.lo[2] = fminf(v0.z, fminf(v1.z, v2.z)); b.hi[2] = fmaxf(v0.z, fmaxf(v1.z, v2.z));
  b._pad[0] = b._pad[1] = 0.0f;
  boxes[t] = b;
}
// the bounds a tree's root node spans (decoded exactly as the traversal decodes them: a superset of the leaf boxes)
__global__ void k_root_bounds(const BvhNode* __restrict__ root, MeshTrav* __restrict__ out) {
  const BvhNode n = *root;
  for (int a = 0; a < 3; a++) {
    uint32_t qlo = 255, qhi = 0;
    for (int k = 0; k < 4; k++)
      if (n.ref[k] != kInvalidRef) { qlo = min(qlo, (n.qlo[a] >> (8 * k)) & 0xffu); qhi = max(qhi, (n.qhi[a] >> (8 * k)) & 0xffu); }
    out->lo[a] = n.origin[a] + (float)qlo * node_scale(n.exp[a]);
    out->hi[a] = n.origin[a] + (float)qhi * node_scale(n.exp[a]);
  }
}
__global__ void k_box_bounds(const Box* __restrict__ b, MeshTrav* __restrict__ out) {
  const Box3 e = inflate_box(Box3{{b->lo[0], b->lo[1], b->lo[2]}, {b->hi[0], b->hi[1], b->hi[2]}});
  for (int a = 0; a < 3; a++) { out->lo[a] = e.lo[a]; out->hi[a] = e.hi[a]; }
}
}  // namespace

hipError_t build_two_level(hipStream_t s, const DeviceScene& S, const MeshInfo* meshes, uint32_t mesh_count, uint32_t instance_count,
                           uint32_t tri_count, uint32_t stack_capacity, LbvhScratch* scratch, LbvhResult* out) {
  *out = LbvhResult{};
  if (tri_count == 0 || instance_count == 0) return hipSuccess;
  hipError_t err = hipSuccess;
  Box *tri_boxes = nullptr, *inst_boxes = nullptr, *mesh_boxes = nullptr;
  BvhNode* nodes_tmp = nullptr;
  Arena tmp;
  size_t tree_bytes = 0, tb = 0;
  char* tree_scratch = nullptr;
  TreeInfo tlas;
  uint32_t base = 0, deepest_blas = 0, max_mesh_tris = 0;
  size_t capacity = instance_count;
  const bool use_ploc = getenv("PTAMD_RADIX_TREE") == nullptr;
  std::vector<MeshTrav> mt(mesh_count);
  for (uint32_t m = 0; m < mesh_count; m++) { capacity += meshes[m].tri_count; max_mesh_tris = std::max(max_mesh_tris, meshes[m].tri_count); }

  // temporaries: one kept allocation (LbvhScratch); the tree builder's part is sized for the largest tree (TLAS or a BLAS)
  LB_CHECK(tree_arena_bytes(instance_count, nullptr, nullptr, &tree_bytes));
  LB_CHECK(tree_arena_bytes(max_mesh_tris, nullptr, nullptr, &tb));
  tree_bytes = std::max(tree_bytes, tb);
  tmp.cap = Arena::pad(sizeof(Box) * (size_t)tri_count) + Arena::pad(sizeof(Box) * (size_t)instance_count) +
            Arena::pad(sizeof(Box) * (size_t)std::max(1u, max_mesh_tris)) + Arena::pad(sizeof(BvhNode) * capacity) + Arena::pad(tree_bytes);
  LB_CHECK(scratch->ensure(tmp.cap));
  tmp.base = scratch->base;
  tri_boxes = tmp.take<Box>(tri_count);
  inst_boxes = tmp.take<Box>(instance_count);
  mesh_boxes = tmp.take<Box>(std::max(1u, max_mesh_tris));
  nodes_tmp = tmp.take<BvhNode>(capacity);
  tree_scratch = tmp.take<char>(tree_bytes);
  LB_CHECK(hipMalloc(&out->tris, sizeof(TriRec) * (size_t)tri_count));
  LB_CHECK(hipMalloc(&out->mesh_trav, sizeof(MeshTrav) * (size_t)mesh_count));
  LB_CHECK(hipMemsetAsync(out->mesh_trav, 0, sizeof(MeshTrav) * (size_t)mesh_count, s));
  // world-space triangles in flattening order (the intersection contract's triangles) and the instances' world boxes
  {
    PrimTables singles;  // one triangle per slot, in flattening order: a BLAS leaf `primitive` of instance i is slot tri_global_base(i) + primitive
    singles.slot_count = tri_count;
    hipLaunchKernelGGL(k_flatten, dim3((tri_count + 255) / 256), dim3(256), 0, s, S, singles, instance_count, out->tris, tri_boxes);
  }
  hipLaunchKernelGGL(k_instance_boxes, dim3(instance_count), dim3(256), 0, s, S, tri_boxes, inst_boxes);
  LB_CHECK(build_tree(s, instance_count, inst_boxes, stack_capacity, use_ploc, nodes_tmp, 0u, kInstBit, true, nullptr, nullptr, tree_scratch, tree_bytes, &tlas));
  base = tlas.node_span;
  for (uint32_t m = 0; m < mesh_count; m++) {
    const MeshInfo mesh = meshes[m];
    if (mesh.tri_count == 0) { mt[m].root_ref = kInvalidRef; continue; }
    TreeInfo blas;
    hipLaunchKernelGGL(k_mesh_boxes, dim3((mesh.tri_count + 255) / 256), dim3(256), 0, s, S, mesh, mesh_boxes);
    LB_CHECK(build_tree(s, mesh.tri_count, mesh_boxes, stack_capacity, use_ploc, nodes_tmp + base, base, kLeafBit, true, nullptr, nullptr, tree_scratch, tree_bytes, &blas));
    LB_CHECK(hipMemcpyAsync(&out->mesh_trav[m].root_ref, &blas.root_ref, sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (blas.node_span) hipLaunchKernelGGL(k_root_bounds, dim3(1), dim3(1), 0, s, nodes_tmp + base, out->mesh_trav + m);
    else hipLaunchKernelGGL(k_box_bounds, dim3(1), dim3(1), 0, s, mesh_boxes, out->mesh_trav + m);
    LB_CHECK(hipStreamSynchronize(s));
    base += blas.node_span;
    deepest_blas = std::max(deepest_blas, blas.depth4);
  }
  if (base) {
    LB_CHECK(hipMalloc(&out->nodes, sizeof(BvhNode) * (size_t)base));
    LB_CHECK(hipMemcpyAsync(out->nodes, nodes_tmp, sizeof(BvhNode) * (size_t)base, hipMemcpyDeviceToDevice, s));
  }
  LB_CHECK(hipGetLastError());
  LB_CHECK(hipStreamSynchronize(s));
  out->root_ref = tlas.root_ref;
  out->node_count = base;
  out->depth4 = tlas.depth4 + deepest_blas;
  out->slot_count = tri_count;

done:
  if (err != hipSuccess) {
    (void)hipFree(out->nodes); (void)hipFree(out->tris); (void)hipFree(out->mesh_trav);
    *out = LbvhResult{};
  }
  return err;
}

}  // namespace pt
