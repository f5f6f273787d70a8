// pt_bvh.h — ray / triangle test and 4-wide quantised-BVH traversal (closest-hit and any-hit).
//
// Replaces Apple's closed `intersector<triangle_data, instancing>::intersect` (kernel.metal:244-251, 293-294,
// 511-512, 623-629; acceleration structures built at renderer_pt.cpp:244-294, 653-749).  Semantics kept:
// closest hit in [min_distance, max_distance], accept-any for shadow rays, no culling, opaque geometry.
//
// Intersection contract: world-space triangles, the Moeller-Trumbore sequence below, closest = min t
// with ties broken by the lowest global triangle id (= lowest (instance, primitive)).  The result does not
// depend on the BVH: the slab test is conservative (boxes are inflated at build time, the comparison carries
// slack), so traversal visits a superset of the triangles that can be hit.
#pragma once
#include "pt_bsdf.h"

namespace pt {

struct RayHit {
  float t, u, v;
  uint32_t tri;  // 2 * (index into DeviceScene::tris, leaf order) + half; kInvalidRef = miss
  uint32_t gid;
};

// Moeller-Trumbore with a fixed operation order.  Returns true and (t,u,v) when |det| > 1e-6 |e1|_1 |d x e2|_1, 0 <= u <= 1, 0 <= v,
// u + v <= 1 and tmin <= t <= tmax.  (v0, e1 = v1 - v0, e2 = v2 - v0: world-space, the caller forms the edges.)
PT_HD bool intersect_triangle(vec3 o, vec3 d, float tmin, float tmax, vec3 v0, vec3 e1, vec3 e2, float* t_out, float* u_out,
                              float* v_out) {
  const vec3 p = cross(d, e2);
  const float det = dot(e1, p);
  // r4: a determinant that is rounding noise (a ray IN the triangle's plane: e1 . p cancels to below 1e-6 of its terms' magnitude) is a miss.
  // With `det == 0` alone such rays were accepted or rejected by noise and the answer depended on which coplanar triangles a traversal
  // happened to test — the one place where hits were not independent of the structure (fuzz seeds 20341, 310601).
  if (!(fabsf(det) > (1e-6f * ((fabsf(e1.x) + fabsf(e1.y)) + fabsf(e1.z))) * ((fabsf(p.x) + fabsf(p.y)) + fabsf(p.z)))) return false;
  const float inv = 1.0f / det;
  const vec3 s = o - v0;
  const float u = dot(s, p) * inv;
  if (!(u >= 0.0f && u <= 1.0f)) return false;
  const vec3 q = cross(s, e1);
  const float v = dot(d, q) * inv;
  if (!(v >= 0.0f && u + v <= 1.0f)) return false;
  const float t = dot(e2, q) * inv;
  if (!(t >= tmin && t <= tmax)) return false;
  *t_out = t;
  *u_out = u;
  *v_out = v;
  return true;
}

// corner c (0..3) of a leaf slot: quarter c of the record
PT_HD vec3 slot_corner(const TriRec* rec, uint32_t c) {
  const float* p = reinterpret_cast<const float*>(rec) + 4u * c;
#if defined(__HIP_DEVICE_COMPILE__)
  struct f3 { float x, y, z; };
  const f3 v = ldg(reinterpret_cast<const f3*>(p));
  return v3(v.x, v.y, v.z);
#else
  return v3(p[0], p[1], p[2]);
#endif
}
// (instance, primitive) of triangle `tri` = 2 * slot + half: what the hit log and pt_trace_primary report
PT_HD void triangle_ids(const DeviceScene& S, uint32_t tri, int32_t* inst, int32_t* prim) {
  const TriRec& tr = S.tris[tri >> 1];
  const uint32_t gid = (tri & 1u) ? tr.gid_b : tr.gid_a, in = tr.inst_code & kSlotInstMask;
  *inst = (int32_t)in;
  *prim = (int32_t)((gid >> 2) - S.instances[in].tri_global_base);
}

// r5 — how far beyond the best hit so far a box is still visited.  fp32 Moeller-Trumbore's t carries an error of ~ulp(|o - v0|) / cos(incidence):
// at grazing incidence it reports a hit up to ~1e-5 t BEFORE the ray enters that triangle's (accurately tested) box.  With the r1-r4 slack of
// 5e-7 t such a triangle was culled or not depending on WHEN it came up — on the tree, and in the wave-cooperative kernels on which rays share
// the wave: the full-size C5 parity test found 2 of 3.8e8 paths that differed between one 46-sample batch and the same samples in batches of
// 16 (both camera rays at 78 degrees incidence on a column, hit within 1e-5 of a shared edge).  Boxes are now culled against best.t * (1 +
// 1e-4): every candidate whose Moeller-Trumbore t is within 1e-4 t of the winner's is tested whatever the order.  That is 200x the old margin;
// the rate of rays whose t error exceeds a margin falls with its square (it needs cos(incidence) < ~3 ulp / margin), i.e. from the observed
// 5e-9 per ray to ~1e-13.  Not a proof: see DESIGN.md section 2 for what would be one (ranking hits by max(t, entry into the triangle's own
// box): tools/experiments/r05_own_box_hit_rule.patch, correct and 12 % slower).
constexpr float kCullSlack = 1.0001f;

// Slack of the slab comparison tn <= tf * kSlabSlack in the node steps.  Error budget of one plane distance in "ray space" (trav_node / trav_node6:
// t = fma(q, A, B) with A = 2^k * inv, B = (origin - o) * inv; u = 2^-24 = half an ulp): inv = v_rcp_f32(d) is within 1 ulp = 2u of 1 / d (r5; an IEEE
// division would be u), A inherits that exactly (a power-of-two scale), B adds the subtraction and the product (4u), the fma rounds once: with
// both terms of one sign |dt| <= 5u t per plane.  The comparison sets the entry on one axis against the exit on ANOTHER (independent errors): 10u t =
// 6.0e-7 t.  r1-r5 carried 1.0000005 (8.4u: sized for the 0.5-ulp reciprocal's 2 x 4u — ADVICE r5); r6 widens it to 1.000001 (16.8u).  Where the two
// terms of the fma CANCEL (ray origin beyond the node's origin plane) the error is relative to the terms, not to t: in position space
// <= 4u (node extent + |origin - o|), which is what the 8e-6 relative inflation of every box (inflate_box: 134u of the coordinate magnitude) is for.
// Neither is a proof for a box hugging a coordinate plane seen from far away; the answer there rests on the hit contract's own tolerance
// (kCullSlack) and on the sweeps (tests/test_stage_functions_host.py perturbs the reciprocal by +-1 ulp: no hit, no radiance bit changes).
#ifndef PT_SLAB_SLACK
#define PT_SLAB_SLACK 1.000001f
#endif
constexpr float kSlabSlack = PT_SLAB_SLACK;

// Conservative slab test: entry distance or -1 if missed. fmin/fmax drop the NaN of 0 * inf.
PT_HD float slab_entry(const float lo[3], const float hi[3], vec3 o, vec3 inv, float tmin, float tmax) {
  float t0 = (lo[0] - o.x) * inv.x, t1 = (hi[0] - o.x) * inv.x;
  float tn = fmaxf(tmin, fminf(t0, t1));
  float tf = fminf(tmax, fmaxf(t0, t1));
  t0 = (lo[1] - o.y) * inv.y; t1 = (hi[1] - o.y) * inv.y;
  tn = fmaxf(tn, fminf(t0, t1));
  tf = fminf(tf, fmaxf(t0, t1));
  t0 = (lo[2] - o.z) * inv.z; t1 = (hi[2] - o.z) * inv.z;
  tn = fmaxf(tn, fminf(t0, t1));
  tf = fminf(tf, fmaxf(t0, t1));
  return (tn <= tf * 1.0000005f + 1e-30f) ? tn : -1.0f;
}

// Per-lane traversal stack: the first kLdsStack entries live in LDS ([depth][lane], bank = lane; row kLdsStack is a
// scratch row that absorbs the writes of the branch-free push), the rest spill to a per-thread HBM slab (only
// pathological trees get there).
// Occupancy of the trace kernels: 7 blocks of 256 threads per CU = 7 waves per SIMD need <= 72 VGPRs and <= 22.8 KB of LDS
// per block: (13 + 1) stack rows + (7 + 1) leaf-queue rows of 1 KB.  Measured against 6 waves (16 + 8 rows): closest-hit
// on C3 10 % faster, C2 unchanged; 8 waves (64 VGPRs, 12 + 6 rows) spills registers and is slower.
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 7
#endif
// r4: with two triangles per leaf slot the closest-hit kernel wants 78 registers; at 7 waves per SIMD (72) it spills 5 of them, at 6 none —
// measured per 128-spp step, C3 / C2: closest 112.5 / 36.8 ms at 7 blocks per CU, 109.3 / 35.3 at 6; shadow (71 registers, no spills) 50.8 /
// 21.4 at 7, 52.2 / 21.5 at 6.  So each kernel gets its own occupancy.
#ifndef PT_CLOSEST_WAVES
#define PT_CLOSEST_WAVES 6
#endif
#ifndef PT_SHADOW_WAVES
#define PT_SHADOW_WAVES PT_TRACE_WAVES
#endif
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 13
#endif
#ifndef PT_PEND_LEAVES
#define PT_PEND_LEAVES 7
#endif
constexpr int kLdsStack = PT_LDS_STACK;
constexpr int kSpillStack = 96 - PT_LDS_STACK;  // total 96 entries: up to 3 pushes per level of a 4-wide tree that is <= 48 levels deep
                                 // (binary LBVH depth <= 95 over 63-bit codes + index tie-break, halved by the collapse)

constexpr int kPendLeaves = PT_PEND_LEAVES;   // per-lane queue of leaf candidates awaiting their triangle test (LDS, [slot][lane])
// The 6-wide kernels split the same 22 rows differently: a node queues ONE entry for all its leaf children (so the queue can be short)
// and pushes up to five children (so the stack runs deeper).  Measured on C3 (closest-hit ms per 64 spp): 13 + 7 rows 59.4, 15 + 5 59.0,
// 16 + 4 58.5, 17 + 3 58.1, 18 + 2 58.5.
#ifndef PT_LDS_STACK6
#define PT_LDS_STACK6 17
#endif
#ifndef PT_PEND_LEAVES6
#define PT_PEND_LEAVES6 3
#endif
constexpr int kLdsStack6 = PT_LDS_STACK6, kPendLeaves6 = PT_PEND_LEAVES6;
static_assert(kLdsStack6 + kPendLeaves6 == kLdsStack + kPendLeaves, "the 6-wide kernels use the LDS budget of the 4-wide ones");
constexpr int kStackTotal = kLdsStack + kSpillStack;  // 96 entries, wherever the LDS part ends

struct TraversalStack {
  uint32_t* lds;     // &lds_stack[0][lane_in_block]
  int lds_stride;    // block size
  uint32_t* spill;   // &spill[0][global_thread]
  int spill_stride;  // threads in grid
  uint32_t* pend;    // &lds_pending[0][lane_in_block], stride lds_stride
  int sp = 0;
  int npend = 0;
  PT_HD void push_leaf(uint32_t ref) { pend[npend * lds_stride] = ref; npend++; }
  PT_HD uint32_t pop_leaf() { npend--; return pend[npend * lds_stride]; }
  // (ROWS = stack rows in LDS: kLdsStack, or kLdsStack6 in the 6-wide kernels — a compile-time constant: as a member it cost 40 bytes of scratch)
  template <int ROWS = kLdsStack> PT_HD void push(uint32_t v) {
    if (sp < ROWS) lds[sp * lds_stride] = v;
    else if (sp < kStackTotal) spill[(size_t)(sp - ROWS) * spill_stride] = v;
    sp++;
  }
  template <int ROWS = kLdsStack> PT_HD uint32_t pop() {
    sp--;
    if (sp < ROWS) return lds[sp * lds_stride];
    if (sp < kStackTotal) return spill[(size_t)(sp - ROWS) * spill_stride];
    return kInvalidRef;  // overflowed entries were dropped; never reached with depth <= 96
  }
};

struct TraversalCount { uint32_t nodes = 0, tris = 0, leaves = 0; };  // node visits, triangle tests, leaf-slot fetches (one or two tests each)

// Resumable traversal: one ray's state lives in registers (+ its LDS/HBM stack and leaf queue) and advances one node
// per trav_node() / one triangle per trav_pending_leaf(), so a persistent kernel can hand a finished lane a new ray
// while its neighbours keep going, and can run the triangle test for many lanes at once (see k_trace_closest).
struct TravState {
  vec3 o, d, inv;
  bool negx, negy, negz;  // direction signs: which side of a child box the ray enters through
  float tmin;
  RayHit best;     // best.t doubles as the current far limit
  uint32_t cur;    // node to visit next
  float payload;   // the `ir` sample handed to the alpha-test intersection function (kernel.metal:510, 625)
  TraversalStack st;
  // ---- two-level traversal only (dead in the one-BVH kernels): the ray in the space of the structure being walked ----
  vec3 co, cinv;          // origin and reciprocal direction there (== o, inv while in the TLAS); t is the same parameter in both spaces
  bool cnegx, cnegy, cnegz;
  float sx, sy, sz;       // slab slack in t per axis: (object-space position error bound of this ray in this instance) * |cinv|
  uint32_t tri_base;      // flattened index of the instance's first triangle: added to the BLAS leaf refs that are queued
#ifdef PT_DEBUG_PID
  bool dbg = false;       // debug build (tools/build_variant.sh dbg -DPT_DEBUG_PID): this lane's ray is the one $PTAMD_DEBUG_RAY names
#endif
};

// intersections.metal:8-39 alphaTestIntersectionFunction: runs for every candidate hit on a non-opaque instance;
// the hit counts iff alpha(baseColor.a x baseTexture.a at the interpolated uv) > payload.
PT_HD bool alpha_test(const DeviceScene& S, uint32_t instanceIdx, uint32_t prim, float u, float v, float r) {
  const InstanceInfo& inst = S.instances[instanceIdx];
  const MeshInfo mesh = S.meshes[inst.mesh];
  const pt_material_gpu& material = S.materials[inst.material_base + S.slots[mesh.tri_base + prim]];
  float alpha = material.baseColor[3];
  if (material.baseTextureId >= 0) {
    const uint32_t* __restrict__ idx = &S.indices[3 * (size_t)(mesh.tri_base + prim)];
    const float* t0 = S.vdata[mesh.vertex_base + idx[0]].texCoords;
    const float* t1 = S.vdata[mesh.vertex_base + idx[1]].texCoords;
    const float* t2 = S.vdata[mesh.vertex_base + idx[2]].texCoords;
    const float w = 1.0f - u - v;
    const vec2 uv = {(w * t0[0] + u * t1[0]) + v * t2[0], (w * t0[1] + u * t1[1]) + v * t2[1]};
    alpha = alpha * tex_sample(S, material.baseTextureId, uv).w;
  }
  return alpha > r;
}

// One candidate triangle against the ray's current best: the closest-hit rule (min t, ties to the lowest global id) or accept-any.
PT_HD bool trav_test(const DeviceScene& S, TravState& ts, vec3 v0, vec3 v1, vec3 v2, uint32_t gid, uint32_t inst, bool cutouts, uint32_t tri, bool any) {
  float t, u, v;
#if defined(PT_DEBUG_PID) && defined(__HIP_DEVICE_COMPILE__)
  if (ts.dbg) {
    float t2 = -1, u2 = -1, v2_ = -1;
    const bool h2 = intersect_triangle(ts.o, ts.d, ts.tmin, kInf, v0, v1 - v0, v2 - v0, &t2, &u2, &v2_);
    printf("dbg  test tri %u gid %u: hit(any t) %d t %.9g u %.9g v %.9g | best.t %.9g best.gid %u\n", tri, gid >> 2, (int)h2, t2, u2, v2_, ts.best.t, ts.best.gid >> 2);
  }
#endif
  if (!intersect_triangle(ts.o, ts.d, ts.tmin, ts.best.t, v0, v1 - v0, v2 - v0, &t, &u, &v)) return false;
  if (cutouts && !alpha_test(S, inst, (gid >> 2) - S.instances[inst].tri_global_base, u, v, ts.payload)) return false;
  if (any) { ts.best.tri = tri; return true; }
  // intersect_triangle admitted t <= best.t; equal t needs the id tie-break
  if (t < ts.best.t || ts.best.tri == kInvalidRef || gid < ts.best.gid) {
    ts.best.t = t; ts.best.u = u; ts.best.v = v; ts.best.tri = tri; ts.best.gid = gid;
  }
  return false;
}

// One leaf slot: triangle A, then triangle B when the slot holds a pair.  The order of the two tests does not matter (min t, ties to the
// lowest global id — the same rule that makes the answer independent of the tree).  B's corners are loaded after A's test (TriRec).
PT_HD void trav_leaf(const DeviceScene& S, TravState& ts, uint32_t ref, bool any, bool* finished, TraversalCount* cnt) {
  const uint32_t slot = ref & ~kLeafBit;
  const TriRec* rec = &S.tris[slot];
  uint32_t gid_b, inst_code;
  {
    const TriRec& tr = *rec;
    const vec3 v0 = v3(tr.q0[0], tr.q0[1], tr.q0[2]), v1 = v3(tr.q1[0], tr.q1[1], tr.q1[2]), v2 = v3(tr.q2[0], tr.q2[1], tr.q2[2]);
    const uint32_t gid_a = tr.gid_a;
    gid_b = tr.gid_b; inst_code = tr.inst_code;
    if (cnt) { cnt->leaves++; cnt->tris++; }
    const uint32_t inst = inst_code & kSlotInstMask;
    const bool cutouts = S.has_alpha && (S.instances[inst].flags & kInstanceNonOpaque);
    if (trav_test(S, ts, v0, v1, v2, gid_a, inst, cutouts, 2u * slot, any)) { *finished = true; return; }
  }
  if (gid_b == kInvalidRef) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_PAIR_HOIST)
  asm volatile("" : "+v"(gid_b), "+v"(inst_code) : "v"(ts.best.t));  // keep B's loads behind A's test (hoisted, they cost A's registers)
#endif
  {
    const vec3 v0 = slot_corner(rec, (inst_code >> 26) & 3u), v1 = slot_corner(rec, (inst_code >> 28) & 3u), v2 = slot_corner(rec, inst_code >> 30);
    if (cnt) cnt->tris++;
    const uint32_t inst = inst_code & kSlotInstMask;
    const bool cutouts = S.has_alpha && (S.instances[inst].flags & kInstanceNonOpaque);
    if (trav_test(S, ts, v0, v1, v2, gid_b, inst, cutouts, 2u * slot + 1u, any)) *finished = true;
  }
}

// Returns true when the ray is finished (ts.best holds the answer). `st` must be the lane's stack.
PT_HD bool trav_init(const DeviceScene& S, TravState& ts, vec3 o, vec3 d, float tmin, float tmax, float payload, TraversalStack st,
                     bool any, TraversalCount* cnt) {
  ts.o = o; ts.d = d; ts.tmin = tmin; ts.payload = payload;
  // A zero (or denormal) direction component would give inv = inf and 0 * inf = NaN in the slab arithmetic; NaNs are
  // dropped by fmin/fmax, i.e. the axis stops constraining anything and an axis-parallel ray walks a large part of the
  // tree (measured: environment-light shadow rays towards the top row of a lat-long map, direction exactly (0,1,0),
  // made k_trace_shadow 250x slower).  A huge finite reciprocal keeps the slab test exact in the limit.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PT_EXACT_RAY_RCP)
  // the reciprocal direction only feeds the slab tests, whose arithmetic has to be conservative, not reproducible (trav_node): v_rcp_f32
  // (1 ulp) instead of three IEEE divisions (~8 instructions each, run for the few lanes of a refill)
  ts.inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
#else
  ts.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
#if defined(PT_TEST_PERTURB_RCP) && !defined(__HIP_DEVICE_COMPILE__)
  PT_TEST_PERTURB_RCP(ts.inv);   // tests/emu only: the device's v_rcp_f32 may differ from 1 / d in the last bit — no answer may depend on it
#endif
#endif
  if (!(fabsf(ts.inv.x) <= 1e30f)) ts.inv.x = copysignf(1e30f, d.x);
  if (!(fabsf(ts.inv.y) <= 1e30f)) ts.inv.y = copysignf(1e30f, d.y);
  if (!(fabsf(ts.inv.z) <= 1e30f)) ts.inv.z = copysignf(1e30f, d.z);
  ts.negx = ts.inv.x < 0.0f; ts.negy = ts.inv.y < 0.0f; ts.negz = ts.inv.z < 0.0f;
  ts.best.t = tmax; ts.best.u = ts.best.v = 0.0f; ts.best.tri = kInvalidRef; ts.best.gid = kInvalidRef;
  ts.st = st;
  ts.st.sp = 0;
  ts.st.npend = 0;
  ts.cur = S.root_ref;
  ts.co = ts.o; ts.cinv = ts.inv;
  ts.cnegx = ts.negx; ts.cnegy = ts.negy; ts.cnegz = ts.negz;
  ts.sx = ts.sy = ts.sz = 0.0f;
  ts.tri_base = 0u;
  if (S.root_ref == kInvalidRef) return true;
  if (S.root_ref & kLeafBit) {  // single-triangle scene
    bool fin = false;
    trav_leaf(S, ts, S.root_ref, any, &fin, cnt);
    return true;
  }
  return false;
}

// ---- node (de)quantisation ---------------------------------------------------------------------------------------
struct Box3 { float lo[3], hi[3]; };

PT_HD float node_scale(uint8_t e) { return u2f((uint32_t)e << 23); }

// Builds one BvhNode from up to four exact child boxes (already inflated). Conservative by construction: every
// quantised coordinate is checked against the very expression the traversal evaluates.
PT_HD BvhNode quantize_node4(const Box3* boxes, const uint32_t* refs, int count) {
  BvhNode n;
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (int k = 0; k < count; k++)
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], boxes[k].lo[a]);
      hi[a] = fmaxf(hi[a], boxes[k].hi[a]);
    }
  float scale[3];
  for (int a = 0; a < 3; a++) {
    n.origin[a] = lo[a];
    // smallest power of two s with 255 * s >= extent (exponent clamped to normal floats)
    const float need = (hi[a] - lo[a]) * (1.0f / 255.0f);
    uint32_t e = (f2u(need) >> 23) & 0xffu;
    if ((f2u(need) & 0x7fffffu) != 0) e += 1;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    // rounding of (hi - lo) / 255 may leave 255 * s a hair short: bump until it covers
    while (e < 254 && lo[a] + 255.0f * node_scale((uint8_t)e) < hi[a]) e += 1;
    n.exp[a] = (uint8_t)e;
    scale[a] = node_scale((uint8_t)e);
  }
  n._pad0 = 0;
  n._pad1[0] = n._pad1[1] = 0;
  for (int a = 0; a < 3; a++) n.qlo[a] = n.qhi[a] = 0;
  for (int k = 0; k < 4; k++) {
    if (k >= count) {
      n.ref[k] = kInvalidRef;
      for (int a = 0; a < 3; a++) n.qlo[a] |= 255u << (8 * k);  // inverted box (lo 255, hi 0)
      continue;
    }
    n.ref[k] = refs[k];
    for (int a = 0; a < 3; a++) {
      const float inv = 1.0f / scale[a];
      int ql = (int)floorf((boxes[k].lo[a] - lo[a]) * inv);
      ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
      while (ql > 0 && lo[a] + (float)ql * scale[a] > boxes[k].lo[a]) ql--;
      int qh = (int)ceilf((boxes[k].hi[a] - lo[a]) * inv);
      qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
      while (qh < 255 && lo[a] + (float)qh * scale[a] < boxes[k].hi[a]) qh++;
      n.qlo[a] |= (uint32_t)ql << (8 * k);
      n.qhi[a] |= (uint32_t)qh << (8 * k);
    }
  }
  return n;
}

// Conservative inflation applied to every exact box before it is quantised into its parent: the slab test must
// never cull a triangle the Moeller-Trumbore test accepts (intersection contract). 8e-6 relative is
// ~64 ulp of the coordinate magnitude.
PT_HD Box3 inflate_box(const Box3& b) {
  Box3 r;
  for (int a = 0; a < 3; a++) {
    const float m = fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a]));
    const float eps = m * 8e-6f + 1e-30f;
    r.lo[a] = b.lo[a] - eps;
    r.hi[a] = b.hi[a] + eps;
  }
  return r;
}

// One node per call (ts.cur must be a node and the leaf queue must have room for four entries).  The node's four child
// slabs are evaluated in "ray space": with A = scale * inv_d and B = (origin - o) * inv_d per axis, t(q) = q * A + B
// (3 VALU per coordinate instead of 5).  This rounds differently from the builder's origin + q * scale, by a few ulp of
// t; the slab slack and the 8e-6 box inflation absorb that.  Leaf children that pass the slab test are NOT tested here:
// they go to the lane's leaf queue, and the caller runs the triangle code when enough lanes of the wave have one queued
// (measured before this split: the in-place triangle loop ran at 11 % lane utilisation on C2, 8 % on C3).
// On return ts.cur is the next node, or kInvalidRef when the node stack is exhausted.
template <bool COUNT, bool TWO = false>
PT_HD void trav_node(const BvhNode* __restrict__ nodes, TravState& ts, TraversalCount* cnt) {
  const BvhNode n = nodes[ts.cur];
  if (COUNT) cnt->nodes++;
  const vec3 ro = TWO ? ts.co : ts.o, ri = TWO ? ts.cinv : ts.inv;
  const bool gx = TWO ? ts.cnegx : ts.negx, gy = TWO ? ts.cnegy : ts.negy, gz = TWO ? ts.cnegz : ts.negz;
  const float ax = node_scale(n.exp[0]) * ri.x, ay = node_scale(n.exp[1]) * ri.y, az = node_scale(n.exp[2]) * ri.z;
  const float bx = (n.origin[0] - ro.x) * ri.x, by = (n.origin[1] - ro.y) * ri.y, bz = (n.origin[2] - ro.z) * ri.z;
  // two-level: the object-space ray carries a rounding error; every slab is widened by its bound (zero while in the TLAS)
  const float bnx = TWO ? bx - ts.sx : bx, bny = TWO ? by - ts.sy : by, bnz = TWO ? bz - ts.sz : bz;
  const float bfx = TWO ? bx + ts.sx : bx, bfy = TWO ? by + ts.sy : by, bfz = TWO ? bz + ts.sz : bz;
  // entry / exit planes per axis by direction sign (one select per dword instead of a min/max pair per child)
  const uint32_t nx = gx ? n.qhi[0] : n.qlo[0], fx = gx ? n.qlo[0] : n.qhi[0];
  const uint32_t ny = gy ? n.qhi[1] : n.qlo[1], fy = gy ? n.qlo[1] : n.qhi[1];
  const uint32_t nz = gz ? n.qhi[2] : n.qlo[2], fz = gz ? n.qlo[2] : n.qhi[2];
  const float far_lim = ts.best.t * kCullSlack;
  float dist[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    // (fused multiply-adds: this arithmetic only has to be conservative, not reproducible — a NaN from 0 * inf is
    //  dropped by fmax/fmin, which widens the box)
    const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 0xffu), ax, bnx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 0xffu), ax, bfx);
    const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 0xffu), ay, bny), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 0xffu), ay, bfy);
    const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 0xffu), az, bnz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 0xffu), az, bfz);
    const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), ts.tmin);
    const float tf = fminf(fminf(fminf(tfx, tfy), tfz), far_lim);
    const bool hit = n.ref[k] != kInvalidRef && tn <= __builtin_fmaf(tf, kSlabSlack, 1e-30f);
    const bool leaf = hit && (n.ref[k] & kLeafBit);
    dist[k] = hit && !leaf ? tn : kInf;  // leaves never go on the node stack (TLAS leaves — kInstBit — do: they are entered like nodes)
    // queue the leaf (branch-free: a non-leaf writes to the scratch row kPendLeaves); two-level: as the flattened triangle
    ts.st.pend[(leaf ? ts.st.npend : kPendLeaves) * ts.st.lds_stride] = TWO ? n.ref[k] + ts.tri_base : n.ref[k];
    ts.st.npend += leaf ? 1 : 0;
  }
  // internal children: nearest becomes `cur`, the others are pushed far-to-near
  // (5-comparator sorting network on (dist, ref) pairs; misses carry +inf and sink to the end)
  float d0 = dist[0], d1 = dist[1], d2 = dist[2], d3 = dist[3];
  uint32_t r0 = n.ref[0], r1 = n.ref[1], r2 = n.ref[2], r3 = n.ref[3];
#define PT_CSWAP(da, ra, db, rb) { const bool sw = db < da; const float td = sw ? db : da; const uint32_t tr = sw ? rb : ra; \
                                   db = sw ? da : db; rb = sw ? ra : rb; da = td; ra = tr; }
  PT_CSWAP(d0, r0, d1, r1) PT_CSWAP(d2, r2, d3, r3) PT_CSWAP(d0, r0, d2, r2) PT_CSWAP(d1, r1, d3, r3) PT_CSWAP(d1, r1, d2, r2)
#undef PT_CSWAP
  if (d0 < kInf) {
    // push far-to-near (a branch-free variant that writes all three entries and redirects misses to the scratch row was
    // measured: neutral on C2, 6 % slower on C3 where deep stacks take the spill path more often)
    if (d3 < kInf) ts.st.push(r3);
    if (d2 < kInf) ts.st.push(r2);
    if (d1 < kInf) ts.st.push(r1);
    ts.cur = r0;
  } else {
    ts.cur = ts.st.sp == 0 ? kInvalidRef : ts.st.pop();
  }
}

// ---- 6-wide nodes (BvhNode6) ------------------------------------------------------------------------------------------------
// Builds one BvhNode6 from up to six exact child boxes (already inflated), the internal children first.  Same quantisation rule as
// quantize_node4: conservative against the very expression the traversal evaluates.
PT_HD BvhNode6 quantize_node6(const Box3* boxes, int n_int, int n_leaf, uint32_t base_node, uint32_t base_leaf) {
  BvhNode6 n;
  const int count = n_int + n_leaf;
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (int k = 0; k < count; k++)
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], boxes[k].lo[a]);
      hi[a] = fmaxf(hi[a], boxes[k].hi[a]);
    }
  n.counts = (uint8_t)(n_int | (n_leaf << 3));
  n.base_node = base_node;
  n.base_leaf = base_leaf;
  n._pad = 0;
  for (int a = 0; a < 3; a++) {
    n.origin[a] = lo[a];
    const float need = (hi[a] - lo[a]) * (1.0f / 255.0f);
    uint32_t e = (f2u(need) >> 23) & 0xffu;
    if ((f2u(need) & 0x7fffffu) != 0) e += 1;
    if (e < 1) e = 1;
    if (e > 254) e = 254;
    while (e < 254 && lo[a] + 255.0f * node_scale((uint8_t)e) < hi[a]) e += 1;
    n.exp[a] = (uint8_t)e;
    const float scale = node_scale((uint8_t)e), inv = 1.0f / scale;
    uint32_t qlo[6], qhi[6];
    for (int k = 0; k < 6; k++) {
      if (k >= count) { qlo[k] = 255u; qhi[k] = 0u; continue; }  // inverted box
      int ql = (int)floorf((boxes[k].lo[a] - lo[a]) * inv);
      ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
      while (ql > 0 && lo[a] + (float)ql * scale > boxes[k].lo[a]) ql--;
      int qh = (int)ceilf((boxes[k].hi[a] - lo[a]) * inv);
      qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
      while (qh < 255 && lo[a] + (float)qh * scale < boxes[k].hi[a]) qh++;
      qlo[k] = (uint32_t)ql; qhi[k] = (uint32_t)qh;
    }
    n.q[a][0] = qlo[0] | qlo[1] << 8 | qlo[2] << 16 | qlo[3] << 24;
    n.q[a][1] = qhi[0] | qhi[1] << 8 | qhi[2] << 16 | qhi[3] << 24;
    n.q[a][2] = qlo[4] | qlo[5] << 8 | qhi[4] << 16 | qhi[5] << 24;
  }
  return n;
}

// One 6-wide node per call (ts.cur must be a node and the leaf queue must have room for ONE entry: the leaf children that pass the slab
// test are queued together as base_leaf << 6 | mask, the triangle rounds take them out one bit at a time).  The nearest internal child
// becomes `cur`, the others are pushed in slot order (host probe, C3: 13.69 nodes per ray against 13.44 fully sorted and 16.95 with
// 4-wide nodes; the 6-element sorting network would cost ~35 VALU and six more live registers).
template <bool COUNT>
PT_HD void trav_node6(const BvhNode* __restrict__ nodes, TravState& ts, TraversalCount* cnt) {
  const BvhNode6 n = reinterpret_cast<const BvhNode6*>(nodes)[ts.cur];
  if (COUNT) cnt->nodes++;
  const float ax = node_scale(n.exp[0]) * ts.inv.x, ay = node_scale(n.exp[1]) * ts.inv.y, az = node_scale(n.exp[2]) * ts.inv.z;
  const float bx = (n.origin[0] - ts.o.x) * ts.inv.x, by = (n.origin[1] - ts.o.y) * ts.inv.y, bz = (n.origin[2] - ts.o.z) * ts.inv.z;
  // entry / exit planes per axis by direction sign: children 0..3 one select per dword, children 4, 5 the halves of the third dword
  const uint32_t nx = ts.negx ? n.q[0][1] : n.q[0][0], fx = ts.negx ? n.q[0][0] : n.q[0][1];
  const uint32_t ny = ts.negy ? n.q[1][1] : n.q[1][0], fy = ts.negy ? n.q[1][0] : n.q[1][1];
  const uint32_t nz = ts.negz ? n.q[2][1] : n.q[2][0], fz = ts.negz ? n.q[2][0] : n.q[2][1];
  const uint32_t nx2 = ts.negx ? n.q[0][2] >> 16 : n.q[0][2], fx2 = ts.negx ? n.q[0][2] : n.q[0][2] >> 16;
  const uint32_t ny2 = ts.negy ? n.q[1][2] >> 16 : n.q[1][2], fy2 = ts.negy ? n.q[1][2] : n.q[1][2] >> 16;
  const uint32_t nz2 = ts.negz ? n.q[2][2] >> 16 : n.q[2][2], fz2 = ts.negz ? n.q[2][2] : n.q[2][2] >> 16;
  const uint32_t n_int = n.counts & 7u, count = n_int + ((n.counts >> 3) & 7u);
  const float far_lim = ts.best.t * kCullSlack;
  float best_d = kInf;
  uint32_t best_k = 0, hits = 0;  // hits: bit k = child k passed the slab test
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint32_t qnx = k < 4 ? (nx >> (8 * k)) & 0xffu : (nx2 >> (8 * (k - 4))) & 0xffu, qfx = k < 4 ? (fx >> (8 * k)) & 0xffu : (fx2 >> (8 * (k - 4))) & 0xffu;
    const uint32_t qny = k < 4 ? (ny >> (8 * k)) & 0xffu : (ny2 >> (8 * (k - 4))) & 0xffu, qfy = k < 4 ? (fy >> (8 * k)) & 0xffu : (fy2 >> (8 * (k - 4))) & 0xffu;
    const uint32_t qnz = k < 4 ? (nz >> (8 * k)) & 0xffu : (nz2 >> (8 * (k - 4))) & 0xffu, qfz = k < 4 ? (fz >> (8 * k)) & 0xffu : (fz2 >> (8 * (k - 4))) & 0xffu;
    const float tnx = __builtin_fmaf((float)qnx, ax, bx), tfx = __builtin_fmaf((float)qfx, ax, bx);
    const float tny = __builtin_fmaf((float)qny, ay, by), tfy = __builtin_fmaf((float)qfy, ay, by);
    const float tnz = __builtin_fmaf((float)qnz, az, bz), tfz = __builtin_fmaf((float)qfz, az, bz);
    const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), ts.tmin);
    const float tf = fminf(fminf(fminf(tfx, tfy), tfz), far_lim);
    const bool hit = (uint32_t)k < count && tn <= __builtin_fmaf(tf, kSlabSlack, 1e-30f);
    hits |= hit ? 1u << k : 0u;
    const bool nearer = hit && (uint32_t)k < n_int && tn < best_d;
    best_d = nearer ? tn : best_d;
    best_k = nearer ? (uint32_t)k : best_k;
  }
#if defined(PT_DEBUG_PID) && defined(__HIP_DEVICE_COMPILE__)
  if (ts.dbg) printf("dbg node %u: n_int %u count %u hits %x base_node %u base_leaf %u best.t %.9g\n", ts.cur, n_int, count, hits, n.base_node, n.base_leaf, ts.best.t);
#endif
  // leaf children that were hit: one queue entry (branch-free: no entry -> the scratch row)
  const uint32_t leaf_mask = hits >> n_int;
  ts.st.pend[(leaf_mask ? ts.st.npend : kPendLeaves6) * ts.st.lds_stride] = n.base_leaf << 6 | leaf_mask;
  ts.st.npend += leaf_mask ? 1 : 0;
  // internal children: the nearest is visited next, the others go on the stack
  uint32_t others = hits & ((1u << n_int) - 1u);
  if (others) {
    others &= ~(1u << best_k);
    while (others) {
      const uint32_t k = (uint32_t)__builtin_ctz(others);
      others &= others - 1u;
      ts.st.push<kLdsStack6>(n.base_node + k);
    }
    ts.cur = n.base_node + best_k;
  } else {
    ts.cur = ts.st.sp == 0 ? kInvalidRef : ts.st.pop<kLdsStack6>();
  }
}

// Tests ONE triangle of the newest leaf-queue entry (6-wide structure: an entry is base_leaf << 6 | mask of the node's leaf children
// still to be tested).  Returns true when an any-hit ray is finished by it.
template <bool ANY, bool COUNT>
PT_HD bool trav_pending_leaf6(const DeviceScene& S, TravState& ts, TraversalCount* cnt) {
  uint32_t* top = &ts.st.pend[(ts.st.npend - 1) * ts.st.lds_stride];
  const uint32_t e = *top;
  const uint32_t mask = e & 63u, r = (uint32_t)__builtin_ctz(mask), rest = mask & (mask - 1u);
  if (rest) *top = (e & ~63u) | rest; else ts.st.npend--;
  bool finished = false;
  trav_leaf(S, ts, kLeafBit | ((e >> 6) + r), ANY, &finished, COUNT ? cnt : nullptr);
  return finished;
}

// ---- two-level traversal: entering and leaving an instance ------------------------------------------------------------
// Called when ts.cur is not a plain node: an exit marker (back to world space) or a TLAS leaf (kInstBit | instance).
// Entering takes the ray to the instance's object space for the BLAS slab tests: co = M^-1 (o - c), cd = M^-1 d, the SAME
// t.  Those are rounded values; the distance between the computed object-space ray and the exact one at parameter t is at
// most ~4 ulp of sum_j |M^-1_ij| (|o_j - c_j| + t |d_j|).  The BLAS slabs are widened by e = 8e-6 * that sum at the far end
// of the instance's bounds (the same relative margin the world-space boxes carry, 30x the rounding bound), so the walk still
// reaches a superset of the triangles the WORLD-space Moeller-Trumbore test can accept: closest hits do not depend on which
// structure was walked.  Needs room for one leaf in the lane's queue (a one-triangle mesh has no nodes).
// trav_leave: the cheap half (exit markers) — run by every lane as soon as one comes up.
PT_HD void trav_leave(TravState& ts) {
  while (ts.cur == kExitMarker) {
    ts.co = ts.o; ts.cinv = ts.inv;
    ts.cnegx = ts.negx; ts.cnegy = ts.negy; ts.cnegz = ts.negz;
    ts.sx = ts.sy = ts.sz = 0.0f;
    ts.tri_base = 0u;
    ts.cur = ts.st.sp == 0 ? kInvalidRef : ts.st.pop();
  }
}
PT_HD void trav_resolve(const DeviceScene& S, TravState& ts) {
  for (;;) {
    if (ts.cur == kExitMarker) {
      ts.co = ts.o; ts.cinv = ts.inv;
      ts.cnegx = ts.negx; ts.cnegy = ts.negy; ts.cnegz = ts.negz;
      ts.sx = ts.sy = ts.sz = 0.0f;
      ts.tri_base = 0u;
      ts.cur = ts.st.sp == 0 ? kInvalidRef : ts.st.pop();
      continue;
    }
    if (ts.cur == kInvalidRef || !(ts.cur & kInstBit)) return;
    if (ts.st.npend > kPendLeaves - 4) return;  // (no room in the leaf queue: the caller runs a triangle round first)
    const InstanceTrav it = ldg(&S.inst_trav[ts.cur & ~kInstBit]);
    const MeshTrav mt = ldg(&S.mesh_trav[it.mesh]);
    if (mt.root_ref == kInvalidRef || (mt.root_ref & kLeafBit)) {
      // a mesh without triangles, or with one: that one is the candidate (no nodes to walk, the ray stays where it is)
      if (mt.root_ref != kInvalidRef) ts.st.push_leaf(kLeafBit | (it.tri_base + (mt.root_ref & ~kLeafBit)));
      ts.cur = ts.st.sp == 0 ? kInvalidRef : ts.st.pop();
      continue;
    }
    const vec3 p = ts.o - v3(it.c[0], it.c[1], it.c[2]);
    const vec3 i0 = v3(it.ic0[0], it.ic0[1], it.ic0[2]), i1 = v3(it.ic1[0], it.ic1[1], it.ic1[2]), i2 = v3(it.ic2[0], it.ic2[1], it.ic2[2]);
    ts.co = (i0 * p.x + i1 * p.y) + i2 * p.z;
    const vec3 cd = (i0 * ts.d.x + i1 * ts.d.y) + i2 * ts.d.z;
    ts.cinv = v3(1.0f / cd.x, 1.0f / cd.y, 1.0f / cd.z);
    if (!(fabsf(ts.cinv.x) <= 1e30f)) ts.cinv.x = copysignf(1e30f, cd.x);
    if (!(fabsf(ts.cinv.y) <= 1e30f)) ts.cinv.y = copysignf(1e30f, cd.y);
    if (!(fabsf(ts.cinv.z) <= 1e30f)) ts.cinv.z = copysignf(1e30f, cd.z);
    ts.cnegx = ts.cinv.x < 0.0f; ts.cnegy = ts.cinv.y < 0.0f; ts.cnegz = ts.cinv.z < 0.0f;
    // where the ray leaves the mesh's bounds (the largest t any hit in this instance can have)
    const float fx = ((ts.cnegx ? mt.lo[0] : mt.hi[0]) - ts.co.x) * ts.cinv.x;
    const float fy = ((ts.cnegy ? mt.lo[1] : mt.hi[1]) - ts.co.y) * ts.cinv.y;
    const float fz = ((ts.cnegz ? mt.lo[2] : mt.hi[2]) - ts.co.z) * ts.cinv.z;
    const float t1 = fminf(fminf(fx, fy), fz);
    const float tfar = fminf(ts.best.t, fmaxf(t1, 0.0f) * 1.0001f + 1e-6f);
    const vec3 ap = v3(fabsf(p.x), fabsf(p.y), fabsf(p.z)), ad = v3(fabsf(ts.d.x), fabsf(ts.d.y), fabsf(ts.d.z));
    const vec3 a0 = v3(fabsf(i0.x), fabsf(i0.y), fabsf(i0.z)), a1 = v3(fabsf(i1.x), fabsf(i1.y), fabsf(i1.z)), a2 = v3(fabsf(i2.x), fabsf(i2.y), fabsf(i2.z));
    const vec3 mp = (a0 * ap.x + a1 * ap.y) + a2 * ap.z, md = (a0 * ad.x + a1 * ad.y) + a2 * ad.z;
    float e = 8e-6f * (fmaxf(mp.x, fmaxf(mp.y, mp.z)) + tfar * fmaxf(md.x, fmaxf(md.y, md.z))) + 1e-30f;
    if (!(e <= 1e30f)) e = 1e30f;
    ts.sx = e * fabsf(ts.cinv.x); ts.sy = e * fabsf(ts.cinv.y); ts.sz = e * fabsf(ts.cinv.z);
    ts.tri_base = it.tri_base;
    ts.st.push(kExitMarker);
    ts.cur = mt.root_ref;
  }
}

// Tests one queued leaf.  Returns true when an any-hit ray is finished by it.
template <bool ANY, bool COUNT>
PT_HD bool trav_pending_leaf(const DeviceScene& S, TravState& ts, TraversalCount* cnt) {
  bool finished = false;
  trav_leaf(S, ts, ts.st.pop_leaf(), ANY, &finished, COUNT ? cnt : nullptr);
  return finished;
}

// One node, then its leaves at once (the scalar formulation: tests/emu, and the reference for the wave-cooperative
// scheduling in kernels.hip — the answer does not depend on the order in which candidates are tested).
template <bool ANY, bool COUNT>
PT_HD bool trav_step(const DeviceScene& S, TravState& ts, TraversalCount* cnt) {
  if (S.wide6) {
    trav_node6<COUNT>(S.nodes, ts, cnt);
    while (ts.st.npend > 0)
      if (trav_pending_leaf6<ANY, COUNT>(S, ts, cnt)) return true;
    return ts.cur == kInvalidRef;
  }
  trav_node<COUNT>(S.nodes, ts, cnt);
  while (ts.st.npend > 0)
    if (trav_pending_leaf<ANY, COUNT>(S, ts, cnt)) return true;
  return ts.cur == kInvalidRef;
}

template <bool ANY, bool COUNT>
PT_HD RayHit traverse(const DeviceScene& S, vec3 o, vec3 d, float tmin, float tmax, float payload, TraversalStack st,
                      TraversalCount* cnt) {
  TravState ts;
  if (trav_init(S, ts, o, d, tmin, tmax, payload, st, ANY, COUNT ? cnt : nullptr)) return ts.best;
  while (!trav_step<ANY, COUNT>(S, ts, cnt)) {}
  return ts.best;
}

}  // namespace pt
