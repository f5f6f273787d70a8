// pt_bvh.h — ray / triangle test and BVH2 traversal (closest-hit and any-hit).
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
#include "pt_device.h"

namespace pt {

struct RayHit {
  float t, u, v;
  uint32_t tri;  // index into DeviceScene::tris (leaf order); kInvalidRef = miss
  uint32_t gid;
};

// Moeller-Trumbore with a fixed operation order.  Returns true and (t,u,v) when det != 0, 0 <= u <= 1, 0 <= v,
// u + v <= 1 and tmin <= t <= tmax.
PT_HD bool intersect_triangle(vec3 o, vec3 d, float tmin, float tmax, const TriRec& tr, float* t_out, float* u_out,
                              float* v_out) {
  const vec3 e1 = v3(tr.e1[0], tr.e1[1], tr.e1[2]);
  const vec3 e2 = v3(tr.e2[0], tr.e2[1], tr.e2[2]);
  const vec3 p = cross(d, e2);
  const float det = dot(e1, p);
  if (det == 0.0f) return false;
  const float inv = 1.0f / det;
  const vec3 s = o - v3(tr.v0[0], tr.v0[1], tr.v0[2]);
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

// Per-lane traversal stack: the first kLdsStack entries live in LDS ([depth][lane], bank = lane), the rest spill
// to a per-thread HBM slab (only pathological trees get there).
constexpr int kLdsStack = 24;
constexpr int kSpillStack = 72;  // total depth 96 >= any LBVH depth over 63-bit codes + index tie-break

struct TraversalStack {
  uint32_t* lds;     // &lds_stack[0][lane_in_block]
  int lds_stride;    // block size
  uint32_t* spill;   // &spill[0][global_thread]
  int spill_stride;  // threads in grid
  int sp = 0;
  PT_HD void push(uint32_t v) {
    if (sp < kLdsStack) lds[sp * lds_stride] = v;
    else if (sp < kLdsStack + kSpillStack) spill[(size_t)(sp - kLdsStack) * spill_stride] = v;
    sp++;
  }
  PT_HD uint32_t pop() {
    sp--;
    if (sp < kLdsStack) return lds[sp * lds_stride];
    if (sp < kLdsStack + kSpillStack) return spill[(size_t)(sp - kLdsStack) * spill_stride];
    return kInvalidRef;  // overflowed entries were dropped; never reached with depth <= 96
  }
};

struct TraversalCount { uint32_t nodes = 0, tris = 0; };

// Resumable traversal: one ray's state lives in registers (+ its LDS/HBM stack) and advances one node per
// trav_step(), so a persistent kernel can hand a finished lane a new ray while its neighbours keep going.
struct TravState {
  vec3 o, d, inv;
  float tmin;
  RayHit best;     // best.t doubles as the current far limit
  uint32_t cur;    // node to visit next
  TraversalStack st;
};

PT_HD void trav_leaf(const DeviceScene& S, TravState& ts, uint32_t ref, bool any, bool* finished, TraversalCount* cnt) {
  const uint32_t ti = ref & ~kLeafBit;
  const TriRec tr = S.tris[ti];
  if (cnt) cnt->tris++;
  float t, u, v;
  if (!intersect_triangle(ts.o, ts.d, ts.tmin, ts.best.t, tr, &t, &u, &v)) return;
  if (any) {
    ts.best.tri = ti;
    *finished = true;
    return;
  }
  // intersect_triangle admitted t <= best.t; equal t needs the id tie-break
  if (t < ts.best.t || ts.best.tri == kInvalidRef || tr.gid < ts.best.gid) {
    ts.best.t = t; ts.best.u = u; ts.best.v = v; ts.best.tri = ti; ts.best.gid = tr.gid;
  }
}

// Returns true when the ray is finished (ts.best holds the answer). `st` must be the lane's stack.
PT_HD bool trav_init(const DeviceScene& S, TravState& ts, vec3 o, vec3 d, float tmin, float tmax, TraversalStack st, bool any,
                     TraversalCount* cnt) {
  ts.o = o; ts.d = d; ts.tmin = tmin;
  ts.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  ts.best.t = tmax; ts.best.u = ts.best.v = 0.0f; ts.best.tri = kInvalidRef; ts.best.gid = kInvalidRef;
  ts.st = st;
  ts.st.sp = 0;
  ts.cur = S.root_ref;
  if (S.root_ref == kInvalidRef) return true;
  if (S.root_ref & kLeafBit) {  // single-triangle scene
    bool fin = false;
    trav_leaf(S, ts, S.root_ref, any, &fin, cnt);
    return true;
  }
  return false;
}

template <bool ANY, bool COUNT>
PT_HD bool trav_step(const DeviceScene& S, TravState& ts, TraversalCount* cnt) {
  const BvhNode n = S.nodes[ts.cur];
  if (COUNT) cnt->nodes++;
  float d0 = slab_entry(n.lo0, n.hi0, ts.o, ts.inv, ts.tmin, ts.best.t);
  float d1 = slab_entry(n.lo1, n.hi1, ts.o, ts.inv, ts.tmin, ts.best.t);
  const uint32_t r0 = n.ref0, r1 = n.ref1;
  bool finished = false;
  // leaves are tested on the spot; they never go on the stack
  if (d0 >= 0.0f && (r0 & kLeafBit)) {
    trav_leaf(S, ts, r0, ANY, &finished, COUNT ? cnt : nullptr);
    if (finished) return true;
    d0 = -1.0f;
  }
  if (d1 >= 0.0f && (r1 & kLeafBit)) {
    trav_leaf(S, ts, r1, ANY, &finished, COUNT ? cnt : nullptr);
    if (finished) return true;
    d1 = -1.0f;
  }
  if (d0 >= 0.0f && d1 >= 0.0f) {
    // both internal children hit: descend into the nearer, defer the farther
    const bool first0 = d0 <= d1;
    ts.st.push(first0 ? r1 : r0);
    ts.cur = first0 ? r0 : r1;
  } else if (d0 >= 0.0f) {
    ts.cur = r0;
  } else if (d1 >= 0.0f) {
    ts.cur = r1;
  } else {
    if (ts.st.sp == 0) return true;
    ts.cur = ts.st.pop();
    if (ts.cur == kInvalidRef) return true;
  }
  return false;
}

template <bool ANY, bool COUNT>
PT_HD RayHit traverse(const DeviceScene& S, vec3 o, vec3 d, float tmin, float tmax, TraversalStack st,
                      TraversalCount* cnt) {
  TravState ts;
  if (trav_init(S, ts, o, d, tmin, tmax, st, ANY, COUNT ? cnt : nullptr)) return ts.best;
  while (!trav_step<ANY, COUNT>(S, ts, cnt)) {}
  return ts.best;
}

}  // namespace pt
