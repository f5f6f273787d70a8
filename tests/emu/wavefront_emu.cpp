// TEST HARNESS ONLY (tests/emu) — runs the product's per-path stage functions (platinum_amd/csrc/pt_*.h, which are
// plain C++ under PT_HD) on the HOST, so that stage logic can be compared with the oracle in the GPU-less
// authoring container.  It is not part of libptamd.so, is never loaded by platinum_amd, and is not a fallback:
// it exists to debug bit-exactness before spending GPU time.  Queues, compaction and the LBVH build (the
// __global__ kernels) are NOT covered here — only `-m gpu` tests exercise those.
//
// The BVH used here is a throw-away host median-split builder that emits the product's BvhNode/TriRec layout.
#include <algorithm>
#include <functional>
#include <cstdio>
#include <cstring>
#include <vector>

// $EMU_RCP_ULP = -1 / +1 (every component) / 2 (+-1 alternating by axis and sign; any even value: +- half of it): the ray's reciprocal direction moved by one ulp before the slab tests, as
// the device's v_rcp_f32 may: hits and radiance must not change (tests/test_stage_functions_host.py)
#include "../../platinum_amd/csrc/pt_math.h"
static int g_emu_rcp_ulp = 0;
static inline void emu_perturb_rcp(pt::vec3& inv) {
  if (!g_emu_rcp_ulp) return;
  float* c[3] = {&inv.x, &inv.y, &inv.z};
  for (int a = 0; a < 3; a++) {
    uint32_t b = pt::f2u(*c[a]);
    if ((b & 0x7f800000u) == 0x7f800000u || (b & 0x7fffffffu) == 0) continue;
    const int dir = (g_emu_rcp_ulp & 1) ? g_emu_rcp_ulp : (((a + (b >> 31)) & 1) ? g_emu_rcp_ulp / 2 : -g_emu_rcp_ulp / 2);   // odd: uniform; even: alternating +- half
    b += (uint32_t)dir;   // magnitude +- 1 ulp
    *c[a] = pt::u2f(b);
  }
}
#define PT_TEST_PERTURB_RCP(v) emu_perturb_rcp(v)
#include "../../platinum_amd/csrc/host_scene.h"
#include "../../platinum_amd/csrc/pt_shade.h"

using namespace pt;

namespace {

static unsigned long long g_nodes = 0, g_tris = 0, g_rays = 0, g_leaves = 0;  // traversal statistics of emu_debug_sample (experiments)
static unsigned long long g_pb_nodes[16] = {0}, g_pb_tris[16] = {0}, g_pb_leaves[16] = {0}, g_pb_rays[16] = {0};  // ... per bounce

struct Emu {
  HostScene hs;
  std::vector<float> lut;
  std::vector<HaltonEntry> halton;
  std::vector<TriRec> tris;
  std::vector<ShadeRec> shade_recs;
  std::vector<LightRec> light_recs;
  std::vector<float> light_cdf;
  std::vector<BvhNode> nodes;
  DeviceScene S{};
  pt_render_params params{};
  // experiment (EMU_WIDE_PROBE): N-wide trees collapsed from the same binary tree, one per width
  struct WideNode { int count; Box3 box[8]; uint32_t ref[8]; };  // ref: kLeafBit | triangle (tris[] order) or index into wide[]
  std::vector<WideNode> wide[9];
  uint32_t wide_root[9] = {0};
  std::vector<std::vector<uint32_t>> groups;  // multi-triangle leaves of the probe trees: ref = kLeafBit | 0x40000000 | group index
};

// The r3 probes below walk trees whose leaf slots hold ONE triangle (they are only run without EMU_PAIRS): triangle A of a slot
static bool probe_intersect(vec3 o, vec3 d, float tmin, float tmax, const TriRec& tr, float* t, float* u, float* v) {
  const vec3 v0 = v3(tr.q0[0], tr.q0[1], tr.q0[2]), v1 = v3(tr.q1[0], tr.q1[1], tr.q1[2]), v2 = v3(tr.q2[0], tr.q2[1], tr.q2[2]);
  return intersect_triangle(o, d, tmin, tmax, v0, v1 - v0, v2 - v0, t, u, v);
}

Box3 tri_box(const TriRec& t, const vec3& v1, const vec3& v2) {
  Box3 b;
  const float* a = t.q0; const float* p1 = &v1.x; const float* p2 = &v2.x;
  for (int k = 0; k < 3; k++) { b.lo[k] = std::min(a[k], std::min(p1[k], p2[k])); b.hi[k] = std::max(a[k], std::max(p1[k], p2[k])); }
  return b;
}

// Binary median-split tree first (host only), then the same even-depth collapse + quantize_node4 the GPU builder uses.
struct BinNode { uint32_t left, right; Box3 box; };

uint32_t build_bin(std::vector<BinNode>& bin, std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t first, uint32_t count) {
  if (count == 1) return kLeafBit | first;
  Box3 bb; for (int k = 0; k < 3; k++) { bb.lo[k] = 1e30f; bb.hi[k] = -1e30f; }
  for (uint32_t i = first; i < first + count; i++)
    for (int k = 0; k < 3; k++) { bb.lo[k] = std::min(bb.lo[k], boxes[order[i]].lo[k]); bb.hi[k] = std::max(bb.hi[k], boxes[order[i]].hi[k]); }
  int axis = 0; float ext = -1;
  for (int k = 0; k < 3; k++) if (bb.hi[k] - bb.lo[k] > ext) { ext = bb.hi[k] - bb.lo[k]; axis = k; }
  uint32_t mid = first + count / 2;
  std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t a, uint32_t b) {
    return boxes[a].lo[axis] + boxes[a].hi[axis] < boxes[b].lo[axis] + boxes[b].hi[axis]; });
  uint32_t idx = (uint32_t)bin.size();
  bin.push_back({});
  uint32_t l = build_bin(bin, order, boxes, first, mid - first);
  uint32_t r = build_bin(bin, order, boxes, mid, first + count - mid);
  bin[idx] = {l, r, bb};
  return idx;
}

static float half_area(const Box3& b) { const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x * y + y * z + z * x; }

// LBVH-like topology (experiments: EMU_MORTON=1): sort by 63-bit Morton code of the box centre, split ranges at the highest
// differing bit (what the Karras tree of lbvh.hip encodes).
static uint64_t expand21(uint64_t v) { v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; }
uint32_t build_morton(std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<uint64_t>& keys, const std::vector<Box3>& boxes,
                      uint32_t first, uint32_t count) {
  if (count == 1) return kLeafBit | first;
  uint32_t split = first + count / 2;
  const uint64_t a = keys[first], b = keys[first + count - 1];
  if (a != b) {
    const int bit = 63 - __builtin_clzll(a ^ b);
    uint32_t lo = first, hi = first + count - 1;  // first index whose `bit` is set
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if ((keys[mid] >> bit) & 1) hi = mid; else lo = mid + 1; }
    split = lo;
  }
  const uint32_t idx = (uint32_t)bin.size();
  bin.push_back({});
  const uint32_t l = build_morton(bin, order, keys, boxes, first, split - first);
  const uint32_t r = build_morton(bin, order, keys, boxes, split, first + count - split);
  Box3 bb;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? boxes[order[ref & ~kLeafBit]] : bin[ref].box; };
  const Box3 bl = box_of(l), br = box_of(r);
  for (int k = 0; k < 3; k++) { bb.lo[k] = std::min(bl.lo[k], br.lo[k]); bb.hi[k] = std::max(bl.hi[k], br.hi[k]); }
  bin[idx] = {l, r, bb};
  return idx;
}

// EMU_SAH_COLLAPSE=1: open the child with the largest surface area until four slots are used
void collapse_sah(Emu& e, const std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t i) {
  uint32_t refs[4] = {bin[i].left, bin[i].right, 0, 0}; int count = 2;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? boxes[order[ref & ~kLeafBit]] : bin[ref].box; };
  while (count < 4) {
    int best = -1; float ba = -1.0f;
    for (int k = 0; k < count; k++) if (!(refs[k] & kLeafBit)) { const float a = half_area(box_of(refs[k])); if (a > ba) { ba = a; best = k; } }
    if (best < 0) break;
    const uint32_t r = refs[best];
    refs[best] = bin[r].left; refs[count++] = bin[r].right;
  }
  Box3 bx[4];
  for (int k = 0; k < count; k++) bx[k] = inflate_box(box_of(refs[k]));
  e.nodes[i] = quantize_node4(bx, refs, count);
  for (int k = 0; k < count; k++) if (!(refs[k] & kLeafBit)) collapse_sah(e, bin, order, boxes, refs[k]);
}

// EMU_WIDE6=1: the product's 6-wide form (BvhNode6, lbvh.hip emit_sah_node6): open the child with the largest surface area until six slots
// are used, internal children first; a node's internal children get consecutive records and its leaf children consecutive triangle slots
// (tri_perm[slot] = position in `order`).  Depth-first here (the GPU numbers level by level): the traversal only needs the two properties.
void collapse6(Emu& e, const std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t i, uint32_t dense,
               std::vector<uint32_t>& tri_perm) {
  uint32_t refs[6] = {bin[i].left, bin[i].right, 0, 0, 0, 0}; int count = 2;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? boxes[order[ref & ~kLeafBit]] : bin[ref].box; };
  while (count < 6) {
    int best = -1; float ba = -1.0f;
    for (int k = 0; k < count; k++) if (!(refs[k] & kLeafBit)) { const float a = half_area(box_of(refs[k])); if (a > ba) { ba = a; best = k; } }
    if (best < 0) break;
    const uint32_t r = refs[best];
    refs[best] = bin[r].left; refs[count++] = bin[r].right;
  }
  uint32_t ints[6], n_int = 0, n_leaf = 0;
  Box3 bi[6], bl[6], bx[6];
  const uint32_t base_leaf = (uint32_t)tri_perm.size();
  for (int k = 0; k < count; k++) {
    if (refs[k] & kLeafBit) { bl[n_leaf++] = inflate_box(box_of(refs[k])); tri_perm.push_back(refs[k] & ~kLeafBit); }
    else { bi[n_int] = inflate_box(box_of(refs[k])); ints[n_int++] = refs[k]; }
  }
  for (uint32_t k = 0; k < n_int; k++) bx[k] = bi[k];
  for (uint32_t k = 0; k < n_leaf; k++) bx[n_int + k] = bl[k];
  const uint32_t base_node = (uint32_t)e.nodes.size();
  e.nodes.resize(e.nodes.size() + n_int);
  const BvhNode6 n6 = quantize_node6(bx, (int)n_int, (int)n_leaf, base_node, base_leaf);
  memcpy(&e.nodes[dense], &n6, sizeof(n6));
  for (uint32_t k = 0; k < n_int; k++) collapse6(e, bin, order, boxes, ints[k], base_node + k, tri_perm);
}

// EMU_SAH_BUILD=1: top-down binned-SAH builder (16 bins, centroid bounds) — a quality yardstick for the GPU's Morton tree
uint32_t build_sah(std::vector<BinNode>& bin, std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t first, uint32_t count) {
  if (count == 1) return kLeafBit | first;
  Box3 bb, cb;
  for (int k = 0; k < 3; k++) { bb.lo[k] = cb.lo[k] = 1e30f; bb.hi[k] = cb.hi[k] = -1e30f; }
  for (uint32_t i = first; i < first + count; i++) {
    const Box3& b = boxes[order[i]];
    for (int k = 0; k < 3; k++) {
      bb.lo[k] = std::min(bb.lo[k], b.lo[k]); bb.hi[k] = std::max(bb.hi[k], b.hi[k]);
      const float c = 0.5f * (b.lo[k] + b.hi[k]);
      cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c);
    }
  }
  constexpr int NB = 16;
  float best_cost = 1e30f; int best_axis = -1, best_split = 0;
  for (int axis = 0; axis < 3; axis++) {
    const float ext = cb.hi[axis] - cb.lo[axis];
    if (!(ext > 0.0f)) continue;
    Box3 bbox[NB]; int bcnt[NB] = {0};
    for (int b = 0; b < NB; b++) for (int k = 0; k < 3; k++) { bbox[b].lo[k] = 1e30f; bbox[b].hi[k] = -1e30f; }
    for (uint32_t i = first; i < first + count; i++) {
      const Box3& b = boxes[order[i]];
      int bi = (int)((0.5f * (b.lo[axis] + b.hi[axis]) - cb.lo[axis]) / ext * NB); bi = std::min(std::max(bi, 0), NB - 1);
      bcnt[bi]++;
      for (int k = 0; k < 3; k++) { bbox[bi].lo[k] = std::min(bbox[bi].lo[k], b.lo[k]); bbox[bi].hi[k] = std::max(bbox[bi].hi[k], b.hi[k]); }
    }
    float right_area[NB]; int right_cnt[NB];
    Box3 acc; for (int k = 0; k < 3; k++) { acc.lo[k] = 1e30f; acc.hi[k] = -1e30f; }
    int c = 0;
    for (int b = NB - 1; b > 0; b--) {
      c += bcnt[b];
      for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], bbox[b].lo[k]); acc.hi[k] = std::max(acc.hi[k], bbox[b].hi[k]); }
      right_area[b] = c ? half_area(acc) : 0.0f; right_cnt[b] = c;
    }
    for (int k = 0; k < 3; k++) { acc.lo[k] = 1e30f; acc.hi[k] = -1e30f; }
    c = 0;
    for (int b = 0; b < NB - 1; b++) {
      c += bcnt[b];
      for (int k = 0; k < 3; k++) { acc.lo[k] = std::min(acc.lo[k], bbox[b].lo[k]); acc.hi[k] = std::max(acc.hi[k], bbox[b].hi[k]); }
      if (c == 0 || right_cnt[b + 1] == 0) continue;
      const float cost = half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = b; }
    }
  }
  uint32_t mid;
  if (best_axis < 0) mid = first + count / 2;
  else {
    const float ext = cb.hi[best_axis] - cb.lo[best_axis];
    auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t a) {
      int bi = (int)((0.5f * (boxes[a].lo[best_axis] + boxes[a].hi[best_axis]) - cb.lo[best_axis]) / ext * NB); bi = std::min(std::max(bi, 0), NB - 1);
      return bi <= best_split; });
    mid = (uint32_t)(it - order.begin());
    if (mid == first || mid == first + count) mid = first + count / 2;
  }
  const uint32_t idx = (uint32_t)bin.size();
  bin.push_back({});
  const uint32_t l = build_sah(bin, order, boxes, first, mid - first);
  const uint32_t r = build_sah(bin, order, boxes, mid, first + count - mid);
  bin[idx] = {l, r, bb};
  return idx;
}

// EMU_PLOC=<radius>: parallel locally-ordered clustering (Meister & Bittner 2018) on the Morton order — prototype of a GPU
// builder that approaches SAH quality: every cluster finds its nearest neighbour (smallest merged surface area) within
// +-radius positions, mutual nearest neighbours merge, the array is compacted, repeat.
uint32_t build_ploc(std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<Box3>& boxes, int radius) {
  struct Cl { uint32_t ref; Box3 box; };
  std::vector<Cl> c(order.size()), next;
  for (size_t i = 0; i < order.size(); i++) c[i] = {kLeafBit | (uint32_t)i, boxes[order[i]]};
  auto merged_area = [](const Box3& a, const Box3& b) {
    Box3 m; for (int k = 0; k < 3; k++) { m.lo[k] = std::min(a.lo[k], b.lo[k]); m.hi[k] = std::max(a.hi[k], b.hi[k]); }
    return half_area(m);
  };
  std::vector<int> nn;
  while (c.size() > 1) {
    const int n = (int)c.size();
    nn.assign(n, -1);
    for (int i = 0; i < n; i++) {
      float best = 1e30f;
      for (int j = std::max(0, i - radius); j <= std::min(n - 1, i + radius); j++) {
        if (j == i) continue;
        const float a = merged_area(c[i].box, c[j].box);
        if (a < best) { best = a; nn[i] = j; }
      }
    }
    next.clear();
    for (int i = 0; i < n; i++) {
      const int j = nn[i];
      if (nn[j] == i) {
        if (i < j) {
          BinNode b; b.left = c[i].ref; b.right = c[j].ref;
          for (int k = 0; k < 3; k++) { b.box.lo[k] = std::min(c[i].box.lo[k], c[j].box.lo[k]); b.box.hi[k] = std::max(c[i].box.hi[k], c[j].box.hi[k]); }
          bin.push_back(b);
          next.push_back({(uint32_t)bin.size() - 1, b.box});
        }
      } else next.push_back(c[i]);
    }
    c.swap(next);
  }
  return c[0].ref;
}

void collapse(Emu& e, const std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t i) {
  uint32_t refs[4]; Box3 bx[4]; int count = 0;
  auto add = [&](uint32_t ref) {
    bx[count] = inflate_box((ref & kLeafBit) ? boxes[order[ref & ~kLeafBit]] : bin[ref].box);
    refs[count++] = ref;
  };
  const uint32_t c[2] = {bin[i].left, bin[i].right};
  for (int k = 0; k < 2; k++) {
    if (c[k] & kLeafBit) add(c[k]);
    else { add(bin[c[k]].left); add(bin[c[k]].right); }
  }
  e.nodes[i] = quantize_node4(bx, refs, count);
  for (int k = 0; k < count; k++) if (!(refs[k] & kLeafBit)) collapse(e, bin, order, boxes, refs[k]);
}

// ---- experiment (EMU_WIDE_PROBE=1): node visits per ray of N-wide trees (N = 4, 6, 8) under different child-ordering rules ------
// The trees are collapsed from the SAME binary tree as the product's 4-wide one (open the internal child with the largest surface
// area until N slots are used), child boxes inflated and quantised to 8 bits against the node's own box exactly like quantize_node4.
// Ordering rules: 0 = all hit children sorted by entry distance (the product's rule), 1 = nearest first, the others in slot order,
// 2 = slots visited in the order slot ^ octant, slots assigned at build time by the child's centroid octant inside the node (what an
// octant-ordered wide BVH does, Ylitie et al. 2017).  Harness-only.
static void quantize_boxes(const Box3* in, int n, Box3* out) {
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  for (int k = 0; k < n; k++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], in[k].lo[a]); hi[a] = fmaxf(hi[a], in[k].hi[a]); }
  for (int a = 0; a < 3; a++) {
    const float need = (hi[a] - lo[a]) * (1.0f / 255.0f);
    uint32_t e = (f2u(need) >> 23) & 0xffu;
    if ((f2u(need) & 0x7fffffu) != 0) e += 1;
    if (e < 1) e = 1; if (e > 254) e = 254;
    while (e < 254 && lo[a] + 255.0f * node_scale((uint8_t)e) < hi[a]) e += 1;
    const float sc = node_scale((uint8_t)e), inv = 1.0f / sc;
    for (int k = 0; k < n; k++) {
      int ql = (int)floorf((in[k].lo[a] - lo[a]) * inv); ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
      while (ql > 0 && lo[a] + (float)ql * sc > in[k].lo[a]) ql--;
      int qh = (int)ceilf((in[k].hi[a] - lo[a]) * inv); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
      while (qh < 255 && lo[a] + (float)qh * sc < in[k].hi[a]) qh++;
      out[k].lo[a] = lo[a] + (float)ql * sc; out[k].hi[a] = lo[a] + (float)qh * sc;
    }
  }
}
static int g_leaf_max = 1;                       // triangles per leaf of the tree being collapsed
static std::vector<uint32_t> g_subtree_leaves;   // per binary node: triangles below it
static void gather_leaves(const std::vector<BinNode>& bin, uint32_t ref, std::vector<uint32_t>& out) {
  if (ref & kLeafBit) { out.push_back(ref & ~kLeafBit); return; }
  gather_leaves(bin, bin[ref].left, out); gather_leaves(bin, bin[ref].right, out);
}
static uint32_t collapse_wide(Emu& e, int store, int N, const std::vector<BinNode>& bin, const std::vector<uint32_t>& order, const std::vector<Box3>& boxes, uint32_t i, bool octant_slots) {
  uint32_t refs[8] = {bin[i].left, bin[i].right}; int count = 2;
  auto box_of = [&](uint32_t ref) { return (ref & kLeafBit) ? boxes[order[ref & ~kLeafBit]] : bin[ref].box; };
  auto closed = [&](uint32_t ref) { return (ref & kLeafBit) || (int)g_subtree_leaves[ref] <= g_leaf_max; };  // stays one child slot
  while (count < N) {
    int best = -1; float ba = -1.0f;
    for (int k = 0; k < count; k++) if (!closed(refs[k])) { const float a = half_area(box_of(refs[k])); if (a > ba) { ba = a; best = k; } }
    if (best < 0) break;
    const uint32_t r = refs[best];
    refs[best] = bin[r].left; refs[count++] = bin[r].right;
  }
  Box3 bx[8], q[8];
  for (int k = 0; k < count; k++) bx[k] = inflate_box(box_of(refs[k]));
  quantize_boxes(bx, count, q);
  const uint32_t me = (uint32_t)e.wide[store].size();
  e.wide[store].push_back({});
  Emu::WideNode w; w.count = N;
  for (int k = 0; k < 8; k++) w.ref[k] = kInvalidRef;
  int slot_of[8];
  if (octant_slots && N == 8) {
    // greedy: children take the free slot closest (Hamming) to the octant of their centroid relative to the node's centre
    float c[3]; Box3 nb = bin[i].box;
    for (int a = 0; a < 3; a++) c[a] = 0.5f * (nb.lo[a] + nb.hi[a]);
    bool used[8] = {false};
    for (int k = 0; k < count; k++) {
      int o = 0;
      for (int a = 0; a < 3; a++) if (0.5f * (bx[k].lo[a] + bx[k].hi[a]) > c[a]) o |= 1 << a;
      int bestslot = -1, bd = 99;
      for (int sl = 0; sl < 8; sl++) if (!used[sl]) { const int d = __builtin_popcount(sl ^ o); if (d < bd) { bd = d; bestslot = sl; } }
      used[bestslot] = true; slot_of[k] = bestslot;
    }
  } else for (int k = 0; k < count; k++) slot_of[k] = k;
  for (int k = 0; k < count; k++) { w.box[slot_of[k]] = q[k]; w.ref[slot_of[k]] = (refs[k] & kLeafBit) ? refs[k] : 0u; }
  for (int k = 0; k < count; k++) {
    if (refs[k] & kLeafBit) continue;
    if (closed(refs[k])) {  // a multi-triangle leaf
      e.groups.push_back({});
      gather_leaves(bin, refs[k], e.groups.back());
      w.ref[slot_of[k]] = kLeafBit | 0x40000000u | (uint32_t)(e.groups.size() - 1);
    } else w.ref[slot_of[k]] = collapse_wide(e, store, N, bin, order, boxes, refs[k], octant_slots);
  }
  e.wide[store][me] = w;
  return me;
}
struct WideCounts { unsigned long long nodes[9][3] = {{0}}, tris[9][3] = {{0}}, rays = 0, mismatch = 0; };
static WideCounts g_wide;
static RayHit wide_closest(const Emu& e, int store, int rule, vec3 o, vec3 d, float tmin, float tmax, unsigned long long* nodes, unsigned long long* tris) {
  const DeviceScene& S = e.S;
  RayHit best; best.t = tmax; best.u = best.v = 0; best.tri = kInvalidRef; best.gid = kInvalidRef;
  vec3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  if (!(fabsf(inv.x) <= 1e30f)) inv.x = copysignf(1e30f, d.x);
  if (!(fabsf(inv.y) <= 1e30f)) inv.y = copysignf(1e30f, d.y);
  if (!(fabsf(inv.z) <= 1e30f)) inv.z = copysignf(1e30f, d.z);
  const int oct = (inv.x < 0 ? 1 : 0) | (inv.y < 0 ? 2 : 0) | (inv.z < 0 ? 4 : 0);
  std::vector<uint32_t> stack;
  uint32_t cur = e.wide_root[store];
  for (;;) {
    const Emu::WideNode& n = e.wide[store][cur];
    (*nodes)++;
    std::pair<float, uint32_t> inner[8]; int ni = 0;
    for (int j = 0; j < 8; j++) {
      const int k = rule == 2 ? (j ^ oct) : j;   // rule 2: a ray going +x visits the low-x slots first
      if (n.ref[k] == kInvalidRef) continue;
      const float tn = slab_entry(n.box[k].lo, n.box[k].hi, o, inv, tmin, best.t * kCullSlack);   // (the shared traverse()'s cull rule: pt_bvh.h)
      if (tn < 0.0f) continue;
      if (n.ref[k] & kLeafBit) {
        static const std::vector<uint32_t> one(1, 0u);
        const bool grp = (n.ref[k] & 0x40000000u) != 0;
        const std::vector<uint32_t>& list = grp ? e.groups[n.ref[k] & 0x3fffffffu] : one;
        for (size_t g = 0; g < list.size(); g++) {
          (*tris)++;
          const uint32_t ti = grp ? list[g] : (n.ref[k] & ~kLeafBit);
          const TriRec& tr = S.tris[ti];
          float t, u, v;
          if (probe_intersect(o, d, tmin, best.t, tr, &t, &u, &v) && (t < best.t || best.tri == kInvalidRef || tr.gid_a < best.gid)) { best.t = t; best.u = u; best.v = v; best.tri = 2 * ti; best.gid = tr.gid_a; }
        }
      } else inner[ni++] = {tn, n.ref[k]};
    }
    if (rule == 0) std::stable_sort(inner, inner + ni, [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
    else if (rule == 1 && ni > 1) { int m = 0; for (int k = 1; k < ni; k++) if (inner[k].first < inner[m].first) m = k; std::swap(inner[0], inner[m]); }
    for (int k = ni - 1; k >= 1; k--) stack.push_back(inner[k].second);
    if (ni > 0) { cur = inner[0].second; continue; }
    if (stack.empty()) return best;
    cur = stack.back(); stack.pop_back();
  }
}

}  // namespace

extern "C" {

void* emu_create(const pt_scene_snapshot* snap, const pt_render_params* p, const void* lut_blob, uint64_t lut_size) {
  auto* e = new Emu();
  e->params = *p;
  g_emu_rcp_ulp = getenv("EMU_RCP_ULP") ? atoi(getenv("EMU_RCP_ULP")) : 0;
  const uint8_t* b = (const uint8_t*)lut_blob;
  uint32_t hdr[32]; memcpy(hdr, b + 12, sizeof(hdr));
  size_t off = 12 + 16 * 8, nf = (lut_size - off) / 4;
  e->lut.resize(nf); memcpy(e->lut.data(), b + off, nf * 4);
  std::string err;
  if (build_host_scene(snap, p, hdr[0], hdr[4], &e->hs, &err) != PT_OK) { fprintf(stderr, "emu: %s\n", err.c_str()); delete e; return nullptr; }
  for (uint32_t c = 2; (int)e->halton.size() < kHaltonDims; c++) {
    bool prime = true;
    for (uint32_t d = 2; d * d <= c; d++) if (c % d == 0) { prime = false; break; }
    if (!prime) continue;
    e->halton.push_back(make_halton_entry(c));
  }
  // flatten to world space (same sequence as lbvh.hip k_flatten): one record per leaf slot = one triangle, or with EMU_PAIRS=1 (what the
  // device build does by default) two consecutive triangles of a mesh that share an edge
  build_primitives(&e->hs, getenv("EMU_PAIRS") != nullptr);
  std::vector<TriRec> tmp; std::vector<Box3> boxes;
  for (uint32_t i = 0; i < e->hs.instances.size(); i++) {
    const InstanceInfo& in = e->hs.instances[i];
    const MeshInfo& m = e->hs.meshes[in.mesh];
    const Xform X = load_xform(in);
    for (uint32_t k = e->hs.mesh_prim_base[in.mesh]; k < e->hs.mesh_prim_base[in.mesh + 1]; k++) {
      const uint32_t t = e->hs.prim_tri[k] & ~kPrimPairBit;
      const bool pair = (e->hs.prim_tri[k] & kPrimPairBit) != 0;
      const uint32_t* idx = &e->hs.indices[3 * (size_t)(m.tri_base + t)];
      const vec3 v0 = transformPoint(ld3(e->hs.positions[m.vertex_base + idx[0]]), X);
      const vec3 v1 = transformPoint(ld3(e->hs.positions[m.vertex_base + idx[1]]), X);
      const vec3 v2 = transformPoint(ld3(e->hs.positions[m.vertex_base + idx[2]]), X);
      vec3 v3_ = v0;
      TriRec r;
      uint32_t bcode = 0;
      r._pad = 0; r.gid_b = kInvalidRef;
      r.gid_a = ((in.tri_global_base + t) << 2) | material_class(e->hs.materials[in.material_base + e->hs.slots[m.tri_base + t]]);
      if (pair) {
        for (int c = 0; c < 3; c++) {
          const uint32_t ib = idx[3 + c];
          uint32_t code = 3u;
          if (ib == idx[0]) code = 0u; else if (ib == idx[1]) code = 1u; else if (ib == idx[2]) code = 2u;
          else v3_ = transformPoint(ld3(e->hs.positions[m.vertex_base + ib]), X);
          bcode |= code << (2 * c);
        }
        r.gid_b = ((in.tri_global_base + t + 1) << 2) | material_class(e->hs.materials[in.material_base + e->hs.slots[m.tri_base + t + 1]]);
      }
      r.inst_code = i | (bcode << kSlotInstBits);
      r.q0[0] = v0.x; r.q0[1] = v0.y; r.q0[2] = v0.z; r.q1[0] = v1.x; r.q1[1] = v1.y; r.q1[2] = v1.z;
      r.q2[0] = v2.x; r.q2[1] = v2.y; r.q2[2] = v2.z; r.q3[0] = v3_.x; r.q3[1] = v3_.y; r.q3[2] = v3_.z;
      Box3 b = tri_box(r, v1, v2);
      const float* p3 = &v3_.x;
      for (int c = 0; c < 3; c++) { b.lo[c] = std::min(b.lo[c], p3[c]); b.hi[c] = std::max(b.hi[c], p3[c]); }
      tmp.push_back(r); boxes.push_back(b);
    }
  }
  std::vector<uint32_t> order(tmp.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
  uint32_t root = kInvalidRef;
  bool wide6 = false;
  std::vector<uint32_t> tri_perm;
  if (!tmp.empty()) {
    std::vector<BinNode> bin;
    if (getenv("EMU_MORTON")) {
      Box3 sb; for (int k = 0; k < 3; k++) { sb.lo[k] = 1e30f; sb.hi[k] = -1e30f; }
      for (auto& b : boxes) for (int k = 0; k < 3; k++) { sb.lo[k] = std::min(sb.lo[k], b.lo[k]); sb.hi[k] = std::max(sb.hi[k], b.hi[k]); }
      const float ext = std::max(sb.hi[0] - sb.lo[0], std::max(sb.hi[1] - sb.lo[1], sb.hi[2] - sb.lo[2]));
      const float scale = ext > 0 ? 2097152.0f / ext : 0.0f;
      std::vector<uint64_t> code(boxes.size());
      for (size_t i = 0; i < boxes.size(); i++) {
        uint64_t c = 0;
        for (int k = 0; k < 3; k++) { float q = (0.5f * (boxes[i].lo[k] + boxes[i].hi[k]) - sb.lo[k]) * scale; q = std::min(std::max(q, 0.0f), 2097151.0f); c |= expand21((uint64_t)q) << (2 - k); }
        code[i] = c;
      }
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
      std::vector<uint64_t> keys(order.size());
      for (size_t i = 0; i < order.size(); i++) keys[i] = code[order[i]];
      if (getenv("EMU_PLOC")) root = build_ploc(bin, order, boxes, atoi(getenv("EMU_PLOC")));
      else root = build_morton(bin, order, keys, boxes, 0, (uint32_t)tmp.size());
    } else if (getenv("EMU_SAH_BUILD")) root = build_sah(bin, order, boxes, 0, (uint32_t)tmp.size());
    else root = build_bin(bin, order, boxes, 0, (uint32_t)tmp.size());
    if (!(root & kLeafBit) && getenv("EMU_WIDE6")) {
      e->nodes.assign(1, BvhNode{});
      e->nodes.reserve(bin.size());
      collapse6(*e, bin, order, boxes, root, 0u, tri_perm);
      root = 0u;
      wide6 = true;
    } else if (!(root & kLeafBit)) {
      e->nodes.assign(bin.size(), BvhNode{});
      if (getenv("EMU_SAH_COLLAPSE")) collapse_sah(*e, bin, order, boxes, root); else collapse(*e, bin, order, boxes, root);
      if (getenv("EMU_WIDE_PROBE")) {
        g_subtree_leaves.assign(bin.size(), 0);
        {
          std::function<uint32_t(uint32_t)> count_leaves = [&](uint32_t ref) -> uint32_t {
            if (ref & kLeafBit) return 1u;
            return g_subtree_leaves[ref] = count_leaves(bin[ref].left) + count_leaves(bin[ref].right);
          };
          count_leaves(root);
        }
        g_leaf_max = 1;
        for (int N : {4, 6, 8}) e->wide_root[N] = collapse_wide(*e, N, N, bin, order, boxes, root, false);
        e->wide_root[0] = collapse_wide(*e, 0, 8, bin, order, boxes, root, true);   // store 0: the 8-wide tree with octant-assigned slots
        // stores 1, 2, 3: 4-wide trees whose leaves hold up to 2, 3, 4 triangles (every binary subtree that small is one leaf)
        for (int M = 2; M <= 4; M++) { g_leaf_max = M; e->wide_root[M - 1] = collapse_wide(*e, M - 1, 4, bin, order, boxes, root, false); }
        g_leaf_max = 1;
      }
    }
  }
  e->tris.resize(tmp.size());
  for (size_t i = 0; i < order.size(); i++) e->tris[i] = tmp[order[wide6 ? tri_perm[i] : i]];

  DeviceScene& S = e->S;
  memset(&S, 0, sizeof(S));
  S.positions = e->hs.positions.data(); S.vdata = e->hs.vdata.data(); S.indices = e->hs.indices.data(); S.slots = e->hs.slots.data();
  S.meshes = e->hs.meshes.data(); S.instances = e->hs.instances.data(); S.materials = e->hs.materials.data();
  S.lights = e->hs.lights.data(); S.nodes = e->nodes.data(); S.tris = e->tris.data(); S.tri_count = e->hs.tri_count; S.slot_count = (uint32_t)e->tris.size();
  S.root_ref = root; S.wide6 = wide6 ? 1u : 0u; S.halton = e->halton.data();
  e->shade_recs.assign(2 * e->tris.size(), ShadeRec{});   // entry 2 * slot + half
  for (size_t i = 0; i < 2 * e->tris.size(); i++) {
    const TriRec& tr = e->tris[i >> 1];
    const uint32_t gid = (i & 1) ? tr.gid_b : tr.gid_a;
    if (gid != kInvalidRef) e->shade_recs[i] = make_shade_rec(S, tr.inst_code & kSlotInstMask, (gid >> 2) - e->hs.instances[tr.inst_code & kSlotInstMask].tri_global_base);
  }
  S.shade_recs = e->shade_recs.data();
  e->light_recs.resize(e->hs.lights.size());
  for (size_t i = 0; i < e->hs.lights.size(); i++) e->light_recs[i] = make_light_rec(S, e->hs.lights[i]);
  S.light_recs = e->light_recs.data();
  for (auto& l : e->hs.lights) e->light_cdf.push_back(l.cumulativePower);
  S.light_cdf = e->light_cdf.data();
  Lut* ls[6] = {&S.luts.E, &S.luts.Eavg, &S.luts.EMs, &S.luts.EavgMs, &S.luts.ETransIn, &S.luts.ETransOut};
  for (int i = 0; i < 6; i++) { ls[i]->w = hdr[4 * i]; ls[i]->h = hdr[4 * i + 1]; ls[i]->depth = hdr[4 * i + 2]; ls[i]->d = e->lut.data() + hdr[4 * i + 3]; }
  S.camera = e->hs.constants.camera; S.idt = e->hs.idt; S.width = p->width; S.height = p->height;
  S.lightCount = e->hs.constants.lightCount; S.totalLightPower = e->hs.constants.totalLightPower;
  S.flags = p->flags; S.integrator = p->integrator; S.max_bounces = p->max_bounces;
  S.tex_data = e->hs.tex_data.data(); S.tex_decode = e->hs.tex_decode.data(); S.textures = e->hs.textures.data(); S.tex_native = e->hs.tex_native; S.env_alias = e->hs.env_alias.data();
  S.env_texture = e->hs.env_texture; S.envLightCount = e->hs.constants.envLightCount; S.has_alpha = e->hs.has_alpha ? 1u : 0u;
  return e;
}
void emu_destroy(void* h) { delete (Emu*)h; }
void emu_get_constants(void* h, pt_constants* out) { *out = ((Emu*)h)->hs.constants; }
uint32_t emu_get_lights(void* h, pt_area_light* out, uint32_t cap) {
  Emu* e = (Emu*)h;
  for (uint32_t i = 0; i < std::min<uint32_t>(cap, (uint32_t)e->hs.lights.size()); i++) out[i] = e->hs.lights[i];
  return (uint32_t)e->hs.lights.size();
}

// ---- experiment (EMU_CULL_PROBE=1): how many node visits would a stack that also carries each entry's ENTRY DISTANCE save? --------
// The product's stack holds child refs only: an entry pushed before a closer hit was found is still fetched and slab-tested when it
// comes up.  With the entry distance beside the ref, a pop can drop it (tn > best.t) without touching the node.  The probe walks the
// same tree with the same child arithmetic as trav_visit (scalar formulation: a node, then its leaf children at once) twice per ray,
// with and without that test, and counts node fetches.  Harness-only; nothing here is compiled into libptamd.so.
struct ProbeCounts { unsigned long long nodes = 0, nodes_cull = 0, tris = 0, tris_cull = 0, rays = 0, mismatch = 0; };
static ProbeCounts g_probe;
static unsigned long long g_stack_hist[64] = {0};  // pushes by the stack depth they write to (non-culling probe)
static RayHit probe_closest(const DeviceScene& S, vec3 o, vec3 d, float tmin, float tmax, bool cull, unsigned long long* nodes, unsigned long long* tris) {
  RayHit best; best.t = tmax; best.u = best.v = 0; best.tri = kInvalidRef; best.gid = kInvalidRef;
  if (S.root_ref == kInvalidRef) return best;
  vec3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  if (!(fabsf(inv.x) <= 1e30f)) inv.x = copysignf(1e30f, d.x);
  if (!(fabsf(inv.y) <= 1e30f)) inv.y = copysignf(1e30f, d.y);
  if (!(fabsf(inv.z) <= 1e30f)) inv.z = copysignf(1e30f, d.z);
  const bool gx = inv.x < 0, gy = inv.y < 0, gz = inv.z < 0;
  auto test_tri = [&](uint32_t ti) {
    (*tris)++;
    const TriRec& tr = S.tris[ti];
    float t, u, v;
    if (!probe_intersect(o, d, tmin, best.t, tr, &t, &u, &v)) return;
    if (t < best.t || best.tri == kInvalidRef || tr.gid_a < best.gid) { best.t = t; best.u = u; best.v = v; best.tri = 2 * ti; best.gid = tr.gid_a; }
  };
  if (S.root_ref & kLeafBit) { test_tri(S.root_ref & ~kLeafBit); return best; }
  std::vector<std::pair<uint32_t, float>> stack;
  uint32_t cur = S.root_ref;
  for (;;) {
    const BvhNode n = S.nodes[cur];
    (*nodes)++;
    const float ax = node_scale(n.exp[0]) * inv.x, ay = node_scale(n.exp[1]) * inv.y, az = node_scale(n.exp[2]) * inv.z;
    const float bx = (n.origin[0] - o.x) * inv.x, by = (n.origin[1] - o.y) * inv.y, bz = (n.origin[2] - o.z) * inv.z;
    const uint32_t nx = gx ? n.qhi[0] : n.qlo[0], fx = gx ? n.qlo[0] : n.qhi[0];
    const uint32_t ny = gy ? n.qhi[1] : n.qlo[1], fy = gy ? n.qlo[1] : n.qhi[1];
    const uint32_t nz = gz ? n.qhi[2] : n.qlo[2], fz = gz ? n.qlo[2] : n.qhi[2];
    std::pair<float, uint32_t> inner[4];
    int ni = 0;
    uint32_t leaves[4]; int nl = 0;
    for (int k = 0; k < 4; k++) {
      const float tnx = __builtin_fmaf((float)((nx >> (8 * k)) & 0xffu), ax, bx), tfx = __builtin_fmaf((float)((fx >> (8 * k)) & 0xffu), ax, bx);
      const float tny = __builtin_fmaf((float)((ny >> (8 * k)) & 0xffu), ay, by), tfy = __builtin_fmaf((float)((fy >> (8 * k)) & 0xffu), ay, by);
      const float tnz = __builtin_fmaf((float)((nz >> (8 * k)) & 0xffu), az, bz), tfz = __builtin_fmaf((float)((fz >> (8 * k)) & 0xffu), az, bz);
      const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), tmin);
      const float tf = fminf(fminf(fminf(tfx, tfy), tfz), best.t * kCullSlack);
      const bool hit = n.ref[k] != kInvalidRef && tn <= __builtin_fmaf(tf, 1.0000005f, 1e-30f);
      if (!hit) continue;
      if (n.ref[k] & kLeafBit) leaves[nl++] = n.ref[k] & ~kLeafBit; else inner[ni++] = {tn, n.ref[k]};
    }
    for (int k = 0; k < nl; k++) test_tri(leaves[k]);
    std::stable_sort(inner, inner + ni, [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
    for (int k = ni - 1; k >= 1; k--) { if (!cull) g_stack_hist[stack.size() < 63 ? stack.size() : 63]++; stack.push_back({inner[k].second, inner[k].first}); }
    if (ni > 0) { cur = inner[0].second; continue; }
    bool got = false;
    while (!stack.empty()) {
      const auto e = stack.back(); stack.pop_back();
      // (the same conservative comparison the slab test uses: an entry is only dropped when its children could not pass it)
      if (cull && !(e.second <= __builtin_fmaf(best.t * kCullSlack, 1.0000005f, 1e-30f))) continue;
      cur = e.first; got = true; break;
    }
    if (!got) return best;
  }
}

// One sample of every pixel, following the kernel sequence of kernels.hip per path.
void emu_debug_sample(void* h, uint32_t sample, float* radiance /*W*H*4*/, int32_t* hits /*B*W*H*2 or null*/) {
  Emu* e = (Emu*)h;
  const DeviceScene& S = e->S;
  const uint32_t W = S.width, H = S.height, B = S.max_bounces, NP = W * H;
  if (hits) for (size_t i = 0; i < (size_t)B * NP * 2; i++) hits[i] = -1;
  std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      const uint32_t pid = y * W + x;
      RayGenOut rg = stage_raygen(S, x, y, sample);
      vec3 o = rg.o, d = rg.d, att = v3(1.0f), L = v3(0.0f);
      float lastPdf = 0.0f; bool lastSpec = false; uint32_t dim = rg.dim;
      for (uint32_t b = 0; b < B; b++) {
        TraversalStack st; st.lds = lds.data(); st.pend = pend.data(); st.lds_stride = 1; st.spill = spill.data(); st.spill_stride = 1;
        TraversalCount tc;
        const float ir = S.has_alpha ? Halton{halton_table(S.halton), rg.offset, dim}.sample1d() : 0.0f;
        RayHit hit = traverse<false, true>(S, o, d, 1e-3f, kInf, ir, st, &tc);
        g_nodes += tc.nodes; g_tris += tc.tris; g_leaves += tc.leaves; g_rays++;
        if (b < 16) { g_pb_nodes[b] += tc.nodes; g_pb_tris[b] += tc.tris; g_pb_leaves[b] += tc.leaves; g_pb_rays[b]++; }
        if (!e->wide[8].empty() && !S.has_alpha) {
          for (int N : {4, 6, 8})
            for (int rule = 0; rule < 3; rule++) {
              if (rule == 2 && N != 8) continue;
              const RayHit w = wide_closest(*e, rule == 2 ? 0 : N, rule, o, d, 1e-3f, kInf, &g_wide.nodes[N][rule], &g_wide.tris[N][rule]);
              if (w.tri != hit.tri || w.t != hit.t) g_wide.mismatch++;
            }
          for (int M = 2; M <= 4; M++) {  // 4-wide, leaves of up to M triangles (counts kept in the unused slots [M - 1][0])
            const RayHit w = wide_closest(*e, M - 1, 0, o, d, 1e-3f, kInf, &g_wide.nodes[M - 1][0], &g_wide.tris[M - 1][0]);
            if (w.tri != hit.tri || w.t != hit.t) g_wide.mismatch++;
          }
          g_wide.rays++;
        }
        if (getenv("EMU_CULL_PROBE") && !S.has_alpha) {
          const RayHit a = probe_closest(S, o, d, 1e-3f, kInf, false, &g_probe.nodes, &g_probe.tris);
          const RayHit c = probe_closest(S, o, d, 1e-3f, kInf, true, &g_probe.nodes_cull, &g_probe.tris_cull);
          g_probe.rays++;
          if (a.tri != hit.tri || c.tri != hit.tri || a.t != hit.t || c.t != hit.t) g_probe.mismatch++;
        }
        if (hits && hit.tri != kInvalidRef) triangle_ids(S, hit.tri, &hits[((size_t)b * NP + pid) * 2], &hits[((size_t)b * NP + pid) * 2 + 1]);
        if (hit.tri == kInvalidRef) {
          if (S.env_texture >= 0) L = L + stage_miss(S, d, att, b, lastPdf, lastSpec);
          L = L + att * 0.0f;  // attenuation * backgroundColor (kernel.metal:311 / :541): NaN for a throughput that is not finite
          break;
        }
        const vec4 qO{o.x, o.y, o.z, lastPdf}, qD{d.x, d.y, d.z, 0.0f};  // the queue entry the stage may re-read
        ShadeIn in; in.o = o; in.d = d; in.att = att; in.rayO = &qO; in.rayD = &qD; in.lastSpecular = lastSpec; in.offset = rg.offset;
        in.dim = dim + 1; in.bounce = b; in.t = hit.t; in.u = hit.u; in.v = hit.v; in.tri = hit.tri;
        ShadeOut out = stage_shade(S, in);
        if (out.has_emitted) L = L + out.emitted;
        if (out.shadow) {
          RayHit sh = traverse<true, false>(S, out.shadow_o, out.shadow_d, 1e-3f, out.shadow_tmax, out.shadow_payload, st, &tc);
          if (sh.tri == kInvalidRef) L = L + out.shadow_contrib;
        }
        if (!out.alive) break;
        o = out.next_o; d = out.next_d; att = out.next_att; lastPdf = out.next_pdf; lastSpec = out.next_specular;
        dim = out.dim & kMetaDimMask;
      }
      radiance[4 * pid] = L.x; radiance[4 * pid + 1] = L.y; radiance[4 * pid + 2] = L.z; radiance[4 * pid + 3] = 1.0f;
    }
}

// ---- the CPU leg of the benchmark: "the same __host__ __device__ kernels built for the host, all host cores" ----
// Renders samples [first, first + ns) of every pixel with the product's own stage functions (stage_raygen, traverse<> over the product's
// 6-wide node form, stage_shade, stage_miss: platinum_amd/csrc/pt_*.h compiled by g++ with -ffp-contract=off), std::thread workers taking
// 16x16 pixel tiles from one atomic cursor, and folds them into `acc` as k_accumulate does (running mean in sample order, n0 samples
// already there).  Per path it is the kernel sequence of kernels.hip; what the host has no use for (queues, compaction, chunk claims) is
// the scheduler, not the arithmetic.  bench.py times this call only (the scene / BVH set-up is emu_create).
}  // extern "C"  (std::thread / std::atomic below)
#include <atomic>
#include <thread>
extern "C" {
static vec3 emu_path(const Emu* e, uint32_t x, uint32_t y, uint32_t sample, uint32_t* lds, uint32_t* spill, uint32_t* pend) {
  const DeviceScene& S = e->S;
  const uint32_t B = S.max_bounces;
  RayGenOut rg = stage_raygen(S, x, y, sample);
  vec3 o = rg.o, d = rg.d, att = v3(1.0f), L = v3(0.0f);
  float lastPdf = 0.0f; bool lastSpec = false; uint32_t dim = rg.dim;
  for (uint32_t b = 0; b < B; b++) {
    TraversalStack st; st.lds = lds; st.pend = pend; st.lds_stride = 1; st.spill = spill; st.spill_stride = 1;
    TraversalCount tc;
    const float ir = S.has_alpha ? Halton{halton_table(S.halton), rg.offset, dim}.sample1d() : 0.0f;
    const RayHit hit = traverse<false, false>(S, o, d, 1e-3f, kInf, ir, st, &tc);
    if (hit.tri == kInvalidRef) {
      if (S.env_texture >= 0) L = L + stage_miss(S, d, att, b, lastPdf, lastSpec);
      L = L + att * 0.0f;
      break;
    }
    const vec4 qO{o.x, o.y, o.z, lastPdf}, qD{d.x, d.y, d.z, 0.0f};
    ShadeIn in; in.o = o; in.d = d; in.att = att; in.rayO = &qO; in.rayD = &qD; in.lastSpecular = lastSpec; in.offset = rg.offset;
    in.dim = dim + 1; in.bounce = b; in.t = hit.t; in.u = hit.u; in.v = hit.v; in.tri = hit.tri;
    const ShadeOut out = stage_shade(S, in);
    if (out.has_emitted) L = L + out.emitted;
    if (out.shadow) {
      const RayHit sh = traverse<true, false>(S, out.shadow_o, out.shadow_d, 1e-3f, out.shadow_tmax, out.shadow_payload, st, &tc);
      if (sh.tri == kInvalidRef) L = L + out.shadow_contrib;
    }
    if (!out.alive) break;
    o = out.next_o; d = out.next_d; att = out.next_att; lastPdf = out.next_pdf; lastSpec = out.next_specular;
    dim = out.dim & kMetaDimMask;
  }
  return L;
}

void emu_render(void* h, uint32_t first, uint32_t ns, float* acc /*W*H*4, running mean*/, uint32_t n0, uint32_t threads) {
  const Emu* e = (const Emu*)h;
  const uint32_t W = e->S.width, H = e->S.height, tx = (W + 15) / 16, ty = (H + 15) / 16;
  std::atomic<uint32_t> cursor{0};
  auto worker = [&]() {
    std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
    for (;;) {
      const uint32_t t = cursor.fetch_add(1);
      if (t >= tx * ty) return;
      const uint32_t x0 = (t % tx) * 16, y0 = (t / tx) * 16;
      for (uint32_t y = y0; y < std::min(y0 + 16, H); y++)
        for (uint32_t x = x0; x < std::min(x0 + 16, W); x++) {
          float* a = acc + 4 * ((size_t)y * W + x);
          for (uint32_t s = 0; s < ns; s++) {
            vec3 L = emu_path(e, x, y, first + s, lds.data(), spill.data(), pend.data());
            const uint32_t n = n0 + s;                 // k_accumulate (kernel.metal:672-684)
            if (n > 0) { L = L + v3(a[0], a[1], a[2]) * (float)n; L = L / (float)(n + 1); }
            a[0] = L.x; a[1] = L.y; a[2] = L.z; a[3] = 1.0f;
          }
        }
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t i = 1; i < std::max(1u, threads); i++) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
}

void emu_trace_primary(void* h, uint32_t sample, pt_hit_record* out) {
  Emu* e = (Emu*)h;
  const DeviceScene& S = e->S;
  std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
  for (uint32_t y = 0; y < S.height; y++)
    for (uint32_t x = 0; x < S.width; x++) {
      RayGenOut rg = stage_raygen(S, x, y, sample);
      TraversalStack st; st.lds = lds.data(); st.pend = pend.data(); st.lds_stride = 1; st.spill = spill.data(); st.spill_stride = 1;
      TraversalCount tc;
      const float ir = S.has_alpha ? Halton{halton_table(S.halton), rg.offset, rg.dim}.sample1d() : 0.0f;
      RayHit hit = traverse<false, false>(S, rg.o, rg.d, 1e-3f, kInf, ir, st, &tc);
      pt_hit_record& r = out[y * S.width + x];
      if (hit.tri != kInvalidRef) { r.t = hit.t; r.u = hit.u; r.v = hit.v; triangle_ids(S, hit.tri, &r.instance, &r.primitive); }
      else { r.t = r.u = r.v = 0; r.instance = r.primitive = -1; }
    }
}
// ---- experiment (emu_packet_probe): camera rays of one 8x8 pixel tile traced as ONE packet (a wave that walks the tree once for its 64
// rays: a node is visited when any ray's slab test passes with that ray's current best t; children ordered by the smallest entry
// distance among the rays that hit them; a leaf's triangle is tested by every ray whose slab test passed).  Counts node visits and
// triangle rounds per PACKET against the per-ray traversal's totals over the same 64 rays.  out: {packets, packet node visits, packet
// triangle rounds, per-ray node visits (sum), per-ray triangle tests (sum), mismatching hits}
void emu_packet_probe(void* h, uint32_t sample, double out[6]) {
  Emu* e = (Emu*)h;
  const DeviceScene& S = e->S;
  std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
  double packets = 0, pn = 0, pt_ = 0, rn = 0, rt = 0, mism = 0;
  for (uint32_t ty = 0; ty < (S.height + 7) / 8; ty++)
    for (uint32_t tx = 0; tx < (S.width + 7) / 8; tx++) {
      struct R { vec3 o, d, inv; RayHit best; bool on; } r[64];
      int n = 0;
      for (uint32_t l = 0; l < 64; l++) {
        const uint32_t x = tx * 8 + (l & 7), y = ty * 8 + (l >> 3);
        if (x >= S.width || y >= S.height) continue;
        const RayGenOut rg = stage_raygen(S, x, y, sample);
        R& q = r[n++];
        q.o = rg.o; q.d = rg.d; q.inv = v3(1.0f / rg.d.x, 1.0f / rg.d.y, 1.0f / rg.d.z); q.on = true;
        q.best.t = kInf; q.best.u = q.best.v = 0; q.best.tri = kInvalidRef; q.best.gid = kInvalidRef;
        TraversalStack st; st.lds = lds.data(); st.pend = pend.data(); st.lds_stride = 1; st.spill = spill.data(); st.spill_stride = 1;
        TraversalCount tc;
        const RayHit ref = traverse<false, true>(S, rg.o, rg.d, 1e-3f, kInf, 0.0f, st, &tc);
        rn += tc.nodes; rt += tc.tris;
        q.best.gid = ref.tri;  // (stash the per-ray answer for the comparison below)
      }
      if (n == 0 || S.root_ref == kInvalidRef || (S.root_ref & kLeafBit)) continue;
      uint32_t want[64];
      for (int k = 0; k < n; k++) { want[k] = r[k].best.gid; r[k].best.gid = kInvalidRef; }
      packets++;
      std::vector<uint32_t> stack;
      uint32_t cur = S.root_ref;
      for (;;) {
        const BvhNode nd = S.nodes[cur];
        pn++;
        float tmin_child[4] = {kInf, kInf, kInf, kInf};
        bool any[4] = {false, false, false, false};
        bool hitk[64][4];
        for (int k = 0; k < n; k++) {
          const R& q = r[k];
          for (int c = 0; c < 4; c++) {
            hitk[k][c] = false;
            if (nd.ref[c] == kInvalidRef) continue;
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
              lo[a] = nd.origin[a] + (float)((nd.qlo[a] >> (8 * c)) & 0xffu) * node_scale(nd.exp[a]);
              hi[a] = nd.origin[a] + (float)((nd.qhi[a] >> (8 * c)) & 0xffu) * node_scale(nd.exp[a]);
            }
            const float tn = slab_entry(lo, hi, q.o, q.inv, 1e-3f, q.best.t * kCullSlack);
            if (tn < 0.0f) continue;
            hitk[k][c] = true; any[c] = true; tmin_child[c] = fminf(tmin_child[c], tn);
          }
        }
        // leaves: one triangle round per hit leaf child (all rays whose slab test passed take part)
        for (int c = 0; c < 4; c++) {
          if (!any[c] || !(nd.ref[c] & kLeafBit)) continue;
          pt_++;
          const uint32_t ti = nd.ref[c] & ~kLeafBit;
          const TriRec& tr = S.tris[ti];
          for (int k = 0; k < n; k++) {
            if (!hitk[k][c]) continue;
            R& q = r[k];
            float t, u, v;
            if (probe_intersect(q.o, q.d, 1e-3f, q.best.t, tr, &t, &u, &v) && (t < q.best.t || q.best.tri == kInvalidRef || tr.gid_a < q.best.gid)) {
              q.best.t = t; q.best.u = u; q.best.v = v; q.best.tri = 2 * ti; q.best.gid = tr.gid_a;
            }
          }
        }
        std::pair<float, uint32_t> inner[4]; int ni = 0;
        for (int c = 0; c < 4; c++) if (any[c] && !(nd.ref[c] & kLeafBit)) inner[ni++] = {tmin_child[c], nd.ref[c]};
        std::stable_sort(inner, inner + ni, [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
        for (int k = ni - 1; k >= 1; k--) stack.push_back(inner[k].second);
        if (ni > 0) { cur = inner[0].second; continue; }
        if (stack.empty()) break;
        cur = stack.back(); stack.pop_back();
      }
      for (int k = 0; k < n; k++) if (r[k].best.tri != want[k]) mism++;
    }
  out[0] = packets; out[1] = pn; out[2] = pt_; out[3] = rn; out[4] = rt; out[5] = mism;
}
// ---- experiment (emu_origin_sort_probe, r5): would sorting a segment's SECONDARY rays by the cell of their origin make its 64-ray chunks touch
// fewer distinct lines?  The rays entering bounce `b` of ONE 8x8 tile under `ns` samples (a segment of the wavefront) are cut into 64-ray chunks
// (a) in the order the stage leaves them (pixel-major, sample-minor, survivors compacted) and (b) sorted by the Morton code of their origin in a
// 32^3 grid over the origins' bounds; per chunk: the number of DISTINCT 64-byte lines (6-wide nodes + leaf slots) its rays fetch, against the
// sum over its rays.  out: {rays, chunks, lines summed over rays, distinct per chunk summed (arrival order), the same (sorted)}.  6-wide trees only.
static void trace_lines(const DeviceScene& S, vec3 o, vec3 d, uint32_t* lds, uint32_t* spill, uint32_t* pend, std::vector<uint32_t>* lines) {
  TraversalStack st; st.lds = lds; st.pend = pend; st.lds_stride = 1; st.spill = spill; st.spill_stride = 1;
  TravState ts;
  if (trav_init(S, ts, o, d, 1e-3f, kInf, 0.0f, st, false, nullptr)) return;
  while (!(ts.cur == kInvalidRef && ts.st.npend == 0)) {
    if (ts.cur != kInvalidRef && ts.st.npend <= kPendLeaves6 - 1) { lines->push_back(ts.cur); trav_node6<false>(S.nodes, ts, nullptr); }
    while (ts.st.npend > 0) {
      const uint32_t e = ts.st.pend[(ts.st.npend - 1) * ts.st.lds_stride];
      lines->push_back(0x80000000u | ((e >> 6) + (uint32_t)__builtin_ctz(e & 63u)));
      trav_pending_leaf6<false, false>(S, ts, nullptr);
    }
  }
}
void emu_origin_sort_probe(void* h, uint32_t tile_x, uint32_t tile_y, uint32_t ns, uint32_t bounce, double out[5]) {
  Emu* e = (Emu*)h;
  const DeviceScene& S = e->S;
  for (int i = 0; i < 5; i++) out[i] = 0;
  if (!S.wide6) return;
  std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
  struct R { vec3 o, d; };
  std::vector<R> rays;
  for (uint32_t pl = 0; pl < 64; pl++)
    for (uint32_t smp = 0; smp < ns; smp++) {
      const uint32_t x = tile_x * 8 + (pl & 7), y = tile_y * 8 + (pl >> 3);
      if (x >= S.width || y >= S.height) continue;
      RayGenOut rg = stage_raygen(S, x, y, smp);
      vec3 o = rg.o, d = rg.d, att = v3(1.0f);
      float lastPdf = 0.0f; bool lastSpec = false; uint32_t dim = rg.dim;
      bool alive = true;
      for (uint32_t b = 0; b < bounce && alive; b++) {
        TraversalStack st; st.lds = lds.data(); st.pend = pend.data(); st.lds_stride = 1; st.spill = spill.data(); st.spill_stride = 1;
        const RayHit hit = traverse<false, false>(S, o, d, 1e-3f, kInf, 0.0f, st, nullptr);
        if (hit.tri == kInvalidRef) { alive = false; break; }
        const vec4 qO{o.x, o.y, o.z, lastPdf}, qD{d.x, d.y, d.z, 0.0f};
        ShadeIn in; in.o = o; in.d = d; in.att = att; in.rayO = &qO; in.rayD = &qD; in.lastSpecular = lastSpec; in.offset = rg.offset;
        in.dim = dim + 1; in.bounce = b; in.t = hit.t; in.u = hit.u; in.v = hit.v; in.tri = hit.tri;
        const ShadeOut so = stage_shade(S, in);
        if (!so.alive) { alive = false; break; }
        o = so.next_o; d = so.next_d; att = so.next_att; lastPdf = so.next_pdf; lastSpec = so.next_specular; dim = so.dim & kMetaDimMask;
      }
      if (alive) rays.push_back({o, d});
    }
  if (rays.empty()) return;
  std::vector<std::vector<uint32_t>> lines(rays.size());
  double total = 0;
  for (size_t i = 0; i < rays.size(); i++) { trace_lines(S, rays[i].o, rays[i].d, lds.data(), spill.data(), pend.data(), &lines[i]); total += (double)lines[i].size(); }
  auto distinct = [&](const std::vector<uint32_t>& order) {
    double sum = 0;
    for (size_t c = 0; c < order.size(); c += 64) {
      std::vector<uint32_t> u;
      for (size_t k = c; k < std::min(order.size(), c + 64); k++) u.insert(u.end(), lines[order[k]].begin(), lines[order[k]].end());
      std::sort(u.begin(), u.end());
      sum += (double)(std::unique(u.begin(), u.end()) - u.begin());
    }
    return sum;
  };
  std::vector<uint32_t> order(rays.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
  const double arrival = distinct(order);
  vec3 lo = rays[0].o, hi = rays[0].o;
  for (auto& r : rays) { lo = v3(fminf(lo.x, r.o.x), fminf(lo.y, r.o.y), fminf(lo.z, r.o.z)); hi = v3(fmaxf(hi.x, r.o.x), fmaxf(hi.y, r.o.y), fmaxf(hi.z, r.o.z)); }
  const float ext = fmaxf(fmaxf(hi.x - lo.x, hi.y - lo.y), fmaxf(hi.z - lo.z, 1e-20f));
  std::vector<uint64_t> key(rays.size());
  for (size_t i = 0; i < rays.size(); i++) {
    const uint64_t cx = (uint64_t)fminf(31.0f, (rays[i].o.x - lo.x) / ext * 32.0f), cy = (uint64_t)fminf(31.0f, (rays[i].o.y - lo.y) / ext * 32.0f),
                   cz = (uint64_t)fminf(31.0f, (rays[i].o.z - lo.z) / ext * 32.0f);
    key[i] = expand21(cx) << 2 | expand21(cy) << 1 | expand21(cz);
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  const double sorted = distinct(order);
  out[0] = (double)rays.size(); out[1] = (double)((rays.size() + 63) / 64); out[2] = total; out[3] = arrival; out[4] = sorted;
}
// ---- experiment (emu_l2_probe, r6): would SCENE-SPACE ray queues keep the closest-hit kernel's BVH lines inside the XCDs' L2s?
// The rays entering bounce `bounce` of a window of the image (tiles [s0, s0 + T) of each of the 4 segment bands, `ns` samples: what the chunk
// tables list consecutively today) are traced on the product's 6-wide tree with every 64-byte line (node or leaf slot) they fetch recorded, then
// REPLAYED through a model of the chip's eight L2s — 4 MB each, 64-byte lines (the counters' miss = one 64-byte fetch: 116 B / 1.93 per ray,
// profiles/r05_pmc_c3.json), 16-way LRU — fed by 768 resident waves per XCD (32 CUs x 4 SIMDs x 6 waves), every wave stepping its 64 rays one
// line per tick, all waves of the chip round-robin, chunks claimed 8 at a time:
//   mode 0  today: ONE cursor over the chunk list in segment order (whichever wave asks next gets the run: XCDs interleave)
//   mode 1  rays sorted by the Morton cell of their ORIGIN (10 bits per axis over the origins' bounds), cut into 8 contiguous ranges, one per
//           XCD, each XCD's waves claiming from their own cursor — the scheduler DESIGN section 7 names
//   mode 2  the sorted list behind ONE cursor (what the sort alone buys, without the partition)
// At bounce 1 the window's ray density per cell equals the full frame's (origins = the window's own primary hits); at bounce >= 2 origins
// scatter and the window's density is LOWER than the full frame's: modes 1 / 2 are then pessimistic (emu_l2_probe_dense covers that case).
// out[mode][6] = {rays counted (chunks claimed after the first fifth of the list), BVH line touches, BVH misses, queue lines streamed,
//                wave-steps, wave-steps in which at least one lane missed (a wave-step waits for its slowest lane)}
struct ProbeRay { float o[3]; uint32_t first, count; };
struct L2Sim {
  static constexpr uint32_t kWays = 16;
  uint32_t sets;
  std::vector<uint32_t> tag, stamp;
  uint32_t clock = 1;
  explicit L2Sim(uint32_t bytes) : sets(bytes / 64u / kWays), tag((size_t)sets * kWays, 0xffffffffu), stamp((size_t)sets * kWays, 0u) {}
  bool touch(uint32_t line) {   // true = hit
    const uint32_t s = (line * 2654435761u >> 7) % sets;   // (the hardware hashes addresses over channels / sets; a plain modulo would alias the two arrays)
    uint32_t* t = &tag[(size_t)s * kWays]; uint32_t* st = &stamp[(size_t)s * kWays];
    uint32_t victim = 0;
    for (uint32_t w = 0; w < kWays; w++) {
      if (t[w] == line) { st[w] = clock++; return true; }
      if (st[w] < st[victim]) victim = w;
    }
    t[victim] = line; st[victim] = clock++;
    return false;
  }
};
static void gen_tile_rays(const Emu* e, uint32_t tile_x, uint32_t tile_y, uint32_t ns, uint32_t bounce, std::vector<ProbeRay>* rays, std::vector<uint32_t>* lines,
                          uint32_t* lds, uint32_t* spill, uint32_t* pend) {
  const DeviceScene& S = e->S;
  std::vector<uint32_t> tmp;
  for (uint32_t pl = 0; pl < 64; pl++)
    for (uint32_t smp = 0; smp < ns; smp++) {
      const uint32_t x = tile_x * 8 + (pl & 7), y = tile_y * 8 + (pl >> 3);
      if (x >= S.width || y >= S.height) continue;
      RayGenOut rg = stage_raygen(S, x, y, smp);
      vec3 o = rg.o, d = rg.d, att = v3(1.0f);
      float lastPdf = 0.0f; bool lastSpec = false; uint32_t dim = rg.dim;
      bool alive = true;
      for (uint32_t b = 0; b < bounce && alive; b++) {
        TraversalStack st; st.lds = lds; st.pend = pend; st.lds_stride = 1; st.spill = spill; st.spill_stride = 1;
        const RayHit hit = traverse<false, false>(S, o, d, 1e-3f, kInf, 0.0f, st, nullptr);
        if (hit.tri == kInvalidRef) { alive = false; break; }
        const vec4 qO{o.x, o.y, o.z, lastPdf}, qD{d.x, d.y, d.z, 0.0f};
        ShadeIn in; in.o = o; in.d = d; in.att = att; in.rayO = &qO; in.rayD = &qD; in.lastSpecular = lastSpec; in.offset = rg.offset;
        in.dim = dim + 1; in.bounce = b; in.t = hit.t; in.u = hit.u; in.v = hit.v; in.tri = hit.tri;
        const ShadeOut so = stage_shade(S, in);
        if (!so.alive) { alive = false; break; }
        o = so.next_o; d = so.next_d; att = so.next_att; lastPdf = so.next_pdf; lastSpec = so.next_specular; dim = so.dim & kMetaDimMask;
      }
      if (!alive) continue;
      tmp.clear();
      trace_lines(S, o, d, lds, spill, pend, &tmp);
      ProbeRay r; r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.first = (uint32_t)lines->size(); r.count = (uint32_t)tmp.size();
      for (uint32_t l : tmp) lines->push_back((l & 0x80000000u) ? S.node_count + (l & 0x7fffffffu) : l);   // one 64-byte line address space: nodes, then slots
      rays->push_back(r);
    }
}
struct ChunkRef { uint32_t first, count; };   // rays order[first .. first + count)
static void l2_replay(const std::vector<ProbeRay>& rays, const std::vector<uint32_t>& lines, const std::vector<uint32_t>& order,
                      const std::vector<std::vector<ChunkRef>>& queues /* 1 (shared) or 8 (one per XCD) */, uint32_t waves_per_xcd, uint32_t l2_bytes,
                      uint32_t line_space, double out[6]) {
  struct Wave { uint32_t next_c = 0, end_c = 0, first = 0, count = 0, step = 0, maxlen = 0; bool active = false, counted = false; };
  std::vector<L2Sim> l2; for (int x = 0; x < 8; x++) l2.emplace_back(l2_bytes);
  std::vector<std::vector<Wave>> waves(8, std::vector<Wave>(waves_per_xcd));
  std::vector<uint32_t> cursor(queues.size(), 0);
  uint32_t stream_line = line_space;   // queue entries: lines nobody has seen before
  double n_rays = 0, touches = 0, misses = 0, streamed = 0, wave_steps = 0, wave_steps_missing = 0;
  bool any = true;
  while (any) {
    any = false;
    for (uint32_t w = 0; w < waves_per_xcd; w++)
      for (int x = 0; x < 8; x++) {
        Wave& wv = waves[x][w];
        const size_t q = queues.size() == 1 ? 0 : (size_t)x;
        if (!wv.active) {
          if (wv.next_c == wv.end_c) {
            if (cursor[q] >= queues[q].size()) continue;
            wv.next_c = cursor[q]; cursor[q] += 8; wv.end_c = std::min<uint32_t>(cursor[q], (uint32_t)queues[q].size());
          }
          const ChunkRef c = queues[q][wv.next_c];
          wv.counted = wv.next_c * 5u >= queues[q].size();
          wv.next_c++;
          wv.first = c.first; wv.count = c.count; wv.step = 0; wv.maxlen = 0; wv.active = true;
          for (uint32_t k = 0; k < c.count; k++) wv.maxlen = std::max(wv.maxlen, rays[order[c.first + k]].count);
          // the chunk's rays: rayO + rayD = 32 B each = 32 lines per 64 rays, streamed in (misses by construction); hit records (16 lines) are written
          for (uint32_t k = 0; k < (c.count * 32u + 63u) / 64u + (c.count * 16u + 63u) / 64u; k++) (void)l2[x].touch(stream_line++);
          if (wv.counted) { n_rays += c.count; streamed += (c.count * 48.0) / 64.0; }
        }
        bool step_missed = false;
        for (uint32_t k = 0; k < wv.count; k++) {
          const ProbeRay& r = rays[order[wv.first + k]];
          if (wv.step < r.count) {
            const bool hit = l2[x].touch(lines[r.first + wv.step]);
            if (wv.counted) { touches += 1; misses += hit ? 0 : 1; }
            step_missed |= !hit;
          }
        }
        if (wv.counted) { wave_steps += 1; wave_steps_missing += step_missed ? 1 : 0; }   // a wave-step waits for its slowest lane
        wv.step++;
        if (wv.step >= wv.maxlen) wv.active = false;
        any = true;
      }
  }
  out[0] = n_rays; out[1] = touches; out[2] = misses; out[3] = streamed; out[4] = wave_steps; out[5] = wave_steps_missing;
}
static void l2_three_modes(const std::vector<ProbeRay>& rays, const std::vector<uint32_t>& lines, const std::vector<uint32_t>& arrival_order,
                           const std::vector<uint32_t>& seg_end /* arrival_order positions where a segment ends */, uint32_t line_space,
                           uint32_t waves_per_xcd, uint32_t l2_bytes, uint32_t cell_bits, double out[18]) {
  auto cut = [](uint32_t first, uint32_t end, std::vector<ChunkRef>* q) { for (uint32_t c = first; c < end; c += 64) q->push_back({c, std::min(64u, end - c)}); };
  // mode 0: segment order, chunks never straddle segments
  {
    std::vector<std::vector<ChunkRef>> q(1);
    uint32_t at = 0;
    for (uint32_t e_ : seg_end) { cut(at, e_, &q[0]); at = e_; }
    l2_replay(rays, lines, arrival_order, q, waves_per_xcd, l2_bytes, line_space, &out[0]);
  }
  // origin cells
  float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
  for (auto& r : rays) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], r.o[a]); hi[a] = fmaxf(hi[a], r.o[a]); }
  const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), fmaxf(hi[2] - lo[2], 1e-20f));
  const float cells = (float)(1u << cell_bits);
  std::vector<uint64_t> key(rays.size());
  for (size_t i = 0; i < rays.size(); i++) {
    uint64_t c[3];
    for (int a = 0; a < 3; a++) c[a] = (uint64_t)fminf(cells - 1.0f, (rays[i].o[a] - lo[a]) / ext * cells);
    key[i] = expand21(c[0]) << 2 | expand21(c[1]) << 1 | expand21(c[2]);
  }
  std::vector<uint32_t> sorted(arrival_order);
  std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
  {
    std::vector<std::vector<ChunkRef>> q(8);
    const uint32_t n = (uint32_t)sorted.size();
    for (uint32_t x = 0; x < 8; x++) cut((uint32_t)((uint64_t)n * x / 8), (uint32_t)((uint64_t)n * (x + 1) / 8), &q[x]);
    l2_replay(rays, lines, sorted, q, waves_per_xcd, l2_bytes, line_space, &out[6]);
  }
  {
    std::vector<std::vector<ChunkRef>> q(1);
    cut(0, (uint32_t)sorted.size(), &q[0]);
    l2_replay(rays, lines, sorted, q, waves_per_xcd, l2_bytes, line_space, &out[12]);
  }
}
void emu_l2_probe(void* h, uint32_t bounce, uint32_t ns, uint32_t s0, uint32_t T, uint32_t waves_per_xcd, uint32_t l2_bytes, uint32_t cell_bits,
                  uint32_t threads, double out[18]) {
  Emu* e = (Emu*)h;
  const DeviceScene& S = e->S;
  for (int i = 0; i < 18; i++) out[i] = 0;
  if (!S.wide6) return;
  const uint32_t tilesX = (S.width + 7) / 8, tilesY = (S.height + 7) / 8, tiles = tilesX * tilesY, per_band = tiles / 4;
  // today's table order: segment sg = 4 * idx + band  ->  tile band * per_band + idx (kernels.hip segment_first_tile)
  std::vector<uint32_t> seg_tile;
  for (uint32_t idx = s0; idx < s0 + T && idx < per_band; idx++) for (uint32_t band = 0; band < 4; band++) seg_tile.push_back(band * per_band + idx);
  std::vector<std::vector<ProbeRay>> tr(seg_tile.size());
  std::vector<std::vector<uint32_t>> tl(seg_tile.size());
  std::atomic<uint32_t> cursor{0};
  auto worker = [&]() {
    std::vector<uint32_t> lds(std::max(kLdsStack, kLdsStack6) + 1), spill(kSpillStack), pend(std::max(kPendLeaves, kPendLeaves6) + 1);
    for (;;) {
      const uint32_t i = cursor.fetch_add(1);
      if (i >= seg_tile.size()) return;
      gen_tile_rays(e, seg_tile[i] % tilesX, seg_tile[i] / tilesX, ns, bounce, &tr[i], &tl[i], lds.data(), spill.data(), pend.data());
    }
  };
  { std::vector<std::thread> pool; for (uint32_t i = 1; i < std::max(1u, threads); i++) pool.emplace_back(worker); worker(); for (auto& t : pool) t.join(); }
  std::vector<ProbeRay> rays; std::vector<uint32_t> lines, order, seg_end;
  for (size_t i = 0; i < seg_tile.size(); i++) {
    const uint32_t base = (uint32_t)lines.size();
    for (auto r : tr[i]) { r.first += base; order.push_back((uint32_t)rays.size()); rays.push_back(r); }
    lines.insert(lines.end(), tl[i].begin(), tl[i].end());
    seg_end.push_back((uint32_t)rays.size());
    std::vector<ProbeRay>().swap(tr[i]); std::vector<uint32_t>().swap(tl[i]);
  }
  if (rays.empty()) return;
  l2_three_modes(rays, lines, order, seg_end, S.node_count + S.slot_count, waves_per_xcd, l2_bytes, cell_bits, out);
}
void emu_get_wide(double out[26]) {
  int i = 0;
  const double r = g_wide.rays ? (double)g_wide.rays : 1.0;
  for (int N : {4, 6, 8}) for (int rule = 0; rule < 3; rule++) { out[i++] = g_wide.nodes[N][rule] / r; out[i++] = g_wide.tris[N][rule] / r; }
  for (int M = 2; M <= 4; M++) { out[i++] = g_wide.nodes[M - 1][0] / r; out[i++] = g_wide.tris[M - 1][0] / r; }
  out[24] = (double)g_wide.rays; out[25] = (double)g_wide.mismatch;
  g_wide = WideCounts{};
}
void emu_get_stack_hist(unsigned long long out[64]) { for (int i = 0; i < 64; i++) { out[i] = g_stack_hist[i]; g_stack_hist[i] = 0; } }
void emu_get_probe(unsigned long long out[6]) {
  out[0] = g_probe.nodes; out[1] = g_probe.nodes_cull; out[2] = g_probe.tris; out[3] = g_probe.tris_cull; out[4] = g_probe.rays; out[5] = g_probe.mismatch;
  g_probe = ProbeCounts{};
}
uint32_t emu_slot_count(void* h) { return ((Emu*)h)->S.slot_count; }
// per-bounce counts of the closest-hit rays of emu_debug_sample (r5 probe): out[16][4] = {nodes, triangle tests, leaf fetches, rays}
void emu_get_counts_per_bounce(unsigned long long out[64]) {
  for (int b = 0; b < 16; b++) { out[4 * b] = g_pb_nodes[b]; out[4 * b + 1] = g_pb_tris[b]; out[4 * b + 2] = g_pb_leaves[b]; out[4 * b + 3] = g_pb_rays[b]; g_pb_nodes[b] = g_pb_tris[b] = g_pb_leaves[b] = g_pb_rays[b] = 0; }
}
void emu_get_counts(unsigned long long out[4]) { out[0] = g_nodes; out[1] = g_tris; out[2] = g_rays; out[3] = g_leaves; g_nodes = g_tris = g_rays = g_leaves = 0; }
float emu_halton(void* h, uint32_t i, uint32_t d) { return halton(halton_table(((Emu*)h)->halton.data()), i, d); }

}  // extern "C"
