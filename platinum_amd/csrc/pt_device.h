// pt_device.h — HBM data layout of the wavefront path tracer (see DESIGN.md "Data layout in HBM").
//
// Scene tables keep the reference's record layouts (pt_shader_defs.hpp, core/mesh.hpp) so the snapshot is
// uploaded by plain memcpy; what the reference reaches through per-mesh / per-instance pointer tables
// (VertexResource / PrimitiveResource / InstanceResource, pt_shader_defs.hpp:117-128) is reached here through
// base offsets into concatenated arrays.
#pragma once
#include "../../include/ptamd.h"
#include "pt_math.h"

namespace pt {

// ---- geometry tables ------------------------------------------------------------------------------------------
struct MeshInfo {          // replaces VertexResource + PrimitiveResource for one mesh
  uint32_t vertex_base;    // into positions[] / vdata[]
  uint32_t tri_base;       // into indices[] (x3) / slots[]
  uint32_t tri_count;
  uint32_t _pad;
};

struct alignas(16) InstanceInfo {  // MTLAccelerationStructureInstanceDescriptor + InstanceResource, 64 B
  float c0[3], c1[3], c2[3], c3[3];  // object->world, four packed columns (renderer_pt.cpp:706-716)
  uint32_t mesh;                     // accelerationStructureIndex
  uint32_t material_base;            // into materials[]: this instance's MaterialGPU array
  uint32_t tri_global_base;          // global id of this instance's first flattened triangle
  uint32_t flags;                    // kInstanceNonOpaque: candidates go through the alpha test (renderer_pt.cpp:714-729)
};
constexpr uint32_t kInstanceNonOpaque = 1u;
static_assert(sizeof(InstanceInfo) == 64, "InstanceInfo");

// ---- BVH -------------------------------------------------------------------------------------------------------
// Leaf slot of the BVH: ONE world-space triangle, or (r4) TWO triangles of one instance that share an edge (every quad of a lat / long
// sphere, every wall: consecutive triangles of a mesh with two vertex indices in common — host_scene.h pair_mesh_triangles).  64 bytes,
// 64-byte aligned: a leaf test is ONE 64-byte L2 request whether it tests one triangle or two.  The slot holds the four world-space
// corners (fp32 transformPoint of the mesh vertices, the intersection contract); the edges e1 = v1 - v0, e2 = v2 - v0 that Moeller-Trumbore
// consumes are formed in registers with the very subtraction k_flatten used to store, so the test sees the same fp32 operands as before and
// the contract (min t, ties to the lowest global id) is untouched.  Triangle A = (q0, q1, q2) in ITS vertex order; triangle B's corners
// are three of q0..q3 in B's own order, named by `code`.  r3 kept one triangle per slot (v0, e1, e2): 66 MB of slots on C3, now 34 MB,
// and the bottom level of the tree went with them.
// Layout: four 16-byte quarters, one corner each, a word of metadata in the fourth lane of the first three.  Triangle A needs quarters
// 0..2 (three dwordx4 loads: what r3's one-triangle record cost); triangle B's corners are fetched AFTER A's test, one 12-byte load per
// corner at quarter `code`, from the line A's loads have just brought into L1 — holding all four corners in registers across A's test cost
// the 72-VGPR trace kernels 11 more spilled registers and 37 % of their speed (measured, r4).
struct alignas(64) TriRec {
  float q0[3]; uint32_t gid_a;      // (global id << 2) | material class; global id = InstanceInfo.tri_global_base + prim is the closest-hit
                                    // tie-break key (the class bits sit below it, so comparing this field orders by global id)
  float q1[3]; uint32_t gid_b;      // the same for triangle B; kInvalidRef: the slot holds one triangle
  float q2[3]; uint32_t inst_code;  // instance id (both triangles) in bits [25:0]; bits [27:26], [29:28], [31:30]: which quarter holds B's v0, v1, v2
  float q3[3]; uint32_t _pad;
};
static_assert(sizeof(TriRec) == 64, "TriRec");
constexpr uint32_t kSlotInstBits = 26, kSlotInstMask = (1u << kSlotInstBits) - 1u;  // < 2^26 instances (checked at pt_start_render)
// A triangle is named by 2 * slot + half (half 1 = triangle B) wherever one is referred to: RayHit::tri, the hit record, shade_recs[].
// Hit record word 3: triangle index (28 bits) | material class << 28; kInvalidRef = miss.  The class (which BSDF lobes the
// material can take: see material_class) lets k_shade put hits of one kind into one wave.
constexpr uint32_t kHitTriMask = 0x0fffffffu;
constexpr uint32_t kShadeClasses = 4;

// What shading needs to know about a hit triangle, resolved once per render (k_shade_records) and indexed like tris[]:
// absolute vertex indices, the absolute index of its MaterialGPU, its instance, and the world-space geometric normal
// (kernel.metal:150-162: normalize(M * normalize(cross(p1 - p0, p2 - p0))) — the same operations in the same order as the
// per-hit formulation, evaluated once per flattened triangle).  32 B = one dwordx4 pair.  Without it every shaded hit walks
// tris -> instances -> meshes -> indices / slots -> vertices / materials (five dependent loads instead of two) and
// re-reads three positions only to rebuild this normal.
struct alignas(16) ShadeRec { uint32_t v[3]; uint32_t material; uint32_t inst; float ng[3]; };
static_assert(sizeof(ShadeRec) == 32, "ShadeRec");

// A read of scene-table memory that is KNOWN to be global (HBM).  k_shade receives the DeviceScene table through a
// pointer, so the pointers inside it are "flat" to the compiler: plain dereferences become flat_load (aperture check,
// both wait counters, 64-bit address VGPRs).  Going through address space 1 yields global_load with an SGPR base.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
__device__ __forceinline__ T ldg(const T* p) {
  T r;
  __builtin_memcpy(&r, (const __attribute__((address_space(1))) void*)p, sizeof(T));
  return r;
}
#else
template <class T>
PT_HD T ldg(const T* p) { return *p; }
#endif

// 4-wide BVH node with child boxes quantised to 8 bits per coordinate relative to the node's own box, 64 B =
// 4 x dwordx4 (one half of a 128-B L2 line).  The traversal kernels are bound by vector-L1 lookups,
// so the node packs four children into the bytes a binary node needed for two.
//   child k box:  lo_a = origin[a] + byte_k(qlo[a]) * scale[a],  hi_a = origin[a] + byte_k(qhi[a]) * scale[a]
//   (one dword per axis and side, child k in byte k: the traversal picks the near/far dword of an axis by the sign of
//   the ray direction with one select and converts bytes with v_cvt_f32_ubyte0..3)
//   scale[a] = 2^(exp[a] - 127)  (a power of two: the product is exact, the sum rounds once — the builder
//   quantises against exactly this expression, so the decoded box always contains the exact child box)
//   ref: kInvalidRef = empty slot; bit31 set = leaf (index into tris[]); else index of an internal node.
struct alignas(16) BvhNode {
  float origin[3];
  uint8_t exp[3];
  uint8_t _pad0;
  uint32_t ref[4];
  uint32_t qlo[3];
  uint32_t qhi[3];
  uint32_t _pad1[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode");
// The 6-wide node of the one-BVH structure (r3; DeviceScene::wide6): the same 64 bytes hold SIX children, so a ray takes ~20 % fewer
// node steps for the same memory traffic per step.  What makes room: a node's internal children are consecutive records and its leaf
// children consecutive triangles (the builder numbers both that way), so two base indices and two counts replace four 32-bit refs.
// Children 0 .. n_int-1 are nodes base_node + k, children n_int .. n_int+n_leaf-1 are triangles base_leaf + (k - n_int).
struct alignas(16) BvhNode6 {
  float origin[3];
  uint8_t exp[3];
  uint8_t counts;        // n_int | n_leaf << 3
  uint32_t base_node;
  uint32_t base_leaf;    // < 2^26 (a leaf-queue entry is base_leaf << 6 | hit mask)
  uint32_t q[3][3];      // per axis: {lo of children 0..3, hi of children 0..3, lo4 | lo5 << 8 | hi4 << 16 | hi5 << 24}
  uint32_t _pad;
};
static_assert(sizeof(BvhNode6) == 64, "BvhNode6");
constexpr uint32_t kMaxWide6Triangles = 1u << 26;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kInvalidRef = 0xffffffffu;

// ---- two-level structure (instanced scenes; replaces the reference's BLAS per mesh + TLAS over instances,
//      renderer_pt.cpp:653-749, for scenes that repeat a mesh many times) --------------------------------------------
// One node array holds the TLAS (over the world boxes of the instances) followed by one BLAS per mesh (object space).
// TLAS leaves are child refs `kInstBit | instance`: they are ordered and stacked like inner nodes and ENTERED when they
// come up (the ray is taken to object space for the slab tests; kExitMarker on the stack brings it back).  BLAS leaves are
// `kLeafBit | primitive`; the candidate that goes to the triangle queue is the flattened triangle `tri_base + primitive`,
// and the triangle test itself stays the world-space one (tris[] in flattening order), so hits do not change by a bit.
constexpr uint32_t kInstBit = 0x40000000u;
constexpr uint32_t kExitMarker = 0xfffffffeu;
struct alignas(16) InstanceTrav {  // 64 B: what entering an instance needs
  float ic0[3], ic1[3], ic2[3];    // columns of the inverse of the instance's 3x3 (computed in double, rounded once)
  float c[3];                      // its translation column
  uint32_t tri_base;               // InstanceInfo::tri_global_base
  uint32_t mesh;                   // index into mesh_trav[]
  uint32_t _pad[2];
};
static_assert(sizeof(InstanceTrav) == 64, "InstanceTrav");
struct alignas(16) MeshTrav { uint32_t root_ref, _pad; float lo[3], hi[3]; };  // 32 B: BLAS root, object-space bounds
static_assert(sizeof(MeshTrav) == 32, "MeshTrav");

// ---- one record per area light: what sampleLightPower / sampleAreaLight (kernel.metal:379-435) read ---------------
// The reference walks light -> instance -> mesh -> three vertices for every light sample; the walk and the light's
// world-space normal do not depend on the sample, so they are resolved once per render (make_light_rec, the same
// arithmetic in the same order) and the stage reads ONE record — from LDS when the table has at most 64 entries.
struct alignas(16) LightRec {   // 128 B
  float q0[3], area;            // object-space vertices (AreaLight::indices resolved), AreaLight::area
  float q1[3], power;
  float q2[3], cumulativePower;
  float c0[3], e0;              // the instance's object->world columns; e = AreaLight::emission
  float c1[3], e1;
  float c2[3], e2;
  float c3[3], _pad0;
  float n[3], _pad1;            // normalize(transformVec(cross(q1 - q0, q2 - q0)))
};
static_assert(sizeof(LightRec) == 128, "LightRec");

// ---- sampler table: one entry per Halton dimension (defs.metal:115-194 holds the 620 primes) ---------------------
// 32-bit integer multiplies are quarter rate on CDNA, so the radical inverse peels `digits` base-`prime` digits per
// division by chunk = prime^digits (the largest power below 2^22) and splits the remainder with exact fp32 arithmetic; a
// quotient below `chunk` is the last remainder as it stands (one division covers a 32-bit index for every prime except
// 163..251 and those from 2048 on, which need two).  The division is the round-up multiply-shift of
// Granlund & Montgomery with a 33-bit multiplier 2^32 + magic, l = ceil(log2 chunk):
//   t = mulhi(magic, n);  q = (t + ((n - t) >> 1)) >> (l - 1)          exact for every 32-bit n
// — ONE quarter-rate multiply per chunk (q * chunk is a 24-bit multiply: q < 2^24 because chunk >= 257, chunk < 2^22).
struct alignas(16) HaltonEntry {
  uint32_t chunk;    // prime^digits, 257 <= chunk < 2^22
  uint32_t magic;    // floor(2^32 * (2^l - chunk) / chunk) + 1
  uint32_t shift;    // l - 1
  float inv;         // 1.0f / (float)prime  (the reference's invB, samplers.metal:172)
  float primef;      // (float)prime
  uint32_t digits;   // digits peeled per chunk (1 for primes >= 257)
  uint32_t prime;
  float hinv;        // 0.5f * inv (exact)
};
constexpr int kHaltonDims = 620;

// ---- LUTs (pt_shader_defs.hpp:130-139) -----------------------------------------------------------------------
struct Lut { const float* d; int w, h, depth; int lds; };  // lds != 0: `d` points at a copy a block staged in LDS (k_shade)
struct LutSet { Lut E, Eavg, EMs, EavgMs, ETransIn, ETransOut; };  // the two *avgTrans tables are never sampled

// ---- scene textures: sampled with repeat + bilinear over LINEAR FLOAT texels (the texture contract) ---
// r4: 8-bit textures MAY stay 8-bit in HBM (4 / 2 / 1 bytes per texel instead of the 16 of a decoded float4) and are then decoded on fetch
// through two 256-entry tables that hold the VERY floats the host decode produces (unorm[b] = (float)b / 255.0f; srgb[b] = the piecewise
// EOTF evaluated in double and rounded once): the filter sees the same operands, the result is the same bit for bit.  host_scene.h
// decode_textures picks the form per scene by footprint.  RGBA32F textures (the environment) are stored as they come.
struct TexInfo { uint32_t offset16, w, h, format; };  // offset16: byte offset into DeviceScene::tex_data / 16; format: PT_TEX_*
constexpr uint32_t kTexDecodeUnorm = 0, kTexDecodeSrgb = 256, kTexDecodeEntries = 512;

struct Mat3 { vec3 c0, c1, c2; };
PT_HD vec3 mul(const Mat3& m, vec3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }

// Everything a kernel needs to know about the scene; passed by value as a kernel argument (constant/SGPR space).
struct DeviceScene {
  const pt_float3* positions;
  const pt_vertex_data* vdata;
  const uint32_t* indices;
  const uint32_t* slots;
  const MeshInfo* meshes;
  const InstanceInfo* instances;
  const pt_material_gpu* materials;
  const pt_area_light* lights;
  const BvhNode* nodes;
  const TriRec* tris;
  const ShadeRec* shade_recs;  // 2 * slot_count records: entry 2 * slot + half (the B entry of a one-triangle slot is unused)
  const LightRec* light_recs;  // lightCount records, same order as lights[]
  const float* light_cdf;      // lightCount values: AreaLight::cumulativePower on its own (what sampleLightPower's binary search reads)
  uint32_t tri_count;      // flattened triangles
  uint32_t slot_count;     // leaf slots in tris[] (<= tri_count: a slot holds one or two triangles)
  uint32_t root_ref;       // kLeafBit|0 for a single-triangle scene, the root's node index otherwise, kInvalidRef when empty
  const InstanceTrav* inst_trav;  // two-level structure only (two_level != 0): root_ref is the TLAS root, tris[] is in flattening order
  const MeshTrav* mesh_trav;
  uint32_t two_level;
  uint32_t wide6;          // nodes[] holds BvhNode6 records (one-BVH structure only)
  uint32_t node_count;     // records in nodes[] (a small two-level structure is staged in LDS by the trace kernels)
  const HaltonEntry* halton;
  LutSet luts;
  const uint8_t* tex_data;          // every texture in its own format, each starting at a multiple of 16 bytes
  const float* tex_decode;          // kTexDecodeEntries floats: unorm[256], srgb[256]
  const TexInfo* textures;
  uint32_t tex_native;              // some texture kept an 8-bit form: taps decode per format (0: every texel is a float4)
  const pt_alias_entry* env_alias;  // EnvironmentLight::alias (pt_shader_defs.hpp:70-73)
  int32_t env_texture;              // -1: no environment light
  uint32_t envLightCount;           // 0 or 1
  uint32_t has_alpha;               // any non-opaque instance: the trace kernels evaluate the alpha-test payload
  // Constants (pt_shader_defs.hpp:105-115) — the fields the kernels read
  pt_camera_data camera;
  Mat3 idt;
  uint32_t width, height;
  uint32_t lightCount;
  float totalLightPower;
  int32_t flags;            // RendererFlags
  uint32_t integrator;      // PT_INTEGRATOR_*
  uint32_t max_bounces;
};

// ---- wavefront state (SoA, one element per path slot; ping-pong between bounces) ---------------------------------
// rayO.w  = pdf of the BSDF sample that generated this ray (lastSample.pdf, kernel.metal:574)
// rayD.w  = bits: [9:0] next Halton dimension, [10] lastSample was specular (kernel.metal:561), [31:11] the path's entry of the
//           per-sample radiance buffer, relative to the window of its segment (kernels.hip lbuf_index; a path never leaves its segment)
// att.w   = bits: Halton offset of this (pixel, sample) (samplers.metal:154-156)
struct PathState {
  vec4* rayO;
  vec4* rayD;
  vec4* att;
};
constexpr uint32_t kMetaDimMask = 0x3ffu;
constexpr uint32_t kMetaSpecular = 1u << 10;
constexpr uint32_t kMetaPidShift = 11;  // bits 11..31: the path's index into Lbuf RELATIVE to its segment's window (< 2^21)

struct ShadowQueue {
  vec4* o;        // origin.xyz, tmax
  vec4* d;        // direction.xyz, bits(pid)
  vec4* contrib;  // attenuation * Ld (kernel.metal:631-637), added to the path's radiance if unoccluded
};

// Per-batch device counters (zeroed by one memset per batch): trace-kernel cursors and the longest segment per queue.
struct BatchCounters {
  uint32_t chunks_closest[64];  // non-empty closest-hit chunks entering bounce b   (b <= 50)
  uint32_t chunks_shadow[64];   // non-empty shadow chunks at bounce b
  uint32_t work_closest[64];  // ordered-claim cursors of the trace kernels
  uint32_t work_shadow[64];
  uint32_t work_shade[64];    // segment-claim cursor of k_shade at bounce b
  uint32_t shaded;        // hits shaded
  uint32_t nonfinite;     // samples with NaN/inf radiance seen by k_accumulate
  uint32_t _pad[2];
  unsigned long long nodes_closest, tris_closest, nodes_shadow, tris_shadow;  // instrumented runs only
#ifdef PT_TAIL_PROBE   // analysis build only (tools/build_variant.sh tail -DPT_TAIL_PROBE): when do the waves of a persistent launch run out of work?
  unsigned long long tail_end_max[3][16], tail_end_sum[3][16], tail_start_inv[3][16], tail_waves[3][16];   // [closest, shade, shadow][bounce]; 100 MHz ticks
  unsigned long long tail_take[3][16], tail_setup[3][16], tail_busy[3][16];   // trace kernels: wave time in ChunkClaims::take, in ray load + trav_init, in all
#endif
};

struct Totals {  // running totals since pt_start_render (folded from BatchCounters after each batch)
  unsigned long long closest_rays, shadow_rays, shaded_hits, paths;
  unsigned long long nodes_closest, tris_closest, nodes_shadow, tris_shadow;
  unsigned long long counted_closest, counted_shadow;
  unsigned long long nonfinite;
};

}  // namespace pt
