/*
 * ptamd.h — C ABI of libptamd.so, the MI355X (gfx950) wavefront path tracer that sits behind the
 * `pt::renderer_pt::Renderer` operator surface of teofum/platinum.
 *
 * The reference has no FFI layer: `Renderer` (src/renderer_pt/renderer_pt.hpp:28-73) is a concrete C++ class
 * that the SDL2/ImGui frontend calls directly and that pulls the scene out of `Store&`.  This header is what a
 * binding for that class would bind: one entry point per public `Renderer` member, Metal handles replaced by
 * plain memory, the scene passed as a flat snapshot whose records keep the reference's exact byte layouts
 * (src/renderer_pt/pt_shader_defs.hpp, src/core/mesh.hpp) so the frontend's buffers can be handed over as-is.
 *
 * Plain C: pointers and sizes only, no C++/torch/HIP types in any signature.
 * Threading: one renderer = one caller thread (the reference is single-threaded, frontend.cpp:188-270).
 * Errors: every call returns PT_OK (0) or a negative pt_error; pt_last_error() returns the message of the last
 * failure on the calling thread (the reference prints to stderr and asserts: metal_utils.mm:172-214).
 * There is NO CPU fallback: pt_create fails with PT_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef PTAMD_H
#define PTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 4u /* 4: pt_get_runtime_info, PT_ERR_RUNTIME_CONFLICT, pt_stats::leaf_slots */

/* ---------------------------------------------------------------------------------------------------------- */
/* Enums (same numeric values as the reference)                                                                 */

/* renderer_pt.hpp:21-26  enum Status */
enum { PT_STATUS_BLOCKED = 0, PT_STATUS_READY = 1, PT_STATUS_BUSY = 4, PT_STATUS_DONE = 8 };
/* renderer_pt.hpp:16-19  enum Integrators (kernel.metal:256 pathtracingKernel, :473 misKernel) */
enum { PT_INTEGRATOR_SIMPLE = 0, PT_INTEGRATOR_MIS = 1 };
/* pt_shader_defs.hpp:75-79  enum RendererFlags */
enum { PT_FLAG_NONE = 0, PT_FLAG_MULTISCATTER_GGX = 1 << 0, PT_FLAG_GMON = 1 << 1 };
/* A path whose radiance is NaN/inf (the reference's BSDF can produce one: e.g. a NaN pdf in the rough-glass eval,
 * about 1 path in 5e8 on C2) poisons its pixel's running mean for good in the reference (kernel.metal:672-684).
 * PROPAGATE keeps that behaviour (parity default); ZERO counts the sample as black and reports it in pt_stats. */
enum { PT_NONFINITE_PROPAGATE = 0, PT_NONFINITE_ZERO = 1 };
/* pt_shader_defs.hpp:85-90  MaterialGPU::MaterialFlags */
enum {
  PT_MATERIAL_THIN_DIELECTRIC = 1 << 0,
  PT_MATERIAL_USE_ALPHA = 1 << 1,
  PT_MATERIAL_EMISSIVE = 1 << 2,
  PT_MATERIAL_ANISOTROPIC = 1 << 3
};

typedef enum pt_error {
  PT_OK = 0,
  PT_ERR_INVALID_ARGUMENT = -1,
  PT_ERR_NO_DEVICE = -2,     /* no usable HIP device: the library never falls back to the CPU */
  PT_ERR_HIP = -3,           /* a HIP runtime call failed; message has file:line and hipGetErrorString */
  PT_ERR_OUT_OF_MEMORY = -4,
  PT_ERR_BAD_STATE = -5,     /* e.g. pt_render_step before pt_start_render */
  PT_ERR_UNSUPPORTED = -6,   /* a scene feature that is not implemented */
  PT_ERR_BAD_LUT = -7,
  PT_ERR_RUNTIME_CONFLICT = -8 /* more than one HIP runtime mapped into the process (pt_get_runtime_info says which) */
} pt_error;

/* ---------------------------------------------------------------------------------------------------------- */
/* Scene snapshot: what crosses the ABI instead of `Store&`.  Byte layouts are the reference's.                */

/* simd float3: 16-byte stride (renderer_pt.cpp:231, core/mesh.cpp:67-69) */
typedef struct pt_float3 { float x, y, z, _pad; } pt_float3;

/* core/mesh.hpp:17-21  VertexData, 48 B: normal @0, tangent(xyzw) @16, texCoords @32 */
typedef struct pt_vertex_data {
  pt_float3 normal;
  float tangent[4];
  float texCoords[2];
  float _pad[2];
} pt_vertex_data;

/* pt_shader_defs.hpp:84-103  MaterialGPU, 96 B */
typedef struct pt_material_gpu {
  float baseColor[4];          /* @0  */
  pt_float3 emission;          /* @16 */
  float emissionStrength;      /* @32 */
  float roughness;             /* @36 */
  float metallic;              /* @40 */
  float transmission;          /* @44 */
  float ior;                   /* @48 */
  float anisotropy;            /* @52 */
  float anisotropyRotation;    /* @56 */
  float clearcoat;             /* @60 */
  float clearcoatRoughness;    /* @64 */
  int32_t flags;               /* @68 */
  int32_t baseTextureId;       /* @72 */
  int32_t rmTextureId;         /* @76 */
  int32_t transmissionTextureId; /* @80 */
  int32_t clearcoatTextureId;  /* @84 */
  int32_t emissionTextureId;   /* @88 */
  int32_t normalTextureId;     /* @92 */
} pt_material_gpu;

/* One mesh = the four shared buffers of core/mesh.hpp:23-60.  `indices` doubles as PrimitiveData[]
 * (pt_shader_defs.hpp:48-50, renderer_pt.cpp:237-239). */
typedef struct pt_mesh {
  const pt_float3* positions;        /* vertex_count x 16 B */
  const pt_vertex_data* vertex_data; /* vertex_count x 48 B */
  const uint32_t* indices;           /* 3 * triangle_count */
  const uint32_t* material_slots;    /* triangle_count (core/mesh.hpp:32,55) */
  uint32_t vertex_count;
  uint32_t triangle_count;
} pt_mesh;

/* MTLAccelerationStructureInstanceDescriptor, 64 B packed (filled at renderer_pt.cpp:706-739):
 * 4 columns x packed float3 @0, options @48, mask @52, intersectionFunctionTableOffset @56,
 * accelerationStructureIndex (= mesh index) @60 */
typedef struct pt_instance {
  float transform[4][3];
  uint32_t options;
  uint32_t mask;
  uint32_t intersectionFunctionTableOffset;
  uint32_t accelerationStructureIndex;
} pt_instance;

/* InstanceResource (pt_shader_defs.hpp:126-128): the per-INSTANCE material array, indexed by the
 * triangle's material slot (renderer_pt.cpp:560-640 duplicates materials per instance). */
typedef struct pt_instance_materials {
  const pt_material_gpu* materials;
  uint32_t material_count;
  uint32_t _pad;
} pt_instance_materials;

/* Camera node: world matrix (scene.cpp:515-534, column-major float4x4) + core/camera.hpp:10-18.
 * The library derives CameraData exactly as Renderer::updateConstants (renderer_pt.cpp:965-1021). */
typedef struct pt_camera {
  float world[4][4];        /* columns */
  float sensor_size[2];     /* mm, default {36,24} */
  float focal_length;       /* mm */
  float aperture;           /* f-number, 0 = pinhole */
  uint32_t aperture_blades;
  float roundness;
  float bokeh_power;
  float focus_distance;
} pt_camera;

/* core/colorspace.hpp:22-42: CIE xy chromaticities of the primaries and the white point */
typedef struct pt_colorspace { float r[2], g[2], b[2], w[2]; } pt_colorspace;

/* Scene textures. The reference uploads these pixel formats (loaders/texture.cpp:30-48) and samples
 * them with address::repeat + filter::linear (bsdf.metal:24, kernel.metal:167, intersections.metal:33). */
enum {
  PT_TEX_RGBA8_SRGB = 0, /* MTL::PixelFormatRGBA8Unorm_sRGB: base colour / emission (decoded to linear before filtering) */
  PT_TEX_RGBA8 = 1,      /* RGBA8Unorm: linear RGB (normal maps) */
  PT_TEX_RG8 = 2,        /* RG8Unorm: roughness, metallic */
  PT_TEX_R8 = 3,         /* R8Unorm: transmission / clearcoat */
  PT_TEX_RGBA32F = 4     /* RGBA32Float: HDR environment maps */
};
typedef struct pt_texture {
  const void* pixels;    /* row-major, top row first, tightly packed */
  uint32_t width, height;
  uint32_t format;       /* PT_TEX_* */
  uint32_t _pad;
} pt_texture;

/* core/environment.hpp:15-19 AliasEntry, 12 B */
typedef struct pt_alias_entry { float pdf, p; uint32_t aliasIdx; } pt_alias_entry;

typedef struct pt_scene_snapshot {
  const pt_mesh* meshes;
  uint32_t mesh_count;
  uint32_t instance_count;
  const pt_instance* instances;                    /* instance_count */
  const pt_instance_materials* instance_materials; /* instance_count */
  pt_camera camera;
  const pt_texture* textures;                      /* texture_count; MaterialGPU::*TextureId index this array */
  uint32_t texture_count;
  int32_t env_texture;                             /* Environment::textureId (scene envmap), -1 = none */
  const pt_alias_entry* env_alias;                 /* width*height entries, or NULL: built by the library exactly as
                                                      Environment::rebuildAliasTable (core/environment.cpp:5-91) */
} pt_scene_snapshot;

/* ---------------------------------------------------------------------------------------------------------- */
/* Derived device constants, exported for parity checks (pt_get_constants)                                     */

/* pt_shader_defs.hpp:52-61 CameraData, 80 B */
typedef struct pt_camera_data {
  pt_float3 position, topLeft, pixelDeltaU, pixelDeltaV;
  float apertureRadius;
  uint32_t apertureBlades;
  float apertureRoundness;
  float bokehPower;
} pt_camera_data;

/* pt_shader_defs.hpp:105-115 Constants, 176 B */
typedef struct pt_constants {
  uint32_t frameIdx, spp, gmonBuckets;
  uint32_t lightCount;
  uint32_t envLightCount;
  uint32_t lutSizeE, lutSizeEavg;
  int32_t flags;
  float totalLightPower;
  uint32_t _pad0;
  uint32_t size[2];
  pt_float3 idt[3];  /* float3x3 columns */
  pt_camera_data camera;
} pt_constants;

/* pt_shader_defs.hpp:63-68 AreaLight, 48 B (built by the library as renderer_pt.cpp:838-917) */
typedef struct pt_area_light {
  uint32_t instanceIdx;
  uint32_t indices[3];
  float area, power, cumulativePower;
  float _pad;
  pt_float3 emission;
} pt_area_light;

/* ---------------------------------------------------------------------------------------------------------- */
/* Renderer                                                                                                      */

typedef struct pt_renderer pt_renderer;

/* Renderer::Renderer(device, queue, store) (renderer_pt.hpp:28-32, renderer_pt.cpp:18-60): builds the
 * pipelines and loads the 8 GGX energy LUTs (renderer_pt.cpp:385-446). */
typedef struct pt_create_info {
  uint32_t abi_version;   /* PT_ABI_VERSION */
  int32_t device_ordinal; /* HIP device; the reference takes the MTL::Device of the window */
  const void* lut_blob;   /* the LUT blob (tools/make_lut_blob.py) in host memory, or NULL ... */
  uint64_t lut_blob_size;
  const char* lut_path;   /* ... to read it from this file (NULL: $PTAMD_LUT_PATH) */
  /* NEW (ABI 3): a DEVICE GROUP.  device_count >= 2 devices share every render: the samples [first_sample, first_sample + spp)
   * are dealt to them in contiguous ranges (whole GMoN buckets with PT_FLAG_GMON), each device renders its range on its own
   * host thread and stream, and the running means are merged when the image is asked for (pt_wait / pt_read_*): ONE RCCL
   * all-reduce of the float accumulator over xGMI (bucket means gathered to the first device and resolved there with GMoN).
   * The merged image lives on device_ordinals[0] (in external_accumulator when given).  A device may be listed more than
   * once (logical shards on one GPU).  device_count == 0: the single device `device_ordinal`. */
  const int32_t* device_ordinals;
  uint32_t device_count;
} pt_create_info;

int pt_create(const pt_create_info* info, pt_renderer** out);
/* Which GPU runtime objects this process holds and which of them serves this library.  No counterpart in the reference (Metal is a
 * system framework); needed here because PyTorch's ROCm wheel bundles private copies of libamdhip64 / libhsa-runtime64 / librccl
 * that can end up mapped BESIDE the system's, and then only the runtime that initialises first sees the GPU.
 * pt_create returns PT_ERR_RUNTIME_CONFLICT when hip_runtimes_mapped exceeds 1 (a second libhsa-runtime64 under one HIP runtime is what
 * rocprofv3's tool library maps: reported, not refused).  Touches no GPU. */
typedef struct pt_runtime_info {
  char hip_runtime_path[512]; /* the libamdhip64 this library's hip* calls resolve to */
  char hsa_runtime_path[512]; /* the (first) libhsa-runtime64 mapped */
  char rccl_path[512];        /* the librccl bound by a device group / pt_rccl_probe; "" until one of them has loaded it */
  uint32_t hip_runtimes_mapped, hsa_runtimes_mapped, rccl_mapped; /* distinct shared objects of each kind in the process */
  int32_t hip_runtime_version; /* hipRuntimeGetVersion of the serving runtime, 0 if the call failed */
  char all_mapped[2048];      /* every matching object, " + " separated: hip | hsa | rccl */
} pt_runtime_info;
int pt_get_runtime_info(pt_runtime_info* out);
/* How a device group deals the samples [0, spp) of a render to its `members` (pure host arithmetic, exported for tests):
 * contiguous ranges, equal up to one sample; with PT_FLAG_GMON whole buckets per member (bucket b = samples
 * [b * ceil(spp / buckets), ...), renderer_pt.cpp:124-126), bucket0/bucket1 = each member's bucket range (may be NULL). */
int pt_group_partition(uint32_t spp, uint32_t members, int32_t flags, uint32_t gmon_buckets, uint64_t* first, uint64_t* count,
                       uint32_t* bucket0, uint32_t* bucket1);
/* Loads librccl.so the way a device group over >= 2 distinct GPUs does (dlopen) and binds the entry points the merge uses
 * (ncclCommInitAll, ncclCommDestroy, ncclAllReduce, ncclGroupStart, ncclGroupEnd, ncclGetErrorString).  No GPU is touched:
 * a build-box check that the multi-GPU path can find its collective library.  PT_OK or PT_ERR_UNSUPPORTED (pt_last_error). */
int pt_rccl_probe(void);
/* The same on a GPU, one step further: a communicator of ONE rank on `device_ordinal`, the merge's grouped in-place
 * ncclAllReduce(sum, float32) on a known pattern, result checked, communicator destroyed — every RCCL call the distinct-device merge
 * makes, with its argument types and stream ordering, as far as a one-GPU box can run them. */
int pt_rccl_selftest(int32_t device_ordinal);
/* Renderer::~Renderer (renderer_pt.hpp:34) */
void pt_destroy(pt_renderer* r);

/* Parameters of Renderer::startRender(camera, size, spp, gmonBuckets, workingSpace, flags)
 * (renderer_pt.hpp:38-45) + selectKernel (:47-53) + what the reference fixes at compile time or lacks. */
typedef struct pt_render_params {
  uint32_t width, height;      /* viewport size */
  uint32_t spp;                /* samples this renderer accumulates (m_accumulationFrames) */
  uint32_t gmon_buckets;       /* used only with PT_FLAG_GMON (1..32, gmon.metal:12); constants.gmonBuckets is 1 otherwise */
  int32_t flags;               /* PT_FLAG_* */
  uint32_t integrator;         /* PT_INTEGRATOR_* (default MIS, renderer_pt.hpp:98) */
  pt_colorspace working_space; /* default BT2020 (pt_viewport.hpp:95) */
  uint32_t max_bounces;        /* NEW: kernel.metal:5 hard-codes 50; 1..50 */
  uint32_t first_sample;       /* NEW: frameIdx of this renderer's first sample (multi-GPU shards) */
  uint32_t samples_in_flight;  /* NEW: samples traced concurrently per batch; 0 = auto */
  uint32_t nonfinite_policy;   /* NEW: PT_NONFINITE_*: what a NaN/inf sample does to the running mean */
  void* external_accumulator;  /* optional DEVICE pointer to W*H float4; NULL = library-owned */
  void* stream;                /* optional hipStream_t to enqueue on; NULL = library-owned stream */
  uint32_t accel_structure;    /* NEW (ABI 3): PT_ACCEL_*.  The reference always builds BLAS per mesh + TLAS over instances
                                  (renderer_pt.cpp:653-749); closest hits are identical whichever structure is walked */
  uint32_t _reserved;
} pt_render_params;
/* PT_ACCEL_AUTO: one BVH over the flattened world-space triangles (fastest: C3 7.4 vs 4.2 Grays/s) unless that would not fit
 * beside the path queues; PT_ACCEL_TWO_LEVEL: TLAS over instances + one object-space BLAS per mesh (C3: 65 KB instead of 82 MB
 * of nodes, staged in LDS by the trace kernels); needs invertible instance transforms, else one BVH is built. */
enum { PT_ACCEL_AUTO = 0, PT_ACCEL_ONE_BVH = 1, PT_ACCEL_TWO_LEVEL = 2 };

/* How pt_start_render sizes the wavefront queues for an image (pure host arithmetic, exported for tests): the samples traced
 * concurrently per batch after every index-width limit has been applied (an explicit samples_in_flight is HALVED until the
 * 16-bit segment slots, the 32-bit queue indices and the chunk tables can address the batch; the image does not change),
 * and the segment geometry.  free_hbm_bytes only matters for samples_in_flight = 0 (auto); tiles_per_seg_override /
 * seg_bands are the $PTAMD_TILES_PER_SEG / $PTAMD_SEG_BANDS tuning knobs (0 / 4 by default). */
typedef struct pt_queue_plan {
  uint32_t samples_in_flight;  /* what a batch will carry */
  uint32_t tiles_per_seg;      /* 8x8 pixel tiles per queue segment */
  uint32_t nseg;               /* segments (<= 32768) */
  uint32_t seg_cap;            /* path slots per segment = tiles_per_seg * samples_in_flight * 64 (<= 65536) */
  uint64_t capacity;           /* path slots per queue array */
  uint64_t lbuf_entries;       /* entries of the per-sample radiance buffer */
} pt_queue_plan;
int pt_plan_queues(uint32_t width, uint32_t height, uint32_t spp, uint32_t samples_in_flight, uint64_t free_hbm_bytes,
                   uint32_t tiles_per_seg_override, uint32_t seg_bands, pt_queue_plan* out);

/* Renderer::startRender + the rebuild* half of the first Renderer::render() (renderer_pt.cpp:72-111,
 * 199-217): copies the snapshot to HBM, builds light table, constants and the LBVH.  The caller owns the
 * snapshot memory only until this returns. Resets progress to 0. */
int pt_start_render(pt_renderer* r, const pt_scene_snapshot* scene, const pt_render_params* params);

/* Renderer::render() steady state (renderer_pt.cpp:113-197): accept up to `max_spp_this_call` further samples (the reference
 * encodes exactly 1) and return without waiting. 0 = all remaining samples.  Progress counts accepted samples, as the reference's
 * m_accumulatedFrames counts encoded ones.  Calls that arrive while the GPU is still executing the previous batch are merged into
 * one batch of up to samples_in_flight samples (a one-sample batch cannot fill the chip); pt_wait, the pt_read_* / present entry
 * points and the call that accepts the render's last sample enqueue whatever is pending.  The image is the same either way. */
int pt_render_step(pt_renderer* r, uint32_t max_spp_this_call);
/* Block until everything enqueued so far has completed (the reference only blocks in readback). */
int pt_wait(pt_renderer* r);

/* Renderer::status() (renderer_pt.cpp:1023-1031), renderProgress() (:1033-1035), renderTime() (:1037) */
int pt_status(const pt_renderer* r);
int pt_progress(const pt_renderer* r, uint64_t* accumulated, uint64_t* total);
uint64_t pt_render_time_ms(const pt_renderer* r);

/* Renderer::gmonOptions() (renderer_pt.hpp:71; pt_shader_defs.hpp:164-166 GmonOptions). With PT_FLAG_GMON the samples
 * are accumulated into `gmon_buckets` bucket images (bucket = sample / ceil(spp / buckets), renderer_pt.cpp:124-139)
 * and the accumulator holds their Gini-weighted median-of-means (shaders/gmon.metal), recomputed after every batch. */
typedef struct pt_gmon_options { float cap; } pt_gmon_options; /* default 1.0 */
int pt_set_gmon_options(pt_renderer* r, const pt_gmon_options* options);
/* One bucket image (W*H*4 floats), for parity checks. */
int pt_read_gmon_bucket(pt_renderer* r, uint32_t bucket, float* rgba_out);

/* ---- post-process chain + tonemap -> RGBA8 ---------------------------------------------------------
 * Renderer::postProcessOptions() / tonemapOptions() / outputColorspace() (renderer_pt.hpp:65-73) edit option structs the
 * UI writes every frame (core/postprocessing.hpp:168-218); pass order exposure, chromaticAberration, contrastSaturation,
 * toneCurve, vignette, tonemap (renderer_pt.cpp:343-353).  Field names follow the reference's structs. */
typedef struct pt_post_options {
  float exposure;                                          /* ExposureOptions */
  float ca_amount, ca_green_shift;                         /* ChromaticAberrationOptions (0, 70) */
  float contrast, saturation;                              /* ContrastSaturationOptions */
  float blacks, shadows, highlights, whites;               /* ToneCurveOptions */
  float vig_amount, vig_midpoint, vig_feather, vig_power, vig_roundness; /* VignetteOptions (0, 0, 50, 20, 100) */
} pt_post_options;

enum { PT_TONEMAP_NONE = 0, PT_TONEMAP_AGX = 1, PT_TONEMAP_KHRONOS_PBR = 2, PT_TONEMAP_FLIM = 3 }; /* postprocess::Tonemapper */

typedef struct pt_tonemap_options {
  uint32_t tonemapper;                                     /* default AgX (postprocessing.hpp:219) */
  float agx_offset[3], agx_slope[3], agx_power[3], agx_saturation;      /* agx::Look (looks::none) */
  float khr_compression_start, khr_desaturation;           /* khronos_pbr::Options (0.8, 0.15) */
  float flim_pre_exposure, flim_pre_formation_filter[3], flim_pre_formation_filter_strength; /* flim::Options */
  float flim_extended_gamut_scale[3], flim_extended_gamut_rotation[3], flim_extended_gamut_mul[3];
  float flim_sigmoid_log2_min, flim_sigmoid_log2_max, flim_sigmoid_toe[2], flim_sigmoid_shoulder[2];
  float flim_negative_exposure, flim_negative_density, flim_print_backlight[3], flim_print_exposure, flim_print_density;
  float flim_black_point;
  uint32_t flim_auto_black_point;
  float flim_post_formation_filter[3], flim_post_formation_filter_strength, flim_midtone_saturation;
  float shadow_color[3], midtone_color[3], highlight_color[3];          /* LiftGammaGain (0.5 each) */
  float shadow_offset, midtone_offset, highlight_offset;
  pt_colorspace output_space;                              /* outputColorspace(), default Display P3 (renderer_pt.hpp:182) */
} pt_tonemap_options;

/* Fill with the reference's defaults (postprocessing.hpp:168-226, flim::presets::flim). */
void pt_default_post_options(pt_post_options* o);
void pt_default_tonemap_options(pt_tonemap_options* o);
int pt_set_post_options(pt_renderer* r, const pt_post_options* o);
int pt_set_tonemap_options(pt_renderer* r, const pt_tonemap_options* o);
/* readbackRenderTarget() (renderer_pt.hpp:57, renderer_pt.cpp:1039-1059): the post-processed, tonemapped RGBA8 image
 * (W*H*4 bytes, row-major, top-left origin). Blocks. */
int pt_read_render_target(pt_renderer* r, uint8_t* rgba8_out);
/* presentRenderTarget() (renderer_pt.hpp:55, used by pt_viewport.cpp:711 to blit): post-processes the current accumulator into
 * the library's RGBA8 render target ON THE DEVICE and returns its device address (W*H*4 bytes, valid until the next
 * pt_start_render / pt_destroy) without a host copy.  The work is enqueued on the renderer's stream; *stream_out (may be NULL)
 * receives that hipStream_t so the caller can order its blit after it.  A device group presents on device_ordinals[0]. */
int pt_present_render_target(pt_renderer* r, void** device_rgba8_out, void** stream_out);

/* The float accumulator: W*H RGBA32F, row-major, top-left origin, running mean, alpha 1
 * (renderer_pt.cpp:812-821, kernel.metal:672-684).  Blocks like readbackRenderTarget (:1039-1059). */
int pt_read_accumulator(pt_renderer* r, float* rgba_out);
/* Device address of the accumulator (for an RCCL reduce by the caller); NULL before pt_start_render. */
void* pt_accumulator_device_ptr(pt_renderer* r);

const char* pt_last_error(void);

/* ---------------------------------------------------------------------------------------------------------- */
/* Parity / measurement surface (no reference counterpart; used by tests and bench.py)                          */

int pt_get_constants(const pt_renderer* r, pt_constants* out);
/* The environment alias table in use (given or built): copies up to `capacity` entries; *count = width*height or 0. */
int pt_get_env_alias(const pt_renderer* r, pt_alias_entry* out, uint64_t capacity, uint64_t* count);
/* Copies up to `capacity` lights; returns the light count in *count. */
int pt_get_lights(const pt_renderer* r, pt_area_light* out, uint32_t capacity, uint32_t* count);

/* Closest-hit record of the camera ray of every pixel for sample `sample_idx` (raygen + traversal only). */
typedef struct pt_hit_record {
  float t, u, v;
  int32_t instance; /* -1 = miss */
  int32_t primitive;
} pt_hit_record;
int pt_trace_primary(pt_renderer* r, uint32_t sample_idx, pt_hit_record* out /* W*H */);

/* Trace ONE sample without touching the accumulator; returns its radiance and the (instance, primitive)
 * hit at every bounce of every pixel's path (-1,-1 where the path was already dead or missed).
 *   radiance_out : W*H*4 floats (rgb, 1)               or NULL
 *   hits_out     : max_bounces * W*H * 2 int32          or NULL */
int pt_debug_sample(pt_renderer* r, uint32_t sample_idx, float* radiance_out, int32_t* hits_out);

typedef struct pt_stats {
  uint64_t triangles;          /* flattened world-space triangles */
  uint64_t bvh_nodes;
  uint32_t bvh_max_depth;      /* levels of the tree (two-level: TLAS + deepest BLAS); the traversal stack holds <= 5 entries per level of the
                                  6-wide form the one-BVH structure is built in by default, <= 3 of the 4-wide form ($PTAMD_BVH4, the fallback builders) */
  uint32_t samples_in_flight;
  double upload_ms;            /* snapshot -> HBM */
  double bvh_build_ms;         /* LBVH build (device time) */
  uint64_t closest_rays;       /* rays traced since pt_start_render */
  uint64_t shadow_rays;
  uint64_t shaded_hits;
  uint64_t paths;              /* pixel*samples started */
  uint64_t nonfinite_samples;  /* samples whose radiance was NaN/inf (zeroed under PT_NONFINITE_ZERO) */
  /* device time per kernel class since pt_start_render, HIP events on the launch stream */
  double ms_raygen, ms_closest, ms_shade, ms_shadow, ms_accumulate;
  uint64_t launches_closest, launches_shadow;
  /* instrumented traversal (pt_measure_traversal): mean BVH nodes / triangles fetched per ray */
  double nodes_per_closest_ray, tris_per_closest_ray;
  double nodes_per_shadow_ray, tris_per_shadow_ray;
  uint32_t accel_two_level;    /* 1: the two-level structure is in use */
  uint32_t batches;            /* batches enqueued since pt_start_render (pt_render_step calls that arrive while the GPU is busy are merged) */
  uint64_t leaf_slots;         /* 64-byte leaf slots of the acceleration structure: a slot holds one triangle, or two of one instance that share an
                                  edge (one-BVH structure); tris_per_*_ray count triangle TESTS, a slot fetch serves one or two of them */
} pt_stats;
int pt_get_stats(pt_renderer* r, pt_stats* out);
/* Enable per-kernel HIP-event timing (adds two event records per launch). */
int pt_set_profiling(pt_renderer* r, int enabled);
/* Run one instrumented sample (outside any timed region) to count node/triangle fetches per ray. */
int pt_measure_traversal(pt_renderer* r, uint32_t sample_idx);

#ifdef __cplusplus
}
#endif
#endif /* PTAMD_H */
