// TEST INFRASTRUCTURE — C API of the CPU oracle (oracle/pt_oracle.cpp). See the header comment there.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
#pragma once
#include <stdint.h>
#include "../include/ptamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_stats {
  uint64_t triangles, bvh_nodes;
  uint64_t closest_rays, shadow_rays, shaded_hits, paths;
  uint64_t nodes_closest, tris_closest, nodes_shadow, tris_shadow;
  uint64_t nonfinite;
} orc_stats;

// use_bvh = 0: brute force over every triangle (the definition); 1: oracle-private median-split BVH (same answers).
orc_scene* orc_scene_create(const pt_scene_snapshot* snap, const pt_render_params* params, const void* lut_blob,
                            uint64_t lut_size, int use_bvh);
void orc_scene_destroy(orc_scene* sc);
int orc_get_constants(const orc_scene* sc, pt_constants* out);
int orc_get_env_alias(const orc_scene* sc, pt_alias_entry* out, uint64_t capacity, uint64_t* count);
int orc_get_lights(const orc_scene* sc, pt_area_light* out, uint32_t capacity, uint32_t* count);
int orc_render(orc_scene* sc, uint32_t first_sample, uint32_t nsamples, float* acc, uint32_t acc_n0, int threads,
               int count_traversal);
int orc_gmon_resolve(const orc_scene* sc, const float* buckets, uint32_t nBuckets, float cap, float* out);
int orc_postprocess(const orc_scene* sc, const float* acc, const pt_post_options* po, const pt_tonemap_options* to, uint8_t* rgba8_out,
                    float* float_out);
int orc_trace_primary(orc_scene* sc, uint32_t sample_idx, pt_hit_record* out);
int orc_debug_sample(orc_scene* sc, uint32_t sample_idx, float* radiance_out, int32_t* hits_out, int threads);
int orc_debug_pixel(orc_scene* sc, uint32_t x, uint32_t y, uint32_t sample_idx, float* L_out);
int orc_render_pixels(orc_scene* sc, const uint32_t* xy, uint32_t npixels, uint32_t first_sample, uint32_t nsamples, float* out,
                      uint32_t acc_n0, int threads);
int orc_get_stats(const orc_scene* sc, orc_stats* out);

uint32_t orc_halton_offset(uint32_t x, uint32_t y, uint32_t sample);
void orc_pcg4d(const uint32_t in[4], uint32_t out[4]);
float orc_halton(uint32_t i, uint32_t d);
uint32_t orc_prime(uint32_t d);
float orc_fresnel(float cosTheta, float ior);
float orc_avg_dielectric_fresnel_fit(float ior);
void orc_sincos(float x, float* s, float* c);
float orc_log2(float x);
float orc_exp2(float x);
void orc_sample_cosine_hemisphere(float u0, float u1, float out[3]);
void orc_sample_tri_uniform(float u0, float u1, float out[2]);
float orc_atan2(float y, float x);
float orc_acos(float x);
void orc_tex_sample(const orc_scene* sc, int texture, float u, float v, float out[4]);
void orc_ray_dir_to_uv(const float dir[3], float out[2]);
void orc_uv_to_ray_dir(const float uv[2], float out[3]);
float orc_lut_sample(const orc_scene* sc, int which, float cx, float cy, float cz);
// LUT generator restatement (ms_lut_gen.metal:337-743): one texel of table `which` re-integrated with the oracle's BSDF pieces
double orc_lut_regen(const orc_scene* sc, int which, float cosTheta, float roughness, float ior, uint32_t nsamples, uint32_t offset,
                     int lambda_mode);
double orc_lut_regen_texel(const orc_scene* sc, int which, int x, int y, int z, uint32_t nsamples, uint32_t seed, int lambda_mode);
float orc_lut_texel(const orc_scene* sc, int which, int x, int y, int z);
void orc_bsdf_sample(const orc_scene* sc, const pt_material_gpu* mat, const float wo[3], const float r[4], const float rc[2],
                     float out_sample[11]);
void orc_bsdf_eval(const orc_scene* sc, const pt_material_gpu* mat, const float wo[3], const float wi[3], float out_eval[4]);

#ifdef __cplusplus
}
#endif
