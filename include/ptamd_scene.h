/* ptamd_scene.h — scene ingestion for libptamd: reads what the reference's own loaders read and
 * produces the `pt_scene_snapshot` that `pt_start_render` consumes, so that real Platinum scenes can feed the
 * MI355X backend without the Mac frontend.  Host-only code (no GPU needed); lives in libptamd.so.
 *
 *   pt_scene_load_json     Scene::Scene(path, device)            /root/reference/src/core/scene.cpp:30-84, 789-902
 *                          (`<stem>.json` + `<stem>_data.bin`, the format written by Scene::saveToFile :536-631)
 *   pt_scene_save_json     Scene::saveToFile                     core/scene.cpp:536-631, 633-787; utils/json.hpp
 *   pt_scene_import_gltf   loaders::gltf::GltfLoader::load       loaders/gltf.cpp:28-113 (meshes :115-248, nodes :253-293,
 *                          materials :304-394, textures :399-420; loaders/texture.cpp:30-48,113-218; tangents
 *                          core/mesh.cpp:135-157 = MikkTSpace on the indexed vertices)
 *   pt_scene_set_environment  Environment::setTexture + rebuildAliasTable is done inside pt_start_render
 *   pt_scene_build_snapshot   what Renderer::startRender gathers: rebuildResourceBuffers (renderer_pt.cpp:448-651: mesh and
 *                          texture index assignment in asset order, per-instance MaterialGPU arrays), Scene::getInstances
 *                          (core/scene.cpp:476-534: visible-filtered LIFO traversal, world = parent * local), camera node
 *                          world transform (core/scene.cpp:463-474)
 * All functions return PT_OK or a PT_ERR_* code (ptamd.h); pt_scene_last_error() has the text. */
#ifndef PTAMD_SCENE_H
#define PTAMD_SCENE_H

#include "ptamd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_scene pt_scene; /* assets (textures, meshes, materials) + node hierarchy: core/scene.hpp without the ECS */

/* loaders/gltf.hpp:20-25 LoadOptions */
enum { PT_GLTF_NONE = 0, PT_GLTF_SKIP_EMPTY_NODES = 1 << 0, PT_GLTF_CREATE_SCENE_NODES = 1 << 1 };

/* MTL::PixelFormat raw values stored in scene.json ("format", core/scene.cpp:733) */
enum { PT_MTL_R8UNORM = 10, PT_MTL_RG8UNORM = 30, PT_MTL_RGBA8UNORM = 70, PT_MTL_RGBA8UNORM_SRGB = 71, PT_MTL_RGBA32FLOAT = 125 };

int pt_scene_create(pt_scene** out);                          /* empty scene with a root node "Scene" (core/scene.cpp:22-28) */
void pt_scene_destroy(pt_scene* s);
int pt_scene_load_json(const char* json_path, pt_scene** out);
int pt_scene_save_json(const pt_scene* s, const char* json_path);
int pt_scene_import_gltf(pt_scene* s, const char* gltf_or_glb_path, int options);
/* Adds an RGBA32F lat-long image as a texture asset and makes it the scene environment (what the frontend's
 * "load environment" does: TextureLoader HDR path + Environment::setTexture). */
int pt_scene_set_environment(pt_scene* s, const float* rgba, uint32_t width, uint32_t height, const char* name);
/* The same from a file, as TextureLoader::loadFromFile(path, name, TextureType::HDR) (loaders/texture.cpp:86-103): ".exr" is read like
 * tinyexr's LoadEXR (scanline NONE / RLE / ZIPS / ZIP / PIZ, HALF / FLOAT), anything else like stbi_loadf (Radiance .hdr). */
int pt_scene_load_environment(pt_scene* s, const char* path);

typedef struct pt_scene_counts {
  uint32_t nodes, meshes, textures, materials, cameras, instances; /* instances: visible nodes with a mesh */
  uint64_t triangles;                                               /* over instances */
} pt_scene_counts;
int pt_scene_get_counts(const pt_scene* s, pt_scene_counts* out);
/* Camera i in Scene::getCameras() order (core/scene.cpp:496-512). name may be NULL. */
int pt_scene_get_camera(const pt_scene* s, uint32_t i, uint64_t* node_id, char* name, uint32_t name_capacity);
/* Adds a camera node under the root (what SceneExplorer's "Camera" menu does, frontend/windows/scene_explorer.cpp:84-90):
 * Camera::withFocalLength(focal_mm) with a tracking transform. Returns its node id. */
int pt_scene_add_camera(pt_scene* s, const char* name, const float position[3], const float target[3], float focal_length_mm,
                        uint64_t* node_id);

/* Builds (or rebuilds) the flat snapshot for rendering through `camera_node`.  The returned pointer and every array it
 * references are owned by the scene and stay valid until the next build/import/destroy. */
int pt_scene_build_snapshot(pt_scene* s, uint64_t camera_node, const pt_scene_snapshot** out);

/* MikkTSpace-compatible tangents for an indexed triangle mesh, written per face-vertex in face order onto the shared
 * vertices exactly as core/mesh.cpp:49-57,135-157 does (exposed for parity tests against the reference's deps/mikkt). */
int pt_generate_tangents(const pt_float3* positions, pt_vertex_data* vertex_data, uint32_t vertex_count, const uint32_t* indices,
                         uint32_t triangle_count);

/* An 8-bit image file in memory -> RGBA8, as stbi_load_from_memory(data, len, &w, &h, nullptr, 4) gives the texture loader
 * (loaders/texture.cpp:111-119): PNG (all colour types, 1-16 bit, tRNS; no Adam7) and JPEG (baseline + progressive, 1 / 3 / 4
 * components, any sampling factors, restart intervals; stb_image's IDCT / upsampling / colour arithmetic, bit-identical to it).
 * Writes width * height * 4 bytes when `rgba_out` is non-NULL and `capacity` suffices; always returns the size. */
int pt_decode_image_rgba8(const uint8_t* data, uint64_t len, uint32_t* width, uint32_t* height, uint8_t* rgba_out, uint64_t capacity);

const char* pt_scene_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
