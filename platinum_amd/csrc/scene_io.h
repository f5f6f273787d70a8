// scene_io.h — internal types of the scene-ingestion component (scene_io.cpp, scene_gltf.cpp).  Public interface:
// include/ptamd_scene.h.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ptamd_scene.h"

namespace ptio {

constexpr uint64_t kNoId = ~0ull;      // std::nullopt of std::optional<AssetID>
constexpr size_t kNoNode = ~(size_t)0;  // entt::null parent

extern thread_local std::string g_error;

struct JV {  // a parsed JSON value
  enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  bool b = false;
  double d = 0;
  uint64_t u = 0;
  bool is_uint = false;
  std::string s;
  std::vector<JV> a;
  std::vector<std::pair<std::string, JV>> o;
  const JV* find(const std::string& k) const;
  const JV& at(const std::string& k) const;
  const JV& at(size_t i) const;
  double num() const;
  uint64_t u64() const;
  bool boolean() const;
  const std::string& str() const;
};
JV json_parse(const std::string& text);
void json_dump(const JV& v, std::string& out);

struct Mat4 { float c[4][4]; };  // [column][row], like simd::float4x4
Mat4 mat_identity();
Mat4 mat_mul(const Mat4& a, const Mat4& b);

struct Transform {  // core/transform.hpp:19-51
  float translation[3] = {0, 0, 0}, rotation[3] = {0, 0, 0}, scale[3] = {1, 1, 1}, target[3] = {0, 0, 0};
  bool track = false;
  Mat4 matrix() const;
};
struct Camera {  // core/camera.hpp:10-18
  float sensor_size[2] = {36.0f, 24.0f};
  float focal_length = 50.0f, aperture = 0.0f;
  uint32_t aperture_blades = 7;
  float roundness = 1.0f, bokeh_power = 0.0f, focus_distance = 1.0f;
};
struct Material {  // core/material.hpp:15-49
  std::string name;
  float base_color[4] = {0.8f, 0.8f, 0.8f, 1.0f};
  float emission[3] = {0, 0, 0};
  float emission_strength = 0.0f, roughness = 1.0f, metallic = 0.0f, transmission = 0.0f, ior = 1.5f;
  float anisotropy = 0.0f, anisotropy_rotation = 0.0f, clearcoat = 0.0f, clearcoat_roughness = 0.05f;
  bool thin_transmission = false;
  std::vector<std::pair<int, uint64_t>> textures;  // TextureSlot -> texture asset id, insertion order (ankerl map)
  uint64_t get_texture(int slot) const { for (auto& t : textures) if (t.first == slot) return t.second; return kNoId; }
  void set_texture(int slot, uint64_t id) { for (auto& t : textures) if (t.first == slot) { t.second = id; return; } textures.emplace_back(slot, id); }
};
struct Texture { std::string name; bool alpha = false; uint32_t width = 0, height = 0, mtl_format = 0; std::vector<uint8_t> bytes; };
struct Mesh { std::vector<pt_float3> positions; std::vector<pt_vertex_data> vdata; std::vector<uint32_t> indices, slots; };
struct Asset {
  enum Type { TEXTURE, MESH, MATERIAL } type = MATERIAL;
  uint64_t id = 0;
  bool retain = true;
  uint32_t rc = 0;
  Texture tex; Mesh mesh; Material mat;
};
struct Node {
  uint64_t id = 0;
  std::string name;
  bool visible = true;
  Transform transform;
  bool has_mesh = false; uint64_t mesh = 0; std::vector<uint64_t> materials;  // kNoId = "default"
  bool has_camera = false; Camera camera;
  size_t parent = kNoNode;
  std::vector<size_t> children;
};
struct Snapshot {
  std::vector<pt_mesh> meshes;
  std::vector<pt_texture> textures;
  std::vector<pt_instance> instances;
  std::vector<std::vector<pt_material_gpu>> material_storage;
  std::vector<pt_instance_materials> instance_materials;
  pt_scene_snapshot snapshot{};
};
struct Scene {
  std::vector<Asset> assets;  // insertion order = the iteration order of the reference's asset map
  uint64_t next_asset_id = 0;
  std::vector<Node> nodes;
  size_t root_index = 0;
  uint64_t next_node_id = 0;
  bool has_env = false; uint64_t env_texture = 0; std::vector<pt_alias_entry> env_alias;
  Snapshot snap;

  Scene();
  Asset* find_asset(uint64_t id);
  const Asset* find_asset(uint64_t id) const;
  uint64_t create_asset(Asset&& a, bool retain);
  size_t create_node(const std::string& name, size_t parent, uint64_t id);
  void retain(uint64_t id);
  void set_mesh(size_t node, uint64_t mesh_id);
  void set_material(size_t node, size_t idx, uint64_t id);
  const pt_scene_snapshot* build_snapshot(uint64_t camera_node);
  void counts(pt_scene_counts* out) const;
  std::vector<size_t> cameras() const;
};

size_t mtl_bytes_per_pixel(uint32_t mtl_format);
std::unique_ptr<Scene> load_json(const std::string& path);
void save_json(const Scene& s, const std::string& path);
void import_gltf(Scene& s, const std::string& path, int options);   // scene_gltf.cpp
void generate_tangents(const pt_float3* positions, pt_vertex_data* vdata, uint32_t vertex_count, const uint32_t* indices,
                       uint32_t triangle_count);                     // scene_gltf.cpp
std::vector<float> read_exr_rgba(const std::string& path, uint32_t* w, uint32_t* h);            // scene_image.cpp
std::vector<float> read_radiance_hdr_rgba(const std::string& path, uint32_t* w, uint32_t* h);   // scene_image.cpp
std::vector<uint8_t> decode_png_rgba8(const uint8_t* data, size_t len, uint32_t* w, uint32_t* h);  // scene_gltf.cpp
bool is_jpeg(const uint8_t* data, size_t len);                                                      // scene_jpeg.cpp
std::vector<uint8_t> decode_jpeg_rgba8(const uint8_t* data, size_t len, uint32_t* w, uint32_t* h); // scene_jpeg.cpp

}  // namespace ptio
