// rt_host.hpp — headless C++ mirror of the reference's Rust host for the compute pass.
//
// The reference host is Rust (winit + wgpu); no Rust toolchain exists in this image, so the host
// side above the C ABI is restated in C++ with the reference's type and method names:
//   Time      <- src/common/time.rs:7-78
//   Camera    <- src/resources/camera.rs:23-131,252-288   (state, CameraRaw, matrices, frame counter)
//   World     <- src/world.rs:21-122, src/resources/objects.rs:26-50, src/resources/material.rs:22-38
//   Raytracer <- src/raytracing.rs:6-201                  (RAII over rt_ctx)
//   App       <- src/app.rs:26-48,93-116                  (update / render ordering)
// Windowing, input, presentation and the wgpu device (lib.rs, screen.rs, render_env.rs) are out of
// scope: this host renders headless.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rt {

// ---- glam 0.24 equivalents used by camera.rs (column-major 4x4, f32) -------------------------
struct Vec3 { float x, y, z; };
struct Mat4 {
  float c[16];  // column-major: column j at c[4j..4j+3]
  static Mat4 from_cols_array(const float (&a)[16]);
  static Mat4 look_to_rh(Vec3 eye, Vec3 dir, Vec3 up);
  Mat4 operator*(const Mat4& rhs) const;
  Mat4 inverse() const;
  Vec3 transform_point3(Vec3 p) const;  // (m * (p, 1)).xyz, no perspective divide (glam Mat4::transform_point3)
};

// ---- deterministic stand-in for rand::thread_rng (the reference's materials are unseeded and
// therefore irreproducible) ---------------------------------------------------
struct SeededRng {
  uint64_t s;
  explicit SeededRng(uint64_t seed) : s(seed) {}
  uint64_t next_u64();
  float gen_f32();                    // uniform [0,1), 24 bits, like rand's Standard for f32
  uint32_t gen_range(uint32_t n);     // uniform 0..n
};

// ---- Time (time.rs) ------------------------------------------------------------------------------
class Time {
 public:
  Time() : time_(0.0f) {}
  void add_delta(float delta) { time_ += delta; }  // time.rs:72-77
  float time() const { return time_; }
  uint32_t bits() const;
  void set_bits(uint32_t b);  // checkpoint restore
 private:
  float time_;
};

// ---- Camera (camera.rs) ---------------------------------------------------------------------------
class Camera {
 public:
  Camera(uint32_t width, uint32_t height);  // Camera::new defaults (camera.rs:59-72)
  rt_camera_raw to_raw() const;             // camera.rs:77-87
  Mat4 calculate_world_to_pixel() const;    // camera.rs:118-131
  void increase_frame() { current_frame += 1; }               // camera.rs:252-265 (progress bar omitted)
  void reset_render() { current_frame = 0; }                  // camera.rs:274-277
  bool render_finished() const { return current_frame >= frames_to_render; }  // camera.rs:279-281
  std::string to_string() const;            // camera.rs:267-272
  // headless forms of the interactive controls (no winit): each returns true when the scene changed,
  // in which case the caller does what scene_was_updated does (reset_render + upload, camera.rs:283-288)
  enum class Key { K, J, I, U, ArrowUp, ArrowDown, Backspace, W, S, D, A, Space, ShiftLeft };
  // camera.rs:133-175 with keys_held = {key}: the reference applies the pressed key's setting change and then
  // one translation per HELD movement key (all through the same inverse rotation, which does not depend on
  // pos), so n keys held together equal n calls here, one per key, in W S D A Space ShiftLeft order. A key
  // outside this enum changes nothing there and cannot be named here.
  bool key_press(Key key);
  void zoom(float val);                     // camera.rs:193-197
  void look_at_pixel_from_center(float off_x, float off_y);  // camera.rs:200-206
  float speed = 0.05f;

  Vec3 pos{0.f, 0.f, 0.f};
  float z_near = 1.0f, z_far = 100000.0f, fov = 0.25f;
  float res_x, res_y;
  uint32_t samples_per_pixel = 1, frames_to_render = 8, current_frame = 0;
  Vec3 look_direction{0.f, 0.f, -1.f};
 private:
  Mat4 clip_space_to_pixel() const;  // camera.rs:89-98
  Mat4 perspective_proj() const;     // camera.rs:100-112
  Mat4 rotation_matrix() const;      // camera.rs:114-116
};

// ---- World (world.rs, objects.rs, material.rs) ------------------------------------------------------
struct Sphere {
  static rt_sphere make(Vec3 center, float radius, uint32_t material_index);                    // objects.rs:42-49
  static std::vector<rt_sphere> random_bunch(uint32_t amount, uint32_t materials_len, SeededRng& rng);  // objects.rs:27-40
};
struct Material {
  static rt_material random_new(SeededRng& rng);  // material.rs:22-37
};
class World {
 public:
  // World::new (world.rs:31-40): 3 random materials, the 4 default spheres.
  static World default_scene(uint64_t seed);
  // the commented-out alternative at world.rs:42: random_bunch(amount, materials.len())
  static World random_bunch(uint32_t amount, uint32_t n_materials, uint64_t seed);
  std::vector<rt_sphere> spheres;
  std::vector<rt_material> materials;
};

// ---- Raytracer (raytracing.rs) --------------------------------------------------------------------
class Raytracer {
 public:
  Raytracer(int device, uint32_t width, uint32_t height, uint32_t flags);  // Raytracer::new
  ~Raytracer();
  Raytracer(const Raytracer&) = delete;
  Raytracer& operator=(const Raytracer&) = delete;
  // Raytracer::compute(encoder, world_bind_group, time_bind_group): the bind groups become the
  // explicit uploads below.
  void bind_world(const World& w);
  void write_camera(const rt_camera_raw& c);
  void write_time(uint32_t bits);
  void compute();
  rt_ctx* ctx() { return ctx_; }
 private:
  rt_ctx* ctx_ = nullptr;
};

// ---- App (app.rs) ----------------------------------------------------------------------------------
class App {
 public:
  App(int device, uint32_t width, uint32_t height, uint32_t flags, World world);
  void update();                   // app.rs:93-97
  void render();                   // app.rs:99-116 (compute pass only)
  void run(uint32_t redraws);      // `redraws` x (update, render), batched into one launch
  // Checkpoint / resume (no counterpart in the reference, whose render restarts from frame 0):
  // the accumulation image, the frame counter, the time and the camera, with a hash of the scene so a
  // checkpoint cannot be resumed against other geometry. A resumed render continues bit-identically.
  void save_checkpoint(const std::string& path);
  void load_checkpoint(const std::string& path);
  uint32_t width() const { return width_; }
  uint32_t height() const { return height_; }
  uint32_t flags() const { return flags_; }
  Time time;
  Camera camera;
  World world;
  Raytracer ray_tracer;
 private:
  uint32_t width_, height_, flags_;
};

// What a checkpoint file says about itself (readable without a GPU).
struct CheckpointInfo {
  uint32_t width, height, flags, current_frame, frames_to_render, samples_per_pixel, time_bits, n_spheres, n_materials;
  uint64_t scene_hash, image_bytes;
};
CheckpointInfo read_checkpoint_info(const std::string& path);  // throws Error{RT_ERR_INVALID} on a malformed file

// thrown by Raytracer/App on a non-zero ABI status (the reference panics via unwrap/expect)
struct Error {
  int code;
  std::string what;
};

// RGBA8 -> PNG (stored deflate, no dependencies). present = true applies what the reference's
// screen pass shows: pow(c, 2.2) (screen_shader.wgsl:36-39) followed by the sRGB surface encode.
bool write_png(const std::string& path, const uint8_t* rgba8, uint32_t width, uint32_t height, bool present);

}  // namespace rt
