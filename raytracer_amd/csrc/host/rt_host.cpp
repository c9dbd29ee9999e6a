// rt_host.cpp — implementation of the headless host mirror (rt_host.hpp) and its C ABI (rt_host.h).
#include "rt_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "rt_host.h"

namespace rt {

// ---------------------------------------------------------------------------------------------------
// glam equivalents. Mat4::inverse is the cofactor scheme glam ports from GLM; glam's exact
// rounding (SSE2 vs scalar paths) is not reproducible without its source, which is why CameraRaw
// is an INPUT of the kernel ABI: kernel parity never depends on these 60 lines.
// ---------------------------------------------------------------------------------------------------
Mat4 Mat4::from_cols_array(const float (&a)[16]) {
  Mat4 m;
  std::memcpy(m.c, a, sizeof m.c);
  return m;
}

static inline Vec3 normalize(Vec3 v) {
  const float l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
  const float r = 1.0f / l;
  return {v.x * r, v.y * r, v.z * r};
}
static inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

Mat4 Mat4::look_to_rh(Vec3 eye, Vec3 dir, Vec3 up) {
  const Vec3 f = normalize(dir);
  const Vec3 s = normalize(cross(f, up));
  const Vec3 u = cross(s, f);
  const float a[16] = {s.x, u.x, -f.x, 0.f, s.y, u.y, -f.y, 0.f, s.z, u.z, -f.z, 0.f, -dot(eye, s), -dot(eye, u), dot(eye, f), 1.f};
  return from_cols_array(a);
}

Mat4 Mat4::operator*(const Mat4& r) const {
  Mat4 o;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
      o.c[4 * j + i] = ((c[i] * r.c[4 * j] + c[4 + i] * r.c[4 * j + 1]) + c[8 + i] * r.c[4 * j + 2]) + c[12 + i] * r.c[4 * j + 3];
  return o;
}

Mat4 Mat4::inverse() const {
  const float m00 = c[0], m01 = c[1], m02 = c[2], m03 = c[3];
  const float m10 = c[4], m11 = c[5], m12 = c[6], m13 = c[7];
  const float m20 = c[8], m21 = c[9], m22 = c[10], m23 = c[11];
  const float m30 = c[12], m31 = c[13], m32 = c[14], m33 = c[15];
  const float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
  const float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
  const float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
  const float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
  const float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
  const float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
  const float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
  const float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
  const float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
  const float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
  const float vec2[4] = {m12, m02, m02, m02}, vec3[4] = {m13, m03, m03, m03};
  const float sa[4] = {1.f, -1.f, 1.f, -1.f}, sb[4] = {-1.f, 1.f, -1.f, 1.f};
  Mat4 inv;
  for (int i = 0; i < 4; ++i) {
    inv.c[i] = ((vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i]) * sa[i];
    inv.c[4 + i] = ((vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i]) * sb[i];
    inv.c[8 + i] = ((vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i]) * sa[i];
    inv.c[12 + i] = ((vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i]) * sb[i];
  }
  const float d0 = m00 * inv.c[0], d1 = m01 * inv.c[4], d2 = m02 * inv.c[8], d3 = m03 * inv.c[12];
  const float det = ((d0 + d1) + d2) + d3;
  const float rcp = 1.0f / det;
  for (float& v : inv.c) v *= rcp;
  return inv;
}

Vec3 Mat4::transform_point3(Vec3 p) const {
  return {((c[0] * p.x + c[4] * p.y) + c[8] * p.z) + c[12], ((c[1] * p.x + c[5] * p.y) + c[9] * p.z) + c[13],
          ((c[2] * p.x + c[6] * p.y) + c[10] * p.z) + c[14]};
}

// ---------------------------------------------------------------------------------------------------
uint64_t SeededRng::next_u64() {  // splitmix64
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
float SeededRng::gen_f32() { return (float)(next_u64() >> 40) * (1.0f / 16777216.0f); }
uint32_t SeededRng::gen_range(uint32_t n) { return (uint32_t)(((next_u64() >> 32) * (uint64_t)n) >> 32); }

void Time::set_bits(uint32_t b) { std::memcpy(&time_, &b, 4); }

uint32_t Time::bits() const {
  uint32_t b;
  std::memcpy(&b, &time_, 4);
  return b;
}

// ---------------------------------------------------------------------------------------------------
Camera::Camera(uint32_t width, uint32_t height) : res_x((float)width), res_y((float)height) {}

Mat4 Camera::clip_space_to_pixel() const {
  const float u = res_x, v = res_y;
  const float a[16] = {u / 2.0f, 0.f, 0.f, 0.f, 0.f, v / 2.0f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, u / 2.0f, v / 2.0f, 0.f, 1.f};
  return Mat4::from_cols_array(a);
}
Mat4 Camera::perspective_proj() const {
  const float n = z_near, f = z_far;
  const float pi = 3.14159265358979323846f;  // std::f32::consts::PI
  const float t = std::tan(fov * pi / 2.0f);
  const float a = res_x / res_y;
  const float m[16] = {1.0f / (a * t), 0.f, 0.f, 0.f, 0.f, 1.0f / t, 0.f, 0.f, 0.f, 0.f, -f / (f - n), -1.f, 0.f, 0.f, -f * n, 0.f};
  return Mat4::from_cols_array(m);
}
Mat4 Camera::rotation_matrix() const { return Mat4::look_to_rh({0.f, 0.f, 0.f}, look_direction, {0.f, 1.f, 0.f}); }

Mat4 Camera::calculate_world_to_pixel() const {
  const float m[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, -pos.x, -pos.y, -pos.z, 1.f};
  const Mat4 world_to_camera_pos = Mat4::from_cols_array(m);
  return clip_space_to_pixel() * perspective_proj() * rotation_matrix() * world_to_camera_pos;
}

rt_camera_raw Camera::to_raw() const {
  rt_camera_raw r{};
  const Mat4 w2p = calculate_world_to_pixel();
  const Mat4 p2w = w2p.inverse();
  std::memcpy(r.world_to_pixel, w2p.c, sizeof w2p.c);
  std::memcpy(r.pixel_to_world, p2w.c, sizeof p2w.c);
  r.pos[0] = pos.x; r.pos[1] = pos.y; r.pos[2] = pos.z; r.pos[3] = pos.z;  // xyzz
  r.focal_length = z_near;
  r.samples_per_pixel = samples_per_pixel;
  r.frames_to_render = frames_to_render;
  r.current_frame = current_frame;
  return r;
}

void Camera::zoom(float val) {
  const float pi = 3.14159265358979323846f;
  const float old = std::log(std::tan(fov * pi / 2.0f));
  const float nw = old + val / 5.0f;
  fov = 2.0f * std::atan(std::exp(nw)) / pi;
}

void Camera::look_at_pixel_from_center(float off_x, float off_y) {
  const Vec3 offset_3d{res_x / 2.0f + off_x, res_y / 2.0f + off_y, 0.0f};
  const Mat4 pixel_to_world = calculate_world_to_pixel().inverse();
  look_direction = normalize(pixel_to_world.transform_point3(offset_3d));
}

bool Camera::key_press(Key key) {
  auto clampu = [](uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
  auto move = [&](Vec3 dir) {
    const Vec3 p = normalize(rotation_matrix().inverse().transform_point3(dir));
    pos = {pos.x + p.x * speed, pos.y + p.y * speed, pos.z + p.z * speed};
  };
  switch (key) {
    case Key::K: samples_per_pixel = clampu(samples_per_pixel * 2u, 1u, 256u); break;
    case Key::J: samples_per_pixel = clampu(samples_per_pixel / 2u, 1u, 256u); break;
    case Key::I: frames_to_render = clampu(frames_to_render * 2u, 1u, 256u); break;
    case Key::U: frames_to_render = clampu(frames_to_render / 2u, 1u, 256u); break;
    case Key::ArrowUp: zoom(-1.0f); break;
    case Key::ArrowDown: zoom(1.0f); break;
    case Key::Backspace: frames_to_render = 1; samples_per_pixel = 1; break;
    case Key::W: move({0.f, 0.f, -1.f}); break;
    case Key::S: move({0.f, 0.f, 1.f}); break;
    case Key::D: move({1.f, 0.f, 0.f}); break;
    case Key::A: move({-1.f, 0.f, 0.f}); break;
    case Key::Space: move({0.f, 1.f, 0.f}); break;
    case Key::ShiftLeft: move({0.f, -1.f, 0.f}); break;
  }
  reset_render();  // every handled key ends in scene_was_updated (camera.rs:172-174)
  return true;
}

std::string Camera::to_string() const {
  return "Frames: " + std::to_string(frames_to_render) + ", RaysPerPixel: " + std::to_string(samples_per_pixel);
}

// ---------------------------------------------------------------------------------------------------
rt_sphere Sphere::make(Vec3 center, float radius, uint32_t material_index) {
  rt_sphere s{};
  s.center[0] = center.x; s.center[1] = center.y; s.center[2] = center.z;
  std::memcpy(&s.padding, &center.z, 4);  // center.xyzz()
  s.radius = radius;
  s.material_index = material_index;
  return s;
}

std::vector<rt_sphere> Sphere::random_bunch(uint32_t amount, uint32_t materials_len, SeededRng& rng) {
  const float root = std::sqrt((float)amount);
  const Vec3 from{-1.0f * root, -1.0f * root, -1.0f * root};
  const Vec3 to{1.0f * root, 1.0f * root, -2.0f * root};
  std::vector<rt_sphere> out;
  out.reserve(amount);
  for (uint32_t i = 0; i < amount; ++i) {
    const float rx = rng.gen_f32(), ry = rng.gen_f32(), rz = rng.gen_f32();
    const Vec3 c{rx * (to.x - from.x) + from.x, ry * (to.y - from.y) + from.y, rz * (to.z - from.z) + from.z};
    // The reference draws gen_range(0..materials_len + 4) (objects.rs:37), which indexes past the
    // material buffer; the result is driver-defined, so this host draws in range instead.
    out.push_back(make(c, 1.0f, rng.gen_range(materials_len)));
  }
  return out;
}

rt_material Material::random_new(SeededRng& rng) {
  rt_material m{};
  for (float& c : m.color) c = rng.gen_f32();
  m.metallic = rng.gen_f32();
  m.roughness = rng.gen_f32();
  m.specular = rng.gen_f32();
  m.padding = 0.0f;
  return m;
}

World World::default_scene(uint64_t seed) {
  SeededRng rng(seed);
  World w;
  w.materials = {Material::random_new(rng), Material::random_new(rng), Material::random_new(rng)};
  w.spheres = {Sphere::make({0.0f, -100.5f, 0.0f}, 100.0f, 0), Sphere::make({-1.0f, 0.0f, -1.0f}, 0.5f, 1),
               Sphere::make({1.0f, 0.0f, -1.0f}, 0.5f, 2), Sphere::make({0.0f, 0.0f, -1.0f}, 0.5f, 0)};
  return w;
}

World World::random_bunch(uint32_t amount, uint32_t n_materials, uint64_t seed) {
  SeededRng rng(seed);
  World w;
  for (uint32_t i = 0; i < n_materials; ++i) w.materials.push_back(Material::random_new(rng));
  w.spheres = Sphere::random_bunch(amount, n_materials, rng);
  return w;
}

// ---------------------------------------------------------------------------------------------------
static void must(int rc, const char* what) {
  if (rc != RT_OK) throw Error{rc, std::string(what) + ": " + rt_last_error()};
}

Raytracer::Raytracer(int device, uint32_t width, uint32_t height, uint32_t flags) {
  must(rt_create(&ctx_, device, width, height, flags), "rt_create");
}
Raytracer::~Raytracer() { rt_destroy(ctx_); }
void Raytracer::bind_world(const World& w) {
  must(rt_set_scene(ctx_, w.spheres.data(), (uint32_t)w.spheres.size(), w.materials.data(), (uint32_t)w.materials.size()), "rt_set_scene");
}
void Raytracer::write_camera(const rt_camera_raw& c) { must(rt_set_camera(ctx_, &c), "rt_set_camera"); }
void Raytracer::write_time(uint32_t bits) { must(rt_set_time(ctx_, bits), "rt_set_time"); }
void Raytracer::compute() { must(rt_compute(ctx_), "rt_compute"); }

// ---------------------------------------------------------------------------------------------------
App::App(int device, uint32_t width, uint32_t height, uint32_t flags, World w)
    : camera(width, height), world(std::move(w)), ray_tracer(device, width, height, flags), width_(width), height_(height), flags_(flags) {
  ray_tracer.bind_world(world);
  ray_tracer.write_camera(camera.to_raw());  // Camera::new -> scene_was_updated (camera.rs:74,283-288)
}

void App::update() {
  time.add_delta(0.01f);  // app.rs:94
  ray_tracer.write_time(time.bits());
}

void App::render() {
  if (!camera.render_finished()) ray_tracer.compute();  // app.rs:106-108
  camera.increase_frame();                              // app.rs:113
  ray_tracer.write_camera(camera.to_raw());             // camera.rs:264
}

void App::run(uint32_t redraws) {
  if (redraws == 0) return;
  std::vector<uint32_t> bits(redraws);
  for (uint32_t i = 0; i < redraws; ++i) {
    time.add_delta(0.01f);
    bits[i] = time.bits();
  }
  ray_tracer.write_camera(camera.to_raw());
  must(rt_render_frames(ray_tracer.ctx(), bits.data(), redraws), "rt_render_frames");
  camera.current_frame += redraws;
}

// ---------------------------------------------------------------------------------------------------
// PNG, stored (uncompressed) deflate blocks
static uint32_t crc_table[256];
static void crc_init() {
  static bool done = false;
  if (done) return;
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crc_table[n] = c;
  }
  done = true;
}
static uint32_t crc_update(uint32_t crc, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
  return crc;
}
static void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
static void chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& data) {
  put_be32(out, (uint32_t)data.size());
  const size_t start = out.size();
  out.insert(out.end(), tag, tag + 4);
  out.insert(out.end(), data.begin(), data.end());
  put_be32(out, crc_update(0xffffffffu, out.data() + start, out.size() - start) ^ 0xffffffffu);
}

bool write_png(const std::string& path, const uint8_t* rgba8, uint32_t width, uint32_t height, bool present) {
  crc_init();
  uint8_t lut[256];
  for (int i = 0; i < 256; ++i) {
    if (!present) { lut[i] = (uint8_t)i; continue; }
    const double lin = std::pow(i / 255.0, 2.2);  // screen_shader.wgsl:36-39
    const double s = lin <= 0.0031308 ? 12.92 * lin : 1.055 * std::pow(lin, 1.0 / 2.4) - 0.055;  // sRGB surface
    lut[i] = (uint8_t)std::lround(std::fmin(std::fmax(s, 0.0), 1.0) * 255.0);
  }
  std::vector<uint8_t> raw;
  raw.reserve((size_t)height * (1 + (size_t)width * 3));
  for (uint32_t y = 0; y < height; ++y) {
    raw.push_back(0);
    for (uint32_t x = 0; x < width; ++x) {
      const uint8_t* p = rgba8 + ((size_t)y * width + x) * 4;
      raw.push_back(lut[p[0]]); raw.push_back(lut[p[1]]); raw.push_back(lut[p[2]]);
    }
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  size_t off = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - off);
    z.push_back(off + n == raw.size() ? 1 : 0);
    z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
    for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
    off += n;
  } while (off < raw.size());
  put_be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, width); put_be32(ihdr, height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  return std::fclose(f) == 0 && ok;
}


// ---------------------------------------------------------------------------------------------------
// Checkpoint file: 128-byte little-endian header, then the accumulation image exactly as the device holds
// it (float4 or rgba8 per pixel).
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr char kCkptMagic[8] = {'R', 'T', 'C', 'K', 'P', 'T', '0', '1'};
struct CkptHeader {
  char magic[8];
  uint32_t width, height, flags, current_frame, frames_to_render, samples_per_pixel, time_bits, n_spheres, n_materials;
  uint32_t reserved0;
  uint64_t scene_hash, image_bytes;
  float pos[3], look[3], z_near, z_far, fov, speed;
  uint8_t pad[128 - 8 - 10 * 4 - 2 * 8 - 10 * 4];
};
static_assert(sizeof(CkptHeader) == 128, "checkpoint header is 128 bytes");

uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}
uint64_t scene_hash(const World& w) {
  uint64_t h = 0xcbf29ce484222325ull;
  h = fnv1a(h, w.spheres.data(), w.spheres.size() * sizeof(rt_sphere));
  return fnv1a(h, w.materials.data(), w.materials.size() * sizeof(rt_material));
}
size_t image_bytes_of(uint32_t w, uint32_t h, uint32_t flags) { return (size_t)w * h * ((flags & RT_ACCUM_F32) ? 16u : 4u); }

CkptHeader read_header(std::FILE* f, const std::string& path) {
  CkptHeader hd;
  if (std::fread(&hd, sizeof hd, 1, f) != 1) throw Error{RT_ERR_INVALID, "checkpoint " + path + ": shorter than its header"};
  if (std::memcmp(hd.magic, kCkptMagic, 8) != 0) throw Error{RT_ERR_INVALID, "checkpoint " + path + ": bad magic"};
  if (hd.width == 0 || hd.height == 0 || hd.width > 65536u || hd.height > 65536u || hd.image_bytes != image_bytes_of(hd.width, hd.height, hd.flags))
    throw Error{RT_ERR_INVALID, "checkpoint " + path + ": inconsistent image size"};
  return hd;
}
}  // namespace

CheckpointInfo read_checkpoint_info(const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw Error{RT_ERR_INVALID, "cannot open checkpoint " + path};
  CkptHeader hd;
  try { hd = read_header(f, path); } catch (...) { std::fclose(f); throw; }
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fclose(f);
  if (size < 0 || (uint64_t)size != sizeof(CkptHeader) + hd.image_bytes) throw Error{RT_ERR_INVALID, "checkpoint " + path + ": truncated image"};
  return {hd.width, hd.height, hd.flags, hd.current_frame, hd.frames_to_render, hd.samples_per_pixel, hd.time_bits, hd.n_spheres, hd.n_materials,
          hd.scene_hash, hd.image_bytes};
}

void App::save_checkpoint(const std::string& path) {
  CkptHeader hd{};
  std::memcpy(hd.magic, kCkptMagic, 8);
  hd.width = width_; hd.height = height_; hd.flags = flags_;
  hd.current_frame = camera.current_frame; hd.frames_to_render = camera.frames_to_render; hd.samples_per_pixel = camera.samples_per_pixel;
  hd.time_bits = time.bits();
  hd.n_spheres = (uint32_t)world.spheres.size(); hd.n_materials = (uint32_t)world.materials.size();
  hd.scene_hash = scene_hash(world);
  hd.image_bytes = image_bytes_of(width_, height_, flags_);
  hd.pos[0] = camera.pos.x; hd.pos[1] = camera.pos.y; hd.pos[2] = camera.pos.z;
  hd.look[0] = camera.look_direction.x; hd.look[1] = camera.look_direction.y; hd.look[2] = camera.look_direction.z;
  hd.z_near = camera.z_near; hd.z_far = camera.z_far; hd.fov = camera.fov; hd.speed = camera.speed;
  std::vector<uint8_t> img(hd.image_bytes);
  const int rc = (flags_ & RT_ACCUM_F32) ? rt_read_accum_f32(ray_tracer.ctx(), img.data(), img.size())
                                         : rt_read_rgba8(ray_tracer.ctx(), img.data(), img.size());
  must(rc, "checkpoint read-back");
  std::FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw Error{RT_ERR_INVALID, "cannot create checkpoint " + path};
  const bool ok = std::fwrite(&hd, sizeof hd, 1, f) == 1 && std::fwrite(img.data(), 1, img.size(), f) == img.size();
  if (std::fclose(f) != 0 || !ok) throw Error{RT_ERR_INVALID, "short write to checkpoint " + path};
}

void App::load_checkpoint(const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw Error{RT_ERR_INVALID, "cannot open checkpoint " + path};
  std::vector<uint8_t> img;
  CkptHeader hd;
  try {
    hd = read_header(f, path);
    // the accumulation format and the dispatch/shading flags decide what the image means; the arithmetic
    // mode may differ (a strict render can be continued in fast mode and vice versa)
    const uint32_t same = ~(uint32_t)RT_MATH_FAST;
    if (hd.width != width_ || hd.height != height_ || (hd.flags & same) != (flags_ & same))
      throw Error{RT_ERR_INVALID, "checkpoint " + path + ": image size or flags differ from this App"};
    if (hd.n_spheres != world.spheres.size() || hd.n_materials != world.materials.size() || hd.scene_hash != scene_hash(world))
      throw Error{RT_ERR_INVALID, "checkpoint " + path + ": was rendered from another scene"};
    img.resize(hd.image_bytes);
    if (std::fread(img.data(), 1, img.size(), f) != img.size()) throw Error{RT_ERR_INVALID, "checkpoint " + path + ": truncated image"};
  } catch (...) { std::fclose(f); throw; }
  std::fclose(f);
  must(rt_write_image(ray_tracer.ctx(), img.data(), img.size()), "checkpoint upload");
  camera.pos = {hd.pos[0], hd.pos[1], hd.pos[2]};
  camera.look_direction = {hd.look[0], hd.look[1], hd.look[2]};
  camera.z_near = hd.z_near; camera.z_far = hd.z_far; camera.fov = hd.fov; camera.speed = hd.speed;
  camera.current_frame = hd.current_frame; camera.frames_to_render = hd.frames_to_render; camera.samples_per_pixel = hd.samples_per_pixel;
  time.set_bits(hd.time_bits);
  ray_tracer.write_camera(camera.to_raw());
  ray_tracer.write_time(time.bits());
}

}  // namespace rt

// ===================================================================================================
// C ABI of the host mirror (include/rt_host.h): lets tests, bench.py and other FFI callers obtain
// the exact byte images the C++ host produces and drive the App loop.
// ===================================================================================================
namespace {
thread_local std::string g_herr;
}

extern "C" {

const char* rth_last_error(void) { return g_herr.c_str(); }

int rth_camera_raw(uint32_t width, uint32_t height, const float pos[3], const float look[3], float fov, uint32_t spp,
                   uint32_t frames, uint32_t current_frame, rt_camera_raw* out) {
  if (!out || width == 0 || height == 0) { g_herr = "rth_camera_raw: bad argument"; return RT_ERR_INVALID; }
  rt::Camera c(width, height);
  if (pos) c.pos = {pos[0], pos[1], pos[2]};
  if (look) c.look_direction = {look[0], look[1], look[2]};
  if (fov > 0.f) c.fov = fov;
  c.samples_per_pixel = spp;
  c.frames_to_render = frames;
  c.current_frame = current_frame;
  *out = c.to_raw();
  return RT_OK;
}

int rth_scene_default(uint64_t seed, rt_sphere* spheres4, rt_material* materials3) {
  if (!spheres4 || !materials3) { g_herr = "rth_scene_default: null output"; return RT_ERR_INVALID; }
  const rt::World w = rt::World::default_scene(seed);
  std::memcpy(spheres4, w.spheres.data(), 4 * sizeof(rt_sphere));
  std::memcpy(materials3, w.materials.data(), 3 * sizeof(rt_material));
  return RT_OK;
}

int rth_scene_random_bunch(uint32_t amount, uint32_t n_materials, uint64_t seed, rt_sphere* spheres, rt_material* materials) {
  if (!spheres || !materials || amount == 0 || n_materials == 0) { g_herr = "rth_scene_random_bunch: bad argument"; return RT_ERR_INVALID; }
  const rt::World w = rt::World::random_bunch(amount, n_materials, seed);
  std::memcpy(spheres, w.spheres.data(), amount * sizeof(rt_sphere));
  std::memcpy(materials, w.materials.data(), n_materials * sizeof(rt_material));
  return RT_OK;
}

int rth_time_table(uint32_t start_bits, uint32_t n, uint32_t* out) {
  if (!out) { g_herr = "rth_time_table: null output"; return RT_ERR_INVALID; }
  float t;
  std::memcpy(&t, &start_bits, 4);
  rt::Time tm;
  tm.add_delta(t);  // 0 + t == t exactly
  for (uint32_t i = 0; i < n; ++i) {
    tm.add_delta(0.01f);
    out[i] = tm.bits();
  }
  return RT_OK;
}

struct rth_camera {
  rt::Camera cam;
};

rth_camera* rth_camera_create(uint32_t width, uint32_t height) { return (width && height) ? new rth_camera{rt::Camera(width, height)} : nullptr; }
void rth_camera_destroy(rth_camera* c) { delete c; }
static bool parse_key(const char* key, rt::Camera::Key* out) {
  static const struct { const char* name; rt::Camera::Key k; } table[] = {
      {"K", rt::Camera::Key::K}, {"J", rt::Camera::Key::J}, {"I", rt::Camera::Key::I}, {"U", rt::Camera::Key::U},
      {"ArrowUp", rt::Camera::Key::ArrowUp}, {"ArrowDown", rt::Camera::Key::ArrowDown}, {"Backspace", rt::Camera::Key::Backspace},
      {"W", rt::Camera::Key::W}, {"S", rt::Camera::Key::S}, {"D", rt::Camera::Key::D}, {"A", rt::Camera::Key::A},
      {"Space", rt::Camera::Key::Space}, {"ShiftLeft", rt::Camera::Key::ShiftLeft}};
  for (const auto& e : table)
    if (std::strcmp(e.name, key) == 0) { *out = e.k; return true; }
  g_herr = std::string("unknown key ") + key;
  return false;
}
int rth_camera_key(rth_camera* c, const char* key) {
  if (!c || !key) { g_herr = "rth_camera_key: null argument"; return RT_ERR_INVALID; }
  rt::Camera::Key k;
  if (!parse_key(key, &k)) return RT_ERR_INVALID;
  c->cam.key_press(k);
  return RT_OK;
}
int rth_camera_zoom(rth_camera* c, float val) { if (!c) return RT_ERR_INVALID; c->cam.zoom(val); c->cam.reset_render(); return RT_OK; }
int rth_camera_look_at_pixel(rth_camera* c, float off_x, float off_y) {
  if (!c) return RT_ERR_INVALID;
  c->cam.look_at_pixel_from_center(off_x, off_y);
  c->cam.reset_render();
  return RT_OK;
}
int rth_camera_increase_frame(rth_camera* c) { if (!c) return RT_ERR_INVALID; c->cam.increase_frame(); return RT_OK; }
int rth_camera_render_finished(rth_camera* c) { return c && c->cam.render_finished() ? 1 : 0; }
int rth_camera_get(rth_camera* c, rt_camera_raw* out, float* fov) {
  if (!c || !out) { g_herr = "rth_camera_get: null argument"; return RT_ERR_INVALID; }
  *out = c->cam.to_raw();
  if (fov) *fov = c->cam.fov;
  return RT_OK;
}

struct rth_app {
  rt::App app;
};

int rth_app_create(rth_app** out, int device, uint32_t width, uint32_t height, uint32_t flags, const rt_sphere* spheres,
                   uint32_t n_spheres, const rt_material* materials, uint32_t n_materials, uint32_t spp, uint32_t frames) {
  if (!out || !spheres || !materials) { g_herr = "rth_app_create: null argument"; return RT_ERR_INVALID; }
  *out = nullptr;
  try {
    rt::World w;
    w.spheres.assign(spheres, spheres + n_spheres);
    w.materials.assign(materials, materials + n_materials);
    rth_app* a = new rth_app{rt::App(device, width, height, flags, std::move(w))};
    a->app.camera.samples_per_pixel = spp;
    a->app.camera.frames_to_render = frames;
    a->app.ray_tracer.write_camera(a->app.camera.to_raw());
    *out = a;
    return RT_OK;
  } catch (const rt::Error& e) {
    g_herr = e.what;
    return e.code;
  } catch (...) {  // nothing may unwind across the C ABI
    g_herr = "host exception (out of memory?)";
    return RT_ERR_INVALID;
  }
}

void rth_app_destroy(rth_app* a) { delete a; }

static int guarded(rth_app* a, void (*fn)(rt::App&, uint32_t), uint32_t n) {
  if (!a) { g_herr = "null app"; return RT_ERR_INVALID; }
  try {
    fn(a->app, n);
    return RT_OK;
  } catch (const rt::Error& e) {
    g_herr = e.what;
    return e.code;
  } catch (...) {  // nothing may unwind across the C ABI
    g_herr = "host exception (out of memory?)";
    return RT_ERR_INVALID;
  }
}

int rth_app_redraw(rth_app* a, uint32_t n) {  // n x (App::update, App::render), one launch per frame
  return guarded(a, [](rt::App& app, uint32_t k) { for (uint32_t i = 0; i < k; ++i) { app.update(); app.render(); } }, n);
}
int rth_app_run(rth_app* a, uint32_t n) {  // same redraws, batched into one launch
  return guarded(a, [](rt::App& app, uint32_t k) { app.run(k); }, n);
}
// App::on_event's camera half (app.rs:50-91 -> camera.rs:133-206) without winit: the control acts on the App's own
// camera, then scene_was_updated (camera.rs:283-288): reset_render + upload of the new CameraRaw.
static int app_camera_changed(rth_app* a) {
  try {
    a->app.camera.reset_render();
    a->app.ray_tracer.write_camera(a->app.camera.to_raw());
    return RT_OK;
  } catch (const rt::Error& e) { g_herr = e.what; return e.code; } catch (...) { g_herr = "host exception (out of memory?)"; return RT_ERR_INVALID; }
}
int rth_app_camera_key(rth_app* a, const char* key) {
  if (!a || !key) { g_herr = "rth_app_camera_key: null argument"; return RT_ERR_INVALID; }
  rt::Camera::Key k;
  if (!parse_key(key, &k)) return RT_ERR_INVALID;
  a->app.camera.key_press(k);
  return app_camera_changed(a);
}
int rth_app_camera_zoom(rth_app* a, float val) {
  if (!a) { g_herr = "null app"; return RT_ERR_INVALID; }
  a->app.camera.zoom(val);
  return app_camera_changed(a);
}
int rth_app_camera_look_at_pixel(rth_app* a, float off_x, float off_y) {
  if (!a) { g_herr = "null app"; return RT_ERR_INVALID; }
  a->app.camera.look_at_pixel_from_center(off_x, off_y);
  return app_camera_changed(a);
}
int rth_app_camera_get(rth_app* a, rt_camera_raw* out, float* fov) {
  if (!a || !out) { g_herr = "rth_app_camera_get: null argument"; return RT_ERR_INVALID; }
  *out = a->app.camera.to_raw();
  if (fov) *fov = a->app.camera.fov;
  return RT_OK;
}
int rth_app_save_checkpoint(rth_app* a, const char* path) {
  if (!a || !path) { g_herr = "rth_app_save_checkpoint: null argument"; return RT_ERR_INVALID; }
  try { a->app.save_checkpoint(path); } catch (const rt::Error& e) { g_herr = e.what; return e.code; } catch (...) { g_herr = "host exception (out of memory?)"; return RT_ERR_INVALID; }
  return RT_OK;
}
int rth_app_load_checkpoint(rth_app* a, const char* path) {
  if (!a || !path) { g_herr = "rth_app_load_checkpoint: null argument"; return RT_ERR_INVALID; }
  try { a->app.load_checkpoint(path); } catch (const rt::Error& e) { g_herr = e.what; return e.code; } catch (...) { g_herr = "host exception (out of memory?)"; return RT_ERR_INVALID; }
  return RT_OK;
}
int rth_checkpoint_info(const char* path, uint32_t out9[9], uint64_t* scene_hash, uint64_t* image_bytes) {
  if (!path || !out9) { g_herr = "rth_checkpoint_info: null argument"; return RT_ERR_INVALID; }
  try {
    const rt::CheckpointInfo i = rt::read_checkpoint_info(path);
    const uint32_t v[9] = {i.width, i.height, i.flags, i.current_frame, i.frames_to_render, i.samples_per_pixel, i.time_bits, i.n_spheres, i.n_materials};
    std::memcpy(out9, v, sizeof v);
    if (scene_hash) *scene_hash = i.scene_hash;
    if (image_bytes) *image_bytes = i.image_bytes;
  } catch (const rt::Error& e) { g_herr = e.what; return e.code; } catch (...) { g_herr = "host exception (out of memory?)"; return RT_ERR_INVALID; }
  return RT_OK;
}
rt_ctx* rth_app_ctx(rth_app* a) { return a ? a->app.ray_tracer.ctx() : nullptr; }
uint32_t rth_app_current_frame(rth_app* a) { return a ? a->app.camera.current_frame : 0; }
uint32_t rth_app_time_bits(rth_app* a) { return a ? a->app.time.bits() : 0; }

int rth_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int present) {
  if (!path || !rgba8) { g_herr = "rth_write_png: null argument"; return RT_ERR_INVALID; }
  if (!rt::write_png(path, rgba8, width, height, present != 0)) { g_herr = std::string("cannot write ") + path; return RT_ERR_INVALID; }
  return RT_OK;
}

}  // extern "C"
