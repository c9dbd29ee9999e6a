// rt_headless — headless CLI over the host mirror: renders a scene with the reference's
// App::update / App::render loop (src/app.rs:93-116) and writes a PNG.
//   rt_headless --width 1920 --height 1080 --spp 16 --frames 256 --scene default --seed 1 --out out.png
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.hpp"

static const char* kUsage =
    "usage: rt_headless [--width W] [--height H] [--spp S] [--frames F] [--scene default|bunch:N|checkers]\n"
    "                   [--seed K] [--materials M] [--accum rgba8|f32] [--math strict|fast] [--dispatch floor16|full]\n"
    "                   [--device D] [--per-frame] [--present] [--out file.png]\n"
    "                   [--resume file.ckpt] [--stop-after N] [--save-checkpoint file.ckpt]\n";

int main(int argc, char** argv) {
  uint32_t width = 2048, height = 1024, spp = 1, frames = 8, n_mat = 3;  // lib.rs:16, camera.rs:66-67
  uint64_t seed = 1;
  uint32_t flags = 0;
  int device = 0;
  bool per_frame = false, present = false;
  std::string scene = "default", out, resume, save;
  uint32_t stop_after = 0;  // redraws to run in this invocation (0 = until frames_to_render)
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto need = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n%s", name, kUsage); std::exit(2); }
      return argv[++i];
    };
    if (a == "--width") width = (uint32_t)std::atoi(need("--width"));
    else if (a == "--height") height = (uint32_t)std::atoi(need("--height"));
    else if (a == "--spp") spp = (uint32_t)std::atoi(need("--spp"));
    else if (a == "--frames") frames = (uint32_t)std::atoi(need("--frames"));
    else if (a == "--materials") n_mat = (uint32_t)std::atoi(need("--materials"));
    else if (a == "--seed") seed = std::strtoull(need("--seed"), nullptr, 10);
    else if (a == "--device") device = std::atoi(need("--device"));
    else if (a == "--scene") scene = need("--scene");
    else if (a == "--out") out = need("--out");
    else if (a == "--accum") { if (std::string(need("--accum")) == "f32") flags |= RT_ACCUM_F32; }
    else if (a == "--math") { if (std::string(need("--math")) == "fast") flags |= RT_MATH_FAST; }
    else if (a == "--dispatch") { if (std::string(need("--dispatch")) == "full") flags |= RT_DISPATCH_FULL; }
    else if (a == "--resume") resume = need("--resume");
    else if (a == "--save-checkpoint") save = need("--save-checkpoint");
    else if (a == "--stop-after") stop_after = (uint32_t)std::atoi(need("--stop-after"));
    else if (a == "--per-frame") per_frame = true;
    else if (a == "--present") present = true;
    else { std::fprintf(stderr, "unknown option %s\n%s", a.c_str(), kUsage); return 2; }
  }
  try {
    rt::World world;
    if (scene == "default") world = rt::World::default_scene(seed);
    else if (scene == "checkers") { world = rt::World::random_bunch(20, n_mat, seed); flags |= RT_SHADE_CHECKER; }
    else if (scene.rfind("bunch:", 0) == 0) world = rt::World::random_bunch((uint32_t)std::atoi(scene.c_str() + 6), n_mat, seed);
    else { std::fprintf(stderr, "unknown scene %s\n%s", scene.c_str(), kUsage); return 2; }
    rt::App app(device, width, height, flags, std::move(world));
    app.camera.samples_per_pixel = spp;
    app.camera.frames_to_render = frames;
    app.camera.reset_render();
    // scene_was_updated (camera.rs:283-288): the constructor uploaded Camera::new's defaults (spp 1, 8 frames);
    // App::render computes BEFORE its own camera upload, so frame 0 of --per-frame needs this one
    app.ray_tracer.write_camera(app.camera.to_raw());
    if (!resume.empty()) {
      app.load_checkpoint(resume);
      std::printf("resumed %s at frame %u of %u\n", resume.c_str(), app.camera.current_frame, app.camera.frames_to_render);
    }
    const uint32_t left = app.camera.current_frame < app.camera.frames_to_render ? app.camera.frames_to_render - app.camera.current_frame : 0u;
    const uint32_t todo = stop_after ? (stop_after < left ? stop_after : left) : left;
    std::printf("\nRendering with:\n%s\n", app.camera.to_string().c_str());  // camera.rs:276
    const auto t0 = std::chrono::steady_clock::now();
    if (per_frame) for (uint32_t f = 0; f < todo; ++f) { app.update(); app.render(); }
    else app.run(todo);
    if (rt_sync(app.ray_tracer.ctx()) != RT_OK) throw rt::Error{RT_ERR_HIP, rt_last_error()};
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const uint32_t cw = (flags & RT_DISPATCH_FULL) ? width : width / 16 * 16, ch = (flags & RT_DISPATCH_FULL) ? height : height / 16 * 16;
    const double samples = (double)cw * ch * app.camera.samples_per_pixel * todo;
    std::printf("%u x %u, %u spp x %u frames: %.3f s, %.3f Gsamples/s\n", width, height, app.camera.samples_per_pixel, todo, sec, samples / sec * 1e-9);
    if (!save.empty()) {
      app.save_checkpoint(save);
      std::printf("saved %s at frame %u of %u\n", save.c_str(), app.camera.current_frame, app.camera.frames_to_render);
    }
    if (!out.empty()) {
      std::vector<uint8_t> img((size_t)width * height * 4);
      if (rt_read_rgba8(app.ray_tracer.ctx(), img.data(), img.size()) != RT_OK) throw rt::Error{RT_ERR_HIP, rt_last_error()};
      if (!rt::write_png(out, img.data(), width, height, present)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
      std::printf("wrote %s\n", out.c_str());
    }
  } catch (const rt::Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what.c_str());
    return 1;
  }
  return 0;
}
