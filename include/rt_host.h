/* rt_host.h — C ABI of librt_host.so, the headless C++ mirror of the reference's Rust host
 * (Camera, World, Time, App; raytracer_amd/csrc/host/rt_host.hpp). It sits ABOVE the kernel
 * boundary rt_hip.h: a Rust host would keep its own camera.rs / world.rs / app.rs and bind
 * rt_hip.h only. These entry points exist so tests, bench.py and the CLI share one host
 * implementation of the inputs.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_app rth_app;

/* Camera::to_raw (reference src/resources/camera.rs:77-131). pos/look may be NULL (defaults 0 and
 * (0,0,-1), camera.rs:61,70); fov <= 0 selects the default 0.25 (camera.rs:65). */
int rth_camera_raw(uint32_t width, uint32_t height, const float pos[3], const float look[3], float fov, uint32_t spp,
                   uint32_t frames, uint32_t current_frame, rt_camera_raw* out);

/* A stateful headless Camera (src/resources/camera.rs:23-36): the interactive controls without winit.
 * rth_camera_key takes the reference's key names: "K"/"J" spp x2 / /2 and "I"/"U" frames x2 / /2, both
 * clamped to [1,256]; "Backspace" -> 1/1; "ArrowUp"/"ArrowDown" zoom; "W","S","D","A","Space","ShiftLeft"
 * move by `speed` = 0.05 along the view axes (camera.rs:133-175). Every handled input resets
 * current_frame, as scene_was_updated does (camera.rs:283-288). */
typedef struct rth_camera rth_camera;
rth_camera* rth_camera_create(uint32_t width, uint32_t height);
void rth_camera_destroy(rth_camera* cam);
int rth_camera_key(rth_camera* cam, const char* key);
int rth_camera_zoom(rth_camera* cam, float val);                              /* camera.rs:193-197 */
int rth_camera_look_at_pixel(rth_camera* cam, float off_x, float off_y);      /* camera.rs:200-206 */
int rth_camera_increase_frame(rth_camera* cam);                               /* camera.rs:252-265 */
int rth_camera_render_finished(rth_camera* cam);                              /* camera.rs:279-281 */
int rth_camera_get(rth_camera* cam, rt_camera_raw* out, float* fov);          /* Camera::to_raw */

/* World::new (src/world.rs:34-40): 4 spheres, 3 materials drawn from a generator seeded with `seed`
 * (the reference uses the unseeded thread_rng, material.rs:22-37). */
int rth_scene_default(uint64_t seed, rt_sphere* spheres4, rt_material* materials3);

/* Sphere::random_bunch (src/resources/objects.rs:27-40) + n_materials random materials; material
 * indices are drawn in range (the reference's 0..len+4 overruns the buffer). */
int rth_scene_random_bunch(uint32_t amount, uint32_t n_materials, uint64_t seed, rt_sphere* spheres, rt_material* materials);

/* Time::add_delta(0.01) applied n times from start_bits (src/common/time.rs:72-77, src/app.rs:94):
 * out[k] = time bits of redraw k. */
int rth_time_table(uint32_t start_bits, uint32_t n, uint32_t* out);

/* App::new / App::update / App::render (src/app.rs:27-48,93-116), compute pass only. */
int rth_app_create(rth_app** out, int device, uint32_t width, uint32_t height, uint32_t flags, const rt_sphere* spheres,
                   uint32_t n_spheres, const rt_material* materials, uint32_t n_materials, uint32_t spp, uint32_t frames);
void rth_app_destroy(rth_app* app);
int rth_app_redraw(rth_app* app, uint32_t n); /* n x (update, render): one launch per frame */
int rth_app_run(rth_app* app, uint32_t n);    /* the same n redraws in one batched launch   */
rt_ctx* rth_app_ctx(rth_app* app);            /* borrow the Raytracer's context for readbacks */
uint32_t rth_app_current_frame(rth_app* app);
uint32_t rth_app_time_bits(rth_app* app);

/* The App's camera under the interactive controls (App::on_event -> Camera::key_press / zoom / look_at_pixel,
 * src/app.rs:50-91, src/resources/camera.rs:133-206) without winit: same key names and rules as rth_camera_*; each
 * call ends with scene_was_updated (camera.rs:283-288): current_frame = 0 and the new CameraRaw is uploaded.
 * rth_app_camera_get returns the 160 bytes the next pass will use. One call is one key pressed with only
 * that key held (keys_held = {key}); several movement keys held at once in the reference equal one call per
 * key here, because every translation goes through the same inverse rotation (camera.rs:149-170). */
int rth_app_camera_key(rth_app* app, const char* key);
int rth_app_camera_zoom(rth_app* app, float val);
int rth_app_camera_look_at_pixel(rth_app* app, float off_x, float off_y);
int rth_app_camera_get(rth_app* app, rt_camera_raw* out, float* fov);

/* Checkpoint / resume. No counterpart in the reference: its accumulation lives in two wgpu
 * textures and a render always restarts from frame 0 (src/resources/camera.rs:274-277). The file holds a
 * 128-byte header (image size, flags, frame counter, time bits, camera, a hash of the scene) and the
 * accumulation image as the device holds it; loading checks size, flags (the arithmetic mode may differ)
 * and scene, restores App::time and App::camera, and the render continues bit-identically to one that was
 * never interrupted. rth_checkpoint_info reads the header without a GPU:
 * out9 = {width, height, flags, current_frame, frames_to_render, samples_per_pixel, time_bits, n_spheres, n_materials}. */
int rth_app_save_checkpoint(rth_app* app, const char* path);
int rth_app_load_checkpoint(rth_app* app, const char* path);
int rth_checkpoint_info(const char* path, uint32_t out9[9], uint64_t* scene_hash, uint64_t* image_bytes);

/* PNG writer. present != 0 applies the reference's screen pass: pow(c, 2.2)
 * (src/shaders/screen_shader.wgsl:36-39) then the sRGB surface encode (src/render_env.rs:48). */
int rth_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height, int present);

const char* rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
