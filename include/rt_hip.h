/* rt_hip.h — C ABI of librt_hip.so: the MI355X (gfx950) replacement for the reference's
 * path-tracing compute pass.
 *
 * The reference has no FFI for this path; its seam is the Rust struct `Raytracer`
 * (reference src/raytracing.rs:6-13). Each entry point below names the piece of that seam it
 * replaces. A Rust host binds these with `extern "C"` (INTEGRATION.md shows the ffi.rs stub).
 *
 * Conventions: opaque handle, POD in/out, plain pointers and sizes, int status (0 = ok, <0 = error,
 * message via rt_last_error()). Nothing throws or aborts. One context drives one GPU from one
 * host thread at a time (the reference is single-threaded: src/lib.rs:29). All work is queued on
 * the context's HIP stream; rt_sync() is the equivalent of queue.submit + wait.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

#define RT_OK 0
#define RT_ERR_INVALID -1   /* bad argument */
#define RT_ERR_HIP -2       /* a HIP runtime call failed */
#define RT_ERR_STATE -3     /* call order: scene/camera not set */
#define RT_ERR_NO_DEVICE -4 /* no usable gfx950 device: the library never falls back to the CPU */

/* Raytracer::new (raytracing.rs:16-156): allocates the W x H accumulation image (zero-initialised,
 * as wgpu zero-initialises color_buffer/color_cache, raytracing.rs:20-52) on HIP device `device`.
 * `flags` = RT_ACCUM_* | RT_DISPATCH_* | RT_SHADE_* | RT_MATH_* | RT_COMBINE_* (rt_types.h). */
int rt_create(rt_ctx** out, int device, uint32_t width, uint32_t height, uint32_t flags);

/* drop(Raytracer) */
void rt_destroy(rt_ctx* ctx);

/* World::new buffer uploads (world.rs:46-56): `n_spheres` x 32 B and `n_materials` x 32 B host
 * images in the reference's Pod layouts. Copied; the caller keeps ownership. Every
 * material_index must be < n_materials (the reference's random_bunch can emit out-of-range
 * indices, objects.rs:37, whose effect is driver-defined; here it is RT_ERR_INVALID). Synchronous; cost grows with
 * the scene: besides the uploads it derives the per-sphere constants of the hit test and, for scenes from about 300
 * spheres where it pays, a uniform grid over the spheres (host side, double precision; a few milliseconds at 16 384
 * spheres) that bounce rays walk instead of testing every sphere - same image bits either way (RT_GRID=0 in the
 * environment switches the grid off). */
int rt_set_scene(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_material* materials,
                 uint32_t n_materials);

/* queue.write_buffer(camera) (camera.rs:264,285-287): the 160-byte CameraRaw, carrying
 * samples_per_pixel, frames_to_render and current_frame. */
int rt_set_camera(rt_ctx* ctx, const rt_camera_raw* camera);

/* Time::add_delta's write of the time uniform (time.rs:75), passed as the f32's bit pattern:
 * the kernel only ever uses the bits (raytrace_kernal.wgsl:260). */
int rt_set_time(rt_ctx* ctx, uint32_t time_bits);

/* Raytracer::compute (raytracing.rs:178-200): one compute pass with the current camera and time —
 * trace samples_per_pixel paths per covered pixel, fold into the accumulation image; the
 * color_buffer -> color_cache copy is implicit (one in-place image). Asynchronous. Does nothing
 * when camera.current_frame >= camera.frames_to_render (raytrace_kernal.wgsl:76).
 * Like the reference's compute, which records the dispatch into a command encoder that App::render submits
 * (app.rs:106-111), rt_compute RECORDS the pass (camera, time). Recorded passes run - as one batched launch of up to
 * 64 passes (RT_DEFER_FRAMES; at >= 4 samples per pixel no more than the staging buffer holds - 16 B per pixel OF THIS
 * CONTEXT'S PARTITION and pass, allocated on first use, at most 4 GiB and never more than a quarter of the device memory that
 * is free when the context first asks: 32 passes of a whole 4K image, 64 of an eighth of it), bit-identical to
 * one launch per pass - when that many wait or at the next call that observes the image, the
 * counters or the timing (rt_sync, rt_read_*, rt_get_counters, rt_timing_get, ...) or that changes what a recorded
 * pass depends on (scene, camera other than current_frame, partition, stream, bound image, rt_clear, rt_write_image);
 * a launch failure of a recorded pass is reported by THAT call (rt_flush / rt_sync are the calls to check when the
 * status matters; rt_destroy runs pending passes and drops their status). RT_DEFER_FRAMES=1 in the environment (read
 * by rt_create) launches every pass at once. Passes run under rt_enable_counters, and passes of a context that runs
 * on a caller's stream (rt_set_stream), launch at once, one by one: there rt_compute enqueues the kernel before it
 * returns. */
int rt_compute(rt_ctx* ctx);

/* Batched form of the App::update/App::render loop (app.rs:93-116) for `n_frames` redraws in ONE
 * launch: frame k uses time_bits[k] and current_frame = camera.current_frame + k; frames at or past
 * frames_to_render are skipped as the reference skips them. On return the context's camera has
 * current_frame advanced by n_frames and its time is time_bits[n_frames-1]. Results are
 * bit-identical to n_frames x (rt_set_time, rt_compute, current_frame += 1). Asynchronous. */
int rt_render_frames(rt_ctx* ctx, const uint32_t* time_bits, uint32_t n_frames);

/* rt_render_frames for callers that cannot allocate (no counterpart in the reference; for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc): the same n_frames redraws with the same results,
 * bit for bit, but this call performs no hipMalloc / hipFree / hipHostMalloc / event creation on any path. The device
 * scratch a launch needs beyond its inputs and outputs (tile queue head, frame table, sky-tile flags) lives in a
 * 2.1 MB array in the library's device image, one copy per GPU, held until the process ends. Limits, rejected with
 * RT_ERR_INVALID and a message: n_frames <= 256; at most 524 288 8x8 tiles in the context's partition (an 8K image has
 * 518 400). RT_ERR_STATE while passes recorded by rt_compute are pending (call rt_sync first). Every call on a GPU
 * shares that GPU's array and rewrites what it uses. Calls from different streams (two contexts of one GPU, each on its own
 * stream) are ordered on the device by the library: each call records one per-GPU event behind its kernels and a call from
 * another stream waits for it (hipStreamWaitEvent; no host wait) - they run one after the other, never overlapped. The
 * exception is a stream under capture, which cannot wait for an event recorded outside its graph: a captured call is not
 * ordered against other users of that GPU's scratch, and neither is a replay of its graph - the caller keeps them apart.
 * The frame table is built on the device from the time bits, which travel as kernel arguments: a captured call (any
 * n_frames <= 256) replays with the seeds and weights it was captured with. Not used here: staged (tile, frame) batches,
 * the cost-ordered tile queue, the per-tile candidate lists of scenes with more than 64 spheres and per-launch timing
 * events, all of which need buffers of their own - expect a few per cent less throughput than rt_render_frames on
 * small scenes and the round-1 loop shape (about 25 % less) on scenes of more than 64 spheres. Asynchronous like
 * rt_render_frames. */
int rt_render_frames_static(rt_ctx* ctx, const uint32_t* time_bits, uint32_t n_frames);

/* queue.submit without the wait (app.rs:111): launches the passes rt_compute has recorded and returns; a launch
 * failure of a recorded pass is reported here (or by whichever observing call comes first, see rt_compute). Does not
 * block, except that the first launch of a scene / image size may allocate its internal buffers, which synchronises
 * the stream once. Stream-ordered callers that keep the context's own stream call this before they record an event or
 * read the bound image from another stream. */
int rt_flush(rt_ctx* ctx);

/* queue.submit + device wait (app.rs:111). */
int rt_sync(rt_ctx* ctx);

/* What Screen::render would sample (color_cache, screen.rs:91-93): W*H*4 bytes RGBA8. In
 * RT_ACCUM_F32 mode the float image is quantised on the fly with the rgba8unorm rule. Synchronous. */
int rt_read_rgba8(rt_ctx* ctx, void* dst, size_t bytes);

/* The float4 accumulation image, W*H*16 bytes (RT_ACCUM_F32 contexts only). Synchronous. */
int rt_read_accum_f32(rt_ctx* ctx, void* dst, size_t bytes);

/* Checkpoint / resume (no counterpart in the reference, whose only persistent state is color_cache +
 * current_frame + time): upload a previously read image in the context's own format (RGBA8: W*H*4
 * bytes as read by rt_read_rgba8; F32: W*H*16 bytes as read by rt_read_accum_f32). Together with
 * camera.current_frame and the time bits this resumes a render bit-identically. Synchronous. */
int rt_write_image(rt_ctx* ctx, const void* src, size_t bytes);

/* Test hook: final RNG state (u32) of every pixel after the most recent pass, W*H*4 bytes; pixels
 * no pass covered read 0. Recording is off until rt_enable_rng_capture(ctx, 1). Synchronous. */
int rt_enable_rng_capture(rt_ctx* ctx, int enable);
int rt_read_rng_state(rt_ctx* ctx, void* dst, size_t bytes);

/* Event counters (rays, sphere tests, ... rt_counters in rt_types.h) accumulated by passes run
 * while counting is enabled; they feed rays/s and the algorithmic-FLOP roofline figure. Counting
 * launches an instrumented kernel variant, so leave it off in timed runs.
 *   enable = 1: the events of the REFERENCE ALGORITHM (every tile traced, every sphere tested by every ray: the oracle's
 *               counters; SURVEY 8d's FLOP formula is defined over these);
 *   enable = 2: the events the production launch EXECUTES - the same loop shape, tile masks / lists, own-sphere drop and
 *               behind cull as a non-counting launch. T = phase-1 filter tests run (8 FMAs each), slots_trace = exact
 *               hit_sphere tests run, D = of those with a discriminant >= 0, R = rays traced (sky-tile samples have none),
 *               slots_reject = samples finished in sky tiles (their N, M, P and rand_calls are included). RT_ERR_STATE at
 *               the launch when the scene has no SPLIT shape (33-64 spheres, RT_SPLIT=0 or RT_SKY_TILES=0). */
int rt_enable_counters(rt_ctx* ctx, int enable);
int rt_get_counters(rt_ctx* ctx, rt_counters* out); /* synchronous */
int rt_reset_counters(rt_ctx* ctx);

/* Multi-GPU tile split (no counterpart in the reference, which drives one device): this context
 * renders only the 8x8 pixel tiles t, numbered row-major over the covered area, with
 * t % world == rank. Pixels of other ranks' tiles stay zero, so the per-GPU images combine with a
 * plain sum (RCCL reduce). Default rank 0 of 1. */
int rt_set_partition(rt_ctx* ctx, uint32_t rank, uint32_t world);

/* Sky tiles (no counterpart in the reference; diagnostic): how many of this context's 8x8 tiles the last pass
 * finished without tracing because no primary ray of theirs can hit a sphere ("sky tiles"), out of
 * `tiles` tiles of its partition. Both 0 before the first pass or with RT_SKY_TILES=0 in the environment. */
int rt_sky_tile_stats(rt_ctx* ctx, uint32_t* sky_tiles, uint32_t* tiles);

/* Plumbing for callers that own device memory / streams (torch.distributed, HIP graphs):
 * rt_accum_device_ptr: device address and byte size of the accumulation image (RGBA8: W*H*4,
 *   F32: W*H*16), valid until rt_destroy. Launches recorded passes first (like rt_flush), so work the caller
 *   enqueues AFTER this call on the context's stream - or after rt_sync on any stream - sees the finished image.
 * rt_set_stream: run on the caller's hipStream_t (NULL = the context's own stream). While a caller's stream is set,
 *   rt_compute launches at once instead of recording (no deferral), so stream order is what the caller sees.
 *   Under hipStreamBeginCapture use rt_render_frames_static after one warm-up call: it never allocates or
 *   synchronises; the other launch paths may do both the first time a buffer is needed.
 * Rule for a bound image (rt_bind_accum) on the context's OWN stream: the caller's streams are not ordered against
 *   it; call rt_sync (or rt_flush + an event of your own) before another stream reads the image. */
int rt_accum_device_ptr(rt_ctx* ctx, void** dptr, size_t* bytes);
/* rt_bind_accum: use caller-owned device memory (e.g. a torch tensor handed to RCCL) as the
 *   accumulation image from now on; `bytes` must equal the image size and the memory must stay
 *   valid until rt_destroy or the next bind. The contents are used as they are (not cleared).
 *   dptr = NULL returns to the context's own image. */
int rt_bind_accum(rt_ctx* ctx, void* dptr, size_t bytes);
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
/* zero the accumulation image and RNG capture (a fresh Raytracer::new without reallocating) */
int rt_clear(rt_ctx* ctx);

/* Timing hook: average device time in milliseconds of the trace kernel over the launches made
 * since the last rt_timing_reset, measured with hipEvents recorded on the launch stream around
 * every launch while timing is enabled. Synchronous. */
int rt_enable_timing(rt_ctx* ctx, int enable);
int rt_timing_reset(rt_ctx* ctx);
int rt_timing_get(rt_ctx* ctx, double* total_ms, uint32_t* launches);

/* Device facts for reports: name, compute units, clock (kHz). */
int rt_device_info(rt_ctx* ctx, char* name, size_t name_bytes, int* compute_units, int* clock_khz);

/* Thread-local message for the last non-zero status returned on this thread. */
const char* rt_last_error(void);

/* Library version string. */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
