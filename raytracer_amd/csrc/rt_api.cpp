// rt_api.cpp — the C ABI of librt_hip.so (include/rt_hip.h): context, scene re-layout, launches.
// Replaces the wgpu half of the reference's Raytracer (src/raytracing.rs:15-201). There is no CPU
// fallback anywhere in this library: without a gfx950 device rt_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_hip.h"
#include "rt_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define RT_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                \
  } while (0)

// ---- scratch of rt_render_frames_static (include/rt_hip.h): the device memory a launch needs beyond its inputs and
// outputs, in a file-scope array instead of hipMalloc'ed buffers. Layout (bytes):
//   [0, 256)                     tile queue head of the persistent kernel (one u32, zeroed before every launch)
//   [256, 256 + 8 * MAX_FRAMES)  frame table {seed_t, bits of cw} of a batched launch
//   [.., .. + 4 * MAX_TILES)     sky-tile flags of the calling context's partition (rewritten by every call)
// Every GPU the library runs on holds its own copy (2.1 MB) until the process ends. `rt_static_scratch` is a host-side
// stand-in, NOT a device address: host code uses the per-device pointer hipGetSymbolAddress returned (scratch_ptr()).
constexpr uint32_t RT_STATIC_MAX_FRAMES = 256;
constexpr uint32_t RT_STATIC_MAX_TILES = 1u << 19;  // 8x8 tiles of one context's partition (7680x4320 has 518 400)
constexpr size_t RT_STATIC_OFF_TABLE = 256;
constexpr size_t RT_STATIC_OFF_SKY = RT_STATIC_OFF_TABLE + 8u * RT_STATIC_MAX_FRAMES;
constexpr size_t RT_STATIC_BYTES = RT_STATIC_OFF_SKY + 4u * (size_t)RT_STATIC_MAX_TILES;
}  // namespace

static __device__ __align__(256) unsigned char rt_static_scratch[RT_STATIC_BYTES];

namespace {

constexpr int RT_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[RT_MAX_DEVICES] = {};
// Every context of a GPU shares that GPU's scratch, and by default every context has a stream of its own: the users of the
// scratch are ordered ON THE DEVICE by one event per GPU - recorded behind each static launch, waited for (hipStreamWaitEvent,
// no host wait, no allocation) by the next static launch that comes from a different stream.
hipEvent_t g_scratch_event[RT_MAX_DEVICES] = {};
hipStream_t g_scratch_last_stream[RT_MAX_DEVICES] = {};
bool g_scratch_used[RT_MAX_DEVICES] = {};

// Device address of this device's copy of the scratch array; resolved once per device, with that device current
// (rt_create calls this right after hipSetDevice).
int resolve_scratch(int device) {
  if (device < 0 || device >= RT_MAX_DEVICES) return fail(RT_ERR_INVALID, "device index beyond RT_MAX_DEVICES");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return RT_OK;
  void* p = nullptr;
  RT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(rt_static_scratch)));
  if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress(rt_static_scratch) returned null");
  RT_HIP(hipEventCreateWithFlags(&g_scratch_event[device], hipEventDisableTiming));
  g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  return RT_OK;
}

}  // namespace

struct rt_ctx {
  int device = 0;
  uint32_t width = 0, height = 0, flags = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  void* accum = nullptr;      // image in use (own_accum or caller-bound memory)
  void* own_accum = nullptr;
  size_t accum_bytes = 0;
  uint32_t* rng = nullptr;  // allocated on first rt_enable_rng_capture
  bool rng_capture = false;
  unsigned long long* counters = nullptr;
  uint32_t* queue = nullptr;  // tile queue head of the persistent kernel
  // cost-ordered tile queue: per-tile cost of the last launch and the order derived from it
  uint32_t* tile_cost = nullptr;
  uint32_t* tile_order = nullptr;
  uint32_t order_tiles = 0;   // my_tiles the buffers are sized for
  uint32_t order_key = 0;     // rank/world the order was measured under
  bool have_order = false;
  uint32_t order_age = 0;     // launches since the cost was last measured
  int use_order = 1;          // env RT_TILE_ORDER=0 disables the feedback
  // sky tiles: per-tile "no primary ray can hit a sphere" flags for the current scene, camera and partition
  uint32_t* sky_flags = nullptr;   // tile masks (0 = sky tile)
  uint16_t* tile_lists = nullptr;  // scenes of > 32 spheres: 16 x u16 per tile (rt_kargs::tile_lists), sized like sky_flags
  uint32_t sky_tiles_cap = 0;  // my_tiles the buffer is sized for
  uint32_t sky_key = 0;        // rank/world the flags were computed under
  bool sky_valid = false;
  int use_sky = 1;             // env RT_SKY_TILES=0 traces every tile (no tile masks at all)
  int use_split = 1;           // env RT_SPLIT=0 keeps scenes of <= 32 spheres on the one-trace-per-iteration kernel
  int compute_units = 0;
  int waves_per_cu = 0;       // launch width override (0 = default); env RT_WAVES_PER_CU
  int reject_tries = 0;       // rejection rounds override (0 = default); env RT_REJECT_TRIES
  int batch_lanes = 0;        // parked-lane batch override (0 = default); env RT_BATCH_LANES
  int counting = 0;           // 0 off, 1 the reference algorithm's events (classic shape, every tile traced), 2 the events the production shape executes
  float* sph_p1 = nullptr;
  rt_f32x4 *sph_test = nullptr, *sph_shade = nullptr, *sph_col = nullptr;
  // uniform grid over the spheres of a big scene (rt_kargs::grid_*); grid_cells == nullptr: none
  uint32_t* grid_cells = nullptr;  // the grid's one buffer (build_grid)
  uint32_t grid_cell0 = 0;
  float grid_org[3] = {0, 0, 0}, grid_inv[3] = {0, 0, 0}, grid_o2max = 0;
  uint32_t grid_dim[3] = {0, 0, 0};
  int use_grid = 1;            // env RT_GRID=0: the general trace always loops over all spheres; 2: a grid whenever one can be built
  bool p1_tame = false;  // every sphere within the magnitude range of the phase-1 error bound
  uint32_t n_spheres = 0;
  bool have_scene = false, have_camera = false, have_time = false;
  rt_camera_raw camera{};
  uint32_t time_bits = 0;
  uint32_t rank = 0, world = 1;
  // time table for batched launches: pinned staging + device copy, guarded by an event
  uint2* time_host = nullptr;
  uint2* time_dev = nullptr;
  size_t time_cap = 0;
  hipEvent_t time_copied = nullptr;
  bool time_pending = false;
  // RGBA8 view of a float image
  void* quant = nullptr;
  // staged launches: frame colours of a batch (n_frames x width x height float4)
  rt_f32x4* staging = nullptr;
  size_t staging_bytes = 0;
  size_t stage_cap = 0;       // largest staging buffer this context allocates (stage_limit(); 0 = not asked yet)
  int use_staging = 1;        // env RT_STAGED_FRAMES=0 keeps batched launches pixel-sequential
  int queue_static = -1;      // env RT_QUEUE_STATIC (experiments)
  int queue_chunk = 0;        // env RT_QUEUE_CHUNK (experiments): units a wave takes per atomic on the queue head (0 = the rule in launch())
  int stage_min_spp = 0;      // env RT_STAGE_MIN_SPP (experiments): samples per pixel from which batches may be staged (0 = the rule below)
  int frame_group = 0;        // env RT_FRAME_GROUP: largest frame group of a staged launch (0 = RT_FRAME_GROUP_MAX)
  // deferred passes: rt_compute records (frame, time bits); the passes run as ONE batched launch when `defer_max` of
  // them wait or when any call observes or changes what they depend on (flush_pending)
  std::vector<uint32_t> pend_frames, pend_time;
  int defer_max = 64;         // env RT_DEFER_FRAMES (<= 1: every rt_compute launches at once); 8 -> 64: each launch ends on a tail of
                              // one pixel's work per lane (0.4 ms at 4K / 64 spp), 0.7 % of an 8-pass launch on one GPU but 6 % on an eighth of the tiles
  // per-launch timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
};

namespace {

int check(rt_ctx* ctx) {
  if (!ctx) return fail(RT_ERR_INVALID, "null context");
  RT_HIP(hipSetDevice(ctx->device));
  return RT_OK;
}

void covered(const rt_ctx* c, uint32_t* cov_w, uint32_t* cov_h) {
  // raytracing.rs:192: dispatch_workgroups(width/16, height/16, 1) of 16x16 invocations
  if (c->flags & RT_DISPATCH_FULL) {
    *cov_w = c->width;
    *cov_h = c->height;
  } else {
    *cov_w = (c->width / RT_WORKGROUP) * RT_WORKGROUP;
    *cov_h = (c->height / RT_WORKGROUP) * RT_WORKGROUP;
  }
}

constexpr uint32_t RT_ORDER_REFRESH = 32;  // launches between two measurements of the per-tile cost
constexpr uint64_t RT_STAGE_BELOW = 32;    // batches are staged per frame when a rank has fewer tiles per wave than this
constexpr uint32_t RT_STAGE_MIN_SPP = 4;   // below that a (tile, frame) unit is too little work for its hand-out (C1: 1 spp got slower) - unless frames are grouped:
constexpr uint64_t RT_STAGE_LOW_SPP_BELOW = 2;  // ... a rank with fewer than 2 spp + 2 tiles per wave (a small image: waves wait more than they issue) stages at 1-3 spp too,
                                                // in units of (tile, <= 16 / spp frames): C1 1.70 -> 0.68 ms per 64-frame render (tools/c1_probe.py, profiles/r3_c1_groups.log); 720p at
                                                // 1 spp +9 %, 1080p at 2 spp +9 %; not 1080p at 1 spp (-5 %: 16 B per sample through the staging buffer and the combine kernel)
constexpr size_t RT_STAGE_MAX_BYTES = (size_t)4 << 30;  // largest staging buffer a context allocates
constexpr uint32_t RT_FRAME_GROUP_MAX = 2u;     // staged launches: frames per unit of work (env RT_FRAME_GROUP overrides; 1 = a unit per frame). C2: 1 / 2 / 4 = 56.4 / 58.1 / 57.8 Gsamples/s, C4 (dear pixels: tails): 5.23 / 5.22 / 4.85 (profiles/r3_sweep_groups.log)
constexpr uint64_t RT_GROUP_MIN_UNITS = 16u;    // ... while every resident wave keeps at least this many units
constexpr uint32_t RT_GROUP_MAX_SAMPLES = 32u;  // ... and a lane's share of a unit stays at or below this many samples

// Largest staging buffer a context may allocate: RT_STAGE_MAX_BYTES, and never more than a quarter of the device memory
// that is free right now (a plain rt_compute loop must not take gigabytes inside an observer call unasked).
size_t stage_limit(rt_ctx* ctx) {
  if (ctx->stage_cap == 0) {  // asked once per context
    size_t free_b = 0, total_b = 0;
    ctx->stage_cap = RT_STAGE_MAX_BYTES;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      if (free_b / 4u < ctx->stage_cap) ctx->stage_cap = free_b / 4u;
    } else {
      (void)hipGetLastError();
    }
  }
  return ctx->stage_cap;
}

float frame_cw(uint32_t frame) {
  // combine_pixel_cache_color (wgsl:107-109): cw = f32(frame) / f32(frame + 1), one IEEE division
  const volatile float f = (float)frame, n = (float)(frame + 1u);
  return f / n;
}

// `scratch` != null: the launch of rt_render_frames_static - queue head and sky flags live in the static scratch array
// (time_table_dev already points into it), and nothing here may allocate: no staging buffer, no cost-ordered queue
// (its two arrays), no timing events.
int launch(rt_ctx* ctx, uint32_t first_frame, uint32_t n_frames, uint32_t time_bits0, const uint2* time_table_dev,
           unsigned char* scratch = nullptr) {
  rt_kargs a{};
  std::memcpy(a.p2w, ctx->camera.pixel_to_world, sizeof a.p2w);
  a.pos[0] = ctx->camera.pos[0];
  a.pos[1] = ctx->camera.pos[1];
  a.pos[2] = ctx->camera.pos[2];
  a.width = ctx->width;
  a.height = ctx->height;
  covered(ctx, &a.cov_w, &a.cov_h);
  a.tiles_x = (a.cov_w + 7u) / 8u;
  a.n_tiles = a.tiles_x * ((a.cov_h + 7u) / 8u);
  a.rank = ctx->rank;
  a.world = ctx->world;
  a.spp = ctx->camera.samples_per_pixel;
  a.first_frame = first_frame;
  a.n_frames = n_frames;
  a.seed_t0 = time_bits0 * 87189u + 189273u;
  a.cw0 = frame_cw(first_frame);
  a.frame_table = time_table_dev;
  a.inv_spp = (a.spp & (a.spp - 1u)) == 0u ? 1.0f / (float)a.spp : 0.0f;
  a.batch_lanes = ctx->batch_lanes > 0 ? (uint32_t)ctx->batch_lanes : 3u;  // profiles/r1b_sweep_batch_lanes.log (short units: below)
  a.n_spheres = ctx->n_spheres;
  a.flags = ctx->flags;
  a.p1_tame = ctx->p1_tame ? 1u : 0u;
  if (ctx->p1_tame && std::fmaf(a.pos[2], a.pos[2], std::fmaf(a.pos[1], a.pos[1], a.pos[0] * a.pos[0])) <= 0x1p40f) a.p1_tame |= 2u;
  a.sph_p1 = ctx->sph_p1;
  a.sph_test = ctx->sph_test;
  a.sph_shade = ctx->sph_shade;
  a.sph_col = ctx->sph_col;
  if (ctx->use_grid && ctx->grid_cells) {
    a.grid_cells = ctx->grid_cells; a.grid_cell0 = ctx->grid_cell0;
    for (int i = 0; i < 3; ++i) { a.grid_org[i] = ctx->grid_org[i]; a.grid_inv[i] = ctx->grid_inv[i]; a.grid_dim[i] = ctx->grid_dim[i]; }
    a.grid_ncells = ctx->grid_dim[0] * ctx->grid_dim[1] * ctx->grid_dim[2];
    a.grid_o2max = ctx->grid_o2max;
  }
  a.accum = ctx->accum;
  a.rng_out = ctx->rng_capture ? ctx->rng : nullptr;
  a.counters = ctx->counters;
  a.queue = scratch ? reinterpret_cast<uint32_t*>(scratch) : ctx->queue;
  if (a.n_tiles == 0) return RT_OK;
  const uint32_t my_tiles = a.n_tiles > a.rank ? (a.n_tiles - a.rank + a.world - 1u) / a.world : 0u;
  a.my_tiles = my_tiles;
  if (my_tiles == 0) return RT_OK;
  if (scratch && my_tiles > RT_STATIC_MAX_TILES)
    return fail(RT_ERR_INVALID, "rt_render_frames_static: " + std::to_string(my_tiles) + " tiles in this context's partition, the static scratch holds " +
                                    std::to_string(RT_STATIC_MAX_TILES) + " (use rt_render_frames)");
  // Rejection rounds per iteration: a lane that sits out pays one trace, so allow more rounds
  // the more a trace costs ("bounded rejection").
  const bool strict_math = (ctx->flags & RT_MATH_FAST) == 0;
  a.reject_tries = a.n_spheres <= 8u ? (strict_math ? 3u : 2u) : (a.n_spheres <= 64u ? 3u : 64u);
  if (ctx->reject_tries > 0) a.reject_tries = (uint32_t)ctx->reject_tries;
  // Persistent launch: enough single-wave workgroups to fill every SIMD, never more than tiles.
  const uint64_t resident = (uint64_t)ctx->compute_units * 24u;  // 6 waves per SIMD (<= 80 VGPRs): the measure of the staging rule
  // STAGED LAUNCH (batches only): a pixel is sequential, so with few tiles per wave the launch would end on
  // the waves that drew the dearest pixels - for all n_frames of them. When a rank has fewer than
  // RT_STAGE_BELOW tiles per wave (and at least RT_STAGE_MIN_SPP samples per pixel), the unit of work becomes (tile, frame): the frames' colours go to a
  // staging buffer and a second kernel folds them into the image in order (same arithmetic, bit-identical).
  a.n_units = my_tiles;
  a.frame_group = 1u;
  a.n_groups = n_frames;
  const size_t stage_bytes = (size_t)n_frames * my_tiles * 64u * sizeof(rt_f32x4);  // the rank's own tiles only
  // FRAME GROUPS: a unit is (tile, g consecutive frames), g a power of two: a lane keeps the pixel it is handed for g frames (one pixel
  // hand-out and 1/g tile fetch per g frame ends). g grows while a lane's share of a unit stays at or below RT_GROUP_MAX_SAMPLES
  // samples (at 64 spp a pixel hand-out is 1/64 of the work already and a longer unit only lengthens the launch's tail) and every
  // resident wave keeps enough units: RT_GROUP_MIN_UNITS of them - the staging rule exists because few units per wave mean long
  // tails - or 3 while a unit is at most 16 samples per lane (short units end evenly). Up to 2 frames, or 16 samples per lane at 1-4 spp.
  uint32_t g = 1u;
  {
    const bool forced = ctx->frame_group > 0;  // env RT_FRAME_GROUP: that group size (rounded down to a power of two), whatever the launch shape
    const uint32_t g_max = forced ? (uint32_t)ctx->frame_group : std::max(RT_FRAME_GROUP_MAX, 16u / std::max(a.spp, 1u));
    while (g * 2u <= g_max && g * 2u <= n_frames) {
      const uint64_t units2 = (uint64_t)my_tiles * ((n_frames + 2u * g - 1u) / (2u * g));
      const uint32_t samples2 = a.spp * g * 2u;
      // (an image of fewer tiles than resident waves - C1 - needs its frames split finely to fill the machine: 16 units per wave there too)
      if (!forced && !(samples2 <= RT_GROUP_MAX_SAMPLES && units2 >= (samples2 <= 16u && (uint64_t)my_tiles >= resident ? 3u : RT_GROUP_MIN_UNITS) * resident)) break;
      g *= 2u;
    }
  }
  const uint32_t min_spp = ctx->stage_min_spp > 0 ? (uint32_t)ctx->stage_min_spp : RT_STAGE_MIN_SPP;
  const bool enough_per_unit = a.spp >= min_spp || ((uint64_t)my_tiles < (RT_STAGE_LOW_SPP_BELOW * a.spp + 2u) * resident && a.spp * g >= std::min(min_spp, 2u));
  if (!scratch && ctx->use_staging && n_frames > 1 && time_table_dev && enough_per_unit && (uint64_t)my_tiles < RT_STAGE_BELOW * resident &&
      stage_bytes <= stage_limit(ctx) && (uint64_t)my_tiles * n_frames < 0xffffffffull / 64u) {
    if (ctx->staging_bytes < stage_bytes) {
      RT_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->staging);
      ctx->staging = nullptr;
      ctx->staging_bytes = 0;
      if (hipMalloc((void**)&ctx->staging, stage_bytes) == hipSuccess) {
        ctx->staging_bytes = stage_bytes;
      } else {
        (void)hipGetLastError();   // no room for the staging buffer: this context keeps its batches pixel-sequential
        ctx->staging = nullptr;
        ctx->use_staging = 0;
      }
    }
    if (ctx->staging) {
      a.staging = ctx->staging;
      a.frame_group = g;
      a.n_groups = (n_frames + g - 1u) / g;
      a.n_units = my_tiles * a.n_groups;
    }
  }
  // Short units (a staged launch whose lane finishes a unit within 32 samples: C2's 16 spp x 2 frames) bring a lane to the service section
  // often: serving six parked lanes at a time instead of three is worth 1.5 % there (profiles/r3_c2_batch_lanes.log); at 64 spp it costs 1 %.
  if (ctx->batch_lanes <= 0 && a.staging && a.n_spheres <= 64u && a.spp >= 8u && a.spp * a.frame_group <= 32u) a.batch_lanes = 6u;
  // HAND-OUT OF UNITS. The queue head serves about 86 M atomics per second - nothing to a launch of long units, the bound of a launch of
  // short ones: one 1080p pass at 1 spp (32 400 units of one sample per lane; what a host that presents every frame launches) took
  // 0.46 ms for 0.09 ms of work, one 8K pass 6.0 ms for 0.9 (profiles/r3_per_pass_matrix.md, r3_queue_chunk.log). By the samples s a lane
  // spends on a unit and the units u per resident wave:
  //   s <= 4: a wave takes c = min(64, u / 4) units per atomic when that is worth 16 samples per lane (8K at 1 spp, 4K at 4 spp); otherwise
  //           NO QUEUE (queue_chunk 0): wave w takes units w, w + waves, ... of the cost-ordered list - what dynamic hand-out could
  //           balance there is less than the atomics cost (640x360 ... 4K at 1 spp: 2.4 - 5.3 x faster, 1080p at 4 spp 2.1 x)
  //   s <= 8, u < 8 (C1's staged (tile, 8 frames) units: +20 %) and s <= 16, u < 4 (640x360 / 720p at 16 spp: +13 % / +4 %): no queue either
  //   longer or more units: one unit per atomic - chunks or strides of them only lengthen the launch's tail (1080p at 16 spp: no queue
  //           -9 %, 4K at 64 spp -28 %; C3, one pass per launch: 2 units per atomic -5 %)
  // Small scenes only: a sample of the 1024-sphere scene costs five times as much, its tiles differ more, and the queue is worth its atomics
  // (C4, one pass per launch: no queue -7 %). env RT_QUEUE_STATIC=n: no queue for units of at most n samples (0: never); RT_QUEUE_CHUNK=n: n units per atomic.
  a.queue_chunk = 1u;
  {
    const uint64_t s = (uint64_t)a.spp * (a.staging ? a.frame_group : n_frames), u = (uint64_t)a.n_units / resident;
    const uint64_t c = std::min<uint64_t>(64u, std::max<uint64_t>(1u, u / 4u));
    if (ctx->queue_static >= 0) { if (s <= (uint64_t)ctx->queue_static) a.queue_chunk = 0u; }
    else if (a.n_spheres <= 64u && ((s <= 4u && c * s < 16u) || (s <= 8u && u < 8u) || (s <= 16u && u < 4u))) a.queue_chunk = 0u;
    if (a.queue_chunk != 0u) {
      if (ctx->queue_chunk > 0) a.queue_chunk = (uint32_t)ctx->queue_chunk;
      else if (s <= 4u && a.n_spheres <= 64u) a.queue_chunk = (uint32_t)c;
    }
  }
  // Launch width: 6 waves per SIMD are resident (<= 80 VGPRs); the SMALL variants fit 7 (<= 72 VGPRs), which pays
  // when there are enough units of work to keep 28 waves per CU busy to the end (C3 -1.5 %, batched C2 -4 %; one C2
  // pass with 4.5 tiles per wave: +4 %, so not there)
  uint32_t wpc = 24u;
  if (ctx->counting == 0 && a.n_spheres <= 64u /* RT_SMALL_MAX */ && (uint64_t)a.n_units >= 8ull * 28ull * (uint64_t)ctx->compute_units) wpc = 28u;
  if (ctx->waves_per_cu > 0) wpc = (uint32_t)ctx->waves_per_cu;
  const uint64_t launch_waves = (uint64_t)ctx->compute_units * wpc;
  const uint32_t n_waves = (uint32_t)(launch_waves < a.n_units ? launch_waves : a.n_units);
  // host-side shape check before the launch: every tile index a block can form is < n_tiles or
  // rejected by the kernel's own guard; every pixel index is < width*height by construction of cov_*
  if (a.cov_w > a.width || a.cov_h > a.height) return fail(RT_ERR_INVALID, "covered area exceeds the image");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timing = ctx->timing && !scratch;
  if (timing) {  // the events bracket the path kernel alone (what rocprofv3 reports as its duration)
    if (ctx->events_used == ctx->events.size()) {
      RT_HIP(hipEventCreate(&e0));
      RT_HIP(hipEventCreate(&e1));
      ctx->events.emplace_back(e0, e1);
    }
    e0 = ctx->events[ctx->events_used].first;
    e1 = ctx->events[ctx->events_used].second;  // events_used is bumped once both are recorded (below)
  }
  const bool strict = (ctx->flags & RT_MATH_FAST) == 0;
  // Cost-ordered queue (only worth it when there are more tiles than waves): hand tiles out dearest
  // first, by the cost measured in an earlier launch of the same partition, scene and camera. The
  // measurement (one atomic per pixel) runs in the first launch and every RT_ORDER_REFRESH-th after.
  const bool ordering = !scratch && ctx->use_order && my_tiles > n_waves;
  bool measure = false;
  const uint32_t key = a.rank * 65536u + a.world;
  if (ordering) {
    if (ctx->order_tiles != my_tiles) {
      RT_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->tile_cost);
      (void)hipFree(ctx->tile_order);
      ctx->tile_cost = ctx->tile_order = nullptr;
      RT_HIP(hipMalloc((void**)&ctx->tile_cost, my_tiles * sizeof(uint32_t)));
      RT_HIP(hipMalloc((void**)&ctx->tile_order, my_tiles * sizeof(uint32_t)));
      ctx->order_tiles = my_tiles;
      ctx->have_order = false;
    }
    if (ctx->order_key != key) ctx->have_order = false;
    ctx->order_key = key;
    // (counting launches run other kernel shapes - the classic one traces the sky tiles too - and neither measure nor age the order)
    measure = ctx->counting == 0 && (!ctx->have_order || ++ctx->order_age >= RT_ORDER_REFRESH);
    if (measure) {
      RT_HIP(hipMemsetAsync(ctx->tile_cost, 0, my_tiles * sizeof(uint32_t), ctx->stream));
      a.tile_cost = ctx->tile_cost;
    }
    a.tile_order = ctx->have_order ? ctx->tile_order : nullptr;
  }
  // Sky tiles: recomputed whenever the scene, the camera or the partition changed - inside the caller's wall clock,
  // outside the hipEvents below, which bracket the path kernel alone (as rocprofv3 reports it); rt_sky_tiles_kernel
  // (15 us at 4K, once per scene/camera) and rt_combine_frames_kernel (0.25 ms per 8-pass 4K batch, 0.3 % of the path
  // kernel) are lines of their own in the rocprofv3 summaries under profiles/.
  // Counting launches trace every tile, so their event counters stay the oracle's.
  if (scratch && ctx->use_sky && ctx->counting != 1) {
    // the scratch is shared by every context on this GPU and keeps the last call's contents: the flags of this
    // context's my_tiles tiles are rewritten by every call (15 us at 4K)
    uint32_t* flags = reinterpret_cast<uint32_t*>(scratch + RT_STATIC_OFF_SKY);
    RT_HIP(rt_launch_tile_masks(a, flags, nullptr, ctx->stream));  // no room for candidate lists in the static scratch
    a.tile_masks = flags;
  } else if (ctx->use_sky && ctx->counting != 1) {
    if (ctx->sky_tiles_cap != my_tiles) {
      RT_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(ctx->sky_flags);
      (void)hipFree(ctx->tile_lists);
      ctx->sky_flags = nullptr;
      ctx->tile_lists = nullptr;
      ctx->sky_tiles_cap = 0;
      RT_HIP(hipMalloc((void**)&ctx->sky_flags, my_tiles * sizeof(uint32_t)));
      RT_HIP(hipMalloc((void**)&ctx->tile_lists, (size_t)my_tiles * 16u * sizeof(uint16_t)));
      ctx->sky_tiles_cap = my_tiles;
      ctx->sky_valid = false;
    }
    if (ctx->sky_key != key) ctx->sky_valid = false;
    ctx->sky_key = key;
    if (!ctx->sky_valid) {
      RT_HIP(rt_launch_tile_masks(a, ctx->sky_flags, ctx->tile_lists, ctx->stream));
      ctx->sky_valid = true;
    }
    a.tile_masks = ctx->sky_flags;
    if (a.n_spheres > 64u /* RT_SMALL_MAX */ && a.n_spheres <= 0xfff0u) a.tile_lists = ctx->tile_lists;
  }
  RT_HIP(hipMemsetAsync(a.queue, 0, sizeof(uint32_t), ctx->stream));
  if (timing) RT_HIP(hipEventRecord(e0, ctx->stream));
  // SPLIT variant (primary rays traced against their tile's candidate mask, primary + first bounce per iteration)
  const bool split = ctx->use_split && a.tile_masks && ctx->counting != 1 && (a.n_spheres <= 32u || a.tile_lists);
  if (ctx->counting == 2 && !split)
    return fail(RT_ERR_STATE, "rt_enable_counters(2): the executed-work counters exist for the SPLIT launch shapes only (<= 32 or > 64 spheres, "
                              "tile masks and RT_SPLIT on)");
  RT_HIP(rt_launch_path_kernel(a, n_waves, strict, ctx->counting != 0, split, ctx->stream));
  if (timing) {
    RT_HIP(hipEventRecord(e1, ctx->stream));
    ctx->events_used++;
  }
  if (a.staging) RT_HIP(rt_launch_combine_frames(a, ctx->stream));
  if (measure) {
    RT_HIP(rt_launch_tile_order(ctx->tile_cost, a.tile_masks, ctx->tile_order, my_tiles, ctx->stream));
    ctx->have_order = true;
    ctx->order_age = 0;
  }
  return RT_OK;
}

// n passes (frames[i] = CameraRaw.current_frame of pass i, time_bits[i] its time uniform) as one launch
int launch_batch(rt_ctx* ctx, const uint32_t* frames, const uint32_t* time_bits, uint32_t n) {
  if (n == 0) return RT_OK;
  if (n == 1) return launch(ctx, frames[0], 1, time_bits[0], nullptr);
  if (ctx->time_pending) {  // the staging buffer may still be in flight from the previous batch
    RT_HIP(hipEventSynchronize(ctx->time_copied));
    ctx->time_pending = false;
  }
  if (n > ctx->time_cap) {
    RT_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->time_dev);
    if (ctx->time_host) (void)hipHostFree(ctx->time_host);
    ctx->time_dev = nullptr;
    ctx->time_host = nullptr;
    ctx->time_cap = 0;
    size_t cap = 256;
    while (cap < n) cap *= 2;
    RT_HIP(hipMalloc((void**)&ctx->time_dev, cap * sizeof(uint2)));
    RT_HIP(hipHostMalloc((void**)&ctx->time_host, cap * sizeof(uint2), hipHostMallocDefault));
    ctx->time_cap = cap;
  }
  for (uint32_t i = 0; i < n; ++i) {
    const float cw = frame_cw(frames[i]);
    uint32_t cw_bits;
    std::memcpy(&cw_bits, &cw, 4);
    ctx->time_host[i] = make_uint2(time_bits[i] * 87189u + 189273u, cw_bits);
  }
  RT_HIP(hipMemcpyAsync(ctx->time_dev, ctx->time_host, n * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
  RT_HIP(hipEventRecord(ctx->time_copied, ctx->stream));
  ctx->time_pending = true;
  return launch(ctx, frames[0], n, time_bits[0], ctx->time_dev);
}

// launch_batch for rt_render_frames_static: the frame table is written into the static scratch by a one-workgroup kernel that
// takes the time bits BY VALUE (1 KiB of kernel arguments) - no pinned buffer, no event, no host wait: the call is the same
// sequence of stream operations whether it runs or is being captured, and a replayed graph carries its own table.
int launch_batch_static(rt_ctx* ctx, uint32_t first_frame, const uint32_t* time_bits, uint32_t n, unsigned char* scratch) {
  if (n == 0) return RT_OK;
  if (n == 1) return launch(ctx, first_frame, 1, time_bits[0], nullptr, scratch);
  uint2* table = reinterpret_cast<uint2*>(scratch + RT_STATIC_OFF_TABLE);
  RT_HIP(rt_launch_frame_table(table, time_bits, first_frame, n, ctx->stream));
  return launch(ctx, first_frame, n, time_bits[0], table, scratch);
}

// Launches the passes rt_compute has recorded. Called by every entry point that observes the image, the counters or
// the timing, or that changes something a recorded pass depends on; an error of the launch is that call's error.
int flush_pending(rt_ctx* ctx) {
  if (ctx->pend_frames.empty()) return RT_OK;
  std::vector<uint32_t> frames, bits;
  frames.swap(ctx->pend_frames);
  bits.swap(ctx->pend_time);
  return launch_batch(ctx, frames.data(), bits.data(), (uint32_t)frames.size());
}

#define RT_FLUSH(ctx)                          \
  do {                                         \
    if (int rc__ = flush_pending(ctx)) return rc__; \
  } while (0)

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
const char* rt_version(void) { return "raytracer_amd 0.1 (gfx950)"; }

int rt_create(rt_ctx** out, int device, uint32_t width, uint32_t height, uint32_t flags) {
  if (!out) return fail(RT_ERR_INVALID, "out is null");
  *out = nullptr;
  if (width == 0 || height == 0 || width > 65536u || height > 65536u) return fail(RT_ERR_INVALID, "bad image size");
  const uint32_t known = RT_ACCUM_F32 | RT_DISPATCH_FULL | RT_SHADE_CHECKER | RT_MATH_FAST | RT_COMBINE_SUM;
  if (flags & ~known) return fail(RT_ERR_INVALID, "unknown flag bits");
  if ((flags & RT_COMBINE_SUM) && !(flags & RT_ACCUM_F32))
    return fail(RT_ERR_INVALID, "RT_COMBINE_SUM needs RT_ACCUM_F32 (the quantised recurrence is order dependent)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device visible; librt_hip has no CPU path");
  if (device < 0 || device >= n_dev) return fail(RT_ERR_INVALID, "device index out of range");
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RT_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
  RT_HIP(hipSetDevice(device));
  if (int rc = resolve_scratch(device)) return rc;  // once per device, with that device current
  rt_ctx* c = new (std::nothrow) rt_ctx();
  if (!c) return fail(RT_ERR_INVALID, "out of host memory");
  c->device = device;
  c->width = width;
  c->height = height;
  c->flags = flags;
  const size_t px = (size_t)width * height;
  c->accum_bytes = px * ((flags & RT_ACCUM_F32) ? 16u : 4u);
  hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&c->own_accum, c->accum_bytes);
  c->accum = c->own_accum;
  if (e == hipSuccess) e = hipMalloc((void**)&c->counters, 14 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMalloc((void**)&c->queue, sizeof(uint32_t));
  c->compute_units = prop.multiProcessorCount;
  if (const char* e1 = std::getenv("RT_WAVES_PER_CU")) c->waves_per_cu = std::atoi(e1);   // tuning knobs for experiments
  if (const char* e2 = std::getenv("RT_REJECT_TRIES")) c->reject_tries = std::atoi(e2);
  if (const char* e3 = std::getenv("RT_BATCH_LANES")) c->batch_lanes = std::atoi(e3);
  if (const char* e4 = std::getenv("RT_TILE_ORDER")) c->use_order = std::atoi(e4);
  if (const char* e5 = std::getenv("RT_STAGED_FRAMES")) c->use_staging = std::atoi(e5);
  if (const char* e6 = std::getenv("RT_FRAME_GROUP")) c->frame_group = std::atoi(e6);
  if (const char* e6 = std::getenv("RT_STAGE_MIN_SPP")) c->stage_min_spp = std::atoi(e6);
  if (const char* e6 = std::getenv("RT_QUEUE_CHUNK")) c->queue_chunk = std::atoi(e6);
  if (const char* e6 = std::getenv("RT_QUEUE_STATIC")) c->queue_static = std::atoi(e6);
  if (const char* e6 = std::getenv("RT_SKY_TILES")) c->use_sky = std::atoi(e6);
  if (const char* e7 = std::getenv("RT_DEFER_FRAMES")) c->defer_max = std::atoi(e7);
  if (const char* e8 = std::getenv("RT_SPLIT")) c->use_split = std::atoi(e8);
  if (const char* e9 = std::getenv("RT_GRID")) c->use_grid = std::atoi(e9);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->time_copied, hipEventDisableTiming);
  if (e == hipSuccess) {
    c->stream = c->own_stream;
    e = hipMemsetAsync(c->accum, 0, c->accum_bytes, c->stream);
  }
  if (e == hipSuccess) e = hipMemsetAsync(c->counters, 0, 14 * sizeof(unsigned long long), c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) {
    const std::string msg = std::string("rt_create: ") + hipGetErrorString(e);
    rt_destroy(c);
    return fail(RT_ERR_HIP, msg);
  }
  *out = c;
  return RT_OK;
}

void rt_destroy(rt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) {
    (void)flush_pending(c);  // recorded passes run, as they did when rt_compute launched them itself
    (void)hipStreamSynchronize(c->stream);
  }
  for (auto& ev : c->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (c->time_copied) (void)hipEventDestroy(c->time_copied);
  (void)hipFree(c->own_accum);
  (void)hipFree(c->rng);
  (void)hipFree(c->counters);
  (void)hipFree(c->queue);
  (void)hipFree(c->tile_cost);
  (void)hipFree(c->tile_order);
  (void)hipFree(c->sky_flags);
  (void)hipFree(c->tile_lists);
  (void)hipFree(c->sph_p1);
  (void)hipFree(c->sph_test);
  (void)hipFree(c->sph_shade);
  (void)hipFree(c->sph_col);
  (void)hipFree(c->grid_cells);
  (void)hipFree(c->time_dev);
  (void)hipFree(c->quant);
  (void)hipFree(c->staging);
  if (c->time_host) (void)hipHostFree(c->time_host);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

// GRID over the spheres of a big scene (rt_kargs::grid_*, DESIGN.md section 4 "grid"). Everything in double; a sphere is listed in
// every cell its INFLATED ball touches. The inflation covers (a) what the literal arithmetic of hit_sphere can add to a sphere: its
// discriminant is non-negative only if the ray's line passes within sqrt(r^2 + 54 u S) of the centre, S = |o|^2 + |c|^2 + r^2,
// u = 2^-24 ("phase-1 filter"); rays with |o|^2 <= o2max use the grid, so S <= S_max = o2max + max(|c|^2 + r^2)
// and r^2 + 2^-18 S_max (64 u S_max) covers it; (b) the traversal's own rounding (cell coordinates in f32: below 2^-18 of a cell)
// with 2^-8 of a cell. No grid (the kernel loops over all spheres, as before) when the scene is degenerate, when big spheres would
// be listed nearly everywhere, or when the walk would not pay (below). Scenes of more than 65535 spheres have a grid too: their
// candidates are tested where they are met (the queue's entries are 16 bits).
static int build_grid(rt_ctx* ctx, const rt_sphere* sp, uint32_t n, const std::vector<float>& p1) {
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, cr2max = 0.0, corner2 = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    const double r = std::fabs((double)sp[i].radius);
    double cc = 0.0;
    for (int a = 0; a < 3; ++a) {
      const double c = sp[i].center[a];
      lo[a] = std::min(lo[a], c - r); hi[a] = std::max(hi[a], c + r);
      cc += c * c;
    }
    cr2max = std::max(cr2max, cc + r * r);
  }
  for (int a = 0; a < 3; ++a) { const double m = std::max(std::fabs(lo[a]), std::fabs(hi[a])); corner2 += m * m; }
  // ray origins the grid serves: within the scene's reach of the coordinate origin (bounce rays start on spheres; 10 % to spare)
  const double o2max = 1.1 * corner2 + 1.0;
  const double infl2 = 0x1p-18 * (o2max + cr2max);  // added to r^2: 64 u S_max against the 54 u S the literal discriminant can be off by
  double ext[3], vol = 1.0;
  std::vector<double> R(n);
  double rmax = 0.0;
  for (uint32_t i = 0; i < n; ++i) { const double r = sp[i].radius; R[i] = std::sqrt(r * r + infl2); rmax = std::max(rmax, R[i]); }
#ifdef RT_BREAK_GRID  // a deliberately unsound grid (spheres listed by 97 % of their radius): tools/fuzz_proofs.py --big must catch it
  for (uint32_t i = 0; i < n; ++i) R[i] *= 0.97;
#endif
  for (int a = 0; a < 3; ++a) { lo[a] -= rmax * 0.01 + 1e-3; hi[a] += rmax * 0.01 + 1e-3; }  // (the inflated balls are re-bounded below)
  for (uint32_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], sp[i].center[a] - R[i]); hi[a] = std::max(hi[a], sp[i].center[a] + R[i]); }
  for (int a = 0; a < 3; ++a) { ext[a] = hi[a] - lo[a]; if (!(ext[a] > 0.0) || !std::isfinite(ext[a])) return RT_OK; vol *= ext[a]; }
  // about four cells per sphere, no cell edge below the median inflated diameter's half, at most 64 cells per axis
  std::vector<double> Rs(R); std::nth_element(Rs.begin(), Rs.begin() + n / 2, Rs.end());
  double cells_per_sphere = 4.0;  // env RT_GRID_DENSITY (experiments; profiles/r3_grid_density.log: 1 / 2 / 4 / 8 / 16 = 0.744 / 0.654 / 0.623 / 0.631 / 0.678 ms per C4 pass)
  if (const char* e = std::getenv("RT_GRID_DENSITY")) { const double v = std::atof(e); if (v > 0.01 && v < 100.0) cells_per_sphere = v; }
  const double edge = std::max(std::cbrt(vol / (cells_per_sphere * n)), Rs[n / 2]);
  uint32_t dim[3]; double cell[3];
  for (int a = 0; a < 3; ++a) {
    dim[a] = (uint32_t)std::min(64.0, std::max(1.0, std::floor(ext[a] / edge)));
    cell[a] = ext[a] / dim[a];
  }
  const double slack = 0x1p-8 * std::max(cell[0], std::max(cell[1], cell[2]));
  // pad the box by the slack so that every inflated + slackened ball lies inside it
  for (int a = 0; a < 3; ++a) { lo[a] -= 2.0 * slack; hi[a] += 2.0 * slack; ext[a] = hi[a] - lo[a]; cell[a] = ext[a] / dim[a]; }
  const size_t ncells = (size_t)dim[0] * dim[1] * dim[2];
  std::vector<std::vector<uint32_t>> lists(ncells);
  size_t total = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const double Ri = R[i] + slack;
    int c0[3], c1[3];
    for (int a = 0; a < 3; ++a) {
      c0[a] = std::max(0, (int)std::floor((sp[i].center[a] - Ri - lo[a]) / cell[a]));
      c1[a] = std::min((int)dim[a] - 1, (int)std::floor((sp[i].center[a] + Ri - lo[a]) / cell[a]));
    }
    for (int z = c0[2]; z <= c1[2]; ++z)
      for (int y = c0[1]; y <= c1[1]; ++y)
        for (int x = c0[0]; x <= c1[0]; ++x) {
          const int ix[3] = {x, y, z};
          double d2 = 0.0;  // squared distance from the centre to the cell's box
          for (int a = 0; a < 3; ++a) {
            const double b0 = lo[a] + ix[a] * cell[a], b1 = b0 + cell[a], c = sp[i].center[a];
            const double d = c < b0 ? b0 - c : (c > b1 ? c - b1 : 0.0);
            d2 += d * d;
          }
          if (d2 <= Ri * Ri) { lists[((size_t)z * dim[1] + y) * dim[0] + x].push_back(i); ++total; }
        }
    if (total > 16u * (size_t)n + 4096u) {  // big spheres listed nearly everywhere: the loop over all spheres is the better trace
      if (std::getenv("RT_GRID_VERBOSE")) std::fprintf(stderr, "rt_set_scene: no grid (%u spheres: more than %zu list entries)\n", n, total);
      return RT_OK;
    }
  }
  // WHEN THE GRID PAYS (tools/grid_crossover.py, profiles/r3_grid_crossover.md: bunches of 65 ... 16384 unit spheres, both traces timed):
  // a wave walks until its longest ray is through - about 1.5 x the mean ray's half of dim0 + dim1 + dim2 cells, one trip of ~60
  // instructions per cell and per listed sphere - while the loop over all spheres costs it ~12 instructions per sphere with the exact
  // tests: the grid is used when trips < n / 5 (the measured crossover of the bunches lies between 200 and 300 spheres; a grid
  // of few cells that list every sphere never gets there). RT_GRID=2 builds it regardless (tests of the walk on small scenes).
  const double trips = 0.75 * (dim[0] + dim[1] + dim[2]) * (1.0 + (double)total / (double)ncells);
  if (ctx->use_grid != 2 && trips > 0.2 * n) {
    if (std::getenv("RT_GRID_VERBOSE")) std::fprintf(stderr, "rt_set_scene: no grid (%u spheres: about %.0f trips per wave through %u x %u x %u cells, %.1f entries per cell)\n", n,
                                                     trips, dim[0], dim[1], dim[2], (double)total / (double)ncells);
    return RT_OK;
  }
  // ONE buffer, addressed by 32-bit byte offsets from its base (a scalar base and a vector offset: no 64-bit address arithmetic
  // per lane). [0, 32): unused, so that offset 0 can mean "empty cell"; then the list entries, 32 bytes each: the filter's constants
  // {cx, cy, cz, K'} TRAVEL WITH the entry (one gather level less than index -> record), then the sphere's index with bit 31 set
  // on the last entry of its cell; then the cell table: per cell the byte offset of its first entry, 0 for an empty cell.
  const size_t cell0 = 32u + total * 32u, bytes = cell0 + ncells * 4u;
  if (bytes > 0x7fffffffu) return RT_OK;
  std::vector<uint32_t> buf(bytes / 4u, 0u);
  size_t at = 32u;
  for (size_t c = 0; c < ncells; ++c) {
    if (lists[c].empty()) continue;
    buf[(cell0 >> 2) + c] = (uint32_t)at;
    for (size_t e = 0; e < lists[c].size(); ++e, at += 32u) {
      const uint32_t i = lists[c][e];
      const float* pp = &p1[(size_t)(i >> 1) * 8u + (i & 1u)];
      const float rec[4] = {pp[0], pp[2], pp[4], pp[6]};
      std::memcpy(&buf[at >> 2], rec, 16);
      buf[(at >> 2) + 4u] = i | (e + 1 == lists[c].size() ? 0x80000000u : 0u);
    }
  }
  RT_HIP(hipMalloc((void**)&ctx->grid_cells, bytes));
  RT_HIP(hipMemcpy(ctx->grid_cells, buf.data(), bytes, hipMemcpyHostToDevice));
  ctx->grid_cell0 = (uint32_t)cell0;
  for (int a = 0; a < 3; ++a) { ctx->grid_org[a] = (float)lo[a]; ctx->grid_inv[a] = (float)(1.0 / cell[a]); ctx->grid_dim[a] = dim[a]; }
  if (std::getenv("RT_GRID_VERBOSE"))
    std::fprintf(stderr, "rt_set_scene: grid %u x %u x %u over %u spheres, %zu list entries (%.2f per cell, %.2f per sphere)\n", dim[0], dim[1], dim[2], n,
                 total, (double)total / (double)ncells, (double)total / n);
  ctx->grid_o2max = (float)o2max;
  return RT_OK;
}

int rt_set_scene(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_material* materials,
                 uint32_t n_materials) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!spheres || !materials || n_spheres == 0 || n_materials == 0) return fail(RT_ERR_INVALID, "empty scene");
  if (n_spheres > (1u << 24)) return fail(RT_ERR_INVALID, "too many spheres");
  // phase-1 constants in sphere pairs {cx0,cx1, cy0,cy1, cz0,cz1, K'0,K'1}; the scene is padded to a multiple
  // of four with spheres at the origin whose K' = 2^100 makes the filter value negative for every tame ray
  const uint32_t n_pad = (n_spheres + 3u) & ~3u;
  std::vector<float> p1((size_t)n_pad * 4u, 0.0f);
  for (uint32_t i = n_spheres; i < n_pad; ++i) p1[(size_t)(i >> 1) * 8u + (i & 1u) + 6u] = 0x1p100f;
  std::vector<rt_f32x4> test(n_spheres), shade(n_spheres), col(n_spheres);
  bool tame = true;
  for (uint32_t i = 0; i < n_spheres; ++i) {
    const rt_sphere& s = spheres[i];
    if (s.material_index >= n_materials)
      return fail(RT_ERR_INVALID, "sphere " + std::to_string(i) + ": material_index out of range");
    const rt_material& m = materials[s.material_index];
    // fl(r*r) is the same single rounding the shader performs per test (wgsl:210); hoisting it is bit-safe
    const volatile float r = s.radius;
    const float rr = r * r;
    // sph_col.w = +1/r when r is a power of two (then x / r == x * (1/r) exactly and the strict
    // kernel multiplies), -1/r otherwise (it divides); the fast kernel uses |sph_col.w|.
    int ex = 0;
    const bool pow2 = r > 0.0f && std::frexp((float)r, &ex) == 0.5f && ex > -100 && ex < 100;
    const float inv_r = pow2 ? 1.0f / r : -std::fabs(1.0f / r);
    // phase-1 filter constant, from exact (double) |c|^2 and r^2, one rounding to f32:
    // K' = (|c|^2 - r^2) - m (|c|^2 + r^2); the filter's error bound needs 2^-40 <= |c|^2 + r^2 <= 2^40
    const double cc = (double)s.center[0] * s.center[0] + (double)s.center[1] * s.center[1] + (double)s.center[2] * s.center[2];
    const double r2 = (double)s.radius * s.radius;
    if (!(cc + r2 >= 0x1p-40 && cc + r2 <= 0x1p40)) tame = false;  // also catches NaN / inf
    float* pp = &p1[(size_t)(i >> 1) * 8u + (i & 1u)];
    pp[0] = s.center[0]; pp[2] = s.center[1]; pp[4] = s.center[2];
    pp[6] = (float)((cc - r2) - (double)RT_P1_MARGIN * (cc + r2));
    test[i] = rt_f32x4{s.center[0], s.center[1], s.center[2], rr};
    shade[i] = rt_f32x4{s.center[0], s.center[1], s.center[2], s.radius};
    col[i] = rt_f32x4{m.color[0], m.color[1], m.color[2], inv_r};
  }
  RT_HIP(hipStreamSynchronize(ctx->stream));  // no launch may still read the old arrays
  (void)hipFree(ctx->sph_p1);
  (void)hipFree(ctx->sph_test);
  (void)hipFree(ctx->sph_shade);
  (void)hipFree(ctx->sph_col);
  ctx->sph_p1 = nullptr;
  ctx->sph_test = ctx->sph_shade = ctx->sph_col = nullptr;
  ctx->have_scene = false;
  const size_t bytes = (size_t)n_spheres * sizeof(rt_f32x4);
  RT_HIP(hipMalloc((void**)&ctx->sph_p1, p1.size() * sizeof(float)));
  RT_HIP(hipMalloc((void**)&ctx->sph_test, bytes));
  RT_HIP(hipMalloc((void**)&ctx->sph_shade, bytes));
  RT_HIP(hipMalloc((void**)&ctx->sph_col, bytes));
  RT_HIP(hipMemcpy(ctx->sph_p1, p1.data(), p1.size() * sizeof(float), hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(ctx->sph_test, test.data(), bytes, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(ctx->sph_shade, shade.data(), bytes, hipMemcpyHostToDevice));
  RT_HIP(hipMemcpy(ctx->sph_col, col.data(), bytes, hipMemcpyHostToDevice));
  (void)hipFree(ctx->grid_cells);
  ctx->grid_cells = nullptr;
  if (tame && n_spheres > 64u) {
    if (int rc = build_grid(ctx, spheres, n_spheres, p1)) return rc;
  }
  ctx->n_spheres = n_spheres;
  ctx->p1_tame = tame;
  ctx->have_order = false;  // per-tile cost belongs to the old scene
  ctx->sky_valid = false;
  ctx->have_scene = true;
  return RT_OK;
}

int rt_set_camera(rt_ctx* ctx, const rt_camera_raw* camera) {
  if (int rc = check(ctx)) return rc;
  if (!camera) return fail(RT_ERR_INVALID, "camera is null");
  if (camera->samples_per_pixel == 0)
    return fail(RT_ERR_INVALID, "samples_per_pixel must be >= 1 (the reference clamps it to [1,256], camera.rs:136-137)");
  // recorded passes share one camera: anything but current_frame changing ends the batch
  if (ctx->have_camera && !ctx->pend_frames.empty()) {
    rt_camera_raw a = ctx->camera, b = *camera;
    a.current_frame = b.current_frame = 0;
    if (std::memcmp(&a, &b, sizeof a) != 0) RT_FLUSH(ctx);
  }
  // a moved camera changes which tiles are dear: measure the per-tile cost again
  if (!ctx->have_camera || std::memcmp(ctx->camera.pixel_to_world, camera->pixel_to_world, sizeof camera->pixel_to_world) != 0 ||
      std::memcmp(ctx->camera.pos, camera->pos, sizeof camera->pos) != 0) {
    ctx->have_order = false;
    ctx->sky_valid = false;
  }
  ctx->camera = *camera;
  ctx->have_camera = true;
  return RT_OK;
}

int rt_set_time(rt_ctx* ctx, uint32_t time_bits) {
  if (int rc = check(ctx)) return rc;
  ctx->time_bits = time_bits;
  ctx->have_time = true;
  return RT_OK;
}

int rt_compute(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  if (!ctx->have_scene || !ctx->have_camera || !ctx->have_time)
    return fail(RT_ERR_STATE, "rt_compute before rt_set_scene / rt_set_camera / rt_set_time");
  if (ctx->camera.current_frame >= ctx->camera.frames_to_render) return RT_OK;  // wgsl:76
  // Like Raytracer::compute, which only records the dispatch (raytracing.rs:178-200; app.rs:111 submits): the pass is
  // recorded, and up to defer_max recorded passes run as one batched launch (staged (tile, frame) units where that
  // pays) - bit-identical to launching them one by one, without each launch's tail. Counting launches stay single.
  // On a caller's stream (rt_set_stream) every pass is enqueued at once: such callers order their own work after ours
  // on that stream (RCCL on the same stream, graph capture) and must find the kernel there.
  if (ctx->defer_max > 1 && ctx->counting == 0 && ctx->stream == ctx->own_stream) {
    ctx->pend_frames.push_back(ctx->camera.current_frame);
    ctx->pend_time.push_back(ctx->time_bits);
    // ... but, where batches are staged (launch(): >= RT_STAGE_MIN_SPP samples per pixel, few tiles per wave), never more than a staged batch has room
    // for (16 B per pixel and pass in the staging buffer): a batch over that limit would fall back to the pixel-sequential
    // shape and its long tail. Batches of fewer samples per pixel are pixel-sequential anyway, and the longer the better
    // (8K, 1 spp: 33 Gsamples/s in batches of 8, 57 in batches of 64).
    uint32_t cov_w = 0, cov_h = 0;
    covered(ctx, &cov_w, &cov_h);
    const uint64_t tiles = (uint64_t)((cov_w + 7u) / 8u) * ((cov_h + 7u) / 8u);
    const uint64_t my_tiles = tiles > ctx->rank ? (tiles - ctx->rank + ctx->world - 1u) / ctx->world : 0u;
    const size_t per_pass = (size_t)my_tiles * 64u * sizeof(rt_f32x4);  // the staging buffer holds this rank's own tiles
    const size_t room = stage_limit(ctx) / (per_pass ? per_pass : 1);
    // launch()'s staging rule: few tiles per resident wave and enough samples per pixel
    const bool staged = ctx->use_staging && ctx->camera.samples_per_pixel >= RT_STAGE_MIN_SPP &&
                        my_tiles < RT_STAGE_BELOW * (uint64_t)ctx->compute_units * 24u;
    const size_t limit = staged && room >= 2 && room < (size_t)ctx->defer_max ? room : (size_t)ctx->defer_max;
    if (ctx->pend_frames.size() >= limit) RT_FLUSH(ctx);
    return RT_OK;
  }
  return launch(ctx, ctx->camera.current_frame, 1, ctx->time_bits, nullptr);
}

int rt_render_frames(rt_ctx* ctx, const uint32_t* time_bits, uint32_t n_frames) {
  if (int rc = check(ctx)) return rc;
  if (!ctx->have_scene || !ctx->have_camera) return fail(RT_ERR_STATE, "rt_render_frames before rt_set_scene / rt_set_camera");
  if (!time_bits || n_frames == 0) return fail(RT_ERR_INVALID, "empty time table");
  const uint32_t cur = ctx->camera.current_frame, total = ctx->camera.frames_to_render;
  const uint32_t live = cur < total ? (n_frames < total - cur ? n_frames : total - cur) : 0u;  // rest are no-ops (wgsl:76)
  RT_FLUSH(ctx);
  int rc = RT_OK;
  if (live > 0) {
    std::vector<uint32_t> frames(live);
    for (uint32_t i = 0; i < live; ++i) frames[i] = cur + i;
    rc = launch_batch(ctx, frames.data(), time_bits, live);
  }
  if (rc != RT_OK) return rc;
  ctx->camera.current_frame = cur + n_frames;  // Camera::increase_frame once per redraw (camera.rs:253)
  ctx->time_bits = time_bits[n_frames - 1];
  ctx->have_time = true;
  return RT_OK;
}

int rt_render_frames_static(rt_ctx* ctx, const uint32_t* time_bits, uint32_t n_frames) {
  if (int rc = check(ctx)) return rc;
  if (!ctx->have_scene || !ctx->have_camera) return fail(RT_ERR_STATE, "rt_render_frames_static before rt_set_scene / rt_set_camera");
  if (!time_bits || n_frames == 0) return fail(RT_ERR_INVALID, "empty time table");
  if (n_frames > RT_STATIC_MAX_FRAMES)
    return fail(RT_ERR_INVALID, "rt_render_frames_static: " + std::to_string(n_frames) + " frames, the static scratch holds a table of " +
                                    std::to_string(RT_STATIC_MAX_FRAMES) + " (split the call or use rt_render_frames)");
  // recorded passes would be launched by the allocating path: the caller has to have them run first
  if (!ctx->pend_frames.empty())
    return fail(RT_ERR_STATE, "rt_render_frames_static: passes recorded by rt_compute are pending; call rt_sync first");
  unsigned char* scratch = g_scratch_ptr[ctx->device];  // this device's copy (resolve_scratch in rt_create)
  if (!scratch) return fail(RT_ERR_STATE, "rt_render_frames_static: scratch address of this device was never resolved");
  const uint32_t cur = ctx->camera.current_frame, total = ctx->camera.frames_to_render;
  const uint32_t live = cur < total ? (n_frames < total - cur ? n_frames : total - cur) : 0u;  // rest are no-ops (wgsl:76)
  if (live > 0) {
    // Order this call behind the scratch's last user when that was another stream (another context of this GPU). A stream
    // that is being captured cannot wait for an event recorded outside the capture: there the caller orders the graph
    // against other users of this GPU's scratch (rt_hip.h).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    RT_HIP(hipStreamIsCapturing(ctx->stream, &cap));
    const bool ordered = cap == hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);  // held over the enqueue: two host threads must not interleave their scratch users either
    if (ordered && g_scratch_used[ctx->device] && g_scratch_last_stream[ctx->device] != ctx->stream)
      RT_HIP(hipStreamWaitEvent(ctx->stream, g_scratch_event[ctx->device], 0));
    if (int rc = launch_batch_static(ctx, cur, time_bits, live, scratch)) return rc;
    if (ordered) {
      RT_HIP(hipEventRecord(g_scratch_event[ctx->device], ctx->stream));
      g_scratch_last_stream[ctx->device] = ctx->stream;
      g_scratch_used[ctx->device] = true;
    }
  }
  ctx->camera.current_frame = cur + n_frames;
  ctx->time_bits = time_bits[n_frames - 1];
  ctx->have_time = true;
  return RT_OK;
}

int rt_flush(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);  // launches, does not wait
  return RT_OK;
}

int rt_sync(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  RT_HIP(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_read_rgba8(rt_ctx* ctx, void* dst, size_t bytes) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  const size_t px = (size_t)ctx->width * ctx->height;
  if (!dst || bytes != px * 4u) return fail(RT_ERR_INVALID, "rt_read_rgba8: need width*height*4 bytes");
  const void* src = ctx->accum;
  if (ctx->flags & RT_ACCUM_F32) {
    if (!ctx->quant) RT_HIP(hipMalloc(&ctx->quant, px * 4u));
    RT_HIP(rt_launch_quantize(ctx->accum, ctx->quant, px, ctx->stream));
    src = ctx->quant;
  }
  RT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_read_accum_f32(rt_ctx* ctx, void* dst, size_t bytes) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!(ctx->flags & RT_ACCUM_F32)) return fail(RT_ERR_STATE, "rt_read_accum_f32 on an RT_ACCUM_RGBA8 context");
  if (!dst || bytes != ctx->accum_bytes) return fail(RT_ERR_INVALID, "rt_read_accum_f32: need width*height*16 bytes");
  RT_HIP(hipMemcpyAsync(dst, ctx->accum, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_write_image(rt_ctx* ctx, const void* src, size_t bytes) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!src || bytes != ctx->accum_bytes) return fail(RT_ERR_INVALID, "rt_write_image: need exactly the image size of this context");
  RT_HIP(hipMemcpyAsync(ctx->accum, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  RT_HIP(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_enable_rng_capture(rt_ctx* ctx, int enable) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (enable && !ctx->rng) {
    const size_t bytes = (size_t)ctx->width * ctx->height * sizeof(uint32_t);
    RT_HIP(hipMalloc((void**)&ctx->rng, bytes));
    RT_HIP(hipMemsetAsync(ctx->rng, 0, bytes, ctx->stream));
  }
  ctx->rng_capture = enable != 0;
  return RT_OK;
}

int rt_read_rng_state(rt_ctx* ctx, void* dst, size_t bytes) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!ctx->rng) return fail(RT_ERR_STATE, "rt_read_rng_state before rt_enable_rng_capture");
  if (!dst || bytes != (size_t)ctx->width * ctx->height * 4u) return fail(RT_ERR_INVALID, "rt_read_rng_state: need width*height*4 bytes");
  RT_HIP(hipMemcpyAsync(dst, ctx->rng, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_enable_counters(rt_ctx* ctx, int enable) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (enable < 0 || enable > 2) return fail(RT_ERR_INVALID, "rt_enable_counters: 0 = off, 1 = the reference algorithm's events, 2 = the events the production launch executes");
  ctx->counting = enable;
  return RT_OK;
}

int rt_get_counters(rt_ctx* ctx, rt_counters* out) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!out) return fail(RT_ERR_INVALID, "out is null");
  unsigned long long h[14];
  RT_HIP(hipMemcpyAsync(h, ctx->counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  RT_HIP(hipStreamSynchronize(ctx->stream));
  out->T = h[0]; out->D = h[1]; out->H = h[2]; out->R = h[3]; out->M = h[4]; out->S = h[5];
  out->Sm = h[6]; out->J = h[7]; out->Sd = h[8]; out->N = h[9]; out->P = h[10]; out->rand_calls = h[11];
  out->slots_trace = h[12]; out->slots_reject = h[13];
  return RT_OK;
}

int rt_reset_counters(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  RT_HIP(hipMemsetAsync(ctx->counters, 0, 14 * sizeof(unsigned long long), ctx->stream));
  return RT_OK;
}

int rt_set_partition(rt_ctx* ctx, uint32_t rank, uint32_t world) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (world == 0 || rank >= world) return fail(RT_ERR_INVALID, "need rank < world");
  ctx->rank = rank;
  ctx->world = world;
  return RT_OK;
}

int rt_sky_tile_stats(rt_ctx* ctx, uint32_t* sky_tiles, uint32_t* tiles) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!sky_tiles || !tiles) return fail(RT_ERR_INVALID, "null out pointer");
  *sky_tiles = *tiles = 0;
  RT_HIP(hipStreamSynchronize(ctx->stream));
  if (!ctx->sky_flags || !ctx->sky_valid || ctx->sky_tiles_cap == 0) return RT_OK;  // no pass has used sky tiles yet
  std::vector<uint32_t> h(ctx->sky_tiles_cap);
  RT_HIP(hipMemcpy(h.data(), ctx->sky_flags, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  uint32_t n = 0;
  for (uint32_t f : h) n += f == 0u;  // mask 0 = no sphere can be hit
  *sky_tiles = n;
  *tiles = ctx->sky_tiles_cap;
  return RT_OK;
}

int rt_accum_device_ptr(rt_ctx* ctx, void** dptr, size_t* bytes) {
  if (int rc = check(ctx)) return rc;
  if (!dptr || !bytes) return fail(RT_ERR_INVALID, "null out pointer");
  RT_FLUSH(ctx);  // whoever asks for the address is about to look at the image: the recorded passes must be on the stream
  *dptr = ctx->accum;
  *bytes = ctx->accum_bytes;
  return RT_OK;
}

int rt_bind_accum(rt_ctx* ctx, void* dptr, size_t bytes) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (dptr && bytes != ctx->accum_bytes) return fail(RT_ERR_INVALID, "rt_bind_accum: size must equal the image size");
  if (dptr && (reinterpret_cast<uintptr_t>(dptr) & 15u)) return fail(RT_ERR_INVALID, "rt_bind_accum: need 16-byte alignment");
  RT_HIP(hipStreamSynchronize(ctx->stream));
  ctx->accum = dptr ? dptr : ctx->own_accum;
  return RT_OK;
}

int rt_set_stream(rt_ctx* ctx, void* hip_stream) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  RT_HIP(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return RT_OK;
}

int rt_clear(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  RT_HIP(hipMemsetAsync(ctx->accum, 0, ctx->accum_bytes, ctx->stream));
  if (ctx->rng) RT_HIP(hipMemsetAsync(ctx->rng, 0, (size_t)ctx->width * ctx->height * 4u, ctx->stream));
  return RT_OK;
}

int rt_enable_timing(rt_ctx* ctx, int enable) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  ctx->timing = enable != 0;
  return RT_OK;
}

int rt_timing_reset(rt_ctx* ctx) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  RT_HIP(hipStreamSynchronize(ctx->stream));
  ctx->events_used = 0;
  return RT_OK;
}

int rt_timing_get(rt_ctx* ctx, double* total_ms, uint32_t* launches) {
  if (int rc = check(ctx)) return rc;
  RT_FLUSH(ctx);
  if (!total_ms || !launches) return fail(RT_ERR_INVALID, "null out pointer");
  RT_HIP(hipStreamSynchronize(ctx->stream));
  double sum = 0.0;
  for (size_t i = 0; i < ctx->events_used; ++i) {
    float ms = 0.f;
    RT_HIP(hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
    sum += ms;
  }
  *total_ms = sum;
  *launches = (uint32_t)ctx->events_used;
  return RT_OK;
}

int rt_device_info(rt_ctx* ctx, char* name, size_t name_bytes, int* compute_units, int* clock_khz) {
  if (int rc = check(ctx)) return rc;
  hipDeviceProp_t prop;
  RT_HIP(hipGetDeviceProperties(&prop, ctx->device));
  if (name && name_bytes) std::snprintf(name, name_bytes, "%s (%s)", prop.name, prop.gcnArchName);
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  return RT_OK;
}

}  // extern "C"
