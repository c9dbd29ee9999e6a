/* rt_oracle.c — CPU restatement of the reference's per-pixel path-tracing compute pass.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under raytracer_amd/ may include, link, import or execute this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED BY THE REFERENCE: atbarry/raytracer ships no tests, golden vectors or
 * known-answer fixtures for this path, and it cannot be built or run in this
 * image (Rust + wgpu 0.18 + a WGSL compiler + a Vulkan ICD are all absent), so there is no
 * oracle/_ref. This file follows the WGSL text line by line; its integer RNG is additionally
 * pinned by tests/test_oracle_kat.py against an independent numpy derivation of the formulas at
 * raytrace_kernal.wgsl:255-275.
 *
 * What is restated (reference file:line given at each function):
 *   src/shaders/raytrace_kernal.wgsl   (every function reachable from `main`)
 *   src/raytracing.rs:20-52,178-200    (RGBA8 textures, floor(w/16) x floor(h/16) dispatch, buffer->cache copy)
 *   src/common/time.rs:72-77, src/app.rs:93-116 (time += 0.01f before each frame; frame counter after)
 *
 * Arithmetic contract ("strict"): WGSL leaves contraction, division/sqrt precision and the
 * expansion of dot/length/normalize to the driver. This oracle fixes ONE choice, the plainest:
 *   - every +,-,*,/ and sqrt is a single correctly-rounded IEEE-754 binary32 operation,
 *   - no FMA contraction, no re-association (build with -ffp-contract=off, no -ffast-math),
 *   - dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - length(x)     = sqrt(dot(x,x));  normalize(x) = x / length(x) (three divisions)
 *   - mat4*vec4     = ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w
 *   - vec op scalar = componentwise
 *   - rgba8unorm store: clamp to [0,1] (NaN -> 0), *255, round-to-nearest-even; load: q/255
 * The HIP kernels in RT_MATH_STRICT mode make the same choices and are compared bit for bit.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- byte layouts (own copies; tests assert they match include/rt_types.h) ---------------- */
typedef struct { float center[3]; uint32_t padding; float radius; uint32_t material_index; uint32_t pad1[2]; } o_sphere;   /* wgsl:1-6   */
typedef struct { float color[4]; float metallic, roughness, specular, padding; } o_material;                              /* wgsl:8-13  */
typedef struct {                                                                                                          /* wgsl:30-39 */
  float world_to_pixel[16]; float pixel_to_world[16]; float pos[4];
  float focal_length; uint32_t samples_per_pixel, frames_to_render, current_frame;
} o_camera;

/* flags: same numeric values as include/rt_types.h (asserted in tests) */
#define O_ACCUM_F32 0x1u
#define O_DISPATCH_FULL 0x2u
#define O_SHADE_CHECKER 0x4u
#define O_COMBINE_SUM 0x10u

typedef struct { uint64_t T, D, H, R, M, S, Sm, J, Sd, N, P, rand_calls; } o_counters;

typedef struct { float x, y, z; } v3;

/* wgsl:53-56 */
static const float RAY_TMAX = 10000000.0f;
static const float RAY_TMIN = 0.001f;
#define MAX_RAY_DEPTH 10

typedef struct {
  /* per-invocation private state (wgsl:62-63,254) */
  uint32_t uvx, uvy, sizex, sizey;
  uint32_t rng;
  /* bindings */
  const o_camera* camera;
  const o_sphere* spheres; uint32_t n_spheres;
  const o_material* materials;
  uint32_t time_bits;
  uint32_t flags;
  o_counters c;
} inv_t;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }

/* wgsl:306-309  length_squared: sqrt, then square */
static inline float length_squared(v3 x) { float len = vlength(x); return len * len; }

/* wgsl:265-275 */
static inline float rng_rand(inv_t* s) {
  uint32_t rng = s->rng;
  rng = rng * 18197u + 182489729u;
  rng ^= (rng << 15);
  rng ^= (rng >> 22);
  rng ^= (rng << 4);
  s->rng = rng;
  s->c.rand_calls++;
  uint32_t bits = (rng & 0x007fffffu) | 0x3f800000u;
  float f; memcpy(&f, &bits, 4);
  return f - 1.0f;
}

/* wgsl:255-263 */
static inline uint32_t rng_seed(uint32_t u, uint32_t v, uint32_t time_bits) {
  uint32_t x = (u << 15) ^ (u * 1729u + 8192374u);
  uint32_t y = (v << 21) ^ (v * 271u + 719827371u);
  uint32_t t = time_bits * 87189u + 189273u;
  return x ^ y ^ t;
}
static inline void initialize_rng(inv_t* s) {
  s->rng = rng_seed(s->uvx, s->uvy, s->time_bits);
  (void)rng_rand(s);
}

/* wgsl:85-88  (pixel_to_world * vec4(f32(x), f32(size.y - y), 0, 1)).xyz ; u32 subtraction wraps */
static inline v3 pixel_to_world(const inv_t* s, uint32_t px, uint32_t py) {
  const float* m = s->camera->pixel_to_world; /* column-major: column c at m[4c..4c+3] */
  float vx = (float)px, vy = (float)(uint32_t)(s->sizey - py), vz = 0.0f, vw = 1.0f;
  v3 r;
  r.x = ((m[0] * vx + m[4] * vy) + m[8] * vz) + m[12] * vw;
  r.y = ((m[1] * vx + m[5] * vy) + m[9] * vz) + m[13] * vw;
  r.z = ((m[2] * vx + m[6] * vy) + m[10] * vz) + m[14] * vw;
  return r;
}

/* wgsl:90-94 */
static inline v3 get_pixel_delta(const inv_t* s) {
  v3 center = pixel_to_world(s, s->uvx, s->uvy);
  v3 uvnext = pixel_to_world(s, 1u + s->uvx, 1u + s->uvy);
  return vsub(uvnext, center);
}

typedef struct { v3 origin, direction; } ray_t;
typedef struct { v3 point, normal; float t; uint32_t index; int hit; uint32_t material_index; } hit_t;

/* wgsl:233-235 */
static inline v3 ray_at(ray_t r, float t) { return vadd(r.origin, vscale(r.direction, t)); }

/* wgsl:200-231 */
static inline int hit_sphere(inv_t* s, ray_t ray, const o_sphere* sp, float tmin, float tmax, hit_t* rec) {
  v3 center = V(sp->center[0], sp->center[1], sp->center[2]);
  v3 oc = vsub(ray.origin, center);
  float a = length_squared(ray.direction);
  float half_b = vdot(oc, ray.direction);
  float c = length_squared(oc) - sp->radius * sp->radius;
  s->c.T++;
  float discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0f) return 0;
  s->c.D++;
  float sqrtd = sqrtf(discriminant);
  float root = (-half_b - sqrtd) / a;
  if (root <= tmin || tmax <= root) {
    root = (-half_b + sqrtd) / a;
    if (root <= tmin || tmax <= root) return 0;
  }
  rec->t = root;
  rec->point = ray_at(ray, root);
  rec->normal = vdivs(vsub(rec->point, center), sp->radius);
  rec->material_index = sp->material_index;
  return 1;
}

/* wgsl:182-198 */
static inline hit_t trace(inv_t* s, ray_t ray) {
  hit_t rec; memset(&rec, 0, sizeof rec);
  hit_t temp; memset(&temp, 0, sizeof temp);
  float closest_so_far = RAY_TMAX;
  s->c.R++;
  for (uint32_t i = 0; i < s->n_spheres; i++) {
    if (hit_sphere(s, ray, &s->spheres[i], RAY_TMIN, closest_so_far, &temp)) {
      closest_so_far = temp.t;
      rec = temp;
      rec.index = i;
      rec.hit = 1;
    }
  }
  if (rec.hit) s->c.H++;
  return rec;
}

/* wgsl:144-148 ; RT_SHADE_CHECKER: sky off */
static inline v3 miss(inv_t* s, ray_t ray) {
  s->c.M++;
  if (s->flags & O_SHADE_CHECKER) return V(0.0f, 0.0f, 0.0f);
  float a = 0.5f * (1.0f - ray.direction.y);
  float ia = 1.0f - a;
  return V(ia * 1.0f + a * 0.5f, ia * 1.0f + a * 0.7f, ia * 1.0f + a * 1.0f);
}

/* wgsl:281-290 */
static inline v3 random_in_unit_sphere(inv_t* s) {
  for (;;) {
    float rx = rng_rand(s), ry = rng_rand(s), rz = rng_rand(s);
    v3 p = V(rx * 2.0f - 1.0f, ry * 2.0f - 1.0f, rz * 2.0f - 1.0f);
    s->c.J++;
    if (length_squared(p) < 1.0f) return p;
  }
}
/* wgsl:292-294 */
static inline v3 random_unit_vector(inv_t* s) {
  v3 p = random_in_unit_sphere(s);
  return vdivs(p, vlength(p));
}
/* wgsl:170-172  dir - 2.0 * dot(dir, normal) * normal */
static inline v3 reflection(v3 dir, v3 n) { return vsub(dir, vscale(n, 2.0f * vdot(dir, n))); }

/* RT_SHADE_CHECKER gate. The reference's dead terms b = sin(4*pi*n.x), c = cos(4*pi*n.y)
 * (wgsl:157-158) have b*c > 0 exactly when floor(4*n.x) + floor(4*n.y + 0.5) is even; the gate is
 * stated in that floor form so CPU and GPU agree bit for bit without a shared sin/cos. */
static inline int checker_glow(v3 n) {
  float fx = floorf(4.0f * n.x);
  float fy = floorf(4.0f * n.y + 0.5f);
  float h = (fx + fy) * 0.5f;
  return h == floorf(h);
}

/* wgsl:150-168 */
static inline v3 scatter(inv_t* s, ray_t* ray, hit_t rec, float* reflectance) {
  const o_material* m = &s->materials[rec.material_index];
  v3 color = V(m->color[0], m->color[1], m->color[2]);
  const float local_reflectance = 0.90f;
  float k = (1.0f - local_reflectance) * (*reflectance);
  if (s->flags & O_SHADE_CHECKER) k = checker_glow(rec.normal) ? (*reflectance) : 0.0f;
  color = vscale(color, k);
  *reflectance = *reflectance * local_reflectance;
  s->c.S++;
  /* wgsl:157-159: b, c, d are dead code at HEAD */
  ray->origin = rec.point;
  if (rec.index % 2u == 0u) {
    s->c.Sm++;
    ray->direction = reflection(ray->direction, rec.normal);
  } else {
    s->c.Sd++;
    ray->direction = vadd(rec.normal, random_unit_vector(s));
  }
  return color;
}

/* wgsl:125-142 */
static inline v3 ray_color(inv_t* s, ray_t start) {
  ray_t ray = start;
  v3 color = V(0.0f, 0.0f, 0.0f);
  float reflectance = 1.0f;
  for (int depth = 0; depth <= MAX_RAY_DEPTH; depth++) {
    hit_t rec = trace(s, ray);
    if (!rec.hit) {
      color = vadd(color, vscale(miss(s, ray), reflectance));
      break;
    } else {
      color = vadd(color, scatter(s, &ray, rec, &reflectance));
    }
  }
  return color;
}

/* wgsl:174-180 */
static inline ray_t get_random_ray(inv_t* s, v3 pixel_center, v3 pixel_delta) {
  float px = -0.5f + rng_rand(s);
  float py = -0.5f + rng_rand(s);
  v3 pixel_sample = vadd(pixel_center, vmul(V(px, py, 1.0f), pixel_delta));
  ray_t r; r.origin = V(s->camera->pos[0], s->camera->pos[1], s->camera->pos[2]); r.direction = pixel_sample;
  return r;
}

/* wgsl:114-123 */
static inline v3 send_rays(inv_t* s) {
  v3 pixel_center = pixel_to_world(s, s->uvx, s->uvy);
  v3 pixel_delta = get_pixel_delta(s);
  v3 color = V(0.0f, 0.0f, 0.0f);
  for (uint32_t sample = 0; sample < s->camera->samples_per_pixel; sample++) {
    ray_t ray = get_random_ray(s, pixel_center, pixel_delta);
    s->c.N++;
    color = vadd(color, ray_color(s, ray));
  }
  return vdivs(color, (float)s->camera->samples_per_pixel);
}

static inline uint8_t unorm8_store(float x) { /* rgba8unorm textureStore */
  float c = fminf(fmaxf(x, 0.0f), 1.0f);
  return (uint8_t)rintf(c * 255.0f);
}
static inline float unorm8_load(uint8_t q) { return (float)q / 255.0f; }

static inline void ctr_add(o_counters* d, const o_counters* s) {
  d->T += s->T; d->D += s->D; d->H += s->H; d->R += s->R; d->M += s->M; d->S += s->S; d->Sm += s->Sm;
  d->J += s->J; d->Sd += s->Sd; d->N += s->N; d->P += s->P; d->rand_calls += s->rand_calls;
}

/* One compute pass = Raytracer::compute (raytracing.rs:178-200): the dispatch plus the
 * color_buffer -> color_cache copy, which makes the two textures one in-place buffer here.
 *
 *  rgba8      W*H*4 bytes, in/out   (RT_ACCUM_RGBA8; may be NULL in F32 mode)
 *  accum      W*H*4 floats, in/out  (RT_ACCUM_F32;   may be NULL in RGBA8 mode)
 *  rng_out    W*H u32, out: final RNG state of every covered pixel (NULL to skip)
 *  counters   out, accumulated (NULL to skip)
 *  tile_rank/tile_world: render only the 8x8 tiles t (row-major over the covered area) with
 *             t % tile_world == tile_rank  (1-process oracle of the multi-GPU tile split)
 * Returns 0, or -1 on bad arguments.
 */
int rto_compute(uint32_t width, uint32_t height, uint32_t flags,
                const o_sphere* spheres, uint32_t n_spheres, const o_material* materials, uint32_t n_materials,
                const o_camera* camera, uint32_t time_bits,
                uint8_t* rgba8, float* accum, uint32_t* rng_out, o_counters* counters,
                uint32_t tile_rank, uint32_t tile_world, int threads) {
  if (!spheres || !materials || !camera || tile_world == 0 || tile_rank >= tile_world) return -1;
  if ((flags & O_ACCUM_F32) ? !accum : !rgba8) return -1;
  for (uint32_t i = 0; i < n_spheres; i++) if (spheres[i].material_index >= n_materials) return -1;
  /* wgsl:76  early return when the render is finished */
  if (camera->current_frame >= camera->frames_to_render) return 0;
  /* raytracing.rs:192  dispatch_workgroups(width/16, height/16, 1) */
  uint32_t cov_w = (flags & O_DISPATCH_FULL) ? width : (width / 16u) * 16u;
  uint32_t cov_h = (flags & O_DISPATCH_FULL) ? height : (height / 16u) * 16u;
  uint32_t tiles_x = (cov_w + 7u) / 8u;
  o_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    o_counters local; memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1) nowait
    for (int64_t y = 0; y < (int64_t)cov_h; y++) {
      for (uint32_t x = 0; x < cov_w; x++) {
        uint32_t tile = ((uint32_t)y / 8u) * tiles_x + x / 8u;
        if (tile % tile_world != tile_rank) continue;
        inv_t s; memset(&s, 0, sizeof s);
        s.uvx = x; s.uvy = (uint32_t)y; s.sizex = width; s.sizey = height;
        s.camera = camera; s.spheres = spheres; s.n_spheres = n_spheres; s.materials = materials;
        s.time_bits = time_bits; s.flags = flags;
        /* wgsl:80-82 */
        initialize_rng(&s);
        v3 pc = send_rays(&s);
        float pixel[4] = {pc.x, pc.y, pc.z, 1.0f};
        s.c.P++;
        /* wgsl:96-112 store_color / combine_pixel_cache_color */
        size_t idx = ((size_t)y * width + x) * 4u;
        uint32_t frame = camera->current_frame;
        float next_frame = (float)(frame + 1u);
        float cw = (float)frame / next_frame;
        float rw = 1.0f - cw;
        for (int ch = 0; ch < 4; ch++) {
          if (flags & O_ACCUM_F32) {
            if (flags & O_COMBINE_SUM) accum[idx + ch] = accum[idx + ch] + pixel[ch];
            else accum[idx + ch] = pixel[ch] * rw + accum[idx + ch] * cw;
          } else {
            float cached = unorm8_load(rgba8[idx + ch]);
            rgba8[idx + ch] = unorm8_store(pixel[ch] * rw + cached * cw);
          }
        }
        if (rng_out) rng_out[(size_t)y * width + x] = s.rng;
        ctr_add(&local, &s.c);
      }
    }
#pragma omp critical
    ctr_add(&total, &local);
  }
  if (counters) ctr_add(counters, &total);
  return 0;
}

/* Time::add_delta (time.rs:72-77): time = time +f32 delta. Returns the bits after `n` additions
 * of 0.01f starting from `start_bits`. */
uint32_t rto_time_advance(uint32_t start_bits, uint32_t n) {
  float t; memcpy(&t, &start_bits, 4);
  const float delta = 0.01f;
  for (uint32_t i = 0; i < n; i++) t += delta;
  uint32_t b; memcpy(&b, &t, 4);
  return b;
}

/* The frame loop of App::update / App::render (app.rs:93-116) for `n_frames` redraws:
 *   update: time += 0.01f           (app.rs:94)
 *   render: if current_frame < frames_to_render: compute   (app.rs:106-108)
 *           current_frame += 1      (camera.rs:253, after the submit)
 * `camera` and `time_bits` are updated in place, as the host state would be. */
int rto_render(uint32_t width, uint32_t height, uint32_t flags,
               const o_sphere* spheres, uint32_t n_spheres, const o_material* materials, uint32_t n_materials,
               o_camera* camera, uint32_t* time_bits, uint32_t n_frames,
               uint8_t* rgba8, float* accum, uint32_t* rng_out, o_counters* counters,
               uint32_t tile_rank, uint32_t tile_world, int threads) {
  for (uint32_t f = 0; f < n_frames; f++) {
    *time_bits = rto_time_advance(*time_bits, 1);
    int rc = rto_compute(width, height, flags, spheres, n_spheres, materials, n_materials, camera, *time_bits,
                         rgba8, accum, rng_out, counters, tile_rank, tile_world, threads);
    if (rc) return rc;
    camera->current_frame += 1u;
  }
  return 0;
}

/* test hooks for the integer known-answer tests */
uint32_t rto_rng_seed(uint32_t u, uint32_t v, uint32_t time_bits) { return rng_seed(u, v, time_bits); }
uint32_t rto_rng_step(uint32_t state, float* out) {
  inv_t s; memset(&s, 0, sizeof s); s.rng = state;
  float f = rng_rand(&s);
  if (out) *out = f;
  return s.rng;
}
uint32_t rto_sizeof(int what) {
  switch (what) { case 0: return sizeof(o_sphere); case 1: return sizeof(o_material); case 2: return sizeof(o_camera); case 3: return sizeof(o_counters); }
  return 0;
}
int rto_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
