// rt_kernels.hip — gfx950 path-tracing kernels behind librt_hip.so.
//
// Replaces the WGSL entry point `main` of the reference (src/shaders/raytrace_kernal.wgsl:72-83)
// and the buffer->cache copy of Raytracer::compute (src/raytracing.rs:195-199). Written for CDNA4,
// not translated from the shader:
//
//  * one 64-lane wavefront = one 8x8 pixel tile (one pixel per lane): primary rays of a wave stay
//    coherent on sphere silhouettes, and each 8-pixel row of float4 accumulators is one 128-B line;
//  * workgroup = 1 wave: no barriers, the dispatcher balances tiles of very different cost;
//  * the sample loop, the bounce loop and (in batched launches) the frame loop of a pixel are
//    FLATTENED into one loop of trace() iterations with path regeneration: a lane whose path ends
//    starts its next sample (or next frame) in the same iteration, so a wave runs for
//    max-over-lanes of TOTAL trace calls instead of sum-over-samples of max-over-lanes of path
//    length. Each lane still consumes its own RNG stream in the reference's order.
//  * the sphere index of the hit loop is wave-uniform: test data {cx,cy,cz,r*r} is read with
//    scalar loads (s_load_dwordx4/x8/x16 through the constant address space) and used straight
//    from SGPRs; only the winning sphere's shading data is fetched per lane;
//  * the accumulation image is read once and written once per launch, in place.
//
// Two arithmetic modes (rt_types.h): RT_MATH_STRICT restates the oracle's IEEE contract operation
// by operation (bit-identical images and RNG states); RT_MATH_FAST uses FMA, v_rcp/v_rsq/v_sqrt and
// dot() for length_squared.  This file must be compiled with -ffp-contract=off: every fused
// operation below is written explicitly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f32x4* const_f32x4_ptr;  // scalar-load path

__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

// raytrace_kernal.wgsl:265-275 (state update) — integer, bit-exact in every mode
__device__ __forceinline__ uint32_t rng_step(uint32_t rng) {
  rng = rng * 18197u + 182489729u;
  rng ^= (rng << 15);
  rng ^= (rng >> 22);
  rng ^= (rng << 4);
  return rng;
}
__device__ __forceinline__ float rng_unit(uint32_t rng) { return as_f32((rng & 0x007fffffu) | 0x3f800000u) - 1.0f; }
#define RT_RAND(dst)        \
  do {                      \
    rng = rng_step(rng);    \
    (dst) = rng_unit(rng);  \
    if (COUNT) c_rand++;    \
  } while (0)

// raytrace_kernal.wgsl:255-261
__device__ __forceinline__ uint32_t rng_seed(uint32_t u, uint32_t v, uint32_t time_bits) {
  uint32_t x = (u << 15) ^ (u * 1729u + 8192374u);
  uint32_t y = (v << 21) ^ (v * 271u + 719827371u);
  uint32_t t = time_bits * 87189u + 189273u;
  return x ^ y ^ t;
}

// strict dot: (ax*bx + ay*by) + az*bz, three roundings for the products, two for the sums
__device__ __forceinline__ float dot_strict(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ float dot_fast(float ax, float ay, float az, float bx, float by, float bz) {
  return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

__device__ __forceinline__ uint32_t unorm8(float x) {  // rgba8unorm store: clamp (NaN -> 0), *255, RNE
  float c = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
  return (uint32_t)__builtin_rintf(c * 255.0f);
}

template <bool STRICT, bool COUNT>
__global__ __launch_bounds__(64) void rt_path_kernel(const rt_kargs a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t tile = blockIdx.x * a.world + a.rank;
  if (tile >= a.n_tiles) return;
  const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const uint32_t u = tx * 8u + (lane & 7u);
  const uint32_t v = ty * 8u + (lane >> 3);
  if (u >= a.cov_w || v >= a.cov_h) return;  // only in RT_DISPATCH_FULL with ragged edges

  const bool checker = (a.flags & RT_SHADE_CHECKER) != 0;
  const bool accum_f32 = (a.flags & RT_ACCUM_F32) != 0;
  const bool combine_sum = (a.flags & RT_COMBINE_SUM) != 0;

  // ---- pixel_to_world / get_pixel_delta (wgsl:85-94): once per pixel, frame-invariant ---------
  const float* m = a.p2w;  // column-major
  float pcx, pcy, pcz, pdx, pdy, pdz;
  {
    const float vx = (float)u, vy = (float)(a.height - v);
    const float wx = (float)(u + 1u), wy = (float)(a.height - (v + 1u));
    pcx = ((m[0] * vx + m[4] * vy) + m[8] * 0.0f) + m[12] * 1.0f;
    pcy = ((m[1] * vx + m[5] * vy) + m[9] * 0.0f) + m[13] * 1.0f;
    pcz = ((m[2] * vx + m[6] * vy) + m[10] * 0.0f) + m[14] * 1.0f;
    const float nx = ((m[0] * wx + m[4] * wy) + m[8] * 0.0f) + m[12] * 1.0f;
    const float ny = ((m[1] * wx + m[5] * wy) + m[9] * 0.0f) + m[13] * 1.0f;
    const float nz = ((m[2] * wx + m[6] * wy) + m[10] * 0.0f) + m[14] * 1.0f;
    pdx = nx - pcx; pdy = ny - pcy; pdz = nz - pcz;
  }

  // ---- accumulator: read once ------------------------------------------------------------------
  const size_t pix = (size_t)v * a.width + u;
  float acc0, acc1, acc2, acc3;
  if (accum_f32) {
    const f32x4 c = reinterpret_cast<const f32x4*>(a.accum)[pix];
    acc0 = c.x; acc1 = c.y; acc2 = c.z; acc3 = c.w;
  } else {
    const uint32_t q = reinterpret_cast<const uint32_t*>(a.accum)[pix];
    acc0 = (float)(q & 255u) / 255.0f; acc1 = (float)((q >> 8) & 255u) / 255.0f;
    acc2 = (float)((q >> 16) & 255u) / 255.0f; acc3 = (float)(q >> 24) / 255.0f;
  }

  const const_f32x4_ptr sph_test = (const_f32x4_ptr)(uintptr_t)a.sph_test;
  const uint32_t n_spheres = a.n_spheres;
  const float spp_f = (float)a.spp;

  uint32_t c_T = 0, c_D = 0, c_H = 0, c_R = 0, c_M = 0, c_S = 0, c_Sm = 0, c_J = 0, c_Sd = 0, c_N = 0, c_P = 0, c_rand = 0;

  // ---- flattened frame x sample x bounce loop ------------------------------------------------
  uint32_t k = 0;  // frame within this launch
  uint32_t rng = rng_seed(u, v, a.time_table ? a.time_table[0] : a.time_bits0);
  rng = rng_step(rng);  // initialize_rng's discarded rand() (wgsl:262)
  if (COUNT) c_rand++;
  uint32_t sample = 0;
  float sumx = 0.0f, sumy = 0.0f, sumz = 0.0f;  // send_rays' `color`
  float ox, oy, oz, dx, dy, dz;                // ray
  float colx, coly, colz, refl;                // ray_color's `color`, `reflectance`
  uint32_t depth;
  {  // get_random_ray (wgsl:174-180) for sample 0
    float r0, r1;
    RT_RAND(r0); RT_RAND(r1);
    const float px = -0.5f + r0, py = -0.5f + r1;
    dx = pcx + px * pdx; dy = pcy + py * pdy; dz = pcz + 1.0f * pdz;
    ox = a.pos[0]; oy = a.pos[1]; oz = a.pos[2];
    colx = coly = colz = 0.0f; refl = 1.0f; depth = 0;
    if (COUNT) c_N++;
  }

  bool alive = true;
  while (alive) {
    // ================= trace (wgsl:182-231) =================
    float closest = RT_RAY_TMAX;
    int hit = -1;
    float aa, inv_a;
    if (STRICT) {
      const float len = __builtin_sqrtf(dot_strict(dx, dy, dz, dx, dy, dz));
      aa = len * len;  // length_squared (wgsl:306-309)
      inv_a = 0.0f;
    } else {
      aa = dot_fast(dx, dy, dz, dx, dy, dz);
      inv_a = __builtin_amdgcn_rcpf(aa);
    }
    if (COUNT) { c_R++; c_T += n_spheres; }
#pragma unroll 4
    for (uint32_t i = 0; i < n_spheres; ++i) {
      const f32x4 s = sph_test[i];  // {cx, cy, cz, r*r}: wave-uniform -> SGPRs
      const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
      if (STRICT) {
        const float hb = dot_strict(ocx, ocy, ocz, dx, dy, dz);
        const float ss = dot_strict(ocx, ocy, ocz, ocx, ocy, ocz);
        // Exact-arithmetic filter. The oracle's c = fl(fl(sqrt(ss))^2) - rr differs from ss - rr by
        // less than (ss+rr)*2^-21, so fl(a*c) differs from fl(a*(ss-rr)) by less than
        // a*(ss+rr)*2^-20.5. If the cheap discriminant is below -a*(ss+rr)*2^-19 the exact one is
        // negative too (a certain miss) and the two IEEE square roots are skipped. Anything else —
        // including NaN/inf, for which the comparison is false — takes the exact path.
        const float hb2 = hb * hb;
        const float dt = hb2 - aa * (ss - s.w);
        const float margin = __builtin_fmaf((ss + s.w) * aa, 0x1p-19f, 1e-30f);
        if (!(dt < -margin)) {
          const float len = __builtin_sqrtf(ss);
          const float c = len * len - s.w;
          const float disc = hb2 - aa * c;
          if (!(disc < 0.0f)) {
            if (COUNT) c_D++;
            const float sq = __builtin_sqrtf(disc);
            float root = (-hb - sq) / aa;
            bool ok = true;
            if (root <= RT_RAY_TMIN || closest <= root) {
              root = (-hb + sq) / aa;
              if (root <= RT_RAY_TMIN || closest <= root) ok = false;
            }
            if (ok) { closest = root; hit = (int)i; }
          }
        }
      } else {
        const float hb = dot_fast(ocx, ocy, ocz, dx, dy, dz);
        const float cc = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -s.w)));
        const float disc = __builtin_fmaf(-aa, cc, hb * hb);
        if (!(disc < 0.0f)) {
          if (COUNT) c_D++;
          const float sq = __builtin_amdgcn_sqrtf(disc);
          float root = (-hb - sq) * inv_a;
          bool ok = true;
          if (root <= RT_RAY_TMIN || closest <= root) {
            root = (-hb + sq) * inv_a;
            if (root <= RT_RAY_TMIN || closest <= root) ok = false;
          }
          if (ok) { closest = root; hit = (int)i; }
        }
      }
    }

    // ================= shade =================
    bool end_path;
    if (hit < 0) {
      // miss (wgsl:144-148); ray_color: color += miss * reflectance; break
      if (COUNT) c_M++;
      if (!checker) {
        const float t = 0.5f * (1.0f - dy);
        const float it = 1.0f - t;
        colx = colx + (it + t * 0.5f) * refl;
        coly = coly + (it + t * 0.7f) * refl;
        colz = colz + (it + t * 1.0f) * refl;
      } else {
        colx = colx + 0.0f * refl; coly = coly + 0.0f * refl; colz = colz + 0.0f * refl;
      }
      end_path = true;
    } else {
      if (COUNT) { c_H++; c_S++; }
      const f32x4 sg = a.sph_shade[hit];  // {cx, cy, cz, r}
      const f32x4 sc = a.sph_col[hit];    // {material r, g, b, 1/r}
      // hit record of the closest sphere (wgsl:225-227): point = ray_at(t), normal = (point - c) / r
      float px, py, pz, nx, ny, nz;
      if (STRICT) {
        px = ox + closest * dx; py = oy + closest * dy; pz = oz + closest * dz;
        nx = (px - sg.x) / sg.w; ny = (py - sg.y) / sg.w; nz = (pz - sg.z) / sg.w;
      } else {
        px = __builtin_fmaf(closest, dx, ox); py = __builtin_fmaf(closest, dy, oy); pz = __builtin_fmaf(closest, dz, oz);
        nx = (px - sg.x) * sc.w; ny = (py - sg.y) * sc.w; nz = (pz - sg.z) * sc.w;
      }
      // scatter (wgsl:150-168)
      float kk = (1.0f - RT_LOCAL_REFLECTANCE) * refl;
      if (checker) {
        const float fx = __builtin_floorf(4.0f * nx);
        const float fy = __builtin_floorf(4.0f * ny + 0.5f);
        const float h = (fx + fy) * 0.5f;
        kk = (h == __builtin_floorf(h)) ? refl : 0.0f;
      }
      colx = colx + sc.x * kk; coly = coly + sc.y * kk; colz = colz + sc.z * kk;
      refl = refl * RT_LOCAL_REFLECTANCE;
      ox = px; oy = py; oz = pz;
      if ((hit & 1) == 0) {
        // reflection (wgsl:170-172): dir - 2*dot(dir, n)*n
        if (COUNT) c_Sm++;
        if (STRICT) {
          const float k2 = 2.0f * dot_strict(dx, dy, dz, nx, ny, nz);
          dx = dx - k2 * nx; dy = dy - k2 * ny; dz = dz - k2 * nz;
        } else {
          const float k2 = -2.0f * dot_fast(dx, dy, dz, nx, ny, nz);
          dx = __builtin_fmaf(k2, nx, dx); dy = __builtin_fmaf(k2, ny, dy); dz = __builtin_fmaf(k2, nz, dz);
        }
      } else {
        // normal + random_unit_vector() (wgsl:281-294): rejection sampling, 3 rand per try
        if (COUNT) c_Sd++;
        float qx, qy, qz, l2;
        for (;;) {
          float r0, r1, r2;
          RT_RAND(r0); RT_RAND(r1); RT_RAND(r2);
          if (COUNT) c_J++;
          if (STRICT) {
            qx = r0 * 2.0f - 1.0f; qy = r1 * 2.0f - 1.0f; qz = r2 * 2.0f - 1.0f;
            l2 = dot_strict(qx, qy, qz, qx, qy, qz);
            // accept iff fl(fl(sqrt(l2))^2) < 1. fl(sqrt(.)) is monotone and fl(sqrt(1)) = 1, and
            // for l2 < 1 the rounded root is <= 1-2^-24 whose square rounds below 1; so the test
            // equals l2 < 1 exactly (no square root needed).
            if (l2 < 1.0f) break;
          } else {
            qx = __builtin_fmaf(r0, 2.0f, -1.0f); qy = __builtin_fmaf(r1, 2.0f, -1.0f); qz = __builtin_fmaf(r2, 2.0f, -1.0f);
            l2 = dot_fast(qx, qy, qz, qx, qy, qz);
            if (l2 < 1.0f) break;
          }
        }
        if (STRICT) {
          const float len = __builtin_sqrtf(l2);
          dx = nx + qx / len; dy = ny + qy / len; dz = nz + qz / len;
        } else {
          const float il = __builtin_amdgcn_rsqf(l2);
          dx = __builtin_fmaf(qx, il, nx); dy = __builtin_fmaf(qy, il, ny); dz = __builtin_fmaf(qz, il, nz);
        }
      }
      depth++;
      end_path = depth > (uint32_t)RT_MAX_RAY_DEPTH;  // for (depth = 0; depth <= MAX_RAY_DEPTH; depth++) (wgsl:131)
    }

    if (end_path) {
      // send_rays: color += ray_color(ray) (wgsl:120)
      sumx = sumx + colx; sumy = sumy + coly; sumz = sumz + colz;
      sample++;
      if (sample == a.spp) {
        // ---- end of this pixel's frame: color / spp (wgsl:122), store_color (wgsl:96-112) ----
        const float p0 = sumx / spp_f, p1 = sumy / spp_f, p2 = sumz / spp_f, p3 = 1.0f;
        if (COUNT) c_P++;
        if (combine_sum) {
          acc0 = acc0 + p0; acc1 = acc1 + p1; acc2 = acc2 + p2; acc3 = acc3 + p3;
        } else {
          const uint32_t frame = a.first_frame + k;
          const float cw = (float)frame / (float)(frame + 1u);
          const float rw = 1.0f - cw;
          acc0 = p0 * rw + acc0 * cw; acc1 = p1 * rw + acc1 * cw; acc2 = p2 * rw + acc2 * cw; acc3 = p3 * rw + acc3 * cw;
        }
        if (!accum_f32) {  // the reference re-quantises to rgba8unorm every frame (raytracing.rs:30,47)
          acc0 = (float)unorm8(acc0) / 255.0f; acc1 = (float)unorm8(acc1) / 255.0f;
          acc2 = (float)unorm8(acc2) / 255.0f; acc3 = (float)unorm8(acc3) / 255.0f;
        }
        k++;
        if (k == a.n_frames) {
          alive = false;
        } else {
          rng = rng_seed(u, v, a.time_table[k]);
          rng = rng_step(rng);
          if (COUNT) c_rand++;
          sample = 0; sumx = sumy = sumz = 0.0f;
        }
      }
      if (alive) {  // path regeneration: get_random_ray for the next sample
        float r0, r1;
        RT_RAND(r0); RT_RAND(r1);
        const float px = -0.5f + r0, py = -0.5f + r1;
        dx = pcx + px * pdx; dy = pcy + py * pdy; dz = pcz + 1.0f * pdz;
        ox = a.pos[0]; oy = a.pos[1]; oz = a.pos[2];
        colx = coly = colz = 0.0f; refl = 1.0f; depth = 0;
        if (COUNT) c_N++;
      }
    }
  }

  // ---- write once ------------------------------------------------------------------------------
  if (accum_f32) {
    f32x4 o; o.x = acc0; o.y = acc1; o.z = acc2; o.w = acc3;
    reinterpret_cast<f32x4*>(a.accum)[pix] = o;
  } else {
    // acc* hold q/255 exactly; recover q
    const uint32_t q = unorm8(acc0) | (unorm8(acc1) << 8) | (unorm8(acc2) << 16) | (unorm8(acc3) << 24);
    reinterpret_cast<uint32_t*>(a.accum)[pix] = q;
  }
  if (a.rng_out) a.rng_out[pix] = rng;

  if (COUNT) {
    const uint32_t vals[12] = {c_T, c_D, c_H, c_R, c_M, c_S, c_Sm, c_J, c_Sd, c_N, c_P, c_rand};
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(&a.counters[i], (unsigned long long)vals[i]);
  }
}

// RGBA8 view of a float4 accumulation image (rt_read_rgba8 on RT_ACCUM_F32 contexts)
__global__ void rt_quantize_kernel(const f32x4* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 c = src[i];
  dst[i] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

}  // namespace

hipError_t rt_launch_path_kernel(const rt_kargs& a, uint32_t my_tiles, bool strict, bool count, hipStream_t stream) {
  if (my_tiles == 0) return hipSuccess;
  const dim3 grid(my_tiles), block(64);
  if (strict) {
    if (count) hipLaunchKernelGGL((rt_path_kernel<true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rt_path_kernel<true, false>), grid, block, 0, stream, a);
  } else {
    if (count) hipLaunchKernelGGL((rt_path_kernel<false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rt_path_kernel<false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t rt_launch_quantize(const void* src_f32x4, void* dst_rgba8, size_t n_pixels, hipStream_t stream) {
  if (n_pixels == 0) return hipSuccess;
  const uint32_t block = 256;
  const uint32_t grid = (uint32_t)((n_pixels + block - 1) / block);
  hipLaunchKernelGGL(rt_quantize_kernel, dim3(grid), dim3(block), 0, stream,
                     reinterpret_cast<const f32x4*>(src_f32x4), reinterpret_cast<uint32_t*>(dst_rgba8), n_pixels);
  return hipGetLastError();
}
