/* rt_types.h — plain-old-data layouts and constants shared by the C ABI (rt_hip.h),
 * the C++ host mirror and any FFI caller (Rust #[repr(C)], ctypes ...).
 *
 * Every struct here is the byte image of a buffer the reference uploads to the GPU:
 *   rt_sphere      <- reference src/resources/objects.rs:17-24,42-49  (WGSL raytrace_kernal.wgsl:1-6)
 *   rt_material    <- reference src/resources/material.rs:5-13        (WGSL raytrace_kernal.wgsl:8-13)
 *   rt_camera_raw  <- reference src/resources/camera.rs:38-48,77-87   (WGSL raytrace_kernal.wgsl:30-39)
 * Pure C so it can be consumed from C, C++, HIP and bindgen.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 32-byte stride. The host writes center.xyzz (objects.rs:44), so `padding` carries the bits of
 * center.z; the shader never reads it. */
typedef struct rt_sphere {
  float center[3];         /* @0  */
  uint32_t padding;        /* @12 bits of center[2] in the reference's upload; ignored */
  float radius;            /* @16 */
  uint32_t material_index; /* @20 */
  uint32_t pad1[2];        /* @24 (u64 padding in objects.rs:23) */
} rt_sphere;

/* 32-byte stride. Only color.xyz is read by the kernel (raytrace_kernal.wgsl:151-152). */
typedef struct rt_material {
  float color[4]; /* @0  */
  float metallic; /* @16 */
  float roughness;/* @20 host order (material.rs:9-11); the shader names them metallic/specular/roughness */
  float specular; /* @24 */
  float padding;  /* @28 */
} rt_material;

/* 160-byte uniform. Matrices are column-major (glam Mat4 / WGSL mat4x4). */
typedef struct rt_camera_raw {
  float world_to_pixel[16];   /* @0   unused by the kernel */
  float pixel_to_world[16];   /* @64  */
  float pos[4];               /* @128 pos.xyzz (camera.rs:81) */
  float focal_length;         /* @144 unused by the kernel */
  uint32_t samples_per_pixel; /* @148 */
  uint32_t frames_to_render;  /* @152 */
  uint32_t current_frame;     /* @156 */
} rt_camera_raw;

#ifdef __cplusplus
static_assert(sizeof(rt_sphere) == 32, "Sphere stride");
static_assert(offsetof(rt_sphere, radius) == 16 && offsetof(rt_sphere, material_index) == 20, "Sphere layout");
static_assert(sizeof(rt_material) == 32, "Material stride");
static_assert(sizeof(rt_camera_raw) == 160, "CameraRaw size");
static_assert(offsetof(rt_camera_raw, pixel_to_world) == 64, "CameraRaw.pixel_to_world");
static_assert(offsetof(rt_camera_raw, pos) == 128, "CameraRaw.pos");
static_assert(offsetof(rt_camera_raw, samples_per_pixel) == 148, "CameraRaw.samples_per_pixel");
static_assert(offsetof(rt_camera_raw, current_frame) == 156, "CameraRaw.current_frame");
#else
_Static_assert(sizeof(rt_sphere) == 32, "Sphere stride");
_Static_assert(sizeof(rt_material) == 32, "Material stride");
_Static_assert(sizeof(rt_camera_raw) == 160, "CameraRaw size");
#endif

/* Shader constants (raytrace_kernal.wgsl:53-56,153). */
#define RT_RAY_TMAX 10000000.0f
#define RT_RAY_TMIN 0.001f
#define RT_MAX_RAY_DEPTH 10
#define RT_LOCAL_REFLECTANCE 0.90f
#define RT_WORKGROUP 16u /* @workgroup_size(16,16,1), raytrace_kernal.wgsl:72 */

/* ---- rt_create flags ---------------------------------------------------------------------- */
/* accumulation buffer format */
#define RT_ACCUM_RGBA8 0x0u /* reference: Rgba8Unorm color_buffer/color_cache, re-quantised every frame (raytracing.rs:30,47) */
#define RT_ACCUM_F32 0x1u   /* float4 running average, un-quantised */
/* which pixels a compute pass covers */
#define RT_DISPATCH_FLOOR16 0x0u /* reference: dispatch_workgroups(w/16, h/16, 1) (raytracing.rs:192) */
#define RT_DISPATCH_FULL 0x2u    /* every pixel */
/* shading variant */
#define RT_SHADE_HEAD 0x0u    /* raytrace_kernal.wgsl at HEAD: sky + 0.1*color emission */
#define RT_SHADE_CHECKER 0x4u /* "glowing checkers" reconstruction (sky off, checker-gated emission); see DESIGN.md */
/* arithmetic */
#define RT_MATH_STRICT 0x0u /* literal WGSL expression order, IEEE-exact, bit-identical to oracle/ */
#define RT_MATH_FAST 0x8u   /* FMA contraction, rcp/rsq, dot for length_squared; within stated image tolerance */
/* combine rule used across frames by rt_compute */
#define RT_COMBINE_AVERAGE 0x0u /* reference running average (raytrace_kernal.wgsl:106-112) */
#define RT_COMBINE_SUM 0x10u    /* F32 only: plain sum of per-frame colours; used by the frame-set split, scaled after the reduce */

typedef struct rt_counters {
  uint64_t T;  /* sphere tests                       (hit_sphere calls)            */
  uint64_t D;  /* tests with discriminant >= 0                                      */
  uint64_t H;  /* trace() calls that hit                                            */
  uint64_t R;  /* trace() calls = rays                                              */
  uint64_t M;  /* miss shadings                                                     */
  uint64_t S;  /* scatter calls                                                     */
  uint64_t Sm; /* mirror bounces                                                    */
  uint64_t J;  /* rejection-sampling tries                                          */
  uint64_t Sd; /* diffuse bounces                                                   */
  uint64_t N;  /* samples (paths)                                                   */
  uint64_t P;  /* pixel-frames                                                      */
  uint64_t rand_calls;
  /* lane-occupancy diagnostics of the HIP kernel (0 in the oracle): 64 x wave iterations */
  uint64_t slots_trace;  /* 64 x (wave loop iterations that ran the hit loop): R / slots_trace = hit-loop lane occupancy */
  uint64_t slots_reject; /* 64 x (rejection rounds run): J / slots_reject = rejection lane occupancy */
} rt_counters;

#ifdef __cplusplus
}
#endif
#endif /* RT_TYPES_H */
