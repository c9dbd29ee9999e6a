// rt_kernels.hip — gfx950 path-tracing kernels behind librt_hip.so.
//
// Replaces the WGSL entry point `main` of the reference (src/shaders/raytrace_kernal.wgsl:72-83)
// and the buffer->cache copy of Raytracer::compute (src/raytracing.rs:195-199). Written for CDNA4,
// not translated from the shader:
//
//  * PERSISTENT WAVES, STREAMED PIXELS. The launch is sized to the chip (a few waves per SIMD), not
//    to the image. A wave pulls 8x8-pixel tiles from a global queue (one atomic per 64 pixels) and
//    hands their pixels to its lanes one by one: a lane that finishes a pixel takes the next one
//    immediately. Lanes therefore stay busy although pixels differ 3x in cost (sky = 1 trace per
//    sample, mirror chains = up to 11), and the chip drains within one pixel's work of the end.
//  * ONE FLAT LOOP PER LANE. Frame loop, sample loop and bounce loop of a pixel are flattened into
//    a loop of trace() iterations with path regeneration. Every lane still consumes its own RNG
//    stream in the reference's order, so results do not depend on which lane renders a pixel.
//  * TWO-PHASE HIT LOOP. Phase 1 walks the spheres with a wave-uniform index: per-sphere constants
//    arrive by scalar loads and are used straight from SGPR pairs, two spheres per v_pk_fma_f32,
//    5-6 VALU instructions per sphere, no branch; the sign bit of a conservative discriminant is
//    shifted into a per-lane 32-sphere bitmask (v_alignbit). Phase 2 lets every lane walk ITS OWN
//    candidate bits (gathered loads), so the expensive root code (sqrt, divisions) runs once per
//    candidate of a lane instead of once per sphere that any lane of the wave might hit.
//  * BOUNDED REJECTION SAMPLING. random_in_unit_sphere (p = pi/6) runs at most `reject_tries`
//    rounds per iteration; lanes still without a direction sit out one trace instead of making 63
//    lanes wait for the unluckiest one.
//  * SKY TILES. A small double-precision kernel proves, per 8x8 tile, that no primary ray of the tile can hit a
//    sphere in the literal arithmetic; the wave that fetches such a tile finishes it on the spot (draws + miss
//    colour, lane = pixel) instead of tracing it.
//  * The accumulation image is read once and written once per pixel and launch, in place.
//
// Two arithmetic modes (rt_types.h). RT_MATH_STRICT: phase 2, shading and accumulation restate
// the oracle's IEEE contract operation by operation (bit-identical images and RNG states); phase 1
// only has to be a superset filter, so both modes share one FMA form with a proven error margin.
// RT_MATH_FAST: FMA, v_rcp/v_rsq/v_sqrt, dot() for length_squared.
// This file must be compiled with -ffp-contract=off: every fused operation is written explicitly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "rt_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }

// raytrace_kernal.wgsl:265-275 (state update) — integer, bit-exact in every mode
#ifndef RT_RNG_MAD64
#define RT_RNG_MAD64 1
#endif
__device__ __forceinline__ uint32_t rng_step(uint32_t rng) {
#if RT_RNG_MAD64
  // rng * 18197 + 182489729 as ONE v_mad_u64_u32 (low half of the 64-bit result): 4.7 issue cycles against
  // 4.35 + 3.1 for v_mul_lo_u32 + v_add_u32 (profiles/r1e_valu_ubench.txt). VOP3 reads one scalar operand at
  // most, so the multiplier travels in a vector register (laundered, or the compiler folds it back into v_mul_lo).
  uint32_t mulv = 18197u;
  asm("" : "+v"(mulv));
  unsigned long long wide, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(wide), "=s"(carry) : "v"(rng), "v"(mulv), "s"(182489729ull));
  rng = (uint32_t)wide;
#else
  rng = rng * 18197u + 182489729u;
#endif
  rng ^= (rng << 15);
  rng ^= (rng >> 22);
  rng ^= (rng << 4);
  return rng;
}
// The two uses of rand() (wgsl:174-180, 281-290) as ONE exact subtraction each. With m = k 2^-23 the
// 23 mantissa bits k of the state, rand() = m exactly, and both -0.5 + m = (k - 2^22) 2^-23 and
// 2 m - 1 = (2k - 2^23) 2^-23 are representable, so every correctly rounded way of computing them gives
// the same bits: (1 + m) - 1.5 and (2 + 2m) - 3, the left operands being bit patterns.
// ((rng & mask) | exponent) as v_bitop3_b32 (table 0xEA = (a & b) | c): 2.8 issue cycles against 4.3 for v_and_or_b32
__device__ __forceinline__ float rng_jitter(uint32_t rng) { return as_f32(__builtin_amdgcn_bitop3_b32(rng, 0x007fffffu, 0x3f800000u, 0xEA)) - 1.5f; }
__device__ __forceinline__ float rng_signed(uint32_t rng) { return as_f32(__builtin_amdgcn_bitop3_b32(rng, 0x007fffffu, 0x40000000u, 0xEA)) - 3.0f; }
#define RT_STEP(dst)        \
  do {                      \
    rng = rng_step(rng);    \
    (dst) = rng;            \
    if (COUNT) c_rand++;    \
  } while (0)

// raytrace_kernal.wgsl:255-261: rng = x ^ y ^ t. The pixel part x ^ y is staged per tile; the frame
// part t = bitcast<u32>(time) * 87189 + 189273 is computed on the host (same wrapping u32 arithmetic).
__device__ __forceinline__ uint32_t rng_seed_pixel(uint32_t u, uint32_t v) {
  uint32_t x = (u << 15) ^ (u * 1729u + 8192374u);
  uint32_t y = (v << 21) ^ (v * 271u + 719827371u);
  return x ^ y;
}

// strict dot: (ax*bx + ay*by) + az*bz, three roundings for the products, two for the sums
__device__ __forceinline__ float dot_strict(float ax, float ay, float az, float bx, float by, float bz) {
  return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ float dot_fast(float ax, float ay, float az, float bx, float by, float bz) {
  return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// Rare-case handling in the helpers below is WAVE-UNIFORM: the common path is computed by every lane,
// and the generic code sits behind `if (ballot(rare))`, one scalar branch. A per-lane `if` would cost
// five to seven scalar instructions of exec-mask bookkeeping per call, and scalar instructions take
// issue slots like vector ones (profiles/r1_valu_ubench.txt, "fma+salu").
#ifndef RT_SMALL_MAX
#define RT_SMALL_MAX 64u  // largest scene (spheres) of the SMALL variants: sphere records in LDS, behind cull (strict), 7 waves per SIMD
#endif
#define RT_RARELY(cond) (__builtin_expect(__ballot(cond) != 0ull, 0))

// Correctly rounded f32 square root (== the oracle's sqrtf). For x >= 2^-96 (and +inf / NaN) this is
// the compiler's own expansion without its denormal pre-scaling and class fix-up: v_sqrt_f32 is
// within 1 ulp, and the two FMA residuals pick the correctly rounded neighbour. Smaller x (0,
// denormals, negatives) takes the generic path.
#ifndef RT_COST_MAX
#define RT_COST_MAX 1
#endif
#ifndef RT_RADIUS_RCP_LDS
#define RT_RADIUS_RCP_LDS 1
#endif
#ifndef RT_SQRT_SIGNS
#define RT_SQRT_SIGNS 1
#endif
#ifndef RT_ROOT2_LAZY
#define RT_ROOT2_LAZY 1
#endif
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
#if RT_SQRT_SIGNS
  // The same decision without compares and selects (two v_cmp + two v_cndmask + the wait states between them): with
  // nrd = dn s - x and nru = up s - x (signs exact: an FMA rounds once, and an exact zero is +0), the result is
  // dn when nrd >= 0, s when nrd < 0 <= nru, up when nru < 0 (then nrd < 0 too) = dn + sign bit(nrd) + sign bit(nru)
  // in units in the last place. +inf and NaN (whose residuals are NaN, of either sign) join the rare path.
  const uint32_t dnb = as_u32(s) - 1u;
  const float nrd = __builtin_fmaf(as_f32(dnb), s, -x), nru = __builtin_fmaf(as_f32(as_u32(s) + 1u), s, -x);
  float r = as_f32(dnb + (as_u32(nrd) >> 31) + (as_u32(nru) >> 31));
  const bool rare = as_u32(x) - 0x0f800000u >= 0x70000000u;  // not in [2^-96, +inf)
#else
  const float dn = as_f32(as_u32(s) - 1u), up = as_f32(as_u32(s) + 1u);
  const float rd = __builtin_fmaf(-dn, s, x), ru = __builtin_fmaf(-up, s, x);
  float r = rd <= 0.0f ? dn : s;
  r = ru > 0.0f ? up : r;
  const bool rare = x < 0x1p-96f;
#endif
  if (RT_RARELY(rare)) {
    if (rare) r = __builtin_sqrtf(x);
  }
  return r;
}

// Correctly rounded f32 division by a divisor whose refined reciprocal is shared. This is the
// compiler's v_div_scale / v_div_fmas / v_div_fixup expansion specialised to operands for which the
// scale is 1 and no fix-up applies: |x|, |y| in [2^-60, 2^60]. rcp_refined(y) = r1 = r0 + r0*(1 - y*r0).
struct rn_divisor { float y, r; bool nice; };
__device__ __forceinline__ rn_divisor make_divisor(float y) {
  rn_divisor d;
  d.y = y;
  const float ay = __builtin_fabsf(y);
  d.nice = ay >= 0x1p-60f && ay <= 0x1p60f;
  const float r0 = __builtin_amdgcn_rcpf(y);
  d.r = __builtin_fmaf(__builtin_fmaf(-y, r0, 1.0f), r0, r0);
  return d;
}
__device__ __forceinline__ float div_core(float x, const rn_divisor& d) {
  float q = x * d.r;
  q = __builtin_fmaf(__builtin_fmaf(-d.y, q, x), d.r, q);
  q = __builtin_fmaf(__builtin_fmaf(-d.y, q, x), d.r, q);
  return q;
}
// x / y for a caller that knows more than the operand ranges: `generic` marks the lanes whose operands may
// leave the range div_core is correctly rounded for (they take the compiler's division), `any_generic` is the
// wave-uniform "some lane does" - both computed once per ray instead of two compares per quotient.
__device__ __forceinline__ float div_known(float x, const rn_divisor& d, bool generic, unsigned long long any_generic) {
  float q = div_core(x, d);
  if (__builtin_expect(any_generic != 0ull, 0)) {  // a ballot: stays in scalar registers, one s_cmp + branch
    if (generic) q = x / d.y;
  }
  return q;
}
// x0 / y, x1 / y and x2 / y
__device__ __forceinline__ void div3_rn(float x0, float x1, float x2, const rn_divisor& d, float& q0, float& q1, float& q2) {
  q0 = div_core(x0, d); q1 = div_core(x1, d); q2 = div_core(x2, d);
  const float lo = __builtin_fminf(__builtin_fminf(__builtin_fabsf(x0), __builtin_fabsf(x1)), __builtin_fabsf(x2));
  const float hi = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)), __builtin_fabsf(x2));
  const bool rare = !(d.nice && lo >= 0x1p-60f && hi <= 0x1p60f);
  if (RT_RARELY(rare)) {
    if (rare) { q0 = x0 / d.y; q1 = x1 / d.y; q2 = x2 / d.y; }
  }
}

__device__ __forceinline__ uint32_t unorm8(float x) {  // rgba8unorm store: clamp (NaN -> 0), *255, RNE
  float c = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
  return (uint32_t)__builtin_rintf(c * 255.0f);
}

// rgba8unorm load, q / 255 (q <= 255): fma(q, c, RN(q * c_lo)) with c = f32(1/255), c_lo = f32(1/255 - c)
// equals the correctly rounded quotient for all 256 inputs (checked exhaustively in tests/test_host.py)
__device__ __forceinline__ float unorm8_to_f32(uint32_t q) {
  const float f = (float)q;
  return __builtin_fmaf(f, 0x1.010102p-8f, f * -0x1.fdfdfep-33f);
}

// Phase 1 of the hit loop for spheres base..base+cnt-1 (base even, cnt <= 32), TWO SPHERES PER
// INSTRUCTION: sph_p1 holds sphere pairs {cx0,cx1, cy0,cy1, cz0,cz1, K'0,K'1} (the scene is padded to a
// multiple of four with spheres that always miss), loaded by scalar loads and used as the SGPR-pair source of
// v_pk_fma_f32; the per-ray operands are broadcast out of VGPR pairs with op_sel. Per pair: 8 v_pk_fma
// (measured 4.35 issue units against 2 x 3.25 for v_fma with an SGPR operand, profiles/r1_valu_ubench.txt),
// [2 v_and_or], 2 v_alignbit. Bit (cnt_pad-1-j) of the result is 1 when sphere base+j is a certain miss,
// cnt_pad = cnt rounded up to a multiple of four.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f32x16* const_f32x16_ptr;
struct p1_ray { f32x2 dx_dy, dz_nod, na_nA, Px_Py, Pz_0; };
// d = s * broadcast(v.SEL1) + broadcast(c.SEL2) and d = s * broadcast(v.SEL) + d; s is an SGPR pair
#define RT_PK_FMA_INIT(d, s, v, sel1, c, sel2) \
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #sel1 "," #sel2 "] op_sel_hi:[1," #sel1 "," #sel2 "]" : "=v"(d) : "s"(s), "v"(v), "v"(c))
#define RT_PK_FMA_ACC(d, s, v, sel) \
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0," #sel ",0] op_sel_hi:[1," #sel ",1]" : "+v"(d) : "s"(s), "v"(v))
template <bool CULL>
__device__ __forceinline__ uint32_t phase1_pair(uint32_t missbits, f32x2 cx, f32x2 cy, f32x2 cz, f32x2 kk, const p1_ray& q) {
  f32x2 nhb, w;
  RT_PK_FMA_INIT(nhb, cz, q.dz_nod, 0, q.dz_nod, 1);  // -hb = c.d - o.d
  RT_PK_FMA_ACC(nhb, cy, q.dx_dy, 1);
  RT_PK_FMA_ACC(nhb, cx, q.dx_dy, 0);
  RT_PK_FMA_INIT(w, kk, q.na_nA, 0, q.na_nA, 1);      // M - a (|o - c|^2 - r^2)
  RT_PK_FMA_ACC(w, cx, q.Px_Py, 0);
  RT_PK_FMA_ACC(w, cy, q.Px_Py, 1);
  RT_PK_FMA_ACC(w, cz, q.Pz_0, 0);
  const f32x2 test = __builtin_elementwise_fma(nhb, nhb, w);
  // BEHIND CULL: w < 0 means the origin is outside the sphere by more than the margin; if then
  // hb > 0 both roots are negative (the sphere lies behind the ray) and hit_sphere rejects them.
  // Proof that the literal arithmetic agrees in DESIGN.md. One v_bitop3: sign = (nhb & w) | test.
  // v_bitop3_b32 (truth table 0xEA = (a & b) | c) issues in 2.8 cycles against 4.3 for v_and_or_b32 (profiles/r1e_valu_ubench.txt)
  const uint32_t key0 = CULL ? __builtin_amdgcn_bitop3_b32(as_u32(nhb.x), as_u32(w.x), as_u32(test.x), 0xEA) : as_u32(test.x);
  const uint32_t key1 = CULL ? __builtin_amdgcn_bitop3_b32(as_u32(nhb.y), as_u32(w.y), as_u32(test.y), 0xEA) : as_u32(test.y);
  missbits = __builtin_amdgcn_alignbit(missbits, key0, 31);  // (missbits << 1) | sign
  return __builtin_amdgcn_alignbit(missbits, key1, 31);
}
// SCALAR FORM (the small-scene variants, round 3): the same eight FMAs per sphere as plain v_fma_f32 / v_fmac_f32 with the sphere
// constant as the scalar operand - twice the instructions of the packed form, but of the class that shares its 4-cycle issue
// window with another wave's plain FP32 / integer instruction, where v_pk_fma_f32 issues alone (profiles/r3_valu_ceiling.txt,
// classes S and X). Same operations in the same order as the packed form: the same bits. C3 strict -4.5 %, C2 -1 %; the
// 1024-sphere scene, whose waves are all in phase 1 at once and have nothing to pair with, +14 % - it keeps the packed form.
#ifndef RT_P1_SCALAR
#define RT_P1_SCALAR 1
#endif
template <bool CULL>
__device__ __forceinline__ uint32_t phase1_one(uint32_t missbits, float cx, float cy, float cz, float kk, const p1_ray& q) {
  float nhb = __builtin_fmaf(cz, q.dz_nod.x, q.dz_nod.y);
  nhb = __builtin_fmaf(cy, q.dx_dy.y, nhb);
  nhb = __builtin_fmaf(cx, q.dx_dy.x, nhb);
  float w = __builtin_fmaf(kk, q.na_nA.x, q.na_nA.y);
  w = __builtin_fmaf(cx, q.Px_Py.x, w);
  w = __builtin_fmaf(cy, q.Px_Py.y, w);
  w = __builtin_fmaf(cz, q.Pz_0.x, w);
  const float test = __builtin_fmaf(nhb, nhb, w);
  const uint32_t key = CULL ? __builtin_amdgcn_bitop3_b32(as_u32(nhb), as_u32(w), as_u32(test), 0xEA) : as_u32(test);
  return __builtin_amdgcn_alignbit(missbits, key, 31);
}
// VGPR FORM (an experiment of round 3, off: -DRT_P1_VGPR=1; MASK shape only). The sphere's constants come from LDS (one ds_read_b128
// per sphere, every lane the same address) and the eight FMAs are v_fma_f32 on vector registers only - class F, which shares its
// issue window with another wave's instruction, where the scalar-operand form is class S (profiles/r3_valu_ceiling.txt). The FMAs
// are written as asm because the compiler turns an accumulating fma into v_fmac_f32, the VOP2 form, which is class S again.
// Measured (profiles/r3_ab_issue_classes.log, "p1v7"): bit-identical, 70 registers, no scratch - and 2.1 % SLOWER on C3, 2.5 % on C2: a
// v_fma_f32 with three vector sources in allocator-chosen banks issues at 2.7 cycles, not 2.05, and twenty ds_read_b128 per iteration
// are not free either. The scalar-operand form stays.
#ifndef RT_P1_VGPR
#define RT_P1_VGPR 0
#endif
#define RT_VFMA_INIT(d, x, y, c) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(c))
#define RT_VFMA_ACC(d, x, y) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y))
template <bool CULL>
__device__ __forceinline__ uint32_t phase1_one_v(uint32_t missbits, f32x4 s, const p1_ray& q) {
  float nhb, w, test;
  RT_VFMA_INIT(nhb, s.z, q.dz_nod.x, q.dz_nod.y);
  RT_VFMA_ACC(nhb, s.y, q.dx_dy.y);
  RT_VFMA_ACC(nhb, s.x, q.dx_dy.x);
  RT_VFMA_INIT(w, s.w, q.na_nA.x, q.na_nA.y);
  RT_VFMA_ACC(w, s.x, q.Px_Py.x);
  RT_VFMA_ACC(w, s.y, q.Px_Py.y);
  RT_VFMA_ACC(w, s.z, q.Pz_0.x);
  RT_VFMA_INIT(test, nhb, nhb, w);
  const uint32_t key = CULL ? __builtin_amdgcn_bitop3_b32(as_u32(nhb), as_u32(w), as_u32(test), 0xEA) : as_u32(test);
  return __builtin_amdgcn_alignbit(missbits, key, 31);
}
// spheres base..base+cnt_pad-1 (base, cnt_pad multiples of 4): one s_load_dwordx16 per four spheres
template <bool CULL, bool SCALAR>
__device__ __forceinline__ uint32_t phase1_missbits(const_f32x16_ptr sph_p1, uint32_t base, uint32_t cnt_pad, const p1_ray& q) {
  uint32_t missbits = 0xffffffffu;
// HYBRID (an experiment of round 3, off): of the sphere pairs of a scalar-form launch, those whose bit is set in RT_P1_PK_MASK (bit =
// pair index within an eight-sphere trip) run in the packed form. The window model said a few packed pairs could pay (9 windows
// nothing shares against 16 that each host a rider); measured on C3, every packed pair costs: one pair in four +1.8 %, two +2.8 %,
// three +4.1 % (profiles/r3_ab_issue_classes.log) - packed instructions also stop the OTHER waves' instructions from pairing
// around them (profiles/r3_valu_ceiling.txt, "path mix").
#ifndef RT_P1_PK_MASK
#define RT_P1_PK_MASK 0x0
#endif
#define RT_P1_PAIR(s, lo, pk)                                                                                                  \
  do {                                                                                                                         \
    if (SCALAR && !(pk)) {                                                                                                     \
      missbits = phase1_one<CULL>(missbits, s[lo], s[lo + 2], s[lo + 4], s[lo + 6], q);                                        \
      missbits = phase1_one<CULL>(missbits, s[lo + 1], s[lo + 3], s[lo + 5], s[lo + 7], q);                                    \
    } else {                                                                                                                   \
      missbits = phase1_pair<CULL>(missbits, __builtin_shufflevector(s, s, lo, lo + 1), __builtin_shufflevector(s, s, lo + 2, lo + 3), \
                                   __builtin_shufflevector(s, s, lo + 4, lo + 5), __builtin_shufflevector(s, s, lo + 6, lo + 7), q); \
    }                                                                                                                          \
  } while (0)
#define RT_P1_FOUR(s, bit0)                                                                                                    \
  do {                                                                                                                         \
    RT_P1_PAIR(s, 0, (RT_P1_PK_MASK >> (bit0)) & 1);                                                                           \
    RT_P1_PAIR(s, 8, (RT_P1_PK_MASK >> ((bit0) + 1)) & 1);                                                                     \
  } while (0)
  const uint32_t n4 = cnt_pad >> 2, b4 = base >> 2;
  uint32_t j = 0;
#ifndef RT_P1_UNROLL
#define RT_P1_UNROLL 1  // 1024 spheres: -1.3 %; 20 spheres: no difference (profiles/r1g_ab_unroll.log)
#endif
#if RT_P1_UNROLL
  for (; j + 2u <= n4; j += 2u) {  // eight spheres per trip: both scalar loads in flight before the first is used
    const f32x16 s0 = sph_p1[b4 + j], s1 = sph_p1[b4 + j + 1u];
    RT_P1_FOUR(s0, 0);
    RT_P1_FOUR(s1, 2);
  }
#endif
  for (; j < n4; ++j) {
    const f32x16 s = sph_p1[b4 + j];  // wave-uniform -> SGPRs
    RT_P1_FOUR(s, 0);
  }
#undef RT_P1_PAIR
#undef RT_P1_FOUR
  return missbits;
}

// Section timing for kernel studies (tools/section_profile.py): -DRT_PROFILE_SECTIONS=1 builds a library whose
// path kernel adds the shader-clock cycles each wave spends in [G], the trace prelude, phase 1, phase 2
// and shading into counters[0..4], its lifetime into [5], its loop iterations into [6] and its lifetime in
// 100 MHz ticks into [7]. Not for production.
#ifndef RT_PROFILE_SECTIONS
#define RT_PROFILE_SECTIONS 0
#endif
#if RT_PROFILE_SECTIONS == 1
// -DRT_PROFILE_ONLY=i times section i alone: two clock reads per iteration instead of twenty (with all five sections
// timed at once the reads themselves are a quarter of the wave's time, most of it booked on the untimed remainder)
#ifndef RT_PROFILE_ONLY
#define RT_PROFILE_ONLY -1
#endif
#define RT_PROF_BEGIN(i) do { if (RT_PROFILE_ONLY < 0 || RT_PROFILE_ONLY == i) prof_b##i = __builtin_amdgcn_s_memtime(); } while (0)
#define RT_PROF_END(i) do { if (RT_PROFILE_ONLY < 0 || RT_PROFILE_ONLY == i) prof_acc##i += __builtin_amdgcn_s_memtime() - prof_b##i; } while (0)
#define RT_PROF_LANES(i)
#elif RT_PROFILE_SECTIONS == 2
// lane occupancy instead of time: counters[i] += active lanes + (1 << 38) per execution of section i
// (0 = [G0] new ray, 1 = [P1] prelude + phase 1, 2 = a rejection round, 3 = scatter, 4 = normalize, 5 = a trip of the walk)
#define RT_PROF_BEGIN(i)
#define RT_PROF_END(i)
#define RT_PROF_LANES(i) do { const unsigned long long m_ = __ballot(true); /* the first active lane counts for the wave */ \
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_, 0u)) == 0u) \
      atomicAdd(&prof_lds[i], (unsigned long long)__popcll(m_) + (1ull << 38)); } while (0)
#elif RT_PROFILE_SECTIONS == 3
// histogram of the candidates a ray brings to the walk: counters[0..4] = primary rays with 1, 2, 3, 4, 5+ candidates,
// counters[8..11] = bounce rays with 1, 2, 3, 4+ (a bounce ray without candidates never reaches the walk)
#define RT_PROF_BEGIN(i)
#define RT_PROF_END(i)
#define RT_PROF_LANES(i)
#else
#define RT_PROF_BEGIN(i)
#define RT_PROF_END(i)
#define RT_PROF_LANES(i)
#endif

// One frame of store_color / combine_pixel_cache_color (wgsl:96-112) on the accumulation pixel `acc`, shared
// by the path kernel (pixel-sequential launches) and rt_combine_frames_kernel (staged launches) so that both
// perform the same operations in the same order. p = this frame's colour (already divided by spp), cw as
// in wgsl:108-109. Returns the rgba8unorm word (RT_ACCUM_RGBA8) after re-quantising acc.
__device__ __forceinline__ uint32_t combine_frame(float& acc0, float& acc1, float& acc2, float& acc3, float p0, float p1, float p2,
                                                  float cw, bool combine_sum, bool accum_f32) {
  const float p3 = 1.0f;
  if (combine_sum) {
    acc0 = acc0 + p0; acc1 = acc1 + p1; acc2 = acc2 + p2; acc3 = acc3 + p3;
  } else {
    const float rw = 1.0f - cw;
    acc0 = p0 * rw + acc0 * cw; acc1 = p1 * rw + acc1 * cw; acc2 = p2 * rw + acc2 * cw; acc3 = p3 * rw + acc3 * cw;
  }
  uint32_t q8 = 0;
  if (!accum_f32) {  // the reference re-quantises to rgba8unorm every frame (raytracing.rs:30,47)
    const uint32_t q0 = unorm8(acc0), q1 = unorm8(acc1), q2 = unorm8(acc2), q3 = unorm8(acc3);
    q8 = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
    acc0 = unorm8_to_f32(q0); acc1 = unorm8_to_f32(q1); acc2 = unorm8_to_f32(q2); acc3 = unorm8_to_f32(q3);
  }
  return q8;
}

// the parked states are 1 and 2 (one unsigned compare), everything runnable is >= ST_READY
enum : uint32_t { ST_DEAD = 0, ST_NEED_PIXEL = 1, ST_FRAME_END = 2, ST_READY = 3, ST_NEED_DIR = 4, ST_NEW_RAY = 5, ST_PATH_DONE = 6, ST_GOT_DIR = 7,
                  ST_TEST = 8 /* SPLIT variants: a ray whose candidates wait for the exact tests of [W] */ };

// STRICT: arithmetic mode. COUNT: event counters for the roofline. SMALL: scenes of at most RT_SMALL_MAX spheres
// (non-counting launches; chosen by rt_launch_path_kernel) - the per-sphere records of phase 2 and of the shading are
// staged in LDS (a gather from LDS returns in ~100 cycles, from L1/L2 in several hundred: C2 strict -5 %, C3 -1 %), and
// strict launches also drop spheres behind the ray in phase 1 (CULL).
// SPLIT (SMALL scenes of at most 32 spheres whose launch has per-tile candidate masks): primary rays are traced by a
// section of their own that takes the tile's mask as its candidates, and a lane can trace its primary ray and its first
// bounce ray in one loop iteration (see [G] / [C] in the loop).
// COLD KERNEL ARGUMENTS. Everything the per-pixel parts need (camera matrix, image geometry, buffers, frame
// table) is read from the kernarg segment where it is used - behind a laundered pointer, so the compiler
// cannot hoist the loads to the kernel entry and then keep ~40 scalar registers alive (and spilled into
// VGPR lanes) across the hot loop. Only what the hit loop and the draws use is taken from `a` by value.
typedef const __attribute__((address_space(4))) rt_kargs* rt_kargs_cptr;
__device__ __forceinline__ rt_kargs_cptr cold_args() {
  rt_kargs_cptr p = (rt_kargs_cptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

template <bool STRICT, bool COUNT, bool SMALL, bool SPLIT>
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 6  // keeps the production variants at <= 80 VGPRs = 6 waves per SIMD (measured: C3 strict -2.7 %)
#endif
// The small-scene variants are held to 72 VGPRs = 7 waves per SIMD: big
// images of small scenes gain 1-2 % from the seventh wave (profiles/r1e_ab_waves.log), the 1024-sphere scene loses 3 %.
#ifndef RT_SPLIT_WAVES
#define RT_SPLIT_WAVES 7
#endif
#ifndef RT_LIST_BATCH
#define RT_LIST_BATCH 56u    // LIST shape: READY lanes the general trace waits for ...
#endif
#ifndef RT_LIST_MAXSKIP
#define RT_LIST_MAXSKIP 4u   // ... for at most this many iterations in a row
#endif
#ifndef RT_WALK_MIN_LANES
#define RT_WALK_MIN_LANES 6u   // MASK shape: lanes that must still hold candidates for a further trip of the walk in the same iteration
#endif
__global__ __launch_bounds__(64, (SMALL && !COUNT) ? (SPLIT ? RT_SPLIT_WAVES : 7) : RT_MIN_WAVES) void rt_path_kernel(const rt_kargs a) {
  // COUNT && SPLIT = the production shape with counters: what the kernel EXECUTES (rt_enable_counters mode 2), as opposed to
  // COUNT && !SPLIT, which traces every tile in the classic shape so that its counters are the reference algorithm's (the oracle's).
  constexpr bool EXEC = COUNT && SPLIT;
  constexpr bool MASK = SPLIT && SMALL;   // <= 32 spheres: the tile's candidates are a bit mask, every section once per iteration
  constexpr bool LIST = SPLIT && !SMALL;  // > 64 spheres: the tile's candidates are a short list; primary and general trace
  // behind cull: strict small-scene launches (fewer exact tests), and every SPLIT launch (there a bounce ray without
  // candidates is finished on the spot, so candidates that only fall to the exact test cost a loop iteration)
  constexpr bool QUEUE = STRICT && !SMALL && (!COUNT || SPLIT);  // the general trace collects a ray's candidates over all blocks before the exact tests
  constexpr bool P1S = SMALL && RT_P1_SCALAR;  // phase 1 in the scalar-operand form (see phase1_one)
  constexpr bool CULL = SMALL && (STRICT || SPLIT);  // (the 1024-sphere scene loses 2 % with it: one more instruction per sphere)
  // The lane index is recomputed where the cold code needs it (two v_mbcnt behind a compiler barrier) instead of
  // living in a register across the hot loop: at 72 registers that one value was the last spill.
  auto lane_id = []() __attribute__((always_inline)) -> uint32_t {
    uint32_t l = __builtin_amdgcn_mbcnt_hi(0xffffffffu, __builtin_amdgcn_mbcnt_lo(0xffffffffu, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };
  const bool checker = (a.flags & RT_SHADE_CHECKER) != 0;
  const bool accum_f32 = (a.flags & RT_ACCUM_F32) != 0;
  const bool combine_sum = (a.flags & RT_COMBINE_SUM) != 0;
  const const_f32x16_ptr sph_p1 = (const_f32x16_ptr)(uintptr_t)a.sph_p1;
  const uint32_t n_spheres = a.n_spheres;
  const float spp_f = (float)a.spp;
  __shared__ f32x4 tile_lds[128];  // per wave: 64 pixel records {pixel_center, v<<16|u} {pixel_delta, seed x^y}
  __shared__ f32x4 sph_lds[SMALL ? 3u * RT_SMALL_MAX : 1u];  // SMALL: sph_test | sph_shade | sph_col of the whole scene
  __shared__ f32x4 p1_lds[(RT_P1_VGPR && SMALL && SPLIT) ? 32u : 1u];  // RT_P1_VGPR: {cx, cy, cz, K'} of the (<= 32, padded) spheres of a MASK launch
  if (RT_P1_VGPR && SMALL && SPLIT) {
    const uint32_t np = (n_spheres + 3u) & ~3u;
    if (threadIdx.x < np && np <= 32u) {
      const float* pp = a.sph_p1 + (threadIdx.x >> 1) * 8u + (threadIdx.x & 1u);
      p1_lds[threadIdx.x] = f32x4{pp[0], pp[2], pp[4], pp[6]};
    }
  }
  if (SMALL) {
    for (uint32_t i = threadIdx.x; i < n_spheres; i += 64u) {
      sph_lds[i] = a.sph_test[i];
      sph_lds[RT_SMALL_MAX + i] = a.sph_shade[i];
      f32x4 col = a.sph_col[i];
#if RT_RADIUS_RCP_LDS
      // strict: a radius that is no power of two (col.w <= 0) is divided by; its refined reciprocal - what make_divisor would
      // compute per hit, v_rcp_f32 and two FMAs - is formed here once per wave and sphere, as -|.| (NaN: outside div_core's range)
      if (STRICT && !(col.w > 0.0f)) {
        const rn_divisor rd = make_divisor(a.sph_shade[i].w);
        col.w = rd.nice ? -__builtin_fabsf(rd.r) : __builtin_nanf("");
      }
#endif
      sph_lds[2u * RT_SMALL_MAX + i] = col;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  uint32_t c_T = 0, c_D = 0, c_H = 0, c_R = 0, c_M = 0, c_S = 0, c_Sm = 0, c_J = 0, c_Sd = 0, c_N = 0, c_P = 0, c_rand = 0;
  uint32_t c_wt = 0, c_wr = 0;  // wave-level iteration counts (diagnostics)
  uint32_t c_X = 0, c_sky = 0;  // EXEC: exact tests run (test_sphere calls); samples finished in sky tiles

  // wave-uniform work queue: pixel slots [q_cur, q_end) of the tile this wave is handing out.
  // slot = (index of the tile in this rank's tile list) * 64 + (pixel within the 8x8 tile)
  uint32_t q_cur = 0, q_end = 0;
  bool more = true;
  uint32_t q_next = a.queue_chunk == 0u ? blockIdx.x : 0u, q_chunk_end = 0;  // wave-uniform: the units [q_next, q_chunk_end) this wave has taken from the queue and not handed out yet (queue_chunk 0: q_next strides)

  // per-lane state machine:
  //   NEED_PIXEL -[A]-> NEW_RAY -[G]-> READY -[C]-> PATH_DONE (miss / depth) | READY (mirror) | NEED_DIR (diffuse)
  //   NEED_DIR -[G]-> GOT_DIR -> READY | PATH_DONE ;  PATH_DONE -[E]-> NEW_RAY | FRAME_END ;  FRAME_END -[F]-> NEW_RAY | NEED_PIXEL ;
  //   NEED_PIXEL -[A]-> DEAD when the queue is empty
  uint32_t state = ST_NEED_PIXEL;
  uint32_t uvp = 0;          // pixel coordinates v << 16 | u (the image index v * width + u is formed where it is used: cold code)
  uint32_t seed_xy = 0;      // pixel part of the RNG seed
  uint32_t my_tile = 0;      // index (in this rank's tile list) of the tile the pixel belongs to
  uint32_t iters = 0;        // trace iterations spent on the current pixel (cost feedback for the tile order)
  uint32_t q_tile = 0, q_tile_new = 0, q_frame = 0;  // wave-uniform: tile (and, staged, frame) being handed out
  uint32_t q_mask = 0, q_mask_new = 0xffffffffu;     // wave-uniform: candidate mask of that tile (SPLIT)
  uint32_t list_skips = 0;                           // wave-uniform (LIST shape): iterations the general trace has waited for more READY lanes
  uint32_t tmask = 0;                                // candidate spheres of the primary rays of this lane's pixel (SPLIT)
  float pcx = 0, pcy = 0, pcz = 0, pdx = 0, pdy = 0, pdz = 0;  // pixel_center, pixel_delta
  uint32_t k = 0, rng = 0, sample = 0, depth = 0;
  float sumx = 0, sumy = 0, sumz = 0;                          // send_rays' `color`
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;        // ray (in NEED_DIR / GOT_DIR: d holds the normal)
  float colx = 0, coly = 0, colz = 0, refl = 1.0f;             // ray_color's `color`, `reflectance`
  float gqx = 0, gqy = 0, gqz = 0, gl2 = 1.0f;                 // accepted point of random_in_unit_sphere, |q|^2

#if RT_PROFILE_SECTIONS
  unsigned long long prof_acc0 = 0, prof_acc1 = 0, prof_acc2 = 0, prof_acc3 = 0, prof_acc4 = 0, prof_acc5 = 0, prof_iters = 0, prof_p2_waves = 0, prof_p2_lanes = 0;
  unsigned long long prof_live0 = 0, prof_live1 = 0, prof_live2 = 0, prof_live3 = 0;
  unsigned long long prof_b0 = 0, prof_b1 = 0, prof_b2 = 0, prof_b3 = 0, prof_b4 = 0, prof_b5 = 0;
  const unsigned long long prof_start = __builtin_amdgcn_s_memtime(), prof_rstart = __builtin_amdgcn_s_memrealtime();
#endif
#if RT_PROFILE_SECTIONS >= 2
  __shared__ unsigned long long prof_lds[16];
  if (threadIdx.x < 16u) prof_lds[threadIdx.x] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
  const uint32_t n_pad32 = (n_spheres + 3u) & ~3u;  // SPLIT (<= 32 spheres): the one block of the candidate walk
  // ---- the pieces of trace (wgsl:182-231) and of miss / scatter (wgsl:144-172); the two loop shapes below compose them ----
  // Per-ray values that outlive the piece that computes them (in the SPLIT shape: across the loop's back edge)
  float a_lit = 1.0f, a_rcp = 1.0f;  // STRICT: length_squared(dir) (wgsl:306-309) and its refined reciprocal; FAST: |d|^2 and v_rcp of it
  bool tame = true;                  // the ray is inside the magnitude range of the phase-1 bound and of div_core
  uint32_t cand = 0;                 // spheres still to be tested exactly, in the bit layout of phase1_missbits
  uint32_t own = 0;                  // SPLIT: the sphere whose surface the current bounce ray starts on (its last hit)
  // MASK shape: `closest` and `hit` of a candidate walk that continues in the next iteration. They share registers with
  // values that are dead while a lane is in ST_TEST: gl2 (|q|^2 of an accepted point, live NEED_DIR -> GOT_DIR only) and
  // own (consumed by the own-sphere drop at the end of [P1], set again by the next scatter).
  float& w_closest = gl2;
  uint32_t& w_hit = own;

  // OWN-SPHERE DROP (SPLIT shape). A bounce ray starts ON the sphere it just hit, so phase 1 can never exclude that
  // sphere (its discriminant is ~ hb^2 >= 0) - and the exact test then rejects both roots almost always: they are ~ 0.
  // This proves the rejection from two dot products. Literal test of sphere `own` for the ray (o, d), wgsl:207-223:
  //   oc = o - c (the three subtractions below ARE the literal ones), hb = dot(oc, d), x = dot(oc, oc) (both computed
  //   as the literal code computes them), c_l = fl(fl(fl(sqrt(x))^2) - fl(r r)), disc = fl(fl(hb^2) - fl(a c_l)),
  //   roots fl(fl(-hb -+ sqrt_rn(disc)) / a). u = 2^-24.
  // Let hb >= 2^-30 and the ray and scene be `tame` (a in [2^-21, 2^21]): hb <= |oc| |d| gives x >= 2^-81, and nothing
  // below under- or overflows. root1 has a negative numerator: rejected. For root2:
  //  * c_l >= 0: fl(a c_l) >= 0, so disc <= fl(hb^2) and sqrt_rn(disc) <= sqrt_rn(fl(hb^2)) = hb (radix 2): the
  //    numerator -hb + sq is <= 0 and root2 <= 0 <= tmin: rejected (or disc < 0: rejected at once).
  //  * c_l < 0, delta = -c_l: fl(len^2) >= x (1 - u)^3, so rr > x (1 - 3u): rr - x is exact (Sterbenz; or rr > 2x, and
  //    then dub > x and the test below fails), T = (rr - x) + 4u x > 0.99 u x > 0, dub = fl(T) >= T (1 - u), and
  //    delta <= (rr - x + 3.01 u x)(1 + u) <= T (1 + u) <= dub (1 + 4u). [So dub <= 0 proves c_l >= 0.]
  //    sq <= sqrt(hb^2 + a delta)(1 + u)^2 <= (hb + a delta / (2 hb))(1 + 2.01u), hence
  //    root2 <= [delta / (2 hb) + 2.1 u hb / a](1 + 6u), which is below tmin when
  //        a dub (1 + 4u) + 4.2 u hb^2 < 2 hb a tmin / (1 + 6u).
  //    The test evaluates  fl(fl(hb^2) 5u + fl(a dub)) <= fl(fl(hb a) * 2 tmin (1 - 2^-10)): five roundings of relative
  //    size u each, against the slack of 2^-10 on the right and of 0.8 u hb^2 on the left.
  // So under `drop` hit_sphere returns false for `own` in the literal arithmetic (it has no side effect then), and the
  // sphere leaves the candidate set. With one inequality instead of separate limits on dub / hb and hb / a, rays that
  // leave a BIG sphere at a grazing angle are covered too (the mirror ground of the default scene, r = 100: x ~ 10^4,
  // dub ~ 2.4e-3 +- a few ulp of x, so every departure with hb >~ 2 - 1.2 degrees above the tangent plane - is dropped;
  // the separate limits needed 11 degrees, and most of the visible ground is seen at less). What keeps its exact test
  // is the last degree above the tangent plane, where the reference's own "shadow acne" lives.
  // `a_d`: |d|^2 - the literal a_lit, or (MASK shape) the FMA form a_fast, which is within 8u of it (each is within a few u
  // of the exact |d|^2): both a-terms of the inequality move by that factor, the 2^-10 on the right is 16384u.
  auto own_sphere_drop = [&](float a_d) __attribute__((always_inline)) -> bool {
    const bool has_own = own != 0xffffffffu;  // LIST: a primary ray that takes the general trace (its tile's list overflowed)
    const uint32_t oi = has_own ? own : 0u;
    const f32x4 so = SMALL ? sph_lds[oi] : a.sph_test[oi];  // {c, fl(r r)}
    const float ex = ox - so.x, ey = oy - so.y, ez = oz - so.z;
    const float x = STRICT ? dot_strict(ex, ey, ez, ex, ey, ez) : dot_fast(ex, ey, ez, ex, ey, ez);
    const float hbo = STRICT ? dot_strict(ex, ey, ez, dx, dy, dz) : dot_fast(ex, ey, ez, dx, dy, dz);
    const float dub = __builtin_fmaf(x, 0x1p-22f, so.w - x);
    const float lhs = __builtin_fmaf(hbo * hbo, 5.0f * 0x1p-24f, a_d * dub);
    const float rhs = (hbo * a_d) * (2.0f * RT_RAY_TMIN * (1.0f - 0x1p-10f));
    return has_own && tame && hbo >= 0x1p-30f && lhs <= rhs;
  };

  // [P] per-ray prelude; FILTER: also the per-ray terms of the phase-1 filter ("phase-1 filter"):
  // exact discriminant D = hb^2 - a (|o - c|^2 - r^2), hb = o.d - c.d. With the margin
  // M = m a (|o|^2 + |c|^2 + r^2), m = RT_P1_MARGIN:
  //   D + M = (o.d - c.d)^2 + [(m - 1) a |o|^2] + (2 a o).c - a K',  K' = (|c|^2 - r^2) - m (|c|^2 + r^2)
  // K' is a per-sphere constant (rt_set_scene), everything else per-ray: 8 FMAs per sphere. The
  // computed value and the oracle's literal discriminant each stay within 51 resp. 54 units of
  // 2^-24 a (|o|^2 + |c|^2 + r^2) of D while all magnitudes are `tame` (no overflow / underflow);
  // m = 2^-16 = 256 such units, so a negative sign here is a certain miss of the literal code too.
  // Rays or scenes outside the tame range skip the filter (every sphere is a candidate).
#ifndef RT_RCP_AT_WALK
#define RT_RCP_AT_WALK 1
#endif
  // what the exact tests need of a ray: length_squared(dir) as the shader computes it (wgsl:306-309: sqrt, then square) and
  // its refined reciprocal
  auto ray_literal_a = [&]() __attribute__((always_inline)) {
    if (STRICT) {
      const float len = sqrt_rn(dot_strict(dx, dy, dz, dx, dy, dz));
      a_lit = len * len;
      a_rcp = make_divisor(a_lit).r;
    } else {
      a_lit = dot_fast(dx, dy, dz, dx, dy, dz);
      a_rcp = __builtin_amdgcn_rcpf(a_lit);
    }
  };
  // returns the FMA form of |d|^2
  auto ray_prelude = [&](auto filter_tag, p1_ray& p1q) __attribute__((always_inline)) -> float {
    constexpr bool FILTER = decltype(filter_tag)::value;
    const float a_fast = dot_fast(dx, dy, dz, dx, dy, dz);
    // a primary ray (the preludes without FILTER) starts at the camera: |o|^2 <= 2^40 is bit 1 of p1_tame, decided by the host
    const float oo = FILTER ? dot_fast(ox, oy, oz, ox, oy, oz) : 0.0f;
    tame = (FILTER ? (a.p1_tame & 1u) != 0u : (a.p1_tame & 2u) != 0u) && a_fast >= 0x1p-20f && a_fast <= 0x1p20f && oo <= 0x1p40f;
    if (FILTER) {
      const float nod = -dot_fast(ox, oy, oz, dx, dy, dz);
      const float na = -a_fast;
      const float nA = (a_fast * oo) * (RT_P1_MARGIN - 1.0f);
      const float a2 = a_fast + a_fast;
      p1q = p1_ray{f32x2{dx, dy}, f32x2{dz, nod}, f32x2{na, nA}, f32x2{a2 * ox, a2 * oy}, f32x2{a2 * oz, 0.0f}};
    }
    // MASK shape: a_lit and its reciprocal (a correctly rounded sqrt, v_rcp_f32 and two FMAs: ~20 instructions) are formed where
    // the exact tests run ([W]: once per iteration for primary and bounce rays together - and only 29 % of the bounce rays
    // get there at all) instead of in both preludes; the own-sphere drop works with a_fast.
    if (!MASK || !RT_RCP_AT_WALK) ray_literal_a();
    return a_fast;
  };

  // [W] hit_sphere (wgsl:200-231) for the lane's candidates among spheres base .., in ascending sphere order.
  // STRICT roots (-hb -+ sq) / a_lit of a tame ray: |hb|, sq < 2^32 and a_lit in [2^-21, 2^21], so the numerator
  // is below div_core's upper limit, and a numerator under its lower limit 2^-60 gives |quotient| < 2^-38 both
  // exactly and from div_core - far below RT_RAY_TMIN, where only the comparison `root <= tmin` is used.
  // Untamed rays take the compiler's division.
  // `beyond(root)`: hit_sphere's `closest <= root` (wgsl:216,220). The shader meets the spheres in ascending index, so among equal
  // roots the lowest index wins; every sphere has ONE root it can be accepted with, whatever `closest` is when it is tested
  // (root1 if root1 > tmin, else root2 - see RT_ROOT2_LAZY), so the loop's result is (smallest such root, lowest index among
  // equals) and may be computed in any order with the index compared on ties: the big-scene variants do (ANYORDER), because the
  // grid trace meets a ray's candidates cell by cell, not by index (and may meet one twice: the second test changes nothing).
  constexpr bool ANYORDER = !SMALL;
  auto test_sphere = [&](uint32_t i, float& closest, int& hit, unsigned long long any_untame) __attribute__((always_inline)) {
    const rn_divisor a_div = {a_lit, a_rcp, false};
    auto beyond = [&](float root) __attribute__((always_inline)) -> bool {
      return ANYORDER ? (closest < root || (closest == root && (int)i >= hit)) : closest <= root;
    };
    if (EXEC) c_X++;
    {
      const f32x4 s = SMALL ? sph_lds[i] : a.sph_test[i];
      const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
      if (STRICT) {
        // hit_sphere (wgsl:200-231), literal
        const float hb = dot_strict(ocx, ocy, ocz, dx, dy, dz);
        const float len = sqrt_rn(dot_strict(ocx, ocy, ocz, ocx, ocy, ocz));
        const float c = len * len - s.w;
        const float disc = hb * hb - a_lit * c;
        if (!(disc < 0.0f)) {
          if (COUNT) c_D++;
          const float sq = sqrt_rn(disc);
          float root = div_known(-hb - sq, a_div, !tame, any_untame);
          bool ok = true;
#if RT_ROOT2_LAZY
          // The second root is only worth a division when the first one lies behind tmin: for a tame ray (a_lit > 0,
          // hb and sq finite, sq >= +0) fl(-hb + sq) >= fl(-hb - sq) and the correctly rounded quotients by a_lit keep
          // that order, so `closest <= root1` implies `closest <= root2` - hit_sphere (wgsl:214-222) rejects both.
          const bool behind1 = root <= RT_RAY_TMIN;
          if (behind1 || beyond(root)) {
            ok = false;
            if (behind1 || !tame) {
              root = div_known(-hb + sq, a_div, !tame, any_untame);
              ok = !(root <= RT_RAY_TMIN || beyond(root));
            }
          }
#else
          if (root <= RT_RAY_TMIN || beyond(root)) {
            root = div_known(-hb + sq, a_div, !tame, any_untame);
            if (root <= RT_RAY_TMIN || beyond(root)) ok = false;
          }
#endif
          if (ok) { closest = root; hit = (int)i; }
        }
      } else {
        const float hb = dot_fast(ocx, ocy, ocz, dx, dy, dz);
        const float cc = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, __builtin_fmaf(ocx, ocx, -s.w)));
        const float disc = __builtin_fmaf(-a_lit, cc, hb * hb);
        if (!(disc < 0.0f)) {  // the filter's margin lets a few near misses through
          if (COUNT) c_D++;
          const float sq = __builtin_amdgcn_sqrtf(disc);
          const float r0 = (-hb - sq) * a_rcp, r1 = (-hb + sq) * a_rcp;
          auto nearer = [&](float r) __attribute__((always_inline)) -> bool {
            return ANYORDER ? (r < closest || (r == closest && (int)i < hit)) : r < closest;
          };
          const bool ok0 = r0 > RT_RAY_TMIN && nearer(r0);
          const float root = ok0 ? r0 : r1;
          if (root > RT_RAY_TMIN && nearer(root)) { closest = root; hit = (int)i; }
        }
      }
    }
  };
  auto walk = [&](uint32_t base, uint32_t cnt_pad, float& closest, int& hit, unsigned long long any_untame) __attribute__((always_inline)) {
    while (cand) {
#if RT_PROFILE_SECTIONS == 1
      prof_p2_waves++; prof_p2_lanes += (unsigned long long)__popcll(__ballot(true));  // (lane 0's view: its own trips)
#endif
      RT_PROF_LANES(5);
      const uint32_t lz = (uint32_t)__builtin_clz(cand);
      cand &= ~(0x80000000u >> lz);
      test_sphere(base + cnt_pad - 32u + lz, closest, hit, any_untame);
    }
  };

  // [M] miss (wgsl:144-148); ray_color: color += miss * reflectance; break
  auto shade_miss = [&]() __attribute__((always_inline)) {
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
    state = ST_PATH_DONE;
  };

  // [S] hit record of the closest sphere (wgsl:225-227) + scatter (wgsl:150-172)
  auto shade_hit = [&](float closest, int hit) __attribute__((always_inline)) {
    if (COUNT) { c_H++; c_S++; }
    const f32x4 sg = SMALL ? sph_lds[RT_SMALL_MAX + (uint32_t)hit] : a.sph_shade[hit];      // {cx, cy, cz, r}
    const f32x4 sc = SMALL ? sph_lds[2u * RT_SMALL_MAX + (uint32_t)hit] : a.sph_col[hit];   // {material r, g, b, +-1/r}
    // point = ray_at(t), normal = (point - c) / r
    float px, py, pz, nx, ny, nz;
    if (STRICT) {
      px = ox + closest * dx; py = oy + closest * dy; pz = oz + closest * dz;
      const float ex = px - sg.x, ey = py - sg.y, ez = pz - sg.z;
      if (sc.w > 0.0f) {
        // r is a power of two (rt_set_scene marks that with a positive 1/r): x / r == x * (1/r)
        // exactly, barring overflow/underflow of a unit-length normal's components
        nx = ex * sc.w; ny = ey * sc.w; nz = ez * sc.w;
      } else {
        rn_divisor rd;
        if (SMALL && RT_RADIUS_RCP_LDS) {  // the reciprocal staged in LDS by the wave's prologue (sign of the radius restored)
          rd.y = sg.w; rd.r = __builtin_copysignf(sc.w, sg.w); rd.nice = sc.w == sc.w;
        } else {
          rd = make_divisor(sg.w);
        }
        div3_rn(ex, ey, ez, rd, nx, ny, nz);
      }
    } else {
      px = __builtin_fmaf(closest, dx, ox); py = __builtin_fmaf(closest, dy, oy); pz = __builtin_fmaf(closest, dz, oz);
      const float ir = __builtin_fabsf(sc.w);
      nx = (px - sg.x) * ir; ny = (py - sg.y) * ir; nz = (pz - sg.z) * ir;
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
    if (SPLIT) own = (uint32_t)hit;
    depth++;
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
      state = depth > (uint32_t)RT_MAX_RAY_DEPTH ? ST_PATH_DONE : ST_READY;
    } else {
      // normal + random_unit_vector(): the direction is drawn in [G]; park the normal in d
      if (COUNT) c_Sd++;
      dx = nx; dy = ny; dz = nz;
      state = ST_NEED_DIR;
    }
  };

  // one rejection round of random_in_unit_sphere (wgsl:281-290) for the lanes in NEED_DIR
  // STRICT: accept iff fl(fl(sqrt(l2))^2) < 1 (wgsl:284,306-309). fl(sqrt(.)) is monotone with
  // fl(sqrt(1)) = 1, and for l2 <= 1 - 2^-24 the rounded root is <= 1 - 2^-24, whose square
  // rounds below 1: the test equals l2 < 1 exactly.
  auto try_direction = [&](uint32_t s0, uint32_t s1, uint32_t s2) __attribute__((always_inline)) {
    if (COUNT) c_J++;
    const float qx = rng_signed(s0), qy = rng_signed(s1), qz = rng_signed(s2);  // rand() * 2 - 1
    const float l2 = STRICT ? dot_strict(qx, qy, qz, qx, qy, qz) : dot_fast(qx, qy, qz, qx, qy, qz);
    gqx = qx; gqy = qy; gqz = qz; gl2 = l2;  // a rejected point is dead: no select needed
    if (l2 < 1.0f) state = ST_GOT_DIR;
  };
  // get_random_ray (wgsl:174-180) from the two jitter draws; ray_color's start values (wgsl:126-130)
  auto start_ray = [&](uint32_t s0, uint32_t s1) __attribute__((always_inline)) {
    const float px = rng_jitter(s0), py = rng_jitter(s1);
    dx = pcx + px * pdx; dy = pcy + py * pdy; dz = pcz + 1.0f * pdz;
    ox = a.pos[0]; oy = a.pos[1]; oz = a.pos[2];
    colx = coly = colz = 0.0f; refl = 1.0f; depth = 0;
    if (COUNT) c_N++;
  };
  // rec.normal + normalize(p) (wgsl:165,292-294), once per accepted lane after all rounds
  auto finish_direction = [&]() __attribute__((always_inline)) {
    if (STRICT) {
      // the components of an accepted point are 0 or multiples of 2^-22 in (-1, 1) and |p| >= 2^-22 unless p = 0:
      // all inside div_core's range (a +0 component gives +0 like the IEEE quotient); p = 0 (0/0) goes the generic way
      const rn_divisor len = make_divisor(sqrt_rn(gl2));
      const bool zero_len = !(gl2 >= 0x1p-50f);
      const unsigned long long any_zero_len = __ballot(zero_len);
      const float ux = div_known(gqx, len, zero_len, any_zero_len), uy = div_known(gqy, len, zero_len, any_zero_len),
                  uz = div_known(gqz, len, zero_len, any_zero_len);
      dx = dx + ux; dy = dy + uy; dz = dz + uz;
    } else {
      const float il = __builtin_amdgcn_rsqf(gl2);
      dx = __builtin_fmaf(gqx, il, dx); dy = __builtin_fmaf(gqy, il, dy); dz = __builtin_fmaf(gqz, il, dz);
    }
    state = depth > (uint32_t)RT_MAX_RAY_DEPTH ? ST_PATH_DONE : ST_READY;  // for (depth = 0; depth <= MAX_RAY_DEPTH; depth++) (wgsl:131)
  };
  const uint32_t valid32 = (n_pad32 >= 32u ? 0xffffffffu : (1u << n_pad32) - 1u) & ~((1u << (n_pad32 - (n_spheres < 32u ? n_spheres : 32u) & 31u)) - 1u);

  // PROBE (-DRT_PROBE_BPERMUTE=1, tools/r3_ab.sh): what moving a ray to another lane costs. The seven dwords phase 1 needs of a
  // ray (o, d, own) go through ds_bpermute_b32 with the lane's own index - the identity, so every result bit stays - in front
  // of phase 1: the floor under the cost of ANY cross-lane compaction of the READY lanes (which also needs the destination
  // lanes - a ballot and two v_mbcnt - and the way back, or the rest of the path's state). profiles/r3_compaction_probe.md.
#ifndef RT_PROBE_BPERMUTE
#define RT_PROBE_BPERMUTE 0
#endif
  auto probe_bpermute = [&]() __attribute__((always_inline)) {
#if RT_PROBE_BPERMUTE
    const int self = (int)(lane_id() << 2);
    ox = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(ox)));
    oy = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(oy)));
    oz = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(oz)));
    dx = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(dx)));
    dy = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(dy)));
    dz = as_f32((uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)as_u32(dz)));
    own = (uint32_t)__builtin_amdgcn_ds_bpermute(self, (int)own);
#endif
  };
  // [C] the general trace: prelude, then per block of 32 spheres phase 1 and the walk of its candidates, then miss or scatter
  auto trace_general = [&]() __attribute__((always_inline)) {
    if (COUNT && __ballot(state == ST_READY)) c_wt++;
    if (state == ST_READY) {
      RT_PROF_LANES(1);
      RT_PROF_BEGIN(1);
      iters++;
      float closest = RT_RAY_TMAX;
      int hit = -1;
      p1_ray p1q;
      probe_bpermute();
      ray_prelude(std::integral_constant<bool, true>{}, p1q);
      const unsigned long long any_untame = __ballot(!tame);
      const uint32_t untame_fill = tame ? 0u : 0xffffffffu;
      const bool own_drop = LIST ? own_sphere_drop(a_lit) : false;
      if (COUNT) c_R++;  // (c_T: below, once it is known whether the lane's filter runs over all spheres or along the grid)
      // CANDIDATE QUEUE (strict launches on scenes of more than RT_SMALL_MAX spheres). A ray of the 1024-sphere scene has
      // 1.4 candidates in all, spread over 32 blocks: walking them block by block ran 18 trips of the exact test per
      // loop iteration with 2.5 of 64 lanes in each (profiles/r2b_section_lanes.log) - a fifth of the kernel's
      // instructions. Instead a lane queues the sphere indices phase 1 leaves it (four 16-bit entries in a register
      // pair, oldest on top) and the wave tests them when the trace is through all blocks - or earlier, when some
      // lane's queue is full: 3-4 trips with a third of the lanes in each. Every lane still meets its candidates in
      // ascending sphere order (hit_sphere's tie rule). In fast arithmetic a trip is 25 instructions and queueing would
      // cost what it saves.
      unsigned long long queue = 0ull;
      uint32_t qn = 0;
      const bool use_queue = n_spheres <= 0xffffu;
      auto drain = [&]() __attribute__((always_inline)) {
        while (qn) {
#if RT_PROFILE_SECTIONS == 1
          prof_p2_waves++; prof_p2_lanes += (unsigned long long)__popcll(__ballot(true));
#endif
          RT_PROF_LANES(5);
          qn--;
          test_sphere((uint32_t)(queue >> (16u * qn)) & 0xffffu, closest, hit, any_untame);
        }
      };
      RT_PROF_END(1);
      // GRID TRACE (scenes of more than RT_SMALL_MAX spheres for which rt_set_scene built a grid; not the counting launches of
      // the reference algorithm). A ray of the 1024-sphere scene passes near a handful of spheres, yet the loop below runs the
      // filter on all of them for every lane. Here each lane walks the cells of a uniform grid along ITS ray (3D-DDA in cell
      // coordinates) and runs the same filter - the eight FMAs of phase1_one, sphere constants gathered per lane - on the
      // spheres listed in those cells only; what passes is queued / tested exactly as below. Why nothing is lost: a sphere's
      // literal discriminant can be non-negative only if the ray's line comes within sqrt(r^2 + 54 u S) of its centre; every
      // such sphere is listed in each cell its ball of radius sqrt(r^2 + 2^-18 S_max) + slack touches (rt_api.cpp, build_grid),
      // and the cells the DDA visits cover the ray's points inside the grid's box to within that slack (its rounding moves
      // a crossing by < 2^-18 of a cell; a corner cell skipped because two crossings tie is touched by the ray only that
      // close to the corner). Spheres arrive out of index order, possibly twice: see ANYORDER. Lanes the grid cannot serve
      // (untamed rays, origins beyond grid_o2max) take the loop over all spheres - `blocks` - as before.
      bool blocks = true;  // this lane still needs the loop over all spheres
      if (!SMALL && !(COUNT && !SPLIT)) {
        const rt_kargs_cptr ga = cold_args();
        if (ga->grid_cells != nullptr) {
          const bool served = tame && dot_fast(ox, oy, oz, ox, oy, oz) <= ga->grid_o2max;
          blocks = !served;
          // the ray in cell coordinates g + t e; the box is [0, n] per axis
          const float gx = (ox - ga->grid_org[0]) * ga->grid_inv[0], gy = (oy - ga->grid_org[1]) * ga->grid_inv[1], gz = (oz - ga->grid_org[2]) * ga->grid_inv[2];
          const float ex = dx * ga->grid_inv[0], ey = dy * ga->grid_inv[1], ez = dz * ga->grid_inv[2];
          const uint32_t nx = ga->grid_dim[0], ny = ga->grid_dim[1], nz = ga->grid_dim[2];
          const float inf = __builtin_inff();
          bool act = served;
          float t0 = 0.0f, t1 = inf;
          float rx, ry, rz;  // 1 / e per axis (inf for an axis the ray does not move along)
          auto slab = [&](float g, float e, float n, float& r) __attribute__((always_inline)) {
            const bool par = !(__builtin_fabsf(e) >= 0x1p-60f);
            r = par ? inf : __builtin_amdgcn_rcpf(e);
            const float ta = (0.0f - g) * r, tb = (n - g) * r;
            const float tn = par ? -inf : __builtin_fminf(ta, tb), tf = par ? inf : __builtin_fmaxf(ta, tb);
            if (par && !(g >= 0.0f && g <= n)) act = false;
            t0 = __builtin_fmaxf(t0, tn); t1 = __builtin_fminf(t1, tf);
          };
          slab(gx, ex, (float)nx, rx); slab(gy, ey, (float)ny, ry); slab(gz, ez, (float)nz, rz);
          act = act && t0 <= t1;
          // first cell: the one that holds the point at t0 (clamped into the box), and per axis the t of the next crossing
          auto first = [&](float g, float e, float r, uint32_t n, float& tm, int& ic) __attribute__((always_inline)) {
            const float pc = __builtin_fminf(__builtin_fmaxf(__builtin_fmaf(e, t0, g), 0.0f), (float)n - 0.5f);
            ic = (int)pc;  // floor: pc >= 0
            tm = r == inf ? inf : ((float)(ic + (e > 0.0f ? 1 : 0)) - g) * r;
          };
          float tmx, tmy, tmz; int ix, iy, iz;
          first(gx, ex, rx, nx, tmx, ix); first(gy, ey, ry, ny, tmy, iy); first(gz, ez, rz, nz, tmz, iz);
          // ONE FLAT LOOP per wave instead of "for each cell: for each entry" (where every cell cost the wave as many trips as its
          // fullest list among 64 rays in 64 different cells): per trip a lane EITHER takes the next entry of its cell's list (`cur`,
          // a byte offset into the grid's buffer; 0: none) OR steps to the next cell along its ray (`coff`, the byte offset of the
          // cell's table entry; DONE past the box), whose list offset was fetched one step ahead (`nxt`). Offsets are 32-bit
          // against a scalar base: no per-lane 64-bit address arithmetic.
          const char* const gbase = reinterpret_cast<const char*>(ga->grid_cells);
          const char* const cbase = gbase + ga->grid_cell0;  // the cell table; `coff` is relative to it
          const uint32_t coff_last = 4u * (nx * ny * nz - 1u), DONE = 0xffffffffu;
          const int sxs = ex > 0.0f ? 4 : -4, sys = ey > 0.0f ? (int)(4u * nx) : -(int)(4u * nx), szs = ez > 0.0f ? (int)(4u * nx * ny) : -(int)(4u * nx * ny);
          const float t_end = t1 * (1.0f + 0x1p-16f) + 0x1p-30f;  // walk while the entry of the next cell is not past the box's far side
          auto cell_list = [&](uint32_t co) __attribute__((always_inline)) -> uint32_t {
            const uint32_t c = co <= coff_last ? co : coff_last;  // (a rounding step off the box - below 0 wraps - reads a valid cell: more candidates, never fewer)
            return *reinterpret_cast<const uint32_t*>(cbase + c);
          };
          uint32_t coff = act ? 4u * (uint32_t)((iz * (int)ny + iy) * (int)nx + ix) : DONE;
          uint32_t cur = 0u, nxt = 0u;
          if (act) nxt = cell_list(coff);
          // a candidate the lane queued last is not queued again (a sphere is listed in every cell it touches, and a ray that comes
          // near it usually crosses several of them): testing a sphere twice changes nothing, so neither does leaving the repeat out
          uint32_t lastj = own_drop ? own : 0xffffffffu;
          const uint32_t skipj = lastj;
          RT_PROF_BEGIN(2);
          for (;;) {
            if (cur != 0u || coff != DONE) { RT_PROF_LANES(2); }  // (profile builds: lanes still walking in this trip)
            if (cur == 0u && coff != DONE) {  // into the cell `coff`, and the crossing that leaves it
              cur = nxt;
              const float tc = __builtin_fminf(__builtin_fminf(tmx, tmy), tmz);
              const bool bx = tmx == tc, by = !bx && tmy == tc;
              const int inc = bx ? sxs : (by ? sys : szs);
              tmx += bx ? __builtin_fabsf(rx) : 0.0f; tmy += by ? __builtin_fabsf(ry) : 0.0f; tmz += (!bx && !by) ? __builtin_fabsf(rz) : 0.0f;
              coff = tc <= t_end ? coff + (uint32_t)inc : DONE;
              if (coff != DONE) nxt = cell_list(coff);
            }
            bool pass = false;
            uint32_t j = 0;
            if (cur != 0u) {
              RT_PROF_LANES(4);  // (profile builds: lanes that take a list entry in this trip)
              const f32x4 sc = *reinterpret_cast<const f32x4*>(gbase + cur);
              const uint32_t jw = *reinterpret_cast<const uint32_t*>(gbase + cur + 16u);
              cur = (int)jw < 0 ? 0u : cur + 32u;
              j = jw & 0x7fffffffu;
              float nhb = __builtin_fmaf(sc.z, p1q.dz_nod.x, p1q.dz_nod.y);
              nhb = __builtin_fmaf(sc.y, p1q.dx_dy.y, nhb);
              nhb = __builtin_fmaf(sc.x, p1q.dx_dy.x, nhb);
              float w = __builtin_fmaf(sc.w, p1q.na_nA.x, p1q.na_nA.y);
              w = __builtin_fmaf(sc.x, p1q.Px_Py.x, w);
              w = __builtin_fmaf(sc.y, p1q.Px_Py.y, w);
              w = __builtin_fmaf(sc.z, p1q.Pz_0.x, w);
              const float test = __builtin_fmaf(nhb, nhb, w);
              pass = (int)as_u32(test) >= 0 && j != lastj && j != skipj;
              if (EXEC) c_T++;
            }
            // (both arithmetic modes queue here: an exact test inside this loop would run for the one or two lanes that have a
            // candidate in most trips of the wave)
            if (use_queue) {
              if (pass) { queue = (queue << 16) | (unsigned long long)j; qn++; lastj = j; }
              if (__ballot(qn == 4u)) drain();
            } else if (pass) {  // more than 65535 spheres: their indices do not fit the queue's 16-bit entries
              lastj = j;
              test_sphere(j, closest, hit, any_untame);
            }
            // (Measured and left out, profiles/r3_grid_suspend.log: stopping the loop when fewer than 12 / 20 / 28 / 36 / 44 lanes are
            // still walking - they park their walk in LDS and take it up in the next general trace, beside the fresh rays - is
            // bit-identical and 1.5 / 2 / 4 / 8 / 13 % SLOWER: lane-trips are conserved, and the lanes set free wait for the batch.)
            if (!__ballot(cur != 0u || coff != DONE)) break;
          }
          RT_PROF_END(2);
        }
      }
      if (COUNT && blocks) c_T += n_spheres;
      if (__ballot(blocks))
      for (uint32_t base = 0; base < n_spheres; base += 32u) {
        const uint32_t cnt = n_spheres - base < 32u ? n_spheres - base : 32u;
        // ---- phase 1: sign of D + M for spheres base..base+cnt-1 -> bitmask ----
        // bit (cnt_pad-1-j) of `missbits` = sign of sphere (base+j)'s value (1 = certain miss)
        const uint32_t cnt_pad = (cnt + 3u) & ~3u;
        RT_PROF_BEGIN(2);
        const uint32_t missbits = phase1_missbits<CULL, P1S>(sph_p1, base, cnt_pad, p1q);
#if RT_PROFILE_SECTIONS
        asm volatile("" :: "v"(missbits));  // keep the loop inside the timed region
#endif
        RT_PROF_END(2);
        RT_PROF_BEGIN(3);
        // ---- phase 2: each lane walks its own candidates in ascending sphere order ----
        // (untamed rays take every sphere; the lowest cnt_pad - cnt bits are padding spheres)
        const uint32_t valid = (cnt_pad == 32u ? 0xffffffffu : (1u << cnt_pad) - 1u) & ~((1u << (cnt_pad - cnt)) - 1u);  // wave-uniform
        cand = blocks ? (~missbits | untame_fill) & valid : 0u;
        if (LIST) {  // own-sphere drop (see own_sphere_drop): the sphere the ray starts on, when it is in this block
          const uint32_t rel = own - base;
          if (own_drop && rel < cnt) cand &= ~(1u << (cnt_pad - 1u - rel));
        }
        if (QUEUE && use_queue) {
          // every lane meets its candidates in ascending order: queued ones are older than what is still in `cand`,
          // and the queues are empty when a block is walked directly
          auto push = [&]() __attribute__((always_inline)) {
            while (cand != 0u && qn < 4u) {
              const uint32_t lz = (uint32_t)__builtin_clz(cand);
              cand &= ~(0x80000000u >> lz);
              queue = (queue << 16) | (unsigned long long)(base + cnt_pad - 32u + lz);  // newest entry lowest
              qn++;
            }
          };
          push();
          if (__ballot(cand != 0u)) {  // some lane's queue is full: everyone tests what is queued, then the rest of this block
            drain();
            push();
            if (__ballot(cand != 0u)) {  // a block with more than four candidates for a lane (untamed rays: every sphere)
              drain();
              walk(base, cnt_pad, closest, hit, any_untame);
            }
          }
        } else {
          walk(base, cnt_pad, closest, hit, any_untame);
        }
        RT_PROF_END(3);
      }
      if ((QUEUE && use_queue) || !SMALL) {  // (the grid trace queues in fast arithmetic too)
        RT_PROF_BEGIN(3);
        drain();
        RT_PROF_END(3);
      }
      RT_PROF_BEGIN(4);
      if (hit < 0) shade_miss(); else shade_hit(closest, hit);
      RT_PROF_END(4);
    }
  };

  for (;;) {
#if RT_PROFILE_SECTIONS
    prof_iters++;
#endif
#if RT_PROFILE_SECTIONS == 2
    {  // iterations by the number of lanes that hold a path (anything but dead / parked)
      const uint32_t live = (uint32_t)__popcll(__ballot(state >= ST_READY));
      if (live == 0u) prof_live0++; else if (live <= 16u) prof_live1++; else if (live <= 48u) prof_live2++; else prof_live3++;
    }
#endif
    // ================= [E] path end: next sample (cheap, every lane that finished a path) =================
    if (state == ST_PATH_DONE) {
      // send_rays: color += ray_color(ray) (wgsl:120)
      sumx = sumx + colx; sumy = sumy + coly; sumz = sumz + colz;
      sample++;
      state = sample == a.spp ? ST_FRAME_END : ST_NEW_RAY;
    }

    // ================= [F] frame end + [A] pixel hand-out, batched =================
    // Both are rare per lane (once per spp samples) but, with 64 desynchronised lanes, would run at
    // 1/64 occupancy in every other iteration. Lanes park in FRAME_END / NEED_PIXEL until `batch_lanes`
    // of them wait, or nothing else is runnable in the wave.
    const unsigned long long parked = __ballot(state - ST_NEED_PIXEL < 2u);  // NEED_PIXEL or FRAME_END
    // wave-uniform; most iterations have no parked lane and skip all of this with one scalar branch
    if (__builtin_expect(parked != 0ull, 0)) {
    const rt_kargs_cptr ca = cold_args();
    if ((uint32_t)__popcll(parked) >= ca->batch_lanes || !__ballot(state >= ST_READY)) {
    if (state == ST_FRAME_END) {
      // ---- end of this pixel's frame: color / spp (wgsl:122), store_color (wgsl:96-112) ----
      float p0, p1, p2;
      if (ca->inv_spp > 0.0f) {  // spp is a power of two: x / spp == x * (1/spp) exactly
        p0 = sumx * ca->inv_spp; p1 = sumy * ca->inv_spp; p2 = sumz * ca->inv_spp;
      } else {
        p0 = sumx / spp_f; p1 = sumy / spp_f; p2 = sumz / spp_f;
      }
      if (COUNT) c_P++;
      const size_t pix = (size_t)(uvp >> 16) * ca->width + (uvp & 0xffffu);
      bool done;          // nothing left to do for this lane's unit of work
      bool last_frame;    // the RNG state is the one the capture hook wants
      if (ca->staging) {
        // STAGED LAUNCH: the unit of work is (pixel, frame k). The frame's colour goes to the staging buffer;
        // rt_combine_frames_kernel folds the frames into the image in order afterwards.
        // The lane keeps its pixel for the frames of the unit's group (frame_group, a power of two): the next frame of the
        // group starts right below, without a pixel hand-out.
        f32x4 o; o.x = p0; o.y = p1; o.z = p2; o.w = 1.0f;
        ca->staging[((size_t)k * ca->my_tiles + my_tile) * 64u + ((uvp >> 13) & 56u) + (uvp & 7u)] = o;  // (v & 7) * 8 + (u & 7)
        k++;
        last_frame = k == ca->n_frames;
        done = last_frame || (k & (ca->frame_group - 1u)) == 0u;
      } else {
        // The accumulation pixel is read, combined and written back HERE, per frame (it belongs to this lane for the
        // whole launch, so the read sees the lane's own previous write): carrying it in four registers across the
        // frames of a launch cost four registers across the hot loop - and, at 72, spills to scratch memory.
        float acc0, acc1, acc2, acc3;
        if (accum_f32) {
          const f32x4 c = reinterpret_cast<const f32x4*>(ca->accum)[pix];
          acc0 = c.x; acc1 = c.y; acc2 = c.z; acc3 = c.w;
        } else {
          const uint32_t q = reinterpret_cast<const uint32_t*>(ca->accum)[pix];
          acc0 = unorm8_to_f32(q & 255u); acc1 = unorm8_to_f32((q >> 8) & 255u);
          acc2 = unorm8_to_f32((q >> 16) & 255u); acc3 = unorm8_to_f32(q >> 24);
        }
        // cw = f32(frame) / f32(frame + 1) (wgsl:108-109), computed on the host with the same IEEE division
        const float cw = ca->frame_table ? as_f32(ca->frame_table[k].y) : ca->cw0;
        const uint32_t q8 = combine_frame(acc0, acc1, acc2, acc3, p0, p1, p2, cw, combine_sum, accum_f32);
        k++;
        done = last_frame = k == ca->n_frames;
        if (accum_f32) {
          f32x4 o; o.x = acc0; o.y = acc1; o.z = acc2; o.w = acc3;
          reinterpret_cast<f32x4*>(ca->accum)[pix] = o;
        } else {
          reinterpret_cast<uint32_t*>(ca->accum)[pix] = q8;
        }
      }
      if (done) {
        if (ca->rng_out && last_frame) ca->rng_out[pix] = rng;
#if RT_COST_MAX
        // the tile's dearest pixel: most pixels are not, and a plain load tells (the atomics reach the memory side as write requests)
        if (ca->tile_cost && __builtin_nontemporal_load(&ca->tile_cost[my_tile]) < iters) atomicMax(&ca->tile_cost[my_tile], iters);
#else
        if (ca->tile_cost) atomicAdd(&ca->tile_cost[my_tile], iters);
#endif
        state = ST_NEED_PIXEL;  // ask for the next pixel
      } else {
        rng = seed_xy ^ ca->frame_table[k].x;
        rng = rng_step(rng);
        if (COUNT) c_rand++;
        sample = 0; sumx = sumy = sumz = 0.0f;
        state = ST_NEW_RAY;
      }
    }

    // ================= [A] hand out pixels =================
    const unsigned long long need = __ballot(state == ST_NEED_PIXEL);
    if (need) {
      const uint32_t n_need = (uint32_t)__popcll(need);
      // rank of this lane among the needing ones; recomputed per use (two v_mbcnt) rather than kept across the fetch
      auto my_rank = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t m = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        asm volatile("" : "+v"(m));
        return m;
      };
      const uint32_t avail = q_end - q_cur;
      const bool wants = state == ST_NEED_PIXEL;
      // the lane takes pixel record `w` of the tile staged in LDS (unit: tile slot `t_slot`, frame `t_frame`)
      auto take_pixel = [&](uint32_t w, uint32_t t_slot, uint32_t t_frame, uint32_t t_mask) __attribute__((always_inline)) {
        const f32x4 rec0 = tile_lds[w * 2u], rec1 = tile_lds[w * 2u + 1u];
        if (as_u32(rec0.w) != 0xffffffffu) {  // else: outside the covered area, the lane takes another slot next round
          my_tile = t_slot; k = t_frame; tmask = t_mask;
          pcx = rec0.x; pcy = rec0.y; pcz = rec0.z;
          pdx = rec1.x; pdy = rec1.y; pdz = rec1.z;
          uvp = as_u32(rec0.w);
          seed_xy = as_u32(rec1.w);
          // k = first frame of this unit: 0, or the unit's frame in a staged launch
          rng = seed_xy ^ (ca->frame_table ? ca->frame_table[k].x : ca->seed_t0);
          rng = rng_step(rng);  // initialize_rng's discarded rand() (wgsl:262)
          if (COUNT) c_rand++;
          sample = 0; sumx = sumy = sumz = 0.0f;
          iters = 0;
          state = ST_NEW_RAY;
        }
      };
      // pixels of the tile already staged in LDS first: their records are consumed before a new tile overwrites them
      // (and before the fetch below, so that nothing of them stays live across it)
      {
        const uint32_t my = my_rank();
        if (wants && my < avail) take_pixel((q_cur + my) & 63u, q_tile, q_frame, q_mask);
      }
      if (n_need > avail) {
        uint32_t base = 0xffffffffu, q_frame_new = 0;
        // Fetch units until one has pixels to hand out. unit -> (tile slot, frame): a staged launch queues every frame
        // of a tile as a unit of its own. Tiles come in the order of the previous launch's measured cost, dearest
        // first, so the launch ends on cheap pixels (results do not depend on the order).
        for (;;) {
          base = 0xffffffffu;
          if (more) {
            // one atomic per `queue_chunk` units: a launch of many short units (an 8K pass at 1 spp: 518 400 tiles of one sample per lane)
            // is otherwise bound by the rate of atomics on this one address (profiles/r3_queue_chunk.log)
            if (ca->queue_chunk == 0u) {
              // NO QUEUE: wave w takes units w, w + waves, w + 2 waves, ... of the (cost-ordered) list. For short units the queue head's
              // ~86 M atomics per second are the launch's bound, and what dynamic hand-out could balance is small (rt_api.cpp, launch())
              q_chunk_end = q_next < ca->n_units ? q_next + 1u : q_next;
            } else if (q_next == q_chunk_end) {
              uint32_t b = 0;
              if (lane_id() == 0u) b = atomicAdd(ca->queue, ca->queue_chunk);  // all 64 lanes are active here (uniform loop)
              b = __builtin_amdgcn_readfirstlane(b);
              q_next = b;
              q_chunk_end = b < ca->n_units ? (b + ca->queue_chunk < ca->n_units ? b + ca->queue_chunk : ca->n_units) : b;
            }
            if (q_next < q_chunk_end) { base = q_next; q_next += ca->queue_chunk == 0u ? gridDim.x : 1u; }
            else { more = false; base = 0xffffffffu; }
          }
          if (base == 0xffffffffu) break;
          uint32_t slot = base;
          q_frame_new = 0;
          if (ca->staging) { slot = base / ca->n_groups; q_frame_new = (base - slot * ca->n_groups) * ca->frame_group; }
          q_tile_new = ca->tile_order ? ca->tile_order[slot] : slot;
          // tile_masks[tile]: 0 = sky tile, else the spheres a primary ray of the tile can hit (SPLIT; all ones otherwise)
          q_mask_new = 0xffffffffu;
          if ((COUNT && !SPLIT) || !ca->tile_masks || (q_mask_new = __builtin_amdgcn_readfirstlane(ca->tile_masks[q_tile_new])) != 0u) break;
          // ---- SKY TILE, finished right here by all 64 lanes (lane = pixel, no divergence; the lanes' own paths
          // wait): rt_tile_masks_kernel proved that every primary ray of this tile misses every sphere in the literal
          // arithmetic ("sky tiles"). Each sample is then get_random_ray's two draws and the miss
          // colour (wgsl:174-180, 144-148, 131-134) - no trace.
          const uint32_t tile = q_tile_new * ca->world + ca->rank;
          const uint32_t tx = tile % ca->tiles_x, ty = tile / ca->tiles_x;
          const uint32_t lane = lane_id();
          const uint32_t pu = tx * 8u + (lane & 7u), pv = ty * 8u + (lane >> 3);
          if (pu < ca->cov_w && pv < ca->cov_h) {
            const float vx = (float)pu, vy = (float)(ca->height - pv);
            const float wx = (float)(pu + 1u), wy = (float)(ca->height - (pv + 1u));
            const __attribute__((address_space(4))) float* m = ca->p2w;  // column-major
            const float spcy = ((m[1] * vx + m[5] * vy) + m[9] * 0.0f) + m[13] * 1.0f;           // pixel_center.y
            const float spdy = (((m[1] * wx + m[5] * wy) + m[9] * 0.0f) + m[13] * 1.0f) - spcy;  // pixel_delta.y
            const size_t spix = (size_t)pv * ca->width + pu;
            const uint32_t sseed = rng_seed_pixel(pu, pv);
            float b0 = 0, b1 = 0, b2 = 0, b3 = 0;
            if (ca->staging) {
              // the image is touched by rt_combine_frames_kernel only
            } else if (accum_f32) {
              const f32x4 c = reinterpret_cast<const f32x4*>(ca->accum)[spix];
              b0 = c.x; b1 = c.y; b2 = c.z; b3 = c.w;
            } else {
              const uint32_t q = reinterpret_cast<const uint32_t*>(ca->accum)[spix];
              b0 = unorm8_to_f32(q & 255u); b1 = unorm8_to_f32((q >> 8) & 255u);
              b2 = unorm8_to_f32((q >> 16) & 255u); b3 = unorm8_to_f32(q >> 24);
            }
            const uint32_t kf0 = ca->staging ? q_frame_new : 0u;
            const uint32_t kf1 = ca->staging && q_frame_new + ca->frame_group < ca->n_frames ? q_frame_new + ca->frame_group : ca->n_frames;
            uint32_t srng = 0, q8 = 0;
            for (uint32_t kf = kf0; kf < kf1; ++kf) {
              srng = rng_step(sseed ^ (ca->frame_table ? ca->frame_table[kf].x : ca->seed_t0));  // initialize_rng (wgsl:254-263)
              if (EXEC) { c_P++; c_N += ca->spp; c_M += ca->spp; c_sky += ca->spp; c_rand += 1u + 2u * ca->spp; }
              float ssx = 0.0f, ssy = 0.0f, ssz = 0.0f;
              for (uint32_t sm = 0; sm < ca->spp; ++sm) {
                srng = rng_step(srng);  // px: does not reach the miss colour
                srng = rng_step(srng);
                const float sdy = spcy + rng_jitter(srng) * spdy;
                float mx = 0.0f, my = 0.0f, mz = 0.0f;  // ray_color: color = 0, reflectance = 1
                if (!checker) {
                  const float t = 0.5f * (1.0f - sdy);
                  const float it = 1.0f - t;
                  mx = mx + (it + t * 0.5f) * 1.0f; my = my + (it + t * 0.7f) * 1.0f; mz = mz + (it + t * 1.0f) * 1.0f;
                } else {
                  mx = mx + 0.0f * 1.0f; my = my + 0.0f * 1.0f; mz = mz + 0.0f * 1.0f;
                }
                ssx = ssx + mx; ssy = ssy + my; ssz = ssz + mz;  // send_rays (wgsl:120)
              }
              float p0, p1, p2;
              if (ca->inv_spp > 0.0f) {
                p0 = ssx * ca->inv_spp; p1 = ssy * ca->inv_spp; p2 = ssz * ca->inv_spp;
              } else {
                p0 = ssx / spp_f; p1 = ssy / spp_f; p2 = ssz / spp_f;
              }
              if (ca->staging) {
                f32x4 o; o.x = p0; o.y = p1; o.z = p2; o.w = 1.0f;
                ca->staging[((size_t)kf * ca->my_tiles + q_tile_new) * 64u + lane] = o;  // lane = (v & 7) * 8 + (u & 7)
              } else {
                const float cw = ca->frame_table ? as_f32(ca->frame_table[kf].y) : ca->cw0;
                q8 = combine_frame(b0, b1, b2, b3, p0, p1, p2, cw, combine_sum, accum_f32);
              }
            }
            if (!ca->staging) {
              if (accum_f32) {
                f32x4 o; o.x = b0; o.y = b1; o.z = b2; o.w = b3;
                reinterpret_cast<f32x4*>(ca->accum)[spix] = o;
              } else {
                reinterpret_cast<uint32_t*>(ca->accum)[spix] = q8;
              }
            }
            if (ca->rng_out && kf1 == ca->n_frames) ca->rng_out[spix] = srng;
          }
        }
        if (base != 0xffffffffu) {
          // ---- stage tile `base`: all 64 lanes compute the per-pixel constants of its 64 pixels at full
          // occupancy (pixel_to_world, get_pixel_delta wgsl:85-94; the pixel part of the RNG seed
          // wgsl:258-259) into LDS; lanes pick their record up when they are handed the pixel.
          const uint32_t tile = q_tile_new * ca->world + ca->rank;
          const uint32_t tx = tile % ca->tiles_x, ty = tile / ca->tiles_x;
          const uint32_t lane = lane_id();
          const uint32_t pu = tx * 8u + (lane & 7u), pv = ty * 8u + (lane >> 3);
          const float vx = (float)pu, vy = (float)(ca->height - pv);
          const float wx = (float)(pu + 1u), wy = (float)(ca->height - (pv + 1u));
          const __attribute__((address_space(4))) float* m = ca->p2w;  // column-major
          const float cx = ((m[0] * vx + m[4] * vy) + m[8] * 0.0f) + m[12] * 1.0f;
          const float cy = ((m[1] * vx + m[5] * vy) + m[9] * 0.0f) + m[13] * 1.0f;
          const float cz = ((m[2] * vx + m[6] * vy) + m[10] * 0.0f) + m[14] * 1.0f;
          const float nx = ((m[0] * wx + m[4] * wy) + m[8] * 0.0f) + m[12] * 1.0f;
          const float ny = ((m[1] * wx + m[5] * wy) + m[9] * 0.0f) + m[13] * 1.0f;
          const float nz = ((m[2] * wx + m[6] * wy) + m[10] * 0.0f) + m[14] * 1.0f;
          const bool inside = pu < ca->cov_w && pv < ca->cov_h;  // false only on ragged edges in RT_DISPATCH_FULL
          f32x4 w0, w1;
          w0.x = cx; w0.y = cy; w0.z = cz; w0.w = as_f32(inside ? (pv << 16 | pu) : 0xffffffffu);
          w1.x = nx - cx; w1.y = ny - cy; w1.z = nz - cz; w1.w = as_f32(rng_seed_pixel(pu, pv));
          __builtin_amdgcn_wave_barrier();  // the reads of the old tile above stay before these writes
          tile_lds[lane * 2u] = w0; tile_lds[lane * 2u + 1u] = w1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const uint32_t my = my_rank();
          if (wants && my >= avail) take_pixel(my - avail, q_tile_new, q_frame_new, q_mask_new);
          q_tile = q_tile_new; q_frame = q_frame_new; q_mask = q_mask_new;
          q_cur = base * 64u + (n_need - avail);
          q_end = base * 64u + 64u;
        } else {
          if (wants && my_rank() >= avail) state = ST_DEAD;  // the queue is empty
          q_cur = q_end;
        }
      } else {
        q_cur += n_need;
      }
    }
    if (!__ballot(state != ST_DEAD)) break;  // every lane drained and the queue is empty (lanes die in [A] only)
    }  // service
    }  // parked

    // ================= [G] random draws, [C] trace + shade =================
    if (!SPLIT) {
      // CLASSIC SHAPE, one trace per lane and iteration. Round 0 of the draws is shared: lanes starting a sample draw
      // the two jitter numbers of get_random_ray (wgsl:174-180) with the same instructions that give lanes in
      // random_in_unit_sphere (wgsl:281-290) the first two of their three numbers; further rounds serve NEED_DIR only.
      RT_PROF_BEGIN(0);
      if (state == ST_NEW_RAY || state == ST_NEED_DIR) {
        uint32_t s0, s1;
        RT_STEP(s0); RT_STEP(s1);
        if (state == ST_NEW_RAY) {
          start_ray(s0, s1);
          state = ST_READY;
        } else {
          uint32_t s2;
          RT_STEP(s2);
          try_direction(s0, s1, s2);
        }
      }
      if (COUNT) c_wr++;
#ifndef RT_ROUND_MIN_LANES
#define RT_ROUND_MIN_LANES 1u  // a further round runs while at least this many lanes still need a direction
#endif
      for (uint32_t round = 1; round < a.reject_tries && (uint32_t)__popcll(__ballot(state == ST_NEED_DIR)) >= RT_ROUND_MIN_LANES; ++round) {
        if (COUNT) c_wr++;
        if (state == ST_NEED_DIR) {
          uint32_t s0, s1, s2;
          RT_STEP(s0); RT_STEP(s1); RT_STEP(s2);
          try_direction(s0, s1, s2);
        }
      }
      if (state == ST_GOT_DIR) finish_direction();
      RT_PROF_END(0);
      trace_general();
    } else if (MASK) {
      // SPLIT SHAPE (scenes of <= 32 spheres whose launch has per-tile candidate masks; DESIGN.md, "primary / bounce
      // split"). Every section appears once, and a lane passes through up to two rays per iteration:
      //   [G0] NEW_RAY: jitter draws, prelude; candidates := the tile's mask (no phase 1: every sphere outside the mask
      //        is a certain miss of every primary ray of the tile in the literal arithmetic, and hit_sphere has no
      //        side effect when it returns false)                                              -> ST_TEST
      //   [W]  ST_TEST: exact tests of the candidates - of a fresh primary ray, or of the bounce ray whose phase 1 ran
      //        at the end of the previous iteration - then miss or scatter                  -> PATH_DONE | READY | NEED_DIR
      //   [G1] NEED_DIR: rejection rounds, normalize                                           -> READY
      //   [P1] READY: prelude + phase 1 over all spheres; no candidate = miss, at once         -> PATH_DONE | ST_TEST
      // 93 % of the primary rays of the 20-sphere scene's traced tiles hit and 71 % of its bounce rays have no
      // candidate, so the common path "primary hit, bounce miss" is ONE iteration (the classic shape needs two), and
      // the rounds, the walk and the scatter are each paid once per iteration for primary and bounce lanes together.
      RT_PROF_BEGIN(0);
      if (state == ST_NEW_RAY) {
        RT_PROF_LANES(0);
        uint32_t s0, s1;
        RT_STEP(s0); RT_STEP(s1);
        start_ray(s0, s1);
        p1_ray unused;
        iters++;
        if (EXEC) c_R++;
        ray_prelude(std::integral_constant<bool, false>{}, unused);
        cand = tmask;
        w_closest = RT_RAY_TMAX; w_hit = 0xffffffffu;
        state = ST_TEST;
      }
      RT_PROF_END(0);
      RT_PROF_BEGIN(3);
      if (state == ST_TEST) {
        // The walk's `closest` and `hit` live in w_closest / w_hit across iterations (two registers that are free while
        // a lane is in ST_TEST, see their declaration): after the first trip, a further trip runs only while at least
        // RT_WALK_MIN_LANES lanes still hold candidates - a trip costs the wave ~85 instructions whoever takes part, and
        // the few rays with a second or third candidate (a third of the bounce rays, hardly any primary ray since the
        // occlusion drop of the tile masks) would otherwise make every iteration pay two or three of them. Such a lane
        // keeps ST_TEST and takes its next candidate in the next iteration's first trip, which runs anyway: one loop
        // iteration more for that path, in the same ascending sphere order (so with the same result).
        float closest = w_closest;
        int hit = (int)w_hit;
        if (RT_RCP_AT_WALK) ray_literal_a();
#if RT_PROFILE_SECTIONS == 3
        if (hit < 0 && closest == RT_RAY_TMAX) {
          const uint32_t pc = (uint32_t)__popc(cand);
          if (pc) atomicAdd(&prof_lds[depth == 0u ? (pc > 5u ? 5u : pc) - 1u : 8u + (pc > 4u ? 4u : pc) - 1u], 1ull);
        }
#endif
        const unsigned long long any_untame = __ballot(!tame);
        for (;;) {
          if (cand) {
            RT_PROF_LANES(5);
            const uint32_t lz = (uint32_t)__builtin_clz(cand);
            cand &= ~(0x80000000u >> lz);
            test_sphere(n_pad32 - 32u + lz, closest, hit, any_untame);
          }
          const uint32_t left = (uint32_t)__popcll(__ballot(cand != 0u));
          if (left < RT_WALK_MIN_LANES) break;  // (also: nothing left)
        }
        RT_PROF_END(3);
        RT_PROF_BEGIN(4);
        if (cand != 0u) { w_closest = closest; w_hit = (uint32_t)hit; }
        else if (hit < 0) shade_miss();
        else { RT_PROF_LANES(3); shade_hit(closest, hit); }
        RT_PROF_END(4);
        RT_PROF_BEGIN(3);
      }
      RT_PROF_END(3);
      RT_PROF_BEGIN(0);
      for (uint32_t round = 0; round < a.reject_tries && __ballot(state == ST_NEED_DIR) != 0ull; ++round) {
        if (state == ST_NEED_DIR) {
          RT_PROF_LANES(2);
          uint32_t s0, s1, s2;
          RT_STEP(s0); RT_STEP(s1); RT_STEP(s2);
          try_direction(s0, s1, s2);
        }
      }
      if (state == ST_GOT_DIR) { RT_PROF_LANES(4); finish_direction(); }
      RT_PROF_END(0);
      if (state == ST_READY) {
        RT_PROF_LANES(1);
        RT_PROF_BEGIN(1);
        iters++;
        if (EXEC) { c_R++; c_T += n_spheres; }
        p1_ray p1q;
        probe_bpermute();
        const float a_fast = ray_prelude(std::integral_constant<bool, true>{}, p1q);
        RT_PROF_END(1);
        RT_PROF_BEGIN(2);
#if RT_P1_VGPR
        uint32_t missbits = 0xffffffffu;
        for (uint32_t i = 0; i < n_pad32; i += 2u) {
          const f32x4 s0 = p1_lds[i], s1 = p1_lds[i + 1u];
          missbits = phase1_one_v<CULL>(missbits, s0, p1q);
          missbits = phase1_one_v<CULL>(missbits, s1, p1q);
        }
#else
        const uint32_t missbits = phase1_missbits<CULL, P1S>(sph_p1, 0u, n_pad32, p1q);
#endif
        cand = (~missbits | (tame ? 0u : 0xffffffffu)) & valid32 & ~(own_sphere_drop(RT_RCP_AT_WALK ? a_fast : a_lit) ? 1u << (n_pad32 - 1u - own) : 0u);
        w_closest = RT_RAY_TMAX; w_hit = 0xffffffffu;  // (own is dead from here on: w_hit shares its register)
        state = ST_TEST;
        if (cand == 0u) shade_miss();
        RT_PROF_END(2);
      }
    } else {
      // LIST SHAPE (scenes of more than 64 spheres with per-tile candidate LISTS, rt_tile_masks_kernel): the primary ray
      // walks its tile's list - the up to 15 spheres a primary ray of the tile can hit - instead of running phase 1 over
      // the whole scene; the general trace [C] then serves the bounce rays (and the primary rays of the few tiles whose
      // list overflowed), with the own-sphere drop. A lane passes its primary and its first bounce ray through one
      // iteration, and phase 1 - two thirds of the time of the 1024-sphere scene - runs for bounce rays only.
      RT_PROF_BEGIN(0);
      uint32_t lcnt = 0;
      if (state == ST_NEW_RAY) {
        RT_PROF_LANES(0);
        uint32_t s0, s1;
        RT_STEP(s0); RT_STEP(s1);
        start_ray(s0, s1);
        own = 0xffffffffu;
        lcnt = a.tile_lists[(size_t)my_tile * 16u];
        state = lcnt > 15u ? ST_READY : ST_TEST;  // 0xffff: the list overflowed (or the tile's rays are not provably tame)
      }
      RT_PROF_END(0);
      RT_PROF_BEGIN(3);
      if (state == ST_TEST) {
        p1_ray unused;
        iters++;
        if (EXEC) c_R++;
        ray_prelude(std::integral_constant<bool, false>{}, unused);
        float closest = RT_RAY_TMAX;
        int hit = -1;
        const unsigned long long any_untame = __ballot(!tame);
        const uint16_t* list = a.tile_lists + (size_t)my_tile * 16u + 1u;
        for (uint32_t j = 0; j < lcnt; ++j) { RT_PROF_LANES(3); test_sphere(list[j], closest, hit, any_untame); }  // ascending sphere index
        if (hit < 0) shade_miss(); else shade_hit(closest, hit);
      }
      RT_PROF_END(3);
      RT_PROF_BEGIN(0);
      for (uint32_t round = 0; round < a.reject_tries && __ballot(state == ST_NEED_DIR) != 0ull; ++round) {
        if (state == ST_NEED_DIR) {
          uint32_t s0, s1, s2;
          RT_STEP(s0); RT_STEP(s1); RT_STEP(s2);
          try_direction(s0, s1, s2);
        }
      }
      if (state == ST_GOT_DIR) finish_direction();
      RT_PROF_END(0);
      // The general trace costs the wave ~6 instructions per sphere of the scene whoever takes part, the sections above
      // a few hundred in all: lanes whose primary ray missed (or whose direction is not drawn yet) would sit it out.
      // So the trace waits - at most RT_LIST_MAXSKIP iterations in a row - until RT_LIST_BATCH lanes are READY,
      // while the other lanes go round the cheap sections again and join them.
      {
        const uint32_t n_ready = (uint32_t)__popcll(__ballot(state == ST_READY));
        const bool others = __ballot(state == ST_NEW_RAY || state == ST_PATH_DONE || state == ST_NEED_DIR) != 0ull;
        if (n_ready >= RT_LIST_BATCH || !others || list_skips >= RT_LIST_MAXSKIP) { list_skips = 0; trace_general(); }
        else if (n_ready) list_skips++;
      }
    }
  }

  const rt_kargs_cptr ca = cold_args();
#if RT_PROFILE_SECTIONS
  if (lane_id() == 0u) {
    const unsigned long long life = __builtin_amdgcn_s_memtime() - prof_start;
    const unsigned long long rlife = __builtin_amdgcn_s_memrealtime() - prof_rstart;  // 100 MHz
#if RT_PROFILE_SECTIONS >= 2
    // (lanes in the low 38 bits, executions / 16 above them: the sum over all waves of a 20-pass launch stays inside its 26 bits)
    auto repack = [](unsigned long long v) { return (v & ((1ull << 38) - 1ull)) | ((v >> 42) << 38); };
    const unsigned long long pv[8] = {repack(prof_lds[0]), repack(prof_lds[1]), repack(prof_lds[2]), repack(prof_lds[3]), repack(prof_lds[4]), life, prof_iters, rlife};
    prof_p2_lanes = prof_lds[5] & ((1ull << 38) - 1ull); prof_p2_waves = prof_lds[5] >> 38;
#else
    const unsigned long long pv[8] = {prof_acc0, prof_acc1, prof_acc2, prof_acc3, prof_acc4, life, prof_iters, rlife};
#endif
    for (int i = 0; i < 8; ++i) atomicAdd(&ca->counters[i], pv[i]);
    // launch shape: [8] = ~min start, [9] = max end, [10] = sum of starts, [11] = sum of ends (100 MHz ticks),
    // [12], [13] = lane- and wave-iterations of the phase-2 candidate loop
#if RT_PROFILE_SECTIONS == 3
    for (int i = 0; i < 4; ++i) atomicAdd(&ca->counters[8 + i], prof_lds[8 + i]);
#elif RT_PROFILE_SECTIONS == 2
    atomicAdd(&ca->counters[8], prof_live0);   // iterations with no / 1-16 / 17-48 / 49-64 lanes holding a path
    atomicAdd(&ca->counters[9], prof_live1);
    atomicAdd(&ca->counters[10], prof_live2);
    atomicAdd(&ca->counters[11], prof_live3);
#else
    atomicMax(&ca->counters[8], ~prof_rstart);
    atomicMax(&ca->counters[9], prof_rstart + rlife);
    atomicAdd(&ca->counters[10], prof_rstart & 0xffffffffull);
    atomicAdd(&ca->counters[11], (prof_rstart + rlife) & 0xffffffffull);
#endif
    atomicAdd(&ca->counters[12], prof_p2_lanes);  // candidate walk: lane-iterations
    atomicAdd(&ca->counters[13], prof_p2_waves);  // candidate walk: wave-iterations
  }
#endif
  if (COUNT) {
    const uint32_t vals[12] = {c_T, c_D, c_H, c_R, c_M, c_S, c_Sm, c_J, c_Sd, c_N, c_P, c_rand};
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(&ca->counters[i], (unsigned long long)vals[i]);
    if (EXEC) {  // slots 12 / 13 carry the exact tests run and the samples finished in sky tiles (rt_hip.h, rt_enable_counters mode 2)
      atomicAdd(&ca->counters[12], (unsigned long long)c_X);
      atomicAdd(&ca->counters[13], (unsigned long long)c_sky);
    } else if (lane_id() == 0u) {
      atomicAdd(&ca->counters[12], 64ull * c_wt);
      atomicAdd(&ca->counters[13], 64ull * c_wr);
    }
  }
}

// Second half of a STAGED launch: one thread per covered pixel of this rank's tiles folds the n_frames staged
// colours into the accumulation pixel with combine_frame, in frame order - the recurrence the path kernel
// runs per pixel in a pixel-sequential launch, operation for operation.
__global__ __launch_bounds__(256) void rt_combine_frames_kernel(const rt_kargs a) {
  const uint32_t pu = blockIdx.x * 64u + (threadIdx.x & 63u), pv = blockIdx.y * 4u + (threadIdx.x >> 6);
  if (pu >= a.cov_w || pv >= a.cov_h) return;
  const uint32_t tile = (pv >> 3) * a.tiles_x + (pu >> 3);
  if (tile % a.world != a.rank) return;
  const bool accum_f32 = (a.flags & RT_ACCUM_F32) != 0, combine_sum = (a.flags & RT_COMBINE_SUM) != 0;
  const size_t pix = (size_t)pv * a.width + pu;
  // staging is indexed by this rank's own tiles: ((frame * my_tiles + tile slot) * 64 + pixel within the tile), slot = tile / world
  const size_t sidx = (size_t)(tile / a.world) * 64u + (pv & 7u) * 8u + (pu & 7u), sstride = (size_t)a.my_tiles * 64u;
  float acc0, acc1, acc2, acc3;
  if (accum_f32) {
    const f32x4 c = reinterpret_cast<const f32x4*>(a.accum)[pix];
    acc0 = c.x; acc1 = c.y; acc2 = c.z; acc3 = c.w;
  } else {
    const uint32_t q = reinterpret_cast<const uint32_t*>(a.accum)[pix];
    acc0 = unorm8_to_f32(q & 255u); acc1 = unorm8_to_f32((q >> 8) & 255u);
    acc2 = unorm8_to_f32((q >> 16) & 255u); acc3 = unorm8_to_f32(q >> 24);
  }
  uint32_t q8 = 0;
  for (uint32_t k = 0; k < a.n_frames; ++k) {
    const f32x4 p = a.staging[(size_t)k * sstride + sidx];
    q8 = combine_frame(acc0, acc1, acc2, acc3, p.x, p.y, p.z, as_f32(a.frame_table[k].y), combine_sum, accum_f32);
  }
  if (accum_f32) {
    f32x4 o; o.x = acc0; o.y = acc1; o.z = acc2; o.w = acc3;
    reinterpret_cast<f32x4*>(a.accum)[pix] = o;
  } else {
    reinterpret_cast<uint32_t*>(a.accum)[pix] = q8;
  }
}

// TILE MASKS ("sky tiles" and "primary / bounce split"): masks[j] says which spheres a primary ray of this
// rank's j-th 8x8 tile can hit. Sphere i is LEFT OUT when EVERY primary ray the pixels of the tile can draw is a certain
// miss of it in the literal arithmetic of hit_sphere (wgsl:200-231). A sufficient condition per sphere, evaluated in
// double precision with margins far above the f32 rounding of the literal code; a sphere that fails it stays a candidate
// and gets the exact test. masks[j] == 0: a SKY TILE, which the path kernel finishes without tracing. Scenes of at
// most 32 spheres get the mask in the bit layout of the path kernel's candidate walk (sphere i = bit cnt_pad-1-i,
// cnt_pad = n rounded up to 4), larger scenes only 0 / all ones.
//   directions of the tile: d = c0 u + c1 (H - v) + c3 + (px dx, py dy, dz), u, v over the 8x8 pixels, px, py in [-.5, .5)
//   (wgsl:85-94,174-180; c0, c1, c3 columns of pixel_to_world, delta = c0 - c1): a ball of radius rho around dc.
//   Per sphere (centre c, radius r, seen from the origin o at distance L under the angle psi against dc), with
//   S = |o|^2 + |c|^2 + r^2 and theta = asin(rho / |dc|):
//   (i)  L^2 - r^2 > 2^-12 S: the origin is outside, by 4000 times the rounding of the literal `c`; and
//   (A)  L^2 sin^2(min(psi, pi - psi) - theta) > r^2 + 2^-12 S: no line of the cone comes near the sphere, the exact
//        discriminant is below -2^-12 a S and the literal one (within 54 2^-24 a S of it) is negative; or
//   (B)  cos(psi - theta) < -2^-8: the sphere lies behind every ray, literal half_b > 0 and c > 0, both roots <= 0 <= tmin.
//   Rays must be `tame` (no overflow / underflow in the literal code): |d|^2 in [2^-19, 2^19], |o|^2 <= 2^39, tame scene.
// WAVE = false: one thread per tile (scenes of at most 32 spheres: a mask). WAVE = true: one WAVE per tile, lane l classifies spheres
// l, l + 64, ... - for 1024 spheres the thread-per-tile form is 2 x 1024 classifications in a row per thread with 500 waves on the
// whole chip (0.78 ms per camera change at 1080p, more than a pass of that scene takes); same classification, same lists.
template <bool WAVE>
__global__ __launch_bounds__(256) void rt_tile_masks_kernel(const rt_kargs a, uint32_t* __restrict__ masks, uint16_t* __restrict__ lists) {
  const uint32_t j = WAVE ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = WAVE ? threadIdx.x & 63u : 0u, stride = WAVE ? 64u : 1u;
  if (j >= a.my_tiles) return;  // (WAVE: the whole wave)
  const uint32_t tile = j * a.world + a.rank;
  const uint32_t tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const double c0[3] = {a.p2w[0], a.p2w[1], a.p2w[2]}, c1[3] = {a.p2w[4], a.p2w[5], a.p2w[6]}, c3[3] = {a.p2w[12], a.p2w[13], a.p2w[14]};
  const double o[3] = {a.pos[0], a.pos[1], a.pos[2]};
  const double H = (double)a.height;
  const double um = 8.0 * tx + 3.5, vm = H - (8.0 * ty + 3.5);
  double dc[3], n0 = 0, n1 = 0, n3 = 0, nd = 0, oo = 0;
  for (int i = 0; i < 3; ++i) {
    dc[i] = c0[i] * um + c1[i] * vm + c3[i];
    n0 += c0[i] * c0[i]; n1 += c1[i] * c1[i]; n3 += c3[i] * c3[i];
    nd += (c0[i] - c1[i]) * (c0[i] - c1[i]);
    oo += o[i] * o[i];
  }
  dc[2] += c0[2] - c1[2];  // + 1.0 * pixel_delta.z
  n0 = sqrt(n0); n1 = sqrt(n1); n3 = sqrt(n3); nd = sqrt(nd);
  const double mag = n0 * (8.0 * tx + 9.0) + n1 * (H + 1.0) + n3 + nd;  // magnitude of the f32 terms that form d
  const double rho = (3.5 * (n0 + n1) + 0.5 * (fabs(c0[0] - c1[0]) + fabs(c0[1] - c1[1]))) * (1.0 + 0x1p-20) + 0x1p-18 * mag;
  const double ndc = sqrt(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
  bool ok = a.p1_tame != 0u && rho < 0.25 * ndc && oo <= 0x1p39 && (ndc - rho) * (ndc - rho) >= 0x1p-19 && (ndc + rho) * (ndc + rho) <= 0x1p19;
  // The third column is multiplied by 0.0 (wgsl:86) and does not enter the bound, but a non-finite entry there makes
  // every direction NaN - and a NaN root passes hit_sphere's range tests, i.e. such rays HIT: never a sky tile.
  float colsum = 0.0f;
  for (int i = 0; i < 16; ++i) colsum += a.p2w[i] * 0.0f;
  ok = ok && colsum == 0.0f;
  const bool per_sphere = a.n_spheres <= 32u;
  const uint32_t cnt_pad = per_sphere ? (a.n_spheres + 3u) & ~3u : 32u;
  const uint32_t everything = per_sphere ? (cnt_pad == 32u ? 0xffffffffu : (1u << cnt_pad) - 1u) & ~((1u << (cnt_pad - a.n_spheres)) - 1u) : 0xffffffffu;
  uint32_t mask = everything;  // the rays are not provably tame: every sphere stays a candidate
  uint32_t n_list = 0xffffu;   // > 32 spheres: length of the tile's candidate list, 0xffff = no list (overflow / not tame)
  uint16_t* list = lists && !per_sphere ? lists + (size_t)j * 16u : nullptr;
  if (ok) {  // also false for NaN / inf in the columns used above
    mask = 0u;
    n_list = 0u;
    const double sin_t = rho / ndc, cos_t = sqrt(1.0 - sin_t * sin_t);
    // Per sphere: certain miss?; `low` = lower bound of every root it can be accepted with and `up` = upper bound of
    // the root of a CERTAIN hit, both as distances along the ray (OCCLUSION, below; up = +inf when the hit is not certain)
    auto classify = [&](uint32_t i, double& low, double& up) -> bool {
      const rt_f32x4 sp = a.sph_shade[i];
      const double c[3] = {sp.x, sp.y, sp.z}, r = fabs((double)sp.w), r2 = r * r;  // (hit_sphere sees r only as r r)
      const double v[3] = {c[0] - o[0], c[1] - o[1], c[2] - o[2]};
      const double L2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      const double m = 0x1p-12 * (oo + c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + r2);
      low = -1.0e300; up = 1.0e300;
      if (!(L2 - r2 > m)) return false;
      const double L = sqrt(L2);
      const double cpsi = fmax(-1.0, fmin(1.0, (v[0] * dc[0] + v[1] * dc[1] + v[2] * dc[2]) / (L * ndc)));
      const double ac = fabs(cpsi), sp1 = sqrt(1.0 - ac * ac);
      const double sdiff = sp1 * cos_t - ac * sin_t;  // sin(min(psi, pi - psi) - theta)
      const bool clear = sdiff > 0.0 && L2 * sdiff * sdiff > (r2 + m) * (1.0 + 0x1p-20);
      const bool behind = cpsi * cos_t + sp1 * sin_t < -0x1p-8;  // cos(psi - theta)
      if (clear || behind) return true;
      // the rays of the tile meet the line to the centre under angles in [max(0, psi - theta), psi + theta]
      const double smin = cpsi > 0.0 && sdiff <= 0.0 ? 0.0 : sp1 * cos_t - cpsi * sin_t;
      const double cmin = cpsi > 0.0 && sdiff <= 0.0 ? 1.0 : cpsi * cos_t + sp1 * sin_t;
      if (cmin > 0.0) low = (L * cmin - sqrt(fmax(0.0, r2 - L2 * smin * smin)) - 0x1p-7 * sqrt(L2 + r2)) * (1.0 - 0x1p-20);
      const double smax = sp1 * cos_t + cpsi * sin_t, cmax = cpsi * cos_t - sp1 * sin_t;
      const double inner = r2 - L2 * smax * smax;  // r^2 - (distance of the cone's worst line from the centre)^2
      const double ea = 0x1p-6 * r + 0x1p-19 * L;
      if (cpsi > 0.0 && cmax > 0.0 && inner >= 0x1p-12 * (L2 + r2) && (L - r) - ea > RT_RAY_TMIN * (ndc + rho) * (1.0 + 0x1p-10))
#ifdef RT_BREAK_OCCLUSION  // harness check of tools/fuzz_proofs.py: an unsound bound must show up as mismatches
        up = (L * cmax - sqrt(inner)) * 0.8;
#else
        up = (L * cmax - sqrt(inner) + ea) * (1.0 + 0x1p-20);
#endif
      return false;
    };
    // OCCLUSION. hit_sphere's loop ends with the smallest accepted root and, among equal ones, the lowest index
    // (every sphere has ONE root it can be accepted with - its smallest one above tmin - whatever `closest` is when
    // it is tested). So a sphere B may be left out when some sphere A of the tile is hit by EVERY primary ray of the
    // tile with a root strictly below every root B can be accepted with. In distances along the ray (root x |d|):
    //   A: all lines of the cone pass the centre within sqrt(r^2 - 2^-12 (L^2 + r^2)); the literal discriminant is
    //      within 2^-18 a (L^2 + r^2) of the exact one (oc = o - c is ONE rounding per component, so every error is
    //      relative to |oc| and r), hence positive and known to 2^-6; the accepted root is the first one (the
    //      origin is outside by more than tmin |d|), at most l_A(psi + theta) + 2^-6 r + 2^-19 L = `up`;
    //   B: whatever its conditioning, a literal root is at least l_B(max(0, psi - theta)) - 2^-7 sqrt(L^2 + r^2) =
    //      `low` (l(phi) = L cos phi - sqrt(r^2 - L^2 sin^2 phi), increasing in phi; a literal hit on an exact
    //      miss lies within that margin of the tangent length).
    // up_min < low(B) drops B; the sphere that attains up_min has low < up and stays. Two passes over the spheres
    // (first up_min, then the candidates) instead of a per-thread array: the kernel keeps no scratch memory.
    double up_min = 1.0e300;
    bool any = false;
    for (uint32_t i = lane; i < a.n_spheres; i += stride) {
      double low, up;
      if (classify(i, low, up)) continue;
      any = true;
      if (!per_sphere && !list) break;  // only "sky tile or not" is wanted
      up_min = fmin(up_min, up);
    }
    if (WAVE) {  // the wave's lanes hold the spheres of ONE tile: combine
      any = __ballot(any) != 0ull;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) up_min = fmin(up_min, __shfl_xor(up_min, m, 64));
    }
    if (any && !per_sphere) mask = everything;
    if (any && (per_sphere || list)) {
      uint32_t kept = 0;
      if (WAVE) {
        for (uint32_t base = 0; base < a.n_spheres; base += 64u) {  // (uniform trip count: every lane takes part in the ballots)
          const uint32_t i = base + lane;
          double low, up;
          const bool keep = i < a.n_spheres && !(classify(i, low, up) || up_min < low);
          const unsigned long long m = __ballot(keep);
          const uint32_t at = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (keep && at < 15u) list[1u + at] = (uint16_t)i;  // ascending index: chunks in order, lanes in order within a chunk
          kept += (uint32_t)__popcll(m);
        }
      } else {
        for (uint32_t i = 0; i < a.n_spheres; ++i) {
          double low, up;
          if (classify(i, low, up) || up_min < low) continue;
          if (per_sphere) mask |= 1u << (cnt_pad - 1u - i);
          else if (kept < 15u) list[1u + kept] = (uint16_t)i;  // ascending index: the order hit_sphere's tie rule needs
          ++kept;
        }
      }
      if (!per_sphere) n_list = kept > 15u ? 0xffffu : kept;
    }
  }
  if (lane == 0u) {
    masks[j] = mask;
    if (list) list[0] = (uint16_t)n_list;
  }
}

// Tile order for the next launch: counting sort of this launch's per-tile cost, dearest first.
// One workgroup; order[] becomes a permutation of [0, n) whatever the costs are.
__global__ __launch_bounds__(1024) void rt_tile_order_kernel(const uint32_t* __restrict__ cost, const uint32_t* __restrict__ masks, uint32_t* __restrict__ order, uint32_t n) {
  // bucket 0 = dearest ... 255 = cheapest traced tile, 256 = tiles without cost: sky tiles, which the wave that fetches
  // them finishes on the spot. The traced tiles are handed out dearest first; the sky tiles are SPREAD EVENLY between them
  // (round 3): a sky unit is a few microseconds of work behind one atomic on the queue head, and at the end of the queue -
  // where a plain sort puts all of them - 7168 waves fetching nothing else ran into the rate of atomics on one address
  // (C3: 7 % of the launch with idle issue slots, profiles/r3_tile_order.md).
  __shared__ uint32_t s_max;
  __shared__ uint32_t s_hist[257];
  __shared__ uint32_t s_off[257];
  const uint32_t t = threadIdx.x;
  if (t == 0) s_max = 1u;
  if (t < 257u) s_hist[t] = 0u;
  __syncthreads();
  uint32_t mx = 0;
  for (uint32_t i = t; i < n; i += blockDim.x) mx = cost[i] > mx ? cost[i] : mx;
  atomicMax(&s_max, mx);
  __syncthreads();
  const float scale = 255.0f / (float)s_max;
  auto bucket = [&](uint32_t i) -> uint32_t {
    const uint32_t c = cost[i];
    if (masks ? masks[i] == 0u : c == 0u) return 256u;  // a sky tile by its mask (a launch that traced it anyway gave it a cost)
    const uint32_t b = 255u - (uint32_t)((float)c * scale);
    return b < 256u ? b : 0u;
  };
  for (uint32_t i = t; i < n; i += blockDim.x) atomicAdd(&s_hist[bucket(i)], 1u);
  __syncthreads();
  if (t == 0) {
    uint32_t acc = 0;
    for (int b = 0; b < 257; ++b) { s_off[b] = acc; acc += s_hist[b]; }
  }
  __syncthreads();
  const unsigned long long n_sky = s_hist[256], n_rest = n - s_hist[256];
  for (uint32_t i = t; i < n; i += blockDim.x) {
    const unsigned long long r = atomicAdd(&s_off[bucket(i)], 1u);  // rank in the plain sort: traced tiles, then sky tiles
    unsigned long long pos;
    if (r < n_rest) {
      pos = r + (n_sky ? r * n_sky / n_rest : 0ull);                      // floor(r n_sky / n_rest) sky tiles go before traced tile r
    } else {
      const unsigned long long sk = r - n_rest;
      unsigned long long before = n_rest ? ((sk + 1ull) * n_rest + n_sky - 1ull) / n_sky : 0ull;  // traced tiles before sky tile sk
      if (before > n_rest) before = n_rest;
      pos = sk + before;
    }
    order[pos] = i;
  }
}

// Frame table of a batched launch {bits(time) * 87189 + 189273 (wgsl:260), bits of cw = f32(frame) / f32(frame + 1) (wgsl:108)}
// written on the device from time bits passed by value (rt_render_frames_static: nothing staged in host memory).
struct rt_time_bits_arg { uint32_t bits[256]; };
__global__ __launch_bounds__(256) void rt_frame_table_kernel(uint2* __restrict__ table, const rt_time_bits_arg tb, uint32_t first_frame, uint32_t n) {
  const uint32_t i = threadIdx.x;
  if (i >= n) return;
  const float f = (float)(first_frame + i), g = (float)(first_frame + i + 1u);
  table[i] = make_uint2(tb.bits[i] * 87189u + 189273u, as_u32(f / g));  // IEEE division (no fast-math in this file), as frame_cw() on the host
}

// RGBA8 view of a float4 accumulation image (rt_read_rgba8 on RT_ACCUM_F32 contexts)
__global__ void rt_quantize_kernel(const f32x4* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4 c = src[i];
  dst[i] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (unorm8(c.w) << 24);
}

}  // namespace

hipError_t rt_launch_path_kernel(const rt_kargs& a, uint32_t n_waves, bool strict, bool count, bool split, hipStream_t stream) {
  if (n_waves == 0 || a.my_tiles == 0) return hipSuccess;
  const dim3 grid(n_waves), block(64);
  // SPLIT needs what makes its primary section correct: candidate masks (<= 32 spheres) or lists (> RT_SMALL_MAX), no counting
  const bool small = a.n_spheres <= RT_SMALL_MAX;
  if (split && (!a.tile_masks || (a.n_spheres > 32u && (small || !a.tile_lists)))) return hipErrorInvalidValue;
  // SMALL variants while the scene fits their LDS arrays. The behind cull they carry in strict mode saves exact tests
  // for one more instruction per sphere test (measured: 20 spheres -3 %, 1024 spheres +2 %). Counting launches keep
  // every candidate: the oracle's D counts their discriminants.
  if (strict) {
    if (count && split && small) hipLaunchKernelGGL((rt_path_kernel<true, true, true, true>), grid, block, 0, stream, a);
    else if (count && split) hipLaunchKernelGGL((rt_path_kernel<true, true, false, true>), grid, block, 0, stream, a);
    else if (count) hipLaunchKernelGGL((rt_path_kernel<true, true, false, false>), grid, block, 0, stream, a);
    else if (split && small) hipLaunchKernelGGL((rt_path_kernel<true, false, true, true>), grid, block, 0, stream, a);
    else if (split) hipLaunchKernelGGL((rt_path_kernel<true, false, false, true>), grid, block, 0, stream, a);
    else if (small) hipLaunchKernelGGL((rt_path_kernel<true, false, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rt_path_kernel<true, false, false, false>), grid, block, 0, stream, a);
  } else {
    if (count && split && small) hipLaunchKernelGGL((rt_path_kernel<false, true, true, true>), grid, block, 0, stream, a);
    else if (count && split) hipLaunchKernelGGL((rt_path_kernel<false, true, false, true>), grid, block, 0, stream, a);
    else if (count) hipLaunchKernelGGL((rt_path_kernel<false, true, false, false>), grid, block, 0, stream, a);
    else if (split && small) hipLaunchKernelGGL((rt_path_kernel<false, false, true, true>), grid, block, 0, stream, a);
    else if (split) hipLaunchKernelGGL((rt_path_kernel<false, false, false, true>), grid, block, 0, stream, a);
    else if (small) hipLaunchKernelGGL((rt_path_kernel<false, false, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((rt_path_kernel<false, false, false, false>), grid, block, 0, stream, a);
  }
  return hipGetLastError();
}

hipError_t rt_launch_combine_frames(const rt_kargs& a, hipStream_t stream) {
  if (!a.staging || !a.frame_table || a.n_frames == 0 || a.cov_w == 0 || a.cov_h == 0) return hipErrorInvalidValue;
  const dim3 grid((a.cov_w + 63u) / 64u, (a.cov_h + 3u) / 4u), block(256);
  hipLaunchKernelGGL(rt_combine_frames_kernel, grid, block, 0, stream, a);
  return hipGetLastError();
}

hipError_t rt_launch_tile_masks(const rt_kargs& a, uint32_t* masks, uint16_t* lists, hipStream_t stream) {
  if (a.my_tiles == 0) return hipSuccess;
  if (lists && a.n_spheres > 0xfff0u) lists = nullptr;  // list entries are 16 bits
  if (a.n_spheres > 64u) hipLaunchKernelGGL(rt_tile_masks_kernel<true>, dim3((a.my_tiles + 3u) / 4u), dim3(256), 0, stream, a, masks, lists);  // a wave per tile
  else hipLaunchKernelGGL(rt_tile_masks_kernel<false>, dim3((a.my_tiles + 255u) / 256u), dim3(256), 0, stream, a, masks, lists);
  return hipGetLastError();
}

hipError_t rt_launch_tile_order(const uint32_t* cost, const uint32_t* masks, uint32_t* order, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(rt_tile_order_kernel, dim3(1), dim3(1024), 0, stream, cost, masks, order, n);
  return hipGetLastError();
}

hipError_t rt_launch_frame_table(uint2* table, const uint32_t* time_bits_host, uint32_t first_frame, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  if (n > 256u) return hipErrorInvalidValue;
  rt_time_bits_arg tb{};
  for (uint32_t i = 0; i < n; ++i) tb.bits[i] = time_bits_host[i];
  hipLaunchKernelGGL(rt_frame_table_kernel, dim3(1), dim3(256), 0, stream, table, tb, first_frame, n);
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
