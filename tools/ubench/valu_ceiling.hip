// VALU issue ceiling of gfx950, measured so that the wall clock and the waves' own clocks must agree.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ceiling valu_ceiling.hip
//   ./valu_ceiling            full table (about 60 ms per cell)
//   ./valu_ceiling pmc        one short dispatch per instruction kind at 8 waves per SIMD, for a rocprofv3 --pmc pass
//
// What is different from valu_ubench.hip (whose two columns disagreed):
//  * 256 independent instructions per loop trip (the 3 scalar instructions of the loop are 1 % of the stream, not 16 %);
//  * every cell runs >= 50 ms, so launch ramp and tail are < 0.1 %;
//  * EVERY wave stamps s_memtime / s_memrealtime around its loop and reads its HW_ID / XCC_ID: the table gives, over all
//    SIMDs, the span from the first loop start to the last loop end of the SIMD's waves per instruction issued on it
//    (must agree with the wall column), the waves' own cycles per instruction (they do NOT interleave evenly: the oldest
//    waves issue as if alone and finish first - which is what made round 1's single-wave "real" column meaningless),
//    the number of distinct SIMDs seen and the min / max waves per SIMD - placement is verified, not assumed;
//  * residency is forced by dynamic LDS (160 KiB / (4 w) per single-wave workgroup), so w waves sit on every SIMD;
//  * the wall column uses the clock the waves measured (s_memtime ticks per 100 MHz s_memrealtime tick), not 2.4 GHz.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BODY = 256;  // vector instructions per loop trip

typedef float f2 __attribute__((ext_vector_type(2)));

// 16 independent destination registers %0..%15, two more inputs %16 %17 (VGPR) and %18 (SGPR / SGPR pair)
#define REPT16(T) ".rept 16\n" T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(8) T(9) T(10) T(11) T(12) T(13) T(14) T(15) ".endr\n"
#define OUTS(r) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), \
                "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

enum Kind { K_FMA, K_FMA_SGPR, K_ADD, K_MUL, K_PK_FMA, K_PK_FMA_SGPR, K_PK_MUL, K_ALIGNBIT, K_BITOP3, K_AND_OR, K_XOR, K_LSHL, K_LSHL_XOR,
            K_MAD_U64, K_MUL_LO, K_SQRT, K_RCP, K_CNDMASK, K_CMP, K_MOV, K_FMA_SALU5, K_FMA_SALU2, K_FMA_SALU1, K_FMA_DEP, K_MIX_PATH, K_COUNT };
static const char* kind_names[K_COUNT] = {
    "v_fma_f32", "v_fma_f32 (sgpr src)", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_fma_f32 (sgpr pair, op_sel)", "v_pk_mul_f32",
    "v_alignbit_b32", "v_bitop3_b32", "v_and_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshlrev+v_xor (per instr)", "v_mad_u64_u32", "v_mul_lo_u32",
    "v_sqrt_f32", "v_rcp_f32", "v_cndmask_b32 (vcc)", "v_cmp_lt_f32 (vcc)", "v_mov_b32", "5 v_fma : 1 s_add (per VALU)", "2 v_fma : 1 s_add (per VALU)",
    "1 v_fma : 1 s_add (per VALU)", "v_fma_f32 dependent chain", "path mix 8fma:4pk:2bitop:1alignbit:1mad64 (per instr)"};

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* __restrict__ stamps, float* __restrict__ out, int iters, float s0) {
  extern __shared__ char lds_pad[];
  float r[16];
  unsigned u[16];
  f2 p[16];
  for (int i = 0; i < 16; ++i) { r[i] = 1.0f + threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7919u + i * 104729u; p[i] = f2{r[i], 0.5f + i}; }
  float va = 1.0001f + threadIdx.x * 1e-7f, vb = 0.25f;
  unsigned ua = 0x9e3779b9u + threadIdx.x, ub = 0x7f4a7c15u;
  f2 pa = f2{va, vb};
  const float sc = s0;
  const f2 sp = f2{s0, s0 * 0.5f};
  unsigned long long s64 = 182489729ull + (unsigned long long)(s0 > 1e30f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == K_FMA) {
#define T(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(va), "v"(vb));
#undef T
    } else if (KIND == K_FMA_SGPR) {
#define T(i) "v_fma_f32 %" #i ", %" #i ", %17, %16\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(vb), "s"(sc));
#undef T
    } else if (KIND == K_ADD) {
#define T(i) "v_add_f32 %" #i ", %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(vb));
#undef T
    } else if (KIND == K_MUL) {
#define T(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(va));
#undef T
    } else if (KIND == K_PK_FMA) {
#define T(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %16\n"
      asm volatile(REPT16(T) : OUTS(p) : "v"(pa));
#undef T
    } else if (KIND == K_PK_FMA_SGPR) {  // the form phase 1 uses: SGPR pair x broadcast VGPR half + accumulator
#define T(i) "v_pk_fma_f32 %" #i ", %17, %16, %" #i " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
      asm volatile(REPT16(T) : OUTS(p) : "v"(pa), "s"(sp));
#undef T
    } else if (KIND == K_PK_MUL) {
#define T(i) "v_pk_mul_f32 %" #i ", %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(p) : "v"(pa));
#undef T
    } else if (KIND == K_ALIGNBIT) {
#define T(i) "v_alignbit_b32 %" #i ", %" #i ", %16, 31\n"
      asm volatile(REPT16(T) : OUTS(u) : "v"(ua));
#undef T
    } else if (KIND == K_BITOP3) {
#define T(i) "v_bitop3_b32 %" #i ", %" #i ", %16, %17 bitop3:0xea\n"
      asm volatile(REPT16(T) : OUTS(u) : "v"(ua), "v"(ub));
#undef T
    } else if (KIND == K_AND_OR) {
#define T(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
      asm volatile(REPT16(T) : OUTS(u) : "v"(ua), "v"(ub));
#undef T
    } else if (KIND == K_XOR) {
#define T(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(u) : "v"(ua));
#undef T
    } else if (KIND == K_LSHL) {
#define T(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
      asm volatile(REPT16(T) : OUTS(u));
#undef T
    } else if (KIND == K_LSHL_XOR) {  // the RNG's shape: shift into a temporary, xor back (8 chains, 16 instructions)
      asm volatile(".rept 16\n"
                   "v_lshlrev_b32 %8, 15, %0\n v_lshlrev_b32 %9, 15, %1\n v_lshlrev_b32 %10, 15, %2\n v_lshlrev_b32 %11, 15, %3\n"
                   "v_lshlrev_b32 %12, 15, %4\n v_lshlrev_b32 %13, 15, %5\n v_lshlrev_b32 %14, 15, %6\n v_lshlrev_b32 %15, 15, %7\n"
                   "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %10\n v_xor_b32 %3, %3, %11\n"
                   "v_xor_b32 %4, %4, %12\n v_xor_b32 %5, %5, %13\n v_xor_b32 %6, %6, %14\n v_xor_b32 %7, %7, %15\n"
                   ".endr\n" : OUTS(u));
    } else if (KIND == K_MAD_U64) {  // rng * A + C as the path kernel issues it (VGPR multiplier, SGPR-pair addend); dst pair, low half reused
      unsigned long long d[8];
      asm volatile(".rept 32\n"
                   "v_mad_u64_u32 %0, vcc, %8, %16, %17\n v_mad_u64_u32 %1, vcc, %9, %16, %17\n v_mad_u64_u32 %2, vcc, %10, %16, %17\n v_mad_u64_u32 %3, vcc, %11, %16, %17\n"
                   "v_mad_u64_u32 %4, vcc, %12, %16, %17\n v_mad_u64_u32 %5, vcc, %13, %16, %17\n v_mad_u64_u32 %6, vcc, %14, %16, %17\n v_mad_u64_u32 %7, vcc, %15, %16, %17\n"
                   ".endr\n"
                   : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                   : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]), "v"(ua), "s"(s64) : "vcc");
      for (int i = 0; i < 8; ++i) u[i] += (unsigned)d[i];
    } else if (KIND == K_MUL_LO) {
#define T(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(u) : "v"(ua));
#undef T
    } else if (KIND == K_SQRT) {
#define T(i) "v_sqrt_f32 %" #i ", %" #i "\n"
      asm volatile(REPT16(T) : OUTS(r));
#undef T
    } else if (KIND == K_RCP) {
#define T(i) "v_rcp_f32 %" #i ", %" #i "\n"
      asm volatile(REPT16(T) : OUTS(r));
#undef T
    } else if (KIND == K_CNDMASK) {
#define T(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
      asm volatile("v_cmp_lt_f32 vcc, %16, %17\n" REPT16(T) : OUTS(r) : "v"(va), "v"(vb) : "vcc");
#undef T
    } else if (KIND == K_CMP) {
#define T(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(va) : "vcc");
#undef T
    } else if (KIND == K_MOV) {
#define T(i) "v_mov_b32 %" #i ", %16\n"
      asm volatile(REPT16(T) : OUTS(r) : "v"(va));
#undef T
    } else if (KIND == K_FMA_SALU5) {  // 240 VALU + 48 SALU per trip; reported per VALU instruction
      unsigned sa = 1;
      asm volatile(".rept 48\n"
                   "v_fma_f32 %0, %0, %17, %18\n v_fma_f32 %1, %1, %17, %18\n v_fma_f32 %2, %2, %17, %18\n v_fma_f32 %3, %3, %17, %18\n v_fma_f32 %4, %4, %17, %18\n"
                   "s_add_u32 %16, %16, 3\n"
                   ".endr\n"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                     "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+s"(sa) : "v"(va), "v"(vb) : "scc");
      u[0] += sa;
    } else if (KIND == K_FMA_SALU2) {  // 256 VALU + 128 SALU
      unsigned sa = 1;
      asm volatile(".rept 128\n"
                   "v_fma_f32 %0, %0, %17, %18\n v_fma_f32 %1, %1, %17, %18\n"
                   "s_add_u32 %16, %16, 3\n"
                   ".endr\n"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                     "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+s"(sa) : "v"(va), "v"(vb) : "scc");
      u[0] += sa;
    } else if (KIND == K_FMA_SALU1) {  // 256 VALU + 256 SALU
      unsigned sa = 1;
      asm volatile(".rept 128\n"
                   "v_fma_f32 %0, %0, %17, %18\n s_add_u32 %16, %16, 3\n v_fma_f32 %1, %1, %17, %18\n s_add_u32 %16, %16, 5\n"
                   ".endr\n"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                     "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+s"(sa) : "v"(va), "v"(vb) : "scc");
      u[0] += sa;
    } else if (KIND == K_FMA_DEP) {
      asm volatile(".rept 256\n v_fma_f32 %0, %0, %1, %2\n .endr\n" : "+v"(r[0]) : "v"(va), "v"(vb));
    } else if (KIND == K_MIX_PATH) {  // roughly the path kernel's hot mix: plain f32, packed FMA, bit ops, RNG multiply
      unsigned long long d0, d1 = u[1];
      asm volatile(".rept 16\n"
                   "v_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %8, %18, %19, %8 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_fma_f32 %1, %1, %16, %17\n v_bitop3_b32 %12, %12, %20, %21 bitop3:0xea\n"
                   "v_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %9, %18, %19, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_fma_f32 %3, %3, %16, %17\n v_alignbit_b32 %13, %13, %20, 31\n"
                   "v_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %10, %18, %19, %10 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_fma_f32 %5, %5, %16, %17\n v_bitop3_b32 %14, %14, %20, %21 bitop3:0xea\n"
                   "v_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %11, %18, %19, %11 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_fma_f32 %7, %7, %16, %17\n v_mad_u64_u32 %15, vcc, %23, %20, %22\n"
                   ".endr\n"
                   : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]),
                     "+v"(u[12]), "+v"(u[13]), "+v"(u[14]), "=&v"(d0)
                   : "v"(va), "v"(vb), "s"(sp), "v"(pa), "v"(ua), "v"(ub), "s"(s64), "v"(u[1]) : "vcc");
      u[0] = (unsigned)d0 + (unsigned)d1;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    unsigned long long* s = stamps + 4ull * blockIdx.x;
    s[0] = t1 - t0; s[1] = w1 - w0; s[2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32); s[3] = w0;
  }
  // only the registers this kind works on stay live across the loop (<= 64 VGPRs: 8 waves per SIMD)
  constexpr bool use_p = KIND == K_PK_FMA || KIND == K_PK_FMA_SGPR || KIND == K_PK_MUL || KIND == K_MIX_PATH;
  constexpr bool use_u = KIND == K_ALIGNBIT || KIND == K_BITOP3 || KIND == K_AND_OR || KIND == K_XOR || KIND == K_LSHL || KIND == K_LSHL_XOR ||
                         KIND == K_MAD_U64 || KIND == K_MUL_LO || KIND == K_MIX_PATH || KIND == K_FMA_SALU5 || KIND == K_FMA_SALU2 || KIND == K_FMA_SALU1;
  constexpr bool use_r = !use_p && !use_u || KIND == K_MIX_PATH || KIND == K_FMA_SALU5 || KIND == K_FMA_SALU2 || KIND == K_FMA_SALU1;
  float acc = 0;
  for (int i = 0; i < 16; ++i) acc += (use_r ? r[i] : 0.0f) + (use_p ? p[i].x + p[i].y : 0.0f) + (use_u ? (float)u[i] : 0.0f);
  if (acc == 12345.678f) out[0] = acc + lds_pad[threadIdx.x];
}

// vector instructions per loop trip that the row's "per instruction" refers to
static int valu_per_trip(int kind) { return kind == K_FMA_SALU5 ? 240 : BODY; }

struct cell { double wall, mean, mn, mx, own_mean, own_mn, own_mx, ghz; int simds, wmin, wmax; float ms; };

template <int KIND>
static cell run(int w, int cus, unsigned long long* d_st, float* d_out, double target_instr_per_simd) {
  const int n_waves = cus * 4 * w;
  const size_t lds = std::min<size_t>(65536, (160 * 1024) / (4 * w)) & ~size_t(255);  // w single-wave workgroups per SIMD fit, w + 1 do not
  const int iters = std::max(1, (int)(target_instr_per_simd / ((double)w * valu_per_trip(KIND))));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(n_waves), dim3(64), lds, 0, d_st, d_out, iters / 4 + 1, 1.0001f);  // warm-up (settles the clock)
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(n_waves), dim3(64), lds, 0, d_st, d_out, iters, 1.0001f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  cell c{};
  CHECK(hipEventElapsedTime(&c.ms, e0, e1));
  std::vector<unsigned long long> st(4ull * n_waves);
  CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  struct simd_t { int n = 0; unsigned long long first = ~0ull, last = 0; };
  std::map<unsigned long long, simd_t> per_simd;
  // placement key: XCC, SE, SH, CU, SIMD (HW_ID bits 4-5 SIMD, 8-11 CU, 12 SH, 13-15 SE; pipe / wave slot left out)
  auto key = [](unsigned long long v) { return ((v >> 32) << 16) | (v & 0xff30ull); };
  double ticks = 0, rticks = 0;
  for (int i = 0; i < n_waves; ++i) {
    simd_t& sd = per_simd[key(st[4 * i + 2])];
    sd.n++; sd.first = std::min(sd.first, st[4 * i + 3]); sd.last = std::max(sd.last, st[4 * i + 3] + st[4 * i + 1]);
    ticks += (double)st[4 * i]; rticks += (double)st[4 * i + 1];
  }
  c.ghz = ticks / rticks * 0.1;
  c.simds = (int)per_simd.size();
  c.wmin = 1 << 30; c.wmax = 0;
  const double instr = (double)iters * valu_per_trip(KIND);
  // in-wave view of a SIMD: from the first loop start to the last loop end of ITS waves (100 MHz stamps), in shader
  // cycles, per vector instruction issued on it
  c.mn = 1e30; c.mx = 0;
  for (auto& kv : per_simd) {
    c.wmin = std::min(c.wmin, kv.second.n); c.wmax = std::max(c.wmax, kv.second.n);
    const double v = (double)(kv.second.last - kv.second.first) * 10.0 * c.ghz / (instr * kv.second.n);
    c.mean += v / per_simd.size(); c.mn = std::min(c.mn, v); c.mx = std::max(c.mx, v);
  }
  // a wave's own view: its cycles per own instruction (4 = it issued as if alone; 4 w = perfectly interleaved with w - 1 others)
  c.own_mn = 1e30; c.own_mx = 0;
  for (int i = 0; i < n_waves; ++i) {
    const double v = (double)st[4 * i] / instr;
    c.own_mean += v / n_waves; c.own_mn = std::min(c.own_mn, v); c.own_mx = std::max(c.own_mx, v);
  }
  c.wall = c.ms * 1e-3 * c.ghz * 1e9 / (instr * w);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return c;
}

template <int KIND>
static void row(int cus, unsigned long long* d_st, float* d_out) {
  printf("%-52s", kind_names[KIND]);
  for (int w : {1, 2, 4, 7, 8}) {
    const cell c = run<KIND>(w, cus, d_st, d_out, 64e6);
    printf(" | %5.2f %5.2f [%5.2f %5.2f] own %5.1f [%4.1f %5.1f] %4.2fG %4d %d-%d %3.0fms", c.wall, c.mean, c.mn, c.mx, c.own_mean, c.own_mn, c.own_mx, c.ghz, c.simds,
           c.wmin, c.wmax, c.ms);
  }
  printf("\n");
  fflush(stdout);
}

template <int KIND>
static void pmc_dispatch(int cus, unsigned long long* d_st, float* d_out) {
  const int w = 8, n_waves = cus * 4 * w, iters = 4000;
  hipLaunchKernelGGL(k<KIND>, dim3(n_waves), dim3(64), 5120, 0, d_st, d_out, iters, 1.0001f);
  CHECK(hipDeviceSynchronize());
  printf("pmc k<%d> %-52s waves %d  vector instructions per wave %lld  scalar %lld\n", KIND, kind_names[KIND], n_waves,
         (long long)iters * valu_per_trip(KIND) + (KIND == K_CNDMASK ? iters : 0),
         (long long)iters * (KIND == K_FMA_SALU5 ? 48 : KIND == K_FMA_SALU2 ? 128 : KIND == K_FMA_SALU1 ? 256 : 0));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_st; float* d_out;
  CHECK(hipMalloc(&d_st, 4ull * 8 * cus * 4 * 8)); CHECK(hipMalloc(&d_out, 4));
#define ALL(F) F<K_FMA>(cus, d_st, d_out); F<K_FMA_SGPR>(cus, d_st, d_out); F<K_ADD>(cus, d_st, d_out); F<K_MUL>(cus, d_st, d_out); F<K_PK_FMA>(cus, d_st, d_out); \
  F<K_PK_FMA_SGPR>(cus, d_st, d_out); F<K_PK_MUL>(cus, d_st, d_out); F<K_ALIGNBIT>(cus, d_st, d_out); F<K_BITOP3>(cus, d_st, d_out); F<K_AND_OR>(cus, d_st, d_out); \
  F<K_XOR>(cus, d_st, d_out); F<K_LSHL>(cus, d_st, d_out); F<K_LSHL_XOR>(cus, d_st, d_out); F<K_MAD_U64>(cus, d_st, d_out); F<K_MUL_LO>(cus, d_st, d_out); \
  F<K_SQRT>(cus, d_st, d_out); F<K_RCP>(cus, d_st, d_out); F<K_CNDMASK>(cus, d_st, d_out); F<K_CMP>(cus, d_st, d_out); F<K_MOV>(cus, d_st, d_out); \
  F<K_FMA_SALU5>(cus, d_st, d_out); F<K_FMA_SALU2>(cus, d_st, d_out); F<K_FMA_SALU1>(cus, d_st, d_out); F<K_FMA_DEP>(cus, d_st, d_out); F<K_MIX_PATH>(cus, d_st, d_out);
  if (argc > 1 && !strcmp(argv[1], "pmc")) {
    ALL(pmc_dispatch)
    return 0;
  }
  printf("device %s, %d CUs (%d SIMDs), nominal clock %d kHz\n", prop.gcnArchName, cus, cus * 4, prop.clockRate);
  printf("columns per waves-per-SIMD w = 1, 2, 4, 7, 8:  wall | per-SIMD mean [min max] | own mean [min max] | clock | SIMDs seen | waves per SIMD min-max | kernel ms\n");
  printf("  wall    = kernel time (hipEvents) x measured clock / (vector instructions per wave x w): SIMD cycles per wave64 instruction\n");
  printf("  per-SIMD = (last loop end - first loop start of the SIMD's waves, s_memrealtime) x clock / vector instructions issued on that SIMD, over all SIMDs\n");
  printf("  own      = a wave's s_memtime ticks per OWN instruction: 4 = it ran as if alone, 4 w = evenly interleaved with the w - 1 others\n");
  ALL(row)
  return 0;
}
