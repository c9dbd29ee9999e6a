// Do FP32 MFMA waves and VALU waves on the SAME SIMD overlap on gfx950? (the round-1 test interleaved
// both kinds in one wave's stream, which measures that wave's issue order, not the pipes.)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coexec mfma_coexec.hip && ./mfma_coexec
// 512-thread workgroups, 4 per CU (forced by 40 KiB of dynamic LDS each): 8 waves per SIMD, of which waves 0-3 of a workgroup
// (one per SIMD) take the VALU role and waves 4-7 the MFMA role, so every SIMD holds 4 VALU and 4 MFMA waves - checked per
// wave with HW_ID. Modes: VALU waves alone (the others exit), MFMA waves alone, both, and every wave interleaving both kinds
// in its own stream (8 VALU per MFMA). Every wave stamps its loop; wall time by hipEvents.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { M_VALU_ONLY, M_MFMA_ONLY, M_BOTH, M_INTERLEAVED, M_ALL_MFMA, M_ALL_VALU };
enum { V_FMA, V_PK_FMA, V_INT };

// 256 vector instructions
template <int VK>
__device__ __forceinline__ void valu_trip(float (&r)[8], f2 (&p)[8], unsigned (&u)[8], float va, float vb, f2 sp, unsigned ua) {
  if (VK == V_FMA) {
    asm volatile(".rept 32\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n .endr\n"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(va), "v"(vb));
  } else if (VK == V_PK_FMA) {
    asm volatile(".rept 32\n v_pk_fma_f32 %0, %9, %8, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %9, %8, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %2, %9, %8, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %9, %8, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %4, %9, %8, %4 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %5, %9, %8, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                 "v_pk_fma_f32 %6, %9, %8, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %7, %9, %8, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n .endr\n"
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(f2{va, vb}), "s"(sp));
  } else {
    asm volatile(".rept 32\n v_xor_b32 %0, %0, %8\n v_alignbit_b32 %1, %1, %8, 31\n v_xor_b32 %2, %2, %8\n v_bitop3_b32 %3, %3, %8, %8 bitop3:0xea\n"
                 "v_xor_b32 %4, %4, %8\n v_alignbit_b32 %5, %5, %8, 31\n v_xor_b32 %6, %6, %8\n v_bitop3_b32 %7, %7, %8, %8 bitop3:0xea\n .endr\n"
                 : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]) : "v"(ua));
  }
}
// 64 MFMAs on four independent accumulators (dependent latency 40 cycles < 4 x 32)
__device__ __forceinline__ void mfma_trip(f4 (&acc)[4], float a, float b) {
  asm volatile(".rept 16\n v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
               "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n .endr\n"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b));
}
// interleaved in ONE stream: per MFMA 8 vector instructions (64 MFMA + 512 VALU per trip)
template <int VK>
__device__ __forceinline__ void mixed_trip(f4 (&acc)[4], float (&r)[8], float a, float b, float va, float vb) {
  asm volatile(".rept 16\n"
               "v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n"
               "v_fma_f32 %8, %8, %14, %15\n v_fma_f32 %9, %9, %14, %15\n v_fma_f32 %10, %10, %14, %15\n v_fma_f32 %11, %11, %14, %15\n"
               "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n"
               "v_fma_f32 %8, %8, %14, %15\n v_fma_f32 %9, %9, %14, %15\n v_fma_f32 %10, %10, %14, %15\n v_fma_f32 %11, %11, %14, %15\n"
               "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n"
               "v_fma_f32 %8, %8, %14, %15\n v_fma_f32 %9, %9, %14, %15\n v_fma_f32 %10, %10, %14, %15\n v_fma_f32 %11, %11, %14, %15\n"
               "v_mfma_f32_16x16x4_f32 %3, %12, %13, %3\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n"
               "v_fma_f32 %8, %8, %14, %15\n v_fma_f32 %9, %9, %14, %15\n v_fma_f32 %10, %10, %14, %15\n v_fma_f32 %11, %11, %14, %15\n"
               ".endr\n"
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
               : "v"(a), "v"(b), "v"(va), "v"(vb));
}

template <int MODE, int VK>
__global__ __launch_bounds__(512) void k(unsigned long long* __restrict__ stamps, float* __restrict__ out, int trips_v, int trips_m, float s0, int opts) {
  extern __shared__ char lds_pad[];
  const unsigned wave = threadIdx.x >> 6;
  // opts bit 0: the MFMA role goes to waves 0-3 (the OLDER waves of each SIMD); bit 1: MFMA waves raise their priority
  const bool mfma_role = MODE == M_ALL_MFMA || (MODE != M_ALL_VALU && MODE != M_INTERLEAVED && ((opts & 1) ? wave < 4u : wave >= 4u));
  if ((opts & 2) && mfma_role) __builtin_amdgcn_s_setprio(3);
  const bool works = MODE == M_BOTH || MODE == M_INTERLEAVED || MODE == M_ALL_MFMA || MODE == M_ALL_VALU || (MODE == M_VALU_ONLY && !mfma_role) ||
                     (MODE == M_MFMA_ONLY && mfma_role);
  float r[8]; f2 p[8]; unsigned u[8]; f4 acc[4];
  for (int i = 0; i < 8; ++i) { r[i] = 1.0f + threadIdx.x * 0.001f + i; p[i] = f2{r[i], 0.5f}; u[i] = threadIdx.x * 7919u + i; }
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.0f, 1.0f, 2.0f, 3.0f};
  const float va = 1.0001f + threadIdx.x * 1e-7f, vb = 0.25f, a = 1.0f + (threadIdx.x & 15) * 0.01f, b = 0.5f;
  const f2 sp = f2{s0, s0 * 0.5f};
  const unsigned ua = 0x9e3779b9u + threadIdx.x;
  unsigned long long t0 = 0, t1 = 0, w0 = 0, w1 = 0;
  if (works) {  // wave-uniform
    t0 = __builtin_amdgcn_s_memtime(); w0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == M_INTERLEAVED) {
      for (int it = 0; it < trips_m; ++it) mixed_trip<VK>(acc, r, a, b, va, vb);
    } else if (mfma_role) {
      for (int it = 0; it < trips_m; ++it) mfma_trip(acc, a, b);
    } else {
      for (int it = 0; it < trips_v; ++it) valu_trip<VK>(r, p, u, va, vb, sp, ua);
    }
    t1 = __builtin_amdgcn_s_memtime(); w1 = __builtin_amdgcn_s_memrealtime();
  }
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63u) == 0u) {
    unsigned long long* s = stamps + 4ull * (blockIdx.x * 8u + wave);
    s[0] = t1 - t0; s[1] = w1 - w0; s[2] = (unsigned long long)hwid | ((unsigned long long)xcc << 32); s[3] = (mfma_role ? 1ull : 0ull) | (works ? 2ull : 0ull);
  }
  float x = 0;
  for (int i = 0; i < 8; ++i) x += r[i] + p[i].x + p[i].y + (float)u[i];
  for (int i = 0; i < 4; ++i) x += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (x == 12345.678f) out[0] = x + lds_pad[threadIdx.x];
}

template <int MODE, int VK>
static void run(const char* label, int cus, unsigned long long* d_st, float* d_out, int trips_v, int trips_m, int opts = 0) {
  const int n_wg = cus * 4, n_waves = n_wg * 8;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, VK>), dim3(n_wg), dim3(512), 40960, 0, d_st, d_out, trips_v / 4 + 1, trips_m / 4 + 1, 1.0001f, opts);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, VK>), dim3(n_wg), dim3(512), 40960, 0, d_st, d_out, trips_v, trips_m, 1.0001f, opts);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> st(4ull * n_waves);
  CHECK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
  auto key = [](unsigned long long v) { return ((v >> 32) << 16) | (v & 0xff30ull); };
  std::map<unsigned long long, std::pair<int, int>> per_simd;  // waves in the VALU / MFMA role
  double tick[2] = {0, 0}, rtick[2] = {0, 0}, mx[2] = {0, 0};
  int n[2] = {0, 0};
  for (int i = 0; i < n_waves; ++i) {
    const int role = (int)(st[4 * i + 3] & 1ull);
    auto& c = per_simd[key(st[4 * i + 2])];
    (role ? c.second : c.first)++;
    if (!(st[4 * i + 3] & 2ull)) continue;
    tick[role] += (double)st[4 * i]; rtick[role] += (double)st[4 * i + 1]; mx[role] = std::max(mx[role], (double)st[4 * i]); n[role]++;
  }
  int balanced = 0;
  for (auto& kv : per_simd) balanced += kv.second.first == 4 && kv.second.second == 4;
  printf("%-44s wall %7.2f ms |", label, ms);
  const bool inter = MODE == M_INTERLEAVED;
  for (int role = 0; role < 2; ++role) {
    if (!n[role]) { printf(" %s: -                                   |", role ? "MFMA waves" : "VALU waves"); continue; }
    const double cyc = tick[role] / n[role], ghz = tick[role] / rtick[role] * 0.1;
    // SIMD cycles per instruction of this role: the wave's cycles / its instructions / waves of the role on the SIMD
    const double wr = (MODE == M_ALL_MFMA || MODE == M_ALL_VALU || inter) ? 8.0 : 4.0;
    const double per = role ? cyc / (64.0 * trips_m) / wr : cyc / ((inter ? 512.0 * trips_m : 256.0 * trips_v)) / wr;
    printf(" %s: mean %6.2f ms (max %6.2f) %6.2f cyc/instr/SIMD @%4.2f GHz |", role ? "MFMA waves" : (inter ? "mixed waves (per VALU)" : "VALU waves"),
           cyc / (ghz * 1e6), mx[role] / (ghz * 1e6), per, ghz);
  }
  printf(" SIMDs %zu, with 4 VALU + 4 MFMA waves: %d\n", per_simd.size(), balanced);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* d_st; float* d_out;
  CHECK(hipMalloc(&d_st, 4ull * 8 * cus * 4 * 8)); CHECK(hipMalloc(&d_out, 4));
  printf("device %s, %d CUs. v_mfma_f32_16x16x4_f32 (2048 FLOP) against vector instructions, 4 + 4 waves per SIMD\n", prop.gcnArchName, cus);
  const int tm = 4000;  // 4 waves x 4000 x 64 MFMA x 32 cycles = 32.8 M cycles per SIMD
#define SET(VK, name, tv)                                                                     \
  run<M_VALU_ONLY, VK>(name ": VALU waves alone (4 per SIMD)", cus, d_st, d_out, tv, tm);  \
  run<M_MFMA_ONLY, VK>(name ": MFMA waves alone (4 per SIMD)", cus, d_st, d_out, tv, tm);  \
  run<M_BOTH, VK>(name ": both (4 + 4 per SIMD)", cus, d_st, d_out, tv, tm);
  SET(V_FMA, "v_fma_f32", 16000)
  SET(V_PK_FMA, "v_pk_fma_f32", 8000)
  SET(V_INT, "xor/alignbit/bitop3", 12000)
  run<M_BOTH, V_FMA>("v_fma_f32: both, MFMA waves are the older ones", cus, d_st, d_out, 16000, tm, 1);
  run<M_BOTH, V_FMA>("v_fma_f32: both, MFMA waves at s_setprio 3", cus, d_st, d_out, 16000, tm, 2);
  run<M_BOTH, V_PK_FMA>("v_pk_fma_f32: both, MFMA waves older", cus, d_st, d_out, 8000, tm, 1);
  run<M_BOTH, V_PK_FMA>("v_pk_fma_f32: both, MFMA waves at s_setprio 3", cus, d_st, d_out, 8000, tm, 2);
  run<M_BOTH, V_INT>("xor/alignbit/bitop3: both, MFMA waves older", cus, d_st, d_out, 12000, tm, 1);
  run<M_BOTH, V_INT>("xor/alignbit/bitop3: both, MFMA waves at s_setprio 3", cus, d_st, d_out, 12000, tm, 2);
  run<M_ALL_VALU, V_FMA>("v_fma_f32: all 8 waves VALU", cus, d_st, d_out, 8000, tm);
  run<M_ALL_MFMA, V_FMA>("all 8 waves MFMA", cus, d_st, d_out, 8000, tm / 2);
  run<M_INTERLEAVED, V_FMA>("one stream: 8 v_fma per MFMA, all 8 waves", cus, d_st, d_out, 8000, tm / 2);
  return 0;
}
