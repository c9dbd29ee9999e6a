// VALU throughput microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the
// instruction kinds the path tracer uses, at 1/2/4/8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ubench valu_ubench.hip && ./valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 32768;
constexpr int UNROLL = 16;  // 16 independent instructions per iteration (8 regs x 2)

#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, float s0, int n, unsigned long long* stamps) {
  float r[8], q[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[8];
  for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 0.001f + i; q[i] = i + 0.5f; p[i] = f2{r[i], q[i]}; }
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7919u + i;
  const float c0 = s0, c1 = s0 * 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < n; ++it) {
    if (KIND == 0) {  // v_fma_f32 vgpr operands
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(c1), "v"(c0));
      BODY8(OP)
#undef OP
    } else if (KIND == 1) {  // v_pk_fma_f32
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 2) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 2) {  // v_sub_f32 with SGPR operand
#define OP(i) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(r[i]) : "s"(c0)); asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(q[i]) : "s"(c1));
      BODY8(OP)
#undef OP
    } else if (KIND == 3) {  // v_alignbit_b32
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(r[i])); asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(q[i]));
      BODY8(OP)
#undef OP
    } else if (KIND == 4) {  // v_mul_lo_u32
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 5) {  // v_sqrt_f32
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i])); asm volatile("v_sqrt_f32 %0, %0" : "+v"(q[i]));
      BODY8(OP)
#undef OP
    } else if (KIND == 6) {  // v_rcp_f32
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])); asm volatile("v_rcp_f32 %0, %0" : "+v"(q[i]));
      BODY8(OP)
#undef OP
    } else if (KIND == 7) {  // v_xor + v_lshlrev (int ALU)
#define OP(i) asm volatile("v_lshlrev_b32 %0, 15, %1" : "=v"(u[(i + 1) & 7]) : "v"(u[i])); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 8) {  // v_cmp + v_cndmask
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(q[i]), "v"(c0) : "vcc");
      BODY8(OP)
#undef OP
    } else if (KIND == 9) {  // v_mad_u32_u24
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 3) & 7]), "v"(u[(i + 4) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 10) {  // v_mul_f32 (e32)
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c1)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(c0));
      BODY8(OP)
#undef OP
    } else if (KIND == 11) {  // v_fma_f32 with SGPR operand (as the hit loop uses)
#define OP(i) asm volatile("v_fma_f32 %0, %0, %0, -%1" : "+v"(r[i]) : "s"(c0)); asm volatile("v_fma_f32 %0, %0, %0, -%1" : "+v"(q[i]) : "s"(c1));
      BODY8(OP)
#undef OP
    } else if (KIND == 12) {  // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7])); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 2) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 13) {  // dependent chain v_fma_f32
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(q[i]), "v"(c1)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(c1), "v"(c0));
      BODY8(OP)
#undef OP
    } else if (KIND == 14) {  // v_fma_f32 interleaved 1:1 with SALU (does scalar issue cost VALU slots?)
      unsigned sa = 1, sb = 2;
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa) : "s"(sb) : "scc"); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(c1), "v"(c0)); asm volatile("s_xor_b32 %0, %0, %1" : "+s"(sb) : "s"(sa) : "scc");
      BODY8(OP)
#undef OP
      u[0] += sa + sb;
    } else if (KIND == 15) {  // v_fma_f32 interleaved 2:1 with SALU
      unsigned sa = 1, sb = 2;
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa) : "s"(sb) : "scc"); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(c1), "v"(c0));
      BODY8(OP)
#undef OP
      u[0] += sa + sb;
    } else if (KIND == 17) {  // v_pk_fma_f32 with an SGPR-pair source and broadcast (op_sel) VGPR sources, as a packed hit loop would use
      f2 sc = f2{c0, c1};
#define OP(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(sc), "v"(p[(i + 1) & 7])); asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(p[i]) : "s"(sc), "v"(p[(i + 2) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 18) {  // v_mad_u64_u32: rng * A + C in one instruction (VGPR multiplier, SGPR-pair addend)
      unsigned long long cadd = 182489729ull + (unsigned long long)(c0 > 1e30f);
      unsigned long long d64;
      unsigned mulv = 18197u + (threadIdx.x >> 8);
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(d64) : "v"(u[i]), "v"(mulv), "s"(cadd) : "vcc"); u[i] = (unsigned)d64;
      BODY8(OP) BODY8(OP)
#undef OP
    } else if (KIND == 19) {  // v_and_or_b32
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])); asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 3) & 7]), "v"(u[(i + 4) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 20) {  // v_xor_b32 alone
#define OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 21) {  // v_add_u32 alone
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 22) {  // v_rsq_f32
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i])); asm volatile("v_rsq_f32 %0, %0" : "+v"(q[i]));
      BODY8(OP)
#undef OP
    } else if (KIND == 23) {  // v_mov_b32
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(q[(i + 1) & 7])); asm volatile("v_mov_b32 %0, %1" : "=v"(q[i]) : "v"(r[(i + 3) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 24) {  // v_bitop3_b32
#define OP(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x3a" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7])); asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x3a" : "+v"(u[i]) : "v"(u[(i + 3) & 7]), "v"(u[(i + 4) & 7]));
      BODY8(OP)
#undef OP
    } else if (KIND == 25) {  // v_med3_f32
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(c1), "v"(c0));
      BODY8(OP)
#undef OP
    } else if (KIND == 26) {  // v_cmp_lt_f32 -> vcc, v_addc_co_u32 (shift a sign into a mask through the carry)
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(q[i]), "v"(c0) : "vcc");
      BODY8(OP)
#undef OP
    } else if (KIND == 27) {  // s_nop 1 between VALU (hazard filler as the compiler emits it)
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n\ts_nop 1" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(c1), "v"(c0));
      BODY8(OP)
#undef OP
    } else if (KIND == 16) {  // v_fma + exec-mask juggling as a divergent `if` compiles to (v_cmp, s_and_saveexec, s_or)
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(q[i]), "v"(c1)); asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_fma_f32 %0, %0, %2, %1\n\ts_or_b64 exec, exec, s[20:21]" : "+v"(q[i]) : "v"(c1), "v"(c0) : "vcc", "s20", "s21");
      BODY8(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  if (stamps && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { stamps[0] = t1 - t0; stamps[1] = w1 - w0; }
  float acc = 0;
  for (int i = 0; i < 8; ++i) acc += r[i] + q[i] + p[i].x + p[i].y + (float)u[i];
  if (acc == 12345.678f) out[0] = acc;
}

template <int KIND>
double run(int waves_per_simd, float* d, int cus, unsigned long long* stamps, double* clock_ghz, double* cyc_real) {
  dim3 grid(cus * 4 * waves_per_simd), block(64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 1.0001f, ITERS, (unsigned long long*)nullptr);  // warm (also settles DVFS)
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, d, 1.0001f, ITERS, stamps);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2]; CHECK(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
  *clock_ghz = (double)h[0] / (double)h[1] * 0.1;
  *cyc_real = (double)h[0] / ((double)waves_per_simd * ITERS * UNROLL);
  // wave-instructions per SIMD
  double instr_per_simd = (double)waves_per_simd * ITERS * UNROLL;
  return ms * 1e-3 * 2.4e9 / instr_per_simd;  // cycles (at 2.4 GHz) per wave-instruction per SIMD
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  float* d; CHECK(hipMalloc(&d, 4));
  unsigned long long* st; CHECK(hipMalloc(&st, 16));
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_subrev_f32(sgpr)", "v_alignbit_b32", "v_mul_lo_u32", "v_sqrt_f32", "v_rcp_f32",
                         "v_lshlrev+v_xor", "v_cmp+v_cndmask", "v_mad_u32_u24", "v_mul_f32", "v_fma_f32(sgpr)", "v_pk_mul/add_f32", "v_fma_f32 dep-chain", "fma+salu 1:1 (per fma)", "fma+salu 2:1 (per fma)", "fma,[cmp,saveexec,fma,or]", "v_pk_fma_f32(sgpr pair,bcast)", "v_mad_u64_u32", "v_and_or_b32", "v_xor_b32", "v_add_u32", "v_rsq_f32", "v_mov_b32", "v_bitop3_b32", "v_med3_f32", "v_cmp+v_addc (pair)", "fma+s_nop 1 (per fma)"};
  printf("%-22s %8s %8s %8s %8s   (cycles @2.4GHz per wave64 instruction per SIMD)\n", "instruction", "1w/SIMD", "2w", "4w", "8w");
#define ROW(K) { printf("%-22s", names[K]); for (int w : {1, 2, 4, 8}) { double g, c; double x = run<K>(w, d, cus, st, &g, &c); printf(" %6.2f (%4.2f real @%4.2fGHz)", x, c, g); } printf("\n"); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27)
  return 0;
}
