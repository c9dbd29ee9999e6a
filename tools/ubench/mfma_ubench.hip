// Does FP32 MFMA overlap with VALU work on gfx950? Times (a) v_mfma_f32_16x16x4_f32 alone, (b) v_pk_fma_f32
// alone, (c) both interleaved, at 4/6/8 waves per SIMD. If (c) ~ max(a, b) the matrix pipe is a second
// issue stream for the hit loop's dot products; if (c) ~ a + b it is not.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_ubench mfma_ubench.hip && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 16384;

template <int MF, int PK, int KIND>  // per loop iteration: MF mfma and PK VALU ops (KIND 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_xor_b32), interleaved
__global__ __launch_bounds__(64) void k(float* out, float s0, int n) {
  f4 c[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
  f2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f2{threadIdx.x * 0.001f + i, i + 0.5f};
  const float a = threadIdx.x * 0.01f, b = s0;
  const f2 q = {s0, s0 * 0.5f};
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MF > u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < PK / 4; ++v) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(u * (PK / 4) + v) & 7]) : "v"(q));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(p[(u * (PK / 4) + v) & 7].x) : "v"(q.x));
        if (KIND == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(p[(u * (PK / 4) + v) & 7].x) : "v"(q.x));
      }
    }
  }
  float acc = 0;
  for (int i = 0; i < 4; ++i) acc += c[i].x + c[i].y + c[i].z + c[i].w;
  for (int i = 0; i < 8; ++i) acc += p[i].x + p[i].y;
  if (acc == 12345.678f) out[0] = acc;
}

template <int MF, int PK, int KIND>
double run(int wps, float* d, int cus) {
  dim3 grid(cus * 4 * wps), block(64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MF, PK, KIND>), grid, block, 0, 0, d, 1.0001f, ITERS);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MF, PK, KIND>), grid, block, 0, 0, d, 1.0001f, ITERS);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3 * 2.4e9 / ((double)wps * ITERS);  // 2.4 GHz cycles per loop iteration per SIMD-wave slot
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  float* d; CHECK(hipMalloc(&d, 4));
  printf("%s: cycles @2.4GHz per loop iteration per wave and SIMD (iteration = 4 mfma_16x16x4_f32 and/or N v_pk_fma_f32)\n", prop.gcnArchName);
  printf("%-34s %8s %8s %8s\n", "mix", "4w/SIMD", "6w", "8w");
#define ROW(name, MF, PK) ROWK(name, MF, PK, 0)
#define ROWK(name, MF, PK, KIND) { printf("%-34s", name); for (int w : {4, 6, 8}) printf(" %8.2f", run<MF, PK, KIND>(w, d, prop.multiProcessorCount)); printf("\n"); }
  ROW("4 mfma", 4, 0)
  ROW("16 pk_fma", 0, 16)
  ROW("32 pk_fma", 0, 32)
  ROW("4 mfma + 16 pk_fma", 4, 16)
  ROW("4 mfma + 32 pk_fma", 4, 32)
  ROW("2 mfma + 32 pk_fma", 2, 32)
  ROWK("32 v_fma", 0, 32, 1)
  ROWK("4 mfma + 32 v_fma", 4, 32, 1)
  ROWK("32 v_xor", 0, 32, 2)
  ROWK("4 mfma + 32 v_xor", 4, 32, 2)
  ROWK("1 mfma + 32 v_fma", 1, 32, 1)
  return 0;
}
