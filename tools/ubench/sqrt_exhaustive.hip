// Is v_sqrt_f32 of gfx950 correctly rounded? Exhaustive over every non-negative finite float: compares the raw instruction with
// the IEEE result (the compiler's correctly rounded expansion, __builtin_sqrtf with -fno-fast-math) and counts the inputs where
// they differ. (rt_kernels.hip's sqrt_rn pays ~9 instructions per call to pick the correctly rounded neighbour.)
//   hipcc --offload-arch=gfx950 -O3 -o sqrt_exhaustive sqrt_exhaustive.hip && ./sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k(unsigned long long* out, unsigned* examples) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long diff = 0, diff_norm = 0, diff_rcp = 0, lo1 = 0, hi1 = 0, far = 0, rlo = 0, rhi = 0, rfar = 0, bad_form = 0, checked_form = 0;
  for (unsigned long long i = tid; i < 0x7f800000ull; i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (unsigned)i);
    const float hw = __builtin_amdgcn_sqrtf(x);
    const float rn = __builtin_sqrtf(x);
    if (__builtin_bit_cast(unsigned, hw) != __builtin_bit_cast(unsigned, rn)) {
      diff++;
      const long long dl = (long long)__builtin_bit_cast(unsigned, hw) - (long long)__builtin_bit_cast(unsigned, rn);
      if (dl == -1) lo1++; else if (dl == 1) hi1++; else far++;
      if (x >= 0x1p-96f) {
        if (diff_norm == 0) { const unsigned slot = atomicAdd(&examples[0], 1u); if (slot < 15u) examples[1 + slot] = (unsigned)i; }
        diff_norm++;
      }
    }
    // the sign form of rt_kernels.hip's sqrt_rn (-DRT_SQRT_SIGNS=1), on its domain [2^-96, +inf)
    if ((unsigned)i - 0x0f800000u < 0x70000000u) {
      const unsigned dnb = __builtin_bit_cast(unsigned, hw) - 1u;
      const float nrd = __builtin_fmaf(__builtin_bit_cast(float, dnb), hw, -x), nru = __builtin_fmaf(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, hw) + 1u), hw, -x);
      const unsigned r = dnb + (__builtin_bit_cast(unsigned, nrd) >> 31) + (__builtin_bit_cast(unsigned, nru) >> 31);
      if (r != __builtin_bit_cast(unsigned, rn)) bad_form++;
      checked_form++;
    }
    // v_rcp_f32 against the correctly rounded 1 / x (normal range only)
    if (x >= 0x1p-100f && x <= 0x1p100f) {
      const float hr = __builtin_amdgcn_rcpf(x), rr = 1.0f / x;
      if (__builtin_bit_cast(unsigned, hr) != __builtin_bit_cast(unsigned, rr)) {
        diff_rcp++;
        const long long dl = (long long)__builtin_bit_cast(unsigned, hr) - (long long)__builtin_bit_cast(unsigned, rr);
        if (dl == -1) rlo++; else if (dl == 1) rhi++; else rfar++;
      }
    }
  }
  atomicAdd(&out[0], diff); atomicAdd(&out[1], diff_norm); atomicAdd(&out[2], diff_rcp);
  atomicAdd(&out[3], lo1); atomicAdd(&out[4], hi1); atomicAdd(&out[5], far); atomicAdd(&out[6], rlo); atomicAdd(&out[7], rhi); atomicAdd(&out[8], rfar); atomicAdd(&out[9], bad_form); atomicAdd(&out[10], checked_form);
}

int main() {
  unsigned long long* d; unsigned* ex;
  CHECK(hipMalloc(&d, 88)); CHECK(hipMalloc(&ex, 64));
  CHECK(hipMemset(d, 0, 88)); CHECK(hipMemset(ex, 0, 64));
  hipLaunchKernelGGL(k, dim3(256 * 32), dim3(256), 0, 0, d, ex);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[11]; unsigned he[16];
  CHECK(hipMemcpy(h, d, 88, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(he, ex, 64, hipMemcpyDeviceToHost));
  printf("v_sqrt_f32 vs correctly rounded sqrt over all %llu non-negative finite floats: %llu differ (%llu of them at x >= 2^-96)\n", 0x7f800000ull, h[0], h[1]);
  printf("  v_sqrt_f32: one ulp below %llu, one ulp above %llu, further %llu\n", h[3], h[4], h[5]);
  printf("  v_rcp_f32: one ulp below %llu, one ulp above %llu, further %llu\n", h[6], h[7], h[8]);
  printf("sign form of sqrt_rn (dn + sign(dn s - x) + sign(up s - x)) vs correctly rounded sqrt on [2^-96, inf): %llu inputs, %llu differ\n", h[10], h[9]);
  printf("v_rcp_f32 vs correctly rounded 1/x over x in [2^-100, 2^100]: %llu differ\n", h[2]);
  for (unsigned i = 0; i < (he[0] < 15u ? he[0] : 15u); ++i) printf("  example input bits 0x%08x\n", he[1 + i]);
  return 0;
}
