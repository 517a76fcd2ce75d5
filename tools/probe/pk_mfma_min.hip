// Minimal reproducer: on gfx950 (MI355X) a packed-FP32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
// can return a wrong result when waves of ANOTHER kernel issue MFMAs densely on the same SIMD.  Each victim kernel contains
// exactly one packed form (inline asm; everything else is scalar, built with -fno-slp-vectorize) and checks every result
// against the scalar instruction sequence in the same lane.  No memory is read in the checked path.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probe/pk_mfma_min.hip -o pk_mfma_min && ./pk_mfma_min
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void mfma_aggressor(float* out, int iters) {   // registers only; ~3 workgroups per CU
  __shared__ uint4 pad[48 * 1024 / 16];
  pad[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
  __syncthreads();
  const uint4 a = pad[(threadIdx.x * 7) & 255], b = pad[(threadIdx.x * 13) & 255];
  f32x4 acc[4] = {};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

__device__ __forceinline__ float rnd12(unsigned& s) {   // a float in [1, 2)
  s = s * 1664525u + 1013904223u;
  return __uint_as_float(0x3f800000u | (s >> 9));
}
template <int FORM> __device__ __forceinline__ v2f packed(v2f a, v2f b, v2f c) {
  v2f d = c;
  if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
  if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(d) : "v"(a), "v"(b));
  if (FORM == 6) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  if (FORM == 7) asm volatile("v_mov_b64 %0, %1" : "=v"(d) : "v"(a));
  return d;
}
template <int FORM> __device__ __forceinline__ v2f scalar(v2f a, v2f b, v2f c) {
#pragma clang fp contract(off)
  v2f d;
  if (FORM == 0) { d.x = a.x * b.x; d.y = a.y * b.y; }
  if (FORM == 1) { d.x = a.x * b.x; d.y = a.x * b.y; }
  if (FORM == 2) { d.x = a.x - b.x; d.y = a.y - b.x; }
  if (FORM == 3) { d.x = a.y * b.x; d.y = a.x * b.y; }
  if (FORM == 4) { d.x = __builtin_fmaf(a.x, b.x, c.x); d.y = __builtin_fmaf(a.y, b.y, c.y); }
  if (FORM == 5) { d.x = __builtin_fmaf(-a.x, b.x, c.x); d.y = __builtin_fmaf(-a.x, b.y, c.y); }
  if (FORM == 6) { d.x = a.x + b.x; d.y = a.y + b.y; }
  if (FORM == 7) d = a;
  return d;
}
// result[0] = wrong results, result[1] = checked results, result[2..9] = one example (lane, a, b, c, got, want as bits)
template <int FORM> __global__ __launch_bounds__(256) void victim(unsigned* result, int loops) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, wrong = 0;
  for (int it = 0; it < loops; ++it) {
    v2f c = {rnd12(s), rnd12(s)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {           // a dependent chain: every result feeds the next operation
      const v2f a = {rnd12(s), rnd12(s)}, b = {c.y - 0.5f, rnd12(s)};
      const v2f got = packed<FORM>(a, b, c), want = scalar<FORM>(a, b, c);
      if (__float_as_uint(got.x) != __float_as_uint(want.x) || __float_as_uint(got.y) != __float_as_uint(want.y)) {
        if (!wrong && atomicCAS(result + 2, 0u, 1u + (threadIdx.x & 63)) == 0u) {
          const float e[8] = {a.x, a.y, b.x, b.y, got.x, got.y, want.x, want.y};
          for (int q = 0; q < 8; ++q) result[3 + q] = __float_as_uint(e[q]);
        }
        ++wrong;
      }
      c.x = __builtin_fmaf(want.x, 0.25f, 1.0f);   // continue from the correct value, kept in [1, 2.x); scalar on purpose:
      c.y = __builtin_fmaf(want.y, 0.25f, 1.0f);   // v2f arithmetic would itself compile to v_pk_fma_f32
    }
  }
  if (wrong) atomicAdd(result, wrong);
  if (threadIdx.x == 0) atomicAdd(result + 1, 256u * 8u * (unsigned)loops);
}

template <int FORM> void run(const char* name, hipStream_t s1, hipStream_t s2, unsigned* res, float* aout, bool aggress) {
  CK(hipMemsetAsync(res, 0, 64, s1));
  CK(hipDeviceSynchronize());
  for (int it = 0; it < 20; ++it) {
    if (aggress) hipLaunchKernelGGL(mfma_aggressor, dim3(768), dim3(256), 0, s2, aout, 20000);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(victim<FORM>, dim3(1024), dim3(256), 0, s1, res, 500);
  }
  CK(hipDeviceSynchronize());
  unsigned h[16]; CK(hipMemcpy(h, res, 64, hipMemcpyDeviceToHost));
  printf("%-62s %s: %u wrong of %u results", name, aggress ? "beside MFMA waves" : "alone            ", h[0], h[1]);
  if (h[0]) {
    float f[8]; memcpy(f, h + 3, 32);
    printf("   e.g. lane %u: a = (%.9g, %.9g) b = (%.9g, %.9g) got (%.9g, %.9g) [%08x %08x] want (%.9g, %.9g) [%08x %08x]",
           h[2] - 1, f[0], f[1], f[2], f[3], f[4], f[5], h[7], h[8], f[6], f[7], h[9], h[10]);
  }
  printf("\n"); fflush(stdout);
}

int main() {
  unsigned* res; float* aout; hipStream_t s1, s2;
  CK(hipMalloc(&res, 64)); CK(hipMalloc(&aout, 768 * 256 * 4)); CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (int aggress = 0; aggress < 2; ++aggress) {
    run<0>("v_pk_mul_f32 d, a, b", s1, s2, res, aout, aggress);
    run<1>("v_pk_mul_f32 d, a, b op_sel_hi:[0,1]", s1, s2, res, aout, aggress);
    run<2>("v_pk_add_f32 d, a, b op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]", s1, s2, res, aout, aggress);
    run<3>("v_pk_mul_f32 d, a, b op_sel:[1,0] op_sel_hi:[0,1]", s1, s2, res, aout, aggress);
    run<4>("v_pk_fma_f32 d, a, b, d", s1, s2, res, aout, aggress);
    run<5>("v_pk_fma_f32 d, a, b, d op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]", s1, s2, res, aout, aggress);
    run<6>("v_pk_add_f32 d, a, b", s1, s2, res, aout, aggress);
    run<7>("v_mov_b64 d, a", s1, s2, res, aout, aggress);
  }
  return 0;
}
