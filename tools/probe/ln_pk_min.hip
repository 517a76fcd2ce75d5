// Minimal standalone reproducer (no library, no framework) of DESIGN.md section 7's finding on gfx950 / MI355X:
// a LayerNorm-backward kernel whose per-element float pairs hipcc's SLP vectoriser turned into packed-FP32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) returns rows of dx computed from WRONG row sums when waves of another
// kernel issue MFMAs densely on the same SIMDs.  Same inputs, same kernel, launched again and again: alone every launch is
// bit-identical; beside the MFMA loop most launches differ.  Built with -fno-slp-vectorize: never.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/ln_pk_min.hip -o ln_pk_min && ./ln_pk_min            (fails)
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probe/ln_pk_min.hip -o ln_pk_ok && ./ln_pk_ok   (control)
// Victim variants (bits of V) bisect what the fault needs: 1 no prefetch of the next row, 2 no dgamma / dbeta
// accumulators, 4 operands generated in registers (no global loads in the row loop), 8 D = 512 (one chunk per lane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {   // registers only; 48 KB of LDS: ~3 workgroups per CU
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

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
  v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(w[q] << 16); f[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(u16, (__bf16)lo) | ((unsigned)__builtin_bit_cast(u16, (__bf16)hi) << 16);
}
__device__ __forceinline__ uint4 hash4(unsigned s) {   // four dwords of bf16 pairs in [1, 2)
  uint4 v; unsigned* w = (unsigned*)&v;
#pragma unroll
  for (int q = 0; q < 4; ++q) { s = s * 1664525u + 1013904223u; w[q] = 0x3f803f80u | (s & 0x007f007fu); }
  return v;
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma, xhat = (x - mean) * rstd; one wave per 4 rows
template <int V>
__global__ __launch_bounds__(256) void ln_bwd(const u16* __restrict__ dy, const u16* __restrict__ x, const float* __restrict__ gamma,
                                              const float* __restrict__ mean, const float* __restrict__ rstd, u16* __restrict__ dx,
                                              float* __restrict__ dgb, int rows, int D) {
  constexpr int MAXC = (V & 8) ? 1 : 2;
  const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6), nch = D >> 3;
  const int r0 = wid * 4, r1 = min(rows, r0 + 4);
  float ag[MAXC][8] = {}, ab[MAXC][8] = {}, gam[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) gam[c][q] = lane + 64 * c < nch ? gamma[(lane + 64 * c) * 8 + q] : 0.f;
  uint4 nd[MAXC], nx[MAXC];
  auto fetch = [&](const int row) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (lane + 64 * c < nch && row < r1) {
        if (V & 4) { nd[c] = hash4(row * 977u + lane * 13u + c); nx[c] = hash4(row * 331u + lane * 7u + c + 5u); }
        else { nd[c] = *(const uint4*)(dy + (long long)row * D + (lane + 64 * c) * 8); nx[c] = *(const uint4*)(x + (long long)row * D + (lane + 64 * c) * 8); }
      }
  };
  if (!(V & 1)) fetch(r0);
  for (int row = r0; row < r1; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float g[MAXC][8], xh[MAXC][8], s1 = 0.f, s2 = 0.f;
    uint4 cd[MAXC], cx[MAXC];
    if (V & 1) fetch(row);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { cd[c] = nd[c]; cx[c] = nx[c]; }
    if (!(V & 1)) fetch(row + 1);
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (lane + 64 * c < nch) {
        float d[8], xx[8];
        unpack8(cd[c], d); unpack8(cx[c], xx);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          xh[c][q] = (xx[q] - mu) * rs; g[c][q] = d[q] * gam[c][q];
          s1 += g[c][q]; s2 += g[c][q] * xh[c][q];
          if (!(V & 2)) { ag[c][q] += d[q] * xh[c][q]; ab[c][q] += d[q]; }
        }
      }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (lane + 64 * c < nch) {
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = rs * (g[c][q] - s1 - xh[c][q] * s2);
        *(uint4*)(dx + (long long)row * D + (lane + 64 * c) * 8) = make_uint4(pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7]));
      }
  }
  if (!(V & 2))
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (lane + 64 * c < nch)
#pragma unroll
        for (int q = 0; q < 8; ++q) { atomicAdd(dgb + (lane + 64 * c) * 8 + q, ag[c][q]); atomicAdd(dgb + D + (lane + 64 * c) * 8 + q, ab[c][q]); }
}

__global__ void compare(const uint4* a, const uint4* b, int n16, int* counters) {   // counters: {launches that differ, launches}
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n16; i += blockDim.x)
    if (a[i].x != b[i].x || a[i].y != b[i].y || a[i].z != b[i].z || a[i].w != b[i].w) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(counters, bad); atomicAdd(counters + 1, 1); }
}

static u16 f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }
template <int V> void run(const char* what, hipStream_t s1, hipStream_t s2, const u16* dy, const u16* x, const float* gam, const float* mean,
                          const float* rstd, u16* dx, u16* ref, float* dgb, float* aout, int* cnt) {
  const int rows = 228, D = (V & 8) ? 512 : 768, nblk = ((rows + 3) / 4 + 3) / 4;
  hipLaunchKernelGGL(ln_bwd<V>, dim3(nblk), dim3(256), 0, s1, dy, x, gam, mean, rstd, ref, dgb, rows, D);
  for (int beside = 0; beside < 2; ++beside) {
    CK(hipDeviceSynchronize()); CK(hipMemset(cnt, 0, 8));
    for (int it = 0; it < 100; ++it) {
      if (beside) hipLaunchKernelGGL(mfma_loop, dim3(768), dim3(256), 0, s2, aout, 20000);
      for (int k = 0; k < 8; ++k) {
        hipLaunchKernelGGL(ln_bwd<V>, dim3(nblk), dim3(256), 0, s1, dy, x, gam, mean, rstd, dx, dgb, rows, D);
        hipLaunchKernelGGL(compare, dim3(1), dim3(1024), 0, s1, (const uint4*)dx, (const uint4*)ref, rows * D / 8, cnt);
      }
    }
    CK(hipDeviceSynchronize());
    int h[2]; CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
    printf("%-58s %s: %d of %d launches differ from the first\n", what, beside ? "beside the MFMA loop" : "alone               ", h[0], h[1]); fflush(stdout);
  }
}

int main() {
  const int rows = 228, D = 768;
  u16* hx = (u16*)malloc(rows * D * 2); u16* hdy = (u16*)malloc(rows * D * 2); float hm[228], hr[228], hg[768];
  unsigned s = 1;
  for (int i = 0; i < rows * D; ++i) { s = s * 1664525u + 1013904223u; hx[i] = f2bf((float)(s >> 8) / 8388608.f - 1.f); s = s * 1664525u + 1013904223u; hdy[i] = f2bf(1e-4f * ((float)(s >> 8) / 8388608.f - 1.f)); }
  for (int r = 0; r < rows; ++r) { hm[r] = 0.01f * (r % 7); hr[r] = 1.7f + 0.001f * r; }
  for (int c = 0; c < D; ++c) hg[c] = 1.f;
  u16 *x, *dy, *dx, *ref; float *gam, *mean, *rstd, *dgb, *aout; int* cnt; hipStream_t s1, s2;
  CK(hipMalloc(&x, rows * D * 2)); CK(hipMalloc(&dy, rows * D * 2)); CK(hipMalloc(&dx, rows * D * 2)); CK(hipMalloc(&ref, rows * D * 2));
  CK(hipMalloc(&gam, D * 4)); CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4)); CK(hipMalloc(&dgb, 2 * D * 4)); CK(hipMalloc(&aout, 768 * 256 * 4)); CK(hipMalloc(&cnt, 8));
  CK(hipMemcpy(x, hx, rows * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy, rows * D * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(gam, hg, D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mean, hm, rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rstd, hr, rows * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dgb, 0, 2 * D * 4)); CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
#define RUN(V, what) run<V>(what, s1, s2, dy, x, gam, mean, rstd, dx, ref, dgb, aout, cnt)
  RUN(0, "as in the library (round 3)");
  RUN(1, "no prefetch of the next row");
  RUN(2, "no dgamma / dbeta accumulators");
  RUN(3, "neither");
  RUN(4, "operands generated in registers (no loads in the loop)");
  RUN(6, "... and no dgamma / dbeta accumulators");
  RUN(8, "D = 512: one 8-element chunk per lane");
  RUN(10, "D = 512, no dgamma / dbeta accumulators");
  return 0;
}
