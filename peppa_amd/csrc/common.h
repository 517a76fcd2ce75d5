// Shared device helpers for the gfx950 (MI355X) kernels of libpeppa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/peppa_hip.h"

// The 16-bit operand / activation type of this build.  libpeppa_hip.so is built with bf16 (BASELINE configs[1]);
// the same sources built with -DPP_F16 give libpeppa_hip_f16.so, IEEE half as in the reference's `precision: 16` AMP runs
// (hparams_base.yaml:45; BASELINE configs[4]).  Both accumulate in fp32 on the matrix cores; statistics, norms, softmax,
// the heads, the loss and the optimizer are fp32 in either.  (Entry points keep "bf16" in their names in both builds:
// there it means "the library's 16-bit type"; pp_dtype() tells which.)
#ifdef PP_F16
typedef _Float16 h16;
#define PP_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 h16;
#define PP_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short h16raw;  // raw bits of the 16-bit type in HBM / LDS

#define PP_WAVE 64

// ---- error plumbing (never throws, never syncs) -------------------------------------
void pp_set_error(const char* fmt, ...);
#define PP_CHECK_ARG(cond, ...)                        \
  do {                                                 \
    if (!(cond)) {                                     \
      pp_set_error(__VA_ARGS__);                       \
      return PP_ERR_INVALID;                           \
    }                                                  \
  } while (0)
#define PP_LAUNCH_CHECK()                                                  \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      pp_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,       \
                   hipGetErrorString(e_));                                 \
      return PP_ERR_HIP;                                                   \
    }                                                                      \
  } while (0)

// ---- 16-bit <-> f32 ---------------------------------------------------------------------
#ifdef PP_F16
__device__ __forceinline__ float h2f(h16raw v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ h16raw f2h(float f) {
  _Float16 b = (_Float16)f;            // v_cvt_f16_f32: RNE, overflow -> inf (the loss scaler's job to avoid)
  return __builtin_bit_cast(h16raw, b);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const half2_t h = __builtin_bit_cast(half2_t, w[q]);
    f[2 * q] = (float)h[0];
    f[2 * q + 1] = (float)h[1];
  }
}
#else
__device__ __forceinline__ float h2f(h16raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ h16raw f2h(float f) {
  // plain cast lowers to v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(h16raw, b);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
#endif
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]);
  v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]);
  return v;
}
// gfx950 transposing LDS read (ds_read_b64_tr_b16) through the type-agnostic i16 builtin
__device__ __forceinline__ h16x4 ds_read_tr16(const unsigned char* lds_addr) {
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
  return __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)lds_addr));
}

// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through Abramowitz & Stegun 7.1.26 (erfc(|u|) = poly(t) e^{-u^2}, absolute error
// <= 1.5e-7: two orders below the bf16 rounding of anything it feeds) from e = exp(-u^2), which the GELU derivative needs
// anyway as its Gaussian factor: one v_exp, one v_rcp and a handful of FMAs instead of libm's erff, in kernels that are
// VALU-bound on it (the wav2vec2 conv0 passes, the GELU epilogues).  The lower tail is formed without cancellation.
__device__ __forceinline__ float gauss_cdf_from_exp(float u, float e) {
  const float t = __frcp_rn(1.0f + 0.3275911f * fabsf(u));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float h = 0.5f * poly * e;                      // = 0.5 erfc(|u|)
  return u < 0.f ? h : 1.0f - h;
}
__device__ __forceinline__ float gelu_f(float x) {
  const float u = x * 0.70710678118654752f;
  return x * gauss_cdf_from_exp(u, __expf(-u * u));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float u = x * 0.70710678118654752f;
  const float e = __expf(-u * u);                       // = exp(-x^2 / 2)
  return gauss_cdf_from_exp(u, e) + x * 0.3989422804014327f * e;
}

// ---- counter-based dropout mask (pp_dropout_*, the GEMM epilogue, pp_gelu_bwd_dropout) ------------------------------
// element i of the flat tensor is kept iff hash16(seed, i) >= p * 65536; chunk = i / 8
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void keep8(uint32_t seed, long long chunk, uint32_t thr, bool* keep) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = mix32((uint32_t)(chunk * 4 + q) * 0x9E3779B9u + seed + (uint32_t)((chunk * 4 + q) >> 32) * 0x7F4A7C15u);
    keep[2 * q] = (h & 0xffffu) >= thr;
    keep[2 * q + 1] = (h >> 16) >= thr;
  }
}

// ---- wave / block reductions ----------------------------------------------------------
// sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane, added in the order
// (r0 + r1) + (r2 + r3) like `v += shfl_xor(v, 16); v += shfl_xor(v, 32)`.  gfx950's v_permlane16_swap / 32_swap do it
// on the VALU; __shfl_xor goes through ds_bpermute (an LDS round trip per step).
__device__ __forceinline__ float sum_rows4(float v) {
  unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // {r0,r0,r2,r2}, {r1,r1,r3,r3}
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lo,lo}, {hi,hi}
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// full-wave reductions on the VALU: DPP inside a 16-lane row (quad swaps, half-row mirror, row mirror), then the two
// permlane swaps across rows; every lane gets the result.  (__shfl_xor = ds_bpermute: six LDS round trips.)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f<0x141>(v);    // row_half_mirror: the other quad of the 8-lane half row
  v += dpp_f<0x140>(v);    // row_mirror: the other half of the 16-lane row
  return sum_rows4(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over a block of NW waves; `red` needs NW floats of LDS; result valid in all threads
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

// ---- exact unsigned division by an invariant (n < 2^31), host-prepared ------------------
struct FastDiv {
  uint32_t mul;
  uint32_t shift;
  uint32_t div;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.div = d;
  if (d == 1) { f.mul = 0; f.shift = 0; return f; }
  uint32_t s = 0;
  while ((1u << s) < d) ++s;  // ceil(log2 d)
  uint64_t m = ((1ull << (32 + s)) + d - 1) / d;  // fits in 33 bits; valid for n < 2^31 when truncated? use 64-bit mul
  f.mul = (uint32_t)(m - (1ull << 32));            // store low 32 bits of (m - 2^32)
  f.shift = s;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  if (f.div == 1) return n;
  // q = (n*m) >> (32+s) with m = 2^32 + mul  ->  ((mulhi(n,mul) + n) >> s), overflow-safe form
  const uint32_t t = __umulhi(n, f.mul);
  return (t + ((n - t) >> 1)) >> (f.shift - 1);
}
