// LayerNorm, attention softmax, wav2vec2 layer-0 conv + GroupNorm, positional-conv weight-norm
// (gfx950).  All HBM/latency-bound; one wave per row for the row-wise ops, wave-64 reductions.
// Replaces torch.nn.LayerNorm / softmax / GroupNorm / weight_norm inside torchaudio 0.9.1
// wav2vec2_base as reached from pig/models.py:101-105.
#include "common.h"

namespace {

constexpr int LN_MAXC = 2;  // chunks (of 8) per lane -> D <= 1024
constexpr int LN_BWD_ALONE_LDS = 92 * 1024;   // + the kernel's own 32 KB = 124 KB of a CU's 160

// experiment switches for tools/probe/ln_variants.sh (run-to-run differences of ln_bwd_kernel inside the step):
// 1 wave sums through __shfl_xor, 2 no prefetch of the next row, 4 multiply by 1 / D instead of dividing, 8 full vmcnt wait
// after the row's loads, 16 no LDS (no dgamma / dbeta)
#ifndef PP_LN_VARIANT
#define PP_LN_VARIANT 0
#endif
}
extern const int pp_exp_ln_variant = PP_LN_VARIANT;   // reported by pp_experimental_build()
namespace {
__device__ __forceinline__ float ln_wave_sum(float v) {
#if PP_LN_VARIANT & 1
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
  return v;
#else
  return wave_sum(v);
#endif
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const h16raw* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, h16raw* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = D >> 3;
  float f[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      unpack8(*(const uint4*)(x + (long long)row * D + ch * 8), f[c]);
#pragma unroll
      for (int q = 0; q < 8; ++q) s += f[c][q];
    }
  }
  const float mu = wave_sum(s) / D;
  float v = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
    if (lane + 64 * c < nch)
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float d = f[c][q] - mu; v += d * d; }
  const float rs = rsqrtf(wave_sum(v) / D + eps);
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float o[8];
      const float4 g0 = *(const float4*)(gamma + ch * 8), g1 = *(const float4*)(gamma + ch * 8 + 4);
      const float4 b0 = *(const float4*)(beta + ch * 8), b1 = *(const float4*)(beta + ch * 8 + 4);
      const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (f[c][q] - mu) * rs * gm[q] + bt[q];
      *(uint4*)(y + (long long)row * D + ch * 8) = pack8(o);
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const h16raw* __restrict__ dy, const h16raw* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, h16raw* __restrict__ dx,
                                                     float* dgamma, float* dbeta, int rows, int D, int rows_per_wave,
                                                     float* __restrict__ ws) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nch = D >> 3;
  float ag[LN_MAXC][8], ab[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) ag[c][q] = ab[c][q] = 0.f;
  const int r0 = wid * rows_per_wave;
  const int r1 = min(rows, r0 + rows_per_wave);
  float gam[LN_MAXC][8];                     // this lane's gamma, once for all its rows
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int q = 0; q < 8; ++q) gam[c][q] = ch < nch ? gamma[ch * 8 + q] : 0.f;
  }
  // the next row's operands are requested before this row's two wave reductions (a row is one dependent chain)
  uint4 nd[LN_MAXC], nx[LN_MAXC];
  auto fetch = [&](const int row) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch && row < r1) {
        nd[c] = *(const uint4*)(dy + (long long)row * D + ch * 8);
        nx[c] = *(const uint4*)(x + (long long)row * D + ch * 8);
      }
    }
  };
  fetch(r0);
  for (int row = r0; row < r1; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXC][8], xh[LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
    uint4 cd[LN_MAXC], cx[LN_MAXC];
#if PP_LN_VARIANT & 2
    fetch(row);
#endif
#if PP_LN_VARIANT & 8
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) { cd[c] = nd[c]; cx[c] = nx[c]; }
#if !(PP_LN_VARIANT & 2)
    fetch(row + 1);
#endif
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float d[8], xx[8];
        unpack8(cd[c], d);
        unpack8(cx[c], xx);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          xh[c][q] = (xx[q] - mu) * rs;
          g[c][q] = d[q] * gam[c][q];
          s1 += g[c][q];
          s2 += g[c][q] * xh[c][q];
          ag[c][q] += d[q] * xh[c][q];
          ab[c][q] += d[q];
        }
      }
    }
#if PP_LN_VARIANT & 4
    const float inv_d = 1.f / (float)D;
    s1 = ln_wave_sum(s1) * inv_d;
    s2 = ln_wave_sum(s2) * inv_d;
#else
    s1 = ln_wave_sum(s1) / D;
    s2 = ln_wave_sum(s2) / D;
#endif
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = rs * (g[c][q] - s1 - xh[c][q] * s2);
        *(uint4*)(dx + (long long)row * D + ch * 8) = pack8(o);
      }
    }
  }
#if !(PP_LN_VARIANT & 16)   // (16: no LDS at all -- dgamma / dbeta are then not produced)
  // combine the block's 4 waves in LDS, then one atomic per channel per block
  __shared__ float red[2][4][64 * 8 * LN_MAXC];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        red[0][w][ch * 8 + q] = ag[c][q];
        red[1][w][ch * 8 + q] = ab[c][q];
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    const float sg = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    const float sb = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
    if (ws) {   // per-block partials, summed by ln_bwd_partials_kernel: no same-address atomics, deterministic
      ws[((long long)blockIdx.x * 2 + 0) * D + i] = sg;
      ws[((long long)blockIdx.x * 2 + 1) * D + i] = sb;
    } else {
      atomicAdd(dgamma + i, sg);
      atomicAdd(dbeta + i, sb);
    }
  }
#endif
}

__global__ __launch_bounds__(256) void ln_bwd_partials_kernel(const float* __restrict__ ws, int nblk, int D,
                                                              float* dgamma, float* dbeta) {
  // 16 columns x 16 slices of the partial rows per workgroup (columns run over 2 * D: dgamma then dbeta)
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (i < 2 * D) {
    const int which = i >= D, c = i - which * D;
    for (int b = part; b < nblk; b += 16) s += ws[((long long)b * 2 + which) * D + c];
  }
  red[part][cl] = s;
  __syncthreads();
  if (part == 0 && i < 2 * D) {
    for (int q = 1; q < 16; ++q) s += red[q][cl];
    const int which = i >= D, c = i - which * D;
    (which ? dbeta : dgamma)[c] += s;
  }
}

constexpr int SM_MAXC = 16;  // columns per lane, template SMC = 4 / 8 / 16 -> T <= 256 / 512 / 1024 (the unfused attention path:
                            // clips beyond the fused kernel's 320 frames, e.g. > 2.3 s of 44.1-kHz audio fed unresampled)

template <int SMC>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, int lds, h16raw* __restrict__ P, int ldp,
                                                          long long rows, int T, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[SMC];
  float mx = -3.0e38f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int j = lane + 64 * c;
    v[c] = j < T ? S[row * lds + j] * scale : -3.0e38f;
    mx = fmaxf(mx, v[c]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int j = lane + 64 * c;
    v[c] = j < T ? __expf(v[c] - mx) : 0.f;
    sum += v[c];
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int j = lane + 64 * c;
    if (j < ldp) P[row * ldp + j] = f2h(v[c] * inv);
  }
}

template <int SMC>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, int lds, const h16raw* __restrict__ P,
                                                          int ldp, h16raw* __restrict__ dS, long long rows, int T, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float p[SMC], d[SMC];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int j = lane + 64 * c;
    p[c] = j < T ? h2f(P[row * ldp + j]) : 0.f;
    d[c] = j < T ? dP[row * lds + j] : 0.f;
    dot += p[c] * d[c];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < SMC; ++c) {
    const int j = lane + 64 * c;
    if (j < ldp) dS[row * ldp + j] = f2h(scale * p[c] * (d[c] - dot));
  }
}

// ---- wav2vec2 feature-extractor layer 0 ---------------------------------------------------------
constexpr int C0 = 512, K0 = 10, S0 = 5, TT0 = 64;  // channels, kernel, stride, frames per block

template <class F>
__device__ __forceinline__ void conv0_tile(const float* __restrict__ wave, int L, int T0, const float* __restrict__ w,
                                           float* xs, F f) {
  // grid: (tiles-per-block groups, B); each block walks tiles blockIdx.x, +gridDim.x, ... so that the
  // per-thread reductions of the callers end in few atomics; thread handles channels 2*tid, 2*tid+1
  const int b = blockIdx.y;
  const int c = threadIdx.x * 2;
  float w0[K0], w1[K0];
#pragma unroll
  for (int k = 0; k < K0; ++k) { w0[k] = w[c * K0 + k]; w1[k] = w[(c + 1) * K0 + k]; }
  const int ntiles = (T0 + TT0 - 1) / TT0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int t0 = tile * TT0;
    const int nt = min(TT0, T0 - t0);
    const int nx = (nt - 1) * S0 + K0;
    __syncthreads();
    for (int i = threadIdx.x; i < nx; i += 256) xs[i] = wave[(long long)b * L + (long long)t0 * S0 + i];
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      float y0 = 0.f, y1 = 0.f;
#pragma unroll
      for (int k = 0; k < K0; ++k) {
        const float xv = xs[t * S0 + k];
        y0 += w0[k] * xv;
        y1 += w1[k] * xv;
      }
      f(b, t0 + t, t, c, y0, y1);
    }
  }
}

__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* wave, int L, int T0, const float* w, float* stats) {
  __shared__ float xs[(TT0 - 1) * S0 + K0];
  float a0 = 0.f, q0 = 0.f, a1 = 0.f, q1 = 0.f;
  conv0_tile(wave, L, T0, w, xs, [&](int, int, int, int, float y0, float y1) {
    a0 += y0; q0 += y0 * y0; a1 += y1; q1 += y1 * y1;
  });
  const int c = threadIdx.x * 2;
  float* st = stats + ((long long)blockIdx.y * C0 + c) * 2;
  atomicAdd(st + 0, a0); atomicAdd(st + 1, q0); atomicAdd(st + 2, a1); atomicAdd(st + 3, q1);
}

struct GN0 {
  float mu0, rs0, mu1, rs1;
};
__device__ __forceinline__ GN0 gn0_load(const float* stats, int b, int c, int T0, float eps) {
  const float* st = stats + ((long long)b * C0 + c) * 2;
  GN0 g;
  g.mu0 = st[0] / T0; g.rs0 = rsqrtf(fmaxf(st[1] / T0 - g.mu0 * g.mu0, 0.f) + eps);
  g.mu1 = st[2] / T0; g.rs1 = rsqrtf(fmaxf(st[3] / T0 - g.mu1 * g.mu1, 0.f) + eps);
  return g;
}

__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* wave, int L, int T0, const float* w, const float* stats,
                                                          const float* gamma, const float* beta, float eps, h16raw* out) {
  __shared__ float xs[(TT0 - 1) * S0 + K0];
  const int c = threadIdx.x * 2;
  const GN0 g = gn0_load(stats, blockIdx.y, c, T0, eps);
  const float g0 = gamma[c], g1 = gamma[c + 1], b0 = beta[c], b1 = beta[c + 1];
  conv0_tile(wave, L, T0, w, xs, [&](int b, int tg, int, int cc, float y0, float y1) {
    const float u0 = (y0 - g.mu0) * g.rs0 * g0 + b0, u1 = (y1 - g.mu1) * g.rs1 * g1 + b1;
    *(uint32_t*)(out + ((long long)b * T0 + tg) * C0 + cc) = pack2(gelu_f(u0), gelu_f(u1));
  });
}

__global__ __launch_bounds__(256) void conv0_bwd_reduce_kernel(const float* wave, int L, int T0, const float* w,
                                                               const float* stats, const float* gamma, const float* beta,
                                                               float eps, const h16raw* dout, float* red) {
  __shared__ float xs[(TT0 - 1) * S0 + K0];
  const int c = threadIdx.x * 2;
  const GN0 g = gn0_load(stats, blockIdx.y, c, T0, eps);
  const float g0 = gamma[c], g1 = gamma[c + 1], b0 = beta[c], b1 = beta[c + 1];
  float r00 = 0.f, r01 = 0.f, r10 = 0.f, r11 = 0.f;
  conv0_tile(wave, L, T0, w, xs, [&](int b, int tg, int, int cc, float y0, float y1) {
    const float xh0 = (y0 - g.mu0) * g.rs0, xh1 = (y1 - g.mu1) * g.rs1;
    const uint32_t dv = *(const uint32_t*)(dout + ((long long)b * T0 + tg) * C0 + cc);
    const float du0 = h2f((h16raw)(dv & 0xffff)) * gelu_grad_f(xh0 * g0 + b0);
    const float du1 = h2f((h16raw)(dv >> 16)) * gelu_grad_f(xh1 * g1 + b1);
    r00 += du0; r01 += du0 * xh0; r10 += du1; r11 += du1 * xh1;
  });
  float* rp = red + ((long long)blockIdx.y * C0 + c) * 2;
  atomicAdd(rp + 0, r00); atomicAdd(rp + 1, r01); atomicAdd(rp + 2, r10); atomicAdd(rp + 3, r11);
}

__global__ __launch_bounds__(256) void conv0_bwd_apply_kernel(const float* wave, int L, int T0, const float* w,
                                                              const float* stats, const float* gamma, const float* beta,
                                                              float eps, const h16raw* dout, const float* red, float* dw,
                                                              float* dgamma, float* dbeta, float* ws) {
  __shared__ float xs[(TT0 - 1) * S0 + K0];
  const int c = threadIdx.x * 2;
  const GN0 g = gn0_load(stats, blockIdx.y, c, T0, eps);
  const float g0 = gamma[c], g1 = gamma[c + 1], b0 = beta[c], b1 = beta[c + 1];
  const float* rp = red + ((long long)blockIdx.y * C0 + c) * 2;
  const float m00 = rp[0] / T0, m01 = rp[1] / T0, m10 = rp[2] / T0, m11 = rp[3] / T0;
  float dw0[K0], dw1[K0];
#pragma unroll
  for (int k = 0; k < K0; ++k) dw0[k] = dw1[k] = 0.f;
  conv0_tile(wave, L, T0, w, xs, [&](int b, int tg, int tl, int cc, float y0, float y1) {
    const float xh0 = (y0 - g.mu0) * g.rs0, xh1 = (y1 - g.mu1) * g.rs1;
    const uint32_t dv = *(const uint32_t*)(dout + ((long long)b * T0 + tg) * C0 + cc);
    const float du0 = h2f((h16raw)(dv & 0xffff)) * gelu_grad_f(xh0 * g0 + b0);
    const float du1 = h2f((h16raw)(dv >> 16)) * gelu_grad_f(xh1 * g1 + b1);
    const float dy0 = g.rs0 * g0 * (du0 - m00 - xh0 * m01), dy1 = g.rs1 * g1 * (du1 - m10 - xh1 * m11);
#pragma unroll
    for (int k = 0; k < K0; ++k) {
      const float xv = xs[tl * S0 + k];
      dw0[k] += dy0 * xv;
      dw1[k] += dy1 * xv;
    }
  });
  if (ws) {     // deterministic: this clip's partial row (gridDim.x == 1); conv0_bwd_sum_kernel adds the clips in order
    float* wp = ws + (long long)blockIdx.y * (C0 * K0);
#pragma unroll
    for (int k = 0; k < K0; ++k) { wp[c * K0 + k] = dw0[k]; wp[(c + 1) * K0 + k] = dw1[k]; }
    return;
  }
#pragma unroll
  for (int k = 0; k < K0; ++k) { atomicAdd(dw + c * K0 + k, dw0[k]); atomicAdd(dw + (c + 1) * K0 + k, dw1[k]); }
  if (blockIdx.x == 0) {
    atomicAdd(dbeta + c, rp[0]); atomicAdd(dgamma + c, rp[1]);
    atomicAdd(dbeta + c + 1, rp[2]); atomicAdd(dgamma + c + 1, rp[3]);
  }
}

// deterministic tail of conv0_bwd_apply: dw[i] += sum_b ws[b][i]; dbeta / dgamma[c] += sum_b red[b][c][0 / 1], clips in order
__global__ void conv0_bwd_sum_kernel(const float* __restrict__ ws, const float* __restrict__ red, int B, float* dw,
                                     float* dgamma, float* dbeta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C0 * K0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += ws[(long long)b * (C0 * K0) + i];
    dw[i] += s;
  }
  if (i < C0) {
    float sb = 0.f, sg = 0.f;
    for (int b = 0; b < B; ++b) { sb += red[((long long)b * C0 + i) * 2]; sg += red[((long long)b * C0 + i) * 2 + 1]; }
    dbeta[i] += sb;
    dgamma[i] += sg;
  }
}

// ---- weight norm over dims (0,1) of v [Co][Ci][Kk] ------------------------------------------------
__global__ void wn_sumsq_kernel(const float* __restrict__ v, long long rows, int Kk, int rows_per_blk, float* normsq) {
  const int k = threadIdx.x;  // blockDim.x == Kk
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  const long long r1 = min(rows, r0 + rows_per_blk);
  float s = 0.f;
  for (long long r = r0; r < r1; ++r) { const float t = v[r * Kk + k]; s += t * t; }
  atomicAdd(normsq + k, s);
}
__global__ void wn_sqrt_kernel(float* n, int Kk) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < Kk) n[k] = sqrtf(n[k]);
}
__global__ void wn_apply_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norm,
                                int Co, int Ci, int Kk, h16raw* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)Co * Kk * Ci;
       i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    const long long t = i / Ci;
    const int k = (int)(t % Kk), co = (int)(t / Kk);
    out[i] = f2h(v[((long long)co * Ci + ci) * Kk + k] * g[k] / norm[k]);
  }
}
__global__ void wn_bwd_reduce_kernel(const float* __restrict__ dwt, const float* __restrict__ v, int Co, int Ci, int Kk,
                                     int rows_per_blk, float* dot) {
  // dot[k] = sum_{co,ci} dw[co][ci][k] * v[co][ci][k]; dwt is [Co][Kk][Ci]
  const int k = threadIdx.x;
  const long long rows = (long long)Co * Ci;
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  const long long r1 = min(rows, r0 + rows_per_blk);
  float s = 0.f;
  for (long long r = r0; r < r1; ++r) {
    const int co = (int)(r / Ci), ci = (int)(r % Ci);
    s += dwt[((long long)co * Kk + k) * Ci + ci] * v[r * Kk + k];
  }
  atomicAdd(dot + k, s);
}
__global__ void wn_bwd_apply_kernel(const float* __restrict__ dwt, const float* __restrict__ v, const float* __restrict__ g,
                                    const float* __restrict__ norm, const float* __restrict__ dot, int Co, int Ci, int Kk,
                                    float* __restrict__ dv, float* __restrict__ dg) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)Co * Ci * Kk;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kk);
    const long long r = i / Kk;
    const int ci = (int)(r % Ci), co = (int)(r / Ci);
    const float dw = dwt[((long long)co * Kk + k) * Ci + ci];
    const float n = norm[k];
    dv[i] = g[k] / n * (dw - v[i] * dot[k] / (n * n));
    if (i < Kk) dg[i] = dot[i] / norm[i];
  }
}

}  // namespace

#define S_ ((hipStream_t)s)

extern "C" int pp_layernorm_fwd(const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                                float* rstd, int rows, int D, pp_stream_t s) {
  PP_CHECK_ARG(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC, "pp_layernorm_fwd: D=%d unsupported", D);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, S_, (const h16raw*)x, gamma, beta, eps, (h16raw*)y, mean,
                     rstd, rows, D);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern int pp_opt_deterministic;
extern int pp_opt_ln_bwd_alone;
extern "C" int pp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dgamma, float* dbeta, int rows, int D, float* ws, int ws_blocks,
                                pp_stream_t s) {
  PP_CHECK_ARG(rows > 0 && D > 0 && D % 8 == 0 && D <= 64 * 8 * LN_MAXC, "pp_layernorm_bwd: D=%d unsupported", D);
  // with a workspace ([ws_blocks][2][D] fp32): 4 rows per wave, per-block partials + a second pass; without: few, fat
  // waves so that the same-address atomics of dgamma / dbeta stay cheap
  int waves = ws ? (rows + 3) / 4 : (rows + 7) / 8;
  const int cap = ws ? 4 * ws_blocks : 512;
  PP_CHECK_ARG(!ws || ws_blocks > 0, "pp_layernorm_bwd: ws_blocks");
  if (waves > cap) waves = cap;
  const int rows_per_wave = (rows + waves - 1) / waves;
  waves = (rows + rows_per_wave - 1) / rows_per_wave;
  const int nblk = (waves + 3) / 4;
  // pp_set_option("ln_bwd_alone", 1): the workgroup also reserves LN_BWD_ALONE_LDS bytes of LDS it never touches, which
  // leaves less of a CU's 160 KB than any LDS-using kernel of this library needs, so none shares its CUs.  This was the first
  // cure for a run-to-run difference of this kernel beside the register-staged GEMM / weight-gradient kernels (about one
  // launch in ten returned one row of dx from slightly wrong sums; tools/probe/ln_vs_kernels.py, det_trace.py).  The cause
  // turned out to be tied to the packed-FP32 instructions the SLP vectoriser formed here (v_pk_fma_f32 and friends): built
  // with -fno-slp-vectorize (peppa_amd/build.py) the kernel is reproducible beside every neighbour WITHOUT the reservation,
  // which therefore stays off; the switch remains for A/B runs.
  size_t dyn_lds = 0;
  if (pp_opt_ln_bwd_alone > 0) {
    static bool attr_set = false;
    if (!attr_set) {
      const hipError_t e = hipFuncSetAttribute((const void*)ln_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LN_BWD_ALONE_LDS);
      PP_CHECK_ARG(e == hipSuccess, "pp_layernorm_bwd: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
      attr_set = true;
    }
    dyn_lds = LN_BWD_ALONE_LDS;
  }
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), dyn_lds, S_, (const h16raw*)dy, (const h16raw*)x, gamma, mean,
                     rstd, (h16raw*)dx, dgamma, dbeta, rows, D, rows_per_wave, ws);
  if (ws) hipLaunchKernelGGL(ln_bwd_partials_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, S_, ws, nblk, D, dgamma, dbeta);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_softmax_fwd(const float* S, int lds, void* P, int ldp, int nb, int T, float scale, pp_stream_t s) {
  PP_CHECK_ARG(nb > 0 && T > 0 && T <= 64 * SM_MAXC && ldp >= T && ldp <= 64 * SM_MAXC && lds >= T, "pp_softmax_fwd: T=%d ldp=%d unsupported", T, ldp);
  const long long rows = (long long)nb * T;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (ldp <= 256) hipLaunchKernelGGL(softmax_fwd_kernel<4>, grid, block, 0, S_, S, lds, (h16raw*)P, ldp, rows, T, scale);
  else if (ldp <= 512) hipLaunchKernelGGL(softmax_fwd_kernel<8>, grid, block, 0, S_, S, lds, (h16raw*)P, ldp, rows, T, scale);
  else hipLaunchKernelGGL(softmax_fwd_kernel<16>, grid, block, 0, S_, S, lds, (h16raw*)P, ldp, rows, T, scale);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_softmax_bwd(const float* dP, int lds, const void* P, int ldp, void* dS, int nb, int T, float scale,
                              pp_stream_t s) {
  PP_CHECK_ARG(nb > 0 && T > 0 && T <= 64 * SM_MAXC && ldp >= T && ldp <= 64 * SM_MAXC && lds >= T, "pp_softmax_bwd: T=%d ldp=%d unsupported", T, ldp);
  const long long rows = (long long)nb * T;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (ldp <= 256) hipLaunchKernelGGL(softmax_bwd_kernel<4>, grid, block, 0, S_, dP, lds, (const h16raw*)P, ldp, (h16raw*)dS, rows, T, scale);
  else if (ldp <= 512) hipLaunchKernelGGL(softmax_bwd_kernel<8>, grid, block, 0, S_, dP, lds, (const h16raw*)P, ldp, (h16raw*)dS, rows, T, scale);
  else hipLaunchKernelGGL(softmax_bwd_kernel<16>, grid, block, 0, S_, dP, lds, (const h16raw*)P, ldp, (h16raw*)dS, rows, T, scale);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern int pp_opt_deterministic;
static int conv0_red_blocks(int T0, int B) {
  if (pp_opt_deterministic) return 1;   // one workgroup per clip walks every tile: each (clip, channel) sum has ONE adder
  // enough workgroups to fill 256 CUs a few times over, few enough that the final atomics stay cheap
  const int ntiles = (T0 + TT0 - 1) / TT0;
  int per_b = (1024 + B - 1) / B;
  if (per_b > ntiles) per_b = ntiles;
  return per_b < 1 ? 1 : per_b;
}
static int conv0_check(int B, int L, int T0, const char* who) {
  PP_CHECK_ARG(B > 0 && L >= K0 && T0 == (L - K0) / S0 + 1, "%s: T0=%d does not match L=%d", who, T0, L);
  return PP_OK;
}
extern "C" int pp_conv0_stats(const float* wave, int B, int L, int T0, const float* w, float* stats, pp_stream_t s) {
  if (int rc = conv0_check(B, L, T0, "pp_conv0_stats")) return rc;
  hipLaunchKernelGGL(conv0_stats_kernel, dim3(conv0_red_blocks(T0, B), B), dim3(256), 0, S_, wave, L, T0, w, stats);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_conv0_apply(const float* wave, int B, int L, int T0, const float* w, const float* stats, const float* gamma,
                              const float* beta, float eps, void* out, pp_stream_t s) {
  if (int rc = conv0_check(B, L, T0, "pp_conv0_apply")) return rc;
  hipLaunchKernelGGL(conv0_apply_kernel, dim3((T0 + TT0 - 1) / TT0, B), dim3(256), 0, S_, wave, L, T0, w, stats, gamma, beta, eps,
                     (h16raw*)out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_conv0_bwd_reduce(const float* wave, int B, int L, int T0, const float* w, const float* stats,
                                   const float* gamma, const float* beta, float eps, const void* dout, float* red,
                                   pp_stream_t s) {
  if (int rc = conv0_check(B, L, T0, "pp_conv0_bwd_reduce")) return rc;
  hipLaunchKernelGGL(conv0_bwd_reduce_kernel, dim3(conv0_red_blocks(T0, B), B), dim3(256), 0, S_, wave, L, T0, w, stats, gamma, beta,
                     eps, (const h16raw*)dout, red);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_conv0_bwd_apply(const float* wave, int B, int L, int T0, const float* w, const float* stats,
                                  const float* gamma, const float* beta, float eps, const void* dout, const float* red,
                                  float* dw, float* dgamma, float* dbeta, float* ws, pp_stream_t s) {
  if (int rc = conv0_check(B, L, T0, "pp_conv0_bwd_apply")) return rc;
  PP_CHECK_ARG(!pp_opt_deterministic || ws, "pp_conv0_bwd_apply: the deterministic mode needs ws (fp32 [B][512 * 10])");
  float* const slab = pp_opt_deterministic ? ws : nullptr;
  hipLaunchKernelGGL(conv0_bwd_apply_kernel, dim3(conv0_red_blocks(T0, B), B), dim3(256), 0, S_, wave, L, T0, w, stats, gamma, beta,
                     eps, (const h16raw*)dout, red, dw, dgamma, dbeta, slab);
  if (slab) hipLaunchKernelGGL(conv0_bwd_sum_kernel, dim3((C0 * K0 + 255) / 256), dim3(256), 0, S_, slab, red, B, dw, dgamma, dbeta);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_weightnorm_fwd(const float* v, const float* g, int Co, int Ci, int Kk, float* norm, void* out,
                                 pp_stream_t s) {
  PP_CHECK_ARG(Co > 0 && Ci > 0 && Kk > 0 && Kk <= 1024, "pp_weightnorm_fwd: sizes");
  if (hipMemsetAsync(norm, 0, (size_t)Kk * 4, S_) != hipSuccess) { pp_set_error("pp_weightnorm_fwd: memset"); return PP_ERR_HIP; }
  const long long rows = (long long)Co * Ci;
  const int rpb = pp_opt_deterministic ? (int)rows : 64;     // deterministic: one workgroup, rows in order
  hipLaunchKernelGGL(wn_sumsq_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(Kk), 0, S_, v, rows, Kk, rpb, norm);
  hipLaunchKernelGGL(wn_sqrt_kernel, dim3((Kk + 255) / 256), dim3(256), 0, S_, norm, Kk);
  const long long n = rows * Kk;
  hipLaunchKernelGGL(wn_apply_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, S_, v, g,
                     norm, Co, Ci, Kk, (h16raw*)out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_weightnorm_bwd(const float* dwt, const float* v, const float* g, const float* norm, int Co, int Ci, int Kk,
                                 float* dv, float* dg, float* dot_ws, pp_stream_t s) {
  PP_CHECK_ARG(Co > 0 && Ci > 0 && Kk > 0 && Kk <= 1024 && dot_ws, "pp_weightnorm_bwd: sizes");
  if (hipMemsetAsync(dot_ws, 0, (size_t)Kk * 4, S_) != hipSuccess) { pp_set_error("pp_weightnorm_bwd: memset"); return PP_ERR_HIP; }
  const long long rows = (long long)Co * Ci;
  const int rpb = pp_opt_deterministic ? (int)rows : 64;
  hipLaunchKernelGGL(wn_bwd_reduce_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(Kk), 0, S_, dwt, v, Co, Ci, Kk, rpb,
                     dot_ws);
  const long long n = rows * Kk;
  hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, S_, dwt,
                     v, g, norm, dot_ws, Co, Ci, Kk, dv, dg);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
