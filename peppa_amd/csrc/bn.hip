// Train-mode BatchNorm3d on channels-last bf16 activations [M rows][Cp channels] (gfx950).
// HBM-bound streaming kernels: 16-byte (8-channel) accesses, per-channel partial sums kept
// deterministic (per-block slabs + a finalize pass, no float atomics).
// Replaces the 37 torch.nn.BatchNorm3d layers of torchvision r2plus1d_18 (pig/models.py:141-150).
#include "common.h"

extern int pp_opt_bn_nt;     // bit 0: non-temporal loads, bit 1: non-temporal stores in the streaming passes
extern int pp_opt_bn_grid;   // workgroup cap of the streaming passes

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16(const h16raw* p) {
  if (NT) {
    const u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
  return *(const uint4*)p;
}
template <bool NT>
__device__ __forceinline__ void st16(h16raw* p, const uint4 v) {
  if (NT) {
    u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (u32x4_t*)p);
  } else {
    *(uint4*)p = v;
  }
}

// Column-partial reducer: block b handles rows [b*rows_per_blk, ...); thread owns one 8-channel
// chunk and strides over rows; NACC running sums per channel.
template <int NACC, class F>
__device__ __forceinline__ void col_reduce(long long M, int Cp, int rows_per_blk, float* partials, F f) {
  extern __shared__ float red[];
  const int cpr = Cp >> 3;               // chunks per row
  const int rpb = 256 / cpr;             // rows processed concurrently
  const int tid = threadIdx.x;
  const int ch = tid % cpr, rsub = tid / cpr;
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk;
  if (r1 > M) r1 = M;
  float acc[NACC][8];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[a][q] = 0.f;
  if (rsub < rpb)
    for (long long r = r0 + rsub; r < r1; r += rpb) f(r, ch, acc);
  // reduce across rsub through LDS: red[rsub][NACC][Cp]
  if (rsub < rpb) {
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int q = 0; q < 8; ++q) red[(rsub * NACC + a) * Cp + ch * 8 + q] = acc[a][q];
  }
  __syncthreads();
  for (int i = tid; i < NACC * Cp; i += 256) {
    float s = 0.f;
    for (int r = 0; r < rpb; ++r) s += red[r * NACC * Cp + i];
    partials[(long long)blockIdx.x * NACC * Cp + i] = s;
  }
}

__global__ __launch_bounds__(256) void colstats_kernel(const h16raw* __restrict__ y, long long M, int Cp,
                                                       int rows_per_blk, float* partials) {
  col_reduce<2>(M, Cp, rows_per_blk, partials, [&](long long r, int ch, float (*acc)[8]) {
    const uint4 v = *(const uint4*)(y + r * Cp + ch * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int q = 0; q < 8; ++q) { acc[0][q] += f[q]; acc[1][q] += f[q] * f[q]; }
  });
}

// first level of the statistics reduction: [nblk][2][ld] -> [nslice][2][ld]
__global__ __launch_bounds__(256) void partials_reduce_kernel(const float* __restrict__ partials, int nblk, int ld,
                                                              int per_slice, float* __restrict__ out) {
  __shared__ float s1s[16][17], s2s[16][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int b0 = blockIdx.y * per_slice, b1 = min(nblk, b0 + per_slice);
  float s1 = 0.f, s2 = 0.f;
  if (c < ld)
    for (int b = b0 + part; b < b1; b += 16) {
      s1 += partials[((long long)b * 2 + 0) * ld + c];
      s2 += partials[((long long)b * 2 + 1) * ld + c];
    }
  s1s[part][cl] = s1;
  s2s[part][cl] = s2;
  __syncthreads();
  if (part == 0 && c < ld) {
    for (int q = 1; q < 16; ++q) { s1 += s1s[q][cl]; s2 += s2s[q][cl]; }
    out[((long long)blockIdx.y * 2 + 0) * ld + c] = s1;
    out[((long long)blockIdx.y * 2 + 1) * ld + c] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials, int nblk, int ldstat,
                                                          double count, int C, int Cp, const float* gamma,
                                                          const float* beta, float eps, float momentum,
                                                          float* running_mean, float* running_var, float* mean,
                                                          float* rstd, float* scale, float* shift) {
  __shared__ double s1s[16][17], s2s[16][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < ldstat)
    for (int b = part; b < nblk; b += 16) {
      s1 += partials[((long long)b * 2 + 0) * ldstat + c];
      s2 += partials[((long long)b * 2 + 1) * ldstat + c];
    }
  s1s[part][cl] = s1;
  s2s[part][cl] = s2;
  __syncthreads();
  if (part == 0 && c < Cp) {
    for (int q = 1; q < 16; ++q) { s1 += s1s[q][cl]; s2 += s2s[q][cl]; }
    if (c < C) {
      const double mu = s1 / count;
      double var = s2 / count - mu * mu;
      if (var < 0.0) var = 0.0;
      const float rs = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[c] * rs;
      mean[c] = (float)mu;
      rstd[c] = rs;
      scale[c] = sc;
      shift[c] = beta[c] - (float)mu * sc;
      if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    } else {
      mean[c] = 0.f; rstd[c] = 0.f; scale[c] = 0.f; shift[c] = 0.f;
    }
  }
}

// Streaming passes: a thread keeps ONE 8-channel chunk (its per-channel parameters live in registers for the whole
// launch) and walks rows; 256 / cpr rows are covered per block pass, so a pass reads one contiguous span.  Two rows
// are in flight per thread and iteration.
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&f)[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn_apply_kernel(const h16raw* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const h16raw* __restrict__ res,
                                                       int relu, h16raw* __restrict__ z, long long M, int cpr) {
  const int rpb = 256 / cpr;
  const int tid = threadIdx.x;
  if (tid >= rpb * cpr) return;
  const int ch = tid % cpr, rsub = tid / cpr;
  float sc[8], sh[8];
  load8(scale + ch * 8, sc);
  load8(shift + ch * 8, sh);
  auto one = [&](const uint4 v, const uint4 rv) __attribute__((always_inline)) -> uint4 {
    float f[8], r[8];
    unpack8(v, f);
    if (res) unpack8(rv, r);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float t = f[q] * sc[q] + sh[q];
      if (res) t += r[q];
      if (relu) t = fmaxf(t, 0.f);
      f[q] = t;
    }
    return pack8(f);
  };
  const long long stride = (long long)gridDim.x * rpb;
  long long r = (long long)blockIdx.x * rpb + rsub;
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  for (; r + stride < M; r += 2 * stride) {
    const long long i0 = (r * cpr + ch) * 8, i1 = ((r + stride) * cpr + ch) * 8;
    const uint4 v0 = ld16<NTL>(y + i0), v1 = ld16<NTL>(y + i1);
    const uint4 r0 = res ? ld16<NTL>(res + i0) : zero4, r1 = res ? ld16<NTL>(res + i1) : zero4;
    st16<NTS>(z + i0, one(v0, r0));
    st16<NTS>(z + i1, one(v1, r1));
  }
  if (r < M) {
    const long long i0 = (r * cpr + ch) * 8;
    st16<NTS>(z + i0, one(ld16<NTL>(y + i0), res ? ld16<NTL>(res + i0) : zero4));
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const h16raw* __restrict__ dz, const h16raw* __restrict__ y,
                                                            const h16raw* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, long long M, int Cp,
                                                            int rows_per_blk, float* partials) {
  const int my_ch = threadIdx.x % (Cp >> 3);   // col_reduce gives this thread the same chunk on every row
  float mu[8], rs[8], sc[8], sh[8];
  load8(mean + my_ch * 8, mu);
  load8(rstd + my_ch * 8, rs);
  if (relu && !z) { load8(scale + my_ch * 8, sc); load8(shift + my_ch * 8, sh); }
  col_reduce<2>(M, Cp, rows_per_blk, partials, [&](long long r, int ch, float (*acc)[8]) {
    const long long o = r * Cp + ch * 8;
    float d[8], yy[8], zz[8];
    unpack8(*(const uint4*)(dz + o), d);
    unpack8(*(const uint4*)(y + o), yy);
    if (relu) {
      if (z) unpack8(*(const uint4*)(z + o), zz);
      else  // no residual: the ReLU mask is recomputed from y instead of reading z (one stream less)
#pragma unroll
        for (int q = 0; q < 8; ++q) zz[q] = yy[q] * sc[q] + sh[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float g = (relu && !(zz[q] > 0.f)) ? 0.f : d[q];
      const float xh = (yy[q] - mu[q]) * rs[q];
      acc[0][q] += g;
      acc[1][q] += g * xh;
    }
  });
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, double count,
                                                               int C, int Cp, const float* gamma, const float* rstd,
                                                               float* dgamma, float* dbeta, float* coef) {
  // 16 channels x 64 row groups per block (round 4; was x 16): only ceil(Cp / 16) blocks read the up to 2048 partial rows
  // (2.4 MB at 144 channels), so the loads in flight per block set the time -- 35 us at layer 1 with 256 threads
  constexpr int PARTS = 64;
  __shared__ double s1s[PARTS][17], s2s[PARTS][17];
  const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < Cp) {
    // four independent chains: the loop is a string of dependent ~250 ns loads otherwise
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    int b = part;
    for (; b + 3 * PARTS < nblk; b += 4 * PARTS)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a1[u] += partials[((long long)(b + PARTS * u) * 2 + 0) * Cp + c];
        a2[u] += partials[((long long)(b + PARTS * u) * 2 + 1) * Cp + c];
      }
    for (; b < nblk; b += PARTS) {
      a1[0] += partials[((long long)b * 2 + 0) * Cp + c];
      a2[0] += partials[((long long)b * 2 + 1) * Cp + c];
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  }
  s1s[part][cl] = s1;
  s2s[part][cl] = s2;
  __syncthreads();
  if (part == 0 && c < Cp) {
    for (int q = 1; q < PARTS; ++q) { s1 += s1s[q][cl]; s2 += s2s[q][cl]; }
    if (c < C) {
      dbeta[c] = (float)s1;
      dgamma[c] = (float)s2;
      coef[c] = gamma[c] * rstd[c];
      coef[Cp + c] = (float)(s1 / count);
      coef[2 * Cp + c] = (float)(s2 / count);
    } else {
      coef[c] = 0.f; coef[Cp + c] = 0.f; coef[2 * Cp + c] = 0.f;
    }
  }
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const h16raw* __restrict__ dz, const h16raw* __restrict__ y,
                                                           const h16raw* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ coef,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, h16raw* __restrict__ dy, h16raw* __restrict__ dres,
                                                           long long M, int cpr, int Cp) {
  const int rpb = 256 / cpr;
  const int tid = threadIdx.x;
  if (tid >= rpb * cpr) return;
  const int ch = tid % cpr, rsub = tid / cpr;
  float mu[8], rs[8], k0[8], k1[8], k2[8], sc[8], sh[8];
  load8(mean + ch * 8, mu);
  load8(rstd + ch * 8, rs);
  load8(coef + ch * 8, k0);
  load8(coef + Cp + ch * 8, k1);
  load8(coef + 2 * Cp + ch * 8, k2);
  if (relu && !z) { load8(scale + ch * 8, sc); load8(shift + ch * 8, sh); }
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  auto one = [&](const long long i, const uint4 vd, const uint4 vy, const uint4 vz) __attribute__((always_inline)) {
    float d[8], yy[8], zz[8], o[8];
    unpack8(vd, d);
    unpack8(vy, yy);
    if (relu) {
      if (z) unpack8(vz, zz);
      else
#pragma unroll
        for (int q = 0; q < 8; ++q) zz[q] = yy[q] * sc[q] + sh[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float g = (relu && !(zz[q] > 0.f)) ? 0.f : d[q];
      const float xh = (yy[q] - mu[q]) * rs[q];
      o[q] = k0[q] * (g - k1[q] - xh * k2[q]);
      d[q] = g;
    }
    st16<NTS>(dy + i, pack8(o));
    if (dres) st16<NTS>(dres + i, pack8(d));
  };
  const long long stride = (long long)gridDim.x * rpb;
  long long r = (long long)blockIdx.x * rpb + rsub;
  for (; r + stride < M; r += 2 * stride) {
    const long long i0 = (r * cpr + ch) * 8, i1 = ((r + stride) * cpr + ch) * 8;
    const uint4 d0 = ld16<NTL>(dz + i0), d1 = ld16<NTL>(dz + i1);
    const uint4 y0 = ld16<NTL>(y + i0), y1 = ld16<NTL>(y + i1);
    const uint4 z0 = (relu && z) ? ld16<NTL>(z + i0) : zero4, z1 = (relu && z) ? ld16<NTL>(z + i1) : zero4;
    one(i0, d0, y0, z0);
    one(i1, d1, y1, z1);
  }
  if (r < M) {
    const long long i0 = (r * cpr + ch) * 8;
    one(i0, ld16<NTL>(dz + i0), ld16<NTL>(y + i0), (relu && z) ? ld16<NTL>(z + i0) : zero4);
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, int Cp, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  if (c < C) {
    const float sc = gamma[c] * rsqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  } else {
    scale[c] = 0.f;
    shift[c] = 0.f;
  }
}

int stream_grid(long long nchunks) {
  long long b = (nchunks + 255) / 256;
  if (b > pp_opt_bn_grid) b = pp_opt_bn_grid;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define CHECK_CP(Cp, who) PP_CHECK_ARG((Cp) > 0 && (Cp) % 8 == 0 && (Cp) <= 2048, who ": Cp=%d must be a multiple of 8, <= 2048", (Cp))

extern "C" int pp_colstats_bf16(const void* y, long long M, int Cp, float* partials, int nblk, pp_stream_t s) {
  CHECK_CP(Cp, "pp_colstats_bf16");
  PP_CHECK_ARG(M > 0 && nblk > 0, "pp_colstats_bf16: bad sizes");
  const int rows_per_blk = (int)((M + nblk - 1) / nblk);
  const int rpb = 256 / (Cp / 8);
  hipLaunchKernelGGL(colstats_kernel, dim3(nblk), dim3(256), (size_t)rpb * 2 * Cp * 4, (hipStream_t)s, (const h16raw*)y, M,
                     Cp, rows_per_blk, partials);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// column sums of a partials table: out[2][ld] = sum_b partials[b][2][ld] (the quantity SyncBN all-reduces across ranks)
extern "C" int pp_partials_sum(const float* partials, int nblk, int ld, float* ws /* [64][2][ld] */, float* out,
                               pp_stream_t s) {
  PP_CHECK_ARG(partials && out && ws && nblk > 0 && ld > 0, "pp_partials_sum: bad arguments");
  const int nslice = nblk > 64 ? 64 : 1, per_slice = (nblk + nslice - 1) / nslice;
  float* first = nslice > 1 ? ws : out;
  hipLaunchKernelGGL(partials_reduce_kernel, dim3((ld + 15) / 16, nslice), dim3(256), 0, (hipStream_t)s, partials, nblk, ld,
                     per_slice, first);
  if (nslice > 1)
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((ld + 15) / 16, 1), dim3(256), 0, (hipStream_t)s, ws, nslice, ld, nslice, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_finalize(const float* partials, int nblk, int ldstat, long long count, int C, int Cp,
                              const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* mean, float* rstd, float* scale, float* shift, float* ws,
                              pp_stream_t s) {
  PP_CHECK_ARG(C > 0 && Cp >= C && ldstat >= Cp && nblk > 0 && count > 0, "pp_bn_finalize: bad sizes");
  if (ws && nblk > 256) {  // two-level reduction: 64 slices in parallel, then the finalize over 64 rows
    const int nslice = 64, per_slice = (nblk + nslice - 1) / nslice;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((ldstat + 15) / 16, nslice), dim3(256), 0, (hipStream_t)s, partials, nblk,
                       ldstat, per_slice, ws);
    partials = ws;
    nblk = nslice;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((Cp + 15) / 16), dim3(256), 0, (hipStream_t)s, partials, nblk, ldstat,
                     (double)count, C, Cp, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_apply(const void* y, const float* scale, const float* shift, const void* res, int relu, void* z,
                           long long M, int Cp, pp_stream_t s) {
  CHECK_CP(Cp, "pp_bn_apply");
  const long long nchunks = M * (Cp / 8);
#define LAUNCH_APPLY(NTL, NTS)                                                                                           \
  hipLaunchKernelGGL((bn_apply_kernel<NTL, NTS>), dim3(stream_grid(nchunks / 2)), dim3(256), 0, (hipStream_t)s, (const h16raw*)y, \
                     scale, shift, (const h16raw*)res, relu, (h16raw*)z, M, Cp / 8)
  switch (pp_opt_bn_nt & 3) {
    case 0: LAUNCH_APPLY(false, false); break;
    case 1: LAUNCH_APPLY(true, false); break;
    case 2: LAUNCH_APPLY(false, true); break;
    default: LAUNCH_APPLY(true, true); break;
  }
#undef LAUNCH_APPLY
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_bwd_reduce(const void* dz, const void* y, const void* z, const float* mean, const float* rstd,
                                const float* scale, const float* shift, int relu, float* partials, int nblk, long long M,
                                int Cp, pp_stream_t s) {
  CHECK_CP(Cp, "pp_bn_bwd_reduce");
  PP_CHECK_ARG(!relu || z || (scale && shift), "pp_bn_bwd_reduce: relu needs z or scale/shift");
  const int rows_per_blk = (int)((M + nblk - 1) / nblk);
  const int rpb = 256 / (Cp / 8);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), (size_t)rpb * 2 * Cp * 4, (hipStream_t)s,
                     (const h16raw*)dz, (const h16raw*)y, (const h16raw*)z, mean, rstd, scale, shift, relu, M, Cp, rows_per_blk, partials);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_bwd_finalize(const float* partials, int nblk, long long count, int C, int Cp, const float* gamma,
                                  const float* rstd, float* dgamma, float* dbeta, float* coef, float* ws, pp_stream_t s) {
  PP_CHECK_ARG(partials && nblk > 0 && count > 0 && C > 0 && Cp >= C, "pp_bn_bwd_finalize: bad sizes");
  if (ws && nblk > 256) {  // two-level reduction, as pp_bn_finalize: 64 slices in parallel (fixed order), then 64 rows
    const int nslice = 64, per_slice = (nblk + nslice - 1) / nslice;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((Cp + 15) / 16, nslice), dim3(256), 0, (hipStream_t)s, partials, nblk, Cp,
                       per_slice, ws);
    partials = ws;
    nblk = nslice;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((Cp + 15) / 16), dim3(1024), 0, (hipStream_t)s, partials, nblk,
                     (double)count, C, Cp, gamma, rstd, dgamma, dbeta, coef);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_bwd_apply(const void* dz, const void* y, const void* z, const float* mean, const float* rstd,
                               const float* coef, const float* scale, const float* shift, int relu, void* dy, void* dres,
                               long long M, int Cp, pp_stream_t s) {
  PP_CHECK_ARG(!relu || z || (scale && shift), "pp_bn_bwd_apply: relu needs z or scale/shift");
  CHECK_CP(Cp, "pp_bn_bwd_apply");
  const long long nchunks = M * (Cp / 8);
#define LAUNCH_BAPPLY(NTL, NTS)                                                                                                   \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<NTL, NTS>), dim3(stream_grid(nchunks / 2)), dim3(256), 0, (hipStream_t)s, (const h16raw*)dz, \
                     (const h16raw*)y, (const h16raw*)z, mean, rstd, coef, scale, shift, relu, (h16raw*)dy, (h16raw*)dres, M, Cp / 8, Cp)
  switch (pp_opt_bn_nt & 3) {
    case 0: LAUNCH_BAPPLY(false, false); break;
    case 1: LAUNCH_BAPPLY(true, false); break;
    case 2: LAUNCH_BAPPLY(false, true); break;
    default: LAUNCH_BAPPLY(true, true); break;
  }
#undef LAUNCH_BAPPLY
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, int C, int Cp, float* scale, float* shift, pp_stream_t s) {
  PP_CHECK_ARG(C > 0 && Cp >= C, "pp_bn_eval_affine: sizes");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((Cp + 255) / 256), dim3(256), 0, (hipStream_t)s, gamma, beta, running_mean,
                     running_var, eps, C, Cp, scale, shift);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
