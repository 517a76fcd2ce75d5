// Streaming / layout kernels (HBM-bound, 16-byte accesses) for libpeppa_hip.so (gfx950).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";
void pp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pp_last_error(void) { return g_err; }
extern "C" int pp_version(void) { return 200; }
extern const int pp_exp_win_ablate, pp_exp_tw_ablate, pp_exp_ln_variant;   // igemm_win.hip, wgrad_tw.hip, norm.hip
extern "C" int pp_experimental_build(void) {
  return (pp_exp_win_ablate != 0) | ((pp_exp_tw_ablate != 0) << 1) | ((pp_exp_ln_variant != 0) << 2);
}
#ifdef PP_F16
extern "C" int pp_dtype(void) { return PP_DTYPE_F16; }
#else
extern "C" int pp_dtype(void) { return PP_DTYPE_BF16; }
#endif

extern int pp_opt_deterministic;

namespace {

inline int sgrid(long long n, int per = 256) {
  long long b = (n + per - 1) / per;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}
#define GSTRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__global__ void gelu_fwd_kernel(const h16raw* __restrict__ x, h16raw* __restrict__ y, long long nch) {
  GSTRIDE(i, nch) {
    float f[8];
    unpack8(*(const uint4*)(x + i * 8), f);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = gelu_f(f[q]);
    *(uint4*)(y + i * 8) = pack8(f);
  }
}
__global__ void gelu_bwd_kernel(const h16raw* __restrict__ dy, const h16raw* __restrict__ x, h16raw* __restrict__ dx,
                                long long nch) {
  GSTRIDE(i, nch) {
    float f[8], d[8];
    unpack8(*(const uint4*)(x + i * 8), f);
    unpack8(*(const uint4*)(dy + i * 8), d);
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] *= gelu_grad_f(f[q]);
    *(uint4*)(dx + i * 8) = pack8(d);
  }
}
__global__ void add_kernel(const h16raw* __restrict__ a, const h16raw* __restrict__ b, h16raw* __restrict__ o, long long nch) {
  GSTRIDE(i, nch) {
    float f[8], d[8];
    unpack8(*(const uint4*)(a + i * 8), f);
    unpack8(*(const uint4*)(b + i * 8), d);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] += d[q];
    *(uint4*)(o + i * 8) = pack8(f);
  }
}
__global__ void cast_f2b_kernel(const float* __restrict__ in, h16raw* __restrict__ out, long long n) {
  GSTRIDE(i, n) out[i] = f2h(in[i]);
}
__global__ void cast_b2f_kernel(const h16raw* __restrict__ in, float* __restrict__ out, long long n) {
  GSTRIDE(i, n) out[i] = h2f(in[i]);
}
__global__ void fill_kernel(float* p, float v, long long n) { GSTRIDE(i, n) p[i] = v; }

__global__ void copy2d_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out, int rows,
                              int cols) {
  GSTRIDE(i, (long long)rows * cols) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[(long long)r * ld_out + c] = in[(long long)r * ld_in + c];
  }
}

// out[r][c] (rows_out x ld_out, zero padded) = in[r][c] or in[c][r]
__global__ void cast_pad_2d_kernel(const float* __restrict__ in, int rows, int cols, int ld_in, h16raw* __restrict__ out,
                                   int rows_out, int cols_out, int ld_out, int transpose) {
  GSTRIDE(i, (long long)rows_out * cols_out) {
    const int r = (int)(i / cols_out), c = (int)(i % cols_out);
    float v = 0.f;
    if (r < rows && c < cols) v = transpose ? in[(long long)c * ld_in + r] : in[(long long)r * ld_in + c];
    out[(long long)r * ld_out + c] = f2h(v);
  }
}

// the same, for a device table of jobs in one launch (blockIdx.y = job): the ~140 per-step weight casts of the audio
// tower are 3-8 us launches each; as one launch they are a few
struct CastItem { const float* in; void* out; long long rows, cols, ld_in, rows_out, cols_out, ld_out, transpose, out_f32; };
__global__ __launch_bounds__(256) void cast_pad_2d_multi_kernel(const CastItem* __restrict__ items) {
  // 32 x 32 output tiles through LDS, so that the transposing jobs read AND write whole rows
  __shared__ float tile[32][33];
  const CastItem it = items[blockIdx.y];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
  const long long tr = (it.rows_out + 31) / 32, tc = (it.cols_out + 31) / 32;
  for (long long t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const long long r0 = (t / tc) * 32, c0 = (t % tc) * 32;
    if (it.transpose) {   // out[r][c] = in[c][r]: read rows of `in` (index c), columns r0 + tx
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long c = c0 + ty + 8 * k, r = r0 + tx;
        tile[ty + 8 * k][tx] = (r < it.rows && c < it.cols) ? it.in[c * it.ld_in + r] : 0.f;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long r = r0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (r < it.rows && c < it.cols) ? it.in[r * it.ld_in + c] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long r = r0 + ty + 8 * k, c = c0 + tx;
      if (r < it.rows_out && c < it.cols_out) {
        const float v = it.transpose ? tile[tx][ty + 8 * k] : tile[ty + 8 * k][tx];
        if (it.out_f32) ((float*)it.out)[r * it.ld_out + c] = v;
        else ((h16raw*)it.out)[r * it.ld_out + c] = f2h(v);
      }
    }
    __syncthreads();
  }
}

// w [Co][Ci][taps] -> out[row][tap][cg]; row = co (or ci when transpose_io), channel = ci (or co)
__global__ void prep_conv_kernel(const float* __restrict__ w, int Co, int Ci, int taps, h16raw* __restrict__ out,
                                 int rows_out, int cg, int transpose_io, int flip, float scale) {
  GSTRIDE(i, (long long)rows_out * taps * cg) {
    const int c = (int)(i % cg);
    const long long t2 = i / cg;
    int tap = (int)(t2 % taps);
    const int row = (int)(t2 / taps);
    if (flip) tap = taps - 1 - tap;
    const int co = transpose_io ? c : row, ci = transpose_io ? row : c;
    float v = 0.f;
    if (co < Co && ci < Ci) v = w[((long long)co * Ci + ci) * taps + tap] * scale;
    out[i] = f2h(v);
  }
}
// the same for a device table of weights in one launch: a tower's ~40 convolutions x 2 layouts are 4-8 us launches each on
// its critical stream.  One block = 2048 consecutive output elements of one weight; item i owns blocks [blk0, next item's blk0).
struct PrepItem { const float* w; h16raw* out; long long Co, Ci, taps, rows_out, cg, transpose_io, flip, blk0; float scale; int pad; };
__global__ __launch_bounds__(256) void prep_conv_multi_kernel(const PrepItem* __restrict__ items, const int n) {
  int lo = 0, hi = n - 1;                       // last item whose first block is <= blockIdx.x (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PrepItem it = items[lo];
  // a thread converts ONE 8-channel chunk (cg is a multiple of 8): two 32-bit divisions per 16-byte store instead of
  // three 64-bit ones per element, which is where the per-weight kernel spends its time
  const unsigned Co = (unsigned)it.Co, Ci = (unsigned)it.Ci, taps = (unsigned)it.taps, cg8 = (unsigned)it.cg >> 3;
  const unsigned nchunks = (unsigned)it.rows_out * taps * cg8;       // (the launcher checked < 2^31 elements)
  const unsigned q = (unsigned)(blockIdx.x - (unsigned)it.blk0) * 256u + threadIdx.x;
  if (q >= nchunks) return;
  const unsigned c0 = (q % cg8) * 8u, t2 = q / cg8;
  unsigned tap = t2 % taps;
  const unsigned row = t2 / taps;
  if (it.flip) tap = taps - 1 - tap;
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const unsigned c = c0 + u;
    const unsigned co = it.transpose_io ? c : row, ci = it.transpose_io ? row : c;
    v[u] = (co < Co && ci < Ci) ? it.w[((size_t)co * Ci + ci) * taps + tap] * it.scale : 0.f;
  }
  *(uint4*)(it.out + (size_t)q * 8) = pack8(v);
}
__global__ void unprep_conv_kernel(const float* __restrict__ g, int Co, int Ci, int taps, int cg, float* __restrict__ dw) {
  GSTRIDE(i, (long long)Co * Ci * taps) {
    const int tap = (int)(i % taps);
    const long long t2 = i / taps;
    const int ci = (int)(t2 % Ci), co = (int)(t2 / Ci);
    dw[i] = g[((long long)co * taps + tap) * cg + ci];
  }
}

struct TapSel { int v[32]; };
__global__ void select_taps_kernel(const uint4* __restrict__ w, int taps, int cg8, TapSel sel, int nsel, uint4* __restrict__ out,
                                   long long n) {
  GSTRIDE(i, n) {   // i over rows * nsel * cg8 16-byte chunks
    const int c = (int)(i % cg8);
    const long long t = i / cg8;
    const int si = (int)(t % nsel);
    const long long r = t / nsel;
    out[i] = w[(r * taps + sel.v[si]) * cg8 + c];
  }
}

// batched bf16 transpose through LDS: in [R][ld_in] (C cols) -> out [C][ld_out] (R cols, zero padded)
__global__ __launch_bounds__(256) void transpose_kernel(const h16raw* __restrict__ in, long long in_bs, int ld_in,
                                                        h16raw* __restrict__ out, long long out_bs, int ld_out, int R, int C,
                                                        int inner, long long in_s1, long long out_s1, int r_pad) {
  __shared__ h16raw tile[32][33];
  const int z = blockIdx.z;
  const h16raw* ip = in + (z / inner) * in_bs + (z % inner) * in_s1;
  h16raw* op = out + (z / inner) * out_bs + (z % inner) * out_s1;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? ip[(long long)r * ld_in + c] : (h16raw)0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < r_pad) op[(long long)c * ld_out + r] = tile[tx][k];
  }
}

// x fp32 [B][3][T][H][W] -> out bf16 [B][T][H][W][CPP] (CPP = 8, or 4: two pixels per 16-byte chunk, see pp_prep_conv_weight_pairs)
template <int CPP>
__global__ void video_norm_kernel(const float* __restrict__ x, h16raw* __restrict__ out, long long npos, long long thw,
                                  float m0, float m1, float m2, float i0, float i1, float i2) {
  GSTRIDE(i, npos) {
    const long long b = i / thw, p = i % thw;
    const float* xp = x + b * 3 * thw + p;
    float f[8] = {(xp[0] - m0) * i0, (xp[thw] - m1) * i1, (xp[2 * thw] - m2) * i2, 0, 0, 0, 0, 0};
    const uint4 v = pack8(f);
    if (CPP == 8) *(uint4*)(out + i * 8) = v;
    else *(uint2*)(out + i * 4) = make_uint2(v.x, v.y);
  }
}
// Stem weights for an input of PAIRED pixels.  A stride-2 conv over [W][4 channels] reads, for output column w, the
// pixels 2 w + dw - pw; pixel pairs (2 j, 2 j + 1) are the 16-byte chunks [W/2][8] of the same memory, so it equals a
// stride-1 conv over pairs j = w + dj - pj with kw' = dj_max - dj_min + 1 taps (7 -> 4): 8-channel chunks carry two real
// pixels instead of one real pixel and five zeros, K = 49 x 8 -> 28 x 8 for the (1,7,7) stem.
//   w [Co][Ci][kth][kw] fp32 -> out [Co][kth * kwp][8] 16-bit, element e = q * 4 + c of pair tap dj <- w[co][c][a][2 (dj - pj) + q + pw]
__global__ void prep_conv_pairs_kernel(const float* __restrict__ w, int Co, int Ci, int kth, int kw, int pw, int kwp, int pj,
                                       h16raw* __restrict__ out) {
  GSTRIDE(i, (long long)Co * kth * kwp * 8) {
    const int e = (int)(i & 7);
    const long long t = i >> 3;
    const int dj = (int)(t % kwp);
    const long long t2 = t / kwp;
    const int a = (int)(t2 % kth), co = (int)(t2 / kth);
    const int q = e >> 2, c = e & 3;
    const int dw = 2 * (dj - pj) + q + pw;
    float v = 0.f;
    if (c < Ci && dw >= 0 && dw < kw) v = w[(((long long)co * Ci + c) * kth + a) * kw + dw];
    out[i] = f2h(v);
  }
}
// ... and the way back for the gradient: g [Co][kth * kwp][8] fp32 -> dw [Co][Ci][kth][kw]
__global__ void unprep_conv_pairs_kernel(const float* __restrict__ g, int Co, int Ci, int kth, int kw, int pw, int kwp, int pj,
                                         float* __restrict__ dwt) {
  GSTRIDE(i, (long long)Co * Ci * kth * kw) {
    const int dw = (int)(i % kw);
    const long long t = i / kw;
    const int a = (int)(t % kth);
    const long long t2 = t / kth;
    const int c = (int)(t2 % Ci), co = (int)(t2 / Ci);
    const int u = dw - pw + 2 * pj;          // = 2 dj + q
    const int dj = u >> 1, q = u & 1;
    dwt[i] = g[(((long long)co * kth + a) * kwp + dj) * 8 + q * 4 + c];
  }
}

// column sums of bf16 [M][ld] -> out fp32 [N] (atomics over row slabs; out zeroed by the launcher)
__global__ __launch_bounds__(256) void colsum_kernel(const h16raw* __restrict__ x, long long M, int N, int ld,
                                                     int rows_per_blk, float* out) {
  const int cpr = (N + 7) / 8;
  const long long r0 = (long long)blockIdx.x * rows_per_blk;
  long long r1 = r0 + rows_per_blk;
  if (r1 > M) r1 = M;
  for (int ch = threadIdx.x; ch < cpr; ch += 256) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long r = r0; r < r1; ++r) {
      float f[8];
      unpack8(*(const uint4*)(x + r * ld + ch * 8), f);
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += f[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (ch * 8 + q < N) atomicAdd(out + ch * 8 + q, acc[q]);
  }
}

}  // namespace

#define S_ ((hipStream_t)s)
#define CHK8(n, who) PP_CHECK_ARG((n) > 0 && (n) % 8 == 0, who ": n=%lld must be a positive multiple of 8", (long long)(n))

extern "C" int pp_gelu_fwd(const void* x, void* y, long long n, pp_stream_t s) {
  CHK8(n, "pp_gelu_fwd");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, (const h16raw*)x, (h16raw*)y, n / 8);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_gelu_bwd(const void* dy, const void* x, void* dx, long long n, pp_stream_t s) {
  CHK8(n, "pp_gelu_bwd");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, (const h16raw*)dy, (const h16raw*)x, (h16raw*)dx, n / 8);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_add_bf16(const void* a, const void* b, void* out, long long n, pp_stream_t s) {
  CHK8(n, "pp_add_bf16");
  hipLaunchKernelGGL(add_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, (const h16raw*)a, (const h16raw*)b, (h16raw*)out, n / 8);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_cast_f32_to_bf16(const float* in, void* out, long long n, pp_stream_t s) {
  PP_CHECK_ARG(n > 0, "pp_cast_f32_to_bf16: n");
  hipLaunchKernelGGL(cast_f2b_kernel, dim3(sgrid(n)), dim3(256), 0, S_, in, (h16raw*)out, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_cast_bf16_to_f32(const void* in, float* out, long long n, pp_stream_t s) {
  PP_CHECK_ARG(n > 0, "pp_cast_bf16_to_f32: n");
  hipLaunchKernelGGL(cast_b2f_kernel, dim3(sgrid(n)), dim3(256), 0, S_, (const h16raw*)in, out, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_fill_f32(float* p, float v, long long n, pp_stream_t s) {
  PP_CHECK_ARG(n > 0, "pp_fill_f32: n");
  hipLaunchKernelGGL(fill_kernel, dim3(sgrid(n)), dim3(256), 0, S_, p, v, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_copy_2d_f32(const float* in, int ld_in, float* out, int ld_out, int rows, int cols, pp_stream_t s) {
  PP_CHECK_ARG(rows > 0 && cols > 0, "pp_copy_2d_f32: sizes");
  hipLaunchKernelGGL(copy2d_kernel, dim3(sgrid((long long)rows * cols)), dim3(256), 0, S_, in, ld_in, out, ld_out, rows, cols);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_cast_pad_2d(const float* in, int rows, int cols, int ld_in, void* out, int rows_out, int cols_out,
                              int ld_out, int transpose, pp_stream_t s) {
  PP_CHECK_ARG(rows > 0 && cols > 0 && rows_out >= rows && cols_out >= cols && ld_out >= cols_out, "pp_cast_pad_2d: sizes");
  hipLaunchKernelGGL(cast_pad_2d_kernel, dim3(sgrid((long long)rows_out * cols_out)), dim3(256), 0, S_, in, rows, cols, ld_in,
                     (h16raw*)out, rows_out, cols_out, ld_out, transpose);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_cast_pad_2d_multi(const void* items, int n, int blocks_per_item, pp_stream_t s) {
  PP_CHECK_ARG(items && n > 0 && n <= 65535 && blocks_per_item > 0, "pp_cast_pad_2d_multi: bad arguments");
  static_assert(sizeof(CastItem) == 80, "pp_cast2d_item layout (10 x 8 bytes)");
  hipLaunchKernelGGL(cast_pad_2d_multi_kernel, dim3(blocks_per_item, n), dim3(256), 0, S_, (const CastItem*)items);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_prep_conv_weight(const float* w, int Co, int Ci, int taps, void* out, int rows_out, int cg,
                                   int transpose_io, int flip, float scale, pp_stream_t s) {
  PP_CHECK_ARG(Co > 0 && Ci > 0 && taps > 0 && rows_out > 0 && cg > 0 && cg % 8 == 0, "pp_prep_conv_weight: sizes");
  PP_CHECK_ARG(transpose_io ? (rows_out >= Ci && cg >= Co) : (rows_out >= Co && cg >= Ci), "pp_prep_conv_weight: pad too small");
  hipLaunchKernelGGL(prep_conv_kernel, dim3(sgrid((long long)rows_out * taps * cg)), dim3(256), 0, S_, w, Co, Ci, taps,
                     (h16raw*)out, rows_out, cg, transpose_io, flip, scale);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
// many contiguous fp32 copies in one launch (gradient buckets: one per bucket instead of one hipMemcpyAsync per tensor)
struct CopyItem { const float* src; float* dst; long long n, blk0; };
__global__ __launch_bounds__(256) void copy_f32_multi_kernel(const CopyItem* __restrict__ items, const int n) {
  int lo = 0, hi = n - 1;                       // last item whose first block is <= blockIdx.x (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const CopyItem it = items[lo];
  const long long base = ((long long)blockIdx.x - it.blk0) * 4096;
  if ((((uintptr_t)it.src | (uintptr_t)it.dst) & 15) == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = base + (u * 256 + threadIdx.x) * 4;
      if (i + 3 < it.n) {
        *(float4*)(it.dst + i) = *(const float4*)(it.src + i);
      } else {
        for (long long j = i; j < it.n; ++j) it.dst[j] = it.src[j];
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const long long i = base + u * 256 + threadIdx.x;
      if (i < it.n) it.dst[i] = it.src[i];
    }
  }
}
extern "C" int pp_copy_f32_multi(const void* items, int n, long long total_blocks, pp_stream_t s) {
  PP_CHECK_ARG(items && n > 0 && total_blocks > 0 && total_blocks < 0x7fffffffLL, "pp_copy_f32_multi: bad arguments");
  static_assert(sizeof(CopyItem) == 32, "pp_copy_item layout (4 x 8 bytes)");
  hipLaunchKernelGGL(copy_f32_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, S_, (const CopyItem*)items, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_prep_conv_weight_multi(const void* items, int n, long long total_blocks, pp_stream_t s) {
  PP_CHECK_ARG(items && n > 0 && total_blocks > 0 && total_blocks < 0x7fffffffLL, "pp_prep_conv_weight_multi: bad arguments");
  // (every `out` has fewer than 2^31 elements and is 16-byte aligned: the caller's table, peppa_amd/hip.py checks both)
  static_assert(sizeof(PrepItem) == 88, "pp_prep_item layout (10 x 8 bytes, float scale, 4 bytes of padding)");
  hipLaunchKernelGGL(prep_conv_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, S_, (const PrepItem*)items, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_select_taps(const void* w, int rows, int taps, int cg, const int* sel, int nsel, void* out, pp_stream_t s) {
  PP_CHECK_ARG(rows > 0 && taps > 0 && cg > 0 && cg % 8 == 0 && nsel > 0 && nsel <= 32 && sel, "pp_select_taps: sizes");
  TapSel ts;
  for (int i = 0; i < nsel; ++i) {
    PP_CHECK_ARG(sel[i] >= 0 && sel[i] < taps, "pp_select_taps: tap %d out of range", sel[i]);
    ts.v[i] = sel[i];
  }
  const long long n = (long long)rows * nsel * (cg / 8);
  hipLaunchKernelGGL(select_taps_kernel, dim3(sgrid(n)), dim3(256), 0, S_, (const uint4*)w, taps, cg / 8, ts, nsel, (uint4*)out, n);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_unprep_conv_grad(const float* g, int Co, int Ci, int taps, int cg, float* dw, pp_stream_t s) {
  PP_CHECK_ARG(Co > 0 && Ci > 0 && taps > 0 && cg >= Ci, "pp_unprep_conv_grad: sizes");
  hipLaunchKernelGGL(unprep_conv_kernel, dim3(sgrid((long long)Co * Ci * taps)), dim3(256), 0, S_, g, Co, Ci, taps, cg, dw);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_transpose_bf16(const void* in, long long in_bs, int ld_in, void* out, long long out_bs, int ld_out, int nb,
                                 int R, int C, int inner, long long in_s1, long long out_s1, int r_pad, pp_stream_t s) {
  if (r_pad <= 0) r_pad = ld_out;
  PP_CHECK_ARG(nb > 0 && R > 0 && C > 0 && r_pad >= R && ld_out >= r_pad && ld_in >= C, "pp_transpose_bf16: sizes");
  if (inner <= 0) inner = 1;
  dim3 grid((C + 31) / 32, (r_pad + 31) / 32, nb);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, S_, (const h16raw*)in, in_bs, ld_in, (h16raw*)out, out_bs, ld_out, R, C,
                     inner, in_s1, out_s1, r_pad);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_video_normalize_ndhwc(const float* x, void* out, int B, int T, int H, int W, const float* mean3,
                                        const float* std3, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0 && mean3 && std3, "pp_video_normalize_ndhwc: sizes");
  const long long thw = (long long)T * H * W;
  hipLaunchKernelGGL(video_norm_kernel<8>, dim3(sgrid(B * thw)), dim3(256), 0, S_, x, (h16raw*)out, B * thw, thw, mean3[0],
                     mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_video_normalize_ndhwc4(const float* x, void* out, int B, int T, int H, int W, const float* mean3,
                                         const float* std3, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0 && mean3 && std3, "pp_video_normalize_ndhwc4: sizes");
  const long long thw = (long long)T * H * W;
  hipLaunchKernelGGL(video_norm_kernel<4>, dim3(sgrid(B * thw)), dim3(256), 0, S_, x, (h16raw*)out, B * thw, thw, mean3[0],
                     mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
// pair geometry of a stride-2 kernel row: taps dj_min .. dj_max of the pair index, see prep_conv_pairs_kernel
static inline void pair_taps(int kw, int pw, int* kwp, int* pj) {
  auto fl2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
  const int lo = fl2(-pw), hi = fl2(kw - 1 - pw);
  *kwp = hi - lo + 1;
  *pj = -lo;
}
extern "C" int pp_prep_conv_weight_pairs(const float* w, int Co, int Ci, int kth, int kw, int pw, void* out, pp_stream_t s) {
  PP_CHECK_ARG(w && out && Co > 0 && Ci > 0 && Ci <= 4 && kth > 0 && kw > 0 && pw >= 0, "pp_prep_conv_weight_pairs: sizes (Ci <= 4)");
  int kwp, pj;
  pair_taps(kw, pw, &kwp, &pj);
  hipLaunchKernelGGL(prep_conv_pairs_kernel, dim3(sgrid((long long)Co * kth * kwp * 8)), dim3(256), 0, S_, w, Co, Ci, kth, kw, pw,
                     kwp, pj, (h16raw*)out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_unprep_conv_grad_pairs(const float* g, int Co, int Ci, int kth, int kw, int pw, float* dw, pp_stream_t s) {
  PP_CHECK_ARG(g && dw && Co > 0 && Ci > 0 && Ci <= 4 && kth > 0 && kw > 0 && pw >= 0, "pp_unprep_conv_grad_pairs: sizes (Ci <= 4)");
  int kwp, pj;
  pair_taps(kw, pw, &kwp, &pj);
  hipLaunchKernelGGL(unprep_conv_pairs_kernel, dim3(sgrid((long long)Co * Ci * kth * kw)), dim3(256), 0, S_, g, Co, Ci, kth, kw,
                     pw, kwp, pj, dw);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_colsum_bf16(const void* x, long long M, int N, int ld, float* out, pp_stream_t s) {
  PP_CHECK_ARG(M > 0 && N > 0 && ld % 8 == 0 && ld >= ((N + 7) & ~7), "pp_colsum_bf16: sizes");
  hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, S_);
  if (e != hipSuccess) { pp_set_error("pp_colsum_bf16: memset failed"); return PP_ERR_HIP; }
  int nblk = (int)((M + 63) / 64);
  if (nblk > 1024) nblk = 1024;
  if (pp_opt_deterministic) nblk = 1;     // one workgroup, rows in order: a single adder per column
  const int rows_per_blk = (int)((M + nblk - 1) / nblk);
  hipLaunchKernelGGL(colsum_kernel, dim3(nblk), dim3(256), 0, S_, (const h16raw*)x, M, N, ld, rows_per_blk, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- MaxPool (1,3,3)/(1,2,2)/(0,1,1) on channels-last bf16 (torchvision resnet18.maxpool, pig/models.py:185) ----
namespace {
__device__ __forceinline__ void max8(float* m, const float* v) {
#pragma unroll
  for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], v[q]);
}
__global__ void maxpool_fwd_kernel(const h16raw* __restrict__ x, h16raw* __restrict__ y, int N, int H, int W, int Ho, int Wo,
                                   int cpr) {
  GSTRIDE(i, (long long)N * Ho * Wo * cpr) {
    const int c = (int)(i % cpr);
    long long t = i / cpr;
    const int xo = (int)(t % Wo); t /= Wo;
    const int yo = (int)(t % Ho);
    const long long n = t / Ho;
    float m[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = -3.0e38f;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) {
        const int yi = yo * 2 - 1 + dy, xi = xo * 2 - 1 + dx;
        if ((unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W) {
          float v[8];
          unpack8(*(const uint4*)(x + (((n * H + yi) * W + xi) * cpr + c) * 8), v);
          max8(m, v);
        }
      }
    *(uint4*)(y + i * 8) = pack8(m);
  }
}
// gather form of the backward: an input position receives dY of every window whose FIRST maximum it is
__global__ void maxpool_bwd_kernel(const h16raw* __restrict__ x, const h16raw* __restrict__ dy, h16raw* __restrict__ dx, int N,
                                   int H, int W, int Ho, int Wo, int cpr) {
  GSTRIDE(i, (long long)N * H * W * cpr) {
    const int c = (int)(i % cpr);
    long long t = i / cpr;
    const int xi = (int)(t % W); t /= W;
    const int yi = (int)(t % H);
    const long long n = t / H;
    float self[8], acc[8];
    unpack8(*(const uint4*)(x + i * 8), self);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int yo = (yi + 1) / 2 - ((yi + 1) % 2 == 0 ? 1 : 0); yo <= (yi + 1) / 2; ++yo) {
      if ((unsigned)yo >= (unsigned)Ho) continue;
      for (int xo = (xi + 1) / 2 - ((xi + 1) % 2 == 0 ? 1 : 0); xo <= (xi + 1) / 2; ++xo) {
        if ((unsigned)xo >= (unsigned)Wo) continue;
        // is (yi, xi) the first maximum of window (yo, xo)?  before = strictly-greater needed, after = greater-or-equal
        bool first[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) first[q] = true;
        for (int dyy = 0; dyy < 3; ++dyy)
          for (int dxx = 0; dxx < 3; ++dxx) {
            const int y2 = yo * 2 - 1 + dyy, x2 = xo * 2 - 1 + dxx;
            if ((unsigned)y2 >= (unsigned)H || (unsigned)x2 >= (unsigned)W || (y2 == yi && x2 == xi)) continue;
            float v[8];
            unpack8(*(const uint4*)(x + (((n * H + y2) * W + x2) * cpr + c) * 8), v);
            const bool before = (y2 < yi) || (y2 == yi && x2 < xi);
#pragma unroll
            for (int q = 0; q < 8; ++q) first[q] = first[q] && (before ? (v[q] < self[q]) : (v[q] <= self[q]));
          }
        float g[8];
        unpack8(*(const uint4*)(dy + (((n * Ho + yo) * Wo + xo) * cpr + c) * 8), g);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += first[q] ? g[q] : 0.f;
      }
    }
    *(uint4*)(dx + i * 8) = pack8(acc);
  }
}
}  // namespace

extern "C" int pp_maxpool3x3s2_fwd(const void* x, void* y, int N, int H, int W, int Cp, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "pp_maxpool3x3s2_fwd: sizes");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(sgrid((long long)N * Ho * Wo * (Cp / 8))), dim3(256), 0, S_, (const h16raw*)x,
                     (h16raw*)y, N, H, W, Ho, Wo, Cp / 8);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int Cp, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0, "pp_maxpool3x3s2_bwd: sizes");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(sgrid((long long)N * H * W * (Cp / 8))), dim3(256), 0, S_, (const h16raw*)x,
                     (const h16raw*)dy, (h16raw*)dx, N, H, W, Ho, Wo, Cp / 8);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- dropout (torchaudio wav2vec2 components: feature projection, transformer, attention, feed-forward) ------
// Counter-based mask: element i is kept iff hash16(seed, i) >= p * 65536; the same (seed, i) regenerates the
// mask in the backward pass, so no mask tensor is stored.  y = keep ? x / (1 - p) : 0  (+ res).
namespace {
__global__ void dropout_bf16_kernel(const h16raw* __restrict__ x, const h16raw* __restrict__ res, h16raw* __restrict__ y, long long nch,
                                    uint32_t thr, float scale, uint32_t seed) {
  GSTRIDE(i, nch) {
    float f[8], r[8];
    bool keep[8];
    unpack8(*(const uint4*)(x + i * 8), f);
    keep8(seed, i, thr, keep);
    if (res) unpack8(*(const uint4*)(res + i * 8), r);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = keep[q] ? f[q] * scale : 0.f;
      if (res) v += r[q];
      f[q] = v;
    }
    *(uint4*)(y + i * 8) = pack8(f);
  }
}
// dx = dropout_bwd(dy) * gelu'(x) in one pass (the feed-forward's intermediate dropout sits right behind its GELU)
__global__ void gelu_bwd_dropout_kernel(const h16raw* __restrict__ dy, const h16raw* __restrict__ x, h16raw* __restrict__ dx,
                                        long long nch, uint32_t thr, float scale, uint32_t seed) {
  GSTRIDE(i, nch) {
    float f[8], d[8];
    bool keep[8];
    unpack8(*(const uint4*)(x + i * 8), f);
    unpack8(*(const uint4*)(dy + i * 8), d);
    keep8(seed, i, thr, keep);
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = keep[q] ? d[q] * scale * gelu_grad_f(f[q]) : 0.f;
    *(uint4*)(dx + i * 8) = pack8(d);
  }
}
__global__ void dropout_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long nch, uint32_t thr, float scale,
                                   uint32_t seed) {
  GSTRIDE(i, nch) {
    bool keep[8];
    keep8(seed, i, thr, keep);
#pragma unroll
    for (int q = 0; q < 8; ++q) y[i * 8 + q] = keep[q] ? x[i * 8 + q] * scale : 0.f;
  }
}
}  // namespace

extern "C" int pp_dropout_bf16(const void* x, const void* res, void* y, long long n, float p, unsigned seed, pp_stream_t s) {
  CHK8(n, "pp_dropout_bf16");
  PP_CHECK_ARG(p >= 0.f && p < 1.f, "pp_dropout_bf16: p=%f", (double)p);
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, (const h16raw*)x, (const h16raw*)res, (h16raw*)y, n / 8,
                     (uint32_t)(p * 65536.f + 0.5f), 1.f / (1.f - p), seed);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_gelu_bwd_dropout(const void* dy, const void* x, void* dx, long long n, float p, unsigned seed, pp_stream_t s) {
  CHK8(n, "pp_gelu_bwd_dropout");
  PP_CHECK_ARG(p >= 0.f && p < 1.f, "pp_gelu_bwd_dropout: p=%f", (double)p);
  hipLaunchKernelGGL(gelu_bwd_dropout_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, (const h16raw*)dy, (const h16raw*)x,
                     (h16raw*)dx, n / 8, (uint32_t)(p * 65536.f + 0.5f), 1.f / (1.f - p), seed);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_dropout_f32(const float* x, float* y, long long n, float p, unsigned seed, pp_stream_t s) {
  CHK8(n, "pp_dropout_f32");
  PP_CHECK_ARG(p >= 0.f && p < 1.f, "pp_dropout_f32: p=%f", (double)p);
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(sgrid(n / 8)), dim3(256), 0, S_, x, y, n / 8, (uint32_t)(p * 65536.f + 0.5f),
                     1.f / (1.f - p), seed);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
