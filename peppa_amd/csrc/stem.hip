// The paired-pixel stem convolution of r2plus1d_18 / r3d_18 (torchvision `stem[0]`, Conv3d(3, 45 | 64, (1,7,7), stride (1,2,2),
// padding (0,3,3)), reached from pig/models.py:141-150) as a WINDOW kernel (round 4).
//
// Input: [image = b*T + t][Hi][Wp pixel pairs][8] (two pixels x four channels, pp_video_normalize_ndhwc4); the convolution over
// pairs has 7 kernel rows x 4 pair taps (pp_prep_conv_weight_pairs: pair column w reads pairs w - 2 .. w + 1), stride 2 along H.
// The gather kernel asks the L2 for 28 x 16 bytes per output row -- 1.44 GB of 16-byte requests per pass at batch 64 for a
// 103 MB input.  Here a workgroup owns 4 output rows of one image: the 13 input rows they touch sit in
// LDS (13 x 64 pairs x 16 B), each wave takes one output row (up to 64 pairs = four 16-row MFMA tiles), one kernel row is one
// 32-deep MFMA k-block whose four k-slices are its four pair taps -- so an A fragment is ONE 16-byte LDS read at
// (2 oh' + dh, ow + dj) -- and the 21 weight fragments (7 k-blocks x 3 column tiles of 16) stay in registers.
// Same K order as pp_igemm's kernels (k = tap * 8 + channel, ascending 32-deep blocks): same bits in y.
#include "common.h"

namespace {

constexpr int KR = 7;            // kernel rows = MFMA k-blocks
constexpr int WS = 64 + 4;       // window row stride in pairs (64 columns of outputs + the 3 halo pairs, one spare)
constexpr int WROWS = 13;        // input rows under 4 output rows: 2 * 3 + 7
constexpr int NJ = 3;            // column tiles of 16: up to 48 output channels
constexpr int STG = 112;         // staging row stride in bytes (48 columns x 2 + 16)

__global__ __launch_bounds__(256, 2) void stem_pairs_fwd_kernel(const h16raw* __restrict__ x, const h16raw* __restrict__ wf,
                                                             h16raw* __restrict__ y, float* __restrict__ colstats, const int Hi,
                                                             const int Wp, const int Ho, const int Co, const int ldc,
                                                             const int ldstat, const int tiles_per_image, const int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char win[WROWS * WS * 16 + 64];
  __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 64 * STG];
  __shared__ float statbuf[4 * NJ * 16 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;

  // the 21 weight fragments: column n = j * 16 + fr, k-slice fq of k-block kb (= kernel row kb, pair tap fq)
  h16x8 bfr[KR][NJ];
#pragma unroll
  for (int kb = 0; kb < KR; ++kb)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = j * 16 + fr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < Co) v = *(const uint4*)(wf + ((long long)n * KR * 4 + kb * 4 + fq) * 8);
      bfr[kb][j] = __builtin_bit_cast(h16x8, v);
    }
  // persistent workgroups: the weight fragments are fetched once, then tile after tile; the NEXT tile's window is requested
  // (four 16-byte chunks per thread, in registers) before the current one is multiplied and written out
  constexpr int NWC = (WROWS * WS + 255) / 256;
  uint4 wr[NWC];
  auto fetch_window = [&](const int t) __attribute__((always_inline)) {
    const int im = t / tiles_per_image, o0 = (t % tiles_per_image) * 4;
#pragma unroll
    for (int k = 0; k < NWC; ++k) {
      const int idx = tid + 256 * k;
      const int r = idx / WS, c = idx - r * WS;
      const int ih = 2 * o0 - 3 + r, pw = c - 2;
      wr[k] = make_uint4(0, 0, 0, 0);
      if (idx < WROWS * WS && (unsigned)ih < (unsigned)Hi && (unsigned)pw < (unsigned)Wp)
        wr[k] = *(const uint4*)(x + (((long long)im * Hi + ih) * Wp + pw) * 8);
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_window(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int img = tile / tiles_per_image, oh0 = (tile % tiles_per_image) * 4;
  if (tile != (int)blockIdx.x) __syncthreads();      // (everyone is done with the previous tile's window, slabs and sums)
  // the window: rows 2 oh0 - 3 .. 2 oh0 + 9 of the image, pair columns -2 .. WS - 3 (zeros outside the image)
#pragma unroll
  for (int k = 0; k < NWC; ++k) {
    const int idx = tid + 256 * k;
    if (idx < WROWS * WS) *(uint4*)(win + idx * 16) = wr[k];
  }
  __syncthreads();
  if (tile + (int)gridDim.x < ntiles) fetch_window(tile + gridDim.x);

  const int oh = oh0 + wave;
  const bool row_ok = oh < Ho;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[rt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    if (rt * 16 >= Wp) break;            // (wave-uniform: narrower frames skip the empty row tiles)
#pragma unroll
    for (int kb = 0; kb < KR; ++kb) {
      const h16x8 af = *(const h16x8*)(win + ((2 * wave + kb) * WS + rt * 16 + fr + fq) * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[rt][j] = PP_MFMA16(af, bfr[kb][j], acc[rt][j], 0, 0, 0);
    }
  }

  // ---- epilogue: bf16 rows through a wave-private LDS slab, then whole 16-byte chunks; column sums of the fp32 values
  unsigned char* stg = stage + wave * 64 * STG;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) *(h16raw*)(stg + (rt * 16 + fq * 4 + r) * STG + (j * 16 + fr) * 2) = f2h(acc[rt][j][r]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const int nch = ((Co + 7) & ~7) >> 3;                // 8-column chunks per row that are stored
  if (row_ok) {
    const long long m0 = ((long long)img * Ho + oh) * Wp;
    for (int cid = lane; cid < Wp * nch; cid += 64) {
      const int row = cid / nch, ch = cid - row * nch;
      *(uint4*)(y + (m0 + row) * ldc + ch * 8) = *(const uint4*)(stg + row * STG + ch * 16);
    }
  }
  if (colstats) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = row_ok && rt * 16 + fq * 4 + r < Wp;
          const float v = ok ? acc[rt][j][r] : 0.f;
          s1 += v;
          s2 += v * v;
        }
      s1 = sum_rows4(s1);
      s2 = sum_rows4(s2);
      if (fq == 0) {
        statbuf[((wave * NJ + j) * 16 + fr) * 2 + 0] = s1;
        statbuf[((wave * NJ + j) * 16 + fr) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < NJ * 16 && tid < ldstat) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {           // (rows past Ho contributed zeros)
        a += statbuf[((w * NJ) * 16 + tid) * 2 + 0];
        b += statbuf[((w * NJ) * 16 + tid) * 2 + 1];
      }
      colstats[((long long)tile * 2 + 0) * ldstat + tid] = a;
      colstats[((long long)tile * 2 + 1) * ldstat + tid] = b;
    }
  }
  }
}


// ---- weight gradient on the same window ---------------------------------------------------------------------------------------
// dW[co][tap * 8 + c] = sum over output positions m of dY[m][co] * X[m's window at tap][c].  The reduce index of the MFMA is m:
// 32 consecutive output columns of one output row per k-block.  Both fragments are transposing LDS reads (ds_read_b64_tr_b16,
// the same row -> k permutation on either side): dY^T from the tile's dY rows ([m][48 columns], 96-byte rows), X^T from the
// window seen as a matrix of 16-byte rows (one pair) and 16 columns (two pair taps x 8 channels) starting at pair ow + dj.
// A wave owns the kernel rows {wave, wave + 4} (x two halves of the four pair taps): 3 x 4 accumulator tiles; persistent
// workgroups add their sums to dW once, at the end (fp32 atomics: not for the deterministic mode).
constexpr int DYS = 96;          // dY tile row stride in bytes (48 columns): 32 x 3, conflict-free transposing reads

__device__ __forceinline__ h16x8 tr_frag32(const unsigned char* tile, int stride, int lane) {
  // rows {4g..4g+3} and {16+4g..16+4g+3} (g = lane >> 4) of 16 columns; lane (4q+p) supplies row q, columns 4p..4p+3
  const int gq = lane >> 4, li = lane & 15;
  const unsigned char* a0 = tile + (4 * gq + (li >> 2)) * stride + (li & 3) * 8;
  const h16x4 lo = ds_read_tr16(a0);
  const h16x4 hi = ds_read_tr16(a0 + 16 * stride);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 4) void stem_pairs_wgrad_kernel(const h16raw* __restrict__ x, const h16raw* __restrict__ dy,
                                                                  float* __restrict__ dw, const int Hi, const int Wp, const int Ho,
                                                                  const int Co, const int ldy, const int tiles_per_image,
                                                                  const int ntiles) {
  __shared__ __attribute__((aligned(16))) unsigned char win[WROWS * WS * 16 + 64];
  __shared__ __attribute__((aligned(16))) unsigned char dyt[4 * 64 * DYS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[NJ][4];           // [co tile][slot]: slot = (kernel row wave or wave + 4) x (pair taps {0,1} or {2,3})
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) acc[j][sl] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool second = wave + 4 < KR;             // (wave 3 owns kernel row 3 only)

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int img = tile / tiles_per_image, oh0 = (tile % tiles_per_image) * 4;
    if (tile != (int)blockIdx.x) __syncthreads();
    for (int idx = tid; idx < WROWS * WS; idx += 256) {
      const int r = idx / WS, c = idx - r * WS;
      const int ih = 2 * oh0 - 3 + r, pw = c - 2;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((unsigned)ih < (unsigned)Hi && (unsigned)pw < (unsigned)Wp) v = *(const uint4*)(x + (((long long)img * Hi + ih) * Wp + pw) * 8);
      *(uint4*)(win + idx * 16) = v;
    }
    // the tile's dY: 4 output rows x 64 positions (zeros past Wp / Ho) x 48 columns (6 chunks of 8)
    for (int idx = tid; idx < 4 * 64 * 6; idx += 256) {
      const int ch = idx % 6, m = idx / 6;
      const int orow = m >> 6, ow = m & 63;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (oh0 + orow < Ho && ow < Wp && ch * 8 < ldy) v = *(const uint4*)(dy + (((long long)img * Ho + oh0 + orow) * Wp + ow) * ldy + ch * 8);
      *(uint4*)(dyt + m * DYS + ch * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int orow = 0; orow < 4; ++orow) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if (mb * 32 >= Wp) break;
        h16x8 af[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) af[j] = tr_frag32(dyt + (orow * 64 + mb * 32) * DYS + j * 32, DYS, lane);
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          if (sl >= 2 && !second) break;
          const int dh = wave + 4 * (sl >> 1), dj = 2 * (sl & 1);
          const h16x8 bf = tr_frag32(win + ((2 * orow + dh) * WS + mb * 32 + dj) * 16, 16, lane);
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j][sl] = PP_MFMA16(af[j], bf, acc[j][sl], 0, 0, 0);
        }
      }
    }
  }
  // acc[j][sl][r] = dW[co = j * 16 + fq * 4 + r][k = (dh * 4 + dj) * 8 + fr]
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    if (sl >= 2 && !second) break;
    const int dh = wave + 4 * (sl >> 1), dj = 2 * (sl & 1);
    const int k = (dh * 4 + dj) * 8 + fr;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = j * 16 + fq * 4 + r;
        if (co < Co) atomicAdd(dw + (long long)co * (KR * 4 * 8) + k, acc[j][sl][r]);
      }
  }
}

}  // namespace

// rows of column statistics pp_stem_pairs_fwd writes: one per (image, 4 output rows)
extern "C" long long pp_stem_pairs_stat_rows(int images, int Hi) {
  const int Ho = (Hi + 2 * 3 - 7) / 2 + 1;
  return (long long)images * ((Ho + 3) / 4);
}

extern "C" int pp_stem_pairs_fwd(const void* x, const void* wf, void* y, float* colstats, int images, int Hi, int Wp, int Co,
                                 int ldc, int ldstat, pp_stream_t s) {
  PP_CHECK_ARG(x && wf && y && images > 0 && Hi >= 7 && Wp >= 4 && Wp <= 64 && Co > 0 && Co <= 48, "pp_stem_pairs_fwd: sizes (Wp <= 64, Co <= 48)");
  PP_CHECK_ARG(ldc % 8 == 0 && ldc >= ((Co + 7) & ~7) && (!colstats || ldstat >= Co), "pp_stem_pairs_fwd: ldc / ldstat");
  const int Ho = (Hi + 2 * 3 - 7) / 2 + 1;
  const int tiles = (Ho + 3) / 4;
  PP_CHECK_ARG((long long)images * tiles < 0x7fffffffLL && (long long)images * Hi * Wp * 8 < 0x7fffffffLL, "pp_stem_pairs_fwd: too large");
  const long long ntiles = (long long)images * tiles;
  const long long gx = ntiles < 256 * 2 ? ntiles : 256 * 2;       // two workgroups per CU (216 registers; three spill: 321 us against 150)
  hipLaunchKernelGGL(stem_pairs_fwd_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)s, (const h16raw*)x,
                     (const h16raw*)wf, (h16raw*)y, colstats, Hi, Wp, Ho, Co, ldc, ldstat, tiles, (int)ntiles);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

/* dw [Co][7 x 4 taps][8] fp32 += the stem's weight gradient (the caller zeroes dw); x, Hi, Wp, Co as pp_stem_pairs_fwd,
 * dy [images * Ho * Wp][ldy] bf16.  fp32 atomics across workgroups: not bitwise reproducible. */
extern "C" int pp_stem_pairs_wgrad(const void* x, const void* dy, float* dw, int images, int Hi, int Wp, int Co, int ldy,
                                   pp_stream_t s) {
  PP_CHECK_ARG(x && dy && dw && images > 0 && Hi >= 7 && Wp >= 4 && Wp <= 64 && Co > 0 && Co <= 48, "pp_stem_pairs_wgrad: sizes (Wp <= 64, Co <= 48)");
  PP_CHECK_ARG(ldy % 8 == 0 && ldy >= ((Co + 7) & ~7), "pp_stem_pairs_wgrad: ldy");
  const int Ho = (Hi + 2 * 3 - 7) / 2 + 1;
  const int tiles = (Ho + 3) / 4;
  const long long ntiles = (long long)images * tiles;
  PP_CHECK_ARG(ntiles < 0x7fffffffLL && (long long)images * Hi * Wp * 8 < 0x7fffffffLL && (long long)images * Ho * Wp * ldy < 0x7fffffffLL, "pp_stem_pairs_wgrad: too large");
  const long long gx = ntiles < 256 * 4 ? ntiles : 256 * 4;       // four workgroups per CU (90 registers, 39 KB of LDS): 2: 216, 3: 177, 4: 164 us
  hipLaunchKernelGGL(stem_pairs_wgrad_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)s, (const h16raw*)x, (const h16raw*)dy,
                     dw, Hi, Wp, Ho, Co, ldy, tiles, (int)ntiles);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
