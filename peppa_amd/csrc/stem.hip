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
