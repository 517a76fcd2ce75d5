// Weight-gradient contraction on MFMA (gfx950):
//
//   dW[i, j] += sum_m dY[m, i] * X_gather[m, j]        i < Ni (output channels), j < Kj (taps x channels)
//
// The reduce index m (output positions; up to millions of rows for r2plus1d layer1) is the slow
// memory axis of BOTH operands, so each 32-row slab is staged in LDS exactly as loaded
// (row-major, 16-byte chunks) and the MFMA fragments (8 consecutive m per lane) are fetched with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read.  M is split across workgroups; partial
// tiles are combined with fp32 global atomics (dW is small: <= 1152 x 10368).
// Replaces the conv/linear weight-gradient of torch autograd for the modules named in igemm.hip.
#include "common.h"

int pp_validate_gather(const pp_gather& g, int K, const char* who);

namespace {

constexpr int TJ = 128;   // j extent per workgroup (8 MFMA tiles, 2 per wave)
constexpr int MS = 32;    // m rows per step
constexpr int QS = 288;   // Q row stride in bytes (256 + 32: 32*odd -> conflict-free tr reads)

struct WGeom {
  FastDiv dRw, dRh, dRt;
};

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int stride, int col0, int lane) {
  // lanes 16g..16g+15 fetch rows {4g..4g+3} and {16+4g..16+4g+3} of columns col0..col0+15;
  // lane (4q+p) supplies the address of row q, columns 4p..4p+3 and receives column (lane&15).
  const int gq = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  const unsigned char* a0 = tile + (4 * gq + q) * stride + (col0 + 4 * pp) * 2;
  const unsigned char* a1 = a0 + 16 * stride;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int WI, int MODE>
__global__ __launch_bounds__(256) void wgrad_kernel(const pp_wgrad_desc p, const WGeom wg, const int nblk_i,
                                                    const int nblk_j, const int rows_per_split) {
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;  // P row stride (bytes), 32*odd
  constexpr int P_BYTES = MS * PS;
  constexpr int Q_BYTES = MS * QS;
  constexpr int NPI = (WI * 64 + 255) / 256;           // P chunk iterations per thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[P_BYTES + Q_BYTES];
  __shared__ int lut[128];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int ib = bid % nblk_i; bid /= nblk_i;
  const int jb = bid % nblk_j; bid /= nblk_j;
  const int split = bid;
  const int z = blockIdx.z;
  const bfraw* __restrict__ X = (const bfraw*)p.X + z * p.x_s;
  const bfraw* __restrict__ dY = (const bfraw*)p.dY + z * p.dy_s;
  float* __restrict__ dW = p.dW + z * p.dw_s;
  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    if (tid < 128) {
      int e = 0;
      if (tid < ntaps) {
        const int dw = tid % g.kw;
        const int t2 = tid / g.kw;
        e = (t2 / g.kh) | ((t2 % g.kh) << 8) | (dw << 16);
      }
      lut[tid] = e;
    }
    __syncthreads();
  }

  const int i0 = ib * TI, j0 = jb * TJ;
  const int m_begin = split * rows_per_split;
  const int m_end = min(p.M, m_begin + rows_per_split);

  // Q (gathered X) chunk owned by this thread: rows (tid>>4) and 16+(tid>>4), columns j0 + 8*(tid&15)
  const int qrow = tid >> 4, qch = tid & 15;
  const int jq = j0 + qch * 8;
  const bool jq_ok = jq < p.Kj;
  int dt = 0, dh = 0, dw = 0, cch = 0;
  if (MODE != PP_DENSE && jq_ok) {
    const int tap = jq / g.cg;
    cch = jq % g.cg;
    const int e = lut[tap];
    dt = e & 0xff; dh = (e >> 8) & 0xff; dw = (e >> 16) & 0xff;
  }

  const int qt0 = dt - g.pt, qh0 = dh - g.ph, qw0 = dw - g.pw;
  auto load_q = [&](int m) -> uint4 {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (!jq_ok || m >= m_end) return zero;
    if (MODE == PP_DENSE) return *(const uint4*)(X + (m * g.lda + jq));
    const uint32_t t1 = fdiv((uint32_t)m, wg.dRw);
    const int rw = m - t1 * g.Rw;
    const uint32_t t2 = fdiv(t1, wg.dRh);
    const int rh = t1 - t2 * g.Rh;
    const uint32_t n = fdiv(t2, wg.dRt);
    const int rt = t2 - n * g.Rt;
    const int gt = rt * g.st + qt0, gh = rh * g.sh + qh0, gw = rw * g.sw + qw0;
    if ((unsigned)gt >= (unsigned)g.Gt || (unsigned)gh >= (unsigned)g.Gh || (unsigned)gw >= (unsigned)g.Gw)
      return zero;
    // 32-bit element offsets (host-checked: the source tensor has < 2^31 elements)
    return *(const uint4*)(X + ((((int)n * g.Gt + gt) * g.Gh + gh) * g.Gw + gw) * g.cstride + cch);
  };
  auto load_p = [&](int it, int mbase) -> uint4 {
    const int cid = tid + 256 * it;
    const int row = cid / (2 * WI), ch = cid % (2 * WI);
    const int m = mbase + row;
    const int i = i0 + ch * 8;
    if (row < MS && m < m_end && i < p.ldy) return *(const uint4*)(dY + (m * p.ldy + i));
    return make_uint4(0, 0, 0, 0);
  };

  f32x4 acc[WI][2];
#pragma unroll
  for (int a = 0; a < WI; ++a) acc[a][0] = acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // optional bias gradient (column sums of dY), taken from the P tiles this block stages anyway
  const bool do_bias = p.dbias != nullptr && jb == 0;
  float bsum[NPI][8];
#pragma unroll
  for (int it = 0; it < NPI; ++it)
#pragma unroll
    for (int q = 0; q < 8; ++q) bsum[it][q] = 0.f;

  uint4 rq[2], rp[NPI];
  rq[0] = load_q(m_begin + qrow);
  rq[1] = load_q(m_begin + 16 + qrow);
#pragma unroll
  for (int it = 0; it < NPI; ++it) rp[it] = load_p(it, m_begin);

  unsigned char* Pt = smem;
  unsigned char* Qt = smem + P_BYTES;
  for (int mb = m_begin; mb < m_end; mb += MS) {
    *(uint4*)(Qt + qrow * QS + qch * 16) = rq[0];
    *(uint4*)(Qt + (16 + qrow) * QS + qch * 16) = rq[1];
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      const int cid = tid + 256 * it;
      const int row = cid / (2 * WI), ch = cid % (2 * WI);
      if (row < MS) *(uint4*)(Pt + row * PS + ch * 16) = rp[it];
      if (do_bias) {
        float f[8];
        unpack8(rp[it], f);
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum[it][q] += f[q];
      }
    }
    __syncthreads();
    const int mn = mb + MS;
    if (mn < m_end) {
      rq[0] = load_q(mn + qrow);
      rq[1] = load_q(mn + 16 + qrow);
#pragma unroll
      for (int it = 0; it < NPI; ++it) rp[it] = load_p(it, mn);
    }
    const bf16x8 b0 = tr_frag(Qt, QS, (2 * wave) * 16, lane);
    const bf16x8 b1 = tr_frag(Qt, QS, (2 * wave + 1) * 16, lane);
#pragma unroll
    for (int a = 0; a < WI; ++a) {
      const bf16x8 af = tr_frag(Pt, PS, a * 16, lane);
      acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b0, acc[a][0], 0, 0, 0);
      acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b1, acc[a][1], 0, 0, 0);
    }
    __syncthreads();
  }

  if (do_bias) {
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      const int cid = tid + 256 * it;
      const int row = cid / (2 * WI), ch = cid % (2 * WI);
      if (row < MS)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = i0 + ch * 8 + q;
          if (i < p.Ni) atomicAdd(p.dbias + z * p.dbias_s + i, bsum[it][q]);
        }
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < WI; ++a)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = j0 + (2 * wave + jj) * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + a * 16 + fq * 4 + r;
        if (i < p.Ni && j < p.Kj) atomicAdd(dW + (long long)i * p.ldw + j, acc[a][jj][r]);
      }
    }
}

int pick_wi(int n16) {
  static const int cand[] = {15, 9, 8, 4, 3, 2};
  static const float eff[] = {1.0f, 0.97f, 0.95f, 0.8f, 0.7f, 0.55f};
  int best = 2;
  float best_cost = 1e30f;
  for (int i = 0; i < 6; ++i) {
    const int c = cand[i];
    const float cost = (float)(((n16 + c - 1) / c) * c) / eff[i];
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

template <int WI>
int launch_wi(const pp_wgrad_desc& d, hipStream_t s) {
  const int nblk_i = (d.Ni + 16 * WI - 1) / (16 * WI);
  const int nblk_j = (d.Kj + TJ - 1) / TJ;
  int msplit = d.msplit;
  const long long steps = ((long long)d.M + MS - 1) / MS;
  if (msplit <= 0) {
    // aim for ~2048 workgroups with at least 24 steps each: every split adds one fp32 atomic per
    // output element (chip-wide atomic rate ~1.3 TB/s), so short splits are atomics-bound
    const long long tiles = (long long)nblk_i * nblk_j * d.nbatch;
    long long want = (2048 + tiles - 1) / tiles;
    const long long maxs = (steps + 23) / 24;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    msplit = (int)want;
  }
  long long sps = (steps + msplit - 1) / msplit;  // steps per split
  msplit = (int)((steps + sps - 1) / sps);
  const int rows_per_split = (int)(sps * MS);
  WGeom wg;
  wg.dRw = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rw));
  wg.dRh = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rh));
  wg.dRt = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rt));
  const long long gx = (long long)nblk_i * nblk_j * msplit;
  dim3 grid((unsigned)gx, 1, (unsigned)d.nbatch), block(256);
  if (d.g.mode == PP_DENSE)
    hipLaunchKernelGGL((wgrad_kernel<WI, PP_DENSE>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split);
  else
    hipLaunchKernelGGL((wgrad_kernel<WI, PP_CONV_FWD>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

extern "C" int pp_wgrad(const pp_wgrad_desc* dp, pp_stream_t stream) {
  PP_CHECK_ARG(dp != nullptr, "pp_wgrad: null descriptor");
  pp_wgrad_desc d = *dp;
  PP_CHECK_ARG(d.M > 0 && d.Ni > 0 && d.Kj > 0, "pp_wgrad: bad sizes");
  PP_CHECK_ARG(d.X && d.dY && d.dW, "pp_wgrad: null operand");
  PP_CHECK_ARG(d.ldy % 8 == 0 && d.ldy >= ((d.Ni + 7) & ~7), "pp_wgrad: ldy=%d too small/unaligned for Ni=%d", d.ldy, d.Ni);
  PP_CHECK_ARG(d.Kj % 8 == 0 && d.ldw >= d.Kj, "pp_wgrad: Kj=%d must be a multiple of 8 and <= ldw", d.Kj);
  PP_CHECK_ARG(d.g.mode == PP_DENSE || d.g.mode == PP_CONV_FWD, "pp_wgrad: gather mode must be dense or conv-fwd");
  PP_CHECK_ARG(((uintptr_t)d.X & 15) == 0 && ((uintptr_t)d.dY & 15) == 0, "pp_wgrad: operands must be 16-byte aligned");
  if (d.nbatch <= 0) d.nbatch = 1;
  const int rc = pp_validate_gather(d.g, d.Kj, "pp_wgrad");
  if (rc != PP_OK) return rc;
  PP_CHECK_ARG((long long)d.M * d.ldy < 0x7fffffffLL, "pp_wgrad: dY has >= 2^31 elements");
  if (d.g.mode == PP_DENSE) {
    PP_CHECK_ARG((long long)d.M * d.g.lda < 0x7fffffffLL, "pp_wgrad: X has >= 2^31 elements");
  } else {
    const long long rows = (long long)d.g.Rt * d.g.Rh * d.g.Rw;
    PP_CHECK_ARG(d.M % rows == 0 && (d.M / rows) * d.g.Gt * d.g.Gh * d.g.Gw * d.g.cstride < 0x7fffffffLL,
                 "pp_wgrad: gathered tensor >= 2^31 elements or M not a multiple of Rt*Rh*Rw");
  }
  hipStream_t s = (hipStream_t)stream;
  switch (pick_wi((d.Ni + 15) / 16)) {
    case 15: return launch_wi<15>(d, s);
    case 9: return launch_wi<9>(d, s);
    case 8: return launch_wi<8>(d, s);
    case 4: return launch_wi<4>(d, s);
    case 3: return launch_wi<3>(d, s);
    default: return launch_wi<2>(d, s);
  }
}
