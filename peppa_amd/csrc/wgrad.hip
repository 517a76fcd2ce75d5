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
extern int pp_opt_xcd_remap_wgrad;

namespace {

constexpr int TJ = 128;   // j extent per workgroup (8 MFMA tiles, 2 per wave)
constexpr int MS = 64;    // m rows per step (two 32-deep MFMA sub-steps)
constexpr int QS = 288;   // Q row stride in bytes (256 + 32: 32*odd -> conflict-free tr reads)
constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer offset that is always out of range -> loads zeros

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

template <int WI, int MODE, bool BIAS>
__global__ __launch_bounds__(256, (WI <= 9 ? 2 : 1)) void wgrad_kernel(const pp_wgrad_desc p, const WGeom wg,
                                                                        const int nblk_i, const int nblk_j,
                                                                        const int rows_per_split, const int xcd_remap) {
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;  // P row stride (bytes), 32*odd
  constexpr int P_BYTES = MS * PS;
  constexpr int Q_BYTES = MS * QS;
  constexpr int BUF = P_BYTES + Q_BYTES;
  constexpr int NQI = 4;                                // Q chunks per thread and step
  constexpr int NPI = (WI * 128 + 255) / 256;           // P chunks per thread and step
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];
  __shared__ int lut[128];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order (see igemm.hip): the (i, j) tiles of one M-split read the same dY / X rows, so keep
  // consecutive tile indices on one XCD's L2
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  const int ib = bid % nblk_i; bid /= nblk_i;
  const int jb = bid % nblk_j; bid /= nblk_j;
  const int split = bid;
  const int z = blockIdx.z;
  const bfraw* X = (const bfraw*)p.X + z * p.x_s;
  const bfraw* dY = (const bfraw*)p.dY + z * p.dy_s;
  float* __restrict__ dW = p.dW + z * p.dw_s;
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);
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

  // Q (gathered X) chunks owned by this thread: rows (tid>>4) + 16 i, columns j0 + 8*(tid&15)
  const int qrow = tid >> 4, qch = tid & 15;
  const int jq = j0 + qch * 8;
  const bool jq_ok = jq < p.Kj;
  int qt0 = 0, qh0 = 0, qw0 = 0, cch = 0;
  if (MODE != PP_DENSE && jq_ok) {
    const int tap = jq / g.cg;
    cch = jq % g.cg;
    const int e = lut[tap];
    qt0 = (e & 0xff) - g.pt; qh0 = ((e >> 8) & 0xff) - g.ph; qw0 = ((e >> 16) & 0xff) - g.pw;
  }
  auto q_offset = [&](int m) __attribute__((always_inline)) -> unsigned {
    if (!jq_ok || m >= m_end) return OOB;
    if (MODE == PP_DENSE) return (unsigned)(m * g.lda + jq) * 2u;
    const uint32_t t1 = fdiv((uint32_t)m, wg.dRw);
    const int rw = m - (int)t1 * g.Rw;
    const uint32_t t2 = fdiv(t1, wg.dRh);
    const int rh = (int)t1 - (int)t2 * g.Rh;
    const int n = (int)fdiv(t2, wg.dRt);
    const int rt = (int)t2 - n * g.Rt;
    const int gt = rt * g.st + qt0, gh = rh * g.sh + qh0, gw = rw * g.sw + qw0;
    const bool ok = (unsigned)gt < (unsigned)g.Gt && (unsigned)gh < (unsigned)g.Gh && (unsigned)gw < (unsigned)g.Gw;
    return ok ? (unsigned)((((n * g.Gt + gt) * g.Gh + gh) * g.Gw + gw) * g.cstride + cch) * 2u : OOB;
  };
  // P (dY) chunks: cid = tid + 256 it -> row cid / (2 WI), chunk cid % (2 WI)
  int prow[NPI], pch[NPI];
#pragma unroll
  for (int it = 0; it < NPI; ++it) {
    const int cid = tid + 256 * it;
    prow[it] = cid / (2 * WI);
    pch[it] = cid % (2 * WI);
  }

  f32x4 acc[WI][2];
#pragma unroll
  for (int a = 0; a < WI; ++a) acc[a][0] = acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // optional bias gradient (column sums of dY), taken from the P tiles this block stages anyway
  const bool do_bias = BIAS && jb == 0;
  float bsum[BIAS ? NPI : 1][8];
#pragma unroll
  for (int it = 0; it < (BIAS ? NPI : 1); ++it)
#pragma unroll
    for (int q = 0; q < 8; ++q) bsum[it][q] = 0.f;

  u32x4 rq[NQI], rp[NPI];
  auto load_stage = [&](int mbase) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NQI; ++i) rq[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, q_offset(mbase + qrow + 16 * i), 0, 0);
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      const int m = mbase + prow[it];
      const int i = i0 + pch[it] * 8;
      const bool ok = prow[it] < MS && m < m_end && i < p.ldy;
      rp[it] = __builtin_amdgcn_raw_buffer_load_b128(rsY, ok ? (unsigned)(m * p.ldy + i) * 2u : OOB, 0, 0);
    }
  };
  auto store_stage = [&](unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NQI; ++i) *(u32x4*)(buf + P_BYTES + (qrow + 16 * i) * QS + qch * 16) = rq[i];
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      if (prow[it] < MS) *(u32x4*)(buf + prow[it] * PS + pch[it] * 16) = rp[it];
      if (BIAS && do_bias) {
        float f[8];
        unpack8(make_uint4(rp[it][0], rp[it][1], rp[it][2], rp[it][3]), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) bsum[BIAS ? it : 0][q] += f[q];
      }
    }
  };
  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned char* Pt = buf + sub * 32 * PS;
      const unsigned char* Qt = buf + P_BYTES + sub * 32 * QS;
      const bf16x8 b0 = tr_frag(Qt, QS, (2 * wave) * 16, lane);
      const bf16x8 b1 = tr_frag(Qt, QS, (2 * wave + 1) * 16, lane);
#pragma unroll
      for (int a = 0; a < WI; ++a) {
        const bf16x8 af = tr_frag(Pt, PS, a * 16, lane);
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b0, acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b1, acc[a][1], 0, 0, 0);
        // keep the scheduler from hoisting every fragment read to the top (register pressure)
        if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // one register stage of loads in flight under the MFMAs, two LDS buffers, one barrier per 64-row step
  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps > 0) {
    load_stage(m_begin);
    store_stage(smem);
  }
  __syncthreads();
  int cur = 0;
  for (int st = 0; st < nsteps; ++st) {
    const bool more = st + 1 < nsteps;
    if (more) load_stage(m_begin + (st + 1) * MS);
    compute(smem + cur * BUF);
    cur ^= 1;
    if (more) store_stage(smem + cur * BUF);
    __syncthreads();
  }

  if (BIAS) {
    // bias gradient: combine the block's row-threads in LDS first, then ONE global atomic per column
    float* bred = (float*)smem;   // the loop buffers are free after the final barrier
    for (int i = tid; i < TI; i += 256) bred[i] = 0.f;
    __syncthreads();
    if (do_bias) {
#pragma unroll
      for (int it = 0; it < NPI; ++it)
        if (prow[it] < MS)
#pragma unroll
          for (int q = 0; q < 8; ++q) atomicAdd(bred + pch[it] * 8 + q, bsum[BIAS ? it : 0][q]);
    }
    __syncthreads();
    if (do_bias)
      for (int i = tid; i < TI; i += 256)
        if (i0 + i < p.Ni) atomicAdd(p.dbias + z * p.dbias_s + i0 + i, bred[i]);
    __syncthreads();
  }
  // Stage the fp32 tile in LDS (row = i, 128 j columns) so that every atomic wave-instruction adds 256
  // contiguous bytes: the chip-wide float-atomic rate needs 128-256 B segments (MI355X_MICROARCH.md).
  const int fr = lane & 15, fq = lane >> 4;
  float* tile = (float*)smem;
  constexpr int CH = (2 * BUF) / (16 * TJ * 4) < WI ? (2 * BUF) / (16 * TJ * 4) : WI;  // i-tiles staged per pass
#pragma unroll
  for (int a0 = 0; a0 < WI; a0 += CH) {
    if (a0 > 0) __syncthreads();
#pragma unroll
    for (int a = a0; a < a0 + CH && a < WI; ++a)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[((a - a0) * 16 + fq * 4 + r) * TJ + (2 * wave + jj) * 16 + fr] = acc[a][jj][r];
    __syncthreads();
    const int nrows = (WI - a0 < CH ? WI - a0 : CH) * 16;
    // splits of one tile finish together: start each split at a different row so they do not queue up
    // on the same 256-byte lines at the memory-side atomic units
    const int nq = nrows / 4;
    for (int k = 0; k < nq; ++k) {
      const int row = ((k + split * 7) % nq) * 4 + wave;
      const int i = i0 + a0 * 16 + row;
      if (i >= p.Ni) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = j0 + h * 64 + lane;
        if (j < p.Kj) atomicAdd(dW + (long long)i * p.ldw + j, tile[row * TJ + h * 64 + lane]);
      }
    }
  }
}

int pick_wi(int n16) {
  static const int cand[] = {15, 9, 8, 4, 3, 2};
  static const float eff[] = {1.0f, 0.96f, 0.97f, 0.8f, 0.7f, 0.55f};  // ties between 9 and 8 go to 8 (fewer registers)
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
    // ~384 workgroups (1.5 per CU) with at least 24 steps per split: measured optimum for the M = 7296
    // transformer GEMMs; every split adds one fp32 atomic per output element
    const long long tiles = (long long)nblk_i * nblk_j * d.nbatch;
    long long want = (384 + tiles / 2) / tiles;
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
  if (d.g.mode == PP_DENSE) {
    if (d.dbias) hipLaunchKernelGGL((wgrad_kernel<WI, PP_DENSE, true>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad);
    else hipLaunchKernelGGL((wgrad_kernel<WI, PP_DENSE, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad);
  } else {
    hipLaunchKernelGGL((wgrad_kernel<WI, PP_CONV_FWD, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad);
  }
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
  PP_CHECK_ARG(!d.dbias || d.g.mode == PP_DENSE, "pp_wgrad: the fused bias gradient is only built for dense operands");
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
