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
extern int pp_opt_wgrad_flat;
extern int pp_opt_wgrad_group_ring;
extern int pp_opt_wgrad_big;
extern int pp_opt_ring_wgrad;
extern int pp_opt_sw_wgrad;
extern int pp_opt_deterministic;
int pp_wgrad_sw_try(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query);
int pp_wgrad_tw_try(const pp_wgrad_desc& d, hipStream_t s, bool force, long long* ws_query);
bool pp_wgrad_tw_ok(const pp_wgrad_desc& d, bool force);

namespace {

constexpr int TJ = 128;   // j extent per workgroup (8 MFMA tiles, 2 per wave)
constexpr int MS = 64;    // m rows per step (two 32-deep MFMA sub-steps)
constexpr int QS = 288;   // Q row stride in bytes (256 + 32: 32*odd -> conflict-free tr reads)
constexpr unsigned OOB = 0xFFFFFFF0u;  // buffer offset that is always out of range -> loads zeros

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WGeom {
  FastDiv dRw, dRh, dRt;
};


__device__ __forceinline__ h16x8 tr_frag(const unsigned char* tile, int stride, int col0, int lane) {
  // lanes 16g..16g+15 fetch rows {4g..4g+3} and {16+4g..16+4g+3} of columns col0..col0+15;
  // lane (4q+p) supplies the address of row q, columns 4p..4p+3 and receives column (lane&15).
  const int gq = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  const unsigned char* a0 = tile + (4 * gq + q) * stride + (col0 + 4 * pp) * 2;
  const unsigned char* a1 = a0 + 16 * stride;
  const h16x4 lo = ds_read_tr16(a0);
  const h16x4 hi = ds_read_tr16(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int WI, int MODE, bool BIAS>
__global__ __launch_bounds__(256, (WI <= 9 ? 2 : 1)) void wgrad_kernel(const pp_wgrad_desc p, const WGeom wg,
                                                                        const int nblk_i, const int nblk_j,
                                                                        const int rows_per_split, const int xcd_remap,
                                                                        const int flat) {
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;  // P row stride (bytes), 32*odd
  constexpr int P_BYTES = MS * PS;
  constexpr int Q_BYTES = MS * QS;
  constexpr int BUF = P_BYTES + Q_BYTES;
  constexpr int NQI = 4;                                // Q chunks per thread and step
  constexpr int NPI = (WI * 128 + 255) / 256;           // P chunks per thread and step
  // one LDS object: two loop buffers, the tap table and the four-deep row table (see decode_rows)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + 1024 + 4 * MS * 8];
  int* const lut = (int*)(smem + 2 * BUF);
  int2* const rowtab = (int2*)(smem + 2 * BUF + 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order (see igemm.hip): the (i, j) tiles of one M-split read the same dY / X rows, so keep
  // consecutive tile indices on one XCD's L2
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  // `flat` (grouped launches without an M split): the grid is ONE dimension over (problem, tile) and an XCD walks a
  // contiguous run of it in which the index of the SHORTER tile axis runs fastest.  A tile streams two M x 128 slabs (dY
  // columns of its ib, X columns of its jb); tiles that run together on one XCD share them through its L2.  With ib
  // fastest and one problem's tiles dealt over all eight XCDs (the order below, round 3) every dY slab of ffn1 (24 x 6
  // tiles) was fetched by six XCDs: 3.64 GB for 0.79 GB of operands (profiles/r03_pmc_traffic.md).  Here the tiles that share
  // the slab of the LONGER axis (the big operand) are neighbours on one XCD and fetch it once; the slabs of the shorter axis
  // (the small operand) are re-fetched once per round of resident workgroups.
  int ib, jb, split, z;
  if (flat) {
    const int tiles = nblk_i * nblk_j;
    z = bid / tiles;
    const int r = bid - z * tiles;
    if (nblk_j <= nblk_i) { jb = r % nblk_j; ib = r / nblk_j; } else { ib = r % nblk_i; jb = r / nblk_i; }
    split = 0;
  } else {
    ib = bid % nblk_i; bid /= nblk_i;
    jb = bid % nblk_j; bid /= nblk_j;
    split = bid;
    z = blockIdx.z;
  }
  const h16raw* X = (const h16raw*)p.X + z * p.x_s;
  const h16raw* dY = (const h16raw*)p.dY + z * p.dy_s;
  float* __restrict__ dW = p.dW + z * p.dw_s;
  float* dbias_z = BIAS ? p.dbias + z * p.dbias_s : nullptr;
  // deterministic mode: this (problem, split)'s slab of p.ws -- Ni rows of ldw floats, then Ni bias floats
  const long long slab_floats = (long long)p.Ni * p.ldw + p.Ni;
  const long long slab_w = ((long long)z * (flat ? 1 : gridDim.x / (nblk_i * nblk_j)) + split) * slab_floats;
  const long long slab_b = slab_w + (long long)p.Ni * p.ldw;
  if (p.ptr_table) {      // grouped launch: problem z has its own operands (uniform scalar loads)
    const unsigned long long* e = p.ptr_table + 4 * z;
    X = (const h16raw*)e[0];
    dY = (const h16raw*)e[1];
    dW = (float*)e[2];
    dbias_z = BIAS ? (float*)e[3] : nullptr;
  }
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);
  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    for (int tp = tid; tp < 256; tp += blockDim.x) {   // up to 256 taps (r3d_18's (3,7,7) stem has 147)
      int e = 0;
      if (tp < ntaps) {
        const int dw = tp % g.kw;
        const int t2 = tp / g.kw;
        e = (t2 / g.kh) | ((t2 % g.kh) << 8) | (dw << 16);
      }
      lut[tp] = e;
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
  int qt0 = 0, qh0 = 0, qw0 = 0, cch = 0, qtap = 0;
  unsigned tapoff = 0;   // byte offset of this thread's (tap, channel chunk) relative to a row's tap (0,0,0)
  if (MODE != PP_DENSE && jq_ok) {
    qtap = jq / g.cg;
    cch = jq % g.cg;
    const int e = lut[qtap];
    const int dt = e & 0xff, dh = (e >> 8) & 0xff, dw = (e >> 16) & 0xff;
    qt0 = dt - g.pt; qh0 = dh - g.ph; qw0 = dw - g.pw;
    tapoff = (unsigned)((((dt * g.Gh + dh) * g.Gw + dw) * g.cstride + cch) * 2);
  }
  // Row table (conv gathers with <= 32 taps): the address arithmetic of a row -- three exact divisions, the origin
  // offset and the validity of every tap -- is done ONCE per row and step by one lane and shared through LDS,
  // instead of by each of the 16 lanes that fetch the row's chunks.  Entry = {byte offset of tap (0,0,0) (may be
  // "virtual", i.e. outside the tensor), bit t set <=> tap t lies inside the tensor}.
  const bool use_tab = MODE != PP_DENSE && ntaps <= 32;
  auto decode_rows = [&](const int mbase, const int par) __attribute__((always_inline)) {
    const int m = mbase + lane;
    int base = 0;
    unsigned mask = 0;
    if (m < m_end) {
      const uint32_t t1 = fdiv((uint32_t)m, wg.dRw);
      const int rw = m - (int)t1 * g.Rw;
      const uint32_t t2 = fdiv(t1, wg.dRh);
      const int rh = (int)t1 - (int)t2 * g.Rh;
      const int n = (int)fdiv(t2, wg.dRt);
      const int rt = (int)t2 - n * g.Rt;
      const int ct = rt * g.st - g.pt, chh = rh * g.sh - g.ph, cw = rw * g.sw - g.pw;
      base = ((((n * g.Gt + ct) * g.Gh + chh) * g.Gw + cw) * g.cstride) * 2;
      unsigned vw = 0, mhw = 0;
      for (int d = 0; d < g.kw; ++d) vw |= (unsigned)((unsigned)(cw + d) < (unsigned)g.Gw) << d;
      for (int d = 0; d < g.kh; ++d) mhw |= ((unsigned)(chh + d) < (unsigned)g.Gh) ? vw << (d * g.kw) : 0u;
      for (int d = 0; d < g.kt; ++d) mask |= ((unsigned)(ct + d) < (unsigned)g.Gt) ? mhw << (d * g.kh * g.kw) : 0u;
    }
    rowtab[par * MS + lane] = make_int2(base, (int)mask);
  };
  auto q_offset = [&](int m) __attribute__((always_inline)) -> unsigned {   // direct form (dense, or > 32 taps)
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

  // Two register sets: the loads of steps s + 1 and s + 2 are in flight while step s is multiplied (an HBM miss
  // costs ~2 us here, several 64-row steps).  `set` is a literal at every call site.
  u32x4 rq[2][NQI], rp[2][NPI];
  auto load_stage = [&](const int set, const int step) __attribute__((always_inline)) {
    const int mbase = m_begin + step * MS;
    if (use_tab) {
      const int par = step & 3;
#pragma unroll
      for (int i = 0; i < NQI; ++i) {
        const int2 e = rowtab[par * MS + qrow + 16 * i];   // (rows past m_end carry an empty mask)
        const bool ok = jq_ok && (((unsigned)e.y >> qtap) & 1u);
        rq[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? (unsigned)e.x + tapoff : OOB, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NQI; ++i) rq[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, q_offset(mbase + qrow + 16 * i), 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      const int m = mbase + prow[it];
      const int i = i0 + pch[it] * 8;
      const bool ok = prow[it] < MS && m < m_end && i < p.ldy;
      rp[set][it] = __builtin_amdgcn_raw_buffer_load_b128(rsY, ok ? (unsigned)(m * p.ldy + i) * 2u : OOB, 0, 0);
    }
  };
  auto store_stage = [&](const int set, unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NQI; ++i) *(u32x4*)(buf + P_BYTES + (qrow + 16 * i) * QS + qch * 16) = rq[set][i];
#pragma unroll
    for (int it = 0; it < NPI; ++it) {
      if (prow[it] < MS) *(u32x4*)(buf + prow[it] * PS + pch[it] * 16) = rp[set][it];
      if (BIAS && do_bias) {
        float f[8];
        unpack8(make_uint4(rp[set][it][0], rp[set][it][1], rp[set][it][2], rp[set][it][3]), f);
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
      const h16x8 b0 = tr_frag(Qt, QS, (2 * wave) * 16, lane);
      const h16x8 b1 = tr_frag(Qt, QS, (2 * wave + 1) * 16, lane);
#pragma unroll
      for (int a = 0; a < WI; ++a) {
        const h16x8 af = tr_frag(Pt, PS, a * 16, lane);
        acc[a][0] = PP_MFMA16(af, b0, acc[a][0], 0, 0, 0);
        acc[a][1] = PP_MFMA16(af, b1, acc[a][1], 0, 0, 0);
        // keep the scheduler from hoisting every fragment read to the top (register pressure)
        if ((a & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // one register stage of loads in flight under the MFMAs, two LDS buffers, one barrier per 64-row step
  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (use_tab) {   // row tables of steps 0..3 (step s lives in table s & 3)
    decode_rows(m_begin + wave * MS, wave);
    __syncthreads();
  }
  // Pipeline: step s is loaded into register set s & 1 three iterations ahead, written to LDS buffer s & 1 after step
  // s - 1 has been multiplied, multiplied one iteration later.  (Loads past the last step fetch nothing: every lane
  // is out of range.)
  load_stage(0, 0);
  store_stage(0, smem);
  load_stage(1, 1);
  load_stage(0, 2);
  __syncthreads();
  auto iteration = [&](const int st, const int set, unsigned char* cur, unsigned char* nxt) __attribute__((always_inline)) {
    // the row table of step st (read three iterations ago) is free: refill it for step st + 4
    if (use_tab && wave == (st & 3)) decode_rows(m_begin + (st + 4) * MS, st & 3);
    compute(cur);
    store_stage(set, nxt);          // step st + 1: waits for its own loads only
    load_stage(set, st + 3);        // ... and its registers go straight back into flight
    __syncthreads();
  };
  for (int st = 0; st < nsteps; st += 2) {
    iteration(st, 1, smem, smem + BUF);
    if (st + 1 < nsteps) iteration(st + 1, 0, smem + BUF, smem);
  }
  if (BIAS) {
    // bias gradient: the block's row-threads meet in LDS and are summed IN ROW ORDER (no LDS float atomics: their
    // order changes from run to run), then ONE global add per column -- with msplit = 1 its only adder
    float* part = (float*)smem;   // [cid = row * 2 WI + chunk][8]; the loop buffers are free after the final barrier
    static_assert(NPI * 256 * 8 * 4 <= 2 * BUF, "bias partials fit the loop buffers");
    if (do_bias) {
#pragma unroll
      for (int it = 0; it < NPI; ++it)
#pragma unroll
        for (int q = 0; q < 8; ++q) part[(tid + 256 * it) * 8 + q] = bsum[BIAS ? it : 0][q];
    }
    __syncthreads();
    if (do_bias)
      for (int i = tid; i < TI; i += 256) {
        float v = 0.f;
        for (int r = 0; r < MS; ++r) v += part[(r * 2 * WI + (i >> 3)) * 8 + (i & 7)];
        if (i0 + i < p.Ni) {
          if (p.ws) p.ws[slab_b + i0 + i] = v;            // deterministic: this split's slab, summed in split order later
          else atomicAdd(dbias_z + i0 + i, v);
        }
      }
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
        if (j < p.Kj) {
          if (p.ws) p.ws[slab_w + (long long)i * p.ldw + j] = tile[row * TJ + h * 64 + lane];
          else atomicAdd(dW + (long long)i * p.ldw + j, tile[row * TJ + h * 64 + lane]);
        }
      }
    }
  }
}

// ---- 256 x 256 tiles (round 4) -------------------------------------------------------------------------------------------
// Every GEMM-shaped kernel of this library delivers 8-10 TB/s from the L2 to its CUs, whatever it does with the bytes
// (DESIGN.md section 5): the 128 x 128 tile above asks for M * (Ni * Kj / 128 + Kj * Ni / 128) * 2 bytes, a 256 x 256 tile for
// half of that.  Eight waves as a 2 x 4 grid of 128 x 64 wave tiles (32 MFMAs per 12 fragment reads), one workgroup per CU,
// the same register-staged pipeline (two register sets in flight, two LDS buffers, one barrier per 64-row step).  No fused
// bias gradient (the caller's launcher keeps the 128 x 128 kernel for that), no M split slabs (deterministic mode likewise).
constexpr int BTI = 256, BTJ = 256;
constexpr int BPS = BTI * 2 + 32, BQS = BTJ * 2 + 32;     // row strides in bytes: 32 x odd (conflict-free transposing reads)

template <int MODE, bool BIAS>
__global__ __launch_bounds__(512, 1) void wgrad_big_kernel(const pp_wgrad_desc p, const WGeom wg, const int nblk_i, const int nblk_j,
                                                           const int rows_per_split, const int xcd_remap, const int flat) {
  constexpr int P_BYTES = MS * BPS, Q_BYTES = MS * BQS, BUF = P_BYTES + Q_BYTES;
  constexpr int NCH = 4;                                  // 16-byte chunks per thread, operand and step (64 rows x 32 chunks / 512)
  constexpr int BIAS_BYTES = BIAS ? 512 * 8 * 4 : 0;      // the bias gradient's running sums: eight floats per thread (below)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + 1024 + 4 * MS * 8 + BIAS_BYTES];
  static_assert(2 * BUF + 1024 + 4 * MS * 8 + BIAS_BYTES <= 160 * 1024, "LDS budget");
  int* const lut = (int*)(smem + 2 * BUF);
  int2* const rowtab = (int2*)(smem + 2 * BUF + 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  int ib, jb, split, z;
  if (flat) {          // grouped launch, no M split: (problem, tile) in slab-sharing order (see wgrad_kernel)
    const int tiles = nblk_i * nblk_j;
    z = bid / tiles;
    const int r = bid - z * tiles;
    if (nblk_j <= nblk_i) { jb = r % nblk_j; ib = r / nblk_j; } else { ib = r % nblk_i; jb = r / nblk_i; }
    split = 0;
  } else {
    ib = bid % nblk_i; bid /= nblk_i;
    jb = bid % nblk_j; bid /= nblk_j;
    split = bid;
    z = blockIdx.z;
  }
  const h16raw* X = (const h16raw*)p.X + z * p.x_s;
  const h16raw* dY = (const h16raw*)p.dY + z * p.dy_s;
  float* __restrict__ dW = p.dW + z * p.dw_s;
  float* dbias_z = BIAS ? p.dbias + z * p.dbias_s : nullptr;
  if (p.ptr_table) {
    const unsigned long long* e = p.ptr_table + 4 * z;
    X = (const h16raw*)e[0];
    dY = (const h16raw*)e[1];
    dW = (float*)e[2];
    dbias_z = BIAS ? (float*)e[3] : nullptr;
  }
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);
  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    for (int tp = tid; tp < 256; tp += blockDim.x) {
      int e = 0;
      if (tp < ntaps) {
        const int dw = tp % g.kw;
        const int t2 = tp / g.kw;
        e = (t2 / g.kh) | ((t2 % g.kh) << 8) | (dw << 16);
      }
      lut[tp] = e;
    }
    __syncthreads();
  }
  const int i0 = ib * BTI, j0 = jb * BTJ;
  const int m_begin = split * rows_per_split;
  const int m_end = min(p.M, m_begin + rows_per_split);

  // chunks owned by this thread: rows (tid >> 5) + 16 i, chunk column tid & 31 -- of Q (gathered X, columns j0 + 8 ch) and of P (dY)
  const int crow = tid >> 5, cch = tid & 31;
  const int jq = j0 + cch * 8;
  const bool jq_ok = jq < p.Kj;
  const int ip = i0 + cch * 8;
  const bool ip_ok = ip < p.ldy;
  int qtap = 0;
  unsigned tapoff = 0;
  if (MODE != PP_DENSE && jq_ok) {
    qtap = jq / g.cg;
    const int cc = jq % g.cg;
    const int e = lut[qtap];
    const int dt = e & 0xff, dh = (e >> 8) & 0xff, dw = (e >> 16) & 0xff;
    tapoff = (unsigned)((((dt * g.Gh + dh) * g.Gw + dw) * g.cstride + cc) * 2);
  }
  // row table (conv gathers; the launcher only sends <= 32 taps here): see wgrad_kernel
  auto decode_rows = [&](const int mbase, const int par) __attribute__((always_inline)) {
    const int m = mbase + lane;
    int base = 0;
    unsigned mask = 0;
    if (m < m_end) {
      const uint32_t t1 = fdiv((uint32_t)m, wg.dRw);
      const int rw = m - (int)t1 * g.Rw;
      const uint32_t t2 = fdiv(t1, wg.dRh);
      const int rh = (int)t1 - (int)t2 * g.Rh;
      const int n = (int)fdiv(t2, wg.dRt);
      const int rt = (int)t2 - n * g.Rt;
      const int ct = rt * g.st - g.pt, chh = rh * g.sh - g.ph, cw = rw * g.sw - g.pw;
      base = ((((n * g.Gt + ct) * g.Gh + chh) * g.Gw + cw) * g.cstride) * 2;
      unsigned vw = 0, mhw = 0;
      for (int d = 0; d < g.kw; ++d) vw |= (unsigned)((unsigned)(cw + d) < (unsigned)g.Gw) << d;
      for (int d = 0; d < g.kh; ++d) mhw |= ((unsigned)(chh + d) < (unsigned)g.Gh) ? vw << (d * g.kw) : 0u;
      for (int d = 0; d < g.kt; ++d) mask |= ((unsigned)(ct + d) < (unsigned)g.Gt) ? mhw << (d * g.kh * g.kw) : 0u;
    }
    rowtab[par * MS + lane] = make_int2(base, (int)mask);
  };

  const int wi = wave >> 2, wj = wave & 3;       // 2 x 4 wave grid: rows [128 wi, +128) x columns [64 wj, +64) of the tile
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // optional bias gradient (column sums of dY) from the P chunks this thread stages anyway: its chunk column is the same on
  // every row, so eight running sums do; the workgroups of column block 0 own it.  The sums live in the thread's OWN 32 bytes
  // of LDS, not in registers: eight more live registers across the K loop spill (254 + 8) and cost every workgroup of the
  // launch 20 % (profiles/r04_probe_wgrad_big.log)
  const bool do_bias = BIAS && jb == 0;
  float* const bmine = (float*)(smem + 2 * BUF + 1024 + 4 * MS * 8) + tid * 8;
  if (BIAS && do_bias) {
    *(float4*)bmine = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(bmine + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  u32x4 rq[2][NCH], rp[2][NCH];
  auto load_stage = [&](const int set, const int step) __attribute__((always_inline)) {
    const int mbase = m_begin + step * MS;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int row = crow + 16 * i;
      unsigned off;
      if (MODE == PP_DENSE) {
        const int m = mbase + row;
        off = (jq_ok && m < m_end) ? (unsigned)(m * g.lda + jq) * 2u : OOB;
      } else {
        const int2 e = rowtab[(step & 3) * MS + row];          // (rows past m_end carry an empty mask)
        off = (jq_ok && (((unsigned)e.y >> qtap) & 1u)) ? (unsigned)e.x + tapoff : OOB;
      }
      rq[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int m = mbase + crow + 16 * i;
      rp[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (ip_ok && m < m_end) ? (unsigned)(m * p.ldy + ip) * 2u : OOB, 0, 0);
    }
  };
  auto store_stage = [&](const int set, unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) *(u32x4*)(buf + (crow + 16 * i) * BPS + cch * 16) = rp[set][i];
    if (BIAS && do_bias) {
      float t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {          // (rows in ascending order: a fixed summation order)
        float f[8];
        unpack8(make_uint4(rp[set][i][0], rp[set][i][1], rp[set][i][2], rp[set][i][3]), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] += f[q];
      }
      float4 lo = *(float4*)bmine, hi = *(float4*)(bmine + 4);
      lo.x += t[0]; lo.y += t[1]; lo.z += t[2]; lo.w += t[3];
      hi.x += t[4]; hi.y += t[5]; hi.z += t[6]; hi.w += t[7];
      *(float4*)bmine = lo;
      *(float4*)(bmine + 4) = hi;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) *(u32x4*)(buf + P_BYTES + (crow + 16 * i) * BQS + cch * 16) = rq[set][i];
  };
  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned char* Pt = buf + sub * 32 * BPS;
      const unsigned char* Qt = buf + P_BYTES + sub * 32 * BQS;
      h16x8 bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = tr_frag(Qt, BQS, (wj * 4 + b) * 16, lane);
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const h16x8 af = tr_frag(Pt, BPS, (wi * 8 + a) * 16, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = PP_MFMA16(af, bf[b], acc[a][b], 0, 0, 0);
        if ((a & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // (keep the fragment reads from piling up at the top)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  const bool use_tab = MODE != PP_DENSE;
  if (use_tab) {
    if (wave < 4) decode_rows(m_begin + wave * MS, wave);
    __syncthreads();
  }
  load_stage(0, 0);
  store_stage(0, smem);
  load_stage(1, 1);
  load_stage(0, 2);
  __syncthreads();
  auto iteration = [&](const int st, const int set, unsigned char* cur, unsigned char* nxt) __attribute__((always_inline)) {
    if (use_tab && wave == (st & 3)) decode_rows(m_begin + (st + 4) * MS, st & 3);
    compute(cur);
    store_stage(set, nxt);
    load_stage(set, st + 3);
    __syncthreads();
  };
  for (int st = 0; st < nsteps; st += 2) {
    iteration(st, 1, smem, smem + BUF);
    if (st + 1 < nsteps) iteration(st + 1, 0, smem + BUF, smem);
  }
  if (BIAS) {
    // the sixteen row-threads of a chunk column meet in LDS and are summed IN ROW ORDER, then one add per column
    const float* part = (const float*)(smem + 2 * BUF + 1024 + 4 * MS * 8);      // [crow][chunk column][8] = [tid][8]
    if (do_bias && tid < BTI) {       // (the loop's last barrier made every thread's sums visible)
      float v = 0.f;
      for (int r = 0; r < 16; ++r) v += part[(r * 32 + (tid >> 3)) * 8 + (tid & 7)];
      if (i0 + tid < p.Ni) atomicAdd(dbias_z + i0 + tid, v);
    }
    __syncthreads();
  }
  // the fp32 tile through LDS, 128 rows (one wave row) per pass, so that every atomic wave-instruction adds 256 contiguous bytes
  const int fr = lane & 15, fq = lane >> 4;
  float* tile = (float*)smem;
  static_assert(128 * BTJ * 4 <= 2 * BUF, "half a tile fits the loop buffers");
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass > 0) __syncthreads();
    if (wi == pass) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) tile[(a * 16 + fq * 4 + r) * BTJ + (wj * 4 + b) * 16 + fr] = acc[a][b][r];
    }
    __syncthreads();
    for (int k = 0; k < 16; ++k) {
      const int row = ((k + split * 7) & 15) * 8 + wave;       // splits start at different rows (see wgrad_kernel)
      const int i = i0 + pass * 128 + row;
      if (i >= p.Ni) continue;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int j = j0 + h * 64 + lane;
        if (j < p.Kj) atomicAdd(dW + (long long)i * p.ldw + j, tile[row * BTJ + h * 64 + lane]);
      }
    }
  }
}

// Deterministic mode: out[z][i][j] += sum over the splits, in split order, of the slabs the kernels above stored.
// Slab (z, split) = ws[(z * nsplit + split) * slab_floats ..]: Ni rows of ldw floats, then (bias) Ni floats.
__global__ __launch_bounds__(256) void wgrad_slab_sum_kernel(const float* __restrict__ ws, const int nsplit, const long long slab_floats,
                                                             const int Ni, const int Kj, const int ldw, float* dW,
                                                             const long long dw_s, float* dbias, const long long dbias_s,
                                                             const unsigned long long* __restrict__ ptr_table) {
  const int z = blockIdx.z;
  float* out = dW + z * dw_s;
  float* bout = dbias ? dbias + z * dbias_s : nullptr;
  if (ptr_table) {
    out = (float*)ptr_table[4 * z + 2];
    bout = dbias ? (float*)ptr_table[4 * z + 3] : nullptr;
  }
  const float* base = ws + (long long)z * nsplit * slab_floats;
  const long long n = (long long)Ni * Kj;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n + (bout ? Ni : 0); e += (long long)gridDim.x * 256) {
    long long off;
    float* dst;
    if (e < n) {
      const int i = (int)(e / Kj), j = (int)(e - (long long)i * Kj);
      off = (long long)i * ldw + j;
      dst = out + off;
    } else {
      off = (long long)Ni * ldw + (e - n);
      dst = bout + (e - n);
    }
    float sum = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) sum += base[sp * slab_floats + off];
    *dst += sum;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LDS-DMA ring variant: one persistent-size workgroup per CU (NWV waves, TJ = 32 * NWV columns of dW), operands go
// global -> LDS by buffer_load ... lds into a three-slot ring with two 64-row steps in flight across raw barriers
// (counted vmcnt), exactly as in igemm.hip's ring.  A DMA piece is 1 KiB, lane-linear, so a lane's (row, chunk) in a
// padded slab row comes from dividing its LDS offset by the row stride; pad lanes fetch OOB (zeros).
// The fused bias gradient needs no extra loads: the wave that owns j-tiles 0,1 multiplies every dY fragment with a
// fragment of ones on the matrix core (column 0 of that accumulator is the column sum).
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {   // n is wave-uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (conservative)
  }
}

template <int WI, int NWV, int MODE, bool BIAS>
__global__ __launch_bounds__(64 * NWV, 1) void wgrad_ring_kernel(const pp_wgrad_desc p, const WGeom wg, const int nblk_i,
                                                                   const int nblk_j, const int rows_per_split,
                                                                   const int xcd_remap, const int flat) {
  constexpr int NT = 64 * NWV;
  constexpr int TJR = 32 * NWV;                                   // columns of dW per workgroup
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;             // slab row strides: 32 bytes x odd
  constexpr int QSR = (NWV & 1) ? TJR * 2 : TJR * 2 + 32;
  constexpr int P_BYTES = MS * PS, Q_BYTES = MS * QSR;
  constexpr int SLOT = P_BYTES + Q_BYTES;
  constexpr int NPIECE_P = P_BYTES / 1024, NPIECE_Q = Q_BYTES / 1024;
  static_assert(P_BYTES % 1024 == 0 && Q_BYTES % 1024 == 0, "slabs are whole DMA pieces");
  constexpr int NPP = (NPIECE_P + NWV - 1) / NWV, NPQ = (NPIECE_Q + NWV - 1) / NWV;   // pieces per wave and step
  static_assert(3 * SLOT + 1024 <= 160 * 1024, "ring does not fit the LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SLOT + 1024];   // one LDS object (see igemm.hip)
  int* const lut = (int*)(smem + 3 * SLOT);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  int ib, jb, split;
  const h16raw* X = (const h16raw*)p.X;
  const h16raw* dY = (const h16raw*)p.dY;
  float* __restrict__ dW = p.dW;
  float* dbias_z = BIAS ? p.dbias : nullptr;
  if (flat) {      // grouped launch (pointer table), no M split: (problem, tile) in slab-sharing order, see wgrad_kernel
    const int tiles = nblk_i * nblk_j;
    const int z = bid / tiles;
    const int r = bid - z * tiles;
    if (nblk_j <= nblk_i) { jb = r % nblk_j; ib = r / nblk_j; } else { ib = r % nblk_i; jb = r / nblk_i; }
    split = 0;
    const unsigned long long* e = p.ptr_table + 4 * z;
    X = (const h16raw*)e[0];
    dY = (const h16raw*)e[1];
    dW = (float*)e[2];
    dbias_z = BIAS ? (float*)e[3] : nullptr;
  } else {
    ib = bid % nblk_i; bid /= nblk_i;
    jb = bid % nblk_j; bid /= nblk_j;
    split = bid;
  }
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);
  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    for (int tp = tid; tp < 256; tp += blockDim.x) {   // up to 256 taps (r3d_18's (3,7,7) stem has 147)
      int e = 0;
      if (tp < ntaps) {
        const int dw = tp % g.kw;
        const int t2 = tp / g.kw;
        e = (t2 / g.kh) | ((t2 % g.kh) << 8) | (dw << 16);
      }
      lut[tp] = e;
    }
    __syncthreads();
  }
  const int i0 = ib * TI, j0 = jb * TJR;
  const int m_begin = split * rows_per_split;
  const int m_end = min(p.M, m_begin + rows_per_split);

  // ---- this lane's slot in each DMA piece its wave issues -----------------------------------------------------
  int p_row[NPP];
  unsigned p_col[NPP];        // byte offset of the chunk inside a dY row, or OOB for pad / out-of-range columns
  int np_w = 0;               // pieces of P this wave really issues (wave-uniform)
#pragma unroll
  for (int it = 0; it < NPP; ++it) {
    const int piece = wave + NWV * it;
    const int o = piece * 1024 + lane * 16;
    p_row[it] = o / PS;
    const int cb = o % PS;
    const int i = i0 + cb / 2;
    p_col[it] = (cb < TI * 2 && i < p.ldy) ? (unsigned)i * 2u : OOB;
    if (piece < NPIECE_P) ++np_w;
  }
  int q_row[NPQ], q_t0[NPQ], q_h0[NPQ], q_w0[NPQ];
  unsigned q_col[NPQ];        // dense: byte offset inside an X row; conv: channel byte offset inside the tap; OOB = pad
  int nq_w = 0;
#pragma unroll
  for (int it = 0; it < NPQ; ++it) {
    const int piece = wave + NWV * it;
    const int o = piece * 1024 + lane * 16;
    q_row[it] = o / QSR;
    const int cb = o % QSR;
    const int jq = j0 + cb / 2;
    const bool ok = cb < TJR * 2 && jq < p.Kj;
    q_t0[it] = q_h0[it] = q_w0[it] = 0;
    if (MODE == PP_DENSE) {
      q_col[it] = ok ? (unsigned)jq * 2u : OOB;
    } else {
      q_col[it] = OOB;
      if (ok) {
        const int tap = jq / g.cg;
        q_col[it] = (unsigned)(jq - tap * g.cg) * 2u;
        const int e = lut[tap];
        q_t0[it] = (e & 0xff) - g.pt; q_h0[it] = ((e >> 8) & 0xff) - g.ph; q_w0[it] = ((e >> 16) & 0xff) - g.pw;
      }
    }
    if (piece < NPIECE_Q) ++nq_w;
  }
  const int per_step = __builtin_amdgcn_readfirstlane(np_w + nq_w);   // DMA instructions this wave issues per step

  auto dma_stage = [&](unsigned char* buf, const int mbase) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NPP; ++it) {
      if (it < np_w) {
        const int m = mbase + p_row[it];
        const unsigned off = (p_col[it] != OOB & m < m_end) ? (unsigned)(m * p.ldy) * 2u + p_col[it] : OOB;
        lds_dma16(rsY, buf + (wave + NWV * it) * 1024, off);
      }
    }
#pragma unroll
    for (int it = 0; it < NPQ; ++it) {
      if (it < nq_w) {
        const int m = mbase + q_row[it];
        const bool live = q_col[it] != OOB && m < m_end;
        unsigned off;   // (branch-free: everything is computed, then selected)
        if (MODE == PP_DENSE) {
          off = (unsigned)(m * g.lda) * 2u + q_col[it];
        } else {
          const uint32_t t1 = fdiv((uint32_t)m, wg.dRw);
          const int rw = m - (int)t1 * g.Rw;
          const uint32_t t2 = fdiv(t1, wg.dRh);
          const int rh = (int)t1 - (int)t2 * g.Rh;
          const int n = (int)fdiv(t2, wg.dRt);
          const int rt = (int)t2 - n * g.Rt;
          const int gt = rt * g.st + q_t0[it], gh = rh * g.sh + q_h0[it], gw = rw * g.sw + q_w0[it];
          const bool in = (unsigned)gt < (unsigned)g.Gt & (unsigned)gh < (unsigned)g.Gh & (unsigned)gw < (unsigned)g.Gw;
          off = (unsigned)((((n * g.Gt + gt) * g.Gh + gh) * g.Gw + gw) * g.cstride) * 2u + q_col[it];
          off = in ? off : OOB;
        }
        off = live ? off : OOB;
        lds_dma16(rsX, buf + P_BYTES + (wave + NWV * it) * 1024, off);
      }
    }
  };

  f32x4 acc[WI][2];
#pragma unroll
  for (int a = 0; a < WI; ++a) acc[a][0] = acc[a][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = BIAS && jb == 0 && wave == 0;
  f32x4 bacc[BIAS ? WI : 1];
#pragma unroll
  for (int a = 0; a < (BIAS ? WI : 1); ++a) bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h16x8 ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (h16)1.0f;

  // Fragment reads go through inline asm: hipcc drains vmcnt (every LDS-DMA in flight) before a
  // ds_read_b64_tr_b16 issued through the builtin, which would serialise the ring.  The asm reads are invisible to
  // the compiler's counters, hence the explicit lgkmcnt waits; the second half-step's reads are issued ahead of the
  // first half-step's MFMAs so their latency is covered.
  const int gq = lane >> 4, li = lane & 15;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned p_lane = (unsigned)((4 * gq + (li >> 2)) * PS + (li & 3) * 8);    // tr_frag's address pattern
  const unsigned q_lane = (unsigned)((4 * gq + (li >> 2)) * QSR + (li & 3) * 8 + (2 * wave) * 32);
  // A fragment = two 64-bit transposing reads (rows 0-15 / 16-31 of the half-step); the raw halves stay in their own
  // registers until the wait below has "modified" them, so nothing the compiler derives from them (the 4-register
  // MFMA operand tuples) can be scheduled ahead of the data's arrival.
  struct Frags { u32x2 ql[2], qh[2], pl[WI], ph[WI]; };
  auto read_frags = [&](const unsigned slot_off, const int sub, Frags& f) __attribute__((always_inline)) {
    const unsigned pa = lds0 + slot_off + (unsigned)(sub * 32 * PS) + p_lane;
    const unsigned qa = lds0 + slot_off + (unsigned)(P_BYTES + sub * 32 * QSR) + q_lane;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.ql[jj]) : "v"(qa + jj * 32) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.qh[jj]) : "v"(qa + jj * 32 + 16 * QSR) : "memory");
    }
#pragma unroll
    for (int a = 0; a < WI; ++a) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.pl[a]) : "v"(pa + a * 32) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.ph[a]) : "v"(pa + a * 32 + 16 * PS) : "memory");
    }
  };
  auto wait_frags = [&](Frags& f) __attribute__((always_inline)) {
    static_assert(WI == 8 || WI == 9, "operand list below");
    if constexpr (WI == 9) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.ql[0]), "+v"(f.qh[0]), "+v"(f.ql[1]), "+v"(f.qh[1]), "+v"(f.pl[0]), "+v"(f.ph[0]), "+v"(f.pl[1]),
                     "+v"(f.ph[1]), "+v"(f.pl[2]), "+v"(f.ph[2]), "+v"(f.pl[3]), "+v"(f.ph[3]), "+v"(f.pl[4]), "+v"(f.ph[4]),
                     "+v"(f.pl[5]), "+v"(f.ph[5]), "+v"(f.pl[6]), "+v"(f.ph[6]), "+v"(f.pl[7]), "+v"(f.ph[7]),
                     "+v"(f.pl[WI - 1]), "+v"(f.ph[WI - 1])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.ql[0]), "+v"(f.qh[0]), "+v"(f.ql[1]), "+v"(f.qh[1]), "+v"(f.pl[0]), "+v"(f.ph[0]), "+v"(f.pl[1]),
                     "+v"(f.ph[1]), "+v"(f.pl[2]), "+v"(f.ph[2]), "+v"(f.pl[3]), "+v"(f.ph[3]), "+v"(f.pl[4]), "+v"(f.ph[4]),
                     "+v"(f.pl[5]), "+v"(f.ph[5]), "+v"(f.pl[6]), "+v"(f.ph[6]), "+v"(f.pl[7]), "+v"(f.ph[7])
                   :
                   : "memory");
    }
  };
  auto mfma_frags = [&](const Frags& f) __attribute__((always_inline)) {
    h16x8 qf[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) qf[jj] = __builtin_bit_cast(h16x8, (u32x4){f.ql[jj][0], f.ql[jj][1], f.qh[jj][0], f.qh[jj][1]});
#pragma unroll
    for (int a = 0; a < WI; ++a) {
      const h16x8 pf = __builtin_bit_cast(h16x8, (u32x4){f.pl[a][0], f.pl[a][1], f.ph[a][0], f.ph[a][1]});
      acc[a][0] = PP_MFMA16(pf, qf[0], acc[a][0], 0, 0, 0);
      acc[a][1] = PP_MFMA16(pf, qf[1], acc[a][1], 0, 0, 0);
      if (BIAS && do_bias) bacc[a] = PP_MFMA16(pf, ones, bacc[a], 0, 0, 0);
    }
  };
  auto compute = [&](const unsigned slot_off) __attribute__((always_inline)) {
    Frags f0, f1;
    read_frags(slot_off, 0, f0);
    wait_frags(f0);
    read_frags(slot_off, 1, f1);   // in flight under the first half-step's MFMAs
    mfma_frags(f0);
    wait_frags(f1);
    mfma_frags(f1);
  };

  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps > 0) dma_stage(smem, m_begin);
  if (nsteps > 1) dma_stage(smem + SLOT, m_begin + MS);
  int sl = 0;
  for (int st = 0; st < nsteps; ++st) {
    wait_vmcnt_dyn(st + 1 < nsteps ? per_step : 0);   // this wave's pieces of step st have landed
    __builtin_amdgcn_s_barrier();                      // ... and everyone's; slot (st + 2) % 3 is free again
    if (st + 2 < nsteps) dma_stage(smem + (sl >= 1 ? sl - 1 : 2) * SLOT, m_begin + (st + 2) * MS);
    compute((unsigned)(sl * SLOT));
    sl = sl == 2 ? 0 : sl + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  if (BIAS && do_bias && fr == 0) {   // column 0 of each bias accumulator holds the column sums of dY
#pragma unroll
    for (int a = 0; a < WI; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + a * 16 + fq * 4 + r;
        if (i < p.Ni) atomicAdd(dbias_z + i, bacc[a][r]);
      }
  }
  // fp32 tile through LDS so that every atomic wave-instruction adds 256 contiguous bytes
  float* tile = (float*)smem;
  constexpr int CHMAX = (3 * SLOT) / (16 * TJR * 4);
  constexpr int CH = CHMAX < WI ? CHMAX : WI;
#pragma unroll
  for (int a0 = 0; a0 < WI; a0 += CH) {
    if (a0 > 0) __syncthreads();
#pragma unroll
    for (int a = a0; a < a0 + CH && a < WI; ++a)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[((a - a0) * 16 + fq * 4 + r) * TJR + (2 * wave + jj) * 16 + fr] = acc[a][jj][r];
    __syncthreads();
    const int nrows = (WI - a0 < CH ? WI - a0 : CH) * 16;
    const int nq = (nrows + NWV - 1) / NWV;
    for (int k = 0; k < nq; ++k) {
      const int row = ((k + split * 7) % nq) * NWV + wave;   // splits start at different rows (see above)
      const int i = i0 + a0 * 16 + row;
      if (row >= nrows || i >= p.Ni) continue;
#pragma unroll
      for (int h = 0; h < TJR / 64; ++h) {
        const int j = j0 + h * 64 + lane;
        if (j < p.Kj) atomicAdd(dW + (long long)i * p.ldw + j, tile[row * TJR + h * 64 + lane]);
      }
    }
  }
}

int pick_wi(int n16) {
  static const int cand[] = {15, 9, 8, 4, 3, 2};
  // measured efficiency per row-tile count.  The 240-row tile (15) runs ONE workgroup per CU (120 accumulator registers)
  // and loses the latency hiding of a second one: on the shapes that used to pick it, 128-row tiles are 1.4-1.8x faster
  // (layer-2.0 strided conv, Ni = 230: 777 -> 538 us; ffn1, Ni = 3072: 143 -> 78 us), so it only wins by a wide margin.
  static const float eff[] = {0.72f, 0.96f, 0.97f, 0.8f, 0.7f, 0.55f};  // ties between 9 and 8 go to 8 (fewer registers)
  int best = 2;
  float best_cost = 1e30f;
  for (int i = 0; i < 6; ++i) {
    const int c = cand[i];
    const float cost = (float)(((n16 + c - 1) / c) * c) / eff[i];
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

inline long long slab_floats_total(const pp_wgrad_desc& d, int msplit) {
  return ((long long)d.Ni * d.ldw + d.Ni) * msplit * (d.nbatch > 0 ? d.nbatch : 1);
}
inline void launch_slab_sum(const pp_wgrad_desc& d, int msplit, hipStream_t s) {
  const long long n = (long long)d.Ni * d.Kj + (d.dbias ? d.Ni : 0);
  long long gx = (n + 255) / 256;
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(wgrad_slab_sum_kernel, dim3((unsigned)gx, 1, (unsigned)(d.nbatch > 0 ? d.nbatch : 1)), dim3(256), 0, s, (const float*)d.ws,
                     msplit, (long long)d.Ni * d.ldw + d.Ni, d.Ni, d.Kj, d.ldw, d.dW, d.dw_s, d.dbias, d.dbias_s, d.ptr_table);
}

template <int WI>
int launch_wi(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query = nullptr) {
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
  // grouped problems without an M split: one flat grid in slab-sharing order (see the kernel)
  const int flat = (pp_opt_wgrad_flat && d.ptr_table && msplit == 1 && gx * d.nbatch < 0x7fffffffLL) ? 1 : 0;
  dim3 grid((unsigned)(flat ? gx * d.nbatch : gx), 1, (unsigned)(flat ? 1 : d.nbatch)), block(256);
  pp_wgrad_desc k = d;                       // what the kernel sees: ws only when this launch really is split
  const bool slabs = pp_opt_deterministic && msplit > 1;
  if (ws_query) { *ws_query = slabs ? slab_floats_total(d, msplit) : 0; return PP_OK; }
  if (slabs) {
    PP_CHECK_ARG(d.ws && d.ws_floats >= slab_floats_total(d, msplit), "pp_wgrad: deterministic mode needs ws of pp_wgrad_ws_floats(d) = %lld floats",
                 slab_floats_total(d, msplit));
  } else {
    k.ws = nullptr;
  }
  if (d.g.mode == PP_DENSE) {
    if (d.dbias) hipLaunchKernelGGL((wgrad_kernel<WI, PP_DENSE, true>), grid, block, 0, s, k, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
    else hipLaunchKernelGGL((wgrad_kernel<WI, PP_DENSE, false>), grid, block, 0, s, k, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  } else {
    hipLaunchKernelGGL((wgrad_kernel<WI, PP_CONV_FWD, false>), grid, block, 0, s, k, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  }
  if (slabs) launch_slab_sum(d, msplit, s);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int launch_big(const pp_wgrad_desc& d, hipStream_t s) {
  const int nblk_i = (d.Ni + BTI - 1) / BTI;
  const int nblk_j = (d.Kj + BTJ - 1) / BTJ;
  const long long steps = ((long long)d.M + MS - 1) / MS;
  const long long tiles = (long long)nblk_i * nblk_j * d.nbatch;
  // one workgroup per CU: the M split that fills whole rounds of 256 workgroups best, >= 16 steps per split
  long long best = 1;
  double best_eff = 0.0;
  const long long maxs = steps / 16 > 0 ? steps / 16 : 1;
  for (long long ms = 1; ms <= maxs && ms * tiles <= 4096; ++ms) {
    const long long gx = ms * tiles;
    const double eff = (double)gx / (double)(((gx + 255) / 256) * 256) - 0.002 * (double)ms;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ms; }
  }
  int msplit = d.msplit > 0 ? d.msplit : (int)best;
  const int flat = d.ptr_table ? 1 : 0;          // grouped: every tile reduces its whole M
  if (flat) msplit = 1;
  const long long sps = (steps + msplit - 1) / msplit;
  msplit = (int)((steps + sps - 1) / sps);
  const int rows_per_split = (int)(sps * MS);
  WGeom wg;
  wg.dRw = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rw));
  wg.dRh = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rh));
  wg.dRt = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rt));
  const long long gx = (long long)nblk_i * nblk_j * msplit;
  dim3 grid((unsigned)(flat ? gx * d.nbatch : gx), 1, (unsigned)(flat ? 1 : d.nbatch)), block(512);
  if (d.g.mode == PP_DENSE) {
    if (d.dbias) hipLaunchKernelGGL((wgrad_big_kernel<PP_DENSE, true>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
    else hipLaunchKernelGGL((wgrad_big_kernel<PP_DENSE, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  } else {
    hipLaunchKernelGGL((wgrad_big_kernel<PP_CONV_FWD, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

template <int WI, int NWV>
int launch_ring(const pp_wgrad_desc& d, hipStream_t s) {
  constexpr int TJR = 32 * NWV;
  const int nblk_i = (d.Ni + 16 * WI - 1) / (16 * WI);
  const int nblk_j = (d.Kj + TJR - 1) / TJR;
  const long long steps = ((long long)d.M + MS - 1) / MS;
  const long long tiles = (long long)nblk_i * nblk_j;
  // one workgroup per CU: pick the M split that fills whole rounds of 256 workgroups best, >= 16 steps per split
  long long best = 1;
  double best_eff = 0.0;
  const long long maxs = steps / 16 > 0 ? steps / 16 : 1;
  for (long long ms = 1; ms <= maxs && ms * tiles <= 4096; ++ms) {
    const long long gx = ms * tiles;
    const double eff = (double)gx / (double)(((gx + 255) / 256) * 256) - 0.002 * (double)ms;   // splits cost atomics
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ms; }
  }
  int msplit = d.msplit > 0 ? d.msplit : (int)best;
  const int flat = d.ptr_table ? 1 : 0;          // grouped: every tile reduces its whole M (no sum crosses workgroups)
  if (flat) msplit = 1;
  const long long sps = (steps + msplit - 1) / msplit;
  msplit = (int)((steps + sps - 1) / sps);
  const int rows_per_split = (int)(sps * MS);
  WGeom wg;
  wg.dRw = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rw));
  wg.dRh = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rh));
  wg.dRt = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.Rt));
  dim3 grid((unsigned)(tiles * msplit * (flat ? d.nbatch : 1)), 1, 1), block(64 * NWV);
  if (d.g.mode == PP_DENSE) {
    if (d.dbias) hipLaunchKernelGGL((wgrad_ring_kernel<WI, NWV, PP_DENSE, true>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
    else hipLaunchKernelGGL((wgrad_ring_kernel<WI, NWV, PP_DENSE, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  } else {
    hipLaunchKernelGGL((wgrad_ring_kernel<WI, NWV, PP_CONV_FWD, false>), grid, block, 0, s, d, wg, nblk_i, nblk_j, rows_per_split, pp_opt_xcd_remap_wgrad, flat);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

// x_bn_scale / x_bn_shift: only the temporal sliding-window kernel applies a producer BatchNorm to X
static bool xbn_ok(const pp_wgrad_desc& d) {
  return pp_opt_sw_wgrad && (long long)d.M >= pp_opt_sw_wgrad && d.Ni <= 64 && pp_wgrad_tw_ok(d, pp_opt_sw_wgrad == 1);
}
extern "C" int pp_wgrad_xbn_supported(const pp_wgrad_desc* dp) {
  if (!dp) return 0;
  pp_wgrad_desc d = *dp;
  if (d.nbatch <= 0) d.nbatch = 1;
  if (d.M <= 0 || d.Ni <= 0 || d.Kj <= 0 || d.g.mode != PP_CONV_FWD || pp_validate_gather(d.g, d.Kj, "pp_wgrad_xbn_supported") != PP_OK) return 0;
  return xbn_ok(d) ? 1 : 0;
}

static int wgrad_dispatch(const pp_wgrad_desc* dp, pp_stream_t stream, long long* ws_query) {
  PP_CHECK_ARG(dp != nullptr, "pp_wgrad: null descriptor");
  pp_wgrad_desc d = *dp;
  PP_CHECK_ARG(d.M > 0 && d.Ni > 0 && d.Kj > 0, "pp_wgrad: bad sizes");
  PP_CHECK_ARG(ws_query || d.ptr_table || (d.X && d.dY && d.dW), "pp_wgrad: null operand");
  PP_CHECK_ARG(!d.ptr_table || (d.g.mode == PP_DENSE && d.nbatch >= 1 && !d.x_bn_scale && !d.x_bn_shift),
               "pp_wgrad: a pointer table (grouped launch) needs a dense gather, nbatch >= 1 and no fused BatchNorm");
  PP_CHECK_ARG(d.ldy % 8 == 0 && d.ldy >= ((d.Ni + 7) & ~7), "pp_wgrad: ldy=%d too small/unaligned for Ni=%d", d.ldy, d.Ni);
  PP_CHECK_ARG(d.Kj % 8 == 0 && d.ldw >= d.Kj, "pp_wgrad: Kj=%d must be a multiple of 8 and <= ldw", d.Kj);
  PP_CHECK_ARG(d.g.mode == PP_DENSE || d.g.mode == PP_CONV_FWD, "pp_wgrad: gather mode must be dense or conv-fwd");
  PP_CHECK_ARG(d.g.mode == PP_DENSE || (d.g.kt > 0 && d.g.kh > 0 && d.g.kw > 0 && d.g.kt * d.g.kh * d.g.kw <= 256 && d.g.kt < 256 &&
                                        d.g.kh < 256 && d.g.kw < 256), "pp_wgrad: taps %dx%dx%d unsupported (<=256 total)", d.g.kt, d.g.kh, d.g.kw);
  PP_CHECK_ARG(!d.dbias || d.g.mode == PP_DENSE, "pp_wgrad: the fused bias gradient is only built for dense operands");
  PP_CHECK_ARG(d.ptr_table || (((uintptr_t)d.X & 15) == 0 && ((uintptr_t)d.dY & 15) == 0), "pp_wgrad: operands must be 16-byte aligned");
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
  if (d.x_bn_scale || d.x_bn_shift) {
    PP_CHECK_ARG(d.x_bn_scale && d.x_bn_shift && xbn_ok(d),
                 "pp_wgrad: x_bn_scale / x_bn_shift (BatchNorm apply of X's producer) is not available for this problem: "
                 "ask pp_wgrad_xbn_supported first");
    return pp_wgrad_tw_try(d, s, pp_opt_sw_wgrad == 1, ws_query);
  }
  if (pp_opt_sw_wgrad && (long long)d.M >= pp_opt_sw_wgrad && !d.ptr_table) {   // (1,3,3) stride-1 convs: window along m
    const int rc_sw = pp_wgrad_sw_try(d, s, ws_query);
    if (rc_sw != 1) return rc_sw;
    const int rc_tw = pp_wgrad_tw_try(d, s, pp_opt_sw_wgrad == 1, ws_query);          // (3,1,1) stride-1 convs: window over time
    if (rc_tw != 1) return rc_tw;
  }
  const int n16 = (d.Ni + 15) / 16;
  // 256 x 256 tiles (round 4; pp_opt_wgrad_big = the M from which every eligible problem takes them, 0 = never): half the L2
  // traffic of the 128 x 128 tiles.  Not in deterministic mode (no slabs), <= 32 taps (row table), and only where the tiles
  // fill the chip: a single problem splits M to do so; a grouped launch (no split) needs >= 0.6 of its last round of 256.
  if (pp_opt_wgrad_big && !pp_opt_deterministic && d.Ni >= 192 && d.Kj >= 192 &&
      (d.g.mode == PP_DENSE || d.g.kt * d.g.kh * d.g.kw <= 32) && (d.nbatch == 1 || d.ptr_table)) {
    const long long tiles = (long long)((d.Ni + BTI - 1) / BTI) * ((d.Kj + BTJ - 1) / BTJ);
    bool take;
    if (d.ptr_table) {
      const long long all = tiles * d.nbatch;
      take = d.M >= 4096 && (double)all / (double)(((all + 255) / 256) * 256) >= 0.6;
    } else {
      take = (long long)d.M >= pp_opt_wgrad_big || (tiles >= 32 && d.M >= 4096);
    }
    if (take) {
      if (ws_query) { *ws_query = 0; return PP_OK; }
      return launch_big(d, s);
    }
  }
  if (d.ptr_table && pp_opt_wgrad_group_ring && d.g.mode == PP_DENSE && n16 > 4 && d.M >= 1024) {
    // grouped Linear weight gradients on the ring form: 128 / 144 x 192 / 256 tiles of dW (eight waves, one workgroup per CU)
    // halve the number of tiles that re-read a dY / X slab against the 128 x 128 tiles of the register-staged kernel; every
    // tile still reduces its whole M, so its single fp32 add per element lands on a zero: bitwise reproducible
    if (ws_query) { *ws_query = 0; return PP_OK; }
    const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
    const int j6 = ((d.Kj + 191) / 192) * 192, j8 = ((d.Kj + 255) / 256) * 256;
    const bool w9 = c9 <= c8, v6 = j6 < j8;
    if (w9) return v6 ? launch_ring<9, 6>(d, s) : launch_ring<9, 8>(d, s);
    return v6 ? launch_ring<8, 6>(d, s) : launch_ring<8, 8>(d, s);
  }
  if (pp_opt_ring_wgrad && !pp_opt_deterministic && d.nbatch == 1 && !d.ptr_table && n16 > 4 && (long long)d.M >= pp_opt_ring_wgrad) {
    // ring tiles: 128 or 144 rows of dW (less padding wins) x 192 or 256 columns (ditto; 192 on ties)
    const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
    const int j6 = ((d.Kj + 191) / 192) * 192, j8 = ((d.Kj + 255) / 256) * 256;
    const bool w9 = c9 <= c8, v6 = j6 <= j8;
    if (w9) return v6 ? launch_ring<9, 6>(d, s) : launch_ring<9, 8>(d, s);
    return v6 ? launch_ring<8, 6>(d, s) : launch_ring<8, 8>(d, s);
  }
  switch (pick_wi(n16)) {
    case 15: return launch_wi<15>(d, s, ws_query);
    case 9: return launch_wi<9>(d, s, ws_query);
    case 8: return launch_wi<8>(d, s, ws_query);
    case 4: return launch_wi<4>(d, s, ws_query);
    case 3: return launch_wi<3>(d, s, ws_query);
    default: return launch_wi<2>(d, s, ws_query);
  }
}

extern "C" int pp_wgrad(const pp_wgrad_desc* dp, pp_stream_t stream) { return wgrad_dispatch(dp, stream, nullptr); }

extern "C" long long pp_wgrad_ws_floats(const pp_wgrad_desc* dp) {
  if (!pp_opt_deterministic) return 0;
  long long need = 0;
  const int rc = wgrad_dispatch(dp, nullptr, &need);
  return rc == PP_OK ? need : 0;
}

void pp_wgrad_slab_sum(const float* ws, int nsplit, long long slab_floats, int Ni, int Kj, int ldw, float* dW, hipStream_t s) {
  const long long n = (long long)Ni * Kj;
  long long gx = (n + 255) / 256;
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(wgrad_slab_sum_kernel, dim3((unsigned)gx, 1, 1), dim3(256), 0, s, ws, nsplit, slab_floats, Ni, Kj, ldw, dW, 0LL,
                     (float*)nullptr, 0LL, (const unsigned long long*)nullptr);
}
