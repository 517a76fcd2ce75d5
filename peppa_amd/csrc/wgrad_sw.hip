// Sliding-window weight gradient for the (1,3,3) stride-1 "spatial" convolutions of r2plus1d / resnet (gfx950):
//
//   dW[i, tap, c] += sum_m dY[m, i] * X[m + off(tap), c] * inside(m, tap)        off = (dh-1) * W + (dw-1)
//
// The generic kernel (wgrad.hip) gathers the nine shifted copies of X from global memory: 9 x the bytes, 9 x the
// address arithmetic, and at layer-1 sizes (M = 3.2 M rows) it is bound by load instructions and their latency, not
// by the matrix cores.  Here a workgroup keeps a WINDOW of X rows in LDS and walks along m:
//   * one workgroup = one 144- (or 128-) row block of dW x one 64-channel block of X x ALL nine taps
//     (36 column tiles of 16; twelve waves, three tiles each -> 108 accumulator registers per lane);
//   * per 64-row step it fetches 64 new rows of X (8 KB) and the 64 x TI slab of dY, by LDS-DMA, three steps
//     ahead (512-row ring for X, three slots for dY, counted vmcnt across raw barriers as in igemm.hip's ring);
//     a step reads X rows up to W + 1 before / after its own 64: one 64-row group on either side (XA = 1, W <= 63) or two
//     (XA = 2, W <= 127: the reference's own 100 x 180 clips are 50 x 90 at layer 1) -- the X stream then runs a group further ahead;
//   * the nine taps are nine row-shifted views of the window: the transposing LDS reads (ds_read_b64_tr_b16) take
//     per-lane row addresses, so a shift is an address offset.  Rows whose tap falls outside the image (the window
//     holds the neighbouring row / frame there) are cleared with a 16-bit AND mask per (row, tap), which one wave
//     prepares per step.
// Global traffic drops to the compulsory bytes (X once, dY once per channel block).  Replaces, for these shapes, the
// same autograd weight gradient as wgrad.hip (torchvision Conv2Plus1D spatial conv, pig/models.py:113-154).
#include "common.h"
#include <type_traits>

extern int pp_opt_xcd_remap_wgrad;
extern int pp_opt_deterministic;
void pp_wgrad_slab_sum(const float* ws, int nsplit, long long slab_floats, int Ni, int Kj, int ldw, float* dW, hipStream_t s);

namespace {

constexpr int MS = 64;                 // rows per step
constexpr int NWV = 12;                // waves per workgroup
constexpr int NT = 64 * NWV;
constexpr int XS = 160;                // X window row stride (128 data bytes + 32: 32 x odd -> conflict-free tr reads)
constexpr int XROWS = 512;             // X window ring (rows)
constexpr int X_BYTES = XROWS * XS;
constexpr int XPIECES = MS * XS / 1024;   // DMA pieces per 64-row group (10)
constexpr int NPSLOT = 3;              // dY ring slots
constexpr int MASK_BYTES = 2 * 9 * MS * 2;
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {   // n is wave-uniform, 0..3 here
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- the index can then be an asm immediate
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int OFF>
__device__ __forceinline__ void ds_read_tr(u32x2& v, const unsigned addr) {   // address + immediate offset: no VGPR per fragment
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

struct SwGeom {
  FastDiv dW_, dH_;   // exact division by the image width / height
  int W, H;
  int M;              // rows of dY / X (N * T * H * W)
  int cstride;        // X row stride (elements)
  int cg;             // channels per tap in dW's layout (= padded Ci)
};

template <int WI, int XA>
__global__ __launch_bounds__(NT, 1) void wgrad_sw_kernel(const h16raw* __restrict__ X, const h16raw* __restrict__ dY,
                                                          float* __restrict__ dW, const SwGeom g, const int Ni,
                                                          const int ldy, const int ldw, const int nblk_i,
                                                          const int nblk_c, const int rows_per_split,
                                                          const int xcd_remap, float* __restrict__ slab) {
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;   // dY slab row stride (32 x odd)
  constexpr int P_BYTES = MS * PS;
  constexpr int PPIECES = P_BYTES / 1024;
  static_assert(P_BYTES % 1024 == 0, "dY slab = whole DMA pieces");
  constexpr int NPIECES = PPIECES + XPIECES;            // pieces per step, dealt round-robin to the waves
  constexpr int NK = (NPIECES + NWV - 1) / NWV;
  constexpr int SMEM = X_BYTES + NPSLOT * P_BYTES + MASK_BYTES;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];   // one LDS object (see igemm.hip)
  unsigned char* const xwin = smem;
  unsigned char* const pring = smem + X_BYTES;
  unsigned short* const masktab = (unsigned short*)(smem + X_BYTES + NPSLOT * P_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  const int ib = bid % nblk_i; bid /= nblk_i;
  const int cb = bid % nblk_c; bid /= nblk_c;
  const int split = bid;
  const int i0 = ib * TI, c0 = cb * 64;
  const int m_begin = split * rows_per_split;            // multiple of 64
  const int m_end = min(g.M, m_begin + rows_per_split);
  const int nsteps = (m_end - m_begin + MS - 1) / MS;
  if (nsteps <= 0) return;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);

  // ---- this lane's place in the (up to NK) DMA pieces its wave issues per step ---------------------------------
  // piece q = wave + NWV * k: q < PPIECES -> piece q of the dY slab, else piece q - PPIECES of the X group
  int d_row[NK];         // row inside the 64-row slab / group
  unsigned d_col[NK];    // byte offset inside the source row, OOB for pad lanes
  int npiece = 0;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int q = wave + NWV * k;
    if (q < NPIECES) ++npiece;
    if (q < PPIECES) {
      const int o = q * 1024 + lane * 16;
      d_row[k] = o / PS;
      const int cbyte = o % PS;
      const int i = i0 + cbyte / 2;
      d_col[k] = (cbyte < TI * 2 && i < ldy) ? (unsigned)i * 2u : OOB;
    } else {
      const int o = (q - PPIECES) * 1024 + lane * 16;
      d_row[k] = o / XS;
      const int cbyte = o % XS;
      d_col[k] = (cbyte < 128) ? (unsigned)(c0 * 2 + cbyte) : OOB;
    }
  }
  npiece = __builtin_amdgcn_readfirstlane(npiece);
  // issue this wave's pieces of step `st` (dY rows of the step, X rows of group st + XA relative to the step);
  // steps outside [0, nsteps) and rows outside the tensor fetch zeros
  auto dma_step = [&](const int st) __attribute__((always_inline)) {
    const int mP = m_begin + st * MS;              // first dY row of the step
    const int mX = mP + XA * MS;                   // first row of the X group that step st brings in (XA groups ahead)
    unsigned char* const pdst = pring + (st % NPSLOT) * P_BYTES;
    unsigned char* const xdst = xwin + ((mX & (XROWS - 1)) * XS);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = wave + NWV * k;                // wave-uniform
      if (q < PPIECES) {
        const int m = mP + d_row[k];
        const bool ok = (d_col[k] != OOB) & (m < m_end) & (st < nsteps);
        lds_dma16(rsY, pdst + q * 1024, ok ? (unsigned)(m * ldy) * 2u + d_col[k] : OOB);
      } else if (q < NPIECES) {
        const int m = mX + d_row[k];
        const bool ok = (d_col[k] != OOB) & ((unsigned)m < (unsigned)g.M);
        lds_dma16(rsX, xdst + (q - PPIECES) * 1024, ok ? (unsigned)(m * g.cstride) * 2u + d_col[k] : OOB);
      }
    }
  };
  // AND masks of step `st`: masktab[st & 1][tap][row] = 0xFFFF if tap of row m lies inside the image, else 0
  auto make_masks = [&](const int st) __attribute__((always_inline)) {
    const int m = m_begin + st * MS + lane;
    const uint32_t q1 = fdiv((uint32_t)m, g.dW_);
    const int w = m - (int)q1 * g.W;
    const int h = (int)q1 - (int)fdiv(q1, g.dH_) * g.H;
    unsigned short* mt = masktab + (st & 1) * 9 * MS + lane;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const bool ok = ((unsigned)(h + dh - 1) < (unsigned)g.H) & ((unsigned)(w + dw - 1) < (unsigned)g.W);
        mt[(dh * 3 + dw) * MS] = ok ? 0xFFFFu : 0u;
      }
  };

  // ---- fragment addressing (see tr_frag in wgrad.hip): lane (4q+p) of 16-lane group gq supplies the address of row
  // 4 gq + q (and +16), columns 4p..4p+3, and receives column (lane & 15) of rows 4 gq .. 4 gq + 3 --------------------
  const int gq = lane >> 4, li = lane & 15;
  const int frow = 4 * gq + (li >> 2);            // row this lane addresses inside a 32-row half-step
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned p_lane = lds0 + X_BYTES + (unsigned)(frow * PS + (li & 3) * 8);
  // the wave's three column tiles: J = 3 wave + jt -> tap J / 4, 16-channel block J % 4
  int q_off[3];          // row shift of the tap
  unsigned q_colb[3];    // byte offset of this lane's columns inside a window row
  unsigned q_mask[3];    // byte offset of the tap's mask row (without step parity / half-step)
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    const int J = 3 * wave + jt;
    const int tap = J >> 2, cblk = J & 3;
    q_off[jt] = (tap / 3 - 1) * g.W + (tap % 3 - 1);
    q_colb[jt] = (unsigned)(cblk * 32 + (li & 3) * 8);
    q_mask[jt] = lds0 + (unsigned)(X_BYTES + NPSLOT * P_BYTES) + (unsigned)((tap * MS + 4 * gq) * 2);
  }

  f32x4 acc[WI][3];
#pragma unroll
  for (int a = 0; a < WI; ++a)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) acc[a][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // All LDS reads of the loop are inline asm (a builtin ds_read_tr would make hipcc drain vmcnt, i.e. the DMAs in
  // flight); the explicit waits "modify" the registers they cover, so no use can be scheduled above them.
  struct QFrag { u32x2 lo, hi, mlo, mhi; };
  auto read_q = [&](const int mrow0, const unsigned mpar, const int sub, const int jt, QFrag& f) __attribute__((always_inline)) {
    const int r = mrow0 + sub * 32 + frow + q_off[jt];
    const unsigned a_lo = lds0 + (unsigned)((r & (XROWS - 1)) * XS) + q_colb[jt];
    const unsigned a_hi = lds0 + (unsigned)(((r + 16) & (XROWS - 1)) * XS) + q_colb[jt];
    const unsigned a_m = q_mask[jt] + mpar + (unsigned)(sub * 32 * 2);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a_lo) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a_hi) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(f.mlo) : "v"(a_m) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(f.mhi) : "v"(a_m) : "memory");
  };
  auto compute = [&](const int st) __attribute__((always_inline)) {
    const int mrow0 = m_begin + st * MS;
    const unsigned mpar = (unsigned)((st & 1) * 9 * MS * 2);
    const unsigned pslot = (unsigned)((st % NPSLOT) * P_BYTES);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // the three X fragments, one after the other (registers are tight at three waves per SIMD): fragment jt + 1
      // is in flight while fragment jt is masked
      h16x8 qv[3];
      QFrag qf[2];
      read_q(mrow0, mpar, sub, 0, qf[0]);
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) {
        QFrag& f = qf[jt & 1];
        if (jt + 1 < 3) {
          read_q(mrow0, mpar, sub, jt + 1, qf[(jt + 1) & 1]);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f.lo), "+v"(f.hi), "+v"(f.mlo), "+v"(f.mhi) : : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi), "+v"(f.mlo), "+v"(f.mhi) : : "memory");
        }
        qv[jt] = __builtin_bit_cast(h16x8, (u32x4){f.lo[0] & f.mlo[0], f.lo[1] & f.mlo[1], f.hi[0] & f.mhi[0], f.hi[1] & f.mhi[1]});
      }
      const unsigned pa = p_lane + pslot + (unsigned)(sub * 32 * PS);
      u32x2 plo[2], phi[2];
      ds_read_tr<0>(plo[0], pa);
      ds_read_tr<16 * PS>(phi[0], pa);
      static_for<0, WI>([&](auto ic) __attribute__((always_inline)) {
        constexpr int a = decltype(ic)::value;
        constexpr int cur = a & 1;
        if constexpr (a + 1 < WI) {   // next dY fragment under these MFMAs
          ds_read_tr<(a + 1) * 32>(plo[cur ^ 1], pa);
          ds_read_tr<(a + 1) * 32 + 16 * PS>(phi[cur ^ 1], pa);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(plo[cur]), "+v"(phi[cur]) : : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(plo[cur]), "+v"(phi[cur]) : : "memory");
        }
        const h16x8 pv = __builtin_bit_cast(h16x8, (u32x4){plo[cur][0], plo[cur][1], phi[cur][0], phi[cur][1]});
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) acc[a][jt] = PP_MFMA16(pv, qv[jt], acc[a][jt], 0, 0, 0);
      });
    }
  };

  // ---- prologue: dma_step(st) brings dY step st and X group st + XA (step st reads groups st - XA .. st + XA), so the
  // X stream starts 2 XA groups early: groups -XA .. XA - 1 alone, then steps 0 and 1; masks of step 0 -----------------
  auto dma_x_only = [&](const int grp) __attribute__((always_inline)) {   // X group `grp` only (prologue)
    const int mX = m_begin + grp * MS;
    unsigned char* const xdst = xwin + ((mX & (XROWS - 1)) * XS);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = wave + NWV * k;
      if (q >= PPIECES && q < NPIECES) {
        const int m = mX + d_row[k];
        const bool ok = (d_col[k] != OOB) & ((unsigned)m < (unsigned)g.M);
        lds_dma16(rsX, xdst + (q - PPIECES) * 1024, ok ? (unsigned)(m * g.cstride) * 2u + d_col[k] : OOB);
      }
    }
  };
#pragma unroll
  for (int grp = -XA; grp < XA; ++grp) dma_x_only(grp);
  dma_step(0);
  dma_step(1);
  if (wave == 0) make_masks(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- main loop.  Iteration st: issue step st + 2, prepare the masks of step st + 1, multiply step st, then wait for
  // everything issued BEFORE this iteration (step st + 1 complete) and meet at the barrier. --------------------------
  for (int st = 0; st < nsteps; ++st) {
    dma_step(st + 2);
    if (wave == (st + 1) % NWV) make_masks(st + 1);
    compute(st);
    wait_vmcnt_dyn(npiece);    // this wave's pieces of step st + 1 have landed; step st + 2's may still fly
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: one 16-row block of dW at a time through LDS, so that every atomic wave-instruction adds the 256
  // contiguous bytes of one (row, tap): [TI][9][64] fp32 of this channel block --------------------------------------
  float* stage = (float*)smem;                 // 16 x 576 floats = 36 KB
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < WI; ++a) {
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      const int J = 3 * wave + jt;
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(fq * 4 + r) * 576 + J * 16 + fr] = acc[a][jt][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * 576; idx += NT) {
      const int row = idx / 576, col = idx - row * 576;
      const int i = i0 + a * 16 + ((row + split) & 15);          // splits start at different rows
      const int tap = col >> 6, c = c0 + (col & 63);
      if (i < Ni && c < g.cg) {
        const long long o = (long long)i * ldw + tap * g.cg + c;
        if (slab) slab[(long long)split * Ni * ldw + o] = stage[((row + split) & 15) * 576 + col];      // deterministic mode
        else atomicAdd(dW + o, stage[((row + split) & 15) * 576 + col]);
      }
    }
  }
}

template <int WI, int XA>
int launch_sw(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query) {
  const pp_gather& gg = d.g;
  SwGeom g;
  g.W = gg.Gw; g.H = gg.Gh; g.M = d.M; g.cstride = gg.cstride; g.cg = gg.cg;
  g.dW_ = make_fastdiv((uint32_t)gg.Gw);
  g.dH_ = make_fastdiv((uint32_t)gg.Gh);
  const int nblk_i = (d.Ni + 16 * WI - 1) / (16 * WI);
  const int nblk_c = (gg.cg + 63) / 64;
  const long long steps = ((long long)d.M + MS - 1) / MS;
  const long long tiles = (long long)nblk_i * nblk_c;
  // one workgroup per CU: the M split that fills whole rounds of 256 best, at least 12 steps per split
  long long best = 1;
  double best_eff = 0.0;
  const long long maxs = steps / 12 > 0 ? steps / 12 : 1;
  for (long long ms = 1; ms <= maxs && ms * tiles <= 2048; ++ms) {
    const long long gx = ms * tiles;
    const double eff = (double)gx / (double)(((gx + 255) / 256) * 256) - 0.0005 * (double)ms;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ms; }
  }
  int msplit = d.msplit > 0 ? d.msplit : (int)best;
  const long long sps = (steps + msplit - 1) / msplit;
  msplit = (int)((steps + sps - 1) / sps);
  const int rows_per_split = (int)(sps * MS);
  dim3 grid((unsigned)(tiles * msplit), 1, 1), block(NT);
  const bool slabs = pp_opt_deterministic && msplit > 1;
  const long long need = slabs ? (long long)msplit * d.Ni * d.ldw : 0;
  if (ws_query) { *ws_query = need; return PP_OK; }
  if (slabs) PP_CHECK_ARG(d.ws && d.ws_floats >= need, "pp_wgrad: deterministic mode needs ws of pp_wgrad_ws_floats(d) = %lld floats", need);
  hipLaunchKernelGGL((wgrad_sw_kernel<WI, XA>), grid, block, 0, s, (const h16raw*)d.X, (const h16raw*)d.dY, d.dW, g, d.Ni, d.ldy,
                     d.ldw, nblk_i, nblk_c, rows_per_split, pp_opt_xcd_remap_wgrad, slabs ? d.ws : (float*)nullptr);
  if (slabs) pp_wgrad_slab_sum(d.ws, msplit, (long long)d.Ni * d.ldw, d.Ni, d.Kj, d.ldw, d.dW, s);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

// Returns PP_OK if the sliding-window kernel took the problem, 1 if the shape is not one it handles (the caller falls
// through to the generic kernel), or a negative error.
int pp_wgrad_sw_try(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query) {
  const pp_gather& g = d.g;
  const bool shape_ok = g.mode == PP_CONV_FWD && d.nbatch == 1 && !d.dbias && g.kt == 1 && g.kh == 3 && g.kw == 3 &&
                        g.st == 1 && g.sh == 1 && g.sw == 1 && g.pt == 0 && g.ph == 1 && g.pw == 1 && g.Gt == g.Rt &&
                        g.Gh == g.Rh && g.Gw == g.Rw && g.Gw + 1 <= 2 * MS && g.cg % 64 == 0 && d.Kj == 9 * g.cg &&
                        d.Ni >= 128 && (long long)d.M * g.cstride < 0x7fffffffLL && (long long)d.M * d.ldy < 0x7fffffffLL;
  if (!shape_ok) return 1;
  const int n16 = (d.Ni + 15) / 16;
  const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
  if (g.Gw + 1 > MS) return c9 <= c8 ? launch_sw<9, 2>(d, s, ws_query) : launch_sw<8, 2>(d, s, ws_query);   // frames 64..127 wide
  return c9 <= c8 ? launch_sw<9, 1>(d, s, ws_query) : launch_sw<8, 1>(d, s, ws_query);
}
