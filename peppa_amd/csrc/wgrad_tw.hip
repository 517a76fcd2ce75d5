// Sliding-window weight gradient for the (3,1,1) stride-1 "temporal" convolutions of r2plus1d (gfx950):
//
//   dW[i, dt, c] += sum_{n, t, hw} dY[(n, t, hw), i] * X[(n, t + dt - 1, hw), c]          (0 <= t + dt - 1 < T)
//
// The three taps read X one frame (H*W rows) apart, so a walk along flat m shares nothing between them.  The sum over
// m can be taken in any order, though: a workgroup walks a COLUMN -- 64 positions hw of one clip -- through time, and
// keeps the X blocks of frames t-1, t, t+1 (plus two in flight) in a five-slot LDS ring.  Every block is fetched once
// and serves all three taps; dY is fetched once per 144-channel block of X.
//   * one workgroup = TI rows of dW x 144 channels of X x 3 taps = 27 column tiles; nine waves, wave w owns tap w / 3
//     and column tiles 3 (w % 3) .. + 2, so a tap outside the clip (t = 0, t = T-1) is a wave-uniform skip: no masks;
//   * LDS-DMA three steps ahead for X, two for dY; counted vmcnt across raw barriers (igemm.hip's ring protocol);
//   * transposing LDS reads through inline asm with immediate offsets (see wgrad_sw.hip).
// Replaces, for these shapes, the same autograd weight gradient as wgrad.hip (torchvision Conv2Plus1D temporal conv,
// pig/models.py:113-154).
#include "common.h"
#include <type_traits>

extern int pp_opt_xcd_remap_wgrad;
extern int pp_opt_deterministic;
extern int pp_opt_tw_narrow;
extern int pp_opt_tw_producers;
void pp_wgrad_slab_sum(const float* ws, int nsplit, long long slab_floats, int Ni, int Kj, int ldw, float* dW, hipStream_t s);

// Timing ablations for tools/probe/tw_la_sweep.sh (results are WRONG with any bit set; the shipped library has 0):
// 1 no fragment reads / MFMAs, 2 no BatchNorm pass, 4 no DMAs after the prologue, 8 no epilogue
#ifndef PP_TW_ABLATE
#define PP_TW_ABLATE 0
#endif

namespace {

constexpr int ABL = PP_TW_ABLATE;
}
extern const int pp_exp_tw_ablate = PP_TW_ABLATE;   // reported by pp_experimental_build()
namespace {
constexpr int MS = 64;                 // rows (positions hw) per step
constexpr int NWV = 9;                 // multiplying waves
constexpr int NT = 64 * NWV;
constexpr int CB_WIDE = 144;           // X channels per workgroup (9 column tiles per tap); the narrow form takes 48

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {   // n is wave-uniform (look-ahead x pieces per wave)
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
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
    case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
    case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
    case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
    case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
    case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
    case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
    case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
    case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
    case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
    case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
    case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
    case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
    case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    case 57: asm volatile("s_waitcnt vmcnt(57)" ::: "memory"); break;
    case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
    case 59: asm volatile("s_waitcnt vmcnt(59)" ::: "memory"); break;
    case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    case 61: asm volatile("s_waitcnt vmcnt(61)" ::: "memory"); break;
    case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
    case 63: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int OFF>
__device__ __forceinline__ void ds_read_tr(u32x2& v, const unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

struct TwGeom {
  FastDiv dT_, dNHB_;   // exact division by T and by the number of 64-position blocks per frame
  int T, HW, NHB;
  int NS;               // steps in total: clips x NHB x T
  int cstride, cg;      // X row stride (elements), channels per tap in dW's layout
  unsigned x_bytes, y_bytes;   // sizes of X and dY (< 2^31: a lane without a source adds 2^31 to its offset and zeros land)
};

// BNA: X is the raw output y of a BatchNorm unit; every X block gets z = relu?(y * scale + shift) applied once, in LDS,
// after it has landed and before its first use (pp_wgrad_desc.x_bn_*): the activated tensor is never materialised.
// CW = column tiles per wave: 3 = 144-channel X blocks; 1 = 48-channel blocks (the stem's temporal convolution: 45 input
// channels -- in the wide form two thirds of every wave's MFMAs and of the X ring were padding).
// LA = look-ahead in steps: iteration s issues dY step s + LA + 1 and X block s + LA + 2 and ends by waiting for the batch
// of iteration s - LA, so LA batches stay in flight across the barrier.  The wide form has LDS for LA = 1 only (about
// 26 KB per step); the narrow form moves 14 KB per step and at LA = 1 was bound by that latency (1.9 TB/s): LA = 5.
// NPROD = producer waves (waves 9 ..): they issue every LDS-DMA of the step and do nothing else, the nine multiplying
// waves issue none.  Measured on the narrow form (tools/probe/tw_la_sweep.sh): DMAs alone 163 us, fragments + MFMAs alone
// 95 us, together 215 us whatever the look-ahead -- a wave that multiplies is not at its DMA instructions when the memory
// pipeline has room for them, so the two costs ADD; a wave that only issues sits blocked on exactly that.
template <int WI, bool BNA, int CW, int LA, int NPROD>
__global__ __launch_bounds__(NT + 64 * NPROD, 1) void wgrad_tw_kernel(const h16raw* __restrict__ X, const h16raw* __restrict__ dY,
                                                          float* __restrict__ dW, const TwGeom g, const int Ni,
                                                          const int ldy, const int ldw, const int nblk_i,
                                                          const int nblk_c, const int steps_per_split,
                                                          const int xcd_remap, const float* __restrict__ bn_scale,
                                                          const float* __restrict__ bn_shift, const int bn_relu,
                                                          float* __restrict__ slab) {
  constexpr int CB = 48 * CW;
  constexpr int XS = CB * 2;             // X block row stride: 288 = 32 x 9, 96 = 32 x 3 (odd) -> conflict-free tr reads
  constexpr int XSLOT = MS * XS;
  constexpr int XPIECES = XSLOT / 1024;  // 18 / 6
  static_assert(XSLOT % 1024 == 0, "X block = whole DMA pieces");
  constexpr int NPSLOT = LA + 2;
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;
  constexpr int PSLOT = MS * PS;
  constexpr int PPIECES = PSLOT / 1024;
  static_assert(PSLOT % 1024 == 0, "dY slab = whole DMA pieces");
  constexpr int NPIECES = PPIECES + XPIECES;
  constexpr int NP = NPROD ? NPROD : NWV;          // waves that issue DMAs
  constexpr int NK = (NPIECES + NP - 1) / NP;
  static_assert(LA * NK <= 63, "vmcnt is a six-bit counter");
  // X ring: blocks s - 1, s, s + 1 in use and LA + 1 in flight; BNA: one more, so that block s + 2 has landed when
  // iteration s starts and its BatchNorm pass can ride inside the iteration, published by the iteration's own barrier
  constexpr int XA = LA + (BNA ? 3 : 2);           // the X block issued at iteration s is s + XA
  constexpr int NXSLOT = XA + 2;
  constexpr int X_BYTES = NXSLOT * XSLOT;
  constexpr int SMEM = X_BYTES + NPSLOT * PSLOT;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(16 * 3 * CB * 4 <= SMEM, "epilogue stage");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];   // one LDS object (see igemm.hip)
  unsigned char* const pring = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_prod = NPROD ? wave >= NWV : true, is_cons = NPROD ? wave < NWV : true;   // (wave-uniform)
  const int pw = NPROD ? wave - NWV : wave;        // index among the issuing waves
  int bid;
  {
    const int nwg = gridDim.x, b0 = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b0 & 7;
    bid = xcd_remap ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  const int ib = bid % nblk_i; bid /= nblk_i;
  const int cb = bid % nblk_c; bid /= nblk_c;
  const int split = bid;
  const int i0 = ib * TI, c0 = cb * CB;
  const int s_begin = split * steps_per_split;
  const int s_end = min(g.NS, s_begin + steps_per_split);
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)g.x_bytes, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)g.y_bytes, 0x00020000);

  // step index (over the whole problem) -> first row of its 64-position block, rows inside the frame, frame index
  struct StepPos { int m0, nvalid, t; };
  auto decode = [&](const int s) __attribute__((always_inline)) -> StepPos {
    StepPos r;
    const int col = (int)fdiv((uint32_t)s, g.dT_);
    r.t = s - col * g.T;
    const int n = (int)fdiv((uint32_t)col, g.dNHB_);
    const int hb = col - n * g.NHB;
    r.m0 = (n * g.T + r.t) * g.HW + hb * MS;
    r.nvalid = min(MS, g.HW - hb * MS);
    return r;
  };

  // ---- this lane's place in the DMA pieces its wave issues per step (piece q = pw + NP k): ONE byte offset per piece,
  // relative to the step's first row, prepared once; a step adds one scalar.  (A producer wave runs alone on its issue
  // slot: at ~250 instructions per step -- row and column arithmetic per piece -- one producer was slower than none.)
  // A lane without a source (pad columns of the slab, channels past cg) carries NOSRC = 2^31: its sum with any step base
  // stays beyond the resource's size and zeros land.  Steps outside the split / the problem are clamped instead: their
  // slots are never multiplied (compute() skips taps outside the clip, dY steps outside the split are not walked).
  constexpr unsigned NOSRC = 0x80000000u;
  int d_row[NK];
  unsigned d_off[NK];
  int npiece = 0;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int q = (is_prod ? pw : 0) + NP * k;
    if (q < NPIECES) ++npiece;
    if (q < PPIECES) {
      const int o = q * 1024 + lane * 16;
      d_row[k] = o / PS;
      const int cbyte = o % PS;
      const int i = i0 + cbyte / 2;
      d_off[k] = (cbyte < TI * 2 && i < ldy) ? (unsigned)(d_row[k] * ldy + i) * 2u : NOSRC;
    } else {
      const int o = (q - PPIECES) * 1024 + lane * 16;
      d_row[k] = o / XS;
      const int c = c0 + (o % XS) / 2;
      d_off[k] = c < g.cg ? (unsigned)(d_row[k] * g.cstride + c) * 2u : NOSRC;
    }
  }
  npiece = __builtin_amdgcn_readfirstlane(npiece);
  const bool ragged = (g.HW & (MS - 1)) != 0;     // frames that end inside a 64-position block: rows past them are masked
  // issue the pieces of dY step sp and of X block sx (absolute step indices)
  auto dma_pair = [&](const int sp, const int sx, const bool do_p) __attribute__((always_inline)) {
    if (!is_prod) return;
    const StepPos pp = decode(min(max(sp, s_begin), s_end - 1)), px = decode(min(max(sx, 0), g.NS - 1));
    const unsigned pbase = (unsigned)(pp.m0 * ldy) * 2u, xbase = (unsigned)(px.m0 * g.cstride) * 2u;
    unsigned char* const pdst = pring + ((sp - s_begin + NPSLOT) % NPSLOT) * PSLOT;
    unsigned char* const xdst = smem + ((sx + NXSLOT) % NXSLOT) * XSLOT;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = pw + NP * k;
      if (q < PPIECES) {
        if (do_p) lds_dma16(rsY, pdst + q * 1024, (ragged && d_row[k] >= pp.nvalid) ? NOSRC : pbase + d_off[k]);
      } else if (q < NPIECES) {
        lds_dma16(rsX, xdst + (q - PPIECES) * 1024, (ragged && d_row[k] >= px.nvalid) ? NOSRC : xbase + d_off[k]);
      }
    }
  };

  // ---- fragment addressing (tr_frag of wgrad.hip) ---------------------------------------------------------------------
  const int gq = lane >> 4, li = lane & 15;
  const int frow = 4 * gq + (li >> 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned p_lane = lds0 + X_BYTES + (unsigned)(frow * PS + (li & 3) * 8);
  const int my_tap = wave / 3;                                            // 0, 1, 2  <->  frame t - 1, t, t + 1
  const unsigned q_lane = lds0 + (unsigned)(frow * XS + (wave % 3) * (CW * 32) + (li & 3) * 8);

  f32x4 acc[WI][CW];
#pragma unroll
  for (int a = 0; a < WI; ++a)
#pragma unroll
    for (int jt = 0; jt < CW; ++jt) acc[a][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int s) __attribute__((always_inline)) {
    if (!is_cons) return;
    const int col = (int)fdiv((uint32_t)s, g.dT_);
    const int t = s - col * g.T;
    if ((unsigned)(t + my_tap - 1) >= (unsigned)g.T) return;              // this wave's tap leaves the clip
    const unsigned xslot = (unsigned)(((s + my_tap - 1 + NXSLOT) % NXSLOT) * XSLOT);
    const unsigned pslot = (unsigned)(((s - s_begin) % NPSLOT) * PSLOT);
    if constexpr (WI == 4) {
      // 64 rows of dW: every fragment of the step is requested up front (28 / 20 reads in flight: the registers are there)
      // and the step pays two LDS round trips instead of ten -- with a dozen MFMAs per wave and step, the serial
      // request -> wait -> multiply chain WAS the step (narrow form: 1.15 us per step whatever the DMA look-ahead)
      u32x2 q[2][CW][2], pp[2][WI][2];
      static_for<0, 2>([&](auto sc) __attribute__((always_inline)) {
        constexpr int sub = decltype(sc)::value;
        const unsigned qa = q_lane + xslot + (unsigned)(sub * 32 * XS);
        const unsigned pa = p_lane + pslot + (unsigned)(sub * 32 * PS);
        static_for<0, CW>([&](auto jc) __attribute__((always_inline)) {
          constexpr int jt = decltype(jc)::value;
          ds_read_tr<jt * 32>(q[sub][jt][0], qa);
          ds_read_tr<jt * 32 + 16 * XS>(q[sub][jt][1], qa);
        });
        static_for<0, WI>([&](auto ic) __attribute__((always_inline)) {
          constexpr int a = decltype(ic)::value;
          ds_read_tr<a * 32>(pp[sub][a][0], pa);
          ds_read_tr<a * 32 + 16 * PS>(pp[sub][a][1], pa);
        });
      });
#define PP_Q(s_, j_) "+v"(q[s_][j_][0]), "+v"(q[s_][j_][1])
#define PP_P(s_, a_) "+v"(pp[s_][a_][0]), "+v"(pp[s_][a_][1])
      static_for<0, 2>([&](auto sc) __attribute__((always_inline)) {
        constexpr int sub = decltype(sc)::value;
        // (lgkmcnt retires in order: the second half's 2 CW + 2 WI reads may still be in flight)
        if constexpr (CW == 3) {
          if constexpr (sub == 0)
            asm volatile("s_waitcnt lgkmcnt(14)" : PP_Q(0, 0), PP_Q(0, 1), PP_Q(0, 2), PP_P(0, 0), PP_P(0, 1), PP_P(0, 2), PP_P(0, 3) : : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : PP_Q(1, 0), PP_Q(1, 1), PP_Q(1, 2), PP_P(1, 0), PP_P(1, 1), PP_P(1, 2), PP_P(1, 3) : : "memory");
        } else {
          if constexpr (sub == 0)
            asm volatile("s_waitcnt lgkmcnt(10)" : PP_Q(0, 0), PP_P(0, 0), PP_P(0, 1), PP_P(0, 2), PP_P(0, 3) : : "memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)" : PP_Q(1, 0), PP_P(1, 0), PP_P(1, 1), PP_P(1, 2), PP_P(1, 3) : : "memory");
        }
        h16x8 qv[CW];
#pragma unroll
        for (int jt = 0; jt < CW; ++jt)
          qv[jt] = __builtin_bit_cast(h16x8, (u32x4){q[sub][jt][0][0], q[sub][jt][0][1], q[sub][jt][1][0], q[sub][jt][1][1]});
#pragma unroll
        for (int a = 0; a < WI; ++a) {
          const h16x8 pv = __builtin_bit_cast(h16x8, (u32x4){pp[sub][a][0][0], pp[sub][a][0][1], pp[sub][a][1][0], pp[sub][a][1][1]});
#pragma unroll
          for (int jt = 0; jt < CW; ++jt) acc[a][jt] = PP_MFMA16(pv, qv[jt], acc[a][jt], 0, 0, 0);
        }
      });
#undef PP_Q
#undef PP_P
      return;
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned qa = q_lane + xslot + (unsigned)(sub * 32 * XS);
      const unsigned pa = p_lane + pslot + (unsigned)(sub * 32 * PS);
      u32x2 qlo[CW], qhi[CW], plo[2], phi[2];
      ds_read_tr<0>(qlo[0], qa);  ds_read_tr<16 * XS>(qhi[0], qa);
      if constexpr (CW == 3) {
        ds_read_tr<32>(qlo[1], qa); ds_read_tr<32 + 16 * XS>(qhi[1], qa);
        ds_read_tr<64>(qlo[2], qa); ds_read_tr<64 + 16 * XS>(qhi[2], qa);
      }
      ds_read_tr<0>(plo[0], pa);
      ds_read_tr<16 * PS>(phi[0], pa);
      if constexpr (CW == 3) {
        asm volatile("s_waitcnt lgkmcnt(2)"
                     : "+v"(qlo[0]), "+v"(qhi[0]), "+v"(qlo[1]), "+v"(qhi[1]), "+v"(qlo[2]), "+v"(qhi[2])
                     :
                     : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(qlo[0]), "+v"(qhi[0]) : : "memory");
      }
      h16x8 qv[CW];
#pragma unroll
      for (int jt = 0; jt < CW; ++jt) qv[jt] = __builtin_bit_cast(h16x8, (u32x4){qlo[jt][0], qlo[jt][1], qhi[jt][0], qhi[jt][1]});
      static_for<0, WI>([&](auto ic) __attribute__((always_inline)) {
        constexpr int a = decltype(ic)::value;
        constexpr int cur = a & 1;
        if constexpr (a + 1 < WI) {
          ds_read_tr<(a + 1) * 32>(plo[cur ^ 1], pa);
          ds_read_tr<(a + 1) * 32 + 16 * PS>(phi[cur ^ 1], pa);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(plo[cur]), "+v"(phi[cur]) : : "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(plo[cur]), "+v"(phi[cur]) : : "memory");
        }
        const h16x8 pv = __builtin_bit_cast(h16x8, (u32x4){plo[cur][0], plo[cur][1], phi[cur][0], phi[cur][1]});
#pragma unroll
        for (int jt = 0; jt < CW; ++jt) acc[a][jt] = PP_MFMA16(pv, qv[jt], acc[a][jt], 0, 0, 0);
      });
    }
  };

  // ---- BNA: a block is 64 rows x 18 octets of 8 channels = 1152 16-byte chunks, two per thread; thread t keeps octet
  // t % 18 (576 = 32 x 18), so its sixteen parameters live in registers.  (Narrow form: 64 x 6 chunks, one for each of the
  // first 384 threads.)  Rows past the frame and channels past cg hold zeros that the pass may turn into relu(shift): their
  // dY rows / dW columns are zero / unwritten.
  float bsc[8], bsh[8];
  constexpr int OC = CB / 8, RPP = NT / OC;        // octets per row; rows one pass of the workgroup covers
  static_assert(NT % OC == 0 && (RPP == 32 || RPP >= MS), "BNA pass: two rows per thread, or at most one");
  const int b_c8 = tid % OC, b_row = tid / OC;
  if (BNA) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + b_c8 * 8 + q;
      bsc[q] = c < g.cg ? bn_scale[c] : 0.f;
      bsh[q] = c < g.cg ? bn_shift[c] : 0.f;
    }
  }
  auto bn_block = [&](const int sx) __attribute__((always_inline)) {     // X block of absolute step sx, in place
    if (!is_cons) return;
    const unsigned base = lds0 + (unsigned)(((sx + NXSLOT) % NXSLOT) * XSLOT + b_c8 * 16);
    // (inline asm: plain LDS accesses make hipcc wait for the LDS-DMAs in flight first)
    const unsigned a0 = base + (unsigned)(b_row * XS), a1 = a0 + (unsigned)(32 * XS);
    u32x4 v0, v1;
    if constexpr (RPP == 32) {
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
    } else {
      if (b_row >= MS) return;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0) : "v"(a0) : "memory");
      v1 = v0;
    }
#pragma unroll
    for (int k = 0; k < (RPP == 32 ? 2 : 1); ++k) {
      const u32x4 vv = k ? v1 : v0;
      float x[8];
      unpack8(make_uint4(vv[0], vv[1], vv[2], vv[3]), x);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float zv = x[q] * bsc[q] + bsh[q];
        x[q] = (bn_relu && !(zv > 0.f)) ? 0.f : zv;
      }
      const uint4 o = pack8(x);
      const u32x4 ov = {o.x, o.y, o.z, o.w};
      asm volatile("ds_write_b128 %0, %1" ::"v"(k ? a1 : a0), "v"(ov) : "memory");
    }
  };

  // ---- prologue: X blocks s_begin - 1 .. s_begin + XA - 1, dY steps s_begin .. s_begin + LA -------------------------
#pragma unroll
  for (int i = 0; i <= XA; ++i) dma_pair(s_begin + i - (XA - LA), s_begin - 1 + i, i >= XA - LA);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (BNA) {                      // blocks s_begin - 1 .. s_begin + 1; block s + 2 follows inside iteration s
    bn_block(s_begin - 1);
    bn_block(s_begin);
    bn_block(s_begin + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- main loop.  Iteration s: issue dY step s + LA + 1 and X block s + XA (its slot held block s - 2), multiply step s,
  // [BNA: activate block s + 2, which landed before the barrier that ended iteration s - 1 and is first read at step
  // s + 1], wait for everything issued before iteration s - LA + 1, barrier. -------------------------------------------
  for (int s = s_begin; s < s_end; ++s) {
    if (!(ABL & 4)) dma_pair(s + LA + 1, s + XA, true);
    if (!(ABL & 1)) compute(s);
    if (BNA && !(ABL & 2)) bn_block(s + 2);       // (behind the step's MFMAs: its LDS round trip and VALU work run while they drain)
    if (is_prod) wait_vmcnt_dyn((ABL & 4) ? 0 : LA * npiece);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: 16 rows of dW at a time through LDS; an atomic wave-instruction adds 256 contiguous bytes ------------
  if (ABL & 8) return;
  float* stage = (float*)smem;                 // 16 x 432 floats
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < WI; ++a) {
    __syncthreads();
    if (is_cons) {
#pragma unroll
      for (int jt = 0; jt < CW; ++jt) {
        const int col = my_tap * CB + ((wave % 3) * CW + jt) * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(fq * 4 + r) * (3 * CB) + col] = acc[a][jt][r];
      }
    }
    __syncthreads();
    for (int idx = is_cons ? tid : 16 * 3 * CB; idx < 16 * 3 * CB; idx += NT) {
      const int row0 = idx / (3 * CB), col = idx - row0 * (3 * CB);
      const int row = (row0 + split) & 15;                        // splits start at different rows
      const int i = i0 + a * 16 + row;
      const int tap = col / CB, c = c0 + col - tap * CB;
      if (i < Ni && c < g.cg) {
        const long long o = (long long)i * ldw + tap * g.cg + c;
        if (slab) slab[(long long)split * Ni * ldw + o] = stage[row * (3 * CB) + col];      // deterministic mode
        else atomicAdd(dW + o, stage[row * (3 * CB) + col]);
      }
    }
  }
}

template <int WI, int CW, int NPROD>
int launch_tw(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query) {
  constexpr int CB = 48 * CW;
#ifndef PP_TW_LA
#define PP_TW_LA 5
#endif
  constexpr int LA = CW == 1 ? PP_TW_LA : 1;
  const pp_gather& gg = d.g;
  TwGeom g;
  g.T = gg.Gt; g.HW = gg.Gh * gg.Gw; g.NHB = (g.HW + MS - 1) / MS;
  const int clips = d.M / (gg.Gt * g.HW);
  g.NS = clips * g.NHB * g.T;
  g.cstride = gg.cstride; g.cg = gg.cg;
  g.x_bytes = (unsigned)((long long)d.M * gg.cstride * 2);
  g.y_bytes = (unsigned)((long long)d.M * d.ldy * 2);
  g.dT_ = make_fastdiv((uint32_t)g.T);
  g.dNHB_ = make_fastdiv((uint32_t)g.NHB);
  const int nblk_i = (d.Ni + 16 * WI - 1) / (16 * WI);
  const int nblk_c = (gg.cg + CB - 1) / CB;
  const long long tiles = (long long)nblk_i * nblk_c;
  // one workgroup per CU: the split of the step stream that fills whole rounds of 256 best, >= 12 steps per split
  long long best = 1;
  double best_eff = 0.0;
  const long long maxs = g.NS / 12 > 0 ? g.NS / 12 : 1;
  for (long long ms = 1; ms <= maxs && ms * tiles <= 2048; ++ms) {
    const long long gx = ms * tiles;
    const double eff = (double)gx / (double)(((gx + 255) / 256) * 256) - 0.0005 * (double)ms;
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ms; }
  }
  int msplit = d.msplit > 0 ? d.msplit : (int)best;
  const int sps = (g.NS + msplit - 1) / msplit;
  msplit = (g.NS + sps - 1) / sps;
  dim3 grid((unsigned)(tiles * msplit), 1, 1), block(NT + 64 * NPROD);
  const bool slabs = pp_opt_deterministic && msplit > 1;
  const long long need = slabs ? (long long)msplit * d.Ni * d.ldw : 0;
  if (ws_query) { *ws_query = need; return PP_OK; }
  if (slabs) PP_CHECK_ARG(d.ws && d.ws_floats >= need, "pp_wgrad: deterministic mode needs ws of pp_wgrad_ws_floats(d) = %lld floats", need);
  float* const slab = slabs ? d.ws : nullptr;
  if (d.x_bn_scale) {
    if constexpr (WI == 4) {      // (64 rows of dW: the six-slot X ring does not fit beside the dY ring of 128)
      hipLaunchKernelGGL((wgrad_tw_kernel<WI, true, CW, LA, NPROD>), grid, block, 0, s, (const h16raw*)d.X, (const h16raw*)d.dY, d.dW, g, d.Ni, d.ldy,
                         d.ldw, nblk_i, nblk_c, sps, pp_opt_xcd_remap_wgrad, d.x_bn_scale, d.x_bn_shift, d.x_bn_relu, slab);
    } else {
      pp_set_error("pp_wgrad: x_bn_scale / x_bn_shift need Ni <= 64 in the temporal sliding-window kernel");
      return PP_ERR_INVALID;
    }
  } else
    hipLaunchKernelGGL((wgrad_tw_kernel<WI, false, CW, LA, NPROD>), grid, block, 0, s, (const h16raw*)d.X, (const h16raw*)d.dY, d.dW, g, d.Ni, d.ldy,
                       d.ldw, nblk_i, nblk_c, sps, pp_opt_xcd_remap_wgrad, (const float*)nullptr, (const float*)nullptr, 0, slab);
  if (slabs) pp_wgrad_slab_sum(d.ws, msplit, (long long)d.Ni * d.ldw, d.Ni, d.Kj, d.ldw, d.dW, s);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

// Is `d` a problem the temporal sliding-window kernel takes?  (`force`: without the is-it-worth-it rule)
bool pp_wgrad_tw_ok(const pp_wgrad_desc& d, const bool force) {
  const pp_gather& g = d.g;
  const long long frame = (long long)g.Gh * g.Gw;
  const bool shape_ok = g.mode == PP_CONV_FWD && d.nbatch == 1 && !d.dbias && g.kt == 3 && g.kh == 1 && g.kw == 1 &&
                        g.st == 1 && g.sh == 1 && g.sw == 1 && g.pt == 1 && g.ph == 0 && g.pw == 0 && g.Gt == g.Rt &&
                        g.Gh == g.Rh && g.Gw == g.Rw && g.cg >= 48 && g.cg % 16 == 0 && d.Kj == 3 * g.cg && d.Ni >= 64 &&
                        d.M % (g.Gt * frame) == 0 && (long long)d.M * g.cstride * 2 < 0x7fffffffLL &&
                        (long long)d.M * d.ldy * 2 < 0x7fffffffLL;
  if (!shape_ok) return false;
  // worth it only where the 64-position blocks are mostly full and the clip is long enough that few steps lose a tap
  // (measured: layer 1/2 shapes 1.3-2.1x faster than the gather kernel, 14x14 / 7x7 frames with T <= 4 slower; the stem's
  // 48 channels fill a third of a 144-channel block and still run 292 vs 333 us)
  const long long nhb = (frame + MS - 1) / MS;
  return force || !(frame * 10 < nhb * MS * 9 || g.Gt < 4);
}

// PP_OK if the temporal sliding-window kernel took the problem, 1 if the shape is not one it handles, < 0 on error.
// `force` (tests: pp_set_option("sw_wgrad", 1)) skips the is-it-worth-it rule.
int pp_wgrad_tw_try(const pp_wgrad_desc& d, hipStream_t s, const bool force, long long* ws_query) {
  if (!pp_wgrad_tw_ok(d, force)) return 1;
  const int n16 = (d.Ni + 15) / 16;
#ifndef PP_TW_NPROD
#define PP_TW_NPROD 3
#endif
  if (pp_opt_tw_producers) {
    if (n16 <= 4 && d.g.cg <= 48 && pp_opt_tw_narrow) return launch_tw<4, 1, PP_TW_NPROD>(d, s, ws_query);
    return n16 <= 4 ? launch_tw<4, 3, PP_TW_NPROD>(d, s, ws_query) : launch_tw<8, 3, PP_TW_NPROD>(d, s, ws_query);
  }
  if (n16 <= 4 && d.g.cg <= 48 && pp_opt_tw_narrow) return launch_tw<4, 1, 0>(d, s, ws_query);
  return n16 <= 4 ? launch_tw<4, 3, 0>(d, s, ws_query) : launch_tw<8, 3, 0>(d, s, ws_query);
}
