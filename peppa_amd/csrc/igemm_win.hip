// Window implicit GEMM for the (1,3,3) stride-1 "spatial" convolutions (forward and data gradient), gfx950.
//
//   C[m, n] = sum_{tap, c} A[m + off(tap), c] * Bt[n, tap, c]        off = +-((dh-1) * W + (dw-1)), zero outside the image
//
// igemm.hip gathers the nine shifted copies of A from global memory, one 64-deep K-step at a time: at layer-1 sizes
// that is nine times the load instructions / L2 traffic the data needs, and the kernel runs at a fifth of the matrix
// peak.  Here a 256-row tile keeps its rows PLUS a halo of W + 1 rows on either side in LDS (one 48- or 64-channel
// chunk at a time, double buffered) and every tap is an address offset into that window:
//   * the window of the next (tile, chunk) arrives by LDS-DMA, one 1-KiB piece per K-step, behind the weight slices;
//   * the weights stream through the same three-slot ring as igemm.hip's ring kernel (K order inside a chunk is
//     (tap, channel), flat, so 48-channel chunks -- 144, 240, 288 ... channels -- lose only the last half K-step);
//   * taps that fall outside the image read a zero row instead (per-row 9-bit masks, computed once per tile);
//   * counted vmcnt waits across raw barriers: each iteration waits for everything but the batch it issued last.
// Same bf16 results as igemm.hip (same products, fp32 accumulation in a different order).
// Replaces torchvision Conv2Plus1D's spatial Conv3d forward / input gradient (pig/models.py:113-154 call site).
#include "common.h"
#include <type_traits>

extern int pp_opt_xcd_remap_igemm;
extern int pp_opt_win_tall;
extern int pp_opt_win_temporal;
extern int pp_opt_win_out_nt;
extern int pp_opt_persist_cus;
extern int pp_opt_win_igemm;
extern int pp_opt_win_stagger;
extern int pp_opt_win_producers;
extern int pp_opt_win_s2d;
extern int pp_opt_win_partial;
extern int pp_opt_win_ragged;
extern int pp_opt_win_kpb;

// Timing ablations for tools/probe/win_ablate.py (results are WRONG with any bit set; the shipped library has 0):
// 1 weights only for a workgroup's first tile, 2 windows likewise, 4 no fragment reads / MFMAs, 8 no epilogue,
// 16 fragment reads but no MFMAs, 32 epilogue without its global stores, 64 s_memtime stamps per wave and segment
// (pp_debug_win_stamps copies them out: [workgroup][wave][wait, barrier, issue, multiply, tile turn, epilogue, total, -])
#ifndef PP_WIN_ABLATE
#define PP_WIN_ABLATE 0
#endif
// where in a K-step the window waves issue their LDS-DMA pieces: 0 before the fragment requests, 1 after them (where the
// weight waves issue theirs), 2 between the two halves' MFMAs, 3 after the MFMAs
#ifndef PP_WIN_WPOS
#define PP_WIN_WPOS 0
#endif

namespace {

constexpr int ABL = PP_WIN_ABLATE;
}
extern const int pp_exp_win_ablate = PP_WIN_ABLATE;   // reported by pp_experimental_build()
namespace {
#if PP_WIN_ABLATE & 64
__device__ unsigned long long pp_win_stamp_buf[256 * 8 * 8];
#define PP_STAMP(i)                                                      \
  {                                                                      \
    const unsigned long long t_ = stamp_now();                          \
    tsum[i] += t_ - tlast;                                               \
    tlast = t_;                                                          \
  }
#else
#define PP_STAMP(i)
#endif
constexpr int BK = 64;
constexpr int NW = 8, NT = 64 * NW;
constexpr int HALO = 64;                      // rows kept on either side of the tile (>= W + 1); template HL = 96: frames up to
                                              // 95 wide (the reference's own 100 x 180 clips: layer 1 is 50 x 90)
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {   // n is wave-uniform (a producer wave keeps up to 35 pieces in flight)
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

struct WinGeom {
  FastDiv dW_, dH_;
  int W, H;
  int M;         // rows of A and of C (N * T * H * W)
  int cstride;   // A row stride (elements)
  int cg;        // A channels per tap (= K / 9)
  int sign;      // +1 forward, -1 data gradient (source = m - off)
  // temporal (3,1,1) variant: a tile is ALL T frames of PB = BM / T consecutive positions of one clip
  FastDiv dBlk;  // position blocks per clip (ceil(H*W / PB): the last one may be ragged)
  FastDiv dPB;   // tile row -> (frame, position): frame = lr / PB
  int nblk, T, HW, PB;
  int nstat;     // partial rows of column statistics the launch writes (two per tile)
  FastDiv dNb;   // column blocks per row block (tile -> (row block, column block))
  int oH, oW;    // S2D: the (H, W) of dx; W, H above are then those of dy
  int Mout;      // S2D: rows of dx
};

struct WinArgs {
  const h16raw* A;
  const h16raw* Bt;
  h16raw* C;
  const h16raw* residual;
  float* colstats;
  int N, b_rows, ldb, ldc, ldr, ldstat;
  unsigned a_bytes, b_bytes;   // sizes of A and Bt: the DMAs rely on the buffer resources' range check (rows outside A, absent weights)
  // BatchNorm-backward sums of the consumer of C, taken in the epilogue (pp_igemm_desc.bnr_*)
  const h16raw* bnr_y;
  const h16raw* bnr_z;
  const float *bnr_mean, *bnr_rstd, *bnr_scale, *bnr_shift;
  int bnr_relu;
  float* bnr_partials;
  // BatchNorm apply (+ ReLU) of the PRODUCER of A, done on each window in LDS (BNA): A is then the raw conv output y and
  // the activated tensor z = relu(y * scale + shift) is never materialised
  const float *bna_scale, *bna_shift;
  int bna_relu;
};

// MT = 16-row tiles per wave (2: 256-row workgroup tile; 4: 512 rows -- narrow outputs, where a weight fragment would
// otherwise feed only two MFMAs per load and the per-K-step bookkeeping outweighs the matrix work).
// NBS = weight ring slots (3: two K-steps in flight; 2, where LDS is short: one).
// TW = temporal: the (3,1,1) stride-1 convolutions.  Their taps are +-H*W rows apart, far more than a halo can hold, so
// the tile is turned instead: all T frames of PB = BM / T consecutive positions of one clip (window row = frame * PB +
// position).  Then every tap is the window offset +-PB, frames -1 and T are the masked zero row, no halo is loaded at
// all and every activation byte is read once (the gather kernel re-read the layer-1 input 2.6 times).
// The temporal form keeps ALL weight K-steps of a tile resident in the ring (NBS >= K-steps per tile, loaded once per
// workgroup): its K-steps are short (16 MFMAs per wave with 64 output columns), and re-streaming 8 KB of weights per step
// from L2 two steps ahead left every step waiting for that DMA.  With resident weights there is no DMA and no barrier
// inside a phase: one barrier per window.
// BNR = the epilogue also accumulates the BatchNorm-backward sums (sum g, sum g * xhat) of the layer that consumes this
// output as its dz: the tile is in registers / LDS anyway, so bn_bwd_reduce's pass over dz (2 B per element of HBM
// traffic, a launch) disappears; its read of y moves here.
// STG = staggered halves (spatial form, three weight slots): see the K-step below.
// S2D = the data gradient of a (1,3,3) convolution with stride (1,2,2) as ONE launch that reads dy once.  A tile is 256 rows
// of dy; for dx position (2 i + p, 2 j + q) only the taps of parity class (p, q) contribute, each reading dy at (i + di, j + dj)
// with di, dj in {0, 1}: the window keeps the tile's rows plus W' + 1 rows AFTER it (none before), a K-step is one tap of one
// 64-channel chunk and accumulates into its class's column tiles (4 classes x WN tiles per row tile), and the epilogue
// scatters four output rows per dy row.  The per-class launches of the gather kernel this replaces each streamed dy again
// (3.5x the traffic, 0.07-0.14 of the matrix peak: profiles/r03_shapes.md).  dy's channel count need not be a multiple of
// 64: the last chunk's surplus channels read the neighbouring bytes and meet zero weights (absent DMA lanes).
// PROD = four producer waves (8, 9: weights; 10, 11: windows) issue every LDS-DMA and do nothing else; the eight
// multiplying waves issue none and never wait on vmcnt (wgrad_tw.hip measured why: a wave that multiplies is not at its
// DMA instructions when the memory pipeline has room for them).  Twelve waves = three per SIMD = 168 registers each.
template <int WN, int CC, bool RES, int MT, int NBS, bool TW, bool BNR = false, bool BNA = false, bool STG = false, bool PROD = false, int HL = HALO, bool S2D = false, int KPB = 1>
__global__ __launch_bounds__(PROD ? NT + 256 : NT, 1) void igemm_win_kernel(const WinArgs p, const WinGeom g, const int nblk_n,
                                                          const int ntiles, const int xcd_remap, const int out_nt) {
  constexpr int BM = 16 * MT * NW;
  // TRC: the MFMAs take (weights, activations) instead of (activations, weights), i.e. accumulate the TRANSPOSED tile: a lane
  // then holds four consecutive output COLUMNS of one output row (row = lane & 15, columns 4 (lane >> 4) ..), which is 8
  // contiguous bytes of C -- the epilogue stages a row tile in LDS with 18 eight-byte writes per lane instead of the 72
  // two-byte writes the other layout (four consecutive rows of one column) needs, before the same coalesced 16-byte
  // stores.  (Storing the 8 bytes straight from the accumulators was tried: 32-byte partial-line writes, step +0.5 ms.)
  // Not with BNR (its epilogue keeps a fixed 8-column chunk per lane for the BatchNorm-backward sums; bnr_built() kernels
  // keep the untransposed form).
#ifndef PP_WIN_TRANSPOSED_ACC
#define PP_WIN_TRANSPOSED_ACC 1
#endif
  constexpr bool TRC = PP_WIN_TRANSPOSED_ACC && !BNR;
  // temporal tiles with narrow outputs: no halo rows at all and THREE window buffers -- the window of the phase after
  // next is in flight too, because a phase (3 short K-steps) is far shorter than an HBM round trip
  constexpr int HALO_ = (TW && WN <= 4) ? 0 : HL;
  constexpr int HB = S2D ? 0 : HALO_, HA = HALO_;          // window rows before / after the tile's own
  constexpr int NCLS = S2D ? 4 : 1;                        // output parity classes (column-tile groups of the accumulators)
  static_assert(!S2D || (!TW && CC == 64 && !BNR && !BNA && !STG && MT == 2), "S2D: spatial form, 64-channel chunks");
  // (three window buffers in the SPATIAL narrow form as well -- the window of the phase after next in flight -- were tried
  // in round 4 for the layer-1 data gradient, whose phases of seven 16-MFMA K-steps are shorter than an HBM round trip:
  // 630-636 us against 613-623, i.e. its parked waves do not wait for windows; profiles/r04_probe_gemm_w3.log)
  constexpr int NWIN = (TW && WN <= 4) ? 3 : 2;
  constexpr int D = NWIN - 1;                                 // phases of window look-ahead
  constexpr int WROWS = BM + HB + HA;
  constexpr int BN = 16 * WN;
  constexpr int B_BYTES = BN * 128;
  // KPB = K-steps per barrier (round 4).  With 64 output columns a K-step is 16 MFMAs per wave -- 256 clocks of matrix work
  // between two workgroup barriers, each of which also exposes the LDS latency of the K-step's first fragments.  KPB = 2
  // makes a ring slot hold TWO K-steps (a "group"): one wait + barrier + DMA issue per group, and the second K-step's
  // fragment reads fly under the first one's MFMAs.  The ring logic is unchanged with "K-step" read as "group".
  constexpr int B_SLOT = KPB * B_BYTES;
  static_assert(KPB == 1 || (KPB == 2 && !TW && !STG && !BNR && !BNA), "KPB = 2: spatial form, plain epilogue");
  constexpr int XS = CC == 64 ? 128 : CC * 2 + 16;            // window row stride; 128-byte rows are XOR-swizzled
  constexpr int WIN_BYTES = WROWS * XS;
  constexpr int WPIECES = WIN_BYTES / 1024;
  static_assert(WIN_BYTES % 1024 == 0, "window = whole DMA pieces");
  // Wave roles for the LDS-DMA: vmcnt retires in order, so a wave that has a long-latency window piece (HBM) in flight
  // cannot tell that the weight slice it issued afterwards (L2) has landed.  Waves 0..3 therefore issue only weights
  // and wait for them step by step; waves 4..7 issue only window pieces -- the WHOLE next window at the start of a
  // phase -- and wait for them once per phase, so a full window (40-70 KB per CU) is in flight under the phase's
  // matrix work instead of the one or two pieces a per-step wait allows.  All eight waves multiply.
#ifndef PP_WIN_PROD_B
#define PP_WIN_PROD_B 2
#endif
  // producer waves (PROD): two for the weight slices, two for the windows (tools/probe/win_prod_sweep.sh: three + one is
  // no better for the 128- / 144-column tiles and loses the 512-row tile's gain -- its windows are 70 pieces a phase)
  // temporal form (resident weights, loaded once per workgroup): the two weight producers would idle after the prologue, so
  // ALL FOUR producers issue window pieces (DUAL) -- one wave sustains ~13 GB/s of LDS-DMA (wgrad_tw.hip's measurement) and
  // the kernel is HBM-bound.  (A single weight producer is not an option: a weight wave's pieces must lie 16 rows apart for
  // its one precomputed swizzle to hold.)
#ifndef PP_WIN_DUAL
#define PP_WIN_DUAL 1
#endif
  constexpr int NPB = TW ? 2 : PP_WIN_PROD_B, NPX = 4 - NPB;
  constexpr bool DUAL = PROD && TW && PP_WIN_DUAL;
  constexpr int NWB = PROD ? NPB : NW / 2;                    // waves in the weight role
  constexpr int NWX = PROD ? (DUAL ? 4 : NPX) : NW / 2;       // waves in the window role
  static_assert(!PROD || (!BNR && !STG), "producer waves: plain epilogue");
  constexpr int NWP = (WPIECES + NWX - 1) / NWX;              // window pieces per window wave and phase
  constexpr int NBI = (BN + 8 * NWB - 1) / (8 * NWB);         // weight pieces per weight wave and K-step (last one maybe absent)
  constexpr int NTAP = TW ? 3 : 9;
  constexpr int KC = NTAP * CC;                               // flat K of one channel chunk: (tap, channel)
  constexpr int NKC = (KC + BK - 1) / BK;                     // K-steps per chunk
  constexpr bool RW = TW;                                     // resident weights
  static_assert(NBS == 2 || NBS == 3 || (RW && NBS <= 9), "weight ring");
  constexpr int STG_STRIDE = BN * 2 + 16;
  constexpr int STG_BYTES = NW * 16 * STG_STRIDE;
  constexpr int STAT_BYTES = NW * BN * 2 * 4;
  static_assert(STG_BYTES <= WIN_BYTES && STAT_BYTES <= B_BYTES, "the epilogue stages in a window buffer / weight slot");
  static_assert(!RW || WN > 4 || STG_BYTES + STAT_BYTES <= WIN_BYTES, "resident weights: the statistics stage behind the output");
  constexpr int BNA_CH = 160;                                 // channels of the BatchNorm parameter table (BNA)
  constexpr int SMEM = NWIN * WIN_BYTES + NBS * B_SLOT + 256 + 64 + (BNA ? 2 * BNA_CH * 4 : 0);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(!BNA || (TW && CC == 48), "BNA: temporal form, 48-channel chunks");
  static_assert(!STG || (!TW && NBS == 3 && !BNA), "staggered halves: spatial form with a three-slot weight ring");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];   // one LDS object (see igemm.hip)
  unsigned char* const bring = smem + NWIN * WIN_BYTES;
  unsigned char* const zrow = smem + NWIN * WIN_BYTES + NBS * B_SLOT;    // 256 zero bytes

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // roles (wave-uniform)
  const bool is_comp = !PROD || wave < NW;
  const bool win_wave = PROD ? (DUAL ? wave >= NW : wave >= NW + NPB) : wave >= NWB;
  const bool wgt_wave = PROD ? (wave >= NW && wave < NW + NPB) : wave < NWB;
  // index inside the role (a DUAL producer has both)
  const int rwave_x = PROD ? (DUAL ? wave - NW : (win_wave ? wave - NW - NPB : 0)) : (win_wave ? wave - NWB : 0);
  const int rwave_b = PROD ? (wgt_wave ? wave - NW : 0) : (wgt_wave ? wave : 0);
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x, bid = blockIdx.x;
  auto tile_index = [&](int it) __attribute__((always_inline)) -> int {   // persistent walk, XCD-contiguous (igemm.hip)
    const int base = it * G;
    const int cnt = ntiles - base < G ? ntiles - base : G;
    if (bid >= cnt) return -1;
    const int xq = cnt >> 3, xr = cnt & 7, xcd = bid & 7;
    const int t = (xcd_remap && cnt >= 8) ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3) : bid;
    return base + t;
  };
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, (short)0, (int)p.a_bytes, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.Bt, (short)0, (int)p.b_bytes, 0x00020000);
  const int nchunk = CC == 64 ? (g.cg + CC - 1) / CC : g.cg / CC;   // (64-channel chunks: the last one may be partial, see dma_weights)

  if (tid < 64) ((unsigned*)zrow)[tid] = 0u;
  float* const bna_tab = (float*)(zrow + 256 + 64);          // [scale BNA_CH][shift BNA_CH]
  if (BNA) {
    for (int i = tid; i < BNA_CH; i += NT) {
      bna_tab[i] = i < g.cg ? p.bna_scale[i] : 0.f;
      bna_tab[BNA_CH + i] = i < g.cg ? p.bna_shift[i] : 0.f;
    }
  }
  __syncthreads();
  auto tw_row = [&](const int tile_m, const int lr) __attribute__((always_inline)) -> int {   // global row of tile row lr
    // (any frame count 2..32 since round 4: PB = 256 / T positions per tile, so up to T - 1 tile rows lie past frame T - 1,
    // and the last position block of a clip may reach past H*W: both kinds of row are dead -- no taps (setup_tile), no
    // store (the sentinel g.M fails every `m < g.M`), and no live row ever reads them as a tap)
    const int b = (int)fdiv((uint32_t)tile_m, g.dBlk);
    const int frame = (int)fdiv((uint32_t)lr, g.dPB);
    const int pos = (tile_m - b * g.nblk) * g.PB + (lr - frame * g.PB);
    return (frame < g.T && pos < g.HW) ? (b * g.T + frame) * g.HW + pos : g.M;
  };

  // ---- window DMA: piece q = rwave + 4 k lands 1 KiB lane-linear.  Everything that depends on the lane is one byte
  // offset per piece RELATIVE to the phase's first row, prepared once per workgroup; a phase adds one scalar.  Rows before
  // the tensor wrap to offsets near 2^32 and rows past it lie beyond the resource's size: the hardware lands zeros for
  // both, so there is no per-piece bounds arithmetic.  (48-channel rows: the 16 pad bytes of a 112-byte row fetch the next
  // eight channels -- bytes of a line that is fetched anyway and that no fragment reads.) -----------------------------
  // (spatial form, 64-channel rows: piece k + 1 lies 32 rows after piece k with the same swizzle, so one offset + a
  // scalar stride do)
  constexpr bool WLIN = CC == 64 && !TW;
  constexpr int NVO = WLIN ? 1 : NWP;
  unsigned w_voff[NVO];
#pragma unroll
  for (int k = 0; k < NVO; ++k) {
    const int q = rwave_x + NWX * k;
    int row, cb;
    if (CC == 64) {
      row = q * 8 + (lane >> 3);
      cb = ((lane & 7) ^ swz(row)) * 16;            // slot (lane & 7) holds chunk slot ^ swz(row)
    } else {
      const int o = q * 1024 + lane * 16;
      row = o / XS;
      cb = o - row * XS;
    }
    const int lr = row - HB;
    const int lfr = TW ? (int)fdiv((uint32_t)(lr < 0 ? 0 : lr), g.dPB) : 0;
    const int rel = TW ? lfr * g.HW + (lr - lfr * g.PB) : lr;
    w_voff[k] = (unsigned)(rel * g.cstride * 2 + cb);
  }
  const unsigned w_step = WLIN ? (unsigned)(8 * NWX * g.cstride * 2) : 0u;   // bytes between a wave's consecutive pieces
  // temporal form with a halo (kept only as staging room for wide outputs): its pieces are never read, so never fetched
  auto piece_live = [&](const int k) __attribute__((always_inline)) -> bool {
    const int q = rwave_x + NWX * k;
    if (TW && HALO_ > 0) return q * 1024 >= HALO_ * XS && (q + 1) * 1024 <= (HALO_ + BM) * XS;
    return q < WPIECES;
  };
  bool abl_first = true;               // (ablations: still in the workgroup's first tile)
  // byte offset of the first row of (tile, chunk)'s window
  auto phase_base = [&](const int tile_, const int chunk_) __attribute__((always_inline)) -> unsigned {
    const int mb_ = (int)fdiv((uint32_t)tile_, g.dNb);
    int row0;
    if (TW) {
      const int b = (int)fdiv((uint32_t)mb_, g.dBlk);
      row0 = b * g.T * g.HW + (mb_ - b * g.nblk) * g.PB;
    } else {
      row0 = mb_ * BM;
    }
    return (unsigned)(row0 * g.cstride + chunk_ * CC) * 2u;
  };
  auto dma_window_piece = [&](const int k, unsigned char* wbuf, const unsigned sbase) __attribute__((always_inline)) {
    const int q = rwave_x + NWX * k;
    if ((ABL & 2) && !abl_first) return;
    if (piece_live(k)) lds_dma16(rsA, wbuf + q * 1024, WLIN ? sbase + (unsigned)k * w_step + w_voff[0] : sbase + w_voff[k]);
  };
  // pieces this (window) wave owns
  int npieces = 0;
#pragma unroll
  for (int k = 0; k < NWP; ++k) npieces += piece_live(k) ? 1 : 0;
  // PBNA: the fused BatchNorm apply (BNA) done by the PRODUCER waves.  Round 2's form has the eight multiplying waves run
  // an LDS pass over each window between two barriers of their own (+100 us on the layer-1 temporal forward: 414 vs 314 us).
  // Here every producer activates exactly the 16-byte slots it fetched itself -- its pieces of the NEXT phase's window, as
  // soon as its own vmcnt says they have landed, while the multiplying waves are in the current phase's K-steps: no wave
  // reads another wave's DMA results, so no barrier is added, and the multiplying waves never see a raw window.
  // MEASURED (round 4, tools/probe/bna_tw.py, bit-identical outputs): 461-467 us against 414 us -- four producer waves that also
  // run the LDS pass (seven 16-byte slots per lane and phase: five reads + a write each) are late with the next window's DMAs
  // and become the critical path of a phase whose three K-steps are short.  Built only with -DPP_WIN_PBNA=1.
#ifndef PP_WIN_PBNA
#define PP_WIN_PBNA 0
#endif
  constexpr bool PBNA = PP_WIN_PBNA && BNA && PROD && DUAL && NWIN == 3;
  constexpr int SPR = XS / 16;                               // 16-byte slots per window row (the last one is padding)
  int bna_c8[PBNA ? NWP : 1];
  if (PBNA) {
#pragma unroll
    for (int k = 0; k < NWP; ++k) {
      const int q = rwave_x + NWX * k;
      const int c8 = (q * 64 + lane) % SPR;
      bna_c8[k] = (q < WPIECES && c8 < CC / 8) ? c8 : -1;
    }
  }
  auto bna_own = [&](unsigned char* wbuf, const int chunk_) __attribute__((always_inline)) {
    if constexpr (PBNA) {
      constexpr int NB1 = (NWP + 1) / 2;                       // two batches: all reads of a batch, one wait, the writes
#pragma unroll
      for (int b0 = 0; b0 < NWP; b0 += NB1) {
        u32x4 vv[NB1], s0[NB1], s1[NB1], h0[NB1], h1[NB1];
        unsigned adr[NB1];
#pragma unroll
        for (int k = b0; k < b0 + NB1 && k < NWP; ++k) {
          const int c8 = bna_c8[k] < 0 ? 0 : bna_c8[k];
          adr[k - b0] = (unsigned)(uintptr_t)(lds_ptr)(wbuf + (rwave_x + NWX * k) * 1024 + lane * 16);
          const unsigned t = (unsigned)(uintptr_t)(lds_ptr)(bna_tab + chunk_ * CC + c8 * 8);
          asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                       "ds_read_b128 %3, %6 offset:%7\n\tds_read_b128 %4, %6 offset:%8"
                       : "=&v"(vv[k - b0]), "=&v"(s0[k - b0]), "=&v"(s1[k - b0]), "=&v"(h0[k - b0]), "=&v"(h1[k - b0])
                       : "v"(bna_c8[k] < 0 ? (unsigned)(uintptr_t)(lds_ptr)zrow : adr[k - b0]), "v"(t), "n"(BNA_CH * 4), "n"(BNA_CH * 4 + 16)
                       : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < NB1; ++k) asm volatile("" : "+v"(vv[k]), "+v"(s0[k]), "+v"(s1[k]), "+v"(h0[k]), "+v"(h1[k]));
#pragma unroll
        for (int k = b0; k < b0 + NB1 && k < NWP; ++k) {
          if (bna_c8[k] >= 0) {
            const int i = k - b0;
            float x[8];
            unpack8(make_uint4(vv[i][0], vv[i][1], vv[i][2], vv[i][3]), x);
            const float sc[8] = {__uint_as_float(s0[i][0]), __uint_as_float(s0[i][1]), __uint_as_float(s0[i][2]), __uint_as_float(s0[i][3]),
                                 __uint_as_float(s1[i][0]), __uint_as_float(s1[i][1]), __uint_as_float(s1[i][2]), __uint_as_float(s1[i][3])};
            const float sh[8] = {__uint_as_float(h0[i][0]), __uint_as_float(h0[i][1]), __uint_as_float(h0[i][2]), __uint_as_float(h0[i][3]),
                                 __uint_as_float(h1[i][0]), __uint_as_float(h1[i][1]), __uint_as_float(h1[i][2]), __uint_as_float(h1[i][3])};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float zv = x[q] * sc[q] + sh[q];
              x[q] = (p.bna_relu && !(zv > 0.f)) ? 0.f : zv;
            }
            const uint4 o = pack8(x);
            const u32x4 ov = {o.x, o.y, o.z, o.w};
            asm volatile("ds_write_b128 %0, %1" ::"v"(adr[i]), "v"(ov) : "memory");
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // ---- weight DMA (waves 0..3): rows 8 wave + (lane >> 3) + 32 i, slot lane & 7 holds K chunk (lane & 7) ^ swz(row).
  // Source offset = row part (per tile) + K part of the K-step (per workgroup; flat (tap, channel) K inside a chunk, so a
  // 48-channel chunk's K-step straddles taps and the part is per lane) + chunk (scalar).  0x80000000 marks "absent"
  // (row past the matrix, K past the last tap): the sum stays beyond the resource's size and zeros land. ------------
  constexpr unsigned ABSENT = 0x80000000u, ABSENT_K = 0x40000000u;   // (their sum does not wrap; Bt is < 2^30 bytes)
  const int brow0 = 8 * rwave_b + (lane >> 3);
  const int kqB = (lane & 7) ^ swz(brow0);
  const bool b_last = 8 * rwave_b + 8 * NWB * (NBI - 1) < BN;   // does this wave own a piece in the last weight pass
  const int nB = b_last ? NBI : NBI - 1;
  unsigned bvoff[NBI];
  // (64-channel chunks: K-step j is tap j, channels kqB * 8 ..: the lane's part moves into the row offset)
  constexpr int NKV = CC == 64 ? 1 : NKC;
  unsigned kvoff[NKV];
#pragma unroll
  for (int j = 0; j < NKV; ++j) {
    const int kf = j * BK + kqB * 8;
    const int tap = kf / CC, c = kf - tap * CC;
    kvoff[j] = tap < NTAP ? (unsigned)(tap * g.cg + c) * 2u : ABSENT_K;
  }
  auto dma_weights = [&](unsigned char* slot, const int chunk_, const int j) __attribute__((always_inline)) {
    if ((ABL & 1) && !abl_first) return;
    // 64-channel chunks of a tensor whose channel count is not a multiple of 64 (464, 928: the mid-planes of layers 3 / 4;
    // 240 in the S2D form): the last chunk's surplus channels read the bytes that follow the row's data -- the next row, or
    // zeros past the tensor -- and meet ABSENT (zero) weights here
    const bool k_absent = CC == 64 && chunk_ * CC + kqB * 8 >= g.cg;
    const unsigned koff = (CC == 64 ? ((j < NTAP && !k_absent) ? (unsigned)(j * g.cg) * 2u + kvoff[0] : ABSENT_K) : kvoff[j]) + (unsigned)(chunk_ * CC) * 2u;
    unsigned char* dst = slot + (8 * rwave_b) * 128;
#pragma unroll
    for (int i = 0; i < NBI; ++i)
      if (i < NBI - 1 || b_last) lds_dma16(rsB, dst + 8 * NWB * i * 128, bvoff[i] + koff);
  };

  // all K-steps of group `gi` (KPB consecutive K-steps, the last group of a chunk may be short) into ring slot `slot`;
  // returns the number of DMA instructions this wave issued
  constexpr int NKC_ = (NTAP * CC + BK - 1) / BK;
  constexpr int NG = (NKC_ + KPB - 1) / KPB;                 // groups per chunk
  auto dma_group = [&](unsigned char* slot, const int chunk_, const int gi) __attribute__((always_inline)) -> int {
    int n = 0;
#pragma unroll
    for (int t = 0; t < KPB; ++t)
      if (gi * KPB + t < NKC_) { dma_weights(slot + t * B_BYTES, chunk_, gi * KPB + t); n += nB; }
    return n;
  };

  // ---- per-tile state -------------------------------------------------------------------------------------------
  int mb = 0, nb = 0;
  unsigned vmask[MT];    // bit t: tap t of fragment row (wave * 16 MT + mt * 16 + fr) lies inside the image
  auto setup_tile = [&](const int tile) __attribute__((always_inline)) {
    mb = (int)fdiv((uint32_t)tile, g.dNb);
    nb = tile - mb * nblk_n;
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int brow = brow0 + 8 * NWB * i;
      const int n = nb * BN + brow;
      bvoff[i] = (brow < BN && n < p.b_rows) ? (unsigned)(n * p.ldb) * 2u : ABSENT;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (TW) {
        const int lr = wave * (16 * MT) + mt * 16 + fr;
        const int frame = (int)fdiv((uint32_t)lr, g.dPB);
        const int b = (int)fdiv((uint32_t)mb, g.dBlk);
        const bool live = frame < g.T && (mb - b * g.nblk) * g.PB + (lr - frame * g.PB) < g.HW;     // (dead rows: see tw_row)
        unsigned v = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) v |= (unsigned)((unsigned)(frame + g.sign * (t - 1)) < (unsigned)g.T) << t;
        vmask[mt] = live ? v : 0u;
        continue;
      }
      const int m = mb * BM + wave * (16 * MT) + mt * 16 + fr;
      const uint32_t q1 = fdiv((uint32_t)m, g.dW_);
      const int w = m - (int)q1 * g.W;
      const int h = (int)q1 - (int)fdiv(q1, g.dH_) * g.H;
      unsigned v = 0;
      if (S2D) {      // tap (dh, dw) reads dy at (h + (dh == 0), w + (dw == 0))
#pragma unroll
        for (int t = 0; t < 9; ++t) v |= (unsigned)((h + (t / 3 == 0 ? 1 : 0) < g.H) & (w + (t % 3 == 0 ? 1 : 0) < g.W)) << t;
        vmask[mt] = m < g.M ? v : 0u;
        continue;
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int sh = g.sign * (t / 3 - 1), sw = g.sign * (t % 3 - 1);
        v |= (unsigned)(((unsigned)(h + sh) < (unsigned)g.H) & ((unsigned)(w + sw) < (unsigned)g.W)) << t;
      }
      vmask[mt] = m < g.M ? v : 0u;
    }
  };

  f32x4 acc[MT][NCLS * WN];
  // ---- fragment addressing, prepared once per workgroup.  Window row of fragment row (mt, fr) = rowA[mt] + tap offset;
  // 64-channel chunks: a K-step is one tap (scalar offset), rows are XOR-swizzled; 48-channel chunks: the lane's k-octet
  // of K-step j, half ks has its own (tap, channel) -- byte offset and tap bit per (j, ks) ----------------------------
  int lutv[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
    lutv[t] = TW ? g.sign * (t - 1) * g.PB : S2D ? (t / 3 == 0 ? g.W : 0) + (t % 3 == 0 ? 1 : 0) : g.sign * ((t / 3 - 1) * g.W + (t % 3 - 1));
  int rowA[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) rowA[mt] = wave * (16 * MT) + mt * 16 + fr + HB;
  unsigned aoff[CC == 64 ? 1 : NKC][2], abit[CC == 64 ? 1 : NKC][2];
  if (CC != 64) {
#pragma unroll
    for (int j = 0; j < NKC; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k0 = j * BK + ks * 32;                        // the k-octets of this half start here
        const int t0 = k0 / CC, r = k0 - t0 * CC + fq * 8;      // (fq * 8 <= 24 < CC: at most one tap further)
        const bool hi = r >= CC;
        const int tap = t0 + (hi ? 1 : 0), c = hi ? r - CC : r;
        const int l0 = t0 < NTAP ? lutv[t0 < NTAP ? t0 : 0] : 0, l1 = t0 + 1 < NTAP ? lutv[t0 + 1 < NTAP ? t0 + 1 : 0] : 0;
        aoff[j][ks] = (unsigned)((hi ? l1 : l0) * XS + c * 2);
        abit[j][ks] = tap < NTAP ? 1u << tap : 0u;
      }
  }
  const unsigned zaddr = (unsigned)(uintptr_t)(lds_ptr)(zrow + fr * 16);
  const unsigned bfr0 = (unsigned)(fr * 128 + ((fq ^ swz(fr)) << 4));     // weight fragment (column fr, half 0) in a slot
  // K-step j of a chunk (j is a compile-time constant after unrolling)
  // `issue_dmas` runs between the fragment requests and the MFMAs: an LDS-DMA instruction takes 100+ cycles to issue, the
  // fragments about as long to arrive
  // `sync` = this K-step's wait + barrier.  Staggered halves (STG): the eight waves of the lockstep form request their
  // fragments together (176 KB per K-step: the LDS is busy, the matrix pipes idle), then multiply together (LDS idle) and
  // meet at the next barrier.  With `early` (waves 4-7 from a phase's second K-step on) a wave requests its fragments
  // BEFORE the barrier: while it waits there, its SIMD partner (waves w and w + 4 share a SIMD) is still multiplying, and
  // after the barrier it multiplies while the partner reads -- reads and MFMAs of the two halves alternate instead of
  // coinciding.  What that needs: the weights of step s + 1 have landed when the barrier of step s opens (the weight waves
  // wait for ALL their DMAs, not for all but the youngest batch), and a phase's first K-step stays in lockstep (its window
  // is published by that step's barrier).
  auto compute = [&](auto comp_c, const int j, const unsigned char* win, const unsigned char* bslot, auto issue_dmas, auto sync, const bool early) __attribute__((always_inline)) {
    // every fragment of the K-step is requested before the first MFMA: written as "load one weight fragment, use it"
    // hipcc keeps ONE fragment register and waits lgkmcnt(0) before every MFMA pair, i.e. one exposed LDS round trip
    // (~100 cycles) per 32 cycles of matrix work
    // Wide outputs (WN >= 8): the second half's weight fragments take the registers of the first half's, each requested
    // right after the MFMAs that consumed its predecessor (18 x 4 registers of weights per half would not fit beside
    // the 72 accumulators without spilling).
    constexpr bool REUSE_B = WN >= 8;
    h16x8 af[2][MT], bfm[REUSE_B ? 1 : 2][WN];
    if constexpr (!decltype(comp_c)::value) {       // producer wave: this K-step's wait, barrier and DMAs
      sync();
      issue_dmas(-1);
      return;
    }
    if (ABL & 4) {
      sync();
      issue_dmas(-1);
      return;
    }
    if (!early) {
      sync();
      issue_dmas(0);
    }
    const unsigned wbase = (unsigned)(uintptr_t)(lds_ptr)win;
    // (laundered: otherwise hipcc prepares the 2 x WN fragment addresses of every ring slot once per workgroup -- 54
    // registers that spill under the producer form's 168-register budget -- instead of two bases + immediate offsets)
    unsigned bbase = (unsigned)(uintptr_t)(lds_ptr)bslot + bfr0;
    asm volatile("" : "+v"(bbase));
    unsigned bbase1 = bbase ^ 64u;
    asm volatile("" : "+v"(bbase1));
    auto load_a = [&](const int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        unsigned a;
        bool ok;
        if (CC == 64) {
          // (the row is laundered through an empty asm: otherwise hipcc hoists the nine taps' addresses of every row tile
          // out of the chunk loop and the 36 registers they occupy spill)
          int r0 = rowA[mt];
          asm volatile("" : "+v"(r0));
          const int wrow = r0 + lutv[j < NTAP ? j : 0];
          a = wbase + (unsigned)(wrow * XS + (((ks * 4 + fq) ^ swz(wrow)) << 4));
          ok = j < NTAP && ((vmask[mt] >> j) & 1u) != 0u;
        } else {
          a = wbase + (unsigned)(rowA[mt] * XS) + aoff[j][ks];
          ok = (vmask[mt] & abit[j][ks]) != 0u;               // (the K tail has no bit)
        }
        af[ks][mt] = *(const h16x8*)(lds_ptr)(uintptr_t)(ok ? a : zaddr);
      }
    };
    auto load_b = [&](const int ks, const int jn) __attribute__((always_inline)) -> h16x8 {
      return *(const h16x8*)(lds_ptr)(uintptr_t)((ks ? bbase1 : bbase) + jn * 2048);
    };
    load_a(0);
#pragma unroll
    for (int jn = 0; jn < WN; ++jn) bfm[0][jn] = load_b(0, jn);
    load_a(1);
    if (!REUSE_B) {
#pragma unroll
      for (int jn = 0; jn < WN; ++jn) bfm[REUSE_B ? 0 : 1][jn] = load_b(1, jn);
    }
    if (early) {
      __builtin_amdgcn_sched_barrier(0);     // (the requests stay in front of the barrier)
      sync();
      issue_dmas(0);
    }
    issue_dmas(1);
    if (ABL & 16) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(af[ks][mt]));
#pragma unroll
        for (int jn = 0; jn < WN; ++jn) asm volatile("" ::"v"(bfm[REUSE_B ? 0 : ks][jn]));
      }
      issue_dmas(2);
      issue_dmas(3);
      return;
    }
    // S2D: tap j = (dh, dw) belongs to output parity class ((dh + 1) & 1, (dw + 1) & 1) -- its own column tiles
    const int c0 = S2D ? ((((j / 3) + 1) & 1) * 2 + (((j % 3) + 1) & 1)) * WN : 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int jn = 0; jn < WN; ++jn) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][c0 + jn] = TRC ? PP_MFMA16(bfm[REUSE_B ? 0 : ks][jn], af[ks][mt], acc[mt][c0 + jn], 0, 0, 0)
                                 : PP_MFMA16(af[ks][mt], bfm[REUSE_B ? 0 : ks][jn], acc[mt][c0 + jn], 0, 0, 0);
        if (REUSE_B && ks == 0) {          // (pinned: left alone, hipcc sinks these reads to just before their MFMAs)
          bfm[0][jn] = load_b(1, jn);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      issue_dmas(2 + ks);
    }
  };

  // ---- epilogue (plain bf16 store, optional residual add, optional BatchNorm column statistics); igemm.hip's ------
  const int ncols_store = (p.N + 7) & ~7;
  // S2D: dy row m (tile space) and parity class cls -> dx row, or -1 where that position does not exist
  auto s2d_row = [&](const int m, const int cls) __attribute__((always_inline)) -> int {
    const uint32_t q1 = fdiv((uint32_t)m, g.dW_);
    const int j = m - (int)q1 * g.W;
    const uint32_t nt = fdiv(q1, g.dH_);
    const int i = (int)q1 - (int)nt * g.H;
    const int u = 2 * i + (cls >> 1), v = 2 * j + (cls & 1);
    return (m < g.M && u < g.oH && v < g.oW) ? ((int)nt * g.oH + u) * g.oW + v : -1;
  };
  auto epilogue = [&](const int mb_e, const int nb_e, unsigned char* const ebuf, unsigned char* const sbuf, const int cls) __attribute__((always_inline)) {
    const int m_wave = mb_e * BM + wave * (16 * MT);
    // temporal form: the rows' global indices (tw_row: two exact divisions each since the ragged form of round 4) are computed
    // HERE.  They depend on the tile and the lane only, so the compiler computed all MT x NIT of them before the K loop and,
    // with 168 registers per wave in the producer form, spilled them: +11 exposed scratch loads per tile, 315 -> 383 us on
    // the 144-column layer-1 data gradient (profiles/r04_probe_tw_regress.log).  Two opaque copies keep them behind the loop.
    int mb_t = mb_e, lr0 = wave * (16 * MT);
    if (TW) {
      asm volatile("" : "+s"(mb_t));
      asm volatile("" : "+v"(lr0));
    }
    const int ca = cls * WN;                          // this class's column tiles of the accumulators (S2D; else 0)
    const int m_lim = S2D ? g.Mout : g.M;
    unsigned char* stg = ebuf + wave * 16 * STG_STRIDE;
    unsigned char* stg_w = stg + (fq * 4) * STG_STRIDE + fr * 2;
    // BNR: a lane keeps ONE 8-column chunk for the whole tile (RPI rows in flight per wave pass), so that its sixteen
    // running sums and the chunk's BatchNorm parameters live in registers; otherwise chunks are dealt lane-linearly
    constexpr int CPR = 2 * WN;                       // 8-column chunks per staged row
    constexpr int RPI = 64 / CPR;                     // rows per pass with a fixed chunk per lane
    constexpr int NIT = BNR ? (16 + RPI - 1) / RPI : (32 * WN + 63) / 64;
    const int b_rsub = lane / CPR, b_ch = lane % CPR;
    float s1[8], s2[8], mu[8], rs[8], sc[8], sh[8];
    // the consumer's y chunks of the WHOLE tile are requested up front (MT x NIT 16-byte loads per lane in flight): read
    // one by one where they are used, every chunk exposed a full HBM round trip inside an epilogue nothing overlaps
    // (measured: the step 2.8 ms slower with the sums fused than with the separate pp_bn_bwd_reduce pass)
    uint4 ypre[BNR ? MT * NIT : 1];
    // likewise the residual rows (RES): one 16-byte load per staged chunk, all in flight before the first row tile is staged
    // (read where they are added, each exposed an HBM round trip: +120-140 us on the layer-1 data gradient)
    // (512-row tiles, MT = 4: the four row tiles' chunks beside 64 accumulators spill under the producer form's 168
    // registers and cost more than they save -- there only the next row tile's chunks are in flight)
    constexpr int RPF = MT <= 2 ? MT : 1;             // row tiles of residual chunks requested ahead
    uint4 rpre[RES ? MT * NIT : 1];
    auto fetch_res = [&](const int mt) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cid = lane + 64 * it;
        const int row = BNR ? b_rsub + RPI * it : cid / CPR;
        const int ch = BNR ? b_ch : cid % CPR;
        const bool live = BNR ? (b_rsub < RPI && row < 16) : cid < 32 * WN;
        int m = TW ? tw_row(mb_t, lr0 + mt * 16 + (live ? row : 0)) : m_wave + mt * 16 + row;
        if (S2D) { m = s2d_row(m, cls); if (m < 0) m = m_lim; }
        const int col = nb_e * BN + ch * 8;
        rpre[mt * NIT + it] = (live && m < m_lim && col < ncols_store) ? *(const uint4*)(p.residual + (long long)m * p.ldr + col) : make_uint4(0, 0, 0, 0);
      }
    };
    if (RES) {
#pragma unroll
      for (int mt = 0; mt < RPF; ++mt) fetch_res(mt);
    }
    if (BNR) {
      const int c0 = nb_e * BN + b_ch * 8;
      const bool cok = b_rsub < RPI && c0 < ncols_store;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = b_rsub + RPI * it;
          const bool live = cok && row < 16;
          const int m = TW ? tw_row(mb_t, lr0 + mt * 16 + (live ? row : 0)) : m_wave + mt * 16 + row;
          ypre[mt * NIT + it] = (live && m < g.M) ? *(const uint4*)(p.bnr_y + (long long)m * p.ldc + c0) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        s1[q] = 0.f; s2[q] = 0.f;
        mu[q] = cok ? p.bnr_mean[c0 + q] : 0.f;
        rs[q] = cok ? p.bnr_rstd[c0 + q] : 0.f;
        sc[q] = cok ? p.bnr_scale[c0 + q] : 0.f;
        sh[q] = cok ? p.bnr_shift[c0 + q] : 0.f;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (RES && mt + RPF < MT) fetch_res(mt + RPF);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        if (TRC) {        // four consecutive columns of row fr: one 8-byte write (acc[mt][j][r] = C[row fr][j * 16 + 4 fq + r])
          const u32x2 w = {pack2(acc[mt][ca + j][0], acc[mt][ca + j][1]), pack2(acc[mt][ca + j][2], acc[mt][ca + j][3])};
          *(u32x2*)(stg + fr * STG_STRIDE + j * 32 + fq * 8) = w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) *(h16raw*)(stg_w + r * STG_STRIDE + j * 32) = f2h(acc[mt][ca + j][r]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // all of the pass's staged chunks are requested, then ONE wait (one wait per chunk exposed an LDS round trip each).
      // (inline asm: a plain LDS load here makes hipcc drain the DMAs in flight, see igemm.hip)
      u32x4 vvs[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cid = lane + 64 * it;
        const int row = BNR ? b_rsub + RPI * it : cid / CPR;
        const int ch = BNR ? b_ch : cid % CPR;
        const bool live = BNR ? (b_rsub < RPI && row < 16) : cid < 32 * WN;
        const unsigned a = (unsigned)(uintptr_t)(lds_ptr)(live ? stg + row * STG_STRIDE + ch * 16 : stg);
        asm volatile("ds_read_b128 %0, %1" : "=v"(vvs[it]) : "v"(a) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vvs[it]));
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cid = lane + 64 * it;
        const int row = BNR ? b_rsub + RPI * it : cid / CPR;
        const int ch = BNR ? b_ch : cid % CPR;
        const bool live = BNR ? (b_rsub < RPI && row < 16) : cid < 32 * WN;
        int m = TW ? tw_row(mb_t, lr0 + mt * 16 + (live ? row : 0)) : m_wave + mt * 16 + row;
        if (S2D) { m = s2d_row(m, cls); if (m < 0) m = m_lim; }
        const int col = nb_e * BN + ch * 8;
        if (live && m < m_lim && col < ncols_store) {
          const u32x4 vv = vvs[it];
          uint4 v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
          if (RES) {
            const uint4 rv = rpre[mt * NIT + it];
            float x[8], y[8];
            unpack8(v, x);
            unpack8(rv, y);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] += y[q];
            v = pack8(x);
          }
          if (ABL & 32) {
            asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          } else if (out_nt) {      // non-temporal: the tile is next read by another kernel, long after it has left the L2
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, (u32x4*)(p.C + (long long)m * p.ldc + col));
          } else {
            *(uint4*)(p.C + (long long)m * p.ldc + col) = v;
          }
          if (BNR) {   // the sums use the bf16 value just stored: exactly what pp_bn_bwd_reduce would read back as dz
            float d[8], yy[8], zz[8];
            unpack8(v, d);
            unpack8(ypre[mt * NIT + it], yy);
            if (p.bnr_relu) {
              if (p.bnr_z) unpack8(*(const uint4*)(p.bnr_z + (long long)m * p.ldc + col), zz);
              else
#pragma unroll
                for (int q = 0; q < 8; ++q) zz[q] = yy[q] * sc[q] + sh[q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const float gq = (p.bnr_relu && !(zz[q] > 0.f)) ? 0.f : d[q];
              s1[q] += gq;
              s2[q] += gq * ((yy[q] - mu[q]) * rs[q]);
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if (BNR) {
      // lanes b_ch, b_ch + CPR, ... hold partial sums of the same chunk: fold them onto the first; then the eight waves
      // meet in LDS and every 256 rows of the tile become one partial row (fixed order: deterministic)
      float* statbuf = (float*)sbuf;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float a = s1[q], b = s2[q];
        if (CPR == 8) {          // lanes l, l ^ 8 (row rotate by 8 on the VALU), then the four 16-lane rows
          a += dpp_f<0x128>(a); b += dpp_f<0x128>(b);
          a = sum_rows4(a); b = sum_rows4(b);
        } else if (CPR == 16) {
          a = sum_rows4(a); b = sum_rows4(b);
        } else {
          float ta = a, tb = b;
#pragma unroll
          for (int r = 1; r < RPI; ++r) { ta += __shfl(a, b_ch + r * CPR); tb += __shfl(b, b_ch + r * CPR); }
          a = ta; b = tb;
        }
        if (lane < CPR) {
          statbuf[(wave * BN + b_ch * 8 + q) * 2 + 0] = a;
          statbuf[(wave * BN + b_ch * 8 + q) * 2 + 1] = b;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      constexpr int HALVES = BM / 256, WPH = NW / HALVES;     // 256-row partial blocks per tile, waves per block
      for (int idx = tid; idx < BN * HALVES; idx += NT) {
        const int c = idx % BN, h = idx / BN;
        const int n = nb_e * BN + c;
        const long long prow = (long long)mb_e * HALVES + h;
        if (n < p.ldc && prow * 256 < g.M) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int w = 0; w < WPH; ++w) {
            a += statbuf[((WPH * h + w) * BN + c) * 2 + 0];
            b += statbuf[((WPH * h + w) * BN + c) * 2 + 1];
          }
          p.bnr_partials[(prow * 2 + 0) * p.ldc + n] = a;
          p.bnr_partials[(prow * 2 + 1) * p.ldc + n] = b;
        }
      }
    }
    if (MT == 2 && !S2D && p.colstats) {   // per-column sum / sum of squares per 128 output rows, deterministic (igemm.hip)
      float* statbuf = (float*)sbuf;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        if (TRC) {
          // a column's 32 rows of this wave: two row tiles in the lane, sixteen lanes of a DPP row (same fq) across
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a0 = acc[0][j][r], a1 = acc[MT == 2 ? 1 : 0][j][r];
            float s1 = a0 + a1, s2 = a0 * a0 + a1 * a1;
            s1 += dpp_f<0xB1>(s1);  s2 += dpp_f<0xB1>(s2);     // lane ^ 1
            s1 += dpp_f<0x4E>(s1);  s2 += dpp_f<0x4E>(s2);     // lane ^ 2
            s1 += dpp_f<0x141>(s1); s2 += dpp_f<0x141>(s2);    // the other quad of the half row
            s1 += dpp_f<0x140>(s1); s2 += dpp_f<0x140>(s2);    // the other half of the 16-lane row
            if (fr == 0) {
              statbuf[(wave * BN + j * 16 + 4 * fq + r) * 2 + 0] = s1;
              statbuf[(wave * BN + j * 16 + 4 * fq + r) * 2 + 1] = s2;
            }
          }
          continue;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < (MT == 2 ? 2 : 0); ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[mt][j][r];
            s1 += v;
            s2 += v * v;
          }
        s1 = sum_rows4(s1);
        s2 = sum_rows4(s2);
        if (fq == 0) {
          statbuf[(wave * BN + j * 16 + fr) * 2 + 0] = s1;
          statbuf[(wave * BN + j * 16 + fr) * 2 + 1] = s2;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int idx = tid; idx < BN * (NW / 4); idx += NT) {
        const int c = idx % BN, h = idx / BN;
        const int n = nb_e * BN + c;
        const long long prow = (long long)mb_e * (NW / 4) + h;
        if (n < p.ldstat && prow < g.nstat) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            a += statbuf[((4 * h + w) * BN + c) * 2 + 0];
            b += statbuf[((4 * h + w) * BN + c) * 2 + 1];
          }
          p.colstats[(prow * 2 + 0) * p.ldstat + n] = a;
          p.colstats[(prow * 2 + 1) * p.ldstat + n] = b;
        }
      }
    }
  };

  // ---- tile loop.  Phase = (tile, chunk); its window sits in buffer (phase counter mod NWIN).  K-step j of a phase:
  //   wait for every DMA except the batch issued by the previous K-step; barrier; issue the weights of the K-step
  //   NBS - 1 ahead and a share of the window of the phase D ahead; multiply.  The K-steps of a chunk are unrolled: which
  //   pieces, which tap, which fragment offsets are compile-time, and a K-step costs its MFMAs, its fragment reads and a
  //   few dozen other instructions (it used to cost several hundred: cursors, divisions and a switch per piece).
  // The loop is instantiated per role (PROD): the multiplying waves' copy holds no DMA tables and never touches vmcnt, the
  // producers' copy no accumulators or fragments -- as ONE body the two sets of registers were live together and the
  // 168-register budget of twelve waves spilled.  Without PROD there is one copy and every wave plays both parts.
  auto main_loop = [&](auto comp_c) __attribute__((always_inline)) {
  constexpr bool COMP = decltype(comp_c)::value;
  constexpr bool DMA = !PROD || !COMP;
  int it = 0;
  int tile = tile_index(0);
  if (tile < 0) return;
  setup_tile(tile);
  int wsel = 0;                       // window buffer of the current phase
  int bsl = 0;                        // weight ring slot of the current K-step
  const int S = nchunk * NKC;         // K-steps per tile
  // (PMC note: the 512-row 48-channel-chunk variant fetches 1.43 GB for the 0.93-GB layer-1 data-gradient input, the
  // 256-row one 0.95 GB: consecutive chunks share 128-byte lines of the 288-byte rows, and 32 CUs x 2 x 70 KB of windows
  // in flight no longer fit one XCD's 4-MB L2.  It is still the faster of the two: profiles/r02_pmc_traffic.md.)
  // the phase `ahead` phases after (it_, chunk_): false past the last one, else its window's base offset
  auto phase_ahead = [&](const int it_, const int chunk_, const int ahead, const int cur_tile, unsigned& sbase) __attribute__((always_inline)) -> bool {
    int pc = chunk_ + ahead, pit = it_;
    while (pc >= nchunk) { pc -= nchunk; ++pit; }
    const int ptile = pit == it_ ? cur_tile : tile_index(pit);
    if (ptile < 0) return false;
    sbase = phase_base(ptile, pc);
    return true;
  };
  constexpr int LA = RW ? 0 : NBS - 1;                          // weight K-steps in flight ahead of the multiply
  constexpr int PPK = TW ? NWP : (NWP + NKC - 1) / NKC;       // window pieces per window wave and K-step
  int last_win = 0;                   // window pieces this (window) wave issued at the previous phase start
  if (DMA && wgt_wave) {
    if (RW) {
      for (int c = 0; c < nchunk; ++c)      // (the launcher checked S <= NBS; same N block for every tile)
#pragma unroll
        for (int j = 0; j < NKC; ++j) dma_weights(bring + (c * NKC + j) * B_BYTES, c, j);
    } else {
#pragma unroll
      for (int a = 0; a < LA; ++a) dma_group(bring + a * B_SLOT, 0, a);      // (NG >= 4 > LA)
    }
  }
  if (DMA && win_wave) {        // (a DUAL producer: after its share of the resident weights; the first wait drains both)
#pragma unroll
    for (int a = 0; a < D; ++a) {
      unsigned sb = 0;
      last_win = 0;
      if (phase_ahead(0, 0, a, tile, sb)) {
#pragma unroll
        for (int k = 0; k < NWP; ++k) dma_window_piece(k, smem + a * WIN_BYTES, sb);
        last_win = npieces;
      }
    }
    if (PBNA) {       // window 0 (and this wave's resident weights) have landed once only window 1 is in flight
      wait_vmcnt_dyn(last_win);
      bna_own(smem, 0);
    }
  }
  int last_batch = 0;                 // weight DMAs this (weight) wave issued in the previous iteration
#if PP_WIN_ABLATE & 64
  auto stamp_now = [&]() __attribute__((always_inline)) -> unsigned long long {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = stamp_now();
  const unsigned long long tbegin = tlast;
#endif
  // (a tile's first waits also cover the previous epilogue's stores.  They are younger than the DMAs waited for, so a
  // counted wait could leave them in flight: measured no faster.  Neither is leaving the last K-steps of a phase free of
  // window pieces so that the youngest has longer to land.)
  bool drain = true;                  // first step of a tile: wait for everything
  auto slot_ahead = [&](const int sl) __attribute__((always_inline)) -> int {    // ring slot LA steps after slot sl
    return NBS == 3 ? (sl >= 1 ? sl - 1 : 2) : (sl ^ 1);
  };
  auto next_slot = [&](int sl) __attribute__((always_inline)) { return sl + 1 == NBS ? 0 : sl + 1; };
  while (true) {
    if constexpr (COMP) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NCLS * WN; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int next_tile = tile_index(it + 1);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      unsigned char* const win = smem + wsel * WIN_BYTES;
      unsigned char* const nwin = smem + (wsel + D >= NWIN ? wsel + D - NWIN : wsel + D) * WIN_BYTES;   // of the phase D ahead
      unsigned nbase = 0;
      const bool nvalid = DMA && win_wave && phase_ahead(it, chunk, D, tile, nbase);
      unsigned nbase1 = 0;
      const bool nvalid1 = PBNA && DMA && win_wave && phase_ahead(it, chunk, 1, tile, nbase1);    // is there a next phase
      unsigned char* const win1 = smem + (wsel + 1 >= NWIN ? wsel + 1 - NWIN : wsel + 1) * WIN_BYTES;
#pragma unroll
      for (int j = 0; j < NKC; ++j) {
        // weight waves, three slots: everything but the batch of the previous iteration (the weights of step s + 1) must
        // have landed; two slots: that batch IS the weights of this step.  Window waves: this phase's window, issued at
        // the start of the previous phase, must have landed when the phase starts; nothing to wait for inside a phase.
        const bool gfirst = j % KPB == 0;     // first K-step of its group: the group's wait, barrier and DMA issue (KPB)
        auto sync = [&]() __attribute__((always_inline)) {
        if (!gfirst) return;
        if (!DMA) {
        } else if (win_wave) {
          // this phase's window has landed once only the younger one (D = 2) is still in flight; a tile's first phase
          // also waits for the epilogue's stores, which sit between them in the counter
          // (producer waves issue no stores: their look-ahead survives the tile boundary)
#ifndef PP_WIN_PROD_DRAIN
#define PP_WIN_PROD_DRAIN 0
#endif
          if (j == 0) wait_vmcnt_dyn(((drain && (!PROD || PP_WIN_PROD_DRAIN)) || D == 1) ? 0 : last_win);
        } else if (wgt_wave) {
          wait_vmcnt_dyn((drain || NBS == 2 || RW || STG) ? 0 : last_batch);
        }
        drain = false;
        PP_STAMP(0)
        if (!RW || j == 0) __builtin_amdgcn_s_barrier();
        if (PBNA) {                          // (the producers activated this window during the previous phase)
        } else if (BNA && j == 0 && !COMP) {        // (producer waves: only the pass's closing barrier)
          __builtin_amdgcn_s_barrier();
        } else if (BNA && j == 0) {
          // z = relu(y * scale + shift) on the window that has just landed: 256 rows x six 8-channel octets, three per
          // thread; then every wave sees the activated rows (the masked taps' zero row is the conv's zero padding of z)
          constexpr int NOCT = BM * (CC / 8);
          constexpr int NPT = (NOCT + NT - 1) / NT;
          // (inline asm throughout: plain LDS accesses here make hipcc wait for every LDS-DMA in flight -- the windows of
          // the next two phases -- before them and again in the K-steps.  All reads first, one wait.)
          u32x4 vv[NPT], s0[NPT], s1[NPT], h0[NPT], h1[NPT];
          unsigned adr[NPT];
#pragma unroll
          for (int k = 0; k < NPT; ++k) {
            const int idx = tid + NT * k < NOCT ? tid + NT * k : 0;       // (threads past the end re-read chunk 0, unused)
            const int row = idx / (CC / 8), c8 = idx % (CC / 8);
            adr[k] = (unsigned)(uintptr_t)(lds_ptr)(win + row * XS + c8 * 16);
            const unsigned t = (unsigned)(uintptr_t)(lds_ptr)(bna_tab + chunk * CC + c8 * 8);
            asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %6\n\tds_read_b128 %2, %6 offset:16\n\t"
                         "ds_read_b128 %3, %6 offset:%7\n\tds_read_b128 %4, %6 offset:%8"
                         : "=&v"(vv[k]), "=&v"(s0[k]), "=&v"(s1[k]), "=&v"(h0[k]), "=&v"(h1[k])
                         : "v"(adr[k]), "v"(t), "n"(BNA_CH * 4), "n"(BNA_CH * 4 + 16)
                         : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < NPT; ++k)
            asm volatile("" : "+v"(vv[k]), "+v"(s0[k]), "+v"(s1[k]), "+v"(h0[k]), "+v"(h1[k]));
#pragma unroll
          for (int k = 0; k < NPT; ++k) {
            if (tid + NT * k < NOCT) {
              float x[8];
              unpack8(make_uint4(vv[k][0], vv[k][1], vv[k][2], vv[k][3]), x);
              const float sc[8] = {__uint_as_float(s0[k][0]), __uint_as_float(s0[k][1]), __uint_as_float(s0[k][2]), __uint_as_float(s0[k][3]),
                                   __uint_as_float(s1[k][0]), __uint_as_float(s1[k][1]), __uint_as_float(s1[k][2]), __uint_as_float(s1[k][3])};
              const float sh[8] = {__uint_as_float(h0[k][0]), __uint_as_float(h0[k][1]), __uint_as_float(h0[k][2]), __uint_as_float(h0[k][3]),
                                   __uint_as_float(h1[k][0]), __uint_as_float(h1[k][1]), __uint_as_float(h1[k][2]), __uint_as_float(h1[k][3])};
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const float zv = x[q] * sc[q] + sh[q];
                x[q] = (p.bna_relu && !(zv > 0.f)) ? 0.f : zv;
              }
              const uint4 o = pack8(x);
              const u32x4 ov = {o.x, o.y, o.z, o.w};
              asm volatile("ds_write_b128 %0, %1" ::"v"(adr[k]), "v"(ov) : "memory");
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
        };
        PP_STAMP(1)
        auto issue_dmas = [&](const int pos) __attribute__((always_inline)) {
          if (!gfirst) return;
          if (pos >= 0 && pos != (win_wave ? PP_WIN_WPOS : (STG ? 0 : 1))) return;   // (STG: the weights have ONE step to land: issue first)
          if (!DMA) {
          } else if (win_wave) {
            if (TW) {
              if (j == 0) {
                last_win = 0;
                if (nvalid) {
#pragma unroll
                  for (int k = 0; k < NWP; ++k) dma_window_piece(k, nwin, nbase);
                  last_win = npieces;
                }
                if (PBNA && nvalid1) {   // the next phase's window has landed once only the one just issued is in flight
                  wait_vmcnt_dyn(last_win);
                  bna_own(win1, chunk + 1 < nchunk ? chunk + 1 : 0);
                }
              }
            } else if (nvalid) {
#pragma unroll
              for (int k = 0; k < NWP; ++k)
                if (k >= j * PPK && k < (j + KPB) * PPK) dma_window_piece(k, nwin, nbase);
            }
          } else if (wgt_wave && !RW) {
            // the group LA ahead: (chunk, j / KPB + LA), or the first ones of the next chunk
            const int g2 = j / KPB + LA < NG ? j / KPB + LA : j / KPB + LA - NG;
            const int c2 = j / KPB + LA < NG ? chunk : chunk + 1;
            last_batch = 0;
            if (c2 < nchunk) last_batch = dma_group(bring + slot_ahead(bsl) * B_SLOT, c2, g2);
          }
        };
        PP_STAMP(2)
        compute(comp_c, j, win, RW ? bring + (chunk * NKC + j) * B_BYTES : bring + bsl * B_SLOT + (j % KPB) * B_BYTES, issue_dmas, sync,
                STG && win_wave && j > 0);
        PP_STAMP(3)
        if (j % KPB == KPB - 1 || j == NKC - 1) bsl = next_slot(bsl);
      }
      wsel = wsel + 1 == NWIN ? 0 : wsel + 1;
    }
    const int mb_done = mb, nb_done = nb;
    unsigned char* const ebuf = smem + (wsel == 0 ? NWIN - 1 : wsel - 1) * WIN_BYTES;   // the window just consumed stages the output
    unsigned char* const sbuf = RW ? ebuf + STG_BYTES                                // ... the weight slot just consumed the statistics
                                   : bring + (bsl == 0 ? NBS - 1 : bsl - 1) * B_SLOT;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // every wave is done reading it
    if (next_tile >= 0) {                                       // the next tile's first weight steps fly under the epilogue
      setup_tile(next_tile);
      // (two slots: the slot just consumed holds the statistics during the epilogue, so only the other one is refilled:
      // it is the slot of step 0 of the next tile because bsl already points past the consumed one)
      if (DMA && wgt_wave && !RW) {
        int sl = bsl;
#pragma unroll
        for (int a = 0; a < LA; ++a) {
          dma_group(bring + sl * B_SLOT, 0, a);
          sl = next_slot(sl);
        }
      }
    }
    PP_STAMP(4)
    if constexpr (COMP) {
      if (!(ABL & 8)) {
#pragma unroll
        for (int cls = 0; cls < NCLS; ++cls) epilogue(mb_done, nb_done, ebuf, sbuf, cls);
      }
    } else if (MT == 2 && !S2D && p.colstats) {
      __builtin_amdgcn_s_barrier();     // (the statistics' barrier inside the epilogue)
    }
    PP_STAMP(5)
    abl_first = false;
    if (next_tile < 0) break;
    tile = next_tile;
    ++it;
    drain = true;
  }
#if PP_WIN_ABLATE & 64
  tsum[6] = stamp_now() - tbegin;
  if (lane == 0 && bid < 256)
    for (int i = 0; i < 8; ++i) pp_win_stamp_buf[(bid * 8 + wave) * 8 + i] = tsum[i];
#endif
  };
  if constexpr (PROD) {
    if (is_comp) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  } else {
    main_loop(std::true_type{});
  }
}

// (WN, CC, MT, TW) combinations the data gradients of r2plus1d_18 / r3d_18 / mc3_18 / resnet18 dispatch to get the BNR form
template <int WN, int CC, int MT, bool TW>
constexpr bool bnr_built() { return TW ? (CC == 64 && WN == 4) : (WN == 4 || WN == 8); }   // (MT x NIT <= 8 chunks of y per lane)

template <int WN, int CC, int MT, int NBS, bool TW = false, int HL = HALO, bool S2D = false, int KPB = 1>
int launch_win(const pp_igemm_desc& d, hipStream_t s) {
  constexpr int BN = 16 * WN, BM = 16 * MT * NW;
  const pp_gather& gg = d.g;
  WinGeom g;
  g.W = gg.Gw; g.H = gg.Gh; g.M = d.M; g.cstride = gg.cstride; g.cg = gg.cg;
  g.oH = gg.Rh; g.oW = gg.Rw; g.Mout = d.M;
  if (S2D) g.M = (int)((long long)d.M / ((long long)gg.Rt * gg.Rh * gg.Rw) * gg.Gt * gg.Gh * gg.Gw);   // tiles walk the rows of dy
  g.sign = gg.mode == PP_CONV_FWD ? 1 : -1;
  g.dW_ = make_fastdiv((uint32_t)gg.Gw);
  g.dH_ = make_fastdiv((uint32_t)gg.Gh);
  g.T = gg.Gt; g.HW = gg.Gh * gg.Gw; g.PB = TW ? BM / gg.Gt : 1;
  g.dPB = make_fastdiv((uint32_t)g.PB);
  g.nblk = (g.HW + g.PB - 1) / g.PB;
  g.dBlk = make_fastdiv((uint32_t)(g.nblk > 0 ? g.nblk : 1));
  const int nblk_n = (d.N + BN - 1) / BN;
  g.dNb = make_fastdiv((uint32_t)nblk_n);
  WinArgs a;
  a.A = (const h16raw*)d.A; a.Bt = (const h16raw*)d.Bt; a.C = (h16raw*)d.C; a.residual = (const h16raw*)d.residual;
  a.colstats = d.colstats;
  a.N = d.N; a.b_rows = d.b_rows; a.ldb = d.ldb; a.ldc = d.ldc; a.ldr = d.ldr; a.ldstat = d.ldstat;
  a.bnr_y = (const h16raw*)d.bnr_y; a.bnr_z = (const h16raw*)d.bnr_z;
  a.bnr_mean = d.bnr_mean; a.bnr_rstd = d.bnr_rstd; a.bnr_scale = d.bnr_scale; a.bnr_shift = d.bnr_shift;
  a.bnr_relu = d.bnr_relu; a.bnr_partials = d.bnr_partials;
  const bool bna = d.a_bn_scale != nullptr;
  a.bna_scale = d.a_bn_scale; a.bna_shift = d.a_bn_shift; a.bna_relu = d.a_bn_relu;
  a.a_bytes = (unsigned)((((long long)g.M - 1) * gg.cstride + gg.cg) * 2);
  const long long b_bytes = (((long long)d.b_rows - 1) * d.ldb + d.K) * 2;
  if (b_bytes <= 0 || b_bytes >= 0x40000000LL) { pp_set_error("pp_igemm: weight matrix too large for the window kernel"); return PP_ERR_INVALID; }
  a.b_bytes = (unsigned)b_bytes;
  const long long nblk_m = TW ? (long long)(d.M / ((long long)gg.Gt * g.HW)) * g.nblk : ((long long)g.M + BM - 1) / BM;
  g.nstat = TW ? (int)(nblk_m * (BM / 128)) : (int)(((long long)g.M + 127) / 128);   // (spatial: rows [128 r, 128 r + 128) that exist)
  const bool ragged = TW && (g.HW % g.PB != 0 || gg.Gt * g.PB != BM);    // dead rows inside tiles: no BatchNorm-backward sums in the epilogue
  const long long ntiles = nblk_m * nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) { pp_set_error("pp_igemm: grid too large"); return PP_ERR_INVALID; }
  const long long gx = ntiles < pp_opt_persist_cus ? ntiles : pp_opt_persist_cus;
  dim3 grid((unsigned)gx, 1, 1), block(NT);
  if constexpr (S2D) {
    if (bna) { pp_set_error("pp_igemm: fused BatchNorm apply is built for the temporal window kernel only"); return PP_ERR_INVALID; }
    if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, false, false, false, false, HL, true, KPB>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
    else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, false, false, false, HL, true, KPB>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
    PP_LAUNCH_CHECK();
    return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
  } else if constexpr (!TW && (HL != HALO || KPB != 1)) {
    // frames wider than 63 (HL = 96 rows of halo), and the two-K-steps-per-barrier form (KPB = 2) of the 64-column tiles:
    // the producer form only -- no BatchNorm-backward sums in the epilogue (the caller runs pp_bn_bwd_reduce:
    // PP_BNR_SKIPPED), no staggered / lockstep variants
    if (bna) { pp_set_error("pp_igemm: fused BatchNorm apply is built for the temporal window kernel only"); return PP_ERR_INVALID; }
    dim3 pblock(NT + 256);
    if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, false, false, false, true, HL, false, KPB>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
    else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, false, false, true, HL, false, KPB>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
    PP_LAUNCH_CHECK();
    return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
  } else {
  if constexpr (TW) {
    if (pp_opt_win_producers >= 3 && !(d.bnr_partials && !ragged && bnr_built<WN, CC, MT, TW>())) {     // (3: also the temporal form)
      dim3 pblock(NT + 256);
      if (bna) {
        if constexpr (CC == 48) {
          if (d.residual || gg.cg > 160 || d.bnr_partials) { pp_set_error("pp_igemm: fused BatchNorm apply: no residual, no backward sums, <= 160 channels"); return PP_ERR_INVALID; }
          hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, true, false, true>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
          PP_LAUNCH_CHECK();
          return PP_OK;
        }
      } else {
        if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, false, false, false, true>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
        else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, false, false, true>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
        PP_LAUNCH_CHECK();
        return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
      }
    }
  }
  if constexpr (!TW) {
    // (1: where it pays -- tiles up to 128 columns; 144-column tiles keep 72 accumulators + 36 weight-fragment registers
    // and lose more to the 168-register budget than the producers give back: layer-1 forward 765 -> 845 us; 2: always)
    if ((pp_opt_win_producers == 2 || pp_opt_win_producers == 4 || (pp_opt_win_producers != 0 && WN <= 8)) && !bna && !(d.bnr_partials && bnr_built<WN, CC, MT, TW>())) {
      dim3 pblock(NT + 256);
      if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, false, false, false, true>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, false, false, true>), grid, pblock, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      PP_LAUNCH_CHECK();
      return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
    }
  }
  if constexpr (bnr_built<WN, CC, MT, TW>()) {
    if (d.bnr_partials && !ragged) {
      if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, true>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, true>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      PP_LAUNCH_CHECK();
      return PP_OK;
    }
  }
  if (bna) {
    if constexpr (TW && CC == 48 && WN == 4) {
      if (d.residual || gg.cg > 160 || d.bnr_partials) { pp_set_error("pp_igemm: fused BatchNorm apply: no residual, no backward sums, <= 160 channels"); return PP_ERR_INVALID; }
      hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, true>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      PP_LAUNCH_CHECK();
      return PP_OK;
    } else {
      pp_set_error("pp_igemm: fused BatchNorm apply is built for the temporal window kernel with 48-channel chunks only");
      return PP_ERR_INVALID;
    }
  }
  if constexpr (!TW && NBS == 3) {
    if (pp_opt_win_stagger) {    // staggered halves (see the kernel's K-step)
      if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW, false, false, true>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW, false, false, true>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
      PP_LAUNCH_CHECK();
      return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
    }
  }
  if (d.residual) hipLaunchKernelGGL((igemm_win_kernel<WN, CC, true, MT, NBS, TW>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
  else hipLaunchKernelGGL((igemm_win_kernel<WN, CC, false, MT, NBS, TW>), grid, block, 0, s, a, g, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, pp_opt_win_out_nt);
  PP_LAUNCH_CHECK();
  return d.bnr_partials ? PP_BNR_SKIPPED : PP_OK;
  }
}

}  // namespace

#if PP_WIN_ABLATE & 64
extern "C" int pp_debug_win_stamps(void* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pp_win_stamp_buf), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -1;
}
#endif

// temporal (3,1,1) stride-1 convs with 4, 8, 16 or 32 frames: 256-row tiles = all frames of 64 / 32 / 16 / 8 positions
static bool tw_shape_ok(const pp_igemm_desc& d) {
  const pp_gather& g = d.g;
  const bool conv = g.mode == PP_CONV_FWD || g.mode == PP_CONV_DGRAD;
  // 2..32 frames (round 4; was 4 / 8 / 16 / 32 with H*W a multiple of 256 / T): PB = 256 / T positions per tile, dead rows masked
  const bool pow2 = g.Gt == 4 || g.Gt == 8 || g.Gt == 16 || g.Gt == 32;
  const bool even = pow2 && (g.Gh * g.Gw) % (256 / g.Gt) == 0 && d.M % 256 == 0;
  const long long rows = (long long)g.Gt * g.Gh * g.Gw;
  return conv && d.nbatch == 1 && !d.c_fp32 && !d.bias && d.act == PP_ACT_NONE && d.drop_p == 0.f &&
         !d.Cpre && !d.omap && g.kt == 3 && g.kh == 1 && g.kw == 1 && g.st == 1 && g.sh == 1 && g.sw == 1 &&
         g.pt == 1 && g.ph == 0 && g.pw == 0 && g.Gt == g.Rt && g.Gh == g.Rh && g.Gw == g.Rw && g.Gt >= 2 && g.Gt <= 32 &&
         (even || pp_opt_win_ragged) && d.K == 3 * g.cg && d.M % rows == 0 &&
         (long long)d.M * g.cstride < 0x7f000000LL && (!d.residual || d.ldr % 8 == 0);
}

// Would pp_igemm apply a producer BatchNorm (a_bn_scale / a_bn_shift) for this problem?  (the temporal form from 48-channel
// chunks with at most 64 output columns: the layer-1 temporal convs and the stem's)
int pp_igemm_abn_ok(const pp_igemm_desc& d) {
  const pp_gather& g = d.g;
  const int n16 = (d.N + 15) / 16;
  return pp_opt_win_igemm && (long long)d.M >= pp_opt_win_igemm && pp_opt_win_temporal && g.mode == PP_CONV_FWD && tw_shape_ok(d) &&
         !d.residual && !d.bnr_partials && (g.cg == 48 || g.cg == 144) && n16 <= 4;
}

// Partial rows of column statistics ([rows][2][ldstat] fp32) pp_igemm writes for `d`: one per 128 output rows, except the
// temporal window form, whose tiles are all frames of 256 / T positions -- two rows per TILE, and with a frame count or
// frame size that does not divide evenly there are more tiles than M / 256.  Mirrors pp_igemm_win_try's dispatch.
long long pp_igemm_win_stat_rows(const pp_igemm_desc& d) {
  const pp_gather& g = d.g;
  const long long dflt = ((long long)d.M + 127) / 128;
  if (!(pp_opt_win_igemm && (long long)d.M >= pp_opt_win_igemm && pp_opt_win_temporal && tw_shape_ok(d))) return dflt;
  const int n16 = (d.N + 15) / 16;
  const bool taken = (g.cg == 64 && n16 <= 4) || (g.cg == 64 && n16 <= 9 && !d.colstats) || (g.cg == 48 && n16 <= 4) || (g.cg == 144 && n16 <= 4);
  if (!taken) return dflt;
  const int pb = 256 / g.Gt;
  const long long nblk = ((long long)g.Gh * g.Gw + pb - 1) / pb;
  return (long long)d.M / ((long long)g.Gt * g.Gh * g.Gw) * nblk * 2;
}

// PP_OK if the window kernel took the problem, 1 if the shape is not one it handles (caller falls through), < 0 on error.
// `d` has been validated by pp_igemm.
int pp_igemm_win_try(const pp_igemm_desc& d, hipStream_t s) {
  const pp_gather& g = d.g;
  const bool conv = g.mode == PP_CONV_FWD || g.mode == PP_CONV_DGRAD;
  const bool tw_ok = pp_opt_win_temporal && tw_shape_ok(d);
  if (tw_ok) {
    // resident weights: every K-step of a tile has its own ring slot (3 per channel chunk), one column block
    const int n16 = (d.N + 15) / 16;
    if (g.cg == 64 && n16 <= 4) return launch_win<4, 64, 2, 3, true>(d, s);
    if (g.cg == 64 && n16 <= 9 && !d.colstats) return launch_win<9, 64, 2, 3, true>(d, s);
    if (g.cg == 48 && n16 <= 4) return launch_win<4, 48, 2, 3, true>(d, s);
    if (g.cg == 144 && n16 <= 4) return launch_win<4, 48, 2, 9, true>(d, s);
    return 1;
  }
  // the data gradient of a (1,3,3) convolution with stride (1,2,2) (layers 2.0 / 3.0 / 4.0): one launch, dy read once
  const bool s2d_ok = pp_opt_win_s2d && g.mode == PP_CONV_DGRAD && d.nbatch == 1 && !d.c_fp32 && !d.bias && d.act == PP_ACT_NONE && !d.Cpre &&
                      !d.omap && d.drop_p == 0.f && g.kt == 1 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 2 && g.sw == 2 && g.pt == 0 &&
                      g.ph == 1 && g.pw == 1 && g.Gt == g.Rt && g.Gh == (g.Rh + 1) / 2 && g.Gw == (g.Rw + 1) / 2 && g.Gw + 1 <= HALO &&
                      d.K == 9 * g.cg && g.cg % 8 == 0 && g.cstride >= ((g.cg + 63) & ~63) - 56 &&
                      (long long)d.M * 2 < 0x7f000000LL && (long long)d.M / 2 * g.cstride < 0x7f000000LL && (!d.residual || d.ldr % 8 == 0);
  if (s2d_ok) return pp_opt_win_kpb == 2 ? launch_win<4, 64, 2, 3, false, HALO, true, 2>(d, s) : launch_win<4, 64, 2, 3, false, HALO, true>(d, s);
  constexpr int HALO_WIDE = 96;
  const bool shape_ok = conv && d.nbatch == 1 && !d.c_fp32 && !d.bias && d.act == PP_ACT_NONE && !d.Cpre && !d.omap && d.drop_p == 0.f &&
                        g.kt == 1 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 1 && g.sw == 1 && g.pt == 0 &&
                        g.ph == 1 && g.pw == 1 && g.Gt == g.Rt && g.Gh == g.Rh && g.Gw == g.Rw && g.Gw + 1 <= HALO_WIDE &&
                        d.K == 9 * g.cg && (g.cg % 64 == 0 || g.cg % 48 == 0 || (g.cg % 8 == 0 && g.cg > 128 && pp_opt_win_partial)) &&
                        (long long)d.M * g.cstride < 0x7f000000LL && (!d.residual || d.ldr % 8 == 0);
  if (!shape_ok) return 1;
  const int n16 = (d.N + 15) / 16;
  const bool c64 = g.cg % 64 == 0 || g.cg % 48 != 0;      // 64-channel chunks (the last one partial where cg % 64 != 0)
  if (g.Gw + 1 > HALO) {
    // frames 64..95 wide (the reference's own clips: layer 1 is 50 x 90): 96 rows of halo.  Two (256 + 192)-row windows of
    // 64-channel rows are 112 KB, so the wide tiles keep a TWO-slot weight ring (one K-step of flight for a weight slice);
    // the 48-channel data gradient keeps 256-row tiles (the 512-row tile's windows would be 154 KB)
    const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
    if (c64) {
      if (n16 <= 4) return launch_win<4, 64, 2, 3, false, HALO_WIDE>(d, s);
      return c9 <= c8 ? launch_win<9, 64, 2, 2, false, HALO_WIDE>(d, s) : launch_win<8, 64, 2, 2, false, HALO_WIDE>(d, s);
    }
    if (n16 <= 4) return pp_opt_win_kpb == 2 ? launch_win<4, 48, 2, 3, false, HALO_WIDE, false, 2>(d, s) : launch_win<4, 48, 2, 3, false, HALO_WIDE>(d, s);
    return c9 <= c8 ? launch_win<9, 48, 2, 2, false, HALO_WIDE>(d, s) : launch_win<8, 48, 2, 2, false, HALO_WIDE>(d, s);
  }
  // tile widths: 64 columns (narrow outputs) or 128 / 144 (whichever pads N less)
  if (c64) {
    if (n16 <= 4) return launch_win<4, 64, 2, 3>(d, s);
    const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
    return c9 <= c8 ? launch_win<9, 64, 2, 3>(d, s) : launch_win<8, 64, 2, 3>(d, s);
  }
  if (n16 <= 4) {
    // narrow output from 48-channel chunks (the layer-1 data gradient): 512-row tiles, four row tiles per wave, when
    // there are enough rows to give every CU a few of them and no statistics are asked for
    if (pp_opt_win_tall && !d.colstats && (pp_opt_win_tall == 2 || (long long)d.M >= 512LL * 256 * 2)) return launch_win<4, 48, 4, 2>(d, s);   // (2 = forced: tests)
    // (two K-steps per barrier where the producer form runs and nothing is asked of the epilogue: the layer-1 data gradient)
    if (pp_opt_win_kpb == 2 && pp_opt_win_producers && !d.colstats && !d.bnr_partials) return launch_win<4, 48, 2, 3, false, HALO, false, 2>(d, s);
    return launch_win<4, 48, 2, 3>(d, s);
  }
  const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
  return c9 <= c8 ? launch_win<9, 48, 2, 3>(d, s) : launch_win<8, 48, 2, 3>(d, s);
}
