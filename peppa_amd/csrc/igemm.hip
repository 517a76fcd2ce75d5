// Implicit-GEMM forward / data-gradient / dense "NT" GEMM on MFMA (gfx950).
//
//   C[M,N] = act(A_gather[M,K] x Bt[N,K]^T + bias) (+ residual)        bf16 in, fp32 accumulate
//
// One kernel serves every contraction of the PeppaPig step whose reduce index is contiguous in
// both operands: 3-D convolutions of r2plus1d_18 (spatial 1x3x3, temporal 3x1x1, 1x1x1 stride-2
// shortcut; torchvision modules reached from pig/models.py:141-150), their data gradients, the
// strided Conv1d stack and grouped positional conv of wav2vec2 and all Linear layers
// (pig/models.py:101-105).  The left operand is gathered on the fly (no im2col in HBM).
//
// Tiling: (32*NW) x (16*WN) x 64 per workgroup of NW waves (4 or 8) stacked along M (32 rows each),
// v_mfma_f32_16x16x32_bf16, register-staged global loads one K-step ahead of the MFMAs, LDS
// tiles with 64-byte rows XOR-swizzled for conflict-free ds_read_b128, epilogue staged through
// LDS so every global store is a full 16-byte row segment.
#include "common.h"
#include <string.h>
#include <type_traits>

int pp_opt_xcd_remap_igemm = 1;
int pp_opt_persistent = 1;
int pp_opt_win_tall = 0;       // window kernel: 512-row tiles (four row tiles per wave) for narrow outputs: 1 with enough rows, 2 always.
                               // Default 0 since round 4: with the producer waves the 256-row tile is as fast alone (614 vs 618 us on the
                               // layer-1 data gradient), reads 1.0x instead of 1.8x its operand (two 70-KB windows per CU no longer fit an
                               // XCD's L2 between chunk passes) and the step is 0.1 ms faster with it (tools/ab_step.py win_tall)
int pp_opt_deterministic = 0;   // ordered reductions instead of fp32 atomics wherever a sum crosses workgroups (slower; see the header)
int pp_opt_tw_producers = 1;   // temporal sliding-window weight gradient: three extra waves issue the LDS-DMAs, the nine multiplying waves none
int pp_opt_tw_narrow = 1;      // temporal sliding-window weight gradient: 48-channel blocks with a deep look-ahead for cg <= 48
int pp_opt_ring_producers = 1;  // LDS-DMA ring GEMM: four extra waves issue the DMAs (tiles up to 128 columns)
int pp_opt_ln_bwd_alone = 0;    // LayerNorm backward keeps LDS-using kernels off its CUs (A/B switch; see pp_layernorm_bwd)
int pp_opt_win_producers = 4;  // window kernel: four extra waves issue the LDS-DMAs (0: never; 1: spatial form, tiles up to 128 columns; 2: every spatial tile; 3 = 1 + the temporal form; 4 = 2 + the temporal form, the default)
int pp_opt_win_stagger = 0;    // window kernel, spatial form: waves 4-7 request their fragments ahead of the K-step's barrier
int pp_opt_win_temporal = 1;   // window kernel also for (3,1,1) stride-1 convs (frames-by-positions tiles)
int pp_opt_win_igemm = 1024;   // window kernel for (1,3,3) stride-1 convs (forward / data gradient) once M >= this (0 = never)
int pp_opt_sw_wgrad = 4096;    // sliding-window weight gradient for (1,3,3) stride-1 convs once M >= this (0 = never)
int pp_opt_ring_wgrad = 0;      // LDS-DMA ring weight gradient once the reduce dimension has this many rows (0 = never)
int pp_opt_ring = 128; // LDS-DMA ring variant once there are this many 256-row tiles (0 = never)
int pp_opt_persist_cus = 256;   // workgroups of the persistent (one per CU) ring / window kernels
int pp_opt_ring_wn = 0;  // dense ring tile width in 16-column units (6, 8, 9; 0 = chosen per problem)
int pp_opt_xcd_remap_wgrad = 1;
int pp_opt_wgrad_flat = 1;
int pp_opt_win_s2d = 1;
int pp_opt_win_partial = 1;
int pp_opt_win_ragged = 1;
int pp_opt_win_kpb = 2;
int pp_opt_wgrad_group_ring = 0;
int pp_opt_igemm_big = 0;         // 256 x 256 forward / data-gradient tiles from this M; > 0: plain epilogues only, < 0: fused ones too.  OFF: bit-identical
                                  // and 15 % faster alone on the audio convolutions' data gradients, no change of the step (they have slack)
int pp_opt_wgrad_big = 32768;        // 256 x 256 weight-gradient tiles once the reduce dimension has this many rows (0 = never)
// BatchNorm streaming passes (tools/bench_bn.py, layer-1 shapes): non-temporal STORES + 32 k workgroups instead of plain
// stores + 4 k: apply 417 -> 370 us, backward apply 576 -> 485 us at 144 channels (4.4 / 4.8 -> 5.0 / 5.7 TB/s); non-temporal
// loads on top bought nothing.  bit 0: non-temporal loads, bit 1: non-temporal stores.
// Non-temporal stores of the output tile (window kernels: always; gather / ring kernels: outputs over 64 MB -- the
// transformer's 11-MB activations are re-read from the caches by the next kernel).  tools/bench_gemm.py, layer-1 shapes:
// forward 841 -> 828 us, data gradient 777 -> 766 us, temporal 393 -> 387 / 306 -> 300 us.
int pp_opt_win_out_nt = 1;
int pp_opt_bn_nt = 2;
int pp_opt_bn_grid = 32768;
static inline int out_nt_for(const pp_igemm_desc& d) {
  return pp_opt_win_out_nt && !d.c_fp32 && (long long)d.M * d.ldc * 2 > (64LL << 20);
}

extern "C" int pp_set_option(const char* name, int value) {
  if (!name) return PP_ERR_INVALID;
  if (!strcmp(name, "xcd_remap_igemm")) { pp_opt_xcd_remap_igemm = value; return PP_OK; }
  if (!strcmp(name, "win_tall")) { pp_opt_win_tall = value; return PP_OK; }
  if (!strcmp(name, "deterministic")) { pp_opt_deterministic = value ? 1 : 0; return PP_OK; }
  if (!strcmp(name, "tw_producers")) { pp_opt_tw_producers = value; return PP_OK; }
  if (!strcmp(name, "tw_narrow")) { pp_opt_tw_narrow = value; return PP_OK; }
  if (!strcmp(name, "ring_producers")) { pp_opt_ring_producers = value; return PP_OK; }
  if (!strcmp(name, "ln_bwd_alone")) { pp_opt_ln_bwd_alone = value; return PP_OK; }
  if (!strcmp(name, "win_producers")) { pp_opt_win_producers = value; return PP_OK; }
  if (!strcmp(name, "win_stagger")) { pp_opt_win_stagger = value; return PP_OK; }
  if (!strcmp(name, "win_igemm")) { pp_opt_win_igemm = value; return PP_OK; }
  if (!strcmp(name, "win_temporal")) { pp_opt_win_temporal = value; return PP_OK; }
  if (!strcmp(name, "sw_wgrad")) { pp_opt_sw_wgrad = value; return PP_OK; }
  if (!strcmp(name, "ring_wgrad")) { pp_opt_ring_wgrad = value; return PP_OK; }
  if (!strcmp(name, "ring_igemm")) { pp_opt_ring = value; return PP_OK; }
  if (!strcmp(name, "persist_cus")) { pp_opt_persist_cus = (value >= 8 && value <= 256) ? value : 256; return PP_OK; }
  if (!strcmp(name, "ring_wn")) { pp_opt_ring_wn = (value == 6 || value == 8 || value == 9) ? value : 0; return PP_OK; }
  if (!strcmp(name, "persistent_igemm")) { pp_opt_persistent = value; return PP_OK; }
  if (!strcmp(name, "xcd_remap_wgrad")) { pp_opt_xcd_remap_wgrad = value; return PP_OK; }
  if (!strcmp(name, "wgrad_flat")) { pp_opt_wgrad_flat = value; return PP_OK; }
  if (!strcmp(name, "win_s2d")) { pp_opt_win_s2d = value; return PP_OK; }
  if (!strcmp(name, "win_partial")) { pp_opt_win_partial = value; return PP_OK; }
  if (!strcmp(name, "win_ragged")) { pp_opt_win_ragged = value; return PP_OK; }
  if (!strcmp(name, "win_kpb")) { pp_opt_win_kpb = value == 2 ? 2 : 1; return PP_OK; }
  if (!strcmp(name, "wgrad_group_ring")) { pp_opt_wgrad_group_ring = value; return PP_OK; }
  if (!strcmp(name, "wgrad_big")) { pp_opt_wgrad_big = value; return PP_OK; }
  if (!strcmp(name, "igemm_big")) { pp_opt_igemm_big = value; return PP_OK; }
  if (!strcmp(name, "bn_nt")) { pp_opt_bn_nt = value; return PP_OK; }
  if (!strcmp(name, "win_out_nt")) { pp_opt_win_out_nt = value; return PP_OK; }
  if (!strcmp(name, "bn_grid")) { pp_opt_bn_grid = value > 0 ? value : 32768; return PP_OK; }
  pp_set_error("pp_set_option: unknown option %s", name);
  return PP_ERR_INVALID;
}

int pp_igemm_win_try(const pp_igemm_desc& d, hipStream_t s);
int pp_igemm_abn_ok(const pp_igemm_desc& d);
long long pp_igemm_win_stat_rows(const pp_igemm_desc& d);

namespace {

constexpr int BK = 64;
constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond every buffer: the buffer load returns zeros

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 128-byte LDS rows (64 bf16): 16-byte chunk c of row r lives at chunk c ^ swz(r); with this XOR the
// four 16-lane groups of a ds_read_b128 fragment read (rows 0-3/12-15 at chunk q, rows 4-11 at q+1) hit
// 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

struct RowDiv {  // exact division by the row-decomposition extents (host-prepared magic numbers)
  FastDiv dRw, dRh, dRt;
};

struct RowInfo {
  unsigned base;     // byte offset of the row origin in the source tensor (conv: may be "virtual")
  int bt, bh, bw;    // conv: origin coordinates; invalid rows carry bt = -2^20 so every range check fails
  int nbase;         // strided data-gradient only: n * Gt*Gh*Gw
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform dst + 16 * lane; a lane whose offset is out of range
// writes zeros.  (Kept out of the kernel template: inside it the host pass silently drops the kernel's stub.)
__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// RING = false: register-staged loads, two LDS buffers, 2-3 workgroups per CU hide each other's latencies.
// RING = true : 256-row tiles (8 waves), one workgroup per CU; operands go global -> LDS by LDS-DMA
//               (buffer_load ... lds) into a three-slot ring with two K-steps in flight across raw barriers and
//               counted vmcnt waits.  The swizzled LDS image is the same: the XOR moves to the source address.
// PROD (ring only): four producer waves (threads 512 ..767) issue every LDS-DMA of a K-step, the eight multiplying waves
// none (wgrad_tw.hip / igemm_win.hip measured why).  Twelve waves = 168 registers each; the tile loop is instantiated per
// role so that the producers' row tables and the multipliers' accumulators are never live together.
template <int WN, int MODE, bool FULL, int NW, bool RING, bool PROD = false>
__global__ __launch_bounds__(64 * NW + (PROD ? 256 : 0), (RING ? 1 : (WN <= 9 ? 2 : 1))) void igemm_kernel(const pp_igemm_desc p, const int nblk_n,
                                                                       const RowDiv rd, const int xcd_remap, const int ntiles,
                                                                       const int out_nt) {
  static_assert(!RING || (NW == 8 && WN <= 9), "ring variant: 8 waves, BN <= 144");
  static_assert(!PROD || RING, "producer waves: ring variant");
  constexpr int BM = 32 * NW;      // rows per workgroup: one 32-row slab per wave
  constexpr int NT = 64 * NW;      // multiplying threads
  constexpr int NTI = PROD ? 256 : NT;   // threads that load (stage) the operands
  constexpr int RS = NTI / 8;      // row stride between a thread's chunks (8 chunk columns per 128-byte row)
  constexpr int BN = 16 * WN;
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BN * 128;
  constexpr int STG_STRIDE = BN * 2 + 16;
  constexpr int STG_BYTES = NW * 16 * STG_STRIDE;
  constexpr int STAT_BYTES = NW * BN * 2 * 4;
  constexpr int LOOP_BYTES = A_BYTES + B_BYTES;
  constexpr int EPI_BYTES = STG_BYTES + STAT_BYTES;
  constexpr int NSLOT = RING ? 3 : 2;
  constexpr int SMEM = NSLOT * LOOP_BYTES > EPI_BYTES ? NSLOT * LOOP_BYTES : EPI_BYTES;
  static_assert(!RING || EPI_BYTES <= LOOP_BYTES, "the epilogue staging must fit one ring slot");
  constexpr int NAI = BM / RS;                 // A chunks per loading thread and K-step (4; producers: 8)
  constexpr int NBI = (BN * 8 + NTI - 1) / NTI;  // B chunks per loading thread and K-step
  // one LDS object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM + (MODE != PP_DENSE ? 2048 : 0)];
  int* const lut = (int*)(smem + SMEM);            // packed (dt, dh, dw) per tap
  int* const lut_off = (int*)(smem + SMEM + 1024);  // byte offset of the tap inside the source tensor (linear part)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const bool is_comp = !PROD || tid < NT;      // (wave-uniform)
  const int itid = PROD ? (tid - NT) & (NTI - 1) : tid;   // index among the loading threads (multiplying waves: unused)
  const int iwave = itid >> 6;
  // Persistent workgroups: block b walks tiles b, b + G, b + 2G, ... (G = gridDim.x), prefetching the first K-step of
  // its next tile under the epilogue of the current one, so the cold-start load latency and the prologue are paid
  // once per workgroup instead of once per tile.
  // XCD-aware order inside each wave of G tiles: workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels
  // the XCD), so each XCD gets a CONTIGUOUS tile range -- neighbouring M-tiles share their gather halo (rows
  // m +/- W, +/- HW) and hit the same private L2 instead of every XCD re-fetching it (bijective for any G).
  const int G = gridDim.x, bid = blockIdx.x;
  auto tile_index = [&](int it) __attribute__((always_inline)) -> int {
    const int base = it * G;
    const int cnt = ntiles - base < G ? ntiles - base : G;
    if (bid >= cnt) return -1;
    const int xq = cnt >> 3, xr = cnt & 7, xcd = bid & 7;
    const int t = (xcd_remap && cnt >= 8) ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3) : bid;
    return base + t;
  };
  int mb = 0, nb = 0;
  const int z = blockIdx.z;
  const int zo = z / p.inner, zi = z % p.inner;

  const h16raw* A = (const h16raw*)p.A + zo * p.a_s0 + zi * p.a_s1;
  const h16raw* Bt = (const h16raw*)p.Bt + zo * p.b_s0 + zi * p.b_s1;
  const long long c_off = zo * p.c_s0 + zi * p.c_s1;
  const float* __restrict__ bias = p.bias ? p.bias + zo * p.bias_s0 + zi * p.bias_s1 : nullptr;
  // raw buffer descriptors (wave-uniform): an offset of OOB is out of range and loads zeros, which
  // replaces every validity branch of the gather by one v_cndmask on the offset
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, (int)OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bt, (short)0, (int)OOB, 0x00020000);

  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    for (int tp = tid; tp < 256; tp += NT) {   // up to 256 taps (r3d_18's (3,7,7) stem has 147)
      int e = 0, o = 0;
      if (tp < ntaps) {
        const int dw = tp % g.kw;
        const int t2 = tp / g.kw;
        const int dh = t2 % g.kh;
        const int dt = t2 / g.kh;
        e = dt | (dh << 8) | (dw << 16);
        o = ((dt * g.Gh + dh) * g.Gw + dw) * g.cstride * 2;
      }
      lut[tp] = e;
      lut_off[tp] = o;
    }
    __syncthreads();
  }
  // stride-1 gathers are linear in the tap: offset = row_base +/- lut_off[tap]
  const bool unit_stride = (g.st == 1 && g.sh == 1 && g.sw == 1);
  const int sft = g.st == 2, sfh = g.sh == 2, sfw = g.sw == 2;

  // ---- per-thread bookkeeping: 4 A rows (tid>>3 + RS i) and NBI B rows, one 16-byte chunk column kq ----
  // register staging: thread loads chunk kq and stores it at slot kq ^ swz(row); LDS-DMA lands lane-linear, so the
  // thread owning slot (tid & 7) fetches chunk (tid & 7) ^ swz(row) instead (swz(row) is the same for all its rows)
  const int kq = RING ? ((itid & 7) ^ swz(itid >> 3)) : (itid & 7);
  RowInfo ri[NAI];
  unsigned bbase[NBI];
  int kcur = 0;                 // this thread's k within the current K-step
  int tap = 0, cch = 0;         // conv modes: k = tap*cg + cch
  auto setup_tile = [&](int tile) __attribute__((always_inline)) {
    nb = tile % nblk_n;
    mb = tile / nblk_n;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int m = mb * BM + (itid >> 3) + RS * i;
      const bool valid = m < p.M;
      const int mm = valid ? m : 0;
      ri[i].nbase = 0;
      if (MODE == PP_DENSE) {
        ri[i].base = valid ? (unsigned)(mm * g.lda) * 2u : OOB;
        ri[i].bt = ri[i].bh = ri[i].bw = 0;
      } else {
        const uint32_t t1 = fdiv((uint32_t)mm, rd.dRw);
        const int rw = mm - (int)t1 * g.Rw;
        const uint32_t t2 = fdiv(t1, rd.dRh);
        const int rh = (int)t1 - (int)t2 * g.Rh;
        const int n = (int)fdiv(t2, rd.dRt);
        const int rt = (int)t2 - n * g.Rt;
        ri[i].nbase = n * g.Gt * g.Gh * g.Gw;
        if (MODE == PP_CONV_FWD) {
          ri[i].bt = rt * g.st - g.pt;
          ri[i].bh = rh * g.sh - g.ph;
          ri[i].bw = rw * g.sw - g.pw;
        } else {
          ri[i].bt = rt + g.pt;
          ri[i].bh = rh + g.ph;
          ri[i].bw = rw + g.pw;
        }
        ri[i].base = (unsigned)((ri[i].nbase + (ri[i].bt * g.Gh + ri[i].bh) * g.Gw + ri[i].bw) * g.cstride) * 2u;
        if (!valid) ri[i].bt = -(1 << 20);
      }
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int brow = (itid >> 3) + RS * i;
      const int n = nb * BN + brow;
      bbase[i] = (brow < BN && n < p.b_rows) ? (unsigned)(n * p.ldb) * 2u : OOB;
    }
    kcur = kq * 8;
    if (MODE != PP_DENSE) {
      tap = kcur / g.cg;
      cch = kcur % g.cg;
    }
  };

  u32x4 ra[RING ? 1 : NAI], rb[RING ? 1 : NBI];
  // byte offsets of this thread's chunks for the next K-step (OOB where the gather leaves the tensor), then advance
  auto stage_offsets = [&](unsigned (&offA)[NAI], unsigned (&offB)[NBI]) __attribute__((always_inline)) {
    const bool k_ok = kcur < p.K;
    if (MODE == PP_DENSE) {
#pragma unroll
      for (int i = 0; i < NAI; ++i) offA[i] = (k_ok && ri[i].base != OOB) ? ri[i].base + (unsigned)kcur * 2u : OOB;
    } else {
      const bool tap_ok = tap < ntaps;
      const int e = lut[tap & 255];
      const int toff = lut_off[tap & 255] + cch * 2;
      const int dt = e & 0xff, dh = (e >> 8) & 0xff, dw = (e >> 16) & 0xff;
#pragma unroll
      for (int i = 0; i < NAI; ++i) {
        const RowInfo& r = ri[i];
        unsigned off;
        bool ok;
        if (MODE == PP_CONV_FWD) {
          const int gt = r.bt + dt, gh = r.bh + dh, gw = r.bw + dw;
          ok = tap_ok && (unsigned)gt < (unsigned)g.Gt && (unsigned)gh < (unsigned)g.Gh && (unsigned)gw < (unsigned)g.Gw;
          off = r.base + (unsigned)toff;
        } else {
          const int nt = r.bt - dt, nh = r.bh - dh, nw = r.bw - dw;
          if (unit_stride) {
            ok = tap_ok && (unsigned)nt < (unsigned)g.Gt && (unsigned)nh < (unsigned)g.Gh && (unsigned)nw < (unsigned)g.Gw;
            off = r.base - (unsigned)lut_off[tap & 255] + (unsigned)cch * 2u;
          } else {
            const int gt = nt >> sft, gh = nh >> sfh, gw = nw >> sfw;
            ok = tap_ok && (((nt & sft) | (nh & sfh) | (nw & sfw)) == 0) && nt >= 0 && nh >= 0 && nw >= 0 &&
                 gt < g.Gt && gh < g.Gh && gw < g.Gw;
            off = (unsigned)((r.nbase + (gt * g.Gh + gh) * g.Gw + gw) * g.cstride + cch) * 2u;
          }
        }
        offA[i] = ok ? off : OOB;
      }
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) offB[i] = (k_ok && bbase[i] != OOB) ? bbase[i] + (unsigned)kcur * 2u : OOB;
    kcur += BK;
    if (MODE != PP_DENSE) {
      cch += BK;
      while (cch >= g.cg) { cch -= g.cg; ++tap; }
    }
  };
  auto load_stage = [&]() __attribute__((always_inline)) {
    if (!RING) {
      unsigned offA[NAI], offB[NBI];
      stage_offsets(offA, offB);
#pragma unroll
      for (int i = 0; i < NAI; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[i], 0, 0);
#pragma unroll
      for (int i = 0; i < NBI; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB[i], 0, 0);
    }
  };
  // LDS-DMA: one wave-instruction lands 64 x 16 B = eight 128-byte rows, lane-linear from a wave-uniform base; lanes
  // whose offset is OOB write zeros.  Every wave issues NAI pieces of A and NBI or NBI - 1 pieces of B per K-step.
  const bool b_last = 8 * iwave + RS * (NBI - 1) < BN;   // wave-uniform: does this wave own a piece in the last B pass
  auto dma_stage = [&](unsigned char* buf) __attribute__((always_inline)) {
    if (RING) {
      unsigned offA[NAI], offB[NBI];
      stage_offsets(offA, offB);
      unsigned char* dst = buf + (8 * iwave) * 128;
#pragma unroll
      for (int i = 0; i < NAI; ++i)
        lds_dma16(rsA, dst + RS * i * 128, offA[i]);
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        if (i < NBI - 1 || b_last)
          lds_dma16(rsB, dst + A_BYTES + RS * i * 128, offB[i]);
    }
  };
  // wait until at most the newest stage of this wave's DMAs is still in flight (or none)
  auto wait_stage = [&](const bool keep_one) __attribute__((always_inline)) {
    if (!keep_one) wait_vmcnt<0>();
    else if (b_last) wait_vmcnt<NAI + NBI>();
    else wait_vmcnt<NAI + NBI - 1>();
  };
  auto store_stage = [&](unsigned char* buf) __attribute__((always_inline)) {
    if (RING) return;
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
      const int row = (tid >> 3) + RS * i;
      *(u32x4*)(buf + row * 128 + ((kq ^ swz(row)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int row = (tid >> 3) + RS * i;
      if (row < BN) *(u32x4*)(buf + A_BYTES + row * 128 + ((kq ^ swz(row)) << 4)) = rb[i];
    }
  };

  f32x4 acc[2][WN];

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fsw = ((ks * 4 + fq) ^ swz(fr)) << 4;
      h16x8 af[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[mt] = *(const h16x8*)(buf + (wave * 32 + mt * 16 + fr) * 128 + fsw);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const h16x8 bfm = *(const h16x8*)(buf + A_BYTES + (j * 16 + fr) * 128 + fsw);
        acc[0][j] = PP_MFMA16(af[0], bfm, acc[0][j], 0, 0, 0);
        acc[1][j] = PP_MFMA16(af[1], bfm, acc[1][j], 0, 0, 0);
      }
    }
  };

  // ---- epilogue of one finished tile (mb_e, nb_e); defined below the tile loop's helpers -------------------
  const int ncols_store = (p.N + 7) & ~7;
  const uint32_t drop_thr = FULL ? (uint32_t)(p.drop_p * 65536.f + 0.5f) : 0u;
  const float drop_scale = 1.f / (1.f - p.drop_p);
  auto epilogue = [&](const int mb_e, const int nb_e, unsigned char* const ebuf) __attribute__((always_inline)) {
  // FULL = bias / activation / residual / pre-activation copy; otherwise plain store (+ optional
  // BatchNorm column statistics).  Flags are tested once, outside the per-value loops.
  const int m_wave = mb_e * BM + wave * 32;
  if (FULL && bias) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = nb_e * BN + j * 16 + fr;
      const float bv = n < p.N ? bias[n] : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][j][r] += bv;
    }
  }
  auto activate = [&]() __attribute__((always_inline)) {
    if (p.act == PP_ACT_GELU) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][j][r] = gelu_f(acc[mt][j][r]);
    } else if (p.act == PP_ACT_RELU) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][j][r] = fmaxf(acc[mt][j][r], 0.f);
    }
  };
  if (p.c_fp32) {
    if (FULL) activate();
    float* C = (float*)p.C + c_off;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = nb_e * BN + j * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m_wave + mt * 16 + fq * 4 + r;
          if (m < p.M && n < p.N) C[(long long)m * p.ldc + n] = acc[mt][j][r];
        }
      }
    return;
  }

  // Each wave stages its own 16 x BN slab (bf16) in LDS and writes it out as full 16-byte row segments.
  // The slab is wave-private, so LDS program order (+ lgkmcnt waits) is the only synchronisation needed.
  unsigned char* stg = ebuf + wave * 16 * STG_STRIDE;
  unsigned char* stg_w = stg + (fq * 4) * STG_STRIDE + fr * 2;
  auto write_out = [&](h16raw* Cout, const h16raw* residual) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) *(h16raw*)(stg_w + r * STG_STRIDE + j * 32) = f2h(acc[mt][j][r]);
      if (RING) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a fence would also drain the LDS-DMAs in flight)
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // ring, plain epilogue: every staged chunk of the pass is requested, then ONE wait (a wait per chunk exposed an LDS
      // round trip each)
      constexpr int NIT = (32 * WN + 63) / 64;
      constexpr bool BATCH = RING && !FULL;
      u32x4 vvs[BATCH ? NIT : 1];
      if (BATCH) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int cid = lane + 64 * it;
          const unsigned a = (unsigned)(uintptr_t)(lds_ptr)(cid < 32 * WN ? stg + (cid / (2 * WN)) * STG_STRIDE + (cid % (2 * WN)) * 16 : stg);
          asm volatile("ds_read_b128 %0, %1" : "=v"(vvs[it]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" : "+v"(vvs[it]));
      }
      // producer form: the multiplying waves hold no loader state, so the fused epilogue can be unrolled, and the residual
      // chunks of the row tile are all requested before the first is added (one by one, each exposed a memory round trip
      // inside an epilogue that nothing overlaps: the transformer's data gradients with residual are a dozen K-steps long)
      constexpr bool RPRE = FULL && PROD && MODE == PP_DENSE && WN <= 8;   // (the 144-column tile spills unrolled)
      uint4 rres[RPRE ? NIT : 1];
      if (RPRE && residual) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int cid = lane + 64 * it;
          const int m = m_wave + mt * 16 + cid / (2 * WN);
          const int col = nb_e * BN + (cid % (2 * WN)) * 8;
          rres[it] = (cid < 32 * WN && m < p.M && col < ncols_store) ? *(const uint4*)(residual + c_off + (long long)m * p.ldr + col)
                                                                        : make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll((FULL && !RPRE) ? 1 : NIT)   // (rolled for the lockstep fused epilogues: their unrolled form spills)
      for (int it = 0; it < NIT; ++it) {
        const int cid = lane + 64 * it;
        const int row = cid / (2 * WN);
        const int ch = cid % (2 * WN);
        const int m = m_wave + mt * 16 + row;
        const int col = nb_e * BN + ch * 8;
        if (cid < 32 * WN && m < p.M && col < ncols_store) {
          uint4 v;
          if (BATCH) {
            const u32x4 vv = vvs[BATCH ? it : 0];
            v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
          } else if (RING) {
            // (hipcc drains vmcnt -- the next tile's LDS-DMAs and every earlier store -- before a plain LDS load here)
            const unsigned a = (unsigned)(uintptr_t)(lds_ptr)(stg + row * STG_STRIDE + ch * 16);
            u32x4 vv;   // (a native vector type: the host pass must accept the constraint as well)
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(vv) : "v"(a) : "memory");
            v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
          } else {
            v = *(const uint4*)(stg + row * STG_STRIDE + ch * 16);
          }
          long long orow = m;
          if (MODE != PP_DENSE && p.omap) {   // scatter compact rows of a parity class into the full tensor
            const uint32_t t1 = fdiv((uint32_t)m, rd.dRw);
            const int rw = m - (int)t1 * g.Rw;
            const uint32_t t2 = fdiv(t1, rd.dRh);
            const int rh = (int)t1 - (int)t2 * g.Rh;
            const int n = (int)fdiv(t2, rd.dRt);
            const int rt = (int)t2 - n * g.Rt;
            orow = (((long long)n * p.Ot + rt * p.os_t + p.oo_t) * p.Oh + rh * p.os_h + p.oo_h) * p.Ow + rw * p.os_w + p.oo_w;
          }
          if (FULL && drop_thr && Cout == (h16raw*)p.C) {   // (not the pre-activation copy)
            float x[8];
            bool keep[8];
            unpack8(v, x);
            keep8(p.drop_seed, (c_off + orow * p.ldc + col) >> 3, drop_thr, keep);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = keep[q] ? x[q] * drop_scale : 0.f;
            v = pack8(x);
          }
          if (FULL && residual) {
            const uint4 rv = RPRE ? rres[RPRE ? it : 0] : *(const uint4*)(residual + c_off + orow * p.ldr + col);
            float x[8], y[8];
            unpack8(v, x);
            unpack8(rv, y);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] += y[q];
            v = pack8(x);
          }
          if (out_nt) {      // large outputs are next read long after they left the caches: keep them out of the L2
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, (u32x4*)(Cout + c_off + orow * p.ldc + col));
          } else {
            *(uint4*)(Cout + c_off + orow * p.ldc + col) = v;
          }
        }
      }
      if (RING) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (FULL) {
    if (p.Cpre) write_out((h16raw*)p.Cpre, nullptr);
    activate();
    write_out((h16raw*)p.C, (const h16raw*)p.residual);
    return;
  }
  write_out((h16raw*)p.C, nullptr);
  if (p.colstats) {
    // per-column sum / sum of squares over this block's 128 rows (fp32 accumulators), deterministic:
    // in-lane over 8 rows, xor-shuffles over the 4 row groups, LDS over the 4 waves
    float* statbuf = (float*)(ebuf + STG_BYTES);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
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
    if (RING) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    // partial rows are always per 128 output rows (independent of the workgroup height)
    for (int idx = tid; idx < BN * (NW / 4); idx += NT) {
      const int c = idx % BN, h = idx / BN;
      const int n = nb_e * BN + c;
      const long long prow = (long long)mb_e * (NW / 4) + h;
      if (n < p.ldstat && prow * 128 < p.M) {
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

  // ---- tile loop -------------------------------------------------------------------------------------------
  const int nk = (p.K + BK - 1) / BK;
  int it = 0;
  const int first = tile_index(0);
  if (first < 0) return;
  setup_tile(first);
  if (RING) {
    auto ring_loop = [&](auto comp_c) __attribute__((always_inline)) {
    constexpr bool COMP = decltype(comp_c)::value;
    constexpr bool DMA = !PROD || !COMP;
    // K-step s (counted over all tiles of this workgroup) lives in slot s % 3.  Per step: wait for this wave's
    // DMAs of the step (leaving the next one in flight), barrier (every wave's pieces have landed, and every wave
    // is done reading the slot about to be refilled), issue the step after next, compute.
    int s0 = 0;   // slot of this tile's first K-step
    auto slot = [&](int s) __attribute__((always_inline)) { return smem + (s >= 3 ? s - 3 : s) * LOOP_BYTES; };
    if constexpr (DMA) {
      dma_stage(slot(s0));
      if (nk > 1) dma_stage(slot(s0 + 1));
    }
    bool first_tile = true;
    while (true) {
      if constexpr (COMP) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int j = 0; j < WN; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      int sl = s0;
      for (int kt = 0; kt < nk; ++kt) {
        // (the first step of a later tile also has the previous epilogue's stores on the counter: drain them all)
        if constexpr (DMA) wait_stage(kt + 1 < nk && (kt > 0 || first_tile));
        __builtin_amdgcn_s_barrier();
        if constexpr (DMA) {
          if (kt + 2 < nk) dma_stage(slot(sl + 2));
        }
        if constexpr (COMP) compute(slot(sl));
        sl = sl == 2 ? 0 : sl + 1;
      }
      first_tile = false;
      const int mb_done = mb, nb_done = nb;
      const int next = tile_index(++it);
      unsigned char* const last = slot(sl == 0 ? 2 : sl - 1);   // the slot just consumed stages the output
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave is done reading it
      if (next >= 0) {                // the next tile's first two K-steps fly under the epilogue
        setup_tile(next);
        if constexpr (DMA) {
          dma_stage(slot(sl));
          if (nk > 1) dma_stage(slot(sl + 1));
        }
      }
      if constexpr (COMP) {
        epilogue(mb_done, nb_done, last);
      } else if (!p.c_fp32 && p.colstats) {
        __builtin_amdgcn_s_barrier();   // (the statistics' barrier inside the epilogue)
      }
      if (next < 0) break;
      s0 = sl;
    }
    };
    if constexpr (PROD) {
      if (is_comp) ring_loop(std::true_type{});
      else ring_loop(std::false_type{});
    } else {
      ring_loop(std::true_type{});
    }
    return;
  }
  // register staging: one stage of global loads flies under the MFMAs of the current K-step; two LDS buffers, one
  // barrier per 64-deep K-step; the next tile's first stage is issued before a plain-store epilogue
  load_stage();
  while (true) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    store_stage(smem);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {   // single loop body + runtime buffer toggle: half the registers of a 2x unroll
      const bool more = kt + 1 < nk;
      if (more) load_stage();
      compute(smem + cur * LOOP_BYTES);
      cur ^= 1;
      if (more) store_stage(smem + cur * LOOP_BYTES);
      __syncthreads();
    }
    const int mb_done = mb, nb_done = nb;
    const int next = tile_index(++it);
    // plain-store epilogues are light on registers: the cold loads of the next tile fly under them.  The fused
    // (bias / residual / statistics) epilogues need the registers themselves, so there the loads are issued after.
    if (!FULL && next >= 0) {
      setup_tile(next);
      load_stage();
    }
    epilogue(mb_done, nb_done, smem);
    if (FULL || next < 0) break;   // fused epilogues run one tile per workgroup (the launcher sizes the grid so)
    __syncthreads();   // the staging slabs of the epilogue alias the loop buffers
  }
}

// ---- 256 x 256 tiles (round 4) -------------------------------------------------------------------------------------------
// Every GEMM-shaped kernel of this library runs at 8-10 TB/s of L2 -> CU traffic, so a large-M problem
// goes as fast as its tile makes bytes per FLOP small: the 256 x 128 ring tile reads A once per 128 columns, this one once per
// 256.  Eight waves as a 2 x 4 grid of 128 x 64 wave tiles (32 MFMAs per 12 fragment reads), one workgroup per CU, operands
// register-staged with two register sets in flight and two LDS buffers (the form that beat the LDS-DMA ring for the weight
// gradient, wgrad.hip), one tile per workgroup.  Gathers: dense, conv-forward (any stride) and unit-stride conv data gradient
// with at most 32 taps -- the row's origin and the validity of each tap sit in a per-tile LDS table, a K-step adds the tap's
// linear offset.  Epilogue: bias / activation / pre-activation copy / dropout / residual / output row map, bf16 output.
constexpr int GBM = 256, GBN = 256;

struct BigArgs {
  FastDiv dcg;          // k -> (tap, channel): tap = k / cg
};

template <int MODE, bool FULL>
__global__ __launch_bounds__(512, 1) void igemm_big_kernel(const pp_igemm_desc p, const RowDiv rd, const BigArgs ba, const int nblk_n,
                                                           const int ntiles, const int xcd_remap, const int out_nt) {
  constexpr int A_BYTES = GBM * 128, B_BYTES = GBN * 128, BUF = A_BYTES + B_BYTES;
  constexpr int STG_STRIDE = 64 * 2 + 16;
  constexpr int NCH = 4;        // 16-byte chunks per thread, operand and K-step (256 rows x 8 chunks / 512)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + 2048 + GBM * 8];
  static_assert(2 * BUF + 2048 + GBM * 8 <= 160 * 1024 && 8 * 16 * STG_STRIDE <= BUF, "LDS budget");
  int* const lut_off = (int*)(smem + 2 * BUF);                 // byte offset of each tap inside the source tensor
  int2* const rowtab = (int2*)(smem + 2 * BUF + 2048);         // per tile row: {origin byte offset, tap validity mask}

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile;
  {
    const int b0 = blockIdx.x, xq = ntiles >> 3, xr = ntiles & 7, xcd = b0 & 7;
    tile = (xcd_remap && ntiles >= 8) ? (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b0 >> 3) : b0;
  }
  const int nb = tile % nblk_n, mb = tile / nblk_n;       // (the column tiles of a row block are neighbours on one XCD)
  const h16raw* A = (const h16raw*)p.A;
  const h16raw* Bt = (const h16raw*)p.Bt;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, (short)0, (int)OOB, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bt, (short)0, (int)OOB, 0x00020000);
  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    if (tid < 32) {
      int o = 0;
      if (tid < ntaps) {
        const int dw = tid % g.kw;
        const int t2 = tid / g.kw;
        o = (((t2 / g.kh) * g.Gh + (t2 % g.kh)) * g.Gw + dw) * g.cstride * 2;
      }
      lut_off[tid] = o;
    }
    if (tid < GBM) {       // this tile's row table
      const int m = mb * GBM + tid;
      int base = 0;
      unsigned mask = 0;
      if (m < p.M) {
        const uint32_t t1 = fdiv((uint32_t)m, rd.dRw);
        const int rw = m - (int)t1 * g.Rw;
        const uint32_t t2 = fdiv(t1, rd.dRh);
        const int rh = (int)t1 - (int)t2 * g.Rh;
        const int n = (int)fdiv(t2, rd.dRt);
        const int rt = (int)t2 - n * g.Rt;
        // origin of the row's taps; forward: tap d sits at origin + d, data gradient (unit stride): at origin - d
        const int ct = MODE == PP_CONV_FWD ? rt * g.st - g.pt : rt + g.pt;
        const int chh = MODE == PP_CONV_FWD ? rh * g.sh - g.ph : rh + g.ph;
        const int cw = MODE == PP_CONV_FWD ? rw * g.sw - g.pw : rw + g.pw;
        const int sgn = MODE == PP_CONV_FWD ? 1 : -1;
        base = ((((n * g.Gt + ct) * g.Gh + chh) * g.Gw + cw) * g.cstride) * 2;
        unsigned vw = 0, mhw = 0;
        for (int d = 0; d < g.kw; ++d) vw |= (unsigned)((unsigned)(cw + sgn * d) < (unsigned)g.Gw) << d;
        for (int d = 0; d < g.kh; ++d) mhw |= ((unsigned)(chh + sgn * d) < (unsigned)g.Gh) ? vw << (d * g.kw) : 0u;
        for (int d = 0; d < g.kt; ++d) mask |= ((unsigned)(ct + sgn * d) < (unsigned)g.Gt) ? mhw << (d * g.kh * g.kw) : 0u;
      }
      rowtab[tid] = make_int2(base, (int)mask);
    }
    __syncthreads();
  }

  // chunks owned by this thread: chunk column kq of rows lrow + 64 i, of A (gathered rows) and of Bt (output columns).
  // Nothing per row is kept in registers across the K loop (128 accumulators + two register sets of 8 chunks leave no room):
  // a K-step re-reads the rows' table entries from LDS and rebuilds the Bt offsets from two scalars.
  const int kq = tid & 7, lrow = tid >> 3;
  if (MODE == PP_DENSE) {       // the row table of a dense operand: {byte offset of the row or OOB, -}
    if (tid < GBM) {
      const int m = mb * GBM + tid;
      rowtab[tid] = make_int2(m < p.M ? (int)((unsigned)(m * g.lda) * 2u) : (int)OOB, 0);
    }
    __syncthreads();
  }
  const int nb0 = nb * GBN + lrow;
  const int wi = wave >> 2, wj = wave & 3;       // 2 x 4 wave grid: rows [128 wi, +128) x columns [64 wj, +64) of the tile
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2][NCH], rb[2][NCH];
  auto load_stage = [&](const int set, const int step) __attribute__((always_inline)) {
    const int k = step * BK + kq * 8;
    const bool k_ok = k < p.K;
    unsigned koff = (unsigned)k * 2u;
    unsigned tapbit = 0;
    if (MODE != PP_DENSE) {
      const int tap = (int)fdiv((uint32_t)k, ba.dcg);
      const int cch = k - tap * g.cg;
      const unsigned lo = (unsigned)lut_off[tap & 31];
      koff = MODE == PP_CONV_FWD ? lo + (unsigned)cch * 2u : (unsigned)cch * 2u - lo;
      tapbit = k_ok ? 1u << (tap & 31) : 0u;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int2 e = rowtab[lrow + 64 * i];
      unsigned off;
      if (MODE == PP_DENSE) off = (k_ok && (unsigned)e.x != OOB) ? (unsigned)e.x + koff : OOB;
      else off = ((unsigned)e.y & tapbit) ? (unsigned)e.x + koff : OOB;
      ra[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int n = nb0 + 64 * i;
      rb[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (k_ok && n < p.b_rows) ? (unsigned)(n * p.ldb + k) * 2u : OOB, 0, 0);
    }
  };
  auto store_stage = [&](const int set, unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int r = lrow + 64 * i;
      *(u32x4*)(buf + r * 128 + ((kq ^ swz(r)) << 4)) = ra[set][i];
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int r = lrow + 64 * i;
      *(u32x4*)(buf + A_BYTES + r * 128 + ((kq ^ swz(r)) << 4)) = rb[set][i];
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](const unsigned char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fsw = ((ks * 4 + fq) ^ swz(fr)) << 4;      // (swz(row) only sees the row's low four bits)
      h16x8 bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = *(const h16x8*)(buf + A_BYTES + (wj * 64 + b * 16 + fr) * 128 + fsw);
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const h16x8 af = *(const h16x8*)(buf + (wi * 128 + a * 16 + fr) * 128 + fsw);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = PP_MFMA16(af, bf[b], acc[a][b], 0, 0, 0);
        if ((a & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // (keep the fragment reads from piling up at the top)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // pipeline: K-step s is loaded into register set s & 1 three iterations ahead, written to LDS buffer s & 1 after step s - 1
  // has been multiplied, multiplied one iteration later (loads past the last step fetch nothing)
  const int nk = (p.K + BK - 1) / BK;
  load_stage(0, 0);
  store_stage(0, smem);
  load_stage(1, 1);
  load_stage(0, 2);
  __syncthreads();
  auto iteration = [&](const int st, const int set, unsigned char* cur, unsigned char* nxt) __attribute__((always_inline)) {
    compute(cur);
    store_stage(set, nxt);
    load_stage(set, st + 3);
    __syncthreads();
  };
  for (int st = 0; st < nk; st += 2) {
    iteration(st, 1, smem, smem + BUF);
    if (st + 1 < nk) iteration(st + 1, 0, smem + BUF, smem);
  }

  // ---- epilogue: same order of operations and roundings as igemm_kernel's (bias, [pre-activation copy], activation, bf16,
  // dropout, residual); each wave stages 16 x 64 of its tile at a time and writes whole 16-byte row segments
  const int ncols_store = (p.N + 7) & ~7;
  const uint32_t drop_thr = FULL ? (uint32_t)(p.drop_p * 65536.f + 0.5f) : 0u;
  const float drop_scale = 1.f / (1.f - p.drop_p);
  const int m_wave = mb * GBM + wi * 128, n_wave = nb * GBN + wj * 64;
  if (FULL && p.bias) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n_wave + b * 16 + fr;
      const float bv = n < p.N ? p.bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += bv;
    }
  }
  unsigned char* stg = smem + wave * 16 * STG_STRIDE;       // (wave-private; the loop's last barrier freed the buffers)
  unsigned char* stg_w = stg + (fq * 4) * STG_STRIDE + fr * 2;
  auto write_out = [&](h16raw* Cout, const h16raw* residual) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 8; ++a) {      // (unrolled: a rolled loop indexes the accumulators dynamically, i.e. through scratch)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) *(h16raw*)(stg_w + r * STG_STRIDE + b * 32) = f2h(acc[a][b][r]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int cid = lane + 64 * it;
        const int row = cid >> 3, ch = cid & 7;
        const int m = m_wave + a * 16 + row;
        const int col = n_wave + ch * 8;
        if (m < p.M && col < ncols_store) {
          uint4 v = *(const uint4*)(stg + row * STG_STRIDE + ch * 16);
          long long orow = m;
          if (MODE != PP_DENSE && p.omap) {   // scatter compact rows of a parity class into the full tensor
            const uint32_t t1 = fdiv((uint32_t)m, rd.dRw);
            const int rw = m - (int)t1 * g.Rw;
            const uint32_t t2 = fdiv(t1, rd.dRh);
            const int rh = (int)t1 - (int)t2 * g.Rh;
            const int n = (int)fdiv(t2, rd.dRt);
            const int rt = (int)t2 - n * g.Rt;
            orow = (((long long)n * p.Ot + rt * p.os_t + p.oo_t) * p.Oh + rh * p.os_h + p.oo_h) * p.Ow + rw * p.os_w + p.oo_w;
          }
          if (FULL && drop_thr && Cout == (h16raw*)p.C) {   // (not the pre-activation copy)
            float x[8];
            bool keep[8];
            unpack8(v, x);
            keep8(p.drop_seed, (orow * p.ldc + col) >> 3, drop_thr, keep);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = keep[q] ? x[q] * drop_scale : 0.f;
            v = pack8(x);
          }
          if (FULL && residual) {
            const uint4 rv = *(const uint4*)(residual + orow * p.ldr + col);
            float x[8], y[8];
            unpack8(v, x);
            unpack8(rv, y);
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] += y[q];
            v = pack8(x);
          }
          if (out_nt) {
            u32x4 w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, (u32x4*)(Cout + orow * p.ldc + col));
          } else {
            *(uint4*)(Cout + orow * p.ldc + col) = v;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  };
  if (FULL) {
    if (p.Cpre) write_out((h16raw*)p.Cpre, nullptr);
    if (p.act == PP_ACT_GELU) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] = gelu_f(acc[a][b][r]);
    } else if (p.act == PP_ACT_RELU) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] = fmaxf(acc[a][b][r], 0.f);
    }
    write_out((h16raw*)p.C, (const h16raw*)p.residual);
  } else {
    write_out((h16raw*)p.C, nullptr);
  }
}

int pick_wn(int n16) {
  // padded tile count weighted by a per-shape efficiency guess (narrow tiles re-read A more often)
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

RowDiv make_rowdiv(const pp_igemm_desc& d) {
  RowDiv rd;
  const bool dense = d.g.mode == PP_DENSE;
  rd.dRw = make_fastdiv((uint32_t)(dense ? 1 : d.g.Rw));
  rd.dRh = make_fastdiv((uint32_t)(dense ? 1 : d.g.Rh));
  rd.dRt = make_fastdiv((uint32_t)(dense ? 1 : d.g.Rt));
  return rd;
}

template <int WN>
int launch_wn(const pp_igemm_desc& d, hipStream_t s) {
  // 8-wave (256-row) register-staged workgroups were measured 3-10 % slower than 4-wave ones on the hot shapes, so
  // that family only has 4-wave variants; the 256-row tiles belong to the LDS-DMA ring variant (launch_ring).
  constexpr int NW = 4, BM = 32 * NW;
  const int nblk_n = (d.N + 16 * WN - 1) / (16 * WN);
  const long long nblk_m = ((long long)d.M + BM - 1) / BM;
  const long long ntiles = nblk_m * nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) { pp_set_error("pp_igemm: grid too large"); return PP_ERR_INVALID; }
  // persistent grid: as many workgroups as stay resident (LDS-limited: 3 / 2 / 1 per CU) once there are >= 4 waves of
  // tiles, otherwise one tile per workgroup
  const bool full = d.bias || d.act != PP_ACT_NONE || d.residual || d.Cpre || d.drop_p > 0.f;
  const long long resident = 256LL * (WN <= 4 ? 3 : (WN <= 9 ? 2 : 1));
  long long gx = ntiles;
  if (pp_opt_persistent && !full && d.nbatch == 1 && ntiles >= 4 * resident) gx = resident;
  dim3 grid((unsigned)gx, 1, (unsigned)d.nbatch), block(64 * NW);
  const RowDiv rd = make_rowdiv(d);
#define PP_LAUNCH_IGEMM(MODE_)                                                                                       \
  if (full) hipLaunchKernelGGL((igemm_kernel<WN, MODE_, true, NW, false>), grid, block, 0, s, d, nblk_n, rd, pp_opt_xcd_remap_igemm, (int)ntiles, out_nt_for(d)); \
  else hipLaunchKernelGGL((igemm_kernel<WN, MODE_, false, NW, false>), grid, block, 0, s, d, nblk_n, rd, pp_opt_xcd_remap_igemm, (int)ntiles, out_nt_for(d))
  switch (d.g.mode) {
    case PP_DENSE: PP_LAUNCH_IGEMM(PP_DENSE); break;
    case PP_CONV_FWD: PP_LAUNCH_IGEMM(PP_CONV_FWD); break;
    case PP_CONV_DGRAD: PP_LAUNCH_IGEMM(PP_CONV_DGRAD); break;
    default: pp_set_error("pp_igemm: bad gather mode %d", d.g.mode); return PP_ERR_INVALID;
  }
#undef PP_LAUNCH_IGEMM
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// LDS-DMA ring variant: 256-row tiles, one persistent workgroup per CU
template <int WN>
int launch_ring(const pp_igemm_desc& d, hipStream_t s) {
  constexpr int BM = 256;
  const int nblk_n = (d.N + 16 * WN - 1) / (16 * WN);
  const long long nblk_m = ((long long)d.M + BM - 1) / BM;
  const long long ntiles = nblk_m * nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) { pp_set_error("pp_igemm: grid too large"); return PP_ERR_INVALID; }
  const long long gx = ntiles < pp_opt_persist_cus ? ntiles : pp_opt_persist_cus;
  dim3 grid((unsigned)gx, 1, (unsigned)d.nbatch), block(512), pblock(768);
  const RowDiv rd = make_rowdiv(d);
  const bool full = d.bias || d.act != PP_ACT_NONE || d.residual || d.Cpre || d.drop_p > 0.f;
  // producer form (pp_opt_ring_producers): tiles up to 128 columns -- the 144-column tile's multipliers do not fit 168 registers
  const bool prod = pp_opt_ring_producers && WN <= 9;
#define PP_LAUNCH_RING(MODE_, FULL_)                                                                                                   \
  do {                                                                                                                                 \
    if constexpr (WN <= 9) {                                                                                                           \
      if (prod) {                                                                                                                      \
        hipLaunchKernelGGL((igemm_kernel<WN, MODE_, FULL_, 8, true, true>), grid, pblock, 0, s, d, nblk_n, rd, pp_opt_xcd_remap_igemm, \
                           (int)ntiles, out_nt_for(d));                                                                                \
        break;                                                                                                                         \
      }                                                                                                                                \
    }                                                                                                                                  \
    hipLaunchKernelGGL((igemm_kernel<WN, MODE_, FULL_, 8, true>), grid, block, 0, s, d, nblk_n, rd, pp_opt_xcd_remap_igemm,            \
                       (int)ntiles, out_nt_for(d));                                                                                    \
  } while (0)
  switch (d.g.mode) {   // (fused epilogues: dense, and conv-forward at 128 columns -- the other conv forms would spill)
    case PP_DENSE:
      if (full) PP_LAUNCH_RING(PP_DENSE, true);
      else PP_LAUNCH_RING(PP_DENSE, false);
      break;
    case PP_CONV_FWD:
      if constexpr (WN == 8) {
        if (full) { PP_LAUNCH_RING(PP_CONV_FWD, true); break; }
      }
      if (full) { pp_set_error("pp_igemm: internal: fused conv epilogue on a ring tile without one"); return PP_ERR_INVALID; }
      if constexpr (WN >= 8) { PP_LAUNCH_RING(PP_CONV_FWD, false); break; }
      pp_set_error("pp_igemm: internal: conv gather on a dense-only ring tile");
      return PP_ERR_INVALID;
    case PP_CONV_DGRAD:
      if (full) { pp_set_error("pp_igemm: internal: fused conv epilogue on a ring tile without one"); return PP_ERR_INVALID; }
      if constexpr (WN >= 8) { PP_LAUNCH_RING(PP_CONV_DGRAD, false); break; }
      pp_set_error("pp_igemm: internal: conv gather on a dense-only ring tile");
      return PP_ERR_INVALID;
    default: pp_set_error("pp_igemm: bad gather mode %d", d.g.mode); return PP_ERR_INVALID;
  }
#undef PP_LAUNCH_RING
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// does the 256 x 256 tile take this problem?  (pp_opt_igemm_big = the M from which it does, 0 = never)
bool igemm_big_ok(const pp_igemm_desc& d) {
  const pp_gather& g = d.g;
  if (!pp_opt_igemm_big || (long long)d.M < (pp_opt_igemm_big < 0 ? -pp_opt_igemm_big : pp_opt_igemm_big)) return false;
  if (d.c_fp32 || d.nbatch != 1 || d.inner != 1 || d.colstats || d.a_bn_scale || d.a_bn_shift) return false;
  // fused epilogues (bias / activation / pre-activation copy / dropout / residual) are built and bit-identical, but one tile
  // per workgroup leaves their two staged outputs uncovered: the audio convolutions' forward (GELU + the saved
  // pre-activation) measured 627 -> 618 / 254 -> 289 us, so only plain epilogues are sent here (a NEGATIVE option value
  // sends the fused ones too: tools/probe/igemm_big.py)
  const bool full = d.bias || d.act != PP_ACT_NONE || d.residual || d.Cpre || d.drop_p > 0.f;
  if (full && pp_opt_igemm_big > 0) return false;
  if (d.K < 2 * BK || d.N < 192) return false;
  if (g.mode != PP_DENSE) {
    if (g.kt * g.kh * g.kw > 32) return false;
    if (g.mode == PP_CONV_DGRAD && !(g.st == 1 && g.sh == 1 && g.sw == 1)) return false;
  }
  const long long tiles = (((long long)d.M + GBM - 1) / GBM) * ((d.N + GBN - 1) / GBN);
  return tiles >= 192;       // (at least three quarters of the CUs busy; M >= 32 768 makes that >= 128 row blocks anyway)
}

int launch_igemm_big(const pp_igemm_desc& d, hipStream_t s) {
  const int nblk_n = (d.N + GBN - 1) / GBN;
  const long long ntiles = (((long long)d.M + GBM - 1) / GBM) * nblk_n;
  if (ntiles <= 0 || ntiles > 0x7fffffffLL) { pp_set_error("pp_igemm: grid too large"); return PP_ERR_INVALID; }
  const RowDiv rd = make_rowdiv(d);
  BigArgs ba;
  ba.dcg = make_fastdiv((uint32_t)(d.g.mode == PP_DENSE ? 1 : d.g.cg));
  const bool full = d.bias || d.act != PP_ACT_NONE || d.residual || d.Cpre || d.drop_p > 0.f;
  dim3 grid((unsigned)ntiles), block(512);
#define PP_LAUNCH_BIG(MODE_)                                                                                                            \
  do {                                                                                                                                  \
    if (full) hipLaunchKernelGGL((igemm_big_kernel<MODE_, true>), grid, block, 0, s, d, rd, ba, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, out_nt_for(d)); \
    else hipLaunchKernelGGL((igemm_big_kernel<MODE_, false>), grid, block, 0, s, d, rd, ba, nblk_n, (int)ntiles, pp_opt_xcd_remap_igemm, out_nt_for(d));     \
  } while (0)
  switch (d.g.mode) {
    case PP_DENSE: PP_LAUNCH_BIG(PP_DENSE); break;
    case PP_CONV_FWD: PP_LAUNCH_BIG(PP_CONV_FWD); break;
    default: PP_LAUNCH_BIG(PP_CONV_DGRAD); break;
  }
#undef PP_LAUNCH_BIG
  PP_LAUNCH_CHECK();
  return PP_OK;
}

}  // namespace

int pp_validate_gather(const pp_gather& g, int K, const char* who) {
  if (g.mode == PP_DENSE) {
    PP_CHECK_ARG(g.lda > 0 && g.lda % 8 == 0, "%s: dense lda=%d must be a positive multiple of 8", who, g.lda);
    return PP_OK;
  }
  PP_CHECK_ARG(g.mode == PP_CONV_FWD || g.mode == PP_CONV_DGRAD, "%s: bad gather mode %d", who, g.mode);
  PP_CHECK_ARG(g.Rt > 0 && g.Rh > 0 && g.Rw > 0 && g.Gt > 0 && g.Gh > 0 && g.Gw > 0, "%s: bad extents", who);
  PP_CHECK_ARG(g.kt > 0 && g.kh > 0 && g.kw > 0 && g.kt * g.kh * g.kw <= 256 && g.kt < 256 && g.kh < 256 &&
                   g.kw < 256, "%s: taps %dx%dx%d unsupported (<=256 total)", who, g.kt, g.kh, g.kw);
  PP_CHECK_ARG(g.cg > 0 && g.cg % 8 == 0 && g.cstride % 8 == 0 && g.cstride >= g.cg,
               "%s: cg=%d cstride=%d must be multiples of 8", who, g.cg, g.cstride);
  PP_CHECK_ARG(K == g.kt * g.kh * g.kw * g.cg, "%s: K=%d != taps*cg=%d", who, K, g.kt * g.kh * g.kw * g.cg);
  if (g.mode == PP_CONV_DGRAD)
    PP_CHECK_ARG((g.st == 1 || g.st == 2) && (g.sh == 1 || g.sh == 2) && (g.sw == 1 || g.sw == 2),
                 "%s: dgrad strides must be 1 or 2", who);
  else
    PP_CHECK_ARG(g.st > 0 && g.sh > 0 && g.sw > 0, "%s: bad strides", who);
  return PP_OK;
}

extern "C" int pp_igemm_abn_supported(const pp_igemm_desc* dp) {
  if (!dp) return 0;
  pp_igemm_desc d = *dp;
  if (d.b_rows <= 0) d.b_rows = d.N;
  if (d.nbatch <= 0) d.nbatch = 1;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || pp_validate_gather(d.g, d.K, "pp_igemm_abn_supported") != PP_OK) return 0;
  return pp_igemm_abn_ok(d);
}

extern "C" long long pp_igemm_stat_rows(const pp_igemm_desc* dp) {
  if (!dp || dp->M <= 0) return 0;
  pp_igemm_desc d = *dp;
  if (d.b_rows <= 0) d.b_rows = d.N;
  if (d.nbatch <= 0) d.nbatch = 1;
  if (d.g.mode == PP_DENSE || pp_validate_gather(d.g, d.K, "pp_igemm_stat_rows") != PP_OK) return ((long long)d.M + 127) / 128;
  return pp_igemm_win_stat_rows(d);
}

extern "C" int pp_igemm(const pp_igemm_desc* dp, pp_stream_t stream) {
  PP_CHECK_ARG(dp != nullptr, "pp_igemm: null descriptor");
  pp_igemm_desc d = *dp;
  PP_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "pp_igemm: bad sizes M=%d N=%d K=%d", d.M, d.N, d.K);
  PP_CHECK_ARG(d.K % 8 == 0, "pp_igemm: K=%d must be a multiple of 8", d.K);
  PP_CHECK_ARG(d.A && d.Bt && d.C, "pp_igemm: null operand");
  PP_CHECK_ARG(d.ldb % 8 == 0 && d.ldb >= d.K, "pp_igemm: ldb=%d must be a multiple of 8 and >= K", d.ldb);
  if (d.b_rows <= 0) d.b_rows = d.N;
  if (d.nbatch <= 0) d.nbatch = 1;
  if (d.inner <= 0) d.inner = 1;
  if (!d.c_fp32) {
    PP_CHECK_ARG(d.ldc % 8 == 0 && d.ldc >= ((d.N + 7) & ~7), "pp_igemm: ldc=%d too small / unaligned for N=%d", d.ldc, d.N);
    PP_CHECK_ARG(((uintptr_t)d.C & 15) == 0, "pp_igemm: C must be 16-byte aligned");
    if (d.residual) PP_CHECK_ARG(d.ldr % 8 == 0, "pp_igemm: residual needs ldr%%8==0 (batched: same offsets as C)");
  } else {
    PP_CHECK_ARG(!d.residual && !d.Cpre && !d.colstats, "pp_igemm: fp32 output supports bias/act only");
  }
  PP_CHECK_ARG(d.drop_p >= 0.f && d.drop_p < 1.f, "pp_igemm: drop_p=%f", (double)d.drop_p);
  if (d.drop_p > 0.f) PP_CHECK_ARG(!d.c_fp32 && !d.colstats && !d.omap && d.nbatch == 1,
                                   "pp_igemm: epilogue dropout needs a plain bf16 output (no fp32 / statistics / row map / batches)");
  if (d.omap) PP_CHECK_ARG(d.g.mode != PP_DENSE && !d.c_fp32 && !d.colstats && d.os_t > 0 && d.os_h > 0 && d.os_w > 0,
                           "pp_igemm: the output row map needs a conv gather and a plain bf16 store");
  if (d.colstats) PP_CHECK_ARG(!d.bias && d.act == PP_ACT_NONE && !d.residual && !d.Cpre && d.nbatch == 1 && d.ldstat >= d.N,
                               "pp_igemm: colstats needs a plain epilogue (no bias/act/residual/pre) and nbatch 1");
  PP_CHECK_ARG(((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.Bt & 15) == 0, "pp_igemm: operands must be 16-byte aligned");
  if (d.bnr_partials)
    PP_CHECK_ARG(d.bnr_y && d.bnr_mean && d.bnr_rstd && d.bnr_scale && d.bnr_shift && !d.c_fp32 && !d.omap && d.nbatch == 1 &&
                     !d.colstats && !d.bias && d.act == PP_ACT_NONE && d.drop_p == 0.f && ((uintptr_t)d.bnr_y & 15) == 0 &&
                     (!d.bnr_z || ((uintptr_t)d.bnr_z & 15) == 0),
                 "pp_igemm: the BatchNorm-backward sums need y + mean / rstd / scale / shift and a plain bf16 data-gradient store");
  const int rc = pp_validate_gather(d.g, d.K, "pp_igemm");
  if (rc != PP_OK) return rc;
  if (d.g.mode != PP_DENSE) {
    const long long rows = (long long)d.g.Rt * d.g.Rh * d.g.Rw;
    PP_CHECK_ARG(d.M % rows == 0, "pp_igemm: M=%d is not a multiple of Rt*Rh*Rw=%lld", d.M, rows);
  }
  if (d.g.mode != PP_DENSE) {
    const long long src = (long long)(d.M / ((long long)d.g.Rt * d.g.Rh * d.g.Rw)) * d.g.Gt * d.g.Gh * d.g.Gw * d.g.cstride;
    PP_CHECK_ARG(src < 0x7ffffff0LL, "pp_igemm: gathered tensor has %lld elements (>= 2^31)", src);
  } else {
    PP_CHECK_ARG((long long)d.M * d.g.lda < 0x7fffffffLL, "pp_igemm: dense operand >= 2^31 elements");
  }
  PP_CHECK_ARG((long long)d.b_rows * d.ldb < 0x7ffffff0LL, "pp_igemm: Bt has >= 2^31 elements");
  if (d.a_bn_scale || d.a_bn_shift)
    PP_CHECK_ARG(d.a_bn_scale && d.a_bn_shift && pp_igemm_abn_ok(d),
                 "pp_igemm: a_bn_scale / a_bn_shift (BatchNorm apply of A's producer) is not available for this problem: "
                 "ask pp_igemm_abn_supported first");
  hipStream_t s = (hipStream_t)stream;
  if (pp_opt_win_igemm && (long long)d.M >= pp_opt_win_igemm) {   // (1,3,3) stride-1 convs: A window in LDS
    const int rc_win = pp_igemm_win_try(d, s);
    if (rc_win != 1) return rc_win;
  }
  if (igemm_big_ok(d)) {      // large-M problems: 256 x 256 tiles (half the L2 traffic of the ring tile per FLOP)
    const int r = launch_igemm_big(d, s);
    return (r == PP_OK && d.bnr_partials) ? PP_BNR_SKIPPED : r;
  }
  const int n16 = (d.N + 15) / 16;
  const bool full = d.bias || d.act != PP_ACT_NONE || d.residual || d.Cpre || d.drop_p > 0.f;
  if (pp_opt_ring && !d.c_fp32 && d.nbatch == 1 && n16 > 4 && d.K >= 2 * BK) {
    // Ring tiles are 128 or 144 columns wide (whichever pads N less).  Narrow outputs (N <= 64) stay on the
    // register-staged kernel, which measured faster there; so do GEMMs too small to give every CU a 256-row tile.
    // Fused epilogues (bias / activation / residual): dense at either width, conv-forward at 128 columns only.
    const int c8 = ((n16 + 7) / 8) * 8, c9 = ((n16 + 8) / 9) * 9;
    int best = c9 <= c8 ? 9 : 8;
    const long long mblk = ((long long)d.M + 255) / 256;
    if (d.g.mode == PP_DENSE) {
      // Dense: the width also decides how many rounds of 256 tiles the persistent workgroups walk.  M = 7296 x N = 768
      // (the transformer's output projections) is 174 tiles of 128 columns -- two thirds of the CUs, one round -- but 232
      // tiles of 96 columns, also one round, each a quarter less work.  Cost of a round ~ A rows + B rows streamed per
      // K-step and the MFMAs, both linear in the width: (8 + width) fits the measured 96 / 128 / 144 column tiles.
      long long best_cost = 0;
      for (const int wn : {9, 8, 6}) {
        const long long tiles = mblk * ((n16 + wn - 1) / wn);
        const long long cost = ((tiles + 255) / 256) * (8 + wn);
        if (best_cost == 0 || cost < best_cost) { best_cost = cost; best = wn; }
      }
      if (pp_opt_ring_wn) best = pp_opt_ring_wn;
    }
    const bool epi_ok = !full || d.g.mode == PP_DENSE || (d.g.mode == PP_CONV_FWD && best == 8);
    const long long tiles = mblk * ((n16 + best - 1) / best);
    if (epi_ok && tiles >= pp_opt_ring) {
      int r;
      if (best == 6) {
        if (d.g.mode != PP_DENSE) { pp_set_error("pp_igemm: internal: 96-column ring tiles are dense only"); return PP_ERR_INVALID; }
        r = launch_ring<6>(d, s);
      } else {
        r = best == 9 ? launch_ring<9>(d, s) : launch_ring<8>(d, s);
      }
      return (r == PP_OK && d.bnr_partials) ? PP_BNR_SKIPPED : r;
    }
  }
  int r;
  switch (pick_wn(n16)) {
    case 15: r = launch_wn<15>(d, s); break;
    case 9: r = launch_wn<9>(d, s); break;
    case 8: r = launch_wn<8>(d, s); break;
    case 4: r = launch_wn<4>(d, s); break;
    case 3: r = launch_wn<3>(d, s); break;
    default: r = launch_wn<2>(d, s); break;
  }
  return (r == PP_OK && d.bnr_partials) ? PP_BNR_SKIPPED : r;   // (these kernels do not take the BatchNorm-backward sums)
}
