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
void pp_wgrad_slab_sum(const float* ws, int nsplit, long long slab_floats, int Ni, int Kj, int ldw, float* dW, hipStream_t s);

namespace {

constexpr int MS = 64;                 // rows (positions hw) per step
constexpr int NWV = 9;
constexpr int NT = 64 * NWV;
constexpr int CB = 144;                // X channels per workgroup (9 column tiles per tap)
constexpr int XS = CB * 2;             // X block row stride: 288 = 32 x 9 (odd) -> conflict-free tr reads
constexpr int XSLOT = MS * XS;
constexpr int XPIECES = XSLOT / 1024;  // 18
constexpr int NPSLOT = 3;
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) buffer_rsrc;

__device__ __forceinline__ void lds_dma16(const buffer_rsrc rs, unsigned char* dst, const unsigned off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)dst, 16, off, 0, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt_dyn(const int n) {   // n is wave-uniform, 0..4 here
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
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
};

// BNA: X is the raw output y of a BatchNorm unit; every X block gets z = relu?(y * scale + shift) applied once, in LDS,
// after it has landed and before its first use (pp_wgrad_desc.x_bn_*): the activated tensor is never materialised.
template <int WI, bool BNA>
__global__ __launch_bounds__(NT, 1) void wgrad_tw_kernel(const h16raw* __restrict__ X, const h16raw* __restrict__ dY,
                                                          float* __restrict__ dW, const TwGeom g, const int Ni,
                                                          const int ldy, const int ldw, const int nblk_i,
                                                          const int nblk_c, const int steps_per_split,
                                                          const int xcd_remap, const float* __restrict__ bn_scale,
                                                          const float* __restrict__ bn_shift, const int bn_relu,
                                                          float* __restrict__ slab) {
  constexpr int TI = 16 * WI;
  constexpr int PS = (WI & 1) ? TI * 2 : TI * 2 + 32;
  constexpr int PSLOT = MS * PS;
  constexpr int PPIECES = PSLOT / 1024;
  static_assert(PSLOT % 1024 == 0, "dY slab = whole DMA pieces");
  constexpr int NPIECES = PPIECES + XPIECES;
  constexpr int NK = (NPIECES + NWV - 1) / NWV;
  // X ring: blocks s - 1, s, s + 1 in use and two in flight; BNA: three in flight, so that block s + 2 has landed when
  // iteration s starts and its BatchNorm pass can ride inside the iteration, published by the iteration's own barrier
  constexpr int NXSLOT = BNA ? 6 : 5;
  constexpr int XA = BNA ? 4 : 3;                  // the X block issued at iteration s is s + XA
  constexpr int X_BYTES = NXSLOT * XSLOT;
  constexpr int SMEM = X_BYTES + NPSLOT * PSLOT;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(16 * 3 * CB * 4 <= SMEM, "epilogue stage");
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];   // one LDS object (see igemm.hip)
  unsigned char* const pring = smem + X_BYTES;

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
  const int i0 = ib * TI, c0 = cb * CB;
  const int s_begin = split * steps_per_split;
  const int s_end = min(g.NS, s_begin + steps_per_split);
  const int nsteps = s_end - s_begin;
  if (nsteps <= 0) return;

  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, (short)0, (int)OOB, 0x00020000);
  const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)dY, (short)0, (int)OOB, 0x00020000);

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

  // ---- this lane's place in the DMA pieces its wave issues per step (piece q = wave + 9 k) ---------------------------
  int d_row[NK];
  unsigned d_col[NK];
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
      const int c = c0 + (o % XS) / 2;
      d_col[k] = c < g.cg ? (unsigned)c * 2u : OOB;
    }
  }
  npiece = __builtin_amdgcn_readfirstlane(npiece);
  // issue the pieces of dY step sp and of X block sx (absolute step indices); either may be out of range -> zeros
  auto dma_pair = [&](const int sp, const int sx, const bool do_p) __attribute__((always_inline)) {
    const bool p_ok = do_p && sp >= s_begin && sp < s_end;
    const bool x_ok = sx >= 0 && sx < g.NS;
    const StepPos pp = decode(p_ok ? sp : 0), px = decode(x_ok ? sx : 0);
    unsigned char* const pdst = pring + ((sp - s_begin + NPSLOT) % NPSLOT) * PSLOT;
    unsigned char* const xdst = smem + ((sx + NXSLOT) % NXSLOT) * XSLOT;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int q = wave + NWV * k;
      if (q < PPIECES) {
        if (do_p) {
          const bool ok = p_ok & (d_col[k] != OOB) & (d_row[k] < pp.nvalid);
          lds_dma16(rsY, pdst + q * 1024, ok ? (unsigned)((pp.m0 + d_row[k]) * ldy) * 2u + d_col[k] : OOB);
        }
      } else if (q < NPIECES) {
        const bool ok = x_ok & (d_col[k] != OOB) & (d_row[k] < px.nvalid);
        lds_dma16(rsX, xdst + (q - PPIECES) * 1024, ok ? (unsigned)((px.m0 + d_row[k]) * g.cstride) * 2u + d_col[k] : OOB);
      }
    }
  };

  // ---- fragment addressing (tr_frag of wgrad.hip) ---------------------------------------------------------------------
  const int gq = lane >> 4, li = lane & 15;
  const int frow = 4 * gq + (li >> 2);
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)smem;
  const unsigned p_lane = lds0 + X_BYTES + (unsigned)(frow * PS + (li & 3) * 8);
  const int my_tap = wave / 3;                                            // 0, 1, 2  <->  frame t - 1, t, t + 1
  const unsigned q_lane = lds0 + (unsigned)(frow * XS + (wave % 3) * 96 + (li & 3) * 8);

  f32x4 acc[WI][3];
#pragma unroll
  for (int a = 0; a < WI; ++a)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) acc[a][jt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](const int s) __attribute__((always_inline)) {
    const int col = (int)fdiv((uint32_t)s, g.dT_);
    const int t = s - col * g.T;
    if ((unsigned)(t + my_tap - 1) >= (unsigned)g.T) return;              // this wave's tap leaves the clip
    const unsigned xslot = (unsigned)(((s + my_tap - 1 + NXSLOT) % NXSLOT) * XSLOT);
    const unsigned pslot = (unsigned)(((s - s_begin) % NPSLOT) * PSLOT);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned qa = q_lane + xslot + (unsigned)(sub * 32 * XS);
      const unsigned pa = p_lane + pslot + (unsigned)(sub * 32 * PS);
      u32x2 qlo[3], qhi[3], plo[2], phi[2];
      ds_read_tr<0>(qlo[0], qa);  ds_read_tr<16 * XS>(qhi[0], qa);
      ds_read_tr<32>(qlo[1], qa); ds_read_tr<32 + 16 * XS>(qhi[1], qa);
      ds_read_tr<64>(qlo[2], qa); ds_read_tr<64 + 16 * XS>(qhi[2], qa);
      ds_read_tr<0>(plo[0], pa);
      ds_read_tr<16 * PS>(phi[0], pa);
      asm volatile("s_waitcnt lgkmcnt(2)"
                   : "+v"(qlo[0]), "+v"(qhi[0]), "+v"(qlo[1]), "+v"(qhi[1]), "+v"(qlo[2]), "+v"(qhi[2])
                   :
                   : "memory");
      h16x8 qv[3];
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) qv[jt] = __builtin_bit_cast(h16x8, (u32x4){qlo[jt][0], qlo[jt][1], qhi[jt][0], qhi[jt][1]});
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
        for (int jt = 0; jt < 3; ++jt) acc[a][jt] = PP_MFMA16(pv, qv[jt], acc[a][jt], 0, 0, 0);
      });
    }
  };

  // ---- BNA: a block is 64 rows x 18 octets of 8 channels = 1152 16-byte chunks, two per thread; thread t keeps octet
  // t % 18 (576 = 32 x 18), so its sixteen parameters live in registers.  Rows past the frame and channels past cg hold
  // zeros that the pass may turn into relu(shift): their dY rows / dW columns are zero / unwritten.
  float bsc[8], bsh[8];
  const int b_c8 = tid % (CB / 8), b_row = tid / (CB / 8);
  if (BNA) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + b_c8 * 8 + q;
      bsc[q] = c < g.cg ? bn_scale[c] : 0.f;
      bsh[q] = c < g.cg ? bn_shift[c] : 0.f;
    }
  }
  auto bn_block = [&](const int sx) __attribute__((always_inline)) {     // X block of absolute step sx, in place
    const unsigned base = lds0 + (unsigned)(((sx + NXSLOT) % NXSLOT) * XSLOT + b_c8 * 16);
    // (inline asm: plain LDS accesses make hipcc wait for the LDS-DMAs in flight first)
    const unsigned a0 = base + (unsigned)(b_row * XS), a1 = a0 + (unsigned)(32 * XS);
    u32x4 v0, v1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
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

  // ---- prologue: X blocks s_begin - 1 .. s_begin + XA - 1, dY steps s_begin, s_begin + 1 ----------------------------
  dma_pair(0, s_begin - 1, false);
  dma_pair(0, s_begin, false);
  dma_pair(s_begin, s_begin + 1, true);
  dma_pair(s_begin + 1, s_begin + 2, true);
  if (BNA) dma_pair(0, s_begin + 3, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (BNA) {                      // blocks s_begin - 1 .. s_begin + 1; block s + 2 follows inside iteration s
    bn_block(s_begin - 1);
    bn_block(s_begin);
    bn_block(s_begin + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- main loop.  Iteration s: issue dY step s + 2 and X block s + XA (its slot held block s - 2), multiply step s, [BNA: activate
  // block s + 2, which landed before the barrier that ended iteration s - 1 and is first read at step s + 1],
  // wait for everything issued before this iteration, barrier. --------------------------------------------------------
  for (int s = s_begin; s < s_end; ++s) {
    dma_pair(s + 2, s + XA, true);
    compute(s);
    if (BNA) bn_block(s + 2);       // (behind the step's MFMAs: its LDS round trip and VALU work run while they drain)
    wait_vmcnt_dyn(npiece);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: 16 rows of dW at a time through LDS; an atomic wave-instruction adds 256 contiguous bytes ------------
  float* stage = (float*)smem;                 // 16 x 432 floats
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < WI; ++a) {
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      const int col = my_tap * CB + ((wave % 3) * 3 + jt) * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(fq * 4 + r) * (3 * CB) + col] = acc[a][jt][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 16 * 3 * CB; idx += NT) {
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

template <int WI>
int launch_tw(const pp_wgrad_desc& d, hipStream_t s, long long* ws_query) {
  const pp_gather& gg = d.g;
  TwGeom g;
  g.T = gg.Gt; g.HW = gg.Gh * gg.Gw; g.NHB = (g.HW + MS - 1) / MS;
  const int clips = d.M / (gg.Gt * g.HW);
  g.NS = clips * g.NHB * g.T;
  g.cstride = gg.cstride; g.cg = gg.cg;
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
  dim3 grid((unsigned)(tiles * msplit), 1, 1), block(NT);
  const bool slabs = pp_opt_deterministic && msplit > 1;
  const long long need = slabs ? (long long)msplit * d.Ni * d.ldw : 0;
  if (ws_query) { *ws_query = need; return PP_OK; }
  if (slabs) PP_CHECK_ARG(d.ws && d.ws_floats >= need, "pp_wgrad: deterministic mode needs ws of pp_wgrad_ws_floats(d) = %lld floats", need);
  float* const slab = slabs ? d.ws : nullptr;
  if (d.x_bn_scale) {
    if constexpr (WI == 4) {      // (64 rows of dW: the six-slot X ring does not fit beside the dY ring of 128)
      hipLaunchKernelGGL((wgrad_tw_kernel<WI, true>), grid, block, 0, s, (const h16raw*)d.X, (const h16raw*)d.dY, d.dW, g, d.Ni, d.ldy,
                         d.ldw, nblk_i, nblk_c, sps, pp_opt_xcd_remap_wgrad, d.x_bn_scale, d.x_bn_shift, d.x_bn_relu, slab);
    } else {
      pp_set_error("pp_wgrad: x_bn_scale / x_bn_shift need Ni <= 64 in the temporal sliding-window kernel");
      return PP_ERR_INVALID;
    }
  } else
    hipLaunchKernelGGL((wgrad_tw_kernel<WI, false>), grid, block, 0, s, (const h16raw*)d.X, (const h16raw*)d.dY, d.dW, g, d.Ni, d.ldy,
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
                        d.M % (g.Gt * frame) == 0 && (long long)d.M * g.cstride < 0x7fffffffLL &&
                        (long long)d.M * d.ldy < 0x7fffffffLL;
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
  return n16 <= 4 ? launch_tw<4>(d, s, ws_query) : launch_tw<8>(d, s, ws_query);
}
