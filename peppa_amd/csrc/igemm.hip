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
// Tiling: 128 x (16*WN) x 32 per 256-thread workgroup, 4 waves stacked along M (32 rows each),
// v_mfma_f32_16x16x32_bf16, register-staged global loads one K-step ahead of the MFMAs, LDS
// tiles with 64-byte rows XOR-swizzled for conflict-free ds_read_b128, epilogue staged through
// LDS so every global store is a full 16-byte row segment.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;

__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

struct RowInfo {
  int base;          // dense: row offset; conv: offset of (n, bt, bh, bw) in elements (may be "virtual")
  int nbase;         // conv: n * Gt*Gh*Gw (positions)
  int bt, bh, bw;
  int valid;
};

template <int WN, int MODE>
__global__ __launch_bounds__(256, (WN <= 9 ? 2 : 1)) void igemm_kernel(const pp_igemm_desc p, const int nblk_n) {
  constexpr int BN = 16 * WN;
  constexpr int A_BYTES = BM * 64;
  constexpr int B_BYTES = BN * 64;
  constexpr int STG_STRIDE = BN * 2 + 16;
  constexpr int STG_BYTES = 4 * 16 * STG_STRIDE;
  constexpr int STAT_BYTES = 4 * BN * 2 * 4;
  constexpr int LOOP_BYTES = A_BYTES + B_BYTES;
  constexpr int EPI_BYTES = STG_BYTES + STAT_BYTES;
  constexpr int SMEM = 2 * LOOP_BYTES > EPI_BYTES ? 2 * LOOP_BYTES : EPI_BYTES;
  constexpr int NBI = (BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  __shared__ int lut[128];      // packed (dt, dh, dw) per tap
  __shared__ int lut_off[128];  // element offset of the tap inside the source tensor (linear part)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int nb = blockIdx.x % nblk_n;
  const int mb = blockIdx.x / nblk_n;
  const int z = blockIdx.z;
  const int zo = z / p.inner, zi = z % p.inner;

  const bfraw* __restrict__ A = (const bfraw*)p.A + zo * p.a_s0 + zi * p.a_s1;
  const bfraw* __restrict__ Bt = (const bfraw*)p.Bt + zo * p.b_s0 + zi * p.b_s1;
  const long long c_off = zo * p.c_s0 + zi * p.c_s1;
  const float* __restrict__ bias = p.bias ? p.bias + zo * p.bias_s0 + zi * p.bias_s1 : nullptr;

  const pp_gather& g = p.g;
  const int ntaps = g.kt * g.kh * g.kw;
  if (MODE != PP_DENSE) {
    if (tid < 128) {
      int e = 0;
      if (tid < ntaps) {
        const int dw = tid % g.kw;
        const int t2 = tid / g.kw;
        const int dh = t2 % g.kh;
        const int dt = t2 / g.kh;
        e = dt | (dh << 8) | (dw << 16);
        lut_off[tid] = ((dt * g.Gh + dh) * g.Gw + dw) * g.cstride;
      }
      lut[tid] = e;
    }
    __syncthreads();
  }
  // stride-1 gathers are linear in the tap: offset = row_base +/- lut_off[tap] (32-bit element offsets;
  // the host checks that the source tensor has < 2^31 elements)
  const bool unit_stride = (g.st == 1 && g.sh == 1 && g.sw == 1);

  // ---- per-thread row bookkeeping (two A rows per thread) --------------------------------
  const int kq = tid & 3;
  RowInfo ri[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = mb * BM + (tid >> 2) + 64 * i;
    ri[i].valid = m < p.M;
    const int mm = ri[i].valid ? m : 0;
    if (MODE == PP_DENSE) {
      ri[i].base = mm * g.lda;
      ri[i].nbase = 0;
      ri[i].bt = ri[i].bh = ri[i].bw = 0;
    } else {
      const int rw = mm % g.Rw;
      int t = mm / g.Rw;
      const int rh = t % g.Rh;
      t /= g.Rh;
      const int rt = t % g.Rt;
      const int n = t / g.Rt;
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
      ri[i].base = (ri[i].nbase + (ri[i].bt * g.Gh + ri[i].bh) * g.Gw + ri[i].bw) * g.cstride;
    }
  }
  const int sft = g.st == 2, sfh = g.sh == 2, sfw = g.sw == 2;

  int kcur = kq * 8;            // this thread's k within the current K-step
  int tap = 0, cch = 0;         // conv modes: k = tap*cg + cch
  if (MODE != PP_DENSE) {
    tap = kcur / g.cg;
    cch = kcur % g.cg;
  }

  auto load_a = [&](int i) -> uint4 {
    const RowInfo& r = ri[i];
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (MODE == PP_DENSE) {
      if (r.valid && kcur < p.K) return *(const uint4*)(A + (r.base + kcur));
      return zero;
    } else {
      if (!r.valid || tap >= ntaps) return zero;
      const int e = lut[tap];
      const int dt = e & 0xff, dh = (e >> 8) & 0xff, dw = (e >> 16) & 0xff;
      if (MODE == PP_CONV_FWD) {
        const int gt = r.bt + dt, gh = r.bh + dh, gw = r.bw + dw;
        if ((unsigned)gt >= (unsigned)g.Gt || (unsigned)gh >= (unsigned)g.Gh || (unsigned)gw >= (unsigned)g.Gw)
          return zero;
        return *(const uint4*)(A + (r.base + lut_off[tap] + cch));
      } else {
        const int nt = r.bt - dt, nh = r.bh - dh, nw = r.bw - dw;
        if (unit_stride) {
          if ((unsigned)nt >= (unsigned)g.Gt || (unsigned)nh >= (unsigned)g.Gh || (unsigned)nw >= (unsigned)g.Gw)
            return zero;
          return *(const uint4*)(A + (r.base - lut_off[tap] + cch));
        }
        bool ok = (((nt & sft) | (nh & sfh) | (nw & sfw)) == 0) && nt >= 0 && nh >= 0 && nw >= 0;
        const int gt = nt >> sft, gh = nh >> sfh, gw = nw >> sfw;
        ok = ok && gt < g.Gt && gh < g.Gh && gw < g.Gw;
        if (!ok) return zero;
        return *(const uint4*)(A + ((r.nbase + (gt * g.Gh + gh) * g.Gw + gw) * g.cstride + cch));
      }
    }
  };
  const bfraw* brow_ptr[NBI];
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int brow = (tid >> 2) + 64 * i;
    const int n = nb * BN + brow;
    brow_ptr[i] = (brow < BN && n < p.b_rows) ? Bt + (long long)n * p.ldb : nullptr;
  }
  auto load_b = [&](int i) -> uint4 {
    if (brow_ptr[i] != nullptr && kcur < p.K) return *(const uint4*)(brow_ptr[i] + kcur);
    return make_uint4(0, 0, 0, 0);
  };
  auto advance_k = [&]() {
    kcur += BK;
    if (MODE != PP_DENSE) {
      cch += BK;
      while (cch >= g.cg) { cch -= g.cg; ++tap; }
    }
  };

  f32x4 acc[2][WN];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[mt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- main loop: two register stages of global loads in flight (prefetch distance 2) and two LDS
  // buffers, so each K-step costs one barrier and its loads had a full step (+ compute) to land.
  const int nk = (p.K + BK - 1) / BK;
  uint4 ra0[2], rb0[NBI], ra1[2], rb1[NBI];
  auto load_stage = [&](uint4* ra, uint4* rb) {
    ra[0] = load_a(0);
    ra[1] = load_a(1);
#pragma unroll
    for (int i = 0; i < NBI; ++i) rb[i] = load_b(i);
    advance_k();
  };
  auto store_stage = [&](const uint4* ra, const uint4* rb, unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 2) + 64 * i;
      *(uint4*)(buf + row * 64 + ((kq ^ swz(row)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
      const int row = (tid >> 2) + 64 * i;
      if (row < BN) *(uint4*)(buf + A_BYTES + row * 64 + ((kq ^ swz(row)) << 4)) = rb[i];
    }
  };
  const int fr = lane & 15, fq = lane >> 4;
  const int fsw = (fq ^ swz(fr)) << 4;
  auto compute = [&](const unsigned char* buf) {
    bf16x8 af[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) af[mt] = *(const bf16x8*)(buf + (wave * 32 + mt * 16 + fr) * 64 + fsw);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const bf16x8 bfm = *(const bf16x8*)(buf + A_BYTES + (j * 16 + fr) * 64 + fsw);
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bfm, acc[0][j], 0, 0, 0);
      acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bfm, acc[1][j], 0, 0, 0);
    }
  };
  unsigned char* buf0 = smem;
  unsigned char* buf1 = smem + LOOP_BYTES;
  load_stage(ra0, rb0);
  if (nk > 1) load_stage(ra1, rb1);
  store_stage(ra0, rb0, buf0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 2 < nk) load_stage(ra0, rb0);
    compute(buf0);
    if (kt + 1 < nk) store_stage(ra1, rb1, buf1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 3 < nk) load_stage(ra1, rb1);
    compute(buf1);
    if (kt + 2 < nk) store_stage(ra0, rb0, buf0);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------------------------
  const int m_wave = mb * BM + wave * 32;
  if (p.c_fp32) {
    float* C = (float*)p.C + c_off;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = nb * BN + j * 16 + fr;
        const float bv = (bias && n < p.N) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m_wave + mt * 16 + fq * 4 + r;
          float v = acc[mt][j][r] + bv;
          if (p.act == PP_ACT_GELU) v = gelu_f(v);
          else if (p.act == PP_ACT_RELU) v = fmaxf(v, 0.f);
          if (m < p.M && n < p.N) C[(long long)m * p.ldc + n] = v;
        }
      }
    return;
  }

  unsigned char* stg = smem + wave * 16 * STG_STRIDE;
  float* statbuf = (float*)(smem + STG_BYTES);
  const int ncols_store = (p.N + 7) & ~7;
  float s1[WN], s2[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) s1[j] = s2[j] = 0.f;
  const int npass = p.Cpre ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const bool pre_pass = (npass == 2 && pass == 0);
    bfraw* Cout = (bfraw*)(pre_pass ? p.Cpre : p.C) + c_off;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = j * 16 + fr;
        const int n = nb * BN + col;
        const float bv = (bias && n < p.N) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[mt][j][r] + bv;
          if (!pre_pass) {
            if (p.act == PP_ACT_GELU) v = gelu_f(v);
            else if (p.act == PP_ACT_RELU) v = fmaxf(v, 0.f);
          }
          const bfraw b = f2bf(v);
          *(bfraw*)(stg + (fq * 4 + r) * STG_STRIDE + col * 2) = b;
          if (p.colstats && !pre_pass) {
            const float fb = bf2f(b);
            s1[j] += fb;
            s2[j] += fb * fb;
          }
        }
      }
      __syncthreads();
      for (int cid = lane; cid < 32 * WN; cid += 64) {
        const int row = cid / (2 * WN);
        const int ch = cid % (2 * WN);
        const int m = m_wave + mt * 16 + row;
        const int col = nb * BN + ch * 8;
        if (m < p.M && col < ncols_store) {
          uint4 v = *(const uint4*)(stg + row * STG_STRIDE + ch * 16);
          if (p.residual && !pre_pass) {
            const uint4 rv = *(const uint4*)((const bfraw*)p.residual + c_off + (long long)m * p.ldr + col);
            float a[8], b[8];
            unpack8(v, a);
            unpack8(rv, b);
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += b[q];
            v = pack8(a);
          }
          *(uint4*)(Cout + (long long)m * p.ldc + col) = v;
        }
      }
      __syncthreads();
    }
  }
  if (p.colstats) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      s1[j] += __shfl_xor(s1[j], 16);
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 16);
      s2[j] += __shfl_xor(s2[j], 32);
      if (fq == 0) {
        statbuf[(wave * BN + j * 16 + fr) * 2 + 0] = s1[j];
        statbuf[(wave * BN + j * 16 + fr) * 2 + 1] = s2[j];
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      const int n = nb * BN + c;
      if (n < p.ldstat) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += statbuf[(w * BN + c) * 2 + 0];
          b += statbuf[(w * BN + c) * 2 + 1];
        }
        p.colstats[((long long)mb * 2 + 0) * p.ldstat + n] = a;
        p.colstats[((long long)mb * 2 + 1) * p.ldstat + n] = b;
      }
    }
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

template <int WN>
int launch_wn(const pp_igemm_desc& d, hipStream_t s) {
  const int nblk_n = (d.N + 16 * WN - 1) / (16 * WN);
  const long long nblk_m = ((long long)d.M + BM - 1) / BM;
  const long long gx = nblk_m * nblk_n;
  if (gx <= 0 || gx > 0x7fffffffLL) { pp_set_error("pp_igemm: grid too large"); return PP_ERR_INVALID; }
  dim3 grid((unsigned)gx, 1, (unsigned)d.nbatch), block(256);
  switch (d.g.mode) {
    case PP_DENSE: hipLaunchKernelGGL((igemm_kernel<WN, PP_DENSE>), grid, block, 0, s, d, nblk_n); break;
    case PP_CONV_FWD: hipLaunchKernelGGL((igemm_kernel<WN, PP_CONV_FWD>), grid, block, 0, s, d, nblk_n); break;
    case PP_CONV_DGRAD: hipLaunchKernelGGL((igemm_kernel<WN, PP_CONV_DGRAD>), grid, block, 0, s, d, nblk_n); break;
    default: pp_set_error("pp_igemm: bad gather mode %d", d.g.mode); return PP_ERR_INVALID;
  }
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
  PP_CHECK_ARG(g.kt > 0 && g.kh > 0 && g.kw > 0 && g.kt * g.kh * g.kw <= 128 && g.kt < 256 && g.kh < 256 &&
                   g.kw < 256, "%s: taps %dx%dx%d unsupported (<=128 total)", who, g.kt, g.kh, g.kw);
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
  if (d.colstats) PP_CHECK_ARG(!d.bias && d.nbatch == 1 && d.ldstat >= d.N, "pp_igemm: colstats needs no bias, nbatch 1");
  PP_CHECK_ARG(((uintptr_t)d.A & 15) == 0 && ((uintptr_t)d.Bt & 15) == 0, "pp_igemm: operands must be 16-byte aligned");
  const int rc = pp_validate_gather(d.g, d.K, "pp_igemm");
  if (rc != PP_OK) return rc;
  if (d.g.mode != PP_DENSE) {
    const long long rows = (long long)d.g.Rt * d.g.Rh * d.g.Rw;
    PP_CHECK_ARG(d.M % rows == 0, "pp_igemm: M=%d is not a multiple of Rt*Rh*Rw=%lld", d.M, rows);
  }
  if (d.g.mode != PP_DENSE) {
    const long long src = (long long)(d.M / ((long long)d.g.Rt * d.g.Rh * d.g.Rw)) * d.g.Gt * d.g.Gh * d.g.Gw * d.g.cstride;
    PP_CHECK_ARG(src < 0x7fffffffLL, "pp_igemm: gathered tensor has %lld elements (>= 2^31)", src);
  } else {
    PP_CHECK_ARG((long long)d.M * d.g.lda < 0x7fffffffLL, "pp_igemm: dense operand >= 2^31 elements");
  }
  hipStream_t s = (hipStream_t)stream;
  const int n16 = (d.N + 15) / 16;
  switch (pick_wn(n16)) {
    case 15: return launch_wn<15>(d, s);
    case 9: return launch_wn<9>(d, s);
    case 8: return launch_wn<8>(d, s);
    case 4: return launch_wn<4>(d, s);
    case 3: return launch_wn<3>(d, s);
    default: return launch_wn<2>(d, s);
  }
}
