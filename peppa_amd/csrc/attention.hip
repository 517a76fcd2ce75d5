// Fused self-attention core of the wav2vec2 encoder layers (torchaudio SelfAttention, T <= 128 frames, 64-wide heads),
// forward and backward, one workgroup per (clip, head) -- gfx950.
//
//   forward : S = scale * Q K^T -> P = softmax(S) -> Pd = dropout(P) -> O = Pd V
//   backward: recomputes S, P, Pd from Q, K (nothing is saved but the dropout seed), then
//             dV = Pd^T dO, dP = dO V^T, dS = scale * P o (dPd - rowsum(dPd o P)), dQ = dS K, dK = dS^T Q
//
// The unfused path needs 5 + 12 launches per layer (batched GEMMs on 114 x 114 tiles, six transposes, softmax, dropout)
// and round-trips S / P / dP through HBM; here Q, K, V (and dO) of one head sit in LDS, the 128 x 128 score tile lives in
// MFMA accumulators, the softmax runs in the accumulator layout (a row is spread over 16 lanes x 8 tiles: in-lane
// reduction + xor-shuffles inside the 16-lane group), and operands whose reduce index is the slow memory axis are
// fetched with the transposing LDS read.  Outputs are produced transposed (D = W X^T) so that a lane owns four
// consecutive head channels of one frame: 8-byte stores, a full 128-byte line per frame and head.
// Dropout uses the counter-based mask of elementwise.hip with the element index of the unfused P tensor [B*H][T][Tp],
// so both paths drop the same probabilities.  Replaces torchaudio.models.wav2vec2.components.SelfAttention's
// softmax(QK^T)V and its autograd (pig/models.py:66-109 call site).
#include "common.h"

namespace {

constexpr int TT = 128;                 // frames per tile (T <= 128)
constexpr int DH = 64;                  // head width
constexpr int QS = 160;                 // row stride of the [128][64] tiles: 32 x 5 bytes -> conflict-free transposing reads
constexpr int PS = 288;                 // row stride of the [128][128] tiles: 32 x 9
constexpr int QK_BYTES = TT * QS;
constexpr int P_BYTES = TT * PS;


__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
// keep-bit of element e of the dropout stream `seed` (identical to elementwise.hip's keep8: one hash per element pair)
__device__ __forceinline__ bool keep_elem(uint32_t seed, long long e, uint32_t thr) {
  const long long w = e >> 1;
  const uint32_t h = mix32((uint32_t)w * 0x9E3779B9u + seed + (uint32_t)(w >> 32) * 0x7F4A7C15u);
  return ((e & 1) ? (h >> 16) : (h & 0xffffu)) >= thr;
}

// fragment of an operand stored [row][k] (k contiguous): lane l -> row (l & 15), k = k0 + 8 (l >> 4) .. + 7
__device__ __forceinline__ h16x8 frag_rowmajor(const unsigned char* tile, int stride, int row0, int k0, int lane) {
  return *(const h16x8*)(tile + (row0 + (lane & 15)) * stride + (k0 + 8 * (lane >> 4)) * 2);
}
// fragment of an operand stored [k][col] (k is the slow axis): same register layout as above -- lane l receives column
// (l & 15) for k = k0 + 8 (l >> 4) .. + 7 -- through two transposing reads (lane 4q+p of a 16-lane group supplies the
// address of k-row q, columns 4p .. 4p+3)
__device__ __forceinline__ h16x8 frag_kmajor(const unsigned char* tile, int stride, int col0, int k0, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const unsigned char* a0 = tile + (k0 + 8 * g + (li >> 2)) * stride + (col0 + 4 * (li & 3)) * 2;
  const h16x4 lo = ds_read_tr16(a0);
  const h16x4 hi = ds_read_tr16(a0 + 4 * stride);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// reductions over a 16-lane row on the VALU (DPP quad swaps, half-row mirror, row mirror; common.h)
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  return v;
}

struct AttnArgs {
  const h16raw* qkv;     // [B*T][3*Hn*64]: q | k | v
  const h16raw* dctx;    // backward: [B*T][Hn*64]
  h16raw* ctx;           // forward out [B*T][Hn*64]
  h16raw* dqkv;          // backward out [B*T][3*Hn*64]
  int T, Tp, Hn;
  float scale;          // applied to the scores
  uint32_t drop_thr;    // p * 65536 (0 = no dropout)
  float drop_scale;     // 1 / (1 - p)
  uint32_t seed;
};

// load one [T][64] head slice (row stride ld elements) into a zero-padded [128][64] LDS tile
template <int NTH>
__device__ __forceinline__ void load_tile(unsigned char* tile, const h16raw* src, int ld, int T, int tid) {
#pragma unroll
  for (int i = 0; i < TT / (NTH / 8); ++i) {
    const int row = (tid >> 3) + (NTH / 8) * i, ch = tid & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < T) v = *(const uint4*)(src + (long long)row * ld + ch * 8);
    *(uint4*)(tile + row * QS + ch * 16) = v;
  }
}

// scores of this wave's 16 MT query rows against all 128 keys, softmax, dropout: P (fp32, registers) and Pd (bf16, LDS)
template <int MT>
__device__ __forceinline__ void scores_softmax(const AttnArgs& p, const unsigned char* Qs, const unsigned char* Ks,
                                               unsigned char* Ps, int bh, int wave, int lane, f32x4 (&P)[MT][8]) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) P[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h16x8 aq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aq[mt] = frag_rowmajor(Qs, QS, wave * (16 * MT) + mt * 16, ks * 32, lane);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const h16x8 bk = frag_rowmajor(Ks, QS, nt * 16, ks * 32, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) P[mt][nt] = PP_MFMA16(aq[mt], bk, P[mt][nt], 0, 0, 0);
    }
  }
  // accumulator layout: column j = fr + 16 nt, rows t = 16 MT wave + 16 mt + 4 fq + r
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = wave * (16 * MT) + mt * 16 + fq * 4 + r;
      float mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float v = (fr + 16 * nt) < p.T ? P[mt][nt][r] * p.scale : -3.0e38f;
        P[mt][nt][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = group16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float e = (fr + 16 * nt) < p.T ? __expf(P[mt][nt][r] - mx) : 0.f;
        P[mt][nt][r] = e;
        sum += e;
      }
      sum = group16_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int j = fr + 16 * nt;
        const float pr = P[mt][nt][r] * inv;
        P[mt][nt][r] = pr;
        float pd = h2f(f2h(pr));       // the unfused path rounds P to bf16 before the dropout
        if (p.drop_thr) pd = keep_elem(p.seed, ((long long)bh * p.T + t) * p.Tp + j, p.drop_thr) ? pd * p.drop_scale : 0.f;
        *(h16raw*)(Ps + t * PS + j * 2) = (t < p.T) ? f2h(pd) : (h16raw)0;
      }
    }
}

// out[(row)][c0 + 4 fq .. + 3] (bf16, row stride ld) from a transposed accumulator tile: lane column = row (fr), rows = channels
__device__ __forceinline__ void store_t(h16raw* out, long long ld, int row, int T, int ch, const f32x4& a) {
  if (row < T) {
    uint2 v;
    v.x = (uint32_t)f2h(a[0]) | ((uint32_t)f2h(a[1]) << 16);
    v.y = (uint32_t)f2h(a[2]) | ((uint32_t)f2h(a[3]) << 16);
    *(uint2*)(out + (long long)row * ld + ch) = v;
  }
}

__global__ __launch_bounds__(256, 1) void attention_fwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[3 * QK_BYTES + P_BYTES];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + QK_BYTES;
  unsigned char* Vs = smem + 2 * QK_BYTES;
  unsigned char* Ps = smem + 3 * QK_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / p.Hn, h = bh % p.Hn;
  const int D = p.Hn * DH, D3 = 3 * D;
  const h16raw* base = p.qkv + (long long)b * p.T * D3 + h * DH;
  load_tile<256>(Qs, base, D3, p.T, tid);
  load_tile<256>(Ks, base + D, D3, p.T, tid);
  load_tile<256>(Vs, base + 2 * D, D3, p.T, tid);
  __syncthreads();
  f32x4 P[2][8];
  scores_softmax<2>(p, Qs, Ks, Ps, bh, wave, lane, P);
  __syncthreads();   // (only this wave's rows of Pd are read below; the barrier also orders the 2-byte stores)
  // O^T[d][t] = sum_j V[j][d] Pd[t][j]: first operand indexed by d (V is k-major), second by t (Pd row-major)
  f32x4 O[4][2];
#pragma unroll
  for (int nd = 0; nd < 4; ++nd) O[nd][0] = O[nd][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kj = 0; kj < 4; ++kj) {
    h16x8 ap[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ap[mt] = frag_rowmajor(Ps, PS, wave * 32 + mt * 16, kj * 32, lane);
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const h16x8 bv = frag_kmajor(Vs, QS, nd * 16, kj * 32, lane);
      O[nd][0] = PP_MFMA16(bv, ap[0], O[nd][0], 0, 0, 0);
      O[nd][1] = PP_MFMA16(bv, ap[1], O[nd][1], 0, 0, 0);
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  h16raw* out = p.ctx + (long long)b * p.T * D + h * DH;
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) store_t(out, D, wave * 32 + mt * 16 + fr, p.T, nd * 16 + fq * 4, O[nd][mt]);
}

// Eight waves, one 16-row tile of queries each (BMT = 1): with four waves of two tiles the kernel ran one wave per SIMD and
// every phase (LDS fragment reads -> MFMA chain -> exp / reductions) was exposed; two waves per SIMD overlap them.
constexpr int BMT = 1, BNT = 64 * (8 / BMT);
__global__ __launch_bounds__(BNT, 1) void attention_bwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * QK_BYTES + 2 * P_BYTES];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + QK_BYTES;
  unsigned char* Vs = smem + 2 * QK_BYTES;
  unsigned char* Os = smem + 3 * QK_BYTES;          // dO
  unsigned char* Ps = smem + 4 * QK_BYTES;          // Pd
  unsigned char* Ss = smem + 4 * QK_BYTES + P_BYTES;   // dS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / p.Hn, h = bh % p.Hn;
  const int D = p.Hn * DH, D3 = 3 * D;
  const h16raw* base = p.qkv + (long long)b * p.T * D3 + h * DH;
  load_tile<BNT>(Qs, base, D3, p.T, tid);
  load_tile<BNT>(Ks, base + D, D3, p.T, tid);
  load_tile<BNT>(Vs, base + 2 * D, D3, p.T, tid);
  load_tile<BNT>(Os, p.dctx + (long long)b * p.T * D + h * DH, D, p.T, tid);
  __syncthreads();
  f32x4 P[BMT][8];
  scores_softmax<BMT>(p, Qs, Ks, Ps, bh, wave, lane, P);

  // dP[t][j] = sum_d dO[t][d] V[j][d]  (both row-major in d); through the dropout; dS = scale * P o (dPd - rowsum(dPd o P))
  f32x4 dP[BMT][8];
#pragma unroll
  for (int mt = 0; mt < BMT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) dP[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h16x8 ao[BMT];
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) ao[mt] = frag_rowmajor(Os, QS, wave * (16 * BMT) + mt * 16, ks * 32, lane);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const h16x8 bv = frag_rowmajor(Vs, QS, nt * 16, ks * 32, lane);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) dP[mt][nt] = PP_MFMA16(ao[mt], bv, dP[mt][nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < BMT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = wave * (16 * BMT) + mt * 16 + fq * 4 + r;
      float dot = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int j = fr + 16 * nt;
        float d = dP[mt][nt][r];
        if (p.drop_thr) d = keep_elem(p.seed, ((long long)bh * p.T + t) * p.Tp + j, p.drop_thr) ? d * p.drop_scale : 0.f;
        dP[mt][nt][r] = d;
        dot += d * h2f(f2h(P[mt][nt][r]));   // (the unfused path keeps P in bf16)
      }
      dot = group16_sum(dot);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int j = fr + 16 * nt;
        const float ds = p.scale * h2f(f2h(P[mt][nt][r])) * (dP[mt][nt][r] - dot);
        *(h16raw*)(Ss + t * PS + j * 2) = (t < p.T && j < p.T) ? f2h(ds) : (h16raw)0;
      }
    }
  __syncthreads();   // Pd and dS of every row are in LDS

  h16raw* dq = p.dqkv + (long long)b * p.T * D3 + h * DH;
  h16raw* dk = dq + D;
  h16raw* dv = dq + 2 * D;
  f32x4 acc[4][BMT];
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int nd = 0; nd < 4; ++nd)
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) acc[nd][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // dV^T[d][j] = sum_t dO[t][d] Pd[t][j]   (both k-major in t)
  zero();
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    h16x8 bp[BMT];
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) bp[mt] = frag_kmajor(Ps, PS, wave * (16 * BMT) + mt * 16, kt * 32, lane);
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const h16x8 ao = frag_kmajor(Os, QS, nd * 16, kt * 32, lane);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) acc[nd][mt] = PP_MFMA16(ao, bp[mt], acc[nd][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) store_t(dv, D3, wave * (16 * BMT) + mt * 16 + fr, p.T, nd * 16 + fq * 4, acc[nd][mt]);
  // dQ^T[d][t] = sum_j K[j][d] dS[t][j]    (K k-major in j, dS row-major in j)
  zero();
#pragma unroll
  for (int kj = 0; kj < 4; ++kj) {
    h16x8 bs[BMT];
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) bs[mt] = frag_rowmajor(Ss, PS, wave * (16 * BMT) + mt * 16, kj * 32, lane);
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const h16x8 ak = frag_kmajor(Ks, QS, nd * 16, kj * 32, lane);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) acc[nd][mt] = PP_MFMA16(ak, bs[mt], acc[nd][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) store_t(dq, D3, wave * (16 * BMT) + mt * 16 + fr, p.T, nd * 16 + fq * 4, acc[nd][mt]);
  // dK^T[d][j] = sum_t Q[t][d] dS[t][j]    (both k-major in t)
  zero();
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    h16x8 bs[BMT];
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) bs[mt] = frag_kmajor(Ss, PS, wave * (16 * BMT) + mt * 16, kt * 32, lane);
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const h16x8 aq = frag_kmajor(Qs, QS, nd * 16, kt * 32, lane);
#pragma unroll
      for (int mt = 0; mt < BMT; ++mt) acc[nd][mt] = PP_MFMA16(aq, bs[mt], acc[nd][mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < BMT; ++mt) store_t(dk, D3, wave * (16 * BMT) + mt * 16 + fr, p.T, nd * 16 + fq * 4, acc[nd][mt]);
}

int check(const char* who, const void* qkv, int B, int T, int Hn, float p) {
  PP_CHECK_ARG(qkv && B > 0 && T > 0 && T <= TT && Hn > 0 && Hn <= 64, "%s: B=%d T=%d heads=%d unsupported (T <= 128)", who, B, T, Hn);
  PP_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout p=%f", who, (double)p);
  return PP_OK;
}

AttnArgs make_args(const void* qkv, int T, int Hn, float scale, float p, unsigned seed) {
  AttnArgs a;
  a.qkv = (const h16raw*)qkv; a.dctx = nullptr; a.ctx = nullptr; a.dqkv = nullptr;
  a.T = T; a.Tp = (T + 15) & ~15; a.Hn = Hn; a.scale = scale;
  a.drop_thr = (uint32_t)(p * 65536.f + 0.5f);
  a.drop_scale = 1.f / (1.f - p);
  a.seed = seed;
  return a;
}

}  // namespace

extern "C" int pp_attention_fwd(const void* qkv, int B, int T, int heads, float scale, float drop_p, unsigned seed, void* ctx,
                                pp_stream_t s) {
  if (int rc = check("pp_attention_fwd", qkv, B, T, heads, drop_p)) return rc;
  PP_CHECK_ARG(ctx != nullptr, "pp_attention_fwd: null output");
  AttnArgs a = make_args(qkv, T, heads, scale, drop_p, seed);
  a.ctx = (h16raw*)ctx;
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * heads), dim3(256), 0, (hipStream_t)s, a);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_attention_bwd(const void* qkv, const void* dctx, int B, int T, int heads, float scale, float drop_p,
                                unsigned seed, void* dqkv, pp_stream_t s) {
  if (int rc = check("pp_attention_bwd", qkv, B, T, heads, drop_p)) return rc;
  PP_CHECK_ARG(dctx && dqkv, "pp_attention_bwd: null operand");
  AttnArgs a = make_args(qkv, T, heads, scale, drop_p, seed);
  a.dctx = (const h16raw*)dctx;
  a.dqkv = (h16raw*)dqkv;
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * heads), dim3(BNT), 0, (hipStream_t)s, a);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
