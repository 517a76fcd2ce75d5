// Fused self-attention core of the wav2vec2 encoder layers (torchaudio SelfAttention, T <= 320 frames, 64-wide heads),
// forward and backward -- gfx950.  T <= 128: one workgroup per (clip, head), as described below.  128 < T <= 320 (the
// 229 frames of 4.6-s clips, BASELINE configs[4]; the 316 frames of the reference's own 2.3-s clips at 44.1 kHz fed
// unresampled): the forward runs one workgroup per (clip, head, 64 query rows) against all keys (256 or 320
// key rows in LDS); the backward walks the NB x NB blocks of 128 queries x 128 keys inside one workgroup (NB = 2, 3), with the
// softmax normalisation taken from the forward's log-sum-exp (lse) and D = rowsum(dO o O) instead of in-kernel row statistics.
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

constexpr int DH = 64;                  // head width
constexpr int QS = 160;                 // row stride of the [rows][64] tiles: 32 x 5 bytes -> conflict-free transposing reads
constexpr int BT = 128;                 // block edge of the backward pass (queries and keys)
constexpr int PSB = 288;                // row stride of its [128][128] tiles: 32 x 9
template <int NT> constexpr int ps_of() { return NT * 32 + 32; }   // [rows][16 NT] probability tile: 32 x odd bytes per row

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
  const h16raw* octx;    // backward: the forward's output [B*T][Hn*64]
  h16raw* ctx;           // forward out [B*T][Hn*64]
  h16raw* dqkv;          // backward out [B*T][3*Hn*64]
  float* lse;            // [B*Hn][T]: log-sum-exp of the scaled scores of every query row (forward out, backward in)
  int T, Tp, Hn, nqb;
  float scale;          // applied to the scores
  uint32_t drop_thr;    // p * 65536 (0 = no dropout)
  float drop_scale;     // 1 / (1 - p)
  uint32_t seed;
};

// rows row0 .. row0 + ROWS - 1 of one [T][64] head slice (row stride ld elements) into a zero-padded [ROWS][64] LDS tile
template <int NTH, int ROWS>
__device__ __forceinline__ void load_tile(unsigned char* tile, const h16raw* src, int ld, int row0, int T, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS / (NTH / 8); ++i) {
    const int row = (tid >> 3) + (NTH / 8) * i, ch = tid & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < T) v = *(const uint4*)(src + (long long)(row0 + row) * ld + ch * 8);
    *(uint4*)(tile + row * QS + ch * 16) = v;
  }
}

// scores of this wave's 16 MT query rows (tile rows, global rows q0 + ...) against all 16 NT keys, softmax, dropout:
// P (fp32, registers), Pd (16-bit, LDS) and the rows' log-sum-exp
template <int MT, int NT>
__device__ __forceinline__ void scores_softmax(const AttnArgs& p, const unsigned char* Qs, const unsigned char* Ks,
                                               unsigned char* Ps, int bh, int q0, int wave, int lane, f32x4 (&P)[MT][NT]) {
  constexpr int PS = ps_of<NT>();
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) P[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h16x8 aq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aq[mt] = frag_rowmajor(Qs, QS, wave * (16 * MT) + mt * 16, ks * 32, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
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
      const int tl = wave * (16 * MT) + mt * 16 + fq * 4 + r, t = q0 + tl;
      float mx = -3.0e38f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = (fr + 16 * nt) < p.T ? P[mt][nt][r] * p.scale : -3.0e38f;
        P[mt][nt][r] = v;
        mx = fmaxf(mx, v);
      }
      mx = group16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float e = (fr + 16 * nt) < p.T ? __expf(P[mt][nt][r] - mx) : 0.f;
        P[mt][nt][r] = e;
        sum += e;
      }
      sum = group16_sum(sum);
      const float inv = 1.f / sum;
      if (fr == 0 && t < p.T) p.lse[(long long)bh * p.T + t] = mx + __logf(sum);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int j = fr + 16 * nt;
        const float pr = P[mt][nt][r] * inv;
        P[mt][nt][r] = pr;
        float pd = h2f(f2h(pr));       // the unfused path rounds P to 16 bits before the dropout
        if (p.drop_thr) pd = keep_elem(p.seed, ((long long)bh * p.T + t) * p.Tp + j, p.drop_thr) ? pd * p.drop_scale : 0.f;
        *(h16raw*)(Ps + tl * PS + j * 2) = (t < p.T) ? f2h(pd) : (h16raw)0;
      }
    }
}

// out[(row)][c0 + 4 fq .. + 3] (16-bit, row stride ld) from a transposed accumulator tile: lane column = row (fr), rows = channels
__device__ __forceinline__ void store_t(h16raw* out, long long ld, int row, int T, int ch, const f32x4& a) {
  if (row < T) {
    uint2 v;
    v.x = (uint32_t)f2h(a[0]) | ((uint32_t)f2h(a[1]) << 16);
    v.y = (uint32_t)f2h(a[2]) | ((uint32_t)f2h(a[3]) << 16);
    *(uint2*)(out + (long long)row * ld + ch) = v;
  }
}

// Forward: four waves, 16 MT query rows each, against all 16 NT keys.  <2, 8>: T <= 128, one workgroup per (clip, head).
// <1, 16> / <1, 20>: T <= 256 / 320, one workgroup per (clip, head, 64 query rows) -- K and V of the head are loaded by each of
// them (L2); 320 key rows are what fits: (64 + 2 x 320) x 160 B of Q, K, V + 64 x 672 B of probabilities = 152 KB.
template <int MT, int NT>
__global__ __launch_bounds__(256, 1) void attention_fwd_kernel(const AttnArgs p) {
  constexpr int QB = 64 * MT, KR = 16 * NT, PS = ps_of<NT>();
  static_assert((QB + 2 * KR) * QS + QB * PS <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char smem[(QB + 2 * KR) * QS + QB * PS];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + QB * QS;
  unsigned char* Vs = Ks + KR * QS;
  unsigned char* Ps = Vs + KR * QS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x / p.nqb, q0 = (blockIdx.x % p.nqb) * QB, b = bh / p.Hn, h = bh % p.Hn;
  const int D = p.Hn * DH, D3 = 3 * D;
  const h16raw* base = p.qkv + (long long)b * p.T * D3 + h * DH;
  load_tile<256, QB>(Qs, base, D3, q0, p.T, tid);
  load_tile<256, KR>(Ks, base + D, D3, 0, p.T, tid);
  load_tile<256, KR>(Vs, base + 2 * D, D3, 0, p.T, tid);
  __syncthreads();
  f32x4 P[MT][NT];
  scores_softmax<MT, NT>(p, Qs, Ks, Ps, bh, q0, wave, lane, P);
  __syncthreads();   // (only this wave's rows of Pd are read below; the barrier also orders the 2-byte stores)
  // O^T[d][t] = sum_j V[j][d] Pd[t][j]: first operand indexed by d (V is k-major), second by t (Pd row-major)
  f32x4 O[4][MT];
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) O[nd][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kj = 0; kj < NT / 2; ++kj) {
    h16x8 ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = frag_rowmajor(Ps, PS, wave * (16 * MT) + mt * 16, kj * 32, lane);
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const h16x8 bv = frag_kmajor(Vs, QS, nd * 16, kj * 32, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) O[nd][mt] = PP_MFMA16(bv, ap[mt], O[nd][mt], 0, 0, 0);
    }
  }
  const int fr = lane & 15, fq = lane >> 4;
  h16raw* out = p.ctx + (long long)b * p.T * D + h * DH;
#pragma unroll
  for (int nd = 0; nd < 4; ++nd)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) store_t(out, D, q0 + wave * (16 * MT) + mt * 16 + fr, p.T, nd * 16 + fq * 4, O[nd][mt]);
}

// Backward: eight waves, one workgroup per (clip, head); NB x NB blocks of 128 queries x 128 keys (NB = 1: T <= 128).
// Per block: S = Q K^T and dP = dO V^T for the wave's 16 query rows, P = exp(scale S - lse) (the forward's row
// normalisation), dS = scale P o (dPd - D) with D = rowsum(dO o O); Pd and dS go to LDS, then every wave takes the 16
// KEYS it owns (dV, dK accumulate over the query blocks) and the 16 QUERIES it owns (dQ accumulates over the key blocks).
// Eight waves, one 16-row tile each: two waves per SIMD overlap LDS fragment reads, MFMA chains and the exp / hash work.
constexpr int BNT = 512;
template <int NB>
__global__ __launch_bounds__(BNT, 1) void attention_bwd_kernel(const AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BT * QS + 2 * BT * PSB];
  unsigned char* Qs = smem;
  unsigned char* Ks = smem + BT * QS;
  unsigned char* Vs = smem + 2 * BT * QS;
  unsigned char* Os = smem + 3 * BT * QS;              // dO
  unsigned char* Ps = smem + 4 * BT * QS;              // Pd
  unsigned char* Ss = smem + 4 * BT * QS + BT * PSB;   // dS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / p.Hn, h = bh % p.Hn;
  const int D = p.Hn * DH, D3 = 3 * D;
  const h16raw* base = p.qkv + (long long)b * p.T * D3 + h * DH;
  const h16raw* dO = p.dctx + (long long)b * p.T * D + h * DH;
  const h16raw* Oc = p.octx + (long long)b * p.T * D + h * DH;
  h16raw* dq = p.dqkv + (long long)b * p.T * D3 + h * DH;
  h16raw* dk = dq + D;
  h16raw* dv = dq + 2 * D;
  f32x4 accK[NB][4], accV[NB][4];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) accK[kb][nd] = accV[kb][nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int qb = 0; qb < NB; ++qb) {       // (rolled: only the key-block loop indexes register arrays)
    if (qb * BT >= p.T) break;
    if (qb > 0) __syncthreads();           // every wave is done with the previous query block's tiles
    load_tile<BNT, BT>(Qs, base, D3, qb * BT, p.T, tid);
    load_tile<BNT, BT>(Os, dO, D, qb * BT, p.T, tid);
    if (NB == 1) {
      load_tile<BNT, BT>(Ks, base + D, D3, 0, p.T, tid);
      load_tile<BNT, BT>(Vs, base + 2 * D, D3, 0, p.T, tid);
    }
    // per-row constants of this wave's 16 query rows (rows 4 fq + r of the accumulator layout)
    float lse_r[4], D_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = qb * BT + wave * 16 + fq * 4 + r;
      float part = 0.f;
      lse_r[r] = 0.f;
      if (t < p.T) {
        lse_r[r] = p.lse[(long long)bh * p.T + t];
        const uint2 a = *(const uint2*)(dO + (long long)t * D + fr * 4), o = *(const uint2*)(Oc + (long long)t * D + fr * 4);
        part = h2f((h16raw)(a.x & 0xffff)) * h2f((h16raw)(o.x & 0xffff)) + h2f((h16raw)(a.x >> 16)) * h2f((h16raw)(o.x >> 16)) +
               h2f((h16raw)(a.y & 0xffff)) * h2f((h16raw)(o.y & 0xffff)) + h2f((h16raw)(a.y >> 16)) * h2f((h16raw)(o.y >> 16));
      }
      D_r[r] = group16_sum(part);
    }
    f32x4 accQ[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) accQ[nd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = 0; kb < NB; ++kb) {     // (rolled; the dK / dV tiles of key block 0 / 1 are picked by a uniform branch)
      if (kb * BT >= p.T) break;
      if (NB > 1) {
        if (kb > 0) __syncthreads();       // ... and with the previous key block's K, V, Pd, dS
        load_tile<BNT, BT>(Ks, base + D, D3, kb * BT, p.T, tid);
        load_tile<BNT, BT>(Vs, base + 2 * D, D3, kb * BT, p.T, tid);
      }
      __syncthreads();
      // scores and dP in two halves of 64 keys: 32 accumulator registers live at a time instead of 64 (with the sixteen
      // dK / dV tiles of two key blocks resident the whole block body otherwise spills)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 P[4], dP[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) P[nt] = dP[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const h16x8 aq = frag_rowmajor(Qs, QS, wave * 16, ks * 32, lane);
          const h16x8 ao = frag_rowmajor(Os, QS, wave * 16, ks * 32, lane);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            P[nt] = PP_MFMA16(aq, frag_rowmajor(Ks, QS, (half * 4 + nt) * 16, ks * 32, lane), P[nt], 0, 0, 0);
            dP[nt] = PP_MFMA16(ao, frag_rowmajor(Vs, QS, (half * 4 + nt) * 16, ks * 32, lane), dP[nt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tl = wave * 16 + fq * 4 + r, t = qb * BT + tl;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const int jl = fr + 16 * (half * 4 + nt), j = kb * BT + jl;
            const bool valid = t < p.T && j < p.T;
            const float pb = valid ? h2f(f2h(__expf(P[nt][r] * p.scale - lse_r[r]))) : 0.f;   // (the unfused path keeps P in 16 bits)
            float d = dP[nt][r], pd = pb;
            if (p.drop_thr) {
              const bool keep = keep_elem(p.seed, ((long long)bh * p.T + t) * p.Tp + j, p.drop_thr);
              d = keep ? d * p.drop_scale : 0.f;
              pd = keep ? pb * p.drop_scale : 0.f;
            }
            *(h16raw*)(Ps + tl * PSB + jl * 2) = f2h(pd);
            *(h16raw*)(Ss + tl * PSB + jl * 2) = f2h(p.scale * pb * (d - D_r[r]));
          }
        }
      }
      __syncthreads();   // Pd and dS of every row of the block are in LDS
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        // dV^T[d][j] += sum_t dO[t][d] Pd[t][j], dK^T[d][j] += sum_t Q[t][d] dS[t][j]   (all k-major in t; this wave's 16 keys j)
        const h16x8 bp = frag_kmajor(Ps, PSB, wave * 16, kt * 32, lane);
        const h16x8 bs = frag_kmajor(Ss, PSB, wave * 16, kt * 32, lane);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) {
          const h16x8 ao = frag_kmajor(Os, QS, nd * 16, kt * 32, lane), aq = frag_kmajor(Qs, QS, nd * 16, kt * 32, lane);
          if (NB == 1 || kb == 0) {
            accV[0][nd] = PP_MFMA16(ao, bp, accV[0][nd], 0, 0, 0);
            accK[0][nd] = PP_MFMA16(aq, bs, accK[0][nd], 0, 0, 0);
          } else if (NB == 2 || kb == 1) {
            accV[1 < NB ? 1 : 0][nd] = PP_MFMA16(ao, bp, accV[1 < NB ? 1 : 0][nd], 0, 0, 0);
            accK[1 < NB ? 1 : 0][nd] = PP_MFMA16(aq, bs, accK[1 < NB ? 1 : 0][nd], 0, 0, 0);
          } else {
            accV[NB - 1][nd] = PP_MFMA16(ao, bp, accV[NB - 1][nd], 0, 0, 0);
            accK[NB - 1][nd] = PP_MFMA16(aq, bs, accK[NB - 1][nd], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int kj = 0; kj < 4; ++kj) {
        // dQ^T[d][t] += sum_j K[j][d] dS[t][j]    (K k-major in j, dS row-major in j; this wave's 16 queries t)
        const h16x8 bs = frag_rowmajor(Ss, PSB, wave * 16, kj * 32, lane);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) accQ[nd] = PP_MFMA16(frag_kmajor(Ks, QS, nd * 16, kj * 32, lane), bs, accQ[nd], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) store_t(dq, D3, qb * BT + wave * 16 + fr, p.T, nd * 16 + fq * 4, accQ[nd]);
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      store_t(dv, D3, kb * BT + wave * 16 + fr, p.T, nd * 16 + fq * 4, accV[kb][nd]);
      store_t(dk, D3, kb * BT + wave * 16 + fr, p.T, nd * 16 + fq * 4, accK[kb][nd]);
    }
}

int check(const char* who, const void* qkv, int B, int T, int Hn, float p) {
  PP_CHECK_ARG(qkv && B > 0 && T > 0 && T <= 320 && Hn > 0 && Hn <= 64, "%s: B=%d T=%d heads=%d unsupported (T <= 320)", who, B, T, Hn);
  PP_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout p=%f", who, (double)p);
  return PP_OK;
}

AttnArgs make_args(const void* qkv, int T, int Hn, float scale, float p, unsigned seed) {
  AttnArgs a;
  a.qkv = (const h16raw*)qkv; a.dctx = nullptr; a.octx = nullptr; a.ctx = nullptr; a.dqkv = nullptr; a.lse = nullptr;
  a.T = T; a.Tp = (T + 15) & ~15; a.Hn = Hn; a.nqb = 1; a.scale = scale;
  a.drop_thr = (uint32_t)(p * 65536.f + 0.5f);
  a.drop_scale = 1.f / (1.f - p);
  a.seed = seed;
  return a;
}

}  // namespace

extern "C" int pp_attention_fwd(const void* qkv, int B, int T, int heads, float scale, float drop_p, unsigned seed, void* ctx,
                                float* lse, pp_stream_t s) {
  if (int rc = check("pp_attention_fwd", qkv, B, T, heads, drop_p)) return rc;
  PP_CHECK_ARG(ctx != nullptr && lse != nullptr, "pp_attention_fwd: null output");
  AttnArgs a = make_args(qkv, T, heads, scale, drop_p, seed);
  a.ctx = (h16raw*)ctx;
  a.lse = lse;
  if (T <= 128) {
    hipLaunchKernelGGL((attention_fwd_kernel<2, 8>), dim3(B * heads), dim3(256), 0, (hipStream_t)s, a);
  } else {
    a.nqb = (T + 63) / 64;
    if (T <= 256) hipLaunchKernelGGL((attention_fwd_kernel<1, 16>), dim3(B * heads * a.nqb), dim3(256), 0, (hipStream_t)s, a);
    else hipLaunchKernelGGL((attention_fwd_kernel<1, 20>), dim3(B * heads * a.nqb), dim3(256), 0, (hipStream_t)s, a);
  }
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_attention_bwd(const void* qkv, const void* ctx, const float* lse, const void* dctx, int B, int T, int heads,
                                float scale, float drop_p, unsigned seed, void* dqkv, pp_stream_t s) {
  if (int rc = check("pp_attention_bwd", qkv, B, T, heads, drop_p)) return rc;
  PP_CHECK_ARG(ctx && lse && dctx && dqkv, "pp_attention_bwd: null operand");
  AttnArgs a = make_args(qkv, T, heads, scale, drop_p, seed);
  a.dctx = (const h16raw*)dctx;
  a.octx = (const h16raw*)ctx;
  a.lse = (float*)lse;
  a.dqkv = (h16raw*)dqkv;
  if (T <= 128) hipLaunchKernelGGL((attention_bwd_kernel<1>), dim3(B * heads), dim3(BNT), 0, (hipStream_t)s, a);
  else if (T <= 256) hipLaunchKernelGGL((attention_bwd_kernel<2>), dim3(B * heads), dim3(BNT), 0, (hipStream_t)s, a);
  else hipLaunchKernelGGL((attention_bwd_kernel<3>), dim3(B * heads), dim3(BNT), 0, (hipStream_t)s, a);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
