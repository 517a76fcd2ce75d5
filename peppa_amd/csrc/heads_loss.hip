// fp32 tail of the step (gfx950): pooling heads, projection + L2-normalise, and the N x N
// cosine-similarity triplet loss.  The contractions run on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 from LDS-staged tiles; reductions are wave-64 shuffles.
//   pig/models.py:30-43  Attention      pig/models.py:213-221 VideoAttention
//   pig/models.py:106-109 / 148-150     project + F.normalize
//   pig/loss.py:33-55    TripletLoss = contrastive(cosine_matrix(V, A))
#include "common.h"

extern int pp_opt_deterministic;

namespace {

// C(m,n) (+)= act(sum_k A(m,k) B(k,n) + bias[n]); arbitrary strides (transposes by stride).
struct SG {
  const float* A; long long sa_m, sa_k;
  const float* B; long long sb_k, sb_n;
  float* C; long long sc_m, sc_n;
  int M, N, K;
  const float* bias;
  const float* aux;   // act==2: C = acc * (1 - aux^2), aux indexed like C
  int act;            // 0 none, 1 tanh, 2 tanh-grad
  int beta;           // 1: C += result (non-atomic)
  int ksplit;         // >1: atomicAdd partial results into zeroed C
};

// one 32 x 32 tile of the product on the exact-fp32 matrix instruction; acc[r] = C(m0 + wm*16 + (lane>>4)*4 + r, n0 + wn*16 + (lane&15))
__device__ __forceinline__ f32x4 sgemm_tile(const SG& p, int m0, int n0, int kb, int ke, float (*As)[17], float (*Bs)[33]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 256 * i;
      {  // A tile 32 x 16
        const int r = e >> 4, c = e & 15;
        const int m = m0 + r, k = k0 + c;
        As[r][c] = (m < p.M && k < ke) ? p.A[m * p.sa_m + k * p.sa_k] : 0.f;
      }
      {  // B tile 16 x 32
        const int r = e >> 5, c = e & 31;
        const int k = k0 + r, n = n0 + c;
        Bs[r][c] = (k < ke && n < p.N) ? p.B[k * p.sb_k + n * p.sb_n] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float a = As[wm * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
      const float b = Bs[kk * 4 + (lane >> 4)][wn * 16 + (lane & 15)];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  return acc;
}

__device__ __forceinline__ void sgemm_body(const SG& p, int bz) {
  __shared__ float As[32][17];
  __shared__ float Bs[16][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  if (m0 >= p.M || n0 >= p.N) return;   // (pair launches: the grid covers the larger problem)
  const int kchunk = ((p.K + p.ksplit - 1) / p.ksplit + 15) & ~15;
  const int kb = bz * kchunk;
  const int ke = min(p.K, kb + kchunk);
  f32x4 acc = sgemm_tile(p, m0, n0, kb, ke, As, Bs);
  const int n = n0 + wn * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
    if (m < p.M && n < p.N) {
      float v = acc[r];
      if (p.bias && bz == 0) v += p.bias[n];
      float* cp = p.C + m * p.sc_m + n * p.sc_n;
      if (p.ksplit > 1) { atomicAdd(cp, v); continue; }
      if (p.act == 1) v = tanhf(v);
      else if (p.act == 2) { const float h = p.aux[m * p.sc_m + n * p.sc_n]; v *= (1.f - h * h); }
      if (p.beta) v += *cp;
      *cp = v;
    }
  }
}

__global__ __launch_bounds__(256) void sgemm_kernel(const SG p) { sgemm_body(p, blockIdx.z); }
// two independent products in one launch (blockIdx.z picks the problem; no split-K)
__global__ __launch_bounds__(256) void sgemm_pair_kernel(const SG p0, const SG p1) { sgemm_body(blockIdx.z ? p1 : p0, 0); }

int sgemm(hipStream_t s, const float* A, long long sa_m, long long sa_k, const float* B, long long sb_k, long long sb_n, float* C,
          long long sc_m, long long sc_n, int M, int N, int K, const float* bias = nullptr, int act = 0, const float* aux = nullptr,
          int beta = 0, int ksplit = 1) {
  SG p{A, sa_m, sa_k, B, sb_k, sb_n, C, sc_m, sc_n, M, N, K, bias, aux, act, beta, ksplit};
  if (ksplit > 1) {
    if (act != 0 || beta != 0 || sc_n != 1 || sc_m != N) { pp_set_error("sgemm: split-K needs dense plain output"); return PP_ERR_INVALID; }
    if (hipMemsetAsync(C, 0, (size_t)M * N * 4, s) != hipSuccess) { pp_set_error("sgemm: memset failed"); return PP_ERR_HIP; }
  }
  hipLaunchKernelGGL(sgemm_kernel, dim3((N + 31) / 32, (M + 31) / 32, ksplit), dim3(256), 0, s, p);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
inline int pick_ksplit(int M, int N, int K) {
  if (pp_opt_deterministic) return 1;      // split-K sums its partial products with atomics
  const int tiles = ((M + 31) / 32) * ((N + 31) / 32);
  int ks = 512 / (tiles > 0 ? tiles : 1);
  const int maxk = (K + 63) / 64;
  if (ks > maxk) ks = maxk;
  return ks < 1 ? 1 : ks;
}

// out[n] = sum_m X[m*ld + n]   (fp32, N <= a few thousand, M arbitrary) -- atomics over row slabs
__global__ void colsum_f32_kernel(const float* __restrict__ X, int M, int N, int ld, int rows_per_blk, float* out) {
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(M, r0 + rows_per_blk);
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += X[(long long)r * ld + n];
    atomicAdd(out + n, s);
  }
}
int colsum_f32(hipStream_t s, const float* X, int M, int N, int ld, float* out) {
  if (hipMemsetAsync(out, 0, (size_t)N * 4, s) != hipSuccess) { pp_set_error("colsum_f32: memset failed"); return PP_ERR_HIP; }
  const int rpb = pp_opt_deterministic ? M : 32;     // deterministic: one workgroup, rows in order
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((M + rpb - 1) / rpb), dim3(256), 0, s, X, M, N, ld, rpb, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- spatial mean over HW (VideoAttention.spatial_avg) -----------------------------------------
__global__ void spatial_mean_fwd_kernel(const h16raw* __restrict__ x, float* __restrict__ out, int HW, int C, int Cp) {
  const long long bt = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += h2f(x[(bt * HW + p) * Cp + c]);
    out[bt * C + c] = s / HW;
  }
}
__global__ void spatial_mean_bwd_kernel(const float* __restrict__ dout, h16raw* __restrict__ dx, int HW, int C, int Cp) {
  const long long bt = blockIdx.x;
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    const h16raw v = f2h(c < C ? dout[bt * C + c] / HW : 0.f);
    for (int p = 0; p < HW; ++p) dx[(bt * HW + p) * Cp + c] = v;
  }
}

// ---- pig/models.py:45-51 AveragePool: nn.AdaptiveAvgPool2d((S, 1)) applied to the 3-D tensor (B, T, F) -------------
// torch reads a 3-D input as (C, H, W): the pool runs over (T, F) -> (S, 1), i.e. the FEATURE axis is averaged away
// and the TIME axis is resampled to S bins [floor(i T / S), ceil((i + 1) T / S)).  Kept as the reference computes it.
__global__ void avgpool_tf_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int F, int S) {
  extern __shared__ float rowsum[];   // [T]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int t = wave; t < T; t += nw) {
    const float* row = x + ((long long)b * T + t) * F;
    float a = 0.f;
    for (int f = lane; f < F; f += 64) a += row[f];
    a = wave_sum(a);
    if (lane == 0) rowsum[t] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    const int t0 = (int)(((long long)i * T) / S), t1 = (int)((((long long)i + 1) * T + S - 1) / S);
    float a = 0.f;
    for (int t = t0; t < t1; ++t) a += rowsum[t];
    out[(long long)b * S + i] = a / (float)((t1 - t0) * F);
  }
}
__global__ void avgpool_tf_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int T, int F, int S) {
  extern __shared__ float coef[];     // [T]: d out / d x[b, t, any f]
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    // bins that contain t: i with floor(i T / S) <= t < ceil((i + 1) T / S)  <=>  i in [ceil((t + 1) S / T) - 1 ... floor(t S / T)] reversed;
    // scanned directly (S is a few hundred)
    float c = 0.f;
    const int ilo = (int)(((long long)t * S) / T) - 1, ihi = (int)((((long long)t + 1) * S + T - 1) / T);
    for (int i = ilo < 0 ? 0 : ilo; i <= ihi && i < S; ++i) {
      const int t0 = (int)(((long long)i * T) / S), t1 = (int)((((long long)i + 1) * T + S - 1) / S);
      if (t >= t0 && t < t1) c += dout[(long long)b * S + i] / (float)((t1 - t0) * F);
    }
    coef[t] = c;
  }
  __syncthreads();
  for (long long i = threadIdx.x; i < (long long)T * F; i += blockDim.x) dx[(long long)b * T * F + i] = coef[i / F];
}

// ---- softmax over time per feature + weighted sum ------------------------------------------------
// e/alpha [B][T][F] in place; pooled[b][f] = sum_t alpha*x
__global__ void timepool_fwd_kernel(float* __restrict__ ea, const float* __restrict__ x, float* __restrict__ pooled, int T, int F) {
  const int b = blockIdx.x;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float* e = ea + (long long)b * T * F + f;
    const float* xp = x + (long long)b * T * F + f;
    float mx = -3.0e38f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, e[(long long)t * F]);
    float sum = 0.f;
    for (int t = 0; t < T; ++t) { const float v = __expf(e[(long long)t * F] - mx); e[(long long)t * F] = v; sum += v; }
    const float inv = 1.f / sum;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) { const float a = e[(long long)t * F] * inv; e[(long long)t * F] = a; acc += a * xp[(long long)t * F]; }
    pooled[(long long)b * F + f] = acc;
  }
}
// de[t][f] = alpha*(dalpha - sum_t alpha*dalpha), dalpha = dpooled*x ; dx = dpooled*alpha
__global__ void timepool_bwd_kernel(const float* __restrict__ alpha, const float* __restrict__ x, const float* __restrict__ dpooled,
                                    float* __restrict__ de, float* __restrict__ dx, int T, int F) {
  const int b = blockIdx.x;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const long long o = (long long)b * T * F + f;
    const float dp = dpooled[(long long)b * F + f];
    float dot = 0.f;
    for (int t = 0; t < T; ++t) dot += alpha[o + (long long)t * F] * dp * x[o + (long long)t * F];
    for (int t = 0; t < T; ++t) {
      const float a = alpha[o + (long long)t * F];
      de[o + (long long)t * F] = a * (dp * x[o + (long long)t * F] - dot);
      dx[o + (long long)t * F] = dp * a;
    }
  }
}

// ---- row L2 normalise (F.normalize: x / max(||x||, eps)) -----------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ nrm,
                                                         int D, float eps) {
  __shared__ float red[4];
  const long long r = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float v = x[r * D + d]; s += v * v; }
  const float n = sqrtf(block_sum<4>(s, red));
  const float dn = fmaxf(n, eps);
  for (int d = threadIdx.x; d < D; d += 256) y[r * D + d] = x[r * D + d] / dn;
  if (nrm && threadIdx.x == 0) nrm[r] = n;
}
// dx = (dy - y*(y.dy)) / max(||x||,eps)   (for ||x|| >= eps; below eps the clamp makes it dy/eps)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ y, float* __restrict__ dx, int D, float eps) {
  __shared__ float red[4];
  const long long r = blockIdx.x;
  float s = 0.f, dot = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) { const float v = x[r * D + d]; s += v * v; dot += y[r * D + d] * dy[r * D + d]; }
  const float n = sqrtf(block_sum<4>(s, red));
  dot = block_sum<4>(dot, red);
  const bool clamped = n < eps;
  const float dn = fmaxf(n, eps);
  for (int d = threadIdx.x; d < D; d += 256)
    dx[r * D + d] = clamped ? dy[r * D + d] / dn : (dy[r * D + d] - y[r * D + d] * dot) / dn;
}

__global__ void copy_f32_kernel(const float* a, float* b, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}

// ---- triplet loss ----------------------------------------------------------------------------------
// ws layout (floats): Vn[N*D] An[N*D] S/G[N*N] diag[N] rowcnt[N] colcnt[N] vnorm[N] anorm[N] dVn[N*D] dAn[N*D]
struct LossWs {
  float *Vn, *An, *G, *diag, *rowc, *colc, *vnorm, *anorm, *dVn, *dAn;
};
__host__ __device__ inline LossWs loss_ws(void* ws, int N, int D) {
  LossWs w;
  float* p = (float*)ws;
  w.Vn = p; p += (size_t)N * D;
  w.An = p; p += (size_t)N * D;
  w.G = p; p += (size_t)N * N;
  w.diag = p; p += N;
  w.rowc = p; p += N;
  w.colc = p; p += N;
  w.vnorm = p; p += N;
  w.anorm = p; p += N;
  w.dVn = p; p += (size_t)N * D;
  w.dAn = p; p += (size_t)N * D;
  return w;
}

__global__ void diag_kernel(const float* __restrict__ S, float* diag, float* rowc, float* colc, float* loss, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { diag[i] = S[(long long)i * N + i]; rowc[i] = 0.f; colc[i] = 0.f; }
  if (i == 0) loss[0] = 0.f;
}
// one block per row i of S: hinge terms, G (in place of S), indicator counts, loss partial
__global__ __launch_bounds__(256) void hinge_kernel(float* __restrict__ SG_, const float* __restrict__ diag, float* rowc,
                                                    float* colc, float* loss, int N, float margin) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float di = diag[i];
  const float inv = 1.f / ((float)N * (float)N);
  float part = 0.f, rc = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float sij = SG_[(long long)i * N + j];
    float g = 0.f;
    if (j != i) {
      const float hc = margin + sij - diag[j];   // column direction: against S_jj
      const float hr = margin + sij - di;        // row direction: against S_ii
      if (hc > 0.f) { part += hc; g += inv; atomicAdd(colc + j, 1.f); }
      if (hr > 0.f) { part += hr; g += inv; rc += 1.f; }
    }
    SG_[(long long)i * N + j] = g;
  }
  part = block_sum<4>(part, red);
  rc = block_sum<4>(rc, red);
  if (threadIdx.x == 0) { atomicAdd(loss, part * inv); rowc[i] = rc; }
}
__global__ void gdiag_kernel(float* G, const float* rowc, const float* colc, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) G[(long long)i * N + i] = -(rowc[i] + colc[i]) / ((float)N * (float)N);
}
// dX = dloss * (dXn' - Xn*(Xn.dXn')) / ||X||   (cosine_matrix normalisation, no clamp: pig/loss.py:51-55), both operands of
// the loss in one launch (blockIdx.y: 0 = V side, 1 = A side).  The stored coefficient matrix G has a zero diagonal; the
// diagonal term gd_r = -(rowc_r + colc_r) / N^2 (every hinge term pulls its diagonal entry down) is applied here:
// dXn'_r = dXn_r + gd_r * Other_r.  dloss == nullptr: scale 1.
struct CosBwd {
  const float *dXn, *Xn, *nrm, *other;
  float* dX;
};
__global__ __launch_bounds__(256) void cosnorm_bwd_kernel(const CosBwd p0, const CosBwd p1, const float* __restrict__ dloss,
                                                          const float* __restrict__ rowc, const float* __restrict__ colc,
                                                          float invn2, int rows0, int rows1, int D) {
  __shared__ float red[4];
  const CosBwd& p = blockIdx.y ? p1 : p0;
  const long long r = blockIdx.x;
  if (r >= (blockIdx.y ? rows1 : rows0)) return;
  const float gd = rowc ? -(rowc[r] + colc[r]) * invn2 : 0.f;
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float g = p.dXn[r * D + d] + (rowc ? gd * p.other[r * D + d] : 0.f);
    dot += p.Xn[r * D + d] * g;
  }
  dot = block_sum<4>(dot, red);
  const float sc = (dloss ? dloss[0] : 1.f) / p.nrm[r];
  for (int d = threadIdx.x; d < D; d += 256) {
    const float g = p.dXn[r * D + d] + (rowc ? gd * p.other[r * D + d] : 0.f);
    p.dX[r * D + d] = sc * (g - p.Xn[r * D + d] * dot);
  }
}

// Launch 1 of the loss: row r of V and A -> unit rows, norms, the diagonal S_rr = Vn_r . An_r; zeroes the accumulators
// of launch 2.  cosine_matrix divides by the norm WITHOUT a clamp (pig/loss.py:51-55): a zero row gives NaN, as there.
__global__ __launch_bounds__(256) void loss_prep_kernel(const float* __restrict__ V, const float* __restrict__ A, const LossWs w,
                                                        float* __restrict__ loss, int D) {
  __shared__ float red[4];
  const long long r = blockIdx.x;
  float sv = 0.f, sa = 0.f, va = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float x = V[r * D + d], y = A[r * D + d];
    sv += x * x; sa += y * y; va += x * y;
  }
  sv = block_sum<4>(sv, red); sa = block_sum<4>(sa, red); va = block_sum<4>(va, red);
  const float nv = sqrtf(sv), na = sqrtf(sa);
  for (int d = threadIdx.x; d < D; d += 256) { w.Vn[r * D + d] = V[r * D + d] / nv; w.An[r * D + d] = A[r * D + d] / na; }
  if (threadIdx.x == 0) {
    w.vnorm[r] = nv; w.anorm[r] = na; w.rowc[r] = 0.f; w.colc[r] = 0.f;
    w.diag[r] = va / (nv * na);
    if (r == 0) loss[0] = 0.f;
  }
}
// The hinges compare S_ij with the diagonal entries S_jj / S_ii, which belong to OTHER tiles of launch 2; they are
// taken from launch 1's dot product Vn_r . An_r instead of torch.diag(S).  Same value up to the summation order
// (~1e-7, inside the 1e-6 tolerance against the live reference's golden vectors; the loss is continuous in it).
// Launch 2: one 32 x 32 tile of S per workgroup; epilogue = both hinges, G (zero diagonal), active counts, loss partial.
__global__ __launch_bounds__(256) void loss_tile_kernel(const LossWs w, const float* __restrict__ diag, float* __restrict__ loss,
                                                        int N, int D, float margin, float* __restrict__ partials) {
  __shared__ float As[32][17];
  __shared__ float Bs[16][33];
  __shared__ float red[4];
  const SG p{w.Vn, D, 1, w.An, 1, D, nullptr, 0, 0, N, N, D, nullptr, nullptr, 0, 0, 1};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const f32x4 acc = sgemm_tile(p, m0, n0, 0, D, As, Bs);
  const float inv = 1.f / ((float)N * (float)N);
  const int n = n0 + wn * 16 + (lane & 15);
  const float dn = n < N ? diag[n] : 0.f;
  float part = 0.f, cc = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + wm * 16 + (lane >> 4) * 4 + r;
    float g = 0.f, rc = 0.f;
    if (m < N && n < N && m != n) {
      const float hc = margin + acc[r] - dn;         // column direction: against S_nn
      const float hr = margin + acc[r] - diag[m];    // row direction: against S_mm
      if (hc > 0.f) { part += hc; g += inv; cc += 1.f; }
      if (hr > 0.f) { part += hr; g += inv; rc += 1.f; }
    }
    if (m < N && n < N) w.G[(long long)m * N + n] = g;
    // row m's active count: the 16 lanes that share (lane >> 4)
    rc += __shfl_xor(rc, 1); rc += __shfl_xor(rc, 2); rc += __shfl_xor(rc, 4); rc += __shfl_xor(rc, 8);
    if ((lane & 15) == 0 && m < N && rc != 0.f) atomicAdd(w.rowc + m, rc);
  }
  cc += __shfl_xor(cc, 16); cc += __shfl_xor(cc, 32);   // column n: the four lane groups
  if (lane < 16 && n < N && cc != 0.f) atomicAdd(w.colc + n, cc);
  part = block_sum<4>(part, red);
  if (partials) {      // deterministic: one partial per tile, added in tile order by loss_sum_kernel
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = part * inv;
    return;
  }
  if (threadIdx.x == 0 && part != 0.f) atomicAdd(loss, part * inv);
}
__global__ void loss_sum_kernel(const float* __restrict__ partials, int n, float* loss) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += partials[i];
    loss[0] = s;
  }
}
// ---- opt-in extension (NOT in the reference, whose loss sums ALL negatives: pig/loss.py:41-48): in-batch
// hardest-negative mining.  loss = (1/N) sum_i [ relu(m + max_{j != i} S_ij - S_ii) + relu(m + max_{j != i} S_ji - S_ii) ].
// One workgroup per (anchor i, side): lane-parallel scores against every candidate j, then a wavefront-64 arg-max
// butterfly ((value, index) pairs through ds-free lane shuffles; ties go to the smaller index), the four waves meet in LDS.
// The winner's coefficient 1/N is scattered into the dense G of the all-negatives path (at most two adders per entry, both
// adding the same value: order-independent) and rowc / colc carry N x "hinge active", so that pp_triplet_loss_bwd runs
// unchanged: its diagonal term -(rowc + colc) / N^2 becomes -(active_r + active_c) / N.
__global__ __launch_bounds__(256) void hardest_kernel(const LossWs w, int N, int D, float margin, float* __restrict__ partials) {
  extern __shared__ float anchor[];          // D floats, then 4 x (value, index)
  float* wbest = anchor + D;
  const int i = blockIdx.x, side = blockIdx.y;
  const float* arow = (side ? w.An : w.Vn) + (long long)i * D;
  const float* cand = side ? w.Vn : w.An;
  for (int d = threadIdx.x; d < D; d += 256) anchor[d] = arow[d];
  __syncthreads();
  float best = -3.0e38f;
  int bidx = 0x7fffffff;
  for (int j = threadIdx.x; j < N; j += 256) {
    if (j == i) continue;
    const float* c = cand + (long long)j * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += anchor[d] * c[d];
    if (s > best) { best = s; bidx = j; }          // j increases: a tie keeps the smaller index
  }
  // wavefront-64 arg-max: butterfly over (value, index); equal values -> smaller index, so every lane ends with the same pair
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { wbest[2 * wave] = best; wbest[2 * wave + 1] = __int_as_float(bidx); }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      const float ov = wbest[2 * k];
      const int oi = __float_as_int(wbest[2 * k + 1]);
      if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    const float invn = 1.f / (float)N;
    float h = 0.f;
    if (N > 1) h = margin + best - w.diag[i];
    const bool active = N > 1 && h > 0.f;
    if (active) atomicAdd(w.G + (side ? (long long)bidx * N + i : (long long)i * N + bidx), invn);
    (side ? w.colc : w.rowc)[i] = active ? (float)N : 0.f;
    partials[side * N + i] = active ? h * invn : 0.f;
  }
}
__global__ void scale_f32_kernel(float* x, const float* sc, long long n) {
  const float k = sc[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] *= k;
}

}  // namespace

#define S_ ((hipStream_t)s)
#define RC(x) do { int rc_ = (x); if (rc_ != PP_OK) return rc_; } while (0)

extern "C" int pp_spatial_mean_fwd(const void* x, float* out, int B, int T, int HW, int C, int Cp, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && HW > 0 && C > 0 && Cp >= C, "pp_spatial_mean_fwd: sizes");
  hipLaunchKernelGGL(spatial_mean_fwd_kernel, dim3(B * T), dim3(256), 0, S_, (const h16raw*)x, out, HW, C, Cp);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_spatial_mean_bwd(const float* dout, void* dx, int B, int T, int HW, int C, int Cp, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && HW > 0 && C > 0 && Cp >= C, "pp_spatial_mean_bwd: sizes");
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(B * T), dim3(256), 0, S_, dout, (h16raw*)dx, HW, C, Cp);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" size_t pp_attnpool_ws_floats(int B, int T, int F, int Hd, int E) {
  return (size_t)B * E + (size_t)B * F + (size_t)B * T * F + (size_t)B * T * Hd;
}

extern "C" int pp_avgpool_tf_fwd(const float* x, int B, int T, int F, int S, float* out, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && T <= 8192 && F > 0 && S > 0, "pp_avgpool_tf_fwd: sizes");
  hipLaunchKernelGGL(avgpool_tf_fwd_kernel, dim3(B), dim3(256), (size_t)T * 4, S_, x, out, T, F, S);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_avgpool_tf_bwd(const float* dout, int B, int T, int F, int S, float* dx, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && T <= 8192 && F > 0 && S > 0, "pp_avgpool_tf_bwd: sizes");
  hipLaunchKernelGGL(avgpool_tf_bwd_kernel, dim3(B), dim3(256), (size_t)T * 4, S_, dout, dx, T, F, S);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_attnpool_fwd(const float* x, int B, int T, int F, int Hd, int E, const float* W1, const float* b1,
                               const float* W2, const float* b2, const float* Wp, const float* bp, int normalize, float* hid,
                               float* alpha, float* pooled, float* pre, float* out, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && F > 0 && Hd > 0 && E > 0 && (Wp || E == F), "pp_attnpool_fwd: sizes");
  const int BT = B * T;
  RC(sgemm(S_, x, F, 1, W1, 1, F, hid, Hd, 1, BT, Hd, F, b1, 1));
  RC(sgemm(S_, hid, Hd, 1, W2, 1, Hd, alpha, F, 1, BT, F, Hd, b2, 0));
  hipLaunchKernelGGL(timepool_fwd_kernel, dim3(B), dim3(F < 256 ? ((F + 63) / 64) * 64 : 256), 0, S_, alpha, x, pooled, T, F);
  if (Wp) RC(sgemm(S_, pooled, F, 1, Wp, 1, F, pre, E, 1, B, E, F, bp, 0));
  else hipLaunchKernelGGL(copy_f32_kernel, dim3((B * E + 255) / 256), dim3(256), 0, S_, pooled, pre, (long long)B * E);
  if (normalize) hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(B), dim3(256), 0, S_, pre, out, (float*)nullptr, E, 1e-12f);
  else hipLaunchKernelGGL(copy_f32_kernel, dim3((B * E + 255) / 256), dim3(256), 0, S_, pre, out, (long long)B * E);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_attnpool_bwd(const float* dout, const float* x, int B, int T, int F, int Hd, int E, const float* W1,
                               const float* W2, const float* Wp, int normalize, const float* hid, const float* alpha,
                               const float* pooled, const float* pre, const float* out, float* dx, float* dW1, float* db1, float* dW2, float* db2,
                               float* dWp, float* dbp, float* ws, pp_stream_t s) {
  PP_CHECK_ARG(B > 0 && T > 0 && F > 0 && Hd > 0 && E > 0 && ws && (Wp || E == F), "pp_attnpool_bwd: sizes");
  const int BT = B * T;
  float* dpre = ws;
  float* dpooled = dpre + (size_t)B * E;
  float* de = dpooled + (size_t)B * F;
  float* da = de + (size_t)BT * F;
  if (normalize) hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(B), dim3(256), 0, S_, dout, pre, out, dpre, E, 1e-12f);
  else hipLaunchKernelGGL(copy_f32_kernel, dim3((B * E + 255) / 256), dim3(256), 0, S_, dout, dpre, (long long)B * E);
  if (Wp) {
    RC(sgemm(S_, dpre, 1, E, pooled, F, 1, dWp, F, 1, E, F, B));            // dWp[e][f] = sum_b dpre[b][e] pooled[b][f]
    RC(colsum_f32(S_, dpre, B, E, E, dbp));
    RC(sgemm(S_, dpre, E, 1, Wp, F, 1, dpooled, F, 1, B, F, E));            // dpooled = dpre Wp
  } else {
    hipLaunchKernelGGL(copy_f32_kernel, dim3((B * E + 255) / 256), dim3(256), 0, S_, dpre, dpooled, (long long)B * E);
  }
  hipLaunchKernelGGL(timepool_bwd_kernel, dim3(B), dim3(F < 256 ? ((F + 63) / 64) * 64 : 256), 0, S_, alpha, x, dpooled, de, dx, T, F);
  RC(sgemm(S_, de, 1, F, hid, Hd, 1, dW2, Hd, 1, F, Hd, BT, nullptr, 0, nullptr, 0, pick_ksplit(F, Hd, BT)));   // dW2[f][h]
  RC(colsum_f32(S_, de, BT, F, F, db2));
  RC(sgemm(S_, de, F, 1, W2, Hd, 1, da, Hd, 1, BT, Hd, F, nullptr, 2, hid));  // da = (de W2) * (1 - hid^2)
  RC(sgemm(S_, da, 1, Hd, x, F, 1, dW1, F, 1, Hd, F, BT, nullptr, 0, nullptr, 0, pick_ksplit(Hd, F, BT)));       // dW1[h][f]
  RC(colsum_f32(S_, da, BT, Hd, Hd, db1));
  RC(sgemm(S_, da, Hd, 1, W1, F, 1, dx, F, 1, BT, F, Hd, nullptr, 0, nullptr, 1));  // dx += da W1
  return PP_OK;
}

extern "C" size_t pp_triplet_workspace_bytes(int N, int D) {
  return ((size_t)4 * N * D + (size_t)N * N + 5 * (size_t)N) * sizeof(float);
}

extern "C" int pp_triplet_loss_fwd(const float* V, const float* A, int N, int D, float margin, float* loss, void* ws,
                                   size_t ws_bytes, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && D > 0 && V && A && loss && ws, "pp_triplet_loss_fwd: bad arguments");
  PP_CHECK_ARG(ws_bytes >= pp_triplet_workspace_bytes(N, D), "pp_triplet_loss_fwd: workspace too small");
  const LossWs w = loss_ws(ws, N, D);
  // two launches (the N = world x B global loss sits between the embedding all-gather and the backward pass)
  hipLaunchKernelGGL(loss_prep_kernel, dim3(N), dim3(256), 0, S_, V, A, w, loss, D);
  // (deterministic mode: the tiles' loss partials meet in the dVn area of the workspace, which only the backward pass uses)
  const int nt = (N + 31) / 32;
  float* const partials = (pp_opt_deterministic && (size_t)nt * nt <= (size_t)N * D) ? w.dVn : nullptr;
  hipLaunchKernelGGL(loss_tile_kernel, dim3(nt, nt), dim3(256), 0, S_, w, (const float*)w.diag, loss, N, D, margin, partials);
  if (partials) hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, S_, (const float*)partials, nt * nt, loss);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

/* Opt-in hardest-negative variant of pp_triplet_loss_fwd (same workspace, same backward entry: pp_triplet_loss_bwd). */
extern "C" int pp_triplet_loss_hardest_fwd(const float* V, const float* A, int N, int D, float margin, float* loss, void* ws,
                                           size_t ws_bytes, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && D > 0 && V && A && loss && ws, "pp_triplet_loss_hardest_fwd: bad arguments");
  PP_CHECK_ARG(ws_bytes >= pp_triplet_workspace_bytes(N, D) && (size_t)2 * N <= (size_t)N * D, "pp_triplet_loss_hardest_fwd: workspace too small");
  const LossWs w = loss_ws(ws, N, D);
  hipLaunchKernelGGL(loss_prep_kernel, dim3(N), dim3(256), 0, S_, V, A, w, loss, D);
  if (hipMemsetAsync(w.G, 0, (size_t)N * N * 4, S_) != hipSuccess) { pp_set_error("pp_triplet_loss_hardest_fwd: memset"); return PP_ERR_HIP; }
  hipLaunchKernelGGL(hardest_kernel, dim3(N, 2), dim3(256), (size_t)(D + 8) * 4, S_, w, N, D, margin, w.dVn);
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, S_, (const float*)w.dVn, 2 * N, loss);     // ordered: bitwise reproducible
  PP_LAUNCH_CHECK();
  return PP_OK;
}

static int launch_pair(hipStream_t s, const SG& p0, const SG& p1) {
  const int M = p0.M > p1.M ? p0.M : p1.M, N = p0.N > p1.N ? p0.N : p1.N;
  hipLaunchKernelGGL(sgemm_pair_kernel, dim3((N + 31) / 32, (M + 31) / 32, 2), dim3(256), 0, s, p0, p1);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_triplet_loss_bwd(const float* V, const float* A, int N, int D, const float* dloss, const void* ws,
                                   float* dV, float* dA, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && D > 0 && V && A && dloss && ws && dV && dA, "pp_triplet_loss_bwd: bad arguments");
  const LossWs w = loss_ws((void*)ws, N, D);
  const SG gv{w.G, N, 1, w.An, D, 1, w.dVn, D, 1, N, D, N, nullptr, nullptr, 0, 0, 1};   // dVn = G An   (off-diagonal part)
  const SG ga{w.G, 1, N, w.Vn, D, 1, w.dAn, D, 1, N, D, N, nullptr, nullptr, 0, 0, 1};   // dAn = G^T Vn
  RC(launch_pair(S_, gv, ga));
  const CosBwd cv{w.dVn, w.Vn, w.vnorm, w.An, dV}, ca{w.dAn, w.An, w.anorm, w.Vn, dA};
  hipLaunchKernelGGL(cosnorm_bwd_kernel, dim3(N, 2), dim3(256), 0, S_, cv, ca, dloss, (const float*)w.rowc, (const float*)w.colc,
                     1.f / ((float)N * (float)N), N, N, D);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// ---- API-surface helpers: pig/loss.py:51-55 cosine_matrix, pig/loss.py:41-48 contrastive (forward),
//      pig/metrics.py:45-52 triplet_accuracy --------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void triplet_acc_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                          const float* __restrict__ n, int D, int discrete, float* out) {
  __shared__ float red[4];
  const long long r = blockIdx.x;
  float aa = 0.f, pp = 0.f, nn = 0.f, ap = 0.f, an = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float x = a[r * D + d], y = p[r * D + d], z = n[r * D + d];
    aa += x * x; pp += y * y; nn += z * z; ap += x * y; an += x * z;
  }
  aa = block_sum<4>(aa, red); pp = block_sum<4>(pp, red); nn = block_sum<4>(nn, red);
  ap = block_sum<4>(ap, red); an = block_sum<4>(an, red);
  if (threadIdx.x == 0) {
    // torch 1.9.1 F.cosine_similarity: w12 / sqrt(clamp(w1*w2, eps^2)), eps = 1e-8
    const float eps2 = 1e-16f;
    const float sp = ap / sqrtf(fmaxf(aa * pp, eps2)), sn = an / sqrtf(fmaxf(aa * nn, eps2));
    const float diff = sp - sn;
    out[r] = discrete ? ((diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) + 1.f) * 0.5f : diff;
  }
}
// ---- rank-based recall (pig/metrics.py:7-42, 54-81) on a similarity matrix ----------------------------------------
// One workgroup per (query row, index set).  The reference sorts distances 1 - cos ascending and intersects the first n
// positions with the targets; element k sits at position #{k' : d[k'] < d[k], or d[k'] == d[k] and k' < k}, so no sort
// is needed: every target counts the candidates ahead of it.  hist[r] = targets at position r; recall@n = prefix / #targets.
__global__ __launch_bounds__(256) void recall_kernel(const float* __restrict__ S, int Nc, int ld, const int* __restrict__ idx,
                                                     int size, const unsigned char* __restrict__ correct, int Nmax,
                                                     float* __restrict__ out) {
  extern __shared__ float dist[];            // [ncols] distances of this row, then Nmax + 1 ints
  const int j = blockIdx.x, set = blockIdx.y;
  const int ncols = idx ? size : Nc;
  int* hist = (int*)(dist + ncols);          // hist[0 .. Nmax-1], hist[Nmax] = number of targets
  const int* ix = idx ? idx + (long long)set * size : nullptr;
  const int row = ix ? ix[j] : j;
  for (int k = threadIdx.x; k < ncols; k += 256) dist[k] = 1.f - S[(long long)row * ld + (ix ? ix[k] : k)];
  for (int n = threadIdx.x; n <= Nmax; n += 256) hist[n] = 0;
  __syncthreads();
  for (int k = threadIdx.x; k < ncols; k += 256) {
    const bool target = correct ? correct[(long long)row * Nc + (ix ? ix[k] : k)] != 0 : k == j;
    if (!target) continue;
    const float dk = dist[k];
    int pos = 0;
    for (int q = 0; q < ncols; ++q) pos += (dist[q] < dk) || (dist[q] == dk && q < k);
    atomicAdd(&hist[Nmax], 1);
    if (pos < Nmax) atomicAdd(&hist[pos], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nt = (float)hist[Nmax];      // (no targets: 0 / 0 = NaN; the reference raises ZeroDivisionError)
    int acc = 0;
    for (int n = 1; n <= Nmax; ++n) {
      acc += hist[n - 1];
      out[((long long)set * Nmax + (n - 1)) * gridDim.x + j] = (float)acc / nt;
    }
  }
}

}  // namespace

extern "C" int pp_triplet_accuracy(const float* a, const float* p, const float* n, int M, int D, int discrete, float* out,
                                   pp_stream_t s) {
  PP_CHECK_ARG(M > 0 && D > 0 && a && p && n && out, "pp_triplet_accuracy: bad arguments");
  hipLaunchKernelGGL(triplet_acc_kernel, dim3(M), dim3(256), 0, S_, a, p, n, D, discrete, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_recall_at_n(const float* S, int Nr, int Nc, int ld, const int* idx, int nsets, int size,
                              const unsigned char* correct, int Nmax, float* out, pp_stream_t s) {
  PP_CHECK_ARG(S && out && Nr > 0 && Nc > 0 && ld >= Nc && Nmax > 0 && Nmax <= 4096, "pp_recall_at_n: bad arguments");
  PP_CHECK_ARG(idx ? (nsets > 0 && size > 0) : true, "pp_recall_at_n: index sets");
  const int rows = idx ? size : Nr, ncols = idx ? size : Nc, sets = idx ? nsets : 1;
  PP_CHECK_ARG(correct || ncols >= rows, "pp_recall_at_n: identity targets need a column per row");
  const size_t shm = (size_t)ncols * 4 + (size_t)(Nmax + 1) * 4;
  PP_CHECK_ARG(shm <= 64 * 1024, "pp_recall_at_n: %d candidates per row exceed the LDS staging (max ~16000)", ncols);
  hipLaunchKernelGGL(recall_kernel, dim3(rows, sets), dim3(256), shm, S_, S, Nc, ld, idx, size, correct, Nmax, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_cosine_matrix(const float* U, const float* Vv, int Nu, int Nv, int D, float* out, float* ws,
                                pp_stream_t s) {
  PP_CHECK_ARG(Nu > 0 && Nv > 0 && D > 0 && U && Vv && out && ws, "pp_cosine_matrix: bad arguments");
  float* Un = ws;
  float* Vn = ws + (size_t)Nu * D;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(Nu), dim3(256), 0, S_, U, Un, (float*)nullptr, D, 0.f);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(Nv), dim3(256), 0, S_, Vv, Vn, (float*)nullptr, D, 0.f);
  RC(sgemm(S_, Un, D, 1, Vn, 1, D, out, Nv, 1, Nu, Nv, D));
  return PP_OK;
}

extern "C" int pp_contrastive_fwd(const float* S, int N, float margin, float* loss, float* ws, pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && S && loss && ws, "pp_contrastive_fwd: bad arguments");
  float* G = ws;                      // N*N copy (hinge_kernel overwrites its input)
  float* diag = G + (size_t)N * N;
  float* rowc = diag + N;
  float* colc = rowc + N;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((N * N + 255) / 256), dim3(256), 0, S_, S, G, (long long)N * N);
  hipLaunchKernelGGL(diag_kernel, dim3((N + 255) / 256), dim3(256), 0, S_, G, diag, rowc, colc, loss, N);
  hipLaunchKernelGGL(hinge_kernel, dim3(N), dim3(256), 0, S_, G, diag, rowc, colc, loss, N, margin);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// Backward of cosine_matrix (the reference's is an ordinary differentiable torch expression, pig/loss.py:51-55):
// dU = J_norm(U)^T (dS Vn), dV = J_norm(V)^T (dS^T Un).  ws: 2 (Nu + Nv) D + Nu + Nv floats.
extern "C" int pp_cosine_matrix_bwd(const float* U, const float* Vv, int Nu, int Nv, int D, const float* dS, float* dU,
                                    float* dV, float* ws, pp_stream_t s) {
  PP_CHECK_ARG(Nu > 0 && Nv > 0 && D > 0 && U && Vv && dS && dU && dV && ws, "pp_cosine_matrix_bwd: bad arguments");
  float* Un = ws;
  float* Vn = Un + (size_t)Nu * D;
  float* dUn = Vn + (size_t)Nv * D;
  float* dVn = dUn + (size_t)Nu * D;
  float* un = dVn + (size_t)Nv * D;
  float* vn = un + Nu;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(Nu), dim3(256), 0, S_, U, Un, un, D, 0.f);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(Nv), dim3(256), 0, S_, Vv, Vn, vn, D, 0.f);
  const SG gu{dS, Nv, 1, Vn, D, 1, dUn, D, 1, Nu, D, Nv, nullptr, nullptr, 0, 0, 1};   // dUn = dS Vn
  const SG gv{dS, 1, Nv, Un, D, 1, dVn, D, 1, Nv, D, Nu, nullptr, nullptr, 0, 0, 1};   // dVn = dS^T Un
  RC(launch_pair(S_, gu, gv));
  const CosBwd cu{dUn, Un, un, nullptr, dU}, cv{dVn, Vn, vn, nullptr, dV};
  hipLaunchKernelGGL(cosnorm_bwd_kernel, dim3(Nu > Nv ? Nu : Nv, 2), dim3(256), 0, S_, cu, cv, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, 0.f, Nu, Nv, D);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

// Backward of contrastive(M, margin) (pig/loss.py:41-48): dM = dloss * G, G_ij = ([hinge_c > 0] + [hinge_r > 0]) / N^2 off
// the diagonal and minus the number of active terms of row i and column i over N^2 on it.  ws: 3 N + 1 floats.
extern "C" int pp_contrastive_bwd(const float* S, int N, float margin, const float* dloss, float* dS, float* ws,
                                  pp_stream_t s) {
  PP_CHECK_ARG(N > 0 && S && dloss && dS && ws, "pp_contrastive_bwd: bad arguments");
  float* diag = ws;
  float* rowc = diag + N;
  float* colc = rowc + N;
  float* scratch = colc + N;
  hipLaunchKernelGGL(copy_f32_kernel, dim3((N * N + 255) / 256), dim3(256), 0, S_, S, dS, (long long)N * N);
  hipLaunchKernelGGL(diag_kernel, dim3((N + 255) / 256), dim3(256), 0, S_, dS, diag, rowc, colc, scratch, N);
  hipLaunchKernelGGL(hinge_kernel, dim3(N), dim3(256), 0, S_, dS, diag, rowc, colc, scratch, N, margin);
  hipLaunchKernelGGL(gdiag_kernel, dim3((N + 255) / 256), dim3(256), 0, S_, dS, rowc, colc, N);
  hipLaunchKernelGGL(scale_f32_kernel, dim3((N * N + 255) / 256), dim3(256), 0, S_, dS, dloss, (long long)N * N);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
