// Standalone probe for the LayerNorm-backward finding of DESIGN.md section 7: the SLP-vectorised (packed-FP32) build of
// ln_bwd_kernel returns one row of dx from wrong sums in ~1 launch of 10 when MFMA-issuing waves of another kernel share
// its SIMDs.  The victim below is that kernel (this repo's own peppa_amd/csrc/norm.hip) built with hipcc's DEFAULT flags
// (SLP vectoriser on); with DBG it also writes every lane's partial sums and an XOR of the dwords it loaded, so that a
// failing launch says WHICH lane held WHAT.  Aggressors: 0 none, 1 the library's dense weight gradient (dlopen), 2 a
// register-only MFMA loop, 3 an MFMA loop fed by streaming global loads, 4 streaming loads without MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/ln_pk_repro.hip -o /tmp/ln_pk_repro -ldl
//   hipcc ... -fno-slp-vectorize ...   (the control: 0 failures expected)
//   /tmp/ln_pk_repro [path to libpeppa_hip.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/peppa_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef unsigned short h16raw;
constexpr int LN_MAXC = 2;

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
  unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  return (uint32_t)__builtin_bit_cast(h16raw, (__bf16)lo) | ((uint32_t)__builtin_bit_cast(h16raw, (__bf16)hi) << 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v; v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]); v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]); return v;
}

// ---- the victim: ln_bwd_kernel of peppa_amd/csrc/norm.hip (deterministic form: per-block partials to ws) -------------
template <int DBG>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const h16raw* __restrict__ dy, const h16raw* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, h16raw* __restrict__ dx, int rows, int D,
                                                     int rows_per_wave, float* __restrict__ ws, float* __restrict__ dbg) {
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nch = D >> 3;
  float ag[LN_MAXC][8], ab[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) ag[c][q] = ab[c][q] = 0.f;
  const int r0 = wid * rows_per_wave;
  const int r1 = min(rows, r0 + rows_per_wave);
  float gam[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int q = 0; q < 8; ++q) gam[c][q] = ch < nch ? gamma[ch * 8 + q] : 0.f;
  }
  uint4 nd[LN_MAXC], nx[LN_MAXC];
  auto fetch = [&](const int row) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch && row < r1) {
        nd[c] = *(const uint4*)(dy + (long long)row * D + ch * 8);
        nx[c] = *(const uint4*)(x + (long long)row * D + ch * 8);
      }
    }
  };
  fetch(r0);
  for (int row = r0; row < r1; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXC][8], xh[LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
    uint4 cd[LN_MAXC], cx[LN_MAXC];
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) { cd[c] = nd[c]; cx[c] = nx[c]; }
    fetch(row + 1);
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float d[8], xx[8];
        unpack8(cd[c], d);
        unpack8(cx[c], xx);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          xh[c][q] = (xx[q] - mu) * rs;
          g[c][q] = d[q] * gam[c][q];
          s1 += g[c][q];
          s2 += g[c][q] * xh[c][q];
          ag[c][q] += d[q] * xh[c][q];
          ab[c][q] += d[q];
        }
      }
    }
    if (DBG == 1) {
      uint32_t kx = 0, kd = 0;
#pragma unroll
      for (int c = 0; c < LN_MAXC; ++c)
        if (lane + 64 * c < nch) { kx ^= cx[c].x ^ cx[c].y ^ cx[c].z ^ cx[c].w; kd ^= cd[c].x ^ cd[c].y ^ cd[c].z ^ cd[c].w; }
      *(float4*)(dbg + ((long long)row * 64 + lane) * 4) = make_float4(s1, s2, __uint_as_float(kx), __uint_as_float(kd));
    }
    const float part1 = s1, part2 = s2;      // (DBG == 2: this lane's partial sums, stored AFTER the row's dx)
    s1 = wave_sum(s1) / D;
    s2 = wave_sum(s2) / D;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = rs * (g[c][q] - s1 - xh[c][q] * s2);
        *(uint4*)(dx + (long long)row * D + ch * 8) = pack8(o);
      }
    }
    if (DBG == 2) *(float4*)(dbg + ((long long)row * 64 + lane) * 4) = make_float4(part1, part2, s1, s2);
  }
  __shared__ float red[2][4][64 * 8 * LN_MAXC];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch)
#pragma unroll
      for (int q = 0; q < 8; ++q) { red[0][w][ch * 8 + q] = ag[c][q]; red[1][w][ch * 8 + q] = ab[c][q]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 256) {
    ws[((long long)blockIdx.x * 2 + 0) * D + i] = red[0][0][i] + red[0][1][i] + red[0][2][i] + red[0][3][i];
    ws[((long long)blockIdx.x * 2 + 1) * D + i] = red[1][0][i] + red[1][1][i] + red[1][2][i] + red[1][3][i];
  }
}

// ---- self-contained aggressors: MFMA loops that leave room for the victim's waves on every SIMD ----------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int LOADS, int MFMA>
__global__ __launch_bounds__(256) void aggressor_kernel(const uint4* __restrict__ src, long long n16, float* out, int iters) {
  __shared__ uint4 pad[48 * 1024 / 16];   // 48 KB: two or three of these workgroups per CU, like the register-staged kernels
  pad[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
  __syncthreads();
  f32x4 acc[4] = {};
  uint4 a = pad[(threadIdx.x * 7) & 255], b = pad[(threadIdx.x * 13) & 255];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (LOADS) { const uint4 v = src[i % n16]; a.x ^= v.x; b.y ^= v.y; a.z += v.z; b.w += v.w; i += (long long)gridDim.x * 256; }
    if (MFMA) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
    } else {
      acc[0][0] += __uint_as_float(a.x & 0x3fffffffu);
    }
  }
  out[(long long)blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

constexpr int MAXS = 8;
__global__ void cmp_kernel(const uint4* dx, const uint4* ref, int n16, int* nfail, int* nlaunch, uint4* save_dx,
                           const uint4* dbg, uint4* save_dbg, int ndbg16) {
  __shared__ int bad, slot;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int b = 0;
  for (int i = threadIdx.x; i < n16; i += blockDim.x) {
    const uint4 p = dx[i], q = ref[i];
    b |= (p.x != q.x) | (p.y != q.y) | (p.z != q.z) | (p.w != q.w);
  }
  if (b) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(nlaunch, 1); slot = bad ? atomicAdd(nfail, 1) : -1; }
  __syncthreads();
  if (slot >= 0 && slot < MAXS) {
    for (int i = threadIdx.x; i < n16; i += blockDim.x) save_dx[(long long)slot * n16 + i] = dx[i];
    if (dbg) for (int i = threadIdx.x; i < ndbg16; i += blockDim.x) save_dbg[(long long)slot * ndbg16 + i] = dbg[i];
  }
}

static h16raw f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (h16raw)(u >> 16); }
static float bf2f(h16raw h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float rnd(uint64_t& s) {   // ~N(0,1): sum of 12 uniforms
  float t = 0; for (int i = 0; i < 12; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; t += (float)((s >> 40) & 0xffff) / 65536.f; } return t - 6.f;
}

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "peppa_amd/libpeppa_hip.so";
  const int rows = 228, D = 768, iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 150;
  std::vector<h16raw> hx(rows * D), hdy(rows * D);
  std::vector<float> hg(D, 1.f), hmean(rows), hrstd(rows);
  uint64_t seed = 1234;
  for (int r = 0; r < rows; ++r) {
    double s = 0, ss = 0;
    for (int c = 0; c < D; ++c) { hx[r * D + c] = f2bf(rnd(seed)); hdy[r * D + c] = f2bf(1e-4f * rnd(seed)); const double v = bf2f(hx[r * D + c]); s += v; ss += v * v; }
    hmean[r] = (float)(s / D); hrstd[r] = (float)(1.0 / sqrt(ss / D - (s / D) * (s / D) + 1e-5));
  }
  h16raw *x, *dy, *dx, *ref; float *gam, *mean, *rstd, *ws, *dbg, *dbg_ref; int* cnt; uint4 *save_dx, *save_dbg;
  const int waves = (rows + 3) / 4, rpw = 4, nblk = (waves + 3) / 4, n16 = rows * D / 8, ndbg16 = rows * 64;
  CK(hipMalloc(&x, rows * D * 2)); CK(hipMalloc(&dy, rows * D * 2)); CK(hipMalloc(&dx, rows * D * 2)); CK(hipMalloc(&ref, rows * D * 2));
  CK(hipMalloc(&gam, D * 4)); CK(hipMalloc(&mean, rows * 4)); CK(hipMalloc(&rstd, rows * 4)); CK(hipMalloc(&ws, (size_t)nblk * 2 * D * 4));
  CK(hipMalloc(&dbg, (size_t)ndbg16 * 16)); CK(hipMalloc(&dbg_ref, (size_t)ndbg16 * 16)); CK(hipMalloc(&cnt, 8));
  CK(hipMalloc(&save_dx, (size_t)MAXS * n16 * 16)); CK(hipMalloc(&save_dbg, (size_t)MAXS * ndbg16 * 16));
  CK(hipMemcpy(x, hx.data(), rows * D * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), rows * D * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(gam, hg.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mean, hmean.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rstd, hrstd.data(), rows * 4, hipMemcpyHostToDevice));
  // aggressor operands
  const int M = 64 * 114, Ni = 3072, Kj = 768;
  void *aX, *aDY; float *aDW, *aWS, *aout; uint4* stream_src; const long long src16 = (256ll << 20) / 16;
  CK(hipMalloc(&aX, (size_t)M * Kj * 2)); CK(hipMalloc(&aDY, (size_t)M * Ni * 2)); CK(hipMalloc(&aDW, (size_t)Ni * Kj * 4));
  CK(hipMemset(aX, 0x3c, (size_t)M * Kj * 2)); CK(hipMemset(aDY, 0x3c, (size_t)M * Ni * 2)); CK(hipMemset(aDW, 0, (size_t)Ni * Kj * 4));
  CK(hipMalloc(&aout, 1024 * 256 * 4)); CK(hipMalloc(&stream_src, src16 * 16)); CK(hipMemset(stream_src, 0x11, src16 * 16));
  typedef int (*wgrad_fn)(const pp_wgrad_desc*, void*);
  typedef long long (*wsf_fn)(const pp_wgrad_desc*);
  typedef int (*opt_fn)(const char*, int);
  void* lib = dlopen(libpath, RTLD_NOW);
  wgrad_fn wgrad = lib ? (wgrad_fn)dlsym(lib, "pp_wgrad") : nullptr;
  pp_wgrad_desc wd; memset(&wd, 0, sizeof wd);
  if (wgrad) {
    ((opt_fn)dlsym(lib, "pp_set_option"))("deterministic", 1);
    wd.M = M; wd.Ni = Ni; wd.Kj = Kj; wd.g.mode = PP_DENSE; wd.g.lda = Kj; wd.X = aX; wd.dY = aDY; wd.ldy = Ni; wd.dW = aDW; wd.ldw = Kj; wd.nbatch = 1;
    wd.ws_floats = ((wsf_fn)dlsym(lib, "pp_wgrad_ws_floats"))(&wd);
    if (wd.ws_floats) { CK(hipMalloc(&aWS, (size_t)wd.ws_floats * 4)); wd.ws = aWS; }
  } else printf("(no %s: aggressor 1 skipped)\n", libpath);
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  auto victim = [&](int dbgmode, h16raw* out, float* d) {
    if (dbgmode == 1) hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nblk), dim3(256), 0, s1, dy, x, gam, mean, rstd, out, rows, D, rpw, ws, d);
    else if (dbgmode == 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nblk), dim3(256), 0, s1, dy, x, gam, mean, rstd, out, rows, D, rpw, ws, d);
    else hipLaunchKernelGGL(ln_bwd_kernel<0>, dim3(nblk), dim3(256), 0, s1, dy, x, gam, mean, rstd, out, rows, D, rpw, ws, d);
  };
  std::vector<uint4> hsave((size_t)MAXS * n16), hdbg((size_t)MAXS * ndbg16), hdbg_ref(ndbg16), href(n16);
  const char* names[] = {"none", "library dense wgrad (register-staged MFMA)", "MFMA loop, registers only", "MFMA loop + streaming loads", "streaming loads, no MFMA"};
  const char* vnames[] = {"as shipped            ", "debug stores mid-row  ", "debug stores after row"};
  for (int dbgmode = 0; dbgmode < 3; ++dbgmode) {
    victim(dbgmode, ref, dbg_ref); CK(hipDeviceSynchronize());
    CK(hipMemcpy(href.data(), ref, (size_t)n16 * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdbg_ref.data(), dbg_ref, (size_t)ndbg16 * 16, hipMemcpyDeviceToHost));
    for (int ag = 0; ag < 5; ++ag) {
      if (ag == 1 && !wgrad) continue;
      CK(hipMemset(cnt, 0, 8));
      for (int it = 0; it < iters; ++it) {
        if (ag == 1) { const int rc = wgrad(&wd, s2); if (rc) { printf("pp_wgrad rc %d\n", rc); return 2; } }
        if (ag == 2) hipLaunchKernelGGL((aggressor_kernel<0, 1>), dim3(768), dim3(256), 0, s2, stream_src, src16, aout, 20000);
        if (ag == 3) hipLaunchKernelGGL((aggressor_kernel<1, 1>), dim3(768), dim3(256), 0, s2, stream_src, src16, aout, 4000);
        if (ag == 4) hipLaunchKernelGGL((aggressor_kernel<1, 0>), dim3(768), dim3(256), 0, s2, stream_src, src16, aout, 4000);
        for (int k = 0; k < 8; ++k) {
          victim(dbgmode, dx, dbg);
          hipLaunchKernelGGL(cmp_kernel, dim3(1), dim3(1024), 0, s1, (const uint4*)dx, (const uint4*)ref, n16, cnt, cnt + 1, save_dx, dbgmode ? (const uint4*)dbg : nullptr, save_dbg, ndbg16);
        }
      }
      CK(hipDeviceSynchronize());
      int h[2]; CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
      printf("victim %s | aggressor %-44s: %d of %d launches differ\n", vnames[dbgmode], names[ag], h[0], h[1]); fflush(stdout);
      if (!h[0]) continue;
      const int ns = h[0] < MAXS ? h[0] : MAXS;
      CK(hipMemcpy(hsave.data(), save_dx, (size_t)ns * n16 * 16, hipMemcpyDeviceToHost));
      if (dbgmode) CK(hipMemcpy(hdbg.data(), save_dbg, (size_t)ns * ndbg16 * 16, hipMemcpyDeviceToHost));
      for (int s = 0; s < ns; ++s) {
        const h16raw* got = (const h16raw*)(hsave.data() + (size_t)s * n16); const h16raw* want = (const h16raw*)href.data();
        for (int r = 0; r < rows; ++r) {
          int nd = 0; for (int c = 0; c < D; ++c) nd += got[r * D + c] != want[r * D + c];
          if (!nd) continue;
          if (!dbgmode && s > 0) continue;
          printf("  failure %d: row %d (row %d of its wave), %d columns differ\n", s, r, r % rpw, nd);
          if (!dbgmode) continue;
          const uint4* dg = hdbg.data() + (size_t)s * ndbg16 + r * 64; const uint4* dr = hdbg_ref.data() + r * 64;
          int nl = 0;
          if (dbgmode == 2) {     // partial sums (x, y) and the reduced, divided sums (z, w) of every lane
            int npart = 0, ntot = 0;
            for (int l = 0; l < 64; ++l) { npart += dg[l].x != dr[l].x || dg[l].y != dr[l].y; ntot += dg[l].z != dr[l].z || dg[l].w != dr[l].w; }
            float a[4], b[4]; memcpy(a, &dg[0], 16); memcpy(b, &dr[0], 16);
            printf("    lanes whose PARTIAL sums differ from the clean run: %d; lanes whose REDUCED sums differ: %d; lane 0: s1 %.9g (ref %.9g) s2 %.9g (ref %.9g)\n",
                   npart, ntot, a[2], b[2], a[3], b[3]);
            if (s > 1 || nd < 0) continue;
            for (int l = 0; l < 64 && nl < 6; ++l)
              if (dg[l].x != dr[l].x || dg[l].y != dr[l].y) {
                memcpy(a, &dg[l], 16); memcpy(b, &dr[l], 16);
                printf("      lane %2d: partial s1 %.9g (ref %.9g) [%08x %08x]  partial s2 %.9g (ref %.9g) [%08x %08x]\n", l, a[0], b[0], dg[l].x, dr[l].x, a[1], b[1], dg[l].y, dr[l].y);
                ++nl;
              }
            continue;
          }
          for (int l = 0; l < 64; ++l)
            if (memcmp(&dg[l], &dr[l], 16)) {
              float a[4], b[4]; memcpy(a, &dg[l], 16); memcpy(b, &dr[l], 16);
              printf("    lane %2d: s1 %.9g (ref %.9g)  s2 %.9g (ref %.9g)  loaded-x xor %08x (ref %08x)  loaded-dy xor %08x (ref %08x)\n", l,
                     a[0], b[0], a[1], b[1], dg[l].z, dr[l].z, dg[l].w, dr[l].w);
              ++nl;
            }
          if (!nl) printf("    every lane's partial sums and loaded words equal the clean run's: the fault is after them (reduction / epilogue)\n");
        }
      }
    }
  }
  return 0;
}
