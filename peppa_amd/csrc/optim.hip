// Fused multi-tensor BertAdam step (gfx950, HBM-bound: 28 B/param).
// Restates pig/optimization.py:101-179: per-tensor clip_grad_norm_(p, max_grad_norm) (coef =
// max_norm/(norm+1e-6), clamped to 1), m = b1*m + (1-b1)*g, v = b2*v + (1-b2)*g*g,
// update = m/(sqrt(v)+eps) + wd*p, p -= lr_scheduled*update; no bias correction.
#include "common.h"

extern int pp_opt_deterministic;

namespace {

// Both kernels stream with 16-byte accesses where the tensor allows it (every pointer 16-byte aligned; chunk offsets are
// multiples of 65536 elements): four times fewer memory instructions and four times the bytes in flight per thread than
// the scalar loop, which left the 4-GB pass at 3.4 TB/s.
__device__ __forceinline__ bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const pp_tensor_list tl, const int* __restrict__ chunk_tensor,
                                                    const long long* __restrict__ chunk_off, int chunk, float* norms,
                                                    const float* __restrict__ skip, float* __restrict__ partials) {
  __shared__ float red[4];
  if (skip && *skip != 0.f) return;     // an overflowed fp16 step: nothing is updated (uniform for the whole grid)
  const int t = chunk_tensor[blockIdx.x];
  const long long off = chunk_off[blockIdx.x];
  const long long n = tl.numel[t];
  const float* g = tl.g[t];
  const long long end = off + chunk < n ? off + chunk : n;
  float s = 0.f;
  long long done = off;
  if (aligned16(g, g, g, g)) {
    const long long nvec = (end - off) >> 2;
    const float4* g4 = (const float4*)(g + off);
    for (long long i = threadIdx.x; i < nvec; i += 256) {
      const float4 v = g4[i];
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    done = off + (nvec << 2);
  }
  for (long long i = done + threadIdx.x; i < end; i += 256) { const float v = g[i]; s += v * v; }
  s = block_sum<4>(s, red);
  if (partials) {       // deterministic: one partial per chunk; bertadam_kernel adds a tensor's chunks in order
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
    return;
  }
  if (threadIdx.x == 0) atomicAdd(norms + t, s);
}

__global__ __launch_bounds__(256) void bertadam_kernel(const pp_tensor_list tl, const int* __restrict__ chunk_tensor,
                                                       const long long* __restrict__ chunk_off, int chunk,
                                                       const float* __restrict__ norms, float lr, float b1, float b2, float eps,
                                                       float wd, float max_norm, const float* __restrict__ lr_t,
                                                       const float* __restrict__ skip, const float* __restrict__ partials,
                                                       int n_chunks) {
  if (skip && *skip != 0.f) return;
  const int t = chunk_tensor[blockIdx.x];
  if (lr_t) lr = lr_t[t];   // per-tensor scheduled learning rate (tensors whose step counts differ share one launch)
  const long long off = chunk_off[blockIdx.x];
  const long long n = tl.numel[t];
  float* p = tl.p[t];
  const float* g = tl.g[t];
  float* m = tl.m[t];
  float* v = tl.v[t];
  float coef = 1.f;
  if (max_norm > 0.f) {
    float nsq;
    if (partials) {     // the tensor's chunks are consecutive blocks: sum their partial squares first to last
      int c0 = blockIdx.x, c1 = blockIdx.x;
      while (c0 > 0 && chunk_tensor[c0 - 1] == t) --c0;
      while (c1 + 1 < n_chunks && chunk_tensor[c1 + 1] == t) ++c1;
      nsq = 0.f;
      for (int c = c0; c <= c1; ++c) nsq += partials[c];
    } else {
      nsq = norms[t];
    }
    coef = max_norm / (sqrtf(nsq) + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  const long long end = off + chunk < n ? off + chunk : n;
  auto one = [&](const float gr, float& mi, float& vi, float& pi) __attribute__((always_inline)) {
    const float gi = gr * coef;
    mi = mi * b1 + (1.f - b1) * gi;
    vi = vi * b2 + (1.f - b2) * gi * gi;
    float upd = mi / (sqrtf(vi) + eps);
    if (wd > 0.f) upd += wd * pi;
    pi = pi - lr * upd;
  };
  long long done = off;
  if (aligned16(p, g, m, v)) {
    const long long nvec = (end - off) >> 2;
    float4* p4 = (float4*)(p + off);
    const float4* g4 = (const float4*)(g + off);
    float4* m4 = (float4*)(m + off);
    float4* v4 = (float4*)(v + off);
    for (long long i = threadIdx.x; i < nvec; i += 256) {
      const float4 gg = g4[i];
      float4 mm = m4[i], vv = v4[i], pp = p4[i];
      one(gg.x, mm.x, vv.x, pp.x);
      one(gg.y, mm.y, vv.y, pp.y);
      one(gg.z, mm.z, vv.z, pp.z);
      one(gg.w, mm.w, vv.w, pp.w);
      m4[i] = mm;
      v4[i] = vv;
      p4[i] = pp;
    }
    done = off + (nvec << 2);
  }
  for (long long i = done + threadIdx.x; i < end; i += 256) {
    float mi = m[i], vi = v[i], pi = p[i];
    one(g[i], mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

// g /= scale; found_inf = 1 on any non-finite element (torch._amp_foreach_non_finite_check_and_unscale_)
__global__ __launch_bounds__(256) void unscale_check_kernel(const pp_tensor_list tl, const int* __restrict__ chunk_tensor,
                                                            const long long* __restrict__ chunk_off, int chunk,
                                                            const float* __restrict__ scale, float* found_inf) {
  const int t = chunk_tensor[blockIdx.x];
  const long long off = chunk_off[blockIdx.x];
  const long long n = tl.numel[t];
  float* g = (float*)tl.g[t];
  const long long end = off + chunk < n ? off + chunk : n;
  const float k = 1.f / scale[0];
  bool bad = false;
  long long done = off;
  if (aligned16(g, g, g, g)) {
    const long long nvec = (end - off) >> 2;
    float4* g4 = (float4*)(g + off);
    for (long long i = threadIdx.x; i < nvec; i += 256) {
      float4 v = g4[i];
      bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
      v.x *= k; v.y *= k; v.z *= k; v.w *= k;
      g4[i] = v;
    }
    done = off + (nvec << 2);
  }
  for (long long i = done + threadIdx.x; i < end; i += 256) {
    const float v = g[i];
    bad |= !isfinite(v);
    g[i] = v * k;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.f;
}

__global__ void amp_update_scale_kernel(float* scale, int* tracker, const float* found_inf, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*found_inf != 0.f) {
    *scale = *scale * backoff;
    *tracker = 0;
  } else {
    const int succ = *tracker + 1;
    if (succ == interval) {
      const float ns = *scale * growth;
      if (isfinite(ns)) *scale = ns;
      *tracker = 0;
    } else {
      *tracker = succ;
    }
  }
}

}  // namespace

extern "C" int pp_grad_unscale_check(const pp_tensor_list* tl, const int* chunk_tensor, const long long* chunk_off, int n_chunks,
                                     int chunk, const float* scale, float* found_inf, pp_stream_t s) {
  PP_CHECK_ARG(tl && tl->n_tensors > 0 && n_chunks > 0 && chunk > 0 && scale && found_inf, "pp_grad_unscale_check: bad arguments");
  hipLaunchKernelGGL(unscale_check_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)s, *tl, chunk_tensor, chunk_off, chunk,
                     scale, found_inf);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_amp_update_scale(float* scale, int* growth_tracker, const float* found_inf, float growth_factor,
                                   float backoff_factor, int growth_interval, pp_stream_t s) {
  PP_CHECK_ARG(scale && growth_tracker && found_inf && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f &&
               growth_interval > 0, "pp_amp_update_scale: bad arguments");
  hipLaunchKernelGGL(amp_update_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, scale, growth_tracker, found_inf,
                     growth_factor, backoff_factor, growth_interval);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_bertadam_step(const pp_tensor_list* tl, const int* chunk_tensor, const long long* chunk_off, int n_chunks,
                                int chunk, float* norms, float lr_scheduled, float b1, float b2, float eps, float weight_decay,
                                float max_grad_norm, const float* lr_per_tensor, const float* skip_flag, pp_stream_t s) {
  PP_CHECK_ARG(tl && tl->n_tensors > 0 && n_chunks > 0 && chunk > 0 && norms, "pp_bertadam_step: bad arguments");
  hipStream_t st = (hipStream_t)s;
  // deterministic mode: the chunks' partial squares go to norms[n_tensors ..] and every block adds its tensor's in order
  float* const partials = pp_opt_deterministic ? norms + tl->n_tensors : nullptr;
  if (max_grad_norm > 0.f) {
    if (hipMemsetAsync(norms, 0, (size_t)tl->n_tensors * 4, st) != hipSuccess) { pp_set_error("pp_bertadam_step: memset"); return PP_ERR_HIP; }
    hipLaunchKernelGGL(sumsq_kernel, dim3(n_chunks), dim3(256), 0, st, *tl, chunk_tensor, chunk_off, chunk, norms, skip_flag, partials);
  }
  hipLaunchKernelGGL(bertadam_kernel, dim3(n_chunks), dim3(256), 0, st, *tl, chunk_tensor, chunk_off, chunk, norms, lr_scheduled,
                     b1, b2, eps, weight_decay, max_grad_norm, lr_per_tensor, skip_flag, (const float*)partials, n_chunks);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
