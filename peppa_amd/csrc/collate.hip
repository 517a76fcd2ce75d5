// On-device collate: what pig/data.py:60-78 (`featurize`: frame / 255 -> float, (T,H,W,C) -> (C,T,H,W))
// and pig/util.py:19-33 (`pad_video_batch` / `pad_audio_batch`: zero-pad to the longest clip, stack) do on the host,
// done from the decoder's uint8 frames after ONE 1-byte-per-sample host->device copy.  All three kernels are pure
// HBM streams (byte/integer work, bit-exact against the oracle); a clip is an entry of a DEVICE table {pointer, length}.
#include "common.h"

namespace {

#define GSTRIDE(i, n) for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// frame / 255 is a float64 division in numpy, rounded to float32 by `.float()` (pig/data.py:68): 256 possible values,
// built once per workgroup with the same two IEEE operations.
__device__ __forceinline__ void build_lut(float* lut) {
  for (int v = threadIdx.x; v < 256; v += blockDim.x) lut[v] = (float)((double)v / 255.0);
  __syncthreads();
}

// ragged uint8 [T_i][H][W][3] -> fp32 [n][3][Tmax][H][W], zero frames after T_i.  One thread = 4 pixels (12 source
// bytes as 3 dwords, one float4 per colour plane); grid.y = clip.
__global__ __launch_bounds__(256) void collate_video_kernel(const long long* __restrict__ items, int Tmax, long long hw,
                                                            float* __restrict__ out) {
  __shared__ float lut[256];
  build_lut(lut);
  const int b = blockIdx.y;
  const uint8_t* src = (const uint8_t*)items[2 * b];
  const long long npix = items[2 * b + 1] * hw, plane = (long long)Tmax * hw;
  float* o = out + (long long)b * 3 * plane;
  const bool vec = (((uintptr_t)src) & 3) == 0 && (hw & 3) == 0;
  if (vec) {
    GSTRIDE(q, plane / 4) {
      const long long p = q * 4;
      float4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
      if (p < npix) {                       // npix is a multiple of 4 here: a group never straddles the clip's end
        const uint32_t* s3 = (const uint32_t*)(src + p * 3);
        const uint32_t w0 = s3[0], w1 = s3[1], w2 = s3[2];   // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        c0 = {lut[w0 & 255], lut[w0 >> 24], lut[(w1 >> 16) & 255], lut[(w2 >> 8) & 255]};
        c1 = {lut[(w0 >> 8) & 255], lut[w1 & 255], lut[w1 >> 24], lut[(w2 >> 16) & 255]};
        c2 = {lut[(w0 >> 16) & 255], lut[(w1 >> 8) & 255], lut[w2 & 255], lut[w2 >> 24]};
      }
      *(float4*)(o + p) = c0;
      *(float4*)(o + plane + p) = c1;
      *(float4*)(o + 2 * plane + p) = c2;
    }
  } else {
    GSTRIDE(p, plane) {
      const bool in = p < npix;
      o[p] = in ? lut[src[p * 3]] : 0.f;
      o[plane + p] = in ? lut[src[p * 3 + 1]] : 0.f;
      o[2 * plane + p] = in ? lut[src[p * 3 + 2]] : 0.f;
    }
  }
}

// ragged byte rows -> [n][row_bytes], zero after each row's own length (padded uint8 video, padded fp32 audio)
__global__ __launch_bounds__(256) void collate_rows_kernel(const long long* __restrict__ items, long long row_bytes,
                                                           uint8_t* __restrict__ out) {
  const int b = blockIdx.y;
  const uint8_t* src = (const uint8_t*)items[2 * b];
  const long long len = items[2 * b + 1];
  uint8_t* o = out + (long long)b * row_bytes;
  const bool vec = ((((uintptr_t)src) | (uintptr_t)o | (uintptr_t)row_bytes) & 15) == 0;
  if (vec) {
    GSTRIDE(q, row_bytes / 16) {
      const long long p = q * 16;
      uint4 v = {0, 0, 0, 0};
      if (p + 16 <= len) {
        v = *(const uint4*)(src + p);
      } else if (p < len) {                 // the one group that straddles the end
        uint8_t t[16];
        for (int k = 0; k < 16; ++k) t[k] = p + k < len ? src[p + k] : 0;
        v = *(const uint4*)t;
      }
      *(uint4*)(o + p) = v;
    }
  } else {
    GSTRIDE(p, row_bytes) o[p] = p < len ? src[p] : 0;
  }
}

// uint8 [B][T][H][W][3] -> normalised bf16 [B*T*H*W][8] (channels 3..7 zero): the stem's input, same arithmetic as
// video_norm_kernel on the fp32 batch ((x/255 - mean) * (1/std), RNE to bf16), so both routes give identical bits.
template <int CPP>
__global__ __launch_bounds__(256) void video_norm_u8_kernel(const uint8_t* __restrict__ x, h16raw* __restrict__ out,
                                                            long long npos, float m0, float m1, float m2, float i0,
                                                            float i1, float i2) {
  __shared__ float lut[256];
  build_lut(lut);
  if ((npos & 3) == 0) {
    GSTRIDE(q, npos / 4) {
      const uint32_t* s3 = (const uint32_t*)(x + q * 12);
      const uint32_t w0 = s3[0], w1 = s3[1], w2 = s3[2];
      const uint32_t px[4][3] = {{w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255},
                                 {w0 >> 24, w1 & 255, (w1 >> 8) & 255},
                                 {(w1 >> 16) & 255, w1 >> 24, w2 & 255},
                                 {(w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24}};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f[8] = {(lut[px[k][0]] - m0) * i0, (lut[px[k][1]] - m1) * i1, (lut[px[k][2]] - m2) * i2, 0, 0, 0, 0, 0};
        const uint4 v = pack8(f);
        if (CPP == 8) *(uint4*)(out + (q * 4 + k) * 8) = v;
        else *(uint2*)(out + (q * 4 + k) * 4) = make_uint2(v.x, v.y);
      }
    }
  } else {
    GSTRIDE(i, npos) {
      float f[8] = {(lut[x[i * 3]] - m0) * i0, (lut[x[i * 3 + 1]] - m1) * i1, (lut[x[i * 3 + 2]] - m2) * i2,
                    0, 0, 0, 0, 0};
      const uint4 v = pack8(f);
      if (CPP == 8) *(uint4*)(out + i * 8) = v;
      else *(uint2*)(out + i * 4) = make_uint2(v.x, v.y);
    }
  }
}

inline int grid_x(long long work, int n) {      // >= 8 workgroups per CU over the whole launch, <= 4096 per clip
  long long b = (work + 255) / 256;
  const long long cap = n >= 8 ? 512 : 4096;
  if (b > cap) b = cap;
  return b < 1 ? 1 : (int)b;
}
}  // namespace

#define S_ ((hipStream_t)s)

extern "C" int pp_collate_video_u8(const void* items, int n, int Tmax, int H, int W, float* out, pp_stream_t s) {
  PP_CHECK_ARG(items && out && n > 0 && Tmax > 0 && H > 0 && W > 0, "pp_collate_video_u8: sizes");
  const long long hw = (long long)H * W;
  hipLaunchKernelGGL(collate_video_kernel, dim3(grid_x(Tmax * hw / 4 + 1, n), n), dim3(256), 0, S_,
                     (const long long*)items, Tmax, hw, out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_collate_rows(const void* items, int n, long long row_bytes, void* out, pp_stream_t s) {
  PP_CHECK_ARG(items && out && n > 0 && row_bytes > 0, "pp_collate_rows: sizes");
  hipLaunchKernelGGL(collate_rows_kernel, dim3(grid_x(row_bytes / 16 + 1, n), n), dim3(256), 0, S_,
                     (const long long*)items, row_bytes, (uint8_t*)out);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

extern "C" int pp_video_normalize_u8_ndhwc(const void* x, void* out, int B, int T, int H, int W, const float* mean3,
                                           const float* std3, pp_stream_t s) {
  PP_CHECK_ARG(x && out && B > 0 && T > 0 && H > 0 && W > 0 && mean3 && std3, "pp_video_normalize_u8_ndhwc: sizes");
  PP_CHECK_ARG((((uintptr_t)x) & 3) == 0, "pp_video_normalize_u8_ndhwc: x must be 4-byte aligned");
  const long long npos = (long long)B * T * H * W;
  hipLaunchKernelGGL(video_norm_u8_kernel<8>, dim3(grid_x(npos / 4 + 1, 1)), dim3(256), 0, S_, (const uint8_t*)x,
                     (h16raw*)out, npos, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
extern "C" int pp_video_normalize_u8_ndhwc4(const void* x, void* out, int B, int T, int H, int W, const float* mean3,
                                            const float* std3, pp_stream_t s) {
  PP_CHECK_ARG(x && out && B > 0 && T > 0 && H > 0 && W > 0 && mean3 && std3, "pp_video_normalize_u8_ndhwc4: sizes");
  PP_CHECK_ARG((((uintptr_t)x) & 3) == 0, "pp_video_normalize_u8_ndhwc4: x must be 4-byte aligned");
  const long long npos = (long long)B * T * H * W;
  hipLaunchKernelGGL(video_norm_u8_kernel<4>, dim3(grid_x(npos / 4 + 1, 1)), dim3(256), 0, S_, (const uint8_t*)x,
                     (h16raw*)out, npos, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  PP_LAUNCH_CHECK();
  return PP_OK;
}
