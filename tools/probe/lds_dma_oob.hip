// Probe: does buffer_load ... lds (LDS-DMA) write zeros for out-of-range lanes, or leave LDS untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xAAAAAAAAu;   // poison
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, (int)0xFFFFFFF0, 0x00020000);
  unsigned off = (threadIdx.x & 1) ? 0xFFFFFFF0u : threadIdx.x * 16u;   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  unsigned h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane0: %x %x %x %x | lane1 (OOB): %x %x %x %x | lane2: %x\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
  return 0;
}
