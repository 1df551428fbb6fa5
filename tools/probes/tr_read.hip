#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint32_t h_addr[64]; uint16_t h_out[256];
  uint32_t* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) h_addr[l] = l * 8;                       // contiguous 8 B per lane
      else if (variant == 1) h_addr[l] = (l / 4) * 64 + (l % 4) * 8;  // 4 lanes per 64-byte row
      else h_addr[l] = ((l % 16) / 4) * 64 + (l % 4) * 8 + (l / 16) * 1024;  // per 16-lane group: 4 rows x 32 B, groups far apart
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4u(elem %4u): %4u %4u %4u %4u\n", l, h_addr[l], h_addr[l] / 2, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]);
  }
  return 0;
}
