#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x * 2] = r[0];
  out[threadIdx.x * 2 + 1] = r[1];
}
__global__ void k2(unsigned* out) {  // two back-to-back swaps on freshly written registers
  unsigned a = threadIdx.x * 3, b = 100 + threadIdx.x, c = 200 + threadIdx.x, d = 300 + threadIdx.x;
  asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n s_nop 0\n"
               "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n s_nop 3"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1000u));
  out[threadIdx.x * 4] = a; out[threadIdx.x * 4 + 1] = b; out[threadIdx.x * 4 + 2] = c; out[threadIdx.x * 4 + 3] = d;
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, 1024);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("swap(a=lane, b=100+lane):\n");
  for (int l = 0; l < 64; ++l) printf("%s[%2d]=(%3u,%3u)%s", l % 8 == 0 ? " " : " ", l, h[2 * l], h[2 * l + 1], l % 8 == 7 ? "\n" : "");
  hipLaunchKernelGGL(k2, 1, 64, 0, 0, d); hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  printf("k2 (a=3l+1000,b=1100+l | c=1200+l,d=1300+l; swap(a,c), swap(b,d)):\n");
  for (int l = 0; l < 64; ++l) printf(" [%2d]=(%4u,%4u,%4u,%4u)%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], l % 4 == 3 ? "\n" : "");
  return 0;
}
