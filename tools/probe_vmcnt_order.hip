// Does a wave's LOAD retire (for s_waitcnt vmcnt) before an OLDER, slower STORE of the same wave?  gfx950, one wave.
//   case A: load (L2-hot line) alone, wait vmcnt(0)                      -> load latency
//   case B: store to a cold line, wait vmcnt(0)                          -> store acknowledge latency (sc1 / plain / nt)
//   case C: store to a cold line, THEN the hot load, wait vmcnt(1)       -> if ~A: the load retired past the store (out of order
//           between the kinds); if ~B: in-order retirement, the load's wait is the store's
// hipcc -O3 --offload-arch=gfx950 tools/probe_vmcnt_order.hip -o tools/bin/probe_vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void probe(const int* hot, int* cold, unsigned long long* out, int stride_ints) {
  const int lane = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs_hot = __builtin_amdgcn_make_buffer_rsrc((void*)hot, 0, 1 << 20, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_cold = __builtin_amdgcn_make_buffer_rsrc((void*)cold, 0, 0x7fffffff, 0x00020000);
  int sink = 0;
  for (int w = 0; w < 4; ++w) sink += __builtin_amdgcn_raw_buffer_load_b32(rs_hot, lane * 4, 0, 0);  // warm the line into L2 / L1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t[8];
  for (int rep = 0; rep < 4; ++rep) {
    const int cold_off = ((rep * 64 + lane) * stride_ints) * 4;  // every lane its own far-apart cold line
    const v4i data = {lane, rep, 3, 4};
    // A
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[0] = __builtin_amdgcn_s_memtime();
    int a = __builtin_amdgcn_raw_buffer_load_b32(rs_hot, lane * 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a)::"memory");
    t[1] = __builtin_amdgcn_s_memtime();
    // B
    __builtin_amdgcn_raw_buffer_store_b128(data, rs_cold, cold_off, 0, AUX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[2] = __builtin_amdgcn_s_memtime();
    // C
    __builtin_amdgcn_raw_buffer_store_b128(data, rs_cold, cold_off + (1 << 26), 0, AUX);
    int c = __builtin_amdgcn_raw_buffer_load_b32(rs_hot, lane * 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(c)::"memory");
    t[3] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[4] = __builtin_amdgcn_s_memtime();
    sink += a + c;
    if (lane == 0) {
      out[rep * 4 + 0] = t[1] - t[0];
      out[rep * 4 + 1] = t[2] - t[1];
      out[rep * 4 + 2] = t[3] - t[2];
      out[rep * 4 + 3] = t[4] - t[3];
    }
  }
  if (sink == 123456789) out[63] = sink;
}
int main() {
  int *hot, *cold;
  unsigned long long* out;
  hipMalloc(&hot, 1 << 20);
  hipMalloc(&cold, (size_t)1 << 31);
  hipMalloc(&out, 64 * 8);
  hipMemset(hot, 0, 1 << 20);
  const char* names[3] = {"sc1 (aux 16)", "plain (aux 0)", "nt (aux 2)"};
  for (int v = 0; v < 3; ++v) {
    for (int round = 0; round < 2; ++round) {
      hipMemset(cold, round, (size_t)1 << 31);  // evict everything
      hipDeviceSynchronize();
      if (v == 0) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, hot, cold, out, 4096);
      if (v == 1) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, hot, cold, out, 4096);
      if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, hot, cold, out, 4096);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    for (int rep = 1; rep < 4; ++rep)
      printf("vmcnt order, store %-14s rep %d: A load alone %5llu   B store alone %5llu   C store then load, wait vmcnt(1) %5llu   then vmcnt(0) +%llu  (cycles of s_memtime)\n",
             names[v], rep, h[rep * 4], h[rep * 4 + 1], h[rep * 4 + 2], h[rep * 4 + 3]);
  }
  return 0;
}
