// Standalone probe (GPU box): empirical lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950.
//   1. which (row, k-block) a lane's E8M0 scale byte applies to, per opsel
//   2. accumulation precision inside one 128-deep MFMA
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o gpurun_out/probe_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int OPSEL>
__global__ void k_scale(const int* sa, const int* sb, float* d, int which) {
  int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  v4f acc = {0, 0, 0, 0};
  if (which == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPSEL, sa[l], 0, sb[l]);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa[l], OPSEL, sb[l]);
  for (int j = 0; j < 4; ++j) d[l * 4 + j] = acc[j];
}

__global__ void k_data(const v8i* a, const v8i* b, float* d) {
  int l = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int j = 0; j < 4; ++j) d[l * 4 + j] = acc[j];
}

__global__ void k_data_s(const v8i* a, const v8i* b, const int* sa, const int* sb, float* d) {
  int l = threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int j = 0; j < 4; ++j) d[l * 4 + j] = acc[j];
}

static void run_scale(int opsel, int which, int* dsa, int* dsb, float* dd, float* hd) {
  switch (opsel) {
    case 0: hipLaunchKernelGGL(k_scale<0>, 1, 64, 0, 0, dsa, dsb, dd, which); break;
    case 1: hipLaunchKernelGGL(k_scale<1>, 1, 64, 0, 0, dsa, dsb, dd, which); break;
    case 2: hipLaunchKernelGGL(k_scale<2>, 1, 64, 0, 0, dsa, dsb, dd, which); break;
    case 3: hipLaunchKernelGGL(k_scale<3>, 1, 64, 0, 0, dsa, dsb, dd, which); break;
  }
  hipMemcpy(hd, dd, 256 * 4, hipMemcpyDeviceToHost);
}

// fp8 e4m3 encode of a power of two 2^e (e in [-6, 8])
static uint8_t p2(int e) { return (uint8_t)((e + 7) << 3); }

int main() {
  int *dsa, *dsb; float* dd; v8i *da, *db;
  hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 1024); hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32);
  int hsa[64], hsb[64]; float hd[256];
  // ---- 1. scale map.  D layout: lane l, reg j -> row(first operand index) = 4*(l>>4)+j, col = l&15
  for (int which = 0; which < 2; ++which)
    for (int opsel = 0; opsel < 4; ++opsel)
      for (int by = 0; by < 4; ++by) {
        printf("SCALE operand=%c opsel=%d byte=%d:", which ? 'B' : 'A', opsel, by);
        int hits = 0;
        for (int L = 0; L < 64; ++L) {
          for (int i = 0; i < 64; ++i) { hsa[i] = 0x7f7f7f7f; hsb[i] = 0x7f7f7f7f; }
          int* tgt = which ? hsb : hsa;
          tgt[L] = (0x7f7f7f7f & ~(0xff << (8 * by))) | (0x80 << (8 * by));
          hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
          run_scale(opsel, which, dsa, dsb, dd, hd);
          // find changed elements
          int nchg = 0, r0 = -1, c0 = -1; float delta = 0; int rowmask = 0, colmask = 0;
          for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            float v = hd[l * 4 + j];
            if (v != 128.0f) { nchg++; r0 = 4 * (l >> 4) + j; c0 = l & 15; delta = v - 128.0f; rowmask |= 1 << r0; colmask |= 1 << c0; }
          }
          if (nchg) { hits++; if (L < 64 && (L % 16 < 2 || L % 16 == 15)) printf(" [L%d: n=%d rows=%04x cols=%04x d=%g]", L, nchg, rowmask, colmask, delta); }
        }
        printf(" hits=%d\n", hits);
      }
  // ---- 2. data map: A one-hot at (lane LA, byte BA) = 1.0; B all 1.0 -> tells which output row lane LA feeds;
  //         and B one-hot pairing: A (LA,BA)=1, B (LB,BB)=1 -> D != 0 iff paired.
  {
    uint8_t ha[64 * 32], hb[64 * 32];
    printf("DATA pairing (A lane/byte -> B lane-group/byte that pairs), sample:\n");
    for (int LA : {0, 1, 17, 35, 63}) for (int BA : {0, 5, 16, 31}) {
      memset(ha, 0, sizeof ha); ha[LA * 32 + BA] = 0x38;
      // B all ones first: which row
      memset(hb, 0x38, sizeof hb);
      hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_data, 1, 64, 0, 0, da, db, dd); hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
      int rowmask = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (hd[l * 4 + j] != 0) rowmask |= 1 << (4 * (l >> 4) + j);
      // find pairing in B col 0 lanes (lanes 0,16,32,48) x 32 bytes
      int found = -1;
      for (int q = 0; q < 4 && found < 0; ++q) for (int bb = 0; bb < 32 && found < 0; ++bb) {
        memset(hb, 0, sizeof hb); hb[(q * 16) * 32 + bb] = 0x38;
        hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_data, 1, 64, 0, 0, da, db, dd); hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; ++i) if (hd[i] != 0) found = q * 32 + bb;
      }
      printf("  A(lane %d, byte %d): rows=%04x pairs with B(q=%d, byte=%d)\n", LA, BA, rowmask, found / 32, found % 32);
    }
  }
  // ---- 2b. which scale lane covers which data region (lane group q, 16-byte half h) of row 0
  {
    uint8_t ha[64 * 32], hb[64 * 32];
    memset(hb, 0x38, sizeof hb);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    for (int qa = 0; qa < 4; ++qa) for (int h = 0; h < 2; ++h) {
      memset(ha, 0, sizeof ha);
      for (int b = 0; b < 16; ++b) ha[(qa * 16) * 32 + 16 * h + b] = 0x38;
      hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
      printf("REGION A(q=%d, half=%d) scaled by A-scale lanes:", qa, h);
      for (int Ls = 0; Ls < 64; Ls += 16) {
        for (int i = 0; i < 64; ++i) { hsa[i] = 0x7f7f7f7f; hsb[i] = 0x7f7f7f7f; }
        hsa[Ls] = 0x7f7f7f80;
        hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_data_s, 1, 64, 0, 0, da, db, dsa, dsb, dd); hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        printf(" L%d->%g", Ls, hd[0]);
      }
      printf("\n");
    }
  }
  // ---- 3. accumulation precision: one product 2^E big, 127 products 2^s small (exact sum known)
  {
    uint8_t ha[64 * 32], hb[64 * 32];
    for (int big = 16; big >= 8; big -= 8) for (int s = big - 8; s >= big - 30 && s >= -12; --s) {
      // row 0 of A: k=0 -> 2^(big/2), others 2^sa ; col 0 of B: k=0 -> 2^(big/2), others 2^sb ; s = sa+sb
      int sa = s / 2, sb2 = s - sa;
      if (sa < -6 || sb2 < -6 || sa > 8 || sb2 > 8) continue;
      memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
      for (int q = 0; q < 4; ++q) for (int b = 0; b < 32; ++b) {
        ha[(q * 16 + 0) * 32 + b] = p2(sa); hb[(q * 16 + 0) * 32 + b] = p2(sb2);
      }
      ha[0] = p2(big / 2); hb[0] = p2(big / 2);
      hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_data, 1, 64, 0, 0, da, db, dd); hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
      double exact = ldexp(1.0, big) + 127.0 * ldexp(1.0, s);
      printf("ACC big=2^%d small=2^%d x127: got %.10g exact %.10g (fp32-rounded exact %.10g) diff %.6g\n", big, s, (double)hd[0], exact,
             (double)(float)exact, (double)hd[0] - exact);
    }
  }
  hipError_t e = hipDeviceSynchronize();
  printf("done: %s\n", hipGetErrorString(e));
  return 0;
}
