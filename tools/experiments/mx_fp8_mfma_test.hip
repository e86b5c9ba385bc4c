#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// C[16][16] = sum_k A[m][k] * B[n][k], A and B e4m3 bytes [16][128], scaled by 2^-11 * 2^-8
__global__ void k(const unsigned char* A, const unsigned char* B, float* C, int sa, int sb) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  v8i a, b;
  // lane (r, g): bytes of row r: chunk g (16 B) and chunk 4 + g (16 B)  -- the (s, g) fragment mapping of the GEMM kernels
  const int4* ar = (const int4*)(A + r * 128);
  const int4* br = (const int4*)(B + r * 128);
  int4 a0 = ar[g], a1 = ar[4 + g], b0 = br[g], b1 = br[4 + g];
  a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
  b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  // operands swapped as in the GEMM kernels: D[row = n][col = m]
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, sb, 0, sa);
  for (int q = 0; q < 4; ++q) C[r * 16 + 4 * g + q] = c[q];     // C[m = r][n = 4g + q]
}
static float e4m3_to_f(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
int main() {
  unsigned char hA[16 * 128], hB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = rand() % 120 + (rand() & 1) * 128; hB[i] = rand() % 120 + (rand() & 1) * 128; }
  unsigned char *dA, *dB; float* dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, 127 - 11, 127 - 8);
  float hC[256]; hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  double worst = 0, big = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double s = 0; for (int kk = 0; kk < 128; ++kk) s += (double)e4m3_to_f(hA[m * 128 + kk]) * e4m3_to_f(hB[n * 128 + kk]);
    s *= ldexp(1.0, -19);
    worst = fmax(worst, fabs(s - hC[m * 16 + n])); big = fmax(big, fabs(s));
  }
  printf("max |err| %.3e of max |ref| %.3e  (C[0][0] = %g)\n", worst, big, hC[0]);
  return worst < 1e-5 * big ? 0 : 1;
}
