// Micro-benchmark behind DESIGN.md section 4's "chip-wide fill rate" question: how many bytes per second can ALL CUs pull
// from L2-resident operand panels (the access pattern of the 256x256x64 GEMM tile: 64 KiB per K-tile and workgroup), by
// LDS-DMA or by plain global loads into VGPRs, alone or while the matrix pipes are busy?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fill_bench.hip -o gpurun_out/fill_bench && gpurun_out/fill_bench
// Diagnostic tool, not part of the library.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 64;

__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_dst) : "memory");
}

// MODE bit 0: LDS-DMA fill; bit 1: VGPR fill; bit 2: MFMA work (64 MFMAs per wave and K-tile = the GEMM's rate) on register
// operands; bits 2+3: the MFMAs take their operands from the filled LDS tile (the GEMM's real inner loop, simple schedule)
template <int MODE>
__global__ __launch_bounds__(512, 2) void fill_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int64_t ld,
                                                      int nk, int tiles_m, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // same XCD-aware tile order idea as the GEMM: workgroups of one XCD take neighbouring tiles
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int id = (bid & 7) * (nwg >> 3) + (bid >> 3);
  const int tiles_n = nwg / tiles_m, width = 8 * tiles_n, group = id / width, first_m = group * 8;
  const int gsz = min(tiles_m - first_m, 8), in_group = id - group * width;
  const int tm = first_m + in_group % gsz, tn = in_group / gsz;          // 8-tall column groups inside an XCD's chunk
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)smem;
  const unsigned ld2 = (unsigned)(ld * 2);
  // 8 rows x 128 B per wave-instruction; 16-byte chunk c of row r lands at chunk c ^ ((r >> 1) & 7) (the GEMM's image)
  const unsigned voff = (unsigned)(lane >> 3) * ld2 + (unsigned)(((lane & 7) ^ (((wave & 1) << 2) + (lane >> 4))) << 4);
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa, fb;
#pragma unroll
  for (int e = 0; e < 8; ++e) { fa[e] = (short)(lane * 37 + e); fb[e] = (short)(lane * 11 + e); }
  bf16x8 keep = fa;
  bf16x8 st[2][8];   // VGPR mode: two K-tiles in flight
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned dst = lds0 + (kt & 1) * 65536;
    // this wave's 8 pieces of the K-tile: 4 of the A panel (rows tm*256 + ...), 4 of the B panel
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = wave + (j & 3) * 8;                      // 32 pieces of 8 rows per panel
      const uint16_t* base = (j < 4 ? A + (int64_t)(tm * 256 + piece * 8) * ld : B + (int64_t)(tn * 256 + piece * 8) * ld) + kt * BK;
      if (MODE & 1) glds16((const char*)base, voff, dst + (j < 4 ? 0 : 32768) + piece * 1024);
      if (MODE & 2) st[kt & 1][j] = *(const bf16x8*)((const char*)base + voff);
    }
    if ((MODE & 12) == 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
    }
    if (MODE & 16) {           // round 3: the GEMM's 24 fragment reads per wave and K-tile from a resident tile, NO fill; bit 3: + its 64 MFMAs
      const char* t = smem + (kt & 1) * 65536;
      const int gp = wave >> 2, wc = wave & 3;
      auto frag = [&](const char* base, int row, int chunk) {
        return *(const bf16x8*)(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      };
      bf16x8 bF[2][2][2];
#pragma unroll
      for (int jh = 0; jh < 2; ++jh)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int sk = 0; sk < 2; ++sk)
            bF[jh][j][sk] = frag(t + 32768, jh * 128 + wc * 32 + j * 16 + (lane & 15), sk * 4 + (lane >> 4));
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        bf16x8 aF[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int sk = 0; sk < 2; ++sk) aF[i][sk] = frag(t, ih * 128 + gp * 64 + i * 16 + (lane & 15), sk * 4 + (lane >> 4));
        if (MODE & 8) {
#pragma unroll
          for (int jh = 0; jh < 2; ++jh)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  acc2[ih][jh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bF[jh][j][sk], aF[i][sk], acc2[ih][jh][i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) keep ^= aF[i][sk];
        }
      }
      if (!(MODE & 8)) {
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) keep ^= bF[jh][j][sk];
      }
      __builtin_amdgcn_s_barrier();
      continue;
    }
    if ((MODE & 12) == 12) {   // K-tile kt-1 (issued one iteration ago) has landed for every wave before anyone reads it
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if ((MODE & 12) == 12 && kt > 0) {
      // the GEMM's real inner loop on the K-tile that landed one iteration ago: 8 waves = 2 row groups x 4 column groups,
      // a wave owns rows {ih*128 + gp*64 + 16i} x columns {jh*128 + wc*32 + 16j}; operands read from LDS (ds_read_b128)
      const char* t = smem + ((kt - 1) & 1) * 65536;
      const int gp = wave >> 2, wc = wave & 3;
      auto frag = [&](const char* base, int row, int chunk) {
        return *(const bf16x8*)(base + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      };
      bf16x8 bF[2][2][2];
#pragma unroll
      for (int jh = 0; jh < 2; ++jh)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int sk = 0; sk < 2; ++sk)
            bF[jh][j][sk] = frag(t + 32768, jh * 128 + wc * 32 + j * 16 + (lane & 15), sk * 4 + (lane >> 4));
#pragma unroll
      for (int ih = 0; ih < 2; ++ih) {
        bf16x8 aF[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int sk = 0; sk < 2; ++sk) aF[i][sk] = frag(t, ih * 128 + gp * 64 + i * 16 + (lane & 15), sk * 4 + (lane >> 4));
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
#pragma unroll
          for (int sk = 0; sk < 2; ++sk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc2[ih][jh][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bF[jh][j][sk], aF[i][sk], acc2[ih][jh][i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_barrier();   // every wave is done reading this slot before the next iteration's LDS-DMA refills it
    }
    if ((MODE & 1) && (MODE & 12) != 12) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // one K-tile (8 instructions) stays in flight
    if (MODE & 2) {
      if (kt > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) keep ^= st[(kt - 1) & 1][j];   // consumes the previous K-tile (the compiler waits exactly)
      }
    }
    if ((MODE & 12) != 12) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc2[a][b][i][j][1];
  s += (float)keep[0] + (float)keep[7];
  if (MODE & 1) s += (float)smem[tid * 16];
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
double run(const uint16_t* A, const uint16_t* B, int64_t ld, int nk, int tiles_m, int tiles_n, float* sink, int iters, int threads = 512) {
  auto kern = fill_kernel<MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(threads), 131072, 0, A, B, ld, nk, tiles_m, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(threads), 131072, 0, A, B, ld, nk, tiles_m, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  const int K = 8192, rows = 8192;
  const int64_t ld = K;
  std::vector<uint16_t> h((size_t)rows * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3f80 + (i * 2654435761u >> 25));
  uint16_t *A, *B; float* sink;
  CK(hipMalloc(&A, h.size() * 2)); CK(hipMalloc(&B, h.size() * 2)); CK(hipMalloc(&sink, 16));
  CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int nk = K / BK;
  printf("fill of 256x256x64 operand tiles, K = %d (128 K-tiles of 64 KiB per workgroup), 512 threads, 1 workgroup per CU\n", K);
  printf("%-10s %-26s %10s %12s %12s\n", "grid", "mode", "ms", "fill TB/s", "MFMA TF/s");
  for (int tiles_m : {32, 20}) {
    const int tiles_n = tiles_m == 32 ? 8 : 8;            // 256 workgroups (every CU) and 160 (the N = 2048 GEMMs of the path)
    const double bytes = (double)tiles_m * tiles_n * nk * 65536.0;
    const double flops = (double)tiles_m * tiles_n * nk * 8 * 64 * 2.0 * 16 * 16 * 32;   // 64 MFMAs per wave per K-tile
    struct { const char* name; double ms; bool mfma; } res[] = {
        {"LDS-DMA only", run<1>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), false},
        {"VGPR loads only", run<2>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), false},
        {"MFMA only", run<4>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), true},
        {"LDS-DMA + MFMA", run<5>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), true},
        {"VGPR loads + MFMA", run<6>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), true},
        {"LDS-DMA + LDS reads + MFMA", run<13>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), true},
        {"LDS reads only (no fill)", run<16>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), false},
        {"LDS reads + MFMA (no fill)", run<24>(A, B, ld, nk, tiles_m, tiles_n, sink, 20), true},
        {"MFMA only, 4 waves (1/SIMD)", run<4>(A, B, ld, nk, tiles_m, tiles_n, sink, 20, 256), true},     // half the MFMAs of the 8-wave rows: TF/s column x 0.5
    };
    for (auto& r : res)
      printf("%-10d %-26s %10.4f %12.2f %12.1f\n", tiles_m * tiles_n, r.name, r.ms,
             (r.name[0] == 'M' && r.name[1] == 'F') ? 0.0 : (r.name[4] == 'r' ? 3.0 : 1.0) * bytes / (r.ms * 1e-3) / 1e12,   // "LDS reads": 192 KiB of fragment reads per K-tile
             r.mfma ? flops / (r.ms * 1e-3) / 1e12 : 0.0);
  }
  return 0;
}
