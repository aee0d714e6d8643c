// Micro-benchmark (not part of the library): do MFMA and VALU work overlap on one gfx950 SIMD
//   (a) across two co-resident waves, (b) inside one wave's instruction stream?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/overlap.hip -o gpurun_out/overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

// mode bits per wave role: 1 = MFMA stream, 2 = VALU stream (fma + exp mix like a softmax), 3 = both interleaved in ONE wave
template <int MODE_LO, int MODE_HI>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const int mode = wave < 4 ? MODE_LO : MODE_HI;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  f32x16 acc0 = {0}, acc1 = {0};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  const float c = 0.3f, m = 0.7f;
  if (mode == 0) { out[blockIdx.x * 512 + threadIdx.x] = 0; return; }
  for (int it = 0; it < iters; ++it) {
    if (mode == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { MFMA(acc0, a, b); MFMA(acc1, a, b); }
    } else if (mode == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(fmaf(v[i], c, -m)) + v[(i + 1) & 15];
    } else {
      // 16 MFMAs with 6 x (fma, exp, add) = 18 VALU-class ops hand-placed behind each one (same totals as mode 1 + mode 2)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j & 1) MFMA(acc1, a, b); else MFMA(acc0, a, b);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int r = (j * 6 + i) & 15;
          v[r] = __builtin_amdgcn_exp2f(fmaf(v[r], c, -m)) + v[(r + 1) & 15];
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int LO, int HI>
float run(float* out, int iters, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<LO, HI>), dim3(256), dim3(512), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<LO, HI>), dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.3f ms\n", name, ms);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  run<1, 0>(out, iters, "waves 0-3: 16 MFMA/iter            | waves 4-7: idle");
  run<2, 0>(out, iters, "waves 0-3: 96 fma+96 exp+96 add    | waves 4-7: idle");
  run<1, 2>(out, iters, "waves 0-3: MFMA                    | waves 4-7: VALU  (two waves per SIMD)");
  run<1, 1>(out, iters, "waves 0-3: MFMA                    | waves 4-7: MFMA");
  run<2, 2>(out, iters, "waves 0-3: VALU                    | waves 4-7: VALU");
  run<3, 0>(out, iters, "waves 0-3: MFMA+VALU interleaved   | waves 4-7: idle  (one wave per SIMD)");
  run<3, 3>(out, iters, "waves 0-3: interleaved             | waves 4-7: interleaved");
  return 0;
}
