// Micro-benchmark (not part of the library): which vector instructions run in the shadow of an MFMA issued by the SAME wave
// on gfx950, and what does each cost alone?  One loop iteration = NM MFMAs, each followed by NV copies of one vector op
// (independent registers).  Variants: MFMA only, op only, both.  Launched with 4 waves (one per SIMD) and 8 waves (two).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/shadow.hip -o gpurun_out/shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Op { FMA, PKFMA, EXP, CVT, MAX3, ADD, PKADD, PKMUL, MOV, LDS128, LDSTR, NOP_ };

template <int OP>
__device__ __forceinline__ void vop(float (&v)[16], f32x2 (&w)[8], unsigned (&u)[8], f32x4& l, const int i, const unsigned addr) {
  const int r = i & 7;
  if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(v[8]), "v"(v[9]));
  if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[r]) : "v"(w[(r + 1) & 7]), "v"(w[(r + 2) & 7]));
  if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
  if (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(v[r]), "v"(v[r + 8]));
  if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(v[8]), "v"(v[9]));
  if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[8]));
  if (OP == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[r]) : "v"(w[(r + 1) & 7]));
  if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[r]) : "v"(w[(r + 1) & 7]));
  if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
  if (OP == LDS128) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(addr));
  if (OP == LDSTR) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(w[r]) : "v"(addr));
}

template <int OP, int NV, bool WITH_MFMA, bool WITH_OP>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  f32x16 acc0 = {0}, acc1 = {0};
  float v[16];
  f32x2 w[8];
  unsigned u[8];
  f32x4 l = {0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
  for (int i = 0; i < 8; ++i) { w[i] = f32x2{v[i], v[i + 8]}; u[i] = threadIdx.x + i; }
  const unsigned addr = (unsigned)(uintptr_t)lds + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (WITH_MFMA) {
        if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      }
      if (WITH_OP) {
#pragma unroll
        for (int i = 0; i < NV; ++i) vop<OP>(v, w, u, l, j * NV + i, addr);
      }
    }
    if (OP == LDS128 || OP == LDSTR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = l[0] + l[1] + l[2] + l[3];
  for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += w[i][0] + w[i][1] + (float)u[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int OP, int NV, bool M, bool O>
float run(float* out, int iters, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, NV, M, O>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, NV, M, O>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int OP, int NV>
void row(float* out, const char* name) {
  const int iters = 20000;
  for (int threads = 256; threads <= 512; threads += 256) {
    const float m = run<OP, NV, true, false>(out, iters, threads);
    const float o = run<OP, NV, false, true>(out, iters, threads);
    const float b = run<OP, NV, true, true>(out, iters, threads);
    const double n_m = 8.0 * iters, n_o = 8.0 * NV * iters;
    printf("%-22s NV=%d waves/SIMD=%d | mfma only %6.2f ns/mfma | op only %5.2f ns/op | both %7.3f ms = mfma + %5.2f ns/op (hidden %3.0f%% of the cheaper)\n",
           name, NV, threads / 256, m * 1e6 / n_m, o * 1e6 / n_o, b, (b - m) * 1e6 / n_o,
           100.0 * (m + o - b) / (m < o ? m : o));
  }
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  row<FMA, 8>(out, "v_fma_f32");
  row<PKFMA, 8>(out, "v_pk_fma_f32");
  row<PKADD, 8>(out, "v_pk_add_f32");
  row<PKMUL, 8>(out, "v_pk_mul_f32");
  row<EXP, 8>(out, "v_exp_f32");
  row<CVT, 8>(out, "v_cvt_pk_bf16_f32");
  row<MAX3, 8>(out, "v_max3_f32");
  row<ADD, 8>(out, "v_add_f32");
  row<MOV, 8>(out, "v_mov_b32");
  row<LDS128, 2>(out, "ds_read_b128");
  row<LDSTR, 4>(out, "ds_read_b64_tr_b16");
  row<FMA, 4>(out, "v_fma_f32");
  row<FMA, 16>(out, "v_fma_f32");
  return 0;
}
