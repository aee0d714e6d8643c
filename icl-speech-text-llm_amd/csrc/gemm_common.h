// gemm_common.h — parameter block and epilogue helpers shared by the GEMM translation units (gemm.hip, gemm256.hip).
#pragma once
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace iclg {


struct GemmParams {
  const __bf16* A;
  const __bf16* W;
  void* C;
  const float* bias;
  const void* R;
  float* ws;
  int64_t lda, ldw, ldc, ldr, sA, sC, sR;
  int M, N, K, epi, out_dtype, res_dtype, split_k, tiles_m, tiles_n;
  int group_m;   // M-tiles per super-tile of the block -> tile map (0 = GROUP_M)
  int xcd_sync;  // block -> tile map variant (block_to_tile)
};

constexpr int GROUP_M = 8;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


// ---- epilogue helpers shared by all tile shapes ---------------------------------------------------------
// v[0..3] = C[m][n..n+3] (4 consecutive n owned by one lane).
__device__ __forceinline__ void epi_store4(const GemmParams& p, int z, int m, int n, f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  const bool has_bias = p.epi & ICL_EPI_BIAS, has_gelu = p.epi & ICL_EPI_GELU, has_res = p.epi & ICL_EPI_RESIDUAL;
  char* Cb = (char*)p.C;
  const int64_t cz = (int64_t)z * p.sC, rz = (int64_t)z * p.sR;
  const bool vec_ok = ((p.ldc & 3) == 0) && (!has_res || (p.ldr & 3) == 0);
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  const bool full = (n + 3 < p.N);
  if (has_bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (full || n + r < p.N) v[r] += p.bias[n + r];
  }
  if (has_gelu) {
    const f32x4 g = gelu_erf4(f32x4{v[0], v[1], v[2], v[3]});
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = g[r];
  }
  const int64_t coff = cz + (int64_t)m * p.ldc + n;
  if (full && vec_ok) {
    if (has_res) {
      const int64_t roff = rz + (int64_t)m * p.ldr + n;
      if (p.res_dtype == ICL_F32) {
        f32x4 rv = *(const f32x4*)((const char*)p.R + roff * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv[r];
      } else {
        const unsigned short* rp = (const unsigned short*)p.R + roff;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bf16_bits_to_f32(rp[r]);
      }
    }
    if (p.out_dtype == ICL_BF16) {
      u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *(u32x2*)(Cb + coff * 2) = pk;
    } else {
      *(f32x4*)(Cb + coff * 4) = f32x4{v[0], v[1], v[2], v[3]};
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r >= p.N) continue;
      float x = v[r];
      if (has_res) {
        const int64_t roff = rz + (int64_t)m * p.ldr + n + r;
        x += (p.res_dtype == ICL_F32) ? ((const float*)p.R)[roff]
                                      : bf16_bits_to_f32(((const unsigned short*)p.R)[roff]);
      }
      if (p.out_dtype == ICL_BF16)
        ((unsigned short*)Cb)[coff + r] = f32_to_bf16_bits(x);
      else
        ((float*)Cb)[coff + r] = x;
    }
  }
}
// gate block at interleaved rows nt + fq4 + r, up block 16 rows later; output column nt/2 + fq4 + r
__device__ __forceinline__ void epi_store_swiglu(const GemmParams& p, int z, int m, int nt, int fq4, f32x4 g4, f32x4 u4) {
  if (m >= p.M || nt >= p.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float g = g4[r], u = u4[r];
    if (p.epi & ICL_EPI_BIAS) {
      g += p.bias[nt + fq4 + r];
      u += p.bias[nt + 16 + fq4 + r];
    }
    v[r] = silu_f(g) * u;
  }
  const int64_t off = (int64_t)z * p.sC + (int64_t)m * p.ldc + (nt >> 1) + fq4;
  if (p.out_dtype == ICL_BF16) {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)((char*)p.C + off * 2) = pk;
  } else {
    *(f32x4*)((char*)p.C + off * 4) = f32x4{v[0], v[1], v[2], v[3]};
  }
}
__device__ __forceinline__ void epi_store_partial(const GemmParams& p, int z, int m, int n, f32x4 acc) {
  if (m >= p.M || n >= p.N) return;
  float* dst = p.ws + (int64_t)z * p.M * p.N + (int64_t)m * p.N + n;
  if (n + 3 < p.N && (p.N & 3) == 0) {
    *(f32x4*)dst = acc;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) dst[r] = acc[r];
  }
}

// ---- interior-tile fast path: no bounds checks, bias as one 16-B load per n-fragment (prefetched before the K loop),
// ---- residual fragments loaded as ONE batch (independent loads in flight together), then add + convert + store.
__device__ __forceinline__ f32x4 load_res4(const GemmParams& p, int64_t roff) {
  if (p.res_dtype == ICL_F32) return *(const f32x4*)((const char*)p.R + roff * 4);
  const u32x2 raw = *(const u32x2*)((const char*)p.R + roff * 2);
  return f32x4{__uint_as_float(raw[0] << 16), __uint_as_float(raw[0] & 0xffff0000u),
               __uint_as_float(raw[1] << 16), __uint_as_float(raw[1] & 0xffff0000u)};
}
__device__ __forceinline__ void store_out4(const GemmParams& p, int64_t coff, f32x4 v) {
  if (p.out_dtype == ICL_BF16) {
    u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *(u32x2*)((char*)p.C + coff * 2) = pk;
  } else {
    *(f32x4*)((char*)p.C + coff * 4) = v;
  }
}
// Tile-INDEPENDENT vector-path predicate: whether a row's arithmetic order (accumulator init = bias + residual) may
// depend only on the problem, never on which tile of the grid the row falls in -> results are batch-invariant.
__device__ __forceinline__ bool vec_path_ok(const GemmParams& p) {
  const bool has_res = p.epi & ICL_EPI_RESIDUAL;
  return ((p.ldc & 3) == 0) && ((p.N & 3) == 0) && (!has_res || (p.ldr & 3) == 0) &&
         (!(p.epi & ICL_EPI_BIAS) || (((uintptr_t)p.bias & 15) == 0));
}
__device__ __forceinline__ bool tile_is_interior(const GemmParams& p, int m0, int n0, int BM, int BN) {
  return (m0 + BM <= p.M) && (n0 + BN <= p.N) && vec_path_ok(p);
}
__device__ __forceinline__ void block_to_tile(const GemmParams& p, int bid, int& tm, int& tn) {
  // Blocks are dealt round-robin over the 8 XCDs (bid & 7 labels the XCD), each with its own L2.  Tiles are ordered in
  // super-tiles of gm M-tiles x all N-tiles (M fastest), so the ~32 tiles an XCD runs together form a gm x 32/gm window
  // sharing gm A panels and 32/gm W panels in that L2.
  //   xcd_sync = 0: XCD x walks a contiguous 1/8 of the tile sequence (bijective for any tile count).
  //   xcd_sync = 1: XCD x walks super-tiles x, x + 8, x + 16, ...: the eight XCDs sweep the SAME N range at the same time, each on
  //                 its own M rows — a W panel is pulled into the Infinity Cache once and serves eight L2 fills while it is hot;
  //                 the super-tiles past the last full round of eight fall back to the contiguous split (still bijective).
  const int nwg = p.tiles_m * p.tiles_n;
  const int gm = p.group_m > 0 ? p.group_m : GROUP_M;
  const int per_group = gm * p.tiles_n;
  int covered = 0;
  if (p.xcd_sync) {
    covered = ((p.tiles_m / gm) >> 3) * 8 * per_group;
    if (bid < covered) {
      const int x = bid & 7, j = bid >> 3;
      const int r = j / per_group, i = j - r * per_group;
      tm = (8 * r + x) * gm + i % gm;
      tn = i / gm;
      return;
    }
  }
  const int nt = nwg - covered, b = bid - covered;
  const int xcd = b & 7, q = nt >> 3, r = nt & 7;
  const int wgid = covered + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);   // bijective XCD remap
  const int group = wgid / per_group;
  const int first_m = group * gm;
  const int gsize = min(p.tiles_m - first_m, gm);
  const int in_group = wgid - group * per_group;
  tm = first_m + in_group % gsize;
  tn = in_group / gsize;
}

// Fused RoPE + KV-cache append for the QKV projection (icl_gemm_rope_kv_bf16): the row phase of the staged epilogue.
// head_dim = 128, so a 256-column tile holds two whole heads of q, of k or of v; the staged tile is bf16, i.e. the
// rotation sees exactly the values the unfused path would have read back from HBM (same rounding points, rope_rot8).
struct RopeFuse {
  const float* cosT;
  const float* sinT;
  const int* pos;
  const int* seq_ids;
  unsigned short* kc;
  unsigned short* vc;
  int k_off, v_off, H, max_len;
  int kv_rows_to_c;   // 0: k / v go to the cache only (the prefill attention reads them there)
};

// 256x256 rolling-pipeline tile (gemm256.hip): one kernel instantiation per epilogue kind, selected on the host
int launch_tile256(GemmParams& p, int batch, hipStream_t stream, const RopeFuse* rope = nullptr);

}  // namespace iclg
