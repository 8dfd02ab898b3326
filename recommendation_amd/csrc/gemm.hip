// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain),
// with the tower epilogues fused: bias, accumulate, relu, dropout, relu/dropout-backward mask.
// Block tile 128x128x16, 4 waves (2x2), each wave owns a 64x64 sub-tile = 2x2 MFMA tiles.
// Both operands are staged k-major in LDS ([k][m] / [k][n]) so an MFMA operand fetch is one
// conflict-free ds_read_b32 (32 consecutive floats per half-wave).  Small-M/N, long-K shapes
// (weight gradients) are split along K into fixed-order partial slabs, reduced by a second
// kernel: no atomics, bitwise reproducible.
#include "common.h"

namespace ctr {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_K = BM + 2;   // stride for operands loaded k-contiguous (scalar transposing writes)
constexpr int LDS_MN = BM + 4;  // stride for operands loaded mn-contiguous (16-byte writes)

struct Epi {
  const float* bias;
  int relu, accumulate;
  Drop drop;
  const float* mask_src;
  int64_t mask_ld;
  float mask_scale;
};

__device__ static inline float epilogue(const Epi& e, float v, int m, int n, int N, const float* cptr) {
  if (e.bias) v += e.bias[n];
  if (e.accumulate) v += *cptr;
  if (e.relu == 1) v = fmaxf(v, 0.f);
  else if (e.relu == 2) v = 1.f / (1.f + expf(-v));
  if (e.drop.on) v *= drop_mul(e.drop, (uint64_t)m * (uint64_t)N + (uint64_t)n);
  if (e.mask_src) v = e.mask_src[(int64_t)m * e.mask_ld + n] > 0.f ? v * e.mask_scale : 0.f;
  return v;
}

// Fetch this thread's share of a 128 x 16 operand tile into 8 registers.
//   KCONTIG: element (r, k) at P[r*ld + k]   (rows = the M or N index)
//   else:    element (r, k) at P[k*ld + r]
template <bool KCONTIG>
__device__ static inline void tile_load(const float* __restrict__ P, int64_t ld, bool vec, int r0, int rmax,
                                        int k0, int kend, int t, float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (KCONTIG) {
      const int r = r0 + (t >> 2) + 64 * j;
      const int k = k0 + (t & 3) * 4;
      const float* p = P + (int64_t)r * ld + k;
      if (r < rmax && vec && k + 3 < kend) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        reg[4 * j + 0] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[4 * j + i] = (r < rmax && k + i < kend) ? p[i] : 0.f;
      }
    } else {
      const int k = k0 + (t >> 5) + 8 * j;
      const int r = r0 + (t & 31) * 4;
      const float* p = P + (int64_t)k * ld + r;
      if (k < kend && vec && r + 3 < rmax) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        reg[4 * j + 0] = v.x; reg[4 * j + 1] = v.y; reg[4 * j + 2] = v.z; reg[4 * j + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[4 * j + i] = (k < kend && r + i < rmax) ? p[i] : 0.f;
      }
    }
  }
}

template <bool KCONTIG>
__device__ static inline void tile_store(float* __restrict__ S, int t, const float (&reg)[8]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (KCONTIG) {
      const int r = (t >> 2) + 64 * j;
      const int k = (t & 3) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) S[(k + i) * LDS_K + r] = reg[4 * j + i];
    } else {
      const int k = (t >> 5) + 8 * j;
      const int r = (t & 31) * 4;
      *reinterpret_cast<float4*>(&S[k * LDS_MN + r]) =
          make_float4(reg[4 * j + 0], reg[4 * j + 1], reg[4 * j + 2], reg[4 * j + 3]);
    }
  }
}

// A_K: A stored [M,K] (k contiguous).  B_K: B stored [N,K] (k contiguous).
template <bool A_K, bool B_K>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A,
                                                      int64_t lda, bool vecA, const float* __restrict__ Bm,
                                                      int64_t ldb, bool vecB, float* __restrict__ C,
                                                      int64_t ldc, Epi epi, int k_chunk,
                                                      float* __restrict__ partial) {
  constexpr int SA = A_K ? LDS_K : LDS_MN;
  constexpr int SB = B_K ? LDS_K : LDS_MN;
  __shared__ __attribute__((aligned(16))) float As[BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * SB];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_chunk;
  const int kend = kbeg + k_chunk < K ? kbeg + k_chunk : K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[8], rb[8];
  tile_load<A_K>(A, lda, vecA, m0, M, kbeg, kend, t, ra);
  tile_load<B_K>(Bm, ldb, vecB, n0, N, kbeg, kend, t, rb);

  const int lr = lane & 31, lk = lane >> 5;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    tile_store<A_K>(As, t, ra);
    tile_store<B_K>(Bs, t, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next tile while this one is multiplied
      tile_load<A_K>(A, lda, vecA, m0, M, k0 + BK, kend, t, ra);
      tile_load<B_K>(Bm, ldb, vecB, n0, N, k0 + BK, kend, t, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + lk;
      const float a0 = As[k * SA + wm * 64 + lr];
      const float a1 = As[k * SA + wm * 64 + 32 + lr];
      const float b0 = Bs[k * SB + wn * 64 + lr];
      const float b1 = Bs[k * SB + wn * 64 + 32 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int n = n0 + wn * 64 + j * 32 + lr;
        if (m < M && n < N) {
          if (partial) {
            partial[((int64_t)blockIdx.z * M + m) * N + n] = acc[i][j][r];
          } else {
            float* cp = C + (int64_t)m * ldc + n;
            *cp = epilogue(epi, acc[i][j][r], m, n, N, cp);
          }
        }
      }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                           int M, int N, float* __restrict__ C, int64_t ldc,
                                                           Epi epi) {
  const int64_t total = (int64_t)M * N;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p % N);
    const int m = (int)(p / N);
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += partial[(int64_t)s * total + p];
    float* cp = C + (int64_t)m * ldc + n;
    *cp = epilogue(epi, v, m, n, N, cp);
  }
}

struct Plan {
  int tiles_m, tiles_n, splits, k_chunk;
};

static Plan make_plan(int M, int N, int K) {
  Plan p;
  p.tiles_m = (int)cdiv(M, BM);
  p.tiles_n = (int)cdiv(N, BN);
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  int splits = 1;
  if (tiles < 256 && K >= 8 * BK) {
    splits = (int)cdiv(512, tiles);
    const int max_splits = K / (4 * BK);  // keep at least 4 k-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 128) splits = 128;
    if (splits < 1) splits = 1;
  }
  int k_chunk = (int)cdiv(cdiv(K, splits), BK) * BK;
  if (k_chunk < BK) k_chunk = BK;
  p.splits = (int)cdiv(K > 0 ? K : 1, k_chunk);
  p.k_chunk = k_chunk;
  return p;
}

}  // namespace ctr

using namespace ctr;

extern "C" size_t ctr_gemm_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan p = make_plan(M, N, K);
  return p.splits > 1 ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int ctr_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, const ctr_epilogue_t* epi,
                            void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "ctr_gemm_f32: negative size");
  if (M == 0 || N == 0) return CTR_OK;
  CTR_REQUIRE(K > 0, "ctr_gemm_f32: K must be positive");
  CTR_REQUIRE(A && B && C, "ctr_gemm_f32: null pointer");
  CTR_REQUIRE(lda >= (trans_a ? M : K), "ctr_gemm_f32: lda=%lld too small", (long long)lda);
  CTR_REQUIRE(ldb >= (trans_b ? K : N), "ctr_gemm_f32: ldb=%lld too small", (long long)ldb);
  CTR_REQUIRE(ldc >= N, "ctr_gemm_f32: ldc=%lld too small", (long long)ldc);
  Epi e;
  e.bias = nullptr; e.relu = 0; e.accumulate = 0; e.drop = make_drop(nullptr);
  e.mask_src = nullptr; e.mask_ld = 0; e.mask_scale = 1.f;
  if (epi) {
    e.bias = epi->bias; e.relu = epi->relu; e.accumulate = epi->accumulate;
    e.drop = make_drop(epi->dropout);
    e.mask_src = epi->mask_src; e.mask_ld = epi->mask_ld; e.mask_scale = epi->mask_scale;
    CTR_REQUIRE(!e.mask_src || e.mask_ld >= N, "ctr_gemm_f32: mask_ld too small");
  }
  const Plan p = make_plan(M, N, K);
  float* partial = nullptr;
  if (p.splits > 1) {
    const size_t need = (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      set_error("ctr_gemm_f32: workspace %zu < %zu", workspace_bytes, need);
      return CTR_EWORKSPACE;
    }
    partial = (float*)workspace;
  }
  const bool vecA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const bool vecB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  const dim3 grid(p.tiles_n, p.tiles_m, p.splits), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(AK, BK_)                                                                                 \
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_>), grid, block, 0, st, M, N, K, A, lda, vecA, B, ldb, vecB, C, \
                     ldc, e, p.k_chunk, partial)
  if (!trans_a && !trans_b) LAUNCH(true, false);
  else if (!trans_a && trans_b) LAUNCH(true, true);
  else if (trans_a && !trans_b) LAUNCH(false, false);
  else LAUNCH(false, true);
#undef LAUNCH
  if (partial) {
    const int64_t total = (int64_t)M * N;
    const int64_t blocks = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, p.splits, M, N,
                       C, ldc, e);
  }
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
