// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain),
// with the tower epilogues fused: bias, accumulate, relu, dropout, relu/dropout-backward mask.
// Block tile 128x128x16, 4 waves (2x2), each wave owns a 64x64 sub-tile = 2x2 MFMA tiles.
// Both operands are staged k-major in LDS ([k][m] / [k][n]) so an MFMA operand fetch is one
// conflict-free ds_read_b32 (32 consecutive floats per half-wave).  Small-M/N, long-K shapes
// (weight gradients) are split along K into fixed-order partial slabs, reduced by a second
// kernel: no atomics, bitwise reproducible.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace ctr {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BK = 16;
// LDS row stride: tile rows + 2 for operands loaded k-contiguous (scalar transposing writes land on
// distinct banks), + 4 for operands loaded mn-contiguous (16-byte aligned vector writes).

// FM terms of the front end's backward, added to the first `cols` columns of the product (see ctr_fm_bwd_t)
struct FmBwd {
  int on, E, cols, eshift;   // eshift: log2(E) when E is a power of two, else -1
  const float *g1, *g2, *s, *z;
  int64_t g1_ld, g2_ld, z_ld;
};

struct Epi {
  const float* bias;
  int relu, accumulate;
  Drop drop;
  const float* mask_src;
  int64_t mask_ld;
  float mask_scale;
  FmBwd fm;
  float* amax_out;   // optional: max |C| over the stored elements is raised here (bits of a non-negative float, atomicMax)
};

__device__ static inline float epilogue(const Epi& e, float v, int m, int n, int N, const float* cptr) {
  if (e.bias) v += e.bias[n];
  if (e.accumulate) v += *cptr;
  if (e.relu == 1) v = fmaxf(v, 0.f);
  else if (e.relu == 2) v = 1.f / (1.f + expf(-v));
  if (e.drop.on) v *= drop_mul(e.drop, (uint64_t)m * (uint64_t)N + (uint64_t)n);
  if (e.mask_src) v = e.mask_src[(int64_t)m * e.mask_ld + n] > 0.f ? v * e.mask_scale : 0.f;
  if (e.fm.on && n < e.fm.cols) {
    const int f = n / e.fm.E, c = n - f * e.fm.E;
    if (e.fm.g1) v += e.fm.g1[(int64_t)m * e.fm.g1_ld + f];
    if (e.fm.g2)
      v = fmaf(e.fm.g2[(int64_t)m * e.fm.g2_ld + c],
               e.fm.s[(int64_t)m * e.fm.E + c] - (e.fm.z ? e.fm.z[(int64_t)m * e.fm.z_ld + n] : 0.f), v);
  }
  return v;
}

// Fetch this thread's share of an R x BK operand tile (R = 32..256) as float4 slots.
//   KCONTIG: element (r, k) at P[r*ld + k]   (rows = the M or N index); slot -> (r = s/(BK/4), k = 4*(s%(BK/4)))
//   else:    element (r, k) at P[k*ld + r];                            slot -> (k = s/(R/4), r = 4*(s%(R/4)))
// FULLK: the whole k range [k0, k0 + BK) is inside the matrix (no k bound checks on the fast path).
template <bool KCONTIG, int R, bool FULLK>
__device__ static inline void tile_load(const float* __restrict__ P, int64_t ld, bool vec, int r0, int rmax,
                                        int k0, int kend, int t, float (&reg)[(R * (BK / 4) + 255) / 256 * 4]) {
  constexpr int SLOTS = R * (BK / 4);
#pragma unroll
  for (int j = 0; j < (SLOTS + 255) / 256; ++j) {
    const int s = t + 256 * j;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (SLOTS % 256 == 0 || s < SLOTS) {
      if (KCONTIG) {
        const int r = r0 + s / (BK / 4);
        const int k = k0 + (s % (BK / 4)) * 4;
        const float* p = P + (int64_t)r * ld + k;
        if (r < rmax && vec && (FULLK || k + 3 < kend)) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
        } else if (r < rmax) {
          if (k + 0 < kend) v0 = p[0];
          if (k + 1 < kend) v1 = p[1];
          if (k + 2 < kend) v2 = p[2];
          if (k + 3 < kend) v3 = p[3];
        }
      } else {
        const int k = k0 + s / (R / 4);
        const int r = r0 + (s % (R / 4)) * 4;
        const float* p = P + (int64_t)k * ld + r;
        if ((FULLK || k < kend) && vec && r + 3 < rmax) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
        } else if (k < kend) {
          if (r + 0 < rmax) v0 = p[0];
          if (r + 1 < rmax) v1 = p[1];
          if (r + 2 < rmax) v2 = p[2];
          if (r + 3 < rmax) v3 = p[3];
        }
      }
    }
    reg[4 * j + 0] = v0; reg[4 * j + 1] = v1; reg[4 * j + 2] = v2; reg[4 * j + 3] = v3;
  }
}

template <bool KCONTIG, int R>
__device__ static inline void tile_store(float* __restrict__ S, int t, const float (&reg)[(R * (BK / 4) + 255) / 256 * 4]) {
  constexpr int SLOTS = R * (BK / 4);
  constexpr int LD = KCONTIG ? R + 2 : R + 4;
#pragma unroll
  for (int j = 0; j < (SLOTS + 255) / 256; ++j) {
    const int s = t + 256 * j;
    if (SLOTS % 256 == 0 || s < SLOTS) {
      if (KCONTIG) {
        const int r = s / (BK / 4);
        const int k = (s % (BK / 4)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) S[(k + i) * LD + r] = reg[4 * j + i];
      } else {
        const int k = s / (R / 4);
        const int r = (s % (R / 4)) * 4;
        *reinterpret_cast<float4*>(&S[k * LD + r]) =
            make_float4(reg[4 * j + 0], reg[4 * j + 1], reg[4 * j + 2], reg[4 * j + 3]);
      }
    }
  }
}

// Epilogue shared by the tiled kernels.  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).  Each 32 x 32 tile goes through a per-wave LDS tile and
// leaves as 16-byte stores (8 rows x 128 bytes per wave instruction): a quarter of the store
// instructions of the register layout.  smem must hold 4 * 32 * 36 floats and no longer be in use.
// PRE: the operands of the mask and FM epilogues are requested a row of wave tiles at a time (registers the
// 128-register fp32-MFMA kernel does not have: it fetches them where they are used).
template <int TM, int TN, bool PRE>
__device__ static inline void tile_epilogue(f32x16 (&acc)[TM][TN], float* smem, int M, int N, int m0, int n0, int wm,
                                            int wn, float* __restrict__ C, int64_t ldc, const Epi& epi,
                                            float* __restrict__ partial, int zslice) {
  constexpr int CS = 36;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  __syncthreads();  // every wave is done with As / Bs
  float* S = smem + wave * (32 * CS);
  float* out = partial ? partial + (int64_t)zslice * M * N : C;
  const int64_t ldo = partial ? N : ldc;
  const bool vec_out = (ldo % 4 == 0) && ((uintptr_t)out % 16 == 0);
  const bool vec_mask = epi.mask_src && (epi.mask_ld % 4 == 0) && ((uintptr_t)epi.mask_src % 16 == 0);
  const int srow = lane >> 3, sc4 = (lane & 7) * 4;
  // FM terms (ctr_fm_bwd_t): with E in {4, 8, 16, 32} every 32-column tile of a row sees the same columns of s and
  // g_second, and all operands of a row of wave tiles are requested before the first tile is written (requested
  // tile by tile, sixteen dependent round trips per block cost half as much as the product itself)
  const bool fm_fast = PRE && epi.fm.on && !partial && epi.fm.eshift >= 2 && epi.fm.eshift <= 5;
  const bool g2_pairs = fm_fast && epi.fm.g2 && ((uintptr_t)epi.fm.g2 % 8 == 0) && (epi.fm.g2_ld % 2 == 0);
  // a lane's four bias values depend on its column tile only: fetched once, not once per row group
  float bias4[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (32 * TN) + j * 32 + sc4;
#pragma unroll
    for (int u = 0; u < 4; ++u) bias4[j][u] = (epi.bias && !partial && n + u < N) ? epi.bias[n + u] : 0.f;
  }
  float amx = 0.f;   // max |value stored| of this lane (epi.amax_out)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    float4 fz[TN][4], fs[4];
    float fg1[TN][4], fg2[4][4];
    if (fm_fast) {
      const int nb = n0 + wn * (32 * TN) + sc4;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m0 + wm * (32 * TM) + i * 32 + it * 8 + srow;
        fs[it] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) fg2[it][u] = 0.f;
        if (m < M && nb < epi.fm.cols && epi.fm.g2) {
          const int c = nb & (epi.fm.E - 1);
          fs[it] = *reinterpret_cast<const float4*>(epi.fm.s + (int64_t)m * epi.fm.E + c);
          const float* g2 = epi.fm.g2 + (int64_t)m * epi.fm.g2_ld + c;
          if (g2_pairs) {   // rows of g_second 8-byte aligned: two loads instead of four
            const float2 lo = *reinterpret_cast<const float2*>(g2), hi = *reinterpret_cast<const float2*>(g2 + 2);
            fg2[it][0] = lo.x; fg2[it][1] = lo.y; fg2[it][2] = hi.x; fg2[it][3] = hi.y;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) fg2[it][u] = g2[u];
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = nb + j * 32;
          fz[j][it] = make_float4(0.f, 0.f, 0.f, 0.f);
          fg1[j][it] = 0.f;
          if (m < M && n < epi.fm.cols) {
            if (epi.fm.g2 && epi.fm.z) fz[j][it] = *reinterpret_cast<const float4*>(epi.fm.z + (int64_t)m * epi.fm.z_ld + n);
            if (epi.fm.g1) fg1[j][it] = epi.fm.g1[(int64_t)m * epi.fm.g1_ld + (n >> epi.fm.eshift)];
          }
        }
      }
    }
    // relu / dropout-backward masks of the whole row of wave tiles, requested together (one round trip instead of
    // one per row group of every tile)
    float4 mkv[TN][4];
    const bool mk_pre = PRE && epi.mask_src && !partial && vec_mask;
    if (mk_pre) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = m0 + wm * (32 * TM) + i * 32 + it * 8 + srow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * (32 * TN) + j * 32 + sc4;
          mkv[j][it] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m < M && n + 3 < N) mkv[j][it] = *reinterpret_cast<const float4*>(epi.mask_src + (int64_t)m * epi.mask_ld + n);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * lk) * CS + lr] = acc[i][j][r];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + srow;
        const float4 v4 = *reinterpret_cast<const float4*>(&S[row * CS + sc4]);
        const int m = m0 + wm * (32 * TM) + i * 32 + row;
        const int n = n0 + wn * (32 * TN) + j * 32 + sc4;
        if (m >= M || n >= N) continue;
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float* cp = out + (int64_t)m * ldo + n;
        const bool full = n + 3 < N;
        if (!partial) {
          float cold[4] = {0.f, 0.f, 0.f, 0.f}, mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (epi.accumulate) {
            if (full && vec_out) {
              const float4 c4 = *reinterpret_cast<const float4*>(cp);
              cold[0] = c4.x; cold[1] = c4.y; cold[2] = c4.z; cold[3] = c4.w;
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (n + u < N) cold[u] = cp[u];
            }
          }
          if (epi.mask_src) {
            const float* mp = epi.mask_src + (int64_t)m * epi.mask_ld + n;
            if (full && vec_mask) {
              const float4 m4 = PRE ? mkv[j][it] : *reinterpret_cast<const float4*>(mp);
              mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w;
            } else {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (n + u < N) mk[u] = mp[u];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (n + u >= N) break;
            float x = v[u];
            if (epi.bias) x += bias4[j][u];
            if (epi.accumulate) x += cold[u];
            if (epi.relu == 1) x = fmaxf(x, 0.f);
            else if (epi.relu == 2) x = 1.f / (1.f + expf(-x));
            if (epi.drop.on) x *= drop_mul(epi.drop, (uint64_t)m * (uint64_t)N + (uint64_t)(n + u));
            if (epi.mask_src) x = mk[u] > 0.f ? x * epi.mask_scale : 0.f;
            v[u] = x;
          }
          if (epi.fm.on && n < epi.fm.cols) {   // cols % 4 == 0: the four columns belong to one row of Z
            if (fm_fast) {
              if (epi.fm.g1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += fg1[j][it];
              }
              if (epi.fm.g2) {
                v[0] = fmaf(fg2[it][0], fs[it].x - fz[j][it].x, v[0]);
                v[1] = fmaf(fg2[it][1], fs[it].y - fz[j][it].y, v[1]);
                v[2] = fmaf(fg2[it][2], fs[it].z - fz[j][it].z, v[2]);
                v[3] = fmaf(fg2[it][3], fs[it].w - fz[j][it].w, v[3]);
              }
            } else {
              const int f = n / epi.fm.E, c = n - f * epi.fm.E;
              if (epi.fm.g1) {
                const float t = epi.fm.g1[(int64_t)m * epi.fm.g1_ld + f];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += t;
              }
              if (epi.fm.g2) {
                const float4 zz = epi.fm.z ? *reinterpret_cast<const float4*>(epi.fm.z + (int64_t)m * epi.fm.z_ld + n)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 ss = *reinterpret_cast<const float4*>(epi.fm.s + (int64_t)m * epi.fm.E + c);
                const float* g2 = epi.fm.g2 + (int64_t)m * epi.fm.g2_ld + c;
                v[0] = fmaf(g2[0], ss.x - zz.x, v[0]);
                v[1] = fmaf(g2[1], ss.y - zz.y, v[1]);
                v[2] = fmaf(g2[2], ss.z - zz.z, v[2]);
                v[3] = fmaf(g2[3], ss.w - zz.w, v[3]);
              }
            }
          }
        }
        if (full && vec_out) {
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
          amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (n + u < N) { cp[u] = v[u]; amx = fmaxf(amx, fabsf(v[u])); }
        }
      }
    }
  }
  if (epi.amax_out && !partial) raise_amax(epi.amax_out, amx);
}

// Which tile a workgroup computes.  The hardware hands workgroups to the eight XCDs round-robin in launch order (x
// fastest), and each XCD has an L2 of its own: in a split-K launch with a handful of tiles per k-slice (the tower's
// weight gradients: 8 tiles, 64-256 slices) the plain blockIdx mapping spreads the tiles of one slice, which share
// its rows of both operands, over all XCDs, and each fetches them again from beyond its L2 (measured: 2.4 x the
// operand bytes).  There XCD x gets a contiguous range of the logical tile order instead, so a slice's tiles run on
// one XCD within a few dispatches of each other (429 x 256 x 65536: 97.7 -> 86.5 us).  Launches with many tiles per
// slice keep the plain order, which measured faster for them (4096^3: 548 vs 597 us).  Placement only changes who
// computes what, never a result; on any other dispatch order the mapping is still a bijection.
__device__ static inline void xcd_tile(int& bx, int& by, int& bz) {
  const unsigned nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (nz == 1 || nx * ny > 16) return;
  const unsigned total = nx * ny * nz;
  const unsigned lin = (blockIdx.z * ny + blockIdx.y) * nx + blockIdx.x;
  const unsigned per = total / 8u, rem = total % 8u, x = lin % 8u, slot = lin / 8u;
  const unsigned L = x * per + (x < rem ? x : rem) + slot;
  bx = (int)(L % nx);
  by = (int)((L / nx) % ny);
  bz = (int)(L / (nx * ny));
}

// A_K: A stored [M,K] (k contiguous).  B_K: B stored [N,K] (k contiguous).
// Tile TBM x TBN x BK, four waves as WGM x (4/WGM); a wave owns TM x TN MFMA tiles of 32x32.
template <bool A_K, bool B_K, int TBM, int TBN, int WGM>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A,
                                                      int64_t lda, bool vecA, const float* __restrict__ Bm,
                                                      int64_t ldb, bool vecB, float* __restrict__ C,
                                                      int64_t ldc, Epi epi, int k_chunk,
                                                      float* __restrict__ partial) {
  epi.drop = resolve_drop(epi.drop);
  constexpr int WGN = 4 / WGM;
  constexpr int TM = TBM / WGM / 32, TN = TBN / WGN / 32;
  constexpr int SA = A_K ? TBM + 2 : TBM + 4;
  constexpr int SB = B_K ? TBN + 2 : TBN + 4;
  constexpr int RA = (TBM * (BK / 4) + 255) / 256 * 4, RB = (TBN * (BK / 4) + 255) / 256 * 4;
  constexpr int CS = 36;  // row stride of the per-wave 32 x 32 staging tile of the epilogue
  constexpr int SMEM = BK * (SA + SB) > 4 * 32 * CS ? BK * (SA + SB) : 4 * 32 * CS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* As = smem;
  float* Bs = smem + BK * SA;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int m0 = blockIdx.y * TBM, n0 = blockIdx.x * TBN;
  const int kbeg = blockIdx.z * k_chunk;
  const int kend = kbeg + k_chunk < K ? kbeg + k_chunk : K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float ra[RA], rb[RB];
  if (kbeg + BK <= kend) {
    tile_load<A_K, TBM, true>(A, lda, vecA, m0, M, kbeg, kend, t, ra);
    tile_load<B_K, TBN, true>(Bm, ldb, vecB, n0, N, kbeg, kend, t, rb);
  } else {
    tile_load<A_K, TBM, false>(A, lda, vecA, m0, M, kbeg, kend, t, ra);
    tile_load<B_K, TBN, false>(Bm, ldb, vecB, n0, N, kbeg, kend, t, rb);
  }

  const int lr = lane & 31, lk = lane >> 5;
  // number of this wave's 32-row / 32-column tiles that reach into the matrix
  const int mrem = M - (m0 + wm * (32 * TM)), nrem = N - (n0 + wn * (32 * TN));
  const int tm_live = mrem <= 0 ? 0 : (mrem + 31) / 32 < TM ? (mrem + 31) / 32 : TM;
  const int tn_live = nrem <= 0 ? 0 : (nrem + 31) / 32 < TN ? (nrem + 31) / 32 : TN;
  const bool whole = tm_live == TM && tn_live == TN;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    tile_store<A_K, TBM>(As, t, ra);
    tile_store<B_K, TBN>(Bs, t, rb);
    __syncthreads();
    // prefetch the next tile while this one is multiplied; only the last tile of a ragged K takes
    // the bound-checked path
    if (k0 + 2 * BK <= kend) {
      tile_load<A_K, TBM, true>(A, lda, vecA, m0, M, k0 + BK, kend, t, ra);
      tile_load<B_K, TBN, true>(Bm, ldb, vecB, n0, N, k0 + BK, kend, t, rb);
    } else if (k0 + BK < kend) {
      tile_load<A_K, TBM, false>(A, lda, vecA, m0, M, k0 + BK, kend, t, ra);
      tile_load<B_K, TBN, false>(Bm, ldb, vecB, n0, N, k0 + BK, kend, t, rb);
    }
    if (whole) {
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int k = 2 * kk + lk;
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[k * SA + wm * (32 * TM) + i * 32 + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[k * SB + wn * (32 * TN) + j * 32 + lr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // edge block: 32 x 32 tiles that lie wholly outside the matrix are skipped, which frees their
      // matrix-core slots for the other blocks resident on the SIMD
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int k = 2 * kk + lk;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (i >= tm_live) break;
          const float a = As[k * SA + wm * (32 * TM) + i * 32 + lr];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (j >= tn_live) break;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[k * SB + wn * (32 * TN) + j * 32 + lr], acc[i][j],
                                                             0, 0, 0);
          }
        }
      }
    }
  }

  tile_epilogue<TM, TN, false>(acc, smem, M, N, m0, n0, wm, wn, C, ldc, epi, partial, blockIdx.z);
}

// ---------------------------------------------------------------------------------------
// The same tiled GEMM on the bf16 matrix cores with fp32-level accuracy.  Every fp32 operand is
// split exactly into three bf16 terms x = hi + mid + lo (8 + 8 + 8 significant bits), and the
// product is accumulated from the six term pairs of order <= 2^-16:
//   hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi           (dropped pairs are <= 2^-25 relative)
// Each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result differs
// from an fp32 FMA chain only in rounding order (~1e-7 relative), while six 32x32x16 bf16 MFMAs cost
// 192 matrix-core cycles per 16 values of k against 512 for eight 32x32x2 fp32 MFMAs.
// The split happens once, on the way from global memory into LDS; LDS holds three bf16 planes per
// operand, k-contiguous in 32-byte rows whose halves are swizzled (lds_at: conflict-free 16-byte reads and writes).
// A non-finite input turns into NaN (inf - inf in the split).
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int SK = 16;    // k per stage = one bf16 MFMA step
constexpr int SROW = SK;      // 16-bit elements per LDS row: 32 bytes, no padding
constexpr int SKG = SK / 8;   // groups of 8 k per row
// Element offset of (row, k group kg) inside a plane.  Rows are 32 bytes apart, so rows 8 apart would meet on the
// same banks in a 16-byte fragment read (its lane groups span rows 0-3, 12-15, 20-27 / 4-11, 16-19, 28-31) and rows
// 4 apart in a 16-byte store (8 consecutive rows per group, 32 banks): the two 16-byte halves of a row swap places
// by the parity of row bits 2 and 3, which makes both conflict-free.  Against a padded 48-byte row this is a third
// less LDS per stage, so a fourth block fits on a CU.
__device__ static inline int lds_at(int row, int kg) {
  return row * SROW + ((kg ^ (((row >> 2) ^ (row >> 3)) & 1)) << 3);
}

// eight consecutive-k elements (k0 .. k0+7) of row r; KCONTIG: element (r, k) at P[r*ld + k], else P[k*ld + r].
// FAST (whole stage inside K, vector loads allowed): no bound checks at all; a row past the end is
// clamped to the last row, whose products only reach output rows that are never stored.
template <bool KCONTIG, bool FAST>
__device__ static inline void split_load8(const float* __restrict__ P, int64_t ld, bool vec, int r, int rmax, int k0,
                                          int kend, float (&v)[8]) {
  if (FAST) {
    const int rc = r < rmax ? r : rmax - 1;
    if (KCONTIG) {
      const float* p = P + (int64_t)rc * ld + k0;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const float* p = P + (int64_t)k0 * ld + rc;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)j * ld];
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (r >= rmax || k0 >= kend) return;
  const float* p = KCONTIG ? P + (int64_t)r * ld + k0 : P + (int64_t)k0 * ld + r;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (k0 + j < kend) v[j] = KCONTIG ? p[j] : p[(int64_t)j * ld];
}

// Two-term fp16 split (PL = 2): x * scale = hi + lo with hi = fp16(x * scale), lo = fp16(x * scale - hi), both rounded
// to nearest: 22 significant bits where lo is a normal fp16 number.  `scale` is a power of two that brings the
// matrix's largest magnitude just under 2^15 (see scale_for), so x * scale is exact, nothing overflows, and the part
// of an element that falls below fp16's smallest subnormal is at most 2^-40 of the matrix maximum.  The product is
// accumulated from three fp16 MFMAs, lo*hi + hi*lo + hi*hi (the dropped lo*lo is <= 2^-22 relative): ~3e-7
// relative, against ~1e-7 for the three-term bf16 split, at half the matrix-core work and two thirds of the LDS
// traffic.  It needs the two operands' largest magnitudes before the launch (ctr_epilogue_t.a_amax / b_amax).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int F16_TARGET_EXP = 15;   // the largest magnitude lands in [2^14, 2^15)

// 2^k with amax * 2^k in [2^(F16_TARGET_EXP-1), 2^F16_TARGET_EXP); 1 for amax = 0, inf or NaN.  k is returned too.
__device__ static inline float scale_for(float amax, int& k) {
  const unsigned bits = __float_as_uint(amax);
  const int e = (int)((bits >> 23) & 0xFFu);          // biased exponent: amax in [2^(e-127), 2^(e-126))
  k = 0;
  if (e == 0 || e == 255) return 1.f;                 // zero / subnormal maximum, or non-finite: leave the values alone
  k = F16_TARGET_EXP - 1 - (e - 127);
  if (k > 100) k = 100;
  if (k < -100) k = -100;
  return __uint_as_float((unsigned)(127 + k) << 23);
}

__device__ static inline void split_store8_f16(_Float16* __restrict__ S, int R, int row, int kg, const float (&v)[8],
                                               float scale) {
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * scale;
    const _Float16 h = (_Float16)x;
    hi[j] = h;
    lo[j] = (_Float16)(x - (float)h);
  }
  _Float16* p = S + lds_at(row, kg);
  *reinterpret_cast<f16x8*>(p) = hi;
  *reinterpret_cast<f16x8*>(p + R * SROW) = lo;
}

// S: PL planes [PL][R][SROW] of bf16; writes the hi / mid / lo terms of v at (row, 8*kg .. 8*kg+7).  PL = 1 keeps
// only the leading term: v rounded to bf16 (round to nearest even), the plain bf16-in / fp32-accumulate product.
template <int PL>
__device__ static inline void split_store8(__bf16* __restrict__ S, int R, int row, int kg, const float (&v)[8]) {
  bf16x8 hi, mid, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    if (PL == 3) {
      const float r1 = v[j] - (float)h;
      const __bf16 m = (__bf16)r1;
      const float r2 = r1 - (float)m;
      mid[j] = m; lo[j] = (__bf16)r2;
    }
  }
  __bf16* p = S + lds_at(row, kg);
  *reinterpret_cast<bf16x8*>(p) = hi;
  if (PL == 3) {
    *reinterpret_cast<bf16x8*>(p + R * SROW) = mid;
    *reinterpret_cast<bf16x8*>(p + 2 * R * SROW) = lo;
  }
}

// Pipeline: LDS holds two stages.  In iteration s a wave (1) issues the global loads of stage s+2 into one
// register set, (2) multiplies stage s out of LDS buffer s%2 while it splits the registers of stage s+1 and
// writes them to buffer (s+1)%2 - the split's VALU and LDS writes fill the issue slots between MFMAs - and
// (3) meets the block at ONE barrier.  The loop is unrolled by two so the register sets alternate without
// moves.
// PL = 3: the three-term bf16 split; PL = 1: operands rounded to bf16; PL = 2: the two-term fp16 split, whose
// scales come from the operands' largest magnitudes (sc.a_amax / sc.b_amax, device words written before the launch).
struct Scales {
  const float *a_amax, *b_amax;
  float a_floor, b_floor;     // lower bounds of the two maxima (a column of ones the producer did not see)
};

template <bool A_K, bool B_K, int TBM, int TBN, int WGM, int NW = 4, int PL = 3>
__global__ __launch_bounds__(64 * NW, 2) void gemm_split_kernel(int M, int N, int K, const float* __restrict__ A,
                                                           int64_t lda, bool vecA, const float* __restrict__ Bm,
                                                           int64_t ldb, bool vecB, float* __restrict__ C,
                                                           int64_t ldc, Epi epi, int k_chunk,
                                                           float* __restrict__ partial, Scales sc) {
  epi.drop = resolve_drop(epi.drop);
  float scale_a = 1.f, scale_b = 1.f, unscale_a = 1.f, unscale_b = 1.f;
  if (PL == 2) {
    int ka, kb;
    scale_a = scale_for(fmaxf(read_amax(sc.a_amax), sc.a_floor), ka);
    scale_b = scale_for(fmaxf(read_amax(sc.b_amax), sc.b_floor), kb);
    unscale_a = __uint_as_float((unsigned)(127 - ka) << 23);    // |k| <= 100: both factors are normal numbers
    unscale_b = __uint_as_float((unsigned)(127 - kb) << 23);
  }
  constexpr int NT = 64 * NW;                            // threads: NW waves, each a 64 x 64 output tile
  constexpr int WGN = NW / WGM;
  constexpr int TM = TBM / WGM / 32, TN = TBN / WGN / 32;
  constexpr int GA = TBM * SKG, GB = TBN * SKG;          // groups of 8 k per operand tile
  constexpr int NA = (GA + NT - 1) / NT, NB = (GB + NT - 1) / NT;
  constexpr int STAGE = PL * (TBM + TBN) * SROW;         // bf16 elements of one stage (A planes, then B planes)
  constexpr int OPER = 2 * STAGE * 2;                    // bytes
  constexpr int SMEM = OPER > NW * 32 * 36 * 4 ? OPER : NW * 32 * 36 * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* S0 = reinterpret_cast<__bf16*>(smem_raw);
  static_assert(SMEM <= 160 * 1024, "tile too large for the LDS");
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  int bx, by, bz;
  xcd_tile(bx, by, bz);
  const int m0 = by * TBM, n0 = bx * TBN;
  const int kbeg = bz * k_chunk;
  const int kend = kbeg + k_chunk < K ? kbeg + k_chunk : K;
  const int nstages = (kend - kbeg + SK - 1) / SK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // group g of an operand tile with R rows: k-contiguous operands walk (row, kg) = (g / SKG, g % SKG) so a
  // row's bytes sit in neighbouring threads; the others walk (g % R, g / R) so lanes read consecutive rows
  // k-contiguous operands need 16-byte alignment for the unchecked path; the others never do
  const bool vec_ok = (!A_K || vecA) && (!B_K || vecB);
  // only_fast: the caller guarantees a whole stage inside K and vec_ok.  The steady-state loop uses it so that
  // its loads sit in straight-line code: behind a branch the compiler can no longer count the loads in
  // flight and waits for ALL of them - the stage just requested included - before the older stage is used.
  auto load_stage = [&](int st, auto only_fast, float (&va)[NA][8], float (&vb)[NB][8]) {
    const int k0 = kbeg + st * SK;
    const bool fast = decltype(only_fast)::value || (k0 + SK <= kend && vec_ok);
    if (fast) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int g = t + NT * q;
        if (GA % NT == 0 || g < GA)
          split_load8<A_K, true>(A, lda, vecA, m0 + (A_K ? g / SKG : g % TBM), M, k0 + 8 * (A_K ? g % SKG : g / TBM), kend,
                                 va[q]);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int g = t + NT * q;
        if (GB % NT == 0 || g < GB)
          split_load8<B_K, true>(Bm, ldb, vecB, n0 + (B_K ? g / SKG : g % TBN), N, k0 + 8 * (B_K ? g % SKG : g / TBN), kend,
                                 vb[q]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int g = t + NT * q;
        if (GA % NT == 0 || g < GA)
          split_load8<A_K, false>(A, lda, vecA, m0 + (A_K ? g / SKG : g % TBM), M, k0 + 8 * (A_K ? g % SKG : g / TBM), kend,
                                  va[q]);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int g = t + NT * q;
        if (GB % NT == 0 || g < GB)
          split_load8<B_K, false>(Bm, ldb, vecB, n0 + (B_K ? g / SKG : g % TBN), N, k0 + 8 * (B_K ? g % SKG : g / TBN), kend,
                                  vb[q]);
      }
    }
  };
  auto store_stage = [&](__bf16* S, const float (&va)[NA][8], const float (&vb)[NB][8]) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int g = t + NT * q;
      if (GA % NT == 0 || g < GA) {
        if (PL == 2)
          split_store8_f16(reinterpret_cast<_Float16*>(S), TBM, A_K ? g / SKG : g % TBM, A_K ? g % SKG : g / TBM, va[q],
                           scale_a);
        else
          split_store8<PL == 2 ? 1 : PL>(S, TBM, A_K ? g / SKG : g % TBM, A_K ? g % SKG : g / TBM, va[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int g = t + NT * q;
      if (GB % NT == 0 || g < GB) {
        if (PL == 2)
          split_store8_f16(reinterpret_cast<_Float16*>(S + PL * TBM * SROW), TBN, B_K ? g / SKG : g % TBN,
                           B_K ? g % SKG : g / TBN, vb[q], scale_b);
        else
          split_store8<PL == 2 ? 1 : PL>(S + PL * TBM * SROW, TBN, B_K ? g / SKG : g % TBN, B_K ? g % SKG : g / TBN, vb[q]);
      }
    }
  };
  const int lr = lane & 31, lk = lane >> 5;
  // STORE: also split the register set (va, vb) into buffer Snext; the scheduling hints at the end ask
  // for one MFMA followed by a handful of the split's VALU instructions, LDS writes spread in between
  auto multiply = [&](const __bf16* S, auto store_tag, __bf16* Snext, const float (&va)[NA][8],
                      const float (&vb)[NB][8]) {
    constexpr bool STORE = decltype(store_tag)::value;
    const __bf16* As = S;
    const __bf16* Bs = S + PL * TBM * SROW;
    // fragments: lane (row lr, k half lk) holds 8 consecutive k of its row, one 16-byte read per plane
    bf16x8 a[TM][PL], b[TN][PL];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        a[i][p] = *reinterpret_cast<const bf16x8*>(As + p * TBM * SROW + lds_at(wm * (32 * TM) + i * 32 + lr, lk));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < PL; ++p)
        b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * TBN * SROW + lds_at(wn * (32 * TN) + j * 32 + lr, lk));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x16 c = acc[i][j];
        if (PL == 2) {                              // fp16 planes: lo*hi, hi*lo, hi*hi
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][1]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][1]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][0]), __builtin_bit_cast(f16x8, b[j][0]), c, 0, 0, 0);
          acc[i][j] = c;
          continue;
        }
        if (PL == 3) {                              // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PL - 1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][PL - 1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PL / 2], b[j][PL / 2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PL / 2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][PL / 2], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (STORE) {
      store_stage(Snext, va, vb);
      if (PL == 2) {
        constexpr int NMFMA = TM * TN * 3;
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);   // the fragment reads first
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);             // six VALU of the split
          if (q % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS write now and then
        }
      }
      if (PL == 3) {
        constexpr int NMFMA = TM * TN * 6;
        __builtin_amdgcn_sched_group_barrier(0x100, 3 * (TM + TN), 0);   // the fragment reads first
#pragma unroll
        for (int q = 0; q < NMFMA; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);             // five VALU of the split
          if (q % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS write now and then
        }
      }
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  float xa[NA][8], xb[NB][8], ya[NA][8], yb[NB][8];
  __bf16* S1 = S0 + STAGE;
  const int nfull = vec_ok ? (kend - kbeg) / SK : 0;   // stages the unchecked loads may fetch
  load_stage(0, no{}, xa, xb);
  store_stage(S0, xa, xb);
  if (nstages > 1) load_stage(1, no{}, xa, xb);    // x holds stage 1
  __syncthreads();
  // loop invariant: buffer 0 holds stage st (st even), x holds stage st+1
  int st = 0;
  while (st + 3 < nfull) {                         // steady state: whole stages, branch-free loads
    load_stage(st + 2, yes{}, ya, yb);
    multiply(S0, yes{}, S1, xa, xb);               // stage st out of buffer 0, stage st+1 into buffer 1
    __syncthreads();
    load_stage(st + 3, yes{}, xa, xb);
    multiply(S1, yes{}, S0, ya, yb);               // stage st+1 out of buffer 1, stage st+2 into buffer 0
    __syncthreads();
    st += 2;
  }
  while (st + 3 < nstages) {                       // the same with checked loads (a ragged last stage, unaligned rows)
    load_stage(st + 2, no{}, ya, yb);
    multiply(S0, yes{}, S1, xa, xb);
    __syncthreads();
    load_stage(st + 3, no{}, xa, xb);
    multiply(S1, yes{}, S0, ya, yb);
    __syncthreads();
    st += 2;
  }
  const int left = nstages - st;                   // 1, 2 or 3 stages remain
  if (left == 3) {
    load_stage(st + 2, no{}, ya, yb);
    multiply(S0, yes{}, S1, xa, xb);
    __syncthreads();
    multiply(S1, yes{}, S0, ya, yb);
    __syncthreads();
    multiply(S0, no{}, S1, xa, xb);
  } else if (left == 2) {
    multiply(S0, yes{}, S1, xa, xb);
    __syncthreads();
    multiply(S1, no{}, S0, xa, xb);
  } else {
    multiply(S0, no{}, S1, xa, xb);
  }
  if (PL == 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * unscale_a) * unscale_b;
  }
  tile_epilogue<TM, TN, true>(acc, reinterpret_cast<float*>(smem_raw), M, N, m0, n0, wm, wn, C, ldc, epi, partial, bz);
}

// ---------------------------------------------------------------------------------------
// Tall-skinny GEMM for the narrow tower layers: C[M,N] = A[M,K] B, M = batch (huge), K <= 256 and a
// chunk of NC = 16*NT <= 128 output columns per block (grid.y walks the chunks).  Such a layer moves
// far more bytes than flops: the B chunk (K x NC) sits in LDS for the whole kernel, the A rows stream
// straight from global memory into MFMA operands (no LDS staging, no barrier in the loop): lane
// (row l16, k-quad kq) loads A[row][16j + 4kq .. +3] as one float4 and feeds element u to MFMA step u,
// whose B operand is B[16j + 4kq + u][n].  Persistent waves walk 32-row strips.
// B_K: B stored [N,K] (k contiguous), else [K,N].
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NT, bool B_K>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(int M, int N, int K, const float* __restrict__ A,
                                                         int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, Epi epi) {
  epi.drop = resolve_drop(epi.drop);
  extern __shared__ float Ws[];  // [K][SN]
  constexpr int NC = 16 * NT, SN = NC + 4;
  const int n0 = blockIdx.y * NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  for (int q = threadIdx.x; q < K * NC; q += 256) {
    int k, c;
    if (B_K) { c = q / K; k = q % K; } else { k = q / NC; c = q % NC; }
    const int n = n0 + c;
    float v = 0.f;
    if (n < N) v = B_K ? Bm[(int64_t)n * ldb + k] : Bm[(int64_t)k * ldb + n];
    Ws[k * SN + c] = v;
  }
  __syncthreads();
  const int strips = (M + 31) / 32;
  const int ksteps = K / 16;
  constexpr int KC = 2;  // k-steps fetched per batch: 4 float4 loads in flight per lane, 4 more prefetched
  const int nchunks = (ksteps + KC - 1) / KC;
  const bool vec_out = (ldc % 4 == 0) && ((uintptr_t)C % 16 == 0);
  const bool vec_mask = epi.mask_src && (epi.mask_ld % 4 == 0) && ((uintptr_t)epi.mask_src % 16 == 0);
  for (int strip = blockIdx.x * 4 + wave; strip < strips; strip += gridDim.x * 4) {
    const int r0 = strip * 32;
    f32x4 acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[0][t] = acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ra = r0 + l16 < M ? r0 + l16 : M - 1, rb = r0 + 16 + l16 < M ? r0 + 16 + l16 : M - 1;
    const float* a0p = A + (int64_t)ra * lda + 4 * kq;
    const float* a1p = A + (int64_t)rb * lda + 4 * kq;
    float4 cur[KC][2], nxt[KC][2];
#pragma unroll
    for (int s4 = 0; s4 < KC; ++s4) {
      cur[s4][0] = cur[s4][1] = nxt[s4][0] = nxt[s4][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s4 < ksteps) {
        cur[s4][0] = *reinterpret_cast<const float4*>(a0p + 16 * s4);
        cur[s4][1] = *reinterpret_cast<const float4*>(a1p + 16 * s4);
      }
    }
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) {
#pragma unroll
        for (int s4 = 0; s4 < KC; ++s4) {
          const int j = (c + 1) * KC + s4;
          if (j < ksteps) {
            nxt[s4][0] = *reinterpret_cast<const float4*>(a0p + 16 * j);
            nxt[s4][1] = *reinterpret_cast<const float4*>(a1p + 16 * j);
          }
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < KC; ++s4) {
        const int j = c * KC + s4;
        if (j >= ksteps) break;
        const float av0[4] = {cur[s4][0].x, cur[s4][0].y, cur[s4][0].z, cur[s4][0].w};
        const float av1[4] = {cur[s4][1].x, cur[s4][1].y, cur[s4][1].z, cur[s4][1].w};
        const float* wrow = Ws + (16 * j + 4 * kq) * SN + l16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const float b = wrow[u * SN + 16 * t];
            // B fragment as the first operand: the accumulator then holds C transposed, four consecutive
            // columns of one row per lane
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, av0[u], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, av1[u], acc[1][t], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < KC; ++s4) { cur[s4][0] = nxt[s4][0]; cur[s4][1] = nxt[s4][1]; }
    }
    // element r of lane (kq, l16) of tile (rt, t): row r0 + 16 rt + l16, column n0 + 16 t + 4 kq + r
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int m = r0 + 16 * rt + l16;
      if (m >= M) continue;
      // relu/dropout-backward masks of the whole row segment first (independent 16-byte loads)
      float4 mk[NT];
      if (vec_mask) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = n0 + 16 * t + 4 * kq;
          mk[t] = n + 3 < N ? *reinterpret_cast<const float4*>(epi.mask_src + (int64_t)m * epi.mask_ld + n)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int n = n0 + 16 * t + 4 * kq;
        if (n >= N) continue;
        float* cp = C + (int64_t)m * ldc + n;
        float v[4];
        if (vec_mask && n + 3 < N && !epi.bias && !epi.accumulate && !epi.relu && !epi.drop.on && !epi.fm.on) {
          const float mv[4] = {mk[t].x, mk[t].y, mk[t].z, mk[t].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = mv[r] > 0.f ? acc[rt][t][r] * epi.mask_scale : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = n + r < N ? epilogue(epi, acc[rt][t][r], m, n + r, N, cp + r) : 0.f;
        }
        if (vec_out && n + 3 < N) {
          *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) cp[r] = v[r];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                           int M, int N, float* __restrict__ C, int64_t ldc,
                                                           Epi epi) {
  epi.drop = resolve_drop(epi.drop);
  const int64_t total = (int64_t)M * N;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p % N);
    const int m = (int)(p / N);
    float v = 0.f;
    int s = 0;
    for (; s + 7 < splits; s += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partial[(int64_t)(s + u) * total + p];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < splits; ++s) v += partial[(int64_t)s * total + p];
    float* cp = C + (int64_t)m * ldc + n;
    *cp = epilogue(epi, v, m, n, N, cp);
  }
}

// ---------------------------------------------------------------------------------------
// Weight-gradient GEMM for small outputs:  C[M,N] = A^T B,  A stored [K,M], B stored [K,N],
// K = batch (huge), M*N small.  No LDS staging: for a fixed k the M (resp. N) index is
// contiguous in memory, which IS the MFMA operand layout (lane l holds A[k0 + (l>>5)][m0 + (l&31)]),
// so fragments are loaded straight from global memory in 128-byte segments.  Each wave owns a
// (32*TM) x (32*TN) output tile over its own K slice; the four waves of a block cover four
// consecutive K slices and are summed through LDS in a fixed order; blocks write partial slabs
// that a second kernel adds in slab order.  The column sums of B (= the bias gradient) fall out
// of the B fragments for free.
// ---------------------------------------------------------------------------------------
template <int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_stream_kernel(int M, int N, int K, const float* __restrict__ A,
                                                          int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                                          int k_per_wave, float* __restrict__ part,
                                                          float* __restrict__ bias_part) {
  extern __shared__ float red[];  // 2 * (TM*TN*16*64) floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.y * (32 * TM), n0 = blockIdx.x * (32 * TN);
  const int64_t kbeg = ((int64_t)blockIdx.z * 4 + wave) * k_per_wave;
  const int64_t kend = kbeg + k_per_wave < K ? kbeg + k_per_wave : K;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bsum[j] = 0.f;
  bool mok[TM], nok[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) mok[i] = m0 + i * 32 + lr < M;
#pragma unroll
  for (int j = 0; j < TN; ++j) nok[j] = n0 + j * 32 + lr < N;
  // per-tile row pointers walk down K two rows at a time (this lane's k parity is lk)
  const float* ap = A + (kbeg + lk) * lda + m0 + lr;
  const float* bp = Bm + (kbeg + lk) * ldb + n0 + lr;
  const int64_t astep = 2 * lda, bstep = 2 * ldb;
  const int steps = (int)((kend - kbeg + 1) / 2);
  const int last_live = (int)(kend - kbeg) - lk;   // this lane's k offset must stay below this (exclusive: k < kend)
#pragma unroll 8
  for (int it = 0; it < steps; ++it) {
    const bool kok = 2 * it < last_live;
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = (kok && mok[i]) ? ap[i * 32] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      b[j] = (kok && nok[j]) ? bp[j * 32] : 0.f;
      bsum[j] += b[j];
    }
    ap += astep;
    bp += bstep;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  // fixed-order sum over the four waves: (w2 -> w0, w3 -> w1), then (w1 -> w0)
  constexpr int NACC = TM * TN * 16;
  float* slot = red + (wave & 1) * (NACC * 64);
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += slot[((i * TN + j) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r] + red[((i * TN + j) * 16 + r) * 64 + lane];
          const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int n = n0 + j * 32 + lr;
          if (m < M && n < N) part[((int64_t)blockIdx.z * M + m) * N + n] = v;
        }
  }
  if (bias_part && blockIdx.y == 0) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float v = bsum[j] + __shfl_xor(bsum[j], 32);  // k parity halves
      if (lk == 0) red[(wave * TN + j) * 32 + lr] = v;
    }
    __syncthreads();
    if (wave == 0 && lk == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + lr;
        const float v = (red[(0 * TN + j) * 32 + lr] + red[(2 * TN + j) * 32 + lr]) +
                        (red[(1 * TN + j) * 32 + lr] + red[(3 * TN + j) * 32 + lr]);
        if (n < N) bias_part[(int64_t)blockIdx.z * N + n] = v;
      }
    }
  }
}

// out[j] (+)= sum over slabs of part[s*cols + j]; block (64,16): slab phase ty takes slabs ty, ty+16, ...
// eight loads in flight per thread; the sixteen phases are added in a fixed order.  (Hundreds of slabs over
// four phases made this launch as long as the GEMM that fed it.)
// A second problem (part2, cols2, out2) rides in the same launch on the blocks past the first one's.
constexpr int SLAB_PH = 16;
__global__ __launch_bounds__(64 * SLAB_PH) void slab_sum_kernel(const float* __restrict__ part, int nslabs, int cols,
                                                               float* __restrict__ out, int accumulate,
                                                               const float* __restrict__ part2, int cols2,
                                                               float* __restrict__ out2) {
  __shared__ float red[SLAB_PH][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int bx = blockIdx.x;
  const int nb1 = (cols + 63) / 64;
  if (bx >= nb1) { bx -= nb1; part = part2; cols = cols2; out = out2; }
  const int j = bx * 64 + tx;
  float acc = 0.f;
  if (j < cols) {
    int s = ty;
    for (; s + 7 * SLAB_PH < nslabs; s += 8 * SLAB_PH) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(s + SLAB_PH * u) * cols + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < nslabs; s += SLAB_PH) acc += part[(int64_t)s * cols + j];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && j < cols) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < SLAB_PH; ++q) v += red[q][tx];
    out[j] = accumulate ? out[j] + v : v;
  }
}

struct WPlan {
  int tm, tn, tiles_m, tiles_n, nkb, k_per_wave;
};

static WPlan make_wplan(int M, int N, int K) {
  WPlan p;
  p.tm = M > 32 ? 2 : 1;   // 64x64 per wave at most: 64 accumulator registers keep two waves per SIMD
  p.tn = N > 32 ? 2 : 1;
  p.tiles_m = (int)cdiv(M, 32 * p.tm);
  p.tiles_n = (int)cdiv(N, 32 * p.tn);
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  int64_t nkb = cdiv(512, tiles);              // aim at >= 512 blocks
  int64_t max_nkb = cdiv(K, 4 * 32);           // at least 32 samples per wave
  if (nkb > max_nkb) nkb = max_nkb;
  if (nkb < 1) nkb = 1;
  p.k_per_wave = (int)(cdiv(cdiv(K, nkb * 4), 2) * 2);
  p.nkb = (int)cdiv(K, (int64_t)p.k_per_wave * 4);
  return p;
}

struct Plan {
  int bm, bn, tiles_m, tiles_n, splits, k_chunk;
};

static int g_gemm_mode = CTR_GEMM_SPLIT_BF16;
static int g_f16_split = 1;   // calls that bring both operands' maxima take the two-term fp16 split (split mode only)

static Plan make_plan(int M, int N, int K, bool wide_ok) {
  Plan p;
  // narrow outputs get tall tiles so the MFMA work is not spent on padding columns; wide outputs (split kernel
  // only) get a 128 x 256 tile on eight waves, so a row block of A is fetched and split once per 256 columns
  if (N <= 32) { p.bm = 256; p.bn = 32; }
  else if (N <= 64) { p.bm = 256; p.bn = 64; }
  else if (wide_ok && N > 192 && M >= 128 && K >= 512) { p.bm = 128; p.bn = 256; }   // short K: no gain in a step
  else { p.bm = 128; p.bn = 128; }
  p.tiles_m = (int)cdiv(M, p.bm);
  p.tiles_n = (int)cdiv(N, p.bn);
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  int splits = 1;
  if (tiles < 256 && K >= 8 * BK) {
    splits = (int)cdiv(512, tiles);
    const int max_splits = K / (4 * BK);  // keep at least 4 k-steps per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 512) splits = 512;
    if (splits < 1) splits = 1;
  }
  int k_chunk = (int)cdiv(cdiv(K, splits), 32) * 32;  // whole stages of either kernel
  if (k_chunk < BK) k_chunk = BK;
  p.splits = (int)cdiv(K > 0 ? K : 1, k_chunk);
  p.k_chunk = k_chunk;
  return p;
}

}  // namespace ctr

using namespace ctr;


static size_t split_lds_bytes(int tbm, int tbn, int planes = 3) {
  const size_t oper = (size_t)2 * planes * (tbm + tbn) * ctr::SROW * 2;
  const size_t epi = (size_t)8 * 32 * 36 * 4;
  return oper > epi ? oper : epi;
}

extern "C" int ctr_set_gemm_mode(int mode) {
  CTR_REQUIRE(mode == CTR_GEMM_FP32_MFMA || mode == CTR_GEMM_SPLIT_BF16 || mode == CTR_GEMM_BF16,
              "ctr_set_gemm_mode: unknown mode %d", mode);
  g_gemm_mode = mode;
  return CTR_OK;
}

extern "C" int ctr_get_gemm_mode(void) { return g_gemm_mode; }

extern "C" int ctr_set_gemm_f16_split(int on) {
  const int prev = g_f16_split;
  g_f16_split = on ? 1 : 0;
  return prev;
}

// Diagnostic: resident blocks per CU the runtime grants the 128x128 split-bf16 tile kernel (it is written
// for two; one means its two-blocks-per-CU overlap of split work and matrix-core work is not happening).
extern "C" int ctr_gemm_split_blocks_per_cu(void) {
  const size_t lds = split_lds_bytes(128, 128);
  const void* fn = reinterpret_cast<const void*>(&gemm_split_kernel<true, false, 128, 128, 2>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds) != hipSuccess) return -1;
  return nb;
}

namespace ctr {
// *out = max(*out, max |X[r, c]|) over a [rows, cols] view (leading dimension ld): the largest magnitude of a GEMM
// operand, for producers that do not raise it themselves.  V4: 16 bytes per lane.
template <bool V4>
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int cols,
                                                    float* __restrict__ out) {
  constexpr int W = V4 ? 4 : 1;
  const int cw = (cols + W - 1) / W;
  const int64_t total = rows * cw;
  float m = 0.f;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = q / cw;
    const int c = (int)(q - r * cw) * W;
    if (V4) {
      const float4 v = *reinterpret_cast<const float4*>(X + r * ld + c);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      m = fmaxf(m, fabsf(X[r * ld + c]));
    }
  }
  raise_amax(out, m);
}

void launch_absmax(const float* X, int64_t ld, int64_t rows, int cols, float* out, hipStream_t st) {
  const bool v4 = cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X % 16) == 0;
  const int64_t total = rows * (v4 ? cols / 4 : cols);
  const int64_t b = cdiv(total, 256 * 4);
  const unsigned blocks = (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
  if (v4) hipLaunchKernelGGL(absmax_kernel<true>, dim3(blocks), dim3(256), 0, st, X, ld, rows, cols, out);
  else hipLaunchKernelGGL(absmax_kernel<false>, dim3(blocks), dim3(256), 0, st, X, ld, rows, cols, out);
}

// out[k][n] = in[n][k]: a weight matrix stored k-contiguous becomes n-contiguous (32x32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t ldi, int N, int K,
                                                       float* __restrict__ out, int64_t ldo) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (n0 + r < N && k0 + tx < K) tile[r][tx] = in[(int64_t)(n0 + r) * ldi + k0 + tx];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    if (k0 + r < K && n0 + tx < N) out[(int64_t)(k0 + r) * ldo + n0 + tx] = tile[tx][r];
}
}  // namespace ctr

// A k-contiguous B (trans_b) costs the tile kernels 64-byte row pieces at a row stride apart; when B is a
// large square-ish weight matrix (small next to M rows of A) it is turned once into the n-contiguous layout
// first.  Measured: 8192x1024x1024 gains 19%; short K (<= 256) and ragged N (429, 1677) lose to the extra launch.
static bool pretranspose_b(int M, int N, int K) {
  return M >= 2048 && (int64_t)N * K <= (int64_t)1 << 23 && N >= 512 && N % 128 == 0 && K >= 512;
}
static size_t bt_floats(int N, int K) { return (size_t)K * (size_t)((N + 3) / 4 * 4); }

extern "C" size_t ctr_gemm_f32_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // the transposition flags are not known here: the larger of the two tile plans' split-K slab counts
  const Plan pn = make_plan(M, N, K, false), pw = make_plan(M, N, K, true);
  const int splits = pn.splits > pw.splits ? pn.splits : pw.splits;
  size_t need = splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (pretranspose_b(M, N, K)) need = align_up(need, 256) + bt_floats(N, K) * sizeof(float);
  return need;
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
  e.fm.on = 0; e.fm.E = e.fm.cols = 0; e.fm.eshift = -1;
  e.fm.g1 = e.fm.g2 = e.fm.s = e.fm.z = nullptr; e.fm.g1_ld = e.fm.g2_ld = e.fm.z_ld = 0;
  e.amax_out = nullptr;
  Scales sc;
  sc.a_amax = sc.b_amax = nullptr; sc.a_floor = sc.b_floor = 0.f;
  if (epi) {
    e.amax_out = epi->c_amax_out;
    sc.a_amax = epi->a_amax; sc.b_amax = epi->b_amax;
    sc.a_floor = epi->a_amax_floor > 0.f ? epi->a_amax_floor : 0.f;
    sc.b_floor = epi->b_amax_floor > 0.f ? epi->b_amax_floor : 0.f;
    e.bias = epi->bias; e.relu = epi->relu; e.accumulate = epi->accumulate;
    e.drop = make_drop(epi->dropout);
    e.mask_src = epi->mask_src; e.mask_ld = epi->mask_ld; e.mask_scale = epi->mask_scale;
    CTR_REQUIRE(!e.mask_src || e.mask_ld >= N, "ctr_gemm_f32: mask_ld too small");
    if (epi->fm_bwd) {
      const ctr_fm_bwd_t* q = epi->fm_bwd;
      CTR_REQUIRE(q->n_rows > 0 && q->E > 0 && q->E % 4 == 0 && (int64_t)q->n_rows * q->E <= N,
                  "ctr_gemm_f32: fm_bwd needs n_rows*E <= N and E %% 4 == 0");
      CTR_REQUIRE(!q->g_first || q->g_first_ld >= q->n_rows, "ctr_gemm_f32: fm_bwd g_first_ld too small");
      CTR_REQUIRE(!q->g_second || (q->g_second_ld >= q->E && q->s && ((uintptr_t)q->s % 16) == 0 &&
                                   (!q->z || (q->z_ld >= (int64_t)q->n_rows * q->E && q->z_ld % 4 == 0 &&
                                              ((uintptr_t)q->z % 16) == 0))),
                  "ctr_gemm_f32: fm_bwd with g_second needs a 16-byte aligned s and, if given, a 16-byte aligned z with "
                  "z_ld %% 4 == 0");
      e.fm.on = 1; e.fm.E = q->E; e.fm.cols = q->n_rows * q->E;
      e.fm.eshift = -1;
      for (int sh = 2; sh < 31; ++sh)
        if ((1 << sh) == q->E) e.fm.eshift = sh;
      e.fm.g1 = q->g_first; e.fm.g1_ld = q->g_first_ld; e.fm.g2 = q->g_second; e.fm.g2_ld = q->g_second_ld;
      e.fm.s = q->s; e.fm.z = q->z; e.fm.z_ld = q->z_ld;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  // narrow tower layers: the whole B chunk in LDS, A streamed (see skinny_gemm_kernel)
  if (!trans_a && M >= 1024 && K <= 256 && K % 16 == 0 && N <= 256 && lda % 4 == 0 && ((uintptr_t)A % 16) == 0) {
    int nc = N <= 32 ? 32 : N <= 64 ? 64 : 128;
    if ((size_t)K * (nc + 4) * sizeof(float) > 72 * 1024 && nc == 128) nc = 64;
    const size_t lds = (size_t)K * (nc + 4) * sizeof(float);
    const int chunks = (int)cdiv(N, nc);
    // large M: only the narrow (N <= 64) untransposed layers win over the tile kernels (measured); a few thousand
    // rows leave the 128 x 128 tiles a handful of blocks, so every shape that fits goes the strip way there
    if (lds <= 72 * 1024 && chunks == 1 && ((!trans_b && N <= 64) || M < 8192)) {
      const int strips = (int)cdiv(M, 32);
      int blocks = (int)cdiv(strips, 4);
      const int cap = 256 * (lds > 40 * 1024 ? 2 : 4) / chunks;  // resident blocks the LDS allows
      if (blocks > cap) blocks = cap > 0 ? cap : 1;
      const dim3 sgrid(blocks, chunks), sblock(256);
#define SK(NT_, BK_)                                                                                        \
  do {                                                                                                      \
    if (lds > 64 * 1024)                                                                                    \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_kernel<NT_, BK_>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
    hipLaunchKernelGGL((skinny_gemm_kernel<NT_, BK_>), sgrid, sblock, lds, st, M, N, K, A, lda, B, ldb, C,   \
                       ldc, e);                                                                             \
  } while (0)
      if (trans_b) { if (nc == 32) SK(2, true); else if (nc == 64) SK(4, true); else SK(8, true); }
      else { if (nc == 32) SK(2, false); else if (nc == 64) SK(4, false); else SK(8, false); }
#undef SK
      if (e.amax_out) launch_absmax(C, ldc, M, N, e.amax_out, st);   // this kernel does not track its output's maximum
      CTR_LAUNCH_CHECK();
      return CTR_OK;
    }
  }
  static const bool no_wide = getenv("CTR_GEMM_NO_WIDE") != nullptr;   // measurement switch: 128 x 128 tiles everywhere
  const Plan p = make_plan(M, N, K, !no_wide && g_gemm_mode != CTR_GEMM_FP32_MFMA && !trans_a);   // wide tiles lose on A^T B (measured)
  float* partial = nullptr;
  if (p.splits > 1) {
    const size_t need = (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) {
      set_error("ctr_gemm_f32: workspace %zu < %zu", workspace_bytes, need);
      return CTR_EWORKSPACE;
    }
    partial = (float*)workspace;
  }
  if (trans_b && !trans_a && g_gemm_mode != CTR_GEMM_FP32_MFMA && pretranspose_b(M, N, K) && workspace) {
    const size_t off = align_up(partial ? (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float) : 0, 256);
    if (workspace_bytes >= off + bt_floats(N, K) * sizeof(float)) {   // else: the direct path below
      float* Bt = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
      const int64_t ldt = (N + 3) / 4 * 4;
      hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(K, 32), (unsigned)cdiv(N, 32)), dim3(256), 0, st, B,
                         ldb, N, K, Bt, ldt);
      B = Bt;
      ldb = ldt;
      trans_b = 0;
    }
  }
  const bool vecA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const bool vecB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  const dim3 grid(p.tiles_n, p.tiles_m, p.splits), block(256);
  // both operands' largest magnitudes are known: the two-term fp16 split (half the matrix-core work of the bf16 one)
  const bool use_f16 = g_gemm_mode == CTR_GEMM_SPLIT_BF16 && g_f16_split && sc.a_amax && sc.b_amax;
  // measurement switch: extra dynamic LDS per block, i.e. fewer resident blocks per CU
  static const size_t lds_pad = getenv("CTR_GEMM_LDS_PAD") ? (size_t)atoi(getenv("CTR_GEMM_LDS_PAD")) : 0;
  // measurement switches (results are wrong with them): every row of A read from its first row (the operand stream
  // served by the caches), every row of C stored over its first row (no write traffic)
  static const bool dbg_a0 = getenv("CTR_GEMM_DEBUG_A0") != nullptr, dbg_c0 = getenv("CTR_GEMM_DEBUG_C0") != nullptr;
  if (dbg_a0 && !trans_a) lda = 0;
  if (dbg_c0) ldc = 0;
#define LAUNCH_SPLIT(AK, BK_, TBM, TBN, WGM, NW_, PL_)                                                      \
  do {                                                                                                      \
    const size_t lds_ = split_lds_bytes(TBM, TBN, PL_) + lds_pad;                                           \
    if (lds_ > 64 * 1024)                                                                                   \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<AK, BK_, TBM, TBN, WGM, NW_, PL_>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));                  \
    hipLaunchKernelGGL((gemm_split_kernel<AK, BK_, TBM, TBN, WGM, NW_, PL_>), grid, dim3(64 * NW_), lds_, st, M, N, K, \
                       A, lda, vecA, B, ldb, vecB, C, ldc, e, p.k_chunk, partial, sc);                      \
  } while (0)
#define LAUNCH3(AK, BK_, TBM, TBN, WGM)                                                                     \
  do {                                                                                                      \
    if (use_f16) LAUNCH_SPLIT(AK, BK_, TBM, TBN, WGM, 4, 2);                                                \
    else if (g_gemm_mode == CTR_GEMM_SPLIT_BF16) LAUNCH_SPLIT(AK, BK_, TBM, TBN, WGM, 4, 3);                \
    else if (g_gemm_mode == CTR_GEMM_BF16) LAUNCH_SPLIT(AK, BK_, TBM, TBN, WGM, 4, 1);                      \
    else                                                                                                    \
      hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, TBM, TBN, WGM>), grid, block, 0, st, M, N, K, A, lda, vecA, B,  \
                         ldb, vecB, C, ldc, e, p.k_chunk, partial);                                         \
  } while (0)
#define LAUNCH_WIDE(AK, BK_)                                                                                 \
  do {                                                                                                      \
    if (use_f16) LAUNCH_SPLIT(AK, BK_, 128, 256, 2, 8, 2);                                                  \
    else if (g_gemm_mode == CTR_GEMM_BF16) LAUNCH_SPLIT(AK, BK_, 128, 256, 2, 8, 1);                        \
    else LAUNCH_SPLIT(AK, BK_, 128, 256, 2, 8, 3);                                                          \
  } while (0)
#define LAUNCH(AK, BK_)                                    \
  do {                                                     \
    if (p.bn == 32) LAUNCH3(AK, BK_, 256, 32, 4);          \
    else if (p.bn == 64) LAUNCH3(AK, BK_, 256, 64, 4);     \
    else if (p.bn == 256) LAUNCH_WIDE(AK, BK_);            \
    else LAUNCH3(AK, BK_, 128, 128, 2);                    \
  } while (0)
  if (!trans_a && !trans_b) LAUNCH(true, false);
  else if (!trans_a && trans_b) LAUNCH(true, true);
  else if (trans_a && !trans_b) LAUNCH(false, false);
  else LAUNCH(false, true);
#undef LAUNCH
#undef LAUNCH_WIDE
#undef LAUNCH3
#undef LAUNCH_SPLIT
  if (partial) {
    const int64_t total = (int64_t)M * N;
    const int64_t blocks = cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, partial, p.splits, M, N,
                       C, ldc, e);
    if (e.amax_out) launch_absmax(C, ldc, M, N, e.amax_out, st);
  }
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_absmax(const float* X, int64_t ld, int64_t rows, int cols, float* out, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, "ctr_absmax: bad size");
  if (rows == 0 || cols == 0) return CTR_OK;
  CTR_REQUIRE(X && out, "ctr_absmax: null pointer");
  launch_absmax(X, ld, rows, cols, out, (hipStream_t)stream);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

namespace ctr {
// max |.| of tensor blockIdx.y of a flat buffer: 16 bytes per lane, four loads in flight (tensors start 16-byte
// aligned in the flat buffer; a tensor that does not, or its last 1-3 elements, goes element by element)
__global__ __launch_bounds__(256) void absmax_tensors_kernel(const float* __restrict__ flat,
                                                            const int64_t* __restrict__ offsets,
                                                            const int64_t* __restrict__ counts, float* __restrict__ out) {
  const int64_t off = offsets[blockIdx.y], n = counts[blockIdx.y];
  const float* p = flat + off;
  float m = 0.f;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  if (((uintptr_t)p & 15) == 0) {
    const int64_t n4 = n / 4;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    int64_t i = tid;
    for (; i + 3 * nth < n4; i += 4 * nth) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p4[i + u * nth];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (; i < n4; i += nth) {
      const float4 v = p4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t j = 4 * n4 + tid; j < n; j += nth) m = fmaxf(m, fabsf(p[j]));
  } else {
    for (int64_t j = tid; j < n; j += nth) m = fmaxf(m, fabsf(p[j]));
  }
  raise_amax(out + (int64_t)blockIdx.y * CTR_AMAX_FLOATS, m);
}
}  // namespace ctr

extern "C" int ctr_absmax_tensors(const float* flat, const int64_t* offsets, const int64_t* counts, int n_tensors,
                                  float* out, void* stream) {
  CTR_REQUIRE(n_tensors >= 0 && n_tensors <= 65535, "ctr_absmax_tensors: bad tensor count");
  if (n_tensors == 0) return CTR_OK;
  CTR_REQUIRE(flat && offsets && counts && out, "ctr_absmax_tensors: null pointer");
  hipLaunchKernelGGL(absmax_tensors_kernel, dim3(128, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, flat,
                     offsets, counts, out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// Stream path is used when the output is small next to K: M*N <= 16384, N <= 128, K >= 2048 - except a
// several-tile output under a very long K (DIN's attention layers, K = B*T), where its tiles re-read the
// operands and the split-K tile kernel measures 1.7x faster.
static bool wgrad_streams(int M, int N, int K) {
  if ((int64_t)M * N >= 8192 && K >= 262144) return false;
  return (int64_t)M * N <= 16384 && N <= 128 && K >= 2048;
}

// Row chunks of the bias gradient's column sums: a long batch gets 512 blocks (with 128, a 128-column gradient kept
// 16 k threads busy and the pass took 21 us for 33 MB)
static int wgrad_colsum_chunks(int K) { return K < 128 * 64 ? (int)cdiv(K, 64) : K < 32768 ? 128 : 512; }

extern "C" size_t ctr_dense_wgrad_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (wgrad_streams(M, N, K)) {
    const WPlan p = make_wplan(M, N, K);
    return (size_t)p.nkb * ((size_t)M * N + N) * sizeof(float);
  }
  const size_t a = ctr_gemm_f32_workspace_bytes(M, N, K);
  const size_t b = (size_t)wgrad_colsum_chunks(K) * N * sizeof(float);   // column-sum partials
  return a > b ? a : b;
}

namespace ctr {
// column sums of X[B,K] into per-chunk partial rows; V4: 16 bytes per lane (K % 4 == 0, ld % 4 == 0)
template <bool V4>
__global__ __launch_bounds__(256) void wgrad_colsum_partial(const float* __restrict__ X, int64_t ld, int B, int K,
                                                           int rows_per_chunk, float* __restrict__ part) {
  constexpr int W = V4 ? 4 : 1;
  __shared__ float red[4][64 * W];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = (blockIdx.x * 64 + tx) * W;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < B ? r0 + rows_per_chunk : B;
  float acc[W];
#pragma unroll
  for (int u = 0; u < W; ++u) acc[u] = 0.f;
  if (j < K) {
    int64_t r = r0 + ty;
    for (; r + 12 < r1; r += 16) {   // four independent row loads in flight
      if (V4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(X + (r + 4 * u) * ld + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[0] += v[u].x; acc[1 % W] += v[u].y; acc[2 % W] += v[u].z; acc[3 % W] += v[u].w; }
      } else {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = X[(r + 4 * u) * ld + j];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[0] += v[u];
      }
    }
    for (; r < r1; r += 4) {
      if (V4) {
        const float4 v = *reinterpret_cast<const float4*>(X + r * ld + j);
        acc[0] += v.x; acc[1 % W] += v.y; acc[2 % W] += v.z; acc[3 % W] += v.w;
      } else {
        acc[0] += X[r * ld + j];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < W; ++u) red[ty][tx * W + u] = acc[u];
  __syncthreads();
  if (ty == 0 && j < K) {
#pragma unroll
    for (int u = 0; u < W; ++u)
      part[(int64_t)blockIdx.y * K + j + u] =
          (red[0][tx * W + u] + red[1][tx * W + u]) + (red[2][tx * W + u] + red[3][tx * W + u]);
  }
}
}  // namespace ctr

extern "C" int ctr_dense_wgrad(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                               float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  return ctr_dense_wgrad_scaled(M, N, K, A, lda, B, ldb, dW, db, nullptr, 0.f, nullptr, workspace, workspace_bytes,
                                stream);
}

extern "C" int ctr_dense_wgrad_scaled(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                                      float* dW, float* db, const float* a_amax, float a_amax_floor,
                                      const float* b_amax, void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(M > 0 && N > 0 && K > 0, "ctr_dense_wgrad: sizes must be positive");
  CTR_REQUIRE(A && B && dW, "ctr_dense_wgrad: null pointer");
  CTR_REQUIRE(lda >= M && ldb >= N, "ctr_dense_wgrad: leading dimension too small");
  const size_t need = ctr_dense_wgrad_workspace_bytes(M, N, K);
  if (!workspace || workspace_bytes < need) {
    set_error("ctr_dense_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!wgrad_streams(M, N, K)) {
    ctr_epilogue_t ep = {};
    ep.mask_scale = 1.f;
    ep.a_amax = a_amax; ep.a_amax_floor = a_amax_floor; ep.b_amax = b_amax;
    const int rc = ctr_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, dW, N, &ep, workspace, workspace_bytes, stream);
    if (rc != CTR_OK || !db) return rc;
    // bias gradient: fixed-order column sums of B (the slab region is free again: same stream)
    const int nchunks = wgrad_colsum_chunks(K);
    const int rows_per_chunk = (int)cdiv(K, nchunks);
    float* part = (float*)workspace;
    if (N % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)B % 16) == 0)
      hipLaunchKernelGGL(wgrad_colsum_partial<true>, dim3((unsigned)cdiv(N, 256), nchunks), dim3(256), 0, st, B, ldb,
                         K, N, rows_per_chunk, part);
    else
      hipLaunchKernelGGL(wgrad_colsum_partial<false>, dim3((unsigned)cdiv(N, 64), nchunks), dim3(256), 0, st, B, ldb,
                         K, N, rows_per_chunk, part);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)cdiv(N, 64)), dim3(64 * SLAB_PH), 0, st, part, nchunks, N, db, 0,
                       (const float*)nullptr, 0, (float*)nullptr);
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  const WPlan p = make_wplan(M, N, K);
  float* part = (float*)workspace;
  float* bias_part = db ? part + (size_t)p.nkb * M * N : nullptr;
  const dim3 grid(p.tiles_n, p.tiles_m, p.nkb), block(256);
  const size_t lds = (size_t)2 * p.tm * p.tn * 16 * 64 * sizeof(float);
#define WLAUNCH(TM_, TN_)                                                                                  \
  hipLaunchKernelGGL((wgrad_stream_kernel<TM_, TN_>), grid, block, lds, st, M, N, K, A, lda, B, ldb, p.k_per_wave, \
                     part, bias_part)
  if (p.tm == 2 && p.tn == 2) WLAUNCH(2, 2);
  else if (p.tm == 2 && p.tn == 1) WLAUNCH(2, 1);
  else if (p.tm == 1 && p.tn == 2) WLAUNCH(1, 2);
  else WLAUNCH(1, 1);
#undef WLAUNCH
  // one launch adds the dW slabs and, on its last blocks, the bias slabs
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)(cdiv((int64_t)M * N, 64) + (db ? cdiv(N, 64) : 0))), dim3(64 * SLAB_PH), 0,
                     st, part, p.nkb, M * N, dW, 0, (const float*)bias_part, N, db);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
