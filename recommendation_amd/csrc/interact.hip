// Per-sample interaction kernels of PNN (inner / outer product terms) and AFM (pairwise
// products as the reference evaluates them).  One wave per sample; the sample's Z tile
// ([N, E] fp32) is staged in LDS once and every product is formed on chip: neither the
// [B, P, E] projections of PNN nor the [B, N(N-1)/2, E] pair tensor of AFM ever exist in HBM.
#include "common.h"

namespace ctr {

constexpr int WAVES = 4;

// ---------------------------------------------------------------------------------------
// PNN inner-product term (PNN/PNN.py:75-81):  u[b,p,e] = sum_n Z[b,n,e] * w[p,n]
//   forward : usq[b,p] = sum_e u[b,p,e]^2
//   backward: du[b,p,e] = coef(b,p) * u[b,p,e]  ->  dZ[b,n,e] += sum_p du * w[p,n]
//                                                  dw[p,n]  += sum_{b,e} du * Z[b,n,e]
// LDS: w [P*N] shared by the block, then per wave Z [N*E] and u [P*E].
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
// PMT x PNT: number of 16 x 16 dw tiles (rows p, columns n) the backward instance keeps in registers
template <bool BWD, int PMT, int PNT>
__global__ __launch_bounds__(256) void pnn_inner_kernel(const float* __restrict__ z, int64_t z_ld,
                                                       const float* __restrict__ w, int B, int N, int E, int P,
                                                       float* __restrict__ usq, const float* __restrict__ coef,
                                                       int64_t coef_ld, float* __restrict__ dz, int64_t dz_ld,
                                                       float* __restrict__ dw_part) {
  extern __shared__ float lds[];
  float* ws = lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ES = E + 1;  // padded row: lanes that differ in n (or p) at a fixed e land in different banks
  float* Zs = ws + P * N + wave * (N * ES + P * ES);
  float* us = Zs + N * ES;
  for (int i = threadIdx.x; i < P * N; i += 256) ws[i] = w[i];
  const int l16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[BWD ? PMT : 1][BWD ? PNT : 1];  // dw tiles: rows p (16 per tile), columns n
  if (BWD) {
#pragma unroll
    for (int m = 0; m < PMT; ++m)
#pragma unroll
      for (int t = 0; t < PNT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    __syncthreads();  // previous sample's tiles fully consumed (and w staged, first time)
    if (live)
      for (int i = lane; i < N * E; i += 64) Zs[(i / E) * ES + i % E] = z[b * z_ld + i];
    __syncthreads();
    if (live) {
      for (int q = lane; q < P * E; q += 64) {
        const int p = q / E, e = q % E;
        float u0 = 0.f, u1 = 0.f;
        int n = 0;
        for (; n + 1 < N; n += 2) {
          u0 += Zs[n * ES + e] * ws[p * N + n];
          u1 += Zs[(n + 1) * ES + e] * ws[p * N + n + 1];
        }
        if (n < N) u0 += Zs[n * ES + e] * ws[p * N + n];
        const float u = u0 + u1;
        if (!BWD) {
          float sq = u * u;
          for (int o = E / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
          if (e == 0) usq[b * P + p] = sq;
        } else {
          us[p * ES + e] = u * (coef_ld ? coef[b * coef_ld + p] : coef[p]);
        }
      }
    }
    if (BWD) {
      __syncthreads();
      if (live) {
        for (int q = lane; q < N * E; q += 64) {
          const int n = q / E, e = q % E;
          float d = 0.f;
          for (int p = 0; p < P; ++p) d += us[p * ES + e] * ws[p * N + n];
          dz[b * dz_ld + q] += d;
        }
        // dw[p,n] += sum_e du[p,e] Z[n,e] on the matrix cores: rows p, columns n, 4 values of e per step
        for (int e0 = 0; e0 < E; e0 += 4) {
          float av[PMT], bvv[PNT];
#pragma unroll
          for (int m = 0; m < PMT; ++m) av[m] = 16 * m + l16 < P ? us[(16 * m + l16) * ES + e0 + g4] : 0.f;
#pragma unroll
          for (int t = 0; t < PNT; ++t) bvv[t] = 16 * t + l16 < N ? Zs[(16 * t + l16) * ES + e0 + g4] : 0.f;
#pragma unroll
          for (int m = 0; m < PMT; ++m)
            if (16 * m < P) {
#pragma unroll
              for (int t = 0; t < PNT; ++t)
                if (16 * t < N) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bvv[t], acc[m][t], 0, 0, 0);
            }
        }
      }
    }
  }
  if (BWD) {
    const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    // element r of lane (g4, l16) of tile (m, t): p = 16 m + 4 g4 + r, n = 16 t + l16
#pragma unroll
    for (int m = 0; m < PMT; ++m)
#pragma unroll
      for (int t = 0; t < PNT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = 16 * m + 4 * g4 + r, n = 16 * t + l16;
          if (pp < P && n < N) dw_part[row * (P * N) + pp * N + n] = acc[m][t][r];
        }
  }
}

// The same term for E = 16 with every contraction on the matrix cores.  One wave per sample:
//   u = W Z      : A = W fragments (registers, loaded once per wave), B = Z straight from global memory -
//                  lane (k-slot g, column l16) reads Z[4s + g][l16], a contiguous 256 bytes per step;
//   dZ = W^T du  : the accumulator layout of u (lane holds rows p = 16m + 4g + r of column e = l16) IS a B
//                  operand when the contraction index p is walked in the order (m, r) with k-slot g, so du
//                  never moves; A = W^T fragments gathered in that order once per wave;
//   dW += du Z^T : contraction over e needs both operands transposed: du and Z go through a wave-private
//                  LDS tile once per sample.
// PT, NT: 16-row tiles of p and of n.
template <bool BWD, int PT, int NT>
__global__ __launch_bounds__(256, 2) void pnn_inner16_kernel(const float* __restrict__ z, int64_t z_ld,
                                                         const float* __restrict__ w, int B, int N, int P,
                                                         float* __restrict__ usq, const float* __restrict__ coef,
                                                         int64_t coef_ld, float* __restrict__ dz, int64_t dz_ld,
                                                         float* __restrict__ dw_part) {
  constexpr int E = 16, NS = 4 * NT, ES = E + 1;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  float* us = lds + wave * ((16 * PT + 16 * NT) * ES);
  float* Zs = us + 16 * PT * ES;
  float wa[PT][NS];
#pragma unroll
  for (int m = 0; m < PT; ++m)
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int pp = 16 * m + l16, n = 4 * s2 + g;
      wa[m][s2] = (pp < P && n < N) ? w[pp * N + n] : 0.f;
    }
  // W for the dZ product sits in LDS, padded to [16 PT][WS] with zeros (WS = 68: the four k-slots of a
  // fragment read land 16 banks apart); as registers it pushed the backward past 256 and to one wave per SIMD
  constexpr int WS = 68;
  float* Wl = lds + WAVES * ((16 * PT + 16 * NT) * ES);
  f32x4 dwacc[BWD ? PT : 1][BWD ? NT : 1];
  if (BWD) {
    for (int q = threadIdx.x; q < 16 * PT * WS; q += 256) {
      const int pp = q / WS, n = q - pp * WS;
      Wl[q] = (pp < P && n < N) ? w[pp * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < PT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) dwacc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  // the next sample's Z fragments (and this sample's coef and old dZ values) are requested before the matrix
  // work of the current one: a wave runs its samples back to back with the loads a sample ahead
  float zn[NS];
  {
    const int64_t b0 = (int64_t)blockIdx.x * WAVES + wave;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int n = 4 * s2 + g;
      zn[s2] = (b0 < B && n < N) ? z[b0 * z_ld + n * E + l16] : 0.f;
    }
  }
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    float zf[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) zf[s2] = zn[s2];
    {
      const int64_t bn = b + stride;
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        const int n = 4 * s2 + g;
        zn[s2] = (bn < B && n < N) ? z[bn * z_ld + n * E + l16] : 0.f;
      }
    }
    float cf[BWD ? PT : 1][4], dzo[BWD ? NT : 1][4];
    if (BWD) {
#pragma unroll
      for (int m = 0; m < PT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = 16 * m + 4 * g + r;
          cf[m][r] = (live && pp < P) ? (coef_ld ? coef[b * coef_ld + pp] : coef[pp]) : 0.f;
        }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = 16 * t + 4 * g + r;
          dzo[t][r] = (live && n < N) ? dz[b * dz_ld + n * E + l16] : 0.f;
        }
    }
    f32x4 u[PT];
#pragma unroll
    for (int m = 0; m < PT; ++m) u[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int m = 0; m < PT; ++m) u[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][s2], zf[s2], u[m], 0, 0, 0);
    if (!BWD) {
      // after the in-row sums every lane of a row of 16 holds the row's 4 PT totals; lane l16 keeps the one
      // of (m, r) = (l16 / 4, l16 % 4), so the sample's P values leave in one store
      float keep = 0.f;
#pragma unroll
      for (int m = 0; m < PT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sq = u[m][r] * u[m][r];
          sq = dpp_add<DPP_XOR1>(sq);
          sq = dpp_add<DPP_XOR2>(sq);
          sq = dpp_add<DPP_ROR + 4>(sq);
          sq = dpp_add<DPP_ROR + 8>(sq);
          if (l16 == 4 * m + r) keep = sq;
        }
      const int pp = 16 * (l16 >> 2) + 4 * g + (l16 & 3);
      if (live && (l16 >> 2) < PT && pp < P) usq[b * P + pp] = keep;
    } else {
#pragma unroll
      for (int m = 0; m < PT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u[m][r] *= cf[m][r];
        }
      f32x4 dza[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) dza[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < PT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            dza[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(16 * m + 4 * g + r) * WS + 16 * t + l16], u[m][r], dza[t], 0,
                                                          0, 0);
      if (live) {
        float* dr = dz + b * dz_ld;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * t + 4 * g + r;
            if (n < N) dr[n * E + l16] = dzo[t][r] + dza[t][r];
          }
      }
      // the tiles are the wave's own: ordering inside the wave is all they need (no block barrier per sample)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the tiles of the previous sample are fully read
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int m = 0; m < PT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) us[(16 * m + 4 * g + r) * ES + l16] = u[m][r];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) Zs[(4 * s2 + g) * ES + l16] = zf[s2];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e0 = 0; e0 < E; e0 += 4) {
        float av[PT], bv[NT];
#pragma unroll
        for (int m = 0; m < PT; ++m) av[m] = us[(16 * m + l16) * ES + e0 + g];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = Zs[(16 * t + l16) * ES + e0 + g];
#pragma unroll
        for (int m = 0; m < PT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            dwacc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[t], dwacc[m][t], 0, 0, 0);
      }
    }
  }
  if (BWD) {
    const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
#pragma unroll
    for (int m = 0; m < PT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pp = 16 * m + 4 * g + r, n = 16 * t + l16;
          if (pp < P && n < N) dw_part[row * (P * N) + pp * N + n] = dwacc[m][t][r];
        }
  }
}

// Fixed-order sum of the per-wave partial rows; the summed row is split over up to three
// destinations [n0 | n1 | rest].  A block owns 16 columns; its 16 row groups each add every 16th
// row with 8 loads in flight, then the groups are added in order.
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int rows, int cols,
                                                      float* __restrict__ o0, int n0, float* __restrict__ o1,
                                                      int n1, float* __restrict__ o2) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float a = 0.f;
  if (j < cols) {
    int r = rg;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(r + 16 * u) * cols + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; r < rows; r += 16) a += part[(int64_t)r * cols + j];
  }
  red[rg][c] = a;
  __syncthreads();
  if (rg == 0 && j < cols) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][c];
    if (j < n0) o0[j] = t;
    else if (j < n0 + n1) o1[j - n0] = t;
    else o2[j - n0 - n1] = t;
  }
}

// ss[b, e*E + f] = s[b,e] * s[b,f]                                   (PNN/PNN.py:83-84)
__global__ __launch_bounds__(256) void outer_fwd_kernel(const float* __restrict__ s, int B, int E,
                                                       float* __restrict__ ss) {
  const int64_t total = (int64_t)B * E * E;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(p % E);
    const int e = (int)((p / E) % E);
    const int64_t b = p / ((int64_t)E * E);
    ss[p] = s[b * E + e] * s[b * E + f];
  }
}

// ds[b,e] = sum_f (dss[b,e,f] + dss[b,f,e]) * s[b,f]
__global__ __launch_bounds__(256) void outer_bwd_kernel(const float* __restrict__ dss, const float* __restrict__ s,
                                                       int B, int E, float* __restrict__ ds) {
  const int64_t total = (int64_t)B * E;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(p % E);
    const int64_t b = p / E;
    const float* d = dss + b * E * E;
    float a = 0.f;
    for (int f = 0; f < E; ++f) a += (d[e * E + f] + d[f * E + e]) * s[b * E + f];
    ds[p] = a;
  }
}

// x[r,c] = act(x[r,c] + b1[c] + b2[c]) with optional relu and dropout (in place)
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ x, int64_t ld, int rows, int cols,
                                                      const float* __restrict__ b1, const float* __restrict__ b2,
                                                      int relu, Drop d) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    float v = x[r * ld + c];
    if (b1) v += b1[c];
    if (b2) v += b2[c];
    if (relu) v = fmaxf(v, 0.f);
    v *= drop_mul(d, (uint64_t)p);
    x[r * ld + c] = v;
  }
}

// kind 0: out = sqrt(a); kind 1: out = b != 0 ? a / b : 0
__global__ void vec_op_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b,
                              float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 0) out[i] = sqrtf(a[i]);
  else out[i] = b[i] != 0.f ? a[i] / b[i] : 0.f;
}

// ---------------------------------------------------------------------------------------
// AFM pair layer as the reference evaluates it (AFM/AFM.py:60-85): the softmax runs over a
// size-1 axis, so every attention weight is exactly 1 and
//   x[b,e] = sum_{i<j} m_ij * Z[b,i,e] * Z[b,j,e],   m_ij = dropout multiplier of pair (i,j)
// (pair k of sample b uses dropout element b*P + k, pairs in lexicographic order).
// Row form: r_i[e] = sum_{j != i} m_ij Z[j,e];  x = 1/2 sum_i Z_i * r_i;  dZ_i = dx * r_i.
// ---------------------------------------------------------------------------------------
__device__ static inline int64_t pair_index(int i, int j, int N) {  // i < j
  return (int64_t)i * N - (int64_t)i * (i + 1) / 2 + (j - i - 1);
}

template <bool BWD>
__global__ __launch_bounds__(256) void afm_pair_kernel(const float* __restrict__ z, int64_t z_ld, int B, int N,
                                                      int E, Drop d, float* __restrict__ x_out,
                                                      const float* __restrict__ dx, float* __restrict__ dz,
                                                      int64_t dz_ld) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ES = E + 1;
  const int P = N * (N - 1) / 2;
  float* Zs = lds + wave * (2 * N * ES + P);
  float* Rs = Zs + N * ES;
  float* ms = Rs + N * ES;  // [P] dropout multiplier of each pair: hashed once, not once per (i, e)
  const int64_t b = (int64_t)blockIdx.x * WAVES + wave;
  const bool live = b < B;
  if (live) {
    for (int i = lane; i < N * E; i += 64) Zs[(i / E) * ES + i % E] = z[b * z_ld + i];
    for (int k = lane; k < P; k += 64) ms[k] = drop_mul(d, (uint64_t)(b * P + k));
  }
  __syncthreads();
  if (live) {
    for (int q = lane; q < N * E; q += 64) {
      const int i = q / E, e = q % E;
      float r0 = 0.f, r1 = 0.f;
      // pairs (j, i), j < i: index base(j) + i - j - 1;  pairs (i, j), j > i: contiguous from base(i)
      int k = i - 1;
      for (int j = 0; j < i; ++j) {
        r0 += ms[k] * Zs[j * ES + e];
        k += N - j - 2;
      }
      const int base = (int)pair_index(i, i + 1, N);
      for (int j = i + 1; j < N; ++j) r1 += ms[base + j - i - 1] * Zs[j * ES + e];
      const float r = r0 + r1;
      if (BWD) dz[b * dz_ld + q] = dx[b * E + e] * r;
      else Rs[i * ES + e] = Zs[i * ES + e] * r;
    }
  }
  if (!BWD) {
    __syncthreads();
    if (live)
      for (int e = lane; e < E; e += 64) {
        float a = 0.f;
        for (int i = 0; i < N; ++i) a += Rs[i * ES + e];
        x_out[b * E + e] = 0.5f * a;
      }
  }
}

// ---------------------------------------------------------------------------------------
// AFM with attention over the pairs (softmax across the N(N-1)/2 pair scores; the form the AFM
// paper defines — the reference's softmax over a size-1 axis is ctr_afm_pair_* above).
//   p_k = Z_i * Z_j;  t_k = relu(p_k W + b);  e_k = t_k . h;  a = softmax_k(e);  x = sum_k a_k m_k p_k
// One wave per sample, lanes over pairs; Z, W, b, h in LDS; the pair tensor never leaves the chip.
// Backward recomputes the scores and needs per pair only (de_k, c_k = a_k m_k, relu mask bits):
//   dZ_i[e] = sum_{j != i} (c_k dx[e] + de_k G_k[e]) Z_j[e],  G_k[e] = sum_a mask_ka h_a W[e,a]
//   dW[e,a] = h_a sum_k mask_ka de_k p_k[e];  db_a = h_a sum_k mask_ka de_k;  dh_a = sum_k de_k t_ka
// Per-wave parameter partials are added in a fixed order afterwards: bitwise reproducible.
// ---------------------------------------------------------------------------------------
__device__ static inline int pair_base16(int i, int N) { return i * N - i * (i + 1) / 2; }

// The pair layer for E = 16 on the matrix cores.  With per-pair multipliers m_k the sample's pair sum is
//   x[e] = sum_{i<j} m_ij z_i[e] z_j[e] = 1/2 sum_i z_i[e] Y[i][e],   Y = M Z,
// M the symmetric N x N matrix of the multipliers (zero diagonal).  One wave per sample: the 741 hashes of a
// 39-field sample are scattered into a dense M in LDS (both triangles), Y = M Z runs as 16x16x4 fp32 MFMAs
// with the Z fragments straight from global memory, and the accumulator layout (lane = column e, rows i) is
// what both outputs need: forward 1/2 sum_i z_i[e] Y[i][e], backward dZ[i][e] = dx[e] Y[i][e].
// (The scalar form walks N-1 LDS pairs per output element: 13 k cycles per sample against about 2 k here.)
template <bool BWD, int NT>
__global__ __launch_bounds__(256) void afm_pair16_kernel(const float* __restrict__ z, int64_t z_ld, int B, int N,
                                                        Drop d, float* __restrict__ x_out,
                                                        const float* __restrict__ dx, float* __restrict__ dz,
                                                        int64_t dz_ld) {
  constexpr int E = 16, NP = 16 * NT, MS = NP + 1;
  extern __shared__ float lds[];
  const int P = N * (N - 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  uint32_t* ij = reinterpret_cast<uint32_t*>(lds);                  // [P]  i | j << 16, lexicographic pair order
  float* Ms = reinterpret_cast<float*>(ij + P) + wave * (NP * MS);  // this wave's M, rows padded to MS
  for (int i = threadIdx.x; i < N - 1; i += 256) {
    const int base = pair_base16(i, N);
    for (int j = i + 1; j < N; ++j) ij[base + j - i - 1] = (uint32_t)i | (uint32_t)j << 16;
  }
  for (int q = lane; q < NP * MS; q += 64) Ms[q] = 0.f;             // diagonal and padding stay zero for good
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  float zn[4 * NT];
  {
    const int64_t b0 = (int64_t)blockIdx.x * WAVES + wave;
#pragma unroll
    for (int s2 = 0; s2 < 4 * NT; ++s2) {
      const int n = 4 * s2 + g;
      zn[s2] = (b0 < B && n < N) ? z[b0 * z_ld + n * E + l16] : 0.f;
    }
  }
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    float zf[4 * NT];
#pragma unroll
    for (int s2 = 0; s2 < 4 * NT; ++s2) zf[s2] = zn[s2];
    {
      const int64_t bn = b + stride;
#pragma unroll
      for (int s2 = 0; s2 < 4 * NT; ++s2) {
        const int n = 4 * s2 + g;
        zn[s2] = (bn < B && n < N) ? z[bn * z_ld + n * E + l16] : 0.f;
      }
    }
    // rows of Z in the accumulator layout (row 16 t + 4 g + r, column l16) and, backward, dx[e]
    float zr[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * t + 4 * g + r;
        zr[t][r] = (!BWD && live && n < N) ? z[b * z_ld + n * E + l16] : 0.f;
      }
    const float dxe = (BWD && live) ? dx[b * E + l16] : 0.f;
    __syncthreads();   // the previous sample's M is fully read
    if (live)
      for (int k = lane; k < P; k += 64) {
        const uint32_t pr = ij[k];
        const int i = pr & 0xffffu, j = pr >> 16;
        const float m = drop_mul(d, (uint64_t)(b * P + k));
        Ms[i * MS + j] = m;
        Ms[j * MS + i] = m;
      }
    __syncthreads();
    f32x4 y[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) y[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 4 * NT; ++s2)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        y[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ms[(16 * t + l16) * MS + 4 * s2 + g], zf[s2], y[t], 0, 0, 0);
    if (BWD) {
      if (live) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = 16 * t + 4 * g + r;
            if (n < N) dz[b * dz_ld + n * E + l16] = dxe * y[t][r];
          }
      }
    } else {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a += zr[t][r] * y[t][r];
      a = row_swap_sum(a, a);     // the four k-slot groups hold disjoint rows of the same column
      a = half_swap_sum(a, a);
      if (live && g == 0) x_out[b * E + l16] = 0.5f * a;
    }
  }
}

__device__ static inline int pair_base(int i, int N) { return i * N - i * (i + 1) / 2; }

// LDS rows of Z are padded to E + 1 floats: lanes that walk pairs read rows j, j+1, ... at a fixed e,
// and a stride of E floats would put all of them in E/64-th of the banks.
template <int E, bool BWD>
__global__ __launch_bounds__(256, 2) void afm_attn_kernel(const float* __restrict__ z, int64_t z_ld, int B, int N, int A,
                                                      const float* __restrict__ W, const float* __restrict__ bv,
                                                      const float* __restrict__ hv, Drop drop,
                                                      float* __restrict__ x_out, const float* __restrict__ dx,
                                                      float* __restrict__ dz, int64_t dz_ld,
                                                      float* __restrict__ part /*[waves][E*A + 2A]*/) {
  extern __shared__ float lds[];
  constexpr int ES = E + 1;
  constexpr int MT = (E + 15) / 16;  // 16-row tiles of dW over e
  const int P = N * (N - 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g4 = lane >> 4;
  float* Wt = lds;                 // [A*E]  W transposed: Wt[a*E + e]
  float* bl = Wt + E * A;          // [A]
  float* hl = bl + A;              // [A]
  uint32_t* ij = reinterpret_cast<uint32_t*>(hl + A);  // [P]  i | j << 16, lexicographic pair order
  const int per_wave = N * ES + E + (BWD ? 4 * P : P);
  float* Zs = reinterpret_cast<float*>(ij + P) + wave * per_wave;
  float* dxs = Zs + N * ES;        // [E]
  float* sc = dxs + E;             // [P] scores, then attention weights
  float* de = sc + P;              // [P]
  float* ck = de + P;              // [P]
  uint32_t* mk = reinterpret_cast<uint32_t*>(ck + P);  // [P] relu mask bits
  for (int q = threadIdx.x; q < E * A; q += 256) Wt[(q % A) * E + q / A] = W[q];
  for (int q = threadIdx.x; q < A; q += 256) { bl[q] = bv[q]; hl[q] = hv[q]; }
  for (int i = threadIdx.x; i < N - 1; i += 256) {
    const int base = pair_base(i, N);
    for (int j = i + 1; j < N; ++j) ij[base + j - i - 1] = (uint32_t)i | (uint32_t)j << 16;
  }
  // pass-1 operands held for the whole kernel: W^T (A operand: row a = 16 nt + l16, k = e = 4 s + g4)
  // and b, h at this lane's accumulator rows a = 16 nt + 4 g4 + r; zero beyond A
  float wreg[E / 4][2], breg[2][4], hreg[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int s4 = 0; s4 < E / 4; ++s4) wreg[s4][nt] = 16 * nt + l16 < A ? W[(4 * s4 + g4) * A + 16 * nt + l16] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = 16 * nt + 4 * g4 + r;
      breg[nt][r] = a < A ? bv[a] : 0.f;
      hreg[nt][r] = a < A ? hv[a] : 0.f;
    }
  }
  // backward state that lives across the samples of this wave
  f32x4 qacc[BWD ? MT : 1][2];   // Q[e,a] = sum_k mask_ka de_k p_k[e]  (MFMA 16x16 tiles over e x a)
  float racc[2] = {0.f, 0.f};    // lane's share of R_a = sum_k mask_ka de_k, a = 16 nt + l16
  if (BWD) {
#pragma unroll
    for (int m = 0; m < MT; ++m) qacc[m][0] = qacc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    __syncthreads();
    if (live) {
      for (int q = lane; q < N * E; q += 64) Zs[(q / E) * ES + q % E] = z[b * z_ld + q];
      if (BWD && lane < E) dxs[lane] = dx[b * E + lane];
    }
    __syncthreads();
    // pass 1: scores and relu masks, 16 pairs per step on the matrix cores: T^T = W^T P^T with rows a,
    // columns pair, contraction over e; element r of lane (g4, l16) is t[a = 4 g4 + r][pair l16]
    float mx = -INFINITY;
    if (live) {
      for (int k0 = 0; k0 < P; k0 += 16) {
        const int k = k0 + l16;
        const bool ok = k < P;
        const uint32_t pr = ok ? ij[k] : 0u;
        const float* zi = Zs + (pr & 0xffffu) * ES;
        const float* zj = Zs + (pr >> 16) * ES;
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
        float pv[E / 4];
#pragma unroll
        for (int s4 = 0; s4 < E / 4; ++s4) pv[s4] = zi[4 * s4 + g4] * zj[4 * s4 + g4];
#pragma unroll
        for (int s4 = 0; s4 < E / 4; ++s4) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s4][0], pv[s4], c0, 0, 0, 0);
        if (A > 16) {
#pragma unroll
          for (int s4 = 0; s4 < E / 4; ++s4)
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s4][1], pv[s4], c1, 0, 0, 0);
        }
        float ek = 0.f;
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t0 = c0[r] + breg[0][r];
          if (t0 > 0.f) { ek += t0 * hreg[0][r]; bits |= 1u << (4 * g4 + r); }
          const float t1 = c1[r] + breg[1][r];
          if (t1 > 0.f) { ek += t1 * hreg[1][r]; bits |= 1u << (16 + 4 * g4 + r); }
        }
        ek += __shfl_xor(ek, 16);
        ek += __shfl_xor(ek, 32);
        bits |= __shfl_xor(bits, 16);
        bits |= __shfl_xor(bits, 32);
        if (ok && g4 == 0) {
          sc[k] = ek;
          if (BWD) mk[k] = bits;
        }
        if (ok) mx = fmaxf(mx, ek);
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    if (live)
      for (int k = lane; k < P; k += 64) sum += expf(sc[k] - mx);
    sum = wave_sum(sum);
    const float rs = 1.f / sum;
    // pass 2: attention weights; forward pools, backward forms da_k and the softmax dot
    float xacc[BWD ? 1 : E];
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) xacc[e] = 0.f;
    }
    float sdot = 0.f;
    if (live) {
      for (int k = lane; k < P; k += 64) {
        const uint32_t pr = ij[k];
        const float* zi = Zs + (pr & 0xffffu) * ES;
        const float* zj = Zs + (pr >> 16) * ES;
        const float a = expf(sc[k] - mx) * rs;
        const float m = drop_mul(drop, (uint64_t)(b * P + k));
        if (!BWD) {
          const float am = a * m;
#pragma unroll
          for (int e = 0; e < E; ++e) xacc[e] += am * zi[e] * zj[e];
        } else {
          float da = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) da += dxs[e] * zi[e] * zj[e];
          da *= m;
          sc[k] = a;
          de[k] = da;      // staged: da_k
          ck[k] = a * m;
          sdot += a * da;
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) xacc[e] = wave_sum(xacc[e]);
      if (live && lane == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) x_out[b * E + e] = xacc[e];
      }
      continue;
    }
    sdot = wave_sum(sdot);
    if (live)
      for (int k = lane; k < P; k += 64) de[k] = sc[k] * (de[k] - sdot);  // de_k = a_k (da_k - sum a da)
    __syncthreads();
    if (live) {
      // dZ_i[e] = dx[e] (C Z)[i,e] + sum_a h_a W[e,a] (V_a Z)[i,e] with the symmetric N x N matrices
      // C[i,j] = c_k and V_a[i,j] = mask_ka de_k (k = pair of i and j, zero diagonal): products on the
      // matrix cores, rows i (16 per tile), contraction over j (4 per step), columns e.
      for (int et = 0; et < MT; ++et) {
        const int e = 16 * et + l16;
        const bool eok = e < E;
        for (int i0 = 0; i0 < N; i0 += 16) {
          const int i = i0 + l16;
          float outv[4] = {0.f, 0.f, 0.f, 0.f};
          for (int a0 = 0; a0 < A; a0 += 16) {
            f32x4 T[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) T[a] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 Tc = {0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < N; j0 += 4) {
              const int j = j0 + g4;
              const bool ok = i < N && j < N && i != j;
              const int lo = i < j ? i : j, hi = i < j ? j : i;
              const int k = ok ? pair_base(lo, N) + (hi - lo - 1) : 0;
              const float dek = ok ? de[k] : 0.f;
              const uint32_t bits = ok ? mk[k] >> a0 : 0u;
              const float zv = (j < N && eok) ? Zs[j * ES + e] : 0.f;
              Tc = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? ck[k] : 0.f, zv, Tc, 0, 0, 0);
#pragma unroll
              for (int a = 0; a < 16; ++a)  // units at and beyond A have no mask bits and no weights: skipped
                if (a0 + a < A)
                  T[a] = __builtin_amdgcn_mfma_f32_16x16x4f32((bits >> a & 1u) ? dek : 0.f, zv, T[a], 0, 0, 0);
            }
            // element r of lane (g4, l16): row i0 + 4 g4 + r, column e
            if (eok) {
              if (a0 == 0) {
                const float dxe = dxs[e];
#pragma unroll
                for (int r = 0; r < 4; ++r) outv[r] += dxe * Tc[r];
              }
#pragma unroll
              for (int a = 0; a < 16; ++a) {
                const float hwv = a0 + a < A ? hl[a0 + a] * Wt[(a0 + a) * E + e] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) outv[r] += hwv * T[a][r];
              }
            }
          }
          if (eok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ii = i0 + 4 * g4 + r;
              if (ii < N) dz[b * dz_ld + ii * E + e] = outv[r];
            }
          }
        }
      }
      // Q += P^T V on the matrix cores: rows e, columns a, contraction over pairs (4 per step);
      // A operand lane (row l16, k g4), B operand lane (k g4, column l16)
      // 16 pairs per trip, operands of all four MFMA steps fetched together
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (nt == 1 && A <= 16) break;
        f32x4 q[MT];
        float rsum = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) q[m] = qacc[m][nt];
        for (int k0 = 0; k0 < P; k0 += 16) {
          uint32_t pr[4], bt[4];
          float dk[4], pv[4][MT];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u + g4;
            const bool ok = k < P;
            pr[u] = ok ? ij[k] : 0u;
            dk[u] = ok ? de[k] : 0.f;
            bt[u] = ok ? mk[k] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float* zi = Zs + (pr[u] & 0xffffu) * ES;
            const float* zj = Zs + (pr[u] >> 16) * ES;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const int e = 16 * m + l16;
              pv[u][m] = e < E ? zi[e] * zj[e] : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float v = (bt[u] >> (16 * nt + l16) & 1u) ? dk[u] : 0.f;
            rsum += v;
#pragma unroll
            for (int m = 0; m < MT; ++m) q[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[u][m], v, q[m], 0, 0, 0);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) qacc[m][nt] = q[m];
        racc[nt] += rsum;
      }
    }
  }
  if (BWD) {
    // accumulator element r of lane (g4, l16) is Q[e = 16 m + 4 g4 + r][a = 16 nt + l16]
    //   dW = h_a Q;  db_a = h_a R_a;  dh_a = sum_e W[e,a] Q[e,a] + b_a R_a   (relu'(t) t = mask t)
    const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    float* out = part + row * (E * A + 2 * A);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int a = 16 * nt + l16;
      const bool aok = a < A;
      float r = racc[nt];
      r += __shfl_xor(r, 16);
      r += __shfl_xor(r, 32);
      float sh = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int e = 16 * m + 4 * g4 + rr;
          if (aok && e < E) {
            const float qv = qacc[m][nt][rr];
            out[e * A + a] = hl[a] * qv;
            sh += Wt[a * E + e] * qv;
          }
        }
      }
      sh += __shfl_xor(sh, 16);
      sh += __shfl_xor(sh, 32);
      if (aok && g4 == 0) {
        out[E * A + a] = hl[a] * r;
        out[E * A + A + a] = sh + bl[a] * r;
      }
    }
  }
}

static size_t afm_pair_lds(int N, int E) {
  return (size_t)WAVES * (2 * (size_t)N * (E + 1) + (size_t)N * (N - 1) / 2) * 4;
}

static int afm_blocks(int B) {
  int64_t n = cdiv(B, WAVES);
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

constexpr size_t AFM_LDS_MAX = 160 * 1024;  // LDS of one CU

static size_t afm_attn_lds(int N, int E, int A, bool bwd) {
  const int P = N * (N - 1) / 2;
  return ((size_t)E * A + 2 * A + P + (size_t)WAVES * (N * (E + 1) + E + (bwd ? 4 : 1) * P)) * sizeof(float);
}

static unsigned grid_for(int64_t total) {
  const int64_t b = cdiv(total, 256);
  return (unsigned)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

static int pnn_blocks(int B) {
  int64_t n = cdiv(B, WAVES);
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

static bool pow2_le64(int E) { return E >= 4 && E <= 64 && (E & (E - 1)) == 0; }

static size_t pnn_lds_bytes(int N, int E, int P) {
  return ((size_t)P * N + WAVES * ((size_t)N * (E + 1) + (size_t)P * (E + 1))) * 4;
}

}  // namespace ctr

using namespace ctr;

extern "C" int ctr_pnn_inner_fwd(const float* z, int64_t z_ld, const float* w, int B, int N, int E, int P,
                                 float* usq_out, void* stream) {
  CTR_REQUIRE(B >= 0 && N > 0 && P > 0, "ctr_pnn_inner_fwd: bad size");
  CTR_REQUIRE(pow2_le64(E), "ctr_pnn_inner_fwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(pnn_lds_bytes(N, E, P) <= 64 * 1024, "ctr_pnn_inner_fwd: N=%d,E=%d,P=%d exceed the LDS budget", N, E, P);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && w && usq_out && z_ld >= (int64_t)N * E, "ctr_pnn_inner_fwd: null pointer / z_ld");
  if (E == 16 && N <= 64 && P <= 64) {   // every contraction on the matrix cores
    const int pt = P <= 16 ? 1 : P <= 32 ? 2 : 4, nt = (N + 15) / 16;
#define PF(PT_, NT_)                                                                                            \
  hipLaunchKernelGGL((pnn_inner16_kernel<false, PT_, NT_>), dim3(pnn_blocks(B)), dim3(256), 0, (hipStream_t)stream, z, \
                     z_ld, w, B, N, P, usq_out, (const float*)nullptr, (int64_t)0, (float*)nullptr, (int64_t)0,   \
                     (float*)nullptr)
#define PFN(PT_)                                                                   \
  do {                                                                             \
    if (nt == 1) PF(PT_, 1); else if (nt == 2) PF(PT_, 2); else if (nt == 3) PF(PT_, 3); else PF(PT_, 4); \
  } while (0)
    if (pt == 1) PFN(1); else if (pt == 2) PFN(2); else PFN(4);
#undef PFN
#undef PF
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  hipLaunchKernelGGL((pnn_inner_kernel<false, 1, 1>), dim3(pnn_blocks(B)), dim3(256), pnn_lds_bytes(N, E, P),
                     (hipStream_t)stream, z, z_ld, w, B, N, E, P, usq_out, (const float*)nullptr, (int64_t)0,
                     (float*)nullptr, (int64_t)0, (float*)nullptr);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" size_t ctr_pnn_inner_bwd_workspace_bytes(int B, int N, int P) {
  return (size_t)pnn_blocks(B) * WAVES * (size_t)P * N * sizeof(float);
}

extern "C" int ctr_pnn_inner_bwd(const float* z, int64_t z_ld, const float* w, const float* coef, int64_t coef_ld,
                                 int B, int N, int E, int P, float* dz, int64_t dz_ld, float* dw_out,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B > 0 && N > 0 && P > 0, "ctr_pnn_inner_bwd: bad size");
  CTR_REQUIRE(pow2_le64(E), "ctr_pnn_inner_bwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(P <= 64 && N <= 128, "ctr_pnn_inner_bwd: needs product_size <= 64 and N <= 128 (got %d, %d)", P, N);
  CTR_REQUIRE(pnn_lds_bytes(N, E, P) <= 64 * 1024, "ctr_pnn_inner_bwd: sizes exceed the LDS budget");
  CTR_REQUIRE(z && w && coef && dz && dw_out && workspace, "ctr_pnn_inner_bwd: null pointer");
  CTR_REQUIRE(z_ld >= (int64_t)N * E && dz_ld >= (int64_t)N * E, "ctr_pnn_inner_bwd: leading dimension");
  if (workspace_bytes < ctr_pnn_inner_bwd_workspace_bytes(B, N, P)) {
    set_error("ctr_pnn_inner_bwd: workspace %zu < %zu", workspace_bytes, ctr_pnn_inner_bwd_workspace_bytes(B, N, P));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = pnn_blocks(B);
  float* part = (float*)workspace;
  if (E == 16 && N <= 64 && P <= 64) {   // every contraction on the matrix cores
    const int pt = P <= 16 ? 1 : P <= 32 ? 2 : 4, nt = (N + 15) / 16;
    const size_t lds16 = ((size_t)WAVES * (16 * pt + 16 * nt) * 17 + (size_t)16 * pt * 68) * sizeof(float);
#define PB(PT_, NT_)                                                                                            \
  hipLaunchKernelGGL((pnn_inner16_kernel<true, PT_, NT_>), dim3(nb), dim3(256), lds16, st, z, z_ld, w, B, N, P,  \
                     (float*)nullptr, coef, coef_ld, dz, dz_ld, part)
#define PBN(PT_)                                                                   \
  do {                                                                             \
    if (nt == 1) PB(PT_, 1); else if (nt == 2) PB(PT_, 2); else if (nt == 3) PB(PT_, 3); else PB(PT_, 4); \
  } while (0)
    if (pt == 1) PBN(1); else if (pt == 2) PBN(2); else PBN(4);
#undef PBN
#undef PB
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)cdiv((int64_t)P * N, 16)), dim3(256), 0, st, part, nb * WAVES,
                       P * N, dw_out, P * N, (float*)nullptr, 0, (float*)nullptr);
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  const size_t lds = pnn_lds_bytes(N, E, P);
#define PL(MT, NT)                                                                                              \
  hipLaunchKernelGGL((pnn_inner_kernel<true, MT, NT>), dim3(nb), dim3(256), lds, st, z, z_ld, w, B, N, E, P,      \
                     (float*)nullptr, coef, coef_ld, dz, dz_ld, part)
#define PLN(MT)                                                                                                 \
  do {                                                                                                          \
    if (N <= 32) PL(MT, 2);                                                                                     \
    else if (N <= 64) PL(MT, 4);                                                                                \
    else PL(MT, 8);                                                                                             \
  } while (0)
  if (P <= 16) PLN(1);
  else if (P <= 32) PLN(2);
  else PLN(4);
#undef PLN
#undef PL
  hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)cdiv((int64_t)P * N, 16)), dim3(256), 0, st, part, nb * WAVES,
                     P * N, dw_out, P * N, (float*)nullptr, 0, (float*)nullptr);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_outer_fwd(const float* s, int B, int E, float* ss_out, void* stream) {
  CTR_REQUIRE(B >= 0 && E > 0, "ctr_outer_fwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(s && ss_out, "ctr_outer_fwd: null pointer");
  hipLaunchKernelGGL(outer_fwd_kernel, dim3(grid_for((int64_t)B * E * E)), dim3(256), 0, (hipStream_t)stream, s, B,
                     E, ss_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_outer_bwd(const float* dss, const float* s, int B, int E, float* ds_out, void* stream) {
  CTR_REQUIRE(B >= 0 && E > 0, "ctr_outer_bwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(dss && s && ds_out, "ctr_outer_bwd: null pointer");
  hipLaunchKernelGGL(outer_bwd_kernel, dim3(grid_for((int64_t)B * E)), dim3(256), 0, (hipStream_t)stream, dss, s, B,
                     E, ds_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_bias_act(float* x, int64_t ld, int rows, int cols, const float* bias1, const float* bias2,
                            int relu, const ctr_dropout_t* drop, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_bias_act: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(x && ld >= cols, "ctr_bias_act: null pointer / ld");
  hipLaunchKernelGGL(bias_act_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, ld,
                     rows, cols, bias1, bias2, relu, make_drop(drop));
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_vec_op(int kind, const float* a, const float* b, float* out, int n, void* stream) {
  CTR_REQUIRE(n >= 0 && (kind == 0 || kind == 1), "ctr_vec_op: bad argument");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(a && out && (kind == 0 || b), "ctr_vec_op: null pointer");
  hipLaunchKernelGGL(vec_op_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, kind, a, b, out, n);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_afm_pair_fwd(const float* z, int64_t z_ld, int B, int N, int E, const ctr_dropout_t* pair_drop,
                                float* x_out, void* stream) {
  CTR_REQUIRE(B >= 0 && N > 0 && E > 0, "ctr_afm_pair_fwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && x_out && z_ld >= (int64_t)N * E, "ctr_afm_pair_fwd: null pointer / z_ld");
  if (E == 16 && N >= 2 && N <= 64) {   // the pair sums as M Z on the matrix cores
    const int nt = (N + 15) / 16;
    const size_t lds16 = (size_t)(N * (N - 1) / 2) * 4 + (size_t)WAVES * (16 * nt) * (16 * nt + 1) * 4;
#define AP(NT_)                                                                                              \
  do {                                                                                                       \
    if (lds16 > 64 * 1024)                                                                                   \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&afm_pair16_kernel<false, NT_>),             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));                  \
    hipLaunchKernelGGL((afm_pair16_kernel<false, NT_>), dim3(pnn_blocks(B)), dim3(256), lds16, (hipStream_t)stream, z, \
                       z_ld, B, N, make_drop(pair_drop), x_out, (const float*)nullptr, (float*)nullptr, (int64_t)0); \
  } while (0)
    if (nt == 1) AP(1); else if (nt == 2) AP(2); else if (nt == 3) AP(3); else AP(4);
#undef AP
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  CTR_REQUIRE(afm_pair_lds(N, E) <= 64 * 1024, "ctr_afm_pair_fwd: N*E=%d exceeds the LDS budget", N * E);
  hipLaunchKernelGGL(afm_pair_kernel<false>, dim3((unsigned)cdiv(B, WAVES)), dim3(256), afm_pair_lds(N, E),
                     (hipStream_t)stream, z, z_ld, B, N, E, make_drop(pair_drop), x_out, (const float*)nullptr,
                     (float*)nullptr, (int64_t)0);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_afm_pair_bwd(const float* z, int64_t z_ld, const float* dx, int B, int N, int E,
                                const ctr_dropout_t* pair_drop, float* dz, int64_t dz_ld, void* stream) {
  CTR_REQUIRE(B >= 0 && N > 0 && E > 0, "ctr_afm_pair_bwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && dx && dz && z_ld >= (int64_t)N * E && dz_ld >= (int64_t)N * E, "ctr_afm_pair_bwd: null pointer / ld");
  if (E == 16 && N >= 2 && N <= 64) {
    const int nt = (N + 15) / 16;
    const size_t lds16 = (size_t)(N * (N - 1) / 2) * 4 + (size_t)WAVES * (16 * nt) * (16 * nt + 1) * 4;
#define AP(NT_)                                                                                              \
  do {                                                                                                       \
    if (lds16 > 64 * 1024)                                                                                   \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&afm_pair16_kernel<true, NT_>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));                  \
    hipLaunchKernelGGL((afm_pair16_kernel<true, NT_>), dim3(pnn_blocks(B)), dim3(256), lds16, (hipStream_t)stream, z, \
                       z_ld, B, N, make_drop(pair_drop), (float*)nullptr, dx, dz, dz_ld);                    \
  } while (0)
    if (nt == 1) AP(1); else if (nt == 2) AP(2); else if (nt == 3) AP(3); else AP(4);
#undef AP
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  CTR_REQUIRE(afm_pair_lds(N, E) <= 64 * 1024, "ctr_afm_pair_bwd: N*E=%d exceeds the LDS budget", N * E);
  hipLaunchKernelGGL(afm_pair_kernel<true>, dim3((unsigned)cdiv(B, WAVES)), dim3(256), afm_pair_lds(N, E),
                     (hipStream_t)stream, z, z_ld, B, N, E, make_drop(pair_drop), (float*)nullptr, dx, dz, dz_ld);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_afm_attn_fwd(const float* z, int64_t z_ld, int B, int N, int E, int A, const float* W,
                                const float* b, const float* h, const ctr_dropout_t* pair_drop, float* x_out,
                                void* stream) {
  CTR_REQUIRE(B >= 0 && N >= 2 && A >= 1 && A <= 32, "ctr_afm_attn_fwd: need N >= 2 and 1 <= A <= 32");
  CTR_REQUIRE(pow2_le64(E), "ctr_afm_attn_fwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(afm_attn_lds(N, E, A, false) <= AFM_LDS_MAX, "ctr_afm_attn_fwd: N=%d, E=%d exceed the LDS budget", N, E);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && W && b && h && x_out && z_ld >= (int64_t)N * E, "ctr_afm_attn_fwd: null pointer / z_ld");
  const dim3 grid(afm_blocks(B)), block(256);
  const size_t lds = afm_attn_lds(N, E, A, false);
  hipStream_t st = (hipStream_t)stream;
  const Drop d = make_drop(pair_drop);
#define AL(EE)                                                                                                  \
  if (lds > 64 * 1024)                                                                                          \
    CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&afm_attn_kernel<EE, false>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
  hipLaunchKernelGGL((afm_attn_kernel<EE, false>), grid, block, lds, st, z, z_ld, B, N, A, W, b, h, d, x_out,    \
                     (const float*)nullptr, (float*)nullptr, (int64_t)0, (float*)nullptr)
  switch (E) { case 4: AL(4); break; case 8: AL(8); break; case 16: AL(16); break; case 32: AL(32); break; default: AL(64); }
#undef AL
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" size_t ctr_afm_attn_bwd_workspace_bytes(int B, int E, int A) {
  return (size_t)afm_blocks(B) * WAVES * ((size_t)E * A + 2 * A) * sizeof(float);
}

extern "C" int ctr_afm_attn_bwd(const float* z, int64_t z_ld, const float* dx, int B, int N, int E, int A,
                                const float* W, const float* b, const float* h, const ctr_dropout_t* pair_drop,
                                float* dz, int64_t dz_ld, float* dW, float* db, float* dh, void* workspace,
                                size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B > 0 && N >= 2 && A >= 1 && A <= 32, "ctr_afm_attn_bwd: need B > 0, N >= 2 and 1 <= A <= 32");
  CTR_REQUIRE(pow2_le64(E), "ctr_afm_attn_bwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(afm_attn_lds(N, E, A, true) <= AFM_LDS_MAX, "ctr_afm_attn_bwd: N=%d, E=%d exceed the LDS budget", N, E);
  CTR_REQUIRE(z && dx && W && b && h && dz && dW && db && dh && workspace, "ctr_afm_attn_bwd: null pointer");
  CTR_REQUIRE(z_ld >= (int64_t)N * E && dz_ld >= (int64_t)N * E, "ctr_afm_attn_bwd: leading dimension");
  if (workspace_bytes < ctr_afm_attn_bwd_workspace_bytes(B, E, A)) {
    set_error("ctr_afm_attn_bwd: workspace %zu < %zu", workspace_bytes, ctr_afm_attn_bwd_workspace_bytes(B, E, A));
    return CTR_EWORKSPACE;
  }
  const int nb = afm_blocks(B);
  const dim3 grid(nb), block(256);
  const size_t lds = afm_attn_lds(N, E, A, true);
  hipStream_t st = (hipStream_t)stream;
  const Drop d = make_drop(pair_drop);
  float* part = (float*)workspace;
#define AL(EE)                                                                                                  \
  if (lds > 64 * 1024)                                                                                          \
    CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&afm_attn_kernel<EE, true>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
  hipLaunchKernelGGL((afm_attn_kernel<EE, true>), grid, block, lds, st, z, z_ld, B, N, A, W, b, h, d,            \
                     (float*)nullptr, dx, dz, dz_ld, part)
  switch (E) { case 4: AL(4); break; case 8: AL(8); break; case 16: AL(16); break; case 32: AL(32); break; default: AL(64); }
#undef AL
  // fixed-order sums of the per-wave partial rows: [dW (E*A) | db (A) | dh (A)]
  const int cols = E * A + 2 * A;
  hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)cdiv(cols, 16)), dim3(256), 0, st, part, nb * WAVES, cols, dW,
                     E * A, db, A, dh);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
