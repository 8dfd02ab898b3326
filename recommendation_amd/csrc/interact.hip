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
template <bool BWD>
__global__ __launch_bounds__(256) void pnn_inner_kernel(const float* __restrict__ z, int64_t z_ld,
                                                       const float* __restrict__ w, int B, int N, int E, int P,
                                                       float* __restrict__ usq, const float* __restrict__ coef,
                                                       int64_t coef_ld, float* __restrict__ dz, int64_t dz_ld,
                                                       float* __restrict__ dw_part) {
  extern __shared__ float lds[];
  float* ws = lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Zs = ws + P * N + wave * (N * E + P * E);
  float* us = Zs + N * E;
  for (int i = threadIdx.x; i < P * N; i += 256) ws[i] = w[i];
  constexpr int MAXACC = 64;  // dw accumulators per lane (P*N <= 4096)
  float acc[BWD ? MAXACC : 1];
  if (BWD) {
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) acc[i] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    __syncthreads();  // previous sample's tiles fully consumed (and w staged, first time)
    if (live)
      for (int i = lane; i < N * E; i += 64) Zs[i] = z[b * z_ld + i];
    __syncthreads();
    if (live) {
      for (int q = lane; q < P * E; q += 64) {
        const int p = q / E, e = q % E;
        float u = 0.f;
        for (int n = 0; n < N; ++n) u += Zs[n * E + e] * ws[p * N + n];
        if (!BWD) {
          float sq = u * u;
          for (int o = E / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
          if (e == 0) usq[b * P + p] = sq;
        } else {
          us[q] = u * (coef_ld ? coef[b * coef_ld + p] : coef[p]);
        }
      }
    }
    if (BWD) {
      __syncthreads();
      if (live) {
        for (int q = lane; q < N * E; q += 64) {
          const int n = q / E, e = q % E;
          float d = 0.f;
          for (int p = 0; p < P; ++p) d += us[p * E + e] * ws[p * N + n];
          dz[b * dz_ld + q] += d;
        }
#pragma unroll
        for (int i = 0; i < MAXACC; ++i) {
          const int q = lane + 64 * i;
          if (q < P * N) {
            const int p = q / N, n = q % N;
            float d = 0.f;
            for (int e = 0; e < E; ++e) d += us[p * E + e] * Zs[n * E + e];
            acc[i] += d;
          }
        }
      }
    }
  }
  if (BWD) {
    const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
#pragma unroll
    for (int i = 0; i < MAXACC; ++i) {
      const int q = lane + 64 * i;
      if (q < P * N) dw_part[row * (P * N) + q] = acc[i];
    }
  }
}

// fixed-order sum of the per-wave dw partial rows
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int rows, int cols,
                                                      float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += part[(int64_t)r * cols + j];
  out[j] = a;
}

// like rows_sum_kernel, with the summed row split over three destinations [n0 | n1 | rest]
__global__ __launch_bounds__(256) void rows_sum_strided_kernel(const float* __restrict__ part, int rows, int cols,
                                                              float* __restrict__ o0, int n0, float* __restrict__ o1,
                                                              int n1, float* __restrict__ o2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  float a = 0.f;
  for (int r = 0; r < rows; ++r) a += part[(int64_t)r * cols + j];
  if (j < n0) o0[j] = a;
  else if (j < n0 + n1) o1[j - n0] = a;
  else o2[j - n0 - n1] = a;
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
  float* Zs = lds + wave * (2 * N * E);
  float* Rs = Zs + N * E;
  const int64_t b = (int64_t)blockIdx.x * WAVES + wave;
  const bool live = b < B;
  const int64_t P = (int64_t)N * (N - 1) / 2;
  if (live)
    for (int i = lane; i < N * E; i += 64) Zs[i] = z[b * z_ld + i];
  __syncthreads();
  if (live) {
    for (int q = lane; q < N * E; q += 64) {
      const int i = q / E, e = q % E;
      float r = 0.f;
      for (int j = 0; j < N; ++j) {
        if (j == i) continue;
        const int64_t k = i < j ? pair_index(i, j, N) : pair_index(j, i, N);
        r += drop_mul(d, (uint64_t)(b * P + k)) * Zs[j * E + e];
      }
      if (BWD) dz[b * dz_ld + q] = dx[b * E + e] * r;
      else Rs[q] = Zs[q] * r;
    }
  }
  if (!BWD) {
    __syncthreads();
    if (live)
      for (int e = lane; e < E; e += 64) {
        float a = 0.f;
        for (int i = 0; i < N; ++i) a += Rs[i * E + e];
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
__device__ static inline int64_t pair_base(int i, int N) { return (int64_t)i * N - (int64_t)i * (i + 1) / 2; }

template <int E, bool BWD>
__global__ __launch_bounds__(256) void afm_attn_kernel(const float* __restrict__ z, int64_t z_ld, int B, int N, int A,
                                                      const float* __restrict__ W, const float* __restrict__ bv,
                                                      const float* __restrict__ hv, Drop drop,
                                                      float* __restrict__ x_out, const float* __restrict__ dx,
                                                      float* __restrict__ dz, int64_t dz_ld,
                                                      float* __restrict__ part /*[waves][E*A + 2A]*/) {
  extern __shared__ float lds[];
  const int P = N * (N - 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Wl = lds;                 // [E*A]
  float* hW = Wl + E * A;          // [E*A]  h_a * W[e,a]
  float* bl = hW + E * A;          // [A]
  float* hl = bl + A;              // [A]
  const int per_wave = N * E + (BWD ? 4 * P : P);
  float* Zs = hl + A + wave * per_wave;
  float* sc = Zs + N * E;          // [P] scores, then attention weights
  float* de = sc + P;              // [P]
  float* ck = de + P;              // [P]
  uint32_t* mk = reinterpret_cast<uint32_t*>(ck + P);  // [P] relu mask bits
  for (int q = threadIdx.x; q < E * A; q += 256) {
    Wl[q] = W[q];
    hW[q] = W[q] * hv[q % A];
  }
  for (int q = threadIdx.x; q < A; q += 256) { bl[q] = bv[q]; hl[q] = hv[q]; }
  constexpr int MAXW = (E * 32 + 63) / 64;  // dW elements per lane (A <= 32)
  float dWacc[BWD ? MAXW : 1];
  float dha[BWD ? 32 : 1], dba[BWD ? 32 : 1];  // this lane's share of dh_a and of sum_k mask_ka de_k
  if (BWD) {
#pragma unroll
    for (int u = 0; u < MAXW; ++u) dWacc[u] = 0.f;
#pragma unroll
    for (int a = 0; a < 32; ++a) dha[a] = dba[a] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  const int64_t iters = (B + stride - 1) / stride;
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t b = it * stride + (int64_t)blockIdx.x * WAVES + wave;
    const bool live = b < B;
    __syncthreads();
    if (live)
      for (int q = lane; q < N * E; q += 64) Zs[q] = z[b * z_ld + q];
    __syncthreads();
    // pass 1: scores (and relu masks)
    float mx = -INFINITY;
    if (live) {
      int i = 0;
      for (int k = lane; k < P; k += 64) {
        while (k >= pair_base(i + 1, N)) ++i;
        const int j = i + 1 + (int)(k - pair_base(i, N));
        float p[E];
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = Zs[i * E + e] * Zs[j * E + e];
        float ek = 0.f;
        uint32_t bits = 0;
        for (int a = 0; a < A; ++a) {
          float t = bl[a];
#pragma unroll
          for (int e = 0; e < E; ++e) t += p[e] * Wl[e * A + a];
          if (t > 0.f) { ek += t * hl[a]; bits |= 1u << a; }
        }
        sc[k] = ek;
        if (BWD) mk[k] = bits;
        mx = fmaxf(mx, ek);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    if (live)
      for (int k = lane; k < P; k += 64) sum += expf(sc[k] - mx);
    sum = wave_sum(sum);
    const float rs = 1.f / sum;
    // pass 2: attention weights; forward pools, backward forms da_k and the softmax dot
    float xacc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) xacc[e] = 0.f;
    float sdot = 0.f;
    if (live) {
      int i = 0;
      for (int k = lane; k < P; k += 64) {
        while (k >= pair_base(i + 1, N)) ++i;
        const int j = i + 1 + (int)(k - pair_base(i, N));
        const float a = expf(sc[k] - mx) * rs;
        const float m = drop_mul(drop, (uint64_t)(b * P + k));
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < E; ++e) xacc[e] += a * m * Zs[i * E + e] * Zs[j * E + e];
        } else {
          float da = 0.f;
#pragma unroll
          for (int e = 0; e < E; ++e) da += dx[b * E + e] * Zs[i * E + e] * Zs[j * E + e];
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
    // pass 3: de_k = a_k (da_k - sdot); db / dh partial sums stay in this lane's registers
    if (live) {
      int i = 0;
      for (int k = lane; k < P; k += 64) {
        while (k >= pair_base(i + 1, N)) ++i;
        const int j = i + 1 + (int)(k - pair_base(i, N));
        const float dek = sc[k] * (de[k] - sdot);
        de[k] = dek;
        float p[E];
#pragma unroll
        for (int e = 0; e < E; ++e) p[e] = Zs[i * E + e] * Zs[j * E + e];
        const uint32_t bits = mk[k];
#pragma unroll
        for (int a = 0; a < 32; ++a) {
          if (a < A && (bits >> a & 1u)) {
            float t = bl[a];
#pragma unroll
            for (int e = 0; e < E; ++e) t += p[e] * Wl[e * A + a];
            dha[a] += dek * t;
            dba[a] += dek;
          }
        }
      }
    }
    __syncthreads();
    // dZ: lane (i, e) walks j != i
    if (live) {
      for (int q = lane; q < N * E; q += 64) {
        const int i = q / E, e = q % E;
        const float dxe = dx[b * E + e];
        float acc = 0.f;
        for (int j = 0; j < N; ++j) {
          if (j == i) continue;
          const int lo = i < j ? i : j, hi = i < j ? j : i;
          const int k = (int)pair_base(lo, N) + (hi - lo - 1);
          const uint32_t bits = mk[k];
          float g = 0.f;
          for (int a = 0; a < A; ++a)
            if (bits >> a & 1u) g += hW[e * A + a];
          acc += (ck[k] * dxe + de[k] * g) * Zs[j * E + e];
        }
        dz[b * dz_ld + q] = acc;
      }
      // dW: lane-owned (e, a) elements walk all pairs
#pragma unroll
      for (int u = 0; u < MAXW; ++u) {
        const int q = lane + 64 * u;
        if (q < E * A) {
          const int e = q / A, a = q % A;
          int i = 0, j = 1;
          float acc = 0.f;
          for (int k = 0; k < P; ++k) {
            if (mk[k] >> a & 1u) acc += de[k] * Zs[i * E + e] * Zs[j * E + e];
            if (++j == N) { ++i; j = i + 1; }
          }
          dWacc[u] += acc * hl[a];
        }
      }
    }
  }
  if (BWD) {
    const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    float* out = part + row * (E * A + 2 * A);
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
      const int q = lane + 64 * u;
      if (q < E * A) out[q] = dWacc[u];
    }
#pragma unroll
    for (int a = 0; a < 32; ++a) {
      if (a < A) {
        const float sb = wave_sum(dba[a]), sh = wave_sum(dha[a]);
        if (lane == 0) {
          out[E * A + a] = sb * hl[a];
          out[E * A + A + a] = sh;
        }
      }
    }
  }
}

static int afm_blocks(int B) {
  int64_t n = cdiv(B, WAVES);
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

static size_t afm_attn_lds(int N, int E, int A, bool bwd) {
  const int P = N * (N - 1) / 2;
  return ((size_t)2 * E * A + 2 * A + (size_t)WAVES * (N * E + (bwd ? 4 : 1) * P)) * sizeof(float);
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

static size_t pnn_lds_bytes(int N, int E, int P) { return ((size_t)P * N + WAVES * ((size_t)N * E + (size_t)P * E)) * 4; }

}  // namespace ctr

using namespace ctr;

extern "C" int ctr_pnn_inner_fwd(const float* z, int64_t z_ld, const float* w, int B, int N, int E, int P,
                                 float* usq_out, void* stream) {
  CTR_REQUIRE(B >= 0 && N > 0 && P > 0, "ctr_pnn_inner_fwd: bad size");
  CTR_REQUIRE(pow2_le64(E), "ctr_pnn_inner_fwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(pnn_lds_bytes(N, E, P) <= 64 * 1024, "ctr_pnn_inner_fwd: N=%d,E=%d,P=%d exceed the LDS budget", N, E, P);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && w && usq_out && z_ld >= (int64_t)N * E, "ctr_pnn_inner_fwd: null pointer / z_ld");
  hipLaunchKernelGGL(pnn_inner_kernel<false>, dim3(pnn_blocks(B)), dim3(256), pnn_lds_bytes(N, E, P),
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
  CTR_REQUIRE((int64_t)P * N <= 4096, "ctr_pnn_inner_bwd: P*N=%lld exceeds 4096", (long long)P * N);
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
  hipLaunchKernelGGL(pnn_inner_kernel<true>, dim3(nb), dim3(256), pnn_lds_bytes(N, E, P), st, z, z_ld, w, B, N, E,
                     P, (float*)nullptr, coef, coef_ld, dz, dz_ld, part);
  hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)cdiv((int64_t)P * N, 256)), dim3(256), 0, st, part, nb * WAVES,
                     P * N, dw_out);
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
  CTR_REQUIRE((size_t)WAVES * 2 * N * E * 4 <= 64 * 1024, "ctr_afm_pair_fwd: N*E=%d exceeds the LDS budget", N * E);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && x_out && z_ld >= (int64_t)N * E, "ctr_afm_pair_fwd: null pointer / z_ld");
  hipLaunchKernelGGL(afm_pair_kernel<false>, dim3((unsigned)cdiv(B, WAVES)), dim3(256), (size_t)WAVES * 2 * N * E * 4,
                     (hipStream_t)stream, z, z_ld, B, N, E, make_drop(pair_drop), x_out, (const float*)nullptr,
                     (float*)nullptr, (int64_t)0);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_afm_pair_bwd(const float* z, int64_t z_ld, const float* dx, int B, int N, int E,
                                const ctr_dropout_t* pair_drop, float* dz, int64_t dz_ld, void* stream) {
  CTR_REQUIRE(B >= 0 && N > 0 && E > 0, "ctr_afm_pair_bwd: bad size");
  CTR_REQUIRE((size_t)WAVES * 2 * N * E * 4 <= 64 * 1024, "ctr_afm_pair_bwd: N*E=%d exceeds the LDS budget", N * E);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && dx && dz && z_ld >= (int64_t)N * E && dz_ld >= (int64_t)N * E, "ctr_afm_pair_bwd: null pointer / ld");
  hipLaunchKernelGGL(afm_pair_kernel<true>, dim3((unsigned)cdiv(B, WAVES)), dim3(256), (size_t)WAVES * 2 * N * E * 4,
                     (hipStream_t)stream, z, z_ld, B, N, E, make_drop(pair_drop), (float*)nullptr, dx, dz, dz_ld);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_afm_attn_fwd(const float* z, int64_t z_ld, int B, int N, int E, int A, const float* W,
                                const float* b, const float* h, const ctr_dropout_t* pair_drop, float* x_out,
                                void* stream) {
  CTR_REQUIRE(B >= 0 && N >= 2 && A >= 1 && A <= 32, "ctr_afm_attn_fwd: need N >= 2 and 1 <= A <= 32");
  CTR_REQUIRE(pow2_le64(E), "ctr_afm_attn_fwd: E=%d must be a power of two in [4,64]", E);
  CTR_REQUIRE(afm_attn_lds(N, E, A, false) <= 64 * 1024, "ctr_afm_attn_fwd: N=%d, E=%d exceed the LDS budget", N, E);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(z && W && b && h && x_out && z_ld >= (int64_t)N * E, "ctr_afm_attn_fwd: null pointer / z_ld");
  const dim3 grid(afm_blocks(B)), block(256);
  const size_t lds = afm_attn_lds(N, E, A, false);
  hipStream_t st = (hipStream_t)stream;
  const Drop d = make_drop(pair_drop);
#define AL(EE)                                                                                                  \
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
  CTR_REQUIRE(afm_attn_lds(N, E, A, true) <= 64 * 1024, "ctr_afm_attn_bwd: N=%d, E=%d exceed the LDS budget", N, E);
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
  hipLaunchKernelGGL((afm_attn_kernel<EE, true>), grid, block, lds, st, z, z_ld, B, N, A, W, b, h, d,            \
                     (float*)nullptr, dx, dz, dz_ld, part)
  switch (E) { case 4: AL(4); break; case 8: AL(8); break; case 16: AL(16); break; case 32: AL(32); break; default: AL(64); }
#undef AL
  // fixed-order sums of the per-wave partial rows: [dW (E*A) | db (A) | dh (A)]
  const int cols = E * A + 2 * A;
  float* tmp = part;  // rows_sum writes to the destinations directly
  hipLaunchKernelGGL(rows_sum_strided_kernel, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, st, tmp, nb * WAVES, cols,
                     dW, E * A, db, A, dh);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
