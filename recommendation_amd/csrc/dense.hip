// Small dense helpers around the towers: fixed-order column sums (bias / output-weight /
// continuous-embedding gradients), the output unit (row dot), the loss heads and strided
// copies.  All reductions run in a fixed order so results are bitwise reproducible.
#include "common.h"

namespace ctr {

constexpr int CS_ROWS = 64;   // rows per partial chunk (narrow matrices need many chunks to fill the GPU)

// grid (ceil(K/64), nchunks); block (64, 4)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t ld,
                                                            const float* __restrict__ w, int64_t w_ld,
                                                            int w_group, int B, int K, int rows_per_chunk,
                                                            float* __restrict__ part) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < B ? r0 + rows_per_chunk : B;
  float acc = 0.f;
  if (j < K) {
    const int wj = w ? j / w_group : 0;
    int64_t r = r0 + ty;
    for (; r + 28 < r1; r += 32) {   // eight independent row loads in flight, added in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = X[(r + 4 * u) * ld + j];
        if (w) v[u] *= w[(r + 4 * u) * w_ld + wj];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < r1; r += 4) {
      float v = X[r * ld + j];
      if (w) v *= w[r * w_ld + wj];
      acc += v;
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && j < K) part[(int64_t)blockIdx.y * K + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// block (64, 4): row phase ty sums chunks ty, ty+4, ...; the four phases are added in a fixed order
constexpr int CS_PH = 16;   // chunk phases of the final sum: hundreds of partial rows over four phases made this
                            // launch longer than the pass that produced them
__global__ __launch_bounds__(64 * CS_PH) void colsum_final_kernel(const float* __restrict__ part, int nchunks, int K,
                                                                 float* __restrict__ out, int accumulate) {
  __shared__ float red[CS_PH][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (j < K) {
    int c = ty;
    for (; c + 7 * CS_PH < nchunks; c += 8 * CS_PH) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(c + CS_PH * u) * K + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; c < nchunks; c += CS_PH) acc += part[(int64_t)c * K + j];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && j < K) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < CS_PH; ++q) v += red[q][tx];
    out[j] = accumulate ? out[j] + v : v;
  }
}

__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ X, int64_t ld,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias, int B, int K,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += X[b * ld + k] * w[k];
  acc = wave_sum(acc);
  if (lane == 0) out[b] = acc + (bias ? bias[0] : 0.f);
}

// The same for rows of a few hundred floats at most: a row per 16 lanes (four rows per wave instead of one wave per
// row, whose lanes past K / 64 * 64 idle), the sum inside the 16-lane row on the data-parallel path.
__global__ __launch_bounds__(256) void rowdot_fwd_r16_kernel(const float* __restrict__ X, int64_t ld,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, int B, int K,
                                                            float* __restrict__ out) {
  const int l16 = threadIdx.x & 15;
  const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t bc = b < B ? b : (int64_t)B - 1;       // every lane takes part in the row sums
  float acc = 0.f;
  for (int k = l16; k < K; k += 16) acc += X[bc * ld + k] * w[k];
  acc = dpp_add<DPP_XOR1>(acc);
  acc = dpp_add<DPP_XOR2>(acc);
  acc = dpp_add<DPP_ROR + 4>(acc);
  acc = dpp_add<DPP_ROR + 8>(acc);
  if (l16 == 0 && b < B) out[b] = acc + (bias ? bias[0] : 0.f);
}

__global__ __launch_bounds__(256) void rowdot_bwd_x_kernel(const float* __restrict__ g,
                                                          const float* __restrict__ w, int B, int K,
                                                          float* __restrict__ dX, int64_t ld, int accumulate) {
  const int64_t total = (int64_t)B * K;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(p % K);
    const int64_t b = p / K;
    const float v = g[b] * w[k];
    float* d = dX + b * ld + k;
    *d = accumulate ? *d + v : v;
  }
}

__device__ static inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ static inline float sce_(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }

// Pass 1: 256 samples per block -> y_out, dx and one partial loss sum per block.
// Pass 2: one block adds the partials in a fixed order.
__global__ __launch_bounds__(256) void loss_head_kernel(int kind, const float* __restrict__ x,
                                                       const float* __restrict__ y, int B, float grad_scale,
                                                       float* __restrict__ y_out, float* __restrict__ dx,
                                                       float* __restrict__ part) {
  __shared__ float red[256];
  const float invB = grad_scale / (float)B;
  const float eps = 1e-7f;
  float l = 0.f;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B) {
    const float xv = x[b], yv = y ? y[b] : 0.f;
    float p, d;
    if (kind == CTR_LOSS_SIGMOID_LOGLOSS || kind == CTR_LOSS_IDENTITY_LOGLOSS) {
      p = kind == CTR_LOSS_SIGMOID_LOGLOSS ? sigmoidf_(xv) : xv;
      l = -yv * logf(p + eps) - (1.f - yv) * logf(1.f - p + eps);
      d = -yv / (p + eps) + (1.f - yv) / (1.f - p + eps);
      if (kind == CTR_LOSS_SIGMOID_LOGLOSS) d *= p * (1.f - p);
    } else if (kind == CTR_LOSS_SIGMOID_SCE_PROB) {
      p = sigmoidf_(xv);
      l = sce_(p, yv);
      d = (sigmoidf_(p) - yv) * p * (1.f - p);
    } else if (kind == CTR_LOSS_SCE_LOGIT) {
      p = sigmoidf_(xv);
      l = sce_(xv, yv);
      d = p - yv;
    } else {  // CTR_LOSS_PROB_SCE
      p = xv;
      l = sce_(xv, yv);
      d = sigmoidf_(xv) - yv;
    }
    if (y_out) y_out[b] = p;
    if (dx) dx[b] = d * invB;
  }
  if (!part) return;
  red[threadIdx.x] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// ESMM head (ESMM/ESMM.py:95-98): y_out = sigmoid(a) * sigmoid(c); loss = sce(labels=y, logits=y_out)
__global__ __launch_bounds__(256) void esmm_head_kernel(const float* __restrict__ a, const float* __restrict__ c,
                                                       const float* __restrict__ y, int B, float grad_scale,
                                                       float* __restrict__ y_out, float* __restrict__ da,
                                                       float* __restrict__ dc, float* __restrict__ part) {
  __shared__ float red[256];
  const float invB = grad_scale / (float)B;
  float l = 0.f;
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B) {
    const float sa = sigmoidf_(a[b]), sc = sigmoidf_(c[b]);
    const float p = sa * sc;
    if (y_out) y_out[b] = p;
    if (y) {
      const float yv = y[b];
      l = sce_(p, yv);
      const float d = (sigmoidf_(p) - yv) * invB;
      if (da) da[b] = d * p * (1.f - sa);
      if (dc) dc[b] = d * p * (1.f - sc);
    }
  }
  if (!part) return;
  red[threadIdx.x] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ part, int n, int B,
                                                        float* __restrict__ loss_out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = red[0] / (float)B;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int64_t src_ld,
                                                    float* __restrict__ dst, int64_t dst_ld, int rows,
                                                    int cols, int accumulate) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    const float v = src[r * src_ld + c];
    float* d = dst + r * dst_ld + c;
    *d = accumulate ? *d + v : v;
  }
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(float* __restrict__ x, int64_t ld, int rows,
                                                           int cols, Drop d) {
  d = resolve_drop(d);
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    x[r * ld + c] *= drop_mul(d, (uint64_t)p);
  }
}

// Two column blocks under two dropouts; LW = log2 of the lanes given to a row (>= cols_a + cols_b), so the row and
// column of a thread are a shift and a mask.
__global__ __launch_bounds__(256) void dropout_apply2_kernel(float* __restrict__ x, int64_t ld, int rows, int cols_a,
                                                            Drop da, int cols_b, Drop db, int LW) {
  da = resolve_drop(da); db = resolve_drop(db);
  const int c = threadIdx.x & ((1 << LW) - 1);
  const int rstep = 256 >> LW;
  if (c >= cols_a + cols_b) return;
  const bool first = c < cols_a;
  for (int64_t r = (int64_t)blockIdx.x * rstep + (threadIdx.x >> LW); r < rows; r += (int64_t)gridDim.x * rstep) {
    const float k = first ? drop_mul(da, (uint64_t)r * cols_a + c) : drop_mul(db, (uint64_t)r * cols_b + (c - cols_a));
    x[r * ld + c] *= k;
  }
}

// g *= (act > 0 ? scale : 0): backward of relu followed by dropout, given the stored output
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ g, int64_t g_ld,
                                                       const float* __restrict__ act, int64_t act_ld, int rows,
                                                       int cols, float scale) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    float* d = g + r * g_ld + c;
    *d = act[r * act_ld + c] > 0.f ? *d * scale : 0.f;
  }
}

static unsigned grid_for(int64_t total) {
  const int64_t b = cdiv(total, 256);
  return (unsigned)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

static int cs_chunks(int B) {
  int64_t n = cdiv(B, CS_ROWS);
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

}  // namespace ctr

using namespace ctr;

extern "C" size_t ctr_colsum_workspace_bytes(int B, int K) {
  return (size_t)cs_chunks(B) * (size_t)(K > 0 ? K : 1) * sizeof(float);
}

extern "C" int ctr_colsum(const float* X, int64_t ld, const float* w, int64_t w_ld, int w_group, int B,
                          int K, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream) {
  CTR_REQUIRE(B >= 0 && K >= 0, "ctr_colsum: bad size");
  if (K == 0) return CTR_OK;
  CTR_REQUIRE(X && out && workspace, "ctr_colsum: null pointer");
  CTR_REQUIRE(!w || w_group > 0, "ctr_colsum: w_group must be positive");
  if (workspace_bytes < ctr_colsum_workspace_bytes(B, K)) {
    set_error("ctr_colsum: workspace %zu < %zu", workspace_bytes, ctr_colsum_workspace_bytes(B, K));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = cs_chunks(B);
  const int rows_per_chunk = (int)cdiv(B > 0 ? B : 1, nchunks);
  float* part = (float*)workspace;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(K, 64), nchunks), dim3(256), 0, st, X, ld, w,
                     w_ld, w_group, B, K, rows_per_chunk, part);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(K, 64)), dim3(64 * CS_PH), 0, st, part, nchunks, K,
                     out, accumulate);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_rowdot_fwd(const float* X, int64_t ld, const float* w, const float* bias, int B, int K,
                              float* out, void* stream) {
  CTR_REQUIRE(B >= 0 && K >= 0, "ctr_rowdot_fwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(X && w && out, "ctr_rowdot_fwd: null pointer");
  if (K <= 512)
    hipLaunchKernelGGL(rowdot_fwd_r16_kernel, dim3((unsigned)cdiv(B, 16)), dim3(256), 0, (hipStream_t)stream, X, ld,
                       w, bias, B, K, out);
  else
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, X, ld,
                       w, bias, B, K, out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_rowdot_bwd_x(const float* g, const float* w, int B, int K, float* dX, int64_t ld,
                                int accumulate, void* stream) {
  CTR_REQUIRE(B >= 0 && K >= 0, "ctr_rowdot_bwd_x: bad size");
  if ((int64_t)B * K == 0) return CTR_OK;
  CTR_REQUIRE(g && w && dX, "ctr_rowdot_bwd_x: null pointer");
  hipLaunchKernelGGL(rowdot_bwd_x_kernel, dim3(grid_for((int64_t)B * K)), dim3(256), 0, (hipStream_t)stream,
                     g, w, B, K, dX, ld, accumulate);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

namespace ctr {
constexpr int UB_KMAX = 255;             // widest unit of the thread-per-column kernel (+ bias slot)
constexpr int UB_KWIDE = 4096;           // widest unit of the columns-per-thread kernel
constexpr int UB_BLOCKS = 1024;
constexpr int UB_U = 8;                  // steps in flight per thread

// Backward of the output unit y = X w + b in one pass over X (DeepFM/DeepFM.py:82-88):
//   dX[b,k] = dl[b] * w[k]  (times mask_scale * [X[b,k] > 0] for k >= mask_from: the relu + dropout of the
//   tower's last layer, whose output is that column range of X),  dw[k] = sum_b X[b,k] dl[b],  db = sum_b dl[b].
// A block owns a contiguous range of samples; thread t works on column t % K of sample row t / K of each step
// (256 / K samples per step, so a step reads and writes one contiguous piece of X and dX), keeps the dw term
// of its column in a register, and the block folds its row slots in LDS into one partial row.
__global__ __launch_bounds__(256) void unit_bwd_kernel(int B, int K, const float* __restrict__ dl,
                                                      const float* __restrict__ w, const float* __restrict__ X,
                                                      int64_t ldx, float* __restrict__ dX, int64_t lddx,
                                                      int mask_from, float mask_scale, int rows_per_block,
                                                      float* __restrict__ part, int keep_a_cols, Drop keep_a,
                                                      int keep_b_cols, Drop keep_b, float* __restrict__ amax_out) {
  keep_a = resolve_drop(keep_a); keep_b = resolve_drop(keep_b);
  float amx = 0.f;                                  // max |dX| over the masked columns (amax_out)
  __shared__ float red[256], dred[256];
  const int rps = 256 / K;                          // sample rows per step
  const int r = threadIdx.x / K, k = threadIdx.x - r * K;
  const bool on = r < rps;
  const float wv = on ? w[k] : 0.f;
  const bool masked = k >= mask_from;
  // columns [0, keep_a_cols) and the keep_b_cols after them carry the keep masks of two dropouts (what
  // ctr_dropout_apply2 would put on dX afterwards)
  const bool in_a = keep_a.on && k < keep_a_cols;
  const bool in_b = keep_b.on && k >= keep_a_cols && k < keep_a_cols + keep_b_cols;
  const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t b1 = b0 + rows_per_block < B ? b0 + rows_per_block : B;
  float acc = 0.f, dsum = 0.f;
  if (on) {
    for (int64_t b = b0 + r; b < b1; b += UB_U * rps) {
      float x[UB_U], d[UB_U];
#pragma unroll
      for (int u = 0; u < UB_U; ++u) {
        const int64_t bb = b + u * rps;
        const bool live = bb < b1;
        x[u] = live ? X[bb * ldx + k] : 0.f;
        d[u] = live ? dl[bb] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UB_U; ++u) {
        const int64_t bb = b + u * rps;
        acc += x[u] * d[u];
        if (k == 0) dsum += d[u];
        float g = d[u] * wv;
        if (masked) g = x[u] > 0.f ? g * mask_scale : 0.f;
        if (in_a) g *= drop_mul(keep_a, (uint64_t)bb * keep_a_cols + k);
        else if (in_b) g *= drop_mul(keep_b, (uint64_t)bb * keep_b_cols + (k - keep_a_cols));
        if (bb < b1) {
          dX[bb * lddx + k] = g;
          if (masked) amx = fmaxf(amx, fabsf(g));
        }
      }
    }
  }
  if (amax_out) raise_amax(amax_out, amx);
  red[threadIdx.x] = acc;
  dred[threadIdx.x] = dsum;
  __syncthreads();
  if ((int)threadIdx.x <= K) {
    float v = 0.f;
    if ((int)threadIdx.x < K) {
      for (int q = 0; q < rps; ++q) v += red[q * K + threadIdx.x];
    } else {
      for (int q = 0; q < rps; ++q) v += dred[q * K];
    }
    part[(int64_t)blockIdx.x * (K + 1) + threadIdx.x] = v;
  }
}

// The same for a wide unit (K > 255, e.g. DCN's [x_L | tower] stack): thread t owns columns t, t + 256, ...
// (NC of them) of every sample row of the block's range; two rows in flight.
template <int NC>
__global__ __launch_bounds__(256) void unit_bwd_wide_kernel(int B, int K, const float* __restrict__ dl,
                                                           const float* __restrict__ w, const float* __restrict__ X,
                                                           int64_t ldx, float* __restrict__ dX, int64_t lddx,
                                                           int mask_from, float mask_scale, int rows_per_block,
                                                           float* __restrict__ part, float* __restrict__ amax_out) {
  const int t = threadIdx.x;
  float amx = 0.f;                                  // max |dX| over the masked columns (amax_out)
  float wv[NC], acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = t + 256 * c;
    wv[c] = k < K ? w[k] : 0.f;
    acc[c] = 0.f;
  }
  const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t b1 = b0 + rows_per_block < B ? b0 + rows_per_block : B;
  float dsum = 0.f;
  for (int64_t b = b0; b < b1; b += 2) {
    const bool two = b + 1 < b1;
    const float d0 = dl[b], d1 = two ? dl[b + 1] : 0.f;
    dsum += d0 + d1;
    float x0[NC], x1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = t + 256 * c;
      x0[c] = k < K ? X[b * ldx + k] : 0.f;
      x1[c] = (two && k < K) ? X[(b + 1) * ldx + k] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = t + 256 * c;
      if (k < K) {
        acc[c] += x0[c] * d0;
        acc[c] += x1[c] * d1;
        float g0 = d0 * wv[c], g1 = d1 * wv[c];
        if (k >= mask_from) {
          g0 = x0[c] > 0.f ? g0 * mask_scale : 0.f;
          g1 = x1[c] > 0.f ? g1 * mask_scale : 0.f;
          amx = fmaxf(amx, fmaxf(fabsf(g0), fabsf(g1)));         // g1 is zero past the last row
        }
        dX[b * lddx + k] = g0;
        if (two) dX[(b + 1) * lddx + k] = g1;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = t + 256 * c;
    if (k < K) part[(int64_t)blockIdx.x * (K + 1) + k] = acc[c];
  }
  if (t == 0) part[(int64_t)blockIdx.x * (K + 1) + K] = dsum;   // every thread walked the same dl values
  if (amax_out) raise_amax(amax_out, amx);
}

// out[k] = sum over the partial rows in a fixed order; column K is the bias gradient
__global__ __launch_bounds__(1024) void unit_bwd_final_kernel(const float* __restrict__ part, int nrows, int K,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 16 row phases
  const int k = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (k <= K) {
    int r = ty;
    for (; r + 7 * 16 < nrows; r += 8 * 16) {               // eight loads in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(r + 16 * u) * (K + 1) + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < nrows; r += 16) acc += part[(int64_t)r * (K + 1) + k];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && k <= K) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][tx];
    if (k < K) dw[k] = v;
    else if (db) db[0] = v;
  }
}
}  // namespace ctr

static int unit_bwd_rows_per_block(int B, int K) {
  const int64_t rows = cdiv(B > 0 ? B : 1, UB_BLOCKS);
  const int lo = K <= UB_KMAX ? 32 : 8;   // a wide row keeps a block busy on its own
  return (int)(rows < lo ? lo : rows);
}
static int unit_bwd_blocks(int B, int K) { return (int)cdiv(B > 0 ? B : 1, unit_bwd_rows_per_block(B, K)); }

extern "C" int ctr_unit_bwd_supported(int K) { return K >= 1 && K <= UB_KWIDE; }

extern "C" size_t ctr_unit_bwd_workspace_bytes(int B, int K) {
  return (size_t)unit_bwd_blocks(B, K) * (size_t)(K + 1) * sizeof(float);
}

extern "C" int ctr_unit_bwd(int B, int K, const float* dl, const float* w, const float* X, int64_t ldx, float* dX,
                            int64_t lddx, int mask_from, float mask_scale, float* dw, float* db, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return ctr_unit_bwd_keep(B, K, dl, w, X, ldx, dX, lddx, mask_from, mask_scale, 0, nullptr, 0, nullptr, nullptr, dw,
                           db, workspace, workspace_bytes, stream);
}

extern "C" int ctr_unit_bwd_keep(int B, int K, const float* dl, const float* w, const float* X, int64_t ldx, float* dX,
                                 int64_t lddx, int mask_from, float mask_scale, int keep_a_cols,
                                 const ctr_dropout_t* keep_a, int keep_b_cols, const ctr_dropout_t* keep_b,
                                 float* masked_amax_out, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(keep_a_cols >= 0 && keep_b_cols >= 0 && keep_a_cols + keep_b_cols <= K,
              "ctr_unit_bwd_keep: keep blocks wider than K");
  const Drop ka = make_drop(keep_a), kb = make_drop(keep_b);
  CTR_REQUIRE(B > 0 && ctr_unit_bwd_supported(K), "ctr_unit_bwd: need B > 0 and 1 <= K <= %d", UB_KWIDE);
  CTR_REQUIRE(dl && w && X && dX && dw && workspace, "ctr_unit_bwd: null pointer");
  CTR_REQUIRE(ldx >= K && lddx >= K, "ctr_unit_bwd: leading dimension < K");
  if (workspace_bytes < ctr_unit_bwd_workspace_bytes(B, K)) {
    set_error("ctr_unit_bwd: workspace %zu < %zu", workspace_bytes, ctr_unit_bwd_workspace_bytes(B, K));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = unit_bwd_blocks(B, K);
  float* part = (float*)workspace;
  if (mask_from < 0 || mask_from > K) mask_from = K;
  const int rpb = unit_bwd_rows_per_block(B, K);
  if (K <= UB_KMAX) {
    hipLaunchKernelGGL(unit_bwd_kernel, dim3(nb), dim3(256), 0, st, B, K, dl, w, X, ldx, dX, lddx, mask_from, mask_scale,
                       rpb, part, keep_a_cols, ka, keep_b_cols, kb, masked_amax_out);
  } else {
#define UW(NC_)                                                                                                  \
  hipLaunchKernelGGL(unit_bwd_wide_kernel<NC_>, dim3(nb), dim3(256), 0, st, B, K, dl, w, X, ldx, dX, lddx, mask_from, \
                     mask_scale, rpb, part, masked_amax_out)
    const int nc = (int)cdiv(K, 256);
    if (nc <= 2) UW(2); else if (nc <= 4) UW(4); else if (nc <= 8) UW(8); else if (nc <= 12) UW(12); else UW(16);
#undef UW
  }
  hipLaunchKernelGGL(unit_bwd_final_kernel, dim3((unsigned)cdiv(K + 1, 64)), dim3(1024), 0, st, part, nb, K, dw, db);
  CTR_LAUNCH_CHECK();
  // the wide kernel leaves the keep masks to a pass of their own
  if (K > UB_KMAX && (ka.on || kb.on))
    return ctr_dropout_apply2(dX, lddx, B, keep_a_cols, keep_a, keep_b_cols, keep_b, stream);
  return CTR_OK;
}

extern "C" size_t ctr_loss_head_workspace_bytes(int B) { return (size_t)cdiv(B > 0 ? B : 1, 256) * sizeof(float); }

extern "C" int ctr_loss_head(int kind, const float* x, const float* y, int B, float grad_scale, float* y_out,
                             float* loss_out, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(kind >= 0 && kind <= CTR_LOSS_PROB_SCE, "ctr_loss_head: unknown kind %d", kind);
  CTR_REQUIRE(B > 0 && x, "ctr_loss_head: bad argument");
  CTR_REQUIRE(y || (!loss_out && !dx), "ctr_loss_head: labels are required for loss / gradient");
  const int nb = (int)cdiv(B, 256);
  float* part = nullptr;
  if (loss_out) {
    if (!workspace || workspace_bytes < ctr_loss_head_workspace_bytes(B)) {
      set_error("ctr_loss_head: workspace %zu < %zu", workspace_bytes, ctr_loss_head_workspace_bytes(B));
      return CTR_EWORKSPACE;
    }
    part = (float*)workspace;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_head_kernel, dim3(nb), dim3(256), 0, st, kind, x, y, B, grad_scale, y_out, dx, part);
  if (loss_out) hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, part, nb, B, loss_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols,
                          int accumulate, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_copy2d: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(src && dst, "ctr_copy2d: null pointer");
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream,
                     src, src_ld, dst, dst_ld, rows, cols, accumulate);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

namespace ctr {
__global__ __launch_bounds__(256) void mul2d_kernel(const float* __restrict__ a, int64_t lda,
                                                   const float* __restrict__ b, int64_t ldb, float* __restrict__ out,
                                                   int64_t ldo, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = p / cols;
    const int c = (int)(p % cols);
    out[r * ldo + c] = a[r * lda + c] * b[r * ldb + c];
  }
}
}  // namespace ctr

extern "C" int ctr_mul2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int rows,
                         int cols, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_mul2d: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(a && b && out, "ctr_mul2d: null pointer");
  hipLaunchKernelGGL(mul2d_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                     ldb, out, ldo, rows, cols);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dropout_apply(float* x, int64_t ld, int rows, int cols, const ctr_dropout_t* drop,
                                 void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_dropout_apply: bad size");
  const Drop d = make_drop(drop);
  if ((int64_t)rows * cols == 0 || !d.on) return CTR_OK;
  CTR_REQUIRE(x, "ctr_dropout_apply: null pointer");
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0,
                     (hipStream_t)stream, x, ld, rows, cols, d);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dropout_apply2(float* x, int64_t ld, int rows, int cols_a, const ctr_dropout_t* drop_a, int cols_b,
                                  const ctr_dropout_t* drop_b, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols_a >= 0 && cols_b >= 0 && ld >= cols_a + cols_b, "ctr_dropout_apply2: bad size");
  const Drop da = make_drop(drop_a), db = make_drop(drop_b);
  if ((int64_t)rows * (cols_a + cols_b) == 0 || (!da.on && !db.on)) return CTR_OK;
  CTR_REQUIRE(x, "ctr_dropout_apply2: null pointer");
  if (cols_a + cols_b > 256) {   // wide blocks: one plain launch each
    int rc = CTR_OK;
    if (da.on) rc = ctr_dropout_apply(x, ld, rows, cols_a, drop_a, stream);
    if (rc == CTR_OK && db.on) rc = ctr_dropout_apply(x + cols_a, ld, rows, cols_b, drop_b, stream);
    return rc;
  }
  int lw = 0;
  while ((1 << lw) < cols_a + cols_b) ++lw;
  const int64_t steps = cdiv(rows, 256 >> lw);
  hipLaunchKernelGGL(dropout_apply2_kernel, dim3((unsigned)(steps < 4096 ? steps : 4096)), dim3(256), 0,
                     (hipStream_t)stream, x, ld, rows, cols_a, da, cols_b, db, lw);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_relu_mask_apply(float* g, int64_t g_ld, const float* act, int64_t act_ld, int rows, int cols,
                                   float scale, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_relu_mask_apply: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(g && act, "ctr_relu_mask_apply: null pointer");
  hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream,
                     g, g_ld, act, act_ld, rows, cols, scale);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_esmm_head(const float* ctr_logit, const float* cvr_logit, const float* y, int B,
                             float grad_scale, float* y_out, float* loss_out, float* dctr, float* dcvr,
                             void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B > 0 && ctr_logit && cvr_logit, "ctr_esmm_head: bad argument");
  CTR_REQUIRE(y || (!loss_out && !dctr && !dcvr), "ctr_esmm_head: labels are required for loss / gradient");
  const int nb = (int)cdiv(B, 256);
  float* part = nullptr;
  if (loss_out) {
    if (!workspace || workspace_bytes < ctr_loss_head_workspace_bytes(B)) {
      set_error("ctr_esmm_head: workspace %zu < %zu", workspace_bytes, ctr_loss_head_workspace_bytes(B));
      return CTR_EWORKSPACE;
    }
    part = (float*)workspace;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(esmm_head_kernel, dim3(nb), dim3(256), 0, st, ctr_logit, cvr_logit, y, B, grad_scale, y_out,
                     dctr, dcvr, part);
  if (loss_out) hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, part, nb, B, loss_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
