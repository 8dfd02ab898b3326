// DCN cross network, all layers fused, O(B*D) traffic.
//
// x_{l+1} = x0 * (x_l . w_l) + b_l + x_l keeps every x_l in span{x0, prefix sums of b}:
//   x_l = x0 * c_l + Bl_l,  Bl_l = sum_{l'<l} b_l',  c_0 = 1,
//   s_l = x_l . w_l = c_l * p_l + q_l,  p_l = x0 . w_l,  q_l = Bl_l . w_l,  c_{l+1} = c_l + s_l.
// So the forward is one pass over x0 (L simultaneous dot products, row held in registers),
// a scalar recursion, and one write of x_L; the backward is a second pass of the same shape
// plus fixed-order batch reductions for dw / db.  One wave per sample.
#include "common.h"

namespace ctr {

constexpr int LMAX = 8;

// Bl[l][j] for l = 0..L (prefix sums of b) and q[l] = Bl_l . w_l.  One block.
__global__ __launch_bounds__(256) void cross_prep_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                        int64_t wld, int D, int L, float* __restrict__ Bl,
                                                        float* __restrict__ q) {
  __shared__ float red[256];
  float qacc[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) qacc[l] = 0.f;
  for (int j = threadIdx.x; j < D; j += 256) {
    float run = 0.f;
    for (int l = 0; l <= L; ++l) {
      Bl[(int64_t)l * D + j] = run;
      if (l < L) {
        qacc[l] += run * w[(int64_t)l * wld + j];
        run += b[(int64_t)l * wld + j];
      }
    }
  }
  for (int l = 0; l < L; ++l) {
    red[threadIdx.x] = qacc[l];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) q[l] = red[0];
    __syncthreads();
  }
}

template <int NPL>
__global__ __launch_bounds__(256) void cross_fwd_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                       const float* __restrict__ w, int64_t wld,
                                                       const float* __restrict__ Bl,
                                                       const float* __restrict__ q, int B, int D, int L,
                                                       float* __restrict__ xL, int64_t xL_ld,
                                                       float* __restrict__ aux) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float x[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int j = lane + 64 * i;
    x[i] = j < D ? x0[b * x0_ld + j] : 0.f;
  }
  float c = 1.f;
  for (int l = 0; l < L; ++l) {
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int j = lane + 64 * i;
      if (j < D) p += x[i] * w[(int64_t)l * wld + j];
    }
    p = wave_sum(p);
    const float s = c * p + q[l];
    c += s;
    if (lane == 0 && aux) {
      aux[b * 2 * L + l] = p;
      aux[b * 2 * L + L + l] = c;  // c_{l+1}
    }
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int j = lane + 64 * i;
    if (j < D) xL[b * xL_ld + j] = x[i] * c + Bl[(int64_t)L * D + j];
  }
}

template <int NPL>
__global__ __launch_bounds__(256) void cross_bwd_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                       const float* __restrict__ w, int64_t wld,
                                                       const float* __restrict__ aux,
                                                       const float* __restrict__ g, int64_t g_ld, int B, int D,
                                                       int L, float* __restrict__ dx0, int64_t dx0_ld,
                                                       int accumulate, float* __restrict__ dpq) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float gv[NPL];
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int j = lane + 64 * i;
    gv[i] = j < D ? g[b * g_ld + j] : 0.f;
    if (j < D) a += gv[i] * x0[b * x0_ld + j];
  }
  a = wave_sum(a);
  // scalar backward of c_{l+1} = c_l * (1 + p_l) + q_l
  float dp[LMAX];
  float dc = a;
  const float cL = aux[b * 2 * L + L + (L - 1)];
#pragma unroll
  for (int l = LMAX - 1; l >= 0; --l) {
    dp[l] = 0.f;
    if (l < L) {
      const float c_l = l == 0 ? 1.f : aux[b * 2 * L + L + (l - 1)];
      const float p_l = aux[b * 2 * L + l];
      dp[l] = dc * c_l;
      if (lane == 0) {
        dpq[b * 2 * L + l] = dp[l];
        dpq[b * 2 * L + L + l] = dc;  // dq_l
      }
      dc = dc * (1.f + p_l);
    }
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int j = lane + 64 * i;
    if (j < D) {
      float v = gv[i] * cL;
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < L) v += dp[l] * w[(int64_t)l * wld + j];
      float* d = dx0 + b * dx0_ld + j;
      *d = accumulate ? *d + v : v;
    }
  }
}

// part[chunk][l][j] = sum_{b in chunk} dp_l[b] * x0[b,j]   (l < L);   part[chunk][L][j] = sum_b g[b,j]
// grid (ceil(D/64), nchunks), block (64, 4)
__global__ __launch_bounds__(256) void cross_bwd_partial_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                               const float* __restrict__ g, int64_t g_ld,
                                                               const float* __restrict__ dpq, int B, int D,
                                                               int L, int rows_per_chunk,
                                                               float* __restrict__ part) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < B ? r0 + rows_per_chunk : B;
  float acc[LMAX + 1];
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) acc[l] = 0.f;
  if (j < D) {
    for (int64_t r = r0 + ty; r < r1; r += 4) {
      const float xv = x0[r * x0_ld + j];
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < L) acc[l] += dpq[r * 2 * L + l] * xv;
      acc[LMAX] += g[r * g_ld + j];
    }
  }
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) {
    if (l < L || l == LMAX) {
      red[ty][tx] = acc[l];
      __syncthreads();
      if (ty == 0 && j < D) {
        const int slot = l == LMAX ? L : l;
        part[((int64_t)blockIdx.y * (L + 1) + slot) * D + j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
      }
      __syncthreads();
    }
  }
}

// sdq[l] = sum_b dq_l[b]; one block, fixed order.
__global__ __launch_bounds__(256) void cross_bwd_sdq_kernel(const float* __restrict__ dpq, int B, int L,
                                                           float* __restrict__ sdq) {
  __shared__ float red[256];
  for (int l = 0; l < L; ++l) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) acc += dpq[(int64_t)b * 2 * L + L + l];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) sdq[l] = red[0];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void cross_bwd_final_kernel(const float* __restrict__ part, int nchunks,
                                                             const float* __restrict__ w, int64_t wld,
                                                             const float* __restrict__ Bl,
                                                             const float* __restrict__ sdq, int D, int L,
                                                             float* __restrict__ dw, float* __restrict__ db) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D) return;
  float gsum = 0.f;
  for (int c = 0; c < nchunks; ++c) gsum += part[((int64_t)c * (L + 1) + L) * D + j];
  // dBl_L = colsum(g); dBl_l = sdq_l * w_l (l < L); db_l' = sum_{l > l'} dBl_l
  float run = gsum;
  for (int l = L - 1; l >= 0; --l) {
    db[(int64_t)l * wld + j] = run;
    run += sdq[l] * w[(int64_t)l * wld + j];
    float acc = 0.f;
    for (int c = 0; c < nchunks; ++c) acc += part[((int64_t)c * (L + 1) + l) * D + j];
    dw[(int64_t)l * wld + j] = acc + sdq[l] * Bl[(int64_t)l * D + j];
  }
}

static int cross_chunks(int B) {
  int64_t n = cdiv(B, 256);
  if (n < 1) n = 1;
  if (n > 512) n = 512;
  return (int)n;
}

static size_t cross_prep_floats(int D, int L) { return align_up((size_t)(L + 1) * D + LMAX, 64); }

}  // namespace ctr

using namespace ctr;

extern "C" size_t ctr_cross_fwd_workspace_bytes(int D, int L) { return cross_prep_floats(D, L) * sizeof(float); }

extern "C" size_t ctr_cross_bwd_workspace_bytes(int B, int D, int L) {
  return (cross_prep_floats(D, L) + align_up((size_t)B * 2 * L, 64) + 64 +
          (size_t)cross_chunks(B) * (L + 1) * D) * sizeof(float);
}

#define CROSS_DISPATCH(KERNEL, ...)                                                              \
  do {                                                                                           \
    const int npl = (int)cdiv(D, 64);                                                            \
    const dim3 grid((unsigned)cdiv(B, 4)), block(256);                                           \
    if (npl <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, st, __VA_ARGS__);                \
    else if (npl <= 8) hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, st, __VA_ARGS__);           \
    else if (npl <= 16) hipLaunchKernelGGL(KERNEL<16>, grid, block, 0, st, __VA_ARGS__);         \
    else if (npl <= 32) hipLaunchKernelGGL(KERNEL<32>, grid, block, 0, st, __VA_ARGS__);         \
    else if (npl <= 64) hipLaunchKernelGGL(KERNEL<64>, grid, block, 0, st, __VA_ARGS__);         \
    else hipLaunchKernelGGL(KERNEL<128>, grid, block, 0, st, __VA_ARGS__);                       \
  } while (0)

extern "C" int ctr_cross_fwd(const float* x0, int64_t x0_ld, const float* w, const float* b, int64_t wld,
                             int B, int D, int L, float* xL, int64_t xL_ld, float* aux_out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B >= 0 && D > 0 && L >= 1 && L <= LMAX, "ctr_cross_fwd: need 1 <= L <= %d", LMAX);
  CTR_REQUIRE(D <= 8192, "ctr_cross_fwd: D=%d exceeds the register-resident limit 8192", D);
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(x0 && w && b && xL && workspace, "ctr_cross_fwd: null pointer");
  CTR_REQUIRE(wld >= D, "ctr_cross_fwd: wld < D");
  if (workspace_bytes < ctr_cross_fwd_workspace_bytes(D, L)) {
    set_error("ctr_cross_fwd: workspace %zu < %zu", workspace_bytes, ctr_cross_fwd_workspace_bytes(D, L));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* Bl = (float*)workspace;
  float* q = Bl + (size_t)(L + 1) * D;
  hipLaunchKernelGGL(cross_prep_kernel, dim3(1), dim3(256), 0, st, w, b, wld, D, L, Bl, q);
  CROSS_DISPATCH(cross_fwd_kernel, x0, x0_ld, w, wld, Bl, q, B, D, L, xL, xL_ld, aux_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_cross_bwd(const float* x0, int64_t x0_ld, const float* w, const float* b, int64_t wld,
                             const float* aux, const float* g, int64_t g_ld, int B, int D, int L,
                             float* dx0, int64_t dx0_ld, int accumulate, float* dw, float* db,
                             void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B > 0 && D > 0 && L >= 1 && L <= LMAX, "ctr_cross_bwd: need B > 0 and 1 <= L <= %d", LMAX);
  CTR_REQUIRE(D <= 8192, "ctr_cross_bwd: D=%d exceeds the register-resident limit 8192", D);
  CTR_REQUIRE(x0 && w && b && aux && g && dx0 && dw && db && workspace, "ctr_cross_bwd: null pointer");
  CTR_REQUIRE(wld >= D, "ctr_cross_bwd: wld < D");
  if (workspace_bytes < ctr_cross_bwd_workspace_bytes(B, D, L)) {
    set_error("ctr_cross_bwd: workspace %zu < %zu", workspace_bytes, ctr_cross_bwd_workspace_bytes(B, D, L));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* Bl = (float*)workspace;
  float* q = Bl + (size_t)(L + 1) * D;
  float* dpq = Bl + cross_prep_floats(D, L);
  float* sdq = dpq + align_up((size_t)B * 2 * L, 64);
  float* part = sdq + 64;
  const int nchunks = cross_chunks(B);
  const int rows_per_chunk = (int)cdiv(B, nchunks);
  hipLaunchKernelGGL(cross_prep_kernel, dim3(1), dim3(256), 0, st, w, b, wld, D, L, Bl, q);
  CROSS_DISPATCH(cross_bwd_kernel, x0, x0_ld, w, wld, aux, g, g_ld, B, D, L, dx0, dx0_ld, accumulate, dpq);
  hipLaunchKernelGGL(cross_bwd_partial_kernel, dim3((unsigned)cdiv(D, 64), nchunks), dim3(256), 0, st, x0,
                     x0_ld, g, g_ld, dpq, B, D, L, rows_per_chunk, part);
  hipLaunchKernelGGL(cross_bwd_sdq_kernel, dim3(1), dim3(256), 0, st, dpq, B, L, sdq);
  hipLaunchKernelGGL(cross_bwd_final_kernel, dim3((unsigned)cdiv(D, 256)), dim3(256), 0, st, part, nchunks, w,
                     wld, Bl, sdq, D, L, dw, db);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
