// DCN cross network, all layers fused, O(B*D) traffic.
//
// x_{l+1} = x0 * (x_l . w_l) + b_l + x_l keeps every x_l in span{x0, prefix sums of b}:
//   x_l = x0 * c_l + Bl_l,  Bl_l = sum_{l'<l} b_l',  c_0 = 1,
//   s_l = x_l . w_l = c_l * p_l + q_l,  p_l = x0 . w_l,  q_l = Bl_l . w_l,  c_{l+1} = c_l + s_l.
// So the forward is one pass over x0 (L simultaneous dot products, row held in registers),
// a scalar recursion, and one write of x_L; the backward is a second pass of the same shape
// plus fixed-order batch reductions for dw / db.  One wave per sample.
#include "common.h"
#include <algorithm>

namespace ctr {

constexpr int LMAX = 8;
static inline int cross_dp_of(int D) { return (D + 3) / 4 * 4; }
constexpr int PREP_THREADS = 1024;

// Four consecutive floats row[j .. j+3] of a row of D; vec: the row may be read 16 bytes at a time.
__device__ static inline float4 row4(const float* __restrict__ row, int j, int D, bool vec) {
  if (vec && j + 3 < D) return *reinterpret_cast<const float4*>(row + j);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < D) v.x = row[j];
  if (j + 1 < D) v.y = row[j + 1];
  if (j + 2 < D) v.z = row[j + 2];
  if (j + 3 < D) v.w = row[j + 3];
  return v;
}
__device__ static inline void put4(float* __restrict__ row, int j, int D, bool vec, float4 v) {
  if (vec && j + 3 < D) {
    *reinterpret_cast<float4*>(row + j) = v;
    return;
  }
  if (j < D) row[j] = v.x;
  if (j + 1 < D) row[j + 1] = v.y;
  if (j + 2 < D) row[j + 2] = v.z;
  if (j + 3 < D) row[j + 3] = v.w;
}
__device__ static inline bool rows16(const void* p, int64_t ld) { return ((uintptr_t)p % 16) == 0 && ld % 4 == 0; }
__device__ static inline float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// Bl[l][j] for l = 0..L (prefix sums of b, rows DP = D rounded up to 4 apart) and q[l] = Bl_l . w_l.  One block.
__global__ __launch_bounds__(PREP_THREADS) void cross_prep_kernel(const float* __restrict__ w,
                                                                 const float* __restrict__ b, int64_t wld, int D,
                                                                 int DP, int L, float* __restrict__ Bl,
                                                                 float* __restrict__ q) {
  __shared__ float red[LMAX][PREP_THREADS / 64];
  float qacc[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) qacc[l] = 0.f;
  for (int j = threadIdx.x; j < D; j += PREP_THREADS) {
    float run = 0.f;
#pragma unroll
    for (int l = 0; l <= LMAX; ++l) {
      if (l <= L) Bl[(int64_t)l * DP + j] = run;
      if (l < L) {
        qacc[l] += run * w[(int64_t)l * wld + j];
        run += b[(int64_t)l * wld + j];
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    const float v = wave_sum(qacc[l]);
    if (lane == 0) red[l][wave] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < L) {
    float v = 0.f;
    for (int k = 0; k < PREP_THREADS / 64; ++k) v += red[threadIdx.x][k];
    q[threadIdx.x] = v;
  }
}

// Persistent form (the one used whenever the L weight rows and the bias prefix row fit the LDS): a block keeps
// w_0 .. w_{L-1} and Bl_L in LDS for its whole life and its four waves walk the samples, the next sample's row
// already in flight (registers) while the current one is reduced and written.  A sample costs its own bytes from
// memory and nothing else; the weight rows, re-read per sample, come out of the LDS.
// lds: [L][DP] weight rows (zero-padded to DP), then [DP] Bl_L.  (The staging loop is written out in both kernels:
// as a shared helper it crashed this compiler's inliner.)
template <int NQ>
__global__ __launch_bounds__(256) void cross_fwd_lds_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                           const float* __restrict__ w, int64_t wld,
                                                           const float* __restrict__ Bl, int DP,
                                                           const float* __restrict__ q, int B, int D, int L,
                                                           float* __restrict__ xL, int64_t xL_ld,
                                                           float* __restrict__ aux) {
  extern __shared__ __attribute__((aligned(16))) float cross_lds[];
  {  // stage w_0 .. w_{L-1} and Bl_L: 16 bytes per load, eight loads of a thread in flight before the first store
    const int q4 = DP / 4, n4 = (L + 1) * q4;
    const bool vw = rows16(w, wld);
    for (int t0 = threadIdx.x; t0 < n4; t0 += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256;
        const int l = t / q4, j = 4 * (t % q4);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < n4) v[u] = l < L ? row4(w + (int64_t)l * wld, j, D, vw) : row4(Bl + (int64_t)L * DP, j, D, true);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256;
        if (t < n4) *reinterpret_cast<float4*>(cross_lds + 4 * t) = v[u];
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const bool vx = rows16(x0, x0_ld), vo = rows16(xL, xL_ld);
  float ql[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) ql[l] = l < L ? q[l] : 0.f;
  float4 x[NQ], xn[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) xn[i] = row4(x0 + b * x0_ld, 256 * i + 4 * lane, D, vx);
  for (; b < B; b += stride) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) x[i] = xn[i];
    const int64_t bn = b + stride < B ? b + stride : b;            // the last sample re-reads itself (harmless)
#pragma unroll
    for (int i = 0; i < NQ; ++i) xn[i] = row4(x0 + bn * x0_ld, 256 * i + 4 * lane, D, vx);
    float p[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      p[l] = 0.f;
      if (l < L) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)      // lanes past the row hold zeros and must not touch the LDS behind it
          if (256 * i + 4 * lane < DP)
            p[l] += dot4(x[i], *reinterpret_cast<const float4*>(cross_lds + l * DP + 256 * i + 4 * lane));
      }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < L) p[l] = wave_sum(p[l]);
    float c = 1.f;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
      if (l < L) {
        const float s = c * p[l] + ql[l];
        c += s;
        if (lane == 0 && aux) {
          aux[b * 2 * L + l] = p[l];
          aux[b * 2 * L + L + l] = c;  // c_{l+1}
        }
      }
    }
    float* orow = xL + b * xL_ld;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int j = 256 * i + 4 * lane;
      if (j < D) {
        const float4 bl = *reinterpret_cast<const float4*>(cross_lds + L * DP + j);
        put4(orow, j, D, vo, make_float4(x[i].x * c + bl.x, x[i].y * c + bl.y, x[i].z * c + bl.z, x[i].w * c + bl.w));
      }
    }
  }
}

// ---- Aligned forms: rows 16-byte aligned and at least DP floats apart, buffers below 2 GiB ------------------------------
// The kernels above spend their time issuing instructions, not waiting for memory (about 1500 per sample where the
// arithmetic needs 300: every conditional load or store is a branch, the predicates of a row's tail fill the scalar
// registers until they spill, and six 64-lane reductions per sample cost six steps each).  The forms below address
// memory through buffer descriptors, whose range check stands in for the branches: a lane that has nothing to load
// or store passes an offset beyond the buffer and the hardware returns zeros or drops the store.
__device__ static inline float4 quad_mask(float4 v, int j, int D) {
  v.x = j < D ? v.x : 0.f;
  v.y = j + 1 < D ? v.y : 0.f;
  v.z = j + 2 < D ? v.z : 0.f;
  v.w = j + 3 < D ? v.w : 0.f;
  return v;
}
// The quad at columns j .. j+3 of a row of D, at byte offset `off` of the buffer: whole where it fits, float by float
// where the row ends inside it, not at all past the row.
__device__ static inline void quad_st_tail(__amdgpu_buffer_rsrc_t r, unsigned off, int j, int D, float4 v) {
  const bool full = j + 3 < D;
  buf_st4(r, full ? off : OOB, v);
  buf_st1(r, !full && j < D ? off : OOB, v.x);
  buf_st1(r, !full && j + 1 < D ? off + 4 : OOB, v.y);
  buf_st1(r, !full && j + 2 < D ? off + 8 : OOB, v.z);
}
__device__ static inline float row16_sum(float v) {          // total over the 16 lanes of a DPP row, in every lane
  v = dpp_add<DPP_XOR1>(v);
  v = dpp_add<DPP_XOR2>(v);
  v = dpp_add<DPP_ROR + 4>(v);
  return dpp_add<DPP_ROR + 8>(v);
}

// Stages `nrows` rows of DPL floats into the LDS: row l < L is w_l, row L (if staged) is BlL; zeros from column D on.
// Four-byte loads, a wave's lanes on consecutive floats, eight in flight, no branches.
template <int DPL, int NT>
__device__ static inline void stage_rows(float* __restrict__ lds, const float* __restrict__ w, int64_t wld,
                                         const float* __restrict__ BlL, int L, int nrows, int D) {
  const int n = nrows * DPL;
  for (int t0 = threadIdx.x; t0 < n; t0 += 8 * NT) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * NT < n ? t0 + u * NT : 0;
      const int l = t / DPL, j = t - l * DPL;
      const int jc = j < D ? j : 0;
      const float* src = l < L ? w + (int64_t)l * wld + jc : BlL + jc;
      v[u] = *src;
      v[u] = j < D ? v[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (t0 + u * NT < n) lds[t0 + u * NT] = v[u];
  }
}

// Forward, a sample per 16-lane row (four samples per wave, NQ quads of the row per lane, lane k of a row on quads
// k, k + 16, ...): a layer's dot product is reduced over 16 lanes in four DPP steps, the four samples of a wave
// share every LDS read of the weights, and nothing in the loop branches.  The last four quads of a lane are the only
// ones that can lie across or past the end of the row (NQ is the quad count rounded up to a multiple of four).
// One block of eight waves per CU stages the weights once; its first rows are requested before the staging, which
// then runs under their latency.
// LDS: [L + 1][NQ * 64] rows (w_0 .. w_{L-1}, Bl_L; zeros from column D on), then q[L].
constexpr int R16_NT = 512;
template <int NQ>
__global__ __launch_bounds__(R16_NT) void cross_fwd_r16_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                              const float* __restrict__ w, int64_t wld,
                                                              const float* __restrict__ BlL,
                                                              const float* __restrict__ q, int B, int D, int L,
                                                              float* __restrict__ xL, int64_t xL_ld,
                                                              float* __restrict__ aux) {
  constexpr int DPL = NQ * 64;
  extern __shared__ __attribute__((aligned(16))) float cross_lds[];
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = buf_of(x0, (unsigned)(((int64_t)(B - 1) * x0_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t ro = buf_of(xL, (unsigned)(((int64_t)(B - 1) * xL_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t ra = buf_of(aux, (unsigned)((int64_t)B * 2 * L * 4));
  const float* ql = cross_lds + (L + 1) * DPL;
  const int64_t stride = (int64_t)gridDim.x * (R16_NT / 16);
  int64_t b0 = ((int64_t)blockIdx.x * (R16_NT / 64) + (threadIdx.x >> 6)) * 4;
  float4 x[NQ];
  auto load_rows = [&](int64_t b) {
    const unsigned xoff = b < B ? (unsigned)(b * x0_ld * 4) + 16 * l16 : OOB;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      if (i < NQ - 4) x[i] = buf_ld4(rx, xoff + 256 * i);
      else x[i] = buf_ld4(rx, 64 * i + 4 * l16 < D ? xoff + 256 * i : OOB);
    }
  };
  load_rows(b0 + grp);
  stage_rows<DPL, R16_NT>(cross_lds, w, wld, BlL, L, L + 1, D);
  if ((int)threadIdx.x < L) cross_lds[(L + 1) * DPL + threadIdx.x] = q[threadIdx.x];
  __syncthreads();
  while (b0 < B) {
    const int64_t b = b0 + grp;
#pragma unroll
    for (int i = NQ - 4; i < NQ; ++i) x[i] = quad_mask(x[i], 64 * i + 4 * l16, D);
    float c = 1.f, av = 0.f;               // lane l < L of a row keeps p_l, lane L + l keeps c_{l+1}
    const float* wl = cross_lds + 4 * l16;
    for (int l = 0; l < L; ++l, wl += DPL) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + 64 * i);
        acc.x += x[i].x * wv.x; acc.y += x[i].y * wv.y; acc.z += x[i].z * wv.z; acc.w += x[i].w * wv.w;
      }
      const float p = row16_sum((acc.x + acc.y) + (acc.z + acc.w));
      c += c * p + ql[l];
      av = l16 == l ? p : av;
      av = l16 == L + l ? c : av;
    }
    buf_st1(ra, b < B && l16 < 2 * L ? (unsigned)(b * 2 * L + l16) * 4 : OOB, av);
    const unsigned ooff = b < B ? (unsigned)(b * xL_ld * 4) + 16 * l16 : OOB;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const float4 bl = *reinterpret_cast<const float4*>(wl + 64 * i);          // wl is at row L here: Bl_L
      const float4 o = make_float4(x[i].x * c + bl.x, x[i].y * c + bl.y, x[i].z * c + bl.z, x[i].w * c + bl.w);
      if (i < NQ - 4) buf_st4(ro, ooff + 256 * i, o);
      else quad_st_tail(ro, ooff + 256 * i, 64 * i + 4 * l16, D, o);
    }
    b0 += stride;
    if (b0 < B) load_rows(b0 + grp);
  }
}

// Backward of the same shape: a = g . x0 per sample (one 16-lane reduction), the scalar recursion backwards from
// the forward's p_l and c_l (two floats per layer and sample, fetched by the row's first 2L lanes and passed
// around the row), dx0 = g c_L + sum_l dp_l w_l (+ what dx0 held).  The gradient row is scaled in place and the
// registers of the x0 row take the old dx0 once the dot product is done.
// LDS: [L][NQ * 64] weight rows.
template <int NQ>
__global__ __launch_bounds__(R16_NT) void cross_bwd_r16_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                              const float* __restrict__ w, int64_t wld,
                                                              const float* __restrict__ aux,
                                                              const float* __restrict__ g, int64_t g_ld, int B, int D,
                                                              int L, float* __restrict__ dx0, int64_t dx0_ld,
                                                              int accumulate, float* __restrict__ dpq) {
  constexpr int DPL = NQ * 64;
  extern __shared__ __attribute__((aligned(16))) float cross_lds[];
  const int lane = threadIdx.x & 63, l16 = lane & 15, grp = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = buf_of(x0, (unsigned)(((int64_t)(B - 1) * x0_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t rg = buf_of(g, (unsigned)(((int64_t)(B - 1) * g_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t rd = buf_of(dx0, (unsigned)(((int64_t)(B - 1) * dx0_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t ra = buf_of(aux, (unsigned)((int64_t)B * 2 * L * 4));
  const __amdgpu_buffer_rsrc_t rq = buf_of(dpq, (unsigned)((int64_t)B * 2 * L * 4));
  const int64_t stride = (int64_t)gridDim.x * (R16_NT / 16);
  int64_t b0 = ((int64_t)blockIdx.x * (R16_NT / 64) + (threadIdx.x >> 6)) * 4;
  float4 x[NQ], gv[NQ];
  float A = 0.f;
  auto load_rows = [&](int64_t b) {
    const unsigned xoff = b < B ? (unsigned)(b * x0_ld * 4) + 16 * l16 : OOB;
    const unsigned goff = b < B ? (unsigned)(b * g_ld * 4) + 16 * l16 : OOB;
    A = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                      ra, b < B && l16 < 2 * L ? (unsigned)(b * 2 * L + l16) * 4 : OOB, 0, 0));
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const bool in = i < NQ - 4 || 64 * i + 4 * l16 < D;
      x[i] = buf_ld4(rx, in ? xoff + 256 * i : OOB);
      gv[i] = buf_ld4(rg, in ? goff + 256 * i : OOB);
    }
  };
  load_rows(b0 + grp);
  stage_rows<DPL, R16_NT>(cross_lds, w, wld, w, L, L, D);
  __syncthreads();
  while (b0 < B) {
    const int64_t b = b0 + grp;
#pragma unroll
    for (int i = NQ - 4; i < NQ; ++i) {
      x[i] = quad_mask(x[i], 64 * i + 4 * l16, D);
      gv[i] = quad_mask(gv[i], 64 * i + 4 * l16, D);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      acc.x += gv[i].x * x[i].x; acc.y += gv[i].y * x[i].y; acc.z += gv[i].z * x[i].z; acc.w += gv[i].w * x[i].w;
    }
    const unsigned doff = b < B ? (unsigned)(b * dx0_ld * 4) + 16 * l16 : OOB;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {           // what dx0 holds (zeros if it is not to be kept), into the x0 registers
      const bool in = accumulate && (i < NQ - 4 || 64 * i + 4 * l16 < D);
      x[i] = buf_ld4(rd, in ? doff + 256 * i : OOB);
    }
    float dc = row16_sum((acc.x + acc.y) + (acc.z + acc.w));
    const float cL = __shfl(A, 2 * L - 1, 16);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      gv[i].x *= cL; gv[i].y *= cL; gv[i].z *= cL; gv[i].w *= cL;
    }
    float keep = 0.f;                      // lane l < L of a row keeps dp_l, lane L + l keeps dq_l
    const float* wl = cross_lds + (L - 1) * DPL + 4 * l16;
    for (int l = L - 1; l >= 0; --l, wl -= DPL) {          // backward of c_{l+1} = c_l (1 + p_l) + q_l
      const float p_l = __shfl(A, l, 16);
      const float c_l = l == 0 ? 1.f : __shfl(A, L + l - 1, 16);
      const float dp = dc * c_l;
      keep = l16 == l ? dp : keep;
      keep = l16 == L + l ? dc : keep;
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(wl + 64 * i);
        gv[i].x += dp * wv.x; gv[i].y += dp * wv.y; gv[i].z += dp * wv.z; gv[i].w += dp * wv.w;
      }
      dc *= 1.f + p_l;
    }
    buf_st1(rq, b < B && l16 < 2 * L ? (unsigned)(b * 2 * L + l16) * 4 : OOB, keep);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const float4 o = make_float4(gv[i].x + x[i].x, gv[i].y + x[i].y, gv[i].z + x[i].z, gv[i].w + x[i].w);
      if (i < NQ - 4) buf_st4(rd, doff + 256 * i, o);
      else quad_st_tail(rd, doff + 256 * i, 64 * i + 4 * l16, D, o);
    }
    b0 += stride;
    if (b0 < B) load_rows(b0 + grp);
  }
}

// One wave per sample (rows too long for the LDS-resident form); lane holds columns 256*i + 4*lane .. +3 for
// i < NQ.  The L dot products x0 . w_l do not depend on the recursion, so they are taken in one sweep over the row
// in registers and reduced together.
template <int NQ>
__global__ __launch_bounds__(256) void cross_fwd_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                       const float* __restrict__ w, int64_t wld,
                                                       const float* __restrict__ Bl, int DP,
                                                       const float* __restrict__ q, int B, int D, int L,
                                                       float* __restrict__ xL, int64_t xL_ld,
                                                       float* __restrict__ aux) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const bool vx = rows16(x0, x0_ld), vw = rows16(w, wld), vo = rows16(xL, xL_ld);
  const float* xr = x0 + b * x0_ld;
  float4 x[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) x[i] = row4(xr, 256 * i + 4 * lane, D, vx);
  float p[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    p[l] = 0.f;
    if (l < L) {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        if (256 * i < D) p[l] += dot4(x[i], row4(w + (int64_t)l * wld, 256 * i + 4 * lane, D, vw));
    }
  }
#pragma unroll
  for (int l = 0; l < LMAX; ++l)
    if (l < L) p[l] = wave_sum(p[l]);
  float c = 1.f;
#pragma unroll
  for (int l = 0; l < LMAX; ++l) {
    if (l < L) {
      const float s = c * p[l] + q[l];
      c += s;
      if (lane == 0 && aux) {
        aux[b * 2 * L + l] = p[l];
        aux[b * 2 * L + L + l] = c;  // c_{l+1}
      }
    }
  }
  float* orow = xL + b * xL_ld;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int j = 256 * i + 4 * lane;
    if (j < D) {
      const float4 bl = row4(Bl + (int64_t)L * DP, j, D, true);
      put4(orow, j, D, vo, make_float4(x[i].x * c + bl.x, x[i].y * c + bl.y, x[i].z * c + bl.z, x[i].w * c + bl.w));
    }
  }
}

// Backward of the same shape, persistent, weights in LDS: per sample g and x0 (prefetched one sample ahead), the
// scalar recursion backwards, dx0 = g c_L + sum_l dp_l w_l.
template <int NQ>
__global__ __launch_bounds__(256) void cross_bwd_lds_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                           const float* __restrict__ w, int64_t wld, int DP,
                                                           const float* __restrict__ aux,
                                                           const float* __restrict__ g, int64_t g_ld, int B, int D,
                                                           int L, float* __restrict__ dx0, int64_t dx0_ld,
                                                           int accumulate, float* __restrict__ dpq) {
  extern __shared__ __attribute__((aligned(16))) float cross_lds[];
  {  // stage w_0 .. w_{L-1}: 16 bytes per load, eight loads of a thread in flight before the first store
    const int q4 = DP / 4, n4 = L * q4;
    const bool vw = rows16(w, wld);
    for (int t0 = threadIdx.x; t0 < n4; t0 += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < n4) v[u] = row4(w + (int64_t)(t / q4) * wld, 4 * (t % q4), D, vw);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256;
        if (t < n4) *reinterpret_cast<float4*>(cross_lds + 4 * t) = v[u];
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const bool vx = rows16(x0, x0_ld), vg = rows16(g, g_ld), vd = rows16(dx0, dx0_ld);
  float4 gv[NQ], xv[NQ], gn[NQ], xn[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    gn[i] = row4(g + b * g_ld, 256 * i + 4 * lane, D, vg);
    xn[i] = row4(x0 + b * x0_ld, 256 * i + 4 * lane, D, vx);
  }
  for (; b < B; b += stride) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) { gv[i] = gn[i]; xv[i] = xn[i]; }
    const int64_t bn = b + stride < B ? b + stride : b;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      gn[i] = row4(g + bn * g_ld, 256 * i + 4 * lane, D, vg);
      xn[i] = row4(x0 + bn * x0_ld, 256 * i + 4 * lane, D, vx);
    }
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
      if (256 * i < D) a += dot4(gv[i], xv[i]);
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
    float* drow = dx0 + b * dx0_ld;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int j = 256 * i + 4 * lane;
      if (j < D) {
        float4 v = make_float4(gv[i].x * cL, gv[i].y * cL, gv[i].z * cL, gv[i].w * cL);
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
          if (l < L) {
            const float4 wv = *reinterpret_cast<const float4*>(cross_lds + l * DP + j);
            v.x += dp[l] * wv.x; v.y += dp[l] * wv.y; v.z += dp[l] * wv.z; v.w += dp[l] * wv.w;
          }
        if (accumulate) {
          const float4 o = row4(drow, j, D, vd);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        put4(drow, j, D, vd, v);
      }
    }
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void cross_bwd_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                       const float* __restrict__ w, int64_t wld,
                                                       const float* __restrict__ aux,
                                                       const float* __restrict__ g, int64_t g_ld, int B, int D,
                                                       int L, float* __restrict__ dx0, int64_t dx0_ld,
                                                       int accumulate, float* __restrict__ dpq) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const bool vx = rows16(x0, x0_ld), vw = rows16(w, wld), vg = rows16(g, g_ld), vd = rows16(dx0, dx0_ld);
  float4 gv[NQ];
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int j = 256 * i + 4 * lane;
    gv[i] = row4(g + b * g_ld, j, D, vg);
    if (256 * i < D) a += dot4(gv[i], row4(x0 + b * x0_ld, j, D, vx));
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
  float* drow = dx0 + b * dx0_ld;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int j = 256 * i + 4 * lane;
    if (j < D) {
      float4 v = make_float4(gv[i].x * cL, gv[i].y * cL, gv[i].z * cL, gv[i].w * cL);
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < L) {
          const float4 wv = row4(w + (int64_t)l * wld, j, D, vw);
          v.x += dp[l] * wv.x; v.y += dp[l] * wv.y; v.z += dp[l] * wv.z; v.w += dp[l] * wv.w;
        }
      if (accumulate) {
        const float4 o = row4(drow, j, D, vd);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      put4(drow, j, D, vd, v);
    }
  }
}

// part[chunk][l][j] = sum_{b in chunk} dp_l[b] * x0[b,j]   (l < L);   part[chunk][L][j] = sum_b g[b,j];
// psdq[chunk][l] = sum_{b in chunk} dq_l[b] (written by the blocks of the first column group).
// grid (ceil(D/256), nchunks), block 256 = 4 waves over the chunk's rows; a lane owns four consecutive columns
__global__ __launch_bounds__(256) void cross_bwd_partial_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                               const float* __restrict__ g, int64_t g_ld,
                                                               const float* __restrict__ dpq, int B, int D, int DP,
                                                               int L, int rows_per_chunk,
                                                               float* __restrict__ part, float* __restrict__ psdq) {
  __shared__ float4 red[4][64];
  __shared__ float sred[4][LMAX];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 256 + 4 * tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < B ? r0 + rows_per_chunk : B;
  const bool vx = rows16(x0, x0_ld), vg = rows16(g, g_ld);
  float4 acc[LMAX + 1];
  float sdq[LMAX];
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) acc[l] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int l = 0; l < LMAX; ++l) sdq[l] = 0.f;
  // The per-row scalars (dp_l, dq_l) of a stretch of rows are staged in LDS first (every lane of a wave needs the
  // same 2L words of a row: read from global memory they were 2L more load instructions per row than the two
  // loads that carry the row itself).
  constexpr int RF = 8;                            // rows in flight per wave (rows r, r + 4, ..., r + 28)
  constexpr int STRETCH = 128;
  __shared__ float sdp[STRETCH * 2 * LMAX];
  for (int64_t base = r0; base < r1; base += STRETCH) {
    const int64_t top = base + STRETCH < r1 ? base + STRETCH : r1;
    __syncthreads();
    for (int t = threadIdx.x; t < (int)(top - base) * 2 * L; t += 256) sdp[t] = dpq[base * 2 * L + t];
    __syncthreads();
    for (int64_t r = base + ty; r < top; r += 4 * RF) {
      float4 xs[RF], gs[RF];
      float ws[RF];
      int rl[RF];
#pragma unroll
      for (int u = 0; u < RF; ++u) {
        const bool ok = r + 4 * u < top;
        const int64_t rr = ok ? r + 4 * u : r;
        rl[u] = (int)(rr - base);
        ws[u] = ok ? 1.f : 0.f;
        xs[u] = row4(x0 + rr * x0_ld, j, D, vx);
        gs[u] = row4(g + rr * g_ld, j, D, vg);
      }
#pragma unroll
      for (int u = 0; u < RF; ++u) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
          if (l < L) {
            const float d = sdp[rl[u] * 2 * L + l] * ws[u];
            acc[l].x += d * xs[u].x; acc[l].y += d * xs[u].y; acc[l].z += d * xs[u].z; acc[l].w += d * xs[u].w;
            sdq[l] += sdp[rl[u] * 2 * L + L + l] * ws[u];
          }
        acc[LMAX].x += gs[u].x * ws[u]; acc[LMAX].y += gs[u].y * ws[u];
        acc[LMAX].z += gs[u].z * ws[u]; acc[LMAX].w += gs[u].w * ws[u];
      }
    }
  }
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) {
    if (l < L || l == LMAX) {
      red[ty][tx] = acc[l];
      __syncthreads();
      if (ty == 0 && j < D) {
        const int slot = l == LMAX ? L : l;
        const float4 a0 = red[0][tx], a1 = red[1][tx], a2 = red[2][tx], a3 = red[3][tx];
        put4(part + ((int64_t)blockIdx.y * (L + 1) + slot) * DP, j, D, true,
             make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                         (a0.w + a1.w) + (a2.w + a3.w)));
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
    if (tx == 0)
#pragma unroll
      for (int l = 0; l < LMAX; ++l) sred[ty][l] = sdq[l];
    __syncthreads();
    if ((int)threadIdx.x < L)
      psdq[(int64_t)blockIdx.y * LMAX + threadIdx.x] =
          (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
  }
}

// Sums the chunk partials in a fixed order and finishes dw / db.  Block = 64 columns x 16 chunk phases.
constexpr int CF_PH = 16;
__global__ __launch_bounds__(64 * CF_PH) void cross_bwd_final_kernel(const float* __restrict__ part,
                                                                    const float* __restrict__ psdq, int nchunks,
                                                                    const float* __restrict__ w, int64_t wld,
                                                                    const float* __restrict__ Bl, int D, int DP, int L,
                                                                    float* __restrict__ dw, float* __restrict__ db) {
  __shared__ float red[CF_PH][LMAX + 1][64];
  __shared__ float sdq[LMAX];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  if (ty < LMAX) {                                 // wave l totals the dq sums of layer l over the chunks
    float v = 0.f;
    if (ty < L)
      for (int c = tx; c < nchunks; c += 64) v += psdq[(int64_t)c * LMAX + ty];
    v = wave_sum(v);
    if (tx == 0) sdq[ty] = v;
  }
  float acc[LMAX + 1];
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) acc[l] = 0.f;
  if (j < D)
    for (int c = ty; c < nchunks; c += CF_PH) {
#pragma unroll
      for (int l = 0; l <= LMAX; ++l)
        if (l <= L) acc[l] += part[((int64_t)c * (L + 1) + l) * DP + j];
    }
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) red[ty][l][tx] = acc[l];
  __syncthreads();
  if (ty != 0 || j >= D) return;
  float tot[LMAX + 1];
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < CF_PH; ++q) v += red[q][l][tx];
    tot[l] = v;
  }
  // dBl_L = colsum(g); dBl_l = sdq_l * w_l (l < L); db_l' = sum_{l > l'} dBl_l
  float run = 0.f;
#pragma unroll
  for (int l = 0; l <= LMAX; ++l)
    if (l == L) run = tot[l];
#pragma unroll
  for (int l = LMAX - 1; l >= 0; --l)
    if (l < L) {
      db[(int64_t)l * wld + j] = run;
      run += sdq[l] * w[(int64_t)l * wld + j];
      dw[(int64_t)l * wld + j] = tot[l] + sdq[l] * Bl[(int64_t)l * DP + j];
    }
}

// cross_bwd_partial_kernel for aligned rows: the rows come through buffer descriptors (a row past the chunk reads as
// zeros, so the loop carries no row predicate and its sixteen loads go out together), 64 rows per chunk, and only
// the first column group adds up the dq sums.  grid (ceil(D/256), nchunks), block 256 = 4 waves over the rows.
constexpr int PF_ROWS = 64;
__global__ __launch_bounds__(256) void cross_bwd_partial_fast_kernel(const float* __restrict__ x0, int64_t x0_ld,
                                                                    const float* __restrict__ g, int64_t g_ld,
                                                                    const float* __restrict__ dpq, int B, int D, int DP,
                                                                    int L, float* __restrict__ part,
                                                                    float* __restrict__ psdq) {
  __shared__ float4 red[4][64];
  __shared__ float sdp[PF_ROWS * 2 * LMAX];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 256 + 4 * tx;
  const int r0 = blockIdx.y * PF_ROWS;
  const int nr = B - r0 < PF_ROWS ? B - r0 : PF_ROWS;
  const __amdgpu_buffer_rsrc_t rx = buf_of(x0, (unsigned)(((int64_t)(B - 1) * x0_ld + D) * 4));
  const __amdgpu_buffer_rsrc_t rg = buf_of(g, (unsigned)(((int64_t)(B - 1) * g_ld + D) * 4));
  constexpr int RF = 8;                            // rows in flight per wave (rows ty, ty + 4, ..., ty + 28)
  float4 xs[2][RF], gs[2][RF];
  auto load_round = [&](int k) {
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const int r = ty + 4 * (RF * k + u);
      const bool in = r < nr && j < D;
      xs[k][u] = buf_ld4(rx, in ? (unsigned)((int64_t)(r0 + r) * x0_ld + j) * 4 : OOB);
      gs[k][u] = buf_ld4(rg, in ? (unsigned)((int64_t)(r0 + r) * g_ld + j) * 4 : OOB);
    }
  };
  load_round(0);
  load_round(1);
  for (int t = threadIdx.x; t < PF_ROWS * 2 * L; t += 256) sdp[t] = t < nr * 2 * L ? dpq[(int64_t)r0 * 2 * L + t] : 0.f;
  __syncthreads();
  float4 acc[LMAX + 1];
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) acc[l] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int u = 0; u < RF; ++u) {
      const float* dr = sdp + (ty + 4 * (RF * k + u)) * 2 * L;
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < L) {
          const float d = dr[l];
          acc[l].x += d * xs[k][u].x; acc[l].y += d * xs[k][u].y; acc[l].z += d * xs[k][u].z; acc[l].w += d * xs[k][u].w;
        }
      acc[LMAX].x += gs[k][u].x; acc[LMAX].y += gs[k][u].y; acc[LMAX].z += gs[k][u].z; acc[LMAX].w += gs[k][u].w;
    }
#pragma unroll
  for (int l = 0; l <= LMAX; ++l) {
    if (l < L || l == LMAX) {
      red[ty][tx] = acc[l];
      __syncthreads();
      if (ty == 0 && j < D) {
        const int slot = l == LMAX ? L : l;
        const float4 a0 = red[0][tx], a1 = red[1][tx], a2 = red[2][tx], a3 = red[3][tx];
        put4(part + ((int64_t)blockIdx.y * (L + 1) + slot) * DP, j, D, true,
             make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                         (a0.w + a1.w) + (a2.w + a3.w)));
      }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0 && ty == 0) {                // dq sums of the chunk: lane r adds row r's, then over the wave
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < L) {
        const float v = wave_sum(sdp[tx * 2 * L + L + l]);          // rows past the chunk were staged as zeros
        if (tx == 0) psdq[(int64_t)blockIdx.y * LMAX + l] = v;
      }
  }
}

// cross_bwd_final_kernel with a block per (64 columns, slot): slot l < L finishes dw_l, slot L the column sums of g
// and with them every db row.  The bias prefix sums Bl_l are rebuilt from b here (L loads a column).
// part: [nchunks][L + 1][DP]; block 256 = 64 columns x 4 chunk phases, eight loads in flight per thread.
__global__ __launch_bounds__(256) void cross_bwd_final_fast_kernel(const float* __restrict__ part,
                                                                  const float* __restrict__ psdq, int nchunks,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ b, int64_t wld, int D,
                                                                  int DP, int L, float* __restrict__ dw,
                                                                  float* __restrict__ db) {
  __shared__ float red[4][64];
  __shared__ float sdq[LMAX];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx, slot = blockIdx.y;
  for (int l = ty; l < L; l += 4) {                // wave ty totals the dq sums of layers ty, ty + 4 over the chunks
    float v = 0.f;
    for (int c = tx; c < nchunks; c += 64) v += psdq[(int64_t)c * LMAX + l];
    v = wave_sum(v);
    if (tx == 0) sdq[l] = v;
  }
  float acc = 0.f;
  const int jc = j < D ? j : 0;
  for (int c0 = ty; c0 < nchunks; c0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + 4 * u;
      v[u] = part[((int64_t)(c < nchunks ? c : 0) * (L + 1) + slot) * DP + jc];
      v[u] = c < nchunks ? v[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty != 0 || j >= D) return;
  const float tot = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  if (slot < L) {                                  // dw_l = sum_b dp_l x0 + sdq_l Bl_l
    float bl = 0.f;
    for (int l = 0; l < slot; ++l) bl += b[(int64_t)l * wld + j];
    dw[(int64_t)slot * wld + j] = tot + sdq[slot] * bl;
  } else {                                         // dBl_L = colsum(g); dBl_l = sdq_l w_l; db_l' = sum_{l > l'} dBl_l
    float run = tot;
    for (int l = L - 1; l >= 0; --l) {
      db[(int64_t)l * wld + j] = run;
      run += sdq[l] * w[(int64_t)l * wld + j];
    }
  }
}

static int cross_chunks(int B) {
  int64_t n = cdiv(B, 128);       // 128 rows per chunk: 4 waves x 8 rows in flight x 4 rounds
  if (n < 1) n = 1;
  if (n > 128) n = 128;
  return (int)n;
}

// LDS bytes of the persistent kernels ((L or L + 1) rows of DP floats); above the limit the per-sample kernels run
static size_t cross_lds_bytes(int D, int rows) { return (size_t)rows * cross_dp_of(D) * sizeof(float); }
constexpr size_t CROSS_LDS_LIMIT = 64 * 1024;

// rows that the aligned kernels may read a whole quad at a time, up to column DP, through a 32-bit buffer offset
static bool quads_ok(const void* p, int64_t ld, int DP, int B) {
  return ((uintptr_t)p % 16) == 0 && ld % 4 == 0 && ld >= DP && (int64_t)B * ld * 4 < ((int64_t)1 << 31);
}
// quads per lane of the sample-per-16-lanes kernels (a multiple of four), their LDS bytes, and whether they apply
static int r16_nq(int D) { return (int)(cdiv(cdiv(D, 4), 16) + 3) / 4 * 4; }
static size_t r16_lds_bytes(int D, int rows) { return ((size_t)rows * r16_nq(D) * 64 + LMAX) * sizeof(float); }
static unsigned r16_blocks(int B) { return (unsigned)std::min<int64_t>(cdiv(B, R16_NT / 16), 256); }   // one per CU
#define R16_DISPATCH(KERNEL, NQV, GRID, BLOCK, LDS, ST, ...)                                  \
  switch (NQV) {                                                                              \
    case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;          \
    case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;          \
    case 12: hipLaunchKernelGGL(KERNEL<12>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
    case 16: hipLaunchKernelGGL(KERNEL<16>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
    case 20: hipLaunchKernelGGL(KERNEL<20>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
    case 24: hipLaunchKernelGGL(KERNEL<24>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
    case 28: hipLaunchKernelGGL(KERNEL<28>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
    default: hipLaunchKernelGGL(KERNEL<32>, GRID, BLOCK, LDS, ST, __VA_ARGS__); break;        \
  }
static bool r16_ok(int D, int rows) { return r16_nq(D) <= 32 && r16_lds_bytes(D, rows) <= CROSS_LDS_LIMIT; }
static int cross_dp(int D) { return cross_dp_of(D); }     // row stride of the workspace's own [.., D] arrays
static size_t cross_prep_floats(int D, int L) { return align_up((size_t)(L + 1) * cross_dp(D) + LMAX, 64); }

}  // namespace ctr

using namespace ctr;

extern "C" size_t ctr_cross_fwd_workspace_bytes(int D, int L) { return cross_prep_floats(D, L) * sizeof(float); }

static int cross_chunks_fast(int B) { return (int)cdiv(B, PF_ROWS); }
extern "C" size_t ctr_cross_bwd_workspace_bytes(int B, int D, int L) {
  const size_t chunks = (size_t)std::max(cross_chunks(B), cross_chunks_fast(B));
  return (cross_prep_floats(D, L) + align_up((size_t)B * 2 * L, 64) + chunks * LMAX +
          chunks * (L + 1) * cross_dp(D)) * sizeof(float);
}

#define CROSS_DISPATCH(KERNEL, ...)                                                              \
  do {                                                                                           \
    const int nq = (int)cdiv(D, 256);                                                            \
    const dim3 grid((unsigned)cdiv(B, 4)), block(256);                                           \
    if (nq <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, block, 0, st, __VA_ARGS__);                 \
    else if (nq <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, st, __VA_ARGS__);            \
    else if (nq <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, st, __VA_ARGS__);            \
    else if (nq <= 8) hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, st, __VA_ARGS__);            \
    else if (nq <= 16) hipLaunchKernelGGL(KERNEL<16>, grid, block, 0, st, __VA_ARGS__);          \
    else hipLaunchKernelGGL(KERNEL<32>, grid, block, 0, st, __VA_ARGS__);                        \
  } while (0)

// persistent launches: two blocks per CU keep 2048 waves walking the samples
#define CROSS_DISPATCH_LDS(KERNEL, LDS_BYTES, ...)                                               \
  do {                                                                                           \
    const int nq = (int)cdiv(D, 256);                                                            \
    int64_t nb = cdiv(B, 4);                                                                     \
    if (nb > 512) nb = 512;                                                                      \
    const dim3 grid((unsigned)nb), block(256);                                                   \
    if (nq <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, block, LDS_BYTES, st, __VA_ARGS__);         \
    else if (nq <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, LDS_BYTES, st, __VA_ARGS__);    \
    else if (nq <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, LDS_BYTES, st, __VA_ARGS__);    \
    else hipLaunchKernelGGL(KERNEL<8>, grid, block, LDS_BYTES, st, __VA_ARGS__);                 \
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
  const int DP = cross_dp(D);
  float* Bl = (float*)workspace;
  float* q = Bl + (size_t)(L + 1) * DP;
  hipLaunchKernelGGL(cross_prep_kernel, dim3(1), dim3(PREP_THREADS), 0, st, w, b, wld, D, DP, L, Bl, q);
  const bool lds_fits = D <= 2048 && cross_lds_bytes(D, L + 1) <= CROSS_LDS_LIMIT;
  if (aux_out && r16_ok(D, L + 1) && quads_ok(x0, x0_ld, DP, B) && quads_ok(xL, xL_ld, DP, B)) {
    const size_t lds = r16_lds_bytes(D, L + 1);
    const dim3 grid(r16_blocks(B)), block(R16_NT);
    R16_DISPATCH(cross_fwd_r16_kernel, r16_nq(D), grid, block, lds, st, x0, x0_ld, w, wld, Bl + (size_t)L * DP, q, B, D, L,
                 xL, xL_ld, aux_out);
  }
  else if (lds_fits)
    CROSS_DISPATCH_LDS(cross_fwd_lds_kernel, cross_lds_bytes(D, L + 1), x0, x0_ld, w, wld, Bl, DP, q, B, D, L, xL, xL_ld,
                       aux_out);
  else
    CROSS_DISPATCH(cross_fwd_kernel, x0, x0_ld, w, wld, Bl, DP, q, B, D, L, xL, xL_ld, aux_out);
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
  const int DP = cross_dp(D);
  float* Bl = (float*)workspace;
  float* q = Bl + (size_t)(L + 1) * DP;
  float* dpq = Bl + cross_prep_floats(D, L);
  if (r16_ok(D, L) && quads_ok(x0, x0_ld, DP, B) && quads_ok(g, g_ld, DP, B) && quads_ok(dx0, dx0_ld, DP, B)) {
    const int nchunks = cross_chunks_fast(B);
    float* psdq = dpq + align_up((size_t)B * 2 * L, 64);
    float* part = psdq + (size_t)nchunks * LMAX;
    const dim3 grid(r16_blocks(B)), block(R16_NT);
    R16_DISPATCH(cross_bwd_r16_kernel, r16_nq(D), grid, block, r16_lds_bytes(D, L), st, x0, x0_ld, w, wld, aux, g, g_ld, B, D,
                 L, dx0, dx0_ld, accumulate, dpq);
    hipLaunchKernelGGL(cross_bwd_partial_fast_kernel, dim3((unsigned)cdiv(D, 256), nchunks), dim3(256), 0, st, x0, x0_ld, g,
                       g_ld, dpq, B, D, DP, L, part, psdq);
    hipLaunchKernelGGL(cross_bwd_final_fast_kernel, dim3((unsigned)cdiv(D, 64), L + 1), dim3(256), 0, st, part, psdq,
                       nchunks, w, b, wld, D, DP, L, dw, db);
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  const int nchunks = cross_chunks(B);
  const int rows_per_chunk = (int)cdiv(B, nchunks);
  float* psdq = dpq + align_up((size_t)B * 2 * L, 64);
  float* part = psdq + (size_t)nchunks * LMAX;
  hipLaunchKernelGGL(cross_prep_kernel, dim3(1), dim3(PREP_THREADS), 0, st, w, b, wld, D, DP, L, Bl, q);
  if (D <= 2048 && cross_lds_bytes(D, L) <= CROSS_LDS_LIMIT)
    CROSS_DISPATCH_LDS(cross_bwd_lds_kernel, cross_lds_bytes(D, L), x0, x0_ld, w, wld, DP, aux, g, g_ld, B, D, L, dx0, dx0_ld,
                       accumulate, dpq);
  else
    CROSS_DISPATCH(cross_bwd_kernel, x0, x0_ld, w, wld, aux, g, g_ld, B, D, L, dx0, dx0_ld, accumulate, dpq);
  hipLaunchKernelGGL(cross_bwd_partial_kernel, dim3((unsigned)cdiv(D, 256), nchunks), dim3(256), 0, st, x0,
                     x0_ld, g, g_ld, dpq, B, D, DP, L, rows_per_chunk, part, psdq);
  hipLaunchKernelGGL(cross_bwd_final_kernel, dim3((unsigned)cdiv(D, 64)), dim3(64 * CF_PH), 0, st, part, psdq, nchunks, w,
                     wld, Bl, D, DP, L, dw, db);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
