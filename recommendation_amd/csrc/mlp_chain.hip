// Chained narrow tower layers in one kernel: h1 = act(x W1 + b1), h2 = act(h1 W2 + b2), h3 = act(h2 W3 + b3)
// with act = dropout(relu(.)), every h written once (the backward needs them) and never read back.
//
// A wave owns a 16-row strip.  Layer 1 streams its rows of x from global memory straight into
// 16x16x4 fp32 MFMA operands (lane (row l16, k-quad kq) loads x[row][16j + 4kq .. +3]) against W1 held
// in LDS.  The accumulators are kept transposed (weights as the first MFMA operand), so lane (l16, kq)
// ends with h[row l16][16t + 4kq + r], r = 0..3 - which is exactly the A-operand fragment of k-step t of
// the next layer.  Layers 2 and 3 therefore take their input from registers with no data movement;
// their weights (small) come from L2 as B fragments.
#include "common.h"

namespace ctr {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// bias: the lane's four bias values (columns n .. n + 3), fetched once per kernel
__device__ static inline f32x4 act4(f32x4 v, f32x4 bias, int n, const Drop& d, int64_t m, int N) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = fmaxf(v[r] + bias[r], 0.f);
    if (d.on) x *= drop_mul(d, (uint64_t)m * (uint64_t)N + (uint64_t)(n + r));
    o[r] = x;
  }
  return o;
}

// the last layer's output may sit at any column of a wider buffer (a concat input of the output unit)
__device__ static inline void store4(float* p, f32x4 v, bool scalar) {
  if (scalar) {
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = v[r];
  } else {
    *reinterpret_cast<f32x4*>(p) = v;
  }
}

// widths N1 = 16*NT1, N2 = 16*NT2, N3 = 16*NT3 (NT3 = 0: two layers); a wave owns 16 rows
template <int NT1, int NT2, int NT3>
__global__ __launch_bounds__(512) void mlp_chain_fwd_kernel(int M, int K0, const float* __restrict__ X, int64_t ldx,
                                                           const float* __restrict__ W1, const float* __restrict__ b1,
                                                           const float* __restrict__ W2, const float* __restrict__ b2,
                                                           const float* __restrict__ W3, const float* __restrict__ b3,
                                                           Drop d1, Drop d2, Drop d3, float* __restrict__ H1, int64_t ld1,
                                                           float* __restrict__ H2, int64_t ld2, float* __restrict__ H3,
                                                           int64_t ld3, int last_scalar) {
  d1 = resolve_drop(d1); d2 = resolve_drop(d2); d3 = resolve_drop(d3);
  extern __shared__ float Ws[];  // W1 as [K0][N1 + 4]
  constexpr int N1 = 16 * NT1, N2 = 16 * NT2, N3 = 16 * NT3, SN = N1 + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  for (int q = threadIdx.x; q < K0 * N1; q += 512) Ws[(q / N1) * SN + q % N1] = W1[q];
  __syncthreads();
  const int strips = (M + 15) / 16;
  const int ksteps = K0 / 16;
  f32x4 b1v[NT1], b2v[NT2], b3v[NT3 > 0 ? NT3 : 1];
#pragma unroll
  for (int t = 0; t < NT1; ++t) b1v[t] = f32x4{b1[16 * t + 4 * kq], b1[16 * t + 4 * kq + 1], b1[16 * t + 4 * kq + 2], b1[16 * t + 4 * kq + 3]};
#pragma unroll
  for (int t = 0; t < NT2; ++t) b2v[t] = f32x4{b2[16 * t + 4 * kq], b2[16 * t + 4 * kq + 1], b2[16 * t + 4 * kq + 2], b2[16 * t + 4 * kq + 3]};
#pragma unroll
  for (int t = 0; t < NT3; ++t) b3v[t] = f32x4{b3[16 * t + 4 * kq], b3[16 * t + 4 * kq + 1], b3[16 * t + 4 * kq + 2], b3[16 * t + 4 * kq + 3]};
  for (int strip = blockIdx.x * 8 + wave; strip < strips; strip += gridDim.x * 8) {
    const int r0 = strip * 16;
    const int64_t m = r0 + l16;
    const bool live = m < M;
    const float* x0 = X + (live ? m : (int64_t)M - 1) * ldx + 4 * kq;
    // ---- layer 1: x from global memory (four k-steps of the row in flight), W1 from LDS
    f32x4 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xq[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) xq[s4] = *reinterpret_cast<const float4*>(x0 + 16 * (s4 < ksteps ? s4 : 0));
    for (int j0 = 0; j0 < ksteps; j0 += 4) {
      float4 cur[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) cur[s4] = xq[s4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int jn = j0 + 4 + s4;
        if (jn < ksteps) xq[s4] = *reinterpret_cast<const float4*>(x0 + 16 * jn);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int j = j0 + s4;
        if (j >= ksteps) break;
        const float av[4] = {cur[s4].x, cur[s4].y, cur[s4].z, cur[s4].w};
        const float* wrow = Ws + (16 * j + 4 * kq) * SN + l16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int t = 0; t < NT1; ++t)
            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[u * SN + 16 * t], av[u], acc1[t], 0, 0, 0);
        }
        asm volatile("" ::: "memory");   // one k-step of W1 fragments in registers at a time
      }
    }
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int n = 16 * t + 4 * kq;
      acc1[t] = act4(acc1[t], b1v[t], n, d1, m, N1);
      if (live) *reinterpret_cast<f32x4*>(H1 + m * ld1 + n) = acc1[t];
    }
    // ---- layer 2: input = acc1 (k-step t of this layer is accumulator tile t of the last), W2 from L2
    f32x4 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wp = W2 + (int64_t)(16 * j + 4 * kq + u) * N2 + l16;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
          acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[16 * t], acc1[j][u], acc2[t], 0, 0, 0);
      }
      asm volatile("" ::: "memory");   // keeps the weight loads of later k-steps from piling up in registers
    }
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      const int n = 16 * t + 4 * kq;
      acc2[t] = act4(acc2[t], b2v[t], n, d2, m, N2);
      if (live) store4(H2 + m * ld2 + n, acc2[t], NT3 == 0 && last_scalar);
    }
    // ---- layer 3
    if (NT3 > 0) {
      f32x4 acc3[NT3 > 0 ? NT3 : 1];
#pragma unroll
      for (int t = 0; t < NT3; ++t) acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* wp = W3 + (int64_t)(16 * j + 4 * kq + u) * N3 + l16;
#pragma unroll
          for (int t = 0; t < NT3; ++t)
            acc3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[16 * t], acc2[j][u], acc3[t], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int t = 0; t < NT3; ++t) {
        const int n = 16 * t + 4 * kq;
        const f32x4 o = act4(acc3[t], b3v[t], n, d3, m, N3);
        if (live) store4(H3 + m * ld3 + n, o, last_scalar);
      }
    }
  }
}

// Backward of the same tail with respect to the activations, two layers in one launch:
//   g_mid = (g_in  W_b^T) * [h_mid > 0] * scale_mid      (W_b: [MID, IN],  the layer mid -> in)
//   g_out = (g_mid W_a^T) * [h_out > 0] * scale_out      (W_a: [OUT, MID], the layer out -> mid)
// g_in is the (already masked) gradient of the last layer's output.  Same register flow as the forward: a
// layer's accumulators (lane holds 4 consecutive columns of its row) are the next product's operand; the
// transposed weights sit in LDS for the whole kernel.  Both gradients are written (the weight gradients read them).
template <int NTI, int NTM, int NTO>
__global__ __launch_bounds__(256) void mlp_chain_bwd_kernel(int M, const float* __restrict__ Gin, int64_t ldgi,
                                                           int in_scalar, const float* __restrict__ Wb,
                                                           const float* __restrict__ Hm, int64_t ldhm, float scale_m,
                                                           float* __restrict__ Gm, int64_t ldgm,
                                                           const float* __restrict__ Wa,
                                                           const float* __restrict__ Ho, int64_t ldho, float scale_o,
                                                           float* __restrict__ Go, int64_t ldgo,
                                                           float* __restrict__ go_amax) {
  extern __shared__ float Ws[];
  constexpr int IN = 16 * NTI, MID = 16 * NTM, OUT = 16 * NTO, SB = MID + 4, SA = OUT + 4;
  float amx = 0.f;                 // largest magnitude this lane writes to Go
  float* Wbt = Ws;                 // [IN][SB]:  Wbt[k][n] = Wb[n][k]
  float* Wat = Ws + IN * SB;       // [MID][SA]: Wat[k][n] = Wa[n][k]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  for (int q = threadIdx.x; q < MID * IN; q += 256) Wbt[(q % IN) * SB + q / IN] = Wb[q];
  for (int q = threadIdx.x; q < OUT * MID; q += 256) Wat[(q % MID) * SA + q / MID] = Wa[q];
  __syncthreads();
  const int strips = (M + 15) / 16;
  for (int strip = blockIdx.x * 4 + wave; strip < strips; strip += gridDim.x * 4) {
    const int64_t m = strip * 16 + l16;
    const bool live = m < M;
    const int64_t mc = live ? m : (int64_t)M - 1;
    f32x4 gi[NTI];
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      const float* p = Gin + mc * ldgi + 16 * j + 4 * kq;
      if (in_scalar) gi[j] = f32x4{p[0], p[1], p[2], p[3]};
      else gi[j] = *reinterpret_cast<const f32x4*>(p);
    }
    f32x4 hm[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t) hm[t] = *reinterpret_cast<const f32x4*>(Hm + mc * ldhm + 16 * t + 4 * kq);
    f32x4 gm[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t) gm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NTI; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wrow = Wbt + (16 * j + 4 * kq + u) * SB + l16;
#pragma unroll
        for (int t = 0; t < NTM; ++t) gm[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[16 * t], gi[j][u], gm[t], 0, 0, 0);
      }
    f32x4 ho[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) ho[t] = *reinterpret_cast<const f32x4*>(Ho + mc * ldho + 16 * t + 4 * kq);
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gm[t][r] = hm[t][r] > 0.f ? gm[t][r] * scale_m : 0.f;
      if (live) *reinterpret_cast<f32x4*>(Gm + m * ldgm + 16 * t + 4 * kq) = gm[t];
    }
    f32x4 go[NTO];
#pragma unroll
    for (int t = 0; t < NTO; ++t) go[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NTM; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wrow = Wat + (16 * j + 4 * kq + u) * SA + l16;
#pragma unroll
        for (int t = 0; t < NTO; ++t) go[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[16 * t], gm[j][u], go[t], 0, 0, 0);
      }
#pragma unroll
    for (int t = 0; t < NTO; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) go[t][r] = ho[t][r] > 0.f ? go[t][r] * scale_o : 0.f;
      if (live) {
        *reinterpret_cast<f32x4*>(Go + m * ldgo + 16 * t + 4 * kq) = go[t];
        amx = fmaxf(fmaxf(amx, fmaxf(fabsf(go[t][0]), fabsf(go[t][1]))), fmaxf(fabsf(go[t][2]), fabsf(go[t][3])));
      }
    }
  }
  if (go_amax) raise_amax(go_amax, amx);
}


// Weight and bias gradients of the last two (narrow) tower layers in one launch:
//   dWa = Xa^T Ga [KA x NA],  dba = sum_rows Ga,   dWb = Xb^T Gb [KB x NB],  dbb = sum_rows Gb
// (Xa, Xb the layers' inputs, Ga, Gb the gradients at their outputs, all M rows).  Launched per layer these
// products are two passes over M-row operands with a few tiles of output each, latency-bound; here a wave
// walks 16-row strips, stages a strip of all four operands in its own LDS area with whole-row 16-byte loads
// (the next strip's loads in flight meanwhile) and feeds rows^T x rows 16x16x4 fp32 MFMAs whose accumulator
// tiles (KA/16 * NA/16 + KB/16 * NB/16 of them) stay in registers for the whole launch.  The four waves of a
// block add their tiles in wave order through LDS; tail_wgrad2_reduce_kernel adds the block partials in a
// fixed order.
template <int KAT, int NAT, int KBT, int NBT>
struct TailWgradLayout {
  static constexpr int KA = 16 * KAT, NA = 16 * NAT, KB = 16 * KBT, NB = 16 * NBT;
  static constexpr int SXA = KA + 4, SGA = NA + 4, SXB = KB + 4, SGB = NB + 4;       // LDS strides of a strip
  static constexpr int XA = 0, GA = 16 * SXA, XB = GA + 16 * SGA, GB = XB + 16 * SXB, WAVE = GB + 16 * SGB;
  static constexpr int WA = 0, WB = KA * NA, BA = WB + KB * NB, BB = BA + NA, PART = BB + NB;   // a partial's layout
};

template <int KAT, int NAT, int KBT, int NBT>
__global__ __launch_bounds__(256, 1) void tail_wgrad2_kernel(int M, const float* __restrict__ Xa, int64_t ldxa,
                                                            const float* __restrict__ Ga, int64_t ldga,
                                                            const float* __restrict__ Xb, int64_t ldxb,
                                                            const float* __restrict__ Gb, int64_t ldgb, int gb_scalar,
                                                            float* __restrict__ part) {
  using L = TailWgradLayout<KAT, NAT, KBT, NBT>;
  constexpr int KA = L::KA, NA = L::NA, KB = L::KB, NB = L::NB;
  constexpr int FXA = 16 * KA / 4, FGA = 16 * NA / 4, FXB = 16 * KB / 4, FGB = 16 * NB / 4;   // float4s per strip
  constexpr int NXA = (FXA + 63) / 64, NGA = (FGA + 63) / 64, NXB = (FXB + 63) / 64, NGB = (FGB + 63) / 64;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  float* my = lds + wave * L::WAVE;
  f32x4 accA[KAT][NAT], accB[KBT][NBT];
#pragma unroll
  for (int i = 0; i < KAT; ++i)
#pragma unroll
    for (int t = 0; t < NAT; ++t) accA[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KBT; ++i)
#pragma unroll
    for (int t = 0; t < NBT; ++t) accB[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sba[NAT], sbb[NBT];
#pragma unroll
  for (int t = 0; t < NAT; ++t) sba[t] = 0.f;
#pragma unroll
  for (int t = 0; t < NBT; ++t) sbb[t] = 0.f;
  struct Regs { f32x4 xa[NXA], ga[NGA], xb[NXB], gb[NGB]; };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // piece f of a strip of C-column rows: row f / (C/4), columns 4 (f % (C/4)) .. +3; rows past M are zeros
  auto load_strip = [&](int64_t strip, Regs& r) {
    const int64_t m0 = strip * 16;
    const int nrows = M - m0 < 16 ? (int)(M - m0) : 16;
#pragma unroll
    for (int i = 0; i < NXA; ++i) {
      const int f = lane + 64 * i, row = f / (KA / 4), c = 4 * (f % (KA / 4));
      r.xa[i] = (f < FXA && row < nrows) ? *reinterpret_cast<const f32x4*>(Xa + (m0 + row) * ldxa + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
      const int f = lane + 64 * i, row = f / (NA / 4), c = 4 * (f % (NA / 4));
      r.ga[i] = (f < FGA && row < nrows) ? *reinterpret_cast<const f32x4*>(Ga + (m0 + row) * ldga + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NXB; ++i) {
      const int f = lane + 64 * i, row = f / (KB / 4), c = 4 * (f % (KB / 4));
      r.xb[i] = (f < FXB && row < nrows) ? *reinterpret_cast<const f32x4*>(Xb + (m0 + row) * ldxb + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NGB; ++i) {
      const int f = lane + 64 * i, row = f / (NB / 4), c = 4 * (f % (NB / 4));
      f32x4 v = zero4;
      if (f < FGB && row < nrows) {
        const float* p = Gb + (m0 + row) * ldgb + c;
        if (gb_scalar) v = f32x4{p[0], p[1], p[2], p[3]};
        else v = *reinterpret_cast<const f32x4*>(p);
      }
      r.gb[i] = v;
    }
  };
  auto store_strip = [&](const Regs& r) {
#pragma unroll
    for (int i = 0; i < NXA; ++i) {
      const int f = lane + 64 * i;
      if (FXA % 64 == 0 || f < FXA) *reinterpret_cast<f32x4*>(my + L::XA + (f / (KA / 4)) * L::SXA + 4 * (f % (KA / 4))) = r.xa[i];
    }
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
      const int f = lane + 64 * i;
      if (FGA % 64 == 0 || f < FGA) *reinterpret_cast<f32x4*>(my + L::GA + (f / (NA / 4)) * L::SGA + 4 * (f % (NA / 4))) = r.ga[i];
    }
#pragma unroll
    for (int i = 0; i < NXB; ++i) {
      const int f = lane + 64 * i;
      if (FXB % 64 == 0 || f < FXB) *reinterpret_cast<f32x4*>(my + L::XB + (f / (KB / 4)) * L::SXB + 4 * (f % (KB / 4))) = r.xb[i];
    }
#pragma unroll
    for (int i = 0; i < NGB; ++i) {
      const int f = lane + 64 * i;
      if (FGB % 64 == 0 || f < FGB) *reinterpret_cast<f32x4*>(my + L::GB + (f / (NB / 4)) * L::SGB + 4 * (f % (NB / 4))) = r.gb[i];
    }
  };
  const int64_t strips = ((int64_t)M + 15) / 16;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  Regs regs;
  if (gw < strips) load_strip(gw, regs);
  for (int64_t strip = gw; strip < strips; strip += nw) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the last strip's reads are done before it is overwritten
    __builtin_amdgcn_wave_barrier();
    store_strip(regs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (strip + nw < strips) load_strip(strip + nw, regs);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * kq + u;                              // a product step takes rows u, 4+u, 8+u, 12+u
      float xa[KAT], ga[NAT], xb[KBT], gb[NBT];
#pragma unroll
      for (int i = 0; i < KAT; ++i) xa[i] = my[L::XA + row * L::SXA + 16 * i + l16];
#pragma unroll
      for (int t = 0; t < NAT; ++t) ga[t] = my[L::GA + row * L::SGA + 16 * t + l16];
#pragma unroll
      for (int i = 0; i < KBT; ++i) xb[i] = my[L::XB + row * L::SXB + 16 * i + l16];
#pragma unroll
      for (int t = 0; t < NBT; ++t) gb[t] = my[L::GB + row * L::SGB + 16 * t + l16];
#pragma unroll
      for (int i = 0; i < KAT; ++i)
#pragma unroll
        for (int t = 0; t < NAT; ++t) accA[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], ga[t], accA[i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < KBT; ++i)
#pragma unroll
        for (int t = 0; t < NBT; ++t) accB[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i], gb[t], accB[i][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NAT; ++t) sba[t] += ga[t];
#pragma unroll
      for (int t = 0; t < NBT; ++t) sbb[t] += gb[t];
    }
  }
  // column sums: lanes kq = 0..3 hold different rows of the same column
#pragma unroll
  for (int t = 0; t < NAT; ++t) { sba[t] += __shfl_xor(sba[t], 16); sba[t] += __shfl_xor(sba[t], 32); }
#pragma unroll
  for (int t = 0; t < NBT; ++t) { sbb[t] += __shfl_xor(sbb[t], 16); sbb[t] += __shfl_xor(sbb[t], 32); }
  // the block's four waves add their tiles in wave order in LDS (the staging areas are free now)
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
      // tile (i, t): lane (l16, kq) holds D[16 i + 4 kq + r][16 t + l16]
#pragma unroll
      for (int i = 0; i < KAT; ++i)
#pragma unroll
        for (int t = 0; t < NAT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = lds + L::WA + (16 * i + 4 * kq + r) * NA + 16 * t + l16;
            *q = w == 0 ? accA[i][t][r] : *q + accA[i][t][r];
          }
#pragma unroll
      for (int i = 0; i < KBT; ++i)
#pragma unroll
        for (int t = 0; t < NBT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = lds + L::WB + (16 * i + 4 * kq + r) * NB + 16 * t + l16;
            *q = w == 0 ? accB[i][t][r] : *q + accB[i][t][r];
          }
      if (kq == 0) {
#pragma unroll
        for (int t = 0; t < NAT; ++t) {
          float* q = lds + L::BA + 16 * t + l16;
          *q = w == 0 ? sba[t] : *q + sba[t];
        }
#pragma unroll
        for (int t = 0; t < NBT; ++t) {
          float* q = lds + L::BB + 16 * t + l16;
          *q = w == 0 ? sbb[t] : *q + sbb[t];
        }
      }
    }
    __syncthreads();
  }
  float* P = part + (int64_t)blockIdx.x * L::PART;
  for (int i = threadIdx.x; i < L::PART; i += 256) P[i] = lds[i];
}

// The two-layer backward chain and the two layers' weight / bias gradients in ONE pass over the activations:
//   g_mid = (g_in W_b^T) * [h_mid > 0] * scale_mid,   g_out = (g_mid W_a^T) * [h_out > 0] * scale_out   (written)
//   dW_b = h_mid^T g_in,  db_b = sum g_in,   dW_a = h_out^T g_mid,  db_a = sum g_mid
// with (OUT, MID, IN) = 16 * (KAT, NAT = KBT, NBT).  A wave stages a 16-row strip of h_out, h_mid and g_in in its LDS
// area (whole-row 16-byte loads, the next strip in flight), runs the chain on row-layout fragments read back from
// there (transposed weights in LDS, as mlp_chain_bwd_kernel), parks g_mid beside them and then reads all four
// column-wise for the rows^T x rows products, whose accumulator tiles stay in registers for the whole launch.
// g_mid never goes to memory; h_out, h_mid and g_in are read once.  Partials as tail_wgrad2_kernel.
template <int KAT, int NAT, int NBT>
__global__ __launch_bounds__(256, 1) void tail_bwd_fused_kernel(int M, const float* __restrict__ Gin, int64_t ldgi,
                                                               int in_scalar, const float* __restrict__ Wb,
                                                               const float* __restrict__ Hm, int64_t ldhm, float scale_m,
                                                               const float* __restrict__ Wa, const float* __restrict__ Ho,
                                                               int64_t ldho, float scale_o, float* __restrict__ Go,
                                                               int64_t ldgo, float* __restrict__ part) {
  using L = TailWgradLayout<KAT, NAT, NAT, NBT>;
  constexpr int OUT = 16 * KAT, MID = 16 * NAT, IN = 16 * NBT, SB = MID + 4, SA = OUT + 4;
  constexpr int FHO = 16 * OUT / 4, FHM = 16 * MID / 4, FGI = 16 * IN / 4;
  constexpr int NHO = (FHO + 63) / 64, NHM = (FHM + 63) / 64, NGI = (FGI + 63) / 64;
  extern __shared__ float lds[];
  float* Wbt = lds;                          // [IN][SB]:  Wbt[k][n] = Wb[n][k]
  float* Wat = Wbt + IN * SB;                // [MID][SA]: Wat[k][n] = Wa[n][k]
  float* stage = Wat + MID * SA;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  float* my = stage + wave * L::WAVE;
  for (int q = threadIdx.x; q < MID * IN; q += 256) Wbt[(q % IN) * SB + q / IN] = Wb[q];
  for (int q = threadIdx.x; q < OUT * MID; q += 256) Wat[(q % MID) * SA + q / MID] = Wa[q];
  __syncthreads();
  f32x4 accA[KAT][NAT], accB[NAT][NBT];
#pragma unroll
  for (int i = 0; i < KAT; ++i)
#pragma unroll
    for (int t = 0; t < NAT; ++t) accA[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NAT; ++i)
#pragma unroll
    for (int t = 0; t < NBT; ++t) accB[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sba[NAT], sbb[NBT];
#pragma unroll
  for (int t = 0; t < NAT; ++t) sba[t] = 0.f;
#pragma unroll
  for (int t = 0; t < NBT; ++t) sbb[t] = 0.f;
  struct Regs { f32x4 ho[NHO], hm[NHM], gi[NGI]; };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_strip = [&](int64_t strip, Regs& r) {
    const int64_t m0 = strip * 16;
    const int nrows = M - m0 < 16 ? (int)(M - m0) : 16;
#pragma unroll
    for (int i = 0; i < NHO; ++i) {
      const int f = lane + 64 * i, row = f / (OUT / 4), c = 4 * (f % (OUT / 4));
      r.ho[i] = (f < FHO && row < nrows) ? *reinterpret_cast<const f32x4*>(Ho + (m0 + row) * ldho + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NHM; ++i) {
      const int f = lane + 64 * i, row = f / (MID / 4), c = 4 * (f % (MID / 4));
      r.hm[i] = (f < FHM && row < nrows) ? *reinterpret_cast<const f32x4*>(Hm + (m0 + row) * ldhm + c) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
      const int f = lane + 64 * i, row = f / (IN / 4), c = 4 * (f % (IN / 4));
      f32x4 v = zero4;
      if (f < FGI && row < nrows) {
        const float* p = Gin + (m0 + row) * ldgi + c;
        if (in_scalar) v = f32x4{p[0], p[1], p[2], p[3]};
        else v = *reinterpret_cast<const f32x4*>(p);
      }
      r.gi[i] = v;
    }
  };
  auto store_strip = [&](const Regs& r) {
#pragma unroll
    for (int i = 0; i < NHO; ++i) {
      const int f = lane + 64 * i;
      if (FHO % 64 == 0 || f < FHO) *reinterpret_cast<f32x4*>(my + L::XA + (f / (OUT / 4)) * L::SXA + 4 * (f % (OUT / 4))) = r.ho[i];
    }
#pragma unroll
    for (int i = 0; i < NHM; ++i) {
      const int f = lane + 64 * i;
      if (FHM % 64 == 0 || f < FHM) *reinterpret_cast<f32x4*>(my + L::XB + (f / (MID / 4)) * L::SXB + 4 * (f % (MID / 4))) = r.hm[i];
    }
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
      const int f = lane + 64 * i;
      if (FGI % 64 == 0 || f < FGI) *reinterpret_cast<f32x4*>(my + L::GB + (f / (IN / 4)) * L::SGB + 4 * (f % (IN / 4))) = r.gi[i];
    }
  };
  const int64_t strips = ((int64_t)M + 15) / 16;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  Regs regs;
  if (gw < strips) load_strip(gw, regs);
  for (int64_t strip = gw; strip < strips; strip += nw) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the last strip's reads are done before it is overwritten
    __builtin_amdgcn_wave_barrier();
    store_strip(regs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (strip + nw < strips) load_strip(strip + nw, regs);
    const int64_t m = strip * 16 + l16;
    const bool live = m < M;
    // ---- the chain, on row-layout fragments: lane (row l16, quad kq) holds columns 16 t + 4 kq .. +3
    f32x4 gi[NBT], hm[NAT], gm[NAT];
#pragma unroll
    for (int j = 0; j < NBT; ++j) gi[j] = *reinterpret_cast<const f32x4*>(my + L::GB + l16 * L::SGB + 16 * j + 4 * kq);
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
      hm[t] = *reinterpret_cast<const f32x4*>(my + L::XB + l16 * L::SXB + 16 * t + 4 * kq);
      gm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < NBT; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* wrow = Wbt + (16 * j + 4 * kq + u) * SB + l16;
#pragma unroll
        for (int t = 0; t < NAT; ++t) gm[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[16 * t], gi[j][u], gm[t], 0, 0, 0);
        asm volatile("" ::: "memory");
      }
#pragma unroll
    for (int t = 0; t < NAT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) gm[t][r] = hm[t][r] > 0.f ? gm[t][r] * scale_m : 0.f;
      *reinterpret_cast<f32x4*>(my + L::GA + l16 * L::SGA + 16 * t + 4 * kq) = gm[t];     // parked for dW_a
    }
    {
      f32x4 go[KAT];
#pragma unroll
      for (int t = 0; t < KAT; ++t) go[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NAT; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* wrow = Wat + (16 * j + 4 * kq + u) * SA + l16;
#pragma unroll
          for (int t = 0; t < KAT; ++t) go[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[16 * t], gm[j][u], go[t], 0, 0, 0);
          asm volatile("" ::: "memory");
        }
#pragma unroll
      for (int t = 0; t < KAT; ++t) {
        const f32x4 ho = *reinterpret_cast<const f32x4*>(my + L::XA + l16 * L::SXA + 16 * t + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) go[t][r] = ho[r] > 0.f ? go[t][r] * scale_o : 0.f;
        if (live) *reinterpret_cast<f32x4*>(Go + m * ldgo + 16 * t + 4 * kq) = go[t];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // g_mid is in LDS before it is read column-wise
    __builtin_amdgcn_wave_barrier();
    // ---- the weight gradients: rows^T x rows, a product step takes rows u, 4+u, 8+u, 12+u
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * kq + u;
      float xa[KAT], ga[NAT], xb[NAT], gb[NBT];
#pragma unroll
      for (int i = 0; i < KAT; ++i) xa[i] = my[L::XA + row * L::SXA + 16 * i + l16];
#pragma unroll
      for (int t = 0; t < NAT; ++t) ga[t] = my[L::GA + row * L::SGA + 16 * t + l16];
#pragma unroll
      for (int i = 0; i < NAT; ++i) xb[i] = my[L::XB + row * L::SXB + 16 * i + l16];
#pragma unroll
      for (int t = 0; t < NBT; ++t) gb[t] = my[L::GB + row * L::SGB + 16 * t + l16];
#pragma unroll
      for (int i = 0; i < KAT; ++i)
#pragma unroll
        for (int t = 0; t < NAT; ++t) accA[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i], ga[t], accA[i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NAT; ++i)
#pragma unroll
        for (int t = 0; t < NBT; ++t) accB[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i], gb[t], accB[i][t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NAT; ++t) sba[t] += ga[t];
#pragma unroll
      for (int t = 0; t < NBT; ++t) sbb[t] += gb[t];
    }
  }
#pragma unroll
  for (int t = 0; t < NAT; ++t) { sba[t] += __shfl_xor(sba[t], 16); sba[t] += __shfl_xor(sba[t], 32); }
#pragma unroll
  for (int t = 0; t < NBT; ++t) { sbb[t] += __shfl_xor(sbb[t], 16); sbb[t] += __shfl_xor(sbb[t], 32); }
  __syncthreads();
  float* red = stage;                        // the staging areas are free now
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < KAT; ++i)
#pragma unroll
        for (int t = 0; t < NAT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = red + L::WA + (16 * i + 4 * kq + r) * MID + 16 * t + l16;
            *q = w == 0 ? accA[i][t][r] : *q + accA[i][t][r];
          }
#pragma unroll
      for (int i = 0; i < NAT; ++i)
#pragma unroll
        for (int t = 0; t < NBT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* q = red + L::WB + (16 * i + 4 * kq + r) * IN + 16 * t + l16;
            *q = w == 0 ? accB[i][t][r] : *q + accB[i][t][r];
          }
      if (kq == 0) {
#pragma unroll
        for (int t = 0; t < NAT; ++t) {
          float* q = red + L::BA + 16 * t + l16;
          *q = w == 0 ? sba[t] : *q + sba[t];
        }
#pragma unroll
        for (int t = 0; t < NBT; ++t) {
          float* q = red + L::BB + 16 * t + l16;
          *q = w == 0 ? sbb[t] : *q + sbb[t];
        }
      }
    }
    __syncthreads();
  }
  float* P = part + (int64_t)blockIdx.x * L::PART;
  for (int i = threadIdx.x; i < L::PART; i += 256) P[i] = red[i];
}

// Block partials -> the four gradients, in a fixed order: thread (group pg, element e) adds partials pg, pg + 8, ...
__global__ __launch_bounds__(256) void tail_wgrad2_reduce_kernel(const float* __restrict__ part, int nparts, int PART,
                                                                int nwa, int nwb, int nba, float* __restrict__ dWa,
                                                                float* __restrict__ dWb, float* __restrict__ dba,
                                                                float* __restrict__ dbb) {
  __shared__ float sm[8][32];
  const int pg = threadIdx.x >> 5, e = blockIdx.x * 32 + (threadIdx.x & 31);
  float s = 0.f;
  if (e < PART) {
    int p = pg;
    for (; p + 56 < nparts; p += 64) {               // eight independent loads in flight
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + 8 * u) * PART + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nparts; p += 8) s += part[(int64_t)p * PART + e];
  }
  sm[pg][threadIdx.x & 31] = s;
  __syncthreads();
  if (pg == 0 && e < PART) {
    float t = sm[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sm[g][threadIdx.x];
    if (e < nwa) dWa[e] = t;
    else if (e < nwa + nwb) dWb[e - nwa] = t;
    else if (e < nwa + nwb + nba) dba[e - nwa - nwb] = t;
    else dbb[e - nwa - nwb - nba] = t;
  }
}

}  // namespace ctr

using namespace ctr;

// Reports whether the chained kernel covers these widths (the caller falls back to one GEMM per layer).
// A 128-wide first layer is not offered: its 32 accumulator tiles plus operands leave one block per CU
// and it measures slower than a GEMM launch per layer.
extern "C" int ctr_mlp_chain_supported(int K0, int N1, int N2, int N3) {
  if (K0 < 16 || K0 % 16 || K0 > 256) return 0;
  return N1 == 64 && N2 == 32 && (N3 == 16 || N3 == 0);
}

extern "C" int ctr_mlp_chain_fwd(int M, int K0, const float* X, int64_t ldx, int N1, const float* W1, const float* b1,
                                 const ctr_dropout_t* drop1, float* H1, int64_t ld1, int N2, const float* W2,
                                 const float* b2, const ctr_dropout_t* drop2, float* H2, int64_t ld2, int N3,
                                 const float* W3, const float* b3, const ctr_dropout_t* drop3, float* H3, int64_t ld3,
                                 void* stream) {
  CTR_REQUIRE(M >= 0, "ctr_mlp_chain_fwd: negative M");
  CTR_REQUIRE(ctr_mlp_chain_supported(K0, N1, N2, N3), "ctr_mlp_chain_fwd: widths %d -> %d -> %d -> %d not covered", K0,
              N1, N2, N3);
  if (M == 0) return CTR_OK;
  CTR_REQUIRE(X && W1 && b1 && H1 && W2 && b2 && H2 && (N3 == 0 || (W3 && b3 && H3)), "ctr_mlp_chain_fwd: null pointer");
  CTR_REQUIRE(ldx >= K0 && ldx % 4 == 0 && ((uintptr_t)X % 16) == 0, "ctr_mlp_chain_fwd: x must be 16-byte aligned rows");
  CTR_REQUIRE(ld1 >= N1 && ld2 >= N2 && (N3 == 0 || ld3 >= N3), "ctr_mlp_chain_fwd: output leading dimension");
  const float* last = N3 ? H3 : H2;
  const int64_t ldl = N3 ? ld3 : ld2;
  const int last_scalar = (ldl % 4 != 0 || ((uintptr_t)last % 16) != 0) ? 1 : 0;
  CTR_REQUIRE(ld1 % 4 == 0 && ((uintptr_t)H1 % 16) == 0 && (N3 == 0 || (ld2 % 4 == 0 && ((uintptr_t)H2 % 16) == 0)),
              "ctr_mlp_chain_fwd: outputs before the last must be 16-byte aligned rows");
  const size_t lds = (size_t)K0 * (N1 + 4) * sizeof(float);
  const int strips = (int)cdiv(M, 16);
  int blocks = (int)cdiv(strips, 8);
  if (blocks > 512) blocks = 512;   // two blocks of eight waves per CU (one: 35 us against 30 at 65536 rows)
  hipStream_t st = (hipStream_t)stream;
  const Drop d1 = make_drop(drop1), d2 = make_drop(drop2), d3 = make_drop(drop3);
#define CH(A_, B_, C_)                                                                                       \
  do {                                                                                                       \
    if (lds > 64 * 1024)                                                                                     \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_chain_fwd_kernel<A_, B_, C_>),          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                    \
    hipLaunchKernelGGL((mlp_chain_fwd_kernel<A_, B_, C_>), dim3(blocks), dim3(512), lds, st, M, K0, X, ldx, W1, b1, W2, \
                       b2, W3, b3, d1, d2, d3, H1, ld1, H2, ld2, H3, ld3, last_scalar);                                   \
  } while (0)
  if (N3 == 16) CH(4, 2, 1);
  else CH(4, 2, 0);
#undef CH
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// Widths (in, mid, out) of the two-layer backward chain: the gradient walks in -> mid -> out.
extern "C" int ctr_mlp_chain_bwd_supported(int in, int mid, int out) {
  return (in == 32 && mid == 64 && out == 128) || (in == 16 && mid == 32 && out == 64);
}

extern "C" int ctr_mlp_chain_bwd(int M, int in, const float* Gin, int64_t ldgi, int mid, const float* Wb,
                                 const float* Hmid, int64_t ldhm, float scale_mid, float* Gmid, int64_t ldgm, int out,
                                 const float* Wa, const float* Hout, int64_t ldho, float scale_out, float* Gout,
                                 int64_t ldgo, float* gout_amax, void* stream) {
  CTR_REQUIRE(M >= 0, "ctr_mlp_chain_bwd: negative M");
  CTR_REQUIRE(ctr_mlp_chain_bwd_supported(in, mid, out), "ctr_mlp_chain_bwd: widths %d -> %d -> %d not covered", in, mid,
              out);
  if (M == 0) return CTR_OK;
  CTR_REQUIRE(Gin && Wb && Hmid && Gmid && Wa && Hout && Gout, "ctr_mlp_chain_bwd: null pointer");
  CTR_REQUIRE(ldgi >= in && ldhm >= mid && ldgm >= mid && ldho >= out && ldgo >= out, "ctr_mlp_chain_bwd: leading dimension");
  const auto rows16 = [](const void* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p % 16) == 0; };
  CTR_REQUIRE(rows16(Hmid, ldhm) && rows16(Gmid, ldgm) && rows16(Hout, ldho) && rows16(Gout, ldgo),
              "ctr_mlp_chain_bwd: activations and output gradients must be 16-byte aligned rows");
  const int in_scalar = rows16(Gin, ldgi) ? 0 : 1;
  const size_t lds = ((size_t)in * (mid + 4) + (size_t)mid * (out + 4)) * sizeof(float);
  const int strips = (int)cdiv(M, 16);
  int blocks = (int)cdiv(strips, 4);
  if (blocks > 512) blocks = 512;   // every block transposes both weight matrices into LDS first
  hipStream_t st = (hipStream_t)stream;
#define CB(A_, B_, C_)                                                                                         \
  hipLaunchKernelGGL((mlp_chain_bwd_kernel<A_, B_, C_>), dim3(blocks), dim3(256), lds, st, M, Gin, ldgi, in_scalar, Wb, \
                     Hmid, ldhm, scale_mid, Gmid, ldgm, Wa, Hout, ldho, scale_out, Gout, ldgo, gout_amax)
  if (in == 32) CB(2, 4, 8);
  else CB(1, 2, 4);
#undef CB
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// Widths the fused two-layer weight gradient covers: (KA, NA) = the earlier layer, (KB, NB) = the later one.
extern "C" int ctr_tail_wgrad2_supported(int ka, int na, int kb, int nb) {
  return (ka == 128 && na == 64 && kb == 64 && nb == 32) || (ka == 64 && na == 32 && kb == 32 && nb == 16);
}

static int tail_wgrad2_blocks(int M) {
  const int64_t need = cdiv(cdiv(M, 16), 4);
  return (int)(need < 256 ? (need > 0 ? need : 1) : 256);
}

extern "C" size_t ctr_tail_wgrad2_workspace_bytes(int M, int ka, int na, int kb, int nb) {
  if (M <= 0 || !ctr_tail_wgrad2_supported(ka, na, kb, nb)) return 0;
  return (size_t)tail_wgrad2_blocks(M) * ((size_t)ka * na + (size_t)kb * nb + na + nb) * sizeof(float);
}

extern "C" int ctr_tail_wgrad2(int M, int ka, const float* Xa, int64_t ldxa, int na, const float* Ga, int64_t ldga,
                               float* dWa, float* dba, int kb, const float* Xb, int64_t ldxb, int nb, const float* Gb,
                               int64_t ldgb, float* dWb, float* dbb, void* workspace, size_t workspace_bytes,
                               void* stream) {
  CTR_REQUIRE(M > 0, "ctr_tail_wgrad2: M must be positive");
  CTR_REQUIRE(ctr_tail_wgrad2_supported(ka, na, kb, nb), "ctr_tail_wgrad2: widths %d x %d, %d x %d not covered", ka, na,
              kb, nb);
  CTR_REQUIRE(Xa && Ga && dWa && dba && Xb && Gb && dWb && dbb && workspace, "ctr_tail_wgrad2: null pointer");
  CTR_REQUIRE(ldxa >= ka && ldga >= na && ldxb >= kb && ldgb >= nb, "ctr_tail_wgrad2: leading dimension");
  const auto rows16 = [](const void* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p % 16) == 0; };
  CTR_REQUIRE(rows16(Xa, ldxa) && rows16(Ga, ldga) && rows16(Xb, ldxb),
              "ctr_tail_wgrad2: Xa, Ga and Xb must be 16-byte aligned rows");
  const size_t need = ctr_tail_wgrad2_workspace_bytes(M, ka, na, kb, nb);
  if (workspace_bytes < need) {
    set_error("ctr_tail_wgrad2: workspace %zu < %zu", workspace_bytes, need);
    return CTR_EWORKSPACE;
  }
  const int gb_scalar = rows16(Gb, ldgb) ? 0 : 1;
  const int blocks = tail_wgrad2_blocks(M);
  float* part = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
#define TW(A_, B_, C_, D_)                                                                                          \
  do {                                                                                                              \
    using L = TailWgradLayout<A_, B_, C_, D_>;                                                                      \
    const size_t a = (size_t)4 * L::WAVE, b = (size_t)L::PART;                                                      \
    const size_t shm = (a > b ? a : b) * sizeof(float);                                                             \
    if (shm > 48 * 1024)                                                                                            \
      CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_wgrad2_kernel<A_, B_, C_, D_>),               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                           \
    hipLaunchKernelGGL((tail_wgrad2_kernel<A_, B_, C_, D_>), dim3(blocks), dim3(256), shm, st, M, Xa, ldxa, Ga, ldga, \
                       Xb, ldxb, Gb, ldgb, gb_scalar, part);                                                        \
    hipLaunchKernelGGL(tail_wgrad2_reduce_kernel, dim3((unsigned)cdiv(L::PART, 32)), dim3(256), 0, st, part, blocks, \
                       (int)L::PART, ka * na, kb * nb, na, dWa, dWb, dba, dbb);                                     \
  } while (0)
  if (ka == 128) TW(8, 4, 4, 2);
  else TW(4, 2, 2, 1);
#undef TW
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// The chain backward fused with the two layers' weight gradients (same widths as ctr_mlp_chain_bwd).
extern "C" int ctr_tail_bwd_fused_supported(int in, int mid, int out) { return ctr_mlp_chain_bwd_supported(in, mid, out); }

extern "C" size_t ctr_tail_bwd_fused_workspace_bytes(int M, int in, int mid, int out) {
  return ctr_tail_wgrad2_workspace_bytes(M, out, mid, mid, in);
}

extern "C" int ctr_tail_bwd_fused(int M, int in, const float* Gin, int64_t ldgi, int mid, const float* Wb,
                                  const float* Hmid, int64_t ldhm, float scale_mid, int out, const float* Wa,
                                  const float* Hout, int64_t ldho, float scale_out, float* Gout, int64_t ldgo,
                                  float* dWa, float* dba, float* dWb, float* dbb, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(M > 0, "ctr_tail_bwd_fused: M must be positive");
  CTR_REQUIRE(ctr_tail_bwd_fused_supported(in, mid, out), "ctr_tail_bwd_fused: widths %d -> %d -> %d not covered", in,
              mid, out);
  CTR_REQUIRE(Gin && Wb && Hmid && Wa && Hout && Gout && dWa && dba && dWb && dbb && workspace,
              "ctr_tail_bwd_fused: null pointer");
  CTR_REQUIRE(ldgi >= in && ldhm >= mid && ldho >= out && ldgo >= out, "ctr_tail_bwd_fused: leading dimension");
  const auto rows16 = [](const void* p, int64_t ld) { return ld % 4 == 0 && ((uintptr_t)p % 16) == 0; };
  CTR_REQUIRE(rows16(Hmid, ldhm) && rows16(Hout, ldho) && rows16(Gout, ldgo),
              "ctr_tail_bwd_fused: activations and the output gradient must be 16-byte aligned rows");
  const size_t need = ctr_tail_bwd_fused_workspace_bytes(M, in, mid, out);
  if (workspace_bytes < need) {
    set_error("ctr_tail_bwd_fused: workspace %zu < %zu", workspace_bytes, need);
    return CTR_EWORKSPACE;
  }
  const int in_scalar = rows16(Gin, ldgi) ? 0 : 1;
  const int blocks = tail_wgrad2_blocks(M);
  float* part = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
#define TF(A_, B_, C_)                                                                                              \
  do {                                                                                                              \
    using L = TailWgradLayout<A_, B_, B_, C_>;                                                                      \
    const size_t wts = (size_t)in * (mid + 4) + (size_t)mid * (out + 4);                                            \
    const size_t a = (size_t)4 * L::WAVE, b = (size_t)L::PART;                                                      \
    const size_t shm = (wts + (a > b ? a : b)) * sizeof(float);                                                     \
    CTR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_bwd_fused_kernel<A_, B_, C_>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                             \
    hipLaunchKernelGGL((tail_bwd_fused_kernel<A_, B_, C_>), dim3(blocks), dim3(256), shm, st, M, Gin, ldgi, in_scalar, \
                       Wb, Hmid, ldhm, scale_mid, Wa, Hout, ldho, scale_out, Gout, ldgo, part);                     \
    hipLaunchKernelGGL(tail_wgrad2_reduce_kernel, dim3((unsigned)cdiv(L::PART, 32)), dim3(256), 0, st, part, blocks, \
                       (int)L::PART, out * mid, mid * in, mid, dWa, dWb, dba, dbb);                                 \
  } while (0)
  if (in == 32) TF(8, 4, 2);
  else TF(4, 2, 1);
#undef TF
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
