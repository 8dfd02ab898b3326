// DIN target attention, the scoring network (DIN/DIN.py:169-185), without the [B*T, 4H] concat:
//   din_all = [q, k, q - k, q*k];  d1 = sigmoid(din_all W1 + b1) [80];  d2 = sigmoid(d1 W2 + b2) [40];
//   score = d2 w3 + b3
// Since din_all W1 = q (W1a + W1c) + k (W1b - W1c) + (q*k) W1d (W1 = [W1a; W1b; W1c; W1d], H rows each), a row
// needs only k, q*k and a per-sample vector.  One wave owns a sample and walks its T positions in strips of
// 16 rows on 16x16x4 fp32 MFMAs, with the accumulators kept transposed (weights are the first operand): lane
// (row l16, quad kq) ends a layer holding h[row][16t + 4kq + r], r = 0..3, which is the operand fragment of
// k-step t of the next layer - the three layers chain through registers.  Weights sit in LDS (re-packed once per
// block).  The backward runs the same chain in reverse (g2, g1, d[q,k,q*k]) and a third kernel forms every
// weight gradient as rows^T x rows products with the accumulators living in registers for the whole launch.
// Sums are taken in a fixed order everywhere: results are bitwise reproducible.
#include "common.h"

namespace ctr {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int DN1 = 80, DN2 = 40;        // widths of f1_att / f2_att (DIN/DIN.py:181-182)
constexpr int DT1 = DN1 / 16;            // 5 accumulator tiles of layer 1
constexpr int DT2 = 3;                   // layer 2 padded to 48 columns (the last 8 are zero weights)
constexpr int DS1 = DN1 + 4;             // LDS row strides: 4 rows apart = 16 banks apart
constexpr int DS2 = 16 * DT2 + 4;

// 1 / (1 + e^-x) on the hardware exp2 and reciprocal: 128 sigmoids per row would otherwise cost more
// vector-instruction time than the row's matrix products.  Both instructions are good to about 1 ulp, but
// exp2(-x log2 e) also inherits the rounding of its argument (|x log2 e| * 2^-24, five ulp at |x| = 5), and the
// weight gradients sum these values over hundreds of thousands of rows with cancellation: the argument's rounding
// residual (one fma) is put back as a first-order factor, and the reciprocal gets one Newton step.  ~1.5 ulp.
__device__ static inline float sigm(float x) {
  const float L2E_HI = 1.44269502f, L2E_LO = 1.92596299e-8f;    // log2(e) = hi + lo
  const float t = -x * L2E_HI;
  float r = fmaf(-x, L2E_HI, -t);                               // exact: what the product's rounding dropped
  r = fmaf(-x, L2E_LO, r);
  float e = __builtin_amdgcn_exp2f(t);
  e = fmaf(e, r * 0.693147181f, e);                             // 2^(t + r) = 2^t (1 + r ln 2 + ...)
  const float d = 1.f + e;
  float y = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, y, 1.f), y, y);
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// LDS image shared by the forward and the backward: Wq = W1a + W1c, Wk = W1b - W1c, Wd = W1d, each [H][DS1];
// W2 as [80][DS2]; b1 [80]; b2, w3 padded to 48 with zeros.
template <int H>
struct FwdLds {
  static constexpr int WQ = 0, WK = H * DS1, WD = 2 * H * DS1, W2 = 3 * H * DS1, B1 = W2 + DN1 * DS2, B2 = B1 + DN1,
                       W3 = B2 + 16 * DT2, END = W3 + 16 * DT2;
};

template <int H>
__device__ static inline void stage_fwd_weights(float* lds, const float* __restrict__ W1, const float* __restrict__ b1,
                                                const float* __restrict__ W2, const float* __restrict__ b2,
                                                const float* __restrict__ w3) {
  using L = FwdLds<H>;
  for (int i = threadIdx.x; i < H * DN1; i += blockDim.x) {
    const int h = i / DN1, n = i % DN1;
    const float a = W1[(int64_t)h * DN1 + n], b = W1[(int64_t)(H + h) * DN1 + n], c = W1[(int64_t)(2 * H + h) * DN1 + n],
                d = W1[(int64_t)(3 * H + h) * DN1 + n];
    lds[L::WQ + h * DS1 + n] = a + c;
    lds[L::WK + h * DS1 + n] = b - c;
    lds[L::WD + h * DS1 + n] = d;
  }
  for (int i = threadIdx.x; i < DN1 * 16 * DT2; i += blockDim.x) {
    const int k = i / (16 * DT2), n = i % (16 * DT2);
    lds[L::W2 + k * DS2 + n] = n < DN2 ? W2[k * DN2 + n] : 0.f;
  }
  for (int i = threadIdx.x; i < DN1; i += blockDim.x) lds[L::B1 + i] = b1[i];
  for (int i = threadIdx.x; i < 16 * DT2; i += blockDim.x) {
    lds[L::B2 + i] = i < DN2 ? b2[i] : 0.f;
    lds[L::W3 + i] = i < DN2 ? w3[i] : 0.f;
  }
}

// rows of sample b that count: its sequence length clamped to [0, T]; all T without lengths
__device__ __forceinline__ int seq_rows(const int32_t* __restrict__ sl, int64_t b, int T) {
  if (!sl) return T;
  const int n = sl[b];
  return n < 0 ? 0 : (n > T ? T : n);
}

template <int HT>
__global__ __launch_bounds__(256, (HT > 4 ? 1 : 2)) void din_mlp_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, int B,
                                                         int T, const float* __restrict__ W1, const float* __restrict__ b1,
                                                         const float* __restrict__ W2, const float* __restrict__ b2,
                                                         const float* __restrict__ w3, const float* __restrict__ b3,
                                                         float* __restrict__ d1, float* __restrict__ d2,
                                                         float* __restrict__ score, const int32_t* __restrict__ sl) {
  constexpr int H = 16 * HT;
  using L = FwdLds<H>;
  extern __shared__ float lds[];
  stage_fwd_weights<H>(lds, W1, b1, W2, b2, w3);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const float bias3 = b3[0];
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
    // positions from the sample's length on are masked out by the softmax that follows (DIN/DIN.py:186-190): their
    // scores are never read, so whole strips of them are not computed (a strip that holds one valid row is)
    const int strips = (seq_rows(sl, b, T) + 15) / 16;
    f32x4 qf[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) qf[j] = *reinterpret_cast<const f32x4*>(q + b * H + 16 * j + 4 * kq);
    // the sample's share of layer 1: q (W1a + W1c) + b1, the same for every row
    f32x4 accq[DT1];
#pragma unroll
    for (int t = 0; t < DT1; ++t) accq[t] = *reinterpret_cast<const f32x4*>(lds + L::B1 + 16 * t + 4 * kq);
#pragma unroll
    for (int j = 0; j < HT; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* w = lds + L::WQ + (16 * j + 4 * kq + u) * DS1 + l16;
#pragma unroll
        for (int t = 0; t < DT1; ++t) accq[t] = MFMA4(w[16 * t], qf[j][u], accq[t]);
      }
    f32x4 kn[HT];                                   // the next strip's rows of k, in flight while this one multiplies
#pragma unroll
    for (int j = 0; j < HT; ++j)
      kn[j] = *reinterpret_cast<const f32x4*>(k + (b * T + (l16 < T ? l16 : T - 1)) * H + 16 * j + 4 * kq);
    for (int s = 0; s < strips; ++s) {
      const int row = 16 * s + l16;
      const bool live = row < T;
      const int64_t m = b * T + (live ? row : T - 1);
      f32x4 kf[HT];
#pragma unroll
      for (int j = 0; j < HT; ++j) kf[j] = kn[j];
      if (s + 1 < strips) {
        const int64_t mn = b * T + (row + 16 < T ? row + 16 : T - 1);
#pragma unroll
        for (int j = 0; j < HT; ++j) kn[j] = *reinterpret_cast<const f32x4*>(k + mn * H + 16 * j + 4 * kq);
      }
      f32x4 a1[DT1];
#pragma unroll
      for (int t = 0; t < DT1; ++t) a1[t] = accq[t];
#pragma unroll
      for (int j = 0; j < HT; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* wk = lds + L::WK + (16 * j + 4 * kq + u) * DS1 + l16;
          const float* wd = lds + L::WD + (16 * j + 4 * kq + u) * DS1 + l16;
          const float xk = kf[j][u], xd = kf[j][u] * qf[j][u];
#pragma unroll
          for (int t = 0; t < DT1; ++t) {
            a1[t] = MFMA4(wk[16 * t], xk, a1[t]);
            a1[t] = MFMA4(wd[16 * t], xd, a1[t]);
          }
          asm volatile("" ::: "memory");   // LDS weights are re-read per use, not parked in registers
        }
#pragma unroll
      for (int t = 0; t < DT1; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[t][r] = sigm(a1[t][r]);
        if (d1 && live) *reinterpret_cast<f32x4*>(d1 + m * DN1 + 16 * t + 4 * kq) = a1[t];
      }
      f32x4 a2[DT2];
#pragma unroll
      for (int t = 0; t < DT2; ++t) a2[t] = *reinterpret_cast<const f32x4*>(lds + L::B2 + 16 * t + 4 * kq);
#pragma unroll
      for (int j = 0; j < DT1; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* w = lds + L::W2 + (16 * j + 4 * kq + u) * DS2 + l16;
#pragma unroll
          for (int t = 0; t < DT2; ++t) a2[t] = MFMA4(w[16 * t], a1[j][u], a2[t]);
          asm volatile("" ::: "memory");   // LDS weights are re-read per use, not parked in registers
        }
      float p = 0.f;
#pragma unroll
      for (int t = 0; t < DT2; ++t) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + L::W3 + 16 * t + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a2[t][r] = sigm(a2[t][r]);
          p += a2[t][r] * w[r];                       // padded columns carry zero weights
        }
        if (d2 && live && 16 * t + 4 * kq < DN2) *reinterpret_cast<f32x4*>(d2 + m * DN2 + 16 * t + 4 * kq) = a2[t];
      }
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (live && kq == 0) score[m] = p + bias3;
    }
  }
}

// Backward of the chain with respect to its inputs.  Per row: g2 = ds w3 * d2 (1 - d2);  g1 = (g2 W2^T) * d1 (1 - d1);
// dx = g1 [Wq; Wk; Wd]^T (3H columns);  dk = dx_k + dx_d * q (added to what dk holds when acc_k);
// dq = sum_rows (dx_q + dx_d * k).  g1 and g2 are written for the weight-gradient kernel.
// LDS: W2t [48][DS1] (W2t[n2][n1] = W2[n1][n2], rows >= 40 zero), Wt [80][3H + 4] (Wt[n1][c] = [Wq; Wk; Wd][c][n1]),
// w3 padded to 48.
template <int HT>
__global__ __launch_bounds__(256, (HT > 4 ? 1 : 2)) void din_mlp_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, int B,
                                                         int T, const float* __restrict__ W1, const float* __restrict__ W2,
                                                         const float* __restrict__ w3, const float* __restrict__ d1,
                                                         const float* __restrict__ d2, const float* __restrict__ ds,
                                                         float* __restrict__ g1, float* __restrict__ g2,
                                                         float* __restrict__ dq, float* __restrict__ dk, int acc_k,
                                                         const int32_t* __restrict__ sl) {
  constexpr int H = 16 * HT, XT = 3 * HT, SX = 3 * H + 4;
  extern __shared__ float lds[];
  float* W2t = lds;                         // [48][DS1]
  float* Wt = W2t + 16 * DT2 * DS1;         // [80][SX]
  float* w3s = Wt + DN1 * SX;               // [48]
  for (int i = threadIdx.x; i < 16 * DT2 * DN1; i += blockDim.x) {
    const int n2 = i / DN1, n1 = i % DN1;
    W2t[n2 * DS1 + n1] = n2 < DN2 ? W2[n1 * DN2 + n2] : 0.f;
  }
  for (int i = threadIdx.x; i < H * DN1; i += blockDim.x) {
    const int h = i / DN1, n = i % DN1;
    const float a = W1[(int64_t)h * DN1 + n], b = W1[(int64_t)(H + h) * DN1 + n], c = W1[(int64_t)(2 * H + h) * DN1 + n],
                d = W1[(int64_t)(3 * H + h) * DN1 + n];
    Wt[n * SX + h] = a + c;
    Wt[n * SX + H + h] = b - c;
    Wt[n * SX + 2 * H + h] = d;
  }
  for (int i = threadIdx.x; i < 16 * DT2; i += blockDim.x) w3s[i] = i < DN2 ? w3[i] : 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
    // masked positions have dscore = 0, hence zero g1, g2, dq and dk contributions: their strips are skipped (dk keeps
    // what it holds there, g1 / g2 stay unwritten and are not read by the weight-gradient pass either)
    const int strips = (seq_rows(sl, b, T) + 15) / 16;
    f32x4 qf[HT], dqa[HT];
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      qf[j] = *reinterpret_cast<const f32x4*>(q + b * H + 16 * j + 4 * kq);
      dqa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // a strip's operands (k, d1, d2, ds of its rows) are fetched one strip ahead
    struct In { f32x4 k[HT], y1[DT1], y2[DT2]; float ds; };
    auto fetch = [&](int s, In& o) {
      const int row = 16 * s + l16;
      const bool live = row < T;
      const int64_t m = b * T + (live ? row : T - 1);
#pragma unroll
      for (int j = 0; j < HT; ++j) o.k[j] = *reinterpret_cast<const f32x4*>(k + m * H + 16 * j + 4 * kq);
#pragma unroll
      for (int t = 0; t < DT1; ++t) o.y1[t] = *reinterpret_cast<const f32x4*>(d1 + m * DN1 + 16 * t + 4 * kq);
#pragma unroll
      for (int t = 0; t < DT2; ++t)
        o.y2[t] = 16 * t + 4 * kq < DN2 ? *reinterpret_cast<const f32x4*>(d2 + m * DN2 + 16 * t + 4 * kq)
                                        : f32x4{0.f, 0.f, 0.f, 0.f};
      o.ds = live ? ds[m] : 0.f;
    };
    In nxt;
    fetch(0, nxt);
    for (int s = 0; s < strips; ++s) {
      const int row = 16 * s + l16;
      const bool live = row < T;
      const int64_t m = b * T + (live ? row : T - 1);
      const In cur = nxt;
      if (s + 1 < strips) fetch(s + 1, nxt);
      const float dsv = cur.ds;
      f32x4 kf[HT];
#pragma unroll
      for (int j = 0; j < HT; ++j) kf[j] = cur.k[j];
      f32x4 gg2[DT2];
#pragma unroll
      for (int t = 0; t < DT2; ++t) {
        const bool in = 16 * t + 4 * kq < DN2;
        const f32x4 y = cur.y2[t];
        const f32x4 w = *reinterpret_cast<const f32x4*>(w3s + 16 * t + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) gg2[t][r] = dsv * w[r] * (y[r] * (1.f - y[r]));
        if (in && live) *reinterpret_cast<f32x4*>(g2 + m * DN2 + 16 * t + 4 * kq) = gg2[t];
      }
      f32x4 gg1[DT1];
#pragma unroll
      for (int t = 0; t < DT1; ++t) gg1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < DT2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* w = W2t + (16 * j + 4 * kq + u) * DS1 + l16;
#pragma unroll
          for (int t = 0; t < DT1; ++t) gg1[t] = MFMA4(w[16 * t], gg2[j][u], gg1[t]);
          asm volatile("" ::: "memory");   // LDS weights are re-read per use, not parked in registers
        }
#pragma unroll
      for (int t = 0; t < DT1; ++t) {
        const f32x4 y = cur.y1[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) gg1[t][r] *= y[r] * (1.f - y[r]);
        if (live) *reinterpret_cast<f32x4*>(g1 + m * DN1 + 16 * t + 4 * kq) = gg1[t];
      }
      f32x4 dx[XT];
#pragma unroll
      for (int t = 0; t < XT; ++t) dx[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < DT1; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* w = Wt + (16 * j + 4 * kq + u) * SX + l16;
#pragma unroll
          for (int t = 0; t < XT; ++t) dx[t] = MFMA4(w[16 * t], gg1[j][u], dx[t]);
          asm volatile("" ::: "memory");   // LDS weights are re-read per use, not parked in registers
        }
#pragma unroll
      for (int j = 0; j < HT; ++j) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = dx[HT + j][r] + dx[2 * HT + j][r] * qf[j][r];
          dqa[j][r] += dx[j][r] + dx[2 * HT + j][r] * kf[j][r];      // rows past T carry zero gradients
        }
        if (live) {
          float* p = dk + m * H + 16 * j + 4 * kq;
          if (acc_k) {
            const f32x4 old = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += old[r];
          }
          *reinterpret_cast<f32x4*>(p) = o;
        }
      }
    }
    // dq: the sample's rows sit in the 16 lanes of a data-parallel-path row
#pragma unroll
    for (int j = 0; j < HT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = dqa[j][r];
        v = dpp_add<DPP_XOR1>(v);
        v = dpp_add<DPP_XOR2>(v);
        v = dpp_add<DPP_ROR + 4>(v);
        v = dpp_add<DPP_ROR + 8>(v);
        dqa[j][r] = v;
      }
      if (l16 == 0) *reinterpret_cast<f32x4*>(dq + b * H + 16 * j + 4 * kq) = dqa[j];
    }
  }
}

// Weight gradients of the chain, all of them in one launch.  A wave walks samples and keeps, for the whole
// launch, the accumulator tiles of  dW2 = d1^T g2 [80 x 40 (48)]  and  dW' = [q, k, q*k]^T g1 [3H x 80], plus
// per-column running sums for db1 = sum g1, db2 = sum g2, dw3 = sum ds d2, db3 = sum ds.  Both operands of a
// product are rows^T: lane (l16, kq) needs column 16i + l16 of row 4kq + u, read from a 16-row strip that the
// wave staged in its own LDS area with whole-row 16-byte loads.  Every wave writes its partial to `part`;
// din_mlp_wgrad_reduce_kernel adds the partials in a fixed order and _final unfolds dW' into dW1:
//   dW1 = [dWq; dWk; dWq - dWk; dWd].
// The 3H/16 x 5 tiles of dW' fit a wave's registers up to H = 32.  Wider rows are cut into XS column slices of 32
// (blockIdx.y): a slice's waves walk all samples but keep only the tiles of their 32 columns of q, k and q*k (and
// read only those columns of k); slice 0 also carries dW2 and the column sums.
template <int HT, int XS>
struct WgradLayout {
  static constexpr int H = 16 * HT, HTS = HT / XS, XT = 3 * HTS;
  static constexpr int W2 = 0;                              // [80][48]                     (slice 0)
  static constexpr int WX = W2 + DN1 * 16 * DT2;            // [3][16 * HTS][80]: the slice's rows of dWq, dWk, dWd
  static constexpr int SB1 = WX + 3 * 16 * HTS * DN1;       // [80]                         (slice 0, like the rest)
  static constexpr int SB2 = SB1 + DN1;                     // [48]
  static constexpr int SW3 = SB2 + 16 * DT2;                // [48]
  static constexpr int SB3 = SW3 + 16 * DT2;                // [1] (+3 pad)
  static constexpr int PART = SB3 + 4;
  // where row c of [dWq; dWk; dWd] ([3H][80]) sits in the totals [XS][PART]
  __host__ __device__ static constexpr int wx_at(int c, int n) {
    return ((c % H) / 16 / HTS) * PART + WX + (((c / H) * HTS + ((c % H) / 16) % HTS) * 16 + (c % 16)) * DN1 + n;
  }
};

template <int HTS>
struct WgradLds {                       // one wave's staging area: a strip of 16 rows of every operand
  static constexpr int HS = 16 * HTS;   // the slice's columns of k
  static constexpr int S1 = DN1 + 4, S2 = DN2 + 4, SK = HS + 4;    // strides: rows 4 apart sit 16 banks apart
  static constexpr int D1 = 0, G1 = 16 * S1, D2 = 2 * 16 * S1, G2 = D2 + 16 * S2, KK = G2 + 16 * S2, DS = KK + 16 * SK,
                       END = DS + 16;
};

template <int HT, int XS>
__global__ __launch_bounds__(256, 1) void din_mlp_wgrad_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              int B, int T, const float* __restrict__ d1,
                                                              const float* __restrict__ d2, const float* __restrict__ ds,
                                                              const float* __restrict__ g1, const float* __restrict__ g2,
                                                              float* __restrict__ part, const int32_t* __restrict__ sl) {
  using WL = WgradLayout<HT, XS>;
  constexpr int HTS = WL::HTS, XT = WL::XT, HS = 16 * HTS;
  using SL = WgradLds<HTS>;
  constexpr int H = 16 * HT;
  constexpr int F1 = 16 * DN1 / 4, F2 = 16 * DN2 / 4, FK = 16 * HS / 4;   // float4s of a strip of d1 / d2 / k's slice
  constexpr int N1 = (F1 + 63) / 64, N2 = (F2 + 63) / 64, NK = (FK + 63) / 64;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const int slice = blockIdx.y;
  const bool lead = slice == 0;                                  // carries dW2 and the column sums
  const int c0 = slice * HS;                                     // first column of q / k of this slice
  float* my = lds + wave * SL::END;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  f32x4 aw2[DT1][DT2], awx[XT][DT1];
#pragma unroll
  for (int i = 0; i < DT1; ++i)
#pragma unroll
    for (int t = 0; t < DT2; ++t) aw2[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < XT; ++i)
#pragma unroll
    for (int t = 0; t < DT1; ++t) awx[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sb1[DT1], sb2[DT2], sw3[DT2], sb3 = 0.f;
#pragma unroll
  for (int t = 0; t < DT1; ++t) sb1[t] = 0.f;
#pragma unroll
  for (int t = 0; t < DT2; ++t) sb2[t] = sw3[t] = 0.f;
  // A strip's operands are contiguous in memory (rows r0 .. r0+15 of one sample): whole 16-byte pieces per lane
  // into registers (in flight while the previous strip multiplies), then into the wave's LDS area, from where
  // the products read them in MFMA order.  Rows past T are zeros.
  struct Regs { f32x4 d1[N1], g1[N1], d2[N2], g2[N2], k[NK]; float ds; };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load_strip = [&](int64_t b, int s, Regs& r) {
    const int r0 = 16 * s;
    const int nrows = T - r0 < 16 ? T - r0 : 16;
    const int64_t m0 = b * T + r0;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int f = lane + 64 * i;
      const bool ok = f < nrows * (DN1 / 4);
      r.d1[i] = ok && lead ? *reinterpret_cast<const f32x4*>(d1 + m0 * DN1 + 4 * f) : zero4;
      r.g1[i] = ok ? *reinterpret_cast<const f32x4*>(g1 + m0 * DN1 + 4 * f) : zero4;
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int f = lane + 64 * i;
      const bool ok = lead && f < nrows * (DN2 / 4);
      r.d2[i] = ok ? *reinterpret_cast<const f32x4*>(d2 + m0 * DN2 + 4 * f) : zero4;
      r.g2[i] = ok ? *reinterpret_cast<const f32x4*>(g2 + m0 * DN2 + 4 * f) : zero4;
    }
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int f = lane + 64 * i;                               // row f / (HS/4), float4 f % (HS/4) of the slice
      r.k[i] = f < nrows * (HS / 4) ? *reinterpret_cast<const f32x4*>(k + (m0 + f / (HS / 4)) * H + c0 + 4 * (f % (HS / 4)))
                                    : zero4;
    }
    r.ds = lead && lane < nrows ? ds[m0 + lane] : 0.f;
  };
  auto store_strip = [&](const Regs& r) {
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int f = lane + 64 * i;
      if (F1 % 64 == 0 || f < F1) {
        *reinterpret_cast<f32x4*>(my + SL::D1 + (f / (DN1 / 4)) * SL::S1 + 4 * (f % (DN1 / 4))) = r.d1[i];
        *reinterpret_cast<f32x4*>(my + SL::G1 + (f / (DN1 / 4)) * SL::S1 + 4 * (f % (DN1 / 4))) = r.g1[i];
      }
    }
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      const int f = lane + 64 * i;
      if (F2 % 64 == 0 || f < F2) {
        *reinterpret_cast<f32x4*>(my + SL::D2 + (f / (DN2 / 4)) * SL::S2 + 4 * (f % (DN2 / 4))) = r.d2[i];
        *reinterpret_cast<f32x4*>(my + SL::G2 + (f / (DN2 / 4)) * SL::S2 + 4 * (f % (DN2 / 4))) = r.g2[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int f = lane + 64 * i;
      if (FK % 64 == 0 || f < FK) *reinterpret_cast<f32x4*>(my + SL::KK + (f / (HS / 4)) * SL::SK + 4 * (f % (HS / 4))) = r.k[i];
    }
    if (lane < 16) my[SL::DS + lane] = r.ds;
  };
  // the wave's samples gw, gw + nw, ... and of each the strips that hold a valid row (masked positions carry zero
  // gradients: g1, g2 and dscore of their strips are neither written nor read)
  auto advance = [&](int64_t& b, int& s) {                   // the strip after (b, s); b >= B when there is none
    ++s;
    while (b < B && 16 * s >= seq_rows(sl, b, T)) {
      b += nw;
      s = 0;
    }
  };
  int64_t b = gw, bn;
  int s = -1, sn;
  advance(b, s);
  Regs regs;
  if (b < B) load_strip(b, s, regs);
  float qv[HTS];
  int64_t bq = -1;
  for (; b < B; b = bn, s = sn) {
    if (b != bq) {
      bq = b;
#pragma unroll
      for (int j = 0; j < HTS; ++j) qv[j] = q[b * H + c0 + 16 * j + l16];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the last strip's reads are done before it is overwritten
    __builtin_amdgcn_wave_barrier();
    store_strip(regs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bn = b;
    sn = s;
    advance(bn, sn);
    if (bn < B) load_strip(bn, sn, regs);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * kq + u;                              // a product step takes rows u, 4+u, 8+u, 12+u
      float g1v[DT1], kv[HTS];
#pragma unroll
      for (int t = 0; t < DT1; ++t) g1v[t] = my[SL::G1 + row * SL::S1 + 16 * t + l16];
#pragma unroll
      for (int j = 0; j < HTS; ++j) kv[j] = my[SL::KK + row * SL::SK + 16 * j + l16];
#pragma unroll
      for (int j = 0; j < HTS; ++j)
#pragma unroll
        for (int t = 0; t < DT1; ++t) {
          awx[j][t] = MFMA4(qv[j], g1v[t], awx[j][t]);
          awx[HTS + j][t] = MFMA4(kv[j], g1v[t], awx[HTS + j][t]);
          awx[2 * HTS + j][t] = MFMA4(qv[j] * kv[j], g1v[t], awx[2 * HTS + j][t]);
        }
      if (lead) {
        float d1v[DT1], g2v[DT2], d2v[DT2];
#pragma unroll
        for (int t = 0; t < DT1; ++t) d1v[t] = my[SL::D1 + row * SL::S1 + 16 * t + l16];
#pragma unroll
        for (int t = 0; t < DT2; ++t) {
          const bool in = 16 * t + l16 < DN2;
          g2v[t] = in ? my[SL::G2 + row * SL::S2 + 16 * t + l16] : 0.f;
          d2v[t] = in ? my[SL::D2 + row * SL::S2 + 16 * t + l16] : 0.f;
        }
        const float dsv = my[SL::DS + row];
#pragma unroll
        for (int i = 0; i < DT1; ++i)
#pragma unroll
          for (int t = 0; t < DT2; ++t) aw2[i][t] = MFMA4(d1v[i], g2v[t], aw2[i][t]);
#pragma unroll
        for (int t = 0; t < DT1; ++t) sb1[t] += g1v[t];
#pragma unroll
        for (int t = 0; t < DT2; ++t) {
          sb2[t] += g2v[t];
          sw3[t] += dsv * d2v[t];
        }
        sb3 += dsv;
      }
    }
  }
  // the per-column sums: lanes kq = 0..3 hold different rows of the same column
#pragma unroll
  for (int t = 0; t < DT1; ++t) { sb1[t] += __shfl_xor(sb1[t], 16); sb1[t] += __shfl_xor(sb1[t], 32); }
#pragma unroll
  for (int t = 0; t < DT2; ++t) {
    sb2[t] += __shfl_xor(sb2[t], 16); sb2[t] += __shfl_xor(sb2[t], 32);
    sw3[t] += __shfl_xor(sw3[t], 16); sw3[t] += __shfl_xor(sw3[t], 32);
  }
  sb3 += __shfl_xor(sb3, 16); sb3 += __shfl_xor(sb3, 32);   // every l16 lane summed the same rows
  float* P = part + ((int64_t)slice * nw + gw) * WL::PART;     // [slice][wave][PART]
  // tile (i, t): lane (l16, kq) holds D[16i + 4kq + r][16t + l16]
#pragma unroll
  for (int i = 0; i < XT; ++i)
#pragma unroll
    for (int t = 0; t < DT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[WL::WX + (16 * i + 4 * kq + r) * DN1 + 16 * t + l16] = awx[i][t][r];
  if (!lead) return;                                           // the other slices' totals are read in WX only
#pragma unroll
  for (int i = 0; i < DT1; ++i)
#pragma unroll
    for (int t = 0; t < DT2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[WL::W2 + (16 * i + 4 * kq + r) * (16 * DT2) + 16 * t + l16] = aw2[i][t][r];
  if (kq == 0) {
#pragma unroll
    for (int t = 0; t < DT1; ++t) P[WL::SB1 + 16 * t + l16] = sb1[t];
#pragma unroll
    for (int t = 0; t < DT2; ++t) {
      P[WL::SB2 + 16 * t + l16] = sb2[t];
      P[WL::SW3 + 16 * t + l16] = sw3[t];
    }
    if (l16 == 0) P[WL::SB3] = sb3;
  }
}

// Partial sums -> totals, in a fixed order: thread (group pg, element e) adds the partials pg, pg + 8, ... (eight
// independent loads in flight), the eight group sums are then added in group order.  The sums run in double
// precision: these gradients are sums over B*T rows of terms of both signs, and a thousand fp32 partials added in
// fp32 cost more accuracy (1e-5 of the largest gradient at 60 000 rows) than all the arithmetic before them.
// blockIdx.y = column slice: its partials are part[slice][nparts][PART], its totals tot[slice][PART]; of the slices
// after the first only elements [wx_lo, wx_hi) (the dW' tiles) were written and are summed.
__global__ __launch_bounds__(256) void din_mlp_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, int PART,
                                                                  int wx_lo, int wx_hi, float* __restrict__ tot) {
  __shared__ double sm[8][32];
  const int pg = threadIdx.x >> 5, e = blockIdx.x * 32 + (threadIdx.x & 31);
  part += (int64_t)blockIdx.y * nparts * PART;
  tot += (int64_t)blockIdx.y * PART;
  if (blockIdx.y > 0 && (blockIdx.x * 32 + 31 < wx_lo || blockIdx.x * 32 >= wx_hi)) return;   // whole block outside
  const bool mine = e < PART && (blockIdx.y == 0 || (e >= wx_lo && e < wx_hi));
  double s = 0.0;
  if (mine) {
    int p = pg;
    for (; p + 56 < nparts; p += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + 8 * u) * PART + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; p < nparts; p += 8) s += (double)part[(int64_t)p * PART + e];
  }
  sm[pg][threadIdx.x & 31] = s;
  __syncthreads();
  if (pg == 0 && mine) {
    double t = sm[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += sm[g][threadIdx.x];
    tot[e] = (float)t;
  }
}

// Totals -> parameter gradients: dW2 un-padded, dW' = [dWq; dWk; dWd] unfolded into dW1 = [dWq; dWk; dWq - dWk; dWd].
// One thread per element of the un-sliced order [dW2 (80 x 48) | dW' (3H x 80) | db1 | db2 | dw3 | db3].
template <int HT, int XS>
__global__ __launch_bounds__(256) void din_mlp_wgrad_final_kernel(const float* __restrict__ tot, float* __restrict__ dW1,
                                                                 float* __restrict__ db1, float* __restrict__ dW2,
                                                                 float* __restrict__ db2, float* __restrict__ dw3,
                                                                 float* __restrict__ db3) {
  using WL = WgradLayout<HT, XS>;
  constexpr int H = 16 * HT;
  constexpr int NW2 = DN1 * 16 * DT2, NWX = 3 * H * DN1;
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e < NW2) {
    const int n1 = e / (16 * DT2), n2 = e % (16 * DT2);
    if (n2 < DN2) dW2[n1 * DN2 + n2] = tot[WL::W2 + e];
    return;
  }
  e -= NW2;
  if (e < NWX) {
    const int c = e / DN1, n = e % DN1;                           // row c of [dWq; dWk; dWd]
    const float s = tot[WL::wx_at(c, n)];
    if (c < H) {                                                  // dWq -> W1a, and the plus half of W1c
      dW1[(int64_t)c * DN1 + n] = s;
      dW1[(int64_t)(2 * H + c) * DN1 + n] = s - tot[WL::wx_at(H + c, n)];
    } else if (c < 2 * H) {
      dW1[(int64_t)c * DN1 + n] = s;                              // dWk -> W1b
    } else {
      dW1[(int64_t)(c + H) * DN1 + n] = s;                        // dWd -> W1d
    }
    return;
  }
  e -= NWX;
  if (e < DN1) db1[e] = tot[WL::SB1 + e];
  else if (e < DN1 + DN2) db2[e - DN1] = tot[WL::SB2 + e - DN1];
  else if (e < DN1 + 2 * DN2) dw3[e - DN1 - DN2] = tot[WL::SW3 + e - DN1 - DN2];
  else if (e == DN1 + 2 * DN2) db3[0] = tot[WL::SB3];
}

}  // namespace ctr

using namespace ctr;

static bool din_h_ok(int H) { return H == 16 || H == 32 || H == 64 || H == 128; }

extern "C" int ctr_din_attn_mlp_supported(int H, int n1, int n2) { return din_h_ok(H) && n1 == DN1 && n2 == DN2; }

static int din_grid(int B, int per_cu) {
  const int64_t need = cdiv(B, 4);
  const int64_t cap = 256 * per_cu;
  return (int)(need < cap ? need : cap);
}
// resident blocks per CU of the forward / backward kernels: their weight image in LDS is 33 KB at H = 16, 49 KB at
// 32, 81 KB at 64 and 146 KB at 128 (a block re-stages it, so no more blocks than stay resident)
static int din_blocks_per_cu(int H) { return H <= 32 ? 3 : 1; }

extern "C" int ctr_din_attn_mlp_fwd_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H,
                                       const float* W1, const float* b1, const float* W2, const float* b2,
                                       const float* w3, const float* b3, float* d1_out, float* d2_out,
                                       float* score_out, void* stream) {
  CTR_REQUIRE(B >= 0 && T > 0 && din_h_ok(H), "ctr_din_attn_mlp_fwd: H=%d must be 16, 32, 64 or 128 (T > 0)", H);
  CTR_REQUIRE(q && k && W1 && b1 && W2 && b2 && w3 && b3 && score_out, "ctr_din_attn_mlp_fwd: null pointer");
  CTR_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)d1_out % 16) == 0 &&
                  ((uintptr_t)d2_out % 16) == 0,
              "ctr_din_attn_mlp_fwd: q, k, d1_out and d2_out must be 16-byte aligned");
  if (B == 0) return CTR_OK;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HT)                                                                                               \
  do {                                                                                                           \
    const size_t shm = (size_t)FwdLds<16 * HT>::END * sizeof(float);                                             \
    if (shm > 48 * 1024)                                                                                         \
      CTR_HIP(hipFuncSetAttribute((const void*)din_mlp_fwd_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)shm));                                                                    \
    hipLaunchKernelGGL(din_mlp_fwd_kernel<HT>, dim3(din_grid(B, din_blocks_per_cu(H))), dim3(256), shm, st, q, k, B, T, W1, b1, W2, \
                       b2, w3, b3, d1_out, d2_out, score_out, sl);                                               \
  } while (0)
  if (H == 16) LAUNCH(1);
  else if (H == 32) LAUNCH(2);
  else if (H == 64) LAUNCH(4);
  else LAUNCH(8);
#undef LAUNCH
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_attn_mlp_fwd(const float* q, const float* k, int B, int T, int H, const float* W1,
                                    const float* b1, const float* W2, const float* b2, const float* w3,
                                    const float* b3, float* d1_out, float* d2_out, float* score_out, void* stream) {
  return ctr_din_attn_mlp_fwd_sl(q, k, nullptr, B, T, H, W1, b1, W2, b2, w3, b3, d1_out, d2_out, score_out, stream);
}

extern "C" int ctr_din_attn_mlp_bwd_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H,
                                       const float* W1, const float* W2, const float* w3, const float* d1,
                                       const float* d2, const float* dscore, float* g1_out, float* g2_out,
                                       float* dq_out, float* dk, int accumulate_dk, void* stream) {
  CTR_REQUIRE(B >= 0 && T > 0 && din_h_ok(H), "ctr_din_attn_mlp_bwd: H=%d must be 16, 32, 64 or 128 (T > 0)", H);
  CTR_REQUIRE(q && k && W1 && W2 && w3 && d1 && d2 && dscore && g1_out && g2_out && dq_out && dk,
              "ctr_din_attn_mlp_bwd: null pointer");
  CTR_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)d1 % 16) == 0 &&
                  ((uintptr_t)d2 % 16) == 0 && ((uintptr_t)g1_out % 16) == 0 && ((uintptr_t)g2_out % 16) == 0 &&
                  ((uintptr_t)dq_out % 16) == 0 && ((uintptr_t)dk % 16) == 0,
              "ctr_din_attn_mlp_bwd: buffers must be 16-byte aligned");
  if (B == 0) return CTR_OK;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HT)                                                                                               \
  do {                                                                                                           \
    const size_t shm = (size_t)(16 * DT2 * DS1 + DN1 * (3 * 16 * HT + 4) + 16 * DT2) * sizeof(float);            \
    if (shm > 48 * 1024)                                                                                         \
      CTR_HIP(hipFuncSetAttribute((const void*)din_mlp_bwd_kernel<HT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)shm));                                                                    \
    hipLaunchKernelGGL(din_mlp_bwd_kernel<HT>, dim3(din_grid(B, din_blocks_per_cu(H))), dim3(256), shm, st, q, k, B, T, W1, W2, w3, \
                       d1, d2, dscore, g1_out, g2_out, dq_out, dk, accumulate_dk, sl);                           \
  } while (0)
  if (H == 16) LAUNCH(1);
  else if (H == 32) LAUNCH(2);
  else if (H == 64) LAUNCH(4);
  else LAUNCH(8);
#undef LAUNCH
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_attn_mlp_bwd(const float* q, const float* k, int B, int T, int H, const float* W1,
                                    const float* W2, const float* w3, const float* d1, const float* d2,
                                    const float* dscore, float* g1_out, float* g2_out, float* dq_out, float* dk,
                                    int accumulate_dk, void* stream) {
  return ctr_din_attn_mlp_bwd_sl(q, k, nullptr, B, T, H, W1, W2, w3, d1, d2, dscore, g1_out, g2_out, dq_out, dk,
                                 accumulate_dk, stream);
}

static int wgrad_parts(int B) { return din_grid(B, 1) * 4; }
static int wgrad_slices(int H) { return H <= 32 ? 1 : H / 32; }
static size_t wgrad_part_floats(int H) {
  return H == 16 ? WgradLayout<1, 1>::PART : WgradLayout<2, 1>::PART;      // a slice of a wider row is 32 columns
}

extern "C" size_t ctr_din_attn_mlp_wgrad_workspace_bytes(int B, int H) {
  if (!din_h_ok(H) || B <= 0) return 0;
  // [slices][waves] partials and [slices] totals
  return (size_t)wgrad_slices(H) * (wgrad_parts(B) + 1) * wgrad_part_floats(H) * sizeof(float);
}

extern "C" int ctr_din_attn_mlp_wgrad_sl(const float* q, const float* k, const int32_t* sl, int B, int T, int H,
                                         const float* d1, const float* d2, const float* dscore, const float* g1,
                                         const float* g2, float* dW1, float* db1, float* dW2, float* db2, float* dw3,
                                         float* db3, void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(B > 0 && T > 0 && din_h_ok(H), "ctr_din_attn_mlp_wgrad: H=%d must be 16, 32, 64 or 128 (B, T > 0)", H);
  CTR_REQUIRE(q && k && d1 && d2 && dscore && g1 && g2 && dW1 && db1 && dW2 && db2 && dw3 && db3 && workspace,
              "ctr_din_attn_mlp_wgrad: null pointer");
  const size_t need = ctr_din_attn_mlp_wgrad_workspace_bytes(B, H);
  if (workspace_bytes < need) {
    set_error("ctr_din_attn_mlp_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int blocks = din_grid(B, 1), parts = blocks * 4;
  float* part = (float*)workspace;
#define LAUNCH(HT, XS)                                                                                              \
  do {                                                                                                              \
    using WL = WgradLayout<HT, XS>;                                                                                 \
    const size_t shm = (size_t)4 * WgradLds<HT / XS>::END * sizeof(float);                                          \
    if (shm > 48 * 1024)                                                                                            \
      CTR_HIP(hipFuncSetAttribute((const void*)din_mlp_wgrad_kernel<HT, XS>,                                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                           \
    hipLaunchKernelGGL((din_mlp_wgrad_kernel<HT, XS>), dim3(blocks, XS), dim3(256), shm, st, q, k, B, T, d1, d2,    \
                       dscore, g1, g2, part, sl);                                                                   \
    float* tot = part + (size_t)XS * parts * WL::PART;                                                              \
    hipLaunchKernelGGL(din_mlp_wgrad_reduce_kernel, dim3((unsigned)cdiv(WL::PART, 32), XS), dim3(256), 0, st, part, \
                       parts, (int)WL::PART, (int)WL::WX, (int)WL::SB1, tot);                                       \
    constexpr int NOUT = DN1 * 16 * DT2 + 3 * 16 * HT * DN1 + DN1 + 2 * DN2 + 1;                                    \
    hipLaunchKernelGGL((din_mlp_wgrad_final_kernel<HT, XS>), dim3((unsigned)cdiv(NOUT, 256)), dim3(256), 0, st,     \
                       tot, dW1, db1, dW2, db2, dw3, db3);                                                          \
  } while (0)
  if (H == 16) LAUNCH(1, 1);
  else if (H == 32) LAUNCH(2, 1);
  else if (H == 64) LAUNCH(4, 2);
  else LAUNCH(8, 4);
#undef LAUNCH
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_attn_mlp_wgrad(const float* q, const float* k, int B, int T, int H, const float* d1,
                                      const float* d2, const float* dscore, const float* g1, const float* g2,
                                      float* dW1, float* db1, float* dW2, float* db2, float* dw3, float* db3,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return ctr_din_attn_mlp_wgrad_sl(q, k, nullptr, B, T, H, d1, d2, dscore, g1, g2, dW1, db1, dW2, db2, dw3, db3,
                                   workspace, workspace_bytes, stream);
}
