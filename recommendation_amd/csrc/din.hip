// DIN pieces that are not GEMMs (DIN/DIN.py:17-71,149-196): the double-indirection id builder,
// the [q, k, q-k, q*k] attention input and its backward, the masked/scaled softmax pooling over the
// behaviour sequence and its backward, inference-mode batch normalisation and Dice.
#include "common.h"

namespace ctr {

// out[k] = (id, cate_list[id]) — the item id and its category id (DIN/DIN.py:24-27,38-41).
__global__ __launch_bounds__(256) void din_build_idx_kernel(const int32_t* __restrict__ ids,
                                                           const int32_t* __restrict__ cate_list, int64_t n,
                                                           int item_count, int32_t* __restrict__ out) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const int id = ids[k];
    const bool ok = id >= 0 && id < item_count;
    out[2 * k] = ok ? id : -1;
    out[2 * k + 1] = ok ? cate_list[id] : -1;
  }
}

// fixed-order block reductions over 256 threads (4 waves): wave result, then the four in order
__device__ static inline float block_sum256(float v, float* red4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}
__device__ static inline float block_max256(float v, float* red4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
}

// row (b,t) of out = [q_b, k_bt, q_b - k_bt, q_b * k_bt]           (DIN/DIN.py:171-174)
__global__ __launch_bounds__(256) void din_concat_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        int B, int T, int H, float* __restrict__ out) {
  const int64_t total = (int64_t)B * T * H;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(p % H);
    const int64_t bt = p / H;
    const int64_t b = bt / T;
    const float qv = q[b * H + h], kv = k[p];
    float* o = out + bt * 4 * H;
    o[h] = qv; o[H + h] = kv; o[2 * H + h] = qv - kv; o[3 * H + h] = qv * kv;
  }
}

// dk[b,t,h] (+)= d1 - d2 + d3*q ;  dq[b,h] (+)= sum_t (d0 + d2 + d3*k).  One wave per sample.
__global__ __launch_bounds__(256) void din_concat_bwd_kernel(const float* __restrict__ d_all,
                                                            const float* __restrict__ q,
                                                            const float* __restrict__ k, int B, int T, int H,
                                                            float* __restrict__ dq, int acc_q,
                                                            float* __restrict__ dk, int acc_k) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  for (int h = lane; h < H; h += 64) {
    const float qv = q[b * H + h];
    float a = 0.f;
    for (int t = 0; t < T; ++t) {
      const int64_t bt = b * T + t;
      const float* d = d_all + bt * 4 * H;
      const float kv = k[bt * H + h];
      const float g = d[H + h] - d[2 * H + h] + d[3 * H + h] * qv;
      float* o = dk + bt * H + h;
      *o = acc_k ? *o + g : g;
      a += d[h] + d[2 * H + h] + d[3 * H + h] * kv;
    }
    float* o = dq + b * H + h;
    *o = acc_q ? *o + a : a;
  }
}

// The same with a block per sample for H dividing 256: thread (t-group, h) walks every (256/H)-th history
// position, four positions in flight, so the loads of one step are whole contiguous rows of d_all and k;
// the t-groups' dq terms are added through LDS in group order.
__global__ __launch_bounds__(256) void din_concat_bwd_block_kernel(const float* __restrict__ d_all,
                                                                  const float* __restrict__ q,
                                                                  const float* __restrict__ k, int B, int T, int H,
                                                                  float* __restrict__ dq, int acc_q,
                                                                  float* __restrict__ dk, int acc_k) {
  __shared__ float red[256];
  const int64_t b = blockIdx.x;
  const int G = 256 / H;
  const int tg = threadIdx.x / H, h = threadIdx.x - tg * H;
  const float qv = q[b * H + h];
  float a = 0.f;
  for (int t0 = tg; t0 < T; t0 += 4 * G) {
    float d0[4], d1[4], d2[4], d3[4], kv[4], old[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      const bool live = t < T;
      const int64_t bt = b * T + (live ? t : 0);
      const float* d = d_all + bt * 4 * H;
      d0[u] = live ? d[h] : 0.f;
      d1[u] = live ? d[H + h] : 0.f;
      d2[u] = live ? d[2 * H + h] : 0.f;
      d3[u] = live ? d[3 * H + h] : 0.f;
      kv[u] = live ? k[bt * H + h] : 0.f;
      old[u] = (live && acc_k) ? dk[bt * H + h] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      if (t < T) dk[(b * T + t) * H + h] = old[u] + (d1[u] - d2[u] + d3[u] * qv);
      a += d0[u] + d2[u] + d3[u] * kv[u];
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (tg == 0) {
    float v = 0.f;
    for (int g2 = 0; g2 < G; ++g2) v += red[g2 * H + h];
    float* o = dq + b * H + h;
    *o = acc_q ? *o + v : v;
  }
}

// att[b,:] = softmax( where(t < sl[b], score[b,t], -2^32+1) / sqrt(H) );  out[b,:] = att[b,:] . K[b]
// One wave per sample; T is swept in 64-wide steps (max, then sum of exponentials, then the pooling).
__global__ __launch_bounds__(256) void din_attn_pool_fwd_kernel(const float* __restrict__ score,
                                                               const int32_t* __restrict__ sl,
                                                               const float* __restrict__ K, int B, int T, int H,
                                                               float* __restrict__ att, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float inv = 1.f / sqrtf((float)H);
  const float pad = -4294967295.f;  // -2**32 + 1
  const int len = sl[b];
  float mx = -INFINITY;
  for (int t = lane; t < T; t += 64) {
    const float v = (t < len ? score[b * T + t] : pad) * inv;
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float v = (t < len ? score[b * T + t] : pad) * inv;
    sum += expf(v - mx);
  }
  sum = wave_sum(sum);
  const float rs = 1.f / sum;
  for (int t = lane; t < T; t += 64) {
    const float v = (t < len ? score[b * T + t] : pad) * inv;
    att[b * T + t] = expf(v - mx) * rs;
  }
  // pooling: every lane recomputes the weights from the scores (no cross-lane traffic through memory)
  for (int h = lane; h < H; h += 64) {
    float a = 0.f;
    for (int t = 0; t < T; ++t) {
      const float v = (t < len ? score[b * T + t] : pad) * inv;
      a += expf(v - mx) * rs * K[(b * T + t) * H + h];
    }
    out[b * H + h] = a;
  }
}

// The same with a block per sample (H divides 256): the softmax over T on all 256 threads, the weights
// staged in LDS, then thread (t-group, h) pools every (256/H)-th position - whole rows of K per step.
__global__ __launch_bounds__(256) void din_attn_pool_fwd_block_kernel(const float* __restrict__ score,
                                                                     const int32_t* __restrict__ sl,
                                                                     const float* __restrict__ K, int B, int T, int H,
                                                                     float* __restrict__ att, float* __restrict__ out) {
  extern __shared__ float sm[];       // att_s[T], red[256], red4[4]
  float* att_s = sm;
  float* red = sm + T;
  float* red4 = red + 256;
  const int64_t b = blockIdx.x;
  const float inv = 1.f / sqrtf((float)H);
  const float pad = -4294967295.f;  // -2**32 + 1
  const int len = sl[b];
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) mx = fmaxf(mx, (t < len ? score[b * T + t] : pad) * inv);
  mx = block_max256(mx, red4);
  float sum = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) sum += expf((t < len ? score[b * T + t] : pad) * inv - mx);
  sum = block_sum256(sum, red4);
  const float rs = 1.f / sum;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float a = expf((t < len ? score[b * T + t] : pad) * inv - mx) * rs;
    att[b * T + t] = a;
    att_s[t] = a;
  }
  __syncthreads();
  const int G = 256 / H;
  const int tg = threadIdx.x / H, h = threadIdx.x - tg * H;
  const float* Kb = K + b * T * H;
  // masked positions have weight exactly 0 (exp of -2^32 below the maximum) unless every position is masked (length 0:
  // uniform weights): their rows of K add nothing and are not read
  const int Tv = len > 0 && len < T ? len : T;
  float acc = 0.f;
  for (int t0 = tg; t0 < Tv; t0 += 4 * G) {
    float kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      kv[u] = t < Tv ? Kb[(int64_t)t * H + h] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      if (t < Tv) acc += att_s[t] * kv[u];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (tg == 0) {
    float v = 0.f;
    for (int g2 = 0; g2 < G; ++g2) v += red[g2 * H + h];
    out[b * H + h] = v;
  }
}

// datt[t] = dout . K[t];  dscore[t] = att[t] * (datt[t] - sum_j att[j] datt[j]) / sqrt(H)  (0 where masked)
// dK[b,t,:] (+)= att[t] * dout
__global__ __launch_bounds__(256) void din_attn_pool_bwd_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ att,
                                                               const int32_t* __restrict__ sl,
                                                               const float* __restrict__ K, int B, int T, int H,
                                                               float* __restrict__ dscore, float* __restrict__ dK,
                                                               int acc_k) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float inv = 1.f / sqrtf((float)H);
  const int len = sl[b];
  float dot = 0.f;
  for (int t = lane; t < T; t += 64) {
    float da = 0.f;
    for (int h = 0; h < H; ++h) da += dout[b * H + h] * K[(b * T + t) * H + h];
    dscore[b * T + t] = da;  // staged: datt
    dot += att[b * T + t] * da;
  }
  dot = wave_sum(dot);
  for (int t = lane; t < T; t += 64) {
    const float a = att[b * T + t];
    const float da = dscore[b * T + t];
    dscore[b * T + t] = t < len ? a * (da - dot) * inv : 0.f;
    for (int h = 0; h < H; ++h) {
      float* o = dK + (b * T + t) * H + h;
      const float g = a * dout[b * H + h];
      *o = acc_k ? *o + g : g;
    }
  }
}

// Block per sample (H divides 256, H % 4 == 0): thread t forms datt[t] from its own row of K with 16-byte
// loads, the block adds sum_j att[j] datt[j] in a fixed order, then thread (t-group, h) writes dK row-wise.
__global__ __launch_bounds__(256) void din_attn_pool_bwd_block_kernel(const float* __restrict__ dout,
                                                                     const float* __restrict__ att,
                                                                     const int32_t* __restrict__ sl,
                                                                     const float* __restrict__ K, int B, int T, int H,
                                                                     float* __restrict__ dscore,
                                                                     float* __restrict__ dK, int acc_k) {
  extern __shared__ float sm[];       // att_s[T], do_s[H], red4[4]
  float* att_s = sm;
  float* do_s = sm + T;
  float* red4 = do_s + H;
  const int64_t b = blockIdx.x;
  const float inv = 1.f / sqrtf((float)H);
  const int len = sl[b];
  for (int h = threadIdx.x; h < H; h += 256) do_s[h] = dout[b * H + h];
  __syncthreads();
  const float* Kb = K + b * T * H;
  // masked positions (weight exactly 0 unless the length is 0: uniform weights) add nothing to the dot product and
  // get dscore = 0: their rows of K are not read
  const int Tv = len > 0 && len < T ? len : T;
  float dot = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    float da = 0.f;
    if (t < Tv)
      for (int h = 0; h < H; h += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(Kb + (int64_t)t * H + h);
        da += (do_s[h] * kv.x + do_s[h + 1] * kv.y) + (do_s[h + 2] * kv.z + do_s[h + 3] * kv.w);
      }
    const float a = att[b * T + t];
    att_s[t] = a;
    dscore[b * T + t] = da;   // staged: datt
    dot += a * da;
  }
  dot = block_sum256(dot, red4);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float da = dscore[b * T + t];
    dscore[b * T + t] = t < len ? att_s[t] * (da - dot) * inv : 0.f;
  }
  const int G = 256 / H;
  const int tg = threadIdx.x / H, h = threadIdx.x - tg * H;
  const float dv = do_s[h];
  float* dKb = dK + b * T * H;
  for (int t0 = tg; t0 < T; t0 += 4 * G) {
    float old[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      old[u] = (acc_k && t < T) ? dKb[(int64_t)t * H + h] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      if (t < T) dKb[(int64_t)t * H + h] = old[u] + att_s[t] * dv;
    }
  }
}

// y = gamma * (x - mean) * rsqrt(var + eps) + beta  (any of gamma/beta/mean may be NULL: 1/0/0);
// xhat (optional) = (x - mean) * rsqrt(var + eps).  tf.layers.batch_normalization, training=False.
__global__ __launch_bounds__(256) void bn_infer_kernel(const float* __restrict__ x, int64_t x_ld, int rows, int cols,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, float* __restrict__ y, int64_t y_ld,
                                                      float* __restrict__ xhat, int64_t xhat_ld) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    const float xh = (x[r * x_ld + c] - (mean ? mean[c] : 0.f)) * rsqrtf((var ? var[c] : 1.f) + eps);
    if (xhat) xhat[r * xhat_ld + c] = xh;
    y[r * y_ld + c] = (gamma ? gamma[c] : 1.f) * xh + (beta ? beta[c] : 0.f);
  }
}

// Dice (DIN/DIN.py:149-166, inference-mode statistics): p = sigmoid(x * rsqrt(1 + eps));
// y = alpha*(1-p)*x + p*x.   Backward: dx = dy * (alpha*(1-p) + p + x*(1-alpha)*p*(1-p)*c),
// dalpha_src = dy * (1-p) * x  (column-summed by the caller).
__global__ __launch_bounds__(256) void dice_kernel(int bwd, const float* __restrict__ x, int64_t x_ld, int rows,
                                                  int cols, const float* __restrict__ alpha, float eps,
                                                  const float* __restrict__ dy, int64_t dy_ld, float* __restrict__ out,
                                                  int64_t out_ld, float* __restrict__ dalpha_src, int64_t da_ld) {
  const float c0 = rsqrtf(1.f + eps);
  const int64_t total = (int64_t)rows * cols;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(q % cols);
    const int64_t r = q / cols;
    const float xv = x[r * x_ld + c];
    const float p = 1.f / (1.f + expf(-xv * c0));
    const float a = alpha[c];
    if (!bwd) {
      out[r * out_ld + c] = a * (1.f - p) * xv + p * xv;
    } else {
      const float g = dy[r * dy_ld + c];
      out[r * out_ld + c] = g * (a * (1.f - p) + p + xv * (1.f - a) * p * (1.f - p) * c0);
      if (dalpha_src) dalpha_src[r * da_ld + c] = g * (1.f - p) * xv;
    }
  }
}

// Elementwise backward of a per-column layer and the column sums its parameters need, in one launch, for the
// narrow [rows, cols] tensors after the attention (rows = batch): a block owns 16 columns and walks all rows in 64
// row phases (thread = (column, phase); four rows in flight), the phases are added in phase order through LDS, so the
// sums are bitwise reproducible.  Few blocks: meant for batches up to some ten thousand rows, where the separate
// elementwise pass, two-stage column sums and their five launches cost more than the arithmetic.
//   BN (inference statistics, DIN/DIN.py:45,53): dx = gamma * (dy * rsqrt(var + eps)); sum_a = sum dy * xhat (dgamma),
//       sum_b = sum dy (dbeta).
//   DICE (DIN/DIN.py:149-166): dx as dice_kernel's backward; sum_a = sum dy * (1 - p) * x (dalpha).
constexpr int CB_COLS = 16, CB_PH = 64;
template <bool DICE>
__global__ __launch_bounds__(CB_COLS * CB_PH) void colwise_bwd_kernel(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ dy, int64_t dy_ld, int rows, int cols,
    const float* __restrict__ par, const float* __restrict__ var, float eps, float* __restrict__ dx, int64_t dx_ld,
    float* __restrict__ sum_a, float* __restrict__ sum_b) {
  __shared__ float red_a[CB_PH][CB_COLS + 1], red_b[CB_PH][CB_COLS + 1];
  const int lc = threadIdx.x % CB_COLS, ph = threadIdx.x / CB_COLS;
  const int c = blockIdx.x * CB_COLS + lc;
  const bool on = c < cols;
  const float pv = on ? par[c] : 0.f;                                   // gamma / alpha
  const float rs = (!DICE && on) ? rsqrtf((var ? var[c] : 1.f) + eps) : 0.f;
  const float c0 = rsqrtf(1.f + eps);
  float acc_a = 0.f, acc_b = 0.f;
  if (on) {
    for (int r0 = ph; r0 < rows; r0 += 4 * CB_PH) {
      float xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * CB_PH;
        const bool live = r < rows;
        xv[u] = live ? x[(int64_t)r * x_ld + c] : 0.f;
        gv[u] = live ? dy[(int64_t)r * dy_ld + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * CB_PH;
        if (r >= rows) break;
        float d;
        if (DICE) {
          const float p = 1.f / (1.f + expf(-xv[u] * c0));
          d = gv[u] * (pv * (1.f - p) + p + xv[u] * (1.f - pv) * p * (1.f - p) * c0);
          acc_a += gv[u] * (1.f - p) * xv[u];
        } else {
          d = pv * (gv[u] * rs);
          acc_a += gv[u] * xv[u];
          acc_b += gv[u];
        }
        dx[(int64_t)r * dx_ld + c] = d;
      }
    }
  }
  red_a[ph][lc] = acc_a;
  red_b[ph][lc] = acc_b;
  __syncthreads();
  if (ph == 0 && on) {
    float a = 0.f, b = 0.f;
    for (int q = 0; q < CB_PH; ++q) { a += red_a[q][lc]; b += red_b[q][lc]; }
    sum_a[c] = a;
    if (!DICE && sum_b) sum_b[c] = b;
  }
}

// g[r,c] *= y[r,c] * (1 - y[r,c])
__global__ __launch_bounds__(256) void dsigmoid_kernel(float* __restrict__ g, int64_t g_ld,
                                                      const float* __restrict__ y, int64_t y_ld, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(p % cols);
    const int64_t r = p / cols;
    const float yv = y[r * y_ld + c];
    g[r * g_ld + c] *= yv * (1.f - yv);
  }
}

// mf_auc partial: cnt = #{b : x_i[b] - x_j[b] > 0}; single block, exact integer count
__global__ __launch_bounds__(256) void count_positive_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int n, float* __restrict__ out_mean) {
  __shared__ int red[256];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += 256) c += (a[i] - b[i]) > 0.f;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_mean[0] = (float)red[0] / (float)n;
}

static unsigned grid_for(int64_t total) {
  const int64_t b = cdiv(total, 256);
  return (unsigned)(b < 8192 ? (b > 0 ? b : 1) : 8192);
}

}  // namespace ctr

using namespace ctr;

extern "C" int ctr_din_build_idx(const int32_t* ids, const int32_t* cate_list, int64_t n, int item_count,
                                 int32_t* out, void* stream) {
  CTR_REQUIRE(n >= 0 && item_count > 0, "ctr_din_build_idx: bad size");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(ids && cate_list && out, "ctr_din_build_idx: null pointer");
  hipLaunchKernelGGL(din_build_idx_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ids, cate_list, n,
                     item_count, out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_concat_fwd(const float* q, const float* k, int B, int T, int H, float* out, void* stream) {
  CTR_REQUIRE(B >= 0 && T >= 0 && H > 0, "ctr_din_concat_fwd: bad size");
  if ((int64_t)B * T == 0) return CTR_OK;
  CTR_REQUIRE(q && k && out, "ctr_din_concat_fwd: null pointer");
  hipLaunchKernelGGL(din_concat_kernel, dim3(grid_for((int64_t)B * T * H)), dim3(256), 0, (hipStream_t)stream, q, k,
                     B, T, H, out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_concat_bwd(const float* d_all, const float* q, const float* k, int B, int T, int H,
                                  float* dq, int accumulate_q, float* dk, int accumulate_k, void* stream) {
  CTR_REQUIRE(B >= 0 && T >= 0 && H > 0, "ctr_din_concat_bwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(d_all && q && k && dq && dk, "ctr_din_concat_bwd: null pointer");
  if (H <= 256 && 256 % H == 0)
    hipLaunchKernelGGL(din_concat_bwd_block_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, d_all, q, k,
                       B, T, H, dq, accumulate_q, dk, accumulate_k);
  else
    hipLaunchKernelGGL(din_concat_bwd_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, d_all, q,
                       k, B, T, H, dq, accumulate_q, dk, accumulate_k);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_attn_pool_fwd(const float* score, const int32_t* sl, const float* K, int B, int T, int H,
                                     float* att_out, float* out, void* stream) {
  CTR_REQUIRE(B >= 0 && T > 0 && H > 0, "ctr_din_attn_pool_fwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(score && sl && K && att_out && out, "ctr_din_attn_pool_fwd: null pointer");
  if (H <= 256 && 256 % H == 0 && T <= 8192)
    hipLaunchKernelGGL(din_attn_pool_fwd_block_kernel, dim3((unsigned)B), dim3(256), (size_t)(T + 260) * sizeof(float),
                       (hipStream_t)stream, score, sl, K, B, T, H, att_out, out);
  else
    hipLaunchKernelGGL(din_attn_pool_fwd_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, score,
                       sl, K, B, T, H, att_out, out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_din_attn_pool_bwd(const float* dout, const float* att, const int32_t* sl, const float* K, int B,
                                     int T, int H, float* dscore, float* dK, int accumulate_k, void* stream) {
  CTR_REQUIRE(B >= 0 && T > 0 && H > 0, "ctr_din_attn_pool_bwd: bad size");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(dout && att && sl && K && dscore && dK, "ctr_din_attn_pool_bwd: null pointer");
  if (H <= 256 && 256 % H == 0 && H % 4 == 0 && T <= 8192 && ((uintptr_t)K % 16) == 0)
    hipLaunchKernelGGL(din_attn_pool_bwd_block_kernel, dim3((unsigned)B), dim3(256),
                       (size_t)(T + H + 4) * sizeof(float), (hipStream_t)stream, dout, att, sl, K, B, T, H, dscore, dK,
                       accumulate_k);
  else
    hipLaunchKernelGGL(din_attn_pool_bwd_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, dout,
                       att, sl, K, B, T, H, dscore, dK, accumulate_k);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_bn_infer(const float* x, int64_t x_ld, int rows, int cols, const float* gamma, const float* beta,
                            const float* mean, const float* var, float eps, float* y, int64_t y_ld, float* xhat,
                            int64_t xhat_ld, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_bn_infer: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(x && y && x_ld >= cols && y_ld >= cols, "ctr_bn_infer: null pointer / ld");
  hipLaunchKernelGGL(bn_infer_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, x,
                     x_ld, rows, cols, gamma, beta, mean, var, eps, y, y_ld, xhat, xhat_ld);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dice_fwd(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps, float* y,
                            int64_t y_ld, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_dice_fwd: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(x && alpha && y, "ctr_dice_fwd: null pointer");
  hipLaunchKernelGGL(dice_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, 0, x, x_ld,
                     rows, cols, alpha, eps, (const float*)nullptr, (int64_t)0, y, y_ld, (float*)nullptr, (int64_t)0);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dice_bwd(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps,
                            const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld, float* dalpha_src,
                            int64_t dalpha_ld, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_dice_bwd: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(x && alpha && dy && dx, "ctr_dice_bwd: null pointer");
  hipLaunchKernelGGL(dice_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, 1, x, x_ld,
                     rows, cols, alpha, eps, dy, dy_ld, dx, dx_ld, dalpha_src, dalpha_ld);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_bn_infer_bwd(const float* dy, int64_t dy_ld, const float* xhat, int64_t xhat_ld, int rows, int cols,
                                const float* gamma, const float* var, float eps, float* dx, int64_t dx_ld,
                                float* dgamma, float* dbeta, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_bn_infer_bwd: bad size");
  if (cols == 0) return CTR_OK;
  CTR_REQUIRE(dy && xhat && gamma && dx && dgamma && dy_ld >= cols && xhat_ld >= cols && dx_ld >= cols,
              "ctr_bn_infer_bwd: null pointer / ld");
  hipLaunchKernelGGL(colwise_bwd_kernel<false>, dim3((unsigned)cdiv(cols, CB_COLS)), dim3(CB_COLS * CB_PH), 0,
                     (hipStream_t)stream, xhat, xhat_ld, dy, dy_ld, rows, cols, gamma, var, eps, dx, dx_ld, dgamma, dbeta);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dice_bwd_sum(const float* x, int64_t x_ld, int rows, int cols, const float* alpha, float eps,
                                const float* dy, int64_t dy_ld, float* dx, int64_t dx_ld, float* dalpha, void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_dice_bwd_sum: bad size");
  if (cols == 0) return CTR_OK;
  CTR_REQUIRE(x && alpha && dy && dx && dalpha && x_ld >= cols && dy_ld >= cols && dx_ld >= cols,
              "ctr_dice_bwd_sum: null pointer / ld");
  hipLaunchKernelGGL(colwise_bwd_kernel<true>, dim3((unsigned)cdiv(cols, CB_COLS)), dim3(CB_COLS * CB_PH), 0,
                     (hipStream_t)stream, x, x_ld, dy, dy_ld, rows, cols, alpha, (const float*)nullptr, eps, dx, dx_ld,
                     dalpha, (float*)nullptr);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_dsigmoid_apply(float* g, int64_t g_ld, const float* y, int64_t y_ld, int rows, int cols,
                                  void* stream) {
  CTR_REQUIRE(rows >= 0 && cols >= 0, "ctr_dsigmoid_apply: bad size");
  if ((int64_t)rows * cols == 0) return CTR_OK;
  CTR_REQUIRE(g && y, "ctr_dsigmoid_apply: null pointer");
  hipLaunchKernelGGL(dsigmoid_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, (hipStream_t)stream, g,
                     g_ld, y, y_ld, rows, cols);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_mean_positive_diff(const float* a, const float* b, int n, float* out_mean, void* stream) {
  CTR_REQUIRE(n > 0 && a && b && out_mean, "ctr_mean_positive_diff: bad argument");
  hipLaunchKernelGGL(count_positive_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, n, out_mean);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
