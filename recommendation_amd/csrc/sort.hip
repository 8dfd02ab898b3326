// Stable LSD radix sort of (uint32 key, uint32 value) pairs, 9 bits per pass, and the
// (key, src) pair builder of the embedding backward.  Each pass is three launches:
// per-tile digit histogram -> exclusive scan of the [digit][tile] counts -> stable scatter.
// Stability inside a tile comes from wave-level peer masks (ballots over the digit bits)
// plus per-wave running digit counters in LDS, so the result is bitwise reproducible.
#include "common.h"
#include <mutex>

namespace ctr {

constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_CHUNKS = 8;                       // 64-key chunks per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_CHUNKS;  // 2048 keys per block
// digit width RB: 9 bits (26-bit row keys sort in 3 passes), or 10 where that saves a pass (19-20, 28-30 bits)
constexpr int RBITS_MAX = 10;
constexpr int RBINS_MAX = 1 << RBITS_MAX;
static inline int digit_bits(int key_bits) {
  return (key_bits + 9) / 10 < (key_bits + 8) / 9 ? 10 : 9;
}

// One block per sort tile: builds the tile's pairs and, while the keys are in registers, the tile's digit
// histogram of the first radix pass (counts[digit][tile], what radix_hist_kernel would write for shift 0), so the
// first pass starts at its scan.  32-bit index arithmetic: B*F < 2^32 is required by the callers.
template <int RB>
__global__ __launch_bounds__(SORT_THREADS) void build_pairs_kernel(
    const int32_t* __restrict__ idx, const int64_t* __restrict__ row_offset, int B, int F, int N,
    int64_t total_rows, int rmode, uint32_t* __restrict__ keys, uint32_t* __restrict__ src, int nblocks,
    uint32_t* __restrict__ counts) {  constexpr int RBITS = RB, RBINS = 1 << RB, BINS_PER_THREAD = RBINS / SORT_THREADS;

  __shared__ uint32_t hist[RBINS];
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) hist[d] = 0;
  __syncthreads();
  const uint32_t n = (uint32_t)B * (uint32_t)F;
  const uint32_t base = blockIdx.x * (uint32_t)SORT_TILE;
  int x[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = base + c * SORT_THREADS + threadIdx.x;
    x[c] = i < n ? idx[i] : 0;
  }
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = base + c * SORT_THREADS + threadIdx.x;
    if (i >= n) continue;
    const uint32_t b = i / (uint32_t)F, f = i - b * (uint32_t)F;
    const int64_t off = row_offset[f];
    const int64_t V = row_offset[f + 1] - off;
    const bool ok = x[c] >= 0 && x[c] < V;
    uint32_t k, s;
    if (rmode) {
      k = ok ? (uint32_t)(off + 1) : (uint32_t)total_rows;
      s = ok ? (uint32_t)((int64_t)b * N + off + x[c]) : 0u;
    } else {
      k = ok ? (uint32_t)(off + x[c]) : (uint32_t)total_rows;
      s = b * (uint32_t)N + f;
    }
    keys[i] = k;
    src[i] = s;
    atomicAdd(&hist[k & (RBINS - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) counts[(int64_t)d * nblocks + blockIdx.x] = hist[d];
}

// Pair builder for multi-hot bags (CSR): one pair per member j of bag (b, f), key = the member's table row,
// src = the bag's row of dZ.  A thread walks one bag, so the pairs come out in member order.
__global__ __launch_bounds__(256) void build_bag_pairs_kernel(
    const int32_t* __restrict__ idx, const int32_t* __restrict__ bag_offsets, const int64_t* __restrict__ row_offset,
    int B, int F, int N, int64_t total_rows, uint32_t* __restrict__ keys, uint32_t* __restrict__ src) {
  const int64_t nbags = (int64_t)B * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbags; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int64_t b = i / F;
    const int64_t off = row_offset[f];
    const int64_t V = row_offset[f + 1] - off;
    const int lo = bag_offsets[i], hi = bag_offsets[i + 1];
    for (int j = lo; j < hi; ++j) {
      const int x = idx[j];
      keys[j] = (x >= 0 && x < V) ? (uint32_t)(off + x) : (uint32_t)total_rows;
      src[j] = (uint32_t)(b * N + f);
    }
  }
}

template <int RB>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(
    const uint32_t* __restrict__ keys, int64_t n, int shift, int nblocks, uint32_t* __restrict__ counts) {  constexpr int RBITS = RB, RBINS = 1 << RB, BINS_PER_THREAD = RBINS / SORT_THREADS;

  __shared__ uint32_t hist[RBINS];
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  uint32_t k[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = base + c * SORT_THREADS + threadIdx.x;
    k[c] = p < n ? keys[p] : 0u;
  }
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = base + c * SORT_THREADS + threadIdx.x;
    if (p < n) atomicAdd(&hist[(k[c] >> shift) & (RBINS - 1)], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) counts[(int64_t)d * nblocks + blockIdx.x] = hist[d];
}

// Block d turns row d of counts[digit][tile] into its exclusive prefix over the tiles and
// stores the row total in totals[d]; the scatter kernel adds the prefix over the digits itself.
__global__ __launch_bounds__(256) void radix_scan_kernel(uint32_t* __restrict__ counts, int nblocks,
                                                        uint32_t* __restrict__ totals) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* row = counts + (int64_t)blockIdx.x * nblocks;
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    // four consecutive tiles per thread, then a scan over the threads
    uint32_t c[4], loc = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 4 * threadIdx.x + u;
      c[u] = i < nblocks ? row[i] : 0u;
      loc += c[u];
    }
    uint32_t incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) woff += wsum[w];
      tot += wsum[w];
    }
    uint32_t run = carry + woff + incl - loc;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + 4 * threadIdx.x + u;
      if (i < nblocks) row[i] = run;
      run += c[u];
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <int RB>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t n, int shift,
    int nblocks, const uint32_t* __restrict__ scanned, const uint32_t* __restrict__ totals,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {  constexpr int RBITS = RB, RBINS = 1 << RB, BINS_PER_THREAD = RBINS / SORT_THREADS;

  __shared__ uint32_t whist[SORT_WAVES][RBINS];
  __shared__ uint32_t dbase[RBINS];
  __shared__ uint32_t wtot[SORT_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // the tile's keys first: their loads overlap the prefix work below
  const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * SORT_CHUNKS * 64;
  uint32_t key[SORT_CHUNKS], val[SORT_CHUNKS], rank[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    const bool valid = p < n;
    key[c] = valid ? keys_in[p] : 0u;
    val[c] = valid ? vals_in[p] : 0u;
  }
  {  // exclusive prefix of the digit totals; thread t owns digits BINS_PER_THREAD*t ...
    uint32_t tv[BINS_PER_THREAD], loc = 0;
#pragma unroll
    for (int u = 0; u < BINS_PER_THREAD; ++u) {
      tv[u] = totals[BINS_PER_THREAD * threadIdx.x + u];
      loc += tv[u];
    }
    uint32_t incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    for (int i = threadIdx.x; i < SORT_WAVES * RBINS; i += SORT_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    uint32_t run = incl - loc;
    for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int u = 0; u < BINS_PER_THREAD; ++u) {
      dbase[BINS_PER_THREAD * threadIdx.x + u] = run;
      run += tv[u];
    }
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    const bool valid = p < n;
    const uint32_t d = (key[c] >> shift) & (RBINS - 1);
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < RBITS; ++bit) {
      const bool set = (d >> bit) & 1u;
      const unsigned long long m = __ballot(valid && set);
      peers &= set ? m : ~m;
    }
    uint32_t prior = 0;
    if (valid) prior = whist[wave][d];
    const uint32_t in_chunk = (uint32_t)__popcll(peers & lt_mask);
    rank[c] = prior + in_chunk;
    // the lowest peer lane publishes the chunk's count for this digit
    if (valid && in_chunk == 0) whist[wave][d] = prior + (uint32_t)__popcll(peers);
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves + this tile's global base
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) {
    uint32_t run = dbase[d] + scanned[(int64_t)d * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const uint32_t c = whist[w][d];
      whist[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    if (p < n) {
      const uint32_t d = (key[c] >> shift) & (RBINS - 1);
      const uint32_t dst = whist[wave][d] + rank[c];
      keys_out[dst] = key[c];
      vals_out[dst] = val[c];
    }
  }
}

static int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 32 && (max_value >> b) != 0) ++b;
  return b;
}

struct SortWs {
  uint32_t *keys_tmp, *vals_tmp, *counts, *totals;
  int nblocks;
};

static size_t sort_ws_bytes(int64_t n) {
  const int64_t nblocks = cdiv(n > 0 ? n : 1, SORT_TILE);
  return align_up((size_t)n * 4, 256) * 2 + align_up((size_t)nblocks * RBINS_MAX * 4, 256) + RBINS_MAX * 4;
}

static SortWs carve(void* ws, int64_t n) {
  SortWs w;
  char* p = (char*)ws;
  w.nblocks = (int)cdiv(n > 0 ? n : 1, SORT_TILE);
  w.keys_tmp = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
  w.vals_tmp = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
  w.counts = (uint32_t*)p; p += align_up((size_t)w.nblocks * RBINS_MAX * 4, 256);
  w.totals = (uint32_t*)p;
  return w;
}

// Sorts the pairs currently in (ka, va); the result lands in (keys_out, vals_out).
// (ka, va) must be (keys_out, vals_out) when the pass count is even, else the tmp pair.
static int run_passes(uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, int64_t n, int passes, int rb,
                      const SortWs& w, hipStream_t st, bool first_hist_done = false) {
  for (int p = 0; p < passes; ++p) {
    const int shift = rb * p;
    if (p > 0 || !first_hist_done) {
      if (rb == 10)
        hipLaunchKernelGGL(radix_hist_kernel<10>, dim3(w.nblocks), dim3(SORT_THREADS), 0, st, ka, n, shift, w.nblocks,
                           w.counts);
      else
        hipLaunchKernelGGL(radix_hist_kernel<9>, dim3(w.nblocks), dim3(SORT_THREADS), 0, st, ka, n, shift, w.nblocks,
                           w.counts);
    }
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1 << rb), dim3(256), 0, st, w.counts, w.nblocks, w.totals);
    if (rb == 10)
      hipLaunchKernelGGL(radix_scatter_kernel<10>, dim3(w.nblocks), dim3(SORT_THREADS), 0, st, ka, va, n, shift,
                         w.nblocks, w.counts, w.totals, kb, vb);
    else
      hipLaunchKernelGGL(radix_scatter_kernel<9>, dim3(w.nblocks), dim3(SORT_THREADS), 0, st, ka, va, n, shift,
                         w.nblocks, w.counts, w.totals, kb, vb);
    uint32_t* t;
    t = ka; ka = kb; kb = t;
    t = va; va = vb; vb = t;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// Evaluation metrics on the device: AUC with average ranks for ties (sort the scores with the radix
// sort above, then every positive finds its tie group by two binary searches) and the log loss of
// tf.losses.log_loss; double-precision sums in a fixed order.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void auc_keys_kernel(const float* __restrict__ score, const float* __restrict__ label,
                                                      int64_t n, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t u = __float_as_uint(score[i] + 0.f);            // -0 -> +0 so equal scores get equal keys
    keys[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);    // unsigned order == float order
    vals[i] = label[i] > 0.5f ? 1u : 0u;
  }
}

constexpr int AUC_BLOCKS = 256;

// part[blk] = (sum of average ranks of the positives, #positives, sum of log-loss terms) over the block's slice
__global__ __launch_bounds__(256) void auc_partial_kernel(const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals, int64_t n,
                                                         const float* __restrict__ score,
                                                         const float* __restrict__ label, double* __restrict__ part) {
  __shared__ double red[3][256];
  double rs = 0.0, np_ = 0.0, ll = 0.0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo0 = (int64_t)blockIdx.x * per, hi0 = lo0 + per < n ? lo0 + per : n;
  for (int64_t i = lo0 + threadIdx.x; i < hi0; i += 256) {
    if (vals[i]) {
      const uint32_t k = keys[i];
      int64_t a = 0, b = i;                 // first index with key >= k
      while (a < b) { const int64_t m = (a + b) >> 1; if (keys[m] < k) a = m + 1; else b = m; }
      const int64_t first = a;
      a = i; b = n;                         // first index with key > k
      while (a < b) { const int64_t m = (a + b) >> 1; if (keys[m] <= k) a = m + 1; else b = m; }
      rs += 0.5 * (double)(first + a - 1) + 1.0;   // ranks first+1 .. a, averaged
      np_ += 1.0;
    }
    const double p = (double)score[i], y = label[i] > 0.5f ? 1.0 : 0.0;
    ll += -y * log(p + 1e-7) - (1.0 - y) * log(1.0 - p + 1e-7);
  }
  red[0][threadIdx.x] = rs; red[1][threadIdx.x] = np_; red[2][threadIdx.x] = ll;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    for (int c = 0; c < 3; ++c) part[blockIdx.x * 3 + c] = red[c][0];
}

__global__ void auc_final_kernel(const double* __restrict__ part, int nblocks, int64_t n, float* __restrict__ auc_out,
                                 float* __restrict__ logloss_out) {
  double rs = 0.0, np_ = 0.0, ll = 0.0;
  for (int b = 0; b < nblocks; ++b) { rs += part[3 * b]; np_ += part[3 * b + 1]; ll += part[3 * b + 2]; }
  const double nn = (double)n - np_;
  if (auc_out) auc_out[0] = (np_ == 0.0 || nn == 0.0) ? __uint_as_float(0x7FC00000u)
                                                      : (float)((rs - np_ * (np_ + 1.0) / 2.0) / (np_ * nn));
  if (logloss_out) logloss_out[0] = (float)(ll / (double)n);
}

}  // namespace ctr

using namespace ctr;

extern "C" size_t ctr_auc_workspace_bytes(int64_t n) {
  return sort_ws_bytes(n) + 2 * align_up((size_t)(n > 0 ? n : 1) * 4, 256) + AUC_BLOCKS * 3 * sizeof(double);
}

extern "C" int ctr_auc(const float* score, const float* label, int64_t n, float* auc_out, float* logloss_out,
                       void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(n > 0 && n < 0x7FFFFFFFll, "ctr_auc: n must be in (0, 2^31)");
  CTR_REQUIRE(score && label && workspace && (auc_out || logloss_out), "ctr_auc: null pointer");
  if (workspace_bytes < ctr_auc_workspace_bytes(n)) {
    set_error("ctr_auc: workspace %zu < %zu", workspace_bytes, ctr_auc_workspace_bytes(n));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* p = (char*)workspace;
  uint32_t* keys = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
  uint32_t* vals = (uint32_t*)p; p += align_up((size_t)n * 4, 256);
  double* part = (double*)p; p += AUC_BLOCKS * 3 * sizeof(double);
  const SortWs w = carve(p, n);
  const int64_t blocks = cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048;
  const int rb = digit_bits(32), passes = (32 + rb - 1) / rb;
  // the pass count decides which buffer pair the sort starts from so that it ends in (keys, vals)
  uint32_t* ka = (passes % 2 == 0) ? keys : w.keys_tmp;
  uint32_t* va = (passes % 2 == 0) ? vals : w.vals_tmp;
  uint32_t* kb = (passes % 2 == 0) ? w.keys_tmp : keys;
  uint32_t* vb = (passes % 2 == 0) ? w.vals_tmp : vals;
  hipLaunchKernelGGL(auc_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, st, score, label, n, ka, va);
  run_passes(ka, va, kb, vb, n, passes, rb, w, st);
  const int nb = (int)(cdiv(n, 256) < AUC_BLOCKS ? cdiv(n, 256) : AUC_BLOCKS);
  hipLaunchKernelGGL(auc_partial_kernel, dim3(nb), dim3(256), 0, st, keys, vals, n, score, label, part);
  hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(1), 0, st, part, nb, n, auc_out, logloss_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" size_t ctr_emb_sort_workspace_bytes(int64_t n_items) { return sort_ws_bytes(n_items); }

extern "C" int ctr_emb_sort(const int32_t* idx, const int64_t* row_offset, int B, int F, int N,
                            int64_t total_rows, int rmode, uint32_t* keys_out, uint32_t* src_out,
                            void* workspace, size_t workspace_bytes, void* stream) {
  const int64_t n = (int64_t)B * F;
  CTR_REQUIRE(B >= 0 && F >= 0 && N > 0, "ctr_emb_sort: bad size");
  CTR_REQUIRE(total_rows > 0 && total_rows < 0xFFFFFFFFll, "ctr_emb_sort: total_rows must fit 32 bits");
  CTR_REQUIRE((int64_t)B * N < 0xFFFFFFFFll, "ctr_emb_sort: B*N must fit 32 bits");
  CTR_REQUIRE(n < 0xFFFFF000ll, "ctr_emb_sort: B*F must stay below 2^32 - 4096");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(idx && row_offset && keys_out && src_out && workspace, "ctr_emb_sort: null pointer");
  if (workspace_bytes < sort_ws_bytes(n)) {
    set_error("ctr_emb_sort: workspace %zu < %zu", workspace_bytes, sort_ws_bytes(n));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const SortWs w = carve(workspace, n);
  const int kbits = bits_for((uint64_t)total_rows), rb = digit_bits(kbits), passes = (kbits + rb - 1) / rb;
  uint32_t* ka = (passes % 2 == 0) ? keys_out : w.keys_tmp;
  uint32_t* va = (passes % 2 == 0) ? src_out : w.vals_tmp;
  uint32_t* kb = (passes % 2 == 0) ? w.keys_tmp : keys_out;
  uint32_t* vb = (passes % 2 == 0) ? w.vals_tmp : src_out;
  if (rb == 10)
    hipLaunchKernelGGL(build_pairs_kernel<10>, dim3((unsigned)w.nblocks), dim3(SORT_THREADS), 0, st, idx, row_offset, B,
                       F, N, total_rows, rmode, ka, va, w.nblocks, w.counts);
  else
    hipLaunchKernelGGL(build_pairs_kernel<9>, dim3((unsigned)w.nblocks), dim3(SORT_THREADS), 0, st, idx, row_offset, B,
                       F, N, total_rows, rmode, ka, va, w.nblocks, w.counts);
  run_passes(ka, va, kb, vb, n, passes, rb, w, st, true);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_emb_bag_sort(const int32_t* idx, const int32_t* bag_offsets, int64_t nnz,
                                const int64_t* row_offset, int B, int F, int N, int64_t total_rows,
                                uint32_t* keys_out, uint32_t* src_out, void* workspace, size_t workspace_bytes,
                                void* stream) {
  CTR_REQUIRE(B >= 0 && F >= 0 && N > 0 && nnz >= 0 && nnz <= 0x7FFFFFFFll, "ctr_emb_bag_sort: bad size");
  CTR_REQUIRE(total_rows > 0 && total_rows < 0xFFFFFFFFll, "ctr_emb_bag_sort: total_rows must fit 32 bits");
  CTR_REQUIRE((int64_t)B * N < 0xFFFFFFFFll, "ctr_emb_bag_sort: B*N must fit 32 bits");
  if (nnz == 0 || (int64_t)B * F == 0) return CTR_OK;
  CTR_REQUIRE(idx && bag_offsets && row_offset && keys_out && src_out && workspace, "ctr_emb_bag_sort: null pointer");
  if (workspace_bytes < sort_ws_bytes(nnz)) {
    set_error("ctr_emb_bag_sort: workspace %zu < %zu", workspace_bytes, sort_ws_bytes(nnz));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const SortWs w = carve(workspace, nnz);
  const int kbits = bits_for((uint64_t)total_rows), rb = digit_bits(kbits), passes = (kbits + rb - 1) / rb;
  uint32_t* ka = (passes % 2 == 0) ? keys_out : w.keys_tmp;
  uint32_t* va = (passes % 2 == 0) ? src_out : w.vals_tmp;
  uint32_t* kb = (passes % 2 == 0) ? w.keys_tmp : keys_out;
  uint32_t* vb = (passes % 2 == 0) ? w.vals_tmp : src_out;
  const int64_t nbags = (int64_t)B * F;
  const int64_t blocks = cdiv(nbags, 256) < 2048 ? cdiv(nbags, 256) : 2048;
  hipLaunchKernelGGL(build_bag_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, bag_offsets, row_offset,
                     B, F, N, total_rows, ka, va);
  run_passes(ka, va, kb, vb, nnz, passes, rb, w, st);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// ---------------------------------------------------------------------------------------
// The pair sort with its scratch in a file-scope device array instead of a caller-provided workspace, for
// callers that cannot allocate device memory.  Each GPU holds its own copy of the array (at its own address)
// from the moment the library is loaded on it until the process ends.
// ---------------------------------------------------------------------------------------
constexpr size_t SORT_STATIC_BYTES = (size_t)24 << 20;
static __device__ __align__(256) unsigned char sort_static_scratch[SORT_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_sort_scratch_ptr[MAX_DEVICES];      // device address of sort_static_scratch on each device, once resolved
std::mutex g_sort_scratch_mutex;
}  // namespace

// Device address of the current device's copy of the array (resolved once per device, with that device current).
static int sort_static_ptr(void** out) {
  int dev = -1;
  CTR_HIP(hipGetDevice(&dev));
  CTR_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "ctr_emb_sort_static: device ordinal %d not supported", dev);
  std::lock_guard<std::mutex> lock(g_sort_scratch_mutex);
  if (!g_sort_scratch_ptr[dev]) {
    void* p = nullptr;
    CTR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(sort_static_scratch)));
    CTR_REQUIRE(p && ((uintptr_t)p % 256) == 0, "ctr_emb_sort_static: scratch symbol did not resolve");
    g_sort_scratch_ptr[dev] = p;
  }
  *out = g_sort_scratch_ptr[dev];
  return CTR_OK;
}

extern "C" int ctr_emb_sort_static(const int32_t* idx, const int64_t* row_offset, int B, int F, int N,
                                   int64_t total_rows, int rmode, uint32_t* keys_out, uint32_t* src_out,
                                   void* stream) {
  const int64_t n = (int64_t)B * F;
  CTR_REQUIRE(B >= 0 && F >= 0, "ctr_emb_sort_static: bad size");
  CTR_REQUIRE(n <= CTR_EMB_SORT_STATIC_MAX_ITEMS && sort_ws_bytes(n) <= SORT_STATIC_BYTES,
              "ctr_emb_sort_static: B*F = %lld exceeds the built-in scratch (at most %d lookups per call); "
              "use ctr_emb_sort with a workspace", (long long)n, CTR_EMB_SORT_STATIC_MAX_ITEMS);
  void* scratch = nullptr;
  const int rc = sort_static_ptr(&scratch);
  if (rc != CTR_OK) return rc;
  // every word of the scratch that a pass reads (tile histograms, digit totals, ping-pong pairs) is written
  // earlier in the same call, so nothing has to be cleared between calls
  return ctr_emb_sort(idx, row_offset, B, F, N, total_rows, rmode, keys_out, src_out, scratch, SORT_STATIC_BYTES,
                      stream);
}

extern "C" int ctr_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, int64_t n, int key_bits,
                              uint32_t* keys_out, uint32_t* vals_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(n >= 0 && key_bits >= 1 && key_bits <= 32, "ctr_sort_pairs: bad size");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(keys_in && vals_in && keys_out && vals_out && workspace, "ctr_sort_pairs: null pointer");
  if (workspace_bytes < sort_ws_bytes(n)) {
    set_error("ctr_sort_pairs: workspace %zu < %zu", workspace_bytes, sort_ws_bytes(n));
    return CTR_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const SortWs w = carve(workspace, n);
  const int rb = digit_bits(key_bits), passes = (key_bits + rb - 1) / rb;
  uint32_t* ka = (passes % 2 == 0) ? keys_out : w.keys_tmp;
  uint32_t* va = (passes % 2 == 0) ? vals_out : w.vals_tmp;
  uint32_t* kb = (passes % 2 == 0) ? w.keys_tmp : keys_out;
  uint32_t* vb = (passes % 2 == 0) ? w.vals_tmp : vals_out;
  hipError_t e = hipMemcpyAsync(ka, keys_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(va, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { set_error("ctr_sort_pairs: copy failed: %s", hipGetErrorString(e)); return (int)e; }
  run_passes(ka, va, kb, vb, n, passes, rb, w, st);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
