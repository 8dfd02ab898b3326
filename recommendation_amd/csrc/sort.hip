// Stable LSD radix sort of (uint32 key, uint32 value) pairs, 9 or 10 bits per pass, and the
// (key, src) pair builder of the embedding backward.  A pass is three launches: per-tile digit
// histogram -> exclusive scan of the [tile][digit] counts -> stable scatter; the builder leaves
// the first pass's histogram behind, so that pass starts at its scan.  The embedding backward's pairs of batches
// between 1024 and 131072 samples take the field-segmented sort further down instead (a launch per pass).
// Stability inside a tile: the lanes of a 64-key chunk that carry the same digit find each other
// through a 64-bit lane mask per digit in LDS (an atomic OR of runs of neighbouring lanes, which
// commutes), and per-wave running digit counters in LDS order the chunks; the result is bitwise
// reproducible.  Between passes the pairs travel as one uint2 per pair, and a tile is put into
// digit order in LDS before it is written, so that neighbouring lanes store to neighbouring
// addresses (one 4-byte store per lane and array to ~500 places per tile is bound by the number
// of write requests, not by bytes).
#include "common.h"
#include <stdlib.h>
#include <mutex>

namespace ctr {

constexpr int SORT_THREADS = 512;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_CHUNKS = 8;                       // 64-key chunks per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_CHUNKS;  // 4096 keys per block
constexpr int SCAN_PARTS = 64, SCAN_DIGITS = 16;      // the count scan: tile ranges per digit, digits per block
// digit width RB: 9 bits (26-bit row keys sort in 3 passes), or 10 where that saves a pass (19-20, 28-30 bits)
constexpr int RBITS_MAX = 10;
constexpr int RBINS_MAX = 1 << RBITS_MAX;
static inline int digit_bits(int key_bits) {
  return (key_bits + 9) / 10 < (key_bits + 8) / 9 ? 10 : 9;
}

// Exclusive prefix over the block's BPT * SORT_THREADS values (thread t owns entries BPT*t .. BPT*t + BPT - 1).
// Contains one __syncthreads(); wtot is SORT_WAVES words of LDS that no other scan uses at the same time.
template <int BPT>
__device__ __forceinline__ void block_excl_scan(uint32_t (&v)[BPT], uint32_t* wtot, int lane, int wave) {
  uint32_t loc = 0;
#pragma unroll
  for (int u = 0; u < BPT; ++u) loc += v[u];
  uint32_t incl = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  uint32_t run = incl - loc;
  for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const uint32_t t = v[u];
    v[u] = run;
    run += t;
  }
}

// Lane mask of the run of neighbouring lanes that carry the same digit, for the run's first lane (its head);
// other lanes get head == false.  Sorted or Zipf-distributed keys put long runs of one digit into a chunk, and
// same-address LDS atomics are served one lane at a time: a run then costs one atomic instead of one per lane.
// Invalid lanes (past the end of the input: always the chunk's last lanes) join no run.
__device__ __forceinline__ unsigned long long run_mask(uint32_t d, bool valid, bool& head) {
  const int lane = threadIdx.x & 63;
  const uint32_t dd = valid ? d : 0xFFFFFFFFu;
  const uint32_t dprev = (uint32_t)__shfl_up((int)dd, 1);
  const bool first = lane == 0 || dd != dprev;
  const unsigned long long heads = __ballot(first);
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long above = heads & ~(below | (1ull << lane));   // heads of later runs
  head = first && valid;
  return ((above & (0ull - above)) - 1ull) & ~below;                   // this lane .. the lane before the next head
}

// Adds a chunk's digits to the tile histogram, one atomic per run of equal digits.
__device__ __forceinline__ void hist_add_chunk(uint32_t* hist, uint32_t d, bool valid) {
  bool head;
  const unsigned long long run = run_mask(d, valid, head);
  if (head) atomicAdd(&hist[d], (uint32_t)__popcll(run));
}

constexpr int BUILD_OFFS_LDS = 1024;   // field offsets are staged in LDS up to this many entries

// One block per sort tile: builds the tile's pairs and, while the keys are in registers, the tile's digit
// histogram of the first radix pass (counts[tile][digit], what radix_hist_kernel would write for shift 0), so the
// first pass starts at its scan.  32-bit index arithmetic: B*F < 2^32 - 4096 is required by the callers.
template <int RB>
__global__ __launch_bounds__(SORT_THREADS) void build_pairs_kernel(
    const int32_t* __restrict__ idx, const int64_t* __restrict__ row_offset, int B, int F, int N,
    int64_t total_rows, int rmode, uint2* __restrict__ pairs, uint32_t* __restrict__ counts) {
  constexpr int RBINS = 1 << RB;
  __shared__ uint32_t hist[RBINS];
  __shared__ int64_t offs[BUILD_OFFS_LDS];
  const bool offs_lds = F < BUILD_OFFS_LDS;
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) hist[d] = 0;
  if (offs_lds)
    for (int f = threadIdx.x; f <= F; f += SORT_THREADS) offs[f] = row_offset[f];
  const uint32_t n = (uint32_t)B * (uint32_t)F;
  const uint32_t base = blockIdx.x * (uint32_t)SORT_TILE;
  int x[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = base + c * SORT_THREADS + threadIdx.x;
    x[c] = i < n ? idx[i] : 0;
  }
  // (sample, field) of the thread's first item; every further item is SORT_THREADS items on
  uint32_t b = (base + threadIdx.x) / (uint32_t)F, f = (base + threadIdx.x) - b * (uint32_t)F;
  const uint32_t step_b = (uint32_t)SORT_THREADS / (uint32_t)F, step_f = (uint32_t)SORT_THREADS % (uint32_t)F;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = base + c * SORT_THREADS + threadIdx.x;
    const bool valid = i < n;
    uint32_t k = 0;
    if (valid) {
      const int64_t off = offs_lds ? offs[f] : row_offset[f];
      const int64_t V = (offs_lds ? offs[f + 1] : row_offset[f + 1]) - off;
      const bool ok = x[c] >= 0 && x[c] < V;
      uint32_t s;
      if (rmode) {
        k = ok ? (uint32_t)(off + 1) : (uint32_t)total_rows;
        s = ok ? (uint32_t)((int64_t)b * N + off + x[c]) : 0u;
      } else {
        k = ok ? (uint32_t)(off + x[c]) : (uint32_t)total_rows;
        s = b * (uint32_t)N + f;
      }
      pairs[i] = make_uint2(k, s);
    }
    hist_add_chunk(hist, k & (RBINS - 1), valid);
    b += step_b;
    f += step_f;
    if (f >= (uint32_t)F) { f -= (uint32_t)F; ++b; }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) counts[(int64_t)blockIdx.x * RBINS + d] = hist[d];
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

// counts[tile][digit] of one pass.  SPLIT: the keys are an array of their own; otherwise the .x of uint2 pairs.
template <int RB, bool SPLIT>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(
    const void* __restrict__ in, int64_t n, int shift, uint32_t* __restrict__ counts) {
  constexpr int RBINS = 1 << RB;
  __shared__ uint32_t hist[RBINS];
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) hist[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  uint32_t k[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = base + c * SORT_THREADS + threadIdx.x;
    k[c] = p < n ? (SPLIT ? ((const uint32_t*)in)[p] : ((const uint2*)in)[p].x) : 0u;
  }
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = base + c * SORT_THREADS + threadIdx.x;
    hist_add_chunk(hist, (k[c] >> shift) & (RBINS - 1), p < n);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RBINS; d += SORT_THREADS) counts[(int64_t)blockIdx.x * RBINS + d] = hist[d];
}

// Turns every column of counts[tile][digit] into its exclusive prefix over the tiles and stores the column
// totals in totals[digit]; the scatter kernel adds the prefix over the digits itself.  A block owns SCAN_DIGITS
// digits and splits the tiles into SCAN_PARTS ranges, a thread per (range, digit): up to 8 tiles per range stay in
// registers, so the kernel is one round of loads, one exchange through LDS and one round of stores.
__global__ __launch_bounds__(SCAN_PARTS * SCAN_DIGITS) void radix_scan_kernel(uint32_t* __restrict__ counts,
                                                                              int nblocks, int rbins,
                                                                              uint32_t* __restrict__ totals) {
  __shared__ uint32_t psum[SCAN_PARTS][SCAN_DIGITS + 1];
  const int dl = threadIdx.x % SCAN_DIGITS, part = threadIdx.x / SCAN_DIGITS;
  const int d = blockIdx.x * SCAN_DIGITS + dl;
  const int per = (nblocks + SCAN_PARTS - 1) / SCAN_PARTS;
  const int lo = part * per < nblocks ? part * per : nblocks, hi = lo + per < nblocks ? lo + per : nblocks;
  uint32_t* col = counts + d;
  uint32_t c[8], s = 0;
  const bool in_regs = per <= 8;
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = lo + u < hi ? col[(int64_t)(lo + u) * rbins] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += c[u];
  } else {
    int t = lo;
    for (; t + 8 <= hi; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = col[(int64_t)(t + u) * rbins];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += c[u];
    }
    for (; t < hi; ++t) s += col[(int64_t)t * rbins];
  }
  psum[part][dl] = s;
  __syncthreads();
  uint32_t run = 0, tot = 0;
#pragma unroll 8
  for (int q = 0; q < SCAN_PARTS; ++q) {
    const uint32_t v = psum[q][dl];
    if (q < part) run += v;
    tot += v;
  }
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (lo + u < hi) col[(int64_t)(lo + u) * rbins] = run;
      run += c[u];
    }
  } else {
    int t = lo;
    for (; t + 8 <= hi; t += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = col[(int64_t)(t + u) * rbins];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        col[(int64_t)(t + u) * rbins] = run;
        run += c[u];
      }
    }
    for (; t < hi; ++t) {
      const uint32_t v = col[(int64_t)t * rbins];
      col[(int64_t)t * rbins] = run;
      run += v;
    }
  }
  if (part == 0) totals[d] = tot;
}

// One pass's stable scatter.  IN_SPLIT / OUT_SPLIT: keys and values in two arrays (in_a / in_b, out_a / out_b: what
// the callers hand in and get back) instead of one uint2 per pair in in_a / out_a (between passes).
//
// Ranking inside a wave: the lanes of a 64-key chunk that carry the same digit find each other through a 64-bit
// lane mask per digit in LDS (an atomic OR of the bits of each run of neighbouring lanes with that digit, read back,
// cleared; the OR commutes, so the mask does not depend on the order the hardware serves the lanes in).  The chunk's lowest lane of a digit adds the group's
// size to the wave's running count of that digit and receives the count before it (LDS operations of a wave
// execute in program order, so chunk c is counted before chunk c + 1); the other lanes of the group fetch it
// from that lane.  A key's rank in the wave is that count plus the number of lower lanes in its group.
// The running counts are 16-bit halves of 32-bit words (a wave holds 512 keys, so no half overflows).
template <int RB, bool IN_SPLIT, bool OUT_SPLIT>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const void* __restrict__ in_a, const void* __restrict__ in_b, int64_t n, int shift,
    const uint32_t* __restrict__ scanned, const uint32_t* __restrict__ totals, void* __restrict__ out_a,
    void* __restrict__ out_b) {
  constexpr int RBINS = 1 << RB, BPT = RBINS / SORT_THREADS, MSLOTS = 512;
  static_assert(SORT_WAVES * MSLOTS == SORT_TILE, "the lane masks and the staged tile share one LDS array");
  __shared__ uint32_t whist32[SORT_WAVES][RBINS / 2];   // per wave: running digit counts, then the wave's offset in the run
  __shared__ uint32_t tstart[RBINS];                    // where digit d's run starts in the tile's digit order
  __shared__ uint32_t goff[RBINS];                      // (global position of that run) - tstart[d]
  __shared__ unsigned long long lds64[SORT_TILE];       // lane masks [wave][digit mod 512]; later the tile in digit order
  __shared__ uint32_t wtot_a[SORT_WAVES], wtot_b[SORT_WAVES];
  uint16_t* whist = (uint16_t*)&whist32[0][0];          // [wave][digit]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tbase = (int64_t)blockIdx.x * SORT_TILE;
  // the tile's pairs first: their loads overlap the prefix work below
  const int64_t wbase = tbase + (int64_t)wave * SORT_CHUNKS * 64;
  uint32_t key[SORT_CHUNKS], val[SORT_CHUNKS], rank[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    key[c] = 0u;
    val[c] = 0u;
    if (p < n) {
      if (IN_SPLIT) {
        key[c] = ((const uint32_t*)in_a)[p];
        val[c] = ((const uint32_t*)in_b)[p];
      } else {
        const uint2 kv = ((const uint2*)in_a)[p];
        key[c] = kv.x;
        val[c] = kv.y;
      }
    }
  }
  uint32_t gb[BPT], sc[BPT];   // thread t owns digits BPT*t ...: global start of the digit, this tile's offset in it
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    gb[u] = totals[BPT * threadIdx.x + u];
    sc[u] = scanned[(int64_t)blockIdx.x * RBINS + BPT * threadIdx.x + u];
  }
  for (int i = threadIdx.x; i < SORT_WAVES * RBINS / 2; i += SORT_THREADS) (&whist32[0][0])[i] = 0u;
  for (int i = threadIdx.x; i < SORT_TILE; i += SORT_THREADS) lds64[i] = 0ull;
  block_excl_scan<BPT>(gb, wtot_a, lane, wave);   // its barrier also orders the clearing above
  unsigned long long* mrow = lds64 + wave * MSLOTS;
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    const bool valid = p < n;
    const uint32_t d = (key[c] >> shift) & (RBINS - 1);
    const uint32_t slot = d & (MSLOTS - 1);
    bool head;
    const unsigned long long run = run_mask(d, valid, head);
    if (head) atomicOr(&mrow[slot], run);
    unsigned long long peers = mrow[slot];
    mrow[slot] = 0ull;
    if (RB > 9) {   // digits that share a slot differ in bit 9
      const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)d, 9, 1);   // all ones if the bit is set
      const unsigned long long m = __ballot(valid && t != 0u);
      peers &= ~(m ^ (((unsigned long long)t << 32) | t));
    }
    if (!valid) peers = 1ull << lane;
    const uint32_t in_chunk =
        __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
    const uint32_t hshift = (d & 1u) * 16u;
    uint32_t prior = 0;
    if (valid && in_chunk == 0) prior = atomicAdd(&whist32[wave][d >> 1], (uint32_t)__popcll(peers) << hshift);
    prior = (uint32_t)__shfl((int)prior, __ffsll((long long)peers) - 1);
    rank[c] = ((prior >> hshift) & 0xFFFFu) + in_chunk;
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves, the tile's count
  uint32_t ts[BPT];
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int d = BPT * threadIdx.x + u;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const uint32_t c = whist[w * RBINS + d];
      whist[w * RBINS + d] = (uint16_t)run;
      run += c;
    }
    ts[u] = run;
  }
  block_excl_scan<BPT>(ts, wtot_b, lane, wave);
#pragma unroll
  for (int u = 0; u < BPT; ++u) {
    const int d = BPT * threadIdx.x + u;
    tstart[d] = ts[u];
    goff[d] = gb[u] + sc[u] - ts[u];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const int64_t p = wbase + c * 64 + lane;
    if (p < n) {
      const uint32_t d = (key[c] >> shift) & (RBINS - 1);
      lds64[tstart[d] + whist[wave * RBINS + d] + rank[c]] = (unsigned long long)key[c] | ((unsigned long long)val[c] << 32);
    }
  }
  __syncthreads();
  const int64_t left = n - tbase;
  const uint32_t tile_n = left < SORT_TILE ? (uint32_t)left : (uint32_t)SORT_TILE;
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = c * SORT_THREADS + threadIdx.x;
    if (i < tile_n) {
      const unsigned long long kv = lds64[i];
      const uint32_t k = (uint32_t)kv;
      const uint32_t dst = goff[(k >> shift) & (RBINS - 1)] + i;
      if (OUT_SPLIT) {
        ((uint32_t*)out_a)[dst] = k;
        ((uint32_t*)out_b)[dst] = (uint32_t)(kv >> 32);
      } else {
        ((unsigned long long*)out_a)[dst] = kv;
      }
    }
  }
}

static int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 32 && (max_value >> b) != 0) ++b;
  return b;
}

struct SortWs {
  uint2 *pa, *pb;              // the pairs between passes (ping-pong)
  uint32_t *counts, *totals;   // counts[tile][digit], totals[digit]
  int nblocks;
};

static size_t sort_ws_bytes(int64_t n) {
  const int64_t nblocks = cdiv(n > 0 ? n : 1, SORT_TILE);
  return align_up((size_t)n * 8, 256) * 2 + align_up((size_t)nblocks * RBINS_MAX * 4, 256) + RBINS_MAX * 4;
}

static SortWs carve(void* ws, int64_t n) {
  SortWs w;
  char* p = (char*)ws;
  w.nblocks = (int)cdiv(n > 0 ? n : 1, SORT_TILE);
  w.pa = (uint2*)p; p += align_up((size_t)n * 8, 256);
  w.pb = (uint2*)p; p += align_up((size_t)n * 8, 256);
  w.counts = (uint32_t*)p; p += align_up((size_t)w.nblocks * RBINS_MAX * 4, 256);
  w.totals = (uint32_t*)p;
  return w;
}

template <int RB>
static void launch_pass(const void* ia, const void* ib, bool in_split, void* oa, void* ob, bool out_split, int64_t n,
                        int shift, bool need_hist, const SortWs& w, hipStream_t st) {
  const dim3 grid(w.nblocks), block(SORT_THREADS);
  if (need_hist) {
    if (in_split)
      hipLaunchKernelGGL((radix_hist_kernel<RB, true>), grid, block, 0, st, ia, n, shift, w.counts);
    else
      hipLaunchKernelGGL((radix_hist_kernel<RB, false>), grid, block, 0, st, ia, n, shift, w.counts);
  }
  hipLaunchKernelGGL(radix_scan_kernel, dim3((1 << RB) / SCAN_DIGITS), dim3(SCAN_PARTS * SCAN_DIGITS), 0, st, w.counts,
                     w.nblocks, 1 << RB, w.totals);
  if (in_split && out_split)
    hipLaunchKernelGGL((radix_scatter_kernel<RB, true, true>), grid, block, 0, st, ia, ib, n, shift, w.counts, w.totals,
                       oa, ob);
  else if (in_split)
    hipLaunchKernelGGL((radix_scatter_kernel<RB, true, false>), grid, block, 0, st, ia, ib, n, shift, w.counts,
                       w.totals, oa, ob);
  else if (out_split)
    hipLaunchKernelGGL((radix_scatter_kernel<RB, false, true>), grid, block, 0, st, ia, ib, n, shift, w.counts,
                       w.totals, oa, ob);
  else
    hipLaunchKernelGGL((radix_scatter_kernel<RB, false, false>), grid, block, 0, st, ia, ib, n, shift, w.counts,
                       w.totals, oa, ob);
}

// Sorts n pairs into (keys_out, vals_out).  The input is (keys_in, vals_in), or, with keys_in == nullptr, the
// uint2 pairs already in w.pa (first_hist_done: w.counts then holds the first pass's tile histograms).
// (keys_in, vals_in) may be (keys_out, vals_out) when there are at least two passes.
static int run_passes(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                      int64_t n, int passes, int rb, const SortWs& w, hipStream_t st, bool first_hist_done = false) {
  const void *ia = keys_in ? (const void*)keys_in : (const void*)w.pa, *ib = vals_in;
  bool in_split = keys_in != nullptr;
  uint2* next = keys_in ? w.pa : w.pb;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    void* oa = last ? (void*)keys_out : (void*)next;
    void* ob = last ? (void*)vals_out : nullptr;
    const bool need_hist = p > 0 || !first_hist_done;
    if (rb == 10)
      launch_pass<10>(ia, ib, in_split, oa, ob, last, n, rb * p, need_hist, w, st);
    else
      launch_pass<9>(ia, ib, in_split, oa, ob, last, n, rb * p, need_hist, w, st);
    ia = next;
    ib = nullptr;
    in_split = false;
    next = next == w.pa ? w.pb : w.pa;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// Field-segmented pair sort of the embedding backward.  A key is (field, id): the builder writes the pairs
// field-major (field f's pairs at [f*B, (f+1)*B) in sample order), which sorts them by field for free, and the radix
// passes then run inside each field on the bits ITS ids need: ceil(bits(V_f) / 9) passes of equal width <= 9 bits -
// one pass for a table of up to 511 rows, three for ten million - instead of three passes over every pair for
// 26-bit global row numbers.  A field is a segment of Tf = ceil(B / 4096) tiles, so a scatter block finds its
// offsets from the Tf digit counts of its own field: no scan launch.  Two forms of a pass:
//   Tf <= FS_FUSED_MAX_TF (B <= 16384): ONE launch.  A block ranks its tile, publishes the tile's digit counts (one
//     word per digit with a "written" bit, so a word is its own flag), and reads the counts of the other Tf - 1
//     tiles of its field as they appear - no read-modify-write on global memory (a device-scope atomic is one fabric
//     transaction: a per-tile flush of digit counts by integer atomics was measured at +10 us per launch).  Blocks
//     wait only for blocks of their own field, at most three block indices away, so with workgroups dispatched in
//     index order the wait ends as soon as the field's blocks have ranked their tiles; a wait that lasts
//     FS_WAIT_TICKS stops the kernel instead of hanging it.  The count words of pass p live in region p & 1; the
//     builder clears region 0 and a pass's block clears its own slot of the other region for the next pass, so
//     nothing has to be cleared between calls.
//   more tiles: a histogram launch, then the scatter, which requests the finished counts together with its keys.
//     A field's 16 blocks are dispatched at different times when other kernels share the GPU, and blocks that wait
//     hold their CU meanwhile: inside the DeepFM step at B = 65536 waiting passes made the step 1.7 % slower although
//     the sort alone is 4 us faster with them.
// Blocks of fields whose passes are done return at once; a field's last pass writes the final split arrays with
// key = row_offset[f] + id.  Ids outside their table travel as id = V_f (behind every valid id of the field); the
// last pass moves them behind all fields with key = total_rows, ordered by field.
// ---------------------------------------------------------------------------------------
constexpr int FS_BINS = 512;
constexpr int FS_MAX_TF = 32;            // tiles per field (B <= 131072): a scatter thread holds one count per tile
constexpr int FS_SB = 128;               // samples per builder block
constexpr int FS_FG = 32;                // fields per builder block
constexpr int FS_FUSED_MAX_TF = 4;          // tiles per field up to which a pass is one launch
constexpr uint32_t FS_WRITTEN = 0x80000000u;             // a published count word (counts are <= 4096)
constexpr unsigned long long FS_WAIT_TICKS = 200000000ull;   // 2 s of the 100 MHz wall clock

// What a field's passes look like: ids 0 .. V-1 are valid, Vs is the id an invalid lookup travels with.
// R-mode: every valid pair of a field has the same key, so the id is 0 (valid) or 1.
__device__ __forceinline__ void fs_plan(int64_t V, int rmode, uint32_t& Vs, int& passes, int& rb) {
  Vs = rmode ? 1u : (uint32_t)V;
  const int bits = Vs == 0u ? 1 : 32 - __builtin_clz(Vs);
  passes = (bits + 8) / 9;
  rb = (bits + passes - 1) / passes;
}

// grid (ceil(B / FS_SB), ceil(F / FS_FG)): a block turns its [samples][fields] slab of ids into pairs and writes
// every field's 128 pairs as one contiguous piece.  All of a thread's id loads are in flight together.  It also
// counts the slab's invalid lookups per field (inv_fine[f][blockIdx.x]) and clears its share of count region 0
// (count_words words).
__global__ __launch_bounds__(256) void fs_build_kernel(const int32_t* __restrict__ idx,
                                                       const int64_t* __restrict__ row_offset, int B, int F, int N,
                                                       int rmode, uint2* __restrict__ pairs,
                                                       uint32_t* __restrict__ inv_fine, uint32_t* __restrict__ counts0,
                                                       int64_t count_words) {
  __shared__ int32_t ids[FS_SB * (FS_FG + 1)];      // [sample][field], odd pitch: a wave reads one field of 64 samples
  __shared__ int64_t offs[FS_FG + 1];
  __shared__ uint32_t ninv[FS_FG];
  constexpr int PER = FS_SB * FS_FG / 256;          // 16 ids per thread
  const int f0 = blockIdx.y * FS_FG, fgc = F - f0 < FS_FG ? F - f0 : FS_FG;
  const int b0 = blockIdx.x * FS_SB, sb = B - b0 < FS_SB ? B - b0 : FS_SB;
  const int lf = threadIdx.x & (FS_FG - 1), ls = threadIdx.x / FS_FG;      // field of the slab, first sample
  int32_t v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int s = ls + u * (256 / FS_FG);
    v[u] = (lf < fgc && s < sb) ? idx[(int64_t)(b0 + s) * F + f0 + lf] : 0;
  }
  for (int i = threadIdx.x; i <= fgc; i += 256) offs[i] = row_offset[f0 + i];
  if (threadIdx.x < FS_FG) ninv[threadIdx.x] = 0u;
  {
    const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * 256;
    const int64_t me = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    for (int64_t q = me; q * 4 < count_words; q += nthreads) ((uint4*)counts0)[q] = make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) ids[(ls + u * (256 / FS_FG)) * (FS_FG + 1) + lf] = v[u];
  __syncthreads();
#pragma unroll 4
  for (int u = 0; u < PER; ++u) {
    const int j = threadIdx.x + 256 * u;
    const int ff = j / FS_SB, s = j % FS_SB;
    if (ff >= fgc || s >= sb) continue;
    const int f = f0 + ff;
    const uint32_t b = (uint32_t)(b0 + s);
    const int x = ids[s * (FS_FG + 1) + ff];
    const int64_t off = offs[ff], V = offs[ff + 1] - off;
    const bool ok = x >= 0 && x < V;
    if (!ok) atomicAdd(&ninv[ff], 1u);
    const uint32_t sid = ok ? (rmode ? 0u : (uint32_t)x) : (rmode ? 1u : (uint32_t)V);
    const uint32_t src = rmode ? (ok ? (uint32_t)((int64_t)b * N + off + x) : 0u) : b * (uint32_t)N + (uint32_t)f;
    pairs[(uint32_t)f * (uint32_t)B + b] = make_uint2(sid, src);
  }
  __syncthreads();
  if ((int)threadIdx.x < fgc) inv_fine[(int64_t)(f0 + threadIdx.x) * gridDim.x + blockIdx.x] = ninv[threadIdx.x];
}

// counts[f][t][digit] of one pass for the fields that still have one; block (f, t) = blockIdx.x; the first pass also
// counts the tile's invalid lookups into inv_tile[f][t].  Used for batches of more than FS_FUSED_MAX_TF tiles per
// field, where the scatter reads finished counts instead of waiting for them.
__global__ __launch_bounds__(SORT_THREADS) void fs_hist_kernel(const uint2* __restrict__ in,
                                                              const int64_t* __restrict__ row_offset, int B, int Tf,
                                                              int pass, int rmode, uint32_t* __restrict__ counts,
                                                              uint32_t* __restrict__ inv_tile) {
  __shared__ uint32_t hist[FS_BINS];
  __shared__ uint32_t ninv;
  const int f = blockIdx.x / Tf, t = blockIdx.x - f * Tf;
  uint32_t Vs;
  int passes, rb;
  fs_plan(row_offset[f + 1] - row_offset[f], rmode, Vs, passes, rb);
  if (pass >= passes) return;
  const int shift = pass * rb;
  const uint32_t mask = (1u << rb) - 1u;
  const int left = B - t * SORT_TILE;
  const uint32_t cnt = left < SORT_TILE ? (uint32_t)left : (uint32_t)SORT_TILE;
  const uint2* tile = in + ((uint32_t)f * (uint32_t)B + (uint32_t)t * SORT_TILE);
  uint32_t k[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = c * SORT_THREADS + threadIdx.x;
    k[c] = i < cnt ? tile[i].x : 0u;
  }
  hist[threadIdx.x] = 0u;
  if (threadIdx.x == 0) ninv = 0u;
  __syncthreads();
  uint32_t bad = 0u;
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const bool valid = c * SORT_THREADS + threadIdx.x < cnt;
    hist_add_chunk(hist, (k[c] >> shift) & mask, valid);
    bad += (uint32_t)__popcll(__ballot(valid && k[c] == Vs));
  }
  if (pass == 0 && bad != 0u && (threadIdx.x & 63) == 0) atomicAdd(&ninv, bad);
  __syncthreads();
  counts[(int64_t)blockIdx.x * FS_BINS + threadIdx.x] = hist[threadIdx.x];
  if (pass == 0 && threadIdx.x == 0) inv_tile[blockIdx.x] = ninv;
}

// One pass's stable scatter inside the fields: block (f, t) = blockIdx.x moves tile t of field f.  in / out: the uint2
// pairs between passes; keys_out / src_out: the final arrays (written by a field's last pass); counts / counts_next:
// this pass's and the next pass's [F][Tf][512] count words; inv_fine: [F][nb] invalid lookups per builder block
// (nb = ceil(B / 128)); inv_tile: [F][Tf] the same per tile, written by pass 0 for the later launches.  The ranking
// inside the tile is that of radix_scatter_kernel.
template <bool FUSED>
__global__ __launch_bounds__(SORT_THREADS) void fs_scatter_kernel(
    const uint2* __restrict__ in, uint2* __restrict__ out, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ src_out, const int64_t* __restrict__ row_offset, int B, int F, int Tf, int pass, int rmode,
    int64_t total_rows, uint32_t* counts, uint32_t* __restrict__ counts_next, const uint32_t* __restrict__ inv_fine,
    int nb, uint32_t* __restrict__ inv_tile) {
  constexpr int RBINS = FS_BINS, MSLOTS = 512;
  constexpr int FINE_PER_TILE = SORT_TILE / FS_SB;
  __shared__ uint32_t whist32[SORT_WAVES][RBINS / 2];
  __shared__ uint32_t tstart[RBINS];
  __shared__ uint32_t goff[RBINS];                       // (position in the field of digit d's run in this tile) - tstart[d]
  __shared__ unsigned long long lds64[SORT_TILE];
  __shared__ uint32_t wtot_a[SORT_WAVES], wtot_b[SORT_WAVES];
  __shared__ uint32_t inv_red[3][SORT_WAVES];
  uint16_t* whist = (uint16_t*)&whist32[0][0];
  const int f = blockIdx.x / Tf, t = blockIdx.x - f * Tf;
  const int64_t off = row_offset[f], V = row_offset[f + 1] - off;
  uint32_t Vs;
  int passes, rb;
  fs_plan(V, rmode, Vs, passes, rb);
  if (pass >= passes) return;                            // the whole block: this field is in its final place
  const bool last = pass == passes - 1;
  const int shift = pass * rb;
  const uint32_t mask = (1u << rb) - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int left = B - t * SORT_TILE;
  const uint32_t cnt = left < SORT_TILE ? (uint32_t)left : (uint32_t)SORT_TILE;
  const uint32_t fbase = (uint32_t)f * (uint32_t)B, tbase = fbase + (uint32_t)t * SORT_TILE;
  uint32_t key[SORT_CHUNKS], val[SORT_CHUNKS], rank[SORT_CHUNKS];
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = wave * (SORT_CHUNKS * 64) + c * 64 + lane;
    key[c] = 0u;
    val[c] = 0u;
    if (i < cnt) {
      const uint2 kv = in[tbase + i];
      key[c] = kv.x;
      val[c] = kv.y;
    }
  }
  // !FUSED: a histogram launch finished the counts before this one; they are requested with the keys (digit d =
  // threadIdx.x: the field's count over its Tf tiles and the count in the tiles before this one)
  uint32_t gb[1] = {0u}, sc = 0u;
  if (!FUSED) {
    const uint32_t* ccol = counts + (int64_t)f * Tf * RBINS + threadIdx.x;
    uint32_t cc[FS_MAX_TF];
#pragma unroll
    for (int tt = 0; tt < FS_MAX_TF; ++tt) cc[tt] = tt < Tf ? ccol[(int64_t)tt * RBINS] : 0u;
#pragma unroll
    for (int tt = 0; tt < FS_MAX_TF; ++tt) {
      gb[0] += cc[tt];
      if (tt < t) sc += cc[tt];
    }
  }
  // invalid lookups (last pass only): in the fields before this one, in this field, in all.  A fused pass 0 reads the
  // builder's counts (the other tiles' sums are being written by this very launch); every other pass reads the
  // per-tile sums an earlier launch wrote.
  uint32_t inv_before = 0u, inv_f = 0u, inv_total = 0u;
  if (last) {
    const bool fine = FUSED && pass == 0;
    const uint32_t* src = fine ? inv_fine : inv_tile;
    const int per = fine ? nb : Tf;
    for (int g = threadIdx.x; g < F * per; g += SORT_THREADS) {
      const uint32_t c = src[g];
      const int gf = g / per;
      inv_total += c;
      if (gf < f) inv_before += c;
      if (gf == f) inv_f += c;
    }
  }
  if (FUSED && pass == 0 && wave == 0) {
    const int g = t * FINE_PER_TILE + lane;
    uint32_t c = (lane < FINE_PER_TILE && g < nb) ? inv_fine[(int64_t)f * nb + g] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if (lane == 0) inv_tile[blockIdx.x] = c;
  }
  if (FUSED && !last) counts_next[(int64_t)blockIdx.x * RBINS + threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < SORT_WAVES * RBINS / 2; i += SORT_THREADS) (&whist32[0][0])[i] = 0u;
  for (int i = threadIdx.x; i < SORT_TILE; i += SORT_THREADS) lds64[i] = 0ull;
  if (last) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      inv_total += (uint32_t)__shfl_xor((int)inv_total, o);
      inv_before += (uint32_t)__shfl_xor((int)inv_before, o);
      inv_f += (uint32_t)__shfl_xor((int)inv_f, o);
    }
    if (lane == 0) { inv_red[0][wave] = inv_total; inv_red[1][wave] = inv_before; inv_red[2][wave] = inv_f; }
  }
  if (!FUSED) block_excl_scan<1>(gb, wtot_a, lane, wave);  // its barrier also orders the clearing above
  else __syncthreads();
  if (last) {
    inv_total = inv_before = inv_f = 0u;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) { inv_total += inv_red[0][w]; inv_before += inv_red[1][w]; inv_f += inv_red[2][w]; }
  }
  unsigned long long* mrow = lds64 + wave * MSLOTS;
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = wave * (SORT_CHUNKS * 64) + c * 64 + lane;
    const bool valid = i < cnt;
    const uint32_t d = (key[c] >> shift) & mask;
    bool head;
    const unsigned long long run = run_mask(d, valid, head);
    if (head) atomicOr(&mrow[d], run);
    unsigned long long peers = mrow[d];
    mrow[d] = 0ull;
    if (!valid) peers = 1ull << lane;
    const uint32_t in_chunk =
        __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
    const uint32_t hshift = (d & 1u) * 16u;
    uint32_t prior = 0;
    if (valid && in_chunk == 0) prior = atomicAdd(&whist32[wave][d >> 1], (uint32_t)__popcll(peers) << hshift);
    prior = (uint32_t)__shfl((int)prior, __ffsll((long long)peers) - 1);
    rank[c] = ((prior >> hshift) & 0xFFFFu) + in_chunk;
  }
  __syncthreads();
  // digit d = threadIdx.x: the tile's count, published at once for the field's other tiles
  uint32_t ts[1];
  {
    const int d = threadIdx.x;
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const uint32_t c = whist[w * RBINS + d];
      whist[w * RBINS + d] = (uint16_t)run;
      run += c;
    }
    ts[0] = run;
  }
  uint32_t* col = counts + (int64_t)f * Tf * RBINS + threadIdx.x;
  const uint32_t mine = ts[0];
  if (FUSED) __hip_atomic_store(col + (int64_t)t * RBINS, mine | FS_WRITTEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  block_excl_scan<1>(ts, wtot_b, lane, wave);
  tstart[threadIdx.x] = ts[0];
  if (!FUSED) goff[threadIdx.x] = gb[0] + sc - ts[0];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = wave * (SORT_CHUNKS * 64) + c * 64 + lane;
    if (i < cnt) {
      const uint32_t d = (key[c] >> shift) & mask;
      lds64[tstart[d] + whist[wave * RBINS + d] + rank[c]] = (unsigned long long)key[c] | ((unsigned long long)val[c] << 32);
    }
  }
  if (FUSED) {
    // the field's count of digit d (over its Tf tiles) and the count in the tiles before this one, as the other
    // tiles publish theirs; every outstanding word is requested in each round
    bool gave_up = false;
    uint32_t tot = mine;
    uint32_t pending = (Tf >= 32 ? 0xFFFFFFFFu : (1u << Tf) - 1u) & ~(1u << t);
    const unsigned long long t0 = wall_clock64();
    while (__any(pending != 0u)) {
      uint32_t got[FS_MAX_TF];
#pragma unroll
      for (int tt = 0; tt < FS_MAX_TF; ++tt)
        got[tt] = (pending >> tt) & 1u
                      ? __hip_atomic_load(col + (int64_t)tt * RBINS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : 0u;
#pragma unroll
      for (int tt = 0; tt < FS_MAX_TF; ++tt)
        if (got[tt] & FS_WRITTEN) {
          const uint32_t c = got[tt] & ~FS_WRITTEN;
          tot += c;
          if (tt < t) sc += c;
          pending &= ~(1u << tt);
        }
      if (wall_clock64() - t0 > FS_WAIT_TICKS) {
        gave_up = true;
        break;
      }
      if (pending != 0u) __builtin_amdgcn_s_sleep(8);
    }
    gb[0] = tot;
    if (__syncthreads_or(gave_up)) __builtin_trap();      // a tile of this field never arrived: stop, do not hang
    block_excl_scan<1>(gb, wtot_a, lane, wave);
    goff[threadIdx.x] = gb[0] + sc - ts[0];
  }
  __syncthreads();
  if (!last) {
#pragma unroll
    for (int c = 0; c < SORT_CHUNKS; ++c) {
      const uint32_t i = c * SORT_THREADS + threadIdx.x;
      if (i < cnt) {
        const unsigned long long kv = lds64[i];
        const uint32_t pos = goff[((uint32_t)kv >> shift) & mask] + i;            // position in the field
        ((unsigned long long*)out)[fbase + pos] = kv;
      }
    }
    return;
  }
  // last pass of this field: final keys, invalid lookups behind all fields
  const uint32_t n = (uint32_t)B * (uint32_t)F;
  const uint32_t key_base = (uint32_t)off + (rmode ? 1u : 0u);
#pragma unroll
  for (int c = 0; c < SORT_CHUNKS; ++c) {
    const uint32_t i = c * SORT_THREADS + threadIdx.x;
    if (i < cnt) {
      const unsigned long long kv = lds64[i];
      const uint32_t k = (uint32_t)kv;
      const uint32_t pos = goff[(k >> shift) & mask] + i;
      const bool good = k != Vs;
      const uint32_t dst = good ? fbase - inv_before + pos : n - inv_total + inv_before + (pos - ((uint32_t)B - inv_f));
      keys_out[dst] = good ? key_base + (rmode ? 0u : k) : (uint32_t)total_rows;
      src_out[dst] = (uint32_t)(kv >> 32);
    }
  }
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
  const int rb = digit_bits(32), passes = (32 + rb - 1) / rb;   // four passes: sorting in place is allowed
  hipLaunchKernelGGL(auc_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, st, score, label, n, keys, vals);
  run_passes(keys, vals, keys, vals, n, passes, rb, w, st);
  const int nb = (int)(cdiv(n, 256) < AUC_BLOCKS ? cdiv(n, 256) : AUC_BLOCKS);
  hipLaunchKernelGGL(auc_partial_kernel, dim3(nb), dim3(256), 0, st, keys, vals, n, score, label, part);
  hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(1), 0, st, part, nb, n, auc_out, logloss_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" size_t ctr_emb_sort_workspace_bytes(int64_t n_items) { return sort_ws_bytes(n_items); }

// The field-segmented path: batches of at least 1024 samples (below that a field's tile is mostly empty and the
// global sort has fewer blocks) and at most FS_MAX_TF tiles per field.  CTR_SORT_FIELDS=0 keeps the global sort.
static bool fs_eligible(int64_t B, int64_t F) {
  static const bool off = getenv("CTR_SORT_FIELDS") && getenv("CTR_SORT_FIELDS")[0] == '0';
  return !off && F >= 1 && B >= 1024 && cdiv(B, SORT_TILE) <= FS_MAX_TF;
}
static int fs_max_passes(int64_t total_rows, int rmode) { return rmode ? 1 : (bits_for((uint64_t)total_rows) + 8) / 9; }
static size_t fs_counter_bytes(int64_t B, int64_t F) {
  // two regions of count words, the builder's invalid counts, the per-tile invalid counts
  return align_up(((size_t)F * cdiv(B, SORT_TILE) * (2 * FS_BINS + 1) + (size_t)F * cdiv(B, FS_SB)) * 4, 256);
}
static size_t fs_ws_bytes(int64_t B, int64_t F) { return 2 * align_up((size_t)(B * F) * 8, 256) + fs_counter_bytes(B, F); }

extern "C" size_t ctr_emb_sort_workspace_bytes_fields(int B, int F, int64_t total_rows, int rmode) {
  const int64_t n = (int64_t)B * F;
  size_t need = sort_ws_bytes(n);
  if (fs_eligible(B, F) && fs_ws_bytes(B, F) > need) need = fs_ws_bytes(B, F);
  return need;
}

static void fs_sort(const int32_t* idx, const int64_t* row_offset, int B, int F, int N, int64_t total_rows, int rmode,
                    uint32_t* keys_out, uint32_t* src_out, void* workspace, hipStream_t st) {
  const int64_t n = (int64_t)B * F;
  const int Tf = (int)cdiv(B, SORT_TILE), pmax = fs_max_passes(total_rows, rmode);
  char* p = (char*)workspace;
  uint2* pa = (uint2*)p; p += align_up((size_t)n * 8, 256);
  uint2* pb = (uint2*)p; p += align_up((size_t)n * 8, 256);
  const int nb = (int)cdiv(B, FS_SB);
  const int64_t words = (int64_t)F * Tf * FS_BINS;          // one region of count words: [F][Tf][512], 16-byte multiple
  uint32_t* counts[2] = {(uint32_t*)p, (uint32_t*)p + words};
  uint32_t* inv_tile = counts[1] + words;                  // [F][Tf]
  uint32_t* inv_fine = inv_tile + (size_t)F * Tf;          // [F][nb]
  // one launch per pass while a field has few tiles (its blocks are dispatched together, and the launches are what
  // the sort costs); with more tiles a histogram launch per pass and no waiting inside the scatter: beside other
  // kernels a field's blocks become resident at different times, and blocks that wait hold their CU meanwhile
  // (measured inside the DeepFM step at B = 65536: the step 1 % slower with waiting passes, although the sort alone
  // is 10 us faster).  CTR_FS_FUSED_MAX_TF moves the border (measurement switch).
  static const int fused_max_tf = getenv("CTR_FS_FUSED_MAX_TF") ? atoi(getenv("CTR_FS_FUSED_MAX_TF")) : FS_FUSED_MAX_TF;
  const bool fused = Tf <= fused_max_tf;
  hipLaunchKernelGGL(fs_build_kernel, dim3((unsigned)nb, (unsigned)cdiv(F, FS_FG)), dim3(256), 0, st, idx, row_offset, B,
                     F, N, rmode, pa, inv_fine, counts[0], fused ? words : (int64_t)0);
  for (int ps = 0; ps < pmax; ++ps) {
    const uint2* in = (ps & 1) ? pb : pa;
    uint2* out = (ps & 1) ? pa : pb;
    uint32_t* cur = fused ? counts[ps & 1] : counts[0];
    if (fused) {
      hipLaunchKernelGGL(fs_scatter_kernel<true>, dim3((unsigned)(F * Tf)), dim3(SORT_THREADS), 0, st, in, out, keys_out,
                         src_out, row_offset, B, F, Tf, ps, rmode, total_rows, cur, counts[(ps + 1) & 1], inv_fine, nb,
                         inv_tile);
    } else {
      hipLaunchKernelGGL(fs_hist_kernel, dim3((unsigned)(F * Tf)), dim3(SORT_THREADS), 0, st, in, row_offset, B, Tf, ps,
                         rmode, cur, inv_tile);
      hipLaunchKernelGGL(fs_scatter_kernel<false>, dim3((unsigned)(F * Tf)), dim3(SORT_THREADS), 0, st, in, out, keys_out,
                         src_out, row_offset, B, F, Tf, ps, rmode, total_rows, cur, counts[(ps + 1) & 1], inv_fine, nb,
                         inv_tile);
    }
  }
}

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
  if (fs_eligible(B, F) && (int64_t)F * cdiv(B, SORT_TILE) < 0x7FFFFFFFll &&
      workspace_bytes >= fs_ws_bytes(B, F)) {
    fs_sort(idx, row_offset, B, F, N, total_rows, rmode, keys_out, src_out, workspace, st);
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  const SortWs w = carve(workspace, n);
  const int kbits = bits_for((uint64_t)total_rows), rb = digit_bits(kbits), passes = (kbits + rb - 1) / rb;
  if (rb == 10)
    hipLaunchKernelGGL(build_pairs_kernel<10>, dim3((unsigned)w.nblocks), dim3(SORT_THREADS), 0, st, idx, row_offset, B,
                       F, N, total_rows, rmode, w.pa, w.counts);
  else
    hipLaunchKernelGGL(build_pairs_kernel<9>, dim3((unsigned)w.nblocks), dim3(SORT_THREADS), 0, st, idx, row_offset, B,
                       F, N, total_rows, rmode, w.pa, w.counts);
  run_passes(nullptr, nullptr, keys_out, src_out, n, passes, rb, w, st, true);
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
  // the builder's split pairs go where the result will: with one pass the sort would read and write the same
  // arrays, so that case builds into the second ping-pong buffer instead (unused by a one-pass sort)
  uint32_t* bk = passes > 1 ? keys_out : (uint32_t*)w.pb;
  uint32_t* bv = passes > 1 ? src_out : (uint32_t*)w.pb + nnz;
  const int64_t nbags = (int64_t)B * F;
  const int64_t blocks = cdiv(nbags, 256) < 2048 ? cdiv(nbags, 256) : 2048;
  hipLaunchKernelGGL(build_bag_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, idx, bag_offsets, row_offset,
                     B, F, N, total_rows, bk, bv);
  run_passes(bk, bv, keys_out, src_out, nnz, passes, rb, w, st);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// ---------------------------------------------------------------------------------------
// The pair sort with its scratch in a file-scope device array instead of a caller-provided workspace, for
// callers that cannot allocate device memory.  Each GPU holds its own copy of the array (at its own address)
// from the moment the library is loaded on it until the process ends.
// ---------------------------------------------------------------------------------------
constexpr size_t SORT_STATIC_BYTES = (size_t)36 << 20;
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
  CTR_REQUIRE(passes > 1 || (keys_in != keys_out && vals_in != vals_out),
              "ctr_sort_pairs: a one-pass sort cannot work in place");
  run_passes(keys_in, vals_in, keys_out, vals_out, n, passes, rb, w, st);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
