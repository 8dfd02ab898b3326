// Row-wise sharding helpers of the multi-GPU embedding exchange: route every lookup to the rank that
// owns its table row, and move E-float rows between the strided [B, ld] activation layout and the
// packed send/receive buffers in the order a stable sort by destination rank produced.
#include "common.h"

namespace ctr {

// lookup p = b*F + f with id x:  dest = (x + f) mod W,  local key = local_offset[dest][f] + x / W
// (local_offset[j][f] = first row of field f inside rank j's concatenated shard).  The owner is rotated by the
// field number so that the hottest row of every field (often the same small id) does not land on one rank.  An id
// outside [0, V_f) keeps the zero-row semantics of the single-GPU gather: it goes to rank f mod W with key -1 and is
// counted in oor_count (optional) - here, on the rank whose batch holds it, because the owner cannot tell such a
// slot from an unfilled one.
__global__ __launch_bounds__(256) void shard_route_kernel(const int32_t* __restrict__ idx,
                                                         const int64_t* __restrict__ row_offset,
                                                         const int32_t* __restrict__ local_offset, int64_t n, int F,
                                                         int W, uint32_t* __restrict__ dest,
                                                         int32_t* __restrict__ lkey, int32_t* __restrict__ oor_count) {
  int bad = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(p % F);
    const int x = idx[p];
    const int64_t V = row_offset[f + 1] - row_offset[f];
    if (x >= 0 && x < V) {
      const int d = (int)(((int64_t)x + f) % W);
      dest[p] = (uint32_t)d;
      lkey[p] = local_offset[d * F + f] + x / W;
    } else {
      dest[p] = (uint32_t)(f % W);
      lkey[p] = -1;
      ++bad;
    }
  }
  if (bad && oor_count) atomicAdd(oor_count, bad);
}

constexpr uint32_t EMPTY_SLOT = 0xFFFFFFFFu;   // a slot of a fixed-capacity bucket that no lookup fills

// Fixed-capacity send buckets.  dest_sorted / perm: the lookups stably sorted by destination rank.  Lookup number r
// of destination d goes to slot d*cap + r: slot_key = its key inside the owner's shard, slot_perm = its position
// p = b*F + f in the batch.  Lookups beyond a bucket's capacity are dropped and counted in overflow (the caller
// treats a non-zero count as an error); with z_zero set, the E floats such a lookup owns in the [B, z_ld] activation
// buffer are zeroed, so that no row of an earlier batch stands in for it.  Both outputs were filled with 0xFF bytes
// beforehand (empty slots).
__global__ __launch_bounds__(256) void shard_bucket_kernel(const uint32_t* __restrict__ dest_sorted,
                                                          const uint32_t* __restrict__ perm,
                                                          const int32_t* __restrict__ lkey, int64_t n, int W, int64_t cap,
                                                          int32_t* __restrict__ slot_key, uint32_t* __restrict__ slot_perm,
                                                          int32_t* __restrict__ overflow, float* __restrict__ z_zero,
                                                          int64_t z_ld, int F, int E) {
  __shared__ int64_t start[64];        // first sorted position of every destination
  if ((int)threadIdx.x < W) {
    const uint32_t d = threadIdx.x;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (dest_sorted[mid] < d) lo = mid + 1; else hi = mid;
    }
    start[d] = lo;
  }
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t d = dest_sorted[j];
    const int64_t r = j - start[d];
    if (r < cap) {
      const uint32_t p = perm[j];
      slot_key[d * cap + r] = lkey[p];
      slot_perm[d * cap + r] = p;
    } else {
      atomicAdd(overflow, 1);
      if (z_zero) {
        const uint32_t p = perm[j];
        float* row = z_zero + (int64_t)(p / (uint32_t)F) * z_ld + (int64_t)(p % (uint32_t)F) * E;
        for (int e = 0; e < E; ++e) row[e] = 0.f;
      }
    }
  }
}

// out[s, 0:E] = src[b, f*E : (f+1)*E] with (b, f) = divmod(perm[s], F) (zeros where perm[s] marks an empty slot);
// 32-bit words, any payload
__global__ __launch_bounds__(256) void rows_gather_kernel(const uint32_t* __restrict__ src, int64_t ld,
                                                         const uint32_t* __restrict__ perm, int64_t n, int F, int E,
                                                         uint32_t* __restrict__ out) {
  const int64_t total = n * E;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = q / E;
    const int e = (int)(q % E);
    const uint32_t p = perm[s];
    out[q] = p == EMPTY_SLOT ? 0u : src[(int64_t)(p / (uint32_t)F) * ld + (int64_t)(p % (uint32_t)F) * E + e];
  }
}

// dst[b, f*E : (f+1)*E] = rows[s, 0:E] with (b, f) = divmod(perm[s], F); empty slots are skipped
__global__ __launch_bounds__(256) void rows_scatter_kernel(const uint32_t* __restrict__ rows,
                                                          const uint32_t* __restrict__ perm, int64_t n, int F, int E,
                                                          uint32_t* __restrict__ dst, int64_t ld) {
  const int64_t total = n * E;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = q / E;
    const int e = (int)(q % E);
    const uint32_t p = perm[s];
    if (p != EMPTY_SLOT) dst[(int64_t)(p / (uint32_t)F) * ld + (int64_t)(p % (uint32_t)F) * E + e] = rows[q];
  }
}

static unsigned grid_of(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (unsigned)(b < 8192 ? (b > 0 ? b : 1) : 8192);
}

}  // namespace ctr

using namespace ctr;

extern "C" int ctr_shard_route(const int32_t* idx, const int64_t* row_offset, const int32_t* local_offset, int B,
                               int F, int W, uint32_t* dest_out, int32_t* lkey_out, int32_t* oor_count,
                               void* stream) {
  CTR_REQUIRE(B >= 0 && F >= 0 && W >= 1, "ctr_shard_route: bad size");
  const int64_t n = (int64_t)B * F;
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(idx && row_offset && local_offset && dest_out && lkey_out, "ctr_shard_route: null pointer");
  hipLaunchKernelGGL(shard_route_kernel, dim3(grid_of(n)), dim3(256), 0, (hipStream_t)stream, idx, row_offset,
                     local_offset, n, F, W, dest_out, lkey_out, oor_count);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_shard_bucket(const uint32_t* dest_sorted, const uint32_t* perm, const int32_t* lkey, int64_t n, int W,
                                int64_t cap, int32_t* slot_key_out, uint32_t* slot_perm_out, int32_t* overflow_count,
                                float* z_zero, int64_t z_ld, int F, int E, void* stream) {
  CTR_REQUIRE(n >= 0 && W >= 1 && W <= 64 && cap >= 1, "ctr_shard_bucket: bad size (1 <= W <= 64, cap >= 1)");
  CTR_REQUIRE(!z_zero || (F >= 1 && E >= 1 && z_ld >= (int64_t)F * E), "ctr_shard_bucket: bad F / E / z_ld");
  CTR_REQUIRE(slot_key_out && slot_perm_out && overflow_count, "ctr_shard_bucket: null output");
  hipStream_t st = (hipStream_t)stream;
  CTR_HIP(hipMemsetAsync(slot_key_out, 0xFF, (size_t)W * cap * 4, st));
  CTR_HIP(hipMemsetAsync(slot_perm_out, 0xFF, (size_t)W * cap * 4, st));
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(dest_sorted && perm && lkey, "ctr_shard_bucket: null input");
  hipLaunchKernelGGL(shard_bucket_kernel, dim3(grid_of(n)), dim3(256), 0, st, dest_sorted, perm, lkey, n, W, cap,
                     slot_key_out, slot_perm_out, overflow_count, z_zero, z_ld, z_zero ? F : 1, E);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_rows_gather(const void* src, int64_t ld, const uint32_t* perm, int64_t n, int F, int E, void* out,
                               void* stream) {
  CTR_REQUIRE(n >= 0 && F >= 1 && E >= 1 && ld >= (int64_t)F * E, "ctr_rows_gather: bad size / ld");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(src && perm && out, "ctr_rows_gather: null pointer");
  hipLaunchKernelGGL(rows_gather_kernel, dim3(grid_of(n * E)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)src, ld, perm, n, F, E, (uint32_t*)out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_rows_scatter(const void* rows, const uint32_t* perm, int64_t n, int F, int E, void* dst, int64_t ld,
                                void* stream) {
  CTR_REQUIRE(n >= 0 && F >= 1 && E >= 1 && ld >= (int64_t)F * E, "ctr_rows_scatter: bad size / ld");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(rows && perm && dst, "ctr_rows_scatter: null pointer");
  hipLaunchKernelGGL(rows_scatter_kernel, dim3(grid_of(n * E)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)rows, perm, n, F, E, (uint32_t*)dst, ld);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
