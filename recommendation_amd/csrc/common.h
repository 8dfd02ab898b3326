// Shared helpers for libctrhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ctrhip.h"

namespace ctr {

void set_error(const char* fmt, ...);

#define CTR_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ctr::set_error(__VA_ARGS__);        \
      return CTR_EINVAL;                  \
    }                                     \
  } while (0)

#define CTR_HIP(call)                                                \
  do {                                                               \
    hipError_t e__ = (call);                                         \
    if (e__ != hipSuccess) {                                         \
      ctr::set_error("%s: %s", __func__, hipGetErrorString(e__));    \
      return (int)e__;                                               \
    }                                                                \
  } while (0)

#define CTR_LAUNCH_CHECK()                                           \
  do {                                                               \
    hipError_t e__ = hipGetLastError();                              \
    if (e__ != hipSuccess) {                                         \
      ctr::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return (int)e__;                                               \
    }                                                                \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// splitmix64-finalizer hash -> 24 random bits. Same function on host and device.
// 24-bit counter hash of (seed, stream, element index).  The (seed, stream) half is uniform over a
// launch and folds into scalar registers; per element it is three to four 32-bit multiplies and a
// few shifts (murmur3-style finaliser), which matters in the GEMM epilogues that drop every output.
__host__ __device__ static inline uint32_t hash24(uint64_t seed, uint64_t stream_id, uint64_t index) {
  uint64_t k = seed * 0xBF58476D1CE4E5B9ull + stream_id * 0x94D049BB133111EBull;
  k ^= k >> 29;
  const uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
  const uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
  uint32_t h = (lo ^ k0) * 0x9E3779B1u;
  h ^= h >> 15;
  h = (h ^ k1 ^ (hi * 0x85EBCA77u)) * 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >> 8;
}

struct Drop {
  const uint64_t* seed_add;   // optional device word added to the seed (the step count of a replayed graph)
  uint64_t seed, stream_id;
  uint32_t thr;
  float scale;
  int on;
};

static inline Drop make_drop(const ctr_dropout_t* d) {
  Drop r;
  r.seed_add = nullptr; r.seed = 0; r.stream_id = 0; r.thr = 1u << 24; r.scale = 1.f; r.on = 0;
  if (d && d->keep_threshold > 0 && d->keep_threshold < (1u << 24)) {
    r.seed = d->seed; r.stream_id = d->stream_id; r.thr = d->keep_threshold; r.scale = d->scale; r.on = 1;
    r.seed_add = d->seed_add;
  }
  return r;
}

// multiplier (0 or scale) of element `index` under dropout d
__device__ static inline float drop_mul(const Drop& d, uint64_t index) {
  if (!d.on) return 1.f;
  return hash24(d.seed + (d.seed_add ? *d.seed_add : 0ull), d.stream_id, index) < d.thr ? d.scale : 0.f;
}

// The descriptor with the device word already added to its seed.  A kernel that draws many masks calls this once on
// its by-value copy: inside drop_mul the word is a load the compiler must repeat after every store that might alias it.
__device__ static inline Drop resolve_drop(Drop d) {
  if (d.on && d.seed_add) d.seed += *d.seed_add;
  d.seed_add = nullptr;
  return d;
}

// x + (x of the lane CTRL names inside this lane's row of 16), on the VALU's data-parallel path (no LDS)
template <int CTRL>
__device__ static inline float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false);
  return x + __builtin_bit_cast(float, y);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR = 0x120;

// Two values per lane in, one out: lanes 0-31 get a summed over the lane pairs (l, l + 32), lanes 32-63 get b.
__device__ static inline float half_swap_sum(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
// Even rows of 16 lanes get a summed over the row pairs (0,1) and (2,3), odd rows get b.
__device__ static inline float row_swap_sum(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// Sum over the 64 lanes, the same bits in every lane: four in-row steps on the data-parallel path, then the
// rows of 16 through two lane-swap instructions (no LDS crossbar traffic).
__device__ static inline float wave_sum(float v) {
  v = dpp_add<DPP_XOR1>(v);
  v = dpp_add<DPP_XOR2>(v);
  v = dpp_add<DPP_ROR + 4>(v);
  v = dpp_add<DPP_ROR + 8>(v);
  v = row_swap_sum(v, v);
  return half_swap_sum(v, v);
}

// Buffer-descriptor addressing: the hardware checks every access against the buffer's size, so a lane that has
// nothing to load or store passes an offset beyond the buffer (OOB) and gets zeros back, or has its store dropped,
// where plain pointers need a branch around the access.  Buffers must stay below 2 GiB (32-bit byte offsets).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;            // an offset no buffer here reaches

__device__ static inline __amdgpu_buffer_rsrc_t buf_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ static inline float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ static inline void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0);
}
__device__ static inline void buf_st1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
__device__ static inline float buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ static inline int buf_ld1i(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
}
static inline bool fits_buffer(int64_t bytes) { return bytes >= 0 && bytes < ((int64_t)1 << 31); }

// A "largest magnitude" record is CTR_AMAX_FLOATS floats: the maximum is kept in AMAX_WORDS words that lie
// AMAX_STRIDE floats (128 bytes) apart, and its value is the largest of them.  Thousands of waves raise a record as
// they finish; atomics on one word are served one after the other (about 3.5 ns each: a 16 K-wave kernel would end
// with 50 us of queued atomics), spread over 32 lines they cost nothing measurable, and no wave has to read the
// record back first.
constexpr int AMAX_WORDS = 32, AMAX_STRIDE = 32;
static_assert(AMAX_WORDS * AMAX_STRIDE == CTR_AMAX_FLOATS, "record size");

// Raises the record to the largest of the wave's non-negative values: one integer atomic per wave.
__device__ static inline void raise_amax(float* rec, float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0 && v > 0.f) {
    const unsigned word = (blockIdx.x + 13u * blockIdx.y + (threadIdx.x >> 6)) & (AMAX_WORDS - 1);
    atomicMax(reinterpret_cast<unsigned int*>(rec) + word * AMAX_STRIDE, __float_as_uint(v));
  }
}

// The record's value, the same in every lane (a kernel reads records that earlier launches wrote).
__device__ static inline float read_amax(const float* rec) {
  const int lane = threadIdx.x & 63;
  float v = lane < AMAX_WORDS ? rec[lane * AMAX_STRIDE] : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// *out = max(*out, max |X|) over a [rows, cols] view (gemm.hip): for producers that do not track it themselves
void launch_absmax(const float* X, int64_t ld, int64_t rows, int cols, float* out, hipStream_t st);

}  // namespace ctr
