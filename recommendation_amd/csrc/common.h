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
  uint64_t seed, stream_id;
  uint32_t thr;
  float scale;
  int on;
};

static inline Drop make_drop(const ctr_dropout_t* d) {
  Drop r;
  r.seed = 0; r.stream_id = 0; r.thr = 1u << 24; r.scale = 1.f; r.on = 0;
  if (d && d->keep_threshold > 0 && d->keep_threshold < (1u << 24)) {
    r.seed = d->seed; r.stream_id = d->stream_id; r.thr = d->keep_threshold; r.scale = d->scale; r.on = 1;
  }
  return r;
}

// multiplier (0 or scale) of element `index` under dropout d
__device__ static inline float drop_mul(const Drop& d, uint64_t index) {
  if (!d.on) return 1.f;
  return hash24(d.seed, d.stream_id, index) < d.thr ? d.scale : 0.f;
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace ctr
