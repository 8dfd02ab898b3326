// Embedding + FM front end: gather, value-scaled continuous rows, first/second order,
// deep-input concat, and the backward with respect to Z.  One wave per sample; a row of
// E floats is covered by G = E/4 lanes (16 B per lane), so a wave moves R = 64/G rows
// per step and every row read/write is one contiguous 4*E-byte segment.
#include "common.h"

namespace ctr {

__device__ static inline float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Sum over the lanes that share lane % G inside a row of 16 lanes (G <= 16); every lane gets a total.
template <int G>
__device__ static inline float row_stride_sum(float x) {
  if (G <= 1) x = dpp_add<DPP_ROR + 1>(x);
  if (G <= 2) x = dpp_add<DPP_ROR + 2>(x);
  if (G <= 4) x = dpp_add<DPP_ROR + 4>(x);
  if (G <= 8) x = dpp_add<DPP_ROR + 8>(x);
  return x;
}

// x + (x of the lane that the mirror names inside this lane's row of 16): halves of 8 (0x141) or the whole row (0x140)
constexpr int DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

// Sum over the G consecutive lanes that hold one table row (G <= 16, aligned groups); every lane gets the total.
template <int G>
__device__ static inline float row_group_sum(float t) {
  if (G >= 2) t = dpp_add<DPP_XOR1>(t);
  if (G >= 4) t = dpp_add<DPP_XOR2>(t);
  if (G >= 8) t = dpp_add<DPP_HALF_MIRROR>(t);     // the two quads of a half hold equal values by now
  if (G >= 16) t = dpp_add<DPP_ROW_MIRROR>(t);
  return t;
}

// Front end for E <= 64.  A row of 16 lanes (one DPP row) owns one sample, so a wave carries four samples and a
// block sixteen; inside the row, G = E/4 lanes cover one table row (16 B per lane) and SLOTS = 16/G rows move per
// step.  CH steps are issued together: first the CH ids of every lane, then the CH row loads they address, so a
// lane keeps CH independent 16-byte loads in flight (a sample's rows do not wait for one another) and every
// reduction of the FM terms stays inside the 16-lane row (data-parallel-path adds, no LDS, no cross-row swaps).
// The field offsets sit in LDS.  DENSE: Z already sits in z_out (N = F rows per sample); only the reductions run.
template <int G, bool DENSE>
__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const float* __restrict__ tables, const int64_t* __restrict__ row_offset,
    const int32_t* __restrict__ idx, const float* __restrict__ x_cont,
    const float* __restrict__ cont_w, int B, int F, int C, int E,
    float* __restrict__ z_out, int64_t z_ld, int cont_tail,
    float* __restrict__ first_out, int64_t first_ld, float* __restrict__ s_out,
    float* __restrict__ second_out, int64_t second_ld, Drop d_first, Drop d_second,
    int32_t* __restrict__ oor_count, const float* __restrict__ blk, const int64_t* __restrict__ fbase,
    const int64_t* __restrict__ fstride) {
  d_first = resolve_drop(d_first); d_second = resolve_drop(d_second);
  constexpr int SLOTS = 16 / G;
  constexpr int CH = 8;
  extern __shared__ int64_t offs_lds[];                     // [F + 1] first row of every field (gather path only)
  const bool gathers = !DENSE && !blk && F > 0;
  if (gathers) {
    for (int f = threadIdx.x; f <= F; f += blockDim.x) offs_lds[f] = row_offset[f];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int l16 = lane & 15;
  const int g = l16 % G;
  const int slot = l16 / G;
  const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = b < B;
  const int N = F + (cont_w ? C : 0);
  float4 s = f4_zero(), q = f4_zero();
  for (int n0 = 0; n0 < N; n0 += SLOTS * CH) {
    int x[CH];
    float4 z[CH];
    if (gathers) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int n = n0 + u * SLOTS + slot;
        x[u] = (live && n < F) ? idx[b * F + n] : -1;
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS + slot;
      z[u] = f4_zero();
      if (!live || n >= N) continue;
      if (DENSE) {
        z[u] = *reinterpret_cast<const float4*>(z_out + b * z_ld + (int64_t)n * E + g * 4);
      } else if (n < F && blk) {   // rows already fetched by their owners, one block of columns per owner
        z[u] = *reinterpret_cast<const float4*>(blk + fbase[n] + b * fstride[n] + g * 4);
      } else if (n < F) {
        const int64_t off = offs_lds[n];
        const int64_t V = offs_lds[n + 1] - off;
        if (x[u] >= 0 && x[u] < V) {
          z[u] = *reinterpret_cast<const float4*>(tables + (off + x[u]) * E + g * 4);
        } else if (oor_count && g == 0) {
          atomicAdd(oor_count, 1);
        }
      } else {
        const int c = n - F;
        const float v = x_cont[b * C + c];
        const float4 w = *reinterpret_cast<const float4*>(cont_w + (int64_t)c * E + g * 4);
        z[u] = make_float4(w.x * v, w.y * v, w.z * v, w.w * v);
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS + slot;
      const bool active = live && n < N;
      const float4 zz = z[u];
      if (!DENSE && z_out && active) *reinterpret_cast<float4*>(z_out + b * z_ld + (int64_t)n * E + g * 4) = zz;
      s.x += zz.x; s.y += zz.y; s.z += zz.z; s.w += zz.w;
      q.x += zz.x * zz.x; q.y += zz.y * zz.y; q.z += zz.z * zz.z; q.w += zz.w * zz.w;
      if (first_out) {
        const float t = row_group_sum<G>((zz.x + zz.y) + (zz.z + zz.w));
        if (active && g == 0) first_out[b * first_ld + n] = t * drop_mul(d_first, (uint64_t)b * N + n);
      }
    }
  }
  if (s_out || second_out) {
    // column sums over the row slots of the sample: lanes G apart inside the 16-lane row
    s.x = row_stride_sum<G>(s.x); s.y = row_stride_sum<G>(s.y); s.z = row_stride_sum<G>(s.z); s.w = row_stride_sum<G>(s.w);
    q.x = row_stride_sum<G>(q.x); q.y = row_stride_sum<G>(q.y); q.z = row_stride_sum<G>(q.z); q.w = row_stride_sum<G>(q.w);
    if (live && slot == 0) {
      if (s_out) *reinterpret_cast<float4*>(s_out + b * E + g * 4) = s;
      if (second_out) {
        const uint64_t base = (uint64_t)b * E + g * 4;
        float* o = second_out + b * second_ld + g * 4;
        o[0] = 0.5f * (s.x * s.x - q.x) * drop_mul(d_second, base + 0);
        o[1] = 0.5f * (s.y * s.y - q.y) * drop_mul(d_second, base + 1);
        o[2] = 0.5f * (s.z * s.z - q.z) * drop_mul(d_second, base + 2);
        o[3] = 0.5f * (s.w * s.w - q.w) * drop_mul(d_second, base + 3);
      }
    }
  }
  if (!DENSE && cont_tail && z_out && live) {
    for (int c = l16; c < C; c += 16) z_out[b * z_ld + (int64_t)N * E + c] = x_cont[b * C + c];
  }
}

// embed_fwd_kernel's gather path (tables in this process, no DENSE / assembled input) written so that the loop
// issues few instructions: the kernel above spends its time issuing them (about 2300 per wave for 39 fields of 16
// floats: a branch around every conditional access, 64-bit address arithmetic for every row of Z), not waiting for
// memory.  Here the ids, x_cont, Z and the first-order terms go through buffer descriptors (32-bit offsets, a lane
// with nothing to do passes an out-of-range offset, so no access sits behind a branch); a lane whose id is out of
// range reads the offsets array instead of a table row and the row is zeroed afterwards.  A lane keeps the rows
// n = slot, slot + SLOTS, ... of its sample, table rows first, then the continuous ones, in the same order as the
// kernel above, and adds them up in that order, so both kernels return the same bits.
template <int G>
__global__ __launch_bounds__(256) void embed_fwd_lean_kernel(
    const float* __restrict__ tables, const int64_t* __restrict__ row_offset,
    const int32_t* __restrict__ idx, const float* __restrict__ x_cont,
    const float* __restrict__ cont_w, int B, int F, int C,
    float* __restrict__ z_out, int64_t z_ld, int cont_tail,
    float* __restrict__ first_out, int64_t first_ld, float* __restrict__ s_out,
    float* __restrict__ second_out, int64_t second_ld, Drop d_first, Drop d_second,
    int32_t* __restrict__ oor_count, float* __restrict__ amax_out) {
  d_first = resolve_drop(d_first); d_second = resolve_drop(d_second);
  constexpr int SLOTS = 16 / G, CH = 4, E = 4 * G;
  float amx = 0.f;                                          // largest magnitude this lane writes to z_out
  extern __shared__ int64_t offs_lds[];                     // [F + 1] first row of every field
  for (int f = threadIdx.x; f <= F; f += 256) offs_lds[f] = row_offset[f];
  __syncthreads();
  const int l16 = threadIdx.x & 15, g = l16 % G, slot = l16 / G;
  const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = b < B;
  const int NC = cont_w ? C : 0, N = F + NC;
  const __amdgpu_buffer_rsrc_t ri = buf_of(idx, (unsigned)((int64_t)B * F * 4));
  const __amdgpu_buffer_rsrc_t rc = buf_of(x_cont, x_cont ? (unsigned)((int64_t)B * C * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rz = buf_of(z_out, z_out ? (unsigned)((int64_t)B * z_ld * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rf = buf_of(first_out, first_out ? (unsigned)((int64_t)B * first_ld * 4) : 0u);
  const unsigned ibase = live ? (unsigned)(b * F + slot) * 4 : OOB;
  const unsigned zbase = live ? (unsigned)(b * z_ld + slot * E + g * 4) * 4 : OOB;
  const unsigned fbase = live && g == 0 ? (unsigned)(b * first_ld + slot) * 4 : OOB;
  const float* const dummy = reinterpret_cast<const float*>(row_offset);      // 16 readable bytes (F >= 1)
  float4 s = f4_zero(), q = f4_zero();
  auto take = [&](float4 zz, int n, bool active) {          // one row of Z: store, sums, first-order term
    buf_st4(rz, active ? zbase + (unsigned)n * (E * 4) - (unsigned)slot * (E * 4) : OOB, zz);
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(zz.x), fabsf(zz.y))), fmaxf(fabsf(zz.z), fabsf(zz.w)));   // inactive rows are zero
    s.x += zz.x; s.y += zz.y; s.z += zz.z; s.w += zz.w;
    q.x += zz.x * zz.x; q.y += zz.y * zz.y; q.z += zz.z * zz.z; q.w += zz.w * zz.w;
    if (first_out) {
      const float t = row_group_sum<G>((zz.x + zz.y) + (zz.z + zz.w));
      buf_st1(rf, active ? fbase + (unsigned)(n - slot) * 4 : OOB, t * drop_mul(d_first, (uint64_t)b * N + n));
    }
  };
  for (int n0 = slot; n0 < F + slot; n0 += SLOTS * CH) {    // table rows n0 + u * SLOTS of this lane
    int x[CH];
    float4 z[CH];
    bool ok[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS;
      x[u] = buf_ld1i(ri, n < F ? ibase + (unsigned)(n - slot) * 4 : OOB);
    }
    bool bad = false;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS, nn = n < F ? n : F - 1;
      const int64_t off = offs_lds[nn], V = offs_lds[nn + 1] - off;
      ok[u] = live && n < F && x[u] >= 0 && x[u] < V;
      bad |= live && n < F && !ok[u];
      const float* src = ok[u] ? tables + (off + x[u]) * E + g * 4 : dummy;
      z[u] = *reinterpret_cast<const float4*>(src);
    }
    if (oor_count && __builtin_amdgcn_ballot_w64(bad)) {    // rare: count the ids outside their tables
#pragma unroll
      for (int u = 0; u < CH; ++u)
        if (live && n0 + u * SLOTS < F && !ok[u] && g == 0) atomicAdd(oor_count, 1);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      float4 zz = z[u];
      zz.x = ok[u] ? zz.x : 0.f; zz.y = ok[u] ? zz.y : 0.f; zz.z = ok[u] ? zz.z : 0.f; zz.w = ok[u] ? zz.w : 0.f;
      take(zz, n0 + u * SLOTS, live && n0 + u * SLOTS < F);
    }
  }
  // continuous rows: this lane's next row numbers n = slot (mod SLOTS) from F on
  const unsigned cbase = live ? (unsigned)(b * C) * 4 : OOB;
  for (int n0 = F + (slot - F % SLOTS + SLOTS) % SLOTS; n0 < N; n0 += SLOTS * CH) {
    float v[CH];
    float4 wv[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS, c = n < N ? n - F : 0;
      v[u] = buf_ld1(rc, n < N ? cbase + (unsigned)c * 4 : OOB);
      wv[u] = *reinterpret_cast<const float4*>(cont_w + (int64_t)c * E + g * 4);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int n = n0 + u * SLOTS;
      const float m = live && n < N ? v[u] : 0.f;
      take(make_float4(wv[u].x * m, wv[u].y * m, wv[u].z * m, wv[u].w * m), n, live && n < N);
    }
  }
  if (s_out || second_out) {
    // column sums over the row slots of the sample: lanes G apart inside the 16-lane row
    s.x = row_stride_sum<G>(s.x); s.y = row_stride_sum<G>(s.y); s.z = row_stride_sum<G>(s.z); s.w = row_stride_sum<G>(s.w);
    q.x = row_stride_sum<G>(q.x); q.y = row_stride_sum<G>(q.y); q.z = row_stride_sum<G>(q.z); q.w = row_stride_sum<G>(q.w);
    if (live && slot == 0) {
      if (s_out) *reinterpret_cast<float4*>(s_out + b * E + g * 4) = s;
      if (second_out) {
        const uint64_t base = (uint64_t)b * E + g * 4;
        float* o = second_out + b * second_ld + g * 4;
        o[0] = 0.5f * (s.x * s.x - q.x) * drop_mul(d_second, base + 0);
        o[1] = 0.5f * (s.y * s.y - q.y) * drop_mul(d_second, base + 1);
        o[2] = 0.5f * (s.z * s.z - q.z) * drop_mul(d_second, base + 2);
        o[3] = 0.5f * (s.w * s.w - q.w) * drop_mul(d_second, base + 3);
      }
    }
  }
  if (cont_tail && z_out && live) {
    for (int c = l16; c < C; c += 16) {
      const float v = x_cont[b * C + c];
      z_out[b * z_ld + (int64_t)N * E + c] = v;
      amx = fmaxf(amx, fabsf(v));
    }
  }
  if (amax_out) raise_amax(amax_out, amx);
}

// Same front end for E = 128 and 256 (G = 32, 64 lanes per row): one wave per sample.
template <int G, bool DENSE>
__global__ __launch_bounds__(256) void embed_fwd_wide_kernel(
    const float* __restrict__ tables, const int64_t* __restrict__ row_offset,
    const int32_t* __restrict__ idx, const float* __restrict__ x_cont,
    const float* __restrict__ cont_w, int B, int F, int C, int E,
    float* __restrict__ z_out, int64_t z_ld, int cont_tail,
    float* __restrict__ first_out, int64_t first_ld, float* __restrict__ s_out,
    float* __restrict__ second_out, int64_t second_ld, Drop d_first, Drop d_second,
    int32_t* __restrict__ oor_count, const float* __restrict__ blk, const int64_t* __restrict__ fbase,
    const int64_t* __restrict__ fstride) {
  d_first = resolve_drop(d_first); d_second = resolve_drop(d_second);
  constexpr int R = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int g = lane % G;
  const int r = lane / G;
  const int N = F + (cont_w ? C : 0);
  float4 s = f4_zero(), q = f4_zero();
  for (int n0 = 0; n0 < N; n0 += R) {
    const int n = n0 + r;
    const bool active = n < N;
    float4 z = f4_zero();
    if (active && DENSE) {
      z = *reinterpret_cast<const float4*>(z_out + b * z_ld + (int64_t)n * E + g * 4);
    } else if (active) {
      if (n < F && blk) {
        z = *reinterpret_cast<const float4*>(blk + fbase[n] + b * fstride[n] + g * 4);
      } else if (n < F) {
        const int x = idx[b * F + n];
        const int64_t off = row_offset[n];
        const int64_t V = row_offset[n + 1] - off;
        if (x >= 0 && x < V) {
          z = *reinterpret_cast<const float4*>(tables + (off + x) * E + g * 4);
        } else if (oor_count && g == 0) {
          atomicAdd(oor_count, 1);
        }
      } else {
        const int c = n - F;
        const float v = x_cont[b * C + c];
        const float4 w = *reinterpret_cast<const float4*>(cont_w + (int64_t)c * E + g * 4);
        z = make_float4(w.x * v, w.y * v, w.z * v, w.w * v);
      }
      if (!DENSE && z_out) *reinterpret_cast<float4*>(z_out + b * z_ld + (int64_t)n * E + g * 4) = z;
    }
    s.x += z.x; s.y += z.y; s.z += z.z; s.w += z.w;
    q.x += z.x * z.x; q.y += z.y * z.y; q.z += z.z * z.z; q.w += z.w * z.w;
    if (first_out) {
      float t = (z.x + z.y) + (z.z + z.w);
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (active && g == 0) first_out[b * first_ld + n] = t * drop_mul(d_first, (uint64_t)b * N + n);
    }
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    s.x += __shfl_xor(s.x, o); s.y += __shfl_xor(s.y, o); s.z += __shfl_xor(s.z, o); s.w += __shfl_xor(s.w, o);
    q.x += __shfl_xor(q.x, o); q.y += __shfl_xor(q.y, o); q.z += __shfl_xor(q.z, o); q.w += __shfl_xor(q.w, o);
  }
  if (r == 0) {
    if (s_out) *reinterpret_cast<float4*>(s_out + b * E + g * 4) = s;
    if (second_out) {
      const uint64_t base = (uint64_t)b * E + g * 4;
      float* o = second_out + b * second_ld + g * 4;
      o[0] = 0.5f * (s.x * s.x - q.x) * drop_mul(d_second, base + 0);
      o[1] = 0.5f * (s.y * s.y - q.y) * drop_mul(d_second, base + 1);
      o[2] = 0.5f * (s.z * s.z - q.z) * drop_mul(d_second, base + 2);
      o[3] = 0.5f * (s.w * s.w - q.w) * drop_mul(d_second, base + 3);
    }
  }
  if (!DENSE && cont_tail && z_out) {
    for (int c = lane; c < C; c += 64) z_out[b * z_ld + (int64_t)N * E + c] = x_cont[b * C + c];
  }
}

// Reference-exact one-hot block: every (row n of Z, 16-byte slot) is a pure function of
// (b, n): find the field that owns row n, compare with the sample's id, copy row 1 or zero.
__global__ __launch_bounds__(256) void embed_onehot_fwd_kernel(
    const float* __restrict__ tables, const int64_t* __restrict__ row_offset,
    const int32_t* __restrict__ idx, const float* __restrict__ x_cont, int B, int F, int C, int E,
    float* __restrict__ z_out, int64_t z_ld, int cont_tail, int32_t* __restrict__ pos_out) {
  extern __shared__ int64_t offs[];  // [F+1] offsets, then [F] ids as int64
  int64_t* ids = offs + (F + 1);
  const int64_t b = blockIdx.x;
  for (int f = threadIdx.x; f <= F; f += blockDim.x) offs[f] = row_offset[f];
  for (int f = threadIdx.x; f < F; f += blockDim.x) ids[f] = idx[b * F + f];
  __syncthreads();
  const int G = E / 4;
  const int64_t total = offs[F];
  for (int64_t p = threadIdx.x; p < total * G; p += blockDim.x) {
    const int64_t n = p / G;
    const int g = (int)(p % G);
    int lo = 0, hi = F;  // field f with offs[f] <= n < offs[f+1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offs[mid] <= n) lo = mid; else hi = mid;
    }
    float4 z = f4_zero();
    if (n - offs[lo] == ids[lo]) z = *reinterpret_cast<const float4*>(tables + (offs[lo] + 1) * E + g * 4);
    *reinterpret_cast<float4*>(z_out + b * z_ld + n * E + g * 4) = z;
  }
  if (cont_tail)
    for (int c = threadIdx.x; c < C; c += blockDim.x) z_out[b * z_ld + total * E + c] = x_cont[b * C + c];
  if (pos_out)
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
      const int64_t V = offs[f + 1] - offs[f];
      pos_out[b * F + f] = (ids[f] >= 0 && ids[f] < V) ? (int32_t)(offs[f] + ids[f]) : -1;
    }
}

__global__ __launch_bounds__(256) void embed_bwd_dz_kernel(
    const float* gz, int64_t gz_ld, const float* __restrict__ g_first, int64_t g_first_ld,
    const float* __restrict__ g_second, int64_t g_second_ld, const float* __restrict__ s,
    const float* __restrict__ z, int64_t z_ld, const float* __restrict__ g_sum, int B, int N, int E,
    Drop d_first, Drop d_second, float* dz, int64_t dz_ld, float* __restrict__ blk, const int64_t* __restrict__ fbase,
    const int64_t* __restrict__ fstride, int F) {
  d_first = resolve_drop(d_first); d_second = resolve_drop(d_second);
  const int G = E / 4;
  const int64_t total = (int64_t)B * N * G;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(p % G);
    const int64_t bn = p / G;
    const int n = (int)(bn % N);
    const int64_t b = bn / N;
    float4 a = f4_zero();
    if (gz) a = *reinterpret_cast<const float4*>(gz + b * gz_ld + (int64_t)n * E + g * 4);
    if (g_first) {
      const float t = g_first[b * g_first_ld + n] * drop_mul(d_first, (uint64_t)b * N + n);
      a.x += t; a.y += t; a.z += t; a.w += t;
    }
    if (g_second) {
      const float4 zz = *reinterpret_cast<const float4*>(z + b * z_ld + (int64_t)n * E + g * 4);
      const float4 ss = *reinterpret_cast<const float4*>(s + b * E + g * 4);
      const float* g2 = g_second + b * g_second_ld + g * 4;
      const uint64_t base = (uint64_t)b * E + g * 4;
      // the fused multiply-add is spelled out: the GEMM epilogue that folds this pass into the tower's last product
      // (ctr_fm_bwd_t) rounds the same way
      a.x = fmaf(g2[0] * drop_mul(d_second, base + 0), ss.x - zz.x, a.x);
      a.y = fmaf(g2[1] * drop_mul(d_second, base + 1), ss.y - zz.y, a.y);
      a.z = fmaf(g2[2] * drop_mul(d_second, base + 2), ss.z - zz.z, a.z);
      a.w = fmaf(g2[3] * drop_mul(d_second, base + 3), ss.w - zz.w, a.w);
    }
    if (g_sum) {
      const float4 gs = *reinterpret_cast<const float4*>(g_sum + b * E + g * 4);
      a.x += gs.x; a.y += gs.y; a.z += gs.z; a.w += gs.w;
    }
    // rows of sharded fields go straight into their owner's block of the exchange buffer
    float* dst = (blk && n < F) ? blk + fbase[n] + b * fstride[n] + g * 4 : dz + b * dz_ld + (int64_t)n * E + g * 4;
    *reinterpret_cast<float4*>(dst) = a;
  }
}

// Plain lookup, no reductions, for inputs with very few fields (DIN's item / category pair, a bias column): the
// sample-per-16-lanes layout above would leave most lanes without a row.  G lanes per (sample, field) row,
// 256 / G rows per block step, four steps in flight (ids first, then the rows they address).
template <int G>
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ tables,
                                                          const int64_t* __restrict__ row_offset,
                                                          const int32_t* __restrict__ idx, int B, int F, int E,
                                                          float* __restrict__ z_out, int64_t z_ld,
                                                          int32_t* __restrict__ oor_count) {
  constexpr int ROWS = 256 / G, UNR = 4;
  const int g = threadIdx.x % G;
  const uint32_t total = (uint32_t)B * (uint32_t)F;
  for (uint32_t base = blockIdx.x * (uint32_t)(ROWS * UNR); base < total; base += gridDim.x * (uint32_t)(ROWS * UNR)) {
    int x[UNR];
    uint32_t r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      r[u] = base + u * ROWS + threadIdx.x / G;
      x[u] = r[u] < total ? idx[r[u]] : -1;
    }
    float4 z[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      z[u] = f4_zero();
      if (r[u] >= total) continue;
      const uint32_t f = r[u] % (uint32_t)F;
      const int64_t off = row_offset[f];
      const int64_t V = row_offset[f + 1] - off;
      if (x[u] >= 0 && x[u] < V) z[u] = *reinterpret_cast<const float4*>(tables + (off + x[u]) * E + g * 4);
      else if (oor_count && g == 0) atomicAdd(oor_count, 1);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (r[u] >= total) continue;
      const uint32_t b = r[u] / (uint32_t)F, f = r[u] - b * (uint32_t)F;
      *reinterpret_cast<float4*>(z_out + (int64_t)b * z_ld + (int64_t)f * E + g * 4) = z[u];
    }
  }
}

// Multi-hot front end, first step: every bag (b, f) of a CSR list is summed into Z[b, f, :] in member order (a
// fixed order, so the result is bitwise reproducible); G = E/4 lanes cover a row, a bag's members go four at a
// time (ids first, then the four row loads they address).  Continuous rows and the x_cont tail are written as in
// embed_fwd_kernel, so the reductions can run over the finished Z.
template <int G>
__global__ __launch_bounds__(256) void embed_bag_pool_kernel(
    const float* __restrict__ tables, const int64_t* __restrict__ row_offset, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ bag_offsets, const float* __restrict__ x_cont, const float* __restrict__ cont_w,
    int B, int F, int C, int E, float* __restrict__ z_out, int64_t z_ld, int cont_tail,
    int32_t* __restrict__ oor_count) {
  constexpr int ROWS = 256 / G;                        // rows of Z a block writes per step
  const int g = threadIdx.x % G;
  const int N = F + (cont_w ? C : 0);
  const int64_t total = (int64_t)B * N;
  for (int64_t r = (int64_t)blockIdx.x * ROWS + threadIdx.x / G; r < total; r += (int64_t)gridDim.x * ROWS) {
    const int64_t b = r / N;
    const int n = (int)(r % N);
    float4 acc = f4_zero();
    if (n < F) {
      const int64_t off = row_offset[n];
      const int64_t V = row_offset[n + 1] - off;
      const int lo = bag_offsets[b * F + n], hi = bag_offsets[b * F + n + 1];
      for (int j = lo; j < hi; j += 4) {
        int x[4];
        float4 z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = j + u < hi ? idx[j + u] : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          z[u] = f4_zero();
          if (x[u] >= 0 && x[u] < V) z[u] = *reinterpret_cast<const float4*>(tables + (off + x[u]) * E + g * 4);
          else if (j + u < hi && oor_count && g == 0) atomicAdd(oor_count, 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += z[u].x; acc.y += z[u].y; acc.z += z[u].z; acc.w += z[u].w; }
      }
    } else {
      const float v = x_cont[b * C + (n - F)];
      const float4 w = *reinterpret_cast<const float4*>(cont_w + (int64_t)(n - F) * E + g * 4);
      acc = make_float4(w.x * v, w.y * v, w.z * v, w.w * v);
    }
    *reinterpret_cast<float4*>(z_out + b * z_ld + (int64_t)n * E + g * 4) = acc;
    if (cont_tail && n == 0)
      for (int c = g; c < C; c += G) z_out[b * z_ld + (int64_t)N * E + c] = x_cont[b * C + c];
  }
}

static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

}  // namespace ctr

using namespace ctr;

static int embed_fwd_impl(const float* tables, const int64_t* row_offset, const int32_t* idx,
                          const float* x_cont, const float* cont_w, int B, int F, int C, int E,
                          float* z_out, int64_t z_ld, int cont_tail, float* first_out,
                          int64_t first_ld, float* s_out, float* second_out, int64_t second_ld,
                          const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                          int32_t* oor_count, const float* blk, const int64_t* fbase, const int64_t* fstride,
                          void* stream, float* amax_out = nullptr) {
  CTR_REQUIRE(B >= 0 && F >= 0 && C >= 0, "ctr_embed_fwd: negative size");
  CTR_REQUIRE(!amax_out || z_out, "ctr_embed_fwd: amax_out is the largest magnitude written to z_out, which is null");
  CTR_REQUIRE(pow2(E) && E >= 4 && E <= 256, "ctr_embed_fwd: E=%d must be a power of two in [4,256]", E);
  CTR_REQUIRE(F == 0 || (tables && row_offset && idx) || (blk && fbase && fstride),
              "ctr_embed_fwd: null table/index pointer");
  CTR_REQUIRE(!z_out || (z_ld % 4 == 0 && ((uintptr_t)z_out % 16) == 0),
              "ctr_embed_fwd: z_out must be 16-byte aligned with z_ld %% 4 == 0");
  CTR_REQUIRE(!(cont_w || cont_tail) || x_cont || C == 0, "ctr_embed_fwd: x_cont missing");
  const int N = F + (cont_w ? C : 0);
  CTR_REQUIRE(!z_out || z_ld >= (int64_t)N * E + (cont_tail ? C : 0), "ctr_embed_fwd: z_ld too small");
  if (B == 0) return CTR_OK;
  CTR_REQUIRE(F <= 4096, "ctr_embed_fwd: at most 4096 fields");
  const Drop d1 = make_drop(drop_first), d2 = make_drop(drop_second);
  const dim3 grid((unsigned)cdiv(B, 16)), wgrid((unsigned)cdiv(B, 4)), block(256);
  const size_t lds = (size_t)(F + 1) * sizeof(int64_t);
  hipStream_t st = (hipStream_t)stream;
  if (F > 0 && F <= 4 && !blk && z_out && !first_out && !s_out && !second_out && N == F && !cont_tail &&
      (int64_t)B * F < 0xFFFF0000ll) {
    const int G = E / 4;
    const int64_t steps = cdiv((int64_t)B * F, (256 / G) * 4);
    const dim3 ggrid((unsigned)(steps < 16384 ? steps : 16384));
#define GATHER(GG) hipLaunchKernelGGL((embed_gather_kernel<GG>), ggrid, block, 0, st, tables, row_offset, idx, B, F, E, \
                                      z_out, z_ld, oor_count)
    switch (G) {
      case 1: GATHER(1); break;
      case 2: GATHER(2); break;
      case 4: GATHER(4); break;
      case 8: GATHER(8); break;
      case 16: GATHER(16); break;
      case 32: GATHER(32); break;
      case 64: GATHER(64); break;
    }
#undef GATHER
    if (amax_out) launch_absmax(z_out, z_ld, B, F * E, amax_out, st);
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  if (!blk && F > 0 && E <= 64 && fits_buffer((int64_t)B * F * 4) && fits_buffer((int64_t)B * C * 4) &&
      fits_buffer((int64_t)B * z_ld * 4) && fits_buffer((int64_t)B * first_ld * 4)) {
#define LEAN(GG)                                                                                           \
  hipLaunchKernelGGL((embed_fwd_lean_kernel<GG>), grid, block, lds, st, tables, row_offset, idx, x_cont, cont_w, B, F, C, \
                     z_out, z_ld, cont_tail, first_out, first_ld, s_out, second_out, second_ld, d1, d2, oor_count, \
                     z_out ? amax_out : nullptr)
    switch (E / 4) {
      case 1: LEAN(1); break;
      case 2: LEAN(2); break;
      case 4: LEAN(4); break;
      case 8: LEAN(8); break;
      case 16: LEAN(16); break;
    }
#undef LEAN
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
#define LAUNCH(GG)                                                                                  \
  hipLaunchKernelGGL((embed_fwd_kernel<GG, false>), grid, block, lds, st, tables, row_offset, idx, x_cont, \
                     cont_w, B, F, C, E, z_out, z_ld, cont_tail, first_out, first_ld, s_out,       \
                     second_out, second_ld, d1, d2, oor_count, blk, fbase, fstride)
#define LAUNCHW(GG)                                                                                 \
  hipLaunchKernelGGL((embed_fwd_wide_kernel<GG, false>), wgrid, block, 0, st, tables, row_offset, idx, x_cont, \
                     cont_w, B, F, C, E, z_out, z_ld, cont_tail, first_out, first_ld, s_out,       \
                     second_out, second_ld, d1, d2, oor_count, blk, fbase, fstride)
  switch (E / 4) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    case 16: LAUNCH(16); break;
    case 32: LAUNCHW(32); break;
    case 64: LAUNCHW(64); break;
  }
#undef LAUNCH
#undef LAUNCHW
  if (amax_out) launch_absmax(z_out, z_ld, B, N * E + (cont_tail ? C : 0), amax_out, st);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_embed_fwd_amax(const float* tables, const int64_t* row_offset, const int32_t* idx,
                                  const float* x_cont, const float* cont_w, int B, int F, int C, int E,
                                  float* z_out, int64_t z_ld, int cont_tail, float* first_out,
                                  int64_t first_ld, float* s_out, float* second_out, int64_t second_ld,
                                  const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                                  int32_t* oor_count, float* amax_out, void* stream) {
  return embed_fwd_impl(tables, row_offset, idx, x_cont, cont_w, B, F, C, E, z_out, z_ld, cont_tail, first_out,
                        first_ld, s_out, second_out, second_ld, drop_first, drop_second, oor_count, nullptr, nullptr,
                        nullptr, stream, amax_out);
}

extern "C" int ctr_embed_fwd(const float* tables, const int64_t* row_offset, const int32_t* idx,
                             const float* x_cont, const float* cont_w, int B, int F, int C, int E,
                             float* z_out, int64_t z_ld, int cont_tail, float* first_out,
                             int64_t first_ld, float* s_out, float* second_out, int64_t second_ld,
                             const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                             int32_t* oor_count, void* stream) {
  return embed_fwd_impl(tables, row_offset, idx, x_cont, cont_w, B, F, C, E, z_out, z_ld, cont_tail, first_out,
                        first_ld, s_out, second_out, second_ld, drop_first, drop_second, oor_count, nullptr, nullptr,
                        nullptr, stream);
}

extern "C" int ctr_embed_assemble_fwd(const float* blocks, const int64_t* fbase, const int64_t* fstride,
                                      const float* x_cont, const float* cont_w, int B, int F, int C, int E,
                                      float* z_out, int64_t z_ld, int cont_tail, float* first_out,
                                      int64_t first_ld, float* s_out, float* second_out, int64_t second_ld,
                                      const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second,
                                      void* stream) {
  CTR_REQUIRE(F == 0 || (blocks && fbase && fstride && ((uintptr_t)blocks % 16) == 0),
              "ctr_embed_assemble_fwd: null / misaligned block pointer");
  return embed_fwd_impl(nullptr, nullptr, nullptr, x_cont, cont_w, B, F, C, E, z_out, z_ld, cont_tail, first_out,
                        first_ld, s_out, second_out, second_ld, drop_first, drop_second, nullptr, blocks, fbase,
                        fstride, stream);
}

extern "C" int ctr_fm_fwd(const float* z, int64_t z_ld, int B, int N, int E, float* first_out,
                         int64_t first_ld, float* s_out, float* second_out, int64_t second_ld,
                         const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second, void* stream) {
  CTR_REQUIRE(B >= 0 && N >= 0, "ctr_fm_fwd: negative size");
  CTR_REQUIRE(pow2(E) && E >= 4 && E <= 256, "ctr_fm_fwd: E=%d must be a power of two in [4,256]", E);
  CTR_REQUIRE(z && z_ld % 4 == 0 && ((uintptr_t)z % 16) == 0 && z_ld >= (int64_t)N * E, "ctr_fm_fwd: z alignment/size");
  if (B == 0) return CTR_OK;
  const Drop d1 = make_drop(drop_first), d2 = make_drop(drop_second);
  const dim3 grid((unsigned)cdiv(B, 16)), wgrid((unsigned)cdiv(B, 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  float* zz = const_cast<float*>(z);  // the DENSE instantiation only reads it
#define LAUNCH(KERNEL, GRID, GG)                                                                    \
  hipLaunchKernelGGL((KERNEL<GG, true>), GRID, block, 0, st, (const float*)nullptr,                 \
                     (const int64_t*)nullptr, (const int32_t*)nullptr, (const float*)nullptr,      \
                     (const float*)nullptr, B, N, 0, E, zz, z_ld, 0, first_out, first_ld, s_out,   \
                     second_out, second_ld, d1, d2, (int32_t*)nullptr, (const float*)nullptr,      \
                     (const int64_t*)nullptr, (const int64_t*)nullptr)
  switch (E / 4) {
    case 1: LAUNCH(embed_fwd_kernel, grid, 1); break;
    case 2: LAUNCH(embed_fwd_kernel, grid, 2); break;
    case 4: LAUNCH(embed_fwd_kernel, grid, 4); break;
    case 8: LAUNCH(embed_fwd_kernel, grid, 8); break;
    case 16: LAUNCH(embed_fwd_kernel, grid, 16); break;
    case 32: LAUNCH(embed_fwd_wide_kernel, wgrid, 32); break;
    case 64: LAUNCH(embed_fwd_wide_kernel, wgrid, 64); break;
  }
#undef LAUNCH
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_embed_bag_fwd(const float* tables, const int64_t* row_offset, const int32_t* idx,
                                 const int32_t* bag_offsets, int64_t nnz, const float* x_cont, const float* cont_w,
                                 int B, int F, int C, int E, float* z_out, int64_t z_ld, int cont_tail,
                                 float* first_out, int64_t first_ld, float* s_out, float* second_out,
                                 int64_t second_ld, const ctr_dropout_t* drop_first,
                                 const ctr_dropout_t* drop_second, int32_t* oor_count, void* stream) {
  CTR_REQUIRE(B >= 0 && F >= 0 && C >= 0 && nnz >= 0 && nnz <= 0x7FFFFFFFll, "ctr_embed_bag_fwd: bad size");
  CTR_REQUIRE(pow2(E) && E >= 4 && E <= 256, "ctr_embed_bag_fwd: E=%d must be a power of two in [4,256]", E);
  CTR_REQUIRE(F == 0 || (tables && row_offset && bag_offsets && (idx || nnz == 0)), "ctr_embed_bag_fwd: null pointer");
  CTR_REQUIRE(z_out && z_ld % 4 == 0 && ((uintptr_t)z_out % 16) == 0,
              "ctr_embed_bag_fwd: z_out is required, 16-byte aligned with z_ld %% 4 == 0");
  CTR_REQUIRE(!(cont_w || cont_tail) || x_cont || C == 0, "ctr_embed_bag_fwd: x_cont missing");
  const int N = F + (cont_w ? C : 0);
  CTR_REQUIRE(z_ld >= (int64_t)N * E + (cont_tail ? C : 0), "ctr_embed_bag_fwd: z_ld too small");
  CTR_REQUIRE(!cont_tail || N > 0, "ctr_embed_bag_fwd: cont_tail needs at least one row of Z");
  if (B == 0 || N == 0) return CTR_OK;
  const int G = E / 4;
  const int64_t steps = cdiv((int64_t)B * N, 256 / G);
  const dim3 grid((unsigned)(steps < 8192 ? steps : 8192)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(GG)                                                                                          \
  hipLaunchKernelGGL((embed_bag_pool_kernel<GG>), grid, block, 0, st, tables, row_offset, idx, bag_offsets, \
                     x_cont, cont_w, B, F, C, E, z_out, z_ld, cont_tail, oor_count)
  switch (G) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    case 16: LAUNCH(16); break;
    case 32: LAUNCH(32); break;
    case 64: LAUNCH(64); break;
  }
#undef LAUNCH
  CTR_LAUNCH_CHECK();
  if (!first_out && !s_out && !second_out) return CTR_OK;
  return ctr_fm_fwd(z_out, z_ld, B, N, E, first_out, first_ld, s_out, second_out, second_ld, drop_first, drop_second,
                    stream);
}

extern "C" int ctr_embed_onehot_fwd(const float* tables, const int64_t* row_offset,
                                    const int32_t* idx, const float* x_cont, int B, int F, int C,
                                    int E, float* z_out, int64_t z_ld, int cont_tail,
                                    int32_t* pos_out, void* stream) {
  CTR_REQUIRE(B >= 0 && F > 0 && C >= 0, "ctr_embed_onehot_fwd: bad size");
  CTR_REQUIRE(E >= 4 && E % 4 == 0, "ctr_embed_onehot_fwd: E=%d must be a multiple of 4", E);
  CTR_REQUIRE(tables && row_offset && idx && z_out, "ctr_embed_onehot_fwd: null pointer");
  CTR_REQUIRE(z_ld % 4 == 0 && ((uintptr_t)z_out % 16) == 0, "ctr_embed_onehot_fwd: z_out alignment");
  CTR_REQUIRE(F <= 4096, "ctr_embed_onehot_fwd: F too large");
  if (B == 0) return CTR_OK;
  const size_t shmem = (size_t)(2 * F + 1) * sizeof(int64_t);
  hipLaunchKernelGGL(embed_onehot_fwd_kernel, dim3(B), dim3(256), shmem, (hipStream_t)stream, tables,
                     row_offset, idx, x_cont, B, F, C, E, z_out, z_ld, cont_tail, pos_out);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

static int embed_bwd_dz_impl(const float* gz, int64_t gz_ld, const float* g_first, int64_t g_first_ld,
                             const float* g_second, int64_t g_second_ld, const float* s, const float* z,
                             int64_t z_ld, const float* g_sum, int B, int N, int E,
                             const ctr_dropout_t* drop_first, const ctr_dropout_t* drop_second, float* dz,
                             int64_t dz_ld, float* blk, const int64_t* fbase, const int64_t* fstride, int F,
                             void* stream) {
  CTR_REQUIRE(B >= 0 && N >= 0 && E >= 4 && E % 4 == 0, "ctr_embed_bwd_dz: bad size");
  CTR_REQUIRE(dz && dz_ld % 4 == 0 && ((uintptr_t)dz % 16) == 0, "ctr_embed_bwd_dz: dz alignment");
  CTR_REQUIRE(!gz || (gz_ld % 4 == 0 && ((uintptr_t)gz % 16) == 0), "ctr_embed_bwd_dz: gz alignment");
  CTR_REQUIRE(!g_second || (s && z && z_ld % 4 == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)s % 16) == 0),
              "ctr_embed_bwd_dz: g_second needs s and z (16-byte aligned)");
  CTR_REQUIRE(!g_sum || ((uintptr_t)g_sum % 16) == 0, "ctr_embed_bwd_dz: g_sum alignment");
  const int64_t total = (int64_t)B * N * (E / 4);
  if (total == 0) return CTR_OK;
  const int64_t blocks = cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096;
  hipLaunchKernelGGL(embed_bwd_dz_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gz,
                     gz_ld, g_first, g_first_ld, g_second, g_second_ld, s, z, z_ld, g_sum, B, N, E,
                     make_drop(drop_first), make_drop(drop_second), dz, dz_ld, blk, fbase, fstride, F);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_embed_bwd_dz(const float* gz, int64_t gz_ld, const float* g_first,
                                int64_t g_first_ld, const float* g_second, int64_t g_second_ld,
                                const float* s, const float* z, int64_t z_ld, const float* g_sum, int B,
                                int N, int E, const ctr_dropout_t* drop_first,
                                const ctr_dropout_t* drop_second, float* dz, int64_t dz_ld, void* stream) {
  return embed_bwd_dz_impl(gz, gz_ld, g_first, g_first_ld, g_second, g_second_ld, s, z, z_ld, g_sum, B, N, E,
                           drop_first, drop_second, dz, dz_ld, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int ctr_embed_bwd_dz_blocks(const float* gz, int64_t gz_ld, const float* g_first,
                                       int64_t g_first_ld, const float* g_second, int64_t g_second_ld,
                                       const float* s, const float* z, int64_t z_ld, const float* g_sum, int B,
                                       int N, int E, const ctr_dropout_t* drop_first,
                                       const ctr_dropout_t* drop_second, float* dz, int64_t dz_ld, float* blocks_out,
                                       const int64_t* fbase, const int64_t* fstride, int F, void* stream) {
  CTR_REQUIRE(F >= 0 && F <= N && (F == 0 || (blocks_out && fbase && fstride && ((uintptr_t)blocks_out % 16) == 0)),
              "ctr_embed_bwd_dz_blocks: block layout arguments");
  return embed_bwd_dz_impl(gz, gz_ld, g_first, g_first_ld, g_second, g_second_ld, s, z, z_ld, g_sum, B, N, E,
                           drop_first, drop_second, dz, dz_ld, blocks_out, fbase, fstride, F, stream);
}
