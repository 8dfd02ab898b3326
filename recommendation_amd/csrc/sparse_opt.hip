// Segment-reduce over sorted (key, src) pairs + fused row-wise optimizer, and the dense
// (flat-buffer) optimizers.  A row of E floats is owned by G = E/4 lanes; runs of equal keys are
// added with a segmented scan in a fixed order (bitwise reproducible) and each touched table
// row is updated exactly once.
#include "common.h"

namespace ctr {

struct Opt {
  int kind;
  float lr, b1, b2, eps, l1, l2;
  const float* lr_dev;   // when set, the step size is read from device memory (a captured graph replays with new values)
  __device__ float LR() const { return lr_dev ? *lr_dev : lr; }
};

// The descriptor with the device step size already read.  A kernel calls this once on its by-value copy: inside the
// update formulas the word is a load the compiler must repeat after every store that might alias it.
__device__ static inline Opt resolve_opt(Opt o) {
  o.lr = o.LR();
  o.lr_dev = nullptr;
  return o;
}

__device__ static inline void ftrl1(const Opt& o, float g, float& w, float& n, float& z) {
  const float nn = n + g * g;
  z += g - (sqrtf(nn) - sqrtf(n)) / o.LR() * w;
  const float sgn = z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f);
  w = fabsf(z) <= o.l1 ? 0.f : -(z - sgn * o.l1) / (sqrtf(nn) / o.LR() + 2.f * o.l2);
  n = nn;
}

__device__ static inline void opt_apply4(const Opt& o, float4 g, float4* w, float4* s0, float4* s1) {
  if (o.kind == CTR_OPT_ADAM) {
    float4 m = *s0, v = *s1, p = *w;
    m.x = o.b1 * m.x + (1.f - o.b1) * g.x; m.y = o.b1 * m.y + (1.f - o.b1) * g.y;
    m.z = o.b1 * m.z + (1.f - o.b1) * g.z; m.w = o.b1 * m.w + (1.f - o.b1) * g.w;
    v.x = o.b2 * v.x + (1.f - o.b2) * g.x * g.x; v.y = o.b2 * v.y + (1.f - o.b2) * g.y * g.y;
    v.z = o.b2 * v.z + (1.f - o.b2) * g.z * g.z; v.w = o.b2 * v.w + (1.f - o.b2) * g.w * g.w;
    p.x -= o.LR() * m.x / (sqrtf(v.x) + o.eps); p.y -= o.LR() * m.y / (sqrtf(v.y) + o.eps);
    p.z -= o.LR() * m.z / (sqrtf(v.z) + o.eps); p.w -= o.LR() * m.w / (sqrtf(v.w) + o.eps);
    *s0 = m; *s1 = v; *w = p;
  } else if (o.kind == CTR_OPT_ADAGRAD) {
    float4 a = *s0, p = *w;
    a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
    p.x -= o.LR() * g.x / sqrtf(a.x); p.y -= o.LR() * g.y / sqrtf(a.y);
    p.z -= o.LR() * g.z / sqrtf(a.z); p.w -= o.LR() * g.w / sqrtf(a.w);
    *s0 = a; *w = p;
  } else if (o.kind == CTR_OPT_SGD) {
    float4 p = *w;
    p.x -= o.LR() * g.x; p.y -= o.LR() * g.y; p.z -= o.LR() * g.z; p.w -= o.LR() * g.w;
    *w = p;
  } else if (o.kind == CTR_OPT_FTRL) {
    float4 p = *w, n = *s0, z = *s1;
    ftrl1(o, g.x, p.x, n.x, z.x); ftrl1(o, g.y, p.y, n.y, z.y);
    ftrl1(o, g.z, p.z, n.z, z.z); ftrl1(o, g.w, p.w, n.w, z.w);
    *w = p; *s0 = n; *s1 = z;
  } else {
    *s0 = g;
  }
}

// The sorted array is cut into blocks of SEG pairs, one wave per block.
//   pass 1: every lane group fetches the gradient row of its OWN pair (R = 64/G pairs per round, all
//           rounds' loads in flight together), then a segmented inclusive scan over the groups adds
//           rows of equal key (keys are sorted, so "same key d groups back" means one run); a carry
//           links the rounds.  The last pair of each run holds the run's sum inside the block.
//           A run that lies inside the block is finished here (optimizer applied once).  A run that
//           starts here and continues goes to part_head[block] (+ head_key[block] = key); the part of a
//           run that started in an earlier block goes to part[block].
//   pass 2: one wave per block with a recorded head adds part_head + the continuation pieces in
//           block order (R pieces per step in fixed lane slots, then a fixed-order shuffle tree)
//           and applies the optimizer.
// Fixed summation order and no atomics: bitwise reproducible.
constexpr int SEG = 64;
constexpr uint32_t NO_KEY = 0xFFFFFFFFu;

__device__ static inline float4 shfl_up4(float4 v, int d) {
  return make_float4(__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.w, d));
}
__device__ static inline float4 shfl4(float4 v, int l) {
  return make_float4(__shfl(v.x, l), __shfl(v.y, l), __shfl(v.z, l), __shfl(v.w, l));
}

template <int G>
__global__ __launch_bounds__(256) void sparse_apply_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ src, int64_t n_items, int N, int E,
    int64_t total_rows, const float* __restrict__ dz, int64_t dz_ld, const float* __restrict__ src_scale,
    float* __restrict__ tables, float* __restrict__ slot0, float* __restrict__ slot1, int64_t slot_ld, Opt opt,
    float* __restrict__ part, float* __restrict__ part_head, uint32_t* __restrict__ head_key) {
  opt = resolve_opt(opt);
  constexpr int R = 64 / G;        // pairs per round
  constexpr int ROUNDS = SEG / R;  // = G
  constexpr int BATCH = ROUNDS < 4 ? ROUNDS : 4;
  const int lane = threadIdx.x & 63;
  const int g = lane % G, q = lane / G;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t start = blk * SEG;
  if (start >= n_items) return;
  const int64_t end = start + SEG < n_items ? start + SEG : n_items;
  const uint32_t first_key = keys[start];
  const bool first_is_head = start == 0 || keys[start - 1] != first_key;
  float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t carry_key = NO_KEY;
  uint32_t recorded = NO_KEY;      // key of the run that starts in this block and leaves it
  for (int r0 = 0; r0 < ROUNDS; r0 += BATCH) {
    uint32_t key[BATCH], nkey[BATCH];
    float4 val[BATCH];
    float sc[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int64_t item = start + (int64_t)(r0 + u) * R + q;
      const bool ok = item < end;
      key[u] = ok ? keys[item] : NO_KEY;
      nkey[u] = ok && item + 1 < n_items ? keys[item + 1] : NO_KEY;
      const uint32_t s = ok ? src[item] : 0u;
      val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      sc[u] = 1.f;
      if (ok && (int64_t)key[u] < total_rows) {
        const int64_t b = s / (uint32_t)N;
        const int64_t n = s % (uint32_t)N;
        val[u] = *reinterpret_cast<const float4*>(dz + b * dz_ld + n * E + g * 4);
        if (src_scale) sc[u] = src_scale[s];
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int64_t item = start + (int64_t)(r0 + u) * R + q;
      float4 v = make_float4(val[u].x * sc[u], val[u].y * sc[u], val[u].z * sc[u], val[u].w * sc[u]);
      const uint32_t k = key[u];
      if (q == 0 && k == carry_key) { v.x += carry.x; v.y += carry.y; v.z += carry.z; v.w += carry.w; }
#pragma unroll
      for (int d = 1; d < R; d <<= 1) {
        const float4 pv = shfl_up4(v, d * G);
        const uint32_t pk = __shfl_up(k, d * G);
        if (q >= d && pk == k) { v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w; }
      }
      carry = shfl4(v, (R - 1) * G + g);
      carry_key = __shfl(k, (R - 1) * G);
      const bool last = item == end - 1;
      if (item < end && (int64_t)k < total_rows && (nkey[u] != k || last)) {
        const bool leaves = last && nkey[u] == k;
        const bool open = k == first_key && !first_is_head;
        if (open) {
          *reinterpret_cast<float4*>(part + blk * E + g * 4) = v;
        } else if (leaves) {
          *reinterpret_cast<float4*>(part_head + blk * E + g * 4) = v;
          recorded = k;
        } else {
          const int64_t off = (int64_t)k * E + g * 4, soff = (int64_t)k * slot_ld + g * 4;
          opt_apply4(opt, v, reinterpret_cast<float4*>(tables + off),
                     slot0 ? reinterpret_cast<float4*>(slot0 + soff) : nullptr,
                     slot1 ? reinterpret_cast<float4*>(slot1 + soff) : nullptr);
        }
      }
      if (last && g == 0) head_key[blk] = recorded;
    }
  }
}

// opt_apply4 on registers (same expressions, so the same bits): the caller loads and stores the rows.
__device__ static inline void opt_math4(const Opt& o, float4 g, float4& p, float4& a, float4& c) {
  if (o.kind == CTR_OPT_ADAM) {
    float4& m = a; float4& v = c;
    m.x = o.b1 * m.x + (1.f - o.b1) * g.x; m.y = o.b1 * m.y + (1.f - o.b1) * g.y;
    m.z = o.b1 * m.z + (1.f - o.b1) * g.z; m.w = o.b1 * m.w + (1.f - o.b1) * g.w;
    v.x = o.b2 * v.x + (1.f - o.b2) * g.x * g.x; v.y = o.b2 * v.y + (1.f - o.b2) * g.y * g.y;
    v.z = o.b2 * v.z + (1.f - o.b2) * g.z * g.z; v.w = o.b2 * v.w + (1.f - o.b2) * g.w * g.w;
    p.x -= o.LR() * m.x / (sqrtf(v.x) + o.eps); p.y -= o.LR() * m.y / (sqrtf(v.y) + o.eps);
    p.z -= o.LR() * m.z / (sqrtf(v.z) + o.eps); p.w -= o.LR() * m.w / (sqrtf(v.w) + o.eps);
  } else if (o.kind == CTR_OPT_ADAGRAD) {
    a.x += g.x * g.x; a.y += g.y * g.y; a.z += g.z * g.z; a.w += g.w * g.w;
    p.x -= o.LR() * g.x / sqrtf(a.x); p.y -= o.LR() * g.y / sqrtf(a.y);
    p.z -= o.LR() * g.z / sqrtf(a.z); p.w -= o.LR() * g.w / sqrtf(a.w);
  } else if (o.kind == CTR_OPT_SGD) {
    p.x -= o.LR() * g.x; p.y -= o.LR() * g.y; p.z -= o.LR() * g.z; p.w -= o.LR() * g.w;
  } else if (o.kind == CTR_OPT_FTRL) {
    ftrl1(o, g.x, p.x, a.x, c.x); ftrl1(o, g.y, p.y, a.y, c.y);
    ftrl1(o, g.z, p.z, a.z, c.z); ftrl1(o, g.w, p.w, a.w, c.w);
  } else {
    a = g;
  }
}

// Exact n / d for 32-bit n and a divisor fixed at launch (multiply-high, two shifts), in place of the hardware-less
// integer division the compiler expands to some twenty-five instructions.
struct DivU32 {
  uint32_t m;
  int sh1, sh2;
  __device__ uint32_t quot(uint32_t n) const {
    const uint32_t t = __umulhi(m, n);
    return (t + ((n - t) >> sh1)) >> sh2;
  }
};
static inline DivU32 make_div(uint32_t d) {
  int l = 0;
  while ((1ull << l) < d) ++l;
  DivU32 r;
  r.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  r.sh1 = l < 1 ? l : 1;
  r.sh2 = l > 1 ? l - 1 : 0;
  return r;
}

// sparse_apply_kernel written for instruction count: the kernel above issues about 2000 instructions per 64 pairs (a segmented scan of four
// rounds through the LDS crossbar, a division per pair to find its gradient row, every access behind a branch) and
// is bound by that, not by memory.  Here a group of G lanes owns G consecutive pairs: their keys and sources come
// straight from memory (no lane exchange), the rows are added along the run inside the lane, and one segmented scan
// over the groups' last runs links the groups.  A pair past the end or with a key outside the tables reads gradient
// row 0 (its sum joins no run that is applied); table rows of the pairs that end no run are not touched (their
// lanes load row 0 and store nothing).  Same outputs as the kernel above (part, part_head, head_key
// for the tail pass); a run's rows are added in pair order, grouped as ((lane-serial) + groups scan).
// FM: the gradient rows lack the FM second-order term's product with the looked-up row; a second sum along every run,
// of g2[sample, :] (the gradient with respect to the second-order term), is carried through the same chain, scan and
// pieces (part2 / part_head2), and a run's gradient becomes sum(dz) - sum(g2) * W_old[key].
template <int G, bool FM = false>
__global__ __launch_bounds__(256) void sparse_apply_lean_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ src, int64_t n_items, int N, DivU32 divN,
    int64_t total_rows, const float* __restrict__ dz, int64_t dz_ld, const float* __restrict__ src_scale,
    float* __restrict__ tables, float* __restrict__ slot0, float* __restrict__ slot1, int64_t slot_ld, Opt opt,
    float* __restrict__ part, float* __restrict__ part_head, uint32_t* __restrict__ head_key,
    const float* __restrict__ g2, int64_t g2_ld, float* __restrict__ part2, float* __restrict__ part_head2) {
  opt = resolve_opt(opt);
  constexpr int R = 64 / G, IPG = G, E = 4 * G;
  const int lane = threadIdx.x & 63;
  const int g = lane % G, q = lane / G;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t start = blk * SEG;
  if (start >= n_items) return;
  const int cnt = (int)(n_items - start < SEG ? n_items - start : SEG);
  const __amdgpu_buffer_rsrc_t rk = buf_of(keys, (unsigned)(n_items * 4));
  const __amdgpu_buffer_rsrc_t rs = buf_of(src, (unsigned)(n_items * 4));
  const uint32_t first_key = keys[start];
  const bool first_is_head = start == 0 || keys[start - 1] != first_key;
  const unsigned kbase = (unsigned)(start + q * IPG) * 4;
  // keys of this group's pairs, of the pair before the group and of the pair after each one (0 past the array)
  uint32_t k[IPG], nk[IPG], sr[IPG];
  const uint32_t kprev = (uint32_t)buf_ld1i(rk, q > 0 ? kbase - 4 : OOB);
#pragma unroll
  for (int u = 0; u < IPG; ++u) {
    k[u] = (uint32_t)buf_ld1i(rk, kbase + 4 * u);
    sr[u] = (uint32_t)buf_ld1i(rs, kbase + 4 * u);
  }
  nk[IPG - 1] = (uint32_t)buf_ld1i(rk, kbase + 4 * IPG);
#pragma unroll
  for (int u = 0; u + 1 < IPG; ++u) nk[u] = k[u + 1];
  // Which pairs end a run, and how, follows from the keys alone: the table rows of those that get an optimizer step
  // are requested here, together with the gradient rows, so a wave waits for memory twice (keys, then rows), not
  // three times.  A pair that ends nothing loads row 0 and stores nothing.
  float4 val[IPG], p[IPG], a[IPG], c[IPG], gq[FM ? IPG : 1];
  bool in[IPG], live[IPG], more[IPG];                  // in the block; key inside the tables; next pair has the same key
  bool app[IPG], to_part[IPG], to_head[IPG];
  uint32_t recorded = NO_KEY;
#pragma unroll
  for (int u = 0; u < IPG; ++u) {
    const int i = q * IPG + u;
    in[u] = i < cnt;
    live[u] = in[u] && (int64_t)k[u] < total_rows;
    more[u] = start + i + 1 < n_items && nk[u] == k[u];
    const bool last = i == cnt - 1;
    const bool ends = live[u] && (!more[u] || last);
    const bool open = k[u] == first_key && !first_is_head;   // started in an earlier block: a piece for the tail pass
    const bool leaves = last && more[u];                      // goes on in the next block: the head piece
    app[u] = ends && !open && !leaves;
    to_part[u] = ends && open;
    to_head[u] = ends && !open && leaves;
    if (to_head[u]) recorded = k[u];
    if (in[u] && last && g == 0) head_key[blk] = recorded;
    const int64_t off = app[u] ? (int64_t)k[u] * E + g * 4 : 0;
    const int64_t soff = app[u] ? (int64_t)k[u] * slot_ld + g * 4 : 0;
    p[u] = a[u] = c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FM || opt.kind != CTR_OPT_GRAD_ONLY) p[u] = *reinterpret_cast<const float4*>(tables + off);
    if (opt.kind == CTR_OPT_ADAM || opt.kind == CTR_OPT_ADAGRAD || opt.kind == CTR_OPT_FTRL)
      a[u] = *reinterpret_cast<const float4*>(slot0 + soff);
    if (opt.kind == CTR_OPT_ADAM || opt.kind == CTR_OPT_FTRL) c[u] = *reinterpret_cast<const float4*>(slot1 + soff);
  }
#pragma unroll
  for (int u = 0; u < IPG; ++u) {
    const uint32_t sb = divN.quot(sr[u]), sn = sr[u] - sb * (uint32_t)N;      // source = sample * N + row of the sample
    // a pair outside the block or the tables reads row 0: its sum joins no run that is applied
    val[u] = *reinterpret_cast<const float4*>(dz + (live[u] ? (int64_t)sb * dz_ld + sn * E : 0) + g * 4);
    if (FM) {   // rows of g2 are 8-byte aligned (a column block of a wider buffer)
      const float* q2 = g2 + (live[u] ? (int64_t)sb * g2_ld : 0) + g * 4;
      const float2 lo = *reinterpret_cast<const float2*>(q2), hi = *reinterpret_cast<const float2*>(q2 + 2);
      gq[u] = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
  }
  if (src_scale) {
#pragma unroll
    for (int u = 0; u < IPG; ++u) {
      const float sc = src_scale[live[u] ? sr[u] : 0u];
      val[u].x *= sc; val[u].y *= sc; val[u].z *= sc; val[u].w *= sc;
    }
  }
  // along the run inside the lane; chain[u]: pairs 0..u of the group are one run
  bool chain[IPG];
  chain[0] = true;
#pragma unroll
  for (int u = 1; u < IPG; ++u) {
    const bool same = in[u] && k[u] == k[u - 1];
    chain[u] = chain[u - 1] && same;
    if (same) { val[u].x += val[u - 1].x; val[u].y += val[u - 1].y; val[u].z += val[u - 1].z; val[u].w += val[u - 1].w; }
    if (FM && same) { gq[u].x += gq[u - 1].x; gq[u].y += gq[u - 1].y; gq[u].z += gq[u - 1].z; gq[u].w += gq[u - 1].w; }
  }
  // the groups' last runs: T = sum of the group's last run, head = it does not continue the previous group's
  const bool joins = q > 0 && in[0] && k[0] == kprev;        // the group's first pair continues the previous group
  float4 T = val[IPG - 1], T2 = FM ? gq[IPG - 1] : make_float4(0.f, 0.f, 0.f, 0.f);
  bool head = !(chain[IPG - 1] && joins);
#pragma unroll
  for (int d = 1; d < R; d <<= 1) {
    const float4 pT = shfl_up4(T, d * G);
    float4 pT2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FM) pT2 = shfl_up4(T2, d * G);
    const int ph = __shfl_up((int)head, d * G);
    if (q >= d && !head) {
      T.x = pT.x + T.x; T.y = pT.y + T.y; T.z = pT.z + T.z; T.w = pT.w + T.w;
      if (FM) { T2.x = pT2.x + T2.x; T2.y = pT2.y + T2.y; T2.z = pT2.z + T2.z; T2.w = pT2.w + T2.w; }
      head = ph != 0;
    }
  }
  float4 carry = shfl_up4(T, G);                             // the run that ends the previous group, whole
  float4 carry2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FM) carry2 = shfl_up4(T2, G);
  if (!joins) carry = carry2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < IPG; ++u) {
    if (chain[u]) { val[u].x = carry.x + val[u].x; val[u].y = carry.y + val[u].y; val[u].z = carry.z + val[u].z; val[u].w = carry.w + val[u].w; }
    if (to_part[u]) *reinterpret_cast<float4*>(part + blk * E + g * 4) = val[u];
    if (to_head[u]) *reinterpret_cast<float4*>(part_head + blk * E + g * 4) = val[u];
    if (FM) {
      if (chain[u]) { gq[u].x = carry2.x + gq[u].x; gq[u].y = carry2.y + gq[u].y; gq[u].z = carry2.z + gq[u].z; gq[u].w = carry2.w + gq[u].w; }
      if (to_part[u]) *reinterpret_cast<float4*>(part2 + blk * E + g * 4) = gq[u];
      if (to_head[u]) *reinterpret_cast<float4*>(part_head2 + blk * E + g * 4) = gq[u];
      // the run's gradient: minus (sum of g2) times the row as it stood before this update
      val[u].x = fmaf(-gq[u].x, p[u].x, val[u].x); val[u].y = fmaf(-gq[u].y, p[u].y, val[u].y);
      val[u].z = fmaf(-gq[u].z, p[u].z, val[u].z); val[u].w = fmaf(-gq[u].w, p[u].w, val[u].w);
    }
  }
#pragma unroll
  for (int u = 0; u < IPG; ++u) opt_math4(opt, val[u], p[u], a[u], c[u]);
#pragma unroll
  for (int u = 0; u < IPG; ++u) {
    if (!app[u]) continue;
    const int64_t off = (int64_t)k[u] * E + g * 4, soff = (int64_t)k[u] * slot_ld + g * 4;
    if (opt.kind != CTR_OPT_GRAD_ONLY) *reinterpret_cast<float4*>(tables + off) = p[u];
    if (opt.kind != CTR_OPT_SGD) *reinterpret_cast<float4*>(slot0 + soff) = a[u];
    if (opt.kind == CTR_OPT_ADAM || opt.kind == CTR_OPT_FTRL) *reinterpret_cast<float4*>(slot1 + soff) = c[u];
  }
}

// Pass 2: one wave per block with a recorded head adds part_head + the continuation pieces in block order and applies
// the optimizer.  Most heads belong to short runs that end in the next block or two, so the first trip looks at R
// blocks only; a hot key's run spans hundreds of blocks, and from the second trip on the wave asks for the first keys
// and the pieces of 8 * R blocks together (no load waits for another: a block past the arrays reads the last one's).
// Every lane adds its pieces in ascending order, then the lanes are combined by a fixed shuffle tree.
template <int G, bool FM = false>
__global__ __launch_bounds__(256) void sparse_apply_tail_kernel(
    const uint32_t* __restrict__ keys, int64_t n_items, int64_t n_blocks, int E,
    float* __restrict__ tables, float* __restrict__ slot0, float* __restrict__ slot1, int64_t slot_ld, Opt opt,
    const float* __restrict__ part, const float* __restrict__ part_head,
    const uint32_t* __restrict__ head_key, const float* __restrict__ part2, const float* __restrict__ part_head2) {
  opt = resolve_opt(opt);
  constexpr int R = 64 / G;
  constexpr int TU = 8;
  const int lane = threadIdx.x & 63;
  const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= n_blocks) return;
  const uint32_t key = head_key[blk];
  if (key == NO_KEY) return;
  const int g = lane % G, r = lane / G;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r == 0) acc = *reinterpret_cast<const float4*>(part_head + blk * E + g * 4);
  if (FM && r == 0) acc2 = *reinterpret_cast<const float4*>(part_head2 + blk * E + g * 4);
  // continuation pieces sit in blocks blk+1, blk+2, ... while the block's first key is `key`
  bool more;
  {
    const int64_t k = blk + 1 + r, kc = k < n_blocks ? k : n_blocks - 1;
    const uint32_t fk = keys[kc * SEG];
    const float4 v = *reinterpret_cast<const float4*>(part + kc * E + g * 4);
    const bool ok = k < n_blocks && fk == key;
    if (ok) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (FM) {
      const float4 v2 = *reinterpret_cast<const float4*>(part2 + kc * E + g * 4);
      if (ok) { acc2.x += v2.x; acc2.y += v2.y; acc2.z += v2.z; acc2.w += v2.w; }
    }
    more = __ballot(ok) == __ballot(1);            // the run ends inside a trip as soon as one slot misses
  }
  for (int64_t k0 = blk + 1 + R; more && k0 < n_blocks; k0 += TU * R) {
    uint32_t fk[TU];
    float4 v[TU], v2[FM ? TU : 1];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int64_t k = k0 + u * R + r, kc = k < n_blocks ? k : n_blocks - 1;
      fk[u] = keys[kc * SEG];
      v[u] = *reinterpret_cast<const float4*>(part + kc * E + g * 4);
      if (FM) v2[u] = *reinterpret_cast<const float4*>(part2 + kc * E + g * 4);
    }
    bool all = true;
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      // a block past the run's first miss can carry the key again only if it is the same run: keys are sorted
      const bool ok = k0 + u * R + r < n_blocks && fk[u] == key;
      if (ok) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      if (FM && ok) { acc2.x += v2[u].x; acc2.y += v2[u].y; acc2.z += v2[u].z; acc2.w += v2[u].w; }
      all = all && ok;
    }
    more = __ballot(all) == __ballot(1);
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
    acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    if (FM) {
      acc2.x += __shfl_xor(acc2.x, o); acc2.y += __shfl_xor(acc2.y, o);
      acc2.z += __shfl_xor(acc2.z, o); acc2.w += __shfl_xor(acc2.w, o);
    }
  }
  if (r == 0) {
    const int64_t off = (int64_t)key * E + g * 4, soff = (int64_t)key * slot_ld + g * 4;
    if (FM) {
      const float4 w0 = *reinterpret_cast<const float4*>(tables + off);
      acc.x = fmaf(-acc2.x, w0.x, acc.x); acc.y = fmaf(-acc2.y, w0.y, acc.y);
      acc.z = fmaf(-acc2.z, w0.z, acc.z); acc.w = fmaf(-acc2.w, w0.w, acc.w);
    }
    opt_apply4(opt, acc, reinterpret_cast<float4*>(tables + off),
               slot0 ? reinterpret_cast<float4*>(slot0 + soff) : nullptr,
               slot1 ? reinterpret_cast<float4*>(slot1 + soff) : nullptr);
  }
}

__global__ __launch_bounds__(256) void dense_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ s0, float* __restrict__ s1,
                                                          int64_t n, Opt o) {
  o = resolve_opt(o);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    if (o.kind == CTR_OPT_ADAM) {
      const float m = o.b1 * s0[i] + (1.f - o.b1) * gi;
      const float v = o.b2 * s1[i] + (1.f - o.b2) * gi * gi;
      s0[i] = m; s1[i] = v;
      p[i] -= o.LR() * m / (sqrtf(v) + o.eps);
    } else if (o.kind == CTR_OPT_ADAGRAD) {
      const float a = s0[i] + gi * gi;
      s0[i] = a;
      p[i] -= o.LR() * gi / sqrtf(a);
    } else if (o.kind == CTR_OPT_FTRL) {
      float w = p[i], n = s0[i], z = s1[i];
      ftrl1(o, gi, w, n, z);
      p[i] = w; s0[i] = n; s1[i] = z;
    } else {
      p[i] -= o.LR() * gi;
    }
  }
}

// A row's byte: bit 0 = the row heads a run of this step's sorted pairs (the row-wise pass updates it), bit 1 = the
// row has received a gradient since the map was cleared, i.e. its m and v may be nonzero.
// touched[key] = 3 for every table row that heads a run of the sorted pairs (keys outside the tables mark nothing).
__global__ __launch_bounds__(256) void mark_rows_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t total_rows,
                                                       uint8_t* __restrict__ touched) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t k = keys[i];
    if ((int64_t)k < total_rows && (i == 0 || keys[i - 1] != k)) touched[k] = 3;
  }
}

// tf.train.AdamOptimizer's step on the rows that received no gradient: m = b1 m, v = b2 v, w -= lr m / (sqrt(v) + eps)
// (the expressions of dense_update_kernel with g = 0, so the same bits); rows marked in `touched` (bit 0) were
// updated by the row-wise pass and are skipped.  Four floats whose m and v are all +0 stay exactly as they are under
// these expressions (b1 * 0 = 0, b2 * 0 = 0, w - lr * 0 / (0 + eps) = w), so rows without optimizer state cost
// nothing but their byte (bit 0 of every byte is cleared on the way: the map is ready for the next step): on tables of tens of millions of rows most rows have never received a gradient.
// A wave walks chunks of 1024 rows: a lane reads the bytes of 16 rows at once, the rows to work on (bit 0 clear, bit 1
// set) are listed in the wave's piece of LDS, and the wave then takes the listed rows' quads 128 at a time: m, v and w
// requested together, written only where some value has state.
constexpr int AU_ROWS = 1024;        // rows per wave chunk (16 per lane)
__global__ __launch_bounds__(256) void adam_untouched_kernel(float* __restrict__ w, float* __restrict__ m,
                                                            float* __restrict__ v, int64_t rows, int q_per_row,
                                                            uint8_t* __restrict__ touched, Opt o) {
  __shared__ uint16_t lists[4][AU_ROWS];
  o = resolve_opt(o);
  // exact only while 0 / (0 + eps) is 0 and lr * 0 is 0
  const bool may_skip = o.eps > 0.f && o.LR() - o.LR() == 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  volatile uint16_t* list = lists[wave];
  const int64_t nchunks = (rows + AU_ROWS - 1) / AU_ROWS;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (int64_t)gridDim.x * 4) {
    const int64_t r0 = chunk * AU_ROWS + lane * 16;
    uint32_t by[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};     // past the end: "skip"
    if (r0 + 16 <= rows) {
      const uint4 q = *reinterpret_cast<const uint4*>(touched + r0);
      by[0] = q.x; by[1] = q.y; by[2] = q.z; by[3] = q.w;
    } else {
      for (int k = 0; k < 16; ++k)
        if (r0 + k < rows) by[k >> 2] = (by[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)touched[r0 + k] << (8 * (k & 3)));
    }
    // this lane is the only reader of its 16 bytes: it clears the step's marks (bit 0) on its way
    if (r0 + 16 <= rows) {
      if ((by[0] | by[1] | by[2] | by[3]) & 0x01010101u)
        *reinterpret_cast<uint4*>(touched + r0) = make_uint4(by[0] & ~0x01010101u, by[1] & ~0x01010101u,
                                                             by[2] & ~0x01010101u, by[3] & ~0x01010101u);
    } else {
      for (int k = 0; k < 16; ++k)
        if (r0 + k < rows && ((by[k >> 2] >> (8 * (k & 3))) & 1u)) touched[r0 + k] = (uint8_t)((by[k >> 2] >> (8 * (k & 3))) & 0xFEu);
    }
    uint32_t mask = 0u;                                  // bit k: row r0 + k is worked on
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t t = (by[k >> 2] >> (8 * (k & 3))) & 0xFFu;
      if (!(t & 1u) && ((t & 2u) || !may_skip)) mask |= 1u << k;
    }
    const uint32_t cnt = (uint32_t)__popc(mask);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (total == 0u) continue;                           // the whole chunk: rows that never received a gradient
    uint32_t pos = incl - cnt;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (mask & (1u << k)) list[pos++] = (uint16_t)(lane * 16 + k);
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = total * (uint32_t)q_per_row;      // quads of the listed rows
    const int64_t rbase = chunk * AU_ROWS;
    constexpr int U = 2;
    for (uint32_t base = 0; base < n; base += 64 * U) {
      float4 mm[U], vv[U], p[U];
      int64_t j[U];
      bool live[U];
      bool any_live = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t idx = base + u * 64 + lane;
        const bool ok = idx < n;
        const uint32_t li = ok ? idx / (uint32_t)q_per_row : 0u;
        const uint32_t qq = ok ? idx - li * (uint32_t)q_per_row : 0u;
        j[u] = (rbase + list[li]) * q_per_row + qq;
        mm[u] = reinterpret_cast<const float4*>(m)[j[u]];
        vv[u] = reinterpret_cast<const float4*>(v)[j[u]];
        p[u] = reinterpret_cast<const float4*>(w)[j[u]];   // listed rows have state as a rule: no second trip for w
        live[u] = ok;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t bits = __float_as_uint(mm[u].x) | __float_as_uint(mm[u].y) | __float_as_uint(mm[u].z) |
                              __float_as_uint(mm[u].w) | __float_as_uint(vv[u].x) | __float_as_uint(vv[u].y) |
                              __float_as_uint(vv[u].z) | __float_as_uint(vv[u].w);
        live[u] = live[u] && (bits != 0u || !may_skip);
        any_live = any_live || live[u];
      }
      if (!__any(any_live)) continue;                    // listed rows whose state is all zeros (marked, never updated)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        const float g = 0.f;
        float4 a = mm[u], c = vv[u], x = p[u];
        a.x = o.b1 * a.x + (1.f - o.b1) * g; a.y = o.b1 * a.y + (1.f - o.b1) * g;
        a.z = o.b1 * a.z + (1.f - o.b1) * g; a.w = o.b1 * a.w + (1.f - o.b1) * g;
        c.x = o.b2 * c.x + (1.f - o.b2) * g * g; c.y = o.b2 * c.y + (1.f - o.b2) * g * g;
        c.z = o.b2 * c.z + (1.f - o.b2) * g * g; c.w = o.b2 * c.w + (1.f - o.b2) * g * g;
        x.x -= o.LR() * a.x / (sqrtf(c.x) + o.eps); x.y -= o.LR() * a.y / (sqrtf(c.y) + o.eps);
        x.z -= o.LR() * a.z / (sqrtf(c.z) + o.eps); x.w -= o.LR() * a.w / (sqrtf(c.w) + o.eps);
        reinterpret_cast<float4*>(m)[j[u]] = a;
        reinterpret_cast<float4*>(v)[j[u]] = c;
        reinterpret_cast<float4*>(w)[j[u]] = x;
      }
    }
    __builtin_amdgcn_wave_barrier();                     // the list is rewritten by the next chunk
  }
}

// second launch: clears bit 0 of the marks (a row's threads of the pass above all read its mark, so none of them may
// clear it); bit 1 stays
__global__ __launch_bounds__(256) void unmark_rows_kernel(const uint32_t* __restrict__ keys, int64_t n, int64_t total_rows,
                                                         uint8_t* __restrict__ touched) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t k = keys[i];
    if ((int64_t)k < total_rows && (i == 0 || keys[i - 1] != k)) touched[k] = 2;
  }
}

static Opt make_opt(const ctr_opt_t* o) {
  Opt r;
  r.kind = o->kind; r.lr = o->lr; r.b1 = o->beta1; r.b2 = o->beta2; r.eps = o->eps; r.l1 = o->l1; r.l2 = o->l2;
  r.lr_dev = o->lr_dev;
  return r;
}

}  // namespace ctr

using namespace ctr;

static size_t apply_ws_bytes(int64_t n_items, int E, bool fm = false) {
  const int64_t nb = cdiv(n_items > 0 ? n_items : 1, SEG);
  return (fm ? 4 : 2) * align_up((size_t)nb * E * sizeof(float), 256) + align_up((size_t)nb * sizeof(uint32_t), 256);
}

extern "C" size_t ctr_emb_sparse_apply_fm_workspace_bytes(int64_t n_items, int E) {
  return apply_ws_bytes(n_items, E, true);
}

extern "C" size_t ctr_emb_sparse_apply_workspace_bytes(int64_t n_items, int E) {
  return apply_ws_bytes(n_items, E);
}

extern "C" int ctr_emb_sparse_apply(const uint32_t* keys_sorted, const uint32_t* src_sorted,
                                    int64_t n_items, int N, int E, int64_t total_rows, const float* dz,
                                    int64_t dz_ld, const float* src_scale, float* tables, float* slot0,
                                    float* slot1, const ctr_opt_t* opt, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return ctr_emb_sparse_apply_ld(keys_sorted, src_sorted, n_items, N, E, total_rows, dz, dz_ld, src_scale, tables,
                                 slot0, slot1, E, opt, workspace, workspace_bytes, stream);
}

static int apply_impl(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items, int N, int E,
                      int64_t total_rows, const float* dz, int64_t dz_ld, const float* src_scale, const float* g2,
                      int64_t g2_ld, float* tables, float* slot0, float* slot1, int64_t slot_ld, const ctr_opt_t* opt,
                      void* workspace, size_t workspace_bytes, void* stream);

extern "C" int ctr_emb_sparse_apply_ld(const uint32_t* keys_sorted, const uint32_t* src_sorted,
                                       int64_t n_items, int N, int E, int64_t total_rows, const float* dz,
                                       int64_t dz_ld, const float* src_scale, float* tables, float* slot0,
                                       float* slot1, int64_t slot_ld, const ctr_opt_t* opt, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return apply_impl(keys_sorted, src_sorted, n_items, N, E, total_rows, dz, dz_ld, src_scale, nullptr, 0, tables, slot0,
                    slot1, slot_ld, opt, workspace, workspace_bytes, stream);
}

extern "C" int ctr_emb_sparse_apply_fm(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items, int N,
                                       int E, int64_t total_rows, const float* dz, int64_t dz_ld, const float* g_second,
                                       int64_t g_second_ld, float* tables, float* slot0, float* slot1, int64_t slot_ld,
                                       const ctr_opt_t* opt, void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(g_second && tables, "ctr_emb_sparse_apply_fm: g_second and tables are required");
  CTR_REQUIRE(E <= 32 && g_second_ld >= E && g_second_ld % 2 == 0 && ((uintptr_t)g_second % 8) == 0,
              "ctr_emb_sparse_apply_fm: rows of at most 32 floats; g_second 8-byte aligned with an even row pitch >= E");
  return apply_impl(keys_sorted, src_sorted, n_items, N, E, total_rows, dz, dz_ld, nullptr, g_second, g_second_ld, tables,
                    slot0, slot1, slot_ld, opt, workspace, workspace_bytes, stream);
}

static int apply_impl(const uint32_t* keys_sorted, const uint32_t* src_sorted, int64_t n_items, int N, int E,
                      int64_t total_rows, const float* dz, int64_t dz_ld, const float* src_scale, const float* g2,
                      int64_t g2_ld, float* tables, float* slot0, float* slot1, int64_t slot_ld, const ctr_opt_t* opt,
                      void* workspace, size_t workspace_bytes, void* stream) {
  CTR_REQUIRE(opt, "ctr_emb_sparse_apply: opt is null");
  CTR_REQUIRE(slot_ld >= E && slot_ld % 4 == 0 && ((uintptr_t)slot0 % 16) == 0 && ((uintptr_t)slot1 % 16) == 0,
              "ctr_emb_sparse_apply: slot_ld must be a multiple of 4 floats, at least E, and the slots 16-byte aligned");
  CTR_REQUIRE(n_items >= 0 && N > 0, "ctr_emb_sparse_apply: bad size");
  CTR_REQUIRE(E >= 4 && E <= 256 && (E & (E - 1)) == 0, "ctr_emb_sparse_apply: E=%d must be a power of two in [4,256]", E);
  if (n_items == 0) return CTR_OK;
  CTR_REQUIRE(keys_sorted && src_sorted && dz, "ctr_emb_sparse_apply: null pointer");
  CTR_REQUIRE(dz_ld % 4 == 0 && ((uintptr_t)dz % 16) == 0, "ctr_emb_sparse_apply: dz alignment");
  switch (opt->kind) {
    case CTR_OPT_GRAD_ONLY: CTR_REQUIRE(slot0, "ctr_emb_sparse_apply: GRAD_ONLY needs slot0"); break;
    case CTR_OPT_ADAM: CTR_REQUIRE(tables && slot0 && slot1, "ctr_emb_sparse_apply: Adam needs tables, m, v"); break;
    case CTR_OPT_ADAGRAD: CTR_REQUIRE(tables && slot0, "ctr_emb_sparse_apply: Adagrad needs tables, acc"); break;
    case CTR_OPT_SGD: CTR_REQUIRE(tables, "ctr_emb_sparse_apply: SGD needs tables"); break;
    case CTR_OPT_FTRL: CTR_REQUIRE(tables && slot0 && slot1, "ctr_emb_sparse_apply: FTRL needs tables, n, z"); break;
    default: CTR_REQUIRE(false, "ctr_emb_sparse_apply: unknown optimizer kind %d", opt->kind);
  }
  const bool fm = g2 != nullptr;
  if (!workspace || workspace_bytes < apply_ws_bytes(n_items, E, fm)) {
    set_error("ctr_emb_sparse_apply: workspace %zu < %zu", workspace_bytes, apply_ws_bytes(n_items, E, fm));
    return CTR_EWORKSPACE;
  }
  const Opt o = make_opt(opt);
  hipStream_t st = (hipStream_t)stream;
  const int G = E / 4;
  const int64_t nb = cdiv(n_items, SEG);
  char* wsp = (char*)workspace;
  float* part = (float*)wsp; wsp += align_up((size_t)nb * E * sizeof(float), 256);
  float* part_head = (float*)wsp; wsp += align_up((size_t)nb * E * sizeof(float), 256);
  float *part2 = nullptr, *part_head2 = nullptr;
  if (fm) {
    part2 = (float*)wsp; wsp += align_up((size_t)nb * E * sizeof(float), 256);
    part_head2 = (float*)wsp; wsp += align_up((size_t)nb * E * sizeof(float), 256);
  }
  uint32_t* head_key = (uint32_t*)wsp;
  const dim3 grid((unsigned)cdiv(nb, 4)), block(256);  // one wave per block of SEG pairs; it also sets head_key
  const dim3 grid2((unsigned)cdiv(nb, 4));
  // GRAD_ONLY writes through slot0 only; give the kernel a non-null table pointer it never uses.
  float* tab = tables ? tables : slot0;
  // pair arrays addressable with 32-bit offsets: the short kernel
  CTR_REQUIRE(!fm || (G <= 8 && fits_buffer(n_items * 4 + 4)), "ctr_emb_sparse_apply_fm: at most 2^30 pairs");
  if (fm) {
    const DivU32 divN = make_div((uint32_t)N);
#define LEANFM(GG)                                                                                              \
  hipLaunchKernelGGL((sparse_apply_lean_kernel<GG, true>), grid, block, 0, st, keys_sorted, src_sorted, n_items, N, divN, \
                     total_rows, dz, dz_ld, src_scale, tab, slot0, slot1, slot_ld, o, part, part_head, head_key, g2,      \
                     g2_ld, part2, part_head2);                                                                           \
  hipLaunchKernelGGL((sparse_apply_tail_kernel<GG, true>), grid2, block, 0, st, keys_sorted, n_items, nb, E, tab, slot0,  \
                     slot1, slot_ld, o, part, part_head, head_key, part2, part_head2)
    switch (G) {
      case 1: LEANFM(1); break;
      case 2: LEANFM(2); break;
      case 4: LEANFM(4); break;
      case 8: LEANFM(8); break;
    }
#undef LEANFM
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
  if (G <= 8 && fits_buffer(n_items * 4 + 4) && tables) {
    const DivU32 divN = make_div((uint32_t)N);
#define LEAN(GG)                                                                                          \
  hipLaunchKernelGGL(sparse_apply_lean_kernel<GG>, grid, block, 0, st, keys_sorted, src_sorted, n_items, N, divN,      \
                     total_rows, dz, dz_ld, src_scale, tab, slot0, slot1, slot_ld, o, part, part_head, head_key,       \
                     (const float*)nullptr, (int64_t)0, (float*)nullptr, (float*)nullptr);                             \
  hipLaunchKernelGGL(sparse_apply_tail_kernel<GG>, grid2, block, 0, st, keys_sorted, n_items, nb, E, tab, \
                     slot0, slot1, slot_ld, o, part, part_head, head_key, (const float*)nullptr, (const float*)nullptr)
    switch (G) {
      case 1: LEAN(1); break;
      case 2: LEAN(2); break;
      case 4: LEAN(4); break;
      case 8: LEAN(8); break;
    }
#undef LEAN
    CTR_LAUNCH_CHECK();
    return CTR_OK;
  }
#define LAUNCH(GG)                                                                                   \
  hipLaunchKernelGGL(sparse_apply_kernel<GG>, grid, block, 0, st, keys_sorted, src_sorted, n_items, N, \
                     E, total_rows, dz, dz_ld, src_scale, tab, slot0, slot1, slot_ld, o, part, part_head, head_key); \
  hipLaunchKernelGGL(sparse_apply_tail_kernel<GG>, grid2, block, 0, st, keys_sorted, n_items, nb, E, tab, \
                     slot0, slot1, slot_ld, o, part, part_head, head_key, (const float*)nullptr, (const float*)nullptr)
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
  return CTR_OK;
}

extern "C" int ctr_dense_update(float* param, const float* grad, float* slot0, float* slot1, int64_t n,
                                const ctr_opt_t* opt, void* stream) {
  CTR_REQUIRE(opt && n >= 0, "ctr_dense_update: bad argument");
  if (n == 0) return CTR_OK;
  CTR_REQUIRE(param && grad, "ctr_dense_update: null pointer");
  CTR_REQUIRE(opt->kind == CTR_OPT_ADAM || opt->kind == CTR_OPT_ADAGRAD || opt->kind == CTR_OPT_SGD ||
                  opt->kind == CTR_OPT_FTRL,
              "ctr_dense_update: unknown optimizer kind %d", opt->kind);
  CTR_REQUIRE(opt->kind != CTR_OPT_ADAM || (slot0 && slot1), "ctr_dense_update: Adam needs m and v");
  CTR_REQUIRE(opt->kind != CTR_OPT_FTRL || (slot0 && slot1), "ctr_dense_update: FTRL needs n and z");
  CTR_REQUIRE(opt->kind != CTR_OPT_ADAGRAD || slot0, "ctr_dense_update: Adagrad needs the accumulator");
  const int64_t blocks = cdiv(n, 256) < 8192 ? cdiv(n, 256) : 8192;
  hipLaunchKernelGGL(dense_update_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                     grad, slot0, slot1, n, make_opt(opt));
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

// Per-step scalars of a replayed graph, written by a one-thread kernel from by-value arguments: stream-ordered
// (no host buffer whose lifetime must outlast an asynchronous copy) and cheaper than two pageable copies.
__global__ void set_step_words_kernel(uint64_t* seed_add, uint64_t seed_add_value, float* lr_dev, float lr_value) {
  if (seed_add) *seed_add = seed_add_value;
  if (lr_dev) *lr_dev = lr_value;
}

extern "C" int ctr_set_step_words(uint64_t* seed_add, uint64_t seed_add_value, float* lr_dev, float lr_value,
                                  void* stream) {
  if (!seed_add && !lr_dev) return CTR_OK;
  hipLaunchKernelGGL(set_step_words_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed_add, seed_add_value,
                     lr_dev, lr_value);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_mark_rows(const uint32_t* keys_sorted, int64_t n_items, int64_t total_rows, uint8_t* touched, int value,
                             void* stream) {
  CTR_REQUIRE(n_items >= 0 && total_rows > 0, "ctr_mark_rows: bad size");
  if (n_items == 0) return CTR_OK;
  CTR_REQUIRE(keys_sorted && touched, "ctr_mark_rows: null pointer");
  const int64_t blocks = cdiv(n_items, 256) < 4096 ? cdiv(n_items, 256) : 4096;
  if (value)
    hipLaunchKernelGGL(mark_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, keys_sorted, n_items,
                       total_rows, touched);
  else
    hipLaunchKernelGGL(unmark_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, keys_sorted, n_items,
                       total_rows, touched);
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}

extern "C" int ctr_adam_untouched_rows(float* tables, float* m, float* v, int64_t rows, int E, uint8_t* touched,
                                       const ctr_opt_t* opt, void* stream) {
  CTR_REQUIRE(opt && opt->kind == CTR_OPT_ADAM, "ctr_adam_untouched_rows: the optimizer must be Adam");
  CTR_REQUIRE(rows >= 0 && E >= 4 && E % 4 == 0, "ctr_adam_untouched_rows: bad size (E a multiple of 4)");
  if (rows == 0) return CTR_OK;
  CTR_REQUIRE(tables && m && v && touched, "ctr_adam_untouched_rows: null pointer");
  CTR_REQUIRE(((uintptr_t)tables % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
              "ctr_adam_untouched_rows: 16-byte alignment");
  CTR_REQUIRE(((uintptr_t)touched % 16) == 0, "ctr_adam_untouched_rows: the row map must be 16-byte aligned");
  CTR_REQUIRE(rows < ((int64_t)1 << 40) && (int64_t)AU_ROWS * (E / 4) < 0xFFFFFFFFll, "ctr_adam_untouched_rows: size");
  const int64_t chunks = cdiv(rows, AU_ROWS);
  const int64_t blocks = cdiv(chunks, 4) < 32768 ? cdiv(chunks, 4) : 32768;
  hipLaunchKernelGGL(adam_untouched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tables, m, v, rows,
                     E / 4, touched, make_opt(opt));
  CTR_LAUNCH_CHECK();
  return CTR_OK;
}
