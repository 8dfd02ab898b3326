// Criteo display-advertising text -> the three arrays a training step is fed with, on the host's cores.
// A line is  label \t I1 .. I13 \t C1 .. C26 \n : the label 0/1, thirteen integer counts and twenty-six
// categorical values written as up to eight hexadecimal digits; any field may be empty.  At 10^7 - 10^8
// samples per second the GPU path consumes 156 bytes per sample, so the text has to be turned into ids at
// several GB/s: the buffer is cut into one piece per thread at line boundaries, every piece counts its lines,
// then parses them into its slice of the outputs (no locks, no allocation per line).
//   dense[r][j]  = I_j as a float (empty: 0), or log(1 + max(I_j, 0)) with log_dense
//   ids[r][f]    = value(C_f) mod cards[f], value = the hexadecimal number (empty: 0): the "hashing trick" with
//                  the token's own 32 bits as the hash, so a table of cards[f] rows serves any vocabulary
//   label[r]     = 0 or 1
// A line with fewer than 40 fields has its missing trailing fields empty; extra fields are ignored.
#include "common.h"
#include <math.h>
#include <string.h>
#include <thread>
#include <vector>

namespace {

constexpr int N_DENSE = 13, N_CATE = 26;

// value of a hexadecimal digit (anything else counts as 0)
struct HexTable {
  uint8_t v[256];
  HexTable() {
    memset(v, 0, sizeof(v));
    for (int c = '0'; c <= '9'; ++c) v[c] = (uint8_t)(c - '0');
    for (int c = 'a'; c <= 'f'; ++c) v[c] = (uint8_t)(c - 'a' + 10);
    for (int c = 'A'; c <= 'F'; ++c) v[c] = (uint8_t)(c - 'A' + 10);
  }
};
const HexTable g_hex;

// a mod d for 32-bit a and d by two multiplications (M = ceil(2^64 / d)): exact for every 32-bit a
inline uint32_t fastmod(uint32_t a, uint64_t M, uint32_t d) {
  const uint64_t low = M * a;
  return (uint32_t)(((__uint128_t)low * d) >> 64);
}

inline const char* line_end(const char* p, const char* end) {
  const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
  return q ? q : end;
}

int64_t count_lines(const char* p, const char* end) {
  int64_t n = 0;
  while (p < end) {
    const char* q = line_end(p, end);
    if (q > p && !(q - p == 1 && *p == '\r')) ++n;      // empty lines carry no sample
    p = q + 1;
  }
  return n;
}

void parse_piece(const char* p, const char* end, const int64_t* cards, int log_dense, float* label, float* dense,
                 int32_t* ids, int64_t max_rows) {
  uint64_t M[N_CATE];
  uint32_t D[N_CATE];
  for (int f = 0; f < N_CATE; ++f) {
    D[f] = (uint32_t)cards[f];
    M[f] = UINT64_MAX / D[f] + 1;
  }
  int64_t r = 0;
  while (p < end && r < max_rows) {
    const char* q = line_end(p, end);
    const char* le = (q > p && q[-1] == '\r') ? q - 1 : q;
    if (le == p) { p = q + 1; continue; }
    const char* c = p;
    // label
    int lab = 0;
    while (c < le && *c != '\t') { if (*c >= '0' && *c <= '9') lab = lab * 10 + (*c - '0'); ++c; }
    label[r] = lab ? 1.f : 0.f;
    if (c < le) ++c;
    for (int j = 0; j < N_DENSE; ++j) {
      bool neg = false, any = false;
      int64_t v = 0;
      if (c < le && *c == '-') { neg = true; ++c; }
      while (c < le && *c != '\t') { if (*c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); any = true; } ++c; }
      if (c < le) ++c;
      if (neg) v = -v;
      float x = any ? (float)v : 0.f;
      if (log_dense) x = log1pf(x > 0.f ? x : 0.f);
      dense[r * N_DENSE + j] = x;
    }
    for (int f = 0; f < N_CATE; ++f) {
      uint32_t h = 0;
      while (c < le && *c != '\t') {
        h = (h << 4) | g_hex.v[(unsigned char)*c];
        ++c;
      }
      if (c < le) ++c;
      ids[r * N_CATE + f] = (int32_t)(D[f] == 1u ? 0u : fastmod(h, M[f], D[f]));
    }
    ++r;
    p = q + 1;
  }
}

}  // namespace

extern "C" int64_t ctr_criteo_tsv_count(const char* text, int64_t len) {
  if (!text || len <= 0) return 0;
  return count_lines(text, text + len);
}

extern "C" int ctr_criteo_tsv_parse(const char* text, int64_t len, const int64_t* cards, int log_dense, int n_threads,
                                    int64_t max_rows, float* label, float* dense, int32_t* ids, int64_t* rows_out) {
  CTR_REQUIRE(rows_out, "ctr_criteo_tsv_parse: rows_out is null");
  *rows_out = 0;
  CTR_REQUIRE(len >= 0 && max_rows >= 0, "ctr_criteo_tsv_parse: bad size");
  if (len == 0 || max_rows == 0) return CTR_OK;
  CTR_REQUIRE(text && cards && label && dense && ids, "ctr_criteo_tsv_parse: null pointer");
  for (int f = 0; f < N_CATE; ++f)
    CTR_REQUIRE(cards[f] > 0 && cards[f] <= 0x7FFFFFFFll, "ctr_criteo_tsv_parse: cards[%d] = %lld must be in (0, 2^31)", f,
                (long long)cards[f]);
  int nt = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
  if (len < (int64_t)nt * 4096) nt = 1;
  // pieces start right after a newline
  std::vector<const char*> cut(nt + 1);
  const char* end = text + len;
  cut[0] = text;
  cut[nt] = end;
  for (int t = 1; t < nt; ++t) {
    const char* p = text + len / nt * t;
    if (p < cut[t - 1]) p = cut[t - 1];
    const char* q = line_end(p, end);
    cut[t] = q < end ? q + 1 : end;
  }
  std::vector<int64_t> cnt(nt, 0);
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { cnt[t] = count_lines(cut[t], cut[t + 1]); });
    cnt[0] = count_lines(cut[0], cut[1]);
    for (auto& x : th) x.join();
  }
  std::vector<int64_t> first(nt + 1, 0);
  for (int t = 0; t < nt; ++t) first[t + 1] = first[t] + cnt[t];
  auto run = [&](int t) {
    const int64_t lo = first[t] < max_rows ? first[t] : max_rows;
    const int64_t room = max_rows - lo;
    if (room > 0)
      parse_piece(cut[t], cut[t + 1], cards, log_dense, label + lo, dense + lo * N_DENSE, ids + lo * N_CATE,
                  room < cnt[t] ? room : cnt[t]);
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& x : th) x.join();
  }
  *rows_out = first[nt] < max_rows ? first[nt] : max_rows;
  return CTR_OK;
}
