// Error reporting and version of libctrhip.
#include "common.h"
#include <string.h>

namespace ctr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ctr

extern "C" const char* ctr_last_error(void) { return ctr::g_err; }
extern "C" int ctr_version(void) { return 100; }
extern "C" uint32_t ctr_keep_hash24(uint64_t seed, uint64_t stream_id, uint64_t index) {
  return ctr::hash24(seed, stream_id, index);
}
