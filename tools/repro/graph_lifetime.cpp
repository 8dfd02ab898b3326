// Stand-alone check (no torch) of the HIP-graph lifetime pattern the training step uses: a stream capture
// with a fork/join over a second stream (two events), instantiated, replayed, and then DESTROYED before
// the next, differently shaped capture is instantiated and launched.  Mirrors what happens when a process
// captures one model after another and the older model's graph is garbage-collected.
//   graph_lifetime destroy   every graph exec is destroyed (then a device sync, as torch's destructor does)
//   graph_lifetime nosync    destroyed without the sync
//   graph_lifetime keep      all graph execs are kept to the end
//   graph_lifetime late      graph r-1 is destroyed AFTER graph r was captured and instantiated, BEFORE r's first
//                            launch (the order a garbage collector produces); latesync adds the device sync
// Prints one line per round; exit code 0 when every replay produced the expected sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)

struct Big { float scale; int pad[31]; const float* indirect; };   // a by-value struct argument, as the real kernels take

__global__ void add_kernel(float* x, int n, Big p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += p.scale * (p.indirect ? *p.indirect : 1.f);
}

struct Captured { hipGraphExec_t exec; hipStream_t side; hipEvent_t fork, join; int main_k, side_k; };

static int capture(hipStream_t cap, float* a, float* b, const float* word, int n, int main_k, int side_k, Captured* out) {
  out->main_k = main_k; out->side_k = side_k;
  CK(hipStreamCreateWithFlags(&out->side, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&out->fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&out->join, hipEventDisableTiming));
  Big p; memset(&p, 0, sizeof(p)); p.scale = 1.f; p.indirect = word;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal));
  for (int k = 0; k < main_k / 2; ++k) hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, a, n, p);
  CK(hipEventRecord(out->fork, cap));
  CK(hipStreamWaitEvent(out->side, out->fork, 0));
  for (int k = 0; k < side_k; ++k) hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, out->side, b, n, p);
  for (int k = main_k / 2; k < main_k; ++k) hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, a, n, p);
  CK(hipEventRecord(out->join, out->side));
  CK(hipStreamWaitEvent(cap, out->join, 0));
  hipLaunchKernelGGL(add_kernel, dim3((n + 255) / 256), dim3(256), 0, cap, a, n, p);
  hipGraph_t g;
  CK(hipStreamEndCapture(cap, &g));
  CK(hipGraphInstantiateWithFlags(&out->exec, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
  CK(hipGraphDestroy(g));
  return 0;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "destroy";
  const int rounds = argc > 2 ? atoi(argv[2]) : 24;
  const int n = 1 << 16;
  float *a, *b, *word;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&word, 4));
  const float one = 1.f;
  CK(hipMemcpy(word, &one, 4, hipMemcpyHostToDevice));
  hipStream_t cap, run;
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
  std::vector<Captured> live;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    Captured c;
    const int main_k = 6 + 7 * (r % 5), side_k = 2 + 3 * (r % 4);
    CK(hipMemsetAsync(a, 0, n * 4, run)); CK(hipMemsetAsync(b, 0, n * 4, run));
    CK(hipStreamSynchronize(run));
    if (capture(cap, a, b, word, n, main_k, side_k, &c)) return 2;
    if (!strncmp(mode, "late", 4) && !live.empty()) {
      Captured& o = live.back();
      CK(hipGraphExecDestroy(o.exec));
      CK(hipEventDestroy(o.fork)); CK(hipEventDestroy(o.join)); CK(hipStreamDestroy(o.side));
      if (!strcmp(mode, "latesync")) CK(hipDeviceSynchronize());
      live.pop_back();
    }
    const int replays = 3;
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(c.exec, run));
    float ha = -1, hb = -1;
    CK(hipMemcpyAsync(&ha, a + (n - 1), 4, hipMemcpyDeviceToHost, run));
    CK(hipMemcpyAsync(&hb, b + 7, 4, hipMemcpyDeviceToHost, run));
    CK(hipStreamSynchronize(run));
    const bool ok = ha == (float)(replays * (main_k + 1)) && hb == (float)(replays * side_k);
    bad += !ok;
    printf("round %2d  main %2d side %2d  a=%g b=%g  %s\n", r, main_k, side_k, ha, hb, ok ? "ok" : "WRONG");
    fflush(stdout);
    if (!strcmp(mode, "keep") || !strncmp(mode, "late", 4)) { live.push_back(c); continue; }
    CK(hipGraphExecDestroy(c.exec));
    CK(hipEventDestroy(c.fork)); CK(hipEventDestroy(c.join)); CK(hipStreamDestroy(c.side));
    if (!strcmp(mode, "destroy")) CK(hipDeviceSynchronize());
  }
  for (auto& c : live) { CK(hipGraphExecDestroy(c.exec)); CK(hipEventDestroy(c.fork)); CK(hipEventDestroy(c.join)); CK(hipStreamDestroy(c.side)); }
  CK(hipDeviceSynchronize());
  printf("mode %s: %d rounds, %d wrong\n", mode, rounds, bad);
  return bad ? 1 : 0;
}
