// HIP-only reproducer attempt for the order-dependent hipGraphLaunch crash of the test suite: process-lifetime side
// streams + events (what csrc/engine.hip's attn_side() / event_pool() hold) reused by MANY stream captures whose graphs are destroyed
// in between, against fresh events per capture.   hipcc --offload-arch=gfx950 -O2 repro_graph_events.cpp -o repro_graph_events
//   ./repro_graph_events static  |  ./repro_graph_events fresh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void work(float* p, int n, float a) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i] * a + 1.0f; }
int main(int argc, char** argv) {
  const bool fresh = argc > 1 && !strcmp(argv[1], "fresh");
  const int N = 1 << 22, LAYERS = 12, ROUNDS = 40;
  float* buf[4]; for (auto& b : buf) { CK(hipMalloc(&b, N * 4)); CK(hipMemset(b, 0, N * 4)); }
  hipStream_t origin, side[3];
  CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<hipEvent_t> pool(64);
  for (auto& e : pool) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  std::vector<hipGraphExec_t> keep;      // every third executable graph stays alive (like engines of earlier tests still referenced)
  for (int r = 0; r < ROUNDS; ++r) {
    if (fresh) { for (auto& e : pool) { CK(hipEventDestroy(e)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); } }
    // a few eager passes through the same streams / events first (engines warm up eagerly before they capture)
    for (int pass = 0; pass < 2; ++pass) {
      const bool cap = pass == 1;
      hipGraph_t g = nullptr;
      if (cap) CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
      int ev = 0;
      for (int l = 0; l < LAYERS; ++l) {
        work<<<N / 256, 256, 0, origin>>>(buf[0], N, 1.0001f);
        CK(hipEventRecord(pool[ev], origin));                       // fork: two row chains + the weight-gradient stream
        for (int k = 0; k < 3; ++k) CK(hipStreamWaitEvent(side[k], pool[ev], 0));
        ++ev;
        for (int k = 0; k < 3; ++k) work<<<N / 256, 256, 0, side[k]>>>(buf[1 + k], N, 1.0002f);
        work<<<N / 256, 256, 0, origin>>>(buf[0], N, 0.9999f);
        for (int k = 0; k < 3; ++k) { CK(hipEventRecord(pool[ev], side[k])); CK(hipStreamWaitEvent(origin, pool[ev], 0)); ++ev; }
      }
      if (cap) {
        CK(hipStreamEndCapture(origin, &g));
        hipGraphExec_t x;
        CK(hipGraphInstantiate(&x, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(x, origin));
        CK(hipStreamSynchronize(origin));
        CK(hipGraphDestroy(g));
        if (r % 3 == 0) keep.push_back(x); else CK(hipGraphExecDestroy(x));
      } else {
        CK(hipDeviceSynchronize());
      }
    }
    if (r % 10 == 9) { for (auto x : keep) CK(hipGraphLaunch(x, origin)); CK(hipStreamSynchronize(origin)); printf("round %d ok (%zu graphs alive)\n", r, keep.size()); fflush(stdout); }
  }
  printf("%s events: %d capture rounds survived\n", fresh ? "fresh" : "static", ROUNDS);
  return 0;
}
