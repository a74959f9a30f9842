// Stand-alone check of the HIP runtime (no torch): does a stream capture survive (a) events that are destroyed while the capture
// is still open, (b) cross-stream waits issued from a second host thread in relaxed mode?   hipcc -o capture_events capture_events.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void k(float* p) { p[threadIdx.x] += 1.0f; }

static void fork_join(hipStream_t a, hipStream_t b, float* p, bool destroy_now, std::vector<hipEvent_t>& later) {
  hipEvent_t e1, e2;
  CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventRecord(e1, a)); CK(hipStreamWaitEvent(b, e1, 0));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, b, p);
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  CK(hipEventRecord(e2, b)); CK(hipStreamWaitEvent(a, e2, 0));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, a, p);
  if (destroy_now) { CK(hipEventDestroy(e1)); CK(hipEventDestroy(e2)); void* junk = malloc(4096); free(junk); }
  else { later.push_back(e1); later.push_back(e2); }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;   // bit 0: destroy events inside the capture, bit 1: second thread issues half of the work
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  std::vector<hipEvent_t> later;
  hipGraph_t graph;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
  for (int i = 0; i < 100; ++i) fork_join(s0, s1, p, mode & 1, later);
  if (mode & 2) {
    std::thread t([&] { for (int i = 0; i < 100; ++i) { fork_join(s0, s2, p, mode & 1, later); fork_join(s1, s2, p, mode & 1, later); } });
    t.join();
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventRecord(e, s1)); CK(hipStreamWaitEvent(s0, e, 0)); later.push_back(e);
  }
  printf("mode %d: ending capture\n", mode); fflush(stdout);
  CK(hipStreamEndCapture(s0, &graph));
  for (hipEvent_t e : later) CK(hipEventDestroy(e));
  hipGraphExec_t exec; CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, s0)); CK(hipStreamSynchronize(s0));
  float h = 0; CK(hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost));
  printf("mode %d: ok, p[0] = %.0f\n", mode, h);
  return 0;
}
