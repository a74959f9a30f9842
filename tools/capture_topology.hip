// Stand-alone reproducer for the hipStreamEndCapture stack overflow met by a captured training step: replays a
// sequence of cross-stream waits inside one relaxed-mode capture that stream M begins and ends.
//   ./capture_topology "S<M A<S S<A M<S"        token X<Y: kernel on Y, event recorded on Y, X waits for it, kernel on X
// Streams are single letters; M is the origin.  Exit code 0 = captured, instantiated, replayed; a crash is the finding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void k(float* p) { atomicAdd(p, 1.0f); }

int main(int argc, char** argv) {
  std::map<char, hipStream_t> st;
  auto stream = [&](char c) { if (!st.count(c)) { hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); st[c] = s; } return st[c]; };
  float* p; CK(hipMalloc(&p, 256)); CK(hipMemset(p, 0, 256));
  std::istringstream seq(argc > 1 ? argv[1] : "S<M M<S");
  std::string tok;
  hipStream_t M = stream('M');
  { std::istringstream pre(argc > 1 ? argv[1] : ""); while (pre >> tok) { stream(tok[0]); stream(tok[2]); } }   // create every stream before the capture
  CK(hipStreamBeginCapture(M, hipStreamCaptureModeRelaxed));
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, M, p);
  int n = 1;
  while (seq >> tok) {
    hipStream_t x = stream(tok[0]), y = stream(tok[2]);
    hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, y, p);
    CK(hipEventRecord(e, y)); CK(hipStreamWaitEvent(x, e, 0));
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, x, p);
    n += 2;
  }
  printf("ending capture of %d kernels over %zu streams\n", n, st.size()); fflush(stdout);
  hipGraph_t graph;
  hipError_t e = hipStreamEndCapture(M, &graph);
  if (e != hipSuccess) { printf("hipStreamEndCapture: %s\n", hipGetErrorString(e)); return 3; }
  hipGraphExec_t exec; CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CK(hipGraphLaunch(exec, M)); CK(hipStreamSynchronize(M));
  float h = 0; CK(hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost));
  printf("ok: %.0f of %d kernels ran\n", h, n);
  return h == n ? 0 : 4;
}
