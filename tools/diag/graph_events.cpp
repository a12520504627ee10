// Can a replayed hipGraph carry timed events?  ROCm 7.2 / gfx950 findings (run on the GPU box):
//  (a) hipEventRecord inside a stream capture: records without error, hipEventElapsedTime -> invalid resource handle;
//  (b) explicit hipGraphAddEventRecordNode on a hand-built graph: works (elapsed == the eager figure);
//  (c) the same node added to a graph UNDER CAPTURE (hipStreamGetCaptureInfo_v2 + hipStreamUpdateCaptureDependencies), and
//      hipGraphExecEventRecordNodeSetEvent to give every replay its own event pair: tested below.
// Build: hipcc --offload-arch=gfx950 tools/diag/graph_events.cpp -o tools/diag/graph_events.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess || verbose) printf("%-70s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
static bool verbose = true;
__global__ void spin(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f; p[threadIdx.x] = a; }

static hipGraphNode_t add_record(hipStream_t s, hipEvent_t ev) {
  hipStreamCaptureStatus st; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t nd = 0;
  CK(hipStreamGetCaptureInfo_v2(s, &st, &id, &g, &deps, &nd));
  printf("   capture status %d, graph %p, %zu deps\n", (int)st, (void*)g, nd);
  hipGraphNode_t node = nullptr;
  CK(hipGraphAddEventRecordNode(&node, g, deps, nd, ev));
  CK(hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies));
  return node;
}

int main() {
  float* d; (void)hipMalloc(&d, 4096);
  hipStream_t s; (void)hipStreamCreate(&s);
  const int K = 4;
  hipEvent_t e0[K], e1[K];
  for (int i = 0; i < K; ++i) { (void)hipEventCreate(&e0[i]); (void)hipEventCreate(&e1[i]); }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  spin<<<1, 256, 0, s>>>(d, 1000);
  hipGraphNode_t r0 = add_record(s, e0[0]);
  spin<<<256, 256, 0, s>>>(d, 200000);
  hipGraphNode_t r1 = add_record(s, e1[0]);
  spin<<<1, 256, 0, s>>>(d, 1000);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int r = 0; r < K; ++r) {
    CK(hipGraphExecEventRecordNodeSetEvent(ge, r0, e0[r]));
    CK(hipGraphExecEventRecordNodeSetEvent(ge, r1, e1[r]));
    CK(hipGraphLaunch(ge, s));
    verbose = false;
  }
  verbose = true;
  CK(hipStreamSynchronize(s));
  for (int r = 0; r < K; ++r) { float ms = -1; CK(hipEventElapsedTime(&ms, e0[r], e1[r])); printf("replay %d: elapsed %.3f us\n", r, ms * 1e3); }
  (void)hipEventRecord(e0[0], s); spin<<<256, 256, 0, s>>>(d, 200000); (void)hipEventRecord(e1[0], s); (void)hipStreamSynchronize(s);
  float ms = -1; (void)hipEventElapsedTime(&ms, e0[0], e1[0]); printf("eager: elapsed %.3f us\n", ms * 1e3);
  return 0;
}
