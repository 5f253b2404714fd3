// host cost of ONE kernel launch on this box, as the training step pays it ~4 200 times (host enqueue):
//   (a) an empty kernel with 16 / 64 / 320 bytes of arguments through hipLaunchKernelGGL,
//   (b) the library's own entry points (vm_add on 4 KiB: the C wrapper + VM_LAUNCH_CHECK),
// enqueue loops of N launches on one stream, the stream idle at the start and synchronised at the end; reported: host microseconds per
// launch of the enqueue loop alone, and the GPU's own pace (total / N) — if the second is the larger, the loop measured a full queue.
// usage: launch_bench [N = 20000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../../include/vividmed_hip.h"

struct P16 { void* p; int n; int pad; };
struct P64 { void* p[8]; };
struct P320 { void* p[40]; };
template <class P> __global__ void empty_k(const P a) { if (threadIdx.x == 9999) ((int*)a.p)[0] = 1; }
__global__ void empty16_k(const P16 a) { if (threadIdx.x == 9999) ((int*)a.p)[0] = 1; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F> static void run(const char* name, int n, hipStream_t s, F&& f) {
  for (int i = 0; i < 200; ++i) f();
  (void)hipStreamSynchronize(s);
  const double t0 = now();
  for (int i = 0; i < n; ++i) f();
  const double t1 = now();
  (void)hipStreamSynchronize(s);
  const double t2 = now();
  printf("%-58s host %6.2f us per launch   (enqueue + drain %6.2f us per launch)\n", name, (t1 - t0) / n, (t2 - t0) / n);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  void* buf; (void)hipMalloc(&buf, 1 << 20);
  P16 a16{buf, 0, 0}; P64 a64{}; a64.p[0] = buf; P320 a320{}; a320.p[0] = buf;
  run("empty kernel, 16 B of arguments, 1 x 64 threads", n, s, [&] { hipLaunchKernelGGL(empty16_k, dim3(1), dim3(64), 0, s, a16); });
  run("empty kernel, 64 B of arguments", n, s, [&] { hipLaunchKernelGGL(empty_k<P64>, dim3(1), dim3(64), 0, s, a64); });
  run("empty kernel, 320 B of arguments", n, s, [&] { hipLaunchKernelGGL(empty_k<P320>, dim3(1), dim3(64), 0, s, a320); });
  run("empty kernel, 320 B, 256 workgroups x 256 threads", n, s, [&] { hipLaunchKernelGGL(empty_k<P320>, dim3(256), dim3(256), 0, s, a320); });
  run("empty kernel, 320 B, 64 KiB dynamic LDS", n, s, [&] { hipLaunchKernelGGL(empty_k<P320>, dim3(256), dim3(256), 65536, s, a320); });
  run("vm_add (bf16, 2048 elements) through the C ABI", n, s, [&] { vm_add(buf, buf, (char*)buf + 65536, 2048, 0, s); });
  run("vm_cast (bf16 -> f32, 2048 elements)", n, s, [&] { vm_cast(buf, 0, (char*)buf + 65536, 1, 2048, s); });
  // the default (NULL) stream, as a reference
  run("empty kernel, 64 B, NULL stream", n, nullptr, [&] { hipLaunchKernelGGL(empty_k<P64>, dim3(1), dim3(64), 0, nullptr, a64); });
  // with an event record between launches (what bench.py's sampled kernel events add)
  hipEvent_t ev; (void)hipEventCreate(&ev);
  run("empty kernel, 64 B + hipEventRecord", n, s, [&] { hipLaunchKernelGGL(empty_k<P64>, dim3(1), dim3(64), 0, s, a64); (void)hipEventRecord(ev, s); });
  return 0;
}
