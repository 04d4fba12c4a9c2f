// Probe: do kernels of two HIP streams run at the same time on this GPU?  Each kernel uses `wgs` workgroups that spin
// for a fixed number of clock ticks; N launches per stream.  hipcc --offload-arch=gfx950 -O3 two_streams.hip -o two_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long ticks, int *out) {
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (ticks < 0) out[0] = 1;
}
static double run(int nstreams, int wgs, int n, long ticks, int *out, unsigned flags) {
  hipStream_t s[2];
  for (int i = 0; i < nstreams; ++i) hipStreamCreateWithFlags(&s[i], flags);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n; ++k)
    for (int i = 0; i < nstreams; ++i) spin<<<wgs, 256, 0, s[i]>>>(ticks, out);
  hipDeviceSynchronize();
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (int i = 0; i < nstreams; ++i) hipStreamDestroy(s[i]);
  return ms;
}
int main() {
  int *out; hipMalloc(&out, 64);
  const long ticks = 2000;   // 100 MHz wall clock: 20 us
  for (int wgs : {64, 256, 1024, 4096}) {
    run(1, wgs, 50, ticks, out, hipStreamNonBlocking);
    double a = run(1, wgs, 200, ticks, out, hipStreamNonBlocking), b = run(2, wgs, 200, ticks, out, hipStreamNonBlocking);
    printf("%5d workgroups x 20 us, 200 launches per stream: 1 stream %.2f ms, 2 streams %.2f ms (x%.2f)\n", wgs, a, b, b / a);
  }
  return 0;
}
