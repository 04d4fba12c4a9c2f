// What does the FIRST batch of loads behind a kernel boundary cost on this part?  (the fused decoder
// kernels spend 8-13 us in their first memory round trip.)   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/boundary_probe.hip -o /tmp/boundary_probe && /tmp/boundary_probe
// A producer kernel (256 workgroups x 1024 threads) leaves `kb` KB per workgroup dirty; the consumer kernel (same grid) has
// every thread load its share of `kb` KB and stamps wall_clock64() (100 MHz) around that one round trip.  Modes:
//   same    : the consumer's workgroup i reads what the producer's workgroup i wrote (same XCD: id mod 8)
//   other   : ... what workgroup i + 1 wrote (another XCD)
//   all8    : workgroup i reads the slice of workgroup (i / 8) * 8 + j for its j-th eighth (8 XCDs' data, like a row tile
//             whose rows come from 8 streams)
//   readonly: a buffer no kernel has written since the start (weights); the producer still runs in front
//   rerun   : readonly, consumer launched twice back to back without a producer: the second launch's time
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void producer(float4 *p, int n4) {   // n4 float4 per workgroup
  float4 *d = p + (long)blockIdx.x * n4;
  for (int i = threadIdx.x; i < n4; i += 1024) d[i] = make_float4(i, blockIdx.x, 1.f, 2.f);
}

__global__ __launch_bounds__(1024) void consumer(const float4 *p, int n4, int mode, long long *ticks, float *sink) {
  const int b = blockIdx.x;
  const long long t0 = wall_clock64();
  float acc = 0.f;
  constexpr int U = 8;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 1024 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 1024;
      int src = b;
      if (mode == 1) src = (b + 1) % gridDim.x;
      if (mode == 2) src = (b / 8) * 8 + (i * 8 / n4);
      v[u] = i < n4 ? p[(long)src * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 123456.789f) sink[0] = acc;
  __syncthreads();
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[b] = t1 - t0;
}

int main() {
  const int G = 256;
  for (int kb : {16, 40, 150}) {
    const int n4 = kb * 1024 / 16;
    float4 *P, *R;
    long long *T;
    float *sink;
    hipMalloc(&P, (size_t)G * n4 * 16);
    hipMalloc(&R, (size_t)G * n4 * 16);
    hipMalloc(&T, G * sizeof(long long));
    hipMalloc(&sink, 4);
    hipMemset(R, 0, (size_t)G * n4 * 16);
    hipDeviceSynchronize();
    const char *names[] = {"same", "other", "all8", "readonly", "rerun"};
    for (int mode = 0; mode < 5; ++mode) {
      std::vector<double> med;
      for (int rep = 0; rep < 30; ++rep) {
        if (mode < 4) producer<<<G, 1024>>>(P, n4);
        if (mode == 4) consumer<<<G, 1024>>>(R, n4, 0, T, sink);
        consumer<<<G, 1024>>>(mode >= 3 ? R : P, n4, mode >= 3 ? 0 : mode, T, sink);
        hipDeviceSynchronize();
        std::vector<long long> h(G);
        hipMemcpy(h.data(), T, G * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        if (rep >= 5) med.push_back(h[G / 2] / 100.0);
      }
      std::sort(med.begin(), med.end());
      printf("%4d KB per workgroup (%5.1f MB per launch)  %-8s  median workgroup: %6.2f us for the round trip (runs: %5.2f .. %5.2f)\n", kb,
             G * kb / 1024.0, names[mode], med[med.size() / 2], med.front(), med.back());
    }
    hipFree(P); hipFree(R); hipFree(T); hipFree(sink);
  }
  return 0;
}
