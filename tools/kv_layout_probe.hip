// Does a head-group-major K|V layout help the attention walks?  (Next 1 (c).)
//   hipcc --offload-arch=gfx950 -O3 tools/kv_layout_probe.hip -o /tmp/kv_layout_probe && timeout 120 /tmp/kv_layout_probe
// The cross-attention walk of a full bucket: 128 streams x 2 workgroups (four heads each) x 1024 threads; a stream has T rows
// of 2 KB = [K of 8 heads | V of 8 heads] (d = 256, fp32), a workgroup needs the K and V columns of ITS four heads of
// every row: two 512-byte runs 1 KB apart, rows 2 KB apart (layout "rows"; what csrc/decoder_layer.hip reads today).
// Layout "groups" stores [K of heads 0-3 | V of heads 0-3] of all rows, then the same for heads 4-7: 1 KB contiguous per
// row and workgroup, consecutive rows adjacent.  One wave instruction = 64 lanes x 16 B = one row's 1 KB in both layouts.
// Prints the time and GB/s of both (cold: 768 MB of other data read in between), T = 790 and 400.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool GROUPS>
__global__ __launch_bounds__(1024) void walk_k(const float4 *kv, int T, float *sink) {
  const int s = blockIdx.x >> 1, hg = blockIdx.x & 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 *base = kv + (long)s * T * 128;   // 2 KB = 128 float4 per row
  float4 acc = make_float4(0, 0, 0, 0);
  for (int t = wave; t < T; t += 64) {   // 4 rows of a wave in flight
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = t + 16 * u;
      const int tc = tt < T ? tt : T - 1;
      long off;
      if (GROUPS) off = (long)hg * T * 64 + (long)tc * 64 + lane;                       // 1 KB per row, rows adjacent
      else off = (long)tc * 128 + (lane >> 5) * 64 + hg * 32 + (lane & 31);              // K run | V run of the head group
      v[u] = base[off];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y + v[u].z + v[u].w; }
  }
  if (acc.x + acc.y == 12345.678f) sink[0] = acc.x;
}

__global__ __launch_bounds__(1024) void flush_k(const float4 *buf, long n4, float *sink) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long)gridDim.x * 1024) acc += buf[i].x;
  if (acc == 12345.678f) sink[1] = acc;
}

int main() {
  const long flush_bytes = 768L << 20;
  const int S = 128;
  float4 *flush, *kv;
  float *sink;
  CHECK(hipMalloc(&flush, flush_bytes));
  CHECK(hipMalloc(&kv, (long)S * 800 * 2048));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(flush, 0, flush_bytes));
  CHECK(hipMemset(kv, 0, (long)S * 800 * 2048));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int T : {790, 400}) {
    for (int groups = 0; groups < 2; ++groups) {
      std::vector<float> ts;
      for (int r = 0; r < 7; ++r) {
        flush_k<<<256, 1024>>>(flush, flush_bytes / 16, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (groups) walk_k<true><<<2 * S, 1024>>>(kv, T, sink);
        else walk_k<false><<<2 * S, 1024>>>(kv, T, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms * 1e3f);
      }
      std::sort(ts.begin(), ts.end());
      const double bytes = (double)S * T * 2048;
      printf("T = %3d, layout %-6s: %.1f us = %.0f GB/s (min %.1f us)\n", T, groups ? "groups" : "rows", ts[3], bytes / ts[3] * 1e-3, ts[0]);
    }
  }
  return 0;
}
