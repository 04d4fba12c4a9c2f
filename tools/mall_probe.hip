// Is a K|V walk faster out of the 256 MB Infinity Cache than out of HBM, and can a side stream warm it while a decoder
// kernel is in its non-HBM phases?  (the walks of the decoder layer kernels run at 5.6 TB/s - the HBM
// limit - for 30 % of a layer's time; the prologues, projections and the feed-forward leave the HBM idle.)
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && timeout 120 /tmp/mall_probe
//   1. read_k over `mb` MB, 256 workgroups x 1024 threads, 16 B per lane:  cold (512 MB of other data read in between)
//      against warm (the same buffer again)                                       -> GB/s each
//   2. spin_k (256 x 1024 threads, 124 KB LDS, no memory traffic, ~40 us) alone, and with prefetch_k (G workgroups x 256
//      threads touching the buffer with loads whose values are dropped) on a second stream -> does the spin kernel slow
//      down, how long does the prefetch take, and is the read behind it warm?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void read_k(const float4 *buf, long n4, float *sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  const long per = n4 / gridDim.x;
  const float4 *s = buf + (long)blockIdx.x * per;
  for (long i = threadIdx.x; i < per; i += 4096) {
    float4 v0 = s[i], v1 = i + 1024 < per ? s[i + 1024] : acc, v2 = i + 2048 < per ? s[i + 2048] : acc,
           v3 = i + 3072 < per ? s[i + 3072] : acc;
    acc.x += v0.x + v1.x + v2.x + v3.x;
    acc.y += v0.y + v1.y + v2.y + v3.y;
  }
  if (acc.x + acc.y == 12345.678f) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void prefetch_k(const float4 *buf, long n4, float *sink) {
  // one 16-byte load per 128-byte line is enough to bring the line in; lanes take consecutive lines
  const long lines = n4 / 8;
  float acc = 0.f;
  for (long l = (long)blockIdx.x * 256 + threadIdx.x; l < lines; l += (long)gridDim.x * 256) acc += buf[l * 8].x;
  if (acc == 12345.678f) sink[1] = acc;
}

__global__ __launch_bounds__(1024) void spin_k(int iters, float *sink) {
  extern __shared__ float lds[];
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  lds[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
    a = a * b + lds[(threadIdx.x + i) & 1023];
    b = b * 0.99999f + 1e-6f;
  }
  if (a == 12345.678f) sink[2] = a;
}

int main() {
  const long flush_bytes = 768L << 20;
  float4 *flush, *buf;
  float *sink;
  CHECK(hipMalloc(&flush, flush_bytes));
  CHECK(hipMalloc(&buf, 256L << 20));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(flush, 0, flush_bytes));
  CHECK(hipMemset(buf, 0, 256L << 20));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(spin_k), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 1024));
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1, p0, p1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&p0)); CHECK(hipEventCreate(&p1));
  auto time_read = [&](long bytes, hipStream_t st) -> float {
    hipEventRecord(e0, st);
    read_k<<<256, 1024, 0, st>>>(buf, bytes / 16, sink);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
  };
  auto do_flush = [&]() {
    read_k<<<256, 1024, 0, s0>>>(flush, flush_bytes / 16, sink);
    hipStreamSynchronize(s0);
  };
  for (long mb : {32L, 92L, 168L, 232L}) {
    const long bytes = mb << 20;
    std::vector<float> cold, warm;
    for (int r = 0; r < 5; ++r) {
      do_flush();
      cold.push_back(time_read(bytes, s0));
      warm.push_back(time_read(bytes, s0));
      warm.push_back(time_read(bytes, s0));
    }
    std::sort(cold.begin(), cold.end()); std::sort(warm.begin(), warm.end());
    printf("read %3ld MB: cold %.1f us = %.0f GB/s   warm %.1f us = %.0f GB/s\n", mb, cold[2], bytes / cold[2] * 1e-3, warm[5],
           bytes / warm[5] * 1e-3);
  }
  // spin alone
  int iters = 3000;
  for (int cal = 0; cal < 3; ++cal) {
    hipEventRecord(e0, s0);
    spin_k<<<256, 1024, 124 * 1024, s0>>>(iters, sink);
    hipEventRecord(e1, s0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("spin alone (%d iterations): %.1f us\n", iters, ms * 1e3f);
    if (cal == 0) iters = (int)(iters * 40.f / (ms * 1e3f));
  }
  for (int G : {32, 64, 128, 256, 512}) {
    for (long mb : {92L, 168L}) {
      const long bytes = mb << 20;
      std::vector<float> ts, tp, tr;
      for (int r = 0; r < 5; ++r) {
        do_flush();
        hipEventRecord(e0, s0);
        hipEventRecord(p0, s1);
        spin_k<<<256, 1024, 124 * 1024, s0>>>(iters, sink);
        prefetch_k<<<G, 256, 0, s1>>>(buf, bytes / 16, sink);
        hipEventRecord(e1, s0);
        hipEventRecord(p1, s1);
        hipEventSynchronize(e1); hipEventSynchronize(p1);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, p0, p1);
        ts.push_back(a * 1e3f); tp.push_back(b * 1e3f);
        tr.push_back(time_read(bytes, s0));
      }
      std::sort(ts.begin(), ts.end()); std::sort(tp.begin(), tp.end()); std::sort(tr.begin(), tr.end());
      printf("prefetch %3d wgs, %3ld MB: spin %.1f us, prefetch %.1f us (%.0f GB/s), read behind it %.1f us = %.0f GB/s\n", G, mb,
             ts[2], tp[2], bytes / tp[2] * 1e-3, tr[2], bytes / tr[2] * 1e-3);
    }
  }
  return 0;
}
