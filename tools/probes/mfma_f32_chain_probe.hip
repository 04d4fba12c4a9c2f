// Probe: throughput of v_mfma_f32_16x16x4_f32 per SIMD as a function of (waves per SIMD) x (independent accumulators per wave).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_chain_probe.hip -o mfma_probe ; run on the GPU box.
// Question behind it (round 6): the four-head self kernel's projection runs 384 MFMAs per SIMD in 11 us (64 cycles each)
// with 4 waves per SIMD x 2 accumulator chains per wave - is that the pipe, or the chains?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void probe(float *out, long long *cyc, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the layer kernels' pattern: 8 k-steps x NT accumulators, EVERY MFMA with its own A and B registers (a k-block of fragments)
template <int NT>
__global__ __launch_bounds__(1024) void probe_regs(const float *src, float *out, long long *cyc, int iters) {
  f32x4 acc[NT];
  float av[8], bv[NT][8];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    av[j] = src[threadIdx.x * 8 + j];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t][j] = src[8192 + (t * 8 + j) * 1024 + threadIdx.x];
  }
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[t][j], acc[t], 0, 0, 0);
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NT; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NT>
void run_regs(int threads, const float *src, float *out, long long *cyc) {
  const int iters = 64;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe_regs<NT>, dim3(256), dim3(threads), 0, 0, src, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += h[i];
  m /= 256;
  const double per_simd = (double)iters * 8 * NT * (threads / 256);
  printf("own registers per MFMA: threads %4d (waves/SIMD %d) tiles %d: %.0f ticks, %.3f ticks per MFMA per SIMD\n", threads, threads / 256, NT, m, m / per_simd);
}
template <int NACC>
void run(int threads, float *out, long long *cyc) {
  const int iters = 256;
  hipLaunchKernelGGL(probe<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(probe<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += h[i];
  m /= 256;
  const int waves_per_simd = threads / 256;
  const double per_simd = (double)iters * 8 * NACC * waves_per_simd;
  // s_memtime ticks at 100 MHz on gfx950?  report both raw ticks and ticks per MFMA; calibrate with the 1-wave 8-acc case
  printf("threads %4d (waves/SIMD %d) acc/wave %d: %.0f ticks, %.3f ticks per MFMA per SIMD\n", threads, waves_per_simd, NACC, m, m / per_simd);
}
int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  for (int threads : {256, 512, 1024}) {
    run<1>(threads, out, cyc); run<2>(threads, out, cyc); run<4>(threads, out, cyc); run<8>(threads, out, cyc);
  }
  float *src;
  hipMalloc(&src, (8192 + 64 * 1024) * 4); hipMemset(src, 0, (8192 + 64 * 1024) * 4);
  for (int threads : {256, 1024}) {
    run_regs<2>(threads, src, out, cyc); run_regs<6>(threads, src, out, cyc);
  }
  return 0;
}
