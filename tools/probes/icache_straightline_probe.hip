// Probe (round 6): what does straight-line (fully unrolled, executed once per launch) MFMA code cost when the instruction
// cache is cold?  The layer kernels' projections are ~100 MFMAs of straight-line code per wave and run at ~52 cycles per
// MFMA and SIMD where a loop of the same instructions runs at 32 (mfma_f32_chain_probe.hip).
// probe<N, ID>: N x 8 unrolled v_mfma_f32_16x16x4_f32 (8 bytes each -> N x 64 bytes of code), 16 waves per workgroup, one
// workgroup per CU.  Warm: the same kernel launched repeatedly.  Cold: four different 32-KB kernels launched between two
// launches of the measured one (> 64 KB of other code through the instruction cache).
// Build: hipcc -O3 --offload-arch=gfx950 icache_straightline_probe.hip -o icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int N, int ID>
__global__ __launch_bounds__(1024) void probe(float *out, long long *cyc) {
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, (float)ID};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+s"(const_cast<long long &>(t0)));   // (the MFMAs depend on values defined behind the first stamp)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < N; ++r)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 3], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;                 // (needs the results: the MFMA pipe is asynchronous)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static double mean_ticks(long long *cyc) {
  long long h[256];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < 256; ++i) m += h[i];
  return m / 256;
}
template <int N>
void run(float *out, long long *cyc, long long *junk) {
  // warm: third launch in a row
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<N, 0>), dim3(256), dim3(1024), 0, 0, out, cyc);
  hipDeviceSynchronize();
  const double warm = mean_ticks(cyc);
  // cold: other code through the instruction cache first
  hipLaunchKernelGGL((probe<512, 1>), dim3(256), dim3(1024), 0, 0, out, junk);
  hipLaunchKernelGGL((probe<512, 2>), dim3(256), dim3(1024), 0, 0, out, junk);
  hipLaunchKernelGGL((probe<512, 3>), dim3(256), dim3(1024), 0, 0, out, junk);
  hipLaunchKernelGGL((probe<512, 4>), dim3(256), dim3(1024), 0, 0, out, junk);
  hipLaunchKernelGGL((probe<N, 0>), dim3(256), dim3(1024), 0, 0, out, cyc);
  hipDeviceSynchronize();
  const double cold = mean_ticks(cyc);
  const double per_simd = N * 8.0 * 4;
  printf("%4d MFMAs per wave (%5.1f KB of code): warm %.1f ticks per MFMA per SIMD, cold %.1f\n", N * 8, N * 64 / 1024.0, warm / per_simd, cold / per_simd);
}
int main() {
  float *out; long long *cyc, *junk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&junk, 256 * 8);
  run<4>(out, cyc, junk); run<12>(out, cyc, junk); run<16>(out, cyc, junk); run<32>(out, cyc, junk); run<64>(out, cyc, junk); run<128>(out, cyc, junk);
  return 0;
}
