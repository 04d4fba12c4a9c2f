// Probe (round 6): how fast can ALL workgroups of a launch read the SAME weight fragments from their XCD's L2?
// The four-head self kernel's q|k|v projection reads 384 KB per workgroup (16 waves x 24 KB, 1 KB per wave load) of a
// 768 KB matrix that 128 workgroups read at the same time in the same order; its phase stamps say 11-12 us = 33 GB/s per CU.
// Variants: loads in flight per wave (DEPTH), every workgroup in the same order (rot = 0) or from its own start (rot = 1),
// slice per wave 24 KB.  Prints us per launch (HIP events over 20 launches) and GB/s per CU.
// Build: hipcc -O3 --offload-arch=gfx950 l2_shared_weights_probe.hip -o l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ __launch_bounds__(1024) void probe(const f4 *w, float *out, int nblk, int rot, int halves) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave's slice: nblk blocks of 1 KB (64 lanes x 16 B); workgroup (x, y): y selects one half of the matrix (as blockIdx.y = head group)
  const f4 *base = w + ((size_t)(blockIdx.y % halves) * 16 + wave) * nblk * 64 + lane;
  const int start = rot ? (blockIdx.x * 5 + wave * 3) % nblk : 0;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 buf[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = base[(size_t)((start + d) % nblk) * 64];
  for (int i = 0; i < nblk; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += buf[d];
      const int nx = i + DEPTH + d;
      if (nx < nblk) buf[d] = base[(size_t)((start + nx) % nblk) * 64];
    }
  }
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 1024 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int DEPTH>
void run(const f4 *w, float *out, int nblk, int rot, int halves) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<DEPTH>, dim3(128, 2), dim3(1024), 0, 0, w, out, nblk, rot, halves);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<DEPTH>, dim3(128, 2), dim3(1024), 0, 0, w, out, nblk, rot, halves);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 20, kb = 16.0 * nblk;
  printf("depth %2d nblk %2d (%3.0f KB per workgroup) rot %d halves %d: %6.2f us per launch, %6.1f GB/s per CU, %5.2f TB/s chip\n", DEPTH, nblk, kb, rot, halves,
         us, kb * 1024 / us * 1e-3, kb * 1024 * 256 / us * 1e-6);
}
int main() {
  f4 *w; float *out;
  hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
  hipMalloc(&out, 256 * 1024 * 4);
  for (int halves : {2, 1})
    for (int rot : {0, 1}) {
      run<4>(w, out, 24, rot, halves);
      run<8>(w, out, 24, rot, halves);
      run<12>(w, out, 24, rot, halves);
      run<24>(w, out, 24, rot, halves);
    }
  run<12>(w, out, 48, 0, 2);
  run<12>(w, out, 96, 0, 2);
  return 0;
}
