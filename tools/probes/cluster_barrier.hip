// Probe: cost of a barrier + all-to-all hand-off among the 8 workgroups of one "stream cluster" inside ONE launch,
// with placement-independent primitives (system-scope relaxed atomics for payload and counter; no L2 write-back /
// invalidate fences).  Each round: every workgroup publishes a 10 KB partial (W=10 rows x 256 floats), arrives at the
// cluster counter, waits for all 8, then sums the 8 partials.  Prints us per round for 1 / 8 / 32 clusters.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cluster_barrier.hip -o /tmp/cluster_barrier && /tmp/cluster_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void st_sys(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ld_sys(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

typedef float v4f __attribute__((ext_vector_type(4)));
// 16-byte system-scope (sc0 sc1) accesses: write-through stores / cache-bypassing loads, valid for any placement
__device__ __forceinline__ void st4_sys(float *p, v4f v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld4x8_sys(const float *p, long stride, v4f (&v)[8]) {   // 8 loads in flight
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p), "v"(p + stride), "v"(p + 2 * stride), "v"(p + 3 * stride), "v"(p + 4 * stride), "v"(p + 5 * stride),
        "v"(p + 6 * stride), "v"(p + 7 * stride)
      : "memory");
}

// mode 0: counter barrier only; mode 1: + 10 KB partial per workgroup, 16-byte sc0 sc1 stores / loads (8 in flight)
__global__ __launch_bounds__(256) void probe2(float *part, unsigned *bar, float *out, int rounds, int *timeout_flag, int mode) {
  const int c = blockIdx.x, h = blockIdx.y, H = gridDim.y, tid = threadIdx.x;
  const int NEL = 2560;   // floats per partial = 640 float4: threads 0..255 own float4 tid, tid+256, (tid+512 < 640)
  v4f acc[3];
  for (int i = 0; i < 3; ++i) acc[i] = v4f{h + 1.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
    float *mine = part + (((long)(r & 1) * gridDim.x + c) * H + h) * NEL;
    if (mode)
      for (int i = 0; i < 3; ++i)
        if (tid + 256 * i < 640) st4_sys(mine + 4 * (tid + 256 * i), v4f{acc[i].x * 0.125f, 0.f, 0.f, 0.f});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(bar + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned target = (unsigned)(r + 1) * H;
      int spins = 0;
      while (__hip_atomic_load(bar + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *timeout_flag = 1; break; }
      }
    }
    __syncthreads();
    if (mode) {
      const float *base = part + ((long)(r & 1) * gridDim.x + c) * H * NEL;
      for (int i = 0; i < 3; ++i)
        if (tid + 256 * i < 640) {
          v4f v[8];
          ld4x8_sys(base + 4 * (tid + 256 * i), NEL, v);
          float s = 0.f;
          for (int g = 0; g < 8; ++g) s += v[g].x;
          acc[i].x = s;
        }
    }
  }
  if (tid == 0 && h == 0) out[c] = acc[0].x;
}

__global__ __launch_bounds__(256) void probe(float *part /*[2][clusters][8][2560]*/, unsigned *bar /*[clusters]*/, float *out,
                                             int rounds, int *timeout_flag) {
  const int c = blockIdx.x, h = blockIdx.y, H = gridDim.y, tid = threadIdx.x;
  const int NEL = 2560;
  float acc[10];
  for (int i = 0; i < 10; ++i) acc[i] = (float)(h + 1);
  for (int r = 0; r < rounds; ++r) {
    float *mine = part + (((long)(r & 1) * gridDim.x + c) * H + h) * NEL;
    for (int i = 0; i < 10; ++i) st_sys(mine + tid + 256 * i, acc[i] * 0.125f);
    __syncthreads();
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(bar + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned target = (unsigned)(r + 1) * H;
      int spins = 0;
      while (__hip_atomic_load(bar + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *timeout_flag = 1; break; }
      }
    }
    __syncthreads();
    const float *base = part + ((long)(r & 1) * gridDim.x + c) * H * NEL;
    for (int i = 0; i < 10; ++i) {
      float s = 0.f;
      for (int g = 0; g < H; ++g) s += ld_sys(base + (long)g * NEL + tid + 256 * i);
      acc[i] = s;
    }
  }
  if (tid == 0 && h == 0) out[c] = acc[0];
}

int main() {
  const int H = 8, rounds = 588;   // 14 layers x 3 hand-offs x 14 ... one decode iteration has 42
  for (int clusters : {1, 8, 32, 128}) {
    float *part, *out; unsigned *bar; int *tf;
    CHECK(hipMalloc(&part, sizeof(float) * 2 * clusters * H * 2560));
    CHECK(hipMalloc(&out, sizeof(float) * clusters));
    CHECK(hipMalloc(&bar, sizeof(unsigned) * clusters));
    CHECK(hipMalloc(&tf, sizeof(int)));
    CHECK(hipMemset(tf, 0, sizeof(int)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemset(bar, 0, sizeof(unsigned) * clusters));
      CHECK(hipEventRecord(a));
      probe<<<dim3(clusters, H), 256>>>(part, bar, out, rounds, tf);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
      CHECK(hipEventElapsedTime(&ms, a, b));
    }
    int t = 0; float o = 0;
    CHECK(hipMemcpy(&t, tf, sizeof(int), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&o, out, sizeof(float), hipMemcpyDeviceToHost));
    printf("clusters %3d: %.2f us per barrier + 80 KB all-to-all with 4-byte system atomics (timeout %d, check %.3f)\n", clusters, ms * 1e3 / rounds, t, o);
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(bar, 0, sizeof(unsigned) * clusters));
        CHECK(hipEventRecord(a));
        probe2<<<dim3(clusters, H), 256>>>(part, bar, out, rounds, tf, mode);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
      }
      CHECK(hipMemcpy(&t, tf, sizeof(int), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&o, out, sizeof(float), hipMemcpyDeviceToHost));
      printf("             %.2f us per %s (timeout %d, check %.3f)\n", ms * 1e3 / rounds,
             mode ? "barrier + 80 KB all-to-all with 16-byte sc0 sc1 accesses" : "counter barrier alone", t, o);
    }
  }
  return 0;
}
