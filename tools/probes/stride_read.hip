// Probe: does the access pattern of the cross-attention K/V stream (256 B pieces at
// a 2 KB stride, 8 heads interleaved) cost HBM bandwidth against a head-major
// contiguous layout?  hipcc --offload-arch=gfx950 -O3 stride_read.hip -o stride_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ base, float* out, int T, long wg_stride_f4,
                                          long row_stride_f4, long head_ofs_f4, int layers, long layer_stride_f4, int li) {
  const int head = blockIdx.x, s = blockIdx.y;
  const float4* p = base + (long)s * layers * layer_stride_f4 + (long)li * layer_stride_f4 + head * head_ofs_f4;
  const int g = threadIdx.x >> 4, c = threadIdx.x & 15;   // 16 lanes x 16 B = 256 B per row
  float acc = 0.f;
  for (int t = g; t < T; t += 16) {
    float4 v = p[(long)t * row_stride_f4 + c];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const int S = 128, H = 8, Ld = 14, TCAP = 1600, T = 270;
  const long per_stream = (long)Ld * TCAP * 512;   // floats
  float *buf, *out;
  hipMalloc(&buf, (size_t)S * per_stream * 4);
  hipMalloc(&out, 64);
  hipMemset(buf, 0, (size_t)S * per_stream * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    // mode 0: [S][Ld][TCAP][H][64 floats]  (frame-major, head pieces 256 B at 2 KB stride)
    // mode 1: [S][Ld][H][TCAP][64 floats]  (head-major, contiguous per workgroup)
    const long row_stride = mode == 0 ? 128 : 16;            // in float4
    const long head_ofs = mode == 0 ? 16 : (long)TCAP * 16;  // in float4
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int li = 0; li < Ld; ++li)
        rd<<<dim3(H, S), 256>>>((const float4*)buf, out, T, 0, row_stride, head_ofs, Ld, (long)TCAP * 128, li);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double bytes = (double)S * H * T * 256.0 * Ld;
      printf("mode %d rep %d: %.1f us per layer, %.2f TB/s\n", mode, rep, ms * 1e3 / Ld, bytes / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
