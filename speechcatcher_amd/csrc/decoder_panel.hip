// Row-panel kernel of the decoder layer: for a panel of 16 hypothesis rows one
// workgroup runs
//     x  += ctx . W1^T + b1              (attention output projection + residual)
//     xn  = LayerNorm(x)                 (pre-LN of the next sub-block)
//     q   = xn . W2^T + b2               (optional: the cross-attention query)
// (reference: decoder_layer.py:101-123 - `x = residual + self_attn(...)`,
// `x = norm2(x)`, `src_attn(x, memory, memory)` whose first op is linear_q,
// multi_head_attention.py:58-60).
//
// Why a dedicated kernel: at M = streams * beam rows (1280 for 128 streams) the
// d x d projections are launch/latency bound, not MFMA bound - three dependent
// launches (GEMM, split-K reduce + LayerNorm, GEMM) for 0.34 GFLOP.  A 16-row
// panel has the whole feature dimension (d <= 256) in one workgroup, so the
// LayerNorm needs no cross-workgroup step and the second projection starts
// from LDS.
//
// MFMA: v_mfma_f32_16x16x4_f32.  Lane l supplies A[row l%16][k l/16] and
// B[k l/16][col l%16] and receives C[rows 4*(l/16)+j][col l%16].  The k index
// is permuted consistently for both operands (lane kk covers k = 32*ki+8*kk+j,
// j = 0..7), so a lane's operands for 8 MFMAs are 32 contiguous bytes.
//
// Memory: adjacent lanes of the B operand are adjacent COLUMNS, i.e. weight
// rows 1 KB apart - loading fragments straight from a row-major weight costs
// one cache-line lookup per lane (measured: 10 us per 16x256x256 GEMM on one
// CU).  The weights are therefore pre-packed once (sc_pack_panel_weight) in
// fragment order [tile][ki][half][lane][4]: every wave load is 1 KB contiguous.
// Activations enter and leave through LDS so that all global traffic of the
// kernel is full-line coalesced.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PanelArgs {
  const float *A; int lda;
  const float *W1p, *b1;
  float *X; int ldx;
  const float *g, *be; float eps;
  float *XN; int ldn;
  const float *W2p, *b2;
  float *Q; int ldq;
  const int *rows;   // optional row table
  int M;
};

__device__ __forceinline__ f32x4 mfma8(f32x4 acc, const float4 &a0, const float4 &a1, const float4 &b0,
                                       const float4 &b1) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
  return acc;
}

// W [N][K] row-major (torch Linear weight) -> fragment order
//   out[((((tile*KI + ki)*2 + half)*64 + lane)*4 + c] = W[tile*16 + lane%16][ki*32 + 8*(lane/16) + 4*half + c]
__global__ void pack_panel_weight_kernel(const float *W, int N, int K, float *out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * K) return;
  const int KI = K / 32;
  const int c = idx & 3, lane = (idx >> 2) & 63, half = (idx >> 8) & 1;
  const int ki = (idx >> 9) % KI, tile = (idx >> 9) / KI;
  out[idx] = W[(long)(tile * 16 + (lane & 15)) * K + ki * 32 + 8 * (lane >> 4) + 4 * half + c];
}

// D = feature dim, NW waves per workgroup, NT 16-column tiles per wave (NW*NT*16 == D)
template <int D, int NW, int NT>
__global__ __launch_bounds__(NW * 64) void proj_ln_proj_kernel(PanelArgs p) {
  static_assert(NW * NT * 16 == D, "tiling");
  constexpr int KI = D / 32;
  constexpr int LD = D + 4;
  constexpr int RW = 16 / NW;   // rows per wave in the row-wise phases
  constexpr int EL = D / 64;    // elements per lane in the row-wise phases
  constexpr int NTH = NW * 64;
  __shared__ __attribute__((aligned(16))) float PA[16 * LD];  // MFMA A operand (ctx, later LN(x))
  __shared__ __attribute__((aligned(16))) float PC[16 * LD];  // MFMA results in row-major form
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.x * 16;
  // panel row i -> buffer row (clamped for loads; stores are masked by m0 + i < M)
  auto brow = [&](int i) -> long {
    const int m = min(m0 + i, p.M - 1);
    return p.rows ? p.rows[m] : m;
  };

  // ---- all global loads of GEMM 1 are issued before anything waits ----
  float4 a[KI][2], b[KI][NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 *wp = reinterpret_cast<const float4 *>(p.W1p) + (long)(wave * NT + t) * KI * 128 + lane;
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      b[ki][t][0] = wp[ki * 128];
      b[ki][t][1] = wp[ki * 128 + 64];
    }
  }
  // ctx panel -> LDS, 16 B per thread, 128 B contiguous per 8 threads
  constexpr int NV = 16 * D / 4;  // float4 elements of the panel
  float4 stage[(NV + NTH - 1) / NTH];
#pragma unroll
  for (int q = 0; q < (NV + NTH - 1) / NTH; ++q) {
    const int e = threadIdx.x + q * NTH;
    stage[q] = *reinterpret_cast<const float4 *>(p.A + brow(e / (D / 4)) * p.lda + 4 * (e % (D / 4)));
  }
  // residual rows in the row-wise layout (wave -> rows, lane -> columns lane+64e)
  float res[RW][EL];
  long xrow[RW];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    xrow[rr] = brow(wave * RW + rr);
#pragma unroll
    for (int e = 0; e < EL; ++e) res[rr][e] = p.X[xrow[rr] * p.ldx + lane + 64 * e];
  }
  float gam[EL], bet[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) {
    gam[e] = p.g[lane + 64 * e];
    bet[e] = p.be[lane + 64 * e];
  }
  float bias1[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bias1[t] = p.b1 ? p.b1[(wave * NT + t) * 16 + r] : 0.f;
#pragma unroll
  for (int q = 0; q < (NV + NTH - 1) / NTH; ++q) {
    const int e = threadIdx.x + q * NTH;
    if (e < NV) *reinterpret_cast<float4 *>(PA + (e / (D / 4)) * LD + 4 * (e % (D / 4))) = stage[q];
  }
  __syncthreads();
  {
    const float *ap = PA + r * LD + 8 * kk;
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      a[ki][0] = *reinterpret_cast<const float4 *>(ap + ki * 32);
      a[ki][1] = *reinterpret_cast<const float4 *>(ap + ki * 32 + 4);
    }
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ki = 0; ki < KI; ++ki)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma8(acc[t], a[ki][0], a[ki][1], b[ki][t][0], b[ki][t][1]);

  // second projection's weights: in flight during the epilogue and the LayerNorm
  float bias2[NT];
  if (p.W2p) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 *wp = reinterpret_cast<const float4 *>(p.W2p) + (long)(wave * NT + t) * KI * 128 + lane;
      bias2[t] = p.b2 ? p.b2[(wave * NT + t) * 16 + r] : 0.f;
#pragma unroll
      for (int ki = 0; ki < KI; ++ki) {
        b[ki][t][0] = wp[ki * 128];
        b[ki][t][1] = wp[ki * 128 + 64];
      }
    }
  }

  // ---- epilogue 1: acc + bias -> LDS (row-major) ----
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) PC[(4 * kk + j) * LD + (wave * NT + t) * 16 + r] = acc[t][j] + bias1[t];
  __syncthreads();

  // ---- residual + LayerNorm (two-pass statistics, eps inside the sqrt), row-wise ----
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int i = wave * RW + rr;
    const bool live = m0 + i < p.M;
    float x[EL];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      x[e] = res[rr][e] + PC[i * LD + lane + 64 * e];
      if (live) p.X[xrow[rr] * p.ldx + lane + 64 * e] = x[e];
      s += x[e];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const float c = x[e] - mean;
      q += c * c;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + p.eps);
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      const int c = lane + 64 * e;
      const float y = (x[e] - mean) * rstd * gam[e] + bet[e];
      PA[i * LD + c] = y;
      if (p.XN && live) p.XN[xrow[rr] * p.ldn + c] = y;
    }
  }
  if (!p.W2p) return;
  __syncthreads();

  // ---- GEMM 2: A fragments from the LDS panel ----
  {
    const float *ap = PA + r * LD + 8 * kk;
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      a[ki][0] = *reinterpret_cast<const float4 *>(ap + ki * 32);
      a[ki][1] = *reinterpret_cast<const float4 *>(ap + ki * 32 + 4);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ki = 0; ki < KI; ++ki)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma8(acc[t], a[ki][0], a[ki][1], b[ki][t][0], b[ki][t][1]);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) PC[(4 * kk + j) * LD + (wave * NT + t) * 16 + r] = acc[t][j] + bias2[t];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int i = wave * RW + rr;
    if (m0 + i < p.M) {
#pragma unroll
      for (int e = 0; e < EL; ++e) p.Q[xrow[rr] * p.ldq + lane + 64 * e] = PC[i * LD + lane + 64 * e];
    }
  }
}

extern "C" int sc_proj_ln_proj_supported(int D) { return D == 256 || D == 128 || D == 64; }

extern "C" int sc_proj_ln_proj(const float *A, int lda, const float *W1p, const float *b1, float *X, int ldx,
                               const float *ln_g, const float *ln_b, float ln_eps, float *XN, int ldn,
                               const float *W2p, const float *b2, float *Q, int ldq, const int32_t *rows,
                               int M, int D, void *stream) {
  SC_CHECK_ARG(A && W1p && X && ln_g && ln_b, "null operand");
  SC_CHECK_ARG(M > 0, "M must be positive");
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D), "feature dim must be 64, 128 or 256");
  SC_CHECK_ARG(lda % 4 == 0 && lda >= D && ldx >= D, "leading dimensions");
  SC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W1p & 15) == 0 && ((uintptr_t)W2p & 15) == 0, "16-byte alignment");
  SC_CHECK_ARG(!W2p || Q, "second projection needs an output");
  SC_CHECK_ARG(XN || W2p, "nothing to produce after the LayerNorm");
  PanelArgs p{A, lda, W1p, b1, X, ldx, ln_g, ln_b, ln_eps, XN, ldn, W2p, b2, Q, ldq, rows, M};
  hipStream_t st = (hipStream_t)stream;
  const int grid = cdiv(M, 16);
  ProfScope prof = sc_prof_begin(st);
  if (D == 256) proj_ln_proj_kernel<256, 8, 2><<<grid, 512, 0, st>>>(p);
  else if (D == 128) proj_ln_proj_kernel<128, 8, 1><<<grid, 512, 0, st>>>(p);
  else proj_ln_proj_kernel<64, 4, 1><<<grid, 256, 0, st>>>(p);
  // algorithmic: 2*D*D flop per row and projection; A read, X read + written, the
  // weights read once, XN / Q written
  const int np = W2p ? 2 : 1;
  sc_prof_end(prof, SC_PROF_PROJ_LN_PROJ, 2.0 * M * D * D * np,
              4.0 * ((double)M * D * (3 + (XN ? 1 : 0) + (W2p ? 1 : 0)) + (double)np * D * D));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

extern "C" int sc_pack_panel_weight(const float *W, int N, int K, float *out, void *stream) {
  SC_CHECK_ARG(W && out, "null");
  SC_CHECK_ARG(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "N must be a multiple of 16, K of 32");
  const long total = (long)N * K;
  pack_panel_weight_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, N, K, out);
  SC_CHECK_LAUNCH();
  return SC_OK;
}
