// Row-panel kernel of the decoder layer: for a panel of R hypothesis rows one
// workgroup runs
//     x  += ctx . W1^T + b1              (attention output projection + residual)
//     xn  = LayerNorm(x)                 (pre-LN of the next sub-block)
//     q   = xn . W2^T + b2               (optional: the cross-attention query)
// (reference: decoder_layer.py:101-123 - `x = residual + self_attn(...)`,
// `x = norm2(x)`, `src_attn(x, memory, memory)` whose first op is linear_q,
// multi_head_attention.py:58-60).
//
// Why a dedicated kernel: at M = streams * beam rows (<= 1280 for 128 streams) the
// d x d projections are launch/latency bound, not MFMA bound - three dependent
// launches (GEMM, split-K reduce + LayerNorm, GEMM) for 0.34 GFLOP.  A row panel
// has the whole feature dimension (d <= 256) in one workgroup, so the LayerNorm
// needs no cross-workgroup step and the second projection starts from LDS.
//
// History (profiles/): a first version on v_mfma_f32_16x16x4_f32 with
// 16-row panels spent 2 x 3.4 us of MFMA per workgroup; loading its B fragments
// straight from row-major weights cost one cache-line lookup per lane (10 us per
// GEMM) - weights are therefore pre-packed once in the order the lanes read them.
// The current kernel uses the 16-block 4x4x1 MFMA: panels of 4 / 8 / 16 rows cost
// MFMA cycles in proportion to their rows, so small row counts get short
// per-workgroup chains on many CUs (14.6 -> 8.9 us at 80 rows, 15.3 -> 11.4 us at
// 1280 rows for proj + LN + proj).
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

extern "C" int sc_proj_ln_proj_supported(int D) { return D == 256 || D == 128 || D == 64; }

extern "C" int sc_pack_panel_weight(const float *W, int N, int K, float *out, void *stream) {
  SC_CHECK_ARG(W && out, "null");
  SC_CHECK_ARG(N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "N must be a multiple of 16, K of 32");
  const long total = (long)N * K;
  pack_panel_weight_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, N, K, out);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ===========================================================================
// v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4x1): a panel of R = 4*RG rows costs
// R/16 of the MFMA cycles of a 16-row tile.  Lane l of the
// MFMA supplies A[row l%4] (same for all blocks) and B[column 64*tile + l]
// and receives C[rows 0..3][column 64*tile + l]: a lane owns ONE output column.
// Weights are packed [tile][k/4][lane][4] (sc_pack_lane_weight): lane l reads
// W[64*tile + l][4*q .. 4*q+3], one 1 KB wave load per 4 k.  The 8 waves are
// (column tile) x (k part); the k parts meet in LDS.
// ===========================================================================
template <int D, int RG>
__global__ __launch_bounds__(512) void proj_ln_proj_kernel(PanelArgs p) {
  constexpr int R = 4 * RG, NT = D / 64, KS = 8 / NT, KW = D / KS, NL = KW / 4;
  constexpr int LD = D + 4, EL = D / 64;
  __shared__ __attribute__((aligned(16))) float PA[R * LD];
  __shared__ __attribute__((aligned(16))) float PC[KS][R * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = wave % NT, ks = wave / NT, k0 = ks * KW;
  const int m0 = blockIdx.x * R;
  auto brow = [&](int i) -> long {
    const int m = min(m0 + i, p.M - 1);
    return p.rows ? p.rows[m] : m;
  };
  // ---- loads of GEMM 1 ----
  float4 b[NL];
  {
    const float4 *wp = reinterpret_cast<const float4 *>(p.W1p) + ((long)tile * (D / 4) + k0 / 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < NL; ++q) b[q] = wp[q * 64];
  }
  constexpr int NV = R * D / 4;
  constexpr int NQ = (NV + 511) / 512;
  float4 stage[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = min((int)threadIdx.x + q * 512, NV - 1);
    stage[q] = *reinterpret_cast<const float4 *>(p.A + brow(e / (D / 4)) * p.lda + 4 * (e % (D / 4)));
  }
  // row-wise phases: wave handles rows wave, wave + 8, ...
  constexpr int RPW = (R + 7) / 8;
  float res[RPW][EL];
  long xrow[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    xrow[rr] = brow(min(wave + 8 * rr, R - 1));
#pragma unroll
    for (int e = 0; e < EL; ++e) res[rr][e] = p.X[xrow[rr] * p.ldx + lane + 64 * e];
  }
  float gam[EL], bet[EL], bia1[EL], bia2[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) {
    gam[e] = p.g[lane + 64 * e];
    bet[e] = p.be[lane + 64 * e];
    bia1[e] = p.b1 ? p.b1[lane + 64 * e] : 0.f;
    bia2[e] = (p.W2p && p.b2) ? p.b2[lane + 64 * e] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = threadIdx.x + q * 512;
    if (e < NV) *reinterpret_cast<float4 *>(PA + (e / (D / 4)) * LD + 4 * (e % (D / 4))) = stage[q];
  }
  __syncthreads();

  auto gemm = [&](f32x4 (&acc)[RG]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap = PA + (lane & 3) * LD + k0;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      float4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const float4 *>(ap + 4 * rg * LD + 4 * q);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].x, b[q].x, acc[rg], 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].y, b[q].y, acc[rg], 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].z, b[q].z, acc[rg], 0, 0, 0);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].w, b[q].w, acc[rg], 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int r = 0; r < 4; ++r) PC[ks][(4 * rg + r) * LD + 64 * tile + lane] = acc[rg][r];
  };
  f32x4 acc[RG];
  gemm(acc);
  if (p.W2p) {   // second projection's weights: in flight during the LayerNorm
    const float4 *wp = reinterpret_cast<const float4 *>(p.W2p) + ((long)tile * (D / 4) + k0 / 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < NL; ++q) b[q] = wp[q * 64];
  }
  __syncthreads();
  // ---- k parts + bias + residual, LayerNorm ----
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = wave + 8 * rr;
    if (i < R) {   // wave-uniform
      const bool live = m0 + i < p.M;
      float x[EL];
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        float y = PC[0][i * LD + lane + 64 * e];
#pragma unroll
        for (int z = 1; z < KS; ++z) y += PC[z][i * LD + lane + 64 * e];
        x[e] = res[rr][e] + (y + bia1[e]);
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
  }
  if (!p.W2p) return;
  __syncthreads();
  gemm(acc);
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = wave + 8 * rr;
    if (i < R && m0 + i < p.M) {
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        float y = PC[0][i * LD + lane + 64 * e];
#pragma unroll
        for (int z = 1; z < KS; ++z) y += PC[z][i * LD + lane + 64 * e];
        p.Q[xrow[rr] * p.ldq + lane + 64 * e] = y + bia2[e];
      }
    }
  }
}

// W [N][K] -> [tile = n/64][k/4][lane = n%64][4]
__global__ void pack_lane_weight_kernel(const float *W, int N, int K, float *out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * K) return;
  const int c = idx & 3, lane = (idx >> 2) & 63;
  const long q = (idx >> 8) % (K / 4), tile = (idx >> 8) / (K / 4);
  out[idx] = W[(tile * 64 + lane) * K + 4 * q + c];
}

extern "C" int sc_pack_lane_weight(const float *W, int N, int K, float *out, void *stream) {
  SC_CHECK_ARG(W && out, "null");
  SC_CHECK_ARG(N > 0 && K > 0 && N % 64 == 0 && K % 4 == 0, "N must be a multiple of 64, K of 4");
  const long total = (long)N * K;
  pack_lane_weight_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, N, K, out);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

template <int D>
static void launch_panel4(const PanelArgs &p, int rg, hipStream_t st) {
  if (rg == 1) proj_ln_proj_kernel<D, 1><<<cdiv(p.M, 4), 512, 0, st>>>(p);
  else if (rg == 2) proj_ln_proj_kernel<D, 2><<<cdiv(p.M, 8), 512, 0, st>>>(p);
  else proj_ln_proj_kernel<D, 4><<<cdiv(p.M, 16), 512, 0, st>>>(p);
}

extern "C" int sc_proj_ln_proj(const float *A, int lda, const float *W1q, const float *b1, float *X, int ldx,
                               const float *ln_g, const float *ln_b, float ln_eps, float *XN, int ldn,
                               const float *W2q, const float *b2, float *Q, int ldq, const int32_t *rows,
                               int M, int D, void *stream) {
  SC_CHECK_ARG(A && W1q && X && ln_g && ln_b, "null operand");
  SC_CHECK_ARG(M > 0, "M must be positive");
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D), "feature dim must be 64, 128 or 256");
  SC_CHECK_ARG(lda % 4 == 0 && lda >= D && ldx >= D, "leading dimensions");
  SC_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W1q & 15) == 0 && ((uintptr_t)W2q & 15) == 0, "16-byte alignment");
  SC_CHECK_ARG(!W2q || Q, "second projection needs an output");
  SC_CHECK_ARG(XN || W2q, "nothing to produce after the LayerNorm");
  // rows per panel: the smallest that keeps the launch within one round of workgroups and
  // the weight re-streaming (every workgroup reads all of W) modest (tools/panel4_probe.py)
  int rpp = M <= 1024 ? 4 : (M <= 2048 ? 8 : 16);   // <= 256 workgroups (measured sweep: one round is what matters)
  if (const char *e = sc_hook("SC_PANEL_ROWS")) {
    const int v = atoi(e);
    if (v == 4 || v == 8 || v == 16) rpp = v;
  }
  PanelArgs p{A, lda, W1q, b1, X, ldx, ln_g, ln_b, ln_eps, XN, ldn, W2q, b2, Q, ldq, rows, M};
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  if (D == 256) launch_panel4<256>(p, rpp / 4, st);
  else if (D == 128) launch_panel4<128>(p, rpp / 4, st);
  else launch_panel4<64>(p, rpp / 4, st);
  // algorithmic: 2*D*D flop per row and projection; A read, X read + written, the
  // weights read once, XN / Q written
  const int np = W2q ? 2 : 1;
  sc_prof_end(prof, SC_PROF_PROJ_LN_PROJ, 2.0 * M * D * D * np,
              4.0 * ((double)M * D * (3 + (XN ? 1 : 0) + (W2q ? 1 : 0)) + (double)np * D * D));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ===========================================================================
// Reduce + LayerNorm + projection: the tail of the fused feed-forward and the
// head of what follows it (the next layer's Q|K|V projection, or the output
// layer) in one launch:
//     x_out[r] = x_in[r] + sum_z part[z][m] + b2            (split sums of sc_ffn_ln)
//     q[r]     = LayerNorm(x_out[r]) . Wq^T + bq             (N = NB * D output columns)
// Grid (row panels, NB column blocks of D).  Every column block reduces the
// panel's rows itself (partials are L2-resident: just written), block 0 also
// stores x_out (and LN(x_out) if asked).  x_in and x_out MUST be different
// buffers: sibling workgroups read x_in at unrelated times.
// Replaces gemm_splitk_reduce_ln + a 64x64-tile GEMM: one launch instead of
// two, and the 4x4x1 MFMA row panels (see above) for the projection.
// ===========================================================================
struct RedProjArgs {
  const float *part; int npart; int part_M;
  const float *b2;
  const float *Xin; float *Xout;
  const int *rows; int M;
  const float *g, *be; float eps;
  float *XN;
  const float *Wq, *bq;
  float *Q; int ldq;
  int by_row;   // partial sums indexed by row id (xrow) instead of by position m (sc_dec_layer_ffn)
  int part_half;   // the partial sums hold fp16 elements (sc_search.act_half)
  int cb;          // column blocks per workgroup (grid.y = N / D / cb): the reduce + LayerNorm of a panel is done once for them
};

typedef _Float16 rp_h4 __attribute__((ext_vector_type(4)));
// WH: Wq holds the fp16 copy of the lane-packed weight (sc_search.out_w_qh): one v_mfma_f32_4x4x4_16B_f16 per 4 k values
// instead of four v_mfma_f32_4x4x1_16B_f32 (fp32 accumulation; the LayerNorm rows are rounded to fp16 on their way
// from LDS) - the output layer of the fp16 decoder mode
template <int D, int RG, bool WH = false>
__global__ __launch_bounds__(512) void reduce_ln_proj_kernel(RedProjArgs p) {
  typedef typename std::conditional<WH, rp_h4, float4>::type BF;
  constexpr int R = 4 * RG, NT = D / 64, KS = 8 / NT, KW = D / KS, NL = KW / 4;
  constexpr int LD = D + 4, EL = D / 64, RPW = (R + 7) / 8;
  __shared__ __attribute__((aligned(16))) float PA[R * LD];
  __shared__ __attribute__((aligned(16))) float PC[KS][R * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = wave % NT, ks = wave / NT, k0 = ks * KW;
  const int m0 = blockIdx.x * R, nb0 = blockIdx.y * p.cb;
  // projection weights of the first column block: in flight during the reduce + LayerNorm
  BF b[NL];
  auto load_w = [&](int nb) {
    const BF *wp = reinterpret_cast<const BF *>(p.Wq) + ((long)(nb * NT + tile) * (D / 4) + k0 / 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < NL; ++q) b[q] = wp[q * 64];
  };
  load_w(nb0);
  float gam[EL], bet[EL], bia2[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) {
    gam[e] = p.g[lane + 64 * e];
    bet[e] = p.be[lane + 64 * e];
    bia2[e] = p.b2 ? p.b2[lane + 64 * e] : 0.f;
  }
  long xrow[RPW];
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = wave + 8 * rr;
    const int m = min(m0 + min(i, R - 1), p.M - 1);
    xrow[rr] = p.rows ? p.rows[m] : m;
    if (i < R) {   // wave-uniform
      const bool live = m0 + i < p.M;
      float x[EL], y[EL];
      float s = 0.f;
      // partial sums: all EL x 16 loads of a row are issued together (one memory round trip per
      // 16 slices), then summed in the canonical order (the same as gemm_splitk_reduce_ln_kernel)
      for (int z0 = 0; z0 < p.npart; z0 += 16) {
        float pv[EL][16];
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const long pe = ((long)min(z0 + q, p.npart - 1) * p.part_M + (p.by_row ? xrow[rr] : (long)m)) * D + lane + 64 * e;
            pv[e][q] = p.part_half ? (float)reinterpret_cast<const _Float16 *>(p.part)[pe] : p.part[pe];
          }
        // canonical order (common.h): balanced tree over the aligned index pairs, 8 at a time, the batches in order
#pragma unroll
        for (int e = 0; e < EL; ++e) {
          float lo[8], hi[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            lo[q] = z0 + q < p.npart ? pv[e][q] : 0.f;
            hi[q] = z0 + 8 + q < p.npart ? pv[e][8 + q] : 0.f;
          }
          float t = sc_tree8f(lo);
          if (z0 + 8 < p.npart) t = t + sc_tree8f(hi);
          y[e] = z0 == 0 ? t : y[e] + t;
        }
      }
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        const int c = lane + 64 * e;
        x[e] = p.Xin[xrow[rr] * D + c] + (y[e] + bia2[e]);
        if (live && blockIdx.y == 0) p.Xout[xrow[rr] * D + c] = x[e];
        s += x[e];
      }
      const float mean = wave_sum(s) / (float)D;
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        const float c = x[e] - mean;
        q2 += c * c;
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + p.eps);
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        const int c = lane + 64 * e;
        const float y = (x[e] - mean) * rstd * gam[e] + bet[e];
        PA[i * LD + c] = y;
        if (p.XN && live && blockIdx.y == 0) p.XN[xrow[rr] * D + c] = y;
      }
    }
  }
  __syncthreads();
  for (int cbi = 0; cbi < p.cb; ++cbi) {
  const int nb = nb0 + cbi;
  if (cbi > 0) load_w(nb);   // (the fragments of a column block fill the register budget: no double buffering)
  float biaq[EL];
#pragma unroll
  for (int e = 0; e < EL; ++e) biaq[e] = p.bq ? p.bq[nb * D + lane + 64 * e] : 0.f;
  f32x4 acc[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) acc[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float *ap = PA + (lane & 3) * LD + k0;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      float4 a[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) a[rg] = *reinterpret_cast<const float4 *>(ap + 4 * rg * LD + 4 * q);
      if constexpr (WH) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
          acc[rg] = __builtin_amdgcn_mfma_f32_4x4x4f16(rp_h4{(_Float16)a[rg].x, (_Float16)a[rg].y, (_Float16)a[rg].z, (_Float16)a[rg].w},
                                                       b[q], acc[rg], 0, 0, 0);
      } else {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].x, b[q].x, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].y, b[q].y, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].z, b[q].z, acc[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[rg].w, b[q].w, acc[rg], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int r = 0; r < 4; ++r) PC[ks][(4 * rg + r) * LD + 64 * tile + lane] = acc[rg][r];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int i = wave + 8 * rr;
    if (i < R && m0 + i < p.M) {
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        float y = PC[0][i * LD + lane + 64 * e];
#pragma unroll
        for (int z = 1; z < KS; ++z) y += PC[z][i * LD + lane + 64 * e];
        p.Q[xrow[rr] * p.ldq + nb * D + lane + 64 * e] = y + biaq[e];
      }
    }
  }
  if (cbi + 1 < p.cb) __syncthreads();   // PC is rewritten by the next column block
  }
}

template <int D, bool WH = false>
static void launch_redproj(const RedProjArgs &p, int rg, int nblocks, hipStream_t st) {
  if (rg == 1) reduce_ln_proj_kernel<D, 1, WH><<<dim3(cdiv(p.M, 4), nblocks / p.cb), 512, 0, st>>>(p);
  else if (rg == 2) reduce_ln_proj_kernel<D, 2, WH><<<dim3(cdiv(p.M, 8), nblocks / p.cb), 512, 0, st>>>(p);
  else reduce_ln_proj_kernel<D, 4, WH><<<dim3(cdiv(p.M, 16), nblocks / p.cb), 512, 0, st>>>(p);
}

// internal (common.h): called by sc_ffn_ln_proj after the fused FFN kernel wrote its partial sums
int sc_launch_reduce_ln_proj(const float *part, int npart, int part_M, const float *b2, const float *Xin, float *Xout,
                             const int32_t *rows, int M, int D, const float *ln_g, const float *ln_b, float ln_eps,
                             float *XN, const float *Wq, const float *bq, float *Q, int N, hipStream_t st,
                             int by_row, int half_mode) {
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D) && N % D == 0 && N > 0, "projection width must be a multiple of D");
  SC_CHECK_ARG(Xin != Xout, "x_in and x_out must be different buffers");
  const int nblocks = N / D;
  // rows per panel: keep panels * column blocks within one round of workgroups
  int rpp = (long)cdiv(M, 4) * nblocks <= 256 ? 4 : ((long)cdiv(M, 8) * nblocks <= 256 ? 8 : 16);
  if (const char *e = sc_hook("SC_PANEL_ROWS")) {
    const int v = atoi(e);
    if (v == 4 || v == 8 || v == 16) rpp = v;
  }
  // half_mode (fp16 decoder mode, sc_search.act_half): bit 0 = Wq holds fp16 elements, bit 1 = the partial sums do
  // more than one round of workgroups even with 16-row panels (a full bucket's output layer: 80 panels x 4 column
  // blocks): two column blocks per workgroup - the panel's partial sums are reduced half as often and the launch is
  // one round (27.6 -> ~16 us at 1280 rows x 1024 columns)
  const int cb = (rpp == 16 && (long)cdiv(M, 16) * nblocks > 256 && nblocks % 2 == 0) ? 2 : 1;
  RedProjArgs p{part, npart, part_M, b2, Xin, Xout, rows, M, ln_g, ln_b, ln_eps, XN, Wq, bq, Q, N, by_row, (half_mode & 2) ? 1 : 0, cb};
  ProfScope prof = sc_prof_begin(st);
  if (half_mode & 1) {
    SC_CHECK_ARG(D == 256 || D == 128, "fp16 output layer: d must be 128 or 256");
    if (D == 256) launch_redproj<256, true>(p, rpp / 4, nblocks, st);
    else launch_redproj<128, true>(p, rpp / 4, nblocks, st);
  } else if (D == 256) launch_redproj<256>(p, rpp / 4, nblocks, st);
  else if (D == 128) launch_redproj<128>(p, rpp / 4, nblocks, st);
  else launch_redproj<64>(p, rpp / 4, nblocks, st);
  sc_prof_end(prof, SC_PROF_PROJ_LN_PROJ, 2.0 * M * D * N,
              4.0 * ((double)M * D * (2 + npart) + (double)N * D + (double)M * N));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Tail of the decoder, first half (round 5): x = x_in + b2 + tree sum of the last layer's feed-forward partial sums (by row
// id, canonical order: common.h), after_norm -> XN.  One wave per row.  The output layer is then ONE tiled GEMM over all
// rows (sc_gemm): the row-panel kernel that did both (reduce_ln_proj_kernel) re-reduced every panel's partial sums in each
// of its column-block workgroups and re-streamed the 1 MB output matrix per 16-row panel - 44 MB of traffic for 7.6 MB of
// operands, 36 us at a full bucket for 4.3 us of matrix work.
__global__ __launch_bounds__(256) void reduce_ln_rows_kernel(const float *__restrict__ part, int npart, long part_M, const float *__restrict__ b2,
                                                             const float *__restrict__ Xin, float *__restrict__ Xout, const int *__restrict__ rows,
                                                             int M, int D, const float *__restrict__ g, const float *__restrict__ be, float eps,
                                                             float *__restrict__ XN, int part_half) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long row = rows ? rows[m] : m;
  const int c4 = lane;                 // D / 4 <= 64 float4 pieces per row
  const bool act = c4 < D / 4;
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act) {
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z0 = 0; z0 < npart; z0 += 8) {
      float4 p[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long pe = ((long)min(z0 + q, npart - 1) * part_M + row) * D + 4 * c4;
        if (part_half) {
          const rp_h4 h = *reinterpret_cast<const rp_h4 *>(reinterpret_cast<const _Float16 *>(part) + pe);
          p[q] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        } else {
          p[q] = *reinterpret_cast<const float4 *>(part + pe);
        }
        if (z0 + q >= npart) p[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const float4 t = sc_tree8(p);
      y = z0 == 0 ? t : sc_add4(y, t);
    }
    const float4 xi = *reinterpret_cast<const float4 *>(Xin + row * D + 4 * c4);
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b2) bb = *reinterpret_cast<const float4 *>(b2 + 4 * c4);
    x = make_float4(xi.x + (y.x + bb.x), xi.y + (y.y + bb.y), xi.z + (y.z + bb.z), xi.w + (y.w + bb.w));
    if (Xout) *reinterpret_cast<float4 *>(Xout + row * D + 4 * c4) = x;
  }
  const float s = act ? (x.x + x.y) + (x.z + x.w) : 0.f;
  const float mean = wave_sum(s) / (float)D;
  const float a = x.x - mean, b = x.y - mean, c = x.z - mean, e = x.w - mean;
  const float q2 = act ? (a * a + b * b) + (c * c + e * e) : 0.f;
  const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + eps);
  if (act) {
    const float4 gm = *reinterpret_cast<const float4 *>(g + 4 * c4), bt = *reinterpret_cast<const float4 *>(be + 4 * c4);
    *reinterpret_cast<float4 *>(XN + row * D + 4 * c4) =
        make_float4(a * rstd * gm.x + bt.x, b * rstd * gm.y + bt.y, c * rstd * gm.z + bt.z, e * rstd * gm.w + bt.w);
  }
}

// Tail of the head-parallel decoder (decoder_layer.hip): sum of the last layer's feed-forward partial sums
// (by row id) + b2 + residual, after_norm, output layer -> sb->logits (transformer_decoder.py:243-249).
extern "C" int sc_dec_output_logits(const sc_search *sbp, const float *xin, float *xout, const float *ffn_part,
                                    int n_ffn_part, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xin && xout && xin != xout && ffn_part && n_ffn_part > 0, "null / aliased");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(sb.out_w_q && sb.V % sb.d == 0, "needs the lane-packed output layer (V a multiple of d)");
  const int M = sb.rowmap ? sb.n_rows : sb.S * sb.W;
  const bool hm = (sb.act_half & 4) != 0 && sb.out_w_qh != nullptr;
  SC_CHECK_ARG(!(sb.act_half & 4) || hm, "fp16 output layer needs out_w_qh");
  if (!hm && sb.d % 4 == 0 && sb.d <= 256 && sb.dq && sb.out_w && !sc_hook("SC_LOGITS_PANEL")) {
    // (round 5) reduce + after_norm once per row, then the output layer as one tiled GEMM over the bucket's rows - the
    // same two launches for every bucket size (the form must not depend on the row count: common.h)
    hipStream_t st = (hipStream_t)stream;
    ProfScope prof = sc_prof_begin(st);
    reduce_ln_rows_kernel<<<cdiv(M, 4), 256, 0, st>>>(ffn_part, n_ffn_part, (long)sb.S * sb.W, sb.layers[sb.n_layers - 1].b2, xin, xout,
                                                      sb.rowmap, M, sb.d, sb.dec_norm_g, sb.dec_norm_b, sb.ln_eps, sb.dq, (sb.act_half & 2) ? 1 : 0);
    SC_CHECK_LAUNCH();
    sc_prof_end(prof, SC_PROF_PROJ_LN_PROJ, 0.0, 4.0 * (double)M * sb.d * (2 + n_ffn_part));
    return sc_gemm(sb.dq, sb.rowmap, sb.d, sb.out_w, sb.out_b, sb.logits, sb.rowmap, sb.V, M, sb.V, sb.d, 0, 0, stream);
  }
  return sc_launch_reduce_ln_proj(ffn_part, n_ffn_part, sb.S * sb.W, sb.layers[sb.n_layers - 1].b2, xin, xout,
                                  sb.rowmap, M, sb.d, sb.dec_norm_g, sb.dec_norm_b, sb.ln_eps, nullptr,
                                  hm ? (const float *)sb.out_w_qh : sb.out_w_q, sb.out_b, sb.logits, sb.V,
                                  (hipStream_t)stream, 1, (hm ? 1 : 0) | (sb.act_half & 2));
}
