// fp32 GEMM  C = A . W^T (+bias, ReLU, residual) on the f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157 TFLOP/s peak on MI355X).
//
// Every nn.Linear of the hot path and the second Conv2d of the subsampling
// (as an implicit GEMM over a channels-last conv1 output) run through this
// kernel.  Both operands are K-contiguous ([M,K] activations, [N,K] weights),
// tiles are staged through LDS transposed ([k][m], leading dim BM+1) so that
//   * global loads are 16 B/lane, 128 B contiguous per 8 lanes,
//   * LDS stores are bank-conflict free ((4*kq+i)*(BM+1)+r covers 32 banks),
//   * MFMA operand reads are one conflict-free ds_read_b32 per operand
//     (lane l reads [k + l/32][m + l%32]).
// Row gather (a_rows) / scatter (c_rows) tables let ragged per-stream buffers
// be consumed and produced without staging copies.
#define SC_STAMP_ON (p.dbg_stamp)
#include "common.h"
#include <mutex>
#include <unordered_map>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// fp16 hi | lo split of four fp32 values: hi = fp16(x), lo = fp16((x - hi) * 2^11) - see ffn_fused_kernel (WF = 2)
__device__ __forceinline__ void ffn_split4(const float4 &x, h16x4 &hi, h16x4 &lo) {
  hi = h16x4{(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
  lo = h16x4{(_Float16)((x.x - (float)hi[0]) * 2048.f), (_Float16)((x.y - (float)hi[1]) * 2048.f),
             (_Float16)((x.z - (float)hi[2]) * 2048.f), (_Float16)((x.w - (float)hi[3]) * 2048.f)};
}

struct GemmArgs {
  const float *A;
  const int *a_rows;
  int lda;
  const float *W;
  const float *bias;
  float *C;
  const int *c_rows;
  int ldc;
  int M, N, K, flags, conv_f1;
  float *part;  // split-K partial sums [ksplit][M][N] (workspace), or null
  int kslice;   // K extent handled by one block along grid.z (multiple of 32)
  // column blocks (sc_gemm_colblocks): output column n lands in block n / ncb at C + block * cb_stride, column n % ncb of
  // the row (ncb = 0: plain).  One launch for the cross-attention K|V rows of ALL decoder layers (streams.hip: project_rows)
  int ncb;
  long cb_stride;
};

__device__ __forceinline__ long gemm_kofs(const GemmArgs &g, int k0) {
  if (g.conv_f1 > 0) {
    int tap = k0 / g.lda;
    return (long)((tap / 3) * g.conv_f1 + tap % 3) * g.lda + (k0 % g.lda);
  }
  return k0;
}

// SPLIT (flag SC_GEMM_SPLIT16): the same product sums on the fp16 matrix pipe - both tiles are split into fp16 hi | lo
// when they are staged into LDS (k-contiguous rows instead of the transposed fp32 tiles), three
// v_mfma_f32_32x32x16_f16 per 16 k values instead of eight v_mfma_f32_32x32x2_f32 (see ffn_fused_kernel, WF = 2).
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK = 32, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs g) {
  constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
  constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
  constexpr int KQ = BK / 4;            // float4 per K row of a tile
  constexpr int RP = 256 / KQ;          // tile rows loaded per pass of the 256 threads
  constexpr int AI = BM / RP, BI = BN / RP;
  constexpr int LDH = BK + 8;           // SPLIT: row stride (fp16 elements) of the k-contiguous hi / lo tiles
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(!SPLIT || BK == 32, "the split form stages 32-wide K tiles");
  constexpr size_t SM_BYTES = SPLIT ? (size_t)(BM + BN) * LDH * 2 * sizeof(_Float16) : (size_t)BK * (LDA_S + LDB_S) * sizeof(float);
  __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[SM_BYTES];
  float *As = reinterpret_cast<float *>(gemm_smem), *Bs = As + BK * LDA_S;
  _Float16 *AsH = reinterpret_cast<_Float16 *>(gemm_smem), *AsL = AsH + BM * LDH, *BsH = AsL + BM * LDH, *BsL = BsH + BN * LDH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware tile mapping: workgroup b runs on XCD b % 8 (each XCD has its own
  // L2).  Give every XCD a CONTIGUOUS range of tiles (n fastest), so the tiles
  // that share an A row panel sit behind one L2 instead of being fetched by all
  // eight.  Bijective for any grid size; affects speed / traffic only.
  int tile_n, tile_m;
  {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int b = blockIdx.x + gx * blockIdx.y;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    tile_n = tile % gx;
    tile_m = tile / gx;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kq = tid % KQ, lr = tid / KQ;
  const int kbeg = blockIdx.z * g.kslice;
  const int kend = (kbeg + g.kslice < g.K) ? kbeg + g.kslice : g.K;

  // Loads are UNCONDITIONAL (rows clamped into range): out-of-range rows and
  // columns compute garbage that the epilogue never stores.  Only the "-1 =
  // zero row" gather entries need masking, done with a select on the VALUE so
  // that hipcc does not branch around each load (which costs a vmcnt(0) wait
  // per load and serialises the whole K loop).
  long abase[AI];
  float amask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + lr + RP * i;
    m = m < g.M ? m : g.M - 1;
    int row = g.a_rows ? g.a_rows[m] : m;
    amask[i] = row < 0 ? 0.f : 1.f;
    abase[i] = (long)(row < 0 ? 0 : row) * g.lda;
  }
  long wbase[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + lr + RP * i;
    wbase[i] = (long)(n < g.N ? n : g.N - 1) * g.K;
  }

  f32x16 acc[TM][TN];
  f32x16 accc[SPLIT ? TM : 1][SPLIT ? TN : 1];   // SPLIT: the 2^11-scaled cross terms
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if (SPLIT) accc[i][j][r] = 0.f;
      }

  float4 ra[AI], rb[BI];

  {
    long ko = gemm_kofs(g, kbeg);
#pragma unroll
    for (int i = 0; i < AI; ++i)
      ra[i] = *reinterpret_cast<const float4 *>(g.A + abase[i] + ko + kq * 4);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      rb[i] = *reinterpret_cast<const float4 *>(g.W + wbase[i] + kbeg + kq * 4);
  }

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        h16x4 hi, lo;
        ffn_split4(make_float4(ra[i].x * amask[i], ra[i].y * amask[i], ra[i].z * amask[i], ra[i].w * amask[i]), hi, lo);
        *reinterpret_cast<h16x4 *>(AsH + (lr + RP * i) * LDH + kq * 4) = hi;
        *reinterpret_cast<h16x4 *>(AsL + (lr + RP * i) * LDH + kq * 4) = lo;
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        h16x4 hi, lo;
        ffn_split4(rb[i], hi, lo);
        *reinterpret_cast<h16x4 *>(BsH + (lr + RP * i) * LDH + kq * 4) = hi;
        *reinterpret_cast<h16x4 *>(BsL + (lr + RP * i) * LDH + kq * 4) = lo;
      }
    } else {
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        float *p = As + (kq * 4) * LDA_S + lr + RP * i;
        p[0] = ra[i].x * amask[i];
        p[LDA_S] = ra[i].y * amask[i];
        p[2 * LDA_S] = ra[i].z * amask[i];
        p[3 * LDA_S] = ra[i].w * amask[i];
      }
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        float *p = Bs + (kq * 4) * LDB_S + lr + RP * i;
        p[0] = rb[i].x;
        p[LDB_S] = rb[i].y;
        p[2 * LDB_S] = rb[i].z;
        p[3 * LDB_S] = rb[i].w;
      }
    }
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next K tile into registers
      long ko = gemm_kofs(g, k0 + BK);
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = *reinterpret_cast<const float4 *>(g.A + abase[i] + ko + kq * 4);
#pragma unroll
      for (int i = 0; i < BI; ++i)
        rb[i] = *reinterpret_cast<const float4 *>(g.W + wbase[i] + (k0 + BK) + kq * 4);
    }
    if constexpr (SPLIT) {   // lane l: row l % 32 of a 32-row tile, the 8 k values 8 * (l / 32) .. of a 16-wide k step
      const int ro = (lane & 31) * LDH + (lane >> 5) * 8;
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        h16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = *reinterpret_cast<const h16x8 *>(AsH + (wm * (TM * 32) + i * 32) * LDH + ro + ks);
          al[i] = *reinterpret_cast<const h16x8 *>(AsL + (wm * (TM * 32) + i * 32) * LDH + ro + ks);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const h16x8 *>(BsH + (wn * (TN * 32) + j * 32) * LDH + ro + ks);
          bl[j] = *reinterpret_cast<const h16x8 *>(BsL + (wn * (TN * 32) + j * 32) * LDH + ro + ks);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accc[i][j], 0, 0, 0);
            accc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accc[i][j], 0, 0, 0);
          }
      }
    } else {
      const float *ap = As + (lane >> 5) * LDA_S + wm * (TM * 32) + (lane & 31);
      const float *bp = Bs + (lane >> 5) * LDB_S + wn * (TN * 32) + (lane & 31);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = ap[kk * LDA_S + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = bp[kk * LDB_S + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if constexpr (SPLIT) {
      if (k0 + BK >= kend) {   // last K tile: fold the cross terms in
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accc[i][j][r] * (1.f / 2048.f);
      }
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Row-table and residual loads are
  // issued unconditionally on clamped addresses (16 in flight, one wait);
  // only the stores are predicated.
  if (g.part) {  // split-K: raw partial sums, reduced by gemm_splitk_reduce_kernel
    float *pz = g.part + (long)blockIdx.z * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < g.M && n < g.N) pz[(long)m * g.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  const bool relu = g.flags & SC_GEMM_RELU, resid = g.flags & SC_GEMM_RESIDUAL;
  const bool has_rows = g.c_rows != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    long roff[16];
    bool rok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const bool mv = m < g.M;
      const int mm = mv ? m : g.M - 1;
      const int crow = has_rows ? g.c_rows[mm] : mm;
      rok[r] = mv && crow >= 0;
      roff[r] = (long)(crow < 0 ? 0 : crow) * g.ldc;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (TN * 32) + j * 32 + (lane & 31);
      const bool nv = n < g.N;
      const int nn = nv ? n : g.N - 1;
      const float bv = g.bias ? g.bias[nn] : 0.f;
      const long coff = g.ncb > 0 ? (long)(nn / g.ncb) * g.cb_stride + nn % g.ncb : (long)nn;
      float old[16];
      if (resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = g.C[roff[r] + coff];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (resid) v = old[r] + v;
        if (rok[r] && nv) g.C[roff[r] + coff] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Skinny GEMM (M <= 64: one stream's decoder rows / one encoder block).  These
// are weight-streaming, latency-bound problems: no LDS staging, no barriers in
// the K loop.  Each lane loads its MFMA operands straight from global memory
// as float4 (lane (i, half) reads k..k+3 of row i at column offset 4*half; the
// 32x32x2 MFMA then pairs k with k+4, which is as good as any pairing), the 4
// waves of a workgroup interleave 8-wide k groups, grid.y splits K further so
// that >= ~256 workgroups stream the weight matrix concurrently.
// ---------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  __shared__ float red[4][RT][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int mbase = blockIdx.z * (32 * RT);
  const int kbeg = blockIdx.y * g.kslice;
  const int kend = (kbeg + g.kslice < g.K) ? kbeg + g.kslice : g.K;
  long abase[RT];
  float amask[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int m = mbase + rt * 32 + i;
    m = m < g.M ? m : g.M - 1;
    int row = g.a_rows ? g.a_rows[m] : m;
    amask[rt] = row < 0 ? 0.f : 1.f;
    abase[rt] = (long)(row < 0 ? 0 : row) * g.lda + 4 * half;
  }
  const int n = n0 + i;
  const float *__restrict__ wp = g.W + (long)(n < g.N ? n : g.N - 1) * g.K + 4 * half;
  const float *__restrict__ ap = g.A;
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  // two 8-wide k groups per trip: 2*(1+RT) independent 16-byte loads in flight
  auto step = [&](const float4 &b4, const float4 (&a4)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].x * amask[rt], b4.x, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].y * amask[rt], b4.y, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].z * amask[rt], b4.z, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[rt].w * amask[rt], b4.w, acc[rt], 0, 0, 0);
    }
  };
  // batches of NB 8-wide k groups: all NB*(1+RT) 16-byte loads are issued
  // before the first MFMA, so a K = 256 slice costs ONE memory round trip.
  constexpr int NB = 8;
  int k = kbeg + wave * 8;
  for (; k + 32 * (NB - 1) < kend; k += 32 * NB) {
    float4 bq[NB], aq[NB][RT];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const long ko = gemm_kofs(g, k + 32 * u);
      bq[u] = *reinterpret_cast<const float4 *>(wp + k + 32 * u);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) aq[u][rt] = *reinterpret_cast<const float4 *>(ap + abase[rt] + ko);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) step(bq[u], aq[u]);
  }
  for (; k < kend; k += 32) {
    const long ko0 = gemm_kofs(g, k);
    const float4 b0 = *reinterpret_cast<const float4 *>(wp + k);
    float4 a0[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a0[rt] = *reinterpret_cast<const float4 *>(ap + abase[rt] + ko0);
    step(b0, a0);
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][rt][r][lane] = acc[rt][r];
  __syncthreads();
  // wave w finishes accumulator registers r = w, w+4, w+8, w+12 (fixed order)
  const bool relu = g.flags & SC_GEMM_RELU, resid = g.flags & SC_GEMM_RESIDUAL;
  const float bv = (g.bias && n < g.N && !g.part) ? g.bias[n] : 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wave + 4 * q;
      float v = ((red[0][rt][r][lane] + red[1][rt][r][lane]) + red[2][rt][r][lane]) + red[3][rt][r][lane];
      const int m = mbase + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m >= g.M || n >= g.N) continue;
      if (g.part) {
        g.part[((long)blockIdx.y * g.M + m) * g.N + n] = v;
      } else {
        const int crow = g.c_rows ? g.c_rows[m] : m;
        if (crow < 0) continue;
        float *p = g.C + (long)crow * g.ldc + n;
        v += bv;
        if (relu) v = fmaxf(v, 0.f);
        if (resid) v = *p + v;
        *p = v;
      }
    }
  }
}

// out[c_rows[m], n] = epilogue(bias[n] + sum_z part[z][m][n]), z in fixed order
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g, int ksplit) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = g.N >> 2;
  if (idx >= (long)g.M * n4) return;
  const int m = idx / n4, n = (idx % n4) * 4;
  // partials are fetched 8 at a time (independent loads in flight) and summed in the canonical order (common.h: sc_tree8)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z0 = 0; z0 < ksplit; z0 += 8) {
    float4 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int z = min(z0 + i, ksplit - 1);
      p[i] = *reinterpret_cast<const float4 *>(g.part + ((long)z * g.M + m) * g.N + n);
      if (z0 + i >= ksplit) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 t = sc_tree8(p);
    acc = z0 == 0 ? t : sc_add4(acc, t);
  }
  if (g.bias) {
    const float4 b = *reinterpret_cast<const float4 *>(g.bias + n);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  if (g.flags & SC_GEMM_RELU) {
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
  }
  const int crow = g.c_rows ? g.c_rows[m] : m;
  if (crow < 0) return;
  float4 *dst = reinterpret_cast<float4 *>(g.C + (long)crow * g.ldc + n);
  if (g.flags & SC_GEMM_RESIDUAL) {
    const float4 o = *dst;
    acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
  }
  *dst = acc;
}

// ... of the fused feed-forward's split sums (residual add into x) TOGETHER with the context hand-off of the contextual
// block encoder (contextual_block_encoder_layer.py:252-267; ctx_handoff_kernel, encoder.hip: slot 0 of every block <- last row of the stream's previous block, the
// first block's from / the last block's to the per-stream state): the thread that owns element n of a block's LAST row
// also writes it where the chain sends it; slot-0 rows are written by nobody else (their own sums are discarded by the
// hand-off anyway).  blkinfo[blk] = {next block of the chain or -1, bit 0: first block of a chain without saved state | bit 1: in a chain,
// state row base (stream * n_layers) if last block else -1, first block of the chain if last and the state is valid else -1}
__global__ __launch_bounds__(256) void ffn_reduce_handoff_kernel(GemmArgs g, int ksplit, ScHandoff h) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = g.N >> 2;
  if (idx >= (long)g.M * n4) return;
  const int m = idx / n4, n = (idx % n4) * 4;
  const int blk = m / h.R, r = m % h.R;
  int4 info = make_int4(-1, 0, -1, -1);
  if (r == h.R - 1 || r == 0) info = *reinterpret_cast<const int4 *>(h.blkinfo + 4 * blk);
  // slot 0 of a block that belongs to a chain is written by the chain (its own sums are discarded by the hand-off); a
  // block OUTSIDE every chain (info.y bit 1 clear: not covered by the job table) keeps the plain reduce for its slot 0
  if (r == 0 && (info.y & 2)) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z0 = 0; z0 < ksplit; z0 += 8) {
    float4 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int z = min(z0 + i, ksplit - 1);
      p[i] = *reinterpret_cast<const float4 *>(g.part + ((long)z * g.M + m) * g.N + n);
      if (z0 + i >= ksplit) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 t = sc_tree8(p);
    acc = z0 == 0 ? t : sc_add4(acc, t);
  }
  if (g.bias) {
    const float4 b = *reinterpret_cast<const float4 *>(g.bias + n);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  }
  float4 *dst = reinterpret_cast<float4 *>(g.C + (long)m * g.ldc + n);
  const float4 o = *dst;
  acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
  *dst = acc;
  if (r != h.R - 1) return;
  if (info.x >= 0) *reinterpret_cast<float4 *>(g.C + (long)info.x * h.R * g.ldc + n) = acc;
  if (info.y & 1) *reinterpret_cast<float4 *>(g.C + (long)blk * h.R * g.ldc + n) = acc;
  if (info.z >= 0) {
    float4 *st = reinterpret_cast<float4 *>(h.state + (long)(info.z + h.layer) * g.N + n);
    if (info.w >= 0) *reinterpret_cast<float4 *>(g.C + (long)info.w * h.R * g.ldc + n) = *st;
    *st = acc;
  }
}

// split-K reduce fused with the LayerNorm that follows the residual add:
// one wave per row (N <= 1024).  C[m] = epilogue(sum_z part[z][m]),
// ln_out[m] = LN(C[m]) * gamma + beta.
__global__ __launch_bounds__(256) void gemm_splitk_reduce_ln_kernel(GemmArgs g, int ksplit,
                                                                    const float *gamma, const float *beta,
                                                                    float eps, float *ln_out, int ld_ln) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= g.M) return;
  const int nv = g.N >> 2;
  const int crow = g.c_rows ? g.c_rows[m] : m;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      // partials are fetched 8 at a time (independent loads in flight) and summed in the canonical order (common.h: sc_tree8)
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int z0 = 0; z0 < ksplit; z0 += 8) {
        float4 p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int z = min(z0 + q, ksplit - 1);
          p[q] = reinterpret_cast<const float4 *>(g.part + ((long)z * g.M + m) * g.N)[c];
          if (z0 + q >= ksplit) p[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 t = sc_tree8(p);
        acc = z0 == 0 ? t : sc_add4(acc, t);
      }
      if (g.bias) {
        const float4 b = reinterpret_cast<const float4 *>(g.bias)[c];
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      }
      if (g.flags & SC_GEMM_RELU) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
      }
      if (crow >= 0) {
        float4 *dst = reinterpret_cast<float4 *>(g.C + (long)crow * g.ldc) + c;
        if (g.flags & SC_GEMM_RESIDUAL) {
          const float4 o = *dst;
          acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
        }
        *dst = acc;
      }
      v[i] = acc;
      s += (acc.x + acc.y) + (acc.z + acc.w);
    }
  }
  const float mean = wave_sum(s) / (float)g.N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + e * e);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)g.N + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < nv) {
      const float4 gm = reinterpret_cast<const float4 *>(gamma)[c];
      const float4 bt = reinterpret_cast<const float4 *>(beta)[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
      const long lrow = (g.flags & SC_GEMM_LN_AT_CROWS) ? crow : m;
      if (lrow >= 0) reinterpret_cast<float4 *>(ln_out + lrow * ld_ln)[c] = o;
    }
  }
}

// scalar reference kernel: one thread per output, k-ordered fmaf chain
__global__ void gemm_naive_kernel(GemmArgs g) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)g.M * g.N) return;
  int m = idx / g.N, n = idx % g.N;
  int row = g.a_rows ? g.a_rows[m] : m;
  float acc = 0.f;
  if (row >= 0) {
    const float *a = g.A + (long)row * g.lda;
    const float *w = g.W + (long)n * g.K;
    for (int k = 0; k < g.K; ++k) {
      long ko = k;
      if (g.conv_f1 > 0) {
        int tap = k / g.lda;
        ko = (long)((tap / 3) * g.conv_f1 + tap % 3) * g.lda + (k % g.lda);
      }
      acc = fmaf(a[ko], w[k], acc);
    }
  }
  float v = acc + (g.bias ? g.bias[n] : 0.f);
  if (g.flags & SC_GEMM_RELU) v = fmaxf(v, 0.f);
  int crow = g.c_rows ? g.c_rows[m] : m;
  if (crow < 0) return;
  float *p = g.C + (long)crow * g.ldc + n;
  if (g.flags & SC_GEMM_RESIDUAL) v = *p + v;
  *p = v;
}

static int g_force_naive = -1;

// ---- optional per-launch timing with HIP events (bench.py roofline leg) ----
struct ProfRec { hipEvent_t a, b; double flops, bytes; int variant; };
static std::vector<ProfRec> g_recs;
static int g_prof_every = 0;
static long long g_gemm_calls = 0;

// shared by every launcher of the library (common.h): events around ONE kernel launch
ProfScope sc_prof_begin(hipStream_t st) {
  ProfScope p{false, nullptr, nullptr, st};
  if (g_prof_every <= 0 || (g_gemm_calls++ % g_prof_every) != 0) return p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;   // never record timing events into a stream capture
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return p;
  (void)hipEventCreate(&p.a);
  (void)hipEventCreate(&p.b);
  (void)hipEventRecord(p.a, st);
  p.on = true;
  return p;
}
void sc_prof_end(ProfScope &p, int kind, double flops, double bytes) {
  if (!p.on) return;
  (void)hipEventRecord(p.b, p.st);
  g_recs.push_back(ProfRec{p.a, p.b, flops, bytes, kind});
  p.on = false;
}

extern "C" int sc_prof_enable(int sample_every) {
  g_prof_every = sample_every;
  g_gemm_calls = 0;
  return SC_OK;
}

// Median elapsed time of an EMPTY hipEvent pair on `stream` (what bracketing a
// launch with events adds to its measured duration); used to de-bias
// sc_prof_collect for kernels that only run a few microseconds.
extern "C" double sc_prof_event_overhead_ms(void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const int n = 64;
  float t[n];
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  for (int i = 0; i < n; ++i) {
    (void)hipEventRecord(a, st);
    (void)hipEventRecord(b, st);
    (void)hipEventSynchronize(b);
    t[i] = 0.f;
    (void)hipEventElapsedTime(&t[i], a, b);
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  for (int i = 1; i < n; ++i)  // insertion sort, n is tiny
    for (int j = i; j > 0 && t[j] < t[j - 1]; --j) { float x = t[j]; t[j] = t[j - 1]; t[j - 1] = x; }
  return (double)t[n / 2];
}

// ms[v], flops[v], n[v] per kernel kind v (scasr.h: SC_PROF_*), nkinds entries
extern "C" int sc_prof_collect_kinds(double *ms, double *flops, double *bytes, long long *n, int nkinds) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { sc_set_error("sc_prof_collect: %s", hipGetErrorString(e)); return SC_ERR_LAUNCH; }
  for (int v = 0; v < nkinds; ++v) { ms[v] = 0; flops[v] = 0; n[v] = 0; if (bytes) bytes[v] = 0; }
  for (auto &r : g_recs) {
    float t = 0.f;
    if (r.variant < nkinds && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.variant] += t;
      flops[r.variant] += r.flops;
      if (bytes) bytes[r.variant] += r.bytes;
      n[r.variant] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  return SC_OK;
}

static int g_bk64 = 0;               // SC_GEMM_BK=64: 64-deep K tiles for the 64x64 kernel (A/B switch)
static int g_skinny_max_m = 64;     // SC_SKINNY_MAX_M overrides (A-B tests: the LDS-tiled kernel wins for M > 64, docs/profiles_r1-r3/r01_gemm_skinny_ab.txt)
// split-K workspaces: one per HIP stream (independent StreamBatches run
// concurrently on their own streams), plus a default for unregistered streams
struct Workspace { float *ptr; size_t bytes; };
static Workspace g_ws_default{nullptr, 0};
static std::unordered_map<void *, Workspace> g_ws_by_stream;
static std::mutex g_ws_mutex;
static thread_local float *g_ws = nullptr;       // resolved per sc_gemm call
static thread_local size_t g_ws_bytes = 0;

const char *sc_hook(const char *name) {
  static const bool enabled = [] {
    const char *e = getenv("SC_TEST_HOOKS");
    return e && atoi(e) != 0;
  }();
  return enabled ? getenv(name) : nullptr;
}

extern "C" int sc_set_workspace(void *ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  g_ws_default = Workspace{(float *)ptr, bytes};
  return SC_OK;
}

extern "C" int sc_set_stream_workspace(void *stream, void *ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (ptr) g_ws_by_stream[stream] = Workspace{(float *)ptr, bytes};
  else g_ws_by_stream.erase(stream);
  return SC_OK;
}

static void resolve_workspace(void *stream) {
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  auto it = g_ws_by_stream.find(stream);
  const Workspace &w = it != g_ws_by_stream.end() ? it->second : g_ws_default;
  g_ws = w.ptr;
  g_ws_bytes = w.bytes;
}

extern "C" size_t sc_workspace_bytes(void *stream) {
  resolve_workspace(stream);
  return g_ws ? g_ws_bytes : 0;
}

// Launches the main kernel.  force_part: always leave raw partial sums in the
// workspace (ksplit >= 1) for a fused reduce epilogue.  Returns the number of
// K slices through *ksplit_out (0: the kernel wrote the final result itself).
static int gemm_dispatch(GemmArgs &g, bool force_part, int *ksplit_out, int *variant_out, hipStream_t st) {
  const int M = g.M, N = g.N, K = g.K;
  if (g_force_naive < 0) {
    const char *e = sc_hook("SC_GEMM_NAIVE");
    g_force_naive = (e && e[0] == '1') ? 1 : 0;
    if (const char *m = sc_hook("SC_SKINNY_MAX_M")) g_skinny_max_m = atoi(m);
    if (const char *m = sc_hook("SC_GEMM_BK")) g_bk64 = atoi(m) == 64;
  }
  bool aligned = (K % 32 == 0) && (g.lda % 4 == 0) && (((uintptr_t)g.A & 15) == 0) &&
                 (((uintptr_t)g.W & 15) == 0) && (g.conv_f1 == 0 || g.lda % 32 == 0);
  // variant: 0 scalar, 1 skinny register-direct (tools only), 2 = 128x128 tile, 3 = 64x64 tile
  // CANONICAL SUMMATION ORDER (round 5, bit-reproducible serving): how the K dimension of a product is cut and in which
  // order its pieces are added is a function of (N, K) ALONE - never of M, which is the number of rows that happen to be
  // in flight (streams of an encoder group, frames of a chunk).  Every aligned problem takes the LDS-tiled kernel (k runs
  // in order inside a slice, both tile sizes issue the same v_mfma_f32_32x32x2_f32 chain per output element); K >= 2560
  // (the subsampling Linear: K = 19 * d) is cut into 8 slices that the reduce kernel adds as a tree (common.h), anything
  // shorter is one chain.  The register-direct skinny kernel (round 1-4: M <= 64) interleaves k over its waves and is
  // kept for A/B runs only (SC_SKINNY_MAX_M).
  int variant;
  const bool can_part_any = g_ws && (N % 4 == 0) && (g.ldc % 4 == 0) && (((uintptr_t)g.C & 15) == 0) &&
                            (!g.bias || (((uintptr_t)g.bias & 15) == 0)) &&
                            (size_t)M * N * sizeof(float) <= g_ws_bytes;
  int ksplit = 1;
  const bool skinny_hook = sc_hook("SC_SKINNY_MAX_M") != nullptr;
  if ((g.flags & SC_GEMM_NAIVE) || g_force_naive || !aligned) {
    variant = 0;
  } else if (skinny_hook && (M <= 64 || (M <= g_skinny_max_m && (K <= 256 || can_part_any))) && M <= g_skinny_max_m) {
    variant = 1;
    if (can_part_any) {
      if (M <= 64) {
        const int ct = cdiv(N, 32);
        ksplit = K / 64;
        const int want = cdiv(256, ct);
        if (ksplit > want) ksplit = want;
      } else {
        ksplit = K / 256;
      }
      if (ksplit < 1) ksplit = 1;
    }
  } else {
    // tile size by a small cost model in cycles (wave quantisation dominates these small GEMMs): it changes which
    // workgroup computes an element, not how
    if (K >= 2560 && can_part_any && (size_t)8 * M * N * sizeof(float) <= g_ws_bytes) ksplit = 8;   // (callers slab M: gemm_slab_rows)
    const double c0 = 4.0;
    double best = 1e30;
    variant = 3;
    for (int v = 2; v <= 3; ++v) {
      const int bm = v == 2 ? 128 : 64;
      const long tiles = (long)cdiv(M, bm) * cdiv(N, bm);
      const long slots = v == 2 ? 512 : 1024;        // resident workgroups on 256 CUs
      const double step1 = v == 2 ? 4096.0 : 1024.0;   // cycles per K-step, one workgroup on a CU
      const int per_cu = v == 2 ? 2 : 4;
      const long work = tiles * ksplit;
      const long full = work / slots, rest = work % slots;
      const double steps = (double)cdiv(K / 32, ksplit) + c0;
      double t = full * steps * step1 * per_cu;
      if (rest) t += steps * step1 * (double)cdiv((int)rest, 256);
      if (t < best) { best = t; variant = v; }
    }
  }
  const bool can_part = can_part_any && variant != 0;
  g.kslice = K;
  if (ksplit > 1) {
    g.kslice = cdiv(K / 32, ksplit) * 32;
    ksplit = cdiv(K, g.kslice);
  }
  const bool part = can_part && (ksplit > 1 || force_part);
  g.part = part ? g_ws : nullptr;
  if (!part) { ksplit = 1; g.kslice = K; }
  if (variant == 0) {
    long total = (long)M * N;
    gemm_naive_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(g);
  } else if (variant == 1) {
    if (M <= 32) gemm_skinny_kernel<1><<<dim3(cdiv(N, 32), ksplit, 1), 256, 0, st>>>(g);
    else gemm_skinny_kernel<2><<<dim3(cdiv(N, 32), ksplit, cdiv(M, 64)), 256, 0, st>>>(g);
  } else if (variant == 2) {
    if (g.flags & SC_GEMM_SPLIT16)
      gemm_mfma_kernel<128, 128, 2, 2, 32, true><<<dim3(cdiv(N, 128), cdiv(M, 128), ksplit), 256, 0, st>>>(g);
    else
      gemm_mfma_kernel<128, 128, 2, 2><<<dim3(cdiv(N, 128), cdiv(M, 128), ksplit), 256, 0, st>>>(g);
  } else if (g.flags & SC_GEMM_SPLIT16) {
    gemm_mfma_kernel<64, 64, 2, 2, 32, true><<<dim3(cdiv(N, 64), cdiv(M, 64), ksplit), 256, 0, st>>>(g);
  } else {
    if (g_bk64 && g.kslice % 64 == 0 && K % 64 == 0 && (g.conv_f1 == 0 || g.lda % 64 == 0))
      gemm_mfma_kernel<64, 64, 2, 2, 64><<<dim3(cdiv(N, 64), cdiv(M, 64), ksplit), 256, 0, st>>>(g);
    else
      gemm_mfma_kernel<64, 64, 2, 2><<<dim3(cdiv(N, 64), cdiv(M, 64), ksplit), 256, 0, st>>>(g);
  }
  *ksplit_out = part ? ksplit : 0;
  *variant_out = variant;
  return SC_OK;
}

extern "C" int sc_prof_collect2(double *ms, double *flops, double *bytes, long long *n) {
  return sc_prof_collect_kinds(ms, flops, bytes, n, 4);
}

extern "C" int sc_prof_collect(double *ms, double *flops, long long *n) {
  return sc_prof_collect2(ms, flops, nullptr, n);
}

// rows per launch: a split-K product (K >= 2560: 8 slices, gemm_dispatch) whose partial sums do not fit the workspace is
// computed in row slabs - the K split, and with it the summation order, never gives way to the row count
static int gemm_slab_rows(int M, int N, int K) {
  if (K < 2560 || !g_ws || (size_t)8 * M * N * sizeof(float) <= g_ws_bytes) return M;
  const long fit = (long)(g_ws_bytes / ((size_t)8 * N * sizeof(float))) / 128 * 128;
  return fit >= 128 ? (int)fit : M;
}

extern "C" int sc_gemm(const float *A, const int32_t *a_rows, int lda, const float *W,
                       const float *bias, float *C, const int32_t *c_rows, int ldc, int M, int N,
                       int K, int flags, int conv_f1, void *stream) {
  SC_CHECK_ARG(A && W && C, "null pointer");
  SC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && lda > 0 && ldc >= N, "bad dimensions");
  if (M == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  resolve_workspace(stream);
  ProfScope prof = sc_prof_begin(st);
  int variant = 0;
  const int slab = gemm_slab_rows(M, N, K);
  for (int m0 = 0; m0 < M; m0 += slab) {
    const int mm = M - m0 < slab ? M - m0 : slab;
    GemmArgs g{a_rows ? A : A + (long)m0 * lda, a_rows ? a_rows + m0 : nullptr, lda, W, bias, c_rows ? C : C + (long)m0 * ldc,
               c_rows ? c_rows + m0 : nullptr, ldc, mm, N, K, flags, conv_f1, nullptr, K};
    int ksplit = 0;
    gemm_dispatch(g, false, &ksplit, &variant, st);
    if (ksplit > 0) {
      const long n4 = (long)mm * (N / 4);
      gemm_splitk_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(g, ksplit);
    }
  }
  // algorithmic bytes: read A and W once, write C (read it too for the residual form)
  sc_prof_end(prof, variant, 2.0 * M * N * K,
              4.0 * ((double)M * K + (double)N * K + (double)M * N * ((flags & SC_GEMM_RESIDUAL) ? 2.0 : 1.0)));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// C block j (columns j*ncb .. of the product, j < N / ncb) = A W[j*ncb ..]^T + bias[j*ncb ..], block j stored at
// C + j * cb_stride with rows through c_rows / ldc like sc_gemm: `N / ncb` products that share A in ONE launch (K < 2560:
// a single k chain per element - the same sums, bit for bit, as one sc_gemm per block).  W: [N][K], the blocks' weights
// one behind the other.  Returns SC_ERR_ARG when the problem does not take the tiled kernel (the caller loops instead).
int sc_gemm_colblocks(const float *A, const int32_t *a_rows, int lda, const float *W, const float *bias, float *C,
                      const int32_t *c_rows, int ldc, int M, int N, int K, int ncb, long cb_stride, int flags, void *stream) {
  SC_CHECK_ARG(A && W && C, "null pointer");
  SC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K < 2560 && lda > 0 && ncb > 0 && N % ncb == 0 && ncb % 128 == 0 && ldc >= ncb,
               "bad dimensions");
  SC_CHECK_ARG(!(flags & (SC_GEMM_RESIDUAL | SC_GEMM_NAIVE)) && (K % 32 == 0) && (lda % 4 == 0) &&
                   (((uintptr_t)A & 15) == 0) && (((uintptr_t)W & 15) == 0),
               "not a tiled problem");
  if (M == 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  resolve_workspace(stream);
  if (sc_hook("SC_GEMM_NAIVE") || sc_hook("SC_SKINNY_MAX_M") || sc_hook("SC_KV_PER_LAYER")) return SC_ERR_ARG;   // A/B and test forms
  ProfScope prof = sc_prof_begin(st);
  GemmArgs g{A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, flags, 0, nullptr, K, ncb, cb_stride};
  int ksplit = 0, variant = 0;
  float *ws_keep = g_ws;
  g_ws = nullptr;   // no split-K form here
  gemm_dispatch(g, false, &ksplit, &variant, st);
  g_ws = ws_keep;
  sc_prof_end(prof, variant, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// GEMM whose output rows are immediately layer-normalised (pre-LN transformer:
// x += proj(...); xn = LN(x)).  ln_out[m] (dense, leading dim ld_ln) receives
// LN(C[c_rows[m]]).  Fused into the split-K reduce when possible, otherwise
// GEMM followed by a LayerNorm launch.
extern "C" int sc_gemm_ln(const float *A, const int32_t *a_rows, int lda, const float *W,
                          const float *bias, float *C, const int32_t *c_rows, int ldc, int M, int N,
                          int K, int flags, int conv_f1, const float *ln_g, const float *ln_b,
                          float ln_eps, float *ln_out, int ld_ln, void *stream) {
  SC_CHECK_ARG(A && W && C && ln_g && ln_b && ln_out, "null pointer");
  SC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && lda > 0 && ldc >= N && ld_ln >= N, "bad dimensions");
  if (M == 0) return SC_OK;
  GemmArgs g{A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, flags, conv_f1, nullptr, K};
  hipStream_t st = (hipStream_t)stream;
  resolve_workspace(stream);
  const bool fusable = N <= 1024 && N % 4 == 0 && ld_ln % 4 == 0;
  int ksplit = 0, variant = 0;
  gemm_dispatch(g, fusable, &ksplit, &variant, st);
  if (ksplit > 0 && fusable) {
    gemm_splitk_reduce_ln_kernel<<<cdiv(M, 4), 256, 0, st>>>(g, ksplit, ln_g, ln_b, ln_eps, ln_out, ld_ln);
    SC_CHECK_LAUNCH();
    return SC_OK;
  }
  if (ksplit > 0) {
    const long n4 = (long)M * (N / 4);
    gemm_splitk_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(g, ksplit);
  }
  SC_CHECK_LAUNCH();
  return sc_layernorm(C, c_rows, ldc, ln_out, (flags & SC_GEMM_LN_AT_CROWS) ? c_rows : nullptr, ld_ln, M, N, ln_g,
                      ln_b, ln_eps, stream);
}

// ===========================================================================
// Fused position-wise feed-forward:  x += W2 . relu(W1 . xn + b1) + b2  (+ LN)
// (reference: feed_forward.py:48-50 behind the residual of encoder_layer /
// decoder_layer / contextual_block_encoder_layer).
//
// One workgroup owns RT = 16*RTT rows and CPW consecutive 128-wide chunks of
// the hidden dimension.  Per chunk:  h = relu(xn_tile . W1c^T + b1c) goes to
// LDS, never to HBM, and is immediately contracted with W2c; the RT x D
// partial result stays in MFMA accumulators across the CPW chunks.  Partials
// of the F/128/CPW chunk groups land in the split-K workspace and are reduced
// in fixed order by the same reduce(+LayerNorm) kernels as the split-K GEMM.
//   * MFMA v_mfma_f32_16x16x4_f32; A operands (activations) from LDS with the
//     k-permuted 32-byte-per-lane fetch, B operands (weights) straight from
//     HBM/L2 into registers out of the fragment-packed copies of W1 / W2
//     (sc_pack_panel_weight): every wave load is 1 KB contiguous;
//   * 8 waves: wave w owns hidden columns [16w,16w+16) of the chunk in GEMM 1
//     and output columns [w*D/8, (w+1)*D/8) in GEMM 2.
// Algorithmic work per row: 4*D*F flop; HBM bytes per row ~ 2*D*4 (+ the
// partial round trip 2*D*4*F/128/CPW).
// ===========================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef SC_FFN_EARLY_B2
#define SC_FFN_EARLY_B2 1
#endif

__device__ __forceinline__ f32x4 ffn_mfma8(f32x4 acc, const float4 &a0, const float4 &a1, const float4 &b0,
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

struct FfnArgs {
  const float *XN;
  const int *rows;
  const float *W1p, *b1, *W2p;
  float *part;  // [F/128/cpw][M][D]
  int M, F, cpw;
  // PRO (decoder layer, sc_dec_layer_ffn): the row tile is computed instead of loaded -
  //   x[row] = Xin[row] + (sum_{h < nph} PH[(row*nph + h)*D + :] + pbias);  tile = LayerNorm(x; g, b)
  // workgroups of chunk group 0 store x to Xout (!= Xin); the partial sums are stored by ROW ID
  // ([grp][part_rows][D]) so that a per-stream consumer finds them without the compaction map
  const float *PH;
  int nph;
  const float *pbias, *Xin;
  float *Xout;
  const float *g, *b;
  float eps;
  int part_rows;
  int w_form;   // 0: fp32 weights; 1: W1p / W2p hold fp16 elements (same fragment order): fp16 MFMA inputs, fp32 accumulation;
                // 2: W1p / W2p hold the fp16 hi | lo SPLIT of the fp32 weights (see WF below)
  int act_half; // PRO: PH and part hold fp16 elements (same element offsets; sc_search.act_half)
  int dbg_stamp;   // SC_PHASE_DBG builds: this launch leaves phase stamps (common.h)
  int pgrp;     // PRO: heads per partial product of PH the consumer still has to group: 4 = one partial per head (sum aligned groups
                // of four first), 1 = the producer summed them (common.h: canonical order of the head partials)
};

// WF = 1 (WH): fp16 weights (the same fragment order, 2-byte elements) and fp16 MFMA inputs (v_mfma_f32_16x16x16_f16: the
// 8 k values a lane holds per 32-wide k block feed 2 instructions instead of 8), fp32 accumulation, fp32 partial sums
// and LayerNorms.  The row tile and the hidden activations then live in LDS as fp16 (converted once, when staged).
__device__ __forceinline__ f32x4 ffn_mma_h(f32x4 acc, const h16x8 &a, const h16x4 &b0, const h16x4 &b1) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h16x4{a[0], a[1], a[2], a[3]}, b0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h16x4{a[4], a[5], a[6], a[7]}, b1, acc, 0, 0, 0);
  return acc;
}
// WF = 2 (WS): fp32-grade arithmetic on the fp16 matrix pipe.  Every fp32 operand x is split into two fp16 numbers,
//   hi = fp16(x),  lo = fp16((x - hi) * 2^11)        (x - hi is exact in fp32; x = hi + lo / 2^11 to ~2^-23 |x|)
// and a product sum a.b is evaluated as  sum(a_hi b_hi) + 2^-11 (sum(a_hi b_lo) + sum(a_lo b_hi))  with three
// v_mfma_f32_16x16x32_f16 (fp16 products are exact in the fp32 accumulators) instead of eight v_mfma_f32_16x16x4_f32:
// 51 instead of 256 matrix-pipe cycles per 8 k values of a lane.  Only the a_lo b_lo term (<= 2^-22 |a b|) is
// dropped: the result differs from fp32 arithmetic by a few ulp of fp32 (tests/test_gpu_ops.py), not by fp16's
// 2^-11.  The weights are split offline (weights.py: the hi and lo halves of a lane's 8 k values take the places of
// the two fp32 slabs of the fragment order, same bytes), the row tile and the hidden activations when they are staged
// into LDS (hi8 | lo8 per 8 k values: the same 32 bytes per lane and row stride as the fp32 tile).  Operands must lie
// within fp16's range (|x| < 65504: LayerNorm outputs and ReLU activations of any real model do).
// (ffn_split4: top of the file)
// LDS bytes of one workgroup (host and device); wf: 0 fp32, 1 fp16, 2 split
__host__ __device__ static inline size_t ffn_lds_bytes(int D, int RT, bool pro, int wf) {
  const bool wh = wf == 1;
  if (!wh) return (size_t)(RT * (D + 4) + RT * (128 + 4) + (pro ? RT : 0)) * sizeof(float);
  return (size_t)RT * (D + 4) * 4 + (size_t)RT * (D + 8) * 2 + (size_t)RT * (128 + 8) * 2 + (pro ? RT * 4 : 0);
}

template <int D, int RTT, bool PRO = false, int WF = 0>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnArgs p) {
  constexpr bool WH = WF == 1, WS = WF == 2;
  constexpr int RT = 16 * RTT, FC = 128;
  constexpr int KI1 = D / 32, KI2 = FC / 32, NT2 = D / 128;
  constexpr int LDX = D + 4, LDH = FC + 4;
  constexpr int LDXH = D + 8, LDHH = FC + 8;   // WH: row strides of the fp16 tiles (elements)
  constexpr int LDXS = 2 * D + 8, LDHS = 2 * FC + 8;   // WS: row strides of the hi8 | lo8 tiles (fp16 elements; = the fp32 bytes)
  extern __shared__ __attribute__((aligned(16))) float ffn_smem[];
  float *Xs = ffn_smem;            // [RT][LDX]  xn tile (WH: fp32 staging of the prologue only); re-used to stage the partial result
  float *Hs = ffn_smem + RT * LDX; // [RT][LDH]  relu(h) of the current chunk (not WH)
  _Float16 *XsH = reinterpret_cast<_Float16 *>(ffn_smem + RT * LDX);   // WH: [RT][LDXH] xn tile
  _Float16 *HsH = XsH + RT * LDXH;                                       // WH: [RT][LDHH] relu(h)
  int *rowid = WH ? reinterpret_cast<int *>(HsH + RT * LDHH)
                  : reinterpret_cast<int *>(ffn_smem + RT * LDX + RT * LDH);   // PRO: [RT] row ids of the tile
  _Float16 *XsS = reinterpret_cast<_Float16 *>(ffn_smem);               // WS: [RT][LDXS] xn tile, IN PLACE of Xs
  _Float16 *HsS = reinterpret_cast<_Float16 *>(ffn_smem + RT * LDX);    // WS: [RT][LDHS] relu(h), in place of Hs
  // 4 weight elements (WS: the hi, in the second slab the lo, halves of the 8 k values of a lane)
  typedef typename std::conditional<WS, h16x8, typename std::conditional<WH, h16x4, float4>::type>::type BF;
  // WS: 4 consecutive columns c.. of a row -> hi4 at, lo4 8 elements behind, this offset
  auto split_at = [](int c) { return (c >> 3) * 16 + (c & 7); };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int grp = blockIdx.x, m0 = blockIdx.y * RT;
  const int KF = p.F / 32;  // k-blocks of W2's packed layout

  // xn tile -> LDS (coalesced, rows through the optional table).  Three separate
  // unrolled phases so that all row-index loads, then all data loads are in flight together.
  constexpr int NV = RT * D / 4;
  constexpr int NQ = NV / 512;
  static_assert(NV % 512 == 0, "tile must be a multiple of the workgroup");
  SC_STAMP(PRO ? 2 : 3, 0);
  BF bf[KI1][2];  // GEMM 1 weights of this wave's 16 hidden columns (all of K)
  auto load_b1 = [&](int chunk) {
    const BF *wp = reinterpret_cast<const BF *>(p.W1p) + ((long)(chunk * 8 + wave) * KI1) * 128 + lane;
#pragma unroll
    for (int ki = 0; ki < KI1; ++ki) {
      bf[ki][0] = wp[ki * 128];
      bf[ki][1] = wp[ki * 128 + 64];
    }
  };
  // (requesting the first chunk's W1 / W2 fragments ahead of the prologue's partial sums was measured and lost:
  // loads return in issue order, the prologue then waits for 256 KB of weights - 197 -> 233 stamp units per launch;
  // the GEMM phase itself is bound by the 256 MFMAs per SIMD of a 16-row tile, not by its loads)
  BF b2f[KI2][NT2][2];
  auto load_b2 = [&](int chunk) {
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      const BF *wp = reinterpret_cast<const BF *>(p.W2p) +
                         ((long)(wave * NT2 + t) * KF + chunk * KI2) * 128 + lane;
#pragma unroll
      for (int k = 0; k < KI2; ++k) {
        b2f[k][t][0] = wp[k * 128];
        b2f[k][t][1] = wp[k * 128 + 64];
      }
    }
  };
  if (PRO) {
    // sum of the producer's per-head partial products (fixed head order) + residual, then LayerNorm - recomputed
    // by every chunk group of the row tile (cheap, L2-resident) so that the attention output projection needs no
    // launch of its own.  Element-parallel: every thread owns float4 pieces of the RT x D tile and has the loads of
    // all head partials of a piece in flight together (one memory round trip behind the row ids); the LayerNorm
    // then runs on the LDS tile with 16 lanes per row (DPP reductions).
    constexpr int C4 = D / 4, Q4 = D / 64;
    const int sub = threadIdx.x & 15;
    float4 gam[Q4], bet[Q4];   // issued first: they have arrived when the partial sums have
#pragma unroll
    for (int q = 0; q < Q4; ++q) {
      gam[q] = *reinterpret_cast<const float4 *>(p.g + 4 * (sub + 16 * q));
      bet[q] = *reinterpret_cast<const float4 *>(p.b + 4 * (sub + 16 * q));
    }
    long rowv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = min(m0 + (threadIdx.x + q * 512) / C4, p.M - 1);
      rowv[q] = p.rows ? p.rows[m] : m;
    }
    SC_STAMP_WAIT(2, 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * 512, i = e / C4, c4 = e % C4;
      const long row = rowv[q];
      if (c4 == 0) rowid[i] = (int)row;
      const float4 xi = *reinterpret_cast<const float4 *>(p.Xin + row * D + 4 * c4);
      float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.pbias) pb = *reinterpret_cast<const float4 *>(p.pbias + 4 * c4);
      float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
      auto ldp = [&](long elem) -> float4 {
        if (p.act_half) {
          const h16x4 h = *reinterpret_cast<const h16x4 *>(reinterpret_cast<const _Float16 *>(p.PH) + elem);
          return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
        return *reinterpret_cast<const float4 *>(p.PH + elem);
      };
      // canonical order of the head partials (common.h): aligned groups of four heads in head order, the groups in order.
      // pgrp = 1: the producer (four heads per workgroup) has summed each group already
      if (p.pgrp == 1 && p.nph == 2) {
        const float4 p0 = ldp((row * 2) * D + 4 * c4);
        const float4 p1 = ldp((row * 2 + 1) * D + 4 * c4);
        y = sc_add4(p0, p1);
      } else
      for (int z0 = 0; z0 < p.nph; z0 += 8) {
        float4 pv[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          pv[z] = ldp((row * p.nph + min(z0 + z, p.nph - 1)) * D + 4 * c4);
          if (z0 + z >= p.nph) pv[z] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (p.pgrp == 4) {
          const float4 g0 = sc_seq4(pv[0], pv[1], pv[2], pv[3]);
          y = z0 == 0 ? g0 : sc_add4(y, g0);
          if (z0 + 4 < p.nph) y = sc_add4(y, sc_seq4(pv[4], pv[5], pv[6], pv[7]));
        } else {
#pragma unroll
          for (int z = 0; z < 8; ++z)
            if (z0 + z < p.nph) y = (z0 + z == 0) ? pv[0] : sc_add4(y, pv[z]);
        }
      }
      const float4 x = make_float4(xi.x + (y.x + pb.x), xi.y + (y.y + pb.y), xi.z + (y.z + pb.z), xi.w + (y.w + pb.w));
      if (grp == 0 && m0 + i < p.M) *reinterpret_cast<float4 *>(p.Xout + row * D + 4 * c4) = x;
      *reinterpret_cast<float4 *>(Xs + i * LDX + 4 * c4) = x;
    }
    SC_STAMP_WAIT(2, 2);
    load_b1(grp * p.cpw);   // GEMM 1's first weight fragments travel during the LayerNorm (the partial sums have arrived)
    __syncthreads();
    SC_STAMP(2, 3);
    for (int i = threadIdx.x >> 4; i < RT; i += 32) {   // uniform per 16-lane row group
      float4 x[Q4];
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        x[q] = *reinterpret_cast<const float4 *>(Xs + i * LDX + 4 * (sub + 16 * q));
        sum += (x[q].x + x[q].y) + (x[q].z + x[q].w);
      }
      const float mean = group_sum<16>(sum) / (float)D;
      float q2 = 0.f;
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        const float a = x[q].x - mean, b = x[q].y - mean, c = x[q].z - mean, e = x[q].w - mean;
        q2 += (a * a + b * b) + (c * c + e * e);
      }
      const float rstd = 1.0f / sqrtf(group_sum<16>(q2) / (float)D + p.eps);
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        const float4 o = make_float4((x[q].x - mean) * rstd * gam[q].x + bet[q].x, (x[q].y - mean) * rstd * gam[q].y + bet[q].y,
                                     (x[q].z - mean) * rstd * gam[q].z + bet[q].z, (x[q].w - mean) * rstd * gam[q].w + bet[q].w);
        if (WS) {   // in place of the fp32 row this 16-lane group has just read
          h16x4 hi, lo;
          ffn_split4(o, hi, lo);
          _Float16 *d = XsS + i * LDXS + split_at(4 * (sub + 16 * q));
          *reinterpret_cast<h16x4 *>(d) = hi;
          *reinterpret_cast<h16x4 *>(d + 8) = lo;
        } else if (WH)
          *reinterpret_cast<h16x4 *>(XsH + i * LDXH + 4 * (sub + 16 * q)) =
              h16x4{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
        else
          *reinterpret_cast<float4 *>(Xs + i * LDX + 4 * (sub + 16 * q)) = o;
      }
    }
    SC_STAMP(2, 4);
  } else {
    long rowv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = min(m0 + (threadIdx.x + q * 512) / (D / 4), p.M - 1);
      rowv[q] = p.rows ? p.rows[m] : m;
    }
    float4 stage[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      stage[q] = *reinterpret_cast<const float4 *>(p.XN + rowv[q] * D + 4 * ((threadIdx.x + q * 512) % (D / 4)));
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * 512;
      if (WS) {
        h16x4 hi, lo;
        ffn_split4(stage[q], hi, lo);
        _Float16 *d = XsS + (e / (D / 4)) * LDXS + split_at(4 * (e % (D / 4)));
        *reinterpret_cast<h16x4 *>(d) = hi;
        *reinterpret_cast<h16x4 *>(d + 8) = lo;
      } else if (WH)
        *reinterpret_cast<h16x4 *>(XsH + (e / (D / 4)) * LDXH + 4 * (e % (D / 4))) =
            h16x4{(_Float16)stage[q].x, (_Float16)stage[q].y, (_Float16)stage[q].z, (_Float16)stage[q].w};
      else
        *reinterpret_cast<float4 *>(Xs + (e / (D / 4)) * LDX + 4 * (e % (D / 4))) = stage[q];
    }
  }
  f32x4 acc2[RTT][NT2];
  f32x4 acc2c[WS ? RTT : 1][WS ? NT2 : 1];   // WS: the 2^11-scaled cross terms
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      acc2[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (WS) acc2c[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  if (!PRO) load_b1(grp * p.cpw);
  __syncthreads();
  SC_STAMP(PRO ? 2 : 3, PRO ? 5 : 1);
  if (PRO) SC_STAMP_WAIT(2, 6);   // (the first weight fragments are in)

  for (int cc = 0; cc < p.cpw; ++cc) {
    const int chunk = grp * p.cpw + cc;
    const float bias = p.b1 ? p.b1[chunk * FC + wave * 16 + r] : 0.f;
#if SC_FFN_EARLY_B2
    // GEMM 2's weight fragments of this chunk are requested BEFORE GEMM 1: they travel behind its MFMAs.  Requested
    // after it (round 1-3) only the short epilogue lay between the request and their first use, and both waves of a
    // SIMD reach that point together (barrier): the matrix pipe idled for an L2 round trip per chunk.
    load_b2(chunk);
#endif
    // ---- GEMM 1: h[RT x 16] of this wave ----
    f32x4 acc1[RTT];
    f32x4 acc1c[WS ? RTT : 1];
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt) {
      acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (WS) acc1c[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (WS) {
      constexpr int NS = RTT * KI1;
      const _Float16 *ab = XsS + r * LDXS + 16 * kk;
      h16x8 ah = *reinterpret_cast<const h16x8 *>(ab), al = *reinterpret_cast<const h16x8 *>(ab + 8);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        h16x8 nh = ah, nl = al;
        if (st + 1 < NS) {
          const _Float16 *ap = ab + ((st + 1) % RTT) * 16 * LDXS + ((st + 1) / RTT) * 64;
          nh = *reinterpret_cast<const h16x8 *>(ap);
          nl = *reinterpret_cast<const h16x8 *>(ap + 8);
        }
        acc1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bf[ki][0], acc1[rt], 0, 0, 0);
        acc1c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bf[ki][1], acc1c[rt], 0, 0, 0);
        acc1c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bf[ki][0], acc1c[rt], 0, 0, 0);
        ah = nh;
        al = nl;
      }
    } else if constexpr (WH) {
      constexpr int NS = RTT * KI1;
      const _Float16 *ab = XsH + r * LDXH + 8 * kk;
      h16x8 a = *reinterpret_cast<const h16x8 *>(ab);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        h16x8 n = a;
        if (st + 1 < NS) n = *reinterpret_cast<const h16x8 *>(ab + ((st + 1) % RTT) * 16 * LDXH + ((st + 1) / RTT) * 32);
        acc1[rt] = ffn_mma_h(acc1[rt], a, bf[ki][0], bf[ki][1]);
        a = n;
      }
    } else {  // A operands are fetched one step ahead of the MFMAs that use them
      constexpr int NS = RTT * KI1;
      const float *ab = Xs + r * LDX + 8 * kk;
      float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        float4 n0 = a0, n1 = a1;
        if (st + 1 < NS) {
          const float *ap = ab + ((st + 1) % RTT) * 16 * LDX + ((st + 1) / RTT) * 32;
          n0 = *reinterpret_cast<const float4 *>(ap);
          n1 = *reinterpret_cast<const float4 *>(ap + 4);
        }
        acc1[rt] = ffn_mfma8(acc1[rt], a0, a1, bf[ki][0], bf[ki][1]);
        a0 = n0;
        a1 = n1;
      }
    }
#if !SC_FFN_EARLY_B2
    // GEMM 2 weights of this chunk: in flight during the epilogue
    load_b2(chunk);
#endif
    if (cc > 0) __syncthreads();  // previous chunk's GEMM 2 is done reading Hs
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float hv = fmaxf((WS ? acc1[rt][j] + acc1c[WS ? rt : 0][j] * (1.f / 2048.f) : acc1[rt][j]) + bias, 0.f);
        if (WS) {
          const _Float16 hi = (_Float16)hv;
          _Float16 *d = HsS + (rt * 16 + 4 * kk + j) * LDHS + split_at(wave * 16 + r);
          d[0] = hi;
          d[8] = (_Float16)((hv - (float)hi) * 2048.f);
        } else if (WH) HsH[(rt * 16 + 4 * kk + j) * LDHH + wave * 16 + r] = (_Float16)hv;
        else Hs[(rt * 16 + 4 * kk + j) * LDH + wave * 16 + r] = hv;
      }
    if (cc + 1 < p.cpw) load_b1(chunk + 1);  // next chunk's GEMM 1 weights, overlapped with GEMM 2
    __syncthreads();
    // ---- canonical summation order (bit-reproducible serving): every 128-wide hidden chunk is ONE accumulation chain that
    // starts from zero, the chunks' results are added as a balanced binary tree over the chunk index - so the result does not
    // depend on how many chunks a workgroup takes (cpw = 1: the consumer adds neighbours; cpw = 2: they are added here), i.e.
    // not on the row count that the tile shape is chosen by.  The first chunk's result is parked in the LDS row tile, which
    // is free from here on (every wave is past the last GEMM 1), and the second chunk's is added to it below.
    const bool tree_last = cc > 0 && cc + 1 == p.cpw;
    if (tree_last) {
#pragma unroll
      for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            Xs[(rt * 16 + 4 * kk + j) * LDX + (wave * NT2 + t) * 16 + r] =
                WS ? acc2[rt][t][j] + acc2c[WS ? rt : 0][WS ? t : 0][j] * (1.f / 2048.f) : acc2[rt][t][j];
          acc2[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (WS) acc2c[WS ? rt : 0][WS ? t : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- GEMM 2: partial y[RT x D/8] of this wave ----
    if constexpr (WS) {
      constexpr int NS = RTT * KI2;
      const _Float16 *ab = HsS + r * LDHS + 16 * kk;
      h16x8 ah = *reinterpret_cast<const h16x8 *>(ab), al = *reinterpret_cast<const h16x8 *>(ab + 8);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int k = st / RTT, rt = st % RTT;
        h16x8 nh = ah, nl = al;
        if (st + 1 < NS) {
          const _Float16 *ap = ab + ((st + 1) % RTT) * 16 * LDHS + ((st + 1) / RTT) * 64;
          nh = *reinterpret_cast<const h16x8 *>(ap);
          nl = *reinterpret_cast<const h16x8 *>(ap + 8);
        }
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
          acc2[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b2f[k][t][0], acc2[rt][t], 0, 0, 0);
          acc2c[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, b2f[k][t][1], acc2c[rt][t], 0, 0, 0);
          acc2c[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, b2f[k][t][0], acc2c[rt][t], 0, 0, 0);
        }
        ah = nh;
        al = nl;
      }
    } else if constexpr (WH) {
      constexpr int NS = RTT * KI2;
      const _Float16 *ab = HsH + r * LDHH + 8 * kk;
      h16x8 a = *reinterpret_cast<const h16x8 *>(ab);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int k = st / RTT, rt = st % RTT;
        h16x8 n = a;
        if (st + 1 < NS) n = *reinterpret_cast<const h16x8 *>(ab + ((st + 1) % RTT) * 16 * LDHH + ((st + 1) / RTT) * 32);
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[rt][t] = ffn_mma_h(acc2[rt][t], a, b2f[k][t][0], b2f[k][t][1]);
        a = n;
      }
    } else {
      constexpr int NS = RTT * KI2;
      const float *ab = Hs + r * LDH + 8 * kk;
      float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int k = st / RTT, rt = st % RTT;
        float4 n0 = a0, n1 = a1;
        if (st + 1 < NS) {
          const float *ap = ab + ((st + 1) % RTT) * 16 * LDH + ((st + 1) / RTT) * 32;
          n0 = *reinterpret_cast<const float4 *>(ap);
          n1 = *reinterpret_cast<const float4 *>(ap + 4);
        }
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[rt][t] = ffn_mfma8(acc2[rt][t], a0, a1, b2f[k][t][0], b2f[k][t][1]);
        a0 = n0;
        a1 = n1;
      }
    }
  }
  // ---- partial result -> LDS (row-major) -> workspace, full lines ----
  // (Xs is free: its last readers finished before the barrier that preceded the last GEMM 2)
#pragma unroll
  for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float *xp = Xs + (rt * 16 + 4 * kk + j) * LDX + (wave * NT2 + t) * 16 + r;
        const float v = WS ? acc2[rt][t][j] + acc2c[WS ? rt : 0][WS ? t : 0][j] * (1.f / 2048.f) : acc2[rt][t][j];
        *xp = p.cpw > 1 ? *xp + v : v;   // (cpw > 1: the element this lane parked above + the last chunk's chain)
      }
  __syncthreads();
  SC_STAMP(PRO ? 2 : 3, PRO ? 7 : 2);
  float *dst = p.part + ((long)grp * p.M + m0) * D;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = threadIdx.x + q * 512;
    const int i = e / (D / 4), c4 = e % (D / 4);
    if (m0 + i < p.M) {
      const float4 v = *reinterpret_cast<const float4 *>(Xs + i * LDX + 4 * c4);
      if (PRO && p.act_half) {
        *reinterpret_cast<h16x4 *>(reinterpret_cast<_Float16 *>(p.part) + ((long)grp * p.part_rows + rowid[i]) * D + 4 * c4) =
            h16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
      } else {
        float *o = PRO ? p.part + ((long)grp * p.part_rows + rowid[i]) * D : dst + (long)i * D;
        // (round 6) the prologue-less form behind the stream-resident decoder layer (sc_dec_layer_ffn_xn): split sums by ROW ID too
        if (!PRO && p.part_rows > 0) o = p.part + ((long)grp * p.part_rows + (p.rows ? p.rows[m0 + i] : m0 + i)) * D;
        *reinterpret_cast<float4 *>(o + 4 * c4) = v;
      }
    }
  }
  SC_STAMP_END(PRO ? 2 : 3, PRO ? 8 : 3);
}
SC_PHASE_GETTER(sc_phase_debug_ffn)

template <int D, int RTT, bool PRO, int WH>
static void launch_ffn_wh(const FfnArgs &p, int ngrp, hipStream_t st) {
  constexpr int RT = 16 * RTT;
  const size_t lds = ffn_lds_bytes(D, RT, PRO, WH);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_fused_kernel<D, RTT, PRO, WH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  ffn_fused_kernel<D, RTT, PRO, WH><<<dim3(ngrp, cdiv(p.M, RT)), 512, lds, st>>>(p);
}
template <int D, int RTT, bool PRO = false>
static void launch_ffn(const FfnArgs &p, int ngrp, hipStream_t st) {
  if (p.w_form == 2) launch_ffn_wh<D, RTT, PRO, 2>(p, ngrp, st);
  else if (p.w_form) launch_ffn_wh<D, RTT, PRO, 1>(p, ngrp, st);
  else launch_ffn_wh<D, RTT, PRO, 0>(p, ngrp, st);
}

template <int D, bool PRO = false>
static void launch_ffn_rtt(const FfnArgs &p, int rtt, int ngrp, hipStream_t st) {
  switch (rtt) {
    case 1: launch_ffn<D, 1, PRO>(p, ngrp, st); break;
    case 2: launch_ffn<D, 2, PRO>(p, ngrp, st); break;
    case 3: launch_ffn<D, 3, PRO>(p, ngrp, st); break;
    case 4: launch_ffn<D, 4, PRO>(p, ngrp, st); break;
    default: launch_ffn<D, 5, PRO>(p, ngrp, st); break;
  }
}

// ===========================================================================
// Row-tile projection:  C = [LN](A) . W^T + b  [+ R]  [-> LN2]      (K = D)
// The attention projections of an encoder layer (multi_head_attention.py:63-90
// q/k/v Linear behind norm1, and the output Linear + residual followed by
// norm2, contextual_block_encoder_layer.py:178-271) on the machinery of the
// fused FFN's first GEMM: a workgroup owns RT = 16*RTT complete input rows
// (K = D, so the LayerNorm in front of the projection is a prologue on the LDS
// tile) and streams fragment-packed weights (sc_pack_panel_weight) straight
// into registers.
//   FULL = false: N a multiple of 128; grid.x = N/128/cpw groups of column
//                 chunks, result staged per chunk through LDS -> full lines;
//   FULL = true : N == D; the workgroup produces whole output rows, adds the
//                 residual and (optionally) LayerNorms them into LN2.
// Algorithmic work per row: 2*D*N flop; bytes per row 4*(D + N) (+ 2*4*D FULL).
// ===========================================================================
struct RowProjArgs {
  const float *A;
  int lda;
  const float *ln_g, *ln_b;  // LayerNorm of the input rows (nullptr: none)
  const float *Wp, *bias;
  const float *R;  // FULL: residual rows [M][ldc] (may be C itself)
  float *C;
  int ldc;
  const float *g2, *b2;  // FULL: LayerNorm of the result rows -> LN2 [M][D]
  float *LN2;
  float eps;
  int M, N, cpw;
  int w_form;   // 0: fp32 weights; 1: Wp holds fp16 elements (same fragment order): fp16 MFMA inputs, fp32 accumulation;
                // 2: Wp holds the fp16 hi | lo split of the fp32 weights (ffn_fused_kernel, WF = 2)
};

template <int D>
__device__ __forceinline__ float4 rowtile_ln(const float4 v, bool act, const float *g, const float *b, float eps,
                                             int lane) {
  float s = act ? (v.x + v.y) + (v.z + v.w) : 0.f;
  const float mean = wave_sum(s) / (float)D;
  const float a = v.x - mean, bb = v.y - mean, c = v.z - mean, e = v.w - mean;
  const float q = act ? (a * a + bb * bb) + (c * c + e * e) : 0.f;
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float4 o = v;
  if (act) {
    const float4 gm = reinterpret_cast<const float4 *>(g)[lane];
    const float4 bt = reinterpret_cast<const float4 *>(b)[lane];
    o = make_float4(a * rstd * gm.x + bt.x, bb * rstd * gm.y + bt.y, c * rstd * gm.z + bt.z, e * rstd * gm.w + bt.w);
  }
  return o;
}

// WH: fp16 weights (the fragment order of sc_pack_panel_weight with 2-byte elements) and fp16 MFMA inputs
// (v_mfma_f32_16x16x16_f16, as in ffn_fused_kernel<.., WH>): the (normalised) row tile is converted to fp16 once, after
// the LayerNorm prologue; accumulation, bias, residual and the output LayerNorm stay fp32.
// WF = 2 (WS): the fp16 hi | lo split of both operands, three v_mfma_f32_16x16x32_f16 per product sum - fp32-grade
// results on the fp16 matrix pipe, see ffn_fused_kernel.  The row tile is split IN PLACE (hi8 | lo8 per 8 k values:
// the bytes of the fp32 row) by the wave that normalises the row.
template <int D, int RTT, bool FULL, int WF = 0>
__global__ __launch_bounds__(512) void rowtile_proj_kernel(RowProjArgs p) {
  constexpr bool WH = WF == 1, WS = WF == 2;
  constexpr int RT = 16 * RTT, FC = 128, KI1 = D / 32;
  constexpr int LDXS = 2 * D + 8;       // WS: row stride of the hi8 | lo8 tile (fp16 elements)
  constexpr int LDX = D + 4, LDO = FULL ? D + 4 : FC + 4;
  constexpr int LDXH = D + 8;           // WH: row stride of the fp16 tile (elements)
  extern __shared__ __attribute__((aligned(16))) float rowtile_smem[];
  float *Xs = rowtile_smem;             // [RT][LDX] (normalised) input rows
  float *Os = rowtile_smem + RT * LDX;  // [RT][LDO] result staging
  _Float16 *XsH = reinterpret_cast<_Float16 *>(rowtile_smem + RT * LDX + RT * LDO);   // WH: [RT][LDXH]
  _Float16 *XsS = reinterpret_cast<_Float16 *>(rowtile_smem);                          // WS: [RT][LDXS], in place of Xs
  typedef typename std::conditional<WS, h16x8, typename std::conditional<WH, h16x4, float4>::type>::type BF;   // 4 weight elements
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const int grp = blockIdx.x, m0 = blockIdx.y * RT;
  const int nch = FULL ? D / FC : p.cpw, ch0 = FULL ? 0 : grp * p.cpw;

  BF bf[KI1][2];
  auto load_b = [&](int chunk) {
    const BF *wp = reinterpret_cast<const BF *>(p.Wp) + ((long)(chunk * 8 + wave) * KI1) * 128 + lane;
#pragma unroll
    for (int ki = 0; ki < KI1; ++ki) {
      bf[ki][0] = wp[ki * 128];
      bf[ki][1] = wp[ki * 128 + 64];
    }
  };
  load_b(ch0);
  constexpr int NV = RT * D / 4, NQ = NV / 512;
  static_assert(NV % 512 == 0, "tile must be a multiple of the workgroup");
  {
    float4 stage[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * 512;
      const int m = min(m0 + e / (D / 4), p.M - 1);
      stage[q] = *reinterpret_cast<const float4 *>(p.A + (long)m * p.lda + 4 * (e % (D / 4)));
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * 512;
      *reinterpret_cast<float4 *>(Xs + (e / (D / 4)) * LDX + 4 * (e % (D / 4))) = stage[q];
    }
  }
  __syncthreads();
  if (p.ln_g || WS) {  // wave w normalises (WS: and splits) rows w, w+8, ... in place
    const bool act = lane < D / 4;
    for (int i = wave; i < RT; i += 8) {
      float4 *row = reinterpret_cast<float4 *>(Xs + i * LDX);
      const float4 v = act ? row[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 o = p.ln_g ? rowtile_ln<D>(v, act, p.ln_g, p.ln_b, p.eps, lane) : v;
      if (WS) {   // every lane of the wave has read its piece of the row
        h16x4 hi, lo;
        ffn_split4(o, hi, lo);
        _Float16 *d = XsS + i * LDXS + ((4 * lane) >> 3) * 16 + ((4 * lane) & 7);
        if (act) {
          *reinterpret_cast<h16x4 *>(d) = hi;
          *reinterpret_cast<h16x4 *>(d + 8) = lo;
        }
      } else if (act) row[lane] = o;
    }
    __syncthreads();
  }
  if constexpr (WH) {   // fp16 copy of the (normalised) tile: the A operands of the fp16 MFMAs
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int e = threadIdx.x + q * 512;
      const float4 v = *reinterpret_cast<const float4 *>(Xs + (e / (D / 4)) * LDX + 4 * (e % (D / 4)));
      *reinterpret_cast<h16x4 *>(XsH + (e / (D / 4)) * LDXH + 4 * (e % (D / 4))) =
          h16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
    __syncthreads();
  }
  for (int cc = 0; cc < nch; ++cc) {
    const int chunk = ch0 + cc;
    const float bias = p.bias ? p.bias[chunk * FC + wave * 16 + r] : 0.f;
    f32x4 acc[RTT];
    f32x4 accc[WS ? RTT : 1];   // WS: the 2^11-scaled cross terms
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt) {
      acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (WS) accc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (WS) {
      constexpr int NS = RTT * KI1;
      const _Float16 *ab = XsS + r * LDXS + 16 * kk;
      h16x8 ah = *reinterpret_cast<const h16x8 *>(ab), al = *reinterpret_cast<const h16x8 *>(ab + 8);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        h16x8 nh = ah, nl = al;
        if (st + 1 < NS) {
          const _Float16 *ap = ab + ((st + 1) % RTT) * 16 * LDXS + ((st + 1) / RTT) * 64;
          nh = *reinterpret_cast<const h16x8 *>(ap);
          nl = *reinterpret_cast<const h16x8 *>(ap + 8);
        }
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bf[ki][0], acc[rt], 0, 0, 0);
        accc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bf[ki][1], accc[rt], 0, 0, 0);
        accc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bf[ki][0], accc[rt], 0, 0, 0);
        ah = nh;
        al = nl;
      }
    } else if constexpr (WH) {
      constexpr int NS = RTT * KI1;
      const _Float16 *ab = XsH + r * LDXH + 8 * kk;
      h16x8 a = *reinterpret_cast<const h16x8 *>(ab);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        h16x8 n = a;
        if (st + 1 < NS) n = *reinterpret_cast<const h16x8 *>(ab + ((st + 1) % RTT) * 16 * LDXH + ((st + 1) / RTT) * 32);
        acc[rt] = ffn_mma_h(acc[rt], a, bf[ki][0], bf[ki][1]);
        a = n;
      }
    } else {
      constexpr int NS = RTT * KI1;
      const float *ab = Xs + r * LDX + 8 * kk;
      float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const int ki = st / RTT, rt = st % RTT;
        float4 n0 = a0, n1 = a1;
        if (st + 1 < NS) {
          const float *ap = ab + ((st + 1) % RTT) * 16 * LDX + ((st + 1) / RTT) * 32;
          n0 = *reinterpret_cast<const float4 *>(ap);
          n1 = *reinterpret_cast<const float4 *>(ap + 4);
        }
        acc[rt] = ffn_mfma8(acc[rt], a0, a1, bf[ki][0], bf[ki][1]);
        a0 = n0;
        a1 = n1;
      }
    }
    if (cc + 1 < nch) load_b(chunk + 1);  // next chunk's weights, in flight during the staging
    if (!FULL && cc > 0) __syncthreads();  // the previous chunk's stores out of Os are done
    const int ocol = (FULL ? chunk * FC : 0) + wave * 16 + r;
#pragma unroll
    for (int rt = 0; rt < RTT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        Os[(rt * 16 + 4 * kk + j) * LDO + ocol] = (WS ? acc[rt][j] + accc[WS ? rt : 0][j] * (1.f / 2048.f) : acc[rt][j]) + bias;
    if (!FULL) {
      __syncthreads();
      for (int e = threadIdx.x; e < RT * (FC / 4); e += 512) {
        const int i = e / (FC / 4), c4 = e % (FC / 4);
        if (m0 + i < p.M)
          *reinterpret_cast<float4 *>(p.C + (long)(m0 + i) * p.ldc + chunk * FC + 4 * c4) =
              *reinterpret_cast<const float4 *>(Os + i * LDO + 4 * c4);
      }
    }
  }
  if (FULL) {
    __syncthreads();
    const bool act = lane < D / 4;
    for (int i = wave; i < RT; i += 8) {
      const int m = m0 + i;
      if (m >= p.M) break;  // uniform per wave
      float4 v = act ? reinterpret_cast<const float4 *>(Os + i * LDO)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (act) {
        if (p.R) {
          const float4 rr = reinterpret_cast<const float4 *>(p.R + (long)m * p.ldc)[lane];
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        reinterpret_cast<float4 *>(p.C + (long)m * p.ldc)[lane] = v;
      }
      if (p.LN2) {
        const float4 o = rowtile_ln<D>(v, act, p.g2, p.b2, p.eps, lane);
        if (act) reinterpret_cast<float4 *>(p.LN2 + (long)m * D)[lane] = o;
      }
    }
  }
}

template <int D, int RTT, bool FULL, int WH>
static void launch_rowtile_wh(const RowProjArgs &p, int ngrp, hipStream_t st) {
  constexpr int RT = 16 * RTT;
  const size_t lds = (size_t)(RT * (D + 4) + RT * ((FULL ? D : 128) + 4)) * sizeof(float) + (WH == 1 ? (size_t)RT * (D + 8) * 2 : 0);
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rowtile_proj_kernel<D, RTT, FULL, WH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  rowtile_proj_kernel<D, RTT, FULL, WH><<<dim3(ngrp, cdiv(p.M, RT)), 512, lds, st>>>(p);
}
template <int D, int RTT, bool FULL>
static void launch_rowtile(const RowProjArgs &p, int ngrp, hipStream_t st) {
  if (p.w_form == 2) launch_rowtile_wh<D, RTT, FULL, 2>(p, ngrp, st);
  else if (p.w_form) launch_rowtile_wh<D, RTT, FULL, 1>(p, ngrp, st);
  else launch_rowtile_wh<D, RTT, FULL, 0>(p, ngrp, st);
}

template <int D, bool FULL>
static void launch_rowtile_rtt(const RowProjArgs &p, int rtt, int ngrp, hipStream_t st) {
  switch (rtt) {
    case 1: launch_rowtile<D, 1, FULL>(p, ngrp, st); break;
    case 2: launch_rowtile<D, 2, FULL>(p, ngrp, st); break;
    case 3: launch_rowtile<D, 3, FULL>(p, ngrp, st); break;
    default: launch_rowtile<D, 4, FULL>(p, ngrp, st); break;
  }
}

extern "C" int sc_rowtile_proj_supported(int D, int N) {
  return (D == 256 || D == 128) && N % 128 == 0 && N >= 128;
}

static int rowtile_run(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                       float eps, const float *Wp, const float *bias, int N, const float *R, float *C,
                       int ldc, const float *g2, const float *b2, float *LN2, void *stream, int w_form) {
  SC_CHECK_ARG(A && Wp && C, "null pointer");
  SC_CHECK_ARG(sc_rowtile_proj_supported(D, N), "unsupported dimensions");
  SC_CHECK_ARG((!ln_g) == (!ln_b) && (!LN2 || (g2 && b2)), "LayerNorm parameters missing");
  const bool full = R || LN2;
  SC_CHECK_ARG(!full || N == D, "residual / output LayerNorm need N == D");
  SC_CHECK_ARG(lda % 4 == 0 && ldc % 4 == 0 && lda >= D && ldc >= N, "leading dimensions");
  SC_CHECK_ARG((((uintptr_t)A | (uintptr_t)Wp | (uintptr_t)C | (uintptr_t)R | (uintptr_t)LN2) & 15) == 0,
               "16-byte alignment");
  if (M <= 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nch = N / 128;
  // tile height (16*rtt rows) and column chunks per workgroup (cpw) by a cost model fitted to
  // tools/rowtile_sweep.py: a workgroup takes 4.3 + 1.0*rtt + 2.1*rtt*cpw us (+8 % for rtt = 4, whose 100 KB of LDS
  // leave it alone on its CU); tiles up to rtt = 3 fit two workgroups per CU, which then take 1.75x as long each
  int best_rtt = 1, best_cpw = full ? nch : 1;
  double best = 1e30;
  for (int cpw = full ? nch : 1; cpw <= nch; ++cpw) {
    if (nch % cpw) continue;
    for (int rtt = 1; rtt <= 4; ++rtt) {
      const long wgs = (long)(nch / cpw) * cdiv(M, 16 * rtt);
      const int conc = (rtt <= 3 && wgs > 256) ? 2 : 1;
      const double t_wg = (4.3 + 1.0 * rtt + 2.1 * rtt * cpw) * (rtt == 4 ? 1.08 : 1.0);
      const double t = (double)((wgs + 256 * conc - 1) / (256 * conc)) * t_wg * (conc == 2 ? 1.75 : 1.0);
      if (t < best) { best = t; best_rtt = rtt; best_cpw = cpw; }
    }
  }
  if (const char *f = sc_hook("SC_ROWTILE_FORCE")) {   // tools/rowtile_bench.py sweep: "rtt,cpw"
    int r = 0, c = 0;
    if (sscanf(f, "%d,%d", &r, &c) == 2 && r >= 1 && r <= 4 && c >= 1 && nch % c == 0) {
      best_rtt = r;
      if (!full) best_cpw = c;
    }
  }
  RowProjArgs p{A, lda, ln_g, ln_b, Wp, bias, R, C, ldc, g2, b2, LN2, eps, M, N, best_cpw, w_form};
  ProfScope prof = sc_prof_begin(st);
  const int ngrp = full ? 1 : nch / best_cpw;
  if (D == 256) {
    if (full) launch_rowtile_rtt<256, true>(p, best_rtt, ngrp, st);
    else launch_rowtile_rtt<256, false>(p, best_rtt, ngrp, st);
  } else {
    if (full) launch_rowtile_rtt<128, true>(p, best_rtt, ngrp, st);
    else launch_rowtile_rtt<128, false>(p, best_rtt, ngrp, st);
  }
  sc_prof_end(prof, SC_PROF_ROWTILE_PROJ, 2.0 * (double)M * D * N,
              4.0 * ((double)M * (D + N) + (double)D * N / (w_form == 1 ? 2.0 : 1.0) + (full ? 2.0 * M * D : 0.0)));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

extern "C" int sc_rowtile_proj(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                               float eps, const float *Wp, const float *bias, int N, const float *R, float *C,
                               int ldc, const float *g2, const float *b2, float *LN2, void *stream) {
  return rowtile_run(A, lda, M, D, ln_g, ln_b, eps, Wp, bias, N, R, C, ldc, g2, b2, LN2, stream, 0);
}
// ... with fp16 weights: Wh = the fragment-packed copy with 2-byte elements; fp16 MFMA inputs (the normalised row tile is
// rounded to fp16 when staged), fp32 accumulation, bias, residual and output LayerNorm
extern "C" int sc_rowtile_proj_h(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                                 float eps, const void *Wh, const float *bias, int N, const float *R, float *C,
                                 int ldc, const float *g2, const float *b2, float *LN2, void *stream) {
  return rowtile_run(A, lda, M, D, ln_g, ln_b, eps, reinterpret_cast<const float *>(Wh), bias, N, R, C, ldc, g2, b2, LN2,
                     stream, 1);
}

// ... with the fp16 hi | lo split of the fp32 weights (weights.py split_panel_weight): fp32-grade results from three fp16
// MFMAs per product sum (rowtile_proj_kernel, WF = 2)
extern "C" int sc_rowtile_proj_s(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                                 float eps, const void *Ws, const float *bias, int N, const float *R, float *C,
                                 int ldc, const float *g2, const float *b2, float *LN2, void *stream) {
  return rowtile_run(A, lda, M, D, ln_g, ln_b, eps, reinterpret_cast<const float *>(Ws), bias, N, R, C, ldc, g2, b2, LN2,
                     stream, 2);
}


// tallest row tile (16 * rtt rows) per weight form: the split form carries two sets of accumulators and spills
// beyond 48 rows at D = 256 (tools/kernel_resources.py)
static inline int ffn_rtt_max(int D, int w_form) { return (w_form == 2 && D == 256) ? 3 : 5; }

extern "C" int sc_ffn_ln_supported(int D, int F) { return (D == 256 || D == 128) && F % 128 == 0 && F >= 128; }

// Wq != nullptr: the partial sums are reduced by the reduce + LayerNorm + projection row-panel
// kernel (decoder_panel.hip) which writes x to Xout (!= X) and the projection to Q [.., N]
static int ffn_run(const float *XN, const int32_t *rows, int M, int D, int F, const float *W1p,
                   const float *b1, const float *W2p, const float *b2, float *X, const float *ln_g,
                   const float *ln_b, float ln_eps, float *ln_out, float *Xout, const float *Wq,
                   const float *bq, float *Q, int N, void *stream, int w_form = 0,   // FfnArgs.w_form
                   const ScHandoff *ho = nullptr) {
  SC_CHECK_ARG(XN && W1p && W2p && X, "null pointer");
  SC_CHECK_ARG(sc_ffn_ln_supported(D, F), "unsupported dimensions");
  SC_CHECK_ARG(!ln_out || (ln_g && ln_b), "LayerNorm parameters missing");
  SC_CHECK_ARG((((uintptr_t)XN | (uintptr_t)W1p | (uintptr_t)W2p | (uintptr_t)X) & 15) == 0, "16-byte alignment");
  if (M <= 0) return SC_OK;
  hipStream_t st = (hipStream_t)stream;
  resolve_workspace(stream);
  SC_CHECK_ARG(g_ws && g_ws_bytes >= (size_t)16 * D * sizeof(float), "sc_ffn_ln needs the split-K workspace");
  const int nch = F / 128;
  const long ws_rows_per_part = (long)(g_ws_bytes / sizeof(float) / D);   // rows*parts that fit
  // rows in slabs so that the partials fit the workspace; per slab pick the
  // tile height (16*rtt rows) and chunks per workgroup (cpw) with the fewest
  // rounds of 256 workgroups, ties towards more partial groups (= less serial work)
  int m_done = 0;
  while (m_done < M) {
    int best_rtt = 5, best_cpw = nch;
    long slab = M - m_done;
    double best = 1e30;
    for (int cpw = 1; cpw <= 2; cpw *= 2) {   // (the canonical summation order covers one chunk or an aligned pair per workgroup)
      if (nch % cpw) continue;
      const int ngrp = nch / cpw;
      long fit = ws_rows_per_part / ngrp;
      if (fit < 16) continue;
      const long mm = fit < (M - m_done) ? (fit / 80) * 80 : (M - m_done);
      if (mm <= 0) continue;
      for (int rtt = 1; rtt <= ffn_rtt_max(D, w_form); ++rtt) {
        const long wgs = (long)ngrp * ((mm + 16 * rtt - 1) / (16 * rtt));
        const double rounds = (double)((wgs + 255) / 256);
        // per workgroup: fixed ~2.5 us + 4.0 us per 16 rows and chunk (measured); reduce: bytes of the partials
        double t = rounds * (2.5 + 4.0 * rtt * cpw) + 1.5 + 2.0 * (double)mm * D * 4.0 * ngrp / 3.0e6;   // fitted: tools/ffn_sweep.py
        t *= (double)(M - m_done) / (double)mm;   // slabs needed at this size
        if (t < best) { best = t; best_rtt = rtt; best_cpw = cpw; slab = mm; }
      }
    }
    SC_CHECK_ARG(best < 1e29, "workspace too small for sc_ffn_ln");
    SC_CHECK_ARG(!Wq || slab == M, "workspace too small for the fused projection (rows do not fit one slab)");
    SC_CHECK_ARG(!ho || (slab == M && !rows && !ln_out && !Wq), "context hand-off: all rows in one slab, no row table / LayerNorm");
    if (const char *f = sc_hook("SC_FFN_FORCE")) {   // tools/ffn_sweep.py: "rtt,cpw"
      int r = 0, c = 0;
      if (sscanf(f, "%d,%d", &r, &c) == 2 && r >= 1 && r <= 5 && c >= 1 && c <= 2 && nch % c == 0 &&
          (long)(M - m_done) * (nch / c) <= ws_rows_per_part) {
        best_rtt = r;
        best_cpw = c;
        slab = M - m_done;
      }
    }
    const int ngrp = nch / best_cpw;
    FfnArgs p{XN, rows ? rows + m_done : nullptr, W1p, b1, W2p, g_ws, (int)slab, F, best_cpw};
    p.w_form = w_form;
    p.dbg_stamp = sc_phase_take(3);
    // without a row table the slab is addressed by offsetting the base pointers
    const float *xn_base = rows ? XN : XN + (long)m_done * D;
    p.XN = xn_base;
    ProfScope prof = sc_prof_begin(st);
    if (D == 256) launch_ffn_rtt<256>(p, best_rtt, ngrp, st);
    else launch_ffn_rtt<128>(p, best_rtt, ngrp, st);
    // algorithmic: 4*D*F flop per row; W1 + W2 once, x in, x out.  The split sums this decomposition
    // writes (ngrp x slab x D) and the reduce kernel re-reads are traffic, not algorithmic bytes
    sc_prof_end(prof, SC_PROF_FFN_FUSED, 4.0 * (double)slab * D * F,
                4.0 * (2.0 * (double)slab * D + 2.0 * (double)D * F));
    SC_CHECK_LAUNCH();
    GemmArgs g{nullptr, nullptr, D, nullptr, b2, rows ? X : X + (long)m_done * D, rows ? rows + m_done : nullptr, D,
               (int)slab, D, F, SC_GEMM_RESIDUAL | (rows ? SC_GEMM_LN_AT_CROWS : 0), 0, g_ws, 0};
    if (Wq) {
      int rc = sc_launch_reduce_ln_proj(g_ws, ngrp, (int)slab, b2, X, Xout, rows, M, D, ln_g, ln_b, ln_eps, ln_out,
                                        Wq, bq, Q, N, st);
      if (rc != SC_OK) return rc;
    } else if (ho) {
      const long n4 = (long)slab * (D / 4);
      ffn_reduce_handoff_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(g, ngrp, *ho);
    } else if (ln_out) {
      float *lo = rows ? ln_out : ln_out + (long)m_done * D;
      gemm_splitk_reduce_ln_kernel<<<cdiv((int)slab, 4), 256, 0, st>>>(g, ngrp, ln_g, ln_b, ln_eps, lo, D);
    } else {
      const long n4 = (long)slab * (D / 4);
      gemm_splitk_reduce_kernel<<<dim3((unsigned)((n4 + 255) / 256)), 256, 0, st>>>(g, ngrp);
    }
    SC_CHECK_LAUNCH();
    m_done += (int)slab;
  }
  return SC_OK;
}

extern "C" int sc_ffn_ln(const float *XN, const int32_t *rows, int M, int D, int F, const float *W1p,
                         const float *b1, const float *W2p, const float *b2, float *X, const float *ln_g,
                         const float *ln_b, float ln_eps, float *ln_out, void *stream) {
  return ffn_run(XN, rows, M, D, F, W1p, b1, W2p, b2, X, ln_g, ln_b, ln_eps, ln_out, nullptr, nullptr, nullptr,
                 nullptr, 0, stream);
}

int sc_ffn_ln_handoff(const float *XN, int M, int D, int F, const void *W1, const float *b1, const void *W2, const float *b2,
                      float *X, int w_form, const ScHandoff &ho, void *stream) {
  SC_CHECK_ARG(ho.blkinfo && ho.state && ho.R > 1 && M % ho.R == 0, "hand-off table");
  return ffn_run(XN, nullptr, M, D, F, (const float *)W1, b1, (const float *)W2, b2, X, nullptr, nullptr, 0.f, nullptr, nullptr,
                 nullptr, nullptr, nullptr, 0, stream, w_form, &ho);
}

// fp16 weights (fragment-packed like W1p / W2p, 2-byte elements): fp16 MFMA inputs, fp32 accumulation and output
extern "C" int sc_ffn_ln_h(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1h,
                           const float *b1, const void *W2h, const float *b2, float *X, const float *ln_g,
                           const float *ln_b, float ln_eps, float *ln_out, void *stream) {
  return ffn_run(XN, rows, M, D, F, (const float *)W1h, b1, (const float *)W2h, b2, X, ln_g, ln_b, ln_eps, ln_out, nullptr,
                 nullptr, nullptr, nullptr, 0, stream, 1);
}
// fp16 hi | lo split of the fp32 weights (weights.py split_panel_weight): fp32-grade results from three fp16 MFMAs per
// product sum (ffn_fused_kernel, WF = 2)
extern "C" int sc_ffn_ln_s(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1s,
                           const float *b1, const void *W2s, const float *b2, float *X, const float *ln_g,
                           const float *ln_b, float ln_eps, float *ln_out, void *stream) {
  return ffn_run(XN, rows, M, D, F, (const float *)W1s, b1, (const float *)W2s, b2, X, ln_g, ln_b, ln_eps, ln_out, nullptr,
                 nullptr, nullptr, nullptr, 0, stream, 2);
}
extern "C" int sc_ffn_ln_proj_s(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1s,
                                const float *b1, const void *W2s, const float *b2, const float *Xin, float *Xout,
                                const float *ln_g, const float *ln_b, float ln_eps, float *ln_out, const float *Wq,
                                const float *bq, float *Q, int N, void *stream) {
  SC_CHECK_ARG(Xin && Xout && Xin != Xout && Wq && Q && ln_g && ln_b, "null / aliased operand");
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D) && N > 0 && N % D == 0, "projection width must be a multiple of D");
  return ffn_run(XN, rows, M, D, F, (const float *)W1s, b1, (const float *)W2s, b2, const_cast<float *>(Xin), ln_g, ln_b,
                 ln_eps, ln_out, Xout, Wq, bq, Q, N, stream, 2);
}
extern "C" int sc_ffn_ln_proj_h(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1h,
                                const float *b1, const void *W2h, const float *b2, const float *Xin, float *Xout,
                                const float *ln_g, const float *ln_b, float ln_eps, float *ln_out, const float *Wq,
                                const float *bq, float *Q, int N, void *stream) {
  SC_CHECK_ARG(Xin && Xout && Xin != Xout && Wq && Q && ln_g && ln_b, "null / aliased operand");
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D) && N > 0 && N % D == 0, "projection width must be a multiple of D");
  return ffn_run(XN, rows, M, D, F, (const float *)W1h, b1, (const float *)W2h, b2, const_cast<float *>(Xin), ln_g, ln_b,
                 ln_eps, ln_out, Xout, Wq, bq, Q, N, stream, 1);
}

extern "C" int sc_ffn_ln_proj(const float *XN, const int32_t *rows, int M, int D, int F, const float *W1p,
                              const float *b1, const float *W2p, const float *b2, const float *Xin, float *Xout,
                              const float *ln_g, const float *ln_b, float ln_eps, float *ln_out, const float *Wq,
                              const float *bq, float *Q, int N, void *stream) {
  SC_CHECK_ARG(Xin && Xout && Xin != Xout && Wq && Q && ln_g && ln_b, "null / aliased operand");
  SC_CHECK_ARG(sc_proj_ln_proj_supported(D) && N > 0 && N % D == 0, "projection width must be a multiple of D");
  return ffn_run(XN, rows, M, D, F, W1p, b1, W2p, b2, const_cast<float *>(Xin), ln_g, ln_b, ln_eps, ln_out, Xout, Wq,
                 bq, Q, N, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Decoder layer, launch C of the head-parallel form (decoder_layer.hip): the feed-forward of layer `layer` with the
// cross-attention output projection folded into its prologue -
//   x'' = xin + bo2 + sum_heads ph2[row][h];  xout = x'';  partial sums of W2 . relu(W1 . LayerNorm3(x'') + b1)
// per chunk group into ffn_part[grp][S*W][d] (by row id).  The consumer (sc_dec_layer_self of the next layer or
// sc_dec_output_logits) adds b2 and the residual.  *n_part (HOST) receives the number of chunk groups.
extern "C" int sc_dec_layer_ffn(const sc_search *sbp, int layer, const float *xin, float *xout, float *ffn_part,
                                int max_part, int *n_part, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xin && xout && xin != xout && ffn_part && n_part && sbp->ph2, "null / aliased");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  const int D = sb.d, F = sb.F;
  SC_CHECK_ARG(sc_ffn_ln_supported(D, F), "unsupported dimensions");
  const sc_dec_layer &w = sb.layers[layer];
  const int32_t *rows = sb.rowmap;
  const int M = rows ? sb.n_rows : sb.S * sb.W;
  SC_CHECK_ARG(M > 0 && M <= sb.S * sb.W && max_part >= 1, "n_rows / max_part out of range");
  const int nch = F / 128;
  // weight form: split copies (fp32-grade on the fp16 matrix pipe) before plain fp16 copies before fp32
  const int wf = (w.w1_s && w.w2_s) ? 2 : (w.w1_h && w.w2_h) ? 1 : 0;
  const void *w1x = wf == 2 ? w.w1_s : wf == 1 ? w.w1_h : (const void *)w.w1_p;
  const void *w2x = wf == 2 ? w.w2_s : wf == 1 ? w.w2_h : (const void *)w.w2_p;
  // tile height (16*rtt rows) and chunks per workgroup (cpw): fewest rounds of 256 workgroups, ties towards
  // more partial groups - the model fitted for sc_ffn_ln (tools/ffn_sweep.py)
  int best_rtt = 5, best_cpw = nch;
  double best = 1e30;
  for (int cpw = 1; cpw <= 2; cpw *= 2) {   // (canonical summation order: one chunk or an aligned pair per workgroup)
    if (nch % cpw || nch / cpw > max_part) continue;
    //  the consumers' batches of 8 are a balanced tree over the chunk index for up to 16 partials only: a model with
    // linear_units > 2048 always takes aligned pairs, so that the shape - and with it the order - never depends on the row count
    if (nch > 16 && nch % 2 == 0 && cpw != 2) continue;
    const int ngrp = nch / cpw;
    for (int rtt = 1; rtt <= ffn_rtt_max(D, wf); ++rtt) {
      const long wgs = (long)ngrp * ((M + 16 * rtt - 1) / (16 * rtt));
      const double rounds = (double)((wgs + 255) / 256);
      const double t = rounds * (2.5 + 4.0 * rtt * cpw) + 1.5 + 2.0 * (double)M * D * 4.0 * ngrp / 3.0e6;
      if (t < best) { best = t; best_rtt = rtt; best_cpw = cpw; }
    }
  }
  SC_CHECK_ARG(best < 1e29, "max_part too small");
  if (const char *f = sc_hook("SC_DEC_FFN_FORCE")) {   // A/B runs: "min_rows,rtt,cpw" for buckets of at least min_rows rows
    int mr = 0, r = 0, c = 0;
    if (sscanf(f, "%d,%d,%d", &mr, &r, &c) == 3 && M >= mr && r >= 1 && r <= ffn_rtt_max(D, wf) && c >= 1 && c <= 2 && nch % c == 0 &&
        nch / c <= max_part) {
      best_rtt = r;
      best_cpw = c;
    }
  }
  const int ngrp = nch / best_cpw;
  const int nph = sb.H / sc_dec_layer_hpw(sb);   // partial products per row left by sc_dec_layer_cross (decoder_layer.hip)
  FfnArgs p{nullptr, rows, (const float *)w1x, w.b1, (const float *)w2x, ffn_part, M, F,
            best_cpw, sb.ph2, nph, w.bo2, xin, xout, w.ln3_g, w.ln3_b, sb.ln_eps, sb.S * sb.W, wf, (sb.act_half & 2) ? 1 : 0,
            sc_phase_take(2), nph == sb.H ? 4 : 1};
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  if (D == 256) launch_ffn_rtt<256, true>(p, best_rtt, ngrp, st);
  else launch_ffn_rtt<128, true>(p, best_rtt, ngrp, st);
  // algorithmic: 4*D*F flop per row; x + H head partials read, W1 + W2 read once, x and the partials written
  // algorithmic: 4*D*F flop per row; weights once + x in + x out - the head partials read and the
  // split sums written are TRAFFIC of this decomposition, not algorithmic bytes
  sc_prof_end(prof, SC_PROF_FFN_PRO, 4.0 * (double)M * D * F, 4.0 * (2.0 * (double)M * D + 2.0 * (double)D * F));
  SC_CHECK_LAUNCH();
  *n_part = ngrp;
  return SC_OK;
}

// (round 6) launch C' of the stream-resident form (decoder_stream.hip): the same feed-forward WITHOUT a prologue - its input
// rows xn = LayerNorm3(x'') were written by sc_dec_layer_stream, which owns complete rows.  Split sums by row id into
// ffn_part[grp][S*W][d]; the consumer (sc_dec_layer_stream of the next layer or sc_dec_output_logits) adds b2 and the residual.
extern "C" int sc_dec_layer_ffn_xn(const sc_search *sbp, int layer, const float *xn, float *ffn_part, int max_part, int *n_part,
                                   void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xn && ffn_part && n_part, "null");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  const int D = sb.d, F = sb.F;
  SC_CHECK_ARG(sc_ffn_ln_supported(D, F), "unsupported dimensions");
  const sc_dec_layer &w = sb.layers[layer];
  const int32_t *rows = sb.rowmap;
  const int M = rows ? sb.n_rows : sb.S * sb.W;
  SC_CHECK_ARG(M > 0 && M <= sb.S * sb.W && max_part >= 1, "n_rows / max_part out of range");
  const int nch = F / 128;
  const int wf = (w.w1_s && w.w2_s) ? 2 : (w.w1_h && w.w2_h) ? 1 : 0;
  const void *w1x = wf == 2 ? w.w1_s : wf == 1 ? w.w1_h : (const void *)w.w1_p;
  const void *w2x = wf == 2 ? w.w2_s : wf == 1 ? w.w2_h : (const void *)w.w2_p;
  // tile height and chunks per workgroup: the attention launches of this form hold ONE compute unit per stream and the encoder
  // groups run beside them, so the grid is sized for the compute units the decode chain has to itself (SC_STREAM_FFN_CUS)
  int cus = SC_STREAM_FFN_CUS;
  if (const char *e = sc_hook("SC_STREAM_FFN_CUS")) cus = atoi(e);
  int best_rtt = 5, best_cpw = nch;
  double best = 1e30;
  for (int cpw = 1; cpw <= 2; cpw *= 2) {   // (canonical summation order: one chunk or an aligned pair per workgroup)
    if (nch % cpw || nch / cpw > max_part) continue;
    if (nch > 16 && nch % 2 == 0 && cpw != 2) continue;   // (see sc_dec_layer_ffn)
    const int ngrp = nch / cpw;
    for (int rtt = 1; rtt <= ffn_rtt_max(D, wf); ++rtt) {
      const long wgs = (long)ngrp * ((M + 16 * rtt - 1) / (16 * rtt));
      const double rounds = (double)((wgs + cus - 1) / cus);
      const double t = rounds * (2.5 + 4.0 * rtt * cpw) + 1.5 + 2.0 * (double)M * D * 4.0 * ngrp / 3.0e6;
      if (t < best) { best = t; best_rtt = rtt; best_cpw = cpw; }
    }
  }
  SC_CHECK_ARG(best < 1e29, "max_part too small");
  if (const char *f = sc_hook("SC_DEC_FFN_FORCE")) {   // A/B runs: "min_rows,rtt,cpw"
    int mr = 0, r = 0, c = 0;
    if (sscanf(f, "%d,%d,%d", &mr, &r, &c) == 3 && M >= mr && r >= 1 && r <= ffn_rtt_max(D, wf) && c >= 1 && c <= 2 && nch % c == 0 &&
        nch / c <= max_part) {
      best_rtt = r;
      best_cpw = c;
    }
  }
  const int ngrp = nch / best_cpw;
  FfnArgs p{xn, rows, (const float *)w1x, w.b1, (const float *)w2x, ffn_part, M, F, best_cpw};
  p.part_rows = sb.S * sb.W;
  p.w_form = wf;
  p.dbg_stamp = sc_phase_take(3);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  if (D == 256) launch_ffn_rtt<256, false>(p, best_rtt, ngrp, st);
  else launch_ffn_rtt<128, false>(p, best_rtt, ngrp, st);
  sc_prof_end(prof, SC_PROF_FFN_PRO, 4.0 * (double)M * D * F, 4.0 * (2.0 * (double)M * D + 2.0 * (double)D * F));
  SC_CHECK_LAUNCH();
  *n_part = ngrp;
  return SC_OK;
}
