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
#include "common.h"
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

__device__ __forceinline__ long gemm_kofs(const GemmArgs &g, int k0) {
  if (g.conv_f1 > 0) {
    int tap = k0 / g.lda;
    return (long)((tap / 3) * g.conv_f1 + tap % 3) * g.lda + (k0 % g.lda);
  }
  return k0;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs g) {
  constexpr int BK = 32;
  constexpr int TM = BM / (32 * WAVES_M), TN = BN / (32 * WAVES_N);
  constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
  constexpr int AI = BM / 32, BI = BN / 32;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  __shared__ float As[BK * LDA_S];
  __shared__ float Bs[BK * LDB_S];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kq = tid & 7, lr = tid >> 3;

  // Loads are UNCONDITIONAL (rows clamped into range): out-of-range rows and
  // columns compute garbage that the epilogue never stores.  Only the "-1 =
  // zero row" gather entries need masking, done with a select on the VALUE so
  // that hipcc does not branch around each load (which costs a vmcnt(0) wait
  // per load and serialises the whole K loop).
  long abase[AI];
  float amask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + lr + 32 * i;
    m = m < g.M ? m : g.M - 1;
    int row = g.a_rows ? g.a_rows[m] : m;
    amask[i] = row < 0 ? 0.f : 1.f;
    abase[i] = (long)(row < 0 ? 0 : row) * g.lda;
  }
  long wbase[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + lr + 32 * i;
    wbase[i] = (long)(n < g.N ? n : g.N - 1) * g.K;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[AI], rb[BI];

  {
    long ko = gemm_kofs(g, 0);
#pragma unroll
    for (int i = 0; i < AI; ++i)
      ra[i] = *reinterpret_cast<const float4 *>(g.A + abase[i] + ko + kq * 4);
#pragma unroll
    for (int i = 0; i < BI; ++i)
      rb[i] = *reinterpret_cast<const float4 *>(g.W + wbase[i] + kq * 4);
  }

  for (int k0 = 0; k0 < g.K; k0 += BK) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      float *p = As + (kq * 4) * LDA_S + lr + 32 * i;
      p[0] = ra[i].x * amask[i];
      p[LDA_S] = ra[i].y * amask[i];
      p[2 * LDA_S] = ra[i].z * amask[i];
      p[3 * LDA_S] = ra[i].w * amask[i];
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      float *p = Bs + (kq * 4) * LDB_S + lr + 32 * i;
      p[0] = rb[i].x;
      p[LDB_S] = rb[i].y;
      p[2 * LDB_S] = rb[i].z;
      p[3 * LDB_S] = rb[i].w;
    }
    __syncthreads();
    if (k0 + BK < g.K) {  // prefetch the next K tile into registers
      long ko = gemm_kofs(g, k0 + BK);
#pragma unroll
      for (int i = 0; i < AI; ++i)
        ra[i] = *reinterpret_cast<const float4 *>(g.A + abase[i] + ko + kq * 4);
#pragma unroll
      for (int i = 0; i < BI; ++i)
        rb[i] = *reinterpret_cast<const float4 *>(g.W + wbase[i] + (k0 + BK) + kq * 4);
    }
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
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Row-table and residual loads are
  // issued unconditionally on clamped addresses (16 in flight, one wait);
  // only the stores are predicated.
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
      float old[16];
      if (resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = g.C[roff[r] + nn];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (relu) v = fmaxf(v, 0.f);
        if (resid) v = old[r] + v;
        if (rok[r] && nv) g.C[roff[r] + nn] = v;
      }
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
struct ProfRec { hipEvent_t a, b; double flops; int variant; };
static std::vector<ProfRec> g_recs;
static int g_prof_every = 0;
static long long g_gemm_calls = 0;

extern "C" int sc_prof_enable(int sample_every) {
  g_prof_every = sample_every;
  g_gemm_calls = 0;
  return SC_OK;
}

// ms[v], flops[v], n[v] for v = 0 naive, 1 = 32x128 tile, 2 = 128x128, 3 = 64x64
extern "C" int sc_prof_collect(double *ms, double *flops, long long *n) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { sc_set_error("sc_prof_collect: %s", hipGetErrorString(e)); return SC_ERR_LAUNCH; }
  for (int v = 0; v < 4; ++v) { ms[v] = 0; flops[v] = 0; n[v] = 0; }
  for (auto &r : g_recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.variant] += t;
      flops[r.variant] += r.flops;
      n[r.variant] += 1;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  g_recs.clear();
  return SC_OK;
}

extern "C" int sc_gemm(const float *A, const int32_t *a_rows, int lda, const float *W,
                       const float *bias, float *C, const int32_t *c_rows, int ldc, int M, int N,
                       int K, int flags, int conv_f1, void *stream) {
  SC_CHECK_ARG(A && W && C, "null pointer");
  SC_CHECK_ARG(M >= 0 && N > 0 && K > 0 && lda > 0 && ldc >= N, "bad dimensions");
  if (M == 0) return SC_OK;
  if (g_force_naive < 0) {
    const char *e = getenv("SC_GEMM_NAIVE");
    g_force_naive = (e && e[0] == '1') ? 1 : 0;
  }
  GemmArgs g{A, a_rows, lda, W, bias, C, c_rows, ldc, M, N, K, flags, conv_f1};
  hipStream_t st = (hipStream_t)stream;
  bool aligned = (K % 32 == 0) && (lda % 4 == 0) && (((uintptr_t)A & 15) == 0) &&
                 (((uintptr_t)W & 15) == 0) && (conv_f1 == 0 || lda % 32 == 0);
  int variant;
  if ((flags & SC_GEMM_NAIVE) || g_force_naive || !aligned) variant = 0;
  else if (M <= 32) variant = 1;
  else if ((long)cdiv(M, 128) * cdiv(N, 128) >= 192) variant = 2;
  else variant = 3;
  ProfRec rec;
  const bool sample = g_prof_every > 0 && (g_gemm_calls++ % g_prof_every == 0);
  if (sample) {
    (void)hipEventCreate(&rec.a);
    (void)hipEventCreate(&rec.b);
    rec.flops = 2.0 * M * N * K;
    rec.variant = variant;
    (void)hipEventRecord(rec.a, st);
  }
  if (variant == 0) {
    long total = (long)M * N;
    gemm_naive_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(g);
  } else if (variant == 1) {
    gemm_mfma_kernel<32, 128, 1, 4><<<dim3(cdiv(N, 128), cdiv(M, 32)), 256, 0, st>>>(g);
  } else if (variant == 2) {
    gemm_mfma_kernel<128, 128, 2, 2><<<dim3(cdiv(N, 128), cdiv(M, 128)), 256, 0, st>>>(g);
  } else {
    gemm_mfma_kernel<64, 64, 2, 2><<<dim3(cdiv(N, 64), cdiv(M, 64)), 256, 0, st>>>(g);
  }
  if (sample) {
    (void)hipEventRecord(rec.b, st);
    g_recs.push_back(rec);
  }
  SC_CHECK_LAUNCH();
  return SC_OK;
}
