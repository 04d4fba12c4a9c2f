// Stream-resident decoder layer of libscasr (gfx950), round 6: ONE workgroup per stream runs the whole attention half of
// a decoder layer - both attentions, all H heads - and two launches make a layer:
//
//   A'  sc_dec_layer_stream   grid (stream)                     512 threads = 8 waves = the H (8) heads
//         x    = x_in + b2' + sum of the previous layer's feed-forward split sums   (layer 0: embed*sqrt(d) + PE)
//         q|k|v = LayerNorm1(x) . Wqkv^T + b of ALL heads; K|V row appended; self-attention over the distinct pool rows
//         x'   = x + bo + linear_out(self-attention)                               (complete rows: no partial products)
//         q    = LayerNorm2(x') . Wq^T + bq;  cross-attention over the stream's shared encoder K|V
//         x''  = x' + bo2 + linear_out(cross-attention) -> x_out;   LayerNorm3(x'') -> xn_out
//   C'  sc_dec_layer_ffn_xn   (gemm.hip: ffn_fused_kernel without prologue)   split sums of W2 . relu(W1 . xn + b1), by row id
//
// reference semantics: speechcatcher/model/decoder/decoder_layer.py:80-132, model/attention/multi_head_attention.py:63-133,
// transformer_decoder.py:231 (embedding).
//
// Why (docs/r06_findings.md): the three-launch form (decoder_layer.hip) puts a stream on TWO compute units
// (four heads each) and a full 128-stream bucket on all 256 - and every one of those workgroups spends 3/4 of its time in
// latency chains (partial-sum fetches, weight bursts from L2, merges) during which its CU does nothing, while the encoder
// side, whose kernels need whole CUs as well, waits its turn: wall = decode time + encoder time (29.5 % of the round-5
// wall).  Here a stream owns ONE CU for the layer: the rows never leave the workgroup between the two attentions (no
// partial products, no second prologue, no LayerNorm recomputed by a sibling), the matrix pipe of that CU is busy for
// ~2/3 of the kernel instead of 1/5, and a full bucket leaves 128 CUs to the encoder groups, which then run BESIDE the
// decode chain instead of between its kernels.
//
// BITS: every sum is evaluated in the canonical order of common.h - the result of a stream is bit for bit what the
// one-head and four-head workgroups of decoder_layer.hip produce (tests/test_gpu_ops.py lock-step, bit-reproducible
// serving):  projection = four K quarters, each one MFMA chain from zero, added in order, then the bias;  attention walk =
// lists of 512 positions, tiles of 16 entries, slot w takes tiles w, w+4, .., rescale per batch of two tiles, the slots
// merged in order;  output projection = one chain per head from zero, heads added in aligned groups of four in head
// order, the two groups in order, then  x + (sum + bias).
#define SC_STAMP_ON (p.dbg_stamp)
#include "common.h"
#include "attn.h"
#include <mutex>
#include <type_traits>

#define CTRL(s, f) sb.ctrl[(s) * 8 + (f)]
#define YSEQ(pp, s, h) (sb.yseq + (((long)(pp) * sb.S + (s)) * sb.W + (h)) * sb.LCAP)
#define ANC(pp, s) (sb.anc + ((long)(pp) * sb.S + (s)) * sb.LCAP * sb.W)

typedef float ds_f32x4 __attribute__((ext_vector_type(4)));

// 8 k-steps of one 32-wide k block for NA independent accumulators, interleaved (consecutive MFMAs never share an
// accumulator); per accumulator the steps run j = 0..7 - the chain decoder_layer.hip's dl_mfma8_il builds
template <int NA>
__device__ __forceinline__ void ds_mfma8_il(ds_f32x4 (&acc)[NA], const float4 &a0, const float4 &a1, const float4 (&b0)[NA],
                                            const float4 (&b1)[NA]) {
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const float4 &bb = j < 4 ? b0[t] : b1[t];
      const float bj = (j & 3) == 0 ? bb.x : (j & 3) == 1 ? bb.y : (j & 3) == 2 ? bb.z : bb.w;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bj, acc[t], 0, 0, 0);
    }
}
// ... with one A tile PER accumulator (the output projection: one accumulator per head)
template <int NA>
__device__ __forceinline__ void ds_mfma8_heads(ds_f32x4 (&acc)[NA], const float4 (&a0)[NA], const float4 (&a1)[NA],
                                               const float4 (&b0)[NA], const float4 (&b1)[NA]) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < NA; ++t) {
      const float4 &aa = j < 4 ? a0[t] : a1[t];
      const float4 &bb = j < 4 ? b0[t] : b1[t];
      const float aj = (j & 3) == 0 ? aa.x : (j & 3) == 1 ? aa.y : (j & 3) == 2 ? aa.z : aa.w;
      const float bj = (j & 3) == 0 ? bb.x : (j & 3) == 1 ? bb.y : (j & 3) == 2 ? bb.z : bb.w;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj, bj, acc[t], 0, 0, 0);
    }
}

struct DecStreamArgs {
  sc_search sb;
  int li;
  const float *xin;   // residual stream before this layer [S*W][d] (layer 0: unused)
  float *xout;        // ... after both attentions (x'')
  float *xn;          // LayerNorm3(x'') [S*W][d]: the feed-forward's input
  // prologue:  x[row] = xin[row] + (tree sum_{z < npart} part[z*zs + row*d + :] + pbias)
  const float *part;
  int npart;
  long zs;
  const float *pbias;
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *ln3_g, *ln3_b;
  const float *wqkv, *bqkv, *wo, *bo;    // panel-packed Wqkv [3d][d], Wo [d][d] of self_attn
  const float *wq, *bq, *wo2, *bo2;      // panel-packed Wq, Wo2 of src_attn
  int dbg_stamp;
};

namespace dstream {
constexpr int D = 256, DK = 32, H = 8, WM = 10, NTH = 512;
constexpr int LDX = D + 4, KI = D / 32, C4 = D / 4;
constexpr int PCS = 512;               // positions per row list (canonical)
constexpr int LDC = 36;                // row stride of a head's context tile (A operand of the output projection)
constexpr int NPS = 5;                 // partial attention states per head (self: 4 slots + the new token's row)
constexpr int LPO = DK + 1;
constexpr int AP = NPS * (2 * 16 + WM * LPO);   // floats of one head's partial states: pm | pl [NPS][16], pO [NPS][WM][LPO]
// LDS map (floats)
constexpr int O_XN = 0;                          // [16][LDX] LayerNorm tile / staging of the projections' results
constexpr int O_GB = O_XN + 16 * LDX;            // [3][2][D] LayerNorm gamma | beta of norm1..3
constexpr int O_ANCS = O_GB + 6 * D;             // [16] pool rows of the new tokens
constexpr int O_ROWS = O_ANCS + 16;              // [PCS * WM] row list + [16] wave totals
constexpr int O_QS = O_ROWS + PCS * WM + 16;     // [H][16][DK] queries / sqrt(dk)
constexpr int O_KVC = O_QS + H * 16 * DK;        // union: kvn [H][WM][2 DK] (self: new token's k|v)  |  ctx [H][16][LDC]
constexpr int KVC = (H * WM * 2 * DK > H * 16 * LDC) ? H * WM * 2 * DK : H * 16 * LDC;
constexpr int O_U = O_KVC + KVC;                 // partial attention states [H][AP] (free during the walks)
constexpr int LDS_FLOATS = O_U + H * AP;
}   // namespace dstream

// partial state of a wave -> slot `slot` of its head (compact: only the WM live hypothesis rows of pO are kept; the
// arithmetic is attn.h's mattn_store_partial)
__device__ __forceinline__ void ds_store_partial(const MAttn<32> &st, float *pm, float *pl, float *pO, int slot, int lane) {
  using namespace dstream;
  const int n = lane & 15, kg = lane >> 4;
  float l = st.l;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (kg == 0) {
    pm[slot * 16 + n] = st.m;
    pl[slot * 16 + n] = l;
  }
  if (n < WM) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) pO[(slot * WM + n) * LPO + (4 * kg + jj) * 2 + dt] = st.o[dt][jj];
  }
}
// context element (h, c) out of NP partial states, in slot order (attn.h: mattn_final)
template <int NP>
__device__ __forceinline__ float ds_final(const float *pm, const float *pl, const float *pO, int h, int c) {
  using namespace dstream;
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < NP; ++w) M = fmaxf(M, pm[w * 16 + h]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    const float mw = pm[w * 16 + h];
    const float g = (mw == -INFINITY) ? 0.f : __expf(mw - M);
    den = fmaf(g, pl[w * 16 + h], den);
    num = fmaf(g, pO[(w * WM + h) * LPO + c], num);
  }
  return den > 0.f ? num / den : 0.f;
}

// wave-private LDS hand-off (a wave's lanes exchange data through its OWN LDS region: no workgroup barrier): the DS
// operations of a wave are executed in issue order, the wait + compiler barrier keep the accesses from being re-ordered
__device__ __forceinline__ void ds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// 512 threads = 8 waves = the 8 heads: wave h owns head h from the projection of its columns to its context tile (all four
// K quarters of the projection, all four slots of the attention walk, the merge) WITHOUT a workgroup barrier in between;
// 256 registers per lane buy three K|V batches of the walk and two k blocks of weight fragments in flight per wave.
// Two K|V tiles (t0, t0 + 4) of a walk in registers WITHOUT their hypothesis masks (attn.h's MBatch carries them: 8 of 40
// registers; here three batches per wave are in flight and the masks are looked up when the batch is worked off - the row list
// is in LDS, the cross-attention's mask is a comparison).  The arithmetic is mattn_load / mattn_batch of attn.h, bit for bit.
struct DsTiles {
  float kr[2][8], vr[2][4][2];
};
template <bool KVH, class RowFn>
__device__ __forceinline__ void ds_load2(DsTiles &b, const float *kv, int t0, int lane, RowFn rowfn) {
  using namespace dstream;
  const int n = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#ifdef SC_DS_PROBE_SAMEROWS   // timing probe only (tools/build_variant.sh): every tile = tile 0, i.e. a walk whose loads hit the L1
    const int t = i * 0 * t0;
#else
    const int t = t0 + 4 * i;
#endif
    long ke;
    unsigned unused;
    rowfn(16 * t + n, ke, unused);
    kv_loadn<4, KVH>(kv, ke + 8 * kg, &b.kr[i][0]);
    kv_loadn<4, KVH>(kv, ke + 8 * kg + 4, &b.kr[i][4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long ve;
      rowfn(16 * t + 4 * kg + j, ve, unused);
      kv_loadn<2, KVH>(kv, ve + D + n * 2, b.vr[i][j]);
    }
  }
}
// the batch's arithmetic in two halves, so that a wave can issue the score MFMAs of batch k + 1 AHEAD of the softmax of batch k
// (the matrix pipe works through them while the VALU does the masks, maxima and exponentials of batch k - with one monolithic
// routine per batch a wave's matrix work stops for every softmax, and two waves per SIMD do not fill the holes by themselves:
// 29 us for 21 us of matrix work in a probe whose loads all hit the L1).  Same instructions per accumulator, same order.
__device__ __forceinline__ void ds_scores(f32x4v (&s)[2], const DsTiles &b, const float (&qb)[8]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) s[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.kr[i][q], qb[q], s[i], 0, 0, 0);
}
template <class RowFn>
__device__ __forceinline__ void ds_finish(MAttn<32> &st, const DsTiles &b, f32x4v (&s)[2], int t0, int ntiles, int lane, RowFn rowfn) {
  const int n = lane & 15, kg = lane >> 4;
  float mloc = -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long unused;
      unsigned hm;
      rowfn(16 * (t0 + 4 * i) + 4 * kg + j, unused, hm);
      if (t0 + 4 * i >= ntiles) hm = 0u;
      s[i][j] = ((hm >> n) & 1u) ? s[i][j] : -INFINITY;
      mloc = fmaxf(mloc, s[i][j]);
    }
  }
  mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
  mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
  const float mnew = fmaxf(st.m, mloc);
  const float muse = (mnew == -INFINITY) ? 0.f : mnew;
  const float corr = (st.m == -INFINITY) ? 0.f : __expf(st.m - muse);
  st.m = mnew;
  st.l *= corr;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) st.o[dt] *= corr;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float pr = (s[i][j] == -INFINITY) ? 0.f : __expf(s[i][j] - muse);
      st.l += pr;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) st.o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.vr[i][j][dt], pr, st.o[dt], 0, 0, 0);
    }
  }
}

template <bool FIRST, bool KVH>
__global__ __launch_bounds__(512, 2) void dec_layer_stream_kernel(DecStreamArgs p) {
  using namespace dstream;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sc_search &sb = p.sb;
  if ((int)blockIdx.x >= (sb.rowmap ? sb.n_rows / sb.W : sb.S)) return;
  const int s = sb.rowmap ? sb.rowmap[blockIdx.x * sb.W] / sb.W : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = wave;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int nh = CTRL(s, SC_C_NHYP);
  if (nh <= 0) return;
  const int L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR), T = CTRL(s, SC_C_T);
  const int W = sb.W;
  float *Xn = smem + O_XN;
  float *gb = smem + O_GB;
  int *ancs = reinterpret_cast<int *>(smem + O_ANCS);
  int *srows = reinterpret_cast<int *>(smem + O_ROWS), *swtot = srows + PCS * WM;
  float *qs = smem + O_QS + head * 16 * DK;
  float *kvn = smem + O_KVC + head * WM * 2 * DK;
  float *ctxs = smem + O_KVC;                       // [H][16][LDC]
  float *pm = smem + O_U + head * AP, *pl = pm + NPS * 16, *pO = pl + NPS * 16;
  const int r = lane & 15, kk = lane >> 4;
  const long skv0 = ((long)s * sb.n_layers + p.li) * sb.kv_rows * 2 * D + head * DK;
  const long ckv0 = ((long)s * sb.n_layers + p.li) * sb.TCAP * 2 * D + head * DK;

  SC_STAMP(0, 0);
  if (tid < 16) ancs[tid] = ANC(cur, s)[(long)(L - 1) * W + min(tid, nh - 1)];
  // LayerNorm parameters of the three norms -> LDS (requested first, parked behind the partial sums)
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 6 * C4) {
    const float *src = tid < C4 ? p.ln1_g : tid < 2 * C4 ? p.ln1_b : tid < 3 * C4 ? p.ln2_g : tid < 4 * C4 ? p.ln2_b : tid < 5 * C4 ? p.ln3_g : p.ln3_b;
    gbv = *reinterpret_cast<const float4 *>(src + 4 * (tid % C4));
  }
  // ancestor slots of the first 512 positions: the row list of the self-attention is built while the partial sums travel
  int slp[WM] = {};
  {
    const int *anc0 = ANC(cur, s);
    const bool live0 = tid < L - 1;
#pragma unroll
    for (int h = 0; h < WM; ++h) slp[h] = anc0[(long)(live0 ? tid : 0) * W + min(h, nh - 1)];
  }
  int U0 = 0;

  // ------------------------------------------------------------------ projection machinery (used three lines below already:
  // the first weight fragments of the Q|K|V projection are requested BEFORE the prologue's partial sums - they arrive first
  // and wait in registers; 256 registers per lane pay for it)
  // NTP tiles of 16 output columns per pass; the wave walks ALL eight k blocks: the four K quarters (two k blocks each) are
  // four MFMA chains from zero, added in order - ((P0 + P1) + P2) + P3 - as the canonical order wants (common.h).
  // B operands straight from the fragment-packed weights, two k blocks ahead of the MFMAs that use them.
  auto load_b = [&](const float *wp, auto tile_of, int ki, auto &x0, auto &x1) {
    constexpr int NTP = sizeof(x0) / sizeof(x0[0]);
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const float4 *wq = reinterpret_cast<const float4 *>(wp) + ((long)tile_of(t) * KI + ki) * 128 + lane;
      x0[t] = wq[0];
      x1[t] = wq[64];
    }
  };
  // PRE: the fragments of k blocks 0 and 1 are in (b0, b1), (c0, c1) already
  auto proj_pass = [&](const float *wp, auto tile_of, auto &sum, auto &b0, auto &b1, auto &c0, auto &c1) {
    constexpr int NTP = sizeof(sum) / sizeof(sum[0]);
    ds_f32x4 acc[NTP];
#pragma unroll
    for (int ki = 0; ki < KI; ++ki) {
      if ((ki & 1) == 0) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) acc[t] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const float *ab = Xn + r * LDX + ki * 32 + 8 * kk;
      const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
      float4 n0[NTP], n1[NTP];
      if (ki + 2 < KI) load_b(wp, tile_of, ki + 2, n0, n1);   // in flight during two k blocks of MFMAs
      ds_mfma8_il<NTP>(acc, a0, a1, b0, b1);
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        b0[t] = c0[t];
        b1[t] = c1[t];
        if (ki + 2 < KI) {
          c0[t] = n0[t];
          c1[t] = n1[t];
        }
      }
      if (ki & 1) {   // a K quarter is complete
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          if (ki == 1) sum[t] = acc[t];
          else sum[t] += acc[t];
        }
      }
    }
  };
  auto tile_qkv0 = [&](int t) { return ((t / 2) * D + head * DK) / 16 + (t % 2); };               // q0 q1 k0
  auto tile_qkv1 = [&](int t) { return (((t + 3) / 2) * D + head * DK) / 16 + ((t + 3) % 2); };   // k1 v0 v1
  auto tile_q = [&](int t) { return (head * DK) / 16 + t; };
  float4 qb0[3], qb1[3], qc0[3], qc1[3];
  float bqkv[6];   // this lane's bias elements of the six column tiles
  auto early_w = [&]() {   // issued BEHIND the prologue's first loads (loads return in issue order: the rows come first)
    load_b(p.wqkv, tile_qkv0, 0, qb0, qb1);
    load_b(p.wqkv, tile_qkv0, 1, qc0, qc1);
#pragma unroll
    for (int t = 0; t < 6; ++t) bqkv[t] = p.bqkv[(t / 2) * D + head * DK + (t % 2) * 16 + r];
  };

  // ------------------------------------------------------------------ prologue: x rows.  Thread tid owns the float4 pieces
  // (row tid / 64, columns 4 (tid % 64)) and (row 8 + tid / 64, ...) of the 16 x D tile for the whole kernel: the residual
  // stream stays in its registers
  const int xc4 = tid % C4;
  const int xrow[2] = {tid / C4, 8 + tid / C4};
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 xres[2] = {zero4, zero4};
  if (FIRST) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (xrow[q] < W) {
        const int tok = YSEQ(cur, s, min(xrow[q], nh - 1))[L - 1];
        const float sq = sqrtf((float)D);
        const float4 ev = *reinterpret_cast<const float4 *>(sb.embed + (long)tok * D + 4 * xc4);
        const float4 pe = *reinterpret_cast<const float4 *>(sb.pe + (long)(L - 1) * D + 4 * xc4);
        xres[q] = make_float4(ev.x * sq + pe.x, ev.y * sq + pe.y, ev.z * sq + pe.z, ev.w * sq + pe.w);
      }
    early_w();
    U0 = mattn_build_rows<WM, true, NTH, PCS>(srows, swtot, ANC(cur, s), 0, L - 1, W, nh, tid, lane, wave, slp);
  } else {
    float4 xi[2] = {zero4, zero4}, pbv = zero4;
    if (p.pbias) pbv = *reinterpret_cast<const float4 *>(p.pbias + 4 * xc4);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (xrow[q] < W) xi[q] = *reinterpret_cast<const float4 *>(p.xin + ((long)s * W + xrow[q]) * D + 4 * xc4);
    float4 yv[2] = {zero4, zero4};
    for (int z0 = 0; z0 < p.npart; z0 += 8) {
      float4 pv[2][8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          pv[q][z] = zero4;   // (the pieces of the padding rows request nothing)
          if (xrow[q] < W && z0 + z < p.npart)
            pv[q][z] = *reinterpret_cast<const float4 *>(p.part + (long)(z0 + z) * p.zs + ((long)s * W + xrow[q]) * D + 4 * xc4);
        }
      if (z0 == 0) {
        early_w();
        U0 = mattn_build_rows<WM, true, NTH, PCS>(srows, swtot, ANC(cur, s), 0, L - 1, W, nh, tid, lane, wave, slp);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 h8 = sc_tree8(pv[q]);   // common.h: ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)), batches added in order
        yv[q] = z0 == 0 ? h8 : sc_add4(yv[q], h8);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (xrow[q] < W)
        xres[q] = make_float4(xi[q].x + (yv[q].x + pbv.x), xi[q].y + (yv[q].y + pbv.y), xi[q].z + (yv[q].z + pbv.z),
                              xi[q].w + (yv[q].w + pbv.w));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<float4 *>(Xn + xrow[q] * LDX + 4 * xc4) = xres[q];   // rows >= W: zeros
  if (tid < 6 * C4) *reinterpret_cast<float4 *>(gb + 4 * tid) = gbv;
  // pre-zero what the MFMA tiles pad with: query rows >= WM of every head
  for (int e = tid; e < H * (16 - WM) * DK; e += NTH) {
    const int hh = e / ((16 - WM) * DK), rem = e % ((16 - WM) * DK);
    smem[O_QS + hh * 16 * DK + WM * DK + rem] = 0.f;
  }
  __syncthreads();
  SC_STAMP(0, 1);

  // LayerNorm of the tile in place: 16 lanes per row, all 16 rows at once (the code of decoder_layer.hip / ffn_fused_kernel)
  auto layer_norm = [&](const float *g) {
    if (tid < 256) {
      constexpr int Q4 = D / 64;
      const int i = tid >> 4, sb16 = tid & 15;
      float4 x[Q4];
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        x[q] = *reinterpret_cast<const float4 *>(Xn + i * LDX + 4 * (sb16 + 16 * q));
        sum += (x[q].x + x[q].y) + (x[q].z + x[q].w);
      }
      const float mean = group_sum<16>(sum) / (float)D;
      float q2 = 0.f;
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        const float a = x[q].x - mean, b = x[q].y - mean, c = x[q].z - mean, e = x[q].w - mean;
        q2 += (a * a + b * b) + (c * c + e * e);
      }
      const float rstd = 1.0f / sqrtf(group_sum<16>(q2) / (float)D + sb.ln_eps);
      if (i < W) {   // rows >= W stay zero
#pragma unroll
        for (int q = 0; q < Q4; ++q) {
          const float4 gm = *reinterpret_cast<const float4 *>(g + 4 * (sb16 + 16 * q));
          const float4 bt = *reinterpret_cast<const float4 *>(g + D + 4 * (sb16 + 16 * q));
          *reinterpret_cast<float4 *>(Xn + i * LDX + 4 * (sb16 + 16 * q)) =
              make_float4((x[q].x - mean) * rstd * gm.x + bt.x, (x[q].y - mean) * rstd * gm.y + bt.y,
                          (x[q].z - mean) * rstd * gm.z + bt.z, (x[q].w - mean) * rstd * gm.w + bt.w);
        }
      }
    }
    __syncthreads();
  };
  layer_norm(gb);
  SC_STAMP(0, 2);

  // ------------------------------------------------------------------ q|k|v of this head: two passes of three column tiles
  {
    ds_f32x4 s0[3], s1[3];
    proj_pass(p.wqkv, tile_qkv0, s0, qb0, qb1, qc0, qc1);
    load_b(p.wqkv, tile_qkv1, 0, qb0, qb1);
    load_b(p.wqkv, tile_qkv1, 1, qc0, qc1);
    proj_pass(p.wqkv, tile_qkv1, s1, qb0, qb1, qc0, qc1);
    SC_STAMP(0, 3);
    const float scale = sqrtf((float)DK);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int which = t / 2, c = (t % 2) * 16 + r;
      const float bias = bqkv[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = 4 * kk + j;
        if (w < WM) {
          float v = 0.f;
          if (w < W) v = (t < 3 ? s0[t % 3][j] : s1[t % 3][j]) + bias;
          if (which == 0) {
            qs[w * DK + c] = v / scale;
          } else {
            kvn[w * 2 * DK + (which - 1) * DK + c] = v;
            // append this token's K|V row into its pool row; later steps read it from the cache
            if (w < nh) kv_store1<KVH>(sb.skv, skv0 + (long)ancs[w] * 2 * D + (which - 1) * D + c, v);
          }
        }
      }
    }
  }
  ds_wave_sync();   // (qs / kvn of this head are read by this wave only)
  SC_STAMP(0, 4);

  // ------------------------------------------------------------------ attention walks: the wave carries the partial states of
  // all four slots of its head (slot w = tiles w, w + 4, ... of every list, batches of two tiles: attn.h); the batches of a
  // group of eight tiles are (w, w + 4), w = 0..3: their loads run two batches ahead of the arithmetic
  MAttn<DK> st[4];
  auto walk4 = [&](const float *kv, int ntiles, auto rowfn) {
    float qb[DK / 4];
#pragma unroll
    for (int i = 0; i < DK / 4; ++i) qb[i] = qs[(lane & 15) * DK + (DK / 4) * (lane >> 4) + i];
    const int nb = 4 * (ntiles / 8) + min(ntiles % 8, 4);   // batches with a first tile < ntiles, in (group, slot) order
    // three buffers, batch k of the sequence in buffer k % 3 for slot k % 4: the loop body is unrolled over 12 batches;
    // the loads run TWO batches ahead of the arithmetic (8 waves x 3 batches x 8 KB in flight or in use per CU)
    DsTiles B0, B1, B2;
    auto ld = [&](DsTiles &b, int i) {
      if (i < nb) ds_load2<KVH>(b, kv, 8 * (i >> 2) + (i & 3), lane, rowfn);
    };
    auto sc = [&](f32x4v (&sv)[2], const DsTiles &b, int i) {
      if (i < nb) ds_scores(sv, b, qb);
    };
    auto fin = [&](MAttn<DK> &stw, const DsTiles &b, f32x4v (&sv)[2], int i) {
      if (i < nb) ds_finish(stw, b, sv, 8 * (i >> 2) + (i & 3), ntiles, lane, rowfn);
    };
    f32x4v sA[2], sB[2];
    ld(B0, 0);
    ld(B1, 1);
    sc(sA, B0, 0);
    for (int i = 0; i < nb; i += 12) {
      ld(B2, i + 2);  sc(sB, B1, i + 1);  fin(st[0], B0, sA, i);
      ld(B0, i + 3);  sc(sA, B2, i + 2);  fin(st[1], B1, sB, i + 1);
      ld(B1, i + 4);  sc(sB, B0, i + 3);  fin(st[2], B2, sA, i + 2);
      ld(B2, i + 5);  sc(sA, B1, i + 4);  fin(st[3], B0, sB, i + 3);
      ld(B0, i + 6);  sc(sB, B2, i + 5);  fin(st[0], B1, sA, i + 4);
      ld(B1, i + 7);  sc(sA, B0, i + 6);  fin(st[1], B2, sB, i + 5);
      ld(B2, i + 8);  sc(sB, B1, i + 7);  fin(st[2], B0, sA, i + 6);
      ld(B0, i + 9);  sc(sA, B2, i + 8);  fin(st[3], B1, sB, i + 7);
      ld(B1, i + 10); sc(sB, B0, i + 9);  fin(st[0], B2, sA, i + 8);
      ld(B2, i + 11); sc(sA, B1, i + 10); fin(st[1], B0, sB, i + 9);
      ld(B0, i + 12); sc(sB, B2, i + 11); fin(st[2], B1, sA, i + 10);
      ld(B1, i + 13); sc(sA, B0, i + 12); fin(st[3], B2, sB, i + 11);
    }
  };
  // output projection of ALL heads' contexts: wave w owns the column tiles 2 w and 2 w + 1; one chain per head from zero,
  // heads added in aligned groups of four in head order, the groups in order; y -> Xn tile; x <- x + (y + bias) by the owners
  float4 ob0[2][4], ob1[2][4];   // fragments of heads 0..3, requested ahead of the merge
  float4 obias[1];               // ... and the bias piece of this thread's x pieces (both rows: the same columns)
  auto out_fetch = [&](const float *wop, int half) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const float4 *wq = reinterpret_cast<const float4 *>(wop) + ((long)(2 * wave + t) * KI + half * 4 + hh) * 128 + lane;
        ob0[t][hh] = wq[0];
        ob1[t][hh] = wq[64];
      }
  };
  // a head's partial states -> its context tile [16][LDC] (rows >= nh zero): stores and merge by the head's own wave
  auto merge_ctx = [&](auto np_tag, const float *wop, const float *obp) {
    constexpr int NP = decltype(np_tag)::value;
    ds_wave_sync();
    SC_STAMP(1, NP == 5 ? 0 : 4);
    out_fetch(wop, 0);   // the output projection's first fragments travel during the merge (the walk's registers are free)
    obias[0] = *reinterpret_cast<const float4 *>(obp + 4 * xc4);
    SC_STAMP(1, NP == 5 ? 1 : 5);
    float cv[WM * DK / 64];
#pragma unroll
    for (int q = 0; q < WM * DK / 64; ++q) {
      const int e = lane + 64 * q, h = e / DK, c = e % DK;
      cv[q] = h < nh ? ds_final<NP>(pm, pl, pO, h, c) : 0.f;
    }
    ds_wave_sync();   // (kvn of this head - read for the fifth state - lies where the context tiles go: all reads are done)
    SC_STAMP(1, NP == 5 ? 2 : 6);
#pragma unroll
    for (int q = 0; q < WM * DK / 64; ++q) {
      const int e = lane + 64 * q, h = e / DK, c = e % DK;
      ctxs[(head * 16 + h) * LDC + c] = cv[q];
    }
    SC_STAMP(1, NP == 5 ? 3 : 7);
  };
  auto out_proj = [&](const float *wop, const float4 bv) {
    ds_f32x4 y[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 a0[4], a1[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const float *ab = ctxs + ((half * 4 + hh) * 16 + r) * LDC + 8 * kk;
        a0[hh] = *reinterpret_cast<const float4 *>(ab);
        a1[hh] = *reinterpret_cast<const float4 *>(ab + 4);
      }
      ds_f32x4 acc0[4], acc1[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        acc0[hh] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[hh] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      ds_mfma8_heads<4>(acc0, a0, a1, ob0[0], ob1[0]);
      ds_mfma8_heads<4>(acc1, a0, a1, ob0[1], ob1[1]);
      if (half == 0) out_fetch(wop, 1);   // (behind the MFMAs that read the registers)
      ds_f32x4 g0 = acc0[0], g1 = acc1[0];
      g0 += acc0[1]; g0 += acc0[2]; g0 += acc0[3];
      g1 += acc1[1]; g1 += acc1[2]; g1 += acc1[3];
      if (half == 0) { y[0] = g0; y[1] = g1; }
      else { y[0] += g0; y[1] += g1; }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * kk + j < WM) Xn[(4 * kk + j) * LDX + (2 * wave + t) * 16 + r] = y[t][j];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (xrow[q] < W) {
        const float4 yv = *reinterpret_cast<const float4 *>(Xn + xrow[q] * LDX + 4 * xc4);
        xres[q] = make_float4(xres[q].x + (yv.x + bv.x), xres[q].y + (yv.y + bv.y), xres[q].z + (yv.z + bv.z), xres[q].w + (yv.w + bv.w));
      }
      *reinterpret_cast<float4 *>(Xn + xrow[q] * LDX + 4 * xc4) = xres[q];   // (a piece of y is read by its owner only) rows >= W: zeros
    }
    __syncthreads();
  };

  // ---- self-attention
#pragma unroll
  for (int w = 0; w < 4; ++w) mattn_init(st[w]);
  {
    const int *anc = ANC(cur, s);
    const int Lc = L - 1;   // cached positions; the new token's own row is the fifth partial state
    int urows = nh + U0;
    auto rowfn = [&](int idx, long &ke, unsigned &hm) {
      const int e = srows[min(idx, PCS * W - 1)];   // entries >= U are zero: no hypothesis
      hm = (unsigned)e >> 16;
      ke = skv0 + (long)(e & 0xFFFF) * 2 * D;
    };
    walk4(sb.skv, cdiv(U0, 16), rowfn);
    for (int c0 = PCS; c0 < Lc; c0 += PCS) {
      __syncthreads();   // the list is rebuilt for the next positions
      const int U = mattn_build_rows<WM, false, NTH, PCS>(srows, swtot, anc, c0, Lc, W, nh, tid, lane, wave, slp);
      urows += U;
      walk4(sb.skv, cdiv(U, 16), rowfn);
    }
    if (sb.stat_rows && tid == 0) atomicAdd(&sb.stat_rows[2], (unsigned long long)urows);
    // the new token: hypothesis h attends to its own row (slot h at position L-1, still in LDS) only
    if (lane < 16) {
      const int h = lane;
      float sdot = -INFINITY;
      if (h < nh) {
        sdot = 0.f;
#pragma unroll
        for (int c = 0; c < DK; ++c) sdot = fmaf(qs[h * DK + c], kvn[h * 2 * DK + c], sdot);
#pragma unroll
        for (int c = 0; c < DK; ++c) pO[(4 * WM + h) * LPO + c] = kvn[h * 2 * DK + DK + c];
      }
      pm[4 * 16 + h] = sdot;
      pl[4 * 16 + h] = h < nh ? 1.f : 0.f;
    }
  }
  SC_STAMP(0, 5);
#pragma unroll
  for (int w = 0; w < 4; ++w) ds_store_partial(st[w], pm, pl, pO, w, lane);
  SC_STAMP(1, 8);
  __syncthreads();   // (the context tiles lie where OTHER heads' kvn rows are: every wave is past its fifth state)
  merge_ctx(std::integral_constant<int, 5>{}, p.wo, p.bo);
  __syncthreads();
  SC_STAMP(0, 6);
  out_proj(p.wo, obias[0]);
  // the cross-attention's q fragments travel during the LayerNorm
  float4 xb0[2], xb1[2], xc0[2], xc1[2];
  load_b(p.wq, tile_q, 0, xb0, xb1);
  load_b(p.wq, tile_q, 1, xc0, xc1);
  const float bq2[2] = {p.bq[head * DK + r], p.bq[head * DK + 16 + r]};
  layer_norm(gb + 2 * D);
  SC_STAMP(0, 7);

  // ------------------------------------------------------------------ cross-attention: q of this head (two column tiles)
  {
    ds_f32x4 sq[2];
    proj_pass(p.wq, tile_q, sq, xb0, xb1, xc0, xc1);
    const float scale = sqrtf((float)DK);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = t * 16 + r;
      const float bias = bq2[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = 4 * kk + j;
        if (w < WM) qs[w * DK + c] = (w < W ? sq[t][j] + bias : 0.f) / scale;
      }
    }
  }
  ds_wave_sync();
  SC_STAMP(0, 8);
#pragma unroll
  for (int w = 0; w < 4; ++w) mattn_init(st[w]);
  {
    const unsigned all = (1u << nh) - 1u;
    walk4(sb.ckv, cdiv(T, 16), [&](int idx, long &ke, unsigned &hm) {
      hm = idx < T ? all : 0u;
      ke = ckv0 + (long)min(idx, T - 1) * 2 * D;
    });
  }
  SC_STAMP(0, 9);
#pragma unroll
  for (int w = 0; w < 4; ++w) ds_store_partial(st[w], pm, pl, pO, w, lane);
  merge_ctx(std::integral_constant<int, 4>{}, p.wo2, p.bo2);   // (the context tile of a head is written and was read by its own wave only ...
  __syncthreads();                               //  ... until the output projection, which reads all of them)
  SC_STAMP(0, 10);
  out_proj(p.wo2, obias[0]);
  // x'' -> the residual stream; LayerNorm3(x'') -> the feed-forward's input rows
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (xrow[q] < W) *reinterpret_cast<float4 *>(p.xout + ((long)s * W + xrow[q]) * D + 4 * xc4) = xres[q];
  layer_norm(gb + 4 * D);
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (xrow[q] < W)
      *reinterpret_cast<float4 *>(p.xn + ((long)s * W + xrow[q]) * D + 4 * xc4) = *reinterpret_cast<const float4 *>(Xn + xrow[q] * LDX + 4 * xc4);
  SC_STAMP_END(0, 11);
}
SC_PHASE_GETTER(sc_phase_debug_stream)

// Which buckets take this form: the model dimensions it is instantiated for (d 256, 8 heads of 32, beam <= 10), fp32
// K|V rows and weights, and at least SC_STREAM_MIN_ROWS hypothesis rows in flight (below that the one-head workgroups of
// decoder_layer.hip spread a stream over more CUs: the chain is latency-bound there and the chip is mostly empty anyway).
// SC_DEC_STREAM = 0 | 1 (test hook) forces the choice for every bucket the form supports.
extern "C" int sc_dec_layer_stream_supported(int d, int H, int W, int F) {
  return d == dstream::D && H == dstream::H && W >= 1 && W <= dstream::WM && sc_ffn_ln_supported(d, F);
}
int sc_dec_layer_stream_form(const sc_search &sb) {
  if (!sc_dec_layer_stream_supported(sb.d, sb.H, sb.W, sb.F) || sb.act_half || !sb.dq || !sb.ffn_part) return 0;
  if (!sb.layers || !sb.layers[0].wqkv_pp || !sb.layers[0].wq_pp || !sb.layers[0].wo_pp || !sb.layers[0].wo2_pp) return 0;
  int min_rows = SC_STREAM_MIN_ROWS;
  if (const char *e = sc_hook("SC_STREAM_MIN")) min_rows = atoi(e);
  int on = (sb.rowmap ? sb.n_rows : sb.S * sb.W) >= min_rows;
  if (const char *e = sc_hook("SC_DEC_STREAM")) on = atoi(e) != 0;
  return on;
}

template <bool FIRST, bool KVH>
static void launch_stream_variant(const DecStreamArgs &p, int ns, hipStream_t st) {
  const size_t lds = (size_t)dstream::LDS_FLOATS * sizeof(float);
  static std::once_flag once;
  std::call_once(once, [&]() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dec_layer_stream_kernel<FIRST, KVH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  dec_layer_stream_kernel<FIRST, KVH><<<dim3(ns), dstream::NTH, lds, st>>>(p);
}

extern "C" int sc_dec_layer_stream(const sc_search *sbp, int layer, const float *xin, float *xout, float *xn_out,
                                   const float *ffn_part, int n_ffn_part, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xout && xn_out, "null");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  SC_CHECK_ARG(sc_dec_layer_stream_supported(sb.d, sb.H, sb.W, sb.F), "unsupported dimensions");
  SC_CHECK_ARG(!sb.act_half, "the stream-resident layer has no fp16 weight form");
  SC_CHECK_ARG(layer == 0 || (xin && xin != xout && ffn_part && n_ffn_part > 0), "x_in / partial sums missing");
  const sc_dec_layer &w = sb.layers[layer];
  SC_CHECK_ARG(w.wqkv_pp && w.wo_pp && w.wq_pp && w.wo2_pp, "panel-packed Wqkv / Wo / Wq / Wo2 missing");
  DecStreamArgs p{sb, layer, xin, xout, xn_out, ffn_part, n_ffn_part, (long)sb.S * sb.W * sb.d,
                  layer > 0 ? sb.layers[layer - 1].b2 : nullptr, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, w.ln3_g, w.ln3_b,
                  w.wqkv_pp, w.bqkv, w.wo_pp, w.bo, w.wq_pp, w.bq, w.wo2_pp, w.bo2, sc_phase_take(0)};
  hipStream_t st = (hipStream_t)stream;
  const int ns = sb.rowmap ? sb.n_rows / sb.W : sb.S;
  ProfScope prof = sc_prof_begin(st);
  if (sb.kv_half) {
    if (layer == 0) launch_stream_variant<true, true>(p, ns, st);
    else launch_stream_variant<false, true>(p, ns, st);
  } else {
    if (layer == 0) launch_stream_variant<true, false>(p, ns, st);
    else launch_stream_variant<false, false>(p, ns, st);
  }
  SC_CHECK_LAUNCH();
  {   // MFMA part: Q|K|V + q projections and the two output projections of the bucket's rows; bytes without the K|V rows
      // (bench.py adds them: cross from T, self from the device counter): the four weight matrices once, x in, x and xn out
    const double M = sb.rowmap ? sb.n_rows : sb.S * sb.W;
    sc_prof_end(prof, SC_PROF_LAYER_STREAM, 12.0 * M * sb.d * sb.d, 4.0 * (6.0 * sb.d * sb.d + 3.0 * M * sb.d));
  }
  return SC_OK;
}
