// Stream-resident decoder layer of libscasr (gfx950), round 6: ONE workgroup per stream runs the whole attention half of
// a decoder layer - both attentions, all H heads - and two launches make a layer:
//
//   A'  sc_dec_layer_stream   grid (stream)                     1024 threads = 16 waves = H (8) heads x 2 waves
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
constexpr int D = 256, DK = 32, H = 8, WM = 10, NTH = 1024;
constexpr int LDX = D + 4, KI = D / 32, C4 = D / 4;
constexpr int PCS = 512;               // positions per row list (canonical)
constexpr int LDP = 100;               // row stride of the projection's parked partial (6 tiles of 16 + 4)
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
constexpr int O_U = O_KVC + KVC;                 // union: parked projection partial [H][WM][LDP]  |  partial states [H][AP]
constexpr int UF = (H * WM * LDP > H * AP) ? H * WM * LDP : H * AP;
constexpr int LDS_FLOATS = O_U + UF;
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

template <bool FIRST, bool KVH>
__global__ __launch_bounds__(1024, 4) void dec_layer_stream_kernel(DecStreamArgs p) {
  using namespace dstream;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sc_search &sb = p.sb;
  if ((int)blockIdx.x >= (sb.rowmap ? sb.n_rows / sb.W : sb.S)) return;
  const int s = sb.rowmap ? sb.rowmap[blockIdx.x * sb.W] / sb.W : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = wave >> 1, sub = wave & 1;   // a head's two waves: K quarters / attention slots {sub, sub + 2}
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
  float *S01 = smem + O_U + head * WM * LDP;
  float *pm = smem + O_U + head * AP, *pl = pm + NPS * 16, *pO = pl + NPS * 16;
  const int r = lane & 15, kk = lane >> 4;

  SC_STAMP(0, 0);
  if (tid < 16) ancs[tid] = ANC(cur, s)[(long)(L - 1) * W + min(tid, nh - 1)];
  // LayerNorm parameters of the three norms -> LDS (requested first, parked behind the partial sums)
  float4 gbv[2];
  {
    const float *src[6] = {p.ln1_g, p.ln1_b, p.ln2_g, p.ln2_b, p.ln3_g, p.ln3_b};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + NTH * q;   // float4 pieces of [6][D]: 384
      gbv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 6 * C4) gbv[q] = *reinterpret_cast<const float4 *>(src[e / C4] + 4 * (e % C4));
    }
  }
  // ancestor slots of the first 512 positions: the row list of the self-attention is built while the partial sums travel
  int slp[WM] = {};
  {
    const int *anc0 = ANC(cur, s);
    const bool live0 = tid < PCS && tid < L - 1;
#pragma unroll
    for (int h = 0; h < WM; ++h) slp[h] = anc0[(long)(live0 ? tid : 0) * W + min(h, nh - 1)];
  }
  int U0 = 0;

  // ------------------------------------------------------------------ prologue: x rows (thread tid owns the float4 piece
  // (row tid / 64, columns 4 (tid % 64)) of the 16 x D tile for the whole kernel: the residual stays in its registers)
  const int xi_ = tid / C4, xc4 = tid % C4;
  float4 xres = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FIRST) {
    if (xi_ < W) {
      const int tok = YSEQ(cur, s, min(xi_, nh - 1))[L - 1];
      const float sq = sqrtf((float)D);
      const float4 ev = *reinterpret_cast<const float4 *>(sb.embed + (long)tok * D + 4 * xc4);
      const float4 pe = *reinterpret_cast<const float4 *>(sb.pe + (long)(L - 1) * D + 4 * xc4);
      xres = make_float4(ev.x * sq + pe.x, ev.y * sq + pe.y, ev.z * sq + pe.z, ev.w * sq + pe.w);
    }
    U0 = mattn_build_rows<WM, true, NTH, PCS>(srows, swtot, ANC(cur, s), 0, L - 1, W, nh, tid, lane, wave, slp);
  } else {
    const long row = (long)s * W + min(xi_, W - 1);
    const bool live = xi_ < W;   // (the pieces of the padding rows request nothing)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xi = zero4, pbv = zero4;
    if (live) xi = *reinterpret_cast<const float4 *>(p.xin + row * D + 4 * xc4);
    if (live && p.pbias) pbv = *reinterpret_cast<const float4 *>(p.pbias + 4 * xc4);
    float4 yv = zero4;
    for (int z0 = 0; z0 < p.npart; z0 += 8) {
      float4 pv[8];
#pragma unroll
      for (int z = 0; z < 8; ++z) {
        pv[z] = zero4;
        if (live && z0 + z < p.npart) pv[z] = *reinterpret_cast<const float4 *>(p.part + (long)(z0 + z) * p.zs + row * D + 4 * xc4);
      }
      if (z0 == 0) U0 = mattn_build_rows<WM, true, NTH, PCS>(srows, swtot, ANC(cur, s), 0, L - 1, W, nh, tid, lane, wave, slp);
      const float4 h8 = sc_tree8(pv);   // common.h: ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)), batches added in order
      yv = z0 == 0 ? h8 : sc_add4(yv, h8);
    }
    if (xi_ < W)
      xres = make_float4(xi.x + (yv.x + pbv.x), xi.y + (yv.y + pbv.y), xi.z + (yv.z + pbv.z), xi.w + (yv.w + pbv.w));
  }
  *reinterpret_cast<float4 *>(Xn + xi_ * LDX + 4 * xc4) = xres;   // rows >= W: zeros
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + NTH * q;
    if (e < 6 * C4) *reinterpret_cast<float4 *>(gb + 4 * e) = gbv[q];
  }
  // pre-zero what the MFMA tiles pad with: query rows >= WM of every head, context rows >= WM
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

  // ------------------------------------------------------------------ projection of this head's columns
  // NTP tiles of 16 output columns per pass; this wave takes the K quarters 2 sub and 2 sub + 1 (k blocks 4 sub .. 4 sub + 3),
  // each quarter its own MFMA chain from zero.  B operands straight from the fragment-packed weights.
  //   tile index of column tile t of the head: tile0 + t for the q-only projection; ((t / 2) * D + head * DK) / 16 + t % 2 for q|k|v
  auto proj_pass = [&](const float *wp, auto tile_of, auto &accA, auto &accB) {
    constexpr int NTP = sizeof(accA) / sizeof(accA[0]);
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      accA[t] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
      accB[t] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float4 b0[NTP], b1[NTP];
    auto load_b = [&](int ki, float4 (&x0)[NTP], float4 (&x1)[NTP]) {
#pragma unroll
      for (int t = 0; t < NTP; ++t) {
        const float4 *wq = reinterpret_cast<const float4 *>(wp) + ((long)tile_of(t) * KI + ki) * 128 + lane;
        x0[t] = wq[0];
        x1[t] = wq[64];
      }
    };
    load_b(4 * sub, b0, b1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int ki = 4 * sub + st;
      const float *ab = Xn + r * LDX + ki * 32 + 8 * kk;
      const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
      float4 n0[NTP], n1[NTP];
      if (st + 1 < 4) load_b(ki + 1, n0, n1);   // in flight during this k block's MFMAs
      if (st < 2) ds_mfma8_il<NTP>(accA, a0, a1, b0, b1);
      else ds_mfma8_il<NTP>(accB, a0, a1, b0, b1);
      if (st + 1 < 4) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          b0[t] = n0[t];
          b1[t] = n1[t];
        }
      }
    }
  };

  const long skv0 = ((long)s * sb.n_layers + p.li) * sb.kv_rows * 2 * D + head * DK;
  const long ckv0 = ((long)s * sb.n_layers + p.li) * sb.TCAP * 2 * D + head * DK;
  {
    // q|k|v: two passes of three column tiles (q0 q1 k0 | k1 v0 v1); sub 0 parks P0 + P1, sub 1 finishes from its registers
    ds_f32x4 a0A[3], a0B[3], a1A[3], a1B[3];
    proj_pass(p.wqkv, [&](int t) { return ((t / 2) * D + head * DK) / 16 + (t % 2); }, a0A, a0B);
    proj_pass(p.wqkv, [&](int t) { return (((t + 3) / 2) * D + head * DK) / 16 + ((t + 3) % 2); }, a1A, a1B);
    if (sub == 0) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * kk + j < WM) {
            const float va = t < 3 ? a0A[t % 3][j] : a1A[t % 3][j], vb = t < 3 ? a0B[t % 3][j] : a1B[t % 3][j];
            S01[(4 * kk + j) * LDP + t * 16 + r] = va + vb;   // P0 + P1
          }
    }
    __syncthreads();
    SC_STAMP(0, 3);
    if (sub == 1) {
      const float scale = sqrtf((float)DK);
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int which = t / 2, c = (t % 2) * 16 + r;
        const float bias = p.bqkv[which * D + head * DK + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int w = 4 * kk + j;
          if (w < WM) {
            const float va = t < 3 ? a0A[t % 3][j] : a1A[t % 3][j], vb = t < 3 ? a0B[t % 3][j] : a1B[t % 3][j];
            float v = 0.f;
            if (w < W) {
              v = S01[w * LDP + t * 16 + r];
              v += va;   // + P2
              v += vb;   // + P3
              v += bias;
            }
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
  }
  __syncthreads();
  SC_STAMP(0, 4);

  // ------------------------------------------------------------------ attention walks: this wave carries the partial states
  // of slots sub and sub + 2 of its head (slot w = tiles w, w + 4, ... of every list, batches of two tiles: attn.h)
  MAttn<DK> stA, stB;
  auto walk2 = [&](const float *kv, int ntiles, auto rowfn) {
    float qb[DK / 4];
#pragma unroll
    for (int i = 0; i < DK / 4; ++i) qb[i] = qs[(lane & 15) * DK + (DK / 4) * (lane >> 4) + i];
    for (int t0 = 0; t0 < ntiles; t0 += 8) {
      const int ta = t0 + sub, tb = t0 + sub + 2;
      if (ta < ntiles) {
        MBatch<DK, 2> b;
        mattn_load<DK, 2, KVH>(b, kv, D, ntiles, ta, lane, rowfn);
        mattn_batch<DK, 2>(stA, b, qb, lane);
      }
      if (tb < ntiles) {
        MBatch<DK, 2> b;
        mattn_load<DK, 2, KVH>(b, kv, D, ntiles, tb, lane, rowfn);
        mattn_batch<DK, 2>(stB, b, qb, lane);
      }
    }
  };
  // merge of a head's partial states -> its context tile [16][LDC] (rows >= nh zero); the waves' states are in LDS
  auto merge_ctx = [&](auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    for (int e = tid; e < H * WM * DK; e += NTH) {
      const int hh = e / (WM * DK), rem = e % (WM * DK), h = rem / DK, c = rem % DK;
      const float *bpm = smem + O_U + hh * AP;
      ctxs[(hh * 16 + h) * LDC + c] = h < nh ? ds_final<NP>(bpm, bpm + NPS * 16, bpm + 2 * NPS * 16, h, c) : 0.f;
    }
  };
  // output projection of ALL heads' contexts: wave w owns column tile w; one chain per head from zero, heads added in
  // aligned groups of four in head order, the groups in order; y -> Xn tile; then x <- x + (y + bias) by the piece owners
  auto out_proj = [&](const float *wop, const float *bias) {
    const float4 bv = *reinterpret_cast<const float4 *>(bias + 4 * xc4);
    ds_f32x4 y;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 b0[4], b1[4], a0[4], a1[4];
      ds_f32x4 acc[4];
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const int hd = half * 4 + hh;
        const float4 *wq = reinterpret_cast<const float4 *>(wop) + ((long)wave * KI + hd) * 128 + lane;
        b0[hh] = wq[0];
        b1[hh] = wq[64];
        const float *ab = ctxs + (hd * 16 + r) * LDC + 8 * kk;
        a0[hh] = *reinterpret_cast<const float4 *>(ab);
        a1[hh] = *reinterpret_cast<const float4 *>(ab + 4);
        acc[hh] = ds_f32x4{0.f, 0.f, 0.f, 0.f};
      }
      ds_mfma8_heads<4>(acc, a0, a1, b0, b1);
      ds_f32x4 g = acc[0];
      g += acc[1];
      g += acc[2];
      g += acc[3];
      if (half == 0) y = g;
      else y += g;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * kk + j < WM) Xn[(4 * kk + j) * LDX + wave * 16 + r] = y[j];
    __syncthreads();
    if (xi_ < W) {
      const float4 yv = *reinterpret_cast<const float4 *>(Xn + xi_ * LDX + 4 * xc4);
      xres = make_float4(xres.x + (yv.x + bv.x), xres.y + (yv.y + bv.y), xres.z + (yv.z + bv.z), xres.w + (yv.w + bv.w));
    }
    *reinterpret_cast<float4 *>(Xn + xi_ * LDX + 4 * xc4) = xres;   // (a piece of y is read by its owner only) rows >= W: zeros
    __syncthreads();
  };

  // ---- self-attention
  mattn_init(stA);
  mattn_init(stB);
  {
    const int *anc = ANC(cur, s);
    const int Lc = L - 1;   // cached positions; the new token's own row is the fifth partial state
    int urows = nh + U0;
    auto rowfn = [&](int idx, long &ke, unsigned &hm) {
      const int e = srows[min(idx, PCS * W - 1)];   // entries >= U are zero: no hypothesis
      hm = (unsigned)e >> 16;
      ke = skv0 + (long)(e & 0xFFFF) * 2 * D;
    };
    walk2(sb.skv, cdiv(U0, 16), rowfn);
    for (int c0 = PCS; c0 < Lc; c0 += PCS) {
      __syncthreads();   // the list is rebuilt for the next positions
      const int U = mattn_build_rows<WM, false, NTH, PCS>(srows, swtot, anc, c0, Lc, W, nh, tid, lane, wave, slp);
      urows += U;
      walk2(sb.skv, cdiv(U, 16), rowfn);
    }
    if (sb.stat_rows && tid == 0) atomicAdd(&sb.stat_rows[1], (unsigned long long)urows);
    // the new token: hypothesis h attends to its own row (slot h at position L-1, still in LDS) only
    if (sub == 0 && lane < 16) {
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
  ds_store_partial(stA, pm, pl, pO, sub, lane);
  ds_store_partial(stB, pm, pl, pO, sub + 2, lane);
  __syncthreads();   // (also: every wave is done with kvn - the context tiles take its place)
  merge_ctx(std::integral_constant<int, 5>{});
  __syncthreads();
  SC_STAMP(0, 6);
  out_proj(p.wo, p.bo);
  layer_norm(gb + 2 * D);
  SC_STAMP(0, 7);

  // ------------------------------------------------------------------ cross-attention: q of this head (two column tiles)
  {
    ds_f32x4 aA[2], aB[2];
    proj_pass(p.wq, [&](int t) { return (head * DK) / 16 + t; }, aA, aB);
    if (sub == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * kk + j < WM) S01[(4 * kk + j) * LDP + t * 16 + r] = aA[t][j] + aB[t][j];
    }
    __syncthreads();
    if (sub == 1) {
      const float scale = sqrtf((float)DK);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c = t * 16 + r;
        const float bias = p.bq[head * DK + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int w = 4 * kk + j;
          if (w < WM) {
            float v = 0.f;
            if (w < W) {
              v = S01[w * LDP + t * 16 + r];
              v += aA[t][j];
              v += aB[t][j];
              v += bias;
            }
            qs[w * DK + c] = v / scale;
          }
        }
      }
    }
  }
  __syncthreads();
  SC_STAMP(0, 8);
  mattn_init(stA);
  mattn_init(stB);
  {
    const unsigned all = (1u << nh) - 1u;
    walk2(sb.ckv, cdiv(T, 16), [&](int idx, long &ke, unsigned &hm) {
      hm = idx < T ? all : 0u;
      ke = ckv0 + (long)min(idx, T - 1) * 2 * D;
    });
  }
  SC_STAMP(0, 9);
  ds_store_partial(stA, pm, pl, pO, sub, lane);
  ds_store_partial(stB, pm, pl, pO, sub + 2, lane);
  __syncthreads();
  merge_ctx(std::integral_constant<int, 4>{});
  __syncthreads();
  SC_STAMP(0, 10);
  out_proj(p.wo2, p.bo2);
  // x'' -> the residual stream; LayerNorm3(x'') -> the feed-forward's input rows
  if (xi_ < W) *reinterpret_cast<float4 *>(p.xout + ((long)s * W + xi_) * D + 4 * xc4) = xres;
  layer_norm(gb + 4 * D);
  if (xi_ < W)
    *reinterpret_cast<float4 *>(p.xn + ((long)s * W + xi_) * D + 4 * xc4) = *reinterpret_cast<const float4 *>(Xn + xi_ * LDX + 4 * xc4);
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
  if ((long)dstream::PCS * sb.W > (long)dstream::PCS * dstream::WM) return 0;
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
