// Search-side kernels of libscasr (gfx950): decoder attention with K/V caches,
// log-softmax + pre-beam top-K, CTC prefix scan, score fusion, beam pruning.
// Every kernel is batched over S streams x W hypotheses and masked by the
// per-stream ctrl rows (see scasr.h), so one launch serves all streams.
#include "common.h"
#include "attn.h"
#include <mutex>
#include <unordered_map>

#define CTRL(s, f) sb.ctrl[(s) * 8 + (f)]
// rows of the CTC table / forward variables seen by a decode step (scasr.h: SC_C_TCTC)
#define SC_CTC_T(s) (CTRL(s, SC_C_TCTC) > 0 ? CTRL(s, SC_C_TCTC) : CTRL(s, SC_C_T))

// per-index helpers for the ping-pong buffers
#define YSEQ(pp, s, h) (sb.yseq + (((long)(pp) * sb.S + (s)) * sb.W + (h)) * sb.LCAP)
#define XPOS(pp, s, h) (sb.xpos + (((long)(pp) * sb.S + (s)) * sb.W + (h)) * sb.LCAP)
#define ANC(pp, s) (sb.anc + ((long)(pp) * sb.S + (s)) * sb.LCAP * sb.W)
#define CTCR(pp, s) (sb.ctc_r + ((long)(pp) * sb.S + (s)) * sb.TCAP * 2 * sb.W)
#define CTCRS(pp, s) (sb.ctc_rs + ((long)(pp) * sb.S + (s)) * sb.TCAP * sb.W)   // log(exp r^n + exp r^b) per frame (scasr.h)

// ---------------------------------------------------------------------------
// CTC state extension: r^n[t] = logzero, r^b[t] = r^b[t-1] + x[t, blank]
// ---------------------------------------------------------------------------
__global__ void ctc_extend_state_kernel(sc_search sb) {
  const int s = blockIdx.x, h = threadIdx.x;
  if (!CTRL(s, SC_C_ACTIVE) || !CTRL(s, SC_C_HAS)) return;
  const int T = CTRL(s, SC_C_T), told = CTRL(s, SC_C_TOLD), nh = CTRL(s, SC_C_NHYP);
  if (h >= nh || told >= T) return;
  float *r = CTCR(CTRL(s, SC_C_CUR), s);
  float *rs = CTCRS(CTRL(s, SC_C_CUR), s);
  const float *x = sb.ctcx + (long)s * sb.TCAP * sb.V;
  const int t0 = told < 1 ? 1 : told;
  float rb = r[((long)(t0 - 1) * 2 + 1) * sb.W + h];
  for (int t = t0; t < T; ++t) {
    rb = rb + x[(long)t * sb.V + sb.blank];
    r[((long)t * 2 + 0) * sb.W + h] = SC_LOGZERO;
    r[((long)t * 2 + 1) * sb.W + h] = rb;
    rs[(long)t * sb.W + h] = rb;   // lse2(logzero, rb) == rb exactly: exp2 of -1e10 is 0, log of 1 is 0
  }
}

// Column-major copy of the new table rows: ctcxT[s][v][t] = ctcx[s][t][v] for t in [told, T).  The prefix scan
// walks ONE column per lane over all frames: from the row-major table that is a 4-byte gather with a 4 KB stride
// (one cache line per lane and frame); from the transposed copy every lane streams its own contiguous column
// with 16-byte loads.  32 x 32 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void ctc_table_transpose_kernel(sc_search sb) {
  __shared__ float tile[32][33];
  const int s = blockIdx.z;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int T = CTRL(s, SC_C_T), told = CTRL(s, SC_C_TOLD);
  if (told >= T) return;
  const int V = sb.V, v0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float *x = sb.ctcx + (long)s * sb.TCAP * V;
  float *xt = sb.ctcxT + (long)s * V * sb.tct;
  for (int t0 = (told & ~31) + blockIdx.y * 32; t0 < T; t0 += gridDim.y * 32) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = t0 + ty + 8 * j, v = v0 + tx;
      tile[ty + 8 * j][tx] = (t < T && v < V) ? x[(long)t * V + v] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = v0 + ty + 8 * j, t = t0 + tx;
      if (v < V && t >= told && t < T) xt[(long)v * sb.tct + t] = tile[tx][ty + 8 * j];
    }
    __syncthreads();
  }
}

extern "C" int sc_ctc_extend_state(const sc_search *sbp, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  if (sbp->ctcxT)
    ctc_table_transpose_kernel<<<dim3(cdiv(sbp->V, 32), 4, sbp->S), 256, 0, (hipStream_t)stream>>>(*sbp);
  ctc_extend_state_kernel<<<sbp->S, 64, 0, (hipStream_t)stream>>>(*sbp);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// Cross-attention K|V rows of all layers, fp32 staging [n_layers][m][2d] -> fp16 cache rows (kv_half): row j of
// layer li goes to cache row rows[j] + li*TCAP (the row table of the projection GEMMs).
__global__ void kv_rows_to_half_kernel(const float *stage, const int32_t *rows, int m, int TCAP, int d2, _Float16 *ckv) {
  const int li = blockIdx.y;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)m * d2; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e / d2), c = (int)(e % d2);
    ckv[((long)rows[j] + (long)li * TCAP) * d2 + c] = (_Float16)stage[((long)li * m + j) * d2 + c];
  }
}

extern "C" int sc_kv_rows_to_half(const float *stage, const int32_t *rows, int m, int n_layers, int TCAP, int d2,
                                  void *ckv_half, void *stream) {
  SC_CHECK_ARG(stage && rows && ckv_half && m >= 0 && n_layers > 0 && d2 > 0, "bad arguments");
  if (m == 0) return SC_OK;
  const int gx = (int)(((long)m * d2 + 255) / 256 > 1024 ? 1024 : ((long)m * d2 + 255) / 256);
  kv_rows_to_half_kernel<<<dim3(gx, n_layers), 256, 0, (hipStream_t)stream>>>(stage, rows, m, TCAP, d2, (_Float16 *)ckv_half);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// Self-attention K|V pool (scasr.h: sc_search.skv / anc): rows are handed out by beam_prune_kernel below - the rows of the
// new hypotheses' newest tokens are chosen when the hypotheses are made - so a decode step starts with its rows in
// place and no launch of its own.
#define SC_KV_MAX_ROWS 65536
// ---------------------------------------------------------------------------
__global__ void dec_embed_kernel(sc_search sb, float sq) {
  const int row = blockIdx.x, s = row / sb.W, h = row % sb.W;
  if (!CTRL(s, SC_C_ACTIVE) || h >= CTRL(s, SC_C_NHYP)) return;
  const int L = CTRL(s, SC_C_L);
  const int tok = YSEQ(CTRL(s, SC_C_CUR), s, h)[L - 1];
  for (int c = threadIdx.x; c < sb.d; c += blockDim.x)
    sb.dx[(long)row * sb.d + c] = sb.embed[(long)tok * sb.d + c] * sq + sb.pe[(long)(L - 1) * sb.d + c];
}

extern "C" int sc_dec_embed(const sc_search *sbp, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  dec_embed_kernel<<<sbp->S * sbp->W, 256, 0, (hipStream_t)stream>>>(*sbp, sqrtf((float)sbp->d));
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// Single-pass ("flash decoding") attention of the decoder, shared by the self-
// and the cross-attention: one workgroup per (stream, head) serves ALL
// hypotheses of the stream.
//   * rows: SELF  - the DISTINCT (position, slot) K/V rows referenced through
//                   the ancestor table (hypotheses of a beam share almost all
//                   of their history: 1.2 distinct rows per position for 10
//                   hypotheses on the XL fixture), compacted per 128-position
//                   chunk into an LDS list with the bit set of hypotheses
//                   using each row;
//           CROSS - the T encoder frames, used by every hypothesis.
//   * arithmetic on the matrix cores (attn.h: mattn_walk): tiles of 16 K/V rows x all hypotheses,
//     S^T = K.Q^T and O^T += V^T.P with v_mfma_f32_16x16x4_f32; the four waves take tiles round-robin,
//     K rows are loaded 32 B per lane (full lines per 4 lanes), V rows 8 B per lane (full lines per 16 lanes);
//   * online softmax per batch of NTW tiles (running maximum per hypothesis = per lane), the waves' partial
//     states and - SELF - the new token's own row are merged through LDS.
// Scores use q/sqrt(dk) . k and exp2-based __expf: differences to the reference's
// softmax(q.k/sqrt(dk)) are rounding-level (parity tests: 2e-4).
// ---------------------------------------------------------------------------
// UNR selects the batch depth: 8 -> 4 tiles per wave in flight (at most half of the streams active: occupancy is
// no issue, serial HBM round trips are), otherwise 2 (all S*H workgroups of a full batch resident at once).
template <int DK, int WM, bool SELF, int UNR, bool KVH = false>
__global__ __launch_bounds__(256, (UNR <= 4 && WM <= 10 && DK <= 32) ? 4 : 1) void dec_attn_flash_kernel(sc_search sb, int li) {
  constexpr int PCH = 128;       // positions per chunk of the row list (SELF)
  constexpr int NTW = (UNR >= 8) ? 4 : 2;   // key tiles per wave and batch (attn.h: mattn_walk)
  constexpr int NP = SELF ? 5 : 4;          // partial states: one per wave (+ the new token's own row)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // grid.y runs over the compaction bucket: the k-th stream of rowmap's active-first order (scasr.h: rowmap)
  const int head = blockIdx.x, s = sb.rowmap ? sb.rowmap[blockIdx.y * sb.W] / sb.W : blockIdx.y;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int nh = CTRL(s, SC_C_NHYP);
  if (nh <= 0) return;
  const int L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR), T = CTRL(s, SC_C_T);
  const int W = sb.W, d = sb.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // LDS: partial states pm / pl [NP][16], pO [NP*16][DK+1]; rows[PCH*W]; wtot[8]; qs [16][DK]
  float *pm = smem;
  float *pl = pm + NP * 16;
  float *pO = pl + NP * 16;
  int *rows = (int *)(smem + mattn_partial_floats(DK, NP));
  int *wtot = rows + (SELF ? PCH * W : 0);
  float *qs = (float *)(wtot + 8);   // queries / sqrt(dk), rows >= nh zero

  // SELF: q|k|v of the new token in dqkv (hypothesis h at + h*3d); CROSS: q in dq
  const float *qbase = SELF ? sb.dqkv + (long)s * W * 3 * d + head * DK : sb.dq + (long)s * W * d + head * DK;
  const int qld = SELF ? 3 * d : d;
  // element offsets of this (stream, layer, head) in the K|V caches (fp32, or fp16 when KVH: same offsets)
  const long skv0 = ((long)s * sb.n_layers + li) * sb.kv_rows * 2 * d + head * DK;
  const long ckv0 = ((long)s * sb.n_layers + li) * sb.TCAP * 2 * d + head * DK;
  if (SELF) {
    // append this token's K|V rows into the pool rows beam_prune_kernel gave the hypotheses (anc[L-1][h]); this launch
    // reads them from dqkv, later steps from the cache
    const int *ancn = ANC(cur, s) + (long)(L - 1) * W;
    for (int e = tid; e < nh * DK; e += 256) {
      const int h = e / DK, c = e % DK;
      const long dst = skv0 + (long)ancn[h] * 2 * d;
      kv_store1<KVH>(sb.skv, dst + c, qbase[(long)h * 3 * d + d + c]);
      kv_store1<KVH>(sb.skv, dst + d + c, qbase[(long)h * 3 * d + 2 * d + c]);
    }
  }
  // queries of the hypotheses, pre-divided by sqrt(dk); the rows that pad the MFMA tile are zero
  const float scale = sqrtf((float)DK);
  for (int e = tid; e < 16 * DK; e += 256) {
    const int h = e / DK, c = e % DK;
    qs[e] = h < nh ? qbase[(long)h * qld + c] / scale : 0.f;
  }
  MAttn<DK> st;
  mattn_init(st);

  if (SELF) {
    const int *anc = ANC(cur, s);
    const int Lc = L - 1;   // cached positions; the new token's row (still in dqkv) is the fifth partial state
    const int nchunk = cdiv(Lc, PCH);
    if (nchunk == 0) __syncthreads();   // qs
    int urows = nh;   // distinct K|V rows of this (stream, layer): the new tokens' rows + the walked ones
    for (int ch = 0; ch < nchunk; ++ch) {
      const int c0 = ch * PCH;
      const int nopre[WM] = {};
      const int U = mattn_build_rows<WM>(rows, wtot, anc, c0, Lc, W, nh, tid, lane, wave, nopre);   // (attn.h)
      urows += U;
      mattn_walk<DK, NTW, KVH>(st, qs, sb.skv, d, cdiv(U, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
        const int e = rows[min(idx, PCH * W - 1)];   // entries >= U are zero: no hypothesis
        hm = (unsigned)e >> 16;
        ke = skv0 + (long)(e & 0xFFFF) * 2 * d;
      });
      if (ch + 1 < nchunk) __syncthreads();  // rows is rebuilt by the next chunk
    }
    if (sb.stat_rows && head == 0 && tid == 0) atomicAdd(&sb.stat_rows[0], (unsigned long long)urows);
    // the new token: hypothesis h attends to its own row (position L-1, still in dqkv) only
    if (tid < 16) {
      const int h = tid;
      float sdot = -INFINITY;
      if (h < nh) {
        const float *kp = qbase + (long)h * 3 * d + d;
        sdot = 0.f;
#pragma unroll
        for (int c = 0; c < DK; ++c) sdot = fmaf(qs[h * DK + c], kp[c], sdot);
#pragma unroll
        for (int c = 0; c < DK; ++c) pO[(4 * 16 + h) * (DK + 1) + c] = kp[d + c];
      }
      pm[4 * 16 + h] = sdot;
      pl[4 * 16 + h] = h < nh ? 1.f : 0.f;
    }
  } else {
    __syncthreads();   // qs
    const unsigned all = (1u << nh) - 1u;
    mattn_walk<DK, NTW, KVH>(st, qs, sb.ckv, d, cdiv(T, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
      hm = idx < T ? all : 0u;
      ke = ckv0 + (long)min(idx, T - 1) * 2 * d;
    });
  }
  mattn_store_partial<DK>(st, pm, pl, pO, wave, lane);
  __syncthreads();
  for (int e = tid; e < nh * DK; e += 256) {
    const int h = e / DK, c = e % DK;
    sb.datt[((long)s * W + h) * d + head * DK + c] = mattn_final<DK, NP>(pm, pl, pO, h, c);
  }
}

static size_t attn_flash_lds(const sc_search &sb, int dk, bool self) {
  return ((size_t)mattn_partial_floats(dk, self ? 5 : 4) + (self ? (size_t)128 * sb.W : 0) + 8 + (size_t)16 * dk) * sizeof(float);
}

template <int DK, bool SELF, bool KVH>
static void launch_attn_flash_kvh(const sc_search &sb, int layer, hipStream_t st) {
  const dim3 grid(sb.H, sb.rowmap ? sb.n_rows / sb.W : sb.S);   // streams of the compaction bucket only
  const size_t lds = attn_flash_lds(sb, DK, SELF);
  // compaction bucket of this launch (scasr.h: rowmap / n_rows): at most half of the streams active
  bool deep = (sb.rowmap && 2 * sb.n_rows <= sb.S * sb.W) || sb.S * sb.H <= 256;
  if (const char *fd = sc_hook("SC_ATTN_DEEP")) deep = atoi(fd) != 0;   // tests: force either variant at any size
  if (sb.W <= 5) dec_attn_flash_kernel<DK, 5, SELF, 4, KVH><<<grid, 256, lds, st>>>(sb, layer);
  else if (sb.W <= 10) {
    if (deep) dec_attn_flash_kernel<DK, 10, SELF, 8, KVH><<<grid, 256, lds, st>>>(sb, layer);
    else dec_attn_flash_kernel<DK, 10, SELF, 2, KVH><<<grid, 256, lds, st>>>(sb, layer);
  } else dec_attn_flash_kernel<DK, 16, SELF, 2, KVH><<<grid, 256, lds, st>>>(sb, layer);
}

template <int DK, bool SELF>
static void launch_attn_flash(const sc_search &sb, int layer, hipStream_t st) {
  if (sb.kv_half) launch_attn_flash_kvh<DK, SELF, true>(sb, layer, st);
  else launch_attn_flash_kvh<DK, SELF, false>(sb, layer, st);
}

extern "C" int sc_dec_self_attn(const sc_search *sbp, int layer, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  const sc_search &sb = *sbp;
  const int dk = sb.d / sb.H;
  hipStream_t st = (hipStream_t)stream;
  SC_CHECK_ARG(sb.W <= 16 && (dk == 64 || dk == 32 || dk == 16) && attn_flash_lds(sb, dk, true) <= 64 * 1024,
               "unsupported head dim / beam width");
  ProfScope prof = sc_prof_begin(st);
  if (dk == 64) launch_attn_flash<64, true>(sb, layer, st);
  else if (dk == 32) launch_attn_flash<32, true>(sb, layer, st);
  else launch_attn_flash<16, true>(sb, layer, st);
  sc_prof_end(prof, SC_PROF_ATTN_SELF, 0.0, 0.0);   // traffic depends on device-side state (L, ancestors)
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// decoder cross-attention over the stream's shared encoder K|V: one workgroup per (stream, head) walks all T frames
// once for all hypotheses (dec_attn_flash_kernel<.., SELF = false>); also the faster form for few streams (measured:
// 1 / 8 / 16 streams 58.3 / 266.9 / 452.5 vs 58.3 / 260.8 / 433.6 audio-s/s for a split-T + merge pair of launches)
extern "C" int sc_dec_cross_attn(const sc_search *sbp, int layer, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  const sc_search &sb = *sbp;
  const int dk = sb.d / sb.H;
  hipStream_t st = (hipStream_t)stream;
  SC_CHECK_ARG(sb.W <= 16 && (dk == 64 || dk == 32 || dk == 16) && attn_flash_lds(sb, dk, false) <= 64 * 1024,
               "unsupported head dim / beam width");
  ProfScope prof = sc_prof_begin(st);
  if (dk == 64) launch_attn_flash<64, false>(sb, layer, st);
  else if (dk == 32) launch_attn_flash<32, false>(sb, layer, st);
  else launch_attn_flash<16, false>(sb, layer, st);
  sc_prof_end(prof, SC_PROF_ATTN_CROSS, 0.0, 0.0);  // bytes = sum_s T_s * 2d * 4: the host knows T (bench.py)
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// fuse_logits: the last layer's reduce kernel projects the output layer too (sb.logits is
// written here, the after_norm rows are not stored); returns that through *logits_done.
static int decoder_layers_impl(const sc_search *sbp, void *stream, bool fuse_logits, bool *logits_done) {
  SC_CHECK_ARG(sbp && sbp->layers, "null");
  const sc_search &sb = *sbp;
  const int d = sb.d, F = sb.F;
  if (logits_done) *logits_done = false;
  // dense kernels run over the compacted rows of the active streams (scasr.h: rowmap)
  const int32_t *rows = sb.rowmap;
  const int n = rows ? sb.n_rows : sb.S * sb.W;
  SC_CHECK_ARG(n > 0 && n <= sb.S * sb.W, "n_rows out of range");
  const int lnf = rows ? SC_GEMM_LN_AT_CROWS : 0;
  // SC_DEC_PANEL=0 keeps the three-launch form (GEMM, reduce+LN, GEMM) for A/B runs
  const char *pe = sc_hook("SC_DEC_PANEL");
  const bool panel_env = !(pe && atoi(pe) == 0);
  const bool panel = panel_env && sc_proj_ln_proj_supported(d);
  const char *fe = sc_hook("SC_FFN_FUSED");      // =0: two GEMMs with the hidden activations in HBM
  const bool ffn_fused = !(fe && atoi(fe) == 0) && sc_ffn_ln_supported(d, F) &&
                         sc_workspace_bytes(stream) >= (size_t)(F / 128) * 80 * d * sizeof(float);
  // the FFN's reduce kernel also projects what consumes its LayerNorm (next layer's Q|K|V, output
  // layer): needs the row panels, the fused FFN, lane-packed weights and a workspace for all rows.
  // x then ping-pongs between dx and dxn (the reduce kernel must not update x in place), and the
  // LayerNorm before the FFN goes to dq (free after the cross-attention).
  const char *qe = sc_hook("SC_FFN_PROJ");
  const bool chain = !(qe && atoi(qe) == 0) && panel && ffn_fused && sb.layers[0].wqkv_q &&
                     sc_workspace_bytes(stream) >= (size_t)(F / 128) * n * d * sizeof(float);
  int rc;
#define SC_TRY(call) do { rc = (call); if (rc != SC_OK) return rc; } while (0)
  float *x = sb.dx, *xalt = sb.dxn;
  float *ffn_in = chain ? sb.dq : sb.dxn;
  // LN1 of layer 0 is the only stand-alone LayerNorm; every other LayerNorm is
  // fused into the kernel that produces its input.
  {
    float *ln1 = chain ? sb.dq : sb.dxn;
    SC_TRY(sc_layernorm(sb.dx, rows, d, ln1, rows, d, n, d, sb.layers[0].ln1_g, sb.layers[0].ln1_b, sb.ln_eps, stream));
    SC_TRY(sc_gemm(ln1, rows, d, sb.layers[0].wqkv, sb.layers[0].bqkv, sb.dqkv, rows, 3 * d, n, 3 * d, d, 0, 0, stream));
  }
  for (int li = 0; li < sb.n_layers; ++li) {
    const sc_dec_layer &w = sb.layers[li];
    const bool last = li + 1 == sb.n_layers;
    const float *ng = last ? sb.dec_norm_g : sb.layers[li + 1].ln1_g;
    const float *nb = last ? sb.dec_norm_b : sb.layers[li + 1].ln1_b;
    SC_TRY(sc_dec_self_attn(sbp, li, stream));
    if (panel) {
      // out-projection + residual + norm2 + cross-attention query in one row-panel kernel
      SC_TRY(sc_proj_ln_proj(sb.datt, d, w.wo_p, w.bo, x, d, w.ln2_g, w.ln2_b, sb.ln_eps, nullptr, d,
                             w.wq_p, w.bq, sb.dq, d, rows, n, d, stream));
      SC_TRY(sc_dec_cross_attn(sbp, li, stream));
      SC_TRY(sc_proj_ln_proj(sb.datt, d, w.wo2_p, w.bo2, x, d, w.ln3_g, w.ln3_b, sb.ln_eps, ffn_in, d,
                             nullptr, nullptr, nullptr, d, rows, n, d, stream));
    } else {
      SC_TRY(sc_gemm_ln(sb.datt, rows, d, w.wo, w.bo, x, rows, d, n, d, d, SC_GEMM_RESIDUAL | lnf, 0,
                        w.ln2_g, w.ln2_b, sb.ln_eps, sb.dxn, d, stream));
      SC_TRY(sc_gemm(sb.dxn, rows, d, w.wq, w.bq, sb.dq, rows, d, n, d, d, 0, 0, stream));
      SC_TRY(sc_dec_cross_attn(sbp, li, stream));
      SC_TRY(sc_gemm_ln(sb.datt, rows, d, w.wo2, w.bo2, x, rows, d, n, d, d, SC_GEMM_RESIDUAL | lnf, 0,
                        w.ln3_g, w.ln3_b, sb.ln_eps, sb.dxn, d, stream));
    }
    // feed-forward + residual, then the NEXT layer's LN1 + Q|K|V (or after_norm + output layer)
    if (chain && (!last || (fuse_logits && sb.out_w_q && sb.V % d == 0))) {
      const bool wh = w.w1_h && w.w2_h;   // fp16 weights: fp16 MFMA inputs, fp32 accumulation
      const float *pw = last ? sb.out_w_q : sb.layers[li + 1].wqkv_q, *pb = last ? sb.out_b : sb.layers[li + 1].bqkv;
      float *pq = last ? sb.logits : sb.dqkv;
      const int pn = last ? sb.V : 3 * d;
      if (w.w1_s && w.w2_s)   // fp16 hi | lo split of the fp32 weights: fp32-grade on the fp16 matrix pipe
        SC_TRY(sc_ffn_ln_proj_s(ffn_in, rows, n, d, F, w.w1_s, w.b1, w.w2_s, w.b2, x, xalt, ng, nb, sb.ln_eps, nullptr, pw, pb,
                                pq, pn, stream));
      else if (wh)
        SC_TRY(sc_ffn_ln_proj_h(ffn_in, rows, n, d, F, w.w1_h, w.b1, w.w2_h, w.b2, x, xalt, ng, nb, sb.ln_eps, nullptr, pw, pb,
                                pq, pn, stream));
      else
        SC_TRY(sc_ffn_ln_proj(ffn_in, rows, n, d, F, w.w1_p, w.b1, w.w2_p, w.b2, x, xalt, ng, nb, sb.ln_eps, nullptr, pw, pb,
                              pq, pn, stream));
      if (last && logits_done) *logits_done = true;
      float *t = x; x = xalt; xalt = t;
      continue;
    }
    float *ln_next = chain ? (x == sb.dx ? sb.dxn : sb.dx) : sb.dxn;   // a buffer that is not x
    if (ffn_fused && w.w1_s && w.w2_s) {
      SC_TRY(sc_ffn_ln_s(ffn_in, rows, n, d, F, w.w1_s, w.b1, w.w2_s, w.b2, x, ng, nb, sb.ln_eps, ln_next, stream));
    } else if (ffn_fused && w.w1_h && w.w2_h) {
      SC_TRY(sc_ffn_ln_h(ffn_in, rows, n, d, F, w.w1_h, w.b1, w.w2_h, w.b2, x, ng, nb, sb.ln_eps, ln_next, stream));
    } else if (ffn_fused) {
      SC_TRY(sc_ffn_ln(ffn_in, rows, n, d, F, w.w1_p, w.b1, w.w2_p, w.b2, x, ng, nb, sb.ln_eps, ln_next, stream));
    } else {
      SC_TRY(sc_gemm(ffn_in, rows, d, w.w1, w.b1, sb.dffh, rows, F, n, F, d, SC_GEMM_RELU, 0, stream));
      SC_TRY(sc_gemm_ln(sb.dffh, rows, F, w.w2, w.b2, x, rows, d, n, d, F, SC_GEMM_RESIDUAL | lnf, 0,
                        ng, nb, sb.ln_eps, ln_next, d, stream));
    }
    if (!last) {
      SC_TRY(sc_gemm(ln_next, rows, d, sb.layers[li + 1].wqkv, sb.layers[li + 1].bqkv, sb.dqkv, rows, 3 * d, n, 3 * d, d,
                     0, 0, stream));
    } else if (ln_next != sb.dxn) {
      // contract of sc_decoder_layers: after_norm(x) in dxn
      SC_TRY(sc_copy_rows(ln_next, rows, sb.dxn, rows, n, d, stream));
    }
  }
  return SC_OK;
}

extern "C" int sc_decoder_layers(const sc_search *sbp, void *stream) {
  return decoder_layers_impl(sbp, stream, false, nullptr);   // leaves after_norm(x) in dxn
}

// ---------------------------------------------------------------------------
// Exact top-k shared by the pre-beam and the per-hypothesis top-W: bitonic sort
// (descending) of 64-bit composites (orderable(key) << 32 | ~index) in LDS.
// Descending composite order == key descending, ties towards the LOWEST index
// (deterministic; torch.topk leaves tie order unspecified).  55 stages for
// V = 1024 instead of the O(V^2) rank count (150 us -> ~10 us per row).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sort_key(float v, int idx) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
  return ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)idx);
}
__device__ __forceinline__ int sort_key_index(unsigned long long c) {
  return (int)(0xFFFFFFFFu - (unsigned)(c & 0xFFFFFFFFull));
}

// sorts comp[0..N) descending; N power of two; all threads of the block call it
__device__ void block_bitonic_sort_desc(unsigned long long *comp, int N) {
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = comp[i], b = comp[ixj];
          const bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) {
            comp[i] = b;
            comp[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ int next_pow2(int v) {
  int n = 1;
  while (n < v) n <<= 1;
  return n;
}

// ---- wave-level bitonic primitives: one 64-bit composite per lane, no LDS, no barriers ----
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const unsigned lo = __shfl_xor((unsigned)v, mask, 64), hi = __shfl_xor((unsigned)(v >> 32), mask, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
  return ((unsigned long long)hi << 32) | lo;
}
// sorts the 64 values of a wave descending (lane i ends with rank i)
__device__ __forceinline__ unsigned long long wave_sort64_desc(unsigned long long v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned long long o = shfl_xor_u64(v, j);
      const bool keep_max = ((lane & j) == 0) == ((lane & k) == 0);
      v = keep_max ? (v > o ? v : o) : (v < o ? v : o);
    }
  }
  return v;
}
// a, b sorted descending over the lanes -> the 64 largest of the union, sorted descending
__device__ __forceinline__ unsigned long long wave_merge64_desc(unsigned long long a, unsigned long long b, int lane) {
  const unsigned long long br = shfl_u64(b, 63 - lane);
  unsigned long long v = a > br ? a : br;   // bitonic sequence holding the top 64
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    const unsigned long long o = shfl_xor_u64(v, j);
    v = ((lane & j) == 0) ? (v > o ? v : o) : (v < o ? v : o);
  }
  return v;
}

// log-softmax + exact top-K (K <= 64, V <= 1024): every wave sorts its 4 x 64 keys
// in registers and merges them to its top 64; wave 0 merges the four wave results.
// Same composite keys as the full sort below, so the same ids in the same order.
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(sc_search sb) {
  __shared__ float red[8];
  __shared__ unsigned long long tops[4][64];
  const int row = blockIdx.x, s = row / sb.W, h = row % sb.W;
  if (!CTRL(s, SC_C_ACTIVE) || h >= CTRL(s, SC_C_NHYP)) return;
  const int V = sb.V, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *x = sb.logits + (long)row * V;
  float xv[4];
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = wave * 256 + q * 64 + lane;
    xv[q] = c < V ? x[c] : -INFINITY;
    m = fmaxf(m, xv[q]);
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) sum += (wave * 256 + q * 64 + lane < V) ? expf(xv[q] - m) : 0.f;
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float ls = logf(sum);
  unsigned long long key[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = wave * 256 + q * 64 + lane;
    if (c < V) {
      const float lp = (xv[q] - m) - ls;
      sb.logp[(long)row * V + c] = lp;
      key[q] = sort_key(__fmul_rn(sb.w_dec, lp), c);
    } else {
      key[q] = 0ull;  // below every real key
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) key[q] = wave_sort64_desc(key[q], lane);
  const unsigned long long m0 = wave_merge64_desc(key[0], key[1], lane);
  const unsigned long long m1 = wave_merge64_desc(key[2], key[3], lane);
  tops[wave][lane] = wave_merge64_desc(m0, m1, lane);
  __syncthreads();
  if (wave == 0) {
    const unsigned long long a = wave_merge64_desc(tops[0][lane], tops[1][lane], lane);
    const unsigned long long b = wave_merge64_desc(tops[2][lane], tops[3][lane], lane);
    const unsigned long long f = wave_merge64_desc(a, b, lane);
    if (lane < sb.K) sb.pre_ids[(long)row * sb.K + lane] = sort_key_index(f);
  }
}

// generic path (V > 1024 or K > 64): full bitonic sort in LDS
__global__ __launch_bounds__(256) void logsoftmax_topk_sort_kernel(sc_search sb) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long comp[];
  __shared__ float red[8];
  const int row = blockIdx.x, s = row / sb.W, h = row % sb.W;
  if (!CTRL(s, SC_C_ACTIVE) || h >= CTRL(s, SC_C_NHYP)) return;
  const int V = sb.V, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NP = next_pow2(V);
  const float *x = sb.logits + (long)row * V;
  float m = -INFINITY;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, x[c]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += expf(x[c] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float ls = logf(sum);
  for (int c = tid; c < NP; c += 256) {
    if (c < V) {
      float lp = (x[c] - m) - ls;
      sb.logp[(long)row * V + c] = lp;
      comp[c] = sort_key(__fmul_rn(sb.w_dec, lp), c);
    } else {
      comp[c] = 0ull;  // below every real key
    }
  }
  __syncthreads();
  block_bitonic_sort_desc(comp, NP);
  for (int k = tid; k < sb.K; k += 256) sb.pre_ids[(long)row * sb.K + k] = sort_key_index(comp[k]);
}

extern "C" int sc_logsoftmax_topk(const sc_search *sbp, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  int np = 1;
  while (np < sbp->V) np <<= 1;
  if (sbp->V <= 1024 && sbp->K <= 64)
    logsoftmax_topk_kernel<<<sbp->S * sbp->W, 256, 0, (hipStream_t)stream>>>(*sbp);
  else
    logsoftmax_topk_sort_kernel<<<sbp->S * sbp->W, 256, np * sizeof(unsigned long long), (hipStream_t)stream>>>(*sbp);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// CTC prefix scan (Watanabe Alg. 2): one lane per (hypothesis, candidate),
// sequential over encoder frames, log-domain, logzero = -1e10.
// ---------------------------------------------------------------------------
// The scan walks the column-major table copy (sb.ctcxT): every lane streams its candidate's column and the
// blank column with 16-byte loads, 16 frames (4 loads each) ahead of the recurrence, chunks aligned to 16 frames.
// What remains on the critical path is the recurrence itself: two log-add-exps per frame (~60 cycles), i.e.
// ~0.025 us per frame instead of the 0.11 us of gathers from the row-major table (the round-1 kernel: one cache
// line per lane and frame); 45 us at T = 400; T = 4500 is a 180 s segment of the CLI.
struct CtcChunkT {
  float4 xc[4], xb[4];
  float pn[16], pb[16];   // pn: log(exp r^n + exp r^b) of the prefix at t-1 (sc_search.ctc_rs; round 5 - was r^n), pb: its r^b
};
// Forward variables of the candidates are kept at CHECKPOINT frames only (t % 16 == 15): ctc_rnew [S][tck][2][W*K],
// tck = ceil(TCAP / 16).  The W x K candidates of a stream used to write r[t] for EVERY frame (3.2 KB per frame at beam
// 10: 47 MB per launch at 128 streams, 1.9 GB per step at T = 4500) although only the W winners' histories survive the
// step; ctc_gather_state_kernel now rebuilds the winners' full-resolution r[t] from the checkpoints - the same
// recurrence (ctc_frame below, bit for bit) over 16 frames per thread, all segments in parallel.
#define SC_CTC_CK 16
__host__ __device__ static inline int ctc_tck(int TCAP) { return (TCAP + SC_CTC_CK - 1) / SC_CTC_CK; }
// one frame of Watanabe Alg. 2 for one (hypothesis, candidate): (r_n, r_b) at t-1 -> t; pb = r^b of the PREFIX at t-1
// prs = log(exp r^n + exp r^b) of the PREFIX at t-1: the same for the K candidates of a hypothesis - it comes from
// sc_search.ctc_rs (round 5) instead of two transcendentals per frame and lane
__device__ __forceinline__ void ctc_frame(float &r_n, float &r_b, float prs, float pb, bool same, float xc, float xb, float &phi) {
  phi = same ? pb : prs;         // select, not a branch: lanes of a wave differ in `same`
  const float nr_n = lse2(r_n, phi) + xc;
  const float nr_b = lse2(r_n, r_b) + xb;
  r_n = nr_n;
  r_b = nr_b;
}
// running log-sum-exp (pm = maximum so far, ps = sum of exp(term - pm)) + one term v.  One of the two exponentials of the
// textbook update is exp(0) = 1: only the other one is evaluated (round 5: the scan is bound by its transcendentals -
// 8 per frame and lane until round 4, 7 with this)
__device__ __forceinline__ void ctc_psi_add(float &pm, float &ps, float v) {
  const float d = v - pm;
  const float e = sc_exp_neg(-fabsf(d));
  ps = d > 0.f ? ps * e + 1.f : ps + e;
  pm = sc_max_raw(pm, v);
}
// a stream's scan is split over T (ctc_prefix_scan_tpar_kernel) when it has at least split_min frames to walk
__device__ __forceinline__ bool ctc_scan_is_split(int T, int L, int split_min) {
  const int out_len = L - 1;
  int start = out_len > 1 ? out_len : 1;
  if (start > T) start = T;
  return split_min > 0 && T - start >= split_min;
}
#define CTC_NSEG 32   // segments per (hypothesis, candidate) pair of the T-parallel scan
// (the T-parallel form parks CTC_NSEG segment states per pair where the checkpoints live: the hosts switch it off for tables
// that short - ctc_split_min_ok)
static inline int ctc_split_min_ok(int TCAP, int split_min) { return ctc_tck(TCAP) >= CTC_NSEG ? split_min : 0; }

__global__ __launch_bounds__(256) void ctc_prefix_scan_colmajor_kernel(sc_search sb, int split_min) {
  const int s = blockIdx.y;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int nh = CTRL(s, SC_C_NHYP), K = sb.K, W = sb.W, V = sb.V;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nh * K) return;
  const int h = e / K, k = e % K;
  const int T = SC_CTC_T(s), L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR);
  if (ctc_scan_is_split(T, L, split_min)) return;   // ctc_prefix_scan_tpar_kernel's stream
  const bool has = CTRL(s, SC_C_HAS);
  const long row = (long)s * W + h;
  const int c = sb.pre_ids[row * K + k];
  const int last = YSEQ(cur, s, h)[L - 1];
  const bool same = (c == last);
  const int tct = sb.tct;
  const float *__restrict__ xcol = sb.ctcxT + ((long)s * V + c) * tct;
  const float *__restrict__ xblk = sb.ctcxT + ((long)s * V + sb.blank) * tct;
  const float *__restrict__ rp = CTCR(cur, s);
  const float *__restrict__ rsp = CTCRS(cur, s);
  float *rn = sb.ctc_rnew + (long)s * ctc_tck(sb.TCAP) * 2 * (W * K);   // checkpoints: r[16 j + 15] at row j
  const int WK = W * K;
  const int out_len = L - 1;
  int start = out_len > 1 ? out_len : 1;
  if (start > T) start = T;
  for (int j = 0; SC_CTC_CK * j + SC_CTC_CK - 1 < start - 1; ++j) {   // frames before start-1: logzero
    rn[((long)j * 2) * WK + e] = SC_LOGZERO;
    rn[((long)j * 2 + 1) * WK + e] = SC_LOGZERO;
  }
  float r_n = (out_len == 0) ? xcol[0] : SC_LOGZERO;  // r[start-1][n]; start == 1 when out_len == 0
  float r_b = SC_LOGZERO;
  if (((start - 1) & (SC_CTC_CK - 1)) == SC_CTC_CK - 1) {
    rn[((long)((start - 1) / SC_CTC_CK) * 2) * WK + e] = r_n;
    rn[((long)((start - 1) / SC_CTC_CK) * 2 + 1) * WK + e] = r_b;
  }
  float cum = 0.f;   // running blank log-prob sum of the initial (state None) hypothesis: sum_{tau < start} x[tau, blank]
  if (!has)
    for (int t = 0; t < start; ++t) cum += xblk[t];

  auto fetch = [&](CtcChunkT &q, int tb) {   // frames [tb, tb+16), tb a multiple of 16
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t4 = min(tb + 4 * j, tct - 4);
      q.xc[j] = *reinterpret_cast<const float4 *>(xcol + t4);
      q.xb[j] = *reinterpret_cast<const float4 *>(xblk + t4);
    }
    if (has) {  // wave-uniform
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int t = tb + i;
        t = t < T ? (t < 1 ? 1 : t) : T - 1;
        q.pn[i] = rsp[(long)(t - 1) * W + h];
        q.pb[i] = rp[((long)(t - 1) * 2 + 1) * W + h];
      }
    }
  };
  float pm = r_n, ps = 1.f;  // psi = logsumexp over {phi[t-1] + x[t,c]} and r[start-1][n]: running max / scaled sum
  auto frame = [&](const CtcChunkT &q, int tb, int i) {
    const int t = tb + i;
    const float4 c4 = q.xc[i >> 2], b4 = q.xb[i >> 2];
    const float xc = (i & 3) == 0 ? c4.x : (i & 3) == 1 ? c4.y : (i & 3) == 2 ? c4.z : c4.w;
    const float xb = (i & 3) == 0 ? b4.x : (i & 3) == 1 ? b4.y : (i & 3) == 2 ? b4.z : b4.w;
    const float prs = has ? q.pn[i] : cum;    // (no prefix state: r^n is logzero, r^n (+) r^b = the running blank sum)
    const float pb = has ? q.pb[i] : cum;     // r_prev[t-1]
    float phi;
    ctc_frame(r_n, r_b, prs, pb, same, xc, xb, phi);
    if ((i & (SC_CTC_CK - 1)) == SC_CTC_CK - 1) {   // (chunks are 16-frame aligned: i == 15 <=> t % 16 == 15; compile-time per unrolled frame)
      rn[((long)(t / SC_CTC_CK) * 2) * WK + e] = r_n;
      rn[((long)(t / SC_CTC_CK) * 2 + 1) * WK + e] = r_b;
    }
    ctc_psi_add(pm, ps, phi + xc);
    if (!has) cum += xb;
  };
  auto advance = [&](const CtcChunkT &q, int tb) {
    if (tb >= start && tb + 16 <= T) {   // interior chunk: no per-frame bounds
#pragma unroll
      for (int i = 0; i < 16; ++i) frame(q, tb, i);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (tb + i >= start && tb + i < T) frame(q, tb, i);
    }
  };
  if (start < T) {
    CtcChunkT qa, qb;
    int tb = start & ~15;
    fetch(qa, tb);
    while (true) {
      if (tb + 16 < T) fetch(qb, tb + 16);
      advance(qa, tb);
      tb += 16;
      if (tb >= T) break;
      if (tb + 16 < T) fetch(qa, tb + 16);
      advance(qb, tb);
      tb += 16;
      if (tb >= T) break;
    }
  }
  float psi = pm + logf(ps);
  const float rsum_last = has ? rsp[(long)(T - 1) * W + h] : cum;
  if (c == sb.eos) psi = rsum_last;
  if (c == sb.blank) psi = SC_LOGZERO;
  sb.psi[row * K + k] = psi;
  if (k == 0) sb.psi_eos[row] = rsum_last;
}

// ---------------------------------------------------------------------------
// The scan split over T for long tables and few active streams (a 180 s segment has T = 4500: 0.66 ms per decode
// iteration for the sequential walk, whatever the number of streams).  The recurrence is affine in the probability
// domain -  n' = (n + phi) * pc,  b' = (n + b) * pb  - so a segment of frames acts on its start state as
//     n_end = A * n0 + Un,      b_end = C * n0 + D * b0 + Ub
// with five coefficients that a thread accumulates over its segment WITHOUT knowing the start state (log domain:
// A += xc, D += xb, Un = lse(Un, phi) + xc, C = lse(A, C) + xb, Ub = lse(Un, Ub) + xb).  A workgroup holds 16
// (hypothesis, candidate) pairs x 32 segments: ONE pass = coefficients (+ the segment's share of psi, which does not
// depend on the state at all), a 32-step combine in LDS gives every segment its start state - stored; the forward
// variables of the W winners are rebuilt from them by ctc_gather_state_kernel (round 5; rounds 2-4 re-walked all W x K
// pairs here: twice the arithmetic).  Rounding differs from the sequential walk at the level of the fp32 log-add-exps
// (same magnitudes).  WHICH form a stream's scan takes is a function of ITS table (frames to walk >= the threshold),
// never of the number of streams in the step: the same audio gives the same bits whoever else is on the GPU.
// ---------------------------------------------------------------------------
#define CTC_EB (256 / CTC_NSEG)   // pairs per workgroup
__global__ __launch_bounds__(256) void ctc_prefix_scan_tpar_kernel(sc_search sb, int split_min) {
  __shared__ float co[8][CTC_NSEG][CTC_EB];    // A, C, D, Un, Ub, psi max, psi sum, blank sum of every (segment, pair)
  __shared__ float st0[2][CTC_NSEG][CTC_EB];   // start state (n, b) of every segment
  __shared__ float cum0[CTC_NSEG][CTC_EB];     // !has: running blank sum at the segment start
  const int s = blockIdx.y;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int nh = CTRL(s, SC_C_NHYP), K = sb.K, W = sb.W, V = sb.V;
  const int T = SC_CTC_T(s), L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR);
  if (!ctc_scan_is_split(T, L, split_min)) return;   // the sequential kernel's stream
  if ((int)blockIdx.x * CTC_EB >= nh * K) return;
  const int el = threadIdx.x % CTC_EB, p = threadIdx.x / CTC_EB;
  const int e_raw = blockIdx.x * CTC_EB + el;
  const bool valid = e_raw < nh * K;
  const int e = valid ? e_raw : nh * K - 1;   // surplus lanes shadow the last pair (no stores)
  const int h = e / K, k = e % K;
  const bool has = CTRL(s, SC_C_HAS);
  const long row = (long)s * W + h;
  const int c = sb.pre_ids[row * K + k];
  const int last = YSEQ(cur, s, h)[L - 1];
  const bool same = (c == last);
  const int tct = sb.tct;
  const float *__restrict__ xcol = sb.ctcxT + ((long)s * V + c) * tct;
  const float *__restrict__ xblk = sb.ctcxT + ((long)s * V + sb.blank) * tct;
  const float *__restrict__ rp = CTCR(cur, s);
  const float *__restrict__ rsp = CTCRS(cur, s);
  float *rn = sb.ctc_rnew + (long)s * ctc_tck(sb.TCAP) * 2 * (W * K);   // checkpoints: r[16 j + 15] at row j
  const int WK = W * K;
  const int out_len = L - 1;
  int start = out_len > 1 ? out_len : 1;
  if (start > T) start = T;
  const float r_n0 = (out_len == 0) ? xcol[0] : SC_LOGZERO;  // r[start-1][n]; start == 1 when out_len == 0
  const float r_b0 = SC_LOGZERO;
  // segments: boundaries at multiples of 16 frames (the loads are 16-frame chunks)
  const int base = start & ~15;
  const int seg = 16 * cdiv(T - base, 16 * CTC_NSEG);
  const int t_lo = max(start, base + p * seg), t_hi = min(T, base + (p + 1) * seg);

  auto fetch = [&](CtcChunkT &q, int tb) {   // frames [tb, tb+16), tb a multiple of 16
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t4 = min(tb + 4 * j, tct - 4);
      q.xc[j] = *reinterpret_cast<const float4 *>(xcol + t4);
      q.xb[j] = *reinterpret_cast<const float4 *>(xblk + t4);
    }
    if (has) {  // uniform per workgroup
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int t = tb + i;
        t = t < T ? (t < 1 ? 1 : t) : T - 1;
        q.pn[i] = rsp[(long)(t - 1) * W + h];
        q.pb[i] = rp[((long)(t - 1) * 2 + 1) * W + h];
      }
    }
  };
  // walk the frames [lo, hi) in 16-frame chunks, one chunk fetched ahead: body(q, i, t) per frame
  auto walk = [&](int lo, int hi, auto body) {
    if (lo >= hi) return;
    CtcChunkT qa, qb;
    int tb = lo & ~15;
    fetch(qa, tb);
    while (true) {
      if (tb + 16 < hi) fetch(qb, tb + 16);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (tb + i >= lo && tb + i < hi) body(qa, i, tb + i);
      tb += 16;
      if (tb >= hi) break;
      if (tb + 16 < hi) fetch(qa, tb + 16);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (tb + i >= lo && tb + i < hi) body(qb, i, tb + i);
      tb += 16;
      if (tb >= hi) break;
    }
  };
  auto comp = [](const float4 &v, int i) { return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w; };

  // ---- pass 0 (no previous prefix): the blank log-prob sum in front of every segment ----
  float cum = 0.f;
  if (!has) {
    float bs = 0.f;
    if (p == 0)
      for (int t = 0; t < start; ++t) bs += xblk[t];
    walk(t_lo, t_hi, [&](const CtcChunkT &q, int i, int) { bs += comp(q.xb[i >> 2], i); });
    co[7][p][el] = bs;
    __syncthreads();
    for (int q = 0; q < p; ++q) cum += co[7][q][el];
    if (p == 0) {
      cum = 0.f;
      for (int t = 0; t < start; ++t) cum += xblk[t];
    } else {
      // segment 0's sum already contains the frames before start
    }
    cum0[p][el] = cum;
  }
  // ---- pass 1: the segment's affine map and its share of psi ----
  float A = 0.f, C = SC_LOGZERO, D = 0.f, Un = SC_LOGZERO, Ub = SC_LOGZERO;
  float pm = SC_LOGZERO, ps = 0.f;
  {
    float cu = cum;
    walk(t_lo, t_hi, [&](const CtcChunkT &q, int i, int) {
      const float xc = comp(q.xc[i >> 2], i), xb = comp(q.xb[i >> 2], i);
      const float pb = has ? q.pb[i] : cu;     // r_prev[t-1]
      const float phi = same ? pb : (has ? q.pn[i] : cu);
      const float nUb = lse2(Un, Ub) + xb;
      const float nC = lse2(A, C) + xb;
      Un = lse2(Un, phi) + xc;
      Ub = nUb;
      C = nC;
      A += xc;
      D += xb;
      ctc_psi_add(pm, ps, phi + xc);
      if (!has) cu += xb;
    });
  }
  co[0][p][el] = A; co[1][p][el] = C; co[2][p][el] = D; co[3][p][el] = Un; co[4][p][el] = Ub;
  co[5][p][el] = pm; co[6][p][el] = ps;
  __syncthreads();
  // ---- combine: start state of every segment, psi ----
  if (p == 0) {
    float n = r_n0, b = r_b0;
    float gm = r_n0, gs = 1.f;   // psi = logsumexp over {phi[t-1] + x[t,c]} and r[start-1][n]
    for (int q = 0; q < CTC_NSEG; ++q) {
      st0[0][q][el] = n;
      st0[1][q][el] = b;
      const float nn = lse2(co[0][q][el] + n, co[3][q][el]);
      const float nb = lse2(lse2(co[1][q][el] + n, co[2][q][el] + b), co[4][q][el]);
      n = nn;
      b = nb;
      const float m = sc_max_raw(gm, co[5][q][el]);
      gs = gs * sc_exp_neg(gm - m) + co[6][q][el] * sc_exp_neg(co[5][q][el] - m);
      gm = m;
    }
    if (valid) {
      float psi = gm + logf(gs);
      float total = 0.f;   // !has: blank sum over all T frames
      if (!has)
        for (int q = 0; q < CTC_NSEG; ++q) total += co[7][q][el];
      const float rsum_last = has ? rsp[(long)(T - 1) * W + h] : total;
      if (c == sb.eos) psi = rsum_last;
      if (c == sb.blank) psi = SC_LOGZERO;
      sb.psi[row * K + k] = psi;
      if (k == 0) sb.psi_eos[row] = rsum_last;
    }
  }
  __syncthreads();
  // ---- (round 5) no second pass: the start state of every segment goes to rows 0 .. CTC_NSEG-1 of the stream's ctc_rnew
  // (the sequential kernel keeps its 16-frame checkpoints there); ctc_gather_state_kernel re-walks the W WINNERS' segments
  // from them - the candidates' r[t] were computed twice here only to be thrown away for all but W of the W x K pairs
  if (valid) {
    rn[((long)p * 2) * WK + e] = st0[0][p][el];
    rn[((long)p * 2 + 1) * WK + e] = st0[1][p][el];
  }
}

// The scan and the state rebuild must agree on which streams were split over T (the T-parallel scan parks 32 segment START
// states in ctc_rnew where the sequential scan leaves its 16-frame checkpoints): the effective split_min of the LAST scan of a
// batch (keyed by its ctc_rnew buffer) is remembered here and sc_ctc_gather_state_split refuses another value - since ABI 5 the
// legacy sc_ctc_gather_state() behind a split scan silently rebuilt the state from the wrong kind of rows.
static std::mutex g_scan_split_mutex;
static std::unordered_map<const void *, int> g_scan_split;   // ctc_rnew -> effective split_min of its last scan
static void scan_split_note(const sc_search *sbp, int eff) {
  std::lock_guard<std::mutex> lk(g_scan_split_mutex);
  g_scan_split[sbp->ctc_rnew] = eff;
}
static int scan_split_last(const sc_search *sbp) {   // -1: no scan of this batch seen yet
  std::lock_guard<std::mutex> lk(g_scan_split_mutex);
  auto it = g_scan_split.find(sbp->ctc_rnew);
  return it == g_scan_split.end() ? -1 : it->second;
}

// split_min > 0: streams with at least split_min frames to walk take the T-parallel kernel
extern "C" int sc_ctc_prefix_scan_split(const sc_search *sbp, int split_min, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  SC_CHECK_ARG(sbp->ctcxT && sbp->tct >= 4 && sbp->tct % 4 == 0, "the prefix scan needs the column-major table copy (sc_search.ctcxT)");
  split_min = ctc_split_min_ok(sbp->TCAP, split_min);
  scan_split_note(sbp, split_min);
  dim3 grid(cdiv(sbp->W * sbp->K, 256), sbp->S);
  ctc_prefix_scan_colmajor_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(*sbp, split_min);
  if (split_min > 0)
    ctc_prefix_scan_tpar_kernel<<<dim3(cdiv(sbp->W * sbp->K, CTC_EB), sbp->S), 256, 0, (hipStream_t)stream>>>(*sbp, split_min);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

extern "C" int sc_ctc_prefix_scan(const sc_search *sbp, void *stream) { return sc_ctc_prefix_scan_split(sbp, 0, stream); }

// ---------------------------------------------------------------------------
// score fusion + per-hypothesis top-W
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sort_key_value(unsigned long long c) {
  unsigned u = (unsigned)(c >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;   // inverse of the monotone map in sort_key
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void fuse_topw_kernel(sc_search sb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int need_full;
  const int row = blockIdx.x, s = row / sb.W, h = row % sb.W;
  if (!CTRL(s, SC_C_ACTIVE) || h >= CTRL(s, SC_C_NHYP)) return;
  const int V = sb.V, K = sb.K, W = sb.W, tid = threadIdx.x;
  float *comb = smem, *ctc = smem + V;
  if (!(sb.w_ctc > 0.f)) {
    // ctc_weight <= 0: the reference builds NO CTC scorer (beam_search.py:925) - the search is decoder-only, the top-W of
    // w_dec * logp are the first W pre-beam candidates (the same composite keys: ties towards the lowest id), and
    // Hypothesis.scores never gets a "ctc" entry (cand_ctc = 0 keeps sc_ctc at 0)
    if (tid < W) {
      const int c = sb.pre_ids[(long)row * K + tid];
      sb.cand_tok[(long)row * W + tid] = c;
      sb.cand_score[(long)row * W + tid] = __fmul_rn(sb.w_dec, sb.logp[(long)row * V + c]);
      sb.cand_ctc[(long)row * W + tid] = 0.f;
    }
    return;
  }
  const float s_prev = CTRL(s, SC_C_HAS) ? sb.ctc_s[((long)CTRL(s, SC_C_CUR) * sb.S + s) * W + h] : 0.f;
  // ---- fast path: only the K pre-beam candidates and eos carry a CTC score
  // above logzero, so the top-W of the fused scores over V is the top-W over
  // these <= K+1 tokens whenever its W-th score beats the best score any other
  // token can reach (w_dec * logp <= 0 plus w_ctc * (logzero - s_prev)).
  // One wave sorts them in registers; otherwise fall through to the full sort.
  if (K < 64 && W <= 64 && sb.w_dec >= 0.f) {
    if (tid < 64) {
      const int lane = tid;
      int c = -1;
      float ps = SC_LOGZERO;
      if (lane < K) {
        c = sb.pre_ids[(long)row * K + lane];
        ps = sb.psi[(long)row * K + lane];
      }
      const bool eos_listed = __ballot(c == sb.eos) != 0ull;
      if (lane == K && !eos_listed) c = sb.eos;
      if (c == sb.eos) ps = sb.psi_eos[row];
      if (c == sb.blank) ps = SC_LOGZERO;
      unsigned long long key = 0ull;
      if (c >= 0) {
        const float cv = __fsub_rn(ps, s_prev);
        ctc[c] = cv;
        key = sort_key(__fadd_rn(__fmul_rn(sb.w_dec, sb.logp[(long)row * V + c]), __fmul_rn(sb.w_ctc, cv)), c);
      }
      key = wave_sort64_desc(key, lane);
      const float bound = __fmul_rn(sb.w_ctc, __fsub_rn(SC_LOGZERO, s_prev));
      const unsigned long long kw = shfl_u64(key, W - 1);
      const bool ok = kw != 0ull && sort_key_value(kw) > bound;
      if (lane == 0) need_full = ok ? 0 : 1;
      if (ok && lane < W) {
        const int i = sort_key_index(key);
        sb.cand_tok[(long)row * W + lane] = i;
        sb.cand_score[(long)row * W + lane] = sort_key_value(key);
        sb.cand_ctc[(long)row * W + lane] = ctc[i];
      }
    }
    __syncthreads();
    if (!need_full) return;
    __syncthreads();
  }
  for (int v = tid; v < V; v += 256) ctc[v] = SC_LOGZERO;
  __syncthreads();
  for (int k = tid; k < K; k += 256) ctc[sb.pre_ids[(long)row * K + k]] = sb.psi[(long)row * K + k];
  __syncthreads();
  if (tid == 0) {
    ctc[sb.eos] = sb.psi_eos[row];
    ctc[sb.blank] = SC_LOGZERO;
  }
  __syncthreads();
  for (int v = tid; v < V; v += 256) {
    const float cv = __fsub_rn(ctc[v], s_prev);
    ctc[v] = cv;
    comb[v] = __fadd_rn(__fmul_rn(sb.w_dec, sb.logp[(long)row * V + v]), __fmul_rn(sb.w_ctc, cv));
  }
  __syncthreads();
  const int NP = next_pow2(V);
  unsigned long long *comp = reinterpret_cast<unsigned long long *>(smem + 2 * V);
  for (int c = tid; c < NP; c += 256) comp[c] = c < V ? sort_key(comb[c], c) : 0ull;
  __syncthreads();
  block_bitonic_sort_desc(comp, NP);
  for (int k = tid; k < W; k += 256) {
    const int i = sort_key_index(comp[k]);
    sb.cand_tok[(long)row * W + k] = i;
    sb.cand_score[(long)row * W + k] = comb[i];
    sb.cand_ctc[(long)row * W + k] = ctc[i];
  }
}

extern "C" int sc_fuse_topw(const sc_search *sbp, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  int np = 1;
  while (np < sbp->V) np <<= 1;
  size_t smem = 2 * sbp->V * sizeof(float) + np * sizeof(unsigned long long);
  fuse_topw_kernel<<<sbp->S * sbp->W, 256, smem, (hipStream_t)stream>>>(*sbp);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// beam_prune: one workgroup per stream.  Candidates (h, j) are ranked by the
// float64 total (stable, hypothesis-major order), the best W become the next
// hypotheses; all bookkeeping of Hypothesis objects happens here.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beam_prune_kernel(sc_search sb) {
  extern __shared__ __attribute__((aligned(16))) double tot[];
  __shared__ int fl_any, fl_all, fl_best, fl_rep;
  __shared__ int win_h[64], win_tok[64];   // per output rank: source hypothesis, appended token
  // K|V pool rows the NEW hypotheses descend from (bit per row): marked while their histories are copied, then the
  // lowest unmarked rows become the rows of their newest tokens (position L) - what the next decode step appends to.
  // No allocator state survives the launch: the ancestor table of a side IS the state.  A step that is rolled back
  // (beam_search.py:827-836) leaves the other side as it was: its rows for position L-1 are still in ITS table.
  __shared__ unsigned bm[SC_KV_MAX_ROWS / 32];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  for (int w = tid; w < (sb.kv_rows + 31) / 32; w += 256) bm[w] = 0u;
  const int W = sb.W, K = sb.K;
  const int nh = CTRL(s, SC_C_NHYP), L = CTRL(s, SC_C_L), T = CTRL(s, SC_C_T);
  const int cur = CTRL(s, SC_C_CUR), o = 1 - cur;
  const int n = nh * W, nout = n < W ? n : W;
  if (tid == 0) { fl_any = 0; fl_all = 1; fl_best = 0; fl_rep = 0; }
  for (int e = tid; e < n; e += 256) {
    const int h = e / W, j = e % W;
    tot[e] = sb.score[((long)cur * sb.S + s) * W + h] + (double)sb.cand_score[((long)s * W + h) * W + j];
  }
  __syncthreads();
  // ---- rank the nh*W candidates; the winners do the O(1) bookkeeping ----
  for (int e = tid; e < n; e += 256) {
    const double te = tot[e];
    int rk = 0;
    for (int j = 0; j < n; ++j) rk += (tot[j] > te) || (tot[j] == te && j < e);
    if (rk >= nout) continue;
    const int h = e / W, j = e % W, i = rk;
    const long prow = (long)s * W + h;
    const int tok = sb.cand_tok[prow * W + j];
    win_h[i] = h;
    win_tok[i] = tok;
    const long oi = ((long)o * sb.S + s) * W + i, ci = ((long)cur * sb.S + s) * W + h;
    sb.score[oi] = te;
    sb.sc_dec[oi] = sb.sc_dec[ci] + (double)sb.logp[prow * sb.V + tok];
    sb.sc_ctc[oi] = sb.sc_ctc[ci] + (double)sb.cand_ctc[prow * W + j];
    int kk = -1;
    for (int q = 0; q < K; ++q)
      if (sb.pre_ids[prow * K + q] == tok) { kk = q; break; }
    float sval;
    if (tok == sb.blank) sval = SC_LOGZERO;
    else if (tok == sb.eos) sval = sb.psi_eos[prow];
    else if (kk >= 0) sval = sb.psi[prow * K + kk];
    else sval = SC_LOGZERO;
    sb.ctc_s[oi] = sval;
    sb.sel[((long)s * W + i) * 2] = h;
    sb.sel[((long)s * W + i) * 2 + 1] = kk < 0 ? 0 : kk;
    const bool is_eos = tok == sb.eos;
    if (is_eos) atomicOr(&fl_any, 1);
    else atomicAnd(&fl_all, 0);
    if (i == 0 && is_eos) atomicOr(&fl_best, 1);
  }
  __syncthreads();
  // ---- histories of the winners: all threads copy (rank i, position p) pairs ----
  const int *asrc = ANC(cur, s);
  int *adst = ANC(o, s);
  bool rep = false;
  // (source and destination are the two sides of the ping-pong buffers: the loads of a batch of UB elements are all
  // issued before the first store - one memory round trip per batch instead of one per element)
  constexpr int UB = 4;
  const int ntot = nout * (L + 1);
  for (int e0 = tid; e0 < ntot; e0 += 256 * UB) {
    int yv[UB], xv[UB], av[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = min(e0 + 256 * u, ntot - 1);
      const int i = e / (L + 1), p = min(e % (L + 1), L - 1);
      const int h = win_h[i];
      yv[u] = YSEQ(cur, s, h)[p];
      xv[u] = XPOS(cur, s, h)[p];
      av[u] = asrc[(long)p * W + h];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + 256 * u;
      if (e >= ntot) break;
      const int i = e / (L + 1), p = e % (L + 1);
      const int tok = win_tok[i];
      int *ydst = YSEQ(o, s, i), *xdst = XPOS(o, s, i);
      if (p < L) {
        ydst[p] = yv[u];
        xdst[p] = xv[u];
        if (p >= 1 && yv[u] == tok && tok != sb.sos && tok != sb.eos) rep = true;
        adst[(long)p * W + i] = av[u];   // pool rows of the parent's history, incl. its newest token's row (position L-1)
        atomicOr(&bm[(av[u] >> 5) & (SC_KV_MAX_ROWS / 32 - 1)], 1u << (av[u] & 31));
      } else {
        ydst[L] = tok;
        xdst[L] = T - 1;
      }
    }
  }
  if (rep) atomicOr(&fl_rep, 1);
  __syncthreads();
  if (tid == 0)
    sb.flags[s] = (fl_any ? SC_F_ANY_EOS : 0) | (fl_best ? SC_F_BEST_EOS : 0) |
                  (fl_all ? SC_F_ALL_EOS : 0) | (fl_rep ? SC_F_REPEAT : 0);
  if (tid >= 64 || L + 1 > sb.LCAP) return;   // (a stream at max_tokens is failed by the host before its next step)
  // ---- pool rows for position L of the nout new hypotheses: the lowest free rows, in hypothesis order (wave 0)
  const int lane = tid, NR = sb.kv_rows, nwords = (NR + 31) / 32;
  int *out = adst + (long)L * W;
  int got = 0;
  for (int w0 = 0; w0 < nwords && got < nout; w0 += 64) {
    const int w = w0 + lane;
    unsigned fr = w < nwords ? ~bm[w] : 0u;
    if (w == nwords - 1 && (NR & 31)) fr &= (1u << (NR & 31)) - 1u;
    const int c = __popc(fr);
    int incl = c;
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) {
      const int t = __shfl_up(incl, o2, 64);
      if (lane >= o2) incl += t;
    }
    int k = got + incl - c;   // index of this lane's first free row among all free rows
    while (fr && k < nout) {
      const int b = __ffs(fr) - 1;
      fr &= fr - 1u;
      out[k++] = w * 32 + b;
    }
    got += __shfl(incl, 63, 64);
  }
  if (got < nout && lane >= got && lane < nout) out[lane] = NR - 1;   // exhausted: in-bounds garbage, the host fails the stream
  if (lane == 0 && sb.kvflags) sb.kvflags[s] = got < nout ? 1 : 0;
}

extern "C" int sc_beam_prune(const sc_search *sbp, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  SC_CHECK_ARG(sbp->W <= 64, "beam wider than 64");
  SC_CHECK_ARG(sbp->kv_rows >= sbp->W && sbp->kv_rows <= SC_KV_MAX_ROWS, "kv_rows out of range (beam .. 65536)");
  size_t smem = (size_t)sbp->W * sbp->W * sizeof(double);
  beam_prune_kernel<<<sbp->S, 256, smem, (hipStream_t)stream>>>(*sbp);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// CTCPrefixScorer.select_state (scorers.py:382-431) for the W winners of the step - and the rebuild of their forward
// variables from the checkpoints the scan left (see SC_CTC_CK above): thread = (winner i, 16-frame segment j) walks its
// segment with the scan's own recurrence from the checkpoint in front of it (or from the scan's initial state in the
// segment that holds frame start-1) and writes r[t] of all its frames into the other side of ctc_r.
__global__ __launch_bounds__(256) void ctc_gather_state_kernel(sc_search sb, int split_min) {
  const int s = blockIdx.y;
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int W = sb.W, K = sb.K, V = sb.V, T = SC_CTC_T(s), nh = CTRL(s, SC_C_NHYP), L = CTRL(s, SC_C_L);
  const int nout = nh * W < W ? nh * W : W;
  const int cur = CTRL(s, SC_C_CUR), o = 1 - cur;
  const bool has = CTRL(s, SC_C_HAS);
  const int WK = W * K, tct = sb.tct;
  float *dst = CTCR(o, s);
  float *dst_rs = CTCRS(o, s);
  const float *__restrict__ rp = CTCR(cur, s);
  const float *__restrict__ rsp = CTCRS(cur, s);
  const float *ck = sb.ctc_rnew + (long)s * ctc_tck(sb.TCAP) * 2 * WK;
  const int out_len = L - 1;
  int start = out_len > 1 ? out_len : 1;
  if (start > T) start = T;
  // the stream's scan was split over T (ctc_prefix_scan_tpar_kernel): CTC_NSEG segments with the scan's own boundaries,
  // their start states in rows 0 .. CTC_NSEG-1 of ck; otherwise 16-frame segments behind the checkpoints
  const bool split = ctc_scan_is_split(T, L, split_min);
  const int base = start & ~15;
  const int seg = split ? 16 * cdiv(T - base, 16 * CTC_NSEG) : SC_CTC_CK;
  const int nseg = split ? CTC_NSEG : cdiv(T, SC_CTC_CK);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nout * nseg; idx += gridDim.x * blockDim.x) {
    const int i = idx % nout, j = idx / nout;
    const int h = sb.sel[((long)s * W + i) * 2], k = sb.sel[((long)s * W + i) * 2 + 1];
    const int e = h * K + k;
    const int c = sb.pre_ids[((long)s * W + h) * K + k];
    const bool same = c == YSEQ(cur, s, h)[L - 1];
    const float *__restrict__ xcol = sb.ctcxT + ((long)s * V + c) * tct;
    const float *__restrict__ xblk = sb.ctcxT + ((long)s * V + sb.blank) * tct;
    // frames [t_lo, t_hi) of this thread and the state in front of them
    int t_lo, t_hi;
    float r_n, r_b;
    if (split) {
      t_lo = max(start, base + j * seg);
      t_hi = min(T, base + (j + 1) * seg);
      r_n = ck[((long)j * 2) * WK + e];
      r_b = ck[((long)j * 2 + 1) * WK + e];
      if (j == 0) {   // ... and the frames in front of the walk: logzero, then the initial state at start-1
        for (int t = 0; t < start - 1; ++t) {
          dst[((long)t * 2) * W + i] = SC_LOGZERO;
          dst[((long)t * 2 + 1) * W + i] = SC_LOGZERO;
          dst_rs[(long)t * W + i] = lse2(SC_LOGZERO, SC_LOGZERO);
        }
        dst[((long)(start - 1) * 2) * W + i] = (out_len == 0) ? xcol[0] : SC_LOGZERO;
        dst[((long)(start - 1) * 2 + 1) * W + i] = SC_LOGZERO;
        dst_rs[(long)(start - 1) * W + i] = lse2((out_len == 0) ? xcol[0] : SC_LOGZERO, SC_LOGZERO);
      }
    } else {
      const int t0 = j * SC_CTC_CK, t1 = min(t0 + SC_CTC_CK, T);
      if (t0 <= start - 1) {   // the segment that holds frame start-1 (or lies in front of it)
        r_n = (out_len == 0) ? xcol[0] : SC_LOGZERO;
        r_b = SC_LOGZERO;
        for (int t = t0; t < min(start - 1, t1); ++t) {
          dst[((long)t * 2) * W + i] = SC_LOGZERO;
          dst[((long)t * 2 + 1) * W + i] = SC_LOGZERO;
          dst_rs[(long)t * W + i] = lse2(SC_LOGZERO, SC_LOGZERO);
        }
        if (start - 1 < t1) {
          dst[((long)(start - 1) * 2) * W + i] = r_n;
          dst[((long)(start - 1) * 2 + 1) * W + i] = r_b;
          dst_rs[(long)(start - 1) * W + i] = lse2(r_n, r_b);
        }
        t_lo = start;
      } else {
        r_n = ck[((long)(j - 1) * 2) * WK + e];
        r_b = ck[((long)(j - 1) * 2 + 1) * WK + e];
        t_lo = t0;
      }
      t_hi = t1;
    }
    if (t_lo >= t_hi) continue;
    float cum = 0.f;   // !has: running blank log-prob sum of the initial (state None) hypothesis, in the scan's order
    if (!has)
      for (int t = 0; t < t_lo; ++t) cum += xblk[t];
    for (int tb = t_lo & ~(SC_CTC_CK - 1); tb < t_hi; tb += SC_CTC_CK) {
      // all inputs of 16 frames are requested before the recurrence walks them (table column, blank column, r of the
      // prefix: independent of the recurrence - issued inside the loop they cost one L2 round trip per frame)
      float xc[SC_CTC_CK], xb[SC_CTC_CK], pn[SC_CTC_CK], pb[SC_CTC_CK];
#pragma unroll
      for (int q = 0; q < SC_CTC_CK / 4; ++q) {
        const int t4 = min(tb + 4 * q, tct - 4);   // (tb is a multiple of 16, tct of 4: aligned; frames >= T are not used)
        const float4 c4 = *reinterpret_cast<const float4 *>(xcol + t4), b4 = *reinterpret_cast<const float4 *>(xblk + t4);
        xc[4 * q] = c4.x; xc[4 * q + 1] = c4.y; xc[4 * q + 2] = c4.z; xc[4 * q + 3] = c4.w;
        xb[4 * q] = b4.x; xb[4 * q + 1] = b4.y; xb[4 * q + 2] = b4.z; xb[4 * q + 3] = b4.w;
      }
#pragma unroll
      for (int q = 0; q < SC_CTC_CK; ++q) {
        const int tp = min(max(tb + q - 1, 0), T - 1);
        pn[q] = has ? rsp[(long)tp * W + h] : 0.f;          // r^n (+) r^b of the prefix (ctc_frame)
        pb[q] = has ? rp[((long)tp * 2 + 1) * W + h] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < SC_CTC_CK; ++q) {
        const int t = tb + q;
        if (t >= t_lo && t < t_hi) {
          float phi;
          ctc_frame(r_n, r_b, has ? pn[q] : cum, has ? pb[q] : cum, same, xc[q], xb[q], phi);
          dst[((long)t * 2) * W + i] = r_n;
          dst[((long)t * 2 + 1) * W + i] = r_b;
          dst_rs[(long)t * W + i] = lse2(r_n, r_b);
          if (!has) cum += xb[q];
        }
      }
    }
  }
}

extern "C" int sc_ctc_gather_state_split(const sc_search *sbp, int split_min, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  SC_CHECK_ARG(sbp->ctcxT && sbp->tct >= 4, "the state rebuild needs the column-major table copy (sc_search.ctcxT)");
  int gx = cdiv(ctc_tck(sbp->TCAP) * sbp->W, 256);
  if (gx > 16) gx = 16;
  const int eff = ctc_split_min_ok(sbp->TCAP, split_min), last = scan_split_last(sbp);
  if (last >= 0 && last != eff) {
    sc_set_error("sc_ctc_gather_state_split: the last prefix scan of this batch ran with split_min = %d, the rebuild was asked for %d - "
                 "streams split over T leave segment states, not checkpoints: pass the scan's value (sc_ctc_gather_state() = 0)", last, eff);
    return SC_ERR_ARG;
  }
  ctc_gather_state_kernel<<<dim3(gx, sbp->S), 256, 0, (hipStream_t)stream>>>(*sbp, eff);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

extern "C" int sc_ctc_gather_state(const sc_search *sbp, void *stream) { return sc_ctc_gather_state_split(sbp, 0, stream); }

// ---------------------------------------------------------------------------
// The head-parallel layer kernels (decoder_layer.hip: 3 launches per layer) are used for compaction buckets of at
// most SC_FUSED_MAX_ROWS hypothesis rows.  They win where the step is a chain of latency-bound launches (few
// active streams); at large buckets every (stream, head) workgroup re-reading all partial sums of its stream costs
// as much fabric traffic as the launches save, and the six-launch form (row panels + stand-alone attention) is
// level or faster (docs/profiles_r1-r3/r02_fused_threshold_sweep.txt, tools/fused_threshold_sweep.sh: strict lock-step
// 27.21 / 26.95 / 26.65 / 27.01 ms per chunk step with the limit at 320 / 640 / 960 / 1280 rows).
static bool dec_fused_ok(const sc_search &sb) {
  if (const char *e = sc_hook("SC_DEC_FUSED"))   // tests: "0" forces the six-launch layers, "1" the fused ones at any size
    return atoi(e) != 0 && sc_dec_layer_fused_supported(sb.d, sb.H, sb.W, sb.F) && sb.ph1 && sb.out_w_q;
  // (round 5) the form is a function of the MODEL, never of the bucket size: the six-launch layers sum in another order,
  // and which bucket a stream decodes in depends on who else is on the GPU.  SC_FUSED_MAX (tools only) restores the old
  // row threshold for A/B runs.
  if (const char *e = sc_hook("SC_FUSED_MAX"))
    if ((sb.rowmap ? sb.n_rows : sb.S * sb.W) > atoi(e) && sc_dec_layer_hpw(sb) < 2) return false;
  return sc_dec_layer_fused_supported(sb.d, sb.H, sb.W, sb.F) && sb.ph1 && sb.ph2 && sb.ffn_part &&
         sb.max_ffn_part >= 1 && sb.out_w_q && sb.V % sb.d == 0 && sb.layers && sb.layers[0].wqkv_pp &&
         sb.layers[0].wq_pp && sb.layers[0].wo_pp && sb.layers[0].w1_p;
}

// which form of the decoder layers sc_decode_step runs for this bucket: 0 six launches per layer, 1 head-parallel, 2 stream-resident
// (3 launches) - streams.hip accounts the cross-attention's K|V traffic by kernel family
int sc_decode_step_form(const sc_search *sbp) { return dec_fused_ok(*sbp) ? (sc_dec_layer_stream_form(*sbp) ? 2 : 1) : 0; }

extern "C" int sc_decode_step(const sc_search *sbp, void *stream) { return sc_decode_step_ex(sbp, 0, stream); }

extern "C" int sc_decode_step_ex(const sc_search *sbp, int scan_split_min, void *stream) {
  SC_CHECK_ARG(sbp, "null");
  const sc_search &sb = *sbp;
  const int n = sb.rowmap ? sb.n_rows : sb.S * sb.W;
  int rc;
  if (dec_fused_ok(sb) && sc_dec_layer_stream_form(sb)) {
    // (round 6) stream-resident form, 2 launches per layer: x'' ping-pongs dx <-> dxn, LayerNorm3(x'') travels in dq
    float *xa = sb.dx, *xb = sb.dxn;
    int npart = 0;
    for (int li = 0; li < sb.n_layers; ++li) {
      SC_TRY(sc_dec_layer_stream(sbp, li, xa, xb, sb.dq, sb.ffn_part, npart, stream));
      SC_TRY(sc_dec_layer_ffn_xn(sbp, li, sb.dq, sb.ffn_part, sb.max_ffn_part, &npart, stream));
      float *t = xa; xa = xb; xb = t;
    }
    SC_TRY(sc_dec_output_logits(sbp, xa, xb, sb.ffn_part, npart, stream));
  } else if (dec_fused_ok(sb)) {
    // 3 launches per layer; x ping-pongs dx <-> dxn
    float *xa = sb.dx, *xb = sb.dxn;
    int npart = 0;
    for (int li = 0; li < sb.n_layers; ++li) {
      SC_TRY(sc_dec_layer_self(sbp, li, xa, xb, sb.ffn_part, npart, stream));
      SC_TRY(sc_dec_layer_cross(sbp, li, xb, xa, stream));
      if (sc_dec_layer_split_ffn(sb)) {   // (round 6, large buckets) the head partials summed once per row, feed-forward without prologue
        SC_TRY(sc_dec_layer_reduce_ln(sbp, li, xa, xb, sb.dq, stream));
        SC_TRY(sc_dec_layer_ffn_xn(sbp, li, sb.dq, sb.ffn_part, sb.max_ffn_part, &npart, stream));
      } else
      SC_TRY(sc_dec_layer_ffn(sbp, li, xa, xb, sb.ffn_part, sb.max_ffn_part, &npart, stream));
      float *t = xa; xa = xb; xb = t;
    }
    SC_TRY(sc_dec_output_logits(sbp, xa, xb, sb.ffn_part, npart, stream));
  } else {
  SC_TRY(sc_dec_embed(sbp, stream));
  bool logits_done = false;
  SC_TRY(decoder_layers_impl(sbp, stream, true, &logits_done));  // logits, or after_norm(x) in dxn
  if (!logits_done)
    SC_TRY(sc_gemm(sb.dxn, sb.rowmap, sb.d, sb.out_w, sb.out_b, sb.logits, sb.rowmap, sb.V, n, sb.V, sb.d, 0, 0, stream));
  }
  SC_TRY(sc_logsoftmax_topk(sbp, stream));
  const bool use_ctc = sb.w_ctc > 0.f;   // ctc_weight <= 0: decoder-only search (beam_search.py:925), no scan, no CTC state
  if (use_ctc) SC_TRY(sc_ctc_prefix_scan_split(sbp, scan_split_min, stream));
  SC_TRY(sc_fuse_topw(sbp, stream));
  SC_TRY(sc_beam_prune(sbp, stream));
  if (use_ctc) SC_TRY(sc_ctc_gather_state_split(sbp, scan_split_min, stream));
  return SC_OK;
}
