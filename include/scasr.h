/*
 * libscasr - C ABI of the MI355X (gfx950) streaming-ASR hot path.
 *
 * Drop-in boundary for speechcatcher's native decoder path.
 * The reference is pure Python/PyTorch and has no FFI of its own; each entry
 * point below replaces the torch-op call sites of one reference function and
 * cites it (paths relative to the reference repository root).  The host side
 * (speechcatcher_amd/engine.py, Python like the reference) binds these with
 * ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless marked HOST.
 *   - all functions are asynchronous launches on `stream` (a hipStream_t
 *     passed as void*); return 0 on success, a negative code on error and
 *     never throw.  sc_last_error() returns a thread-local message.
 *   - caller owns all memory; nothing is allocated or freed by the library.
 *   - "rows" tables are int32 ROW indices into a 2-D buffer (multiplied by the
 *     leading dimension inside the kernel); NULL means the identity mapping;
 *     -1 in a gather table reads a zero row.
 *   - all floating point is fp32 (IEEE, no fast-math); hypothesis scores are
 *     accumulated in fp64 like the reference's Python floats.
 */
#ifndef SCASR_H
#define SCASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_ARG (-1)
#define SC_ERR_LAUNCH (-2)
#define SC_ERR_CAPACITY (-3) /* sc_push, per stream: a capacity limit of the batch (frames, tokens, pcm, chunk length) */
#define SC_ERR_INPUT (-4)    /* sc_push, per stream: an input the reference itself raises on */

/* flags of sc_gemm */
#define SC_GEMM_RELU 1
#define SC_GEMM_RESIDUAL 2
#define SC_GEMM_LN_AT_CROWS 8 /* sc_gemm_ln: ln_out rows follow c_rows instead of 0..M-1 */
#define SC_GEMM_NAIVE 4 /* force the scalar reference kernel (debugging) */
#define SC_GEMM_SPLIT16 16 /* tiled kernels only: fp16 hi | lo split of both operands when the tiles are staged, three fp16
                              MFMAs per product sum, fp32 accumulation - fp32-grade results at a fraction of the f32
                              matrix-pipe time (see sc_ffn_ln_s); the small-M kernels ignore it */

/* beam_prune stop flags (one int32 per stream) */
#define SC_F_ANY_EOS 1
#define SC_F_BEST_EOS 2
#define SC_F_ALL_EOS 4
#define SC_F_REPEAT 8

/* ctrl row (int32[8]) per stream, uploaded by the host before every search op */
#define SC_C_ACTIVE 0 /* stream takes part in this launch */
#define SC_C_CUR 1    /* live ping-pong hypothesis buffer (0/1); outputs go to 1-cur */
#define SC_C_FINAL 2
#define SC_C_T 3    /* encoder frames visible to this decode block */
#define SC_C_L 4    /* tokens per live hypothesis, incl. sos */
#define SC_C_NHYP 5 /* live hypotheses (1 or W) */
#define SC_C_HAS 6  /* live hypotheses carry a CTC state */
#define SC_C_TOLD 7 /* sc_ctc_extend_state: rows the CTC states covered before this block */
/* decode-step launches (sc_ctc_prefix_scan, sc_ctc_gather_state): rows of the CTC table and of the
 * forward variables, when that differs from SC_C_T (0: same).  The reference never clears
 * CTCPrefixScorer.impl on reset() (scorers.py:342-350): the next utterance on the same object is
 * scored over the stale table until its own frames outgrow it, while attention sees SC_C_T frames. */
#define SC_C_TCTC 7

typedef struct sc_enc_layer {
  const float *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo;
  const float *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
  const float *w1_p, *w2_p; /* sc_pack_panel_weight of w1, w2 (used when sc_ffn_ln_supported(d, F)) */
  const float *wqkv_p, *wo_p; /* sc_pack_panel_weight of wqkv, wo (used when sc_rowtile_proj_supported(d, d)) */
  const void *w1_h, *w2_h;    /* fp16 copies of w1_p / w2_p (same fragment order) or NULL: the fused FFN then runs
                                 fp16 MFMA inputs with fp32 accumulation (sc_ffn_ln_h) */
  const void *wqkv_h, *wo_h;  /* fp16 copies of wqkv_p / wo_p or NULL: the attention projections with fp16 MFMA inputs
                                 (sc_rowtile_proj_h) */
  const void *w1_s, *w2_s;    /* fp16 hi | lo SPLIT of w1_p / w2_p (weights.py split_panel_weight; same bytes) or NULL: the
                                 fused FFN then computes fp32-grade results with three fp16 MFMAs per product sum
                                 (sc_ffn_ln_s); takes precedence over w1_h / w2_h */
  const void *wqkv_s, *wo_s;  /* the same split of wqkv_p / wo_p or NULL (sc_rowtile_proj_s); precedence over wqkv_h / wo_h */
} sc_enc_layer;

typedef struct sc_dec_layer {
  const float *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo;
  const float *ln2_g, *ln2_b, *wq, *bq, *wo2, *bo2;
  const float *ln3_g, *ln3_b, *w1, *b1, *w2, *b2;
  const float *wo_p, *wq_p, *wo2_p; /* sc_pack_lane_weight of wo, wq, wo2 (used when sc_proj_ln_proj_supported(d)) */
  const float *w1_p, *w2_p;         /* sc_pack_panel_weight of w1, w2 (used when sc_ffn_ln_supported(d, F)) */
  const float *wqkv_q;              /* sc_pack_lane_weight of wqkv (sc_ffn_ln_proj of the layer before) */
  const float *wqkv_pp, *wq_pp, *wo_pp, *wo2_pp; /* sc_pack_panel_weight of wqkv, wq, wo, wo2 (sc_dec_layer_self / _cross), or NULL */
  const void *w1_h, *w2_h;          /* fp16 copies of w1_p / w2_p or NULL (see sc_enc_layer) */
  const void *w1_s, *w2_s;          /* fp16 hi | lo split of w1_p / w2_p or NULL (see sc_enc_layer) */
  const void *wqkv_pph, *wq_pph, *wo_pph, *wo2_pph; /* fp16 copies of wqkv_pp / wq_pp / wo_pp / wo2_pp (same fragment order)
                                       or NULL: the layer kernels' projections with fp16 MFMA inputs, fp32 accumulation
                                       (BASELINE configs[4], with sc_search.act_half / out_w_qh; needs kv_half) */
} sc_dec_layer;

/* Search-side buffers of one StreamBatch (S streams, beam W, pre-beam K). */
typedef struct sc_search {
  int32_t S, W, K, V, d, H, F, n_layers, TCAP, LCAP;
  int32_t blank, eos, sos;
  float w_dec, w_ctc, ln_eps;
  const int32_t *ctrl; /* [S][8] */
  int32_t *flags;      /* [S] */
  const float *ctcx;   /* [S][TCAP][V]  CTC posterior table (quirk A1 rows) */
  const float *ckv;    /* [S][n_layers][TCAP][2d] cross-attention K|V */
  float *skv;          /* [S][n_layers][kv_rows][2d] self-attention K|V rows of a stream: a POOL (round 4) - a row holds
                          the K|V of one (position, hypothesis) token; sc_beam_prune hands every NEW hypothesis the row
                          of its newest token (the lowest rows no new hypothesis descends from: rows of pruned
                          hypotheses are free again), `anc` is the index.  The reference keeps a per-hypothesis output
                          cache instead and copies it when hypotheses fork (transformer_decoder.py:210-249).  Hypotheses of a beam share almost all of their
                          history (1.2 distinct rows per position at beam 10), so kv_rows ~ 1.5 x LCAP instead of the
                          W x LCAP of one slot per (position, hypothesis) */
  int32_t *yseq, *xpos; /* [2][S][W][LCAP] */
  int32_t *anc;         /* [2][S][LCAP][W] pool row (of skv) of the K|V row hypothesis h attends to at position p */
  double *score, *sc_dec, *sc_ctc; /* [2][S][W] */
  float *ctc_r;    /* [2][S][TCAP][2][W] forward variables r^n, r^b per hypothesis */
  float *ctc_s;    /* [2][S][W] prefix score log_psi of the chosen token */
  float *ctc_rnew; /* [S][ceil(TCAP/16)][2][W*K] r of every (hypothesis, candidate) at the CHECKPOINT frames t % 16 == 15
                      (row j = frame 16 j + 15): sc_ctc_gather_state rebuilds the winners' r[t] at every frame from them */
  float *dx, *dxn, *dqkv, *datt, *dq, *dffh, *logits, *logp; /* [S*W][...] */
  int32_t *pre_ids;       /* [S*W][K] */
  float *psi, *psi_eos;   /* [S*W][K], [S*W] */
  float *cand_score;      /* [S*W][W] */
  int32_t *cand_tok;      /* [S*W][W] */
  float *cand_ctc;        /* [S*W][W] */
  int32_t *sel;           /* [S][W][2] (parent hypothesis, candidate index) */
  const float *embed, *pe, *dec_norm_g, *dec_norm_b, *out_w, *out_b;
  const sc_dec_layer *layers; /* HOST array [n_layers] */
  /* Ragged-batch compaction of the dense decoder kernels (GEMMs, row panels,
   * LayerNorm): they process the first n_rows entries of rowmap, a permutation
   * of the S*W hypothesis rows that lists the rows of the ACTIVE streams first
   * (streams leave the lock-step decode loop at different steps).  rowmap NULL:
   * all S*W rows in order.  n_rows is a launch parameter (one captured graph
   * per bucket); rowmap is re-uploaded by the host together with ctrl. */
  const int32_t *rowmap;
  int32_t n_rows;
  const float *out_w_q; /* sc_pack_lane_weight of out_w (sc_ffn_ln_proj of the last layer), or NULL */
  /* head-parallel decoder layers (sc_dec_layer_*): per-head partial products of the two attention output
   * projections [S*W][H][d] and the feed-forward partial sums [max_ffn_part][S*W][d]; NULL: not used */
  float *ph1, *ph2, *ffn_part;
  int32_t max_ffn_part;
  /* column-major copy of the CTC table [S][V][tct] (tct = TCAP rounded up to a multiple of 4), maintained by
   * sc_ctc_extend_state and streamed by sc_ctc_prefix_scan (required) */
  int32_t tct;
  float *ctcxT;
  /* 1: the K|V caches ckv / skv hold IEEE fp16 elements (same element offsets, half the bytes); attention
   * arithmetic, softmax and everything else stay fp32.  Written through sc_kv_rows_to_half (cross) and by the
   * self-attention kernels (self); read by the single-pass attention kernels. */
  int32_t kv_half;
  /* measurement aid (bench.py: algorithmic bytes of the self-attention): when not NULL, every self-attention launch
   * adds the DISTINCT (position, slot) K|V rows it walked for a stream (plus its new rows) to stat_rows[0]
   * (sc_dec_self_attn) / stat_rows[1] (sc_dec_layer_self) / stat_rows[2] (sc_dec_layer_stream) - one atomic per (stream, layer) */
  unsigned long long *stat_rows;
  /* fp16 decoder mode (BASELINE configs[4]; opt-in, never the parity mode): out_w_qh = fp16 copy of out_w_q (the
   * output layer with fp16 MFMA inputs) or NULL; act_half bit 0 (1): the layer kernels use the fp16 weight copies
   * (*_pph) with fp16 MFMA inputs; bit 1 (2): the partial products between the decoder's kernels (ph1, ph2, ffn_part)
   * hold fp16 elements (same element offsets); bit 2 (4): the output layer uses out_w_qh.  LayerNorm, softmax,
   * log-softmax, CTC, scores: fp32. */
  const void *out_w_qh;
  int32_t act_half;
  int32_t kv_rows;   /* rows of the self-attention K|V pool per (stream, layer), <= 65536 */
  int32_t *kvflags;  /* [S] written by sc_beam_prune: 1 = the stream's pool is exhausted (its next step would compute
                        garbage: the host fails the stream with SC_ERR_CAPACITY), 0 = fine */
  float *ctc_rs;     /* [2][S][TCAP][W] log(exp r^n + exp r^b) of ctc_r, frame by frame (round 5): written wherever ctc_r is
                        (sc_ctc_extend_state, sc_ctc_gather_state) and read by the prefix scan, whose W x K lanes all
                        needed this sum of THEIR hypothesis at every frame - 2 of the 7 transcendentals per frame and lane */
} sc_search;

const char *sc_last_error(void);
/* ABI revision of this header: bumped whenever a struct layout or a signature changes incompatibly.  sc_version()
 * returns the revision the LIBRARY was built with; a host compares the two before it passes any struct
 * (speechcatcher_amd/_abi.py does at load time, the C hosts in tests/ at start-up). */
#define SC_ABI_VERSION 6
int sc_version(void);

/* hipGraph capture / replay of any sequence of the launches below on a
 * non-default stream: begin, issue the launches, end -> executable graph. */
int sc_graph_capture_begin(void *stream);
int sc_graph_capture_end(void *stream, void **graph_exec /*HOST out*/);
int sc_graph_launch(void *graph_exec, void *stream);
int sc_graph_destroy(void *graph_exec);

/* ---- generic building blocks -------------------------------------------- */

/* C[c_rows[m], n] (+)= sum_k A[a_rows[m]*lda + kofs(k)] * W[n*K + k] + bias[n]
 * (torch.nn.Linear call sites: multi_head_attention.py:81-83,133,
 * feed_forward.py:50, subsampling.py:98, ctc.py:40, transformer_decoder.py:249;
 * with conv_f1 > 0 the second Conv2d of subsampling.py:87-93 as an implicit
 * GEMM: lda = channels, tap = k / lda, kofs = ((tap/3)*conv_f1 + tap%3)*lda + k%lda).
 * f32 MFMA (v_mfma_f32_32x32x2_f32), K must be a multiple of 32. */
int sc_gemm(const float *A, const int32_t *a_rows, int lda, const float *W, const float *bias,
            float *C, const int32_t *c_rows, int ldc, int M, int N, int K, int flags, int conv_f1,
            void *stream);

/* sc_gemm followed by LayerNorm of the produced rows (pre-LN transformer:
 * x += proj(...); xn = LN(x): decoder_layer.py:101-123, transformer_decoder.py:243).
 * ln_out[m*ld_ln ..] = LN(C[c_rows[m]]).  The LayerNorm is fused into the
 * split-K reduce when the workspace is set. */
int sc_gemm_ln(const float *A, const int32_t *a_rows, int lda, const float *W, const float *bias,
               float *C, const int32_t *c_rows, int ldc, int M, int N, int K, int flags, int conv_f1,
               const float *ln_g, const float *ln_b, float ln_eps, float *ln_out, int ld_ln,
               void *stream);

/* Row-panel form of "attention output projection + residual + LayerNorm
 * (+ next projection)" for the decoder layer (decoder_layer.py:101-123:
 * x = residual + self_attn(..); x = norm2(x); src_attn's linear_q,
 * multi_head_attention.py:58-60):
 *   X[m] += A[m] . W1^T + b1;  XN[m] = LN(X[m]) (if XN);  Q[m] = LN(X[m]) . W2^T + b2 (if W2).
 * One workgroup owns 4, 8 or 16 complete rows (D = 64, 128 or 256), so the LayerNorm
 * and the second projection need no second launch.  v_mfma_f32_4x4x1_16B_f32.
 * A must be 16-byte aligned with lda % 4 == 0.  rows (optional): the M rows of
 * A / X / XN / Q to process (NULL: rows 0..M-1). */
int sc_proj_ln_proj(const float *A, int lda, const float *W1q, const float *b1, float *X, int ldx,
                    const float *ln_g, const float *ln_b, float ln_eps, float *XN, int ldn,
                    const float *W2q, const float *b2, float *Q, int ldq, const int32_t *rows,
                    int M, int D, void *stream);
int sc_proj_ln_proj_supported(int D);
/* W1q / W2q: the [D][D] Linear weights re-ordered ONCE so that lane l of a wave reads
 * W[64*tile + l][4*q .. 4*q+3] as one 16-byte element (N % 64 == 0, K % 4 == 0):
 *   out[((tile*(K/4) + q)*64 + lane)*4 + c] = W[tile*64 + lane][4*q + c]            */
int sc_pack_lane_weight(const float *W, int N, int K, float *out, void *stream);
/* Fragment order of the 16x16x4-MFMA kernels (sc_ffn_ln; N % 16 == 0, K % 32 == 0):
 *   out[((((tile*(K/32) + ki)*2 + half)*64 + lane)*4 + c]
 *       = W[tile*16 + lane%16][ki*32 + 8*(lane/16) + 4*half + c]              */
int sc_pack_panel_weight(const float *W, int N, int K, float *out, void *stream);

/* Fused position-wise feed-forward (feed_forward.py:48-50 + the residual of the
 * encoder / decoder layer, + the LayerNorm that follows when ln_out != NULL):
 *   X[r] += W2 . relu(W1 . XN[r] + b1) + b2;  ln_out[r] = LN(X[r])   for r in rows[0..M)
 * (rows NULL: r = 0..M-1; ln_out may alias XN).  The hidden activations stay in
 * LDS; W1p [F][D] and W2p [D][F] are sc_pack_panel_weight copies.  Needs the
 * split-K workspace (sc_set_workspace / sc_set_stream_workspace). */
int sc_ffn_ln(const float *XN, const int32_t *rows, int M, int D, int F, const float *W1p,
              const float *b1, const float *W2p, const float *b2, float *X, const float *ln_g,
              const float *ln_b, float ln_eps, float *ln_out, void *stream);
int sc_ffn_ln_supported(int D, int F);
/* Row-tile projection with the LayerNorms around it folded in (K = D):
 *   C[r] = LN(A[r]; ln_g, ln_b) . W^T + bias            (ln_g NULL: A[r] as is)
 *   R != NULL or LN2 != NULL (needs N == D):  C[r] += R[r];  LN2[r] = LN(C[r]; g2, b2)
 * the encoder layer's q|k|v Linear behind norm1 (multi_head_attention.py:63-90,
 * contextual_block_encoder_layer.py:190-215) and its attention output Linear + residual
 * followed by norm2 (:216-240), one launch each.  W [N][D] in sc_pack_panel_weight order,
 * N a multiple of 128, D in {128, 256}; R may be C.  LN2 has leading dimension D. */
int sc_rowtile_proj(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                    float eps, const float *Wp, const float *bias, int N, const float *R, float *C,
                    int ldc, const float *g2, const float *b2, float *LN2, void *stream);
/* ... with fp16 weights (Wh = the fragment-packed copy with 2-byte elements) and fp16 MFMA inputs: BASELINE configs[4] */
int sc_rowtile_proj_h(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                      float eps, const void *Wh, const float *bias, int N, const float *R, float *C,
                      int ldc, const float *g2, const float *b2, float *LN2, void *stream);
/* ... with the fp16 hi | lo split of the fp32 weights (see sc_ffn_ln_s): fp32-grade results from fp16 MFMA inputs */
int sc_rowtile_proj_s(const float *A, int lda, int M, int D, const float *ln_g, const float *ln_b,
                      float eps, const void *Ws, const float *bias, int N, const float *R, float *C,
                      int ldc, const float *g2, const float *b2, float *LN2, void *stream);
int sc_rowtile_proj_supported(int D, int N);
/* The same feed-forward followed by the projection that consumes its LayerNorm - the next decoder
 * layer's Q|K|V (decoder_layer.py:85-100 of layer l+1) or the output layer
 * (transformer_decoder.py:243-249) - with the split-sum reduce, residual, LayerNorm and projection
 * in ONE row-panel launch:
 *   Xout[r] = Xin[r] + FFN(XN[r]);  ln_out[r] = LN(Xout[r]) (optional);  Q[r] = LN(Xout[r]) . Wq^T + bq
 * Wq [N][D] in sc_pack_lane_weight order, N a multiple of D, Q leading dimension N.  Xin and Xout
 * must be different buffers.  Needs a workspace that holds all M rows (else returns an error). */
int sc_ffn_ln_proj(const float *XN, const int32_t *rows, int M, int D, int F, const float *W1p,
                   const float *b1, const float *W2p, const float *b2, const float *Xin, float *Xout,
                   const float *ln_g, const float *ln_b, float ln_eps, float *ln_out, const float *Wq,
                   const float *bq, float *Q, int N, void *stream);
/* ... with fp16 weights: W1h / W2h = the fragment-packed copies with 2-byte elements; fp16 MFMA inputs (activations
 * rounded to fp16 when staged), fp32 accumulation, bias, ReLU input, partial sums, residual and LayerNorm in fp32 */
int sc_ffn_ln_h(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1h, const float *b1,
                const void *W2h, const float *b2, float *X, const float *ln_g, const float *ln_b, float ln_eps,
                float *ln_out, void *stream);
int sc_ffn_ln_proj_h(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1h, const float *b1,
                     const void *W2h, const float *b2, const float *Xin, float *Xout, const float *ln_g,
                     const float *ln_b, float ln_eps, float *ln_out, const float *Wq, const float *bq, float *Q, int N,
                     void *stream);
/* ... with the fp16 hi | lo SPLIT of the fp32 weights (W1s / W2s: weights.py split_panel_weight - per lane and 32-wide
 * k block the fp16 roundings of its 8 k values in the first slab of the fragment order, fp16((w - hi) * 2^11) in the
 * second; the same bytes as the fp32 copies): every product sum is evaluated as
 * sum(a_hi w_hi) + 2^-11 (sum(a_hi w_lo) + sum(a_lo w_hi)) with fp16 MFMA inputs and fp32 accumulation, the
 * activations split the same way when staged.  Differs from the fp32 kernel by a few fp32 ulp (the dropped
 * a_lo w_lo terms: <= 2^-22 relative), at ~1/5 of its matrix-pipe cycles.  Operands must lie within fp16's range. */
int sc_ffn_ln_s(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1s, const float *b1,
                const void *W2s, const float *b2, float *X, const float *ln_g, const float *ln_b, float ln_eps,
                float *ln_out, void *stream);
int sc_ffn_ln_proj_s(const float *XN, const int32_t *rows, int M, int D, int F, const void *W1s, const float *b1,
                     const void *W2s, const float *b2, const float *Xin, float *Xout, const float *ln_g,
                     const float *ln_b, float ln_eps, float *ln_out, const float *Wq, const float *bq, float *Q, int N,
                     void *stream);
/* bytes of the split-K workspace registered for `stream` (0: none).  sc_decoder_layers /
 * sc_encoder_layers use the fused FFN only when a workspace is available and fall back to
 * two GEMMs otherwise. */
size_t sc_workspace_bytes(void *stream);

/* Workspace (device memory, caller-owned) for the deterministic split-K path of
 * sc_gemm: partial sums [ksplit][M][N] reduced in fixed order.  Without it
 * sc_gemm never splits K. */
int sc_set_workspace(void *ptr, size_t bytes);
/* per-stream workspace (several StreamBatches running concurrently on their own
 * streams must not share partial-sum storage); ptr == NULL unregisters */
int sc_set_stream_workspace(void *stream, void *ptr, size_t bytes);

/* Optional per-launch timing of sc_gemm with HIP events on the launch stream
 * (every `sample_every`-th launch; 0 disables).  sc_prof_collect synchronises
 * and returns, per kernel variant v (0 scalar, 1 = 32x128 tile, 2 = 128x128,
 * 3 = 64x64): summed milliseconds, summed algorithmic flops (2*M*N*K) and the
 * number of sampled launches (variant 1 = skinny register-direct kernel for M <= 64).  All three pointers are HOST arrays of 4. */
/* kernel kinds of the per-launch timing records */
#define SC_PROF_GEMM_NAIVE 0
#define SC_PROF_GEMM_SKINNY 1
#define SC_PROF_GEMM_128 2
#define SC_PROF_GEMM_64 3
#define SC_PROF_PROJ_LN_PROJ 4
#define SC_PROF_FFN_FUSED 5
#define SC_PROF_ATTN_SELF 6
#define SC_PROF_ATTN_CROSS 7
#define SC_PROF_ROWTILE_PROJ 8
#define SC_PROF_FFN_PRO 9 /* ffn_fused_kernel<.., PRO>: sc_dec_layer_ffn (head-partial reduce + norm3 prologue) */
#define SC_PROF_LAYER_SELF 10  /* dec_layer_attn_kernel<.., SELF>: sc_dec_layer_self (head-parallel layer, small buckets) */
#define SC_PROF_LAYER_CROSS 11 /* dec_layer_attn_kernel<.., cross>: sc_dec_layer_cross */
#define SC_PROF_LAYER_STREAM 12 /* dec_layer_stream_kernel: sc_dec_layer_stream (stream-resident layer, large buckets; round 6) */
#define SC_PROF_KINDS 13
/* decoder layers run as head-parallel launches (sc_dec_layer_*) for compaction buckets up to this many rows */
#define SC_FUSED_MAX_ROWS 960
/* ... with one head per workgroup; from SC_HPW_MIN_ROWS rows on with FOUR heads per workgroup (1024 threads: the
 * prologue reduce + LayerNorm once per four heads, H/4 partial products per row).  Measured on the default bench
 * (continuous, tools/ab_hpw.sh, SC_HPW_MIN = never / 0 / 160 / 320 / 640 / 800 / 960): 2967 / 3109 / 3074 / 3024 / 3094 /
 * 3055 / 3035 audio-s/s; strict lock-step, where the medium buckets count (tools/ab_hpw_strict.sh, 96 / 320 / 480 / 640 /
 * 960 / never): 2038 / 2137 / 2159 / 2126 / 2195 / 2151 - four heads from half of a 128-stream batch on */
#define SC_HPW_MIN_ROWS 640
/* (round 6) ... and from SC_STREAM_MIN_ROWS rows on in the STREAM-RESIDENT form (decoder_stream.hip: one 1024-thread workgroup per
 * stream runs both attentions of a layer for all heads, two launches per layer): a full 128-stream bucket then holds 128 of the
 * 256 compute units and the encoder groups run beside the decode chain instead of between its kernels.  Same bits as the
 * other two forms (canonical summation, common.h).  SC_STREAM_FFN_CUS: compute units its feed-forward launch is sized for. */
#define SC_STREAM_MIN_ROWS 1281
#define SC_STREAM_FFN_CUS 256
int sc_prof_collect_kinds(double *ms, double *flops, double *bytes, long long *n, int nkinds);
int sc_prof_enable(int sample_every);
int sc_prof_collect(double *ms, double *flops, long long *n);
/* same, plus summed ALGORITHMIC bytes (A + W read once, C written; C read too with SC_GEMM_RESIDUAL) */
int sc_prof_collect2(double *ms, double *flops, double *bytes, long long *n);
/* median duration (ms) of an empty event pair on `stream`: the bias that event
 * bracketing adds to each sampled launch */
double sc_prof_event_overhead_ms(void *stream);

/* LayerNorm over rows (model/layers/normalization.py:7-24, eps 1e-12). */
int sc_layernorm(const float *src, const int32_t *src_rows, int lds, float *dst,
                 const int32_t *dst_rows, int ldd, int M, int d, const float *gamma,
                 const float *beta, float eps, void *stream);

int sc_copy_rows(const float *src, const int32_t *src_rows, float *dst, const int32_t *dst_rows,
                 int n, int width, void *stream);

/* in-place log_softmax of selected rows (scorers.py:133-134, first block only) */
int sc_log_softmax_rows(float *x, const int32_t *rows, int n, int V, void *stream);

/* ---- frontend ------------------------------------------------------------ */

/* STFTFrontend.forward (model/frontend/stft_frontend.py:87-154) fused with the
 * global MVN and frame trimming of Speech2TextStreaming.apply_frontend
 * (speech2text_streaming.py:355-389).  jobs[j] = {stream, seg_start, seg_len,
 * eff_len, keep_lo, keep_n, dst_row0, 0}.  mvn_mode: 0 none, 1 fp32, 2 fp64. */
int sc_logmel(const float *pcm, int pcm_stride, const int32_t *jobs, int n_jobs, int max_keep,
              const float *window, const float *mel_fb, const float *twiddle, const double *mean,
              const double *stdv, int mvn_mode, int n_fft, int hop, int win, int n_mels,
              float *feat, void *stream);

/* ---- encoder --------------------------------------------------------------- */

/* first Conv2d(1->d,3,stride 2)+ReLU (subsampling.py:87-93), channels-last out.
 * jobs[j] = {src_row0, T_in, c1_row0, T1}. */
int sc_conv1(const float *feat, int n_mels, const int32_t *jobs, int n_jobs, int max_t1,
             const float *w, const float *b, int d, float *c1, void *stream);

/* block assembly (contextual_block_transformer_encoder.py:354-380).
 * jobs[b] = {src_row0, chunk_len, pe_off_frames, pe_off_ctx, short, 0}. */
int sc_block_pack(const float *sub, const int32_t *jobs, int nb, int R, const float *pe, int d,
                  float *xblk, void *stream);

/* context hand-off (contextual_block_encoder_layer.py:253-269).
 * jobs[j] = {b0, nblk, state_row, has_state}; state row used = state_row + layer. */
int sc_ctx_handoff(float *x, int R, const int32_t *jobs, int ns, float *state, int layer, int d,
                   void *stream);

/* masked block self-attention (multi_head_attention.py:92-133 with the mask of
 * contextual_block_transformer_encoder.py:524-528). */
int sc_enc_attention(const float *qkv, float *att, int nblk, int R, int H, int d, int masked,
                     void *stream);

/* all encoder layers on (nblk, R, d) blocks (contextual_block_encoder_layer.py:178-271). */
int sc_encoder_layers(const sc_enc_layer *layers /*HOST*/, int n_layers, float *x, int nblk, int R,
                      int masked, const int32_t *jobs, int ns, float *past_ctx, float *xn,
                      float *qkv, float *att, float *ffh, int d, int H, int F, float eps,
                      void *stream);

/* ---- search ---------------------------------------------------------------- */

/* CTCPrefixScoreTH.extend_state (ctc_prefix_score_full.py:349-368) */
int sc_ctc_extend_state(const sc_search *sb /*HOST*/, void *stream);
/* embed*sqrt(d)+PE of the newest token (transformer_decoder.py:231) */
int sc_dec_embed(const sc_search *sb, void *stream);
/* kv_half: cross-attention K|V rows of all layers from an fp32 staging buffer [n_layers][m][d2] (the output of
 * the K|V projection GEMMs for m new encoder frames) into the fp16 cache: row j of layer li -> row
 * rows[j] + li*TCAP. */
int sc_kv_rows_to_half(const float *stage, const int32_t *rows, int m, int n_layers, int TCAP, int d2,
                       void *ckv_half, void *stream);
/* decoder self-attention with K/V cache + ancestor table (decoder_layer.py:85-101) */
int sc_dec_self_attn(const sc_search *sb, int layer, void *stream);
/* decoder cross-attention over the shared per-stream K/V (decoder_layer.py:106-115) */
int sc_dec_cross_attn(const sc_search *sb, int layer, void *stream);
int sc_decoder_layers(const sc_search *sb, void *stream);
/* log_softmax + pre-beam top-K (transformer_decoder.py:249, beam_search.py:150-154) */
int sc_logsoftmax_topk(const sc_search *sb, void *stream);
/* CTCPrefixScoreTH.__call__ on the K candidates (ctc_prefix_score_full.py:88-291) */
int sc_ctc_prefix_scan(const sc_search *sb, void *stream);
/* the same scores; streams with at least split_min (> 0) frames behind their first scanned frame are walked by 16
 * threads per (hypothesis, candidate) - segment-wise affine maps of the recurrence, combined, then re-walked - instead
 * of one: for long tables (T = 4500: a 180 s segment) while few streams are active */
int sc_ctc_prefix_scan_split(const sc_search *sb, int split_min, void *stream);
/* score fusion + per-hypothesis top-W (beam_search.py:113-185,723) */
int sc_fuse_topw(const sc_search *sb, void *stream);
/* expand / prune / bookkeeping / stop flags (beam_search.py:721-809, hypothesis.py:132-142) */
int sc_beam_prune(const sc_search *sb, void *stream);
/* CTCPrefixScorer.select_state (scorers.py:382-431) */
int sc_ctc_gather_state(const sc_search *sb, void *stream);
/* ... behind sc_ctc_prefix_scan_split(sb, split_min): the streams whose scan was split over T (the same rule: at least
 * split_min frames to walk) left the start states of their CTC_NSEG = 32 segments instead of 16-frame checkpoints */
int sc_ctc_gather_state_split(const sc_search *sb, int split_min, void *stream);
/* one full beam-search step = all of the above in order (beam_search.py:701-758) */
int sc_decode_step(const sc_search *sb, void *stream);
/* ... with the CTC prefix scan of streams that have >= scan_split_min frames to walk split over T
 * (sc_ctc_prefix_scan_split; 0: never) */
int sc_decode_step_ex(const sc_search *sb, int scan_split_min, void *stream);
/* ---- head-parallel decoder layers: 3 launches per layer (decoder_layer.py:80-132) --------------------
 * The residual stream x ping-pongs between two [S*W][d] buffers (x_in != x_out in every call): sibling
 * workgroups read x_in while the owner of a row writes x_out.  Grid (stream, head) for the two attention
 * launches; every workgroup sums the producer's partial products itself (fixed order) and recomputes the
 * LayerNorm of its stream's W rows, so neither the LayerNorms nor the attention output projections need
 * launches of their own.  d in {128, 256}, head dim in {16, 32} and (round 6, d = 256) 64 - the reference's geometry when
 * config.yaml names no heads (speech2text_streaming.py:221-227) -, W <= 16, sc_ffn_ln_supported(d, F). */
int sc_dec_layer_fused_supported(int d, int H, int W, int F);
/* A: x = x_in + b2[layer-1] + sum_z ffn_part[z][row]  (layer 0: embed*sqrt(d)+PE, transformer_decoder.py:231;
 * x_in / ffn_part unused) -> x_out; q|k|v = norm1(x) . Wqkv^T + b of every head; K|V row appended to the
 * cache; self-attention through the ancestor table (decoder_layer.py:85-101); ph1 = per-head partial of
 * self_attn.linear_out. */
int sc_dec_layer_self(const sc_search *sb, int layer, const float *x_in, float *x_out, const float *ffn_part,
                      int n_ffn_part, void *stream);
/* B: x = x_in + bo + sum_h ph1 -> x_out; q = norm2(x) . Wq^T + bq; cross-attention over the stream's shared
 * encoder K|V (decoder_layer.py:106-115); ph2 = per-head partial of src_attn.linear_out. */
int sc_dec_layer_cross(const sc_search *sb, int layer, const float *x_in, float *x_out, void *stream);
/* C: x = x_in + bo2 + sum_h ph2 -> x_out; feed-forward of norm3(x) (decoder_layer.py:117-123,
 * feed_forward.py:48-50) as partial sums ffn_part[z][row], z < *n_part (HOST out; <= max_part). */
int sc_dec_layer_ffn(const sc_search *sb, int layer, const float *x_in, float *x_out, float *ffn_part,
                     int max_part, int *n_part /*HOST*/, void *stream);
/* ---- stream-resident decoder layers: 2 launches per layer (round 6; decoder_layer.py:80-132) -------------------
 * d = 256, 8 heads of 32, W <= 10, fp32 weights.  One workgroup per stream: no partial products between the attentions. */
int sc_dec_layer_stream_supported(int d, int H, int W, int F);
/* A': x = x_in + b2[layer-1] + sum_z ffn_part[z][row] (layer 0: embed*sqrt(d)+PE); self-attention block (norm1, q|k|v of all
 * heads, K|V row appended, attention through the ancestor table, linear_out + residual: decoder_layer.py:85-101);
 * cross-attention block (norm2, q, attention over the stream's encoder K|V, linear_out + residual: :106-115) -> x_out;
 * xn_out = norm3(x_out) (:117). */
int sc_dec_layer_stream(const sc_search *sb, int layer, const float *x_in, float *x_out, float *xn_out,
                        const float *ffn_part, int n_ffn_part, void *stream);
/* C': feed-forward of the rows xn (feed_forward.py:48-50) as partial sums ffn_part[z][row] by row id, z < *n_part. */
int sc_dec_layer_ffn_xn(const sc_search *sb, int layer, const float *xn, float *ffn_part, int max_part,
                        int *n_part /*HOST*/, void *stream);
/* B2 (round 6; the four-head form at large buckets): x = x_in + bo2 + sum_h ph2 -> x_out, xn_out = norm3(x) once per row
 * (decoder_layer.py:113-117) - the feed-forward then runs without prologue (sc_dec_layer_ffn_xn) */
int sc_dec_layer_reduce_ln(const sc_search *sb, int layer, const float *x_in, float *x_out, float *xn_out, void *stream);
/* tail: x = x_in + b2[last] + sum_z ffn_part -> x_out; logits = after_norm(x) . out_w^T + out_b
 * (transformer_decoder.py:243-249); needs sb->out_w_q. */
int sc_dec_output_logits(const sc_search *sb, const float *x_in, float *x_out, const float *ffn_part,
                         int n_ffn_part, void *stream);

/* ---- Conformer building blocks (north-star components, unused by the shipped
 * contextual-block Transformer models: SURVEY.md section 8(f) rank 3) ---------- */

/* GLU -> depthwise Conv1d(ksize, same padding) -> BatchNorm1d(eval) -> Swish of
 * ConvolutionModule.forward (model/layers/convolution.py:103-112); the two
 * pointwise convolutions around it are sc_gemm calls.  y [B][T][2C] -> out [B][T][C]. */
int sc_glu_dwconv_bn_swish(const float *y, int B, int T, int C, int ksize, const float *dw_w,
                           const float *dw_b, const float *bn_g, const float *bn_b,
                           const float *bn_mean, const float *bn_var, float bn_eps, float *out,
                           void *stream);

/* RelPositionMultiHeadedAttention.forward core (model/attention/multi_head_attention.py:
 * 343-378, rel_shift :300-314): qkv [B*T][3d] projected q|k|v, p [T][d] = linear_pos(pos_emb),
 * bias_u / bias_v [H][dk] -> out [B*T][d] (before linear_out).  Any T. */
int sc_relpos_attention(const float *qkv, const float *p, const float *bias_u, const float *bias_v,
                        float *out, int B, int T, int H, int d, void *stream);
/* ... with the reference's attention masks (:366-372): mask_mode 1 = mask [B][T] over keys (batch, 1, time_k),
 * 2 = mask [B][T][T] (batch, time_q, time_k), bytes, 0 = masked out; a fully masked query row gives zeros. */
int sc_relpos_attention_masked(const float *qkv, const float *p, const float *bias_u, const float *bias_v,
                               float *out, int B, int T, int H, int d, const uint8_t *mask, int mask_mode,
                               void *stream);

/* ==== stream-level API: the decoder as a C library ==========================================
 * What a reference maintainer binds instead of Speech2TextStreaming's torch modules:
 *   engine  = one weight replica on one GPU            (load_model, speechcatcher.py:126-227)
 *   streams = S independent recognitions sharing it    (S x Speech2TextStreaming, speech2text_streaming.py:43-263)
 *   sc_push = Speech2TextStreaming.__call__ for the listed streams, batched   (:402-539)
 * The library owns all device memory of these objects.  Handles are not thread-safe: one host thread and
 * one HIP stream per sc_streams.  Functions return SC_OK / a negative code and never throw. */
typedef struct sc_config {
  int32_t d_model, enc_heads, enc_layers, dec_heads, dec_layers, ffn_dim, vocab_size;
  int32_t n_mels, n_fft, win_length, hop_length, sample_rate;
  int32_t block_size, hop_size, look_ahead, subsample, conv_freq1, conv_freq2;
  int32_t blank_id, sos_id, eos_id, pe_max_len;
  int32_t mvn_mode; /* 0: no global MVN, 1: fp32 statistics, 2: fp64 statistics */
  float ln_eps;
} sc_config;

typedef struct sc_named_tensor {
  const char *name; /* PackedWeights names: "window", "conv2_w", "enc.3.wqkv_p", "dec.0.wkv", ... */
  const void *data; /* DEVICE pointer, borrowed for the lifetime of the engine */
  int64_t numel;
  int32_t dtype; /* 0 = f32, 1 = f64 (mean64 / std64), 2 = f16 (w1_h / w2_h) */
} sc_named_tensor;

typedef struct sc_stream_options {
  int32_t n_streams, beam_size;
  float ctc_weight;          /* 0.3 (speechcatcher.py:221) */
  int32_t use_bbd;           /* block boundary detection (beam_search.py:771-790) */
  int32_t max_frames;        /* encoder frames per utterance (capacity) */
  int32_t max_tokens;        /* tokens per hypothesis incl. sos (capacity) */
  int32_t pcm_capacity;      /* samples buffered per stream (0: 1 << 20) */
  int32_t max_chunk_samples; /* longest single call (0: 32768) */
  int32_t strict_reference;  /* reset() leaves the stale CTC table / PE counter like the reference (scorers.py:342-350) */
  int32_t kv_half;           /* K|V caches in fp16 (fp32 arithmetic): BASELINE configs[4]'s storage mode; 0 = fp32 */
  int32_t kv_pool_rows;      /* rows of the self-attention K|V pool per stream and layer (capacity; at most max_tokens x beam =
                                one row per (position, hypothesis)).  0 = default: that maximum - a stream can then never run out
                                of rows before max_tokens - unless it would take more than a quarter of the device's free memory
                                (thousands of streams): then what that budget holds, at least 1.5 x max_tokens + 4 x beam.  A
                                stream that needs more rows than the pool has fails with SC_ERR_CAPACITY at the step that is
                                actually TAKEN (a step that is rolled back or rewound never fails) and is reset */
} sc_stream_options;

typedef struct sc_stream_info_t {
  int32_t enc_frames, processed_block, process_idx, n_hyp, hyp_len, pcm_buffered, frontend_started;
  int64_t decode_steps;
} sc_stream_info_t;

typedef struct sc_engine sc_engine;
typedef struct sc_streams sc_streams;

/* weights already on the device, in the layout of speechcatcher_amd.weights.PackedWeights (Python host) */
int sc_engine_create(const sc_config *cfg, const sc_named_tensor *tensors, int n_tensors, int device, sc_engine **out);
/* ... or from a packed model file written by PackedWeights.save_packed (C / C++ hosts: no Python, no torch) */
int sc_engine_load(const char *packed_model_path, int device, sc_engine **out);
int sc_engine_config(const sc_engine *engine, sc_config *out);
void sc_engine_destroy(sc_engine *engine);

int sc_streams_create(sc_engine *engine, const sc_stream_options *options, sc_streams **out);
void sc_streams_destroy(sc_streams *streams);
/* One chunk step for n streams: stream_ids[i] (each stream at most once per call) gets n_samples[i] float samples
 * in +-1 from pcm[i] (HOST; NULL = already resident in the device PCM buffer, see sc_streams_pcm) with is_final[i].
 * The host chunks are staged in pinned memory and reach the device with ONE copy.  Runs frontend -> encoder ->
 * every decode block that became ready (run to completion).  status[i] (HOST out, may be NULL): 1 output, 0 the
 * reference's early `return []`, SC_ERR_CAPACITY / SC_ERR_INPUT: this stream failed and was reset (message:
 * sc_stream_last_error), the others are unaffected. */
int sc_push(sc_streams *streams, const int *stream_ids, const float *const *pcm, const int *n_samples,
            const uint8_t *is_final, int n, int *status);
/* the same for 2-D (T, n_mels) already-normalised feature matrices (speech2text_streaming.py:438-449) */
int sc_push_features(sc_streams *streams, const int *stream_ids, const float *const *feats, const int *n_frames,
                     const uint8_t *is_final, int n, int *status);
/* Continuous batching - the reference server's concurrency unit is an independent stream that is called, answers,
 * and is called again (recognize_ws / process_audio_chunk, speechcatcher_server.py:205-296,359-397), not a batch in
 * lock-step.  sc_submit hands the engine ONE chunk for each listed stream (copied; a stream has at most one chunk
 * outstanding) and returns at once: the chunks are planned and their frontend + encoder stage is issued as one group
 * on the encoder HIP stream.  sc_poll runs the decode side - one beam-search step per tick for EVERY stream that is
 * inside a block, whichever call it belongs to - until at least min_done outstanding chunks are complete (or none is
 * outstanding) and reports up to max_done of them: done_ids[i] = stream, status[i] as in sc_push.  Returns the
 * number reported, < 0 on error.  A stream's reply is ready when ITS blocks are done; streams that finish early
 * get their next chunk while the stragglers of the previous one are still decoding.  Per stream the blocks, steps
 * and results are those of sc_push (the kernel form of a decode step follows the number of streams in it, and with it the
 * fp32 summation order: scores agree to ~1e-5, a beam cut closer than that can fall either way).  sc_push may be mixed in: it also advances the submitted streams. */
int sc_submit(sc_streams *streams, const int *stream_ids, const float *const *pcm, const int *n_samples,
              const uint8_t *is_final, int n);
int sc_poll(sc_streams *streams, int min_done, int max_done, int *done_ids /*HOST out*/, int *status /*HOST out, may be NULL*/);
int sc_streams_outstanding(const sc_streams *streams);
/* sc_submit policy: a chunk's own decode block only sees frames that were there before it (the block schedule runs one
 * hop behind the encoder, beam_search.py:590-634), so its frontend + encoder stage has a whole chunk period of slack:
 * the stages of successive admissions are merged and issued as ONE group when it holds min_streams streams (default:
 * half of the streams) or as soon as a queued decode block needs its frames.  1: every admission is issued at once.
 * Results do not depend on it in the fp32 form.  With split-precision projections ("split16") the group's row count
 * selects between the tiled GEMM (split operands) and the small-M kernels (fp32) for the subsampling / CTC / cross-K|V
 * projections: both are fp32-grade, results can differ in the last bits of fp32 between group sizes. */
int sc_streams_set_encoder_batch(sc_streams *streams, int min_streams);
/* Chunks a stream may have outstanding at a time (1..8; default 1 = the reference's call -> reply -> next call,
 * speechcatcher_server.py:359-397).  depth > 1: sc_submit accepts the next chunk(s) of a stream while an earlier one
 * is still being decoded, for hosts that already have the audio (a file: the reference CLI's chunk loop,
 * speechcatcher.py:574-592; a backlog): the frontend + encoder of chunk k+1 run beside the decoding of chunk k and the
 * stream does not idle between its reply and its next call.  Per stream the chunks are processed and reported in
 * order, at most one per sc_poll call, each with the results of the one-at-a-time protocol; the hypotheses of a
 * reported chunk are a copy taken when it completed, returned by sc_get_hyps / sc_get_hyps_batch until the NEXT
 * sc_poll call.  Nothing can be queued behind a final chunk; a chunk that fails (at admission or while decoding)
 * fails the chunks queued behind it as well - each is reported in its turn (also those submitted before the last of
 * them has been reported), the stream is reset once, when the chunk that failed is reported.  The depth can only be
 * changed while nothing is outstanding.  sc_push on such a handle works as always (it needs the stream idle). */
int sc_streams_set_queue_depth(sc_streams *streams, int depth);
/* message of the stream's last failure (status < 0 from sc_push / sc_poll); "" if it never failed */
const char *sc_stream_last_error(const sc_streams *streams, int stream);
/* live hypotheses, best first (BeamState.hypotheses: yseq, xpos, score, scores{decoder, ctc}; hypothesis.py).
 * Returns how many were written (<= nbest) or a negative code. */
int sc_get_hyps(sc_streams *streams, int stream, int nbest, int max_len, int32_t *ids, int32_t *xpos, int *lens,
                double *scores, double *score_dec, double *score_ctc);
/* ... of n streams at once - what a batch of Speech2TextStreaming.__call__s hands back (speech2text_streaming.py:
 * 466-539): ONE pack launch and ONE device-to-host copy into pinned memory.  ids / xpos [n][nbest][max_len],
 * lens / scores / score_dec / score_ctc [n][nbest], n_hyps [n]; any output but n_hyps may be NULL. */
int sc_get_hyps_batch(sc_streams *streams, const int *stream_ids, int n, int nbest, int max_len, int32_t *ids,
                      int32_t *xpos, int *lens, int *n_hyps, double *scores, double *score_dec, double *score_ctc);
/* Speech2TextStreaming.reset (speech2text_streaming.py:252-263); not while the stream has a chunk outstanding */
int sc_reset(sc_streams *streams, int stream);
int sc_stream_info(const sc_streams *streams, int stream, sc_stream_info_t *out);
/* rows of the self-attention K|V pool per (stream, layer) this batch was created with (sc_stream_options.kv_pool_rows, or the
 * default: one row per (position, hypothesis) within min(a quarter of the free device memory, SC_KV_POOL_DEFAULT_MAX_GIB)) */
#define SC_KV_POOL_DEFAULT_MAX_GIB 24
int sc_streams_kv_rows(const sc_streams *streams);
int sc_streams_stats(const sc_streams *streams, long *enc_calls, long *dec_steps, long *dec_blocks);
/* measurement aid: encoder-layer hipGraphs captured so far (one per shape of an encoder group) and the host seconds
 * that took */
int sc_streams_capture_stats(const sc_streams *streams, long *n_captures, double *seconds);
/* measurement aids (bench.py): hipGraph replay on/off (off: launches can be bracketed by the sc_prof_* events);
 * encoder K|V rows the cross-attention has read since the last call (returned and cleared) */
int sc_streams_set_graphs(sc_streams *streams, int on);
/* measurement aid (tools/prof_bench.sh): an empty kernel named sc_marker_kernel<id> (id 0..3) on `stream` - brackets a window
 * of the run in a rocprofv3 trace / counter collection, so that the launches of THAT window can be told from the rest */
int sc_marker(int id, void *stream);
/* host seconds the decode step loop spent issuing a step (ctrl upload + graph launch) and waiting for its stop
 * flags, since the last call (returned and cleared) */
int sc_streams_host_times(sc_streams *streams, double *launch_s, double *wait_s);
/* ... and by compaction bucket: seconds[17], iterations[17] (index = bucket size in units of n_streams/16) */
int sc_streams_bucket_times(sc_streams *streams, double *seconds, long *iterations);
long sc_streams_take_xattn_rows(sc_streams *streams);
/* ... split by the kernel that read them (HOST [3]): rows[0] dec_attn_flash, rows[1] sc_dec_layer_cross, rows[2] sc_dec_layer_stream */
int sc_streams_take_xattn_rows_by_kernel(sc_streams *streams, long *rows);
/* all attention counters since the last call (returned and cleared), each summed over decode iterations, active streams
 * and decoder layers, by the form of the decoder layers that did the work - [0] stand-alone attention kernels (six launches per
 * layer) / [1] head-parallel layer kernels / [2] stream-resident layer kernel:
 *   out[0..2] encoder K|V rows the cross-attention read, out[3..5] token positions the self-attention covered (L per
 *   hypothesis: the algorithmic length), out[6..8] DISTINCT self-attention K|V rows read (device counter) */
int sc_streams_take_attn_counters(sc_streams *streams, long *out /*HOST [9]*/);
/* the batch's HIP stream and its device PCM ring [n_streams][capacity] (bench: inputs resident in HBM) */
void *sc_streams_hip_stream(sc_streams *streams);
float *sc_streams_pcm(sc_streams *streams, long *capacity);
/* host <-> device copies: write samples into a stream's device PCM ring; read back the samples buffered by the
 * frontend (frontend_states["waveform_buffer"], speech2text_streaming.py:300-338) and the encoder output so far
 * (beam_search.encoder_buffer).  The read functions return the number of samples / frames. */
int sc_streams_write_pcm(sc_streams *streams, int stream, long offset, const float *host, long n);
long sc_streams_read_pcm_buffer(sc_streams *streams, int stream, float *host, long max_n);
int sc_streams_read_enc(sc_streams *streams, int stream, float *host, int max_frames);

#ifdef __cplusplus
}
#endif
#endif /* SCASR_H */
