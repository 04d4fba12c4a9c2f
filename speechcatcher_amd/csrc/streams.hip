// Stream-level C ABI of libscasr (include/scasr.h: sc_engine_* / sc_streams_* / sc_push / sc_get_hyps /
// sc_reset): the host state machine that turns the kernel-level entry points into a streaming decoder,
// in C++ - no Python in the loop.  It is the host half of the reference's
//   Speech2TextStreaming.__call__ / apply_frontend      speechcatcher/speech2text_streaming.py:278-539
//   ContextualBlockTransformerEncoder.forward_infer     speechcatcher/model/encoder/contextual_block_transformer_encoder.py:241-419
//   BlockwiseSynchronousBeamSearch.process_block / _decode_one_block / reset
//                                                       speechcatcher/beam_search/beam_search.py:343-356,507-838
// for S streams at once: pure integer bookkeeping (the three nested carry-over buffers of SURVEY.md
// Appendix D, the block schedule, the step loop with its stop flags, rollback and rewind) that decides
// WHAT to launch; all arithmetic runs in the HIP kernels.  speechcatcher_amd/engine.py is the same logic in
// Python (it also runs on the CPU spec backend, which is how the logic is checked against the reference
// fixtures without a GPU); tests/test_gpu_native.py holds both to the same fixtures.
//
// Run-to-completion only (every block finishes inside its push: the reference's per-call semantics).
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

#define HIP_TRY(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      sc_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__));  \
      return SC_ERR_LAUNCH;                                                        \
    }                                                                              \
  } while (0)
#define RC_TRY(call)                 \
  do {                               \
    int rc__ = (call);               \
    if (rc__ != SC_OK) return rc__;  \
  } while (0)

struct Tensor {
  void *ptr = nullptr;
  int64_t numel = 0;
  int dtype = 0;  // 0 f32, 1 f64, 2 f16
  bool owned = false;
};

struct StreamFault {
  int stream;
  int code;  // SC_ERR_CAPACITY / SC_ERR_INPUT
  std::string msg;
};

}  // namespace

struct sc_engine {
  sc_config cfg{};
  int device = 0;
  std::map<std::string, Tensor> t;
  std::vector<sc_enc_layer> enc;
  std::vector<sc_dec_layer> dec;
  std::vector<const float *> wkv, bkv;

  const float *f(const std::string &name, bool required = true) const {
    auto it = t.find(name);
    if (it == t.end()) {
      if (required) sc_set_error("sc_engine: tensor '%s' is missing", name.c_str());
      return nullptr;
    }
    return (const float *)it->second.ptr;
  }
  ~sc_engine() {
    for (auto &kv : t)
      if (kv.second.owned && kv.second.ptr) (void)hipFree(kv.second.ptr);
  }
};

namespace {

// host mirror of one stream's scalar state (engine.py: StreamState)
struct St {
  bool fe_started = false;
  long pcm_start = 0, pcm_end = 0;
  bool enc_started = false;
  int fpp = 0, nfeat = 0, upp = 0, nsub = 0;
  bool has_sub = false;
  int n_blocks = 0;
  bool has_addin = false, has_ctx = false;
  int short_pos = 0;
  int T_enc = 0, processed_block = 0, process_idx = 0;
  bool prev_valid = false, started = false;
  int cur = 0, L = 1, nhyp = 1;
  bool has_ctc = false;
  int T_ctc = 0, T_kv = 0, output_index = 0;
  long n_steps_total = 0;
};

template <typename T>
int dalloc(T **p, size_t n) {
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIP_TRY(hipMalloc((void **)p, bytes));
  HIP_TRY(hipMemset(*p, 0, bytes));
  return SC_OK;
}

}  // namespace

struct EncPlan {
  std::vector<int32_t> conv_jobs, a_rows, lin_dst, feat_src, feat_dst, blk_jobs, sjobs, emit_src, emit_dst, sub_src,
      sub_dst;
  struct Short { int s, ubase, U, pe_pos, dst_T; };
  std::vector<Short> shorts;
  int n_conv = 0, max_t1 = 0;
};

// frontend + encoder launches of one push, planned (host state already advanced) but not yet issued
struct PendingEnc {
  bool valid = false;
  std::vector<int32_t> fe_jobs;
  int n_fe = 0, max_keep = 0;
  EncPlan P;
};

struct sc_streams {
  sc_engine *eng = nullptr;
  sc_config cfg{};
  int S = 0, W = 0, K = 0, TCAP = 0, LCAP = 0, FCAP = 0, UCAP = 0, max_feat_new = 0, max_t1 = 0, max_t2 = 0,
      max_blocks = 0, max_chunk = 0, max_length = 500;
  long PCAP = 0;
  bool use_bbd = false, strict = true;
  hipStream_t stream = nullptr;
  // Encoder side of a chunk step on its OWN HIP stream: frontend + encoder of a push do not feed the decode blocks
  // whose frames were already there before the push (the block schedule runs one hop behind the encoder, SURVEY
  // A7), so both proceed concurrently - the encoder's MFMA-bound grids fill the CUs that the latency-bound decode
  // iterations leave idle.  `es` is the stream the current phase launches into.
  hipStream_t stream_enc = nullptr, es = nullptr;
  hipEvent_t ev_enc_done = nullptr;
  hipEvent_t ev_iter[2] = {nullptr, nullptr};   // end of decode iteration k (k & 1)
  int32_t *flags_dev = nullptr, *ring_dev = nullptr;   // stop flags of a step (device); device view of flags_host [2][S]
  int32_t *rm_host[2] = {nullptr, nullptr};     // pinned rowmap images (double-buffered: one may still be in a copy queue)
  int rm_idx = 0;
  bool rm_dirty = false;                        // rm_host[rm_idx] differs from the device rowmap
  int scan_split_min = 256, scan_split_streams = 48;   // T-parallel CTC scan: frames to walk >=, bucket streams <=
  bool speculate = false;                       // device-side step control: enqueue iteration i+1 before reading the
                                                // flags of i (opt-in, graphs on only; measured: no gain - DESIGN 4 (q))
  bool scan_long = false;                       // this step: a live stream has >= scan_split_min frames to walk
  int step_split_min() const { return (scan_long && n_rows_step <= scan_split_streams * W) ? scan_split_min : 0; }
  int graph_key() const { return n_rows_step * 4 + (speculate ? 2 : 0) + (step_split_min() > 0 ? 1 : 0); }
  long spec_launched = 0, spec_wasted = 0;
  bool enc_pending = false;      // the encoder stage of this push has been launched and may still be running
  PendingEnc *pend = nullptr;   // ... has been planned but not launched yet (launched when the decode loop thins out)
  int enc_start_thr = 0;         // launch it when at most this many streams are still in the step loop
  void *ws = nullptr, *ws_enc = nullptr;
  std::vector<void *> owned;
  // device buffers
  float *pcm = nullptr, *featbuf = nullptr, *subbuf = nullptr, *prev_addin = nullptr, *past_ctx = nullptr, *enc = nullptr,
        *c1 = nullptr, *c2 = nullptr, *xblk = nullptr, *ws_xn = nullptr, *ws_qkv = nullptr, *ws_att = nullptr,
        *ws_ffh = nullptr;
  int32_t *jobs_ctx = nullptr, *ctrlmap = nullptr, *arena_dev = nullptr;
  float *kv_stage = nullptr;   // kv_half: fp32 staging [dec_layers][kv_stage_rows][2d] of the K|V projections
  int kv_stage_rows = 0;
  sc_search sb{};
  // pinned host
  int32_t *ctrlmap_host = nullptr, *ctrl0_host = nullptr, *flags_host = nullptr, *arena_host = nullptr;
  size_t arena_cap = 1 << 22, arena_off = 0;
  std::vector<St> st;
  std::vector<int> rowmap_key;
  int row_bucket = 1, n_rows_step = 0;
  bool decode_prepared = false;
  std::map<int, hipGraphExec_t> dec_graphs;
  std::map<std::vector<long>, hipGraphExec_t> enc_graphs;
  long dec_steps = 0, dec_blocks = 0, enc_calls = 0, xattn_rows[2] = {0, 0};   // [0] flash kernels, [1] layer kernels
  double t_launch = 0, t_wait = 0, t_host = 0;   // seconds in the step loop: issuing, waiting for the flags, bookkeeping
  double t_bucket[17] = {0};                     // ... waiting, by compaction bucket (n_rows_step / (row_bucket*W))
  long n_bucket[17] = {0};
  bool use_graphs = true;

  ~sc_streams() {
    for (auto &g : dec_graphs) (void)hipGraphExecDestroy(g.second);
    for (auto &g : enc_graphs) (void)hipGraphExecDestroy(g.second);
    if (stream) (void)sc_set_stream_workspace(stream, nullptr, 0);
    if (stream_enc) {
      (void)sc_set_stream_workspace(stream_enc, nullptr, 0);
      (void)hipStreamDestroy(stream_enc);
    }
    if (ev_enc_done) (void)hipEventDestroy(ev_enc_done);
    for (int i = 0; i < 2; ++i) {
      if (ev_iter[i]) (void)hipEventDestroy(ev_iter[i]);
      if (rm_host[i]) (void)hipHostFree(rm_host[i]);
    }
    delete pend;
    for (void *p : owned) (void)hipFree(p);
    if (ctrlmap_host) (void)hipHostFree(ctrlmap_host);
    if (ctrl0_host) (void)hipHostFree(ctrl0_host);
    if (flags_host) (void)hipHostFree(flags_host);
    if (arena_host) (void)hipHostFree(arena_host);
    if (stream) (void)hipStreamDestroy(stream);
  }

  template <typename T>
  int alloc(T **p, size_t n) {
    RC_TRY(dalloc(p, n));
    owned.push_back(*p);
    return SC_OK;
  }
  int32_t *ctrl_host() { return ctrlmap_host; }              // [S][8]
  int32_t *rowmap_host() { return rm_host[rm_idx]; }          // [S*W]

  // host int table -> device (pinned arena, async copy on the batch's stream; launches that read it are
  // ordered behind the copy; the arena is recycled at the start of every push)
  int itensor(const std::vector<int32_t> &a, const int32_t **out) {
    const size_t n = a.size();
    if (arena_off + n > arena_cap) {
      sc_set_error("sc_push: job-table arena exhausted");
      return SC_ERR_ARG;
    }
    const size_t off = arena_off;
    arena_off = off + ((n + 63) & ~size_t(63));
    if (n) {
      memcpy(arena_host + off, a.data(), n * sizeof(int32_t));
      HIP_TRY(hipMemcpyAsync(arena_dev + off, arena_host + off, n * sizeof(int32_t), hipMemcpyHostToDevice, es));
    }
    *out = arena_dev + off;
    return SC_OK;
  }
};

namespace {

constexpr int F_ANY_EOS = 1, F_BEST_EOS = 2, F_ALL_EOS = 4, F_REPEAT = 8;

void init_hyp(sc_streams *b, int s) {
  // create_initial_hypothesis (hypothesis.py:75-91): yseq=[sos], xpos=[0], scores 0 - side 0, slot 0
  const sc_search &sb = b->sb;
  const int32_t sos = b->cfg.sos_id, zero = 0;
  const double dz = 0.0;
  const size_t o = ((size_t)0 * b->S + s) * b->W + 0;
  (void)hipMemcpyAsync(sb.yseq + o * b->LCAP, &sos, 4, hipMemcpyHostToDevice, b->stream);
  (void)hipMemcpyAsync(sb.xpos + o * b->LCAP, &zero, 4, hipMemcpyHostToDevice, b->stream);
  (void)hipMemcpyAsync(sb.score + o, &dz, 8, hipMemcpyHostToDevice, b->stream);
  (void)hipMemcpyAsync(sb.sc_dec + o, &dz, 8, hipMemcpyHostToDevice, b->stream);
  (void)hipMemcpyAsync(sb.sc_ctc + o, &dz, 8, hipMemcpyHostToDevice, b->stream);
  (void)hipStreamSynchronize(b->stream);   // the sources are stack variables
}

void reset_stream(sc_streams *b, int s) {
  // Speech2TextStreaming.reset + BlockwiseSynchronousBeamSearch.reset (speech2text_streaming.py:252-263,
  // beam_search.py:343-356).  strict: CTCPrefixScorer.impl is never cleared by the reference
  // (scorers.py:342-350: the stale table stays) and the short-segment PE counter keeps counting (A13).
  St old = b->st[s], ns;
  if (b->strict) {
    ns.short_pos = old.short_pos;
    ns.T_ctc = old.T_ctc;
  }
  b->st[s] = ns;
  init_hyp(b, s);
}

// apply_frontend planning (speech2text_streaming.py:300-400)
struct FePlan { bool emit; long seg_start, seg_len, eff_len; int lo, n; };
FePlan plan_frontend(const sc_config &c, St &st, bool is_final) {
  const int win = c.win_length, hop = c.hop_length;
  const long N = st.pcm_end - st.pcm_start;
  const bool first = !st.fe_started;
  const long seg_start = st.pcm_start;
  const int trim = ((win + hop - 1) / hop + 1) / 2;
  FePlan p{false, 0, 0, 0, 0, 0};
  if (!(N > win) && !is_final) {
    st.fe_started = true;
    return p;
  }
  if (is_final) {
    const long eff = N > win ? N : win;
    const int total = 1 + (int)(eff / hop);
    int lo = 0, n = total;
    if (!first && total > trim) { lo = trim; n = total - trim; }
    st.fe_started = false;
    st.pcm_start = st.pcm_end;
    return FePlan{true, seg_start, N, eff, lo, n};
  }
  const long n_frames = (N - (win - hop)) / hop, n_res = (N - (win - hop)) % hop;
  const long proc = (win - hop) + n_frames * hop;
  const int total = 1 + (int)(proc / hop);
  st.pcm_start = st.pcm_end - (win - hop) - n_res;
  st.fe_started = true;
  if (first) {
    const int n = total > trim ? total - trim : total;
    return FePlan{true, seg_start, proc, proc, 0, n};
  }
  if (total > 2 * trim) return FePlan{true, seg_start, proc, proc, trim, total - 2 * trim};
  return p;  // "too short after trimming": frames are lost
}

int compact_pcm(sc_streams *b, int s) {
  St &st = b->st[s];
  const long n = st.pcm_end - st.pcm_start;
  if (st.pcm_start > 0) {
    float *base = b->pcm + (long)s * b->PCAP;
    if (n > 0) {
      // ranges may overlap: go through a temporary device buffer
      float *tmp = nullptr;
      HIP_TRY(hipMalloc((void **)&tmp, n * sizeof(float)));
      HIP_TRY(hipMemcpyAsync(tmp, base + st.pcm_start, n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
      HIP_TRY(hipMemcpyAsync(base, tmp, n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
      HIP_TRY(hipStreamSynchronize(b->stream));
      (void)hipFree(tmp);
    }
    st.pcm_start = 0;
    st.pcm_end = n;
  }
  return SC_OK;
}

struct Chunk { int s; const float *pcm; long n; bool fin; };


// forward_infer planning for the listed streams; pure host state changes.
// Throws StreamFault for per-stream failures.
void encode_plan(sc_streams *b, const std::vector<int> &streams, const std::map<int, int> &feat_new,
                 const std::map<int, bool> &finals, EncPlan &P) {
  const sc_config &c = b->cfg;
  const int S = b->S, F1 = c.conv_freq1, F2 = c.conv_freq2, sub = c.subsample;
  int c1_rows = 0;
  std::map<int, std::array<int, 3>> per;
  for (int s : streams) {
    St &st = b->st[s];
    const bool fin = finals.at(s);
    const int nbuf = st.enc_started ? st.nfeat : 0;
    const int Tf = nbuf + feat_new.at(s);
    const int base = (st.fpp * S + s) * b->FCAP;
    st.enc_started = true;
    int t_use, keep;
    if (fin) {
      t_use = Tf;
      keep = 0;
      if (Tf < 7) {
        char m[256];
        snprintf(m, sizeof m, "Calculated padded input size per channel is smaller than the 3x3 subsampling kernel "
                 "(stream %d: %d feature frames in a final chunk)", s, Tf);   // the reference dies inside Conv2d (A3)
        throw StreamFault{s, SC_ERR_INPUT, m};
      }
    } else {
      const int n_s = Tf / sub - 1;
      if (n_s < 2) {
        st.nfeat = Tf;
        continue;
      }
      keep = Tf % sub + sub * 2;
      t_use = n_s * sub;
    }
    const int t1 = (t_use - 3) / 2 + 1, t2 = (t1 - 3) / 2 + 1;
    P.conv_jobs.insert(P.conv_jobs.end(), {base, t_use, c1_rows, t1});
    P.n_conv++;
    P.max_t1 = std::max(P.max_t1, t1);
    for (int i2 = 0; i2 < t2; ++i2)
      for (int f2 = 0; f2 < F2; ++f2) P.a_rows.push_back(c1_rows * F1 + 2 * i2 * F1 + 2 * f2);
    c1_rows += t1;
    if (keep) {
      const int obase = ((1 - st.fpp) * S + s) * b->FCAP;
      for (int i = 0; i < keep; ++i) {
        P.feat_src.push_back(base + Tf - keep + i);
        P.feat_dst.push_back(obase + i);
      }
      st.fpp = 1 - st.fpp;
    }
    st.nfeat = keep;
    const int nsub = st.has_sub ? st.nsub : 0;
    const int ubase = (st.upp * S + s) * b->UCAP;
    if (nsub + t2 > b->UCAP) throw StreamFault{s, SC_ERR_CAPACITY, "subsampled-frame buffer capacity exceeded"};
    for (int i = 0; i < t2; ++i) P.lin_dst.push_back(ubase + nsub + i);
    per[s] = {t2, nsub, ubase};
  }
  if (!P.n_conv) return;
  const int R = c.block_size + 2;
  const int offset = c.block_size - c.look_ahead - c.hop_size;
  int n_blk_jobs = 0;
  for (int s : streams) {
    auto it = per.find(s);
    if (it == per.end()) continue;
    St &st = b->st[s];
    const bool fin = finals.at(s);
    const int t2 = it->second[0], nsub = it->second[1], ubase = it->second[2];
    const int U = nsub + t2;
    int nb;
    if (fin) {
      nb = (int)std::ceil((double)(U - offset - c.look_ahead) / (double)c.hop_size);
      if (st.n_blocks == 0 && U <= c.block_size) {
        if (st.T_enc + U > b->TCAP) throw StreamFault{s, SC_ERR_CAPACITY, "encoder-frame capacity exceeded (max_frames)"};
        // short-segment path (:345-351): the host state advances at planning time (the launch may come later)
        P.shorts.push_back({s, ubase, U, st.short_pos, st.T_enc});
        st.short_pos += U;   // StreamPositionalEncoding's internal counter (A13)
        st.T_enc += U;
        continue;
      }
    } else {
      if (U <= c.block_size) {
        st.has_sub = true;
        st.nsub = U;
        continue;
      }
      const int overlap = c.block_size - c.hop_size;
      nb = std::max(0, U - overlap) / c.hop_size;
      const int res = U - c.hop_size * nb;
      const int obase = ((1 - st.upp) * S + s) * b->UCAP;
      for (int i = 0; i < res; ++i) {
        P.sub_src.push_back(ubase + U - res + i);
        P.sub_dst.push_back(obase + i);
      }
      st.upp = 1 - st.upp;
      st.has_sub = true;
      st.nsub = res;
    }
    nb = std::max(nb, 0);
    const int b0 = n_blk_jobs;
    for (int i = 0; i < nb; ++i) {
      const int cur_hop = i * c.hop_size;
      const int clen = std::min(c.block_size, U - cur_hop);
      P.blk_jobs.insert(P.blk_jobs.end(), {ubase + cur_hop, clen, cur_hop + c.hop_size * st.n_blocks, i + st.n_blocks, 0, 0});
      ++n_blk_jobs;
    }
    if (nb > 0) {
      P.sjobs.insert(P.sjobs.end(), {b0, nb, s, (int)st.has_addin, (int)st.has_ctx});
      st.has_addin = st.has_ctx = true;
    }
    // output extraction (_extract_output_from_blocks_infer :500-522)
    const bool first = st.n_blocks == 0;
    const int y_len = fin ? (first ? U : U - offset) : nb * c.hop_size + (first ? offset : 0);
    std::vector<int32_t> src(std::max(y_len, 0), -1);
    if (first && nb > 0)
      for (int i = 0; i < offset && i < y_len; ++i) src[i] = b0 * R + 1 + i;
    for (int i = 0; i < nb; ++i) {
      const int cur_hop = i * c.hop_size + (first ? offset : 0);
      const int clen = (i == nb - 1 && fin) ? std::min(c.block_size - offset, y_len - cur_hop) : c.hop_size;
      for (int j = 0; j < clen; ++j)
        if (cur_hop + j >= 0 && cur_hop + j < y_len) src[cur_hop + j] = (b0 + i) * R + 1 + offset + j;
    }
    if (st.T_enc + y_len > b->TCAP) throw StreamFault{s, SC_ERR_CAPACITY, "encoder-frame capacity exceeded (max_frames)"};
    for (int i = 0; i < y_len; ++i) {
      P.emit_src.push_back(src[i]);
      P.emit_dst.push_back(s * b->TCAP + st.T_enc + i);
    }
    st.T_enc += y_len;
    st.n_blocks += nb;
  }
  for (int s : streams)  // final call: next_states = None (:407-408)
    if (finals.at(s)) {
      St &st = b->st[s];
      st.enc_started = false;
      st.nfeat = st.nsub = st.n_blocks = 0;
      st.has_sub = st.has_addin = st.has_ctx = false;
    }
}

int enc_layers_launch(sc_streams *b, int nblk, int R, bool masked, const int32_t *jobs, int ns) {
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  auto launch = [&]() {
    return sc_encoder_layers(e->enc.data(), (int)e->enc.size(), b->xblk, nblk, R, masked ? 1 : 0, jobs, ns, b->past_ctx,
                             b->ws_xn, b->ws_qkv, b->ws_att, b->ws_ffh, c.d_model, c.enc_heads, c.ffn_dim, c.ln_eps,
                             b->es);
  };
  if (!b->use_graphs) return launch();
  // ~6 launches per layer: replayed from a hipGraph keyed by everything that shapes the launch sequence
  std::vector<long> key{nblk, R, (long)masked, (long)(intptr_t)jobs, ns};
  auto it = b->enc_graphs.find(key);
  if (it == b->enc_graphs.end()) {
    if (b->enc_graphs.size() >= 16) return launch();   // ragged callers: do not hoard graphs
    RC_TRY(sc_graph_capture_begin(b->es));
    const int rc = launch();
    void *g = nullptr;
    const int rc2 = sc_graph_capture_end(b->es, &g);
    if (rc != SC_OK) return rc;
    if (rc2 != SC_OK) return rc2;
    it = b->enc_graphs.emplace(key, (hipGraphExec_t)g).first;
  }
  return sc_graph_launch(it->second, b->es);
}

int encode_short(sc_streams *b, const EncPlan::Short &sh) {
  // short-segment path (:345-351): one un-blocked pass, no mask, no context slots, A13 counter
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  const int U = sh.U;
  const int32_t *jd, *src, *dst;
  RC_TRY(b->itensor({sh.ubase, U, sh.pe_pos, 0, 1, 0}, &jd));
  RC_TRY(sc_block_pack(b->subbuf, jd, 1, U, e->f("pe"), c.d_model, b->xblk, b->es));
  RC_TRY(sc_encoder_layers(e->enc.data(), (int)e->enc.size(), b->xblk, 1, U, 0, nullptr, 0, b->past_ctx, b->ws_xn,
                           b->ws_qkv, b->ws_att, b->ws_ffh, c.d_model, c.enc_heads, c.ffn_dim, c.ln_eps, b->es));
  std::vector<int32_t> a(U), d(U);
  for (int i = 0; i < U; ++i) { a[i] = i; d[i] = sh.s * b->TCAP + sh.dst_T + i; }
  RC_TRY(b->itensor(a, &src));
  RC_TRY(b->itensor(d, &dst));
  RC_TRY(sc_layernorm(b->xblk, src, c.d_model, b->enc, dst, c.d_model, U, c.d_model, e->f("enc_norm_g"), e->f("enc_norm_b"),
                      c.ln_eps, b->es));
  return SC_OK;
}

int encode_launch(sc_streams *b, EncPlan &P) {
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  const int d = c.d_model, F1 = c.conv_freq1, F2 = c.conv_freq2, R = c.block_size + 2;
  b->enc_calls++;
  const int32_t *cj, *ar, *ld;
  RC_TRY(b->itensor(P.conv_jobs, &cj));
  RC_TRY(sc_conv1(b->featbuf, c.n_mels, cj, P.n_conv, P.max_t1, e->f("conv1_w"), e->f("conv1_b"), d, b->c1, b->es));
  RC_TRY(b->itensor(P.a_rows, &ar));
  RC_TRY(sc_gemm(b->c1, ar, d, e->f("conv2_w"), e->f("conv2_b"), b->c2, nullptr, d, (int)P.a_rows.size(), d, 9 * d,
                 SC_GEMM_RELU, F1, b->es));
  RC_TRY(b->itensor(P.lin_dst, &ld));
  RC_TRY(sc_gemm(b->c2, nullptr, F2 * d, e->f("sub_out_w"), e->f("sub_out_b"), b->subbuf, ld, d, (int)P.lin_dst.size(), d,
                 F2 * d, 0, 0, b->es));
  if (!P.feat_src.empty()) {
    const int32_t *a, *z;
    RC_TRY(b->itensor(P.feat_src, &a));
    RC_TRY(b->itensor(P.feat_dst, &z));
    RC_TRY(sc_copy_rows(b->featbuf, a, b->featbuf, z, (int)P.feat_src.size(), c.n_mels, b->es));
  }
  const int nbk = (int)P.blk_jobs.size() / 6, ns = (int)P.sjobs.size() / 5;
  if (nbk > 0) {
    if (nbk > b->max_blocks) {
      sc_set_error("sc_push: too many encoder blocks in one call");
      return SC_ERR_ARG;
    }
    const int32_t *bj, *ja;
    RC_TRY(b->itensor(P.blk_jobs, &bj));
    RC_TRY(sc_block_pack(b->subbuf, bj, nbk, R, e->f("pe"), d, b->xblk, b->es));
    std::vector<int32_t> j_add(ns * 4), j_ctx(ns * 4);
    for (int i = 0; i < ns; ++i) {
      const int32_t *sj = &P.sjobs[i * 5];
      j_add[i * 4 + 0] = sj[0]; j_add[i * 4 + 1] = sj[1]; j_add[i * 4 + 2] = sj[2]; j_add[i * 4 + 3] = sj[3];
      j_ctx[i * 4 + 0] = sj[0]; j_ctx[i * 4 + 1] = sj[1]; j_ctx[i * 4 + 2] = sj[2] * c.enc_layers; j_ctx[i * 4 + 3] = sj[4];
    }
    RC_TRY(b->itensor(j_add, &ja));
    RC_TRY(sc_ctx_handoff(b->xblk, R, ja, ns, b->prev_addin, 0, d, b->es));
    // the layer loop reads its job table from a persistent buffer (stable address: it is replayed from a graph)
    const int32_t *jc;
    RC_TRY(b->itensor(j_ctx, &jc));
    HIP_TRY(hipMemcpyAsync(b->jobs_ctx, jc, j_ctx.size() * sizeof(int32_t), hipMemcpyDeviceToDevice, b->es));
    RC_TRY(enc_layers_launch(b, nbk, R, true, b->jobs_ctx, ns));
  }
  if (!P.emit_src.empty()) {
    const int32_t *a, *z;
    RC_TRY(b->itensor(P.emit_src, &a));
    RC_TRY(b->itensor(P.emit_dst, &z));
    RC_TRY(sc_layernorm(b->xblk, a, d, b->enc, z, d, (int)P.emit_src.size(), d, e->f("enc_norm_g"), e->f("enc_norm_b"),
                        c.ln_eps, b->es));
  }
  for (auto &sh : P.shorts) RC_TRY(encode_short(b, sh));
  if (!P.sub_src.empty()) {
    const int32_t *a, *z;
    RC_TRY(b->itensor(P.sub_src, &a));
    RC_TRY(b->itensor(P.sub_dst, &z));
    RC_TRY(sc_copy_rows(b->subbuf, a, b->subbuf, z, (int)P.sub_src.size(), d, b->es));
  }
  return SC_OK;
}

// the host's ctrl rows -> device (block start, and whenever the host, not the advance kernel, decides the next step)
int upload_ctrl(sc_streams *b) {
  HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrlmap_host, (size_t)b->S * 8 * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  return SC_OK;
}
int upload_rowmap(sc_streams *b) {
  HIP_TRY(hipMemcpyAsync(b->ctrlmap + b->S * 8, b->rm_host[b->rm_idx], (size_t)b->S * b->W * sizeof(int32_t),
                         hipMemcpyHostToDevice, b->stream));
  b->rm_dirty = false;
  return SC_OK;
}

// dense decoder kernels process the first n_rows_step entries of rowmap: the hypothesis rows of the streams still
// in the step loop (active first, both parts in stream order), rounded up to a bucket of row_bucket streams
void set_rowmap(sc_streams *b, const std::vector<int> &active) {
  const int S = b->S, W = b->W;
  const int na = (int)active.size();
  const int nb = std::min(S, (na + b->row_bucket - 1) / b->row_bucket * b->row_bucket);
  if (active != b->rowmap_key) {
    b->rowmap_key = active;
    b->rm_idx ^= 1;       // the other image may still be queued for a copy (at most one step back)
    b->rm_dirty = true;
    int32_t *rm = b->rowmap_host();
    std::vector<char> isact(S, 0);
    for (int s : active) isact[s] = 1;
    int k = 0;
    for (int pass = 1; pass >= 0; --pass)
      for (int s = 0; s < S; ++s)
        if (isact[s] == pass)
          for (int h = 0; h < W; ++h) rm[k++] = s * W + h;
  }
  b->n_rows_step = nb * W;
}

// one decode step; with the device-side step control also the advance kernel (ctrl rows of the next step, flags ->
// ring).  Without it the prune kernel stores the stop flags straight into the host-mapped array (sb.flags = ring).
static int step_and_advance(sc_streams *b) {
  b->sb.flags = b->speculate ? b->flags_dev : b->ring_dev;
  // CTC prefix scan split over T while few streams are active (a function of the compaction bucket, so every
  // captured graph has one form): the sequential walk costs 0.15 us per frame whatever the number of streams
  RC_TRY(sc_decode_step_ex(&b->sb, b->step_split_min(), b->stream));
  if (!b->speculate) return SC_OK;
  return sc_step_advance(&b->sb, b->use_bbd ? 1 : 0, b->ring_dev, b->stream);
}

int decode_step_launch(sc_streams *b) {
  b->sb.n_rows = b->n_rows_step;
  if (!b->use_graphs) return step_and_advance(b);
  auto it = b->dec_graphs.find(b->graph_key());
  if (it == b->dec_graphs.end()) {
    RC_TRY(step_and_advance(b));   // warm-up launch (also validates arguments); executes the step
    RC_TRY(sc_graph_capture_begin(b->stream));
    const int rc = step_and_advance(b);
    void *g = nullptr;
    const int rc2 = sc_graph_capture_end(b->stream, &g);
    if (rc != SC_OK) return rc;
    if (rc2 != SC_OK) return rc2;
    b->dec_graphs[b->graph_key()] = (hipGraphExec_t)g;
    return SC_OK;   // the warm-up launch already executed this step
  }
  return sc_graph_launch(it->second, b->stream);
}

int prepare_decode(sc_streams *b) {
  // capture the decode graph of every compaction bucket up front, on dry steps (all ctrl rows inactive)
  memset(b->ctrl_host(), 0, (size_t)b->S * 8 * sizeof(int32_t));
  RC_TRY(upload_ctrl(b));
  const int keep = b->n_rows_step;
  for (int nb = b->row_bucket; nb < b->S + b->row_bucket; nb += b->row_bucket) {
    b->n_rows_step = std::min(nb, b->S) * b->W;
    if (!b->dec_graphs.count(b->graph_key())) RC_TRY(decode_step_launch(b));
  }
  b->n_rows_step = keep;
  return SC_OK;
}

struct Todo { int s, T; bool fin; };
int launch_encoder(sc_streams *b);

// _decode_one_block (beam_search.py:655-838) for a lock-step group of streams
int decode_blocks(sc_streams *b, const std::vector<Todo> &todo, std::vector<StreamFault> &faults) {
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  const int S = b->S, W = b->W, d = c.d_model, Ld = c.dec_layers, V = c.vocab_size;
  const int n = (int)todo.size();
  b->dec_blocks += n;
  std::vector<int> cur(n), L(n), nhyp(n), pidx(n), told(n), tkv(n), Ttab(n), T(n), nhyp_prev(n), out_idx(n, 0);
  std::vector<char> has(n), pvalid(n), fin(n), live(n, 1), took_out(n, 0), has_prev(n);
  std::vector<long> nsteps(n, 0);
  for (int i = 0; i < n; ++i) {
    const St &x = b->st[todo[i].s];
    T[i] = todo[i].T; fin[i] = todo[i].fin;
    cur[i] = x.cur; L[i] = x.L; nhyp[i] = x.nhyp; has[i] = x.has_ctc; pidx[i] = x.process_idx; pvalid[i] = x.prev_valid;
    told[i] = x.T_ctc; tkv[i] = x.T_kv;
    Ttab[i] = std::max(T[i], told[i]);   // the CTC table never shrinks (stale table after reset())
    if (T[i] > b->TCAP) { sc_set_error("sc_push: max_frames exceeded"); return SC_ERR_ARG; }
  }
  // ---- extend_scorers (:403-464): CTC rows, cross-attention K/V rows, r states
  std::vector<int32_t> rows, lsm, krows, kv0;
  bool same_rows = true;
  for (int i = 0; i < n; ++i) {
    const int s = todo[i].s;
    same_rows = same_rows && told[i] == tkv[i];
    for (int t = told[i]; t < T[i]; ++t) rows.push_back(s * b->TCAP + t);
    if (T[i] > told[i] && told[i] == 0)   // quirk A1: only the first block is log-softmaxed
      for (int t = 0; t < T[i]; ++t) lsm.push_back(s * b->TCAP + t);
    for (int t = tkv[i]; t < T[i]; ++t) {
      krows.push_back(s * b->TCAP + t);
      kv0.push_back(s * Ld * b->TCAP + t);
    }
  }
  const int32_t *ar = nullptr;
  if (!rows.empty()) {
    RC_TRY(b->itensor(rows, &ar));
    RC_TRY(sc_gemm(b->enc, ar, d, e->f("ctc_w"), e->f("ctc_b"), const_cast<float *>(b->sb.ctcx), ar, V, (int)rows.size(), V,
                   d, 0, 0, b->stream));
    if (!lsm.empty()) {
      const int32_t *lr;
      RC_TRY(b->itensor(lsm, &lr));
      RC_TRY(sc_log_softmax_rows(const_cast<float *>(b->sb.ctcx), lr, (int)lsm.size(), V, b->stream));
    }
  }
  if (!kv0.empty()) {
    if (!same_rows || !ar) RC_TRY(b->itensor(krows, &ar));
    const int32_t *kvt;
    RC_TRY(b->itensor(kv0, &kvt));   // one row table for all layers: layer li's rows start li*TCAP rows further
    const int m = (int)kv0.size();
    if (!b->sb.kv_half) {
      for (int li = 0; li < Ld; ++li)
        RC_TRY(sc_gemm(b->enc, ar, d, e->wkv[li], e->bkv[li], const_cast<float *>(b->sb.ckv) + (size_t)li * b->TCAP * 2 * d,
                       kvt, 2 * d, m, 2 * d, d, 0, 0, b->stream));
    } else {
      // fp16 cache: project into the fp32 staging buffer (dense rows), then convert + scatter all layers at once
      for (int r0 = 0; r0 < m; r0 += b->kv_stage_rows) {
        const int mm = std::min(b->kv_stage_rows, m - r0);
        for (int li = 0; li < Ld; ++li)
          RC_TRY(sc_gemm(b->enc, ar + r0, d, e->wkv[li], e->bkv[li], b->kv_stage + (size_t)li * mm * 2 * d, nullptr, 2 * d,
                         mm, 2 * d, d, 0, 0, b->stream));
        RC_TRY(sc_kv_rows_to_half(b->kv_stage, kvt + r0, mm, Ld, b->TCAP, 2 * d, const_cast<float *>(b->sb.ckv), b->stream));
      }
    }
  }
  if (!b->decode_prepared) {
    b->decode_prepared = true;
    RC_TRY(prepare_decode(b));
  }
  // block-start ctrl rows (own pinned buffer: every earlier use was followed by a flag read-back sync)
  memset(b->ctrl0_host, 0, (size_t)S * 8 * sizeof(int32_t));
  for (int i = 0; i < n; ++i) {
    int32_t *r = b->ctrl0_host + todo[i].s * 8;
    r[0] = 1; r[1] = cur[i]; r[2] = fin[i]; r[3] = Ttab[i]; r[4] = L[i]; r[5] = nhyp[i]; r[6] = has[i]; r[7] = told[i];
  }
  HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrl0_host, (size_t)S * 8 * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  RC_TRY(sc_ctc_extend_state(&b->sb, b->stream));
  for (int i = 0; i < n; ++i) {
    St &x = b->st[todo[i].s];
    x.T_ctc = Ttab[i];
    x.T_kv = std::max(T[i], tkv[i]);
    x.output_index = 0;
    nhyp_prev[i] = nhyp[i];
    has_prev[i] = has[i];
  }
  int32_t *ctrl = b->ctrl_host();
  memset(ctrl, 0, (size_t)S * 8 * sizeof(int32_t));
  // ---- step loop (:701-821)
  // Device-side step control (opt-in, sc_streams_set_speculation): sc_step_advance behind every step derives the
  // next ctrl rows on the device, so iteration k+1 can be enqueued BEFORE the host has read the stop flags of
  // iteration k ("speculated": same compaction bucket, the streams that stop in k are inactive in k+1 by their device
  // ctrl row); the host reads the flags of k (ring slot L & 1) while k+1 runs.  Not speculated: a step after which a
  // stream would hit max_length / max_tokens (the host decides those), fewer than two live streams (the likely last
  // iteration of a round would run for nothing), buckets that can still shrink, graphs off.  Measured at 128 streams:
  // 3050-3071 with, 3052-3065 audio-s/s without - the host turnaround is not on the critical path - hence off.
  std::vector<int> active;
  bool inflight = false;   // iteration `iter` has already been enqueued
  long iter = 0;
  while (true) {
    active.clear();
    bool any = false;
    for (int i = 0; i < n; ++i) {
      const bool act = live[i] && pidx[i] < b->max_length;
      live[i] = act;
      if (act && L[i] + 1 > b->LCAP) {   // the stream leaves the loop here and is reset by push()
        faults.push_back({todo[i].s, SC_ERR_CAPACITY, "max_tokens exceeded"});
        live[i] = 0;
        continue;
      }
      if (act) { any = true; active.push_back(todo[i].s); }
    }
    if (!any) {
      if (inflight) b->spec_wasted++;   // runs with every stream inactive
      break;
    }
    const auto tp0 = std::chrono::steady_clock::now();
    std::sort(active.begin(), active.end());
    set_rowmap(b, active);
    b->scan_long = false;   // the graph with the T-parallel scan only when a stream's table is long enough for it
    for (int i = 0; i < n && b->scan_split_min > 0; ++i)
      if (live[i] && Ttab[i] - std::max(L[i] - 1, 1) >= b->scan_split_min) b->scan_long = true;
    if (!inflight) {
      for (int i = 0; i < n; ++i) {
        int32_t *r = ctrl + todo[i].s * 8;
        r[0] = live[i]; r[1] = cur[i]; r[2] = fin[i]; r[3] = T[i]; r[4] = L[i]; r[5] = nhyp[i]; r[6] = has[i]; r[7] = Ttab[i];
      }
      // the encoder stage of this push fills the CUs that the thinned-out step loop leaves idle
      if (b->pend->valid && (int)active.size() <= b->enc_start_thr) RC_TRY(launch_encoder(b));
      if (b->rm_dirty) {   // one copy command for both (nothing is in flight here: the pinned image is free)
        memcpy(b->ctrlmap_host + S * 8, b->rm_host[b->rm_idx], (size_t)S * W * sizeof(int32_t));
        HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrlmap_host, (size_t)(S * 8 + S * W) * sizeof(int32_t), hipMemcpyHostToDevice,
                               b->stream));
        b->rm_dirty = false;
      } else {
        RC_TRY(upload_ctrl(b));
      }
      RC_TRY(decode_step_launch(b));
      HIP_TRY(hipEventRecord(b->ev_iter[iter & 1], b->stream));
    }
    for (int i = 0; i < n; ++i)
      if (live[i]) b->xattn_rows[b->n_rows_step > SC_FUSED_MAX_ROWS ? 0 : 1] += (long)T[i] * Ld;   // K|V rows the cross-attention reads this step (bench roofline)
    // (speculating at every size measured 2 % SLOWER than not at all: the compaction bucket then follows the active
    // set one iteration late, which costs what the hidden host turnaround saves - so only where the bucket cannot
    // shrink much any more: the tail of the step loop (at most 8 streams))
    bool spec = b->speculate && b->use_graphs && active.size() >= 2 &&
                (int)active.size() <= std::max(b->row_bucket, std::min(8, S / 4));
    for (int i = 0; i < n && spec; ++i)
      if (live[i] && (pidx[i] + 1 >= b->max_length || L[i] + 2 > b->LCAP)) spec = false;
    if (spec) {   // iteration iter + 1, assuming nothing about iter's outcome
      if (b->pend->valid && (int)active.size() <= b->enc_start_thr) RC_TRY(launch_encoder(b));
      if (b->rm_dirty) RC_TRY(upload_rowmap(b));
      RC_TRY(decode_step_launch(b));
      HIP_TRY(hipEventRecord(b->ev_iter[(iter + 1) & 1], b->stream));
      b->spec_launched++;
    }
    // the stop flags live in host-mapped pinned memory (the advance kernel stores them there: no copy command)
    const auto tp1 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventSynchronize(b->ev_iter[iter & 1]));
    const auto tp2 = std::chrono::steady_clock::now();
    b->dec_steps++;
    b->t_launch += std::chrono::duration<double>(tp1 - tp0).count();
    b->t_wait += std::chrono::duration<double>(tp2 - tp1).count();
    {
      const int bk = std::min(16, (int)((long)b->n_rows_step * 16 / std::max(1, S * W)));
      b->t_bucket[bk] += std::chrono::duration<double>(tp2 - tp0).count();
      b->n_bucket[bk] += 1;
    }
    inflight = spec;
    iter++;
    for (int i = 0; i < n; ++i) {
      if (!live[i]) continue;
      // with the device-side step control: the ring slot of the step that ran at this length
      const int f = b->flags_host[(b->speculate ? (L[i] & 1) * S : 0) + todo[i].s];
      const bool f_any = f & F_ANY_EOS, f_best = f & F_BEST_EOS, f_all = f & F_ALL_EOS, f_rep = f & F_REPEAT;
      out_idx[i] += 1;
      nsteps[i] += 1;
      const bool stop_eos = f_any && (!fin[i] || f_best);
      const bool stop_bbd = b->use_bbd && !stop_eos && f_rep && !fin[i];
      const bool stop_all = !stop_eos && !stop_bbd && f_all && fin[i];
      const bool accept = !(stop_eos || stop_bbd || stop_all);
      const bool take = stop_eos || stop_all || accept;
      if (stop_bbd) out_idx[i] -= 1;
      const int nh_out = std::min(W, nhyp[i] * W);
      if (take) {
        nhyp_prev[i] = nhyp[i];
        has_prev[i] = has[i];
        cur[i] = 1 - cur[i];
        L[i] += 1;
        nhyp[i] = nh_out;
        has[i] = 1;
      }
      if (stop_eos || stop_all) took_out[i] = 1;
      live[i] = accept;
      if (accept) {
        pvalid[i] = 1;   // prev_hyps = copy(H_out)
        pidx[i] += 1;    // process_idx += 1
      }
    }
  }
  // ---- rewind (:827-836)
  for (int i = 0; i < n; ++i) {
    const bool rw = pidx[i] > 1 && pvalid[i];
    const bool r2 = rw && took_out[i];   // live state is a non-accepted H_out: go back to its H_in
    if (r2) {
      cur[i] = 1 - cur[i];
      L[i] -= 1;
      nhyp[i] = nhyp_prev[i];
      has[i] = has_prev[i];
    }
    if (rw) { pidx[i] -= 1; pvalid[i] = 0; }
    St &x = b->st[todo[i].s];
    x.cur = cur[i]; x.L = L[i]; x.nhyp = nhyp[i]; x.has_ctc = has[i];
    x.process_idx = pidx[i]; x.prev_valid = pvalid[i];
    x.output_index = out_idx[i];
    x.n_steps_total += nsteps[i];
  }
  return SC_OK;
}

// t_old (optional): encoder frames every stream had BEFORE this push's encoder stage, which may still be running on
// the encoder stream: a block that only sees those frames does not wait for it
int stage_decode(sc_streams *b, const std::map<int, int> &feat_new, const std::map<int, bool> &finals,
                 std::vector<StreamFault> &faults, const std::vector<int> *t_old = nullptr) {
  // decode schedule (beam_search.py:590-634), rounds of lock-step blocks
  const sc_config &c = b->cfg;
  std::map<int, bool> done_final;
  while (true) {
    std::vector<Todo> todo;
    for (auto &kv : feat_new) {
      const int s = kv.first;
      bool faulted = false;
      for (auto &f : faults) faulted = faulted || f.stream == s;
      if (faulted || done_final.count(s)) continue;
      St &st = b->st[s];
      const int cur_end = c.block_size - c.look_ahead + c.hop_size * st.processed_block;
      const int t_avail = st.T_enc;
      if (t_avail > 0 && cur_end < t_avail) todo.push_back({s, cur_end, false});
      else if (finals.at(s) && t_avail > 0) {
        todo.push_back({s, t_avail, true});
        done_final[s] = true;
      }
    }
    if (todo.empty()) break;
    if (b->enc_pending || b->pend->valid) {
      bool needs_new = t_old == nullptr;
      for (auto &t : todo) needs_new = needs_new || t.T > (*t_old)[t.s];
      if (needs_new) {   // a block of this round sees frames of this push: order the decode stream behind the encoder
        RC_TRY(launch_encoder(b));
        if (b->enc_pending) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_enc_done, 0));
        b->enc_pending = false;
      }
    }
    RC_TRY(decode_blocks(b, todo, faults));
    for (auto &t : todo)
      if (!t.fin) b->st[t.s].processed_block += 1;
  }
  return SC_OK;
}

int stage_encode(sc_streams *b, const std::vector<Chunk> &chunks, std::map<int, int> &feat_new, std::map<int, bool> &finals,
                 std::vector<int> &has_out) {
  const sc_config &c = b->cfg;
  std::vector<int32_t> fe_jobs;
  int n_fe = 0, max_keep = 0;
  for (size_t k = 0; k < chunks.size(); ++k) {
    const Chunk &ch = chunks[k];
    St &st = b->st[ch.s];
    finals[ch.s] = ch.fin;
    if (st.pcm_end + ch.n > b->PCAP) throw StreamFault{ch.s, SC_ERR_CAPACITY, "pcm buffer capacity exceeded"};
    if (ch.n > 0 && ch.pcm)
      if (hipMemcpyAsync(b->pcm + (long)ch.s * b->PCAP + st.pcm_end, ch.pcm, ch.n * sizeof(float), hipMemcpyHostToDevice,
                         b->es) != hipSuccess)
        throw StreamFault{ch.s, SC_ERR_LAUNCH, "copy of the PCM chunk failed"};
    st.pcm_end += ch.n;
    FePlan p = plan_frontend(c, st, ch.fin);
    if (!p.emit) continue;
    if (p.n > b->max_feat_new) throw StreamFault{ch.s, SC_ERR_CAPACITY, "chunk produces more feature frames than max_chunk_samples allows"};
    const int nbuf = st.enc_started ? st.nfeat : 0;
    const int dst_row0 = (st.fpp * b->S + ch.s) * b->FCAP + nbuf;
    fe_jobs.insert(fe_jobs.end(), {ch.s, (int)p.seg_start, (int)p.seg_len, (int)p.eff_len, p.lo, p.n, dst_row0, 0});
    ++n_fe;
    max_keep = std::max(max_keep, p.n);
    feat_new[ch.s] = p.n;
    has_out[k] = 1;
  }
  // plan the encoder BEFORE anything that changes device state is launched: a per-stream fault thrown by the
  // planning leaves the device untouched (the PCM copies above only append behind pcm_end)
  EncPlan P;
  std::vector<int> enc_streams;
  for (auto &kv : feat_new) {
    St &st = b->st[kv.first];
    if (!st.started) st.started = true;   // running_hyps = [initial hypothesis]
    if (kv.second >= 3) enc_streams.push_back(kv.first);   // n < 3: encoder skipped, frames discarded (:551-559)
  }
  if (!enc_streams.empty()) encode_plan(b, enc_streams, feat_new, finals, P);
  PendingEnc &pe = *b->pend;
  pe.valid = n_fe > 0 || P.n_conv > 0;
  pe.fe_jobs = std::move(fe_jobs);
  pe.n_fe = n_fe;
  pe.max_keep = max_keep;
  pe.P = std::move(P);
  return SC_OK;
}

// issue the planned frontend + encoder launches (on the encoder stream when there is one)
int launch_encoder(sc_streams *b) {
  PendingEnc &pe = *b->pend;
  if (!pe.valid) return SC_OK;
  pe.valid = false;
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  b->es = b->stream_enc ? b->stream_enc : b->stream;
  int rc = SC_OK;
  if (pe.n_fe) {
    const int32_t *jobs;
    rc = b->itensor(pe.fe_jobs, &jobs);
    if (rc == SC_OK)
      rc = sc_logmel(b->pcm, (int)b->PCAP, jobs, pe.n_fe, pe.max_keep, e->f("window"), e->f("mel_fb"), e->f("twiddle"),
                     (const double *)e->f("mean64"), (const double *)e->f("std64"), c.mvn_mode, c.n_fft, c.hop_length,
                     c.win_length, c.n_mels, b->featbuf, b->es);
  }
  if (rc == SC_OK && pe.P.n_conv) rc = encode_launch(b, pe.P);
  if (b->stream_enc) {
    (void)hipEventRecord(b->ev_enc_done, b->stream_enc);
    b->enc_pending = true;
  }
  b->es = b->stream;
  return rc;
}

}  // namespace

// =====================================================================================================
extern "C" int sc_engine_create(const sc_config *cfg, const sc_named_tensor *tensors, int n_tensors, int device,
                                sc_engine **out) {
  SC_CHECK_ARG(cfg && tensors && out && n_tensors > 0, "null");
  sc_engine *e = new sc_engine;
  e->cfg = *cfg;
  e->device = device;
  for (int i = 0; i < n_tensors; ++i) {
    Tensor t;
    t.ptr = const_cast<void *>(tensors[i].data);
    t.numel = tensors[i].numel;
    t.dtype = tensors[i].dtype;
    e->t[tensors[i].name] = t;
  }
  for (const char *nm : {"window", "mel_fb", "twiddle", "pe", "mean64", "std64", "conv1_w", "conv1_b", "conv2_w", "conv2_b",
                         "sub_out_w", "sub_out_b", "enc_norm_g", "enc_norm_b", "embed", "dec_norm_g", "dec_norm_b", "out_w",
                         "out_b", "ctc_w", "ctc_b"})
    if (!e->f(nm)) { delete e; return SC_ERR_ARG; }
  e->enc.resize(cfg->enc_layers);
  for (int i = 0; i < cfg->enc_layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    sc_enc_layer &l = e->enc[i];
#define ENC(fld) l.fld = e->f(p + #fld); if (!l.fld) { delete e; return SC_ERR_ARG; }
    ENC(ln1_g) ENC(ln1_b) ENC(wqkv) ENC(bqkv) ENC(wo) ENC(bo) ENC(ln2_g) ENC(ln2_b) ENC(w1) ENC(b1) ENC(w2) ENC(b2)
    ENC(w1_p) ENC(w2_p) ENC(wqkv_p) ENC(wo_p)
#undef ENC
    l.w1_h = e->f(p + "w1_h", false);   // optional fp16 copies: fp16 MFMA inputs in the fused FFN
    l.w2_h = e->f(p + "w2_h", false);
  }
  e->dec.resize(cfg->dec_layers);
  e->wkv.resize(cfg->dec_layers);
  e->bkv.resize(cfg->dec_layers);
  for (int i = 0; i < cfg->dec_layers; ++i) {
    const std::string p = "dec." + std::to_string(i) + ".";
    sc_dec_layer &l = e->dec[i];
#define DEC(fld) l.fld = e->f(p + #fld); if (!l.fld) { delete e; return SC_ERR_ARG; }
#define DEC_OPT(fld) l.fld = e->f(p + #fld, false);
    DEC(ln1_g) DEC(ln1_b) DEC(wqkv) DEC(bqkv) DEC(wo) DEC(bo) DEC(ln2_g) DEC(ln2_b) DEC(wq) DEC(bq) DEC(wo2) DEC(bo2)
    DEC(ln3_g) DEC(ln3_b) DEC(w1) DEC(b1) DEC(w2) DEC(b2) DEC(wo_p) DEC(wq_p) DEC(wo2_p) DEC(w1_p) DEC(w2_p) DEC(wqkv_q)
    DEC_OPT(wqkv_pp) DEC_OPT(wq_pp) DEC_OPT(wo_pp) DEC_OPT(wo2_pp) DEC_OPT(w1_h) DEC_OPT(w2_h)
#undef DEC
#undef DEC_OPT
    e->wkv[i] = e->f(p + "wkv");
    e->bkv[i] = e->f(p + "bkv");
    if (!e->wkv[i] || !e->bkv[i]) { delete e; return SC_ERR_ARG; }
  }
  *out = e;
  return SC_OK;
}

// packed model file (speechcatcher_amd.weights.PackedWeights.save_packed): "SCPK1\0\0\0", int32 sizeof(sc_config),
// sc_config, int32 n, then n x { int32 name_len, name, int32 dtype, int64 numel, data }
extern "C" int sc_engine_load(const char *path, int device, sc_engine **out) {
  SC_CHECK_ARG(path && out, "null");
  FILE *fp = fopen(path, "rb");
  if (!fp) { sc_set_error("sc_engine_load: cannot open %s", path); return SC_ERR_ARG; }
  auto bad = [&](const char *m) { sc_set_error("sc_engine_load: %s (%s)", m, path); fclose(fp); return SC_ERR_ARG; };
  char magic[8];
  int32_t csz = 0, n = 0;
  sc_config cfg{};
  if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "SCPK1\0\0\0", 8) != 0) return bad("not a packed model file");
  if (fread(&csz, 4, 1, fp) != 1 || csz != (int32_t)sizeof(sc_config)) return bad("config size mismatch");
  if (fread(&cfg, sizeof cfg, 1, fp) != 1 || fread(&n, 4, 1, fp) != 1 || n <= 0) return bad("truncated header");
  if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice failed");
  std::vector<std::string> names(n);
  std::vector<sc_named_tensor> tens(n);
  std::vector<void *> ptrs;
  std::vector<char> buf;
  for (int i = 0; i < n; ++i) {
    int32_t nl = 0, dt = 0;
    int64_t numel = 0;
    if (fread(&nl, 4, 1, fp) != 1 || nl <= 0 || nl > 256) return bad("bad tensor name");
    names[i].resize(nl);
    if (fread(&names[i][0], 1, nl, fp) != (size_t)nl || fread(&dt, 4, 1, fp) != 1 || fread(&numel, 8, 1, fp) != 1)
      return bad("truncated tensor header");
    const size_t bytes = (size_t)numel * (dt == 1 ? 8 : dt == 2 ? 2 : 4);
    buf.resize(bytes);
    if (bytes && fread(buf.data(), 1, bytes, fp) != bytes) return bad("truncated tensor data");
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return bad("hipMalloc failed");
    if (bytes && hipMemcpy(p, buf.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return bad("upload failed");
    ptrs.push_back(p);
    tens[i] = sc_named_tensor{names[i].c_str(), p, numel, dt};
  }
  fclose(fp);
  const int rc = sc_engine_create(&cfg, tens.data(), n, device, out);
  if (rc != SC_OK) {
    for (void *p : ptrs) (void)hipFree(p);
    return rc;
  }
  for (auto &kv : (*out)->t) kv.second.owned = true;
  return SC_OK;
}

extern "C" void sc_engine_destroy(sc_engine *e) { delete e; }

extern "C" int sc_engine_config(const sc_engine *e, sc_config *out) {
  SC_CHECK_ARG(e && out, "null");
  *out = e->cfg;
  return SC_OK;
}

extern "C" int sc_streams_create(sc_engine *e, const sc_stream_options *o, sc_streams **out) {
  SC_CHECK_ARG(e && o && out, "null");
  const sc_config &c = e->cfg;
  SC_CHECK_ARG(o->n_streams > 0 && o->beam_size > 0 && o->max_frames > 0 && o->max_tokens > 1, "bad options");
  const int pre_beam = 40;
  const int K = std::min(pre_beam, c.vocab_size);
  SC_CHECK_ARG(o->beam_size <= K && o->beam_size <= 16, "beam size must not exceed 16 (and the pre-beam size 40)");
  const int dk = c.d_model / c.dec_heads;
  SC_CHECK_ARG(c.d_model % c.dec_heads == 0 && (dk == 16 || dk == 32 || (dk == 64 && o->beam_size <= 10)),
               "decoder head dim must be 16, 32 or 64 (64: beam <= 10)");
  HIP_TRY(hipSetDevice(e->device));
  sc_streams *b = new sc_streams;
  b->eng = e;
  b->cfg = c;
  const int S = b->S = o->n_streams, W = b->W = o->beam_size;
  b->K = K;
  b->TCAP = o->max_frames;
  b->LCAP = o->max_tokens;
  b->max_chunk = o->max_chunk_samples > 0 ? o->max_chunk_samples : 32768;
  b->PCAP = o->pcm_capacity > 0 ? o->pcm_capacity : (1 << 20);
  b->use_bbd = o->use_bbd != 0;
  b->strict = o->strict_reference != 0;
  const int d = c.d_model, V = c.vocab_size, F = c.ffn_dim;
  b->max_feat_new = 2 + b->max_chunk / c.hop_length + 8;
  b->FCAP = b->max_feat_new + 16;
  b->UCAP = c.block_size + b->FCAP / 4 + 8;
  b->max_t1 = (b->FCAP - 3) / 2 + 1;
  b->max_t2 = (b->max_t1 - 3) / 2 + 1;
  b->max_blocks = S * (b->UCAP / c.hop_size + 1);
  const int R = c.block_size + 2;
  const size_t m_enc = (size_t)b->max_blocks * R, n = (size_t)S * W;
  int rc = SC_OK;
#define A(ptr, count) if (rc == SC_OK) rc = b->alloc(&ptr, (size_t)(count))
  if (hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, -1) != hipSuccess) {
    sc_set_error("sc_streams_create: hipStreamCreate failed");
    delete b;
    return SC_ERR_LAUNCH;
  }
  A(b->pcm, (size_t)S * b->PCAP);
  A(b->featbuf, (size_t)2 * S * b->FCAP * c.n_mels);
  A(b->subbuf, (size_t)2 * S * b->UCAP * d);
  A(b->prev_addin, (size_t)S * d);
  A(b->past_ctx, (size_t)S * c.enc_layers * d);
  A(b->enc, (size_t)S * b->TCAP * d);
  A(b->c1, (size_t)S * b->max_t1 * c.conv_freq1 * d);
  A(b->c2, (size_t)S * b->max_t2 * c.conv_freq2 * d);
  A(b->xblk, m_enc * d);
  A(b->ws_xn, m_enc * d);
  A(b->ws_qkv, m_enc * 3 * d);
  A(b->ws_att, m_enc * d);
  A(b->ws_ffh, m_enc * F);
  A(b->jobs_ctx, (size_t)S * 4);
  A(b->ctrlmap, (size_t)S * 8 + n);
  A(b->arena_dev, b->arena_cap);
  sc_search &sb = b->sb;
  sb.S = S; sb.W = W; sb.K = K; sb.V = V; sb.d = d; sb.H = c.dec_heads; sb.F = F; sb.n_layers = c.dec_layers;
  sb.TCAP = b->TCAP; sb.LCAP = b->LCAP; sb.xchunk = 256; sb.blank = c.blank_id; sb.eos = c.eos_id; sb.sos = c.sos_id;
  sb.w_dec = 1.0f - o->ctc_weight; sb.w_ctc = o->ctc_weight; sb.ln_eps = c.ln_eps;
  float *ctcx = nullptr, *ckv = nullptr;
  A(ctcx, (size_t)S * b->TCAP * V);
  sb.kv_half = o->kv_half != 0;
  const size_t kvdiv = sb.kv_half ? 2 : 1;   // fp16 elements: half the bytes
  A(ckv, (size_t)S * c.dec_layers * b->TCAP * 2 * d / kvdiv + 4);
  sb.ctcx = ctcx; sb.ckv = ckv;
  if (sb.kv_half) {
    b->kv_stage_rows = std::max(256, S * 24);
    A(b->kv_stage, (size_t)c.dec_layers * b->kv_stage_rows * 2 * d);
  }
  sb.tct = (b->TCAP + 3) / 4 * 4;
  A(sb.ctcxT, (size_t)S * V * sb.tct);
  A(sb.skv, (size_t)S * c.dec_layers * b->LCAP * W * 2 * d / kvdiv + 4);
  A(sb.yseq, (size_t)2 * n * b->LCAP);
  A(sb.xpos, (size_t)2 * n * b->LCAP);
  A(sb.anc, (size_t)2 * S * b->LCAP * W);
  A(sb.score, 2 * n); A(sb.sc_dec, 2 * n); A(sb.sc_ctc, 2 * n);
  A(sb.ctc_r, (size_t)2 * S * b->TCAP * 2 * W);
  A(sb.ctc_s, 2 * n);
  A(sb.ctc_rnew, (size_t)S * b->TCAP * 2 * W * K);
  A(b->flags_dev, (size_t)S);   // (sb.flags: set per step - flags_dev, or the host-mapped array itself)
  A(sb.dx, n * d); A(sb.dxn, n * d); A(sb.dqkv, n * 3 * d); A(sb.datt, n * d); A(sb.dq, n * d); A(sb.dffh, n * F);
  A(sb.logits, n * V); A(sb.logp, n * V);
  A(sb.pre_ids, n * K); A(sb.psi, n * K); A(sb.psi_eos, n);
  A(sb.cand_score, n * W); A(sb.cand_tok, n * W); A(sb.cand_ctc, n * W);
  A(sb.sel, n * 2);
  const int nch = (b->TCAP + 255) / 256;
  A(sb.xpart, n * c.dec_heads * nch * (d / c.dec_heads + 2));
  if (sc_dec_layer_fused_supported(d, c.dec_heads, W, F) && V % d == 0 && e->dec[0].wqkv_pp && e->f("out_w_q", false)) {
    A(sb.ph1, n * c.dec_heads * d);
    A(sb.ph2, n * c.dec_heads * d);
    A(sb.ffn_part, (size_t)(F / 128) * n * d);
    sb.max_ffn_part = F / 128;
    if (sc_dec_cluster_supported(d, c.dec_heads, W, F) && rc == SC_OK) {   // persistent stream-cluster decoder
      sc_dec_layer *ld = nullptr;
      A(ld, e->dec.size());
      if (rc == SC_OK && hipMemcpy(ld, e->dec.data(), e->dec.size() * sizeof(sc_dec_layer), hipMemcpyHostToDevice) != hipSuccess)
        rc = SC_ERR_LAUNCH;
      sb.layers_dev = ld;
      A(sb.cbar, (size_t)S);
      A(sb.cl_err, 1);
    }
  }
  if (rc == SC_OK && hipMalloc(&b->ws, (size_t)128 << 20) != hipSuccess) rc = SC_ERR_LAUNCH;
  if (rc == SC_OK) b->owned.push_back(b->ws);
#undef A
  if (rc != SC_OK) {
    sc_set_error("sc_streams_create: device allocation failed");
    delete b;
    return rc;
  }
  sb.ctrl = b->ctrlmap;
  sb.rowmap = b->ctrlmap + S * 8;
  sb.n_rows = S * W;
  sb.embed = e->f("embed"); sb.pe = e->f("pe"); sb.dec_norm_g = e->f("dec_norm_g"); sb.dec_norm_b = e->f("dec_norm_b");
  sb.out_w = e->f("out_w"); sb.out_b = e->f("out_b"); sb.out_w_q = e->f("out_w_q", false);
  sb.layers = e->dec.data();
  (void)sc_set_stream_workspace(b->stream, b->ws, (size_t)128 << 20);
  b->es = b->stream;
  b->pend = new PendingEnc;
  {
    // The encoder stage starts when at most 7 % of the streams are still in the step loop (small batches: at once).
    // Measured at 128 streams (profiles/r02_encoder_overlap_sweep.txt): serial 33.4 ms per chunk step; started with
    // the first decode iteration 32.2 (its large grids delay the full-batch decode kernels); at 50 % / 25 % / 10 % /
    // 7 % / 3 % of the streams 31.3 / 31.1 / 30.5 / 30.5 / 30.8.  SC_ENC_START overrides the percentage.
    const char *th = sc_hook("SC_ENC_START");
    b->enc_start_thr = S < 16 ? S : std::max(1, (int)((long)S * (th ? atoi(th) : 7) / 100));
  }
  {
    // second stream for the encoder side (see sc_streams::stream_enc).  SC_ENC_OVERLAP=0 keeps everything on one
    // stream; SC_ENC_CUS=n restricts the encoder stream to the first n compute units (hipExtStreamCreateWithCUMask)
    const char *ov = sc_hook("SC_ENC_OVERLAP");
    if (!(ov && atoi(ov) == 0)) {
      const char *cu = sc_hook("SC_ENC_CUS");
      const int ncu = cu ? atoi(cu) : 0;
      hipError_t er;
      if (ncu > 0 && ncu < 256) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ncu; ++i) mask[i >> 5] |= 1u << (i & 31);
        er = hipExtStreamCreateWithCUMask(&b->stream_enc, 8, mask);
      } else {
        er = hipStreamCreateWithPriority(&b->stream_enc, hipStreamNonBlocking, 0);
      }
      if (er == hipSuccess && hipMalloc(&b->ws_enc, (size_t)128 << 20) == hipSuccess &&
          hipEventCreateWithFlags(&b->ev_enc_done, hipEventDisableTiming) == hipSuccess) {
        b->owned.push_back(b->ws_enc);
        (void)sc_set_stream_workspace(b->stream_enc, b->ws_enc, (size_t)128 << 20);
      } else {
        (void)hipGetLastError();
        if (b->stream_enc) (void)hipStreamDestroy(b->stream_enc);
        b->stream_enc = nullptr;
      }
    }
  }
  const size_t cm = ((size_t)S * 8 + n) * sizeof(int32_t);
  if (hipHostMalloc((void **)&b->ctrlmap_host, cm) != hipSuccess || hipHostMalloc((void **)&b->ctrl0_host, (size_t)S * 32) != hipSuccess ||
      hipHostMalloc((void **)&b->flags_host, (size_t)S * 8) != hipSuccess ||
      hipHostMalloc((void **)&b->rm_host[0], n * sizeof(int32_t)) != hipSuccess ||
      hipHostMalloc((void **)&b->rm_host[1], n * sizeof(int32_t)) != hipSuccess ||
      hipEventCreateWithFlags(&b->ev_iter[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&b->ev_iter[1], hipEventDisableTiming) != hipSuccess ||
      hipHostMalloc((void **)&b->arena_host, b->arena_cap * 4) != hipSuccess) {
    sc_set_error("sc_streams_create: pinned host allocation failed");
    delete b;
    return SC_ERR_LAUNCH;
  }
  memset(b->flags_host, 0, (size_t)S * 8);
  {
    void *dv = nullptr;
    if (hipHostGetDevicePointer(&dv, b->flags_host, 0) != hipSuccess || !dv) {
      sc_set_error("sc_streams_create: pinned flags are not device-accessible");
      delete b;
      return SC_ERR_LAUNCH;
    }
    b->ring_dev = (int32_t *)dv;
    sb.flags = b->ring_dev;
  }
  if (const char *sp = sc_hook("SC_SPECULATE")) b->speculate = atoi(sp) != 0;   // tests / A-B runs
  if (const char *sp = sc_hook("SC_SCAN_SPLIT_MIN")) b->scan_split_min = atoi(sp);   // tests: 0 = never, small = always
  memset(b->ctrlmap_host, 0, cm);
  for (size_t i = 0; i < n; ++i) b->rm_host[0][i] = b->rm_host[1][i] = (int32_t)i;
  (void)hipMemcpy(b->ctrlmap, b->ctrlmap_host, (size_t)S * 8 * sizeof(int32_t), hipMemcpyHostToDevice);
  (void)hipMemcpy(b->ctrlmap + S * 8, b->rm_host[0], n * sizeof(int32_t), hipMemcpyHostToDevice);
  b->rowmap_key.resize(S);
  for (int s = 0; s < S; ++s) b->rowmap_key[s] = s;
  b->row_bucket = std::max(1, S / 32);   // 32 compaction buckets (graphs): 16 -> 32 measured +1 % at 128 streams, 64 nothing more
  if (const char *rb = sc_hook("SC_ROW_BUCKETS")) b->row_bucket = std::max(1, S / std::max(1, atoi(rb)));   // tools: sweep
  b->n_rows_step = S * W;
  b->st.assign(S, St());
  for (int s = 0; s < S; ++s) init_hyp(b, s);
  *out = b;
  return SC_OK;
}

extern "C" void sc_streams_destroy(sc_streams *b) {
  if (b) {
    (void)hipStreamSynchronize(b->stream);
    delete b;
  }
}

extern "C" int sc_reset(sc_streams *b, int stream) {
  SC_CHECK_ARG(b && stream >= 0 && stream < b->S, "stream out of range");
  reset_stream(b, stream);
  return SC_OK;
}

extern "C" void *sc_streams_hip_stream(sc_streams *b) { return b ? (void *)b->stream : nullptr; }

extern "C" float *sc_streams_pcm(sc_streams *b, long *capacity) {
  if (!b) return nullptr;
  if (capacity) *capacity = b->PCAP;
  return b->pcm;
}

// One chunk step (Speech2TextStreaming.__call__ for raw audio, for the listed streams at once).
// pcm[i] == NULL: the samples are already resident in the stream's device PCM buffer (sc_streams_pcm) behind
// what it has received so far - only the count is taken.  status[i] (HOST out): 1 = the call produced output,
// 0 = the reference's early `return []` (speech2text_streaming.py:432-433), < 0 = this stream failed
// (SC_ERR_CAPACITY: a capacity limit; SC_ERR_INPUT: an input the reference itself raises on, A3) - it has been
// reset, sc_last_error() holds the message, and every other stream of the call is decoded as if it had not been
// there.  Returns SC_OK unless the call as a whole failed.
extern "C" int sc_push(sc_streams *b, const int *stream_ids, const float *const *pcm, const int *n_samples,
                       const uint8_t *is_final, int n, int *status) {
  SC_CHECK_ARG(b && stream_ids && n_samples && is_final && n >= 0, "null");
  HIP_TRY(hipSetDevice(b->eng->device));
  std::vector<Chunk> chunks;
  std::vector<int> pos;   // index in the caller's arrays
  for (int i = 0; i < n; ++i) {
    SC_CHECK_ARG(stream_ids[i] >= 0 && stream_ids[i] < b->S && n_samples[i] >= 0, "stream id / sample count out of range");
    chunks.push_back({stream_ids[i], pcm ? pcm[i] : nullptr, n_samples[i], is_final[i] != 0});
    pos.push_back(i);
    if (status) status[i] = 0;
  }
  b->arena_off = 0;
  std::vector<StreamFault> faults;
  std::map<int, int> feat_new;
  std::map<int, bool> finals;
  std::vector<int> has_out, t_old;
  while (true) {
    for (auto &ch : chunks) {   // compaction moves device data: settle it before the snapshot
      St &st = b->st[ch.s];
      if (st.pcm_end + ch.n > b->PCAP) RC_TRY(compact_pcm(b, ch.s));
    }
    std::vector<St> snap = b->st;
    feat_new.clear();
    finals.clear();
    has_out.assign(chunks.size(), 0);
    try {
      b->es = b->stream_enc ? b->stream_enc : b->stream;   // (the PCM chunks are copied on the encoder's stream)
      const int rc_enc = stage_encode(b, chunks, feat_new, finals, has_out);
      b->es = b->stream;
      if (rc_enc != SC_OK) return rc_enc;
      if (!b->stream_enc) RC_TRY(launch_encoder(b));        // one stream: encoder first, as the reference does
      t_old.assign(b->S, 0);
      for (int s = 0; s < b->S; ++s) t_old[s] = snap[s].T_enc;
      break;
    } catch (const StreamFault &f) {
      b->es = b->stream;
      b->st = snap;   // planning is pure host work that precedes every launch: undo = restore the mirrors
      b->arena_off = 0;
      faults.push_back(f);
      for (size_t k = 0; k < chunks.size(); ++k)
        if (chunks[k].s == f.stream) {
          chunks.erase(chunks.begin() + k);
          pos.erase(pos.begin() + k);
          break;
        }
      if (chunks.empty()) break;
    }
  }
  if (!chunks.empty()) RC_TRY(stage_decode(b, feat_new, finals, faults, &t_old));
  RC_TRY(launch_encoder(b));   // (if no decode iteration got to it)
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (b->stream_enc) HIP_TRY(hipStreamSynchronize(b->stream_enc));
  b->enc_pending = false;
  if (status)
    for (size_t k = 0; k < chunks.size(); ++k) status[pos[k]] = has_out[k];
  for (auto &f : faults) {
    reset_stream(b, f.stream);
    sc_set_error("stream %d: %s", f.stream, f.msg.c_str());
    if (status)
      for (int i = 0; i < n; ++i)
        if (stream_ids[i] == f.stream) status[i] = f.code;
  }
  return SC_OK;
}

// live hypotheses of a stream, best first: ids / xpos [nbest][max_len] (row-major, caller-allocated), lens[nbest],
// scores / score_dec / score_ctc [nbest] (any of them may be NULL).  Returns the number of hypotheses written
// (<= nbest), or a negative error.
extern "C" int sc_get_hyps(sc_streams *b, int stream, int nbest, int max_len, int32_t *ids, int32_t *xpos, int *lens,
                           double *scores, double *score_dec, double *score_ctc) {
  SC_CHECK_ARG(b && stream >= 0 && stream < b->S && nbest >= 0 && max_len >= 0, "bad arguments");
  const St &st = b->st[stream];
  if (!st.started) return 0;
  const int n = std::min(nbest, st.nhyp), L = st.L;
  SC_CHECK_ARG(ids == nullptr || max_len >= L, "max_len is smaller than the hypotheses");
  const sc_search &sb = b->sb;
  const size_t o = ((size_t)st.cur * b->S + stream) * b->W;
  for (int i = 0; i < n; ++i) {
    if (ids) HIP_TRY(hipMemcpy(ids + (size_t)i * max_len, sb.yseq + (o + i) * b->LCAP, (size_t)L * 4, hipMemcpyDeviceToHost));
    if (xpos) HIP_TRY(hipMemcpy(xpos + (size_t)i * max_len, sb.xpos + (o + i) * b->LCAP, (size_t)L * 4, hipMemcpyDeviceToHost));
    if (lens) lens[i] = L;
  }
  if (scores && n) HIP_TRY(hipMemcpy(scores, sb.score + o, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (score_dec && n) HIP_TRY(hipMemcpy(score_dec, sb.sc_dec + o, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (score_ctc && n) HIP_TRY(hipMemcpy(score_ctc, sb.sc_ctc + o, (size_t)n * 8, hipMemcpyDeviceToHost));
  return n;
}

// 2-D (T, n_mels) already-normalised features instead of PCM (the reference's 2-D / 3-D input path,
// speech2text_streaming.py:438-449): feats[i] HOST [n_frames[i]][n_mels].  Same status convention as sc_push.
extern "C" int sc_push_features(sc_streams *b, const int *stream_ids, const float *const *feats, const int *n_frames,
                                const uint8_t *is_final, int n, int *status) {
  SC_CHECK_ARG(b && stream_ids && feats && n_frames && is_final && n >= 0, "null");
  HIP_TRY(hipSetDevice(b->eng->device));
  const sc_config &c = b->cfg;
  struct Item { int s; const float *f; int n; bool fin; int pos; };
  std::vector<Item> items;
  for (int i = 0; i < n; ++i) {
    SC_CHECK_ARG(stream_ids[i] >= 0 && stream_ids[i] < b->S && n_frames[i] >= 0 && feats[i], "bad item");
    items.push_back({stream_ids[i], feats[i], n_frames[i], is_final[i] != 0, i});
    if (status) status[i] = 0;
  }
  b->arena_off = 0;
  std::vector<StreamFault> faults;
  std::map<int, int> feat_new;
  std::map<int, bool> finals;
  while (!items.empty()) {
    std::vector<St> snap = b->st;
    feat_new.clear();
    finals.clear();
    try {
      for (auto &it : items) {   // capacity of every item BEFORE anything is copied
        const St &st = b->st[it.s];
        const int nbuf = st.enc_started ? st.nfeat : 0;
        if (it.n > b->max_feat_new || nbuf + it.n > b->FCAP)
          throw StreamFault{it.s, SC_ERR_CAPACITY, "feature frames of one call exceed the batch's capacity (max_chunk_samples)"};
      }
      EncPlan P;
      std::vector<int> enc_streams;
      std::vector<std::pair<const Item *, size_t>> copies;
      for (auto &it : items) {
        St &st = b->st[it.s];
        const int nbuf = st.enc_started ? st.nfeat : 0;
        copies.push_back({&it, ((size_t)(st.fpp * b->S + it.s) * b->FCAP + nbuf) * c.n_mels});
        feat_new[it.s] = it.n;
        finals[it.s] = it.fin;
        st.started = true;
        if (it.n >= 3) enc_streams.push_back(it.s);
      }
      if (!enc_streams.empty()) encode_plan(b, enc_streams, feat_new, finals, P);
      for (auto &cp : copies)   // appended rows only
        if (cp.first->n > 0)
          HIP_TRY(hipMemcpyAsync(b->featbuf + cp.second, cp.first->f, (size_t)cp.first->n * c.n_mels * sizeof(float),
                                 hipMemcpyHostToDevice, b->stream));
      if (P.n_conv) RC_TRY(encode_launch(b, P));
      break;
    } catch (const StreamFault &f) {
      b->st = snap;
      b->arena_off = 0;
      faults.push_back(f);
      for (size_t k = 0; k < items.size(); ++k)
        if (items[k].s == f.stream) { items.erase(items.begin() + k); break; }
    }
  }
  if (!items.empty()) RC_TRY(stage_decode(b, feat_new, finals, faults));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (status)
    for (auto &it : items) status[it.pos] = 1;
  for (auto &f : faults) {
    reset_stream(b, f.stream);
    sc_set_error("stream %d: %s", f.stream, f.msg.c_str());
    if (status)
      for (int i = 0; i < n; ++i)
        if (stream_ids[i] == f.stream) status[i] = f.code;
  }
  return SC_OK;
}

// host <-> device copies of a stream's PCM ring and encoder output (tests, bench preload, the drop-in class's
// frontend_states / encoder_buffer views)
extern "C" int sc_streams_write_pcm(sc_streams *b, int stream, long offset, const float *host, long n) {
  SC_CHECK_ARG(b && host && stream >= 0 && stream < b->S && offset >= 0 && n >= 0 && offset + n <= b->PCAP, "out of range");
  HIP_TRY(hipMemcpy(b->pcm + (long)stream * b->PCAP + offset, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  return SC_OK;
}
extern "C" long sc_streams_read_pcm_buffer(sc_streams *b, int stream, float *host, long max_n) {
  if (!b || stream < 0 || stream >= b->S) return SC_ERR_ARG;
  const St &st = b->st[stream];
  const long n = std::min<long>(st.pcm_end - st.pcm_start, max_n);
  if (host && n > 0 &&
      hipMemcpy(host, b->pcm + (long)stream * b->PCAP + st.pcm_start, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return SC_ERR_LAUNCH;
  return n;
}
extern "C" int sc_streams_read_enc(sc_streams *b, int stream, float *host, int max_frames) {
  if (!b || stream < 0 || stream >= b->S) return SC_ERR_ARG;
  const int T = std::min(b->st[stream].T_enc, max_frames);
  if (host && T > 0 &&
      hipMemcpy(host, b->enc + (size_t)stream * b->TCAP * b->cfg.d_model, (size_t)T * b->cfg.d_model * sizeof(float),
                hipMemcpyDeviceToHost) != hipSuccess)
    return SC_ERR_LAUNCH;
  return T;
}

extern "C" int sc_stream_info(const sc_streams *b, int stream, sc_stream_info_t *out) {
  SC_CHECK_ARG(b && out && stream >= 0 && stream < b->S, "bad arguments");
  const St &st = b->st[stream];
  out->enc_frames = st.T_enc;
  out->processed_block = st.processed_block;
  out->process_idx = st.process_idx;
  out->n_hyp = st.started ? st.nhyp : 0;
  out->hyp_len = st.L;
  out->decode_steps = st.n_steps_total;
  out->pcm_buffered = (int32_t)(st.pcm_end - st.pcm_start);
  out->frontend_started = st.fe_started;
  return SC_OK;
}

// measurement aids (bench.py roofline leg): hipGraph replay off so that every launch can be bracketed by HIP
// events (sc_prof_enable); encoder K|V rows read by the cross-attention since the last call (returns and clears)
extern "C" int sc_streams_set_graphs(sc_streams *b, int on) {
  SC_CHECK_ARG(b, "null");
  b->use_graphs = on != 0;
  return SC_OK;
}
extern "C" long sc_streams_take_xattn_rows(sc_streams *b) {
  if (!b) return 0;
  const long v = b->xattn_rows[0] + b->xattn_rows[1];
  b->xattn_rows[0] = b->xattn_rows[1] = 0;
  return v;
}
extern "C" int sc_streams_take_xattn_rows_by_kernel(sc_streams *b, long *rows) {
  SC_CHECK_ARG(b && rows, "null");
  rows[0] = b->xattn_rows[0];
  rows[1] = b->xattn_rows[1];
  b->xattn_rows[0] = b->xattn_rows[1] = 0;
  return SC_OK;
}

// seconds and iterations of the decode step loop by compaction bucket (index = active streams / (S/16), 17 entries),
// since the last call (returned and cleared)
extern "C" int sc_streams_bucket_times(sc_streams *b, double *seconds, long *iterations) {
  SC_CHECK_ARG(b && seconds && iterations, "null");
  for (int i = 0; i < 17; ++i) {
    seconds[i] = b->t_bucket[i];
    iterations[i] = b->n_bucket[i];
    b->t_bucket[i] = 0;
    b->n_bucket[i] = 0;
  }
  return SC_OK;
}

extern "C" int sc_streams_host_times(sc_streams *b, double *launch_s, double *wait_s) {
  SC_CHECK_ARG(b && launch_s && wait_s, "null");
  *launch_s = b->t_launch;
  *wait_s = b->t_wait;
  b->t_launch = b->t_wait = 0;
  return SC_OK;
}

// decode iterations enqueued ahead of their predecessor's stop flags, and how many of those found every stream stopped
extern "C" int sc_streams_speculation(const sc_streams *b, long *launched, long *wasted) {
  SC_CHECK_ARG(b, "null");
  if (launched) *launched = b->spec_launched;
  if (wasted) *wasted = b->spec_wasted;
  return SC_OK;
}
extern "C" int sc_streams_set_speculation(sc_streams *b, int on) {
  SC_CHECK_ARG(b, "null");
  b->speculate = on != 0;   // the decode graphs of the other mode stay cached (graph_key)
  return SC_OK;
}

extern "C" int sc_streams_stats(const sc_streams *b, long *enc_calls, long *dec_steps, long *dec_blocks) {
  SC_CHECK_ARG(b, "null");
  if (enc_calls) *enc_calls = b->enc_calls;
  if (dec_steps) *dec_steps = b->dec_steps;
  if (dec_blocks) *dec_blocks = b->dec_blocks;
  return SC_OK;
}
