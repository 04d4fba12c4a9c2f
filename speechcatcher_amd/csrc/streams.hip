// Stream-level C ABI of libscasr (include/scasr.h: sc_engine_* / sc_streams_* / sc_push / sc_submit / sc_poll /
// sc_get_hyps(_batch) / sc_reset): the host state machine that turns the kernel-level entry points into a
// streaming decoder, in C++ - no Python in the loop.  It is the host half of the reference's
//   Speech2TextStreaming.__call__ / apply_frontend      speechcatcher/speech2text_streaming.py:278-539
//   ContextualBlockTransformerEncoder.forward_infer     speechcatcher/model/encoder/contextual_block_transformer_encoder.py:241-419
//   BlockwiseSynchronousBeamSearch.process_block / _decode_one_block / reset
//                                                       speechcatcher/beam_search/beam_search.py:343-356,507-838
// for S streams at once: pure integer bookkeeping (the three nested carry-over buffers of SURVEY.md
// Appendix D, the block schedule, the step loop with its stop flags, rollback and rewind) that decides
// WHAT to launch; all arithmetic runs in the HIP kernels.  speechcatcher_amd/engine.py is the same logic in
// Python (it runs on the CPU spec backend, which is how the logic is checked against the reference
// fixtures without a GPU); tests/test_gpu_native.py holds both to the same fixtures.
//
// One engine, two ways to drive it.  The decode side is a TICK engine over per-stream state: every stream owns a
// queue of ready decode blocks and the state of the block it is in; one tick starts the blocks that are ready,
// runs ONE beam-search step for every stream that is inside a block, reads the stop flags and does the
// accept / rollback / rewind bookkeeping per stream.  The encoder side is a sequence of admission GROUPS
// (frontend + encoder + CTC / cross-attention K|V projections of the chunks admitted together), each on the
// encoder HIP stream with an event; a block that sees frames of a group waits for that group only.
//   sc_push            = admit the listed chunks as one group, tick until every one of them is complete (the
//                        reference's per-call semantics: strict lock-step of the batch);
//   sc_submit/sc_poll  = continuous batching: chunks are admitted whenever the host has them, a stream's reply is
//                        ready when ITS blocks are done, streams that finished early start their next chunk while the
//                        stragglers of the previous one are still decoding (the reference's concurrency unit is an
//                        independent stream: speechcatcher_server.py:331-371).
// Per stream both give the same blocks, the same steps, the same results.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <new>
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
  float *wkv_all = nullptr, *bkv_all = nullptr;   // the layers' K|V weights [Ld * 2d][d] and biases one behind the other (project_rows)

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
    if (wkv_all) (void)hipFree(wkv_all);
    if (bkv_all) (void)hipFree(bkv_all);
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
  // encoder frames whose CTC rows / cross-attention K|V rows have been projected (by the encoder stage that
  // emitted them); after a strict reset() T_proj restarts at the stale table's extent (scorers.py:342-350)
  int T_proj = 0, T_projkv = 0;
  long n_steps_total = 0;
};

// a decode block of the schedule (beam_search.py:590-634): the T frames [0, T) it presents to the scorers, whether
// it is the final block, and the encoder group whose frames it sees (0: only frames that were there before)
// ... and the chunk (Job::seq of its stream) that queued it
struct Blk { int T; bool fin; long gen; long job; };

// the block a stream is inside (the loop state of _decode_one_block, beam_search.py:655-838)
struct Run {
  bool inblk = false, live = false, took = false, pvalid = false, has = false, hasp = false, fin = false;
  int cur = 0, L = 1, nhyp = 1, nhp = 1, pidx = 0, T = 0, Tc = 0, out = 0;
  long nsteps = 0;
  long job = 0;   // Job::seq of the chunk the block belongs to
};

// a stream's outstanding chunk (sc_push / sc_submit): open until it has been reported.  With a queue depth > 1
// (sc_streams_set_queue_depth) further chunks of the stream wait behind it (sc_streams::ahead), in order.
struct Job {
  bool open = false;
  int has_out = 0;   // 1: the call produced output, 0: the reference's early `return []`
  int fault = 0;     // SC_ERR_*: the chunk failed, the stream is reset when it is reported
  long seq = 0;      // per-stream sequence number of the chunk (tags its decode blocks)
  bool fin = false;  // is_final chunk: nothing may be queued behind it
  bool dropped = false;   // failed only because an EARLIER chunk of the stream failed (the reset has happened by then)
  bool started = false;   // St::started right after THIS chunk's admission (a later admission may set it before this one is reported)
};

// hypotheses of a stream's last complete chunk, copied aside when later chunks of the stream may go on decoding
// (queue depth > 1): what sc_get_hyps_batch returns for the stream until the next sc_poll call
struct Snap {
  bool valid = false, reported = false;
  long seq = 0;
  int L = 1, nhyp = 0;
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

// One admission group: the frontend + encoder launches of the chunks admitted together, planned (host state already
// advanced) and issued on the encoder stream either at once (sc_submit) or when the decode loop has thinned out
// (sc_push).  `ev` completes when the group's encoder output, CTC rows and cross-attention K|V rows are in place.
struct EncGroup {
  long gen = 0;
  bool launched = false, features = false;
  bool open = false;                         // sc_submit: later admissions may still be merged into this (unlaunched) group
  bool deferred = false;                     // sc_submit: issued when it is full or when a decode block needs its frames
  int slot = 0;                              // job-table arena slot
  int stage_slot = -1;                       // pinned staging slot holding the admission's host input (-1: none)
  std::vector<long long> copy_jobs;          // [n][3] staging offset, destination offset, count (floats)
  size_t stage_floats = 0;
  std::vector<int32_t> fe_jobs;
  int n_fe = 0, max_keep = 0;
  EncPlan P;
  std::vector<int32_t> ctc_rows, kv_src, kv_dst;   // eager projections of the frames the group emits
  bool same_rows = true;
  std::vector<int> streams;                  // streams with work in this group (each at most once)
  bool empty() const { return !n_fe && !P.n_conv; }
};

constexpr int N_ARENA = 8;   // job-table arena slots = encoder groups that can be in flight
constexpr int N_STAGE = 4;   // pinned staging slots for host input (a slot is free again when its H2D copy is done)
constexpr int N_RING = 4;    // block-start tables (ctrl rows, log-softmax row lists): pinned ring

struct sc_streams {
  sc_engine *eng = nullptr;
  sc_config cfg{};
  int S = 0, W = 0, K = 0, TCAP = 0, LCAP = 0, FCAP = 0, UCAP = 0, max_feat_new = 0, max_t1 = 0, max_t2 = 0,
      max_blocks = 0, max_chunk = 0, max_length = 500;
  long PCAP = 0;
  bool use_bbd = false, strict = true;
  hipStream_t stream = nullptr;
  // Encoder side on its OWN HIP stream: frontend + encoder of a chunk do not feed the decode blocks whose frames
  // were already there before the chunk (the block schedule runs one hop behind the encoder), so both
  // are in flight together: the encoder's kernels run in the gaps of the decode chain (between a step's last kernel
  // and the next step's first - the host reads the stop flags there) and behind its thin iterations.  Kernel traces
  // (tools/rocpd_busy.py) show the kernels of the two streams ALTERNATING, not sharing CUs: 1.5 % of the wall time
  // has both running - every kernel of the path fills the registers or the LDS of the CUs it occupies.
  // `es` is the stream the current phase launches into.
  hipStream_t stream_enc = nullptr, es = nullptr;
  hipEvent_t ev_iter[2] = {nullptr, nullptr};   // end of decode iteration k (k & 1)
  int32_t *ring_dev = nullptr;                  // device view of flags_host [S]: the prune kernel stores the stop flags there
  int32_t *rm_host[2] = {nullptr, nullptr};     // pinned rowmap images (double-buffered: one may still be in a copy queue)
  int rm_idx = 0;
  bool rm_dirty = false;                        // rm_host[rm_idx] differs from the device rowmap
  // T-parallel CTC scan: frames to walk >= scan_split_min (tools: and bucket streams <= scan_split_streams, 0 = any).  1024
  // (round 5; 256 with a 48-stream bucket limit until round 4): WHICH form a stream's scan takes must depend on that
  // stream alone (bit-reproducible serving), and with every stream of a full bucket in it the T-parallel form costs more
  // than the sequential walk (3181 against 3258 audio-s/s at 390 frames to walk, profiles/r05_ab_scan_logits.txt) - it is
  // for the long tables of CLI segments (T = 4500: 61 against 588 us per step)
  int scan_split_min = 1024, scan_split_streams = 0;
  bool scan_long = false;                       // this step: a live stream has >= scan_split_min frames to walk
  // (round 5) the T-parallel kernel is part of a step whenever ONE live stream has a table long enough for it - whatever the
  // size of the bucket: which form a stream's scan takes must depend on that stream alone (bit-reproducible serving);
  // scan_split_streams (tools) restores the old bucket limit for A/B runs
  int step_split_min() const { return (scan_long && (scan_split_streams <= 0 || n_rows_step <= scan_split_streams * W)) ? scan_split_min : 0; }
  int graph_key() const { return n_rows_step * 2 + (step_split_min() > 0 ? 1 : 0); }
  int enc_start_thr = 0;         // sc_push: launch the planned encoder group when at most this many streams are still decoding
  int enc_batch_min = 0;         // sc_submit: launch the open encoder group when it holds this many streams (sc_streams_set_encoder_batch)
  int gemm_flags = 0;            // SC_GEMM_SPLIT16 when the engine carries split-precision copies of the PROJECTIONS (wqkv_s): the tiled
                                 // GEMMs of the encoder stage (conv2, subsampling Linear, CTC / cross-K|V projections) in the same form
                                 // (weights.py checks the range of their operands at load time)
  void *ws = nullptr, *ws_enc = nullptr;
  std::vector<void *> owned, owned_host;
  // device buffers
  float *pcm = nullptr, *featbuf = nullptr, *subbuf = nullptr, *prev_addin = nullptr, *past_ctx = nullptr, *enc = nullptr,
        *c1 = nullptr, *c2 = nullptr, *xblk = nullptr, *ws_xn = nullptr, *ws_qkv = nullptr, *ws_att = nullptr,
        *ws_ffh = nullptr;
  int32_t *jobs_ctx = nullptr, *ctrlmap = nullptr;
  float *kv_stage = nullptr;   // kv_half: fp32 staging [dec_layers][kv_stage_rows][2d] of the K|V projections
  int kv_stage_rows = 0;
  sc_search sb{};
  // pinned host
  int32_t *ctrlmap_host = nullptr, *flags_host = nullptr;
  // ---- job tables of the encoder groups: N_ARENA slots of one pinned / one device arena --------------------------
  int32_t *arena_host = nullptr, *arena_dev = nullptr;
  size_t slot_cap = 0, arena_off = 0;
  int cur_slot = 0;
  hipEvent_t ev_group[N_ARENA] = {nullptr};
  long slot_gen[N_ARENA] = {0};   // group that owns the slot (0: free)
  // ---- block-start tables (decode side) ------------------------------------------------------------------------------
  int32_t *bs_host = nullptr, *bs_dev = nullptr;   // [N_RING][S*8 + S*block_size]: ctrl rows, then log-softmax rows
  size_t bs_cap = 0;
  int bs_i = 0;
  // ---- host input staging (sc_push / sc_submit with host pointers): one H2D copy + one scatter launch per group ----
  float *stage_host = nullptr, *stage_dev = nullptr;
  size_t stage_cap = 0;           // floats per slot
  hipEvent_t ev_stage[N_STAGE] = {nullptr};
  bool stage_busy[N_STAGE] = {false};
  int stage_next = 0;
  long long *cjobs_host = nullptr, *cjobs_dev = nullptr;   // [N_STAGE][S*3] scatter jobs of a staging slot
  // ---- batched hypothesis read-back (sc_get_hyps_batch): pack kernel -> one D2H into pinned memory -------------------
  int32_t *pack_dev = nullptr, *pack_host = nullptr, *pjobs_host = nullptr, *pjobs_dev = nullptr;
  size_t pack_cap = 0;            // int32 elements
  // ---- tick engine -----------------------------------------------------------------------------------------------------
  std::vector<St> st;
  std::vector<Run> run;
  std::vector<std::deque<Blk>> bq;
  std::vector<Job> job;                    // the OLDEST outstanding chunk of a stream (reported next)
  std::vector<std::deque<Job>> ahead;      // chunks queued behind it (queue_depth > 1)
  std::vector<long> job_seq;               // sequence number of the stream's latest chunk
  int queue_depth = 1;                     // outstanding chunks a stream may have (sc_streams_set_queue_depth)
  std::vector<Snap> snap;
  std::vector<long> done_at;               // sc_poll: order in which the streams' oldest chunks became complete (0: not yet)
  long done_counter = 0;
  int32_t *snap_yseq = nullptr, *snap_xpos = nullptr;   // [S*W][LCAP]
  double *snap_score = nullptr;                         // [S*W][3]
  hipEvent_t ev_snap = nullptr;
  std::vector<std::string> fault_msg;
  std::vector<long> enc_gen;       // group of the stream's latest encoder stage
  std::deque<EncGroup *> groups;   // planned or in flight, oldest first
  long gen_next = 1, gen_done = 0, gen_ordered = 0;   // groups: issued, known complete, main stream ordered behind
  int n_open = 0;                  // outstanding chunks
  long iter = 0;
  bool inflight = false;           // a decode step has been enqueued and not collected yet (between two sc_poll calls)
  std::vector<int> tick_active;    // ... for these streams
  std::chrono::steady_clock::time_point tick_t0, tick_t1;
  hipStream_t stream_rb = nullptr; // hypothesis read-back (sc_get_hyps_batch) beside an enqueued decode step
  bool poisoned = false;           // a call failed half-way: device and host state may disagree
  std::vector<int> rowmap_key;
  int row_bucket = 1, n_rows_step = 0;
  bool decode_prepared = false;
  std::map<int, hipGraphExec_t> dec_graphs;
  std::map<std::vector<long>, hipGraphExec_t> enc_graphs;
  long dec_steps = 0, dec_blocks = 0, enc_calls = 0, xattn_rows[3] = {0, 0, 0};   // [0] flash kernels, [1] head-parallel layer kernels, [2] stream-resident layer kernel
  long sattn_pos[3] = {0, 0, 0};           // token positions the self-attention covered (sum of L over steps, streams, layers)
  unsigned long long *stat_rows = nullptr; // device [3]: distinct self-attention K|V rows read (sc_search.stat_rows)
  double t_launch = 0, t_wait = 0;               // seconds in the step loop: issuing, waiting for the flags
  double t_bucket[17] = {0};                     // ... by compaction bucket (n_rows_step / (row_bucket*W))
  long n_bucket[17] = {0};
  bool use_graphs = true;
  long n_enc_captures = 0;        // encoder-layer graphs captured (one per group shape) and the host time that took
  double t_enc_capture = 0;

  ~sc_streams() {
    for (auto &g : dec_graphs) (void)hipGraphExecDestroy(g.second);
    for (auto &g : enc_graphs) (void)hipGraphExecDestroy(g.second);
    if (stream) (void)sc_set_stream_workspace(stream, nullptr, 0);
    if (stream_enc) {
      (void)sc_set_stream_workspace(stream_enc, nullptr, 0);
      (void)hipStreamDestroy(stream_enc);
    }
    for (int i = 0; i < 2; ++i)
      if (ev_iter[i]) (void)hipEventDestroy(ev_iter[i]);
    for (int i = 0; i < N_ARENA; ++i)
      if (ev_group[i]) (void)hipEventDestroy(ev_group[i]);
    for (int i = 0; i < N_STAGE; ++i)
      if (ev_stage[i]) (void)hipEventDestroy(ev_stage[i]);
    if (ev_snap) (void)hipEventDestroy(ev_snap);
    for (EncGroup *g : groups) delete g;
    for (void *p : owned) (void)hipFree(p);
    for (void *p : owned_host) (void)hipHostFree(p);
    if (stream_rb) (void)hipStreamDestroy(stream_rb);
    if (stream) (void)hipStreamDestroy(stream);
  }

  template <typename T>
  int alloc(T **p, size_t n) {
    RC_TRY(dalloc(p, n));
    owned.push_back(*p);
    return SC_OK;
  }
  template <typename T>
  int halloc(T **p, size_t n) {   // pinned, device-visible
    HIP_TRY(hipHostMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    owned_host.push_back(*p);
    memset(*p, 0, std::max<size_t>(n, 1) * sizeof(T));
    return SC_OK;
  }
  int32_t *ctrl_host() { return ctrlmap_host; }              // [S][8]
  int32_t *rowmap_host() { return rm_host[rm_idx]; }          // [S*W]

  // host int table -> device (slot `cur_slot` of the pinned arena, async copy on the encoder-side stream; the launches
  // that read it are ordered behind the copy; a slot is recycled when its group's event has completed)
  int itensor(const std::vector<int32_t> &a, const int32_t **out) {
    const size_t n = a.size();
    if (arena_off + n > slot_cap) {
      sc_set_error("job-table arena exhausted (%zu + %zu of %zu entries)", arena_off, n, slot_cap);
      return SC_ERR_ARG;
    }
    const size_t off = (size_t)cur_slot * slot_cap + arena_off;
    arena_off += (n + 63) & ~size_t(63);
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

// create_initial_hypothesis (hypothesis.py:75-91): yseq=[sos], xpos=[0], scores 0 - side 0, slot 0.  A kernel on
// the batch's stream: ordered behind whatever still reads the old hypotheses, no host synchronisation.
__global__ void init_hyp_kernel(sc_search sb, int s) {
  const size_t o = ((size_t)0 * sb.S + s) * sb.W + 0;
  sb.yseq[o * sb.LCAP] = sb.sos;
  sb.xpos[o * sb.LCAP] = 0;
  sb.anc[((size_t)0 * sb.S + s) * sb.LCAP * sb.W] = 0;   // K|V pool row of the sos token (position 0, hypothesis 0): row 0
  sb.score[o] = 0.0;
  sb.sc_dec[o] = 0.0;
  sb.sc_ctc[o] = 0.0;
}

// host input of an admission group, staged contiguously -> its places in the per-stream buffers (PCM ring rows /
// feature-buffer rows).  jobs[j] = {staging offset, destination offset, count} in floats.
__global__ __launch_bounds__(256) void scatter_f32_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                          const long long *__restrict__ jobs) {
  const long long so = jobs[blockIdx.y * 3], dof = jobs[blockIdx.y * 3 + 1], n = jobs[blockIdx.y * 3 + 2];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[dof + i] = src[so + i];
}

// live hypotheses of the listed streams -> one packed buffer (sc_get_hyps_batch).  jobs[j] = {hypothesis row
// ((side*S + s)*W + h), L, destination offset (int32 units), 0}: ids at [off, off+L), positions at [off+L, off+2L),
// then the three float64 totals (score, decoder, ctc) at the next 8-byte boundary.
// jobs[j][3] = 1: the row is a row (s*W + h) of the snapshot copies (snap_*) instead of the live state.
__global__ __launch_bounds__(128) void pack_hyps_kernel(sc_search sb, const int32_t *__restrict__ jobs, int32_t *__restrict__ out,
                                                        const int32_t *__restrict__ snap_yseq, const int32_t *__restrict__ snap_xpos,
                                                        const double *__restrict__ snap_score) {
  const int32_t *j = jobs + (size_t)blockIdx.x * 4;
  const size_t row = (size_t)j[0];
  const int L = j[1];
  const bool snap = j[3] != 0;
  int32_t *o = out + (size_t)j[2];
  const int32_t *ys = (snap ? snap_yseq : sb.yseq) + row * sb.LCAP, *xp = (snap ? snap_xpos : sb.xpos) + row * sb.LCAP;
  for (int i = threadIdx.x; i < L; i += 128) {
    o[i] = ys[i];
    o[L + i] = xp[i];
  }
  if (threadIdx.x == 0) {
    double *sc = (double *)(o + ((2 * L + 1) & ~1));
    sc[0] = snap ? snap_score[row * 3] : sb.score[row];
    sc[1] = snap ? snap_score[row * 3 + 1] : sb.sc_dec[row];
    sc[2] = snap ? snap_score[row * 3 + 2] : sb.sc_ctc[row];
  }
}

// live hypotheses of up to 32 streams -> the snapshot copies (grid: entry x hypothesis slot)
struct SnapArgs { int n; int s[32], cur[32], L[32], nhyp[32]; };
__global__ __launch_bounds__(128) void snapshot_hyps_kernel(sc_search sb, SnapArgs a, int32_t *__restrict__ snap_yseq,
                                                            int32_t *__restrict__ snap_xpos, double *__restrict__ snap_score) {
  const int e = blockIdx.x, h = blockIdx.y;
  if (h >= a.nhyp[e]) return;
  const size_t src = ((size_t)a.cur[e] * sb.S + a.s[e]) * sb.W + h, dst = (size_t)a.s[e] * sb.W + h;
  for (int i = threadIdx.x; i < a.L[e]; i += 128) {
    snap_yseq[dst * sb.LCAP + i] = sb.yseq[src * sb.LCAP + i];
    snap_xpos[dst * sb.LCAP + i] = sb.xpos[src * sb.LCAP + i];
  }
  if (threadIdx.x == 0) {
    snap_score[dst * 3] = sb.score[src];
    snap_score[dst * 3 + 1] = sb.sc_dec[src];
    snap_score[dst * 3 + 2] = sb.sc_ctc[src];
  }
}

void init_hyp(sc_streams *b, int s) {
  init_hyp_kernel<<<1, 1, 0, b->stream>>>(b->sb, s);
}

void reset_stream(sc_streams *b, int s) {
  // Speech2TextStreaming.reset + BlockwiseSynchronousBeamSearch.reset (speech2text_streaming.py:252-263,
  // beam_search.py:343-356).  strict: CTCPrefixScorer.impl is never cleared by the reference
  // (scorers.py:342-350: the stale table stays) and the short-segment PE counter keeps counting (A13).
  St old = b->st[s], ns;
  if (b->strict) {
    ns.short_pos = old.short_pos;
    ns.T_ctc = old.T_ctc;
    ns.T_proj = old.T_ctc;   // rows behind the stale table's extent belong to the next utterance
  }
  b->st[s] = ns;
  b->run[s] = Run();
  b->bq[s].clear();
  b->snap[s] = Snap();
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
    if (b->enc_graphs.size() >= 192) return launch();   // (one per group shape; continuous batching varies it)
    const auto tc0 = std::chrono::steady_clock::now();
    RC_TRY(sc_graph_capture_begin(b->es));
    const int rc = launch();
    void *g = nullptr;
    const int rc2 = sc_graph_capture_end(b->es, &g);
    b->n_enc_captures++;
    b->t_enc_capture += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
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
                 SC_GEMM_RELU | b->gemm_flags, F1, b->es));
  RC_TRY(b->itensor(P.lin_dst, &ld));
  RC_TRY(sc_gemm(b->c2, nullptr, F2 * d, e->f("sub_out_w"), e->f("sub_out_b"), b->subbuf, ld, d, (int)P.lin_dst.size(), d,
                 F2 * d, b->gemm_flags, 0, b->es));
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

// one beam-search step for every active stream (the prune kernel stores the stop flags straight into the host-mapped
// flag array).  CTC prefix scan split over T while few streams are active (a function of the compaction bucket, so
// every captured graph has one form): the sequential walk costs 0.15 us per frame whatever the number of streams.
int decode_step_launch(sc_streams *b) {
  b->sb.n_rows = b->n_rows_step;
  b->sb.flags = b->ring_dev;
  auto step = [&]() { return sc_decode_step_ex(&b->sb, b->step_split_min(), b->stream); };
  if (!b->use_graphs) return step();
  auto it = b->dec_graphs.find(b->graph_key());
  if (it == b->dec_graphs.end()) {
    RC_TRY(step());   // warm-up launch (also validates arguments); executes the step
    RC_TRY(sc_graph_capture_begin(b->stream));
    const int rc = step();
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
  HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrlmap_host, (size_t)b->S * 8 * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  const int keep = b->n_rows_step;
  for (int nb = b->row_bucket; nb < b->S + b->row_bucket; nb += b->row_bucket) {
    b->n_rows_step = std::min(nb, b->S) * b->W;
    if (!b->dec_graphs.count(b->graph_key())) RC_TRY(decode_step_launch(b));
  }
  b->n_rows_step = keep;
  HIP_TRY(hipStreamSynchronize(b->stream));   // (the pinned ctrl image is rewritten by the first real step)
  return SC_OK;
}

// ---- encoder groups -----------------------------------------------------------------------------------------------
// groups complete in order (one in-order stream): pop every leading group whose event has fired
int retire_groups(sc_streams *b) {
  while (!b->groups.empty()) {
    EncGroup *g = b->groups.front();
    if (!g->launched) break;
    const hipError_t q = hipEventQuery(b->ev_group[g->slot]);
    if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
    if (q != hipSuccess) {
      sc_set_error("encoder group %ld failed: %s", g->gen, hipGetErrorString(q));
      return SC_ERR_LAUNCH;
    }
    b->gen_done = g->gen;
    b->slot_gen[g->slot] = 0;
    b->groups.pop_front();
    delete g;
  }
  return SC_OK;
}

// host-blocking wait for the group `gen` (and, the stream being in order, every group before it)
int launch_pending_groups(sc_streams *b, long upto = -1);
int wait_group(sc_streams *b, long gen) {
  if (gen <= b->gen_done) return SC_OK;
  RC_TRY(launch_pending_groups(b, gen));
  const int slot = (int)(gen % N_ARENA);
  HIP_TRY(hipEventSynchronize(b->ev_group[slot]));
  return retire_groups(b);
}

// CTC rows and cross-attention K|V rows of the frames a group emits (extend_scorers' encoder-side half,
// beam_search.py:403-464; scorers.py:342-350 stores RAW logits - the log-softmax of a stream's first block, quirk
// A1, is applied when that block starts).  They depend on the encoder output only, so they belong to the encoder
// stage: a decode block then starts with nothing but its CTC state extension.
int project_rows(sc_streams *b, EncGroup &g) {
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  const int d = c.d_model, Ld = c.dec_layers, V = c.vocab_size;
  const int32_t *ar = nullptr;
  if (!g.ctc_rows.empty()) {
    RC_TRY(b->itensor(g.ctc_rows, &ar));
    RC_TRY(sc_gemm(b->enc, ar, d, e->f("ctc_w"), e->f("ctc_b"), const_cast<float *>(b->sb.ctcx), ar, V, (int)g.ctc_rows.size(),
                   V, d, b->gemm_flags, 0, b->es));
  }
  if (!g.kv_dst.empty()) {
    if (!g.same_rows || !ar) RC_TRY(b->itensor(g.kv_src, &ar));
    const int32_t *kvt;
    RC_TRY(b->itensor(g.kv_dst, &kvt));   // one row table for all layers: layer li's rows start li*TCAP rows further
    const int m = (int)g.kv_dst.size();
    // all layers in one launch (same sums: one k chain per element); per layer when that form does not apply
    const bool all = e->wkv_all && (2 * d) % 128 == 0;
    if (!b->sb.kv_half) {
      if (all && sc_gemm_colblocks(b->enc, ar, d, e->wkv_all, e->bkv_all, const_cast<float *>(b->sb.ckv), kvt, 2 * d, m, Ld * 2 * d, d,
                                   2 * d, (long)b->TCAP * 2 * d, b->gemm_flags, b->es) == SC_OK)
        return SC_OK;
      for (int li = 0; li < Ld; ++li)
        RC_TRY(sc_gemm(b->enc, ar, d, e->wkv[li], e->bkv[li], const_cast<float *>(b->sb.ckv) + (size_t)li * b->TCAP * 2 * d,
                       kvt, 2 * d, m, 2 * d, d, b->gemm_flags, 0, b->es));
    } else {
      // fp16 cache: project into the fp32 staging buffer (dense rows), then convert + scatter all layers at once
      for (int r0 = 0; r0 < m; r0 += b->kv_stage_rows) {
        const int mm = std::min(b->kv_stage_rows, m - r0);
        if (!(all && sc_gemm_colblocks(b->enc, ar + r0, d, e->wkv_all, e->bkv_all, b->kv_stage, nullptr, 2 * d, mm, Ld * 2 * d, d, 2 * d,
                                       (long)mm * 2 * d, b->gemm_flags, b->es) == SC_OK))
          for (int li = 0; li < Ld; ++li)
            RC_TRY(sc_gemm(b->enc, ar + r0, d, e->wkv[li], e->bkv[li], b->kv_stage + (size_t)li * mm * 2 * d, nullptr, 2 * d,
                           mm, 2 * d, d, b->gemm_flags, 0, b->es));
        RC_TRY(sc_kv_rows_to_half(b->kv_stage, kvt + r0, mm, Ld, b->TCAP, 2 * d, const_cast<float *>(b->sb.ckv), b->es));
      }
    }
  }
  return SC_OK;
}

// host input of an admission -> device, at once: ONE host-to-device copy of the pinned staging slot, one scatter
// launch to the chunks' places behind what each stream has buffered (PCM ring / feature buffer)
int stage_copy(sc_streams *b, EncGroup &g) {
  if (g.copy_jobs.empty()) return SC_OK;
  hipStream_t es = b->stream_enc ? b->stream_enc : b->stream;
  const int ss = g.stage_slot;
  const size_t so = (size_t)ss * b->stage_cap, jo = (size_t)ss * b->S * 3;
  const int nj = (int)g.copy_jobs.size() / 3;
  memcpy(b->cjobs_host + jo, g.copy_jobs.data(), g.copy_jobs.size() * sizeof(long long));
  HIP_TRY(hipMemcpyAsync(b->stage_dev + so, b->stage_host + so, g.stage_floats * sizeof(float), hipMemcpyHostToDevice, es));
  HIP_TRY(hipMemcpyAsync(b->cjobs_dev + jo, b->cjobs_host + jo, g.copy_jobs.size() * sizeof(long long), hipMemcpyHostToDevice, es));
  long long mx = 1;
  for (int j = 0; j < nj; ++j) mx = std::max(mx, g.copy_jobs[j * 3 + 2]);
  dim3 grid((unsigned)std::min<long long>(64, (mx + 1023) / 1024), nj);
  scatter_f32_kernel<<<grid, 256, 0, es>>>(b->stage_dev + so, g.features ? b->featbuf : b->pcm, b->cjobs_dev + jo);
  HIP_TRY(hipEventRecord(b->ev_stage[ss], es));   // the pinned slot is free again behind this
  b->stage_busy[ss] = true;
  g.copy_jobs.clear();
  return SC_OK;
}

// src (planned after dst, disjoint streams) joins dst: job tables are concatenated, indices into the group-local
// scratch buffers (conv1 output rows, encoder block slots) are shifted behind dst's
void merge_group(sc_streams *b, EncGroup &dst, EncGroup &src) {
  const sc_config &c = b->cfg;
  const int F1 = c.conv_freq1, R = c.block_size + 2;
  EncPlan &D = dst.P, &P = src.P;
  int c1_rows = 0;
  if (D.n_conv) c1_rows = D.conv_jobs[(D.n_conv - 1) * 4 + 2] + D.conv_jobs[(D.n_conv - 1) * 4 + 3];
  const int nblk = (int)D.blk_jobs.size() / 6;
  for (int j = 0; j < P.n_conv; ++j) P.conv_jobs[j * 4 + 2] += c1_rows;
  for (auto &v : P.a_rows) v += c1_rows * F1;
  for (size_t j = 0; j < P.sjobs.size(); j += 5) P.sjobs[j] += nblk;
  for (auto &v : P.emit_src)
    if (v >= 0) v += nblk * R;
  auto cat = [](std::vector<int32_t> &a, const std::vector<int32_t> &x) { a.insert(a.end(), x.begin(), x.end()); };
  cat(D.conv_jobs, P.conv_jobs); cat(D.a_rows, P.a_rows); cat(D.lin_dst, P.lin_dst); cat(D.feat_src, P.feat_src);
  cat(D.feat_dst, P.feat_dst); cat(D.blk_jobs, P.blk_jobs); cat(D.sjobs, P.sjobs); cat(D.emit_src, P.emit_src);
  cat(D.emit_dst, P.emit_dst); cat(D.sub_src, P.sub_src); cat(D.sub_dst, P.sub_dst);
  D.shorts.insert(D.shorts.end(), P.shorts.begin(), P.shorts.end());
  D.n_conv += P.n_conv;
  D.max_t1 = std::max(D.max_t1, P.max_t1);
  cat(dst.fe_jobs, src.fe_jobs);
  dst.n_fe += src.n_fe;
  dst.max_keep = std::max(dst.max_keep, src.max_keep);
  cat(dst.ctc_rows, src.ctc_rows); cat(dst.kv_src, src.kv_src); cat(dst.kv_dst, src.kv_dst);
  dst.same_rows = dst.same_rows && src.same_rows;
  dst.streams.insert(dst.streams.end(), src.streams.begin(), src.streams.end());
}

// issue a planned group: frontend, encoder, projections (on the encoder stream when there is one)
int launch_group(sc_streams *b, EncGroup *g) {
  if (g->launched) return SC_OK;
  g->launched = true;
  g->open = false;
  const sc_config &c = b->cfg;
  sc_engine *e = b->eng;
  b->es = b->stream_enc ? b->stream_enc : b->stream;
  b->cur_slot = g->slot;
  b->arena_off = 0;
  int rc = SC_OK;
  if (rc == SC_OK && g->n_fe) {
    const int32_t *jobs;
    rc = b->itensor(g->fe_jobs, &jobs);
    if (rc == SC_OK)
      rc = sc_logmel(b->pcm, (int)b->PCAP, jobs, g->n_fe, g->max_keep, e->f("window"), e->f("mel_fb"), e->f("twiddle"),
                     (const double *)e->f("mean64"), (const double *)e->f("std64"), c.mvn_mode, c.n_fft, c.hop_length,
                     c.win_length, c.n_mels, b->featbuf, b->es);
  }
  if (rc == SC_OK && g->P.n_conv) rc = encode_launch(b, g->P);
  if (rc == SC_OK) rc = project_rows(b, *g);
  (void)hipEventRecord(b->ev_group[g->slot], b->es);
  if (b->es == b->stream) b->gen_ordered = std::max(b->gen_ordered, g->gen);
  b->es = b->stream;
  return rc;
}

// groups are launched in order; upto >= 0: only the groups up to that generation
int launch_pending_groups(sc_streams *b, long upto) {
  for (EncGroup *g : b->groups)
    if (!g->launched && (upto < 0 || g->gen <= upto)) RC_TRY(launch_group(b, g));
  return SC_OK;
}

// ---- the tick engine ------------------------------------------------------------------------------------------------
// _decode_one_block's epilogue for a stream whose step loop has ended: rewind (beam_search.py:827-836), back to the
// stream state; a non-final block advances the schedule (:619-621)
void finish_block(sc_streams *b, int s) {
  Run &r = b->run[s];
  const bool rw = r.pidx > 1 && r.pvalid;
  const bool r2 = rw && r.took;   // live state is a non-accepted H_out: go back to its H_in
  if (r2) {
    r.cur = 1 - r.cur;
    r.L -= 1;
    r.nhyp = r.nhp;
    r.has = r.hasp;
  }
  if (rw) { r.pidx -= 1; r.pvalid = false; }
  St &x = b->st[s];
  x.cur = r.cur; x.L = r.L; x.nhyp = r.nhyp; x.has_ctc = r.has;
  x.process_idx = r.pidx; x.prev_valid = r.pvalid;
  x.output_index = r.out;
  x.n_steps_total += r.nsteps;
  if (!r.fin) x.processed_block += 1;
  r.inblk = false;
}

// the block the stream is inside cannot go on: the chunk that owns it fails, and with it the chunks queued behind it
// (blocks run in order: every earlier chunk of the stream has been decoded by then)
void fault_stream(sc_streams *b, int s, int code, const std::string &msg) {
  const long seq = b->run[s].inblk ? b->run[s].job : b->job[s].seq;
  bool hit = false;
  auto mark = [&](Job &j) {
    if (!j.open) return;
    if (j.seq == seq) { j.fault = code; hit = true; }
    else if (hit && !j.fault) { j.fault = code; j.dropped = true; }
  };
  mark(b->job[s]);
  for (Job &j : b->ahead[s]) mark(j);
  b->fault_msg[s] = msg;
  b->run[s] = Run();
  b->bq[s].clear();
}

inline bool chunk_decoded(const sc_streams *b, int s);

// a block of a LATER chunk of the stream starts only when the hypotheses of the chunk before it have been copied
// aside (snapshot_completed; queue depth > 1) - until then the stream is held back
inline bool may_start(const sc_streams *b, int s) {
  const Job &j = b->job[s];
  const long q = b->bq[s].front().job;
  if (!j.open || q <= j.seq) return true;
  return q == j.seq + 1 && b->snap[s].valid && b->snap[s].seq == j.seq;
}

// queue depth > 1: the hypotheses of every stream whose oldest chunk has just been decoded -> the snapshot copies, on
// the batch's stream, i.e. before any later step changes them.  One snapshot per stream: a chunk that completes while
// the previous one's snapshot has not been handed out yet (sc_poll) waits, and holds the stream's later blocks back.
int snapshot_completed(sc_streams *b) {
  if (b->queue_depth <= 1) return SC_OK;
  SnapArgs a;
  a.n = 0;
  bool any = false;
  auto flush = [&]() -> int {
    if (!a.n) return SC_OK;
    snapshot_hyps_kernel<<<dim3(a.n, b->W), 128, 0, b->stream>>>(b->sb, a, b->snap_yseq, b->snap_xpos, b->snap_score);
    SC_CHECK_LAUNCH();
    a.n = 0;
    return SC_OK;
  };
  for (int s = 0; s < b->S; ++s) {
    const Job &j = b->job[s];
    Snap &sn = b->snap[s];
    if (!j.open || j.fault || sn.valid || !chunk_decoded(b, s)) continue;
    const St &x = b->st[s];
    sn.valid = true;
    sn.reported = false;
    sn.seq = j.seq;
    sn.L = x.L;
    sn.nhyp = j.started ? x.nhyp : 0;
    if (sn.nhyp > 0) {
      a.s[a.n] = s; a.cur[a.n] = x.cur; a.L[a.n] = x.L; a.nhyp[a.n] = x.nhyp;
      a.n++;
      any = true;
      if (a.n == 32) RC_TRY(flush());
    }
  }
  RC_TRY(flush());
  if (any) HIP_TRY(hipEventRecord(b->ev_snap, b->stream));
  return SC_OK;
}

// idle streams whose next block is ready start it: extend_scorers' search-side half (beam_search.py:403-464) -
// log-softmax of the first block's CTC rows (quirk A1), CTC state extension - and the loop state of the block
int start_blocks(sc_streams *b) {
  const int S = b->S;
  const long gen_ok = std::max(b->gen_done, b->gen_ordered);
  std::vector<int> starts;
  for (int s = 0; s < S; ++s)
    if (!b->run[s].inblk && !b->bq[s].empty() && b->bq[s].front().gen <= gen_ok && may_start(b, s)) starts.push_back(s);
  if (starts.empty()) return SC_OK;
  if (!b->decode_prepared) {
    b->decode_prepared = true;
    RC_TRY(prepare_decode(b));
  }
  b->bs_i = (b->bs_i + 1) % N_RING;
  int32_t *ctrl0 = b->bs_host + (size_t)b->bs_i * b->bs_cap, *lsm = ctrl0 + (size_t)S * 8;
  memset(ctrl0, 0, (size_t)S * 8 * sizeof(int32_t));
  int n_lsm = 0;
  for (int s : starts) {
    const Blk k = b->bq[s].front();
    b->bq[s].pop_front();
    St &x = b->st[s];
    Run &r = b->run[s];
    r = Run();
    r.inblk = r.live = true;
    r.T = k.T; r.fin = k.fin; r.job = k.job;
    r.cur = x.cur; r.L = x.L; r.nhyp = x.nhyp; r.has = x.has_ctc; r.pidx = x.process_idx; r.pvalid = x.prev_valid;
    const int told = x.T_ctc;
    r.Tc = std::max(k.T, told);   // the CTC table never shrinks (stale table after reset())
    if (k.T > told && told == 0)  // quirk A1: only the rows of the stream's first block are log-softmaxed
      for (int t = 0; t < k.T && n_lsm < (int)(b->bs_cap - (size_t)S * 8); ++t) lsm[n_lsm++] = s * b->TCAP + t;
    int32_t *c = ctrl0 + s * 8;
    c[0] = 1; c[1] = r.cur; c[2] = r.fin; c[3] = r.Tc; c[4] = r.L; c[5] = r.nhyp; c[6] = r.has; c[7] = told;
    x.T_ctc = r.Tc;
    x.T_kv = std::max(k.T, x.T_kv);
    x.output_index = 0;
    r.nhp = r.nhyp; r.hasp = r.has;
    b->dec_blocks++;
  }
  HIP_TRY(hipMemcpyAsync(b->ctrlmap, ctrl0, (size_t)S * 8 * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  if (n_lsm) {
    int32_t *ld = b->bs_dev + (size_t)b->bs_i * b->bs_cap;
    HIP_TRY(hipMemcpyAsync(ld, lsm, (size_t)n_lsm * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    RC_TRY(sc_log_softmax_rows(const_cast<float *>(b->sb.ctcx), ld, n_lsm, b->cfg.vocab_size, b->stream));
  }
  RC_TRY(sc_ctc_extend_state(&b->sb, b->stream));
  return SC_OK;
}

// One tick = issue (start the ready blocks, close the ended ones, enqueue ONE beam-search step for every stream that is
// inside a block) + collect (wait for the step, read its stop flags, apply the accept / stop rules).  sc_poll returns
// to its caller BETWEEN the two: the device runs the step while the host handles the replies of the streams that
// finished (read-back on the read-back stream, the next chunks' admission on the encoder stream).
// *progress: a block was started or finished, or a step was enqueued.  Nothing to do -> *progress = false.
int tick_collect(sc_streams *b);
int tick_issue(sc_streams *b, bool *progress) {
  const int S = b->S, W = b->W, Ld = b->cfg.dec_layers;
  *progress = false;
  if (b->inflight) RC_TRY(tick_collect(b));
  RC_TRY(retire_groups(b));
  RC_TRY(snapshot_completed(b));
  RC_TRY(start_blocks(b));
  {
    // a queued block that sees frames of a group which has not even been issued (sc_submit defers and merges the
    // encoder stages of successive admissions): issue it now, it will have completed a few ticks on
    long need = 0;
    for (int s = 0; s < S; ++s)
      if (!b->run[s].inblk && !b->bq[s].empty()) need = std::max(need, b->bq[s].front().gen);
    long upto = -1;
    for (EncGroup *g : b->groups)
      if (!g->launched && g->deferred && g->gen <= need) upto = g->gen;
    if (upto >= 0) RC_TRY(launch_pending_groups(b, upto));
  }
  // streams whose step loop has ended (stop flags of the previous tick, or process_idx at its bound: :701)
  std::vector<int> &active = b->tick_active;
  active.clear();
  for (int s = 0; s < S; ++s) {
    Run &r = b->run[s];
    if (!r.inblk) continue;
    const bool act = r.live && r.pidx < b->max_length;
    if (act && r.L + 1 > b->LCAP) {   // the stream leaves the loop here and is reset when its chunk is reported
      fault_stream(b, s, SC_ERR_CAPACITY, "max_tokens exceeded");
      *progress = true;
      continue;
    }
    if (!act) {
      finish_block(b, s);
      *progress = true;
      continue;
    }
    active.push_back(s);
  }
  if (active.empty()) {
    if (*progress) return SC_OK;
    // nobody is decoding: a block that waits for its encoder group is ordered behind it ON THE DEVICE
    long need = 0;
    for (int s = 0; s < S; ++s)
      if (!b->run[s].inblk && !b->bq[s].empty() && (need == 0 || b->bq[s].front().gen < need)) need = b->bq[s].front().gen;
    if (need > std::max(b->gen_done, b->gen_ordered)) {
      RC_TRY(launch_pending_groups(b, need));
      if (b->stream_enc) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_group[need % N_ARENA], 0));
      b->gen_ordered = std::max(b->gen_ordered, need);
      *progress = true;
    }
    return SC_OK;
  }
  *progress = true;
  b->tick_t0 = std::chrono::steady_clock::now();
  set_rowmap(b, active);
  b->scan_long = false;   // the graph with the T-parallel scan only when a stream's table is long enough for it
  int32_t *ctrl = b->ctrl_host();
  memset(ctrl, 0, (size_t)S * 8 * sizeof(int32_t));
  for (int s : active) {
    const Run &r = b->run[s];
    if (b->scan_split_min > 0 && r.Tc - std::max(r.L - 1, 1) >= b->scan_split_min) b->scan_long = true;
    int32_t *c = ctrl + s * 8;
    c[0] = 1; c[1] = r.cur; c[2] = r.fin; c[3] = r.T; c[4] = r.L; c[5] = r.nhyp; c[6] = r.has; c[7] = r.Tc;
  }
  // sc_push: the planned encoder stage fills the CUs that the thinned-out step loop leaves idle
  if ((int)active.size() <= b->enc_start_thr)
    for (EncGroup *g : b->groups)
      if (!g->launched && !g->open) RC_TRY(launch_group(b, g));
  if (b->rm_dirty) {   // one copy command for ctrl rows + rowmap (nothing is in flight here: the pinned image is free)
    memcpy(b->ctrlmap_host + S * 8, b->rm_host[b->rm_idx], (size_t)S * W * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrlmap_host, (size_t)(S * 8 + S * W) * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    b->rm_dirty = false;
  } else {
    HIP_TRY(hipMemcpyAsync(b->ctrlmap, b->ctrlmap_host, (size_t)S * 8 * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  }
  RC_TRY(decode_step_launch(b));
  HIP_TRY(hipEventRecord(b->ev_iter[b->iter & 1], b->stream));
  const int form = sc_decode_step_form(&b->sb);   // 0 six launches per layer, 1 head-parallel, 2 stream-resident
  for (int s : active) {   // bench roofline: K|V rows the cross-attention reads, positions the self-attention covers
    b->xattn_rows[form] += (long)b->run[s].T * Ld;
    b->sattn_pos[form] += (long)b->run[s].L * Ld;
  }
  b->tick_t1 = std::chrono::steady_clock::now();
  b->inflight = true;
  return SC_OK;
}

int tick_collect(sc_streams *b) {
  if (!b->inflight) return SC_OK;
  const int S = b->S, W = b->W;
  b->inflight = false;
  const auto tpw = std::chrono::steady_clock::now();
  HIP_TRY(hipEventSynchronize(b->ev_iter[b->iter & 1]));   // the stop flags live in host-mapped pinned memory
  const auto tp2 = std::chrono::steady_clock::now();
  b->iter++;
  b->dec_steps++;
  b->t_launch += std::chrono::duration<double>(b->tick_t1 - b->tick_t0).count();
  b->t_wait += std::chrono::duration<double>(tp2 - tpw).count();
  {
    const int bk = std::min(16, (int)((long)b->n_rows_step * 16 / std::max(1, S * W)));
    b->t_bucket[bk] += std::chrono::duration<double>(tp2 - b->tick_t0).count();
    b->n_bucket[bk] += 1;
  }
  // ---- the accept / stop rules of the step loop (:759-821)
  for (int s : b->tick_active) {
    Run &r = b->run[s];
    // the prune kernel found no free K|V pool row for the new hypotheses' newest tokens: the side it wrote cannot be
    // decoded from.  That only matters if the step is TAKEN and stays taken (see below): a step that is rolled back
    // (stop_bbd) or rewound when the block closes leaves the current side, which is fully valid.
    const bool kv_full = b->flags_host[S + s] != 0;
    b->flags_host[S + s] = 0;
    const int f = b->flags_host[s];
    const bool f_any = f & F_ANY_EOS, f_best = f & F_BEST_EOS, f_all = f & F_ALL_EOS, f_rep = f & F_REPEAT;
    r.out += 1;
    r.nsteps += 1;
    const bool stop_eos = f_any && (!r.fin || f_best);
    const bool stop_bbd = b->use_bbd && !stop_eos && f_rep && !r.fin;
    const bool stop_all = !stop_eos && !stop_bbd && f_all && r.fin;
    const bool accept = !(stop_eos || stop_bbd || stop_all);
    const bool take = stop_eos || stop_all || accept;
    if (kv_full && take && !(!accept && r.pidx > 1 && r.pvalid)) {   // (not accepted + rewind pending: finish_block goes back to H_in)
      if (sc_hook("SC_DEBUG_POOL"))
        fprintf(stderr, "[scasr] pool exhausted: stream %d L %d nhyp %d T %d fin %d pidx %d flags %d kv_rows %d iter %ld\n", s, r.L, r.nhyp,
                r.T, (int)r.fin, r.pidx, f, b->sb.kv_rows, b->iter);
      fault_stream(b, s, SC_ERR_CAPACITY, "self-attention K|V pool exhausted (kv_pool_rows / max_tokens)");
      continue;
    }
    if (stop_bbd) r.out -= 1;
    if (take) {
      r.nhp = r.nhyp;
      r.hasp = r.has;
      r.cur = 1 - r.cur;
      r.L += 1;
      r.nhyp = std::min(W, r.nhyp * W);
      r.has = true;
    }
    if (stop_eos || stop_all) r.took = true;
    r.live = accept;
    if (accept) {
      r.pvalid = true;   // prev_hyps = copy(H_out)
      r.pidx += 1;       // process_idx += 1
    }
    // the step loop of this block has ended (stop flags, or process_idx at its bound: :701): close the block NOW, so
    // that the chunk counts as complete in the sc_poll call that collected its last step - closing it in the next
    // tick_issue cost a stream one more idle tick between its reply and its next chunk (11.15 -> 10.2 decode
    // iterations per chunk step at 128 streams)
    if (!(r.live && r.pidx < b->max_length)) finish_block(b, s);
  }
  return SC_OK;
}

int engine_tick(sc_streams *b, bool *progress) {
  RC_TRY(tick_issue(b, progress));
  return tick_collect(b);
}

// a stream's OLDEST outstanding chunk is decoded: the stream is not inside one of its blocks and none is queued
// (blocks of chunks queued behind it carry a larger Job::seq)
inline bool chunk_decoded(const sc_streams *b, int s) {
  const long q = b->job[s].seq;
  return !(b->run[s].inblk && b->run[s].job <= q) && (b->bq[s].empty() || b->bq[s].front().job > q);
}
// ... and can be reported: with a queue depth > 1 its hypotheses must have been copied aside first
inline bool chunk_complete(const sc_streams *b, int s) {
  const Job &j = b->job[s];
  if (!j.open) return false;
  if (j.fault != 0) return true;
  if (!chunk_decoded(b, s)) return false;
  return b->queue_depth <= 1 || (b->snap[s].valid && b->snap[s].seq == j.seq);
}

// ---- admission ------------------------------------------------------------------------------------------------------
struct Chunk { int s; const float *host; long n; bool fin; int pos; };

// the group that carries the stream's LATEST encoder stage (its frames are what "the frames that were there before"
// means for the next admission: a block that only sees those still has to wait for that group)
inline long st_gen(const sc_streams *b, int s) { return b->enc_gen[s] > b->gen_done ? b->enc_gen[s] : 0; }

// apply_frontend + forward_infer planning of one group, pure host work: stages the host input,
// advances the host mirrors, fills the group's job tables.  Throws StreamFault for per-stream failures.
void plan_group(sc_streams *b, const std::vector<Chunk> &chunks, bool features, EncGroup &g, std::map<int, int> &feat_new,
                std::map<int, bool> &finals, std::vector<int> &has_out) {
  const sc_config &c = b->cfg;
  float *stage = g.stage_slot >= 0 ? b->stage_host + (size_t)g.stage_slot * b->stage_cap : nullptr;
  for (size_t k = 0; k < chunks.size(); ++k) {
    const Chunk &ch = chunks[k];
    St &st = b->st[ch.s];
    finals[ch.s] = ch.fin;
    if (features) {
      // 2-D (T, n_mels) already-normalised features (speech2text_streaming.py:438-449): appended to the feature buffer
      const int nbuf = st.enc_started ? st.nfeat : 0;
      if (ch.n > b->max_feat_new || nbuf + ch.n > b->FCAP)
        throw StreamFault{ch.s, SC_ERR_CAPACITY, "feature frames of one call exceed the batch's capacity (max_chunk_samples)"};
      if (ch.n > 0) {
        const size_t cnt = (size_t)ch.n * c.n_mels;
        memcpy(stage + g.stage_floats, ch.host, cnt * sizeof(float));
        g.copy_jobs.insert(g.copy_jobs.end(), {(long long)g.stage_floats,
                                               (long long)(((size_t)(st.fpp * b->S + ch.s) * b->FCAP + nbuf) * c.n_mels), (long long)cnt});
        g.stage_floats += cnt;
      }
      feat_new[ch.s] = (int)ch.n;
      has_out[k] = 1;
      continue;
    }
    if (st.pcm_end + ch.n > b->PCAP) throw StreamFault{ch.s, SC_ERR_CAPACITY, "pcm buffer capacity exceeded"};
    if (ch.n > 0 && ch.host) {
      if (ch.n > b->max_chunk + 16 * c.hop_length)
        throw StreamFault{ch.s, SC_ERR_CAPACITY, "chunk is longer than max_chunk_samples allows"};
      memcpy(stage + g.stage_floats, ch.host, (size_t)ch.n * sizeof(float));
      g.copy_jobs.insert(g.copy_jobs.end(), {(long long)g.stage_floats, (long long)ch.s * b->PCAP + st.pcm_end, (long long)ch.n});
      g.stage_floats += (size_t)ch.n;
    }
    st.pcm_end += ch.n;
    FePlan p = plan_frontend(c, st, ch.fin);
    if (!p.emit) continue;
    if (p.n > b->max_feat_new) throw StreamFault{ch.s, SC_ERR_CAPACITY, "chunk produces more feature frames than max_chunk_samples allows"};
    const int nbuf = st.enc_started ? st.nfeat : 0;
    const int dst_row0 = (st.fpp * b->S + ch.s) * b->FCAP + nbuf;
    g.fe_jobs.insert(g.fe_jobs.end(), {ch.s, (int)p.seg_start, (int)p.seg_len, (int)p.eff_len, p.lo, p.n, dst_row0, 0});
    ++g.n_fe;
    g.max_keep = std::max(g.max_keep, p.n);
    feat_new[ch.s] = p.n;
    has_out[k] = 1;
  }
  // plan the encoder BEFORE anything that changes device state is launched: a per-stream fault thrown by the
  // planning leaves the device untouched (the staged input only appends behind pcm_end / the buffered features)
  std::vector<int> enc_streams;
  for (auto &kv : feat_new) {
    St &st = b->st[kv.first];
    if (!st.started) st.started = true;   // running_hyps = [initial hypothesis]
    if (kv.second >= 3) enc_streams.push_back(kv.first);   // n < 3: encoder skipped, frames discarded (:551-559)
  }
  if (!enc_streams.empty()) encode_plan(b, enc_streams, feat_new, finals, g.P);
}

// Admit the chunks of one call: plan them (per-stream faults drop that stream's chunk only), copy the host input
// to the device, queue the decode blocks the new frames make ready (beam_search.py:590-634), open the jobs.
// sc_push (defer = false): the admission is one encoder group of its own, issued by the tick loop when the decode
// loop has thinned out.  sc_submit (defer = true): the admission JOINS the open group - the frontend + encoder stages
// of successive small admissions run as one large batch (a chunk's own decode block only sees frames that were there
// before it, SURVEY A7, so its encoder stage has a whole chunk period of slack) - which is issued when it holds
// enc_batch_min streams, or as soon as a queued decode block needs its frames.
int admit(sc_streams *b, std::vector<Chunk> chunks, bool features, bool defer, std::vector<int> *has_out_by_pos) {
  const sc_config &c = b->cfg;
  RC_TRY(retire_groups(b));
  bool host_input = false;
  for (auto &ch : chunks) host_input = host_input || (ch.host && ch.n > 0);
  int sslot = -1;
  if (host_input) {
    sslot = b->stage_next;
    b->stage_next = (b->stage_next + 1) % N_STAGE;
    if (b->stage_busy[sslot]) HIP_TRY(hipEventSynchronize(b->ev_stage[sslot]));
    b->stage_busy[sslot] = false;
  }
  std::unique_ptr<EncGroup> gp(new EncGroup);   // owned here until it joins b->groups (or is merged / dropped)
  EncGroup *g = gp.get();
  std::map<int, int> feat_new;
  std::map<int, bool> finals;
  std::vector<int> has_out;
  std::vector<St> snap;
  while (true) {
    if (!features)
      for (auto &ch : chunks) {   // compaction moves device data: settle it before the snapshot
        St &st = b->st[ch.s];
        if (st.pcm_end + ch.n > b->PCAP && st.pcm_start > 0) {
          int rc = launch_pending_groups(b);   // (a frontend that still has to read the old places)
          if (rc == SC_OK && b->stream_enc && hipStreamSynchronize(b->stream_enc) != hipSuccess) rc = SC_ERR_LAUNCH;
          if (rc == SC_OK) rc = compact_pcm(b, ch.s);
          if (rc != SC_OK) return rc;
        }
      }
    snap = b->st;
    *g = EncGroup();
    g->stage_slot = sslot; g->features = features; g->deferred = defer;
    feat_new.clear();
    finals.clear();
    has_out.assign(chunks.size(), 0);
    try {
      plan_group(b, chunks, features, *g, feat_new, finals, has_out);
      break;
    } catch (const StreamFault &f) {
      b->st = snap;   // planning is pure host work that precedes every launch: undo = restore the mirrors
      for (size_t k = 0; k < chunks.size(); ++k)
        if (chunks[k].s == f.stream) {
          chunks.erase(chunks.begin() + k);
          break;
        }
      {
        Job fj;
        fj.open = true;
        fj.fault = f.code;
        fj.seq = ++b->job_seq[f.stream];
        b->fault_msg[f.stream] = f.msg;
        if (b->job[f.stream].open) b->ahead[f.stream].push_back(fj);   // the earlier chunks of the stream go on
        else {
          b->job[f.stream] = fj;
          b->run[f.stream] = Run();
          b->bq[f.stream].clear();
        }
        b->n_open++;
      }
      if (chunks.empty()) {   // every chunk of the call failed: nothing of the last (aborted) plan may survive
        *g = EncGroup();
        feat_new.clear();
        finals.clear();
        has_out.clear();
        break;
      }
    }
  }
  {
    const int rc = stage_copy(b, *g);
    if (rc != SC_OK) return rc;
  }
  // eager projections: CTC rows / cross-attention K|V rows of every frame this admission emits
  for (auto &ch : chunks) {
    St &st = b->st[ch.s];
    const int t0 = snap[ch.s].T_enc, t1 = st.T_enc;
    if (feat_new.count(ch.s)) g->streams.push_back(ch.s);
    if (t1 <= t0) continue;
    const int c0 = std::max(t0, st.T_proj), k0 = std::max(t0, st.T_projkv);
    g->same_rows = g->same_rows && c0 == k0;
    for (int t = c0; t < t1; ++t) g->ctc_rows.push_back(ch.s * b->TCAP + t);
    for (int t = k0; t < t1; ++t) {
      g->kv_src.push_back(ch.s * b->TCAP + t);
      g->kv_dst.push_back(ch.s * c.dec_layers * b->TCAP + t);
    }
    st.T_proj = std::max(st.T_proj, t1);
    st.T_projkv = std::max(st.T_projkv, t1);
  }
  // which group carries the work: the open one (sc_submit; never two admissions of one stream in a group), or a new one
  long gen = 0;
  if (!g->empty()) {
    EncGroup *open = (!b->groups.empty() && b->groups.back()->open && !b->groups.back()->launched) ? b->groups.back() : nullptr;
    if (open) {
      bool clash = !defer;
      for (int s : g->streams) clash = clash || std::find(open->streams.begin(), open->streams.end(), s) != open->streams.end();
      if (clash) {
        open->open = false;
        if (defer) {
          const int rc = launch_group(b, open);
          if (rc != SC_OK) return rc;
        }
        open = nullptr;
      }
    }
    if (open) {
      merge_group(b, *open, *g);
      gp.reset();
      g = open;
    } else {
      g->gen = b->gen_next++;
      g->slot = (int)(g->gen % N_ARENA);
      if (b->slot_gen[g->slot]) {   // N_ARENA groups in flight: wait for the oldest
        const int rc = wait_group(b, b->slot_gen[g->slot]);
        if (rc != SC_OK) return rc;
      }
      b->slot_gen[g->slot] = g->gen;
      g->open = defer;
      b->groups.push_back(gp.release());
    }
    gen = g->gen;
  } else {
    gp.reset();
    g = nullptr;
  }
  std::map<int, long> seq_of;   // the chunks of this admission: their per-stream sequence numbers
  for (auto &ch : chunks) seq_of[ch.s] = ++b->job_seq[ch.s];
  // decode schedule (beam_search.py:590-634) as per-stream queues; a block that sees frames of this admission waits
  // for the group that carries it
  for (auto &kv : feat_new) {
    const int s = kv.first;
    St &st = b->st[s];
    std::deque<Blk> &q = b->bq[s];
    int pb = st.processed_block + (b->run[s].inblk && !b->run[s].fin ? 1 : 0);
    for (auto &k : q) pb += k.fin ? 0 : 1;
    const int t_avail = st.T_enc, t_old = snap[s].T_enc;
    while (t_avail > 0) {
      const int cur_end = c.block_size - c.look_ahead + c.hop_size * pb;
      if (!(cur_end < t_avail)) break;
      q.push_back({cur_end, false, cur_end > t_old ? gen : st_gen(b, s), seq_of.at(s)});
      ++pb;
    }
    if (finals.at(s) && t_avail > 0) q.push_back({t_avail, true, t_avail > t_old ? gen : st_gen(b, s), seq_of.at(s)});
  }
  if (gen)
    for (int s : g->streams) b->enc_gen[s] = gen;
  for (size_t k = 0; k < chunks.size(); ++k) {
    Job j;
    j.open = true;
    j.has_out = has_out[k];
    j.seq = seq_of.at(chunks[k].s);
    j.fin = chunks[k].fin;
    j.started = b->st[chunks[k].s].started;
    if (b->job[chunks[k].s].open) b->ahead[chunks[k].s].push_back(j);   // behind the stream's outstanding chunk(s)
    else b->job[chunks[k].s] = j;
    b->n_open++;
    if (has_out_by_pos) (*has_out_by_pos)[chunks[k].pos] = has_out[k];
  }
  if (g && defer && !g->launched && (int)g->streams.size() >= std::max(1, b->enc_batch_min)) RC_TRY(launch_group(b, g));
  return SC_OK;
}

// a call failed half-way (launch error, exhausted arena): settle the device and refuse further work on the handle
int poison(sc_streams *b, int rc) {
  (void)hipStreamSynchronize(b->stream);
  if (b->stream_enc) (void)hipStreamSynchronize(b->stream_enc);
  (void)hipGetLastError();
  b->inflight = false;
  b->poisoned = true;
  return rc;
}

#define SC_API_BEGIN try {
#define SC_API_END                                                            \
  } catch (const std::bad_alloc &) {                                         \
    sc_set_error("%s: out of host memory", __func__);                        \
    return SC_ERR_ARG;                                                       \
  } catch (const std::exception &ex) {                                       \
    sc_set_error("%s: %s", __func__, ex.what());                             \
    return SC_ERR_ARG;                                                       \
  } catch (...) {                                                            \
    sc_set_error("%s: unexpected exception", __func__);                      \
    return SC_ERR_ARG;                                                       \
  }

// depth: outstanding chunks a stream may have INCLUDING the new one (sc_push, sc_push_features: 1)
int check_chunks(sc_streams *b, const int *stream_ids, const int *counts, int n, const char *what, int depth = 1) {
  std::vector<char> seen(b->S, 0);
  for (int i = 0; i < n; ++i) {
    if (stream_ids[i] < 0 || stream_ids[i] >= b->S || counts[i] < 0) {
      sc_set_error("%s: stream id / count out of range (entry %d)", what, i);
      return SC_ERR_ARG;
    }
    if (seen[stream_ids[i]]) {
      sc_set_error("%s: stream %d is listed twice (one chunk per stream and call)", what, stream_ids[i]);
      return SC_ERR_ARG;
    }
    seen[stream_ids[i]] = 1;
    const int s = stream_ids[i];
    const int out = (b->job[s].open ? 1 : 0) + (int)b->ahead[s].size();
    if (out >= depth) {
      sc_set_error(depth > 1 ? "%s: stream %d already has %d chunks outstanding (the queue depth; sc_poll reports them)"
                             : "%s: stream %d still has a chunk outstanding (sc_poll reports it)", what, s, out);
      return SC_ERR_ARG;
    }
    if (out > 0) {   // queueing behind an outstanding chunk
      const Job &last = b->ahead[s].empty() ? b->job[s] : b->ahead[s].back();
      if (last.fin && !last.fault) {
        sc_set_error("%s: stream %d: nothing can be queued behind a final chunk (sc_poll reports it first)", what, s);
        return SC_ERR_ARG;
      }
    }
  }
  return SC_OK;
}

}  // namespace

// =====================================================================================================
extern "C" int sc_engine_create(const sc_config *cfg, const sc_named_tensor *tensors, int n_tensors, int device,
                                sc_engine **out) {
  SC_CHECK_ARG(cfg && tensors && out && n_tensors > 0, "null");
  SC_API_BEGIN
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
    l.w1_s = e->f(p + "w1_s", false);   // optional fp16 hi | lo split: fp32-grade fused FFN on the fp16 matrix pipe
    l.w2_s = e->f(p + "w2_s", false);
    l.wqkv_h = e->f(p + "wqkv_h", false);   // optional fp16 copies: fp16 MFMA inputs in the attention projections
    l.wo_h = e->f(p + "wo_h", false);
    l.wqkv_s = e->f(p + "wqkv_s", false);   // optional fp16 hi | lo split of the attention projections
    l.wo_s = e->f(p + "wo_s", false);
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
    DEC_OPT(w1_s) DEC_OPT(w2_s) DEC_OPT(wqkv_pph) DEC_OPT(wq_pph) DEC_OPT(wo_pph) DEC_OPT(wo2_pph)
#undef DEC
#undef DEC_OPT
    e->wkv[i] = e->f(p + "wkv");
    e->bkv[i] = e->f(p + "bkv");
    if (!e->wkv[i] || !e->bkv[i]) { delete e; return SC_ERR_ARG; }
  }
  {
    // one weight matrix for the cross-attention K|V rows of all decoder layers: project_rows runs ONE product per group
    // instead of one per layer (14 launches of ~600 x 512 x 256 at 6 TFLOP/s in round 4).  A copy: 2 MB per layer at d = 256
    const size_t d = cfg->d_model, per = 2 * d * d, Ld = cfg->dec_layers;
    if (Ld > 1 && hipMalloc(&e->wkv_all, Ld * per * sizeof(float)) == hipSuccess &&
        hipMalloc(&e->bkv_all, Ld * 2 * d * sizeof(float)) == hipSuccess) {
      bool ok = true;
      for (size_t i = 0; i < Ld && ok; ++i)
        ok = hipMemcpy(e->wkv_all + i * per, e->wkv[i], per * sizeof(float), hipMemcpyDeviceToDevice) == hipSuccess &&
             hipMemcpy(e->bkv_all + i * 2 * d, e->bkv[i], 2 * d * sizeof(float), hipMemcpyDeviceToDevice) == hipSuccess;
      if (!ok) { (void)hipFree(e->wkv_all); (void)hipFree(e->bkv_all); e->wkv_all = e->bkv_all = nullptr; }
    } else {
      (void)hipGetLastError();
      if (e->wkv_all) (void)hipFree(e->wkv_all);
      e->wkv_all = e->bkv_all = nullptr;
    }
  }
  *out = e;
  return SC_OK;
  SC_API_END
}

// packed model file (speechcatcher_amd.weights.PackedWeights.save_packed): "SCPK1\0\0\0", int32 sizeof(sc_config),
// sc_config, int32 n, then n x { int32 name_len, name, int32 dtype, int64 numel, data }
extern "C" int sc_engine_load(const char *path, int device, sc_engine **out) {
  SC_CHECK_ARG(path && out, "null");
  SC_API_BEGIN
  FILE *fp = fopen(path, "rb");
  if (!fp) { sc_set_error("sc_engine_load: cannot open %s", path); return SC_ERR_ARG; }
  std::vector<void *> ptrs;
  auto bad = [&](const char *m) {
    sc_set_error("sc_engine_load: %s (%s)", m, path);
    fclose(fp);
    for (void *p : ptrs) (void)hipFree(p);   // tensors uploaded so far
    return SC_ERR_ARG;
  };
  char magic[8];
  int32_t csz = 0, n = 0;
  sc_config cfg{};
  if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "SCPK1\0\0\0", 8) != 0) return bad("not a packed model file");
  if (fread(&csz, 4, 1, fp) != 1 || csz != (int32_t)sizeof(sc_config)) return bad("config size mismatch");
  if (fread(&cfg, sizeof cfg, 1, fp) != 1 || fread(&n, 4, 1, fp) != 1 || n <= 0 || n > 100000) return bad("truncated header");
  if (hipSetDevice(device) != hipSuccess) return bad("hipSetDevice failed");
  std::vector<std::string> names(n);
  std::vector<sc_named_tensor> tens(n);
  std::vector<char> buf;
  for (int i = 0; i < n; ++i) {
    int32_t nl = 0, dt = 0;
    int64_t numel = 0;
    if (fread(&nl, 4, 1, fp) != 1 || nl <= 0 || nl > 256) return bad("bad tensor name");
    names[i].resize(nl);
    if (fread(&names[i][0], 1, nl, fp) != (size_t)nl || fread(&dt, 4, 1, fp) != 1 || fread(&numel, 8, 1, fp) != 1)
      return bad("truncated tensor header");
    if (dt < 0 || dt > 2 || numel < 0 || numel > ((int64_t)1 << 33)) return bad("bad tensor dtype / size");
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
  SC_API_END
}

extern "C" void sc_engine_destroy(sc_engine *e) { delete e; }

extern "C" int sc_engine_config(const sc_engine *e, sc_config *out) {
  SC_CHECK_ARG(e && out, "null");
  *out = e->cfg;
  return SC_OK;
}

extern "C" int sc_streams_create(sc_engine *e, const sc_stream_options *o, sc_streams **out) {
  SC_CHECK_ARG(e && o && out, "null");
  SC_API_BEGIN
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
  // (SC_PRIO_DEC / SC_PRIO_ENC: test hooks for the priority A/B of profiles/r05_ab_priority.txt)
  if (hipStreamCreateWithPriority(&b->stream, hipStreamNonBlocking, sc_hook("SC_PRIO_DEC") ? atoi(sc_hook("SC_PRIO_DEC")) : -1) != hipSuccess) {
    sc_set_error("sc_streams_create: hipStreamCreate failed");
    delete b;
    return SC_ERR_LAUNCH;
  }
  if (hipStreamCreateWithPriority(&b->stream_rb, hipStreamNonBlocking, -1) != hipSuccess) {
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
  // job-table arena: one slot per encoder group in flight; the largest tables are the conv2 row gather
  // (frames x conv_freq2 per stream) and the per-frame row lists
  b->slot_cap = (((size_t)S * ((size_t)b->max_t2 * (c.conv_freq2 + 12) + 512) + 8192) + 63) & ~size_t(63);
  A(b->arena_dev, b->slot_cap * N_ARENA);
  b->bs_cap = (size_t)S * 8 + (size_t)S * c.block_size;
  A(b->bs_dev, b->bs_cap * N_RING);
  b->stage_cap = (size_t)S * std::max<size_t>((size_t)b->max_chunk + 16 * c.hop_length, (size_t)b->max_feat_new * c.n_mels);
  A(b->stage_dev, b->stage_cap * N_STAGE);
  b->pack_cap = n * (2 * (size_t)b->LCAP + 8);
  A(b->pack_dev, b->pack_cap);
  A(b->pjobs_dev, n * 4);
  A(b->cjobs_dev, (size_t)N_STAGE * S * 3);
  sc_search &sb = b->sb;
  sb.S = S; sb.W = W; sb.K = K; sb.V = V; sb.d = d; sb.H = c.dec_heads; sb.F = F; sb.n_layers = c.dec_layers;
  sb.TCAP = b->TCAP; sb.LCAP = b->LCAP; sb.blank = c.blank_id; sb.eos = c.eos_id; sb.sos = c.sos_id;
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
  // self-attention K|V: a pool of rows per stream and layer (scasr.h: sc_search.skv).  A beam's hypotheses share almost
  // all of their history (1.05-1.2 distinct rows per token position measured), but nothing bounds that: two or three
  // alternatives that survive a whole utterance need 2-3 rows per position.  Default: ONE ROW PER (position, hypothesis)
  // - a stream can then never run out of rows before max_tokens - unless that takes more than a quarter of the device's
  // free memory (thousands of streams): then as many rows as that budget holds, never fewer than 1.5 per position + 4W.
  // Rows are handed out lowest-first, so a larger pool costs address space, not traffic.
  {
    const long full = (long)b->LCAP * W;
    long rows = o->kv_pool_rows;
    if (rows <= 0) {
      const long floor_rows = (long)b->LCAP + b->LCAP / 2 + 4 * W;
      size_t free_b = 0, total_b = 0;
      rows = floor_rows;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t per_row = (size_t)S * c.dec_layers * 2 * d * sizeof(float) / kvdiv;
        // ... and never more than SC_KV_POOL_DEFAULT_MAX_GIB by default, whatever happens to be free (a size that
        // follows "free memory" alone depends on allocation order and on who else uses the device; the chosen row count is
        // reported by sc_streams_kv_rows and in the bench line)
        const size_t budget = std::min<size_t>(free_b / 4, (size_t)SC_KV_POOL_DEFAULT_MAX_GIB << 30);
        rows = std::max<long>(floor_rows, (long)std::min<size_t>((size_t)full, budget / per_row));
      } else {
        (void)hipGetLastError();
      }
    }
    rows = std::max<long>(std::min(rows, full), 2 * W);
    if (rows > 65536) {
      if (o->kv_pool_rows > 0) {
        sc_set_error("sc_streams_create: the self-attention K|V pool is limited to 65536 rows per stream (max_tokens / kv_pool_rows)");
        delete b;
        return SC_ERR_ARG;
      }
      rows = 65536;
    }
    sb.kv_rows = (int32_t)rows;
  }
  A(sb.skv, (size_t)S * c.dec_layers * sb.kv_rows * 2 * d / kvdiv + 4);
  A(sb.yseq, (size_t)2 * n * b->LCAP);
  A(sb.xpos, (size_t)2 * n * b->LCAP);
  A(sb.anc, (size_t)2 * S * b->LCAP * W);
  A(sb.score, 2 * n); A(sb.sc_dec, 2 * n); A(sb.sc_ctc, 2 * n);
  A(sb.ctc_r, (size_t)2 * S * b->TCAP * 2 * W);
  A(sb.ctc_rs, (size_t)2 * S * b->TCAP * W);
  A(sb.ctc_s, 2 * n);
  A(sb.ctc_rnew, (size_t)S * ((b->TCAP + 15) / 16) * 2 * W * K);   // checkpoints every 16 frames (search.hip: SC_CTC_CK)
  A(sb.dx, n * d); A(sb.dxn, n * d); A(sb.dqkv, n * 3 * d); A(sb.datt, n * d); A(sb.dq, n * d); A(sb.dffh, n * F);
  A(sb.logits, n * V); A(sb.logp, n * V);
  A(sb.pre_ids, n * K); A(sb.psi, n * K); A(sb.psi_eos, n);
  A(sb.cand_score, n * W); A(sb.cand_tok, n * W); A(sb.cand_ctc, n * W);
  A(sb.sel, n * 2);
  A(b->stat_rows, 3);
  sb.stat_rows = b->stat_rows;
  if (sc_dec_layer_fused_supported(d, c.dec_heads, W, F) && V % d == 0 && e->dec[0].wqkv_pp && e->f("out_w_q", false)) {
    A(sb.ph1, n * c.dec_heads * d);
    A(sb.ph2, n * c.dec_heads * d);
    A(sb.ffn_part, (size_t)(F / 128) * n * d);
    sb.max_ffn_part = F / 128;
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
  // fp16 decoder mode (BASELINE configs[4]): the engine carries the fp16 copies AND the K|V caches are fp16
  sb.out_w_qh = nullptr;
  sb.act_half = 0;
  if (sb.kv_half && e->f("out_w_qh", false) && !e->dec.empty() && e->dec[0].wqkv_pph && e->dec[0].wq_pph && e->dec[0].wo_pph &&
      e->dec[0].wo2_pph && sb.ph1) {
    sb.out_w_qh = e->f("out_w_qh", false);
    // fp16 weight fragments in the layer kernels + fp16 partial products.  NOT the output layer (bit 4): its result goes
    // straight into the scores - measured on 256 streams x 7 chunks against the fp32 engine (tools/fp16_mode_stats.py,
    // profiles/r04_fp16_mode_stats.txt): bits 1 | 2 move the best hypothesis of 1 stream, the fp16 output layer alone of 18
    sb.act_half = 1 | 2;
    if (const char *ah = sc_hook("SC_ACT_HALF")) sb.act_half = atoi(ah) & 7;   // tools/fp16_mode_stats.py: any subset
  }
  sb.layers = e->dec.data();
  (void)sc_set_stream_workspace(b->stream, b->ws, (size_t)128 << 20);
  b->es = b->stream;
  {
    // The encoder stage starts when at most 7 % of the streams are still in the step loop (small batches: at once).
    // Measured at 128 streams (docs/profiles_r1-r3/r02_encoder_overlap_sweep.txt): serial 33.4 ms per chunk step; started with
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
        er = hipStreamCreateWithPriority(&b->stream_enc, hipStreamNonBlocking, sc_hook("SC_PRIO_ENC") ? atoi(sc_hook("SC_PRIO_ENC")) : 0);
      }
      if (er == hipSuccess && hipMalloc(&b->ws_enc, (size_t)128 << 20) == hipSuccess) {
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
  bool ok = b->halloc(&b->ctrlmap_host, (size_t)S * 8 + n) == SC_OK && b->halloc(&b->flags_host, (size_t)S * 2) == SC_OK &&
            b->halloc(&b->rm_host[0], n) == SC_OK && b->halloc(&b->rm_host[1], n) == SC_OK &&
            b->halloc(&b->arena_host, b->slot_cap * N_ARENA) == SC_OK && b->halloc(&b->bs_host, b->bs_cap * N_RING) == SC_OK &&
            b->halloc(&b->stage_host, b->stage_cap * N_STAGE) == SC_OK && b->halloc(&b->pack_host, b->pack_cap) == SC_OK &&
            b->halloc(&b->pjobs_host, n * 4) == SC_OK && b->halloc(&b->cjobs_host, (size_t)N_STAGE * S * 3) == SC_OK;
  for (int i = 0; i < 2 && ok; ++i) ok = hipEventCreateWithFlags(&b->ev_iter[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < N_ARENA && ok; ++i) ok = hipEventCreateWithFlags(&b->ev_group[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < N_STAGE && ok; ++i) ok = hipEventCreateWithFlags(&b->ev_stage[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
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
    sb.kvflags = b->ring_dev + S;   // second half of the host-mapped flag array: "K|V pool exhausted" per stream
  }
  if (const char *sp = sc_hook("SC_SCAN_SPLIT_MIN")) b->scan_split_min = atoi(sp);   // tests: 0 = never, small = always
  if (const char *sp = sc_hook("SC_SCAN_SPLIT_STREAMS")) b->scan_split_streams = atoi(sp);   // tools: bucket limit of the T-parallel scan
  memset(b->ctrlmap_host, 0, cm);
  for (size_t i = 0; i < n; ++i) b->rm_host[0][i] = b->rm_host[1][i] = (int32_t)i;
  (void)hipMemcpy(b->ctrlmap, b->ctrlmap_host, (size_t)S * 8 * sizeof(int32_t), hipMemcpyHostToDevice);
  (void)hipMemcpy(b->ctrlmap + S * 8, b->rm_host[0], n * sizeof(int32_t), hipMemcpyHostToDevice);
  b->rowmap_key.resize(S);
  for (int s = 0; s < S; ++s) b->rowmap_key[s] = s;
  b->enc_batch_min = std::max(1, S / 2);
  // (the projections' split copies, i.e. proj_dtype = "split16": the feed-forward's alone leave every projection fp32)
  b->gemm_flags = (!b->eng->enc.empty() && b->eng->enc[0].wqkv_s) ? SC_GEMM_SPLIT16 : 0;
  b->row_bucket = std::max(1, S / 32);   // 32 compaction buckets (graphs): 16 -> 32 measured +1 % at 128 streams, 64 nothing more
  if (const char *rb = sc_hook("SC_ROW_BUCKETS")) b->row_bucket = std::max(1, S / std::max(1, atoi(rb)));   // tools: sweep
  b->n_rows_step = S * W;
  b->st.assign(S, St());
  b->run.assign(S, Run());
  b->bq.assign(S, std::deque<Blk>());
  b->job.assign(S, Job());
  b->ahead.assign(S, std::deque<Job>());
  b->job_seq.assign(S, 0);
  b->snap.assign(S, Snap());
  b->done_at.assign(S, 0);
  b->fault_msg.assign(S, std::string());
  b->enc_gen.assign(S, 0);
  for (int s = 0; s < S; ++s) init_hyp(b, s);
  *out = b;
  return SC_OK;
  SC_API_END
}

extern "C" void sc_streams_destroy(sc_streams *b) {
  if (b) {
    (void)hipStreamSynchronize(b->stream);
    if (b->stream_enc) (void)hipStreamSynchronize(b->stream_enc);
    delete b;
  }
}

extern "C" int sc_reset(sc_streams *b, int stream) {
  SC_CHECK_ARG(b && stream >= 0 && stream < b->S, "stream out of range");
  SC_CHECK_ARG(!b->job[stream].open, "the stream has a chunk outstanding (sc_poll reports it first)");
  reset_stream(b, stream);
  return SC_OK;
}

extern "C" void *sc_streams_hip_stream(sc_streams *b) { return b ? (void *)b->stream : nullptr; }

extern "C" float *sc_streams_pcm(sc_streams *b, long *capacity) {
  if (!b) return nullptr;
  if (capacity) *capacity = b->PCAP;
  return b->pcm;
}

namespace {

// report a complete chunk: status, reset of a failed stream, job closed
int report_chunk(sc_streams *b, int s) {
  Job &j = b->job[s];
  const int status = j.fault ? j.fault : j.has_out;
  if (j.fault && !j.dropped) reset_stream(b, s);   // (fault_msg[s] keeps the message: sc_stream_last_error)
  if (b->snap[s].valid && b->snap[s].seq == j.seq) b->snap[s].reported = true;   // handed out: free at the next sc_poll
  b->done_at[s] = 0;
  if (b->ahead[s].empty()) j = Job();
  else {   // the next chunk of the stream is now its oldest
    j = b->ahead[s].front();
    b->ahead[s].pop_front();
  }
  b->n_open--;
  return status;
}

// tick until every listed stream's chunk is decoded (sc_push, sc_push_features)
int run_to_completion(sc_streams *b, const std::vector<int> &streams) {
  while (true) {
    bool all = true;
    for (int s : streams) all = all && (b->job[s].fault != 0 || chunk_decoded(b, s));
    if (all) return SC_OK;
    bool progress = false;
    RC_TRY(engine_tick(b, &progress));
    if (!progress) {
      sc_set_error("decode schedule stalled (internal error)");
      return SC_ERR_LAUNCH;
    }
  }
}

int push_impl(sc_streams *b, const int *stream_ids, const float *const *host, const int *counts, const uint8_t *is_final,
              int n, int *status, bool features, const char *what) {
  if (b->poisoned) {
    sc_set_error("%s: the handle was left inconsistent by an earlier failed call; destroy it", what);
    return SC_ERR_LAUNCH;
  }
  HIP_TRY(hipSetDevice(b->eng->device));
  RC_TRY(check_chunks(b, stream_ids, counts, n, what));
  for (int i = 0; i < n; ++i) b->snap[stream_ids[i]] = Snap();   // (queue depth > 1) a new call supersedes the copy of the last reply
  std::vector<Chunk> chunks;
  std::vector<int> streams, has_out(n, 0);
  for (int i = 0; i < n; ++i) {
    chunks.push_back({stream_ids[i], host ? host[i] : nullptr, counts[i], is_final[i] != 0, i});
    streams.push_back(stream_ids[i]);
    if (status) status[i] = 0;
  }
  if (n == 0) return SC_OK;
  int rc = admit(b, chunks, features, /*launch_now=*/false, &has_out);
  if (rc == SC_OK) rc = run_to_completion(b, streams);
  if (rc == SC_OK) rc = launch_pending_groups(b);   // (if no decode iteration got to it)
  if (rc != SC_OK) return poison(b, rc);
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (b->stream_enc) HIP_TRY(hipStreamSynchronize(b->stream_enc));
  RC_TRY(retire_groups(b));
  std::string msgs;
  for (int i = 0; i < n; ++i) {
    const int s = stream_ids[i];
    const bool failed = b->job[s].fault != 0;
    const int st = report_chunk(b, s);
    if (status) status[i] = st;
    if (failed) msgs += (msgs.empty() ? "" : "; ") + std::string("stream ") + std::to_string(s) + ": " + b->fault_msg[s];
  }
  if (!msgs.empty()) sc_set_error("%s", msgs.c_str());
  return SC_OK;
}

}  // namespace

// One chunk step (Speech2TextStreaming.__call__ for raw audio, for the listed streams at once).
// pcm[i] == NULL: the samples are already resident in the stream's device PCM buffer (sc_streams_pcm) behind
// what it has received so far - only the count is taken.  status[i] (HOST out): 1 = the call produced output,
// 0 = the reference's early `return []` (speech2text_streaming.py:432-433), < 0 = this stream failed
// (SC_ERR_CAPACITY: a capacity limit; SC_ERR_INPUT: an input the reference itself raises on, A3) - it has been
// reset, sc_stream_last_error() holds its message (sc_last_error() all messages of the call), and every other stream
// of the call is decoded as if it had not been there.  Returns SC_OK unless the call as a whole failed.
extern "C" int sc_push(sc_streams *b, const int *stream_ids, const float *const *pcm, const int *n_samples,
                       const uint8_t *is_final, int n, int *status) {
  SC_CHECK_ARG(b && stream_ids && n_samples && is_final && n >= 0, "null");
  SC_API_BEGIN
  return push_impl(b, stream_ids, pcm, n_samples, is_final, n, status, false, "sc_push");
  SC_API_END
}

// 2-D (T, n_mels) already-normalised features instead of PCM (the reference's 2-D / 3-D input path,
// speech2text_streaming.py:438-449): feats[i] HOST [n_frames[i]][n_mels].  Same status convention as sc_push.
extern "C" int sc_push_features(sc_streams *b, const int *stream_ids, const float *const *feats, const int *n_frames,
                                const uint8_t *is_final, int n, int *status) {
  SC_CHECK_ARG(b && stream_ids && feats && n_frames && is_final && n >= 0, "null");
  for (int i = 0; i < n; ++i) SC_CHECK_ARG(feats[i] || n_frames[i] == 0, "null feature matrix");
  SC_API_BEGIN
  return push_impl(b, stream_ids, feats, n_frames, is_final, n, status, true, "sc_push_features");
  SC_API_END
}

// Continuous batching, part 1: hand the engine one chunk for each listed stream and return at once.  The chunks are
// copied (the caller may free them), planned and their encoder stage is issued as ONE group; decoding happens inside
// sc_poll.  A stream has at most one chunk outstanding: submit its next chunk after sc_poll has reported this one
// (the reference's session loop is call -> result -> next call, speechcatcher_server.py:359-397).
extern "C" int sc_submit(sc_streams *b, const int *stream_ids, const float *const *pcm, const int *n_samples,
                         const uint8_t *is_final, int n) {
  SC_CHECK_ARG(b && stream_ids && n_samples && is_final && n >= 0, "null");
  SC_API_BEGIN
  if (b->poisoned) {
    sc_set_error("sc_submit: the handle was left inconsistent by an earlier failed call; destroy it");
    return SC_ERR_LAUNCH;
  }
  HIP_TRY(hipSetDevice(b->eng->device));
  RC_TRY(check_chunks(b, stream_ids, n_samples, n, "sc_submit", b->queue_depth));
  if (n == 0) return SC_OK;
  std::vector<Chunk> chunks;
  for (int i = 0; i < n; ++i) {
    const int s = stream_ids[i];
    int failed = b->job[s].open ? b->job[s].fault : 0;
    for (const Job &j : b->ahead[s])
      if (j.fault) failed = j.fault;
    if (failed) {   // queued behind a chunk that has failed (the host cannot know yet): it fails with it, reported in order
      Job j;
      j.open = true;
      j.fault = failed;
      j.dropped = true;
      j.seq = ++b->job_seq[s];
      b->ahead[s].push_back(j);
      b->n_open++;
      continue;
    }
    chunks.push_back({s, pcm ? pcm[i] : nullptr, n_samples[i], is_final[i] != 0, i});
  }
  if (chunks.empty()) return SC_OK;
  const int rc = admit(b, chunks, false, /*launch_now=*/true, nullptr);
  return rc == SC_OK ? SC_OK : poison(b, rc);
  SC_API_END
}

// Continuous batching, part 2: run the engine (decode ticks over every stream that is inside a block) until at least
// min_done outstanding chunks are complete - or none is outstanding - and report up to max_done of them:
// done_ids[i] = stream, status[i] as in sc_push.  Returns the number reported (0: nothing outstanding), < 0 on error.
// The hypotheses of a reported stream are those of ITS call (sc_get_hyps / sc_get_hyps_batch) until its next chunk
// is submitted.  min_done <= 0: report what is complete now without running a tick.
extern "C" int sc_poll(sc_streams *b, int min_done, int max_done, int *done_ids, int *status) {
  SC_CHECK_ARG(b && done_ids && max_done > 0, "bad arguments");
  SC_API_BEGIN
  if (b->poisoned) {
    sc_set_error("sc_poll: the handle was left inconsistent by an earlier failed call; destroy it");
    return SC_ERR_LAUNCH;
  }
  HIP_TRY(hipSetDevice(b->eng->device));
  min_done = std::min(std::min(min_done, max_done), b->n_open);
  for (Snap &sn : b->snap)   // the hypotheses handed out by the previous call have been read: their copies are free
    if (sn.valid && sn.reported) sn.valid = false;
  int stalled = 0;
  while (true) {
    int rc = tick_collect(b);   // the step enqueued by the previous call (or iteration)
    if (rc == SC_OK) rc = retire_groups(b);
    if (rc == SC_OK) rc = snapshot_completed(b);
    if (rc != SC_OK) return poison(b, rc);
    int n_complete = 0;
    for (int s = 0; s < b->S; ++s)
      if (chunk_complete(b, s)) {
        ++n_complete;
        if (!b->done_at[s]) b->done_at[s] = ++b->done_counter;   // (streams that complete in one tick: by index)
      }
    const bool enough = n_complete >= min_done || b->n_open == 0;
    // the next step is enqueued BEFORE the replies go back to the caller: the device decodes the streams that are still
    // inside their blocks while the host reads the finished streams' hypotheses and admits their next chunks
    bool progress = false;
    if (b->n_open > n_complete || !enough) {
      do {   // (a tick that only started / closed blocks enqueues nothing: go on until a step is in flight or all is idle)
        rc = tick_issue(b, &progress);
        if (rc != SC_OK) return poison(b, rc);
      } while (progress && !b->inflight);
    }
    if (enough) break;
    if (b->inflight || progress) stalled = 0;
    else if (n_complete > 0 && b->queue_depth > 1) break;   // the rest waits for these replies to be handed out (one snapshot per stream)
    else if (++stalled > 1) {   // (once: blocks that were closed without a step change the count above)
      sc_set_error("sc_poll: schedule stalled (internal error)");
      return poison(b, SC_ERR_LAUNCH);
    }
  }
  // oldest completion first: with a small max_done no stream waits behind lower-numbered ones that completed later
  std::vector<std::pair<long, int>> ready;
  for (int s = 0; s < b->S; ++s)
    if (chunk_complete(b, s)) ready.push_back({b->done_at[s] ? b->done_at[s] : ++b->done_counter, s});
  std::sort(ready.begin(), ready.end());
  int n = 0;
  for (size_t i = 0; i < ready.size() && n < max_done; ++i, ++n) {
    const int s = ready[i].second;
    done_ids[n] = s;
    const int st = report_chunk(b, s);
    if (status) status[n] = st;
  }
  return n;
  SC_API_END
}

extern "C" int sc_streams_outstanding(const sc_streams *b) { return b ? b->n_open : 0; }

// Chunks a stream may have outstanding at a time (default 1: the reference's call -> reply -> next call).  depth > 1:
// sc_submit accepts the NEXT chunk(s) of a stream while an earlier one is still being decoded - a host that has the
// audio already (a file, a backlog) lets the frontend + encoder of chunk k+1 run beside the decoding of chunk k, so
// that the stream does not idle between its reply and its next call.  Chunks of a stream are processed and reported
// in order, one per sc_poll call; per chunk the results are those of the one-at-a-time protocol.  The hypotheses of a
// reported chunk are a copy taken when it completed: sc_get_hyps / sc_get_hyps_batch return them until the NEXT
// sc_poll call.  Nothing can be queued behind a final chunk; a failure fails the chunks queued behind it.
extern "C" int sc_streams_set_queue_depth(sc_streams *b, int depth) {
  SC_CHECK_ARG(b && depth >= 1 && depth <= 8, "bad arguments");
  SC_API_BEGIN
  SC_CHECK_ARG(b->n_open == 0, "chunks are outstanding");
  HIP_TRY(hipSetDevice(b->eng->device));
  if (depth > 1 && !b->snap_yseq) {
    RC_TRY(b->alloc(&b->snap_yseq, (size_t)b->S * b->W * b->LCAP));
    RC_TRY(b->alloc(&b->snap_xpos, (size_t)b->S * b->W * b->LCAP));
    RC_TRY(b->alloc(&b->snap_score, (size_t)b->S * b->W * 3));
    HIP_TRY(hipEventCreateWithFlags(&b->ev_snap, hipEventDisableTiming));
  }
  for (Snap &sn : b->snap) sn = Snap();
  b->queue_depth = depth;
  return SC_OK;
  SC_API_END
}

// sc_submit policy: the frontend + encoder stages of successive admissions are merged and issued as ONE group when it
// holds min_streams streams (default: half of the streams), or as soon as a queued decode block needs its frames.
// 1: every admission is issued at once.  Results do not depend on it.
extern "C" int sc_streams_set_encoder_batch(sc_streams *b, int min_streams) {
  SC_CHECK_ARG(b && min_streams >= 1, "bad arguments");
  b->enc_batch_min = min_streams;
  return SC_OK;
}

// message of the last failure of this stream (status < 0 from sc_push / sc_poll), "" if none; valid until the
// stream fails again or the handle is destroyed
extern "C" const char *sc_stream_last_error(const sc_streams *b, int stream) {
  if (!b || stream < 0 || stream >= b->S) return "";
  return b->fault_msg[stream].c_str();
}

// live hypotheses of a stream, best first: ids / xpos [nbest][max_len] (row-major, caller-allocated), lens[nbest],
// scores / score_dec / score_ctc [nbest] (any of them may be NULL).  Returns the number of hypotheses written
// (<= nbest), or a negative error.
extern "C" int sc_get_hyps(sc_streams *b, int stream, int nbest, int max_len, int32_t *ids, int32_t *xpos, int *lens,
                           double *scores, double *score_dec, double *score_ctc) {
  SC_CHECK_ARG(b && stream >= 0 && stream < b->S, "bad arguments");
  int nh = 0;
  const int rc = sc_get_hyps_batch(b, &stream, 1, nbest, max_len, ids, xpos, lens, &nh, scores, score_dec, score_ctc);
  return rc == SC_OK ? nh : rc;
}

// The same for n streams with ONE pack launch and ONE device-to-host copy (into pinned memory): ids / xpos
// [n][nbest][max_len], lens / scores / score_dec / score_ctc [n][nbest], n_hyps [n] (any output may be NULL except
// n_hyps).  What Speech2TextStreaming.__call__ returns for a batch of sessions (speech2text_streaming.py:466-539:
// every running hypothesis with yseq, xpos, score, scores{decoder, ctc}).
extern "C" int sc_get_hyps_batch(sc_streams *b, const int *stream_ids, int n, int nbest, int max_len, int32_t *ids,
                                 int32_t *xpos, int *lens, int *n_hyps, double *scores, double *score_dec,
                                 double *score_ctc) {
  SC_CHECK_ARG(b && stream_ids && n_hyps && n >= 0 && nbest >= 0 && max_len >= 0, "bad arguments");
  SC_API_BEGIN
  HIP_TRY(hipSetDevice(b->eng->device));
  std::vector<int32_t> jobs;
  size_t off = 0;
  bool from_snapshot = false;
  // the job tables hold one entry per hypothesis row of the batch (S*W): every stream at most once
  SC_CHECK_ARG(n <= b->S, "more streams listed than the batch has");
  std::vector<char> listed(b->S, 0);
  for (int i = 0; i < n; ++i) {
    const int s = stream_ids[i];
    SC_CHECK_ARG(s >= 0 && s < b->S, "stream out of range");
    SC_CHECK_ARG(!listed[s], "a stream is listed twice");
    listed[s] = 1;
    const Snap &sn = b->snap[s];
    if (sn.valid) {   // queue depth > 1: the copy taken when the stream's last reported chunk completed
      const int nh = std::min(nbest, sn.nhyp);
      n_hyps[i] = nh;
      SC_CHECK_ARG(nh == 0 || (!ids && !xpos) || max_len >= sn.L, "max_len is smaller than the hypotheses");
      for (int h = 0; h < nh; ++h) {
        jobs.insert(jobs.end(), {(int32_t)((size_t)s * b->W + h), sn.L, (int32_t)off, 1});
        off += (size_t)((2 * sn.L + 1) & ~1) + 6;
      }
      from_snapshot = true;
      continue;
    }
    SC_CHECK_ARG(!b->run[s].inblk && b->bq[s].empty(), "stream is inside a decode block (its chunk has not been reported yet)");
    const St &st = b->st[s];
    const int nh = st.started ? std::min(nbest, st.nhyp) : 0;
    n_hyps[i] = nh;
    SC_CHECK_ARG(nh == 0 || (!ids && !xpos) || max_len >= st.L, "max_len is smaller than the hypotheses");
    for (int h = 0; h < nh; ++h) {
      jobs.insert(jobs.end(), {(int32_t)(((size_t)st.cur * b->S + s) * b->W + h), st.L, (int32_t)off, 0});
      off += (size_t)((2 * st.L + 1) & ~1) + 6;
    }
  }
  const int m = (int)jobs.size() / 4;
  if (m == 0) return SC_OK;
  SC_CHECK_ARG(off <= b->pack_cap && (size_t)m <= (size_t)b->S * b->W, "hypotheses exceed the read-back buffer");
  memcpy(b->pjobs_host, jobs.data(), jobs.size() * sizeof(int32_t));
  // on the read-back stream: a decode step of OTHER streams may be in flight on the batch's stream (sc_poll); the
  // listed streams' last steps have been collected (host-synchronised) by then
  HIP_TRY(hipMemcpyAsync(b->pjobs_dev, b->pjobs_host, jobs.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->stream_rb));
  if (from_snapshot) HIP_TRY(hipStreamWaitEvent(b->stream_rb, b->ev_snap, 0));   // the copies are written on the batch's stream
  pack_hyps_kernel<<<m, 128, 0, b->stream_rb>>>(b->sb, b->pjobs_dev, b->pack_dev, b->snap_yseq, b->snap_xpos, b->snap_score);
  HIP_TRY(hipMemcpyAsync(b->pack_host, b->pack_dev, off * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream_rb));
  HIP_TRY(hipStreamSynchronize(b->stream_rb));
  int k = 0;
  for (int i = 0; i < n; ++i)
    for (int h = 0; h < n_hyps[i]; ++h, ++k) {
      const int L = jobs[k * 4 + 1];
      const int32_t *src = b->pack_host + jobs[k * 4 + 2];
      const size_t o = (size_t)i * nbest + h;
      if (ids) memcpy(ids + o * max_len, src, (size_t)L * 4);
      if (xpos) memcpy(xpos + o * max_len, src + L, (size_t)L * 4);
      if (lens) lens[o] = L;
      const double *sc = (const double *)(src + ((2 * L + 1) & ~1));
      if (scores) scores[o] = sc[0];
      if (score_dec) score_dec[o] = sc[1];
      if (score_ctc) score_ctc[o] = sc[2];
    }
  return SC_OK;
  SC_API_END
}

// host <-> device copies of a stream's PCM ring and encoder output (tests, bench preload, the drop-in class's
// frontend_states / encoder_buffer views)
extern "C" int sc_streams_write_pcm(sc_streams *b, int stream, long offset, const float *host, long n) {
  SC_CHECK_ARG(b && host && stream >= 0 && stream < b->S && offset >= 0 && n >= 0 && offset + n <= b->PCAP, "out of range");
  HIP_TRY(hipMemcpy(b->pcm + (long)stream * b->PCAP + offset, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  return SC_OK;
}
// (reads of encoder-side buffers first settle the encoder groups that sc_submit may still hold back or have in flight)
static int settle_encoder(sc_streams *b) {
  RC_TRY(tick_collect(b));
  RC_TRY(launch_pending_groups(b));
  if (b->stream_enc) HIP_TRY(hipStreamSynchronize(b->stream_enc));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return retire_groups(b);
}
extern "C" long sc_streams_read_pcm_buffer(sc_streams *b, int stream, float *host, long max_n) {
  if (!b || stream < 0 || stream >= b->S) return SC_ERR_ARG;
  if (settle_encoder(b) != SC_OK) return SC_ERR_LAUNCH;
  const St &st = b->st[stream];
  const long n = std::min<long>(st.pcm_end - st.pcm_start, max_n);
  if (host && n > 0 &&
      hipMemcpy(host, b->pcm + (long)stream * b->PCAP + st.pcm_start, (size_t)n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return SC_ERR_LAUNCH;
  return n;
}
extern "C" int sc_streams_read_enc(sc_streams *b, int stream, float *host, int max_frames) {
  if (!b || stream < 0 || stream >= b->S) return SC_ERR_ARG;
  if (settle_encoder(b) != SC_OK) return SC_ERR_LAUNCH;
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
extern "C" int sc_streams_kv_rows(const sc_streams *b) { return b ? b->sb.kv_rows : 0; }

extern "C" long sc_streams_take_xattn_rows(sc_streams *b) {
  if (!b) return 0;
  const long v = b->xattn_rows[0] + b->xattn_rows[1] + b->xattn_rows[2];
  b->xattn_rows[0] = b->xattn_rows[1] = b->xattn_rows[2] = 0;
  return v;
}
extern "C" int sc_streams_take_xattn_rows_by_kernel(sc_streams *b, long *rows) {
  SC_CHECK_ARG(b && rows, "null");
  for (int i = 0; i < 3; ++i) {
    rows[i] = b->xattn_rows[i];
    b->xattn_rows[i] = 0;
  }
  return SC_OK;
}

extern "C" int sc_streams_take_attn_counters(sc_streams *b, long *out) {
  SC_CHECK_ARG(b && out, "null");
  unsigned long long dr[3] = {0, 0, 0};
  HIP_TRY(hipSetDevice(b->eng->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(dr, b->stat_rows, sizeof dr, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(b->stat_rows, 0, sizeof dr));
  for (int i = 0; i < 3; ++i) {
    out[i] = b->xattn_rows[i];
    out[3 + i] = b->sattn_pos[i];
    out[6 + i] = (long)dr[i];
    b->xattn_rows[i] = b->sattn_pos[i] = 0;
  }
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

// encoder-layer hipGraphs captured so far (one per shape of an encoder group) and the host seconds spent capturing
extern "C" int sc_streams_capture_stats(const sc_streams *b, long *n_captures, double *seconds) {
  SC_CHECK_ARG(b && n_captures && seconds, "null");
  *n_captures = b->n_enc_captures;
  *seconds = b->t_enc_capture;
  return SC_OK;
}

extern "C" int sc_streams_stats(const sc_streams *b, long *enc_calls, long *dec_steps, long *dec_blocks) {
  SC_CHECK_ARG(b, "null");
  if (enc_calls) *enc_calls = b->enc_calls;
  if (dec_steps) *dec_steps = b->dec_steps;
  if (dec_blocks) *dec_blocks = b->dec_blocks;
  return SC_OK;
}

