// Shared helpers for the gfx950 kernels of libscasr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

#include "../../include/scasr.h"

#define SC_LOGZERO (-10000000000.0f)  // ctc_prefix_score_full.py:58

void sc_set_error(const char *fmt, ...);

// Test / measurement hooks.  The SC_* environment switches (kernel variants forced for parity tests, A/B runs and
// tuning sweeps) are consulted ONLY when the process was started with SC_TEST_HOOKS=1 (tests/conftest.py and the
// tools/ scripts set it); a production process never reads them, so nothing a server's environment contains can
// change which kernels a captured graph holds.  gemm.hip.
const char *sc_hook(const char *name);

#define SC_CHECK_ARG(cond, msg)      \
  do {                               \
    if (!(cond)) {                   \
      sc_set_error("%s: %s", __func__, msg); \
      return SC_ERR_ARG;             \
    }                                \
  } while (0)

#define SC_CHECK_LAUNCH()                                              \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      sc_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return SC_ERR_LAUNCH;                                            \
    }                                                                  \
  } while (0)

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// HIP events around one kernel launch (enabled by sc_prof_enable; gemm.hip)
struct ProfScope { bool on; hipEvent_t a, b; hipStream_t st; };
ProfScope sc_prof_begin(hipStream_t st);
void sc_prof_end(ProfScope &p, int kind, double flops, double bytes);

// search.hip: form of the decoder layers sc_decode_step picks for sb->n_rows (0 six-launch, 1 head-parallel, 2 stream-resident)
int sc_decode_step_form(const sc_search *sb);
// N / ncb products that share A in one launch, block j of the columns stored at C + j * cb_stride (gemm.hip)
int sc_gemm_colblocks(const float *A, const int32_t *a_rows, int lda, const float *W, const float *bias, float *C,
                      const int32_t *c_rows, int ldc, int M, int N, int K, int ncb, long cb_stride, int flags, void *stream);

// decoder_layer.hip: heads per workgroup of the head-parallel layer kernels for the bucket sb.n_rows (1, 2 or 4): the
// consumer of their partial products (the next layer kernel, sc_dec_layer_ffn) reduces sb.H / hpw of them per row
int sc_dec_layer_hpw(const sc_search &sb);
// decoder_stream.hip: 1 when this bucket's decoder layers run in the stream-resident form (one workgroup per stream, two
// launches per layer), 0: the head-parallel launches of decoder_layer.hip
int sc_dec_layer_stream_form(const sc_search &sb);
// decoder_layer.hip: 1 when the four-head form of this bucket sums the cross-attention's partial products in a launch of its own
// (sc_dec_layer_reduce_ln) and runs the feed-forward without prologue (sc_dec_layer_ffn_xn)
int sc_dec_layer_split_ffn(const sc_search &sb);

// decoder_panel.hip: reduce of the fused-FFN partial sums + LayerNorm + projection (sc_ffn_ln_proj)
int sc_launch_reduce_ln_proj(const float *part, int npart, int part_M, const float *b2, const float *Xin, float *Xout,
                             const int32_t *rows, int M, int D, const float *ln_g, const float *ln_b, float ln_eps,
                             float *XN, const float *Wq, const float *bq, float *Q, int N, hipStream_t st,
                             int by_row = 0, int half_mode = 0);

// gemm.hip: sc_ffn_ln / _h / _s (w_form 0 / 1 / 2) whose split-sum reduce also does the encoder's context hand-off of layer
// `layer` (sc_ctx_handoff fused into the reduce: one launch less per encoder layer).  blkinfo [nblk][4] comes from
// sc_ctx_blkinfo (encoder.hip); rows are the nblk * R rows of the block buffer, all in one slab of the workspace.
struct ScHandoff { const int32_t *blkinfo; int R; float *state; int layer; };
int sc_ffn_ln_handoff(const float *XN, int M, int D, int F, const void *W1, const float *b1, const void *W2, const float *b2,
                      float *X, int w_form, const ScHandoff &ho, void *stream);

// ---- device helpers (wave = 64 lanes on gfx950) ---------------------------
// DPP lane permutations (no LDS round trip, unlike the ds_bpermute behind __shfl)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define SC_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define SC_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define SC_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8 lanes
#define SC_DPP_ROR4 0x124        // rotate by 4 inside each 16 lanes
#define SC_DPP_ROR8 0x128        // rotate by 8 inside each 16 lanes
#define SC_DPP_ROW_MIRROR 0x140  // lane i <-> 15-i inside each 16 lanes

// sum over the LPR (4, 8 or 16) adjacent lanes of a row group; every lane gets the total
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_mov<SC_DPP_XOR1>(v);
  v += dpp_mov<SC_DPP_XOR2>(v);
  if (LPR >= 8) v += dpp_mov<SC_DPP_HALF_MIRROR>(v);
  if (LPR == 16) v += dpp_mov<SC_DPP_ROW_MIRROR>(v);
  return v;
}

// Wave-wide reductions (all 64 lanes must be active - every call site is wave-uniform): four DPP steps reduce each
// 16-lane row in registers, the four row results are combined through scalar registers.  The six dependent
// ds_bpermute steps of a __shfl_xor butterfly cost ~0.3 us per reduction - 2 us for the LayerNorm of a
// decoder layer's ten rows (tools/layer_phase_times.py).
__device__ __forceinline__ float sc_lane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<SC_DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<SC_DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<SC_DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_mov<SC_DPP_ROW_MIRROR>(v));
  return fmaxf(fmaxf(sc_lane(v, 0), sc_lane(v, 16)), fmaxf(sc_lane(v, 32), sc_lane(v, 48)));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<SC_DPP_XOR1>(v);
  v += dpp_mov<SC_DPP_XOR2>(v);
  v += dpp_mov<SC_DPP_HALF_MIRROR>(v);
  v += dpp_mov<SC_DPP_ROW_MIRROR>(v);
  return (sc_lane(v, 0) + sc_lane(v, 16)) + (sc_lane(v, 32) + sc_lane(v, 48));
}

// log(exp(a)+exp(b)) the way torch.logsumexp does it: max first.  One of the two exponentials is exp(0) == 1
// exactly, so only the other one is evaluated - with the bare hardware transcendentals (v_exp_f32 / v_log_f32:
// the argument of the log lies in [1, 2], the exponent is <= 0, so none of the range / denormal fix-ups that
// __expf / __logf carry under -fno-fast-math are needed; they made up most of the prefix scan's 170
// instructions per frame).
#define SC_LOG2E 1.4426950408889634f
#define SC_LN2 0.6931471805599453f
__device__ __forceinline__ float sc_exp_neg(float x) {   // exp(x) for x <= 0 (flushes below 2^-126)
  return __builtin_amdgcn_exp2f(x * SC_LOG2E);
}
__device__ __forceinline__ float sc_max_raw(float a, float b) {   // one v_max_f32 (fmaxf adds two canonicalising
  float m;                                                         // v_max x,x under IEEE mode; no NaNs occur here)
  asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
  return m;
}
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = sc_max_raw(a, b);
  return m + SC_LN2 * __builtin_amdgcn_logf(1.f + sc_exp_neg(-fabsf(a - b)));
}

// ---- canonical summation of partial sums (bit-reproducible serving) ---------------------------------------------------------
// Which tile shape a launch takes depends on how many rows are in flight, i.e. on who else is on the GPU.  A stream's result must
// not: every sum of partial results is therefore evaluated in ONE order, whatever the decomposition that produced the partials.
//   * split sums of the fused feed-forward (one per 128-wide hidden chunk, or per aligned PAIR of chunks added in that order
//     by the producer): a balanced binary tree over the aligned index pairs, evaluated in batches of 8 -
//     ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)), batches added in order (a tree again for <= 16 partials).  A missing entry
//     is +0 (x + 0 is exact).  The same routine reduces every split-K GEMM, whose slice count is a function of (N, K) alone.
//   * per-head partial products of the decoder's output projections: heads in aligned groups of four, each group summed in
//     head order, the groups in order - what the four-heads-per-workgroup kernels do in LDS is what the consumer of the
//     one-head-per-workgroup kernels does in registers.
__device__ __forceinline__ float4 sc_add4(const float4 &a, const float4 &b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 sc_tree4(const float4 &a, const float4 &b, const float4 &c, const float4 &d) {
  return sc_add4(sc_add4(a, b), sc_add4(c, d));
}
__device__ __forceinline__ float4 sc_tree8(const float4 (&v)[8]) {
  return sc_add4(sc_tree4(v[0], v[1], v[2], v[3]), sc_tree4(v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ float sc_tree8f(const float (&v)[8]) {
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
__device__ __forceinline__ float4 sc_seq4(const float4 &a, const float4 &b, const float4 &c, const float4 &d) {
  return sc_add4(sc_add4(sc_add4(a, b), c), d);
}

// Phase stamps for tools/layer_phase_times.py: compiled in only with -DSC_PHASE_DBG (make EXTRA=-DSC_PHASE_DBG, never
// in the shipped library).  Thread 0 of EVERY workgroup (linear id < SC_PHASE_WGS) stores the shader clock (s_memtime:
// one tick per shader cycle) at phase boundaries into a per-source-file table that sc_phase_debug_<file>() copies out;
// slots 14 / 15 of a workgroup hold the 100 MHz real-time counter at its first / last stamp (the shader clocks of the
// eight XCDs are not synchronised: workgroups are aligned on the real-time counter), slot 13 the launch's grid size.  A launch
// overwrites the stamps of the one before: what is read is the LAST launch of every kind.  (Round 4 stamped workgroup
// (0,0) only: the first workgroup of a launch is not representative of the 256 - phases of different workgroups do not
// line up.)
#ifdef SC_PHASE_DBG
#ifndef SC_PHASE_MIN_GRID
#define SC_PHASE_MIN_GRID 0   // -DSC_PHASE_MIN_GRID=200: only launches of at least that many workgroups leave stamps (full buckets)
#endif
#define SC_PHASE_WGS 512
#define SC_PHASE_RING 32   // the last 32 launches of every kind are kept (slot = launch number mod 32)
static __device__ long long sc_phase_stamps[4][SC_PHASE_RING][SC_PHASE_WGS][16];
// SC_STAMP_ON: expression in the kernel's scope (its argument struct carries the host's decision: sc_phase_debug_*_arm(n)
// lets only the next n launches of a kind leave stamps - e.g. the 14 layers of ONE decode step, whose last launch then stays)
#ifndef SC_STAMP_ON
#define SC_STAMP_ON 1
#endif
#define SC_STAMP_AT(k, i, rt)                                                                 \
  do {                                                                                        \
    const int wg__ = blockIdx.x + gridDim.x * blockIdx.y;                                     \
    if (threadIdx.x == 0 && (SC_STAMP_ON) && wg__ < SC_PHASE_WGS && (int)(gridDim.x * gridDim.y) >= SC_PHASE_MIN_GRID) { \
      long long *row__ = sc_phase_stamps[k][((SC_STAMP_ON) - 1) & (SC_PHASE_RING - 1)][wg__]; \
      row__[i] = (long long)__builtin_amdgcn_s_memtime();                                     \
      if ((rt) >= 0) row__[rt] = (long long)__builtin_amdgcn_s_memrealtime();                 \
      if ((i) == 0) {                                                                         \
        row__[13] = (long long)(gridDim.x * gridDim.y);                                       \
        row__[12] = (long long)(SC_STAMP_ON);                                                 \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define SC_STAMP(k, i) SC_STAMP_AT(k, i, ((i) == 0 ? 14 : -1))
#define SC_STAMP_END(k, i) SC_STAMP_AT(k, i, 15)
// ... by the thread for which `cond` holds (one per workgroup): the free row entries 9..11 - e.g. when the waves of a role are done
#define SC_STAMP_BY(k, i, cond)                                                               \
  do {                                                                                        \
    const int wg__ = blockIdx.x + gridDim.x * blockIdx.y;                                     \
    if ((cond) && (SC_STAMP_ON) && wg__ < SC_PHASE_WGS && (int)(gridDim.x * gridDim.y) >= SC_PHASE_MIN_GRID)           \
      sc_phase_stamps[k][((SC_STAMP_ON) - 1) & (SC_PHASE_RING - 1)][wg__][i] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
// ... behind a wait for everything the wave has in flight: the stamp then closes a memory round trip (changes the timing
// a little: the loads of the next stage are not under way yet)
#define SC_STAMP_WAIT(k, i)                                                                   \
  do {                                                                                        \
    __builtin_amdgcn_s_waitcnt(0);                                                            \
    SC_STAMP(k, i);                                                                           \
  } while (0)
#define SC_PHASE_GETTER(name)                                                                 \
  extern "C" int name(long long *out) {                                                       \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sc_phase_stamps), sizeof(long long) * 4 * SC_PHASE_RING * SC_PHASE_WGS * 16) == hipSuccess ? 0 : -1; \
  }                                                                                           \
  static int sc_phase_budget[4] = {-1, -1, -1, -1};   /* launches of a kind that still leave stamps (-1: all) */ \
  static int sc_phase_count[4] = {0, 0, 0, 0};        /* launch number of the kind (1-based; the ring slot) */   \
  static inline int sc_phase_take(int k) {            /* 0: no stamps, else the launch number */                 \
    if (sc_phase_budget[k] == 0) return 0;                                                    \
    if (sc_phase_budget[k] > 0) --sc_phase_budget[k];                                         \
    return ++sc_phase_count[k];                                                               \
  }                                                                                           \
  extern "C" int name##_arm(int n) {   /* clear the table; the next n launches of every kind leave stamps (-1: all) */ \
    static long long z[4 * SC_PHASE_RING * SC_PHASE_WGS * 16];                                \
    for (int k = 0; k < 4; ++k) { sc_phase_budget[k] = n; sc_phase_count[k] = 0; }            \
    return hipMemcpyToSymbol(HIP_SYMBOL(sc_phase_stamps), z, sizeof z) == hipSuccess ? 0 : -1; \
  }
#else
#define SC_STAMP(k, i) do {} while (0)
#define SC_STAMP_END(k, i) do {} while (0)
#define SC_STAMP_BY(k, i, cond) do {} while (0)
#define SC_STAMP_WAIT(k, i) do {} while (0)
#define SC_PHASE_GETTER(name)                                                                 \
  static inline int sc_phase_take(int) { return 0; }
#endif
