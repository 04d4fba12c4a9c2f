// Head-parallel decoder layer kernels of libscasr (gfx950): three launches per decoder
// layer instead of six.
//
//   A  sc_dec_layer_self   grid (stream, head)
//        x   = x_in + b2' + sum of the previous layer's feed-forward partial sums   (layer 0: embed*sqrt(d) + PE)
//        q|k|v(head) = LayerNorm1(x) . Wqkv[head slices]^T + b      (f32 MFMA 16x16x4, K split over the 4 waves)
//        ctx(head)   = self-attention over the ancestor-indexed K/V cache + the new row (single pass, online softmax)
//        ph1[row][head][:] = ctx(head) . Wo[:, head slice]^T         (split-K over heads of the output projection)
//   B  sc_dec_layer_cross  grid (stream, head)
//        x'  = x + bo + sum_heads ph1;   q(head) = LayerNorm2(x') . Wq[head slice]^T + bq
//        ctx(head) = cross-attention over the stream's shared encoder K|V;   ph2 = ctx(head) . Wo2[:, head slice]^T
//   C  sc_dec_layer_ffn    (gemm.hip: ffn_fused_kernel<.., PRO>)  x'' = x' + bo2 + sum_heads ph2, LayerNorm3, FFN partials
//
// reference semantics: speechcatcher/model/decoder/decoder_layer.py:80-132 (pre-LN layer with caches),
// model/attention/multi_head_attention.py:63-133, transformer_decoder.py:231 (embedding).
//
// Why this shape on MI355X: the decode step is a chain of ~90 dependent launches of 7-11 us each when few
// streams are active (docs/profiles_r1-r3/r02_*timeline*), so launches are what to remove.  A LayerNorm needs complete
// rows and an output projection needs all heads, which used to force a launch boundary after every attention.
// Here every (stream, head) workgroup REDUCES THE PRODUCER'S PARTIAL SUMS ITSELF (W <= 16 rows x d: 10 KB per
// partial, L2-resident - blockIdx.x is the stream, so the H workgroups of a stream share an XCD and its L2)
// and recomputes the cheap row-local LayerNorm redundantly (HPW heads per workgroup, see below: H / HPW times), while
// the expensive parts - weight streaming,
// K/V streaming, MFMA work - are split across heads without redundancy: Wqkv / Wq by output columns of the
// head, Wo / Wo2 by the head's K slice (their partial products are summed by the next kernel's prologue in
// fixed head order: deterministic).  No cross-workgroup hand-off inside a launch (an agent-scope release /
// acquire costs an L2 write-back + invalidate on this part); the kernel boundary is the only barrier.
//
// HPW (heads per workgroup, round 4).  With one head per workgroup every one of the H workgroups of a stream re-reads
// ALL partial sums of its stream's rows (8 x redundant at XL dims): cheap while few streams are active, but at a full
// bucket (1280 rows) it is ~80 MB per layer through the fabric and costs more than the three launches the form saves
// (result (k)).  HPW = 4 puts four heads into one 1024-thread workgroup (four head groups of four
// waves, each running exactly the single-head code on its own LDS region): the prologue - reduce + LayerNorm - is
// done ONCE per four heads and shared through LDS, the four heads' shares of the output projection are summed in LDS
// before they are stored, so the next kernel reduces H / HPW = 2 partial products per row instead of 8.  256 workgroups
// of 16 waves fill the 256 CUs of a full 128-stream bucket exactly like the 1024 workgroups of 4 waves did.
#define SC_STAMP_ON (p.dbg_stamp)
#include "common.h"
#include "attn.h"
#include <mutex>

#define CTRL(s, f) sb.ctrl[(s) * 8 + (f)]
#define YSEQ(pp, s, h) (sb.yseq + (((long)(pp) * sb.S + (s)) * sb.W + (h)) * sb.LCAP)
#define ANC(pp, s) (sb.anc + ((long)(pp) * sb.S + (s)) * sb.LCAP * sb.W)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dl_h4 __attribute__((ext_vector_type(4)));

// fp16 decoder mode (WH; BASELINE configs[4], scasr.h: sc_dec_layer.wqkv_pph ...): the weight fragments are the fp16
// copies (same fragment order, 8 bytes per lane and half k-block instead of 16), the A operands are rounded to fp16 on
// their way from LDS, two v_mfma_f32_16x16x16_f16 cover the 8 k values a lane holds per 32-wide k block (fp32
// accumulation).  The same 8 k-steps for NA independent accumulators, as dl_mfma8_il below.
template <int NA>
__device__ __forceinline__ void dl_mfma8_il_h(f32x4 (&acc)[NA], const float4 &a0, const float4 &a1, const dl_h4 (&b0)[NA],
                                              const dl_h4 (&b1)[NA]) {
  const dl_h4 h0{(_Float16)a0.x, (_Float16)a0.y, (_Float16)a0.z, (_Float16)a0.w};
  const dl_h4 h1{(_Float16)a1.x, (_Float16)a1.y, (_Float16)a1.z, (_Float16)a1.w};
#pragma unroll
  for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(h0, b0[t], acc[t], 0, 0, 0);
#pragma unroll
  for (int t = 0; t < NA; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(h1, b1[t], acc[t], 0, 0, 0);
}

// partial products between the decoder's kernels: fp32, or fp16 elements at the same element offsets (sc_search.act_half)
__device__ __forceinline__ float4 dl_load_part(const float *base, long elem, bool half) {
  if (half) {
    const dl_h4 h = *reinterpret_cast<const dl_h4 *>(reinterpret_cast<const _Float16 *>(base) + elem);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  }
  return *reinterpret_cast<const float4 *>(base + elem);
}
__device__ __forceinline__ void dl_store_part(float *base, long elem, const float4 &v, bool half) {
  if (half)
    *reinterpret_cast<dl_h4 *>(reinterpret_cast<_Float16 *>(base) + elem) =
        dl_h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  else
    *reinterpret_cast<float4 *>(base + elem) = v;
}

__device__ __forceinline__ f32x4 dl_mfma8(f32x4 acc, const float4 &a0, const float4 &a1, const float4 &b0,
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

// the same 8 k-steps for NA independent accumulators, interleaved: consecutive MFMAs never share an accumulator
// (v_mfma_f32_16x16x4_f32 issues every 32 cycles per SIMD but a dependent accumulate waits 40: with one wave per
// SIMD - small buckets - eight back-to-back steps on one accumulator run at 80 % of the matrix rate)
template <int NA>
__device__ __forceinline__ void dl_mfma8_il(f32x4 (&acc)[NA], const float4 &a0, const float4 &a1, const float4 (&b0)[NA],
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

struct DecLayerArgs {
  sc_search sb;
  int li;
  const float *xin;  // residual stream before this kernel [S*W][d]
  float *xout;       // ... after its prologue (!= xin: sibling workgroups read xin at unrelated times)
  // prologue:  x[row] = xin[row] + (sum_{z < npart} part[z*zs + row*rs + :] + pbias)
  const float *part;
  int npart;
  int pgrp;          // canonical order of the sum over the partials (see the prologue): 0 tree, 4 groups of four heads, 1 in order
  long zs, rs;
  const float *pbias;
  const float *ln_g, *ln_b;
  const float *wp, *bp;  // projection behind the LayerNorm: sc_pack_panel_weight copy of Wqkv [3d][d] / Wq [d][d], bias
  const float *wop;      // output projection [d][d] (self: linear_out of self_attn, cross: of src_attn), panel-packed
  float *ph;             // [S*W][H][d] partial output projection of every head
  int dbg_stamp;         // SC_PHASE_DBG builds: this launch leaves phase stamps (common.h)
};

// LDS floats of one workgroup (host and device use the same formula)
// region of a head group: a union over the phases - LayerNorm tile (one head per workgroup only: the four-head form keeps it
// in the shared area), split-K partials of the projection, attention partial states (+ the group's own row list of 512
// positions, one head per workgroup), A tile + staging of the output projection
__host__ __device__ static inline int dl_region_floats(int D, int DK, int W, int WM, bool self, int hpw = 1) {
  const int nt = (self ? 3 : 1) * (DK / 16);
  const int xn = hpw > 1 ? 0 : 16 * (D + 4);
  const int ps = 4 * WM * (nt * 16 + 4);                 // (the MFMA tiles' rows >= WM are padding: not stored)
  const int attn = mattn_partial_floats(DK, self ? 5 : 4) + ((self && hpw == 1) ? 512 * W : 0) + 8;   // partial states, row list, wtot
  const int outp = 16 * (32 * (DK > 32 ? DK / 32 : 1) + 4) + WM * (D + 4);   // A tile of the head's k blocks + staging
  int r = xn > ps ? xn : ps;
  r = r > outp ? r : outp;
  return r > attn ? r : attn;
}
// ... of one head group: region, qs, kvn, ctx
__host__ __device__ static inline int dl_group_floats(int D, int DK, int W, int WM, bool self, int hpw = 1) {
  return dl_region_floats(D, DK, W, WM, self, hpw) + 16 * DK /*qs*/ + (self ? WM * 2 * DK : 0) /*kvn*/ + WM * DK /*ctx*/;
}
// HPW head groups + LayerNorm gamma | beta + (HPW > 1) the LayerNorm tile shared by the head groups and, behind it, the
// self-attention's shared row list (128 * hpw positions x W entries + 16 wave totals): an area of its own since round 5 -
// the list is built at the START of the kernel, while the producer's partial sums travel (the ancestor table does not
// depend on this layer's x), so the LayerNorm tile and the list are alive together
__host__ __device__ static inline int dl_lds_floats(int D, int DK, int W, int WM, bool self, int hpw = 1) {
  const int shared = hpw > 1 ? 16 * (D + 4) + (self ? 128 * hpw * W + 16 : 0) : 0;
  return hpw * dl_group_floats(D, DK, W, WM, self, hpw) + 2 * D + 16 /*pool rows of the new tokens*/ + shared;
}

// key tiles per wave in flight in the attention walk of the few-streams variant (UNR = 8).  Measured in round 3 with 8
// instead of 4 (two batches of dependent HBM round trips instead of three at T = 600): strict lock-step 2269 -> 2080
// audio-s/s, single stream 6.8 -> 7.5-9.2 ms per hop - the walk is not bound by the number of round trips.
#ifndef SC_LAYER_NTW
#define SC_LAYER_NTW 4
#endif
// A/B switches of the four-heads-per-workgroup form (tools/build_variant.sh <name> "-DSC_EARLY_KV=0 ..."; defaults = the product):
#ifndef SC_EARLY_LIST
#define SC_EARLY_LIST 1   // 1: the self-attention's row list is built in the prologue (0: after the projection, round 4)
#endif
#ifndef SC_EARLY_KV
#define SC_EARLY_KV 1     // 1: a wave's first batch of K|V tiles is requested behind the projection's MFMAs (needs SC_EARLY_LIST)
#endif
#ifndef SC_HPW_NTW
#define SC_HPW_NTW 2      // tiles per wave in flight in the four-head form's SELF walk (2 | 4; the arithmetic is the same: canonical batches)
#endif
#ifndef SC_HPW_NTW_CROSS
#define SC_HPW_NTW_CROSS 2   // ... in its CROSS walk
#endif
#ifndef SC_HPW_TOUCH
#define SC_HPW_TOUCH 0       // 1: the four-head form requests a dword of every weight line ahead of its prologue (rounds 4-5)
#endif
// (round 6, A/B switch, default OFF: measured flat) SELF kernel of the four-head form with its q|k|v projection split over
// the four waves of a head by COLUMN TILES instead of by K quarter.  One wave projects k, one v - two column tiles over all
// of K, one MFMA chain per K quarter, the quarters added in registers in the canonical order, + bias, straight into the new
// token's row; two waves share q (K quarters 0-1 and 2-3) and leave P0+P1, P2, P3 in LDS, which the q hand-off adds in the
// same order: no reduce of all 96 columns through LDS.  The roles rotate over the head groups, so that every SIMD (wave w
// of every group) carries one wave of each role.  Same sums in the same order - bit-identical to the K-split form
// (tests/test_gpu_baseline_size.py -k stream_resident with this switch on: identical token ids and float64 totals).
// What it measured (docs/r06_findings.md section 6, profiles/r06_phase_times_role_projection.txt): the hand-off costs 0.45 us
// where the reduce cost 2.0, but the projection itself takes 12-13 us for 5.5 us of matrix work - in this form a wave waits
// ~2.1 us for every refill of its fragment registers (128 registers per lane hold one K quarter; two: spills, 18 us), in
// the K-split form the same time goes elsewhere (a probe without loads is no faster).  3373 / 3386 against 3388 / 3402
// audio-s/s in one job: not the product.
#ifndef SC_SELF_ROLES
#define SC_SELF_ROLES 0
#endif
#ifndef SC_ROLE_PF
#define SC_ROLE_PF 2         // k-blocks of weight fragments a wave holds (= one K quarter; 4 = two quarters: spills)
#endif
#ifndef SC_ROLE_PRIO
#define SC_ROLE_PRIO 0       // 1: instruction priority by head group during the projection (younger waves first: same total)
#endif
#ifndef SC_ROLE_UNROLL
#define SC_ROLE_UNROLL 0     // 1: the loop over the K quarters unrolled (instruction footprint: no difference)
#endif
template <int D, int DK, int WM, bool SELF, int UNR, bool FIRST, bool KVH, int HPW = 1, bool WH = false>
__global__ __launch_bounds__(256 * HPW, (UNR <= 4 && WM <= 10) ? (DK > 32 ? 2 : 4) : 1) void dec_layer_attn_kernel(DecLayerArgs p) {
  typedef typename std::conditional<WH, dl_h4, float4>::type BF;   // 4 weight elements of a fragment
  const bool acth = (p.sb.act_half & 2) != 0;                       // partial products in fp16
  constexpr int NTH = 256 * HPW;   // threads: HPW head groups of 4 waves
  constexpr int PCH = 256;       // positions per chunk of a head group's own row list (SELF, one head per workgroup): every
                                 // thread one position (128 until round 4: half as many build phases per walk now)
  constexpr int LDX = D + 4, KI = D / 32, KPW = KI / 4;
  constexpr int NTQ = DK / 16, NT = (SELF ? 3 : 1) * NTQ, LDP = NT * 16 + 4;
  static_assert(KI % 4 == 0 && (DK == 16 || DK == 32 || DK == 64), "d_model must be a multiple of 128, head dim 16, 32 or 64");
  // (round 6) head dim 64 - the reference's geometry when config.yaml names no heads (4 heads of 64 at d = 256,
  // speech2text_streaming.py:221-227, 236-244): its 12 column tiles of q|k|v are projected in passes of NTP tiles (the weight
  // fragments of one pass in registers at a time); one pass, i.e. the code of rounds 2-5, for head dims 16 and 32
  constexpr int NTP = NT > 6 ? 4 : NT, NPASS = NT / NTP;
  static_assert(NT % NTP == 0, "whole passes");
  static_assert(DK <= 32 || HPW == 1, "head dim 64 runs one head per workgroup");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sc_search &sb = p.sb;
  // grid.x runs over the compaction bucket: the k-th stream of rowmap's active-first order (scasr.h: rowmap)
  // (grid.x is padded to a multiple of 8: workgroup (x, y) has linear id x + gridDim.x * y and runs on XCD id % 8,
  // so the H workgroups of a stream - and the same stream's workgroups of the next launch - share one XCD's L2)
  if ((int)blockIdx.x >= (sb.rowmap ? sb.n_rows / sb.W : sb.S)) return;
  const int s = sb.rowmap ? sb.rowmap[blockIdx.x * sb.W] / sb.W : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr bool ROLES = SC_SELF_ROLES && HPW > 1 && SELF && DK == 32;
  // (ROLES: the wave's number as a scalar - its role is a scalar branch, the fragment addresses SGPR bases)
  const int wvi = ROLES ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int g = ROLES ? wvi >> 2 : tid >> 8, gt = tid & 255, wave = ROLES ? wvi & 3 : gt >> 6;   // head group, thread and wave inside it
  const int head = blockIdx.y * HPW + g;
  const int role = ROLES ? (wave + g) & 3 : 0;      // 0: q, K quarters 0-1   1: q, K quarters 2-3   2: k   3: v
  const int rcb = role < 2 ? 0 : role - 1;         // the role's column block of the fused q|k|v weights
  if (!CTRL(s, SC_C_ACTIVE)) return;
  const int nh = CTRL(s, SC_C_NHYP);
  if (nh <= 0) return;
  const int L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR), T = CTRL(s, SC_C_T);
  const int W = sb.W, NPH = sb.H / HPW;
  const int GS = dl_group_floats(D, DK, W, WM, SELF, HPW);   // LDS floats of one head group
  float *region = smem + g * GS;
  float *qs = region + dl_region_floats(D, DK, W, WM, SELF, HPW);   // [16][DK] queries / sqrt(dk), rows >= W zero
  float *kvn = qs + 16 * DK;                                 // SELF: [WM][2*DK] k|v of the new token
  float *ctx = kvn + (SELF ? WM * 2 * DK : 0);               // [WM][DK] attention output of this head
  float *gb = smem + HPW * GS;                               // [2][D] LayerNorm gamma | beta (shared)
  int *ancs = reinterpret_cast<int *>(gb + 2 * D);           // [16] SELF: pool rows of the new tokens (sc_kv_alloc)
  float *Xsh = HPW > 1 ? gb + 2 * D + 16 : region;           // [16][D+4] LayerNorm tile: shared by the head groups
  constexpr int PCS = 128 * (HPW > 1 ? HPW : 4);             // positions per row list (512 in every form: canonical tiles)
  int *srows = reinterpret_cast<int *>(Xsh + 16 * (D + 4)), *swtot = srows + PCS * sb.W;   // HPW > 1, SELF: the shared row list

  // ------------------------------------------------------------------ L2 warm-up of this head's weight slices
  // One dword per 128-B line of the projection fragments and of the output-projection fragments, issued before
  // anything else: when few streams are active the weights come from the Infinity Cache / HBM (0.4-0.9 us per
  // dependent miss); the loads that feed the MFMAs later then hit L2.  The values only keep the loads alive.
  SC_STAMP(SELF ? 0 : 1, 0);
  if (SELF && tid < 16) ancs[tid] = ANC(cur, s)[(long)(L - 1) * W + min(tid, nh - 1)];   // (read behind the prologue's barriers)
  // LayerNorm parameters: issued first, parked in LDS once the partial sums (issued later, returned later) are in
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < D / 2) gbv = *reinterpret_cast<const float4 *>((tid < D / 4 ? p.ln_g : p.ln_b) + 4 * (tid % (D / 4)));
  // (round 6) the bias elements of this thread's share of the split-K reduce: requested here, they used to be one more
  // dependent round trip in the middle of the kernel (the elements e = gt + 256 k of the WM x NT*16 tile)
  constexpr int NBP = (WM * ((SELF ? 3 : 1) * (DK / 16)) * 16 + 255) / 256;
  float bpre[NBP];   // (requested behind the prologue, when its registers are free: they travel during the projection)
  // PF (few streams active: registers are plentiful): the projection's B fragments are fetched into registers
  // right here instead, so that the MFMAs behind the LayerNorm wait for nothing.
  constexpr bool PF = UNR >= 8;
  static_assert(!(PF && HPW > 1), "the few-streams variant runs one head per workgroup");
  // EARLY (HPW > 1, full buckets; phase stamps r04: 6.7 of the self kernel's 40 us went into a projection with 1.3 us of
  // MFMA work - its first B fragments were requested behind the LayerNorm's barrier): k-block 0 of the projection is
  // requested as soon as the x rows are in LDS and travels during the LayerNorm; the output projection's fragments
  // travel during the merge of the attention's partial states.
  constexpr bool EARLY = HPW > 1;
  BF pfb[ROLES ? SC_ROLE_PF * 4 : (PF || EARLY) ? NTP * 2 : 1];   // k-block 0 of this wave (first pass); the later ones are fetched behind the MFMAs of their predecessor
                                                                  // (ROLES: the first SC_ROLE_PF k-blocks of the wave's two column tiles)
  float rbias[2] = {0.f, 0.f}, qbias = 0.f;                       // ROLES: bias of the lane's column of either tile (k, v); of the thread's q column in the hand-off
  // ... and the ancestor slots of the first 128 positions (the row list of the self-attention starts from them)
  // (HPW > 1, round 5: of the first 512 positions, one per thread of the workgroup - the list is BUILT before the prologue's
  // partial sums have arrived, see early_list below)
  int slp[WM] = {};
  if ((PF || (HPW > 1 && SC_EARLY_LIST)) && SELF) {
    const int *anc0 = ANC(cur, s);
    const int pt = HPW > 1 ? tid : gt;
    const bool live0 = pt < (HPW > 1 ? PCS : PCH) && pt < L - 1;
#pragma unroll
    for (int h = 0; h < WM; ++h) slp[h] = anc0[(long)(live0 ? pt : 0) * W + min(h, nh - 1)];
  }
  int U0 = 0;   // HPW > 1, SELF: entries of the row list of the first 512 positions
  auto early_list = [&]() {
    if constexpr (HPW > 1 && SELF && SC_EARLY_LIST)
      U0 = mattn_build_rows<WM, true, NTH, PCS>(srows, swtot, ANC(cur, s), 0, L - 1, W, nh, tid, lane, tid >> 6, slp);
  };
  float touch = 0.f, kvtouch = 0.f;
  // (loads return in issue order: the fragments are requested right BEHIND the first batch of partial sums, which
  // the LayerNorm needs first)
  const int rki0 = role == 1 ? KI / 2 : 0;   // ROLES: the wave's first k-block
  auto role_frag = [&](int ki, BF (&f0)[2], BF (&f1)[2]) {   // k-block ki of the wave's two column tiles
#ifdef SC_ROLE_PROBE_L1    // timing probe only (wrong sums): every k-block = the first one, i.e. a projection whose fragment loads hit the L1
    ki = rki0;
#endif
#ifdef SC_ROLE_PROBE_ROT   // timing probe only (wrong sums): every workgroup / head walks the k-blocks from another start
    {
      const int nkb = role < 2 ? KI / 2 : KI;
      ki = rki0 + ((ki - rki0 + ((blockIdx.x * 3 + head) % nkb)) % nkb);
    }
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tile = (rcb * D + head * DK) / 16 + t;
      const BF *wq = reinterpret_cast<const BF *>(p.wp) + ((long)tile * KI + ki) * 128 + lane;
      f0[t] = wq[0];
      f1[t] = wq[64];
    }
  };
  auto prefetch_w = [&]() {
    if constexpr (ROLES) {
#pragma unroll
      for (int kb = 0; kb < SC_ROLE_PF; ++kb) {
        BF f0[2], f1[2];
        role_frag(rki0 + kb, f0, f1);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          pfb[kb * 4 + t * 2] = f0[t];
          pfb[kb * 4 + t * 2 + 1] = f1[t];
        }
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
      const int tile = ((t / NTQ) * D + head * DK) / 16 + (t % NTQ);
      const BF *wq = reinterpret_cast<const BF *>(p.wp) + ((long)tile * KI + wave * KPW) * 128 + lane;
      pfb[t * 2] = wq[0];
      pfb[t * 2 + 1] = wq[64];
    }
  };
  // (round 6) NOT in the four-head form: at the buckets it serves every XCD's L2 holds the fragments anyway (32 workgroups per
  // XCD read the same ones), and the 384 KB of lines a workgroup touched stood IN FRONT of its partial sums in the load queue
  // (loads return in issue order) - the stream-resident kernel fetches the same rows in 4.3 us where this prologue took 11
  constexpr bool TOUCH = HPW == 1 || SC_HPW_TOUCH;
  if constexpr (TOUCH) {
    if (PF) {
      if (FIRST) prefetch_w();
    } else {
      constexpr int NW = (SELF ? 3 : 1);
      const int ofs = gt * 32;   // floats: 128 B per thread, 32 KB per head group and instruction
#pragma unroll
      for (int wh = 0; wh < NW; ++wh) {
        if (WH) {   // 2-byte elements: the same fragments are half as many bytes
          const _Float16 *base = reinterpret_cast<const _Float16 *>(p.wp) + ((long)((wh * D + head * DK) / 16) * KI) * 512;
          if (2 * ofs < NTQ * KI * 512) touch += (float)base[2 * ofs];
        } else {
          const float *base = p.wp + ((long)((wh * D + head * DK) / 16) * KI) * 512;   // NTQ tiles x KI k-blocks x 2 KB
          if (ofs < NTQ * KI * 512) touch += base[ofs];
        }
      }
    }
    // output projection: D/16 tiles, k-block (head*DK)/32, 2 KB each
    constexpr int TPW = D / 16 * 16;   // lines: 16 per tile
    if (!WH && gt < TPW) touch += p.wop[((long)((gt >> 4) * KI + (head * DK) / 32) * 2) * 256 + (gt & 15) * 32];
  }

  // ------------------------------------------------------------------ prologue: x rows, LayerNorm -> Xn
  // element-parallel: every thread owns float4 pieces of the W x D tile and sums the producer's partial sums for
  // them with ALL loads of a batch in flight at once (one memory round trip per ZB partial sums); the row-wise
  // LayerNorm then runs on the LDS tile.
  {
    float *Xn = Xsh;  // [16][LDX], rows >= W zero
    constexpr int C4 = D / 4, QT = (16 * C4 + NTH - 1) / NTH;   // float4 pieces per row; pieces per thread (16 rows)
    constexpr int QN = (WM * C4 + NTH - 1) / NTH;               // ... of the rows that can be live (W <= WM)
    constexpr int ZB = (UNR >= 8 || HPW > 1) ? 8 : 4;   // partial sums per batch (register budget of the variant; HPW > 1: one piece per thread)
    float4 xv[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int e = tid + NTH * q, i = e / C4, c4 = e % C4;
      xv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < W) {
        if (FIRST) {
          // embed*sqrt(d) + PE of the newest token (transformer_decoder.py:231); rows >= nh repeat row nh-1
          const int tok = YSEQ(cur, s, min(i, nh - 1))[L - 1];
          const float sq = sqrtf((float)D);
          const float4 ev = *reinterpret_cast<const float4 *>(sb.embed + (long)tok * D + 4 * c4);
          const float4 pe = *reinterpret_cast<const float4 *>(sb.pe + (long)(L - 1) * D + 4 * c4);
          xv[q] = make_float4(ev.x * sq + pe.x, ev.y * sq + pe.y, ev.z * sq + pe.z, ev.w * sq + pe.w);
        }
      }
    }
    if (FIRST) early_list();
    if (!FIRST) {
      // the residual rows and the bias are requested TOGETHER with the first batch of partial sums: behind them they were
      // one more dependent round trip of ~2 us (rows another XCD wrote; tools/boundary_probe.hip)
      float4 xi[QN], pbv[QN];
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int e = tid + NTH * q, i = min(e / C4, W - 1), c4 = e % C4;
        xi[q] = *reinterpret_cast<const float4 *>(p.xin + ((long)s * W + i) * D + 4 * c4);
        pbv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.pbias) pbv[q] = *reinterpret_cast<const float4 *>(p.pbias + 4 * c4);
      }
      // canonical order of the producer's partial sums (common.h) - the same for every kernel form:
      //   pgrp == 0: split sums of the fused feed-forward -> balanced tree over the aligned index pairs, 8 at a time
      //   pgrp >= 1: per-head partial products -> aligned groups of four heads in head order (pgrp == 4; pgrp == 1: the
      //              four-heads-per-workgroup producer has done that), the groups in order
      float4 yv[QN];
      for (int z0 = 0; z0 < p.npart; z0 += 8) {
        float4 h8[QN];   // this batch of 8
#pragma unroll
        for (int zb = 0; zb < 8; zb += ZB) {
          if (zb > 0 && z0 + zb >= p.npart) break;
          float4 pv[QN][ZB];
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int e = tid + NTH * q, i = min(e / C4, W - 1), c4 = e % C4;
            const long row = (long)s * W + i;
#pragma unroll
            for (int z = 0; z < ZB; ++z) {
              pv[q][z] = dl_load_part(p.part, (long)min(z0 + zb + z, p.npart - 1) * p.zs + row * p.rs + 4 * c4, acth);
              if (z0 + zb + z >= p.npart) pv[q][z] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
          if (PF && z0 == 0 && zb == 0) prefetch_w();
          if (z0 == 0 && zb == 0) early_list();   // (two barriers; the partial sums requested above travel meanwhile)
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            if (p.pgrp == 0) {          // tree: (p0+p1)+(p2+p3) [+ (p4+p5)+(p6+p7)]
              float4 t = sc_tree4(pv[q][0], pv[q][1], pv[q][2], pv[q][3]);
              if (ZB == 8) t = sc_add4(t, sc_tree4(pv[q][ZB - 4], pv[q][ZB - 3], pv[q][ZB - 2], pv[q][ZB - 1]));
              h8[q] = zb == 0 ? t : sc_add4(h8[q], t);
            } else if (p.pgrp == 4) {   // groups of four heads
              float4 t = sc_seq4(pv[q][0], pv[q][1], pv[q][2], pv[q][3]);
              h8[q] = zb == 0 ? t : sc_add4(h8[q], t);
              if (ZB == 8 && z0 + zb + 4 < p.npart) h8[q] = sc_add4(h8[q], sc_seq4(pv[q][ZB - 4], pv[q][ZB - 3], pv[q][ZB - 2], pv[q][ZB - 1]));
            } else {                    // in order
#pragma unroll
              for (int z = 0; z < ZB; ++z)
                if (z0 + zb + z < p.npart) h8[q] = (zb + z == 0) ? pv[q][0] : sc_add4(h8[q], pv[q][z]);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) yv[q] = z0 == 0 ? h8[q] : sc_add4(yv[q], h8[q]);
      }
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int e = tid + NTH * q, i = e / C4;
        if (i < W)
          xv[q] = make_float4(xi[q].x + (yv[q].x + pbv[q].x), xi[q].y + (yv[q].y + pbv[q].y), xi[q].z + (yv[q].z + pbv[q].z),
                              xi[q].w + (yv[q].w + pbv[q].w));
      }
    }
    if (PF && !SELF) {
      // warm-up of this head's encoder K|V lines (one element per 128-B K line and V line of every frame): requested
      // now, behind everything the prologue waits for, they travel from HBM / the Infinity Cache while the LayerNorm
      // and the q projection run; the attention's operand loads then hit L2.  Consumed at the very end.
      const long ck0 = ((long)s * sb.n_layers + p.li) * sb.TCAP * 2 * D + head * DK;
      for (int t = tid; t < T; t += 256) {
        float e2[1];
        kv_loadn<1, KVH>(sb.ckv, ck0 + (long)t * 2 * D, e2);
        kvtouch += e2[0];
        kv_loadn<1, KVH>(sb.ckv, ck0 + (long)t * 2 * D + D, e2);
        kvtouch += e2[0];
      }
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      const int e = tid + NTH * q, i = e / C4, c4 = e % C4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                   // rows >= W: zeros
      if (q < QN) v = xv[q];
      if (16 * C4 % NTH == 0 || e < 16 * C4) {
        *reinterpret_cast<float4 *>(Xn + i * LDX + 4 * c4) = v;
        if (blockIdx.y == 0 && i < W) *reinterpret_cast<float4 *>(p.xout + ((long)s * W + i) * D + 4 * c4) = v;
      }
    }
    if (EARLY) prefetch_w();
    if constexpr (ROLES) {
      qbias = p.bp[head * DK + (gt & (DK - 1))];   // q hand-off: thread gt owns column gt % DK of two rows
      if (role >= 2)
#pragma unroll
        for (int t = 0; t < 2; ++t) rbias[t] = p.bp[rcb * D + head * DK + t * 16 + (lane & 15)];
    } else
#pragma unroll
    for (int k = 0; k < NBP; ++k) {
      const int n = (gt + 256 * k) % (((SELF ? 3 : 1) * (DK / 16)) * 16);
      bpre[k] = p.bp[(n / DK) * D + head * DK + n % DK];
    }
    if (tid < D / 2) *reinterpret_cast<float4 *>(gb + 4 * tid) = gbv;
    if (touch == 123456.789f) Xn[0] = touch;   // never true: keeps the warm-up loads (they have returned by now:
    __syncthreads();                           // loads return in order and the partial sums were waited for)
    SC_STAMP(SELF ? 0 : 1, 1);
    if (HPW == 1 || tid < 256) {   // LayerNorm in place: 16 lanes per row, all 16 rows of the tile at once (DPP reductions)
      constexpr int Q4 = D / 64;
      const int i = tid >> 4, sub = tid & 15;
      float4 x[Q4];
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        x[q] = *reinterpret_cast<const float4 *>(Xn + i * LDX + 4 * (sub + 16 * q));
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
          const float4 gm = *reinterpret_cast<const float4 *>(gb + 4 * (sub + 16 * q));
          const float4 bt = *reinterpret_cast<const float4 *>(gb + D + 4 * (sub + 16 * q));
          *reinterpret_cast<float4 *>(Xn + i * LDX + 4 * (sub + 16 * q)) =
              make_float4((x[q].x - mean) * rstd * gm.x + bt.x, (x[q].y - mean) * rstd * gm.y + bt.y,
                          (x[q].z - mean) * rstd * gm.z + bt.z, (x[q].w - mean) * rstd * gm.w + bt.w);
        }
      }
    }
  }
  __syncthreads();
  SC_STAMP(SELF ? 0 : 1, 2);

  constexpr int NTW_ = (UNR >= 8) ? SC_LAYER_NTW : (HPW > 1 ? (SELF ? SC_HPW_NTW : SC_HPW_NTW_CROSS) : 2);   // tiles per wave in flight in the attention walk
  MBatch<DK, NTW_> kvb0;                                 // HPW > 1, SELF: the wave's first batch, requested ahead (below)
  const long skv0 = ((long)s * sb.n_layers + p.li) * sb.kv_rows * 2 * D + head * DK;   // element offset (fp32 or fp16 pool)
  if constexpr (ROLES) {
    // ---------------------------------------------------------------- projection by role (SC_SELF_ROLES): two column tiles
    // per wave, one MFMA chain per K quarter (KPW k-blocks); canonical order of a column's sum: ((P0 + P1) + P2) + P3 + bias
    constexpr int LQP = DK + 4;                                   // q partials in LDS: [3][WM][LQP] = P0+P1 | P2 | P3,
    float *QP = region + mattn_partial_floats(DK, 5) + 8;         // behind the attention's partial states (written later)
    static_assert(5 * 16 * (DK + 3) + 8 + 3 * WM * LQP <= 4 * WM * (NT * 16 + 4), "the q partials fit the region (dl_region_floats: ps)");
    const float *Xn = Xsh;
    const int r = lane & 15, kk = lane >> 4;
#if SC_ROLE_PRIO
    // the issue arbiter prefers the oldest wave: the waves of head group 3 finished their FIRST K quarter 6.4 us after the
    // barrier (group 0: 1.4) with their fragments in registers since the prologue - the younger groups go first instead
    if (g == 1) __builtin_amdgcn_s_setprio(1);
    else if (g == 2) __builtin_amdgcn_s_setprio(2);
    else if (g == 3) __builtin_amdgcn_s_setprio(3);
#endif
    // a LOOP over the role's K quarters (KPW k-blocks each, their fragment buffers refilled KPW k-blocks ahead): the same
    // 16 * KPW MFMAs per pass through ONE piece of code (SC_ROLE_UNROLL = 1: every quarter its own code, as the K-split form)
    auto proj = [&](int nq, auto on_quarter) {
      constexpr int QA = SC_ROLE_PF / KPW;   // K quarters whose fragments a wave holds (1 | 2): the loop refills QA quarters ahead
      static_assert(SC_ROLE_PF % KPW == 0 && (QA == 1 || QA == 2), "whole K quarters in flight");
      BF f0[SC_ROLE_PF][2], f1[SC_ROLE_PF][2];
#pragma unroll
      for (int kb = 0; kb < SC_ROLE_PF; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f0[kb][t] = pfb[kb * 4 + t * 2];
          f1[kb][t] = pfb[kb * 4 + t * 2 + 1];
        }
#if SC_ROLE_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
      for (int q0 = 0; q0 < nq; q0 += QA) {
#pragma unroll
        for (int h = 0; h < QA; ++h) {
          const int q = q0 + h;
          f32x4 accq[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) accq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < KPW; ++kb) {
            const int ki = rki0 + q * KPW + kb;
            const float *ab = Xn + r * LDX + ki * 32 + 8 * kk;
            const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
            if constexpr (WH) dl_mfma8_il_h<2>(accq, a0, a1, f0[h * KPW + kb], f1[h * KPW + kb]);
            else dl_mfma8_il<2>(accq, a0, a1, f0[h * KPW + kb], f1[h * KPW + kb]);
            if (q + QA < nq) role_frag(ki + SC_ROLE_PF, f0[h * KPW + kb], f1[h * KPW + kb]);   // in flight during the next quarters' MFMAs
          }
          on_quarter(q, accq);
        }
      }
    };
    auto store_qp = [&](int z, const f32x4 (&v)[2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * kk + j < WM) QP[(z * WM + 4 * kk + j) * LQP + t * 16 + r] = v[t][j];
    };
    if (role < 2) {
      f32x4 run[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      proj(2, [&](int Q, const f32x4 (&a)[2]) {
        if (role == 0) {   // P0, P1 -> P0 + P1
#pragma unroll
          for (int t = 0; t < 2; ++t) run[t] = Q == 0 ? a[t] : run[t] + a[t];
          if (Q == 1) store_qp(0, run);
        } else {           // P2, P3: added by the hand-off, one after the other
          store_qp(1 + Q, a);
        }
      });
    } else {
      f32x4 run[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      proj(4, [&](int Q, const f32x4 (&a)[2]) {
#ifdef SC_ROLE_STAMP_Q
        SC_STAMP_BY(0, 9 + Q, g == SC_ROLE_STAMP_Q - 1 && role == 2 && lane == 0 && Q < 3);   // (K quarters 0..2 of the k role's wave are done)
#endif
#pragma unroll
        for (int t = 0; t < 2; ++t) run[t] = Q == 0 ? a[t] : run[t] + a[t];
      });
      // the new token's k (role 2) / v (role 3): LDS for its own attention state, and appended into its pool row - later
      // steps read it from the cache
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * kk + j;
          if (row < WM) {
            const float v = row < W ? run[t][j] + rbias[t] : 0.f;
            kvn[row * 2 * DK + (role - 2) * DK + t * 16 + r] = v;
            if (row < nh) kv_store1<KVH>(sb.skv, skv0 + (long)ancs[row] * 2 * D + (role - 2) * D + t * 16 + r, v);
          }
        }
    }
#ifdef SC_ROLE_STAMP_Q
#elif defined(SC_ROLE_STAMP_G3)
    SC_STAMP_BY(0, 9, g == 3 && wave == 0 && lane == 0);
    SC_STAMP_BY(0, 10, g == 3 && wave == 3 && lane == 0);
    SC_STAMP_BY(0, 11, g == 1 && wave == 2 && lane == 0);
#else
    SC_STAMP_BY(0, 8 + wave, g == 0 && wave > 0 && lane == 0);   // (roles 1..3 of head group 0 are done)
#endif
#if SC_ROLE_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // the wave's first batch of K|V tiles (as in the K-split form below): travels during the hand-off
    if constexpr (SC_EARLY_LIST && SC_EARLY_KV)
      mattn_load<DK, NTW_, KVH>(kvb0, sb.skv, D, cdiv(U0, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
        const int e = srows[min(idx, PCS * W - 1)];
        hm = (unsigned)e >> 16;
        ke = skv0 + (long)(e & 0xFFFF) * 2 * D;
      });
    __syncthreads();
    SC_STAMP(0, 3);
    {
      const float scale = sqrtf((float)DK);
#pragma unroll
      for (int k = 0; k < 16 * DK / 256; ++k) {
        const int e = gt + 256 * k, h = e / DK, c = e % DK;   // (c = gt % DK for every k)
        float v = 0.f;   // hypothesis rows the MFMA tiles pad with
        if (h < W) {
          v = QP[(0 * WM + h) * LQP + c];
          v += QP[(1 * WM + h) * LQP + c];
          v += QP[(2 * WM + h) * LQP + c];
          v += qbias;
          v = v / scale;
        }
        qs[e] = v;
      }
    }
    __syncthreads();
    SC_STAMP(0, 4);
  } else {
  // ------------------------------------------------------------------ projection of the head's columns
  // NT tiles of 16 output columns, K = D split over the 4 waves (KPW k-blocks of 32 each); B operands
  // straight from the fragment-packed weights (1 KB contiguous per wave load)
  {
    const float *Xn = Xsh;
    const int r = lane & 15, kk = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {   // (one pass for head dims 16 / 32)
      auto load_b = [&](int ki, BF (&b0)[NTP], BF (&b1)[NTP]) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          const int tt = ps * NTP + t;
          const int tile = ((tt / NTQ) * D + head * DK) / 16 + (tt % NTQ);
          const BF *wq = reinterpret_cast<const BF *>(p.wp) + ((long)tile * KI + ki) * 128 + lane;
          b0[t] = wq[0];
          b1[t] = wq[64];
        }
      };
      BF b0[NTP], b1[NTP];
      f32x4 accp[NTP];
#pragma unroll
      for (int t = 0; t < NTP; ++t) accp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((PF || EARLY) && ps == 0) {
#pragma unroll
        for (int t = 0; t < NTP; ++t) {
          b0[t] = pfb[t * 2];
          b1[t] = pfb[t * 2 + 1];
        }
      } else {
        load_b(wave * KPW, b0, b1);
      }
#pragma unroll
      for (int kq = 0; kq < KPW; ++kq) {
        const int ki = wave * KPW + kq;
        const float *ab = Xn + r * LDX + ki * 32 + 8 * kk;
#ifdef SC_PROJ_PROBE_NOLDS   // timing probe only (wrong sums): the A operand without its LDS reads
        const float4 a0 = make_float4((float)lane, 1.f, 2.f, (float)ki), a1 = make_float4(3.f, (float)kk, 4.f, 5.f);
#else
        const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#endif
        BF n0[NTP], n1[NTP];
#ifdef SC_PROJ_PROBE_NOLOAD   // timing probe only (wrong sums): the later k-blocks re-use the fragments of the first - a projection without memory
        if (kq + 1 < KPW) {
#pragma unroll
          for (int t = 0; t < NTP; ++t) {
            n0[t] = b0[t];
            n1[t] = b1[t];
          }
        }
#else
        if (kq + 1 < KPW) load_b(ki + 1, n0, n1);   // in flight during this k-block's MFMAs
#endif
        if constexpr (WH) dl_mfma8_il_h<NTP>(accp, a0, a1, b0, b1);
        else dl_mfma8_il<NTP>(accp, a0, a1, b0, b1);
#if defined(SC_PROJ_STAMPS) && SC_PROJ_STAMPS == 2   // (9 / 10: thread 0's / the youngest wave's FIRST MFMA of the phase has its result - its fragments have arrived)
        if (ps == 0 && kq == 0) {
          asm volatile("s_nop 0" : "+v"(accp[0]));
          SC_STAMP_BY(SELF ? 0 : 1, 9, tid == 0);
          SC_STAMP_BY(SELF ? 0 : 1, 10, tid == NTH - 64);
        }
#endif
        if (kq + 1 < KPW) {
#pragma unroll
          for (int t = 0; t < NTP; ++t) {
            b0[t] = n0[t];
            b1[t] = n1[t];
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NTP; ++t) acc[ps * NTP + t] = accp[t];
    }
    // HPW > 1, SELF (round 5): the wave's FIRST batch of K|V tiles is requested here - the rows of the cached positions do not
    // depend on this step's x, the list has been there since the prologue - and travels while the split sums are reduced
    // and the new token's row is appended (1.4 us of a 22 us kernel during which HBM idled)
    if constexpr (HPW > 1 && SELF && SC_EARLY_LIST && SC_EARLY_KV) {
      const long skv0e = ((long)s * sb.n_layers + p.li) * sb.kv_rows * 2 * D + head * DK;
      mattn_load<DK, NTW_, KVH>(kvb0, sb.skv, D, cdiv(U0, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
        const int e = srows[min(idx, PCS * W - 1)];
        hm = (unsigned)e >> 16;
        ke = skv0e + (long)(e & 0xFFFF) * 2 * D;
      });
    }
#if defined(SC_PROJ_STAMPS) && SC_PROJ_STAMPS != 2   // (9: thread 0's MFMAs issued + first K|V batch requested; 10: the youngest wave's; 11: behind the barrier)
    SC_STAMP_BY(SELF ? 0 : 1, 9, tid == 0);
    SC_STAMP_BY(SELF ? 0 : 1, 10, tid == NTH - 64);
#endif
    __syncthreads();  // every wave is done reading Xn: the region becomes the partial products
#ifdef SC_PROJ_STAMPS
    SC_STAMP_BY(SELF ? 0 : 1, 11, tid == 0);
#endif
    float *Ps = region;  // [4][WM][LDP]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * kk + j < WM) Ps[(wave * WM + 4 * kk + j) * LDP + t * 16 + r] = acc[t][j];
  }
  __syncthreads();
  SC_STAMP(SELF ? 0 : 1, 3);
  {
    const float *Ps = region;
    const float scale = sqrtf((float)DK);
    for (int e = WM * DK + gt; e < 16 * DK; e += 256) qs[e] = 0.f;   // hypothesis rows the MFMA tiles pad with
#pragma unroll
    for (int k = 0; k < NBP; ++k) {
      const int e = gt + 256 * k;
      if (e >= WM * NT * 16) break;
      const int w = e / (NT * 16), n = e % (NT * 16);
      const int which = n / DK, c = n % DK;
      float v = 0.f;
      if (w < W) {
        v = Ps[(0 * WM + w) * LDP + n];
        v += Ps[(1 * WM + w) * LDP + n];
        v += Ps[(2 * WM + w) * LDP + n];
        v += Ps[(3 * WM + w) * LDP + n];
        v += bpre[k];
      }
      if (which == 0) {
        qs[w * DK + c] = v / scale;
      } else if (SELF) {
        kvn[w * 2 * DK + (which - 1) * DK + c] = v;
        // append this token's K|V row into its pool row; later steps read it from the cache
        if (w < nh) kv_store1<KVH>(sb.skv, skv0 + (long)ancs[w] * 2 * D + (which - 1) * D + c, v);
      }
    }
  }
  __syncthreads();
  SC_STAMP(SELF ? 0 : 1, 4);
  }   // !ROLES

  // ------------------------------------------------------------------ attention of this head (all hypotheses)
  // matrix-core form (attn.h: mattn_*): the hypotheses are the N dimension, 16 K/V rows a tile; the four waves
  // take tiles round-robin and leave one partial state each (+ one for the new token's own row, SELF)
  // PF: the output projection's B fragments travel while the attention runs
  constexpr int TWO = D / 16 / 4;
  constexpr int KBO = DK > 32 ? DK / 32 : 1;   // k blocks of 32 input columns the head's share of the output projection spans
  BF ob[(PF || EARLY) ? KBO * TWO * 2 : 1];
  auto prefetch_o = [&]() {
#pragma unroll
    for (int kb2 = 0; kb2 < KBO; ++kb2)
#pragma unroll
      for (int t = 0; t < TWO; ++t) {
        const BF *wq = reinterpret_cast<const BF *>(p.wop) + ((long)(wave * TWO + t) * KI + (head * DK) / 32 + kb2) * 128 + lane;
        ob[(kb2 * TWO + t) * 2] = wq[0];
        ob[(kb2 * TWO + t) * 2 + 1] = wq[64];
      }
  };
  if (PF) prefetch_o();
  constexpr int NTW = NTW_;   // tiles per wave and batch
  constexpr int NP = SELF ? 5 : 4;
  float *pm = region;
  float *pl = pm + NP * 16;
  float *pO = pl + NP * 16;
  int *rows = (int *)(region + mattn_partial_floats(DK, NP));
  int *wtot = rows + (SELF ? 2 * PCH * W : 0);
  const long ckv0 = ((long)s * sb.n_layers + p.li) * sb.TCAP * 2 * D + head * DK;
  MAttn<DK> st;
  mattn_init(st);

  if (SELF) {
    const int *anc = ANC(cur, s);
    const int Lc = L - 1;  // cached positions; the new token's row is the fifth partial state
    int urows = nh;   // distinct K|V rows of this (stream, layer): the new tokens' rows + the walked ones
    if constexpr (HPW > 1) {
      // ONE row list per workgroup, 128 * HPW positions at a time, built by all its threads and walked by every head
      // group (it lives where the LayerNorm tile was: the projection is done)
      auto rowfn = [&](int idx, long &ke, unsigned &hm) {
        const int e = srows[min(idx, PCS * W - 1)];   // entries >= U are zero: no hypothesis
        hm = (unsigned)e >> 16;
        ke = skv0 + (long)(e & 0xFFFF) * 2 * D;
      };
      // positions [0, 512): the list was built in the prologue, every wave's first batch is in its registers
      if constexpr (!SC_EARLY_LIST)
        U0 = mattn_build_rows<WM, false, NTH, PCS>(srows, swtot, anc, 0, Lc, W, nh, tid, lane, tid >> 6, slp);
      urows += U0;
      if constexpr (SC_EARLY_LIST && SC_EARLY_KV)
        mattn_walk<DK, NTW, KVH, true>(st, qs, sb.skv, D, cdiv(U0, 16), wave, lane, rowfn, &kvb0);
      else
        mattn_walk<DK, NTW, KVH>(st, qs, sb.skv, D, cdiv(U0, 16), wave, lane, rowfn);
      for (int c0 = PCS; c0 < Lc; c0 += PCS) {
        __syncthreads();  // the list is rebuilt for the next positions
        const int U = mattn_build_rows<WM, false, NTH, PCS>(srows, swtot, anc, c0, Lc, W, nh, tid, lane, tid >> 6, slp);
        urows += U;
        mattn_walk<DK, NTW, KVH>(st, qs, sb.skv, D, cdiv(U, 16), wave, lane, rowfn);
      }
    } else
    // one head per workgroup: the SAME lists of 512 positions (the tiles of a list, and with them the order of the sums,
    // are what every form must share), built by the 256 threads in two passes of 256 positions
    for (int c0 = 0; c0 < Lc; c0 += 2 * PCH) {
      int U = (PF && c0 == 0) ? mattn_build_rows<WM, true, 256, PCH>(rows, wtot, anc, c0, Lc, W, nh, gt, lane, wave, slp, 0, 2 * PCH * W)
                              : mattn_build_rows<WM, false, 256, PCH>(rows, wtot, anc, c0, Lc, W, nh, gt, lane, wave, slp, 0, 2 * PCH * W);   // (attn.h)
      if (c0 + PCH < Lc) U = mattn_build_rows<WM, false, 256, PCH>(rows, wtot, anc, c0 + PCH, Lc, W, nh, gt, lane, wave, slp, U, 0);
      urows += U;
      mattn_walk<DK, NTW, KVH>(st, qs, sb.skv, D, cdiv(U, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
        const int e = rows[min(idx, 2 * PCH * W - 1)];   // entries >= U are zero: no hypothesis
        hm = (unsigned)e >> 16;
        ke = skv0 + (long)(e & 0xFFFF) * 2 * D;
      });
      if (c0 + 2 * PCH < Lc) __syncthreads();  // rows is rebuilt for the next positions
    }
    if (sb.stat_rows && head == 0 && gt == 0) atomicAdd(&sb.stat_rows[1], (unsigned long long)urows);
    // the new token: hypothesis h attends to its own row (slot h at position L-1, still in LDS) only
    if (gt < 16) {
      const int h = gt;
      float sdot = -INFINITY;
      if (h < nh) {
        sdot = 0.f;
#pragma unroll
        for (int c = 0; c < DK; ++c) sdot = fmaf(qs[h * DK + c], kvn[h * 2 * DK + c], sdot);
#pragma unroll
        for (int c = 0; c < DK; ++c) pO[(4 * 16 + h) * (DK + 1) + c] = kvn[h * 2 * DK + DK + c];
      }
      pm[4 * 16 + h] = sdot;
      pl[4 * 16 + h] = h < nh ? 1.f : 0.f;
    }
  } else {
    const unsigned all = (1u << nh) - 1u;
    mattn_walk<DK, NTW, KVH>(st, qs, sb.ckv, D, cdiv(T, 16), wave, lane, [&](int idx, long &ke, unsigned &hm) {
      hm = idx < T ? all : 0u;
      ke = ckv0 + (long)min(idx, T - 1) * 2 * D;
    });
  }
  SC_STAMP(SELF ? 0 : 1, 5);
  mattn_store_partial<DK>(st, pm, pl, pO, wave, lane);
  if (EARLY) prefetch_o();   // (the walk's registers are free now)
  __syncthreads();
  for (int e = gt; e < WM * DK; e += 256) {
    const int h = e / DK, c = e % DK;
    ctx[e] = h < nh ? mattn_final<DK, NP>(pm, pl, pO, h, c) : 0.f;  // rows >= nh: zero context (their partial products stay finite)
  }
  __syncthreads();
  SC_STAMP(SELF ? 0 : 1, 6);

  // ------------------------------------------------------------------ this head's share of the output projection
  // ph[row][head][n] = sum_c ctx[w][c] * Wo[n][head*DK + c]: f32 MFMA over the k-block of 32 input columns that
  // holds this head's DK columns (the columns of a neighbouring head in the same block meet zeros of the A tile);
  // B operands from the fragment-packed copy of Wo, result staged through LDS for full-line stores.
  {
    constexpr int KW = 32 * KBO, LDA = KW + 4, NTO = D / 16, TW = NTO / 4, LDO = D + 4;
    float *As = region;              // [16][LDA]
    float *Os = region + 16 * LDA;   // [WM][LDO] (the tiles' rows >= WM are padding: not stored)
    const int kb = (head * DK) / 32, koff = (head * DK) % 32;
    for (int e = gt; e < 16 * KW; e += 256) {
      const int w = e / KW, c = e % KW;
      As[w * LDA + c] = (w < WM && c >= koff && c < koff + DK) ? ctx[w * DK + c - koff] : 0.f;
    }
    const int r = lane & 15, kk = lane >> 4;
    static_assert(TW == TWO, "tiles per wave");
    BF b0[KBO][TW], b1[KBO][TW];
#pragma unroll
    for (int kb2 = 0; kb2 < KBO; ++kb2)
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        if (PF || EARLY) {
          b0[kb2][t] = ob[(kb2 * TW + t) * 2];
          b1[kb2][t] = ob[(kb2 * TW + t) * 2 + 1];
        } else {
          const BF *wq = reinterpret_cast<const BF *>(p.wop) + ((long)(wave * TW + t) * KI + kb + kb2) * 128 + lane;
          b0[kb2][t] = wq[0];
          b1[kb2][t] = wq[64];
        }
      }
    __syncthreads();
    f32x4 oacc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) oacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb2 = 0; kb2 < KBO; ++kb2) {   // (one chain per head over its k blocks in order)
      const float *ab = As + r * LDA + 32 * kb2 + 8 * kk;
      const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
      if constexpr (WH) dl_mfma8_il_h<TW>(oacc, a0, a1, b0[kb2], b1[kb2]);
      else dl_mfma8_il<TW>(oacc, a0, a1, b0[kb2], b1[kb2]);
    }
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * kk + j < WM) Os[(4 * kk + j) * LDO + (wave * TW + t) * 16 + r] = oacc[t][j];
    __syncthreads();
    SC_STAMP(SELF ? 0 : 1, 7);
    // (HPW > 1: the head groups' shares are summed here, in head order - one partial product per workgroup)
    for (int e = tid; e < W * (D / 4); e += NTH) {
      const int w = e / (D / 4), c4 = e % (D / 4);
      float4 o = *reinterpret_cast<const float4 *>(smem + 16 * LDA + w * LDO + 4 * c4);
#pragma unroll
      for (int gg = 1; gg < HPW; ++gg) {
        const float4 t = *reinterpret_cast<const float4 *>(smem + gg * GS + 16 * LDA + w * LDO + 4 * c4);
        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
      }
      dl_store_part(p.ph, (((long)s * W + w) * NPH + blockIdx.y) * D + 4 * c4, o, acth);
    }
  }
  if (kvtouch == 123456.789f) p.ph[0] = kvtouch;   // never true: keeps the K|V warm-up loads
  SC_STAMP_END(SELF ? 0 : 1, 8);
}
SC_PHASE_GETTER(sc_phase_debug_layer)

// ---------------------------------------------------------------------------------------------------------------
// Heads per workgroup for this launch's compaction bucket (see the top of the file).  One head per workgroup while few
// streams are active (latency-bound chains: more, smaller workgroups spread the weight and K/V streaming over more
// CUs); four from SC_HPW_MIN_ROWS hypothesis rows on, where the redundant prologue reads are what costs.
// SC_DEC_HPW = 1 | 4 forces one form at any size (tests, sweeps).  (Two heads per workgroup, 512 threads, was measured
// between the two: 3005 against 2890 / 3068 audio-s/s for one / four at every bucket size - not instantiated.)
int sc_dec_layer_hpw(const sc_search &sb) {
  const int dk = sb.d / sb.H;
  const bool can = sb.d == 256 && dk == 32 && sb.W <= 10;   // the instantiated HPW > 1 variants (beams <= 5 run in the 10-row tiles too)
  int hpw = 1;
  int min_rows = sb.W > 5 ? SC_HPW_MIN_ROWS : SC_FUSED_MAX_ROWS + 1;   // narrow beams: the 5-row one-head kernels up to the old limit of the form
  if (const char *e = sc_hook("SC_HPW_MIN")) min_rows = atoi(e);   // tools: threshold sweep
  if (can && (sb.rowmap ? sb.n_rows : sb.S * sb.W) >= min_rows) hpw = 4;
  if (const char *e = sc_hook("SC_DEC_HPW")) {
    const int v = atoi(e);
    if (v == 1 || (can && v == 4)) hpw = v;
  }
  while (hpw > 1 && sb.H % hpw) hpw >>= 1;
  return hpw;
}

template <int D, int DK, int WM, bool SELF, int UNR, bool FIRST, bool KVH, int HPW, bool WH = false>
static void launch_dec_layer_variant(const DecLayerArgs &p, int ns, hipStream_t st) {
  const sc_search &sb = p.sb;
  // one or two streams: pad grid.x to 8 so that all H workgroups of a stream land on ONE XCD (linear id x + 8*y) and
  // share its L2 - single stream 5.41 -> 5.25 ms per hop; with more streams the padding concentrates the work on
  // fewer XCDs and costs 1 % at 128 streams, so larger buckets keep their natural spread
  const dim3 grid(ns <= 2 ? 8 : ns, sb.H / HPW);
  const size_t lds = (size_t)dl_lds_floats(D, DK, sb.W, WM, SELF, HPW) * sizeof(float);
  // the attribute is set ONCE per instantiation, to the largest size any beam <= WM of it can ask for (it used to be
  // the size of whichever beam width came first - a beam-5 engine followed by a beam-10 one in the same process), thread-safe
  static std::once_flag attr_once;
  std::call_once(attr_once, [&]() {
    const size_t lds_max = (size_t)dl_lds_floats(D, DK, WM, WM, SELF, HPW) * sizeof(float);
    if (lds_max > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&dec_layer_attn_kernel<D, DK, WM, SELF, UNR, FIRST, KVH, HPW, WH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
  });
  dec_layer_attn_kernel<D, DK, WM, SELF, UNR, FIRST, KVH, HPW, WH><<<grid, 256 * HPW, lds, st>>>(p);
}

template <int D, int DK, bool SELF, bool FIRST, bool KVH>
static int launch_dec_layer_kvh(const DecLayerArgs &p, hipStream_t st) {
  const sc_search &sb = p.sb;
  const int ns = sb.rowmap ? sb.n_rows / sb.W : sb.S;   // streams of the compaction bucket only
  // compaction bucket of this launch (scasr.h: rowmap / n_rows): at most half of the streams active
  bool deep = (sb.rowmap && 2 * sb.n_rows <= sb.S * sb.W) || sb.S * sb.H <= 256;
  if (const char *fd = sc_hook("SC_ATTN_DEEP")) deep = atoi(fd) != 0;   // tests: force either variant at any size
  const int hpw = sc_dec_layer_hpw(sb);
  // fp16 decoder mode (sc_search.act_half; needs the fp16 K|V caches): the *_pph fragments are in p.wp / p.wop
  if constexpr (KVH) {
    if (sb.act_half & 1) {
      if constexpr (D == 256 && DK == 32) {
        if (hpw == 4) { launch_dec_layer_variant<D, DK, 10, SELF, 2, FIRST, KVH, 4, true>(p, ns, st); SC_CHECK_LAUNCH(); return SC_OK; }
      }
      if (sb.W <= 5) launch_dec_layer_variant<D, DK, 5, SELF, 4, FIRST, KVH, 1, true>(p, ns, st);
      else if (sb.W <= 10) {
        if (deep) launch_dec_layer_variant<D, DK, 10, SELF, 8, FIRST, KVH, 1, true>(p, ns, st);
        else launch_dec_layer_variant<D, DK, 10, SELF, 2, FIRST, KVH, 1, true>(p, ns, st);
      } else launch_dec_layer_variant<D, DK, 16, SELF, 2, FIRST, KVH, 1, true>(p, ns, st);
      SC_CHECK_LAUNCH();
      return SC_OK;
    }
  }
  if constexpr (D == 256 && DK == 32) {
    if (hpw == 4) { launch_dec_layer_variant<D, DK, 10, SELF, 2, FIRST, KVH, 4>(p, ns, st); SC_CHECK_LAUNCH(); return SC_OK; }
  }
  if (sb.W <= 5) {
    launch_dec_layer_variant<D, DK, 5, SELF, 4, FIRST, KVH, 1>(p, ns, st);
  } else if (sb.W <= 10) {
    if (deep) launch_dec_layer_variant<D, DK, 10, SELF, 8, FIRST, KVH, 1>(p, ns, st);
    else launch_dec_layer_variant<D, DK, 10, SELF, 2, FIRST, KVH, 1>(p, ns, st);
  } else {
    launch_dec_layer_variant<D, DK, 16, SELF, 2, FIRST, KVH, 1>(p, ns, st);
  }
  SC_CHECK_LAUNCH();
  return SC_OK;
}

template <int D, int DK, bool SELF, bool FIRST>
static int launch_dec_layer(const DecLayerArgs &p, hipStream_t st) {
  return p.sb.kv_half ? launch_dec_layer_kvh<D, DK, SELF, FIRST, true>(p, st) : launch_dec_layer_kvh<D, DK, SELF, FIRST, false>(p, st);
}

template <bool SELF, bool FIRST>
static int launch_dec_layer_dims(const DecLayerArgs &p, hipStream_t st) {
  const int d = p.sb.d, dk = p.sb.d / p.sb.H;
  if (d == 256 && dk == 32) return launch_dec_layer<256, 32, SELF, FIRST>(p, st);
  if (d == 256 && dk == 64) return launch_dec_layer<256, 64, SELF, FIRST>(p, st);
  if (d == 256 && dk == 16) return launch_dec_layer<256, 16, SELF, FIRST>(p, st);
  if (d == 128 && dk == 32) return launch_dec_layer<128, 32, SELF, FIRST>(p, st);
  if (d == 128 && dk == 16) return launch_dec_layer<128, 16, SELF, FIRST>(p, st);
  sc_set_error("sc_dec_layer_*: unsupported dimensions d=%d head dim %d", d, dk);
  return SC_ERR_ARG;
}

extern "C" int sc_dec_layer_fused_supported(int d, int H, int W, int F) {
  if (H <= 0 || d % H) return 0;
  const int dk = d / H;
  return ((d == 256 && (dk == 64 || dk == 32 || dk == 16)) || (d == 128 && (dk == 32 || dk == 16))) && W >= 1 && W <= 16 &&
         sc_ffn_ln_supported(d, F);
}

extern "C" int sc_dec_layer_self(const sc_search *sbp, int layer, const float *xin, float *xout, const float *ffn_part,
                                 int n_ffn_part, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xout && sbp->ph1, "null");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  SC_CHECK_ARG(sc_dec_layer_fused_supported(sb.d, sb.H, sb.W, sb.F), "unsupported dimensions");
  SC_CHECK_ARG(layer == 0 || (xin && xin != xout && ffn_part && n_ffn_part > 0), "x_in / partial sums missing");
  const sc_dec_layer &w = sb.layers[layer];
  SC_CHECK_ARG(w.wqkv_pp && w.wo_pp, "panel-packed Wqkv / Wo missing");
  SC_CHECK_ARG(!sb.act_half || (sb.kv_half && w.wqkv_pph && w.wo_pph), "fp16 decoder mode needs fp16 K|V caches and the *_pph weights");
  const bool wh = (sb.act_half & 1) != 0;
  DecLayerArgs p{sb, layer, xin, xout, ffn_part, n_ffn_part, 0, (long)sb.S * sb.W * sb.d, (long)sb.d,
                 layer > 0 ? sb.layers[layer - 1].b2 : nullptr, w.ln1_g, w.ln1_b,
                 wh ? (const float *)w.wqkv_pph : w.wqkv_pp, w.bqkv, wh ? (const float *)w.wo_pph : w.wo_pp, sb.ph1, sc_phase_take(0)};
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  const int rc = layer == 0 ? launch_dec_layer_dims<true, true>(p, st) : launch_dec_layer_dims<true, false>(p, st);
  {   // MFMA part: Q|K|V projection + out-projection of the bucket's rows (the attention itself depends on device state)
    const double M = sb.rowmap ? sb.n_rows : sb.S * sb.W;
    // algorithmic bytes without the K|V rows (bench.py adds them from the device counter): Wqkv + Wo once, x in, x out
    sc_prof_end(prof, SC_PROF_LAYER_SELF, 8.0 * M * sb.d * sb.d, 4.0 * (4.0 * sb.d * sb.d + 2.0 * M * sb.d));
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// (round 6) launch B2 of the four-head form at large buckets: the cross-attention's output projection summed ONCE per row -
//   x'' = x' + ((p0 + p1) + bo2) -> xout;   xn = LayerNorm3(x'') -> xn_out
// so that the feed-forward runs WITHOUT its prologue (sc_dec_layer_ffn_xn, gemm.hip).  ffn_fused_kernel<PRO> repeats this sum
// and the LayerNorm in each of the 8 chunk-group workgroups of a row tile (144 KB of residual + partials per workgroup, 31 MB
// through the fabric per launch, 8.5 of its 36 us at a full bucket); here 20 workgroups of 64 rows read every element once.
// Same arithmetic, element for element: the sum in the canonical order of the head partials (common.h), the LayerNorm with 16
// lanes per row and the summation order of the layer kernels / ffn_fused_kernel<PRO> - the bits do not depend on which of
// the two paths a bucket takes.
__global__ __launch_bounds__(256) void dec_head_reduce_ln_kernel(const float *__restrict__ ph, int nph, int pgrp, const float *__restrict__ pbias,
                                                                 const float *__restrict__ xin, float *__restrict__ xout, float *__restrict__ xn,
                                                                 const int *__restrict__ rows, int M, const float *__restrict__ g,
                                                                 const float *__restrict__ be, float eps) {
  constexpr int D = 256, Q4 = D / 64;
  const int m = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  const long row = rows ? rows[min(m, M - 1)] : min(m, M - 1);
  float4 x[Q4];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < Q4; ++q) {
    const int c4 = sub + 16 * q;
    const float4 xi = *reinterpret_cast<const float4 *>(xin + row * D + 4 * c4);
    float4 pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pbias) pb = *reinterpret_cast<const float4 *>(pbias + 4 * c4);
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pgrp == 1 && nph == 2) {
      y = sc_add4(*reinterpret_cast<const float4 *>(ph + (row * 2) * D + 4 * c4), *reinterpret_cast<const float4 *>(ph + (row * 2 + 1) * D + 4 * c4));
    } else {
      for (int z0 = 0; z0 < nph; z0 += 8) {
        float4 pv[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          pv[z] = *reinterpret_cast<const float4 *>(ph + (row * nph + min(z0 + z, nph - 1)) * D + 4 * c4);
          if (z0 + z >= nph) pv[z] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (pgrp == 4) {
          const float4 g0 = sc_seq4(pv[0], pv[1], pv[2], pv[3]);
          y = z0 == 0 ? g0 : sc_add4(y, g0);
          if (z0 + 4 < nph) y = sc_add4(y, sc_seq4(pv[4], pv[5], pv[6], pv[7]));
        } else {
#pragma unroll
          for (int z = 0; z < 8; ++z)
            if (z0 + z < nph) y = (z0 + z == 0) ? pv[0] : sc_add4(y, pv[z]);
        }
      }
    }
    x[q] = make_float4(xi.x + (y.x + pb.x), xi.y + (y.y + pb.y), xi.z + (y.z + pb.z), xi.w + (y.w + pb.w));
    if (m < M) *reinterpret_cast<float4 *>(xout + row * D + 4 * c4) = x[q];
    sum += (x[q].x + x[q].y) + (x[q].z + x[q].w);
  }
  const float mean = group_sum<16>(sum) / (float)D;
  float q2 = 0.f;
#pragma unroll
  for (int q = 0; q < Q4; ++q) {
    const float a = x[q].x - mean, b = x[q].y - mean, c = x[q].z - mean, e = x[q].w - mean;
    q2 += (a * a + b * b) + (c * c + e * e);
  }
  const float rstd = 1.0f / sqrtf(group_sum<16>(q2) / (float)D + eps);
  if (m < M) {
#pragma unroll
    for (int q = 0; q < Q4; ++q) {
      const float4 gm = *reinterpret_cast<const float4 *>(g + 4 * (sub + 16 * q));
      const float4 bt = *reinterpret_cast<const float4 *>(be + 4 * (sub + 16 * q));
      *reinterpret_cast<float4 *>(xn + row * D + 4 * (sub + 16 * q)) =
          make_float4((x[q].x - mean) * rstd * gm.x + bt.x, (x[q].y - mean) * rstd * gm.y + bt.y,
                      (x[q].z - mean) * rstd * gm.z + bt.z, (x[q].w - mean) * rstd * gm.w + bt.w);
    }
  }
}

// does this bucket's four-head form sum the cross-attention's partial products in a launch of its own (then: feed-forward
// without prologue)?  d = 256, fp32 partial products, four heads per workgroup.  SC_DEC_FFN_SPLIT = 0 | 1 (test hook).
// MEASURED AND NOT TAKEN by default (profiles/r06_ab_split_ffn.txt): the feed-forward loses its 8.5 us prologue (36 -> ~30 us at
// a full bucket, 27.5 -> 24.5 at 84 streams), but the launch that replaces it costs 5.5 us - a kernel boundary plus two dependent
// round trips of a 50-workgroup kernel: 3364 / 3345 against 3373 / 3376 audio-s/s.  Kept as an A/B hook (same bits either way).
int sc_dec_layer_split_ffn(const sc_search &sb) {
  const char *e = sc_hook("SC_DEC_FFN_SPLIT");
  return e && atoi(e) != 0 && sb.d == 256 && !sb.act_half && sb.dq && sc_dec_layer_hpw(sb) > 1;
}

extern "C" int sc_dec_layer_reduce_ln(const sc_search *sbp, int layer, const float *xin, float *xout, float *xn_out, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xin && xout && xn_out && xin != xout && sbp->ph2, "null / aliased");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  SC_CHECK_ARG(sb.d == 256 && !(sb.act_half & 2), "d = 256, fp32 partial products");
  const sc_dec_layer &w = sb.layers[layer];
  const int M = sb.rowmap ? sb.n_rows : sb.S * sb.W;
  const int nph = sb.H / sc_dec_layer_hpw(sb);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  dec_head_reduce_ln_kernel<<<cdiv(M, 16), 256, 0, st>>>(sb.ph2, nph, nph == sb.H ? 4 : 1, w.bo2, xin, xout, xn_out, sb.rowmap, M, w.ln3_g,
                                                        w.ln3_b, sb.ln_eps);
  SC_CHECK_LAUNCH();
  sc_prof_end(prof, SC_PROF_PROJ_LN_PROJ, 0.0, 4.0 * (double)M * sb.d * (3 + nph));
  return SC_OK;
}

extern "C" int sc_dec_layer_cross(const sc_search *sbp, int layer, const float *xin, float *xout, void *stream) {
  SC_CHECK_ARG(sbp && sbp->layers && xin && xout && xin != xout && sbp->ph1 && sbp->ph2, "null / aliased");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(layer >= 0 && layer < sb.n_layers, "layer out of range");
  SC_CHECK_ARG(sc_dec_layer_fused_supported(sb.d, sb.H, sb.W, sb.F), "unsupported dimensions");
  const sc_dec_layer &w = sb.layers[layer];
  SC_CHECK_ARG(w.wq_pp && w.wo2_pp, "panel-packed Wq / Wo2 missing");
  const int nph = sb.H / sc_dec_layer_hpw(sb);   // partial products per row left by sc_dec_layer_self (same bucket, same form)
  SC_CHECK_ARG(!sb.act_half || (sb.kv_half && w.wq_pph && w.wo2_pph), "fp16 decoder mode needs fp16 K|V caches and the *_pph weights");
  const bool wh = (sb.act_half & 1) != 0;
  DecLayerArgs p{sb, layer, xin, xout, sb.ph1, nph, nph == sb.H ? 4 : 1, (long)sb.d, (long)nph * sb.d, w.bo, w.ln2_g, w.ln2_b,
                 wh ? (const float *)w.wq_pph : w.wq_pp, w.bq, wh ? (const float *)w.wo2_pph : w.wo2_pp, sb.ph2, sc_phase_take(1)};
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof = sc_prof_begin(st);
  const int rc = launch_dec_layer_dims<false, false>(p, st);
  {   // MFMA part: q projection + out-projection; K|V bytes = sum_s T_s * 2d * 4: the host knows T (bench.py)
    const double M = sb.rowmap ? sb.n_rows : sb.S * sb.W;
    sc_prof_end(prof, SC_PROF_LAYER_CROSS, 4.0 * M * sb.d * sb.d, 4.0 * (2.0 * sb.d * sb.d + 2.0 * M * sb.d));
  }
  return rc;
}
