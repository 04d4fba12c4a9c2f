// Device helpers shared by the decoder attention kernels (search.hip, decoder_layer.hip):
struct AttnState {
  float m, l;
  float4 a;
};

template <int CTRL>
__device__ __forceinline__ void attn_merge_dpp(AttnState &st) {
  const float pm = dpp_mov<CTRL>(st.m), pl = dpp_mov<CTRL>(st.l);
  const float px = dpp_mov<CTRL>(st.a.x), py = dpp_mov<CTRL>(st.a.y);
  const float pz = dpp_mov<CTRL>(st.a.z), pw = dpp_mov<CTRL>(st.a.w);
  const float M = fmaxf(st.m, pm);
  const float ca = (st.m == -INFINITY) ? 0.f : __expf(st.m - M);
  const float cb = (pm == -INFINITY) ? 0.f : __expf(pm - M);
  st.m = M;
  st.l = st.l * ca + pl * cb;
  st.a.x = st.a.x * ca + px * cb;
  st.a.y = st.a.y * ca + py * cb;
  st.a.z = st.a.z * ca + pz * cb;
  st.a.w = st.a.w * ca + pw * cb;
}


// K|V caches in half precision (sc_search.kv_half: fp16 storage, fp32 arithmetic): element offsets are the same
// as for the fp32 caches, the elements are 2 bytes.  4 consecutive elements <-> one float4.
typedef _Float16 sc_half4 __attribute__((ext_vector_type(4)));
template <bool KVH>
__device__ __forceinline__ float4 kv_load4(const float *base, long elem) {
  if (KVH) {
    const sc_half4 h = *reinterpret_cast<const sc_half4 *>(reinterpret_cast<const _Float16 *>(base) + elem);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  return *reinterpret_cast<const float4 *>(base + elem);
}
template <bool KVH>
__device__ __forceinline__ void kv_store1(float *base, long elem, float v) {
  if (KVH) reinterpret_cast<_Float16 *>(base)[elem] = (_Float16)v;
  else base[elem] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core form of the decoder attention of one (stream, head) workgroup.  All (up to 16) hypotheses of the
// stream are the N dimension of v_mfma_f32_16x16x4_f32, 16 K/V rows ("keys") the M dimension:
//     S^T[key][hyp]  = K[key][:] . Q[hyp][:]          A = K rows (lane: key m, dims DPL*kg .. +DPL), B = Q
//     O^T[dim][hyp] += V[key][dim] * P[key][hyp]       A = V^T (lane: dims m*NDT .. +NDT of key 4*kg+j), B = P
// The accumulator layout of S^T (lane: hyp n = lane & 15, keys 4*kg + j) IS the B-operand layout of P, so the
// probabilities never leave their registers, and every per-hypothesis factor (running maximum, rescale) is
// lane-local.  The VALU form this replaces spent ~30 instructions per (key, hypothesis) on a 16-lane-wide SIMD
// (4 cycles per wave instruction): 4-7 us of a 14-19 us decoder layer launch, and the flash kernels of a full
// 128-stream bucket were VALU-bound at 2-3x their HBM time (tools/layer_phase_times.py).
// A wave walks the tiles wave, wave+4, ... NT tiles at a time (two passes per batch: scores and their maximum,
// then exponentials and P.V), so the online-softmax rescale happens once per batch.  Masked keys (not an
// ancestor of the hypothesis / beyond the end) get probability 0.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int DK>
struct MAttn {
  f32x4v o[DK / 16];   // O^T: o[dt][jj] = dim (4*kg + jj) * NDT + dt of hypothesis n
  float m, l;          // running maximum (same in the 4 lanes of a hypothesis), this lane's share of the sum
};
template <int DK>
__device__ __forceinline__ void mattn_init(MAttn<DK> &a) {
#pragma unroll
  for (int t = 0; t < DK / 16; ++t) a.o[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  a.m = -INFINITY;
  a.l = 0.f;
}

typedef _Float16 sc_half2 __attribute__((ext_vector_type(2)));
// SC_KV_NT (A/B switch, default 0): the K|V rows of a walk are read ONCE by ONE workgroup - with non-temporal loads they do
// not displace the weight fragments that all workgroups of an XCD share in its 4 MB L2
#ifndef SC_KV_NT
#define SC_KV_NT 0
#endif
template <class T>
__device__ __forceinline__ T sc_ld_stream(const T *p) {
#if SC_KV_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
typedef float sc_f32x2 __attribute__((ext_vector_type(2)));
template <int N, bool KVH>
__device__ __forceinline__ void kv_loadn(const float *base, long elem, float *out) {
  static_assert(N == 1 || N == 2 || N == 4, "1, 2 or 4 elements");
  if (KVH) {
    const _Float16 *h = reinterpret_cast<const _Float16 *>(base) + elem;
    if (N == 4) {
      const sc_half4 v = sc_ld_stream(reinterpret_cast<const sc_half4 *>(h));
      out[0] = (float)v.x; out[1] = (float)v.y; out[2] = (float)v.z; out[3] = (float)v.w;
    } else if (N == 2) {
      const sc_half2 v = sc_ld_stream(reinterpret_cast<const sc_half2 *>(h));
      out[0] = (float)v.x; out[1] = (float)v.y;
    } else {
      out[0] = (float)sc_ld_stream(h);
    }
  } else {
    if (N == 4) {
      const sc_f32x4 v = sc_ld_stream(reinterpret_cast<const sc_f32x4 *>(base + elem));
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else if (N == 2) {
      const sc_f32x2 v = sc_ld_stream(reinterpret_cast<const sc_f32x2 *>(base + elem));
      out[0] = v.x; out[1] = v.y;
    } else {
      out[0] = sc_ld_stream(base + elem);
    }
  }
}

// rowfn(idx, k_elem, hyp_mask): element offset of K row `idx` of the walk (V follows vofs elements later) and the
// bit set of hypotheses that attend to it; it must be safe for any idx (clamp) - tiles >= ntiles are masked here.
// qs: LDS [16][DK] queries / sqrt(dk), rows of unused hypotheses zero.
// The walk is split into its two halves so that a kernel can have the FIRST batch of a wave's tiles under way long before
// the queries exist (round 5: the K|V rows of the cached positions do not depend on this step's x - their loads travel
// while the projection's split sums are reduced):  mattn_load = the NT tiles wave, wave + 4, ... of a wave -> registers,
// mattn_batch = their arithmetic.
template <int DK, int NT>
struct MBatch {
  float kr[NT][DK / 4], vr[NT][4][DK / 16];
  unsigned hm[NT][4];
};
template <int DK, int NT, bool KVH, class RowFn>
__device__ __forceinline__ void mattn_load(MBatch<DK, NT> &b, const float *kv, int vofs, int ntiles, int t0, int lane, RowFn rowfn) {
  constexpr int DPL = DK / 4, NDT = DK / 16;
  const int n = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int t = t0 + 4 * i;
    long ke;
    unsigned unused;
    rowfn(16 * t + n, ke, unused);
#pragma unroll
    for (int q = 0; q < DPL / 4; ++q) kv_loadn<4, KVH>(kv, ke + DPL * kg + 4 * q, &b.kr[i][4 * q]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      long ve;
      rowfn(16 * t + 4 * kg + j, ve, b.hm[i][j]);
      kv_loadn<NDT, KVH>(kv, ve + vofs + n * NDT, b.vr[i][j]);
      if (t >= ntiles) b.hm[i][j] = 0u;
    }
  }
}
// CANONICAL BATCHES (bit-reproducible serving): the online-softmax rescale happens per batch of TWO tiles (t, t + 4) in
// every form of the kernels - a form that has NT = 4 tiles in flight (few streams active) works them off as two
// batches, exactly the batches the NT = 2 forms see.  A batch of masked tiles changes nothing (factor exp(0) = 1).
template <int DK, int NT>
__device__ __forceinline__ void mattn_batch(MAttn<DK> &st, const MBatch<DK, NT> &b, const float (&qb)[DK / 4], int lane) {
  constexpr int DPL = DK / 4, NDT = DK / 16;
  static_assert(NT % 2 == 0, "tiles in flight: whole canonical batches of two");
  const int n = lane & 15;
#pragma unroll
  for (int b0 = 0; b0 < NT; b0 += 2) {
    f32x4v s[2];
    float mloc = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) s[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // k-steps outermost: consecutive MFMAs go to different tiles' accumulators (a dependent accumulate waits 40
    // cycles, the pipe issues every 32)
#pragma unroll
    for (int q = 0; q < DPL; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i) s[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.kr[b0 + i][q], qb[q], s[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[i][j] = ((b.hm[b0 + i][j] >> n) & 1u) ? s[i][j] : -INFINITY;
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
    for (int dt = 0; dt < NDT; ++dt) st.o[dt] *= corr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = (s[i][j] == -INFINITY) ? 0.f : __expf(s[i][j] - muse);
        st.l += p;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) st.o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.vr[b0 + i][j][dt], p, st.o[dt], 0, 0, 0);
      }
    }
  }
}
// PRE: the wave's first batch (tiles wave, wave + 4, ... < 4 NT) was loaded by the caller (mattn_load with t0 = wave and the
// same ntiles / rowfn)
template <int DK, int NT, bool KVH, bool PRE = false, class RowFn>
__device__ __forceinline__ void mattn_walk(MAttn<DK> &st, const float *qs, const float *kv, int vofs, int ntiles,
                                           int wave, int lane, RowFn rowfn, const MBatch<DK, NT> *pre = nullptr) {
  constexpr int DPL = DK / 4;
  const int n = lane & 15, kg = lane >> 4;
  float qb[DPL];
#pragma unroll
  for (int i = 0; i < DPL; ++i) qb[i] = qs[n * DK + DPL * kg + i];
  int t0 = wave;
  if constexpr (PRE) {
    if (t0 < ntiles) mattn_batch<DK, NT>(st, *pre, qb, lane);
    t0 += 4 * NT;
  }
  for (; t0 < ntiles; t0 += 4 * NT) {
    MBatch<DK, NT> b;
    mattn_load<DK, NT, KVH>(b, kv, vofs, ntiles, t0, lane, rowfn);
    mattn_batch<DK, NT>(st, b, qb, lane);
  }
}

// partial state of a wave -> LDS slot `slot`: pm / pl [slots][16], pO [slots*16][DK+1]
template <int DK>
__device__ __forceinline__ void mattn_store_partial(const MAttn<DK> &st, float *pm, float *pl, float *pO, int slot, int lane) {
  constexpr int NDT = DK / 16, LDP = DK + 1;
  const int n = lane & 15, kg = lane >> 4;
  float l = st.l;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (kg == 0) {
    pm[slot * 16 + n] = st.m;
    pl[slot * 16 + n] = l;
  }
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) pO[(slot * 16 + n) * LDP + (4 * kg + jj) * NDT + dt] = st.o[dt][jj];
}

// context element (h, c) out of NP partial states (a hypothesis without any key: 0)
template <int DK, int NP>
__device__ __forceinline__ float mattn_final(const float *pm, const float *pl, const float *pO, int h, int c) {
  constexpr int LDP = DK + 1;
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < NP; ++w) M = fmaxf(M, pm[w * 16 + h]);
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int w = 0; w < NP; ++w) {
    const float mw = pm[w * 16 + h];
    const float g = (mw == -INFINITY) ? 0.f : __expf(mw - M);
    den = fmaf(g, pl[w * 16 + h], den);
    num = fmaf(g, pO[(w * 16 + h) * LDP + c], num);
  }
  return den > 0.f ? num / den : 0.f;
}
// LDS floats of the partial states
__host__ __device__ static inline int mattn_partial_floats(int DK, int np) { return np * 16 * (DK + 3); }

// ---------------------------------------------------------------------------------------------------------------
// Row list of the decoder self-attention for the positions [c0, c0 + 128) of one stream: the DISTINCT K|V pool rows the
// live hypotheses attend to there (ancestor table anc[p][h] = pool row; hypotheses of a beam share almost all of
// their history), each with the bit set of hypotheses that use it:   entry = pool row | hypothesis bit set << 16.
// Called by the 256 threads of a head group (gt = thread, wave = its wave in the group); thread p handles position
// c0 + p: duplicates among its <= WM rows are found by comparison in registers, a wave scan + the wave totals give
// every position its place in the list.  rw [128 * W] and wtot [8] are LDS of the group; contains __syncthreads()
// (all threads of the workgroup must call it the same number of times).  Returns the number of entries.
// PRE: the thread's ancestor rows were loaded earlier (`pre`, a register array: it is taken by reference and selected at
// compile time - a run-time pointer to it would force the array into scratch memory, +5 us per launch measured).
// NTHR threads cooperate (gt = index among them, wave = gt / 64) on PCH positions: 256 / 128 = the four waves of one head
// group and its own list; 1024 / 512 = a whole four-head workgroup building ONE list that its head groups share (the
// list depends on the stream only) - a 400-token hypothesis then costs one build phase (two barriers, one dependent
// round trip for the ancestor rows) instead of four, and the walk runs over all its tiles without draining in between.
// ofs / zero_n: a list can be built in two passes (one-head workgroups: 256 threads on the 512-position lists that every
// form of the layer kernels walks - the tiles, and with them the summation order, must not depend on the form): the first
// pass zeroes the whole list (zero_n entries), the second continues behind the first's `ofs` entries (zero_n = 0).
// Returns the number of entries in the list (ofs included).
template <int WM, bool PRE = false, int NTHR = 256, int PCH = 128>
__device__ __forceinline__ int mattn_build_rows(int *rw, int *wtot, const int *anc, int c0, int Lc, int W, int nh,
                                                int gt, int lane, int wave, const int (&pre)[WM], int ofs = 0, int zero_n = -1) {
  static_assert(PCH % 64 == 0 && PCH <= NTHR, "positions: whole waves of the cooperating threads");
  if (zero_n < 0) zero_n = PCH * W;
  for (int e = gt; e < zero_n; e += NTHR) rw[e] = 0;
  const int pp = c0 + gt;
  const bool live = gt < PCH && pp < Lc;
  int r[WM];
#pragma unroll
  for (int h = 0; h < WM; ++h) r[h] = PRE ? pre[h] : anc[(long)(live ? pp : 0) * W + min(h, nh - 1)];
  unsigned first = 0;   // hypotheses that are the first to name their row
  int rep[WM];          // ... and for every hypothesis the first one with the same row
#pragma unroll
  for (int h = 0; h < WM; ++h) {
    rep[h] = h;
#pragma unroll
    for (int h2 = h - 1; h2 >= 0; --h2) rep[h] = (r[h2] == r[h]) ? h2 : rep[h];
    if (h < nh && rep[h] == h) first |= 1u << h;
  }
  const int cnt = live ? __popc(first) : 0;
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63 && wave < PCH / 64) wtot[wave] = incl;
  __syncthreads();   // also orders the zero fill before the ORs
  int base = ofs + incl - cnt, U = ofs;
#pragma unroll
  for (int w2 = 0; w2 < PCH / 64; ++w2) {
    const int t = wtot[w2];
    if (w2 < wave) base += t;
    U += t;
  }
  if (live) {
#pragma unroll
    for (int h = 0; h < WM; ++h) {
      if (h < nh) {
        const int rank = __popc(first & ((1u << rep[h]) - 1u));
        atomicOr(&rw[base + rank], (r[h] & 0xFFFF) | (1 << (16 + h)));
      }
    }
  }
  __syncthreads();   // list complete; wtot may be rewritten
  return U;
}
