// Persistent "stream cluster" decoder of libscasr (gfx950): ALL decoder layers of one beam-search step in ONE launch
// for small compaction buckets.
//
// When few streams are still in the step loop (the straggler tail of a lock-step batch, or a single real-time
// stream) a decode iteration is a chain of dependent launches, and each launch is a chain of ~1 us memory hops:
// 48-92 launches x 7-19 us = 0.85 ms, with more than 95 % of the chip idle (profiles/r02_timeline_small_bucket_*).
// Here the H workgroups of a stream (one per attention head, one CU each) stay resident for the whole step and walk
// the layers together; what a kernel boundary used to order is ordered by a CLUSTER BARRIER of those H workgroups:
//
//   per layer:  A  x += b2' + sum_h PF[h]   (layer 0: embedding); norm1; q|k|v of the head; K|V append; self-attention;
//                  PH1[h] = ctx_h . Wo[:, head slice]^T                                            -> barrier
//               B  x += bo + sum_h PH1[h]; norm2; q of the head; cross-attention; PH2[h] = ... Wo2 -> barrier
//               C  x += bo2 + sum_h PH2[h]; norm3; feed-forward over the head's 1/H slice of the hidden units
//                  (W1 rows / W2 columns [h*F/H, (h+1)*F/H)); PF[h] = partial W2 product           -> barrier
//   tail:       head 0 stores x; sc_dec_output_logits adds b2 + sum_h PF[h], after_norm, output layer (next launch).
//
// Every workgroup keeps its own copy of the stream's residual rows x (W <= 16 rows x d) in LDS - the row-local
// arithmetic (partial-sum reduce, LayerNorm) is recomputed by all H workgroups, only the per-head partial products
// travel.  They travel through memory with 16-byte SYSTEM-scope accesses (global_store/load_dwordx4 sc0 sc1:
// write-through stores, cache-bypassing loads), and the barrier is one system-scope counter per stream - correct for
// any workgroup placement, no L2 write-back / invalidate fences (an agent-scope release costs 2-6 us on this part).
// Measured (tools/probes/cluster_barrier.hip): barrier + 80 KB all-to-all = 2.5-2.9 us for 1-8 clusters, 6 us at
// 32, 26 us at 128 - which is why this form is used for buckets of at most SC_CLUSTER_MAX_STREAMS streams.
//
// All H workgroups of a cluster must become resident for the barriers to complete: the grid is at most
// SC_CLUSTER_MAX_STREAMS x H <= 256 workgroups of 256 threads, one per CU; every spin is bounded (a stuck cluster
// sets sb.flags-independent err word and falls through instead of hanging the device).
//
// reference semantics: speechcatcher/model/decoder/decoder_layer.py:80-132, transformer_decoder.py:231-251.
#include "common.h"
#include "attn.h"

#define CTRL(s, f) sb.ctrl[(s) * 8 + (f)]
#define YSEQ(pp, s, h) (sb.yseq + (((long)(pp) * sb.S + (s)) * sb.W + (h)) * sb.LCAP)
#define ANC(pp, s) (sb.anc + ((long)(pp) * sb.S + (s)) * sb.LCAP * sb.W)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 dc_mfma8(f32x4 acc, const float4 &a0, const float4 &a1, const float4 &b0,
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

// 16-byte system-scope accesses (valid for any workgroup placement: the stores write through, the loads bypass the
// caches that another XCD's stores cannot reach)
__device__ __forceinline__ void st4_sys(float *p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld4_sys_issue(f32x4 &v, const float *p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void vm_wait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct DecClusterArgs {
  sc_search sb;
  float *xout;        // [S*W][d]: the residual stream after the last layer's cross-attention block (head 0 stores it)
  unsigned *cbar;     // [S] cluster barrier counters, zeroed before the launch
  int *err;           // set to 1 if a barrier timed out
  long long *dbg;     // profiling aid (SC_CLUSTER_DBG): shader-clock stamps of workgroup (first stream, head 0), or NULL
};

template <int D, int DK, int WM, int UNR, bool KVH>
__global__ __launch_bounds__(256, 1) void dec_cluster_kernel(DecClusterArgs p) {
  constexpr int LPR = DK / 4, NG = 256 / LPR, PCH = 128, GPR = 16 / LPR, NPART = NG / GPR;
  constexpr int EL = D / 64, LDX = D + 4, KI = D / 32, KPW = KI / 4, C4 = D / 4;
  constexpr int NTQ = DK / 16;
  static_assert(KI % 4 == 0 && NPART == 16 && DK <= 32, "d_model a multiple of 128, head dim 16 or 32");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sc_search &sb = p.sb;
  const int s = sb.rowmap ? sb.rowmap[blockIdx.x * sb.W] / sb.W : blockIdx.x;
  const int head = blockIdx.y;
  if (!CTRL(s, SC_C_ACTIVE)) return;          // the whole cluster of an idle stream leaves: nobody waits for it
  const int nh = CTRL(s, SC_C_NHYP);
  if (nh <= 0) return;
  const int L = CTRL(s, SC_C_L), cur = CTRL(s, SC_C_CUR), T = CTRL(s, SC_C_T);
  const int W = sb.W, LCAP = sb.LCAP, H = sb.H, F = sb.F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  const int g = tid / LPR, cq = tid % LPR;
  const int FS = F / H;                        // hidden units of this workgroup's feed-forward slice
  // ---- LDS
  float *Xres = smem;                          // [16][LDX] residual rows of the stream (rows >= W zero)
  float *Xn = Xres + 16 * LDX;                 // [16][LDX] LayerNorm output = MFMA A operand
  float *region = Xn + 16 * LDX;               // projection partials / attention scratch / FFN hidden / staging
  constexpr int NT_MAX = 3 * NTQ;
  constexpr int PS_FLOATS = 4 * 16 * (NT_MAX * 16 + 4);
  const int red_floats = NPART * W * (DK + 2);
  const int attn_floats = red_floats + PCH * W + 8;
  constexpr int OUT_FLOATS = 16 * 36 + 16 * (D + 4);
  const int hid_floats = 16 * (FS + 4);
  int region_floats = PS_FLOATS > attn_floats ? PS_FLOATS : attn_floats;
  region_floats = region_floats > OUT_FLOATS ? region_floats : OUT_FLOATS;
  region_floats = region_floats > hid_floats ? region_floats : hid_floats;
  float *qs = region + region_floats;          // [WM][DK]
  float *kvn = qs + WM * DK;                   // [WM][2*DK]
  float *ctx = kvn + WM * 2 * DK;              // [WM][DK]

  int dbg_n = 0;
  auto stamp = [&]() {
    if (p.dbg && blockIdx.x == 0 && head == 0 && tid == 0 && dbg_n < 256) p.dbg[dbg_n++] = (long long)__builtin_amdgcn_s_memtime();
  };
  unsigned bar_target = 0;
  auto cluster_barrier = [&]() {
    // this workgroup's system-scope stores have been issued by all its threads: drain them, then arrive
    vm_wait0();
    __syncthreads();
    bar_target += (unsigned)H;
    if (tid == 0) {
      __hip_atomic_fetch_add(p.cbar + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      int spins = 0;
      while (__hip_atomic_load(p.cbar + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < bar_target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) {   // never in a healthy run: a peer did not arrive; do not hang the device
          *p.err = 1;
          break;
        }
      }
    }
    __syncthreads();
  };

  // x += bias + sum_{z<H} part[(row*H + z)*D + :]  (or ffn-style layout part[(z*NR + row)*D + :]); then LayerNorm -> Xn
  auto reduce_ln = [&](const float *part, bool by_head, const float *bias, const float *lg, const float *lb) {
    constexpr int QN = (WM * C4 + 255) / 256;
    const long NR = (long)sb.S * W;
    float gam[EL], bet[EL];   // in flight together with the partial sums
#pragma unroll
    for (int e = 0; e < EL; ++e) {
      gam[e] = lg[lane + 64 * e];
      bet[e] = lb[lane + 64 * e];
    }
    if (part) {
      f32x4 pv[QN][8];
      static_assert(QN <= 3, "register budget of the partial-sum batch");
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int e = tid + 256 * q, i = min(e / C4, W - 1), c4 = e % C4;
        const long row = (long)s * W + i;
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          const int zz = min(z, H - 1);
          const float *src = by_head ? part + (row * H + zz) * D + 4 * c4 : part + ((long)zz * NR + row) * D + 4 * c4;
          ld4_sys_issue(pv[q][z], src);
        }
      }
      vm_wait0();
#pragma unroll
      for (int q = 0; q < QN; ++q) {
        const int e = tid + 256 * q, i = e / C4, c4 = e % C4;
        if (i < W) {
          f32x4 y = pv[q][0];
#pragma unroll
          for (int z = 1; z < 8; ++z)
            if (z < H) y += pv[q][z];
          const float4 bb = *reinterpret_cast<const float4 *>(bias + 4 * c4);
          float4 x = *reinterpret_cast<float4 *>(Xres + i * LDX + 4 * c4);
          x.x += y[0] + bb.x; x.y += y[1] + bb.y; x.z += y[2] + bb.z; x.w += y[3] + bb.w;
          *reinterpret_cast<float4 *>(Xres + i * LDX + 4 * c4) = x;
        }
      }
      __syncthreads();
    }
    for (int i = wave; i < W; i += 4) {   // LayerNorm, one wave per row
      float x[EL];
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        x[e] = Xres[i * LDX + lane + 64 * e];
        sum += x[e];
      }
      const float mean = wave_sum(sum) / (float)D;
      float q2 = 0.f;
#pragma unroll
      for (int e = 0; e < EL; ++e) {
        const float c = x[e] - mean;
        q2 += c * c;
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + sb.ln_eps);
#pragma unroll
      for (int e = 0; e < EL; ++e) Xn[i * LDX + lane + 64 * e] = (x[e] - mean) * rstd * gam[e] + bet[e];
    }
    __syncthreads();
  };

  // NT tiles of 16 output columns of LN(x) . Wp^T: K = D split over the 4 waves, partial products -> region (Ps)
  auto project = [&](const float *wp, int nt, auto tile_of) {
    f32x4 acc[NT_MAX];
    float4 b0[KPW][NT_MAX], b1[KPW][NT_MAX];   // every weight fragment of this wave in flight before the first MFMA
#pragma unroll
    for (int kq = 0; kq < KPW; ++kq)
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t)
        if (t < nt) {
          const float4 *wq = reinterpret_cast<const float4 *>(wp) + ((long)tile_of(t) * KI + wave * KPW + kq) * 128 + lane;
          b0[kq][t] = wq[0];
          b1[kq][t] = wq[64];
        }
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kq = 0; kq < KPW; ++kq) {
      const int ki = wave * KPW + kq;
      const float *ab = Xn + r16 * LDX + ki * 32 + 8 * kk;
      const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t)
        if (t < nt) acc[t] = dc_mfma8(acc[t], a0, a1, b0[kq][t], b1[kq][t]);
    }
    const int ldp = nt * 16 + 4;
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t)
      if (t < nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) region[(wave * 16 + 4 * kk + j) * ldp + t * 16 + r16] = acc[t][j];
    __syncthreads();
  };

  // attention of this head for all hypotheses (single pass, online softmax) -> ctx; SELF walks the ancestor-indexed
  // cache + the new row in kvn, CROSS the stream's encoder K|V
  auto attention = [&](bool self, int li) {
    float *red_m = region, *red_l = red_m + NPART * W, *red_a = red_l + NPART * W;
    int *rows = (int *)(red_a + NPART * W * DK);
    int *wtot = rows + PCH * W;
    const long skv0 = ((long)s * sb.n_layers + li) * LCAP * W * 2 * D + head * DK;
    const long ckv0 = ((long)s * sb.n_layers + li) * sb.TCAP * 2 * D + head * DK;
    AttnState st[WM];
#pragma unroll
    for (int h = 0; h < WM; ++h) {
      st[h].m = -INFINITY;
      st[h].l = 0.f;
      st[h].a = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto process = [&](const float4 &k, const float4 &v, unsigned hm) {
#pragma unroll
      for (int h = 0; h < WM; ++h) {
        const float4 qh = *reinterpret_cast<const float4 *>(qs + h * DK + 4 * cq);
        float sdot = qh.x * k.x;
        sdot = fmaf(qh.y, k.y, sdot);
        sdot = fmaf(qh.z, k.z, sdot);
        sdot = fmaf(qh.w, k.w, sdot);
        sdot = group_sum<LPR>(sdot);
        if ((hm >> h) & 1u) {
          if (sdot > st[h].m) {
            const float corr = __expf(st[h].m - sdot);
            st[h].l *= corr;
            st[h].a.x *= corr; st[h].a.y *= corr; st[h].a.z *= corr; st[h].a.w *= corr;
            st[h].m = sdot;
          }
          const float pe = __expf(sdot - st[h].m);
          st[h].l += pe;
          st[h].a.x = fmaf(pe, v.x, st[h].a.x);
          st[h].a.y = fmaf(pe, v.y, st[h].a.y);
          st[h].a.z = fmaf(pe, v.z, st[h].a.z);
          st[h].a.w = fmaf(pe, v.w, st[h].a.w);
        }
      }
    };
    if (self) {
      const int *anc = ANC(cur, s);
      const int Lc = L - 1;
      for (int ch = 0; ch < cdiv(Lc, PCH); ++ch) {
        const int c0 = ch * PCH;
        // distinct (position, slot) rows of positions [c0, c0+PCH): entry = local position | slot << 8 | hyp bits << 12
        for (int e = tid; e < PCH * W; e += 256) rows[e] = 0;
        const int pp = c0 + tid;
        const bool live = tid < PCH && pp < Lc;
        int sl[WM];
#pragma unroll
        for (int h = 0; h < WM; ++h) sl[h] = anc[(long)(live ? pp : 0) * W + min(h, nh - 1)];
        unsigned mask = 0;
#pragma unroll
        for (int h = 0; h < WM; ++h)
          if (h < nh) mask |= 1u << sl[h];
        const int cnt = live ? __popc(mask) : 0;
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o, 64);
          if (lane >= o) incl += t;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        const int base = incl - cnt + (wave >= 1 ? wtot[0] : 0);
        const int U = wtot[0] + wtot[1];
        if (live) {
#pragma unroll
          for (int h = 0; h < WM; ++h)
            if (h < nh) {
              const int rank = __popc(mask & ((1u << sl[h]) - 1u));
              atomicOr(&rows[base + rank], tid | (sl[h] << 8) | (1 << (12 + h)));
            }
        }
        __syncthreads();
        for (int j0 = g; j0 < U; j0 += NG * UNR) {
          int e[UNR];
          float4 k[UNR], v[UNR];
#pragma unroll
          for (int i = 0; i < UNR; ++i) {
            e[i] = rows[min(j0 + i * NG, U - 1)];
            const long kp = skv0 + ((long)(c0 + (e[i] & 255)) * W + ((e[i] >> 8) & 15)) * 2 * D;
            k[i] = kv_load4<KVH>(sb.skv, kp + 4 * cq);
            v[i] = kv_load4<KVH>(sb.skv, kp + D + 4 * cq);
          }
#pragma unroll
          for (int i = 0; i < UNR; ++i)
            if (j0 + i * NG < U) process(k[i], v[i], (unsigned)e[i] >> 12);
        }
        __syncthreads();   // rows / wtot are rebuilt by the next chunk (or become the partial states)
      }
      if (g < nh) {   // the new token: hypothesis h attends to its own row only
        const float4 k = *reinterpret_cast<const float4 *>(kvn + g * 2 * DK + 4 * cq);
        const float4 v = *reinterpret_cast<const float4 *>(kvn + g * 2 * DK + DK + 4 * cq);
        process(k, v, 1u << g);
      }
    } else {
      const unsigned all = (1u << nh) - 1u;
      for (int j0 = g; j0 < T; j0 += NG * UNR) {
        float4 k[UNR], v[UNR];
#pragma unroll
        for (int i = 0; i < UNR; ++i) {
          const long kp = ckv0 + (long)min(j0 + i * NG, T - 1) * 2 * D;
          k[i] = kv_load4<KVH>(sb.ckv, kp + 4 * cq);
          v[i] = kv_load4<KVH>(sb.ckv, kp + D + 4 * cq);
        }
#pragma unroll
        for (int i = 0; i < UNR; ++i)
          if (j0 + i * NG < T) process(k[i], v[i], all);
      }
    }
#pragma unroll
    for (int h = 0; h < WM; ++h) {
      if (LPR == 4) attn_merge_dpp<SC_DPP_ROR4>(st[h]);
      attn_merge_dpp<SC_DPP_ROR8>(st[h]);
    }
    if ((g % GPR) == 0) {
      const int pp = g / GPR;
#pragma unroll
      for (int h = 0; h < WM; ++h)
        if (h < nh) {
          if (cq == 0) {
            red_m[pp * W + h] = st[h].m;
            red_l[pp * W + h] = st[h].l;
          }
          *reinterpret_cast<float4 *>(red_a + ((long)(pp * W + h)) * DK + 4 * cq) = st[h].a;
        }
    }
    __syncthreads();
    if (tid < nh * NPART) {
      const int h = tid / NPART, pp = tid % NPART;
      const float mp = red_m[pp * W + h];
      float M = mp;
      M = fmaxf(M, dpp_mov<SC_DPP_XOR1>(M));
      M = fmaxf(M, dpp_mov<SC_DPP_XOR2>(M));
      M = fmaxf(M, dpp_mov<SC_DPP_HALF_MIRROR>(M));
      M = fmaxf(M, dpp_mov<SC_DPP_ROW_MIRROR>(M));
      const float w = (mp == -INFINITY) ? 0.f : __expf(mp - M);
      float den = w * red_l[pp * W + h];
      den += dpp_mov<SC_DPP_XOR1>(den);
      den += dpp_mov<SC_DPP_XOR2>(den);
      den += dpp_mov<SC_DPP_HALF_MIRROR>(den);
      den += dpp_mov<SC_DPP_ROW_MIRROR>(den);
      red_m[pp * W + h] = w;
      if (pp == 0) red_l[h] = den;
    }
    __syncthreads();
    for (int e = tid; e < WM * DK; e += 256) {
      const int h = e / DK, c = e % DK;
      float o = 0.f;
      if (h < nh) {
        float num = 0.f;
#pragma unroll
        for (int pp = 0; pp < NPART; ++pp) num = fmaf(red_m[pp * W + h], red_a[((long)(pp * W + h)) * DK + c], num);
        o = num / red_l[h];
      }
      ctx[e] = o;
    }
    __syncthreads();
  };

  // out[row][head][:] = A . Wp[:, k-block(s) kb0..kb0+nkb)]^T for the 16 x (32*nkb) tile A in LDS (leading dim lda), system stores
  auto head_partial = [&](const float *A, int lda, int nkb, const float *wp, int kiw, int kb0, float *out, bool by_head) {
    constexpr int NTO = D / 16, TW = NTO / 4, LDO = D + 4;
    float *Os = region + 16 * 36;   // staging (the A tile of the attention output projection sits at region[0..16*36))
    f32x4 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 b0[2][TW], b1[2][TW];
    auto loadb = [&](int kb, int buf) {
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        const float4 *wq = reinterpret_cast<const float4 *>(wp) + ((long)(wave * TW + t) * kiw + kb0 + kb) * 128 + lane;
        b0[buf][t] = wq[0];
        b1[buf][t] = wq[64];
      }
    };
    loadb(0, 0);
    for (int kb = 0; kb < nkb; kb += 2) {   // two k-blocks per trip: the next block's fragments load during the MFMAs
      if (kb + 1 < nkb) loadb(kb + 1, 1);
      {
        const float *ab = A + r16 * lda + kb * 32 + 8 * kk;
        const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = dc_mfma8(acc[t], a0, a1, b0[0][t], b1[0][t]);
      }
      if (kb + 2 < nkb) loadb(kb + 2, 0);
      if (kb + 1 < nkb) {
        const float *ab = A + r16 * lda + (kb + 1) * 32 + 8 * kk;
        const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = dc_mfma8(acc[t], a0, a1, b0[1][t], b1[1][t]);
      }
    }
    __syncthreads();   // (the hidden tile / A tile may alias the staging area)
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) Os[(4 * kk + j) * LDO + (wave * TW + t) * 16 + r16] = acc[t][j];
    __syncthreads();
    const long NR = (long)sb.S * W;
    for (int e = tid; e < W * C4; e += 256) {
      const int w = e / C4, c4 = e % C4;
      const long row = (long)s * W + w;
      float *dst = by_head ? out + (row * H + head) * D + 4 * c4 : out + ((long)head * NR + row) * D + 4 * c4;
      const float4 v = *reinterpret_cast<const float4 *>(Os + w * LDO + 4 * c4);
      st4_sys(dst, f32x4{v.x, v.y, v.z, v.w});
    }
  };

  // ------------------------------------------------------------------ x of layer 0: embedding (transformer_decoder.py:231)
  for (int e = tid; e < 16 * C4; e += 256) {
    const int i = e / C4, c4 = e % C4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < W) {
      const int tok = YSEQ(cur, s, min(i, nh - 1))[L - 1];
      const float sq = sqrtf((float)D);
      const float4 ev = *reinterpret_cast<const float4 *>(sb.embed + (long)tok * D + 4 * c4);
      const float4 pe = *reinterpret_cast<const float4 *>(sb.pe + (long)(L - 1) * D + 4 * c4);
      v = make_float4(ev.x * sq + pe.x, ev.y * sq + pe.y, ev.z * sq + pe.z, ev.w * sq + pe.w);
    }
    *reinterpret_cast<float4 *>(Xres + i * LDX + 4 * c4) = v;
    *reinterpret_cast<float4 *>(Xn + i * LDX + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  const float scale = sqrtf((float)DK);
  for (int li = 0; li < sb.n_layers; ++li) {
    const sc_dec_layer &w = p.sb.layers[li];   // (device copy of the layer table: see launcher)
    // ============================== A: self-attention block
    stamp();   // 0
    reduce_ln(li == 0 ? nullptr : sb.ffn_part, false, li == 0 ? nullptr : p.sb.layers[li - 1].b2, w.ln1_g, w.ln1_b);
    stamp();   // 1 reduce+LN1
    project(w.wqkv_pp, 3 * NTQ, [&](int t) { return ((t / NTQ) * D + head * DK) / 16 + (t % NTQ); });
    stamp();   // 2 qkv projection
    {
      const int nt = 3 * NTQ, ldp = nt * 16 + 4;
      const long skv0 = ((long)s * sb.n_layers + li) * LCAP * W * 2 * D + head * DK;
      for (int e = tid; e < WM * nt * 16; e += 256) {
        const int ww = e / (nt * 16), n = e % (nt * 16);
        const int which = n / DK, c = n % DK;
        float v = 0.f;
        if (ww < W) {
          v = region[(0 * 16 + ww) * ldp + n];
          v += region[(1 * 16 + ww) * ldp + n];
          v += region[(2 * 16 + ww) * ldp + n];
          v += region[(3 * 16 + ww) * ldp + n];
          v += w.bqkv[which * D + head * DK + c];
        }
        if (which == 0) qs[ww * DK + c] = v / scale;
        else {
          kvn[ww * 2 * DK + (which - 1) * DK + c] = v;
          if (ww < nh) kv_store1<KVH>(sb.skv, skv0 + ((long)(L - 1) * W + ww) * 2 * D + (which - 1) * D + c, v);
        }
      }
      __syncthreads();
    }
    stamp();   // 3 qkv reduce
    attention(true, li);
    stamp();   // 4 self attention
    {
      const int kb = (head * DK) / 32, koff = (head * DK) % 32;
      for (int e = tid; e < 16 * 32; e += 256) {
        const int ww = e / 32, c = e % 32;
        region[ww * 36 + c] = (ww < WM && c >= koff && c < koff + DK) ? ctx[ww * DK + c - koff] : 0.f;
      }
      __syncthreads();
      head_partial(region, 36, 1, w.wo_pp, KI, kb, sb.ph1, true);
    }
    stamp();   // 5 out-projection partial
    cluster_barrier();
    stamp();   // 6 barrier A
    // ============================== B: cross-attention block
    reduce_ln(sb.ph1, true, w.bo, w.ln2_g, w.ln2_b);
    stamp();   // 7 reduce+LN2
    project(w.wq_pp, NTQ, [&](int t) { return head * DK / 16 + t; });
    {
      const int nt = NTQ, ldp = nt * 16 + 4;
      for (int e = tid; e < WM * nt * 16; e += 256) {
        const int ww = e / (nt * 16), n = e % (nt * 16);
        float v = 0.f;
        if (ww < W) {
          v = region[(0 * 16 + ww) * ldp + n];
          v += region[(1 * 16 + ww) * ldp + n];
          v += region[(2 * 16 + ww) * ldp + n];
          v += region[(3 * 16 + ww) * ldp + n];
          v += w.bq[head * DK + n];
        }
        qs[ww * DK + n] = v / scale;
      }
      __syncthreads();
    }
    stamp();   // 8 q projection
    attention(false, li);
    stamp();   // 9 cross attention
    {
      const int kb = (head * DK) / 32, koff = (head * DK) % 32;
      for (int e = tid; e < 16 * 32; e += 256) {
        const int ww = e / 32, c = e % 32;
        region[ww * 36 + c] = (ww < WM && c >= koff && c < koff + DK) ? ctx[ww * DK + c - koff] : 0.f;
      }
      __syncthreads();
      head_partial(region, 36, 1, w.wo2_pp, KI, kb, sb.ph2, true);
    }
    cluster_barrier();
    stamp();   // 10 out-projection 2 + barrier B
    // ============================== C: feed-forward over this workgroup's slice of the hidden units
    reduce_ln(sb.ph2, true, w.bo2, w.ln3_g, w.ln3_b);
    stamp();   // 11 reduce+LN3
    {
      // hidden[16][FS] = relu(LN3(x) . W1[slice]^T + b1[slice]); wave wv owns hidden tiles wv, wv+4, ...
      float *Hs = region;              // [16][FS+4]
      const int ldh = FS + 4, nth = FS / 16, tile0 = head * nth;
      float4 wa[2][KI], wb[2][KI];   // double-buffered: all k-blocks of a hidden tile, the next tile loading meanwhile
      auto loadt = [&](int tt, int buf) {
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) {
          const float4 *wq = reinterpret_cast<const float4 *>(w.w1_p) + ((long)(tile0 + tt) * KI + ki) * 128 + lane;
          wa[buf][ki] = wq[0];
          wb[buf][ki] = wq[64];
        }
      };
      auto tile = [&](int tt, int buf) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) {
          const float *ab = Xn + r16 * LDX + ki * 32 + 8 * kk;
          const float4 a0 = *reinterpret_cast<const float4 *>(ab), a1 = *reinterpret_cast<const float4 *>(ab + 4);
          acc = dc_mfma8(acc, a0, a1, wa[buf][ki], wb[buf][ki]);
        }
        const float bias = w.b1[(tile0 + tt) * 16 + r16];
#pragma unroll
        for (int j = 0; j < 4; ++j) Hs[(4 * kk + j) * ldh + tt * 16 + r16] = fmaxf(acc[j] + bias, 0.f);
      };
      if (wave < nth) loadt(wave, 0);
      for (int tt = wave; tt < nth; tt += 8) {
        if (tt + 4 < nth) loadt(tt + 4, 1);
        tile(tt, 0);
        if (tt + 8 < nth) loadt(tt + 8, 0);
        if (tt + 4 < nth) tile(tt + 4, 1);
      }
      __syncthreads();
      stamp();   // 12 FFN GEMM 1
      // PF[head] = hidden . W2[:, slice]^T: k-blocks head*FS/32 ... of W2's packed layout (F/32 k-blocks per tile)
      head_partial(Hs, ldh, FS / 32, w.w2_p, F / 32, head * (FS / 32), sb.ffn_part, false);
    }
    stamp();   // 13 FFN GEMM 2
    cluster_barrier();
    stamp();   // 14 barrier C
  }
  // ------------------------------------------------------------------ tail: x before the last feed-forward's residual
  if (head == 0)
    for (int e = tid; e < W * C4; e += 256) {
      const int i = e / C4, c4 = e % C4;
      *reinterpret_cast<float4 *>(p.xout + ((long)s * W + i) * D + 4 * c4) = *reinterpret_cast<const float4 *>(Xres + i * LDX + 4 * c4);
    }
}

// ---------------------------------------------------------------------------------------------------------------
#define SC_CLUSTER_MAX_STREAMS 16
static long long *g_cluster_dbg = nullptr;
// profiling aid: the shader-clock stamps of the last launch (SC_TEST_HOOKS=1 SC_CLUSTER_DBG=1); returns how many were copied
extern "C" int sc_dec_cluster_debug(long long *host_out, int n) {
  if (!g_cluster_dbg || !host_out || n <= 0) return 0;
  if (n > 256) n = 256;
  if (hipMemcpy(host_out, g_cluster_dbg, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  return n;
}

extern "C" int sc_dec_cluster_supported(int d, int H, int W, int F) {
  if (H <= 0 || d % H || F % H) return 0;
  const int dk = d / H, fs = F / H;
  return (d == 256 || d == 128) && (dk == 32 || dk == 16) && W >= 1 && W <= 10 && H <= 8 && fs % 32 == 0 &&
         sc_ffn_ln_supported(d, F);
}

template <int D, int DK>
static size_t cluster_lds(const sc_search &sb, int wm) {
  const int W = sb.W, ntq = DK / 16, fs = sb.F / sb.H;
  const int ps = 4 * 16 * (3 * ntq * 16 + 4), attn = 16 * W * (DK + 2) + 128 * W + 8, outp = 16 * 36 + 16 * (D + 4),
            hid = 16 * (fs + 4);
  int region = ps > attn ? ps : attn;
  region = region > outp ? region : outp;
  region = region > hid ? region : hid;
  return (size_t)(2 * 16 * (D + 4) + region + wm * DK + wm * 2 * DK + wm * DK) * sizeof(float);
}

template <int D, int DK, int WM, bool KVH>
static int launch_cluster(const DecClusterArgs &p, hipStream_t st) {
  const sc_search &sb = p.sb;
  const size_t lds = cluster_lds<D, DK>(sb, WM);
  auto kern = dec_cluster_kernel<D, DK, WM, 8, KVH>;
  static bool attr_done = false;
  if (!attr_done && lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const dim3 grid(sb.rowmap ? sb.n_rows / sb.W : sb.S, sb.H);
  kern<<<grid, 256, lds, st>>>(p);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// All decoder layers of one step for the streams of the compaction bucket (at most SC_CLUSTER_MAX_STREAMS).
// dev_layers: DEVICE copy of sb->layers (the kernel walks the layers itself); cbar [S] and err [1] device words.
// Leaves x (before the last feed-forward's residual) in xout and the feed-forward partial sums of the last layer in
// sb->ffn_part[h], h < H: follow with sc_dec_output_logits(sb, xout, <other buffer>, sb->ffn_part, H).
extern "C" int sc_dec_cluster_layers(const sc_search *sbp, const sc_dec_layer *dev_layers, float *xout, unsigned *cbar,
                                     int *err, void *stream) {
  SC_CHECK_ARG(sbp && dev_layers && xout && cbar && err && sbp->ph1 && sbp->ph2 && sbp->ffn_part, "null");
  const sc_search &sb = *sbp;
  SC_CHECK_ARG(sc_dec_cluster_supported(sb.d, sb.H, sb.W, sb.F) && sb.max_ffn_part >= sb.H, "unsupported dimensions");
  const int nstreams = sb.rowmap ? sb.n_rows / sb.W : sb.S;
  SC_CHECK_ARG(nstreams >= 1 && nstreams <= SC_CLUSTER_MAX_STREAMS, "bucket too large for the cluster kernel");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(cbar, 0, (size_t)sb.S * sizeof(unsigned), st) != hipSuccess) {
    sc_set_error("sc_dec_cluster_layers: memset failed");
    return SC_ERR_LAUNCH;
  }
  static long long *dbg_dev = nullptr;
  if (sc_hook("SC_CLUSTER_DBG") && !dbg_dev) (void)hipMalloc((void **)&dbg_dev, 256 * sizeof(long long));
  g_cluster_dbg = dbg_dev;
  DecClusterArgs p{sb, xout, cbar, err, dbg_dev};
  p.sb.layers = dev_layers;
  const int dk = sb.d / sb.H;
#define SC_CL(DD, KK)                                                                              \
  if (sb.d == DD && dk == KK) {                                                                    \
    if (sb.W <= 5) return sb.kv_half ? launch_cluster<DD, KK, 5, true>(p, st) : launch_cluster<DD, KK, 5, false>(p, st); \
    return sb.kv_half ? launch_cluster<DD, KK, 10, true>(p, st) : launch_cluster<DD, KK, 10, false>(p, st);            \
  }
  SC_CL(256, 32) SC_CL(256, 16) SC_CL(128, 32) SC_CL(128, 16)
#undef SC_CL
  sc_set_error("sc_dec_cluster_layers: unsupported dimensions");
  return SC_ERR_ARG;
}

extern "C" int sc_dec_cluster_max_streams(void) { return SC_CLUSTER_MAX_STREAMS; }
