// Conformer building blocks named by the north star but not wired into any
// shipped speechcatcher model: stand-alone
// gfx950 kernels for the convolution module and the relative-position
// attention.  The pointwise convolutions / projections are sc_gemm calls.
#include "common.h"

// ---------------------------------------------------------------------------
// GLU -> depthwise Conv1d(k, same padding) -> BatchNorm1d(eval) -> Swish
//   reference: speechcatcher/model/layers/convolution.py:103-112
//   y [B][T][2C] (output of pointwise_conv1, channels-last) -> out [B][T][C]
// One workgroup per (batch, 64-frame tile, 64-channel tile): the GLU values of
// the tile plus its k-1 halo frames are staged in LDS once, the k taps then
// read them conflict-free (lanes = consecutive channels).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void glu_dwconv_bn_swish_kernel(
    const float *y, int B, int T, int C, int ksize, const float *dw_w, const float *dw_b,
    const float *bn_g, const float *bn_b, const float *bn_mean, const float *bn_var, float bn_eps,
    float *out) {
  extern __shared__ __attribute__((aligned(16))) float glu[];  // [(64 + ksize - 1)][64]
  const int pad = (ksize - 1) / 2;
  const int b = blockIdx.z, t0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int cl = threadIdx.x & 63, tq = threadIdx.x >> 6;  // 4 time rows per pass
  const int c = c0 + cl;
  const int rows = 64 + ksize - 1;
  for (int r = tq; r < rows; r += 4) {
    const int t = t0 + r - pad;
    float v = 0.f;  // zero padding outside [0, T)
    if (t >= 0 && t < T && c < C) {
      const float *yr = y + ((long)b * T + t) * 2 * C;
      const float a = yr[c], g = yr[C + c];
      v = a * (1.0f / (1.0f + expf(-g)));
    }
    glu[r * 64 + cl] = v;
  }
  __syncthreads();
  if (c >= C) return;
  const float scale = bn_g[c] / sqrtf(bn_var[c] + bn_eps);
  const float shift = bn_b[c] - bn_mean[c] * scale;
  const float bias = dw_b ? dw_b[c] : 0.f;
  for (int r = tq; r < 64; r += 4) {
    const int t = t0 + r;
    if (t >= T) break;
    float acc = 0.f;
    for (int k = 0; k < ksize; ++k) acc = fmaf(dw_w[c * ksize + k], glu[(r + k) * 64 + cl], acc);
    float v = (acc + bias) * scale + shift;
    v = v * (1.0f / (1.0f + expf(-v)));
    out[((long)b * T + t) * C + c] = v;
  }
}

extern "C" int sc_glu_dwconv_bn_swish(const float *y, int B, int T, int C, int ksize, const float *dw_w,
                                      const float *dw_b, const float *bn_g, const float *bn_b,
                                      const float *bn_mean, const float *bn_var, float bn_eps, float *out,
                                      void *stream) {
  SC_CHECK_ARG(y && dw_w && bn_g && bn_b && bn_mean && bn_var && out, "null pointer");
  SC_CHECK_ARG(ksize % 2 == 1 && ksize <= 129, "kernel size must be odd and <= 129");
  if (B <= 0 || T <= 0) return SC_OK;
  dim3 grid(cdiv(C, 64), cdiv(T, 64), B);
  size_t smem = (size_t)(64 + ksize - 1) * 64 * sizeof(float);
  glu_dwconv_bn_swish_kernel<<<grid, 256, smem, (hipStream_t)stream>>>(y, B, T, C, ksize, dw_w, dw_b, bn_g, bn_b,
                                                                        bn_mean, bn_var, bn_eps, out);
  SC_CHECK_LAUNCH();
  return SC_OK;
}

// ---------------------------------------------------------------------------
// Relative-position multi-head attention (Transformer-XL / ESPnet legacy form)
//   reference: speechcatcher/model/attention/multi_head_attention.py:300-378
//   scores[i][j] = ((q_i + u) . k_j + shift((q + v) . p^T)[i][j]) / sqrt(dk)
//   shift(x)[i][j] = padded[T + i*T + j], padded = x with a zero column in front
// One workgroup per (batch, head); Q, K, V, P of the head live in LDS
// (T <= 128), lane = query row, online softmax over the keys.
// ---------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(128) void relpos_attention_kernel(const float *qkv, const float *p,
                                                               const float *bias_u, const float *bias_v,
                                                               float *out, int T, int H, int d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Qs = smem, *Ks = Qs + T * DK, *Vs = Ks + T * DK, *Ps = Vs + T * DK;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const float *base = qkv + (long)b * T * 3 * d + h * DK;
  for (int e = threadIdx.x; e < T * DK; e += blockDim.x) {
    const int r = e / DK, c = e % DK;
    Qs[e] = base[(long)r * 3 * d + c];
    Ks[e] = base[(long)r * 3 * d + d + c];
    Vs[e] = base[(long)r * 3 * d + 2 * d + c];
    Ps[e] = p[(long)r * d + h * DK + c];
  }
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= T) return;
  float qu[DK], acc[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    qu[c] = Qs[i * DK + c] + bias_u[h * DK + c];
    acc[c] = 0.f;
  }
  const float scale = sqrtf((float)DK);
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < T; ++j) {
    float ac = 0.f;
#pragma unroll
    for (int c = 0; c < DK; ++c) ac = fmaf(qu[c], Ks[j * DK + c], ac);
    // rel_shift: element [i][j] of the shifted matrix
    const int f = T + i * T + j;
    const int jj = f % (T + 1);
    float bd = 0.f;
    if (jj != 0) {
      const int ii = f / (T + 1);
      const float *qr = Qs + ii * DK, *pr = Ps + (jj - 1) * DK;
#pragma unroll
      for (int c = 0; c < DK; ++c) bd = fmaf(qr[c] + bias_v[h * DK + c], pr[c], bd);
    }
    const float s = (ac + bd) / scale;
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn), pe = expf(s - mn);
    l = l * corr + pe;
#pragma unroll
    for (int c = 0; c < DK; ++c) acc[c] = acc[c] * corr + pe * Vs[j * DK + c];
    m = mn;
  }
  float *o = out + ((long)b * T + i) * d + h * DK;
#pragma unroll
  for (int c = 0; c < DK; ++c) o[c] = acc[c] / l;
}

// The same attention for any T and with the reference's masks (multi_head_attention.py:366-372: masked scores get
// the most negative float before the softmax and their probabilities are zeroed after it; a fully masked row gives
// zeros).  Grid (batch*head, query tiles of 128): lane = query row; K and V stream through LDS in tiles of 64 keys,
// Q and P stay in global memory (the rel_shift term of (i, j) reads Q row i or i+1 and P row j-i-2 or T+j-i-1:
// see the index algebra below).  mask_mode 0: none; 1: mask [B][T] over keys (the reference's (batch, 1, time_k));
// 2: mask [B][T][T].
template <int DK>
__global__ __launch_bounds__(128) void relpos_attention_tiled_kernel(const float *qkv, const float *p, const float *bias_u,
                                                                     const float *bias_v, float *out, int T, int H, int d,
                                                                     const uint8_t *mask, int mask_mode) {
  constexpr int KT = 64;
  __shared__ __attribute__((aligned(16))) float Ks[KT * DK], Vs[KT * DK];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const float *base = qkv + (long)b * T * 3 * d + h * DK;
  const int i = blockIdx.y * 128 + threadIdx.x;
  const bool live = i < T;
  const int ic = live ? i : T - 1;
  // shifted[i][j] = padded[T + i*T + j] with padded rows of T+1 columns (a zero in front):
  //   j >= i+1: row i+1, column j-i-1 (0 -> the zero pad, else P row j-i-2);   j <= i: row i, column T+j-i -> P row T+j-i-1
  float qu[DK], qv0[DK], qv1[DK], acc[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    const float q0 = base[(long)ic * 3 * d + c];
    const float q1 = base[(long)min(ic + 1, T - 1) * 3 * d + c];
    qu[c] = q0 + bias_u[h * DK + c];
    qv0[c] = q0 + bias_v[h * DK + c];
    qv1[c] = q1 + bias_v[h * DK + c];
    acc[c] = 0.f;
  }
  const float scale = sqrtf((float)DK);
  float m = -INFINITY, l = 0.f;
  for (int j0 = 0; j0 < T; j0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * DK; e += blockDim.x) {
      const int r = min(j0 + e / DK, T - 1), c = e % DK;
      Ks[e] = base[(long)r * 3 * d + d + c];
      Vs[e] = base[(long)r * 3 * d + 2 * d + c];
    }
    __syncthreads();
    if (!live) continue;
    const int jn = min(KT, T - j0);
    for (int jj = 0; jj < jn; ++jj) {
      const int j = j0 + jj;
      if (mask_mode == 1 && !mask[(long)b * T + j]) continue;
      if (mask_mode == 2 && !mask[((long)b * T + i) * T + j]) continue;
      float ac = 0.f;
#pragma unroll
      for (int c = 0; c < DK; ++c) ac = fmaf(qu[c], Ks[jj * DK + c], ac);
      float bd = 0.f;
      if (j != i + 1) {
        const int prow = j > i ? j - i - 2 : T + j - i - 1;
        const float *pr = p + (long)prow * d + h * DK;
        if (j > i) {
#pragma unroll
          for (int c = 0; c < DK; ++c) bd = fmaf(qv1[c], pr[c], bd);
        } else {
#pragma unroll
          for (int c = 0; c < DK; ++c) bd = fmaf(qv0[c], pr[c], bd);
        }
      }
      const float sc = (ac + bd) / scale;
      const float mn = fmaxf(m, sc);
      const float corr = expf(m - mn), pe = expf(sc - mn);
      l = l * corr + pe;
#pragma unroll
      for (int c = 0; c < DK; ++c) acc[c] = acc[c] * corr + pe * Vs[jj * DK + c];
      m = mn;
    }
  }
  if (!live) return;
  float *o = out + ((long)b * T + i) * d + h * DK;
#pragma unroll
  for (int c = 0; c < DK; ++c) o[c] = l > 0.f ? acc[c] / l : 0.f;   // every key masked: zeros, like the reference
}

extern "C" int sc_relpos_attention_masked(const float *qkv, const float *p, const float *bias_u, const float *bias_v,
                                          float *out, int B, int T, int H, int d, const uint8_t *mask, int mask_mode,
                                          void *stream) {
  SC_CHECK_ARG(qkv && p && bias_u && bias_v && out, "null pointer");
  SC_CHECK_ARG(T >= 1 && d % H == 0 && mask_mode >= 0 && mask_mode <= 2 && (mask_mode == 0 || mask), "bad arguments");
  if (B <= 0) return SC_OK;
  const int dk = d / H;
  hipStream_t st = (hipStream_t)stream;
  if (mask_mode == 0 && T <= 128) {   // everything of a head in LDS
    const size_t smem = (size_t)4 * T * dk * sizeof(float);
    if (dk == 32) relpos_attention_kernel<32><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
    else if (dk == 16) relpos_attention_kernel<16><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
    else if (dk == 64) relpos_attention_kernel<64><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
    else { sc_set_error("sc_relpos_attention: unsupported head dim %d", dk); return SC_ERR_ARG; }
  } else {
    const dim3 grid(B * H, cdiv(T, 128));
    if (dk == 32) relpos_attention_tiled_kernel<32><<<grid, 128, 0, st>>>(qkv, p, bias_u, bias_v, out, T, H, d, mask, mask_mode);
    else if (dk == 16) relpos_attention_tiled_kernel<16><<<grid, 128, 0, st>>>(qkv, p, bias_u, bias_v, out, T, H, d, mask, mask_mode);
    else if (dk == 64) relpos_attention_tiled_kernel<64><<<grid, 128, 0, st>>>(qkv, p, bias_u, bias_v, out, T, H, d, mask, mask_mode);
    else { sc_set_error("sc_relpos_attention: unsupported head dim %d", dk); return SC_ERR_ARG; }
  }
  SC_CHECK_LAUNCH();
  return SC_OK;
}

extern "C" int sc_relpos_attention(const float *qkv, const float *p, const float *bias_u, const float *bias_v,
                                   float *out, int B, int T, int H, int d, void *stream) {
  return sc_relpos_attention_masked(qkv, p, bias_u, bias_v, out, B, T, H, d, nullptr, 0, stream);
}
