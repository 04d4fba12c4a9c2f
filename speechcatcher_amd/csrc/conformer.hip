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

extern "C" int sc_relpos_attention(const float *qkv, const float *p, const float *bias_u, const float *bias_v,
                                   float *out, int B, int T, int H, int d, void *stream) {
  SC_CHECK_ARG(qkv && p && bias_u && bias_v && out, "null pointer");
  SC_CHECK_ARG(T >= 1 && T <= 128, "T must be <= 128 (one workgroup holds Q, K, V, P of a head in LDS)");
  SC_CHECK_ARG(d % H == 0, "d % H");
  if (B <= 0) return SC_OK;
  const int dk = d / H;
  hipStream_t st = (hipStream_t)stream;
  size_t smem = (size_t)4 * T * dk * sizeof(float);
  if (dk == 32) relpos_attention_kernel<32><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
  else if (dk == 16) relpos_attention_kernel<16><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
  else if (dk == 64) relpos_attention_kernel<64><<<B * H, 128, smem, st>>>(qkv, p, bias_u, bias_v, out, T, H, d);
  else { sc_set_error("sc_relpos_attention: unsupported head dim %d", dk); return SC_ERR_ARG; }
  SC_CHECK_LAUNCH();
  return SC_OK;
}
