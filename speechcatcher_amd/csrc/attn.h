// Device helpers shared by the decoder attention kernels (search.hip, decoder_layer.hip):
// DPP lane permutations, row-group reductions and the online-softmax state of one hypothesis.
#pragma once
#include "common.h"

// DPP lane permutations (no LDS round trip, unlike ds_bpermute-based __shfl)
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
