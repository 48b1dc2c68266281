// fused_common.hpp — helpers shared by the whole-stage kernels (fused_phi / fused_rho / fused_gnn).
#pragma once
#include "common.hpp"

namespace sn {

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ f32x4 lds_ld4(const float* p) {
  float4 t = *reinterpret_cast<const float4*>(p);
  return f32x4{t.x, t.y, t.z, t.w};
}
__device__ __forceinline__ void lds_st4(float* p, f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Weight fragments are fetched with buffer loads: descriptor in SGPRs (wave-uniform base), one VGPR of
// per-lane offset (lane*16) and an immediate/SGPR fragment offset — no 64-bit address VGPR per fragment.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float* p, unsigned bytes) {
  unsigned long long a = reinterpret_cast<unsigned long long>(p);
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 wfrag(__amdgpu_buffer_rsrc_t rs, int voff, int frag) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, frag * 1024, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}


// Single-row-tile GEMM:  for every output tile ot < NTO, acc = W[ot-th 16 outputs] . in  and epi(ot, acc)
// consumes it.  W is packed [NTO][NTI][64][4].  Two accumulators (even / odd k-chunks) keep two independent
// MFMA chains in flight (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency).  The weight fragments are
// truly double buffered (A/B): tile ot+1's loads are issued before tile ot's MFMAs (pinned with sched_barrier).
template <int NTI>
__device__ __forceinline__ f32x4 mfma_tile(const float4 (&w)[NTI], const f32x4 (&in)[NTI]) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < NTI; kk += 2) {
    a0 = mfma16(w[kk].x, in[kk][0], a0);
    if (kk + 1 < NTI) a1 = mfma16(w[kk + 1].x, in[kk + 1][0], a1);
    a0 = mfma16(w[kk].y, in[kk][1], a0);
    if (kk + 1 < NTI) a1 = mfma16(w[kk + 1].y, in[kk + 1][1], a1);
    a0 = mfma16(w[kk].z, in[kk][2], a0);
    if (kk + 1 < NTI) a1 = mfma16(w[kk + 1].z, in[kk + 1][2], a1);
    a0 = mfma16(w[kk].w, in[kk][3], a0);
    if (kk + 1 < NTI) a1 = mfma16(w[kk + 1].w, in[kk + 1][3], a1);
  }
  return a0 + a1;
}

template <int NTI, int NTO, typename Epi>
__device__ __forceinline__ void gemm_rows2(const float* __restrict__ wp, const f32x4 (&in)[NTI], int lane, Epi epi) {
  const __amdgpu_buffer_rsrc_t rs = weight_rsrc(wp, NTO * NTI * 1024);
  const int voff = lane * 16;
  float4 wA[NTI], wB[NTI];
#pragma unroll
  for (int kk = 0; kk < NTI; ++kk) wA[kk] = wfrag(rs, voff, kk);
#pragma unroll
  for (int ot = 0; ot < NTO; ot += 2) {
    if (ot + 1 < NTO) {
#pragma unroll
      for (int kk = 0; kk < NTI; ++kk) wB[kk] = wfrag(rs, voff, (ot + 1) * NTI + kk);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi(ot, mfma_tile<NTI>(wA, in));
    __builtin_amdgcn_sched_barrier(0);
    if (ot + 1 < NTO) {
      if (ot + 2 < NTO) {
#pragma unroll
        for (int kk = 0; kk < NTI; ++kk) wA[kk] = wfrag(rs, voff, (ot + 2) * NTI + kk);
      }
      __builtin_amdgcn_sched_barrier(0);
      epi(ot + 1, mfma_tile<NTI>(wB, in));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int NT, typename Epi>
__device__ __forceinline__ void gemm_rows(const float* __restrict__ wp, const f32x4 (&in)[NT], int lane, Epi epi) {
  gemm_rows2<NT, NT>(wp, in, lane, epi);
}

// sum over the 4 lane groups (lanes l, l^16, l^32, l^48) that hold one activation row
__device__ __forceinline__ float row_allsum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

}  // namespace sn
