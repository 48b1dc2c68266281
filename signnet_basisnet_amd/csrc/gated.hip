// gated.hip — GatedGCN's edge-gated aggregation and its adjoint.
//
// Replaces the DGL message passing of GatedGCNLayer.forward (GraphPrediction/layers/gatedgcn_layer.py:51-56):
//   apply_edges(u_add_v('Dh','Eh')) ; e = DEh + Ce ; sigma = sigmoid(e)
//   update_all(u_mul_e('Bh','sigma'), sum) ; update_all(copy_e('sigma'), sum) ; h = Ah + sum_sigma_h / (sum_sigma + 1e-6)
// in ONE pass over the destination-sorted CSR of the batch plan: a thread owns (destination node, channel), walks the
// node's in-edges (edge-id order -> deterministic sums), writes each edge's new feature row once and the node's new
// feature.  HBM-bound gather; algorithmic bytes 4*d*(3E + 4N) (Ce in, e out, Bh/Dh gathered per edge; Ah, Eh in, h, den out).
// The adjoint is two passes without atomics: by destination (d e, d Eh, d Ah) and by source over the reverse CSR (d Dh, d Bh).
#include "common.hpp"
#include <algorithm>

namespace sn {
namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ldn: row stride of the four node arrays (they may be column blocks of ONE [N, 4C] GEMM output).  Optional fused eval-mode
// epilogue (hs/ht, es/et = folded BatchNorm scale/shift of bn_node_h / bn_node_e; hres / eres = residual inputs):
//   h_out = [hres +] relu(hs*h + ht),  e_out = [eres +] relu(es*e + et)        (gatedgcn_layer.py:64-72)
struct GatedEpi { const float *hs, *ht, *es, *et, *hres, *eres; };
__global__ __launch_bounds__(256) void k_gated_fwd(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                   const float* __restrict__ Dh, const float* __restrict__ Eh, int ldn,
                                                   const float* __restrict__ Ce, int64_t N, int C,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ eperm, float* __restrict__ h_out,
                                                   float* __restrict__ e_out, float* __restrict__ den_out, GatedEpi ep) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t i = idx / C;
  const int c = (int)(idx - i * C);
  const float eh = Eh[i * ldn + c];
  const bool fuse = ep.hs != nullptr;
  const float es = fuse ? ep.es[c] : 1.f, et = fuse ? ep.et[c] : 0.f;
  float num = 0.f, den = 0.f;
  for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
    const int64_t j = col[s], e = eperm[s];
    const float en = Dh[j * ldn + c] + eh + Ce[e * C + c];
    const float sg = sigmoidf_(en);
    float eo = en;
    if (fuse) {
      eo = fmaxf(en * es + et, 0.f);
      if (ep.eres) eo += ep.eres[e * C + c];
    }
    e_out[e * C + c] = eo;
    num += Bh[j * ldn + c] * sg;
    den += sg;
  }
  float h = Ah[i * ldn + c] + num / (den + 1e-6f);
  if (fuse) {
    h = fmaxf(h * ep.hs[c] + ep.ht[c], 0.f);
    if (ep.hres) h += ep.hres[idx];
  }
  h_out[idx] = h;
  if (den_out) den_out[idx] = den;
}

// by destination: g = dh / (den+eps); d den = -dh * (h - Ah) / (den+eps)   [since num/(den+eps) = h - Ah]
//   d sigma_e = g * Bh_src + d den ;  d e_e = de_e + d sigma_e * sigma_e (1 - sigma_e) ;  d Eh_i = sum_in d e_e ;  d Ah = dh
__global__ __launch_bounds__(256) void k_gated_bwd_dst(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                       const float* __restrict__ e_new, const float* __restrict__ h_new,
                                                       const float* __restrict__ den, const float* __restrict__ dh,
                                                       const float* __restrict__ de, int64_t N, int C,
                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const int32_t* __restrict__ eperm, float* __restrict__ dE,
                                                       float* __restrict__ de_new /* [E,C]: d(Dh_src + Eh_dst + Ce) */,
                                                       float* __restrict__ gnum /* [N,C]: dh/(den+eps), for the source pass */) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t i = idx / C;
  const int c = (int)(idx - i * C);
  const float inv = 1.0f / (den[idx] + 1e-6f);
  const float g = dh[idx] * inv;
  const float dden = -g * (h_new[idx] - Ah[idx]);
  float acc = 0.f;
  for (int s = rowptr[i]; s < rowptr[i + 1]; ++s) {
    const int64_t j = col[s], e = eperm[s];
    const float sg = sigmoidf_(e_new[e * C + c]);
    const float dsg = g * Bh[j * C + c] + dden;
    const float d = (de ? de[e * C + c] : 0.f) + dsg * sg * (1.0f - sg);
    de_new[e * C + c] = d;
    acc += d;
  }
  dE[idx] = acc;
  gnum[idx] = g;
}
// by source over the reverse CSR (rcol = destination, rperm = edge id): d Dh_j = sum_out d e_e ; d Bh_j = sum_out sigma_e * g_dst
__global__ __launch_bounds__(256) void k_gated_bwd_src(const float* __restrict__ e_new, const float* __restrict__ de_new,
                                                       const float* __restrict__ gnum, int64_t N, int C,
                                                       const int32_t* __restrict__ rrow, const int32_t* __restrict__ rcol,
                                                       const int32_t* __restrict__ rperm, float* __restrict__ dD,
                                                       float* __restrict__ dB) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t j = idx / C;
  const int c = (int)(idx - j * C);
  float ad = 0.f, ab = 0.f;
  for (int s = rrow[j]; s < rrow[j + 1]; ++s) {
    const int64_t i = rcol[s], e = rperm[s];
    ad += de_new[e * C + c];
    ab += sigmoidf_(e_new[e * C + c]) * gnum[i * C + c];
  }
  dD[idx] = ad;
  dB[idx] = ab;
}

// The same pass with the recipe of the GIN / GINE gathers (ops.hip): a thread owns FOUR channels of a destination node (float4 rows),
// keeps four in-edges in flight (their ids, then 3 x 4 row loads, before any arithmetic) and the workgroups are ordered so that runs of
// consecutive nodes share an XCD (its L2 then serves the source rows of a graph's edges).  One edge and one float at a time the pass
// ran at 0.35-0.42 of HBM.  Element order of every sum unchanged: bit-identical to k_gated_fwd.
constexpr int GATED_KU = 4;
__global__ __launch_bounds__(256) void k_gated_fwd_v4(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                      const float* __restrict__ Dh, const float* __restrict__ Eh, int ldn,
                                                      const float* __restrict__ Ce, int64_t N, int C,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                      const int32_t* __restrict__ eperm, float* __restrict__ h_out,
                                                      float* __restrict__ e_out, float* __restrict__ den_out, GatedEpi ep, int xcd_chunk) {
  const int CV = C >> 2;
  const int64_t idx = xcd_remap(blockIdx.x, xcd_chunk) * 256 + threadIdx.x;
  if (idx >= N * CV) return;
  const int64_t i = idx / CV;
  const int c = 4 * (int)(idx - i * CV);
  auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  const int lo = rowptr[i], deg = rowptr[i + 1] - lo;
  const float4 eh = ld4(Eh + i * ldn + c), ah = ld4(Ah + i * ldn + c);
  const bool fuse = ep.hs != nullptr;
  float4 es = make_float4(1.f, 1.f, 1.f, 1.f), et = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fuse) { es = ld4(ep.es + c); et = ld4(ep.et + c); }
  float num[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < deg; k0 += GATED_KU) {
    int64_t j[GATED_KU], e[GATED_KU];
#pragma unroll
    for (int u = 0; u < GATED_KU; ++u) {
      j[u] = 0; e[u] = 0;
      if (k0 + u < deg) { j[u] = col[lo + k0 + u]; e[u] = eperm[lo + k0 + u]; }
    }
    float4 dv[GATED_KU], bv[GATED_KU], cv[GATED_KU], rv[GATED_KU];
#pragma unroll
    for (int u = 0; u < GATED_KU; ++u) {
      dv[u] = bv[u] = cv[u] = rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + u < deg) {
        dv[u] = ld4(Dh + j[u] * ldn + c); bv[u] = ld4(Bh + j[u] * ldn + c); cv[u] = ld4(Ce + e[u] * C + c);
        if (fuse && ep.eres) rv[u] = ld4(ep.eres + e[u] * C + c);
      }
    }
#pragma unroll
    for (int u = 0; u < GATED_KU; ++u) {
      if (k0 + u < deg) {
        const float d4[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w}, b4[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
        const float c4[4] = {cv[u].x, cv[u].y, cv[u].z, cv[u].w}, r4[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
        const float e4[4] = {eh.x, eh.y, eh.z, eh.w}, s4[4] = {es.x, es.y, es.z, es.w}, t4[4] = {et.x, et.y, et.z, et.w};
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float en = d4[r] + e4[r] + c4[r];
          const float sg = sigmoidf_(en);
          float eo = en;
          if (fuse) {
            eo = fmaxf(en * s4[r] + t4[r], 0.f);
            if (ep.eres) eo += r4[r];
          }
          o[r] = eo;
          num[r] += b4[r] * sg;
          den[r] += sg;
        }
        *reinterpret_cast<float4*>(e_out + e[u] * C + c) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  const float a4[4] = {ah.x, ah.y, ah.z, ah.w};
  float h[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = a4[r] + num[r] / (den[r] + 1e-6f);
  if (fuse) {
    const float4 hs = ld4(ep.hs + c), ht = ld4(ep.ht + c);
    const float hs4[4] = {hs.x, hs.y, hs.z, hs.w}, ht4[4] = {ht.x, ht.y, ht.z, ht.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r] * hs4[r] + ht4[r], 0.f);
    if (ep.hres) {
      const float4 hr = ld4(ep.hres + i * C + c);
      h[0] += hr.x; h[1] += hr.y; h[2] += hr.z; h[3] += hr.w;
    }
  }
  *reinterpret_cast<float4*>(h_out + i * C + c) = make_float4(h[0], h[1], h[2], h[3]);
  if (den_out) *reinterpret_cast<float4*>(den_out + i * C + c) = make_float4(den[0], den[1], den[2], den[3]);
}

}  // namespace
}  // namespace sn

using namespace sn;

extern "C" int sn_gated_aggregate_f32(const float* Ah, const float* Bh, const float* Dh, const float* Eh, int ldn, const float* Ce,
                                      int64_t N, int C, const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                      float* h_out, float* e_out, float* den_out, const float* h_scale, const float* h_shift,
                                      const float* e_scale, const float* e_shift, const float* h_res, const float* e_res,
                                      void* stream) {
  SN_REQUIRE(Ah && Bh && Dh && Eh && Ce && rowptr && col && eperm && h_out && e_out && N >= 0 && C > 0 && ldn >= C,
             "sn_gated_aggregate_f32: bad arguments");
  SN_REQUIRE((h_scale != nullptr) == (h_shift != nullptr) && (h_scale != nullptr) == (e_scale != nullptr) &&
             (e_scale != nullptr) == (e_shift != nullptr), "sn_gated_aggregate_f32: the fused epilogue needs all four scale/shift vectors");
  SN_REQUIRE(h_scale || (!h_res && !e_res), "sn_gated_aggregate_f32: residuals only with the fused epilogue");
  SN_REQUIRE(!h_scale || !den_out, "sn_gated_aggregate_f32: the fused (eval) epilogue and den_out (training) exclude each other");
  if (N == 0) return SN_OK;
  GatedEpi ep{h_scale, h_shift, e_scale, e_shift, h_res, e_res};
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (C % 4 == 0 && ldn % 4 == 0 && al(Ah) && al(Bh) && al(Dh) && al(Eh) && al(Ce) && al(h_out) && al(e_out) && al(den_out) && al(h_scale) &&
      al(h_shift) && al(e_scale) && al(e_shift) && al(h_res) && al(e_res)) {
    const int CV = C / 4;
    const int chunk = (int)std::max<int64_t>(1, (int64_t)(CV >= 16 ? 128 : 512) * CV / 256);       // ~128 (512) consecutive nodes per XCD run
    const int64_t round = (int64_t)8 * chunk;
    const int64_t nblk = cdiv(cdiv(N * CV, 256), round) * round;
    hipLaunchKernelGGL(k_gated_fwd_v4, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, Ah, Bh, Dh, Eh, ldn, Ce, N, C, rowptr, col,
                       eperm, h_out, e_out, den_out, ep, chunk);
    SN_CHECK_LAUNCH("sn_gated_aggregate_f32");
    return SN_OK;
  }
  hipLaunchKernelGGL(k_gated_fwd, dim3((unsigned)cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, Ah, Bh, Dh, Eh, ldn, Ce, N, C,
                     rowptr, col, eperm, h_out, e_out, den_out, ep);
  SN_CHECK_LAUNCH("sn_gated_aggregate_f32");
  return SN_OK;
}

extern "C" int sn_gated_aggregate_bwd_f32(const float* Ah, const float* Bh, const float* e_new, const float* h_new, const float* den,
                                          const float* dh, const float* de /* may be NULL */, int64_t N, int C,
                                          const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                                          const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm,
                                          float* dB, float* dD, float* dE, float* de_new, float* scratch /* [N*C] */, void* stream) {
  SN_REQUIRE(Ah && Bh && e_new && h_new && den && dh && rowptr && rev_rowptr && dB && dD && dE && de_new && scratch && N >= 0 && C > 0,
             "sn_gated_aggregate_bwd_f32: bad arguments");
  if (N == 0) return SN_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(N * C, 256)), blk(256);
  hipLaunchKernelGGL(k_gated_bwd_dst, grid, blk, 0, st, Ah, Bh, e_new, h_new, den, dh, de, N, C, rowptr, col, eperm, dE, de_new, scratch);
  hipLaunchKernelGGL(k_gated_bwd_src, grid, blk, 0, st, e_new, (const float*)de_new, (const float*)scratch, N, C, rev_rowptr, rev_col,
                     rev_eperm, dD, dB);
  SN_CHECK_LAUNCH("sn_gated_aggregate_bwd_f32");
  return SN_OK;
}
