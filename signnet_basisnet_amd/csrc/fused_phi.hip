// fused_phi.hip — phi(x) + phi(-x) for every valid (node, eigenvector slot) row, all layers in ONE launch.
//
// Replaces GNN3d.forward applied to x and -x (Alchemy/sign_net/sign_net.py:28-44,113;
// GINESignNetPyG/core/sign_net.py:30-48,115) in eval mode.  Why one kernel: with BatchNorm folded to
// an affine, every op of every phi layer is local to one (graph, slot, sign) slab of n_graph <= 64
// rows, so the [rows, d] activations never need to touch HBM between layers.
//
// Mapping (SN_PHI_BIN_ROWS = 64 rows per workgroup, 4 waves):
//   * graphs are packed into columns (sn_batch_plan: best-fit-decreasing on the node count, <= 64 rows); bin j of a
//     column holds the slot-j slab of every member graph; wave w owns bin rows [16w, 16w+16) for BOTH signs and a
//     wave whose tile has no rows skips the GEMMs;
//   * a row tile lives in registers in the MFMA operand layout of common.hpp
//     (lane = (row = l&15, g = l>>4) holds channels 16*kk + 4*g + t), so the accumulators of one GEMM
//     are directly the operand of the next — no LDS round trip between the two Linears of a MaskedMLP;
//   * the GIN neighbour sum goes through LDS: every wave writes its rows (both signs) to X[sign][row][ch],
//     one barrier, then each lane gathers its CSR neighbours' rows (ds_read_b128) — the reference's
//     [K,E,d] gather + scatter_add (masked_layers.py:75) becomes LDS traffic; the residual `+ previous_x`
//     (sign_net.py:42) is re-read from the same LDS image;
//   * weights are streamed from L2 in pre-packed fragment order (1 KiB coalesced per wave-load), each
//     fragment feeding 8 MFMAs (4 k-steps x 2 signs).
// Bound: fp32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s chip peak); algorithmic flops per valid row:
// 2 signs * (L-1) layers * 2 GEMMs * 2*d*d.
#include "fused_common.hpp"

namespace sn {

constexpr int PHI_R = SN_PHI_BIN_ROWS;  // rows per bin / workgroup
constexpr int PHI_WAVES = PHI_R / 16;

struct PhiStruct {
  const float* ev;
  const int32_t* graph_ptr;
  const int64_t* evoff;
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* bin_col;    // [nbins]   column of every bin
  const int32_t* col_bin0;   // [ncol+1]  first bin of every column
  const int32_t* col_mem;    // [ncol][8] member graphs (-1 = none)
  const int32_t* col_off;    // [ncol][8] row offset of each member inside a bin
  const int32_t* meta;       // [0] nbins, [1] error
  int64_t max_bins;
  int kmax;
  int K;
  float* out;
};

constexpr int PHI_NBR = 8;   // in-neighbours of a row kept in LDS (more: read from the CSR in global memory)

template <int NT>
__global__ __launch_bounds__(PHI_R * 4, 2) void k_phi_fused(PhiStruct S, sn_phi_params P) {
  constexpr int D = 16 * NT;
  constexpr int LD = D + 4;  // +4 floats: conflict-free ds_write_b128 of 8 consecutive rows
  constexpr int NKB = (NT + 1) / 2;
  using Ring = WRing<NT>;
  extern __shared__ __align__(1024) unsigned char lds_raw[];
  float* X = reinterpret_cast<float*>(lds_raw + Ring::BYTES);   // [PHI_R][LD]  x_l of the sign being computed
  float* xs = X + PHI_R * LD;                                   // [PHI_R] scalar eigenvector entries (layer 0)
  unsigned char* nbr = reinterpret_cast<unsigned char*>(xs + PHI_R);   // [PHI_R][PHI_NBR] bin rows of the first in-neighbours
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = wave * 16 + (lane & 15), g = lane >> 4;
  const int nbins = S.meta[0];
  if (S.meta[1] != 0) return;  // a graph has more than 64 nodes: the host falls back to the layer path
  Ring ring;
  ring.init(lds_raw, wave, lane);
  // first [d,d] Linear of a (bin, sign) pass — where the weight stream (re)starts
  const void* wfirst = (P.hid0 != 1) ? P.l0_w2 : (P.n_layers > 1 ? P.layers[0].w1s : nullptr);
  if (NT >= SPLIT_RING && wfirst != nullptr && nbins > (int)blockIdx.x) ring.prologue(wfirst, NT);

  for (int bin = blockIdx.x; bin < nbins; bin += gridDim.x) {
    // ---------------------------------------------------------------- my row: bin -> column -> member graph
    const int colid = S.bin_col[bin];
    const int slot = bin - S.col_bin0[colid];       // every member contributes its slab of eigenvector `slot`
    int node = -1, gs = 0, row0 = 0, e_lo = 0, e_hi = 0;
    float xval = 0.f;
#pragma unroll 1
    for (int k = 0; k < 8; ++k) {
      const int gi = S.col_mem[colid * 8 + k];
      if (gi < 0) break;
      const int off = S.col_off[colid * 8 + k];
      const int g0 = S.graph_ptr[gi], n = S.graph_ptr[gi + 1] - g0;
      const int kg = (S.kmax > 0 && n > S.kmax) ? S.kmax : n;
      if (r >= off && r < off + n && slot < kg) {
        const int li = r - off;
        node = g0 + li;
        gs = g0;
        row0 = off;
        e_lo = S.rowptr[node];
        e_hi = S.rowptr[node + 1];
        xval = S.ev[S.evoff[gi] + (int64_t)li * n + slot];
      }
    }
    const bool valid = node >= 0;
    const bool wave_live = __ballot(valid) != 0ull;   // a 16-row tile without rows skips all MFMAs (keeps barriers + DMA)
    const int deg = e_hi - e_lo;
    __syncthreads();   // previous bin is done with xs / nbr / X
    if (g == 0) {
      xs[r] = xval;
      for (int e = 0; e < deg && e < PHI_NBR; ++e) nbr[r * PHI_NBR + e] = (unsigned char)(row0 + S.col[e_lo + e] - gs);
    }
    float* XR = X + r * LD;                 // my row
    // ---------------------------------------------------------------- layer 0 aggregate (scalar input, sign-free)
    __syncthreads();
    float a0 = 0.f;
    for (int e = 0; e < deg; ++e) a0 += xs[e < PHI_NBR ? (int)nbr[r * PHI_NBR + e] : row0 + S.col[e_lo + e] - gs];
    {
#pragma clang fp contract(off)
      const float sc = 1.f + *P.l0_eps;
      const float self = xval * sc;
      a0 = a0 + self;
    }
    f32x4 res[NT];
#pragma unroll 1
    for (int sg = 0; sg < 2; ++sg) {
      const float as = sg ? -a0 : a0;          // phi(-x): the aggregate of -x is exactly -(aggregate of x)
      f32x4 in[NT], o[NT];
      Split8 sp[NKB];
      // -------------------------------------------------------------- layer 0
      if (P.hid0 == 1) {
        // Linear(1->1) . BN . ReLU . Linear(1->d) [+b] . BN . ReLU          (core/sign_net.py:20, masked_layers.py:54-64)
        if (wave_live) {
          const float w1 = P.l0_w1[0], s0 = P.l0_bn0_scale[0], h0 = P.l0_bn0_shift[0];
          const float t = fmaxf((as * w1) * s0 + h0, 0.f);
          const float* w2v = reinterpret_cast<const float*>(P.l0_w2);
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            const f32x4 w2 = ld4(w2v + c), s1 = ld4(P.l0_bn_scale + c), h1 = ld4(P.l0_bn_shift + c);
            f32x4 b2 = {0.f, 0.f, 0.f, 0.f};
            if (P.l0_bias2) b2 = ld4(P.l0_bias2 + c);
            in[kk] = relu4((t * w2 + b2) * s1 + h1);
          }
        }
      } else {
        // Linear(1->d) . BN . ReLU . Linear(d->d) [+b] . BN . ReLU           (Alchemy sign_net.py:20)
        if (wave_live) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) {
            const int c = 16 * kk + 4 * g;
            const f32x4 w1 = ld4(P.l0_w1 + c), s0 = ld4(P.l0_bn0_scale + c), h0 = ld4(P.l0_bn0_shift + c);
            o[kk] = relu4((as * w1) * s0 + h0);
          }
          split_rows<NT>(o, sp);
        }
        const void* nxt = P.n_layers > 1 ? P.layers[0].w1s : wfirst;
        wg_gemm_split<NT, NT, false>(ring, P.l0_w2, nxt, wave_live, sp, NoPre(),
                                     [&](int ot, f32x4 acc, f32x4 b2, f32x4 s1, f32x4 h1, f32x4) { in[ot] = relu4((acc + b2) * s1 + h1); });
      }
      if (!valid) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // -------------------------------------------------------------- layers 1 .. L-1
#pragma unroll 1
      for (int l = 1; l < P.n_layers; ++l) {
        const sn_phi_layer& Lp = P.layers[l - 1];
        // publish x_l for the neighbour sums and the residual
        if (wave_live) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) lds_st4(XR + 16 * kk + 4 * g, in[kk]);
        }
        lds_barrier();
        if (wave_live) {
          // GIN aggregate: sum of in-neighbours (edge-id order), then + (1+eps) * self
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) o[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < deg; ++e) {
            const int nb = e < PHI_NBR ? (int)nbr[r * PHI_NBR + e] : row0 + S.col[e_lo + e] - gs;
            const float* np = X + nb * LD + 4 * g;
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) o[kk] += lds_ld4(np + 16 * kk);
          }
          {
#pragma clang fp contract(off)
            const float sc = 1.f + *Lp.eps;
#pragma unroll
            for (int kk = 0; kk < NT; ++kk) {
              const f32x4 sf = in[kk] * sc;
              o[kk] = o[kk] + sf;
            }
          }
          split_rows<NT>(o, sp);
        }
        // MaskedMLP: Linear . BN . ReLU . Linear [+b]
        wg_gemm_split<NT, NT, false>(ring, Lp.w1s, Lp.w2s, wave_live, sp, NoPre(),
                                     [&](int ot, f32x4 acc, f32x4 s0, f32x4 h0, f32x4, f32x4) { o[ot] = relu4(acc * s0 + h0); });
        if (wave_live) split_rows<NT>(o, sp);
        const void* nxt = (l + 1 < P.n_layers) ? P.layers[l].w1s : wfirst;   // next sign / next bin restart the stream here
        // GNN3d: mask . BN . ReLU . + previous_x
        wg_gemm_split<NT, NT, false>(
            ring, Lp.w2s, nxt, wave_live, sp, [&](int ot) { return lds_ld4(XR + 16 * ot + 4 * g); },
            [&](int ot, f32x4 acc, f32x4 b2, f32x4 s1, f32x4 h1, f32x4 prev) { in[ot] = relu4((acc + b2) * s1 + h1) + prev; });
        if (!valid) {
#pragma unroll
          for (int kk = 0; kk < NT; ++kk) in[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        lds_barrier();  // everyone is done reading X before it is overwritten
      }
      if (sg == 0) {
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) res[kk] = in[kk];
      } else if (valid) {
        // ------------------------------------------------------------ phi(x) + phi(-x) -> out[node*K + slot, :]
        float* orow = S.out + ((int64_t)node * S.K + slot) * P.d;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
          const int c = 16 * kk + 4 * g;
          const f32x4 v = res[kk] + in[kk];
          if ((P.d & 3) == 0) {
            if (c < P.d) *reinterpret_cast<float4*>(orow + c) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (c + t < P.d) orow[c + t] = v[t];
          }
        }
      }
    }
  }
  ring.drain();
}

template <int NT>
static int launch_phi(const PhiStruct& S, const sn_phi_params& P, hipStream_t st) {
  constexpr int LD = 16 * NT + 4;
  const size_t lds = (size_t)WRing<NT>::BYTES + (size_t)(PHI_R * LD + PHI_R) * sizeof(float) + (size_t)PHI_R * PHI_NBR;
  static int cus = 0;  // idempotent one-time setup (same values whichever thread wins)
  if (cus == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_phi_fused<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_phi_fused_f32: cannot raise the dynamic LDS limit to %zu", lds);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  int64_t grid = S.max_bins < (int64_t)2 * cus ? S.max_bins : (int64_t)2 * cus;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL((k_phi_fused<NT>), dim3((unsigned)grid), dim3(PHI_R * 4), lds, st, S, P);
  return SN_OK;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_phi_fused_f32(const sn_phi_params* params, const float* eigen_vectors, const int32_t* graph_ptr,
                                const int64_t* evoff, const int32_t* rowptr, const int32_t* col,
                                const sn_plan_bins* bins, int kmax, int K, float* out, void* stream) {
  SN_REQUIRE(params && eigen_vectors && graph_ptr && evoff && rowptr && bins && out, "sn_phi_fused_f32: null pointer");
  SN_REQUIRE(bins->phi_bin_col && bins->phi_col_bin0 && bins->phi_col_mem && bins->phi_col_off && bins->meta,
             "sn_phi_fused_f32: incomplete sn_plan_bins");
  const sn_phi_params& P = *params;
  SN_REQUIRE(P.d > 0 && P.d <= 128, "sn_phi_fused_f32: hidden width %d not in (0, 128]", P.d);
  SN_REQUIRE(P.n_layers >= 1 && P.n_layers <= SN_PHI_MAX_LAYERS, "sn_phi_fused_f32: %d layers unsupported", P.n_layers);
  SN_REQUIRE(P.hid0 == 1 || P.hid0 == P.d, "sn_phi_fused_f32: first hidden width must be 1 or d");
  SN_REQUIRE(P.l0_w1 && P.l0_bn0_scale && P.l0_bn0_shift && P.l0_w2 && P.l0_eps, "sn_phi_fused_f32: layer-0 parameters missing");
  SN_REQUIRE(P.hid0 != 1 || (P.l0_bn_scale && P.l0_bn_shift), "sn_phi_fused_f32: layer-0 BatchNorm vectors missing");
  for (int l = 1; l < P.n_layers; ++l) {
    const sn_phi_layer& L = P.layers[l - 1];
    SN_REQUIRE(L.w1s && L.w2s && L.eps, "sn_phi_fused_f32: layer %d parameters missing", l);
  }
  SN_REQUIRE(K > 0 && bins->phi_max_bins >= 0, "sn_phi_fused_f32: bad K / max_bins");
  if (bins->phi_max_bins == 0) return SN_OK;
  PhiStruct S{eigen_vectors, graph_ptr, evoff, rowptr, col, bins->phi_bin_col, bins->phi_col_bin0, bins->phi_col_mem,
              bins->phi_col_off, bins->meta, bins->phi_max_bins, kmax, K, out};
  hipStream_t st = (hipStream_t)stream;
  int rc = SN_OK;
  switch ((P.d + 15) / 16) {
    case 1: rc = launch_phi<1>(S, P, st); break;
    case 2: rc = launch_phi<2>(S, P, st); break;
    case 3: rc = launch_phi<3>(S, P, st); break;
    case 4: rc = launch_phi<4>(S, P, st); break;
    case 5: rc = launch_phi<5>(S, P, st); break;
    case 6: rc = launch_phi<6>(S, P, st); break;
    case 7: rc = launch_phi<7>(S, P, st); break;
    default: rc = launch_phi<8>(S, P, st); break;
  }
  if (rc != SN_OK) return rc;
  SN_CHECK_LAUNCH("sn_phi_fused_f32");
  return SN_OK;
}
