// eigenspace.hip — BasisNet's one-off preprocessing on the device and the projector-free form of the
// IGN 2->1 contractions (row a20's algebraic shortcut).
//
// Replaces the module-level code of LearningFilters/training.py:47-73:
//     rounded_vals = around(eigvals, 5); uniq_vals, inv, counts = rounded_vals.unique(return_inverse, return_counts)
//     eigenspaces = tensor_split(eigvecs, cumsum(counts), dim=1);  projectors = [V @ V.T ...]  stacked by multiplicity
// (1) sn_eigenspace_group: eigenvalue multiplicities and the multiplicity-major order of the eigenspaces — one workgroup, the
//     eigenvalues must be ascending (what eigh returns; `unique` on a sorted array is run-length grouping);
// (2) sn_eigenspace_projectors_f32: P_s = V_s V_s^T for every eigenspace, written straight into the [n_spaces, N, N] stack in
//     the order the reference's dict {mult: cat(projectors)} has — HBM-write bound (4*n_spaces*N^2 bytes, 2.15 GB for the
//     32x32 grid), V itself (4 MB) stays in L2;
// (3) sn_ign_contract_eigvecs_f32: the five 2->1 contractions of P_s = V_s V_s^T WITHOUT the projector:
//     diag_i = sum_k V_ik^2, rowsum_i = colsum_i = sum_k V_ik (sum_j V_jk), trace = sum_i diag_i, total = sum_k (sum_j V_jk)^2
//     — 4*N*mult bytes per eigenspace instead of 4*N^2 (4.2 MB instead of 2.15 GB per forward on the 32x32 grid); same
//     maths as contractions_2_to_1 (ign.py:344-374) on the projector, different fp32 summation order.
#include "common.hpp"

namespace sn {

constexpr int EIG_MAXN = 8192;

// status bits (meta[2])
constexpr int EIG_ERR_UNSORTED = 1;

// meta: [0] n_spaces, [1] n_mults, [2] error bits, [3] largest multiplicity
__global__ __launch_bounds__(256) void k_eig_group(const float* __restrict__ eigvals, int N, float scale, int32_t* __restrict__ space_of,
                                                   int32_t* __restrict__ space_start, int32_t* __restrict__ space_mult,
                                                   int32_t* __restrict__ space_slot, int32_t* __restrict__ mult_list,
                                                   int32_t* __restrict__ mult_count, int32_t* __restrict__ meta) {
  extern __shared__ int sm[];
  float* key = reinterpret_cast<float*>(sm);       // [N]  around(): round-half-even(x * 10^d)   (the division by 10^d is injective here)
  int* sid = sm + N;                               // [N]  eigenspace of every eigenvector (inclusive scan of the run starts) / later: hist
  int* part = sid + N;                             // [256] scan partials
  __shared__ int s_err, s_ns;
  const int t = threadIdx.x;
  if (t == 0) { s_err = 0; s_ns = 0; }
  for (int i = t; i < N; i += 256) key[i] = rintf(eigvals[i] * scale);
  __syncthreads();
  // run starts, chunked scan: thread t owns the contiguous range [lo, hi)
  const int per = (N + 255) / 256, lo = t * per < N ? t * per : N, hi = lo + per < N ? lo + per : N;
  int c = 0;
  bool bad = false;
  for (int i = lo; i < hi; ++i) {
    const bool st = i == 0 || key[i] != key[i - 1];
    bad = bad || (i > 0 && key[i] < key[i - 1]);
    c += st ? 1 : 0;
    sid[i] = c;
  }
  part[t] = c;
  if (bad) s_err = EIG_ERR_UNSORTED;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int k = 0; k < 256; ++k) { const int v = part[k]; part[k] = run; run += v; }
    s_ns = run;
  }
  __syncthreads();
  const int ns = s_ns;
  for (int i = lo; i < hi; ++i) {
    const int s = sid[i] + part[t] - 1;
    space_of[i] = s;
    if (i == 0 || key[i] != key[i - 1]) space_start[s] = i;
  }
  if (t == 0) space_start[ns] = N;
  __syncthreads();
  __threadfence_block();
  // multiplicities + histogram over multiplicity (reuses sid as hist[0..N])
  for (int i = t; i < N; i += 256) sid[i] = 0;                    // hist[mult - 1], 1 <= mult <= N
  __syncthreads();
  for (int s = t; s < ns; s += 256) {
    const int m = space_start[s + 1] - space_start[s];
    space_mult[s] = m;
    atomicAdd(&sid[m - 1], 1);
  }
  __syncthreads();
  if (t == 0) {
    // sorted unique multiplicities, their counts, and the first slot of each (multiplicity-major order of the reference's dict,
    // eigenvalue order inside a multiplicity) — a serial pass: one-off preprocessing of a single graph
    int nm = 0, base = 0, mmax = 0;
    for (int m = 1; m <= N; ++m) {
      const int cnt = sid[m - 1];
      if (cnt > 0) { mult_list[nm] = m; mult_count[nm] = cnt; ++nm; mmax = m; }
      sid[m - 1] = base;            // hist -> running slot base of multiplicity m
      base += cnt;
    }
    for (int s = 0; s < ns; ++s) {
      const int m = space_mult[s];
      space_slot[s] = sid[m - 1]++;
    }
    meta[0] = ns; meta[1] = nm; meta[2] = s_err; meta[3] = mmax;
  }
}

constexpr int PROJ_TI = 64;     // rows of P per workgroup
constexpr int PROJ_KC = 32;     // eigenvectors of an eigenspace per pass (held in registers)

// one pass over the tile with KCC <= PROJ_KC eigenvectors in registers
template <int KCC>
__device__ __forceinline__ void proj_pass(const float* __restrict__ V, int N, int ldv, int kbase, int mc, int i0, bool first,
                                          const float (&A)[PROJ_TI][PROJ_KC + 1], float* __restrict__ P) {
  for (int j0 = 0; j0 < N; j0 += 256) {
    const int j = j0 + threadIdx.x;
    float b[KCC];
#pragma unroll
    for (int k = 0; k < KCC; ++k) b[k] = (j < N && k < mc) ? V[(int64_t)j * ldv + kbase + k] : 0.f;
    if (j < N) {
      for (int ii = 0; ii < PROJ_TI && i0 + ii < N; ++ii) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KCC; ++k) acc += A[ii][k] * b[k];
        float* o = P + (int64_t)(i0 + ii) * N + j;
        *o = first ? acc : *o + acc;       // eigenspaces wider than 32: later passes accumulate into the first one's tile
      }
    }
  }
}

// P_s[i, j] = sum_k V[i, k0+k] V[j, k0+k]; grid (n_spaces, ceil(N / 64)); thread t owns columns j = j0 + t of a 256-wide strip.
// The pass is instantiated for 2 / 4 / 8 / 32 eigenvectors and picked per eigenspace (block-uniform): the products of the zero
// padding up to 32 cost 16x the useful work on a grid graph, whose eigenspaces have multiplicity 1-2 — the kernel then wrote its
// 2 GB at 1.65 TB/s.
__global__ __launch_bounds__(256) void k_eig_projectors(const float* __restrict__ V, int N, int ldv, const int32_t* __restrict__ space_start,
                                                        const int32_t* __restrict__ space_slot, float* __restrict__ out) {
  __shared__ float A[PROJ_TI][PROJ_KC + 1];
  const int s = blockIdx.x, i0 = blockIdx.y * PROJ_TI;
  const int k0 = space_start[s], m = space_start[s + 1] - k0;
  float* P = out + (int64_t)space_slot[s] * N * N;
  for (int kc = 0; kc < m; kc += PROJ_KC) {
    const int mc = m - kc < PROJ_KC ? m - kc : PROJ_KC;
    __syncthreads();
    for (int e = threadIdx.x; e < PROJ_TI * PROJ_KC; e += 256) {
      const int ii = e / PROJ_KC, k = e - ii * PROJ_KC;
      A[ii][k] = (i0 + ii < N && k < mc) ? V[(int64_t)(i0 + ii) * ldv + k0 + kc + k] : 0.f;
    }
    __syncthreads();
    if (mc <= 2) proj_pass<2>(V, N, ldv, k0 + kc, mc, i0, kc == 0, A, P);
    else if (mc <= 4) proj_pass<4>(V, N, ldv, k0 + kc, mc, i0, kc == 0, A, P);
    else if (mc <= 8) proj_pass<8>(V, N, ldv, k0 + kc, mc, i0, kc == 0, A, P);
    else proj_pass<PROJ_KC>(V, N, ldv, k0 + kc, mc, i0, kc == 0, A, P);
  }
}

// The five 2->1 contractions of P_s = V_s V_s^T from V_s alone; one workgroup per eigenspace; out[slot, i, 0..4] as
// sn_ign_contract_2to1_f32 lays them out: [diag_i, tr/n, rowsum_i/n, colsum_i/n, total/n^2].
__global__ __launch_bounds__(256) void k_eig_contract(const float* __restrict__ V, int N, int ldv, const int32_t* __restrict__ space_start,
                                                      const int32_t* __restrict__ space_slot, float* __restrict__ out) {
  extern __shared__ float cs[];           // [m] column sums, then [256] reduction scratch
  const int s = blockIdx.x;
  const int k0 = space_start[s], m = space_start[s + 1] - k0;
  float* red = cs + m;
  const int t = threadIdx.x;
  // column sums: thread t owns columns k = t, t+256, ... when m is large; for small m the rows are split over threads and folded
  for (int k = 0; k < m; ++k) {
    float a = 0.f;
    for (int j = t; j < N; j += 256) a += V[(int64_t)j * ldv + k0 + k];
    red[t] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    if (t == 0) cs[k] = red[0];
    __syncthreads();
  }
  float tot = 0.f;
  for (int k = 0; k < m; ++k) tot += cs[k] * cs[k];
  float tr_part = 0.f;
  float* O = out + (int64_t)space_slot[s] * N * 5;
  const float inv_n = 1.0f / (float)N;
  for (int i = t; i < N; i += 256) {
    const float* row = V + (int64_t)i * ldv + k0;
    float d = 0.f, r = 0.f;
    for (int k = 0; k < m; ++k) { const float v = row[k]; d += v * v; r += v * cs[k]; }
    O[i * 5 + 0] = d;
    O[i * 5 + 2] = r * inv_n;
    O[i * 5 + 3] = r * inv_n;
    O[i * 5 + 4] = tot * inv_n * inv_n;
    tr_part += d;
  }
  red[t] = tr_part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float tr = red[0] * inv_n;
  for (int i = t; i < N; i += 256) O[i * 5 + 1] = tr;
}

}  // namespace sn

using namespace sn;

extern "C" int sn_eigenspace_group(const float* eigvals, int N, int decimals, int32_t* space_of, int32_t* space_start,
                                   int32_t* space_mult, int32_t* space_slot, int32_t* mult_list, int32_t* mult_count, int32_t* meta,
                                   void* stream) {
  SN_REQUIRE(eigvals && space_of && space_start && space_mult && space_slot && mult_list && mult_count && meta,
             "sn_eigenspace_group: null pointer");
  SN_REQUIRE(N >= 1 && N <= EIG_MAXN, "sn_eigenspace_group: N=%d not in [1, %d]", N, EIG_MAXN);
  SN_REQUIRE(decimals >= 0 && decimals <= 7, "sn_eigenspace_group: decimals=%d not in [0, 7]", decimals);
  float scale = 1.f;
  for (int i = 0; i < decimals; ++i) scale *= 10.f;           // 10**decimals as torch evaluates `x * 10**decimals` in fp32
  const size_t lds = (size_t)(2 * N + 256) * sizeof(int);
  static bool init = false;
  if (!init) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_eig_group), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((2 * EIG_MAXN + 256) * sizeof(int))) != hipSuccess)
      return fail(SN_ERR_LAUNCH, "sn_eigenspace_group: cannot raise the dynamic LDS limit");
    init = true;
  }
  hipLaunchKernelGGL(k_eig_group, dim3(1), dim3(256), lds, (hipStream_t)stream, eigvals, N, scale, space_of, space_start, space_mult,
                     space_slot, mult_list, mult_count, meta);
  SN_CHECK_LAUNCH("sn_eigenspace_group");
  return SN_OK;
}

extern "C" int sn_eigenspace_projectors_f32(const float* eigvecs, int N, int ldv, const int32_t* space_start, const int32_t* space_slot,
                                            int n_spaces, float* out, void* stream) {
  SN_REQUIRE(eigvecs && space_start && space_slot && out && N >= 1 && ldv >= N && n_spaces >= 0, "sn_eigenspace_projectors_f32: bad arguments");
  if (n_spaces == 0) return SN_OK;
  hipLaunchKernelGGL(k_eig_projectors, dim3((unsigned)n_spaces, (unsigned)cdiv(N, PROJ_TI)), dim3(256), 0, (hipStream_t)stream, eigvecs, N,
                     ldv, space_start, space_slot, out);
  SN_CHECK_LAUNCH("sn_eigenspace_projectors_f32");
  return SN_OK;
}

extern "C" int sn_ign_contract_eigvecs_f32(const float* eigvecs, int N, int ldv, const int32_t* space_start, const int32_t* space_slot,
                                           int n_spaces, int max_mult, float* out, void* stream) {
  SN_REQUIRE(eigvecs && space_start && space_slot && out && N >= 1 && ldv >= N && n_spaces >= 0 && max_mult >= 1 && max_mult <= N,
             "sn_ign_contract_eigvecs_f32: bad arguments");
  if (n_spaces == 0) return SN_OK;
  const size_t lds = (size_t)(max_mult + 256) * sizeof(float);
  SN_REQUIRE(lds <= 64 * 1024, "sn_ign_contract_eigvecs_f32: multiplicity %d too large", max_mult);
  hipLaunchKernelGGL(k_eig_contract, dim3((unsigned)n_spaces), dim3(256), lds, (hipStream_t)stream, eigvecs, N, ldv, space_start,
                     space_slot, out);
  SN_CHECK_LAUNCH("sn_ign_contract_eigvecs_f32");
  return SN_OK;
}
