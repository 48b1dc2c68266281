// plan.hip — per-batch structure: graph_ptr / nvalid / dst-sorted CSR, and eigen-data packing.
// Replaces the reference's to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-61),
// the mask construction of SignNet.forward (sign_net.py:100-102) and PyG's per-call COO gather
// (torch_geometric MessagePassing) — see include/signnet_hip.h.
#include "common.hpp"

namespace sn {

// status words: [0] error bits, [1] max nodes per graph, [2] max in-degree, [3] reserved
enum { ST_ERR = 0, ST_NMAX = 1, ST_DEGMAX = 2 };
enum { ERR_UNSORTED = 1, ERR_GRAPH_ID = 2, ERR_EDGE_RANGE = 4, ERR_EDGE_CROSS = 8 };

// K1: per node — graph id, graph boundaries; zero the in-degree counters.
__global__ void k_plan_nodes(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                             int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                             int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) graph_ptr[B] = (int32_t)N;
  if (i >= N) return;
  int64_t g = batch[i];
  deg[i] = 0;
  if (g < 0 || g >= B) {
    atomicOr(&status[ST_ERR], ERR_GRAPH_ID);
    node_graph[i] = 0;
    return;
  }
  node_graph[i] = (int32_t)g;
  int64_t gp = (i == 0) ? -1 : batch[i - 1];
  if (gp > g) atomicOr(&status[ST_ERR], ERR_UNSORTED);
  if (gp != g) {
    // every graph id in (gp, g] starts here (ids in between are empty graphs)
    for (int64_t t = (gp < 0 ? 0 : gp + 1); t <= g; ++t) graph_ptr[t] = (int32_t)i;
  }
  if (i == N - 1)
    for (int64_t t = g + 1; t < B; ++t) graph_ptr[t] = (int32_t)N;
}

// K2: per edge — validate, count in-degree.
__global__ void k_plan_degree(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                              const int32_t* __restrict__ node_graph, int32_t* __restrict__ deg,
                              int32_t* __restrict__ status) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(&status[ST_ERR], ERR_EDGE_RANGE);
    return;
  }
  if (node_graph[s] != node_graph[d]) atomicOr(&status[ST_ERR], ERR_EDGE_CROSS);
  atomicAdd(&deg[d], 1);
}

// K3: one workgroup — exclusive scans: deg -> rowptr (and cursor copy), n_b^2 -> evoff; nvalid; maxima.
__global__ __launch_bounds__(1024) void k_plan_scan(int64_t N, int64_t B, int kmax,
                                                    const int32_t* __restrict__ graph_ptr,
                                                    const int32_t* __restrict__ node_graph,
                                                    int32_t* __restrict__ deg /* in: degree, out: cursor=rowptr */,
                                                    int32_t* __restrict__ rowptr,
                                                    int32_t* __restrict__ nvalid,
                                                    int64_t* __restrict__ evoff,
                                                    int32_t* __restrict__ status) {
  __shared__ long long part[1024];
  __shared__ long long carry_s;
  __shared__ int maxs[2];
  const int T = blockDim.x, t = threadIdx.x;
  if (t == 0) { carry_s = 0; maxs[0] = 0; maxs[1] = 0; }
  __syncthreads();
  // ---- rowptr = exclusive scan of deg (chunked: each thread owns a contiguous run)
  {
    int64_t per = (N + T - 1) / T;
    int64_t lo = (int64_t)t * per, hi = lo + per < N ? lo + per : N;
    long long s = 0;
    int dmax = 0;
    for (int64_t i = lo; i < hi; ++i) { s += deg[i]; dmax = dmax > deg[i] ? dmax : deg[i]; }
    part[t] = s;
    atomicMax(&maxs[1], dmax);
    __syncthreads();
    // Hillis-Steele inclusive scan over the T partials
    for (int off = 1; off < T; off <<= 1) {
      long long v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    long long run = part[t] - s;  // exclusive prefix of this thread's run
    for (int64_t i = lo; i < hi; ++i) {
      int dgi = deg[i];
      rowptr[i] = (int32_t)run;
      deg[i] = (int32_t)run;  // fill cursor
      run += dgi;
    }
    if (t == T - 1) rowptr[N] = (int32_t)part[T - 1];
    __syncthreads();
  }
  // ---- evoff = exclusive scan of n_b^2 over graphs
  {
    int64_t per = (B + T - 1) / T;
    int64_t lo = (int64_t)t * per, hi = lo + per < B ? lo + per : B;
    long long s = 0;
    int nmax = 0;
    for (int64_t b = lo; b < hi; ++b) {
      long long n = graph_ptr[b + 1] - graph_ptr[b];
      s += n * n;
      nmax = nmax > (int)n ? nmax : (int)n;
    }
    part[t] = s;
    atomicMax(&maxs[0], nmax);
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {
      long long v = (t >= off) ? part[t - off] : 0;
      __syncthreads();
      part[t] += v;
      __syncthreads();
    }
    long long run = part[t] - s;
    for (int64_t b = lo; b < hi; ++b) {
      long long n = graph_ptr[b + 1] - graph_ptr[b];
      evoff[b] = run;
      run += n * n;
    }
    if (t == T - 1) evoff[B] = part[T - 1];
    __syncthreads();
  }
  // ---- nvalid per node
  for (int64_t i = t; i < N; i += T) {
    int g = node_graph[i];
    int n = graph_ptr[g + 1] - graph_ptr[g];
    nvalid[i] = (kmax > 0 && n > kmax) ? kmax : n;
  }
  if (t == 0) { status[ST_NMAX] = maxs[0]; status[ST_DEGMAX] = maxs[1]; }
}

// K4: per edge — scatter into its destination's segment (arrival order is arbitrary here ...)
__global__ void k_plan_fill(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                            int32_t* __restrict__ cursor, int32_t* __restrict__ col,
                            int32_t* __restrict__ eperm) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) return;
  int p = atomicAdd(&cursor[d], 1);
  col[p] = (int32_t)s;
  eperm[p] = (int32_t)e;
}

// K5: per node — ... so sort each segment by edge id (insertion sort; molecular degrees are <= 4).
// The in-edge order, and therefore the fp32 summation order of every aggregation, is deterministic.
__global__ void k_plan_sort(int64_t N, const int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                            int32_t* __restrict__ eperm) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int lo = rowptr[i], hi = rowptr[i + 1];
  for (int a = lo + 1; a < hi; ++a) {
    int ke = eperm[a], kc = col[a];
    int b = a - 1;
    while (b >= lo && eperm[b] > ke) {
      eperm[b + 1] = eperm[b];
      col[b + 1] = col[b];
      --b;
    }
    eperm[b + 1] = ke;
    col[b + 1] = kc;
  }
}

// Eigen-data packing (to_dense_list_EVD, transform.py:52-61).
__global__ void k_pack_eig(const float* __restrict__ ev, const float* __restrict__ es,
                           const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ node_graph,
                           const int32_t* __restrict__ nvalid, const int64_t* __restrict__ evoff,
                           int64_t N, int K, float* __restrict__ x0, float* __restrict__ s0) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * K) return;
  int64_t i = idx / K;
  int j = (int)(idx - i * K);
  int g = node_graph[i];
  int gs = graph_ptr[g];
  int n = graph_ptr[g + 1] - gs;
  bool ok = j < nvalid[i];
  float v = 0.f, s = 0.f;
  if (ok) {
    v = ev[evoff[g] + (int64_t)(i - gs) * n + j];
    if (s0) s = es[gs + j];
  }
  x0[idx] = v;
  if (s0) s0[idx] = s;
}


// ---------------------------------------------------------------------------- bins for the fused stages
// Units of graph g: kind 0 -> K_g slabs of n_g rows; kind 1 -> n_g nodes of K_g rows; kind 2 -> 1 graph of n_g rows
// (K_g = min(n_g, kmax)).  Next-fit packing in graph order into bins of R rows.
__device__ __forceinline__ void unit_shape(int kind, int n, int kg, int& usize, int& ucount) {
  if (kind == 0) { usize = n; ucount = kg; }
  else if (kind == 1) { usize = kg; ucount = n; }
  else { usize = n; ucount = n > 0 ? 1 : 0; }
}

__global__ __launch_bounds__(1024) void k_plan_bins(const int32_t* __restrict__ graph_ptr, int64_t B, int kmax,
                                                    int kind, int R, int64_t max_bins,
                                                    int32_t* __restrict__ bin_node, int32_t* __restrict__ bin_slot,
                                                    int32_t* __restrict__ meta) {
  extern __shared__ int sh[];   // [B] bin0, [B] fill0  (start state of every graph)
  int* bin0 = sh;
  int* fill0 = sh + B;
  __shared__ int s_err, s_nbins, s_rows;
  const int t = threadIdx.x, T = blockDim.x;
  if (t == 0) {
    int bin = 0, fill = 0, err = 0, rows = 0;
    for (int64_t g = 0; g < B; ++g) {
      int n = graph_ptr[g + 1] - graph_ptr[g];
      int kg = (kmax > 0 && n > kmax) ? kmax : n;
      int us, uc;
      unit_shape(kind, n, kg, us, uc);
      bin0[g] = bin;
      fill0[g] = fill;
      if (us <= 0 || uc <= 0) continue;
      if (us > R) { err = 1; continue; }
      rows += us * uc;
      int a = (R - fill) / us;
      if (a > uc) a = uc;
      fill += a * us;
      int rem = uc - a;
      if (rem > 0) {
        int per = R / us;
        int nb = (rem + per - 1) / per;
        bin += nb;
        fill = (rem - (nb - 1) * per) * us;
      }
    }
    s_err = err;
    s_nbins = (fill > 0 || bin > 0) ? bin + (fill > 0 ? 1 : 0) : 0;
    // `bin` is the index of the currently open bin; it is counted once it holds rows
    if (fill == 0 && bin > 0) s_nbins = bin;  // cannot happen (a new bin is opened only to hold a unit) — kept for safety
    s_rows = rows;
  }
  __syncthreads();
  const int nbins = s_nbins;
  if (t == 0) {
    meta[0] = nbins;
    meta[1] = s_err | (nbins > max_bins ? 2 : 0);
    meta[2] = s_rows;
  }
  if (nbins > max_bins) return;
  // padding rows = -1
  for (int64_t i = t; i < (int64_t)nbins * R; i += T) { bin_node[i] = -1; bin_slot[i] = -1; }
  __syncthreads();
  // expansion: one thread per (graph, unit, row) — loop graphs, spread (unit,row) over threads
  for (int64_t g = 0; g < B; ++g) {
    const int gs = graph_ptr[g];
    const int n = graph_ptr[g + 1] - gs;
    const int kg = (kmax > 0 && n > kmax) ? kmax : n;
    int us, uc;
    unit_shape(kind, n, kg, us, uc);
    if (us <= 0 || uc <= 0 || us > R) continue;
    int a = (R - fill0[g]) / us;
    if (a > uc) a = uc;
    const int per = R / us;
    for (int i = t; i < us * uc; i += T) {
      int u = i / us, r = i - u * us;
      int bin, row0;
      if (u < a) { bin = bin0[g]; row0 = fill0[g] + u * us; }
      else { int v = u - a; bin = bin0[g] + 1 + v / per; row0 = (v % per) * us; }
      int node, slot;
      if (kind == 0) { node = gs + r; slot = u; }
      else if (kind == 1) { node = gs + u; slot = r; }
      else { node = gs + r; slot = 0; }
      int64_t o = (int64_t)bin * R + row0 + r;
      bin_node[o] = node;
      bin_slot[o] = slot;
    }
  }
}

}  // namespace sn

using namespace sn;

extern "C" int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index,
                             int64_t E, int kmax, int32_t* graph_ptr, int32_t* node_graph,
                             int32_t* nvalid, int64_t* evoff, int32_t* rowptr, int32_t* col,
                             int32_t* eperm, int32_t* status, int32_t* scratch, void* stream) {
  SN_REQUIRE(N >= 0 && B >= 0 && E >= 0, "sn_batch_plan: negative size");
  SN_REQUIRE(N < (1ll << 31) && E < (1ll << 31), "sn_batch_plan: N/E exceed int32");
  SN_REQUIRE(graph_ptr && node_graph && nvalid && evoff && rowptr && status && scratch,
             "sn_batch_plan: null output");
  SN_REQUIRE(N == 0 || batch, "sn_batch_plan: null batch");
  SN_REQUIRE(E == 0 || (edge_index && col && eperm), "sn_batch_plan: null edge arrays");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st) != hipSuccess)
    return fail(SN_ERR_LAUNCH, "sn_batch_plan: memset failed");
  int32_t* deg = scratch;
  const int T = 256;
  hipLaunchKernelGGL(k_plan_nodes, dim3((unsigned)cdiv(N > 0 ? N : 1, T)), dim3(T), 0, st, batch, N, B,
                     graph_ptr, node_graph, deg, status);
  if (E > 0)
    hipLaunchKernelGGL(k_plan_degree, dim3((unsigned)cdiv(E, T)), dim3(T), 0, st, edge_index, E, N,
                       node_graph, deg, status);
  hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, N, B, kmax, graph_ptr, node_graph, deg,
                     rowptr, nvalid, evoff, status);
  if (E > 0) {
    hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)cdiv(E, T)), dim3(T), 0, st, edge_index, E, N, deg,
                       col, eperm);
    hipLaunchKernelGGL(k_plan_sort, dim3((unsigned)cdiv(N, T)), dim3(T), 0, st, N, rowptr, col, eperm);
  }
  SN_CHECK_LAUNCH("sn_batch_plan");
  return SN_OK;
}

extern "C" int sn_pack_eig_f32(const float* eigen_vectors, const float* eigen_values,
                               const int32_t* graph_ptr, const int32_t* node_graph,
                               const int32_t* nvalid, const int64_t* evoff, int64_t N, int K, float* x0,
                               float* s0, void* stream) {
  SN_REQUIRE(K > 0 && N >= 0, "sn_pack_eig_f32: bad sizes");
  SN_REQUIRE(x0 && graph_ptr && node_graph && nvalid && evoff, "sn_pack_eig_f32: null pointer");
  SN_REQUIRE(!s0 || eigen_values, "sn_pack_eig_f32: s0 requested without eigen_values");
  if (N == 0) return SN_OK;
  hipLaunchKernelGGL(k_pack_eig, dim3((unsigned)cdiv(N * K, 256)), dim3(256), 0, (hipStream_t)stream,
                     eigen_vectors, eigen_values, graph_ptr, node_graph, nvalid, evoff, N, K, x0, s0);
  SN_CHECK_LAUNCH("sn_pack_eig_f32");
  return SN_OK;
}

extern "C" int64_t sn_bins_bound(int64_t rows_upper_bound, int R) {
  // next-fit: two consecutive bins always hold more than R rows together
  if (R <= 0) return 0;
  return 2 * cdiv(rows_upper_bound > 0 ? rows_upper_bound : 0, R) + 2;
}

extern "C" int sn_plan_bins(const int32_t* graph_ptr, int64_t B, int kmax, int kind, int R, int64_t max_bins,
                            int32_t* bin_node, int32_t* bin_slot, int32_t* meta, void* stream) {
  SN_REQUIRE(graph_ptr && bin_node && bin_slot && meta && B >= 0 && R > 0 && max_bins >= 0 && kind >= 0 && kind <= 2,
             "sn_plan_bins: bad arguments");
  SN_REQUIRE(B <= 16000, "sn_plan_bins: B=%lld graphs exceed the single-workgroup planner (16000)", (long long)B);
  size_t lds = (size_t)2 * (B > 0 ? B : 1) * sizeof(int);
  hipLaunchKernelGGL(k_plan_bins, dim3(1), dim3(1024), lds, (hipStream_t)stream, graph_ptr, B, kmax, kind, R, max_bins,
                     bin_node, bin_slot, meta);
  SN_CHECK_LAUNCH("sn_plan_bins");
  return SN_OK;
}
