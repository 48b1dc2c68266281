// plan.hip — per-batch structure: graph_ptr / nvalid / dst-sorted CSR, and eigen-data packing.
// Replaces the reference's to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-61),
// the mask construction of SignNet.forward (sign_net.py:100-102) and PyG's per-call COO gather
// (torch_geometric MessagePassing) — see include/signnet_hip.h.
#include "common.hpp"
#include <stdlib.h>

namespace sn {

// status words: [0] error bits, [1] max nodes per graph, [2] max in-degree, [3] reserved
enum { ST_ERR = 0, ST_NMAX = 1, ST_DEGMAX = 2 };
enum { ERR_UNSORTED = 1, ERR_GRAPH_ID = 2, ERR_EDGE_RANGE = 4, ERR_EDGE_CROSS = 8 };

struct BinsOut {           // device-side view of sn_bins_out[3]
  int R[3];
  long long max_bins[3];
  int32_t* node[3];
  int32_t* slot[3];
};

// K1: per node — graph id, graph boundaries; zero the in-degree counters and the status words; mark every
// bin row as padding (-1).  Nothing here is read by another thread of this launch.
__global__ void k_plan_nodes(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                             int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                             int32_t* __restrict__ deg, int32_t* __restrict__ status, BinsOut bo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int k = 0; k < 3; ++k) {
    if (!bo.node[k]) continue;
    const int64_t tot = bo.max_bins[k] * bo.R[k];
    for (int64_t j = i; j < tot; j += stride) { bo.node[k][j] = -1; bo.slot[k][j] = -1; }
  }
  if (i < 4) status[i] = 0;
  if (i == 0) graph_ptr[B] = (int32_t)N;
  if (i >= N) return;
  int64_t g = batch[i];
  deg[i] = 0;
  if (g < 0 || g >= B) { node_graph[i] = 0; return; }   // reported by k_plan_degree
  node_graph[i] = (int32_t)g;
  int64_t gp = (i == 0) ? -1 : batch[i - 1];
  if (gp < -1) gp = -1;
  if (gp != g && gp < g) {
    // every graph id in (gp, g] starts here (ids in between are empty graphs)
    for (int64_t t = gp + 1; t <= g; ++t) graph_ptr[t] = (int32_t)i;
  }
  if (i == N - 1)
    for (int64_t t = g + 1; t < B; ++t) graph_ptr[t] = (int32_t)N;
}

// K2: validation (per node and per edge) and in-degree counting.
__global__ void k_plan_degree(const int64_t* __restrict__ batch, const int64_t* __restrict__ ei, int64_t E,
                              int64_t N, int64_t B, int32_t* __restrict__ deg, int32_t* __restrict__ status) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < N) {
    int64_t g = batch[t];
    if (g < 0 || g >= B) atomicOr(&status[ST_ERR], ERR_GRAPH_ID);
    if (t > 0 && batch[t - 1] > g) atomicOr(&status[ST_ERR], ERR_UNSORTED);
  }
  if (t >= E) return;
  int64_t s = ei[t], d = ei[E + t];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(&status[ST_ERR], ERR_EDGE_RANGE);
    return;
  }
  if (batch[s] != batch[d]) atomicOr(&status[ST_ERR], ERR_EDGE_CROSS);
  atomicAdd(&deg[d], 1);
}

// Units of graph g: kind 0 -> K_g slabs of n_g rows; kind 1 -> n_g nodes of K_g rows; kind 2 -> 1 graph of n_g rows
// (K_g = min(n_g, kmax)).  Next-fit packing in graph order into bins of R rows.
__device__ __forceinline__ void unit_shape(int kind, int n, int kg, int& usize, int& ucount) {
  if (kind == 0) { usize = n; ucount = kg; }
  else if (kind == 1) { usize = kg; ucount = n; }
  else { usize = n; ucount = n > 0 ? 1 : 0; }
}

// Next-fit over the graphs, one wave per stage kind: the per-graph unit shapes are computed by all 64 lanes
// into LDS, then lane 0 walks the graphs (closed form per graph, small-integer divisions done with exact
// float reciprocals) recording the (bin, fill) state every graph starts from.
__device__ __forceinline__ int idiv_small(int a, int b) {   // exact for 0 <= a, 0 < b <= 2^20
  int q = (int)(((float)a + 0.5f) / (float)b);
  return q;
}
__device__ void bins_scan(int kind, int R, long long max_bins, int kmax, const int32_t* __restrict__ graph_ptr, int64_t B,
                          int32_t* __restrict__ bin0, int32_t* __restrict__ fill0, int32_t* __restrict__ meta,
                          int* sh_us, int* sh_uc, int chunk) {
  const int lane = threadIdx.x & 63;
  int bin = 0, fill = 0, err = 0, rows = 0;
  for (int64_t base = 0; base < B; base += chunk) {
    const int cnt = (int)((B - base) < chunk ? (B - base) : chunk);
    for (int i = lane; i < cnt; i += 64) {
      int n = graph_ptr[base + i + 1] - graph_ptr[base + i];
      int kg = (kmax > 0 && n > kmax) ? kmax : n;
      int us, uc;
      unit_shape(kind, n, kg, us, uc);
      sh_us[i] = us;
      sh_uc[i] = uc;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes of this wave are visible to lane 0
    if (lane == 0) {
      for (int i = 0; i < cnt; ++i) {
        const int us = sh_us[i], uc = sh_uc[i];
        bin0[base + i] = bin;
        fill0[base + i] = fill;
        if (us <= 0 || uc <= 0) continue;
        if (us > R) { err = 1; continue; }
        rows += us * uc;
        int a = idiv_small(R - fill, us);
        if (a > uc) a = uc;
        fill += a * us;
        const int rem = uc - a;
        if (rem > 0) {
          const int per = idiv_small(R, us);
          const int nb = idiv_small(rem + per - 1, per);
          bin += nb;                         // the open bin is closed, nb new ones are used, the last stays open
          fill = (rem - (nb - 1) * per) * us;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    const int nbins = fill > 0 ? bin + 1 : bin;
    meta[0] = nbins;
    meta[1] = err | (nbins > max_bins ? 2 : 0);
    meta[2] = rows;
    meta[3] = R;
  }
}

// K3: one workgroup — exclusive scans: deg -> rowptr (and cursor copy), n_b^2 -> evoff; nvalid; maxima; bin states.
__global__ __launch_bounds__(1024) void k_plan_scan(int64_t N, int64_t B, int kmax,
                                                    const int32_t* __restrict__ graph_ptr,
                                                    const int32_t* __restrict__ node_graph,
                                                    int32_t* __restrict__ deg /* in: degree, out: cursor=rowptr */,
                                                    int32_t* __restrict__ rowptr,
                                                    int32_t* __restrict__ nvalid,
                                                    int64_t* __restrict__ evoff,
                                                    int32_t* __restrict__ status, BinsOut bo,
                                                    int32_t* __restrict__ binstate /* [3][2][B] */,
                                                    int32_t* __restrict__ bins_meta /* [3][4] */) {
  __shared__ long long part[1024];
  __shared__ int maxs[2];
  const int T = blockDim.x, t = threadIdx.x;
  if (t == 0) { maxs[0] = 0; maxs[1] = 0; }
  __syncthreads();
  // ---- block 1: the three sequential bin scans (one lane of three waves), concurrent with block 0's scans
  if (blockIdx.x == 1) {
    __shared__ int sh_shape[3][2][1024];
    if ((t >> 6) < 3 && bins_meta) {
      const int k = t >> 6;
      if (bo.node[k]) bins_scan(k, bo.R[k], bo.max_bins[k], kmax, graph_ptr, B, binstate + (2 * k) * B,
                                binstate + (2 * k + 1) * B, bins_meta + 4 * k, sh_shape[k][0], sh_shape[k][1], 1024);
      else if ((t & 63) == 0) { bins_meta[4 * k] = 0; bins_meta[4 * k + 1] = 0; bins_meta[4 * k + 2] = 0; bins_meta[4 * k + 3] = 0; }
    }
    return;
  }
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
    for (int off = 1; off < T; off <<= 1) {   // Hillis-Steele inclusive scan over the T partials
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

// K4: blocks [0, eblocks): per edge — scatter into its destination's segment (arrival order arbitrary, fixed
// by k_plan_sort);  blocks [eblocks, eblocks + B): one graph each — write its bin rows for the three stage kinds.
__global__ __launch_bounds__(256) void k_plan_fill(const int64_t* __restrict__ ei, int64_t E, int64_t N, int eblocks,
                                                   int32_t* __restrict__ cursor, int32_t* __restrict__ col,
                                                   int32_t* __restrict__ eperm, const int32_t* __restrict__ graph_ptr,
                                                   int64_t B, int kmax, BinsOut bo,
                                                   const int32_t* __restrict__ binstate,
                                                   const int32_t* __restrict__ bins_meta) {
  if ((int)blockIdx.x < eblocks) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) return;
    int p = atomicAdd(&cursor[d], 1);
    col[p] = (int32_t)s;
    eperm[p] = (int32_t)e;
    return;
  }
  const int64_t g = (int64_t)blockIdx.x - eblocks;
  if (g >= B) return;
  const int gs = graph_ptr[g];
  const int n = graph_ptr[g + 1] - gs;
  const int kg = (kmax > 0 && n > kmax) ? kmax : n;
  for (int kind = 0; kind < 3; ++kind) {
    if (!bo.node[kind] || bins_meta[4 * kind + 1] != 0) continue;
    const int R = bo.R[kind];
    int us, uc;
    unit_shape(kind, n, kg, us, uc);
    if (us <= 0 || uc <= 0 || us > R) continue;
    const int b0 = binstate[(2 * kind) * B + g], f0 = binstate[(2 * kind + 1) * B + g];
    int a = (R - f0) / us;
    if (a > uc) a = uc;
    const int per = R / us;
    for (int i = threadIdx.x; i < us * uc; i += blockDim.x) {
      int u = i / us, r = i - u * us;
      int bin, row0;
      if (u < a) { bin = b0; row0 = f0 + u * us; }
      else { int v = u - a; bin = b0 + 1 + v / per; row0 = (v % per) * us; }
      int node, slot;
      if (kind == 0) { node = gs + r; slot = u; }
      else if (kind == 1) { node = gs + u; slot = r; }
      else { node = gs + r; slot = 0; }
      const int64_t o = (int64_t)bin * R + row0 + r;
      bo.node[kind][o] = node;
      bo.slot[kind][o] = slot;
    }
  }
}

// K5: per node — sort each CSR segment by edge id (insertion sort; molecular degrees are <= 4).
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

// ---------------------------------------------------------------------------- single-workgroup plan (small batches)
// Everything sn_batch_plan produces, in ONE launch of one 1024-thread workgroup with all intermediate state in
// LDS — for batches with N <= 4096 nodes, E <= 10240 edges, B <= 512 graphs (the reference's batch sizes:
// 128-256 molecules).  Five dependent launches cost ~60 us on MI355X; this path costs one.
constexpr int PS_NMAX = 4096, PS_EMAX = 10240, PS_BMAX = 512, PS_T = 1024;

__device__ __forceinline__ int ps_block_exscan(int v, int* part, int t) {   // exclusive scan of one int per thread
  part[t] = v;
  __syncthreads();
  for (int off = 1; off < PS_T; off <<= 1) {
    int a = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += a;
    __syncthreads();
  }
  return part[t] - v;
}

__global__ __launch_bounds__(PS_T) void k_plan_small(const int64_t* __restrict__ batch, int N, int B,
                                                     const int64_t* __restrict__ ei, int E, int kmax,
                                                     int32_t* __restrict__ graph_ptr, int32_t* __restrict__ node_graph,
                                                     int32_t* __restrict__ nvalid, int64_t* __restrict__ evoff,
                                                     int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                     int32_t* __restrict__ eperm, int32_t* __restrict__ status, BinsOut bo,
                                                     int32_t* __restrict__ bins_meta) {
  extern __shared__ int sm[];
  int* gp = sm;                      // [B+1]
  int* deg = gp + (PS_BMAX + 1);     // [N]  in-degree, then fill cursor
  int* rp = deg + PS_NMAX;           // [N+1]
  int* lcol = rp + (PS_NMAX + 1);    // [E]
  int* lperm = lcol + PS_EMAX;       // [E]
  int* part = lperm + PS_EMAX;       // [PS_T]
  int* bst = part + PS_T;            // [3][2][B] bin start state per graph
  __shared__ int s_err, s_nmax, s_dmax, s_nb[3], s_berr[3];
  const int t = threadIdx.x;
  if (t == 0) { s_err = 0; s_nmax = 0; s_dmax = 0; }
  if (t < 3) { s_nb[t] = 0; s_berr[t] = 1; }
  for (int i = t; i < N; i += PS_T) deg[i] = 0;
  for (int i = t; i <= B; i += PS_T) gp[i] = N;     // graphs after the last node (and gp[B]) start at N
  __syncthreads();
  // ---- nodes: graph ids, boundaries, validation
  for (int i = t; i < N; i += PS_T) {
    const long long g = batch[i];
    const long long gprev = (i == 0) ? -1 : batch[i - 1];
    if (g < 0 || g >= B) { atomicOr(&s_err, ERR_GRAPH_ID); node_graph[i] = 0; continue; }
    node_graph[i] = (int)g;
    if (gprev > g) atomicOr(&s_err, ERR_UNSORTED);
    if (gprev < g) {
      const long long lo = gprev < -1 ? 0 : gprev + 1;
      for (long long k = lo; k <= g; ++k) gp[k] = i;   // ids in (gprev, g) are empty graphs starting here too
    }
  }
  __syncthreads();
  // ---- three waves walk the graphs for the bin packing while the others count in-degrees
  const int wave = t >> 6, lane = t & 63;
  if (wave >= 13) {
    const int k = wave - 13;
    if (lane == 0 && bins_meta) {
      if (bo.node[k]) {
        const int R = bo.R[k];
        int bin = 0, fill = 0, err = 0, rows = 0;
        for (int g = 0; g < B; ++g) {
          const int n = gp[g + 1] - gp[g];
          const int kg = (kmax > 0 && n > kmax) ? kmax : n;
          int us, uc;
          unit_shape(k, n, kg, us, uc);
          bst[(2 * k) * PS_BMAX + g] = bin;
          bst[(2 * k + 1) * PS_BMAX + g] = fill;
          if (us <= 0 || uc <= 0) continue;
          if (us > R) { err = 1; continue; }
          rows += us * uc;
          int a = idiv_small(R - fill, us);
          if (a > uc) a = uc;
          fill += a * us;
          const int rem = uc - a;
          if (rem > 0) {
            const int per = idiv_small(R, us);
            const int nb = idiv_small(rem + per - 1, per);
            bin += nb;
            fill = (rem - (nb - 1) * per) * us;
          }
        }
        const int nbins = fill > 0 ? bin + 1 : bin;
        const int berr = err | (nbins > bo.max_bins[k] ? 2 : 0);
        bins_meta[4 * k] = nbins;
        bins_meta[4 * k + 1] = berr;
        bins_meta[4 * k + 2] = rows;
        bins_meta[4 * k + 3] = R;
        s_nb[k] = nbins;
        s_berr[k] = berr;
      } else {
        bins_meta[4 * k] = 0; bins_meta[4 * k + 1] = 0; bins_meta[4 * k + 2] = 0; bins_meta[4 * k + 3] = 0;
      }
    }
  } else {
    for (int e = t; e < E; e += 13 * 64) {
      const long long s = ei[e], d = ei[(long long)E + e];
      if (s < 0 || s >= N || d < 0 || d >= N) { atomicOr(&s_err, ERR_EDGE_RANGE); continue; }
      if (batch[s] != batch[d]) atomicOr(&s_err, ERR_EDGE_CROSS);
      atomicAdd(&deg[d], 1);
    }
  }
  __syncthreads();
  // ---- rowptr = exclusive scan of deg ; evoff = exclusive scan of n^2
  {
    const int per = (N + PS_T - 1) / PS_T;
    const int lo = t * per, hi = (lo + per < N) ? lo + per : N;
    int s = 0, dmax = 0;
    for (int i = lo; i < hi; ++i) { s += deg[i]; dmax = dmax > deg[i] ? dmax : deg[i]; }
    atomicMax(&s_dmax, dmax);
    int run = ps_block_exscan(s, part, t);
    for (int i = lo; i < hi; ++i) { const int dg = deg[i]; rp[i] = run; deg[i] = run; run += dg; }
    if (t == PS_T - 1) rp[N] = part[PS_T - 1];
    __syncthreads();
    const int n = (t < B) ? gp[t + 1] - gp[t] : 0;
    atomicMax(&s_nmax, n);
    const int ex = ps_block_exscan(n * n, part, t);
    if (t < B) evoff[t] = ex;
    if (t == PS_T - 1) evoff[B] = part[PS_T - 1];
  }
  __syncthreads();
  // ---- fill the CSR segments (arbitrary arrival order), then sort each segment by edge id
  for (int e = t; e < E; e += PS_T) {
    const long long s = ei[e], d = ei[(long long)E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) continue;
    const int p = atomicAdd(&deg[d], 1);
    lcol[p] = (int)s;
    lperm[p] = e;
  }
  __syncthreads();
  for (int i = t; i < N; i += PS_T) {
    const int lo = rp[i], hi = rp[i + 1];
    for (int a = lo + 1; a < hi; ++a) {
      const int ke = lperm[a], kc = lcol[a];
      int b = a - 1;
      while (b >= lo && lperm[b] > ke) { lperm[b + 1] = lperm[b]; lcol[b + 1] = lcol[b]; --b; }
      lperm[b + 1] = ke;
      lcol[b + 1] = kc;
    }
  }
  __syncthreads();
  // ---- write out
  for (int i = t; i <= N; i += PS_T) rowptr[i] = rp[i];
  for (int i = t; i < E; i += PS_T) { col[i] = lcol[i]; eperm[i] = lperm[i]; }
  for (int i = t; i <= B; i += PS_T) graph_ptr[i] = gp[i];
  for (int i = t; i < N; i += PS_T) {
    const long long g = batch[i];
    int nv = 0;
    if (g >= 0 && g < B) { const int n = gp[g + 1] - gp[g]; nv = (kmax > 0 && n > kmax) ? kmax : n; }
    nvalid[i] = nv;
  }
  if (t == 0) { status[ST_ERR] = s_err; status[ST_NMAX] = s_nmax; status[ST_DEGMAX] = s_dmax; status[3] = 0; }
  // ---- bin rows: padding first (tail of every bin is rewritten below where a unit lands), then one wave per graph
  for (int k = 0; k < 3; ++k) {
    if (!bo.node[k] || s_berr[k] != 0) continue;
    const int R = bo.R[k];
    const int nb = s_nb[k];
    for (int i = t; i < nb * R; i += PS_T) { bo.node[k][i] = -1; bo.slot[k][i] = -1; }
  }
  __syncthreads();
  for (int g = wave; g < B; g += PS_T / 64) {
    const int gs = gp[g], n = gp[g + 1] - gs;
    const int kg = (kmax > 0 && n > kmax) ? kmax : n;
    for (int k = 0; k < 3; ++k) {
      if (!bo.node[k] || s_berr[k] != 0) continue;
      const int R = bo.R[k];
      int us, uc;
      unit_shape(k, n, kg, us, uc);
      if (us <= 0 || uc <= 0 || us > R) continue;
      const int b0 = bst[(2 * k) * PS_BMAX + g], f0 = bst[(2 * k + 1) * PS_BMAX + g];
      int a = idiv_small(R - f0, us);
      if (a > uc) a = uc;
      const int per = idiv_small(R, us);
      for (int i = lane; i < us * uc; i += 64) {
        const int u = i / us, r = i - u * us;
        int bin, row0;
        if (u < a) { bin = b0; row0 = f0 + u * us; }
        else { const int v = u - a; bin = b0 + 1 + v / per; row0 = (v % per) * us; }
        int node, slot;
        if (k == 0) { node = gs + r; slot = u; }
        else if (k == 1) { node = gs + u; slot = r; }
        else { node = gs + r; slot = 0; }
        const long long o = (long long)bin * R + row0 + r;
        bo.node[k][o] = node;
        bo.slot[k][o] = slot;
      }
    }
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


}  // namespace sn

using namespace sn;

extern "C" int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index,
                             int64_t E, int kmax, int32_t* graph_ptr, int32_t* node_graph,
                             int32_t* nvalid, int64_t* evoff, int32_t* rowptr, int32_t* col,
                             int32_t* eperm, int32_t* status, const sn_bins_out* bins, int32_t* bins_meta,
                             int32_t* scratch, void* stream) {
  SN_REQUIRE(N >= 0 && B >= 0 && E >= 0, "sn_batch_plan: negative size");
  SN_REQUIRE(N < (1ll << 31) && E < (1ll << 31), "sn_batch_plan: N/E exceed int32");
  SN_REQUIRE(graph_ptr && node_graph && nvalid && evoff && rowptr && status && scratch,
             "sn_batch_plan: null output");
  SN_REQUIRE(N == 0 || batch, "sn_batch_plan: null batch");
  SN_REQUIRE(E == 0 || (edge_index && col && eperm), "sn_batch_plan: null edge arrays");
  BinsOut bo;
  for (int k = 0; k < 3; ++k) {
    bo.R[k] = 0; bo.max_bins[k] = 0; bo.node[k] = nullptr; bo.slot[k] = nullptr;
    if (bins && bins[k].node) {
      SN_REQUIRE(bins[k].slot && bins[k].R > 0 && bins[k].max_bins >= 0 && bins_meta, "sn_batch_plan: bad bins[%d]", k);
      bo.R[k] = bins[k].R; bo.max_bins[k] = bins[k].max_bins; bo.node[k] = bins[k].node; bo.slot[k] = bins[k].slot;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  static const bool small_path = getenv("SN_PLAN_SMALL") != nullptr;   // experimental: measured slower (124 vs 61 us)
  if (small_path && N <= PS_NMAX && E <= PS_EMAX && B <= PS_BMAX && N > 0) {
    const size_t lds = (size_t)((PS_BMAX + 1) + PS_NMAX + (PS_NMAX + 1) + 2 * PS_EMAX + PS_T + 6 * PS_BMAX) * sizeof(int);
    static bool init = false;
    if (!init) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_plan_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return fail(SN_ERR_LAUNCH, "sn_batch_plan: cannot raise the dynamic LDS limit to %zu", lds);
      init = true;
    }
    hipLaunchKernelGGL(k_plan_small, dim3(1), dim3(PS_T), lds, st, batch, (int)N, (int)B, edge_index, (int)E, kmax, graph_ptr,
                       node_graph, nvalid, evoff, rowptr, col, eperm, status, bo, bins_meta);
    SN_CHECK_LAUNCH("sn_batch_plan");
    return SN_OK;
  }
  int32_t* deg = scratch;                       // [N]
  int32_t* binstate = scratch + ((N + 3) / 4) * 4;  // [3][2][B]
  const int T = 256;
  const bool any_bins = bo.node[0] || bo.node[1] || bo.node[2];
  int64_t nb1 = cdiv(N > 0 ? N : 1, T);
  if (nb1 < 64) nb1 = 64;                       // enough threads to clear the bin arrays quickly
  hipLaunchKernelGGL(k_plan_nodes, dim3((unsigned)nb1), dim3(T), 0, st, batch, N, B, graph_ptr, node_graph, deg, status, bo);
  const int64_t ne = N > E ? N : E;
  hipLaunchKernelGGL(k_plan_degree, dim3((unsigned)cdiv(ne > 0 ? ne : 1, T)), dim3(T), 0, st, batch, edge_index, E, N, B, deg,
                     status);
  hipLaunchKernelGGL(k_plan_scan, dim3(any_bins ? 2 : 1), dim3(1024), 0, st, N, B, kmax, graph_ptr, node_graph, deg, rowptr, nvalid, evoff,
                     status, bo, binstate, bins_meta);
  const int eblocks = (int)cdiv(E, T);
  const int64_t gblocks = any_bins ? B : 0;
  if (eblocks + gblocks > 0)
    hipLaunchKernelGGL(k_plan_fill, dim3((unsigned)(eblocks + gblocks)), dim3(T), 0, st, edge_index, E, N, eblocks, deg, col,
                       eperm, graph_ptr, B, kmax, bo, binstate, bins_meta);
  if (E > 0)
    hipLaunchKernelGGL(k_plan_sort, dim3((unsigned)cdiv(N, T)), dim3(T), 0, st, N, rowptr, col, eperm);
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

