/*
 * signnet_hip.h — C ABI of libsignnet_hip.so (gfx950 / MI355X).
 *
 * The reference (cptq/SignNet-BasisNet) has no FFI layer: its operator boundary is the
 * torch.nn.Module surface.  This header is the boundary *below* that surface —
 * what the Python modules in signnet_basisnet_amd/ bind with ctypes — one entry point per
 * third-party / ATen kernel class the reference's forward inherits.  Each
 * declaration cites the reference call site(s) (file:line under /root/reference) it replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (torch); nothing is retained;
 *   - outputs and scratch are pre-allocated by the caller; no allocation happens inside;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - return 0 on success, a negative SN_ERR_* code otherwise; sn_last_error() gives the
 *     thread-local message;  no global mutable state; re-entrant across streams;
 *   - activations are fp32 row-major "row matrices" [R, C]; for eigenvector-slot tensors the row
 *     index is node*K + slot (the reference's [N, K, C] layout, sign_net.py:97-113) and a row is
 *     *valid* iff slot < nvalid[node] (the reference's mask_full, sign_net.py:100-102);
 *   - indices coming from the reference's data objects are int64; the plan converts to int32.
 */
#ifndef SIGNNET_HIP_H
#define SIGNNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_OK 0
#define SN_ERR_ARG (-1)      /* bad argument (null pointer, size, alignment) */
#define SN_ERR_LAUNCH (-2)   /* HIP launch / runtime error */
#define SN_ERR_UNSUPPORTED (-3)

#define SN_ABI_VERSION 2   /* 2: sn_plan_bins.phi_bin_mem, the fused finishes of the training links (trailing struct fields) */

int sn_version(void);
const char* sn_last_error(void);
/* number of compute units / XCDs of the current device (for grid sizing and roofline maths) */
int sn_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz);
/* one wave spins for `cycles` shader cycles and writes the count it reached to out_cycles[0] (device memory): bracketed by events it
 * measures the clock the device runs at (bench.py puts it in its line: a slower box is then distinguishable from a slower kernel) */
int sn_clock_probe(int64_t cycles, int64_t* out_cycles, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batch plan.  Replaces: to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-38),
 * `scatter(ones, batch)` size/mask construction (sign_net.py:100-102), and PyG
 * MessagePassing's per-call edge gather order (torch_geometric==2.0.1, call sites
 * masked_layers.py:75, pyg_gnn_wrapper.py:28) by a destination-sorted CSR built once per batch.
 *
 * Inputs : batch[N] int64 ascending; edge_index[2,E] int64 (row 0 = source, row 1 = target).
 * Outputs (int32, caller-allocated):
 *   graph_ptr[B+1]  first node of each graph;        node_graph[N]  graph id per node
 *   nvalid[N]       min(n_graph, |kmax|) per node (kmax = 0: n_graph) — number of valid slots.  kmax < 0 ("full slots", the DGL
 *                   tree's dense [N, K] encodings): same nvalid, but the phi work bins cover all |kmax| slots of every graph
 *   evoff[B+1]      prefix of n_b^2   (offset of graph b's eigenvector block, int64)
 *   rowptr[N+1], col[E] (source node of each in-edge), eperm[E] (edge id, for edge_attr)
 *     — in-edges of a node are ordered by edge id (deterministic summation order)
 *   status[8]       status[0] != 0 -> malformed batch (unsorted batch, edge across graphs, ...);
 *                   status[1] = max nodes per graph, status[2] = max in-degree, status[3] = flags of the fused GINE
 *                   stage (set by sn_gnn_fused_f32), status[4] = its completion counter (zeroed here), [5..7] reserved
 *   bins            work bins of the fused stages (may be NULL) — see "Fused stages" below
 * scratch: int32[N + 8].  One launch (two workgroups) for batches of <= 4096 nodes / 12288 edges / 1024 graphs,
 * five launches otherwise; never a host synchronisation.
 */
typedef struct {
  /* phi: graphs packed into columns of <= 64 rows; bin j of a column = slot j of its graphs.  The four column arrays are OPTIONAL
   * (all NULL: only the member records below are written — what the stage kernels walk; the planner's write-out is shorter) */
  int32_t* phi_bin_col;   /* [phi_max_bins]  column of each bin                                        */
  int64_t phi_max_bins;   /* capacity of phi_bin_col: sn_phi_bins_bound(B, kmax)                          */
  int32_t* phi_col_bin0;  /* [B+1]           first bin of each column (columns <= graphs)                 */
  int32_t* phi_col_mem;   /* [B][8]          member graphs of each column, -1 = none                      */
  int32_t* phi_col_off;   /* [B][8]          row offset of each member inside a bin                       */
  /* rho (sn_rho_fused_f32): a node's K_g slot rows padded to 16*ceil(K_g/16); 64/pad nodes per bin, per graph */
  int32_t* rho_bin0;      /* [B+1]           first bin of each graph                                      */
  int32_t* meta;          /* [8]  phi: nbins, error, real rows, columns ; rho: nbins, error, real rows ; [7] record bins (phi_bin_mem) */
  const int32_t* node_graph; /* [N] sn_batch_plan's node_graph output (set by the caller): with <= 16 slots per node rho uses
                                node-major bins — four consecutive nodes of the batch per bin, ceil(N/4) bins — instead of rho_bin0 */
  /* Member records of every bin, what the stage kernels walk (may be NULL: sn_phi_fused_f32 / the wide rho kernel then need it):
   * [phi_max_bins][8] word pairs, 16-byte aligned; word 0 = graph | index << 13 | row offset << 19 | (rows - 1) << 25 (-1: no
   * member), word 1 = first node of the graph.  A member is one SLAB of a graph of n nodes: phi reads it as the n node rows of
   * eigenvector slot `index`; with all eigenvectors (kmax = 0) rho reads it as the n slot rows of node `index` (same shapes, same
   * bins).  meta[7] = number of record bins.  kmax != 0: the bins of the columns above.  kmax = 0 (up to 4096 graphs): slabs of
   * ANY graphs packed best-fit-decreasing per bin (98-99 % fill where columns reach 92 %: a column's bins above its shorter
   * members hold only the taller ones); no columns are laid out then (meta[3] = 0, meta[0] = meta[7]).
   * All phi_max_bins rows must be readable: a stage kernel requests the rows of its first bins before the plan's bin count has
   * arrived (rows past meta[7] are read and ignored). */
  int32_t* phi_bin_mem;
} sn_plan_bins;

/* Early report of a batch's flags (sn_batch_plan_ex): the workgroups of the one-launch plan write them to PINNED host memory as they
 * finish — the host learns "can the fused stages serve this batch" ~20 us into a forward instead of after its last kernel.
 * host: int32[16], zeroed by the caller before the launch:
 *   [0] status[0] (malformed batch)   [1] largest graph (nodes)   [2] largest in-degree   [3] 1: a graph has > max_graph_edges in-edges
 *   [4] meta[1] (phi bins: 1 = a graph of > 64 nodes) | 8 = a graph without nodes   [5] meta[5] (rho bins)   [6] 1: a feature id outside its embedding tables
 *   [8..11] set to 1 by the CSR / phi-bin / rho-bin / feature-id workgroup AFTER its words above (poll all four).
 * node_ids / edge_ids (may be NULL): the int64 feature ids of the DiscreteEncoders that consume this batch (every column of data.x /
 * data.edge_attr, model_utils/elements.py:21-37) and the row count of their tables: nn.Embedding's IndexError, decided here. */
typedef struct {
  const int64_t* node_ids; int64_t n_node_ids; int64_t node_vocab;
  const int64_t* edge_ids; int64_t n_edge_ids; int64_t edge_vocab;
  int max_graph_edges;     /* > 0: the in-edge capacity of the fused GINE stage (192) */
  int reserved;
  int32_t* host;
} sn_plan_early;

int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index, int64_t E,
                  int kmax, int32_t* graph_ptr, int32_t* node_graph, int32_t* nvalid, int64_t* evoff,
                  int32_t* rowptr, int32_t* col, int32_t* eperm, int32_t* status,
                  const sn_plan_bins* bins /* host struct of device pointers, or NULL */,
                  int32_t* scratch, void* stream);
/* the same with the early report (one-launch plan only: sn_batch_plan_early_supported(N, E, B) != 0); early = NULL: sn_batch_plan */
int sn_batch_plan_ex(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index, int64_t E,
                     int kmax, int32_t* graph_ptr, int32_t* node_graph, int32_t* nvalid, int64_t* evoff,
                     int32_t* rowptr, int32_t* col, int32_t* eperm, int32_t* status,
                     const sn_plan_bins* bins, int32_t* scratch, const sn_plan_early* early, void* stream);
int sn_batch_plan_early_supported(int64_t N, int64_t E, int64_t B);
int64_t sn_phi_bins_bound(int64_t B, int kmax);

/* Eigen-data packing.  Replaces to_dense_list_EVD (transform.py:52-61): x0[node, j] = V_b[local, j]
 * and s0[node, j] = D_b[j] for j < nvalid[node], else 0.  K = slots per node in the output. */
int sn_pack_eig_f32(const float* eigen_vectors, const float* eigen_values, const int32_t* graph_ptr,
                    const int32_t* node_graph, const int32_t* nvalid, const int64_t* evoff, int64_t N,
                    int K, float* x0, float* s0 /* may be NULL */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight packing: W [d_out, d_in] (nn.Linear layout, row stride ldw) -> MFMA fragment order
 * Wp[ceil(d_out/16)][ceil(d_in/16)][64 lanes][4], zero padded.  Wp needs
 * sn_packed_weight_floats(d_out, d_in) floats. */
int64_t sn_packed_weight_floats(int d_out, int d_in);
int sn_pack_weight_f32(const float* W, int d_out, int d_in, int ldw, float* Wp, void* stream);
/* the same packing of W^T, read in place from the row-major [rows, cols] matrix W (the backward's dX = dY W is a Linear whose weight is
 * the forward weight transposed: no transposed copy); Wp: sn_packed_weight_floats(cols, rows) floats. */
int sn_pack_weight_t_f32(const float* W, int rows, int cols, int ldw, float* Wp, void* stream);

/* Split-packed linear for the fused phi / rho stages, which evaluate their fp32 GEMMs on the bf16 matrix pipe:
 * every fp32 weight is split EXACTLY into three bf16 pieces (8 significand bits each) and every product x*w is
 * summed from its six partial products of weight >= 2^-16 (fp32 accumulate); the dropped remainder is <= 2^-23
 * |x||w|, the size of one fp32 rounding (same accuracy class as nn.Linear in fp32, 2.5x the fp32 matrix rate).
 * Layout: per 16-output tile one chunk of 3*ceil(d_in/32) weight fragments [k block][piece][64 lanes][8 bf16]
 * followed by SN_SPLIT_EPI per-channel fp32 vectors e0,e1,e2 (bias / folded BatchNorm of the Linear's epilogue,
 * length d_out, NULL = zeros) in the accumulator layout — the whole Linear streams through LDS as one buffer. */
#define SN_SPLIT_EPI 3
int64_t sn_split_packed_bytes(int d_out, int d_in);
int sn_pack_split_f32(const float* W, int d_out, int d_in, int ldw, const float* e0, const float* e1,
                      const float* e2, void* Wsp /* 16-byte aligned */, void* stream);

/* ------------------------------------------------------------------------------------------
 * GIN aggregation.  Replaces PyG GINConv(Identity()).propagate + (1+eps)x
 * (masked_layers.py:70,75; pyg_gnn_wrapper.py:11,16) and dgl.nn.pytorch.GINConv's copy_u/sum
 * (GraphPrediction/layers/gnns.py:90-98,113):
 *     out[i, :] = (1 + *eps) * x[i, :] + sum_{e in in(i)} x[col[e], :]
 * x/out: [N, F] fp32 (F = slots*channels floats per node).  eps: device pointer to one float
 * (NULL = 0).  `negate` != 0 computes the aggregate of -x (the phi(-v) branch, sign_net.py:113).
 */
int sn_gin_aggregate_f32(const float* x, float* out, int64_t N, int F, const int32_t* rowptr,
                         const int32_t* col, const float* eps, int negate, void* stream);

/* Same with one workgroup per (graph, channel chunk) staging the graph's feature slab and its
 * CSR slice in LDS, so every feature row is read from HBM exactly once (graphs must be
 * contiguous node ranges given by graph_ptr; edges must stay inside a graph). */
int sn_gin_aggregate_slab_f32(const float* x, float* out, int64_t N, int F, int64_t B,
                              const int32_t* graph_ptr, const int32_t* rowptr, const int32_t* col,
                              const float* eps, int negate, void* stream);

/* GINE aggregation.  Replaces PyG GINEConv.message/aggregate (pyg_gnn_wrapper.py:23,28):
 *     out[i, :] = (1 + *eps) * x[i, :] + sum_{e in in(i)} relu(x[col[e], :] + ea[eperm[e], :]) */
int sn_gine_aggregate_f32(const float* x, const float* ea, float* out, int64_t N, int C,
                          const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                          const float* eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Masked linear with fused epilogue (fp32 MFMA 16x16x4).  Replaces nn.Linear + the masked
 * zeroing / MaskedBN(eval) / ReLU / residual chain of MaskedMLP.forward (masked_layers.py:54-64),
 * GNN3d.forward (sign_net.py:36-43), MLP.forward (elements.py:58-69), the DGL MLP
 * (GraphPrediction/layers/mlp.py:37-56) and the transformer projections
 * (transformer_module.py:85-99,118-124):
 *     y = x @ W^T (+ bias); invalid rows -> 0; [relu]; [y*scale+shift]; [relu]; [+ residual]
 * flags select the stages.  Row validity: nvalid == NULL -> all rows valid, else row r is valid
 * iff (r % K) < nvalid[r / K].  Invalid rows are written as 0.
 */
#define SN_EPI_BIAS 1
#define SN_EPI_RELU_PRE 2   /* activation before the affine (DGL MLP order, mlp.py:40-46) */
#define SN_EPI_AFFINE 4     /* eval-mode BatchNorm folded to scale/shift */
#define SN_EPI_RELU 8       /* activation after the affine (PyG-tree order) */
#define SN_EPI_RESIDUAL 16
#define SN_EPI_BLOCK_BIAS 32 /* + block_bias[row / rows_per_block][:] together with the bias (sn_masked_linear_blockbias_f32 sets it) */
#define SN_EPI_RESIDUAL_PRE 64 /* the residual is added right behind the bias, BEFORE relu_pre / the affine / relu: BatchNorm(x + Linear(h)) of the
                                * DGL Transformer layer (layers/transformer.py:283-290) as ONE launch; excludes SN_EPI_RESIDUAL */
#define SN_EPI_LEAKY 128       /* LeakyReLU(0.01) (nn.LeakyReLU's default slope: PNA's mixing FCLayer, pna_utils.py) after the affine, in RELU's place */
int sn_masked_linear_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out,
                         const float* bias, const int32_t* nvalid, int K, int flags,
                         const float* scale, const float* shift, const float* residual, int ldr,
                         float* y, int ldy, void* stream);
/* The same with a bias per BLOCK of rows_per_block consecutive rows: y = epilogue(x W^T + bias + block_bias[row / rows_per_block]) — the
 * equivariant 1->1 layers of IGN2to1 / DeepSets (LearningFilters/ign.py:405-414, models.py:58-113: Linear over cat[x, mean over the
 * matrix broadcast to its rows]) = W_a x + (W_b mean) per matrix, without materialising the concatenation. */
int sn_masked_linear_blockbias_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out, const float* bias,
                                   const float* block_bias, int64_t rows_per_block, int ldbb, int flags, const float* scale,
                                   const float* shift, float* y, int ldy, void* stream);

/* Eval-mode BatchNorm1d folded to an affine (nn.BatchNorm1d.forward with running statistics,
 * used at masked_layers.py:19, model.py:50, elements.py:64, sign_net.py:50):
 *   scale[c] = weight[c] / sqrt(running_var[c] + eps),  shift[c] = bias[c] - running_mean[c]*scale[c]
 * for c < C, 0 for C <= c < C_pad.  weight/bias may be NULL (affine=False). */
int sn_bn_fold_f32(const float* weight, const float* bias, const float* running_mean,
                   const float* running_var, float eps, int C, int C_pad, float* scale, float* shift,
                   void* stream);

/* The running-statistics side effect of a train-mode nn.BatchNorm1d forward (torch: momentum 0.1, running_var from
 * the UNBIASED batch variance): running = (1-m)*running + m*batch, with mean / var (biased) / count as produced by
 * sn_masked_colstats_f32. */
int sn_bn_running_update_f32(const float* mean, const float* var, const float* count /* device scalar */, float momentum,
                             int C, float* running_mean, float* running_var, void* stream);

/* Per-channel masked statistics for train-mode BatchNorm (MaskedBN on the compacted valid rows,
 * masked_layers.py:19; nn.BatchNorm1d model.py:50): mean[c], biased var[c] over valid rows,
 * count written to *count.  scratch: float[2*C*nblocks] (nblocks = sn_colstats_blocks(R)). */
int sn_colstats_blocks(int64_t R);
int sn_masked_colstats_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K,
                           float* mean, float* var, float* count, float* scratch, void* stream);

/* Train-mode BatchNorm1d statistics in one call (the sequence sn_masked_colstats_f32 -> sn_bn_fold_f32 (x2) ->
 * sn_bn_running_update_f32 with the last four launches fused): mean / biased var / count over the valid rows, rstd, the folded
 * scale = gamma*rstd and shift = beta - mean*scale (gamma / beta may be NULL), and — when running_mean / running_var are given —
 * their momentum update with the unbiased variance.  scratch as for sn_masked_colstats_f32. */
int sn_bn_train_stats_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                          float* var, float* rstd, float* scale, float* shift, float* count, float* scratch, void* stream);

/* A masked Linear followed by a train-mode BatchNorm1d's statistics, the pattern of every MLP layer of the reference in training
 * (GINESignNetPyG/core/model_utils/elements.py MLP: `x = norm(lin(x))`): z = x W^T + b on the valid rows (0 elsewhere), exactly as
 * sn_masked_linear_f32 with SN_EPI_BIAS, and mean / var / rstd / folded (scale, shift) / count / running statistics of z as
 * sn_bn_train_stats_f32 would return them (same values up to the summation order of the moments).  For >= 32 rows, d_in, d_out <= 128 and
 * 16-byte-aligned rows the column moments are taken from the accumulators of the Linear kernel (one partial per wave, Chan's
 * update): z is not read again; other shapes run the two entry points one after the other.  scratch:
 * sn_linear_bn_scratch_floats(R, d_in, d_out) floats. */
int64_t sn_linear_bn_scratch_floats(int64_t R, int d_in, int d_out);
int sn_linear_bn_train_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out, const float* bias,
                           const int32_t* nvalid, int K, float* z, int ldz, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* mean, float* var, float* rstd,
                           float* scale, float* shift, float* count, float* scratch, void* stream);

/* Elementwise y = [relu]( [relu_pre](x) * scale + shift ) [+ residual] on valid rows, 0 elsewhere
 * (the un-fused BatchNorm apply used by the train-mode forward). */
int sn_masked_affine_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K,
                         int flags, const float* scale, const float* shift, const float* residual,
                         int ldr, float* y, int ldy, void* stream);

/* Masked LayerNorm of (x + residual) over channels, eps as given (MaskedLN, masked_layers.py:22-32,
 * used at transformer_module.py:100-101,125-126); invalid rows -> 0. */
int sn_masked_layernorm_f32(const float* x, const float* residual /* may be NULL */, int64_t R, int C,
                            const float* gamma, const float* beta, float eps, const int32_t* nvalid,
                            int K, float* y, void* stream);

/* Per-node multi-head attention over the slot axis (ScaledDotProductAttention,
 * transformer_module.py:50-58, heads split as :85-98): q,k,v,out are [N*K, heads*dk];
 * softmax over the valid slots of the node; invalid query rows -> 0.  prob_mask (may be NULL): the train-mode
 * attention dropout of transformer_module.py:49,55 as an explicit mask [N, heads, K, K] of 0 / 1/(1-p) factors applied to
 * the softmax output (drawn by the caller, so that forward and backward see the same one). */
int sn_set_attention_f32(const float* q, const float* k, const float* v, int64_t N, int K, int heads,
                         int dk, const int32_t* nvalid, const float* prob_mask, float* out, void* stream);

/* The CSR of two disjoint copies of a batch: rowptr2 [2N + 1] = [rowptr | rowptr[1:] + E], col2 [2E] = [col | col + N] — the graph the
 * training step's stacked phi(+x) / phi(-x) aggregation walks in one launch (sign_net.py:113: both sign passes over the same edges). */
int sn_plan_double_i32(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t E, int32_t* rowptr2, int32_t* col2, void* stream);

/* Uniform draws -> a scaled keep-mask, in place: u[i] = u[i] >= p ? scale : 0 — the attention dropout mask of transformer_module.py:49,55
 * (the caller draws u with its own generator; scale = 1 / (1 - p)); u 16-byte aligned. */
int sn_keep_mask_f32(float* u, int64_t n, float p, float scale, void* stream);

/* out[n, :] = sum_k x[n, k, :]  (torch.sum(x, dim=1), sign_net.py:70). */
int sn_slot_sum_f32(const float* x, int64_t N, int K, int C, float* out, void* stream);

/* DiscreteEncoder (elements.py:31-37): out[r,:] = sum_f tables[f][idx[r*ldi + f], :], nf <= 10.
 * `tables` is a HOST array of nf device pointers (each table [table_rows[f], C] fp32), `table_rows` a HOST array of nf row counts.
 * An index outside [0, table_rows[f]) — nn.Embedding raises IndexError there — is never dereferenced: it contributes 0 and sets
 * bit 0 of *status (device int32, may be NULL). */
int sn_embedding_sum_f32(const int64_t* idx, int ldi, int nf, int64_t R, const float* const* tables,
                         const int64_t* table_rows, int C, float* out, int32_t* status, void* stream);

/* Graph pooling (torch_scatter.scatter add/mean, model.py:57-61): out[b,:] over nodes of graph b.
 * mode 0 = add, 1 = mean. */
int sn_segment_pool_f32(const float* x, int64_t B, int C, const int32_t* graph_ptr, int mode,
                        float* out, void* stream);

/* BasisNet: the five equivariant 2->1 contractions of a stack of n x n matrices (eigenspace projectors).
 * Replaces contractions_2_to_1 (LearningFilters/ign.py:344-374, normalization 'inf'), called from
 * layer_2_to_1.forward (ign.py:117-128):  ops[b, i, :] = [X_ii, tr(X)/n, rowsum_i/n, colsum_i/n, sum(X)/n^2],
 * written row-major [b, n, 5] so that the following einsum (ign.py:123) is a 5 -> S masked_linear over the
 * (b, i) rows.  HBM-bound: X is read exactly once (4*b*n*n bytes).
 * scratch: sn_ign_contract_scratch_floats(b, n) floats. */
int64_t sn_ign_contract_scratch_floats(int64_t b, int n);
int sn_ign_contract_2to1_f32(const float* X, int64_t b, int n, float* ops_out, float* scratch, void* stream);

/* Batched Laplacian eigendecomposition (the step before the path).  Replaces, for a whole collated
 * batch on the device, the reference's per-sample host transforms
 *   EVDTransform.__call__ / EVD_Laplacian  (Alchemy/sign_net/transform.py:7-23, GINESignNetPyG/core/transform.py:7-26:
 *     to_undirected -> get_laplacian(normalization) -> dense -> torch.linalg.eigh), and
 *   lap_positional_encoding                (GraphPrediction/data/molecules.py:148-181: I - D^-1/2 A D^-1/2, eig,
 *     ascending sort, eigenvector columns 1..k, zero padding when n <= k).
 * Inputs : edge_index[2,E] int64 (any order; made undirected, self loops dropped, duplicates coalesced);
 *          graph_ptr[B+1] int32 (first node of each graph);  norm 0 = None (D - A), 1 = 'sym'.
 * Outputs: evoff[B+1] int64 = prefix of n_b^2;  eigen_values[N] ascending per graph;
 *          eigen_vectors[total] = the n_b x n_b blocks, row-major V[node, eig] (the reference's flattened wire format,
 *          transform.py:14), total >= sum n_b^2 (the buffer doubles as the dense-adjacency scratch);
 *          pos_enc[N,k] (optional, may be NULL) = V[:, skip : skip+k], zero padded (DGL layout: skip = 1).
 * Eigenvector signs / the basis inside a repeated eigenvalue are arbitrary, as with LAPACK.  Graphs of up to 64 nodes
 * (one-sided Jacobi in registers, 16/32/64 lanes per graph).
 * work: int32[sn_evd_work_ints(B)].  status[4] (zeroed here): status[0] bit 0 = an edge leaves its graph / bad node id,
 * bit 1 = a graph has > 64 nodes (its outputs are left zero), bit 2 = no convergence, bit 3 = `total` too small;
 * status[1] = the largest number of Jacobi sweeps any wave ran (diagnostic). */
int64_t sn_evd_work_ints(int64_t B);
int sn_laplacian_evd_f32(const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, int64_t B, int64_t N,
                         int norm, int64_t* evoff, float* eigen_values, float* eigen_vectors, int64_t total,
                         float* pos_enc, int k, int skip, int32_t* work, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the layer-at-a-time train path.  The reference obtains these from torch.autograd over
 * ATen / PyG / torch_scatter kernels (loss.backward(): Alchemy/main_alchemy.py:108, GINESignNetPyG/core/train.py:62-63);
 * here each forward entry point above has a hand-written adjoint.  Rows r = node*K + slot with slot >= nvalid[node]
 * carry zero gradient.  dX of a Linear is sn_masked_linear_f32 with the transposed weight packed.
 *
 * sn_linear_wgrad_f32: dW[o,i] = sum_r dy[r,o] x[r,i], db[o] = sum_r dy[r,o] over the valid rows (nn.Linear weight / bias
 *   gradients; fp32-input MFMA, per-row-chunk partials + a deterministic reduction).  db may be NULL.
 *   scratch: float[sn_linear_wgrad_scratch_floats(R, d_in, d_out)].
 * sn_bn_act_bwd_f32: train-mode BatchNorm1d (+ ReLU after it) backward.  Forward was a = scale*z + shift with
 *   scale = gamma*rstd, shift = beta - mean*scale (sn_bn_fold_f32 on the batch statistics), y = relu?(a).  Writes
 *   sums[0..C) = d beta = sum g, sums[C..2C) = d gamma = sum g*xhat (g = dy*[a>0], xhat = (z-mean)*rstd) and
 *   dz = scale*(g - d beta/M - xhat*d gamma/M), M = *count.  scratch: float[sn_bn_act_bwd_scratch_floats(R, C)].
 * sn_relu_bwd_f32: dx = dy*[y>0].
 * sn_masked_layernorm_bwd_f32: adjoint of sn_masked_layernorm_f32 (MaskedLN, masked_layers.py:22-32): du (gradient of
 *   both x and residual), d gamma, d beta.  scratch: float[sn_layernorm_bwd_scratch_floats(R, C)].
 * sn_set_attention_bwd_f32: adjoint of sn_set_attention_f32 (softmax recomputed): dq, dk, dv.
 * sn_gine_aggregate_bwd_f32: adjoint of sn_gine_aggregate_f32 w.r.t. the node features (dh) and the per-edge features
 *   (dee, [E,C] in edge-id order) over the REVERSE CSR (rows = source nodes, rev_col = destination, rev_eperm = edge id:
 *   sn_batch_plan on the flipped edge_index).  (GIN's adjoint is sn_gin_aggregate_f32 itself on the reverse CSR.)
 * sn_slot_broadcast_f32 / sn_segment_broadcast_f32: adjoints of sn_slot_sum_f32 (valid slots only) / sn_segment_pool_f32.
 * sn_embedding_sum_bwd_f32: dtables[f][v,:] += sum_{r: idx[r,f] = v} g[r,:] (no atomics: per-chunk partial sums in row order, then
 *   the chunks in order — bitwise reproducible; C <= 512; scratch: float[sn_embedding_bwd_scratch_floats(R, nf, table_rows, C)]);
 * sn_embedding_sum_bwd_layers_f32: the same for L <= 16 gradient planes g[l] ([L][R][C]) that share the index columns — the per-layer
 *   edge encoders of a GINE stack (GNN.forward, model.py:52-60: every layer embeds the same edge_attr with its own table):
 *   dtables[l * nf + f][v,:] += sum_{r: idx[r,f] = v} g[l][r,:], one launch pair for all planes;
 *   scratch: float[sn_embedding_bwd_layers_scratch_floats(R, L, C)].
 *   out-of-range indices contribute nothing and set bit 0 of *status (device int32, may be NULL); table_rows as in the forward.
 * sn_dot_f32: out[0] = sum a[i] b[i] (the GIN / GINE eps gradients).  scratch: float[256].
 * sn_adam_step_f32: one torch.optim.Adam step (no amsgrad; weight_decay added to the gradient) on a flat tensor; the gradient is
 *   read as g*grad_scale (1/world_size after a SUM all-reduce of data-parallel ranks). */
int64_t sn_linear_wgrad_scratch_floats(int64_t R, int d_in, int d_out);
int sn_linear_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, int64_t R, int d_in, int d_out,
                        const int32_t* nvalid, int K, float* dW, float* db, float* scratch, void* stream);
int64_t sn_bn_act_bwd_scratch_floats(int64_t R, int C);
int sn_bn_act_bwd_f32(const float* z, int ldz, const float* dy, int ldd, int64_t R, int C, const int32_t* nvalid, int K,
                      const float* mean, const float* rstd, const float* scale, const float* shift, int relu,
                      const float* count, float* sums, float* dz, int ldo, float* scratch, void* stream);
int sn_relu_bwd_f32(const float* y, const float* dy, int64_t R, int C, const int32_t* nvalid, int K, float* dx, void* stream);
int64_t sn_layernorm_bwd_scratch_floats(int64_t R, int C);
int sn_masked_layernorm_bwd_f32(const float* x, const float* residual, const float* dy, int64_t R, int C, const float* gamma,
                                float eps, const int32_t* nvalid, int K, float* du, float* dgamma, float* dbeta,
                                float* scratch, void* stream);
/* sn_masked_layernorm_bwd_f32 with d gamma / d beta ADDED to the given buffers (a parameter's .grad) */
int sn_masked_layernorm_bwd_acc_f32(const float* x, const float* residual, const float* dy, int64_t R, int C, const float* gamma,
                                    float eps, const int32_t* nvalid, int K, float* du, float* dgamma, float* dbeta,
                                    float* scratch, void* stream);
int sn_set_attention_bwd_f32(const float* q, const float* k, const float* v, const float* dout, int64_t N, int K, int heads,
                             int dk, const int32_t* nvalid, const float* prob_mask, float* dq, float* dk_out, float* dv,
                             void* stream);
int sn_gine_aggregate_bwd_f32(const float* h, const float* ee, const float* g, int64_t N, int C, const int32_t* rev_rowptr,
                              const int32_t* rev_col, const int32_t* rev_eperm, const float* eps, float* dh, float* dee,
                              void* stream);
int sn_slot_broadcast_f32(const float* g, int64_t N, int K, int C, const int32_t* nvalid, float* dx, void* stream);
int sn_segment_broadcast_f32(const float* g, int64_t B, int C, const int32_t* graph_ptr, int mode, float* dx, void* stream);
int64_t sn_embedding_bwd_scratch_floats(int64_t R, int nf, const int64_t* table_rows, int C);
/* forward of the same: out[l][r][:] = sum_f tables[l * nf + f][idx[r, f]] for L <= 16 layers (nf <= 4) that share the index columns */
int sn_embedding_sum_layers_f32(const int64_t* idx, int ldi, int nf, int64_t R, int L, const float* const* tables,
                                const int64_t* table_rows, int C, float* out, int32_t* status, void* stream);
int64_t sn_embedding_bwd_layers_scratch_floats(int64_t R, int L, int C);
int sn_embedding_sum_bwd_layers_f32(const int64_t* idx, int ldi, int nf, int64_t R, int L, float* const* dtables,
                                    const int64_t* table_rows, int C, const float* g, int32_t* status, float* scratch, void* stream);
int sn_embedding_sum_bwd_f32(const int64_t* idx, int ldi, int nf, int64_t R, float* const* dtables, const int64_t* table_rows,
                             int C, const float* g, int32_t* status, float* scratch, void* stream);
int sn_dot_f32(const float* a, const float* b, int64_t n, float* out, float* scratch, void* stream);
int sn_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, void* stream);

/* Message passing of the DGL tree's PNA and sparse-Transformer base networks, over the destination-sorted CSR
 * of sn_batch_plan (in-edges in edge-id order: no atomics, reproducible sums).
 * sn_pna_aggregate_f32: PNATower.reduce_func_for_h (GraphPrediction/layers/pna_layer.py:50-56) with aggregators 'mean max min std'
 *   (pna_utils.py:13-36) and scalers 'identity amplification attenuation' (:68-81, avg_log = avg_d['log']): msg [E, ldm] in edge-id
 *   order -> out[n, off + (4*s + a)*C + c]; with `hself` the node's own C-channel row is copied to out[n, 0:C] first (off = C: the
 *   tower's torch.cat([h, aggregated]), :69), else off = 0.  Nodes without in-edges get zeros.
 * sn_edge_attention_f32: MultiHeadAttentionLayer.propagate_attention (layers/transformer.py:150-195; full_graph False, edge features):
 *   out[i,h,:] = sum_{j->i} s V[j,h,:] / (sum s + 1e-6), s = exp(clamp(sum_c K[j,h,c] Q[i,h,c] / sqrt(dk) * E[e,h,c], -5, 5)); Q/K/V
 *   [N, heads*dk], Ee [E, heads*dk] in edge-id order, dk <= 32.
 * sn_pointwise_f32: y = act((x * rowscale[r]) * scale[c] + shift[c]) + residual, act 0 none / 1 ReLU / 2 LeakyReLU(slope) / 3 ReLU
 *   applied after the residual add; rowscale,
 *   (scale, shift) and residual are optional — PNA's graph_norm (h * snorm_n, pna_layer.py:75-76) + BatchNorm, and the LeakyReLU of
 *   its mixing FCLayer (:126). */
int sn_pna_aggregate_f32(const float* msg, int ldm, const float* hself, int ldh, int C, int64_t N, const int32_t* rowptr,
                         const int32_t* eperm, float avg_log, float* out, int ldo, void* stream);
/* The same reduction with the message formed on the fly: msg(j -> n, edge e) = Ps[j] + Pd[n] + Qe[e] — PNATower.pretrans_edges
 * (pna_layer.py:38-44) is a Linear over cat[h_src, h_dst, e] = W_s h_src + W_d h_dst + (W_e e + b); Ps / Pd [N, >= C] node terms,
 * Qe [E, >= C] the edge term with the bias, col = the CSR's source nodes.  All towers of a layer side by side (C = in_dim). */
int sn_pna_aggregate_gather_f32(const float* Ps, int ldps, const float* Pd, int ldpd, const float* Qe, int ldq, const float* hself, int ldh,
                                int C, int64_t N, const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float avg_log,
                                float* out, int ldo, int tower_width, void* stream);
/* tower_width = 0: the output layout of sn_pna_aggregate_f32 ([13 blocks][C]); tower_width = it > 0: tower-major — tower t = c / it owns the
 * 13 * it contiguous columns [own it | 12 aggregate blocks of it], which is what the towers' posttrans Linears read:
 * sn_grouped_linear_f32: y[:, g*dout..] = ((x[:, g*din..] W_g^T + b_g) * rowscale[row]) * scale + shift for G column groups (a block-
 * diagonal Linear; W [G][dout][din] row-major, din % 4 == 0, din <= 256, dout <= 16; rowscale / scale+shift / bias optional):
 * PNATower's posttrans Linear, graph_norm (h * snorm_n) and the folded BatchNorm of all towers in one launch (pna_layer.py:69-79). */
int sn_grouped_linear_f32(const float* x, int ldx, int64_t R, int G, int din, int dout, const float* W, const float* bias,
                          const float* rowscale, const float* scale, const float* shift, float* y, int ldy, void* stream);
int sn_edge_attention_f32(const float* Q, const float* K, const float* V, const float* Ee, int64_t N, int heads, int dk,
                          const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float* out, void* stream);
/* The same with row strides (in floats): Q / K / V rows ldq apart — the three column blocks of ONE [N, 3*heads*dk] projection
 * (MultiHeadAttentionLayer's Q, K, V Linears, layers/transformer.py:122-126, evaluated as one [3d, d] GEMM) — and Ee rows lde apart
 * (every layer's E projection of the same edge embedding in one [E, L*heads*dk] matrix). */
int sn_edge_attention_strided_f32(const float* Q, const float* K, const float* V, int ldq, const float* Ee, int lde, int64_t N, int heads,
                                  int dk, const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float* out, void* stream);
int sn_pointwise_f32(const float* x, int ldx, int64_t R, int C, const float* rowscale, const float* scale, const float* shift, int act,
                     float slope, const float* residual, int ldr, float* y, int ldy, void* stream);

/* dgl.nn.pytorch.GATConv after its fc Linear, as GATNet builds it (nets/ZINC_graph_regression/gat_net.py:62-66: feat_drop = attn_drop = 0,
 * negative_slope 0.2, no residual, bias, ReLU) — DGL is absent and unpinned by the reference; semantics restated from its published
 * definition: feat [N, heads*C] = fc(h); e_ij = leaky_relu(feat_j . attn_l[h] + feat_i . attn_r[h], slope) over the in-edges j -> i;
 * a = softmax over a node's in-edges; out[i,h,:] = [relu](sum_j a_ij feat[j,h,:] + bias[h,:]).  attn_l / attn_r: [heads*C]; bias may be
 * NULL; C <= 64; in-edges in edge-id order (no atomics).  lse [N, heads] (may be NULL): log-sum-exp of each (node, head), for a backward. */
int sn_gat_aggregate_f32(const float* feat, const float* attn_l, const float* attn_r, const float* bias, int64_t N, int heads, int C,
                         float negative_slope, int relu, const int32_t* rowptr, const int32_t* col, float* out, float* lse, void* stream);

/* Adjoints of the f3 message-passing ops (the reference obtains them from torch.autograd through DGL); CSR walks with one owner per
 * output element — no atomics, bitwise reproducible.
 * sn_edge_rows_sum_f32: out[n,:] = sum over n's CSR range of g[eperm[p],:] — the adjoint of gathering node rows onto edges (h[dst] with
 *   the plan, h[src] with the plan of the flipped edge list: pna_layer.py:38-44 `pretrans_edges`).
 * sn_pna_aggregate_bwd_f32: dout [N, ldo] (layout of sn_pna_aggregate_f32) -> dmsg [E, C], dself [N, C] (NULL when the forward had no
 *   hself); the first maximum / minimum in edge order takes the max / min gradient, the clamped variance passes none.
 * sn_edge_attention_bwd_f32: dQ, dK, dV [N, heads*dk], dE [E, heads*dk]; `out` is the forward result; scratch: float[2*E*heads];
 *   rev_*: the CSR of the flipped edge list (source-side sums).
 * sn_act_bwd_f32: dx = dy * act'(x) [* rowscale[r]], x the pre-activation; act 0 none / 1 ReLU / 2 LeakyReLU(slope). */
int sn_edge_rows_sum_f32(const float* g, int ldg, int C, int64_t N, const int32_t* rowptr, const int32_t* eperm, float* out, int ldo,
                         void* stream);
int sn_pna_aggregate_bwd_f32(const float* msg, int ldm, int C, int64_t N, const int32_t* rowptr, const int32_t* eperm, float avg_log,
                             const float* dout, int ldo, float* dmsg, float* dself, void* stream);
int sn_edge_attention_bwd_f32(const float* Q, const float* K, const float* V, const float* Ee, const float* out, const float* dout,
                              int64_t N, int64_t E, int heads, int dk, const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                              const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm, float* dQ, float* dK, float* dV,
                              float* dE, float* scratch, void* stream);
int sn_act_bwd_f32(const float* x, const float* dy, int64_t R, int C, const float* rowscale, int act, float slope, float* dx, void* stream);
/* sn_gat_aggregate_bwd_f32: adjoint of sn_gat_aggregate_f32 (out, lse from the forward call).  dfeat [N, heads*C]; dbias_rows [N, heads*C] =
 *   dout masked by the ReLU (its column sums are the bias gradient); d_el, d_er [N, heads]: gradients of feat . attn_l (as a source) and
 *   feat . attn_r (as a destination) — the attn_l / attn_r gradients are sum_n d_el[n,h] feat[n,h,:] and sum_n d_er[n,h] feat[n,h,:].
 *   rev_*: the CSR of the flipped edge list (edges grouped by source).  scratch: 2 * E * heads floats. */
int sn_gat_aggregate_bwd_f32(const float* feat, const float* attn_l, const float* attn_r, const float* bias, const float* out,
                             const float* lse, const float* dout, int64_t N, int64_t E, int heads, int C, float negative_slope, int relu,
                             const int32_t* rowptr, const int32_t* col, const int32_t* eperm, const int32_t* rev_rowptr,
                             const int32_t* rev_col, const int32_t* rev_eperm, float* dfeat, float* dbias_rows, float* d_el, float* d_er,
                             float* scratch, void* stream);

/* The GatedGCN network of the DGL tree from its first GatedGCNLayer to the scores, ONE launch, eval mode:
 * replaces the layer loop, readout and MLPReadout of GatedGCNNet.forward (nets/ZINC_graph_regression/gatedgcn_net.py:105-148) and
 * GatedGCNLayer.forward (layers/gatedgcn_layer.py:36-81; batch_norm, no dropout, no graph_norm).  One workgroup per graph
 * (<= 64 nodes, <= sn_gatedgcn_max_edges(d) in-edges: other graphs get a NaN score and status[3] |= 1 / 2 — the caller then uses the
 * layer-at-a-time entry points; status[5] != 0 on entry — an embedding id out of range upstream — makes every score NaN).  d: a multiple of 4 in [4, 96]; dp = max(48, 16*ceil(d/16)); every matrix is sn_pack_split_f32 of its zero-padded form:
 *   wabde: [4*dp, dp] = A | B | D | E stacked (each padded to dp rows), e0 = the four biases
 *   wc:    [dp, dp] = C, (e0, e1, e2) = (bias, folded bn_node_e scale, shift)
 *   h_scale / h_shift: [dp] folded bn_node_h;  residual: 1 when the layer adds its input (gatedgcn_layer.py:29-31,70-72)
 * h [N, d]: embedded node features (read only); e [E, d]: embedded edge features, UPDATED IN PLACE layer by layer (edge-id order);
 * readout: mean (readout_mean = 1) or sum over the nodes, then Linear(d_out, ro_d1) ReLU Linear(ro_d1, ro_d2) ReLU Linear(ro_d2, 1)
 * with plain row-major fp32 weights (layers/mlp_readout_layer.py).  y: [B]. */
typedef struct {
  const void* wabde;
  const void* wc;
  const float* h_scale;
  const float* h_shift;
  int residual, reserved;
} sn_gatedgcn_layer;

#define SN_GATED_MAX_LAYERS 32
typedef struct {
  int d, d_out, n_layers, readout_mean, ro_d1, ro_d2;
  const float *ro_w0, *ro_b0, *ro_w1, *ro_b1, *ro_w2, *ro_b2;
  sn_gatedgcn_layer layers[SN_GATED_MAX_LAYERS];
} sn_gatedgcn_params;

int sn_gatedgcn_max_edges(int d);
int sn_gatedgcn_fused_f32(const sn_gatedgcn_params* params, const float* h, float* e, const int32_t* graph_ptr, int64_t B,
                          const int32_t* rowptr, const int32_t* col, const int32_t* eperm, int32_t* status, float* y, void* stream);

/* Dense multi-head softmax attention over whole sequences, forward and backward — the attention inside the
 * nn.TransformerEncoderLayer stack of LearningFilters/models.py:115-135 (`Transformer`; sequences = the graph's N nodes, head width 3-8).
 * q, k, v, out, dout, dq, dk, dv: [Bt, L, heads*dk] row-major (batch_first; head h owns columns h*dk..); lse, delta: [Bt, heads, L].
 * out[b,i,h,:] = sum_j softmax_j(q_i . k_j / sqrt(dk)) v_j (online softmax); lse = the row's log-sum-exp, kept for the backward, which
 * recomputes the probabilities (dq per query, dk / dv per key: no atomics, reproducible); delta is scratch (dout . out).  dk <= 32. */
int sn_dense_attention_f32(const float* q, const float* k, const float* v, int64_t Bt, int L, int heads, int dk, float* out, float* lse,
                           void* stream);
int sn_dense_attention_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* lse, const float* dout,
                               int64_t Bt, int L, int heads, int dk, float* dq, float* dk_out, float* dv, float* delta, void* stream);

/* BasisNet preprocessing on the device — replaces the module-level code of LearningFilters/training.py:47-73.
 * sn_eigenspace_group: `around(eigvals, decimals)` (round-half-even of x*10^decimals, fp32 as torch evaluates it), `unique(...,
 *   return_counts)` as run-length grouping (eigvals ascending, what eigh returns; else meta[2] bit 0), and the order in which the
 *   reference stacks the projectors: by multiplicity ascending, eigenvalue order inside a multiplicity.  All outputs device int32:
 *   space_of[N] (eigenspace of every eigenvector), space_start[N+1] (first eigenvector of every eigenspace, [n_spaces] = N),
 *   space_mult[N], space_slot[N] (position of the eigenspace in the multiplicity-major stack), mult_list[N] / mult_count[N] (sorted
 *   distinct multiplicities and how many eigenspaces have each), meta[4] = {n_spaces, n_mults, error bits, largest multiplicity}.
 *   One workgroup; N <= 8192.  A one-off per graph: the caller reads meta / mult_list / mult_count back once to size its tensors.
 * sn_eigenspace_projectors_f32: out[space_slot[s], :, :] = V_s V_s^T (`projectors = [V @ V.T ...]`, `torch.cat` by multiplicity):
 *   out is the [n_spaces, N, N] stack; eigvecs row-major [N, ldv] with V[node, eigenvector] (training.py:42-43).
 * sn_ign_contract_eigvecs_f32: the five 2->1 contractions of every P_s = V_s V_s^T (contractions_2_to_1, ign.py:344-374, what
 *   sn_ign_contract_2to1_f32 computes from the projector) evaluated from V_s alone — out [n_spaces, N, 5] in slot order; reads
 *   4*N*mult bytes per eigenspace instead of 4*N^2.  Same maths, different fp32 summation order. */
int sn_eigenspace_group(const float* eigvals, int N, int decimals, int32_t* space_of, int32_t* space_start, int32_t* space_mult,
                        int32_t* space_slot, int32_t* mult_list, int32_t* mult_count, int32_t* meta, void* stream);
int sn_eigenspace_projectors_f32(const float* eigvecs, int N, int ldv, const int32_t* space_start, const int32_t* space_slot,
                                 int n_spaces, float* out, void* stream);
int sn_ign_contract_eigvecs_f32(const float* eigvecs, int N, int ldv, const int32_t* space_start, const int32_t* space_slot,
                                int n_spaces, int max_mult, float* out, void* stream);

/* GatedGCN edge-gated aggregation.  Replaces the DGL message passing of GatedGCNLayer.forward
 * (GraphPrediction/layers/gatedgcn_layer.py:51-56: apply_edges(u_add_v) + sigmoid + two update_all sums):
 *   e_out[e,:] = Dh[src(e),:] + Eh[dst(e),:] + Ce[e,:];   sigma = sigmoid(e_out)
 *   h_out[i,:] = Ah[i,:] + (sum_{e -> i} sigma_e * Bh[src(e),:]) / (sum_{e -> i} sigma_e + 1e-6)
 * over sn_batch_plan's destination-sorted CSR (rowptr / col = source / eperm = edge id); den_out (may be NULL) keeps the
 * per-node sum of gates for the backward.  ldn = row stride of Ah/Bh/Dh/Eh (column blocks of one [N,4C] GEMM output when the
 * four Linears are evaluated as one).  Optional fused eval-mode epilogue (all of h_scale/h_shift/e_scale/e_shift, or none):
 * h_out = [h_res +] relu(h_scale*h + h_shift), e_out = [e_res +] relu(e_scale*e + e_shift) — bn_node_h / bn_node_e folded, ReLU,
 * residual (gatedgcn_layer.py:64-72).  sn_gated_aggregate_bwd_f32: gradients w.r.t. Bh, Dh, Eh (dB, dD, dE, [N,C]) and
 * de_new [E,C] = the gradient of e_out's three addends (= d Ce; d Ah = dh), from dh [N,C] and de [E,C] (NULL = 0); two passes,
 * by destination and by source (reverse CSR = sn_batch_plan of the flipped edge_index), no atomics.  scratch: float[N*C]. */
int sn_gated_aggregate_f32(const float* Ah, const float* Bh, const float* Dh, const float* Eh, int ldn, const float* Ce, int64_t N,
                           int C, const int32_t* rowptr, const int32_t* col, const int32_t* eperm, float* h_out, float* e_out,
                           float* den_out, const float* h_scale, const float* h_shift, const float* e_scale, const float* e_shift,
                           const float* h_res, const float* e_res, void* stream);
int sn_gated_aggregate_bwd_f32(const float* Ah, const float* Bh, const float* e_new, const float* h_new, const float* den,
                               const float* dh, const float* de, int64_t N, int C, const int32_t* rowptr, const int32_t* col,
                               const int32_t* eperm, const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm,
                               float* dB, float* dD, float* dE, float* de_new, float* scratch, void* stream);

/* ==========================================================================================
 * Fused stages (eval mode: BatchNorm folded to per-channel scale/shift).
 *
 * Work is cut into *bins* of 64 activation rows that one workgroup keeps on chip for a whole stage.  A bin
 * holds whole *units* — the rows that exchange data inside the stage — and sn_batch_plan lays them out on the
 * device (sn_plan_bins above), with no host synchronisation:
 *   phi: unit = one (graph, eigenvector slot) slab of n_graph rows (GIN aggregation).  Graphs are packed into
 *        columns of <= 64 rows by best-fit-decreasing on n_graph; a column of height max K_g yields that many bins.
 *   rho: unit = one node's K_g slot rows (attention over slots), padded to a multiple of 16 rows.
 *   The GINE stage needs no bins: one workgroup per graph.
 * meta[1] / meta[5] != 0: a graph has more than 64 nodes / slots — the stage cannot run fused (the caller uses the
 * layer-at-a-time entry points).
 */

/* phi(x) + phi(-x) for every valid (node, slot) row, all L layers in one launch.
 * Replaces the whole of GNN3d.forward called twice (sign_net.py:28-44,113 /
 * core/sign_net.py:30-48,115): per layer GINConv aggregate (masked_layers.py:75) -> MaskedMLP
 * (:54-64) -> mask -> MaskedBN -> ReLU -> +previous.  Activations never leave the CU between
 * layers: the slab rows live in registers (MFMA operand layout) and are exchanged through LDS for
 * the neighbour sums.  All per-channel vectors are zero-padded to d_pad = 16*ceil(d/16) floats;
 * the [d, d] Linears are split-packed (sn_pack_split_f32) with their epilogue vectors.
 *   layer 0 takes the scalar eigenvector entry: Linear(1 -> hid0) [BN, ReLU] Linear(hid0 -> d)
 *   with hid0 == 1 (GINESignNetPyG, core/sign_net.py:20) or hid0 == d (Alchemy, sign_net.py:20).
 * out: [N*K, d] rows (row = node*K + slot); only valid rows are written. */
typedef struct {
  const void* w1s;   /* MaskedMLP.layers[0], split-packed [d, d] with (e0, e1) = MaskedMLP.norms[0] folded (scale, shift) */
  const void* w2s;   /* MaskedMLP.layers[1], split-packed [d, d] with (e0, e1, e2) = (bias or 0, GNN3d.norms[l] scale, shift) */
  const float* eps;  /* device scalar */
} sn_phi_layer;

#define SN_PHI_MAX_LAYERS 16
typedef struct {
  int d, n_layers, hid0, reserved; /* reserved: 0 for sn_phi_fused_f32; sn_deepsigns_phi_f32: the output width (phi_out_dim) */
  const float* l0_w1;        /* [hid0_pad] : Linear(1 -> hid0).weight[:, 0] */
  const float* l0_bn0_scale; /* [hid0_pad] */
  const float* l0_bn0_shift;
  const void* l0_w2;         /* hid0 == 1: float [d_pad] = Linear(1 -> d).weight[:, 0];
                                else split-packed [d, d] with (e0, e1, e2) = (bias or 0, bn scale, bn shift) */
  const float* l0_bias2;     /* hid0 == 1 only; may be NULL */
  const float* l0_bn_scale;  /* hid0 == 1 only */
  const float* l0_bn_shift;  /* hid0 == 1 only */
  const float* l0_eps;
  sn_phi_layer layers[SN_PHI_MAX_LAYERS - 1]; /* layers 1 .. n_layers-1 */
} sn_phi_params;

#define SN_PHI_BIN_ROWS 64
int sn_phi_fused_f32(const sn_phi_params* params /* host struct of device pointers */,
                     const float* eigen_vectors, const int32_t* graph_ptr, const int64_t* evoff,
                     const int32_t* rowptr, const int32_t* col, const sn_plan_bins* bins,
                     int kmax /* as given to sn_batch_plan */, int K /* row stride of out in slots */,
                     float* out, void* stream);

/* The sign-invariant encoder of the DGL tree, enc(g, x) + enc(g, -x) with enc = GIN (GraphPrediction/layers/deepsigns.py:45-46 /
 * :72-73, layers/gnns.py:81-114, layers/mlp.py:37-56), eval mode, ONE launch — the same stage kernel as sn_phi_fused_f32 with the
 * DGL layer order.  Per GIN layer l the caller folds the eval BatchNorms at pack time and passes, all split-packed on widths
 * zero-padded to d (a multiple of 16, 48..112):
 *   layer 0:   l0_w1 = lins.0.weight[:, 0] (Linear(1 -> hidden)), (l0_bn0_scale, l0_bn0_shift) = (1, lins.0.bias): relu(a*w + b);
 *              l0_w2 = lins.1 with the MLP's BatchNorm (after the ReLU) folded in, (e0, e1, e2) = (bias', s, t) where (s, t) is the
 *              folded enc.bns[0] (applied in front of layer 1; (1, 0) after the last layer):  out = (W1' h + e0) * e1 + e2
 *   layer l>0: w1s = lins.0 with (e0, e1) = (1, bias);  w2s as l0_w2 above.           No ReLU and no residual after the second Linear.
 * x: the dense positional encodings [N, ldx] (row = node, K columns used).  All K slots of every node are evaluated (zero-padded
 * columns of small graphs included): sn_batch_plan must have been called with kmax = -K ("full slots").  out: [N*K, params->reserved]
 * (row = node*K + slot), reserved = phi_out_dim <= d.  Graphs of more than 64 nodes: meta[1] != 0, nothing is written. */
int sn_deepsigns_phi_f32(const sn_phi_params* params, const float* x, int ldx, const int32_t* graph_ptr, const int32_t* rowptr,
                         const int32_t* col, const sn_plan_bins* bins, int K, float* out, void* stream);

/* A whole MLP over matrix rows in one launch: y = W_{L-1} relu(... relu(W_0 x + b_0) ...) + b_{L-1} — rho of the DGL sign-invariant
 * nets (layers/deepsigns.py:47-49 and :81-84 with layers/mlp.py:37-56; eval BatchNorms, which sit between a ReLU and the next
 * Linear, folded into that Linear by the caller).  weights: host array of n_layers device pointers, each sn_pack_split_f32 of the
 * [d_pad, d_pad] zero-padded matrix with e0 = bias.  nvalid == NULL: row r of the input is x[r*ldx + 0..d_in); else (masked
 * variant) it is sum_{s < nvalid[r]} x[(r*K + s)*ldx + 0..d_in) — the masked sum over the slot axis (deepsigns.py:76-81). */
#define SN_MLP_MAX_LAYERS 16
int sn_mlp_chain_f32(const float* x, int ldx, int64_t R, int d_in, const int32_t* nvalid, int K, const void* const* weights,
                     int n_layers, int d_pad, float* y, int ldy, int d_out, void* stream);

/* rho: the set-transformer encoder layers over each node's valid slots and the sum over slots, one launch.
 * Replaces SetTransformer.forward up to torch.sum(x, dim=1) (sign_net.py:60-70 / core/sign_net.py:64-75)
 * with its TransformerEncoderLayer stack (transformer_module.py:27-127, 4 heads, post-LN, eps 1e-6) and,
 * when has_pos, the eigenvalue encoder MaskedMLP(1->1->d) added to x (Alchemy sign_net.py:86,108,62).
 * Weight matrices split-packed (sn_pack_split_f32), vectors zero-padded to d_pad.  Bins: sn_plan_bins.rho_bin0.
 *   x:       [N*K, d] = phi(x)+phi(-x) (row = node*K + slot; only valid rows are read)
 *   out_sum: [N, d]   sum over the node's valid slots of the last encoder layer's output            */
typedef struct {
  const void *wq, *wk, *wv, *wfc;  /* split-packed [d,d], no bias (transformer_module.py:67-70) */
  const float *ln1_g, *ln1_b;      /* slf_attn.norm.ln */
  const void *w1, *w2;             /* pos_ffn.w_1 / w_2, split-packed with e0 = their bias */
  const float *ln2_g, *ln2_b;      /* pos_ffn.norm.ln */
} sn_rho_layer;

#define SN_RHO_MAX_LAYERS 8
#define SN_RHO_BIN_ROWS 64
typedef struct {
  int d, n_layers, heads, has_pos;
  float ln_eps;
  int head_pad;              /* 0: natural channel layout.  16 / 32: every weight and vector is zero-padded to heads*head_pad
                                channels, and the OUTPUT rows of wq/wk/wv and the INPUT columns of wfc are permuted so that head h
                                owns channels [h*head_pad, h*head_pad + d/heads): the attention then runs in registers for any
                                head width (used when d/heads is not a multiple of 16 and kmax <= 16) */
  const float* pe_w1;        /* [>=1]  eigen_encoder.layers.0.weight */
  const float* pe_bn0_scale; /* [>=1] */
  const float* pe_bn0_shift;
  const float* pe_w2;        /* [d_pad] eigen_encoder.layers.1.weight[:, 0] */
  const float* pe_bn1_scale; /* [d_pad] */
  const float* pe_bn1_shift;
  sn_rho_layer layers[SN_RHO_MAX_LAYERS];
} sn_rho_params;

int sn_rho_fused_f32(const sn_rho_params* params, const float* x, const float* eigen_values,
                     const int32_t* graph_ptr, int64_t B, int64_t N, const sn_plan_bins* bins, int kmax, int K,
                     float* out_sum, void* stream);

/* The GINE network on top of the positional encoding, one launch: rho's output Linear+BatchNorm on the slot
 * sum (sign_net.py:71), then GNN.forward (model.py:36-64 / core/model.py:44-79): input encoder
 * (DiscreteEncoder elements.py:31-37 or MLP(F,d,1)), Linear(cat[x,pos]), n_layers x [edge encoder, GINEConv
 * (pyg_gnn_wrapper.py:19-28), BatchNorm, ReLU, +previous], add pooling, 2-layer output encoder.
 * One workgroup per graph (graphs of more than SN_GNN_MAX_NODES = 64 nodes or 192 edges are skipped and flagged in
 * status[3] (bits 0 / 1); the caller then uses the layer-at-a-time entry points).  A graph that cannot be evaluated — too large,
 * a feature id out of range (bit 2), or an earlier stage of the same batch flagged in `flags_src` (malformed batch, phi / rho work
 * bins not laid out) — gets a NaN output row: nothing uninitialised is ever handed back.  Every [d,d] Linear is split-packed
 * (sn_pack_split_f32) together with its epilogue vectors; other vectors are zero-padded to d_pad. */
typedef struct {
  const float* etab[10];  /* discrete edge encoder: embedding tables [edge_vocab, d] of layer l */
  const float* ew;        /* float edge encoder: weight [d_pad, F_e] row-major (NOT packed), zero padded rows */
  const float* e_scale;   /*   its folded BatchNorm */
  const float* e_shift;
  const void* w1s;        /* GINE nn.layers.0, split-packed with (e0, e1) = nn.norms.0 folded (scale, shift) */
  const void* w2s;        /* GINE nn.layers.1, split-packed with (e0, e1) = gnn.norms[l] folded (scale, shift) */
  const float* eps;       /* device scalar */
} sn_gnn_layer;

#define SN_GNN_MAX_LAYERS 16
#define SN_GNN_MAX_NODES 64
typedef struct {
  int d, n_layers, n_out, reserved;
  int node_discrete, node_nf; /* discrete: number of int64 feature columns (<=10); float: F (<=16) */
  int edge_discrete, edge_nf;
  int node_vocab, edge_vocab; /* rows of every discrete node / edge table (DiscreteEncoder's max_num_values, elements.py:22): a feature
                                 value outside [0, vocab) — nn.Embedding raises IndexError there — is never dereferenced; the graph's
                                 output row is NaN and status[3] bit 2 is set */
  const float* ntab[10];      /* discrete node encoder tables [node_vocab, d] */
  const float* nw;            /* float node encoder: weight [d_pad, F] row-major (NOT packed), zero padded rows */
  const float* n_scale;
  const float* n_shift;
  const void* rho_out_w;      /* must be NULL: rho.out (Linear + eval BatchNorm, sign_net.py:71) is folded by the caller into lin_b —
                                 one dependent GEMM stage and one barrier fewer in the per-graph chain (field kept for layout) */
  const void* lin_a;          /* gnn.linear.weight[:, :d], split-packed, no epilogue vectors */
  const void* lin_b;          /* W_pos . diag(scale) . W_out with e0 = W_pos . shift + gnn.linear.bias, split-packed
                                 (W_pos = gnn.linear.weight[:, d:], W_out = sign_net.rho.out.0.weight, (scale, shift) = rho.out.1 folded) */
  const void* head_w1;        /* output_encoder.layers.0, split-packed with (e0, e1) = output_encoder.norms.0 folded */
  const void* head_w2;        /* output_encoder.layers.1 [n_out, d], split-packed with e0 = its bias */
  sn_gnn_layer layers[SN_GNN_MAX_LAYERS];
} sn_gnn_params;

int sn_gnn_fused_f32(const sn_gnn_params* params, const void* x, int ldx, const void* edge_attr, int lde,
                     const float* rho_sum, const int32_t* graph_ptr, int64_t B, const int32_t* rowptr,
                     const int32_t* col, const int32_t* eperm, int32_t* status /* sn_batch_plan's */,
                     float* y /* [B, n_out] */,
                     const int32_t* flags_src, int n_flags, int32_t* flags_host /* optional report, see below */,
                     void* stream);

/* Flag report without a separate copy: when flags_host (device-accessible pinned host memory) is not NULL, the last
 * workgroup to finish copies flags_src[0 .. n_flags) (device; typically sn_batch_plan's status words followed by
 * sn_plan_bins.meta) to it, after every workgroup's own flag updates, and then — behind a system-scope fence — stores 1 to
 * flags_host[n_flags] (so flags_host holds n_flags + 1 ints).  A caller that zeroed that word before the launch can poll
 * it from the host: once it reads 1 the flags are there; no event / marker packet on the stream is needed. */

/* The DGL tree's GIN base net, eval mode, one launch on the same per-graph stage kernel (GraphPrediction/nets/ZINC_graph_regression/
 * gin_net.py:83-126 with layers/gin_layer.py and layers/mlp.py:37-56, layers/mlp_readout_layer.py):
 *   h = embedding_h[atom] + embedding_p(p);  L x  h = MLP((1 + eps) h_i + sum_{j -> i} h_j);  readout sum / mean;  MLPReadout.
 * params (all widths zero-padded to d = 64, 96 or 128 by the caller; the same struct as sn_gnn_fused_f32):
 *   node_discrete = 1, node_nf = 1, ntab[0] = embedding_h.weight [node_vocab, d]; edge_nf = 0; rho_out_w = NULL;
 *   lin_a = split-packed identity, lin_b = split-packed embedding_p.weight ([d, d], columns >= kp zero) with e0 = its bias;
 *   layers[l]: w1s = MLP.lins.0 with (e0, e1) = (1, bias), w2s = MLP.lins.1 with the MLP's eval BatchNorm (which follows the ReLU) folded
 *              in and (e0, e1) = (1, bias'); eps = the layer's eps scalar; nothing follows the second Linear;
 *   head_w1, head_mid: MLPReadout.FC_layers.0 / .1 with (e0, e1) = (1, bias); head_w2 = FC_layers.2 with e0 = bias; n_out = 1.
 * atom: int64 [N]; p: [N, ldp] positional encoding, kp columns used; pool_mean: readout 'mean' (else 'sum').  Graphs of more than 64 nodes
 * or 192 in-edges, or an atom type outside the table: NaN row + status[3] bits, as sn_gnn_fused_f32. */
int sn_gin_net_fused_f32(const sn_gnn_params* params, const void* head_mid, int pool_mean, const int64_t* atom, const float* p, int ldp,
                         int kp, const int32_t* graph_ptr, int64_t B, const int32_t* rowptr, const int32_t* col, const int32_t* eperm,
                         int32_t* status, float* y, const int32_t* flags_src, int n_flags, void* stream);

/* The DGL tree's sparse graph Transformer base net, eval mode, one launch on the same per-graph stage kernel (GraphPrediction/nets/
 * ZINC_graph_regression/transformer_net.py:88-140 with layers/transformer.py:150-312; the shipped shape: hidden 64, 8 heads, residual,
 * BatchNorm):
 *   h = embedding_h[atom] + embedding_p(p)  (or pe_proj(cat[.,.]): lin_a = pe_proj.weight[:, :64], lin_b = the two maps behind p folded);  L x [ Q, K, V -> edge attention with E_ij -> x1 = BatchNorm(h + O_h(a)) ->
 *   h = BatchNorm(x1 + FFN2(relu(FFN1(x1)))) ];  readout sum / mean;  MLPReadout.
 * params: as sn_gin_net_fused_f32 with d = 64 (input stage, head, node table); layers[l].etab[0..7] = the layer's eight [64, 64] stage
 * matrices, split-packed: Q, K, V (no epilogue vectors), O_h with (e0, e1, e2) = (bias, BatchNorm1 scale, shift), FFN_h_layer1 rows
 * [0, 64) and [64, 128) with e0 = their bias halves, FFN_h_layer2 columns [0, 64) (no vectors) and [64, 128) with (e0, e1, e2) = (bias,
 * BatchNorm2 scale, shift); layers[l].w1s / w2s repeat etab[0] / etab[1].  e_proj: [E, lde] fp32 in edge-id order, columns
 * [l * 64, (l + 1) * 64) = layer l's E projection of the edge embedding (one Linear over all layers, computed by the caller).
 * Limits and error reporting as sn_gin_net_fused_f32. */
int sn_transformer_net_fused_f32(const sn_gnn_params* params, const void* head_mid, int pool_mean, const int64_t* atom, const float* p,
                                 int ldp, int kp, const float* e_proj, int lde, const int32_t* graph_ptr, int64_t B, const int32_t* rowptr,
                                 const int32_t* col, const int32_t* eperm, int32_t* status, float* y, const int32_t* flags_src, int n_flags,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training-step stage kernels (BASELINE configs[3]).  They replace, for one "Linear -> BatchNorm1d(train) -> ReLU"
 * link of a MaskedMLP / MLP (Alchemy/sign_net/model_utils/masked_layers.py:34-64, GINESignNetPyG/core/model_utils/elements.py:40-69),
 * what loss.backward() (Alchemy/main_alchemy.py:108, GINESignNetPyG/core/train.py:62) runs through ATen: one pass over the rows per
 * direction instead of ~10.  Rows come in G groups of R rows, group-major (the phi(+x) / phi(-x) passes of sign_net.py:113 share every
 * weight but keep separate batch statistics); row r of a group is valid iff nvalid == NULL or (r % K) < nvalid[r / K].  Weights are the
 * RAW nn.Linear parameters (row-major [d_out, d_in], any alignment): nothing is packed per step.  Widths: multiples of 4 up to 128.
 *
 * sn_train_linear_f32: y = [relu](x_hat W^T + b) on valid rows, 0 elsewhere, x_hat = [relu](x * in_scale[g] + in_shift[g]) (the
 *   producer's train-mode BatchNorm folded to an affine, applied as the operand tile is loaded; NULL: x_hat = [relu] x).  With stat_part
 *   (float[G * (2 * nblk * d_out + nblk)], nblk = sn_train_linear_blocks(R, G)) the batch moments of y over the valid rows of every
 *   group are taken from the accumulators; sn_train_bn_finish_f32 merges them:
 *     state[0..4][g][C] = mean, biased variance, rstd, scale = gamma * rstd, shift = beta - mean * scale (every component one
 *     contiguous [G, C] block: state + 3*G*C is the in_scale of the consumer);  count[g] = valid rows;
 *     running statistics updated once per group, in group order (momentum, unbiased variance), when given.
 * sn_train_linear_bwd_f32: the adjoint of one link in one pass.
 *     g  = dy * [mask_scale * zo + mask_shift > 0]     (the ReLU behind this Linear's BatchNorm; mask_* NULL: g = dy)
 *     dz = coef_a * g - coef_b - coef_c * zo           (that BatchNorm's backward, sn_train_bn_bwd_finish_f32; coef_* NULL: dz = g)
 *     x_hat = [relu](x * x_scale + x_shift) or x       (the operand the forward link formed on load)
 *     gx = (dz W) * [x_hat > 0 if x_relu]              -> gx [G*R, d_in]  (NULL: no input gradient)
 *     sums_part[g][blk][0][c] = sum gx, [1][c] = sum gx * (x - x_mean[g])   (x_mean != NULL; for the producer's BatchNorm backward)
 *     dw_part[g*nblk + blk] = sum dz^T x_hat (d_out * d_in floats) followed by sum dz (d_out floats, want_db) — per-workgroup
 *       partials, nblk = sn_train_linear_bwd_blocks(R, G); sn_train_reduce_parts_f32 adds them in block order.  No atomics anywhere.
 * sn_train_bn_bwd_sums_f32: the same column sums for a BatchNorm whose upstream gradient does not come from
 *   sn_train_linear_bwd_f32 (its output feeds an aggregation / a residual): g = dy * [scale * z + shift > 0 if relu];
 *   sums_part: float[G * sn_train_bn_bwd_blocks(R, G) * 2 * C].
 * sn_train_bn_bwd_finish_f32: coef[0..2][g][C] (a, b, c above) from the partial sums, and d gamma / d beta summed over the groups
 *   (written, or added when accumulate != 0).
 * sn_train_bn_apply_f32: y = [relu](z * scale[g] + shift[g]) [+ residual] on valid rows, 0 elsewhere (`state` as written by
 *   sn_train_bn_finish_f32) — the last BatchNorm of a stack, whose output is materialised.
 * sn_train_reduce_parts_f32: out[i] (+)= sum_b part[b * stride + i], i < n. */
typedef struct {
  const float* x; int ldx;
  int64_t R; int G;
  int d_in, d_out;
  const float* W; int ldw;
  const float* bias;
  const int32_t* nvalid; int K;
  const float* in_scale; const float* in_shift;   /* [G, d_in] or NULL */
  int in_relu, out_relu;
  float* y; int ldy;
  float* stat_part;                               /* NULL: no statistics */
  /* ABI 2 — the BatchNorm finish inside the same launch (fin_state != NULL; needs stat_part, d_out <= 128): the LAST workgroup to
   * arrive (an agent-scope ticket behind write-through stores of the partials) merges the moments in block order — the result does
   * not depend on who is last; the slicing and order of sn_train_bn_finish_f32 (equal up to the compiler's FMA contraction: last-bit
   * differences) — and writes fin_state / fin_count / the running statistics.  Measured SLOWER than the separate launch in a
   * replayed HIP graph; the Python side uses it only under SN_TRAIN_FUSE_FINISH=1. */
  const float* fin_gamma; const float* fin_beta; float fin_eps, fin_momentum;
  float* fin_running_mean; float* fin_running_var;
  float* fin_state; float* fin_count;
} sn_train_linear_args;

typedef struct {
  int64_t R; int G;
  const int32_t* nvalid; int K;
  int d_in, d_out;
  const float* dy; int lddy;
  const float* zo; int ldzo;
  const float* coef_a; const float* coef_b; const float* coef_c;   /* [G, d_out] or NULL */
  const float* mask_scale; const float* mask_shift;                /* [G, d_out] or NULL */
  const float* x; int ldx;
  const float* x_scale; const float* x_shift;                      /* [G, d_in] or NULL */
  int x_relu;
  const float* x_mean;                                             /* [G, d_in] or NULL */
  const float* W; int ldw;
  float* gx; int ldgx;
  float* sums_part;
  float* dw_part;
  int want_db;
  int gx_accumulate;               /* gx += instead of gx =: several Linears share one operand (q, k, v of the attention); excludes x_mean / dot_x */
  const float* dot_x; int lddot;   /* optional [G*R, d_in]: dot_part[g*nblk + blk] = sum over the block's rows of gx . dot_x — summed by the */
  double* dot_part;                /*   caller it is the eps gradient of the GIN / GINE aggregation that produced the operand (float64: a cancelling sum) */
  /* ABI 2 — what sn_train_post_link_f32 did behind the link, inside the link's own launch by its last-arriving workgroup (same
   * slicing and order; equal up to FMA contraction): fin_coef != NULL: the BatchNorm-backward finish of the PRODUCER link from sums_part (arguments of
   * sn_train_bn_bwd_finish_f32; C = d_in <= 128; d gamma / d beta written, or added when fin_accumulate);  fin_dot_out != NULL:
   * fin_dot_out[0] += the eps gradient (the sum of dot_part). */
  const float* fin_state; const float* fin_count; const float* fin_gamma;
  float* fin_coef; float* fin_dgamma; float* fin_dbeta; int fin_accumulate;
  float* fin_dot_out;
  /* ABI 2 — the CONSUMER-side form: merge_sums != NULL: coef a | b | c of THIS link's BatchNorm backward are not given (coef_* NULL) but
   * merged by the link itself, in its prologue, from the column-sum partials of sn_train_bn_bwd_sums_f32 / of the consumer link's
   * sums_part (float[G * merge_nblk * 2 * d_out]) with the statistics merge_state ([5][G][d_out]) / merge_count ([G]) of the forward:
   * no finish launch in front of the link.  Workgroup 0 writes d gamma / d beta (added when merge_accumulate). */
  const float* merge_sums; int merge_nblk;
  const float* merge_state; const float* merge_count; const float* merge_gamma;
  float* merge_dgamma; float* merge_dbeta; int merge_accumulate;
} sn_train_linear_bwd_args;

/* What follows a backward link, in ONE launch (a block does one job): the reduction of the link's per-workgroup dW (and db) partials
 * ADDED to the parameters' gradients (mandatory), the BatchNorm-backward finish of the producer link (optional: sums_part != NULL;
 * arguments of sn_train_bn_bwd_finish_f32) and the eps-gradient finish (optional: dot_part != NULL; added to dot_out[0]). */
typedef struct sn_train_post_args {
  const float* dw_part; int nparts; int64_t stride; int64_t n_w; float* dw_out; int64_t n_b; float* db_out;
  const float* sums_part; int nblk; int G; int C; const float* state; const float* count; const float* gamma; float* coef;
  float* dgamma; float* dbeta; int accumulate_bn;
  const double* dot_part; int dot_n; float* dot_out;
} sn_train_post_args;
int sn_train_post_link_f32(const sn_train_post_args* args, void* stream);
int sn_train_linear_blocks(int64_t R, int G);
int sn_train_linear_bwd_blocks(int64_t R, int G);
int sn_train_bn_bwd_blocks(int64_t R, int G);
int64_t sn_train_linear_bwd_part_floats(int64_t R, int G, int d_in, int d_out);
int sn_train_linear_f32(const sn_train_linear_args* args, void* stream);
int sn_train_bn_finish_f32(const float* stat_part, int nblk, int G, int C, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* state, float* count, void* stream);
int sn_train_linear_bwd_f32(const sn_train_linear_bwd_args* args, void* stream);
int sn_train_bn_bwd_sums_f32(const float* dy, int lddy, const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid,
                             int K, const float* state, int relu, float* sums_part, void* stream);
int sn_train_bn_bwd_finish_f32(const float* sums_part, int nblk, int G, int C, const float* state, const float* count,
                               const float* gamma, float* coef, float* dgamma, float* dbeta, int accumulate, void* stream);
int sn_train_bn_apply_f32(const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid, int K, const float* state,
                          int relu, const float* residual, int ldr, float* y, int ldy, void* stream);
int sn_train_reduce_parts_f32(const float* part, int nparts, int64_t stride, int64_t n, float* out, int accumulate, void* stream);
/* The same reduction for up to SN_TRAIN_MAX_REDUCE_JOBS (part, nparts, stride, n, out) tuples in ONE launch: the dW / db partials of
 * every backward link of a step, reduced once at the end of loss.backward() (GINESignNetPyG/core/train.py:62) instead of one launch
 * per link; out[i] (+)= sum_b part[b * stride + i] in block order, job by job independent.  Same arithmetic as the single form. */
#define SN_TRAIN_MAX_REDUCE_JOBS 64
typedef struct { const float* part; int nparts; int64_t stride; int64_t n; float* out; int accumulate; } sn_train_reduce_job;
int sn_train_reduce_jobs_f32(const sn_train_reduce_job* jobs, int njobs, void* stream);
/* out[0] += (float) sum of n float64 partials, for up to SN_TRAIN_MAX_REDUCE_JOBS (part, n, out) in ONE launch: the eps gradients of every
 * aggregation of a step (sn_train_linear_bwd_f32's dot_part), added once at the end of loss.backward(). */
typedef struct { const double* part; int n; float* out; } sn_train_dot_job;
int sn_train_dot_jobs_f64(const sn_train_dot_job* jobs, int njobs, void* stream);
/* sn_train_bn_bwd_sums_f32 + sn_train_bn_bwd_finish_f32 in one launch (C <= 128): the last workgroup to arrive finishes. */
int sn_train_bn_bwd_f32(const float* dy, int lddy, const float* z, int ldz, int64_t R, int G, int C, const int32_t* nvalid, int K,
                        const float* state, const float* count, int relu, const float* gamma, float* sums_part, float* coef,
                        float* dgamma, float* dbeta, int accumulate, void* stream);
/* out[0] (+)= (float) sum of n float64 partials (dot_part of sn_train_linear_bwd_f32: the eps gradient), one launch */
int sn_train_dot_finish_f64(const double* part, int n, float* out, int accumulate, void* stream);

/* Adjoints of the aggregations of a layer whose input also feeds a residual (x -> aggregate -> MLP, y = ... + x; GNN3d.forward
 * sign_net.py:36-43, GNN.forward model.py:52-60): d x = aggregate^T(d a) + d y in ONE pass (`plus` = d y) instead of an extra elementwise add.
 * sn_gin_aggregate_add_f32: out = sn_gin_aggregate_f32(x) + plus on the CSR given (the reverse CSR for an adjoint).
 * sn_gine_aggregate_bwd_add_f32: sn_gine_aggregate_bwd_f32 with dh += plus. */
int sn_gin_aggregate_add_f32(const float* x, const float* plus, float* out, int64_t N, int F, const int32_t* rowptr, const int32_t* col,
                             const float* eps, void* stream);
int sn_gine_aggregate_bwd_add_f32(const float* h, const float* ee, const float* g, const float* plus, int64_t N, int C,
                                  const int32_t* rev_rowptr, const int32_t* rev_col, const int32_t* rev_eperm, const float* eps,
                                  float* dh, float* dee, void* stream);

/* The 1 -> 1 -> d MaskedMLP whose input is one scalar per (node, slot) row — GINESignNetPyG's first phi layer
 * (core/sign_net.py:20: MaskedGINConv(1, n_hid, ...) -> MaskedMLP(1, n_hid) with hidden width 1, masked_layers.py:37) and its
 * eigen_encoder2 (core/sign_net.py:90,111) — in closed form (csrc/train.hip): every column of the second Linear's output is an affine
 * image of the same scalar h = relu(bn_a(w1 a)), so the batch statistics of bn_b follow from the moments of h.
 *   y[g][row][c] = [relu_b](bn_b(w2[c] h + b2[c])),   rows: G <= 2 groups of M, group 1 with a -> -a when negate_second (phi(-x)).
 * sn_train_scalar_mlp_stats_f32: scalar_state[g][8] (float64: n, mean_a, rstd_a, scale_a, shift_a, mean_h, var_h, -), column_state[g][2][d]
 *   (p_c = gamma_c w2_c / s_c, s_c = sqrt(w2_c^2 var_h + eps_b)); running statistics of both BatchNorms updated once per group in order.
 * sn_train_scalar_mlp_apply_f32: one write of y [G*M, d] (0 on invalid rows).
 * sn_train_scalar_mlp_bwd_f32: one read of dy -> d w1, d gamma_a, d beta_a (1 float each), d w2, d gamma_b, d beta_b (d floats each; the
 *   bias of the second Linear sits in front of a batch-statistics BatchNorm: its gradient is 0) written, or added when accumulate != 0,
 *   and da [M] (gradient of the scalar input, both groups).  part: float[G * sn_train_bn_bwd_blocks(M, G) * 2 * d]; row_sums: float[G * M]. */
typedef struct {
  const float* a; int64_t M; int G; int negate_second;
  const int32_t* nvalid; int K; int d;
  const float* w1; const float* gamma_a; const float* beta_a; float eps_a;
  const float* w2; const float* b2; const float* gamma_b; const float* beta_b; float eps_b; int relu_b;
  double* scalar_state;   /* float64: the scalar chain's biases are multiplied by the row count */
  float* column_state;
} sn_train_scalar_mlp_args;

int64_t sn_train_scalar_mlp_work_doubles(int64_t M, int G, int d);   /* size of the `work` buffers below */
int sn_train_scalar_mlp_stats_f32(const sn_train_scalar_mlp_args* args, float momentum_a, float* running_mean_a, float* running_var_a,
                                  float momentum_b, float* running_mean_b, float* running_var_b, double* work, void* stream);
int sn_train_scalar_mlp_apply_f32(const sn_train_scalar_mlp_args* args, float* y, void* stream);
int sn_train_scalar_mlp_bwd_f32(const sn_train_scalar_mlp_args* args, const float* dy, float* part, float* row_sums, float* dw1,
                                float* dgamma_a, float* dbeta_a, float* dw2, float* dgamma_b, float* dbeta_b, float* da,
                                int accumulate, double* work, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * IGN2to1.forward after the 2->1 contractions (LearningFilters/ign.py:29-39), eval mode, ONE launch: o [b, n, 5] ->
 * bn0(relu(W0 o + b0)) -> two equivariant 1->1 layers bn(relu(Wa h + Wb mean_n(h) + b)) (:174-214) -> fc2(relu(fc1(.))) -> y [b, O, n]
 * (already transposed as ign.py:36-39 leaves it).  One workgroup per matrix, the rows' H channels stay in registers (MFMA operand layout).
 * s*, t*: the BatchNorms folded to scale / shift (sn_bn_fold_f32); w1a / w1b: coeffs[:, :, 0]^T / coeffs[:, :, 1]^T ([H_out, H_in]).
 * sn_ign_mlp_supported: H in {16, 32}, n <= 1024, O <= 32 — otherwise the layer-at-a-time entry points serve the module.  Parameter
 * arrays 16-byte aligned. */
typedef struct sn_ign_mlp_params {
  const float *w0, *b0, *s0, *t0;                 /* [H,5], [H], [H], [H] */
  const float *w1a, *w1b, *b1, *s1, *t1;          /* [H,H], [H,H], [H] x3 */
  const float *w2a, *w2b, *b2, *s2, *t2;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;     /* [H,H], [H], [O,H], [O] (fc2_b may be NULL) */
} sn_ign_mlp_params;
int sn_ign_mlp_supported(int n, int H, int O);
int sn_ign_mlp_f32(const float* o, int64_t b, int n, int H, int O, const sn_ign_mlp_params* P, float* y, void* stream);

/* EqDeepSetsEncoder (LearningFilters/models.py:58-113) behind its first layer's Linear, for ONE set (n * widest layer but the last <= 16384), in one launch:
 * z [n, width[0]] = lin1_0(x) + lin2_0(mean x) (pre-activation; or its two halves, see split0); then for i = 1 .. n_layers-1:  h = relu(h); [BatchNorm over the n rows
 * with gamma[i-1], beta[i-1] — track_running_stats = False: batch statistics in eval too];  h = w1[i] h + b1[i] + w2[i] mean_n(h) + b2[i].
 * No ReLU / BatchNorm after the last layer.  w1 / w2 [width[i], width[i-1]] row-major; widths <= 32.  y [n, width[n_layers-1]]. */
#define SN_DEEPSETS_MAX_LAYERS 8
typedef struct sn_deepsets_tail_params {
  int n_layers, use_bn;
  float eps;
  int split0;      /* 1: z is [n, 2 width[0]] = x [W1_0 ; W2_0]^T + [b1_0 ; b2_0] and the kernel forms z[:, :w] + mean_rows(z[:, w:]) */
  int width[SN_DEEPSETS_MAX_LAYERS];
  const float* w1[SN_DEEPSETS_MAX_LAYERS]; const float* b1[SN_DEEPSETS_MAX_LAYERS];
  const float* w2[SN_DEEPSETS_MAX_LAYERS]; const float* b2[SN_DEEPSETS_MAX_LAYERS];
  const float* gamma[SN_DEEPSETS_MAX_LAYERS]; const float* beta[SN_DEEPSETS_MAX_LAYERS];
} sn_deepsets_tail_params;
int sn_deepsets_tail_f32(const float* z, int n, const sn_deepsets_tail_params* P, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGNNET_HIP_H */
