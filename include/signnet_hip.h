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

#define SN_ABI_VERSION 1

int sn_version(void);
const char* sn_last_error(void);
/* number of compute units / XCDs of the current device (for grid sizing and roofline maths) */
int sn_device_info(int* cu_count, int* lds_bytes_per_cu, int* clock_khz);

/* ------------------------------------------------------------------------------------------
 * Batch plan.  Replaces: to_dense_EVD bookkeeping (Alchemy/sign_net/transform.py:26-38),
 * `scatter(ones, batch)` size/mask construction (sign_net.py:100-102), and PyG
 * MessagePassing's per-call edge gather order (torch_geometric==2.0.1, call sites
 * masked_layers.py:75, pyg_gnn_wrapper.py:28) by a destination-sorted CSR built once per batch.
 *
 * Inputs : batch[N] int64 ascending; edge_index[2,E] int64 (row 0 = source, row 1 = target).
 * Outputs (int32, caller-allocated):
 *   graph_ptr[B+1]  first node of each graph;        node_graph[N]  graph id per node
 *   nvalid[N]       min(n_graph, kmax) per node (kmax<=0: n_graph) — number of valid slots
 *   evoff[B+1]      prefix of n_b^2   (offset of graph b's eigenvector block, int64)
 *   rowptr[N+1], col[E] (source node of each in-edge), eperm[E] (edge id, for edge_attr)
 *     — in-edges of a node are ordered by edge id (deterministic summation order)
 *   status[4]       status[0] != 0 -> malformed batch (unsorted batch, edge across graphs, ...)
 * scratch: int32[N + 8].
 */
int sn_batch_plan(const int64_t* batch, int64_t N, int64_t B, const int64_t* edge_index, int64_t E,
                  int kmax, int32_t* graph_ptr, int32_t* node_graph, int32_t* nvalid, int64_t* evoff,
                  int32_t* rowptr, int32_t* col, int32_t* eperm, int32_t* status, int32_t* scratch,
                  void* stream);

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
int sn_masked_linear_f32(const float* x, int ldx, int64_t R, int d_in, const float* Wp, int d_out,
                         const float* bias, const int32_t* nvalid, int K, int flags,
                         const float* scale, const float* shift, const float* residual, int ldr,
                         float* y, int ldy, void* stream);

/* Per-channel masked statistics for train-mode BatchNorm (MaskedBN on the compacted valid rows,
 * masked_layers.py:19; nn.BatchNorm1d model.py:50): mean[c], biased var[c] over valid rows,
 * count written to *count.  scratch: float[2*C*nblocks] (nblocks = sn_colstats_blocks(R)). */
int sn_colstats_blocks(int64_t R);
int sn_masked_colstats_f32(const float* x, int ldx, int64_t R, int C, const int32_t* nvalid, int K,
                           float* mean, float* var, float* count, float* scratch, void* stream);

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
 * softmax over the valid slots of the node; invalid query rows -> 0. */
int sn_set_attention_f32(const float* q, const float* k, const float* v, int64_t N, int K, int heads,
                         int dk, const int32_t* nvalid, float* out, void* stream);

/* out[n, :] = sum_k x[n, k, :]  (torch.sum(x, dim=1), sign_net.py:70). */
int sn_slot_sum_f32(const float* x, int64_t N, int K, int C, float* out, void* stream);

/* DiscreteEncoder (elements.py:31-37): out[r,:] = sum_f tables[f][idx[r*ldi + f], :], nf <= 10.
 * `tables` is a HOST array of nf device pointers (each table [V, C] fp32). */
int sn_embedding_sum_f32(const int64_t* idx, int ldi, int nf, int64_t R, const float* const* tables,
                         int C, float* out, void* stream);

/* Graph pooling (torch_scatter.scatter add/mean, model.py:57-61): out[b,:] over nodes of graph b.
 * mode 0 = add, 1 = mean. */
int sn_segment_pool_f32(const float* x, int64_t B, int C, const int32_t* graph_ptr, int mode,
                        float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGNNET_HIP_H */
