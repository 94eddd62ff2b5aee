/*
 * nsdp_hip.h -- C-ABI of libnsdp_hip.so, the MI355X (gfx950) drop-in for the native boundary of the
 * NSDP TDNet hot path.
 *
 * Conventions (replaces the reference contract described in SURVEY.md section 8 b1):
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - float = IEEE binary32, indices = int32_t; tensors are dense, row-major, contiguous
 *     (the reference asserts the same: _ext-src/include/utils.h:5-25 CHECK_CONTIGUOUS/IS_FLOAT/IS_INT);
 *   - outputs and scratch are allocated BY THE CALLER (the reference allocates with torch::zeros in the
 *     callee, e.g. sampling.cpp:70-76; a C ABI cannot) -- functions that the reference zero-fills
 *     (`*_grad`, ball_query) zero their output themselves, stream-ordered;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is asynchronous on it,
 *     mirroring `at::cuda::getCurrentCUDAStream()` (e.g. sampling_gpu.cu:180);
 *   - return value: 0 on success, a negative NSDP_E* code for bad arguments, or a positive hipError_t.
 *     Nothing ever calls exit() (the reference does on a launch failure, cuda_utils.h:30-39);
 *   - re-entrant, no global state besides a thread-local last-error string.
 */
#ifndef NSDP_HIP_H_
#define NSDP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSDP_EINVAL (-1)  /* bad size / null pointer */
#define NSDP_ENOSUP (-2)  /* unsupported configuration (message in nsdp_last_error) */

/* Version of this ABI (bumped on any signature change). */
int nsdp_abi_version(void);
/* Human-readable description of the last non-zero return on this thread. */
const char *nsdp_last_error(void);
/* Tuning / ablation knobs (not part of the reference contract; timing experiments only -- several produce wrong results):
 * 1 wgrad software pipelining, 3 fp32 GEMM pipeline form, 4 fp32 GEMM ablation bits, 5 float4-operand wgrad, 6 bf16x3 GEMM
 * ablation bits, 7 bf16 wgrad phases (1 no MFMA, 2 no transposition, 4 no DMA), 8 bf16 linear phases (1 no MFMA, 2 no stores).
 * 9 is a HOST HINT, not an ablation: the number of compute units nsdp_linear_wgrad_bf16x3_f32 leaves free (its persistent
 * one-wave-per-SIMD workgroups otherwise hold every CU until the kernel ends); the host sets it around weight-gradient
 * launches that run on a side stream next to the critical chain and resets it to 0 (the workspace query sees the same value).
 * The hint is THREAD-LOCAL: it applies to launches made by the host thread that set it (set / query / launch / reset happen on
 * one thread; another device's backward thread cannot clobber it).
 * 10: 0 = immediate-insertion kNN kernel; 11: 0 = three-launch BatchNorm forms (A/B against the one-launch slab kernels), 2 = slab kernels up to 16384 rows;
 * 12: 0 = one-lane-per-query ball_query / three_nn kernels (A/B against the four-lane plane-tile scans). */
void nsdp_debug_set(int key, int value);
/* Number of HIP devices visible (0 when there is none; never fails). */
int nsdp_device_count(void);

/* ----------------------------------------------------------------------------------------------
 * pointnet2_ops._ext replacements
 * (reference binding table: pointnet2_ops_lib/pointnet2_ops/_ext-src/src/bindings.cpp:6-19)
 * -------------------------------------------------------------------------------------------- */

/* furthest_point_sampling(points(B,N,3), nsamples) -> (B,nsamples) i32
 * replaces sampling.cpp:66-87 + sampling_gpu.cu:69-229.  `tmp` = (B,N) f32 scratch (only touched when
 * N > 8192; may be NULL otherwise).  Indices are identical to the reference kernel's, ties included
 * (tie rule of the block_size = opt_n_threads(N) shared-memory tree is reproduced exactly). */
int nsdp_furthest_point_sampling(const float *xyz, int B, int N, int nsamples, float *tmp,
                                 int32_t *idx_out, void *stream);

/* gather_points(points(B,C,N), idx(B,M)) -> (B,C,M); sampling.cpp:16-41, sampling_gpu.cu:8-30 */
int nsdp_gather_points(const float *points, const int32_t *idx, int B, int C, int N, int M,
                       float *out, void *stream);
/* gather_points_grad(grad_out(B,C,M), idx(B,M), N) -> (B,C,N); sampling.cpp:43-65, sampling_gpu.cu:34-57 */
int nsdp_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int M,
                            float *grad_points, void *stream);

/* group_points(points(B,C,N), idx(B,NP,NS)) -> (B,C,NP,NS); group_points.cpp:14-38, group_points_gpu.cu:8-41 */
int nsdp_group_points(const float *points, const int32_t *idx, int B, int C, int N, int NP, int NS,
                      float *out, void *stream);
/* group_points_grad(grad_out(B,C,NP,NS), idx, N) -> (B,C,N); group_points.cpp:40-65, group_points_gpu.cu:43-75 */
int nsdp_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int NP,
                           int NS, float *grad_points, void *stream);

/* The gradient of gather_points / group_points for callers that hold the inverse of the index map (nsdp_knn_invert on
 * idx viewed as (B,E): offsets (B,N+1), entries (B,E), lists ascending): grad_points(B,C,N)[b,c,s] = sum over the list of s
 * of grad_out(B,C,E)[b,c,entry] -- no atomics, deterministic, a stream over grad_out (the index set is shared by all C
 * channels, so one list build serves them all).  Same result as nsdp_group_points_grad (sampling_gpu.cu:34-57,
 * group_points_gpu.cu:43-75) up to the order of the fp32 sums.  A row of E floats that fits LDS (E <= 32768) is staged
 * whole; longer rows in slices, with a cursor per target into its (ascending) list.  N <= 32768. */
int nsdp_scatter_cm_lists_supported(int B, int C, int N, int E);
int nsdp_scatter_cm_lists(const float *grad_out, const int32_t *offsets, const int32_t *entries, int B, int C, int N,
                          int E, float *grad_points, void *stream);

/* three_interpolate_grad (interpolate_gpu.cu:116-143) without atomics: offsets / entries = nsdp_knn_invert of the (B, n, 3)
 * index map flattened to E = 3 n entries per shape (entry e = 3 j + t); grad_points[b][c][s] = sum over the list of s of
 * grad_out[b][c][e / 3] * weight[b][e], in list order (deterministic).  Rows of up to 8192 targets are staged whole, longer
 * ones in slices (cursors into the ascending lists); m <= 32768 sources. */
int nsdp_three_interpolate_grad_lists_supported(int B, int c, int n, int m);
int nsdp_three_interpolate_grad_lists(const float *grad_out, const float *weight, const int32_t *offsets,
                                      const int32_t *entries, int B, int c, int n, int m, float *grad_points, void *stream);


/* ball_query(new_xyz(B,M,3), xyz(B,N,3), radius, nsample) -> (B,M,nsample) i32;
 * ball_query.cpp:8-34, ball_query_gpu.cu:9-54 */
int nsdp_ball_query(const float *new_xyz, const float *xyz, int B, int N, int M, float radius,
                    int nsample, int32_t *idx_out, void *stream);

/* three_nn(unknown(B,n,3), known(B,m,3)) -> dist2(B,n,3) f32, idx(B,n,3) i32;
 * interpolate.cpp:17-42, interpolate_gpu.cu:9-70 */
int nsdp_three_nn(const float *unknown, const float *known, int B, int n, int m, float *dist2,
                  int32_t *idx, void *stream);
/* three_interpolate(points(B,c,m), idx(B,n,3), weight(B,n,3)) -> (B,c,n); interpolate.cpp:44-72, interpolate_gpu.cu:72-114 */
int nsdp_three_interpolate(const float *points, const int32_t *idx, const float *weight, int B,
                           int c, int m, int n, float *out, void *stream);
/* three_interpolate_grad(grad_out(B,c,n), idx, weight, m) -> (B,c,m); interpolate.cpp:74-101, interpolate_gpu.cu:116-156 */
int nsdp_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                                int B, int c, int n, int m, float *grad_points, void *stream);

/* ----------------------------------------------------------------------------------------------
 * ATen call sites of the model that the path replaces with native kernels
 * -------------------------------------------------------------------------------------------- */

/* kNN: replaces `square_distance(query, source).argsort()[:, :, :k]`
 * (model/utils.py:39-55; call sites model/encoder/blocks.py:101-102, :287-288, model/decoder/blocks.py:50-52).
 * query(B,n,3), source(B,m,3) -> idx(B,n,k) i32 ascending by (distance, index); distance is
 * ((dx*dx + dy*dy) + dz*dz) in fp32 with separately rounded products, bit-identical to the reference.
 * dist2_out(B,n,k) may be NULL.  Never materialises the n x m matrix.  k <= 64, k <= m. */
int nsdp_knn(const float *query, const float *source, int B, int n, int m, int k, int32_t *idx_out,
             float *dist2_out, void *stream);

/* index_points(points(B,N,C), idx(B,S)) -> (B,S,C) (row gather; model/utils.py:58-70) */
int nsdp_gather_rows(const float *points, const int32_t *idx, int B, int N, int C, int S, float *out,
                     void *stream);
/* rel4 (B,n,k,4) = (sign * (query[b,i] - source[b, idx[b,i,j]]), 0): the relative coordinates an attention block feeds to its
 * position-encoding MLP (model/encoder/blocks.py:104-106, :285-286, model/decoder/blocks.py:72-78: index_points + broadcast
 * subtraction), already zero-padded to the K = 4 layer's 16-byte rows.  query (B,n,3), source (B,m,3), idx (B,n,k) int32,
 * sign = +1 (query - source) or -1 (source - query).  One rounding per component, as torch.sub.  No gradient: for coordinates
 * that need one the host keeps the differentiable gather + subtraction. */
int nsdp_rel_coords4(const float *query, const float *source, const int32_t *idx, int B, int n, int m, int k, float sign,
                     float *out4, void *stream);
/* backward of index_points: grad_points(B,N,C) += scatter of grad_out(B,S,C) (zero-filled first). */
int nsdp_scatter_add_rows(const float *grad_out, const int32_t *idx, int B, int N, int C, int S,
                          float *grad_points, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Dense layers of the path on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32)
 * replace the ATen addmm calls behind nn.Linear / 1x1 nn.Conv1d in model/encoder/blocks.py and
 * model/decoder/{blocks,crosstransformer_decoder}.py (e.g. blocks.py:114-117, decoder/blocks.py:78-88,
 * crosstransformer_decoder.py:63-69).
 * -------------------------------------------------------------------------------------------- */

/* Y[M,N] = post( pre(X)[M,K] * W[N,K]^T + bias[N] (+ residual[M,N]) );
 * pre(X) = X, relu(X) (relu_in) and/or X * (mask[M,K] > 0) (ReLU backward of the producing layer);
 * post = ReLU if relu_out, then * (out_mask[M,N] > 0) if out_mask (ReLU backward of a relu_in layer).
 * bias / residual / mask / out_mask may be NULL.  K % 4 == 0, N <= 256, X/W/mask 16-byte aligned,
 * dense row-major. */
int nsdp_linear_f32(const float *X, const float *W, const float *bias, const float *residual,
                    const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                    int relu_in, int relu_out, void *stream);

/* Fragment-major weight packs for nsdp_linear_wp_f32 (one contiguous KiB per wave-wide MFMA operand load
 * instead of 16 rows x 64 B): Wp = pack of W[N,K] (forward operand), WpT = pack of W^T (the operand of
 * dX = dY * W); either may be NULL.  Each holds nsdp_packed_weight_floats(N,K) floats, zero padded to
 * multiples of 16 in both dimensions:
 *   Wp [((tn*ceil(K/16) + kb)*64 + 16g + li)*4 + c] = W[16tn + li][16kb + 4g + c]
 *   WpT[((tk*ceil(N/16) + nb)*64 + 16g + li)*4 + c] = W[16nb + 4g + c][16tk + li]                       */
long long nsdp_packed_weight_floats(int N, int K);
int nsdp_pack_weight_f32(const float *W, int N, int K, float *Wp, float *WpT, void *stream);

/* nsdp_linear_f32 with W given as its fragment-major pack (logical shape still [N,K]); same semantics. */
int nsdp_linear_wp_f32(const float *X, const float *Wp, const float *bias, const float *residual,
                       const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                       int relu_in, int relu_out, void *stream);

/* fp32 dense layer on the bf16 matrix pipe by error-compensated 3-way splitting (x = h + m + l in bf16, 6 of the
 * 9 partial products, fp32 accumulate; accurate to fp32 rounding level, see csrc/gemm_bf16x3.hip).  Same
 * contract as nsdp_linear_f32, with W pre-split by nsdp_pack_weight_bf16x3:
 *   Wp  [ceil(K/32)][ceil(N/16)][plane h,m,l][lane 16 g + li][8 bf16] = planes of W[16 tn + li][32 kb + kperm(g, j)]
 *   WpT [ceil(N/32)][ceil(K/16)][3][64][8]                            = planes of W[32 nb + kperm(g, j)][16 tk + li]
 * (the pack of W^T, operand of dX = dY * W); kperm(g, j) = 16 (j / 4) + 4 g + j % 4 is the order in which a lane's
 * eight activation values of a k block are loaded (the four lane groups of a row then read 64 contiguous bytes per
 * instruction); sizes from nsdp_packed_weight_bf16x3_bytes(N, K, transposed). */
/* Every pack of a model in one launch per 64 descriptors (the optimizer rewrites all weights once per train step, so
 * all packs are rebuilt once per step).  `descs` is a HOST array; kind 0 = nsdp_pack_weight_f32 layout, 1 =
 * nsdp_pack_weight_bf16x3 layout, 3 = Wp is the row-major [N, 4] zero-padded copy of a K <= 4 weight (the 16-byte weight rows of the
 * K = 4 kernels; WpT unused); Wp / WpT may be NULL (not both); results are bit-identical to the single calls. */
typedef struct {
  const float *W; /* [N, K] row-major, device */
  void *Wp;       /* forward pack or NULL */
  void *WpT;      /* pack of W^T or NULL */
  int N, K;
  int kind;
  int reserved;
} NsdpPackDesc;
int nsdp_pack_weights_batched(const NsdpPackDesc *descs, int count, void *stream);

/* The optimizer step of the train step (reference model/__init__.py:10-41 builds torch.optim.Adam,
 * model/deformation_networks.py:63-77 and flow_arbitrary.py:30-48 call optimizer.step()): the Adam update of EVERY
 * parameter tensor in ONE launch (csrc/adam.hip).  Tables in DEVICE memory (static across the replays of a captured step):
 *   descs  [n tensors]       fp32 tensors of `numel` elements each; `step` = the tensor's fp32 step counter t (torch's
 *                            capturable layout of state["step"]), read by every chunk and advanced by one by the launch
 *   chunks [n_chunks][2]     int32 (tensor, chunk of that tensor), chunks of nsdp_adam_chunk_elems() elements, every chunk
 *                            of every tensor exactly once, any order
 *   done   [n tensors]       int32 arrival counters, zero on entry, zero on exit
 * lr_dev (device fp32 scalar, e.g. a schedule changed between replays) overrides lr when not NULL.  Arithmetic: torch's
 * single-tensor fp32 Adam operation by operation (lerp_, mul_/addcmul_, sqrt / div / add, addcdiv_), bias corrections in
 * double; weight_decay is the L2 form (g + wd p), amsgrad is not offered (the reference does not use it). */
typedef struct {
  float *param;       /* updated in place */
  const float *grad;
  float *exp_avg;     /* first moment, in place */
  float *exp_avg_sq;  /* second moment, in place */
  float *step;        /* fp32 scalar, in place (+1) */
  long long numel;
} NsdpAdamDesc;
int nsdp_adam_chunk_elems(void);
int nsdp_adam_multi_f32(const NsdpAdamDesc *descs_dev, const int32_t *chunks_dev, int n_chunks, int32_t *done_dev,
                        const float *lr_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int maximize, void *stream);

long long nsdp_packed_weight_bf16x3_bytes(int N, int K, int transposed);
int nsdp_pack_weight_bf16x3(const float *W, int N, int K, void *Wp, void *WpT, void *stream);
int nsdp_linear_bf16x3_f32(const float *X, const void *Wp, const float *bias, const float *residual,
                           const float *mask, const float *out_mask, float *Y, long long M, int N, int K,
                           int relu_in, int relu_out, void *stream);

/* nsdp_linear_bf16x3_f32 plus a GATHERED difference of two small tables, added in the kernel's epilogue:
 *   Y[r] = post( X[r] W^T + b + (gq[r / g_div] - gk[(r / g_rows_per_shape) * g_nsrc + gidx[r]]) )
 * gq (.., N), gk (shapes * g_nsrc, N) fp32, gidx (M) int32 in [0, g_nsrc).  With X = the hidden layer of a position-encoding
 * MLP, gq = the queries, gk = the key table and gidx = the flattened neighbour indices this IS the logits' input
 * u = q_i - k_j + delta(rel_ij) of a vector-attention block (reference model/encoder/blocks.py:104-116, decoder/blocks.py:72-84)
 * -- the attn_pre pass (read pos, write u) and the pos tensor itself disappear.  The sum is rounded exactly like the separate
 * pass rounds it: fl(fl(X W^T + b) + fl(gq - gk)).  M and both tables' element counts < 2^31; relu_in must be 0.
 * gq == NULL: gk is the ready difference table (one query per shape: q_b - k_bj, shapes * g_nsrc rows), its rows are ADDED;
 * (half the loads).  relu_out must be 0 as well. */
int nsdp_linear_bf16x3_gather_f32(const float *X, const void *Wp, const float *bias, const float *gq, int g_div, const float *gk,
                                  const int32_t *gidx, int g_rows_per_shape, int g_nsrc, float *Y, long long M, int N, int K,
                                  int relu_in, int relu_out, void *stream);

/* nsdp_linear_bf16x3_f32 with a SIGNED residual: Y = post( pre(X) W^T + b + residual_sign * residual ), residual_sign = +1 or -1.
 * -1 turns a projection into "minus a table" in the GEMM itself: the differences q2 - q1, k2 - k1 a set abstraction's second
 * attention needs when it re-uses the first one's u as its position encoding (reference model/encoder/blocks.py:303-308). */
int nsdp_linear_bf16x3_signed_f32(const float *X, const void *Wp, const float *bias, const float *residual, float residual_sign,
                                  float *Y, long long M, int N, int K, int relu_in, int relu_out, void *stream);

/* Y = out_mask_gate( post( (X gated by mask) W^T + b + residual ) ) + addend: the masked form of nsdp_linear_bf16x3_f32 with one more
 * operand added AFTER the output mask.  It is dX of the first layer of a pre-activation residual block x + fc_1(relu(fc_0(relu(x))))
 * (reference model/decoder/blocks.py:99-142): X = d(h), mask = h, out_mask = x, addend = the gradient arriving over the skip
 * connection -- the block's input gradient in one launch instead of a GEMM and an elementwise add. */
int nsdp_linear_bf16x3_addend_f32(const float *X, const void *Wp, const float *bias, const float *residual, const float *mask,
                                  const float *out_mask, const float *addend, float *Y, long long M, int N, int K,
                                  int relu_out, void *stream);

/* Position-encoding MLP fc_delta = Linear(3, K) -> ReLU -> Linear(K, N) (reference model/encoder/blocks.py:86-90, :281-285,
 * model/decoder/blocks.py:30-34) straight from the coordinates: Y[M,N] = relu(X4 W0^T + b0) W^T + bias.  The hidden tensor [M, K]
 * is neither written nor read: the GEMM's operand producer evaluates the K = 4 layer for the 16-byte rows X4 [M,4] (zero-padded
 * coordinates) with the expression of nsdp_linear_f32's K = 4 kernel -- the values are bit for bit those of the two-launch form.
 * W0 [K,4] row-major zero-padded, b0 [K] or NULL, Wp = nsdp_pack_weight_bf16x3 of W [N,K].  gk != NULL: the gathered addend of
 * nsdp_linear_bf16x3_gather_f32 joins in the epilogue (gq == NULL: its one-table form).  Shapes: nsdp_linear_bf16x3_h0_supported. */
int nsdp_linear_bf16x3_h0_supported(long long M, int N, int K);
int nsdp_linear_bf16x3_h0_f32(const float *X4, const float *W0, const float *b0, const void *Wp, const float *bias,
                              const float *gq, int g_div, const float *gk, const int32_t *gidx, int g_rows_per_shape, int g_nsrc,
                              float *Y, long long M, int N, int K, void *stream);

/* dX GEMM of a position-encoding MLP's SECOND layer that takes the FIRST (K = 4) layer's weight gradient along.
 * fc_delta = Linear(3, d) -> ReLU -> Linear(d, d) on relative coordinates (reference model/encoder/blocks.py:86-90, :281-285,
 * model/decoder/blocks.py:30-34); the coordinates need no gradient, so Y = dY W2 -- the gradient of h0 = relu(X4 W0^T + b0) -- has
 * one reader: dW0 = (Y o [h0 > 0])^T X4, db0 = column sums of (Y o [h0 > 0]).  This entry runs the GEMM (WpT = bf16x3 pack of W2^T,
 * dY [M, K], N = width of h0) and forms both in its epilogue: Y is never written (and never read back by a weight-gradient
 * launch), the ReLU mask is recomputed from the 16-byte input rows X4 [M, 4] (zero-padded coordinates) with the forward kernel's
 * own expression (W0 [N, 4] row-major, zero-padded; b0 [N] or NULL).  dW0 is [N, k_out] (k_out = 3 or 4: the layer's real input
 * width), db0 [N] or NULL; accumulate != 0 adds to them.  Deterministic.  Shapes: nsdp_linear_bf16x3_k4tail_ok. */
int nsdp_linear_bf16x3_k4tail_ok(long long M, int N, int K);
size_t nsdp_linear_bf16x3_k4tail_workspace_bytes(long long M, int N);
int nsdp_linear_bf16x3_k4tail_f32(const float *dY, const void *WpT, const float *X4, const float *W0, const float *b0,
                                  float *dW0, float *db0, long long M, int N, int K, int k_out, int accumulate, float *ws,
                                  size_t ws_bytes, void *stream);


/* Weight/bias gradient of the layer above: dW[N,K] (+)= pre(dY)[M,N]^T * pre(X)[M,K], db[N] (+)= colsum(pre(dY));
 * pre(dY) = dY * (mask[M,N] > 0) when mask != NULL; pre(X) = relu(X) when relu_x.  db may be NULL.
 * Deterministic (two-stage
 * reduction through `workspace`, >= nsdp_linear_wgrad_workspace_bytes(M,N,K) bytes, no atomics). */
size_t nsdp_linear_wgrad_workspace_bytes(long long M, int N, int K);
int nsdp_linear_wgrad_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                          float *db, long long M, int N, int K, int accumulate, float *workspace,
                          size_t workspace_bytes, void *stream);
/* K = 4 layers with a fused output ReLU (the first layer of a position-encoding MLP: 16-byte input rows, [M, N] output):
 * the same gradients with the ReLU mask RECOMPUTED from X [M,4], W [N,4] (row-major, zero-padded K = 3) and bias [N] (or
 * NULL) -- relu'(X W^T + bias), bit for bit the decision of nsdp_linear_f32's K = 4 kernel -- instead of read back from
 * the [M, N] output: half the bytes.  Workspace as nsdp_linear_wgrad_f32 with K = 4. */
int nsdp_linear_wgrad_k4_remask_f32(const float *dY, const float *X, const float *W, const float *bias, float *dW,
                                    float *db, long long M, int N, int accumulate, float *workspace,
                                    size_t workspace_bytes, void *stream);

/* nsdp_linear_wgrad_f32 on the bf16 matrix pipe (error-compensated 3-way split of both operands, fp32 rounding-level
 * accuracy, csrc/wgrad_bf16x3.hip).  Same contract; shapes must satisfy nsdp_linear_wgrad_bf16x3_supported
 * (M >= 1024, 16 < N,K <= 256, tensors below 4 GB), workspace >= nsdp_linear_wgrad_bf16x3_workspace_bytes. */
int nsdp_linear_wgrad_bf16x3_supported(long long M, int N, int K);
size_t nsdp_linear_wgrad_bf16x3_workspace_bytes(long long M, int N, int K);
int nsdp_linear_wgrad_bf16x3_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                                 float *db, long long M, int N, int K, int accumulate, float *workspace,
                                 size_t workspace_bytes, void *stream);

/* The same weight gradient in two halves, for callers that reduce the partial sums of MANY layers in one launch (the
 * weight gradients of a train step sit on a side stream, where 98 tiny reduce launches per step wait for compute units behind
 * the critical chain's kernels: 8.6 us each alone, 33 us in the step):
 *   _partials_f32       runs the row kernel only; `workspace` then holds the per-workgroup partials and must stay untouched
 *                       until the reduce; *desc_out describes the pending reduction (workspace, targets, partial count, tiles)
 *   _reduce_batched     dW (+)= the fixed-order sum of the partials, db likewise, for `count` descriptors (host array, copied
 *                       into the kernel arguments, 48 per launch).  The sums are those of nsdp_linear_wgrad_bf16x3_f32, bit for
 *                       bit.  Two descriptors of one launch must not share a target (`accumulate` reads what is there). */
typedef struct {
  const float *ws;
  float *dW, *db;      /* db may be NULL */
  int S, nta, ktb;     /* partials, tile classes of the row kernel (filled by _partials_f32) */
  int N, K;
  int accumulate;
  int reserved;
} NsdpWgradReduceDesc;
int nsdp_linear_wgrad_bf16x3_partials_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW,
                                          float *db, long long M, int N, int K, int accumulate, float *workspace,
                                          size_t workspace_bytes, NsdpWgradReduceDesc *desc_out, void *stream);
int nsdp_wgrad_bf16x3_reduce_batched(const NsdpWgradReduceDesc *descs, int count, void *stream);

/* bf16-storage variants of the attention glue (csrc/attention.hip, same kernels instantiated for a bf16 storage type):
 * q, kf, vf, pos, u, a, y, residual, a_g, v_g and the activation gradients du, dy, da, dpos, dpos_acc are bf16 tensors;
 * lse and the scatter / reduction outputs dq, dkf, dvf, da_g, dv_g are fp32.  Arithmetic is fp32 throughout. */
int nsdp_attn_pre_fwd_bf16(const void *q, const void *kf, const void *pos, const int32_t *idx, int B, int n, int N, int k,
                           int d, int q_per_shape, void *u, void *stream);
int nsdp_attn_pre_bwd_bf16(const void *du, const int32_t *idx, int B, int n, int N, int k, int d, int q_per_shape,
                           float *dq, float *dkf, void *dpos_acc, void *stream);
int nsdp_attn_post_fwd_bf16(const void *a, const void *vf, const void *pos, const int32_t *idx, const void *a_g,
                            const void *v_g, const void *residual, int B, int n, int N, int k, int d, void *y, float *lse,
                            void *stream);
int nsdp_attn_post_bwd_bf16(const void *dy, const void *a, const void *vf, const void *pos, const int32_t *idx,
                            const void *a_g, const void *v_g, const void *y, const void *residual, const float *lse, int B,
                            int n, int N, int k, int d, void *da, void *dpos, float *dvf, float *da_g, float *dv_g,
                            void *stream);

/* ----------------------------------------------------------------------------------------------
 * bf16-STORAGE dense layers (BASELINE config 3: flow_arbitrary.py:30-48 step with activations and saved tensors in
 * bf16, fp32 accumulation, fp32 master weights / weight gradients).  Same layer contract as nsdp_linear_f32 /
 * nsdp_linear_wgrad_f32 with X, residual, mask, out_mask, dY (and Y unless out_f32) as bf16 tensors; one bf16 MFMA
 * product per multiply-add.  csrc/gemm_bf16.hip.
 *   Wp  [ceil(K/32)][ceil(N/16)][lane 16 g + i][8 bf16] = W[chan(nt, i)][32 kb + 8 g + j]
 *   WpT [ceil(N/32)][ceil(K/16)][lane][8]               = W[32 nb + 8 g + j][chan(tk, i)]
 *   chan(t, i) = 32 (t / 2) + 8 (i / 4) + 4 (t % 2) + i % 4 for paired tiles (16 t + i for an unpaired last tile): the
 *   eight accumulator values a lane holds for one row are eight consecutive channels -> 16-byte bf16 stores.
 * nsdp_pack_weights_bf16 takes the NsdpPackDesc array of nsdp_pack_weights_batched (`kind` ignored).
 * Shapes: K % 8 == 0, 8 <= K <= 256, N <= 256, N % 4 == 0 (any N with out_f32); wgrad: N, K even, <= 256. */
long long nsdp_packed_weight_bf16_bytes(int N, int K, int transposed);
int nsdp_pack_weights_bf16(const NsdpPackDesc *descs, int count, void *stream);
int nsdp_linear_bf16(const void *X, const void *Wp, const float *bias, const void *residual, const void *mask,
                     const void *out_mask, void *Y, long long M, int N, int K, int relu_in, int relu_out, int out_f32,
                     void *stream);
size_t nsdp_linear_wgrad_bf16_workspace_bytes(long long M, int N, int K);
/* 1 when nsdp_linear_wgrad_bf16 can take `mask` for this shape (its LDS ring holds the mask rows as well); 0: the caller
 * multiplies the mask into dY first (the call then returns NSDP_ENOSUP with a mask). */
int nsdp_linear_wgrad_bf16_takes_mask(long long M, int N, int K);
int nsdp_linear_wgrad_bf16(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db,
                           long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                           void *stream);

/* Weight gradient of the SECOND layer of the same MLP from the coordinates: dW [N,K] = dY^T relu(X4 W0^T + b0), db [N] = column
 * sums of dY -- nsdp_linear_wgrad_bf16x3_f32 whose X operand is recomputed by the producer (see nsdp_linear_bf16x3_h0_f32) instead
 * of read from an [M, K] tensor.  Workspace: nsdp_linear_wgrad_bf16x3_workspace_bytes(M, N, K).  desc_out != NULL: partial sums
 * only, reduction described for nsdp_wgrad_bf16x3_reduce_batched; NULL: reduced by this call.  Deterministic. */
int nsdp_linear_wgrad_bf16x3_h0_supported(long long M, int N, int K);
int nsdp_linear_wgrad_bf16x3_h0_f32(const float *dY, const float *X4, const float *W0, const float *b0, float *dW, float *db,
                                    long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                    NsdpWgradReduceDesc *desc_out, void *stream);

/* G16 layout: a tensor T[M, C] (M % 16 == 0, C % 4 == 0) stored as [M / 16][C / 4][16 rows][4 floats] -- groups of 16 rows,
 * channel-quad-major inside a group; the same bytes, permuted.  It is the layout of the wide intermediates that ONLY the dense-layer
 * kernels touch: the hidden layer of every Linear -> ReLU -> Linear pair over per-(centre, neighbour) rows (fc_gamma of the attention
 * blocks, reference model/encoder/blocks.py:86-124, model/decoder/blocks.py:30-91) and its gradient -- written by one GEMM, read by
 * the next, by one weight gradient and as a ReLU mask.  In it every wave-wide activation load and every 16 x 16 output tile's store of
 * the bf16x3 kernels is ONE contiguous KiB (row-major: 16 runs of 64 B at the row pitch), and outputs need no staging through LDS.
 *   nsdp_layout_g16_f32            dst = src re-laid out (to_g16 != 0: row-major -> G16, else back); out of place.
 *   nsdp_linear_bf16x3_g16_f32     nsdp_linear_bf16x3_f32 (+ the addend of nsdp_linear_bf16x3_addend_f32) with `layout` = 1: X and
 *                                  mask in G16 (no input ReLU), or 2: Y in G16 (no mask, residual, out_mask, addend).  Everything else
 *                                  stays row-major.  Same arithmetic element for element: bit-identical to the row-major call.
 *   nsdp_linear_wgrad_bf16x3_g16_f32   nsdp_linear_wgrad_bf16x3_f32 / _partials_f32 (desc_out != NULL: partial sums only) with
 *                                  `layout` = 1: dY and mask in G16, or 2: X in G16 (no mask).  dW, db bit-identical to the row-major call.
 * The *_supported predicates say which (shape, layout, operands) combinations are instantiated. */
int nsdp_layout_g16_f32(const float *src, float *dst, long long M, int C, int to_g16, void *stream);
int nsdp_linear_bf16x3_g16_supported(long long M, int N, int K, int layout, int has_mask, int relu_in);
int nsdp_linear_bf16x3_g16_f32(const float *X, const void *Wp, const float *bias, const float *residual, const float *mask,
                               const float *out_mask, const float *addend, float *Y, long long M, int N, int K, int relu_in,
                               int relu_out, int layout, const unsigned char *mask_bits, unsigned char *bits_out, void *stream);
/* ReLU bits: the mask [h > 0] of a hidden tensor h [M, C] that lives in the G16 layout, one BIT per element:
 * [M / 16][ceil(C / 32)][64] bytes, byte (row group, k block kb, 4 * (row % 16) + g) = bits 0-3: channels 32 kb + 4 g .. + 3,
 * bits 4-7: channels 32 kb + 16 + 4 g .. + 3 (the eight values lane (row, g) of the GEMM's fragment convention holds in block kb).
 * nsdp_linear_bf16x3_g16_f32 writes them next to a G16 output with an output ReLU (bits_out != NULL) and takes them in place of
 * `mask` for a G16 input (mask_bits != NULL: the dX GEMM of that layer); nsdp_linear_wgrad_bf16x3_g16_f32 takes them as the mask of
 * a G16 dY (mask_bits != NULL, layout 1).  28 bytes per row of a 200-wide layer where the fp32 mask stream was 800. */
size_t nsdp_relu_bits_bytes(long long M, int C);
int nsdp_linear_wgrad_bf16x3_g16_supported(long long M, int N, int K, int layout, int has_mask);
int nsdp_linear_wgrad_bf16x3_g16_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW, float *db,
                                     long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                     NsdpWgradReduceDesc *desc_out, int layout, const unsigned char *mask_bits, void *stream);

/* nsdp_linear_wgrad_bf16 in two halves, like nsdp_linear_wgrad_bf16x3_partials_f32 / nsdp_wgrad_bf16x3_reduce_batched: the row
 * kernel now (an all-zero *desc_out -- ws == NULL -- means the call had nothing to do), the fixed-order sums of many layers'
 * partials in one launch later (bit-identical to the one-call form; two descriptors of a launch must not share a target). */
typedef struct {
  const float *ws;     /* [S][N * K + N] partials */
  float *dW, *db;      /* db may be NULL */
  int S, N, K;
  int accumulate;
  int reserved;
} NsdpWgradB16ReduceDesc;
int nsdp_linear_wgrad_bf16_partials(const void *dY, const void *X, const void *mask, int relu_x, float *dW, float *db,
                                    long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                    NsdpWgradB16ReduceDesc *desc_out, void *stream);
int nsdp_wgrad_bf16_reduce_batched(const NsdpWgradB16ReduceDesc *descs, int count, void *stream);
/* nsdp_linear_wgrad_f32 (the exact-fp32 kernels of the small layers) likewise: its partials have the same [S][N K + N] layout,
 * its reduce the same eight chains -- the descriptor goes to nsdp_wgrad_bf16_reduce_batched. */
int nsdp_linear_wgrad_partials_f32(const float *dY, const float *X, const float *mask, int relu_x, float *dW, float *db,
                                   long long M, int N, int K, int accumulate, float *workspace, size_t workspace_bytes,
                                   NsdpWgradB16ReduceDesc *desc_out, void *stream);

/* bf16-storage variants of the BatchNorm kernels (x, addend, y, dy, dx bf16; statistics and affine parameters fp32). */
int nsdp_bn_stats_bf16(const void *x, const void *addend, long long R, int C, float eps, float momentum,
                       float *running_mean, float *running_var, float *mean, float *invstd, float *workspace,
                       long long *num_batches_tracked, void *stream);
int nsdp_bn_train_fwd_bf16(const void *x, const void *addend, long long R, int C, float eps, float momentum, int updates,
                           float *running_mean, float *running_var, long long *num_batches_tracked, const float *gamma,
                           const float *beta, int relu, void *y, float *mean, float *invstd, float *workspace,
                           void *stream);
int nsdp_bn_apply_bf16(const void *x, const void *addend, const float *mean, const float *invstd, const float *gamma,
                       const float *beta, long long R, int C, int relu, void *y, void *stream);
int nsdp_bn_backward_bf16(const void *dy, const void *y_relu, const void *x, const void *addend, const float *mean,
                          const float *invstd, const float *gamma, long long R, int C, int training, void *dx,
                          float *dgamma, float *dbeta, float *workspace, void *stream);

/* Atomics-free scatter through inverse neighbour lists (csrc/segment.hip).  nsdp_knn_invert: idx (B, E) int32 with values
 * in [0, N) (E = centres x neighbours) -> offsets (B, N + 1), entries (B, E): entries[b][offsets[b][s] .. offsets[b][s+1])
 * is the ascending list of the flat positions e with idx[b][e] == s.  N <= 32768.  Built once per index set and step.
 * nsdp_segment_sum_rows: out[b][s][:] = scale * sum over that list of src[b][e][:]  (src (B, E, d) fp32 / bf16, out fp32) --
 * the scatter-add of the attention backward (dvf, dkf) as a deterministic gather-reduce. */
int nsdp_knn_invert(const int32_t *idx, int B, int E, int N, int32_t *offsets, int32_t *entries, void *stream);
int nsdp_segment_sum_rows(const float *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                          float scale, float *out, void *stream);
int nsdp_segment_sum_rows_bf16(const void *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                               float scale, float *out, void *stream);
/* ... with the caller's next statement folded in: out = scale * sum + addend (addend (B, N, d) fp32; bit-identical to the call above
 * followed by the addition for scale = +-1). */
int nsdp_segment_sum_rows_add(const float *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                              float scale, const float *addend, float *out, void *stream);
int nsdp_segment_sum_rows_add_bf16(const void *src, const int32_t *offsets, const int32_t *entries, int B, int E, int N, int d,
                                   float scale, const float *addend, float *out, void *stream);

/* Scatter as a GEMM (bf16 storage): table[b][a][c] = sum over the rows r of shape b with idx[b][r] == a of src[b][r][c],
 * computed as one-hot(idx)^T x src on the matrix cores (exact: 1.0 x bf16, fp32 accumulation; no atomics, deterministic).
 * src (B, rows, d) bf16, idx (B, rows) int32 in [0, N), table (B, N, d) fp32 (overwritten).  N even, <= 128; d % 8 == 0.
 * Used for the decoder's 100-anchor tables (dvf, dkf); nsdp_attn_post_bwd_bf16 accepts dvf == NULL with vf != NULL for
 * that (it then only streams). */
size_t nsdp_scatter_rows_onehot_bf16_workspace_bytes(int B, long long rows, int N, int d);
int nsdp_scatter_rows_onehot_bf16(const void *src, const int32_t *idx, int B, long long rows, int N, int d, float *table,
                                  float *workspace, size_t workspace_bytes, void *stream);
/* The same for fp32 storage: src (B, rows, d) fp32 is split into its three bf16 planes on the fly (csrc/wgrad_bf16x3.hip,
 * one-hot mode) and the one-hot operand is exact, so table[b][a][c] is the fp32 sum of the selected rows in a FIXED order --
 * the deterministic replacement of the fp32-atomic scatters (register-table / LDS-table kernels) behind the decoder's
 * anchor-table gradients (reference model/decoder/blocks.py:72-77: the backward of index_points over 100 anchors).
 * N <= 128, 16 < d <= 208, d % 4 == 0; table (B, N, d) is overwritten. */
size_t nsdp_scatter_rows_onehot_f32_workspace_bytes(int B, long long rows, int N, int d);
int nsdp_scatter_rows_onehot_f32(const float *src, const int32_t *idx, int B, long long rows, int N, int d, float *table,
                                 float *workspace, size_t workspace_bytes, void *stream);

/* K = 4 layers with bf16 storage (first layer of every position-encoding MLP: fp32 relative coordinates zero-padded
 * to 4 columns in, bf16 out; csrc/k4_bf16.hip).  W is the plain [N,4] fp32 matrix.  N % 8 == 0.
 * dX of such a layer is an nsdp_linear_bf16 call with 4 outputs. */
int nsdp_linear_k4_bf16(const float *X, const float *W, const float *bias, void *Y, long long M, int N, int relu_out,
                        void *stream);
size_t nsdp_linear_wgrad_k4_bf16_workspace_bytes(long long M, int N);
int nsdp_linear_wgrad_k4_bf16(const void *dY, const float *X, const void *mask, float *dW, float *db, long long M, int N,
                              float *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Point-Transformer vector attention glue (everything between the dense layers of one block), replacing
 * the materialised ATen gather / sub / add / softmax / einsum sequence of model/encoder/blocks.py:104-124,
 * :290-308 and model/decoder/blocks.py:72-91.  Channels-last fp32, d <= 256:
 *   q (B,n,d) per-centre queries;  kf, vf (B,N,d) projected source features;  idx (B,n,k) i32 neighbours
 *   pos, u, a (B,n,k,d) per-(centre,neighbour) tensors;  a_g, v_g (B,d) optional global token (decoder).
 * -------------------------------------------------------------------------------------------- */

/* u = q[:, :, None] - kf[idx] + pos;  q_per_shape = 1: q is (B,1,d), one query vector per shape shared
 * by all centres (decoder: q = w_qs(z), model/decoder/blocks.py:63-66) */
int nsdp_attn_pre_fwd(const float *q, const float *kf, const float *pos, const int32_t *idx, int B, int n,
                      int N, int k, int d, int q_per_shape, float *u, void *stream);
/* dq = sum_j du (summed over the centres too when q_per_shape);  dkf (zero-filled here) -= scatter(du);
 * dpos_acc (B,n,k,d), may be NULL: dpos_acc += du in the same pass -- `pos` feeds both the logits (through u) and
 * the values, so its gradient is d(u) + the dpos of nsdp_attn_post_bwd; accumulating here replaces a separate
 * elementwise add over the largest tensor of the block. */
int nsdp_attn_pre_bwd(const float *du, const int32_t *idx, int B, int n, int N, int k, int d,
                      int q_per_shape, float *dq, float *dkf, float *dpos_acc, void *stream);
/* The dq-only form (the caller scatters dkf itself, through inverse lists) with the caller's next statement folded in:
 * dq (B,n,d) = sum_j du - dq_sub (dq_sub (B,n,d): the upstream gradient of the block's output, whose share of d(pos) the fused
 * d(pos) hand-over counted twice -- hip_attention._AttnPre.backward).  Bit-identical to nsdp_attn_pre_bwd followed by the
 * subtraction. */
int nsdp_attn_pre_bwd_sub(const float *du, const int32_t *idx, int B, int n, int N, int k, int d, const float *dq_sub, float *dq,
                          void *stream);
int nsdp_attn_pre_bwd_sub_bf16(const void *du, const int32_t *idx, int B, int n, int N, int k, int d, const void *dq_sub, float *dq,
                               void *stream);
/* y = sum_j softmax_j(a) * (vf[idx] + pos) [+ softmax weight of a_g * v_g] [+ residual];
 * lse (B,n,d) = log-sum-exp of the logits (kept for the backward pass).
 * vf == NULL: values are `pos` alone (pos_only block); a_g/v_g, residual may be NULL. */
int nsdp_attn_post_fwd(const float *a, const float *vf, const float *pos, const int32_t *idx,
                       const float *a_g, const float *v_g, const float *residual, int B, int n, int N,
                       int k, int d, float *y, float *lse, void *stream);
/* da = w dy (s - y_att), dpos = w dy, dvf (zero-filled here) += scatter(dpos), da_g/dv_g (zero-filled
 * here) = global-token gradients; y/residual/lse as produced by / passed to nsdp_attn_post_fwd. */
int nsdp_attn_post_bwd(const float *dy, const float *a, const float *vf, const float *pos,
                       const int32_t *idx, const float *a_g, const float *v_g, const float *y,
                       const float *residual, const float *lse, int B, int n, int N, int k, int d,
                       float *da, float *dpos, float *dvf, float *da_g, float *dv_g, void *stream);

/* nsdp_attn_post_fwd / _bwd for a block whose `pos` was never materialised: `u` = q_i - k_j + pos (the output of
 * nsdp_linear_bf16x3_gather_f32), `vk` = the table v + k, qsub (B, n, d) = the queries: values = u + vk[idx] - qsub_i.
 * Gradients are those of the original graph: dpos (= w dy) and dvf (= its scatter, or NULL: the caller scatters). */
int nsdp_attn_post_fwd_q(const float *a, const float *vk, const float *u, const int32_t *idx, const float *qsub,
                         const float *residual, int B, int n, int N, int k, int d, float *y, float *lse, void *stream);
int nsdp_attn_post_bwd_q(const float *dy, const float *a, const float *vk, const float *u, const int32_t *idx, const float *qsub,
                         const float *y, const float *residual, const float *lse, int B, int n, int N, int k, int d,
                         float *da, float *dpos, float *dvf, void *stream);

/* nsdp_attn_post_bwd for a block whose value scatter (dvf) is done by the caller (nsdp_scatter_rows_onehot_*): da, dpos and
 * the global-token gradients with NO atomics -- a workgroup owns centres of one shape, partial sums are combined in a
 * fixed order through `workspace` (>= nsdp_attn_post_bwd_det_workspace_bytes).  Bit-reproducible run to run.  vf required;
 * a_g / v_g / da_g / dv_g all present or all NULL. */
size_t nsdp_attn_post_bwd_det_workspace_bytes(int B, int n, int k, int d);
int nsdp_attn_post_bwd_det(const float *dy, const float *a, const float *vf, const float *pos, const int32_t *idx,
                           const float *a_g, const float *v_g, const float *y, const float *residual, const float *lse,
                           int B, int n, int N, int k, int d, float *da, float *dpos, float *da_g, float *dv_g,
                           float *workspace, size_t workspace_bytes, void *stream);
int nsdp_attn_post_bwd_det_bf16(const void *dy, const void *a, const void *vf, const void *pos, const int32_t *idx,
                                const void *a_g, const void *v_g, const void *y, const void *residual, const float *lse,
                                int B, int n, int N, int k, int d, void *da, void *dpos, float *da_g, float *dv_g,
                                float *workspace, size_t workspace_bytes, void *stream);


/* ----------------------------------------------------------------------------------------------
 * Fused cross-attention decoder forward (no-grad / inference path): CrossTransformerDecoder.forward,
 * model/decoder/crosstransformer_decoder.py:45-70 + model/decoder/blocks.py:48-95, for dim = 200,
 * hidden_dim = 128, n_blocks = 5, out_dim = 3.  One wave carries 16 query points through all 18 dense
 * layers in registers (transposed fp32 MFMA chain).  Inputs, all zero-padded to 208 / 128 channels:
 *   xyz_q (B,NQ,3), anchors (B,A,3), idx (B,NQ,KN) i32 nearest anchors,
 *   qk (B,A,208) = w_qs(z) - w_ks(anchor_feats), vtab (B,A,208) = w_vs(anchor_feats),
 *   a_g (B,208) = fc_gamma(w_qs(z) - w_k_global(z)), v_g (B,208) = w_v_global(z),
 *   weights[17], every W in FRAGMENT-MAJOR order [out tile of 16][k block of 16][lane = 16 g + li][4], holding
 *   W_rowmajor[16 tile + li][16 kblock + 4 g + 0..3] (one wave-wide MFMA operand load = one contiguous KiB):
 *                 { fc_delta.0 [208,4] (weight | bias, row-major), fc_delta.2 W [208,208], b [208], fc_gamma.0 W, b,
 *                   fc_gamma.2 W, b, init_enc W [128,208], b [128], fc_c W [5,128,208], b [5,128],
 *                   blocks.fc_0 W [5,128,128], b [5,128], blocks.fc_1 W [5,128,128], b [5,128],
 *                   fc_out W [16,128], b [16] }.
 * -------------------------------------------------------------------------------------------- */
int nsdp_decoder_fused_fwd(const float *xyz_q, const float *anchors, const int32_t *idx, const float *qk,
                           const float *vtab, const float *a_g, const float *v_g,
                           const float *const *weights, int n_weights, int B, int NQ, int A, int KN, int D,
                           int H, float *out, void *stream);

/* ----------------------------------------------------------------------------------------------
 * BatchNorm1d on channels-last rows x[R,C] (R = B*n, C % 4 == 0, C <= 1024), replacing the 35
 * nn.BatchNorm1d calls of the encoder (model/encoder/blocks.py:132, :158, :300-312) together with the
 * residual add in front of them (`addend`, may be NULL: the norm acts on x + addend) and the ReLU behind
 * them (`relu`).  `workspace` >= nsdp_bn_workspace_bytes(C) bytes.
 * -------------------------------------------------------------------------------------------- */
size_t nsdp_bn_workspace_bytes(int C);
/* training statistics: mean[C], invstd[C] = 1/sqrt(biased var + eps); running_mean/var (may be NULL)
 * updated with `momentum` and the unbiased variance, exactly like nn.BatchNorm1d in training mode;
 * *num_batches_tracked (device int64, may be NULL) += 1 (no kernel of its own for the counter).
 * Statistics are accumulated as shifted sums (pivot = row 0), so |mean| >> std does not cancel. */
int nsdp_bn_stats(const float *x, const float *addend, long long R, int C, float eps, float momentum,
                  float *running_mean, float *running_var, float *mean, float *invstd, float *workspace,
                  long long *num_batches_tracked, void *stream);
/* The training-mode forward as ONE call: statistics, running-average update, normalisation (y, and mean / invstd for the
 * backward).  For R <= 16384 rows it is one launch (a workgroup owns a slab of channels and all rows, in registers);
 * larger tensors take nsdp_bn_stats + nsdp_bn_apply.  `updates` >= 1: the momentum update of the running statistics is
 * applied that many times from the same batch statistics and *num_batches_tracked += updates -- what `updates` forward
 * passes of the module over the SAME input leave behind (FlowArbitrary encodes one cloud twice, reference
 * model/flow_arbitrary.py:19-20; this library encodes it once). */
int nsdp_bn_train_fwd(const float *x, const float *addend, long long R, int C, float eps, float momentum, int updates,
                      float *running_mean, float *running_var, long long *num_batches_tracked, const float *gamma,
                      const float *beta, int relu, float *y, float *mean, float *invstd, float *workspace, void *stream);
/* y = ((x + addend) - mean) * invstd * gamma + beta, then ReLU if relu */
int nsdp_bn_apply(const float *x, const float *addend, const float *mean, const float *invstd,
                  const float *gamma, const float *beta, long long R, int C, int relu, float *y, void *stream);
/* backward of the above: dy' = dy * (y_relu > 0) when y_relu != NULL (the forward output of a relu norm);
 * dgamma = sum dy' xhat, dbeta = sum dy'; dx (= gradient of both x and addend) with the batch-statistics
 * terms when training != 0, plain gamma*invstd*dy' otherwise (eval: statistics are constants). */
int nsdp_bn_backward(const float *dy, const float *y_relu, const float *x, const float *addend,
                     const float *mean, const float *invstd, const float *gamma, long long R, int C,
                     int training, float *dx, float *dgamma, float *dbeta, float *workspace, void *stream);

/* ----------------------------------------------------------------------------------------------
 * Kernel timing with HIP events on the launch stream (used by bench.py for the roofline object)
 * -------------------------------------------------------------------------------------------- */
void nsdp_prof_enable(int on);            /* on=1 clears previous records and starts recording */
/* same, for a subset: bit k of `mask` = time the kernels of kind k (nsdp_prof_name(k)); 0 stops.  An event pair per
 * launch keeps consecutive kernels from overlapping head-to-tail: timing all ~500 launches of a train step costs
 * ~3 ms of a 52 ms step, timing one kernel class a few hundred microseconds. */
void nsdp_prof_enable_kinds(unsigned mask);
int nsdp_prof_num_kinds(void);
/* Kernel-variant trace (test instrumentation): while enabled every launcher records the template instance it chose
 * ("linear_bf16x3<2,13,0,8,0>", "attn_post_bwd_lds", ...).  nsdp_trace_enable(1) clears and starts, (0) stops;
 * nsdp_trace_read copies the newline-separated unique names (NUL-terminated, truncated to `capacity`) and returns the
 * size needed. */
void nsdp_trace_enable(int on);
int nsdp_trace_read(char *buf, int capacity);
const char *nsdp_prof_name(int kind);
/* Sums over all recorded launches of `kind`: count, elapsed ms, algorithmic flops and bytes. */
int nsdp_prof_collect(int kind, long long *launches, double *total_ms, double *flops, double *bytes);

/* ----------------------------------------------------------------------------------------------
 * Multi-stream replay of a stream-captured step (csrc/graph_exec.hip).  No counterpart in the reference (its step is
 * enqueued op by op from Python, train.py:150-225); this is the host side of `train_on_batch` taken off the critical
 * path: the step is captured once into a hipGraph_t (our kernels, ATen's, memsets, copies) and replayed from C with the
 * eager schedule's stream concurrency -- hipGraphLaunch of the same graph serialises its branches on this ROCm.
 *   create : `graph` = the captured hipGraph_t (stays owned by the caller and must outlive the executor); nodes are
 *            assigned to at most `max_streams` HIP streams (1 = the caller's stream only).
 *   launch : enqueue one replay behind everything already on `stream`; `stream` continues behind all branches.
 * -------------------------------------------------------------------------------------------- */
int nsdp_graph_exec_create(void *graph, int max_streams, void **out_handle);
int nsdp_graph_exec_info(void *handle, int *nodes, int *kernels, int *streams, int *cross_edges, int *own_graph_nodes);
int nsdp_graph_exec_launch(void *handle, void *stream);
/* One replay with a HIP event pair around every kernel node whose (mangled) name contains `name_substr`; waits for the
 * replay to finish and returns the number of such launches and the sum of their durations AS REPLAYED -- i.e. including the
 * time a kernel shares the chip with the other streams' kernels (measurement, not for timed regions). */
int nsdp_graph_exec_launch_timed(void *handle, void *stream, const char *name_substr, long long *launches, double *total_ms);
int nsdp_graph_exec_destroy(void *handle);

#ifdef __cplusplus
}
#endif
#endif /* NSDP_HIP_H_ */
