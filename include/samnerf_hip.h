/* samnerf_hip.h -- C ABI of libsamnerf_hip.so: the MI355X (gfx950) kernels of the SAM-NeRF
 * render-and-distill hot path.
 *
 * This is the drop-in boundary.  The reference reaches its native code through the tiny-cuda-nn
 * torch bindings (tcnn.Encoding / tcnn.Network / tcnn.NetworkWithInputEncoding modules) and through
 * ordinary torch ops; citations below are relative to the reference repository root and name the
 * interface each entry point replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to fp32 / int32 / uint8 data allocated by the caller
 *     (PyTorch); the library never allocates, frees or retains a pointer past the call;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); kernels are
 *     enqueued on it and the call returns without synchronising; all entry points are re-entrant;
 *   - return value: 0 on success, negative on error (bad argument / launch failure); the message
 *     is available from snf_last_error() on the calling thread.  No C++ exception crosses the ABI;
 *   - tensors are row-major and contiguous unless a leading dimension (`ld*`, in elements) is given.
 */
#ifndef SAMNERF_HIP_H
#define SAMNERF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNF_OK 0
#define SNF_ERR_ARG (-1)
#define SNF_ERR_LAUNCH (-2)

#define SNF_CONTRACT_NONE 0
#define SNF_CONTRACT_LINF 1 /* SceneContraction(order=inf): nerfacto field + proposal nets (nerfstudio/models/nerfacto.py:156) */
#define SNF_CONTRACT_L2 2   /* SceneContraction(): SAMField default (samnerf/sam_field.py:32) */

#define SNF_ACT_NONE 0
#define SNF_ACT_RELU 1
#define SNF_ACT_SIGMOID 2
#define SNF_ACT_GELU 3 /* forward only (nn.GELU, erf form): the image encoder's MLP blocks */

typedef void* snf_stream_t;

int snf_version(void);
const char* snf_last_error(void);

/* ---- task streams on part of the chip.  The reference runs its nerfacto branch and its two feature heads back to back on
 *      torch's current stream (samnerf/sam_model.py:226-301); this library's step schedule runs them as concurrent tasks and may
 *      confine a task -- or single launches of it -- to a share of the 256 CUs, so that the bandwidth-bound table kernels (which fill
 *      every CU they touch) leave CUs to the matrix kernels of the other tasks.
 * snf_stream_create_cu_mask: a HIP stream whose kernels run on `n_cus` CUs only (hipExtStreamCreateWithCUMask with the first n_cus
 * mask bits set -- the driver spreads mask bits evenly over the 8 XCDs; clamped to [8, CU count]).  The handle is a hipStream_t:
 * wrap it with torch.cuda.ExternalStream(handle), destroy it with snf_stream_destroy once nothing is enqueued on it. */
int snf_stream_create_cu_mask(int n_cus, snf_stream_t* out_stream);
/* snf_stream_create_priority: a HIP stream of the given priority (0 default, < 0 higher, > 0 lower; clamped to the device's range). */
int snf_stream_create_priority(int priority, snf_stream_t* out_stream);
int snf_stream_destroy(snf_stream_t stream);

/* ---- a3: UniformLinDispPiecewiseSampler / SpacedSampler.generate_ray_samples
 *      (nerfstudio/model_components/ray_samplers.py:79-126,223-246).
 * nears,fars [R]; t_rand [R] single per-ray jitter or NULL (eval).  Out: sbins, ebins [R,P+1]. */
int snf_sample_spacing(const float* nears, const float* fars, const float* t_rand, int R, int P,
                       float* sbins, float* ebins, snf_stream_t stream);

/* ---- a1+a4: Frustums.get_positions (nerfstudio/cameras/rays.py:48-57) fused with
 *      SceneContraction.forward (field_components/spatial_distortions.py:66-69), the (x+2)/4
 *      normalisation and the (0,1) selector (fields/nerfacto_field.py:244-252,
 *      fields/density_fields.py:103-110, samnerf/sam_field.py:116-118).
 * origins,dirs [R,3]; ebins [R,n+1]; ids [R,K] int32 sample indices or NULL (then K must equal n and
 * every sample is taken in order: the top-K gather of samnerf/sam_model.py:250-255).
 * Out: u [R*K,3] normalised positions, selector [R*K] uint8 (NULL when use_selector == 0). */
int snf_positions(const float* origins, const float* dirs, const float* ebins, const int32_t* ids,
                  int R, int n, int K, int contraction, int use_selector, float* u, uint8_t* selector,
                  snf_stream_t stream);
/* Row-mapped form for the eval render (samnerf/sam_model.py:371-377,392-398: the feature passes render
 * `camera_ray_bundle[hind.flatten(), wind.flatten()]`, an index subset of the rays pass 1 has just sampled):
 * list entry j reads ray src_rows[j] of the chunk (origins / dirs / ebins) and owns row dst_rows[j] of ids [.,K]
 * and u [.*K,3].  Same arithmetic as snf_positions; no selector. */
int snf_positions_rows(const float* origins, const float* dirs, const float* ebins, const int32_t* ids,
                       const int32_t* src_rows, const int32_t* dst_rows, int M, int n, int K, int contraction,
                       float* u, snf_stream_t stream);

/* ---- a6: tcnn.Encoding(HashGrid) forward with the reference's torch semantics
 *      (HashEncoding.pytorch_fwd, nerfstudio/field_components/encodings.py:289-349; call sites
 *      samnerf/sam_field.py:99-109, fields/nerfacto_field.py:157-167, fields/density_fields.py:73-99).
 * u [N,3] in [0,1]; table [L*2^log2_T, F] (level-major rows, feature-minor); scalings [L] (the
 * reference's floor(min_res*g^l) vector, supplied by the caller).  F in {2,8}.
 * Out: out[n*ld_out + col_off + l*F + f]; ld_out = 0 (col_off = 0) selects the level-major layout out[(l*N + n)*F + f],
 * which snf_mlp64_fwd (ldx = 0), snf_linear_bwd_weight (ldx = 0) and, for gradients, snf_mlp64_bwd_data (lddx = 0) and
 * snf_hashgrid_bwd_presorted[_adam] (ld_out = 0: no staging pass) read and write directly. */
int snf_hashgrid_fwd(const float* u, const float* table, const float* scalings, int N, int L, int F,
                     int log2_T, float* out, int ld_out, int col_off, snf_stream_t stream);

/* backward of the above w.r.t. the table only (positions are detached on this path: sam_field.py:116,
 * ray_samplers.py:357): grad_table[row*F+f] += w_corner * grad_out[...]  (atomic accumulation;
 * caller zero-fills grad_table). */
int snf_hashgrid_bwd(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                     int log2_T, int ld_out, int col_off, float* grad_table, snf_stream_t stream);

/* Atomic-free variant of snf_hashgrid_bwd (the default in the product path): contributions are counting-sorted by
 * destination bucket and reduced per bucket in LDS; same result up to fp32 summation order.  `workspace` is a
 * caller-allocated scratch buffer of at least snf_hashgrid_bwd_workspace_bytes(N, L, log2_T) bytes (16-B aligned),
 * free for reuse as soon as the call's kernels have run on `stream`. */
int64_t snf_hashgrid_bwd_workspace_bytes(int N, int L, int log2_T);
int snf_hashgrid_bwd_sorted(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                            int log2_T, int ld_out, int col_off, float* grad_table, void* workspace,
                            int64_t workspace_bytes, snf_stream_t stream);

/* The same with run aggregation in the reduce pass for the first n_run_levels levels: at a coarse level neighbouring
 * records of a bucket carry the same row (consecutive samples of a ray inside one cell); they are summed across adjacent
 * lanes before ranking.  Any n_run_levels in [0, L] gives the same sums up to fp32 summation order. */
int snf_hashgrid_bwd_sorted_ex(const float* u, const float* grad_out, const float* scalings, int N, int L, int F,
                               int log2_T, int ld_out, int col_off, int n_run_levels, float* grad_table, void* workspace,
                               int64_t workspace_bytes, snf_stream_t stream);

/* The sorted backward in two halves.  snf_hashgrid_sort (count / scan / scatter) depends on the positions and the level
 * geometry only, so grids of equal geometry at the same points share it and it can run in the forward pass;
 * snf_hashgrid_bwd_presorted (stage + reduce) then needs the sorted workspace (read-only: may be shared by concurrent
 * streams) and a stage buffer of L*N*F floats.  workspace size: snf_hashgrid_bwd_workspace_bytes(N, L, log2_T). */
int snf_hashgrid_sort(const float* u, const float* scalings, int N, int L, int log2_T, void* workspace,
                      int64_t workspace_bytes, snf_stream_t stream);
int snf_hashgrid_bwd_presorted(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                               int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                               snf_stream_t stream);

/* The same two halves with x-pair records (F = 2 grids): the two x-neighbour corners of a sample share a bucket for all but 2^-11 of the
 * pairs and take the same staged gradient, so snf_hashgrid_sort_xp writes ONE 16-byte record per pair (two single records for a split
 * pair) and snf_hashgrid_bwd_presorted_adam_xp reduces it with one gradient gather per pair -- same sums, bit for bit, as
 * snf_hashgrid_sort + snf_hashgrid_bwd_presorted(_adam) (64-bit fixed-point accumulation is order-independent).  The two sorts leave
 * different workspaces: an _xp sort is read by the _xp backward only.  param == NULL: no optimizer step.  N even, L <= 64. */
int snf_hashgrid_sort_xp(const float* u, const float* scalings, int N, int L, int log2_T, void* workspace, int64_t workspace_bytes,
                         snf_stream_t stream);
int snf_hashgrid_bwd_presorted_adam_xp(const float* grad_out, int N, int L, int log2_T, int ld_out, int col_off, int n_run_levels,
                                       float* grad_table, const void* sorted_workspace, float* stage, int fuse_from_level,
                                       float* param, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                       int step, float grad_scale, snf_stream_t stream);

/* snf_hashgrid_bwd_presorted with the optimizer step folded into its reduce pass for the levels >= fuse_from_level: the
 * workgroup that owns a bucket of rows holds their complete gradient after its last chunk and applies
 * torch.optim.Adam's update (the arithmetic of snf_adam_step: eps outside the bias-corrected sqrt, gradient pre-scaled by
 * grad_scale) to param / exp_avg / exp_avg_sq of those rows -- 24 B per parameter instead of 8 B (gradient
 * read-modify-write) + 32 B (snf_adam_step).  grad_table is left all-zero on those levels.  Replaces, for the hash tables,
 * torch.optim.Adam.step of nerfstudio/engine/optimizers.py:131-147.  Only valid when this call carries the table's whole
 * gradient of the step (one backward per step; one rank, or levels owned by this rank).  Levels < fuse_from_level behave as
 * in snf_hashgrid_bwd_presorted (the caller steps them, e.g. with snf_adam_step_rows). */
int snf_hashgrid_bwd_presorted_adam(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                    int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                    int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                    float beta1, float beta2, float eps, int step, float grad_scale, snf_stream_t stream);

/* Reachable-row levels (round 3).  A level of resolution s addresses at most (s + 3)^3 of its 2^log2_T rows (Encoding.active_rows
 * on the host side; exact: the other rows never receive a gradient and torch.optim.Adam leaves a parameter with g = m = v = 0
 * where it is).  For the leading `sparse_levels` levels of a table the caller passes the static lists of those rows --
 * reach_rows: (level << log2_T) + row, ascending; reach_start[level * B + bucket], B = 2^snf_hashgrid_bucket_bits(N, log2_T), with
 * one closing entry -- and the reduce pass handles those levels by a fixed-point sum over COMPACT row indices with replicated
 * accumulators (no same-address pile-up where hundreds of records share a row) and applies Adam to exactly the listed rows, zero
 * gradient or not: the call then steps the WHOLE table (no gradient is written for those levels, no snf_adam_step_rows pass).
 * sparse_step = 0: those levels only add their sums to grad_table (gradient exchange between ranks first, optimizer off).
 * sparse_levels <= fuse_from_level; sparse_max_rows = the longest list of a bucket, at most snf_hashgrid_sparse_max_rows(F);
 * scratch: SNF_HG_FX_SCRATCH_BYTES private to the call (may be NULL for F = 2 or without reachable-row levels).  Replaces, for the
 * hash tables, torch.optim.Adam.step of nerfstudio/engine/optimizers.py:131-147 like snf_hashgrid_bwd_presorted_adam. */
int snf_hashgrid_bucket_bits(int N, int log2_T);
int snf_hashgrid_sparse_max_rows(int F);
int snf_hashgrid_bwd_presorted_adam_sp(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                       int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                       int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                       float beta1, float beta2, float eps, int step, float grad_scale,
                                       const uint32_t* reach_rows, const uint32_t* reach_start, int sparse_levels,
                                       int sparse_max_rows, int sparse_step, void* scratch, snf_stream_t stream);

/* snf_hashgrid_bwd_presorted_adam_sp for the TWO F = 8 grids of a feature head (clip_encs / clipseg_encs, samnerf/sam_field.py:38-94:
 * same samples N, same log2_T, level-major gradients = ld_out 0): the bucket-wide levels of both tables in ONE reduce launch (one
 * tail instead of two), the reachable-row levels of each in front of it.  Same arithmetic per table as two single-table calls;
 * the hyper-parameters are the group's.  scratch: SNF_HG_FX_SCRATCH_BYTES, needed when either table has reachable-row levels. */
int snf_hashgrid_bwd_presorted_adam_pair(const float* grad_out0, const float* grad_out1, int N, int L0, int L1, int log2_T,
                                         float* grad_table0, float* grad_table1, const void* sorted_workspace0,
                                         const void* sorted_workspace1, int fuse_from_level0, int fuse_from_level1, float* param0,
                                         float* exp_avg0, float* exp_avg_sq0, float* param1, float* exp_avg1, float* exp_avg_sq1,
                                         const uint32_t* reach_rows0, const uint32_t* reach_start0, int sparse_levels0,
                                         const uint32_t* reach_rows1, const uint32_t* reach_start1, int sparse_levels1,
                                         int sparse_max_rows, int sparse_step, void* scratch, float lr, float beta1, float beta2,
                                         float eps, int step, float grad_scale, snf_stream_t stream);

/* The same pass with FIXED-POINT per-row sums (F = 2 and F = 8): a contribution w * g is added to its row as a 64-bit integer
 * LDS atomic, q = rint(w g 2^s) with 2^s = 2^38 / 2^e and 2^e above the level's largest finite |g| (found by a small
 * pre-pass), instead of being sorted by row and summed in fp32 -- ds_add_u64 retires 6.3 lane-ops/clk/CU on gfx950 against
 * 0.37 for ds_add_f32.  The sum is exact, hence independent of the order of the records (bit-reproducible table gradients);
 * every contribution is resolved to 2^-38 of the level's largest gradient; rows that receive a non-finite contribution
 * become NaN as a float sum would.  `scratch`: SNF_HG_FX_SCRATCH_BYTES, private to the launch -- the sorted workspace stays
 * read-only, so grids sharing a sort may run their backward concurrently.  fuse_from_level = L steps no level (gradient
 * accumulation only).  snf_hashgrid_bwd_presorted[_adam] take this path by themselves for F = 2 (scratch inside their
 * workspace; SNF_HG_FX=0 restores the float reduce). */
#define SNF_HG_FX_SCRATCH_BYTES 256
int snf_hashgrid_bwd_presorted_adam_fx(const float* grad_out, int N, int L, int F, int log2_T, int ld_out, int col_off,
                                       int n_run_levels, float* grad_table, const void* sorted_workspace, float* stage,
                                       int fuse_from_level, float* param, float* exp_avg, float* exp_avg_sq, float lr,
                                       float beta1, float beta2, float eps, int step, float grad_scale, void* scratch,
                                       snf_stream_t stream);

/* Render path only (no backward): a feature head's two F = 8 hash grids and its first, hidden layer in ONE kernel -- SAMField.
 * get_outputs' grids -> cat -> first CutlassMLP layer (samnerf/sam_field.py:112-140) + ReLU + MeanRenderer over groups of 16 samples
 * (samnerf/sam_model.py:126-137), as evaluated by the render pass of samnerf/sam_model.py:337-419.  The per-sample features of a
 * 64-sample tile are interpolated straight into LDS as bf16 hi / lo planes and feed the matrix cores from there; the [N, 8 (LA + LB)]
 * encoding never reaches HBM.  Hbar [N / 16, O] = sum_k row_weight[n] relu(enc[n] W^T); the caller applies the linear last layer to it.
 * snf_split_weights_b3 prepares W [O, I] (fp32, constant during a render) as two bf16 planes in matrix-operand order, (O * I) bf16
 * each.  Needs LA + LB even and <= 32, O in {128, 256}, group == 16, N % 64 == 0. */
int snf_split_weights_b3(const float* W, int O, int I, void* hi_plane, void* lo_plane, snf_stream_t stream);
int snf_grid_head_fused_fwd(const float* u, const float* tableA, const float* scalingsA, int LA, const float* tableB,
                            const float* scalingsB, int LB, int log2_T, const void* Whi, const void* Wlo, int O,
                            const float* row_weight, int group, float* Hbar, int N, snf_stream_t stream);

/* Arithmetic of the wide (>= 128 input) dense layers: 1 (default) = bf16 3-term split on the bf16 matrix cores with fp32
 * accumulate (max abs error ~1e-6 on head-shaped data, 1/5 of the matrix cycles), 0 = exact fp32 matrix cores,
 * 2 = as 1 and the fused 64-wide chains (snf_mlp64_*) on the same split (opt-in: -9 % on those kernels, 6x their round-off).
 * Process-wide; other narrow layers always run exact fp32. */
int snf_set_gemm_mode(int mode);
int snf_get_gemm_mode(void);

/* ---- a7: one layer of tcnn.Network (FullyFusedMLP / CutlassMLP) == nerfstudio MLP layer
 *      (field_components/mlp.py:80-99): Y[N,O] = act(X[N,I] W[O,I]^T + bias).  bias may be NULL. */
int snf_linear_fwd(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx,
                   int ldy, int act, float* Y, snf_stream_t stream);
/* The same layer with a caller-provided scratch buffer: when the 128 x 64 output tiles of a long-K layer do not fill the
 * chip (the conv head's [4096 | 256, 2304] x [2304, 256] GEMMs) the k range is split over workgroups and the partial
 * products are summed, biased and activated by a second small kernel.  snf_linear_fwd_workspace_bytes returns the scratch
 * size this shape wants (0: no split; workspace may then be NULL and the call equals snf_linear_fwd). */
int64_t snf_linear_fwd_workspace_bytes(int N, int I, int O);
int snf_linear_fwd_ws(const float* X, const float* W, const float* bias, int N, int I, int O, int ldx, int ldy, int act,
                      float* Y, void* workspace, int64_t workspace_bytes, snf_stream_t stream);
/* Level-major operands (ld = -8): a matrix with I % 8 == 0 columns stored as [I/8][N][8] -- what two F = 8 feature grids
 * write side by side with snf_hashgrid_fwd(ld_out = 0) into one buffer (grid g at float offset g * L * N * 8).  Accepted as X
 * by snf_linear_fwd (ldx = -8) and snf_linear_bwd_weight (ldx = -8), produced as dX by snf_linear_bwd_data (lddx = -8), whose
 * output is then the staged gradient of snf_hashgrid_bwd_presorted[_adam] (ld_out = 0, pointer offset per grid): the
 * [N, 192] row-major encoding of samnerf/sam_field.py:121-137 and its gradient never exist.  Needs gemm mode >= 1,
 * 64 <= I <= 256, I % 16 == 0, O >= 64. */
/* dX[N,I] = (dY * act'(Y)) W ;  Y is the layer OUTPUT (post-activation), may be NULL when act == NONE. */
int snf_linear_bwd_data(const float* dY, const float* Y, const float* W, int N, int I, int O, int lddy,
                        int ldy, int lddx, int act, float* dX, snf_stream_t stream);
/* dW[O,I] += (dY * act'(Y))^T X ; dbias[O] += column sums (dbias may be NULL).  Atomic accumulation
 * into caller-zeroed (or running) buffers. */
int snf_linear_bwd_weight(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy,
                          int ldy, int ldx, int act, float* dW, float* dbias, snf_stream_t stream);

/* The weight gradient with a caller-provided scratch buffer (size: snf_linear_bwd_weight_workspace_bytes; 0 = this shape does
 * not use one).  The feature-head layers (64 <= I, O <= 256, N >= 8192, no bias, gemm mode >= 1) then run a full-width
 * kernel: a workgroup owns a chunk of rows and the whole O x I output, so dY, Y and X are read from HBM exactly once (the
 * tiled kernel re-reads them once per 64 x 64 output tile), per-chunk partial sums go to the scratch buffer and a second
 * small kernel adds them to dW.  Other shapes, a NULL or short workspace: identical to snf_linear_bwd_weight. */
int64_t snf_linear_bwd_weight_workspace_bytes(int N, int I, int O);
int snf_linear_bwd_weight_ws(const float* dY, const float* Y, const float* X, int N, int I, int O, int lddy, int ldy, int ldx,
                             int act, float* dW, float* dbias, void* workspace, int64_t workspace_bytes, snf_stream_t stream);

/* A hidden layer whose ReLU output is only ever (a) rendered -- the weighted mean over `group` consecutive rows (MeanRenderer
 * over the K samples of a ray, samnerf/sam_model.py:126-137, applied BEFORE the network's linear last layer:
 * sum_k w_k (W h_k) = W (sum_k w_k h_k)) -- and (b) differentiated through its ReLU:
 *   snf_linear_fwd_mean   Hbar[n / group, :] = sum_k row_weight[n] * relu(X W^T)[n, :]  ([N/group, O]) and the ReLU mask as bits,
 *                         Ymask [N][O/8] bytes, bit c of a row = (y[c] > 0); the activations themselves go to Y only if Y != NULL.
 *   snf_linear_bwd_*_rows the layer's two gradients when dY[n,:] = row_scale[n] * dYg[n / group, :] (dYg [N/group, lddy],
 *                         row_scale [N] = the detached rendering weights): the loaders form that product, the [N, O] gradient
 *                         snf_feature_mean_bwd would write is never materialised.  y_is_mask != 0: Y is the bit mask above
 *                         (ldy = bytes per row) instead of the fp32 activations.  Same bits as the unfused sequence.
 * Only the weight-stationary / full-width bf16-split kernels implement these (SNF_ERR_ARG otherwise, conditions in the error
 * text): N >= 8192, 64 <= I, O <= 256, gemm mode >= 1, snf_linear_bwd_weight_workspace_bytes(N, I, O) > 0. */
int snf_linear_fwd_mean(const float* X, const float* W, int N, int I, int O, int ldx, const float* row_weight, int group,
                        float* Hbar, uint8_t* Ymask, float* Y, int ldy, snf_stream_t stream);
int snf_linear_bwd_data_rows(const float* dYg, const float* row_scale, int group, const float* Y, int y_is_mask, const float* W,
                             int N, int I, int O, int lddy, int ldy, int lddx, int act, float* dX, snf_stream_t stream);
int snf_linear_bwd_weight_rows(const float* dYg, const float* row_scale, int group, const float* Y, int y_is_mask, const float* X,
                               int N, int I, int O, int lddy, int ldy, int ldx, int act, float* dW, void* workspace,
                               int64_t workspace_bytes, snf_stream_t stream);

/* ---- a7, tiny: the proposal networks' density MLP (nerfstudio/fields/density_fields.py:80-97 with hidden_dim 16:
 *      I -> H (ReLU) -> 1, bias-free) in one launch per direction, one thread per sample.  Built for I = 10, H = 16
 *      (snf_mlp_tiny_supported).  Hid [N,H] receives the hidden activations (may be NULL at inference); the backward
 *      writes dX [N, lddx] (may be NULL) and ACCUMULATES dW0 [H,I], dW1 [H] (one set of atomics per workgroup). */
int snf_mlp_tiny_supported(int I, int H, int O);
int snf_mlp_tiny_fwd(const float* X, int ldx, const float* W0, const float* W1, int I, int H, int64_t N, float* Hid,
                     float* Y, snf_stream_t stream);
/* The proposal density of the eval path in one launch (HashMLPDensityField.get_density, density_fields.py:99-127, no gradients):
 * density[n] = exp(mlp(hashgrid(u[n]))) * selector[n] == snf_hashgrid_fwd (row-major) + snf_mlp_tiny_fwd + snf_trunc_exp_fwd, identical
 * values, no [N, 10] encoding written.  L = 5, F = 2, H = 16 only (snf_mlp_tiny_supported). */
int snf_prop_density_fwd(const float* u, const float* table, const float* scalings, int N, int L, int F, int log2_T, const float* W0,
                         const float* W1, int H, const uint8_t* selector, float* density, snf_stream_t stream);
int snf_mlp_tiny_bwd(const float* dY, const float* X, int ldx, const float* Hid, const float* W0, const float* W1, int I,
                     int H, int64_t N, float* dX, int lddx, float* dW0, float* dW1, snf_stream_t stream);

/* ---- a16: the SAM conv head (samnerf/sam_model.py:196-200,259-264: Conv2d(C,C,k,padding=k/2) -> ReLU -> Conv2d -> mean over
 *      the p x p patch) as GEMMs.  Features stay channel-last [R, C], row = patch*p*p + y*p + x (MeanRenderer's layout).
 *   unfold      : col[row, c*k*k + t] = x[patch, y+dy_t, x+dx_t, c] (0 outside the patch), t = ky*k + kx -- the column
 *                 order of Conv2d.weight [O, C, k, k] viewed as [O, C*k*k], so  conv(x) = snf_linear_fwd(col, W, bias).
 *   fold        : adjoint of unfold (dcol [R, C*k*k] -> dx [R, C]).
 *   unfold_mean : cm[patch, c*k*k + t] = mean over the patch rows of unfold(h): the patch mean moved in front of the
 *                 second convolution's GEMM (both linear), so that GEMM runs on R/p^2 rows.
 *   fold_mean   : adjoint of unfold_mean (dcm [R/p^2, C*k*k] -> dh [R, C]).
 * R must be a multiple of p*p; k odd <= 5; p <= 64 for unfold (the image encoder's 64 x 64 neck), p <= 8 for the others. */
int snf_patch_unfold(const float* x, int R, int p, int C, int k, float* col, snf_stream_t stream);
int snf_patch_fold(const float* dcol, int R, int p, int C, int k, float* dx, snf_stream_t stream);
int snf_patch_unfold_mean(const float* h, int R, int p, int C, int k, float* cm, snf_stream_t stream);
int snf_patch_fold_mean(const float* dcm, int R, int p, int C, int k, float* dh, snf_stream_t stream);

/* ---- a7, fused: a whole 64-wide tiny MLP (tcnn FullyFusedMLP: nerfstudio/fields/nerfacto_field.py:157-175,228-240)
 *      in one launch, activations in registers.  Layers: W0 [64, in_real] (in_real <= 32), W1 [64,64] (n_hidden == 2
 *      only), Wout [out, 64] (out <= 32); ReLU between layers, out_act on the output (NONE or SIGMOID); no biases.
 * X is [N, ldx] with ldx >= 32, ldx % 4 == 0 (columns in_real..31 are ignored but must be readable).
 * H1, H2 ([N,64], may be NULL at inference) receive the hidden activations needed by the backward. */
int snf_mlp64_fwd(const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout,
                  int n_hidden, int out, int out_act, int64_t N, float* H1, float* H2, float* Y, int ldy,
                  snf_stream_t stream);
/* snf_mlp64_fwd + snf_trunc_exp_fwd(Y, ldy, selector, N, density) in one call (the field's base MLP followed by trunc_exp of its output
 * 0, nerfacto_field.py:244-252): for the base net's shape (one hidden layer, 16 linear outputs, H1 = H2 = NULL) the density leaves from
 * the chain's epilogue instead of being re-read at a 64-byte stride; any other shape runs the two kernels.  Identical values. */
int snf_mlp64_fwd_density(const float* X, int ldx, const float* W0, int in_real, const float* W1, const float* Wout, int n_hidden,
                          int out, int out_act, int64_t N, float* H1, float* H2, float* Y, int ldy, const uint8_t* selector,
                          float* density, snf_stream_t stream);
/* data-gradient chain of the same net: dZ[s][o] = dY[s*lddy + dy_col_off + o] (column 0 taken from dY0[s] when dY0 !=
 * NULL), times the sigmoid derivative of Y when out_act == SIGMOID.  Writes the ReLU-masked hidden gradients dH2, dH1
 * ([N,64]), the pre-activation output gradient dZ ([N, lddz >= out], may be NULL) and dX ([N, lddx >= 32], may be NULL).  The
 * weight gradients are then snf_linear_bwd_weight(dZ, Hlast), (dH2, H1), (dH1, X) with act = NONE. */
int snf_mlp64_bwd_data(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy,
                       const float* W0, int in_real, const float* W1, const float* Wout, int n_hidden, int out,
                       int out_act, int64_t N, const float* H1, const float* H2, float* dH1, float* dH2, float* dZ,
                       int lddz, float* dX, int lddx, snf_stream_t stream);

/* The data-gradient chain AND the weight gradients of the same net in one pass: the wave that holds dH^T (and reads H for the
 * ReLU mask anyway) forms dW = dA^T B itself on the bf16 matrix cores (3-term split, fp32 accumulate) through a per-wave LDS
 * transpose, so dH1 / dH2 / dZ are never written and X / H1 / H2 are read once instead of twice (-1.2 GB of HBM traffic per
 * train step for the two nerfacto nets).  dW0 [64, in_real], dW1 [64,64], dWout [out,64] are ACCUMULATED (plain adds by one
 * reduce kernel over per-workgroup partials in `workspace`: snf_mlp64_bwd_fused_workspace_bytes(n_hidden) bytes).  X as in
 * snf_mlp64_fwd (ldx = 0: level-major).  Replaces snf_mlp64_bwd_data + three snf_linear_bwd_weight calls.  * H1 = H2 = NULL ("recompute", gemm mode 1): the hidden activations are formed again from X inside the kernel with the forward's
 * own arithmetic (bit-identical), so snf_mlp64_fwd may be called with H1 = H2 = NULL as well and the [N,64] activations never reach
 * HBM. */
int64_t snf_mlp64_bwd_fused_workspace_bytes(int n_hidden);
int snf_mlp64_bwd_fused(const float* dY, int lddy, int dy_col_off, const float* dY0, const float* Y, int ldy, const float* X,
                        int ldx, const float* W0, int in_real, const float* W1, const float* Wout, int n_hidden, int out,
                        int out_act, int64_t N, const float* H1, const float* H2, float* dX, int lddx, float* dW0, float* dW1,
                        float* dWout, void* workspace, int64_t workspace_bytes, snf_stream_t stream);

/* The colour net (fields/nerfacto_field.py:336-351: `h = cat([SH16(d), geo])` -> mlp_head) with its input row FORMED in the
 * kernel's loader instead of read: features 0..15 = the degree-4 spherical harmonics of the sample's ray direction (utils/math.py:
 * 27-73, the arithmetic of snf_head_input bit for bit), features 16..30 = columns 1..15 of the base net's output row base_out
 * [R*S, ld_base >= 16].  The [N, 32] input tensor of snf_head_input + snf_mlp64_fwd is never written.  gemm mode 1; n_geo = 15.
 * snf_mlp64_bwd_fused_sh is the matching recomputing backward (H1 = H2 = NULL semantics of snf_mlp64_bwd_fused): dY [N, lddy]
 * (columns 0 .. out-1), and only the gradient that has a consumer leaves -- d_geo [N, ld_dgeo >= 16], column j = d(feature 16 + j)
 * (the base net's backward then reads it as its output gradient with dy_col_off = -1). */
int snf_mlp64_fwd_sh(const float* dirs, int R, int S, const float* base_out, int ld_base, int n_geo, const float* W0,
                     const float* W1, const float* Wout, int n_hidden, int out, int out_act, float* H1, float* H2, float* Y,
                     int ldy, snf_stream_t stream);
int snf_mlp64_bwd_fused_sh(const float* dY, int lddy, const float* Y, int ldy, const float* dirs, int R, int S,
                           const float* base_out, int ld_base, int n_geo, const float* W0, const float* W1, const float* Wout,
                           int n_hidden, int out, int out_act, float* d_geo, int ld_dgeo, float* dW0, float* dW1, float* dWout,
                           void* workspace, int64_t workspace_bytes, snf_stream_t stream);

/* ---- a13: SH degree-4 basis of the raw unit direction (nerfstudio/utils/math.py:27-73) written to
 *      the first 16 columns of the colour-MLP input, with the geo features copied behind it
 *      (torch.cat of fields/nerfacto_field.py:336-343).  dirs [R,3]; geo points at h[:,1] of the
 *      base-MLP output [R*S, ld_geo]; out [R*S, ld_out] gets 16 + n_geo columns. */
int snf_head_input(const float* dirs, const float* geo, int R, int S, int n_geo, int ld_geo, float* out,
                   int ld_out, snf_stream_t stream);

/* ---- a8+a9: trunc_exp (field_components/activations.py:24-40), selector mask and
 *      RaySamples.get_weights (nerfstudio/cameras/rays.py:141-163).
 * raw[(r*n+i)*raw_stride] = pre-activation density (is_density == 0: sigma = exp(raw)*selector, backward
 * through the truncated exp) or the density itself (is_density == 1: plain get_weights(densities));
 * selector [R*n] uint8 or NULL; ebins [R,n+1].  Out: weights [R,n]; density [R,n] (may be NULL). */
int snf_weights_fwd(const float* raw, int raw_stride, int is_density, const uint8_t* selector,
                    const float* ebins, int R, int n, float* weights, float* density, snf_stream_t stream);
/* grad_raw[(r*n+i)*raw_stride] = dL/draw  (overwrites that column only). */
int snf_weights_bwd(const float* raw, int raw_stride, int is_density, const uint8_t* selector,
                    const float* ebins, const float* grad_weights, int R, int n, float* grad_raw,
                    snf_stream_t stream);

/* ---- a8 alone: density = trunc_exp(raw) * selector, the tail of Field.get_density
 *      (fields/nerfacto_field.py:260-265, fields/density_fields.py:120-124, activations.py:24-40).
 * raw[t*raw_stride], selector [N] uint8 or NULL -> density [N]; backward writes grad_raw[t*grad_stride]. */
int snf_trunc_exp_fwd(const float* raw, int raw_stride, const uint8_t* selector, int64_t N, float* density,
                      snf_stream_t stream);
int snf_trunc_exp_bwd(const float* raw, int raw_stride, const uint8_t* selector, const float* grad_density,
                      int64_t N, float* grad_raw, int grad_stride, snf_stream_t stream);

/* ---- a10: PDFSampler.generate_ray_samples, include_original=False, single jitter
 *      (model_components/ray_samplers.py:298-367), preceded by the anneal pow of :583.
 * weights [R,P]; sbins_in [R,P+1]; u_rand [R] or NULL (eval); nears,fars [R].
 * Out: sbins, ebins [R,S+1]. */
int snf_pdf_resample(const float* weights, const float* sbins_in, const float* u_rand, const float* nears,
                     const float* fars, int R, int P, int S, float anneal, float histogram_padding,
                     float* sbins, float* ebins, snf_stream_t stream);

/* ---- a14+a15: RGBRenderer('last_sample'), AccumulationRenderer, DepthRenderer('median')
 *      (model_components/renderers.py:97-140,218-223,260-270).
 * rgb [R,S,3]; weights [R,S]; ebins [R,S+1].  Any of the outputs may be NULL. */
int snf_composite_fwd(const float* rgb, const float* weights, const float* ebins, int R, int S,
                      int training, float* out_rgb, float* out_acc, float* out_depth,
                      snf_stream_t stream);
/* training-mode backward of the RGB render: grad_rgb [R,S,3], grad_weights [R,S] (overwritten). */
int snf_composite_bwd(const float* rgb, const float* weights, const float* grad_out_rgb, int R, int S,
                      float* grad_rgb, float* grad_weights, snf_stream_t stream);

/* ---- a16: torch.topk(weights, K) + sharpening + renormalisation (samnerf/sam_model.py:244-248).
 * Out: ids [R,K] int32 (descending weight, ties -> lower index), sam_weights [R,K] (0/0 -> NaN kept). */
int snf_topk_sharpen(const float* weights, int R, int S, int K, float temperature, int32_t* ids,
                     float* sam_weights, snf_stream_t stream);
/* Row-mapped form (see snf_positions_rows): weights row src_rows[j] -> ids / sam_weights row dst_rows[j], j < M. */
int snf_topk_sharpen_rows(const float* weights, const int32_t* src_rows, const int32_t* dst_rows, int M, int S,
                          int K, float temperature, int32_t* ids, float* sam_weights, snf_stream_t stream);

/* ---- a18: MeanRenderer (samnerf/sam_model.py:126-137): out[r,c] = sum_k w[r,k] * embeds[r,k,c]. */
int snf_feature_mean_fwd(const float* embeds, const float* w, int R, int K, int C, float* out,
                         snf_stream_t stream);
int snf_feature_mean_bwd(const float* grad_out, const float* w, int R, int K, int C, float* grad_embeds,
                         snf_stream_t stream);

/* ---- a20: interlevel_loss / lossfun_outer / outer (model_components/losses.py:46-120), one
 *      proposal level.  Out: loss_rows [R] = sum_s clip(w-w_outer,0)^2/(w+1e-7) (caller takes the mean
 *      over R*S); grad_w_prop [R,P] = d(mean loss)/d w_prop * grad_scale (NULL to skip). */
int snf_interlevel(const float* sbins_fine, const float* w_fine, const float* sbins_prop,
                   const float* w_prop, int R, int S, int P, float grad_scale, float* loss_rows,
                   float* grad_w_prop, snf_stream_t stream);

/* ---- a21: distortion_loss / lossfun_distortion (model_components/losses.py:124-143).
 * Out: loss_rows [R]; grad_w [R,S] = d(loss_row)/dw * grad_scale (NULL to skip). */
int snf_distortion(const float* sbins, const float* w, int R, int S, float grad_scale, float* loss_rows,
                   float* grad_w, snf_stream_t stream);

/* ---- a19: L2 losses.  loss = weight * mean over rows r of mean_c (pred - target)^2 ; with nan_skip the mean runs over the
 * rows whose own mean is not NaN ( = mse_loss(.., 'none').mean(-1).nanmean(), samnerf/sam_model.py:316-328; without it
 * nn.MSELoss(), nerfstudio/models/nerfacto.py:326 ).  acc: SNF_ROWMSE_SCRATCH_WORDS zeroed scratch words (the ticket word is
 * left zeroed, so the buffer can be reused by later calls on the same stream); out: {loss, counted rows}.
 * bwd: dpred = gout * weight * 2 (pred - target) / (C * out[1]), zero for skipped rows; gout, out are device scalars. */
#define SNF_ROWMSE_SCRATCH_WORDS 516
int snf_rowmse_loss_fwd(const float* pred, const float* target, int R, int C, float weight, int nan_skip, float* acc,
                        float* out, snf_stream_t stream);
int snf_rowmse_loss_bwd(const float* pred, const float* target, int R, int C, float weight, int nan_skip,
                        const float* gout, const float* out, float* dpred, snf_stream_t stream);

/* ---- a19..a21, scalar tail: what autograd and the loss dict do around the kernels above, for a train step that is a
 * static launch schedule (samnerf_amd/step_program.py) instead of an autograd graph.
 * snf_add_scaled: y[i] += alpha * x[i], product rounded before the sum (= autograd scaling one branch's gradient by a loss
 *   multiplier and adding it to another branch's: nerfstudio/models/nerfacto.py:333 distortion_loss_mult).
 * snf_nerf_loss_summary: NerfactoModel.get_loss_dict / get_metrics_dict scalars (nerfacto.py:316-344) from the row sums:
 *   out = {total, rgb_loss (= rgb_mse[0]), interlevel_scale * sum(interlevel_rows), distortion_mult * distortion,
 *   distortion = distortion_scale * sum(distortion_rows), psnr = -10 log10(rgb_mse[0])}; interlevel_rows may be NULL. */
int snf_add_scaled(int64_t n, float alpha, const float* x, float* y, snf_stream_t stream);
int snf_nerf_loss_summary(const float* rgb_mse, const float* interlevel_rows, float interlevel_scale,
                          const float* distortion_rows, float distortion_scale, float distortion_mult, int R, float* out,
                          snf_stream_t stream);

/* ---- the SAM image encoder forward (samnerf/segment_anything/modeling/image_encoder.py, common.py).
 * Dense layers are snf_linear_fwd (SNF_ACT_GELU for the MLP); these are the pieces around them.  Tokens are rows [B*T, C].
 * snf_patchify: image [B,Cin,S,S] -> rows [B*(S/P)^2, Cin*P*P] in Conv2d.weight.view(E, Cin*P*P) column order (PatchEmbed).
 * snf_layernorm: y = LayerNorm(x + residual) (residual may be NULL; sum_out, if given, receives x + residual); also
 *   LayerNorm2d on channel-last rows.
 * snf_window_partition / snf_window_merge_add: window_partition with zero padding; window_unpartition fused with the
 *   block's `shortcut + x` (ws == 0: plain add).
 * snf_relpos: rel[bh][i][0..n) = q_i . rel_pos_h[ih-kh+n-1], rel[bh][i][n..2n) = q_i . rel_pos_w[iw-kw+n-1] from the qkv rows
 *   [Bw*T, 3*C] (add_decomposed_rel_pos; tables must have 2n-1 rows).  In gemm mode >= 1 grids with n >= 32 (the 64 x 64 global
 *   blocks) form the products on the matrix cores with the 3-term bf16 split (1e-6 relative, as snf_attention_planes_rp does inside the
 *   windowed blocks); gemm mode 0 and small grids keep the fp32 vector-ALU kernels.
 * snf_attention: out[b*T+i, h*hd..] = softmax(scale * q k^T + rel_h + rel_w) v for every (window b, head h), q/k/v read in
 *   place from the qkv rows; rel may be NULL; head_dim <= 96. */
int snf_patchify(const float* img, int B, int Cin, int S, int P, float* rows, snf_stream_t stream);
/* Sam.preprocess (samnerf/segment_anything/modeling/sam.py:164-174), the step between SamPredictor.set_torch_image
 * (predictor.py:70-97) and the encoder: out[b,c,y,x] = (img[b,c,y,x] - mean[c]) / std[c] inside the h x w image, 0 in the padding
 * to S x S.  img: uint8 (is_uint8 != 0; what set_image hands over) or fp32, [B,C,h,w]; mean / std [C] on the device. */
int snf_sam_preprocess(const void* img, int is_uint8, int B, int C, int h, int w, int S, const float* mean, const float* stdv,
                       float* out, snf_stream_t stream);
int snf_layernorm(const float* x, const float* residual, int N, int C, const float* weight, const float* bias, float eps,
                  float* sum_out, float* y, snf_stream_t stream);
int snf_window_partition(const float* x, int B, int H, int W, int C, int ws, float* out, snf_stream_t stream);
int snf_window_merge_add(const float* windows, const float* shortcut, int B, int H, int W, int C, int ws, float* out,
                         snf_stream_t stream);
int snf_relpos(const float* qkv, int Bw, int T, int heads, int head_dim, int n, const float* rel_pos_h,
               const float* rel_pos_w, float* rel, snf_stream_t stream);
int snf_attention(const float* qkv, const float* rel, int Bw, int T, int heads, int head_dim, int n, float scale, float* out,
                  snf_stream_t stream);

/* The encoder block's four token GEMMs (attn.qkv / attn.proj / mlp.lin1 / mlp.lin2, image_encoder.py:164-236, common.py:13-28)
 * on operands that arrive already split into bf16 hi / lo planes (x = hi + lo; a product = hi*hi + hi*lo + lo*hi on the bf16
 * matrix cores, fp32 accumulate: the arithmetic of snf_linear_fwd in gemm mode 1, without its per-tile VALU split).
 *   row-major planes  [Nc][K]      : the constant weights, split once (snf_split_planes; n = Nc*K elements);
 *   k-blocked planes  [K/8][M][8]  : the activations, written by their producer -- snf_layernorm_planes (LayerNorm, with the
 *     window partition's row map: token (b,y,x) lands at its window row, padded rows are never written and must be zero),
 *     snf_attention_planes, or snf_linear_planes_fwd itself (c_hi / c_lo: lin1's GELU output is lin2's operand);
 *     snf_split_planes_kb converts a row-major fp32 matrix (tests, odd producers).
 * snf_linear_planes_fwd: C = act(A W^T + bias), act in {NONE, RELU, GELU}; K % 64 == 0, Nc % 64 == 0; fp32 output C [M][Nc]
 *   and / or plane output [Nc/8][M][8]; all pointers 16-byte aligned. */
int snf_split_planes(const float* x, int64_t n, uint16_t* hi, uint16_t* lo, snf_stream_t stream);
int snf_split_planes_kb(const float* x, int M, int K, uint16_t* hi, uint16_t* lo, snf_stream_t stream);
int snf_linear_planes_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                          int M, int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo, snf_stream_t stream);
/* (tuning / benchmarking: the same with the tile shape (128 rb rows x 32 nb columns; rb = -1: 256 rows, 512-thread workgroup) given
 * instead of chosen) */
int snf_linear_planes_fwd_shape(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo,
                                const float* bias, int M, int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo, int rb,
                                int nb, snf_stream_t stream);
/* ... with the weights k-blocked as well (w_hi / w_lo [K/8][Nc][8] = snf_split_planes_kb of the [Nc][K] matrix): 256 x 320 / 256 x 256
 * tiles with both operands staged through LDS; K % 64 == 0, Nc % 320 == 0 or Nc % 256 == 0.  Same products in the same order per
 * output as snf_linear_planes_fwd (bit-identical results). */
int snf_linear_planes_kb_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, const float* bias,
                             int M, int K, int Nc, int act, float* C, uint16_t* c_hi, uint16_t* c_lo, snf_stream_t stream);
int snf_layernorm_planes(const float* x, const float* residual, int N, int C, const float* weight, const float* bias, float eps,
                         float* sum_out, uint16_t* y_hi, uint16_t* y_lo, int M_out, int H, int W, int ws, snf_stream_t stream);
/* ... with the block's window_unpartition + `shortcut + x` (snf_window_merge_add) folded in: residual_windows are the projection's
 * output rows in WINDOW order ([B * nWh * nWw * res_ws^2, C]); token (b, y, x) adds the row of its window position; sum_out (required)
 * receives the merged token rows, the planes their LayerNorm (norm2 of image_encoder.py:176-181). */
int snf_layernorm_planes_merge(const float* x, const float* residual_windows, int N, int C, const float* weight, const float* bias,
                               float eps, float* sum_out, uint16_t* y_hi, uint16_t* y_lo, int H, int W, int res_ws, snf_stream_t stream);
int snf_attention_planes(const float* qkv, const float* rel, int Bw, int T, int heads, int head_dim, int n, float scale,
                         uint16_t* out_hi, uint16_t* out_lo, snf_stream_t stream);
/* ... for small grids (2n - 1 <= 32: the 14 x 14 windows) with snf_relpos folded in: the decomposed relative-position terms are formed
 * inside the kernel, on the matrix cores, from rel_pos_h / rel_pos_w [2n-1][head_dim] (add_decomposed_rel_pos, image_encoder.py:323-361) */
int snf_attention_planes_rp(const float* qkv, const float* rel_pos_h, const float* rel_pos_w, int Bw, int T, int heads, int head_dim,
                            int n, float scale, uint16_t* out_hi, uint16_t* out_lo, snf_stream_t stream);

/* ---- the batch builder in front of the path (images, feature maps and cameras resident in HBM).
 * snf_pixel_indices: PixelSampler (patch == 1: u [B,3]) / PatchPixelSampler (u [B/patch^2,3]) .sample_method without a
 *   mask (nerfstudio/data/pixel_samplers.py:50-75,246-300): u ~ U[0,1) -> indices [B,3] int64 (camera, row, col).
 * snf_generate_rays: RayGenerator.forward + pinhole Cameras._generate_rays_from_coords, camera optimizer off, no
 *   distortion (model_components/ray_generators.py:44-63, cameras/cameras.py:576-722): c2w [N,3,4], intrinsics [N,4] =
 *   fx, fy, cx, cy -> origins/directions [R,3], pixel_area [R], camera_indices [R] int64.
 * snf_gather_nearest: FeatureDataloader.__call__ (samnerf/data/feature_loader.py:49-56): out[b] = features[cam,
 *   long(row * fh/H), long(col * fw/W)] for the points b*point_stride + point_offset of `points` [.,3]
 *   (stride p^2, offset (p/2)*p + p/2 = the patch centres of samnerf/datamanager.py:106-110; fh = H, fw = W gathers the
 *   image pixels themselves).  features [N, fh, fw, C] fp32. */
int snf_pixel_indices(const float* u, int B, int patch, int num_images, int H, int W, int64_t* indices,
                      snf_stream_t stream);
int snf_generate_rays(const int64_t* indices, int R, const float* c2w, const float* intrinsics, int num_cameras,
                      float* origins, float* directions, float* pixel_area, int64_t* camera_indices, snf_stream_t stream);
int snf_gather_nearest(const int64_t* points, int B, int point_stride, int point_offset, const float* features,
                       int num_images, int fh, int fw, int C, int H, int W, float* out, snf_stream_t stream);

/* ---- optimiser side (nerfstudio/engine/optimizers.py:100-147; torch.optim.Adam, eps 1e-15,
 *      samnerf/samconfigs.py:144-161): fused Adam over one contiguous parameter-arena slice.
 * grads are multiplied by grad_scale first (1/world_size for the data-parallel mean) and are
 * zeroed afterwards when zero_grad != 0. */
int snf_adam_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, int zero_grad, snf_stream_t stream);

/* The same update on a LIST of table rows: rows[i] = element offset (from the bases p, g, m, v) of a row of F = 2 or 8
 * floats.  For the coarse levels of a hash grid only the rows the level's lattice hashes to can ever receive a gradient;
 * every other row keeps g = m = v = 0, for which the Adam update is exactly the identity, so the optimizer visits the
 * reachable rows only (host side: tcnn_compat.Encoding.active_rows, engine.Optimizers). */
int snf_adam_step_rows(float* p, float* g, float* m, float* v, const int32_t* rows, int64_t nrows, int F, float lr,
                       float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad, snf_stream_t stream);

/* Non-finite-gradient guard.  Replaces torch.cuda.amp.GradScaler's inf / NaN check around the optimizer step
 * (nerfstudio/engine/trainer.py:419-437, engine/optimizers.py:138-149: `grad_scaler.step(optimizer)` does not call
 * optimizer.step() when a gradient of that optimizer is inf / NaN -- parameters, moments and state['step'] stay).  Adam is fused
 * into the table backward here, so the verdict has to exist BEFORE those launches and the host must not wait for it: a guard is a
 * device record int32[2] = {veto, skipped}, zero-initialised by the caller, one per group of losses whose gradients reach the
 * same parameters.
 *   snf_guard_update(values, n, guard): first commits the previous verdict (veto set -> skipped += 1), then veto = any of the n
 *     loss values is inf / NaN.  Enqueued behind the loss kernels of a step, in front of its optimizer-side launches.
 *   snf_guard_scan(x, n, guard): veto |= any of x[0 .. n) is inf / NaN.  For the dense parameters (MLP / conv weights) of the group:
 *     a loss can stay finite over an inf weight -- fmaxf(NaN, 0) = 0 in a ReLU, sigmoid(inf) = 1 -- while the gradients behind it are
 *     NaN (0 * inf); with finite losses AND finite weights every gradient of this fp32 path is finite unless a product overflows.
 *   snf_step_guard(guard): binds `guard` (or NULL: none) for the calling host thread.  The optimizer-side entry points issued while
 *     it is bound -- snf_adam_step, snf_adam_step_rows, snf_hashgrid_bwd_presorted_adam / _sp / _xp / _pair / _fx -- read the
 *     record on the device: on a vetoed step p / exp_avg / exp_avg_sq are left as they are (gradients are still cleared where
 *     zero_grad asks for it) and the bias corrections use `step - skipped`, so a vetoed step does not count.  With veto = skipped = 0
 *     every result is bit-identical to an unguarded launch. */
int snf_step_guard(const int32_t* guard);
int snf_guard_update(const float* values, int n, int32_t* guard, snf_stream_t stream);
int snf_guard_scan(const float* x, int64_t n, int32_t* guard, snf_stream_t stream);

/* Tuning hook: launch shape of the Adam kernel (grid cap, threads per block in {64,128,256}, independent 16-byte groups
 * per thread in {1,2,4}).  Process-wide; the default is the measured best for MI355X. */
int snf_set_adam_launch(int max_blocks, int threads, int unroll);

/* Counter-based deterministic fill: x[i] = lo + (hi-lo)*U(seed,i); identical to the CPU restatement in
 * the package (used to initialise full-size tables without shipping them). */
int snf_fill_uniform(float* x, int64_t n, uint64_t seed, float lo, float hi, snf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMNERF_HIP_H */
