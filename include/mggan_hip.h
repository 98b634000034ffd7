/* libmggan_hip.so -- C ABI of the MI355X (gfx950) hot path of MG-GAN training.
 *
 * The reference (selflein/MG-GAN) has no FFI of its own: its boundary for this
 * path is the Python class surface mggan.model.* (SURVEY.md 8b).  This header is
 * the drop-in boundary underneath that surface: every entry point below replaces
 * the implicit ATen/cuDNN/cuBLAS launches behind one reference call site, cited
 * per function as file:line under /root/reference/mggan.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer to
 *     contiguous f32 / int32 / int64 memory owned by the caller (PyTorch-ROCm
 *     tensors are storage only).  No allocation, no ownership transfer.
 *   - every call is asynchronous on `stream`; returns 0 or a negative code, text
 *     via mggan_last_error().  No global mutable state besides that error string.
 *   - "ld*" = row stride in floats, so outputs can be written straight into column
 *     slices of wider buffers (no concatenation kernels).
 *   - gradients of parameters are ACCUMULATED (+=) into caller-zeroed buffers;
 *     reductions are two-phase and deterministic (no float atomics).
 */
#ifndef MGGAN_HIP_H
#define MGGAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mggan_stream_t; /* == hipStream_t */

const char* mggan_last_error(void);
int mggan_version(void);
/* Measurement aid: *slot = the device's 100 MHz wall clock when the stream (or the captured graph) gets here. */
int mggan_timestamp(unsigned long long* slot, mggan_stream_t stream);
/* Measurement aid: while on, every kernel launch of the library is noted; mggan_launch_log_read writes
 * "symbol:threads;symbol:threads;..." (mangled device-function names, as rocprofv3 --kernel-trace reports them) of the
 * launches since the last read into `out` and clears the log.  Process-wide; off by default (one flag load per launch). */
int mggan_launch_log(int on);
int mggan_launch_log_read(char* out, int cap);

/* ---- dense layers: nn.Linear (+activation) forward / input grad / weight grad -------
 * reference: utils.py:134-149 (make_mlp), discriminators.py:46-56,76-108,
 *            standard.py:91-105, social.py:13,39-45, cnn.py:102-107
 * act: 0 none, 1 LeakyReLU(slope) (slope 0 == ReLU), 2 sigmoid, 3 sigmoid*(1-2e-7)+1e-7 (D output) */
int mggan_linear_fwd(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int rows, int K,
                     int N, int act, float slope, mggan_stream_t stream);
/* dZ = dY * act'(Y)  (Y = activation OUTPUT) */
/* A chain of up to three dense stages on 32-row tiles in ONE launch (replaces the nn.Sequential MLP stacks
 * built by the reference's utils.make_mlp, utils.py:134-149: discriminator heads, pred_encoder,
 * in_encoder_fc, PM-network).  `args` points to a host structure
 *   { const float* X; const float* in_mul; float* in_store;
 *     int ldx, rows, K0, ld_in_mul, in_mul_act, ld_in_store, n; float in_mul_slope;
 *     struct { const float* W; const float* bias; const float* mul_src; float* out;
 *              int K, N, ldw, trans, act, mul_act, ld_mul, ld_out, accumulate; float slope, mul_slope; } s[3]; }
 * Stage i: out_i[r][c] = act_i(sum_k in_i[r][k] * (trans ? W[k][c] : W[c][k]) + bias[c]) * act'_mul(mul_src[r][c]);
 * in_0 = X (* act'(in_mul), optionally copied to in_store), in_{i+1} = out_i; widths <= 192.  Forward pass:
 * trans = 0, `out` of the inner stages = the saved hidden activations.  Backward pass: in_mul = the saved
 * output of the last activation, stages run last-to-first with trans = 1 and mul_src = the saved hidden
 * activations; in_store / out hold the gate gradients the weight-gradient GEMMs (mggan_wgrad*) read. */
int mggan_mlp_chain(const void* args, mggan_stream_t stream);
int mggan_act_bwd(const float* dY, int lddy, const float* Y, int ldy, float* dZ, int lddz, int rows, int N, int act,
                  float slope, mggan_stream_t stream);
/* dX (rows x K) (+)= dZ (rows x N) . W (N x K, row stride ldw).  If Yact != NULL the activation derivative is
 * fused into the operand load: dZ = dY * act'(Yact) with dY passed as `dZ` (no separate mggan_act_bwd launch). */
int mggan_linear_bwd_data(const float* dZ, int lddz, const float* W, int ldw, float* dX, int lddx, int rows, int K,
                          int N, int accumulate, const float* Yact, int ld_yact, int act, float slope,
                          mggan_stream_t stream);
/* dW[g] (N x K, row stride lddw) += dZ_g^T X_g ; db[g] (N) += colsum(dZ_g).  Rows may be
 * split in n_groups contiguous segments seg[0..n_groups] (device int32, multiplied by
 * seg_scale) with per-group outputs w_stride / b_stride floats apart (per-generator
 * decoder weights).  seg == NULL, n_groups <= 1: one group over all rows. */
int mggan_wgrad_splits(int rows, int K, int N, int n_groups);
size_t mggan_wgrad_workspace_bytes(int rows, int K, int N, int n_groups);
/* feature_major != 0: dZ is stored [N][lddz] and X [K][ldx] (element (row, f) at p[f*ld + row]) */
int mggan_wgrad(const float* dZ, int lddz, const float* X, int ldx, float* dW, int lddw, float* db, int rows, int K,
                int N, const int* seg, int seg_scale, int n_groups, long w_stride, long b_stride, int feature_major,
                const float* Yact, int ld_yact, int act, float slope, void* workspace, size_t workspace_bytes,
                mggan_stream_t stream);
/* Many independent weight gradients (partial-sum phase of mggan_wgrad with dW == NULL) in one launch per
 * operand layout.  descs: array of n structures
 *   { const float* dZ; const float* X; float* workspace; const int* seg;
 *     int rows, K, N, lddz, ldx, seg_scale, n_groups, feature_major; }
 * with the meaning of the mggan_wgrad arguments of the same names; each workspace must hold
 * mggan_wgrad_workspace_bytes(rows, K, N, n_groups).  Fold the partials with mggan_grad_reduce_multi. */
int mggan_wgrad_multi(const void* descs, int n, mggan_stream_t stream);
/* Deferred reduction: mggan_wgrad with dW == NULL only writes its partial sums into `workspace`
 * ([groups*splits][N*(K+1)]); mggan_grad_reduce_multi then folds MANY such partial buffers into the
 * gradient buffers in one launch.  descs = host array of n structs
 *   { const float* P; float* dW; float* db; long w_stride, b_stride;
 *     int M, Naug, has_bias, lddw, splits, groups, p_stride, block0(ignored); }
 * has_bias bit 0: the last partial column is the bias gradient (-> db); bit 1: store (=) instead of
 * accumulate (+=), for scratch destinations that were not zeroed. */
int mggan_grad_reduce_multi(const void* descs, int n, mggan_stream_t stream);
int mggan_transpose(const float* W, float* WT, int N, int K, mggan_stream_t stream);
/* dst[ped][c] (+)= sum_k src[inv[k*b+ped]][c] : adjoint of "repeat over samples" */
/* The per-pedestrian tail of the rollout adjoint (shared h0, reference: the enc_h_to_dec_h adjoint of
 * /root/reference/mggan/model/modules/standard.py:118-131 summed over a pedestrian's K rollout rows) in one launch:
 * dQe (b,32) = sum over the K rows of a pedestrian of dH0; dEnc (b, ld_enc) = dQe . W_e2d[:, :EIN] (W_e2d row-major, row
 * stride ldw); the last S columns of dEnc additionally receive the K-row sums of dSocR (R,S) (S = 0: none; S <= 32). */
int mggan_rollout_ped_adjoint(const float* dH0, const float* dSocR, const int* inv, const float* W_e2d, int ldw, float* dQe,
                              float* dEnc, int ld_enc, int b, int K, int EIN, int S, mggan_stream_t stream);
int mggan_gather_sum(const float* src, int ld_src, const int* inv, float* dst, int ld_dst, int b, int K, int ncols,
                     int accumulate, mggan_stream_t stream);

/* ---- LSTM trajectory encoder / decoder rollouts ---------------------------------------
 * reference: common_modules.py:48-66 (TrajectoryEncoder), :97-131 (RelativeDecoder),
 *            standard.py:91-94,227-265 (enc_h_to_dec_h, forward_all), :190-214 (gather)
 * mggan_lstm_fold: per group (generator), fold Linear(2,E) into the gate weights and lay
 * the weights out for the rollout kernels.  Block layout (floats):
 *   A[4H][2] | bias[4H] | WhhT[H][4H] | dec: W1T[H+S][H/2] | b1[H/2] | W2[2][H/2] | b2[2] */
int mggan_lstm_prep_size(int H, int S, int dec);
int mggan_lstm_fold(const float* W_emb, const float* b_emb, const float* W_ih, const float* b_ih, const float* b_hh,
                    const float* W_hh, const float* W1, const float* b1, const float* W2, const float* b2,
                    long param_stride, int n_groups, int H, int E, int S, int dec, float* prep, int prep_stride,
                    mggan_stream_t stream);
/* chain rule back from (dA[4H][2] | dbias[4H]) to embedding / W_ih / b_ih / b_hh grads */
int mggan_lstm_unfold_grads(const float* W_emb, const float* b_emb, const float* W_ih, float* dW_emb, float* db_emb,
                            float* dW_ih, float* db_ih, float* db_hh, long param_stride, int n_groups, int H, int E,
                            const float* dprep, int dprep_stride, mggan_stream_t stream);
/* x (T,b,2) -> h_T (b, ld_hout).  Save buffers (all or none): Gt (b,T,4H) gates after
 * activation, Cs (b,T,H), Hp (b,T,H) = h_{t-1}, Din (b,T,2). */
int mggan_lstm_encoder_fwd(const float* x, int T, int b, int H, const float* prep, float* hout, int ld_hout, float* Gt,
                           float* Cs, float* Hp, float* Din, mggan_stream_t stream);
/* dh_T -> dPre (b,T,4H): gradient of the gate pre-activations (weight grads follow via mggan_wgrad) */
int mggan_lstm_encoder_bwd(const float* dhT, int ld_dhT, int T, int b, int H, const float* W_hh, const float* prep,
                           const float* Gt, const float* Cs, float* dPre, mggan_stream_t stream);
/* R rollout rows sorted by generator (generator g owns rows seg[g] .. seg[g+1]-1); row r: pedestrian
 * row_ped[r], noise slot row_slot[r], output position row_pos[r] in (T, Rout, 2).  One workgroup rolls out a
 * 16-row tile on the matrix cores (common_modules.py:97-131, standard.py:227-265).  Save buffers (all or none; private
 * to this pair of entries, tile-blocked with tiles = ceil(R/16) + n_gens slots of 16 rows):
 * Gt (tiles,T,H,16,4), Cs (tiles,T+1,H,16,2), Din (tiles,T,16,2), Aact (tiles,T,4,16,4); row-major E2Din (R,EIN+Z),
 * SocR (R,S) feed the weight-gradient GEMMs. */
int mggan_decoder_rollout_fwd(int R, int T, int b, int H, int EIN, int Z, const float* prep, int prep_stride,
                              const int* seg, int n_gens, const int* row_ped, const int* row_slot, const int* row_pos,
                              const float* enc_h, int ld_enc, const float* noise, const float* soc, int ld_soc,
                              const float* xy0, const float* dxdy0, const float* We2d, const float* be2d,
                              float* out_abs, float* out_rel, int Rout, float* Gt, float* Cs, float* Din,
                              float* Aact, float* E2Din, float* SocR, const float* Qe, float* Nz, mggan_stream_t stream);
/* Qe (b, 32) = be2d + We2d[:, :EIN] enc_h (standard.py:247-252), the part of h0 the K rollout rows of a pedestrian share
 * (W_e2d is one module for all generators): mggan_decoder_rollout_fwd then multiplies the noise columns only and keeps
 * Nz (R, Z), the rows' noise vectors, instead of E2Din (R, EIN+Z).  We2d: (32, ldw) row-major, EIN a multiple of 16. */
int mggan_decoder_e2d_shared(const float* enc_h, int ld_enc, int b, int EIN, const float* We2d, int ldw, const float* be2d,
                             float* Q, mggan_stream_t stream);
/* ---- social attention over in-scene ordered pairs ---------------------------------------
 * reference: social.py:67-104 (features), :33-48 (embedding MLP), :14-30 (attention pooling),
 *            discriminators.py:179-184 (D-side call; only sample block 0 carries features)
 * pair p = (pair_i[p], pair_j[p]); per pedestrian: ped_prow = index of pair (i, first j of its
 * scene), ped_s0 = first pedestrian of its scene, ped_n = scene size.
 * vc (b x 65) = Wh [W3 | b3] from the generic GEMM; sigma_ij = l2_ij . vc_j[:64] + vc_j[64]. */
int mggan_social_w3b(const float* W3, const float* b3, float* W3b, int F, mggan_stream_t stream);
int mggan_social_pairs_fwd(int P, const int* pair_i, const int* pair_j, const float* xy_last, const float* dxdy_last,
                           const float* W1, const float* b1, const float* W2, const float* b2, const float* vc,
                           float* feat, float* l1, float* l2, float* sigma, mggan_stream_t stream);
int mggan_social_softmax_fwd(int b, int H, const int* ped_prow, const int* ped_s0, const int* ped_n,
                             const float* sigma, const float* h, int ld_h, float* att, float* S, int ld_s,
                             mggan_stream_t stream);
int mggan_social_softmax_bwd(int b, int H, const int* ped_prow, const int* ped_s0, const int* ped_n, const float* att,
                             const float* h, int ld_h, const float* dS, int ld_ds, float* dsigma, float* dh, int ld_dh,
                             int accumulate_dh, mggan_stream_t stream);
int mggan_social_pairs_bwd(int P, int b, const int* pair_j, const int* ped_prow, const int* ped_s0, const int* ped_n,
                           const float* dsigma, const float* vc, const float* l1, const float* l2, const float* W2,
                           float* dz2, float* dz1, float* dvc, mggan_stream_t stream);

/* The same math -- INCLUDING the per-pedestrian dense stages Wh_j = W_at h_j + b_at, [v_j | c_j] = Wh_j [W3 | b3] and
 * their adjoints -- as ONE launch per direction for scenes of up to 64 pedestrians (csrc/social_rows.hip): a workgroup
 * owns whole scenes (scenes[s] = {first row, past-last row}), a wave owns attention rows, the pair MLP and its adjoint run
 * on MFMA, the softmax over a scene is wave-level shuffle reductions, and NOTHING per pair is stored: _bwd recomputes the
 * pair MLP from the positions.  H = width of h (32 | 64), F = social feature width (rows of W3 / W_at, <= 64), max_n =
 * largest scene (<= 64); W2, W3 and W_at must be 16-byte aligned (they are staged with 16-byte loads).  xy_mod > 0: the pedestrian rows repeat with that period (xy_last / dxdy_last hold one period:
 * the real and the fake half of a discriminator pair pass share the observed positions).
 * _bwd: dh[j] (+)= sum_i a_ij dS_i + dWh_j W_at; side outputs for the weight-gradient GEMMs of W3 | b3 | W_at | b_at:
 * Wh (rows, F), dWh (rows, F), dvc (rows, ldv) = d[v_j | c_j] (65 columns used); partials != NULL: also the weight
 * gradients of the 3->32 and 32->64 layers, as mggan_social_rows_grid(S, max_n) partial blocks of
 * mggan_social_rows_partial_floats() floats ([64][33] = dW2 | db2, then [32][4] = dW1 | db1) for mggan_grad_reduce_multi.
 * Rows that belong to no scene are not written.  With few scenes the rows of a scene are dealt to
 * mggan_social_rows_splits(S, max_n) workgroups; _bwd then needs `scratch` (splits x dvc_rows x (65 + H) floats) and
 * `tickets` (S words, zero before the first launch; the kernel leaves them at zero): the last workgroup of a scene to
 * arrive folds the shares in split order (results do not depend on the arrival order). */
int mggan_social_rows_splits(int S, int max_n);
int mggan_social_rows_grid(int S, int max_n);
int mggan_social_rows_partial_floats(void);
int mggan_social_rows_fwd(int S, const int* scenes, int H, int F, int max_n, const float* xy_last, const float* dxdy_last,
                          int xy_mod, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                          const float* b3, const float* Wat, const float* bat, const float* h, int ld_h, float* S_out,
                          int ld_s, mggan_stream_t stream);
int mggan_social_rows_bwd(int S, const int* scenes, int H, int F, int max_n, const float* xy_last, const float* dxdy_last,
                          int xy_mod, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                          const float* b3, const float* Wat, const float* bat, const float* h, int ld_h, const float* dS,
                          int ld_ds, float* dvc, int ldv, int dvc_rows, float* Wh, float* dWh, float* dh, int ld_dh,
                          int accumulate_dh, float* partials, float* scratch, unsigned* tickets, mggan_stream_t stream);

/* ---- Social-GAN pooling (--pool_type sgan) ------------------------------------------------
 * reference: social_gan.py:199-229 (PoolHiddenNet.forward): per scene, every pedestrian i and every j of its scene
 * (itself included): X = [Linear(2,E)(p_j - p_i) | h_j] -> mlp_pre_pool (mggan_mlp_chain) -> max over j.
 * pair p: output row pair_o[p], position rows pair_i[p] / pair_j[p] (modulo xy_mod when > 0), hidden row pair_j[p];
 * the pairs of output row o are [ped_prow[o], ped_prow[o] + ped_n[o]); hid_ptr / hid_pairs = CSR list of the pairs
 * that read each hidden row (for the deterministic adjoint of the gather). */
int mggan_pool_pairs_fwd(int P, const int* pair_i, const int* pair_j, const float* xy_last, int xy_mod,
                         const float* We, const float* be, int E, const float* h, int ld_h, int H, float* X, float* rel,
                         mggan_stream_t stream);
int mggan_pool_gather_bwd(int b, int H, int E, const int* hid_ptr, const int* hid_pairs, const float* dX, float* dh,
                          int ld_dh, mggan_stream_t stream);
int mggan_segment_max_fwd(int rows, int B, const int* ped_prow, const int* ped_n, const float* Y, float* out, int* arg,
                          mggan_stream_t stream);
int mggan_segment_max_bwd(int P, int B, const int* pair_o, const int* ped_prow, const float* dOut, int ld_dout,
                          const int* arg, float* dY, mggan_stream_t stream);

/* ---- scene CNN + physical attention -----------------------------------------------------
 * reference: cnn.py:119-160 (Conv_Blocks), :275-282 (CNN.forward), :109-116 (AttentionGlobal.forward)
 * img (B,4,33,33) -> conv1 (never stored at full resolution: ReLU(BN(.)) is monotone, so the pooled activation is
 * ReLU(BN(x_max)) for a positive BatchNorm scale and ReLU(BN(x_min)) for a negative one, and the scale has the sign of
 * the parameter gamma: the raw window extreme xsel (B,C,16,16) and its 2-bit position are kept) -> conv2 raw y2
 * (B,C,16,16) -> [BN+ReLU+pool] ->
 * attention over channels -> out (B,64).
 * Statistics: every kernel leaves ONE row of 2C doubles per workgroup in `part` ((sum, sumsq) forward, (sum g,
 * sum g*xhat) backward).  With a `ticket` (one zeroed word, left at zero) the last workgroup of the launch folds the rows
 * in index order and finalizes BatchNorm itself (scale / shift / stat = mean | invstd / `updates` momentum updates of
 * the running statistics; backward: coef = [gamma*invstd | mean g | mean g*xhat], dgamma / dbeta accumulated).  With
 * ticket == NULL the caller does it: mggan_bn_reduce_rows -> (all-reduce over ranks) -> mggan_bn_finalize /
 * mggan_bn_bwd_coef.
 * `dims` (may be NULL = every image is real): a 16-byte record in DEVICE memory {int n_real; int s_real; float
 * b_padded / n_real; pad} of a batch that the trainer padded to its shape bucket with inert "phantom" pedestrians at the
 * end (the reference loader's ragged batches, /root/reference/mggan/data_utils/trajectories_scene.py:40-78, replayed as one
 * captured graph per bucket): the image loops stop at n_real, the element counts of the statistics shrink with it, and
 * the attention head writes zero features for the phantom rows.  The launch geometry stays that of B. */
/* mggan_conv1_pool, gram != NULL (sharded training; ticket must be NULL): the BatchNorm-1 statistics are those of the batch
 * whose Gram matrix of image patches is `gram` -- the GLOBAL batch's, all-reduced once per iteration -- and are written by
 * the launch itself (see mggan_bn1_from_gram below); no exchange at this point. */
/* `comm` (mggan_conv1_pool, mggan_conv2_fwd2, mggan_scene_attention_bwd; may be NULL; needs ticket != NULL): sharded training
 * over peer-mapped arenas -- a device pointer from mggan_comm_channel_create for the calling stream's channel.  The last
 * workgroup to finish folds the partial rows, exchanges the 2C sums and `count` (THIS rank's element count) with the other
 * ranks and finalizes with the GLOBAL statistics: the BatchNorm exchange point costs no launch of its own (round 5: fold +
 * exchange + finalize were one extra launch, mggan_bn_sync_finalize).  A rank with B == 0 still takes part (one
 * single-workgroup launch).  Backward: coef from the global sums, dgamma / dbeta take this rank's share. */
int mggan_cnn_grid(int B);      /* workgroups (= partial rows) of conv1_pool / conv2_fwd2 / image_gram / conv1_wgrad */
int mggan_cnn_bwd_grid(int B);  /* workgroups (= partial rows) of conv2_bwd */
int mggan_conv1_pool(const float* img, int B, int C, const float* W, const float* bias, float* xsel,
                     unsigned char* code, double* part, unsigned int* ticket, double count, const float* gamma,
                     const float* beta, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                     float eps, int updates, float* scale, float* shift, float* stat, const double* gram,
                     const void* comm, const int* dims, mggan_stream_t stream);
int mggan_conv2_fwd2(const float* xsel, int B, int C, const float* scale1, const float* shift1,
                     const float* W, const float* bias, float* y2, double* part, unsigned int* ticket, double count,
                     const float* gamma, const float* beta, float* run_mean, float* run_var,
                     long long* num_batches_tracked, float momentum, float eps, int updates, float* scale, float* shift,
                     float* stat, const void* comm, const int* dims, mggan_stream_t stream);
int mggan_bn_reduce_rows(const double* part, int rows, int W, double* sums, mggan_stream_t stream);
/* training: 0 = eval (running statistics), n >= 1 = batch statistics and n momentum updates of the running ones */
int mggan_bn_finalize(const double* sums, double count, int C, int training, const float* gamma, const float* beta,
                      float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* stat, mggan_stream_t stream);
/* backward coefficients from partial rows (one rank) ... */
int mggan_bn_bwd_rows_finalize(const double* part, int rows, double count, int C, const float* gamma, const float* stat,
                               float* coef, double* coefd, float* dgamma, float* dbeta, mggan_stream_t stream);
/* ... or from sums over the GLOBAL batch (after the all-reduce); local_sums: this rank's share (dgamma / dbeta) */
int mggan_bn_bwd_coef(const double* sums, const double* local_sums, double count, int C, const float* gamma,
                      const float* stat, float* coef, double* coefd, float* dgamma, float* dbeta, mggan_stream_t stream);
int mggan_scene_attention_fwd(const float* y2, int B, int C, const float* scale2, const float* shift2, const float* Wa,
                              const float* ba, const float* Wb, const float* bb, float* out, int ld_out, float* ysel,
                              unsigned char* ycode, const int* dims, mggan_stream_t stream);
/* ysel / ycode above (both or neither; a forward pass that will be differentiated passes them): per image, channel and
 * pooled cell -- laid out (B, 64 cells, C) -- the raw conv2 value that won its 2x2 window and which of the four it was -- all the adjoint needs
 * of the conv2 output.
 * Adjoint of the attention head INCLUDING the weight gradients of both layers (MFMA; nothing per position is stored):
 * g2sel (B,64,C) = gradient that reaches the raw conv2 output grid, ONE value per pooled cell (it sits on window position
 * ycode of the (B,C,16,16) grid, the other three are zero: mggan_conv2_bwd spreads it); wpart: mggan_scene_attention_grid(B) partial blocks of
 * mggan_scene_attention_partial_floats(C) floats ([32][C+1] = dWa | dba, then [C][33] = dWb | dbb) for
 * mggan_grad_reduce_multi; part: the same number of rows of 2C doubles (BatchNorm-2 adjoint sums per workgroup);
 * ticket != NULL: the launch also finishes the BatchNorm-2 adjoint (coef2, dgamma2 / dbeta2 accumulated). */
int mggan_scene_attention_grid(int B);
int mggan_scene_attention_partial_floats(int C);
int mggan_scene_attention_bwd(const float* ysel, const unsigned char* ycode, int B, int C, const float* scale2,
                              const float* shift2, const float* stat2, const float* Wa, const float* ba, const float* Wb,
                              const float* bb, const float* dout, int ld_dout, float* g2sel, float* wpart, double* part,
                              unsigned int* ticket, double count, const float* gamma2, float* coef2, float* dgamma2,
                              float* dbeta2, const void* comm, const int* dims, mggan_stream_t stream);
/* conv2 adjoint (BatchNorm-2 backward on the fly, dW2 / db2 as per-workgroup partial rows in `workspace`, input
 * gradient routed through ReLU / max-pool of block 1 -> G1c (B,C,16,16)); part1:
 * mggan_cnn_bwd_grid(B) rows; coefd1 = [gamma*invstd | S1 | S2 | mean | invstd] (C each) + count, f64, for
 * mggan_conv1_wgrad.  workspace: mggan_cnn_bwd_grid(B) * (256/(C*C)) * (C*C*9 + C) floats. */
int mggan_conv2_bwd(const float* xsel, int B, int C, const float* scale1,
                    const float* shift1, const float* stat1, const float* y2, const float* g2sel,
                    const unsigned char* ycode, const float* stat2,
                    const float* coef2, const float* W, float* G1c, double* part1, float* dW,
                    float* db, float* workspace, size_t workspace_bytes, unsigned int* ticket, double count1,
                    const float* gamma1, float* coef1, double* coefd1, float* dgamma1, float* dbeta1,
                    const int* dims, mggan_stream_t stream);
/* Gram matrix of the 3x3 patches of a batch of images, gram[s][t] (37 x 37 doubles; tap t = 9*ci + 3*ky + kx,
 * tap 36 = the constant 1): the image-only part of every conv1 weight gradient of the batch (both CNNs, every
 * backward pass).  Computed from the images' autocorrelation (10 channel pairs x 25 offsets, edge rows / columns
 * taken out at assembly; MGGAN_GRAM_KERNEL=mfma: tap by tap on the matrix cores).  workspace:
 * mggan_image_gram_workspace(B) bytes. */
size_t mggan_image_gram_workspace(int B);
int mggan_image_gram(const float* img, int B, double* gram, double* workspace, size_t workspace_bytes,
                     const int* dims, mggan_stream_t stream);
/* conv1 weight gradient (the images need no input gradient): dW (C,4,3,3) += (gamma/sigma) * (A - mean(g) * B -
 * mean(g*xhat) * Chat) with A from G1c / code1 (matrix cores) and B, Chat from the Gram matrix, in f64; the conv1 bias
 * gradient is identically zero in front of a train-mode BatchNorm.  workspace: mggan_cnn_grid(B) * C * 36 doubles. */
int mggan_conv1_wgrad(const float* img, int B, int C, const float* G1c, const unsigned char* code1, const double* gram,
                      const float* W, const float* bias, const double* coefd, float* dW, double* workspace,
                      size_t workspace_bytes, unsigned int* ticket /* two zeroed words: the finalize rides in the launch */,
                      const int* dims, mggan_stream_t stream);
/* Sharded training (scene sharding over the GPUs of a node), layer 1 of the scene CNNs without an exchange of
 * its own: with the batch's GLOBAL Gram matrix (mggan_image_gram, all-reduced once per batch) the BatchNorm-1 FORWARD
 * statistics of any conv1 weights follow without a collective (mggan_bn1_from_gram: scale / shift / stat and the
 * running-statistics update, as mggan_bn_finalize), and what the layer-1 ADJOINT needs of the other ranks is folded into an
 * f64 tail [A (C x 36) | S1 (C) | S2 (C)] (mggan_conv1_tail_fold; wrows = the partial rows mggan_conv1_wgrad leaves when
 * dW == NULL, part1 = those of mggan_conv2_bwd) that travels with the step's gradient all-reduce (mggan_comm_allreduce2);
 * mggan_conv1_tail_finalize then adds dW1 / dgamma1 / dbeta1 of the GLOBAL batch, identically on every rank. */
int mggan_bn1_from_gram(const double* gram, int C, const float* W, const float* bias, const float* gamma,
                        const float* beta, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                        float eps, int updates, float* scale, float* shift, float* stat, mggan_stream_t stream);
int mggan_conv1_tail_floats(int C);
int mggan_conv1_tail_fold(const double* wrows, int rows, const double* part1, int rows1, int C, double* tail,
                          int riders /* doubles behind the sums: copied from rider_src, or set to zero (NULL) */,
                          const double* rider_src, mggan_stream_t stream);
int mggan_conv1_tail_finalize(const double* tail, const double* gram, int C, const float* W, const float* bias,
                              const float* gamma, const float* stat, float* dW, float* dgamma, float* dbeta,
                              mggan_stream_t stream);

/* ---- both discriminator heads over many rows, weight-stationary (csrc/dheads.hip) -------------------------------
 * reference: discriminators.py:76-85,197-204 (discs[0]) and :97-108,211-219 (gen_id_reconstructor) on the K*b rows of
 * the generator step.  X (rows, ldx >= 192) -> Ya (rows,1) = act_a(Linear(96,1)(LeakyReLU(Linear(192,96)(X)))),
 * Yb (rows,g) = Linear(96,g)(LeakyReLU(Linear(192,96)(X))); Ha / Hb (rows,96) keep the hidden activations when not NULL.
 * mggan_dheads_bwd_data: dX (rows, ld_dx) from dYa, dYb -- the input gradient only (frozen discriminator). */
int mggan_dheads_fwd(const float* X, int ldx, int rows, int g, int act_a, const float* W1a, const float* b1a,
                     const float* W2a, const float* b2a, const float* W1b, const float* b1b, const float* W2b,
                     const float* b2b, float* Ha, float* Hb, float* Ya, float* Yb, mggan_stream_t stream);
int mggan_dheads_bwd_data(const float* dYa, const float* dYb, const float* Ya, const float* Ha, const float* Hb, int rows,
                          int g, int act_a, const float* W1a, const float* W2a, const float* W1b, const float* W2b,
                          float* dX, int ld_dx, mggan_stream_t stream);
/* the same adjoint for TRAINABLE heads over a real/fake pair pass (discriminator step): head B covers rows
 * [row0_b, rows) only (Hb and dYb start at that row); besides dX it leaves dH (rows,192) = the first layers' gate
 * gradients [head A | head B] and dza (rows) = dYa * act'(Ya): the operands of the four weight-gradient products */
int mggan_dheads_bwd_train(const float* dYa, const float* dYb, const float* Ya, const float* Ha, const float* Hb, int rows,
                           int row0_b, int g, int act_a, const float* W1a, const float* W2a, const float* W1b,
                           const float* W2b, float* dX, int ld_dx, float* dH, float* dza, mggan_stream_t stream);

/* Lean form of the same heads for the sample blocks k >= 1 of a K-sample pass with a frozen discriminator (the generator
 * step; discriminators.py:179-219): in_enc and scene are shared by a pedestrian's K rows and the social block is zero
 * from block 1 on, so P[ped] = b1 + W1[:, in_enc | scene] [in_enc | scene](ped) is computed once per pedestrian
 * (mggan_dheads_shared, from the block-0 rows of X) and the row product keeps K = 32 (the pred_enc block at c_pe).
 * mask: ceil((rows-row0)/16)*64 64-bit words of LeakyReLU sign bits, written by _fwd (NULL: not kept), read by _bwd,
 * which writes dX[row][c_pe .. c_pe+31] only. */
int mggan_dheads_shared(const float* X, int ldx, int b, int c_in, int c_sc, const float* W1a, const float* b1a,
                        const float* W1b, const float* b1b, float* P, mggan_stream_t stream);
int mggan_dheads_lean_fwd(const float* X, int ldx, int c_pe, int row0, int rows, int b, int g, int act_a, const float* P,
                          const float* W1a, const float* W2a, const float* b2a, const float* W1b, const float* W2b,
                          const float* b2b, unsigned long long* mask, float* Ya, float* Yb, mggan_stream_t stream);
int mggan_dheads_lean_bwd(const float* dYa, const float* dYb, const float* Ya, const unsigned long long* mask, int row0,
                          int rows, int g, int act_a, const float* W1a, const float* W2a, const float* W1b, const float* W2b,
                          int c_pe, float* dX, int ld_dx, mggan_stream_t stream);

/* The lean pass one stage further back (csrc/dheads.hip): rows [row0, rows) from the predicted steps pred (T = 12, rows, 2;
 * row = k*b + ped) through the frozen pred_encoder (discriminators.py:42-43,129-131: Wp1 (64,24), bp1, Wp2 (32,64), bp2)
 * and both heads in ONE launch, and from dYa / dYb to dpred (12, rows, 2) in one: the chain of transposed products keeps
 * the activations in registers (no row copy of the steps, no pred_enc block, no hidden activations in memory). */
int mggan_d_rows_lean_fwd(const float* pred, int T, int row0, int rows, int b, int g, int act_a, const float* Wp1,
                          const float* bp1, const float* Wp2, const float* bp2, const float* P, int c_pe, const float* W1a,
                          const float* W2a, const float* b2a, const float* W1b, const float* W2b, const float* b2b,
                          unsigned long long* mask, float* Ya, float* Yb, mggan_stream_t stream);
int mggan_d_rows_lean_bwd(const float* dYa, const float* dYb, const float* Ya, const unsigned long long* mask, int T,
                          int row0, int rows, int g, int act_a, const float* Wp1, const float* Wp2, int c_pe,
                          const float* W1a, const float* W2a, const float* W1b, const float* W2b, float* dpred,
                          mggan_stream_t stream);
/* pred_encoder (discriminators.py:42-43,129-131: Linear(24,64) - LeakyReLU(0.2) - Linear(64,32)) from time-major steps
 * (T = 12, n_stride, 2) straight into its column block of the classifier input: rows [0, rows_a) from a, the rest from b2
 * (the real and the fake half of a pair pass); h1 (rows,64) / xrows (rows,24): hidden layer and row copy of the steps for
 * the weight gradients, when not NULL.  One launch instead of steps_to_rows + a chain launch. */
int mggan_pred_encoder_fwd(const float* a, const float* b2, int T, int n_stride, int rows_a, int rows, const float* Wp1,
                           const float* bp1, const float* Wp2, const float* bp2, float* X, int ldx, int c_pe, float* h1,
                           float* xrows, mggan_stream_t stream);
/* the first `rows` rows of steps (T, n, 2) as rows (rows, 2T); rows (n, ld) into the first n rows of steps (T, n_out, 2) */
int mggan_steps_to_rows_n(const float* a, int T, int n, int rows, float* out, mggan_stream_t stream);
int mggan_rows_to_steps_n(const float* rows, int ld, int T, int n, int n_out, float* out, mggan_stream_t stream);

/* ---- in-graph all-reduce over peer-mapped memory (csrc/comm.hip; scene-sharded training) ------------
 * No reference counterpart (the reference is single-process); replaces torch.distributed.all_reduce for the <= 360 KB
 * messages of an iteration with a plain, HIP-graph-capturable kernel.  Every rank owns one uncached arena per channel
 * (mggan_comm_arena_bytes(max_elems), max_elems counted in 8-byte elements), exports it with mggan_comm_ipc_handle
 * (64-byte handle) and maps its peers' with mggan_comm_ipc_open.  mggan_comm_allreduce sums `n` elements of `data`
 * (dtype 0 f32, 1 f64, 2 i32) over the ranks IN PLACE, adding the ranks' contributions in rank order (bit-identical on
 * every rank).  All ranks must issue the same collectives in the same order on a channel.  A wait that exceeds the bound
 * (mggan_comm_set_timeout, 30 s by default) sets the arena's error word (mggan_comm_error) and the process's host-mapped
 * error word (mggan_comm_host_error: a host pointer, readable without a device sync) instead of hanging, and the
 * collective leaves NaN (INT_MIN for i32) in `data` -- never a sum over stale slots. */
size_t mggan_comm_arena_bytes(long max_elems);
int mggan_comm_alloc(size_t bytes, void** out);
int mggan_comm_free(void* p);
int mggan_comm_ipc_handle(void* p, void* handle);
int mggan_comm_ipc_open(const void* handle, void** out);
int mggan_comm_ipc_close(void* p);
int mggan_comm_allreduce(void* const* arenas, int rank, int world, long max_elems, void* data, long n, int dtype,
                         mggan_stream_t stream);
/* ... with an f64 tail (data2, n2 doubles) summed in the same collective (same flags, same sequence number) */
int mggan_comm_allreduce2(void* const* arenas, int rank, int world, long max_elems, void* data, long n, int dtype,
                          double* data2, long n2, mggan_stream_t stream);
/* A channel's arguments in device memory (*out), for the scene-CNN kernels that fold their BatchNorm exchange into their
 * last workgroup (the `comm` argument of mggan_conv1_pool, mggan_conv2_fwd2, mggan_scene_attention_bwd). */
int mggan_comm_channel_create(void* const* arenas, int rank, int world, long max_elems, void** out);
int mggan_comm_channel_free(void* p);
int mggan_comm_error(const void* arena, unsigned int* out);
int mggan_comm_set_timeout(double seconds);
int mggan_comm_host_error(unsigned int** out);
/* Sharded scene CNN: one launch per BatchNorm exchange point -- fold this rank's partial rows, all-reduce the 2C sums
 * and the element count over the ranks (peer-mapped arenas of the calling stream's channel), finalize with the global
 * statistics (forward: scale / shift / stat / running statistics; backward: coef / coefd, dgamma / dbeta += this rank's
 * share).  Same results on every rank. */
int mggan_bn_sync_finalize(void* const* arenas, int rank, int world, long max_elems, const double* part, int rows,
                           double local_count, int C, const float* gamma, const float* beta, float* run_mean,
                           float* run_var, long long* num_batches_tracked, float momentum, float eps, int updates,
                           float* scale, float* shift, float* stat, mggan_stream_t stream);
int mggan_bn_bwd_sync_finalize(void* const* arenas, int rank, int world, long max_elems, const double* part, int rows,
                               double local_count, int C, const float* gamma, const float* stat, float* coef,
                               double* coefd, float* dgamma, float* dbeta, mggan_stream_t stream);

/* ---- RCCL all-reduce inside the iteration graph (csrc/rccl.hip; scene-sharded training) --------------
 * No reference counterpart.  The north-star transport ("RCCL all-reduce of discriminator/PM gradients over xGMI") as a
 * launch of THIS library: ncclAllReduce bound from librccl.so (dlopen; the copy torch has mapped) on the caller's
 * stream, so it is captured into the one iteration graph like every other entry -- torch.distributed's collectives cut
 * the capture into segments.  The communicator is the library's own: rank 0 draws a 128-byte id
 * (mggan_rccl_unique_id), the host side hands it to every rank (mggan/devcomm.py: RcclComm), every rank calls
 * mggan_rccl_comm_init on its device.  mggan_rccl_allreduce sums `data` (n elements; dtype 0 f32, 1 f64, 2 i32) in
 * place and, in the same RCCL group (one launch), an optional f64 tail `data2` (n2 doubles) -- the RCCL form of
 * mggan_comm_allreduce2.  mggan_rccl_available: 1 when librccl.so resolves, else 0 (the reason: mggan_last_error after
 * a failed call). */
int mggan_rccl_available(void);
int mggan_rccl_unique_id(void* id);
int mggan_rccl_comm_init(const void* id, int rank, int world, void** comm);
int mggan_rccl_comm_destroy(void* comm);
int mggan_rccl_allreduce(void* comm, void* data, long n, int dtype, double* data2, long n2, mggan_stream_t stream);
int mggan_rccl_async_error(void* comm, int* out);

/* ---- losses (+ gradients), clipping, AdamW ----------------------------------------------
 * reference: abstract_train.py:62-67, utils.py:18-25, train.py:58-75,92-113,181-200,626-639,
 *            train.py:131-135,209-213,656-658, abstract_train.py:45-50 */
/* p = D output rows; loss_r = w_r*scale*L(p_r,label), w_r = inv_count[row_gen[r]] (or 1); dp = dloss/dp.
 * kind 0: L = BCELoss (objectives 'NS' and, with a negative scale on the generator side, 'MM'); kind 1: L = (p-label)^2
 * ('LS'); abstract_train.py:62-75 */
/* label: host scalar, or (label_u != NULL) drawn on the device as label_lo + (label_hi-label_lo) * *label_u */
int mggan_bce_rows(int rows, int kind, const float* p, float label, const float* label_u, float label_lo,
                   float label_hi, float scale, const int* row_gen, const float* inv_count, float* loss_rows, float* dp,
                   mggan_stream_t stream);
/* All adversarial loss terms of one optimizer step in ONE launch (abstract_train.py:62-75 phi_1/phi_2/phi_3,
 * train.py:92-111 generator re-weighting and classifier CE, :184 discriminator-side CE): term A = BCE/MSE over rows
 * [0,nA) of p (optionally weighted 1/count(generator of the row), sign +-1), term B = BCE/MSE over rows [nA,nA+nB),
 * term C = cross entropy over nC rows of g logits.  Each term's mean goes to out[q] (when not NULL), total =
 * A + B + grad_c*C, and the gradients of `total` wrt p and the logits are left in dp / dlogits.  `args` points to
 *   { const float* p; const float* label_u[2]; const int* row_gen; const int* seg; const float* inv_count;
 *     const float* logits; const int* target; float* dp; float* dlogits; float* out[3]; float* total;
 *     double* partial; unsigned* ticket;   (768 doubles of scratch; one word that is zero before the first call)
 *     const int* dims;                     (padded batch, see the scene-CNN section; NULL: every row is real)
 *     float label[2], lo[2], hi[2], scale[3], sign_a, grad_c; int nA, nB, nC, g, ld, kind, weighted_c, bmod; }
 * With dims, row r of every term belongs to pedestrian r % bmod: phantom rows get zero gradients and take no part in the
 * sums, and the scales (reciprocals of the PADDED row counts) are corrected by dims' factor.
 * label_u[q] != NULL: the smoothed label of term q is lo + (hi-lo)*u drawn on the device; counts per generator come
 * from seg (g+1 offsets of the generator-sorted rows) or from inv_count. */
int mggan_gan_losses(const void* args, mggan_stream_t stream);
/* PM-network targets 'l2' (mode 0) / 'endpoint' (mode 1), train.py:616-624,641-647: target[ped] = the generator
 * whose best of E samples is closest to the ground truth; gen_abs (T,E,g,b,2), gt (T,b,2) */
int mggan_pm_target(int b, int T, int E, int g, int mode, const float* gen_abs, const float* gt, int* target,
                    mggan_stream_t stream);
/* PM-network target 'mgan' as the reference computes it (train.py:606-614; the target softmax runs over a singleton
 * axis): loss_r = scale * (-target_weight * sum_j log p_j + reg * sum_j p_j log p_j), p = softmax(logits (b,g)) */
/* reg_dev != NULL: `reg` (the reference's 0.9 ** epoch) is read from that device word instead -- a captured iteration then
 * follows the epochs without being captured again */
int mggan_pm_mgan_loss(int b, int g, const float* logits, float target_weight, float reg, const float* reg_dev, float scale,
                       float* loss_rows, float* dlogits, float* probs, mggan_stream_t stream);
/* Categorical(logits=...).sample((K,)).T on the device (standard.py:217-225): inverse CDF from uniforms u (b,K) */
/* ---- device RNG: every draw of one training iteration in ONE launch (csrc/rng.hip, Philox4x32-10) ---------
 * reference draws being replaced on the `--rng device` path: utils.py:18-25 (label uniforms),
 * utils.py:152-165 (one N(0,1)^Z vector per scene, repeated for its pedestrians), standard.py:217-225 (Categorical
 * sampling, here: the uniforms of the inverse-CDF sampler).  state[0] = seed, state[1] = iteration counter (advanced
 * by the kernel itself: a captured graph draws fresh numbers at every replay); ticket = one zeroed word.
 * noise[set][ped][z], set < n_sets, is identical for the pedestrians of a scene (ped_scene[ped]). */
int mggan_draw_iteration(long long* state, unsigned int* ticket, int n_labels, float* labels, int n_sets, int b, int Z,
                         const int* ped_scene, float* noise, long n_unif, float* unif, mggan_stream_t stream);
int mggan_sample_categorical(int b, int K, int g, const float* logits, const float* u, long long* idx,
                             mggan_stream_t stream);
/* device-side replacement of get_selection_indices + gather bookkeeping (utils.py:234-248,
 * standard.py:190-214): idx (b,K) int64 generator ids -> rollout rows stably sorted by generator.
 * row r: generator row_gen[r], pedestrian row_ped[r], noise slot row_slot[r] (occurrence offset),
 * output position row_pos[r] = k*b+ped; inv = inverse permutation; seg[g+1] = segment offsets;
 * row_gen_pos = generator id per output position; blk_cnt = scratch, 16 * ceil(b*K/1024) ints. */
int mggan_bucket_rows(const long long* idx, int b, int K, int g, int* row_gen, int* row_ped, int* row_slot,
                      int* row_pos, int* inv, int* seg, int* row_gen_pos, int* blk_cnt, mggan_stream_t stream);
/* both of the above behind one entry -- ONE launch up to 2,048 rows (the single-sample rollouts of the discriminator
 * step), the separate launches beyond: idx receives the picks, the row tables are those of mggan_bucket_rows;
 * ticket: reserved (one word, untouched) */
int mggan_sample_bucket_rows(int b, int K, int g, const float* logits, const float* u, long long* idx, int* row_gen,
                             int* row_ped, int* row_slot, int* row_pos, int* inv, int* seg, int* row_gen_pos, int* blk_cnt, int blk_self_reset,
                             unsigned int* ticket, mggan_stream_t stream);
int mggan_scale(float* x, long n, const float* scalar, mggan_stream_t stream);
/* classifier input of the discriminator (discriminators.py:141,185,196): rows k*b+ped =
 * [soc | in_enc | pred_enc | scene], and its adjoint.  soc_all = 0: social features exist for sample block 0 only
 * (soc0 has b rows; the list-repetition quirk of one K-sample call); soc_all = 1: soc0 has K*b rows
 * (K independent single-sample calls batched into one pass, e.g. the real and the fake pass of a D step);
 * soc_all = 2: soc0 has b rows and EVERY sample block gets them (PoolHiddenNet walks the K-times repeated scene list
 * and concatenates K copies of the block-0 result, social_gan.py:212-228). */
int mggan_d_assemble_fwd(int b, int K, int w_soc, int w_in, int w_pred, int w_scene, int soc_all, const float* soc0,
                         const float* in_enc, const float* pred_enc, const float* scene, float* X,
                         mggan_stream_t stream);
int mggan_d_assemble_bwd(int b, int K, int w_soc, int w_in, int w_pred, int w_scene, int soc_all, const float* dX,
                         float* dsoc0, float* din_enc, float* dpred_enc, float* dscene, mggan_stream_t stream);
/* The same classifier input without intermediate copies (csrc/drows.hip): X (K*b, ldx) is written in place by its
 * producers (pred_encoder chain -> columns of pred_enc, social attention -> columns of soc); d_rows_fill broadcasts
 * in_enc / scene of pedestrian `ped` into rows k*b+ped and clears the soc columns [0, w_soc) of sample blocks
 * k >= soc_blocks (blocks without social features); d_rows_reduce is the adjoint of the broadcast
 * (sum over k, fixed order; a NULL destination is skipped).  Widths / offsets / strides: multiples of 4 floats. */
int mggan_d_rows_fill(int b, int K, int soc_blocks, int w_soc, int c_in, int w_in, int c_scene, int w_scene,
                      const float* in_enc, int ld_in, const float* scene, int ld_scene, float* X, int ldx,
                      mggan_stream_t stream);
int mggan_d_rows_reduce(int b, int K, int c_in, int w_in, int c_scene, int w_scene, const float* dX, int ldx, float* din,
                        int ld_in, float* dscene, int ld_scene, mggan_stream_t stream);
/* adjoint of mggan_steps_to_rows: rows (n, 2T) with row stride ld -> time-major (T, n, 2) */
int mggan_rows_to_steps(const float* rows, int ld, int T, int n, float* out, mggan_stream_t stream);
/* Time-major steps -> one row per trajectory: out[r][2t+c] = a[t][r][c] for r < n, and (b != NULL) out[n+r][2t+c] =
 * b[t][r][c] -- the input of the discriminator's pred_encoder (discriminators.py:129-131 permute + reshape; the real and
 * the fake trajectories of a pair pass in one launch) */
int mggan_steps_to_rows(const float* a, const float* b, int T, int n, float* out, mggan_stream_t stream);
int mggan_ce_rows(int rows, int g, const float* logits, int ld, const int* target, const float* inv_count, float scale,
                  float* loss_rows, float* dlogits, int ldd, mggan_stream_t stream);
int mggan_l2_min_scene(int S, int T, int K, int b, const int* scenes, const int* ped_scene, const float* gen_abs,
                       const float* gt, float grad_scale, float* scene_loss, int* scene_arg, float* gabs,
                       const int* dims, mggan_stream_t stream);
int mggan_pm_ml_loss(int b, int T, int E, int g, const float* gen_abs, const float* gt, const float* logits, float sigma,
                     float scale, float* loss_rows, float* dlogits, float* probs, mggan_stream_t stream);
/* the same plus the reductions the trainer needs, in ONE launch: *out = sum of the loss rows, probs_out[g] =
 * probs_scale * column means of the generator probabilities; partial = 17 * 64 doubles of scratch, ticket = one word
 * that is zero before the first call (the kernel leaves it at zero) */
int mggan_pm_ml_loss_mean(int b, int T, int E, int g, const float* gen_abs, const float* gt, const float* logits,
                          float sigma, float scale, float* loss_rows, float* dlogits, float* probs, double* partial,
                          unsigned* ticket, float* out, float* probs_out, float probs_scale, const int* dims,
                          mggan_stream_t stream);
int mggan_sum(const float* x, long n, float alpha, float* out, int accumulate, mggan_stream_t stream);
int mggan_colmean(const float* x, int rows, int g, float scale, float* out, mggan_stream_t stream);
/* dims / bmod as in mggan_gan_losses: rows of phantom pedestrians (row % bmod >= n_real) are not counted */
int mggan_gen_counts(const int* idx, int n, int g, int* counts, float* inv_count, const int* dims, int bmod,
                     mggan_stream_t stream);
int mggan_inv_counts(const int* counts, int g, float* inv_count, mggan_stream_t stream);
/* Sharded training: the generator step's 1/count weights (train.py:94-96 of the reference: counts over the GLOBAL batch)
 * without a collective of their own -- mggan_sample_counts computes, ahead of the sampling launch, how often every generator
 * will be picked on these logits and uniforms (out: 16 doubles, the same CDF arithmetic as mggan_sample_categorical); the
 * doubles ride in the f64 tail of an earlier exchange (mggan_comm_allreduce2) and mggan_inv_counts_f64 turns the sums into
 * the weights. */
int mggan_sample_counts(int b, int K, int g, const float* logits, const float* u, int* scratch /* 17 ints, zero */,
                        double* out, mggan_stream_t stream);
int mggan_inv_counts_f64(const double* counts, int g, float* inv_count, mggan_stream_t stream);
/* flat parameter buffer + segment table: elem_seg[i] = segment of element i (-1 = padding),
 * active[s] = segment takes part in this step (grad not None), seg_step[s] = Adam step count.
 * zero_grad != 0: the consumed gradients are left at 0 instead of their clipped values (saves the caller's
 * memset before the next backward pass).  lr_dev != NULL: the learning rate is read from that device word instead of
 * `lr` (a captured HIP graph then follows the per-epoch schedule without being captured again).
 * ONE launch (gradient norm, clipping, update): workspace = 600 doubles that are ZERO before the first call (partial sums,
 * the two counters of the launch's grid barrier, the tail flag, the finalized tail block; every launch leaves them ready for
 * the next).  One workspace per
 * optimizer; launches that share one must be ordered by their stream. */
/* comm (may be NULL): sharded training over peer-mapped arenas -- a device pointer from mggan_comm_channel_create for the
 * calling stream's channel: the gradient all-reduce (sum over the ranks, rank order) runs INSIDE this launch, chunk by chunk
 * by the workgroup that updates the chunk (round 5: mggan_comm_allreduce2 in front of it); every rank must make the call.
 * tail (may be NULL): the f64 tail that rides with the gradients (mggan_conv1_tail_fold) and what its finalize needs; with
 * comm it is exchanged in the same launch, and its finalize (mggan_conv1_tail_finalize: global-batch dW1 / dgamma1 / dbeta1
 * added into their slots of `grad`) happens in this launch too -- with comm == NULL the tail must already hold the global
 * sums (e.g. mggan_rccl_allreduce in front of this call).  The chunk count of `n` floats + 1 must fit the arena's flags. */
typedef struct {
  double* tail;        /* [A (C x 36) | S1 (C) | S2 (C) | riders], n2 doubles */
  long n2;
  const double* gram;  /* 37 x 37 Gram matrix of the GLOBAL batch's image patches */
  const float* W;      /* conv1 weight (C,4,3,3) */
  const float* bias;
  const float* gamma;  /* BatchNorm-1 weight */
  const float* stat;   /* [mean | invstd] of the forward pass */
  float* dW;           /* slots inside `grad` */
  float* dgamma;
  float* dbeta;
  int C;
} mggan_grad_tail_t;
int mggan_clip_adamw(float* param, float* grad, float* m, float* v, long n, const int* elem_seg, int nseg,
                     const unsigned char* active, int* seg_step, float max_norm, double lr, const double* lr_dev,
                     double beta1, double beta2, double eps, double weight_decay, int zero_grad, double* workspace,
                     float* norm_out, const void* comm, const mggan_grad_tail_t* tail, mggan_stream_t stream);

/* Fused decoder backward: BPTT + in-kernel per-generator weight gradients (dW_hh and dW1[:, :H] on MFMA).
 * n_gens*NW persistent workgroups; workgroup (g, w) leaves one partial block of `wlen` floats at
 * wpart[(g*NW + w) * wlen] laid out as [W_hh | A | bias | W1h | b1 | W2 | b2 | W1s] (offsets from
 * mggan_decoder_bwd_fused_layout); reduce them with mggan_grad_reduce_multi (groups = n_gens, splits = NW).
 * SocR (R, S): the rows' social features as the forward saved them -- with it the block also carries W1s = dW1[:, H:]
 * (16 x 32, the social half of hidden2pos: dQ^T SocR); NULL: that part of the block is zero.
 * dEnc (R, EIN) = dH0 We2d[:, :EIN], or NULL: the caller folds dH0 over the rows of a pedestrian and multiplies once
 * per pedestrian (the adjoint of the Qe form of mggan_decoder_rollout_fwd). */
/* floats of padding between the tile records of the rollout's saved state: Gt is tiles x (T*H*64 + gt_pad) floats, Cs
 * tiles x ((T+1)*H*32 + cs_pad) (see mggan_decoder_rollout_fwd) */
int mggan_decoder_save_pads(int* gt_pad, int* cs_pad);
int mggan_decoder_bwd_fused_layout(int* wlen, int* off_A, int* off_bias, int* off_W1, int* off_b1, int* off_W2,
                                   int* off_b2, int* off_W1s);
int mggan_decoder_rollout_bwd_fused(int n_gens, int NW, int T, int H, int EIN, int Z, const int* seg, const int* row_pos,
                                    const float* W_hh, const float* W1, const float* W2, long param_stride,
                                    const float* We2d, const float* prep, int prep_stride, const float* Gt,
                                    const float* Cs, const float* Din,
                                    const float* Aact, const float* gabs, const float* grel, int Rout, float* dH0,
                                    float* dQ, float* dEnc, float* dSocR, float* wpart, const float* SocR,
                                    mggan_stream_t stream);

/* A ragged batch into the static buffers of its shape bucket (train()'s padded batches, mggan/abstract_train.py
 * IterationGraphs; reference collate /root/reference/mggan/data_utils/trajectories_scene.py:40-78): `descs` = n (<= 8) records
 * { const float* src; float* dst; long inner; int outer, position; }, every tensor (outer, pedestrians, inner) contiguous
 * with b pedestrians in src and b_pad in dst.  The real pedestrians are copied in front, the phantom pedestrians [b, b_pad)
 * get their constant content: 0, and -- position tensors -- element 0 of the inner axis = pedestrian % period (distinct
 * positions inside a phantom scene).  One launch. */
int mggan_pad_batch(const void* descs, int n, int b, int b_pad, int period, mggan_stream_t stream);
/* Crops of the AUGMENTED scene image (training; /root/reference/mggan/data_utils/trajectories_scene.py:276-357: flip ->
 * img.rotate(alpha / pi * 180, expand=True) -> img.resize(..., ANTIALIAS) -> one crop per pedestrian), computed directly from
 * the un-augmented scaled images resident in `atlas` -- bit-identical to Pillow's result.  items: one record of 26 int32 per
 * batch item { int64 img_off; int w, h, flip, rot, nw, nh, sw, sh; int a[6]; int ksh, ksv; int64 kh, bh, kv, bv; } (the
 * 16.16 inverse affine map of Pillow's nearest-neighbour rotation and the offsets of the item's Lanczos tables -- rows of ksh /
 * ksv 22-bit coefficients and (first source index, taps) pairs per output column / row -- in `tables`; built by
 * mggan/data_utils/aug_geometry.py); ped_item[p] = item of pedestrian p, centers (n, 2) = crop centre (x, y) in the resized
 * image; out (n, 4, 2 margin + 1, 2 margin + 1): RGB as -1 + v * 2 / 256 and the one-hot centre channel.  Limits: window <= 33,
 * <= 128 taps per pass (downscale <= 21), staged source rows <= 800 pixels (checked on the host side); `atlas` must be
 * readable 4 bytes past its last pixel (a pixel is fetched as one dword). */
int mggan_crop_patches_aug(const unsigned char* atlas, const void* items, const int* tables, const int* ped_item,
                           const int* centers, int n, int margin, int max_taps /* largest ksh of the batch's items */,
                           float* out, mggan_stream_t stream);
/* ... or, when the pedestrians of an item are many (their windows overlap: a 33 x 33 window of a 64 x 48 image), the WHOLE resized
 * image of every item first -- tile p of n_tiles is the 33 x 33 block around tile_center[2p..] (x, y) of item tile_item[p]'s
 * resized image, written as u8 RGB (sh, sw, 3) at small + small_off[item] -- and the crops as windows of those images
 * (mggan_crop_patches with atlas = small).  Same arithmetic, same bits; the host side picks the cheaper form per batch. */
int mggan_aug_small_images(const unsigned char* atlas, const void* items, const int* tables, const int* tile_item,
                           const int* tile_center, const int* tile_rows /* (n_tiles, 2) or NULL: rows [first, end) of the resized
                           image this entry computes -- the host splits tiles into row bands when a batch has few */,
                           int n_tiles, int max_taps, const long long* small_off, unsigned char* small, mggan_stream_t stream);
/* n (<= 8) small buffers copied in ONE launch: `descs` = n records { const void* src; void* dst; long bytes; } (<= 64 KB each).
 * Snapshot / roll-back of the discriminator's BatchNorm running statistics (nn.BatchNorm2d buffers, reference
 * /root/reference/mggan/model/modules/cnn.py:140-141) around the next iteration's discriminator context when it is issued
 * ahead of time (mggan/model/train.py: _issue_d_context / drain_pipeline). */
int mggan_copy_small(const void* descs, int n, mggan_stream_t stream);
/* ---- input pipeline: per-pedestrian scene crops cut on the GPU -----------------------------
 * Replaces the per-pedestrian PIL crop loop of BaseTrajectories.py:254-288 / trajectories_scene.py:349-356.
 * atlas = the u8 RGB "small" scene images (H,W,3) packed back to back in device memory; pedestrian p reads the
 * image at byte offset img_off[p] of size img_hw[2p] x img_hw[2p+1], window centre centers[2p] (x), [2p+1] (y);
 * out (n,4,2m+1,2m+1): channels 0-2 = -1 + v*2/256 (0 outside the image), channel 3 = one-hot centre. */
int mggan_crop_patches(const unsigned char* atlas, const long long* img_off, const int* img_hw, const int* centers,
                       int n, int margin, float* out, mggan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
