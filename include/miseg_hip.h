/*
 * miseg_hip.h -- C ABI of the MI355X (gfx950) hot path of the semi-supervised segmentation
 * train step (2D U-Net + global/local IIC mutual information + UDA consistency).
 *
 * The reference (jizongFox/MI-based-Regularized-Semi-supervised-Segmentation) is pure Python
 * on stock PyTorch: it has NO native/FFI layer.  The seam this library sits under is therefore
 * the reference's Python object API; every entry point below names the
 * reference call site whose ATen op sequence it replaces ("ref:" = path:line in the reference
 * checkout, "whl:" = inside its vendored deepclustering2 wheel).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch types.  All pointers are DEVICE pointers
 *     owned by the caller (PyTorch allocates; the library never allocates user-visible memory).
 *   - `stream` is a hipStream_t passed as void*; every launch goes on it; no internal threads,
 *     no host synchronisation, so calls may be captured into a hipGraph.
 *   - Return value: 0 = ok, <0 = error (MISEG_E_*).  miseg_last_error() gives the message of
 *     the last failure on the calling thread.  Nothing throws.
 *   - `dt` selects the storage/operand type of U-Net activations and packed weights:
 *     MISEG_F32 (exact fp32, v_mfma_f32_16x16x4_f32), MISEG_BF16 (bf16 operands,
 *     v_mfma_f32_16x16x32_bf16, fp32 accumulate) or MISEG_F16 (the same kernels on IEEE half: 3 more mantissa bits,
 *     5-bit exponent -- activation gradients need the caller's loss scaling).  Statistics, losses, probabilities,
 *     gradients of parameters and the optimiser are always fp32.
 *   - U-Net activations are NHWC ("channels last"): element (n,h,w,c) at ((n*H+h)*W+w)*C+c.
 *     Probability maps handed to the local-MI kernels are NCHW fp32 (the reference layout).
 *   - `flips` = int32[N] per-sample bit mask, bit0 = flip H, bit1 = flip W: the decisions the
 *     reference draws from Python's `random` under FixRandomSeed (ref: semi_seg/epocher.py:148-149,
 *     whl:deepclustering2/augment/tensor_augment.py:31-39).  The host draws; kernels only apply.
 *   - Scratch ("ws") sizes come from the matching *_ws_bytes() query.
 */
#ifndef MISEG_HIP_H
#define MISEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISEG_F32 0
#define MISEG_BF16 1
#define MISEG_F16 2 /* IEEE half storage / operands (v_mfma_*_f16), fp32 accumulate: the U-Net, BN and head entry points that take `dt` */

#define MISEG_OK 0
#define MISEG_E_INVALID (-1) /* bad argument (shape, dtype, null pointer, unsupported size) */
#define MISEG_E_LAUNCH (-2)  /* HIP launch failure; see miseg_last_error() */

int miseg_version(void);
const char* miseg_last_error(void);
/* Make stream `waiter` wait for the work enqueued on stream `producer` so far (both hipStream_t of this device, passed as void*): the
 * cross-stream ordering the host side needs around its side streams (weight gradients, IIC branch; the reference is single-stream, the
 * PyTorch call this replaces is torch.cuda.Stream.wait_stream).  Uses device-scope-release events (no system fence). */
int miseg_stream_wait_stream(void* waiter, void* producer);

/* ------------------------------------------------------------------------------------------
 * Launch tape: one iteration of the reference's hot loop (ref: semi_seg/epocher.py:143-187) issued from ONE C call.
 * The reference drives its step from Python, and so does this repo: ~300 launches on three streams from two Python threads
 * (caller + autograd engine).  With fixed shapes that list of (entry point, arguments, stream) is the same every iteration, so
 * the library records its own entry-point calls once -- every `int miseg_*(void* stream, ...)` below appends itself while a tape
 * is being recorded, whichever thread calls it -- and miseg_tape_replay calls them again in the recorded order: per-stream order
 * and cross-stream waits are exactly those of the recorded (eager) iteration, the host spends a few microseconds per launch.
 *   tape  = miseg_tape_begin()                 start recording (0 if another tape is being recorded); the calls still execute
 *   seg   = miseg_tape_mark(tape)              segment boundary: the host does something between segments (an RCCL all-reduce)
 *   miseg_tape_end(tape)
 *   slot  = miseg_tape_bind(tape, base, span)  every recorded POINTER argument p in [base, base + span) is re-based at replay:
 *                                              p' = values[slot] + (p - base).  Bound are the loader's batch tensors, the pinned
 *                                              staging slots of miseg_upload / miseg_download, the event of miseg_event_record.
 *   miseg_tape_replay(tape, segment, values, nvalues)      nvalues = number of bindings; they are applied at segment 0
 *   miseg_tape_time_op / miseg_tape_timed_ms   HIP-event pair around one op at every replay, on the op's own stream
 * Everything else the recorded calls point at must stay where it is (the host keeps the recorded iteration's allocations in
 * a private pool).  Entry points that read HOST arrays during the call (miseg_flip's strides) record their own copy of them.
 * ------------------------------------------------------------------------------------------ */
int64_t miseg_tape_begin(void);
int miseg_tape_end(int64_t tape);
int miseg_tape_free(int64_t tape);
int64_t miseg_tape_mark(int64_t tape);
int64_t miseg_tape_len(int64_t tape);
int64_t miseg_tape_segments(int64_t tape);
const char* miseg_tape_op_name(int64_t tape, int64_t op);
int64_t miseg_tape_op_stream(int64_t tape, int64_t op);
int64_t miseg_tape_bind(int64_t tape, const void* base, int64_t span);
int64_t miseg_tape_bind_uses(int64_t tape, int64_t slot);
int miseg_tape_replay(int64_t tape, int64_t segment, const void* const* values, int64_t nvalues);
int miseg_tape_time_op(int64_t tape, int64_t op);
int64_t miseg_tape_timed_ms(int64_t tape, int64_t op, float* ms, int64_t cap);
/* Host <-> device traffic of an iteration as entry points (so that it is on the tape): asynchronous copies between PINNED host
 * memory and the device on `stream` (the flip decisions and Adam's scalars go up, ref semi_seg/epocher.py:144-149; the meter values
 * come down, ref :181-185's .item() calls), and the event the host waits for before it reads them. */
int miseg_upload(void* stream, void* dst_dev, const void* src_pinned, int64_t nbytes);
int miseg_download(void* stream, void* dst_pinned, const void* src_dev, int64_t nbytes);
int64_t miseg_event_create(void);
int miseg_event_destroy(int64_t ev);
int miseg_event_record(void* stream, void* ev);
int miseg_event_synchronize(int64_t ev);
int miseg_event_query(int64_t ev);
/* Byte movers on `stream` (kernels: hipMemcpyDtoD / hipMemsetAsync can wait behind a busy device's copy queue); pointers 16-byte
 * aligned.  They stand where the reference's autograd issues zero fills and copies (zero gradient of a part without a loss, a
 * parameter that received no gradient: torch.optim's zero_grad at ref semi_seg/epocher.py:177). */
int miseg_fill_zero(void* stream, void* dst, int64_t nbytes);
int miseg_copy(void* stream, void* dst, const void* src, int64_t nbytes);
/* dst[o][j][0..chunk) = src[o][idx[j]][0..chunk) for o < outer, j < n_idx (fp32; idx = int32 device array of rows < n_src): the windows
 * of one colour group picked out of a per-window tensor -- overlapping patches (ref iic_loss.py:152-189) are differentiated in
 * pairwise-disjoint groups, one launch each. */
int miseg_gather_rows(void* stream, const float* src, float* dst, int64_t outer, int64_t n_src, int64_t n_idx, const int32_t* idx,
                      int64_t chunk);
/* Gradient of the logits batch [labeled | unlabeled | flipped unlabeled] (ref semi_seg/epocher.py:154-159 splits it, autograd
 * scales each loss's gradient by its coefficient, zero-fills the detached part and concatenates): out = [scale0[0] * src0 |
 * scale1[0] * src1 | scale2[0] * src2], fp32, a null src = zeros, a null scale = 1; numel_i multiples of 4. */
int miseg_assemble_rows(void* stream, float* out, const float* src0, const float* scale0, int64_t numel0, const float* src1,
                        const float* scale1, int64_t numel1, const float* src2, const float* scale2, int64_t numel2);

/* ------------------------------------------------------------------------------------------
 * Local (displacement-window) IIC mutual information
 * ref: contrastyou/losses/iic_loss.py:107-149 (IIDSegmentationLoss.__call__),
 *      :152-189 (patch_generator / IIDSegmentationSmallPathLoss).
 * x, y: fp32 NCHW [N,K,H,W] per-pixel simplexes.  `win` = int32[P][4] windows (h0,h1,w0,w1) in
 * reference iteration order; each window is treated as an independent zero-padded map exactly
 * like the reference's crop (:158) followed by conv2d(padding=pad) (:123).  mask: fp32 [N,1,H,W]
 * or NULL (:115-117).
 * precision : 0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32); 1 = bf16 MFMA with hi/lo operand split (3 products,
 *             fp32-class accuracy); 2 = plain bf16 operands; 3 = f16 hi x hi + both cross terms on the block-scaled fp8 MFMA
 *             (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3): the accuracy class of 1 at 2/3 of its matrix time; kernels without that
 *             form run 1.  1/2/3 fall back to 0 for shapes the 16-bit kernels do not cover.
 * joint_fwd : raw[P][T][T][K][K] (T = 2*pad+1), raw[p][a][b][i][j] =
 *             sum_{n,h,w} Xpad[n,i,h+a,w+b] * Y[n,j,h,w]              (the conv2d at :120-123)
 * loss_fwd  : per window: global-min shift +1e-16 (:124), per-displacement normalise (:129),
 *             symmetrise (:132), marginals (:135-136), -sum P(logP - lam logPi - lam logPj)/T^2
 *             (:139-146).  Writes loss[P] and grad_raw[P][T][T][K][K] = d loss[p] / d raw[p].
 * bwd       : gx, gy [N,K,H,W] += sum_p scale[p] * (d loss[p]/d x, y) through the joint and mask.
 *             scale = fp32[P] DEVICE array (upstream grad / P, the average_iter at :186).
 *             The windows of ONE call must be pairwise disjoint (the host colours overlapping patches into
 *             groups).  accumulate = 0: plain stores -- legal only if the call's windows cover every pixel
 *             of the map (e.g. the single whole-map window); accumulate = 1: gx, gy += (caller zero-fills).
 * ------------------------------------------------------------------------------------------ */
int64_t miseg_iic_local_joint_ws_bytes(int64_t N, int64_t K, int64_t H, int64_t W, int64_t pad, int64_t P);
int miseg_iic_local_joint_fwd(void* stream, const float* x, const float* y, const float* mask, int64_t N,
                              int64_t K, int64_t H, int64_t W, int64_t pad, const int32_t* win, int64_t P,
                              float* raw, void* ws, int64_t ws_bytes, int precision);
int miseg_iic_local_loss_fwd(void* stream, const float* raw, int64_t K, int64_t pad, int64_t P, float lamda,
                             float* loss, float* grad_raw);
/* The same epilogue with one block per (window, displacement) instead of one block per window -- same numbers for grad_raw,
 * loss summed over the displacements in d order -- for paddings whose (2 pad + 1)^2 displacements would take the single block
 * several rounds (pad >= 2).  ws: miseg_iic_local_loss_ws_bytes(pad, P) bytes (the per-displacement loss terms). */
int64_t miseg_iic_local_loss_ws_bytes(int64_t pad, int64_t P);
int miseg_iic_local_loss_fwd_ws(void* stream, const float* raw, int64_t K, int64_t pad, int64_t P, float lamda,
                                float* loss, float* grad_raw, void* ws, int64_t ws_bytes);
int miseg_iic_local_bwd(void* stream, const float* x, const float* y, const float* mask, int64_t N, int64_t K,
                        int64_t H, int64_t W, int64_t pad, const int32_t* win, int64_t P,
                        const float* grad_raw, const float* scale, float* gx, float* gy, int accumulate, int precision,
                        void* ws, int64_t ws_bytes);
int64_t miseg_iic_local_bwd_ws_bytes(int64_t K, int64_t pad, int64_t P);
/* All S sub-heads of a tap in one launch (ref semi_seg/epocher.py:264-272 loops `criterion(p[:ub], p[ub:]) for p in probs`):
 * probs fp32 [S][2*UB][K][H][W], x_s = probs[s][:UB], y_s = probs[s][UB:]; raw [S][P][T][T][K][K]; grad_raw likewise;
 * scale [S][P]; gprob like probs.  Workspaces: max of the single-head queries evaluated at P*S and at P (shapes the
 * batched bf16 kernels do not serve run one sub-head per launch). */
int miseg_iic_local_joint_fwd_heads(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                                    int64_t pad, const int32_t* win, int64_t P, float* raw, void* ws, int64_t ws_bytes,
                                    int precision);
int miseg_iic_local_bwd_heads(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                              int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale,
                              float* gprob, int accumulate, int precision, void* ws, int64_t ws_bytes);
/* Local-MI operand planes (K = 20, pad = 3: the f16 + fp8 backward).  The backward multiplies every probability as f16 hi + 8-bit cross
 * terms; with the three images of a map kept in memory PIXEL-major -- p16 [map][H][W][20] f16, p8l / p8h [map][H][W][24] e4m3 of
 * 2^20 (v - hi) and 2^8 v -- a source row of the kernel is a plain memory -> LDS copy (no registers, no conversions, issued a phase
 * ahead) instead of 40 loads and ~150 vector instructions per wave and row.  map = s * 2 UB + m, the order of
 * probs[S][2 UB][K][H][W] (ref contrastyou/losses/iic_loss.py:120-149 differentiated; the operands are the head's outputs,
 * contrastyou/trainer/_utils.py:149-154).  One buffer of miseg_iic_local_planes_bytes() bytes (0: shape not supported), written by
 * miseg_iic_local_make_planes or by the joint forward as a by-product.  miseg_iic_local_bwd_heads_planes = miseg_iic_local_bwd_heads
 * at precision 3, bit for bit, reading the planes in place of probs. */
int64_t miseg_iic_local_planes_bytes(int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W, int64_t pad);
int miseg_iic_local_make_planes(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W, int64_t pad,
                                void* planes, int64_t planes_bytes);
int miseg_iic_local_joint_fwd_heads_planes(void* stream, const float* probs, int64_t S, int64_t UB, int64_t K, int64_t H, int64_t W,
                                           int64_t pad, const int32_t* win, int64_t P, float* raw, void* ws, int64_t ws_bytes,
                                           void* planes, int64_t planes_bytes);   /* = ..._joint_fwd_heads at precision 3 + the planes */
int miseg_iic_local_bwd_heads_planes(void* stream, const void* planes, int64_t planes_bytes, int64_t S, int64_t UB, int64_t K, int64_t H,
                                     int64_t W, int64_t pad, const int32_t* win, int64_t P, const float* grad_raw, const float* scale,
                                     float* gprob, int accumulate, void* ws, int64_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Global IIC mutual information, S sub-heads in one launch
 * ref: contrastyou/losses/iic_loss.py:43-71 (IIDLoss.forward), :74-94 (compute_joint);
 *      wrapper semi_seg/_utils.py:12-15.
 * x, y: fp32 [S][N][K].  Outputs loss[S], loss_no_lamb[S], joint[S][K][K] and, for backward,
 * gx, gy [S][N][K] = upstream[s] * d loss[s]/d x,y  (upstream = fp32[S] device array).
 * ------------------------------------------------------------------------------------------ */
int miseg_iic_global_fwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K,
                         float lamb, float* loss, float* loss_no_lamb, float* joint);
int miseg_iic_global_bwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K,
                         float lamb, const float* upstream, float* gx, float* gy);
/* the same two for the layout the epocher has: prob[S][2N][K] with x = prob[:, :N], y = prob[:, N:] (the two views of the unlabeled
 * batch, semi_seg/epocher.py:258-262) -- no slice copies forward, ONE gradient tensor gprob[S][2N][K] backward. */
int miseg_iic_global_fwd_pair(void* stream, const float* prob, int64_t S, int64_t N, int64_t K, float lamb, float* loss,
                              float* loss_no_lamb, float* joint);
int miseg_iic_global_bwd_pair(void* stream, const float* prob, int64_t S, int64_t N, int64_t K, float lamb,
                              const float* upstream, float* gprob);

/* compute_joint on its own (ref iic_loss.py:74-94): joint [S][K][K] = normalised sum_n x_n (x) y_n, symmetrised iff `symmetric`;
 * bwd: gjoint = dL/djoint -> gx, gy [S][N][K]. */
int miseg_iic_global_joint_fwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, int symmetric,
                               float* joint);
int miseg_iic_global_joint_bwd(void* stream, const float* x, const float* y, int64_t S, int64_t N, int64_t K, int symmetric,
                               const float* joint, const float* gjoint, float* gx, float* gy);

/* ------------------------------------------------------------------------------------------
 * Cluster heads
 * ref: contrastyou/trainer/_utils.py:137-168 (LocalClusterHead: 1x1 conv C->K + bias, channel
 *      softmax, x S sub-heads) and :96-134 (ClusterHead: global avg pool, Linear C->K, softmax).
 * The flip replay of the tapped feature (ref: semi_seg/epocher.py:264-266) and the
 * cat([features_tf, tf_features]) (:269-271) are index math here: sample m of the output reads
 * feature sample src[m] (int32[M]) with flip mask flips[m].
 * feat: NHWC [B,H,W,C] (dt).  w: fp32 [S][K][C], b: fp32 [S][K].  T = softmax temperature.
 * local  fwd: prob fp32 [S][M][K][H][W] (NCHW per sub-head).
 * local  bwd: gprob same shape -> gfeat NHWC [B,H,W,C] (dt): rows src[m] are OVERWRITTEN with the gradient, all
 *             other rows are left as they are (callers pass a zeroed buffer); src[] must be pairwise distinct (each
 *             gfeat element then has one writer -- the epocher's src is an arange); gw [S][K][C], gb [S][K] overwritten.
 * global fwd: prob fp32 [S][M][K]; bwd likewise (flips are irrelevant under global pooling); gfeat rows src[m] are
 *             overwritten (zeroed buffer, distinct src -- as for the local head).
 * ------------------------------------------------------------------------------------------ */
/* simplex_violations (optional, needs K <= 32): += number of (sub-head, sample, pixel) positions whose K probabilities
 * do not sum to 1 within simplex_tol -- the consumer's `assert simplex(prob)` (ref iic_loss.py:28-29) for free. */
int miseg_head_local_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                         const int32_t* src, const int32_t* flips, int64_t M, const float* w, const float* b,
                         int64_t S, int64_t K, float T, float* prob, float simplex_tol, int32_t* simplex_violations);
int64_t miseg_head_local_bwd_ws_bytes(int64_t M, int64_t H, int64_t W, int64_t C, int64_t S, int64_t K);
int miseg_head_local_bwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                         const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S,
                         int64_t K, float T, const float* prob, const float* gprob, void* gfeat, float* gw,
                         float* gb, void* ws, int64_t ws_bytes);
/* the same with a COMPACT gradient: gfeat_rows = dt [rows][H][W][C] holds rows [row0, row0 + rows) of the batch only, every src[m] must
 * lie in that range (the epocher's src is the last 2 UB samples, ref semi_seg/epocher.py:258-259): no zero-filled full-batch tensor;
 * miseg_bn_relu_bwd_dual adds it to the feature's other gradient. */
int miseg_head_local_bwd_rows(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                              const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S,
                              int64_t K, float T, const float* prob, const float* gprob, void* gfeat_rows, int64_t row0,
                              float* gw, float* gb, void* ws, int64_t ws_bytes);
/* miseg_head_local_bwd_rows WITHOUT the saved probabilities: the kernel computes them again from feat, w and b by the forward
 * kernel's own operations (bit-equal p), so the backward reads gprob only -- half the bytes of this HBM-bound kernel.  The autograd
 * node of LocalClusterHead (ref contrastyou/trainer/_utils.py:137-168) keeps its inputs either way.  Shapes of
 * miseg_head_local_bwd_recompute_supported only (16-bit features, C = 16, S = 5, K = 20: the shipped top tap). */
int64_t miseg_head_local_bwd_recompute_supported(int dt, int64_t C, int64_t S, int64_t K);
int miseg_head_local_bwd_recompute(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                                   const int32_t* src, const int32_t* flips, int64_t M, const float* w, const float* b, int64_t S,
                                   int64_t K, float T, const float* gprob, void* gfeat_rows, int64_t row0, float* gw, float* gb,
                                   void* ws, int64_t ws_bytes);
/* the same, ADDING to gfeat_inout in place of storing (rows of src only; one rounding of the sum): for a tapped feature whose other
 * consumer -- DeConv_1x1 at the last decoder block, unet.py:84,129 -- has already written its input gradient there, so that autograd
 * has nothing left to add (a 300 MB elementwise pass on the step's critical path). */
int64_t miseg_head_local_bwd_acc_supported(int dt, int64_t C, int64_t S, int64_t K);
int miseg_head_local_bwd_acc(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                             const int32_t* src, const int32_t* flips, int64_t M, const float* w, int64_t S, int64_t K,
                             float T, const float* prob, const float* gprob, void* gfeat_inout, float* gw, float* gb,
                             void* ws, int64_t ws_bytes);
int miseg_head_global_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                          const int32_t* src, int64_t M, const float* w, const float* b, int64_t S, int64_t K,
                          float T, float* pooled, float* prob);
int miseg_head_global_bwd(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src,
                          int64_t M, const float* w, int64_t S, int64_t K, float T, const float* pooled,
                          const float* prob, const float* gprob, void* gfeat, float* gw, float* gb,
                          float* dz_ws /* fp32[S*M*K] scratch */);
/* compact form, as miseg_head_local_bwd_rows */
int miseg_head_global_bwd_rows(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src,
                               int64_t M, const float* w, int64_t S, int64_t K, float T, const float* pooled,
                               const float* prob, const float* gprob, void* gfeat_rows, int64_t row0, float* gw, float* gb,
                               float* dz_ws);

/* Head VARIANTS of the same two classes (csrc/heads_var.hip): head_type="mlp" (ref contrastyou/trainer/_utils.py:117-126 global,
 * :154-161 local: Linear/1x1-conv C->HID, LeakyReLU(0.01), Linear/1x1-conv HID->K; HID = 128 global, interm_dim = 64 local) and
 * normalize=True (ref :26-33, :113, :123, :150, :158: logits L2-normalised over the class axis, eps 1e-12, before softmax(./T)).
 * HID == 0 selects the single-layer head: w1 fp32 [S][K][C], b1 [S][K], w2/b2 unused (NULL).  HID > 0: w1 [S][HID][C], b1 [S][HID],
 * w2 [S][K][HID], b2 [S][K].  Same gather (src) / flip replay (flips) / layouts / gfeat contract as the linear entry points above;
 * the backward recomputes the forward from the features (nothing but the features is saved), gw1/gb1/gw2/gb2 are overwritten. */
int miseg_head_local_var_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                             const int32_t* src, const int32_t* flips, int64_t M, const float* w1, const float* b1,
                             int64_t HID, const float* w2, const float* b2, int64_t S, int64_t K, float T, int normalize,
                             float* prob);
int64_t miseg_head_local_var_bwd_ws_bytes(int64_t M, int64_t H, int64_t W, int64_t C, int64_t HID, int64_t S, int64_t K);
int miseg_head_local_var_bwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                             const int32_t* src, const int32_t* flips, int64_t M, const float* w1, const float* b1,
                             int64_t HID, const float* w2, const float* b2, int64_t S, int64_t K, float T, int normalize,
                             const float* gprob, void* gfeat, float* gw1, float* gb1, float* gw2, float* gb2, void* ws,
                             int64_t ws_bytes);
int miseg_head_global_var_fwd(void* stream, int dt, const void* feat, int64_t B, int64_t H, int64_t W, int64_t C,
                              const int32_t* src, int64_t M, const float* w1, const float* b1, int64_t HID, const float* w2,
                              const float* b2, int64_t S, int64_t K, float T, int normalize, float* pooled, float* prob);
int miseg_head_global_var_bwd(void* stream, int dt, int64_t B, int64_t H, int64_t W, int64_t C, const int32_t* src, int64_t M,
                              const float* w1, const float* b1, int64_t HID, const float* w2, const float* b2, int64_t S,
                              int64_t K, float T, int normalize, const float* pooled, const float* gprob, void* gfeat,
                              float* gw1, float* gb1, float* gw2, float* gb2, float* dpool_ws /* fp32[S*M*C] scratch */);

/* ------------------------------------------------------------------------------------------
 * Pixel-wise losses on logits (fp32 NHWC [N,H,W,C], C <= 32)
 * softmax_kl : ref whl:deepclustering2/loss/kl_losses.py:107-126 on softmax(logits)
 *              (semi_seg/epocher.py:165-166) with the one-hot of class2one_hot
 *              (whl:deepclustering2/utils/general.py:221-235) taken from int64 labels [N,H,W]:
 *              loss = mean_{n,h,w} sum_c -t*log((p+1e-16)/(t+1e-16)); glogits = upstream * d/dlogits.
 *              A label outside [0,C) sets *bad_label (int32) != 0 (the reference asserts).
 * softmax_mse: ref semi_seg/epocher.py:221-224 -- mean((softmax(a) - softmax(flip(b)))^2) over all
 *              elements, b detached; flip(b) per `flips` is index math.  ga = upstream * d/da.
 * softmax_klcons: the `UDARegCriterion.name: kl` form of the same term (ref semi_seg/trainer.py:137,194 with
 *              whl:deepclustering2/loss/kl_losses.py:107-126): mean_{n,h,w} sum_c -t*log((p+1e-16)/(t+1e-16)),
 *              p = softmax(a), t = softmax(flip(b)) detached.  Same arguments as softmax_mse.
 * `upstream` = fp32 device scalar (may be NULL = 1.0).  Results are deterministic (two-pass sums).
 * ------------------------------------------------------------------------------------------ */
int64_t miseg_loss_ws_bytes(int64_t N, int64_t H, int64_t W);
int miseg_softmax_klcons(void* stream, const float* a, const float* b, const int32_t* flips, int64_t N, int64_t H,
                         int64_t W, int64_t C, const float* upstream, float* loss, float* ga, void* ws, int64_t ws_bytes);
int miseg_softmax_kl(void* stream, const float* logits, const int64_t* labels, int64_t N, int64_t H, int64_t W,
                     int64_t C, const float* upstream, float* loss, float* glogits, int32_t* bad_label, void* ws,
                     int64_t ws_bytes);
int miseg_softmax_mse(void* stream, const float* a, const float* b, const int32_t* flips, int64_t N, int64_t H,
                      int64_t W, int64_t C, const float* upstream, float* loss, float* ga, void* ws,
                      int64_t ws_bytes);
/* per-sample H/W flip of an [N,C,H,W]-indexed tensor with arbitrary element strides (in elements),
 * elem_bytes in {2,4,8}; ref whl:deepclustering2/augment/tensor_augment.py:31-39. Bit-exact.
 * in_strides4 / out_strides4 are HOST arrays of 4 strides (read at launch time); in, out and flips are device pointers. */
int miseg_flip(void* stream, const void* in, void* out, int64_t N, int64_t C, int64_t H, int64_t W,
               const int64_t* in_strides4, const int64_t* out_strides4, int elem_bytes, const int32_t* flips);
/* The network's input batch out = [a | b | flip(b)] (contiguous [*,C,H,W], 4-byte elements; flips = int32[Nb]) in one pass;
 * ref semi_seg/epocher.py:148-153 (stack of per-sample flips, then torch.cat of labeled, unlabeled, transformed unlabeled). */
int miseg_cat_flip(void* stream, const void* a, int64_t Na, const void* b, int64_t Nb, int64_t C, int64_t H, int64_t W,
                   const int32_t* flips, void* out);
/* the same for ONE-channel fp32 images, writing besides `out` the stem's operand: pad8 = dt_pad [N][H][W][8], channel 0 = the pixel
 * rounded to dt_pad (MISEG_BF16 / MISEG_F16), channels 1..7 zero -- what miseg_cast_pad would make of `out` (unet.py:15, Conv1's input). */
int miseg_cat_flip_pad(void* stream, const float* a, int64_t Na, const float* b, int64_t Nb, int64_t H, int64_t W,
                       const int32_t* flips, float* out, int dt_pad, void* pad8);
/* argmax over channels + per-sample per-class intersection/union counts (int64 [N][C] each);
 * ref semi_seg/epocher.py:183 + whl:.../general_dice_meter.py:141-172. pred (int64 [N,H,W]) optional. */
int miseg_argmax_dice(void* stream, const float* logits, const int64_t* labels, int64_t N, int64_t H, int64_t W,
                      int64_t C, int64_t* pred, int64_t* inter, int64_t* uni);

/* The iteration's scalar report in one launch   ref: the .item() reads of semi_seg/epocher.py:115-121 and the inline
 * `if torch.isnan(loss): raise` / `assert simplex(..)` of iic_loss.py:28-29,132-133,184-186.
 * flat[>= C] fp32 (first C entries = the distinct device scalars the R reported values are linear combinations of, coeff[R][C];
 * further entries = vectors that are NaN-tested / float flags), iflat = int32 flags; desc[ncheck][3] = (kind, a, b):
 * 0 -> flat[a], 1 -> number of NaNs in flat[a .. a+b), 2 -> (float) iflat[a].  out[R + ncheck]: values (a non-finite input
 * poisons exactly the rows that use it), then the check flags in desc order. */
int miseg_report_scalars(void* stream, const float* flat, const int32_t* iflat, const float* coeff, int64_t R, int64_t C,
                         const int32_t* desc, int64_t ncheck, float* out);
/* counts the positions of x[outer][C][inner] (fp32) whose channel sum is not within tol of 1 (NaN counts): the device
 * half of `simplex` (ref whl:deepclustering2/utils/general.py simplex = allclose(sum(axis), 1)); adds into *count. */
int miseg_simplex_violations(void* stream, const float* x, int64_t outer, int64_t C, int64_t inner, float tol,
                             int32_t* count);

/* ------------------------------------------------------------------------------------------
 * U-Net building blocks   ref: contrastyou/arch/unet.py:10-40, 86-133
 * Weights arrive in the reference's OIHW fp32 layout and are re-packed per call into the
 * operand layout [tap][n][k] of type dt (pack kinds: 0 = forward: n = Cout, k = Cin; 1 = dgrad: taps
 * mirrored, n = input channels [ci_begin, ci_begin+ci_count) -- one slice per concat source -- k = Cout).
 * The 1-channel stem (unet.py:66) runs through the same kernels with its input zero-padded to one
 * 16-byte channel vector (miseg_cast_pad) and its weight padded likewise by the host.
 * conv3x3: stride 1, pad 1, no bias (unet.py:15,18,33).  Input = channel-concat of up to two
 * NHWC sources (torch.cat((skip, up),1), unet.py:109-125), each optionally read through a
 * nearest x2 upsample (nn.Upsample(scale_factor=2), unet.py:32): src s has C{s} channels and
 * spatial size (H>>ups{s}, W>>ups{s}).
 *   fwd   : out NHWC [N,H,W,Cout] (dt) = raw conv; if stats != NULL also writes per-block
 *           partial (sum, sumsq) of the fp32 accumulators for BatchNorm (see bn_finalize).
 *   dgrad : is conv3x3_fwd on grad_out with pack kind 1, Cin<->Cout.
 *   wgrad : gw OIHW fp32 [Cout][C0+C1][3][3] (overwritten), deterministic split-K.
 * ------------------------------------------------------------------------------------------ */
/* kind: 0 = forward layout, 1 = data-gradient layout of input channels [ci_begin, ci_begin + ci_count); kind = (Csrc << 8) (forward
 * layout only): w_oihw really is [Cout][Csrc][3][3] with Csrc < Cin and the missing input channels are packed as zeros -- the stem,
 * Conv2d(input_dim = 1, 16, 3) at unet.py:15 read as one 16-byte channel vector. */
int miseg_pack_conv3x3_weights(void* stream, int dt, const float* w_oihw, int64_t Cout, int64_t Cin, int kind,
                               int64_t ci_begin, int64_t ci_count, void* packed);
/* ... and the way back for that layer's weight gradient: gw [Cout][Cin][3][3] = the first Cin input channels of the
 * [Cout][Cin_pad][3][3] gradient miseg_conv3x3_wgrad computed. */
int miseg_conv3x3_wgrad_slice(void* stream, const float* gw_padded, int64_t Cout, int64_t Cin_pad, int64_t Cin, float* gw);
/* the same for many layers in ONE launch: jobs_dev = device array of njobs records
 *   struct miseg_pack_job { const float* w_oihw; void* packed; int32_t Cout, Cin, kind, ci_begin, ci_count, first_block; }  (40 bytes)
 * job j owns blocks [first_block[j], first_block[j+1]) of 256 elements each (first_block[0] = 0, ascending);
 * total_blocks = first_block of a virtual job njobs.  All jobs share dt. */
int miseg_pack_conv3x3_weights_multi(void* stream, int dt, const void* jobs_dev, int64_t njobs, int64_t total_blocks);
/* number of per-block (sum, sumsq) partial rows conv3x3_fwd writes: stats_partials = fp32[parts][2][Cout] */
int64_t miseg_conv3x3_stats_parts(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W);
/* ... of miseg_conv3x3_fwd proper (one row per block of the kernel that serves the shape; <= miseg_conv3x3_stats_parts, which stays the
 * row count of miseg_conv3x3_bn_fwd and of the epilogue reduce of miseg_conv3x3_dgrad_bn) */
int64_t miseg_conv3x3_fwd_parts(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout);
int miseg_conv3x3_fwd(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1,
                      int ups1, int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out,
                      float* stats_partials);
/* conv3x3 whose output leaves 2x2 SUM-pooled: out_pooled NHWC [N, H/2, W/2, Cout].  With pack kind 1 this is the data gradient of a
 * convolution that read its input through the nearest x2 upsample (unet.py:32): conv3x3_fwd + miseg_sumpool2x2 without the
 * full-resolution intermediate (4x the bytes of the result); the four fp32 sums of a block are added before the one rounding.
 * Single source, bf16 / fp16; the streaming shapes (Cin <= 32, large maps) and the tiled kernel's 64-channel-slice form (Cout > 32):
 * ask _supported, else use the two calls. */
int64_t miseg_conv3x3_fwd_sumpool_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout);
int miseg_conv3x3_fwd_sumpool(void* stream, int dt, const void* in, int64_t Cin, int64_t N, int64_t H, int64_t W,
                              const void* packed_w, int64_t Cout, void* out_pooled);
/* ... ADDING to inout_pooled instead of storing: the pre-upsample feature is also a local-MI tap whose head backward has already
 * written its gradient there (ops._GradJoin) -- replaces autograd's add of the two.  Streaming shapes only. */
int64_t miseg_conv3x3_fwd_sumpool_acc_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W);
int miseg_conv3x3_fwd_sumpool_acc(void* stream, int dt, const void* in, int64_t Cin, int64_t N, int64_t H, int64_t W,
                                  const void* packed_w, int64_t Cout, void* inout_pooled);
/* Data gradient of a convolution whose input was the channel concat of two full-resolution sources (torch.cat((skip, up), 1),
 * unet.py:109-125): conv3x3_fwd over grad_out [N,H,W,K] with the dgrad pack of ALL C0 + C1 input channels (kind 1, ci_begin 0,
 * ci_count C0 + C1), channels [0, C0) written to out0 [N,H,W,C0] and [C0, C0 + C1) to out1 [N,H,W,C1] -- one launch reads grad_out
 * once where two sliced launches read it twice.  C0 a multiple of 16, C1 of 4. */
int miseg_conv3x3_dgrad_dual(void* stream, int dt, const void* grad_out, int64_t K, int64_t N, int64_t H, int64_t W,
                             const void* packed_w, int64_t C0, void* out0, int64_t C1, void* out1);
/* conv3x3_fwd + bn_finalize in ONE launch (training mode): the block that finishes last sums the partial rows and writes
 * `saved` / the running statistics itself (same formulas as miseg_bn_finalize; the sums are combined in a different but fixed
 * order).  sync_counter: one int32 in device memory, 0 on entry, 0 again when the kernel ends -- the caller may hand the same
 * counter to launch after launch on one stream, not to two launches that can run concurrently.  Only for shapes whose partial
 * matrix one block can sum quickly: ask miseg_conv3x3_bn_fwd_fusable first (0 -> use conv3x3_fwd + bn_finalize). */
int64_t miseg_conv3x3_bn_fwd_fusable(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout);
int miseg_conv3x3_bn_fwd(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1,
                         int ups1, int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out,
                         float* stats_partials, const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* saved,
                         int32_t* sync_counter);
int64_t miseg_conv3x3_wgrad_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout);
int miseg_conv3x3_wgrad(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1,
                        int ups1, int64_t N, int64_t H, int64_t W, const void* gout, int64_t Cout, float* gw_oihw,
                        void* ws, int64_t ws_bytes);

/* BatchNorm2d (training: batch statistics; eval: running statistics) + ReLU
 * ref: unet.py:16-17,19-20,34-35 (nn.BatchNorm2d defaults eps=1e-5, momentum=0.1).
 * bn_finalize   : partials -> mean, invstd, scale=gamma*invstd, shift=beta-mean*scale
 *                 (saved[4][C]); updates running_mean/var (unbiased var) and
 *                 num_batches_tracked (int64 device scalar) when running_mean != NULL.
 * bn_eval_coeffs: saved[2..3] from running stats.
 * bn_relu_fwd   : y = relu(raw*scale+shift), NHWC dt, optional fused 2x2 max-pool second
 *                 output (nn.MaxPool2d(2,2), unet.py:61-64) when pooled != NULL.
 * bn_relu_bwd   : gy (+ optional gpool routed to the first arg-max of each 2x2 window, torch
 *                 semantics) -> graw (dt), ggamma, gbeta.  Two passes (reduce, apply).
 * ------------------------------------------------------------------------------------------ */
int miseg_bn_finalize(void* stream, const float* stats_partials, int64_t nparts, int64_t C, int64_t count,
                      const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, float* saved);
int miseg_bn_eval_coeffs(void* stream, int64_t C, const float* gamma, const float* beta, float eps,
                         const float* running_mean, const float* running_var, float* saved);
int miseg_bn_relu_fwd(void* stream, int dt, const void* raw, int64_t N, int64_t H, int64_t W, int64_t C,
                      const float* saved, void* y, void* pooled);
/* BatchNorm batch statistics without a finalize launch (training mode, ref contrastyou/arch/unet.py:14-21: Conv2d -> BatchNorm2d -> ReLU).
 * miseg_conv3x3_fwd_acc = miseg_conv3x3_fwd whose blocks ADD their per-channel sum and sum of squares into acc -- uint64[2 Cout + 1], 8-byte
 * aligned, ZERO before the launch -- as 2^-20 fixed point (integer adds: the totals do not depend on the arrival order, so results stay
 * bit-reproducible; a block sum of 2^30 or more, or a non-finite one, is counted in acc[2 Cout] and the layer's statistics then come out
 * NaN).  Shapes: miseg_conv3x3_fwd_acc_supported (Cout <= 256, at most 2 048 blocks: the no-wrap bound).  miseg_bn_relu_fwd_acc =
 * miseg_bn_finalize + miseg_bn_relu_fwd: every block turns the totals into scale / shift itself, block 0 writes `saved`
 * ([mean | invstd | scale | shift], what the backward reads) and moves running_mean / running_var / num_batches_tracked.  C <= 256. */
/* The stem convolution and its weight gradient without matrix cores (ref contrastyou/arch/unet.py:58: `conv_block(input_dim = 1, 16)`'s
 * first Conv2d): x = [N][H][W][CP] whose channel 0 is the image -- in 16-bit storage (the padded channel vector the other entry points
 * take; x_f32 = 0) or the fp32 image itself (x_f32 = 1, CP = 1: rounded to the storage type as it is read, so the padded copy need
 * not exist) --, w = the fp32 master weight [Cout][Cin_weight = 1][3][3].  A thread owns a pixel and four output channels; products and
 * accumulation as on the MFMA path (16-bit operands, fp32 sums).  acc_or_null: the BatchNorm accumulator of miseg_conv3x3_fwd_acc
 * (training) or null (evaluation).  miseg_conv3x3_stem_wgrad writes gw [Cout][1][3][3] (no padded gradient, no slice). */
int64_t miseg_conv3x3_stem_supported(int dt, int64_t Cin_weight, int64_t CP, int64_t Cout);
int miseg_conv3x3_stem_fwd(void* stream, int dt, const void* x, int x_f32, int64_t CP, int64_t N, int64_t H, int64_t W, const float* w,
                           int64_t Cin_weight, int64_t Cout, void* out, void* acc_or_null);
int64_t miseg_conv3x3_stem_wgrad_ws_bytes(int64_t Cout);
int miseg_conv3x3_stem_wgrad(void* stream, int dt, const void* x, int x_f32, int64_t CP, int64_t N, int64_t H, int64_t W, const void* graw,
                             int64_t Cout, float* gw, void* ws, int64_t ws_bytes);
int64_t miseg_conv3x3_fwd_acc_supported(int dt, int64_t Cin, int64_t N, int64_t H, int64_t W, int64_t Cout);
int miseg_conv3x3_fwd_acc(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1, int ups1,
                          int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cout, void* out, void* acc);
int miseg_bn_relu_fwd_acc(void* stream, int dt, const void* raw, int64_t N, int64_t H, int64_t W, int64_t C, const void* acc,
                          const float* gamma, const float* beta, float eps, float momentum, float* rmean, float* rvar, int64_t* nbt,
                          float* saved, void* y, void* pooled);
int64_t miseg_bn_bwd_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t C);
int miseg_bn_relu_bwd(void* stream, int dt, const void* raw, const void* y, const void* gy, const void* gpool,
                      int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved,
                      int training, void* graw, float* ggamma, float* gbeta, void* ws, int64_t ws_bytes);
/* the same with the statistics finished by the last reduce block (no separate finalize launch) when sync_counter != NULL and the
 * layer is narrow enough; sync_counter as in miseg_conv3x3_bn_fwd.  sync_counter == NULL is miseg_bn_relu_bwd. */
int miseg_bn_relu_bwd_sync(void* stream, int dt, const void* raw, const void* y, const void* gy, const void* gpool,
                           int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved,
                           int training, void* graw, float* ggamma, float* gbeta, void* ws, int64_t ws_bytes,
                           int32_t* sync_counter);
/* bn_relu_bwd with a SECOND gradient of the activation for samples [n2_begin, n2_end): gy2 = dt [n2_end - n2_begin][H][W][C], added to
 * gy in fp32 inside the loaders.  A tapped feature map feeds the next layer AND a cluster head (ref semi_seg/epocher.py:258-273, which
 * takes the last 2 UB samples); the reference's autograd adds the two gradients in a separate pass over the whole batch. */
int miseg_bn_relu_bwd_dual(void* stream, int dt, const void* raw, const void* gy, const void* gpool, const void* gy2,
                           int64_t n2_begin, int64_t n2_end, int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma,
                           const float* saved, int training, void* graw, float* ggamma, float* gbeta, void* ws,
                           int64_t ws_bytes);
/* The same without a finalize launch: the reduce kernel's blocks add their two sums per channel into acc -- uint64[4 C + 2], 8-byte aligned,
 * ZERO before the launch; two fixed-point tiers (2^-40 units for gradients as they come, 2^-12 for loss-scaled ones) and a misfit count
 * per tier -- and the apply kernel takes the finest tier every block fitted.  Integer adds: order-independent, bit-reproducible.
 * Shapes: miseg_bn_relu_bwd_acc_supported (C <= 256, at most 2 048 reduce blocks). */
int64_t miseg_bn_relu_bwd_acc_supported(int dt, int64_t N, int64_t H, int64_t W, int64_t C);
int miseg_bn_relu_bwd_dual_acc(void* stream, int dt, const void* raw, const void* gy, const void* gpool, const void* gy2, int64_t n2_begin,
                               int64_t n2_end, int64_t N, int64_t H, int64_t W, int64_t C, const float* gamma, const float* saved,
                               int training, void* graw, float* ggamma, float* gbeta, void* ws, int64_t ws_bytes, void* acc);
/* bn_relu_bwd WITHOUT its reduce pass: the per-block sums of dz and dz * xhat come from the epilogue of the convolution that wrote gy
 * (miseg_conv3x3_dgrad_bn called with red_raw / red_saved / red_parts and a plain graw input): finalize from ext_parts
 * [ext_nparts][2][C], then the apply pass.  Layers without the fused pool; ws >= 3 C floats. */
int miseg_bn_relu_bwd_ext(void* stream, int dt, const void* raw, const void* gy, int64_t N, int64_t H, int64_t W, int64_t C,
                          const float* gamma, const float* saved, int training, void* graw, float* ggamma, float* gbeta,
                          const float* ext_parts, int64_t ext_nparts, void* ws, int64_t ws_bytes);
/* BatchNorm backward folded into the convolutions around it (ref: unet.py:14-21, 31-36 -- what autograd runs as
 * threshold_backward -> native_batch_norm_backward -> convolution_backward becomes two launches per layer):
 *   bn_relu_bwd_stats : the statistics half of bn_relu_bwd for a layer WITHOUT the fused pool: ggamma, gbeta and
 *                       bwd_coef = fp32[6][C] (scale | shift | mean | A | P | Q) with which a consumer forms
 *                       graw = [relu(raw*scale+shift) > 0] * A * gy + P + Q * (raw - mean)
 *                       (A = gamma*invstd, P = -A*mean(dz), Q = -A*invstd*mean(dz*xhat); P = Q = 0 in eval mode) -- the tensor
 *                       bn_relu_bwd would have written.  ext_parts != NULL: the per-block sums [ext_nparts][2][C] were already taken
 *                       by the epilogue of the convolution that wrote gy (conv3x3_dgrad_bn's red_* arguments); raw / gy / ws unused.
 *   conv3x3_dgrad_bn  : conv3x3_fwd(pack kind 1) over graw formed in the tile loader from (raw, gy, bwd_coef) -- K = the layer's
 *                       output channels, Cs = channels of the gradient written; gy == NULL: raw_or_graw is a plain graw tensor.
 *                       pool_out: the output leaves 2x2 sum-pooled (conv3x3_fwd_sumpool).  red_raw != NULL (full-resolution output
 *                       only): `out` is the activation gradient of ANOTHER BatchNorm layer whose raw output / saved[4][Cs] are
 *                       red_raw / red_saved; the epilogue writes that layer's per-block backward sums to red_parts
 *                       [conv3x3_dgrad_red_parts(...)][2][Cs] (0 parts: not available for the shape).
 *   conv3x3_wgrad_bn  : conv3x3_wgrad with the same loader in place of gout. */
int miseg_bn_relu_bwd_stats(void* stream, int dt, const void* raw, const void* gy, int64_t N, int64_t H, int64_t W, int64_t C,
                            const float* gamma, const float* saved, int training, float* bwd_coef, float* ggamma,
                            float* gbeta, const float* ext_parts, int64_t ext_nparts, void* ws, int64_t ws_bytes);
int64_t miseg_conv3x3_dgrad_bn_supported(int dt, int64_t K, int64_t N, int64_t H, int64_t W, int64_t Cs, int pool_out);
int64_t miseg_conv3x3_dgrad_red_parts(int dt, int64_t K, int64_t N, int64_t H, int64_t W, int64_t Cs);
int miseg_conv3x3_dgrad_bn(void* stream, int dt, const void* raw_or_graw, const void* gy, const float* bwd_coef, int64_t K,
                           int64_t N, int64_t H, int64_t W, const void* packed_w, int64_t Cs, void* out, int pool_out,
                           const void* red_raw, const float* red_saved, float* red_parts);
int miseg_conv3x3_wgrad_bn(void* stream, int dt, const void* in0, int64_t C0, int ups0, const void* in1, int64_t C1,
                           int ups1, int64_t N, int64_t H, int64_t W, const void* raw, const void* gy,
                           const float* bwd_coef, int64_t Cout, float* gw_oihw, void* ws, int64_t ws_bytes);
/* backward of nearest x2 upsample: out[n,h,w,c] = sum of the 2x2 block of in (NHWC dt). */
int miseg_sumpool2x2(void* stream, int dt, const void* in, int64_t N, int64_t H, int64_t W, int64_t C, void* out,
                     int accumulate);
/* dt tensor elementwise: dst += src (skip-connection gradient joins). */
int miseg_axpy(void* stream, int dt, const void* src, void* dst, int64_t numel);
/* fp32 image [npix][Cin] -> dt [npix][CP], channels >= Cin zero-filled (stem input, CP = one 16-byte vector). */
int miseg_cast_pad(void* stream, const float* in, int64_t npix, int64_t Cin, int dt_out, void* out, int64_t CP);

/* the 1x1 logits head with bias (unet.py:84,129): in NHWC dt [N,H,W,16] -> logits fp32 NHWC [N,H,W,Cout]. */
int miseg_conv1x1_fwd(void* stream, int dt, const void* in, int64_t N, int64_t H, int64_t W, int64_t Cin,
                      const float* w, const float* bias, int64_t Cout, float* out);
int64_t miseg_conv1x1_bwd_ws_bytes(int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout);
int miseg_conv1x1_bwd(void* stream, int dt, const void* in, const float* gout, int64_t N, int64_t H, int64_t W,
                      int64_t Cin, const float* w, int64_t Cout, void* gin, float* gw, float* gbias, void* ws,
                      int64_t ws_bytes);

/* ------------------------------------------------------------------------------------------
 * Adam with L2 weight decay on one flat fp32 buffer   ref: semi_seg/trainer.py:67-72,179-184
 * (torch.optim.Adam semantics).  step/lr live on the host; bias corrections are computed on the
 * host in double and passed in, so the launch is graph-capturable with scalars in `hyper`
 * (fp32[4] device: lr/bc1, 1/sqrt(bc2), eps, weight_decay) refreshed by a tiny memcpy.
 * ------------------------------------------------------------------------------------------ */
int miseg_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    int64_t numel, float beta1, float beta2, const float* hyper);
/* The same update on gradients that carry a static loss scale (the fp16 storage mode seeds backward with
 * grad_scale so that activation gradients stay inside half's range): grad is read as grad / grad_scale. */
int miseg_adam_step_scaled(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                           int64_t numel, float beta1, float beta2, const float* hyper, float grad_scale);
/* ... and guarded: guard = nguard fp32 flags in device memory (the iteration's deferred assertion / NaN flags); if any of them is
 * non-zero or NaN the launch changes nothing -- parameters and moments stay as the reference, which raises before backward
 * (iic_loss.py:147-148, semi_seg/epocher.py:129-130), would have left them.  nguard == 0: miseg_adam_step_scaled.
 * grad_scale == -1: the loss scale lives on the device, hyper is fp32[5] and hyper[4] = 1 / scale (a dynamic loss scale under a
 * replayed launch tape, whose by-value arguments are fixed). */
int miseg_adam_step_guarded(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            int64_t numel, float beta1, float beta2, const float* hyper, float grad_scale,
                            const float* guard, int64_t nguard);
/* Overflow test of the half-precision modes (BASELINE configs[4]: torch.cuda.amp-style loss scaling; the reference itself trains in
 * fp32 and has no such step): count[0] = number of non-finite entries of grad[0..numel), as a float (0 = clean).  The caller appends
 * it to the guard flags of miseg_adam_step_guarded -- an overflowed gradient then moves neither parameters nor moments -- and the
 * host lowers the loss scale when the count arrives with the iteration's scalars. */
int miseg_count_nonfinite(void* stream, const float* grad, int64_t numel, float* count);

/* ------------------------------------------------------------------------------------------
 * Device input pipeline   ref: semi_seg/augment.py:7-52 (ACDCStrongTransforms),
 * contrastyou/augment/sequential_wrapper.py:11-100, whl:deepclustering2/augment/pil_augment.py
 * (RandomRotation / RandomCrop / CenterCrop / flips / ToTensor / ToLabel), torchvision 0.7 ColorJitter,
 * Pillow Image.rotate (NEAREST -> libImaging affine_fixed) and Image.blend.
 *
 * atlas_img / atlas_gt : u8 [n_slices, slice_h, slice_w] resident in HBM (atlas_gt may be NULL with gt_out NULL)
 * jobs_dev             : int32 [njobs, MISEG_AUG_JOB_INTS] on the device, one augmentation job per output slice:
 *    [0] slice index  [1],[2] unused  [3] number of geometric ops (<= MISEG_AUG_MAX_GEO)  [4] number of colour ops (<= 3)
 *    [5..7] colour op codes in application order   [8..10] their factors (fp32 bit patterns)   [11] unused
 *    [12 + 9*g ...] geometric op g in application order: {type, p1..p6, in_w, in_h}:
 *        MISEG_AUG_CROP   p1 = top, p2 = left                       (out(y,x) = in(y+top, x+left), zero outside)
 *        MISEG_AUG_VFLIP / MISEG_AUG_HFLIP
 *        MISEG_AUG_AFFINE p1..p6 = a0,a1,a2,a3,a4,a5 in 16.16 fixed point, a2/a5 already including the half-pixel
 *                         terms, exactly libImaging's affine_fixed: xin = (a2 + y*a1 + x*a0) >> 16, zero outside
 *      in_w, in_h = size of the image the op reads.
 * img_out : fp32 [njobs, out_h, out_w] = u8 / 255 (ToTensor); gt_out : int64 [njobs, out_h, out_w] (ToLabel).
 * out_h * out_w <= 65536.  Bit-exact against the PIL chain for equal parameters.
 * ------------------------------------------------------------------------------------------ */
#define MISEG_AUG_JOB_INTS 48
#define MISEG_AUG_MAX_GEO 4
#define MISEG_AUG_CROP 1
#define MISEG_AUG_VFLIP 2
#define MISEG_AUG_HFLIP 3
#define MISEG_AUG_AFFINE 4
#define MISEG_AUG_BRIGHTNESS 1
#define MISEG_AUG_CONTRAST 2
#define MISEG_AUG_SATURATION 3

/* Host planner: item seeds -> job table, with the reference's random streams (contrastyou/augment/sequential_wrapper.py:27-100;
 * pil_augment get_params; torchvision 0.7 ColorJitter.get_params; CPython random = MT19937).  Pure host function.
 * recipe     : the transform preset (semi_seg/augment.py:7-52) in declarative form
 * item_seeds : SequentialWrapperTwice's global_seed per item (>= 0); slice_ids / widths / heights per item
 * jobs_out   : HOST int32 [views * n_items, MISEG_AUG_JOB_INTS], view-major (all first views, then all second views)
 * out_wh     : HOST int32[2] = output width, height (all items must agree, as default_collate demands) */
#define MISEG_RECIPE_ROTATE 1       /* arg = degrees: uniform(-arg, arg) */
#define MISEG_RECIPE_VFLIP 2        /* arg = probability */
#define MISEG_RECIPE_HFLIP 3
#define MISEG_RECIPE_RANDOM_CROP 4  /* arg = size */
#define MISEG_RECIPE_CENTER_CROP 5
typedef struct {
    int32_t n_geo;
    int32_t geo_kind[4];
    double geo_arg[4];
    int32_t has_jitter;
    double jitter[6]; /* brightness lo,hi, contrast lo,hi, saturation lo,hi */
    int32_t twice;
    int32_t total_freedom;
} miseg_aug_recipe;
int miseg_plan_augment(const miseg_aug_recipe* recipe, int64_t n_items, const int64_t* item_seeds,
                       const int32_t* slice_ids, const int32_t* widths, const int32_t* heights, int32_t* jobs_out,
                       int32_t* out_wh);

int miseg_augment_slices(void* stream, const uint8_t* atlas_img, const uint8_t* atlas_gt, int64_t n_slices,
                         int64_t slice_h, int64_t slice_w, const int32_t* jobs_dev, int64_t njobs, int64_t out_h,
                         int64_t out_w, float* img_out, int64_t* gt_out);

#ifdef __cplusplus
}
#endif
#endif /* MISEG_HIP_H */
