/* act3d_hip.h -- C-ABI of libact3d_hip.so: the MI355X (gfx950) kernels behind the Act3D keypose and
 * ChainedDiffuser trajectory hot path.
 *
 * The reference (zhouxian/act3d-chained-diffuser) is pure Python/PyTorch and has no FFI of its own for this
 * path; every entry point below replaces a sequence of aten ops, cited as reference file:line.  A maintainer
 * binds them with ctypes; the package in this repo does exactly that.
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch's allocator); the library never allocates
 *     or frees device memory and keeps no state besides a thread-local error string.  There is no workspace object
 *     (SURVEY 8b sketched a3d_workspace_create / _destroy): workspaces are caller-owned pointers whose sizes the
 *     `*_ws_bytes` / `*_ws_floats` / `*_sync_ints` / `*_floats` queries return, which keeps every entry capturable;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*): no host synchronisation, no
 *     default-stream use, so a sequence of calls is capturable with hipStreamBeginCapture / torch.cuda.graph;
 *   - return value 0 = ok, negative errno-style code otherwise (-22 bad argument, -5 launch failure);
 *     a3d_last_error_string() describes the last failure on the calling thread; no exceptions, no exit();
 *   - tensors are contiguous row-major fp32 unless stated; token tensors are batch-first (B, N, E) -- the
 *     reference's sequence-first (N, B, E) is a view concern of the Python shim;
 *   - indices are int64 (long long) at the API, as torch.topk / torch.max return them.
 *
 * Attention operand formats (written by a3d_proj_rope_split / a3d_rope_split*, read by a3d_attn_*):
 *   rows  : [B][H][Npad][W] bf16, head dim 15 padded to 16 with zero;
 *             W = 48 for q and k ("QK"): row = hi(16) | lo(16) | lo2(16), x ~= hi + lo + lo2 (fp32-grade logits),
 *             W = 32 for v and dO rows (backward only): row = hi(16) | lo(16), x ~= hi + lo
 *   planes: [B][H][2][16][Npad] bf16 ("VT"), plane 0 = hi, plane 1 = lo (transposed: keys contiguous)
 *   Npad % 64 == 0 for keys/values, Npad % 16 == 0 suffices for queries when written by a3d_rope_split_qk with
 *   Npad % 64 == 0 (callers simply use a multiple of 64 everywhere).
 */
#ifndef ACT3D_HIP_H
#define ACT3D_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int a3d_version(void);
const char* a3d_last_error_string(void);

/* ---- dense layers ------------------------------------------------------------------------------------ */
/* Y[m,n] = act(sum_k X[m,k] * W(n,k) + bias[n]);  W(n,k) = W[n*ldw+k], or W[k*ldw+n] if w_transposed (dgrad).
 * act: 0 none, 1 relu, 2 multiply by (mask[m*ldm+n] > 0) (ReLU backward fused into dgrad), 3 Y += result (a gradient summed in place:
 * the tensor's consumers accumulate into one buffer instead of autograd adding their outputs).
 * Replaces F.linear at multihead_custom_attention.py:246-303,447; layers.py:88-94,313-332; diffusion_head.py:41-49. */
int a3d_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                   const float* mask, int ldm, int M, int N, int K, int act, int w_transposed, void* stream);
/* dW[n*lddw+k] += sum_m dY[m,n] X[m,k];  db[n] += sum_m dY[m,n] (db may be NULL).  Accumulates atomically. */
int a3d_linear_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw, float* db, int M,
                     int N, int K, void* stream);
/* Same, with a caller-provided workspace: for M >= 1024 the M reduction is split into 64..256-row chunks whose partial
 * tiles go to `ws`, and a second kernel adds them into dW/db in a fixed order -- deterministic and free of memory-side
 * float atomics.  a3d_linear_wgrad_ws_bytes returns the size needed (0 = the one-stage atomic kernel is used and ws
 * may be NULL). */
size_t a3d_linear_wgrad_ws_bytes(int M, int N, int K, int has_bias);
int a3d_linear_wgrad_ws(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw, float* db, int M,
                        int N, int K, float* ws, size_t ws_bytes, void* stream);
/* Y = LayerNorm(A + R) * gamma + beta over the last dim (R may be NULL); saves mean/rstd per row.
 * Replaces `output = self.norm(query + dropout(attn_output))` layers.py:308-309, 329-331, 158-159. */
int a3d_add_layernorm_fwd(const float* A, const float* R, const float* gamma, const float* beta, float* Y,
                          float* mean, float* rstd, int M, int E, float eps, void* stream);
/* dS = d(A+R); dgamma/dbeta accumulate atomically (both NULL to skip). */
int a3d_add_layernorm_bwd(const float* A, const float* R, const float* gamma, const float* mean, const float* rstd,
                          const float* dY, float* dS, float* dgamma, float* dbeta, int M, int E, void* stream);

/* ---- RoPE-3D + operand formatting ---------------------------------------------------------------------- */
/* dst(QK format) = split_bf16(rope3d(Y[:, :E] * scale, xyz)); xyz NULL = no rotation.  freq: E/6 device floats
 * = exp(arange(0,E/3,2) * (-ln(1e4)/(E/3))).  Replaces RotaryPositionEncoding3D.forward + embed_rotary
 * (position_encodings.py:31-34,64-97) and `q = q * scaling` (multihead_custom_attention.py:325,348-359). */
int a3d_rope_split_qk(const float* Y, int ldy, const float* xyz, const float* freq, float scale, void* dst, int B,
                      int N, int Npad, int E, int H, void* stream);
int a3d_split_vt(const float* Y, int ldy, void* dst, int B, int N, int Npad, int E, int H, void* stream);
/* Both formats of the same rotated rows in one pass (either output may be NULL); the backward needs q, k, v in both.
 * rows_width: 48 (q, k: hi|lo|lo2) or 32 (v: hi|lo); a3d_rope_split_qk writes 48-wide rows. */
int a3d_rope_split(const float* Y, int ldy, const float* xyz, const float* freq, float scale, void* rows_out,
                   int rows_width, void* planes_out, int B, int N, int Npad, int E, int H, void* stream);
/* In-projection + RoPE + operand formatting in one pass: for output block j in {0, 1} (block 1 optional: rows1 and
 * planes1 both NULL), Y_j = (X W[jE:(j+1)E, :K]^T + bias[jE:(j+1)E]) * scale_j, rotated by xyz_j (NULL: none), written as
 * rows (width 32 | 48, may be NULL) and planes (may be NULL).  X: [B*N][ldx] fp32; the projected rows never reach HBM.
 * Replaces the F.linear in-projections (multihead_custom_attention.py:246-303) feeding a3d_rope_split.
 * K, ldx multiples of 4, X 16-byte aligned (W may be 4-byte aligned: flat parameter buffers); E <= 128. */
int a3d_proj_rope_split(const float* X, int ldx, const float* W, int ldw, const float* bias, int K, const float* xyz0,
                        float scale0, void* rows0, int rows0_width, void* planes0, const float* xyz1, float scale1,
                        void* rows1, int rows1_width, void* planes1, const float* freq, int B, int N, int Npad, int E,
                        int H, void* stream);
/* dY[:, :E] = scale * R(xyz)^T * sum_s dR[s];  dR: [nsplit][B][H][Npad][16] fp32 (grad w.r.t. rotated rows). */
int a3d_rope_merge_bwd(const float* dR, int nsplit, const float* xyz, const float* freq, float scale, float* dY,
                       int ldy, int B, int N, int Npad, int E, int H, void* stream);

/* ---- attention core ------------------------------------------------------------------------------------- */
/* O[b,q,h*15+d] = softmax_k(q.k + mask) v ; LSE[b,h,q] saved for backward.  kmask: [B][S] uint8, 1 = padded key
 * (may be NULL).  nsplit > 1 splits the key range over workgroups (ws >= a3d_attn_fwd_ws_floats floats).
 * Replaces bmm/softmax/bmm + key_padding_mask of multihead_custom_attention.py:386-447. */
int a3d_attn_fwd(const void* Qs, const void* Ks, const void* Vt, const unsigned char* kmask, float* O, float* LSE,
                 float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, void* stream);
size_t a3d_attn_fwd_ws_floats(int B, int H, int Lqp, int nsplit);
/* Backward of the above on split-bf16 MFMA: outputs (gradients w.r.t. the rotated rows) dQp [nsplit][B][H][Lqp][16], dK / dV
 * [B][H][Sp][16].  Needs both formats of q, k, v: rows Qs, Ks (48-wide), Vs (32-wide)
 * and planes Qt, Kt (a3d_rope_split writes both).  Scratch: dOs [B][H][Lqp][32] bf16, dOt [B][H][2][16][Lqp] bf16,
 * D [B][H][Lqp].  Lqp % 64 == 0. */
int a3d_attn_bwd_bf16(const void* Qs, const void* Qt, const void* Ks, const void* Kt, const void* Vs,
                      const unsigned char* kmask, const float* O, const float* dO, const float* LSE, void* dOs,
                      void* dOt, float* D, float* dQp, float* dK, float* dV, int B, int H, int Lq, int Lqp, int S,
                      int Sp, int nsplit, void* stream);

/* ---- training-mode dropout (ChainedDiffuser transformer, p = 0.1: layers.py:10,34,58,82-84; diffusion_head.py:46,183,193;
 *      attention weights multihead_custom_attention.py:413) ---------------------------------------------------------
 * Counter-based (Philox4x32-10): the keep flag of an element is a pure function of (drop_state = {seed, offset} uint64[2]
 * on the device, site, element index), so backward passes regenerate the mask and a training step is capturable.
 * An element is kept iff its 16-bit draw >= round(p * 65536); kept values are scaled by 1 / (1 - p).
 * y = x o keep / (1 - p) over a flat fp32 array; the same call is its own backward (dx from dy).  y may alias x. */
int a3d_dropout(const float* x, float* y, size_t n, const unsigned long long* drop_state, unsigned int site, float p,
                void* stream);
/* The layer and the nn.Dropout behind it in one launch: Y = dropout(act(X W^T + b)) with a3d_dropout's mask over the flat index
 * m * N + n of the contiguous output (ldy == N required; act 3 is not accepted).  Bit-identical to a3d_linear_fwd followed by
 * a3d_dropout(Y -> Y), which is also what it runs when the epilogue does not apply (N % 8 != 0, row counts served by the bf16x3
 * kernel).  Replaces `dropout(relu(linear1(x)))`, `dropout(linear2(...))` layers.py:82-84, `dropout(attn_output)` layers.py:146,181,
 * and with act 2 / w_transposed their backward (dgrad, ReLU mask, dropout of the hidden gradient). */
int a3d_linear_fwd_drop(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                        const float* mask, int ldm, int M, int N, int K, int act, int w_transposed,
                        const unsigned long long* drop_state, unsigned int site, float p, void* stream);
/* a3d_add_layernorm_bwd with a second output dS_drop = dropout(dS) (a separate buffer): the gradient of LayerNorm(x + dropout(branch))
 * with respect to x (dS) and to the branch (dS_drop).  Bit-identical to a3d_add_layernorm_bwd + a3d_dropout(dS -> dS_drop). */
int a3d_add_layernorm_bwd_drop(const float* A, const float* R, const float* gamma, const float* mean, const float* rstd,
                               const float* dY, float* dS, float* dS_drop, float* dgamma, float* dbeta, int M, int E,
                               const unsigned long long* drop_state, unsigned int site, float p, void* stream);
/* out[i] = keep flag (0 / 1) of element i of a block row: c2 == 0xFFFFFFFF selects the flat elementwise indexing of
 * a3d_dropout (c1 ignored); otherwise (c2 = b * H + h, c1 = query) the attention-weight indexing, i = key.  Test hook. */
int a3d_dropout_mask(unsigned char* out, size_t n, const unsigned long long* drop_state, unsigned int c2, unsigned int c1,
                     unsigned int site, float p, void* stream);
/* ---- split-fp16 attention (csrc/attention16.hip): the default attention core --------------------------------------------
 * Replaces the same reference lines as a3d_attn_fwd / a3d_attn_bwd_bf16 (multihead_custom_attention.py:386-447 and its
 * autograd) with half the MFMA work: q, k two-part fp16 (x = hi + lo, fp32-grade logits), P / dS / V / dO single fp16.
 * Operand formats "16": rows16 [B][H][Npad][32] fp16 = hi(16) | lo(16); planes16 [B][H][parts][16][Npad] fp16, transposed
 * (parts & 3 = 2: hi and lo planes, what the kernels read; 1: hi only; parts | 4: padded channel 15 of the hi plane = 1.0 --
 * REQUIRED for the value planes, it is the softmax-denominator channel of the forward's PV product; parts | 8: the same 1.0 in
 * channel 15 of the hi part of the ROWS -- REQUIRED for the value rows of a3d_attn16_fwd_rows).
 * q must carry log2(e) (pass scale * log2 e to the *_split16 writers; a3d_rope_merge_bwd takes the same scale): scores and
 * LSE2 are in log2 units.  drop_state NULL or drop_p == 0: no dropout; otherwise Philox keep flags as a3d_attn_fwd_dropout.
 * Sp <= 16384. */
int a3d_rope_split16(const float* Y, int ldy, const float* xyz, const float* freq, float scale, void* rows_out,
                     void* planes_out, int plane_parts, int B, int N, int Npad, int E, int H, void* stream);
int a3d_proj_rope_split16(const float* X, int ldx, const float* W, int ldw, const float* bias, int K, const float* xyz0,
                          float scale0, void* rows0, void* planes0, int parts0, const float* xyz1, float scale1, void* rows1,
                          void* planes1, int parts1, const float* freq, int B, int N, int Npad, int E, int H, void* stream);
/* O [B][Lq][E] fp32, LSE2 [B][H][Lqp] fp32 (log2 units).  Qr, Kr rows16; Vp planes16 with BOTH parts.  ws as a3d_attn_fwd. */
int a3d_attn16_fwd(const void* Qr, const void* Kr, const void* Vp, const unsigned char* kmask, float* O, float* LSE2,
                   float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                   const unsigned long long* drop_state, unsigned int drop_site, float drop_p, void* stream);
/* The same forward on the ROWS-ONLY operand set (round 6, the default of ops.py): Vr = value rows16 whose padded channel 15 of the hi
 * part is 1.0 (writers: parts | 8).  The V^T fragments of the PV product are transposed LDS reads (ds_read_b64_tr_b16) of the rows
 * tile, so the projection writes ONE layout of K and of V -- half the stores of the planes + rows set, and a3d_attn16_bwd reads the
 * same tensors.  nograd != 0: the caller keeps no gradient (evaluation, sampling): the low part of P is then formed only for
 * chunks that hold a dominant key (weight > 2^-6 of the running denominator); NEVER set it on a pass whose backward will run. */
int a3d_attn16_fwd_rows(const void* Qr, const void* Kr, const void* Vr, const unsigned char* kmask, float* O, float* LSE2,
                        float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                        const unsigned long long* drop_state, unsigned int drop_site, float drop_p, int nograd, void* stream);
/* Needs rows16 of q, k, v (v rows with or without the ones channel).  Qp, Kp (planes16 of q, k) are accepted for ABI stability and
 * IGNORED since round 6 (may be NULL): the dQ kernel forms K^T by transposed LDS reads of the K rows.  Scratch: dOr [B][H][Lqp][32] fp16 (hi | lo), pack
 * (a3d_attn16_bwd_pack_bytes: the query-side operands of the dK / dV kernel as 20 KB LDS images per 64 rows, rows sorted by
 * gradient magnitude -- block floating point over the query axis), D [B][H][Lqp] fp32, rexp [B][H][Lqp] int32.  Outputs:
 * dQp [nsplit][B][H][Lqp][16] (gradient w.r.t. the log2e-scaled, rotated q), dK, dV [B][H][Sp][16] fp32 -- the
 * a3d_rope_merge_bwd layouts.  Lqp % 64 == 0.  A3D_ATTN_FAST=1 in the environment selects single-fp16 P (forward) and dS
 * (backward): ~25 % faster, 3e-4-class instead of 1e-5-class parity (not the default). */
size_t a3d_attn16_bwd_pack_bytes(int B, int H, int Lqp);
int a3d_attn16_bwd(const void* Qr, const void* Qp, const void* Kr, const void* Kp, const void* Vr,
                   const unsigned char* kmask, const float* O, const float* dO, const float* LSE2, void* dOr, void* pack,
                   float* D, int* rexp, float* dQp, float* dK, float* dV, int B, int H, int Lq, int Lqp, int S, int Sp,
                   int nsplit, const unsigned long long* drop_state, unsigned int drop_site, float drop_p, void* stream);

/* OPT-IN fp8 forward (BASELINE.json configs[4], "fp8 MFMA attention"; ops.ATTN_MODE = "fp8"): e4m3fn operands on
 * v_mfma_f32_16x16x32_fp8_fp8 with power-of-two amax scales per (sample, head) (csrc/attention8.hip).  Replaces the same
 * reference code as a3d_attn16_fwd (multihead_custom_attention.py:386-447) at e4m3's tolerance, NOT at the 1e-3 parity bar.
 * Takes the "16" operands of a3d_attn16_fwd (Qr, Kr rows16; Vp two-part planes16 with the ones channel) and derives its own:
 * ops8 = a3d_attn8_operand_bytes(B, H, Sp) bytes of scratch, 256-byte aligned (K8 rows, V8 planes, amax words), rewritten
 * by every call (amax reduction + pack kernel + forward: three launches).  No dropout variant.  O, LSE2, ws as
 * a3d_attn16_fwd, so a3d_attn16_bwd can consume them. */
size_t a3d_attn8_operand_bytes(int B, int H, int Sp);
int a3d_attn8_fwd(const void* Qr, const void* Kr, const void* Vp, void* ops8, const unsigned char* kmask, float* O,
                  float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, void* stream);

/* a3d_attn_fwd / a3d_attn_bwd_bf16 with dropout on the attention weights: O = (keep o softmax(..) / (1 - p)) V. */
int a3d_attn_fwd_dropout(const void* Qs, const void* Ks, const void* Vt, const unsigned char* kmask, float* O, float* LSE,
                         float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                         const unsigned long long* drop_state, unsigned int site, float p, void* stream);
int a3d_attn_bwd_bf16_dropout(const void* Qs, const void* Qt, const void* Ks, const void* Kt, const void* Vs,
                              const unsigned char* kmask, const float* O, const float* dO, const float* LSE, void* dOs,
                              void* dOt, float* D, float* dQp, float* dK, float* dV, int B, int H, int Lq, int Lqp, int S,
                              int Sp, int nsplit, const unsigned long long* drop_state, unsigned int site, float p,
                              void* stream);

/* ---- attention of ONE query per sample over the context (Act3D's query stream, act3d.py:467-480), keys / values never
 *      materialised: o[b][h*15+d] = W_v[h*15+d] . xbar[b][h] + b_v with xbar = softmax-weighted mean of the raw context rows
 *      X [B][S][E] (16-byte aligned); the key projection W_k x + b_k (rows E..2E of in_proj), rotated by xyz (NULL: none), is
 *      recomputed per tile.  qrot: [B][H][16] rotated, scaled query (a3d_rope_rows_f32 with N = Npad = 1).  E = 15 H <= 60.
 *      Outputs xbar [B][H][E], lse [B][H] (saved for backward), o [B][E] (o NULL: skipped -- the fused layer kernel
 *      a3d_qs_post_fwd projects the values).  ws >= a3d_sq_fwd_ws_floats floats. ------------- */
size_t a3d_sq_fwd_ws_floats(int B, int H, int E, int nsplit);
int a3d_sq_attn_fwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv, int ldwv,
                    const float* bv, const float* qrot, const float* freq, float* ws, float* xbar, float* lse, float* o, int B,
                    int S, int E, int H, int nsplit, void* stream);
/* Backward: dX [B][S][E] (written), dqp [nsplit][B][H][16] (rotated-query gradient partials: a3d_rope_merge_bwd layout with
 * Npad = 1), dWk / dbk / dWv / dbv accumulated (+=).  ws >= a3d_sq_bwd_ws_floats floats = dxbar [B][H][E] | cD [B][H] | weight
 * gradient partials; dO NULL: the caller (a3d_qs_post_bwd) has already written dxbar and cD there and owns dWv / dbv. */
size_t a3d_sq_bwd_ws_floats(int B, int H, int E, int nsplit);
int a3d_sq_attn_bwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv, int ldwv,
                    const float* qrot, const float* freq, const float* xbar, const float* lse, const float* dO, float* ws,
                    float* dX, float* dqp, float* dWk, int lddwk, float* dbk, float* dWv, int lddwv, float* dbv, int B, int S,
                    int E, int H, int nsplit, void* stream);
/* the same with dX += instead of dX = when accumulate_dX != 0 (the context's gradient summed in place by its consumers) */
int a3d_sq_attn_bwd_acc(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv, int ldwv,
                        const float* qrot, const float* freq, const float* xbar, const float* lse, const float* dO, float* ws,
                        float* dX, float* dqp, float* dWk, int lddwk, float* dbk, float* dWv, int lddwv, float* dbv, int B, int S,
                        int E, int H, int nsplit, int accumulate_dX, void* stream);
/* dW[n][k] += sum_z partial[z][n][k], db[n] += sum_z partial[z][n][E] for partial [nsplit][E][E + 1] (fixed order) */
int a3d_sq_wgrad_reduce(const float* partial, int nsplit, float* dW, int lddw, float* db, int E, void* stream);

/* ---- scene tokens --------------------------------------------------------------------------------------- */
/* out[b][(cam*h + y)*w + x][:] = bilinear(pcd[(b,cam)], 1/factor)  (act3d.py:379-383, encoder.py:147-158) */
int a3d_pcd_downsample(const float* pcd, float* out_xyz, int B, int C, int Hin, int Win, int factor, void* stream);
/* idx_out[b][:k] = indices of the k nearest points of xyz[b] to pos[b], ascending (distance, index)
 * (act3d.py:244-245).  ws >= a3d_knn_topk_ws_bytes.  dist_out optional [B][k]. */
size_t a3d_knn_topk_ws_bytes(int B, int N);
int a3d_knn_topk(const float* pos, const float* xyz, void* ws, long long* idx_out, float* dist_out, int B, int N,
                 int k, void* stream);
/* find_traj_nn (model/utils/utils.py:39-48) of the multi-scale diffusion head: the k scene points with the smallest SQUARED
 * distance to the nearest of the L trajectory points traj_xyz [B][L][3]; same (distance, index) order and workspace as above. */
int a3d_traj_nn_topk(const float* traj_xyz, int L, const float* xyz, void* ws, long long* idx_out, float* dist_out, int B,
                     int N, int k, void* stream);
/* ctx[b] = [ feat[b][idx[b][0..k)] | extra[b][0..X) ], rows of W floats (idx NULL: identity, k == Npts)
 * (act3d.py:247-260). */
int a3d_build_context(const float* feat, const long long* idx, const float* extra, float* ctx, int B, int Npts,
                      int k, int X, int W, void* stream);
int a3d_build_context_bwd(const float* dctx, const long long* idx, float* dfeat, float* dextra, int B, int Npts,
                          int k, int X, int W, int accumulate, void* stream);

/* Same with bf16 token rows: `feat` is the FPN's bf16 channels-last output read in place ([B][Npts][ldf] bf16 with the
 * W token channels first -- ldf = 64 when the FPN runs channel-padded for MIOpen -- 8-byte aligned, W, ldf % 4 == 0);
 * ctx / extra stay fp32; the backward accumulates into a bf16 gradient map of the same layout (zero-initialised by the
 * caller and shared by every level that gathers from the map; pad channels are never written).
 * `bias` (fp32 [W] or NULL) is added to the gathered rows -- the bias of the FPN's 3x3 output convolution applied to the rows a
 * level reads instead of to the whole map; its gradient is a3d_colsum_rows of d(ctx). */
int a3d_build_context_bf16(const void* feat, int ldf, const long long* idx, const float* extra, const float* bias, float* ctx,
                           int B, int Npts, int k, int X, int W, void* stream);
int a3d_build_context_bwd_bf16(const float* dctx, const long long* idx, void* dfeat, int ldf, float* dextra, int B,
                               int Npts, int k, int X, int W, int accumulate, void* stream);
/* Token-sparse WEIGHT gradient of the FPN's 3x3 output convolution (torchvision FeaturePyramidNetwork.layer_blocks, stride 1,
 * padding 1, 64 -> 64 channels on the bf16 NHWC path; act3d.py:76-77) for a map that is read only through a3d_build_context_bf16's
 * gathers (act3d.py:244-260): dW[tap][co][ci] (+)= sum over b, j < k of G[b][j][co] X[pixel(idx[b][j]) + offset(tap)][ci], computed
 * from the gather's own backward inputs (idx and the fp32 context-gradient rows G [B][g_rows][E], g_rows >= k, E <= 64) -- 12 % of
 * the dense library kernel's work and no dense gradient map.  X: the convolution's INPUT, bf16 [B ncam][H][W][64]; dW fp32
 * [3][3][64 co][64 ci] (tap-major; the caller permutes it to the weight's layout); ws: a3d_conv3x3_wgrad_tokens_ws_floats() floats.
 * Replaces the weight-gradient half of F.conv2d's autograd for that layer (csrc/fpn_sparse.hip). */
size_t a3d_conv3x3_wgrad_tokens_ws_floats(void);
int a3d_conv3x3_wgrad_tokens(const void* X, const long long* idx, const float* G, int g_rows, int E, float* ws, float* dW,
                             int accumulate, int B, int k, int ncam, int H, int W, void* stream);
/* Token-sparse INPUT gradient of the same convolution (the library's dense igemm_bwd over a gradient map that is non-zero on 6 - 12 %
 * of its pixels: 0.65 ms per step at the bench shape).  The output gradient dy (bf16 NHWC [B ncam][H][W][64], scattered into by
 * a3d_build_context_bwd_bf16) is non-zero only on gathered pixels, so dx is non-zero only on their 3x3 neighbourhoods:
 * a3d_conv3x3_mark_tiles marks, per gathered token, the 8 x 32-pixel output tiles its neighbourhood touches (mask: one byte per tile,
 * a3d_conv3x3_tile_count(B ncam, H, W) bytes, zeroed by the caller before the first gather of a backward pass marks it);
 * a3d_conv3x3_dgrad_tiles compacts the marks into a tile list and runs the implicit-GEMM stream kernel of a3d_conv3x3_bn_fwd over the
 * marked tiles only, writing zeros to the others: dx = conv3x3(dy, wt), wt bf16 [64 ci][3][3][64 co] with
 * wt[ci][kh][kw][co] = w[co][ci][2 - kh][2 - kw].  ws: a3d_conv3x3_dgrad_tiles_ws_ints(..) ints.  Replaces the input-gradient half of
 * F.conv2d's autograd for that layer (torchvision FeaturePyramidNetwork.layer_blocks; csrc/conv3x3.hip). */
size_t a3d_conv3x3_tile_count(size_t images, int H, int W);
size_t a3d_conv3x3_dgrad_tiles_ws_ints(size_t images, int H, int W);
int a3d_conv3x3_mark_tiles(const long long* idx, int B, int k, int ncam, int H, int W, unsigned char* mask, void* stream);
int a3d_conv3x3_dgrad_tiles(const void* dy, const void* wt, const unsigned char* mask, int* ws, void* dx, size_t images, int H, int W,
                            void* stream);
/* out[c] += sum over b < B, s < k of src[b][s][c]  (src fp32 [B][S][ld], c < nout <= C <= 64; two launches, fixed summation
 * order; ws: a3d_colsum_rows_ws_floats(B, k, C) floats). */
size_t a3d_colsum_rows_ws_floats(int B, int k, int C);
int a3d_colsum_rows(const float* src, int B, int S, int k, int ld, int C, float* out, int nout, float* ws, void* stream);

/* ---- decoding heads, losses, sampler, optimizer ----------------------------------------------------------- */
int a3d_mask_logits_fwd(const float* q, const float* F, float* out, int B, int Ng, int E, void* stream);
int a3d_mask_logits_bwd(const float* q, const float* F, const float* dlog, float* dF, float* dq, int B, int Ng,
                        int E, int accumulate_dF, void* stream);
int a3d_argmax_gather(const float* logits, const float* ghost, long long* top_idx, float* pos, int B, int Ng,
                      void* stream);
/* loss = coeff * mean_b CE(logits[b], softmax(-|ghost[b]-gt[b]|/spread)); dlogits optional. */
int a3d_soft_ce_loss(const float* ghost, const float* gt, const float* logits, float* loss_b, float* loss,
                     float* dlogits, int B, int Ng, float spread, float label_smoothing, float coeff, void* stream);
/* kind 0 = MSE, 1 = L1; loss = coeff * mean; grad optional. */
int a3d_elem_loss(const float* pred, const float* target, int n, int kind, float coeff, float* loss, float* grad,
                  void* stream);
int a3d_scale_by_scalar(const float* x, const float* scalar, float* y, size_t n, void* stream);
int a3d_quat_sigmoid_fwd(const float* pred, float* rot, float* grip, int B, void* stream);
int a3d_quat_sigmoid_bwd(const float* pred, const float* drot, const float* dgrip, float* dpred, int B, void* stream);
/* The 6D_* rotation heads (act3d.py:529-533; compute_rotation_matrix_from_ortho6d, model/utils/utils.py:93-130):
 * pred [B][7] -> rot [B][3][3] (columns x, y, z of the Gram-Schmidt frame), grip [B][1] = sigmoid(pred[6]). */
int a3d_ortho6d_sigmoid_fwd(const float* pred, float* rot, float* grip, int B, void* stream);
int a3d_ortho6d_sigmoid_bwd(const float* pred, const float* drot, const float* dgrip, float* dpred, int B, void* stream);
/* y[b] = x[b][idx[b]] for x [B][N][W] (the top ghost point's offset / feature row, act3d.py:513-522); the backward
 * writes the full [B][N][W] gradient (zero outside the selected rows). */
int a3d_select_row_fwd(const float* x, const long long* idx, float* y, int B, int N, int W, void* stream);
int a3d_select_row_bwd(const float* dy, const long long* idx, float* dx, int B, int N, int W, void* stream);
/* state = {seed, offset} (2 x uint64, device).  anchor NULL: uniform box (utils.py:68-73); else ball rejection
 * inside the clipped box (utils.py:76-84, act3d.py:417-436) with a bounded number of attempts. */
int a3d_sample_ghost_points(const unsigned long long* state, const float* bounds, const float* anchor, float radius,
                            float* out, int B, int Ng, int level, int max_attempts, void* stream);
int a3d_rng_advance(unsigned long long* state, unsigned long long n, void* stream);
void a3d_philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Host mirror of the device RoPE sin / cos (a3d_common.h fast_sincos: sincos_poly, behind a double-precision reduction for |x| >= 200; the rotary code of
 * position_encodings.py:64-97 evaluated in-kernel): the same fp32 operations, so a CPU test can bound it against float64. */
void a3d_sincos_host(const float* x, float* sn, float* cs, size_t n);
/* torch.optim.AdamW semantics (engine.py:89-102) on a flat buffer; elements [0, n_nodecay) use wd_nodecay.  The decision
 * "this parameter has no gradient -> skip it" is per PARAMETER, as in torch: seg_off [nseg + 1] (element offsets, ascending,
 * seg_off[0] = 0, seg_off[nseg] = n) delimits the parameters, seg_state [nseg][4] = {step, active, lr / bc1, sqrt(bc2)} is
 * the per-parameter step count + scratch (zero-initialised by the caller; step = torch's state[p]["step"]).  A parameter
 * joins the update from the first call in which its gradient segment has a non-zero element.  step[0] counts calls. */
int a3d_adamw_step(float* p, const float* g, float* m, float* v, float* step, const long long* seg_off, float* seg_state,
                   int nseg, size_t n, size_t n_nodecay, float lr, float beta1, float beta2, float eps, float wd_nodecay,
                   float wd_decay, float grad_scale, void* stream);

/* ---- Act3D query stream, fused per layer (csrc/query_stream.hip; act3d.py:467-480, layers.py:293-351) -------------------------
 * One RelativeCrossAttentionLayer + FeedforwardLayer of the ONE-query-per-sample stream as four launches around a3d_sq_attn_fwd /
 * a3d_sq_attn_bwd's key-streaming kernels.  E <= 60, E % 12 == 0, H = E / 15 <= 4, FFN hidden = E; all matrices row-major with
 * row stride E (views into in_proj_weight etc.), fp32. */
typedef struct {
  const float *wv, *bv;            /* value rows [2E, 3E) of in_proj_weight / in_proj_bias */
  const float *wo, *bo;            /* out_proj */
  const float *g1, *b1;            /* RelativeCrossAttentionLayer.norm */
  const float *w1, *c1, *w2, *c2;  /* FeedforwardLayer.linear1 / linear2 (weight, bias) */
  const float *g2, *b2;            /* FeedforwardLayer.norm */
} a3d_qs_params;
typedef struct {                   /* gradient buffers of the same parameters, accumulated into (+=) */
  float *dwv, *dbv, *dwo, *dbo, *dg1, *db1, *dw1, *dc1, *dw2, *dc2, *dg2, *db2;
} a3d_qs_grads;
/* qrot [B][H][16] = RoPE((W_q x + b_q) * scale, xyz): the rotated, scaled query a3d_sq_attn_fwd takes (xyz NULL: no rotation) */
int a3d_qs_pre_fwd(const float* x, const float* wq, const float* bq, const float* xyz, const float* freq, float scale, float* qrot,
                   int B, int E, int H, void* stream);
/* from a3d_sq_attn_bwd's dqp [nsplit][B][H][16]: dW_q += dq^T x, db_q += sum dq, dx += dq W_q  (dx [B][E] in / out) */
int a3d_qs_pre_bwd(const float* dqp, int nsplit, const float* xyz, const float* freq, float scale, const float* x, const float* wq,
                   float* dwq, float* dbq, float* dx, int B, int E, int H, void* stream);
/* xbar [B][H][E] (a3d_sq_attn_fwd) + resid [B][E] (the layer's input) -> y [B][E] = the layer's output; `save`
 * (a3d_qs_save_floats(B, E) floats) keeps o | Y | y1 | h | o2 | LayerNorm statistics per row for the backward */
size_t a3d_qs_save_floats(int B, int E);
int a3d_qs_post_fwd(const float* xbar, const float* resid, const a3d_qs_params* p, float* save, float* y, int B, int E, int H,
                    void* stream);
/* dy [B][E] -> every parameter gradient of `p` (+=), dxbar [B][H][E] and cD [B][H] (= dxbar . xbar) for a3d_sq_attn_bwd's key
 * pass, dresid [B][E] = the gradient of the residual branch w.r.t. the layer input */
int a3d_qs_post_bwd(const float* dy, const float* resid, const float* xbar, const float* save, const a3d_qs_params* p,
                    const a3d_qs_grads* gr, float* dxbar, float* cD, float* dresid, int B, int E, int H, void* stream);

/* ---- DDPM trajectory denoiser (elementwise pieces) ------------------------------------------------------------ */
/* x_t = sqrt(acp[t_b]) x0 + sqrt(1 - acp[t_b]) eps; channels [0,npos) use acp_pos, the rest acp_rot
 * (DDPMScheduler.add_noise; diffusion_model.py:296-305).  t: [B] int64. */
int a3d_ddpm_add_noise(const float* x0, const float* noise, const long long* t, const float* acp_pos,
                       const float* acp_rot, float* out, int B, int L, int D, int npos, void* stream);
/* One reverse step: inpaint (model_out[mask] = cond[mask]), then t == 0 ? out = model_out
 * : out = c_x0[t]*clip(model_out,-1,1) + c_xt[t]*sample + sigma[t]*noise.  coef_*: [T][3] device tables
 * (diffusion_model.py:100-117 + DDPMScheduler.step, prediction_type="sample", fixed_small). */
int a3d_ddpm_step(const float* model_out, const float* sample, const float* noise, const float* cond_data,
                  const unsigned char* cond_mask, const float* coef_pos, const float* coef_rot, float* out, int rows,
                  int D, int npos, int t, void* stream);
/* AdaLN: y = x * (1 + mod[:, :E]) + mod[:, E:]  (layers.py:273-290); mod: [B][2E], x: [B][L][E]. */
int a3d_adaln_fwd(const float* x, const float* mod, float* y, int B, int L, int E, void* stream);
int a3d_adaln_bwd(const float* x, const float* mod, const float* dy, float* dx, float* dmod, int B, int L, int E,
                  void* stream);
/* out[i] = [sin(x_i f_j) | cos(x_i f_j)], f_j = exp(-j ln(1e4)/(E/2-1))  (position_encodings.py:13-20). */
int a3d_sinusoidal_emb(const float* x, float* out, int n, int E, void* stream);
int a3d_silu_fwd(const float* x, float* y, size_t n, void* stream);
int a3d_silu_bwd(const float* x, const float* dy, float* dx, size_t n, void* stream);
/* y[b,l,:] = x[b,l,:] + r[l,:] */
int a3d_add_rows(const float* x, const float* r, float* y, int B, int L, int E, void* stream);
/* dr [L][E] = sum over the batch of dy [B][L][E] (gradient of the shared rows, e.g. instr_position_embedding, act3d.py:201-209) */
int a3d_add_rows_bwd(const float* dy, float* dr, int B, int L, int E, void* stream);
/* out = cat(traj[..., :npos] + upd[..., :npos], upd[..., npos:])  (diffusion_head.py:268-272) */
int a3d_traj_update(const float* traj, const float* upd, float* out, int rows, int D, int npos, void* stream);

/* out[i] = [ (xyz - lo) / (hi - lo) * 2 - 1 | first two columns of R(normalise(quat)) | extra.. ] for pose rows
 * [xyz | quat (w, x, y, z) | extra..] (diffusion_model.py:187-212: normalize_pos + convert_rot, rotation_parametrization "6D");
 * bounds: device [2][3], NULL = leave xyz as is.  a3d_signal_to_pose is the inverse map (unconvert_rot + unnormalize_pos, :192-195,214-230):
 * Gram-Schmidt of the 6D part, quaternion from the best-conditioned trace identity. */
int a3d_pose_to_signal(const float* pose, const float* bounds, float* out, int n, int extra, void* stream);
int a3d_signal_to_pose(const float* signal, const float* bounds, float* out, int n, int extra, void* stream);
/* cols[b][9] of TrajectoryCriterion.compute_metrics (main_trajectory.py:303-343), see diffusion.hip; pred, gt: [B][L][D], D >= 7. */
int a3d_traj_errors(const float* pred, const float* gt, float* cols, int B, int L, int D, void* stream);
/* cols[b][6 + nlev] of LossAndMetrics.compute_metrics (main_keypose.py:431-482), see heads.hip; pos: [nlev + 1][B][3] with
 * slot 0 = the final position, gt rows of ldgt >= 8 floats [xyz | quat | open]. */
int a3d_keypose_errors(const float* pos, const float* rot, const float* grip, const float* gt, int ldgt, float* cols, int B,
                       int nlev, int symmetric, void* stream);
/* coeff * mean_b min(mse(q_b, g_b), mse(q_b, -g_b))  (symmetric_rotation_loss, main_keypose.py:370-376); grad optional [B][4]. */
int a3d_sym_quat_loss(const float* q, const float* gt, int ldgt, float coeff, float* loss, float* grad, int B, void* stream);

/* ---- fused denoising-network evaluation for the sampling loop (inference; diffusion_head.py:200-363 per step of
 *      diffusion_model.py:86-119).  L <= 16 trajectory steps per sample, E = 15 H <= 128.  One step = a3d_dn_head, then per
 *      ParallelAttentionLayer a3d_dn_cross + a3d_dn_rest, then a3d_dn_tail (18 launches instead of ~200).
 *      Weights are the reference's tensors, row-major fp32 ([out][in]); every pointer is a device pointer. ------------- */
typedef struct {
  const float *enc_w0, *enc_b0, *enc_w1, *enc_b1;   /* traj_encoder: Linear(D, E) - ReLU - Linear(E, E)  (diffusion_head.py:41-46) */
  const float* sem;                                 /* [L][E] sinusoidal step-index embedding (position_encodings.py:13-20) */
  const float* lang_kv;                             /* [B][S_lang][2E] = instruction tokens through the packed k | v projection
                                                       of traj_lang_attention; NULL: no instruction branch */
  int S_lang;
  const float *q_w, *q_b, *out_w, *out_b, *ln_g, *ln_b;   /* traj_lang_attention cross_12 (rows 0..E-1 of in_proj) / norm_12 */
} a3d_dn_head_params;
typedef struct {
  const float* sem;          /* [L][E] or NULL */
  const float* mod;          /* [2E] AdaLN (scale | shift) of this layer at this timestep, NULL: no AdaLN */
  const float *q_w, *q_b;    /* rows 0..E-1 of cross_12.in_proj_weight / bias */
  const float* freq;         /* E/6 RoPE frequencies, NULL: no rotation */
  const void* Kf;            /* context keys, projected + rotated: rows16 [B][H][Sp][32] fp16 = hi(16) | lo(16) (round 6; was fp32 rows) */
  const unsigned short* Vt;  /* context values: planes16 [B][H][2][16][Sp] fp16 hi / lo, 1.0 in channel 15 of the hi plane
                              * (both written by ONE a3d_proj_rope_split16 launch: k block rows, parts 2; v block planes, parts 2 | 4) */
} a3d_dn_cross_params;
typedef struct {
  const float *c_out_w, *c_out_b, *c_ln_g, *c_ln_b;                       /* cross_12.out_proj, norm_12 */
  const float* sem;
  const float *s_mod, *s_in_w, *s_in_b, *s_out_w, *s_out_b, *s_ln_g, *s_ln_b;   /* adaln_1 (this step), sa1, norm_1; s_in_w NULL: none */
  const float* freq;
  const unsigned char* kmask;                                              /* [B][L] 1 = padded step, or NULL */
  const float *f_mod, *f_w1, *f_b1, *f_w2, *f_b2, *f_ln_g, *f_ln_b;         /* adaln_ff1 (this step), ffn_12, norm_122; f_w1 NULL: none */
  int F;                                                                   /* FFN hidden width (4E) */
} a3d_dn_rest_params;
typedef struct {
  const float *pos_w0, *pos_b0, *pos_w1, *pos_b1, *rot_w0, *rot_b0, *rot_w1, *rot_b1;   /* regressors (diffusion_head.py:177-199) */
  const float *noise, *cond_data;            /* [B][L][D]; noise NULL at t = 0 */
  const unsigned char* cond_mask;            /* [B][L][D] or NULL */
  const float *coef_pos, *coef_rot;          /* [T][3] posterior tables (see a3d_ddpm_step) */
} a3d_dn_tail_params;
int a3d_dn_head(const float* traj, int D, const a3d_dn_head_params* p, float* x_out, int B, int L, int E, int H, void* stream);
size_t a3d_dn_cross_ws_floats(int B, int H, int nsplit);
/* partial attention outputs of every (sample, head, key split) into ws (>= a3d_dn_cross_ws_floats floats) */
int a3d_dn_cross(const float* x, const float* traj, int D, const a3d_dn_cross_params* p, float* ws, int B, int L, int E,
                 int H, int S, int Sp, int nsplit, void* stream);
/* combine + out-proj + LayerNorm, self-attention block, FFN block of one layer: x_in -> x_out ([B][L][E]) */
int a3d_dn_rest(const float* x_in, const float* traj, int D, const float* ws, const a3d_dn_rest_params* p, float* x_out,
                int B, int L, int E, int H, int nsplit, void* stream);
/* regressors + trajectory update + DDPM reverse step t_step: traj ([B][L][D]) -> traj_out */
int a3d_dn_tail(const float* pos_feats, const float* rot_feats, const float* traj, int D, const a3d_dn_tail_params* p,
                float* traj_out, int B, int L, int E, int t_step, void* stream);
/* ---- persistent sampler: nsteps consecutive denoise steps t_first, t_first - 1, ... of one trajectory batch in ONE launch
 *      (diffusion_model.py:86-119: the loop body `out = prediction_head(...); trajectory = scheduler.step(...)`).  Two workgroup
 *      roles: TWO workgroups per (trajectory, 16-step row tile) run that unit's chain across layers AND steps -- the primary: head,
 *      trajectory stack, position stack, tail + DDPM step; the helper: the rotation stack, concurrently with the position stack (x
 *      handed over through xbuf); the row tiles of a trajectory exchange their self-attention keys / values through kvx --; the
 *      remaining CUs stream the cached context K / V for whichever (unit, layer) is ready.
 *      L <= 64 trajectory steps (scripts/train_trajectory.sh:7-8 and online_evaluation/eval.sh:17 use interpolation_length 50).
 *      layers_dev: DEVICE array of n_traj + n_pos + n_rot entries in stack order (trajectory, position, rotation; diffusion_head.py:
 *      343-357), whose cross.mod / rest.s_mod / rest.f_mod are the BASES of the [T][2E] AdaLN tables (row t is used at step t), and
 *      tail->noise is the BASE of the [T][B][L][D] step noise.  traj is updated in place.  With NT = ceil(L / 16): qbuf: 2 * B * NT *
 *      16 * 128 floats; part: a3d_dn_cross_ws_floats(2 * B * NT, H, a3d_dn_persist_splits(H, nsplit)) floats; kvx:
 *      a3d_dn_persist_kvx_floats floats (NULL when L <= 16); xbuf: a3d_dn_persist_xbuf_floats floats; sync: a3d_dn_persist_sync_ints(..)
 *      ints (zeroed by the call; sync[2] != 0 afterwards = a wait exceeded its bound and the launch gave up: the trajectory is then
 *      invalid).  Needs 2 * B * NT + 16 <= the device's CU count (all workgroups are co-resident); otherwise A3D_ERR_ARG -> per-phase
 *      entry points (L <= 16) or a smaller batch per call. */
typedef struct {
  a3d_dn_cross_params cross;
  a3d_dn_rest_params rest;
} a3d_dn_layer_params;
int a3d_dn_persist_splits(int H, int nsplit);
size_t a3d_dn_persist_kvx_floats(int B, int L, int E);
size_t a3d_dn_persist_xbuf_floats(int B, int L);
/* development aid: 256 phase timestamps (100 MHz ticks) of the last a3d_dn_persist launch under A3D_DN_PROF=1 (host buffer) */
int a3d_dn_persist_prof(const int* sync, int B, int L, int n_layers, int nsteps, long long* out256);
size_t a3d_dn_persist_sync_ints(int B, int L, int n_layers, int nsteps);
int a3d_dn_persist(const a3d_dn_layer_params* layers_dev, int n_traj, int n_pos, int n_rot, const a3d_dn_head_params* head,
                   const a3d_dn_tail_params* tail, float* traj, float* qbuf, float* part, float* kvx, float* xbuf, int* sync, int B,
                   int L, int D, int E, int H, int S, int Sp, int nsplit, int t_first, int nsteps, void* stream);
/* development aid: 18 phase timestamps (100 MHz ticks) of workgroup 0 of the last a3d_dn_rest launch under A3D_DN_PROF=1 (host buffer) */
int a3d_dbg_dn_prof(long long* out18);
/* out[b][h][n][16] fp32 = rope3d(Y[b, n, :E] * scale, xyz) split into heads (column 15 and rows >= N zero): the K cache */
int a3d_rope_rows_f32(const float* Y, int ldy, const float* xyz, const float* freq, float scale, float* out, int B, int N,
                      int Npad, int E, int H, void* stream);

/* ---- frozen-backbone BatchNorm (train-mode statistics) + ReLU + residual, bf16 NHWC -------------- */
/* x, residual, y: bf16, `rows` = N*H*W rows of C channels (torch channels_last storage).  C = 8 * divisor of 256. */
int a3d_bn_nslab(size_t rows, int C);
int a3d_bn_stats(const void* x, float* partial /* [nslab][2][C] */, size_t rows, int C, int nslab, void* stream);
/* train: batch mean / biased variance from `partial`, running stats updated with `momentum` (unbiased variance), as
 * nn.BatchNorm2d in train(); else the running statistics are used.  scale = gamma / sqrt(var + eps), shift = beta - mean*scale. */
int a3d_bn_finalize(const float* partial, int nslab, size_t rows, int C, float eps, float momentum, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float* scale, float* shift, int train,
                    void* stream);
/* y = relu?(x * scale[c] + shift[c] (+ residual)); res_scale / res_shift (or NULL): the residual is a raw convolution output with
 * its own BatchNorm (the bottleneck's downsample branch, clip.py:28-43) and enters as residual * res_scale[c] + res_shift[c] */
int a3d_bn_apply(const void* x, const void* residual, const float* res_scale, const float* res_shift, const float* scale,
                 const float* shift, void* y, size_t rows, int C, int relu, void* stream);
/* Same, followed by the nn.AvgPool2d(2) that the CLIP bottleneck / stem applies to the activation (model/utils/clip.py
 * Bottleneck.avgpool, downsample[0], ModifiedResNet.avgpool): y_pool [N][H/2][W/2][C] = mean of the 2x2 bf16 activations;
 * y_full [N][H][W][C] is also written unless NULL.  scale == NULL: identity (plain average pool of x).  H, W even. */
int a3d_bn_apply_pool2(const void* x, const void* residual, const float* scale, const float* shift, void* y_full,
                       void* y_pool, int N, int H, int W, int C, int relu, void* stream);
/* Workgroups of a3d_bn_apply / a3d_bn_apply_pool2 (grid-stride loops; default 16384, A3D_BN_GRID): cap > 0 sets the cap (at least 64)
 * and returns the previous one, cap <= 0 only queries.  A capture of the frozen backbone that is to run BESIDE other kernels
 * (engine.GraphedStep(prefetch=...)) uses one workgroup per CU so that the concurrent stream finds free wave slots; grids are
 * baked into a captured graph, so the setting only matters while launches are being issued.  (clip.py:28-43 BatchNorm passes.) */
int a3d_bn_grid_cap(int cap);
/* FPN top-down step, bf16 NHWC, exact 2x: y = lat + bias + nearest_up2(top)  (torchvision FeaturePyramidNetwork.forward as the
 * reference instantiates it, act3d.py:60-66, with the lateral 1x1 convolution's bias folded in: the convolution runs
 * bias-free; bias fp32 [nbias <= C] or NULL, channels >= nbias are MIOpen's zero padding; top NULL at the pyramid's top level).  Backward: dtop = 2x2 block sums of dy (dlat = dy;
 * dtop NULL at the top level), dbias[c] += column sums of dy when dbias != NULL (ws: a3d_upsample2_add_bwd_ws_floats floats,
 * C / 4 must divide 256; two-stage, fixed order).  H, W: the fine size, even. */
int a3d_upsample2_add_fwd(const void* lat, const void* top, const float* bias, int nbias, void* y, int N, int H, int W, int C,
                          void* stream);
size_t a3d_upsample2_add_bwd_ws_floats(int N, int H, int W, int C);
int a3d_upsample2_add_bwd(const void* dy, void* dtop, float* dbias, int nbias, float* ws, int N, int H, int W, int C,
                          void* stream);

/* y (bf16, [N][H][W][3] = torch channels_last storage) = (x (fp32 [N][3][H][W]) - mean[c]) / std[c]: CLIP's input
 * normalisation (model/utils/clip.py:19, act3d.py:364) fused with the layout change and the cast the bf16 backbone needs. */
int a3d_rgb_normalize_nhwc_bf16(const float* x, const float* mean, const float* stdv, void* y, size_t N, int H, int W,
                                void* stream);

/* ---- diagnostics ---------------------------------------------------------------------------------------------- */
int a3d_dbg_mfma_bf16(const void* A16x32, const void* B32x16, float* D16x16, void* stream);
int a3d_dbg_mfma_f32(const float* A16x4, const float* B4x16, float* D16x16, void* stream);
/* out[i] = v_cvt_pk_bf16_f32(in[2i], in[2i+1]) -- pins the rounding mode the attention kernel relies on (RNE). */
int a3d_dbg_cvt_pk_bf16(const float* in, void* out, int npairs, void* stream);

/* ---- frozen backbone: 1x1 convolutions as a bf16 MFMA GEMM with the neighbouring BatchNorm work folded in ---------------
 * (CLIP ModifiedResNet bottleneck conv1 / conv3 / downsample, model/utils/clip.py:28-43.)
 * y [M][N] bf16 = f(x [M][K] bf16) w[N][K]^T with f(x) = relu?(x * in_scale[k] + in_shift[k]) (the producer's BatchNorm-apply;
 * in_scale NULL: identity); partial (or NULL): [a3d_conv1x1_nslab(M, K, N)][2][N] per-workgroup (sum, sum of squares) of the
 * rounded outputs = the input a3d_bn_finalize expects for the BatchNorm that follows.
 * Served shapes (a3d_conv1x1_streams(K, N) == 1): K in {64, 128, 256}, N in {64, 128, 256 j}, weight block + activation buffers
 * within 96 KB of LDS -- the HBM-bound layers 1-2 of the ResNet, where the fusion pays; other shapes are refused (A3D_ERR_ARG,
 * a3d_conv1x1_nslab 0): the compute-bound deep layers stay on MIOpen.  Operands 16-byte aligned. */
int a3d_conv1x1_streams(int K, int N);
int a3d_conv1x1_nslab(size_t M, int K, int N);
int a3d_conv1x1_bn_fwd(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu, void* y,
                       float* partial, size_t M, int K, int N, void* stream);
/* The FPN's lateral 1x1 convolution with its bias and the top-down add in the epilogue (torchvision FeaturePyramidNetwork.forward:
 * inner_lateral = inner_blocks[i](x); last_inner = inner_lateral + F.interpolate(last_inner, nearest) -- act3d.py:76-77, :363-369):
 *   y[n][h][w][:] = bf16( x[n][h][w][:] w^T + bias[:] + top[n][h/2][w/2][:] ),  ONE rounding of the fp32 sum
 * x [images][H][W][K] bf16 (NHWC), w [N][K] bf16, bias fp32 (the first nbias <= N channels; NULL: none), top [images][H/2][W/2][N] bf16
 * (NULL: the pyramid's top level), y [images][H][W][N] bf16.  The resident-weight streaming kernel of a3d_conv1x1_bn_fwd with another
 * epilogue: the lateral map is never written and re-read by a3d_upsample2_add_fwd.  Served (a3d_conv1x1_topdown_serves == 1): K in
 * {64, 128, 256}, N in {64, 128}; H and W even with a top map; at most 2^32 rows; operands 16-byte aligned.  The backward is
 * a3d_upsample2_add_bwd's (d top, d bias) plus the convolution's weight gradient. */
int a3d_conv1x1_topdown_serves(int K, int N);
int a3d_conv1x1_topdown_fwd(const void* x, const void* w, const float* bias, int nbias, const void* top, void* y, size_t images,
                            int H, int W, int K, int N, void* stream);
/* Whether the deep-layer GEMM of a3d_conv1x1_bn_fwd (K = 64 j in 128 .. 2048, N = 128 j up to 2048: the 1x1 convolutions of CLIP
 * ModifiedResNet layers 2 - 4, model/utils/clip.py:28-43) takes its shapes: 1 yes (default; A3D_CONV1X1_DEEP), 0 they stay with the
 * library.  Sets the mode and returns the previous one; mode < 0 only queries.  Affects a3d_conv1x1_streams / _nslab / _bn_fwd alike. */
int a3d_conv1x1_deep_mode(int mode);
/* The stem's first convolution (CLIP ModifiedResNet conv1, model/utils/clip.py:22-43: nn.Conv2d(3, 32, 3, stride=2, padding=1, bias=False))
 * on the RAW images with CLIP's normalisation (act3d.py:365) in front of it and the statistics of the BatchNorm behind it folded in:
 * y = bf16(conv(bf16((rgb - mean) / std), w)), zero padding of the NORMALISED map (what F.conv2d does to the map
 * a3d_rgb_normalize_nhwc_bf16 writes).  rgb fp32 planar [N][3][H][W]; w bf16 [32][27] (the torch weight [co][ci][kh][kw]); y bf16 NHWC
 * [N][H/2][W/2][32]; partial (or NULL): [a3d_stem_conv_nslab(..)][2][32] per-workgroup (sum, sum of squares) of the rounded outputs
 * for a3d_bn_finalize.  H a multiple of 16, W of 64 (a3d_stem_conv_nslab returns 0 otherwise).  csrc/stem.hip. */
int a3d_stem_conv_nslab(size_t images, int H, int W);
int a3d_stem_conv_bn_fwd(const float* rgb, const float* mean, const float* stdv, const void* w, void* y, float* partial, size_t images,
                         int H, int W, void* stream);

/* 3x3 stride-1 padding-1 convolution of the frozen backbone's narrow layers (the stem's conv2 / conv3, layer1's conv2:
 * model/utils/clip.py:22-43, torch.nn.Conv2d(.., 3, padding=1, bias=False)) as a bf16 MFMA implicit GEMM with the BatchNorm work
 * around it folded in, like a3d_conv1x1_bn_fwd:  x [images][H][W][Cin] bf16 (NHWC), w [Cout][3][3][Cin] bf16 (the channels_last
 * layout of the torch weight), y [images][H][W][Cout] bf16;  in_scale / in_shift (or NULL): BatchNorm-apply (+ ReLU when in_relu)
 * of the producer applied to x on load, rounded to bf16 as a3d_bn_apply materialises it, zero padding applied AFTER it;
 * partial (or NULL): [a3d_conv3x3_nslab(..)][2][Cout] per-workgroup (sum, sum of squares) of the rounded outputs for
 * a3d_bn_finalize.  Served shapes (a3d_conv3x3_serves == 1): 32 -> 32, 32 -> 64 and 64 -> 64 channels, H a multiple of 8, W a
 * multiple of 32; anything else is refused (wider layers are compute-bound library convolutions).  16-byte aligned operands. */
int a3d_conv3x3_serves(int Cin, int Cout, int H, int W);
int a3d_conv3x3_nslab(size_t images, int H, int W, int Cin, int Cout);
int a3d_conv3x3_bn_fwd(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu, void* y,
                       float* partial, size_t images, int H, int W, int Cin, int Cout, void* stream);

/* ---- data plane ----------------------------------------------------------------------------------------
 * The `Resize` augmentation of datasets/utils.py:40-100 (nearest resize by a random scale, reflect-pad right/bottom, random
 * crop back to H x W; RGB and XYZ share the draws) as one gather pass over the collated batch on the device.
 * src/dst [frames][planes][H][W] fp32 (planes = cameras x 3), params [frames][4] int32 = (resized_h, resized_w, crop_i,
 * crop_j) per frame (host-sampled with the reference's RNG consumption, data.sample_resize_params);
 * dst = src[map(y, x)] * scale + shift (scale = shift = 0.5 un-normalises RGB, dataset_engine.py:134-137). */
int a3d_resize_crop(const float* src, float* dst, const int* params, int frames, int planes, int H, int W, float scale,
                    float shift, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACT3D_HIP_H */
