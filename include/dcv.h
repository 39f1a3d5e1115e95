/* libdcv_hip.so — C ABI of the MI355X-native DiChaViT training hot path.
 *
 * The reference (chaudatascience/diverse_channel_vit) has NO FFI: its hot path is stock ATen ops
 * called from Python.  The drop-in boundary is therefore the Python plugin contract
 * (models/__init__.py:9 `dichavit`, models/dichavit.py:844-865), mirrored by
 * diverse_channel_vit_amd/dichavit.py.  This header is the native layer beneath it: each entry
 * replaces the ATen op sequence cited next to it.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions: every entry returns 0 or a negative DCV_ERR_* code; never throws, never allocates,
 * never synchronises, never reads the environment; all buffers are device pointers owned by the caller;
 * `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); entries are re-entrant and
 * graph-capturable: the only process-wide state is a cache of each device's CU count
 * (hipDeviceGetAttribute, read once per device).  Every tuning choice is an explicit argument (the *_ex forms).
 * bf16 buffers are raw 16-bit brain-float; "f32" is IEEE binary32.  Only gfx950 code is built.
 */
#ifndef DCV_H
#define DCV_H
#ifdef __cplusplus
extern "C" {
#endif

#define DCV_OK 0
#define DCV_ERR_SHAPE (-1)
#define DCV_ERR_ALIGN (-2)
#define DCV_ERR_UNSUPPORTED (-3)
#define DCV_ERR_LAUNCH (-4)
#define DCV_ERR_NULL (-5)

/* dcv_gemm_nt epilogues */
#define DCV_EPI_BIAS_BF16 0      /* out bf16 = acc + bias                         (attn.qkv, vit.py:123)            */
#define DCV_EPI_BIAS_GELU_BF16 1 /* z = acc + bias (fp32): out bf16 = GELU_erf'(z), out2 bf16 = GELU_erf(z) (mlp.fc1 + act, vit.py:77-78) */
#define DCV_EPI_BIAS_RESID_F32 2 /* out f32 = (aux f32 ? aux : out) + s (acc + bias) (attn.proj / mlp.fc2 + residual, vit.py:142,397-398);
                                    s = 1, or with aux2 != NULL and T = rows per sample: s = aux2[row / T] — DropPath's per-sample
                                    keep_b / keep_prob (vit.py:37-56) */
#define DCV_EPI_PLAIN_BF16 3     /* out bf16 = acc                                 (input gradients)                */
#define DCV_EPI_GELU_BWD_BF16 4  /* out bf16 = acc * aux bf16, aux = the saved GELU'(z)  (grad through act, vit.py:78)    */
#define DCV_EPI_PATCH 5          /* tokens: out f32[b,1+t,:] = acc + bias + aux[c(t),:] + aux2[1+i(t),:] ;
                                    out2 f32[b*T+t,:] = acc + bias (optional)       (dichavit.py:377,409-415,565)   */

int dcv_version(void);
const char* dcv_error_string(int code);

/* C[M,N] = A[M,K] . W[N,K]^T (bf16 in, f32 accumulate) + epilogue.  K % 64 == 0, lda/ldw % 8 == 0.
 * Replaces F.linear forward / input-gradient and the Conv3d patch projection (on im2col rows). */
int dcv_gemm_nt(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int epilogue, const float* bias,
                void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux, const float* aux2, int T, int n,
                void* stream);
/* Tile variants of the *_ex forms.  gemm_nt: NARROW = 256 x 128 output tile, WIDE = 256 x 384 (needs N % 384 == 0, not the
 * PATCH epilogue).  gemm_tn_acc: NARROW = 128 x 128, WIDE = 384 x 128 (needs P % 384 == 0, Q % 128 == 0).  AUTO picks by
 * measured shape rules.  Forcing an illegal variant returns DCV_ERR_UNSUPPORTED. */
#define DCV_TILE_AUTO 0
#define DCV_TILE_NARROW 1
#define DCV_TILE_WIDE 2
#define DCV_TILE_PAIR 3 /* dcv_gemm_nt_ex only: 128 x 128 tiles by four-wave workgroups, TWO per CU (one stores while the other computes) */
#define DCV_TILE_ALT 4  /* dcv_gemm_nt_ex only: 256 x 384 tiles whose two 192-column halves are accumulated and stored in alternating phases by the two wave groups of a workgroup (bf16-output epilogues, N % 384 == 0); compiled into variant builds only (-DDCV_NT_ALT=1: measured slower), DCV_ERR_UNSUPPORTED otherwise */
/* dcv_gemm_nt with its launch controls as arguments: both kernels are persistent (one workgroup per CU walks the output
 * tiles, several rounds when there are more tiles than workgroups); grid_cap > 0 caps the number of workgroups (0 = one per
 * CU of the current device) — the data-parallel backward leaves CUs to RCCL's kernels this way. */
int dcv_gemm_nt_ex(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int epilogue, const float* bias,
                   void* out, int ldo, void* out2, int ldo2, const void* aux, int ldaux, const float* aux2, int T, int n,
                   int grid_cap, int tile, void* stream);

/* dW[P,Q] (f32) += sum_m Y[m,P] * X[m,Q] ; dbias[P] (f32, nullable) += sum_m Y[m,P].  bf16 inputs.
 * Replaces the weight/bias gradients of nn.Linear / Conv3d (autograd of vit.py:72-74,123,142; dichavit.py:377). */
int dcv_gemm_tn_acc(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw, float* dbias,
                    void* stream);
/* Residual addition AND the LayerNorm that follows it, from the accumulators of one 256 x 384 tile (N must be 384: a tile spans whole rows):
 *   x_out f32 [M,N] = resid + s (A W^T + bias)      (attn.proj / mlp.fc2 + residual, vit.py:397-398; s = branch_scale[row / T] or 1)
 *   u_out bf16 [M,N] = (x_out - mean) * rstd * gamma + beta,   mean / rstd f32 [M]   (norm2 / the next block's norm1, vit.py:397-398; eps as given)
 * = dcv_gemm_nt(DCV_EPI_BIAS_RESID_F32) + dcv_ln_fwd in one launch, without the re-read of x_out.  Statistics are centred (Chan's pairwise
 * combination), as dcv_ln_fwd's.  Returns DCV_ERR_UNSUPPORTED for N != 384.  Alignment (DCV_ERR_ALIGN otherwise): every pointer 16 bytes — u_out
 * included: its rows leave in 16-byte stores — lda, ldw, ldu multiples of 8, ldo, ldr of 4. */
int dcv_gemm_nt_resid_ln(const void* A, int lda, const void* W, int ldw, int M, int N, int K, const float* bias, const float* resid, int ldr,
                         const float* branch_scale, int T, float* x_out, int ldo, const float* gamma, const float* beta, float eps,
                         void* u_out, int ldu, float* mean, float* rstd, int grid_cap, void* stream);

/* the variant (DCV_TILE_NARROW / DCV_TILE_WIDE) the *_ex forms launch for a problem — pure functions of their arguments */
int dcv_gemm_nt_pick(int M, int N, int K, int epilogue, int tile);
int dcv_gemm_tn_pick(int M, int P, int Q, int tile);
int dcv_gemm_tn_acc_ex(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw, float* dbias,
                       int tile, void* stream);
/* DETERMINISTIC form (the reference trains with cudnn.deterministic = True, utils.py:394-401): the workgroups that split the token
 * rows store their partial tiles to `ws` with plain stores and a second launch adds them to dW / dbias in a fixed order, so two runs on
 * the same inputs give bit-identical gradients (dcv_gemm_tn_acc adds with fp32 atomics in dispatch order).  ws: caller-owned scratch of
 * at least dcv_gemm_tn_det_ws_floats(M, P, Q, tile) floats, 16-byte aligned, contents irrelevant before and after; lddw % 4 == 0. */
long dcv_gemm_tn_det_ws_floats(int M, int P, int Q, int tile);
int dcv_gemm_tn_acc_det(const void* Y, int ldy, const void* X, int ldx, int M, int P, int Q, float* dW, int lddw, float* dbias,
                        int tile, float* ws, long ws_floats, void* stream);

/* Several weight-gradient products over the SAME M token rows in ONE launch (round 3): dW_i[P_i,Q_i] += Y_i[M,P_i]^T X_i[M,Q_i],
 * dbias_i[P_i] += colsum(Y_i) (dbias may be NULL) for i < n <= 8.  Every P_i % 384 == 0 and Q_i % 128 == 0 and the tiles of all products
 * together must fit one resident round (sum of (P_i/384)(Q_i/128) <= CUs), else DCV_ERR_UNSUPPORTED: call the products one by one.  The CUs
 * are filled by the tiles of all products, so each tile is split far fewer ways over the rows than in n separate launches (a block's four
 * gradients: 7 ways instead of 21 / 21 / 28 / 85): a third of the partial-tile traffic and one launch's fixed cost instead of four.
 * ws != NULL: the deterministic form (dcv_gemm_tn_group_ws_floats(items, n, M) floats; partial tiles through ws, fixed-order second pass);
 * ws == NULL: fp32 atomics.  Replaces the four torch.autograd weight-gradient GEMMs of Block.backward (models/vit.py:72-82, 116-142). */
typedef struct dcv_tn_item {
    const void* Y; /* bf16 [M, P], row stride ldy */
    const void* X; /* bf16 [M, Q], row stride ldx */
    float* dW;     /* fp32 [P, Q], row stride lddw, accumulated into */
    float* dbias;  /* fp32 [P] or NULL */
    int ldy, ldx, P, Q, lddw, reserved;
} dcv_tn_item;
long dcv_gemm_tn_group_ws_floats(const dcv_tn_item* items, int n, int M);
int dcv_gemm_tn_group(const dcv_tn_item* items, int n, int M, float* ws, long ws_floats, void* stream);

/* LayerNorm (eps inside the sqrt, biased variance) — vit.py:361,374 / dichavit.py:651.
 * out is bf16 [M,D] (or f32 when out_is_f32); mean/rstd [M] may be NULL. x rows are x_row_stride floats apart. */
int dcv_ln_fwd(const float* x, long x_row_stride, const float* gamma, const float* beta, void* out, int out_is_f32, float* mean,
               float* rstd, int M, int D, float eps, void* stream);
/* dx_out = (dx_in ? dx_in : 0) + LN'(du) ; dx_bf16 (nullable) = bf16(dx_out) ; dgamma/dbeta += ... */
int dcv_ln_bwd(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
               const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
               float* dbeta, int M, int D, void* stream);

/* DETERMINISTIC forms of the other entries that combine partial sums of several workgroups (LayerNorm's dgamma / dbeta, the tokeniser's
 * d(channel_embed) / d(pos), the diversity loss' per-channel sums, the gradient norm): partials go through `ws` (at least *_det_ws_floats
 * floats, 16-byte aligned, contents irrelevant) and are added in a fixed order by a second launch.  Same arguments and results otherwise.
 * With dcv_gemm_tn_acc_det these make a whole training step bit-reproducible (attention, the NT GEMMs and AdamW never were order-dependent). */
long dcv_ln_bwd_det_ws_floats(int M, int D);
int dcv_ln_bwd_det(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                   const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                   float* dbeta, int M, int D, float* ws, long ws_floats, void* stream);
long dcv_patch_bwd_det_ws_floats(int B, int C, int n, int D);
int dcv_patch_bwd_det(const float* dx0, const float* dYloss, void* dY_bf16, float* dE, float* dpos, float* dcls, int B, int C, int n,
                      int D, float* ws, long ws_floats, void* stream);
long dcv_ortho_fwd_det_ws_floats(int B, int C, int n, int D);
int dcv_ortho_fwd_det(const float* Y, float* S, float* selfsq, float* tot, float* inv_norm, float* stats, int B, int C, int n, int D,
                      float* ws, long ws_floats, void* stream);
long dcv_sumsq_det_ws_floats(long n);
int dcv_sumsq_acc_det(const float* x, long n, float* acc, float* ws, long ws_floats, void* stream);

/* dcv_ln_bwd with the bf16 copy scaled per sample: dx_bf16 = bf16(bf16_row_scale[row / rows_per_sample] * dx_out) — the copy feeds the
 * backward of the residual branch below this LayerNorm, whose DropPath factor (vit.py:37-56) multiplies that branch's gradient; dx_out itself
 * (the residual stream) is not scaled.  ws NULL: atomic dgamma / dbeta; else the deterministic form (dcv_ln_bwd_det_ws_floats floats). */
int dcv_ln_bwd_scaled(const void* du, int du_is_f32, const float* x, long x_row_stride, const float* mean, const float* rstd,
                      const float* gamma, const float* dx_in, float* dx_out, long dx_row_stride, void* dx_bf16, float* dgamma,
                      float* dbeta, int M, int D, const float* bf16_row_scale, int rows_per_sample, float* ws, long ws_floats, void* stream);

/* softmax(q k^T * scale) v for packed qkv [B,N,3,H,64] bf16 -> o [B,N,H*64] bf16, lse [B,H,N] f32.
 * Replaces Attention.forward's q@k^T / softmax / @v (vit.py:123-141); the [B,H,N,N] matrix is never stored. */
int dcv_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, int head_dim, float scale, void* stream);
/* dqkv [B,N,3,H,64] bf16 from (qkv, o, dO, lse).  ws: f32 workspace of 2*B*H*N floats holding the row statistics
 *   ws[0 .. BHN) = -delta, delta[b,h,q] = sum_d dO*O;  ws[BHN .. 2 BHN) = lse * log2(e).
 * Two launches: dQ (which also writes ws), then dK/dV (which reads it). */
int dcv_attn_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                 int H, int head_dim, float scale, void* stream);
/* the launches individually (profiling / stream placement).  dcv_attn_bwd_dq fills the dQ slot of dqkv AND ws;
 * dcv_attn_bwd_dkdv fills the dK and dV slots and needs ws filled (by dcv_attn_bwd_dq or by dcv_attn_bwd_delta, which
 * computes only the statistics); its `lse` argument is kept for symmetry and not read. */
int dcv_attn_bwd_delta(const void* o, const void* dO, const float* lse, float* ws, int B, int N, int H, int head_dim, void* stream);
int dcv_attn_bwd_dq(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int H,
                    int head_dim, float scale, void* stream);
int dcv_attn_bwd_dkdv(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int H,
                      int head_dim, float scale, void* stream);
/* Query-row-restricted forms: only the query rows [0, Nq) of every (batch, head) are processed (1 <= Nq <= N); keys and
 * values are always all N rows.  Used for the LAST encoder block, whose output is read at the CLS row only
 * (ChannelVisionTransformer.forward returns norm(x)[:, 0], dichavit.py:651-652): Nq = 1.  Forward writes o / lse rows < Nq
 * only; backward reads dO rows < Nq only, writes dQ = 0 for the rows >= Nq and dK, dV of all keys. */
int dcv_attn_fwd_rows(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, float scale, void* stream);
int dcv_attn_bwd_rows(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int Nq,
                      int H, int head_dim, float scale, void* stream);
int dcv_attn_bwd_dq_rows(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                         int Nq, int H, int head_dim, float scale, void* stream);
int dcv_attn_bwd_dkdv_rows(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                           int H, int head_dim, float scale, void* stream);

/* Pre-scaled-q forms (round 4).  The q part of qkv holds q * scale * log2(e) — the caller scales the operand copy of W_q and the q part of the
 * bias BEFORE the qkv GEMM (dcv_cast_scaled_ranges), so q is still rounded once.  The score accumulators then start at the row constants (-m,
 * -LSE log2 e, -delta) and p = exp2(accumulator): one vector instruction less per score in each kernel.  Outputs as the plain entries: o, lse
 * (natural log), dqkv with dQ the gradient with respect to the UNSCALED q.  The ws of the _ps pair carries -LSE log2 e: do not mix it with the
 * plain pair's.  Replaces nothing in the reference (vit.py:128-133 computes (q @ k^T) * scale on a materialised matrix). */
int dcv_attn_fwd_rows_ps(const void* qkv, void* o, float* lse, int B, int N, int Nq, int H, int head_dim, void* stream);
int dcv_attn_bwd_rows_ps(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N, int Nq,
                         int H, int head_dim, float scale, void* stream);
int dcv_attn_bwd_dq_rows_ps(const void* qkv, const void* o, const void* dO, const float* lse, float* ws, void* dqkv, int B, int N,
                            int Nq, int H, int head_dim, float scale, void* stream);
int dcv_attn_bwd_dkdv_rows_ps(const void* qkv, const void* dO, const float* lse, const float* ws, void* dqkv, int B, int N, int Nq,
                              int H, int head_dim, float scale, void* stream);

/* x [B,Ct,H,W] f32 (x_is_u8 == 0: normalised images, the reference's batch format) or u8 (raw pixels), ch_idx int32[C]
 * (device) -> bf16 [B*C*(H/P)*(W/P), P*P] patch rows (dichavit.py:134/210,377).  scale/shift f32[C] (nullable, indexed by
 * gathered position): x*scale[c] + shift[c], i.e. the (x/255 - mean_c)/std_c of the CPU pipeline
 * (jump_cp_transforms.py:119-121) fused into the tokeniser. */
int dcv_im2col_bf16(const void* x, int x_is_u8, const int* ch_idx, const float* scale, const float* shift, void* out, int B, int Ct,
                    int C, int H, int W, int P, void* stream);
/* one pass over d(tokens) f32 [B,1+C*n,D]: dY_bf16 [B*C*n,D] = dx0[:,1:] (+ dYloss) ; dE [C,D], dpos [1+n,D], dcls [D] += */
int dcv_patch_bwd(const float* dx0, const float* dYloss, void* dY_bf16, float* dE, float* dpos, float* dcls, int B, int C, int n,
                  int D, void* stream);
/* dropout_tokens_hcs (dichavit.py:568-627): scatter == 0: out[b,k,:] = x[b,idx[k],:] (x [B,N,D] -> out [B,Nk,D]);
 * scatter != 0: out[b,idx[k],:] = x[b,k,:] (x [B,Nk,D] -> out [B,N,D], caller zero-fills out first).  f32, D % 4 == 0. */
int dcv_gather_tokens(const float* x, const int* idx, float* out, int B, int N, int Nk, int D, int scatter, void* stream);
int dcv_fill_cls(float* x, const float* cls, const float* pos0, int B, long batch_stride, int D, void* stream);

/* ortho_proj_loss_fn_v2 statistics (loss_fn.py:24-48): stats[b] = (pos_sum, neg_sum) of image b.
 * S [B,C,D], selfsq [B,C], tot [B,D], inv_norm [B,C*n] are outputs kept for the backward. */
int dcv_ortho_fwd(const float* Y, float* S, float* selfsq, float* tot, float* inv_norm, float* stats, int B, int C, int n, int D,
                  void* stream);
/* dY [B,C*n,D] f32 = d loss / dY given coef[b] = (dL/dpos_sum_b, dL/dneg_sum_b). */
int dcv_ortho_bwd(const float* Y, const float* S, const float* tot, const float* inv_norm, const float* coef, float* dY, int B,
                  int C, int n, int D, void* stream);

/* The channel-embedding proxy regulariser in ONE launch, value and both gradients (round 3): emb [C, D] and proxies [C, D] fp32 contiguous;
 * loss[0] = cross_entropy(-cdist(scale * normalize(emb), scale * normalize(proxies))^2, eye(C)) (mean over rows), d_emb / d_proxies [C, D] =
 * its gradients (for an upstream gradient of 1).  Replaces proxy_loss(channel_emb_proxies[cur_channels], channel_embed, eye, scale)
 * (models/dichavit.py:399-402, models/loss_fn.py:7-21) and its autograd backward: ~40 launches on [C, D] tensors.  dcv_proxy_loss_supported:
 * C <= 32, D <= 1024, C * D <= 16384 (two normalised copies in LDS); DCV_ERR_UNSUPPORTED otherwise (the caller keeps the op-by-op form). */
int dcv_proxy_loss_supported(int C, int D);
int dcv_proxy_loss(const float* emb, const float* proxies, int C, int D, float scale, float* loss, float* d_emb, float* d_proxies, void* stream);

/* fused AdamW over a flat fp32 range; g is multiplied by grad_scale first (1/world for data parallel). */
int dcv_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int step, float grad_scale, void* stream);
/* same update with the scalars read from device memory: hyper_dev[8] = {lr, beta1, beta2, eps, weight_decay,
 * 1/(1-beta1^t), 1/sqrt(1-beta2^t), grad_scale}.  Lets a captured HIP graph of the step be replayed while the
 * step count and the schedulers' lr / weight decay advance (dcv_adamw_set_hyper rewrites the 32 bytes between replays). */
int dcv_adamw_dyn(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev, void* stream);
/* writes hyper_dev[8] for step `step` (bias corrections computed in double on the host, passed BY VALUE to a one-thread
 * kernel): stream-ordered, no host staging buffer, so a host that runs many steps ahead of the device cannot overwrite the
 * scalars of a step that has not executed yet. */
int dcv_adamw_set_hyper(float* hyper_dev, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                        float grad_scale, void* stream);
/* Gradient clipping (trainer.py:1003-1004 -> torch.nn.utils.clip_grad_norm_, L2): dcv_sumsq_acc adds sum(x^2) of a flat fp32
 * range to the device scalar *acc (zero it first; call once per gradient buffer); dcv_clip_scale multiplies a range by
 * min(1, max_norm / (sqrt(*sumsq_dev) + 1e-6)).  Everything stays on the device: no host sync, graph-capturable. */
int dcv_sumsq_acc(const float* x, long n, float* acc, void* stream);
int dcv_clip_scale(float* x, long n, const float* sumsq_dev, float max_norm, void* stream);
/* bf16 operand copies of the parameter arena */
int dcv_cast_bf16(const float* src, void* dst, long n, void* stream);
/* desc_dev: device int64 [n_desc][4] = {src offset, dst offset, R, C}; dst[C][R] = bf16(src[R][C]) */
int dcv_cast_transpose_bf16(const float* src_base, void* dst_base, const long long* desc_dev, int n_desc, int max_tiles, void* stream);
/* Stochastically rounded variants for the TRAINING operand copies: bf16 = (fp32 bits + r) >> 16 with r = the low 16
 * bits of lowbias32(element_offset_from_src_base ^ seed_dev[0]*0x9E3779B9).  Both use the offset from the arena base, so a
 * weight and its transposed copy round identically; seed_dev is a device word the host bumps once per step (graph-replay
 * safe).  The reference trains in fp32 (trainer.py:963-1006, no autocast on CPU): round-to-nearest copies lag the fp32
 * master coherently under AdamW's +-lr steps, unbiased rounding removes that lag. */
int dcv_cast_bf16_sr(const float* src, void* dst, long n, const unsigned* seed_dev, void* stream);
int dcv_cast_transpose_bf16_sr(const float* src_base, void* dst_base, const long long* desc_dev, int n_desc, int max_tiles,
                               const unsigned* seed_dev, void* stream);

/* Scaled re-cast of parts of an operand copy.  desc_dev: device int64 [n_desc][5] = {src offset (floats from src_base), dst offset, count,
 * scaled_count, kind}.  kind 0: dst_bf16[dst + i] = bf16(f_i * src[i]), kind 1: dst_f32[dst + i] = f_i * src[i], with f_i = scale for i <
 * scaled_count and 1 after.  seed_dev NULL: round to nearest; else the stochastic rounding of dcv_cast_bf16_sr with the same per-element bits
 * (offset from src_base), so the rewritten range is what that cast would have produced from scale * src.  blocks_per_desc: grid.x. */
int dcv_cast_scaled_ranges(const float* src_base, void* dst_bf16, float* dst_f32, const long long* desc_dev, int n_desc, int blocks_per_desc,
                           float scale, const unsigned* seed_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
