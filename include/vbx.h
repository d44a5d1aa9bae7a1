/*
 * vbx.h -- C ABI of libvbx_hip.so: the MI355X (gfx950) native hot path of
 * lucidrains/voicebox-pytorch (VoiceBox transformer fwd/bwd + CFM sampling loop).
 *
 * The reference has NO FFI / plugin interface: its "operators" are ATen calls made
 * from Python modules.  Each entry point below therefore cites the reference *Python call site*
 * (file:line under /root/reference/voicebox_pytorch/) whose ATen work it replaces.  A maintainer
 * binds these with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers owned by the caller
 *    (PyTorch-allocated); the library never allocates or frees device memory and never
 *    synchronises, so every entry point is hipGraph-capture safe.
 *  - work is enqueued on `stream` (a hipStream_t passed as void*; 0 = default stream).
 *  - return 0 on success; negative = VBX_E* argument error (checked before launch);
 *    positive = hipError_t of the launch.  vbx_last_error() returns a thread-local message.
 *  - dtypes: bf16/fp16 tensors are raw 16-bit; masks are uint8 (torch.bool); statistics fp32.
 *  - activations are token-major: row r = b * Np + n  (Np = frames + register tokens).
 */
#ifndef VBX_H
#define VBX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBX_VERSION 1

enum { VBX_OK = 0, VBX_EINVAL = -1, VBX_EUNSUPPORTED = -2, VBX_EWORKSPACE = -3 };

int vbx_version(void);
const char* vbx_last_error(void);
/* Device sanity: returns 0 iff device `dev` reports gcnArchName gfx950. */
int vbx_check_device(int dev);

/* ------------------------------------------------------------------ GEMM (MFMA bf16, fp32 acc) */
/* mode: how the two operands are laid out (contraction index k):
 *   NT: A[M,K] (k contiguous), B[N,K] (k contiguous)  -> C[M,N] = A . B^T    nn.Linear forward
 *   NN: A[M,K] (k contiguous), B[K,N] (n contiguous)  -> C[M,N] = A . B      dgrad  (dX = dY . W)
 *   TN: A[K,M] (m contiguous), B[K,N] (n contiguous)  -> C[M,N] = A^T . B    wgrad  (dW = dY^T . X)
 * Replaces every nn.Linear / F.linear on the path (voicebox_pytorch.py:314-315,320,333,345,348,
 * 938,966,1078,1092) and their autograd backward. */
enum { VBX_GEMM_NT = 0, VBX_GEMM_NN = 1, VBX_GEMM_TN = 2 };

enum {
  VBX_EPI_BF16 = 0,       /* C bf16 [M,ldc]   (+bias if given)                                     */
  VBX_EPI_F32 = 1,        /* C fp32 [M,ldc]   (+bias) (+resid fp32 [M,ldc]) ; optional bf16 copy   */
  VBX_EPI_QKV = 2,        /* to_qkv + MultiheadRMSNorm + rotary (voicebox_pytorch.py:320-328)       */
  VBX_EPI_GEGLU = 3,      /* FeedForward[0] + GEGLU (voicebox_pytorch.py:338-340,345)               */
  VBX_EPI_SPLITK = 4      /* fp32 partial slabs [splits][M][N] for TN wgrad                        */
};

typedef struct {
  int mode, epilogue;
  int M, N, K;
  int lda, ldb, ldc;
  const void* A;            /* bf16 */
  const void* B;            /* bf16 */
  void* C;                  /* per epilogue */
  const float* bias;        /* [N] or NULL */
  const float* resid;       /* EPI_F32: fp32 [M,ldc] added, or NULL */
  void* C2;                 /* EPI_F32: optional bf16 copy [M,ldc]; EPI_GEGLU: bf16 pre-activation [M,N] or NULL */
  int splits;               /* EPI_SPLITK: number of K splits (>=1) */
  /* EPI_QKV: N = 3*H*64.  rows r = b*Np + n. */
  int Np, H;
  float qk_scale;           /* sqrt(dim_head) (MultiheadRMSNorm.scale); <= 0 disables qk-norm */
  const float* q_gamma;     /* [H,64] */
  const float* k_gamma;     /* [H,64] */
  const float* rot_cos;     /* [Np,32] fp32 (host-built table, rotary freqs are (ang,ang)) */
  const float* rot_sin;     /* [Np,32] */
  void* q16; void* k16;     /* fp16 [B,H,Np,64]  : operands of QK^T */
  void* qb;  void* kb;      /* bf16 [B,H,Np,64]  : backward operands (may be NULL in eval) */
  void* v;                  /* bf16 [B,H,Np,64] */
  float* q_rnorm; float* k_rnorm; /* fp32 [B,H,Np] 1/max(|t|,1e-12) (may be NULL in eval) */
  int f16;                  /* 1: A and B hold fp16 (not bf16) -- NT mode only (the forward GEMMs).  Everything that
                             * feeds the attention logits 10*q.k (|q|=|k|=8, std ~80) is precision critical: bf16
                             * operands (2^-9) perturb the logits by ~0.2, fp16 (2^-11) by ~0.05 at the same MFMA
                             * rate; the backward GEMMs keep bf16 (gradient range).  With f16=1 EPI_GEGLU writes
                             * C as fp16. */
  void* v16;                /* EPI_QKV: fp16 copy of v [B,H,Np,64] (forward P.V operand); v (bf16) may then be NULL */
  void* C3;                 /* EPI_GEGLU: optional bf16 copy of C [M,ldc] (wgrad operand) */
  float q_prescale;         /* EPI_QKV: q16 is written as q-hat * q_prescale (vbx_attn_q_prescale(scale): the attention kernels'
                             * contract); <= 0 means 1.  qb / k16 / kb are never scaled. */
  /* EPI_BF16, NN mode, optional (round 5): this GEMM is the dgrad of Attention.to_out (C = dO bf16 [B*Np, H*64]) and the attention
   * backward's delta[b,h,n] = sum_d dO[b,n,h*64+d] * O[b,n,h*64+d] is written as a by-product (then pass out = NULL to vbx_attn_bwd*).
   * delta_o: the forward output O, fp16, same [M, ldc] layout as C; needs Np, H with N = H*64.  VBX_EUNSUPPORTED when the tile that
   * would serve this shape has no such epilogue (call again without, and let vbx_attn_bwd* run its own pass). */
  const void* delta_o;
  float* delta;
} vbx_gemm_desc;

int vbx_gemm(const vbx_gemm_desc* d, void* stream);
/* Tuning knob (results are identical up to fp32 summation order): which tile serves vbx_gemm / the grouped launch.
 * 0 automatic per shape (default; environment VBX_GEMM_PATH=<n> presets it), 1 the 128-wide kernels only, 2 the 256 x 256
 * 8-wave kernel (gemm3.hip) wherever it can serve, 3 the 128 x 256 two-workgroups-per-CU kernel (gemm4.hip) wherever it can.
 * Not thread safe; call before launching work. */
int vbx_gemm_select(int path);
/* Round 6: to_qkv and FeedForward-in at K = 512 (NT, VBX_EPI_QKV / VBX_EPI_GEGLU, every backward copy or none) run on the
 * WEIGHT-STATIONARY kernel (csrc/gemm5.hip): one 4-wave workgroup per CU keeps a 256-feature weight panel in its registers and walks
 * 32-row activation blocks; the epilogue of a block runs inside the next block's MFMA stream.  It owns a whole CU (512 registers
 * per lane, 128 KiB LDS), so two such launches on two streams cannot share CUs: a caller that runs two of them concurrently
 * (the sampler's two half-batch streams) gives each a share with vbx_gemm5_cu_limit(n) -- the launches then use at most n CUs
 * (0 = all; process-global, read at launch, i.e. baked into a captured graph).  VBX_GEMM5=0 / vbx_gemm_select(1): the tiled kernels.
 * vbx_gemm_select(4): as 0 with this kernel forced on even when VBX_GEMM5=0. */
int vbx_gemm5_cu_limit(int n);
/* n (1..4) TN / VBX_EPI_SPLITK GEMMs in ONE launch of the 256 x 256 tile (same slab layout and results as n vbx_gemm calls):
 * the four weight-gradient GEMMs of a layer are 8-24 such tiles each; together, with 3 K-splits, they fill 198 CUs (92 us in
 * situ against 4 x 38 us as separate 128-wide launches).  With vbx_gemm_select(1) it falls back to n separate launches. */
int vbx_gemm_tn_splitk_grouped(const vbx_gemm_desc* descs, int n, void* stream);
/* Sum split-K slabs [splits][M][N] and scatter into dst (fp32): dst[rowmap(i)][j] (+)= sum_s slab.
 * rowmap: 0 identity; 1 GEGLU de-interleave with (F, Fp): packed row p -> ((p%128)<64 ?
 * (p/128)*64+p%128 : F + (p/128)*64 + p%128-64), rows/cols beyond the valid range dropped. */
int vbx_splitk_reduce(const float* slabs, int splits, int M, int N, float* dst, int dst_rows, int dst_cols,
                      int dst_ld, int rowmap, int F, int accumulate, void* stream);

/* ------------------------------------------------------------------ norms */
/* AdaptiveRMSNorm / RMSNorm forward (voicebox_pytorch.py:246-247, 270-276):
 * y[r,:] = x[r,:]/max(|x[r,:]|,1e-12)*sqrt(D) * gamma[b(r),:] + beta[b(r),:]  -> bf16.
 * gamma/beta: fp32 with batch stride gb_stride (0 for the non-adaptive RMSNorm), beta may be NULL.
 * Rows: for each batch b, rows n in [n0, n0+rows_per_batch) of x (row stride Np per batch);
 * output y is dense [B*rows_per_batch, D]. */
int vbx_rmsnorm_fwd(const float* x, const float* gamma, const float* beta, long gb_stride, void* y_bf16,
                    void* y_f16 /* optional fp16 copy (same dense layout); y_bf16 may then be NULL */,
                    int B, int Np, int n0, int rows_per_batch, int D, void* stream);
/* same, fp32 output rows [B*rows_per_batch, D] (final norm of a standalone Transformer.forward, :479) */
int vbx_rmsnorm_fwd_f32(const float* x, const float* gamma, const float* beta, long gb_stride, float* y_f32, int B, int Np,
                        int n0, int rows_per_batch, int D, void* stream);
/* backward: dx_out = dx_in (or 0 if NULL) + d/dx ; partial dgamma/dbeta sums per row chunk:
 * part[b][chunk][2][D] (chunk count = vbx_rmsnorm_bwd_chunks(rows_per_batch)).  dy is dense bf16 [B*rows, D].
 * dx tensors have the same (Np, n0) row addressing as x.  dxb: optional bf16 copy of dx_out (same addressing). */
int vbx_rmsnorm_bwd_chunks(int rows_per_batch);
int vbx_rmsnorm_bwd(const float* x, const float* gamma, long gb_stride, const void* dy_bf16, const float* dx_in,
                    float* dx_out, void* dxb_bf16, float* part,
                    float* colpart /* optional [B][chunks][D]: per-chunk column sums of dx_in (a fused bias gradient) */,
                    int B, int Np, int n0, int rows_per_batch, int D, void* stream);

/* ------------------------------------------------------------------ attention */
/* Attend.forward math path (attend.py:121-135): softmax(scale * q k^T + key-pad mask) v, fused
 * flash-style (scores never materialised).  q16,k16,v16 fp16 are [B,H,Np,64]; mask uint8
 * [B,Np] or NULL; lse fp32 [B,H,Np] in log2 units (m + log2 l).
 * CONTRACT (round 5, every vbx_attn_* entry point): q16 holds q PRE-MULTIPLIED by scale * log2(e) (vbx_attn_q_prescale(scale)),
 * so that q16 . k16 is directly the exponent of exp2 -- the kernels fold the softmax statistics into the MFMA accumulator
 * (csrc/attn_bwd_fold.inc) and have no per-element scale multiply left.  The to_qkv epilogue writes it that way
 * (vbx_gemm_desc.q_prescale); a caller with plain fp32 q multiplies before rounding to fp16.  qb (the bf16 backward operand)
 * stays UNSCALED; `scale` remains the multiplier of dq / dk, which are gradients w.r.t. the unscaled q / k. */
float vbx_attn_q_prescale(float scale);
int vbx_attn_fwd(const void* q16, const void* k16, const void* v16 /* fp16 */, const uint8_t* mask,
                 void* out16 /* fp16 [B,Np,H*64] */, void* out_bf16 /* optional bf16 copy (backward operand) */,
                 float* lse, int B, int H, int Np, float scale, void* stream);
/* backward (autograd of attend.py:121-135).  dout bf16 [B,Np,H*64]; qb,kb bf16 copies of q,k; delta fp32 [B,H,Np] scratch;
 * dq,dk fp32 [B,H,Np,64]; dv is written bf16 token-major at dv[(b*Np+n)*dv_ld + h*64 + d].
 * One kernel family serves it: the TWO-BODY kernel -- dq and dk/dv bodies in one launch, S / dP evaluated in both, no inter-workgroup
 * waits, deterministic.  By default its softmax statistics are folded into the MFMA accumulator (csrc/attn_bwd_fold.inc: P = exp2(-(L - q.k))
 * with L added inside the matrix pipe); vbx_attn_bwd_select(3) / VBX_ATTN_BWD_FOLD=0 runs the same bodies without the fold (round 3's
 * arithmetic -- what attention dropout always uses); the two differ by fp32 rounding of the exponent only.  Tail tiles of <= 16 rows
 * (Np % 128 <= 16: the register tokens) run a 16 x 16 MFMA role with their ring rows split over the waves (csrc/attn_bwd_ragged.inc).
 * Round 3's ONE-PASS chain kernel (select 2: S / dP once, dq summed by an ordered chain of workgroups through device memory) was correct
 * and 30 - 60 % slower; it was removed in round 6 (docs/history.md): vbx_attn_bwd_select(2) returns VBX_EUNSUPPORTED and
 * vbx_attn_bwd_scratch_bytes() 0.  `scratch` stays in the signatures for ABI stability and is ignored (pass NULL).
 * vbx_attn_bwd_select: 0 automatic (= 1), 1 folded two-body, 3 unfolded two-body. */
size_t vbx_attn_bwd_scratch_bytes(int B, int H, int Np);
int vbx_attn_bwd_select(int variant);
int vbx_attn_bwd_variant(void); /* always 1 (two-body) since round 6 */
int vbx_attn_bwd(const void* q16, const void* k16, const void* qb, const void* kb, const void* v,
                 const uint8_t* mask, const void* out /* forward output [B,Np,H*64]; NULL: `delta` already holds rowsum(dO * O)
                                                         (vbx_gemm_desc.delta) and the pass that computes it is skipped */, int out_is_f16,
                 const void* dout, const float* lse, float* delta, float* dq, float* dk, void* dv, int dv_ld, int B, int H,
                 int Np, float scale, void* scratch, void* stream);
/* backward of MultiheadRMSNorm + rotary (voicebox_pytorch.py:286-287,199): consumes dq/dk fp32
 * [B,H,Np,64] and the saved q16/k16 + rnorm, writes d(raw q|k) bf16 into dqkv[(b*Np+n)*ld + which*H*64
 * + h*64 + d] and partial gamma grads gpart[2][vbx_qknorm_rope_bwd_gpart_rows(B)][H][64]. */
/* vbx_attn_bwd with the backward of rotary + MultiheadRMSNorm (vbx_qknorm_rope_bwd) folded into the epilogues of its two kernels:
 * no fp32 dq / dk round trip; writes d(qkv) bf16 [B*Np, ld] (q | k | v blocks of H*64 columns) directly.  gpart: partial gamma
 * gradients [2][B * vbx_attn_bwd_fused_tiles(Np)][H][64] (q then k), to be summed over the rows (qk_scale > 0 only). */
int vbx_attn_bwd_fused_tiles(int Np);
int vbx_attn_bwd_fused(const void* q16, const void* k16, const void* qb, const void* kb, const void* v, const uint8_t* mask,
                       const void* out, int out_is_f16, const void* dout, const float* lse, float* delta, const float* q_rnorm,
                       const float* k_rnorm, const float* q_gamma, const float* k_gamma, const float* rot_cos, const float* rot_sin,
                       float qk_scale, void* dqkv, int ld, float* gpart, int B, int H, int Np, float scale, void* scratch,
                       void* stream);
/* ---- training-time dropout (attend.py:131 attention probabilities, voicebox_pytorch.py:346 GEGLU output)
 * The mask is a pure function of (seed, stream_id, element index) through Philox4x32-10, 16 random bits per element: an element is
 * kept iff its lot < thr16 = round((1 - p) * 65536), survivors are scaled by vbx_dropout_keep_scale(p) = 65536 / thr16 (exactly
 * unbiased for the realised keep rate).  Counters: attention element (bh, q, key) -> (4 * (key / 32) + (key % 32) / 8, q, bh, stream_id),
 * lot key % 8; FeedForward element (row, col) -> (col / 8, row, 0, stream_id), lot col % 8; Philox key = (seed low, seed high);
 * lot e of a call = 16-bit half e % 2 (low first) of output word e / 2.  The runtime uses stream_id = 2 * layer (attention) and
 * 2 * layer + 1 (FeedForward) with one seed per forward (vbx_io.drop_seed).
 * vbx_attn_dropout_bits writes one layer's keep bits in both orientations the kernels read: bits_rm [B*H][Np][W] (bit key % 32 of
 * word key / 32) and bits_cm [B*H][Np keys][W] (bit q % 32 of word q / 32), W = vbx_dropout_bits_words(Np) 32-bit words per row.
 * The *_dropout attention entry points take them: the forward keeps the softmax statistics of the undropped probabilities; the
 * backward runs on the two-body kernel.  vbx_dropout_rows drops a [rows, cols] 16-bit matrix in place (fp16 copy and / or bf16
 * copy of the same values; cols and ld multiples of 8) -- applied to the GEGLU output in the forward and to its gradient in the
 * backward. */
int vbx_dropout_bits_words(int Np);
float vbx_dropout_keep_scale(float p);
int vbx_attn_dropout_bits(void* bits_rm, void* bits_cm, int BH, int Np, unsigned long long seed, unsigned stream_id, float p,
                          void* stream);
int vbx_dropout_rows(void* x_f16, void* x_bf16, long rows, int cols, int ld, unsigned long long seed, unsigned stream_id, float p,
                     void* stream);
/* the same mask (same seed / stream_id / element index) on an fp32 matrix: precise mode's unrounded GEGLU output */
int vbx_dropout_rows_f32(float* x, long rows, int cols, int ld, unsigned long long seed, unsigned stream_id, float p, void* stream);
int vbx_attn_fwd_dropout(const void* q16, const void* k16, const void* v16, const uint8_t* mask, void* out16, void* out_bf16,
                         float* lse, int B, int H, int Np, float scale, const void* bits_rm, float p, void* stream);
int vbx_attn_bwd_dropout(const void* q16, const void* k16, const void* qb, const void* kb, const void* v, const uint8_t* mask,
                         const void* out, int out_is_f16, const void* dout, const float* lse, float* delta, float* dq, float* dk,
                         void* dv, int dv_ld, int B, int H, int Np, float scale, const void* bits_rm, const void* bits_cm, float p,
                         void* stream);
/* bits_rm == NULL: identical to vbx_attn_bwd_fused */
int vbx_attn_bwd_fused_dropout(const void* q16, const void* k16, const void* qb, const void* kb, const void* v, const uint8_t* mask,
                               const void* out, int out_is_f16, const void* dout, const float* lse, float* delta,
                               const float* q_rnorm, const float* k_rnorm, const float* q_gamma, const float* k_gamma,
                               const float* rot_cos, const float* rot_sin, float qk_scale, void* dqkv, int ld, float* gpart, int B,
                               int H, int Np, float scale, void* scratch, const void* bits_rm, const void* bits_cm, float p,
                               void* stream);
int vbx_qknorm_rope_bwd(const float* dq, const float* dk, const void* q16, const void* k16, const float* q_rnorm,
                        const float* k_rnorm, const float* q_gamma, const float* k_gamma, const float* rot_cos,
                        const float* rot_sin, float qk_scale, void* dqkv, int ld, float* gpart, int B, int H, int Np,
                        float q16_scale /* the factor q16 carries: vbx_attn_q_prescale(scale) */, void* stream);

/* ------------------------------------------------------------------ small / memory-bound ops */
/* x_cat bf16 [B*N, 2*D] = (x, cond * ~cond_mask)   (voicebox_pytorch.py:1035,1075-1076) */
int vbx_pack_embed_input(const float* x, const float* cond, const uint8_t* cond_mask, void* out_f16,
                         void* out_bf16 /* optional copy: wgrad operand */, int B, int N, int D, void* stream);
/* ConvPositionEmbed + residual + register tokens (voicebox_pytorch.py:220-233,1080,422-425):
 * xs[b, R+n, :] = e[b,n,:] + mask*gelu(conv(mask*e)[b,n,:] + bias);  xs[b, r<R, :] = reg[r,:]. */
/* text-conditioned embed input (condition_on_text = True, voicebox_pytorch.py:1035-1076):
 * out[b*N+n, :] = [ x | cond_emb | cond' ] as fp16 (+bf16), width 2*D + E, with
 *   cond'    = drop[b] ? null_cond : cond * ~cond_mask                       (:1035, :1043-1048; drop_mask may be NULL)
 *   cond_emb = table[drop[b] ? null_id : ids[b, .]] resized from T tokens to N frames by interpolate_1d (:89-107, :1057-1066):
 *              F.interpolate bilinear, align_corners=False (identity when T == N).
 * vbx_cond_emb_bwd scatters d(cond_emb) (bf16 [B*N, E], row stride ld) into the table gradient with fp32 atomics. */
int vbx_pack_embed_input_text(const float* x, const float* cond, const uint8_t* cond_mask, const uint8_t* drop_mask,
                              const float* null_cond, const long* ids, int T, const float* table, int E, long null_id,
                              void* out_f16, void* out_bf16, int B, int N, int D, void* stream);
/* the same rows unrounded (fp32 [B*N, 2*D + E]): precise mode's to_embed operand */
int vbx_embed_input_text_f32(const float* x, const float* cond, const uint8_t* cond_mask, const uint8_t* drop_mask,
                             const float* null_cond, const long* ids, int T, const float* table, int E, long null_id,
                             float* out_f32, int B, int N, int D, void* stream);
int vbx_cond_emb_bwd(const void* demb_bf16, int ld, const long* ids, int T, const uint8_t* drop_mask, long null_id,
                     float* gtable, int B, int N, int E, void* stream);
/* DurationPredictor front end (voicebox_pytorch.py:793-823): out fp16 [B*N, E+D] = [ to_phoneme_emb(max(ids,0)) | cond'' ] with
 * cond'' = curtail_or_pad(where(drop[b], null_cond, cond * ~cond_mask), N); ids [B,N] (-1 = padding), cond fp32 [B,S,D],
 * cond_mask [B,S] (may be NULL), drop_mask [B] (may be NULL).  Feeds vbx_gemm (to_embed, f16 operands). */
int vbx_pack_phoneme_input(const long* ids, const float* table, int E, const float* cond, int S, const uint8_t* cond_mask,
                           const uint8_t* drop_mask, const float* null_cond, void* out_f16, int B, int N, int D, void* stream);
/* to_pred = Linear(dim, 1) + squeeze (voicebox_pytorch.py:672-675): out[r] = x[r,:] . w + bias[0]  (bias may be NULL). */
int vbx_rowdot(const float* x, const float* w, const float* bias, float* out, long rows, int D, void* stream);
/* standalone Transformer.forward (voicebox_pytorch.py:417-431, :476-477): residual stream [B,N+R,D] = register tokens (rows
 * n < R) followed by x [B,N,D]; backward: dx = rows n >= R of dxs, dreg[R,D] = sum over the batch of rows n < R. */
int vbx_stack_input(const float* x, const float* reg, float* xs, int B, int N, int R, int D, void* stream);
int vbx_stack_input_bwd(const float* dxs, float* dx, float* dreg /* may be NULL */, int B, int N, int R, int D, void* stream);
/* u-net skip connection (voicebox_pytorch.py:458-463): cat [rows, 2*D] = (x | scale * skip) as fp16 and / or bf16 (the combiner's GEMM operand);
 * backward: dcat fp32 [rows, 2*D] = d(cat) -> dx = dcat[:, :D] (+ bf16 copy), dskip = scale * dcat[:, D:]; and the deferred add of a
 * stored dskip into the gradient of the layer input it was taken from: dx += dskip (+ bf16 copy). */
int vbx_unet_cat(const float* x, const float* skip, float scale, void* cat_f16, void* cat_bf16, long rows, int D, void* stream);
int vbx_unet_split(const float* dcat, float scale, float* dx, void* dx_bf16, float* dskip, long rows, int D, void* stream);
int vbx_unet_addskip(float* dx, void* dx_bf16, const float* dskip, long n, void* stream);
int vbx_convpos_fwd(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* reg,
                    float* xs, int B, int N, int R, int D, int ksize, void* stream);
/* vbx_convpos_fwd with libm's erff in the GELU instead of the fast path's Abramowitz-Stegun form (precise mode) */
int vbx_convpos_fwd_libm(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* reg,
                         float* xs, int B, int N, int R, int D, int ksize, void* stream);
/* ksize: any odd kernel size <= 31 (the reference default is 31, voicebox_pytorch.py:893; one unrolled instantiation per size).
 * backward: de = dxs[:,R:] + conv-transpose(...) ; dw/db partials [chunks][D][ksize+1]; dreg [R,D]. */
int vbx_convpos_bwd(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* dxs,
                    float* dpre_tmp /* fp32 [B,N,D] scratch */, float* de, void* de_bf16,
                    float* wpart /* [chunks][D][64]: k<ksize weight grads, [63] bias grad */, float* dreg, int B, int N,
                    int R, int D, int ksize, void* stream);
int vbx_convpos_bwd_chunks(int B, int N);
/* dw[d][k] = sum_chunks wpart[.][d][k], db[d] = sum_chunks wpart[.][d][63] */
int vbx_conv_wgrad_finalize(const float* wpart, int chunks, int D, int ksize, float* dw, float* db, void* stream);
/* time embedding: LearnedSinusoidalPosEmb -> Linear -> SiLU (voicebox_pytorch.py:163-167,916-920) */
int vbx_time_embed_fwd(const float* times, const float* w_sin, const float* w1, const float* b1, float* four,
                       float* pre, float* temb, int B, int D, int Th, void* stream);
int vbx_time_embed_bwd_scratch_floats(int B, int D);
int vbx_time_embed_bwd(const float* times, const float* w_sin, const float* w1, const float* four, const float* pre,
                       const float* dtemb, float* dw_sin, float* dw1, float* db1, float* scratch /* vbx_time_embed_bwd_scratch_floats */, int B,
                       int D, int Th, void* stream);
/* all adaLN projections at once: ada[b][j] = bias[j] + sum_t temb[b][t] * W[j][t],  W fp16 [J,Th]
 * (J = depth*2 norms*(gamma,beta)*D) (voicebox_pytorch.py:273). */
int vbx_adaln_proj_fwd(const float* temb, const void* w_bf16, const float* bias, float* ada, int B, int Th, int J,
                       int group /* output layout ada[j/group][b][j%group]; <=0 or J: plain [b][j] */, void* stream);
/* dW[j][t] = sum_b dada[b][j]*temb[b][t] (fp32 [J,Th]), dbias[j] = sum_b dada[b][j],
 * dtemb[b][t] = sum_j dada[b][j] * W[j][t]. */
int vbx_adaln_proj_bwd(const float* temb, const void* w_bf16, const float* dada, float* dw, float* dbias, float* dtemb,
                       float* scratch, int B, int Th, int J, int accumulate_dtemb, void* stream);
int vbx_adaln_proj_bwd_scratch_floats(int B, int Th, int J);
/* d(time_emb) of EVERY layer's projections in one launch (+ one reduction): dtemb[b][t] = sum over l, j of dada[l][b][j] * W[l][j][t];
 * W fp16 [L][J][Th] (the packed arena's order), dada fp32 [L][B][J], scratch vbx_adaln_dtemb_all_scratch_floats() floats.  With the
 * weight gradient kept in factor form (vbx_model.adaln_factors) and the bias gradient taken from the norm partial records, this is all
 * that is left of the adaLN backward: once per step instead of two launches per layer. */
long vbx_adaln_dtemb_all_scratch_floats(int L, int B, int Th, int J);
int vbx_adaln_dtemb_all(const void* w_f16, const float* dada, float* dtemb, float* scratch, int L, int B, int Th, int J, void* stream);
/* reduce rmsnorm_bwd partials over chunks: out[b][2][D] = sum_chunk part[b][chunk][2][D] */
int vbx_reduce_norm_partials(const float* part, float* out, long out_b_stride, int B, int chunks, int D, int sum_batch,
                             void* stream);
/* out[d] = sum_{b,chunk} colpart[b][chunk][d]  (the fused column sums of vbx_rmsnorm_bwd) */
int vbx_reduce_col_partials(const float* colpart, float* out, float* tmp /* B*D floats */, int B, int chunks, int D,
                            void* stream);
/* GEGLU backward on the interleaved pre-activation (voicebox_pytorch.py:338-340) */
int vbx_geglu_bwd(const void* h1_bf16, const void* dg_bf16, void* dh1_bf16, int M, int Fp, void* stream);
/* column sums: out[c] (+)= sum_r in[r][c]   (bias grads) ; rowmap as in vbx_splitk_reduce */
int vbx_colsum_bf16(const void* in_bf16, int M, int C, int ld, float* out, int out_len, int rowmap, int F,
                    float* scratch, void* stream);
int vbx_colsum_f32(const float* in, int M, int C, int ld, float* out, float* scratch, void* stream);
int vbx_colsum_scratch_floats(int M, int C);
/* out[j] (+)= sum_{i<rows} in[i*ld + j], j < cols   (partials -> gradients) */
int vbx_sum_rows_f32(const float* in, long rows, long ld, float* out, long cols, int accumulate, void* stream);
/* rows of the gpart buffer of vbx_qknorm_rope_bwd per `which`: gpart is [2][rows][H][64] */
int vbx_qknorm_rope_bwd_gpart_rows(int B);
/* masked MSE (voicebox_pytorch.py:1099-1115): loss scalar fp32; per_b: vbx_masked_mse_scratch_floats(B) floats
 * ([0,B) per-sample loss, [B,2B) denominators, then partial sums). */
int vbx_masked_mse_scratch_floats(int B);
int vbx_masked_mse_fwd(const float* pred, const float* target, const uint8_t* loss_mask, float* per_b, float* loss,
                       int B, int N, int D, void* stream);
int vbx_masked_mse_bwd(const float* pred, const float* target, const uint8_t* loss_mask, const float* per_b,
                       const float* gscale /* device scalar d(loss) or NULL (=1) */, float* dpred, void* dpred_bf16, int B,
                       int N, int D, void* stream);
/* CFM inputs (voicebox_pytorch.py:1404-1410): w = (1-(1-sigma)t)x0 + t x1 ; flow = x1-(1-sigma)x0 */
int vbx_cfm_inputs(const float* x1, const float* x0, const float* times, float sigma, float* w, float* flow, int B,
                   long per_batch, void* stream);
/* y_out = y + coef[idx] * f   (ODE midpoint axpy; coef device-resident so graphs hold no host scalars) */
int vbx_axpy_dev(const float* y, const float* f, const float* coef, int idx, float* out, long n, void* stream);
/* hipGraph-replayable ODE step helpers (replace the host-side loop of torchdiffeq.odeint, call site
 * voicebox_pytorch.py:1295): t and dt come from device tables [2*intervals] indexed by a device counter.
 * table slot 0/1 of interval i at table[2*i + slot]. */
int vbx_ode_set_time(float* times, int B, const float* table, const int* counter, int slot, void* stream);
int vbx_axpy_ctr(const float* y, const float* f, const float* table, const int* counter, int slot, float* out, long n,
                 void* stream);
int vbx_counter_add(int* counter, int inc, void* stream);
/* Sampling: every batch element of a call shares the ODE time, and the grid is known up front, so the time embedding + the adaLN
 * projections of ALL time points (voicebox_pytorch.py:1082, :273 -- a 100 MB weight stream per function evaluation at dim 512 / depth
 * 12) are evaluated once per sample() into table [2 * intervals][L][G] (G = 4 * D: gamma1 | beta1 | gamma2 | beta2 of a layer); this
 * copies the slice of time point 2 * counter + slot into the runtime's ada [L][B][G] (vbx_io.ada_table makes vbx_model_forward do it). */
int vbx_ada_select(float* ada, int L, int B, int G, const float* table, const int* counter, int slot, void* stream);
/* Occupies `stream` with one idle wave for `us` microseconds (0 .. 10000).  The sampler integrates the two halves of a batch as two
 * graphs on two streams and starts the second one ~60 us late, so that different kernels of the two forwards overlap (attention
 * beside GEMMs) instead of the same ones: 16 intervals 81.7 -> 80.1 ms (tools/sample_offset.py). */
int vbx_stream_delay(float us, void* stream);
/* fp32 -> bf16 and/or fp16 weight packing with optional row map / K padding:
 * dst[p][c] = (src row of p valid && c < src_cols) ? src[row][c] : 0 ; dst is [dst_rows, dst_cols] */
int vbx_pack_weight(const float* src, int src_rows, int src_cols, void* dst_bf16 /* or NULL */, void* dst_f16 /* or NULL */,
                    int dst_rows, int dst_cols, int rowmap, int F, void* stream);
int vbx_pack_bias(const float* src, int n, float* dst, int dst_n, int rowmap, int F, void* stream);
/* ------------------------------------------------------------------ GateLoop (optional layer, use_gateloop_layers)
 * gateloop_transformer.SimpleGateLoopLayer(dim, post_ln=True), call sites voicebox_pytorch.py:399,465-466 (third-party,
 * restated in oracle/restate.py:gateloop -- parity unpinned).  qkva = RMSNorm(x) W^T is produced by vbx_rmsnorm_fwd +
 * vbx_gemm; these entries are the gated linear scan and the post-LayerNorm.
 * scan fwd: a = sigmoid(qkva[..,2D:]); h_t = a_t h_{t-1} + qkva[..,D:2D]_t; s_t = qkva[..,:D]_t h_t.
 *   qkva [B,Np,3D] fp32, s [B,Np,D] fp32, hstate [B,Np,D] fp32 or NULL (kept for the backward).
 * scan bwd: ds [B,Np,D] fp32 -> d(qkva) bf16 [B,Np,3D] (sigmoid derivative applied). */
int vbx_gateloop_scan_fwd(const float* qkva, float* s, float* hstate, int B, int Np, int D, void* stream);
int vbx_gateloop_scan_bwd(const float* qkva, const float* hstate, const float* ds, void* dqkva_bf16, int B, int Np, int D,
                          void* stream);
/* y = LayerNorm(s) * w + bias (+ resid), rows of D (biased variance, eps as nn.LayerNorm). */
int vbx_layernorm_fwd(const float* s, const float* w, const float* bias, const float* resid, float* y, long rows, int D,
                      float eps, void* stream);
/* ds and per-16-row partial records part[B][ceil(Np/16)][2][D] (dw | dbias), to be summed by vbx_reduce_norm_partials. */
int vbx_layernorm_bwd(const float* s, const float* w, const float* dy, float* ds, float* part, int B, int Np, int D, float eps,
                      void* stream);

/* Several vbx_splitk_reduce jobs in one launch (e.g. the four weight gradients of a layer, each with its own slab region). */
#define VBX_SKR_MAX 6
typedef struct {
  const float* slabs;
  float* dst;
  float* sq;  /* NULL, or vbx_splitk_reduce_blocks(M, N) floats: sq[b] = sum of squares of the values block b stored (a fixed summation
                 order: the same bits on every run) -- the gradient-norm terms of this tensor without a second pass over it */
  int splits, M, N, dst_rows, dst_cols, dst_ld, rowmap, F, block0, pad_;
} vbx_skr_job;
typedef struct {
  vbx_skr_job job[VBX_SKR_MAX];
  int n;
} vbx_skr_jobs;
int vbx_splitk_reduce_multi(const vbx_skr_jobs* jobs, void* stream);
int vbx_splitk_reduce_blocks(int M, int N); /* blocks (= sq partials) of one job */
/* Several small column reductions in one launch: for job i, out[b][map(c)] = sum over r < rows of
 * src[b*src_bstride + r*row_stride + c], c < cols, b < batches; map = identity or the GEGLU row un-interleave (rowmap = 1, F), columns
 * mapping outside [0, dst_len) are dropped.  block0 is filled in by the library. */
#define VBX_MR_MAX 48
typedef struct {
  const float* src;
  float* dst;
  long src_bstride, dst_bstride, row_stride;
  int rows, cols, batches, dst_len, rowmap, F, block0, pad_;
} vbx_mr_job;
typedef struct {
  vbx_mr_job job[VBX_MR_MAX];
  int n;
} vbx_mr_jobs;
int vbx_multi_reduce(const vbx_mr_jobs* jobs, void* stream);
/* vbx_splitk_reduce_multi(sjobs) and vbx_multi_reduce(mjobs) as ONE launch (the end of a layer's backward); results bit-identical */
int vbx_layer_reduce(const vbx_skr_jobs* sjobs, const vbx_mr_jobs* mjobs, void* stream);
/* vbx_geglu_bwd that also writes per-slab column sums of dh1: scratch[vbx_geglu_bwd_colsum_slabs()][2*Fp] (same interleaved column
 * order as dh1; finish with vbx_multi_reduce + the GEGLU row un-interleave) -- the FeedForward[0].bias gradient without a second
 * pass over dh1 */
int vbx_geglu_bwd_colsum_slabs(void);
int vbx_geglu_bwd_colsum(const void* h1_bf16, const void* dg_bf16, void* dh1_bf16, int M, int Fp, float* scratch, void* stream);
/* first stage of vbx_colsum_bf16 only: scratch[vbx_colsum_slabs()][C] partial column sums (finish with vbx_multi_reduce) */
int vbx_colsum_bf16_partials(const void* in_bf16, int M, int C, int ld, float* scratch, void* stream);
int vbx_colsum_slabs(void);

/* fused Adam (torch.optim.Adam semantics, no weight decay/amsgrad) over a flat fp32 buffer; grads are
 * pre-multiplied by *gscale (device scalar, e.g. clip coefficient) if non-NULL. */
int vbx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  int step, const float* gscale, void* stream);
/* Adam over the flat buffer that ALSO refreshes the packed fp16/bf16 operand copies of the weights it updates, so the
 * next forward needs no vbx_model_pack_weights pass (62 pack launches and a second read of every parameter per step).
 * The flat buffer is described by segments [off, off+count) that tile [0, n): plain ones (dst all NULL) and packed ones
 * (a [rows, cols] weight written to dst_bf16 / dst_f16 with row stride dst_ld, or a bias copied to dst_f32; rowmap = 1
 * applies the GEGLU row interleave of vbx_pack_weight).  vbx_model_adam_segments() writes the table of a model into
 * HOST memory (returns the count, or a negative VBX_E* code; call with out = NULL to size it); the caller keeps a DEVICE
 * copy and hands it to vbx_adam_step_packed(). */
typedef struct {
  long off, count;  /* floats, inside the flat buffer */
  void* dst_bf16;
  void* dst_f16;
  float* dst_f32;
  int cols, dst_ld, rowmap, F;
  long block0;      /* first 2048-element block of this segment in the launch grid */
} vbx_adam_seg;
int vbx_adam_step_packed(float* p, const float* g, float* m, float* v, const vbx_adam_seg* segs_dev, int nsegs,
                         long total_blocks, float lr, float beta1, float beta2, float eps, int step, const float* gscale,
                         void* stream);
/* sum of squares of a flat buffer -> out[0] (two-stage, deterministic) ; scratch >= 1024 floats */
int vbx_sumsq(const float* x, long n, float* out, float* scratch, void* stream);
/* ---- adaLN projection weight gradients in FACTOR form (round 5; vbx_model.adaln_factors).  The gradient of a layer's adaLN weight
 * block W_l [J4 = 4 D, Th] (voicebox_pytorch.py:256-276: to_gamma / to_beta of the two AdaptiveRMSNorms, contiguous) is
 * dada_l^T . temb with dada_l [B, J4] and temb [B, Th] -- half of all parameters, defined by B * (J4 + Th) numbers.  In factor
 * mode the backward entry points leave that block of the gradient buffer UNWRITTEN and the optimizer works from the factors:
 *   vbx_sumsq_adaln_factors : the L * B * B terms (dada_l[b] . dada_l[b']) (temb[b] . temb[b']) whose sum over (b, b') is
 *                             |dada_l^T . temb|_F^2 -> out[(l * B + b) * B + b'];
 *   vbx_sumsq_ranges        : sum of squares over n <= 64 ranges [lo, hi) of x (host array of 2 n longs, multiples of 4 floats)
 *                             plus n_extra values already stored at scratch[1024 ..) -> out[0]; scratch >= 1024 + n_extra floats;
 *   vbx_adam_adaln_factors  : torch.optim.Adam (as vbx_adam_step) on the L blocks at flat offsets w_off[l] with the gradient
 *                             expanded on the fly, refreshing the fp16 operand copies dst_f16[l] ([J4][Th]; may be NULL).
 * vbx_model_adaln_factors gives the factor pointers and the block offsets of a model's training arena. */
int vbx_sumsq_adaln_factors(const float* dada /* [L][B][J4] */, const float* temb /* [B][Th] */, int L, int B, int J4, int Th,
                            float* out /* [L * B * B] */, void* stream);
/* dw [J4, Th] = dada^T . temb for any B (the data-parallel exchange gathers every rank's factors and expands the summed gradient
 * locally: dp.GradBucketReducer(adaln_factors=...)) */
int vbx_adaln_expand_dw(const float* temb /* [B][Th] */, const float* dada /* [B][J4] */, float* dw, int B, int Th, int J4, int reserved,
                        void* stream);
int vbx_sumsq_ranges(const float* x, const long* ranges /* host [2 n] */, int n, int n_extra, float* out, float* scratch, void* stream);
int vbx_adam_adaln_factors(float* p, float* m, float* v, const long* w_off /* host [L] */, void* const* dst_f16 /* host [L] or NULL */,
                           const float* dada, const float* temb, int L, int B, int J4, int Th, float lr, float beta1, float beta2,
                           float eps, int step, const float* gscale, void* stream);
/* gradient-clip coefficient for a buffer holding the SUM over `world` ranks (inv_world = 1/world):
 * norm = sqrt(sumsq)*inv_world; coef[0] = min(1, max_norm/(norm+1e-6)) * inv_world (max_norm <= 0: no clipping);
 * coef[1] = norm.  (accelerator.clip_grad_norm_, trainer.py:274-275) */
int vbx_clip_coef(const float* sumsq, float max_norm, float inv_world, float* coef /* [2] */, void* stream);

/* ------------------------------------------------------------------ stage-level runtime
 * The whole VoiceBox forward / backward as native sequences of launches (no host sync, no allocation:
 * hipGraph-capturable).  Replaces VoiceBox.forward (voicebox_pytorch.py:987-1115) incl.
 * Transformer.forward (:412-479) and, for the backward entry points, their autograd graph.
 * The caller (Python) owns three arenas: flat fp32 parameters (+ same-layout gradients), the packed
 * bf16 weight arena and the activation arena (sizes from the *_bytes queries). */
enum { VBX_P_SINW = 0, VBX_P_T1W, VBX_P_T1B, VBX_P_EMBW, VBX_P_EMBB, VBX_P_CONVW, VBX_P_CONVB, VBX_P_REG, VBX_P_FNG,
       VBX_P_PREDW,
       VBX_P_CEMB /* to_cond_emb.weight [num_cond_tokens + 1, E], read only when vbx_model.E > 0 */, VBX_NG };
/* per layer; the four adaLN weights, and the four adaLN biases, must be contiguous in this order; so must the GateLoop
 * post-LayerNorm weight and bias (GLLNW, GLLNB).  The four GL* slots are read only when vbx_model.gateloop != 0. */
enum { VBX_L_G1W = 0, VBX_L_B1W, VBX_L_G2W, VBX_L_B2W, VBX_L_G1B, VBX_L_B1B, VBX_L_G2B, VBX_L_B2B, VBX_L_QG, VBX_L_KG,
       VBX_L_QKVW, VBX_L_OUTW, VBX_L_FF1W, VBX_L_FF1B, VBX_L_FF2W, VBX_L_FF2B, VBX_L_GLG, VBX_L_GLW, VBX_L_GLLNW, VBX_L_GLLNB,
       VBX_L_N1G, VBX_L_N2G /* plain RMSNorm gammas, read only when vbx_model.plain_norm != 0 */,
       VBX_L_SKW, VBX_L_SKB /* u-net skip combiner Linear(2 * dim, dim) of the second-half layers, read only when vbx_model.unet != 0 */,
       VBX_NL };

typedef struct {
  int B, N, R, D, H, F, Th, L, ksize;
  int qk_norm;            /* attn_qk_norm (voicebox_pytorch.py:897) */
  float attn_scale;       /* Attend scale: 10 with qk-norm, dim_head^-0.5 otherwise (:304, attend.py:111) */
  int training;           /* 1: keep every activation needed by the backward entry points */
  float* params;          /* flat fp32 master parameters */
  float* grads;           /* flat fp32 gradients, same offsets (NULL in eval) */
  const long* off;        /* HOST array [VBX_NG + L*VBX_NL] of offsets (in floats) into params/grads */
  void* wpack;            /* packed bf16 weight arena */
  void* act;              /* activation arena */
  const float* rot_cos;   /* [N+R,32] host-built rotary tables (voicebox_pytorch.py:184-191,436-443) */
  const float* rot_sin;
  int gateloop;           /* use_gateloop_layers (:898): x = GateLoop(x) + x in front of every attention block (:465-466) */
  int stack_only;         /* 1: standalone Transformer.forward (:412-479): io->x is the stack input [B,N,D], io->cond the adaptive
                             norm condition [B,Th] (unused with plain_norm), io->pred the final-norm output [B,N,D]; the backward
                             entry points take d(output) in io->target and write io->dx / io->dcond */
  int E;                  /* dim_cond_emb of a text-conditioned model (condition_on_text, :931-940), 0 = unconditional:
                             to_embed is Linear(2*D + E, D) over [x | cond_emb | cond] (:1071-1076) */
  int V1;                 /* rows of the conditioning embedding table (num_cond_tokens + 1) */
  int plain_norm;         /* 1: non-adaptive RMSNorm (adaptive_rmsnorm = False, :386-389): gammas at VBX_L_N1G / VBX_L_N2G */
  float attn_dropout;     /* attn_dropout (:895, attend.py:131) and ff_dropout (:891, :346) of the module: applied in a forward whose */
  float ff_dropout;       /* vbx_io.dropout != 0 (nn.Dropout: module.training), keyed by vbx_io.drop_seed; with attn_dropout > 0 the
                             arena holds the attention keep bits (per layer when training != 0, for the backward) */
  int Din;                /* dim_in (:884,905): width of x / cond / target / pred and of null_cond; to_embed is Linear(2*Din + E, D)
                             (:938), to_pred Linear(D, Din) (:964-966).  0 = D.  Multiple of 8. */
  int precise;            /* 1: exact-operand forward (see "precise mode" below): every forward matrix product to fp32 accuracy;
                             needs wpack3 (vbx_model_precise_wpack_bytes, filled by vbx_model_pack_weights_precise) and pscratch
                             (vbx_model_precise_scratch_bytes).  The backward entry points are unchanged (bf16 operands). */
  void* wpack3;           /* precise mode: hi/lo-split fp16 weights, K-concatenated */
  void* pscratch;         /* precise mode: fp32 intermediates + the K-concatenated activation operand */
  int unet;               /* use_unet_skip_connection (voicebox_pytorch.py:368-369,391-398,453-463; stack_only models: VoiceBox never
                             enables it): layer l >= L / 2 starts with x = Linear(2 * dim, dim)(cat(x, skip_scale * input of layer L-1-l)) */
  float skip_scale;       /* skip_connect_scale (:390), 2^-0.5 by default */
  int adaln_factors;      /* 1 (training, adaptive norms): vbx_model_backward_layer does NOT write the gradients of the adaLN projection
                             WEIGHTS (slots VBX_L_G1W .. VBX_L_B2W) -- they stay in factor form (vbx_model_adaln_factors; the
                             optimizer expands them, see vbx_adam_adaln_factors) and vbx_model_adam_segments leaves those blocks
                             out of the fused Adam's table.  Biases, d(time_emb) and every other gradient are unchanged. */
  int defer_reduce;       /* 1: the caller runs vbx_model_backward_layer for L-1 .. 0 and reads NO gradient before layer 0 has
                             returned (no per-stage gradient exchange): every layer keeps its partial records (norm gamma / beta, bias
                             column sums, qk-norm gammas) in its own arena region and layer 0 reduces them all in two launches
                             instead of one launch per layer.  0 (default): each layer's small gradients are final when it returns.
                             Same per-tensor summation order either way. */
  float* sq_partials;     /* NULL, or vbx_model_sq_partials(m, NULL) floats of device memory: the slab reduce of every layer's four
                             weight-gradient matrices (to_qkv, to_out, FeedForward in / out) also leaves the sums of squares of what it
                             stored, one float per block -- the gradient norm then needs no second pass over those tensors (their
                             flat ranges: vbx_model_sq_partials).  Meaningful only when the norm is taken over THIS backward's
                             gradient (no accumulation, no exchange in between). */
} vbx_model;

typedef struct {
  const float* x;               /* [B,N,D]  (w in training, y in sampling) */
  const float* cond;            /* [B,N,D] */
  const uint8_t* cond_mask;     /* [B,N] 1 = frame is to be infilled (conditioning zeroed there, :1035) */
  const uint8_t* attn_mask;     /* [B,N] self_attn_mask or NULL */
  const uint8_t* attn_mask_p;   /* [B,N+R] = attn_mask left-padded with True for the registers (:428), or NULL */
  const uint8_t* loss_mask;     /* [B,N] cond_mask & attn_mask (:1099); required when target is given */
  const float* times;           /* [B] */
  const float* target;          /* [B,N,D] or NULL */
  float* pred;                  /* [B,N,D] output */
  float* loss;                  /* [1] output when target != NULL */
  const long* cond_ids;         /* E > 0: [B,T] conditioning token ids (int64) */
  int T;                        /* E > 0: tokens per sample */
  long null_id;                 /* E > 0: id substituted where drop_mask is set (null_cond_id, :933) */
  const uint8_t* drop_mask;     /* E > 0: [B] classifier-free-guidance drop mask (:1040-1053) or NULL */
  const float* null_cond;       /* E > 0: [D] null_cond parameter (:944), required with drop_mask */
  float* dx;                    /* stack_only backward: [B,N,D] gradient of the stack input */
  float* dcond;                 /* stack_only backward: [B,Th] gradient of the adaptive-norm condition (NULL with plain_norm) */
  int dropout;                  /* 1: apply the model's attn_dropout / ff_dropout in this forward (the module is in train() mode) */
  unsigned long long drop_seed; /* Philox key of this forward's masks (the backward entry points must see the same io) */
  const float* ada_table;       /* inference only, or NULL: precomputed adaLN projections [2 * intervals][L][4 * D] (vbx_ada_select); */
  const int* ada_counter;       /* the forward then skips the time embedding and the projection GEMV and takes time point         */
  int ada_slot;                 /* 2 * ada_counter[0] + ada_slot of the table (`times` is not read)                              */
} vbx_io;

size_t vbx_model_wpack_bytes(const vbx_model* m);
size_t vbx_model_act_bytes(const vbx_model* m);
int vbx_model_pack_weights(const vbx_model* m, void* stream);
/* factor form of the adaLN weight gradients of the last backward (valid until the next forward of this arena): dada [L][B][4 D],
 * temb [B][Th]; w_off[l] = flat offset of layer l's weight block, dst_f16[l] = its fp16 operand copy in the wpack arena (HOST arrays
 * of L entries each, either may be NULL) */
int vbx_model_adaln_factors(const vbx_model* m, const float** dada, const float** temb, long* w_off, void** dst_f16);
/* floats vbx_model.sq_partials must hold (0: this configuration does not serve it), and -- ranges != NULL -- the 4 * L flat ranges
 * [lo, hi) of the gradient buffer they cover, in ascending order */
long vbx_model_sq_partials(const vbx_model* m, long* ranges /* host [4 * L][2] or NULL */);
/* segment table for vbx_adam_step_packed (see there) */
int vbx_model_adam_segments(const vbx_model* m, long n_flat, vbx_adam_seg* out, int max_segs, long* total_blocks);
int vbx_model_forward(const vbx_model* m, const vbx_io* io, void* stream);
/* adaLN projections of n <= 16 conditioning rows temb [n, Th] with the model's packed weights -> ada [L][n][4 * D] (what the forward
 * computes per call; the sampler tabulates it over its time grid, see vbx_ada_select) */
int vbx_model_adaln_table(const vbx_model* m, const float* temb, int n, float* ada, void* stream);
/* backward: head (loss, to_pred, final norm) -> layers L-1..0 -> embed (conv, to_embed, time MLP).  Each call
 * finishes the gradients of its own parameters, so the caller can all-reduce them while the next runs. */
int vbx_model_backward_head(const vbx_model* m, const vbx_io* io, const float* gscale, void* stream);
int vbx_model_backward_layer(const vbx_model* m, const vbx_io* io, int layer, void* stream);
int vbx_model_backward_embed(const vbx_model* m, const vbx_io* io, void* stream);

/* tests/debug only: device pointer of a named tensor inside the activation arena (NULL if unknown) */
void* vbx_model_debug_ptr(const vbx_model* m, const char* name, int layer);

/* ------------------------------------------------------------------ precise mode (exact-operand forward)
 * The fast path rounds every forward GEMM / attention operand to fp16; at the reference's own initialisation (qk-normed logits of
 * std ~80, a chaotic 12-layer map) that moves the loss by O(1e-3).  With vbx_model.precise = 1 vbx_model_forward evaluates
 * the same VoiceBox.forward (voicebox_pytorch.py:987-1115) with every matrix product to fp32 accuracy:
 *  - nn.Linear (:320,333,345,348,1078,1092): the same vbx_gemm tiles with both operands split into fp16 hi + lo parts and concatenated
 *    along K -- A' = [A_hi | A_hi 2^-8 | A_lo 2^8] (vbx_split3_f16), W' = [W_hi | W_lo 2^8 | W_hi 2^-8] (vbx_pack_weight3), K' = 3K,
 *    VBX_EPI_F32 (the powers of two keep the lo parts of small weights out of the fp16 subnormals);
 *  - MultiheadRMSNorm + rotary (:286-287,193-199, 323-328) and GEGLU (:338-340) as fp32 kernels on the fp32 GEMM results
 *    (vbx_qknorm_rope_f32, vbx_geglu_f32), which also write the fp16 / bf16 copies the backward entry points read;
 *  - Attend (attend.py:121-135) as an fp32 FMA flash kernel (vbx_attn_fwd_f32), q / k / v / P never rounded;
 *  - AdaptiveRMSNorm's to_gamma / to_beta (:273) from the fp32 master weights (vbx_adaln_proj_f32);
 *  - text conditioning (:1056-1078): the fp32 rows of vbx_embed_input_text_f32 through the same split GEMM (K = 2 * Din + E);
 *  - GateLoop (:465-466): its to_qkva projection through the split GEMM, scan and LayerNorm are fp32 on both paths;
 *  - dropout (attend.py:131, :346): the fast path's Philox masks (vbx_attn_dropout_bits, vbx_dropout_rows[_f32]) on the unrounded
 *    probabilities / GEGLU output, so a precise step and a fast step with the same seed drop the same elements.
 * Serves VoiceBox (not the standalone Transformer stack, u-net skips or plain RMSNorm); those return VBX_EINVAL. */
/* dst fp16 [rows, 3*Kp] = [hi | hi 2^-8 | lo 2^8] of src fp32 [rows, K] (row stride ld floats), columns K..Kp zero; Kp % 8 == 0 */
int vbx_split3_f16(const float* src, long rows, int K, long ld, void* dst_f16, int Kp, void* stream);
/* dst fp16 [dst_rows, 3*dst_cols] = [hi | lo 2^8 | hi 2^-8] of the weight, rows mapped / padded as vbx_pack_weight does */
int vbx_pack_weight3(const float* src, int src_rows, int src_cols, void* dst_f16, int dst_rows, int dst_cols, int rowmap, int F,
                     void* stream);
/* raw fp32 [B*Np, 3*H*64] (to_qkv output) -> q, k (qk-normed when qk_scale > 0, rotated) and v, head-major [B,H,Np,64]: fp32 plus the
 * optional fp16 / bf16 copies and 1/max(|.|,1e-12) rows the backward reads (any of q16 .. k_rnorm may be NULL) */
int vbx_qknorm_rope_f32(const float* raw, int B, int H, int Np, float qk_scale, const float* q_gamma, const float* k_gamma,
                        const float* rot_cos, const float* rot_sin, float* q32, float* k32, float* v32, void* q16, void* k16,
                        void* qb, void* kb, void* v_bf16, void* v16, float* q_rnorm, float* k_rnorm,
                        float q16_scale /* q16 = q * q16_scale: vbx_attn_q_prescale(scale), the attention kernels' contract */,
                        void* stream);
/* attend.py:121-135 in fp32: q, k, v fp32 [B,H,Np,64] -> out32 fp32 [B,Np,H*64] (+ optional fp16 / bf16 copies, log2-LSE [B,H,Np]) */
int vbx_attn_fwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask, float* out32, void* out16, void* out_bf16,
                     float* lse, int B, int H, int Np, float scale, void* stream);
/* ... with attention dropout: bits_rm / p as in vbx_attn_fwd_dropout (the statistics stay those of the undropped probabilities) */
int vbx_attn_fwd_f32_dropout(const float* q, const float* k, const float* v, const uint8_t* mask, float* out32, void* out16,
                             void* out_bf16, float* lse, int B, int H, int Np, float scale, const void* bits_rm, float p, void* stream);
/* GEGLU (libm erff) on the fp32 pre-activation h1 [M, 2*Fp] in the packed column order (128-column blocks: 64 "x", their 64 "gate"):
 * g32 [M, Fp] (+ optional fp16 / bf16 copies of g and the bf16 pre-activation the backward reads) */
int vbx_geglu_f32(const float* h1, float* g32, void* g16, void* g_bf16, void* h1_bf16, long M, int Fp, void* stream);
/* vbx_adaln_proj_fwd with fp32 weights W [J, Th] */
int vbx_adaln_proj_f32(const float* temb, const float* w, const float* bias, float* ada, int B, int Th, int J, int group, void* stream);
size_t vbx_model_precise_wpack_bytes(const vbx_model* m);
size_t vbx_model_precise_scratch_bytes(const vbx_model* m);
int vbx_model_pack_weights_precise(const vbx_model* m, void* stream);

/* ------------------------------------------------------------------ in-situ stage timing (measurement only)
 * While enabled, vbx_model_forward / vbx_model_backward_* bracket each MFMA stage of a layer (to_qkv, attention forward, to_out,
 * ff_in, ff_out, the four dgrads, attention backward, the weight-gradient launch) with a pair of HIP events recorded on the
 * caller's stream, i.e. the launches are timed where they run -- between their real neighbours -- not back to back in
 * isolation.  vbx_prof_collect synchronises on the recorded events, aggregates by label and disables the recording.
 * Not hipGraph-capture safe: do not enable around a capture.  bench.py's roofline.kernels come from here. */
typedef struct {
  char label[24];
  int calls;
  float total_us;
} vbx_prof_entry;
int vbx_prof_enable(int on);
int vbx_prof_collect(vbx_prof_entry* out, int max_entries); /* returns the number of labels (<= max_entries) */

/* ------------------------------------------------------------------ hardware probes (tests only) */
int vbx_probe_tr16(const void* in_u16_4096, const int* lane_elem_off, void* out_u16_256, void* stream);
int vbx_probe_mfma(int which, const float* a, const float* b, float* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VBX_H */
