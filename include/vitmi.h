/* vitmi.h -- C ABI of libvitmi_hip.so: the gfx950 (MI355X) kernels behind the ViT encoder
 * hot path of mrkize/VisionTransformer (mymodel/ViT.py, mymodel/ViT_mask.py).
 *
 * The reference is pure Python on PyTorch and has no FFI of its own; each entry point below
 * replaces the aten kernels that one group of reference lines dispatches, and says which.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error (see VITMI_ERR_*); the message of the
 *     last error on the calling thread is vitmi_last_error().  Nothing throws.
 *   - all pointers are BORROWED device pointers (HBM) unless a comment says "host"; the
 *     library never frees or keeps them.  Temporary storage is a caller-provided workspace
 *     whose size comes from the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it, never synchronised.
 *   - `dt` is the activation/compute storage type: VITMI_F32 or VITMI_BF16.  Parameters that
 *     stay in fp32 in every mode (biases, LayerNorm affine, cls token, positional table) are
 *     typed `const float*`.  Accumulation is always fp32.
 *   - matrices are row-major; "ld" = elements between consecutive rows.
 */
#ifndef VITMI_H
#define VITMI_H

#ifdef __cplusplus
extern "C" {
#endif

#define VITMI_VERSION 100

#define VITMI_F32 0
#define VITMI_BF16 1

#define VITMI_OK 0
#define VITMI_ERR_BAD_ARG (-1)
#define VITMI_ERR_HIP (-2)
#define VITMI_ERR_UNSUPPORTED (-3)

/* GEMM epilogues (vitmi_gemm `epilogue`) */
#define VITMI_EPI_NONE 0  /* C = acc [+ bias[n]] [+ residual(m,n)]                                  */
#define VITMI_EPI_GELU 1  /* pre = acc + bias[n]; aux(m,n) = pre (if aux); C = gelu_erf(pre)        */
#define VITMI_EPI_DGELU 2 /* C = acc * gelu_erf'(aux(m,n))   (aux = the pre-activation saved above) */
/* The same pair with the derivative evaluated in the FORWARD epilogue, where erf and exp of the pre-activation
 * are at hand anyway; the backward epilogue is then one multiply per element instead of an erf evaluation: */
/* aux of this pair is private to it: fp32 in fp32 mode, IEEE half bits (not bf16) in bf16 mode, since the
 * derivative lies in [-0.13, 1.13] and half keeps 11 significant bits of it. */
#define VITMI_EPI_GELU_D 3 /* pre = acc + bias[n]; aux(m,n) = gelu_erf'(pre) (if aux); C = gelu_erf(pre)  */
#define VITMI_EPI_MULAUX 4 /* C = acc * aux(m,n)                                                           */
/* LayerNorm folded into the product (vitmi_gemm_ln only): A holds the raw rows x, B the weight times gamma;
 * C = rstd[m] * (acc - mean[m] * s[n]) + bias[n], s[n] = sum_k B[n][k], bias = W beta + b; _GELU: gelu_erf of that */
#define VITMI_EPI_LN 5
#define VITMI_EPI_LN_GELU 6

/* kernel families for the launch timers (vitmi_prof_*) */
#define VITMI_FAM_GEMM 0
#define VITMI_FAM_ATTN_FWD 1
#define VITMI_FAM_ATTN_BWD 2
#define VITMI_FAM_LAYERNORM 3
#define VITMI_FAM_TOKENS 4
#define VITMI_FAM_OTHER 5
/* one family per bf16 GEMM kernel instantiation: VARIANT0 + 4*layout + epilogue/output, with
 * layout = 2*(A stored [k][m]) + (B stored [k][n]) and epilogue/output in
 * {plain->bf16, plain->f32, gelu->bf16, dgelu->bf16} */
#define VITMI_FAM_GEMM_VARIANT0 8
/* aligned fast paths: nt (both operands contiguous along k) x the same four epilogue/output
 * forms, and tn (weight gradients, split over tokens) */
#define VITMI_FAM_GEMM_NT_FAST0 24
#define VITMI_FAM_GEMM_TN_FAST 28
#define VITMI_FAM_COUNT 29

int vitmi_version(void);
const char* vitmi_last_error(void);

/* Launch timers: when enabled, every launch of a kernel family is bracketed by HIP events on
 * its own stream.  vitmi_prof_read synchronises those events and returns the summed duration
 * (ms), the summed algorithmic work the launches declared (FLOPs for GEMM/attention, bytes
 * for the HBM-bound families) and the launch count.  Host pointers.  No reference counterpart
 * (the reference times whole epochs with time.time(), trainer.py:98,159-161). */
int vitmi_prof_enable(int on);
int vitmi_prof_reset(void);
/* Time only one family (-1: all).  Two event records per launch cost a few microseconds of GPU time each: with
 * every family timed a ViT-B/16 step runs 3 % slower and a 32x32-pixel step 25 % slower, so a benchmark times
 * everything during warm-up and only the kernel it reports on inside its timed region. */
int vitmi_prof_select(int family);
/* Time every `every`-th launch of the timed families (default 1 = all): the averages stay per launch, the GPU time
 * spent on event records drops by the same factor.  Pick a stride coprime to the period of the launch sequence. */
int vitmi_prof_stride(int every);
int vitmi_prof_read(int family, double* total_ms, double* total_work, long long* launches);
const char* vitmi_prof_name(int family);

/* dst[i] = (dst_dt) src[i].  Replaces Tensor.to(dtype) on parameters/inputs. */
int vitmi_cast(const void* src, int src_dt, void* dst, int dst_dt, long long n, void* stream);

/* C(m,n) = sum_k A(m,k) * B(n,k) with a fused epilogue; A(m,k) = A[m*sam + k*sak],
 * B(n,k) = B[n*sbn + k*sbk] (so nn.Linear's y = x W^T is sak = sbk = 1, and the two backward
 * products are strided views of the same call).  A and B are `dt`; C is `c_dt` (dt or
 * VITMI_F32); bias is fp32 [N] or NULL; residual is `dt` with row stride ldr or NULL; aux is
 * `dt` with row stride ldaux or NULL.
 * Replaces: nn.Linear forward/backward (mymodel/ViT.py:66,69,86,89,148), the residual adds
 * (:119-120) and nn.GELU (:67). */
int vitmi_gemm(int dt, int M, int N, int K, const void* A, long long sam, long long sak, const void* B,
               long long sbn, long long sbk, void* C, long long ldc, int c_dt, int epilogue, const float* bias,
               const void* residual, long long ldr, void* aux, long long ldaux, void* stream);

/* vitmi_gemm that also returns colsum[n] = sum_m of the values it writes to C[:, n] (fp32, N
 * entries): with the dGELU epilogue this is the bias gradient of the first MLP layer
 * (autograd of nn.Linear + nn.GELU, mymodel/ViT.py:66-67) without a second pass over the
 * [tokens, mlp_dim] gradient.  On the large aligned shapes the sums come out of the GEMM
 * epilogue (per 128-row partial sums, reduced in a fixed order); otherwise a column-sum kernel
 * runs over C.  workspace: vitmi_gemm_colsum_workspace_bytes(M, N) bytes. */
long long vitmi_gemm_colsum_workspace_bytes(int M, int N);
int vitmi_gemm_colsum(int dt, int M, int N, int K, const void* A, long long sam, long long sak, const void* B,
                      long long sbn, long long sbk, void* C, long long ldc, int c_dt, int epilogue,
                      const float* bias, const void* residual, long long ldr, void* aux, long long ldaux,
                      float* colsum, void* workspace, void* stream);

/* Weight gradient C[p][q] (fp32, row stride ldc) = sum_t U[t][p] * V[t][q]: U is [T,P] (row
 * stride ldu), V is [T,Q] (row stride ldv), both `dt`.  For y = x W^T: dW = wgrad(U = dy, V = x).
 * The contraction over tokens is split into slabs summed in a fixed order (deterministic);
 * workspace holds the slabs.  Replaces the weight-gradient GEMM of nn.Linear's backward. */
long long vitmi_gemm_wgrad_workspace_bytes(int dt, int P, int Q, int T);
int vitmi_gemm_wgrad(int dt, int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv,
                     float* C, long long ldc, void* workspace, void* stream);

/* Many weight gradients over ONE token axis in one launch per 16 jobs: C_j[P_j][Q_j] (fp32) = U_j^T V_j, U_j [T, P_j]
 * and V_j [T, Q_j] bf16 row-major with leading dimensions ldu, ldv (every `x.grad`-side product `dW = dy^T x` of the
 * nn.Linear layers, mymodel/ViT.py:65-68,86,88, gathered after the backward pass instead of one small launch each
 * beside it).  jobs is a HOST array.  Contract per job: the three pointers 16-byte aligned, P, Q, ldu, ldv multiples
 * of 8, ldc a multiple of 4.  workspace: vitmi_gemm_wgrad_multi_workspace_bytes(jobs, n, T) bytes (fp32 partials of
 * the token slabs, summed in slab order; NULL: one slab).  vitmi_ff_block_bwd / vitmi_attn_block_bwd called with
 * both weight-gradient pointers NULL leave these products to the caller; their du / dqkv operands then sit at
 * vitmi_ff_block_bwd_du_offset / vitmi_attn_block_bwd_dqkv_offset bytes into the workspace of that call. */
typedef struct {
    const void* U;
    long long ldu;
    const void* V;
    long long ldv;
    float* C;
    long long ldc;
    int P;
    int Q;
} vitmi_wgrad_job;
long long vitmi_gemm_wgrad_multi_workspace_bytes(const vitmi_wgrad_job* jobs, int n, long long T);
int vitmi_gemm_wgrad_multi(const vitmi_wgrad_job* jobs, int n, long long T, void* workspace, void* stream);
long long vitmi_ff_block_bwd_du_offset(int dt, int rows, int D, int M);
long long vitmi_attn_block_bwd_dqkv_offset(int dt, int B, int N, int D, int H, int DH);

/* The vector gradients of many sub-blocks in one launch per 32 jobs (the bias and LayerNorm-parameter gradients of
 * nn.Linear / nn.LayerNorm's backward, mymodel/ViT.py:55,65-68,88, gathered after the backward pass like the weight
 * gradients above).  A one-launch sub-block backward leaves per-strip partial sums, `strips` rows of
 * [db1 (M) | dgamma (D) | dbeta (D) | db2 (D)] fp32; a job sums them over the strips in strip order (the same order and
 * arithmetic as the reduction the sub-block call runs itself).  M == 0 and db1 == NULL for the attention sub-block
 * (db2 = the output projection's bias gradient, NULL: none).  jobs is a HOST array.
 * vitmi_ff_block_bwd / vitmi_attn_block_bwd called with dgamma == NULL (then every other vector-gradient pointer and
 * both weight-gradient pointers must be NULL too) leave these sums to the caller: the partial sums sit
 * vitmi_*_block_bwd_sums_offset bytes into that call's workspace, vitmi_*_block_bwd_sums_strips rows of them.  A strips
 * value of 0 says that the shape does not run as the one-launch kernel: such a call with dgamma == NULL is refused
 * (VITMI_ERR_BAD_ARG, nothing enqueued), and so is one whose operands miss the 16-byte alignment of that kernel. */
typedef struct {
    const float* partial;
    int strips;
    int M;
    int D;
    float* db1;
    float* dgamma;
    float* dbeta;
    float* db2;
} vitmi_sums_job;
int vitmi_small_sums_multi(const vitmi_sums_job* jobs, int n, void* stream);
long long vitmi_ff_block_bwd_sums_offset(int dt, int rows, int D, int M);
int vitmi_ff_block_bwd_sums_strips(int dt, int rows, int D, int M);
long long vitmi_attn_block_bwd_sums_offset(int dt, int B, int N, int D, int H, int DH);
int vitmi_attn_block_bwd_sums_strips(int dt, int B, int N, int D, int H, int DH);

/* Two weight gradients over the SAME T tokens, C0 [P0,Q0] = U0^T V0 and C1 [P1,Q1] = U1^T V1 (both dense fp32):
 * the two nn.Linear weight gradients of one pre-norm sub-block (mymodel/ViT.py:66,69 resp. :86,89), which autograd
 * computes as two GEMMs.  In bf16 with every extent a multiple of 256 and T a multiple of 64 both run in ONE launch
 * that fills the compute units once (token slabs per output tile plus a shared remainder, fixed-order sum of the
 * partial tiles: deterministic); any other case is two vitmi_gemm_wgrad calls.  workspace:
 * vitmi_gemm_wgrad_pair_workspace_bytes. */
long long vitmi_gemm_wgrad_pair_workspace_bytes(int dt, int T, int P0, int Q0, int P1, int Q1);
int vitmi_gemm_wgrad_pair(int dt, int T, int P0, int Q0, const void* U0, long long ldu0, const void* V0,
                          long long ldv0, float* C0, int P1, int Q1, const void* U1, long long ldu1, const void* V1,
                          long long ldv1, float* C1, void* workspace, void* stream);
/* 1 (default): vitmi_gemm_wgrad_pair may use the one-launch form; 0: always two launches.  Process-wide;
 * benchmarking. */
int vitmi_debug_set_tn_batch(int on);

/* Tuning hook: force the block-tile configuration of the aligned nt GEMM (0 = 128x128, 1 = 256x128,
 * 2 = 256x256; -1 = automatic).  Process-wide; for benchmarking only. */
int vitmi_debug_set_nt_config(int cfg);
/* Tile walk of the persistent 256x256 nt kernel: gm > 0 = groups of gm row panels with the row index moving
 * fastest inside a group (keeps the group's A panels in the XCD's L2 while it sweeps the weight), 0 = column index
 * fastest, -1 = automatic.  Process-wide; for benchmarking only. */
int vitmi_debug_set_nt_group(int gm);
/* Start-up stagger of the persistent nt kernel's blocks, in ticks of the 100 MHz constant clock: blocks that own
 * one tile fewer than the busiest ones (all_blocks != 0: every block) start up to span (by position inside the
 * XCD) + xspan (by XCD) ticks late.  0, 0 = off.  Process-wide; for benchmarking only. */
int vitmi_debug_set_nt_stagger(int span_ticks, int xspan_ticks, int all_blocks);
/* How the bf16 MLP block evaluates nn.GELU() (mymodel/ViT.py:67, the exact erf form) and its derivative; the fp32
 * kernels always call erff.  Process-wide; leave it alone between a forward pass and its backward pass.
 *   0 "fast" (default): VITMI_EPI_GELU_D / VITMI_EPI_MULAUX.  The forward epilogue evaluates gelu and gelu' together
 *      through Phi(x) ~ 1 / (1 + exp(-x q(x^2))), q a fitted even polynomial: |gelu error| <= 3.8e-5,
 *      |gelu' error| <= 9.3e-5 against the erf forms, before the values are rounded to bf16 (8 significant bits,
 *      relative 3.9e-3) and IEEE half (11 bits); the backward multiplies by the stored derivative.
 *   1 "erf": VITMI_EPI_GELU / VITMI_EPI_DGELU.  gelu through Abramowitz-Stegun 7.1.26 erf (|error| <= 1.5e-7), the
 *      pre-activation is stored in bf16 and the backward evaluates the erf-form derivative from it.
 * The Python side exposes it as visiontransformer_amd.set_gelu("fast" | "erf") and model.set_gelu(...). */
int vitmi_set_gelu_mode(int erf_pair);
int vitmi_get_gelu_mode(void);
int vitmi_debug_set_gelu_pair(int classic);     /* older name of vitmi_set_gelu_mode */
/* 1 (default): when the persistent nt kernel's last round would fill less than half of the compute units, the
 * rows of that round run as a second launch of the 128 x 128 kernel; 0: one launch.  Process-wide; benchmarking. */
int vitmi_debug_set_nt_tail_split(int on);
/* 1 (default): bf16 attention over short sequences (<= 96 tokens) of narrow heads (<= 16 wide) gives a workgroup six
 * adjacent heads of an image when the head count is a multiple of six; 0: one (image, head) pair per workgroup for
 * every shape.  Same results bit for bit.  Process-wide; benchmarking. */
int vitmi_debug_set_attn_head_groups(int on);
/* The bf16 attention backward for heads of 64 and 129..208 tokens.  1: ONE kernel (P and dS computed once per pair;
 * dS goes through LDS to the dQ product); 2 (default): the same, and with 193..208 tokens the variant that fetches
 * the next pair's operands under the current pair's products; 0: the dQ kernel and the dK/dV kernel, as for every
 * other shape.  Process-wide; benchmarking. */
int vitmi_debug_set_attn_bwd_fused(int on);
/* The fp32 attention kernels.  1 (default): products of 16 x 16 tiles on the fp32 matrix instruction (heads of 8 /
 * 16 / 32 / 64, up to 288 tokens while two [tokens][head dim] images fit in LDS, 16-byte-aligned tensors), one launch
 * forward and one backward, no workspace use; 0: the row-by-row kernels every other fp32 shape takes (the backward
 * pass of those keeps P and dS in the workspace).  Same IEEE fp32 arithmetic either way.  Process-wide. */
int vitmi_debug_set_attn_f32_mfma(int on);
/* same for the weight-gradient (tn) GEMM: 0 = 128x128, 1 = 256x256, -1 = automatic */
int vitmi_debug_set_tn_config(int cfg);

/* Leave k compute units (a multiple of 8 in [0, 128]; 0 = none, the default) free of the one-workgroup-per-CU GEMM
 * kernels: the persistent nt kernel launches that many fewer blocks and the weight-gradient kernel sizes its token
 * slabs for the smaller chip.  For data-parallel training, where a collective (RCCL) has to find compute units
 * while the GEMMs of the backward pass hold the chip (DistributedDataParallel's overlap, trainer.py:184).
 * Process-wide. */
int vitmi_set_reserved_cus(int k);

/* dst[c][r] (bf16) = src[r][c] (src_dt): the transposed bf16 compute copy of a weight, which
 * turns the input-gradient product dX = dY W into the k-contiguous form of vitmi_gemm.  No
 * reference counterpart (aten reads W through strides). */
int vitmi_transpose_cast(const void* src, int src_dt, void* dst, int rows, int cols, void* stream);

/* out[c] = sum_r x[r*ld + c] in fp32 (bias gradients).  workspace: vitmi_colsum_workspace_bytes.
 * Replaces the bias-gradient reductions autograd runs for nn.Linear / nn.Conv2d. */
long long vitmi_colsum_workspace_bytes(long long rows, int cols);
int vitmi_colsum(const void* x, int dt, long long rows, int cols, long long ld, float* out, void* workspace,
                 void* stream);

/* The patch gather of Conv2d(k = s = p) + flatten(2).transpose(1, 2) (mymodel/ViT.py:33,43) as a
 * matrix: out[(b, gy, gx)][(c, i, j)] = x[b, c, gy*p + i, gx*p + j], row-major [B*(H/p)*(W/p)][C*p*p]
 * in out_dt.  For patch sizes the in-place gather of vitmi_patch_embed_* cannot read in 16-byte
 * pieces (p % 8 != 0) the host side gathers once and runs the ordinary GEMMs on the result. */
int vitmi_patchify(const void* x, int x_dt, void* out, int out_dt, int B, int C, int H, int W, int p, void* stream);

/* The image-side half of JigsawPuzzleMaskedRegion.__call__ (masking_generator.py:144-199: Unfold ->
 * overwrite the masked columns -> Fold) as one patch-level gather that never builds the unfolded
 * tensor.  x, out: [B,C,H,W] in dt (out must not alias x); src: int32 [L] on the device, L =
 * (H/p)*(W/p); patch j of out[b] is a copy of
 *   patch v   of x[b]                  when v = src[j] lies in [0, L)      ("mjp", untouched patches),
 *   patch v-L of bank[bank_idx[b]]     when v lies in [L, 2L)              ("pbf"; bank: [bank_count,C,H,W]
 *                                       images in dt, bank_idx: int32 [B] on the device),
 *   the patch fill[b] ([B,C,p,p], dt)  when v == -2                        ("repeat"),
 *   zeros                              otherwise                           ("zero").
 * bank and fill may be null when no src entry selects them. */
int vitmi_jigsaw_gather(const void* x, void* out, int dt, int B, int C, int H, int W, int p, const int* src,
                        const void* bank, int bank_count, const int* bank_idx, const void* fill, void* stream);

/* y[i] = min(max(x[i] * scale + addend[i], lo), hi): the whole-image branches of JigsawPuzzleMaskedRegion.__call__,
 * "Scale" (masking_generator.py:210-213: x * scale, addend NULL, lo/hi infinite) and "noise" (:215-218: the host's
 * N(0, 0.1) draw added, clipped to [0, 255]).  x, y in dt; addend fp32 [n] or NULL. */
int vitmi_scale_add_clip(const void* x, const float* addend, void* y, int dt, long long n, float scale, float lo,
                         float hi, void* stream);

/* out[b,c,i,j] = (sum_t weight[t] * x[b, c, gy(t)*p+i, gx(t)*p+j]) / L over the L patches t of each image:
 * to_patches[:, :, seq_ord].mean(2) of the "repeat" branch (masking_generator.py:195-196) with weight[t] =
 * the number of times patch t occurs in seq_ord.  weight: fp32 [L] on the device; out: [B,C,p,p] in dt. */
int vitmi_patch_weighted_mean(const void* x, int dt, int B, int C, int H, int W, int p, const float* weight, void* out,
                              void* stream);

/* Attention rollout (vit_rollout.py:18-61; the variant at trainer.py:17-44) over the L exported probability
 * tensors of one forward pass.  probs: HOST array of L device pointers, each [B,H,N,N] in dt, in layer order.
 * Per layer and sample: heads fused (head_fusion 0 mean, 1 max, 2 min), the discard_k smallest entries of the
 * flattened [N*N] matrix zeroed (topk(..., largest=False) + scatter_; the caller passes
 * discard_k = int(N*N*discard_ratio); keep_first != 0 leaves flat entry 0 out of the candidates as
 * trainer.py:32-33 does; among equal values the lowest indices go first, where torch leaves the choice
 * open), then a = (A + I)/2 with every row divided by its sum.  The result is row 0, columns 1..N-1, of
 * a_{L-1} ... a_1 a_0, computed as a chain of vector-matrix products; out: fp32 [B, N-1], final_mode 0 = as is
 * (trainer.py:44), 1 = divided by its maximum and passed through softmax (vit_rollout.py:51-55).
 * workspace: vitmi_rollout_workspace_bytes(L, B, N) bytes (the L normalised fp32 matrices). */
long long vitmi_rollout_workspace_bytes(int L, int B, int N);
int vitmi_rollout(const void* const* probs, int dt, int L, int B, int H, int N, int head_fusion, int discard_k,
                  int keep_first, int final_mode, float* out, void* workspace, void* stream);

/* Gradient-weighted rollout (vit_grad_rollout.py:9-36) over L probability tensors and the L gradients that reached
 * them (grads: HOST array of L device pointers, [B,H,N,N] dt each, layer order).  Per layer and sample: mean over the
 * heads of probs * grads with negative means set to zero, the discard_k smallest entries zeroed, a = (A + I)/2 with
 * entry (i, j) divided by the sum of ROW j (what `a / a.sum(dim=-1)` computes on the reference's batch of one), then
 * row 0 of a_{L-1} ... a_0, columns 1..N-1, divided by its maximum.  The reference is defined for one image; every
 * sample of a batch is treated here as that one image.  out: fp32 [B, N-1]; workspace as for vitmi_rollout. */
int vitmi_grad_rollout(const void* const* probs, const void* const* grads, int dt, int L, int B, int H, int N,
                       int discard_k, float* out, void* workspace, void* stream);

/* out[b][j-1] = heads-fused probs[b, :, 0, j] for j = 1..N-1: last_attn of vit_rollout.py:64-73. fp32 [B, N-1]. */
int vitmi_rollout_last_attn(const void* probs, int dt, int B, int H, int N, int head_fusion, float* out, void* stream);

/* Patch embedding: Conv2d(C, D, kernel=stride=p) + bias, flatten(2).transpose(1,2)
 * (mymodel/ViT.py:33,43) as an im2col-free GEMM: the A operand is gathered straight from the
 * NCHW image.  x is [B,C,H,W] in x_dt; w is [D, C*p*p] in dt; out token j of image b is
 * written at out + b*out_batch_stride + j*D (dt), so it can land inside a [B,N,D] buffer.  */
int vitmi_patch_embed_fwd(const void* x, int x_dt, const void* w, const float* bias, void* out, int dt, int B, int C,
                          int H, int W, int p, int D, long long out_batch_stride, void* stream);
/* dW[d, k] = sum_{b,j} dy[b,j,d] * patch(b,j)[k]  (fp32 [D, C*p*p]).  Conv2d weight gradient.
 * dy is dense [B, tokens, D]; workspace from the query below (may be 0 bytes -> NULL). */
long long vitmi_patch_embed_bwd_w_workspace_bytes(int dt, int B, int C, int H, int W, int p, int D);
int vitmi_patch_embed_bwd_w(const void* x, int x_dt, const void* dy, long long dy_batch_stride, float* dw,
                            void* workspace, int dt, int B, int C, int H, int W, int p, int D, void* stream);

/* The same two products for square patches NARROWER than 8 pixels (p in {1, 2, 4}: the 4 x 4 patches of the 32 x 32
 * geometry, UMAP.py:12-17), bf16 image and bf16 compute, still gathered in place: the product runs over groups of
 * g = 8 / p horizontally adjacent patches (16 contiguous bytes of an image row) against a block form of the weight.
 * vitmi_patch_wide_group: g when the form applies to this geometry (W % 8 == 0, D % 8 == 0), else 0.
 * vitmi_patch_wide_weight: w [D, C*p*p] fp32 (+ bias [D]) -> w_wide [g*D, C*p*8] bf16 (+ bias_wide [g*D] fp32); redo when
 * the weight changes.  vitmi_patch_embed_fwd_wide / _bwd_w_wide: as the calls above (dw is the ordinary [D, C*p*p]
 * gradient; the workspace holds the block-form gradient and the token slabs). */
int vitmi_patch_wide_group(int dt, int x_dt, int B, int C, int H, int W, int p, int D);
int vitmi_patch_wide_weight(const float* w, const float* bias, void* w_wide, float* bias_wide, int D, int C, int p,
                            void* stream);
int vitmi_patch_embed_fwd_wide(const void* x, const void* w_wide, const float* bias_wide, void* out, int B, int C, int H,
                               int W, int p, int D, long long out_batch_stride, void* stream);
long long vitmi_patch_embed_bwd_w_wide_workspace_bytes(int B, int C, int H, int W, int p, int D);
int vitmi_patch_embed_bwd_w_wide(const void* x, const void* dy, float* dw, void* workspace, int B, int C, int H, int W,
                                 int p, int D, void* stream);

/* seq_ord[i] = i*(1 - m[i]) + m[i]*masked_row for i < n: the index arithmetic of ViT_mask.py:162-163
 * (masked_row = unk_embed_index) in one launch; masked_row = -1 marks "no positional row" for the variants
 * that zero the masked positions (ViT_ape.py:186-190).  unk_mask: n entries of int64 (mask_is_int64 != 0) or
 * int32 on the device, values 0/1; seq_ord: int64 [n]. */
int vitmi_seq_ord(const void* unk_mask, int mask_is_int64, long long* seq_ord, int n, long long masked_row, void* stream);

/* Token assembly + positional embedding (mymodel/ViT.py:155-159; ViT_mask.py:156-172):
 *   out[b,0,:]   = cls[:]          + pe[row(0),:]
 *   out[b,1+j,:] = patch[b,j,:]    + pe[row(1+j),:]
 * row(i) = seq_ord[i] when seq_ord != NULL (device int64 [N]; ViT_mask.py:162-164), else i;
 * a negative seq_ord[i] adds nothing to token i (the zero row of ViT_mask.py:166-170);
 * pe == NULL skips the add (model.PE False).  patch is [B,N-1,D] dt, out is [B,N,D] dt.
 * pe_rows bounds-checks nothing on device: the caller guarantees 0 <= seq_ord[i] < pe_rows. */
int vitmi_tokens_assemble_fwd(const void* patch, const float* cls, const float* pe, const long long* seq_ord,
                              void* out, int dt, int B, int N, int D, void* stream);
/* Backward of the above for the parameters: dcls[D] = sum_b dx[b,0,:];
 * dpe[r,:] = sum_b sum_{i: row(i)=r} dx[b,i,:] for r in [0,pe_rows) (rows nobody reads get 0).
 * Deterministic (fixed summation order).  dpe may be NULL. */
long long vitmi_tokens_assemble_bwd_workspace_bytes(int N, int D);
int vitmi_tokens_assemble_bwd(const void* dx, const long long* seq_ord, float* dcls, float* dpe, int pe_rows,
                              void* workspace, int dt, int B, int N, int D, void* stream);

/* LayerNorm over the last dim (nn.LayerNorm, mymodel/ViT.py:55,147; eps as given).
 * x rows are ldx apart (lets the head read x[:,0] in place); y is dense [rows,D]. */
int vitmi_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int dt, long long rows, int D, float eps, void* stream);
/* dx = LN'(dy) [+ dres]; dgamma/dbeta fp32 [D].  dres (dt, dense [rows,D]) may be NULL.
 * dres_colsum (fp32 [D], may be NULL) receives the column sums of dres: dres is the gradient of the
 * residual stream, whose column sum is the bias gradient of the projection that fed the residual,
 * so that reduction rides on this kernel's pass over dres. */
long long vitmi_layernorm_bwd_workspace_bytes(long long rows, int D);
int vitmi_layernorm_bwd(const void* dy, const void* x, long long ldx, const float* gamma, const float* mean,
                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                        float* dres_colsum, void* workspace, int dt, long long rows, int D, void* stream);

/* vitmi_layernorm_bwd without a workspace argument, for callers that cannot allocate (code under HIP graph
 * capture, tight serving / fine-tuning loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * The per-block partial sums live in a device array of the library (768 * 3 * VITMI_LN_BWD_STATIC_MAX_D floats =
 * 9 MiB per GPU, held until the process ends); the call allocates and frees nothing.  D above
 * VITMI_LN_BWD_STATIC_MAX_D returns VITMI_ERR_UNSUPPORTED.  Same arguments, kernels and results (bit for bit) as
 * vitmi_layernorm_bwd.  All calls on one GPU share the array: calls that overlap on one GPU (two streams) race;
 * nothing in it has to be zero between calls.  The device current at the call must be the stream's device. */
#define VITMI_LN_BWD_STATIC_MAX_D 1024
int vitmi_layernorm_bwd_static(const void* dy, const void* x, long long ldx, const float* gamma, const float* mean,
                               const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                               float* dres_colsum, int dt, long long rows, int D, void* stream);

/* Multi-head self-attention core (mymodel/ViT.py:96-103): qkv is [B,N,3*H*DH] laid out as
 * the to_qkv output (q | k | v, each head-major), out is [B,N,H*DH] ('b h n d -> b n (h d)').
 * scores = (q k^T) * DH^-0.5 (scale after the matmul), softmax over keys, out = probs v.
 * lse [B,H,N] fp32 receives log-sum-exp of the scaled scores (kept for backward).
 * probs ([B,H,N,N] dt) may be NULL; when given it receives the post-softmax probabilities the
 * reference hands to forward hooks on Attention.attend (ViT.py:100) / attn_drop.
 * layout: 0, or VITMI_QKV_HEAD_MAJOR when qkv is stored as head-major planes [3][H][B*N][DH] (what
 * vitmi_attn_block_fwd writes for bf16 heads of 64: the [N][DH] matrix of one (q|k|v, image, head) is then contiguous);
 * vitmi_attn_bwd also takes VITMI_DQKV_HEAD_MAJOR for the layout dqkv is WRITTEN in.  Planes need bf16 and DH = 64. */
#define VITMI_QKV_HEAD_MAJOR 1
#define VITMI_DQKV_HEAD_MAJOR 2
int vitmi_attn_fwd(const void* qkv, void* out, float* lse, void* probs, int dt, int B, int N, int H, int DH, int layout,
                   void* stream);
/* dqkv from dout, recomputing the probabilities from qkv and lse.  dprobs ([B,H,N,N] dt, a gradient that reaches
 * the exported probabilities from outside: a loss on the attention maps, the hooks of vit_grad_rollout.py:38-56) may
 * be NULL; when given, its softmax backward (q and k slots) is added, the workspace must have
 * vitmi_attn_bwd_dprobs_workspace_bytes, and dout (with out) may be NULL, in which case dqkv receives the dprobs share
 * alone (v slots zero).  The dprobs share runs four small launches per (image, head): it serves attention-map
 * analysis on a few images, not the training step.  The dout path always needs the workspace of
 * vitmi_attn_bwd_workspace_bytes (bf16: rowsum(dout * out) per query, [B,H,N] fp32, which the dQ kernel leaves for
 * the dK/dV kernel; fp32: the probabilities and their gradient). */
long long vitmi_attn_bwd_workspace_bytes(int dt, int B, int N, int H, int DH);
long long vitmi_attn_bwd_dprobs_workspace_bytes(int dt, int B, int N, int H, int DH);
int vitmi_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const void* dprobs,
                   void* dqkv, void* workspace, int dt, int B, int N, int H, int DH, int layout, void* stream);

/* Whole encoder sub-blocks, each ONE call that enqueues the launches of the sub-block in order on the stream:
 *   ff block    y = x + W2 gelu(W1 LN(x) + b1) + b2       PreNorm(FeedForward) + residual, mymodel/ViT.py:58,65-73,120
 *   attn block  y = x + Wout attn(Wqkv LN(x) [+ bqkv]) [+ bout]   PreNorm(Attention) + residual, ViT.py:58,93-104,119
 * and their backward (what autograd derives for those lines).  No arithmetic of their own: every step is one of
 * the entry points above; they exist so that the host pays one foreign call per sub-block instead of one per
 * kernel.  All matrices are dense row-major in dt (weights as nn.Linear stores them, [out, in]); gamma, beta,
 * biases, mean, rstd, lse and every parameter gradient are fp32.  Saved for backward: xn (normalised input),
 * mean, rstd, and u / h (derivative of GELU at the MLP pre-activation, written by the forward epilogue
 * VITMI_EPI_GELU_D and multiplied in by VITMI_EPI_MULAUX in the backward pass, and the activation) or qkv / o / lse.  w*_t: optional transposed copies
 * [in, out] of the weights for the input-gradient products (NULL: read through strides).  b_qkv, b_out, db_qkv,
 * db_out and probs may be NULL.  workspace: the matching *_workspace_bytes (temporaries + scratch of the steps). */
int vitmi_ff_block_fwd(int dt, int rows, int D, int M, const void* x, const float* gamma, const float* beta, float eps,
                       const void* w1, const float* b1, const void* w2, const float* b2, void* xn, float* mean,
                       float* rstd, void* u, void* h, void* y, void* stream);
long long vitmi_ff_block_bwd_workspace_bytes(int dt, int rows, int D, int M);
int vitmi_ff_block_bwd(int dt, int rows, int D, int M, const void* dy, const void* x, const void* xn, const void* u,
                       const void* h, const float* gamma, const float* mean, const float* rstd, const void* w1,
                       const void* w1_t, const void* w2, const void* w2_t, void* dx, float* dgamma, float* dbeta,
                       float* dw1, float* db1, float* dw2, float* db2, void* workspace, void* stream);
/* At the small geometry (bf16, D <= 256 and a multiple of 32, hidden width a multiple of 32, 16-byte aligned
 * operands) the block calls run as one-launch kernels that keep the sub-block's activations in LDS; same outputs to the
 * bf16 bar.  vitmi_debug_set_small_fused(mask): bit 0 = the MLP sub-block's forward, bit 1 = the attention
 * sub-block's forward (up to 96 tokens, heads of 8 or 16, no probability export), bit 2 = the MLP sub-block's backward,
 * bit 3 = that backward kernel in calls that leave the weight gradients to the caller (dw1 == dw2 == NULL), bit 4 = the
 * attention sub-block's backward in such calls (dw_qkv == dw_out == NULL, no qkv bias, both transposed weight copies given)
 * (default 27: the two forward kernels and bits 3 and 4; a negative mask restores the default);
 * process-wide, for benchmarking and for the tests that compare the two forms. */
int vitmi_debug_set_small_fused(int mask);
/* The two weight-gradient products of a sub-block's backward call may run on a second stream of the library's own,
 * beside the input-gradient chain (forked from and joined back into the caller's stream inside the call; their scratch
 * is a region of its own in the workspace).  mode -1: automatic (default) = for rows * D up to 8 M elements while the
 * caller's stream is being captured into a graph (issued eagerly the event calls cost the host more than the overlap
 * returns); 0: never; 1: always.  Process-wide, for benchmarking. */
int vitmi_debug_set_wgrad_overlap(int mode);
int vitmi_attn_block_fwd(int dt, int B, int N, int D, int H, int DH, const void* x, const float* gamma, const float* beta,
                         float eps, const void* w_qkv, const float* b_qkv, const void* w_out, const float* b_out,
                         void* xn, float* mean, float* rstd, void* qkv, void* o, float* lse, void* probs, void* y,
                         int qkv_layout, void* stream);
/* The layout vitmi_attn_block_fwd should be asked to leave qkv in for this shape (VITMI_QKV_HEAD_MAJOR for bf16 heads
 * of 64, else 0); the same value goes to vitmi_attn_block_bwd, or to vitmi_attn_bwd when the backward pass is driven
 * step by step.  vitmi_debug_set_qkv_planes(mask): bit 0 = recommend planes for qkv, bit 1 = planes for the dqkv that
 * vitmi_attn_block_bwd keeps in its workspace (default 3); process-wide, for benchmarking. */
int vitmi_attn_block_qkv_layout(int dt, int B, int N, int D, int H, int DH);
int vitmi_debug_set_qkv_planes(int mask);
long long vitmi_attn_block_bwd_workspace_bytes(int dt, int B, int N, int D, int H, int DH);
int vitmi_attn_block_bwd(int dt, int B, int N, int D, int H, int DH, const void* dy, const void* x, const void* xn,
                         const void* qkv, const void* o, const float* lse, const float* gamma, const float* mean,
                         const float* rstd, const void* w_qkv, const void* w_qkv_t, const void* w_out,
                         const void* w_out_t, void* dx, float* dgamma, float* dbeta, float* dw_qkv, float* db_qkv,
                         float* dw_out, float* db_out, void* workspace, int qkv_layout, void* stream);

/* The same two sub-blocks for inference (no autograd graph: model.eval() under torch.no_grad(), the mode of
 * mia_attn.py / vit_rollout.py / every evaluation loop of trainer.py), bf16 only, nothing kept for a backward pass.
 * nn.LayerNorm + nn.Linear (mymodel/ViT.py:55,58 in front of :86 and :66) run as ONE product: the normalised rows are
 * never written.  vitmi_row_stats leaves mean and 1/sqrt(var + eps) per row; vitmi_gemm_ln multiplies the RAW rows by
 * the weight with gamma folded in and finishes in its epilogue,
 *     y[m][n] = rstd[m] * (sum_k x[m][k] Wg[n][k] - mean[m] * s[n]) + c[n],
 *     Wg = bf16(W * gamma) [N,K],  s[n] = sum_k Wg[n][k] (fp32, of the rounded values),  c = W beta + b (fp32),
 * VITMI_EPI_LN_GELU applying gelu_erf to that (the MLP's first layer; no derivative is stored).  The folded
 * operands are made once per weight version by the caller.  probs and b_out may be NULL. */
int vitmi_row_stats(const void* x, long long ldx, float* mean, float* rstd, int dt, long long rows, int D, float eps,
                    void* stream);
/* The folded operands of vitmi_gemm_ln from the fp32 parameters (nn.LayerNorm weight / bias of mymodel/ViT.py:55 and the
 * nn.Linear behind it, :66 / :86): w [N,K], gamma / beta [K], bias [N] or NULL -> w_folded bf16 [N,K], w_rowsum [N],
 * bias_folded [N].  Weight preparation: one launch per weight version. */
int vitmi_fold_ln(const float* w, const float* gamma, const float* beta, const float* bias, void* w_folded,
                  float* w_rowsum, float* bias_folded, int N, int K, void* stream);
int vitmi_gemm_ln(int M, int N, int K, const void* x, long long ldx, const void* w_folded, const float* bias_folded,
                  const float* w_rowsum, const float* mean, const float* rstd, void* C, long long ldc, int epilogue,
                  void* stream);
long long vitmi_attn_block_infer_workspace_bytes(int B, int N, int D, int H, int DH);
int vitmi_attn_block_infer(int B, int N, int D, int H, int DH, const void* x, float eps, const void* wq_folded,
                           const float* bq_folded, const float* wq_rowsum, const void* w_out, const float* b_out,
                           void* probs, void* y, void* workspace, void* stream);
long long vitmi_ff_block_infer_workspace_bytes(int rows, int D, int M);
int vitmi_ff_block_infer(int rows, int D, int M, const void* x, float eps, const void* w1_folded,
                         const float* b1_folded, const float* w1_rowsum, const void* w2, const float* b2, void* y,
                         void* workspace, void* stream);

/* Per-sample gradient norms for DP-SGD (defence_method/DPSGD.py:114-136, which clips each sample's gradient and
 * gets per-sample gradients by running the batch one image at a time).  Rows of the [tokens, features]
 * matrices are cut into segments of `seg` consecutive rows (one image each); every routine ADDS its parameter's
 * share of ||g_i||^2 to sqnorm[i] (fp32, one entry per segment, zeroed by the caller) without materialising g_i:
 *   vitmi_wgrad_sqnorm          || U_i^T V_i ||_F^2, the weight of y = x W^T with U = dy [T,P], V = x [T,Q]
 *                               (bf16 aligned shapes: the weight-gradient MFMA kernel with its token slabs cut at
 *                               image boundaries, each block reducing its tile to a sum of squares; otherwise one
 *                               vitmi_gemm_wgrad per image); workspace: vitmi_wgrad_sqnorm_workspace_bytes
 *   vitmi_seg_colsum_sqnorm     || sum_t x[t,:] ||^2            (biases: x = dy)
 *   vitmi_seg_layernorm_sqnorm  || sum_t dy*xhat ||^2 + || sum_t dy ||^2   (LayerNorm weight and bias; dy dense [rows,D])
 *   vitmi_seg_sqnorm            sum_t || x[t,:] ||^2            (positional table, class token: x = dx)
 * Reductions run in a fixed order; each sqnorm entry has a single writer per launch.
 * The two column-sum routines can also leave the per-image sums themselves (`sums` [segments, cols] fp32, NULL = not
 * wanted).  With the clipping factors c_i = min(1, C / ||g_i||) known, the clipped gradient sum_i c_i g_i then needs no
 * second pass over the network (the upstream gradient of image i in a backward pass of sum_i c_i loss_i is c_i times
 * its value in the norm pass):
 *   vitmi_seg_combine           out[c] = sum_i coef[i] x[i][c]  (vector parameters: x = the per-image sums)
 *   vitmi_seg_scale_rows        out[t,:] = x[t,:] * coef[t / seg]: one operand of a weight-gradient product, so that
 *                               vitmi_gemm_wgrad(dy, coef-scaled x) = sum_i c_i dy_i^T x_i */
/* vitmi_wgrad_sqnorm on bf16 images of up to 256 tokens uses || U^T V ||_F^2 = sum_{t,t'} (U U^T)[t,t'] (V V^T)[t,t']:
 * two token Gram matrices per image instead of the gradient (3x fewer flops for the MLP weights at 197 tokens);
 * vitmi_debug_set_ghost_norms(0) switches back to the tiles of the gradient itself (process-wide; for A/B timing). */
int vitmi_debug_set_ghost_norms(int on);
long long vitmi_wgrad_sqnorm_workspace_bytes(int dt, int P, int Q, int T, int seg);
int vitmi_wgrad_sqnorm(int dt, int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                       float* sqnorm, void* workspace, void* stream);
int vitmi_seg_colsum_sqnorm(const void* x, int dt, long long rows, int cols, long long ld, int seg, float* sqnorm,
                            float* sums, void* stream);
int vitmi_seg_layernorm_sqnorm(const void* dy, const void* x, long long ldx, const float* mean, const float* rstd, int dt,
                               long long rows, int D, int seg, float* sqnorm, float* sums_beta, float* sums_gamma,
                               void* stream);
int vitmi_seg_combine(const void* x, int dt, int n, long long cols, long long ld, const float* coef, float* out,
                      void* stream);
int vitmi_seg_scale_rows(const void* x, int dt, long long rows, int cols, long long ld, int seg, const float* coef,
                         void* out, long long ldo, void* stream);
int vitmi_seg_sqnorm(const void* x, int dt, long long rows, int cols, long long ld, int seg, float* sqnorm, void* stream);

/* mean over tokens: out[b,:] = mean_i x[b,i,:]  (pool == 'mean', mymodel/ViT.py:164) and its
 * backward dx[b,i,:] = dout[b,:] / N. */
int vitmi_meanpool_fwd(const void* x, void* out, int dt, int B, int N, int D, void* stream);
int vitmi_meanpool_bwd(const void* dout, void* dx, int dt, int B, int N, int D, void* stream);

/* y = a + b elementwise (dt); gradient accumulation on the residual stream. */
int vitmi_add(const void* a, const void* b, void* y, int dt, long long n, void* stream);

/* nn.Dropout in training mode (mymodel/ViT.py:68,70,90: after GELU, after the second MLP layer, after to_out):
 * y[i] = x[i] * keep(i) / (1 - p) with keep(i) drawn from a counter-based generator keyed by (seed, i), so the
 * backward pass is the same call on the gradient with the same seed and no mask is stored.  In place allowed. */
int vitmi_dropout(const void* x, void* y, int dt, long long n, float p, unsigned long long seed, void* stream);

/* Fused Adam step over one flat fp32 parameter tensor (torch.optim.Adam semantics with
 * weight_decay added to the gradient, trainer.py:204-205), optionally refreshing a bf16 copy
 * of the parameter in the same pass.  step is the 1-based step count. */
int vitmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                    long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale, void* stream);

/* The same update for many tensors in ONE launch (torch.optim.Adam walks its parameter list,
 * trainer.py:204-205; a ViT-B/16 has ~150 parameter tensors, most of them tiny).
 * tensors (device memory) holds n_tensors records; chunks (device memory) holds n_chunks pairs
 * {tensor index, first element}: block c updates elements [first, first + chunk_elems) of its
 * tensor, clipped to the tensor's n.  chunk_elems must be a positive multiple of 4. */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    void* param_bf16; /* may be NULL */
    long long n;
} vitmi_adam_tensor;
typedef struct {
    long long tensor;
    long long first;
} vitmi_adam_chunk;
int vitmi_adam_step_multi(const vitmi_adam_tensor* tensors, int n_tensors, const vitmi_adam_chunk* chunks,
                          int n_chunks, int chunk_elems, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float grad_scale, void* stream);

/* vitmi_adam_step_multi for weight matrices that keep a transposed bf16 copy (the B operand of the input-gradient
 * products, x.grad = dy @ W of every nn.Linear, mymodel/ViT.py:65-68,86,88): one block per 64 x 64 tile updates the
 * tile, writes the row-major bf16 copy (param_bf16, may be NULL) and the transposed one (param_bf16_t[c][r], required)
 * in the same pass.  mats and tile_start live on the device; tile_start[j] = number of 64 x 64 tiles of the matrices
 * before j (ceil(rows/64) * ceil(cols/64) each), total_tiles their sum = the grid.  Contract: cols % 4 == 0, the four
 * fp32 pointers 16-byte aligned, param_bf16 and param_bf16_t 8-byte aligned. */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    void* param_bf16;   /* [rows][cols], may be NULL */
    void* param_bf16_t; /* [cols][rows] */
    long long rows;
    long long cols;
} vitmi_adam_matrix;
int vitmi_adam_step_matrices(const vitmi_adam_matrix* mats, const long long* tile_start, int n_mats,
                             long long total_tiles, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int step, float grad_scale, void* stream);

/* vitmi_transpose_cast (fp32 source) for many matrices in ONE launch: job j writes dst[c][r] = (bf16) src[r][c]
 * for its rows x cols matrix.  jobs and tile_start live on the device; tile_start[j] = number of 64 x 64 tiles
 * of the jobs before j (ceil(rows/64) * ceil(cols/64) each), total_tiles their sum = the grid.  Used after the
 * optimizer step to refresh every transposed weight copy at once (49 launches per step for ViT-B/16 otherwise). */
typedef struct {
    const void* src;
    void* dst;
    long long rows;
    long long cols;
} vitmi_transpose_job;
int vitmi_transpose_cast_multi(const vitmi_transpose_job* jobs, const long long* tile_start, int n_jobs,
                               long long total_tiles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITMI_H */
