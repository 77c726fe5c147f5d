// Whole encoder sub-blocks as single C-ABI calls: each function enqueues, in order and on one stream, the
// launches that make up PreNorm(Attention)+residual or PreNorm(FeedForward)+residual (mymodel/ViT.py:52-58,
// 62-73, 77-104, 119-120) or their backward.  They add no arithmetic of their own -- every step is one of the
// entry points declared in vitmi.h -- and exist so that a host written in a slow language pays for one foreign
// call per sub-block instead of one per kernel (at the 32x32-pixel geometry the step time is host time).
// Temporaries and the scratch of the steps live in one caller-provided workspace: the steps run one after
// another on the stream, so they share its scratch part.
#include "common.h"

namespace {

inline long long align256(long long n) { return (n + 255) & ~255ll; }
inline long long esize(int dt) { return dt == VITMI_F32 ? 4 : 2; }
inline long long max2(long long a, long long b) { return a > b ? a : b; }

#define STEP(call)                      \
    do {                                \
        const int rc__ = (call);        \
        if (rc__ != VITMI_OK) return rc__; \
    } while (0)

// dX = dY W for y = x W^T: through the transposed copy W^T [K,N] when the caller has one, else through strides
int input_grad(int dt, int T, int K, int N, const void* dy, const void* w, const void* w_t, void* dx, int epilogue,
               void* aux, float* colsum, void* scratch, void* stream) {
    const void* b = w_t ? w_t : w;
    const long long sbn = w_t ? N : 1, sbk = w_t ? 1 : K;
    if (colsum)
        return vitmi_gemm_colsum(dt, T, K, N, dy, N, 1, b, sbn, sbk, dx, K, dt, epilogue, nullptr, nullptr, 0, aux,
                                 aux ? K : 0, colsum, scratch, stream);
    return vitmi_gemm(dt, T, K, N, dy, N, 1, b, sbn, sbk, dx, K, dt, epilogue, nullptr, nullptr, 0, aux, aux ? K : 0,
                      stream);
}

// ---- weight gradients beside the input-gradient chain ---------------------------------------------------------------
// At small shapes no launch of a sub-block's backward fills the chip, and the two weight-gradient products are off
// the critical path (nothing downstream of the call reads dW before the optimizer): they are enqueued on a second
// stream of the library's own, each forked from the caller's stream by an event as soon as its operands exist -- the
// first at the start of the call (dY^T H, dY^T O read the call's inputs only), the second behind the launch that
// produces its dY operand -- and joined back before the call returns, so that they run beside the input-gradient
// chain and the LayerNorm backward.  The caller's stream sees
// one call as before (also under stream capture: the fork and join become graph edges).
// (one per host thread and device, created on first use and kept for the life of the process)
struct Side {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, fork2 = nullptr, join = nullptr;
    bool failed = false;
};
constexpr int MAX_DEVICES = 16;
thread_local Side tl_side[MAX_DEVICES];
int g_overlap = -1;  // -1 automatic, 0 never, 1 always

Side* side_stream() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    Side& sd = tl_side[dev];
    if (sd.failed) return nullptr;
    if (!sd.s) {
        if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.fork2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            sd.failed = true;  // stay on the caller's stream from now on (this thread, this device)
            sd.s = nullptr;
            return nullptr;
        }
    }
    return &sd;
}

// Automatic mode: rows * width up to 8 M elements (every product of the sub-block is then a launch of ~10 us that
// leaves most of the chip idle) AND the caller's stream is being captured into a graph.  Issued eagerly, each fork /
// join pair costs the host about 20 us of event calls -- more than the overlap returns -- while in a captured graph
// they are edges that cost nothing at replay (measured at the 32-pixel geometry, batch 256: eager 3.9 -> 5.3 ms per
// step with the forks, graph 3.9 -> 3.3 ms).
Side* overlap_side(long long rows, int D, void* stream) {
    if (g_overlap == 0) return nullptr;
    if (g_overlap > 0) return side_stream();
    if (rows * (long long)D > (8ll << 20)) return nullptr;
    // created at the first small-shape call, which is normally an eager one: no stream creation inside a capture
    Side* sd = side_stream();
    if (!sd) return nullptr;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(as_stream(stream), &st) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return st == hipStreamCaptureStatusActive ? sd : nullptr;
}

// the second stream waits for everything enqueued on the caller's stream so far
int side_fork(Side* sd, void* stream, bool second = false) {
    hipEvent_t ev = second ? sd->fork2 : sd->fork;
    hipError_t e = hipEventRecord(ev, as_stream(stream));
    if (e == hipSuccess) e = hipStreamWaitEvent(sd->s, ev, 0);
    return e == hipSuccess ? VITMI_OK : vitmi_hip_fail(e, "side stream fork");
}
int side_join(Side* sd, void* stream) {
    hipError_t e = hipEventRecord(sd->join, sd->s);
    if (e == hipSuccess) e = hipStreamWaitEvent(as_stream(stream), sd->join, 0);
    return e == hipSuccess ? VITMI_OK : vitmi_hip_fail(e, "side stream join");
}

}  // namespace

extern "C" int vitmi_debug_set_wgrad_overlap(int mode) {
    g_overlap = mode < 0 ? -1 : (mode ? 1 : 0);
    return VITMI_OK;
}

// ------------------------------------------------------------------------------------------------ MLP block
// the sub-block as ONE launch at the small geometry (small.hip); NOT_TAKEN = operands not 16-byte aligned, nothing enqueued
enum { VITMI_SMALL_NOT_TAKEN = 1000 };
bool vitmi_ff_small_ok(int dt, long long rows, int D, int M);
bool vitmi_ff_small_bwd_ok(int dt, long long rows, int D, int M, bool caller_forms_dw);
int vitmi_ff_small_bwd_strips(long long rows, int D, int M);
long long vitmi_ff_small_bwd_partial_bytes(long long rows, int D, int M);
int vitmi_ff_small_bwd(long long rows, int D, int M, const void* dy, const void* x, const void* u, const float* gamma,
                       const float* mean, const float* rstd, const void* w2_t, const void* w1_t, void* du, void* dx,
                       float* partial, int erf_pair, void* stream);
int vitmi_ff_small_bwd_final(long long rows, int D, int M, const float* partial, float* db1, float* dgamma, float* dbeta,
                             float* db2, void* stream);
int vitmi_ff_small_fwd(long long rows, int D, int M, const void* x, const float* gamma, const float* beta, float eps,
                       const void* w1, const float* b1, const void* w2, const float* b2, void* xn, float* mean, float* rstd,
                       void* u, void* h, void* y, int erf_pair, void* stream);
// which epilogue pair the MLP block runs: 0 = GELU_D / MULAUX (u holds gelu'(pre-activation)), 1 = GELU / DGELU (u
// holds the pre-activation and the backward epilogue evaluates the derivative)
static int g_classic_gelu_pair = 0;
extern "C" int vitmi_set_gelu_mode(int erf_pair) {
    g_classic_gelu_pair = erf_pair ? 1 : 0;
    return VITMI_OK;
}
extern "C" int vitmi_get_gelu_mode() { return g_classic_gelu_pair; }
extern "C" int vitmi_debug_set_gelu_pair(int classic) { return vitmi_set_gelu_mode(classic); }   // older name
extern "C" int vitmi_ff_block_fwd(int dt, int rows, int D, int M, const void* x, const float* gamma, const float* beta,
                                  float eps, const void* w1, const float* b1, const void* w2, const float* b2, void* xn,
                                  float* mean, float* rstd, void* u, void* h, void* y, void* stream) {
    VITMI_CHECK_ARG(x && gamma && beta && w1 && b1 && w2 && b2 && xn && mean && rstd && u && h && y,
                    "vitmi_ff_block_fwd: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && M > 0, "vitmi_ff_block_fwd: bad shape rows=%d D=%d M=%d", rows, D, M);
    if (vitmi_ff_small_ok(dt, rows, D, M)) {
        const int rc = vitmi_ff_small_fwd(rows, D, M, x, gamma, beta, eps, w1, b1, w2, b2, xn, mean, rstd, u, h, y,
                                          g_classic_gelu_pair, stream);
        if (rc != VITMI_SMALL_NOT_TAKEN) return rc;
    }
    STEP(vitmi_layernorm_fwd(x, D, gamma, beta, xn, mean, rstd, dt, rows, D, eps, stream));
    STEP(vitmi_gemm(dt, rows, M, D, xn, D, 1, w1, D, 1, h, M, dt, g_classic_gelu_pair ? VITMI_EPI_GELU : VITMI_EPI_GELU_D, b1, nullptr, 0, u,
                    M, stream));
    STEP(vitmi_gemm(dt, rows, D, M, h, M, 1, w2, M, 1, y, D, dt, VITMI_EPI_NONE, b2, x, D, nullptr, 0, stream));
    return VITMI_OK;
}

// scratch of the launches on the caller's stream; the weight-gradient pair has its own region behind it (it may run
// beside them)
static long long ff_chain_scratch(long long rows, int D, int M) {
    return align256(max2(max2(vitmi_gemm_colsum_workspace_bytes(rows, M), vitmi_layernorm_bwd_workspace_bytes(rows, D)),
                         vitmi_ff_small_bwd_partial_bytes(rows, D, M)));
}

extern "C" long long vitmi_ff_block_bwd_workspace_bytes(int dt, int rows, int D, int M) {
    if (rows <= 0 || D <= 0 || M <= 0) return 0;
    return align256((long long)rows * M * esize(dt)) + align256((long long)rows * D * esize(dt)) +
           ff_chain_scratch(rows, D, M) + align256(vitmi_gemm_wgrad_pair_workspace_bytes(dt, rows, D, M, M, D));
}

extern "C" long long vitmi_ff_block_bwd_du_offset(int dt, int rows, int D, int M) {
    (void)dt; (void)rows; (void)D; (void)M;
    return 0;   // du [rows, M] opens the workspace
}

extern "C" long long vitmi_ff_block_bwd_sums_offset(int dt, int rows, int D, int M) {
    if (rows <= 0 || D <= 0 || M <= 0) return 0;
    return align256((long long)rows * M * esize(dt)) + align256((long long)rows * D * esize(dt));   // behind du and dxn
}
extern "C" int vitmi_ff_block_bwd_sums_strips(int dt, int rows, int D, int M) {
    if (rows <= 0 || D <= 0 || M <= 0 || !vitmi_ff_small_bwd_ok(dt, rows, D, M, true)) return 0;
    return vitmi_ff_small_bwd_strips(rows, D, M);
}

extern "C" int vitmi_ff_block_bwd(int dt, int rows, int D, int M, const void* dy, const void* x, const void* xn,
                                  const void* u, const void* h, const float* gamma, const float* mean, const float* rstd,
                                  const void* w1, const void* w1_t, const void* w2, const void* w2_t, void* dx,
                                  float* dgamma, float* dbeta, float* dw1, float* db1, float* dw2, float* db2,
                                  void* workspace, void* stream) {
    // dgamma == null: the vector gradients are the caller's as well, later (vitmi_small_sums_multi), from the partial
    // sums of the one-launch kernel at vitmi_ff_block_bwd_sums_offset in the workspace it keeps until then
    const bool skip_sums = !dgamma;
    VITMI_CHECK_ARG(dy && x && xn && u && h && gamma && mean && rstd && w1 && w2 && dx && workspace &&
                        ((dw1 != nullptr) == (dw2 != nullptr)) &&
                        (skip_sums ? (!dbeta && !db1 && !db2 && !dw1) : (dbeta && db1 && db2)),
                    "vitmi_ff_block_bwd: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && M > 0, "vitmi_ff_block_bwd: bad shape rows=%d D=%d M=%d", rows, D, M);
    VITMI_CHECK_ARG(!skip_sums || (w1_t && w2_t && vitmi_ff_small_bwd_ok(dt, rows, D, M, true)),
                    "vitmi_ff_block_bwd: dgamma == NULL (sums left to the caller) needs the one-launch kernel's shape and "
                    "both transposed weights (vitmi_ff_block_bwd_sums_strips says 0 otherwise)");
    char* ws = (char*)workspace;
    void* du = ws;
    void* dxn = ws + align256((long long)rows * M * esize(dt));
    void* scratch = (char*)dxn + align256((long long)rows * D * esize(dt));
    void* wg_scratch = (char*)scratch + ff_chain_scratch(rows, D, M);
    // dw1 == dw2 == null: the caller forms the weight gradients itself, later (vitmi_gemm_wgrad_multi), from dy, h, xn
    // and du, which it finds at vitmi_ff_block_bwd_du_offset in the workspace it keeps until then
    const bool skip_w = !dw1;
    Side* sd = skip_w ? nullptr : overlap_side(rows, D, stream);
    if (sd) {
        STEP(side_fork(sd, stream));
        STEP(vitmi_gemm_wgrad(dt, D, M, rows, dy, D, h, M, dw2, M, wg_scratch, sd->s));
    }
    if (w1_t && w2_t && vitmi_ff_small_bwd_ok(dt, rows, D, M, skip_w)) {
        // one launch for du, dxn and the LayerNorm backward (small.hip); the sums for db1, dgamma, dbeta, db2 are
        // finished by a small kernel that, like the weight gradients, need not sit on the caller's chain
        const int rc = vitmi_ff_small_bwd(rows, D, M, dy, x, u, gamma, mean, rstd, w2_t, w1_t, du, dx, (float*)scratch,
                                          g_classic_gelu_pair, stream);
        if (rc != VITMI_SMALL_NOT_TAKEN) {
            if (rc != VITMI_OK) return rc;
            if (skip_sums) {
                // nothing: the caller reduces the partial sums
            } else if (skip_w) {
                STEP(vitmi_ff_small_bwd_final(rows, D, M, (const float*)scratch, db1, dgamma, dbeta, db2, stream));
            } else if (sd) {
                STEP(side_fork(sd, stream, true));
                STEP(vitmi_gemm_wgrad(dt, M, D, rows, du, M, xn, D, dw1, D, wg_scratch, sd->s));
                STEP(vitmi_ff_small_bwd_final(rows, D, M, (const float*)scratch, db1, dgamma, dbeta, db2, sd->s));
                STEP(side_join(sd, stream));
            } else {
                STEP(vitmi_gemm_wgrad_pair(dt, rows, D, M, dy, D, h, M, dw2, M, D, du, M, xn, D, dw1, wg_scratch, stream));
                STEP(vitmi_ff_small_bwd_final(rows, D, M, (const float*)scratch, db1, dgamma, dbeta, db2, stream));
            }
            return VITMI_OK;
        }
    }
    if (skip_sums) {
        vitmi_set_error("vitmi_ff_block_bwd: dgamma == NULL, but an operand misses the 16-byte alignment of the one-launch kernel");
        return VITMI_ERR_BAD_ARG;
    }
    STEP(input_grad(dt, rows, M, D, dy, w2, w2_t, du, g_classic_gelu_pair ? VITMI_EPI_DGELU : VITMI_EPI_MULAUX,
                    const_cast<void*>(u), db1, scratch, stream));
    if (sd) {
        STEP(side_fork(sd, stream, true));
        STEP(vitmi_gemm_wgrad(dt, M, D, rows, du, M, xn, D, dw1, D, wg_scratch, sd->s));
    } else if (!skip_w) {
        // both weight gradients of the sub-block share the token axis: one launch where the shapes allow it
        STEP(vitmi_gemm_wgrad_pair(dt, rows, D, M, dy, D, h, M, dw2, M, D, du, M, xn, D, dw1, wg_scratch, stream));
    }
    STEP(input_grad(dt, rows, D, M, du, w1, w1_t, dxn, VITMI_EPI_NONE, nullptr, nullptr, nullptr, stream));
    STEP(vitmi_layernorm_bwd(dxn, x, D, gamma, mean, rstd, dy, dx, dgamma, dbeta, db2, scratch, dt, rows, D, stream));
    if (sd) STEP(side_join(sd, stream));
    return VITMI_OK;
}

// ------------------------------------------------------------------------------------------- attention block
bool vitmi_attn_small_ok(int dt, int B, int N, int D, int H, int DH);
int vitmi_attn_small_fwd(int B, int N, int D, int H, int DH, const void* x, const float* gamma, const float* beta, float eps,
                         const void* w_qkv, const float* b_qkv, const void* w_out, const float* b_out, void* xn, float* mean,
                         float* rstd, void* qkv, void* o, float* lse, void* y, void* stream);
int vitmi_small_fused_mask();
bool vitmi_attn_small_bwd_ok(int dt, int B, int N, int D, int H, int DH);
long long vitmi_attn_small_bwd_partial_bytes(int B, int D);
int vitmi_attn_small_bwd(int B, int N, int D, int H, int DH, const void* dy, const void* x, const void* qkv, const void* o,
                         const float* lse, const float* gamma, const float* mean, const float* rstd, const void* w_out_t,
                         const void* w_qkv_t, void* dqkv, void* dx, float* partial, void* stream);
int vitmi_small_sums_final(int strips, int D, const float* partial, float* dgamma, float* dbeta, float* db, void* stream);
bool vitmi_qkv_planes_ok(int dt, int B, int N, int D, int H, int DH, const void* x, const void* w_qkv, const void* qkv);
int vitmi_gemm_to_planes(int T, int N, int K, const void* xn, const void* w, const float* bias, void* planes, void* stream);
int vitmi_gemm_from_planes(int T, int D, int I3, const void* dqkv_planes, const void* w_t, void* dxn, void* stream);
int vitmi_gemm_wgrad_pair_planes(int T, int P0, int Q0, const void* U0, long long ldu0, const void* V0, long long ldv0,
                                 float* C0, int P1, int Q1, const void* U1_planes, const void* V1, long long ldv1, float* C1,
                                 void* workspace, void* stream);

// Head-major planes for qkv (bit 0) and, inside the backward call, for dqkv (bit 1): heads of 64 in bf16.  The
// (image, head) matrices the attention kernels read are then contiguous 2 * N * 64-byte blocks instead of 128-byte
// pieces 6 * H * 64 bytes apart, and so are the 64-deep k-blocks of the A operand of the input-gradient product.
static int g_qkv_planes = 3;
extern "C" int vitmi_debug_set_qkv_planes(int mask) {
    g_qkv_planes = mask & 3;
    return VITMI_OK;
}
extern "C" int vitmi_attn_block_qkv_layout(int dt, int B, int N, int D, int H, int DH) {
    (void)B;
    (void)N;
    (void)H;
    return ((g_qkv_planes & 1) && dt == VITMI_BF16 && DH == 64 && D % 8 == 0) ? VITMI_QKV_HEAD_MAJOR : 0;
}

extern "C" int vitmi_attn_block_fwd(int dt, int B, int N, int D, int H, int DH, const void* x, const float* gamma,
                                    const float* beta, float eps, const void* w_qkv, const float* b_qkv,
                                    const void* w_out, const float* b_out, void* xn, float* mean, float* rstd, void* qkv,
                                    void* o, float* lse, void* probs, void* y, int qkv_layout, void* stream) {
    VITMI_CHECK_ARG(x && gamma && beta && w_qkv && w_out && xn && mean && rstd && qkv && o && lse && y,
                    "vitmi_attn_block_fwd: null pointer");
    VITMI_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && DH > 0, "vitmi_attn_block_fwd: bad shape");
    VITMI_CHECK_ARG(qkv_layout == 0 || qkv_layout == VITMI_QKV_HEAD_MAJOR, "vitmi_attn_block_fwd: bad qkv layout %d", qkv_layout);
    const int rows = B * N, I = H * DH;
    if (!probs && !qkv_layout && (vitmi_small_fused_mask() & 2) && vitmi_attn_small_ok(dt, B, N, D, H, DH)) {
        const int rc = vitmi_attn_small_fwd(B, N, D, H, DH, x, gamma, beta, eps, w_qkv, b_qkv, w_out, b_out, xn, mean, rstd, qkv,
                                            o, lse, y, stream);
        if (rc != VITMI_SMALL_NOT_TAKEN) return rc;
    }
    STEP(vitmi_layernorm_fwd(x, D, gamma, beta, xn, mean, rstd, dt, rows, D, eps, stream));
    if (qkv_layout) {
        if (!vitmi_qkv_planes_ok(dt, B, N, D, H, DH, xn, w_qkv, qkv)) {
            vitmi_set_error("vitmi_attn_block_fwd: head-major qkv needs bf16, heads of 64, D %% 8 == 0 and 16-byte aligned operands");
            return VITMI_ERR_UNSUPPORTED;
        }
        STEP(vitmi_gemm_to_planes(rows, 3 * I, D, xn, w_qkv, b_qkv, qkv, stream));
    } else {
        STEP(vitmi_gemm(dt, rows, 3 * I, D, xn, D, 1, w_qkv, D, 1, qkv, 3 * I, dt, VITMI_EPI_NONE, b_qkv, nullptr, 0, nullptr,
                        0, stream));
    }
    STEP(vitmi_attn_fwd(qkv, o, lse, probs, dt, B, N, H, DH, qkv_layout, stream));
    STEP(vitmi_gemm(dt, rows, D, I, o, I, 1, w_out, I, 1, y, D, dt, VITMI_EPI_NONE, b_out, x, D, nullptr, 0, stream));
    return VITMI_OK;
}

static long long attn_chain_scratch(int dt, int B, int N, int D, int H, int DH) {
    const long long rows = (long long)B * N;
    long long scratch = vitmi_attn_bwd_workspace_bytes(dt, B, N, H, DH);
    scratch = max2(scratch, vitmi_colsum_workspace_bytes(rows, 3 * H * DH));
    scratch = max2(scratch, vitmi_layernorm_bwd_workspace_bytes(rows, D));
    scratch = max2(scratch, vitmi_attn_small_bwd_partial_bytes(B, D));
    return align256(scratch);
}

extern "C" long long vitmi_attn_block_bwd_workspace_bytes(int dt, int B, int N, int D, int H, int DH) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || DH <= 0) return 0;
    const int rows = B * N, I = H * DH;
    return align256((long long)rows * I * esize(dt)) + align256((long long)rows * 3 * I * esize(dt)) +
           align256((long long)rows * D * esize(dt)) + attn_chain_scratch(dt, B, N, D, H, DH) +
           align256(vitmi_gemm_wgrad_pair_workspace_bytes(dt, rows, D, I, 3 * I, D));
}

extern "C" long long vitmi_attn_block_bwd_dqkv_offset(int dt, int B, int N, int D, int H, int DH) {
    (void)D;
    return align256((long long)B * N * H * DH * esize(dt));   // behind dout [rows, H * DH]
}

extern "C" long long vitmi_attn_block_bwd_sums_offset(int dt, int B, int N, int D, int H, int DH) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || DH <= 0) return 0;
    const long long rows = (long long)B * N, I = (long long)H * DH;   // behind dout, dqkv and dxn
    return align256(rows * I * esize(dt)) + align256(rows * 3 * I * esize(dt)) + align256(rows * D * esize(dt));
}
extern "C" int vitmi_attn_block_bwd_sums_strips(int dt, int B, int N, int D, int H, int DH) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || DH <= 0) return 0;
    return vitmi_attn_small_bwd_ok(dt, B, N, D, H, DH) ? B : 0;
}

extern "C" int vitmi_attn_block_bwd(int dt, int B, int N, int D, int H, int DH, const void* dy, const void* x,
                                    const void* xn, const void* qkv, const void* o, const float* lse, const float* gamma,
                                    const float* mean, const float* rstd, const void* w_qkv, const void* w_qkv_t,
                                    const void* w_out, const void* w_out_t, void* dx, float* dgamma, float* dbeta,
                                    float* dw_qkv, float* db_qkv, float* dw_out, float* db_out, void* workspace,
                                    int qkv_layout, void* stream) {
    // dgamma == null: the vector gradients are the caller's as well, later (vitmi_small_sums_multi), from the per-image
    // partial sums of the one-launch kernel at vitmi_attn_block_bwd_sums_offset in the workspace it keeps until then
    const bool skip_sums = !dgamma;
    VITMI_CHECK_ARG(dy && x && xn && qkv && o && lse && gamma && mean && rstd && w_qkv && w_out && dx && workspace &&
                        ((dw_qkv != nullptr) == (dw_out != nullptr)) &&
                        (skip_sums ? (!dbeta && !db_qkv && !db_out && !dw_qkv) : dbeta != nullptr),
                    "vitmi_attn_block_bwd: null pointer");
    VITMI_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && DH > 0, "vitmi_attn_block_bwd: bad shape");
    VITMI_CHECK_ARG(!skip_sums || (!qkv_layout && w_qkv_t && w_out_t && vitmi_attn_small_bwd_ok(dt, B, N, D, H, DH)),
                    "vitmi_attn_block_bwd: dgamma == NULL (sums left to the caller) needs the one-launch kernel's shape, "
                    "row-major qkv and both transposed weights (vitmi_attn_block_bwd_sums_strips says 0 otherwise)");
    VITMI_CHECK_ARG(qkv_layout == 0 || qkv_layout == VITMI_QKV_HEAD_MAJOR, "vitmi_attn_block_bwd: bad qkv layout %d", qkv_layout);
    const int rows = B * N, I = H * DH;
    char* ws = (char*)workspace;
    void* dout = ws;
    void* dqkv = ws + align256((long long)rows * I * esize(dt));
    void* dxn = (char*)dqkv + align256((long long)rows * 3 * I * esize(dt));
    void* scratch = (char*)dxn + align256((long long)rows * D * esize(dt));
    void* wg_scratch = (char*)scratch + attn_chain_scratch(dt, B, N, D, H, DH);
    // dqkv lives in the workspace, so its layout is this call's own choice: planes when the three consumers below can
    // read them (a qkv bias wants the column sums of the row-major matrix; the input gradient wants the transposed weight)
    // dw_qkv == dw_out == null: the weight gradients are the caller's, later (vitmi_gemm_wgrad_multi), from dy, o, xn
    // and dqkv, which stays row-major at vitmi_attn_block_bwd_dqkv_offset in the workspace the caller keeps until then
    const bool skip_w = !dw_qkv;
    if (skip_w && !db_qkv && !qkv_layout && w_qkv_t && w_out_t && vitmi_attn_small_bwd_ok(dt, B, N, D, H, DH)) {
        // one launch for dO, the softmax backward, dxn and the LayerNorm backward (attention.hip); the sums for dgamma,
        // dbeta and the output bias are finished by a small kernel
        const int rc = vitmi_attn_small_bwd(B, N, D, H, DH, dy, x, qkv, o, lse, gamma, mean, rstd, w_out_t, w_qkv_t, dqkv, dx,
                                            (float*)scratch, stream);
        if (rc != VITMI_SMALL_NOT_TAKEN) {
            if (rc != VITMI_OK) return rc;
            if (skip_sums) return VITMI_OK;   // the caller reduces the partial sums
            return vitmi_small_sums_final(B, D, (const float*)scratch, dgamma, dbeta, db_out, stream);
        }
    }
    if (skip_sums) {
        vitmi_set_error("vitmi_attn_block_bwd: dgamma == NULL, but an operand misses the 16-byte alignment of the one-launch kernel");
        return VITMI_ERR_BAD_ARG;
    }
    const bool dq_planes = !skip_w && (g_qkv_planes & 2) && !db_qkv && w_qkv_t &&
                           vitmi_qkv_planes_ok(dt, B, N, D, H, DH, xn, w_qkv_t, dqkv);
    Side* sd = (dq_planes || skip_w) ? nullptr : overlap_side(rows, D, stream);  // (planes: heads of 64, never a small shape)
    if (sd) {
        STEP(side_fork(sd, stream));
        STEP(vitmi_gemm_wgrad(dt, D, I, rows, dy, D, o, I, dw_out, I, wg_scratch, sd->s));
    }
    STEP(input_grad(dt, rows, I, D, dy, w_out, w_out_t, dout, VITMI_EPI_NONE, nullptr, nullptr, nullptr, stream));
    STEP(vitmi_attn_bwd(qkv, o, dout, lse, nullptr, dqkv, scratch, dt, B, N, H, DH,
                        qkv_layout | (dq_planes ? VITMI_DQKV_HEAD_MAJOR : 0), stream));
    if (dq_planes) {
        STEP(vitmi_gemm_wgrad_pair_planes(rows, D, I, dy, D, o, I, dw_out, 3 * I, D, dqkv, xn, D, dw_qkv, wg_scratch, stream));
        STEP(vitmi_gemm_from_planes(rows, D, 3 * I, dqkv, w_qkv_t, dxn, stream));
    } else {
        if (sd) {
            STEP(side_fork(sd, stream, true));
            STEP(vitmi_gemm_wgrad(dt, 3 * I, D, rows, dqkv, 3 * I, xn, D, dw_qkv, D, wg_scratch, sd->s));
        } else if (!skip_w) {
            STEP(vitmi_gemm_wgrad_pair(dt, rows, D, I, dy, D, o, I, dw_out, 3 * I, D, dqkv, 3 * I, xn, D, dw_qkv, wg_scratch,
                                       stream));
        }
        if (db_qkv) STEP(vitmi_colsum(dqkv, dt, rows, 3 * I, 3 * I, db_qkv, scratch, stream));
        STEP(input_grad(dt, rows, D, 3 * I, dqkv, w_qkv, w_qkv_t, dxn, VITMI_EPI_NONE, nullptr, nullptr, nullptr, stream));
    }
    STEP(vitmi_layernorm_bwd(dxn, x, D, gamma, mean, rstd, dy, dx, dgamma, dbeta, db_out, scratch, dt, rows, D, stream));
    if (sd) STEP(side_join(sd, stream));
    return VITMI_OK;
}

// ------------------------------------------------------------------------------- inference (nothing kept for a backward)
// bf16 only.  LayerNorm never writes its normalised rows: a statistics pass over x, then the product on the raw rows
// with gamma folded into the weight and mean / rstd applied in the epilogue (VITMI_EPI_LN); the MLP's first product
// applies GELU in the same epilogue and stores no derivative.
static long long infer_stats_bytes(long long rows) { return 2 * align256(rows * (long long)sizeof(float)); }

extern "C" long long vitmi_attn_block_infer_workspace_bytes(int B, int N, int D, int H, int DH) {
    if (B <= 0 || N <= 0 || D <= 0 || H <= 0 || DH <= 0) return 0;
    const long long rows = (long long)B * N, I = (long long)H * DH;
    return infer_stats_bytes(rows) + align256(rows * 3 * I * 2) + align256(rows * I * 2) +
           align256((long long)B * H * N * (long long)sizeof(float));
}

extern "C" int vitmi_attn_block_infer(int B, int N, int D, int H, int DH, const void* x, float eps, const void* wq_folded,
                                      const float* bq_folded, const float* wq_rowsum, const void* w_out,
                                      const float* b_out, void* probs, void* y, void* workspace, void* stream) {
    VITMI_CHECK_ARG(x && wq_folded && bq_folded && wq_rowsum && w_out && y && workspace, "vitmi_attn_block_infer: null pointer");
    VITMI_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && DH > 0, "vitmi_attn_block_infer: bad shape");
    const int rows = B * N, I = H * DH;
    char* ws = (char*)workspace;
    float* mean = (float*)ws;
    float* rstd = (float*)(ws + align256((long long)rows * sizeof(float)));
    void* qkv = ws + infer_stats_bytes(rows);
    void* o = (char*)qkv + align256((long long)rows * 3 * I * 2);
    float* lse = (float*)((char*)o + align256((long long)rows * I * 2));
    STEP(vitmi_row_stats(x, D, mean, rstd, VITMI_BF16, rows, D, eps, stream));
    STEP(vitmi_gemm_ln(rows, 3 * I, D, x, D, wq_folded, bq_folded, wq_rowsum, mean, rstd, qkv, 3 * I, VITMI_EPI_LN, stream));
    STEP(vitmi_attn_fwd(qkv, o, lse, probs, VITMI_BF16, B, N, H, DH, 0, stream));
    STEP(vitmi_gemm(VITMI_BF16, rows, D, I, o, I, 1, w_out, I, 1, y, D, VITMI_BF16, VITMI_EPI_NONE, b_out, x, D, nullptr, 0,
                    stream));
    return VITMI_OK;
}

extern "C" long long vitmi_ff_block_infer_workspace_bytes(int rows, int D, int M) {
    if (rows <= 0 || D <= 0 || M <= 0) return 0;
    return infer_stats_bytes(rows) + align256((long long)rows * M * 2);
}

extern "C" int vitmi_ff_block_infer(int rows, int D, int M, const void* x, float eps, const void* w1_folded,
                                    const float* b1_folded, const float* w1_rowsum, const void* w2, const float* b2,
                                    void* y, void* workspace, void* stream) {
    VITMI_CHECK_ARG(x && w1_folded && b1_folded && w1_rowsum && w2 && y && workspace, "vitmi_ff_block_infer: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && M > 0, "vitmi_ff_block_infer: bad shape rows=%d D=%d M=%d", rows, D, M);
    char* ws = (char*)workspace;
    float* mean = (float*)ws;
    float* rstd = (float*)(ws + align256((long long)rows * sizeof(float)));
    void* h = ws + infer_stats_bytes(rows);
    STEP(vitmi_row_stats(x, D, mean, rstd, VITMI_BF16, rows, D, eps, stream));
    STEP(vitmi_gemm_ln(rows, M, D, x, D, w1_folded, b1_folded, w1_rowsum, mean, rstd, h, M, VITMI_EPI_LN_GELU, stream));
    STEP(vitmi_gemm(VITMI_BF16, rows, D, M, h, M, 1, w2, M, 1, y, D, VITMI_BF16, VITMI_EPI_NONE, b2, x, D, nullptr, 0, stream));
    return VITMI_OK;
}
