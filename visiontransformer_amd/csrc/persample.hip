// Per-sample gradient norms for DP-SGD (defence_method/DPSGD.py:114-136 clips every sample's gradient to
// l2_norm_clip before summing).  The reference obtains per-sample gradients by running the batch one image at
// a time; here one backward over the whole batch accumulates ||g_i||^2 per image without materialising g_i:
// rows of the [tokens, features] activations are cut into segments of `seg` consecutive rows (one image each)
// and every routine adds its parameter's share to sqnorm[segment].  Each sqnorm entry has one writer per
// launch and all reductions run in a fixed order, so the norms are reproducible.
#include "common.h"

int vitmi_gemm_tn_seg_tiles(int P, int Q);
int vitmi_gemm_tn_seg_sq(int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                         float* partial, int* tiles, hipStream_t s);
bool vitmi_gemm_tn_fast_ok(int P, int Q, const void* U, long long ldu, const void* V, long long ldv, const void* C,
                           long long ldc);
bool vitmi_gemm_ghost_sq_ok(int P, int Q, int seg, const void* U, long long ldu, const void* V, long long ldv);
int vitmi_gemm_ghost_sq(int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                        float* sqnorm, hipStream_t s);

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// sqnorm[s] += sum over tiles of partial[s][tile]
__global__ void seg_tile_sum_kernel(const float* __restrict__ partial, int tiles, int nseg, float* __restrict__ sqnorm) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    float t = 0.f;
    for (int i = 0; i < tiles; ++i) t += partial[(long long)s * tiles + i];
    sqnorm[s] += t;
}

// sqnorm[s] += sum of squares of a dense matrix of n elements (one block; the fallback path)
__global__ __launch_bounds__(256) void dense_sq_accum_kernel(const float* __restrict__ x, long long n, float* __restrict__ dst) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) acc = fmaf(x[i], x[i], acc);
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) *dst += acc;
}

// sqnorm[s] += || sum_{t in segment s} x[t, :] ||^2
template <typename T>
__global__ __launch_bounds__(256) void seg_colsum_sq_kernel(const T* __restrict__ x, long long ld, long long rows, int cols,
                                                           int seg, float* __restrict__ sqnorm, float* __restrict__ sums) {
    __shared__ float red[4];
    const long long r0 = (long long)blockIdx.x * seg;
    const long long r1 = r0 + seg < rows ? r0 + seg : rows;
    float acc = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s = 0.f;
        for (long long t = r0; t < r1; ++t) s += to_f32(x[t * ld + c]);
        if (sums) sums[(long long)blockIdx.x * cols + c] = s;
        acc = fmaf(s, s, acc);
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) sqnorm[blockIdx.x] += acc;
}

// sqnorm[s] += || sum_t dy[t,:] * xhat[t,:] ||^2 + || sum_t dy[t,:] ||^2, xhat = (x - mean) * rstd: the LayerNorm
// weight and bias gradients of one image
template <typename T>
__global__ __launch_bounds__(256) void seg_ln_sq_kernel(const T* __restrict__ dy, const T* __restrict__ x, long long ldx,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       long long rows, int D, int seg, float* __restrict__ sqnorm,
                                                       float* __restrict__ sums_b, float* __restrict__ sums_g) {
    __shared__ float red[4];
    const long long r0 = (long long)blockIdx.x * seg;
    const long long r1 = r0 + seg < rows ? r0 + seg : rows;
    float acc = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) {
        float sg = 0.f, sb = 0.f;
        for (long long t = r0; t < r1; ++t) {
            const float d = to_f32(dy[t * D + c]);
            sg = fmaf(d, (to_f32(x[t * ldx + c]) - mean[t]) * rstd[t], sg);
            sb += d;
        }
        if (sums_b) {
            sums_b[(long long)blockIdx.x * D + c] = sb;
            sums_g[(long long)blockIdx.x * D + c] = sg;
        }
        acc = fmaf(sg, sg, fmaf(sb, sb, acc));
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) sqnorm[blockIdx.x] += acc;
}

// sqnorm[s] += sum over the segment's rows and all columns of x^2
template <typename T>
__global__ __launch_bounds__(256) void seg_sq_kernel(const T* __restrict__ x, long long ld, long long rows, int cols, int seg,
                                                    float* __restrict__ sqnorm) {
    __shared__ float red[4];
    const long long r0 = (long long)blockIdx.x * seg;
    const long long r1 = r0 + seg < rows ? r0 + seg : rows;
    float acc = 0.f;
    for (long long t = r0; t < r1; ++t)
        for (int c = threadIdx.x; c < cols; c += 256) {
            const float v = to_f32(x[t * ld + c]);
            acc = fmaf(v, v, acc);
        }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) sqnorm[blockIdx.x] += acc;
}

// Wide variants of the two column-sum kernels for 16-byte aligned rows: one 8-wave block per segment.  A wave
// owns a group of 64 * VEC consecutive columns (16 bytes per lane) and every R-th row of the segment; the R row
// groups of a column group are added through LDS in row-group order, then squared and summed in wave order.
template <typename T> struct SegVec { static constexpr int N = 16 / sizeof(T); T v[16 / sizeof(T)]; };
constexpr int SEG_WAVES = 8;

template <typename T, bool LN>
__global__ __launch_bounds__(64 * SEG_WAVES) void seg_colsum_sq_wide_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                            long long ld, long long ldx,
                                                                            const float* __restrict__ mean,
                                                                            const float* __restrict__ rstd, long long rows,
                                                                            int cols, int seg, float* __restrict__ sqnorm,
                                                                            float* __restrict__ sums_b,
                                                                            float* __restrict__ sums_g) {
    constexpr int VEC = SegVec<T>::N;
    typedef SegVec<T> __attribute__((aligned(16))) VT;
    __shared__ float part[SEG_WAVES][64 * VEC * (LN ? 2 : 1)];
    __shared__ float red[SEG_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * seg;
    const long long r1 = r0 + seg < rows ? r0 + seg : rows;
    const int ncg = (cols + 64 * VEC - 1) / (64 * VEC);
    float total = 0.f;                                   // wave 0 .. P-1 accumulate their column groups' squares
    for (int cg0 = 0; cg0 < ncg; cg0 += SEG_WAVES) {
        const int P = ncg - cg0 < SEG_WAVES ? ncg - cg0 : SEG_WAVES;      // column groups in this pass
        const int R = SEG_WAVES / P;                                       // row groups per column group
        const int cg = wave % P, rg = wave / P;
        const int c = ((cg0 + cg) * 64 + lane) * VEC;
        float sb[VEC], sg[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sb[j] = sg[j] = 0.f;
        if (rg < R && c < cols) {
            long long t = r0 + rg;
            for (; t + R < r1; t += 2 * R) {             // two rows in flight
                const VT a = *reinterpret_cast<const VT*>(dy + t * ld + c);
                const VT b = *reinterpret_cast<const VT*>(dy + (t + R) * ld + c);
                if (LN) {
                    const VT xa = *reinterpret_cast<const VT*>(x + t * ldx + c);
                    const VT xb = *reinterpret_cast<const VT*>(x + (t + R) * ldx + c);
                    const float ma = mean[t], ra = rstd[t], mb = mean[t + R], rb = rstd[t + R];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        sg[j] = fmaf(to_f32(a.v[j]), (to_f32(xa.v[j]) - ma) * ra, sg[j]);
                        sg[j] = fmaf(to_f32(b.v[j]), (to_f32(xb.v[j]) - mb) * rb, sg[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) sb[j] += to_f32(a.v[j]) + to_f32(b.v[j]);
            }
            if (t < r1) {
                const VT a = *reinterpret_cast<const VT*>(dy + t * ld + c);
                if (LN) {
                    const VT xa = *reinterpret_cast<const VT*>(x + t * ldx + c);
                    const float ma = mean[t], ra = rstd[t];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) sg[j] = fmaf(to_f32(a.v[j]), (to_f32(xa.v[j]) - ma) * ra, sg[j]);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) sb[j] += to_f32(a.v[j]);
            }
        }
        __syncthreads();                                 // the previous pass has read `part`
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            part[wave][lane * VEC + j] = sb[j];
            if (LN) part[wave][64 * VEC + lane * VEC + j] = sg[j];
        }
        __syncthreads();
        if (wave < P && c < cols) {                      // wave == cg here (rg == 0)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float b = 0.f, g = 0.f;
                for (int r = 0; r < R; ++r) {
                    b += part[r * P + wave][lane * VEC + j];
                    if (LN) g += part[r * P + wave][64 * VEC + lane * VEC + j];
                }
                if (sums_b) {                             // the segment's column sums themselves, for a later weighted sum
                    sums_b[(long long)blockIdx.x * cols + c + j] = b;
                    if (LN) sums_g[(long long)blockIdx.x * cols + c + j] = g;
                }
                total = fmaf(b, b, total);
                if (LN) total = fmaf(g, g, total);
            }
        }
    }
    total = wave_sum(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < SEG_WAVES; ++w) t += red[w];
        sqnorm[blockIdx.x] += t;
    }
}

// out[t][:] = x[t][:] * coef[t / seg]: one operand of a weight-gradient product with every image's rows scaled by
// the image's clipping factor (sum_i c_i dY_i^T X_i = dY^T (c X))
template <typename T, int VEC>
__global__ __launch_bounds__(256) void seg_scale_rows_kernel(const T* __restrict__ x, long long ld, long long rows, int cols,
                                                            int seg, const float* __restrict__ coef, T* __restrict__ out,
                                                            long long ldo) {
    const int cpr = cols / VEC;                            // vectors per row
    const long long n = rows * cpr;
    struct __attribute__((aligned(sizeof(T) * VEC))) V { T v[VEC]; };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long t = i / cpr;
        const int c = (int)(i - t * cpr) * VEC;
        const float k = coef[t / seg];
        V a = *reinterpret_cast<const V*>(x + t * ld + c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a.v[j] = from_f32<T>(to_f32(a.v[j]) * k);
        *reinterpret_cast<V*>(out + t * ldo + c) = a;
    }
}

// out[c] = sum_i coef[i] * x[i][c], i < n in order: a vector parameter's clipped gradient from the per-image sums
template <typename T>
__global__ __launch_bounds__(256) void seg_combine_kernel(const T* __restrict__ x, long long ld, int n, long long cols,
                                                         const float* __restrict__ coef, float* __restrict__ out) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float a0 = 0.f, a1 = 0.f;
    int i = 0;
    for (; i + 1 < n; i += 2) {
        a0 = fmaf(coef[i], to_f32(x[(long long)i * ld + c]), a0);
        a1 = fmaf(coef[i + 1], to_f32(x[(long long)(i + 1) * ld + c]), a1);
    }
    if (i < n) a0 = fmaf(coef[i], to_f32(x[(long long)i * ld + c]), a0);
    out[c] = a0 + a1;
}

inline bool wide_ok(const void* p, long long ld, int cols, int dt) {
    const int vec = dt == VITMI_F32 ? 4 : 8;
    return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && ld % vec == 0 && cols % vec == 0;
}

bool fast_seg_ok(int dt, int P, int Q, const void* U, long long ldu, const void* V, long long ldv) {
    // the alignment rules of the direct-to-LDS tn kernel; the output pointer plays no part in segment mode
    alignas(16) static const char dummy[16] = {0};
    return dt == VITMI_BF16 && vitmi_gemm_tn_fast_ok(P, Q, U, ldu, V, ldv, dummy, (long long)((Q + 3) & ~3));
}

}  // namespace

// 1 (default): per-image weight-gradient norms of images up to 256 tokens through the token Gram matrices; 0: through
// the per-image tiles of the gradient itself (for A/B timing and for checking one against the other)
static int g_ghost_norms = 1;
extern "C" int vitmi_debug_set_ghost_norms(int on) {
    g_ghost_norms = on ? 1 : 0;
    return VITMI_OK;
}

extern "C" long long vitmi_wgrad_sqnorm_workspace_bytes(int dt, int P, int Q, int T, int seg) {
    if (P <= 0 || Q <= 0 || T <= 0 || seg <= 0) return 0;
    const long long nseg = (T + seg - 1) / seg;
    const long long fast = nseg * vitmi_gemm_tn_seg_tiles(P, Q) * (long long)sizeof(float);
    const long long slow = (long long)P * Q * (long long)sizeof(float) + vitmi_gemm_wgrad_workspace_bytes(dt, P, Q, seg) + 256;
    return (fast > slow ? fast : slow) + 256;
}

extern "C" int vitmi_wgrad_sqnorm(int dt, int P, int Q, int T, int seg, const void* U, long long ldu, const void* V,
                                  long long ldv, float* sqnorm, void* workspace, void* stream) {
    VITMI_CHECK_ARG(U && V && sqnorm && workspace, "vitmi_wgrad_sqnorm: null pointer");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_wgrad_sqnorm: bad dtype %d", dt);
    VITMI_CHECK_ARG(P > 0 && Q > 0 && T > 0 && seg > 0 && ldu >= P && ldv >= Q, "vitmi_wgrad_sqnorm: bad shape");
    hipStream_t s = as_stream(stream);
    const int nseg = (T + seg - 1) / seg;
    // images of up to 256 tokens: the norm from the two token Gram matrices (2 n^2 (P + Q) flops instead of 2 n P Q,
    // no per-tile partial sums); worth it once the gradient is larger than the Grams
    if (g_ghost_norms && dt == VITMI_BF16 && seg >= 2 && (long long)P * Q > (long long)seg * (P + Q) &&
        vitmi_gemm_ghost_sq_ok(P, Q, seg, U, ldu, V, ldv)) {
        ProfScope prof(VITMI_FAM_OTHER, s, 2.0 * nseg * (double)seg * seg * (P + Q));
        return vitmi_gemm_ghost_sq(P, Q, T, seg, U, ldu, V, ldv, sqnorm, s);
    }
    if (fast_seg_ok(dt, P, Q, U, ldu, V, ldv)) {
        int tiles = 0;
        int rc = vitmi_gemm_tn_seg_sq(P, Q, T, seg, U, ldu, V, ldv, (float*)workspace, &tiles, s);
        if (rc != VITMI_OK) return rc;
        hipLaunchKernelGGL(seg_tile_sum_kernel, dim3(ceil_div(nseg, 256)), dim3(256), 0, s, (const float*)workspace, tiles,
                           nseg, sqnorm);
        VITMI_LAUNCH_CHECK("vitmi_wgrad_sqnorm");
        return VITMI_OK;
    }
    // any dtype, any alignment: the segment's gradient through the ordinary weight-gradient product, then its norm
    const long long es = dt == VITMI_F32 ? 4 : 2;
    float* g = (float*)workspace;
    void* ws2 = (char*)workspace + (((long long)P * Q * 4 + 255) & ~255ll);
    for (int i = 0; i < nseg; ++i) {
        const int len = (i + 1) * seg <= T ? seg : T - i * seg;
        int rc = vitmi_gemm_wgrad(dt, P, Q, len, (const char*)U + (long long)i * seg * ldu * es, ldu,
                                  (const char*)V + (long long)i * seg * ldv * es, ldv, g, Q, ws2, stream);
        if (rc != VITMI_OK) return rc;
        hipLaunchKernelGGL(dense_sq_accum_kernel, dim3(1), dim3(256), 0, s, g, (long long)P * Q, sqnorm + i);
    }
    VITMI_LAUNCH_CHECK("vitmi_wgrad_sqnorm");
    return VITMI_OK;
}

#define SEG_ARGS_OK(who)                                                                             \
    VITMI_CHECK_ARG(x && sqnorm, who ": null pointer");                                            \
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, who ": bad dtype %d", dt);                \
    VITMI_CHECK_ARG(rows > 0 && cols > 0 && seg > 0 && ld >= cols, who ": bad shape")

extern "C" int vitmi_seg_colsum_sqnorm(const void* x, int dt, long long rows, int cols, long long ld, int seg,
                                       float* sqnorm, float* sums, void* stream) {
    SEG_ARGS_OK("vitmi_seg_colsum_sqnorm");
    hipStream_t s = as_stream(stream);
    const int nseg = (int)((rows + seg - 1) / seg);
    if (wide_ok(x, ld, cols, dt) && seg >= 2 * SEG_WAVES) {
        if (dt == VITMI_F32)
            hipLaunchKernelGGL((seg_colsum_sq_wide_kernel<float, false>), dim3(nseg), dim3(64 * SEG_WAVES), 0, s,
                               (const float*)x, (const float*)nullptr, ld, 0ll, (const float*)nullptr, (const float*)nullptr,
                               rows, cols, seg, sqnorm, sums, (float*)nullptr);
        else
            hipLaunchKernelGGL((seg_colsum_sq_wide_kernel<bf16_t, false>), dim3(nseg), dim3(64 * SEG_WAVES), 0, s,
                               (const bf16_t*)x, (const bf16_t*)nullptr, ld, 0ll, (const float*)nullptr,
                               (const float*)nullptr, rows, cols, seg, sqnorm, sums, (float*)nullptr);
    } else if (dt == VITMI_F32)
        hipLaunchKernelGGL((seg_colsum_sq_kernel<float>), dim3(nseg), dim3(256), 0, s, (const float*)x, ld, rows, cols, seg,
                           sqnorm, sums);
    else
        hipLaunchKernelGGL((seg_colsum_sq_kernel<bf16_t>), dim3(nseg), dim3(256), 0, s, (const bf16_t*)x, ld, rows, cols,
                           seg, sqnorm, sums);
    VITMI_LAUNCH_CHECK("vitmi_seg_colsum_sqnorm");
    return VITMI_OK;
}

extern "C" int vitmi_seg_sqnorm(const void* x, int dt, long long rows, int cols, long long ld, int seg, float* sqnorm,
                                void* stream) {
    SEG_ARGS_OK("vitmi_seg_sqnorm");
    hipStream_t s = as_stream(stream);
    const int nseg = (int)((rows + seg - 1) / seg);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL((seg_sq_kernel<float>), dim3(nseg), dim3(256), 0, s, (const float*)x, ld, rows, cols, seg, sqnorm);
    else
        hipLaunchKernelGGL((seg_sq_kernel<bf16_t>), dim3(nseg), dim3(256), 0, s, (const bf16_t*)x, ld, rows, cols, seg,
                           sqnorm);
    VITMI_LAUNCH_CHECK("vitmi_seg_sqnorm");
    return VITMI_OK;
}

extern "C" int vitmi_seg_layernorm_sqnorm(const void* dy, const void* x, long long ldx, const float* mean,
                                          const float* rstd, int dt, long long rows, int D, int seg, float* sqnorm,
                                          float* sums_beta, float* sums_gamma, void* stream) {
    VITMI_CHECK_ARG(dy && x && mean && rstd && sqnorm, "vitmi_seg_layernorm_sqnorm: null pointer");
    VITMI_CHECK_ARG(!sums_beta == !sums_gamma, "vitmi_seg_layernorm_sqnorm: both sums or neither");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_seg_layernorm_sqnorm: bad dtype %d", dt);
    VITMI_CHECK_ARG(rows > 0 && D > 0 && seg > 0 && ldx >= D, "vitmi_seg_layernorm_sqnorm: bad shape");
    hipStream_t s = as_stream(stream);
    const int nseg = (int)((rows + seg - 1) / seg);
    if (wide_ok(dy, D, D, dt) && wide_ok(x, ldx, D, dt) && seg >= 2 * SEG_WAVES) {
        if (dt == VITMI_F32)
            hipLaunchKernelGGL((seg_colsum_sq_wide_kernel<float, true>), dim3(nseg), dim3(64 * SEG_WAVES), 0, s,
                               (const float*)dy, (const float*)x, (long long)D, ldx, mean, rstd, rows, D, seg, sqnorm,
                               sums_beta, sums_gamma);
        else
            hipLaunchKernelGGL((seg_colsum_sq_wide_kernel<bf16_t, true>), dim3(nseg), dim3(64 * SEG_WAVES), 0, s,
                               (const bf16_t*)dy, (const bf16_t*)x, (long long)D, ldx, mean, rstd, rows, D, seg, sqnorm,
                               sums_beta, sums_gamma);
    } else if (dt == VITMI_F32)
        hipLaunchKernelGGL((seg_ln_sq_kernel<float>), dim3(nseg), dim3(256), 0, s, (const float*)dy, (const float*)x, ldx,
                           mean, rstd, rows, D, seg, sqnorm, sums_beta, sums_gamma);
    else
        hipLaunchKernelGGL((seg_ln_sq_kernel<bf16_t>), dim3(nseg), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, ldx,
                           mean, rstd, rows, D, seg, sqnorm, sums_beta, sums_gamma);
    VITMI_LAUNCH_CHECK("vitmi_seg_layernorm_sqnorm");
    return VITMI_OK;
}

extern "C" int vitmi_seg_scale_rows(const void* x, int dt, long long rows, int cols, long long ld, int seg,
                                    const float* coef, void* out, long long ldo, void* stream) {
    VITMI_CHECK_ARG(x && coef && out, "vitmi_seg_scale_rows: null pointer");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_seg_scale_rows: bad dtype %d", dt);
    VITMI_CHECK_ARG(rows > 0 && cols > 0 && seg > 0 && ld >= cols && ldo >= cols, "vitmi_seg_scale_rows: bad shape");
    hipStream_t s = as_stream(stream);
    const int vec = dt == VITMI_F32 ? 4 : 8;
    const bool wide = wide_ok(x, ld, cols, dt) && wide_ok(out, ldo, cols, dt);
    const long long n = rows * (wide ? cols / vec : cols);
    const int blocks = (int)std::min<long long>((n + 255) / 256, 256 * 16);
    ProfScope prof(VITMI_FAM_OTHER, s, 2.0 * rows * cols * (dt == VITMI_F32 ? 4 : 2));
    if (dt == VITMI_F32) {
        if (wide)
            hipLaunchKernelGGL((seg_scale_rows_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)x, ld, rows, cols,
                               seg, coef, (float*)out, ldo);
        else
            hipLaunchKernelGGL((seg_scale_rows_kernel<float, 1>), dim3(blocks), dim3(256), 0, s, (const float*)x, ld, rows, cols,
                               seg, coef, (float*)out, ldo);
    } else {
        if (wide)
            hipLaunchKernelGGL((seg_scale_rows_kernel<bf16_t, 8>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, ld, rows,
                               cols, seg, coef, (bf16_t*)out, ldo);
        else
            hipLaunchKernelGGL((seg_scale_rows_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, ld, rows,
                               cols, seg, coef, (bf16_t*)out, ldo);
    }
    VITMI_LAUNCH_CHECK("vitmi_seg_scale_rows");
    return VITMI_OK;
}

extern "C" int vitmi_seg_combine(const void* x, int dt, int n, long long cols, long long ld, const float* coef, float* out,
                                 void* stream) {
    VITMI_CHECK_ARG(x && coef && out, "vitmi_seg_combine: null pointer");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_seg_combine: bad dtype %d", dt);
    VITMI_CHECK_ARG(n > 0 && cols > 0 && ld >= cols, "vitmi_seg_combine: bad shape");
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_OTHER, s, 1.0 * n * cols * (dt == VITMI_F32 ? 4 : 2));
    const unsigned blocks = (unsigned)((cols + 255) / 256);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL((seg_combine_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)x, ld, n, cols, coef, out);
    else
        hipLaunchKernelGGL((seg_combine_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, ld, n, cols, coef,
                           out);
    VITMI_LAUNCH_CHECK("vitmi_seg_combine");
    return VITMI_OK;
}
