// HBM-bound kernels of the ViT encoder path: casts, residual adds, column sums, token
// assembly with (gathered) positional embedding, LayerNorm forward/backward, mean pooling,
// fused Adam.  All are coalesced row/stream kernels: 8-16 B per lane, one wave per token row
// where a row reduction is needed, fp32 arithmetic, deterministic two-stage reductions.
#include "common.h"

namespace {

template <typename T, int V> struct Vec { T v[V]; };

template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* p, float (&f)[V]) {
    typedef Vec<T, V> __attribute__((aligned(sizeof(T) * V))) VT;
    VT t = *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = to_f32(t.v[i]);
}
template <typename T, int V>
__device__ __forceinline__ void store_vec(T* p, const float (&f)[V]) {
    typedef Vec<T, V> __attribute__((aligned(sizeof(T) * V))) VT;
    VT t;
#pragma unroll
    for (int i = 0; i < V; ++i) t.v[i] = from_f32<T>(f[i]);
    *reinterpret_cast<VT*>(p) = t;
}

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---------------------------------------------------------------- cast / add
template <typename S, typename D, int V>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const long long stride = (long long)gridDim.x * blockDim.x * V;
    for (; i + V <= n; i += stride) {
        float f[V];
        load_vec<S, V>(src + i, f);
        store_vec<D, V>(dst + i, f);
    }
    if (i < n && i + V > n) {
        for (long long j = i; j < n; ++j) dst[j] = from_f32<D>(to_f32(src[j]));
    }
}

template <typename S, typename D>
int launch_cast(const void* src, void* dst, long long n, hipStream_t s) {
    if (n == 0) return VITMI_OK;
    const bool vec = aligned_to(src, 16) && aligned_to(dst, 16);
    const int threads = 256;
    if (vec) {
        constexpr int V = 4;
        int blocks = (int)std::min<long long>((n + threads * V - 1) / (threads * V), 4096);
        hipLaunchKernelGGL((cast_kernel<S, D, V>), dim3(blocks), dim3(threads), 0, s, (const S*)src, (D*)dst, n);
    } else {
        int blocks = (int)std::min<long long>((n + threads - 1) / threads, 4096);
        hipLaunchKernelGGL((cast_kernel<S, D, 1>), dim3(blocks), dim3(threads), 0, s, (const S*)src, (D*)dst, n);
    }
    VITMI_LAUNCH_CHECK("vitmi_cast");
    return VITMI_OK;
}

template <typename T, int V>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    const long long stride = (long long)gridDim.x * blockDim.x * V;
    for (; i + V <= n; i += stride) {
        float fa[V], fb[V];
        load_vec<T, V>(a + i, fa);
        load_vec<T, V>(b + i, fb);
#pragma unroll
        for (int j = 0; j < V; ++j) fa[j] += fb[j];
        store_vec<T, V>(y + i, fa);
    }
    if (i < n && i + V > n) {
        for (long long j = i; j < n; ++j) y[j] = from_f32<T>(to_f32(a[j]) + to_f32(b[j]));
    }
}

// y = clamp(x * scale + addend, lo, hi): the "Scale" and "noise" image augmentations (masking_generator.py:210-218)
template <typename T>
__global__ void scale_add_clip_kernel(const T* __restrict__ x, const float* __restrict__ addend, T* __restrict__ y,
                                      long long n, float scale, float lo, float hi) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = to_f32(x[i]) * scale;
        if (addend) v += addend[i];
        y[i] = from_f32<T>(fminf(fmaxf(v, lo), hi));
    }
}

// ---------------------------------------------------------------- column sums
// stage 1: block (strip, chunk): partial[chunk][c] = sum over the chunk's rows; lane owns V
// adjacent columns, the 4 waves of a block interleave rows and combine through LDS.
template <typename T, int V>
__global__ void colsum_partial_kernel(const T* __restrict__ x, long long rows, int cols, long long ld,
                                      float* __restrict__ partial, int rows_per_chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * V;
    const long long r_begin = (long long)blockIdx.y * rows_per_chunk;
    const long long r_end = r_begin + rows_per_chunk < rows ? r_begin + rows_per_chunk : rows;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (c0 < cols) {
        // four rows per trip, their loads independent of one another: a wave keeps four requests in flight
        // instead of paying one memory round trip per row
        long long r = r_begin + wave;
        for (; r + 12 < r_end; r += 16) {
            float f[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_vec<T, V>(x + (r + 4 * u) * ld + c0, f[u]);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += (f[0][j] + f[1][j]) + (f[2][j] + f[3][j]);
        }
        for (; r < r_end; r += 4) {
            float f[V];
            load_vec<T, V>(x + r * ld + c0, f);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += f[j];
        }
    }
    __shared__ float red[4][64 * V];
#pragma unroll
    for (int j = 0; j < V; ++j) red[wave][lane * V + j] = acc[j];
    __syncthreads();
    if (wave == 0 && c0 < cols) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float s = red[0][lane * V + j] + red[1][lane * V + j] + red[2][lane * V + j] + red[3][lane * V + j];
            partial[(long long)blockIdx.y * cols + c0 + j] = s;
        }
    }
}

// out[c] = sum_k partial[k][c] in a fixed order: 64 columns x 16 row groups per block, the groups'
// partial sums combined through LDS
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int chunks, int cols,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int k = grp;
        for (; k + 16 < chunks; k += 32) {
            s0 += partial[(long long)k * cols + c];
            s1 += partial[(long long)(k + 16) * cols + c];
        }
        for (; k < chunks; k += 16) s0 += partial[(long long)k * cols + c];
    }
    __shared__ float red[16][64];
    red[grp][lane] = s0 + s1;
    __syncthreads();
    if (grp == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][lane];
        out[c] = s;
    }
}

inline int colsum_chunks(long long rows) {
    long long c = (rows + 63) / 64;
    if (c > 512) c = 512;
    if (c < 1) c = 1;
    return (int)c;
}

template <typename T>
int launch_colsum(const void* x, long long rows, int cols, long long ld, float* out, void* workspace, hipStream_t s) {
    const int chunks = colsum_chunks(rows);
    const int rows_per_chunk = (int)((rows + chunks - 1) / chunks);
    float* partial = (float*)workspace;
    const bool vec = (cols % 4 == 0) && (ld % 4 == 0) && aligned_to(x, 16);
    if (vec) {
        dim3 grid(ceil_div(cols, 64 * 4), chunks);
        hipLaunchKernelGGL((colsum_partial_kernel<T, 4>), grid, dim3(256), 0, s, (const T*)x, rows, cols, ld, partial,
                           rows_per_chunk);
    } else {
        dim3 grid(ceil_div(cols, 64), chunks);
        hipLaunchKernelGGL((colsum_partial_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, rows, cols, ld, partial,
                           rows_per_chunk);
    }
    VITMI_LAUNCH_CHECK("vitmi_colsum(partial)");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 64)), dim3(1024), 0, s, partial, chunks, cols, out);
    VITMI_LAUNCH_CHECK("vitmi_colsum(final)");
    return VITMI_OK;
}

// ---------------------------------------------------------------- token assembly (+ gathered PE)
// one wave per output token row; lanes sweep the row V elements at a time.
template <typename T, int V>
__global__ void tokens_assemble_fwd_kernel(const T* __restrict__ patch, const float* __restrict__ cls,
                                           const float* __restrict__ pe, const long long* __restrict__ seq_ord,
                                           T* __restrict__ out, int B, int N, int D) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (long long)B * N) return;
    const int b = (int)(row / N), i = (int)(row % N);
    // a negative index means "no positional row for this token" (zero embedding)
    const long long pr = pe ? (seq_ord ? seq_ord[i] : (long long)i) : -1;
    const float* pe_row = pr >= 0 ? pe + pr * D : nullptr;
    T* o = out + row * D;
    const T* src = i > 0 ? patch + ((long long)b * (N - 1) + (i - 1)) * D : nullptr;
    for (int c = lane * V; c < D; c += 64 * V) {
        float f[V];
        if (i > 0) {
            load_vec<T, V>(src + c, f);
        } else {
            load_vec<float, V>(cls + c, f);
        }
        if (pe_row) {
            float p[V];
            load_vec<float, V>(pe_row + c, p);
#pragma unroll
            for (int j = 0; j < V; ++j) f[j] += p[j];
        }
        store_vec<T, V>(o + c, f);
    }
}

// dpe[r,:] = sum over positions i (ascending) with row(i) == r of possum[i,:]; dcls = possum[0,:]
__global__ void tokens_scatter_rows_kernel(const float* __restrict__ possum, const long long* __restrict__ seq_ord,
                                           float* __restrict__ dcls, float* __restrict__ dpe, int pe_rows, int N,
                                           int D) {
    const int r = blockIdx.x;  // r == pe_rows -> the cls-gradient block
    if (r == pe_rows) {
        for (int c = threadIdx.x; c < D; c += blockDim.x) dcls[c] = possum[c];
        return;
    }
    if (!dpe) return;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < N; ++i) {
            const long long row = seq_ord ? seq_ord[i] : (long long)i;
            if (row == r) s += possum[(long long)i * D + c];
        }
        dpe[(long long)r * D + c] = s;
    }
}

// ---------------------------------------------------------------- dropout
// y[i] = x[i] * keep(i) / (1 - p), keep(i) = [u32(seed, i) >= p * 2^32]: a counter-based generator (two rounds of the
// splitmix64 finaliser over seed and element index), so the backward pass regenerates the mask from the seed
// instead of storing it.  One 64-bit draw serves two elements.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, unsigned long long seed,
                               unsigned threshold, float scale) {
    const long long pairs = (n + 1) >> 1;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < pairs; k += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r = mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(k + 1)) ^ seed);
        const long long i = 2 * k;
        y[i] = from_f32<T>((unsigned)r >= threshold ? to_f32(x[i]) * scale : 0.f);
        if (i + 1 < n) y[i + 1] = from_f32<T>((unsigned)(r >> 32) >= threshold ? to_f32(x[i + 1]) * scale : 0.f);
    }
}

// ---------------------------------------------------------------- LayerNorm
// ITERS = ceil(D / (64*V)) is a template parameter so the per-lane row slice stays in registers.

template <typename T, int V, int LN_MAX_ITERS>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, long long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long long rows, int D, float eps) {
    // a wave walks rows (grid-strided) one row ahead: the next row's load is in flight while the current row is
    // reduced, normalised and stored; gamma and beta stay in registers across rows
    typedef Vec<T, V> __attribute__((aligned(sizeof(T) * V))) VT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long stride = (long long)gridDim.x * nw;
    float g[LN_MAX_ITERS][V], bt[LN_MAX_ITERS][V];
#pragma unroll
    for (int it = 0; it < LN_MAX_ITERS; ++it) {
        const int c = (it * 64 + lane) * V;
#pragma unroll
        for (int j = 0; j < V; ++j) g[it][j] = bt[it][j] = 0.f;
        if (c < D) {
            load_vec<float, V>(gamma + c, g[it]);
            load_vec<float, V>(beta + c, bt[it]);
        }
    }
    VT nx[LN_MAX_ITERS];
    auto fetch = [&](long long row) {
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) nx[it] = *reinterpret_cast<const VT*>(x + row * ldx + c);
        }
    };
    long long row = (long long)blockIdx.x * nw + wave;
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        float v[LN_MAX_ITERS][V];
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                v[it][j] = c < D ? to_f32(nx[it].v[j]) : 0.f;
                s += v[it][j];
            }
        }
        if (row + stride < rows) fetch(row + stride);
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = v[it][j] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
        T* yr = y + row * (long long)D;
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) o[j] = (v[it][j] - mu) * rs * g[it][j] + bt[it][j];
                store_vec<T, V>(yr + c, o);
            }
        }
    }
}

// each wave walks rows (grid-strided), keeps its columns' dgamma/dbeta partial sums in
// registers, and the block writes one partial row pair; a second kernel adds the partials in
// block order.
template <typename T, int V, int LN_MAX_ITERS>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, long long ldx,
                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const T* __restrict__ dres, T* __restrict__ dx,
                                     float* __restrict__ partial, long long rows, int D, int want_rsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float dg[LN_MAX_ITERS][V], db[LN_MAX_ITERS][V], gm[LN_MAX_ITERS][V], dr[LN_MAX_ITERS][V];
#pragma unroll
    for (int it = 0; it < LN_MAX_ITERS; ++it) {
        const int c = (it * 64 + lane) * V;
#pragma unroll
        for (int j = 0; j < V; ++j) dg[it][j] = db[it][j] = gm[it][j] = dr[it][j] = 0.f;
        if (c < D) load_vec<float, V>(gamma + c, gm[it]);
    }
    // The row loop runs one row ahead: the three loads of the next row (x, dy and the residual gradient) are in
    // flight while the current row is reduced and stored, so a wave always has a row's worth of reads outstanding.
    typedef Vec<T, V> __attribute__((aligned(sizeof(T) * V))) VT;
    VT nx[LN_MAX_ITERS], ndy[LN_MAX_ITERS], ndr[LN_MAX_ITERS];
    float nmu = 0.f, nrs = 0.f;
    const long long stride = (long long)gridDim.x * nw;
    auto fetch = [&](long long row) {
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
                nx[it] = *reinterpret_cast<const VT*>(x + row * ldx + c);
                ndy[it] = *reinterpret_cast<const VT*>(dy + row * (long long)D + c);
                if (dres) ndr[it] = *reinterpret_cast<const VT*>(dres + row * (long long)D + c);
            }
        }
        nmu = mean[row];
        nrs = rstd[row];
    };
    long long row = (long long)blockIdx.x * nw + wave;
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        VT cx[LN_MAX_ITERS], cdy[LN_MAX_ITERS], cdr[LN_MAX_ITERS];
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            cx[it] = nx[it];
            cdy[it] = ndy[it];
            cdr[it] = ndr[it];
        }
        const float mu = nmu, rs = nrs;
        if (row + stride < rows) fetch(row + stride);
        // the normalised input and the scaled gradient are recomputed from the raw row in the second pass rather
        // than held: the registers go to the row in flight
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float dv = to_f32(cdy[it].v[j]);
                    const float xh = (to_f32(cx[it].v[j]) - mu) * rs;
                    const float g = dv * gm[it][j];
                    s1 += g;
                    s2 += g * xh;
                    dg[it][j] += dv * xh;
                    db[it][j] += dv;
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
        T* dxr = dx + row * (long long)D;
#pragma unroll
        for (int it = 0; it < LN_MAX_ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (to_f32(cx[it].v[j]) - mu) * rs;
                    const float g = to_f32(cdy[it].v[j]) * gm[it][j];
                    o[j] = rs * (g - c1 - xh * c2);
                }
                if (dres) {
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const float r = to_f32(cdr[it].v[j]);
                        o[j] += r;
                        dr[it][j] += r;
                    }
                }
                store_vec<T, V>(dxr + c, o);
            }
        }
    }
    // combine the block's waves through LDS, one iteration slice at a time
    __shared__ float red[3][4][64 * V];
#pragma unroll
    for (int it = 0; it < LN_MAX_ITERS; ++it) {
        if (it * 64 * V >= D) break;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < V; ++j) {
            red[0][wave][lane * V + j] = dg[it][j];
            red[1][wave][lane * V + j] = db[it][j];
            red[2][wave][lane * V + j] = dr[it][j];
        }
        __syncthreads();
        if (wave == 0) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float a = 0.f, b = 0.f, r = 0.f;
                    for (int w = 0; w < nw; ++w) {
                        a += red[0][w][lane * V + j];
                        b += red[1][w][lane * V + j];
                        r += red[2][w][lane * V + j];
                    }
                    partial[((long long)blockIdx.x * 3 + 0) * D + c + j] = a;
                    partial[((long long)blockIdx.x * 3 + 1) * D + c + j] = b;
                    if (want_rsum) partial[((long long)blockIdx.x * 3 + 2) * D + c + j] = r;
                }
            }
        }
    }
}

// out[q][c] = sum_k partial[k][q][c] in a fixed order.  Block (column strip, quantity q): 16 row
// groups x 64 columns, each group keeps four independent running sums so that the loads of a
// thread overlap instead of forming one dependent chain.
__global__ __launch_bounds__(1024) void layernorm_bwd_final_kernel(const float* __restrict__ partial, int blocks,
                                                                   int D, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta,
                                                                   float* __restrict__ rsum) {
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int q = blockIdx.y;
    float* out = q == 0 ? dgamma : (q == 1 ? dbeta : rsum);
    if (!out) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < D) {
        const float* p = partial + (long long)q * D + c;
        const long long st = 3ll * D;
        int k = grp;
        for (; k + 48 < blocks; k += 64) {
            a0 += p[(long long)k * st];
            a1 += p[(long long)(k + 16) * st];
            a2 += p[(long long)(k + 32) * st];
            a3 += p[(long long)(k + 48) * st];
        }
        for (; k < blocks; k += 16) a0 += p[(long long)k * st];
    }
    __shared__ float red[16][64];
    red[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && c < D) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][lane];
        out[c] = s;
    }
}

// three 4-wave blocks per CU on 256 CUs (the backward kernel's registers allow three waves per SIMD), every wave
// with the loads of two rows outstanding: the kernel streams 3-4 row-sized tensors
inline int ln_bwd_blocks(long long rows) {
    long long b = (rows + 3) / 4;
    if (b > 768) b = 768;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- mean pooling
template <typename T>
__global__ void meanpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int N, int D) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * D) return;
    const int b = (int)(idx / D), c = (int)(idx % D);
    float s = 0.f;
    for (int i = 0; i < N; ++i) s += to_f32(x[((long long)b * N + i) * D + c]);
    out[idx] = from_f32<T>(s / (float)N);
}
template <typename T>
__global__ void meanpool_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int B, int N, int D) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * N * D) return;
    const int c = (int)(idx % D);
    const int b = (int)(idx / ((long long)N * D));
    dx[idx] = from_f32<T>(to_f32(dout[(long long)b * D + c]) / (float)N);
}

// ---------------------------------------------------------------- patch gather (im2col for k = s = p)
// out[(b, gy, gx)][(c, i, j)] = x[b, c, gy*p + i, gx*p + j]; one thread per output element, writes
// coalesced, reads in runs of p pixels
template <typename S, typename T>
__global__ void patchify_kernel(const S* __restrict__ x, T* __restrict__ out, int C, int H, int W, int p, int gw,
                                int per, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int E = C * p * p;
    const long long t = idx / E;
    const int e = (int)(idx - t * E);
    const int b = (int)(t / per), tt = (int)(t - (long long)b * per);
    const int gy = tt / gw, gx = tt - gy * gw;
    const int c = e / (p * p), q = e - c * p * p;
    const int i = q / p, j = q - i * p;
    out[idx] = from_f32<T>(to_f32(x[(((long long)b * C + c) * H + gy * p + i) * W + gx * p + j]));
}

// ---------------------------------------------------------------- transposed bf16 copies of many fp32 matrices
// dst[c][r] = (bf16) src[r][c] for every job in one launch: block t finds its job by bisection over the
// prefix of 64x64-tile counts, then moves one tile through LDS: 256-byte runs of the fp32 rows in, 128-byte runs of
// the bf16 rows out.
constexpr int TT = 64;
__global__ __launch_bounds__(256) void transpose_cast_multi_kernel(const vitmi_transpose_job* __restrict__ jobs,
                                                                  const long long* __restrict__ tile_start, int n_jobs) {
    __shared__ float tile[TT][TT + 1];
    const long long t = blockIdx.x;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {                   // last job whose first tile is <= t
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    const vitmi_transpose_job job = jobs[lo];
    const float* in = reinterpret_cast<const float*>(job.src);
    bf16_t* out = reinterpret_cast<bf16_t*>(job.dst);
    const int R = (int)job.rows, Cn = (int)job.cols;
    const int tiles_c = (Cn + TT - 1) / TT;
    const int local = (int)(t - tile_start[lo]);
    const int c0 = (local % tiles_c) * TT, r0 = (local / tiles_c) * TT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll 4
    for (int k = ty; k < TT; k += 4) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < Cn) ? in[(long long)r * Cn + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < TT; k += 4) {
        const int c = c0 + k, r = r0 + tx;
        if (c < Cn && r < R) out[(long long)c * R + r] = (bf16_t)tile[tx][k];
    }
}

// ---------------------------------------------------------------- positional-row order from the mask
// seq_ord[i] = i * (1 - m[i]) + m[i] * masked_row  (ViT_mask.py:162-163), int64 out, mask int64 or int32
template <typename M>
__global__ void seq_ord_kernel(const M* __restrict__ mask, long long* __restrict__ out, int n, long long masked_row) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long m = (long long)mask[i];
    out[i] = (long long)i * (1 - m) + m * masked_row;
}

// ---------------------------------------------------------------- patch-level image gather (jigsaw augmentation)
// out[b] patch j is a whole-patch copy chosen by src[j]: v in [0, L) -> patch v of x[b]; v in [L, 2L) -> patch v-L of
// bank image bank_idx[b]; -2 -> the per-sample patch fill[b]; anything else -> zeros.  Pure byte movement in units of
// U bytes (a unit never crosses a patch row), one thread per output unit, reads and writes in runs of one patch row.
template <typename U>
__global__ void jigsaw_gather_kernel(const unsigned char* __restrict__ x, unsigned char* __restrict__ out,
                                     const int* __restrict__ src, const unsigned char* __restrict__ bank, int bank_count,
                                     const int* __restrict__ bank_idx, const unsigned char* __restrict__ fill, int C, int H,
                                     int gw, int L, int prow /* bytes of one patch row */, int p, long long row_bytes,
                                     long long total_units) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total_units) return;
    const int upr = (int)(row_bytes / (int)sizeof(U));          // units per image row
    const long long r = idx / upr;                                // (b*C + c)*H + y
    const int xb = (int)(idx - r * upr) * (int)sizeof(U);        // byte offset inside the row
    const int y = (int)(r % H);
    const long long bc = r / H;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const int gx = xb / prow, within = xb - gx * prow;
    const int gy = y / p, iy = y - gy * p;
    const int v = src[gy * gw + gx];
    U val;
    memset(&val, 0, sizeof(U));
    const unsigned char* from = nullptr;
    if (v >= 0 && v < L) {
        const int sy = v / gw, sx = v - sy * gw;
        from = x + (((long long)b * C + c) * H + sy * p + iy) * row_bytes + (long long)sx * prow + within;
    } else if (v >= L && v < 2 * L && bank != nullptr) {
        const int bi = bank_idx[b];
        if (bi >= 0 && bi < bank_count) {
            const int sp = v - L, sy = sp / gw, sx = sp - sy * gw;
            from = bank + (((long long)bi * C + c) * H + sy * p + iy) * row_bytes + (long long)sx * prow + within;
        }
    } else if (v == -2 && fill != nullptr) {
        from = fill + (((long long)b * C + c) * p + iy) * prow + within;
    }
    if (from) val = *reinterpret_cast<const U*>(from);
    *reinterpret_cast<U*>(out + r * row_bytes + xb) = val;
}

// out[b, c, i, j] = (sum over patches t of weight[t] * x[b, c, gy(t)*p + i, gx(t)*p + j]) / L
template <typename T>
__global__ void patch_weighted_mean_kernel(const T* __restrict__ x, const float* __restrict__ weight, T* __restrict__ out,
                                           int C, int H, int W, int p, int gw, int L, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx % p);
    const int i = (int)((idx / p) % p);
    const long long bc = idx / (p * p);
    const T* img = x + bc * H * W;
    float acc = 0.f;
    for (int t = 0; t < L; ++t) {
        const float w = weight[t];
        if (w != 0.f) {
            const int gy = t / gw, gx = t - gy * gw;
            acc += w * to_f32(img[(long long)(gy * p + i) * W + gx * p + j]);
        }
    }
    out[idx] = from_f32<T>(acc / (float)L);
}

// ---------------------------------------------------------------- Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16_t* __restrict__ p16, long long n, float lr, float b1, float b2,
                            float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
        if (p16) p16[i] = (bf16_t)pi;
    }
}

__device__ __forceinline__ void adam_update(float& pi, float gi, float& mi, float& vi, float lr_c, float b1, float b2,
                                            float eps, float wd, float bc2_sqrt, float gscale) {
    gi *= gscale;
    if (wd != 0.f) gi += wd * pi;
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= lr_c * (mi / denom);
}

// one block per chunk of one tensor; 16-byte accesses on the aligned body
__global__ __launch_bounds__(256) void adam_multi_kernel(const vitmi_adam_tensor* __restrict__ tensors,
                                                         const vitmi_adam_chunk* __restrict__ chunks, int chunk_elems,
                                                         float lr_c, float b1, float b2, float eps, float wd,
                                                         float bc2_sqrt, float gscale) {
    const vitmi_adam_chunk ck = chunks[blockIdx.x];
    const vitmi_adam_tensor t = tensors[ck.tensor];
    const long long end = ck.first + chunk_elems < t.n ? ck.first + chunk_elems : t.n;
    float* __restrict__ p = t.param;
    const float* __restrict__ g = t.grad;
    float* __restrict__ m = t.exp_avg;
    float* __restrict__ v = t.exp_avg_sq;
    bf16_t* __restrict__ p16 = (bf16_t*)t.param_bf16;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(p16) & 7) == 0;
    auto scalar = [&](long long k) {
        float a = p[k], b = m[k], c = v[k];
        adam_update(a, g[k], b, c, lr_c, b1, b2, eps, wd, bc2_sqrt, gscale);
        p[k] = a;
        m[k] = b;
        v[k] = c;
        if (p16) p16[k] = (bf16_t)a;
    };
    if (!vec) {  // unaligned tensor
        for (long long k = ck.first + threadIdx.x; k < end; k += 256) scalar(k);
        return;
    }
    for (long long i = ck.first + 4ll * threadIdx.x; i + 4 <= end; i += 4 * 256) {
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + i);
        f32x4 vv = *reinterpret_cast<const f32x4*>(v + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = pv[k], b = mv[k], c = vv[k];
            adam_update(a, gv[k], b, c, lr_c, b1, b2, eps, wd, bc2_sqrt, gscale);
            pv[k] = a;
            mv[k] = b;
            vv[k] = c;
        }
        *reinterpret_cast<f32x4*>(p + i) = pv;
        *reinterpret_cast<f32x4*>(m + i) = mv;
        *reinterpret_cast<f32x4*>(v + i) = vv;
        if (p16) {
            bf16x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (bf16_t)pv[k];
            *reinterpret_cast<bf16x4*>(p16 + i) = o;
        }
    }
    // the last 1-3 elements of the tensor
    const long long k = ck.first + ((end - ck.first) & ~3ll) + threadIdx.x;
    if (k < end) scalar(k);
}

// The same update for weight matrices that keep a transposed bf16 copy (the operand of the input-gradient GEMMs):
// one block per 64 x 64 tile (found by bisection as in transpose_cast_multi_kernel) updates the tile 16 bytes per
// access, writes the row-major bf16 copy, and sends the rounded values through LDS to write the transposed copy in
// 8-byte pieces: the separate transpose pass and its re-read of every fp32 weight disappear.  cols % 4 == 0 and
// 16-byte-aligned fp32 pointers are the caller's side of the contract (checked by the entry point).
__global__ __launch_bounds__(256) void adam_tiles_kernel(const vitmi_adam_matrix* __restrict__ mats,
                                                         const long long* __restrict__ tile_start, int n_mats,
                                                         float lr_c, float b1, float b2, float eps, float wd,
                                                         float bc2_sqrt, float gscale) {
    __shared__ bf16_t tile[TT][TT + 2];
    const long long t = blockIdx.x;
    int lo = 0, hi = n_mats - 1;
    while (lo < hi) {                   // last matrix whose first tile is <= t
        const int mid = (lo + hi + 1) >> 1;
        if (tile_start[mid] <= t) lo = mid;
        else hi = mid - 1;
    }
    const vitmi_adam_matrix job = mats[lo];
    float* __restrict__ p = job.param;
    const float* __restrict__ g = job.grad;
    float* __restrict__ m = job.exp_avg;
    float* __restrict__ v = job.exp_avg_sq;
    bf16_t* __restrict__ p16 = (bf16_t*)job.param_bf16;
    bf16_t* __restrict__ p16t = (bf16_t*)job.param_bf16_t;
    const int R = (int)job.rows, Cn = (int)job.cols;
    const int tiles_c = (Cn + TT - 1) / TT;
    const int local = (int)(t - tile_start[lo]);
    const int c0 = (local % tiles_c) * TT, r0 = (local / tiles_c) * TT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16; a thread owns 4 consecutive columns
#pragma unroll
    for (int k = ty; k < TT; k += 16) {
        const int r = r0 + k, c = c0 + 4 * tx;
        if (r < R && c < Cn) {
            const long long i = (long long)r * Cn + c;
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 mv = *reinterpret_cast<const f32x4*>(m + i);
            f32x4 vv = *reinterpret_cast<const f32x4*>(v + i);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = pv[j], b = mv[j], cc = vv[j];
                adam_update(a, gv[j], b, cc, lr_c, b1, b2, eps, wd, bc2_sqrt, gscale);
                pv[j] = a;
                mv[j] = b;
                vv[j] = cc;
                o[j] = (bf16_t)a;
                tile[k][4 * tx + j] = o[j];
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(m + i) = mv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
            if (p16) *reinterpret_cast<bf16x4*>(p16 + i) = o;
        }
    }
    __syncthreads();
    const bool vec_t = (R & 3) == 0;
#pragma unroll
    for (int k = ty; k < TT; k += 16) {
        const int c = c0 + k, r = r0 + 4 * tx;
        if (c < Cn && r < R) {
            bf16_t* dst = p16t + (long long)c * R + r;
            if (vec_t) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = tile[4 * tx + j][k];
                *reinterpret_cast<bf16x4*>(dst) = o;
            } else {
                for (int j = 0; j < 4 && r + j < R; ++j) dst[j] = tile[4 * tx + j][k];
            }
        }
    }
}

}  // namespace

// ================================================================ C ABI
extern "C" int vitmi_cast(const void* src, int src_dt, void* dst, int dst_dt, long long n, void* stream) {
    VITMI_CHECK_ARG(n >= 0 && (n == 0 || (src && dst)), "vitmi_cast: null pointer or negative n");
    hipStream_t s = as_stream(stream);
    if (src_dt == VITMI_F32 && dst_dt == VITMI_BF16) return launch_cast<float, bf16_t>(src, dst, n, s);
    if (src_dt == VITMI_BF16 && dst_dt == VITMI_F32) return launch_cast<bf16_t, float>(src, dst, n, s);
    if (src_dt == VITMI_F32 && dst_dt == VITMI_F32) return launch_cast<float, float>(src, dst, n, s);
    if (src_dt == VITMI_BF16 && dst_dt == VITMI_BF16) return launch_cast<bf16_t, bf16_t>(src, dst, n, s);
    vitmi_set_error("vitmi_cast: unsupported dtype pair %d -> %d", src_dt, dst_dt);
    return VITMI_ERR_BAD_ARG;
}

extern "C" int vitmi_scale_add_clip(const void* x, const float* addend, void* y, int dt, long long n, float scale,
                                    float lo, float hi, void* stream) {
    VITMI_CHECK_ARG(x && y && n >= 0, "vitmi_scale_add_clip: bad argument");
    if (n == 0) return VITMI_OK;
    hipStream_t s = as_stream(stream);
    const int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL(scale_add_clip_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, addend, (float*)y,
                           n, scale, lo, hi);
    else if (dt == VITMI_BF16)
        hipLaunchKernelGGL(scale_add_clip_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, addend,
                           (bf16_t*)y, n, scale, lo, hi);
    else {
        vitmi_set_error("vitmi_scale_add_clip: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_scale_add_clip");
    return VITMI_OK;
}

extern "C" int vitmi_add(const void* a, const void* b, void* y, int dt, long long n, void* stream) {
    VITMI_CHECK_ARG(n >= 0 && (n == 0 || (a && b && y)), "vitmi_add: null pointer or negative n");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_add: bad dtype %d", dt);
    if (n == 0) return VITMI_OK;
    hipStream_t s = as_stream(stream);
    const bool vec = aligned_to(a, 16) && aligned_to(b, 16) && aligned_to(y, 16);
    const int threads = 256;
    if (dt == VITMI_F32) {
        if (vec) {
            int blocks = (int)std::min<long long>((n + threads * 4 - 1) / (threads * 4), 4096);
            hipLaunchKernelGGL((add_kernel<float, 4>), dim3(blocks), dim3(threads), 0, s, (const float*)a,
                               (const float*)b, (float*)y, n);
        } else {
            int blocks = (int)std::min<long long>((n + threads - 1) / threads, 4096);
            hipLaunchKernelGGL((add_kernel<float, 1>), dim3(blocks), dim3(threads), 0, s, (const float*)a,
                               (const float*)b, (float*)y, n);
        }
    } else {
        if (vec) {
            int blocks = (int)std::min<long long>((n + threads * 8 - 1) / (threads * 8), 4096);
            hipLaunchKernelGGL((add_kernel<bf16_t, 8>), dim3(blocks), dim3(threads), 0, s, (const bf16_t*)a,
                               (const bf16_t*)b, (bf16_t*)y, n);
        } else {
            int blocks = (int)std::min<long long>((n + threads - 1) / threads, 4096);
            hipLaunchKernelGGL((add_kernel<bf16_t, 1>), dim3(blocks), dim3(threads), 0, s, (const bf16_t*)a,
                               (const bf16_t*)b, (bf16_t*)y, n);
        }
    }
    VITMI_LAUNCH_CHECK("vitmi_add");
    return VITMI_OK;
}

extern "C" int vitmi_dropout(const void* x, void* y, int dt, long long n, float p, unsigned long long seed, void* stream) {
    VITMI_CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "vitmi_dropout: null pointer or negative n");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_dropout: bad dtype %d", dt);
    VITMI_CHECK_ARG(p >= 0.f && p < 1.f, "vitmi_dropout: p = %f outside [0, 1)", (double)p);
    if (n == 0) return VITMI_OK;
    hipStream_t s = as_stream(stream);
    const double t = (double)p * 4294967296.0;
    const unsigned threshold = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    const float scale = 1.0f / (1.0f - p);
    const int blocks = (int)std::min<long long>(((n + 1) / 2 + 255) / 256, 8192);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (float*)y, n, seed, threshold,
                           scale);
    else
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n, seed,
                           threshold, scale);
    VITMI_LAUNCH_CHECK("vitmi_dropout");
    return VITMI_OK;
}

extern "C" long long vitmi_colsum_workspace_bytes(long long rows, int cols) {
    return (long long)colsum_chunks(rows) * cols * (long long)sizeof(float);
}
long long vitmi_colsum_workspace_bytes_impl(long long rows, int cols) { return vitmi_colsum_workspace_bytes(rows, cols); }

// out[c] = sum over the `chunks` rows of a dense [chunks][cols] fp32 partial matrix, fixed order
int vitmi_colsum_final_launch(const float* partial, int chunks, int cols, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(cols, 64)), dim3(1024), 0, s, partial, chunks, cols, out);
    VITMI_LAUNCH_CHECK("colsum(final)");
    return VITMI_OK;
}

extern "C" int vitmi_colsum(const void* x, int dt, long long rows, int cols, long long ld, float* out, void* workspace,
                            void* stream) {
    VITMI_CHECK_ARG(x && out && workspace, "vitmi_colsum: null pointer");
    VITMI_CHECK_ARG(rows > 0 && cols > 0 && ld >= cols, "vitmi_colsum: bad shape rows=%lld cols=%d ld=%lld", rows, cols,
                    ld);
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_OTHER, s, (double)rows * cols * (dt == VITMI_F32 ? 4 : 2));
    if (dt == VITMI_F32) return launch_colsum<float>(x, rows, cols, ld, out, workspace, s);
    if (dt == VITMI_BF16) return launch_colsum<bf16_t>(x, rows, cols, ld, out, workspace, s);
    vitmi_set_error("vitmi_colsum: bad dtype %d", dt);
    return VITMI_ERR_BAD_ARG;
}

extern "C" int vitmi_tokens_assemble_fwd(const void* patch, const float* cls, const float* pe,
                                         const long long* seq_ord, void* out, int dt, int B, int N, int D,
                                         void* stream) {
    VITMI_CHECK_ARG(patch && cls && out, "vitmi_tokens_assemble_fwd: null pointer");
    VITMI_CHECK_ARG(B > 0 && N > 1 && D > 0, "vitmi_tokens_assemble_fwd: bad shape B=%d N=%d D=%d", B, N, D);
    hipStream_t s = as_stream(stream);
    const int esz = dt == VITMI_F32 ? 4 : 2;
    // algorithmic bytes: read the patch rows, write the token rows, read the table rows and the indices once
    ProfScope prof(VITMI_FAM_TOKENS, s,
                   (double)B * (N - 1) * D * esz + (double)B * N * D * esz + (double)N * D * 4 + 8.0 * N);
    const long long rows = (long long)B * N;
    const bool vec = (D % 4 == 0) && aligned_to(patch, 16) && aligned_to(out, 16) && aligned_to(cls, 16) &&
                     (!pe || aligned_to(pe, 16));
    dim3 grid(ceil_div(rows, 4)), block(256);
    if (dt == VITMI_F32) {
        if (vec)
            hipLaunchKernelGGL((tokens_assemble_fwd_kernel<float, 4>), grid, block, 0, s, (const float*)patch, cls, pe,
                               seq_ord, (float*)out, B, N, D);
        else
            hipLaunchKernelGGL((tokens_assemble_fwd_kernel<float, 1>), grid, block, 0, s, (const float*)patch, cls, pe,
                               seq_ord, (float*)out, B, N, D);
    } else if (dt == VITMI_BF16) {
        if (vec)
            hipLaunchKernelGGL((tokens_assemble_fwd_kernel<bf16_t, 4>), grid, block, 0, s, (const bf16_t*)patch, cls,
                               pe, seq_ord, (bf16_t*)out, B, N, D);
        else
            hipLaunchKernelGGL((tokens_assemble_fwd_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)patch, cls,
                               pe, seq_ord, (bf16_t*)out, B, N, D);
    } else {
        vitmi_set_error("vitmi_tokens_assemble_fwd: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_tokens_assemble_fwd");
    return VITMI_OK;
}

extern "C" long long vitmi_tokens_assemble_bwd_workspace_bytes(int N, int D) {
    // position sums [N*D] followed by the column-sum partials (at most 128 chunks of N*D floats)
    return ((long long)N * D + 128ll * N * D) * (long long)sizeof(float);
}

extern "C" int vitmi_tokens_assemble_bwd(const void* dx, const long long* seq_ord, float* dcls, float* dpe,
                                         int pe_rows, void* workspace, int dt, int B, int N, int D, void* stream) {
    VITMI_CHECK_ARG(dx && dcls && workspace, "vitmi_tokens_assemble_bwd: null pointer");
    VITMI_CHECK_ARG(B > 0 && N > 1 && D > 0 && pe_rows >= 0, "vitmi_tokens_assemble_bwd: bad shape");
    hipStream_t s = as_stream(stream);
    const int esz = dt == VITMI_F32 ? 4 : 2;
    ProfScope prof(VITMI_FAM_TOKENS, s, (double)B * N * D * esz + (double)(pe_rows + 1) * D * 4);
    float* possum = (float*)workspace;
    float* partial = possum + (long long)N * D;
    // sum over the batch: a column sum of the [B, N*D] view
    int rc;
    if (dt == VITMI_F32)
        rc = launch_colsum<float>(dx, B, N * D, (long long)N * D, possum, partial, s);
    else if (dt == VITMI_BF16)
        rc = launch_colsum<bf16_t>(dx, B, N * D, (long long)N * D, possum, partial, s);
    else {
        vitmi_set_error("vitmi_tokens_assemble_bwd: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL(tokens_scatter_rows_kernel, dim3(pe_rows + 1), dim3(256), 0, s, possum, seq_ord, dcls, dpe,
                       pe_rows, N, D);
    VITMI_LAUNCH_CHECK("vitmi_tokens_assemble_bwd");
    return VITMI_OK;
}

extern "C" int vitmi_layernorm_fwd(const void* x, long long ldx, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int dt, long long rows, int D, float eps, void* stream) {
    VITMI_CHECK_ARG(x && gamma && beta && y, "vitmi_layernorm_fwd: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && ldx >= D, "vitmi_layernorm_fwd: bad shape rows=%lld D=%d ldx=%lld", rows, D,
                    ldx);
    hipStream_t s = as_stream(stream);
    const int esz = dt == VITMI_F32 ? 4 : 2;
    ProfScope prof(VITMI_FAM_LAYERNORM, s, 2.0 * rows * D * esz);
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && aligned_to(x, 16) && aligned_to(y, 16) &&
                     aligned_to(gamma, 16) && aligned_to(beta, 16);
    const int iters = ceil_div(D, 64 * (vec ? 4 : 1));
    if (iters > 8) {
        vitmi_set_error("vitmi_layernorm_fwd: D=%d exceeds the supported row length %d", D, 64 * 8 * (vec ? 4 : 1));
        return VITMI_ERR_UNSUPPORTED;
    }
    if (dt != VITMI_F32 && dt != VITMI_BF16) {
        vitmi_set_error("vitmi_layernorm_fwd: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    // four 4-wave blocks per CU on 256 CUs, every wave walking rows with the next row's load outstanding
    dim3 grid((unsigned)std::min<long long>(ceil_div(rows, 4), 1024)), block(256);
#define LN_FWD(T, V, I)                                                                                         \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, V, I>), grid, block, 0, s, (const T*)x, ldx, gamma, beta, (T*)y, \
                       mean, rstd, rows, D, eps)
#define LN_FWD_I(T, V)                         \
    do {                                       \
        if (iters <= 1) LN_FWD(T, V, 1);       \
        else if (iters == 2) LN_FWD(T, V, 2);  \
        else if (iters == 3) LN_FWD(T, V, 3);  \
        else if (iters == 4) LN_FWD(T, V, 4);  \
        else if (iters <= 6) LN_FWD(T, V, 6);  \
        else LN_FWD(T, V, 8);                  \
    } while (0)
    if (dt == VITMI_F32) {
        if (vec) LN_FWD_I(float, 4); else LN_FWD_I(float, 1);
    } else {
        if (vec) LN_FWD_I(bf16_t, 4); else LN_FWD_I(bf16_t, 1);
    }
#undef LN_FWD_I
#undef LN_FWD
    VITMI_LAUNCH_CHECK("vitmi_layernorm_fwd");
    return VITMI_OK;
}

// mean and 1/sqrt(var + eps) of every row, nothing else: what a product with LayerNorm folded into its weight
// needs of the rows (VITMI_EPI_LN).  Same row walk as layernorm_fwd_kernel, one row ahead.
template <typename T, int V, int ITERS>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, long long ldx, float* __restrict__ mean,
                                                        float* __restrict__ rstd, long long rows, int D, float eps) {
    typedef Vec<T, V> __attribute__((aligned(sizeof(T) * V))) VT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long stride = (long long)gridDim.x * nw;
    VT nx[ITERS];
    auto fetch = [&](long long row) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) nx[it] = *reinterpret_cast<const VT*>(x + row * ldx + c);
        }
    };
    long long row = (long long)blockIdx.x * nw + wave;
    if (row < rows) fetch(row);
    for (; row < rows; row += stride) {
        float v[ITERS][V];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                v[it][j] = c < D ? to_f32(nx[it].v[j]) : 0.f;
                sum += v[it][j];
            }
        }
        if (row + stride < rows) fetch(row + stride);
        const float mu = wave_sum(sum) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * V;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float d = v[it][j] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// Folds LayerNorm's affine parameters into the weight of the Linear behind it (operands of vitmi_gemm_ln): one wave
// per weight row n; wg[n][k] = bf16(w[n][k] * gamma[k]), s[n] = sum_k float(wg[n][k]) (of the ROUNDED values: the
// epilogue's mean[m] * s[n] then cancels exactly what the product accumulated), c[n] = sum_k w[n][k] beta[k] + b[n].
__global__ void fold_ln_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ bias, bf16_t* __restrict__ wg,
                               float* __restrict__ s, float* __restrict__ c, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* row = w + (long long)n * K;
    float ss = 0.f, cc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = row[k];
        const bf16_t g = (bf16_t)(wv * gamma[k]);
        wg[(long long)n * K + k] = g;
        ss += (float)g;
        cc = fmaf(wv, beta[k], cc);
    }
    ss = wave_sum(ss);
    cc = wave_sum(cc);
    if (lane == 0) {
        s[n] = ss;
        c[n] = cc + (bias ? bias[n] : 0.f);
    }
}

extern "C" int vitmi_fold_ln(const float* w, const float* gamma, const float* beta, const float* bias, void* w_folded,
                             float* w_rowsum, float* bias_folded, int N, int K, void* stream) {
    VITMI_CHECK_ARG(w && gamma && beta && w_folded && w_rowsum && bias_folded, "vitmi_fold_ln: null pointer");
    VITMI_CHECK_ARG(N > 0 && K > 0, "vitmi_fold_ln: bad shape N=%d K=%d", N, K);
    hipStream_t st = as_stream(stream);
    ProfScope prof(VITMI_FAM_OTHER, st, 6.0 * N * K);
    hipLaunchKernelGGL(fold_ln_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, st, w, gamma, beta, bias, (bf16_t*)w_folded,
                       w_rowsum, bias_folded, N, K);
    VITMI_LAUNCH_CHECK("vitmi_fold_ln");
    return VITMI_OK;
}

extern "C" int vitmi_row_stats(const void* x, long long ldx, float* mean, float* rstd, int dt, long long rows, int D,
                               float eps, void* stream) {
    VITMI_CHECK_ARG(x && mean && rstd, "vitmi_row_stats: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && ldx >= D, "vitmi_row_stats: bad shape rows=%lld D=%d ldx=%lld", rows, D, ldx);
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_row_stats: bad dtype %d", dt);
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_LAYERNORM, s, 1.0 * rows * D * (dt == VITMI_F32 ? 4 : 2));
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && aligned_to(x, 16);
    const int iters = ceil_div(D, 64 * (vec ? 4 : 1));
    if (iters > 8) {
        vitmi_set_error("vitmi_row_stats: D=%d exceeds the supported row length %d", D, 64 * 8 * (vec ? 4 : 1));
        return VITMI_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)std::min<long long>(ceil_div(rows, 4), 1024)), block(256);
#define RS(T, V, I) hipLaunchKernelGGL((row_stats_kernel<T, V, I>), grid, block, 0, s, (const T*)x, ldx, mean, rstd, rows, D, eps)
#define RS_I(T, V)                         \
    do {                                   \
        if (iters <= 1) RS(T, V, 1);       \
        else if (iters == 2) RS(T, V, 2);  \
        else if (iters == 3) RS(T, V, 3);  \
        else if (iters == 4) RS(T, V, 4);  \
        else if (iters <= 6) RS(T, V, 6);  \
        else RS(T, V, 8);                  \
    } while (0)
    if (dt == VITMI_F32) {
        if (vec) RS_I(float, 4); else RS_I(float, 1);
    } else {
        if (vec) RS_I(bf16_t, 4); else RS_I(bf16_t, 1);
    }
#undef RS_I
#undef RS
    VITMI_LAUNCH_CHECK("vitmi_row_stats");
    return VITMI_OK;
}

extern "C" long long vitmi_layernorm_bwd_workspace_bytes(long long rows, int D) {
    return (long long)ln_bwd_blocks(rows) * 3 * D * (long long)sizeof(float);
}

// shared body of vitmi_layernorm_bwd and vitmi_layernorm_bwd_static: `partial` is ln_bwd_blocks(rows) * 3 * D floats
static int layernorm_bwd_launch(const char* who, const void* dy, const void* x, long long ldx, const float* gamma,
                                const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma,
                                float* dbeta, float* dres_colsum, float* partial, int dt, long long rows, int D,
                                hipStream_t s) {
    const int esz = dt == VITMI_F32 ? 4 : 2;
    ProfScope prof(VITMI_FAM_LAYERNORM, s, (dres ? 4.0 : 3.0) * rows * D * esz);
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && aligned_to(x, 16) && aligned_to(dy, 16) && aligned_to(dx, 16) &&
                     aligned_to(gamma, 16) && (!dres || aligned_to(dres, 16));
    const int iters = ceil_div(D, 64 * (vec ? 4 : 1));
    if (iters > 8) {
        vitmi_set_error("%s: D=%d exceeds the supported row length %d", who, D, 64 * 8 * (vec ? 4 : 1));
        return VITMI_ERR_UNSUPPORTED;
    }
    if (dt != VITMI_F32 && dt != VITMI_BF16) {
        vitmi_set_error("%s: bad dtype %d", who, dt);
        return VITMI_ERR_BAD_ARG;
    }
    const int blocks = ln_bwd_blocks(rows);
    dim3 grid(blocks), block(256);
#define LN_BWD(T, V, I)                                                                                        \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, V, I>), grid, block, 0, s, (const T*)dy, (const T*)x, ldx, gamma, \
                       mean, rstd, (const T*)dres, (T*)dx, partial, rows, D, dres_colsum ? 1 : 0)
#define LN_BWD_I(T, V)                         \
    do {                                       \
        if (iters <= 1) LN_BWD(T, V, 1);       \
        else if (iters == 2) LN_BWD(T, V, 2);  \
        else if (iters == 3) LN_BWD(T, V, 3);  \
        else if (iters == 4) LN_BWD(T, V, 4);  \
        else if (iters <= 6) LN_BWD(T, V, 6);  \
        else LN_BWD(T, V, 8);                  \
    } while (0)
    if (dt == VITMI_F32) {
        if (vec) LN_BWD_I(float, 4); else LN_BWD_I(float, 1);
    } else {
        if (vec) LN_BWD_I(bf16_t, 4); else LN_BWD_I(bf16_t, 1);
    }
#undef LN_BWD_I
#undef LN_BWD
    VITMI_LAUNCH_CHECK(who);
    if (dgamma || dbeta || dres_colsum) {
        hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3(ceil_div(D, 64), 3), dim3(1024), 0, s, partial, blocks, D,
                           dgamma, dbeta, dres_colsum);
        VITMI_LAUNCH_CHECK(who);
    }
    return VITMI_OK;
}

extern "C" int vitmi_layernorm_bwd(const void* dy, const void* x, long long ldx, const float* gamma, const float* mean,
                                   const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                   float* dres_colsum, void* workspace, int dt, long long rows, int D,
                                   void* stream) {
    VITMI_CHECK_ARG(!dres_colsum || dres, "vitmi_layernorm_bwd: dres_colsum needs dres");
    VITMI_CHECK_ARG(dy && x && gamma && mean && rstd && dx && workspace, "vitmi_layernorm_bwd: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && ldx >= D, "vitmi_layernorm_bwd: bad shape");
    return layernorm_bwd_launch("vitmi_layernorm_bwd", dy, x, ldx, gamma, mean, rstd, dres, dx, dgamma, dbeta,
                                dres_colsum, (float*)workspace, dt, rows, D, as_stream(stream));
}

// ---- vitmi_layernorm_bwd_static: the same launches with the per-block partial sums in a device array of the
// library instead of a caller workspace, so the call allocates nothing.  The array is sized for the largest grid
// (768 blocks) at D <= VITMI_LN_BWD_STATIC_MAX_D; every GPU the library runs on holds one copy until the process
// ends.  The array's name is only a host-side handle: its device address is looked up once per device with
// hipGetSymbolAddress (the device current at that moment) and that pointer is what the kernels receive.
static __device__ __align__(256) float g_ln_bwd_scratch[768 * 3 * VITMI_LN_BWD_STATIC_MAX_D];
static float* g_ln_bwd_scratch_ptr[64] = {nullptr};

static int ln_bwd_scratch(float** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vitmi_hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= 64) {
        vitmi_set_error("vitmi_layernorm_bwd_static: device index %d beyond the 64 the library tracks", dev);
        return VITMI_ERR_UNSUPPORTED;
    }
    float* p = __atomic_load_n(&g_ln_bwd_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        // two threads racing here look up the same address; neither allocates
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_ln_bwd_scratch));
        if (e != hipSuccess) return vitmi_hip_fail(e, "hipGetSymbolAddress(layernorm scratch)");
        if (!p) {
            vitmi_set_error("vitmi_layernorm_bwd_static: no device address for the scratch array");
            return VITMI_ERR_HIP;
        }
        __atomic_store_n(&g_ln_bwd_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
    *out = p;
    return VITMI_OK;
}

extern "C" int vitmi_layernorm_bwd_static(const void* dy, const void* x, long long ldx, const float* gamma,
                                          const float* mean, const float* rstd, const void* dres, void* dx,
                                          float* dgamma, float* dbeta, float* dres_colsum, int dt, long long rows,
                                          int D, void* stream) {
    VITMI_CHECK_ARG(!dres_colsum || dres, "vitmi_layernorm_bwd_static: dres_colsum needs dres");
    VITMI_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "vitmi_layernorm_bwd_static: null pointer");
    VITMI_CHECK_ARG(rows > 0 && D > 0 && ldx >= D, "vitmi_layernorm_bwd_static: bad shape");
    if (D > VITMI_LN_BWD_STATIC_MAX_D) {
        vitmi_set_error("vitmi_layernorm_bwd_static: D=%d exceeds the static scratch (rows of at most %d); use "
                        "vitmi_layernorm_bwd with a workspace", D, VITMI_LN_BWD_STATIC_MAX_D);
        return VITMI_ERR_UNSUPPORTED;
    }
    float* partial = nullptr;
    int rc = ln_bwd_scratch(&partial);
    if (rc != VITMI_OK) return rc;
    // every word of the scratch that the final kernel reads is written by the first kernel of this call, so nothing
    // has to be cleared; two calls in flight on one GPU (different streams) would share it and race
    return layernorm_bwd_launch("vitmi_layernorm_bwd_static", dy, x, ldx, gamma, mean, rstd, dres, dx, dgamma, dbeta,
                                dres_colsum, partial, dt, rows, D, as_stream(stream));
}

extern "C" int vitmi_meanpool_fwd(const void* x, void* out, int dt, int B, int N, int D, void* stream) {
    VITMI_CHECK_ARG(x && out && B > 0 && N > 0 && D > 0, "vitmi_meanpool_fwd: bad argument");
    hipStream_t s = as_stream(stream);
    dim3 grid(ceil_div((long long)B * D, 256)), block(256);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL(meanpool_fwd_kernel<float>, grid, block, 0, s, (const float*)x, (float*)out, B, N, D);
    else if (dt == VITMI_BF16)
        hipLaunchKernelGGL(meanpool_fwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)out, B, N, D);
    else {
        vitmi_set_error("vitmi_meanpool_fwd: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_meanpool_fwd");
    return VITMI_OK;
}

extern "C" int vitmi_meanpool_bwd(const void* dout, void* dx, int dt, int B, int N, int D, void* stream) {
    VITMI_CHECK_ARG(dout && dx && B > 0 && N > 0 && D > 0, "vitmi_meanpool_bwd: bad argument");
    hipStream_t s = as_stream(stream);
    dim3 grid(ceil_div((long long)B * N * D, 256)), block(256);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL(meanpool_bwd_kernel<float>, grid, block, 0, s, (const float*)dout, (float*)dx, B, N, D);
    else if (dt == VITMI_BF16)
        hipLaunchKernelGGL(meanpool_bwd_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)dout, (bf16_t*)dx, B, N, D);
    else {
        vitmi_set_error("vitmi_meanpool_bwd: bad dtype %d", dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_meanpool_bwd");
    return VITMI_OK;
}

extern "C" int vitmi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                               long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int step, float grad_scale, void* stream) {
    VITMI_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "vitmi_adam_step: bad argument");
    if (n == 0) return VITMI_OK;
    hipStream_t s = as_stream(stream);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                       (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    VITMI_LAUNCH_CHECK("vitmi_adam_step");
    return VITMI_OK;
}

extern "C" int vitmi_adam_step_multi(const vitmi_adam_tensor* tensors, int n_tensors, const vitmi_adam_chunk* chunks,
                                     int n_chunks, int chunk_elems, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int step, float grad_scale, void* stream) {
    VITMI_CHECK_ARG(n_tensors >= 0 && n_chunks >= 0 && step >= 1, "vitmi_adam_step_multi: bad argument");
    VITMI_CHECK_ARG(chunk_elems > 0 && chunk_elems % 4 == 0, "vitmi_adam_step_multi: chunk_elems must be a positive multiple of 4");
    if (n_chunks == 0) return VITMI_OK;
    VITMI_CHECK_ARG(tensors && chunks && n_tensors > 0, "vitmi_adam_step_multi: null table");
    hipStream_t s = as_stream(stream);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, s, tensors, chunks, chunk_elems, lr / bc1, beta1,
                       beta2, eps, weight_decay, bc2_sqrt, grad_scale);
    VITMI_LAUNCH_CHECK("vitmi_adam_step_multi");
    return VITMI_OK;
}

extern "C" int vitmi_adam_step_matrices(const vitmi_adam_matrix* mats, const long long* tile_start, int n_mats,
                                        long long total_tiles, float lr, float beta1, float beta2, float eps,
                                        float weight_decay, int step, float grad_scale, void* stream) {
    VITMI_CHECK_ARG(n_mats >= 0 && total_tiles >= 0 && step >= 1, "vitmi_adam_step_matrices: bad argument");
    if (n_mats == 0 || total_tiles == 0) return VITMI_OK;
    VITMI_CHECK_ARG(mats && tile_start && total_tiles <= 0x7fffffffll, "vitmi_adam_step_matrices: bad table");
    hipStream_t s = as_stream(stream);
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_tiles_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, mats, tile_start, n_mats,
                       lr / bc1, beta1, beta2, eps, weight_decay, bc2_sqrt, grad_scale);
    VITMI_LAUNCH_CHECK("vitmi_adam_step_matrices");
    return VITMI_OK;
}

extern "C" int vitmi_patchify(const void* x, int x_dt, void* out, int out_dt, int B, int C, int H, int W, int p,
                              void* stream) {
    VITMI_CHECK_ARG(x && out, "vitmi_patchify: null pointer");
    VITMI_CHECK_ARG(B > 0 && C > 0 && p > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0,
                    "vitmi_patchify: bad geometry B=%d C=%d H=%d W=%d p=%d", B, C, H, W, p);
    hipStream_t s = as_stream(stream);
    const int gw = W / p, per = (H / p) * gw;
    const long long total = (long long)B * per * C * p * p;
    ProfScope prof(VITMI_FAM_OTHER, s, (double)total * ((x_dt == VITMI_F32 ? 4 : 2) + (out_dt == VITMI_F32 ? 4 : 2)));
    dim3 grid(ceil_div(total, 256)), block(256);
#define PATCHIFY(S, T) \
    hipLaunchKernelGGL((patchify_kernel<S, T>), grid, block, 0, s, (const S*)x, (T*)out, C, H, W, p, gw, per, total)
    if (x_dt == VITMI_F32 && out_dt == VITMI_F32) PATCHIFY(float, float);
    else if (x_dt == VITMI_F32 && out_dt == VITMI_BF16) PATCHIFY(float, bf16_t);
    else if (x_dt == VITMI_BF16 && out_dt == VITMI_BF16) PATCHIFY(bf16_t, bf16_t);
    else if (x_dt == VITMI_BF16 && out_dt == VITMI_F32) PATCHIFY(bf16_t, float);
    else {
        vitmi_set_error("vitmi_patchify: bad dtype pair %d -> %d", x_dt, out_dt);
        return VITMI_ERR_BAD_ARG;
    }
#undef PATCHIFY
    VITMI_LAUNCH_CHECK("vitmi_patchify");
    return VITMI_OK;
}

extern "C" int vitmi_jigsaw_gather(const void* x, void* out, int dt, int B, int C, int H, int W, int p, const int* src,
                                   const void* bank, int bank_count, const int* bank_idx, const void* fill, void* stream) {
    VITMI_CHECK_ARG(x && out && src, "vitmi_jigsaw_gather: null pointer");
    VITMI_CHECK_ARG(x != out, "vitmi_jigsaw_gather: the gather cannot run in place");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_jigsaw_gather: bad dtype %d", dt);
    VITMI_CHECK_ARG(B > 0 && C > 0 && p > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0,
                    "vitmi_jigsaw_gather: bad geometry B=%d C=%d H=%d W=%d p=%d", B, C, H, W, p);
    VITMI_CHECK_ARG(bank == nullptr || (bank_count > 0 && bank_idx != nullptr),
                    "vitmi_jigsaw_gather: a bank needs bank_count > 0 and bank_idx");
    hipStream_t s = as_stream(stream);
    const int es = dt == VITMI_F32 ? 4 : 2;
    const int gw = W / p, L = (H / p) * gw, prow = p * es;
    const long long row_bytes = (long long)W * es;
    const unsigned long long align = (unsigned long long)(uintptr_t)x | (uintptr_t)out | (uintptr_t)bank | (uintptr_t)fill |
                                     (unsigned long long)prow;
    int unit = 16;
    while (unit > es && (align & (unsigned long long)(unit - 1))) unit >>= 1;
    const long long total = (long long)B * C * H * (row_bytes / unit);
    ProfScope prof(VITMI_FAM_OTHER, s, 2.0 * B * C * H * (double)row_bytes);
    dim3 grid(ceil_div(total, 256)), block(256);
#define JIGSAW(U)                                                                                                      \
    hipLaunchKernelGGL((jigsaw_gather_kernel<U>), grid, block, 0, s, (const unsigned char*)x, (unsigned char*)out, src,   \
                       (const unsigned char*)bank, bank_count, bank_idx, (const unsigned char*)fill, C, H, gw, L, prow, p, \
                       row_bytes, total)
    if (unit == 16) JIGSAW(uint4);
    else if (unit == 8) JIGSAW(uint2);
    else if (unit == 4) JIGSAW(unsigned int);
    else JIGSAW(unsigned short);
#undef JIGSAW
    VITMI_LAUNCH_CHECK("vitmi_jigsaw_gather");
    return VITMI_OK;
}

extern "C" int vitmi_patch_weighted_mean(const void* x, int dt, int B, int C, int H, int W, int p, const float* weight,
                                         void* out, void* stream) {
    VITMI_CHECK_ARG(x && out && weight, "vitmi_patch_weighted_mean: null pointer");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_patch_weighted_mean: bad dtype %d", dt);
    VITMI_CHECK_ARG(B > 0 && C > 0 && p > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0,
                    "vitmi_patch_weighted_mean: bad geometry B=%d C=%d H=%d W=%d p=%d", B, C, H, W, p);
    hipStream_t s = as_stream(stream);
    const int gw = W / p, L = (H / p) * gw;
    const long long total = (long long)B * C * p * p;
    ProfScope prof(VITMI_FAM_OTHER, s, (double)B * C * H * W * (dt == VITMI_F32 ? 4 : 2));
    dim3 grid(ceil_div(total, 256)), block(256);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL((patch_weighted_mean_kernel<float>), grid, block, 0, s, (const float*)x, weight, (float*)out, C, H,
                           W, p, gw, L, total);
    else
        hipLaunchKernelGGL((patch_weighted_mean_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, weight, (bf16_t*)out,
                           C, H, W, p, gw, L, total);
    VITMI_LAUNCH_CHECK("vitmi_patch_weighted_mean");
    return VITMI_OK;
}

extern "C" int vitmi_seq_ord(const void* unk_mask, int mask_is_int64, long long* seq_ord, int n, long long masked_row,
                             void* stream) {
    VITMI_CHECK_ARG(unk_mask && seq_ord, "vitmi_seq_ord: null pointer");
    VITMI_CHECK_ARG(n > 0, "vitmi_seq_ord: bad length %d", n);
    hipStream_t s = as_stream(stream);
    if (mask_is_int64)
        hipLaunchKernelGGL((seq_ord_kernel<long long>), dim3(ceil_div(n, 256)), dim3(256), 0, s, (const long long*)unk_mask,
                           seq_ord, n, masked_row);
    else
        hipLaunchKernelGGL((seq_ord_kernel<int>), dim3(ceil_div(n, 256)), dim3(256), 0, s, (const int*)unk_mask, seq_ord, n,
                           masked_row);
    VITMI_LAUNCH_CHECK("vitmi_seq_ord");
    return VITMI_OK;
}

extern "C" int vitmi_transpose_cast_multi(const vitmi_transpose_job* jobs, const long long* tile_start, int n_jobs,
                                          long long total_tiles, void* stream) {
    if (n_jobs == 0) return VITMI_OK;
    VITMI_CHECK_ARG(jobs && tile_start && n_jobs > 0 && total_tiles > 0 && total_tiles < (1ll << 31),
                    "vitmi_transpose_cast_multi: bad table");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(transpose_cast_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, jobs, tile_start, n_jobs);
    VITMI_LAUNCH_CHECK("vitmi_transpose_cast_multi");
    return VITMI_OK;
}
