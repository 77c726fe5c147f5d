// Attention rollout on the device (vit_rollout.py:18-61, trainer.py:17-44 of the reference).
//
// The reference fuses heads, zeroes the k smallest entries of each sample's fused [N,N] matrix
// (topk + scatter_), forms (A + I)/2 with rows normalised, multiplies the L matrices together and
// keeps row 0.  Here each layer is ONE launch (a block per sample: head fusion, radix selection of
// the k-th smallest entry, discard, row normalisation; the matrix is read once from the exported
// probabilities and kept in fp32), and the product is ONE launch that carries only row 0 through
// the chain: e0^T a_{L-1} ... a_0 is L vector-matrix products (L*N^2 multiply-adds per sample) instead
// of L matrix-matrix products (L*N^3).  All of it is bound by reading the probabilities from HBM.
#include "common.h"

namespace {

constexpr int RL_THREADS = 1024;

// fusion 0 mean, 1 max, 2 min over the heads; 3 = mean over the heads of probability x gradient with negative means
// set to zero (vit_grad_rollout.py:14-16), g pointing at the gradient entry that matches p
template <typename T>
__device__ __forceinline__ float fuse_heads(const T* __restrict__ p, long long head_stride, int H, int fusion,
                                            const T* __restrict__ g = nullptr) {
    if (fusion == 3) {
        float f = 0.f;
        for (int h = 0; h < H; ++h) f += to_f32(p[h * head_stride]) * to_f32(g[h * head_stride]);
        f = f / (float)H;
        return f > 0.f ? f : 0.f;
    }
    float f = to_f32(p[0]);
    if (fusion == 0) {
        for (int h = 1; h < H; ++h) f += to_f32(p[h * head_stride]);
        return f / (float)H;
    }
    if (fusion == 1) {
        for (int h = 1; h < H; ++h) f = fmaxf(f, to_f32(p[h * head_stride]));
        return f;
    }
    for (int h = 1; h < H; ++h) f = fminf(f, to_f32(p[h * head_stride]));
    return f;
}

// 16 bytes of T that are only element-aligned in memory
template <typename T>
struct __attribute__((packed, aligned(sizeof(T)))) Run {
    T v[16 / sizeof(T)];
};

constexpr int RL_MAXL = 48;        // layers per launch (pointer table passed by value)
struct LayerPtrs {
    const void* p[RL_MAXL];
};

// hist[bin] += 1 for every lane with `on`.  Fused probabilities share their top bits, so most lanes of a wave hit
// the same bin: the lanes that agree with the first active lane add their count once, the others add on their own.
__device__ __forceinline__ void hist_add(unsigned int* hist, unsigned bin, bool on) {
    if (on) {
        const unsigned lbin = (unsigned)__builtin_amdgcn_readfirstlane((int)bin);
        const unsigned long long same = __ballot(bin == lbin);
        if (bin == lbin) {
            if ((int)(threadIdx.x & (WAVE - 1)) == __ffsll((long long)same) - 1)
                atomicAdd(&hist[lbin], (unsigned)__popcll(same));
        } else {
            atomicAdd(&hist[bin], 1u);
        }
    }
}

// a[l][b] = row-normalised ((discard_k(fuse_h probs[l][b,h])) + I) / 2.  One block per (sample, layer).
// Entries are non-negative floats, so their bit patterns order like the values and the k-th smallest is
// found by four 8-bit histogram rounds.  Among entries equal to the k-th value the lowest flat indices are
// discarded first.  `first` = 1 keeps flat entry 0 out of the candidates (the trainer.py variant).
constexpr int RL_REGVALS = 40;     // fused values a thread may hold in registers (REG path: N*N <= ~40 K, N <= 202)

template <typename T, bool REG>
__global__ __launch_bounds__(RL_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void rollout_layer_kernel(LayerPtrs probs, LayerPtrs grads, int l0, float* __restrict__ a_all, int B,
                                                                  int H, int N, int fusion, int k, int first,
                                                                  int norm_mode) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int sel[4];      // prefix, mask, remaining, count of the chosen bin
    __shared__ unsigned int wave_tot[RL_THREADS / WAVE];
    __shared__ unsigned int running;
    const int tid = threadIdx.x, b = blockIdx.x, l = blockIdx.y;
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int NN = N * N;
    const T* p = reinterpret_cast<const T*>(probs.p[l]) + (long long)b * H * NN;
    const T* gp = fusion == 3 ? reinterpret_cast<const T*>(grads.p[l]) + (long long)b * H * NN : nullptr;
    float* a = a_all + ((long long)(l0 + l) * B + b) * NN;

    // Head fusion reads 16 bytes per lane and head; N*N is odd for the usual token counts, so these runs are only
    // element-aligned (the hardware splits such loads, which costs far less than 8 separate 2-byte loads).
    // REG: thread t keeps the runs g = r*1024 + t (elements g*V .. g*V+V-1) and one element of the ragged tail in
    // registers through selection and discard, and the matrix is written once.  Otherwise it lives in a[] (L2).
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int R = RL_REGVALS / V;
    const int nvec = NN / V;
    float f[R][V];
    float ftail = 0.f;
    auto fuse_run = [&](int g, float* out) {
        Run<T> run;
        if (fusion == 3) {
            Run<T> grun;
#pragma unroll
            for (int e = 0; e < V; ++e) out[e] = 0.f;
            for (int h = 0; h < H; ++h) {
                __builtin_memcpy(&run, p + (long long)h * NN + (long long)g * V, sizeof(run));
                __builtin_memcpy(&grun, gp + (long long)h * NN + (long long)g * V, sizeof(grun));
#pragma unroll
                for (int e = 0; e < V; ++e) out[e] += to_f32(run.v[e]) * to_f32(grun.v[e]);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float m = out[e] / (float)H;
                out[e] = m > 0.f ? m : 0.f;
            }
            return;
        }
        __builtin_memcpy(&run, p + (long long)g * V, sizeof(run));
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = to_f32(run.v[e]);
        for (int h = 1; h < H; ++h) {
            __builtin_memcpy(&run, p + (long long)h * NN + (long long)g * V, sizeof(run));
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float x = to_f32(run.v[e]);
                out[e] = fusion == 0 ? out[e] + x : (fusion == 1 ? fmaxf(out[e], x) : fminf(out[e], x));
            }
        }
        if (fusion == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) out[e] = out[e] / (float)H;
        }
    };
    auto store_run = [&](int g, const float* in) {
        Run<float> res[V / 4];
#pragma unroll
        for (int e = 0; e < V; ++e) res[e / 4].v[e % 4] = in[e];
        __builtin_memcpy(a + (long long)g * V, res, sizeof(res));
    };
    if constexpr (REG) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int g = r * RL_THREADS + tid;
            if (g < nvec) {
                fuse_run(g, f[r]);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) f[r][e] = 0.f;
            }
        }
        if (nvec * V + tid < NN) ftail = fuse_heads(p + nvec * V + tid, NN, H, fusion, gp ? gp + nvec * V + tid : nullptr);
    } else {
        for (int g = tid; g < nvec; g += RL_THREADS) {
            float t[V];
            fuse_run(g, t);
            store_run(g, t);
        }
        for (int idx = nvec * V + tid; idx < NN; idx += RL_THREADS)
            a[idx] = fuse_heads(p + idx, NN, H, fusion, gp ? gp + idx : nullptr);
        __syncthreads();
    }
    // fn(idx, bits, live) for every entry, called by whole waves (it votes)
    auto for_each = [&](auto&& fn) {
        if constexpr (REG) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int g = r * RL_THREADS + tid;
#pragma unroll
                for (int e = 0; e < V; ++e) fn(g * V + e, __float_as_uint(f[r][e]), g < nvec);
            }
            fn(nvec * V + tid, __float_as_uint(ftail), nvec * V + tid < NN);
        } else {
            for (int base = 0; base < NN; base += RL_THREADS) {
                const int idx = base + tid;
                fn(idx, idx < NN ? __float_as_uint(a[idx]) : 0u, idx < NN);
            }
        }
    };

    const int cand = NN - first;
    const int kk = k < cand ? k : cand;
    bool tie_pass = false;
    unsigned tbits = 0u, take_equal = 0u;
    if (kk > 0) {
        if (tid == 0) {
            sel[0] = 0u;
            sel[1] = 0u;
            sel[2] = (unsigned)kk;
            sel[3] = (unsigned)cand;
        }
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += RL_THREADS) hist[i] = 0u;
            __syncthreads();
            const unsigned prefix = sel[0], mask = sel[1];
            for_each([&](int idx, unsigned bits, bool live) {
                hist_add(hist, (bits >> shift) & 255u, live && idx >= first && (bits & mask) == prefix);
            });
            __syncthreads();
            if (wave == 0) {             // lane q owns bins 4q..4q+3; a shuffle scan finds the bin that holds rank `rem`
                const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2],
                               c3 = hist[4 * lane + 3];
                const unsigned own = c0 + c1 + c2 + c3;
                unsigned incl = own;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const unsigned up = (unsigned)__shfl_up((int)incl, o, WAVE);
                    if (lane >= o) incl += up;
                }
                const unsigned excl = incl - own, rem = sel[2];
                if (excl < rem && rem <= incl) {
                    unsigned cum = excl;
                    int q = 0;
                    unsigned c = c0;
                    if (cum + c0 >= rem) { q = 0; c = c0; }
                    else if (cum + c0 + c1 >= rem) { q = 1; c = c1; cum += c0; }
                    else if (cum + c0 + c1 + c2 >= rem) { q = 2; c = c2; cum += c0 + c1; }
                    else { q = 3; c = c3; cum += c0 + c1 + c2; }
                    sel[0] = prefix | ((unsigned)(4 * lane + q) << shift);
                    sel[1] = mask | (255u << shift);
                    sel[2] = rem - cum;
                    sel[3] = c;
                }
            }
            __syncthreads();
        }
        tbits = sel[0];
        take_equal = sel[2];
        tie_pass = take_equal < sel[3];  // fewer entries equal to the threshold go than there are: index order decides
    }
    if constexpr (REG) {
        if (kk > 0 && !tie_pass) {       // every entry up to the threshold goes
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if ((r * RL_THREADS + tid) * V + e >= first && __float_as_uint(f[r][e]) <= tbits) f[r][e] = 0.f;
            }
            if (__float_as_uint(ftail) <= tbits) ftail = 0.f;      // the tail never holds flat entry 0 (N*N >= V)
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int g = r * RL_THREADS + tid;
            if (g < nvec) store_run(g, f[r]);
        }
        if (nvec * V + tid < NN) a[nvec * V + tid] = ftail;
        __syncthreads();
    } else if (kk > 0 && !tie_pass) {
        for (int idx = first + tid; idx < NN; idx += RL_THREADS)
            if (__float_as_uint(a[idx]) <= tbits) a[idx] = 0.f;
        __syncthreads();
    }
    if (tie_pass) {                      // discard below the threshold, and ties in index order, in a[]
        if (tid == 0) running = 0u;
        __syncthreads();
        for (int base = 0; base < NN; base += RL_THREADS) {
            const int idx = base + tid;
            const bool live = idx >= first && idx < NN;
            const unsigned bits = live ? __float_as_uint(a[idx]) : 0xffffffffu;
            const bool eq = live && bits == tbits;
            const unsigned long long bal = __ballot(eq);
            const unsigned before = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned off = running;
            for (int w = 0; w < wave; ++w) off += wave_tot[w];
            if (live && (bits < tbits || (eq && off + before < take_equal))) a[idx] = 0.f;
            __syncthreads();
            if (tid == 0) {
                unsigned t = 0u;
                for (int w = 0; w < RL_THREADS / WAVE; ++w) t += wave_tot[w];
                running += t;
            }
            __syncthreads();
        }
    }

    // (A + I) / 2, each row divided by its sum: a wave per row
    if (norm_mode == 0) {
        for (int i = wave; i < N; i += RL_THREADS / WAVE) {
            float* row = a + (long long)i * N;
            float s = 0.f;
            for (int j = lane; j < N; j += WAVE) s += (row[j] + (i == j ? 1.f : 0.f)) * 0.5f;
            s = wave_sum(s);
            for (int j = lane; j < N; j += WAVE) row[j] = (row[j] + (i == j ? 1.f : 0.f)) * 0.5f / s;
        }
        return;
    }
    // norm_mode 1, the arithmetic of vit_grad_rollout.py:27-28 on a batch of one: `a / a.sum(dim=-1)` without keepdim
    // broadcasts the vector of row sums along the LAST axis, so entry (i, j) is divided by the sum of row j
    __shared__ float rowsum[256];
    for (int i = wave; i < N; i += RL_THREADS / WAVE) {
        const float* row = a + (long long)i * N;
        float s = 0.f;
        for (int j = lane; j < N; j += WAVE) s += (row[j] + (i == j ? 1.f : 0.f)) * 0.5f;
        s = wave_sum(s);
        if (lane == 0) rowsum[i] = s;
    }
    __syncthreads();
    for (int idx = tid; idx < NN; idx += RL_THREADS) {
        const int i = idx / N, j = idx - i * N;
        a[idx] = (a[idx] + (i == j ? 1.f : 0.f)) * 0.5f / rowsum[j];
    }
}

// Row 0 of a_{L-1} ... a_1 a_0 per sample, then the class-to-patch part [1:] in one of two final forms:
// final_mode 0 = as is (trainer.py:44), 1 = divided by its maximum, then softmax (vit_rollout.py:51-55), 2 = divided by
// its maximum (vit_grad_rollout.py:35).
// v <- v a_l is split over CH_GROUPS groups of 256 threads (each a slice of the rows, CH_UNROLL running
// sums per thread so that as many loads are on their way) and the partial vectors are added in a fixed order.
constexpr int CH_GROUPS = 4;
constexpr int CH_THREADS = 256 * CH_GROUPS;
constexpr int CH_UNROLL = 8;         // rows of a layer's matrix a thread has on their way at a time
__global__ __launch_bounds__(CH_THREADS) void rollout_chain_kernel(const float* __restrict__ a_all, float* __restrict__ out,
                                                                  int L, int B, int N, int final_mode) {
    extern __shared__ float vbuf[];      // v [N], w [N], partial [CH_GROUPS][N]
    __shared__ float red[CH_THREADS / WAVE];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int g = tid >> 8, t = tid & 255;
    float* v = vbuf;
    float* w = vbuf + N;
    float* part = vbuf + 2 * N;
    for (int j = tid; j < N; j += CH_THREADS) v[j] = j == 0 ? 1.f : 0.f;
    __syncthreads();
    const int per = (N + CH_GROUPS - 1) / CH_GROUPS;
    const int i0 = g * per, i1 = (i0 + per) < N ? (i0 + per) : N;
    for (int l = L - 1; l >= 0; --l) {
        const float* A = a_all + ((long long)l * B + b) * N * N;
        for (int j = t; j < N; j += 256) {
            float sacc[CH_UNROLL];
#pragma unroll
            for (int u = 0; u < CH_UNROLL; ++u) sacc[u] = 0.f;
            int i = i0;
            for (; i + CH_UNROLL <= i1; i += CH_UNROLL) {
                float av[CH_UNROLL];
#pragma unroll
                for (int u = 0; u < CH_UNROLL; ++u) av[u] = A[(long long)(i + u) * N + j];
#pragma unroll
                for (int u = 0; u < CH_UNROLL; ++u) sacc[u] = fmaf(v[i + u], av[u], sacc[u]);
            }
            for (; i < i1; ++i) sacc[0] = fmaf(v[i], A[(long long)i * N + j], sacc[0]);
#pragma unroll
            for (int w2 = CH_UNROLL / 2; w2 > 0; w2 >>= 1)
#pragma unroll
                for (int u = 0; u < w2; ++u) sacc[u] += sacc[u + w2];
            part[g * N + j] = sacc[0];
        }
        __syncthreads();
        for (int j = tid; j < N; j += CH_THREADS) {
            float acc = part[j];
#pragma unroll
            for (int q = 1; q < CH_GROUPS; ++q) acc += part[q * N + j];
            w[j] = acc;
        }
        __syncthreads();
        float* sw = v;
        v = w;
        w = sw;
    }
    float* o = out + (long long)b * (N - 1);
    if (final_mode == 0) {
        for (int j = 1 + tid; j < N; j += CH_THREADS) o[j - 1] = v[j];
        return;
    }
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    constexpr int NW = CH_THREADS / WAVE;
    float m = -INFINITY;
    for (int j = 1 + tid; j < N; j += CH_THREADS) m = fmaxf(m, v[j]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int q = 1; q < NW; ++q) m = fmaxf(m, red[q]);
    __syncthreads();
    if (final_mode == 2) {
        for (int j = 1 + tid; j < N; j += CH_THREADS) o[j - 1] = v[j] / m;
        return;
    }
    // softmax of t = v / m, whose maximum is 1 (m > 0 for any rollout of probabilities)
    float tmax = -INFINITY;
    for (int j = 1 + tid; j < N; j += CH_THREADS) tmax = fmaxf(tmax, v[j] / m);
    tmax = wave_max(tmax);
    if (lane == 0) red[wave] = tmax;
    __syncthreads();
    tmax = red[0];
    for (int q = 1; q < NW; ++q) tmax = fmaxf(tmax, red[q]);
    __syncthreads();
    float s = 0.f;
    for (int j = 1 + tid; j < N; j += CH_THREADS) {
        const float e = expf(v[j] / m - tmax);
        w[j] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = 0.f;
    for (int q = 0; q < NW; ++q) s += red[q];
    for (int j = 1 + tid; j < N; j += CH_THREADS) o[j - 1] = w[j] / s;
}

// out[b][j-1] = fuse_h probs[b, h, 0, j], j = 1..N-1 (vit_rollout.py:64-73)
template <typename T>
__global__ void rollout_last_attn_kernel(const T* __restrict__ probs, float* __restrict__ out, int B, int H, int N,
                                         int fusion) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * (N - 1)) return;
    const int b = idx / (N - 1), j = 1 + idx - b * (N - 1);
    out[idx] = fuse_heads(probs + (long long)b * H * N * N + j, (long long)N * N, H, fusion);
}

}  // namespace

extern "C" long long vitmi_rollout_workspace_bytes(int L, int B, int N) {
    if (L <= 0 || B <= 0 || N <= 0) return 0;
    return (long long)L * B * N * N * (long long)sizeof(float);
}

static int rollout_impl(const char* who, const void* const* probs, const void* const* grads, int dt, int L, int B, int H,
                        int N, int head_fusion, int discard_k, int keep_first, int norm_mode, int final_mode, float* out,
                        void* workspace, void* stream) {
    VITMI_CHECK_ARG(probs && out && workspace, "%s: null pointer", who);
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "%s: bad dtype %d", who, dt);
    VITMI_CHECK_ARG(L > 0 && B > 0 && H > 0 && N > 1, "%s: bad geometry L=%d B=%d H=%d N=%d", who, L, B, H, N);
    VITMI_CHECK_ARG(N <= 256, "%s: N=%d exceeds the supported 256 tokens", who, N);
    VITMI_CHECK_ARG(discard_k >= 0 && discard_k <= N * N, "%s: discard_k %d outside [0, N*N]", who, discard_k);
    for (int l = 0; l < L; ++l)
        VITMI_CHECK_ARG(probs[l] != nullptr && (!grads || grads[l] != nullptr), "%s: tensor %d is null", who, l);
    hipStream_t s = as_stream(stream);
    const int k = discard_k;
    float* a_all = (float*)workspace;
    const int es = dt == VITMI_F32 ? 4 : 2;
    ProfScope prof(VITMI_FAM_OTHER, s, (double)L * B * N * N * ((double)H * es * (grads ? 2 : 1) + 12.0));
    for (int l0 = 0; l0 < L; l0 += RL_MAXL) {
        const int nl = (L - l0) < RL_MAXL ? (L - l0) : RL_MAXL;
        LayerPtrs tab, gtab;
        for (int l = 0; l < RL_MAXL; ++l) {
            tab.p[l] = l < nl ? probs[l0 + l] : nullptr;
            gtab.p[l] = (grads && l < nl) ? grads[l0 + l] : nullptr;
        }
#define RL_LAUNCH(T, REG)                                                                                             \
    hipLaunchKernelGGL((rollout_layer_kernel<T, REG>), dim3(B, nl), dim3(RL_THREADS), 0, s, tab, gtab, l0, a_all, B, H, \
                       N, head_fusion, k, keep_first ? 1 : 0, norm_mode)
        // registers hold the fused matrix when its 16-byte runs fit 40 values per thread (N <= 202) and there are
        // at least two runs per thread (below that most lanes would only vote)
        if (dt == VITMI_F32) {
            const int nvec = (N * N) / 4;
            if (nvec >= 2 * RL_THREADS && nvec <= (RL_REGVALS / 4) * RL_THREADS) RL_LAUNCH(float, true);
            else RL_LAUNCH(float, false);
        } else {
            const int nvec = (N * N) / 8;
            if (nvec >= 2 * RL_THREADS && nvec <= (RL_REGVALS / 8) * RL_THREADS) RL_LAUNCH(bf16_t, true);
            else RL_LAUNCH(bf16_t, false);
        }
#undef RL_LAUNCH
    }
    hipLaunchKernelGGL(rollout_chain_kernel, dim3(B), dim3(CH_THREADS), (2 + CH_GROUPS) * N * sizeof(float), s, a_all, out,
                       L, B, N, final_mode);
    VITMI_LAUNCH_CHECK(who);
    return VITMI_OK;
}

extern "C" int vitmi_rollout(const void* const* probs, int dt, int L, int B, int H, int N, int head_fusion,
                             int discard_k, int keep_first, int final_mode, float* out, void* workspace, void* stream) {
    VITMI_CHECK_ARG(head_fusion >= 0 && head_fusion <= 2, "vitmi_rollout: head_fusion %d (0 mean, 1 max, 2 min)",
                    head_fusion);
    VITMI_CHECK_ARG(final_mode == 0 || final_mode == 1, "vitmi_rollout: final_mode %d (0 raw, 1 max+softmax)", final_mode);
    return rollout_impl("vitmi_rollout", probs, nullptr, dt, L, B, H, N, head_fusion, discard_k, keep_first, 0, final_mode,
                        out, workspace, stream);
}

extern "C" int vitmi_grad_rollout(const void* const* probs, const void* const* grads, int dt, int L, int B, int H, int N,
                                  int discard_k, float* out, void* workspace, void* stream) {
    VITMI_CHECK_ARG(grads, "vitmi_grad_rollout: null pointer");
    return rollout_impl("vitmi_grad_rollout", probs, grads, dt, L, B, H, N, 3, discard_k, 0, 1, 2, out, workspace, stream);
}

extern "C" int vitmi_rollout_last_attn(const void* probs, int dt, int B, int H, int N, int head_fusion, float* out,
                                       void* stream) {
    VITMI_CHECK_ARG(probs && out, "vitmi_rollout_last_attn: null pointer");
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "vitmi_rollout_last_attn: bad dtype %d", dt);
    VITMI_CHECK_ARG(B > 0 && H > 0 && N > 1, "vitmi_rollout_last_attn: bad geometry B=%d H=%d N=%d", B, H, N);
    VITMI_CHECK_ARG(head_fusion >= 0 && head_fusion <= 2, "vitmi_rollout_last_attn: head_fusion %d", head_fusion);
    hipStream_t s = as_stream(stream);
    const int total = B * (N - 1);
    if (dt == VITMI_F32)
        hipLaunchKernelGGL((rollout_last_attn_kernel<float>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                           (const float*)probs, out, B, H, N, head_fusion);
    else
        hipLaunchKernelGGL((rollout_last_attn_kernel<bf16_t>), dim3(ceil_div(total, 256)), dim3(256), 0, s,
                           (const bf16_t*)probs, out, B, H, N, head_fusion);
    VITMI_LAUNCH_CHECK("vitmi_rollout_last_attn");
    return VITMI_OK;
}
