// Encoder sub-blocks at the small geometry as single launches (bf16, embedding width <= 256: the 32x32-pixel / patch-4
// models of config/cifar10 with the UMAP.py:12-17 overrides -- 65 tokens of width 192, MLP 512, 12 heads of 8).
//
// At that size every product of the sub-block is a few microseconds of arithmetic and the step is a chain of
// ~30 dependent launches per encoder block, each paying its own launch, its own fill and drain of a 256-CU grid and a
// round trip of its operands through HBM.  Here one workgroup owns a strip of token rows (MLP) or one image
// (attention) for the WHOLE sub-block: the rows are normalised into LDS, every product reads its activation operand
// from LDS and streams its weight operand straight from L2 into MFMA fragments (the weights of a block are 0.5 MB: they
// stay L2-resident for all workgroups), and only what the backward pass needs is written out.
//
// The shared device helpers (operand convention, streamed products, strip LayerNorm) are in small.h; this file holds
// the MLP sub-block, attention.hip the attention sub-block (it reuses that file's softmax tile).
#include "small.h"

namespace {

struct FfSmallArgs {
    const bf16_t* x;
    const float *gamma, *beta;
    const bf16_t* w1;
    const float* b1;
    const bf16_t* w2;
    const float* b2;
    bf16_t *xn, *u, *h, *y;
    float *mean, *rstd;
    long long rows;
    int D, M;
    float eps;
    int erf_pair;  // 1: u = pre-activation, h = erf GELU; 0: u = GELU'(pre-activation) as IEEE half, h = fast GELU
};

// y = x + W2 gelu(W1 LN(x) + b1) + b2 for a strip of 16 MT rows (mymodel/ViT.py:52-58, 62-73, 120), keeping LN(x), its
// statistics, the GELU output and what the backward epilogue multiplies by.
template <int MT>
__global__ __launch_bounds__(SM_THREADS) void ff_small_fwd_kernel(FfSmallArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int D = a.D, M = a.M, ldx = D + SM_PAD, ldh = M + SM_PAD;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* hs = xs + MT * 16 * ldx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row0 = (long long)blockIdx.x * (MT * 16);
    const int nrows = (int)(a.rows - row0 < MT * 16 ? a.rows - row0 : MT * 16);
    strip_layernorm<MT>(a.x, a.gamma, a.beta, a.xn, a.mean, a.rstd, row0, nrows, D, a.eps, xs, ldx);
    __syncthreads();
    const int tok = lane & 15, f8 = (lane >> 4) * 8;
    const int rot = blockIdx.x * 5;
    // hidden = gelu(xn W1^T + b1): 32 features per wave and pass, 8 consecutive ones per lane
    for (int pi = wave; pi < M / 32; pi += SM_WAVES) {
        const int p = (pi + rot) % (M / 32);  // workgroups walk the weight rows in different orders
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        rows_gemm<MT, 2, 3>(a.w1, D, p * 32, D / 32, xs, ldx, lane, acc);
        const int n = p * 32 + f8;
        const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.b1 + n), bias1 = *reinterpret_cast<const f32x4*>(a.b1 + n + 4);
        float pre[MT][8];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 8; ++t) pre[m][t] = acc[t >> 2][m][t & 3] + (t < 4 ? bias0[t & 3] : bias1[t & 3]);
        bf16x8 hq[MT], uq[MT];
        // one branch per pass, not per element: the MT * 8 evaluations below are independent chains
        if (a.erf_pair) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    uq[m][t] = (bf16_t)pre[m][t];
                    hq[m][t] = (bf16_t)gelu_erf_fast(pre[m][t]);
                }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    float gv, dv;
                    gelu_both_fast(pre[m][t], gv, dv);
                    hq[m][t] = (bf16_t)gv;
                    uq[m][t] = deriv_pack(dv, bf16_t());
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            *reinterpret_cast<bf16x8*>(hs + r * ldh + n) = hq[m];
            if (r < nrows) {
                *reinterpret_cast<bf16x8*>(a.h + (row0 + r) * M + n) = hq[m];
                *reinterpret_cast<bf16x8*>(a.u + (row0 + r) * M + n) = uq[m];
            }
        }
    }
    __syncthreads();
    // y = hidden W2^T + b2 + x
    for (int pi = wave; pi < D / 32; pi += SM_WAVES) {
        const int p = (pi + rot) % (D / 32);
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = p * 32 + f8;
        bf16x8 res[MT];  // the residual pieces, requested ahead of the products
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok < nrows ? m * 16 + tok : 0;
            res[m] = *reinterpret_cast<const bf16x8*>(a.x + (row0 + r) * D + n);
        }
        rows_gemm<MT, 2, 4>(a.w2, M, p * 32, M / 32, hs, ldh, lane, acc);
        const f32x4 bias0 = *reinterpret_cast<const f32x4*>(a.b2 + n), bias1 = *reinterpret_cast<const f32x4*>(a.b2 + n + 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            if (r < nrows) {
                bf16x8 o;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    o[t] = (bf16_t)((acc[t >> 2][m][t & 3] + (t < 4 ? bias0[t & 3] : bias1[t & 3])) + (float)res[m][t]);
                *reinterpret_cast<bf16x8*>(a.y + (row0 + r) * D + n) = o;
            }
        }
    }
}

// ---- backward of the same sub-block (mymodel/ViT.py:52-58, 62-73, 120) for a strip of 16 MT rows ----------------------
//   du  = (dy W2) * gelu'(pre-activation)        -> written out (operand of dW1 = du^T LN(x)) and kept in LDS
//   dxn = du W1                                  -> stays in registers (fp32)
//   dx  = LayerNorm backward of dxn at x, + dy   (the residual branch)
// and the strip's column sums for db1 (of du), dgamma, dbeta (of dxn) and db2 (of dy) as one row of `partial`
// ([strips][M + 3 D]); ff_small_bwd_final_kernel adds the rows in strip order.  W2 and W1 are read through their
// transposed copies ([M][D] and [D][M]: the output feature of each product indexes a row).
struct FfSmallBwdArgs {
    const bf16_t *dy, *x, *u;
    const float *gamma, *mean, *rstd;
    const bf16_t *w2_t, *w1_t;
    bf16_t *du, *dx;
    float* partial;
    long long rows;
    int D, M;
    int erf_pair;  // what u holds: 1 = the pre-activation, 0 = GELU'(pre-activation) as IEEE half
};

template <int MT>
__global__ __launch_bounds__(SM_THREADS) void ff_small_bwd_kernel(FfSmallBwdArgs a) {
    constexpr int R = MT * 16;
    extern __shared__ __align__(16) unsigned char smem[];
    const int D = a.D, M = a.M, ldx = D + SM_PAD, ldh = M + SM_PAD;
    bf16_t* dys = reinterpret_cast<bf16_t*>(smem);
    bf16_t* dus = dys + R * ldx;
    float* red = reinterpret_cast<float*>(dus + R * ldh);  // [2][SM_WAVES][R] row sums of the LayerNorm backward
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row0 = (long long)blockIdx.x * R;
    const int nrows = (int)(a.rows - row0 < R ? a.rows - row0 : R);
    const int tok = lane & 15, f8 = (lane >> 4) * 8;
    float* part = a.partial + (long long)blockIdx.x * (M + 3 * D);
    // the strip of dy into LDS (zeros past the end of the matrix)
    for (int i = threadIdx.x; i < R * (D / 8); i += SM_THREADS) {
        const int r = i / (D / 8), c = (i - r * (D / 8)) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < nrows) v = *reinterpret_cast<const u32x4*>(a.dy + (row0 + r) * D + c);
        *reinterpret_cast<u32x4*>(dys + r * ldx + c) = v;
    }
    __syncthreads();
    for (int p = wave; p < M / 32; p += SM_WAVES) {
        const int n = p * 32 + f8;
        bf16x8 uq[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok < nrows ? m * 16 + tok : 0;
            uq[m] = *reinterpret_cast<const bf16x8*>(a.u + (row0 + r) * M + n);
        }
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        rows_gemm<MT, 2, 3>(a.w2_t, D, p * 32, D / 32, dys, ldx, lane, acc);
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 dq[MT];
        if (a.erf_pair) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 8; ++t) dq[m][t] = (bf16_t)(acc[t >> 2][m][t & 3] * gelu_grad_fast((float)uq[m][t]));
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < 8; ++t) dq[m][t] = (bf16_t)(acc[t >> 2][m][t & 3] * deriv_unpack(uq[m][t]));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
#pragma unroll
            for (int t = 0; t < 8; ++t) cs[t] += (float)dq[m][t];  // rows past the end: dy = 0, so 0
            *reinterpret_cast<bf16x8*>(dus + r * ldh + n) = dq[m];
            if (r < nrows) *reinterpret_cast<bf16x8*>(a.du + (row0 + r) * M + n) = dq[m];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int t = 0; t < 8; ++t) cs[t] += __shfl_xor(cs[t], o, 64);
        if (tok == 0) {
            *reinterpret_cast<f32x4*>(part + n) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *reinterpret_cast<f32x4*>(part + n + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
        }
    }
    __syncthreads();
    // dxn: one 32-feature pass per wave (D <= 256), then the LayerNorm backward across the waves
    const bool active = wave < D / 32;
    const int n = wave * 32 + f8;
    f32x4 acc[2][MT];
    bf16x8 xq[MT];
    float mu[MT], rs[MT];
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok < nrows ? m * 16 + tok : 0;
            xq[m] = *reinterpret_cast<const bf16x8*>(a.x + (row0 + r) * D + n);
            mu[m] = a.mean[row0 + r];
            rs[m] = a.rstd[row0 + r];
        }
        g0 = *reinterpret_cast<const f32x4*>(a.gamma + n);
        g1 = *reinterpret_cast<const f32x4*>(a.gamma + n + 4);
        rows_gemm<MT, 2, 4>(a.w1_t, M, wave * 32, M / 32, dus, ldh, lane, acc);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float xh = ((float)xq[m][t] - mu[m]) * rs[m];
                const float g = acc[t >> 2][m][t & 3] * (t < 4 ? g0[t & 3] : g1[t & 3]);
                s1 += g;
                s2 += g * xh;
            }
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 16) {
                red[wave * R + m * 16 + tok] = s1;
                red[(SM_WAVES + wave) * R + m * 16 + tok] = s2;
            }
        }
    }
    __syncthreads();
    if (active) {
        float dg[8], db[8], dr[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) dg[t] = db[t] = dr[t] = 0.f;
        const int nw = D / 32;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            float c1 = 0.f, c2 = 0.f;
            for (int w = 0; w < nw; ++w) {
                c1 += red[w * R + r];
                c2 += red[(SM_WAVES + w) * R + r];
            }
            c1 /= (float)D;
            c2 /= (float)D;
            const bf16x8 dyv = *reinterpret_cast<const bf16x8*>(dys + r * ldx + n);
            bf16x8 o;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float xh = ((float)xq[m][t] - mu[m]) * rs[m];
                const float dxn = acc[t >> 2][m][t & 3];
                const float g = dxn * (t < 4 ? g0[t & 3] : g1[t & 3]);
                const float res = (float)dyv[t];
                o[t] = (bf16_t)(rs[m] * (g - c1 - xh * c2) + res);
                dg[t] += dxn * xh;  // rows past the end: du = 0, so dxn = 0
                db[t] += dxn;
                dr[t] += res;
            }
            if (r < nrows) *reinterpret_cast<bf16x8*>(a.dx + (row0 + r) * D + n) = o;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                dg[t] += __shfl_xor(dg[t], o, 64);
                db[t] += __shfl_xor(db[t], o, 64);
                dr[t] += __shfl_xor(dr[t], o, 64);
            }
        if (tok == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                part[M + n + t] = dg[t];
                part[M + D + n + t] = db[t];
                part[M + 2 * D + n + t] = dr[t];
            }
        }
    }
}

// out[c] = sum over the strips' rows of partial[strip][c], in strip order; c runs over [db1 (M) | dgamma | dbeta | db2 (D each)]
__device__ __forceinline__ void sums_final_block(const float* __restrict__ partial, int strips, int M, int D, float* __restrict__ db1,
                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db2,
                                                 int block) {
    // 16 row groups x 64 columns; a group keeps four independent running sums so that its loads overlap (the sums
    // are taken in a fixed order either way)
    const int cols = M + 3 * D;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = block * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        const float* p = partial + c;
        int k = grp;
        for (; k + 48 < strips; k += 64) {
            a0 += p[(long long)k * cols];
            a1 += p[(long long)(k + 16) * cols];
            a2 += p[(long long)(k + 32) * cols];
            a3 += p[(long long)(k + 48) * cols];
        }
        for (; k < strips; k += 16) a0 += p[(long long)k * cols];
    }
    __shared__ float red[16][64];
    red[grp][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (grp == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][lane];
        if (c < M)
            db1[c] = s;
        else if (c < M + D)
            dgamma[c - M] = s;
        else if (c < M + 2 * D)
            dbeta[c - M - D] = s;
        else if (db2)
            db2[c - M - 2 * D] = s;
    }
}
__global__ __launch_bounds__(1024) void ff_small_bwd_final_kernel(const float* __restrict__ partial, int strips, int M, int D,
                                                                  float* __restrict__ db1, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ db2) {
    sums_final_block(partial, strips, M, D, db1, dgamma, dbeta, db2, blockIdx.x);
}
// the same for up to SUMS_JOBS sub-blocks: blockIdx.y = job (table by value: capture-safe), blockIdx.x = 64 columns
constexpr int SUMS_JOBS = 32;
struct SumsTable {
    vitmi_sums_job j[SUMS_JOBS];
};
__global__ __launch_bounds__(1024) void small_sums_multi_kernel(SumsTable t) {
    const vitmi_sums_job& j = t.j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.M + 3 * j.D) return;   // whole block: no barrier is skipped by part of one
    sums_final_block(j.partial, j.strips, j.M, j.D, j.db1, j.dgamma, j.dbeta, j.db2, blockIdx.x);
}

// token tiles per workgroup: the fewest grid rounds over the CUs, then the smaller strip
int pick_strip_tiles(long long rows, int cus, size_t lds_per_tile) {
    int best = 2;
    long long best_cost = -1;
    for (int mt = 2; mt <= 6; ++mt) {
        if (mt * lds_per_tile > 160 * 1024) break;
        const long long wgs = (rows + mt * 16 - 1) / (mt * 16);
        const long long cost = ((wgs + cus - 1) / cus) * (mt + 2);  // a round costs a fixed part plus a part per token tile
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = mt;
        }
    }
    return best;
}

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// Raise a kernel's dynamic-LDS limit to the whole 160 KB once per device (bit d of *done_mask = done on device d).
int allow_full_lds(const void* fn, unsigned long long* done_mask, const char* who) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vitmi_hip_fail(e, who);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return VITMI_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return vitmi_hip_fail(e, who);
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return VITMI_OK;
}

template <int MT>
int launch_ff_small(const FfSmallArgs& a, hipStream_t s) {
    const size_t lds = (size_t)MT * 16 * ((a.D + SM_PAD) + (a.M + SM_PAD)) * sizeof(bf16_t);
    static unsigned long long done = 0;
    const int rc = allow_full_lds(reinterpret_cast<const void*>(&ff_small_fwd_kernel<MT>), &done, "ff_small_fwd: LDS attribute");
    if (rc != VITMI_OK) return rc;
    const int grid = (int)((a.rows + MT * 16 - 1) / (MT * 16));
    hipLaunchKernelGGL(ff_small_fwd_kernel<MT>, dim3(grid), dim3(SM_THREADS), lds, s, a);
    VITMI_LAUNCH_CHECK("ff_small_fwd_kernel");
    return VITMI_OK;
}

// (the backward kernel is opt-in: beside the weight gradients on the second stream it is no faster than the four
// launches it replaces -- 3.29 -> 3.46 ms per step at the 32-pixel geometry, batch 256, graph replay; it saves host time
// when the step is issued eagerly: 4.57 -> 4.45 ms)
constexpr int SMALL_DEFAULT = 3 | 8 | 16;
template <int MT>
int launch_ff_small_bwd(const FfSmallBwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)MT * 16 * ((a.D + SM_PAD) + (a.M + SM_PAD)) * sizeof(bf16_t) + (size_t)2 * SM_WAVES * MT * 16 * sizeof(float);
    static unsigned long long done = 0;
    const int rc = allow_full_lds(reinterpret_cast<const void*>(&ff_small_bwd_kernel<MT>), &done, "ff_small_bwd: LDS attribute");
    if (rc != VITMI_OK) return rc;
    const int grid = (int)((a.rows + MT * 16 - 1) / (MT * 16));
    hipLaunchKernelGGL(ff_small_bwd_kernel<MT>, dim3(grid), dim3(SM_THREADS), lds, s, a);
    VITMI_LAUNCH_CHECK("ff_small_bwd_kernel");
    return VITMI_OK;
}

// the strip height of the backward kernel: a function of the shape alone (the partial-sum buffer is sized from it)
int ff_small_bwd_tiles(long long rows, int D, int M) {
    int best = 2;
    long long best_cost = -1;
    for (int mt = 2; mt <= 6; ++mt) {
        const size_t lds = (size_t)mt * 16 * ((D + SM_PAD) + (M + SM_PAD)) * sizeof(bf16_t) + (size_t)2 * SM_WAVES * mt * 16 * sizeof(float);
        if (lds > 160 * 1024) break;
        const long long wgs = (rows + mt * 16 - 1) / (mt * 16);
        const long long cost = ((wgs + 255) / 256) * (mt + 2);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = mt;
        }
    }
    return best;
}

// bit 0: MLP sub-block forward, bit 1: attention sub-block forward, bit 2: MLP sub-block backward, bit 3: the same
// backward kernel in calls that leave the weight gradients to the caller (nothing then runs beside it), bit 4: attention
// sub-block backward in such calls
int g_small_fused = SMALL_DEFAULT;

}  // namespace

int vitmi_small_fused_mask() { return g_small_fused; }
extern "C" int vitmi_debug_set_small_fused(int mask) {
    g_small_fused = mask < 0 ? SMALL_DEFAULT : mask;  // negative: back to the default
    return VITMI_OK;
}

// returned by the one-launch forms when the operands do not meet their alignment: nothing was enqueued
enum { VITMI_SMALL_NOT_TAKEN = 1000 };

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// whether vitmi_ff_block_fwd runs as the one-launch kernel: bf16, width a multiple of 32 up to 256, hidden width a
// multiple of 32 whose strip fits LDS beside the normalised rows
bool vitmi_ff_small_ok(int dt, long long rows, int D, int M) {
    if (!(g_small_fused & 1) || dt != VITMI_BF16 || rows <= 0) return false;
    if (D < 32 || D > 256 || D % 32 != 0 || M < 32 || M % 32 != 0) return false;
    return (size_t)2 * 16 * ((D + SM_PAD) + (M + SM_PAD)) * sizeof(bf16_t) <= 160 * 1024;
}

int vitmi_ff_small_fwd(long long rows, int D, int M, const void* x, const float* gamma, const float* beta, float eps,
                       const void* w1, const float* b1, const void* w2, const float* b2, void* xn, float* mean, float* rstd,
                       void* u, void* h, void* y, int erf_pair, void* stream) {
    if (!(aligned16(x) && aligned16(w1) && aligned16(w2) && aligned16(xn) && aligned16(u) && aligned16(h) && aligned16(y) &&
          aligned16(gamma) && aligned16(beta) && aligned16(b1) && aligned16(b2)))
        return VITMI_SMALL_NOT_TAKEN;  // the caller runs the launch-per-step form
    FfSmallArgs a{(const bf16_t*)x, gamma, beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (bf16_t*)xn, (bf16_t*)u,
                  (bf16_t*)h, (bf16_t*)y, mean, rstd, rows, D, M, eps, erf_pair};
    hipStream_t s = as_stream(stream);
    int mt = pick_strip_tiles(rows, device_cus(), (size_t)16 * ((D + SM_PAD) + (M + SM_PAD)) * sizeof(bf16_t));
    if ((g_small_fused >> 16) & 7) mt = (g_small_fused >> 16) & 7;  // benchmarking override
    switch (mt) {
        case 2: return launch_ff_small<2>(a, s);
        case 3: return launch_ff_small<3>(a, s);
        case 4: return launch_ff_small<4>(a, s);
        case 5: return launch_ff_small<5>(a, s);
        default: return launch_ff_small<6>(a, s);
    }
}

// ---- MLP sub-block backward
bool vitmi_ff_small_bwd_ok(int dt, long long rows, int D, int M, bool caller_forms_dw) {
    if (!(g_small_fused & (caller_forms_dw ? 12 : 4)) || dt != VITMI_BF16 || rows <= 0) return false;
    if (D < 32 || D > 256 || D % 32 != 0 || M < 32 || M % 32 != 0) return false;
    return (size_t)2 * 16 * ((D + SM_PAD) + (M + SM_PAD)) * sizeof(bf16_t) + (size_t)2 * SM_WAVES * 32 * sizeof(float) <= 160 * 1024;
}
// bytes of the per-strip partial sums the backward kernel leaves for its final reduction
long long vitmi_ff_small_bwd_partial_bytes(long long rows, int D, int M) {
    if (rows <= 0 || D <= 0 || M <= 0) return 0;
    const int mt = ff_small_bwd_tiles(rows, D, M);
    return ((rows + mt * 16 - 1) / (mt * 16)) * (long long)(M + 3 * D) * (long long)sizeof(float);
}
// the fused kernel on `stream`; the final reduction of the bias / LayerNorm-parameter gradients on `final_stream`
// (which must already be ordered behind `stream`'s kernel when it differs).  VITMI_SMALL_NOT_TAKEN: nothing enqueued.
int vitmi_ff_small_bwd(long long rows, int D, int M, const void* dy, const void* x, const void* u, const float* gamma,
                       const float* mean, const float* rstd, const void* w2_t, const void* w1_t, void* du, void* dx,
                       float* partial, int erf_pair, void* stream) {
    if (!(aligned16(dy) && aligned16(x) && aligned16(u) && aligned16(gamma) && aligned16(w2_t) && aligned16(w1_t) && aligned16(du) &&
          aligned16(dx) && aligned16(partial)))
        return VITMI_SMALL_NOT_TAKEN;
    FfSmallBwdArgs a{(const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)u, gamma, mean, rstd, (const bf16_t*)w2_t,
                     (const bf16_t*)w1_t, (bf16_t*)du, (bf16_t*)dx, partial, rows, D, M, erf_pair};
    hipStream_t s = as_stream(stream);
    switch (ff_small_bwd_tiles(rows, D, M)) {
        case 2: return launch_ff_small_bwd<2>(a, s);
        case 3: return launch_ff_small_bwd<3>(a, s);
        case 4: return launch_ff_small_bwd<4>(a, s);
        case 5: return launch_ff_small_bwd<5>(a, s);
        default: return launch_ff_small_bwd<6>(a, s);
    }
}
int vitmi_ff_small_bwd_final(long long rows, int D, int M, const float* partial, float* db1, float* dgamma, float* dbeta,
                             float* db2, void* stream) {
    const int mt = ff_small_bwd_tiles(rows, D, M);
    const int strips = (int)((rows + mt * 16 - 1) / (mt * 16));
    hipLaunchKernelGGL(ff_small_bwd_final_kernel, dim3((M + 3 * D + 63) / 64), dim3(1024), 0, as_stream(stream), partial, strips, M,
                       D, db1, dgamma, dbeta, db2);
    VITMI_LAUNCH_CHECK("ff_small_bwd_final_kernel");
    return VITMI_OK;
}
// the same final reduction over `strips` rows of [dgamma | dbeta | db (D each)]: the attention sub-block's partial sums
// (db == null: no output bias)
int vitmi_small_sums_final(int strips, int D, const float* partial, float* dgamma, float* dbeta, float* db, void* stream) {
    hipLaunchKernelGGL(ff_small_bwd_final_kernel, dim3((3 * D + 63) / 64), dim3(1024), 0, as_stream(stream), partial, strips, 0, D,
                       nullptr, dgamma, dbeta, db);
    VITMI_LAUNCH_CHECK("ff_small_bwd_final_kernel");
    return VITMI_OK;
}
int vitmi_ff_small_bwd_strips(long long rows, int D, int M) {
    const int mt = ff_small_bwd_tiles(rows, D, M);
    return (int)((rows + mt * 16 - 1) / (mt * 16));
}

extern "C" int vitmi_small_sums_multi(const vitmi_sums_job* jobs, int n, void* stream) {
    VITMI_CHECK_ARG(jobs && n > 0, "vitmi_small_sums_multi: no jobs");
    for (int i = 0; i < n; ++i) {
        const vitmi_sums_job& j = jobs[i];
        VITMI_CHECK_ARG(j.partial && j.strips > 0 && j.M >= 0 && j.D > 0 && j.dgamma && j.dbeta && (j.M == 0 || j.db1),
                        "vitmi_small_sums_multi: job %d: bad pointer or shape (strips=%d M=%d D=%d)", i, j.strips, j.M, j.D);
    }
    for (int i0 = 0; i0 < n; i0 += SUMS_JOBS) {
        const int nj = n - i0 < SUMS_JOBS ? n - i0 : SUMS_JOBS;
        SumsTable t;
        int cols = 0;
        for (int i = 0; i < SUMS_JOBS; ++i) {
            t.j[i] = jobs[i0 + (i < nj ? i : 0)];
            if (i < nj && t.j[i].M + 3 * t.j[i].D > cols) cols = t.j[i].M + 3 * t.j[i].D;
        }
        hipLaunchKernelGGL(small_sums_multi_kernel, dim3((cols + 63) / 64, nj), dim3(1024), 0, as_stream(stream), t);
        VITMI_LAUNCH_CHECK("small_sums_multi_kernel");
    }
    return VITMI_OK;
}
