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

template <int MT>
int launch_ff_small(const FfSmallArgs& a, hipStream_t s) {
    const size_t lds = (size_t)MT * 16 * ((a.D + SM_PAD) + (a.M + SM_PAD)) * sizeof(bf16_t);
    static size_t granted = 0;
    if (lds > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_small_fwd_kernel<MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return vitmi_hip_fail(e, "ff_small_fwd: hipFuncSetAttribute");
        granted = lds;
    }
    const int grid = (int)((a.rows + MT * 16 - 1) / (MT * 16));
    hipLaunchKernelGGL(ff_small_fwd_kernel<MT>, dim3(grid), dim3(SM_THREADS), lds, s, a);
    VITMI_LAUNCH_CHECK("ff_small_fwd_kernel");
    return VITMI_OK;
}

constexpr int SMALL_DEFAULT = 3;
int g_small_fused = SMALL_DEFAULT;  // bit 0: MLP sub-block forward, bit 1: attention sub-block forward

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
