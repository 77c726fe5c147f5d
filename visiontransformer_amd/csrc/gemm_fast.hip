// Fast bf16 GEMM kernels for the aligned shapes of the ViT encoder (16-byte-aligned operands,
// leading dimensions and inner extents multiples of 8).  Both stage operands HBM -> LDS with
// direct-to-LDS loads (global_load_lds, 16 B per lane, no VGPR round trip), double-buffered, one
// barrier per 64-deep K-step, and compute 128x128 block tiles with 4 waves x (4x4)
// v_mfma_f32_16x16x32_bf16 accumulators.
//
//   nt: C[m][n] = sum_k A[m][k] B[n][k]      both operands contiguous along k
//       LDS image [row][64 k] (128-B rows); 16-B chunk c of row r sits at chunk
//       c ^ swz_nt(r): the swizzle is applied to the per-lane SOURCE address of the direct-to-LDS
//       load (its destination is lane-linear) and again on the ds_read_b128 of a fragment.
//       The MFMA is issued "swapped" (weight rows as the A operand) and weight rows are taken
//       in a permuted order so that a lane ends up with 16 consecutive n of one output row:
//       bias / GELU / residual / store all work on 32-64 contiguous bytes per lane.
//   tn: C[p][q] = sum_t U[t][p] V[t][q]      both operands contiguous along the OUTPUT index
//       (weight gradients: contraction over tokens).  LDS image [64 t][128 cols] (256-B rows),
//       fragments come out transposed through ds_read_b64_tr_b16.  Split over t into
//       blockIdx.z slabs (fp32) that a second kernel sums in slab order (deterministic).
//
// Out-of-range rows/chunks are fetched from a 16-byte zero word instead of being clamped, so
// ragged M, N, K contribute exact zeros.
#include <type_traits>

#include "common.h"

namespace {

constexpr int TK = 64;
constexpr int NT = 256;

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)lds_dst, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// nt kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_nt(int row) { return (row & 7) ^ (((row >> 4) & 1) << 2); }

// stage one [ROWS][64 k] tile with NW waves: wave w issues ROWS/8/NW loads, each covering 8 rows
// kstep: elements between the 64-wide k-blocks of a row.  64 (= TK) for an ordinary row-major operand; for the
// head-major planes of dqkv ([3*H][tokens][64], ld = 64) it is tokens * 64: k-block j of row m lives in plane j
template <int ROWS, int NW>
__device__ __forceinline__ void stage_nt(const bf16_t* __restrict__ g, long long ld, int row0, int rows, int k0, int K,
                                         bf16_t* lds, int wave, int lane, long long kstep = TK) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int rblk = (wave * PER + i) * 8;
        const int r = rblk + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(r);      // logical chunk this lane fetches
        const int gr = row0 + r, gk = k0 + c * 8;
        const void* src = (gr < rows && gk < K) ? (const void*)(g + (long long)gr * ld + (long long)(gk >> 6) * kstep + (gk & 63))
                                                : (const void*)g_zero16;
        glds16(src, lds + rblk * TK);
    }
}

// Patch operand: logical matrix P[token m][element e], m = (image b, gy, gx), e = (c, i, j), read in
// place from the NCHW image: P = x[b, c, gy*ps + i, gx*pw + j], patches ps rows high and pw pixels wide.  With
// pw % 8 == 0 a 16-byte chunk (8 elements) never leaves one patch row, so it is 16 contiguous bytes of the image.
// Square patches narrower than 8 pixels (the 4 x 4 patches of the 32-pixel geometry) are read as "wide" patches:
// pw = 8 spans 8 / ps horizontally adjacent patches, the row of the product is that group of tokens and the weight
// is the block form vitmi_patch_wide_weight builds.
struct PatchGeom {
    int C, H, W, ps, pw, gw, per;  // gw = (wide) patches per image row, per = per image
    int tokens, elems;             // B*per, C*ps*pw
};
__device__ __forceinline__ long long patch_addr(const PatchGeom& g, int m, int e) {
    const int b = m / g.per, t = m - b * g.per;
    const int gy = t / g.gw, gx = t - gy * g.gw;
    const int c = e / (g.ps * g.pw), q = e - c * g.ps * g.pw;
    const int i = q / g.pw, j = q - i * g.pw;
    return (((long long)b * g.C + c) * g.H + (gy * g.ps + i)) * g.W + gx * g.pw + j;
}

template <int ROWS, int NW>
__device__ __forceinline__ void stage_nt_patch(const bf16_t* __restrict__ img, const PatchGeom& pg, int row0, int k0,
                                               bf16_t* lds, int wave, int lane) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int rblk = (wave * PER + i) * 8;
        const int r = rblk + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(r);
        const int gr = row0 + r, gk = k0 + c * 8;
        const void* src = (gr < pg.tokens && gk < pg.elems) ? (const void*)(img + patch_addr(pg, gr, gk))
                                                            : (const void*)g_zero16;
        glds16(src, lds + rblk * TK);
    }
}

__device__ __forceinline__ bf16x8 frag_nt(const bf16_t* lds, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(lds + row * TK + ((chunk ^ swz_nt(row)) << 3));
}

struct FastEpi {
    void* C;
    long long ldc;
    const float* bias;
    const bf16_t* res;
    long long ldr;
    bf16_t* aux;
    long long ldaux;
    int rows_per_group;      // > 0: row m lives at C + (m / rpg) * group_stride + (m % rpg) * ldc
    long long group_stride;
    float* cs_partial;       // phased kernel only: [2 * m-tiles][N] column sums of the stored values
    // VITMI_EPI_LN / VITMI_EPI_LN_GELU: A holds the raw rows x and B the weight with LayerNorm's gamma folded in;
    // acc becomes rstd[m] * (acc - mean[m] * ln_s[n]) (ln_s[n] = row sum of the folded weight) before the bias
    const float* ln_mean;
    const float* ln_rstd;
    const float* ln_s;
    // > 0: C is written as head-major planes [N / 64][hm_T rows][64] (hm_T = M): column block j of row m at
    // C + (j * hm_T + m) * 64.  The layout the attention kernels read qkv in for heads of 64.
    long long hm_T;
    long long a_kstep;       // see stage_nt: elements between the 64-wide k-blocks of an A row (0 = contiguous rows)
};
template <typename OutT>
__device__ __forceinline__ OutT* epi_out_ptr(const FastEpi& e, int m, int n) {
    if (e.hm_T > 0) return reinterpret_cast<OutT*>(e.C) + ((long long)(n >> 6) * e.hm_T + m) * 64 + (n & 63);
    if (e.rows_per_group > 0) {
        const int gi = m / e.rows_per_group;
        return reinterpret_cast<OutT*>(e.C) + (long long)gi * e.group_stride + (long long)(m - gi * e.rows_per_group) * e.ldc + n;
    }
    return reinterpret_cast<OutT*>(e.C) + (long long)m * e.ldc + n;
}

// v[t] belongs to row m, column n + (t & 7) + (t >> 3) * H2: two runs of 8 consecutive columns, H2
// apart (H2 = 8: one run of 16)
template <int EPI, typename OutT, int H2 = 8>
__device__ __forceinline__ void epi_row16(const FastEpi& e, int m, int n, int N, float (&v)[16]) {
    const bool full = (n + H2 + 8 <= N);
    auto col = [&](int t) { return n + (t & 7) + (t >> 3) * H2; };
    if (epi_is_ln(EPI)) {
        const float mu = e.ln_mean[m], rs = e.ln_rstd[m];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = rs * (v[t] - mu * (col(t) < N ? e.ln_s[col(t)] : 0.f));
    }
    if (e.bias) {
        if (full) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + col(4 * t4));
#pragma unroll
                for (int t = 0; t < 4; ++t) v[4 * t4 + t] += b[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (col(t) < N) v[t] += e.bias[col(t)];
        }
    }
    if (epi_is_gelu(EPI)) {
        float w[16];  // what aux receives: the pre-activation, or the derivative of GELU at it
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (EPI == VITMI_EPI_GELU_D) {
                gelu_both_fast(v[t], v[t], w[t]);
            } else {
                w[t] = v[t];
                v[t] = EPI == VITMI_EPI_GELU ? gelu_erf_fast(v[t]) : gelu_fast(v[t]);
            }
        }
        if (e.aux) {
            bf16_t* a = e.aux + (long long)m * e.ldaux + n;
            if (full) {
                bf16x8 p0, p1;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    p0[t] = EPI == VITMI_EPI_GELU_D ? deriv_pack(w[t], bf16_t()) : (bf16_t)w[t];
                    p1[t] = EPI == VITMI_EPI_GELU_D ? deriv_pack(w[8 + t], bf16_t()) : (bf16_t)w[8 + t];
                }
                *reinterpret_cast<bf16x8*>(a) = p0;
                *reinterpret_cast<bf16x8*>(a + H2) = p1;
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (col(t) < N)
                        a[col(t) - n] = EPI == VITMI_EPI_GELU_D ? deriv_pack(w[t], bf16_t()) : (bf16_t)w[t];
            }
        }
    } else if (epi_is_auxmul(EPI)) {
        const bf16_t* a = e.aux + (long long)m * e.ldaux + n;
        if (full) {
            const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(a + H2);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(p0[t]) : gelu_grad_fast((float)p0[t]);
                v[8 + t] *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(p1[t]) : gelu_grad_fast((float)p1[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (col(t) < N)
                    v[t] *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(a[col(t) - n]) : gelu_grad_fast((float)a[col(t) - n]);
        }
    }
    if (e.res) {
        const bf16_t* r = e.res + (long long)m * e.ldr + n;
        if (full) {
            const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(r);
            const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(r + H2);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] += (float)p0[t];
                v[8 + t] += (float)p1[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (col(t) < N) v[t] += (float)r[col(t) - n];
        }
    }
    OutT* c = epi_out_ptr<OutT>(e, m, n);
    if (full) {
        if (sizeof(OutT) == 2) {
            bf16x8 p0, p1;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                p0[t] = (bf16_t)v[t];
                p1[t] = (bf16_t)v[8 + t];
            }
            *reinterpret_cast<bf16x8*>(c) = p0;
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(c) + H2) = p1;
        } else {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(c) + (col(4 * t4) - n)) =
                    f32x4{v[4 * t4], v[4 * t4 + 1], v[4 * t4 + 2], v[4 * t4 + 3]};
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (col(t) < N) c[col(t) - n] = from_f32<OutT>(v[t]);
    }
}

// Epilogue of a whole 128 x 64 wave sub-tile whose 64 columns are all inside the matrix (phased kernel; bf16 output).
// The eight row groups used to run one after the other as [load operand, wait, compute, store], and on this
// architecture stores and loads retire through one in-order counter, so every group's wait also waited for the
// previous group's stores: 8-16 dependent memory round trips per tile with the matrix pipes idle.  Here the bias is
// read once per tile and the row operand (residual, or the saved pre-activation of the dGELU form) of PD row groups
// is requested before the first group is processed, group i + PD right after group i has been stored; all loads of
// a group are older than every store issued after them, so the counted waits the compiler derives never wait for a
// store.  Rows past M read row M - 1 (always present) and are not stored.
template <int EPI, bool CS, bool HAS_OP>
__device__ __forceinline__ void epi_subtile_fast(const FastEpi& e, f32x4 (&acc)[4][8], int mrow0, int n, int M,
                                                 float* cs) {
    constexpr int PD = CS ? 4 : 8;
    f32x4 b[4];
    if (e.bias) {
        b[0] = *reinterpret_cast<const f32x4*>(e.bias + n);
        b[1] = *reinterpret_cast<const f32x4*>(e.bias + n + 4);
        b[2] = *reinterpret_cast<const f32x4*>(e.bias + n + 32);
        b[3] = *reinterpret_cast<const f32x4*>(e.bias + n + 36);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 ls[4];
    float lmu[8], lrs[8];
    if (epi_is_ln(EPI)) {
        ls[0] = *reinterpret_cast<const f32x4*>(e.ln_s + n);
        ls[1] = *reinterpret_cast<const f32x4*>(e.ln_s + n + 4);
        ls[2] = *reinterpret_cast<const f32x4*>(e.ln_s + n + 32);
        ls[3] = *reinterpret_cast<const f32x4*>(e.ln_s + n + 36);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = min(mrow0 + 16 * i, M - 1);
            lmu[i] = e.ln_mean[m];
            lrs[i] = e.ln_rstd[m];
        }
    }
    const bf16_t* op = epi_is_auxmul(EPI) ? e.aux : e.res;
    const long long ldop = epi_is_auxmul(EPI) ? e.ldaux : e.ldr;
    bf16x8 o[8][2];
    auto fetch = [&](int i) {
        const int m = min(mrow0 + 16 * i, M - 1);
        const bf16_t* p = op + (long long)m * ldop + n;
        o[i][0] = *reinterpret_cast<const bf16x8*>(p);
        o[i][1] = *reinterpret_cast<const bf16x8*>(p + 32);
    };
    if (HAS_OP) {
#pragma unroll
        for (int i = 0; i < PD; ++i) fetch(i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const int m = mrow0 + 16 * i;
        const bool live = m < M;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                v[4 * j + r] = epi_is_ln(EPI) ? fmaf(lrs[i], acc[j][i][r] - lmu[i] * ls[j][r], b[j][r]) : acc[j][i][r] + b[j][r];
        if (epi_is_gelu(EPI)) {
            bf16x8 p0, p1;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (EPI == VITMI_EPI_GELU_D) {
                    float d0, d1;
                    gelu_both_fast(v[t], v[t], d0);
                    gelu_both_fast(v[8 + t], v[8 + t], d1);
                    p0[t] = deriv_pack(d0, bf16_t());
                    p1[t] = deriv_pack(d1, bf16_t());
                } else {
                    p0[t] = (bf16_t)v[t];
                    p1[t] = (bf16_t)v[8 + t];
                    v[t] = EPI == VITMI_EPI_GELU ? gelu_erf_fast(v[t]) : gelu_fast(v[t]);
                    v[8 + t] = EPI == VITMI_EPI_GELU ? gelu_erf_fast(v[8 + t]) : gelu_fast(v[8 + t]);
                }
            }
            if (e.aux && live) {
                bf16_t* a = e.aux + (long long)m * e.ldaux + n;
                *reinterpret_cast<bf16x8*>(a) = p0;
                *reinterpret_cast<bf16x8*>(a + 32) = p1;
            }
        }
        if (HAS_OP) {
            if (epi_is_auxmul(EPI)) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    v[t] *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(o[i][0][t]) : gelu_grad_fast((float)o[i][0][t]);
                    v[8 + t] *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(o[i][1][t]) : gelu_grad_fast((float)o[i][1][t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    v[t] += (float)o[i][0][t];
                    v[8 + t] += (float)o[i][1][t];
                }
            }
        }
        if (live) {
            bf16x8 p0, p1;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                p0[t] = (bf16_t)v[t];
                p1[t] = (bf16_t)v[8 + t];
            }
            bf16_t* c = epi_out_ptr<bf16_t>(e, m, n);
            *reinterpret_cast<bf16x8*>(c) = p0;
            *reinterpret_cast<bf16x8*>(c + 32) = p1;
            if (CS) {
#pragma unroll
                for (int t = 0; t < 16; ++t) cs[t] += v[t];
            }
        }
        if (HAS_OP && i + PD < 8) fetch(i + PD);
    }
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a contiguous
// run of tiles; within a run the n-tile index moves fastest (neighbours share the A row panel).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int x = bid & 7, l = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

// BM x BN block tile, WM x WN waves, each wave a (BM/WM) x 64 sub-tile; dynamic LDS holds two
// stages of [BM + BN][64] bf16.
template <int EPI, typename OutT, bool PATCH, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_fast_kernel(const bf16_t* __restrict__ A, long long lda,
                                                                    const bf16_t* __restrict__ B, long long ldb,
                                                                    FastEpi epi, int M, int N, int K, int tiles_n,
                                                                    PatchGeom pg) {
    constexpr int NW = WM * WN, WTM = BM / WM, WTN = BN / WN, MI = WTM / 16, NJ = WTN / 16;
    static_assert(NJ == 4, "a lane owns 16 consecutive n = 4 n-tiles");
    constexpr int STAGE = (BM + BN) * TK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fg = lane >> 4;

    f32x4 acc[NJ][MI];  // [j: n-tile][i: m-tile]
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + TK - 1) / TK;
    // weight rows of n-tile j in the order that leaves n = 16*fg + 4*j + r in accumulator register r
    int wrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wrow[j] = wn * WTN + 16 * (fr >> 2) + 4 * j + (fr & 3);

    // Two LDS stages, one barrier per 64-deep K-tile: the direct-to-LDS loads of tile kt+1 are issued
    // before the MFMAs of tile kt and drained by the barrier's wait.  (A hand-pipelined variant with
    // two register fragment sets measured 10-25 % slower: it spills and defeats hipcc's own
    // interleave of ds_read and MFMA.)
    const long long a_kstep = epi.a_kstep > 0 ? epi.a_kstep : TK;
    if (PATCH)
        stage_nt_patch<BM, NW>(A, pg, m0, 0, smem, wave, lane);
    else
        stage_nt<BM, NW>(A, lda, m0, M, 0, K, smem, wave, lane, a_kstep);
    stage_nt<BN, NW>(B, ldb, n0, N, 0, K, smem + BM * TK, wave, lane);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        bf16_t* As = smem + (kt & 1) * STAGE;
        bf16_t* Bs = As + BM * TK;
        if (kt + 1 < nk) {
            bf16_t* An = smem + ((kt + 1) & 1) * STAGE;
            if (PATCH)
                stage_nt_patch<BM, NW>(A, pg, m0, (kt + 1) * TK, An, wave, lane);
            else
                stage_nt<BM, NW>(A, lda, m0, M, (kt + 1) * TK, K, An, wave, lane, a_kstep);
            stage_nt<BN, NW>(B, ldb, n0, N, (kt + 1) * TK, K, An + BM * TK, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < TK / 32; ++ks) {
            bf16x8 fw[NJ], fx[MI];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fw[j] = frag_nt(Bs, wrow[j], ks * 4 + fg);
#pragma unroll
            for (int i = 0; i < MI; ++i) fx[i] = frag_nt(As, wm * WTM + i * 16 + fr, ks * 4 + fg);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fx[i], acc[j][i], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        const int n = n0 + wn * WTN + 16 * fg;
        if (m < M && n < N) {
            float v[16];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[j][i][r];
            epi_row16<EPI, OutT>(epi, m, n, N, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// nt kernel, phased variant for large problems with K % 64 == 0.  256x256 block tile, 8 waves as
// 2 (m) x 4 (n), 64-deep K-tiles double-buffered in LDS.  A K-tile is four phases of 16 MFMAs (one
// quadrant of the wave's 128x64 sub-tile each); a phase is
//     [fragment reads + one 16-KiB region of direct-to-LDS loads] barrier [16 MFMAs] barrier
// and the two wave groups (m halves; one wave of each per SIMD) run one barrier apart, so one
// group's MFMAs execute beside the other group's LDS/VMEM work.
//
// A K-tile buffer is four regions of 128 rows x 64 k, cut by WHEN a row is read, not by where it
// sits in the matrix:  RA0 = the m rows of quadrants 0/1 of both groups, RB0 = the n rows of the
// two n-tiles used in phases 0 and 3 (the first 32 of each wave's 64 columns), RB1 = those of
// phases 1 and 2, RA1 = the m rows of phases
// 2/3.  Regions are streamed in that order, phase q issues region q+6 of the stream, and a
// counted s_waitcnt vmcnt(8) leaves four regions (64 KiB per CU) in flight across the barriers.
//   RAW: the wait that retires a region precedes the first barrier of the phase BEFORE its first
//        read, so both groups have passed a barrier after every wave's wait.
//   WAR: a region is refilled at least two phases after the phase that last read it.
// ------------------------------------------------------------------------------------------------
constexpr int RG = 128 * TK;  // elements per region

__device__ __forceinline__ void wait_vm_regions(int pend) {  // 2 loads per wave per region
    if (pend >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pend == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (pend == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (pend == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, typename OutT, bool CS>
__global__ __launch_bounds__(512) void gemm_nt_phased_kernel(const bf16_t* __restrict__ A, long long lda,
                                                             const bf16_t* __restrict__ B, long long ldb,
                                                             FastEpi epi, int M, int N, int K, int tiles_n,
                                                             int ntiles, int gm, int span, int xspan, int stagger_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nreg = 4 * (K / TK);  // regions in the stream
    const int tiles_m = ntiles / tiles_n;
    // Optional start-up stagger (ticks of the 100 MHz constant clock): every block would otherwise reach its
    // epilogue at the same moment, so the chip writes its outputs in bursts at fabric speed with the matrix pipes
    // idle.  Only blocks that own one tile fewer than the busiest blocks are delayed (at most `span + xspan` ticks,
    // chosen below one tile time), so the delay costs nothing.
    if (span > 0 || xspan > 0) {
        const int rem = ntiles % gridDim.x;
        if (stagger_all || (rem != 0 && (int)blockIdx.x >= rem)) {
            const unsigned long long d =
                (unsigned long long)((long long)span * (blockIdx.x >> 3) / 32 + (long long)xspan * (blockIdx.x & 7) / 8);
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
        }
    }
    // persistent: the grid is one block per CU (a multiple of 8, so a block keeps its XCD); output
    // stores of a tile drain while the next tile's loads and MFMAs run
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int tile = xcd_remap(vb, ntiles);
    // Tile order inside an XCD's run.  gm == 0: n fastest (the blocks of an XCD share A row panels).  gm > 0: groups
    // of gm row panels, m fastest inside a group, so the 32 blocks of an XCD work on gm row panels x 32/gm weight
    // tiles at a time: the gm A panels stay in the XCD's L2 while the group sweeps the weight once, and a weight
    // that does not fit the L2 is re-fetched once per group instead of once per row panel.
    int tm, tn;
    if (gm > 0) {
        const int per = gm * tiles_n;
        const int g = tile / per, r = tile - g * per;
        const int ge = min(gm, tiles_m - g * gm);
        tn = r / ge;
        tm = g * gm + (r - tn * ge);
    } else {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
    }
    const int m0 = tm * 256, n0 = tn * 256;

    // A region is 16 loads of 8 rows; wave w issues loads 2w and 2w+1 of every region.  src[s][i]:
    // per-lane source of load i of the next region of kind s (0 RA0, 1 RB0, 2 RB1, 3 RA1); rows
    // past the edge read the zero word and do not advance.
    const char* src[4][2];
    long long inc[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rho = (wave * 2 + i) * 8 + (lane >> 3);  // row inside the region
        const int c = (lane & 7) ^ swz_nt(rho);
        const int ra = (rho >> 6) * 128 + (rho & 63);           // RA0 row; RA1 = + 64
        const int rb = (rho >> 5) * 64 + (rho & 31);            // RB0 row; RB1 = + 32
        const int rows[4] = {ra, rb, rb + 32, ra + 64};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const bool isA = (s4 == 0 || s4 == 3);
            const int g = (isA ? m0 : n0) + rows[s4];
            const bool ok = g < (isA ? M : N);
            const bf16_t* base = isA ? A + (long long)g * lda : B + (long long)g * ldb;
            src[s4][i] = ok ? (const char*)(base + c * 8) : (const char*)g_zero16;
            inc[s4][i] = ok ? ((isA && epi.a_kstep > 0) ? epi.a_kstep * 2 : (long long)TK * 2) : 0;
        }
    }
    // issue region h of the stream (kind h & 3 of K-tile h >> 2); regions go out strictly in order
    auto stage = [&](int h, auto kind) {
        constexpr int S = decltype(kind)::value;
        bf16_t* dst = smem + ((h >> 2) & 1) * 4 * RG + S * RG + wave * 2 * 8 * TK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16(src[S][i], dst + i * 8 * TK);
            src[S][i] += inc[S][i];
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // fragment rows inside their regions
    int rowX[4], rowW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowX[i] = wr * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 2; ++j) rowW[j] = wc * 32 + 8 * (fr >> 2) + 4 * j + (fr & 3);
    auto frag = [&](const bf16_t* region, int row, int ks) {
        return *reinterpret_cast<const bf16x8*>(region + row * TK + (((ks * 4 + fg) ^ swz_nt(row)) << 3));
    };

    f32x4 acc[4][8];  // [j: n-tile][i: m-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: regions 0..5, regions 0 and 1 landed
    stage(0, K0{});
    stage(1, K1{});
    stage(2, K2{});
    stage(3, K3{});
    if (nreg > 4) {
        stage(4, K0{});
        stage(5, K1{});
    }
    wait_vm_regions(min(nreg, 6) - 2);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // second group runs one barrier behind

    bf16x8 fx[4][2], fw0[2][2], fw1[2][2];
    for (int q = 0; q < nreg; q += 4) {
        const bf16_t* buf = smem + ((q >> 2) & 1) * 4 * RG;
        // ---- phase 0: quadrant (m 0-3, n 0-1)
        if (q + 6 < nreg) stage(q + 6, K2{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw0[j][ks] = frag(buf + RG, rowW[j], ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fx[i][ks] = frag(buf, rowX[i], ks);
        wait_vm_regions(min(nreg, q + 7) - (q + 3));
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[j][ks], fx[i][ks], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: quadrant (m 0-3, n 2-3)
        if (q + 7 < nreg) stage(q + 7, K3{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw1[j][ks] = frag(buf + 2 * RG, rowW[j], ks);
        wait_vm_regions(min(nreg, q + 8) - (q + 4));
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 + j][i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[j][ks], fx[i][ks], acc[2 + j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (m 4-7, n 2-3)
        if (q + 8 < nreg) stage(q + 8, K0{});
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fx[i][ks] = frag(buf + 3 * RG, rowX[i], ks);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 + j][4 + i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[j][ks], fx[i][ks], acc[2 + j][4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (m 4-7, n 0-1); nothing to read
        if (q + 9 < nreg) stage(q + 9, K1{});
        wait_vm_regions(min(nreg, q + 10) - (q + 6));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j][4 + i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw0[j][ks], fx[i][ks], acc[j][4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // matches the second group's extra barrier

    float cs[CS ? 16 : 1];
    if (CS) {
#pragma unroll
        for (int t = 0; t < 16; ++t) cs[t] = 0.f;
    }
    // whole sub-tile inside the matrix in n, bf16 output, one row operand at most: the pipelined epilogue
    const bool fast_epi = sizeof(OutT) == 2 && n0 + 256 <= N && epi.rows_per_group == 0 &&
                          !(EPI != VITMI_EPI_NONE && epi.res);
    if (fast_epi) {
        const int mrow0 = m0 + wr * 128 + fr, nn = n0 + wc * 64 + 8 * fg;
        if constexpr (epi_is_auxmul(EPI)) {
            epi_subtile_fast<EPI, CS, true>(epi, acc, mrow0, nn, M, cs);
        } else if constexpr (epi_is_gelu(EPI)) {
            epi_subtile_fast<EPI, CS, false>(epi, acc, mrow0, nn, M, cs);
        } else {
            if (epi.res)
                epi_subtile_fast<EPI, CS, true>(epi, acc, mrow0, nn, M, cs);
            else
                epi_subtile_fast<EPI, CS, false>(epi, acc, mrow0, nn, M, cs);
        }
    } else
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // with the column sums the register file is full here: keep hipcc from hoisting the operand
        // loads of all eight row groups to the top (a spill's reload would wait for the tile's stores)
        if (CS && (i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        const int m = m0 + wr * 128 + i * 16 + fr;
        const int n = n0 + wc * 64 + 8 * fg;
        if (m < M && n < N) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[j][i][r];
            epi_row16<EPI, OutT, 32>(epi, m, n, N, v);
            if (CS) {
#pragma unroll
                for (int t = 0; t < 16; ++t) cs[t] += v[t];
            }
        }
    }
    if (CS) {
        // column sums of this wave's 128 rows: the 16 row lanes of a column group combine in a fixed
        // butterfly order, lane fr == 0 writes row (2 * m-tile + wr) of the partial matrix
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float x = cs[t];
            x += __shfl_xor(x, 1, 64);
            x += __shfl_xor(x, 2, 64);
            x += __shfl_xor(x, 4, 64);
            x += __shfl_xor(x, 8, 64);
            cs[t] = x;
        }
        const int n = n0 + wc * 64 + 8 * fg;
        if (fr == 0) {
            float* prow = epi.cs_partial + (long long)((m0 >> 8) * 2 + wr) * N + n;
            if (n + 40 <= N && (N & 3) == 0) {
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
                    *reinterpret_cast<f32x4*>(prow + (t4 & 1) * 4 + (t4 >> 1) * 32) =
                        f32x4{cs[4 * t4], cs[4 * t4 + 1], cs[4 * t4 + 2], cs[4 * t4 + 3]};
            } else {
                for (int t = 0; t < 16; ++t) {
                    const int c = (t & 7) + (t >> 3) * 32;
                    float x = cs[0];
#pragma unroll
                    for (int u = 1; u < 16; ++u) x = (t == u) ? cs[u] : x;
                    if (n + c < N) prow[c] = x;
                }
            }
        }
    }
    }  // tile loop
}

// ------------------------------------------------------------------------------------------------
// tn kernel (weight gradients)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_tn(int t) { return ((t & 3) | (((t >> 3) & 1) << 2)) << 1; }

// stage one [64 t][COLS] tile with NW waves: one load = 1 KiB = 512/COLS rows
// plane: 0 for a row-major operand; for head-major planes [cols / 64][tokens][64] (ld = 64) the elements of one
// plane, tokens * 64: column block j of token t lives at g + j * plane + t * 64
template <int COLS, int NW>
__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ g, long long ld, int t0, int T, int col0, int cols,
                                         bf16_t* lds, int wave, int lane, long long plane = 0) {
    constexpr int CPR = COLS / 8;          // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;          // rows per load instruction
    constexpr int PER = (64 / RPI) / NW;   // loads per wave
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int tblk = (wave * PER + i) * RPI;
        const int t = tblk + lane / CPR;
        const int c = (lane % CPR) ^ swz_tn(t);
        const int gt = t0 + t, gc = col0 + c * 8;
        const long long off = plane > 0 ? (long long)(gc >> 6) * plane + (long long)gt * 64 + (gc & 63) : (long long)gt * ld + gc;
        const void* src = (gt < T && gc < cols) ? (const void*)(g + off) : (const void*)g_zero16;
        glds16(src, lds + tblk * COLS);
    }
}

template <int COLS, int NW>
__device__ __forceinline__ void stage_tn_patch(const bf16_t* __restrict__ img, const PatchGeom& pg, int t0, int col0,
                                               bf16_t* lds, int wave, int lane) {
    constexpr int CPR = COLS / 8, RPI = 64 / CPR, PER = (64 / RPI) / NW;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int tblk = (wave * PER + i) * RPI;
        const int t = tblk + lane / CPR;
        const int c = (lane % CPR) ^ swz_tn(t);
        const int gt = t0 + t, gc = col0 + c * 8;
        const void* src = (gt < pg.tokens && gc < pg.elems) ? (const void*)(img + patch_addr(pg, gt, gc))
                                                            : (const void*)g_zero16;
        glds16(src, lds + tblk * COLS);
    }
}

// fragment for 16 consecutive columns col0..col0+15, k-step ks (32 t): element j of lane (fr,fg)
// = tile[t = 32 ks + 8 fg + j][col0 + fr]
template <int COLS>
__device__ __forceinline__ bf16x8 frag_tn(const bf16_t* lds, int col0, int ks, int lane) {
    const int fg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 4 * p;              // 4 consecutive columns, 8 bytes
    const int t_lo = 32 * ks + 8 * fg + q, t_hi = t_lo + 4;
    const int o_lo = t_lo * COLS + ((((col >> 3) ^ swz_tn(t_lo)) << 3) | (col & 7));
    const int o_hi = t_hi * COLS + ((((col >> 3) ^ swz_tn(t_hi)) << 3) | (col & 7));
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + o_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + o_hi));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// BP x BQ output tile, WP x WQ waves, each wave (BP/WP) x (BQ/WQ); dynamic LDS = two stages of
// [64 t][BP + BQ] bf16
// SEG: the token axis is cut into segments of `kt_per_slab` TOKENS (one image each) instead of slabs of K-tiles;
// the grid is tiles x segment groups, and a block writes, for every segment of its group, the sum of squares of
// its tile of the segment's product to C[segment * tiles + tile] rather than the tile itself (per-sample
// weight-gradient norms).
template <bool PATCH, int BP, int BQ, int WP, int WQ, bool SEG = false>
__global__ __launch_bounds__(64 * WP * WQ) void gemm_tn_fast_kernel(const bf16_t* __restrict__ U, long long ldu,
                                                                    const bf16_t* __restrict__ V, long long ldv,
                                                                    float* __restrict__ C, long long ldc,
                                                                    long long slab_stride, int P, int Q, int T,
                                                                    int tiles_q, int tiles_pq, int kt_per_slab,
                                                                    PatchGeom pg, long long u_plane) {
    constexpr int NW = WP * WQ, WTP = BP / WP, WTQ = BQ / WQ, MI = WTP / 16, NJ = WTQ / 16;
    constexpr int STAGE = TK * (BP + BQ);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    // 1-D grid of tiles x slabs, remapped so that each XCD (blocks b, b+8, ...) owns a contiguous run
    // in which the tile index moves fastest: the blocks of one token slab run on one XCD at about
    // the same time and share its L2 copy of the slab's U and V rows
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = lin / tiles_pq, tile = lin - slab * tiles_pq;
    const int p0 = (tile / tiles_q) * BP, q0 = (tile % tiles_q) * BQ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WQ, wq = wave % WQ;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[NJ][MI];  // [j: q-tile][i: p-tile]
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mfma_tile = [&](const bf16_t* Us, const bf16_t* Vs) {
#pragma unroll
        for (int ks = 0; ks < TK / 32; ++ks) {
            bf16x8 fv[NJ], fu[MI];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fv[j] = frag_tn<BQ>(Vs, wq * WTQ + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) fu[i] = frag_tn<BP>(Us, wp * WTP + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[j], fu[i], acc[j][i], 0, 0, 0);
        }
    };

    if constexpr (SEG) {
        // The block walks a contiguous group of segments with ONE two-stage pipeline: the K-tile sequence is
        // (segment, 64-token tile) flattened, the last tile of a segment zero-masked past the segment's end.  When a
        // segment's last tile has been multiplied the accumulators are reduced to a sum of squares and cleared.
        const int seglen = kt_per_slab;
        const int nseg = (T + seglen - 1) / seglen;
        const int groups = gridDim.x / tiles_pq;
        const int per = (nseg + groups - 1) / groups;
        const int s_lo = slab * per, s_hi = min(nseg, s_lo + per);
        const int ktps = (seglen + TK - 1) / TK;
        const int n_it = s_hi > s_lo ? (s_hi - s_lo) * ktps : 0;
        float* red = reinterpret_cast<float*>(smem + 2 * STAGE);        // [2][NW], beyond the two stages
        auto stage = [&](int it, bf16_t* dst) {
            const int sg = s_lo + it / ktps, kt = it % ktps;
            const int tb = sg * seglen, tl = min(T, tb + seglen);
            stage_tn<BP, NW>(U, ldu, tb + kt * TK, tl, p0, P, dst, wave, lane, u_plane);
            stage_tn<BQ, NW>(V, ldv, tb + kt * TK, tl, q0, Q, dst + TK * BP, wave, lane);
        };
        if (n_it > 0) stage(0, smem);
        __syncthreads();
        for (int it = 0; it < n_it; ++it) {
            bf16_t* Us = smem + (it & 1) * STAGE;
            if (it + 1 < n_it) stage(it + 1, smem + ((it + 1) & 1) * STAGE);
            mfma_tile(Us, Us + TK * BP);
            const bool last = (it % ktps) == ktps - 1;
            const int sg = s_lo + it / ktps;
            if (last) {
                // rows and columns past P, Q were staged as zeros, so the whole tile can be summed
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) sq = fmaf(acc[j][i][r], acc[j][i][r], sq);
                        acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                sq = wave_sum(sq);
                if (lane == 0) red[(sg & 1) * NW + wave] = sq;
            }
            __syncthreads();
            if (last && threadIdx.x == 0) {
                float t = 0.f;
                for (int w = 0; w < NW; ++w) t += red[(sg & 1) * NW + w];
                C[(long long)sg * tiles_pq + tile] = t;
            }
        }
        return;
    }

    const int nk_all = (T + TK - 1) / TK;
    const int kt_begin = slab * kt_per_slab;
    const int kt_end = min(kt_begin + kt_per_slab, nk_all);
    if (kt_begin < kt_end) {
        stage_tn<BP, NW>(U, ldu, kt_begin * TK, T, p0, P, smem, wave, lane, u_plane);
        if (PATCH)
            stage_tn_patch<BQ, NW>(V, pg, kt_begin * TK, q0, smem + TK * BP, wave, lane);
        else
            stage_tn<BQ, NW>(V, ldv, kt_begin * TK, T, q0, Q, smem + TK * BP, wave, lane);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int b = (kt - kt_begin) & 1;
        bf16_t* Us = smem + b * STAGE;
        bf16_t* Vs = Us + TK * BP;
        if (kt + 1 < kt_end) {
            bf16_t* Un = smem + (b ^ 1) * STAGE;
            stage_tn<BP, NW>(U, ldu, (kt + 1) * TK, T, p0, P, Un, wave, lane, u_plane);
            if (PATCH)
                stage_tn_patch<BQ, NW>(V, pg, (kt + 1) * TK, q0, Un + TK * BP, wave, lane);
            else
                stage_tn<BQ, NW>(V, ldv, (kt + 1) * TK, T, q0, Q, Un + TK * BP, wave, lane);
        }
        mfma_tile(Us, Vs);
        __syncthreads();
    }
    // acc[j][i][r] = C[p = p0 + wp*WTP + i*16 + fr][q = q0 + wq*WTQ + j*16 + 4*fg + r]
    float* out = C + (long long)slab * slab_stride;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = p0 + wp * WTP + i * 16 + fr;
        if (p >= P) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wq * WTQ + j * 16 + 4 * fg;
            if (q + 4 <= Q) {
                *reinterpret_cast<f32x4*>(out + (long long)p * ldc + q) = acc[j][i];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (q + r < Q) out[(long long)p * ldc + q + r] = acc[j][i][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Many weight gradients over one token axis in ONE launch: job j is C_j[P_j][Q_j] = U_j^T V_j over the same T tokens
// (the weight gradients of several sub-blocks, formed after the backward pass instead of one small launch each
// beside it).  The grid is slabs x (the 128 x 128 tiles of all jobs); a block finds its job from the tile prefix,
// runs the two-stage loop of gemm_tn_fast_kernel over its slab of token tiles and writes a dense [P_j][Q_j] fp32
// partial (or C_j itself with one slab); tn_multi_reduce_kernel adds the slabs in index order.
// ------------------------------------------------------------------------------------------------
constexpr int TNM_MAX = 16;
struct TnMulti {
    const bf16_t* U[TNM_MAX];
    const bf16_t* V[TNM_MAX];
    float* C[TNM_MAX];
    long long slab_off[TNM_MAX];   // first float of job j's [ns][P][Q] partials inside `slabs`
    int ldu[TNM_MAX], ldv[TNM_MAX], ldc[TNM_MAX], P[TNM_MAX], Q[TNM_MAX], tiles_q[TNM_MAX];
    int tile_begin[TNM_MAX + 1];
    float* slabs;
    int njobs, T, ns, per;         // per: token tiles per slab
};

__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(const TnMulti tm) {
    constexpr int BP = 128, BQ = 128, WP = 2, WQ = 2, NW = 4, WTP = 64, WTQ = 64, MI = 4, NJ = 4;
    constexpr int STAGE = TK * (BP + BQ);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tiles = tm.tile_begin[tm.njobs];
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = lin / tiles;
    int tile = lin - slab * tiles;
    int job = 0;
#pragma unroll
    for (int k = 1; k < TNM_MAX; ++k)
        if (k < tm.njobs && tile >= tm.tile_begin[k]) job = k;
    tile -= tm.tile_begin[job];
    const bf16_t* __restrict__ U = tm.U[job];
    const bf16_t* __restrict__ V = tm.V[job];
    const long long ldu = tm.ldu[job], ldv = tm.ldv[job];
    const int P = tm.P[job], Q = tm.Q[job], tq = tm.tiles_q[job], T = tm.T;
    const int p0 = (tile / tq) * BP, q0 = (tile % tq) * BQ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave / WQ, wq = wave % WQ;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk_all = (T + TK - 1) / TK;
    const int kt_begin = slab * tm.per, kt_end = min(kt_begin + tm.per, nk_all);
    if (kt_begin < kt_end) {
        stage_tn<BP, NW>(U, ldu, kt_begin * TK, T, p0, P, smem, wave, lane);
        stage_tn<BQ, NW>(V, ldv, kt_begin * TK, T, q0, Q, smem + TK * BP, wave, lane);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int b = (kt - kt_begin) & 1;
        bf16_t* Us = smem + b * STAGE;
        bf16_t* Vs = Us + TK * BP;
        if (kt + 1 < kt_end) {
            bf16_t* Un = smem + (b ^ 1) * STAGE;
            stage_tn<BP, NW>(U, ldu, (kt + 1) * TK, T, p0, P, Un, wave, lane);
            stage_tn<BQ, NW>(V, ldv, (kt + 1) * TK, T, q0, Q, Un + TK * BP, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < TK / 32; ++ks) {
            bf16x8 fv[NJ], fu[MI];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fv[j] = frag_tn<BQ>(Vs, wq * WTQ + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i) fu[i] = frag_tn<BP>(Us, wp * WTP + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[j], fu[i], acc[j][i], 0, 0, 0);
        }
        __syncthreads();
    }
    // acc[j][i][r] = C[p = p0 + wp*WTP + i*16 + fr][q = q0 + wq*WTQ + j*16 + 4*fg + r]
    float* out;
    long long ldo;
    if (tm.ns > 1) {
        out = tm.slabs + tm.slab_off[job] + (long long)slab * P * Q;
        ldo = Q;
    } else {
        out = tm.C[job];
        ldo = tm.ldc[job];
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = p0 + wp * WTP + i * 16 + fr;
        if (p >= P) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = q0 + wq * WTQ + j * 16 + 4 * fg;
            if (q + 4 <= Q) *reinterpret_cast<f32x4*>(out + (long long)p * ldo + q) = acc[j][i];   // Q % 8 == 0
        }
    }
}

// C_j[p][q] = sum over slabs of job j's partials, in slab order; grid (blocks, jobs)
__global__ void tn_multi_reduce_kernel(const TnMulti tm) {
    const int job = blockIdx.y;
    const int P = tm.P[job], Q = tm.Q[job];
    const long long n4 = (long long)P * Q / 4, pq = (long long)P * Q;
    const float* __restrict__ src = tm.slabs + tm.slab_off[job];
    float* __restrict__ C = tm.C[job];
    const long long ldc = tm.ldc[job];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 s = *reinterpret_cast<const f32x4*>(src + 4 * i);
        for (int k = 1; k < tm.ns; ++k) s += *reinterpret_cast<const f32x4*>(src + k * pq + 4 * i);
        const long long e = 4 * i, p = e / Q, q = e - p * Q;
        *reinterpret_cast<f32x4*>(C + p * ldc + q) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// tn kernel, phased variant (T % 64 == 0): the schedule of gemm_nt_phased_kernel (four phases of 16
// MFMAs per 64-deep token tile, two wave groups one barrier apart, regions streamed six ahead
// with a counted vmcnt) on [64 t][128 cols] regions read through ds_read_b64_tr_b16:
//   RU0 = p columns of the first 64 of each group's 128, RV0 = first 32 of each wave's 64 q
//   columns, RV1 = the other 32, RU1 = the other 64 p columns.
// The transposed reads are issued from inline assembly: hipcc puts s_waitcnt vmcnt(0) in front of
// the ds_read_tr builtin whenever a direct-to-LDS load is outstanding, which would serialise every
// token tile behind its own loads.  Their results are therefore covered by the explicit
// lgkmcnt(0) that follows each phase's first barrier.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x2 tr_read(unsigned lds_addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
    return r;
}

// One 256 x 256 tile of C = U^T V over the token tiles [kt_begin, kt_end), all 512 threads of the block:
// out[p * ldo + q] for p0 <= p < min(p0 + 256, P), q0 <= q < min(q0 + 256, Q).  Leaves every wave behind the
// same number of barriers with no LDS read outstanding, so a block may call it again (the direct-to-LDS loads of
// the next call queue behind this call's stores in the in-order vmcnt).
__device__ __forceinline__ void tn_phased_tile(const bf16_t* __restrict__ U, long long ldu,
                                               const bf16_t* __restrict__ V, long long ldv, int P, int Q, int p0,
                                               int q0, int kt_begin, int kt_end, float* __restrict__ out,
                                               long long ldo, unsigned char* smem_raw, long long u_plane = 0) {
    constexpr int RGT = TK * 128;  // elements per region: 64 t x 128 columns
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    typedef __attribute__((address_space(3))) unsigned char* lds_u8p;
    const unsigned lds_base = (unsigned)(size_t)(lds_u8p)smem_raw;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nreg = 4 * max(kt_end - kt_begin, 0);

    // a region is 16 loads of 4 token rows; wave w issues loads 2w and 2w+1 of every region
    const char* src[4][2];
    long long inc[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tl = (wave * 2 + i) * 4 + (lane >> 4);  // token row inside the tile
        const int lc = ((lane & 15) ^ swz_tn(tl)) * 8;    // first of 8 logical columns
        const int cu = (lc >> 6) * 128 + (lc & 63);       // RU0 column; RU1 = + 64
        const int cv = (lc >> 5) * 64 + (lc & 31);        // RV0 column; RV1 = + 32
        const int cols[4] = {cu, cv, cv + 32, cu + 64};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const bool isU = (s4 == 0 || s4 == 3);
            const int g = (isU ? p0 : q0) + cols[s4];
            const bool ok = g < (isU ? P : Q);
            const bool planes = isU && u_plane > 0;     // U as head-major planes [P / 64][tokens][64]
            const long long ld = planes ? 64 : (isU ? ldu : ldv);
            const bf16_t* base = (isU ? U : V) + ((long long)kt_begin * TK + tl) * ld +
                                 (planes ? (long long)(g >> 6) * u_plane + (g & 63) : (long long)g);
            src[s4][i] = ok ? (const char*)base : (const char*)g_zero16;
            inc[s4][i] = ok ? ld * TK * 2 : 0;
        }
    }
    auto stage = [&](int h, auto kind) {
        constexpr int S = decltype(kind)::value;
        bf16_t* dst = smem + ((h >> 2) & 1) * 4 * RGT + S * RGT + wave * 2 * 4 * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16(src[S][i], dst + i * 4 * 128);
            src[S][i] += inc[S][i];
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // fragment of 16 consecutive region columns col0.., k-step ks: element j of lane (fr, fg) =
    // region[t = 32 ks + 8 fg + j][col0 + fr]; byte offsets of its two transposed reads
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    auto frag_off = [&](int col0, int ks, int half) {
        const int col = col0 + 4 * tp;
        const int t = 32 * ks + 8 * fg + tq + 4 * half;
        return 2 * (t * 128 + ((((col >> 3) ^ swz_tn(t)) << 3) | (col & 7)));
    };
    auto frag = [&](unsigned region_addr, int col0, int ks) {
        union {
            u32x2 h[2];
            bf16x8 v;
        } u;
        u.h[0] = tr_read(region_addr + frag_off(col0, ks, 0));
        u.h[1] = tr_read(region_addr + frag_off(col0, ks, 1));
        return u.v;
    };

    f32x4 acc[4][8];  // [j: q-tile][i: p-tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nreg > 0) {
        stage(0, K0{});
        stage(1, K1{});
        stage(2, K2{});
        stage(3, K3{});
        if (nreg > 4) {
            stage(4, K0{});
            stage(5, K1{});
        }
    }
    wait_vm_regions(min(nreg, 6) - 2);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();

    bf16x8 fu[4][2], fv0[2][2], fv1[2][2];
    for (int q = 0; q < nreg; q += 4) {
        const unsigned buf = lds_base + ((q >> 2) & 1) * 4 * RGT * 2;
        // ---- phase 0: quadrant (p 0-3, q 0-1)
        if (q + 6 < nreg) stage(q + 6, K2{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fv0[j][ks] = frag(buf + RGT * 2, wc * 32 + j * 16, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fu[i][ks] = frag(buf, wr * 64 + i * 16, ks);
        wait_vm_regions(min(nreg, q + 7) - (q + 3));
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv0[j][ks], fu[i][ks], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: quadrant (p 0-3, q 2-3)
        if (q + 7 < nreg) stage(q + 7, K3{});
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fv1[j][ks] = frag(buf + 2 * RGT * 2, wc * 32 + j * 16, ks);
        wait_vm_regions(min(nreg, q + 8) - (q + 4));
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 + j][i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv1[j][ks], fu[i][ks], acc[2 + j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (p 4-7, q 2-3)
        if (q + 8 < nreg) stage(q + 8, K0{});
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fu[i][ks] = frag(buf + 3 * RGT * 2, wr * 64 + i * 16, ks);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[2 + j][4 + i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv1[j][ks], fu[i][ks], acc[2 + j][4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (p 4-7, q 0-1); nothing to read
        if (q + 9 < nreg) stage(q + 9, K1{});
        wait_vm_regions(min(nreg, q + 10) - (q + 6));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[j][4 + i] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv0[j][ks], fu[i][ks], acc[j][4 + i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();

    // acc[j][i][r] = C[p = p0 + wr*128 + i*16 + fr][q = q0 + wc*64 + j*16 + 4*fg + r]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int pp = p0 + wr * 128 + i * 16 + fr;
        if (pp >= P) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = q0 + wc * 64 + j * 16 + 4 * fg;
            if (qq + 4 <= Q) {
                *reinterpret_cast<f32x4*>(out + (long long)pp * ldo + qq) = acc[j][i];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (qq + r < Q) out[(long long)pp * ldo + qq + r] = acc[j][i][r];
            }
        }
    }
}

__global__ __launch_bounds__(512) void gemm_tn_phased_kernel(const bf16_t* __restrict__ U, long long ldu,
                                                             const bf16_t* __restrict__ V, long long ldv,
                                                             float* __restrict__ C, long long ldc,
                                                             long long slab_stride, int P, int Q, int tiles_q,
                                                             int tiles_pq, int kt_per_slab, int nk_all, long long u_plane) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = lin / tiles_pq, tile = lin - slab * tiles_pq;
    const int p0 = (tile / tiles_q) * 256, q0 = (tile % tiles_q) * 256;
    const int kt_begin = slab * kt_per_slab;
    const int kt_end = min(kt_begin + kt_per_slab, nk_all);
    tn_phased_tile(U, ldu, V, ldv, P, Q, p0, q0, kt_begin, kt_end, C + (long long)slab * slab_stride, ldc, smem_raw, u_plane);
}

// Several weight gradients over the SAME token axis in one launch that fills the chip once (each product alone
// would pay its own launch, partial-tile write and reduction).  All outputs are multiples of 256 x 256; `tiles`
// output tiles in all, nk token tiles each.  The first tiles * nmain blocks ("main") own token tiles
// [slab * a, min((slab + 1) * a, kmain)) of one output tile; the last r = nk - kmain token tiles of ALL output tiles
// form a tile-major list of tiles * r units that the other blocks take in runs of upl = m * r units, i.e. m whole
// output tiles one after the other (r = 0: equal slabs only).  Every piece is written as a dense 256 x 256 fp32
// partial into slot tile * spt + index (main: its slab; run: nmain), which tn_batch_reduce_kernel sums in slot
// order: deterministic, no atomics.
struct TnBatchProblem {
    const bf16_t* U;
    const bf16_t* V;
    float* C;
    long long ldu, ldv, ldc;
    long long u_plane;       // > 0: U is head-major planes [P / 64][tokens][64], this many elements per plane
    int Q, tiles_q, tile_begin, pad;
};
struct TnBatch {
    TnBatchProblem pr[4];
    float* partial;
    int nprob, tiles, nk, nmain, a, r, upl, spt, main_blocks, kmain;
};
constexpr int TN_SLOT = 256 * 256;

__device__ __forceinline__ int tn_batch_problem(const TnBatch& tb, int tile) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < tb.nprob && tile >= tb.pr[k].tile_begin) i = k;
    return i;
}

__global__ __launch_bounds__(512) void gemm_tn_batch_kernel(const TnBatch tb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // blockIdx & 7 is the XCD.  Every XCD takes an equal share of the runs (they share no operands with their
    // neighbours, so one XCD holding all of them would be the slowest); its main blocks are consecutive in the
    // slab-major order, i.e. neighbouring output tiles over the same tokens.
    const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int nq = gridDim.x >> 3, nr = gridDim.x & 7;
    const int runs = gridDim.x - tb.main_blocks, rq = runs >> 3, rr = runs & 7;
    const int main_x = nq + (x < nr) - rq - (x < rr);
    const bool is_main = l < main_x;
    const int run = x * rq + min(x, rr) + (l - main_x);
    int u = 0, u_end = 0, tile, kb, ke, slot;
    if (is_main) {
        const int lin = x * (nq - rq) + min(x, nr) - min(x, rr) + l;
        const int slab = lin / tb.tiles;
        tile = lin - slab * tb.tiles;
        kb = slab * tb.a;
        ke = min(kb + tb.a, tb.kmain);
        slot = tile * tb.spt + slab;
    } else {
        u = run * tb.upl;
        u_end = min(u + tb.upl, tb.tiles * tb.r);
        tile = -1;
    }
    for (;;) {
        if (!is_main) {
            if (u >= u_end) break;
            tile = u / tb.r;
            const int k = u - tile * tb.r;
            const int len = min(tb.r - k, u_end - u);
            kb = tb.kmain + k;
            ke = kb + len;
            slot = tile * tb.spt + tb.nmain + run - (tile * tb.r) / tb.upl;
            u += len;
        }
        const int i = tn_batch_problem(tb, tile);
        const TnBatchProblem& pr = tb.pr[i];
        const int lt = tile - pr.tile_begin;
        const int p0 = (lt / pr.tiles_q) * 256, q0 = (lt % pr.tiles_q) * 256;
        float* out = tb.partial + (long long)slot * TN_SLOT - ((long long)p0 * 256 + q0);
        tn_phased_tile(pr.U, pr.ldu, pr.V, pr.ldv, 1 << 30, 1 << 30, p0, q0, kb, ke, out, 256, smem_raw, pr.u_plane);
        if (is_main) break;
    }
}

// C tile = sum of its partial slots in slot order; block = (tile, one sixteenth of the tile)
__global__ __launch_bounds__(256) void tn_batch_reduce_kernel(const TnBatch tb) {
    const int tile = blockIdx.x >> 4, part = blockIdx.x & 15;
    int n = tb.nmain;
    if (tb.r > 0) n += ((tile + 1) * tb.r - 1) / tb.upl - (tile * tb.r) / tb.upl + 1;
    const int i = tn_batch_problem(tb, tile);
    const TnBatchProblem& pr = tb.pr[i];
    const int lt = tile - pr.tile_begin;
    const int p0 = (lt / pr.tiles_q) * 256, q0 = (lt % pr.tiles_q) * 256;
    const float* base = tb.partial + (long long)tile * tb.spt * TN_SLOT;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e4 = part * 1024 + it * 256 + threadIdx.x;  // float4 index inside the tile
        const int row = e4 >> 6, col = (e4 & 63) * 4;
        f32x4 s = *reinterpret_cast<const f32x4*>(base + 4 * (long long)e4);
        int k = 1;
        for (; k + 1 < n; k += 2) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(base + (long long)k * TN_SLOT + 4 * (long long)e4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(base + (long long)(k + 1) * TN_SLOT + 4 * (long long)e4);
            s += v0;
            s += v1;
        }
        if (k < n) s += *reinterpret_cast<const f32x4*>(base + (long long)k * TN_SLOT + 4 * (long long)e4);
        *reinterpret_cast<f32x4*>(pr.C + (long long)(p0 + row) * pr.ldc + q0 + col) = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-image squared Frobenius norm of a weight gradient without forming it ("ghost" norm):
//     || U_i^T V_i ||_F^2 = sum over token pairs (t, t') of (U_i U_i^T)[t][t'] * (V_i V_i^T)[t][t'],
// U_i [n, P] and V_i [n, Q] the rows of image i (n <= 256 tokens).  Two Gram matrices of n x n cost
// 2 n^2 (P + Q) flops where the gradient itself costs 2 n P Q (3x fewer at 197 tokens for the MLP weights), and
// nothing is written but one float.  One 16-wave workgroup per image: the image's rows stream through LDS in 64-deep
// k-tiles (two stages, rows past n zero-filled, the swizzled row image of the nt kernels), each wave owns up to TPW
// 16 x 16 tiles of the upper triangle (both Grams in the same registers' layout, so the final product is elementwise
// in registers; off-diagonal tiles count twice), and the workgroup's sum is ADDED to sqnorm[i] by one thread.
// ------------------------------------------------------------------------------------------------
template <int TPW>
__global__ __launch_bounds__(1024) void ghost_sqnorm_kernel(const bf16_t* __restrict__ U, long long ldu, int P,
                                                            const bf16_t* __restrict__ V, long long ldv, int Q, int T,
                                                            int seg, int nt, float* __restrict__ sqnorm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    __shared__ float red[16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int r0 = blockIdx.x * seg;
    const int n = min(seg, T - r0);
    const int stage_elems = nt * 16 * TK;
    const int ntiles = nt * (nt + 1) / 2;
    // this wave's tiles: TPW consecutive entries of the row-major upper triangle (mostly one tile row: its fragment is
    // read once per k-step)
    int tr[TPW], tc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        int t = wave * TPW + j, r = 0;
        if (t >= ntiles) t = -1;
        if (t >= 0)
            while (t >= nt - r) {
                t -= nt - r;
                ++r;
            }
        tr[j] = t < 0 ? -1 : r;
        tc[j] = t < 0 ? -1 : r + t;
    }
    f32x4 acc[2][TPW];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[g][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // groups of 8 rows: 2 * nt of them, group g by wave g % 16
    auto stage = [&](const bf16_t* base, long long ld, int K, int k0, bf16_t* dst) {
        for (int g = wave; g < 2 * nt; g += 16) {
            const int r = g * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_nt(r);
            const int gk = k0 + c * 8;
            const void* src = (r < n && gk < K) ? (const void*)(base + (long long)(r0 + r) * ld + gk) : (const void*)g_zero16;
            glds16(src, dst + g * 8 * TK);
        }
    };
    const int nkU = (P + TK - 1) / TK, nkV = (Q + TK - 1) / TK, nk = nkU + nkV;
    auto stage_it = [&](int it, bf16_t* dst) {
        if (it < nkU)
            stage(U, ldu, P, it * TK, dst);
        else
            stage(V, ldv, Q, (it - nkU) * TK, dst);
    };
    stage_it(0, smem);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const bf16_t* cur = smem + (it & 1) * stage_elems;
        if (it + 1 < nk) stage_it(it + 1, smem + ((it + 1) & 1) * stage_elems);
        const bool second = it >= nkU;
#pragma unroll
        for (int ks = 0; ks < TK / 32; ++ks) {
            bf16x8 fa = frag_nt(cur, (tr[0] < 0 ? 0 : tr[0]) * 16 + fr, ks * 4 + fg);
            int have = tr[0];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if (tr[j] < 0) continue;
                if (tr[j] != have) {
                    fa = frag_nt(cur, tr[j] * 16 + fr, ks * 4 + fg);
                    have = tr[j];
                }
                const bf16x8 fb = frag_nt(cur, tc[j] * 16 + fr, ks * 4 + fg);
                if (second)
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[1][j], 0, 0, 0);
                else
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[0][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        if (tr[j] < 0) continue;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fmaf(acc[0][j][r], acc[1][j][r], t);
        tot += tr[j] == tc[j] ? t : 2.f * t;
    }
    tot = wave_sum(tot);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        sqnorm[blockIdx.x] += t;
    }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, long long slab_stride, int nslabs,
                                   float* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        // slabs are added in index order; four loads are issued per trip so that they overlap
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + 4 * i);
        int k = 1;
        for (; k + 3 < nslabs; k += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(slabs + k * slab_stride + 4 * i);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(slabs + (k + 1) * slab_stride + 4 * i);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(slabs + (k + 2) * slab_stride + 4 * i);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(slabs + (k + 3) * slab_stride + 4 * i);
            s += v0;
            s += v1;
            s += v2;
            s += v3;
        }
        for (; k < nslabs; ++k) s += *reinterpret_cast<const f32x4*>(slabs + k * slab_stride + 4 * i);
        *reinterpret_cast<f32x4*>(out + 4 * i) = s;
    }
}

// out[c][r] (bf16) = in[r][c] (fp32 or bf16): transposed compute copies of the weights
template <typename S>
__global__ void transpose_cast_kernel(const S* __restrict__ in, bf16_t* __restrict__ out, int R, int Cn) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < Cn) ? to_f32(in[(long long)r * Cn + c]) : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < Cn && r < R) out[(long long)c * R + r] = (bf16_t)tile[tx][k];
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// true when the nt fast path can take the problem
bool vitmi_gemm_nt_fast_ok(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                           const void* C, long long ldc, int c_dt, const void* res, long long ldr, const void* aux,
                           long long ldaux) {
    (void)M;
    (void)N;
    if (K % 8 || lda % 8 || ldb % 8 || !al16(A) || !al16(B) || !al16(C)) return false;
    if (ldc % (c_dt == VITMI_F32 ? 4 : 8)) return false;
    if (res && (ldr % 8 || !al16(res))) return false;
    if (aux && (ldaux % 8 || !al16(aux))) return false;
    return true;
}

// Raise a kernel's dynamic-LDS limit once per device.
static int ensure_lds_attr(const void* fn, size_t bytes, unsigned long long* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vitmi_hip_fail(e, "hipGetDevice");
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return VITMI_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return vitmi_hip_fail(e, "hipFuncSetAttribute(LDS size)");
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return VITMI_OK;
}

template <int EPI, typename OutT, bool PATCH, int BM, int BN, int WM, int WN>
static int launch_nt_cfg(int M, int N, int K, const bf16_t* a, long long lda, const bf16_t* b, long long ldb,
                         const FastEpi& e, const PatchGeom& pg, hipStream_t s) {
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * TK * sizeof(bf16_t);
    auto fn = gemm_nt_fast_kernel<EPI, OutT, PATCH, BM, BN, WM, WN>;
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(fn), lds, &done);
    if (rc != VITMI_OK) return rc;
    const int tiles_m = ceil_div(M, BM), tiles_n = ceil_div(N, BN);
    hipLaunchKernelGGL(fn, dim3(tiles_m * tiles_n), dim3(64 * WM * WN), lds, s, a, lda, b, ldb, e, M, N, K, tiles_n, pg);
    VITMI_LAUNCH_CHECK("vitmi_gemm(nt fast)");
    return VITMI_OK;
}

// Compute units the one-block-per-CU kernels leave free (for the workgroups of a collective that runs beside
// them): a multiple of 8, so every XCD gives up the same number.
static int g_reserved_cus = 0;
extern "C" int vitmi_set_reserved_cus(int k) {
    if (k < 0 || k > 128 || (k & 7)) {
        vitmi_set_error("vitmi_set_reserved_cus: %d is not a multiple of 8 in [0, 128]", k);
        return VITMI_ERR_BAD_ARG;
    }
    g_reserved_cus = k;
    return VITMI_OK;
}

int vitmi_reserved_cus() { return g_reserved_cus; }

// persistent blocks of the phased nt kernel: one per CU, rounded down to a multiple of 8 so that a
// block's tiles stay on its XCD, minus the reserved ones
// every CU of the current device, rounded down to a multiple of 8: what the workspace-size queries plan with, so
// that a size asked for under one reservation holds under every other
static int full_grid_blocks() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cached[dev & 63];
    if (c == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        c = cus & ~7;
    }
    return c;
}
static int phased_grid_blocks() {
    const int c = full_grid_blocks();
    return c - g_reserved_cus > 8 ? c - g_reserved_cus : 8;
}

// row panels per group of the phased kernel's tile walk: -1 = automatic, 0 = n fastest
static int g_nt_group_override = -1;
extern "C" int vitmi_debug_set_nt_group(int gm) {
    g_nt_group_override = gm;
    return VITMI_OK;
}
static int g_nt_span = 0, g_nt_xspan = 0, g_nt_stagger_all = 0;
extern "C" int vitmi_debug_set_nt_stagger(int span_ticks, int xspan_ticks, int all_blocks) {
    g_nt_span = span_ticks;
    g_nt_xspan = xspan_ticks;
    g_nt_stagger_all = all_blocks;
    return VITMI_OK;
}
static int pick_nt_group(int N, int K) {
    if (g_nt_group_override >= 0) return g_nt_group_override;
    (void)K;
    // measured at 50432 rows (tools/gemm_group_cmp.py, tools/gemm_group_pmc.py): N = 3072 (12 column tiles, a 4.7 MB
    // weight that does not fit one XCD's L2) plain 305 -> 243 us and FETCH_SIZE 485 -> 432 MB with groups of 8 (12:
    // 241 us but 507 MB, 16: 587 MB: the group's A panels no longer fit); N = 2304 178 -> 171 us; three column
    // tiles: no difference in time, more traffic
    const int tiles_n = ceil_div(N, 256);
    return tiles_n >= 6 ? 8 : 0;
}

template <int EPI, typename OutT>
static int launch_nt_phased(int M, int N, int K, const bf16_t* a, long long lda, const bf16_t* b, long long ldb,
                            const FastEpi& e, hipStream_t s) {
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * 4 * (size_t)RG * sizeof(bf16_t);
    const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 256);
    const int ntiles = tiles_m * tiles_n;
    int grid = ntiles;
    if (grid > phased_grid_blocks()) grid = phased_grid_blocks();
    if (e.cs_partial) {
        // column sums are produced by the one instantiation that needs them (dGELU -> bf16)
        if (!epi_is_auxmul(EPI) || sizeof(OutT) != 2) {
            vitmi_set_error("vitmi_gemm(nt phased): column sums are only fused into the dGELU / aux-multiply epilogues");
            return VITMI_ERR_BAD_ARG;
        }
        static unsigned long long done_cs = 0;
        auto fcs = gemm_nt_phased_kernel<epi_is_auxmul(EPI) ? EPI : VITMI_EPI_DGELU, bf16_t, true>;
        int rc = ensure_lds_attr(reinterpret_cast<const void*>(fcs), lds, &done_cs);
        if (rc != VITMI_OK) return rc;
        hipLaunchKernelGGL(fcs, dim3(grid), dim3(512), lds, s, a, lda, b, ldb, e, M, N, K, tiles_n, ntiles,
                           pick_nt_group(N, K), g_nt_span, g_nt_xspan, g_nt_stagger_all);
        VITMI_LAUNCH_CHECK("vitmi_gemm(nt phased + column sums)");
        return VITMI_OK;
    }
    auto fn = gemm_nt_phased_kernel<EPI, OutT, false>;
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(fn), lds, &done);
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, s, a, lda, b, ldb, e, M, N, K, tiles_n, ntiles,
                       pick_nt_group(N, K), g_nt_span, g_nt_xspan, g_nt_stagger_all);
    VITMI_LAUNCH_CHECK("vitmi_gemm(nt phased)");
    return VITMI_OK;
}

// tile configuration: 0 = 128x128 (4 waves), 1 = 256x128 (8 waves, 4x2), 2 = 256x256 (8 waves, 2x4),
// 3 = 256x256 phased (K % 64 == 0, not for the patch operand)
static int g_nt_cfg_override = -1;
extern "C" int vitmi_debug_set_nt_config(int cfg) {
    g_nt_cfg_override = cfg;
    return VITMI_OK;
}

// rows of a thin last round on a second launch of the small-tile kernel: 1 = on
static int g_nt_tail_split = 1;
extern "C" int vitmi_debug_set_nt_tail_split(int on) {
    g_nt_tail_split = on ? 1 : 0;
    return VITMI_OK;
}

static int pick_nt_cfg(int M, int N) {
    if (g_nt_cfg_override >= 0) return g_nt_cfg_override;
    // measured on the ViT-B/16 shapes: 256x256 tiles win (1.2-1.3x over 128x128) once they fill the
    // 256 CUs at least once; 256x128 never wins
    const long long tiles = (long long)ceil_div(M, 256) * ceil_div(N, 256);
    return tiles >= 256 ? 3 : 0;  // 3 falls back to 2 when K % 64 != 0
}

template <int EPI, typename OutT, bool PATCH>
static int launch_nt_epi(int M, int N, int K, const bf16_t* a, long long lda, const bf16_t* b, long long ldb,
                         const FastEpi& e, const PatchGeom& pg, hipStream_t s) {
    switch (pick_nt_cfg(M, N)) {
        case 3:
            if (!PATCH && K % TK == 0 && K > 0) {
                // A last round that fills less than half of the chip costs a whole tile time on the persistent
                // kernel (N = 768 at 50432 rows: 591 tiles = 2 rounds + 79).  The rows of that round go to a second
                // launch of the 128 x 128 kernel instead: four times as many, smaller tiles, one or two per compute
                // unit, no partial sums and no extra traffic.  Not with the fused column sums (only the persistent
                // kernel produces them).
                const int G = phased_grid_blocks();
                const int tiles_m = ceil_div(M, 256), tiles_n = ceil_div(N, 256);
                const long long ntiles = (long long)tiles_m * tiles_n;
                const int rounds = (int)(ntiles / G), rem = (int)(ntiles % G);
                const int head_panels = (int)(((long long)rounds * G) / tiles_n);
                if (g_nt_tail_split && !e.cs_partial && e.rows_per_group == 0 && rounds >= 1 && rem > 0 && 2 * rem < G &&
                    head_panels >= 1 && head_panels < tiles_m) {
                    const int M1 = head_panels * 256;
                    int rc = launch_nt_phased<EPI, OutT>(M1, N, K, a, lda, b, ldb, e, s);
                    if (rc != VITMI_OK) return rc;
                    FastEpi t = e;
                    // head-major planes: the rows of the tail start M1 * 64 elements into every plane, the plane
                    // stride stays that of the whole matrix
                    t.C = reinterpret_cast<OutT*>(e.C) + (long long)M1 * (e.hm_T > 0 ? 64 : e.ldc);
                    if (e.res) t.res = e.res + (long long)M1 * e.ldr;
                    if (e.aux) t.aux = e.aux + (long long)M1 * e.ldaux;
                    if (e.ln_mean) t.ln_mean = e.ln_mean + M1, t.ln_rstd = e.ln_rstd + M1;
                    return launch_nt_cfg<EPI, OutT, PATCH, 128, 128, 2, 2>(M - M1, N, K, a + (long long)M1 * lda, lda, b, ldb,
                                                                         t, pg, s);   // (lda = 64 for a head-major A)
                }
                return launch_nt_phased<EPI, OutT>(M, N, K, a, lda, b, ldb, e, s);
            }
            [[fallthrough]];
        case 2: return launch_nt_cfg<EPI, OutT, PATCH, 256, 256, 2, 4>(M, N, K, a, lda, b, ldb, e, pg, s);
        case 1: return launch_nt_cfg<EPI, OutT, PATCH, 256, 128, 4, 2>(M, N, K, a, lda, b, ldb, e, pg, s);
        default: return launch_nt_cfg<EPI, OutT, PATCH, 128, 128, 2, 2>(M, N, K, a, lda, b, ldb, e, pg, s);
    }
}

template <bool PATCH>
static int launch_nt_fast(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                          const FastEpi& e, int c_dt, int epilogue, const PatchGeom& pg, hipStream_t s) {
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* b = (const bf16_t*)B;
    if (c_dt == VITMI_F32) return launch_nt_epi<VITMI_EPI_NONE, float, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
    if (epilogue == VITMI_EPI_NONE)
        return launch_nt_epi<VITMI_EPI_NONE, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
    if constexpr (PATCH) {
        vitmi_set_error("patch embedding supports only the plain epilogue");
        return VITMI_ERR_BAD_ARG;
    } else {
        if (epilogue == VITMI_EPI_GELU)
            return launch_nt_epi<VITMI_EPI_GELU, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
        if (epilogue == VITMI_EPI_GELU_D)
            return launch_nt_epi<VITMI_EPI_GELU_D, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
        if (epilogue == VITMI_EPI_MULAUX)
            return launch_nt_epi<VITMI_EPI_MULAUX, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
        if (epilogue == VITMI_EPI_LN)
            return launch_nt_epi<VITMI_EPI_LN, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
        if (epilogue == VITMI_EPI_LN_GELU)
            return launch_nt_epi<VITMI_EPI_LN_GELU, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
        return launch_nt_epi<VITMI_EPI_DGELU, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
    }
}

// true when vitmi_gemm_nt_fast would run the phased kernel, the one that can also emit column sums;
// rows of the partial matrix it then writes: 2 per 256-row tile
bool vitmi_gemm_nt_phased_ok(int M, int N, int K) { return pick_nt_cfg(M, N) == 3 && K % TK == 0 && K > 0; }
int vitmi_gemm_nt_colsum_rows(int M) { return 2 * ceil_div(M, 256); }

int vitmi_gemm_nt_fast(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int c_dt, int epilogue, const float* bias, const void* residual, long long ldr,
                       void* aux, long long ldaux, float* cs_partial, hipStream_t s) {
    FastEpi e{C, ldc, bias, (const bf16_t*)residual, ldr, (bf16_t*)aux, ldaux, 0, 0, cs_partial, nullptr, nullptr, nullptr, 0, 0};
    return launch_nt_fast<false>(M, N, K, A, lda, B, ldb, e, c_dt, epilogue, PatchGeom{}, s);
}

// The two products around the attention core with qkv / dqkv as head-major planes [3*H][tokens][64] (bf16, plain
// epilogue, every extent a multiple of 8, N resp. K a multiple of 64):
//   out_planes: C = A B^T + bias written as planes (the qkv projection; N = 3*H*64, tokens = M)
//   a_planes:   A is read from planes (the input gradient behind dqkv; K = 3*H*64, tokens = M), C row-major
int vitmi_gemm_nt_fast_planes(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                              long long ldc, const float* bias, int out_planes, int a_planes, hipStream_t s) {
    FastEpi e{C, ldc, bias, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, out_planes ? (long long)M : 0,
              a_planes ? (long long)M * 64 : 0};
    return launch_nt_fast<false>(M, N, K, A, a_planes ? 64 : lda, B, ldb, e, VITMI_BF16, VITMI_EPI_NONE, PatchGeom{}, s);
}

// C (bf16) = epilogue(LayerNorm(x) W^T + b) without the normalised rows ever being formed: A = x, B = the weight
// with gamma folded in, epilogue VITMI_EPI_LN or VITMI_EPI_LN_GELU, bias = W beta + b, ln_s = row sums of B
int vitmi_gemm_nt_fast_ln(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                          long long ldc, int epilogue, const float* bias, const float* ln_mean, const float* ln_rstd,
                          const float* ln_s, hipStream_t s) {
    FastEpi e{C, ldc, bias, nullptr, 0, nullptr, 0, 0, 0, nullptr, ln_mean, ln_rstd, ln_s, 0, 0};
    return launch_nt_fast<false>(M, N, K, A, lda, B, ldb, e, VITMI_BF16, epilogue, PatchGeom{}, s);
}

// patch-embedding forward on a bf16 image: tokens x D = patches x W^T, rows land at
// out + b*out_batch_stride + j*D
bool vitmi_patch_fast_ok(const void* img, const void* w, const void* out, int W, int p, int D, int E,
                         long long out_batch_stride) {
    return p % 8 == 0 && W % 8 == 0 && D % 8 == 0 && E % 8 == 0 && out_batch_stride % 8 == 0 && al16(img) &&
           al16(w) && al16(out);
}

// pw = p for patches whose width is a multiple of 8; pw = 8 > p for the wide form, where D counts the features of
// the whole group (8 / p tokens) and w / bias are the block forms
int vitmi_patch_fwd_fast(const void* img, const void* w, const float* bias, void* out, int B, int C, int H, int W,
                         int p, int pw, int D, long long out_batch_stride, hipStream_t s) {
    const int gh = H / p, gw = W / pw, per = gh * gw, E = C * p * pw;
    PatchGeom pg{C, H, W, p, pw, gw, per, B * per, E};
    FastEpi e{out, D, bias, nullptr, 0, nullptr, 0, per, out_batch_stride, nullptr, nullptr, nullptr, nullptr, 0, 0};
    return launch_nt_fast<true>(B * per, D, E, img, 0, w, E, e, VITMI_BF16, VITMI_EPI_NONE, pg, s);
}

bool vitmi_gemm_tn_fast_ok(int P, int Q, const void* U, long long ldu, const void* V, long long ldv, const void* C,
                           long long ldc) {
    return !(P % 8 || Q % 8 || ldu % 8 || ldv % 8 || ldc % 4 || !al16(U) || !al16(V) || !al16(C));
}

// tile configuration of the tn kernel: 0 = 128x128 (4 waves), 1 = 256x256 (8 waves), 2 = 256x256 phased
static int g_tn_cfg_override = -1;
extern "C" int vitmi_debug_set_tn_config(int cfg) {
    g_tn_cfg_override = cfg;
    return VITMI_OK;
}
static int pick_tn_cfg(int P, int Q) {
    if (g_tn_cfg_override >= 0) return g_tn_cfg_override;
    return (P >= 512 && Q >= 512) ? 2 : 0;  // 2 falls back to 1 when T % 64 != 0
}

// number of t-slabs the tn kernel would use (the caller sizes the workspace from it)
int vitmi_gemm_tn_slabs(int P, int Q, int T) {
    const int cfg = pick_tn_cfg(P, Q);
    const int bt = cfg >= 1 ? 256 : 128;
    const int tiles = ceil_div(P, bt) * ceil_div(Q, bt);
    const int nk = ceil_div(T, TK);
    // blocks that are resident at once on 256 CUs; rounding DOWN keeps the launch to one round of
    // blocks (one block more than the chip holds would double the kernel's time)
    const int target = (cfg >= 1 ? 1 : 2) * phased_grid_blocks();
    int s = target / tiles;
    if (s > nk) s = nk;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

// an upper bound of vitmi_gemm_tn_slabs over every CU reservation and every tile configuration (a debug override
// included): workspace sizes are computed from this, the launch from vitmi_gemm_tn_slabs
int vitmi_gemm_tn_slabs_bound(int P, int Q, int T) {
    const int nk = ceil_div(T, TK);
    int best = 1;
    for (int bt = 128; bt <= 256; bt += 128) {
        const int tiles = ceil_div(P, bt) * ceil_div(Q, bt);
        int s = (bt == 256 ? 1 : 2) * full_grid_blocks() / tiles;
        if (s > nk) s = nk;
        if (s > 32) s = 32;
        if (s > best) best = s;
    }
    return best;
}

template <bool PATCH, int BP, int BQ, int WP, int WQ>
static int launch_tn_cfg(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                         long long ldc, float* slabs, int ns, const PatchGeom& pg, hipStream_t s, long long u_plane = 0) {
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * (size_t)TK * (BP + BQ) * sizeof(bf16_t);
    auto fn = gemm_tn_fast_kernel<PATCH, BP, BQ, WP, WQ>;
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(fn), lds, &done);
    if (rc != VITMI_OK) return rc;
    const int tiles_p = ceil_div(P, BP), tiles_q = ceil_div(Q, BQ);
    if (!slabs) ns = 1;
    const int nk = ceil_div(T, TK);
    const int per = ceil_div(nk, ns);
    ns = ceil_div(nk, per);
    dim3 grid(tiles_p * tiles_q * ns), block(64 * WP * WQ);
    if (ns == 1) {
        hipLaunchKernelGGL(fn, grid, block, lds, s, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv, C, ldc, 0ll, P, Q, T,
                           tiles_q, tiles_p * tiles_q, per, pg, u_plane);
        VITMI_LAUNCH_CHECK("vitmi_gemm(tn fast)");
        return VITMI_OK;
    }
    // slabs are dense [ns][P][ldc] so that the reduce is a flat sum when C is dense
    const long long slab_stride = (long long)P * ldc;
    hipLaunchKernelGGL(fn, grid, block, lds, s, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv, slabs, ldc, slab_stride,
                       P, Q, T, tiles_q, tiles_p * tiles_q, per, pg, u_plane);
    VITMI_LAUNCH_CHECK("vitmi_gemm(tn fast)");
    const long long n4 = slab_stride / 4;
    int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, slab_stride, ns, C, n4);
    VITMI_LAUNCH_CHECK("vitmi_gemm(tn reduce)");
    return VITMI_OK;
}

static int launch_tn_phased(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                            long long ldc, float* slabs, int ns, hipStream_t s, long long u_plane = 0) {
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * 4 * (size_t)TK * 128 * sizeof(bf16_t);
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(gemm_tn_phased_kernel), lds, &done);
    if (rc != VITMI_OK) return rc;
    const int tiles_p = ceil_div(P, 256), tiles_q = ceil_div(Q, 256);
    if (!slabs) ns = 1;
    const int nk = T / TK;
    const int per = ceil_div(nk, ns);
    ns = ceil_div(nk, per);
    dim3 grid(tiles_p * tiles_q * ns), block(512);
    const bool direct = (ns == 1);
    const long long slab_stride = direct ? 0ll : (long long)P * ldc;
    hipLaunchKernelGGL(gemm_tn_phased_kernel, grid, block, lds, s, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv,
                       direct ? C : slabs, ldc, slab_stride, P, Q, tiles_q, tiles_p * tiles_q, per, nk, u_plane);
    VITMI_LAUNCH_CHECK("vitmi_gemm(tn phased)");
    if (!direct) {
        const long long n4 = slab_stride / 4;
        int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, slab_stride, ns, C, n4);
        VITMI_LAUNCH_CHECK("vitmi_gemm(tn reduce)");
    }
    return VITMI_OK;
}

template <bool PATCH>
static int launch_tn_fast(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                          long long ldc, float* slabs, const PatchGeom& pg, hipStream_t s, long long u_plane = 0) {
    const int ns = vitmi_gemm_tn_slabs(P, Q, T);
    if (!PATCH && pick_tn_cfg(P, Q) == 2 && T % TK == 0 && T > 0)
        return launch_tn_phased(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, ns, s, u_plane);
    if (pick_tn_cfg(P, Q) >= 1)
        return launch_tn_cfg<PATCH, 256, 256, 2, 4>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, ns, pg, s, u_plane);
    return launch_tn_cfg<PATCH, 128, 128, 2, 2>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, ns, pg, s, u_plane);
}

// ---- vitmi_gemm_wgrad_multi: jobs over one token axis, TNM_MAX per launch
static int tn_multi_slabs(int tiles, int T) {
    const int nk = ceil_div(T, TK);
    int ns = 2 * phased_grid_blocks() / (tiles > 0 ? tiles : 1);   // two 64-KiB blocks fit a CU
    if (ns > nk) ns = nk;
    if (ns > 32) ns = 32;
    if (ns < 1) ns = 1;
    return ns;
}
static int tn_multi_slabs_bound(int tiles, int T) {
    const int nk = ceil_div(T, TK);
    int ns = 2 * full_grid_blocks() / (tiles > 0 ? tiles : 1);
    if (ns > nk) ns = nk;
    if (ns > 32) ns = 32;
    if (ns < 1) ns = 1;
    return ns;
}
static bool tn_multi_job_ok(const vitmi_wgrad_job& j) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return j.U && j.V && j.C && al(j.U) && al(j.V) && al(j.C) && j.P > 0 && j.Q > 0 && j.P % 8 == 0 && j.Q % 8 == 0 &&
           j.ldu >= j.P && j.ldv >= j.Q && j.ldu % 8 == 0 && j.ldv % 8 == 0 && j.ldc >= j.Q && j.ldc % 4 == 0 &&
           j.ldu < (1ll << 31) && j.ldv < (1ll << 31) && j.ldc < (1ll << 31);
}

extern "C" long long vitmi_gemm_wgrad_multi_workspace_bytes(const vitmi_wgrad_job* jobs, int n, long long T) {
    if (!jobs || n <= 0 || T <= 0 || T >= (1ll << 31)) return 0;
    long long worst = 0;
    for (int b = 0; b < n; b += TNM_MAX) {
        const int m = n - b < TNM_MAX ? n - b : TNM_MAX;
        int tiles = 0;
        long long elems = 0;
        for (int k = 0; k < m; ++k) {
            tiles += ceil_div(jobs[b + k].P, 128) * ceil_div(jobs[b + k].Q, 128);
            elems += (long long)jobs[b + k].P * jobs[b + k].Q;
        }
        const long long bytes = (long long)tn_multi_slabs_bound(tiles, (int)T) * elems * (long long)sizeof(float);
        if (bytes > worst) worst = bytes;
    }
    return worst;
}

extern "C" int vitmi_gemm_wgrad_multi(const vitmi_wgrad_job* jobs, int n, long long T, void* workspace, void* stream) {
    VITMI_CHECK_ARG(n >= 0 && T > 0 && T < (1ll << 31), "vitmi_gemm_wgrad_multi: bad argument");
    if (n == 0) return VITMI_OK;
    VITMI_CHECK_ARG(jobs, "vitmi_gemm_wgrad_multi: null job table");
    for (int k = 0; k < n; ++k)
        VITMI_CHECK_ARG(tn_multi_job_ok(jobs[k]),
                        "vitmi_gemm_wgrad_multi: job %d: bf16 operands and the fp32 result 16-byte aligned, P, Q and the "
                        "leading dimensions multiples of 8 (ldc of 4)", k);
    hipStream_t s = as_stream(stream);
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * (size_t)TK * 256 * sizeof(bf16_t);
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(gemm_tn_multi_kernel), lds, &done);
    if (rc != VITMI_OK) return rc;
    for (int b = 0; b < n; b += TNM_MAX) {
        TnMulti tm{};
        tm.njobs = n - b < TNM_MAX ? n - b : TNM_MAX;
        tm.T = (int)T;
        long long elems = 0;
        double flops = 0;
        int tiles = 0;
        for (int k = 0; k < tm.njobs; ++k) {
            const vitmi_wgrad_job& j = jobs[b + k];
            tm.U[k] = (const bf16_t*)j.U;
            tm.V[k] = (const bf16_t*)j.V;
            tm.C[k] = j.C;
            tm.ldu[k] = (int)j.ldu;
            tm.ldv[k] = (int)j.ldv;
            tm.ldc[k] = (int)j.ldc;
            tm.P[k] = j.P;
            tm.Q[k] = j.Q;
            tm.tiles_q[k] = ceil_div(j.Q, 128);
            tm.tile_begin[k] = tiles;
            tm.slab_off[k] = elems;     // scaled by the slab count below
            tiles += ceil_div(j.P, 128) * tm.tiles_q[k];
            elems += (long long)j.P * j.Q;
            flops += 2.0 * j.P * j.Q * (double)T;
        }
        for (int k = tm.njobs; k <= TNM_MAX; ++k) tm.tile_begin[k] = tiles;
        tm.tile_begin[tm.njobs] = tiles;
        const int nk = ceil_div((int)T, TK);
        int ns = workspace ? tn_multi_slabs(tiles, (int)T) : 1;
        tm.per = ceil_div(nk, ns);
        ns = ceil_div(nk, tm.per);
        tm.ns = ns;
        for (int k = 0; k < tm.njobs; ++k) tm.slab_off[k] *= ns;
        tm.slabs = (float*)workspace;
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, flops);
        hipLaunchKernelGGL(gemm_tn_multi_kernel, dim3(tiles * ns), dim3(256), lds, s, tm);
        VITMI_LAUNCH_CHECK("vitmi_gemm_wgrad_multi");
        if (ns > 1) {
            hipLaunchKernelGGL(tn_multi_reduce_kernel, dim3(64, tm.njobs), dim3(256), 0, s, tm);
            VITMI_LAUNCH_CHECK("vitmi_gemm_wgrad_multi(reduce)");
        }
    }
    return VITMI_OK;
}

// per-segment tile sums of squares: partial[seg][tile]; returns the number of tiles per segment through *tiles
template <int BT, int WP, int WQ>
static int launch_tn_seg(int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                         float* partial, int* tiles, hipStream_t s) {
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * (size_t)TK * (BT + BT) * sizeof(bf16_t) + 2 * WP * WQ * sizeof(float);
    auto fn = gemm_tn_fast_kernel<false, BT, BT, WP, WQ, true>;
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(fn), lds, &done);
    if (rc != VITMI_OK) return rc;
    const int tiles_p = ceil_div(P, BT), tiles_q = ceil_div(Q, BT), nseg = ceil_div(T, seg);
    const int ntiles = tiles_p * tiles_q;
    *tiles = ntiles;
    // one round of blocks (two per CU for the 128-wide tiles): each block walks `per` consecutive segments
    const int want = (BT == 256 ? 256 : 512) / ntiles;
    const int per = ceil_div(nseg, want < 1 ? 1 : want);
    const int groups = ceil_div(nseg, per);
    hipLaunchKernelGGL(fn, dim3(ntiles * groups), dim3(64 * WP * WQ), lds, s, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv,
                       partial, 0ll, 0ll, P, Q, T, tiles_q, ntiles, seg, PatchGeom{}, 0ll);
    VITMI_LAUNCH_CHECK("vitmi_gemm_wgrad_sqnorm(tn segments)");
    return VITMI_OK;
}
int vitmi_gemm_tn_seg_tiles(int P, int Q) {
    const int bt = (P >= 512 && Q >= 512) ? 256 : 128;
    return ceil_div(P, bt) * ceil_div(Q, bt);
}
int vitmi_gemm_tn_seg_sq(int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                         float* partial, int* tiles, hipStream_t s) {
    if (P >= 512 && Q >= 512) return launch_tn_seg<256, 2, 4>(P, Q, T, seg, U, ldu, V, ldv, partial, tiles, s);
    return launch_tn_seg<128, 2, 2>(P, Q, T, seg, U, ldu, V, ldv, partial, tiles, s);
}

// per-image || U_i^T V_i ||_F^2 ADDED to sqnorm[i] (images = segments of `seg` consecutive rows, seg <= 256): the Gram form
bool vitmi_gemm_ghost_sq_ok(int P, int Q, int seg, const void* U, long long ldu, const void* V, long long ldv) {
    return seg >= 1 && seg <= 256 && P % 8 == 0 && Q % 8 == 0 && ldu % 8 == 0 && ldv % 8 == 0 && al16(U) && al16(V);
}
template <int TPW>
static int launch_ghost(int P, int Q, int T, int seg, int nt, const void* U, long long ldu, const void* V, long long ldv,
                        float* sqnorm, hipStream_t s) {
    static unsigned long long done = 0;
    const size_t lds = 2 * (size_t)nt * 16 * TK * sizeof(bf16_t);
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(ghost_sqnorm_kernel<TPW>), 2 * (size_t)16 * 16 * TK * sizeof(bf16_t), &done);
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL(ghost_sqnorm_kernel<TPW>, dim3(ceil_div(T, seg)), dim3(1024), lds, s, (const bf16_t*)U, ldu, P,
                       (const bf16_t*)V, ldv, Q, T, seg, nt, sqnorm);
    VITMI_LAUNCH_CHECK("vitmi_wgrad_sqnorm(gram)");
    return VITMI_OK;
}
int vitmi_gemm_ghost_sq(int P, int Q, int T, int seg, const void* U, long long ldu, const void* V, long long ldv,
                        float* sqnorm, hipStream_t s) {
    const int nt = ceil_div(seg, 16);
    const int tpw = ceil_div(nt * (nt + 1) / 2, 16);
    switch (tpw) {
        case 1: return launch_ghost<1>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 2: return launch_ghost<2>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 3: return launch_ghost<3>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 4: return launch_ghost<4>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 5: return launch_ghost<5>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 6: return launch_ghost<6>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 7: return launch_ghost<7>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        case 8: return launch_ghost<8>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
        default: return launch_ghost<9>(P, Q, T, seg, nt, U, ldu, V, ldv, sqnorm, s);
    }
}

int vitmi_gemm_tn_fast(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                       long long ldc, float* slabs, hipStream_t s) {
    return launch_tn_fast<false>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, PatchGeom{}, s);
}
// the same with U as head-major planes [P / 64][T][64] (P a multiple of 64)
int vitmi_gemm_tn_fast_planes(int P, int Q, int T, const void* U, const void* V, long long ldv, float* C, long long ldc,
                              float* slabs, hipStream_t s) {
    return launch_tn_fast<false>(P, Q, T, U, 64, V, ldv, C, ldc, slabs, PatchGeom{}, s, (long long)T * 64);
}

// ---- several weight gradients over one token axis in one launch (gemm_tn_batch_kernel)
static int g_tn_batch = 1;
extern "C" int vitmi_debug_set_tn_batch(int on) {
    g_tn_batch = on;
    return VITMI_OK;
}

// geometry of the launch; false when the shapes do not qualify (the caller then runs the products one by one)
static bool tn_batch_plan(int n, const int* P, const int* Q, int T, TnBatch* tb) {
    if (!g_tn_batch || n < 2 || n > 4 || T <= 0 || T % TK) return false;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (P[i] <= 0 || Q[i] <= 0 || P[i] % 256 || Q[i] % 256) return false;
        tb->pr[i].Q = Q[i];
        tb->pr[i].tiles_q = Q[i] / 256;
        tb->pr[i].tile_begin = tiles;
        tb->pr[i].pad = 0;
        tiles += (P[i] / 256) * (Q[i] / 256);
    }
    const int G = phased_grid_blocks(), nk = T / TK;
    const int nmain = G / tiles;
    if (nmain < 1 || nk < 4 * (nmain + 1) || (long long)tiles * nk >= (1ll << 31)) return false;   // unit indices are ints
    const int rem = G - nmain * tiles;
    int a = ceil_div(nk, nmain), r = 0, upl = 1;
    // Blocks left over take the last r token tiles of m whole output tiles each, so that all of them walk the same
    // token tiles at the same time (neighbouring output tiles then share operand panels in L2, as the main blocks
    // do; runs that start inside an output tile were measured slower than they are shorter).  Worth it when the
    // left-over blocks are a fair fraction of the chip and r is more than a few token tiles.
    if (rem * 4 >= tiles) {
        const int m = ceil_div(tiles, rem);
        int best = a;
        const int r0 = nk / (nmain * m + 1);
        for (int rr = r0 - 1; rr <= r0 + 1; ++rr) {
            if (rr < 8 || rr >= nk) continue;
            const int c = ceil_div(nk - rr, nmain);
            const int worst = c > m * rr ? c : m * rr;
            if (worst < best) best = worst, a = c, r = rr, upl = m * rr;
        }
    }
    tb->nprob = n;
    tb->tiles = tiles;
    tb->nk = nk;
    tb->nmain = nmain;
    tb->a = a;
    tb->r = r;
    tb->upl = upl;
    tb->kmain = nk - r;
    tb->spt = nmain + (r > 0 ? 1 : 0);
    tb->main_blocks = nmain * tiles;
    return true;
}

// bytes of partial tiles the batched launch needs; 0 = these shapes run one by one
long long vitmi_gemm_tn_batch_bytes(int n, const int* P, const int* Q, int T) {
    TnBatch tb;
    if (!tn_batch_plan(n, P, Q, T, &tb)) return 0;
    return (long long)tb.tiles * tb.spt * TN_SLOT * (long long)sizeof(float);
}

// an upper bound of vitmi_gemm_tn_batch_bytes over every CU reservation and both states of the batch switch: a tile
// never gets more main slabs than full chip / tiles, plus one slot for a remainder run
long long vitmi_gemm_tn_batch_bytes_bound(int n, const int* P, const int* Q, int T) {
    if (n < 2 || n > 4 || T <= 0 || T % TK) return 0;
    long long tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (P[i] <= 0 || Q[i] <= 0 || P[i] % 256 || Q[i] % 256) return 0;
        tiles += (long long)(P[i] / 256) * (Q[i] / 256);
    }
    const long long nmain = full_grid_blocks() / tiles;
    if (nmain < 1) return 0;
    return tiles * (nmain + 1) * TN_SLOT * (long long)sizeof(float);
}

// C[i] (dense [P[i]][Q[i]] fp32) = U[i]^T V[i] over T tokens for i < n; VITMI_ERR_UNSUPPORTED when the shapes do
// not qualify
int vitmi_gemm_tn_batch(int n, const int* P, const int* Q, int T, const void* const* U, const long long* ldu,
                        const void* const* V, const long long* ldv, float* const* C, float* partial, hipStream_t s,
                        const int* u_planes) {
    TnBatch tb;
    if (!partial || !tn_batch_plan(n, P, Q, T, &tb)) return VITMI_ERR_UNSUPPORTED;
    for (int i = 0; i < n; ++i) {
        if (!vitmi_gemm_tn_fast_ok(P[i], Q[i], U[i], ldu[i], V[i], ldv[i], C[i], Q[i])) return VITMI_ERR_UNSUPPORTED;
        tb.pr[i].u_plane = (u_planes && u_planes[i]) ? (long long)T * 64 : 0;
        tb.pr[i].U = (const bf16_t*)U[i];
        tb.pr[i].V = (const bf16_t*)V[i];
        tb.pr[i].C = C[i];
        tb.pr[i].ldu = ldu[i];
        tb.pr[i].ldv = ldv[i];
        tb.pr[i].ldc = Q[i];
    }
    tb.partial = partial;
    static unsigned long long done = 0;
    constexpr size_t lds = 2 * 4 * (size_t)TK * 128 * sizeof(bf16_t);
    int rc = ensure_lds_attr(reinterpret_cast<const void*>(gemm_tn_batch_kernel), lds, &done);
    if (rc != VITMI_OK) return rc;
    const int runs = tb.r > 0 ? ceil_div(tb.tiles * tb.r, tb.upl) : 0;
    hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3(tb.main_blocks + runs), dim3(512), lds, s, tb);
    VITMI_LAUNCH_CHECK("vitmi_gemm_wgrad_batch(tn phased)");
    hipLaunchKernelGGL(tn_batch_reduce_kernel, dim3(tb.tiles * 16), dim3(256), 0, s, tb);
    VITMI_LAUNCH_CHECK("vitmi_gemm_wgrad_batch(reduce)");
    return VITMI_OK;
}

// patch-embedding weight gradient on a bf16 image: dW[d][e] = sum_t dy[t][d] * P[t][e]
int vitmi_patch_bwd_w_fast(const void* img, const void* dy, float* dw, float* slabs, int B, int C, int H, int W, int p,
                           int pw, int D, hipStream_t s) {
    const int gh = H / p, gw = W / pw, per = gh * gw, E = C * p * pw;
    PatchGeom pg{C, H, W, p, pw, gw, per, B * per, E};
    return launch_tn_fast<true>(D, E, B * per, dy, D, img, 0, dw, E, slabs, pg, s);
}

extern "C" int vitmi_transpose_cast(const void* src, int src_dt, void* dst, int rows, int cols, void* stream) {
    VITMI_CHECK_ARG(src && dst && rows > 0 && cols > 0, "vitmi_transpose_cast: bad argument");
    hipStream_t s = as_stream(stream);
    dim3 grid(ceil_div(cols, 32), ceil_div(rows, 32)), block(256);
    if (src_dt == VITMI_F32)
        hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, block, 0, s, (const float*)src, (bf16_t*)dst, rows, cols);
    else if (src_dt == VITMI_BF16)
        hipLaunchKernelGGL(transpose_cast_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, rows,
                           cols);
    else {
        vitmi_set_error("vitmi_transpose_cast: bad dtype %d", src_dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_transpose_cast");
    return VITMI_OK;
}
