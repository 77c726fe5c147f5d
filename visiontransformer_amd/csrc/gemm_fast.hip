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
template <int ROWS, int NW>
__device__ __forceinline__ void stage_nt(const bf16_t* __restrict__ g, long long ld, int row0, int rows, int k0, int K,
                                         bf16_t* lds, int wave, int lane) {
    constexpr int PER = ROWS / 8 / NW;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int rblk = (wave * PER + i) * 8;
        const int r = rblk + (lane >> 3);
        const int c = (lane & 7) ^ swz_nt(r);      // logical chunk this lane fetches
        const int gr = row0 + r, gk = k0 + c * 8;
        const void* src = (gr < rows && gk < K) ? (const void*)(g + (long long)gr * ld + gk) : (const void*)g_zero16;
        glds16(src, lds + rblk * TK);
    }
}

// Patch operand: logical matrix P[token m][element e], m = (image b, gy, gx), e = (c, i, j), read in
// place from the NCHW image: P = x[b, c, gy*ps + i, gx*ps + j].  With ps % 8 == 0 a 16-byte chunk
// (8 elements) never leaves one patch row, so it is 16 contiguous bytes of the image.
struct PatchGeom {
    int C, H, W, ps, gw, per;  // per = patches per image
    int tokens, elems;         // B*per, C*ps*ps
};
__device__ __forceinline__ long long patch_addr(const PatchGeom& g, int m, int e) {
    const int b = m / g.per, t = m - b * g.per;
    const int gy = t / g.gw, gx = t - gy * g.gw;
    const int c = e / (g.ps * g.ps), q = e - c * g.ps * g.ps;
    const int i = q / g.ps, j = q - i * g.ps;
    return (((long long)b * g.C + c) * g.H + (gy * g.ps + i)) * g.W + gx * g.ps + j;
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
};

template <int EPI, typename OutT>
__device__ __forceinline__ void epi_row16(const FastEpi& e, int m, int n, int N, float (&v)[16]) {
    // v[t] belongs to column n + t of row m
    const bool full = (n + 16 <= N);
    if (e.bias) {
        if (full) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + n + 4 * t4);
#pragma unroll
                for (int t = 0; t < 4; ++t) v[4 * t4 + t] += b[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (n + t < N) v[t] += e.bias[n + t];
        }
    }
    if (EPI == VITMI_EPI_GELU) {
        if (e.aux) {
            bf16_t* a = e.aux + (long long)m * e.ldaux + n;
            if (full) {
                bf16x8 p0, p1;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    p0[t] = (bf16_t)v[t];
                    p1[t] = (bf16_t)v[8 + t];
                }
                *reinterpret_cast<bf16x8*>(a) = p0;
                *reinterpret_cast<bf16x8*>(a + 8) = p1;
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (n + t < N) a[t] = (bf16_t)v[t];
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = gelu_fast(v[t]);
    } else if (EPI == VITMI_EPI_DGELU) {
        const bf16_t* a = e.aux + (long long)m * e.ldaux + n;
        if (full) {
            const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(a);
            const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(a + 8);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] *= gelu_grad_fast((float)p0[t]);
                v[8 + t] *= gelu_grad_fast((float)p1[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (n + t < N) v[t] *= gelu_grad_fast((float)a[t]);
        }
    }
    if (e.res) {
        const bf16_t* r = e.res + (long long)m * e.ldr + n;
        if (full) {
            const bf16x8 p0 = *reinterpret_cast<const bf16x8*>(r);
            const bf16x8 p1 = *reinterpret_cast<const bf16x8*>(r + 8);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] += (float)p0[t];
                v[8 + t] += (float)p1[t];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (n + t < N) v[t] += (float)r[t];
        }
    }
    OutT* c = reinterpret_cast<OutT*>(e.C) + (long long)m * e.ldc + n;
    if (e.rows_per_group > 0) {
        const int gi = m / e.rows_per_group;
        c = reinterpret_cast<OutT*>(e.C) + (long long)gi * e.group_stride +
            (long long)(m - gi * e.rows_per_group) * e.ldc + n;
    }
    if (full) {
        if (sizeof(OutT) == 2) {
            bf16x8 p0, p1;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                p0[t] = (bf16_t)v[t];
                p1[t] = (bf16_t)v[8 + t];
            }
            *reinterpret_cast<bf16x8*>(c) = p0;
            *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(c) + 8) = p1;
        } else {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(c) + 4 * t4) =
                    f32x4{v[4 * t4], v[4 * t4 + 1], v[4 * t4 + 2], v[4 * t4 + 3]};
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (n + t < N) c[t] = from_f32<OutT>(v[t]);
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
    if (PATCH)
        stage_nt_patch<BM, NW>(A, pg, m0, 0, smem, wave, lane);
    else
        stage_nt<BM, NW>(A, lda, m0, M, 0, K, smem, wave, lane);
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
                stage_nt<BM, NW>(A, lda, m0, M, (kt + 1) * TK, K, An, wave, lane);
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
// tn kernel (weight gradients)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz_tn(int t) { return ((t & 3) | (((t >> 3) & 1) << 2)) << 1; }

// stage one [64 t][COLS] tile with NW waves: one load = 1 KiB = 512/COLS rows
template <int COLS, int NW>
__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ g, long long ld, int t0, int T, int col0, int cols,
                                         bf16_t* lds, int wave, int lane) {
    constexpr int CPR = COLS / 8;          // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;          // rows per load instruction
    constexpr int PER = (64 / RPI) / NW;   // loads per wave
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int tblk = (wave * PER + i) * RPI;
        const int t = tblk + lane / CPR;
        const int c = (lane % CPR) ^ swz_tn(t);
        const int gt = t0 + t, gc = col0 + c * 8;
        const void* src = (gt < T && gc < cols) ? (const void*)(g + (long long)gt * ld + gc) : (const void*)g_zero16;
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
template <bool PATCH, int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(64 * WP * WQ) void gemm_tn_fast_kernel(const bf16_t* __restrict__ U, long long ldu,
                                                                    const bf16_t* __restrict__ V, long long ldv,
                                                                    float* __restrict__ C, long long ldc,
                                                                    long long slab_stride, int P, int Q, int T,
                                                                    int tiles_q, int tiles_pq, int kt_per_slab,
                                                                    PatchGeom pg) {
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
    const int nk_all = (T + TK - 1) / TK;
    const int kt_begin = slab * kt_per_slab;
    const int kt_end = min(kt_begin + kt_per_slab, nk_all);

    f32x4 acc[NJ][MI];  // [j: q-tile][i: p-tile]
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (kt_begin < kt_end) {
        stage_tn<BP, NW>(U, ldu, kt_begin * TK, T, p0, P, smem, wave, lane);
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
            stage_tn<BP, NW>(U, ldu, (kt + 1) * TK, T, p0, P, Un, wave, lane);
            if (PATCH)
                stage_tn_patch<BQ, NW>(V, pg, (kt + 1) * TK, q0, Un + TK * BP, wave, lane);
            else
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

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, long long slab_stride, int nslabs,
                                   float* __restrict__ out, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + 4 * i);
        for (int k = 1; k < nslabs; ++k) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + k * slab_stride + 4 * i);
            s += v;
        }
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

// tile configuration: 0 = 128x128 (4 waves), 1 = 256x128 (8 waves, 4x2), 2 = 256x256 (8 waves, 2x4)
static int g_nt_cfg_override = -1;
extern "C" int vitmi_debug_set_nt_config(int cfg) {
    g_nt_cfg_override = cfg;
    return VITMI_OK;
}

static int pick_nt_cfg(int M, int N) {
    if (g_nt_cfg_override >= 0) return g_nt_cfg_override;
    // measured on the ViT-B/16 shapes: 256x256 tiles win (1.2-1.3x over 128x128) once they fill the
    // 256 CUs at least once; 256x128 never wins
    const long long tiles = (long long)ceil_div(M, 256) * ceil_div(N, 256);
    return tiles >= 256 ? 2 : 0;
}

template <int EPI, typename OutT, bool PATCH>
static int launch_nt_epi(int M, int N, int K, const bf16_t* a, long long lda, const bf16_t* b, long long ldb,
                         const FastEpi& e, const PatchGeom& pg, hipStream_t s) {
    switch (pick_nt_cfg(M, N)) {
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
    if (epilogue == VITMI_EPI_GELU)
        return launch_nt_epi<VITMI_EPI_GELU, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
    return launch_nt_epi<VITMI_EPI_DGELU, bf16_t, PATCH>(M, N, K, a, lda, b, ldb, e, pg, s);
}

int vitmi_gemm_nt_fast(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int c_dt, int epilogue, const float* bias, const void* residual, long long ldr,
                       void* aux, long long ldaux, hipStream_t s) {
    FastEpi e{C, ldc, bias, (const bf16_t*)residual, ldr, (bf16_t*)aux, ldaux, 0, 0};
    return launch_nt_fast<false>(M, N, K, A, lda, B, ldb, e, c_dt, epilogue, PatchGeom{}, s);
}

// patch-embedding forward on a bf16 image: tokens x D = patches x W^T, rows land at
// out + b*out_batch_stride + j*D
bool vitmi_patch_fast_ok(const void* img, const void* w, const void* out, int W, int p, int D, int E,
                         long long out_batch_stride) {
    return p % 8 == 0 && W % 8 == 0 && D % 8 == 0 && E % 8 == 0 && out_batch_stride % 8 == 0 && al16(img) &&
           al16(w) && al16(out);
}

int vitmi_patch_fwd_fast(const void* img, const void* w, const float* bias, void* out, int B, int C, int H, int W,
                         int p, int D, long long out_batch_stride, hipStream_t s) {
    const int gh = H / p, gw = W / p, per = gh * gw, E = C * p * p;
    PatchGeom pg{C, H, W, p, gw, per, B * per, E};
    FastEpi e{out, D, bias, nullptr, 0, nullptr, 0, per, out_batch_stride};
    return launch_nt_fast<true>(B * per, D, E, img, 0, w, E, e, VITMI_BF16, VITMI_EPI_NONE, pg, s);
}

bool vitmi_gemm_tn_fast_ok(int P, int Q, const void* U, long long ldu, const void* V, long long ldv, const void* C,
                           long long ldc) {
    return !(P % 8 || Q % 8 || ldu % 8 || ldv % 8 || ldc % 4 || !al16(U) || !al16(V) || !al16(C));
}

// tile configuration of the tn kernel: 0 = 128x128 (4 waves), 1 = 256x256 (8 waves)
static int g_tn_cfg_override = -1;
extern "C" int vitmi_debug_set_tn_config(int cfg) {
    g_tn_cfg_override = cfg;
    return VITMI_OK;
}
static int pick_tn_cfg(int P, int Q) {
    if (g_tn_cfg_override >= 0) return g_tn_cfg_override;
    return (P >= 512 && Q >= 512) ? 1 : 0;
}

// number of t-slabs the tn kernel would use (the caller sizes the workspace from it)
int vitmi_gemm_tn_slabs(int P, int Q, int T) {
    const int cfg = pick_tn_cfg(P, Q);
    const int bt = cfg == 1 ? 256 : 128;
    const int tiles = ceil_div(P, bt) * ceil_div(Q, bt);
    const int nk = ceil_div(T, TK);
    const int target = cfg == 1 ? 256 : 512;  // resident blocks on 256 CUs
    int s = (target + tiles - 1) / tiles;
    if (s > nk) s = nk;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

template <bool PATCH, int BP, int BQ, int WP, int WQ>
static int launch_tn_cfg(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                         long long ldc, float* slabs, int ns, const PatchGeom& pg, hipStream_t s) {
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
                           tiles_q, tiles_p * tiles_q, per, pg);
        VITMI_LAUNCH_CHECK("vitmi_gemm(tn fast)");
        return VITMI_OK;
    }
    // slabs are dense [ns][P][ldc] so that the reduce is a flat sum when C is dense
    const long long slab_stride = (long long)P * ldc;
    hipLaunchKernelGGL(fn, grid, block, lds, s, (const bf16_t*)U, ldu, (const bf16_t*)V, ldv, slabs, ldc, slab_stride,
                       P, Q, T, tiles_q, tiles_p * tiles_q, per, pg);
    VITMI_LAUNCH_CHECK("vitmi_gemm(tn fast)");
    const long long n4 = slab_stride / 4;
    int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, slabs, slab_stride, ns, C, n4);
    VITMI_LAUNCH_CHECK("vitmi_gemm(tn reduce)");
    return VITMI_OK;
}

template <bool PATCH>
static int launch_tn_fast(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                          long long ldc, float* slabs, const PatchGeom& pg, hipStream_t s) {
    const int ns = vitmi_gemm_tn_slabs(P, Q, T);
    if (pick_tn_cfg(P, Q) == 1)
        return launch_tn_cfg<PATCH, 256, 256, 2, 4>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, ns, pg, s);
    return launch_tn_cfg<PATCH, 128, 128, 2, 2>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, ns, pg, s);
}

int vitmi_gemm_tn_fast(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                       long long ldc, float* slabs, hipStream_t s) {
    return launch_tn_fast<false>(P, Q, T, U, ldu, V, ldv, C, ldc, slabs, PatchGeom{}, s);
}

// patch-embedding weight gradient on a bf16 image: dW[d][e] = sum_t dy[t][d] * P[t][e]
int vitmi_patch_bwd_w_fast(const void* img, const void* dy, float* dw, float* slabs, int B, int C, int H, int W, int p,
                           int D, hipStream_t s) {
    const int gh = H / p, gw = W / p, per = gh * gw, E = C * p * p;
    PatchGeom pg{C, H, W, p, gw, per, B * per, E};
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
