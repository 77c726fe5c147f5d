// Device helpers shared by the one-launch sub-block kernels of the small geometry (small.hip, attention.hip): MFMA
// products with the activation operand in LDS and the weight operand streamed from L2 into fragments, and the
// LayerNorm of a strip of rows into LDS.
//
// MFMA operand convention (v_mfma_f32_16x16x32_bf16, D = A B + C, A: 16 x 32, B: 32 x 16):
//   A fragment: lane l holds A[l & 15][8 * (l >> 4) .. +8]      -> 16 weight rows (output features), k contiguous
//   B fragment: lane l holds B[8 * (l >> 4) .. +8][l & 15]      -> 16 token rows read from LDS, k contiguous
//   D fragment: lane l holds D[4 * (l >> 4) + i][l & 15], i<4   -> 4 output features of ONE token row
// i.e. the products are formed transposed (features x tokens), so that a lane ends up with consecutive elements of a
// row of the row-major result.
#pragma once
#include "common.h"

namespace {

constexpr int SM_THREADS = 512;
constexpr int SM_WAVES = SM_THREADS / 64;
constexpr int SM_PAD = 8;  // bf16 elements between LDS rows: consecutive rows start 4 banks apart

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// acc[j][m] += Wtile_j[0 .. 16)[0 .. 32 KS) * act[16 m .. +16][0 .. 32 KS)^T for j < NB feature tiles and m < MT token tiles.
// The NB tiles interleave the 16 NB weight rows n0 .. n0 + 16 NB: row r of tile j is weight row
// n0 + 4 NB (r >> 2) + 4 j + (r & 3), so that the lane with l >> 4 == g ends up with the 4 NB CONSECUTIVE output features
// n0 + 4 NB g .. + 4 NB of its token row (acc[j][m][i] = feature n0 + 4 NB g + 4 j + i): one 16-byte store per token tile
// for NB = 2.  W is row-major in global memory (ldw elements per row, 16-byte aligned rows), act in LDS (lda elements
// per row).  The weight fragments arrive in chunks of CH k-steps through two register buffers: a wave has the loads of
// one or two chunks in flight while it multiplies.
// One 16-byte global load per lane whose completion the COMPILER does not track: the loop below issues a chunk's loads
// two chunks ahead of their use and waits with counted s_waitcnt.  (Written as ordinary loads, the scheduler sinks every
// load to its first use and the loop pays one L2 latency per chunk: 0.6 us per k-step, seven times the MFMA time.)
__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// waits until at most N vector-memory operations are outstanding and makes every fragment of `buf` depend on the wait
template <int N, int CH, int NB>
__device__ __forceinline__ void gwait(u32x4 (&buf)[CH][NB]) {
    static_assert(CH * NB <= 8, "one asm operand per fragment");
    if constexpr (CH * NB == 2)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(buf[0][0]), "+v"(buf[CH - 1][NB - 1]) : "n"(N));
    else if constexpr (CH * NB == 4)
        asm volatile("s_waitcnt vmcnt(%4)"
                     : "+v"(buf[0][0]), "+v"(buf[0][NB - 1]), "+v"(buf[CH - 1][0]), "+v"(buf[CH - 1][NB - 1])
                     : "n"(N));
    else if constexpr (CH * NB == 6)
        asm volatile("s_waitcnt vmcnt(%6)"
                     : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1])
                     : "n"(N));
    else
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1]),
                       "+v"(buf[3][0]), "+v"(buf[3][1])
                     : "n"(N));
}

// the same for a 16-byte LDS read (address = byte offset inside LDS) ...
__device__ __forceinline__ void lread16(u32x4& dst, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
// ... and its counted wait over the MT fragments of one k-step
template <int N, int MT>
__device__ __forceinline__ void lwait(u32x4 (&av)[MT]) {
    static_assert(MT >= 2 && MT <= 6, "one asm operand per fragment");
    if constexpr (MT == 2)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(av[0]), "+v"(av[1]) : "n"(N));
    else if constexpr (MT == 3)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]) : "n"(N));
    else if constexpr (MT == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]) : "n"(N));
    else if constexpr (MT == 5)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]) : "n"(N));
    else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5])
                     : "n"(N));
}

template <int MT, int NB, int CH>
__device__ __forceinline__ void rows_gemm_ch(const bf16_t* __restrict__ w, int ldw, int n0, int KS, const bf16_t* act, int lda,
                                             int lane, f32x4 (&acc)[NB][MT]) {
    static_assert(NB == 2, "gwait lists the fragments of two feature tiles");
    // KS is a multiple of CH.  Every fetch is unconditional (past the end it re-reads the last chunk); only the products
    // are guarded.  Loads return in order, so "at most CH * NB outstanding" after the NEXT chunk's loads were issued
    // means the current chunk has arrived.
    const int r = lane & 15;
    const bf16_t* wp = w + (long long)(n0 + 4 * NB * (r >> 2) + (r & 3)) * ldw + (lane >> 4) * 8;
    typedef __attribute__((address_space(3))) const unsigned char* lds_u8p;
    const unsigned ap = (unsigned)(size_t)(lds_u8p)(const unsigned char*)act + (unsigned)((r * lda + (lane >> 4) * 8) * sizeof(bf16_t));
    u32x4 b0[CH][NB], b1[CH][NB];
    auto fetch = [&](u32x4 (&buf)[CH][NB], int ks0) {
        ks0 = ks0 < KS ? ks0 : KS - CH;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int j = 0; j < NB; ++j) gload16(buf[c][j], wp + (long long)j * 4 * ldw + (ks0 + c) * 32);
    };
    // the activation fragments of k-step c + 1 are read from LDS while the products of k-step c are issued
    auto multiply = [&](const u32x4 (&buf)[CH][NB], int ks0) {
        u32x4 av[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) lread16(av[0][m], ap + (unsigned)((m * 16 * lda + ks0 * 32) * sizeof(bf16_t)));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c + 1 < CH) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    lread16(av[(c + 1) & 1][m], ap + (unsigned)((m * 16 * lda + (ks0 + c + 1) * 32) * sizeof(bf16_t)));
                lwait<MT, MT>(av[c & 1]);
            } else {
                lwait<0, MT>(av[c & 1]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[j][m] = mfma16(__builtin_bit_cast(bf16x8, buf[c][j]), __builtin_bit_cast(bf16x8, av[c & 1][m]), acc[j][m]);
        }
    };
    fetch(b0, 0);
    for (int ks = 0; ks < KS; ks += 2 * CH) {
        fetch(b1, ks + CH);
        gwait<CH * NB>(b0);
        multiply(b0, ks);
        fetch(b0, ks + 2 * CH);
        gwait<CH * NB>(b1);
        if (ks + CH < KS) multiply(b1, ks + CH);
    }
    gwait<0>(b0);  // the trailing re-read: nothing of this routine stays in flight
}

// the chunk depth that divides the k-step count: up to CHMAX steps (2 * CHMAX * NB 1-KB requests in flight per wave)
template <int MT, int NB, int CHMAX>
__device__ __forceinline__ void rows_gemm(const bf16_t* __restrict__ w, int ldw, int n0, int KS, const bf16_t* act, int lda,
                                          int lane, f32x4 (&acc)[NB][MT]) {
    if (CHMAX >= 4 && KS % 4 == 0)
        rows_gemm_ch<MT, NB, (CHMAX >= 4 ? 4 : 1)>(w, ldw, n0, KS, act, lda, lane, acc);
    else if (CHMAX >= 3 && KS % 3 == 0)
        rows_gemm_ch<MT, NB, (CHMAX >= 3 ? 3 : 1)>(w, ldw, n0, KS, act, lda, lane, acc);
    else if (KS % 2 == 0)
        rows_gemm_ch<MT, NB, 2>(w, ldw, n0, KS, act, lda, lane, acc);
    else
        rows_gemm_ch<MT, NB, 1>(w, ldw, n0, KS, act, lda, lane, acc);
}

// LayerNorm of the strip's rows into LDS (and to xn / mean / rstd): a wave per row, four columns per lane.  Same
// arithmetic as layernorm_fwd_kernel (two passes over the registers).  Rows past the end of the matrix become zeros.
// All of the wave's rows are requested before the first is reduced, and the reductions of the rows are independent
// chains the scheduler interleaves: one memory latency and one reduction latency per strip, not one per row.
template <int MT>
__device__ __forceinline__ void strip_layernorm(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, bf16_t* __restrict__ xn,
                                                float* __restrict__ mean, float* __restrict__ rstd, long long row0, int nrows,
                                                int D, float eps, bf16_t* xs, int lds_ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane * 4;
    const bool on = c < D;
    const int cc = on ? c : 0;
    constexpr int RPW = MT * 16 / SM_WAVES;
    bf16x4 raw[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + i * SM_WAVES;
        raw[i] = *reinterpret_cast<const bf16x4*>(x + (row0 + (r < nrows ? r : 0)) * D + cc);
    }
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cc);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + cc);
    float v[RPW][4], mu[RPW], rs[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = on ? (float)raw[i][j] : 0.f;
        mu[i] = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < RPW; ++i) mu[i] += __shfl_xor(mu[i], o, 64);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        mu[i] = mu[i] / (float)D;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mu[i];
            q += d * d;
        }
        rs[i] = on ? q : 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < RPW; ++i) rs[i] += __shfl_xor(rs[i], o, 64);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave + i * SM_WAVES;
        rs[i] = rsqrtf(rs[i] / (float)D + eps);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = r < nrows ? (bf16_t)((v[i][j] - mu[i]) * rs[i] * g[j] + bt[j]) : (bf16_t)0.f;
        if (on) *reinterpret_cast<bf16x4*>(xs + r * lds_ld + c) = o;
        if (r < nrows) {
            if (on) *reinterpret_cast<bf16x4*>(xn + (row0 + r) * D + c) = o;
            if (lane == 0) {
                mean[row0 + r] = mu[i];
                rstd[row0 + r] = rs[i];
            }
        }
    }
}

}  // namespace
