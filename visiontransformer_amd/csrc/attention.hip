// Multi-head self-attention core for ViT sequence lengths (N <= 256 tokens, and up to 608 for bf16 heads of 64:
// all keys of one head fit in LDS and the scores of a query row fit in registers, so the softmax is a single exact
// pass over the whole row -- the online rescale of a streaming flash kernel never triggers at these lengths).
//
// bf16 kernels (v_mfma_f32_16x16x32_bf16), one workgroup per (image, head):
//   forward : K and V^T of the head are staged once in LDS.  A wave owns 16 query rows and
//             computes S^T = K Q^T with the QUERY on the MFMA lane, so the softmax reduction is
//             over that lane's registers plus two cross-lane steps, and the probability
//             registers are directly the B operand of O^T = V^T P^T.  Key rows are fed to the
//             first product in a permuted order chosen so that those registers come out in
//             natural key order.  The bf16-rounded exponentials feed P V and the result is scaled by
//             1 / sum; the optional export to HBM writes the normalised probabilities, and the
//             output is the same bit for bit with and without it.
//   backward: two kernels that both recompute P from (Q, K, lse):
//             dK/dV -- a wave owns 16 keys, sweeps all queries, accumulates dK^T, dV^T in
//                      registers (S, dP computed with the KEY on the lane -> P, dS are B operands);
//             dQ    -- a wave owns 16 queries, sweeps all keys, accumulates dQ^T
//                      (S^T, dP^T with the QUERY on the lane -> dS^T is the B operand).
//             No atomics, no cross-wave reduction: results are bitwise reproducible.
// fp32 kernels: one wave per query (or key) row, plain FMA; the parity path.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int ATTN_THREADS = 512;  // 8 waves per (image, head) workgroup
constexpr size_t LDS_BYTES = 160 * 1024;

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
}

// Running maximum over the eight scores of one chunk in four v_max3_f32.  fmaxf would cost three times as many
// instructions: MFMA results are not known to the compiler as quiet, so it copies each through a canonicalising
// v_max first.  The compiler does not track MFMA-result hazards into inline assembly: callers finish every MFMA
// that feeds these blocks first, fence with sched_barrier, and let the FIRST block spend the wait states.
template <bool FIRST>
__device__ __forceinline__ float max8_raw(float m, const f32x4& a, const f32x4& b) {
    if (FIRST) asm volatile("s_nop 7\n\ts_nop 7");
    asm volatile(
        "v_max3_f32 %0, %0, %1, %2\n\t"
        "v_max3_f32 %0, %0, %3, %4\n\t"
        "v_max3_f32 %0, %0, %5, %6\n\t"
        "v_max3_f32 %0, %0, %7, %8"
        : "+v"(m)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    return m;
}
// eight floats to a bf16 fragment, pairwise: four v_cvt_pk_bf16_f32, no byte shuffles
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    union {
        bf16x2 h[4];
        bf16x8 v;
    } u;
    u.h[0] = __builtin_convertvector(f32x2{a[0], a[1]}, bf16x2);
    u.h[1] = __builtin_convertvector(f32x2{a[2], a[3]}, bf16x2);
    u.h[2] = __builtin_convertvector(f32x2{b[0], b[1]}, bf16x2);
    u.h[3] = __builtin_convertvector(f32x2{b[2], b[3]}, bf16x2);
    return u.v;
}

// the first n elements of a fragment, the others zeroed (whatever they held): n <= 0 clears it, n >= 8 keeps it
__device__ __forceinline__ bf16x8 keep_first(const bf16x8& v, int n) {
    union {
        bf16x8 v;
        u32x4 w;
    } u;
    u.v = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) u.w[i] &= (n > 2 * i ? 0xFFFFu : 0u) | (n > 2 * i + 1 ? 0xFFFF0000u : 0u);
    return u.v;
}

// permuted row -> sequence index inside a 32-wide chunk: the MFMA output register r of lane
// group g (rows 4g+r) of sub-tile u then holds sequence index 32c + 8g + 4u + r.
__device__ __forceinline__ int perm_row(int c, int u, int i) { return 32 * c + 8 * (i >> 2) + 4 * u + (i & 3); }

template <int DH> struct AttnDims {
    static constexpr int KS = DH >= 32 ? DH / 32 : 1;   // 32-wide k-steps over the head dim
    static constexpr int DT = DH >= 16 ? DH / 16 : 1;   // 16-wide output tiles over the head dim
    static constexpr int W = DH;                        // row width of an LDS image
    static constexpr int CPR = W / 8;                   // 16-byte chunks per image row
};

// One LDS image per matrix, [seq][W] bf16 without padding, serves BOTH kinds of MFMA operand:
//   row reads   (ds_read_b128: 8 consecutive head-dim elements of one sequence row) and
//   transposed reads (ds_read_b64_tr_b16: 4 consecutive sequence rows of one head-dim column).
// For 128-byte rows (dh = 64) chunk c of row r is stored at chunk c ^ swz(r) with
// swz(r) = 2*bit1(r) | 4*bit3(r): the 16-lane groups of a row read (rows taken in the permuted
// order of perm_row) and the 32-lane halves of a transposed read (rows 8g + q, g in {0,1}) then
// touch every LDS bank exactly once.  Rows of 16 bytes (dh = 8) need no swizzle: the 16 rows of a row read and the
// 2 x 8 rows of a transposed read lie in distinct banks as they are.  A transposed read always covers 16 head-dim
// columns; with dh = 8 its upper eight come from the NEXT row of the image (16 bytes past the last row for the
// last one, which the launchers allocate): they only feed output rows d >= 8, which are never stored.
template <int DH>
__device__ __forceinline__ int img_swz(int row) {
    return AttnDims<DH>::W == 64 ? ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)) : 0;
}
template <int DH>
__device__ __forceinline__ int img_off(int row, int chunk) {
    return row * AttnDims<DH>::W + ((chunk ^ img_swz<DH>(row)) << 3);
}

// 16-byte fragment of row `row` of a [seq][DH] matrix in HBM (row stride ld), head-dim offset
// ks*32 + 8*g; zero outside the matrix.
template <int DH>
__device__ __forceinline__ bf16x8 global_frag(const bf16_t* base, long long ld, int row, int nrows, int ks, int g) {
    const int d = ks * 32 + g * 8;
    if (row < nrows && d < DH) return *reinterpret_cast<const bf16x8*>(base + (long long)row * ld + d);
    return zero8();
}
// row read: elements [ks*32 + 8g, +8) of image row `row`
template <int DH>
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* img, int row, int ks, int g) {
    const int d = ks * 32 + g * 8;
    if (d < DH) return *reinterpret_cast<const bf16x8*>(img + img_off<DH>(row, d >> 3));
    return zero8();
}
// transposed read: element j of lane (fr, fg) = image[seq0 + 8*fg + j][col0 + fr], j = 0..7
template <int DH>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int seq0, int col0, int lane) {
    const int fg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 4 * p;
    const int r_lo = seq0 + 8 * fg + q, r_hi = r_lo + 4;
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img + img_off<DH>(r_lo, col >> 3) + (col & 7)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(img + img_off<DH>(r_hi, col >> 3) + (col & 7)));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// stage rows [0,NP) of HPW adjacent heads of two HBM matrices (a head = DH columns; row strides lda, ldb) into
// their images, head hh at img + hh * NP * W; rows >= n are zero.  Every load of the thread is issued before the
// first LDS write, so a workgroup pays one HBM round trip for its operands rather than one per 16-byte chunk, and
// consecutive threads take consecutive 16-byte pieces of a row across the heads.  blockDim.x = THREADS.
template <int DH, int NP, int THREADS = ATTN_THREADS, int HPW = 1>
__device__ __forceinline__ void stage_img2(const bf16_t* a, long long lda, bf16_t* img_a, const bf16_t* b, long long ldb,
                                           bf16_t* img_b, int n) {
    using A = AttnDims<DH>;
    constexpr int PER_ROW = A::CPR * HPW;
    constexpr int TOT = NP * PER_ROW;
    constexpr int IT = (TOT + THREADS - 1) / THREADS;
    bf16x8 va[IT], vb[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = (int)threadIdx.x + i * THREADS;
        const int r = c / PER_ROW, col = (c % PER_ROW) * 8;
        va[i] = zero8();
        vb[i] = zero8();
        if (c < TOT && r < n) {
            va[i] = *reinterpret_cast<const bf16x8*>(a + (long long)r * lda + col);
            vb[i] = *reinterpret_cast<const bf16x8*>(b + (long long)r * ldb + col);
        }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int c = (int)threadIdx.x + i * THREADS;
        const int r = c / PER_ROW, rem = c % PER_ROW, hh = rem / A::CPR, kc = rem % A::CPR;
        if (c < TOT) {
            *reinterpret_cast<bf16x8*>(img_a + hh * NP * A::W + img_off<DH>(r, kc)) = va[i];
            *reinterpret_cast<bf16x8*>(img_b + hh * NP * A::W + img_off<DH>(r, kc)) = vb[i];
        }
    }
}

// Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  This returns the (image, head) pair index a block
// works on such that each XCD walks a contiguous range of pairs: the heads of one image, whose q/k/v rows are
// adjacent 128-byte pieces of the same qkv rows, are then fetched by one XCD's L2 close together in time.
__device__ __forceinline__ int xcd_block_order() {
    const int G = gridDim.x, x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = G >> 3, rem = G & 7;
    return x * q + (x < rem ? x : rem) + slot;
}

// elements of one wave's export tile buffer: 16 rows of N probabilities plus the alignment shift, in 16-byte units
__host__ __device__ __forceinline__ int export_tile_elems(int N) { return (16 * N + 7 + 7) & ~7; }

// One 16-query tile of the forward pass against the staged K and V images: scores, exact softmax, P V, the lse and
// output rows, and (EXPORT) the probability tile.  qf holds the tile's query fragments.
template <int DH, int NC, bool EXPORT>
__device__ __forceinline__ void attn_fwd_tile(const bf16_t* Ks, const bf16_t* Vs, const bf16x8 (&qf)[AttnDims<DH>::KS], int qt,
                                              int b, int h, int N, int H, float scale, bf16_t* __restrict__ out,
                                              float* __restrict__ lse, bf16_t* __restrict__ probs, bf16_t* Pw, int lane) {
    using A = AttnDims<DH>;
    const int fr = lane & 15, fg = lane >> 4;
    const int I = H * DH;
    const int q = qt * 16 + fr;
    const float sl2 = scale * LOG2E;
    constexpr int nc_act = NC;                        // the launcher picks NC = ceil(N / 32): every chunk holds a real key
    // S^T tiles: s[c][u][r] = score(query q, key 32c + 8fg + 4u + r)
    f32x4 s[NC][2];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < nc_act) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int krow = perm_row(c, u, fr);
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(Ks, krow, ks, fg), qf[ks], acc, 0, 0,
                                                                  0);
                s[c][u] = acc;
            }
            // up to 8 chunks NC = ceil(N / 32) and only the last chunk can reach past N; the long-sequence sizes
            // (12, 16, 19 chunks) may hold several such chunks
            if ((NC > 8 || c == NC - 1) && 32 * c + 32 > N) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (32 * c + 8 * fg + 4 * u + r >= N) s[c][u][r] = -INFINITY;
            }
            if (c & 1) __builtin_amdgcn_sched_barrier(0);   // keep the K fragment reads of later chunks from piling up in registers
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    mx = max8_raw<true>(mx, s[0][0], s[0][1]);
#pragma unroll
    for (int c = 1; c < NC; ++c) mx = max8_raw<false>(mx, s[c][0], s[c][1]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mb = mx * sl2;
    float sum = 0.f;
    bf16x8 pfr[NC];                                   // without the export the exponentials are rounded as they appear
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < nc_act) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[c][u][r], sl2, -mb));
                    s[c][u][r] = p;
                    sum += p;
                }
            if (!EXPORT) pfr[c] = pack8(s[c][0], s[c][1]);
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    if (fg == 0 && q < N) lse[((long long)b * H + h) * N + q] = mx * scale + __logf(sum);
    // the tile's first element in the export tensor; the LDS copy starts at the same offset modulo 16 bytes,
    // so that 16-byte pieces of the tile are aligned on both sides
    bf16_t* gtile = EXPORT ? probs + (((long long)b * H + h) * N + qt * 16) * N : nullptr;
    const int mis = EXPORT ? (int)((reinterpret_cast<uintptr_t>(gtile) >> 1) & 7) : 0;
    bf16_t* ptile = Pw + mis;
    // O^T = V^T P^T, d-tile by d-tile; V^T fragments come transposed out of the V row image
    f32x4 o[A::DT];
#pragma unroll
    for (int dt = 0; dt < A::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < nc_act) {
            // the rounded exponentials feed P V and the 16 x DH result is scaled by 1 / sum afterwards (16 products
            // per lane instead of one per score); the export rounds the normalised values on its own
            const bf16x8 pf = EXPORT ? pack8(s[c][0], s[c][1]) : pfr[c];
            if (EXPORT) {
                const bf16x8 pn = pack8(s[c][0] * inv, s[c][1] * inv);
                bf16_t* prow = ptile + fr * N + 32 * c + 8 * fg;
                if (32 * c + 32 <= N) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) prow[j] = pn[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (32 * c + 8 * fg + j < N) prow[j] = pn[j];
                }
            }
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Vs, 32 * c, dt * 16, lane), pf, o[dt],
                                                                0, 0, 0);
            if (c & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (EXPORT) {
        const int total = (N - qt * 16 < 16 ? N - qt * 16 : 16) * N;   // rows beyond N are left out
        const int head = (8 - mis) & 7;                               // elements up to the first 16-byte boundary
        if (lane < head && lane < total) gtile[lane] = ptile[lane];
        const int pieces = total > head ? (total - head) >> 3 : 0;
        for (int i = lane; i < pieces; i += 64)
            *reinterpret_cast<bf16x8*>(gtile + head + 8 * i) = *reinterpret_cast<const bf16x8*>(ptile + head + 8 * i);
        const int done = head + 8 * pieces;
        if (done + lane < total) gtile[done + lane] = ptile[done + lane];
    }
    if (q < N) {
        bf16_t* orow = out + ((long long)b * N + q) * I + h * DH;
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) {
            const int d = dt * 16 + fg * 4;
            if (d < DH) {
                bf16x4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)(o[dt][r] * inv);
                *reinterpret_cast<bf16x4*>(orow + d) = pk;
            }
        }
    }
}

// ================================================================ forward (bf16)
// EXPORT is a template flag of the same kernel.  Chunks of 32 keys that lie wholly beyond N are skipped, only the chunk
// that straddles N pays for the key mask; exponentials are raw v_exp_f32.
// HPW > 1: a workgroup takes HPW adjacent heads of one image (short sequences of narrow heads: one pair alone is a
// few kilobytes and five tiles, less than a workgroup's launch and its HBM round trip are worth) and its waves walk
// the HPW * ceil(N/16) tiles; H must be a multiple of HPW.
template <int DH, int NC, bool EXPORT, int THREADS, int HPW>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(NC <= 8 ? 4 : (NC <= 16 ? 2 : 1), NC <= 8 ? 4 : (NC <= 16 ? 2 : 1)))) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, bf16_t* __restrict__ probs, int N,
                                                            int H, float scale) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);     // [HPW] K images, then [HPW] V images
    bf16_t* Vs = Ks + HPW * IMG;
    const int pair0 = xcd_block_order() * HPW;
    const int b = pair0 / H, h0 = pair0 % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const bf16_t* qb = qkv + (long long)b * N * ld + h0 * DH;
    const bf16_t* kb = qb + I;
    const bf16_t* vb = qb + 2 * I;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntq = (N + 15) >> 4, total = HPW * ntq;
    // export: the wave's 16 x N probability tile is assembled in LDS in its final layout (the 16 rows are one
    // contiguous run of the [B,H,N,N] tensor) and written out with 16-byte stores
    bf16_t* Pw = Vs + HPW * IMG + wave * export_tile_elems(N);
    bf16x8 qf[A::KS], qnext[A::KS];                   // the wave's first query tile travels beside the K, V staging
    auto load_q = [&](int t) {                        // tile t = (head t / ntq, rows 16 (t % ntq) ..); zeros past the end
        const int hh = HPW == 1 ? 0 : t / ntq, qt = t - hh * ntq;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks)
            qnext[ks] = global_frag<DH>(qb + hh * DH, ld, t < total ? qt * 16 + fr : N, N, ks, fg);
    };
    load_q(wave);
    stage_img2<DH, NP, THREADS, HPW>(kb, ld, Ks, vb, ld, Vs, N);
    __syncthreads();

    for (int t = wave; t < total; t += nwaves) {
        const int hh = HPW == 1 ? 0 : t / ntq, qt = t - hh * ntq;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) qf[ks] = qnext[ks];
        load_q(t + nwaves);                           // the next tile's queries load under this tile's arithmetic
        attn_fwd_tile<DH, NC, EXPORT>(Ks + hh * IMG, Vs + hh * IMG, qf, qt, b, h0 + hh, N, H, scale, out, lse, probs, Pw, lane);
    }
}

// ================================================================ backward dK, dV (bf16)
template <int DH, int NC, int HPW>
__global__ __launch_bounds__(ATTN_THREADS) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta_in,
                                                                bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Qs_all = reinterpret_cast<bf16_t*>(smem);        // [HPW] Q images, then [HPW] dO images
    bf16_t* dOs_all = Qs_all + HPW * IMG;
    float* lse2_all = reinterpret_cast<float*>(dOs_all + HPW * IMG);   // [HPW][NP], then delta [HPW][NP]
    float* delta_all = lse2_all + HPW * NP;

    const int pair0 = xcd_block_order() * HPW;
    const int b = pair0 / H, h0 = pair0 % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const bf16_t* qb = qkv + (long long)b * N * ld + h0 * DH;
    const bf16_t* kb = qb + I;
    const bf16_t* vb = qb + 2 * I;
    const bf16_t* dob = dout + (long long)b * N * I + h0 * DH;
    // lse (in log2 units) and delta = rowsum(dO * O) per query, the latter left by the dQ kernel; padded queries
    // get p = 0
    for (int i = threadIdx.x; i < HPW * NP; i += blockDim.x) {
        const int hh = i / NP, q = i - hh * NP;
        float l2 = 1e30f, dl = 0.f;
        if (q < N) {
            l2 = lse[((long long)b * H + h0 + hh) * N + q] * LOG2E;
            dl = delta_in[((long long)b * H + h0 + hh) * N + q];
        }
        lse2_all[i] = l2;
        delta_all[i] = dl;
    }
    stage_img2<DH, NP, ATTN_THREADS, HPW>(qb, ld, Qs_all, dob, I, dOs_all, N);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const float sl2 = scale * LOG2E;
    constexpr int nc_act = NC;                        // the launcher picks NC = ceil(N / 32)
    const int ntk = (N + 15) >> 4;
    for (int t = wave; t < HPW * ntk; t += nwaves) {
        const int hh = HPW == 1 ? 0 : t / ntk, kt = t - hh * ntk;
        const int h = h0 + hh;
        const bf16_t* Qs = Qs_all + hh * IMG;
        const bf16_t* dOs = dOs_all + hh * IMG;
        const float* lse2 = lse2_all + hh * NP;
        const float* delta = delta_all + hh * NP;
        const int key = kt * 16 + fr;
        bf16x8 kf[A::KS], vf[A::KS];
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            kf[ks] = global_frag<DH>(kb + hh * DH, ld, key, N, ks, fg);
            vf[ks] = global_frag<DH>(vb + hh * DH, ld, key, N, ks, fg);
        }
        f32x4 dkt[A::DT], dvt[A::DT];
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) dkt[dt] = dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < nc_act; ++c) {          // query chunks wholly beyond N contribute nothing
            bf16x8 pf, dsf;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qrow = perm_row(c, u, fr);
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dpa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) {
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(Qs, qrow, ks, fg), kf[ks], sa, 0, 0, 0);
                    dpa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(dOs, qrow, ks, fg), vf[ks], dpa, 0, 0, 0);
                }
                // this lane's rows are queries 32c + 8fg + 4u + r
                const int q0 = 32 * c + 8 * fg + 4 * u;
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(&lse2[q0]);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(&delta[q0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, -l2[r]));
                    pf[4 * u + r] = (bf16_t)p;
                    dsf[4 * u + r] = (bf16_t)(p * (dpa[r] - dl[r]));     // the 1/sqrt(dh) goes onto the finished dK
                }
            }
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(dOs, 32 * c, dt * 16, lane), pf, dvt[dt], 0,
                                                                  0, 0);
                dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Qs, 32 * c, dt * 16, lane), dsf, dkt[dt], 0,
                                                                  0, 0);
            }
        }
        if (key < N) {
            bf16_t* dk = dqkv + ((long long)b * N + key) * ld + I + h * DH;
            bf16_t* dv = dk + I;
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                const int d = dt * 16 + fg * 4;
                if (d < DH) {
                    bf16x4 a, c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a[r] = (bf16_t)(dkt[dt][r] * scale);
                        c[r] = (bf16_t)dvt[dt][r];
                    }
                    *reinterpret_cast<bf16x4*>(dk + d) = a;
                    *reinterpret_cast<bf16x4*>(dv + d) = c;
                }
            }
        }
    }
}

// ================================================================ backward dQ (bf16)
template <int DH, int NC, int HPW>
__global__ __launch_bounds__(ATTN_THREADS) __attribute__((amdgpu_waves_per_eu(NC <= 8 ? 4 : 2, NC <= 8 ? 4 : 2))) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                               const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse,
                                                               float* __restrict__ delta_out,
                                                               bf16_t* __restrict__ dqkv, int N, int H, float scale) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks_all = reinterpret_cast<bf16_t*>(smem);        // [HPW] K images, then [HPW] V images
    bf16_t* Vs_all = Ks_all + HPW * IMG;

    const int pair0 = xcd_block_order() * HPW;
    const int b = pair0 / H, h0 = pair0 % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const bf16_t* qb = qkv + (long long)b * N * ld + h0 * DH;
    const bf16_t* kb = qb + I;
    const bf16_t* vb = qb + 2 * I;
    const bf16_t* ob = out + (long long)b * N * I + h0 * DH;
    const bf16_t* dob = dout + (long long)b * N * I + h0 * DH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntq = (N + 15) >> 4, total = HPW * ntq;
    // the wave's first query tile (q, dO, O, lse) travels beside the K, V staging; later tiles load one tile ahead
    bf16x8 qf[A::KS], dof[A::KS], qn[A::KS], don[A::KS], on[A::KS];
    float lsen;
    auto load_tile = [&](int t) {                     // tile t = (head t / ntq, rows 16 (t % ntq) ..); zeros past the end
        const int hh = HPW == 1 ? 0 : t / ntq, qt = t - hh * ntq;
        const int q = t < total ? qt * 16 + fr : N;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            qn[ks] = global_frag<DH>(qb + hh * DH, ld, q, N, ks, fg);
            don[ks] = global_frag<DH>(dob + hh * DH, I, q, N, ks, fg);
            on[ks] = global_frag<DH>(ob + hh * DH, I, q, N, ks, fg);
        }
        lsen = q < N ? lse[((long long)b * H + h0 + hh) * N + q] : 1e30f;
    };
    load_tile(wave);
    stage_img2<DH, NP, ATTN_THREADS, HPW>(kb, ld, Ks_all, vb, ld, Vs_all, N);
    __syncthreads();

    const float sl2 = scale * LOG2E;
    constexpr int nc_act = NC;                        // the launcher picks NC = ceil(N / 32)
    for (int t = wave; t < total; t += nwaves) {
        const int hh = HPW == 1 ? 0 : t / ntq, qt = t - hh * ntq;
        const int h = h0 + hh;
        const bf16_t* Ks = Ks_all + hh * IMG;
        const bf16_t* Vs = Vs_all + hh * IMG;
        const int q = qt * 16 + fr;
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            qf[ks] = qn[ks];
            dof[ks] = don[ks];
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)on[ks][j] * (float)dof[ks][j];
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (fg == 0 && q < N) delta_out[((long long)b * H + h) * N + q] = dl;   // the dK, dV kernel reads it
        // p * 1/sqrt(dh) comes straight out of the exponential: log2 of the factor joins the exponent
        const float l2 = q < N ? lsen * LOG2E - __log2f(scale) : 1e30f;
        load_tile(t + nwaves);
        f32x4 dqt[A::DT];
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c = 0; c < nc_act; ++c) {
            f32x4 ds[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int krow = perm_row(c, u, fr);
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dpa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) {
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(Ks, krow, ks, fg), qf[ks], sa, 0, 0, 0);
                    dpa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(Vs, krow, ks, fg), dof[ks], dpa, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[u][r] = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, -l2)) * (dpa[r] - dl);
            }
            // element j of the fragment is key 32c + 8fg + j
            bf16x8 dsf = pack8(ds[0], ds[1]);
            if (32 * c + 32 > N) dsf = keep_first(dsf, N - 32 * c - 8 * fg);   // keys past N contribute nothing
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt)
                dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Ks, 32 * c, dt * 16, lane), dsf, dqt[dt], 0,
                                                                  0, 0);
        }
        if (q < N) {
            bf16_t* dq = dqkv + ((long long)b * N + q) * ld + h * DH;
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                const int d = dt * 16 + fg * 4;
                if (d < DH) {
                    bf16x4 a;
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = (bf16_t)dqt[dt][r];
                    *reinterpret_cast<bf16x4*>(dq + d) = a;
                }
            }
        }
    }
}

// ================================================================ fp32 path
// forward: block per (b,h), K and V in LDS, one wave per query row.
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           float* __restrict__ lse, float* __restrict__ probs, int N,
                                                           int H, int DH, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ks = reinterpret_cast<float*>(smem);        // [N][DH+1]
    float* Vs = Ks + (long long)N * (DH + 1);          // [N][DH+1]
    float* qs = Vs + (long long)N * (DH + 1);          // [4][DH]
    float* ps = qs + 4 * DH;                           // [4][N]
    const int blk = xcd_block_order();
    const int b = blk / H, h = blk % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float* qb = qkv + (long long)b * N * ld + h * DH;
    for (int c = threadIdx.x; c < N * DH; c += blockDim.x) {
        const int r = c / DH, d = c % DH;
        Ks[r * (DH + 1) + d] = qb[(long long)r * ld + I + d];
        Vs[r * (DH + 1) + d] = qb[(long long)r * ld + 2 * I + d];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myq = qs + wave * DH;
    float* myp = ps + wave * N;
    for (int q = wave; q < N; q += 4) {
        for (int d = lane; d < DH; d += 64) myq[d] = qb[(long long)q * ld + d];
        __builtin_amdgcn_wave_barrier();
        float sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = lane + 64 * t;
            float s = -INFINITY;
            if (j < N) {
                s = 0.f;
                for (int d = 0; d < DH; ++d) s = fmaf(myq[d], Ks[j * (DH + 1) + d], s);
                s *= scale;
            }
            sc[t] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sc[t] = (lane + 64 * t < N) ? expf(sc[t] - mx) : 0.f;
            sum += sc[t];
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = lane + 64 * t;
            if (j < N) {
                const float p = sc[t] * inv;
                myp[j] = p;
                if (probs) probs[(((long long)b * H + h) * N + q) * N + j] = p;
            }
        }
        if (lane == 0) lse[((long long)b * H + h) * N + q] = mx + logf(sum);
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < DH; d += 64) {
            float o = 0.f;
            for (int j = 0; j < N; ++j) o = fmaf(myp[j], Vs[j * (DH + 1) + d], o);
            out[((long long)b * N + q) * I + h * DH + d] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// backward pass 1: per query row -> dq, and P, dS rows into the workspace
__global__ __launch_bounds__(256) void attn_bwd_q_f32_kernel(const float* __restrict__ qkv,
                                                             const float* __restrict__ dout,
                                                             const float* __restrict__ lse, float* __restrict__ dqkv,
                                                             float* __restrict__ Pw, float* __restrict__ dSw, int N,
                                                             int H, int DH, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + (long long)N * (DH + 1);
    float* qs = Vs + (long long)N * (DH + 1);   // [4][2*DH]: q row then dO row
    float* ds = qs + 4 * 2 * DH;                // [4][N]
    const int blk = xcd_block_order();
    const int b = blk / H, h = blk % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float* qb = qkv + (long long)b * N * ld + h * DH;
    const float* dob = dout + (long long)b * N * I + h * DH;
    for (int c = threadIdx.x; c < N * DH; c += blockDim.x) {
        const int r = c / DH, d = c % DH;
        Ks[r * (DH + 1) + d] = qb[(long long)r * ld + I + d];
        Vs[r * (DH + 1) + d] = qb[(long long)r * ld + 2 * I + d];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myq = qs + wave * 2 * DH;
    float* mydo = myq + DH;
    float* myds = ds + wave * N;
    for (int q = wave; q < N; q += 4) {
        for (int d = lane; d < DH; d += 64) {
            myq[d] = qb[(long long)q * ld + d];
            mydo[d] = dob[(long long)q * I + d];
        }
        __builtin_amdgcn_wave_barrier();
        const float l = lse[((long long)b * H + h) * N + q];
        float p[4], dp[4];
        float dsum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = lane + 64 * t;
            p[t] = dp[t] = 0.f;
            if (j < N) {
                float s = 0.f, g = 0.f;
                for (int d = 0; d < DH; ++d) {
                    s = fmaf(myq[d], Ks[j * (DH + 1) + d], s);
                    g = fmaf(mydo[d], Vs[j * (DH + 1) + d], g);
                }
                p[t] = expf(s * scale - l);
                dp[t] = g;
                dsum += p[t] * g;
            }
        }
        dsum = wave_sum(dsum);
        const long long rowoff = (((long long)b * H + h) * N + q) * N;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = lane + 64 * t;
            if (j < N) {
                const float dsv = p[t] * (dp[t] - dsum) * scale;
                myds[j] = dsv;
                Pw[rowoff + j] = p[t];
                dSw[rowoff + j] = dsv;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < DH; d += 64) {
            float o = 0.f;
            for (int j = 0; j < N; ++j) o = fmaf(myds[j], Ks[j * (DH + 1) + d], o);
            dqkv[((long long)b * N + q) * ld + h * DH + d] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// backward pass 2: per key row -> dk, dv from the stored P and dS columns
__global__ __launch_bounds__(256) void attn_bwd_kv_f32_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ dout,
                                                              const float* __restrict__ Pw,
                                                              const float* __restrict__ dSw, float* __restrict__ dqkv,
                                                              int N, int H, int DH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Qs = reinterpret_cast<float*>(smem);     // [N][DH+1]
    float* dOs = Qs + (long long)N * (DH + 1);      // [N][DH+1]
    const int blk = xcd_block_order();
    const int b = blk / H, h = blk % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float* qb = qkv + (long long)b * N * ld + h * DH;
    const float* dob = dout + (long long)b * N * I + h * DH;
    for (int c = threadIdx.x; c < N * DH; c += blockDim.x) {
        const int r = c / DH, d = c % DH;
        Qs[r * (DH + 1) + d] = qb[(long long)r * ld + d];
        dOs[r * (DH + 1) + d] = dob[(long long)r * I + d];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long headoff = ((long long)b * H + h) * N * N;
    for (int j = wave; j < N; j += 4) {
        for (int d = lane; d < DH; d += 64) {
            float dk = 0.f, dv = 0.f;
            for (int q = 0; q < N; ++q) {
                const float p = Pw[headoff + (long long)q * N + j];
                const float s = dSw[headoff + (long long)q * N + j];
                dv = fmaf(p, dOs[q * (DH + 1) + d], dv);
                dk = fmaf(s, Qs[q * (DH + 1) + d], dk);
            }
            dqkv[((long long)b * N + j) * ld + I + h * DH + d] = dk;
            dqkv[((long long)b * N + j) * ld + 2 * I + h * DH + d] = dv;
        }
    }
}

// ================================================================ gradient arriving at the exported probabilities
// One (image, head): dS[q][j] = P[q][j] * (dP[q][j] - sum_j' P[q][j'] dP[q][j']) * scale with P recomputed from the
// fp32 scores S = q k^T and the row's log-sum-exp (softmax backward for a gradient dP that reaches the probabilities
// from outside the fused kernel: a loss on the attention maps, gradient rollout).  One wave per query row.
template <typename T>
__global__ __launch_bounds__(256) void attn_dprobs_rows_kernel(const float* __restrict__ S, const float* __restrict__ lse,
                                                               const T* __restrict__ dP, T* __restrict__ dS, int N,
                                                               float scale) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= N) return;
    const float l = lse[q];
    float dot = 0.f;
    for (int j = lane; j < N; j += 64) dot += expf(S[(long long)q * N + j] * scale - l) * to_f32(dP[(long long)q * N + j]);
    dot = wave_sum(dot);
    for (int j = lane; j < N; j += 64) {
        const float p = expf(S[(long long)q * N + j] * scale - l);
        dS[(long long)q * N + j] = from_f32<T>(p * (to_f32(dP[(long long)q * N + j]) - dot) * scale);
    }
}

// ---------------------------------------------------------------- dispatch helpers
inline int pick_nc(int N) {
    const int need = (N + 31) / 32;
    // 1..8 exact (every chunk holds a real key); 12, 16, 19 chunks for sequences up to 608 tokens, where the whole
    // head still fits in LDS (one workgroup per CU) and chunks that lie beyond N are masked like the straddling one
    const int avail[] = {1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 19};
    for (int a : avail)
        if (need <= a) return a;
    return 0;
}

template <int DH, int NC, int HPW = 1>
size_t dkv_smem() {
    return (size_t)HPW * ((size_t)(2 * NC * 32 * AttnDims<DH>::W) * sizeof(bf16_t) + 2 * NC * 32 * sizeof(float)) + 16;
}
template <int DH, int NC, int HPW = 1>
size_t dq_smem() {
    return (size_t)HPW * (2 * NC * 32 * AttnDims<DH>::W) * sizeof(bf16_t) + 16;
}

// Raise a kernel's dynamic-LDS limit once per device (the attribute is per device; one process
// may drive several devices from several threads).
inline int ensure_lds(const void* fn, size_t bytes, unsigned long long* done_mask, const char* who) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return vitmi_hip_fail(e, who);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return VITMI_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return vitmi_hip_fail(e, who);
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    return VITMI_OK;
}

template <int DH, int NC, int HPW = 1>
size_t fwd_smem() {
    return (size_t)HPW * (2 * NC * 32 * AttnDims<DH>::W) * sizeof(bf16_t) + 16;
}

// heads per workgroup: 6 for short sequences (<= 96 tokens) of narrow heads (<= 16 wide) when the head count allows
// it and the grid still covers the chip; otherwise one (image, head) pair per workgroup
constexpr int HPW_SMALL = 6;
inline bool use_head_groups(int B, int N, int H, int DH) {
    return DH <= 16 && N <= 96 && H % HPW_SMALL == 0 && (long long)B * H / HPW_SMALL >= 256;
}

int g_attn_head_groups = 1;      // debug switch: 0 = one (image, head) pair per workgroup for every shape

template <int DH, int NC, bool EXPORT, int THREADS, int HPW = 1>
int launch_fwd_t(const void* qkv, void* out, float* lse, void* probs, int B, int N, int H, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)DH);
    static unsigned long long done = 0;
    // the export variant adds one tile buffer per wave
    const size_t sm = fwd_smem<DH, NC, HPW>() + (EXPORT ? (size_t)(THREADS / 64) * export_tile_elems(N) * sizeof(bf16_t) : 0);
    const size_t sm_max = THREADS > ATTN_THREADS ? LDS_BYTES
                                                 : fwd_smem<DH, NC, HPW>() + (EXPORT ? (size_t)(THREADS / 64) * export_tile_elems(NC * 32) * sizeof(bf16_t) : 0);
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_fwd_bf16_kernel<DH, NC, EXPORT, THREADS, HPW>), sm_max, &done,
                        "vitmi_attn_fwd: LDS attribute");
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL((attn_fwd_bf16_kernel<DH, NC, EXPORT, THREADS, HPW>), dim3(B * H / HPW), dim3(THREADS), sm, s,
                       (const bf16_t*)qkv, (bf16_t*)out, lse, (bf16_t*)probs, N, H, scale);
    VITMI_LAUNCH_CHECK("vitmi_attn_fwd(bf16)");
    return VITMI_OK;
}

// The export variant holds a 16 x N tile per wave in LDS, which leaves room for one workgroup per CU at ViT-B
// lengths: it then runs 16 waves per workgroup (the occupancy of two 8-wave workgroups) when those tiles fit.
template <int DH, int NC, bool EXPORT>
int launch_fwd_x(const void* qkv, void* out, float* lse, void* probs, int B, int N, int H, hipStream_t s) {
    if constexpr (DH <= 16 && NC <= 3) {
        if (g_attn_head_groups && use_head_groups(B, N, H, DH))
            return launch_fwd_t<DH, NC, EXPORT, ATTN_THREADS, HPW_SMALL>(qkv, out, lse, probs, B, N, H, s);
    }
    if constexpr (EXPORT) {
        const size_t one = fwd_smem<DH, NC>() + (size_t)(ATTN_THREADS / 64) * export_tile_elems(N) * sizeof(bf16_t);
        const size_t wide = fwd_smem<DH, NC>() + (size_t)(2 * ATTN_THREADS / 64) * export_tile_elems(N) * sizeof(bf16_t);
        if (2 * one > LDS_BYTES && wide <= LDS_BYTES)
            return launch_fwd_t<DH, NC, EXPORT, 2 * ATTN_THREADS>(qkv, out, lse, probs, B, N, H, s);
    }
    return launch_fwd_t<DH, NC, EXPORT, ATTN_THREADS>(qkv, out, lse, probs, B, N, H, s);
}

template <int DH, int NC>
int launch_fwd_noexport(const void* qkv, void* out, float* lse, int B, int N, int H, hipStream_t s) {
    // the score tile of all keys lives in registers: 19 chunks need the 512 registers of a lone wave per SIMD
    return launch_fwd_t<DH, NC, false, (NC > 16 ? ATTN_THREADS / 2 : ATTN_THREADS)>(qkv, out, lse, nullptr, B, N, H, s);
}

template <int DH, int NC>
int launch_fwd(const void* qkv, void* out, float* lse, void* probs, int B, int N, int H, hipStream_t s) {
    if (probs) return launch_fwd_x<DH, NC, true>(qkv, out, lse, probs, B, N, H, s);
    return launch_fwd_x<DH, NC, false>(qkv, out, lse, probs, B, N, H, s);
}

template <int DH, int NC, int HPW>
int launch_bwd_t(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
                 int N, int H, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)DH);
    static unsigned long long done1 = 0, done2 = 0;
    const size_t s1 = dkv_smem<DH, NC, HPW>(), s2 = dq_smem<DH, NC, HPW>();
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_bwd_dkv_bf16_kernel<DH, NC, HPW>), s1, &done1, "vitmi_attn_bwd: LDS attribute (dkv)");
    if (rc != VITMI_OK) return rc;
    rc = ensure_lds(reinterpret_cast<const void*>(&attn_bwd_dq_bf16_kernel<DH, NC, HPW>), s2, &done2, "vitmi_attn_bwd: LDS attribute (dq)");
    if (rc != VITMI_OK) return rc;
    // dQ first: it leaves delta = rowsum(dO * O) per query in the workspace for the dK, dV kernel
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DH, NC, HPW>), dim3(B * H / HPW), dim3(ATTN_THREADS), s2, s,
                       (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(dq, bf16)");
    hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DH, NC, HPW>), dim3(B * H / HPW), dim3(ATTN_THREADS), s1, s,
                       (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale);
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(dkv, bf16)");
    return VITMI_OK;
}

template <int DH, int NC>
int launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int B,
               int N, int H, hipStream_t s) {
    if constexpr (DH <= 16 && NC <= 3) {
        if (g_attn_head_groups && use_head_groups(B, N, H, DH))
            return launch_bwd_t<DH, NC, HPW_SMALL>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
    }
    return launch_bwd_t<DH, NC, 1>(qkv, out, dout, lse, delta, dqkv, B, N, H, s);
}

#define DISPATCH_NC(FN, DH, ...)                      \
    switch (nc) {                                     \
        case 1: return FN<DH, 1>(__VA_ARGS__);        \
        case 2: return FN<DH, 2>(__VA_ARGS__);        \
        case 3: return FN<DH, 3>(__VA_ARGS__);        \
        case 4: return FN<DH, 4>(__VA_ARGS__);        \
        case 5: return FN<DH, 5>(__VA_ARGS__);        \
        case 6: return FN<DH, 6>(__VA_ARGS__);        \
        case 7: return FN<DH, 7>(__VA_ARGS__);        \
        default: return FN<DH, 8>(__VA_ARGS__);       \
    }
// sequences of 257..608 tokens: head dim 64 only
#define DISPATCH_NC_LONG(FN, ...)                     \
    switch (nc) {                                     \
        case 12: return FN<64, 12>(__VA_ARGS__);      \
        case 16: return FN<64, 16>(__VA_ARGS__);      \
        default: return FN<64, 19>(__VA_ARGS__);      \
    }

inline size_t f32_fwd_smem(int N, int DH) { return ((size_t)2 * N * (DH + 1) + 4 * DH + 4 * N) * sizeof(float); }
inline size_t f32_bwdq_smem(int N, int DH) { return ((size_t)2 * N * (DH + 1) + 8 * DH + 4 * N) * sizeof(float); }
inline size_t f32_bwdkv_smem(int N, int DH) { return ((size_t)2 * N * (DH + 1)) * sizeof(float); }

int check_common(const char* who, int dt, int B, int N, int H, int DH) {
    VITMI_CHECK_ARG(dt == VITMI_F32 || dt == VITMI_BF16, "%s: bad dtype %d", who, dt);
    VITMI_CHECK_ARG(B > 0 && N > 0 && H > 0 && DH > 0, "%s: bad shape B=%d N=%d H=%d DH=%d", who, B, N, H, DH);
    if (N > 608 || (N > 256 && !(dt == VITMI_BF16 && DH == 64))) {
        vitmi_set_error("%s: sequence length N=%d is not supported (up to 256 tokens; up to 608 for bf16 heads of 64)", who,
                        N);
        return VITMI_ERR_UNSUPPORTED;
    }
    if (dt == VITMI_BF16 && !(DH == 8 || DH == 16 || DH == 32 || DH == 64)) {
        vitmi_set_error("%s: bf16 head dim %d not in {8,16,32,64}", who, DH);
        return VITMI_ERR_UNSUPPORTED;
    }
    if (dt == VITMI_F32 && f32_bwdq_smem(N, DH) > 160 * 1024) {
        vitmi_set_error("%s: fp32 path needs %zu B of LDS for N=%d DH=%d (limit 163840)", who, f32_bwdq_smem(N, DH), N,
                        DH);
        return VITMI_ERR_UNSUPPORTED;
    }
    return VITMI_OK;
}

}  // namespace

extern "C" int vitmi_debug_set_attn_head_groups(int on) {
    g_attn_head_groups = on;
    return VITMI_OK;
}

extern "C" int vitmi_attn_fwd(const void* qkv, void* out, float* lse, void* probs, int dt, int B, int N, int H, int DH,
                              void* stream) {
    VITMI_CHECK_ARG(qkv && out && lse, "vitmi_attn_fwd: null pointer");
    int rc = check_common("vitmi_attn_fwd", dt, B, N, H, DH);
    if (rc != VITMI_OK) return rc;
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_ATTN_FWD, s, 4.0 * B * H * (double)N * N * DH);
    if (dt == VITMI_F32) {
        const size_t sm = f32_fwd_smem(N, DH);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_attn_fwd: hipFuncSetAttribute");
        hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(B * H), dim3(256), sm, s, (const float*)qkv, (float*)out, lse,
                           (float*)probs, N, H, DH, 1.0f / sqrtf((float)DH));
        VITMI_LAUNCH_CHECK("vitmi_attn_fwd(f32)");
        return VITMI_OK;
    }
    const int nc = pick_nc(N);
    if (nc > 8) {
        if (probs) {
            vitmi_set_error("vitmi_attn_fwd: the probability export is limited to 256 tokens (N=%d)", N);
            return VITMI_ERR_UNSUPPORTED;
        }
        DISPATCH_NC_LONG(launch_fwd_noexport, qkv, out, lse, B, N, H, s)
    }
    switch (DH) {
        case 8: DISPATCH_NC(launch_fwd, 8, qkv, out, lse, probs, B, N, H, s)
        case 16: DISPATCH_NC(launch_fwd, 16, qkv, out, lse, probs, B, N, H, s)
        case 32: DISPATCH_NC(launch_fwd, 32, qkv, out, lse, probs, B, N, H, s)
        default: DISPATCH_NC(launch_fwd, 64, qkv, out, lse, probs, B, N, H, s)
    }
}

extern "C" long long vitmi_attn_bwd_workspace_bytes(int dt, int B, int N, int H, int DH) {
    (void)DH;
    if (dt == VITMI_F32) return 2ll * B * H * N * N * (long long)sizeof(float);
    return (long long)B * H * N * (long long)sizeof(float);     // delta = rowsum(dO * O) per query
}

static long long align256_ll(long long x) { return (x + 255) & ~255ll; }

extern "C" long long vitmi_attn_bwd_dprobs_workspace_bytes(int dt, int B, int N, int H, int DH) {
    if (B <= 0 || N <= 0 || H <= 0 || DH <= 0) return 0;
    const long long es = dt == VITMI_F32 ? 4 : 2;
    // the ordinary workspace, then: one head's fp32 scores, its dS, and a second [B,N,3*H*DH] gradient
    return align256_ll(vitmi_attn_bwd_workspace_bytes(dt, B, N, H, DH)) + align256_ll((long long)N * N * 4) +
           align256_ll((long long)N * N * es) + align256_ll((long long)B * N * 3 * H * DH * es);
}

// dst [B,N,3*H*DH] <- the share of dqkv that a gradient dprobs [B,H,N,N] on the probabilities contributes: its q and k
// slots (the v slots are zeroed: v does not enter the probabilities).  Per (image, head): S = q k^T (fp32), the row
// kernel above, dq = dS k, dk = dS^T q -- four launches through the generic GEMM; this path serves attention-map
// analysis (gradient rollout on a handful of images), not the training step.
static int attn_bwd_from_dprobs(const void* qkv, const float* lse, const void* dprobs, void* dst, float* S, void* dS,
                                int dt, int B, int N, int H, int DH, void* stream) {
    hipStream_t s = as_stream(stream);
    const long long es = dt == VITMI_F32 ? 4 : 2;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float scale = 1.0f / sqrtf((float)DH);
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)((long long)B * N * ld * es), s);
    if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_attn_bwd: hipMemsetAsync");
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const char* qb = (const char*)qkv + ((long long)b * N * ld + (long long)h * DH) * es;
            const char* kb = qb + (long long)I * es;
            char* dq = (char*)dst + ((long long)b * N * ld + (long long)h * DH) * es;
            char* dk = dq + (long long)I * es;
            const long long head = ((long long)b * H + h);
            const char* dp = (const char*)dprobs + head * N * N * es;
            int rc = vitmi_gemm(dt, N, N, DH, qb, ld, 1, kb, ld, 1, S, N, VITMI_F32, VITMI_EPI_NONE, nullptr, nullptr, 0,
                                nullptr, 0, stream);
            if (rc != VITMI_OK) return rc;
            if (dt == VITMI_F32)
                hipLaunchKernelGGL(attn_dprobs_rows_kernel<float>, dim3(ceil_div(N, 4)), dim3(256), 0, s, S, lse + head * N,
                                   (const float*)dp, (float*)dS, N, scale);
            else
                hipLaunchKernelGGL(attn_dprobs_rows_kernel<bf16_t>, dim3(ceil_div(N, 4)), dim3(256), 0, s, S, lse + head * N,
                                   (const bf16_t*)dp, (bf16_t*)dS, N, scale);
            VITMI_LAUNCH_CHECK("vitmi_attn_bwd(dprobs rows)");
            // dq[q][d] = sum_j dS[q][j] k[j][d]
            rc = vitmi_gemm(dt, N, DH, N, dS, N, 1, kb, 1, ld, dq, ld, dt, VITMI_EPI_NONE, nullptr, nullptr, 0, nullptr, 0,
                            stream);
            if (rc != VITMI_OK) return rc;
            // dk[j][d] = sum_q dS[q][j] q[q][d]
            rc = vitmi_gemm(dt, N, DH, N, dS, 1, N, qb, 1, ld, dk, ld, dt, VITMI_EPI_NONE, nullptr, nullptr, 0, nullptr, 0,
                            stream);
            if (rc != VITMI_OK) return rc;
        }
    return VITMI_OK;
}

static int attn_bwd_from_dout(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                              void* workspace, int dt, int B, int N, int H, int DH, hipStream_t s) {
    // five products of the flash-style backward: 10 * B * H * N^2 * DH
    ProfScope prof(VITMI_FAM_ATTN_BWD, s, 10.0 * B * H * (double)N * N * DH);
    if (dt == VITMI_F32) {
        VITMI_CHECK_ARG(workspace, "vitmi_attn_bwd: fp32 path needs a workspace");
        float* Pw = (float*)workspace;
        float* dSw = Pw + (long long)B * H * N * N;
        const size_t s1 = f32_bwdq_smem(N, DH), s2 = f32_bwdkv_smem(N, DH);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_attn_bwd: hipFuncSetAttribute");
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kv_f32_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return vitmi_hip_fail(e, "vitmi_attn_bwd: hipFuncSetAttribute");
        const float scale = 1.0f / sqrtf((float)DH);
        hipLaunchKernelGGL(attn_bwd_q_f32_kernel, dim3(B * H), dim3(256), s1, s, (const float*)qkv, (const float*)dout,
                           lse, (float*)dqkv, Pw, dSw, N, H, DH, scale);
        VITMI_LAUNCH_CHECK("vitmi_attn_bwd(q, f32)");
        hipLaunchKernelGGL(attn_bwd_kv_f32_kernel, dim3(B * H), dim3(256), s2, s, (const float*)qkv, (const float*)dout,
                           Pw, dSw, (float*)dqkv, N, H, DH);
        VITMI_LAUNCH_CHECK("vitmi_attn_bwd(kv, f32)");
        return VITMI_OK;
    }
    const int nc = pick_nc(N);
    VITMI_CHECK_ARG(workspace, "vitmi_attn_bwd: needs the workspace of vitmi_attn_bwd_workspace_bytes");
    float* delta = (float*)workspace;               // [B, H, N] fp32
    if (nc > 8) DISPATCH_NC_LONG(launch_bwd, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
    switch (DH) {
        case 8: DISPATCH_NC(launch_bwd, 8, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        case 16: DISPATCH_NC(launch_bwd, 16, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        case 32: DISPATCH_NC(launch_bwd, 32, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        default: DISPATCH_NC(launch_bwd, 64, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
    }
}

extern "C" int vitmi_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const void* dprobs,
                              void* dqkv, void* workspace, int dt, int B, int N, int H, int DH, void* stream) {
    VITMI_CHECK_ARG(qkv && lse && dqkv, "vitmi_attn_bwd: null pointer");
    VITMI_CHECK_ARG(dout || dprobs, "vitmi_attn_bwd: neither dout nor dprobs");
    VITMI_CHECK_ARG(!dout || out, "vitmi_attn_bwd: dout needs out");
    int rc = check_common("vitmi_attn_bwd", dt, B, N, H, DH);
    if (rc != VITMI_OK) return rc;
    hipStream_t s = as_stream(stream);
    if (!dprobs) return attn_bwd_from_dout(qkv, out, dout, lse, dqkv, workspace, dt, B, N, H, DH, s);
    VITMI_CHECK_ARG(workspace, "vitmi_attn_bwd: dprobs needs the workspace of vitmi_attn_bwd_dprobs_workspace_bytes");
    const long long es = dt == VITMI_F32 ? 4 : 2;
    char* w = (char*)workspace + align256_ll(vitmi_attn_bwd_workspace_bytes(dt, B, N, H, DH));
    float* S = (float*)w;
    void* dS = w + align256_ll((long long)N * N * 4);
    void* tmp = (char*)dS + align256_ll((long long)N * N * es);
    if (!dout) return attn_bwd_from_dprobs(qkv, lse, dprobs, dqkv, S, dS, dt, B, N, H, DH, stream);
    rc = attn_bwd_from_dout(qkv, out, dout, lse, dqkv, workspace, dt, B, N, H, DH, s);
    if (rc != VITMI_OK) return rc;
    rc = attn_bwd_from_dprobs(qkv, lse, dprobs, tmp, S, dS, dt, B, N, H, DH, stream);
    if (rc != VITMI_OK) return rc;
    return vitmi_add(dqkv, tmp, dqkv, dt, (long long)B * N * 3 * H * DH, stream);
}
