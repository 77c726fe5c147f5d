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
#include "small.h"

int vitmi_reserved_cus();   // gemm_fast.hip

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

// Where q / k / v (which = 0 / 1 / 2) of head h of image b start, and the row stride.  T == 0: the rows of to_qkv's
// output, [tokens][3][H][DH] (a head's rows are 2*DH-byte pieces 6*H*DH bytes apart).  T > 0, the number of tokens in
// the batch: head-major planes [3][H][T][DH], so the [N][DH] matrix of one (which, image, head) is contiguous -- what the
// projection GEMM's epilogue writes for heads of 64 and the GEMMs behind the backward pass read.
__device__ __forceinline__ long long qkv_off(int which, int b, int h, int N, int H, int DH, long long T) {
    if (T > 0) return (((long long)which * H + h) * T + (long long)b * N) * DH;
    return (long long)b * N * 3 * H * DH + (long long)which * H * DH + (long long)h * DH;
}
__device__ __forceinline__ long long qkv_ld(int H, int DH, long long T) { return T > 0 ? (long long)DH : 3ll * H * DH; }

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
                                              float* __restrict__ lse, bf16_t* __restrict__ probs, bf16_t* Pw, int lane,
                                              bf16_t* o_lds = nullptr, int o_ld = 0) {
    // o_lds: also leave the tile's output rows in LDS, [row of the image][H * DH] with o_ld elements per row (the
    // one-launch sub-block kernel multiplies them by the projection weight without a round trip through HBM)
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
                if (o_lds) *reinterpret_cast<bf16x4*>(o_lds + q * o_ld + h * DH + d) = pk;
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
                                                            int H, float scale, long long qT) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);     // [HPW] K images, then [HPW] V images
    bf16_t* Vs = Ks + HPW * IMG;
    const int pair0 = xcd_block_order() * HPW;
    const int b = pair0 / H, h0 = pair0 % H;
    const int I = H * DH;
    const long long ld = qkv_ld(H, DH, qT);          // head-major (qT > 0) only with HPW == 1
    const bf16_t* qb = qkv + qkv_off(0, b, h0, N, H, DH, qT);
    const bf16_t* kb = qkv + qkv_off(1, b, h0, N, H, DH, qT);
    const bf16_t* vb = qkv + qkv_off(2, b, h0, N, H, DH, qT);
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

// ================================================================ attention sub-block forward in one launch (small geometry)
// y = x + Wo softmax(q k^T / sqrt(dh)) v + bo with (q, k, v) = Wqkv LN(x) (+ bqkv), mymodel/ViT.py:52-58, 77-104, 119, for
// sequences of up to 96 tokens of width up to 256 with heads of 8 or 16: one workgroup per image.  The image's rows are
// normalised into LDS; the qkv product (weights streamed from L2, small.h) lands, 16 bytes = one lane's 8 features at a
// time, directly in the per-head q / k / v LDS images that attn_fwd_tile reads (a lane's 8 consecutive features are
// one 16-byte chunk of one head's row); the softmax tiles leave their output rows in LDS for the output projection.
// Written out: what the backward pass reads (LN(x) and its statistics, qkv rows, the attention output, the row
// log-sum-exps) and y.
struct AttnSmallArgs {
    const bf16_t* x;
    const float *gamma, *beta;
    const bf16_t* w_qkv;
    const float* b_qkv;
    const bf16_t* w_out;
    const float* b_out;
    bf16_t *xn, *qkv, *o, *y;
    float *mean, *rstd, *lse;
    int N, D, H;
    float eps;
};

template <int DH, int NC, int MT>
__global__ __launch_bounds__(SM_THREADS) void attn_small_fwd_kernel(AttnSmallArgs a) {
    constexpr int NP = NC * 32, IMG = NP * DH, R = MT * 16;
    static_assert(R <= NP, "the row strip must fit the images");
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, D = a.D, H = a.H, I = H * DH, ldx = D + SM_PAD, ldo = I + SM_PAD;
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);   // [R][ldx] normalised rows
    bf16_t* imgs = xs + R * ldx;                    // [3][H][NP][DH] q, k, v images (+ one row of slack)
    bf16_t* os = imgs + 3 * H * IMG + 8;            // [R][ldo] attention output rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const long long row0 = (long long)b * N;
    strip_layernorm<MT>(a.x, a.gamma, a.beta, a.xn, a.mean, a.rstd, row0, N, D, a.eps, xs, ldx);
    // image rows the product below does not write (R .. NP) and the slack row: zero keys / values
    {
        const bf16x8 z = zero8();
        constexpr int CPR = DH / 8;
        constexpr int extra = (NP - R) * CPR;
        if constexpr (extra > 0) {
            for (int i = threadIdx.x; i < 3 * H * extra; i += SM_THREADS) {
                const int img = i / extra, rem = i - img * extra;
                *reinterpret_cast<bf16x8*>(imgs + img * IMG + R * DH + rem * 8) = z;
            }
        }
        if (threadIdx.x == 0) *reinterpret_cast<bf16x8*>(imgs + 3 * H * IMG) = z;
    }
    __syncthreads();
    const int tok = lane & 15, f8 = (lane >> 4) * 8;
    // (q, k, v) rows: 32 features per wave and pass, 8 consecutive ones per lane
    for (int p = wave; p < 3 * I / 32; p += SM_WAVES) {
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        rows_gemm<MT, 2, 3>(a.w_qkv, D, p * 32, D / 32, xs, ldx, lane, acc);
        const int n = p * 32 + f8;
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
        if (a.b_qkv) {
            bias0 = *reinterpret_cast<const f32x4*>(a.b_qkv + n);
            bias1 = *reinterpret_cast<const f32x4*>(a.b_qkv + n + 4);
        }
        const int which = n / I, rem = n - which * I, h = rem / DH, d0 = rem - h * DH;
        bf16_t* img = imgs + (which * H + h) * IMG;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            bf16x8 v;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = r < N ? (bf16_t)(acc[t >> 2][m][t & 3] + (t < 4 ? bias0[t & 3] : bias1[t & 3])) : (bf16_t)0.f;
            *reinterpret_cast<bf16x8*>(img + img_off<DH>(r, d0 >> 3)) = v;
            if (r < N) *reinterpret_cast<bf16x8*>(a.qkv + (row0 + r) * 3 * I + n) = v;
        }
    }
    __syncthreads();
    // softmax tiles: (head, 16 queries) pairs over the waves
    {
        using A = AttnDims<DH>;
        const int fr = lane & 15, fg = lane >> 4;
        const int ntq = (N + 15) >> 4;
        const float scale = 1.0f / sqrtf((float)DH);
        for (int t = wave; t < H * ntq; t += SM_WAVES) {
            const int h = t / ntq, qt = t - h * ntq;
            bf16x8 qf[A::KS];
#pragma unroll
            for (int ks = 0; ks < A::KS; ++ks) qf[ks] = row_frag<DH>(imgs + h * IMG, qt * 16 + fr, ks, fg);
            attn_fwd_tile<DH, NC, false>(imgs + (H + h) * IMG, imgs + (2 * H + h) * IMG, qf, qt, b, h, N, H, scale, a.o, a.lse,
                                         nullptr, nullptr, lane, os, ldo);
        }
    }
    __syncthreads();
    // y = o Wo^T + bo + x
    for (int p = wave; p < D / 32; p += SM_WAVES) {
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = p * 32 + f8;
        bf16x8 res[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok < N ? m * 16 + tok : 0;
            res[m] = *reinterpret_cast<const bf16x8*>(a.x + (row0 + r) * D + n);
        }
        rows_gemm<MT, 2, 3>(a.w_out, I, p * 32, I / 32, os, ldo, lane, acc);
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
        if (a.b_out) {
            bias0 = *reinterpret_cast<const f32x4*>(a.b_out + n);
            bias1 = *reinterpret_cast<const f32x4*>(a.b_out + n + 4);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            if (r < N) {
                bf16x8 o;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    o[t] = (bf16_t)((acc[t >> 2][m][t & 3] + (t < 4 ? bias0[t & 3] : bias1[t & 3])) + (float)res[m][t]);
                *reinterpret_cast<bf16x8*>(a.y + (row0 + r) * D + n) = o;
            }
        }
    }
}

// ================================================================ attention sub-block backward in one launch (small geometry)
// The backward of attn_small_fwd_kernel's sub-block (mymodel/ViT.py:52-58, 77-104, 119) for one image per workgroup, for
// callers that form the weight gradients themselves:
//   dO   = dy Wo                      -> per-head LDS images (bf16), delta = rowsum(dO * O) beside them
//   dq, dk, dv from the q, k, v, dO images (the tile arithmetic of attn_bwd_dq_bf16_kernel / attn_bwd_dkv_bf16_kernel,
//                                        every operand in LDS) -> rows of dqkv in LDS and in HBM (operand of dWqkv)
//   dxn  = dqkv Wqkv                  -> stays in registers (fp32)
//   dx   = LayerNorm backward of dxn at x, + dy (the residual branch)
// and the image's column sums for dgamma, dbeta (of dxn) and the output bias (of dy) as one row of `partial` ([B][3 D]).
// Wo and Wqkv are read through their transposed copies ([I][D] and [D][3 I]: the output feature of each product indexes
// a row).
// Fragments for the products over a head dimension of 8 or 16 inside a 32-deep MFMA, without a branch: the operand that
// is fixed for a tile is zero beyond the head dimension (row_frag_z), so the streamed one may hold anything finite
// there; its lane groups past the head dimension re-read a piece inside it (row_frag_any).
template <int DH>
__device__ __forceinline__ bf16x8 row_frag_any(const bf16_t* img, int row, int fg) {
    return *reinterpret_cast<const bf16x8*>(img + img_off<DH>(row, fg & (DH / 8 - 1)));
}
template <int DH>
__device__ __forceinline__ bf16x8 row_frag_z(const bf16_t* img, int row, int fg) {
    union {
        bf16x8 v;
        u32x4 w;
    } u;
    u.v = row_frag_any<DH>(img, row, fg);
    const unsigned keep = fg * 8 < DH ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) u.w[i] &= keep;
    return u.v;
}

struct AttnSmallBwdArgs {
    const bf16_t *dy, *x, *qkv, *o;
    const float *lse, *gamma, *mean, *rstd;
    const bf16_t *w_out_t, *w_qkv_t;
    bf16_t *dqkv, *dx;
    float* partial;
    int N, D, H;
};

#ifndef VITMI_ASB_ABL
#define VITMI_ASB_ABL 0   // timing ablations: 1 = no softmax backward, 2 = no dO product, 4 = no dxn product
#endif
template <int DH, int NC, int MT>
__global__ __launch_bounds__(SM_THREADS) void attn_small_bwd_kernel(AttnSmallBwdArgs a) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * DH, R = MT * 16;
    static_assert(R <= NP, "the row strip must fit the images");
    static_assert(A::KS == 1 && A::DT == 1, "heads of 8 or 16");
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, D = a.D, H = a.H, I = H * DH, ldx = D + SM_PAD, ldq = 3 * I + SM_PAD;
    bf16_t* act = reinterpret_cast<bf16_t*>(smem);            // [R][ldx] dy rows, later [R][ldq] dqkv rows
    bf16_t* imgs = act + R * (ldx > ldq ? ldx : ldq);         // [4][H][NP][DH] q, k, v, dO images (+ one row of slack)
    float* lse2 = reinterpret_cast<float*>(imgs + 4 * H * IMG + 8);   // [H][NP] in log2 units, then delta [H][NP]
    float* delta = lse2 + H * NP;
    float* red = delta + H * NP;                              // [2][SM_WAVES][R] row sums of the LayerNorm backward
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tok = lane & 15, f8 = (lane >> 4) * 8;
    const int b = blockIdx.x;
    const long long row0 = (long long)b * N;
    // ---- the image's dy rows, its q, k, v images and log-sum-exps into LDS (zeros past the last token); the dy loads
    // are in flight while the images are staged
    {
        constexpr int DYB = 6;   // R (D / 8) <= 96 * 32 pieces of 16 bytes over 512 threads
        u32x4 dv[DYB];
#pragma unroll
        for (int j = 0; j < DYB; ++j) {
            const int i = threadIdx.x + j * SM_THREADS, r = i / (D / 8), c = (i - r * (D / 8)) * 8;
            dv[j] = u32x4{0u, 0u, 0u, 0u};
            if (i < R * (D / 8) && r < N) dv[j] = *reinterpret_cast<const u32x4*>(a.dy + (row0 + r) * D + c);
        }
        const int cpr = 3 * I / 8;
        for (int i0 = threadIdx.x; i0 < NP * cpr; i0 += 8 * SM_THREADS) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * SM_THREADS, r = i / cpr, n = (i - r * cpr) * 8;
                v[j] = u32x4{0u, 0u, 0u, 0u};
                if (i < NP * cpr && r < N) v[j] = *reinterpret_cast<const u32x4*>(a.qkv + (row0 + r) * 3 * I + n);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * SM_THREADS, r = i / cpr, n = (i - r * cpr) * 8;
                const int which = n / I, rem = n - which * I, h = rem / DH, d0 = rem - h * DH;
                if (i < NP * cpr) *reinterpret_cast<u32x4*>(imgs + (which * H + h) * IMG + img_off<DH>(r, d0 >> 3)) = v[j];
            }
        }
#pragma unroll
        for (int j = 0; j < DYB; ++j) {
            const int i = threadIdx.x + j * SM_THREADS, r = i / (D / 8), c = (i - r * (D / 8)) * 8;
            if (i < R * (D / 8)) *reinterpret_cast<u32x4*>(act + r * ldx + c) = dv[j];
        }
        for (int i = threadIdx.x; i < H * NP; i += SM_THREADS) {
            const int h = i / NP, q = i - h * NP;
            lse2[i] = q < N ? a.lse[((long long)b * H + h) * N + q] * LOG2E : 1e30f;   // padded queries get p = 0
            if (q >= R) delta[i] = 0.f;                                                  // (and no NaN from their delta)
        }
        // rows of the dO images that the product below does not write, and the slack row
        if (R < NP) {
            constexpr int CPR = DH / 8, EXTRA = (NP - R) * CPR;
            for (int i = threadIdx.x; i < H * EXTRA; i += SM_THREADS) {
                const int h = i / (EXTRA > 0 ? EXTRA : 1), rem = i - h * EXTRA;
                *reinterpret_cast<u32x4*>(imgs + (3 * H + h) * IMG + R * DH + rem * 8) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        if (threadIdx.x == 0) *reinterpret_cast<u32x4*>(imgs + 4 * H * IMG) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // ---- dO = dy Wo: 32 features per wave and pass, a lane's 8 consecutive ones lie in one head
    for (int p = wave; p < I / 32 && !(VITMI_ASB_ABL & 2); p += SM_WAVES) {
        f32x4 acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = p * 32 + f8;
        bf16x8 oq[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            oq[m] = zero8();
            if (r < N) oq[m] = *reinterpret_cast<const bf16x8*>(a.o + (row0 + r) * I + n);
        }
        rows_gemm<MT, 2, 3>(a.w_out_t, D, p * 32, D / 32, act, ldx, lane, acc);
        const int h = n / DH, d0 = n - h * DH;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            bf16x8 v;
            float dl = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] = (bf16_t)acc[t >> 2][m][t & 3];   // rows past the last token: dy = 0, so 0
                dl += (float)v[t] * (float)oq[m][t];
            }
            if (DH == 16) dl += __shfl_xor(dl, 16, 64);
            *reinterpret_cast<bf16x8*>(imgs + (3 * H + h) * IMG + img_off<DH>(r, d0 >> 3)) = v;
            if (d0 == 0) delta[h * NP + r] = dl;
        }
    }
    __syncthreads();
    // ---- softmax backward: (head, 16 queries) tiles for dq and (head, 16 keys) tiles for dk and dv, over the waves
    {
        const int fr = lane & 15, fg = lane >> 4;
        const float scale = 1.0f / sqrtf((float)DH);
        const float sl2 = scale * LOG2E, lscale = __log2f(scale);
        // a wave's pass holds one tile of each kind (queries 16 tl .. of head h for dq, keys 16 tl .. for dk and dv): two
        // independent chains of products and exponentials in one block of straight-line code, stored together at its end
        for (int t = wave; t < H * MT && !(VITMI_ASB_ABL & 1); t += SM_WAVES) {
            const int h = t / MT, tl = t - h * MT;
            const bf16_t* Qs = imgs + h * IMG;
            const bf16_t* Ks = imgs + (H + h) * IMG;
            const bf16_t* Vs = imgs + (2 * H + h) * IMG;
            const bf16_t* dOs = imgs + (3 * H + h) * IMG;
            const float* l2p = lse2 + h * NP;
            const float* dlp = delta + h * NP;
            const int row = tl * 16 + fr;                      // the query of the dq tile, the key of the dk / dv tile
            const bf16x8 qf = row_frag_z<DH>(Qs, row, fg), dof = row_frag_z<DH>(dOs, row, fg);
            const bf16x8 kf = row_frag_z<DH>(Ks, row, fg), vf = row_frag_z<DH>(Vs, row, fg);
            // p * 1/sqrt(dh) comes straight out of the exponential: log2 of the factor joins the exponent
            const float l2q = l2p[row] - lscale, dlq = dlp[row];
            f32x4 dqt = {0.f, 0.f, 0.f, 0.f}, dkt = dqt, dvt = dqt;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 ds[2];
                bf16x8 pf, dsk;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int prow = perm_row(c, u, fr);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    // scores and dP with the tile's queries on the lanes (keys prow) ...
                    const f32x4 sq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag_any<DH>(Ks, prow, fg), qf, z, 0, 0, 0);
                    const f32x4 dpq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag_any<DH>(Vs, prow, fg), dof, z, 0, 0, 0);
                    // ... and with the tile's keys on the lanes (queries prow)
                    const f32x4 sk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag_any<DH>(Qs, prow, fg), kf, z, 0, 0, 0);
                    const f32x4 dpk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag_any<DH>(dOs, prow, fg), vf, z, 0, 0, 0);
                    const int q0 = 32 * c + 8 * fg + 4 * u;      // the key-tile lane's rows are queries q0 + r
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(&l2p[q0]);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(&dlp[q0]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ds[u][r] = __builtin_amdgcn_exp2f(fmaf(sq[r], sl2, -l2q)) * (dpq[r] - dlq);
                        const float pr = __builtin_amdgcn_exp2f(fmaf(sk[r], sl2, -l2[r]));
                        pf[4 * u + r] = (bf16_t)pr;
                        dsk[4 * u + r] = (bf16_t)(pr * (dpk[r] - dl[r]));   // the 1/sqrt(dh) goes onto the finished dK
                    }
                }
                bf16x8 dsf = pack8(ds[0], ds[1]);                 // element j of the fragment is key 32c + 8fg + j
                // keys past N contribute nothing (only the last chunk can reach past N: NC = ceil(N / 32))
                if (c == NC - 1) dsf = keep_first(dsf, N - 32 * c - 8 * fg);
                dqt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Ks, 32 * c, 0, lane), dsf, dqt, 0, 0, 0);
                dvt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(dOs, 32 * c, 0, lane), pf, dvt, 0, 0, 0);
                dkt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Qs, 32 * c, 0, lane), dsk, dkt, 0, 0, 0);
            }
            const int d = fg * 4;
            if (d < DH) {
                bf16x4 vq, vk, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vq[r] = row < N ? (bf16_t)dqt[r] : (bf16_t)0.f;
                    vk[r] = row < N ? (bf16_t)(dkt[r] * scale) : (bf16_t)0.f;
                    vv[r] = row < N ? (bf16_t)dvt[r] : (bf16_t)0.f;
                }
                bf16_t* dst = act + row * ldq + h * DH + d;
                *reinterpret_cast<bf16x4*>(dst) = vq;
                *reinterpret_cast<bf16x4*>(dst + I) = vk;
                *reinterpret_cast<bf16x4*>(dst + 2 * I) = vv;
            }
        }
    }
    __syncthreads();
    // ---- the dqkv rows out (the caller's weight gradient reads them): by the waves without a 32-feature pass below
    const int pass_waves = D / 32;
    {
        const int first = pass_waves < SM_WAVES ? (int)threadIdx.x - pass_waves * 64 : (int)threadIdx.x;
        const int count = pass_waves < SM_WAVES ? SM_THREADS - pass_waves * 64 : SM_THREADS;
        const int cpr = 3 * I / 8;
        if (first >= 0)
            for (int i = first; i < N * cpr; i += count) {
                const int r = i / cpr, c = (i - r * cpr) * 8;
                *reinterpret_cast<u32x4*>(a.dqkv + (row0 + r) * 3 * I + c) = *reinterpret_cast<const u32x4*>(act + r * ldq + c);
            }
    }
    // ---- dxn = dqkv Wqkv: one 32-feature pass per wave (D <= 256), then the LayerNorm backward across the waves
    const bool active = wave < pass_waves;
    const int n = wave * 32 + f8;
    f32x4 acc[2][MT];
    bf16x8 xq[MT], dyq[MT];
    float mu[MT], rs[MT];
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok < N ? m * 16 + tok : 0;
            xq[m] = *reinterpret_cast<const bf16x8*>(a.x + (row0 + r) * D + n);
            dyq[m] = *reinterpret_cast<const bf16x8*>(a.dy + (row0 + r) * D + n);
            mu[m] = a.mean[row0 + r];
            rs[m] = a.rstd[row0 + r];
        }
        g0 = *reinterpret_cast<const f32x4*>(a.gamma + n);
        g1 = *reinterpret_cast<const f32x4*>(a.gamma + n + 4);
        if (!(VITMI_ASB_ABL & 4)) rows_gemm<MT, 2, 3>(a.w_qkv_t, 3 * I, wave * 32, 3 * I / 32, act, ldq, lane, acc);
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
        float* part = a.partial + (long long)b * 3 * D;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int r = m * 16 + tok;
            float c1 = 0.f, c2 = 0.f;
            for (int w = 0; w < pass_waves; ++w) {
                c1 += red[w * R + r];
                c2 += red[(SM_WAVES + w) * R + r];
            }
            c1 /= (float)D;
            c2 /= (float)D;
            bf16x8 o;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float xh = ((float)xq[m][t] - mu[m]) * rs[m];
                const float dxn = acc[t >> 2][m][t & 3];   // rows past the last token: dqkv = 0, so 0
                const float g = dxn * (t < 4 ? g0[t & 3] : g1[t & 3]);
                const float res = r < N ? (float)dyq[m][t] : 0.f;
                o[t] = (bf16_t)(rs[m] * (g - c1 - xh * c2) + res);
                dg[t] += dxn * xh;
                db[t] += dxn;
                dr[t] += res;
            }
            if (r < N) *reinterpret_cast<bf16x8*>(a.dx + (row0 + r) * D + n) = o;
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
                part[n + t] = dg[t];
                part[D + n + t] = db[t];
                part[2 * D + n + t] = dr[t];
            }
        }
    }
}

// ================================================================ backward dK, dV (bf16)
template <int DH, int NC, int HPW>
__global__ __launch_bounds__(ATTN_THREADS) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta_in,
                                                                bf16_t* __restrict__ dqkv, int N, int H, float scale,
                                                                long long qT, long long dT) {
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
    const long long ld = qkv_ld(H, DH, qT), ldd = qkv_ld(H, DH, dT);
    const bf16_t* qb = qkv + qkv_off(0, b, h0, N, H, DH, qT);
    const bf16_t* kb = qkv + qkv_off(1, b, h0, N, H, DH, qT);
    const bf16_t* vb = qkv + qkv_off(2, b, h0, N, H, DH, qT);
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
            bf16_t* dk = dqkv + qkv_off(1, b, h, N, H, DH, dT) + (long long)key * ldd;
            bf16_t* dv = dqkv + qkv_off(2, b, h, N, H, DH, dT) + (long long)key * ldd;
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
                                                               bf16_t* __restrict__ dqkv, int N, int H, float scale,
                                                               long long qT, long long dT) {
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks_all = reinterpret_cast<bf16_t*>(smem);        // [HPW] K images, then [HPW] V images
    bf16_t* Vs_all = Ks_all + HPW * IMG;

    const int pair0 = xcd_block_order() * HPW;
    const int b = pair0 / H, h0 = pair0 % H;
    const int I = H * DH;
    const long long ld = qkv_ld(H, DH, qT), ldd = qkv_ld(H, DH, dT);
    const bf16_t* qb = qkv + qkv_off(0, b, h0, N, H, DH, qT);
    const bf16_t* kb = qkv + qkv_off(1, b, h0, N, H, DH, qT);
    const bf16_t* vb = qkv + qkv_off(2, b, h0, N, H, DH, qT);
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
            bf16_t* dq = dqkv + qkv_off(0, b, h, N, H, DH, dT) + (long long)q * ldd;
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

// ================================================================ backward, one kernel (bf16, dh = 64, 129..208 tokens)
// P and dS are computed ONCE per (query, key) instead of once in each of the two kernels above: five matrix products
// per pair instead of seven, half the exponentials, and q, k, v, dO are fetched once.  A 16-wave workgroup holds the
// Q, dO and K images of one (image, head) pair.  Waves 0 .. ceil(N/16)-1 own 16 keys each, exactly as in the dK/dV
// kernel (K and V fragments in registers, S and dP with the KEY on the lane, dK^T and dV^T accumulated in
// registers), and walk the query chunks in step.  dQ needs dS with the QUERY on the lane: each owner writes its
// [16 keys x 32 queries] dS^T tile of the chunk into an LDS buffer, a barrier follows, and waves 13..15 then share
// the eight 16 x 16 tiles of dQ^T of the chunk's 32 queries over ALL keys from that buffer
// (transposed reads of the buffer and of the K image) while the owners go on to the next chunk and the other
// buffer.  A chunk's dQ is complete after its own pass, so nothing is accumulated across waves: no atomics, and
// the result does not depend on timing.
constexpr int FUSED_THREADS = 1024;
#ifndef VITMI_FUSED_ABL
#define VITMI_FUSED_ABL 0
#endif
constexpr int FUSED_TASK_WAVE0 = 13;                 // waves 0..12 can own keys
constexpr int FUSED_DQ_WAVE0 = 14;                   // waves 14, 15 compute dQ

// dS^T chunk buffer: [keys][32 queries] bf16; 16-byte piece p of row r is stored at piece p ^ ((r >> 2) & 3), so that
// the 16 rows of a tile write and the 4 x 32 bytes of a transposed read each cover distinct banks
__device__ __forceinline__ int dsb_off(int row, int col) {
    return row * 32 + ((((col >> 3) ^ (row >> 2)) & 3) << 3) + (col & 7);
}
// element j of lane (fr, fg) = buffer[seq0 + 8*fg + j][col0 + fr]
__device__ __forceinline__ bf16x8 dsb_tr_frag(const bf16_t* buf, int seq0, int col0, int lane) {
    const int fg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int col = col0 + 4 * p;
    const int r_lo = seq0 + 8 * fg + q, r_hi = r_lo + 4;
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(buf + dsb_off(r_lo, col)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(buf + dsb_off(r_hi, col)));
    union {
        s16x4 h[2];
        bf16x8 v;
    } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

template <int NC>
__global__ __launch_bounds__(FUSED_THREADS) void attn_bwd_fused_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                   const bf16_t* __restrict__ out,
                                                                   const bf16_t* __restrict__ dout,
                                                                   const float* __restrict__ lse,
                                                                   bf16_t* __restrict__ dqkv, int N, int H, int npairs, float scale,
                                                                   long long qT, long long dT) {
    constexpr int DH = 64;
    using A = AttnDims<DH>;
    constexpr int NP = NC * 32, IMG = NP * A::W, DSB = NP * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* dOs = Qs + IMG;
    bf16_t* Ks = dOs + IMG;
    bf16_t* dsb0 = Ks + IMG;                                  // two chunk buffers
    float* lse2 = reinterpret_cast<float*>(dsb0 + 2 * DSB);   // [NP], then delta [NP]
    float* delta = lse2 + NP;

    const int I = H * DH;
    const long long ld = qkv_ld(H, DH, qT), ldd = qkv_ld(H, DH, dT);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntk = (N + 15) >> 4;                            // the launcher guarantees ntk <= FUSED_TASK_WAVE0
    const bool owner = wave < ntk;
    const float sl2 = scale * LOG2E;
    // rows of the chunk buffers that no owner writes must not hold NaN patterns: their K rows are zero
    for (int i = threadIdx.x; i < 2 * DSB / 8; i += FUSED_THREADS) *reinterpret_cast<bf16x8*>(dsb0 + 8 * i) = zero8();

    // a workgroup per compute unit walks the pairs base, base + G, ...: at any time an XCD works on a contiguous
    // range of pairs (the heads of a few images)
    for (int pair = xcd_block_order(); pair < npairs; pair += gridDim.x) {
        const int b = pair / H, h = pair % H;
        const bf16_t* qb = qkv + qkv_off(0, b, h, N, H, DH, qT);
        const bf16_t* kb = qkv + qkv_off(1, b, h, N, H, DH, qT);
        const bf16_t* vb = qkv + qkv_off(2, b, h, N, H, DH, qT);
        const bf16_t* ob = out + (long long)b * N * I + h * DH;
        const bf16_t* dob = dout + (long long)b * N * I + h * DH;
        // everything derived from the thread index is formed again per pair: kept across the loop, the LDS offsets of
        // the staging and of both phases would crowd out the accumulators
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, fr = lane & 15, fg = lane >> 4;
        const int key = wave * 16 + fr;

        bf16x8 kf[A::KS], vf[A::KS];
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            kf[ks] = global_frag<DH>(kb, ld, owner ? key : N, N, ks, fg);
            vf[ks] = global_frag<DH>(vb, ld, owner ? key : N, N, ks, fg);
        }
        for (int q = tid; q < NP; q += FUSED_THREADS)             // log2-unit lse; padded queries get p = 0
            lse2[q] = q < N ? lse[((long long)b * H + h) * N + q] * LOG2E : 1e30f;
        {
            // Q, dO, K images; delta = rowsum(dO * O) from the same dO pieces (the 8 pieces of a row sit in 8 adjacent lanes)
            constexpr int TOT = NP * A::CPR, IT = (TOT + FUSED_THREADS - 1) / FUSED_THREADS;
            bf16x8 vq[IT], vk[IT], vd[IT], vo[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int c = tid + i * FUSED_THREADS;
                const int r = c / A::CPR, col = (c % A::CPR) * 8;
                vq[i] = vk[i] = vd[i] = vo[i] = zero8();
                if (c < TOT && r < N && !(VITMI_FUSED_ABL & 8)) {
                    vq[i] = *reinterpret_cast<const bf16x8*>(qb + (long long)r * ld + col);
                    vk[i] = *reinterpret_cast<const bf16x8*>(kb + (long long)r * ld + col);
                    vd[i] = *reinterpret_cast<const bf16x8*>(dob + (long long)r * I + col);
                    vo[i] = *reinterpret_cast<const bf16x8*>(ob + (long long)r * I + col);
                }
            }
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int c = tid + i * FUSED_THREADS;
                const int r = c / A::CPR, kc = c % A::CPR;
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) part += (float)vo[i][j] * (float)vd[i][j];
                part += __shfl_xor(part, 1, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 4, 64);
                if (c < TOT) {
                    *reinterpret_cast<bf16x8*>(Qs + img_off<DH>(r, kc)) = vq[i];
                    *reinterpret_cast<bf16x8*>(Ks + img_off<DH>(r, kc)) = vk[i];
                    *reinterpret_cast<bf16x8*>(dOs + img_off<DH>(r, kc)) = vd[i];
                    if (kc == 0) delta[r] = part;
                }
            }
        }
        __syncthreads();

        if (wave < FUSED_DQ_WAVE0) {
            // ---- key owners (and the spare wave, which only keeps the barriers): dK^T, dV^T, and the chunk's dS^T tile
            f32x4 dkt[A::DT], dvt[A::DT];
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) dkt[dt] = dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                bf16_t* dsb = dsb0 + (c & 1) * DSB;
                if (owner && !(VITMI_FUSED_ABL & 2)) {
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
                        const int q0 = 32 * c + 8 * fg + 4 * u;     // this lane's rows are queries q0 + r
                        const f32x4 l2 = *reinterpret_cast<const f32x4*>(&lse2[q0]);
                        const f32x4 dl = *reinterpret_cast<const f32x4*>(&delta[q0]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, -l2[r]));
                            pf[4 * u + r] = (bf16_t)p;
                            dsf[4 * u + r] = (bf16_t)(p * (dpa[r] - dl[r]));     // the 1/sqrt(dh) goes onto the finished dK, dQ
                        }
                    }
                    // element j is query 32c + 8fg + j of key `key`; keys past N must not reach dQ
                    *reinterpret_cast<bf16x8*>(dsb + dsb_off(key, 8 * fg)) = key < N ? dsf : zero8();
#pragma unroll
                    for (int dt = 0; dt < A::DT; ++dt) {
                        dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(dOs, 32 * c, dt * 16, lane), pf, dvt[dt],
                                                                          0, 0, 0);
                        dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(Qs, 32 * c, dt * 16, lane), dsf, dkt[dt],
                                                                          0, 0, 0);
                    }
                }
                // the chunk's LDS writes are done and visible; stores to HBM still in flight are not waited for
                if (!(VITMI_FUSED_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (owner && key < N) {
                bf16_t* dk = dqkv + qkv_off(1, b, h, N, H, DH, dT) + (long long)key * ldd;
                bf16_t* dv = dqkv + qkv_off(2, b, h, N, H, DH, dT) + (long long)key * ldd;
#pragma unroll
                for (int dt = 0; dt < A::DT; ++dt) {
                    const int d = dt * 16 + fg * 4;
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
        } else {
            // ---- dQ^T of each chunk, [64 head dim x 32 queries], behind the chunk's barrier: each of the two waves
            // takes a 32 x 32 block (two head-dim tiles x two query halves), K^T fragments times dS^T fragments over the
            // key chunks, the fragments of key chunk kc + 1 requested before the products of kc
            const int tw = wave - FUSED_DQ_WAVE0;
            // the owners wait for these two waves at the next barrier: they go first wherever they share a SIMD
            __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                const bf16_t* dsb = dsb0 + (c & 1) * DSB;
                if (!(VITMI_FUSED_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (VITMI_FUSED_ABL & 1) continue;
                f32x4 acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 ka[2][2], sb[2][2];        // [stage][head-dim tile], [stage][query half]
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ka[0][i] = tr_frag<DH>(Ks, 0, (2 * tw + i) * 16, lane);
                    sb[0][i] = dsb_tr_frag(dsb, 0, 16 * i, lane);
                }
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    if (kc + 1 < NC) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            ka[(kc + 1) & 1][i] = tr_frag<DH>(Ks, 32 * (kc + 1), (2 * tw + i) * 16, lane);
                            sb[(kc + 1) & 1][i] = dsb_tr_frag(dsb, 32 * (kc + 1), 16 * i, lane);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kc & 1][i], sb[kc & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 32 * c + 16 * j + fr;
                    if (q < N) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bf16x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[i][j][r] * scale);
                            *reinterpret_cast<bf16x4*>(dqkv + qkv_off(0, b, h, N, H, DH, dT) + (long long)q * ldd + (2 * tw + i) * 16 + fg * 4) = v;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // the images are rewritten for the next pair: the last chunk's dQ must be through with them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---------------------------------------------------------------- the same, pipelined across pairs (193..208 tokens)
// A workgroup fills its compute unit, so nothing else hides the time its operands take to arrive: here they are
// fetched one pair ahead.  The Q and dO images (and lse, delta) exist twice; at the first chunk of pair i every wave
// requests its share of the Q and dO images of pair i+1 straight into the other set (direct-to-LDS loads: no
// registers, so all 52 KB are in flight at once), the O pieces follow through registers and meet the landed dO
// image five chunks later to give delta.  The K image is read by the dQ waves up to the last chunk, so the next K
// travels in registers over the last two chunks and is written behind the barrier that ends the pair; the owners'
// K and V fragments of the next pair are requested as
// soon as the last chunk's score products have used the old ones.  Images hold 208 rows (13 key tiles): the last
// query chunk reads 16 rows past an image, which are finite values of the neighbouring image and meet
// probabilities that are exactly zero; the dQ product masks the key rows past 208 instead.
constexpr int PIPE_NC = 7, PIPE_ROWS = 208, PIPE_NQ = PIPE_NC * 32;
constexpr int PIPE_LAND = PIPE_NC - 3;               // the chunk by whose end the next pair's Q and dO images have landed
__device__ __attribute__((aligned(16))) unsigned int g_attn_zero16[4] = {0u, 0u, 0u, 0u};   // source of the pieces of rows past N
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
constexpr size_t PIPE_LDS = (size_t)5 * PIPE_ROWS * 64 * 2 + (size_t)2 * PIPE_ROWS * 32 * 2 + (size_t)4 * PIPE_NQ * 4;

__global__ __launch_bounds__(FUSED_THREADS) void attn_bwd_pipe_bf16_kernel(const bf16_t* __restrict__ qkv,
                                                                  const bf16_t* __restrict__ out,
                                                                  const bf16_t* __restrict__ dout,
                                                                  const float* __restrict__ lse,
                                                                  bf16_t* __restrict__ dqkv, int N, int H, int npairs,
                                                                  float scale, long long qT, long long dT) {
    constexpr int DH = 64, NC = PIPE_NC, ROWS = PIPE_ROWS, NQ = PIPE_NQ;
    using A = AttnDims<DH>;
    constexpr int IMG = ROWS * A::W, DSB = ROWS * 32, PCS = ROWS * A::CPR;   // PCS: 16-byte pieces of an image
    static_assert(ROWS % 8 == 0 && 2 * (ROWS / 8) == 4 * FUSED_TASK_WAVE0 && PCS <= 2 * FUSED_THREADS && PIPE_LAND >= 0,
                  "the fetch schedule below assumes these");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* img0 = reinterpret_cast<bf16_t*>(smem);          // [Q set 0][dO set 0][Q set 1][dO set 1][K]
    bf16_t* Ks = img0 + 4 * IMG;
    bf16_t* dsb0 = Ks + IMG;                                  // two chunk buffers of dS^T
    float* lsd = reinterpret_cast<float*>(dsb0 + 2 * DSB);    // [set][lse in log2 units | delta][NQ]

    const int I = H * DH;
    const long long ld = qkv_ld(H, DH, qT), ldd = qkv_ld(H, DH, dT);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntk = (N + 15) >> 4;                            // <= 13
    const bool owner = wave < ntk;
    const float sl2 = scale * LOG2E;
    const int G = gridDim.x;

    // every byte of LDS starts finite: rows that are only ever read past an image, chunk-buffer rows no owner writes
    for (int i = threadIdx.x; i < (int)(PIPE_LDS / 16); i += FUSED_THREADS) *reinterpret_cast<bf16x8*>(img0 + 8 * i) = zero8();
    __syncthreads();

    int pair = xcd_block_order();
    bf16x8 kf[A::KS], vf[A::KS];
    if (pair < npairs) {
        // the first pair arrives the plain way: all four matrices at once
        const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fg = lane >> 4;
        const int b = pair / H, h = pair % H;
        const bf16_t* qb = qkv + qkv_off(0, b, h, N, H, DH, qT);
        const bf16_t* kb = qkv + qkv_off(1, b, h, N, H, DH, qT);
        const bf16_t* vb = qkv + qkv_off(2, b, h, N, H, DH, qT);
        const bf16_t* ob = out + (long long)b * N * I + h * DH;
        const bf16_t* dob = dout + (long long)b * N * I + h * DH;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            kf[ks] = global_frag<DH>(kb, ld, owner ? wave * 16 + fr : N, N, ks, fg);
            vf[ks] = global_frag<DH>(vb, ld, owner ? wave * 16 + fr : N, N, ks, fg);
        }
        if (tid < NQ) lsd[tid] = tid < N ? lse[((long long)b * H + h) * N + tid] * LOG2E : 1e30f;
        bf16x8 vq[2], vk[2], vd[2], vo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * FUSED_THREADS;
            const int r = c >> 3, col = (c & 7) * 8;
            vq[i] = vk[i] = vd[i] = vo[i] = zero8();
            if (c < PCS && r < N) {
                vq[i] = *reinterpret_cast<const bf16x8*>(qb + (long long)r * ld + col);
                vk[i] = *reinterpret_cast<const bf16x8*>(kb + (long long)r * ld + col);
                vd[i] = *reinterpret_cast<const bf16x8*>(dob + (long long)r * I + col);
                vo[i] = *reinterpret_cast<const bf16x8*>(ob + (long long)r * I + col);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * FUSED_THREADS;
            const int r = c >> 3, kc = c & 7;
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)vo[i][j] * (float)vd[i][j];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (c < PCS) {
                *reinterpret_cast<bf16x8*>(img0 + img_off<DH>(r, kc)) = vq[i];
                *reinterpret_cast<bf16x8*>(img0 + IMG + img_off<DH>(r, kc)) = vd[i];
                *reinterpret_cast<bf16x8*>(Ks + img_off<DH>(r, kc)) = vk[i];
                if (kc == 0) lsd[NQ + r] = part;
            }
        }
    }
    __syncthreads();

    // per pair: where the next pair's operands come from, which LDS set is read and which is filled
    struct Ctx {
        int b, h;
        bool has_next;
        const bf16_t *qn, *kn, *vn, *on, *don;    // next pair: q, k, v, O, dO
        bf16_t *dq, *dk, *dv;                     // this pair's gradient rows (row stride ldd)
        const float* lsen;
        const bf16_t *Qs, *dOs;
        bf16_t *Qn, *dOn;
        const float *lse2, *delta;
        float *lse2n, *deltan;
    };
    auto make_ctx = [&](int pair, int cur) {
        Ctx x;
        x.b = pair / H;
        x.h = pair % H;
        x.has_next = pair + G < npairs;
        const int nb = (pair + G) / H, nh = (pair + G) % H;
        x.qn = qkv + qkv_off(0, nb, nh, N, H, DH, qT);
        x.kn = qkv + qkv_off(1, nb, nh, N, H, DH, qT);
        x.vn = qkv + qkv_off(2, nb, nh, N, H, DH, qT);
        x.dq = dqkv + qkv_off(0, x.b, x.h, N, H, DH, dT);
        x.dk = dqkv + qkv_off(1, x.b, x.h, N, H, DH, dT);
        x.dv = dqkv + qkv_off(2, x.b, x.h, N, H, DH, dT);
        x.on = out + (long long)nb * N * I + nh * DH;
        x.don = dout + (long long)nb * N * I + nh * DH;
        x.lsen = lse + ((long long)nb * H + nh) * N;
        x.Qs = img0 + cur * 2 * IMG;
        x.dOs = x.Qs + IMG;
        x.Qn = img0 + (cur ^ 1) * 2 * IMG;
        x.dOn = x.Qn + IMG;
        x.lse2 = lsd + cur * 2 * NQ;
        x.delta = x.lse2 + NQ;
        x.lse2n = lsd + (cur ^ 1) * 2 * NQ;
        x.deltan = x.lse2n + NQ;
        return x;
    };
    // a thread's two pieces of the next K image: requested at the start of chunk NC - 2, kept in registers until the
    // pair's last barrier, when the dQ waves are through with the present image
    auto fetch_k = [&](const Ctx& x, int tid, bf16x8& kp0, bf16x8& kp1) {
        int tl = tid;
        asm volatile("" : "+v"(tl));              // addresses are formed where they are used, not ahead of the chunk loop
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = i * FUSED_THREADS + tl, r = p >> 3, col = (p & 7) * 8;
            bf16x8 v = zero8();
            if (p < PCS && r < N) v = *reinterpret_cast<const bf16x8*>(x.kn + (long long)r * ld + col);
            if (i == 0) kp0 = v; else kp1 = v;
        }
    };
    auto commit_k = [&](int tid, const bf16x8& kp0, const bf16x8& kp1) {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        *reinterpret_cast<bf16x8*>(Ks + img_off<DH>(tl >> 3, tl & 7)) = kp0;
        const int p1 = FUSED_THREADS + tl;
        if (p1 < PCS) *reinterpret_cast<bf16x8*>(Ks + img_off<DH>(p1 >> 3, p1 & 7)) = kp1;
    };

    if (wave < FUSED_TASK_WAVE0) {
        // ================= waves 0..12: key owners (a wave past ceil(N/16) only fetches and keeps the barriers)
        int cur = 0;
        for (; pair < npairs; pair += G, cur ^= 1) {
            const Ctx x = make_ctx(pair, cur);
            // everything derived from the thread index is formed again per pair: kept across the loop, the LDS offsets
            // would crowd out the accumulators
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            tid &= FUSED_THREADS - 1;
            const int lane = tid & 63, fr = lane & 15, fg = lane >> 4;
            const int key = wave * 16 + fr;
            bf16x8 kp0 = zero8(), kp1 = zero8();
            f32x4 dkt[A::DT], dvt[A::DT];
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) dkt[dt] = dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                bf16_t* dsb = dsb0 + (c & 1) * DSB;
                if (c == NC - 2 && x.has_next) fetch_k(x, tid, kp0, kp1);
                if (owner) {
                    bf16x8 pf, dsf;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qrow = perm_row(c, u, fr);
                        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dpa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < A::KS; ++ks) {
                            sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(x.Qs, qrow, ks, fg), kf[ks], sa, 0, 0, 0);
                            dpa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<DH>(x.dOs, qrow, ks, fg), vf[ks], dpa, 0, 0, 0);
                        }
                        const int q0 = 32 * c + 8 * fg + 4 * u;     // this lane's rows are queries q0 + r
                        const f32x4 l2 = *reinterpret_cast<const f32x4*>(&x.lse2[q0]);
                        const f32x4 dl = *reinterpret_cast<const f32x4*>(&x.delta[q0]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, -l2[r]));
                            pf[4 * u + r] = (bf16_t)p;
                            dsf[4 * u + r] = (bf16_t)(p * (dpa[r] - dl[r]));     // the 1/sqrt(dh) goes onto the finished dK, dQ
                        }
                    }
                    if (c == NC - 1 && x.has_next) {
                        // the old fragments have served their last product
#pragma unroll
                        for (int ks = 0; ks < A::KS; ++ks) {
                            kf[ks] = global_frag<DH>(x.kn, ld, key, N, ks, fg);
                            vf[ks] = global_frag<DH>(x.vn, ld, key, N, ks, fg);
                        }
                    }
                    // element j is query 32c + 8fg + j of key `key`; keys past N must not reach dQ
                    *reinterpret_cast<bf16x8*>(dsb + dsb_off(key, 8 * fg)) = key < N ? dsf : zero8();
#pragma unroll
                    for (int dt = 0; dt < A::DT; ++dt) {
                        dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(x.dOs, 32 * c, dt * 16, lane), pf, dvt[dt],
                                                                          0, 0, 0);
                        dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<DH>(x.Qs, 32 * c, dt * 16, lane), dsf, dkt[dt],
                                                                          0, 0, 0);
                    }
                }
                if (c == 0 && x.has_next) {
                    // the next pair's Q and dO images, straight into LDS: 52 blocks of 8 rows, four per wave.  The loads
                    // are issued outside the compiler's view, so no LDS read is made to wait for them; this wave waits
                    // for them itself at the end of chunk PIPE_LAND.  They follow the chunk's products because the wait for
                    // the K and V fragments, requested at the end of the pair before, would otherwise wait for them too.
                    int ll = lane;
                    asm volatile("" : "+v"(ll));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = wave * 4 + k;                        // Q blocks, then dO blocks
                        const bool isq = j < ROWS / 8;
                        const int blk = isq ? j : j - ROWS / 8;
                        const int r = blk * 8 + (ll >> 3);
                        const int kc = (ll & 7) ^ img_swz<DH>(r);          // the piece that belongs at this lane's place
                        const bf16_t* src = isq ? x.qn + (long long)r * ld + kc * 8 : x.don + (long long)r * I + kc * 8;
                        const void* from = r < N ? (const void*)src : (const void*)g_attn_zero16;
                        const unsigned to = (unsigned)(unsigned long long)(lds_ptr_t)((isq ? x.Qn : x.dOn) + blk * 8 * A::W);
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(from), "s"(to) : "memory");
                    }
                }
                // chunk PIPE_LAND ends with this wave's direct-to-LDS loads landed (nothing else of its own is in flight)
                if (c == PIPE_LAND && x.has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the chunk's LDS writes are done and visible; stores to HBM still in flight are not waited for
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (owner && key < N) {
                bf16_t* dk = x.dk + (long long)key * ldd;
                bf16_t* dv = x.dv + (long long)key * ldd;
#pragma unroll
                for (int dt = 0; dt < A::DT; ++dt) {
                    const int d = dt * 16 + fg * 4;
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
            // the last chunk's dQ is through with the K image: the next one goes in
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (x.has_next) commit_k(tid, kp0, kp1);
        }
    } else {
        // ================= helper waves 13..15.  All three fetch what needs arithmetic on the way: the next pair's O
        // and dO pieces, two of each per thread and chunk, whose products give delta = rowsum(dO * O), and lse.
        // Waves 14, 15 also compute dQ^T of chunk c - 1, [64 head dim x 32 queries], beside the owners' chunk c: each
        // takes a 32 x 32 block (two head-dim tiles x two query halves), K^T fragments times dS^T fragments over the
        // key chunks, the fragments of key chunk kc + 1 requested before the products of kc.
        constexpr int HT = FUSED_THREADS - FUSED_TASK_WAVE0 * 64;          // helper threads: 192
        constexpr int RP = 2;                                              // pieces per thread and round
        static_assert(RP * HT * (NC - 2) >= PCS && NQ <= 2 * HT, "NC - 2 rounds cover an image");
        const int tw = wave - FUSED_DQ_WAVE0;
        int cur = 0;
        for (; pair < npairs; pair += G, cur ^= 1) {
            const Ctx x = make_ctx(pair, cur);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            tid &= FUSED_THREADS - 1;
            const int lane = tid & 63, fr = lane & 15, fg = lane >> 4;
            const int ht = tid - FUSED_TASK_WAVE0 * 64;
            bf16x8 kp0 = zero8(), kp1 = zero8();
            bf16x8 ro[RP], rd[RP];                 // a round's O pieces and the dO pieces of the same places
            float lv[2] = {0.f, 0.f};
            // round k (requested at the start of chunk k, used at the start of chunk k + 1): pieces ht + (RP k + i) HT
            auto round_load = [&](int k) {
                int hl = ht;
                asm volatile("" : "+v"(hl));
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int p = hl + (RP * k + i) * HT, r = p >> 3, col = (p & 7) * 8;
                    ro[i] = rd[i] = zero8();
                    if (p < PCS && r < N) {
                        ro[i] = *reinterpret_cast<const bf16x8*>(x.on + (long long)r * I + col);
                        rd[i] = *reinterpret_cast<const bf16x8*>(x.don + (long long)r * I + col);
                    }
                }
            };
            auto round_use = [&](int k) {
                int hl = ht;
                asm volatile("" : "+v"(hl));
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int p = hl + (RP * k + i) * HT, r = p >> 3, kc = p & 7;
                    float part = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) part += (float)ro[i][j] * (float)rd[i][j];
                    part += __shfl_xor(part, 1, 64);       // a row's 8 pieces are 8 adjacent lanes
                    part += __shfl_xor(part, 2, 64);
                    part += __shfl_xor(part, 4, 64);
                    if (p < PCS && kc == 0) x.deltan[r] = part;
                }
            };
            auto dq_chunk = [&](int c) {
                const bf16_t* dsb = dsb0 + (c & 1) * DSB;
                int ln = lane;
                asm volatile("" : "+v"(ln));      // fragment addresses are formed per chunk, not kept across the loop
                ln &= 63;
                f32x4 acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 ka[2][2], sb[2][2];        // [stage][head-dim tile], [stage][query half]
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ka[0][i] = tr_frag<DH>(Ks, 0, (2 * tw + i) * 16, ln);
                    sb[0][i] = dsb_tr_frag(dsb, 0, 16 * i, ln);
                }
#pragma unroll
                for (int kc = 0; kc < NC; ++kc) {
                    if (kc + 1 < NC) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            ka[(kc + 1) & 1][i] = tr_frag<DH>(Ks, 32 * (kc + 1), (2 * tw + i) * 16, ln);
                            sb[(kc + 1) & 1][i] = dsb_tr_frag(dsb, 32 * (kc + 1), 16 * i, ln);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (32 * (kc + 1) > ROWS) {
                        // element j of the fragment is key row 32 kc + 8 fg + j: rows past the buffer count as zero
#pragma unroll
                        for (int j = 0; j < 2; ++j) sb[kc & 1][j] = keep_first(sb[kc & 1][j], ROWS - 32 * kc - 8 * (ln >> 4));
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kc & 1][i], sb[kc & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 32 * c + 16 * j + fr;
                    if (q < N) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bf16x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(acc[i][j][r] * scale);
                            *reinterpret_cast<bf16x4*>(x.dq + (long long)q * ldd + (2 * tw + i) * 16 + fg * 4) = v;
                        }
                    }
                }
            };
            // the owners wait for these waves at the next barrier: they go first wherever they share a SIMD
            __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
            for (int c = 0; c < NC; ++c) {
                if (x.has_next) {
                    if (c > 0 && c <= NC - 2) round_use(c - 1);
                    if (c == 0) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (ht + i * HT < N) lv[i] = x.lsen[ht + i * HT];
                    } else if (c == 1) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int q = ht + i * HT;
                            if (q < NQ) x.lse2n[q] = q < N ? lv[i] * LOG2E : 1e30f;
                        }
                    }
                    if (c < NC - 2) round_load(c);
                    if (c == NC - 2) fetch_k(x, tid, kp0, kp1);
                }
                if (c > 0 && tw >= 0) dq_chunk(c - 1);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (tw >= 0) dq_chunk(NC - 1);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (x.has_next) commit_k(tid, kp0, kp1);
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

// ================================================================ fp32 path on the fp32 matrix instruction
// The same arithmetic (IEEE fp32 products and sums, expf / logf) as the row-by-row kernels above, as products of
// 16 x 16 tiles through v_mfma_f32_16x16x4_f32: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] of a
// 16 x 4 / 4 x 16 step and receives D[4 (l >> 4) + r][l & 15].  One block per (image, head), a wave per 16-row tile.
// Score tiles are formed transposed, S^T[key][query] = K Q^T, so that a lane's four values are four keys of ONE
// query: they are exactly the B operand (P^T[key][query]) of the following product O^T = V^T P^T when the four steps
// of a 16-key tile take the keys in the order 4 (l >> 4) + r, r = 0..3 -- a sum may run over its index in any order
// as long as both operands agree -- so probabilities never leave registers.  The operand that changes with the tile
// (K, V rows; Q, dO rows in the second half of the backward pass) is read from an LDS image [rows][LD], one element
// per lane and step; the wave's own rows are fragments in registers.  Images hold 16 NT rows, zero past N and past
// DH, LD = 16 ceil(DH / 16) + 4 floats.
template <int DH>
struct F32A {
    static constexpr int DT = (DH + 15) / 16, W = 16 * DT, LD = W + 4, KS = DH / 4;
};

template <int DH>
__device__ __forceinline__ void f32_stage(const float* __restrict__ src, long long ld, int N, int rows, float* img) {
    using A = F32A<DH>;
    constexpr int CPR = A::LD / 4;         // 16-byte pieces per LDS row
    for (int c = threadIdx.x; c < rows * CPR; c += blockDim.x) {
        const int r = c / CPR, k = (c - r * CPR) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < N && k < DH) v = *reinterpret_cast<const f32x4*>(src + (long long)r * ld + k);
        *reinterpret_cast<f32x4*>(img + r * A::LD + k) = v;
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, float* __restrict__ probs,
                                                                int N, int H, float scale) {
    using A = F32A<DH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + 16 * NT * A::LD;
    const int blk = xcd_block_order();
    const int b = blk / H, h = blk % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float* qb = qkv + (long long)b * N * ld + h * DH;
    f32_stage<DH>(qb + I, ld, N, 16 * NT, Ks);
    f32_stage<DH>(qb + 2 * I, ld, N, 16 * NT, Vs);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int ntk = (N + 15) >> 4;
    for (int qt = wave; qt < ntk; qt += 4) {
        const int q = 16 * qt + fr;
        float qf[A::KS];
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) qf[ks] = q < N ? qb[(long long)q * ld + 4 * ks + fk] : 0.f;
        f32x4 st[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            st[c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (c < ntk) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const float* kr = Ks + (16 * c + fr) * A::LD + fk;
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) acc = mfma4(kr[4 * ks], qf[ks], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (16 * c + 4 * fk + r < N) ? acc[r] * scale : -INFINITY;
                    st[c][r] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(st[c][r] - mx);     // exp(-inf) = 0 for the keys past N
                st[c][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (fk == 0 && q < N) lse[((long long)b * H + h) * N + q] = mx + logf(sum);
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[c][r] *= inv;
        if (probs && q < N) {
            float* prow = probs + (((long long)b * H + h) * N + q) * N;
#pragma unroll
            for (int c = 0; c < NT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * c + 4 * fk + r;
                    if (key < N) prow[key] = st[c][r];
                }
        }
        f32x4 o[A::DT];
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            if (c < ntk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* vr = Vs + (16 * c + 4 * fk + r) * A::LD + fr;
#pragma unroll
                    for (int dt = 0; dt < A::DT; ++dt) o[dt] = mfma4(vr[16 * dt], st[c][r], o[dt]);
                }
            }
        }
        if (q < N) {
            float* orow = out + ((long long)b * N + q) * I + h * DH;
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                const int d = 16 * dt + 4 * fk;
                if (d < DH) *reinterpret_cast<f32x4*>(orow + d) = o[dt];
            }
        }
    }
}

// Backward pass in one launch.  First half: a wave per query tile against the K and V images -- S^T and dP^T = V dO^T
// tiles of all key tiles in registers, delta[q] = sum_k P dP from them, dS^T = P^T (dP^T - delta) / sqrt(dh), and
// dQ^T = K^T dS^T; delta goes to LDS.  Second half: the images are rewritten with Q and dO, a wave per key tile forms
// S = Q K^T and dP = dO V^T tiles (queries as rows now, so that the lane's four values are four queries of one key),
// P from the stored row log-sum-exps, and accumulates dV^T = dO^T P and dK^T = Q^T dS over the query tiles.
template <int DH, int NT>
__global__ __launch_bounds__(256) void attn_bwd_f32_mfma_kernel(const float* __restrict__ qkv,
                                                                const float* __restrict__ dout,
                                                                const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                int N, int H, float scale) {
    using A = F32A<DH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* img0 = reinterpret_cast<float*>(smem);
    float* img1 = img0 + 16 * NT * A::LD;
    float* ls = img1 + 16 * NT * A::LD;      // row log-sum-exp, a large value past N (the probability becomes 0)
    float* dl = ls + 16 * NT;                // delta
    const int blk = xcd_block_order();
    const int b = blk / H, h = blk % H;
    const int I = H * DH;
    const long long ld = 3ll * I;
    const float* qb = qkv + (long long)b * N * ld + h * DH;
    const float* dob = dout + (long long)b * N * I + h * DH;
    float* gb = dqkv + (long long)b * N * ld + h * DH;
    f32_stage<DH>(qb + I, ld, N, 16 * NT, img0);
    f32_stage<DH>(qb + 2 * I, ld, N, 16 * NT, img1);
    for (int t = threadIdx.x; t < 16 * NT; t += blockDim.x) ls[t] = t < N ? lse[((long long)b * H + h) * N + t] : 1e30f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int ntk = (N + 15) >> 4;
    for (int qt = wave; qt < ntk; qt += 4) {
        const int q = 16 * qt + fr;
        float qf[A::KS], dof[A::KS];
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            qf[ks] = q < N ? qb[(long long)q * ld + 4 * ks + fk] : 0.f;
            dof[ks] = q < N ? dob[(long long)q * I + 4 * ks + fk] : 0.f;
        }
        const float lq = ls[q];
        f32x4 pt[NT], dpt[NT];
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            pt[c] = dpt[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < ntk) {
                f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
                const float* kr = img0 + (16 * c + fr) * A::LD + fk;
                const float* vr = img1 + (16 * c + fr) * A::LD + fk;
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) {
                    sa = mfma4(kr[4 * ks], qf[ks], sa);
                    da = mfma4(vr[4 * ks], dof[ks], da);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = (16 * c + 4 * fk + r < N) ? expf(sa[r] * scale - lq) : 0.f;
                    pt[c][r] = pv;
                    dpt[c][r] = da[r];
                    dsum += pv * da[r];
                }
            }
        }
        dsum += __shfl_xor(dsum, 16, 64);
        dsum += __shfl_xor(dsum, 32, 64);
        if (fk == 0) dl[q] = dsum;
        f32x4 dq[A::DT];
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            if (c < ntk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dsv = pt[c][r] * (dpt[c][r] - dsum) * scale;
                    const float* kr = img0 + (16 * c + 4 * fk + r) * A::LD + fr;
#pragma unroll
                    for (int dt = 0; dt < A::DT; ++dt) dq[dt] = mfma4(kr[16 * dt], dsv, dq[dt]);
                }
            }
        }
        if (q < N) {
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                const int d = 16 * dt + 4 * fk;
                if (d < DH) *reinterpret_cast<f32x4*>(gb + (long long)q * ld + d) = dq[dt];
            }
        }
    }
    __syncthreads();   // every wave is through with the K and V images; delta is complete
    f32_stage<DH>(qb, ld, N, 16 * NT, img0);
    f32_stage<DH>(dob, I, N, 16 * NT, img1);
    __syncthreads();
    for (int kt = wave; kt < ntk; kt += 4) {
        const int key = 16 * kt + fr;
        float kf[A::KS], vf[A::KS];
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) {
            kf[ks] = key < N ? qb[(long long)key * ld + I + 4 * ks + fk] : 0.f;
            vf[ks] = key < N ? qb[(long long)key * ld + 2 * I + 4 * ks + fk] : 0.f;
        }
        f32x4 dk[A::DT], dv[A::DT];
#pragma unroll
        for (int dt = 0; dt < A::DT; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qc = 0; qc < ntk; ++qc) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
            const float* qr = img0 + (16 * qc + fr) * A::LD + fk;
            const float* dr = img1 + (16 * qc + fr) * A::LD + fk;
#pragma unroll
            for (int ks = 0; ks < A::KS; ++ks) {
                sa = mfma4(qr[4 * ks], kf[ks], sa);
                da = mfma4(dr[4 * ks], vf[ks], da);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = 16 * qc + 4 * fk + r;
                const float pv = expf(sa[r] * scale - ls[qq]);
                const float dsv = pv * (da[r] - dl[qq]) * scale;
                const float* qrow = img0 + qq * A::LD + fr;
                const float* drow = img1 + qq * A::LD + fr;
#pragma unroll
                for (int dt = 0; dt < A::DT; ++dt) {
                    dv[dt] = mfma4(drow[16 * dt], pv, dv[dt]);
                    dk[dt] = mfma4(qrow[16 * dt], dsv, dk[dt]);
                }
            }
        }
        if (key < N) {
#pragma unroll
            for (int dt = 0; dt < A::DT; ++dt) {
                const int d = 16 * dt + 4 * fk;
                if (d < DH) {
                    *reinterpret_cast<f32x4*>(gb + (long long)key * ld + I + d) = dk[dt];
                    *reinterpret_cast<f32x4*>(gb + (long long)key * ld + 2 * I + d) = dv[dt];
                }
            }
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
// layout of the call being dispatched (set by the entry points below, read where the kernels are launched): tokens in
// the batch when qkv / dqkv are head-major planes, 0 for the rows of to_qkv's output
thread_local long long tl_qT = 0, tl_dT = 0;

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
                       (const bf16_t*)qkv, (bf16_t*)out, lse, (bf16_t*)probs, N, H, scale, tl_qT);
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
                       (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale, tl_qT,
                       tl_dT);
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(dq, bf16)");
    hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DH, NC, HPW>), dim3(B * H / HPW), dim3(ATTN_THREADS), s1, s,
                       (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, (bf16_t*)dqkv, N, H, scale, tl_qT, tl_dT);
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

// compute units the one-workgroup-per-CU kernels may fill: all but those vitmi_set_reserved_cus keeps for a collective
inline int fused_grid_limit(int cus) {
    const int left = cus - vitmi_reserved_cus();
    return left > 8 ? left : 8;
}

// one-kernel backward: heads of 64, 129..208 tokens (9..13 key tiles: the owners, with waves 13..15 on dQ).
// 0 = the dQ + dK/dV pair, 1 = one kernel, 2 = one kernel, operands of the next pair fetched under the current one
// where that variant applies (193..208 tokens).  At B 256 / N 197 / H 12: 260 / 231 / 214 us per launch.
int g_attn_bwd_fused = 2;
inline bool use_fused_bwd(int N, int DH) { return g_attn_bwd_fused && DH == 64 && N > 128 && N <= 16 * FUSED_TASK_WAVE0; }

template <int NC>
int launch_bwd_fused_t(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N,
                       int H, hipStream_t s) {
    const float scale = 1.0f / sqrtf(64.f);
    static unsigned long long done = 0;
    constexpr int NP = NC * 32;
    const size_t sm = (size_t)NP * (3 * 64 * sizeof(bf16_t) + 2 * 32 * sizeof(bf16_t) + 2 * sizeof(float)) + 16;
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_bwd_fused_bf16_kernel<NC>), sm, &done, "vitmi_attn_bwd: LDS attribute (fused)");
    if (rc != VITMI_OK) return rc;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus = fused_grid_limit(cus);
    const int grid = B * H < cus ? B * H : cus;     // one workgroup fills a compute unit
    hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<NC>), dim3(grid), dim3(FUSED_THREADS), sm, s, (const bf16_t*)qkv,
                       (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, B * H, scale, tl_qT, tl_dT);
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(fused, bf16)");
    return VITMI_OK;
}
int launch_bwd_pipe(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N, int H,
                    hipStream_t s) {
    const float scale = 1.0f / sqrtf(64.f);
    static unsigned long long done = 0;
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_bwd_pipe_bf16_kernel), PIPE_LDS, &done, "vitmi_attn_bwd: LDS attribute (pipelined)");
    if (rc != VITMI_OK) return rc;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus = fused_grid_limit(cus);
    const int grid = B * H < cus ? B * H : cus;     // one workgroup fills a compute unit
    hipLaunchKernelGGL(attn_bwd_pipe_bf16_kernel, dim3(grid), dim3(FUSED_THREADS), PIPE_LDS, s, (const bf16_t*)qkv,
                       (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dqkv, N, H, B * H, scale, tl_qT, tl_dT);
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(pipelined, bf16)");
    return VITMI_OK;
}
int launch_bwd_fused(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int N, int H,
                     hipStream_t s) {
    if (g_attn_bwd_fused == 2 && N > 32 * (PIPE_NC - 1) && N <= PIPE_ROWS) return launch_bwd_pipe(qkv, out, dout, lse, dqkv, B, N, H, s);
    switch ((N + 31) / 32) {
        case 5: return launch_bwd_fused_t<5>(qkv, out, dout, lse, dqkv, B, N, H, s);
        case 6: return launch_bwd_fused_t<6>(qkv, out, dout, lse, dqkv, B, N, H, s);
        default: return launch_bwd_fused_t<7>(qkv, out, dout, lse, dqkv, B, N, H, s);
    }
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

// the fp32 kernels on the matrix instruction: heads of 8 / 16 / 32 / 64, up to 18 key tiles, both images in LDS,
// 16-byte-aligned rows (every row of a head starts a multiple of 4 floats into a 16-byte-aligned tensor)
int g_attn_f32_mfma = 1;
template <int DH> constexpr size_t f32_mfma_lds(int nt, bool bwd) {
    return ((size_t)2 * 16 * nt * F32A<DH>::LD + (bwd ? 2 * 16 * nt : 0)) * sizeof(float);
}
inline int f32_mfma_tiles(int N) {
    const int nt = (N + 15) >> 4;
    return nt <= 5 ? 5 : (nt <= 9 ? 9 : (nt <= 13 ? 13 : (nt <= 18 ? 18 : 0)));
}
inline bool f32_mfma_ok(int N, int DH, bool bwd, const void* p0, const void* p1, const void* p2, const void* p3) {
    if (!g_attn_f32_mfma || !(DH == 8 || DH == 16 || DH == 32 || DH == 64)) return false;
    const int nt = f32_mfma_tiles(N);
    if (!nt) return false;
    const size_t ldw = 16 * ((DH + 15) / 16) + 4;
    if (((size_t)2 * 16 * nt * ldw + (bwd ? 2 * 16 * nt : 0)) * sizeof(float) > 160 * 1024) return false;
    const uintptr_t u = reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2) |
                        reinterpret_cast<uintptr_t>(p3);
    return (u & 15) == 0;
}
template <int DH, int NT>
int launch_fwd_f32_mfma(const float* qkv, float* out, float* lse, float* probs, int B, int N, int H, hipStream_t s) {
    static unsigned long long done = 0;
    const size_t sm = f32_mfma_lds<DH>(NT, false);
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_fwd_f32_mfma_kernel<DH, NT>), sm, &done, "vitmi_attn_fwd: LDS attribute (f32)");
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL((attn_fwd_f32_mfma_kernel<DH, NT>), dim3(B * H), dim3(256), sm, s, qkv, out, lse, probs, N, H,
                       1.0f / sqrtf((float)DH));
    VITMI_LAUNCH_CHECK("vitmi_attn_fwd(f32, mfma)");
    return VITMI_OK;
}
template <int DH, int NT>
int launch_bwd_f32_mfma(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int N, int H,
                        hipStream_t s) {
    static unsigned long long done = 0;
    const size_t sm = f32_mfma_lds<DH>(NT, true);
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_bwd_f32_mfma_kernel<DH, NT>), sm, &done, "vitmi_attn_bwd: LDS attribute (f32)");
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL((attn_bwd_f32_mfma_kernel<DH, NT>), dim3(B * H), dim3(256), sm, s, qkv, dout, lse, dqkv, N, H,
                       1.0f / sqrtf((float)DH));
    VITMI_LAUNCH_CHECK("vitmi_attn_bwd(f32, mfma)");
    return VITMI_OK;
}
#define DISPATCH_F32_MFMA(FN, ...)                                     \
    do {                                                               \
        const int nt_ = f32_mfma_tiles(N);                             \
        switch (DH * 100 + nt_) {                                      \
            case 805: return FN<8, 5>(__VA_ARGS__);                    \
            case 809: return FN<8, 9>(__VA_ARGS__);                    \
            case 813: return FN<8, 13>(__VA_ARGS__);                   \
            case 818: return FN<8, 18>(__VA_ARGS__);                   \
            case 1605: return FN<16, 5>(__VA_ARGS__);                  \
            case 1609: return FN<16, 9>(__VA_ARGS__);                  \
            case 1613: return FN<16, 13>(__VA_ARGS__);                 \
            case 1618: return FN<16, 18>(__VA_ARGS__);                 \
            case 3205: return FN<32, 5>(__VA_ARGS__);                  \
            case 3209: return FN<32, 9>(__VA_ARGS__);                  \
            case 3213: return FN<32, 13>(__VA_ARGS__);                 \
            case 3218: return FN<32, 18>(__VA_ARGS__);                 \
            case 6405: return FN<64, 5>(__VA_ARGS__);                  \
            case 6409: return FN<64, 9>(__VA_ARGS__);                  \
            case 6413: return FN<64, 13>(__VA_ARGS__);                 \
            default: return FN<64, 18>(__VA_ARGS__);                   \
        }                                                              \
    } while (0)

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

extern "C" int vitmi_debug_set_attn_f32_mfma(int on) {
    g_attn_f32_mfma = on ? 1 : 0;
    return VITMI_OK;
}

extern "C" int vitmi_debug_set_attn_bwd_fused(int on) {
    g_attn_bwd_fused = on;
    return VITMI_OK;
}

// head-major planes exist for the bf16 kernels with heads of 64 (one (image, head) pair per workgroup)
static int check_layout(const char* who, int layout, int dt, int DH, bool dprobs) {
    VITMI_CHECK_ARG(layout >= 0 && layout <= 3, "%s: bad layout %d", who, layout);
    if (layout && (dt != VITMI_BF16 || DH != 64 || dprobs)) {
        vitmi_set_error("%s: the head-major qkv layout needs bf16 heads of 64 (and no gradient on the probabilities)", who);
        return VITMI_ERR_UNSUPPORTED;
    }
    return VITMI_OK;
}

extern "C" int vitmi_attn_fwd(const void* qkv, void* out, float* lse, void* probs, int dt, int B, int N, int H, int DH,
                              int layout, void* stream) {
    VITMI_CHECK_ARG(qkv && out && lse, "vitmi_attn_fwd: null pointer");
    int rc = check_common("vitmi_attn_fwd", dt, B, N, H, DH);
    if (rc != VITMI_OK) return rc;
    rc = check_layout("vitmi_attn_fwd", layout, dt, DH, false);
    if (rc != VITMI_OK) return rc;
    tl_qT = (layout & VITMI_QKV_HEAD_MAJOR) ? (long long)B * N : 0;
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_ATTN_FWD, s, 4.0 * B * H * (double)N * N * DH);
    if (dt == VITMI_F32 && f32_mfma_ok(N, DH, false, qkv, out, nullptr, nullptr))
        DISPATCH_F32_MFMA(launch_fwd_f32_mfma, (const float*)qkv, (float*)out, lse, (float*)probs, B, N, H, s);
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
    if (dt == VITMI_F32 && f32_mfma_ok(N, DH, true, qkv, dout, dqkv, nullptr))
        DISPATCH_F32_MFMA(launch_bwd_f32_mfma, (const float*)qkv, (const float*)dout, lse, (float*)dqkv, B, N, H, s);
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
    if (use_fused_bwd(N, DH)) return launch_bwd_fused(qkv, out, dout, lse, dqkv, B, N, H, s);
    if (nc > 8) DISPATCH_NC_LONG(launch_bwd, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
    switch (DH) {
        case 8: DISPATCH_NC(launch_bwd, 8, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        case 16: DISPATCH_NC(launch_bwd, 16, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        case 32: DISPATCH_NC(launch_bwd, 32, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
        default: DISPATCH_NC(launch_bwd, 64, qkv, out, dout, lse, delta, dqkv, B, N, H, s)
    }
}

extern "C" int vitmi_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const void* dprobs,
                              void* dqkv, void* workspace, int dt, int B, int N, int H, int DH, int layout, void* stream) {
    VITMI_CHECK_ARG(qkv && lse && dqkv, "vitmi_attn_bwd: null pointer");
    VITMI_CHECK_ARG(dout || dprobs, "vitmi_attn_bwd: neither dout nor dprobs");
    VITMI_CHECK_ARG(!dout || out, "vitmi_attn_bwd: dout needs out");
    int rc = check_common("vitmi_attn_bwd", dt, B, N, H, DH);
    if (rc != VITMI_OK) return rc;
    rc = check_layout("vitmi_attn_bwd", layout, dt, DH, dprobs != nullptr);
    if (rc != VITMI_OK) return rc;
    tl_qT = (layout & VITMI_QKV_HEAD_MAJOR) ? (long long)B * N : 0;
    tl_dT = (layout & VITMI_DQKV_HEAD_MAJOR) ? (long long)B * N : 0;
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

// ---------------------------------------------------------------- one-launch attention sub-block forward (small geometry)
namespace {
size_t attn_small_lds(int D, int H, int DH, int NC, int MT) {
    return ((size_t)MT * 16 * (D + SM_PAD) + (size_t)3 * H * NC * 32 * DH + 8 + (size_t)MT * 16 * (H * DH + SM_PAD)) * sizeof(bf16_t);
}
template <int DH, int NC, int MT>
int launch_attn_small(const AttnSmallArgs& a, int B, hipStream_t s) {
    static unsigned long long done = 0;
    const size_t lds = attn_small_lds(a.D, a.H, DH, NC, MT);
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_small_fwd_kernel<DH, NC, MT>), LDS_BYTES, &done,
                        "vitmi_attn_block_fwd: LDS attribute");
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL((attn_small_fwd_kernel<DH, NC, MT>), dim3(B), dim3(SM_THREADS), lds, s, a);
    VITMI_LAUNCH_CHECK("attn_small_fwd_kernel");
    return VITMI_OK;
}
template <int DH>
int dispatch_attn_small(const AttnSmallArgs& a, int B, hipStream_t s) {
    const int mt = (a.N + 15) / 16;
    if (mt <= 2) return launch_attn_small<DH, 1, 2>(a, B, s);
    if (mt == 3) return launch_attn_small<DH, 2, 3>(a, B, s);
    if (mt == 4) return launch_attn_small<DH, 2, 4>(a, B, s);
    if (mt == 5) return launch_attn_small<DH, 3, 5>(a, B, s);
    return launch_attn_small<DH, 3, 6>(a, B, s);
}
inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

// whether vitmi_attn_block_fwd may run as the one-launch kernel (shape part; the operands' alignment is checked at the call)
bool vitmi_attn_small_ok(int dt, int B, int N, int D, int H, int DH) {
    if (dt != VITMI_BF16 || B <= 0 || N <= 0 || N > 96) return false;
    if (DH != 8 && DH != 16) return false;
    const int I = H * DH;
    if (D < 32 || D > 256 || D % 32 != 0 || I % 32 != 0 || I > 256) return false;
    const int mt = (N + 15) / 16 < 2 ? 2 : (N + 15) / 16;
    return attn_small_lds(D, H, DH, (mt + 1) / 2, mt) <= LDS_BYTES;
}

// returns 1000 (nothing enqueued) when an operand is not 16-byte aligned
int vitmi_attn_small_fwd(int B, int N, int D, int H, int DH, const void* x, const float* gamma, const float* beta, float eps,
                         const void* w_qkv, const float* b_qkv, const void* w_out, const float* b_out, void* xn, float* mean,
                         float* rstd, void* qkv, void* o, float* lse, void* y, void* stream) {
    if (!(al16p(x) && al16p(gamma) && al16p(beta) && al16p(w_qkv) && al16p(b_qkv) && al16p(w_out) && al16p(b_out) && al16p(xn) &&
          al16p(qkv) && al16p(o) && al16p(y)))
        return 1000;
    AttnSmallArgs a{(const bf16_t*)x, gamma, beta, (const bf16_t*)w_qkv, b_qkv, (const bf16_t*)w_out, b_out, (bf16_t*)xn,
                    (bf16_t*)qkv, (bf16_t*)o, (bf16_t*)y, mean, rstd, lse, N, D, H, eps};
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_ATTN_FWD, s, 4.0 * B * H * (double)N * N * DH);
    return DH == 8 ? dispatch_attn_small<8>(a, B, s) : dispatch_attn_small<16>(a, B, s);
}

// ---------------------------------------------------------------- one-launch attention sub-block backward (small geometry)
int vitmi_small_fused_mask();   // small.hip
namespace {
size_t attn_small_bwd_lds(int D, int H, int DH, int NC, int MT) {
    const int I = H * DH, ld = (D > 3 * I ? D : 3 * I) + SM_PAD;
    return ((size_t)MT * 16 * ld + (size_t)4 * H * NC * 32 * DH + 8) * sizeof(bf16_t) +
           ((size_t)2 * H * NC * 32 + (size_t)2 * SM_WAVES * MT * 16) * sizeof(float);
}
template <int DH, int NC, int MT>
int launch_attn_small_bwd(const AttnSmallBwdArgs& a, int B, hipStream_t s) {
    static unsigned long long done = 0;
    const size_t lds = attn_small_bwd_lds(a.D, a.H, DH, NC, MT);
    int rc = ensure_lds(reinterpret_cast<const void*>(&attn_small_bwd_kernel<DH, NC, MT>), LDS_BYTES, &done,
                        "vitmi_attn_block_bwd: LDS attribute");
    if (rc != VITMI_OK) return rc;
    hipLaunchKernelGGL((attn_small_bwd_kernel<DH, NC, MT>), dim3(B), dim3(SM_THREADS), lds, s, a);
    VITMI_LAUNCH_CHECK("attn_small_bwd_kernel");
    return VITMI_OK;
}
template <int DH>
int dispatch_attn_small_bwd(const AttnSmallBwdArgs& a, int B, hipStream_t s) {
    const int mt = (a.N + 15) / 16;
    if (mt <= 2) return launch_attn_small_bwd<DH, 1, 2>(a, B, s);
    if (mt == 3) return launch_attn_small_bwd<DH, 2, 3>(a, B, s);
    if (mt == 4) return launch_attn_small_bwd<DH, 2, 4>(a, B, s);
    if (mt == 5) return launch_attn_small_bwd<DH, 3, 5>(a, B, s);
    return launch_attn_small_bwd<DH, 3, 6>(a, B, s);
}
}  // namespace

// whether vitmi_attn_block_bwd may run as the one-launch kernel when the caller forms the weight gradients (bit 4 of
// the small-geometry mask; shape part: the operands' alignment is checked at the call)
bool vitmi_attn_small_bwd_ok(int dt, int B, int N, int D, int H, int DH) {
    if (!(vitmi_small_fused_mask() & 16) || dt != VITMI_BF16 || B <= 0 || N <= 0 || N > 96) return false;
    if (DH != 8 && DH != 16) return false;
    const int I = H * DH;
    if (D < 32 || D > 256 || D % 32 != 0 || I % 32 != 0 || I > 256) return false;
    const int mt = (N + 15) / 16 < 2 ? 2 : (N + 15) / 16;
    return attn_small_bwd_lds(D, H, DH, (mt + 1) / 2, mt) <= LDS_BYTES;
}
// bytes of the per-image partial sums the kernel leaves for the final reduction (ff_small_bwd_final_kernel's layout
// without a hidden part: [B][dgamma | dbeta | output-bias gradient])
long long vitmi_attn_small_bwd_partial_bytes(int B, int D) { return B > 0 && D > 0 ? (long long)B * 3 * D * (long long)sizeof(float) : 0; }

// returns 1000 (nothing enqueued) when an operand is not 16-byte aligned
int vitmi_attn_small_bwd(int B, int N, int D, int H, int DH, const void* dy, const void* x, const void* qkv, const void* o,
                         const float* lse, const float* gamma, const float* mean, const float* rstd, const void* w_out_t,
                         const void* w_qkv_t, void* dqkv, void* dx, float* partial, void* stream) {
    if (!(al16p(dy) && al16p(x) && al16p(qkv) && al16p(o) && al16p(gamma) && al16p(w_out_t) && al16p(w_qkv_t) && al16p(dqkv) &&
          al16p(dx) && al16p(partial)))
        return 1000;
    AttnSmallBwdArgs a{(const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)qkv, (const bf16_t*)o, lse, gamma, mean, rstd,
                       (const bf16_t*)w_out_t, (const bf16_t*)w_qkv_t, (bf16_t*)dqkv, (bf16_t*)dx, partial, N, D, H};
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_ATTN_BWD, s, 10.0 * B * H * (double)N * N * DH);
    return DH == 8 ? dispatch_attn_small_bwd<8>(a, B, s) : dispatch_attn_small_bwd<16>(a, B, s);
}
