// GEMM kernels of the ViT encoder path.
//
//   C(m,n) = sum_k A(m,k) * B(n,k)   [+ bias[n]] [GELU / dGELU] [+ residual(m,n)]
//
// bf16: 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
//       v_mfma_f32_16x16x32_bf16 accumulators; operands are staged global -> VGPR -> LDS with
//       the next tile's global loads in flight under the current tile's MFMAs; the LDS image is
//       always [row][k] with a 16-byte row pad (144-B rows: conflict-free ds_read_b128), whatever
//       the operand's memory layout (an operand that is contiguous along its row index is
//       transposed on its way into LDS).
// fp32: 64x64x16 tiles on v_mfma_f32_16x16x4_f32 (IEEE fp32 products and sums: the exact parity path) with the same
//       epilogue; products with few output tiles (weight gradients) split K over blockIdx.z into slabs summed in order.
//
// The patch-embedding GEMMs reuse the bf16/fp32 cores with a loader that gathers patch
// elements straight from the NCHW image (no im2col buffer).
#include "common.h"

// fast paths for aligned bf16 problems (gemm_fast.hip)
bool vitmi_gemm_nt_fast_ok(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb,
                           const void* C, long long ldc, int c_dt, const void* res, long long ldr, const void* aux,
                           long long ldaux);
int vitmi_gemm_nt_fast(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                       long long ldc, int c_dt, int epilogue, const float* bias, const void* residual, long long ldr,
                       void* aux, long long ldaux, float* cs_partial, hipStream_t s);
int vitmi_gemm_nt_fast_ln(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                          long long ldc, int epilogue, const float* bias, const float* ln_mean, const float* ln_rstd,
                          const float* ln_s, hipStream_t s);
bool vitmi_gemm_nt_phased_ok(int M, int N, int K);
int vitmi_gemm_nt_colsum_rows(int M);
int vitmi_colsum_final_launch(const float* partial, int chunks, int cols, float* out, hipStream_t s);
long long vitmi_colsum_workspace_bytes_impl(long long rows, int cols);
bool vitmi_gemm_tn_fast_ok(int P, int Q, const void* U, long long ldu, const void* V, long long ldv, const void* C,
                           long long ldc);
int vitmi_gemm_tn_slabs(int P, int Q, int T);
int vitmi_gemm_tn_slabs_bound(int P, int Q, int T);
int vitmi_gemm_tn_fast(int P, int Q, int T, const void* U, long long ldu, const void* V, long long ldv, float* C,
                       long long ldc, float* slabs, hipStream_t s);
long long vitmi_gemm_tn_batch_bytes(int n, const int* P, const int* Q, int T);
long long vitmi_gemm_tn_batch_bytes_bound(int n, const int* P, const int* Q, int T);
int vitmi_gemm_tn_batch(int n, const int* P, const int* Q, int T, const void* const* U, const long long* ldu,
                        const void* const* V, const long long* ldv, float* const* C, float* partial, hipStream_t s,
                        const int* u_planes);
int vitmi_gemm_tn_fast_planes(int P, int Q, int T, const void* U, const void* V, long long ldv, float* C, long long ldc,
                              float* slabs, hipStream_t s);
int vitmi_gemm_nt_fast_planes(int M, int N, int K, const void* A, long long lda, const void* B, long long ldb, void* C,
                              long long ldc, const float* bias, int out_planes, int a_planes, hipStream_t s);
bool vitmi_patch_fast_ok(const void* img, const void* w, const void* out, int W, int p, int D, int E,
                         long long out_batch_stride);
int vitmi_patch_fwd_fast(const void* img, const void* w, const float* bias, void* out, int B, int C, int H, int W,
                         int p, int pw, int D, long long out_batch_stride, hipStream_t s);
int vitmi_patch_bwd_w_fast(const void* img, const void* dy, float* dw, float* slabs, int B, int C, int H, int W, int p,
                           int pw, int D, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_LD = BK + 8;  // elements per LDS row (144 B)
constexpr int NTHREADS = 256;
constexpr int CHUNKS = (BM * BK / 8) / NTHREADS;  // 16-byte chunks per thread per operand tile = 4

struct Epilogue {
    void* C;
    long long ldc;
    const float* bias;
    const void* res;
    long long ldr;
    void* aux;
    long long ldaux;
    // patch-embedding forward writes token j of image b at C + b*batch_stride + j*ldc
    int tokens_per_image;  // 0 = plain row-major
    long long batch_stride;
};

template <int EPI, typename T, typename OutT>
__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float v) {
    if (e.bias) v += e.bias[n];
    if (epi_is_gelu(EPI)) {
        if (e.aux)
            reinterpret_cast<T*>(e.aux)[(long long)m * e.ldaux + n] =
                EPI == VITMI_EPI_GELU_D ? deriv_pack(gelu_erf_grad(v), T()) : from_f32<T>(v);
        v = gelu_erf(v);
    } else if (epi_is_auxmul(EPI)) {
        const T a = reinterpret_cast<const T*>(e.aux)[(long long)m * e.ldaux + n];
        v *= (EPI == VITMI_EPI_MULAUX) ? deriv_unpack(a) : gelu_erf_grad(to_f32(a));
    }
    if (e.res) v += to_f32(reinterpret_cast<const T*>(e.res)[(long long)m * e.ldr + n]);
    long long off;
    if (e.tokens_per_image > 0) {
        const int b = m / e.tokens_per_image, j = m - b * e.tokens_per_image;
        off = (long long)b * e.batch_stride + (long long)j * e.ldc + n;
    } else {
        off = (long long)m * e.ldc + n;
    }
    reinterpret_cast<OutT*>(e.C)[off] = from_f32<OutT>(v);
}

// ------------------------------------------------------------------------------------------------
// Operand loaders.  A tile is ROWS(=128) x BK(=64) bf16 in LDS, [row][k].  Every thread owns 4
// 16-byte chunks.  KC: a chunk is 8 consecutive k of one row.  MC: a chunk is 8 consecutive rows
// at one k (the operand is contiguous along its row index) and is scattered into LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (bf16_t)0.f;
    return z;
}

template <bool KC>
struct StridedLoader {
    const bf16_t* p;
    long long s_r, s_k;  // element strides along the row index and along k
    int R, K;            // bounds
    int vec;             // 16-byte loads allowed (alignment checked on the host)

    __device__ __forceinline__ void load(int r0, int k0, bf16x8 (&v)[CHUNKS]) const {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = threadIdx.x + i * NTHREADS;
            if (KC) {
                const int r = r0 + (c >> 3), k = k0 + (c & 7) * 8;
                if (r < R && k + 8 <= K && vec) {
                    v[i] = *reinterpret_cast<const bf16x8*>(p + (long long)r * s_r + k);
                } else {
                    bf16x8 t = zero8();
                    if (r < R) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (k + j < K) t[j] = p[(long long)r * s_r + (long long)(k + j) * s_k];
                    }
                    v[i] = t;
                }
            } else {
                const int k = k0 + (c >> 4), r = r0 + (c & 15) * 8;
                if (k < K && r + 8 <= R && vec) {
                    v[i] = *reinterpret_cast<const bf16x8*>(p + (long long)k * s_k + r);
                } else {
                    bf16x8 t = zero8();
                    if (k < K) {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (r + j < R) t[j] = p[(long long)(r + j) * s_r + (long long)k * s_k];
                    }
                    v[i] = t;
                }
            }
        }
    }
    __device__ __forceinline__ static void store(bf16_t* lds, const bf16x8 (&v)[CHUNKS]) {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = threadIdx.x + i * NTHREADS;
            if (KC) {
                *reinterpret_cast<bf16x8*>(lds + (c >> 3) * LDS_LD + (c & 7) * 8) = v[i];
            } else {
                const int k = c >> 4, r = (c & 15) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) lds[(r + j) * LDS_LD + k] = v[i][j];
            }
        }
    }
};

// Patch operand: logical matrix P[token m][element e], m = (b, gy, gx), e = (c, i, j),
// P = x[b, c, gy*ps + i, gx*ps + j].  TOKEN_ROWS: tile rows are tokens and k runs over patch
// elements (forward).  !TOKEN_ROWS: tile rows are patch elements and k runs over tokens
// (weight gradient); chunks still run along e, so they are scattered into LDS.
template <typename XT, bool TOKEN_ROWS>
struct PatchLoader {
    const XT* x;
    int C, H, W, ps, gw, gh;
    int tokens, elems;  // B*gh*gw, C*ps*ps

    __device__ __forceinline__ long long addr(int m, int e) const {
        const int per = gh * gw;
        const int b = m / per, t = m - b * per;
        const int gy = t / gw, gx = t - gy * gw;
        const int c = e / (ps * ps), q = e - c * ps * ps;
        const int i = q / ps, j = q - i * ps;
        return (((long long)b * C + c) * H + (gy * ps + i)) * W + gx * ps + j;
    }
    __device__ __forceinline__ bf16x8 chunk(int m, int e0) const {
        bf16x8 t = zero8();
        if (m >= tokens) return t;
        if ((ps & 7) == 0 && e0 + 8 <= elems) {
            // 8 elements of one patch row: contiguous in the image
            const XT* src = x + addr(m, e0);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = (bf16_t)to_f32(src[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (e0 + j < elems) t[j] = (bf16_t)to_f32(x[addr(m, e0 + j)]);
        }
        return t;
    }
    __device__ __forceinline__ void load(int r0, int k0, bf16x8 (&v)[CHUNKS]) const {
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const int c = threadIdx.x + i * NTHREADS;
            if (TOKEN_ROWS)
                v[i] = chunk(r0 + (c >> 3), k0 + (c & 7) * 8);
            else
                v[i] = chunk(k0 + (c >> 4), r0 + (c & 15) * 8);
        }
    }
    __device__ __forceinline__ static void store(bf16_t* lds, const bf16x8 (&v)[CHUNKS]) {
        StridedLoader<TOKEN_ROWS>::store(lds, v);
    }
};

template <int EPI, typename OutT, class LoaderA, class LoaderB>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(LoaderA la, LoaderB lb, Epilogue epi, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) bf16_t As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[BN * LDS_LD];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[CHUNKS], rb[CHUNKS];
    la.load(m0, 0, ra);
    lb.load(n0, 0, rb);
    const int nk = (K + BK - 1) / BK;
    for (int kt = 0; kt < nk; ++kt) {
        LoaderA::store(As, ra);
        LoaderB::store(Bs, rb);
        __syncthreads();
        if (kt + 1 < nk) {
            la.load(m0, (kt + 1) * BK, ra);
            lb.load(n0, (kt + 1) * BK, rb);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 16 + fr) * LDS_LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 64 + j * 16 + fr) * LDS_LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // accumulator element r of tile (i,j): row = 4*fg + r, col = fr
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fr;
                if (n < N) epilogue_store<EPI, bf16_t, OutT>(epi, m, n, acc[i][j][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 path: 64x64x16 tiles, 256 threads, 4x4 outputs per thread, generic strides.
// ------------------------------------------------------------------------------------------------
struct F32Operand {
    const float* p;
    long long s_r, s_k;
    int R;
    // patch gather (s_r/s_k unused when patch != 0)
    int patch, token_rows, C, H, W, ps, gw, gh;
    __device__ __forceinline__ float at(int r, int k, int K) const {
        if (r >= R || k >= K) return 0.f;
        if (!patch) return p[(long long)r * s_r + (long long)k * s_k];
        const int m = token_rows ? r : k, e = token_rows ? k : r;
        const int per = gh * gw;
        const int b = m / per, t = m - b * per;
        const int gy = t / gw, gx = t - gy * gw;
        const int c = e / (ps * ps), q = e - c * ps * ps;
        const int i = q / ps, j = q - i * ps;
        return p[(((long long)b * C + c) * H + (gy * ps + i)) * W + gx * ps + j];
    }
};

// 64 x 64 output tile per 4-wave block, each wave a 32 x 32 quarter as 2 x 2 v_mfma_f32_16x16x4_f32 tiles (IEEE fp32
// products and sums: the exact-arithmetic mode keeps its 1e-4 bar).  Operand fragments of that instruction: lane l
// supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] of a 16 x 4 / 4 x 16 step and receives D[4 (l >> 4) + i][l & 15].
// Both tiles sit in LDS as [row][k], 32-deep, rows 36 floats apart.
// blockIdx.z walks K in chunks of k_chunk (weight gradients: K = tokens, few output tiles): slab z of the partial
// products goes to C + z * slab_stride and f32_slab_sum_kernel adds the slabs in index order.
//
// A thread fetches its share of a K-tile as 16-byte pieces along whichever index its operand is contiguous in (k for
// x and the weights of a forward product, the row for the column-major views of the backward products), when base,
// strides and extent allow it (va / vb), and as single elements otherwise.  The next K-tile is in flight while the
// current one's products are issued.
template <int EPI, int T = 64>
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32Operand A, F32Operand B, Epilogue epi, int M, int N, int K,
                                                       int k_chunk, long long slab_stride, int va, int vb) {
    // T x T output tile: 128 for large products (each wave a 64 x 64 quarter, 4 x 4 MFMA tiles: half the LDS reads
    // per product of the 64 tile), 64 (2 x 2), or, for products of few tiles, 32 (each wave one 16 x 16 tile: four
    // times the blocks)
    constexpr int TK = 32, LD = TK + 4, PER = T * TK / 256, PV = PER / 4, SUB = T / 32;
    __shared__ __attribute__((aligned(16))) float As[T][LD];
    __shared__ __attribute__((aligned(16))) float Bs[T][LD];
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int kz0 = blockIdx.z * k_chunk, kz1 = min(K, kz0 + k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (T / 2), wn = (wave & 1) * (T / 2);
    const int fr = lane & 15, fk = lane >> 4;
    const bool a_kc = A.patch ? (A.token_rows != 0) : (A.s_k == 1);
    const bool b_kc = B.patch ? (B.token_rows != 0) : (B.s_k == 1);
    f32x4 acc[SUB][SUB];
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
        for (int j = 0; j < SUB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // piece i of a thread: 4 consecutive k of one row (kc) or 4 consecutive rows at one k; as scalars: element
    // c = tid + 256 i in the same (row, k) enumeration the single-element form always used
    auto piece_at = [&](bool kc, int i, int& r, int& k) {
        const int c = threadIdx.x + i * 256;
        if (kc) { k = 4 * (c & 7); r = c >> 3; } else { r = 4 * (c % (T / 4)); k = c / (T / 4); }
    };
    auto elem_at = [&](bool kc, int i, int& r, int& k) {
        const int c = threadIdx.x + i * 256;
        if (kc) { k = c & 31; r = c >> 5; } else { r = c % T; k = c / T; }
    };
    auto fetch_op = [&](const F32Operand& X, bool kc, int vec, int row0, int k0, float (&dst)[PER]) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                int r, k;
                piece_at(kc, i, r, k);
                const int gr = row0 + r, gk = k0 + k;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kc ? (gr < X.R && gk + 3 < kz1) : (gr + 3 < X.R && gk < kz1)) {
                    v = *reinterpret_cast<const f32x4*>(X.p + (long long)gr * X.s_r + (long long)gk * X.s_k);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = kc ? X.at(gr, gk + j, kz1) : X.at(gr + j, gk, kz1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[4 * i + j] = v[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int r, k;
                elem_at(kc, i, r, k);
                dst[i] = X.at(row0 + r, k0 + k, kz1);
            }
        }
    };
    auto put_op = [&](float (&S)[T][LD], bool kc, int vec, const float (&src)[PER]) {
        if (vec) {
#pragma unroll
            for (int i = 0; i < PV; ++i) {
                int r, k;
                piece_at(kc, i, r, k);
                if (kc) {
                    *reinterpret_cast<f32x4*>(&S[r][k]) = f32x4{src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) S[r + j][k] = src[4 * i + j];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                int r, k;
                elem_at(kc, i, r, k);
                S[r][k] = src[i];
            }
        }
    };
    float av[PER], bv[PER];
    if (kz0 < kz1) {
        fetch_op(A, a_kc, va, m0, kz0, av);
        fetch_op(B, b_kc, vb, n0, kz0, bv);
    }
    for (int k0 = kz0; k0 < kz1; k0 += TK) {
        __syncthreads();  // the previous tile's fragments have been read
        put_op(As, a_kc, va, av);
        put_op(Bs, b_kc, vb, bv);
        __syncthreads();
        if (k0 + TK < kz1) {
            fetch_op(A, a_kc, va, m0, k0 + TK, av);
            fetch_op(B, b_kc, vb, n0, k0 + TK, bv);
        }
        // A lane reads its row's k = 8 fk .. 8 fk + 7 as two 16-byte pieces; step (h, j) of the eight takes element
        // 4 h + j from every lane, so the four lane groups supply k = 8 fk + 4 h + j, four different k, to both
        // operands alike (a sum may take its index in any order).  Rows are 36 floats apart: the sixteen rows of a
        // piece fall on all 64 banks once.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a[SUB], b[SUB];
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(&As[wm + 16 * i + fr][8 * fk + 4 * h]);
                b[i] = *reinterpret_cast<const f32x4*>(&Bs[wn + 16 * i + fr][8 * fk + 4 * h]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < SUB; ++i)
#pragma unroll
                    for (int jj = 0; jj < SUB; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b[jj][j], acc[i][jj], 0, 0, 0);
        }
    }
    if (slab_stride > 0) epi.C = reinterpret_cast<float*>(epi.C) + (long long)blockIdx.z * slab_stride;
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm + i * 16 + 4 * fk + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int n = n0 + wn + j * 16 + fr;
                if (n < N) epilogue_store<EPI, float, float>(epi, m, n, acc[i][j][r]);
            }
        }
}

// 16-byte pieces of an fp32 operand: contiguous along k (s_k == 1) or along the row (s_r == 1), the other stride a
// multiple of 4 elements, the base 16-byte aligned; patch operands and everything else go element by element
inline int f32_vec_ok(const F32Operand& x) {
    if (x.patch || (reinterpret_cast<uintptr_t>(x.p) & 15)) return 0;
    if (x.s_k == 1) return x.s_r % 4 == 0;
    if (x.s_r == 1) return x.s_k % 4 == 0;
    return 0;
}

// out[m * ldc + n] = sum_z slabs[z][m][n] (dense M x N slabs), in slab order
__global__ void f32_slab_sum_kernel(const float* __restrict__ slabs, int nslabs, int M, int N, float* __restrict__ out,
                                    long long ldc) {
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float s = slabs[i];
        for (int z = 1; z < nslabs; ++z) s += slabs[(long long)z * total + i];
        out[(i / N) * ldc + (i % N)] = s;
    }
}

// The same product for operands that can be fetched 16 bytes at a time throughout: every operand either contiguous
// along k (AKC / BKC: the activations and weights of a forward product) or along its row index (the column-major
// views the backward products take), bases 16-byte aligned, the other stride a multiple of 4, K and the K chunk
// multiples of 32, row counts of row-contiguous operands multiples of 4.  Per-thread source pointers are set up once
// and advanced by a constant, so the K-loop carries no index arithmetic and the kernel needs half the registers of
// the generic one; rows past the edge are fetched as zeros.
template <int EPI, int T, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_f32_vec_kernel(const float* __restrict__ Ap, long long a_ld, int AR,
                                                           const float* __restrict__ Bp, long long b_ld, int BR,
                                                           Epilogue epi, int M, int N, int K, int k_chunk,
                                                           long long slab_stride) {
    constexpr int TK = 32, LD = TK + 4, PV = T * TK / 1024, SUB = T / 32;
    __shared__ __attribute__((aligned(16))) float As[T][LD];
    __shared__ __attribute__((aligned(16))) float Bs[T][LD];
    const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
    const int kz0 = blockIdx.z * k_chunk, kz1 = min(K, kz0 + k_chunk);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (T / 2), wn = (wave & 1) * (T / 2);
    const int fr = lane & 15, fk = lane >> 4;
    f32x4 acc[SUB][SUB];
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
        for (int j = 0; j < SUB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // piece i of a thread: 4 consecutive k of one row (k-contiguous operand) or 4 consecutive rows at one k
    auto piece = [&](bool kc, int i, int& r, int& k) {
        const int c = threadIdx.x + i * 256;
        if (kc) { k = 4 * (c & 7); r = c >> 3; } else { r = 4 * (c % (T / 4)); k = c / (T / 4); }
    };
    const float* ap[PV];
    const float* bp[PV];
    unsigned aok = 0, bok = 0;
#pragma unroll
    for (int i = 0; i < PV; ++i) {
        int r, k;
        piece(AKC, i, r, k);
        ap[i] = AKC ? Ap + (long long)(m0 + r) * a_ld + kz0 + k : Ap + (m0 + r) + (long long)(kz0 + k) * a_ld;
        if (m0 + r < AR) aok |= 1u << i;
        piece(BKC, i, r, k);
        bp[i] = BKC ? Bp + (long long)(n0 + r) * b_ld + kz0 + k : Bp + (n0 + r) + (long long)(kz0 + k) * b_ld;
        if (n0 + r < BR) bok |= 1u << i;
    }
    const long long a_step = AKC ? TK : TK * a_ld, b_step = BKC ? TK : TK * b_ld;
    f32x4 av[PV], bv[PV];
    auto fetch = [&]() {
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            av[i] = (aok >> i) & 1 ? *reinterpret_cast<const f32x4*>(ap[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[i] = (bok >> i) & 1 ? *reinterpret_cast<const f32x4*>(bp[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            ap[i] += a_step;
            bp[i] += b_step;
        }
    };
    if (kz0 < kz1) fetch();
    for (int k0 = kz0; k0 < kz1; k0 += TK) {
        __syncthreads();  // the previous tile's fragments have been read
#pragma unroll
        for (int i = 0; i < PV; ++i) {
            int r, k;
            piece(AKC, i, r, k);
            if (AKC) {
                *reinterpret_cast<f32x4*>(&As[r][k]) = av[i];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) As[r + j][k] = av[i][j];
            }
            piece(BKC, i, r, k);
            if (BKC) {
                *reinterpret_cast<f32x4*>(&Bs[r][k]) = bv[i];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) Bs[r + j][k] = bv[i][j];
            }
        }
        __syncthreads();
        if (k0 + TK < kz1) fetch();
        // fragment order as in gemm_f32_kernel: step (h, j) takes k = 8 fk + 4 h + j from the lane groups
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 a[SUB], b[SUB];
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(&As[wm + 16 * i + fr][8 * fk + 4 * h]);
                b[i] = *reinterpret_cast<const f32x4*>(&Bs[wn + 16 * i + fr][8 * fk + 4 * h]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < SUB; ++i)
#pragma unroll
                    for (int jj = 0; jj < SUB; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], b[jj][j], acc[i][jj], 0, 0, 0);
        }
    }
    if (slab_stride > 0) epi.C = reinterpret_cast<float*>(epi.C) + (long long)blockIdx.z * slab_stride;
#pragma unroll
    for (int i = 0; i < SUB; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm + i * 16 + 4 * fk + r;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int n = n0 + wn + j * 16 + fr;
                if (n < N) epilogue_store<EPI, float, float>(epi, m, n, acc[i][j][r]);
            }
        }
}

// 0: not eligible; 1: k-contiguous; 2: row-contiguous
inline int f32_fast_layout(const F32Operand& x) {
    if (x.patch || (reinterpret_cast<uintptr_t>(x.p) & 15)) return 0;
    if (x.s_k == 1 && x.s_r % 4 == 0) return 1;
    if (x.s_r == 1 && x.s_k % 4 == 0 && x.R % 4 == 0) return 2;
    return 0;
}

// Tile choice.  Below F32_SMALL_BELOW 64 x 64 tiles (four blocks per CU) a product runs on 32 x 32 tiles: a block's
// K-loop is a chain of load latencies, and four times the blocks is four times the chains in flight (cfg 1 step:
// 4.94 -> 4.49 ms).  From F32_BIG_FROM 128 x 128 tiles on (two blocks per CU) the 16-byte kernel runs on those: half
// the LDS reads per product.
constexpr long long F32_SMALL_BELOW = 1024;
constexpr long long F32_BIG_FROM = 512;

template <int EPI, int T>
inline void launch_f32_tile(const F32Operand& a, const F32Operand& b, const Epilogue& epi, int M, int N, int K, int chunk,
                            long long slab_stride, int splits, hipStream_t s) {
    const dim3 grid(ceil_div(N, T), ceil_div(M, T), splits);
    const int la = f32_fast_layout(a), lb = f32_fast_layout(b);
    if (la && lb && K % 32 == 0 && chunk % 32 == 0 && !(la == 2 && lb == 1)) {
        const long long ald = la == 1 ? a.s_r : a.s_k, bld = lb == 1 ? b.s_r : b.s_k;
        if (la == 1 && lb == 1)
            hipLaunchKernelGGL((gemm_f32_vec_kernel<EPI, T, true, true>), grid, dim3(256), 0, s, a.p, ald, a.R, b.p, bld, b.R, epi,
                               M, N, K, chunk, slab_stride);
        else if (la == 1)
            hipLaunchKernelGGL((gemm_f32_vec_kernel<EPI, T, true, false>), grid, dim3(256), 0, s, a.p, ald, a.R, b.p, bld, b.R, epi,
                               M, N, K, chunk, slab_stride);
        else
            hipLaunchKernelGGL((gemm_f32_vec_kernel<EPI, T, false, false>), grid, dim3(256), 0, s, a.p, ald, a.R, b.p, bld, b.R,
                               epi, M, N, K, chunk, slab_stride);
        return;
    }
    // the generic kernel needs the whole register file at 128 x 128: it stops at 64
    constexpr int TG = T > 64 ? 64 : T;
    hipLaunchKernelGGL((gemm_f32_kernel<EPI, TG>), dim3(ceil_div(N, TG), ceil_div(M, TG), splits), dim3(256), 0, s, a, b, epi, M,
                       N, K, chunk, slab_stride, f32_vec_ok(a), f32_vec_ok(b));
}

template <int EPI>
inline void launch_f32_pick(const F32Operand& a, const F32Operand& b, const Epilogue& epi, int M, int N, int K, int chunk,
                            long long slab_stride, int splits, hipStream_t s) {
    if ((long long)ceil_div(N, 128) * ceil_div(M, 128) * splits >= F32_BIG_FROM)
        launch_f32_tile<EPI, 128>(a, b, epi, M, N, K, chunk, slab_stride, splits, s);
    else if ((long long)ceil_div(N, 64) * ceil_div(M, 64) * splits < F32_SMALL_BELOW)
        launch_f32_tile<EPI, 32>(a, b, epi, M, N, K, chunk, slab_stride, splits, s);
    else
        launch_f32_tile<EPI, 64>(a, b, epi, M, N, K, chunk, slab_stride, splits, s);
}

template <int EPI>
inline void launch_f32_epi(const F32Operand& a, const F32Operand& b, const Epilogue& epi, int M, int N, int K, hipStream_t s) {
    launch_f32_pick<EPI>(a, b, epi, M, N, K, K, 0ll, 1, s);
}

// K chunks for an fp32 product whose output has few tiles (weight gradients): enough blocks for two per CU, chunks of
// at least 256.  A function of the shape alone, so that the workspace query and the launch agree.
inline int f32_k_splits(int M, int N, int K) {
    const long long tiles = (long long)ceil_div(M, 64) * ceil_div(N, 64);
    if (tiles >= 256 || K < 1024) return 1;
    long long want = (512 + tiles - 1) / tiles;
    const long long most = K / 256;
    if (want > most) want = most;
    return want < 1 ? 1 : (int)want;
}

// the plain fp32 product C = A B^T, split along K through `slabs` when that pays (slabs may be null: one pass)
inline int launch_f32_plain(const F32Operand& a, const F32Operand& b, Epilogue epi, int M, int N, int K, float* slabs,
                            hipStream_t s, const char* what) {
    int splits = slabs ? f32_k_splits(M, N, K) : 1;
    int chunk = K;
    if (splits > 1) {
        chunk = ((K + splits - 1) / splits + 31) / 32 * 32;
        splits = (K + chunk - 1) / chunk;
    }
    if (splits <= 1) {
        launch_f32_pick<VITMI_EPI_NONE>(a, b, epi, M, N, K, K, 0ll, 1, s);
        VITMI_LAUNCH_CHECK(what);
        return VITMI_OK;
    }
    float* out = reinterpret_cast<float*>(epi.C);
    const long long ldc = epi.ldc;
    epi.C = slabs;
    epi.ldc = N;
    launch_f32_pick<VITMI_EPI_NONE>(a, b, epi, M, N, K, chunk, (long long)M * N, splits, s);
    VITMI_LAUNCH_CHECK(what);
    const int blocks = (int)std::min<long long>(((long long)M * N + 255) / 256, 1024);
    hipLaunchKernelGGL(f32_slab_sum_kernel, dim3(blocks), dim3(256), 0, s, slabs, splits, M, N, out, ldc);
    VITMI_LAUNCH_CHECK(what);
    return VITMI_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int EPI, typename OutT, class LA, class LB>
void launch_bf16(const LA& la, const LB& lb, const Epilogue& epi, int M, int N, int K, hipStream_t s) {
    dim3 grid(ceil_div(N, BN), ceil_div(M, BM));
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, OutT, LA, LB>), grid, dim3(NTHREADS), 0, s, la, lb, epi, M, N, K);
}

template <int EPI, typename OutT>
void dispatch_bf16_layout(bool a_kc, bool b_kc, const void* A, long long sam, long long sak, const void* B,
                          long long sbn, long long sbk, const Epilogue& epi, int M, int N, int K, hipStream_t s) {
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* b = (const bf16_t*)B;
    // 16-byte loads need the contiguous direction to be the chunk direction and every chunk start aligned
    const int va = al16(A) && (a_kc ? (sak == 1 && sam % 8 == 0) : (sam == 1 && sak % 8 == 0));
    const int vb = al16(B) && (b_kc ? (sbk == 1 && sbn % 8 == 0) : (sbn == 1 && sbk % 8 == 0));
    if (a_kc && b_kc)
        launch_bf16<EPI, OutT>(StridedLoader<true>{a, sam, sak, M, K, va}, StridedLoader<true>{b, sbn, sbk, N, K, vb},
                               epi, M, N, K, s);
    else if (a_kc && !b_kc)
        launch_bf16<EPI, OutT>(StridedLoader<true>{a, sam, sak, M, K, va}, StridedLoader<false>{b, sbn, sbk, N, K, vb},
                               epi, M, N, K, s);
    else if (!a_kc && b_kc)
        launch_bf16<EPI, OutT>(StridedLoader<false>{a, sam, sak, M, K, va}, StridedLoader<true>{b, sbn, sbk, N, K, vb},
                               epi, M, N, K, s);
    else
        launch_bf16<EPI, OutT>(StridedLoader<false>{a, sam, sak, M, K, va},
                               StridedLoader<false>{b, sbn, sbk, N, K, vb}, epi, M, N, K, s);
}

}  // namespace

// cs_partial != nullptr asks the phased nt kernel (the caller checked that it will run) for column sums
static int gemm_impl(int dt, int M, int N, int K, const void* A, long long sam, long long sak, const void* B,
                     long long sbn, long long sbk, void* C, long long ldc, int c_dt, int epilogue, const float* bias,
                     const void* residual, long long ldr, void* aux, long long ldaux, float* cs_partial, void* stream) {
    VITMI_CHECK_ARG(A && B && C, "vitmi_gemm: null operand");
    VITMI_CHECK_ARG(M > 0 && N > 0 && K > 0, "vitmi_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    VITMI_CHECK_ARG(ldc >= N, "vitmi_gemm: ldc=%lld < N=%d", ldc, N);
    VITMI_CHECK_ARG(epilogue >= VITMI_EPI_NONE && epilogue <= VITMI_EPI_MULAUX, "vitmi_gemm: bad epilogue %d", epilogue);
    VITMI_CHECK_ARG(!epi_is_auxmul(epilogue) || aux, "vitmi_gemm: this epilogue needs aux");
    VITMI_CHECK_ARG(!residual || ldr >= N, "vitmi_gemm: ldr=%lld < N=%d", ldr, N);
    VITMI_CHECK_ARG(!aux || ldaux >= N, "vitmi_gemm: ldaux=%lld < N=%d", ldaux, N);
    VITMI_CHECK_ARG(sam >= 0 && sak >= 0 && sbn >= 0 && sbk >= 0, "vitmi_gemm: strides must not be negative");
    hipStream_t s = as_stream(stream);
    Epilogue epi{C, ldc, bias, residual, ldr, aux, ldaux, 0, 0};
    int fam = VITMI_FAM_GEMM;
    bool fast_nt = false;
    if (dt == VITMI_BF16) {
        VITMI_CHECK_ARG(c_dt == VITMI_BF16 || (c_dt == VITMI_F32 && epilogue == VITMI_EPI_NONE),
                        "vitmi_gemm: fp32 output supports only the plain epilogue");
        const int layout = ((sam == 1 && sak != 1) ? 2 : 0) + ((sbn == 1 && sbk != 1) ? 1 : 0);
        const int eo = c_dt == VITMI_F32 ? 1 : (epi_is_gelu(epilogue) ? 2 : (epi_is_auxmul(epilogue) ? 3 : 0));
        fam = VITMI_FAM_GEMM_VARIANT0 + 4 * layout + eo;
        fast_nt = sak == 1 && sbk == 1 &&
                  vitmi_gemm_nt_fast_ok(M, N, K, A, sam, B, sbn, C, ldc, c_dt, residual, ldr, aux, ldaux);
        if (fast_nt) fam = VITMI_FAM_GEMM_NT_FAST0 + eo;
    }
    ProfScope prof(fam, s, 2.0 * M * N * K);
    if (fast_nt)
        return vitmi_gemm_nt_fast(M, N, K, A, sam, B, sbn, C, ldc, c_dt, epilogue, bias, residual, ldr, aux, ldaux,
                                  cs_partial, s);
    if (dt == VITMI_F32) {
        VITMI_CHECK_ARG(c_dt == VITMI_F32, "vitmi_gemm: fp32 compute writes fp32 output");
        F32Operand a{(const float*)A, sam, sak, M, 0, 0, 0, 0, 0, 0, 0, 0};
        F32Operand b{(const float*)B, sbn, sbk, N, 0, 0, 0, 0, 0, 0, 0, 0};
        dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
        if (epilogue == VITMI_EPI_NONE)
            return launch_f32_plain(a, b, epi, M, N, K, nullptr, s, "vitmi_gemm(f32)");
        else if (epilogue == VITMI_EPI_GELU)
            launch_f32_epi<VITMI_EPI_GELU>(a, b, epi, M, N, K, s);
        else if (epilogue == VITMI_EPI_GELU_D)
            launch_f32_epi<VITMI_EPI_GELU_D>(a, b, epi, M, N, K, s);
        else if (epilogue == VITMI_EPI_MULAUX)
            launch_f32_epi<VITMI_EPI_MULAUX>(a, b, epi, M, N, K, s);
        else
            launch_f32_epi<VITMI_EPI_DGELU>(a, b, epi, M, N, K, s);
        VITMI_LAUNCH_CHECK("vitmi_gemm(f32)");
        return VITMI_OK;
    }
    VITMI_CHECK_ARG(dt == VITMI_BF16, "vitmi_gemm: bad dtype %d", dt);
    // an operand is K-contiguous unless its row stride is 1 and its k stride is not
    const bool a_kc = !(sam == 1 && sak != 1);
    const bool b_kc = !(sbn == 1 && sbk != 1);
    if (c_dt == VITMI_F32)
        dispatch_bf16_layout<VITMI_EPI_NONE, float>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    else if (epilogue == VITMI_EPI_NONE)
        dispatch_bf16_layout<VITMI_EPI_NONE, bf16_t>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    else if (epilogue == VITMI_EPI_GELU)
        dispatch_bf16_layout<VITMI_EPI_GELU, bf16_t>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    else if (epilogue == VITMI_EPI_GELU_D)
        dispatch_bf16_layout<VITMI_EPI_GELU_D, bf16_t>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    else if (epilogue == VITMI_EPI_MULAUX)
        dispatch_bf16_layout<VITMI_EPI_MULAUX, bf16_t>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    else
        dispatch_bf16_layout<VITMI_EPI_DGELU, bf16_t>(a_kc, b_kc, A, sam, sak, B, sbn, sbk, epi, M, N, K, s);
    VITMI_LAUNCH_CHECK("vitmi_gemm(bf16)");
    return VITMI_OK;
}

extern "C" int vitmi_gemm(int dt, int M, int N, int K, const void* A, long long sam, long long sak, const void* B,
                          long long sbn, long long sbk, void* C, long long ldc, int c_dt, int epilogue,
                          const float* bias, const void* residual, long long ldr, void* aux, long long ldaux,
                          void* stream) {
    return gemm_impl(dt, M, N, K, A, sam, sak, B, sbn, sbk, C, ldc, c_dt, epilogue, bias, residual, ldr, aux, ldaux,
                     nullptr, stream);
}

extern "C" int vitmi_gemm_ln(int M, int N, int K, const void* x, long long ldx, const void* w_folded,
                             const float* bias_folded, const float* w_rowsum, const float* mean, const float* rstd,
                             void* C, long long ldc, int epilogue, void* stream) {
    VITMI_CHECK_ARG(x && w_folded && bias_folded && w_rowsum && mean && rstd && C, "vitmi_gemm_ln: null pointer");
    VITMI_CHECK_ARG(M > 0 && N > 0 && K > 0 && ldx >= K && ldc >= N, "vitmi_gemm_ln: bad shape M=%d N=%d K=%d", M, N, K);
    VITMI_CHECK_ARG(epilogue == VITMI_EPI_LN || epilogue == VITMI_EPI_LN_GELU, "vitmi_gemm_ln: bad epilogue %d", epilogue);
    if (!vitmi_gemm_nt_fast_ok(M, N, K, x, ldx, w_folded, K, C, ldc, VITMI_BF16, nullptr, 0, nullptr, 0) ||
        ((size_t)bias_folded | (size_t)w_rowsum) % 16 || N % 8) {
        vitmi_set_error("vitmi_gemm_ln: operands must be 16-byte aligned with K, N, ldx, ldc multiples of 8");
        return VITMI_ERR_UNSUPPORTED;
    }
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_GEMM_NT_FAST0 + (epilogue == VITMI_EPI_LN ? 0 : 1), s, 2.0 * M * N * K);
    return vitmi_gemm_nt_fast_ln(M, N, K, x, ldx, w_folded, K, C, ldc, epilogue, bias_folded, mean, rstd, w_rowsum, s);
}

extern "C" long long vitmi_gemm_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const long long fused = (long long)vitmi_gemm_nt_colsum_rows(M) * N * (long long)sizeof(float);
    const long long separate = vitmi_colsum_workspace_bytes_impl(M, N);
    return fused > separate ? fused : separate;
}

extern "C" int vitmi_gemm_colsum(int dt, int M, int N, int K, const void* A, long long sam, long long sak,
                                 const void* B, long long sbn, long long sbk, void* C, long long ldc, int c_dt,
                                 int epilogue, const float* bias, const void* residual, long long ldr, void* aux,
                                 long long ldaux, float* colsum, void* workspace, void* stream) {
    VITMI_CHECK_ARG(colsum && workspace, "vitmi_gemm_colsum: null colsum or workspace");
    VITMI_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "vitmi_gemm_colsum: bad operands");
    hipStream_t s = as_stream(stream);
    const bool fused = dt == VITMI_BF16 && c_dt == VITMI_BF16 && epi_is_auxmul(epilogue) && sak == 1 && sbk == 1 &&
                       vitmi_gemm_nt_fast_ok(M, N, K, A, sam, B, sbn, C, ldc, c_dt, residual, ldr, aux, ldaux) &&
                       vitmi_gemm_nt_phased_ok(M, N, K);
    int rc = gemm_impl(dt, M, N, K, A, sam, sak, B, sbn, sbk, C, ldc, c_dt, epilogue, bias, residual, ldr, aux, ldaux,
                       fused ? (float*)workspace : nullptr, stream);
    if (rc != VITMI_OK) return rc;
    if (fused) {
        ProfScope prof(VITMI_FAM_OTHER, s, 0.0);
        return vitmi_colsum_final_launch((const float*)workspace, vitmi_gemm_nt_colsum_rows(M), N, colsum, s);
    }
    return vitmi_colsum(C, c_dt, M, N, ldc, colsum, workspace, stream);
}

extern "C" int vitmi_patch_embed_fwd(const void* x, int x_dt, const void* w, const float* bias, void* out, int dt,
                                     int B, int C, int H, int W, int p, int D, long long out_batch_stride,
                                     void* stream) {
    VITMI_CHECK_ARG(x && w && out, "vitmi_patch_embed_fwd: null pointer");
    VITMI_CHECK_ARG(B > 0 && C > 0 && p > 0 && D > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0,
                    "vitmi_patch_embed_fwd: bad geometry B=%d C=%d H=%d W=%d p=%d D=%d", B, C, H, W, p, D);
    const int gh = H / p, gw = W / p, per = gh * gw;
    const int M = B * per, K = C * p * p, N = D;
    VITMI_CHECK_ARG(out_batch_stride >= (long long)per * D, "vitmi_patch_embed_fwd: out_batch_stride too small");
    hipStream_t s = as_stream(stream);
    Epilogue epi{out, D, bias, nullptr, 0, nullptr, 0, per, out_batch_stride};
    if (dt == VITMI_BF16 && x_dt == VITMI_BF16 && vitmi_patch_fast_ok(x, w, out, W, p, D, K, out_batch_stride)) {
        ProfScope prof(VITMI_FAM_GEMM_NT_FAST0, s, 2.0 * M * N * K);
        return vitmi_patch_fwd_fast(x, w, bias, out, B, C, H, W, p, p, D, out_batch_stride, s);
    }
    ProfScope prof(VITMI_FAM_GEMM, s, 2.0 * M * N * K);
    if (dt == VITMI_F32) {
        VITMI_CHECK_ARG(x_dt == VITMI_F32, "vitmi_patch_embed_fwd: fp32 compute needs an fp32 image");
        F32Operand a{(const float*)x, 0, 0, M, 1, 1, C, H, W, p, gw, gh};
        F32Operand b{(const float*)w, K, 1, N, 0, 0, 0, 0, 0, 0, 0, 0};
        dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
        hipLaunchKernelGGL(gemm_f32_kernel<VITMI_EPI_NONE>, grid, dim3(256), 0, s, a, b, epi, M, N, K, K, 0ll, f32_vec_ok(a), f32_vec_ok(b));
        VITMI_LAUNCH_CHECK("vitmi_patch_embed_fwd(f32)");
        return VITMI_OK;
    }
    VITMI_CHECK_ARG(dt == VITMI_BF16, "vitmi_patch_embed_fwd: bad dtype %d", dt);
    StridedLoader<true> lb{(const bf16_t*)w, K, 1, N, K, al16(w) && K % 8 == 0};
    if (x_dt == VITMI_F32) {
        PatchLoader<float, true> la{(const float*)x, C, H, W, p, gw, gh, M, K};
        launch_bf16<VITMI_EPI_NONE, bf16_t>(la, lb, epi, M, N, K, s);
    } else if (x_dt == VITMI_BF16) {
        PatchLoader<bf16_t, true> la{(const bf16_t*)x, C, H, W, p, gw, gh, M, K};
        launch_bf16<VITMI_EPI_NONE, bf16_t>(la, lb, epi, M, N, K, s);
    } else {
        vitmi_set_error("vitmi_patch_embed_fwd: bad image dtype %d", x_dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_patch_embed_fwd(bf16)");
    return VITMI_OK;
}

extern "C" long long vitmi_patch_embed_bwd_w_workspace_bytes(int dt, int B, int C, int H, int W, int p, int D) {
    if (p <= 0) return 0;
    const int E = C * p * p, T = B * (H / p) * (W / p);
    if (dt == VITMI_F32) {
        const int z = f32_k_splits(D, E, T);
        return z > 1 ? (long long)(z + 1) * D * E * (long long)sizeof(float) : 0;
    }
    if (dt != VITMI_BF16) return 0;
    const int ns = vitmi_gemm_tn_slabs_bound(D, E, T);    // independent of the CU reservation in force
    return ns > 1 ? (long long)ns * D * E * (long long)sizeof(float) : 0;
}

extern "C" int vitmi_patch_embed_bwd_w(const void* x, int x_dt, const void* dy, long long dy_batch_stride, float* dw,
                                       void* workspace, int dt, int B, int C, int H, int W, int p, int D,
                                       void* stream) {
    VITMI_CHECK_ARG(x && dy && dw, "vitmi_patch_embed_bwd_w: null pointer");
    VITMI_CHECK_ARG(B > 0 && C > 0 && p > 0 && D > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0,
                    "vitmi_patch_embed_bwd_w: bad geometry");
    const int gh = H / p, gw = W / p, per = gh * gw;
    const int T = B * per, E = C * p * p;
    VITMI_CHECK_ARG(dy_batch_stride == (long long)per * D,
                    "vitmi_patch_embed_bwd_w: dy must be dense [B, tokens, D] (batch stride %lld != %lld)",
                    dy_batch_stride, (long long)per * D);
    hipStream_t s = as_stream(stream);
    // dW[d, e] = sum_t dy[t, d] * P[t, e]:  rows of A = d (dy is contiguous along d), rows of B = e, k = t
    Epilogue epi{dw, E, nullptr, nullptr, 0, nullptr, 0, 0, 0};
    if (dt == VITMI_BF16 && x_dt == VITMI_BF16 && vitmi_patch_fast_ok(x, dy, dw, W, p, D, E, 8)) {
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * D * E * (double)T);
        return vitmi_patch_bwd_w_fast(x, dy, dw, (float*)workspace, B, C, H, W, p, p, D, s);
    }
    ProfScope prof(VITMI_FAM_GEMM, s, 2.0 * D * E * (double)T);
    if (dt == VITMI_F32) {
        VITMI_CHECK_ARG(x_dt == VITMI_F32, "vitmi_patch_embed_bwd_w: fp32 compute needs an fp32 image");
        F32Operand a{(const float*)dy, 1, D, D, 0, 0, 0, 0, 0, 0, 0, 0};
        F32Operand b{(const float*)x, 0, 0, E, 1, 0, C, H, W, p, gw, gh};
        return launch_f32_plain(a, b, epi, D, E, T, (float*)workspace, s, "vitmi_patch_embed_bwd_w(f32)");
    }
    VITMI_CHECK_ARG(dt == VITMI_BF16, "vitmi_patch_embed_bwd_w: bad dtype %d", dt);
    StridedLoader<false> la{(const bf16_t*)dy, 1, D, D, T, al16(dy) && D % 8 == 0};
    if (x_dt == VITMI_F32) {
        PatchLoader<float, false> lb{(const float*)x, C, H, W, p, gw, gh, T, E};
        launch_bf16<VITMI_EPI_NONE, float>(la, lb, epi, D, E, T, s);
    } else if (x_dt == VITMI_BF16) {
        PatchLoader<bf16_t, false> lb{(const bf16_t*)x, C, H, W, p, gw, gh, T, E};
        launch_bf16<VITMI_EPI_NONE, float>(la, lb, epi, D, E, T, s);
    } else {
        vitmi_set_error("vitmi_patch_embed_bwd_w: bad image dtype %d", x_dt);
        return VITMI_ERR_BAD_ARG;
    }
    VITMI_LAUNCH_CHECK("vitmi_patch_embed_bwd_w(bf16)");
    return VITMI_OK;
}

// ---- square patches narrower than 8 pixels, in place (the 4 x 4 patches of the 32 x 32 geometry, UMAP.py:12-17) ---------
// A 16-byte piece of a bf16 image row is 8 pixels = the row pieces of g = 8 / p horizontally adjacent patches.  The
// product is therefore taken over "wide" patches of p x 8 pixels: row = the group of g consecutive tokens, K = C p 8,
// N = g D outputs (token s of the group, feature d at column s D + d -- the group's g rows of the [tokens, D] result,
// which are contiguous), against the block weight Wg[s D + d][c, i, jj] = W[d][c, i, jj - s p] for s p <= jj < (s+1) p,
// else 0.  g times the arithmetic of a product that is a per-mille of the step, and no gathered copy of the image.
namespace {
__global__ void patch_wide_weight_kernel(const float* __restrict__ w, const float* __restrict__ bias, bf16_t* __restrict__ wg,
                                         float* __restrict__ bias_g, int D, int C, int p) {
    const int g = 8 / p, K = C * p * 8, total = g * D * K;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / K, k = i - n * K;
        const int s = n / D, d = n - s * D;
        const int c = k / (p * 8), q = k - c * p * 8;
        const int r = q >> 3, jj = q & 7;
        const int j = jj - s * p;
        wg[i] = (j >= 0 && j < p) ? (bf16_t)w[((long long)d * C + c) * p * p + r * p + j] : (bf16_t)0.f;
    }
    if (bias && bias_g)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g * D; i += gridDim.x * blockDim.x) bias_g[i] = bias[i % D];
}
// dw[d][c, r, j] = sum_s dwg[s D + d][c, r, s p + j]
__global__ void patch_wide_collapse_kernel(const float* __restrict__ dwg, float* __restrict__ dw, int D, int C, int p) {
    const int g = 8 / p, E = C * p * p, K = C * p * 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D * E; i += gridDim.x * blockDim.x) {
        const int d = i / E, e = i - d * E;
        const int c = e / (p * p), q = e - c * p * p;
        const int r = q / p, j = q - r * p;
        float v = 0.f;
        for (int s = 0; s < g; ++s) v += dwg[(long long)(s * D + d) * K + c * p * 8 + r * 8 + s * p + j];
        dw[i] = v;
    }
}
inline bool wide_geometry_ok(int B, int C, int H, int W, int p, int D) {
    return B > 0 && C > 0 && D > 0 && p > 0 && p < 8 && 8 % p == 0 && H % p == 0 && W % 8 == 0 && D % 8 == 0;
}
}  // namespace

extern "C" int vitmi_patch_wide_group(int dt, int x_dt, int B, int C, int H, int W, int p, int D) {
    return (dt == VITMI_BF16 && x_dt == VITMI_BF16 && wide_geometry_ok(B, C, H, W, p, D)) ? 8 / p : 0;
}

extern "C" int vitmi_patch_wide_weight(const float* w, const float* bias, void* w_wide, float* bias_wide, int D, int C,
                                       int p, void* stream) {
    VITMI_CHECK_ARG(w && w_wide && D > 0 && C > 0 && p > 0 && p < 8 && 8 % p == 0, "vitmi_patch_wide_weight: bad argument");
    VITMI_CHECK_ARG(!bias == !bias_wide, "vitmi_patch_wide_weight: bias and bias_wide go together");
    const int total = (8 / p) * D * C * p * 8;
    hipLaunchKernelGGL(patch_wide_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, bias,
                       (bf16_t*)w_wide, bias_wide, D, C, p);
    VITMI_LAUNCH_CHECK("vitmi_patch_wide_weight");
    return VITMI_OK;
}

extern "C" int vitmi_patch_embed_fwd_wide(const void* x, const void* w_wide, const float* bias_wide, void* out, int B,
                                          int C, int H, int W, int p, int D, long long out_batch_stride, void* stream) {
    VITMI_CHECK_ARG(x && w_wide && out, "vitmi_patch_embed_fwd_wide: null pointer");
    VITMI_CHECK_ARG(wide_geometry_ok(B, C, H, W, p, D), "vitmi_patch_embed_fwd_wide: bad geometry B=%d C=%d H=%d W=%d p=%d D=%d",
                    B, C, H, W, p, D);
    const int g = 8 / p, per = (H / p) * (W / p);
    VITMI_CHECK_ARG(out_batch_stride >= (long long)per * D && out_batch_stride % 8 == 0 && al16(x) && al16(w_wide) && al16(out),
                    "vitmi_patch_embed_fwd_wide: operands must be 16-byte aligned, batch stride a multiple of 8");
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_GEMM_NT_FAST0, s, 2.0 * B * per * D * C * p * 8);
    return vitmi_patch_fwd_fast(x, w_wide, bias_wide, out, B, C, H, W, p, 8, g * D, out_batch_stride, s);
}

extern "C" long long vitmi_patch_embed_bwd_w_wide_workspace_bytes(int B, int C, int H, int W, int p, int D) {
    if (!wide_geometry_ok(B, C, H, W, p, D)) return 0;
    const int g = 8 / p, K = C * p * 8, T = B * (H / p) * (W / 8);
    const long long dwg = ((long long)g * D * K * (long long)sizeof(float) + 255) & ~255ll;
    const int ns = vitmi_gemm_tn_slabs_bound(g * D, K, T);
    return dwg + (ns > 1 ? (long long)ns * g * D * K * (long long)sizeof(float) : 0);
}

extern "C" int vitmi_patch_embed_bwd_w_wide(const void* x, const void* dy, float* dw, void* workspace, int B, int C, int H,
                                            int W, int p, int D, void* stream) {
    VITMI_CHECK_ARG(x && dy && dw && workspace, "vitmi_patch_embed_bwd_w_wide: null pointer");
    VITMI_CHECK_ARG(wide_geometry_ok(B, C, H, W, p, D) && al16(x) && al16(dy), "vitmi_patch_embed_bwd_w_wide: bad geometry");
    const int g = 8 / p, K = C * p * 8;
    const long long dwg_bytes = ((long long)g * D * K * (long long)sizeof(float) + 255) & ~255ll;
    float* dwg = (float*)workspace;
    float* slabs = (float*)((char*)workspace + dwg_bytes);
    hipStream_t s = as_stream(stream);
    {
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * g * D * K * (double)B * (H / p) * (W / 8));
        // dy [tokens, D] read as [groups, g D]: the g tokens of a group are consecutive rows
        const int rc = vitmi_patch_bwd_w_fast(x, dy, dwg, slabs, B, C, H, W, p, 8, g * D, s);
        if (rc != VITMI_OK) return rc;
    }
    hipLaunchKernelGGL(patch_wide_collapse_kernel, dim3(ceil_div(D * C * p * p, 256)), dim3(256), 0, s, dwg, dw, D, C, p);
    VITMI_LAUNCH_CHECK("vitmi_patch_embed_bwd_w_wide");
    return VITMI_OK;
}

extern "C" long long vitmi_gemm_wgrad_workspace_bytes(int dt, int P, int Q, int T) {
    if (dt == VITMI_F32) {
        const int z = f32_k_splits(P, Q, T);  // (+1: the chunk rounding may add a slab)
        return z > 1 ? (long long)(z + 1) * P * Q * (long long)sizeof(float) : 0;
    }
    if (dt != VITMI_BF16) return 0;
    // sized for every CU reservation and tile configuration (vitmi_set_reserved_cus may change between the query
    // and the call; callers cache the answer per shape)
    const int ns = vitmi_gemm_tn_slabs_bound(P, Q, T);
    return ns > 1 ? (long long)ns * P * Q * (long long)sizeof(float) : 0;
}

extern "C" int vitmi_gemm_wgrad(int dt, int P, int Q, int T, const void* U, long long ldu, const void* V,
                                long long ldv, float* C, long long ldc, void* workspace, void* stream) {
    VITMI_CHECK_ARG(U && V && C, "vitmi_gemm_wgrad: null operand");
    VITMI_CHECK_ARG(P > 0 && Q > 0 && T > 0 && ldu >= 0 && ldv >= 0 && ldc >= Q, "vitmi_gemm_wgrad: bad shape");
    hipStream_t s = as_stream(stream);
    if (dt == VITMI_BF16 && ldc == Q && vitmi_gemm_tn_fast_ok(P, Q, U, ldu, V, ldv, C, ldc)) {
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * P * Q * (double)T);
        return vitmi_gemm_tn_fast(P, Q, T, U, ldu, V, ldv, C, ldc, (float*)workspace, s);
    }
    if (dt == VITMI_F32) {
        ProfScope prof(VITMI_FAM_GEMM, s, 2.0 * P * Q * (double)T);
        F32Operand a{(const float*)U, 1, ldu, P, 0, 0, 0, 0, 0, 0, 0, 0};
        F32Operand b{(const float*)V, 1, ldv, Q, 0, 0, 0, 0, 0, 0, 0, 0};
        Epilogue epi{C, ldc, nullptr, nullptr, 0, nullptr, 0, 0, 0};
        return launch_f32_plain(a, b, epi, P, Q, T, (float*)workspace, s, "vitmi_gemm_wgrad(f32)");
    }
    // general path: the strided GEMM with both operands read along their output index
    return vitmi_gemm(dt, P, Q, T, U, 1, ldu, V, 1, ldv, C, ldc, VITMI_F32, VITMI_EPI_NONE, nullptr, nullptr, 0,
                      nullptr, 0, stream);
}

extern "C" long long vitmi_gemm_wgrad_pair_workspace_bytes(int dt, int T, int P0, int Q0, int P1, int Q1) {
    const int P[2] = {P0, P1}, Q[2] = {Q0, Q1};
    // enough for either route: operands that turn out to be misaligned at call time take the two single products
    const long long w0 = vitmi_gemm_wgrad_workspace_bytes(dt, P0, Q0, T), w1 = vitmi_gemm_wgrad_workspace_bytes(dt, P1, Q1, T);
    long long b = w0 > w1 ? w0 : w1;
    if (dt == VITMI_BF16) {
        const long long pair = vitmi_gemm_tn_batch_bytes_bound(2, P, Q, T);
        if (pair > b) b = pair;
    }
    return b;
}

extern "C" int vitmi_gemm_wgrad_pair(int dt, int T, int P0, int Q0, const void* U0, long long ldu0, const void* V0,
                                     long long ldv0, float* C0, int P1, int Q1, const void* U1, long long ldu1,
                                     const void* V1, long long ldv1, float* C1, void* workspace, void* stream) {
    VITMI_CHECK_ARG(U0 && V0 && C0 && U1 && V1 && C1, "vitmi_gemm_wgrad_pair: null operand");
    VITMI_CHECK_ARG(T > 0 && P0 > 0 && Q0 > 0 && P1 > 0 && Q1 > 0 && ldu0 >= 0 && ldv0 >= 0 && ldu1 >= 0 && ldv1 >= 0,
                    "vitmi_gemm_wgrad_pair: bad shape");
    hipStream_t s = as_stream(stream);
    const int P[2] = {P0, P1}, Q[2] = {Q0, Q1};
    if (dt == VITMI_BF16 && workspace && vitmi_gemm_tn_batch_bytes(2, P, Q, T) > 0) {
        const void* U[2] = {U0, U1};
        const void* V[2] = {V0, V1};
        const long long ldu[2] = {ldu0, ldu1}, ldv[2] = {ldv0, ldv1};
        float* C[2] = {C0, C1};
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * ((double)P0 * Q0 + (double)P1 * Q1) * (double)T);
        const int rc = vitmi_gemm_tn_batch(2, P, Q, T, U, ldu, V, ldv, C, (float*)workspace, s, nullptr);
        if (rc != VITMI_ERR_UNSUPPORTED) return rc;
    }
    const int rc = vitmi_gemm_wgrad(dt, P0, Q0, T, U0, ldu0, V0, ldv0, C0, Q0, workspace, stream);
    if (rc != VITMI_OK) return rc;
    return vitmi_gemm_wgrad(dt, P1, Q1, T, U1, ldu1, V1, ldv1, C1, Q1, workspace, stream);
}

// ---- the products around the attention core when qkv / dqkv are head-major planes [3*H][tokens][64] (bf16, heads of 64;
// chosen by the whole-block entry points of blocks.hip, which also own the decision that every operand qualifies)
bool vitmi_qkv_planes_ok(int dt, int B, int N, int D, int H, int DH, const void* x, const void* w_qkv, const void* qkv) {
    (void)B;
    (void)N;
    (void)H;
    return dt == VITMI_BF16 && DH == 64 && D % 8 == 0 && al16(x) && al16(w_qkv) && al16(qkv);
}

// qkv planes = xn W^T (+ bias)
int vitmi_gemm_to_planes(int T, int N, int K, const void* xn, const void* w, const float* bias, void* planes, void* stream) {
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_GEMM_NT_FAST0, s, 2.0 * T * N * K);
    return vitmi_gemm_nt_fast_planes(T, N, K, xn, K, w, K, planes, N, bias, 1, 0, s);
}

// dxn = dqkv W for dqkv in planes: the k-contiguous form on the transposed weight w_t [K_out = D][3I]
int vitmi_gemm_from_planes(int T, int D, int I3, const void* dqkv_planes, const void* w_t, void* dxn, void* stream) {
    hipStream_t s = as_stream(stream);
    ProfScope prof(VITMI_FAM_GEMM_NT_FAST0, s, 2.0 * T * D * I3);
    return vitmi_gemm_nt_fast_planes(T, D, I3, dqkv_planes, 64, w_t, I3, dxn, D, nullptr, 0, 1, s);
}

// the two weight gradients of the attention sub-block with the second product's dy (dqkv) in planes
int vitmi_gemm_wgrad_pair_planes(int T, int P0, int Q0, const void* U0, long long ldu0, const void* V0, long long ldv0,
                                 float* C0, int P1, int Q1, const void* U1_planes, const void* V1, long long ldv1, float* C1,
                                 void* workspace, void* stream) {
    hipStream_t s = as_stream(stream);
    const int P[2] = {P0, P1}, Q[2] = {Q0, Q1};
    if (workspace && vitmi_gemm_tn_batch_bytes(2, P, Q, T) > 0) {
        const void* U[2] = {U0, U1_planes};
        const void* V[2] = {V0, V1};
        const long long ldu[2] = {ldu0, 64}, ldv[2] = {ldv0, ldv1};
        const int planes[2] = {0, 1};
        float* C[2] = {C0, C1};
        ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * ((double)P0 * Q0 + (double)P1 * Q1) * (double)T);
        const int rc = vitmi_gemm_tn_batch(2, P, Q, T, U, ldu, V, ldv, C, (float*)workspace, s, planes);
        if (rc != VITMI_ERR_UNSUPPORTED) return rc;
    }
    int rc = vitmi_gemm_wgrad(VITMI_BF16, P0, Q0, T, U0, ldu0, V0, ldv0, C0, Q0, workspace, stream);
    if (rc != VITMI_OK) return rc;
    if (!vitmi_gemm_tn_fast_ok(P1, Q1, U1_planes, 64, V1, ldv1, C1, Q1)) {
        vitmi_set_error("weight gradient from head-major planes: operands must be 16-byte aligned with extents multiples of 8");
        return VITMI_ERR_UNSUPPORTED;
    }
    ProfScope prof(VITMI_FAM_GEMM_TN_FAST, s, 2.0 * P1 * Q1 * (double)T);
    return vitmi_gemm_tn_fast_planes(P1, Q1, T, U1_planes, V1, ldv1, C1, Q1, (float*)workspace, s);
}
