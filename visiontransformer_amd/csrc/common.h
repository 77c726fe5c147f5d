// Shared device/host helpers for the gfx950 ViT encoder kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vitmi.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// ---- error plumbing -------------------------------------------------------------------------
void vitmi_set_error(const char* fmt, ...);
int vitmi_hip_fail(hipError_t e, const char* what);

#define VITMI_CHECK_ARG(cond, ...)                  \
    do {                                            \
        if (!(cond)) {                              \
            vitmi_set_error(__VA_ARGS__);           \
            return VITMI_ERR_BAD_ARG;               \
        }                                           \
    } while (0)

#define VITMI_LAUNCH_CHECK(name)                                  \
    do {                                                          \
        hipError_t e__ = hipGetLastError();                       \
        if (e__ != hipSuccess) return vitmi_hip_fail(e__, name);  \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- per-kernel-family launch timing (drives bench.py's roofline object) ---------------------
void vitmi_prof_record_begin(int family, hipStream_t s, double work);
void vitmi_prof_record_end(int family, hipStream_t s);

struct ProfScope {
    int fam;
    hipStream_t s;
    ProfScope(int family, hipStream_t st, double work) : fam(family), s(st) { vitmi_prof_record_begin(fam, s, work); }
    ~ProfScope() { vitmi_prof_record_end(fam, s); }
};

// The derivative that VITMI_EPI_GELU_D hands to VITMI_EPI_MULAUX lies in [-0.13, 1.13]: in the 16-bit modes it is
// kept as IEEE half (11 significant bits, no range problem) inside the bf16-typed buffer rather than as bf16 (8 bits),
// whose rounding showed up in the LayerNorm-weight gradients.  fp32 mode keeps it in fp32.
__device__ __forceinline__ __bf16 deriv_pack(float v, __bf16) { return __builtin_bit_cast(__bf16, (_Float16)v); }
__device__ __forceinline__ float deriv_pack(float v, float) { return v; }
__device__ __forceinline__ float deriv_unpack(__bf16 b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ float deriv_unpack(float v) { return v; }

constexpr bool epi_is_gelu(int epi) { return epi == VITMI_EPI_GELU || epi == VITMI_EPI_GELU_D || epi == VITMI_EPI_LN_GELU; }
constexpr bool epi_is_ln(int epi) { return epi == VITMI_EPI_LN || epi == VITMI_EPI_LN_GELU; }
constexpr bool epi_is_auxmul(int epi) { return epi == VITMI_EPI_DGELU || epi == VITMI_EPI_MULAUX; }

// ---- dtype traits --------------------------------------------------------------------------
template <typename T> struct dt_of;
template <> struct dt_of<float> { static constexpr int value = VITMI_F32; };
template <> struct dt_of<bf16_t> { static constexpr int value = VITMI_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp, one v_exp and a 5-term
// polynomial.  Used by the bf16 epilogues, where the result is rounded to 8 bits of mantissa;
// the fp32 path keeps libm's erff.  Returns erf(x/sqrt2) and exp(-x^2/2) together because GELU
// and its derivative need both.
// Every multiply-add below is written as an explicit fmaf and contraction of the remaining operations is switched
// off, so the value does not depend on how the compiler happens to vectorise the calling epilogue (two
// instantiations of one kernel otherwise disagreed in the last bit of a few elements).
__device__ __forceinline__ void erf_exp_fast(float x, float& erf_v, float& e) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __builtin_amdgcn_exp2f((-0.72134752044448170368f * x) * x);  // exp(-x^2/2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = fmaf(-(p * t), e, 1.0f);
    erf_v = copysignf(r, x);
}
// GELU and its derivative for the bf16 epilogues through ONE exponential: Phi(x) ~ s(x) = 1 / (1 + exp(-x q(x^2))),
// q an even polynomial fitted (minimax over [-8, 8] on x s(x) and on its derivative together) to the erf form;
// |x s - gelu_erf(x)| <= 3.8e-5 and |(x s)' - gelu_erf'(x)| <= 9.3e-5, below half a unit in the last place of the
// bf16 / half values they are rounded to.  17 issue slots for the pair instead of 26 with the erf evaluation above:
// the epilogue of the first MLP GEMM is VALU-bound (16 values x 8 row groups per lane with the matrix pipes idle).
// The polynomial's argument is clamped to [-8, 8], beyond which s is 0 or 1 to fp32 precision.
__device__ __forceinline__ void gelu_sigmoid_parts(float x, float& s, float& e, float& xc, float& x2) {
#pragma clang fp contract(off)
    xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    x2 = xc * xc;
    // -log2(e) * {1.5948456719, 0.0740075325, -0.000694974849}
    float q = fmaf(1.0026367671e-3f, x2, -1.0677030009e-1f);
    q = fmaf(q, x2, -2.3008759357f);
    e = __builtin_amdgcn_exp2f(xc * q);  // exp(-x q(x^2))
    s = __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_fast(float x) {
#pragma clang fp contract(off)
    float s, e, xc, x2;
    gelu_sigmoid_parts(x, s, e, xc, x2);
    return x * s;
}
__device__ __forceinline__ void gelu_both_fast(float x, float& g, float& dg) {
#pragma clang fp contract(off)
    float s, e, xc, x2;
    gelu_sigmoid_parts(x, s, e, xc, x2);
    // (x q(x^2))' = c0 + 3 c1 x^2 + 5 c2 x^4
    float zp = fmaf(-3.4748742452e-3f, x2, 2.2202259742e-1f);
    zp = fmaf(zp, x2, 1.5948456719f);
    const float t = (s * (e * s)) * zp;  // s (1 - s) z'
    g = x * s;
    dg = fmaf(x, t, s);
}
// GELU in its erf form through erf_exp_fast (|error| <= 1.5e-7 x): what VITMI_EPI_GELU evaluates, the forward half of
// the "erf" pair
__device__ __forceinline__ float gelu_erf_fast(float x) {
#pragma clang fp contract(off)
    float er, e;
    erf_exp_fast(x, er, e);
    return x * fmaf(0.5f, er, 0.5f);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
#pragma clang fp contract(off)
    float er, e;
    erf_exp_fast(x, er, e);
    return fmaf(x * 0.39894228040143267794f, e, fmaf(0.5f, er, 0.5f));
}

// exact-erf GELU and its derivative (nn.GELU() default form)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
