// common.h — shared device helpers for libcomat_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/comat_hip.h"

typedef unsigned short bf16_t;  // raw bf16 storage

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware conversion (v_cvt_pk_bf16_f32 on gfx950: one instruction instead of the
// six of a software rounding; the compiler pairs neighbouring conversions)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// runtime-dtype scalar access (dtype is wave-uniform)
__device__ __forceinline__ float ld_dt(const void* p, int64_t i, int dt) {
    return dt == COMAT_F32 ? ((const float*)p)[i] : bf16_to_f32(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st_dt(void* p, int64_t i, float v, int dt) {
    if (dt == COMAT_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f32_to_bf16(v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

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
// block-wide sum for blockDim.x == 256 (4 waves); sbuf: >= 4 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* sbuf) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
    __syncthreads();
    return sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sbuf) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sbuf[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sbuf[0], sbuf[1]), fmaxf(sbuf[2], sbuf[3]));
}

// fp8 delayed scaling: fold a block's (wave's) abs-max into a site's running maximum.  |x| >= 0, so the uint order of the float
// bits is the float order; the atomic runs at the memory side (agent scope, coherent across the 8 XCDs' L2s) and is skipped when
// the slot already holds a value at least as large - after the first few arrivals almost every block leaves with one load.
__device__ __forceinline__ void fp8_amax_track(unsigned* slot, float m) {
    const unsigned b = __float_as_uint(m);
    if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_fetch_max(slot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 8 (4) fp32 values * inv -> e4m3fn bytes: the arithmetic of comat_fp8_quantize.  The hardware conversion rounds to nearest even
// but does NOT saturate (beyond +-480 it produces the NaN byte): under a just-in-time scale nothing exceeds 448; under a delayed
// scale a tensor may, so the product is clamped first (v_med3_f32, one instruction; exact for everything in range)
__device__ __forceinline__ float fp8_sat(float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); }
__device__ __forceinline__ uint2 fp8_pack8(const float* v, float inv) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[0] * inv), fp8_sat(v[1] * inv), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[2] * inv), fp8_sat(v[3] * inv), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[4] * inv), fp8_sat(v[5] * inv), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[6] * inv), fp8_sat(v[7] * inv), hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}
__device__ __forceinline__ unsigned fp8_pack4(const float* v, float inv) {
    int lo = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[0] * inv), fp8_sat(v[1] * inv), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(v[2] * inv), fp8_sat(v[3] * inv), lo, true);
    return (unsigned)lo;
}

// ---- host side ------------------------------------------------------------------------------------------
void comat_set_error(const char* fmt, ...);
int comat_check_launch(const char* what);
// tuning options (runtime.hip): environment default read once, comat_set_option() overrides
enum { COMAT_OPT_FLASH_TRIM = 0, COMAT_OPT_FLASH_TR, COMAT_OPT_GEMM2, COMAT_OPT_G2_CFG, COMAT_OPT_G2_SPLITS,
       COMAT_OPT_FORCE_SPLITS, COMAT_OPT_NORM_FUSED, COMAT_OPT_GEMM2_TT, COMAT_OPT_FLASH_KT, COMAT_OPT_FLASH_MERGE,
       COMAT_OPT_G2_ORDER, COMAT_OPT_FLASH_XCD, COMAT_OPT_GEMM3, COMAT_OPT_G3_CFG, COMAT_OPT_FLASH_QS,
       COMAT_N_OPTIONS };
int comat_option(int id);
// which kernel family served the calling thread's last comat_gemm / comat_gemm_segments / comat_conv2d call
// (comat_last_gemm_kernel() in the ABI): 0 general 64x64, 1 pipelined, 2 pipelined k-major, 3 pipelined fp8, 4 grouped k-major,
// 5 lean (gemm3.hip)
void comat_note_gemm_kernel(int id);

#define COMAT_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            comat_set_error(__VA_ARGS__);             \
            return COMAT_EINVAL;                      \
        }                                             \
    } while (0)

static inline bool dtype_ok(int dt) { return dt == COMAT_F32 || dt == COMAT_BF16; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int grid_1d(int64_t n_items, int per_block, int cap = 4096) {
    int64_t g = cdiv64(n_items, per_block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
