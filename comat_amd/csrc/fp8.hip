// fp8.hip — per-tensor fp8 (OCP e4m3fn) quantisation for the fp8-forward configuration (BASELINE.json configs[4]:
// "fp8 MFMA UNet forward with bf16 backward").  The frozen weights are quantised once, the activation entering each
// frozen Linear / conv is quantised per call: one pass for the scale (abs-max / 448), one pass for the bytes.  Both are
// HBM streams (2 + 1 bytes per element).  The products run on gemm2.hip's 32x32x64 fp8 MFMA variant.
//
// Arithmetic (restated bit for bit by oracle/fp8.py):
//   scale = max(amax, 2^-100) / 448          (fp32 division; 448 = largest e4m3fn value)
//   q_i   = e4m3fn_rne(x_i * (1 / scale))    (fp32 reciprocal and product, hardware RNE conversion, saturating)
#include "common.h"

namespace {

constexpr float FP8_MAX = 448.0f;

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *(const uint4*)p;
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

// ws[0]: ticket counter (uint, zero between launches), ws[1]: running maximum as uint bits (zero between launches).
// |x| >= 0, so the uint order of the float bits is the float order and atomicMax is exact and order-independent.
template <typename T> __global__ __launch_bounds__(256) void fp8_scale_kernel(const T* x, int64_t n, float* scale, unsigned* ws,
                                                                              unsigned* amax_bits) {
    __shared__ float sbuf[4];
    float m = 0.0f;
    const int64_t nv = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * 8) m = fmaxf(m, fabsf(ldf<T>(x + nv * 8 + threadIdx.x)));
    m = block_max_256(m, sbuf);
    if (threadIdx.x == 0) {
        // every hand-off here is an agent-scope atomic executed at the memory side (coherent across the 8 XCDs' L2s, like the
        // split-K tickets of gemm_shared.h): no plain stores, hence no fences.  The maximum must have been performed
        // before this block's ticket becomes visible: wait for it (gfx9 counts atomics in vmcnt).
        __hip_atomic_fetch_max(ws + 1, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            const unsigned bits = __hip_atomic_exchange(ws + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *scale = fmaxf(__uint_as_float(bits), 0x1p-100f) / FP8_MAX;
            if (amax_bits) __hip_atomic_fetch_max(amax_bits, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // delayed scaling: tracked too
        }
    }
}

template <typename T> __global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* x, int64_t n, const float* scale,
                                                                                 unsigned char* y) {
    const float inv = 1.0f / *scale;
    const int64_t nv = n / 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
        *(uint2*)(y + i * 8) = fp8_pack8(v, inv);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * 8) {
        const int64_t i = nv * 8 + threadIdx.x;
        const int q = __builtin_amdgcn_cvt_pk_fp8_f32(ldf<T>(x + i) * inv, 0.0f, 0, false);
        y[i] = (unsigned char)(q & 0xff);
    }
}

// Delayed scaling (round 6): the bytes under a scale that is ALREADY in memory (the site's abs-max of the previous optimizer step),
// and this tensor's abs-max folded into the site's running maximum for the next one - ONE launch, no ticket, nobody waits.
template <typename T> __global__ __launch_bounds__(256) void fp8_quantize_scaled_kernel(const T* x, int64_t n, const float* scale,
                                                                                        unsigned char* y, unsigned* amax_bits) {
    __shared__ float sbuf[4];
    const float inv = 1.0f / *scale;
    const int64_t nv = n / 8;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        float v[8];
        load8<T>(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[e]));
        *(uint2*)(y + i * 8) = fp8_pack8(v, inv);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nv * 8) {
        const int64_t i = nv * 8 + threadIdx.x;
        const float xe = ldf<T>(x + i);
        m = fmaxf(m, fabsf(xe));
        const int q = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_sat(xe * inv), 0.0f, 0, false);
        y[i] = (unsigned char)(q & 0xff);
    }
    m = block_max_256(m, sbuf);
    if (threadIdx.x == 0) fp8_amax_track(amax_bits, m);
}

// once per optimizer step: every site that saw a tensor gets scale = max(amax, 2^-100) / 448 and a fresh running maximum
__global__ __launch_bounds__(256) void fp8_scales_update_kernel(unsigned* amax_bits, float* scale, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned b = __hip_atomic_load(amax_bits + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b != 0u) {
        scale[i] = fmaxf(__uint_as_float(b), 0x1p-100f) / FP8_MAX;
        __hip_atomic_store(amax_bits + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

extern "C" int comat_fp8_quantize_scaled(const void* x, int64_t n, int32_t dtype, const float* scale, void* y, uint32_t* amax_bits,
                                         void* stream) {
    COMAT_REQUIRE(x && scale && y && amax_bits && n > 0, "comat_fp8_quantize_scaled: null argument or empty tensor");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_fp8_quantize_scaled: bad dtype");
    COMAT_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, "comat_fp8_quantize_scaled: x / y must be 16 / 8-byte aligned");
    const int grid = grid_1d(n / 8, 256 * 4, 2048);
    if (dtype == COMAT_F32)
        hipLaunchKernelGGL(fp8_quantize_scaled_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, n,
                           scale, (unsigned char*)y, amax_bits);
    else
        hipLaunchKernelGGL(fp8_quantize_scaled_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n,
                           scale, (unsigned char*)y, amax_bits);
    return comat_check_launch("comat_fp8_quantize_scaled");
}

extern "C" int comat_fp8_scales_update(uint32_t* amax_bits, float* scale, int32_t n, void* stream) {
    COMAT_REQUIRE(amax_bits && scale && n > 0, "comat_fp8_scales_update: null argument or no sites");
    hipLaunchKernelGGL(fp8_scales_update_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, amax_bits, scale, n);
    return comat_check_launch("comat_fp8_scales_update");
}

extern "C" int comat_fp8_scale(const void* x, int64_t n, int32_t dtype, float* scale, void* ws, uint32_t* amax_bits, void* stream) {
    COMAT_REQUIRE(x && scale && ws && n > 0, "comat_fp8_scale: null argument or empty tensor");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_fp8_scale: bad dtype");
    COMAT_REQUIRE((((uintptr_t)x) & 15) == 0, "comat_fp8_scale: x must be 16-byte aligned");
    const int grid = grid_1d(n / 8, 256 * 4, 1024);
    if (dtype == COMAT_F32)
        hipLaunchKernelGGL(fp8_scale_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, scale,
                           (unsigned*)ws, amax_bits);
    else
        hipLaunchKernelGGL(fp8_scale_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n, scale,
                           (unsigned*)ws, amax_bits);
    return comat_check_launch("comat_fp8_scale");
}

extern "C" int comat_fp8_quantize(const void* x, int64_t n, int32_t dtype, const float* scale, void* y, void* stream) {
    COMAT_REQUIRE(x && scale && y && n > 0, "comat_fp8_quantize: null argument or empty tensor");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_fp8_quantize: bad dtype");
    COMAT_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0, "comat_fp8_quantize: x / y must be 16 / 8-byte aligned");
    const int grid = grid_1d(n / 8, 256 * 4, 2048);
    if (dtype == COMAT_F32)
        hipLaunchKernelGGL(fp8_quantize_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, n, scale,
                           (unsigned char*)y);
    else
        hipLaunchKernelGGL(fp8_quantize_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, n,
                           scale, (unsigned char*)y);
    return comat_check_launch("comat_fp8_quantize");
}
