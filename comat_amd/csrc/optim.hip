// optim.hip — optimizer tail on the flat fp32 trainable buffers: global-norm reduction and fused clip + AdamW.
// (training_script.py:661-664,692-694: clip_grad_norm_ then AdamW.step, one HBM pass over p/g/m/v.)
#include "common.h"

namespace {

constexpr int NT = 256;

// Two stages with a fixed summation order: the global gradient norm (and with it the clip factor) must come out
// bit-identical on every data-parallel rank, or the replicas drift apart; float atomics would make it order-dependent.
__global__ __launch_bounds__(NT) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n,
                                                           float* __restrict__ ws) {
    __shared__ float sbuf[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) acc += x[i] * x[i];
    acc = block_sum_256(acc, sbuf);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}
__global__ __launch_bounds__(NT) void sumsq_final_kernel(const float* __restrict__ ws, int nparts,
                                                         float* __restrict__ out) {
    __shared__ float sbuf[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += NT) acc += ws[i];
    acc = block_sum_256(acc, sbuf);
    if (threadIdx.x == 0) out[0] += acc;
}

__global__ __launch_bounds__(NT) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                   const int32_t* __restrict__ step_dev,
                                                   const float* __restrict__ gnorm_sq, float max_norm, float grad_scale) {
    // grad_scale: the buffer holds the SUM of the data-parallel ranks' gradients (RCCL all-reduce SUM); 1 / world turns
    // it - and its norm - into the mean here, instead of one more pass over the buffer
    float clip = grad_scale;
    if (step_dev) {  // step count of applied updates lives on the device (see comat_adamw_tick)
        const float t = (float)(*step_dev + 1);
        bc1 = 1.0f - powf(b1, t);
        bc2s = sqrtf(1.0f - powf(b2, t));
    }
    // a non-finite gradient norm (overflow / NaN somewhere in backward) skips the update, like the GradScaler step
    // of the reference's mixed-precision run (accelerate, training_script.py:661-664): parameters and moments stay
    if (gnorm_sq && !isfinite(*gnorm_sq)) return;
    if (gnorm_sq && max_norm > 0.f) {
        const float c = max_norm / (sqrtf(*gnorm_sq) * grad_scale + 1e-6f);
        clip = c < 1.0f ? c * grad_scale : grad_scale;
    }
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float pi = p[i] * (1.0f - lr * wd);
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
        p[i] = pi;
    }
}

__global__ void adamw_tick_kernel(int32_t* __restrict__ counters, const float* __restrict__ gnorm_sq) {
    if (threadIdx.x == 0) counters[isfinite(*gnorm_sq) ? 0 : 1] += 1;
}

}  // namespace

extern "C" int comat_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream) {
    COMAT_REQUIRE(x && out && ws && n > 0, "comat_sumsq: bad args");
    const int parts = grid_1d(n, NT, 1024);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(parts), dim3(NT), 0, (hipStream_t)stream, x, n, ws);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, (const float*)ws, parts, out);
    return comat_check_launch("comat_sumsq");
}

extern "C" int comat_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int32_t step, const int32_t* step_dev,
                           const float* gnorm_sq, float max_norm, float grad_scale, void* stream) {
    COMAT_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev) && grad_scale > 0.f, "comat_adamw: bad args");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2s, step_dev, gnorm_sq, max_norm, grad_scale);
    return comat_check_launch("comat_adamw");
}

extern "C" int comat_adamw_tick(int32_t* counters, const float* gnorm_sq, void* stream) {
    COMAT_REQUIRE(counters && gnorm_sq, "comat_adamw_tick: null pointer");
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, gnorm_sq);
    return comat_check_launch("comat_adamw_tick");
}
