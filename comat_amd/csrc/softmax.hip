// softmax.hip — row softmax of attention scores and its backward (one 64-lane wave per row).
// The probability map written here IS the tensor the reference's AttentionStore clones
// (attn_utils/tc_attn_utils.py:60-68): it is written once with lane-contiguous (coalesced) stores and is kept for the
// backward pass and for the attribute-concentration loss, so the capture costs no extra HBM traffic.
#include "common.h"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void softmax_fwd_kernel(const void* __restrict__ S, void* __restrict__ P, int64_t rows,
                                                         int cols, int q_len, int causal, int causal_offset,
                                                         const int8_t* __restrict__ key_mask, int64_t rows_per_mask,
                                                         int s_dt, int p_dt) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * cols;
    int limit = cols;  // keys j < limit are visible
    if (causal) {
        const int q = (int)(row % q_len);
        const int l = q + causal_offset + 1;
        limit = l < cols ? l : cols;
        if (limit < 0) limit = 0;
    }
    const int8_t* km = key_mask ? key_mask + (row / rows_per_mask) * cols : nullptr;
    float m = -INFINITY;
    for (int j = lane; j < limit; j += 64)
        if (!km || km[j]) m = fmaxf(m, ld_dt(S, off + j, s_dt));
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < limit; j += 64)
        if (!km || km[j]) sum += __expf(ld_dt(S, off + j, s_dt) - m);
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int j = lane; j < cols; j += 64) {
        float p = 0.f;
        if (j < limit && (!km || km[j])) p = __expf(ld_dt(S, off + j, s_dt) - m) * inv;
        st_dt(P, off + j, p, p_dt);
    }
}

__global__ __launch_bounds__(NT) void softmax_bwd_kernel(const void* __restrict__ P, const void* __restrict__ dP,
                                                         void* __restrict__ dS, int64_t rows, int cols, float scale,
                                                         int p_dt, int dp_dt, int ds_dt) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t off = row * cols;
    float dot = 0.f;
    for (int j = lane; j < cols; j += 64) dot += ld_dt(P, off + j, p_dt) * ld_dt(dP, off + j, dp_dt);
    dot = wave_sum(dot);
    for (int j = lane; j < cols; j += 64) {
        const float p = ld_dt(P, off + j, p_dt);
        st_dt(dS, off + j, scale * p * (ld_dt(dP, off + j, dp_dt) - dot), ds_dt);
    }
}

}  // namespace

extern "C" int comat_softmax_fwd(const void* S, void* P, int64_t rows, int32_t cols, int32_t q_len, int32_t causal,
                                 int32_t causal_offset, const int8_t* key_mask, int64_t rows_per_mask,
                                 int32_t s_dtype, int32_t p_dtype, void* stream) {
    COMAT_REQUIRE(S && P, "comat_softmax_fwd: null pointer");
    COMAT_REQUIRE(rows > 0 && cols > 0, "comat_softmax_fwd: bad shape");
    COMAT_REQUIRE(!causal || q_len > 0, "comat_softmax_fwd: causal needs q_len");
    COMAT_REQUIRE(!key_mask || rows_per_mask > 0, "comat_softmax_fwd: key_mask needs rows_per_mask");
    COMAT_REQUIRE(dtype_ok(s_dtype) && dtype_ok(p_dtype), "comat_softmax_fwd: bad dtype");
    COMAT_REQUIRE(cdiv64(rows, NT / 64) < (1ll << 31), "comat_softmax_fwd: too many rows");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)cdiv64(rows, NT / 64)), dim3(NT), 0, (hipStream_t)stream, S,
                       P, rows, cols, q_len > 0 ? q_len : 1, causal, causal_offset, key_mask,
                       rows_per_mask > 0 ? rows_per_mask : 1, s_dtype, p_dtype);
    return comat_check_launch("comat_softmax_fwd");
}

extern "C" int comat_softmax_bwd(const void* P, const void* dP, void* dS, int64_t rows, int32_t cols, float scale,
                                 int32_t p_dtype, int32_t dp_dtype, int32_t ds_dtype, void* stream) {
    COMAT_REQUIRE(P && dP && dS, "comat_softmax_bwd: null pointer");
    COMAT_REQUIRE(rows > 0 && cols > 0, "comat_softmax_bwd: bad shape");
    COMAT_REQUIRE(dtype_ok(p_dtype) && dtype_ok(dp_dtype) && dtype_ok(ds_dtype), "comat_softmax_bwd: bad dtype");
    COMAT_REQUIRE(cdiv64(rows, NT / 64) < (1ll << 31), "comat_softmax_bwd: too many rows");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)cdiv64(rows, NT / 64)), dim3(NT), 0, (hipStream_t)stream, P,
                       dP, dS, rows, cols, scale, p_dtype, dp_dtype, ds_dtype);
    return comat_check_launch("comat_softmax_bwd");
}
