// gemm_shared.h — pieces shared by the two GEMM translation units (gemm.hip: the 64x64 register-staged kernel that
// serves every operand layout and the fp32 parity mode; gemm2.hip: the LDS-DMA pipelined bf16 kernel for the
// k-contiguous shapes that carry the FLOPs).
#pragma once
#include "common.h"
#include <stdlib.h>

struct Epi {
    void* C;
    const float* bias;
    const float* bias2;
    const void* R;
    int64_t ldc, ldr, rows_per_b2;
    float alpha, beta;
    int act, out_dt, r_dt;
    // second epilogue (comat_gemm_params::epi2, pipelined kernel only): GEGLU over value / gate columns interleaved in 16s
    void* C2;
    int64_t ldc2;
    int epi2;
    // epi2 == 4 (tail columns): the last n2 columns of the product go to C2 as alpha2 * product
    int64_t n2;
    float alpha2;
    // GEGLU epilogues under the fp8 forward with delayed scaling (comat_gemm_params::q8): the e4m3 bytes of value * gelu(gate) for the
    // layer that consumes it, its abs-max folded into that site's running maximum; C2 (the bf16 copy) may then be NULL
    unsigned char* q8;
    const float* q_scale;
    unsigned* q_amax;
    int64_t ldq8;
};

// XCD-aware workgroup -> work-item map.  Workgroup b is dispatched to XCD b % 8 (MI355X: 8 XCDs, a private 4 MiB L2
// each).  Give every XCD one CONTIGUOUS chunk of the linear work list, ordered so that neighbours share the same
// activation rows and stream the (small) weight panel: the chunk's operands then stay resident in that XCD's L2
// instead of every XCD thrashing over the whole problem.  Bijective for any count; affects speed only.
__device__ __forceinline__ int64_t xcd_chunk_map(int64_t bid, int64_t n) {
    const int64_t q = n >> 3, r = n & 7, xcd = bid & 7, idx = bid >> 3;
    const int64_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---------------------------------------------------------------------------------------------------------------
// Split-K without a reduce launch.  Every block of a tile's `splits` k-slices stores its fp32 partial tile to its
// slab (lane-linear layout: the reader loads exactly what the writer stored, 16 bytes per lane, fully coalesced), then
// takes a ticket on the tile's counter.  The block that draws the last ticket adds the slabs IN SLICE ORDER (its own
// slice from the slab too, so the summation order does not depend on who arrives last: results are bit-reproducible)
// and runs the fused epilogue.
// Visibility between workgroups (per-CU L1 and per-XCD L2 are not coherent): the slabs are stored WRITE-THROUGH
// (sc1) and read back with sc1 loads, every storing wave drains its stores (s_waitcnt vmcnt(0)) before the block's
// ticket - the fence-free form of the CDNA4 guide's hand-off recipe (cdna_hip_programming.md, guideline 16, R1: "sc1
// loads may replace the acquire only when the producer stored sc1").  The first version of this code used plain stores
// + an agent-scope release fence per block + an acquire in the last arriver: correct, but every block's release
// (buffer_wbl2) writes back its XCD's dirty L2 lines, which cost ~10 us per launch on MI355X
// (profiles/r02_b_*: GroupNorm statistics 9 -> 19 us, 64-slice weight-gradient GEMMs 25 -> 40 us).
// The counter is re-armed (0) by the last arriver, so the caller zeroes the counter region ONCE
// (include/comat_hip.h: COMAT_WS_COUNTER_BYTES).  `lds_flag`: one dword of the block's single LDS array.
// ---------------------------------------------------------------------------------------------------------------
struct SlabIO {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ explicit SlabIO(float* base)
        : rs(__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000)) {}
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    __device__ __forceinline__ void store(int64_t byte_off, const f32x4_t& v) const {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, (int)byte_off, 0, /*sc1*/ 16);
    }
    __device__ __forceinline__ f32x4_t load(int64_t byte_off) const {
        return __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)byte_off, 0, /*sc1*/ 16));
    }
};

__device__ __forceinline__ bool splitk_ticket_is_last(unsigned* counter, int splits, unsigned* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through slab stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old == (unsigned)(splits - 1);
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
        *lds_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

// ---- device pieces shared by the MFMA kernels of gemm2.hip and gemm3.hip -------------------------------------------------
__device__ __forceinline__ void mma_t(f32x16_t& acc, const short8_t& wfrag, const short8_t& xfrag) {
    // D[n][m] += W[n][k] X[m][k]: A operand = weight fragment (lane&31 = n), B operand = activation fragment (lane&31 = m)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wfrag), __builtin_bit_cast(bf16x8_t, xfrag),
                                                  acc, 0, 0, 0);
}

// 16-byte write-through store / L1-bypassing load (sc1): data handed from one workgroup to another inside a launch
__device__ __forceinline__ void store16_wt(void* p, const uint4& v) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const u32x4_t w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

// lanes 32..63 of `lo` <-> lanes 0..31 of `hi`
__device__ __forceinline__ void half_swap(float& lo, float& hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

union Pack16 {
    uint4 u;
    bf16_t h[8];
    float f[4];
};

// 8 consecutive output columns n .. n+7 of output row m
// WT: the 16-byte stores are WRITE-THROUGH (sc1): for a tile that other workgroups of the SAME launch read with sc1 loads
// (no fence needed: cdna_hip_programming.md, guideline 16).  No caller sets it today (the chained launches of round 4 did).
template <bool WT = false>
__device__ __forceinline__ void epilogue_run(const Epi& ep, float* v, int64_t m, int64_t n, int64_t N, bool vec) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= ep.alpha;
    if (vec) {
        if (ep.bias) {
            const float4 b0 = *(const float4*)(ep.bias + n), b1 = *(const float4*)(ep.bias + n + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (ep.bias2) {
            const float* p2 = ep.bias2 + (int64_t)((unsigned)m / (unsigned)ep.rows_per_b2) * N + n;
            const float4 b0 = *(const float4*)p2, b1 = *(const float4*)(p2 + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (ep.act == COMAT_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        } else if (ep.act == COMAT_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        }
        if (ep.R) {
            if (ep.r_dt == COMAT_F32) {
                const float* pr = (const float*)ep.R + m * ep.ldr + n;
                const float4 r0 = *(const float4*)pr, r1 = *(const float4*)(pr + 4);
                v[0] += ep.beta * r0.x; v[1] += ep.beta * r0.y; v[2] += ep.beta * r0.z; v[3] += ep.beta * r0.w;
                v[4] += ep.beta * r1.x; v[5] += ep.beta * r1.y; v[6] += ep.beta * r1.z; v[7] += ep.beta * r1.w;
            } else {
                Pack16 pk;
                pk.u = *(const uint4*)((const bf16_t*)ep.R + m * ep.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ep.beta * bf16_to_f32(pk.h[e]);
            }
        }
        if (ep.out_dt == COMAT_F32) {
            float* pc = (float*)ep.C + m * ep.ldc + n;
            *(float4*)pc = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(pc + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            Pack16 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.h[e] = f32_to_bf16(v[e]);
            if (WT) store16_wt((bf16_t*)ep.C + m * ep.ldc + n, pk.u);
            else *(uint4*)((bf16_t*)ep.C + m * ep.ldc + n) = pk.u;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t col = n + e;
            if (col < N) {
                float x = v[e];
                if (ep.bias) x += ep.bias[col];
                if (ep.bias2) x += ep.bias2[(int64_t)((unsigned)m / (unsigned)ep.rows_per_b2) * N + col];
                if (ep.act == COMAT_ACT_SILU) x = silu_f(x);
                else if (ep.act == COMAT_ACT_GELU) x = gelu_f(x);
                if (ep.R) x += ep.beta * ld_dt(ep.R, m * ep.ldr + col, ep.r_dt);
                st_dt(ep.C, m * ep.ldc + col, x, ep.out_dt);
            }
        }
    }
}

// epi2 == 4: 8 consecutive TAIL columns c .. c + 7 (column index inside C2) of output row m: C2[m, c ..] = alpha2 * v
__device__ __forceinline__ void epilogue_tail(const Epi& ep, const float* v, int64_t m, int64_t c) {
    if (ep.out_dt == COMAT_F32) {
        float* pc = (float*)ep.C2 + m * ep.ldc2 + c;
        *(float4*)pc = make_float4(v[0] * ep.alpha2, v[1] * ep.alpha2, v[2] * ep.alpha2, v[3] * ep.alpha2);
        *(float4*)(pc + 4) = make_float4(v[4] * ep.alpha2, v[5] * ep.alpha2, v[6] * ep.alpha2, v[7] * ep.alpha2);
    } else {
        Pack16 pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) pk.h[e] = f32_to_bf16(v[e] * ep.alpha2);
        *(uint4*)((bf16_t*)ep.C2 + m * ep.ldc2 + c) = pk.u;
    }
}

// GEGLU epilogue (comat_gemm_params::epi2): `v` = the lane's 16 values of one output row after the half swaps - v[0..8) are
// columns nt + 8 h .. (value channels), v[8..16) columns nt + 16 + 8 h .. (their gate channels), nt = first column of the
// 32-column tile.  out = value * gelu(gate) for 8 channels -> ONE 16-byte store into C2 [M, N / 2]; with epi2 == 1 the
// pre-activations are stored too (C, same layout as without the fusion: the backward pass reads them).  Both halves are
// rounded to the storage type BEFORE the product, exactly what the unfused pair of kernels computes.
__device__ __forceinline__ void epilogue_geglu(const Epi& ep, const float* v, int64_t m, int64_t nb, int64_t nt, int h, float qinv,
                                               float& qmax) {
    float a[8], g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = v[e] * ep.alpha;
        g[e] = v[8 + e] * ep.alpha;
    }
    if (ep.bias) {
        const float4 a0 = *(const float4*)(ep.bias + nb), a1 = *(const float4*)(ep.bias + nb + 4);
        const float4 g0 = *(const float4*)(ep.bias + nb + 16), g1 = *(const float4*)(ep.bias + nb + 20);
        a[0] += a0.x; a[1] += a0.y; a[2] += a0.z; a[3] += a0.w; a[4] += a1.x; a[5] += a1.y; a[6] += a1.z; a[7] += a1.w;
        g[0] += g0.x; g[1] += g0.y; g[2] += g0.z; g[3] += g0.w; g[4] += g1.x; g[5] += g1.y; g[6] += g1.z; g[7] += g1.w;
    }
    Pack16 pa, pg, po;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        pa.h[e] = f32_to_bf16(a[e]);
        pg.h[e] = f32_to_bf16(g[e]);
        po.h[e] = f32_to_bf16(bf16_to_f32(pa.h[e]) * gelu_f(bf16_to_f32(pg.h[e])));
    }
    if (ep.epi2 == 1) {
        *(uint4*)((bf16_t*)ep.C + m * ep.ldc + nb) = pa.u;
        *(uint4*)((bf16_t*)ep.C + m * ep.ldc + nb + 16) = pg.u;
    }
    if (ep.C2) *(uint4*)((bf16_t*)ep.C2 + m * ep.ldc2 + (nt >> 1) + 8 * h) = po.u;
    if (ep.q8) {  // the bytes comat_fp8_quantize_scaled would make of the stored (rounded) product
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r[e] = bf16_to_f32(po.h[e]);
            qmax = fmaxf(qmax, fabsf(r[e]));
        }
        *(uint2*)(ep.q8 + m * ep.ldq8 + (nt >> 1) + 8 * h) = fp8_pack8(r, qinv);
    }
}

// GEGLU BACKWARD epilogue (epi2 == 3): `v` = 8 consecutive columns d .. d + 7 of row m of the product dF = g W2 (the gradient of the
// feed-forward's hidden activations f = value * gelu(gate), [M, D]).  C2 holds the saved pre-activations [M, 2 D] in the interleaved
// layout (value channels 16 t .. at columns 32 t .., their gates 16 columns further), C receives their gradient in the same layout:
//   d value = dF * gelu(gate),   d gate = dF * value * gelu'(gate)
// dF is rounded to bf16 first - the bits the two-launch form (GEMM, then comat_geglu_il_bwd) computes.
__device__ __forceinline__ void epilogue_geglu_bwd(const Epi& ep, const float* v, int64_t m, int64_t d) {
    const int64_t cv = ((d >> 4) << 5) + (d & 15), cg = cv + 16;
    Pack16 pa, pg, oa, og;
    pa.u = *(const uint4*)((const bf16_t*)ep.C2 + m * ep.ldc2 + cv);
    pg.u = *(const uint4*)((const bf16_t*)ep.C2 + m * ep.ldc2 + cg);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float go = bf16_to_f32(f32_to_bf16(v[e] * ep.alpha));
        const float a = bf16_to_f32(pa.h[e]), g = bf16_to_f32(pg.h[e]);
        oa.h[e] = f32_to_bf16(go * gelu_f(g));
        og.h[e] = f32_to_bf16(go * a * gelu_grad_f(g));
    }
    *(uint4*)((bf16_t*)ep.C + m * ep.ldc + cv) = oa.u;
    *(uint4*)((bf16_t*)ep.C + m * ep.ldc + cg) = og.u;
}

// ---- host side ------------------------------------------------------------------------------------------------
constexpr int64_t WS_COUNTERS = COMAT_WS_COUNTER_BYTES / 4;  // ticket counters at the head of the workspace

// gemm2.hip: the pipelined bf16 kernel.  Each returns >0 when it took the problem (1 pipelined, 2 k-major), 0 when the shape is not
// eligible (the caller falls through to the general kernel), <0 on error.
int comat_gemm2_try_gemm(const comat_gemm_params* p, void* stream);
int comat_gemm2_try_segments(const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, void* stream);
int comat_gemm2_try_conv(const comat_conv_params* p, void* stream);
// gemm3.hip: the lean k-parallel-wave kernel (-> 5 when it took the problem, 0 otherwise)
int comat_gemm3_try(const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, bool bias_per_batch, void* stream);
