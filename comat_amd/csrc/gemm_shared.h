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

// ---- host side ------------------------------------------------------------------------------------------------
constexpr int64_t WS_COUNTERS = COMAT_WS_COUNTER_BYTES / 4;  // ticket counters at the head of the workspace

// gemm2.hip: the pipelined bf16 kernel.  Each returns >0 when it took the problem (1 pipelined, 2 k-major), 0 when the shape is not
// eligible (the caller falls through to the general kernel), <0 on error.
int comat_gemm2_try_gemm(const comat_gemm_params* p, void* stream);
int comat_gemm2_try_segments(const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, void* stream);
int comat_gemm2_try_conv(const comat_conv_params* p, void* stream);
