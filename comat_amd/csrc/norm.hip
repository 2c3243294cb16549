// norm.hip — GroupNorm(+SiLU) and LayerNorm, forward and data-gradient, on channels-last token matrices.
// HBM-bound kernels: every pass reads whole rows (C contiguous elements) so that a wave's accesses coalesce.
// Statistics are accumulated per thread in fp32 and combined in fp64 in a FIXED order (no floating-point atomics), which
// keeps E[x^2]-E[x]^2 stable enough for the fp32 parity mode and the results bit-reproducible.
#include "gemm_shared.h"  // splitk_arrive_is_last: the agent-scope ticket protocol, reused for the statistics

namespace {

constexpr int NT = 256;
constexpr int MAX_G = 64;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- one-launch GroupNorm: a workgroup owns ONE (sample, group) and keeps its elements in registers ------------------
// The three-launch form below (statistics over row blocks -> fixed-order combine -> apply) uses the whole chip, which the
// VAE's 512^2 tensors need; the UNet's groups are small (64^2 x 10 channels = 41 k elements at most levels, 10 k at
// 16^2) and there each of the three launches is a ~5-9 us latency chain (profiles/r02_k_kernel_trace_eager.txt:
// gn_vstats 9.5 + gn_reduce 5 + gn_vapply2 8.3 us; 2 700 launches per C2 step).  Here B x G workgroups each read their
// group once (every thread holds <= MAXU register units of it), reduce in a fixed order (thread-sequential partials in
// fp32, fixed wave butterfly and wave-sequential combine in fp64), and write the result: one launch, one read of x.
// A register unit is VEC elements: 2 bf16 (one dword), or one element (fp32, or bf16 groups with an odd channel count).
// No floating-point atomics anywhere in this file: a shape that fits neither this form nor the vectorised three-launch
// form is rejected (COMAT_EUNSUPPORTED).
template <typename T, int VEC> struct GnUnit {
    static __device__ __forceinline__ uint32_t load(const T* p) {
        if (sizeof(T) == 4 || VEC == 2) return *(const uint32_t*)p;
        return (uint32_t)*(const uint16_t*)p;
    }
    static __device__ __forceinline__ void store(T* p, uint32_t u) {
        if (sizeof(T) == 4 || VEC == 2) *(uint32_t*)p = u;
        else *(uint16_t*)p = (uint16_t)u;
    }
    static __device__ __forceinline__ float get(uint32_t u, int e) {
        if (sizeof(T) == 4) return __uint_as_float(u);
        return bf16_to_f32((bf16_t)(e == 0 ? (u & 0xffffu) : (u >> 16)));
    }
    static __device__ __forceinline__ uint32_t pack(float a, float b2) {
        if (sizeof(T) == 4) return __float_as_uint(a);
        if (VEC == 1) return (uint32_t)f32_to_bf16(a);
        return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b2) << 16);
    }
};

// block-wide sum of two doubles in a fixed order; NTB / 64 waves
template <int NTB> __device__ __forceinline__ void gn_block_sum2(double& a, double& b, double* s_red) {
    a = wave_sum_f64(a);
    b = wave_sum_f64(b);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[2 * wave] = a;
        s_red[2 * wave + 1] = b;
    }
    __syncthreads();
    a = 0.0;
    b = 0.0;
#pragma unroll
    for (int w = 0; w < NTB / 64; ++w) {
        a += s_red[2 * w];
        b += s_red[2 * w + 1];
    }
}

constexpr int GN_ONE_MAXCPG = 256;

// MODE 0: y = silu?(xhat * gamma + beta), stats[b, g] = (mean, rstd).   MODE 1: dx = rstd * (g - (s1 + xhat * s2) / n) [+ add]
template <typename T, int VEC, int NTB, int MAXU, int MODE>
__global__ __launch_bounds__(NTB) void gn_one_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ stats, T* __restrict__ out, int HW, int C, int G,
                                                     float eps, int silu, const T* __restrict__ add) {
    typedef GnUnit<T, VEC> U;
    __shared__ float s_gamma[GN_ONE_MAXCPG], s_beta[GN_ONE_MAXCPG];
    __shared__ double s_red[2 * (NTB / 64)];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / G, upr = cpg / VEC, n_u = HW * upr;
    for (int c = tid; c < cpg; c += NTB) {
        s_gamma[c] = gamma[g * cpg + c];
        s_beta[c] = beta[g * cpg + c];
    }
    // uniform base pointers (SGPRs) + 32-bit per-lane element offsets (the host checks HW * C < 2^31)
    const int64_t base = (int64_t)b * HW * C + (int64_t)g * cpg;
    const T* xb = x + base;
    const T* gb = MODE == 1 ? dy + base : nullptr;
    const T* ab = (MODE == 1 && add) ? add + base : nullptr;
    T* ob = out + base;
    const int row0 = tid / upr, col0 = tid % upr, drow = NTB / upr, dcol = NTB % upr;
    float mean = 0.f, rstd = 0.f;
    if (MODE == 1) {
        mean = stats[2 * ((int64_t)b * G + g)];
        rstd = stats[2 * ((int64_t)b * G + g) + 1];
    }
    __syncthreads();
    uint32_t ux[MAXU], ug[MODE == 1 ? MAXU : 1];
    float a1 = 0.f, a2 = 0.f;
    {
        int row = row0, col = col0;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (tid + i * NTB < n_u) {
                const unsigned off = (unsigned)row * (unsigned)C + (unsigned)(col * VEC);
                ux[i] = U::load(xb + off);
                if (MODE == 1) ug[i] = U::load(gb + off);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xe = U::get(ux[i], e);
                    if (MODE == 0) {
                        a1 += xe;
                        a2 += xe * xe;
                    } else {
                        const int c = col * VEC + e;
                        const float xh = (xe - mean) * rstd;
                        float gg = U::get(ug[i], e);
                        if (silu) gg *= silu_grad_f(xh * s_gamma[c] + s_beta[c]);
                        gg *= s_gamma[c];
                        a1 += gg;
                        a2 += gg * xh;
                    }
                }
            }
            row += drow;
            col += dcol;
            if (col >= upr) {
                col -= upr;
                ++row;
            }
        }
    }
    double d1 = (double)a1, d2 = (double)a2;
    gn_block_sum2<NTB>(d1, d2, s_red);
    const double count = (double)HW * cpg;
    float s1 = 0.f, s2 = 0.f;
    if (MODE == 0) {
        const double m = d1 / count;
        double var = d2 / count - m * m;
        if (var < 0) var = 0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (tid == 0) {
            stats[2 * ((int64_t)b * G + g)] = mean;
            stats[2 * ((int64_t)b * G + g) + 1] = rstd;
        }
    } else {
        s1 = (float)d1;
        s2 = (float)d2;
    }
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    {
        int row = row0, col = col0;
        asm volatile("" : "+v"(row), "+v"(col));  // recompute the offsets: keeping MAXU of them live would spill
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (tid + i * NTB < n_u) {
                const unsigned off = (unsigned)row * (unsigned)C + (unsigned)(col * VEC);
                uint32_t ua = 0;
                if (MODE == 1 && add) ua = U::load(ab + off);
                float o[2] = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int c = col * VEC + e;
                    const float xh = (U::get(ux[i], e) - mean) * rstd;
                    if (MODE == 0) {
                        float y = xh * s_gamma[c] + s_beta[c];
                        if (silu) y = silu_f(y);
                        o[e] = y;
                    } else {
                        float gg = U::get(ug[i], e);
                        if (silu) gg *= silu_grad_f(xh * s_gamma[c] + s_beta[c]);
                        gg *= s_gamma[c];
                        float d = rstd * (gg - (s1 + xh * s2) * inv_n);
                        if (add) d += U::get(ua, e);
                        o[e] = d;
                    }
                }
                U::store(ob + off, U::pack(o[0], o[1]));
            }
            row += drow;
            col += dcol;
            if (col >= upr) {
                col -= upr;
                ++row;
            }
        }
    }
}

// Which one-launch variant serves a group of n_u register units: 0 = none (too large).  Forward keeps x (MAXU units per
// thread), backward keeps x and dy (2 * MAXU).
static inline int gn_one_variant(int64_t n_u, bool bwd) {
    if (n_u <= 256 * 24) return 1;                  // 256 threads x 24 units
    if (n_u <= 1024 * 16) return 2;                 // 1024 threads x 16
    if (n_u <= 1024 * (bwd ? 40 : 64)) return 3;    // 1024 threads x 64 (forward) / 40 (backward): <= 128 VGPRs per thread
    return 0;
}

template <typename T, int VEC, int MODE>
static bool gn_one_launch(const void* x, const void* dy, const float* gamma, const float* beta, float* stats, void* out,
                          int B, int64_t HW, int C, int G, float eps, int silu, const void* add, hipStream_t st) {
    const int cpg = C / G;
    const int64_t n_u = HW * (cpg / VEC);
    const int v = gn_one_variant(n_u, MODE == 1);
    if (v == 0) return false;
    const dim3 grid((unsigned)G, (unsigned)B);
#define GN_ONE(NTB_, MAXU_)                                                                                                  \
    hipLaunchKernelGGL((gn_one_kernel<T, VEC, NTB_, MAXU_, MODE>), grid, dim3(NTB_), 0, st, (const T*)x, (const T*)dy, gamma, \
                       beta, stats, (T*)out, (int)HW, C, G, eps, silu, (const T*)add)
    if (v == 1) GN_ONE(256, 24);
    else if (v == 2) GN_ONE(1024, 16);
    else if constexpr (MODE == 0) GN_ONE(1024, 64);
    else GN_ONE(1024, 40);
#undef GN_ONE
    return true;
}

// option norm_fused: 3 (default) = one launch wherever a group fits a workgroup's registers, 0 = always three launches,
// 1 / 2 = the two-launch forms.  -> true when the one-launch form took the call
template <int MODE>
static bool gn_try_one(const void* x, const void* dy, const float* gamma, const float* beta, float* stats, void* out, int B,
                       int64_t HW, int C, int G, float eps, int silu, const void* add, int dtype, bool must, hipStream_t st) {
    // Measured on MI355X (profiles/r03_f_mb_gn.txt): B x G workgroups reading 20..160-byte row pieces win where a group is
    // small (8x8 and 16x16 levels: 5.8 vs 10.7 us, 10.0 vs 15.0 us) and lose where the tensor needs the whole chip's
    // bandwidth (64x64 x 320: 28 vs 14 us) - so the one-launch form serves HW <= 256, and whatever the vectorised form
    // cannot take at all (`must`).
    // The backward form holds two tensors per thread and gains less: 16x16 x 1 280 channels 13.8 -> 16.6 us (a loss), x 2 560
    // 21.4 -> 14.4 us, 8x8 13.5 -> 7.4 us - it serves HW <= 64, and HW <= 256 from 1 920 channels on.
    const bool pays = MODE == 0 ? HW <= 256 : (HW <= 64 || (HW <= 256 && C >= 1920));
    const int nf = comat_option(COMAT_OPT_NORM_FUSED);
    if (!must && ((nf != 3 && nf != 4 && nf != 5) || !pays)) return false;
    const int cpg = C / G;
    if (cpg > GN_ONE_MAXCPG || HW * (int64_t)C >= (1ll << 31)) return false;
    if (dtype == COMAT_F32) {
        if ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy | (uintptr_t)add) & 3) != 0) return false;
        return gn_one_launch<float, 1, MODE>(x, dy, gamma, beta, stats, out, B, HW, C, G, eps, silu, add, st);
    }
    const bool even = cpg % 2 == 0 && C % 2 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy | (uintptr_t)add) & 3) == 0;
    if (even) return gn_one_launch<bf16_t, 2, MODE>(x, dy, gamma, beta, stats, out, B, HW, C, G, eps, silu, add, st);
    return gn_one_launch<bf16_t, 1, MODE>(x, dy, gamma, beta, stats, out, B, HW, C, G, eps, silu, add, st);
}

// ---- vectorised GroupNorm (C % (16/sizeof(T)) == 0): 16-byte accesses, fixed channel vector per thread -----------
union GV16 {
    uint4 u;
    float f[4];
    bf16_t h[8];
};
template <typename T> __device__ __forceinline__ float gv_get(const GV16& v, int e) {
    return sizeof(T) == 2 ? bf16_to_f32(v.h[e]) : v.f[e];
}
template <typename T> __device__ __forceinline__ void gv_set(GV16& v, int e, float x) {
    if (sizeof(T) == 2) v.h[e] = f32_to_bf16(x);
    else v.f[e] = x;
}

constexpr int VSLOTS = 2;  // channel vectors per thread: C <= 256 * 2 * EPV
constexpr int GN_TG = 16;   // blocks per first-level ticket counter of the two-launch form

// MODE 0: sum x, sum x^2.   MODE 1 (backward): s1 = sum gy*gamma, s2 = sum gy*gamma*xhat
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void gn_vstats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ stats, double* __restrict__ ws,
                                                       int HW, int C, int G, int silu, int rows_per_block,
                                                       unsigned* __restrict__ tickets, float* __restrict__ stats_out,
                                                       double count, float eps) {
    constexpr int EPV = 16 / sizeof(T);
    // per-(row lane, channel) partial sums: reduced in a FIXED order below, so the block result is deterministic
    // (LDS float atomics from many waves made run-to-run results differ in the last bf16 bit of a few outputs)
    __shared__ __attribute__((aligned(16))) float p_a[4096], p_b[4096];
    const int b = blockIdx.y;
    const int VPR = C / EPV;
    const int R = VPR >= NT ? 1 : NT / VPR;  // rows processed in parallel
    const int cpg = C / G;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, HW);
    const int64_t base = (int64_t)b * HW * C;
#pragma unroll
    for (int sl = 0; sl < VSLOTS; ++sl) {
        int v, rsub;
        if (VPR >= NT) { v = threadIdx.x + sl * NT; rsub = 0; }
        else { v = threadIdx.x % VPR; rsub = threadIdx.x / VPR; if (sl > 0 || rsub >= R) v = VPR; }
        if (v >= VPR) continue;
        const int c0 = v * EPV;
        float a1[EPV], a2[EPV], gm[EPV], bt[EPV], mu[EPV], rs[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            a1[e] = 0.f; a2[e] = 0.f;
            if (MODE == 1) {
                gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e];
                const float* st = stats + ((int64_t)b * G + (c0 + e) / cpg) * 2;
                mu[e] = st[0]; rs[e] = st[1];
            }
        }
        for (int r = r0 + rsub; r < r1; r += R) {
            GV16 xv, gv;
            xv.u = *(const uint4*)(x + base + (int64_t)r * C + c0);
            if (MODE == 1) gv.u = *(const uint4*)(dy + base + (int64_t)r * C + c0);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float xe = gv_get<T>(xv, e);
                if (MODE == 0) {
                    a1[e] += xe;
                    a2[e] += xe * xe;
                } else {
                    const float xh = (xe - mu[e]) * rs[e];
                    float g = gv_get<T>(gv, e);
                    if (silu) g *= silu_grad_f(xh * gm[e] + bt[e]);
                    g *= gm[e];
                    a1[e] += g;
                    a2[e] += g * xh;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            p_a[rsub * C + c0 + e] = a1[e];
            p_b[rsub * C + c0 + e] = a2[e];
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        float f1 = 0.f, f2 = 0.f;
        for (int rr = 0; rr < R; ++rr)
            for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
                f1 += p_a[rr * C + c];
                f2 += p_b[rr * C + c];
            }
        // per-block partial (no atomics): reduced in block order by gn_reduce_kernel -> bit-reproducible statistics
        double* part = ws + ((int64_t)(1 + blockIdx.x) * gridDim.y + b) * G * 2;
        if (tickets) {  // write-through (sc1) stores: another workgroup of THIS launch reads them (gemm_shared.h)
            __hip_atomic_store(part + threadIdx.x * 2 + 0, (double)f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + threadIdx.x * 2 + 1, (double)f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            part[threadIdx.x * 2 + 0] = (double)f1;
            part[threadIdx.x * 2 + 1] = (double)f2;
        }
    }
    if (tickets == nullptr) return;  // three-launch form: gn_reduce(_finalize)_kernel combines the partials
    // Two-launch form: the LAST block of this sample to arrive combines the per-block partials of its G groups, in the
    // same fixed order as gn_reduce(_finalize)_kernel (lane l takes slabs l, l+64, ..., then a fixed butterfly), so the
    // statistics stay bit-reproducible and identical to the three-launch form.
    // Two ticket levels (round 6): a memory-side atomic on ONE address serialises at ~20 ns per arrival - 512 blocks of a sample on one
    // counter cost the last of them ~11 us (measured: the form lost to three launches by exactly that, growing with the block
    // count).  Groups of GN_TG blocks share a first-level counter; the last of each group takes the sample's second-level ticket.
    {
        const int nblk = gridDim.x, ngrp = (nblk + GN_TG - 1) / GN_TG, grp = blockIdx.x / GN_TG;
        const int in_grp = min(GN_TG, nblk - grp * GN_TG);
        if (!splitk_ticket_is_last(tickets + gridDim.y + b * ngrp + grp, in_grp, (unsigned*)p_a)) return;
        if (!splitk_ticket_is_last(tickets + b, ngrp, (unsigned*)p_a)) return;
    }
    // Round 6: the last arriver's reads are ONE burst of independent 16-byte sc1 buffer loads (a (s1, s2) pair each), not a chain of
    // dependent atomic loads (round 5: every one of them waited out a full memory round trip - the two-launch form measured 26.8 us
    // against 14.1 for three launches).  Thread t owns group t % G and takes slabs t / G, t / G + S, ... (S = 256 / G); the S
    // partial sums of a group are then added in slab-lane order through LDS: a fixed order, bit-reproducible.
    {
        const int nblk = gridDim.x;
        const int64_t n = (int64_t)gridDim.y * G * 2;  // doubles per slab
        const int S = NT / G, g2 = threadIdx.x % G, sl = threadIdx.x / G;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ws, 0, 0x7ffffff0, 0x00020000);
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        double s1 = 0.0, s2 = 0.0;
        if (sl < S) {
            const int64_t off0 = ((int64_t)b * G + g2) * 16;  // byte offset of the group's pair inside a slab
#pragma unroll 4
            for (int k = sl; k < nblk; k += S) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((int64_t)(1 + k) * n * 8 + off0), 0, /*sc1*/ 16);
                double d0, d1;
                const unsigned lo[2] = {v[0], v[1]}, hi[2] = {v[2], v[3]};
                __builtin_memcpy(&d0, lo, 8);
                __builtin_memcpy(&d1, hi, 8);
                s1 += d0;
                s2 += d1;
            }
        }
        double* red = (double*)p_b;  // [S][G][2] doubles <= 4 KiB of the 16 KiB array (p_a holds the ticket flag)
        __syncthreads();
        if (sl < S) {
            red[(sl * G + g2) * 2] = s1;
            red[(sl * G + g2) * 2 + 1] = s2;
        }
        __syncthreads();
        if (threadIdx.x < G) {
            double t1 = 0.0, t2 = 0.0;
            for (int q = 0; q < S; ++q) {
                t1 += red[(q * G + threadIdx.x) * 2];
                t2 += red[(q * G + threadIdx.x) * 2 + 1];
            }
            const int64_t i = (int64_t)b * G + threadIdx.x;
            if (MODE == 0) {
                const double mean = t1 / count;
                double var = t2 / count - mean * mean;
                if (var < 0) var = 0;
                stats_out[2 * i] = (float)mean;
                stats_out[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
            } else {
                ws[2 * i] = t1;
                ws[2 * i + 1] = t2;
            }
        }
    }
}

// ws[0 .. n) = sum over blocks of the partial slabs ws[(1+blk)*n + i]: one wave per output, lane l takes slabs
// l, l+64, ... and the lanes are combined by a fixed butterfly -> a fixed summation order, but not a serial loop
__global__ void gn_reduce_kernel(double* __restrict__ ws, int nblk, int n) {
    const int i = blockIdx.x;
    double acc = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) acc += ws[(int64_t)(1 + k) * n + i];
    acc = wave_sum_f64(acc);
    if (threadIdx.x == 0) ws[i] = acc;
}

// forward: the same reduction fused with the mean / rstd finalisation (one wave per (sample, group))
__global__ void gn_reduce_finalize_kernel(const double* __restrict__ ws, int nblk, int ngroups, float* __restrict__ stats,
                                          double count, float eps) {
    const int i = blockIdx.x;
    const int n = ngroups * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        s1 += ws[(int64_t)(1 + k) * n + 2 * i];
        s2 += ws[(int64_t)(1 + k) * n + 2 * i + 1];
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if (threadIdx.x == 0) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        stats[2 * i] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// MODE 0: y = silu?(xhat*gamma+beta).   MODE 1: dx = rstd * (g - (s1 + xhat*s2)/n) [+ add]
// Normalisation / its gradient with every per-channel constant in REGISTERS: a thread owns one 16-byte channel vector
// (two when C > 256 vectors) and walks rows, R = 256 / (vectors per row) rows of a sample in parallel per block.  A
// grid-stride kernel that re-derives group index, statistics, gamma and beta per ELEMENT (an integer division and five
// dependent loads per 2 bytes of data) ran at ~1 TB/s on MI355X (round 1); here the inner loop is one 16-byte load per operand, 8 FMAs and
// one 16-byte store.  Same arithmetic, same operation order per element.
// FIN (option norm_fused = 2): no reduce / finalize launch between the statistics pass and this kernel - every block
// combines the (<= 64) per-block partial slabs of its sample's groups itself, in a fixed order (8 lanes per group take
// slabs sub, sub + 8, ...; then a fixed butterfly), before it touches the data.  Block 0 of a sample stores mean / rstd
// for the backward pass.
struct GnFin {
    const double* part;  // the workspace of the statistics pass (slab 1 + k at part + (1 + k) * B * G * 2)
    float* stats_out;    // MODE 0: [B, G, 2]
    int nblk;
    double count;
    float eps;
};

struct GnQ {  // MODE 0 with Q: the e4m3 bytes of the rounded output + its abs-max (see NormQ below)
    unsigned char* q8;
    const float* scale;
    unsigned* amax_bits;
};
template <typename T, int MODE, bool FIN = false, bool Q = false>
__global__ __launch_bounds__(NT) void gn_vapply2_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stats, const double* __restrict__ ws,
                                                        T* __restrict__ out, int HW, int C, int G, int silu,
                                                        int rows_per_block, const T* __restrict__ add, GnFin fin, GnQ gq) {
    constexpr int EPV = 16 / sizeof(T);
    const int b = blockIdx.y;
    __shared__ float sm_q[Q ? 4 : 1];
    float qinv = 0.f, qmax = 0.f;
    if (Q) qinv = 1.0f / *gq.scale;
    __shared__ float sm_a[FIN ? 64 : 1], sm_b[FIN ? 64 : 1];
    __shared__ __attribute__((aligned(16))) double sm_red[FIN ? 2 * NT : 1];
    if (FIN) {
        // Round 6: ONE burst of independent 16-byte loads per thread (thread t: group t % G, slabs t / G, t / G + S, ..), then the S
        // partial sums of a group are added in slab-lane order through LDS - one memory round trip per block instead of a chain
        // of <= 8 dependent pairs per lane followed by 64-bit shuffles (round-6 call gn_fin: that form cost the C2 step +3.6 ms).
        const int S = NT / G, g2 = threadIdx.x % G, sl = threadIdx.x / G;
        const int64_t n = (int64_t)gridDim.y * G * 2;
        double s1 = 0.0, s2 = 0.0;
        if (sl < S) {
            const double* src = fin.part + 2 * ((int64_t)b * G + g2);
#pragma unroll 8
            for (int k = sl; k < fin.nblk; k += S) {
                const double2 v = *(const double2*)(src + (int64_t)(1 + k) * n);
                s1 += v.x;
                s2 += v.y;
            }
            sm_red[(sl * G + g2) * 2] = s1;
            sm_red[(sl * G + g2) * 2 + 1] = s2;
        }
        __syncthreads();
        if (threadIdx.x < G) {
            double t1 = 0.0, t2 = 0.0;
            for (int q = 0; q < S; ++q) {
                t1 += sm_red[(q * G + threadIdx.x) * 2];
                t2 += sm_red[(q * G + threadIdx.x) * 2 + 1];
            }
            const int64_t i = (int64_t)b * G + threadIdx.x;
            if (MODE == 0) {
                const double mean = t1 / fin.count;
                double var = t2 / fin.count - mean * mean;
                if (var < 0) var = 0;
                const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)fin.eps));
                sm_a[threadIdx.x] = m;
                sm_b[threadIdx.x] = r;
                if (blockIdx.x == 0) {
                    fin.stats_out[2 * i] = m;
                    fin.stats_out[2 * i + 1] = r;
                }
            } else {
                sm_a[threadIdx.x] = (float)t1;
                sm_b[threadIdx.x] = (float)t2;
            }
        }
        __syncthreads();
    }
    const int VPR = C / EPV;
    const int R = VPR >= NT ? 1 : NT / VPR;
    const int cpg = C / G;
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(r0 + rows_per_block, HW);
    const int64_t base = (int64_t)b * HW * C;
#pragma unroll
    for (int sl = 0; sl < VSLOTS; ++sl) {
        int v, rsub;
        if (VPR >= NT) { v = threadIdx.x + sl * NT; rsub = 0; }
        else { v = threadIdx.x % VPR; rsub = threadIdx.x / VPR; if (sl > 0 || rsub >= R) v = VPR; }
        if (v >= VPR) continue;
        const int c0 = v * EPV;
        float gm[EPV], bt[EPV], mu[EPV], rs[EPV], s1[EPV], s2[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int grp = (c0 + e) / cpg;
            const int64_t sg = (int64_t)b * G + grp;
            gm[e] = gamma[c0 + e];
            bt[e] = beta[c0 + e];
            if (FIN && MODE == 0) {
                mu[e] = sm_a[grp];
                rs[e] = sm_b[grp];
            } else {
                mu[e] = stats[2 * sg];
                rs[e] = stats[2 * sg + 1];
            }
            if (MODE == 1) {
                s1[e] = FIN ? sm_a[grp] : (float)ws[2 * sg];
                s2[e] = FIN ? sm_b[grp] : (float)ws[2 * sg + 1];
            }
        }
        for (int r = r0 + rsub; r < r1; r += R) {
            const int64_t off = base + (int64_t)r * C + c0;
            GV16 xv, gv, av, ov;
            xv.u = *(const uint4*)(x + off);
            if (MODE == 1) gv.u = *(const uint4*)(dy + off);
            if (MODE == 1 && add) av.u = *(const uint4*)(add + off);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float xh = (gv_get<T>(xv, e) - mu[e]) * rs[e];
                if (MODE == 0) {
                    float y = xh * gm[e] + bt[e];
                    if (silu) y = silu_f(y);
                    gv_set<T>(ov, e, y);
                } else {
                    float g = gv_get<T>(gv, e);
                    if (silu) g *= silu_grad_f(xh * gm[e] + bt[e]);
                    g *= gm[e];
                    float d = rs[e] * (g - (s1[e] + xh * s2[e]) * inv_n);
                    if (add) d += gv_get<T>(av, e);
                    gv_set<T>(ov, e, d);
                }
            }
            *(uint4*)(out + off) = ov.u;
            if (Q) {
                float r[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    r[e] = gv_get<T>(ov, e);
                    qmax = fmaxf(qmax, fabsf(r[e]));
                }
                if (EPV == 8) *(uint2*)(gq.q8 + off) = fp8_pack8(r, qinv);
                else *(unsigned*)(gq.q8 + off) = fp8_pack4(r, qinv);
            }
        }
    }
    if (Q) {
        qmax = block_max_256(qmax, sm_q);
        if (threadIdx.x == 0) fp8_amax_track(gq.amax_bits, qmax);
    }
}

// rows per block of the apply pass: ~512 blocks, at least 4 rows per row lane
static inline int gn_apply_rows_per_block(int B, int64_t HW, int C, int epv) {
    const int vpr = C / epv;
    const int R = vpr >= NT ? 1 : NT / vpr;
    int64_t rpb = cdiv64((int64_t)B * HW, 512);
    if (rpb < 4 * R) rpb = 4 * R;
    if (rpb > HW) rpb = HW;
    return (int)rpb;
}

static inline int gn_rows_per_block(int B, int64_t HW, int C, int epv) {
    const int vpr = C / epv;
    const int R = vpr >= NT ? 1 : NT / vpr;
    int64_t rpb = cdiv64((int64_t)B * HW, 1024);  // ~4 blocks per CU
    if (rpb < 4 * R) rpb = 4 * R;
    if (rpb < cdiv64(HW, 1024)) rpb = cdiv64(HW, 1024);  // at most 1024 partial slabs in the workspace
    if (rpb > HW) rpb = HW;
    return (int)rpb;
}

// workspace: [GN_TICKETS uint32 ticket counters (zeroed once by the caller, re-armed by the kernels) | doubles]
constexpr int GN_TICKETS = 1024;
// (4 = the one-launch form where it pays, the two-launch ticket form elsewhere; its counters: B second-level + B * ceil(nblk / GN_TG)
// first-level ones - a call that needs more takes the three-launch form)
static inline bool gn_two_launch(int B, int nblk) {
    const int o = comat_option(COMAT_OPT_NORM_FUSED);
    return (o == 1 || o == 4) && (int64_t)B * (1 + (nblk + GN_TG - 1) / GN_TG) <= GN_TICKETS;
}
// norm_fused = 2 / 5: the apply kernel finalises (gn_vapply2_kernel<.., FIN>): every block sums the per-block partials of its sample in
// its prologue (5 = with the one-launch form where that pays, like 3)
// Measured (profiles/r06_ag_gn_fin_burst.txt): forward 18.5 -> 16.4 us at 2 x 64^2 x 640, 28.4 -> 23.0 at 128^2 x 512, equal on the small
// levels; the VAE's 256^2 / 512^2 tensors lose (the slab cap starves their statistics pass): they keep three launches.
constexpr int GN_FIN_MAX_SLABS = 256;
constexpr int64_t GN_FIN_MAX_HW = 16384;
static inline bool gn_fin_in_apply(int G, int64_t HW) {
    const int o = comat_option(COMAT_OPT_NORM_FUSED);
    return (o == 2 || (o == 5 && HW <= GN_FIN_MAX_HW)) && G <= 64;
}

template <typename T>
static void gn_fwd_vec(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws_all, int B,
                       int64_t HW, int C, int G, float eps, int silu, hipStream_t st, const GnQ* q = nullptr) {
    constexpr int EPV = 16 / sizeof(T);
    const bool fin = gn_fin_in_apply(G, HW);
    int rpb = gn_rows_per_block(B, HW, C, EPV);
    if (fin && rpb < cdiv64(HW, GN_FIN_MAX_SLABS)) rpb = (int)cdiv64(HW, GN_FIN_MAX_SLABS);  // bounded prologue of the apply blocks
    dim3 sg((unsigned)cdiv64(HW, rpb), (unsigned)B);
    unsigned* tickets = (unsigned*)ws_all;
    double* ws = ws_all + GN_TICKETS / 2;
    const bool fused = !fin && gn_two_launch(B, (int)sg.x);
    hipLaunchKernelGGL((gn_vstats_kernel<T, 0>), sg, dim3(NT), 0, st, (const T*)x, (const T*)nullptr, gamma, beta,
                       (const float*)nullptr, ws, (int)HW, C, G, silu, rpb, fused ? tickets : (unsigned*)nullptr, stats,
                       (double)HW * (C / G), eps);
    if (!fused && !fin)
        hipLaunchKernelGGL(gn_reduce_finalize_kernel, dim3(B * G), dim3(64), 0, st, (const double*)ws, (int)sg.x, B * G,
                           stats, (double)HW * (C / G), eps);
    const int arpb = gn_apply_rows_per_block(B, HW, C, EPV);
    const dim3 ag((unsigned)cdiv64(HW, arpb), (unsigned)B);
    const GnFin f = {ws, stats, (int)sg.x, (double)HW * (C / G), eps};
    const GnQ noq = {nullptr, nullptr, nullptr};
    if (q && fin)  // (same statistics path as the plain call: the two forms of a norm stay bit-identical in y and stats)
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 0, true, true>), ag, dim3(NT), 0, st, (const T*)x, (const T*)nullptr, gamma, beta,
                           (const float*)stats, (const double*)nullptr, (T*)y, (int)HW, C, G, silu, arpb, (const T*)nullptr,
                           f, *q);
    else if (q)
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 0, false, true>), ag, dim3(NT), 0, st, (const T*)x, (const T*)nullptr, gamma, beta,
                           (const float*)stats, (const double*)nullptr, (T*)y, (int)HW, C, G, silu, arpb, (const T*)nullptr,
                           f, *q);
    else if (fin)
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 0, true>), ag, dim3(NT), 0, st, (const T*)x, (const T*)nullptr, gamma, beta,
                           (const float*)stats, (const double*)nullptr, (T*)y, (int)HW, C, G, silu, arpb, (const T*)nullptr,
                           f, noq);
    else
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 0, false>), ag, dim3(NT), 0, st, (const T*)x, (const T*)nullptr, gamma, beta,
                           (const float*)stats, (const double*)nullptr, (T*)y, (int)HW, C, G, silu, arpb, (const T*)nullptr,
                           f, noq);
}

template <typename T>
static void gn_bwd_vec(const void* dy, const void* x, const float* gamma, const float* beta, const float* stats, void* dx,
                       double* ws_all, int B, int64_t HW, int C, int G, int silu, const void* add, hipStream_t st) {
    constexpr int EPV = 16 / sizeof(T);
    const bool fin = gn_fin_in_apply(G, HW);
    int rpb = gn_rows_per_block(B, HW, C, EPV);
    if (fin && rpb < cdiv64(HW, GN_FIN_MAX_SLABS)) rpb = (int)cdiv64(HW, GN_FIN_MAX_SLABS);
    dim3 sg((unsigned)cdiv64(HW, rpb), (unsigned)B);
    unsigned* tickets = (unsigned*)ws_all;
    double* ws = ws_all + GN_TICKETS / 2;
    const bool fused = !fin && gn_two_launch(B, (int)sg.x);
    hipLaunchKernelGGL((gn_vstats_kernel<T, 1>), sg, dim3(NT), 0, st, (const T*)x, (const T*)dy, gamma, beta, stats, ws,
                       (int)HW, C, G, silu, rpb, fused ? tickets : (unsigned*)nullptr, (float*)nullptr, 0.0, 0.0f);
    if (!fused && !fin) hipLaunchKernelGGL(gn_reduce_kernel, dim3(B * G * 2), dim3(64), 0, st, ws, (int)sg.x, B * G * 2);
    const int arpb = gn_apply_rows_per_block(B, HW, C, EPV);
    const dim3 ag((unsigned)cdiv64(HW, arpb), (unsigned)B);
    const GnFin f = {ws, nullptr, (int)sg.x, 0.0, 0.0f};
    if (fin)
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 1, true>), ag, dim3(NT), 0, st, (const T*)x, (const T*)dy, gamma, beta, stats,
                           (const double*)ws, (T*)dx, (int)HW, C, G, silu, arpb, (const T*)add, f, GnQ{nullptr, nullptr, nullptr});
    else
        hipLaunchKernelGGL((gn_vapply2_kernel<T, 1, false>), ag, dim3(NT), 0, st, (const T*)x, (const T*)dy, gamma, beta, stats,
                           (const double*)ws, (T*)dx, (int)HW, C, G, silu, arpb, (const T*)add, f, GnQ{nullptr, nullptr, nullptr});
}

static inline bool gn_vec_ok(const void* a, const void* b, int C, int dtype, int64_t HW) {
    const int epv = dtype == COMAT_BF16 ? 8 : 4;
    const int vpr = C / epv;
    const int R = vpr >= NT ? 1 : NT / (vpr > 0 ? vpr : 1);
    return (C % epv) == 0 && vpr <= NT * VSLOTS && (int64_t)R * C <= 4096 && ((uintptr_t)a % 16) == 0 &&
           ((uintptr_t)b % 16) == 0 && HW < (1ll << 31);
}

// ---- LayerNorm: one wave per row ----------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ y,
                                                    float* __restrict__ stats, int64_t M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ldf<T>(xr + c);
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = ldf<T>(xr + c) - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    T* yr = y + row * C;
    for (int c = lane; c < C; c += 64) stf<T>(yr + c, (ldf<T>(xr + c) - mean) * rstd * gamma[c] + beta[c]);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ gamma, const float* __restrict__ stats,
                                                    T* __restrict__ dx, int64_t M, int C, const T* __restrict__ add) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + row * C;
    const T* gr = dy + row * C;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float g = ldf<T>(gr + c) * gamma[c];
        const float xh = (ldf<T>(xr + c) - mean) * rstd;
        s1 += g;
        s2 += g * xh;
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    T* dr = dx + row * C;
    for (int c = lane; c < C; c += 64) {
        const float g = ldf<T>(gr + c) * gamma[c];
        const float xh = (ldf<T>(xr + c) - mean) * rstd;
        float d = rstd * (g - s1 - xh * s2);
        if (add) d += ldf<T>(add + row * C + c);  // gradient of the branch that bypasses the norm (residual)
        stf<T>(dr + c, d);
    }
}

// ---- vectorised LayerNorm: one wave per row, the row lives in registers (ONE 16-byte read per element group) --------
// VPL = 16-byte vectors per lane (C <= 64 * VPL * EPV).  Same two-pass formulas as the scalar kernels above.
// Q (fp8 forward, delayed scaling): the e4m3 bytes of the ROUNDED output under the consumer's scale leave with it (q8 [M, C]) and
// its abs-max is folded into the site's running maximum - the bits of comat_fp8_quantize_scaled over y, without its launch and
// without reading y back.
struct NormQ {
    unsigned char* q8;
    const float* scale;
    unsigned* amax_bits;
};
template <typename T, int VPL, bool Q = false>
__global__ __launch_bounds__(NT) void ln_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y,
                                                        float* __restrict__ stats, int64_t M, int C, float eps, NormQ nq) {
    constexpr int EPV = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C / EPV;
    float v[VPL][EPV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        GV16 t;
        t.u = make_uint4(0, 0, 0, 0);
        if (vi < nvec) t.u = *(const uint4*)(x + row * C + (int64_t)vi * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            v[i][e] = vi < nvec ? gv_get<T>(t, e) : 0.f;
            s += v[i][e];
        }
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        if (lane + i * 64 < nvec) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float d = v[i][e] - mean;
                q += d * d;
            }
        }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    float qinv = 0.f, qmax = 0.f;
    if (Q) qinv = 1.0f / *nq.scale;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const float4 g0 = *(const float4*)(gamma + vi * EPV), b0 = *(const float4*)(beta + vi * EPV);
            float gg[8], bb[8];
            gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w;
            bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w;
            if (EPV == 8) {
                const float4 g1 = *(const float4*)(gamma + vi * EPV + 4), b1 = *(const float4*)(beta + vi * EPV + 4);
                gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
                bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
            }
            GV16 o;
#pragma unroll
            for (int e = 0; e < EPV; ++e) gv_set<T>(o, e, (v[i][e] - mean) * rstd * gg[e] + bb[e]);
            *(uint4*)(y + row * C + (int64_t)vi * EPV) = o.u;
            if (Q) {
                float r[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    r[e] = gv_get<T>(o, e);
                    qmax = fmaxf(qmax, fabsf(r[e]));
                }
                unsigned char* qp = nq.q8 + row * C + (int64_t)vi * EPV;
                if (EPV == 8) *(uint2*)qp = fp8_pack8(r, qinv);
                else *(unsigned*)qp = fp8_pack4(r, qinv);
            }
        }
    }
    if (Q) {
        qmax = wave_max(qmax);
        if (lane == 0) fp8_amax_track(nq.amax_bits, qmax);
    }
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

template <typename T, int VPL>
__global__ __launch_bounds__(NT) void ln_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ stats,
                                                        T* __restrict__ dx, int64_t M, int C, const T* __restrict__ add) {
    constexpr int EPV = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C / EPV;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float g[VPL][EPV], xh[VPL][EPV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            GV16 tg, tx;
            tg.u = *(const uint4*)(dy + row * C + (int64_t)vi * EPV);
            tx.u = *(const uint4*)(x + row * C + (int64_t)vi * EPV);
            const float4 g0 = *(const float4*)(gamma + vi * EPV);
            float gg[8];
            gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w;
            if (EPV == 8) {
                const float4 g1 = *(const float4*)(gamma + vi * EPV + 4);
                gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
            }
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                g[i][e] = gv_get<T>(tg, e) * gg[e];
                xh[i][e] = (gv_get<T>(tx, e) - mean) * rstd;
                s1 += g[i][e];
                s2 += g[i][e] * xh[i][e];
            }
        }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            GV16 o, ta;
            if (add) ta.u = *(const uint4*)(add + row * C + (int64_t)vi * EPV);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float d = rstd * (g[i][e] - s1 - xh[i][e] * s2);
                if (add) d += gv_get<T>(ta, e);
                gv_set<T>(o, e, d);
            }
            *(uint4*)(dx + row * C + (int64_t)vi * EPV) = o.u;
        }
    }
}

static inline int ln_vpl(int C, int dtype, const void* a, const void* b, const void* c, const void* d) {
    const int epv = dtype == COMAT_BF16 ? 8 : 4;
    if (C % epv) return 0;
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) != 0) return 0;
    const int vpl = (C / epv + 63) / 64;
    return vpl <= 4 ? vpl : 0;
}

}  // namespace

extern "C" int comat_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   double* ws, int32_t B, int64_t HW, int32_t C, int32_t G, float eps, int32_t silu,
                                   int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && gamma && beta && y && stats && ws, "comat_groupnorm_fwd: null pointer");
    COMAT_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "comat_groupnorm_fwd: bad shape");
    COMAT_REQUIRE(G <= MAX_G && B <= 65535, "comat_groupnorm_fwd: G or B too large");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_groupnorm_fwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = gn_vec_ok(x, y, C, dtype, HW);
    // one launch where a (sample, group) fits a workgroup's registers (every UNet level); the vectorised three-launch form
    // for the large tensors of the VAE; nothing else: a shape neither takes is an error, not a slower kernel
    if (gn_try_one<0>(x, nullptr, gamma, beta, stats, y, B, HW, C, G, eps, silu, nullptr, dtype, !vec, st))
        return comat_check_launch("comat_groupnorm_fwd");
    if (!vec) {
        comat_set_error("comat_groupnorm_fwd: unsupported shape (C = %d, G = %d, HW = %lld): groups of more than 65 536 "
                        "register units need C %% %d == 0 and 16-byte aligned tensors", C, G, (long long)HW,
                        dtype == COMAT_BF16 ? 8 : 4);
        return COMAT_EUNSUPPORTED;
    }
    if (dtype == COMAT_BF16) gn_fwd_vec<bf16_t>(x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, st);
    else gn_fwd_vec<float>(x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, st);
    return comat_check_launch("comat_groupnorm_fwd");
}

// 1 when comat_groupnorm_fwd_q takes the shape: the vectorised three-launch form, i.e. not the shapes the one-launch form serves
extern "C" int comat_groupnorm_fwd_q_ok(int32_t B, int64_t HW, int32_t C, int32_t G, int32_t dtype) {
    if (!dtype_ok(dtype) || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G || G > MAX_G || B > 65535) return 0;
    if (comat_option(COMAT_OPT_NORM_FUSED) >= 3 && HW <= 256 && C / G <= GN_ONE_MAXCPG)
        return 0;  // gn_try_one<0> "pays"
    return gn_vec_ok(nullptr, nullptr, C, dtype, HW) ? 1 : 0;
}

extern "C" int comat_groupnorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y, float* stats, double* ws,
                                     int32_t B, int64_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* q8,
                                     const float* scale, uint32_t* amax_bits, void* stream) {
    COMAT_REQUIRE(x && gamma && beta && y && stats && ws && q8 && scale && amax_bits, "comat_groupnorm_fwd_q: null pointer");
    if (!comat_groupnorm_fwd_q_ok(B, HW, C, G, dtype) || !gn_vec_ok(x, y, C, dtype, HW) || (((uintptr_t)q8) & 7) != 0) {
        comat_set_error("comat_groupnorm_fwd_q: only the vectorised three-launch form emits e4m3 bytes (B = %d, HW = %lld, C = %d, "
                        "G = %d): run comat_groupnorm_fwd and comat_fp8_quantize_scaled", B, (long long)HW, C, G);
        return COMAT_EUNSUPPORTED;
    }
    const GnQ q = {(unsigned char*)q8, scale, amax_bits};
    if (dtype == COMAT_BF16) gn_fwd_vec<bf16_t>(x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, (hipStream_t)stream, &q);
    else gn_fwd_vec<float>(x, gamma, beta, y, stats, ws, B, HW, C, G, eps, silu, (hipStream_t)stream, &q);
    return comat_check_launch("comat_groupnorm_fwd_q");
}

extern "C" int comat_layernorm_fwd_q_ok(int32_t C, int32_t dtype) {
    return dtype_ok(dtype) && ln_vpl(C, dtype, nullptr, nullptr, nullptr, nullptr) ? 1 : 0;
}

extern "C" int comat_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                                   const float* stats, void* dx, double* ws, int32_t B, int64_t HW, int32_t C,
                                   int32_t G, int32_t silu, const void* add, int32_t dtype, void* stream) {
    COMAT_REQUIRE(dy && x && gamma && beta && stats && dx && ws, "comat_groupnorm_bwd: null pointer");
    COMAT_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "comat_groupnorm_bwd: bad shape");
    COMAT_REQUIRE(G <= MAX_G && B <= 65535, "comat_groupnorm_bwd: G or B too large");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_groupnorm_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = gn_vec_ok(x, dx, C, dtype, HW) && ((uintptr_t)dy % 16) == 0 && (!add || ((uintptr_t)add % 16) == 0);
    if (gn_try_one<1>(x, dy, gamma, beta, (float*)stats, dx, B, HW, C, G, 0.f, silu, add, dtype, !vec, st))
        return comat_check_launch("comat_groupnorm_bwd");
    if (!vec) {
        comat_set_error("comat_groupnorm_bwd: unsupported shape (C = %d, G = %d, HW = %lld): groups of more than 40 960 "
                        "register units need C %% %d == 0 and 16-byte aligned tensors", C, G, (long long)HW,
                        dtype == COMAT_BF16 ? 8 : 4);
        return COMAT_EUNSUPPORTED;
    }
    if (dtype == COMAT_BF16) gn_bwd_vec<bf16_t>(dy, x, gamma, beta, stats, dx, ws, B, HW, C, G, silu, add, st);
    else gn_bwd_vec<float>(dy, x, gamma, beta, stats, dx, ws, B, HW, C, G, silu, add, st);
    return comat_check_launch("comat_groupnorm_bwd");
}

extern "C" int comat_layernorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y, float* stats, int64_t M,
                                     int32_t C, float eps, int32_t dtype, void* q8, const float* scale, uint32_t* amax_bits,
                                     void* stream) {
    COMAT_REQUIRE(x && gamma && beta && y && stats && q8 && scale && amax_bits, "comat_layernorm_fwd_q: null pointer");
    COMAT_REQUIRE(M > 0 && C > 0 && dtype_ok(dtype), "comat_layernorm_fwd_q: bad shape or dtype");
    const int vpl = (((uintptr_t)q8) & 7) == 0 ? ln_vpl(C, dtype, x, y, gamma, beta) : 0;
    if (!vpl) {
        comat_set_error("comat_layernorm_fwd_q: only the vectorised form emits e4m3 bytes (C = %d a multiple of %d up to %d, 16-byte "
                        "aligned tensors): run comat_layernorm_fwd and comat_fp8_quantize_scaled", C, dtype == COMAT_BF16 ? 8 : 4,
                        dtype == COMAT_BF16 ? 2048 : 1024);
        return COMAT_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv64(M, NT / 64));
    const NormQ nq = {(unsigned char*)q8, scale, amax_bits};
#define LN_FWDQ(T, V) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, V, true>), grid, dim3(NT), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, C, eps, nq)
    if (dtype == COMAT_BF16) {
        if (vpl == 1) LN_FWDQ(bf16_t, 1); else if (vpl == 2) LN_FWDQ(bf16_t, 2); else if (vpl == 3) LN_FWDQ(bf16_t, 3); else LN_FWDQ(bf16_t, 4);
    } else {
        if (vpl == 1) LN_FWDQ(float, 1); else if (vpl == 2) LN_FWDQ(float, 2); else if (vpl == 3) LN_FWDQ(float, 3); else LN_FWDQ(float, 4);
    }
#undef LN_FWDQ
    return comat_check_launch("comat_layernorm_fwd_q");
}

extern "C" int comat_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   int64_t M, int32_t C, float eps, int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && gamma && beta && y && stats, "comat_layernorm_fwd: null pointer");
    COMAT_REQUIRE(M > 0 && C > 0, "comat_layernorm_fwd: bad shape");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_layernorm_fwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv64(M, NT / 64));
    const int vpl = ln_vpl(C, dtype, x, y, gamma, beta);
    const NormQ noq = {nullptr, nullptr, nullptr};
#define LN_FWD(T, V) hipLaunchKernelGGL((ln_fwd_vec_kernel<T, V>), grid, dim3(NT), 0, st, (const T*)x, gamma, beta, (T*)y, stats, M, C, eps, noq)
    if (vpl && dtype == COMAT_BF16) {
        if (vpl == 1) LN_FWD(bf16_t, 1); else if (vpl == 2) LN_FWD(bf16_t, 2); else if (vpl == 3) LN_FWD(bf16_t, 3); else LN_FWD(bf16_t, 4);
        return comat_check_launch("comat_layernorm_fwd");
    }
    if (vpl && dtype == COMAT_F32) {
        if (vpl == 1) LN_FWD(float, 1); else if (vpl == 2) LN_FWD(float, 2); else if (vpl == 3) LN_FWD(float, 3); else LN_FWD(float, 4);
        return comat_check_launch("comat_layernorm_fwd");
    }
#undef LN_FWD
    if (dtype == COMAT_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, dim3(NT), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y,
                           stats, M, C, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, dim3(NT), 0, st, (const float*)x, gamma, beta, (float*)y,
                           stats, M, C, eps);
    return comat_check_launch("comat_layernorm_fwd");
}

extern "C" int comat_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                   int64_t M, int32_t C, const void* add, int32_t dtype, void* stream) {
    COMAT_REQUIRE(dy && x && gamma && stats && dx, "comat_layernorm_bwd: null pointer");
    COMAT_REQUIRE(M > 0 && C > 0, "comat_layernorm_bwd: bad shape");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_layernorm_bwd: bad dtype");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv64(M, NT / 64));
    const int vpl = (!add || ((uintptr_t)add & 15) == 0) ? ln_vpl(C, dtype, dy, x, dx, gamma) : 0;
#define LN_BWD(T, V) hipLaunchKernelGGL((ln_bwd_vec_kernel<T, V>), grid, dim3(NT), 0, st, (const T*)dy, (const T*)x, gamma, stats, (T*)dx, M, C, (const T*)add)
    if (vpl && dtype == COMAT_BF16) {
        if (vpl == 1) LN_BWD(bf16_t, 1); else if (vpl == 2) LN_BWD(bf16_t, 2); else if (vpl == 3) LN_BWD(bf16_t, 3); else LN_BWD(bf16_t, 4);
        return comat_check_launch("comat_layernorm_bwd");
    }
    if (vpl && dtype == COMAT_F32) {
        if (vpl == 1) LN_BWD(float, 1); else if (vpl == 2) LN_BWD(float, 2); else if (vpl == 3) LN_BWD(float, 3); else LN_BWD(float, 4);
        return comat_check_launch("comat_layernorm_bwd");
    }
#undef LN_BWD
    if (dtype == COMAT_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, dim3(NT), 0, st, (const bf16_t*)dy, (const bf16_t*)x, gamma,
                           stats, (bf16_t*)dx, M, C, (const bf16_t*)add);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, dim3(NT), 0, st, (const float*)dy, (const float*)x, gamma,
                           stats, (float*)dx, M, C, (const float*)add);
    return comat_check_launch("comat_layernorm_bwd");
}
