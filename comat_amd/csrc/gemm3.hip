// gemm3.hip — the "lean" MFMA GEMM for the launch-latency-bound products of the step, and CHAINED launches (a producer
// GEMM and the K-segmented GEMM that consumes its output in ONE launch).
//
// Why a second kernel.  profiles/r03_z_gap_table.txt: 3 208 of the 4 876 MFMA launches of a C2 step have a roofline time
// under 2 us and take 36.9 ms - the LoRA low-rank projections (M x 128 x C: 8 us for a 0.2 us problem), the K-segmented
// "frozen + low-rank" products of the 32^2 / 16^2 / 8^2 levels, the BLIP decoder.  On the pipelined kernel (gemm2.hip)
// such a problem is a LATENCY CHAIN: every 64-byte k-tile is DMA wait -> block barrier -> LDS reads -> a handful of MFMAs,
// ~0.11 us per link whatever the tile shape, 40 links for K = 1280.  This kernel has no chain:
//   * no LDS staging and no barrier in the main loop.  A wave loads its MFMA fragments straight from global memory into
//     registers (each lane reads 64 CONTIGUOUS bytes of its operand row per 64-element k-chunk: four dwordx4 loads that
//     walk one 128-byte line per row pair) and keeps the next chunk's 16 loads in flight while the MFMAs of the current
//     chunk issue (register ping-pong; the compiler counts vmcnt for plain loads);
//   * the waves of a workgroup split K, not the tile: every wave owns the workgroup's WHOLE output tile for 1 / NW of the
//     chunks (k-parallel waves), so a K = 1280 problem is 3 - 5 chunks deep per wave instead of 40 k-tiles;
//   * the partial tiles meet in LDS ONCE, after the loop: NW lane-linear fp32 images (16-byte accesses, conflict-free),
//     one block barrier, then every wave sums one half tile over the NW partials in wave order (fixed order:
//     bit-reproducible) and runs the fused epilogue of gemm2.hip on it (same accumulator layout, same 16-byte stores).
// Global traffic per workgroup equals the LDS-shared design's ((BM + BN) x K x 2 bytes: every operand row of the tile is
// read once, by the one wave that owns its k-range); what is lost is the 4x reuse of a 128 x 128 block tile, so problems
// that fill the chip with big tiles stay on gemm2.hip (option gemm3 / the rule in g3_wants).
// The lane -> k assignment inside a chunk (lane (r, h) of MFMA k-step j holds k = 32 h + 8 j .. + 8 of row r) is the same
// for both operands, so - as in gemm2.hip - the permutation cancels.
//
// CHAINED launch (comat_gemm_chain).  Every LoRA projection is two dependent products: h = s x D^T (M x r, tiny) and
// y = [x | h] [W | U]^T (training_utils/pipeline.py:84-115: `up(down(x))` added to the frozen Linear); the backward pass
// has the same shape (u = s g U, then dx = [g | u] [W^T | D^T]^T).  As two launches the small one costs 5 - 8 us plus a
// kernel boundary, 1 100 times per step.  Here the workgroups with the lowest block ids compute the producer's tiles
// (write-through stores, then one counter per 32/64-row block), the others compute the consumer's tiles and need the
// producer's rows only for their LAST k-segment: by the time a consumer wave gets there - after the frozen part of its
// range - the rows are long done, so the wait is a formality and the producer overlaps the consumer's main product.
// Dispatch order makes it safe: a workgroup is only ever waiting for workgroups with LOWER block ids, which every XCD's
// dispatcher has started before it (and the spin is bounded: a give-up raises a flag the host checks in tests).
// Visibility: producer tiles are stored sc1 (write-through) and drained (vmcnt(0)) before the counter, consumers poll the
// counter relaxed and read the rows with sc1 loads (cdna_hip_programming.md guideline 16, R1 - the split-K protocol of
// gemm_shared.h).  The counters re-arm themselves (the last consumer of a row block zeroes them).
#include "gemm_shared.h"

namespace {

constexpr int CKB = 128;        // bytes of k per chunk and operand row (64 bf16)
constexpr int CKE = 64;         // k elements per chunk
constexpr int MAXSEG3 = 8;
constexpr unsigned SPIN_LIMIT = 1u << 24;

struct Seg3 {
    const char* A;
    const char* B;
    int64_t lda, ldb, sA, sB;  // bytes
    int nch;                   // chunks of this segment
};

struct Prob3 {  // one (batched) K-segmented GEMM
    Seg3 seg[MAXSEG3];
    int nseg, nch;
    int64_t M, N;
    int tiles_m, tiles_n, batch;
    int64_t sC, sR, sBias;  // batch strides (elements)
    Epi ep;
    int vec;
};

struct Args3 {
    Prob3 main;
    Prob3 pre;          // producer of a chained launch (n_pre > 0): main's LAST segment reads pre's output rows
    int n_pre;          // producer work items = the first n_pre workgroups
    int need;           // producer tiles per row block (pre.tiles_n * pre.batch)
    int cons;           // consumer tiles per row block (main.tiles_n * main.batch)
    unsigned* done;     // [tiles_m] producer tiles finished, per row block (zero before the launch, re-armed by it)
    unsigned* seen;     // [tiles_m] consumer tiles finished
    unsigned* err;      // set to 1 when a spin gave up
};

template <int TM, int TN>
struct Frags {
    short8_t x[TM][4];  // activation fragments: [32-row tile][k-step]
    short8_t w[TN][4];  // weight fragments
};

// TM x TN MFMA tiles of 32 x 32 per workgroup, NW waves splitting K
template <int TM, int TN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm3_kernel(Args3 g) {
    constexpr int T = TM * TN, BM = TM * 32, BN = TN * 32;
    __shared__ __attribute__((aligned(16))) float4 smem[NW * T * 4 * 64];  // the ONLY LDS object: NW partial tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool is_pre = (int)blockIdx.x < g.n_pre;  // wave-uniform
    const Prob3& P = is_pre ? g.pre : g.main;

    unsigned lin = is_pre ? blockIdx.x : (unsigned)xcd_chunk_map((int64_t)blockIdx.x - g.n_pre, (int64_t)gridDim.x - g.n_pre);
    const int tn = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)P.tiles_n));
    lin /= (unsigned)P.tiles_n;
    const int tm = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)P.tiles_m));
    const int64_t z = __builtin_amdgcn_readfirstlane((int)(lin / (unsigned)P.tiles_m));
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // this wave's chunk range [c0, c0 + cnt) of the problem's chunk list (segments concatenated)
    const int q = P.nch / NW, rem = P.nch % NW;
    const int c0 = wave * q + (wave < rem ? wave : rem);
    const int cnt = __builtin_amdgcn_readfirstlane(q + (wave < rem ? 1 : 0));

    int64_t arow[TM], brow[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int64_t m = m0 + a * 32 + r;
        arow[a] = m < P.M ? m : P.M - 1;  // rows >= M are never stored: any valid row will do
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int64_t n = n0 + b * 32 + r;
        brow[b] = n < P.N ? n : P.N - 1;
    }

    // ---- chunk cursor over the segment list ------------------------------------------------------------------------------------
    // A consumer wave of a chained launch runs its chunks in two phases: the chunks of the ordinary segments (plain global
    // loads), then the chunks of the LAST segment, whose A rows another workgroup of this launch produced (sc1 buffer loads,
    // after the row block's counter says they are complete).  Each phase is its own software pipeline: the steady-state loop
    // issues the loads of two chunks unconditionally (the compiler's vmcnt bookkeeping is exact only without branches
    // around the loads).
    int seg = 0, seg_left = 0;
    const char* pa[TM];
    const char* pb[TN];
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.seg[0].A, 0, 0x7ffffff0, 0x00020000);
    int voff[TM];
    auto enter_segment = [&](int s, int skip) {  // position the cursor at chunk `skip` of segment s
        const Seg3& sg = P.seg[s];
        seg = s;
        seg_left = sg.nch - skip;
#pragma unroll
        for (int b = 0; b < TN; ++b) pb[b] = sg.B + z * sg.sB + brow[b] * sg.ldb + (int64_t)skip * CKB + h * 64;
#pragma unroll
        for (int a = 0; a < TM; ++a) pa[a] = sg.A + z * sg.sA + arow[a] * sg.lda + (int64_t)skip * CKB + h * 64;
    };
    auto load_plain = [&](Frags<TM, TN>& f) {
        if (seg_left == 0) enter_segment(seg + 1, 0);  // wave-uniform; no loads inside
        --seg_left;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f.x[a][j] = *(const short8_t*)(pa[a] + 16 * j);
            pa[a] += CKB;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f.w[b][j] = *(const short8_t*)(pb[b] + 16 * j);
            pb[b] += CKB;
        }
    };
    auto load_chained = [&](Frags<TM, TN>& f) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                f.x[a][j] = __builtin_bit_cast(short8_t, __builtin_amdgcn_raw_buffer_load_b128(rsA, voff[a] + 16 * j, 0, /*sc1*/ 16));
            voff[a] += CKB;
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) f.w[b][j] = *(const short8_t*)(pb[b] + 16 * j);
            pb[b] += CKB;
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    auto mma_chunk = [&](const Frags<TM, TN>& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) mma_t(acc[a][b], f.w[b][j], f.x[a][j]);
    };
    // n chunks through the register ping-pong: the next chunk's loads are in flight while this chunk's MFMAs issue
    auto run = [&](int n, auto&& load) {
        if (n <= 0) return;
        Frags<TM, TN> f0, f1;
        load(f0);
        int i = 1;
        for (; i + 1 < n; i += 2) {
            load(f1);
            mma_chunk(f0);
            load(f0);
            mma_chunk(f1);
        }
        if (i < n) {
            load(f1);
            mma_chunk(f0);
            mma_chunk(f1);
        } else {
            mma_chunk(f0);
        }
    };

    // ---- main loop ------------------------------------------------------------------------------------------------------------
    const bool has_chain = !is_pre && g.n_pre > 0;
    const int tail = has_chain ? P.seg[P.nseg - 1].nch : 0;  // chunks of the chained segment (the LAST chunks of the list)
    const int n_plain_all = P.nch - tail;
    int n_plain = n_plain_all - c0;
    n_plain = n_plain < 0 ? 0 : (n_plain > cnt ? cnt : n_plain);
    const int n_chain = cnt - n_plain;
    if (n_plain > 0) {
        int s = 0, skip = c0;
        while (s + 1 < P.nseg && skip >= P.seg[s].nch) {
            skip -= P.seg[s].nch;
            ++s;
        }
        enter_segment(s, skip);
        run(n_plain, load_plain);
    }
    if (n_chain > 0) {  // wave-uniform
        const Seg3& sg = P.seg[P.nseg - 1];
        const int skip = c0 + n_plain - n_plain_all;  // first chained chunk of this wave
        // wait until the producer's tiles of this row block are complete (lane 0 polls: relaxed, agent scope)
        if (lane == 0) {
            unsigned n = 0;
            while (__hip_atomic_load(g.done + tm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.need) {
                if (++n > SPIN_LIMIT) {
                    __hip_atomic_store(g.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.A + z * sg.sA), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int a = 0; a < TM; ++a) voff[a] = (int)(arow[a] * sg.lda + (int64_t)skip * CKB + h * 64);
#pragma unroll
        for (int b = 0; b < TN; ++b) pb[b] = sg.B + z * sg.sB + brow[b] * sg.ldb + (int64_t)skip * CKB + h * 64;
        run(n_chain, load_chained);
    }

    // ---- the NW partial tiles meet in LDS: image [wave][tile][quad][lane] of 16 bytes ----------------------------------------
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                smem[((wave * T + a * TN + b) * 4 + qd) * 64 + lane] =
                    make_float4(acc[a][b][4 * qd], acc[a][b][4 * qd + 1], acc[a][b][4 * qd + 2], acc[a][b][4 * qd + 3]);
    __syncthreads();

    // ---- every wave sums half tiles (quads 2 hf, 2 hf + 1 of a 32 x 32 tile = 16 columns) over the partials IN WAVE ORDER and
    // runs the fused epilogue on them.  acc[i] of lane (r, h): row r, column (i & 3) + 8 (i >> 2) + 4 h (gemm2.hip) ----
    Epi ep = P.ep;
    ep.C = (char*)ep.C + z * P.sC * (ep.out_dt == COMAT_F32 ? 4 : 2);
    if (ep.R) ep.R = (const char*)ep.R + z * P.sR * (ep.r_dt == COMAT_F32 ? 4 : 2);
    if (ep.bias) ep.bias += z * P.sBias;
    const bool vec = P.vec != 0;
    for (int u = wave; u < 2 * T; u += NW) {
        const int t = u >> 1, hf = u & 1;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.0f;
#pragma unroll
        for (int p = 0; p < NW; ++p) {
            const float4 x0 = smem[((p * T + t) * 4 + 2 * hf) * 64 + lane];
            const float4 x1 = smem[((p * T + t) * 4 + 2 * hf + 1) * 64 + lane];
            v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w;
            v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) half_swap(v[j], v[4 + j]);
        const int a = t / TN, b = t % TN;
        const int64_t m = m0 + a * 32 + r;
        const int64_t nb = n0 + b * 32 + 16 * hf + 8 * h;
        if (m < P.M && nb < P.N) {
            if (is_pre) epilogue_run<true>(ep, v, m, nb, P.N, vec);
            else epilogue_run<false>(ep, v, m, nb, P.N, vec);
        }
    }

    // ---- chained launch: publish / retire ------------------------------------------------------------------------------------
    if (g.n_pre > 0) {
        if (is_pre) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(g.done + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();  // every wave of this tile is past its wait
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(g.seen + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)(g.cons - 1)) {  // the last consumer of this row block re-arms both counters
                    __hip_atomic_store(g.done + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(g.seen + tm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
enum { G3_AUTO = 0, G3_32x32x8 = 1, G3_64x32x8 = 2, G3_32x64x8 = 3, G3_64x64x4 = 4, G3_64x64x8 = 5, G3_32x32x4 = 6,
       G3_64x32x4 = 7, G3_32x32x16 = 8, G3_64x32x16 = 9, G3_LAST = 9 };

struct Cfg3 {
    int bm, bn, nw;
};
Cfg3 cfg3_dims(int c) {
    switch (c) {
        case G3_32x32x8: return {32, 32, 8};
        case G3_64x32x8: return {64, 32, 8};
        case G3_32x64x8: return {32, 64, 8};
        case G3_64x64x8: return {64, 64, 8};
        case G3_32x32x4: return {32, 32, 4};
        case G3_64x32x4: return {64, 32, 4};
        case G3_32x32x16: return {32, 32, 16};
        case G3_64x32x16: return {64, 32, 16};
        default: return {64, 64, 4};
    }
}

void launch3(int c, const Args3& a, unsigned blocks, hipStream_t st) {
    switch (c) {
        case G3_32x32x8: hipLaunchKernelGGL((gemm3_kernel<1, 1, 8>), dim3(blocks), dim3(512), 0, st, a); break;
        case G3_64x32x8: hipLaunchKernelGGL((gemm3_kernel<2, 1, 8>), dim3(blocks), dim3(512), 0, st, a); break;
        case G3_32x64x8: hipLaunchKernelGGL((gemm3_kernel<1, 2, 8>), dim3(blocks), dim3(512), 0, st, a); break;
        case G3_64x64x8: hipLaunchKernelGGL((gemm3_kernel<2, 2, 8>), dim3(blocks), dim3(512), 0, st, a); break;
        case G3_32x32x4: hipLaunchKernelGGL((gemm3_kernel<1, 1, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        case G3_64x32x4: hipLaunchKernelGGL((gemm3_kernel<2, 1, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        case G3_32x32x16: hipLaunchKernelGGL((gemm3_kernel<1, 1, 16>), dim3(blocks), dim3(1024), 0, st, a); break;
        case G3_64x32x16: hipLaunchKernelGGL((gemm3_kernel<2, 1, 16>), dim3(blocks), dim3(1024), 0, st, a); break;
        default: hipLaunchKernelGGL((gemm3_kernel<2, 2, 4>), dim3(blocks), dim3(256), 0, st, a); break;
    }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Prob3 from the ABI structs; false when the problem is not eligible (k-contiguous bf16, whole 64-element chunks, 16-byte rows)
bool fill_prob(Prob3& P, const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, bool bias_per_batch) {
    if (p->in_dtype != COMAT_BF16 || p->transA || p->transB || p->batch2 > 1 || nseg < 1 || nseg > MAXSEG3) return false;
    if (p->M < 1 || p->N < 8 || p->M >= (1ll << 31) || p->N >= (1ll << 31)) return false;
    const int64_t batch = p->batch1 > 1 ? p->batch1 : 1;
    if (batch > 1 && p->bias2) return false;
    int64_t nch = 0;
    for (int s = 0; s < nseg; ++s) {
        const comat_gemm_segment& q = segs[s];
        if (q.K % CKE || q.lda % 8 || q.ldb % 8 || !al16(q.A) || !al16(q.B)) return false;
        if (batch > 1 && (q.sA % 8 || q.sB % 8)) return false;
        P.seg[s].A = (const char*)q.A; P.seg[s].B = (const char*)q.B;
        P.seg[s].lda = q.lda * 2; P.seg[s].ldb = q.ldb * 2; P.seg[s].sA = q.sA * 2; P.seg[s].sB = q.sB * 2;
        P.seg[s].nch = (int)(q.K / CKE);
        nch += P.seg[s].nch;
    }
    if (nch >= (1ll << 24)) return false;
    P.nseg = nseg;
    P.nch = (int)nch;
    P.M = p->M; P.N = p->N;
    P.batch = (int)batch;
    P.sC = p->sC1; P.sR = p->sR1; P.sBias = (bias_per_batch && p->bias) ? p->N : 0;  // comat_gemm_segments: bias [batch, N]; comat_gemm: shared
    P.ep.C = p->C; P.ep.bias = p->bias; P.ep.bias2 = p->bias2; P.ep.R = p->R;
    P.ep.ldc = p->ldc; P.ep.ldr = p->ldr; P.ep.rows_per_b2 = p->rows_per_bias2 > 0 ? p->rows_per_bias2 : 1;
    P.ep.alpha = p->alpha; P.ep.beta = p->beta; P.ep.act = p->act;
    P.ep.out_dt = p->out_dtype; P.ep.r_dt = p->r_dtype;
    // 16-byte epilogue accesses: 8 columns per lane inside a row, every row start 16-byte aligned
    int vec = 1;
    if (p->N % 8 || p->ldc % 8 || !al16(p->C) || p->sC1 % 8) vec = 0;
    if (p->R && (p->ldr % 8 || !al16(p->R) || p->sR1 % 8)) vec = 0;
    if (p->bias && (!al16(p->bias) || P.sBias % 4)) vec = 0;
    if (p->bias2 && !al16(p->bias2)) vec = 0;
    P.vec = vec;
    return true;
}

void set_tiles(Prob3& P, const Cfg3& d) {
    P.tiles_m = (int)cdiv64(P.M, d.bm);
    P.tiles_n = (int)cdiv64(P.N, d.bn);
}

// tile shape: the largest of 64x64 / 64x32 / 32x32 that still yields >= ~3/4 of a workgroup per CU; 8 k-parallel waves when
// the contraction is long enough to give each of them >= 2 chunks
int pick_cfg(const Prob3& P) {
    const int forced = comat_option(COMAT_OPT_G3_CFG);
    if (forced >= 1 && forced <= G3_LAST) return forced;
    const int64_t b = P.batch;
    const int64_t t64 = cdiv64(P.M, 64) * cdiv64(P.N, 64) * b, t6432 = cdiv64(P.M, 64) * cdiv64(P.N, 32) * b;
    const bool deep = P.nch >= 16;
    if (P.M <= 32) return G3_32x32x4;  // the shapes the rule of g3_wants takes (best of the nine on all of them)
    if (t64 >= 192) return deep ? G3_64x64x8 : G3_64x64x4;
    if (t6432 >= 192) return deep ? G3_64x32x8 : G3_64x32x4;
    return deep ? G3_32x32x8 : G3_32x32x4;
}

// which problems the lean kernel takes from the pipelined one (option gemm3: 0 never, 1 this rule, 2 every eligible problem)
bool g3_wants(const Prob3& P) {
    const int opt = comat_option(COMAT_OPT_GEMM3);
    if (opt == 0) return false;
    if (opt >= 2) return true;
    // Measured (profiles/r04_b_mb_gemm3.txt): fragment-shaped register loads lose to the DMA ring almost everywhere - 32
    // cache lines per wave-instruction keep the texture path busy (gemm 512 x 1280 x 1280: 13 us against 9.1) - and win only
    // where a 64-row tile of the pipelined kernel would be three quarters padding AND the contraction is short: the BLIP
    // text decoder at T = 16 (16 x 768 x 768: 5.4 us against 7.2, 16 x 3072 x 768: 5.4 against 7.4; but 16 x 768 x 3072:
    // 11.7 against 9.9).
    return P.M <= 32 && P.nch <= 16 && P.batch == 1;
}

}  // namespace

// -> 5 when the lean kernel took the problem, 0 when it is not eligible / not wanted
int comat_gemm3_try(const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, bool bias_per_batch, void* stream) {
    if (comat_option(COMAT_OPT_GEMM3) == 0) return 0;
    Args3 a = {};
    if (!fill_prob(a.main, p, segs, nseg, bias_per_batch) || !g3_wants(a.main)) return 0;
    const int c = pick_cfg(a.main);
    set_tiles(a.main, cfg3_dims(c));
    const int64_t blocks = (int64_t)a.main.tiles_m * a.main.tiles_n * a.main.batch;
    if (blocks >= (1ll << 31)) return 0;
    launch3(c, a, (unsigned)blocks, (hipStream_t)stream);
    return 5;
}

// Two dependent K-segmented GEMMs (p0 / segs0, then p1 / segs1 whose LAST segment's A operand is p0's output) - in one launch
// when the lean kernel takes both (-> 5), else 0: the caller issues them one after the other.
int comat_gemm3_try_chain(const comat_gemm_params* p0, const comat_gemm_segment* segs0, int nseg0, const comat_gemm_params* p1,
                          const comat_gemm_segment* segs1, int nseg1, void* stream) {
    if (comat_option(COMAT_OPT_GEMM3_CHAIN) == 0) return 0;
    Args3 a = {};
    if (!fill_prob(a.pre, p0, segs0, nseg0, true) || !fill_prob(a.main, p1, segs1, nseg1, true)) return 0;
    if (a.pre.M != a.main.M || !a.pre.vec || a.pre.ep.out_dt != COMAT_BF16 || !p1->ws) return 0;
    if (comat_option(COMAT_OPT_GEMM3_CHAIN) == 1 && !g3_wants(a.main)) return 0;
    // the consumer's last segment must read inside the producer's output (batch items: column blocks of it)
    const Seg3& last = a.main.seg[a.main.nseg - 1];
    if (last.lda != a.pre.ep.ldc * 2 || a.pre.M * a.pre.ep.ldc * 2 >= (1ll << 31)) return 0;
    const int c = pick_cfg(a.main);
    const Cfg3 d = cfg3_dims(c);
    set_tiles(a.main, d);
    set_tiles(a.pre, d);
    if (a.main.tiles_m != a.pre.tiles_m || 2 * (int64_t)a.main.tiles_m > WS_COUNTERS) return 0;
    a.n_pre = a.pre.tiles_m * a.pre.tiles_n * a.pre.batch;
    a.need = a.pre.tiles_n * a.pre.batch;
    a.cons = a.main.tiles_n * a.main.batch;
    a.done = (unsigned*)p1->ws;
    a.seen = (unsigned*)p1->ws + WS_COUNTERS / 2;
    a.err = (unsigned*)p1->ws + WS_COUNTERS - 1;
    const int64_t blocks = (int64_t)a.n_pre + (int64_t)a.main.tiles_m * a.main.tiles_n * a.main.batch;
    if (blocks >= (1ll << 31)) return 0;
    launch3(c, a, (unsigned)blocks, (hipStream_t)stream);
    return 5;
}
