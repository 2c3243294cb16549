// gemm2.hip — LDS-DMA pipelined MFMA GEMM / K-segmented GEMM / implicit-GEMM conv2d for gfx950 (CDNA4): bf16 operands, an
// fp8 (e4m3) variant of the same kernel, and a k-major x k-major variant for the LoRA weight gradients.
//
// The k-contiguous bf16 contractions carry ~all FLOPs of the CoMat step (every frozen Linear / 1x1 conv, the LoRA
// "frozen + low-rank" K-segmented products, every 3x3 conv of the UNets and the VAE).  The general kernel of gemm.hip
// stages operands global -> VGPR -> ds_write -> LDS with ONE 32x32 MFMA tile per wave: two 16-byte LDS fragment reads
// per MFMA (= the whole 256 B/clk LDS read port at full MFMA rate) plus ~80 B/clk of ds_write traffic make it
// LDS-bound at ~10 % of the matrix peak (profiles/r01_pmc_conv_gemm_attnmap.txt).  This kernel is built around what
// bounds it instead:
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip, no
//     ds_write), NST-deep LDS ring (4..8 stages), ONE s_barrier per k-tile, counted s_waitcnt vmcnt(N) so that NST-3
//     k-tiles stay in flight across every barrier (cdna_hip_programming.md section 5: "pipelining across barriers");
//   * each wave owns 2x2 (or 2x1) MFMA tiles of 32x32: 4 fragment reads feed 4 MFMAs (half the LDS read traffic per
//     FLOP), accumulators stay in registers for the whole k-loop;
//   * LDS image = rows of 64 bytes (32 bf16 of k; 128-byte rows with their own swizzle in the KS = 4 shapes), 16-byte
//     chunk c of row r stored at slot c ^ ((r >> 2) & 3): the four
//     16-lane groups of a ds_read_b128 then touch all 64 banks exactly once (conflict-free without padding, which the
//     lane-linear DMA destination would not allow).  The swizzle is applied on the SOURCE address of the DMA and on the
//     fragment read (both sides, rule 21 of the guide);
//   * conv: the im2col gather is the per-lane DMA source address (tap-uniform per k-tile because Cin % 32 == 0),
//     padding taps read a 256-byte zero page;
//   * accumulators are kept TRANSPOSED (MFMA(A = weight fragment, B = activation fragment): lane <-> output row,
//     registers <-> 4 consecutive output columns), one v_permlane32_swap pass makes every lane own 8 consecutive
//     columns: residual / bias loads and the output store are 16-byte accesses (8x fewer store instructions than the
//     2-byte-per-lane stores of the natural layout);
//   * split-K combines inside the launch (gemm_shared.h), no reduce kernel.
// Fragment rule (as gemm.hip): lane (r = lane & 31, h = lane >> 5) of k-step s holds k = 16 s + 8 h .. + 8 of row r for
// both operands, so the k-permutation inside the MFMA cancels.
#include "gemm_shared.h"
#include <type_traits>

// Diagnostic builds only (make diag -> lib/libcomat_hip_d<N>.so, tools/calls/r6_k.sh): the k-loop of gemm2_body with one of its
// parts deleted - 1: no MFMAs (fragments are still read), 2: no fragment reads (MFMAs on stale registers), 3: no LDS-DMA,
// 4: no barrier, 5: no epilogue.  Results are garbage; the times say which part bounds a k-tile.  The product build has G2_DIAG 0.
#ifndef G2_DIAG
#define G2_DIAG 0
#endif
// Issue order of a k-step's fragment reads against the MFMAs of the step before (round 6).  The source asks for "reads of step
// s + 1, then MFMAs of step s"; hipcc's scheduler sinks the reads into the second half of the MFMA batch and then waits for
// lgkmcnt(0) in front of the next batch, so most of the LDS latency is exposed (ISA: `[k0] M5 r4 M1 r2 M1 [k0] ...`; no-DMA build:
// 48 % MFMA busy at 256 x 256).  1: a scheduling barrier behind the reads (all reads first); 2: reads in pairs between the first MFMAs.
#ifndef G2_PIN
#define G2_PIN 0
#endif
// The interleaved k-loop (round 6, bf16): every MFMA of a batch is followed by its share of the NEXT batch's fragment reads and - in the
// batch behind the barrier - of the next k-tile's LDS-DMA pieces, the order pinned by scheduling barriers.  Why: a DMA piece costs
// the issuing wave ~90 cycles of issue; all pieces in a burst behind the barrier (every wave at once) leave the matrix pipe idle for
// ~350 cycles per k-tile (tools/probes/loop_probe.hip, 256 x 256 shape, L2-resident source: 98.8 % MFMA busy without DMA, 73.8 % with
// the burst, 92.4 % with one piece behind every second MFMA).  0 = the round-5 loop.
#ifndef G2_ILV
#define G2_ILV 1
#endif
#ifndef G2_STAGE
#define G2_STAGE 1  // 0: the round-5 epilogue (lane = row) for A/B builds
#endif
#ifndef G2_NST256
#define G2_NST256 4  // ring depth of the 256 x 256 shape (5 = all 160 KiB of LDS)
#endif

namespace {

constexpr int BK = 32;       // bf16 k elements per k-tile
constexpr int RB = 64;       // bytes per LDS row = bytes of k per k-tile (32 bf16, or 64 fp8 e4m3)
constexpr int MAXSEG2 = 8;

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];  // zero-initialised: source of padding taps

struct Seg2 {
    const char* A;
    const char* B;
    int64_t lda, ldb, sA, sB;  // in BYTES (the host multiplies the element strides by the element size)
    int nkt;  // k-tiles of this segment
};

struct Args2 {
    Seg2 seg[MAXSEG2];
    int nseg;
    int Hin, Win, Cin, Hout, Wout, KW, stride, pad, ups;  // conv only (seg[0].A = X, seg[0].B = W, ldb = K)
    int zins;  // conv, ups == 2: the upsampling INSERTS ZEROS (only even coordinates hold samples): the data-gradient of a
               // stride-2 conv as a stride-1 gather over the zero-stuffed output gradient (comat_conv2d mode 1)
    int64_t M, N;
    int nkt;  // k-tiles in total
    int tiles_m, tiles_n, splits;
    int order;  // 0: work items run column-block-fastest inside an XCD chunk (row-block-major), 1: row-block-fastest
                // (column-block-major) - which operand's panels the 8 private L2s share (finish_launch picks per problem)
    int64_t ntiles;
    int64_t sC, sR, sBias;  // batch strides (elements)
    float* ws;
    Epi ep;
    int vec;  // 16-byte epilogue accesses are legal (alignment / divisibility checked by the host)
    const float* scale_a;  // fp8 operands: device pointers to the per-tensor dequantisation scales (NULL = 1)
    const float* scale_b;
    int64_t s_scale_b;     // batch z uses scale_b[z * s_scale_b]
    // fp8 operands only: bf16 k-tail  + A2k B2k^T  added to the SCALED fp8 product (comat_gemm_params::A2k); strides in bytes
    const char* A2k;
    const char* B2k;
    int64_t lda2k, ldb2k, sA2k, sB2k;
    int K2;
    // epi2 == 4 (tail columns, single-segment GEMM only): rows >= N - ep.n2 of B come from B2 (leading dimension seg[0].ldb)
    const char* B2;
    int64_t sB2t, sC2t;  // batch strides of B2 (bytes) and C2 (elements)
};

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// N hand-issued 16-byte LDS reads x[i] <- LDS[addr + i * STRIDE] (immediate offsets); the compiler does not count them: the
// caller waits with its own s_waitcnt lgkmcnt before any use (gemm2_body: frags_landed)
template <int I, int N, int STRIDE> __device__ __forceinline__ void ds_read_frags(short8_t* x, unsigned addr) {
    if constexpr (I < N) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[I]) : "v"(addr), "n"(I * STRIDE));
        ds_read_frags<I + 1, N, STRIDE>(x, addr);
    }
}
// wait until at most `tiles` k-tiles (L DMA instructions each) of this wave are still in flight; tiles in [0, MAXT]
template <int L, int MAXT> __device__ __forceinline__ void wait_tiles(int tiles) {
    if (tiles >= MAXT) wait_vmcnt<L * MAXT>();
    else if constexpr (MAXT > 0) wait_tiles<L, MAXT - 1>(tiles);
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// fp8 (OCP e4m3): one 32x32x64 MFMA consumes the whole 64-byte k-tile; lane (r, h) supplies the two 16-byte chunks it
// would read for bf16 k-steps 0 and 1 (chunks h and 2 + h of row r) as ONE 32-byte operand.  The instruction's own
// lane -> k assignment is the same for A and B, so - as for bf16 - the k-permutation cancels.  Block scales are unused
// (scale 0 selects the unscaled v_mfma_f32_32x32x64_f8f6f4 form); per-tensor scales are applied in the epilogue.
__device__ __forceinline__ void mma_t_fp8(f32x16_t& acc, const short8_t& w0, const short8_t& w1, const short8_t& x0,
                                          const short8_t& x1) {
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i wa = __builtin_bit_cast(v4i, w0), wb = __builtin_bit_cast(v4i, w1);
    const v4i xa = __builtin_bit_cast(v4i, x0), xb = __builtin_bit_cast(v4i, x1);
    const v8i wv = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
    const v8i xv = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, xv, acc, 0 /* A: fp8 e4m3 */, 0 /* B: fp8 e4m3 */, 0, 0, 0, 0);
}

// split-K combine (inside the launch) + fused epilogue of one block tile; shared by the k-contiguous and the k-major kernel
// -> true for the block that ran the tile's epilogue (every block when the problem is not split)
// STG (round 6): the epilogue runs on a ROW-CONTIGUOUS layout.  The MFMA's transposed accumulators give every lane one output row
// and 8 + 8 of its columns: a store instruction then writes 16 bytes into each of 32 different rows - 64 separate requests, a quarter
// of a cache line each (measured: a 4096 x 4096 bf16 output costs 36 us, 0.95 TB/s, against 5 us for the whole launch without its
// stores: profiles/r06_r_kscan_epilogue.txt).  With STG every wave stages 32 rows x WTN columns of fp32 accumulators in its own
// slice of the (now idle) LDS ring and reads them back with lane = (row, 8-column chunk): 8 lanes cover 128 contiguous bytes of a
// row, a store instruction writes 8 (WTN = 64) whole lines, and the residual loads are as contiguous.  The arithmetic per element
// is the same code in the same order (epilogue_run / _tail / _geglu / _geglu_bwd): identical bits.
template <int TM, int TN, int WTM, int WTN, int NTH, bool WT = false, bool STG = false>
__device__ __forceinline__ bool g2_finish(f32x16_t (&acc)[TM][TN], const Args2& g, int sp, int64_t tile, int64_t z, int64_t m0,
                                          int64_t n0, int wr, int wc, int r, int h, int tid, char* smem, bool prescaled = false) {
    // ---- split-K: combine inside the launch (write-through slab stores, sc1 loads by the last arriver: gemm_shared.h) ----
    if constexpr (TM * TN <= 4) if (g.splits > 1) {  // (wave tiles of more than 4 MFMA tiles are never split: the host keeps splits = 1)
        constexpr int QPT = TM * TN * 4;  // 16-byte vectors per thread
        const SlabIO io(g.ws + WS_COUNTERS);
        const int64_t mine = (((int64_t)sp * g.ntiles + tile) * QPT * NTH + tid) * 16;  // byte offset of vector 0
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    io.store(mine + (int64_t)((a * TN + b) * 4 + q) * NTH * 16, v);
                }
        if (!splitk_ticket_is_last((unsigned*)g.ws + tile, g.splits, (unsigned*)smem)) return false;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
        // slabs are added IN SLICE ORDER; the loads of slab s2 + 1 are in flight while slab s2 is added (a write-through
        // slab comes back from memory, ~1.2 us per dependent round trip: 32 slices cost 38 us when read one after the
        // other - profiles/r02_g_bench_shapes.txt, LoRA weight gradients)
        f32x4_t cur[TM][TN][4], nxt[TM][TN][4] = {};
        const int64_t base0 = (tile * QPT * NTH + tid) * 16, sstep = g.ntiles * QPT * NTH * 16;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[a][b][q] = io.load(base0 + (int64_t)((a * TN + b) * 4 + q) * NTH * 16);
        for (int s2 = 0; s2 < g.splits; ++s2) {
            if (s2 + 1 < g.splits) {
                const int64_t src = base0 + (int64_t)(s2 + 1) * sstep;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) nxt[a][b][q] = io.load(src + (int64_t)((a * TN + b) * 4 + q) * NTH * 16);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[a][b][4 * q] += cur[a][b][q][0];
                        acc[a][b][4 * q + 1] += cur[a][b][q][1];
                        acc[a][b][4 * q + 2] += cur[a][b][q][2];
                        acc[a][b][4 * q + 3] += cur[a][b][q][3];
                        cur[a][b][q] = nxt[a][b][q];
                    }
        }
    }

    // ---- epilogue.  acc[a][b][i]: output row m = lane & 31 of the (a)-th 32-row tile, column (i & 3) + 8 (i >> 2) + 4 h
    // of the (b)-th 32-column tile.  Half-swapping quad 0 <-> 1 and 2 <-> 3 gives lane h = 0 columns 0..7 and 16..23,
    // lane h = 1 columns 8..15 and 24..31 ----
#if G2_DIAG == 5  // no epilogue: the accumulators stay alive through a store that never happens
    {
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) sacc += acc[a][b][i];
        if (sacc == 12345.678f) ((float*)g.ep.C)[0] = sacc;
        return true;
    }
#endif
    Epi ep = g.ep;
    if (!prescaled) {  // (the fp8 kernel with a bf16 k-tail has applied the operand scales to its accumulators already)
        if (g.scale_a) ep.alpha *= *g.scale_a;
        if (g.scale_b) ep.alpha *= g.scale_b[z * g.s_scale_b];
    }
    ep.C = (char*)ep.C + z * g.sC * (ep.out_dt == COMAT_F32 ? 4 : 2);
    if (ep.R) ep.R = (const char*)ep.R + z * g.sR * (ep.r_dt == COMAT_F32 ? 4 : 2);
    if (ep.bias) ep.bias += z * g.sBias;
    if (ep.epi2 == 4) ep.C2 = (char*)ep.C2 + z * g.sC2t * (ep.out_dt == COMAT_F32 ? 4 : 2);
    const int64_t nmain = g.N - (ep.epi2 == 4 ? ep.n2 : 0);  // tail columns: [nmain, N) go to C2 (8-column pieces never straddle)
    const bool vec = g.vec != 0;
    float qinv = 0.f, qmax = 0.f;  // GEGLU epilogue that also emits e4m3 bytes (Epi::q8)
    if (ep.q8) qinv = 1.0f / *ep.q_scale;
    if constexpr (STG && G2_STAGE) {
        constexpr int RS = WTN + 4;   // floats per staged row: 16 bytes of padding put the 8 rows of a write phase on different banks
        constexpr int CH = WTN / 8;   // 8-column chunks per row
        constexpr int RPP = 64 / CH;  // rows per pass of the wave
        static_assert(WTN % 8 == 0 && 64 % CH == 0 && 32 % RPP == 0, "staged epilogue: lanes tile 32 rows");
        const int lane = tid & 63;
        float* stg = (float*)smem + (size_t)(tid >> 6) * (32 * RS);
        __syncthreads();  // every wave is done with the ring (fragment reads, the ticket flag)
        // ONE copy of the epilogue arithmetic in the instruction stream: the loops over row blocks and passes are NOT unrolled (the
        // unrolled form was 19 000 instructions, ~150 KB of code that every wave walks through once per tile: more than the
        // instruction cache holds); only the staging writes are per row block (the accumulator array needs a constant index).
#pragma nounroll
        for (int a = 0; a < TM; ++a) {
            static_for<0, TM>([&](auto at) {
                constexpr int A = decltype(at)::value;
                if (a == A) {
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(float4*)(stg + r * RS + b * 32 + 8 * q + 4 * h) =
                                make_float4(acc[A][b][4 * q], acc[A][b][4 * q + 1], acc[A][b][4 * q + 2], acc[A][b][4 * q + 3]);
                }
            });
            const int64_t mrow = m0 + wr * WTM + a * 32, ncol = n0 + wc * WTN;
            if (ep.epi2 == 1 || ep.epi2 == 2) {  // GEGLU: lane = (row, 32-column tile, half j): value chunk j and its gate chunk
                constexpr int LPR = 2 * TN, RPG = 64 / LPR;
#pragma nounroll
                for (int p = 0; p < 32 / RPG; ++p) {
                    const int row = p * RPG + lane / LPR, t = (lane % LPR) >> 1, j = lane & 1;
                    float v[16];
                    const float* src = stg + row * RS + t * 32 + 8 * j;
                    *(float4*)v = *(const float4*)src;
                    *(float4*)(v + 4) = *(const float4*)(src + 4);
                    *(float4*)(v + 8) = *(const float4*)(src + 16);
                    *(float4*)(v + 12) = *(const float4*)(src + 20);
                    const int64_t m = mrow + row, nt = ncol + t * 32;
                    if (m < g.M && nt < g.N) epilogue_geglu(ep, v, m, nt + 8 * j, nt, j, qinv, qmax);
                }
            } else {
#pragma nounroll
                for (int p = 0; p < 32 / RPP; ++p) {
                    const int row = p * RPP + lane / CH, c = lane % CH;
                    float v[8];
                    const float* src = stg + row * RS + c * 8;
                    *(float4*)v = *(const float4*)src;
                    *(float4*)(v + 4) = *(const float4*)(src + 4);
                    const int64_t m = mrow + row, n = ncol + c * 8;
                    if (m < g.M) {
                        if (ep.epi2 == 3) {
                            if (n < g.N) epilogue_geglu_bwd(ep, v, m, n);
                        } else if (ep.epi2 == 4) {
                            if (n < nmain) epilogue_run<WT>(ep, v, m, n, nmain, vec);
                            else if (n < g.N) epilogue_tail(ep, v, m, n - nmain);
                        } else if (n < g.N) {
                            epilogue_run<WT>(ep, v, m, n, g.N, vec);
                        }
                    }
                }
            }
        }
        if (ep.q8) {
            qmax = wave_max(qmax);
            if (lane == 0) fp8_amax_track(ep.q_amax, qmax);
        }
        return true;
    }
    // lane = row form (kernels without the staging area): one copy of the arithmetic too - the loop over the wave's MFMA tiles is
    // not unrolled, the accumulator tile is picked by a uniform branch chain
#pragma nounroll
    for (int ab = 0; ab < TM * TN; ++ab) {
        const int a = ab / TN, b = ab % TN;
        const int64_t m = m0 + wr * WTM + a * 32 + r;
        float v[16];
        static_for<0, TM * TN>([&](auto t) {
            constexpr int T = decltype(t)::value;
            if (ab == T) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = acc[T / TN][T % TN][i];
            }
        });
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            half_swap(v[j], v[4 + j]);
            half_swap(v[8 + j], v[12 + j]);
        }
        const int64_t nt = n0 + wc * WTN + b * 32, nb = nt + 8 * h;
        if (ep.epi2 == 3) {  // GEGLU backward: the lane's two 8-column pieces of dF -> d value / d gate (N % 16 == 0)
            if (m < g.M) {
                if (nb < g.N) epilogue_geglu_bwd(ep, v, m, nb);
                if (nb + 16 < g.N) epilogue_geglu_bwd(ep, v + 8, m, nb + 16);
            }
        } else if (ep.epi2 == 4) {  // tail columns: the first epilogue for columns < nmain, alpha2 * product into C2 behind
            if (m < g.M) {
                if (nb < nmain) epilogue_run<WT>(ep, v, m, nb, nmain, vec);
                else if (nb < g.N) epilogue_tail(ep, v, m, nb - nmain);
                if (nb + 16 < nmain) epilogue_run<WT>(ep, v + 8, m, nb + 16, nmain, vec);
                else if (nb + 16 < g.N) epilogue_tail(ep, v + 8, m, nb + 16 - nmain);
            }
        } else if (ep.epi2) {  // GEGLU over interleaved value / gate columns (N % 32 == 0: whole tiles only)
            if (m < g.M && nt < g.N) epilogue_geglu(ep, v, m, nb, nt, h, qinv, qmax);
        } else if (m < g.M) {
            if (nb < g.N) epilogue_run<WT>(ep, v, m, nb, g.N, vec);
            if (nb + 16 < g.N) epilogue_run<WT>(ep, v + 8, m, nb + 16, g.N, vec);
        }
    }
    return true;
}

// BM x BN block tile, WM x WN waves, each wave (BM/WM) x (BN/WN) as 32x32 MFMA tiles; NST-deep LDS ring.
// CONV: implicit-GEMM gather.
// EB: bytes per operand element (2 = bf16, 1 = fp8 e4m3).
// KS: MFMA k-steps (of 32 bytes) per k-tile: 2 -> rows of 64 bytes (the host's k-tile unit, Args2::nkt), 4 -> rows of 128
// bytes = two host k-tiles per barrier.  The k-loop of a problem with few rows is a latency chain (DMA wait, barrier,
// LDS reads, MFMAs: ~0.11 us per 64-byte k-tile whatever the tile shape - 2 600 of the 4 000 launches of a C2 step);
// doubling the tile halves the number of links.
// `bid` / `nblk`: this block's index in, and the size of, the grid.
template <int BM, int BN, int WM, int WN, int NST, bool CONV, int EB, int KS>
__device__ __forceinline__ void gemm2_body(const Args2& g, const unsigned bid, const unsigned nblk, char* smem) {
    static_assert(KS == 2 || (KS == 4 && EB == 2), "128-byte k-tiles: bf16 only");
    constexpr int RBK = 32 * KS;   // bytes per LDS row
    constexpr int KSF = KS / 2;    // host k-tiles (64 bytes) per kernel k-tile
    constexpr int KE = RBK / EB;   // k elements per k-tile
    constexpr int LPR = RBK / 16;  // lanes (16-byte chunks) per row
    constexpr int RPI = 64 / LPR;  // rows per DMA wave-instruction (1 KiB)
    constexpr int NW = WM * WN, NTH = NW * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int IA = BM / (RPI * NW), IB = BN / (RPI * NW);  // DMA instructions per wave per k-tile, per operand
    static_assert(IA >= 1 && IB >= 1 && BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile too small for the block");
    static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be whole MFMA tiles");
    constexpr int L = IA + IB;           // DMA instructions per wave per k-tile
    constexpr int SS = (BM + BN) * RBK;  // bytes per ring stage: [A: BM rows][B: BN rows]
    static_assert(NST >= 4 && (NST - 3) * L <= 63, "ring depth / vmcnt range");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave / WN, wc = wave % WN;

    // ---- work item: ((z * tiles_m + tm) * tiles_n + tn) * splits + sp  (all wave-uniform: kept in SGPRs) ----
    unsigned lin = (unsigned)xcd_chunk_map(bid, nblk);
    const int sp = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.splits));
    lin /= (unsigned)g.splits;
    int tm, tn;
    int64_t z;
    if (g.order == 2) {  // grouped (option g2_order = 3): row blocks fastest inside bands of 4 row blocks - the 32 tiles an XCD
                         // runs at a time form a 4 x 8 patch (12 operand panels per k-tile instead of 18 for two whole rows)
        const unsigned per_z = (unsigned)g.tiles_m * (unsigned)g.tiles_n;
        z = __builtin_amdgcn_readfirstlane((int)(lin / per_z));
        lin -= (unsigned)z * per_z;
        const unsigned band = lin / (4u * (unsigned)g.tiles_n), first = band * 4u;
        const unsigned rows = (unsigned)g.tiles_m - first < 4u ? (unsigned)g.tiles_m - first : 4u;
        const unsigned in_band = lin - band * 4u * (unsigned)g.tiles_n;
        tm = __builtin_amdgcn_readfirstlane((int)(first + in_band % rows));
        tn = __builtin_amdgcn_readfirstlane((int)(in_band / rows));
    } else if (g.order) {  // row blocks fastest: an XCD's chunk holds ALL row blocks of a few column blocks (weights stay put)
        tm = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_m));
        lin /= (unsigned)g.tiles_m;
        tn = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_n));
        z = __builtin_amdgcn_readfirstlane((int)(lin / (unsigned)g.tiles_n));
    } else {
        tn = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_n));
        lin /= (unsigned)g.tiles_n;
        tm = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_m));
        z = __builtin_amdgcn_readfirstlane((int)(lin / (unsigned)g.tiles_m));
    }
    // the tile's slot in the split-K bookkeeping (slabs, ticket counter): the row-block-major number whatever the order
    const int64_t tile = (z * g.tiles_m + tm) * g.tiles_n + tn;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int nkt_all = g.nkt / KSF;  // k-tiles of THIS kernel's size (the host guarantees divisibility)
    const int per = (nkt_all + g.splits - 1) / g.splits;
    int kt0 = sp * per;
    if (kt0 > nkt_all) kt0 = nkt_all;
    const int kt1 = kt0 + per < nkt_all ? kt0 + per : nkt_all;
    const int nt = __builtin_amdgcn_readfirstlane(kt1 - kt0);

    // ---- DMA source state.  Wave-instruction i of an operand covers rows (i * NW + wave) * RPI .. + RPI of its tile,
    // lane -> (row = lane / LPR, slot = lane % LPR) of the lane-linear 1 KiB it writes; the lane fetches chunk
    // slot ^ swz(row) of that row (the LDS swizzle, applied on the source side).  swz: 64-byte rows (row >> 2) & 3,
    // 128-byte rows (row >> 1) & 7 - either way the 16 rows one ds_read_b128 phase touches hit 64 distinct banks ----
    auto swz = [](int row) { return KS == 2 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
    // (the swizzle is a function of the row INSIDE the tile: instruction i of wave w holds rows (i NW + w) RPI + drow, and
    // NW RPI is a multiple of the swizzle's 16-row period)
    static_assert((NW * RPI) % 16 == 0, "swizzle period");
    const int drow = lane / LPR;
    const int csrc = (lane % LPR) ^ swz(wave * RPI + drow);
    const char* pb[IB];     // running source pointers, B operand
    int64_t brow_[IB];      // clamped global row of the lane's B chunk
    const char* pa[IA];     // GEMM: running source pointers, A operand
    int64_t arow_[IA];      // GEMM: clamped global row
    int by[IA], bx[IA], bb[IA];  // CONV: oy*stride - pad, ox*stride - pad (upsampled coordinates), b * Hin
    bool rv[IA];
    int seg = 0, seg_left = 0;   // issue-side segment cursor
    int ky = 0, kx = 0, ci0 = 0;  // CONV: issue-side tap cursor

#pragma unroll
    for (int i = 0; i < IB; ++i) {
        int64_t gr = n0 + (i * NW + wave) * RPI + drow;
        brow_[i] = gr < g.N ? gr : g.N - 1;  // columns >= N are never stored: any valid row will do
    }
    if (CONV) {
        const int hw = g.Hout * g.Wout;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int64_t gm = m0 + (i * NW + wave) * RPI + drow;
            rv[i] = gm < g.M;
            // (M < 2^31 - the host checks - so the pixel index splits with 32-bit UNSIGNED divisions: a 64-bit signed division by a
            // run-time value is ~150 dependent instructions per lane, twice per DMA row, in front of the kernel's first load)
            const unsigned gmc = rv[i] ? (unsigned)gm : 0u;
            const unsigned b = gmc / (unsigned)hw, rem = gmc - b * (unsigned)hw;
            const unsigned oy = rem / (unsigned)g.Wout, ox = rem - oy * (unsigned)g.Wout;
            bb[i] = (int)b * g.Hin;
            by[i] = (int)oy * g.stride - g.pad;
            bx[i] = (int)ox * g.stride - g.pad;
        }
        const int k0 = kt0 * KE;
        const int tap = k0 / g.Cin;
        ci0 = k0 - tap * g.Cin;
        ky = tap / g.KW;
        kx = tap - ky * g.KW;
#pragma unroll
        for (int i = 0; i < IB; ++i) pb[i] = g.seg[0].B + brow_[i] * g.seg[0].ldb + (int64_t)kt0 * RBK + csrc * 16;
    } else {
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const int64_t gr = m0 + (i * NW + wave) * RPI + drow;
            arow_[i] = gr < g.M ? gr : g.M - 1;
        }
        // locate k-tile kt0 in the segment list
        int t0 = kt0;
        seg = 0;
        while (seg + 1 < g.nseg && t0 >= g.seg[seg].nkt / KSF) {
            t0 -= g.seg[seg].nkt / KSF;
            ++seg;
        }
        seg_left = g.seg[seg].nkt / KSF - t0;
        const Seg2 sg = g.seg[seg];
#pragma unroll
        for (int i = 0; i < IA; ++i) pa[i] = sg.A + z * sg.sA + arow_[i] * sg.lda + (int64_t)t0 * RBK + csrc * 16;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int64_t ntail = g.B2 ? g.N - g.ep.n2 : g.N;  // rows of B from here on live in B2 (tail columns, one segment only)
            const char* bb2 = brow_[i] < ntail ? sg.B + z * sg.sB + brow_[i] * sg.ldb : g.B2 + z * g.sB2t + (brow_[i] - ntail) * sg.ldb;
            pb[i] = bb2 + (int64_t)t0 * RBK + csrc * 16;
        }
    }

    // the DMA of the next k-tile of this block's range into ring stage `st`, as L = IA + IB pieces (1 KiB wave-instructions) that
    // the interleaved k-loop places one by one between its MFMAs: issue_pre (GEMM: segment switch), issue_piece(j, st) for
    // j = 0 .. L - 1 in any order (A pieces first: j < IA), issue_post (conv: tap cursor).  issue(st) = all of it at once (prologue)
    auto issue_pre = [&]() {
        if (G2_DIAG == 3) return;
        if (!CONV) {
            if (seg_left == 0) {  // next segment (uniform branch)
                ++seg;
                const Seg2 sg = g.seg[seg];
                seg_left = sg.nkt / KSF;
#pragma unroll
                for (int i = 0; i < IA; ++i) pa[i] = sg.A + z * sg.sA + arow_[i] * sg.lda + csrc * 16;
#pragma unroll
                for (int i = 0; i < IB; ++i) pb[i] = sg.B + z * sg.sB + brow_[i] * sg.ldb + csrc * 16;
            }
            --seg_left;
        }
    };
    auto issue_piece = [&](const int j, int st) {  // j: compile-time after unrolling
        if (G2_DIAG == 3) return;
        char* sbase = smem + st * SS + wave * 1024;
        if (j < IA) {
            const int i = j;
            if (CONV) {
                const int lim_y = g.Hin * g.ups, lim_x = g.Win * g.ups;
                int sy = by[i] + ky, sx = bx[i] + kx;
                bool ok = rv[i] && (unsigned)sy < (unsigned)lim_y && (unsigned)sx < (unsigned)lim_x;
                if (g.zins) ok = ok && !((sy | sx) & 1);
                if (g.ups == 2) {
                    sy >>= 1;
                    sx >>= 1;
                }
                const int off = (((bb[i] + sy) * g.Win + sx) * g.Cin + ci0) * EB + csrc * 16;  // bytes (host: < 2^31)
                const void* src = ok ? (const void*)(g.seg[0].A + off) : (const void*)(g_zero_page + (lane & 15) * 16);
                dma16(src, sbase + i * NW * 1024);
            } else {
                dma16(pa[i], sbase + i * NW * 1024);
                pa[i] += RBK;
            }
        } else {
            const int i = j - IA;
            dma16(pb[i], sbase + BM * RBK + i * NW * 1024);
            pb[i] += RBK;
        }
    };
    auto issue_post = [&]() {
        if (G2_DIAG == 3) return;
        if (CONV) {
            ci0 += KE;
            if (ci0 >= g.Cin) {
                ci0 = 0;
                if (++kx == g.KW) {
                    kx = 0;
                    ++ky;
                }
            }
        }
    };
    auto issue = [&](int st) {
        issue_pre();
#pragma unroll
        for (int j = 0; j < L; ++j) issue_piece(j, st);
        issue_post();
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    // fragment addressing: lane (r, h), k-step s reads 16 bytes at row r, slot (2 s + h) ^ swz(r)
    const int sw = swz(r);
    auto fofs = [&](int s2) { return r * RBK + (((2 * s2 + h) ^ sw) * 16); };
    const int fo0 = fofs(0), fo1 = fofs(1);
    const int a_base = wr * WTM * RBK, b_base = BM * RBK + wc * WTN * RBK;

    // ---- pipeline.  DMA runs NST-1 k-tiles ahead of the MFMAs; ONE barrier per k-tile, in the MIDDLE of a tile's MFMA
    // work: the fragments of k-step 1 are requested from LDS before the MFMAs of k-step 0 issue, the fragments of the
    // next tile's k-step 0 before the MFMAs of k-step 1, so that with one wave per SIMD (these problems rarely have two
    // blocks per CU) the LDS latency and the barrier hide under 4 MFMAs instead of stalling the in-order wave.
    // Iteration t, at its barrier: tiles .. t+NST-2 have been issued, tile t+1 must have landed -> up to NST-3 tiles stay
    // in flight across the barrier (counted vmcnt); then tile t+NST-1 is issued into the stage of tile t-1, whose reads
    // were all consumed by MFMAs before any wave reached this barrier. ----
    auto frags = [&](const char* st, int fo, short8_t (&xf)[TM], short8_t (&wf)[TN]) {
        if (G2_DIAG == 2) return;
        if constexpr (G2_PIN == 3 && EB == 2) {
            // hand-issued reads (the compiler neither counts nor moves them): the batch's TM + TN ds_read_b128 go out HERE, and
            // frags_landed<NR>() in front of the MFMAs that use an EARLIER batch waits for lgkmcnt(NR) - everything but this batch
            const unsigned la = (unsigned)(uintptr_t)(st + a_base + fo), lb = (unsigned)(uintptr_t)(st + b_base + fo);
            ds_read_frags<0, TM, 32 * RBK>(xf, la);
            ds_read_frags<0, TN, 32 * RBK>(wf, lb);
            return;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) xf[a] = *(const short8_t*)(st + a_base + a * 32 * RBK + fo);
#pragma unroll
        for (int b = 0; b < TN; ++b) wf[b] = *(const short8_t*)(st + b_base + b * 32 * RBK + fo);
        if (G2_PIN == 1) __builtin_amdgcn_sched_barrier(0);
    };
    // hand-issued reads only: at most `younger` fragment reads (the batches issued after the one about to be used) may still be
    // in flight; the empty asms tie the batch's registers to the wait, so that no MFMA reading them can be scheduled above it
    auto frags_landed = [&](auto younger, short8_t (&xf)[TM], short8_t (&wf)[TN]) {
        if constexpr (G2_PIN == 3 && EB == 2 && G2_DIAG != 2) {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(younger)::value) : "memory");
#pragma unroll
            for (int a = 0; a < TM; ++a) asm volatile("" : "+v"(xf[a]));
#pragma unroll
            for (int b = 0; b < TN; ++b) asm volatile("" : "+v"(wf[b]));
        }
    };
    constexpr std::integral_constant<int, TM + TN> one_batch{};
    constexpr std::integral_constant<int, 0> no_batch{};
    auto mmas = [&](const short8_t (&xf)[TM], const short8_t (&wf)[TN]) {
#if G2_DIAG == 1
#pragma unroll
        for (int a = 0; a < TM; ++a) asm volatile("" ::"v"(xf[a]));
#pragma unroll
        for (int b = 0; b < TN; ++b) asm volatile("" ::"v"(wf[b]));
        return;
#endif
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) mma_t(acc[a][b], wf[b], xf[a]);
#if G2_PIN == 2
        // the batch's MFMAs and the TM + TN fragment reads issued with it: MFMA, two reads, MFMA, two reads, .., the rest of the MFMAs
#pragma unroll
        for (int i = 0; i < (TM + TN + 1) / 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN + 1) / 2, 0);
#endif
    };
    auto mmas8 = [&](const short8_t (&x0)[TM], const short8_t (&w0)[TN], const short8_t (&x1)[TM], const short8_t (&w1)[TN]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) mma_t_fp8(acc[a][b], w0[b], w1[b], x0[a], x1[a]);
    };
    // interleaved batch: the TM x TN MFMAs on (xc, wc); behind MFMA m: reads [m RP, (m + 1) RP) of the next batch's TM + TN fragment
    // reads into (xn, wn) from `stn + fon` (has_next), and - with_dma - pieces of the next k-tile's DMA into ring slot `slot`
    // (do_issue: a tile is left to issue; wave-uniform)
    auto batch = [&](auto with_dma, short8_t (&xc)[TM], short8_t (&wc)[TN], short8_t (&xn)[TM], short8_t (&wn)[TN], const char* stn,
                     int fon, bool has_next, bool do_issue, int slot) {
        constexpr int NM = TM * TN, NR = TM + TN;
        constexpr int RP = (NR + NM - 1) / NM;                // reads behind each MFMA
        constexpr int DP = L >= NM ? (L + NM - 1) / NM : 1;   // DMA pieces behind an MFMA that carries any
        constexpr int DS = L >= NM ? 1 : NM / L;              // ... which is every DS-th MFMA
        constexpr bool DMA = decltype(with_dma)::value;
        if (DMA && do_issue) issue_pre();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#if G2_DIAG == 1
            asm volatile("" ::"v"(xc[m / TN]), "v"(wc[m % TN]));
#else
            mma_t(acc[m / TN][m % TN], wc[m % TN], xc[m / TN]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (G2_DIAG != 2 && has_next) {
#pragma unroll
                for (int q = m * RP; q < (m + 1) * RP && q < NR; ++q) {
                    if (q < TM) xn[q] = *(const short8_t*)(stn + a_base + q * 32 * RBK + fon);
                    else wn[q - TM] = *(const short8_t*)(stn + b_base + (q - TM) * 32 * RBK + fon);
                }
            }
            if (DMA && (m % DS) == DS - 1 && do_issue) {
#pragma unroll
                for (int j = (m / DS) * DP; j < (m / DS + 1) * DP && j < L; ++j) issue_piece(j, slot);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA && do_issue) issue_post();
        static_assert((NM / DS) * DP >= L, "every DMA piece has a place");
    };
    constexpr std::true_type dma_here{};
    constexpr std::false_type no_dma{};
#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
        if (u < nt) issue(u);
    short8_t xf0[TM], wf0[TN], xf1[TM], wf1[TN];
#if G2_DIAG == 2
    for (int a = 0; a < TM; ++a) xf0[a] = xf1[a] = short8_t{(short)lane, 1, 2, 3, 4, 5, 6, 7};
    for (int b = 0; b < TN; ++b) wf0[b] = wf1[b] = short8_t{(short)lane, 7, 6, 5, 4, 3, 2, 1};
#endif
    int stage = 0;  // ring slot of tile t
    if (nt > 0) {   // (ONE region from the first fragment read to the last MFMA: tools/isa_frag_check.py follows every path of it)
        wait_tiles<L, NST - 2>(nt - 1);  // tile 0 landed (this wave's part)
        __builtin_amdgcn_s_barrier();   // ... and every other wave's
        asm volatile("" ::: "memory");
        frags(smem, fo0, xf0, wf0);
    if constexpr (EB == 2 && KS == 4 && G2_ILV == 1) {
        for (int t = 0; t + 1 < nt; ++t) {
            const char* cur = smem + stage * SS;
            batch(no_dma, xf0, wf0, xf1, wf1, cur, fo1, true, false, 0);
            batch(no_dma, xf1, wf1, xf0, wf0, cur, fofs(2), true, false, 0);
            batch(no_dma, xf0, wf0, xf1, wf1, cur, fofs(3), true, false, 0);
            const int nstage = stage + 1 == NST ? 0 : stage + 1;
            wait_tiles<L, NST - 3>(nt - 2 - t);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            batch(dma_here, xf1, wf1, xf0, wf0, smem + nstage * SS, fo0, true, t + NST - 1 < nt, stage == 0 ? NST - 1 : stage - 1);
            stage = nstage;
        }
        {
            const char* cur = smem + stage * SS;
            batch(no_dma, xf0, wf0, xf1, wf1, cur, fo1, true, false, 0);
            batch(no_dma, xf1, wf1, xf0, wf0, cur, fofs(2), true, false, 0);
            batch(no_dma, xf0, wf0, xf1, wf1, cur, fofs(3), true, false, 0);
            batch(no_dma, xf1, wf1, xf0, wf0, smem, fo0, false, false, 0);
        }
    } else if constexpr (EB == 2 && KS == 4) {
        // four k-steps per tile: fragments of step s + 1 are requested before the MFMAs of step s issue; the barrier (and
        // the DMA issue behind it) sits before the MFMAs of the LAST step, with the next tile's step 0 already requested
        for (int t = 0; t + 1 < nt; ++t) {
            const char* cur = smem + stage * SS;
            frags(cur, fo1, xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            frags(cur, fofs(2), xf0, wf0);
            frags_landed(one_batch, xf1, wf1);
            mmas(xf1, wf1);
            frags(cur, fofs(3), xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            const int nstage = stage + 1 == NST ? 0 : stage + 1;
            wait_tiles<L, NST - 3>(nt - 2 - t);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (G2_PIN == 3) frags(smem + nstage * SS, fo0, xf0, wf0);  // reads first: the DMA's address arithmetic covers their latency
            if (t + NST - 1 < nt) issue(stage == 0 ? NST - 1 : stage - 1);
            if (G2_PIN != 3) frags(smem + nstage * SS, fo0, xf0, wf0);
            frags_landed(one_batch, xf1, wf1);
            mmas(xf1, wf1);
            stage = nstage;
        }
        {
            const char* cur = smem + stage * SS;
            frags(cur, fo1, xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            frags(cur, fofs(2), xf0, wf0);
            frags_landed(one_batch, xf1, wf1);
            mmas(xf1, wf1);
            frags(cur, fofs(3), xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            frags_landed(no_batch, xf1, wf1);
            mmas(xf1, wf1);
        }
    } else if constexpr (EB == 2 && G2_ILV == 1) {
        for (int t = 0; t + 1 < nt; ++t) {
            const char* cur = smem + stage * SS;
            batch(no_dma, xf0, wf0, xf1, wf1, cur, fo1, true, false, 0);
            const int nstage = stage + 1 == NST ? 0 : stage + 1;
            wait_tiles<L, NST - 3>(nt - 2 - t);
            if (G2_DIAG != 4) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            batch(dma_here, xf1, wf1, xf0, wf0, smem + nstage * SS, fo0, true, t + NST - 1 < nt, stage == 0 ? NST - 1 : stage - 1);
            stage = nstage;
        }
        {  // last tile
            batch(no_dma, xf0, wf0, xf1, wf1, smem + stage * SS, fo1, true, false, 0);
            batch(no_dma, xf1, wf1, xf0, wf0, smem, fo0, false, false, 0);
        }
    } else if constexpr (EB == 2) {
        for (int t = 0; t + 1 < nt; ++t) {  // steady state: a next tile exists (no data-dependent branch around the LDS reads)
            frags(smem + stage * SS, fo1, xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            const int nstage = stage + 1 == NST ? 0 : stage + 1;
            wait_tiles<L, NST - 3>(nt - 2 - t);
            if (G2_DIAG != 4) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (G2_PIN == 3) frags(smem + nstage * SS, fo0, xf0, wf0);  // reads first: the DMA's address arithmetic covers their latency
            if (t + NST - 1 < nt) issue(stage == 0 ? NST - 1 : stage - 1);  // the slot of tile t-1
            if (G2_PIN != 3) frags(smem + nstage * SS, fo0, xf0, wf0);
            frags_landed(one_batch, xf1, wf1);
            mmas(xf1, wf1);
            stage = nstage;
        }
        {  // last tile
            frags(smem + stage * SS, fo1, xf1, wf1);
            frags_landed(one_batch, xf0, wf0);
            mmas(xf0, wf0);
            frags_landed(no_batch, xf1, wf1);
            mmas(xf1, wf1);
        }
    } else {
        // fp8: one 32x32x64 MFMA per tile pair and k-tile.  Same ring protocol; the second half of tile t's fragments and
        // the first half of tile t+1's are requested before the MFMAs of tile t issue.
        short8_t xn[TM], wn[TN];
        for (int t = 0; t + 1 < nt; ++t) {
            frags(smem + stage * SS, fo1, xf1, wf1);
            const int nstage = stage + 1 == NST ? 0 : stage + 1;
            wait_tiles<L, NST - 3>(nt - 2 - t);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + NST - 1 < nt) issue(stage == 0 ? NST - 1 : stage - 1);
            frags(smem + nstage * SS, fo0, xn, wn);
            mmas8(xf0, wf0, xf1, wf1);
#pragma unroll
            for (int a = 0; a < TM; ++a) xf0[a] = xn[a];
#pragma unroll
            for (int b = 0; b < TN; ++b) wf0[b] = wn[b];
            stage = nstage;
        }
        {
            frags(smem + stage * SS, fo1, xf1, wf1);
            mmas8(xf0, wf0, xf1, wf1);
        }
    }
    }

    // fp8 operands + bf16 k-tail (the LoRA up projection of a frozen projection, K2 = rank): the e4m3 product is scaled in its
    // registers, then K2 / 16 more MFMA steps run on fragments read STRAIGHT from global memory (the tail is 128 .. 256 bytes per
    // row: no ring, one exposed round trip) - by the first k-slice only when the problem is split
    bool prescaled = false;
    if constexpr (EB == 1) {
        if (g.A2k) {
            prescaled = true;
            float sc = 1.0f;
            if (g.scale_a) sc *= *g.scale_a;
            if (g.scale_b) sc *= g.scale_b[z * g.s_scale_b];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] *= sc;
            if (sp == 0) {
                const char* pa2[TM];
                const char* pb2[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    int64_t row = m0 + wr * WTM + a * 32 + r;
                    if (row >= g.M) row = g.M - 1;  // rows beyond M are never stored
                    pa2[a] = g.A2k + z * g.sA2k + row * g.lda2k + h * 16;
                }
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    int64_t row = n0 + wc * WTN + b * 32 + r;
                    if (row >= g.N) row = g.N - 1;
                    pb2[b] = g.B2k + z * g.sB2k + row * g.ldb2k + h * 16;
                }
                const int steps = g.K2 >> 4;
#pragma unroll 2
                for (int s2 = 0; s2 < steps; ++s2) {
                    short8_t xt[TM], wt[TN];
#pragma unroll
                    for (int a = 0; a < TM; ++a) xt[a] = *(const short8_t*)(pa2[a] + s2 * 32);
#pragma unroll
                    for (int b = 0; b < TN; ++b) wt[b] = *(const short8_t*)(pb2[b] + s2 * 32);
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) mma_t(acc[a][b], wt[b], xt[a]);
                }
            }
        }
    }
    static_assert(NW * 32 * (WTN + 4) * 4 <= NST * SS, "the staged epilogue fits the ring");
    g2_finish<TM, TN, WTM, WTN, NTH, false, true>(acc, g, sp, tile, z, m0, n0, wr, wc, r, h, tid, smem, prescaled);
}

template <int BM, int BN, int WM, int WN, int NST, bool CONV, int EB = 2, int KS = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm2_kernel(Args2 g) {
    __shared__ __attribute__((aligned(1024))) char smem[NST * (BM + BN) * 32 * KS];  // the ONLY LDS object of the kernel
    gemm2_body<BM, BN, WM, WN, NST, CONV, EB, KS>(g, blockIdx.x, gridDim.x, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// k-major operands: C[M, N] (+)= A^T B with A stored [K, M] and B stored [K, N] (rows = k).  This is every LoRA weight
// gradient of the step: dU = g^T h and dD = u^T x contract over the TOKEN axis of two row-major token matrices
// (training_utils/pipeline.py:94-115 leaves them to autograd's addmm).  The general kernel gathers such fragments
// with eight 2-byte LDS reads each; here the k-tile is DMA'd as it lies in memory ([32 k-rows][128 columns], 256-byte
// rows) and the MFMA fragments come out of LDS through the hardware transpose read ds_read_b64_tr_b16.
// Its semantics, probed on gfx950 (tools/probes/tr_read_probe.hip, profiles/r02_e_tr_probe.txt): inside every group
// of 16 lanes, lane 4k + q supplies the address of 4 consecutive 16-bit elements = columns 4q .. 4q+3 of row k of a
// 4 x 16 matrix, and lane i receives column i (rows 0 .. 3).  With "row" = k and "column" = the operand's row index,
// two reads hand lane (r, h) its 8 consecutive k of operand row r.
// Bank layout: 16-byte slot p of k-row kk is stored at slot p ^ (4 * (kk & 3)) (source-side swizzle of the DMA, same
// XOR on the read): the 8 x 32-byte pieces that one 32-lane pass of the read touches then cover all 64 banks once.
// One tile shape (128 x 128, 4 waves, 4-deep ring); few output tiles and K = thousands of tokens, so the launch is
// always split along k and combined in-launch (g2_finish).
// ---------------------------------------------------------------------------------------------------------------
struct TTFrag {
    unsigned long long lo, hi;
};

template <int NST> __global__ __launch_bounds__(256) void gemm2_tt_kernel(Args2 g) {
    constexpr int BM = 128, BN = 128, NW = 4, NTH = 256, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int RBT = 256;                    // bytes per k-row of an operand image (128 columns)
    constexpr int OPB = BK * RBT;               // 8 KiB per operand per stage
    constexpr int SS = 2 * OPB;
    constexpr int IO = BK / (4 * NW);           // DMA instructions per wave per operand per k-tile (4 k-rows each)
    constexpr int L = 2 * IO;
    static_assert(IO == 2 && (NST - 3) * L <= 63, "k-major tile geometry");
    __shared__ __attribute__((aligned(1024))) char smem[NST * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    unsigned lin = (unsigned)xcd_chunk_map(blockIdx.x, gridDim.x);
    const int sp = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.splits));
    lin /= (unsigned)g.splits;
    const int64_t tile = __builtin_amdgcn_readfirstlane((int)lin);
    const int tn = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_n));
    lin /= (unsigned)g.tiles_n;
    const int tm = __builtin_amdgcn_readfirstlane((int)(lin % (unsigned)g.tiles_m));
    const int64_t z = __builtin_amdgcn_readfirstlane((int)(lin / (unsigned)g.tiles_m));
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int per = (g.nkt + g.splits - 1) / g.splits;
    int kt0 = sp * per;
    if (kt0 > g.nkt) kt0 = g.nkt;
    const int kt1 = kt0 + per < g.nkt ? kt0 + per : g.nkt;
    const int nt = __builtin_amdgcn_readfirstlane(kt1 - kt0);

    // DMA source: lane -> k-row kr = lane / 16 of the instruction's 4 rows, 16-byte slot p = lane % 16 of the 256-byte
    // row, fetching source chunk p ^ (4 kr).  Columns beyond the operand are clamped (their outputs are never stored).
    const Seg2 sg = g.seg[0];
    const int kr = lane >> 4, chunk = (lane & 15) ^ (4 * kr);
    int64_t ca = m0 + chunk * 8, cb = n0 + chunk * 8;
    if (ca > g.M - 8) ca = g.M - 8;
    if (cb > g.N - 8) cb = g.N - 8;
    const char* pa[IO];
    const char* pb[IO];
#pragma unroll
    for (int i = 0; i < IO; ++i) {
        const int64_t krow = (int64_t)kt0 * BK + (i * NW + wave) * 4 + kr;
        pa[i] = sg.A + z * sg.sA + krow * sg.lda + ca * 2;
        pb[i] = sg.B + z * sg.sB + krow * sg.ldb + cb * 2;
    }
    const int64_t stepa = (int64_t)BK * sg.lda, stepb = (int64_t)BK * sg.ldb;
    auto issue = [&](int st) {
        char* sbase = smem + st * SS + wave * 1024;
#pragma unroll
        for (int i = 0; i < IO; ++i) {
            dma16(pa[i], sbase + i * NW * 1024);
            pa[i] += stepa;
        }
#pragma unroll
        for (int i = 0; i < IO; ++i) {
            dma16(pb[i], sbase + OPB + i * NW * 1024);
            pb[i] += stepb;
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    // transpose-read addressing.  16-lane group gq = lane >> 4: columns 16 (gq & 1) .. + 16 of the 32-row MFMA tile,
    // k half 8 (gq >> 1); inside the group lane 4 kq + q points at row k0 + kq, columns 4 q .. 4 q + 3.  Byte offset in
    // the row: (column * 2) ^ (64 * kq) (the slot swizzle; tile bases are multiples of 64 bytes, kq = row & 3).
    const int kq = (lane & 15) >> 2;
    const int colb = ((wr * WTM + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const int colb_b = ((wc * WTN + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned rowoff = (unsigned)((8 * h + kq) * RBT);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned fa[TM], fb[TN];  // byte offsets of the (k-step 0, first read) fragment piece of each MFMA tile
#pragma unroll
    for (int a = 0; a < TM; ++a) fa[a] = rowoff + (unsigned)((colb + a * 64) ^ (64 * kq));
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[b] = (unsigned)OPB + rowoff + (unsigned)((colb_b + b * 64) ^ (64 * kq));

#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
        if (u < nt) issue(u);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        wait_tiles<L, NST - 2>(nt - 1 - t);
        __builtin_amdgcn_s_barrier();  // tile t landed everywhere; every wave is done reading tile t-1
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) issue(stage == 0 ? NST - 1 : stage - 1);
        const unsigned sb = smem_base + (unsigned)(stage * SS);
        // all 16 transpose reads of the k-tile (2 k-steps x (2 + 2) fragments x 2 reads) in one statement: the compiler
        // does not track asm loads, so the wait sits in the same statement (cdna_hip_programming.md 5.7, form i)
        TTFrag xa[2][TM], xb[2][TN];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:1024\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:1024\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:1024\n\t"
            "ds_read_b64_tr_b16 %8, %16 offset:4096\n\tds_read_b64_tr_b16 %9, %16 offset:5120\n\t"
            "ds_read_b64_tr_b16 %10, %17 offset:4096\n\tds_read_b64_tr_b16 %11, %17 offset:5120\n\t"
            "ds_read_b64_tr_b16 %12, %18 offset:4096\n\tds_read_b64_tr_b16 %13, %18 offset:5120\n\t"
            "ds_read_b64_tr_b16 %14, %19 offset:4096\n\tds_read_b64_tr_b16 %15, %19 offset:5120\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(xa[0][0].lo), "=&v"(xa[0][0].hi), "=&v"(xa[0][1].lo), "=&v"(xa[0][1].hi), "=&v"(xb[0][0].lo),
              "=&v"(xb[0][0].hi), "=&v"(xb[0][1].lo), "=&v"(xb[0][1].hi), "=&v"(xa[1][0].lo), "=&v"(xa[1][0].hi),
              "=&v"(xa[1][1].lo), "=&v"(xa[1][1].hi), "=&v"(xb[1][0].lo), "=&v"(xb[1][0].hi), "=&v"(xb[1][1].lo),
              "=&v"(xb[1][1].hi)
            : "v"(sb + fa[0]), "v"(sb + fa[1]), "v"(sb + fb[0]), "v"(sb + fb[1])
            : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    short8_t xf, wf;
                    __builtin_memcpy(&xf, &xa[s2][a], 16);
                    __builtin_memcpy(&wf, &xb[s2][b], 16);
                    mma_t(acc[a][b], wf, xf);
                }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    g2_finish<TM, TN, WTM, WTN, NTH>(acc, g, sp, tile, z, m0, n0, wr, wc, r, h, tid, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// GROUPED k-major products: up to TT_MAXP independent problems  C_p[M_p, N_p] += A_p^T B_p  (fp32 accumulate in place)
// in ONE launch.  The LoRA weight gradients of a step are ~720 such problems (320 x 128 outputs over 8 192 tokens and
// the like), each far too small to fill the chip and each paying the serial split-K combine of its own launch
// (profiles/r02_h_mb_tt.txt: 24 us per launch, 30 TFLOP/s).  They depend on nothing but saved activations and nobody
// reads them before the optimizer, so the host queues them (comat_amd/ops.py: TTQueue) and hands them over in groups:
// the launch then has thousands of workgroups, the split-K combines of different tiles overlap, and the per-launch
// latencies are paid once per group.
// The problem table travels in the KERNEL ARGUMENTS (72 bytes per problem, < 4 KiB in all): no device-side table to
// upload, and a captured hipGraph node carries it by value.
// Same tile machinery as gemm2_tt_kernel (128 x 128 tile, 4 waves, DMA ring, ds_read_b64_tr_b16 fragments); new here:
// any K >= 1 (k-rows beyond K are DMA'd from the zero page, so the 77-token text projections join the group), and a
// per-problem split count chosen by the host for the GROUP (long slices when the group fills the chip anyway).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TT_MAXP = 48;

struct TTProb {
    const char* A;
    const char* B;
    float* C;
    int M, N, K;
    int lda, ldb;  // bytes
    int ldc;       // elements
    int tiles_n, tiles, splits;
    int tile0, slab0;  // first ticket counter / first 64 KiB slab of this problem in the workspace
};
static_assert(sizeof(TTProb) == 72, "TTProb layout");

struct TTGroupArgs {
    TTProb prob[TT_MAXP];
    int blk0[TT_MAXP + 1];  // first work item of every problem (blk0[nprob] = number of work items)
    int nprob;
    float* ws;
};
static_assert(sizeof(TTGroupArgs) <= 4096, "kernel arguments are limited to 4 KiB");

template <int NST> __global__ __launch_bounds__(256, 2) void gemm2_tt_group_kernel(TTGroupArgs g) {
    constexpr int BM = 128, BN = 128, NW = 4, NTH = 256, WTM = 64, WTN = 64, TM = 2, TN = 2;
    constexpr int RBT = 256, OPB = BK * RBT, SS = 2 * OPB, IO = BK / (4 * NW), L = 2 * IO;
    static_assert(IO == 2 && (NST - 3) * L <= 63, "k-major tile geometry");
    __shared__ __attribute__((aligned(1024))) char smem[NST * SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // work item -> problem (binary search over the kernel-argument table: scalar loads only)
    const int lin = __builtin_amdgcn_readfirstlane((int)xcd_chunk_map(blockIdx.x, gridDim.x));
    int lo = 0, hi = g.nprob;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (lin >= g.blk0[mid]) lo = mid;
        else hi = mid;
    }
    const TTProb& P = g.prob[lo];
    const int local = lin - g.blk0[lo];
    // tile-fastest: neighbouring work items (same XCD chunk) contract the same k-slice of one shared operand
    const int tile = __builtin_amdgcn_readfirstlane(local % P.tiles);
    const int sp = __builtin_amdgcn_readfirstlane(local / P.tiles);
    const int tn = tile % P.tiles_n, tm = tile / P.tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int K = P.K, M = P.M, N = P.N, splits = P.splits;
    const int nkt = (K + BK - 1) / BK;
    const int per = (nkt + splits - 1) / splits;
    int kt0 = sp * per;
    if (kt0 > nkt) kt0 = nkt;
    const int kt1 = kt0 + per < nkt ? kt0 + per : nkt;
    const int nt = __builtin_amdgcn_readfirstlane(kt1 - kt0);

    const int kr = lane >> 4, chunk = (lane & 15) ^ (4 * kr);
    int64_t ca = m0 + chunk * 8, cb = n0 + chunk * 8;
    if (ca > M - 8) ca = M - 8;
    if (cb > N - 8) cb = N - 8;
    const char* abase = P.A + ca * 2;
    const char* bbase = P.B + cb * 2;
    const int64_t lda = P.lda, ldb = P.ldb;
    const char* zsrc = (const char*)g_zero_page + (lane & 15) * 16;
    int krow0 = kt0 * BK + wave * 4 + kr;  // k-row of DMA instruction 0 of the next tile to issue (instruction i: + 16 i)
    const char* pa[IO];
    const char* pb[IO];
#pragma unroll
    for (int i = 0; i < IO; ++i) {
        pa[i] = abase + (int64_t)(krow0 + i * NW * 4) * lda;
        pb[i] = bbase + (int64_t)(krow0 + i * NW * 4) * ldb;
    }
    const int64_t stepa = (int64_t)BK * lda, stepb = (int64_t)BK * ldb;
    auto issue = [&](int st) {
        char* sbase = smem + st * SS + wave * 1024;
#pragma unroll
        for (int i = 0; i < IO; ++i) {
            dma16(krow0 + i * NW * 4 < K ? (const void*)pa[i] : (const void*)zsrc, sbase + i * NW * 1024);
            pa[i] += stepa;
        }
#pragma unroll
        for (int i = 0; i < IO; ++i) {
            dma16(krow0 + i * NW * 4 < K ? (const void*)pb[i] : (const void*)zsrc, sbase + OPB + i * NW * 1024);
            pb[i] += stepb;
        }
        krow0 += BK;
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    const int kq = (lane & 15) >> 2;
    const int colb = ((wr * WTM + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const int colb_b = ((wc * WTN + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
    const unsigned rowoff = (unsigned)((8 * h + kq) * RBT);
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned fa[TM], fb[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) fa[a] = rowoff + (unsigned)((colb + a * 64) ^ (64 * kq));
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[b] = (unsigned)OPB + rowoff + (unsigned)((colb_b + b * 64) ^ (64 * kq));

#pragma unroll
    for (int u = 0; u < NST - 1; ++u)
        if (u < nt) issue(u);
    int stage = 0;
    for (int t = 0; t < nt; ++t) {
        wait_tiles<L, NST - 2>(nt - 1 - t);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NST - 1 < nt) issue(stage == 0 ? NST - 1 : stage - 1);
        const unsigned sb = smem_base + (unsigned)(stage * SS);
        TTFrag xa[2][TM], xb[2][TN];
        asm volatile(
            "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:1024\n\t"
            "ds_read_b64_tr_b16 %2, %17\n\tds_read_b64_tr_b16 %3, %17 offset:1024\n\t"
            "ds_read_b64_tr_b16 %4, %18\n\tds_read_b64_tr_b16 %5, %18 offset:1024\n\t"
            "ds_read_b64_tr_b16 %6, %19\n\tds_read_b64_tr_b16 %7, %19 offset:1024\n\t"
            "ds_read_b64_tr_b16 %8, %16 offset:4096\n\tds_read_b64_tr_b16 %9, %16 offset:5120\n\t"
            "ds_read_b64_tr_b16 %10, %17 offset:4096\n\tds_read_b64_tr_b16 %11, %17 offset:5120\n\t"
            "ds_read_b64_tr_b16 %12, %18 offset:4096\n\tds_read_b64_tr_b16 %13, %18 offset:5120\n\t"
            "ds_read_b64_tr_b16 %14, %19 offset:4096\n\tds_read_b64_tr_b16 %15, %19 offset:5120\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(xa[0][0].lo), "=&v"(xa[0][0].hi), "=&v"(xa[0][1].lo), "=&v"(xa[0][1].hi), "=&v"(xb[0][0].lo),
              "=&v"(xb[0][0].hi), "=&v"(xb[0][1].lo), "=&v"(xb[0][1].hi), "=&v"(xa[1][0].lo), "=&v"(xa[1][0].hi),
              "=&v"(xa[1][1].lo), "=&v"(xa[1][1].hi), "=&v"(xb[1][0].lo), "=&v"(xb[1][0].hi), "=&v"(xb[1][1].lo),
              "=&v"(xb[1][1].hi)
            : "v"(sb + fa[0]), "v"(sb + fa[1]), "v"(sb + fb[0]), "v"(sb + fb[1])
            : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    short8_t xf, wf;
                    __builtin_memcpy(&xf, &xa[s2][a], 16);
                    __builtin_memcpy(&wf, &xb[s2][b], 16);
                    mma_t(acc[a][b], wf, xf);
                }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }

    // split-K combine of this problem's tile (same protocol as g2_finish; slabs of 64 KiB = one 128 x 128 fp32 tile)
    if (splits > 1) {
        constexpr int QPT = TM * TN * 4;
        const SlabIO io(g.ws + WS_COUNTERS);
        const int64_t sstep = (int64_t)P.tiles * QPT * NTH * 16;
        const int64_t base0 = ((int64_t)(P.slab0 + tile) * QPT * NTH + tid) * 16;
        const int64_t mine = base0 + (int64_t)sp * sstep;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4_t v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
                    io.store(mine + (int64_t)((a * TN + b) * 4 + q) * NTH * 16, v);
                }
        if (!splitk_ticket_is_last((unsigned*)g.ws + P.tile0 + tile, splits, (unsigned*)smem)) return;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
        f32x4_t cur[TM][TN][4], nxt[TM][TN][4] = {};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[a][b][q] = io.load(base0 + (int64_t)((a * TN + b) * 4 + q) * NTH * 16);
        for (int s2 = 0; s2 < splits; ++s2) {
            if (s2 + 1 < splits) {
                const int64_t src = base0 + (int64_t)(s2 + 1) * sstep;
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) nxt[a][b][q] = io.load(src + (int64_t)((a * TN + b) * 4 + q) * NTH * 16);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[a][b][4 * q] += cur[a][b][q][0];
                        acc[a][b][4 * q + 1] += cur[a][b][q][1];
                        acc[a][b][4 * q + 2] += cur[a][b][q][2];
                        acc[a][b][4 * q + 3] += cur[a][b][q][3];
                        cur[a][b][q] = nxt[a][b][q];
                    }
        }
    }

    // epilogue: C += acc (fp32, 2 x 16-byte read-modify-write per 8 columns; layout as in g2_finish)
    float* Cp = P.C;
    const int64_t ldc = P.ldc;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int64_t m = m0 + wr * WTM + a * 32 + r;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[a][b][i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half_swap(v[j], v[4 + j]);
                half_swap(v[8 + j], v[12 + j]);
            }
            const int64_t nb = n0 + wc * WTN + b * 32 + 8 * h;
            if (m < M) {
#pragma unroll
                for (int e8 = 0; e8 < 2; ++e8) {
                    const int64_t n = nb + 16 * e8;
                    if (n < N) {
                        float* pc = Cp + m * ldc + n;
                        float4 c0 = *(const float4*)pc, c1 = *(const float4*)(pc + 4);
                        const float* w = v + 8 * e8;
                        c0.x += w[0]; c0.y += w[1]; c0.z += w[2]; c0.w += w[3];
                        c1.x += w[4]; c1.y += w[5]; c1.z += w[6]; c1.w += w[7];
                        *(float4*)pc = c0;
                        *(float4*)(pc + 4) = c1;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
enum { CFG_AUTO = 0, CFG_128x128 = 1, CFG_128x64 = 2, CFG_256x128 = 3, CFG_64x128 = 4, CFG_128x128_D6 = 5, CFG_64x64 = 6,
       CFG_128x128_W8 = 7,
       // 128-byte k-tiles (KS = 4): half as many barrier / wait / issue rounds along k, for latency-bound problems
       CFG_64x64_K4 = 8, CFG_128x64_K4 = 9, CFG_64x128_K4 = 10, CFG_128x128_K4 = 11,
       // 8 waves as 2 x 4 of 128 x 64 (4 x 2 MFMA tiles per wave: 6 fragment reads per 8 MFMAs, half the LDS-DMA bytes per FLOP of
       // 128 x 128), ring of 4 x 32 KB = 128 KB, one block per CU, never split: VAE-sized problems (>= ~256 such tiles)
       CFG_256x256 = 12,
       // half of it: 4 waves of 128 x 64 (one per SIMD), ring of 4 x 24 KB, never split: problems whose N gives 256 x 256 too few tiles
       CFG_256x128_W4 = 13, CFG_LAST = 13 };

struct Cfg2 {
    int bm, bn, nth;
};
static Cfg2 cfg_dims(int c) {
    switch (c) {
        case CFG_128x64: return {128, 64, 256};
        case CFG_256x128: return {256, 128, 512};
        case CFG_64x128: return {64, 128, 256};
        case CFG_64x64: return {64, 64, 256};
        case CFG_128x128_W8: return {128, 128, 512};
        case CFG_256x256: return {256, 256, 512};
        case CFG_256x128_W4: return {256, 128, 256};
        case CFG_64x64_K4: return {64, 64, 256};
        case CFG_128x64_K4: return {128, 64, 256};
        case CFG_64x128_K4: return {64, 128, 256};
        default: return {128, 128, 256};
    }
}
static bool cfg_is_k4(int c) { return c >= CFG_64x64_K4 && c <= CFG_128x128_K4; }
static int cfg_k2_twin(int c) {  // the same block shape with 64-byte k-tiles
    switch (c) {
        case CFG_64x64_K4: return CFG_64x64;
        case CFG_128x64_K4: return CFG_128x64;
        case CFG_64x128_K4: return CFG_64x128;
        case CFG_128x128_K4: return CFG_128x128;
        default: return c;
    }
}

// tile, waves, ring depth (LDS = depth * (BM + BN) * 64 B): 128x128 / 4 deep = 64 KB (two blocks per CU), 128x64 and
// 64x128 / 6 deep = 72 KB (two per CU), 256x128 / 4 deep = 96 KB and 128x128 / 6 deep = 96 KB (one per CU), 64x64 / 8
template <bool CONV, int EB> static void launch_cfg(int c, const Args2& a, unsigned blocks, hipStream_t st) {
    switch (c) {
        case CFG_128x64: hipLaunchKernelGGL((gemm2_kernel<128, 64, 2, 2, 6, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_256x128: hipLaunchKernelGGL((gemm2_kernel<256, 128, 4, 2, 4, CONV, EB>), dim3(blocks), dim3(512), 0, st, a); break;
        case CFG_64x128: hipLaunchKernelGGL((gemm2_kernel<64, 128, 2, 2, 6, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_128x128_D6: hipLaunchKernelGGL((gemm2_kernel<128, 128, 2, 2, 6, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_64x64: hipLaunchKernelGGL((gemm2_kernel<64, 64, 2, 2, 8, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_128x128_W8:  // 8 waves (2 x 4, 64x32 each): two waves per SIMD even when a CU holds a single block
            hipLaunchKernelGGL((gemm2_kernel<128, 128, 2, 4, 4, CONV, EB>), dim3(blocks), dim3(512), 0, st, a); break;
        case CFG_256x256: hipLaunchKernelGGL((gemm2_kernel<256, 256, 2, 4, G2_NST256, CONV, EB>), dim3(blocks), dim3(512), 0, st, a); break;
        case CFG_256x128_W4: hipLaunchKernelGGL((gemm2_kernel<256, 128, 2, 2, 4, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((gemm2_kernel<128, 128, 2, 2, 4, CONV, EB>), dim3(blocks), dim3(256), 0, st, a); break;
    }
}
// 128-byte k-tiles, bf16: ring of 4 stages = the same bytes of k in flight as 8 stages of 64-byte tiles
template <bool CONV> static void launch_cfg_k4(int c, const Args2& a, unsigned blocks, hipStream_t st) {
    switch (c) {
        case CFG_64x64_K4: hipLaunchKernelGGL((gemm2_kernel<64, 64, 2, 2, 4, CONV, 2, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_128x64_K4: hipLaunchKernelGGL((gemm2_kernel<128, 64, 2, 2, 4, CONV, 2, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        case CFG_64x128_K4: hipLaunchKernelGGL((gemm2_kernel<64, 128, 2, 2, 4, CONV, 2, 4>), dim3(blocks), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((gemm2_kernel<128, 128, 2, 2, 4, CONV, 2, 4>), dim3(blocks), dim3(256), 0, st, a); break;
    }
}

// options (runtime.hip): gemm2 = 0 routes everything to gemm.hip's general kernel; g2_cfg / g2_splits force the block
// tile and the split count (tools/mb_gemm2.py sweeps them)
static bool g2_enabled() { return comat_option(COMAT_OPT_GEMM2) != 0; }
static void g2_overrides(int* cfg, int* splits) {
    *cfg = comat_option(COMAT_OPT_G2_CFG);
    *splits = comat_option(COMAT_OPT_G2_SPLITS);
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- plans: block tile and split count per problem ----
// (1) a table of measured plans for the problems of the SD1.5 / SDXL / BLIP steps (tools/tune_gemm2.py on an MI355X ->
//     tools/make_gemm2_plans.py -> gemm2_plans.inc), keyed by (conv?, M, N, k-tiles, batch);
// (2) a rule of thumb for everything else, fitted to the same measurements (profiles/r02_e_g2_tune.jsonl: 73.7 ms for
//     the step's 3535 launches against 65.3 ms with the table and 154 ms with the general kernel): the largest tile
//     that still yields >= 192 (128x128) / >= 128 (half tile) blocks, else 64x64; then cut k until ~1.5 blocks per CU,
//     every slice >= 24 k-tiles (a slice costs an fp32 slab round trip and a ticket).
struct Plan2Entry {
    int conv;
    int64_t M, N;
    int nkt, batch, cfg, splits;
};
#include "gemm2_plans.inc"

static void plan2(bool conv, bool fp8, int64_t M, int64_t N, int nkt, int64_t batch, int64_t ws_bytes, int* cfg_out,
                  int* splits_out) {
    int fc = 0, fs = 0;
    g2_overrides(&fc, &fs);
    int c = fc;
    int64_t s = fs;
    if (c == CFG_AUTO) {  // (kind 0 / 1: bf16 GEMM / conv, 2 / 3: the same with fp8 operands - measured separately, round 6)
        const int kind = (conv ? 1 : 0) + (fp8 ? 2 : 0);
        for (size_t i = 0; i < sizeof(g2_plans) / sizeof(g2_plans[0]); ++i) {
            const Plan2Entry& e = g2_plans[i];
            if (e.conv == kind && e.M == M && e.N == N && e.nkt == nkt && e.batch == batch) {
                c = e.cfg;
                if (s == 0) s = e.splits;
                break;
            }
        }
    }
    if (c == CFG_AUTO) {
        const int half = (N % 128 != 0 && N % 128 <= 64) ? CFG_128x64 : CFG_64x128;  // N = 64, 320, 960: no half-empty tiles
        const int64_t b128 = cdiv64(M, 128) * cdiv64(N, 128) * batch;
        const Cfg2 hd = cfg_dims(half);
        const int64_t bhalf = cdiv64(M, hd.bm) * cdiv64(N, hd.bn) * batch;
        if (M >= 32768 && N >= 128) c = CFG_256x128;  // VAE-sized: plenty of tiles, take the biggest
        else if (b128 >= 192) c = CFG_128x128;
        else if (bhalf >= 128) c = half;
        else c = CFG_64x64;
    }
    const Cfg2 d = cfg_dims(c);
    const int64_t ntiles = cdiv64(M, d.bm) * cdiv64(N, d.bn) * batch;
    const int64_t slab_bytes = ws_bytes - COMAT_WS_COUNTER_BYTES;
    if (slab_bytes <= 0 || ntiles > WS_COUNTERS || c == CFG_256x256 || c == CFG_256x128_W4) s = 1;
    else {
        if (s == 0) {  // about 1.5 blocks per CU, every slice at least 24 k-tiles long
            s = 1;
            if (ntiles < 384) {
                s = cdiv64(384, ntiles);
                if (s > nkt / 24) s = nkt / 24;
            }
        }
        const int64_t cap = slab_bytes / (ntiles * d.bm * d.bn * 4);
        if (s > cap) s = cap;
        if (s > nkt) s = nkt;
        if (s > 32) s = 32;
        if (s < 1) s = 1;
        const int per = (int)cdiv64(nkt, s);
        s = cdiv64(nkt, per);  // no empty slices
    }
    *cfg_out = c;
    *splits_out = (int)s;
}

static void fill_epi(Args2& a, const comat_gemm_params* p) {
    a.ep.C = p->C; a.ep.bias = p->bias; a.ep.bias2 = p->bias2; a.ep.R = p->R;
    a.ep.ldc = p->ldc; a.ep.ldr = p->ldr; a.ep.rows_per_b2 = p->rows_per_bias2 > 0 ? p->rows_per_bias2 : 1;
    a.ep.alpha = p->alpha; a.ep.beta = p->beta; a.ep.act = p->act;
    a.ep.out_dt = p->out_dtype; a.ep.r_dt = p->r_dtype;
    a.ep.C2 = p->C2; a.ep.ldc2 = p->ldc2; a.ep.epi2 = p->epi2;
    a.ep.n2 = p->epi2 == 4 ? p->n2 : 0; a.ep.alpha2 = p->alpha2;
    const bool q = p->q8 && (p->epi2 == 1 || p->epi2 == 2);
    a.ep.q8 = q ? (unsigned char*)p->q8 : nullptr; a.ep.q_scale = q ? p->q_scale : nullptr;
    a.ep.q_amax = q ? p->q_amax : nullptr; a.ep.ldq8 = q ? p->ldq8 : 0;
}

// 16-byte epilogue accesses: 8 columns per lane must stay inside a row and every row start must be 16-byte aligned
static int epi_vec_ok(const Epi& ep, int64_t N, int64_t sC, int64_t sR, int64_t sBias, int64_t M) {
    if (N % 8 || ep.ldc % 8 || !al16(ep.C) || sC % 8) return 0;
    if (ep.R && (ep.ldr % 8 || !al16(ep.R) || sR % 8)) return 0;
    if (ep.bias && (!al16(ep.bias) || sBias % 4)) return 0;
    if (ep.bias2 && !al16(ep.bias2)) return 0;
    if (M >= (1ll << 31)) return 0;
    return 1;
}

// Which way should the work items of one XCD chunk run?  Each of the 8 XCDs has a private L2 and gets a contiguous eighth of
// the items (xcd_chunk_map); an operand panel touched by k chunks is fetched k times.  Modelled L2-miss reads of the two
// orders (tools/xcd_traffic_model.py; within 3 % of FETCH_SIZE on the counter probes): rows x |A row| + columns x |B column|
// summed over the chunks.  Option g2_order: 0 = always row-block-major (rounds 1-3), 1 = always column-block-major,
// 2 = whichever the model prefers by more than 10 %.
static int tile_order(const Args2& a, bool conv, bool fp8, int bm, int bn) {
    const int opt = comat_option(COMAT_OPT_G2_ORDER);
    if (opt == 3) return 2;
    if (opt != 2) return opt == 1;
    const int64_t tm = a.tiles_m, tn = a.tiles_n, items = tm * tn;
    const double eb = fp8 ? 1.0 : 2.0;
    const double a_row = conv ? a.Cin * eb * (a.KW == 3 ? 1.25 : 1.0) : (double)a.nkt * 64.0;  // bytes of one A row
    const double b_col = (double)a.nkt * 64.0;                                               // bytes of one B column
    double cost[2] = {0.0, 0.0};
    const int64_t q = items / 8, r = items % 8;
    int64_t start = 0;
    for (int x = 0; x < 8; ++x) {
        const int64_t n = q + (x < r ? 1 : 0);
        if (n == 0) continue;
        const int64_t lo = start, hi = start + n - 1;
        start += n;
        auto span = [](int64_t l, int64_t h, int64_t blk, int64_t tot) {
            return (h + 1) * blk < tot ? (h + 1) * blk - l * blk : tot - l * blk;
        };
        {   // order 0: lin = tm_i * tn + tn_i
            const double rows = (double)span(lo / tn, hi / tn, bm, a.M);
            const double cols = hi / tn > lo / tn ? (double)a.N : (double)span(lo % tn, hi % tn, bn, a.N);
            cost[0] += rows * a_row + cols * b_col;
        }
        {   // order 1: lin = tn_i * tm + tm_i
            const double cols = (double)span(lo / tm, hi / tm, bn, a.N);
            const double rows = hi / tm > lo / tm ? (double)a.M : (double)span(lo % tm, hi % tm, bm, a.M);
            cost[1] += rows * a_row + cols * b_col;
        }
    }
    return cost[1] < 0.9 * cost[0];
}

static int finish_launch(Args2& a, bool conv, bool fp8, int64_t batch, void* ws, int64_t ws_bytes, void* stream) {
    int c, s;
    plan2(conv, fp8, a.M, a.N, a.nkt, batch, ws ? ws_bytes : 0, &c, &s);
    if (cfg_is_k4(c)) {  // 128-byte k-tiles: bf16, every segment a whole number of them, conv taps uniform per tile
        bool ok = !fp8 && (!conv || a.Cin % 64 == 0);
        for (int i = 0; i < a.nseg; ++i) ok = ok && a.seg[i].nkt % 2 == 0;
        if (!ok) c = cfg_k2_twin(c);
        else {  // split counts are slices of 128-byte tiles here: no empty slices
            const int64_t n2 = a.nkt / 2;
            if (s > n2) s = (int)(n2 > 0 ? n2 : 1);
            s = (int)cdiv64(n2, cdiv64(n2, s));
        }
    }
    const Cfg2 d = cfg_dims(c);
    a.tiles_m = (int)cdiv64(a.M, d.bm);
    a.tiles_n = (int)cdiv64(a.N, d.bn);
    a.ntiles = (int64_t)a.tiles_m * a.tiles_n * batch;
    a.splits = s;
    a.order = tile_order(a, conv, fp8, d.bm, d.bn);
    a.ws = (float*)ws;
    const int64_t blocks = a.ntiles * s;
    if (blocks >= (1ll << 31)) return 0;
    a.vec = epi_vec_ok(a.ep, a.N, a.sC, a.sR, a.sBias, a.M);
    if (fp8) {
        if (conv) launch_cfg<true, 1>(c, a, (unsigned)blocks, (hipStream_t)stream);
        else launch_cfg<false, 1>(c, a, (unsigned)blocks, (hipStream_t)stream);
        return 3;
    }
    if (cfg_is_k4(c)) {
        if (conv) launch_cfg_k4<true>(c, a, (unsigned)blocks, (hipStream_t)stream);
        else launch_cfg_k4<false>(c, a, (unsigned)blocks, (hipStream_t)stream);
        return 1;
    }
    if (conv) launch_cfg<true, 2>(c, a, (unsigned)blocks, (hipStream_t)stream);
    else launch_cfg<false, 2>(c, a, (unsigned)blocks, (hipStream_t)stream);
    return 1;
}

}  // namespace

// k-major x k-major (LoRA weight gradients): C[M, N] = A^T B, A stored [K, M], B stored [K, N]
static int try_gemm_tt(const comat_gemm_params* p, void* stream) {
    if (p->K % BK || p->lda % 8 || p->ldb % 8 || !al16(p->A) || !al16(p->B) || p->M % 8 || p->N % 8) return 0;
    if (p->batch1 > 1 && (p->sA1 % 8 || p->sB1 % 8)) return 0;
    if (p->K < 256 || p->M >= (1ll << 31) || p->N >= (1ll << 31) || !p->ws) return 0;
    Args2 a = {};
    a.seg[0].A = (const char*)p->A; a.seg[0].B = (const char*)p->B;
    a.seg[0].lda = p->lda * 2; a.seg[0].ldb = p->ldb * 2; a.seg[0].sA = p->sA1 * 2; a.seg[0].sB = p->sB1 * 2;
    a.seg[0].nkt = (int)(p->K / BK);
    a.nseg = 1;
    a.nkt = a.seg[0].nkt;
    a.M = p->M; a.N = p->N;
    a.sC = p->sC1; a.sR = p->sR1; a.sBias = 0;
    fill_epi(a, p);
    a.tiles_m = (int)cdiv64(a.M, 128);
    a.tiles_n = (int)cdiv64(a.N, 128);
    a.ntiles = (int64_t)a.tiles_m * a.tiles_n * p->batch1;
    // a handful of output tiles, thousands of tokens to contract.  Launch time ~ (nkt / s) k-tiles of main loop at ~0.35 us
    // each (a lone block per CU is DMA-latency bound) + s slabs for the last arriver at ~0.6 us each (pipelined sc1
    // reads): minimal near s = sqrt(0.6 nkt) - 12 slices for 8192 tokens, not "one block per CU" (32 slices: 42 us
    // measured, of which 38 us were the combine; profiles/r02_g_bench_shapes.txt)
    int fc = 0, fs = 0;
    g2_overrides(&fc, &fs);
    int64_t s = fs;
    if (s <= 0) {
        s = 1;
        while ((s + 1) * (s + 1) * 5 <= (int64_t)a.nkt * 3) ++s;  // floor(sqrt(0.6 nkt))
        if (a.ntiles * s < 16) s = cdiv64(16, a.ntiles);           // very few tiles: keep at least 16 blocks busy
    }
    if (s > a.nkt / 4) s = a.nkt / 4;
    const int64_t slab_bytes = p->ws_bytes - COMAT_WS_COUNTER_BYTES;
    if (slab_bytes <= 0 || a.ntiles > WS_COUNTERS) s = 1;
    else if (s > slab_bytes / (a.ntiles * 128 * 128 * 4)) s = slab_bytes / (a.ntiles * 128 * 128 * 4);
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    s = cdiv64(a.nkt, cdiv64(a.nkt, s));
    a.splits = (int)s;
    a.ws = (float*)p->ws;
    a.vec = epi_vec_ok(a.ep, a.N, a.sC, a.sR, a.sBias, a.M);
    hipLaunchKernelGGL((gemm2_tt_kernel<4>), dim3((unsigned)(a.ntiles * s)), dim3(256), 0, (hipStream_t)stream, a);
    return 2;
}

// fp8 (e4m3) operands run ONLY here (the general kernel has no fp8 path): an ineligible fp8 problem is an error of the
// caller, reported by gemm.hip.  EB = element bytes; a 16-byte chunk holds 16 / EB elements, a k-tile 64 / EB.
int comat_gemm2_try_gemm(const comat_gemm_params* p, void* stream) {
    const bool fp8 = p->in_dtype == COMAT_FP8_E4M3;
    if ((!g2_enabled() && !fp8) || (p->in_dtype != COMAT_BF16 && !fp8) || p->batch2 != 1) return 0;
    if (p->transA && p->transB) return (!fp8 && comat_option(COMAT_OPT_GEMM2_TT)) ? try_gemm_tt(p, stream) : 0;
    if (p->transA || p->transB) return 0;
    const int eb = fp8 ? 1 : 2, ch = 16 / eb, ke = RB / eb;
    if (p->K % ke || p->lda % ch || p->ldb % ch || !al16(p->A) || !al16(p->B)) return 0;
    if (p->batch1 > 1 && (p->sA1 % ch || p->sB1 % ch)) return 0;
    // (round 3: 16-row problems - the BLIP text decoder at T = 16 - come here too: three quarters of a 64-row tile are
    // clamped duplicates whose results are never stored, and the DMA ring still beats the register-staged kernel, 1.3 TFLOP/s
    // there; below 16 rows - time embeddings - the general kernel's split-K stays)
    if ((p->M < 16 && !fp8) || p->M >= (1ll << 31) || p->N >= (1ll << 31)) return 0;
    Args2 a = {};
    a.seg[0].A = (const char*)p->A; a.seg[0].B = (const char*)p->B;
    a.seg[0].lda = p->lda * eb; a.seg[0].ldb = p->ldb * eb; a.seg[0].sA = p->sA1 * eb; a.seg[0].sB = p->sB1 * eb;
    a.seg[0].nkt = (int)(p->K / ke);
    a.nseg = 1;
    a.nkt = a.seg[0].nkt;
    a.M = p->M; a.N = p->N;
    a.sC = p->sC1; a.sR = p->sR1; a.sBias = 0;
    a.scale_a = fp8 ? p->scale_a : nullptr;
    a.scale_b = fp8 ? p->scale_b : nullptr;
    a.s_scale_b = fp8 && p->batch1 > 1 ? p->s_scale_b : 0;
    if (p->A2k || p->B2k) {  // bf16 k-tail of an fp8 product: the caller's contract, checked here (no second kernel behind it)
        if (!fp8 || !p->A2k || !p->B2k || p->K2 <= 0 || p->K2 % 16 || p->K2 > 4096 || p->lda2k % 8 || p->ldb2k % 8 || p->lda2k < p->K2 ||
            p->ldb2k < p->K2 || !al16(p->A2k) || !al16(p->B2k) || p->epi2 || (p->batch1 > 1 && (p->sA2k % 8 || p->sB2k % 8))) {
            comat_set_error("comat_gemm: the bf16 k-tail (A2k, B2k) needs fp8 operands, K2 %% 16 == 0, 16-byte aligned rows and no second epilogue");
            return -1;
        }
        a.A2k = (const char*)p->A2k; a.B2k = (const char*)p->B2k;
        a.lda2k = p->lda2k * 2; a.ldb2k = p->ldb2k * 2; a.sA2k = p->sA2k * 2; a.sB2k = p->sB2k * 2;
        a.K2 = (int)p->K2;
    }
    fill_epi(a, p);
    if (p->epi2 == 4) {  // tail columns: 8-column pieces on both sides of the seam, 16-byte rows of C2 and B2; else two launches
        const int es = p->out_dtype == COMAT_F32 ? 4 : 8;
        if (fp8 || !p->B2 || !p->C2 || p->n2 <= 0 || p->n2 >= p->N || p->n2 % 8 || (p->N - p->n2) % 8 || !al16(p->C2) || !al16(p->B2) ||
            p->ldc2 % es || p->sC2_tail % es || p->sB2_tail % ch || p->ldc2 < p->n2 ||
            !epi_vec_ok(a.ep, p->N, a.sC, a.sR, a.sBias, a.M))
            return 0;
        a.B2 = (const char*)p->B2;
        a.sB2t = p->sB2_tail * eb;
        a.sC2t = p->sC2_tail;
    } else if (p->epi2 == 3) {  // GEGLU backward epilogue: C, C2 [M, 2 N] bf16 with 16-byte rows, nothing else fused
        if (p->out_dtype != COMAT_BF16 || p->N % 16 || p->R || p->bias || p->bias2 || p->act != COMAT_ACT_NONE || p->batch1 > 1 || !p->C ||
            !p->C2 || !al16(p->C) || !al16(p->C2) || p->ldc % 8 || p->ldc2 % 8 || p->ldc < 2 * p->N || p->ldc2 < 2 * p->N)
            return 0;  // declined: comat_gemm runs the two-launch form, which states its own requirements
    } else if (p->epi2) {  // GEGLU epilogue: bf16 output, 16-byte rows, whole 32-column tiles, nothing else fused
        if ((p->epi2 != 1 && p->epi2 != 2) || p->out_dtype != COMAT_BF16 || p->N % 32 || p->R || p->bias2 || p->act != COMAT_ACT_NONE ||
            p->batch1 > 1 || (!p->C2 && !p->q8) || !al16(p->C2) || p->ldc2 % 8 || (p->C2 && p->ldc2 < p->N / 2) || (p->bias && !al16(p->bias)) ||
            (p->epi2 == 1 && (!al16(p->C) || p->ldc % 8))) {
            if (!fp8 && !p->q8) return 0;  // declined: the two-launch form of comat_gemm takes over
            comat_set_error("comat_gemm: GEGLU epilogue with fp8 operands / e4m3 output: bf16 output, N %% 32 == 0, 16-byte aligned rows, "
                            "no residual / bias2 / activation / batch (no second kernel behind this path)");
            return -1;
        }
        if (p->q8 && (!p->q_scale || !p->q_amax || (((uintptr_t)p->q8) & 7) || p->ldq8 % 8 || p->ldq8 < p->N / 2)) {
            comat_set_error("comat_gemm: q8 needs q_scale, q_amax, 8-byte aligned rows and ldq8 >= N / 2");
            return -1;
        }
    }
    return finish_launch(a, false, fp8, p->batch1, p->ws, p->ws_bytes, stream);
}

int comat_gemm2_try_segments(const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg, void* stream) {
    if (!g2_enabled() || p->in_dtype != COMAT_BF16 || nseg > MAXSEG2) return 0;
    if (p->M < 16 || p->M >= (1ll << 31) || p->N >= (1ll << 31)) return 0;
    const int64_t batch = p->batch1 > 1 ? p->batch1 : 1;
    Args2 a = {};
    int64_t nkt = 0;
    for (int s = 0; s < nseg; ++s) {
        if (segs[s].K % BK || segs[s].lda % 8 || segs[s].ldb % 8 || !al16(segs[s].A) || !al16(segs[s].B)) return 0;
        if (batch > 1 && (segs[s].sA % 8 || segs[s].sB % 8)) return 0;
        a.seg[s].A = (const char*)segs[s].A; a.seg[s].B = (const char*)segs[s].B;
        a.seg[s].lda = segs[s].lda * 2; a.seg[s].ldb = segs[s].ldb * 2; a.seg[s].sA = segs[s].sA * 2; a.seg[s].sB = segs[s].sB * 2;
        a.seg[s].nkt = (int)(segs[s].K / BK);
        nkt += a.seg[s].nkt;
    }
    if (nkt >= (1ll << 30)) return 0;
    a.nseg = nseg;
    a.nkt = (int)nkt;
    a.M = p->M; a.N = p->N;
    a.sC = p->sC1; a.sR = p->sR1; a.sBias = p->bias ? p->N : 0;
    fill_epi(a, p);
    return finish_launch(a, false, false, batch, p->ws, p->ws_bytes, stream);
}

// K-segmented problem -> Args2 (no plan yet); false when the pipelined kernel does not take it
static bool fill_segments(Args2& a, const comat_gemm_params* p, const comat_gemm_segment* segs, int nseg) {
    if (p->in_dtype != COMAT_BF16 || nseg > MAXSEG2 || p->transA || p->transB || p->batch2 > 1) return false;
    if (p->M < 16 || p->M >= (1ll << 31) || p->N >= (1ll << 31)) return false;
    const int64_t batch = p->batch1 > 1 ? p->batch1 : 1;
    int64_t nkt = 0;
    for (int s = 0; s < nseg; ++s) {
        if (segs[s].K % BK || segs[s].lda % 8 || segs[s].ldb % 8 || !al16(segs[s].A) || !al16(segs[s].B)) return false;
        if (batch > 1 && (segs[s].sA % 8 || segs[s].sB % 8)) return false;
        a.seg[s].A = (const char*)segs[s].A; a.seg[s].B = (const char*)segs[s].B;
        a.seg[s].lda = segs[s].lda * 2; a.seg[s].ldb = segs[s].ldb * 2; a.seg[s].sA = segs[s].sA * 2; a.seg[s].sB = segs[s].sB * 2;
        a.seg[s].nkt = (int)(segs[s].K / BK);
        nkt += a.seg[s].nkt;
    }
    if (nkt >= (1ll << 30)) return false;
    a.nseg = nseg;
    a.nkt = (int)nkt;
    a.M = p->M; a.N = p->N;
    a.sC = p->sC1; a.sR = p->sR1; a.sBias = p->bias ? p->N : 0;
    fill_epi(a, p);
    return true;
}

int comat_gemm2_try_conv(const comat_conv_params* p, void* stream) {
    const bool fp8 = p->in_dtype == COMAT_FP8_E4M3;
    const int eb = fp8 ? 1 : 2, ke = RB / eb;
    // mode 1 (transposed gather) with stride 2 = the data-gradient of the UNet's downsamplers: three of four taps fall
    // between samples, but the zero-stuffed form runs on THIS kernel (3/4 of its MFMA work is on zeros and it is still 3x
    // faster than the register-staged kernel's 35 TFLOP/s: profiles/r02_l_bench_shapes.txt, 100 us per call)
    const bool zins = p->mode == 1 && p->stride == 2 && p->ups == 1 && !fp8 && comat_option(COMAT_OPT_GEMM2) != 0;
    if ((!g2_enabled() && !fp8) || (p->in_dtype != COMAT_BF16 && !fp8) || (p->mode != 0 && !zins) || p->Cin % ke) return 0;
    if (!al16(p->X) || !al16(p->W)) return 0;
    const int64_t M = (int64_t)p->B * p->Hout * p->Wout;
    const int64_t K = (int64_t)p->KH * p->KW * p->Cin;
    if ((M < 48 && !fp8) || M >= (1ll << 31)) return 0;
    if ((int64_t)p->B * p->Hin * p->Win * p->Cin * eb >= (1ll << 31)) return 0;  // 32-bit byte offsets of the gather
    Args2 a = {};
    a.seg[0].A = (const char*)p->X; a.seg[0].B = (const char*)p->W;
    a.seg[0].lda = 0; a.seg[0].ldb = K * eb; a.seg[0].nkt = (int)(K / ke);
    a.scale_a = fp8 ? p->scale_a : nullptr;
    a.scale_b = fp8 ? p->scale_b : nullptr;
    a.nseg = 1;
    a.nkt = a.seg[0].nkt;
    a.Hin = p->Hin; a.Win = p->Win; a.Cin = p->Cin; a.Hout = p->Hout; a.Wout = p->Wout;
    a.KW = p->KW; a.stride = p->stride; a.pad = p->pad; a.ups = p->ups;
    if (zins) {  // src = (dst + k - pad) / 2 where divisible  ==  stride-1 gather over the 2x zero-stuffed input
        if (p->Hout > 2 * p->Hin || p->Wout > 2 * p->Win) return 0;
        a.stride = 1;
        a.ups = 2;
        a.zins = 1;
    }
    a.M = M; a.N = p->Cout;
    a.sC = a.sR = a.sBias = 0;
    a.ep.C = p->Y; a.ep.bias = p->bias; a.ep.bias2 = p->bias2; a.ep.R = p->R;
    a.ep.ldc = p->Cout; a.ep.ldr = p->Cout; a.ep.rows_per_b2 = (int64_t)p->Hout * p->Wout;
    a.ep.alpha = p->alpha; a.ep.beta = p->beta; a.ep.act = p->act;
    a.ep.out_dt = p->out_dtype; a.ep.r_dt = p->r_dtype;
    return finish_launch(a, true, fp8, 1, p->ws, p->ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// grouped k-major products (include/comat_hip.h: comat_gemm_tt_grouped)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int comat_gemm_tt_grouped(const comat_tt_problem* probs, int32_t nprob, int32_t in_dtype, void* ws,
                                     int64_t ws_bytes, void* stream) {
    COMAT_REQUIRE(probs != nullptr && nprob >= 1, "comat_gemm_tt_grouped: no problems");
    COMAT_REQUIRE(in_dtype == COMAT_BF16, "comat_gemm_tt_grouped: bf16 operands only (fp32 parity mode uses comat_gemm)");
    for (int i = 0; i < nprob; ++i) {
        const comat_tt_problem& q = probs[i];
        COMAT_REQUIRE(q.A && q.B && q.C, "comat_gemm_tt_grouped: null operand in problem %d", i);
        COMAT_REQUIRE(q.M >= 8 && q.N >= 8 && q.K >= 1 && q.M % 8 == 0 && q.N % 8 == 0 && q.M < (1 << 24) && q.N < (1 << 24) &&
                          q.K < (1 << 30),
                      "comat_gemm_tt_grouped: problem %d: M, N must be multiples of 8, K >= 1", i);
        COMAT_REQUIRE(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0 && q.lda >= q.M && q.ldb >= q.N && q.ldc >= q.N &&
                          q.lda < (1 << 29) && q.ldb < (1 << 29) && q.ldc < (1ll << 31),
                      "comat_gemm_tt_grouped: problem %d: leading dimensions must keep rows 16-byte aligned", i);
        COMAT_REQUIRE(al16(q.A) && al16(q.B) && al16(q.C), "comat_gemm_tt_grouped: problem %d: operands must be 16-byte aligned", i);
    }
    const int64_t slab_cap = ws ? (ws_bytes - COMAT_WS_COUNTER_BYTES) / (128 * 128 * 4) : 0;
    int i0 = 0;
    while (i0 < nprob) {
        TTGroupArgs a = {};
        int n = 0;
        int64_t tiles_total = 0, slabs_total = 0;
        int nkt[TT_MAXP];
        int64_t spl[TT_MAXP];
        while (i0 + n < nprob && n < TT_MAXP) {
            const comat_tt_problem& q = probs[i0 + n];
            const int64_t t = cdiv64(q.M, 128) * cdiv64(q.N, 128);
            int64_t sq = slab_cap > 0 ? cdiv64(cdiv64(q.K, BK), 32) : 1;
            if (sq > 32) sq = 32;
            const int64_t need = sq > 1 ? t * sq : 0;  // slabs of this problem (upper bound)
            if (n > 0 && (tiles_total + t > WS_COUNTERS || slabs_total + need > slab_cap)) break;
            COMAT_REQUIRE(t <= WS_COUNTERS && need <= slab_cap,
                          "comat_gemm_tt_grouped: problem %d does not fit the workspace (%lld tiles, %lld slabs of 64 KiB)", i0 + n,
                          (long long)t, (long long)need);
            slabs_total += need;
            TTProb& P = a.prob[n];
            P.A = (const char*)q.A; P.B = (const char*)q.B; P.C = (float*)q.C;
            P.M = (int)q.M; P.N = (int)q.N; P.K = (int)q.K;
            P.lda = (int)(q.lda * 2); P.ldb = (int)(q.ldb * 2); P.ldc = (int)q.ldc;
            P.tiles_n = (int)cdiv64(q.N, 128);
            P.tiles = (int)t;
            nkt[n] = (int)cdiv64(q.K, BK);
            tiles_total += t;
            ++n;
        }
        // split count = a function of the PROBLEM alone (slices of ~32 k-tiles = 1 024 tokens, at most 32): the
        // summation order of a weight gradient must not depend on which other problems happen to share its launch
        // (eager steps, replayed segments and whole-step graphs group differently and still agree bit for bit)
        for (int p = 0; p < n; ++p) {
            int64_t sp = slab_cap > 0 ? cdiv64(nkt[p], 32) : 1;
            if (sp > 32) sp = 32;
            const int64_t per = cdiv64(nkt[p], sp);
            spl[p] = cdiv64(nkt[p], per);  // no empty slices
        }
        int64_t blk = 0, tile0 = 0, slab0 = 0;
        for (int p = 0; p < n; ++p) {
            TTProb& P = a.prob[p];
            P.splits = (int)spl[p];
            P.tile0 = (int)tile0;
            P.slab0 = (int)slab0;
            a.blk0[p] = (int)blk;
            blk += (int64_t)P.tiles * P.splits;
            tile0 += P.tiles;
            if (P.splits > 1) slab0 += (int64_t)P.tiles * P.splits;
        }
        COMAT_REQUIRE(blk < (1ll << 31), "comat_gemm_tt_grouped: too many work items");
        a.blk0[n] = (int)blk;
        a.nprob = n;
        a.ws = (float*)ws;
        hipLaunchKernelGGL((gemm2_tt_group_kernel<4>), dim3((unsigned)blk), dim3(256), 0, (hipStream_t)stream, a);
        i0 += n;
    }
    comat_note_gemm_kernel(4);
    return comat_check_launch("comat_gemm_tt_grouped");
}
