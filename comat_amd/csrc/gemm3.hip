// gemm3.hip — the "lean" MFMA GEMM for the launch-latency-bound products of the step with very few rows.
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
// (Round 4 also ran "chained" launches on this kernel and on gemm2.hip - a producer GEMM and the K-segmented GEMM consuming its
// output in one launch, for the LoRA pairs; measured no faster than two launches (profiles/r04_c_*), and the pairs themselves
// left the dependent chain in round 5 (merged weights).  The code is on branch exp/gemm-chain.)
#include "gemm_shared.h"

namespace {

constexpr int CKB = 128;        // bytes of k per chunk and operand row (64 bf16)
constexpr int CKE = 64;         // k elements per chunk
constexpr int MAXSEG3 = 8;

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
    const Prob3& P = g.main;

    unsigned lin = (unsigned)xcd_chunk_map((int64_t)blockIdx.x, (int64_t)gridDim.x);
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
    // The steady-state loop issues the loads of two chunks unconditionally (the compiler's vmcnt bookkeeping is exact only
    // without branches around the loads).
    int seg = 0, seg_left = 0;
    const char* pa[TM];
    const char* pb[TN];
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
    if (cnt > 0) {
        int s = 0, skip = c0;
        while (s + 1 < P.nseg && skip >= P.seg[s].nch) {
            skip -= P.seg[s].nch;
            ++s;
        }
        enter_segment(s, skip);
        run(cnt, load_plain);
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
        if (m < P.M && nb < P.N) epilogue_run<false>(ep, v, m, nb, P.N, vec);
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
