// loop_probe.hip — what the CU can overlap: ds_read_b128 fragment reads, 32x32x16 bf16 MFMAs, s_barrier, in the shapes of gemm2.hip's
// k-loop (TM + TN reads feed TM x TN MFMAs per k-step; two register sets).  No global memory traffic.  One workgroup per CU
// (LDS-limited), 4 or 8 waves; wave 0 of block 0 reports s_memtime cycles per k-step.  Orders:
//   0 MFMAs only                      1 reads only
//   2 all reads of step s+1, wait for step s (counted lgkmcnt), MFMAs of step s          (gemm2.hip G2_PIN = 3)
//   3 MFMA, read, MFMA, read, ... (one read behind each of the first MFMAs), wait lgkmcnt(0) before the next step
//   4 as 3 with two reads behind each of the first MFMAs
//   5 reads for step s+2 (three register sets) interleaved one per MFMA, counted wait
// BAR = 1: an s_barrier every second k-step (as the k-tile barrier of the 64-byte k-tile shapes).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/loop_probe.hip -o tools/probes/loop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short short8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mma(f32x16_t& acc, const short8_t& a, const short8_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <int OFF> __device__ __forceinline__ void rd(short8_t& x, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
#define FENCE() __builtin_amdgcn_sched_barrier(0)
template <int N> __device__ __forceinline__ void tie(short8_t (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(x[i]));
}

template <int I, int N> struct Rd {
    static __device__ __forceinline__ void run(short8_t* x, unsigned addr) {
        if constexpr (I < N) {
            rd<I * 2048>(x[I], addr);
            Rd<I + 1, N>::run(x, addr);
        }
    }
};
// read i of a k-step's TM + TN reads: the first TM go to xf, the rest to wf
template <int I, int TM, int TN> __device__ __forceinline__ void rd_one(short8_t (&xf)[TM], short8_t (&wf)[TN], unsigned a, unsigned b) {
    if constexpr (I < TM) rd<I * 2048>(xf[I], a);
    else if constexpr (I < TM + TN) rd<(I - TM) * 2048>(wf[I - TM], b);
}

// DMA: LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave-instruction) per wave and k-tile (= two k-steps), L2-resident source, into
// a ring the fragment reads do not touch; at most two k-tiles' pieces stay in flight (counted vmcnt in front of the barrier).
//   DMODE 1: all pieces right behind the barrier (gemm2.hip today)   2: one piece behind every second MFMA
template <int ORDER, int TM, int TN, int NW, int BAR, int DMA = 0, int DMODE = 0, int NOMMA = 0>
__global__ __launch_bounds__(NW * 64) void probe(int iters, unsigned long long* cyc, float* sink, const char* gsrc = nullptr, unsigned win = 0) {
    __shared__ __attribute__((aligned(1024))) char smem[96 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += NW * 64) ((unsigned*)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    // conflict-free fragment addressing as gemm2.hip (64-byte rows, swizzled chunk)
    const int r = lane & 31, h = lane >> 5;
    const unsigned base = (unsigned)(uintptr_t)smem;
    const unsigned a0 = base + (wave & 1) * 8192 + r * 64 + ((h ^ ((r >> 2) & 3)) << 4);
    const unsigned b0 = base + 32768 + (wave >> 1) * 4096 + r * 64 + ((h ^ ((r >> 2) & 3)) << 4);
    const unsigned a1 = a0 + 32, b1 = b0 + 32;
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    short8_t x0[TM], w0[TN], x1[TM], w1[TN], x2[TM], w2[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) x0[a] = x1[a] = x2[a] = short8_t{(short)lane, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll
    for (int b = 0; b < TN; ++b) w0[b] = w1[b] = w2[b] = short8_t{(short)lane, 7, 6, 5, 4, 3, 2, 1};
    constexpr int NR = TM + TN, NM = TM * TN;
    // DMA state: piece p of this wave goes to ring slot (p % 16) of its 16 KiB region behind the fragment area
    // win = 0: every block reads the same few hundred KiB (L2-resident); win > 0: block b streams through its own window of `win` bytes
    // (a wave-instruction = 16 rows of 64 bytes, 4 KiB apart: the operand-tile shape of a K = 2048 GEMM), wrapping inside it
    const size_t lane_off = win ? (size_t)(lane >> 2) * 4096 + (lane & 3) * 16 : (size_t)(lane >> 2) * 640 + (lane & 3) * 16;
    const char* gbase = gsrc ? gsrc + 65536 + (win ? (size_t)blockIdx.x * win : (size_t)(blockIdx.x % 8) * 131072) : nullptr;
    const char* gp = gsrc ? gbase + (win ? (size_t)wave * 65536 : (size_t)wave * 8192) + lane_off : nullptr;
    unsigned gadv = 0;
    char* dring = smem + 64 * 1024 + wave * 4096;
    int dslot = 0;
    auto dma = [&]() {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(dring + (dslot & 3) * 1024), 16, 0, 0);
        if (win) {  // next 64-byte column of the same 16 rows; after 64 columns (one 4 KiB row span) the next 8-wave row group
            gadv += 64;
            const unsigned col = gadv & 4095u, grp = (gadv >> 12) * (unsigned)(NW * 65536);
            gp = gbase + (size_t)((grp + wave * 65536u) % win) + lane_off + col;
        } else {
            gp += 64;
        }
        ++dslot;
    };

    auto mm_all = [&](short8_t (&xf)[TM], short8_t (&wf)[TN]) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) mma(acc[a][b], wf[b], xf[a]);
    };
    // one k-step: MFMAs on (xc, wc), reads into (xn, wn) from (an, bn)
    auto step = [&](short8_t (&xc)[TM], short8_t (&wc)[TN], short8_t (&xn)[TM], short8_t (&wn)[TN], unsigned an, unsigned bn) {
        if constexpr (ORDER == 0) {
            mm_all(xc, wc);
        } else if constexpr (ORDER == 1) {
            Rd<0, TM>::run(xn, an);
            Rd<0, TN>::run(wn, bn);
            lgkm<0>();
            tie(xn); tie(wn);
        } else if constexpr (ORDER == 2) {
            Rd<0, TM>::run(xn, an);
            Rd<0, TN>::run(wn, bn);
            lgkm<NR>();
            tie(xc); tie(wc);
            FENCE();
            mm_all(xc, wc);
            FENCE();
        } else if constexpr (ORDER == 3 || ORDER == 4) {
            constexpr int PER = ORDER == 3 ? 1 : 2;
            lgkm<0>();
            tie(xc); tie(wc);
            FENCE();
            int m = 0;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    mma(acc[a][b], wc[b], xc[a]);
                    FENCE();
                    constexpr int dummy = 0;
                    (void)dummy;
                    ++m;
                }
            (void)m;
        }
    };
    (void)step;

    // explicit per-order loops (the interleaved orders need compile-time read indices)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (ORDER <= 2) {
        for (int it = 0; it < iters; ++it) {
            step(x0, w0, x1, w1, a1, b1);
            if (BAR) __builtin_amdgcn_s_barrier();
            step(x1, w1, x0, w0, a0, b0);
        }
    } else if constexpr (ORDER == 3 || ORDER == 4) {
        constexpr int PER = ORDER == 3 ? 1 : 2;
        auto stp = [&](short8_t (&xc)[TM], short8_t (&wc)[TN], short8_t (&xn)[TM], short8_t (&wn)[TN], unsigned an, unsigned bn) {
            lgkm<0>();
            tie(xc); tie(wc);
            FENCE();
            // MFMA m followed by reads m*PER .. m*PER+PER-1
            auto one = [&](auto mtag) {
                constexpr int m = decltype(mtag)::value;
                if constexpr (!NOMMA) mma(acc[m / TN][m % TN], wc[m % TN], xc[m / TN]);
                FENCE();
                rd_one<m * PER, TM, TN>(xn, wn, an, bn);
                if constexpr (PER == 2) rd_one<m * PER + 1, TM, TN>(xn, wn, an, bn);
                if constexpr (DMODE == 2 && (m & 1) == 1 && (m / 2) < DMA / 2) dma();  // DMA / 2 pieces per k-step, behind MFMAs 1, 3, ..
                FENCE();
            };
            one(std::integral_constant<int, 0>{});
            if constexpr (NM > 1) one(std::integral_constant<int, 1>{});
            if constexpr (NM > 2) one(std::integral_constant<int, 2>{});
            if constexpr (NM > 3) one(std::integral_constant<int, 3>{});
            if constexpr (NM > 4) one(std::integral_constant<int, 4>{});
            if constexpr (NM > 5) one(std::integral_constant<int, 5>{});
            if constexpr (NM > 6) one(std::integral_constant<int, 6>{});
            if constexpr (NM > 7) one(std::integral_constant<int, 7>{});
            static_assert(NM <= 8 && NR <= NM * PER, "probe sizes");
        };
        for (int it = 0; it < iters; ++it) {
            stp(x0, w0, x1, w1, a1, b1);
            if constexpr (DMA > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");  // the pieces of the k-tile before the last have landed
                if (!win && (it & 63) == 63) gp -= 64 * 64 * DMA;            // stay inside the L2-resident window
            }
            if (BAR) __builtin_amdgcn_s_barrier();
            if constexpr (DMODE == 1) {
#pragma unroll
                for (int d = 0; d < DMA; ++d) dma();
                FENCE();
            }
            stp(x1, w1, x0, w0, a0, b0);
        }
    } else {  // ORDER 5: three sets, reads of step s+2 one per MFMA, counted wait: before step s+1 only step s+2's reads may be pending
        auto stp = [&](short8_t (&xc)[TM], short8_t (&wc)[TN], short8_t (&xn)[TM], short8_t (&wn)[TN], unsigned an, unsigned bn) {
            lgkm<NR>();  // the reads issued during the previous step may be pending; those of the step before (this step's data) not
            tie(xc); tie(wc);
            FENCE();
            auto one = [&](auto mtag) {
                constexpr int m = decltype(mtag)::value;
                mma(acc[m / TN][m % TN], wc[m % TN], xc[m / TN]);
                FENCE();
                rd_one<m, TM, TN>(xn, wn, an, bn);
                FENCE();
            };
            one(std::integral_constant<int, 0>{});
            if constexpr (NM > 1) one(std::integral_constant<int, 1>{});
            if constexpr (NM > 2) one(std::integral_constant<int, 2>{});
            if constexpr (NM > 3) one(std::integral_constant<int, 3>{});
            if constexpr (NM > 4) one(std::integral_constant<int, 4>{});
            if constexpr (NM > 5) one(std::integral_constant<int, 5>{});
            if constexpr (NM > 6) one(std::integral_constant<int, 6>{});
            if constexpr (NM > 7) one(std::integral_constant<int, 7>{});
        };
        for (int it = 0; it < iters; it += 3) {  // six k-steps per round: sets rotate 0 -> 1 -> 2
            stp(x0, w0, x2, w2, a0, b0);
            if (BAR) __builtin_amdgcn_s_barrier();
            stp(x1, w1, x0, w0, a1, b1);
            stp(x2, w2, x1, w1, a0, b0);
            if (BAR) __builtin_amdgcn_s_barrier();
            stp(x0, w0, x2, w2, a1, b1);
            stp(x1, w1, x0, w0, a0, b0);
            if (BAR) __builtin_amdgcn_s_barrier();
            stp(x2, w2, x1, w1, a1, b1);
        }
    }
    lgkm<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) s += acc[a][b][0] + acc[a][b][7];
    s += (float)x0[0][0] + (float)x1[0][1] + (float)x2[0][2] + (float)w0[0][0] + (float)w1[0][0] + (float)w2[0][0];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static const char* g_src = nullptr;
static unsigned g_win = 0;
template <int ORDER, int TM, int TN, int NW, int BAR, int DMA = 0, int DMODE = 0, int NOMMA = 0> static void run(const char* name, int blocks, unsigned long long* dcyc, float* dsink) {
    const int iters = 1200;  // multiple of 3; two k-steps per iteration
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<ORDER, TM, TN, NW, BAR, DMA, DMODE, NOMMA><<<blocks, NW * 64>>>(iters, dcyc, dsink, g_src, g_win);
    hipEventRecord(e0);
    probe<ORDER, TM, TN, NW, BAR, DMA, DMODE, NOMMA><<<blocks, NW * 64>>>(iters, dcyc, dsink, g_src, g_win);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), dcyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const double steps = 2.0 * iters;
    const double clk = (double)c[0] / steps;
    const double mf = ORDER == 1 ? 0 : (double)TM * TN * 32.0 * (NW / 4);  // MFMA pipe cycles per k-step and SIMD
    if (DMA) printf("[win %5u KiB, %6.2f TB/s] ", g_win >> 10, (double)DMA * NW * 1024.0 * iters * blocks / (ms * 1e-3) / 1e12);
    printf("%-44s TM=%d TN=%d waves=%d bar=%d dma=%d/%d blocks=%3d : %7.1f clk / k-step (MFMA floor %4.0f, %4.1f %% busy)  wall %.3f ms -> %.2f GHz\n", name, TM,
           TN, NW, BAR, DMA, DMODE, blocks, clk, mf, mf > 0 ? 100.0 * mf / clk : 0.0, ms, (double)c[0] / (ms * 1e6));
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  HIP error: %s\n", hipGetErrorString(err));
}

#define RUN_ALL(TM, TN, NW, BAR, B)                                                    \
    run<0, TM, TN, NW, BAR>("0 MFMAs only", B, dcyc, dsink);                          \
    run<1, TM, TN, NW, BAR>("1 reads only", B, dcyc, dsink);                          \
    run<2, TM, TN, NW, BAR>("2 reads first, counted wait, MFMAs", B, dcyc, dsink);    \
    run<3, TM, TN, NW, BAR>("3 MFMA/read interleaved, wait all", B, dcyc, dsink);     \
    run<5, TM, TN, NW, BAR>("5 three sets, read per MFMA, counted", B, dcyc, dsink);

int main() {
    unsigned long long* dcyc;
    float* dsink;
    hipMalloc(&dcyc, 1024 * sizeof(unsigned long long));
    hipMalloc(&dsink, 64);
    char* src;
    const size_t src_bytes = (size_t)600 << 20;
    hipMalloc(&src, src_bytes);
    hipMemset(src, 0x3c, src_bytes);
    g_src = src;
    for (unsigned win : {0u, 1u << 20, 2u << 20}) {  // per-block windows: shared (L2), 256 MB in all (MALL-sized), 512 MB (HBM)
        g_win = win;
        const int blocks = 256;
        run<3, 4, 2, 8, 1, 4, 2, 1>("reads + 4 DMA pieces, no MFMA", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 4, 2, 0>("MFMAs + reads + 4 DMA pieces between", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 4, 1, 0>("MFMAs + reads + 4 DMA pieces behind barrier", blocks, dcyc, dsink);
        run<3, 2, 2, 8, 1, 2, 2, 1>("reads + 2 DMA pieces, no MFMA", blocks, dcyc, dsink);
        run<3, 2, 2, 8, 1, 2, 2, 0>("MFMAs + reads + 2 DMA pieces between", blocks, dcyc, dsink);
    }
    g_win = 0;
    if (getenv("PROBE_DMA")) for (int blocks : {256}) {
        run<3, 4, 2, 8, 1, 0, 0>("3 interleaved reads, no DMA", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 4, 1>("3 + 4 DMA pieces behind the barrier", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 4, 2>("3 + 4 DMA pieces between MFMAs", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 8, 1>("3 + 8 DMA pieces behind the barrier", blocks, dcyc, dsink);
        run<3, 4, 2, 8, 1, 8, 2>("3 + 8 DMA pieces between MFMAs", blocks, dcyc, dsink);
        run<3, 2, 2, 4, 1, 0, 0>("3 interleaved reads, no DMA", blocks, dcyc, dsink);
        run<3, 2, 2, 4, 1, 4, 1>("3 + 4 DMA pieces behind the barrier", blocks, dcyc, dsink);
        run<3, 2, 2, 4, 1, 4, 2>("3 + 4 DMA pieces between MFMAs", blocks, dcyc, dsink);
        run<3, 2, 2, 8, 1, 2, 1>("3 + 2 DMA pieces behind the barrier", blocks, dcyc, dsink);
        run<3, 2, 2, 8, 1, 2, 2>("3 + 2 DMA pieces between MFMAs", blocks, dcyc, dsink);
    }
    if (getenv("PROBE_ALL")) for (int blocks : {256, 8}) {
        RUN_ALL(4, 2, 8, 0, blocks)
        RUN_ALL(4, 2, 8, 1, blocks)
        run<4, 4, 2, 8, 1>("4 MFMA/2 reads interleaved, wait all", blocks, dcyc, dsink);
        RUN_ALL(2, 2, 4, 0, blocks)
        RUN_ALL(2, 2, 4, 1, blocks)
        run<4, 2, 2, 4, 1>("4 MFMA/2 reads interleaved, wait all", blocks, dcyc, dsink);
        RUN_ALL(2, 2, 8, 1, blocks)
    }
    return 0;
}
