// fill_rate_probe.hip — how many bytes per clock one CU can pull from L2 (and from HBM) into LDS through the LDS-DMA
// (global_load_lds_dwordx4) and into VGPRs through global_load_dwordx4, in the access shape of the GEMM kernels' operand tiles
// (a wave-instruction = 8 rows of 128 bytes, rows `ld` bytes apart) and in fully linear 1 KiB pieces.  One workgroup per CU
// (LDS-limited), 4 or 8 waves, DEPTH instructions in flight per wave.  No MFMA, no LDS reads: this is the ceiling of the
// operand fill alone - what the tile shapes of gemm2.hip divide their FLOPs by.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fill_rate_probe.hip -o tools/probes/fill_rate_probe.bin ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: LDS-DMA, rows; 1: LDS-DMA, linear; 2: VGPR loads, rows; 3: VGPR loads, linear
template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char* __restrict__ src, size_t mask, int ld, int iters, unsigned* out,
                                                       unsigned long long* cyc) {
    __shared__ __attribute__((aligned(1024))) char smem[96 * 1024];  // > 80 KiB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wave_global = (size_t)blockIdx.x * NW + wave;
    // rows: lane -> (row = lane / 8, 16-byte chunk = lane % 8) of 8 rows that lie `ld` bytes apart; linear: lane * 16
    const size_t lane_off = (MODE & 1) ? (size_t)lane * 16 : (size_t)(lane >> 3) * ld + (lane & 7) * 16;
    const size_t step = (MODE & 1) ? 1024 : (size_t)8 * ld;  // bytes of address space one instruction covers
    size_t pos = ((size_t)blockIdx.x * 977 * NW + wave) * step;  // workgroups start apart, their waves at neighbouring pieces
    char* dst = smem + wave * (DEPTH * 1024 > 8192 ? 8192 : DEPTH * 1024);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE < 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const char* p = src + ((pos + lane_off) & mask);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(dst + ((it * 4 + u) & 7) * 1024), 16, 0, 0);
                pos += step * NW;  // the waves of a workgroup interleave: consecutive pieces go to consecutive waves
            }
            wait_vmcnt<DEPTH - 4>();
        }
    } else {
        // plain loads: DEPTH loads issued, then all of them consumed (the compiler waits for each at its use)
        for (int it = 0; it < iters * 4 / DEPTH; ++it) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                v[u] = *(const u32x4*)(src + ((pos + lane_off) & mask));
                pos += step * NW;
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) acc ^= v[u];
        }
    }
    wait_vmcnt<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 0x12345678u && out) out[0] = acc.y;  // keeps the loads alive
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NW, int DEPTH> static void run(const char* name, const char* buf, size_t bytes, int ld, int iters, int ncu,
                                                       unsigned long long* dcyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<MODE, NW, DEPTH>), dim3(ncu), dim3(NW * 64), 0, 0, buf, bytes - 1, ld, iters, (unsigned*)nullptr, dcyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(ncu);
    hipMemcpy(cyc.data(), dcyc, ncu * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : cyc) mean += (double)c;
    mean /= ncu;
    const double per_cu = (double)NW * iters * 4 * 1024;
    printf("%-34s waves %d depth %2d  ws %6.1f MB ld %5d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (%.0f clk)\n", name, NW, DEPTH, bytes / 1048576.0,
           ld, ms * 1e3, per_cu * ncu / (ms * 1e-3) / 1e12, per_cu / mean, mean);
    fflush(stdout);
}

int main() {
    int ncu = 256;
    const size_t big = (size_t)1 << 30;
    char* buf;
    hipMalloc(&buf, big);
    hipMemset(buf, 1, big);
    unsigned long long* dcyc;
    hipMalloc(&dcyc, 4096 * sizeof(unsigned long long));
    const int iters = 512;
    for (size_t ws : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, big}) {
        for (int ld : {640, 2560}) {
            run<0, 4, 8>("lds-dma rows", buf, ws, ld, iters, ncu, dcyc);
            run<0, 4, 16>("lds-dma rows", buf, ws, ld, iters, ncu, dcyc);
            run<0, 8, 8>("lds-dma rows", buf, ws, ld, iters, ncu, dcyc);
            run<0, 8, 16>("lds-dma rows", buf, ws, ld, iters, ncu, dcyc);
            run<2, 4, 16>("vgpr loads rows", buf, ws, ld, iters, ncu, dcyc);
            run<2, 8, 16>("vgpr loads rows", buf, ws, ld, iters, ncu, dcyc);
            run<2, 8, 32>("vgpr loads rows", buf, ws, ld, iters, ncu, dcyc);
        }
        run<1, 4, 16>("lds-dma linear", buf, ws, 0, iters, ncu, dcyc);
        run<1, 8, 16>("lds-dma linear", buf, ws, 0, iters, ncu, dcyc);
        run<3, 4, 16>("vgpr loads linear", buf, ws, 0, iters, ncu, dcyc);
        run<3, 8, 16>("vgpr loads linear", buf, ws, 0, iters, ncu, dcyc);
        run<3, 8, 32>("vgpr loads linear", buf, ws, 0, iters, ncu, dcyc);
    }
    // a single CU busy (the rest of the chip idle): the per-CU limit without fabric contention
    run<0, 8, 16>("lds-dma rows, ONE workgroup", buf, (size_t)2 << 20, 640, iters, 1, dcyc);
    run<2, 8, 32>("vgpr loads rows, ONE workgroup", buf, (size_t)2 << 20, 640, iters, 1, dcyc);
    run<0, 8, 16>("lds-dma rows, 32 workgroups", buf, (size_t)2 << 20, 640, iters, 32, dcyc);
    return 0;
}
