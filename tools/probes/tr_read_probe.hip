// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): which 16-bit elements does each lane receive?
// LDS holds element index e at byte address 2 e.  Three address patterns, one 64-lane wave each; output: 4 u16 per lane.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void probe(unsigned short* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = l * 8;                              // linear: lane l -> elements 4l..4l+3
    else if (pattern == 1) addr = (l & 15) * 64 + (l >> 4) * 8;  // 16 rows of 32 elements; 16-lane group g at column 4g
    else addr = (l & 3) * 8 + ((l >> 2) & 3) * 256 + (l >> 4) * 1024;  // 4 lanes along a row, 4 rows of 128 elements
    addr += (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;  // LDS byte address
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff;
    out[l * 4 + 1] = v.x >> 16;
    out[l * 4 + 2] = v.y & 0xffff;
    out[l * 4 + 3] = v.y >> 16;
}

int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int p = 0; p < 3; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
