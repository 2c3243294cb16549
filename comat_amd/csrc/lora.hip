// lora.hip — merged LoRA weights for every projection of a model in ONE launch.
//
// The reference's LoRA-wrapped Linear computes  y = W x + s * up(down(x))  (training_utils/pipeline.py:94-115), i.e. the
// function  y = (W + s U D) x.  The trained calls of a step (5 UNet forwards + backwards, the discriminator's calls) use
// W_eff = W + s U D directly - one plain GEMM per projection instead of a rank-r product on the dependent chain in front of
// a K-segmented one - and the data-gradient uses W_eff^T.  The factors only change at the optimizer step, so both
// orientations of every W_eff are refreshed once per step here: one launch over a tile table instead of two small launches
// per projection group (SD1.5: 80 groups per UNet).
//
//   Wm_p [N, K]  = bf16( W_p + s * U_p[N, r] D_p[r, K] )        WmT_p [K, N] = Wm_p^T   (the same bits, transposed)
//
// One 64 x 64 tile per workgroup (4 waves, a 32 x 32 MFMA tile each, fragments straight from global memory: r <= 256, the
// operands of a tile are 2 x 16 KB and L2-resident), fp32 accumulation, ONE rounding to bf16; the tile goes through LDS so
// that both orientations leave in 16-byte rows.  HBM-bound by construction (reads W once, writes both copies once).
#include "gemm_shared.h"

namespace {

constexpr int LM_T = 64;       // tile edge
constexpr int LM_LD = LM_T + 8;  // LDS row pitch in bf16 (16-byte aligned rows, no 2-way conflicts on the row reads)

struct LmProblem {  // mirrors the int64 [n, 10] table of comat_lora_merge
    const bf16_t* W;
    const bf16_t* U;
    const bf16_t* Dt;
    bf16_t* Wm;
    bf16_t* WmT;
    int64_t N, K, r, ldu, lddt;
};

__global__ __launch_bounds__(256) void lora_merge_kernel(const LmProblem* __restrict__ problems,
                                                         const int32_t* __restrict__ tiles, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[LM_T * LM_LD];
    const int32_t* t = tiles + (int64_t)blockIdx.x * 3;
    const LmProblem p = problems[t[0]];
    const int n0 = t[1], k0 = t[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int N = (int)p.N, K = (int)p.K, r = (int)p.r;
    // A operand: U rows n (lane & 31), 8 consecutive j per lane; B operand: D^T rows k
    const int an = n0 + wn * 32 + (lane & 31), bk = k0 + wk * 32 + (lane & 31);
    const int jl = (lane >> 5) * 8;
    const bool a_ok = an < N, b_ok = bk < K;
    const bf16_t* ap = p.U + (int64_t)(a_ok ? an : 0) * p.ldu + jl;
    const bf16_t* bp = p.Dt + (int64_t)(b_ok ? bk : 0) * p.lddt + jl;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const short8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < r; j += 16) {
        const short8_t a = a_ok ? *(const short8_t*)(ap + j) : zero;
        const short8_t b = b_ok ? *(const short8_t*)(bp + j) : zero;
        mma_t(acc, a, b);  // acc[n][k] += U[n][j] D^T[k][j]
    }
    // lane holds column k = lane & 31 of the wave tile and rows n = 8 (i / 4) + 4 (lane / 32) + i % 4
    const int kk = wk * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int nn = wn * 32 + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3);
        float w = 0.f;
        if (n0 + nn < N && k0 + kk < K) w = bf16_to_f32(p.W[(int64_t)(n0 + nn) * K + k0 + kk]);
        tile[nn * LM_LD + kk] = f32_to_bf16(w + scale * acc[i]);
    }
    __syncthreads();
    // Wm rows: 8 lanes x 16 bytes per row, 32 rows per pass
    {
        const int c8 = (threadIdx.x & 7) * 8;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 32 + (threadIdx.x >> 3);
            if (n0 + row < N && k0 + c8 < K)
                *(uint4*)(p.Wm + (int64_t)(n0 + row) * K + k0 + c8) = *(const uint4*)(tile + row * LM_LD + c8);
        }
    }
    if (p.WmT) {  // WmT rows k: 8 consecutive n per lane, gathered down a tile column
        const int c8 = (threadIdx.x & 7) * 8;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int krow = pass * 32 + (threadIdx.x >> 3);
            Pack16 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.h[e] = tile[(c8 + e) * LM_LD + krow];
            if (k0 + krow < K && n0 + c8 < N) *(uint4*)(p.WmT + (int64_t)(k0 + krow) * N + n0 + c8) = pk.u;
        }
    }
}

}  // namespace

extern "C" int comat_lora_merge(const int64_t* problems, const int32_t* tiles, int64_t n_tiles, float scale, void* stream) {
    COMAT_REQUIRE(problems && tiles && n_tiles > 0 && n_tiles < (1ll << 31), "comat_lora_merge: bad args");
    static_assert(sizeof(LmProblem) == 10 * sizeof(int64_t), "problem table layout");
    hipLaunchKernelGGL(lora_merge_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const LmProblem*)problems, tiles, scale);
    return comat_check_launch("comat_lora_merge");
}
