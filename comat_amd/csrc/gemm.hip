// gemm.hip — MFMA GEMM and implicit-GEMM conv2d for gfx950 (CDNA4).
//
// One LDS-staged, register-double-buffered tile kernel serves every dense contraction on the CoMat step:
//   * bf16 storage  -> v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
//   * fp32 storage  -> v_mfma_f32_32x32x2_f32   (exact f32: the parity mode, 1/16 of the bf16 rate)
// Block = 256 threads = 4 waves (64 lanes) in a 2x2 arrangement; block tile 128x128 (each wave 2x2 MFMA tiles of
// 32x32) or 64x64 (each wave one tile).  One k-tile is 64 bytes per row for both dtypes (32 bf16 / 16 fp32), so
// the LDS image and the fragment addressing are dtype-independent:
//   k-contiguous operand: LDS [rows][80 B] (64 B data + 16 B pad -> ds_read_b128 of a 16-lane group touches 16
//                         distinct 16-byte slots: 20*r mod 64 is injective for r mod 16);
//   k-major operand     : LDS [k][rows*sizeof(T) + 16 B], fragments gathered with scalar LDS reads.
// Fragment rule: lane (r = lane&31, h = lane>>5) of k-step s holds the 16 bytes at k-offset s*32 + h*16 of row r,
// for A and B alike, so any k-permutation inside the MFMA cancels (sum over k).  C/D mapping of the 32x32 MFMA:
// col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int ROWB = 80;             // bytes per LDS row, k-contiguous image
constexpr int LDS_OP_BYTES = 10240;  // per operand per stage (128 rows * 80 B >= any k-major image)

struct Epi {
    void* C;
    const float* bias;
    const float* bias2;
    const void* R;
    int64_t ldc, ldr, rows_per_b2;
    float alpha, beta;
    int act, out_dt, r_dt;
};

template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { typedef short8_t type; };
template <> struct FragOf<float> { typedef f32x4_t type; };

union Vec16 {
    uint4 u;
    float f[4];
    bf16_t h[8];
};

// ---------------------------------------------------------------------------------------------------------------
// Operand loaders: global -> registers (issued before the MFMA phase) -> LDS (after it).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int ROWS, bool TRANS> struct PlainLoader {
    static constexpr bool kTrans = TRANS;
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int NCH = ROWS * 4 / NT;
    static constexpr int VPR = ROWS / EPV;               // 16-byte vectors per k-row (k-major image)
    static constexpr int RS = ROWS * (int)sizeof(T) + 16;  // k-major LDS row stride in bytes
    const T* P;
    int64_t ld, R0, Rmax, K;
    bool vec_ok;
    uint4 regs[NCH];

    __device__ __forceinline__ void init(const void* p, int64_t ld_, int64_t r0, int64_t rmax, int64_t k_) {
        P = (const T*)p;
        ld = ld_;
        R0 = r0;
        Rmax = rmax;
        K = k_;
        vec_ok = ((ld % EPV) == 0) && ((((uintptr_t)p) & 15) == 0);
    }
    __device__ __forceinline__ void load(int64_t k0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT;
            Vec16 v;
            v.u = make_uint4(0, 0, 0, 0);
            if (!TRANS) {
                const int row = c >> 2, kv = c & 3;
                const int64_t gk = k0 + kv * EPV, gr = R0 + row;
                if (gr < Rmax && gk < K) {
                    const T* src = P + gr * ld + gk;
                    if (vec_ok && gk + EPV <= K) {
                        v.u = *(const uint4*)src;
                    } else {
#pragma unroll
                        for (int e = 0; e < EPV; ++e)
                            if (gk + e < K) {
                                if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)src)[e];
                                else v.f[e] = ((const float*)src)[e];
                            }
                    }
                }
            } else {
                const int kk = c / VPR, mv = c % VPR;
                const int64_t gk = k0 + kk, gr = R0 + (int64_t)mv * EPV;
                if (gk < K && gr < Rmax) {
                    const T* src = P + gk * ld + gr;
                    if (vec_ok && gr + EPV <= Rmax) {
                        v.u = *(const uint4*)src;
                    } else {
#pragma unroll
                        for (int e = 0; e < EPV; ++e)
                            if (gr + e < Rmax) {
                                if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)src)[e];
                                else v.f[e] = ((const float*)src)[e];
                            }
                    }
                }
            }
            regs[i] = v.u;
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT;
            if (!TRANS) *(uint4*)(lds + (c >> 2) * ROWB + (c & 3) * 16) = regs[i];
            else *(uint4*)(lds + (c / VPR) * RS + (c % VPR) * 16) = regs[i];
        }
    }
};

struct ConvGeom {
    int B, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, mode, ups;
};

// im2col gather of a channels-last activation: row m = (b, oy, ox), k = (ky, kx, ci).
template <typename T, int ROWS> struct ConvLoader {
    static constexpr bool kTrans = false;
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int NCH = ROWS * 4 / NT;
    const T* X;
    ConvGeom g;
    int64_t K;
    bool vec_ok;
    int rb[NCH], roy[NCH], rox[NCH];
    bool rvalid[NCH];
    uint4 regs[NCH];

    __device__ __forceinline__ void init(const void* x, const ConvGeom& g_, int64_t r0, int64_t M) {
        X = (const T*)x;
        g = g_;
        K = (int64_t)g.KH * g.KW * g.Cin;
        vec_ok = ((g.Cin % EPV) == 0) && ((((uintptr_t)x) & 15) == 0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT;
            const int64_t gm = r0 + (c >> 2);
            rvalid[i] = gm < M;
            const int64_t hw = (int64_t)g.Hout * g.Wout;
            const int64_t b = gm / hw, rem = gm - b * hw;
            rb[i] = (int)b;
            roy[i] = (int)(rem / g.Wout);
            rox[i] = (int)(rem - (int64_t)roy[i] * g.Wout);
        }
    }
    // pointer to X[b, sy, sx, ci] for output position (oy, ox) and tap, or nullptr when the tap falls in padding
    __device__ __forceinline__ const T* src_ptr(int b, int oy, int ox, int tap, int ci) const {
        const int ky = tap / g.KW, kx = tap - ky * g.KW;
        int sy, sx;
        if (g.mode == 0) {
            sy = oy * g.stride + ky - g.pad;
            sx = ox * g.stride + kx - g.pad;
            if (sy < 0 || sx < 0 || sy >= g.Hin * g.ups || sx >= g.Win * g.ups) return nullptr;
            if (g.ups == 2) {
                sy >>= 1;
                sx >>= 1;
            }
        } else {
            const int ty = oy + ky - g.pad, tx = ox + kx - g.pad;
            if (ty < 0 || tx < 0) return nullptr;
            if ((ty % g.stride) != 0 || (tx % g.stride) != 0) return nullptr;
            sy = ty / g.stride;
            sx = tx / g.stride;
            if (sy >= g.Hin || sx >= g.Win) return nullptr;
        }
        return X + (((int64_t)b * g.Hin + sy) * g.Win + sx) * g.Cin + ci;
    }
    __device__ __forceinline__ void load(int64_t k0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT;
            const int kv = c & 3;
            const int64_t gk = k0 + kv * EPV;
            Vec16 v;
            v.u = make_uint4(0, 0, 0, 0);
            if (rvalid[i] && gk < K) {
                if (vec_ok) {
                    const int tap = (int)(gk / g.Cin), ci = (int)(gk - (int64_t)tap * g.Cin);
                    const T* src = src_ptr(rb[i], roy[i], rox[i], tap, ci);
                    if (src) v.u = *(const uint4*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        const int64_t k = gk + e;
                        if (k < K) {
                            const int tap = (int)(k / g.Cin), ci = (int)(k - (int64_t)tap * g.Cin);
                            const T* src = src_ptr(rb[i], roy[i], rox[i], tap, ci);
                            if (src) {
                                if (sizeof(T) == 2) v.h[e] = *(const bf16_t*)src;
                                else v.f[e] = *(const float*)src;
                            }
                        }
                    }
                }
            }
            regs[i] = v.u;
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = threadIdx.x + i * NT;
            *(uint4*)(lds + (c >> 2) * ROWB + (c & 3) * 16) = regs[i];
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// LDS -> MFMA fragments
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int ROWS, bool TRANS>
__device__ __forceinline__ typename FragOf<T>::type read_frag(const char* lds, int row, int s, int h) {
    typedef typename FragOf<T>::type F;
    if (!TRANS) {
        return *(const F*)(lds + row * ROWB + s * 32 + h * 16);
    } else {
        constexpr int EPV = 16 / sizeof(T);
        constexpr int RS = ROWS * (int)sizeof(T) + 16;
        const int ke0 = s * (32 / (int)sizeof(T)) + h * EPV;
        Vec16 v;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            const char* p = lds + (ke0 + j) * RS + row * (int)sizeof(T);
            if (sizeof(T) == 2) v.h[j] = *(const bf16_t*)p;
            else v.f[j] = *(const float*)p;
        }
        F out;
        __builtin_memcpy(&out, &v, 16);
        return out;
    }
}

__device__ __forceinline__ void mma(f32x16_t& acc, const short8_t& a, const short8_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x16_t& acc, const f32x4_t& a, const f32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Block-level main loop + fused epilogue
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, typename AL, typename BL>
__device__ __forceinline__ void gemm_block(AL& al, BL& bl, int64_t K, int64_t m0, int64_t n0, int64_t M, int64_t N,
                                           const Epi& ep) {
    constexpr int BKE = 64 / sizeof(T);
    constexpr int WTM = BM / 2, WTN = BN / 2, TM = WTM / 32, TN = WTN / 32;
    typedef typename FragOf<T>::type F;
    __shared__ __attribute__((aligned(16))) char smem[2][2][LDS_OP_BYTES];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

    const int64_t nk = (K + BKE - 1) / BKE;
    al.load(0);
    bl.load(0);
    al.store(smem[0][0]);
    bl.store(smem[0][1]);
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = (kt + 1 < nk);
        if (more) {
            al.load((kt + 1) * BKE);
            bl.load((kt + 1) * BKE);
        }
        const char* la = smem[cur][0];
        const char* lb = smem[cur][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            F af[TM], bfr[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = read_frag<T, BM, AL::kTrans>(la, wr * WTM + a * 32 + r, s, h);
#pragma unroll
            for (int b = 0; b < TN; ++b) bfr[b] = read_frag<T, BN, BL::kTrans>(lb, wc * WTN + b * 32 + r, s, h);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) mma(acc[a][b], af[a], bfr[b]);
        }
        if (more) {
            al.store(smem[cur ^ 1][0]);
            bl.store(smem[cur ^ 1][1]);
        }
        __syncthreads();
    }

    // epilogue: v = act(alpha*acc + bias + bias2) + beta*R
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int64_t col = n0 + wc * WTN + b * 32 + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t row = m0 + wr * WTM + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M && col < N) {
                    float v = ep.alpha * acc[a][b][i];
                    if (ep.bias) v += ep.bias[col];
                    if (ep.bias2) v += ep.bias2[(row / ep.rows_per_b2) * N + col];
                    if (ep.act == COMAT_ACT_SILU) v = silu_f(v);
                    else if (ep.act == COMAT_ACT_GELU) v = gelu_f(v);
                    if (ep.R) v += ep.beta * ld_dt(ep.R, row * ep.ldr + col, ep.r_dt);
                    st_dt(ep.C, row * ep.ldc + col, v, ep.out_dt);
                }
            }
        }
}

struct GemmArgs {
    const void* A;
    const void* B;
    int64_t M, N, K, lda, ldb;
    int64_t batch2, sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    int tiles_m;
    Epi ep;
};

template <typename T, int BM, int BN, bool TA, bool TB> __global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
    const int64_t z = blockIdx.z, b1 = z / g.batch2, b2 = z - b1 * g.batch2;
    const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const T* A = (const T*)g.A + b1 * g.sA1 + b2 * g.sA2;
    const T* B = (const T*)g.B + b1 * g.sB1 + b2 * g.sB2;
    Epi ep = g.ep;
    const int64_t coff = b1 * g.sC1 + b2 * g.sC2, roff = b1 * g.sR1 + b2 * g.sR2;
    ep.C = (char*)ep.C + coff * (ep.out_dt == COMAT_F32 ? 4 : 2);
    if (ep.R) ep.R = (const char*)ep.R + roff * (ep.r_dt == COMAT_F32 ? 4 : 2);
    PlainLoader<T, BM, TA> al;
    PlainLoader<T, BN, TB> bl;
    al.init(A, g.lda, m0, g.M, g.K);
    bl.init(B, g.ldb, n0, g.N, g.K);
    gemm_block<T, BM, BN>(al, bl, g.K, m0, n0, g.M, g.N, ep);
}

struct ConvArgs {
    const void* X;
    const void* W;
    ConvGeom geo;
    int64_t M, N, K;
    int tiles_m;
    Epi ep;
};

template <typename T, int BM, int BN> __global__ __launch_bounds__(NT) void conv_kernel(ConvArgs g) {
    const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    ConvLoader<T, BM> al;
    PlainLoader<T, BN, false> bl;
    al.init(g.X, g.geo, m0, g.M);
    bl.init(g.W, g.K, n0, g.N, g.K);
    gemm_block<T, BM, BN>(al, bl, g.K, m0, n0, g.M, g.N, g.ep);
}

template <typename T, int BM, int BN> int launch_gemm_t(const GemmArgs& g, int trans, dim3 grid, hipStream_t st) {
    switch (trans) {
        case 0: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, false>), grid, dim3(NT), 0, st, g); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true, false>), grid, dim3(NT), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, true>), grid, dim3(NT), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true, true>), grid, dim3(NT), 0, st, g); break;
    }
    return 0;
}

bool pick_big_tile(int64_t M, int64_t N, int64_t batch) {
    if (M <= 64 || N <= 64) return false;
    const int64_t tiles = cdiv64(M, 128) * cdiv64(N, 128) * batch;
    return tiles >= 192;
}

}  // namespace

extern "C" int comat_gemm(const comat_gemm_params* p, void* stream) {
    COMAT_REQUIRE(p != nullptr, "comat_gemm: null params");
    COMAT_REQUIRE(p->A && p->B && p->C, "comat_gemm: null operand");
    COMAT_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "comat_gemm: bad shape M=%ld N=%ld K=%ld", (long)p->M,
                  (long)p->N, (long)p->K);
    COMAT_REQUIRE(dtype_ok(p->in_dtype) && dtype_ok(p->out_dtype), "comat_gemm: bad dtype");
    COMAT_REQUIRE(!p->R || dtype_ok(p->r_dtype), "comat_gemm: bad residual dtype");
    COMAT_REQUIRE(p->batch1 >= 1 && p->batch2 >= 1 && p->batch1 * p->batch2 <= 65535, "comat_gemm: bad batch");
    COMAT_REQUIRE(!p->bias2 || p->rows_per_bias2 > 0, "comat_gemm: bias2 needs rows_per_bias2");
    COMAT_REQUIRE(p->lda >= (p->transA ? p->M : p->K) && p->ldb >= (p->transB ? p->N : p->K) && p->ldc >= p->N,
                  "comat_gemm: leading dimension too small");
    GemmArgs g;
    g.A = p->A; g.B = p->B;
    g.M = p->M; g.N = p->N; g.K = p->K; g.lda = p->lda; g.ldb = p->ldb;
    g.batch2 = p->batch2;
    g.sA1 = p->sA1; g.sA2 = p->sA2; g.sB1 = p->sB1; g.sB2 = p->sB2;
    g.sC1 = p->sC1; g.sC2 = p->sC2; g.sR1 = p->sR1; g.sR2 = p->sR2;
    g.ep.C = p->C; g.ep.bias = p->bias; g.ep.bias2 = p->bias2; g.ep.R = p->R;
    g.ep.ldc = p->ldc; g.ep.ldr = p->ldr; g.ep.rows_per_b2 = p->rows_per_bias2 > 0 ? p->rows_per_bias2 : 1;
    g.ep.alpha = p->alpha; g.ep.beta = p->beta; g.ep.act = p->act;
    g.ep.out_dt = p->out_dtype; g.ep.r_dt = p->r_dtype;
    const int64_t batch = p->batch1 * p->batch2;
    const bool big = pick_big_tile(p->M, p->N, batch);
    const int bm = big ? 128 : 64;
    g.tiles_m = (int)cdiv64(p->M, bm);
    const int64_t tiles = (int64_t)g.tiles_m * cdiv64(p->N, bm);
    COMAT_REQUIRE(tiles < (1ll << 31), "comat_gemm: too many tiles");
    dim3 grid((unsigned)tiles, 1, (unsigned)batch);
    hipStream_t st = (hipStream_t)stream;
    const int trans = (p->transA ? 1 : 0) | (p->transB ? 2 : 0);
    if (p->in_dtype == COMAT_BF16) {
        if (big) launch_gemm_t<bf16_t, 128, 128>(g, trans, grid, st);
        else launch_gemm_t<bf16_t, 64, 64>(g, trans, grid, st);
    } else {
        if (big) launch_gemm_t<float, 128, 128>(g, trans, grid, st);
        else launch_gemm_t<float, 64, 64>(g, trans, grid, st);
    }
    return comat_check_launch("comat_gemm");
}

extern "C" int comat_conv2d(const comat_conv_params* p, void* stream) {
    COMAT_REQUIRE(p != nullptr, "comat_conv2d: null params");
    COMAT_REQUIRE(p->X && p->W && p->Y, "comat_conv2d: null operand");
    COMAT_REQUIRE(p->B > 0 && p->Hin > 0 && p->Win > 0 && p->Cin > 0 && p->Hout > 0 && p->Wout > 0 && p->Cout > 0,
                  "comat_conv2d: bad shape");
    COMAT_REQUIRE(p->KH > 0 && p->KW > 0 && p->stride >= 1 && p->pad >= 0, "comat_conv2d: bad kernel geometry");
    COMAT_REQUIRE(p->mode == 0 || p->mode == 1, "comat_conv2d: mode must be 0 or 1");
    COMAT_REQUIRE(p->ups == 1 || (p->ups == 2 && p->mode == 0), "comat_conv2d: ups must be 1, or 2 with mode 0");
    COMAT_REQUIRE(dtype_ok(p->in_dtype) && dtype_ok(p->out_dtype), "comat_conv2d: bad dtype");
    COMAT_REQUIRE(!p->R || dtype_ok(p->r_dtype), "comat_conv2d: bad residual dtype");
    ConvArgs g;
    g.X = p->X; g.W = p->W;
    g.geo.B = p->B; g.geo.Hin = p->Hin; g.geo.Win = p->Win; g.geo.Cin = p->Cin;
    g.geo.Hout = p->Hout; g.geo.Wout = p->Wout; g.geo.KH = p->KH; g.geo.KW = p->KW;
    g.geo.stride = p->stride; g.geo.pad = p->pad; g.geo.mode = p->mode; g.geo.ups = p->ups;
    g.M = (int64_t)p->B * p->Hout * p->Wout;
    g.N = p->Cout;
    g.K = (int64_t)p->KH * p->KW * p->Cin;
    g.ep.C = p->Y; g.ep.bias = p->bias; g.ep.bias2 = p->bias2; g.ep.R = p->R;
    g.ep.ldc = p->Cout; g.ep.ldr = p->Cout; g.ep.rows_per_b2 = (int64_t)p->Hout * p->Wout;
    g.ep.alpha = p->alpha; g.ep.beta = p->beta; g.ep.act = p->act;
    g.ep.out_dt = p->out_dtype; g.ep.r_dt = p->r_dtype;
    const bool big = pick_big_tile(g.M, g.N, 1);
    const int bm = big ? 128 : 64;
    g.tiles_m = (int)cdiv64(g.M, bm);
    const int64_t tiles = (int64_t)g.tiles_m * cdiv64(g.N, bm);
    COMAT_REQUIRE(tiles < (1ll << 31), "comat_conv2d: too many tiles");
    dim3 grid((unsigned)tiles, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    if (p->in_dtype == COMAT_BF16) {
        if (big) hipLaunchKernelGGL((conv_kernel<bf16_t, 128, 128>), grid, dim3(NT), 0, st, g);
        else hipLaunchKernelGGL((conv_kernel<bf16_t, 64, 64>), grid, dim3(NT), 0, st, g);
    } else {
        if (big) hipLaunchKernelGGL((conv_kernel<float, 128, 128>), grid, dim3(NT), 0, st, g);
        else hipLaunchKernelGGL((conv_kernel<float, 64, 64>), grid, dim3(NT), 0, st, g);
    }
    return comat_check_launch("comat_conv2d");
}
