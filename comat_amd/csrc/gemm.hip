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
#include "gemm_shared.h"

namespace {

constexpr int NT = 256;
constexpr int KTB = 64;              // bytes of k per row per k-tile (128 B measured no better: fewer barriers but half the occupancy)
constexpr int ROWB = KTB + 16;       // bytes per LDS row, k-contiguous image (36 dwords: conflict-free b128 reads)
constexpr int CPR = KTB / 16;        // 16-byte chunks per row
constexpr int PF = 4;  // k-tiles kept in flight per thread (register prefetch ring): hides HBM/L2 latency when few
                        // workgroups share a CU

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
// All index arithmetic that does not change along k is done once in init(); load() only advances running
// pointers / tap counters by one k-tile (the k-loop of the small tiles is otherwise VALU-bound on address math).
template <typename T, int ROWS, bool TRANS, int NTH = NT> struct PlainLoader {
    static constexpr bool kTrans = TRANS;
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int BKE = KTB / sizeof(T);
    static constexpr int TOTAL = ROWS * CPR;             // 16-byte chunks of one k-tile image
    static constexpr int NCH = TOTAL >= NTH ? TOTAL / NTH : 1;
    static constexpr bool PARTIAL = TOTAL < NTH;          // more threads than chunks: the surplus threads idle here
    static_assert(TOTAL % NTH == 0 || TOTAL < NTH, "chunks must divide evenly over the block");
    static constexpr int VPR = ROWS / EPV;               // 16-byte vectors per k-row (k-major image)
    static constexpr int RS = ROWS * (int)sizeof(T) + 16;  // k-major LDS row stride in bytes
    const T* ptr[NCH];   // running source pointer of each 16-byte chunk
    int kpos[NCH];       // k index of the chunk's first element
    int rleft[NCH];      // non-trans: 1 if the row exists; trans: number of valid m-elements in the vector (0..EPV)
    int lds_off[NCH];
    int K;
    int64_t step;        // pointer advance per k-tile (elements)
    bool vec_ok;
    uint4 regs[PF][NCH];

    // `tid`: index of the thread among the NTH threads that fill this image (threadIdx.x unless the block holds
    // several wave groups with an image each)
    __device__ __forceinline__ void init(const void* p, int64_t ld, int64_t r0, int64_t rmax, int64_t k_, int64_t kt0,
                                         int tid = -1) {
        const T* P = (const T*)p;
        if (tid < 0) tid = threadIdx.x;
        K = (int)k_;
        vec_ok = ((ld % EPV) == 0) && ((((uintptr_t)p) & 15) == 0);
        step = TRANS ? (int64_t)BKE * ld : (int64_t)BKE;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * NTH;
            if (PARTIAL && c >= TOTAL) {  // no chunk for this thread
                kpos[i] = 0;
                rleft[i] = 0;
                ptr[i] = P;
                lds_off[i] = -1;
                continue;
            }
            if (!TRANS) {
                const int row = c / CPR, kv = c % CPR;
                const int64_t gr = r0 + row;
                kpos[i] = (int)(kt0 * BKE) + kv * EPV;
                rleft[i] = gr < rmax ? 1 : 0;
                ptr[i] = P + (gr < rmax ? gr : 0) * ld + kpos[i];
                lds_off[i] = row * ROWB + kv * 16;
            } else {
                const int kk = c / VPR, mv = c % VPR;
                const int64_t gr = r0 + (int64_t)mv * EPV;
                kpos[i] = (int)(kt0 * BKE) + kk;
                const int64_t left = rmax - gr;
                rleft[i] = left <= 0 ? 0 : (left >= EPV ? EPV : (int)left);
                ptr[i] = P + (int64_t)kpos[i] * ld + (left > 0 ? gr : 0);
                lds_off[i] = kk * RS + mv * 16;
            }
        }
    }
    // loads the current k-tile into register slot `slot`, then advances to the next k-tile
    __device__ __forceinline__ void load(int slot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Vec16 v;
            v.u = make_uint4(0, 0, 0, 0);
            const T* src = ptr[i];
            if (!TRANS) {
                if (rleft[i] && kpos[i] < K) {
                    if (vec_ok && kpos[i] + EPV <= K) {
                        v.u = *(const uint4*)src;
                    } else {
#pragma unroll
                        for (int e = 0; e < EPV; ++e)
                            if (kpos[i] + e < K) {
                                if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)src)[e];
                                else v.f[e] = ((const float*)src)[e];
                            }
                    }
                }
            } else {
                if (rleft[i] > 0 && kpos[i] < K) {
                    if (vec_ok && rleft[i] == EPV) {
                        v.u = *(const uint4*)src;
                    } else {
#pragma unroll
                        for (int e = 0; e < EPV; ++e)
                            if (e < rleft[i]) {
                                if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)src)[e];
                                else v.f[e] = ((const float*)src)[e];
                            }
                    }
                }
            }
            regs[slot][i] = v.u;
            ptr[i] += step;
            kpos[i] += BKE;
        }
    }
    __device__ __forceinline__ void store(char* lds, int slot) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (!PARTIAL || lds_off[i] >= 0) *(uint4*)(lds + lds_off[i]) = regs[slot][i];
    }
};

// K-segmented operand: C = sum_s A_s[M, K_s] B_s[N, K_s]^T with every A_s / B_s k-contiguous.  The k-tile sequence is
// the concatenation of the segments (each padded to whole k-tiles with zeros by the kpos < K mask), so LoRA's
// "frozen weight + low-rank branch" and the sum of several projections' data-gradients run as ONE contraction.
constexpr int MAXSEG = 8;
struct SegTable {
    const void* A[MAXSEG];
    const void* B[MAXSEG];
    int64_t lda[MAXSEG], ldb[MAXSEG];
    int64_t sA[MAXSEG], sB[MAXSEG];  // element stride of the operand between batch items (0: shared)
    int K[MAXSEG];
    int nseg;
};

// LIMIT: the loader delivers at most `total_left` k-tiles and zeros afterwards (a wave group's slice of the k-range)
template <typename T, int ROWS, bool LIMIT = false> struct SegLoader {
    static constexpr bool kTrans = false;
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int BKE = KTB / sizeof(T);
    static constexpr int NCH = ROWS * CPR / NT;
    const T* ptr[NCH];
    int kpos[NCH];
    int64_t grow[NCH];   // clamped global row of the chunk
    int rleft[NCH];
    int lds_off[NCH];
    int K, seg, tiles_left;
    int tid0;            // index of the thread among the NT threads that fill this image
    int64_t total_left;  // LIMIT only
    int64_t zb;          // batch item
    bool vec_ok, is_b;
    uint4 regs[PF][NCH];

    // segment `sg`, starting `toff` k-tiles into it.  The table is scanned with an unrolled select (uniform values:
    // scalar selects, no dynamic indexing of the kernel-argument struct)
    __device__ __forceinline__ void setup(const SegTable& t, int sg, int toff) {
        const void* P = nullptr;
        int64_t ld = 0;
        int k = 0;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s)
            if (s == sg) {
                P = is_b ? (const void*)((const T*)t.B[s] + zb * t.sB[s]) : (const void*)((const T*)t.A[s] + zb * t.sA[s]);
                ld = is_b ? t.ldb[s] : t.lda[s];
                k = t.K[s];
            }
        seg = sg;
        K = k;
        tiles_left = (k + BKE - 1) / BKE - toff;
        vec_ok = ((ld % EPV) == 0) && ((((uintptr_t)P) & 15) == 0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kv = ((LIMIT ? tid0 : (int)threadIdx.x) + i * NT) % CPR;
            kpos[i] = toff * BKE + kv * EPV;
            ptr[i] = (const T*)P + grow[i] * ld + kpos[i];
        }
    }
    __device__ __forceinline__ void init(const SegTable& t, bool b_operand, int64_t r0, int64_t rmax, int64_t kt0,
                                         int64_t z, int tid = -1, int64_t ntiles = 0) {
        is_b = b_operand;
        zb = z;
        if (LIMIT) {
            tid0 = tid < 0 ? (int)threadIdx.x : tid;
            total_left = ntiles;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (LIMIT ? tid0 : (int)threadIdx.x) + i * NT;
            const int row = c / CPR, kv = c % CPR;
            const int64_t gr = r0 + row;
            rleft[i] = gr < rmax ? 1 : 0;
            grow[i] = gr < rmax ? gr : 0;
            lds_off[i] = row * ROWB + kv * 16;
        }
        int sg = 0;
        int64_t t0 = kt0;
#pragma unroll
        for (int s = 0; s < MAXSEG; ++s) {
            const int64_t nt = (t.K[s] + BKE - 1) / BKE;
            if (s == sg && s + 1 < t.nseg && t0 >= nt) {
                t0 -= nt;
                ++sg;
            }
        }
        setup(t, sg, (int)t0);
    }
    __device__ __forceinline__ void load(const SegTable& t, int slot) {
        if (tiles_left <= 0 && seg + 1 < t.nseg) setup(t, seg + 1, 0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Vec16 v;
            v.u = make_uint4(0, 0, 0, 0);
            const T* src = ptr[i];
            if (rleft[i] && kpos[i] < K && (!LIMIT || total_left > 0)) {
                if (vec_ok && kpos[i] + EPV <= K) {
                    v.u = *(const uint4*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; ++e)
                        if (kpos[i] + e < K) {
                            if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)src)[e];
                            else v.f[e] = ((const float*)src)[e];
                        }
                }
            }
            regs[slot][i] = v.u;
            ptr[i] += BKE;
            kpos[i] += BKE;
        }
        --tiles_left;
        if (LIMIT) --total_left;
    }
    __device__ __forceinline__ void store(char* lds, int slot) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *(uint4*)(lds + lds_off[i]) = regs[slot][i];
    }
};

// adapter: gives a SegLoader the load(slot) interface of the other loaders
template <typename T, int ROWS, bool LIMIT = false> struct SegLoaderRef {
    static constexpr bool kTrans = false;
    SegLoader<T, ROWS, LIMIT> l;
    const SegTable* t;
    __device__ __forceinline__ void load(int slot) { l.load(*t, slot); }
    __device__ __forceinline__ void store(char* lds, int slot) const { l.store(lds, slot); }
};

struct ConvGeom {
    int B, Hin, Win, Cin, Hout, Wout, KH, KW, stride, pad, mode, ups;
};

// im2col gather of a channels-last activation: row m = (b, oy, ox), k = (ky, kx, ci).  Per chunk the output position
// is decoded once; along k a running (ky, kx, ci) counter replaces the divisions.
template <typename T, int ROWS, int NTH = NT, bool LIMIT = false> struct ConvLoader {
    static constexpr bool kTrans = false;
    static constexpr int EPV = 16 / sizeof(T);
    static constexpr int BKE = KTB / sizeof(T);
    static constexpr int NCH = ROWS * CPR / NTH;
    static_assert(ROWS * CPR % NTH == 0 && NCH >= 1, "conv A operand: chunks must divide evenly over the block");
    const T* X;
    ConvGeom g;
    bool vec_ok;
    int base_y[NCH], base_x[NCH], brow[NCH];  // oy*stride - pad (mode 0) or oy - pad (mode 1); b*Hin
    int ky[NCH], kx[NCH], ci[NCH];            // running tap / channel of the chunk's first element
    bool rvalid[NCH];
    int tid0;         // index of the thread among the NTH threads that fill this image
    int tiles_left;   // LIMIT: k-tiles this loader may still deliver (zeros afterwards)
    uint4 regs[PF][NCH];

    __device__ __forceinline__ void init(const void* x, const ConvGeom& g_, int64_t r0, int64_t M, int64_t kt0,
                                         int tid = -1, int ntiles = 0x7fffffff) {
        X = (const T*)x;
        g = g_;
        tid0 = tid < 0 ? (int)threadIdx.x : tid;
        tiles_left = ntiles;
        vec_ok = ((g.Cin % EPV) == 0) && ((((uintptr_t)x) & 15) == 0);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid0 + i * NTH;
            const int64_t gm = r0 + (c / CPR);
            rvalid[i] = gm < M;
            const int hw = g.Hout * g.Wout;
            const int b = (int)(gm / hw), rem = (int)(gm - (int64_t)b * hw);
            const int oy = rem / g.Wout, ox = rem - oy * g.Wout;
            brow[i] = b * g.Hin;
            base_y[i] = (g.mode == 0 ? oy * g.stride : oy) - g.pad;
            base_x[i] = (g.mode == 0 ? ox * g.stride : ox) - g.pad;
            const int64_t k = kt0 * BKE + (c % CPR) * EPV;
            const int tap = (int)(k / g.Cin);
            ci[i] = (int)(k - (int64_t)tap * g.Cin);
            ky[i] = tap / g.KW;
            kx[i] = tap - ky[i] * g.KW;
        }
    }
    // element offset of X[b, sy, sx, ci] for the tap, or -1 when it falls into padding / between strided samples
    __device__ __forceinline__ int src_off(int i, int kyy, int kxx, int cii) const {
        int sy = base_y[i] + kyy, sx = base_x[i] + kxx;
        if (g.mode == 0) {
            if (sy < 0 || sx < 0 || sy >= g.Hin * g.ups || sx >= g.Win * g.ups) return -1;
            if (g.ups == 2) {
                sy >>= 1;
                sx >>= 1;
            }
        } else {
            if (sy < 0 || sx < 0) return -1;
            if (g.stride == 2) {
                if ((sy | sx) & 1) return -1;
                sy >>= 1;
                sx >>= 1;
            } else if (g.stride > 1) {
                if ((sy % g.stride) != 0 || (sx % g.stride) != 0) return -1;
                sy /= g.stride;
                sx /= g.stride;
            }
            if (sy >= g.Hin || sx >= g.Win) return -1;
        }
        return ((brow[i] + sy) * g.Win + sx) * g.Cin + cii;
    }
    __device__ __forceinline__ void load(int slot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            Vec16 v;
            v.u = make_uint4(0, 0, 0, 0);
            if (rvalid[i] && ky[i] < g.KH && (!LIMIT || tiles_left > 0)) {
                if (vec_ok) {
                    const int off = src_off(i, ky[i], kx[i], ci[i]);
                    if (off >= 0) v.u = *(const uint4*)(X + off);
                } else {
                    int ky2 = ky[i], kx2 = kx[i], ci2 = ci[i];
#pragma unroll
                    for (int e = 0; e < EPV; ++e) {
                        if (ky2 < g.KH) {
                            const int off = src_off(i, ky2, kx2, ci2);
                            if (off >= 0) {
                                if (sizeof(T) == 2) v.h[e] = ((const bf16_t*)X)[off];
                                else v.f[e] = ((const float*)X)[off];
                            }
                        }
                        if (++ci2 == g.Cin) {
                            ci2 = 0;
                            if (++kx2 == g.KW) {
                                kx2 = 0;
                                ++ky2;
                            }
                        }
                    }
                }
            }
            regs[slot][i] = v.u;
            ci[i] += BKE;  // advance one k-tile
            while (ci[i] >= g.Cin) {
                ci[i] -= g.Cin;
                if (++kx[i] == g.KW) {
                    kx[i] = 0;
                    ++ky[i];
                }
            }
        }
        if (LIMIT) --tiles_left;
    }
    __device__ __forceinline__ void store(char* lds, int slot) const {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid0 + i * NTH;
            *(uint4*)(lds + (c / CPR) * ROWB + (c % CPR) * 16) = regs[slot][i];
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
        const int ke0 = s * (32 / (int)sizeof(T)) + h * EPV;  // k-step s covers 32 bytes of k
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
// split-K context of one block: slice `sp` of `splits` of output tile `tile` (linear over batch x tiles_m x tiles_n)
struct SplitCtx {
    float* slabs;       // [splits][ntiles][4 quads][NT threads] float4
    unsigned* counter;  // this tile's ticket counter
    int splits, sp;
    int64_t tile, ntiles;
};

template <typename T, typename AL, typename BL>
__device__ __forceinline__ void gemm_block(AL& al, BL& bl, int64_t kt0, int64_t kt1, int64_t m0, int64_t n0, int64_t M,
                                           int64_t N, const Epi& ep, const SplitCtx& sk) {
    // 4 waves as 2 x 2, one 32x32 MFMA tile per wave
    typedef typename FragOf<T>::type F;
    constexpr int OPB = 64 * ROWB;  // bytes per operand per stage (>= any image)
    __shared__ __attribute__((aligned(16))) char smem[2][2][OPB];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    // prologue: PF k-tiles in flight, the first one staged to LDS
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kt0 + u < kt1) {
            al.load(u);
            bl.load(u);
        }
    al.store(smem[0][0], 0);
    bl.store(smem[0][1], 0);
    __syncthreads();
    int cur = 0;
    for (int64_t kt = kt0; kt < kt1; kt += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int64_t t = kt + u;  // tile t sits in LDS[cur]; register slot u (its old home) is free
            if (t < kt1) {
                if (t + PF < kt1) {
                    al.load(u);
                    bl.load(u);
                }
                const char* la = smem[cur][0];
                const char* lb = smem[cur][1];
#pragma unroll
                for (int s = 0; s < KTB / 32; ++s) {
                    const F af = read_frag<T, 64, AL::kTrans>(la, wr * 32 + r, s, h);
                    const F bfr = read_frag<T, 64, BL::kTrans>(lb, wc * 32 + r, s, h);
                    mma(acc, af, bfr);
                }
                if (t + 1 < kt1) {
                    al.store(smem[cur ^ 1][0], (u + 1) % PF);
                    bl.store(smem[cur ^ 1][1], (u + 1) % PF);
                }
                __syncthreads();
                cur ^= 1;
            }
        }
    }

    if (sk.splits > 1) {
        // split-K: park the fp32 partial tile (lane-linear float4 slabs), take a ticket; the last arriver of the tile
        // adds the slices in slice order and finishes (gemm_shared.h) - no reduce launch
        const SlabIO io(sk.slabs);
        const int64_t mine = (((int64_t)sk.sp * sk.ntiles + sk.tile) * 4 * NT + threadIdx.x) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_t v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            io.store(mine + (int64_t)q * NT * 16, v);
        }
        if (!splitk_ticket_is_last(sk.counter, sk.splits, (unsigned*)&smem[0][0][0])) return;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        for (int s2 = 0; s2 < sk.splits; ++s2) {
            const int64_t src = (((int64_t)s2 * sk.ntiles + sk.tile) * 4 * NT + threadIdx.x) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t v = io.load(src + (int64_t)q * NT * 16);
                acc[4 * q] += v[0];
                acc[4 * q + 1] += v[1];
                acc[4 * q + 2] += v[2];
                acc[4 * q + 3] += v[3];
            }
        }
    }
    // epilogue: v = act(alpha*acc + bias + bias2) + beta*R
    const int64_t col = n0 + wc * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t row = m0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M && col < N) {
            float v = ep.alpha * acc[i];
            if (ep.bias) v += ep.bias[col];
            if (ep.bias2) v += ep.bias2[(row / ep.rows_per_b2) * N + col];
            if (ep.act == COMAT_ACT_SILU) v = silu_f(v);
            else if (ep.act == COMAT_ACT_GELU) v = gelu_f(v);
            if (ep.R) v += ep.beta * ld_dt(ep.R, row * ep.ldr + col, ep.r_dt);
            st_dt(ep.C, row * ep.ldc + col, v, ep.out_dt);
        }
    }
}

struct GemmArgs {
    const void* A;
    const void* B;
    int64_t M, N, K, lda, ldb;
    int64_t batch2, sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
    int tiles_m, tiles_n;
    int splits;
    int64_t ntiles;  // batch * tiles_m * tiles_n
    float* ws;       // [counters | slabs]
    Epi ep;
};

// k-tile range of split s
__device__ __forceinline__ void split_range(int64_t nk, int splits, int s, int64_t& kt0, int64_t& kt1) {
    const int64_t per = (nk + splits - 1) / splits;
    kt0 = (int64_t)s * per;
    kt1 = kt0 + per < nk ? kt0 + per : nk;
    if (kt0 > nk) kt0 = nk;
}

__device__ __forceinline__ SplitCtx make_split(float* ws, int splits, int sp, int64_t tile, int64_t ntiles) {
    SplitCtx sk;
    sk.slabs = ws + WS_COUNTERS;
    sk.counter = (unsigned*)ws + tile;
    sk.splits = splits;
    sk.sp = sp;
    sk.tile = tile;
    sk.ntiles = ntiles;
    return sk;
}

template <typename T, bool TA, bool TB> __global__ __launch_bounds__(NT) void gemm_kernel(GemmArgs g) {
    // linear work id = ((z * tiles_m + tm) * tiles_n + tn) * splits + split
    int64_t lin = xcd_chunk_map(blockIdx.x, gridDim.x);
    const int sp = (int)(lin % g.splits);
    lin /= g.splits;
    const int64_t tile = lin;
    const int tn = (int)(lin % g.tiles_n);
    lin /= g.tiles_n;
    const int tm = (int)(lin % g.tiles_m);
    const int64_t z = lin / g.tiles_m, b1 = z / g.batch2, b2 = z - b1 * g.batch2;
    constexpr int BKE = KTB / (int)sizeof(T);
    int64_t kt0, kt1;
    split_range((g.K + BKE - 1) / BKE, g.splits, sp, kt0, kt1);
    const int64_t m0 = (int64_t)tm * 64, n0 = (int64_t)tn * 64;
    const T* A = (const T*)g.A + b1 * g.sA1 + b2 * g.sA2;
    const T* B = (const T*)g.B + b1 * g.sB1 + b2 * g.sB2;
    Epi ep = g.ep;
    const int64_t coff = b1 * g.sC1 + b2 * g.sC2, roff = b1 * g.sR1 + b2 * g.sR2;
    ep.C = (char*)ep.C + coff * (ep.out_dt == COMAT_F32 ? 4 : 2);
    if (ep.R) ep.R = (const char*)ep.R + roff * (ep.r_dt == COMAT_F32 ? 4 : 2);
    PlainLoader<T, 64, TA> al;
    PlainLoader<T, 64, TB> bl;
    al.init(A, g.lda, m0, g.M, g.K, kt0);
    bl.init(B, g.ldb, n0, g.N, g.K, kt0);
    gemm_block<T>(al, bl, kt0, kt1, m0, n0, g.M, g.N, ep, make_split(g.ws, g.splits, sp, tile, g.ntiles));
}

struct GemmSegArgs {
    SegTable t;
    int64_t M, N, nk;
    int64_t sC, sR, sBias;  // batch strides (elements) of C, R and bias
    int tiles_m, tiles_n, splits;
    int64_t ntiles;
    float* ws;
    Epi ep;
};

template <typename T> __global__ __launch_bounds__(NT) void gemm_seg_kernel(GemmSegArgs g) {
    int64_t lin = xcd_chunk_map(blockIdx.x, gridDim.x);
    const int sp = (int)(lin % g.splits);
    lin /= g.splits;
    const int64_t tile = lin;
    const int tn = (int)(lin % g.tiles_n);
    lin /= g.tiles_n;
    const int tm = (int)(lin % g.tiles_m);
    const int64_t z = lin / g.tiles_m;
    int64_t kt0, kt1;
    split_range(g.nk, g.splits, sp, kt0, kt1);
    const int64_t m0 = (int64_t)tm * 64, n0 = (int64_t)tn * 64;
    Epi ep = g.ep;
    ep.C = (char*)ep.C + z * g.sC * (ep.out_dt == COMAT_F32 ? 4 : 2);
    if (ep.R) ep.R = (const char*)ep.R + z * g.sR * (ep.r_dt == COMAT_F32 ? 4 : 2);
    if (ep.bias) ep.bias += z * g.sBias;
    SegLoaderRef<T, 64> al, bl;
    al.t = bl.t = &g.t;
    al.l.init(g.t, false, m0, g.M, kt0, z);
    bl.l.init(g.t, true, n0, g.N, kt0, z);
    gemm_block<T>(al, bl, kt0, kt1, m0, n0, g.M, g.N, ep, make_split(g.ws, g.splits, sp, tile, g.ntiles));
}

struct ConvArgs {
    const void* X;
    const void* W;
    ConvGeom geo;
    int64_t M, N, K;
    int tiles_m, tiles_n;
    int splits;
    int64_t ntiles;
    float* ws;
    Epi ep;
};

template <typename T> __global__ __launch_bounds__(NT) void conv_kernel(ConvArgs g) {
    int64_t lin = xcd_chunk_map(blockIdx.x, gridDim.x);
    const int sp = (int)(lin % g.splits);
    lin /= g.splits;
    const int64_t tile = lin;
    const int tn = (int)(lin % g.tiles_n), tm = (int)(lin / g.tiles_n);
    const int64_t m0 = (int64_t)tm * 64, n0 = (int64_t)tn * 64;
    constexpr int BKE = KTB / (int)sizeof(T);
    int64_t kt0, kt1;
    split_range((g.K + BKE - 1) / BKE, g.splits, sp, kt0, kt1);
    ConvLoader<T, 64> al;
    PlainLoader<T, 64, false> bl;
    al.init(g.X, g.geo, m0, g.M, kt0);
    bl.init(g.W, g.K, n0, g.N, g.K, kt0);
    gemm_block<T>(al, bl, kt0, kt1, m0, n0, g.M, g.N, g.ep, make_split(g.ws, g.splits, sp, tile, g.ntiles));
}

template <typename T> void launch_gemm_t(const GemmArgs& g, int trans, dim3 grid, hipStream_t st) {
    switch (trans) {
        case 0: hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(NT), 0, st, g); break;
        case 1: hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(NT), 0, st, g); break;
        case 2: hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(NT), 0, st, g); break;
        default: hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(NT), 0, st, g); break;
    }
}

// Split-K choice for the 64x64 tile kernel.  These problems are short on tiles (M <= 16384, N <= 1280): prefer enough
// workgroups to put several on every CU (the k-loop is latency-bound otherwise), >= 8 k-tiles per split so that the
// slab traffic stays small; bounded by the caller's workspace (slabs) and by the ticket counters (one per tile).
int plan_splits(int64_t M, int64_t N, int64_t K, int bke, int64_t batch, int64_t ws_bytes) {
    const int force = comat_option(COMAT_OPT_FORCE_SPLITS);  // tuning knob (tools/mb_gemm2.py)
    const int64_t ntiles = cdiv64(M, 64) * cdiv64(N, 64) * batch;
    const int64_t nk = cdiv64(K, bke);
    const int64_t slab_bytes = ws_bytes - COMAT_WS_COUNTER_BYTES;
    if (slab_bytes <= 0 || ntiles > WS_COUNTERS) return 1;
    const int64_t cap = slab_bytes / (ntiles * 64 * 64 * 4);
    int64_t s = 1;
    if (force > 0) s = force;
    else if (ntiles < 768 && nk >= 16) {
        s = cdiv64(1024, ntiles);
        if (s > nk / 8) s = nk / 8;
    }
    if (s > cap) s = cap;
    if (s > nk) s = nk;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

}  // namespace

extern "C" int64_t comat_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t batch, int32_t in_dtype) {
    // counters + the slabs of the largest split the planners would pick for this problem (both kernels: <= 64 slices of
    // the padded fp32 output)
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return COMAT_WS_COUNTER_BYTES;
    const int bke = in_dtype == COMAT_BF16 ? KTB / 2 : KTB / 4;
    const int64_t ntiles = cdiv64(M, 64) * cdiv64(N, 64) * batch;
    const int64_t nk = cdiv64(K, bke);
    int64_t s = 1;
    if (ntiles < 768 && nk >= 16) {
        s = cdiv64(1024, ntiles);
        if (s > nk / 8) s = nk / 8;
        if (s > 64) s = 64;
        if (s < 1) s = 1;
    }
    const int64_t pm = cdiv64(M, 128) * 128, pn = cdiv64(N, 128) * 128;
    return COMAT_WS_COUNTER_BYTES + s * batch * pm * pn * 4;
}

extern "C" int comat_gemm(const comat_gemm_params* p, void* stream) {
    COMAT_REQUIRE(p != nullptr, "comat_gemm: null params");
    COMAT_REQUIRE(p->A && p->B && (p->C || p->epi2 == 2), "comat_gemm: null operand");
    COMAT_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, "comat_gemm: bad shape M=%ld N=%ld K=%ld", (long)p->M,
                  (long)p->N, (long)p->K);
    COMAT_REQUIRE((dtype_ok(p->in_dtype) || p->in_dtype == COMAT_FP8_E4M3) && dtype_ok(p->out_dtype), "comat_gemm: bad dtype");
    COMAT_REQUIRE(!p->R || dtype_ok(p->r_dtype), "comat_gemm: bad residual dtype");
    COMAT_REQUIRE(p->batch1 >= 1 && p->batch2 >= 1 && p->batch1 * p->batch2 <= 65535, "comat_gemm: bad batch");
    COMAT_REQUIRE(!p->bias2 || p->rows_per_bias2 > 0, "comat_gemm: bias2 needs rows_per_bias2");
    COMAT_REQUIRE(p->K < (1ll << 30), "comat_gemm: K too large");
    COMAT_REQUIRE(p->lda >= (p->transA ? p->M : p->K) && p->ldb >= (p->transB ? p->N : p->K) &&
                      p->ldc >= p->N - (p->epi2 == 4 ? p->n2 : 0),  // (tail columns, epi2 = 4: C holds the first N - n2 columns)
                  "comat_gemm: leading dimension too small");
    COMAT_REQUIRE(p->epi2 == 0 || ((p->C2 != nullptr || (p->q8 != nullptr && p->epi2 <= 2)) && p->epi2 >= 1 && p->epi2 <= 4),
                  "comat_gemm: bad second epilogue");
    if (p->epi2 == 4)
        COMAT_REQUIRE(p->B2 != nullptr && p->n2 > 0 && p->n2 < p->N && p->ldc2 >= p->n2 && !p->transA && !p->transB && p->batch2 == 1,
                      "comat_gemm: tail columns (epi2 = 4) need B2, 0 < n2 < N, ldc2 >= n2, k-contiguous operands, one batch level");
    if (p->epi2 == 0 && p->in_dtype == COMAT_BF16 && !p->transA && !p->transB && p->batch2 == 1) {  // lean kernel first (option gemm3)
        const comat_gemm_segment one = {p->A, p->B, p->K, p->lda, p->ldb, p->sA1, p->sB1};
        const int rc3 = comat_gemm3_try(p, &one, 1, false, stream);
        if (rc3 > 0) {
            comat_note_gemm_kernel(rc3);
            return comat_check_launch("comat_gemm");
        }
    }
    const int rc2 = comat_gemm2_try_gemm(p, stream);
    comat_note_gemm_kernel(rc2 > 0 ? rc2 : 0);
    if (rc2) return rc2 < 0 ? rc2 : comat_check_launch("comat_gemm");
    COMAT_REQUIRE(!p->q8, "comat_gemm: the e4m3 output of the GEGLU epilogue (q8) exists on the pipelined kernel only");
    if (p->epi2 == 4) {  // tail columns outside the pipelined kernel: the two products one after the other, same results
        comat_gemm_params q = *p;
        q.epi2 = 0; q.C2 = nullptr; q.B2 = nullptr; q.n2 = 0;
        q.N = p->N - p->n2;
        int rc = comat_gemm(&q, stream);
        if (rc) return rc;
        q.B = p->B2; q.sB1 = p->sB2_tail; q.N = p->n2;
        q.C = p->C2; q.ldc = p->ldc2; q.sC1 = p->sC2_tail;
        q.bias = nullptr; q.bias2 = nullptr; q.R = nullptr; q.act = COMAT_ACT_NONE; q.alpha = p->alpha2; q.beta = 0.0f;
        return comat_gemm(&q, stream);
    }
    if (p->epi2 == 3) {  // GEGLU backward epilogue, two-launch form: dF into the workspace, then the elementwise kernel
        COMAT_REQUIRE(p->in_dtype == COMAT_BF16 && p->out_dtype == COMAT_BF16 && p->N % 16 == 0 && p->batch1 * p->batch2 == 1 &&
                          !p->R && !p->bias && !p->bias2 && p->act == COMAT_ACT_NONE && p->ldc == 2 * p->N && p->ldc2 == 2 * p->N,
                      "comat_gemm: the GEGLU backward epilogue needs bf16 operands and output, N %% 16 == 0, ldc == ldc2 == 2 N, nothing else fused");
        const int64_t need = p->M * p->N * 2;
        COMAT_REQUIRE(p->ws && p->ws_bytes >= COMAT_WS_COUNTER_BYTES + need,
                      "comat_gemm: the two-launch form of the GEGLU backward epilogue needs M * N * 2 bytes of workspace behind the counters");
        comat_gemm_params q = *p;
        q.epi2 = 0;
        q.C2 = nullptr;
        q.C = (char*)p->ws + COMAT_WS_COUNTER_BYTES;
        q.ldc = p->N;
        q.ws = nullptr;
        q.ws_bytes = 0;
        const int rc = comat_gemm(&q, stream);
        if (rc) return rc;
        return comat_geglu_il_bwd(q.C, p->C2, p->C, p->M, (int32_t)p->N, COMAT_BF16, stream);
    }
    if (p->epi2 != 0) {
        // The pipelined kernel declined (option gemm2 = 0, or a shape it does not take): the GEGLU epilogue is a fast path, not
        // a different function - compute the same thing as two launches: the plain product into C (epi2 == 2: into
        // the tail of the caller's workspace, unsplit), then the interleaved-layout GEGLU kernel into C2.
        COMAT_REQUIRE(p->in_dtype == COMAT_BF16 && p->out_dtype == COMAT_BF16 && p->N % 32 == 0 && p->batch1 * p->batch2 == 1 &&
                          !p->R && !p->bias2 && p->act == COMAT_ACT_NONE && p->ldc2 == p->N / 2,
                      "comat_gemm: the GEGLU epilogue needs bf16 operands and output, N %% 32 == 0, ldc2 == N / 2, no residual / bias2 / "
                      "activation / batch");
        comat_gemm_params q = *p;
        q.epi2 = 0;
        q.C2 = nullptr;
        if (p->epi2 == 2 || p->ldc != p->N) {
            const int64_t need = p->M * p->N * 2;
            COMAT_REQUIRE(p->ws && p->ws_bytes >= COMAT_WS_COUNTER_BYTES + need,
                          "comat_gemm: the two-launch form of the GEGLU epilogue needs M * N * 2 bytes of workspace behind the counters");
            q.C = (char*)p->ws + COMAT_WS_COUNTER_BYTES;
            q.ldc = p->N;
            q.ws = nullptr;  // the scratch lives where split-K slabs would: run unsplit
            q.ws_bytes = 0;
        }
        const int rc = comat_gemm(&q, stream);
        if (rc) return rc;
        if (p->epi2 == 1 && q.C != p->C) {  // (pre-activations wanted in a strided C: not a shape the step has)
            comat_set_error("comat_gemm: epi2 == 1 needs ldc == N outside the pipelined kernel");
            return COMAT_EUNSUPPORTED;
        }
        return comat_geglu_il_fwd(q.C, p->C2, p->M, (int32_t)(p->N / 2), COMAT_BF16, stream);
    }
    COMAT_REQUIRE(p->in_dtype != COMAT_FP8_E4M3,
                  "comat_gemm: fp8 operands need transA = transB = 0, K %% 64 == 0, 16-byte aligned rows, batch2 == 1");
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
    const bool bf = p->in_dtype == COMAT_BF16;
    const int bke = bf ? KTB / 2 : KTB / 4;
    g.tiles_m = (int)cdiv64(p->M, 64);
    g.tiles_n = (int)cdiv64(p->N, 64);
    g.ntiles = (int64_t)g.tiles_m * g.tiles_n * batch;
    g.splits = plan_splits(p->M, p->N, p->K, bke, batch, p->ws ? p->ws_bytes : 0);
    const int64_t blocks = g.ntiles * g.splits;
    COMAT_REQUIRE(blocks < (1ll << 31), "comat_gemm: too many tiles");
    g.ws = (float*)p->ws;
    dim3 grid((unsigned)blocks, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    const int trans = (p->transA ? 1 : 0) | (p->transB ? 2 : 0);
    if (bf) launch_gemm_t<bf16_t>(g, trans, grid, st);
    else launch_gemm_t<float>(g, trans, grid, st);
    return comat_check_launch("comat_gemm");
}

extern "C" int comat_gemm_segments(const comat_gemm_params* p, const comat_gemm_segment* segs, int32_t nseg,
                                   void* stream) {
    COMAT_REQUIRE(p != nullptr && segs != nullptr, "comat_gemm_segments: null params");
    COMAT_REQUIRE(nseg >= 1 && nseg <= MAXSEG, "comat_gemm_segments: 1..%d segments supported, got %d", MAXSEG, nseg);
    COMAT_REQUIRE(p->C && p->M > 0 && p->N > 0, "comat_gemm_segments: bad output / shape");
    COMAT_REQUIRE(dtype_ok(p->in_dtype) && dtype_ok(p->out_dtype), "comat_gemm_segments: bad dtype");
    COMAT_REQUIRE(!p->R || dtype_ok(p->r_dtype), "comat_gemm_segments: bad residual dtype");
    COMAT_REQUIRE(p->batch2 <= 1 && !p->transA && !p->transB && p->batch1 <= 65535,
                  "comat_gemm_segments: operands must be k-contiguous; one batch level (batch1)");
    const int64_t batch = p->batch1 > 1 ? p->batch1 : 1;
    COMAT_REQUIRE(batch == 1 || !p->bias2, "comat_gemm_segments: bias2 is not supported with a batch");
    COMAT_REQUIRE(!p->bias2 || p->rows_per_bias2 > 0, "comat_gemm_segments: bias2 needs rows_per_bias2");
    COMAT_REQUIRE(p->ldc >= p->N, "comat_gemm_segments: ldc too small");
    for (int s = 0; s < nseg; ++s) {
        COMAT_REQUIRE(segs[s].A && segs[s].B && segs[s].K > 0 && segs[s].K < (1ll << 30),
                      "comat_gemm_segments: bad segment %d", s);
        COMAT_REQUIRE(segs[s].lda >= segs[s].K && segs[s].ldb >= segs[s].K,
                      "comat_gemm_segments: leading dimension of segment %d too small", s);
    }
    const int rc3 = comat_gemm3_try(p, segs, nseg, true, stream);  // lean kernel first (option gemm3)
    if (rc3 > 0) {
        comat_note_gemm_kernel(rc3);
        return comat_check_launch("comat_gemm_segments");
    }
    const int rc2 = comat_gemm2_try_segments(p, segs, nseg, stream);
    comat_note_gemm_kernel(rc2 > 0 ? rc2 : 0);
    if (rc2) return rc2 < 0 ? rc2 : comat_check_launch("comat_gemm_segments");
    GemmSegArgs g;
    const int bke = p->in_dtype == COMAT_BF16 ? KTB / 2 : KTB / 4;
    g.nk = 0;
    for (int s = 0; s < MAXSEG; ++s) {
        if (s < nseg) {
            g.t.A[s] = segs[s].A; g.t.B[s] = segs[s].B;
            g.t.lda[s] = segs[s].lda; g.t.ldb[s] = segs[s].ldb; g.t.K[s] = (int)segs[s].K;
            g.t.sA[s] = segs[s].sA; g.t.sB[s] = segs[s].sB;
            g.nk += cdiv64(segs[s].K, bke);
        } else {
            g.t.A[s] = g.t.B[s] = nullptr;
            g.t.lda[s] = g.t.ldb[s] = 0; g.t.K[s] = 0;
            g.t.sA[s] = g.t.sB[s] = 0;
        }
    }
    g.t.nseg = nseg;
    g.M = p->M; g.N = p->N;
    g.ep.C = p->C; g.ep.bias = p->bias; g.ep.bias2 = p->bias2; g.ep.R = p->R;
    g.ep.ldc = p->ldc; g.ep.ldr = p->ldr; g.ep.rows_per_b2 = p->rows_per_bias2 > 0 ? p->rows_per_bias2 : 1;
    g.ep.alpha = p->alpha; g.ep.beta = p->beta; g.ep.act = p->act;
    g.ep.out_dt = p->out_dtype; g.ep.r_dt = p->r_dtype;
    g.sC = p->sC1; g.sR = p->sR1; g.sBias = p->bias ? p->N : 0;  // bias: [batch, N] when batched
    g.tiles_m = (int)cdiv64(p->M, 64);
    g.tiles_n = (int)cdiv64(p->N, 64);
    g.ntiles = (int64_t)g.tiles_m * g.tiles_n * batch;
    g.splits = plan_splits(p->M, p->N, g.nk * bke, bke, batch, p->ws ? p->ws_bytes : 0);
    const int64_t blocks = g.ntiles * g.splits;
    COMAT_REQUIRE(blocks < (1ll << 31), "comat_gemm_segments: too many tiles");
    g.ws = (float*)p->ws;
    hipStream_t st = (hipStream_t)stream;
    if (p->in_dtype == COMAT_BF16) hipLaunchKernelGGL((gemm_seg_kernel<bf16_t>), dim3((unsigned)blocks), dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((gemm_seg_kernel<float>), dim3((unsigned)blocks), dim3(NT), 0, st, g);
    return comat_check_launch("comat_gemm_segments");
}

extern "C" int comat_conv2d(const comat_conv_params* p, void* stream) {
    COMAT_REQUIRE(p != nullptr, "comat_conv2d: null params");
    COMAT_REQUIRE(p->X && p->W && p->Y, "comat_conv2d: null operand");
    COMAT_REQUIRE(p->B > 0 && p->Hin > 0 && p->Win > 0 && p->Cin > 0 && p->Hout > 0 && p->Wout > 0 && p->Cout > 0,
                  "comat_conv2d: bad shape");
    COMAT_REQUIRE(p->KH > 0 && p->KW > 0 && p->stride >= 1 && p->pad >= 0, "comat_conv2d: bad kernel geometry");
    COMAT_REQUIRE(p->mode == 0 || p->mode == 1, "comat_conv2d: mode must be 0 or 1");
    COMAT_REQUIRE((int64_t)p->B * p->Hin * p->Win * p->Cin < (1ll << 31) &&
                      (int64_t)p->KH * p->KW * p->Cin < (1ll << 30),
                  "comat_conv2d: input tensor too large for 32-bit gather offsets");
    COMAT_REQUIRE(p->ups == 1 || (p->ups == 2 && p->mode == 0), "comat_conv2d: ups must be 1, or 2 with mode 0");
    COMAT_REQUIRE((dtype_ok(p->in_dtype) || p->in_dtype == COMAT_FP8_E4M3) && dtype_ok(p->out_dtype), "comat_conv2d: bad dtype");
    COMAT_REQUIRE(!p->R || dtype_ok(p->r_dtype), "comat_conv2d: bad residual dtype");
    const int rc2 = comat_gemm2_try_conv(p, stream);
    comat_note_gemm_kernel(rc2 > 0 ? rc2 : 0);
    if (rc2) return rc2 < 0 ? rc2 : comat_check_launch("comat_conv2d");
    COMAT_REQUIRE(p->in_dtype != COMAT_FP8_E4M3, "comat_conv2d: fp8 operands need mode 0, Cin %% 64 == 0, 16-byte aligned tensors");
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
    const bool bf = p->in_dtype == COMAT_BF16;
    const int bke = bf ? KTB / 2 : KTB / 4;
    g.tiles_m = (int)cdiv64(g.M, 64);
    g.tiles_n = (int)cdiv64(g.N, 64);
    g.ntiles = (int64_t)g.tiles_m * g.tiles_n;
    g.splits = plan_splits(g.M, g.N, g.K, bke, 1, p->ws ? p->ws_bytes : 0);
    const int64_t blocks = g.ntiles * g.splits;
    COMAT_REQUIRE(blocks < (1ll << 31), "comat_conv2d: too many tiles");
    g.ws = (float*)p->ws;
    dim3 grid((unsigned)blocks, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    if (bf) hipLaunchKernelGGL((conv_kernel<bf16_t>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((conv_kernel<float>), grid, dim3(NT), 0, st, g);
    return comat_check_launch("comat_conv2d");
}
