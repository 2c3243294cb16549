// attention.hip — fused (flash-style) attention forward and backward on MFMA for gfx950.
//
// Used wherever the probability map itself is not needed (UNet self-attention: 4096^2 scores per head at 64x64 latents
// that the reference materialises through its patched Attention.forward, attn_utils/tc_attn_utils.py:126-146; BLIP
// ViT self-attention; cross-attention on steps that do not capture maps).  Scores never touch HBM: per 32-key tile the
// wave computes S^T = K Q^T with MFMA, does the online softmax in registers and feeds P^T straight back into MFMA.
//
// Transposed formulation (the wave64 trick): with S^T [keys x queries] in the 32x32 accumulator layout
// (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), every lane owns ONE query column, so the softmax
// statistics (running max m, sum l, LSE, D) are per-lane scalars and the row reductions are over the lane's own 16
// registers plus one xor-32 shuffle.  The same registers, converted to the storage type, are the B operand of the next
// MFMA (O^T += V^T P^T), because any k-permutation shared by the A and B fragments cancels in the sum: the A fragment
// (V^T, K^T, Q^T, dO^T) is gathered from the LDS tile with the identical key order.
//
// Kernels (T = bf16 -> v_mfma_f32_32x32x16_bf16, T = fp32 -> exact v_mfma_f32_32x32x2_f32; DMAX = padded head dim):
//   flash_fwd   : block = 4 waves x 32 queries, loops over 32-key tiles (K,V staged through LDS, register prefetch)
//   flash_dq    : same geometry as forward; D[q] = sum_d dO[q,d] O[q,d] (stored for flash_dkdv), dQ^T += K^T dS^T
//   flash_dkdv  : block = 4 waves x 32 keys, loops over 32-query tiles; dV^T += dO^T P, dK^T += Q^T dS
#include "common.h"
#include "gemm_shared.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int NT = 256;
// ---- build-time policy of the fused attention kernels (round 4, calls 12 - 15: same-box A/B of one build per setting,
// profiles/r04_l_mb_flash_ab.txt, r04_n_mb_flash_diet.txt; every switch can be overridden with -D for such a run) ----------
// COMAT_FLASH_W2: the bf16 backward kernels for head dims <= 64 hold 160 - 250 registers and the step's shapes give a CU at
//   most two of their 4-wave blocks - they never run more than two waves per SIMD.  Saying so (amdgpu_waves_per_eu(2, 2)) lets
//   the scheduler batch LDS fragment reads ahead of the MFMAs instead of cycling two fragment registers to stay inside a
//   three-waves budget nobody uses: backward 2 x 8 x 4096^2 d=40 265.9 -> 259.1 us, 1 x 8: 173.2 -> 159.1.  Not the forward
//   kernels (d = 64: 144 -> 163 us), not head dims > 64 or fp32 (already at 256 registers: the hint makes them spill).
#ifndef COMAT_FLASH_W2
#define COMAT_FLASH_W2 1
#endif
// COMAT_FLASH_FULL_TILES: tiles with every row inside the matrix come through buffer loads with scalar tile offsets
//   (TileMover::load_full): forward 109.3 -> 101.4 us, backward 270.2 -> 262.8 at 2 x 8 x 4096^2 d=40, every 2-tile shape gains.
#ifndef COMAT_FLASH_FULL_TILES
#define COMAT_FLASH_FULL_TILES 1
#endif
// COMAT_FLASH_SCALE_OUT: dS = P (dP - D) without the softmax scale; dQ and dK are multiplied by it once, when they are stored
//   (one multiply per score less in every backward iteration; the products see dS before instead of after the scaling):
//   backward d=40 262.8 -> 252.1 us, d=64 445 -> 433; the 96-wide kernels (d = 80) lose (83.3 -> 88.9) and keep the old form.
#ifndef COMAT_FLASH_SCALE_OUT
#define COMAT_FLASH_SCALE_OUT 1
#endif
constexpr bool flash_scale_out(int dmax) { return COMAT_FLASH_SCALE_OUT && dmax != 96; }
constexpr bool flash_two_waves(int dmax, int elem_bytes, bool bwd) { return COMAT_FLASH_W2 && bwd && dmax <= 64 && elem_bytes == 2; }
#define FLASH_OCC(DMAX, ESZ, BWD) \
    __attribute__((amdgpu_waves_per_eu(flash_two_waves(DMAX, ESZ, BWD) ? 2 : 1, flash_two_waves(DMAX, ESZ, BWD) ? 2 : 8)))

template <typename T> struct FragOf;
template <> struct FragOf<bf16_t> { typedef short8_t type; };
template <> struct FragOf<float> { typedef f32x4_t type; };

union V16 {
    uint4 u;
    float f[4];
    bf16_t h[8];
};

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

// combine a per-lane value with the value of the lane 32 away (the other half of the 32x32 accumulator's columns):
// v_permlane32_swap hands each lane {own, partner} in one VALU instruction (a __shfl_xor(.., 32) goes through LDS)
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// two fp32 values in a register pair: v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 do two lanes' worth of the softmax / dS
// arithmetic per instruction (the loops are VALU-bound: tools/isa_loop.py).  Component-wise IEEE, i.e. the same bits as the
// scalar forms.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pair_of(const f32x16_t& v, int i) { return f32x2_t{v[i], v[i + 1]}; }
__device__ __forceinline__ f32x2_t splat2(float x) { return f32x2_t{x, x}; }
__device__ __forceinline__ f32x2_t exp2_fast2(f32x2_t x) { return f32x2_t{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
// exp(x) for x given in log2 units (v_exp_f32 is 2^x: no multiply in front of it)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// accumulator register i of lane-half hh  <->  row index inside the 32x32 tile
__device__ __forceinline__ int crow(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

template <typename T, int DMAX> struct Geo {
    static constexpr int KC = 16 / sizeof(T);             // elements per 16-byte chunk
    static constexpr int NKS = DMAX / (2 * KC);           // MFMA k-steps over the head dim
    static constexpr int NJ = 32 / (2 * KC);              // MFMA k-steps over a 32-row tile
    static constexpr int NT32 = DMAX / 32;                // 32-wide output tiles over the head dim
    static constexpr int RS = DMAX * (int)sizeof(T) + 16;  // LDS row stride in bytes (16-B pad)
    static constexpr int CPRW = DMAX * (int)sizeof(T) / 16;
    static constexpr int NCHK = (32 * CPRW + NT - 1) / NT;
    static constexpr int TILE_BYTES = 32 * RS;
    // transposed image [DMAX head-dim columns][32 tile rows] (flag TR of the kernels): the k-major fragments become
    // aligned 8 / 16-byte reads of 4 consecutive tile rows instead of KC scalar reads.  TRS: bytes per column; the pad
    // spreads the 32 lanes of a half-wave (consecutive columns) over all banks (bf16: 18 dwords, fp32: 36 dwords)
    static constexpr int TRS = 32 * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);
    static constexpr int TT_BYTES = DMAX * TRS;
};

// static LDS bytes of the backward bodies (flash_dq_body / flash_dkdv_body and their two-tile forms): the kernels that run one
// body, and the kernel that runs either (flash_bwd_kernel), size their buffer from here
template <typename T, int DMAX, bool TR> struct BwdLds {
    typedef Geo<T, DMAX> G;
    static constexpr bool SWT = TR && sizeof(T) != 2;  // software-transposed tile images (fp32 parity mode)
    static constexpr int DQ_ONE = 2 * G::TILE_BYTES + (SWT ? G::TT_BYTES : 0);
    static constexpr int DQ = 2 * DQ_ONE <= 65536 ? 2 * DQ_ONE : DQ_ONE;
    static constexpr int DKDV_ONE = 2 * G::TILE_BYTES + 256 + (SWT ? 2 * G::TT_BYTES : 0);
    static constexpr int DKDV = 2 * DKDV_ONE <= 65536 ? 2 * DKDV_ONE : DKDV_ONE;
    static constexpr int DQ2 = 8 * G::TILE_BYTES;            // two double-buffered [K | V] tile pairs
    static constexpr int DKDV2 = 8 * G::TILE_BYTES + 1024;   // two double-buffered [Q | dO | lse | D] tile pairs
};

// Every global load issued so far has landed.  Placed (unconditionally) at the end of a kernel's prologue: the prologue's
// own waits sit inside divergent branches (threads that copy no tile chunk skip them), so without this the compiler must
// assume the Q / K / V fragment loads may still be in flight at the loop head and puts an `s_waitcnt vmcnt(0)` in front
// of the loop's first MFMA - behind the NEXT tile's loads, which the loop has just issued: every iteration then pays a
// full global-load latency (round 4, tools/isa_loop.py --order).  A builtin, not inline asm: the waitcnt pass reads it.
__device__ __forceinline__ void loads_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), nothing else

// resource descriptor of one (batch, head) slab [rows x d] with row stride ld: raw (untyped) access, every byte up to the
// end of the last row's head columns in range (the slab must stay below 2 GiB: comat_flash_attn_* check it)
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const T* base, int rows, int64_t ld, int d) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((int64_t)(rows - 1) * ld + d) * (int64_t)sizeof(T)), 0x00020000);
}

// cooperative [32 rows x d] tile copy global -> registers -> LDS (zero padded to DMAX columns / missing rows)
template <typename T, int DMAX> struct TileMover {
    typedef Geo<T, DMAX> G;
    uint4 regs[G::NCHK];
    __device__ __forceinline__ void load(const T* base, int64_t ld, int row0, int nrows, int d) {
#pragma unroll
        for (int i = 0; i < G::NCHK; ++i) {
            const int c = threadIdx.x + i * NT;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < 32 * G::CPRW) {
                const int rr = c / G::CPRW, col = (c % G::CPRW) * G::KC;
                if (row0 + rr < nrows && col < d) v = *(const uint4*)(base + (int64_t)(row0 + rr) * ld + col);
            }
            regs[i] = v;
        }
    }
    // Full tiles (every row inside the matrix - all but the last tile of a loop) come through BUFFER loads: the operand's
    // (batch, head) slab is one resource descriptor in SGPRs, the tile's byte offset a scalar, and a thread's chunk offset
    // inside a tile never changes (prepare(), once per kernel and operand).  Lanes that hold no data - columns beyond d -
    // carry an offset beyond the descriptor's size and read zeros from the range check.  Per tile and iteration: one load
    // instruction, no 64-bit pointer arithmetic, no compare / select / re-zero, no divergent region (they were ~20 of the
    // ~150 VALU instructions of a 64-key forward iteration, and four exec-mask branches).
    unsigned voff[G::NCHK];
    __device__ __forceinline__ void prepare(int64_t ld, int d) {
#pragma unroll
        for (int i = 0; i < G::NCHK; ++i) {
            const int c = threadIdx.x + i * NT;
            const int rr = c / G::CPRW, col = (c % G::CPRW) * G::KC;
            voff[i] = (c < 32 * G::CPRW && col < d) ? (unsigned)((rr * ld + col) * (int64_t)sizeof(T)) : 0x80000000u;
        }
    }
    __device__ __forceinline__ void load_full(__amdgpu_buffer_rsrc_t slab, unsigned tile_off) {
#pragma unroll
        for (int i = 0; i < G::NCHK; ++i) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(slab, voff[i], tile_off, 0);
            __builtin_memcpy(&regs[i], &v, 16);
        }
    }
    __device__ __forceinline__ void store(char* lds) const {
#pragma unroll
        for (int i = 0; i < G::NCHK; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < 32 * G::CPRW) *(uint4*)(lds + (c / G::CPRW) * G::RS + (c % G::CPRW) * 16) = regs[i];
        }
    }
    // transposed image: element (tile row rr, column col) -> ldsT[col * TRS + rr * sizeof(T)]
    __device__ __forceinline__ void store_t(char* ldsT) const {
#pragma unroll
        for (int i = 0; i < G::NCHK; ++i) {
            const int c = threadIdx.x + i * NT;
            if (c < 32 * G::CPRW) {
                const int rr = c / G::CPRW, col0 = (c % G::CPRW) * G::KC;
                V16 v;
                v.u = regs[i];
#pragma unroll
                for (int e = 0; e < G::KC; ++e) {
                    char* p = ldsT + (col0 + e) * G::TRS + rr * (int)sizeof(T);
                    if (sizeof(T) == 2) *(bf16_t*)p = v.h[e];
                    else *(float*)p = v.f[e];
                }
            }
        }
    }
};

// fragment with rows = tile rows, k = head-dim chunk (k-contiguous 16-byte read)
template <typename T, int DMAX>
__device__ __forceinline__ typename FragOf<T>::type frag_kc(const char* lds, int row, int s, int hh) {
    return *(const typename FragOf<T>::type*)(lds + row * Geo<T, DMAX>::RS + s * 32 + hh * 16);
}
// fragment with rows = head-dim index n, k = tile rows in accumulator order (gathered, k-major)
template <typename T, int DMAX>
__device__ __forceinline__ typename FragOf<T>::type frag_km(const char* lds, int n, int j, int hh) {
    constexpr int KC = Geo<T, DMAX>::KC;
    V16 v;
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        const char* p = lds + crow(KC * j + e, hh) * Geo<T, DMAX>::RS + n * (int)sizeof(T);
        if (sizeof(T) == 2) v.h[e] = *(const bf16_t*)p;
        else v.f[e] = *(const float*)p;
    }
    typename FragOf<T>::type out;
    __builtin_memcpy(&out, &v, 16);
    return out;
}
// the same fragment from the transposed image: accumulator rows KC*j .. KC*j+KC-1 are runs of 4 consecutive tile rows
// (bf16: rows 16j+4hh+{0..3} and 16j+8+4hh+{0..3}; fp32: rows 8j+4hh+{0..3}) -> two 8-byte reads / one 16-byte read
template <typename T, int DMAX>
__device__ __forceinline__ typename FragOf<T>::type frag_km_t(const char* ldsT, int n, int j, int hh) {
    typedef Geo<T, DMAX> G;
    const char* col = ldsT + n * G::TRS;
    typename FragOf<T>::type out;
    if (sizeof(T) == 2) {
        const uint2 lo = *(const uint2*)(col + (16 * j + 4 * hh) * 2);
        const uint2 hi = *(const uint2*)(col + (16 * j + 8 + 4 * hh) * 2);
        const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        __builtin_memcpy(&out, &v, 16);
    } else {
        const uint4 v = *(const uint4*)(col + (8 * j + 4 * hh) * 4);
        __builtin_memcpy(&out, &v, 16);
    }
    return out;
}
// accumulator registers KC*j .. KC*j+KC-1 as a fragment of the storage type
template <typename T> __device__ __forceinline__ typename FragOf<T>::type pack_acc(const f32x16_t& a, int j) {
    constexpr int KC = 16 / sizeof(T);
    V16 v;
#pragma unroll
    for (int e = 0; e < KC; ++e) {
        if (sizeof(T) == 2) v.h[e] = f32_to_bf16(a[KC * j + e]);
        else v.f[e] = a[KC * j + e];
    }
    typename FragOf<T>::type out;
    __builtin_memcpy(&out, &v, 16);
    return out;
}
// ---- k-major bf16 fragments through the hardware transpose read (ds_read_b64_tr_b16, gfx950) ------------------------------
// The k-major operand tiles (V in the forward, K in dQ, Q and dO in dK/dV) are stored as they come from memory - rows =
// keys / queries, head dim contiguous - and the MFMA A fragment "8 consecutive k of head-dim row n" is exactly what the
// transpose read hands out (semantics probed on gfx950: tools/probes/tr_read_probe.hip): inside a
// 16-lane group, lane 4k'+q supplies the address of 4 consecutive elements (columns 4q..4q+3) of row k', and lane i
// receives column i of those 4 rows.  With rows = tile rows (the contraction index) and columns = head-dim indices, two
// reads give lane (r, hh) the tile rows 16j + 4hh + {0..3} and 16j + 8 + 4hh + {0..3} of head-dim column n = 32 t2 + r:
// the accumulator order crow() that pack_acc() uses for the other operand.  No transposed LDS image (rounds 1-2 built one
// with 8 two-byte LDS writes per 16-byte chunk), no gathers: every head dim takes this path.
// The compiler does not count asm loads (cdna_hip_programming.md 5.7, form ii): all reads of a phase are issued first,
// then every fragment passes through a wait statement that names its registers before the MFMA consumes it.
struct TrF {
    unsigned long long lo, hi;
};
__device__ __forceinline__ unsigned lds_addr32(const char* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF> __device__ __forceinline__ void tr_read(unsigned long long& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
// per-lane part of the address: row 4 hh + k' of the k-step, columns 16 * (second 16-lane group) + 4 q
template <int RS> __device__ __forceinline__ unsigned tr_lane_off(int lane) {
    return (unsigned)((4 * (lane >> 5) + ((lane & 15) >> 2)) * RS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
}
// the NT32 fragments (32-wide head-dim tiles t2) of k-step J (tile rows 16 J .. 16 J + 15) of one tile
template <int RS, int J, int... T2>
__device__ __forceinline__ void tr_issue_j(unsigned addr, TrF* f, std::integer_sequence<int, T2...>) {
    ((tr_read<J * 16 * RS + T2 * 64>(f[T2].lo, addr), tr_read<J * 16 * RS + T2 * 64 + 8 * RS>(f[T2].hi, addr)), ...);
}
__device__ __forceinline__ short8_t tr_take(TrF& f) {  // wait for the LDS queue, then hand the fragment to the MFMA
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi));
    short8_t out;
    __builtin_memcpy(&out, &f, 16);
    return out;
}

// per-lane fragment (column = this lane's row of the global matrix, chunk 2s+hh of the head dim) straight from HBM
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS>
__device__ __forceinline__ void load_col_frags(typename FragOf<T>::type* f, const T* base, int64_t ld, int row,
                                               int nrows, int d, int hh) {
    typedef Geo<T, DMAX> G;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int col = (2 * s + hh) * G::KC;
        V16 v;
        v.u = make_uint4(0, 0, 0, 0);
        if (row < nrows && col < d) v.u = *(const uint4*)(base + (int64_t)row * ld + col);
        __builtin_memcpy(&f[s], &v, 16);
    }
}

struct FlashArgs {
    const void *Q, *K, *V, *O, *dO;
    void *Out, *dQ, *dK, *dV;
    float* lse;
    float* Dbuf;
    int B, H, Nq, Nk, d;
    int64_t ldq, ldk, ldv, ldo;
    float scale;
    int xcd;       // option flash_xcd: renumber the workgroups so that the blocks of one (batch, head) share an XCD
    int qsplit;    // dK/dV: number of query ranges (blockIdx.z) whose fp32 partials are summed by flash_kv_reduce
    float* part;   // [2][qsplit][B*H][Nk][d] fp32 partial dK / dV (qsplit > 1)
    // forward, fp8 forward with delayed scaling (comat_flash_attn_fwd_q): the e4m3 bytes of the (rounded) output for the projection that
    // consumes it - layout of Out with leading dimension ldq8 - and its abs-max folded into *q_amax
    unsigned char* q8;
    const float* q_scale;
    unsigned* q_amax;
    int64_t ldq8;
};

// NK: MFMA k-steps over the head dim actually issued (< Geo::NKS when the padded tail chunks are all zero)
// TR: the k-major operand (V here) is staged as a TRANSPOSED LDS image and read with aligned 8 / 16-byte loads
// Hardware hands consecutive workgroups (x fastest, then y) to the 8 XCDs round-robin, so the blocks of one (batch, head)
// - which all stream that head's K and V (Q and dO in dK/dV) - land on all 8 private L2s and each fetches the operands once:
// 84 MB modelled, 89 MB measured for 2 x 8 heads 4 096^2 d = 40 against 21 MB algorithmic.  With
// option flash_xcd the linear workgroup id goes through the chunk map of the GEMM kernels (gemm_shared.h): an XCD then owns a
// contiguous, head-major range of blocks.  Which block computes which tile changes, nothing else: bit-identical.
__device__ __forceinline__ void flash_block_xy(int xcd, int& bx, int& by) {
    bx = blockIdx.x;
    by = blockIdx.y;
    if (xcd) {
        const int64_t lin = xcd_chunk_map((int64_t)blockIdx.y * gridDim.x + blockIdx.x, (int64_t)gridDim.x * gridDim.y);
        bx = (int)(lin % gridDim.x);
        by = (int)(lin / gridDim.x);
    }
}

// fp8 forward, delayed scaling: the forward kernels also emit the e4m3 bytes of their output.  Accumulator registers 4 j .. 4 j + 3 of a
// lane are 4 consecutive head-dim columns of its query row (crow): 4 bytes per store, made from the ROUNDED output values - what
// comat_fp8_quantize_scaled would make of Out - and one abs-max per wave for the consumer's site.  Every lane of the wave calls this
// (rows beyond Nq contribute nothing).
template <typename T, int NT32>
__device__ __forceinline__ void flash_store_q8(const FlashArgs& a, const f32x16_t (&oT)[NT32], float inv, int q, int hh, int b, int h) {
    const float qinv = 1.0f / *a.q_scale;
    float qmax = 0.f;
    if (q < a.Nq) {
        unsigned char* qb = a.q8 + ((int64_t)b * a.Nq + q) * a.ldq8 + h * a.d;
#pragma unroll
        for (int t2 = 0; t2 < NT32; ++t2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n0 = t2 * 32 + crow(4 * j, hh);
                if (n0 < a.d) {
                    float r[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float o = oT[t2][4 * j + e] * inv;
                        r[e] = sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(o)) : o;
                        qmax = fmaxf(qmax, fabsf(r[e]));
                    }
                    *(unsigned*)(qb + n0) = fp8_pack4(r, qinv);
                }
            }
    }
    qmax = wave_max(qmax);
    if ((threadIdx.x & 63) == 0) fp8_amax_track(a.q_amax, qmax);
}

template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, sizeof(T), false) void flash_fwd_kernel(FlashArgs a) {
    typedef Geo<T, DMAX> G;
    typedef typename FragOf<T>::type F;
    // LDS: [K tile | V tile (TR: its transposed image)], TWICE when it fits (DB): the next tile is written into the other
    // buffer right after this tile's MFMAs, so a key tile costs ONE block barrier instead of two
    constexpr bool HW = TR && sizeof(T) == 2;  // bf16: hardware transpose reads from the plain tile
    constexpr bool SWT = TR && !HW;            // fp32: software-built transposed image
    constexpr int ONE = G::TILE_BYTES + (SWT ? G::TT_BYTES : G::TILE_BYTES);
    constexpr bool DB = 2 * ONE <= 65536;
    __shared__ __attribute__((aligned(16))) char smem[DB ? 2 * ONE : ONE];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    char* Kt = smem;
    char* Vt = smem + G::TILE_BYTES;  // SWT: the transposed image of the V tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    T* Ob = (T*)a.Out + (int64_t)b * a.Nq * a.ldo + h * a.d;
    const int q = bx * 128 + wave * 32 + r;

    F qf[NK];
    load_col_frags<T, DMAX, NK>(qf, Qb, a.ldq, q, a.Nq, a.d, hh);
    f32x16_t oT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oT[t][i] = 0.f;
    float m = -INFINITY, l = 0.f;  // running max of the RAW scores, running sum of exp(scale * (s - m))
    const float c2 = a.scale * LOG2E;

    TileMover<T, DMAX> km, vm;
    const int ntiles = (a.Nk + 31) / 32;
    km.load(Kb, a.ldk, 0, a.Nk, a.d);
    vm.load(Vb, a.ldv, 0, a.Nk, a.d);
    km.store(Kt);
    if (SWT) vm.store_t(Vt);
    else vm.store(Vt);
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) {
            km.load(Kb, a.ldk, (t + 1) * 32, a.Nk, a.d);
            vm.load(Vb, a.ldv, (t + 1) * 32, a.Nk, a.d);
        }
        f32x16_t st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) mma(st, frag_kc<T, DMAX>(Kt, r, s, hh), qf[s]);
        // online softmax on the RAW scores (c2 = scale * log2 e > 0 keeps their order): 4 VALU ops per score - max,
        // fma + exp2, sum.  Keys beyond Nk exist only in the last tile (uniform branch).
        if ((t + 1) * 32 > a.Nk) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t * 32 + crow(i, hh) >= a.Nk) st[i] = -INFINITY;
        }
        float mt = st[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(mt, st[i]);
        mt = half_max(mt);
        const float m_new = fmaxf(m, mt);  // raw units
        const float neg = -m_new * c2;
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = exp2_fast(__builtin_fmaf(st[i], c2, neg));
            st[i] = p;
            ps += p;
        }
        ps = half_sum(ps);
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {  // some query's maximum moved: rescale (rare after the first tiles)
            const float alpha = exp2_fast((m - m_new) * c2);
            l *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) oT[t2][i] *= alpha;
        }
        l += ps;
        m = m_new;
        if constexpr (HW) {
            const unsigned va = lds_addr32(Vt) + tr_off;
            auto step = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                TrF vf[G::NT32];
                tr_issue_j<G::RS, j>(va, vf, std::make_integer_sequence<int, G::NT32>{});
                const F pb = pack_acc<T>(st, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) mma(oT[t2], tr_take(vf[t2]), pb);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const F pb = pack_acc<T>(st, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) mma(oT[t2], SWT ? frag_km_t<T, DMAX>(Vt, t2 * 32 + r, j, hh) : frag_km<T, DMAX>(Vt, t2 * 32 + r, j, hh), pb);
            }
        }
        if (DB) {
            // the other buffer was last read in iteration t-1, which every wave left through the barrier below
            Kt = smem + ((t + 1) & 1) * ONE;
            Vt = Kt + G::TILE_BYTES;
        } else {
            __syncthreads();
        }
        if (more) {
            km.store(Kt);
            if (SWT) vm.store_t(Vt);
            else vm.store(Vt);
        }
        if (DB || more) __syncthreads();
    }
    if (a.q8) flash_store_q8<T, G::NT32>(a, oT, 1.0f / l, q, hh, b, h);
    if (q < a.Nq) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = t2 * 32 + crow(i, hh);
                if (n < a.d) stf<T>(Ob + (int64_t)q * a.ldo + n, oT[t2][i] * inv);
            }
        if (hh == 0) a.lse[((int64_t)by) * a.Nq + q] = m * a.scale + __logf(l);  // natural-log units
    }
}

// Forward with TWO 32-key tiles per iteration (bf16, hardware transpose reads; option flash_kt = 2): one block barrier, one
// running-max update and one rescale decision per 64 keys, and the two score tiles are independent MFMA chains whose
// results the softmax of the other can overlap.  Same arithmetic per score as flash_fwd_kernel; the running maximum moves
// at 64-key granularity, so results agree with the 32-key kernel to rounding, not bit for bit.
template <int DMAX, int NK>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, 2, false) void flash_fwd2_kernel(FlashArgs a) {
    typedef bf16_t T;
    typedef Geo<T, DMAX> G;
    typedef short8_t F;
    constexpr int ONE = 2 * G::TILE_BYTES;  // [K tile | V tile] of one 32-key tile
    constexpr int PAIR = 2 * ONE;
    static_assert(2 * PAIR <= 65536, "two double-buffered tile pairs must fit the static LDS limit");
    __shared__ __attribute__((aligned(16))) char smem[2 * PAIR];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    char* cur = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    T* Ob = (T*)a.Out + (int64_t)b * a.Nq * a.ldo + h * a.d;
    const int q = bx * 128 + wave * 32 + r;
    F qf[NK];
    load_col_frags<T, DMAX, NK>(qf, Qb, a.ldq, q, a.Nq, a.d, hh);
    f32x16_t oT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) oT[t][i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c2 = a.scale * LOG2E;
    TileMover<T, DMAX> km0, vm0, km1, vm1;
    const int npairs = (a.Nk + 63) / 64;
    const __amdgpu_buffer_rsrc_t k_slab = slab_rsrc(Kb, a.Nk, a.ldk, a.d), v_slab = slab_rsrc(Vb, a.Nk, a.ldv, a.d);
    km0.prepare(a.ldk, a.d);
    vm0.prepare(a.ldv, a.d);
    km1.prepare(a.ldk, a.d);
    vm1.prepare(a.ldv, a.d);
    auto load_pair = [&](int p) {
        if (COMAT_FLASH_FULL_TILES && (p + 1) * 64 <= a.Nk) {
            const unsigned ko = (unsigned)(p * 64 * a.ldk * (int64_t)sizeof(T)), vo = (unsigned)(p * 64 * a.ldv * (int64_t)sizeof(T));
            km0.load_full(k_slab, ko);
            vm0.load_full(v_slab, vo);
            km1.load_full(k_slab, ko + (unsigned)(32 * a.ldk * (int64_t)sizeof(T)));
            vm1.load_full(v_slab, vo + (unsigned)(32 * a.ldv * (int64_t)sizeof(T)));
            return;
        }
        // rows beyond Nk come back as zeros (TileMover), their scores are masked below
        km0.load(Kb, a.ldk, p * 64, a.Nk, a.d);
        vm0.load(Vb, a.ldv, p * 64, a.Nk, a.d);
        km1.load(Kb, a.ldk, p * 64 + 32, a.Nk, a.d);
        vm1.load(Vb, a.ldv, p * 64 + 32, a.Nk, a.d);
    };
    auto store_pair = [&](char* dst) {
        km0.store(dst);
        vm0.store(dst + G::TILE_BYTES);
        km1.store(dst + ONE);
        vm1.store(dst + ONE + G::TILE_BYTES);
    };
    load_pair(0);
    store_pair(cur);
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int p = 0; p < npairs; ++p) {
        const bool more = p + 1 < npairs;
        if (more) load_pair(p + 1);
        f32x16_t s0, s1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            mma(s0, frag_kc<T, DMAX>(cur, r, s, hh), qf[s]);
            mma(s1, frag_kc<T, DMAX>(cur + ONE, r, s, hh), qf[s]);
        }
        if ((p + 1) * 64 > a.Nk) {  // keys beyond Nk exist only in the last pair
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (p * 64 + crow(i, hh) >= a.Nk) s0[i] = -INFINITY;
                if (p * 64 + 32 + crow(i, hh) >= a.Nk) s1[i] = -INFINITY;
            }
        }
        float mt = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mt = fmaxf(fmaxf(mt, s0[i]), s1[i]);  // one v_max3_f32 each (max is exact: any order)
        mt = half_max(mt);
        const float m_new = fmaxf(m, mt);
        const float neg = -m_new * c2;
        const f32x2_t c2v = splat2(c2), negv = splat2(neg);
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2_t p0 = exp2_fast2(__builtin_elementwise_fma(pair_of(s0, i), c2v, negv));
            const f32x2_t p1 = exp2_fast2(__builtin_elementwise_fma(pair_of(s1, i), c2v, negv));
            s0[i] = p0.x; s0[i + 1] = p0.y;
            s1[i] = p1.x; s1[i + 1] = p1.y;
            const f32x2_t t = p0 + p1;  // the pair sums of rows i, i + 1; added to the running sum in row order, as before
            ps += t.x;
            ps += t.y;
        }
        ps = half_sum(ps);
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {
            const float alpha = exp2_fast((m - m_new) * c2);
            l *= alpha;
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) oT[t2][i] *= alpha;
        }
        l += ps;
        m = m_new;
        const unsigned va0 = lds_addr32(cur + G::TILE_BYTES) + tr_off, va1 = lds_addr32(cur + ONE + G::TILE_BYTES) + tr_off;
        auto step = [&](auto jc, unsigned va, const f32x16_t& st) {
            constexpr int j = decltype(jc)::value;
            TrF vf[G::NT32];
            tr_issue_j<G::RS, j>(va, vf, std::make_integer_sequence<int, G::NT32>{});
            const F pb = pack_acc<T>(st, j);
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2) mma(oT[t2], tr_take(vf[t2]), pb);
        };
        step(std::integral_constant<int, 0>{}, va0, s0);
        step(std::integral_constant<int, 1>{}, va0, s0);
        step(std::integral_constant<int, 0>{}, va1, s1);
        step(std::integral_constant<int, 1>{}, va1, s1);
        cur = smem + ((p + 1) & 1) * PAIR;  // last read in iteration p-1, which every wave left through the barrier below
        if (more) store_pair(cur);
        __syncthreads();
    }
    if (a.q8) flash_store_q8<T, G::NT32>(a, oT, 1.0f / l, q, hh, b, h);
    if (q < a.Nq) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = t2 * 32 + crow(i, hh);
                if (n < a.d) stf<T>(Ob + (int64_t)q * a.ldo + n, oT[t2][i] * inv);
            }
        if (hh == 0) a.lse[((int64_t)by) * a.Nq + q] = m * a.scale + __logf(l);
    }
}

// NK: MFMA k-steps over the head dim actually issued (< Geo::NKS when the padded tail chunks are all zero)
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false, bool WRITE_D = true>
__device__ __forceinline__ void flash_dq_body(const FlashArgs& a, char* smem, int bx, int by) {
    constexpr bool SCALE_OUT = flash_scale_out(DMAX);
    typedef Geo<T, DMAX> G;
    typedef typename FragOf<T>::type F;
    constexpr bool HW = TR && sizeof(T) == 2;  // bf16: hardware transpose reads from the plain K tile
    constexpr bool SWT = TR && !HW;
    constexpr int ONE = 2 * G::TILE_BYTES + (SWT ? G::TT_BYTES : 0);
    constexpr bool DB = 2 * ONE <= 65536;  // double-buffered tiles: one barrier per key tile (see flash_fwd_kernel)
    char* Kt = smem;
    char* Vt = smem + G::TILE_BYTES;
    char* KtT = smem + 2 * G::TILE_BYTES;  // TR: transposed image of the K tile (for dQ^T += K^T dS^T)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const T* Gb = (const T*)a.dO + (int64_t)b * a.Nq * a.ldo + h * a.d;
    T* dQb = (T*)a.dQ + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const int q = bx * 128 + wave * 32 + r;
    F qf[NK], gf[NK];
    load_col_frags<T, DMAX, NK>(qf, Qb, a.ldq, q, a.Nq, a.d, hh);
    load_col_frags<T, DMAX, NK>(gf, Gb, a.ldo, q, a.Nq, a.d, hh);
    const float lse_q = (q < a.Nq ? a.lse[(int64_t)by * a.Nq + q] : 0.f) * LOG2E;  // log2 units
    // D[q] = sum_d dO[q, d] O[q, d]: the lane already holds its half of row q of dO; O comes in the same fragments.  Fixed
    // order (chunks ascending, then the two halves), written for the dK/dV pass that follows on the stream: the separate
    // "prep" launch of round 1 (260 launches per C2 step) is gone.
    float D_q;
    {
        const T* Ob = (const T*)a.O + (int64_t)b * a.Nq * a.ldo + h * a.d;
        F of[NK];
        load_col_frags<T, DMAX, NK>(of, Ob, a.ldo, q, a.Nq, a.d, hh);
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            V16 gv, ov;
            __builtin_memcpy(&gv, &gf[s], 16);
            __builtin_memcpy(&ov, &of[s], 16);
#pragma unroll
            for (int e = 0; e < G::KC; ++e)
                part += sizeof(T) == 2 ? bf16_to_f32(gv.h[e]) * bf16_to_f32(ov.h[e]) : gv.f[e] * ov.f[e];
        }
        D_q = half_sum(part);
        if (WRITE_D && q < a.Nq && hh == 0) a.Dbuf[(int64_t)by * a.Nq + q] = D_q;
    }
    const float c2 = a.scale * LOG2E;
    f32x16_t dqT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqT[t][i] = 0.f;
    TileMover<T, DMAX> km, vm;
    const int ntiles = (a.Nk + 31) / 32;
    km.load(Kb, a.ldk, 0, a.Nk, a.d);
    vm.load(Vb, a.ldv, 0, a.Nk, a.d);
    km.store(Kt);
    if (SWT) km.store_t(KtT);
    vm.store(Vt);
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) {
            km.load(Kb, a.ldk, (t + 1) * 32, a.Nk, a.d);
            vm.load(Vb, a.ldv, (t + 1) * 32, a.Nk, a.d);
        }
        f32x16_t st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            mma(st, frag_kc<T, DMAX>(Kt, r, s, hh), qf[s]);
            mma(dp, frag_kc<T, DMAX>(Vt, r, s, hh), gf[s]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = exp2_fast(__builtin_fmaf(st[i], c2, -lse_q));
            st[i] = SCALE_OUT ? p * (dp[i] - D_q) : p * a.scale * (dp[i] - D_q);
        }
        if ((t + 1) * 32 > a.Nk) {  // keys beyond Nk exist only in the last tile
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (t * 32 + crow(i, hh) >= a.Nk) st[i] = 0.f;
        }
        if constexpr (HW) {
            const unsigned ka = lds_addr32(Kt) + tr_off;
            auto step = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                TrF kfr[G::NT32];
                tr_issue_j<G::RS, j>(ka, kfr, std::make_integer_sequence<int, G::NT32>{});
                const F db = pack_acc<T>(st, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) mma(dqT[t2], tr_take(kfr[t2]), db);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const F db = pack_acc<T>(st, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) mma(dqT[t2], SWT ? frag_km_t<T, DMAX>(KtT, t2 * 32 + r, j, hh) : frag_km<T, DMAX>(Kt, t2 * 32 + r, j, hh), db);
            }
        }
        if (DB) {
            Kt = smem + ((t + 1) & 1) * ONE;
            Vt = Kt + G::TILE_BYTES;
            KtT = Kt + 2 * G::TILE_BYTES;
        } else {
            __syncthreads();
        }
        if (more) {
            km.store(Kt);
            if (SWT) km.store_t(KtT);
            vm.store(Vt);
        }
        if (DB || more) __syncthreads();
    }
    if (q < a.Nq) {
#pragma unroll
        for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = t2 * 32 + crow(i, hh);
                if (n < a.d) stf<T>(dQb + (int64_t)q * a.ldq + n, SCALE_OUT ? dqT[t2][i] * a.scale : dqT[t2][i]);
            }
    }
}
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, sizeof(T), true) void flash_dq_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[BwdLds<T, DMAX, TR>::DQ];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    flash_dq_body<T, DMAX, NK, TR>(a, smem, bx, by);
}

// NK: MFMA k-steps over the head dim actually issued (< Geo::NKS when the padded tail chunks are all zero)
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
__device__ __forceinline__ void flash_dkdv_body(const FlashArgs& a, char* smem, int bx, int by, int bz) {
    constexpr bool SCALE_OUT = flash_scale_out(DMAX);
    typedef Geo<T, DMAX> G;
    typedef typename FragOf<T>::type F;
    constexpr bool HW = TR && sizeof(T) == 2;  // bf16: hardware transpose reads from the plain Q / dO tiles
    constexpr bool SWT = TR && !HW;
    constexpr int ONE = 2 * G::TILE_BYTES + 256 + (SWT ? 2 * G::TT_BYTES : 0);
    constexpr bool DB = 2 * ONE <= 65536;  // double-buffered tiles: one barrier per query tile (see flash_fwd_kernel)
    char* Qt = smem;
    char* Gt = smem + G::TILE_BYTES;
    float* lse_s = (float*)(smem + 2 * G::TILE_BYTES);
    float* D_s = lse_s + 32;
    char* QtT = smem + 2 * G::TILE_BYTES + 256;  // TR: transposed images of the Q and dO tiles (dK^T, dV^T products)
    char* GtT = QtT + G::TT_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const T* Gb = (const T*)a.dO + (int64_t)b * a.Nq * a.ldo + h * a.d;
    T* dKb = (T*)a.dK + (int64_t)b * a.Nk * a.ldk + h * a.d;
    T* dVb = (T*)a.dV + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const int key = bx * 128 + wave * 32 + r;
    F kf[NK], vf[NK];
    load_col_frags<T, DMAX, NK>(kf, Kb, a.ldk, key, a.Nk, a.d, hh);
    load_col_frags<T, DMAX, NK>(vf, Vb, a.ldv, key, a.Nk, a.d, hh);
    f32x16_t dkT[G::NT32], dvT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkT[t][i] = 0.f; dvT[t][i] = 0.f; }
    const float* lse_g = a.lse + (int64_t)by * a.Nq;
    const float* D_g = a.Dbuf + (int64_t)by * a.Nq;
    TileMover<T, DMAX> qm, gm;
    // query-tile range of this block: few keys (cross-attention: 77) leave one block per (batch, head), so the
    // query loop is cut into gridDim.z ranges whose partial sums a fixed-order reduce kernel adds up
    const int ntq = (a.Nq + 31) / 32;
    const int per = (ntq + a.qsplit - 1) / a.qsplit;
    const int tbeg = bz * per;
    const int ntiles = tbeg + per < ntq ? tbeg + per : ntq;
    float lse_r = 0.f, D_r = 0.f;  // staged by threads 0..31; lse goes to log2 units when it is STORED to LDS (a multiply
                                   // here would wait for the load right behind its issue)
    const float c2 = a.scale * LOG2E;
    qm.load(Qb, a.ldq, tbeg * 32, a.Nq, a.d);
    gm.load(Gb, a.ldo, tbeg * 32, a.Nq, a.d);
    if (threadIdx.x < 32) {
        const int qi = tbeg * 32 + threadIdx.x;
        lse_r = qi < a.Nq ? lse_g[qi] : 0.f;
        D_r = qi < a.Nq ? D_g[qi] : 0.f;
    }
    qm.store(Qt);
    gm.store(Gt);
    if (SWT) {
        qm.store_t(QtT);
        gm.store_t(GtT);
    }
    if (threadIdx.x < 32) { lse_s[threadIdx.x] = lse_r * LOG2E; D_s[threadIdx.x] = D_r; }
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int t = tbeg; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        if (more) {
            qm.load(Qb, a.ldq, (t + 1) * 32, a.Nq, a.d);
            gm.load(Gb, a.ldo, (t + 1) * 32, a.Nq, a.d);
            if (threadIdx.x < 32) {
                const int qi = (t + 1) * 32 + threadIdx.x;
                lse_r = qi < a.Nq ? lse_g[qi] : 0.f;
                D_r = qi < a.Nq ? D_g[qi] : 0.f;
            }
        }
        f32x16_t sc, dp;  // [query rows x key cols]
#pragma unroll
        for (int i = 0; i < 16; ++i) { sc[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            mma(sc, frag_kc<T, DMAX>(Qt, r, s, hh), kf[s]);
            mma(dp, frag_kc<T, DMAX>(Gt, r, s, hh), vf[s]);
        }
        // rows beyond Nq: last tile only; keys beyond Nk: last key block only - a BLOCK-uniform condition (full tiles skip
        // the compare / select instructions)
        const bool edge = (t + 1) * 32 > a.Nq || bx * 128 + 128 > a.Nk;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = exp2_fast(__builtin_fmaf(sc[i], c2, -lse_s[crow(i, hh)]));
        if (edge) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!((t * 32 + crow(i, hh) < a.Nq) && (key < a.Nk))) sc[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            dp[i] = SCALE_OUT ? sc[i] * (dp[i] - D_s[crow(i, hh)]) : sc[i] * a.scale * (dp[i] - D_s[crow(i, hh)]);
        if constexpr (HW) {
            const unsigned qa = lds_addr32(Qt) + tr_off, ga = lds_addr32(Gt) + tr_off;
            auto step = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                TrF gfr[G::NT32], qfr[G::NT32];
                tr_issue_j<G::RS, j>(ga, gfr, std::make_integer_sequence<int, G::NT32>{});
                tr_issue_j<G::RS, j>(qa, qfr, std::make_integer_sequence<int, G::NT32>{});
                const F pb = pack_acc<T>(sc, j);
                const F db = pack_acc<T>(dp, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) {
                    mma(dvT[t2], tr_take(gfr[t2]), pb);
                    mma(dkT[t2], tr_take(qfr[t2]), db);
                }
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
        } else {
#pragma unroll
            for (int j = 0; j < G::NJ; ++j) {
                const F pb = pack_acc<T>(sc, j);
                const F db = pack_acc<T>(dp, j);
#pragma unroll
                for (int t2 = 0; t2 < G::NT32; ++t2) {
                    mma(dvT[t2], SWT ? frag_km_t<T, DMAX>(GtT, t2 * 32 + r, j, hh) : frag_km<T, DMAX>(Gt, t2 * 32 + r, j, hh), pb);
                    mma(dkT[t2], SWT ? frag_km_t<T, DMAX>(QtT, t2 * 32 + r, j, hh) : frag_km<T, DMAX>(Qt, t2 * 32 + r, j, hh), db);
                }
            }
        }
        if (DB) {
            Qt = smem + ((t - tbeg + 1) & 1) * ONE;
            Gt = Qt + G::TILE_BYTES;
            lse_s = (float*)(Qt + 2 * G::TILE_BYTES);
            D_s = lse_s + 32;
            QtT = Qt + 2 * G::TILE_BYTES + 256;
            GtT = QtT + G::TT_BYTES;
        } else {
            __syncthreads();
        }
        if (more) {
            qm.store(Qt);
            gm.store(Gt);
            if (SWT) {
                qm.store_t(QtT);
                gm.store_t(GtT);
            }
            if (threadIdx.x < 32) { lse_s[threadIdx.x] = lse_r * LOG2E; D_s[threadIdx.x] = D_r; }
        }
        if (DB || more) __syncthreads();
    }
    if (key < a.Nk) {
        const int64_t slab = (int64_t)a.B * a.H * a.Nk * a.d;
        float* pk = a.qsplit > 1
                        ? a.part + ((int64_t)bz * a.B * a.H + by) * a.Nk * a.d + (int64_t)key * a.d
                        : nullptr;
        float* pv = a.qsplit > 1 ? pk + (int64_t)a.qsplit * slab : nullptr;
        if (a.qsplit > 1) {
            // fp32 partials of this query range: accumulator registers 4 j .. 4 j + 3 are 4 consecutive head-dim columns (crow), so a
            // lane writes 16-byte pieces instead of single floats (same values; d is a multiple of 4, the slabs are 16-byte aligned)
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n0 = t2 * 32 + crow(4 * j, hh);
                    if (n0 < a.d) {
                        const float sc = SCALE_OUT ? a.scale : 1.0f;
                        *(float4*)(pk + n0) = SCALE_OUT ? make_float4(dkT[t2][4 * j] * sc, dkT[t2][4 * j + 1] * sc, dkT[t2][4 * j + 2] * sc, dkT[t2][4 * j + 3] * sc)
                                                        : make_float4(dkT[t2][4 * j], dkT[t2][4 * j + 1], dkT[t2][4 * j + 2], dkT[t2][4 * j + 3]);
                        *(float4*)(pv + n0) = make_float4(dvT[t2][4 * j], dvT[t2][4 * j + 1], dvT[t2][4 * j + 2], dvT[t2][4 * j + 3]);
                    }
                }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = t2 * 32 + crow(i, hh);
                    if (n < a.d) {
                        const float dk = SCALE_OUT ? dkT[t2][i] * a.scale : dkT[t2][i];
                        stf<T>(dKb + (int64_t)key * a.ldk + n, dk);
                        stf<T>(dVb + (int64_t)key * a.ldv + n, dvT[t2][i]);
                    }
                }
        }
    }
}
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, sizeof(T), true) void flash_dkdv_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[BwdLds<T, DMAX, TR>::DKDV];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    flash_dkdv_body<T, DMAX, NK, TR>(a, smem, bx, by, blockIdx.z);
}

// ---- backward with TWO 32-row tiles per iteration (bf16, hardware transpose reads; option flash_kt = 2) ---------------------
// As flash_fwd2_kernel: half the block barriers, and four independent MFMA chains (scores and dP of two tiles) whose
// element-wise work overlaps.  Same arithmetic per element as the 32-row kernels (no running state here: the statistics
// come from the forward), so dQ / dK / dV differ from theirs only in the order the two tiles' MFMA products are added.
template <int DMAX, int NK, bool WRITE_D = true>
__device__ __forceinline__ void flash_dq2_body(const FlashArgs& a, char* smem, int bx, int by) {
    constexpr bool SCALE_OUT = flash_scale_out(DMAX);
    typedef bf16_t T;
    typedef Geo<T, DMAX> G;
    typedef short8_t F;
    constexpr int ONE = 2 * G::TILE_BYTES;  // [K tile | V tile]
    constexpr int PAIR = 2 * ONE;
    static_assert(2 * PAIR <= 65536, "two double-buffered tile pairs must fit the static LDS limit");
    char* cur = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const T* Gb = (const T*)a.dO + (int64_t)b * a.Nq * a.ldo + h * a.d;
    T* dQb = (T*)a.dQ + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const int q = bx * 128 + wave * 32 + r;
    F qf[NK], gf[NK];
    load_col_frags<T, DMAX, NK>(qf, Qb, a.ldq, q, a.Nq, a.d, hh);
    load_col_frags<T, DMAX, NK>(gf, Gb, a.ldo, q, a.Nq, a.d, hh);
    const float lse_q = (q < a.Nq ? a.lse[(int64_t)by * a.Nq + q] : 0.f) * LOG2E;
    float D_q;
    {  // D[q] = sum_d dO[q, d] O[q, d] (fixed order), stored for the dK/dV pass - as in flash_dq_kernel
        const T* Ob = (const T*)a.O + (int64_t)b * a.Nq * a.ldo + h * a.d;
        F of[NK];
        load_col_frags<T, DMAX, NK>(of, Ob, a.ldo, q, a.Nq, a.d, hh);
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            V16 gv, ov;
            __builtin_memcpy(&gv, &gf[s], 16);
            __builtin_memcpy(&ov, &of[s], 16);
#pragma unroll
            for (int e = 0; e < G::KC; ++e) part += bf16_to_f32(gv.h[e]) * bf16_to_f32(ov.h[e]);
        }
        D_q = half_sum(part);
        if (WRITE_D && q < a.Nq && hh == 0) a.Dbuf[(int64_t)by * a.Nq + q] = D_q;
    }
    const float c2 = a.scale * LOG2E;
    f32x16_t dqT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqT[t][i] = 0.f;
    TileMover<T, DMAX> km0, vm0, km1, vm1;
    const int npairs = (a.Nk + 63) / 64;
    const __amdgpu_buffer_rsrc_t k_slab = slab_rsrc(Kb, a.Nk, a.ldk, a.d), v_slab = slab_rsrc(Vb, a.Nk, a.ldv, a.d);
    km0.prepare(a.ldk, a.d);
    vm0.prepare(a.ldv, a.d);
    km1.prepare(a.ldk, a.d);
    vm1.prepare(a.ldv, a.d);
    auto load_pair = [&](int p) {
        if (COMAT_FLASH_FULL_TILES && (p + 1) * 64 <= a.Nk) {
            const unsigned ko = (unsigned)(p * 64 * a.ldk * (int64_t)sizeof(T)), vo = (unsigned)(p * 64 * a.ldv * (int64_t)sizeof(T));
            km0.load_full(k_slab, ko);
            vm0.load_full(v_slab, vo);
            km1.load_full(k_slab, ko + (unsigned)(32 * a.ldk * (int64_t)sizeof(T)));
            vm1.load_full(v_slab, vo + (unsigned)(32 * a.ldv * (int64_t)sizeof(T)));
            return;
        }
        km0.load(Kb, a.ldk, p * 64, a.Nk, a.d);
        vm0.load(Vb, a.ldv, p * 64, a.Nk, a.d);
        km1.load(Kb, a.ldk, p * 64 + 32, a.Nk, a.d);
        vm1.load(Vb, a.ldv, p * 64 + 32, a.Nk, a.d);
    };
    auto store_pair = [&](char* dst) {
        km0.store(dst);
        vm0.store(dst + G::TILE_BYTES);
        km1.store(dst + ONE);
        vm1.store(dst + ONE + G::TILE_BYTES);
    };
    load_pair(0);
    store_pair(cur);
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int p = 0; p < npairs; ++p) {
        const bool more = p + 1 < npairs;
        if (more) load_pair(p + 1);
        f32x16_t s0, d0, s1, d1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s0[i] = 0.f; d0[i] = 0.f; s1[i] = 0.f; d1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            mma(s0, frag_kc<T, DMAX>(cur, r, s, hh), qf[s]);
            mma(d0, frag_kc<T, DMAX>(cur + G::TILE_BYTES, r, s, hh), gf[s]);
            mma(s1, frag_kc<T, DMAX>(cur + ONE, r, s, hh), qf[s]);
            mma(d1, frag_kc<T, DMAX>(cur + ONE + G::TILE_BYTES, r, s, hh), gf[s]);
        }
        {
            const f32x2_t c2v = splat2(c2), nl = splat2(-lse_q), nD = splat2(-D_q), scv = splat2(a.scale);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2_t p0 = exp2_fast2(__builtin_elementwise_fma(pair_of(s0, i), c2v, nl));
                const f32x2_t p1 = exp2_fast2(__builtin_elementwise_fma(pair_of(s1, i), c2v, nl));
                const f32x2_t e0 = (SCALE_OUT ? p0 : p0 * scv) * (pair_of(d0, i) + nD);
                const f32x2_t e1 = (SCALE_OUT ? p1 : p1 * scv) * (pair_of(d1, i) + nD);
                s0[i] = e0.x; s0[i + 1] = e0.y;
                s1[i] = e1.x; s1[i + 1] = e1.y;
            }
        }
        if ((p + 1) * 64 > a.Nk) {  // keys beyond Nk exist only in the last pair
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (p * 64 + crow(i, hh) >= a.Nk) s0[i] = 0.f;
                if (p * 64 + 32 + crow(i, hh) >= a.Nk) s1[i] = 0.f;
            }
        }
        const unsigned ka0 = lds_addr32(cur) + tr_off, ka1 = lds_addr32(cur + ONE) + tr_off;
        auto step = [&](auto jc, unsigned ka, const f32x16_t& ds) {
            constexpr int j = decltype(jc)::value;
            TrF kfr[G::NT32];
            tr_issue_j<G::RS, j>(ka, kfr, std::make_integer_sequence<int, G::NT32>{});
            const F db = pack_acc<T>(ds, j);
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2) mma(dqT[t2], tr_take(kfr[t2]), db);
        };
        step(std::integral_constant<int, 0>{}, ka0, s0);
        step(std::integral_constant<int, 1>{}, ka0, s0);
        step(std::integral_constant<int, 0>{}, ka1, s1);
        step(std::integral_constant<int, 1>{}, ka1, s1);
        cur = smem + ((p + 1) & 1) * PAIR;
        if (more) store_pair(cur);
        __syncthreads();
    }
    if (q < a.Nq) {
#pragma unroll
        for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = t2 * 32 + crow(i, hh);
                if (n < a.d) stf<T>(dQb + (int64_t)q * a.ldq + n, SCALE_OUT ? dqT[t2][i] * a.scale : dqT[t2][i]);
            }
    }
}
template <int DMAX, int NK>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, 2, true) void flash_dq2_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[BwdLds<bf16_t, DMAX, true>::DQ2];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    flash_dq2_body<DMAX, NK>(a, smem, bx, by);
}

template <int DMAX, int NK>
__device__ __forceinline__ void flash_dkdv2_body(const FlashArgs& a, char* smem, int bx, int by, int bz) {
    constexpr bool SCALE_OUT = flash_scale_out(DMAX);
    typedef bf16_t T;
    typedef Geo<T, DMAX> G;
    typedef short8_t F;
    constexpr int ONE = 2 * G::TILE_BYTES + 256;  // [Q tile | dO tile | lse (32 floats) | D (32 floats)] of one 32-query tile
    constexpr int PAIR = 2 * ONE;
    static_assert(2 * PAIR <= 65536, "two double-buffered tile pairs must fit the static LDS limit");
    char* cur = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = by / a.H, h = by % a.H;
    const T* Qb = (const T*)a.Q + (int64_t)b * a.Nq * a.ldq + h * a.d;
    const T* Kb = (const T*)a.K + (int64_t)b * a.Nk * a.ldk + h * a.d;
    const T* Vb = (const T*)a.V + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const T* Gb = (const T*)a.dO + (int64_t)b * a.Nq * a.ldo + h * a.d;
    T* dKb = (T*)a.dK + (int64_t)b * a.Nk * a.ldk + h * a.d;
    T* dVb = (T*)a.dV + (int64_t)b * a.Nk * a.ldv + h * a.d;
    const int key = bx * 128 + wave * 32 + r;
    F kf[NK], vf[NK];
    load_col_frags<T, DMAX, NK>(kf, Kb, a.ldk, key, a.Nk, a.d, hh);
    load_col_frags<T, DMAX, NK>(vf, Vb, a.ldv, key, a.Nk, a.d, hh);
    f32x16_t dkT[G::NT32], dvT[G::NT32];
#pragma unroll
    for (int t = 0; t < G::NT32; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkT[t][i] = 0.f; dvT[t][i] = 0.f; }
    const float* lse_g = a.lse + (int64_t)by * a.Nq;
    const float* D_g = a.Dbuf + (int64_t)by * a.Nq;
    TileMover<T, DMAX> qm0, gm0, qm1, gm1;
    // query-tile range [tbeg, tend) of this block (gridDim.z ranges, see flash_dkdv_kernel), walked two tiles at a time; a
    // second tile at or beyond `tend` belongs to the next range (or to nobody): its probabilities are forced to zero
    const int ntq = (a.Nq + 31) / 32;
    const int per = (ntq + a.qsplit - 1) / a.qsplit;
    const int tbeg = bz * per;
    const int tend = tbeg + per < ntq ? tbeg + per : ntq;
    const float c2 = a.scale * LOG2E;
    float lse_r = 0.f, D_r = 0.f;  // staged by threads 0..63: queries of the pair (lse -> log2 units at the LDS store)
    const __amdgpu_buffer_rsrc_t q_slab = slab_rsrc(Qb, a.Nq, a.ldq, a.d), g_slab = slab_rsrc(Gb, a.Nq, a.ldo, a.d);
    qm0.prepare(a.ldq, a.d);
    gm0.prepare(a.ldo, a.d);
    qm1.prepare(a.ldq, a.d);
    gm1.prepare(a.ldo, a.d);
    auto load_pair = [&](int t) {
        // lse / D first: the compiler guards their registers with vmcnt waits (the previous pair's loads were consumed inside a
        // divergent region it cannot see through) - behind the tile loads those waits would cover the tiles as well, and the
        // block's barrier would hand wave 0's stall to everybody
        if (threadIdx.x < 64) {
            const int qi = t * 32 + threadIdx.x;
            lse_r = qi < a.Nq ? lse_g[qi] : 0.f;
            D_r = qi < a.Nq ? D_g[qi] : 0.f;
        }
        if (COMAT_FLASH_FULL_TILES && (t + 2) * 32 <= a.Nq) {
            const unsigned qo = (unsigned)(t * 32 * a.ldq * (int64_t)sizeof(T)), go = (unsigned)(t * 32 * a.ldo * (int64_t)sizeof(T));
            qm0.load_full(q_slab, qo);
            gm0.load_full(g_slab, go);
            qm1.load_full(q_slab, qo + (unsigned)(32 * a.ldq * (int64_t)sizeof(T)));
            gm1.load_full(g_slab, go + (unsigned)(32 * a.ldo * (int64_t)sizeof(T)));
        } else {
            qm0.load(Qb, a.ldq, t * 32, a.Nq, a.d);
            gm0.load(Gb, a.ldo, t * 32, a.Nq, a.d);
            qm1.load(Qb, a.ldq, t * 32 + 32, a.Nq, a.d);
            gm1.load(Gb, a.ldo, t * 32 + 32, a.Nq, a.d);
        }
    };
    auto store_pair = [&](char* dst) {
        qm0.store(dst);
        gm0.store(dst + G::TILE_BYTES);
        qm1.store(dst + ONE);
        gm1.store(dst + ONE + G::TILE_BYTES);
        if (threadIdx.x < 64) {
            float* st = (float*)(dst + (threadIdx.x >> 5) * ONE + 2 * G::TILE_BYTES);
            st[threadIdx.x & 31] = lse_r * LOG2E;
            st[32 + (threadIdx.x & 31)] = D_r;
        }
    };
    if (tbeg < tend) {
        load_pair(tbeg);
        store_pair(cur);
    }
    loads_landed();
    __syncthreads();
    const unsigned tr_off = tr_lane_off<G::RS>(lane);
    for (int t = tbeg; t < tend; t += 2) {
        const bool more = t + 2 < tend;
        if (more) load_pair(t + 2);
        const bool has1 = t + 1 < tend;
        f32x16_t c0, d0, c1, d1;  // [query rows x key cols]
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; d0[i] = 0.f; c1[i] = 0.f; d1[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            mma(c0, frag_kc<T, DMAX>(cur, r, s, hh), kf[s]);
            mma(d0, frag_kc<T, DMAX>(cur + G::TILE_BYTES, r, s, hh), vf[s]);
            mma(c1, frag_kc<T, DMAX>(cur + ONE, r, s, hh), kf[s]);
            mma(d1, frag_kc<T, DMAX>(cur + ONE + G::TILE_BYTES, r, s, hh), vf[s]);
        }
        const float* st0 = (const float*)(cur + 2 * G::TILE_BYTES);
        const float* st1 = (const float*)(cur + ONE + 2 * G::TILE_BYTES);
        // rows beyond Nq / a missing second tile: last pair only; keys beyond Nk: last key block only - a BLOCK-uniform
        // condition, so full tiles skip the 3 x 32 compare / select instructions altogether
        const bool edge = (t + 2) * 32 > a.Nq || bx * 128 + 128 > a.Nk || !has1;
        const f32x2_t c2v = splat2(c2), scv = splat2(a.scale);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {  // rows crow(i), crow(i) + 1 sit side by side in the staged lse / D arrays
            const int qr = crow(i, hh);
            const f32x2_t l0 = *(const f32x2_t*)(st0 + qr), l1 = *(const f32x2_t*)(st1 + qr);
            const f32x2_t p0 = exp2_fast2(__builtin_elementwise_fma(pair_of(c0, i), c2v, -l0));
            const f32x2_t p1 = exp2_fast2(__builtin_elementwise_fma(pair_of(c1, i), c2v, -l1));
            c0[i] = p0.x; c0[i + 1] = p0.y;
            c1[i] = p1.x; c1[i + 1] = p1.y;
        }
        if (edge) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qr = crow(i, hh);
                if (!((t * 32 + qr < a.Nq) && (key < a.Nk))) c0[i] = 0.f;
                if (!(has1 && ((t + 1) * 32 + qr < a.Nq) && (key < a.Nk))) c1[i] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const int qr = crow(i, hh);
            const f32x2_t D0 = *(const f32x2_t*)(st0 + 32 + qr), D1 = *(const f32x2_t*)(st1 + 32 + qr);
            const f32x2_t e0 = (SCALE_OUT ? pair_of(c0, i) : pair_of(c0, i) * scv) * (pair_of(d0, i) - D0);
            const f32x2_t e1 = (SCALE_OUT ? pair_of(c1, i) : pair_of(c1, i) * scv) * (pair_of(d1, i) - D1);
            d0[i] = e0.x; d0[i + 1] = e0.y;
            d1[i] = e1.x; d1[i + 1] = e1.y;
        }
        const unsigned qa0 = lds_addr32(cur) + tr_off, ga0 = lds_addr32(cur + G::TILE_BYTES) + tr_off;
        const unsigned qa1 = lds_addr32(cur + ONE) + tr_off, ga1 = lds_addr32(cur + ONE + G::TILE_BYTES) + tr_off;
        auto step = [&](auto jc, unsigned qa, unsigned ga, const f32x16_t& pr, const f32x16_t& ds) {
            constexpr int j = decltype(jc)::value;
            TrF gfr[G::NT32], qfr[G::NT32];
            tr_issue_j<G::RS, j>(ga, gfr, std::make_integer_sequence<int, G::NT32>{});
            tr_issue_j<G::RS, j>(qa, qfr, std::make_integer_sequence<int, G::NT32>{});
            const F pb = pack_acc<T>(pr, j);
            const F db = pack_acc<T>(ds, j);
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2) {
                mma(dvT[t2], tr_take(gfr[t2]), pb);
                mma(dkT[t2], tr_take(qfr[t2]), db);
            }
        };
        step(std::integral_constant<int, 0>{}, qa0, ga0, c0, d0);
        step(std::integral_constant<int, 1>{}, qa0, ga0, c0, d0);
        step(std::integral_constant<int, 0>{}, qa1, ga1, c1, d1);
        step(std::integral_constant<int, 1>{}, qa1, ga1, c1, d1);
        cur = smem + (((t - tbeg) / 2 + 1) & 1) * PAIR;
        if (more) store_pair(cur);
        __syncthreads();
    }
    if (key < a.Nk) {
        const int64_t slab = (int64_t)a.B * a.H * a.Nk * a.d;
        float* pk = a.qsplit > 1
                        ? a.part + ((int64_t)bz * a.B * a.H + by) * a.Nk * a.d + (int64_t)key * a.d
                        : nullptr;
        float* pv = a.qsplit > 1 ? pk + (int64_t)a.qsplit * slab : nullptr;
        if (a.qsplit > 1) {
            // fp32 partials of this query range: accumulator registers 4 j .. 4 j + 3 are 4 consecutive head-dim columns (crow), so a
            // lane writes 16-byte pieces instead of single floats (same values; d is a multiple of 4, the slabs are 16-byte aligned)
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n0 = t2 * 32 + crow(4 * j, hh);
                    if (n0 < a.d) {
                        const float sc = SCALE_OUT ? a.scale : 1.0f;
                        *(float4*)(pk + n0) = SCALE_OUT ? make_float4(dkT[t2][4 * j] * sc, dkT[t2][4 * j + 1] * sc, dkT[t2][4 * j + 2] * sc, dkT[t2][4 * j + 3] * sc)
                                                        : make_float4(dkT[t2][4 * j], dkT[t2][4 * j + 1], dkT[t2][4 * j + 2], dkT[t2][4 * j + 3]);
                        *(float4*)(pv + n0) = make_float4(dvT[t2][4 * j], dvT[t2][4 * j + 1], dvT[t2][4 * j + 2], dvT[t2][4 * j + 3]);
                    }
                }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < G::NT32; ++t2)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int n = t2 * 32 + crow(i, hh);
                    if (n < a.d) {
                        const float dk = SCALE_OUT ? dkT[t2][i] * a.scale : dkT[t2][i];
                        stf<T>(dKb + (int64_t)key * a.ldk + n, dk);
                        stf<T>(dVb + (int64_t)key * a.ldv + n, dvT[t2][i]);
                    }
                }
        }
    }
}
template <int DMAX, int NK>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, 2, true) void flash_dkdv2_kernel(FlashArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[BwdLds<bf16_t, DMAX, true>::DKDV2];
    int bx, by;
    flash_block_xy(a.xcd, bx, by);
    flash_dkdv2_body<DMAX, NK>(a, smem, bx, by, blockIdx.z);
}

// D[q] = sum_d dO[q, d] O[q, d] on its own: the prologue of the dQ bodies, statement for statement (same fragments, same
// order, same bits), for the one-launch backward below whose dK/dV blocks must not wait for its dQ blocks
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS>
__global__ __launch_bounds__(NT) void flash_delta_kernel(FlashArgs a) {
    typedef Geo<T, DMAX> G;
    typedef typename FragOf<T>::type F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.y / a.H, h = blockIdx.y % a.H;
    const T* Gb = (const T*)a.dO + (int64_t)b * a.Nq * a.ldo + h * a.d;
    const T* Ob = (const T*)a.O + (int64_t)b * a.Nq * a.ldo + h * a.d;
    const int q = blockIdx.x * 128 + wave * 32 + r;
    F gf[NK], of[NK];
    load_col_frags<T, DMAX, NK>(gf, Gb, a.ldo, q, a.Nq, a.d, hh);
    load_col_frags<T, DMAX, NK>(of, Ob, a.ldo, q, a.Nq, a.d, hh);
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        V16 gv, ov;
        __builtin_memcpy(&gv, &gf[s], 16);
        __builtin_memcpy(&ov, &of[s], 16);
#pragma unroll
        for (int e = 0; e < G::KC; ++e)
            part += sizeof(T) == 2 ? bf16_to_f32(gv.h[e]) * bf16_to_f32(ov.h[e]) : gv.f[e] * ov.f[e];
    }
    const float D_q = half_sum(part);
    if (q < a.Nq && hh == 0) a.Dbuf[(int64_t)blockIdx.y * a.Nq + q] = D_q;
}

// dQ and dK/dV of one attention in ONE launch (option flash_merge): blocks [0, nqb) of the x dimension run the dQ body,
// the other nkb * qsplit the dK/dV body.  The two passes are independent once D exists (flash_delta_kernel), and at the
// deep levels of the UNet neither fills the chip on its own (16 x 16 latent, 2 x 8 heads: 32 blocks each on 256 CUs; the
// 32 x 32 level: 128 each), so run side by side they take max(dQ, dK/dV) instead of the sum.  Same bodies, same bits as the
// separate kernels.  DQ2: the dQ role walks two key tiles per iteration (flash_dq2_body).
template <typename T, int DMAX, int NK, bool TR, bool DQ2>
__global__ __launch_bounds__(NT) FLASH_OCC(DMAX, sizeof(T), true) void flash_bwd_kernel(FlashArgs a, int nqb, int nkb) {
    typedef BwdLds<T, DMAX, TR> L;
    constexpr int LDQ = DQ2 ? L::DQ2 : L::DQ;
    __shared__ __attribute__((aligned(16))) char smem[LDQ > L::DKDV ? LDQ : L::DKDV];
    const int bx = blockIdx.x;
    if (bx >= nqb) {
        const int k = bx - nqb;
        flash_dkdv_body<T, DMAX, NK, TR>(a, smem, k % nkb, blockIdx.y, k / nkb);
    } else if constexpr (DQ2) {
        flash_dq2_body<DMAX, NK, false>(a, smem, bx, blockIdx.y);
    } else {
        flash_dq_body<T, DMAX, NK, TR, false>(a, smem, bx, blockIdx.y);
    }
}

// dK / dV = sum over the query ranges of the fp32 partials, in range order (bit-reproducible)
template <typename T> __global__ __launch_bounds__(NT) void flash_kv_reduce_kernel(FlashArgs a) {
    const int64_t slab = (int64_t)a.B * a.H * a.Nk * a.d;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= slab) return;
    const int n = (int)(i % a.d);
    const int key = (int)((i / a.d) % a.Nk);
    const int bh = (int)(i / ((int64_t)a.d * a.Nk));
    const int b = bh / a.H, h = bh % a.H;
    float sk = 0.f, sv = 0.f;
    // (unrolled: eight ranges' loads are in flight before the first add - the sums still run in range order, same bits; as a
    // rolled loop every iteration waited out its own memory round trip, up to 64 of them)
#pragma unroll 8
    for (int z = 0; z < a.qsplit; ++z) {
        sk += a.part[(int64_t)z * slab + i];
        sv += a.part[((int64_t)a.qsplit + z) * slab + i];
    }
    stf<T>((T*)a.dK + ((int64_t)b * a.Nk + key) * a.ldk + h * a.d + n, sk);
    stf<T>((T*)a.dV + ((int64_t)b * a.Nk + key) * a.ldv + h * a.d + n, sv);
}

template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
void launch_fwd(const FlashArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && TR && 8 * Geo<T, DMAX>::TILE_BYTES <= 65536) {
        if (comat_option(COMAT_OPT_FLASH_KT) >= 2 && a.Nk > 64) {  // two key tiles per iteration
            hipLaunchKernelGGL((flash_fwd2_kernel<DMAX, NK>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(NT), 0, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((flash_fwd_kernel<T, DMAX, NK, TR>), dim3((a.Nq + 127) / 128, a.B * a.H), dim3(NT), 0, st, a);
}
template <typename T, int DMAX, int NK = Geo<T, DMAX>::NKS, bool TR = false>
void launch_bwd(const FlashArgs& a, hipStream_t st) {
    // option flash_kt: 1 = one 32-row tile per iteration everywhere, 2 = two in the forward, 3 = + dQ, 4 = + dK/dV for head
    // dims <= 64 (two waves per SIMD there; at 80 the one-tile kernel is faster: profiles/r03_n_mb_flash_kt_vgpr.txt),
    // 5 = + dK/dV for every head dim <= 96
    // option flash_merge: dQ and dK/dV in one launch (flash_bwd_kernel, one-tile dK/dV body) - 0 never, 1 when the two
    // grids together hold at most 768 blocks (neither fills 256 CUs x 2 resident blocks alone), 2 always
    const dim3 gq((a.Nq + 127) / 128, a.B * a.H), gk((a.Nk + 127) / 128, a.B * a.H, a.qsplit);
    const int kt = comat_option(COMAT_OPT_FLASH_KT), mg = comat_option(COMAT_OPT_FLASH_MERGE);
    constexpr bool TWO = sizeof(T) == 2 && TR && 8 * Geo<T, DMAX>::TILE_BYTES + 1024 <= 65536;
    const int64_t blocks = ((int64_t)gq.x + (int64_t)gk.x * gk.z) * gq.y;
    const bool merge = (mg == 2 || (mg == 1 && blocks <= 768)) && (int64_t)gq.x + (int64_t)gk.x * gk.z <= 65535;
    const bool dq2 = TWO && kt >= 3 && a.Nk > 64;
    const bool dkdv2 = TWO && !merge && a.Nq > 64 && (kt >= 5 || (kt == 4 && DMAX <= 64));
    if (merge) {
        hipLaunchKernelGGL((flash_delta_kernel<T, DMAX, NK>), gq, dim3(NT), 0, st, a);
        const dim3 g(gq.x + gk.x * gk.z, gq.y);
        if constexpr (TWO) {
            if (dq2) hipLaunchKernelGGL((flash_bwd_kernel<T, DMAX, NK, TR, true>), g, dim3(NT), 0, st, a, (int)gq.x, (int)gk.x);
            else hipLaunchKernelGGL((flash_bwd_kernel<T, DMAX, NK, TR, false>), g, dim3(NT), 0, st, a, (int)gq.x, (int)gk.x);
        } else {
            hipLaunchKernelGGL((flash_bwd_kernel<T, DMAX, NK, TR, false>), g, dim3(NT), 0, st, a, (int)gq.x, (int)gk.x);
        }
    } else {
        if constexpr (TWO) {
            if (dq2) hipLaunchKernelGGL((flash_dq2_kernel<DMAX, NK>), gq, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((flash_dq_kernel<T, DMAX, NK, TR>), gq, dim3(NT), 0, st, a);
            if (dkdv2) hipLaunchKernelGGL((flash_dkdv2_kernel<DMAX, NK>), gk, dim3(NT), 0, st, a);
            else hipLaunchKernelGGL((flash_dkdv_kernel<T, DMAX, NK, TR>), gk, dim3(NT), 0, st, a);
        } else {
            hipLaunchKernelGGL((flash_dq_kernel<T, DMAX, NK, TR>), gq, dim3(NT), 0, st, a);
            hipLaunchKernelGGL((flash_dkdv_kernel<T, DMAX, NK, TR>), gk, dim3(NT), 0, st, a);
        }
    }
    if (a.qsplit > 1) {
        const int64_t slab = (int64_t)a.B * a.H * a.Nk * a.d;
        hipLaunchKernelGGL((flash_kv_reduce_kernel<T>), dim3((unsigned)cdiv64(slab, NT)), dim3(NT), 0, st, a);
    }
}

template <typename T, bool TR> int dispatch_tr(const FlashArgs& a, bool bwd, bool trim, hipStream_t st) {
    constexpr int KC2 = 2 * (16 / (int)sizeof(T));
    if (trim && a.d > 32 && a.d <= 48) {
        if (bwd) launch_bwd<T, 64, 48 / KC2, TR>(a, st);
        else launch_fwd<T, 64, 48 / KC2, TR>(a, st);
        return 0;
    }
    if (trim && a.d > 64 && a.d <= 80) {
        if (bwd) launch_bwd<T, 96, 80 / KC2, TR>(a, st);
        else launch_fwd<T, 96, 80 / KC2, TR>(a, st);
        return 0;
    }
#define FA_CASE(D)                                             \
    if (a.d <= D) {                                            \
        if (bwd) launch_bwd<T, D, Geo<T, D>::NKS, TR>(a, st);  \
        else launch_fwd<T, D, Geo<T, D>::NKS, TR>(a, st);      \
        return 0;                                              \
    }
    FA_CASE(32)
    FA_CASE(64)
    FA_CASE(96)
    FA_CASE(160)
#undef FA_CASE
    return -1;
}

template <typename T> int dispatch(const FlashArgs& a, bool bwd, hipStream_t st) {
    // Two refinements, both bit-identical to the plain kernels by construction and validated so on MI355X (round 2,
    // profiles/r02_a_*): TRIM skips the MFMA k-steps whose head-dim chunk is pure zero padding (head dim 40 in a 64-wide
    // tile: 3 of 4 steps; 80 in 96: 5 of 6) - the skipped products are exactly zero; TR stages the k-major operand tiles
    // (V; K in dQ; Q and dO in dK/dV) as transposed LDS images, so their MFMA fragments are two 8-byte reads instead of
    // eight 2-byte reads.  Options flash_trim / flash_tr = 0 (COMAT_FLASH_TRIM / COMAT_FLASH_TR) select the plain kernels.
    const int trim_v = comat_option(COMAT_OPT_FLASH_TRIM), tr_v = comat_option(COMAT_OPT_FLASH_TR);
    // flash_tr: 0 never, 2 always, 1 (default) by head dim: building the transposed image costs 8 scalar LDS writes per
    // 16-byte chunk, which pays off for d <= 64 only (profiles/r02_h_mb_flash.txt: d = 80 backward 133 vs 144 us,
    // d = 160 81 vs 92 us without it; d = 40 441 vs 474 us with it)
    // (round 3) bf16: TR = hardware transpose reads straight from the plain tiles (no transposed image to build), which pays
    // for every head dim; the head-dim rule above still governs the software image of the fp32 parity mode
    if (tr_v == 2 || (tr_v == 1 && (a.d <= 64 || sizeof(T) == 2))) return dispatch_tr<T, true>(a, bwd, trim_v == 1, st);
    return dispatch_tr<T, false>(a, bwd, trim_v == 1, st);
}

int check_args(const char* what, const void* Q, const void* K, const void* V, int B, int H, int Nq, int Nk, int d,
               int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int dtype) {
    COMAT_REQUIRE(Q && K && V, "%s: null pointer", what);
    COMAT_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && d > 0 && (int64_t)B * H <= 65535, "%s: bad shape", what);
    COMAT_REQUIRE(dtype_ok(dtype), "%s: bad dtype", what);
    const int kc = dtype == COMAT_BF16 ? 8 : 4;
    COMAT_REQUIRE(d <= 160 && d % kc == 0, "%s: head dim %d unsupported (<=160, multiple of %d)", what, d, kc);
    COMAT_REQUIRE(ldq % kc == 0 && ldk % kc == 0 && ldv % kc == 0 && ldo % kc == 0, "%s: leading dims must be 16-byte multiples", what);
    COMAT_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V) & 15) == 0, "%s: operands must be 16-byte aligned", what);
    // one (batch, head) slab is addressed through a buffer descriptor with 32-bit offsets (TileMover::load_full)
    const int64_t esz = dtype == COMAT_BF16 ? 2 : 4, lim = (int64_t)1 << 31;
    COMAT_REQUIRE(Nq * ldq * esz < lim && Nk * ldk * esz < lim && Nk * ldv * esz < lim && Nq * ldo * esz < lim,
                  "%s: a batch entry of an operand must stay below 2 GiB", what);
    return 0;
}

}  // namespace

static int flash_fwd_impl(const char* what, unsigned char* q8, const float* q_scale, uint32_t* q_amax, int64_t ldq8, const void* Q, const void* K, const void* V, void* O, float* lse, int32_t B,
                                    int32_t H, int32_t Nq, int32_t Nk, int32_t d, int64_t ldq, int64_t ldk,
                                    int64_t ldv, int64_t ldo, float scale, int32_t dtype, void* stream) {
    if (int rc = check_args(what, Q, K, V, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, dtype)) return rc;
    COMAT_REQUIRE(O && lse, "%s: null output", what);
    FlashArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.Out = O; a.lse = lse;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
    a.xcd = comat_option(COMAT_OPT_FLASH_XCD);
    a.q8 = q8; a.q_scale = q_scale; a.q_amax = (unsigned*)q_amax; a.ldq8 = ldq8;
    const int rc = dtype == COMAT_BF16 ? dispatch<bf16_t>(a, false, (hipStream_t)stream)
                                       : dispatch<float>(a, false, (hipStream_t)stream);
    COMAT_REQUIRE(rc == 0, "%s: unsupported head dim", what);
    return comat_check_launch(what);
}

extern "C" int comat_flash_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int32_t B, int32_t H, int32_t Nq,
                                    int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int32_t dtype,
                                    void* stream) {
    return flash_fwd_impl("comat_flash_attn_fwd", nullptr, nullptr, nullptr, 0, Q, K, V, O, lse, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale,
                          dtype, stream);
}

extern "C" int comat_flash_attn_fwd_q(const void* Q, const void* K, const void* V, void* O, float* lse, int32_t B, int32_t H, int32_t Nq,
                                      int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale,
                                      int32_t dtype, void* q8, int64_t ldq8, const float* q_scale, uint32_t* q_amax, void* stream) {
    COMAT_REQUIRE(q8 && q_scale && q_amax && ldq8 >= (int64_t)H * d && ldq8 % 4 == 0 && (((uintptr_t)q8) & 3) == 0 && d % 4 == 0,
                  "comat_flash_attn_fwd_q: q8, q_scale, q_amax, 4-byte aligned rows (ldq8 %% 4 == 0, d %% 4 == 0)");
    return flash_fwd_impl("comat_flash_attn_fwd_q", (unsigned char*)q8, q_scale, q_amax, ldq8, Q, K, V, O, lse, B, H, Nq, Nk, d, ldq, ldk, ldv,
                          ldo, scale, dtype, stream);
}

extern "C" int comat_flash_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                    const float* lse, float* Dbuf, void* dQ, void* dK, void* dV, int32_t B, int32_t H,
                                    int32_t Nq, int32_t Nk, int32_t d, int64_t ldq, int64_t ldk, int64_t ldv,
                                    int64_t ldo, float scale, int32_t dtype, float* ws, int64_t ws_bytes,
                                    void* stream) {
    if (int rc = check_args("comat_flash_attn_bwd", Q, K, V, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, dtype)) return rc;
    COMAT_REQUIRE(O && dO && lse && Dbuf && dQ && dK && dV, "comat_flash_attn_bwd: null pointer");
    COMAT_REQUIRE((((uintptr_t)dO | (uintptr_t)O) & 15) == 0, "comat_flash_attn_bwd: O and dO must be 16-byte aligned");
    FlashArgs a = {};
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.lse = (float*)lse; a.Dbuf = Dbuf;
    a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale;
    a.xcd = comat_option(COMAT_OPT_FLASH_XCD);
    // few key blocks (cross-attention): cut the query loop into ranges so that the grid fills the chip
    a.qsplit = 1;
    a.part = ws;
    const int64_t base_blocks = (int64_t)((Nk + 127) / 128) * B * H, ntq = (Nq + 31) / 32;
    if (ws && (((uintptr_t)ws) & 15) == 0 && d % 4 == 0 && base_blocks < 256 && ntq >= 8) {  // (16-byte partial stores)
        int64_t qs = cdiv64(comat_option(COMAT_OPT_FLASH_QS) > 0 ? comat_option(COMAT_OPT_FLASH_QS) : 512, base_blocks);
        if (qs > ntq / 4) qs = ntq / 4;
        const int64_t cap = ws_bytes / (2 * (int64_t)B * H * Nk * d * 4);
        if (qs > cap) qs = cap;
        if (qs > 64) qs = 64;
        if (qs >= 2) a.qsplit = (int)qs;
    }
    const int rc = dtype == COMAT_BF16 ? dispatch<bf16_t>(a, true, (hipStream_t)stream)
                                       : dispatch<float>(a, true, (hipStream_t)stream);
    COMAT_REQUIRE(rc == 0, "comat_flash_attn_bwd: unsupported head dim");
    return comat_check_launch("comat_flash_attn_bwd");
}
