// eltwise.hip — HBM-bound elementwise / data-movement kernels of the CoMat step (grid-stride, runtime dtype).
#include "common.h"

namespace {

constexpr int NT = 256;
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < (n); i += (int64_t)gridDim.x * NT)

__global__ __launch_bounds__(NT) void unary_kernel(int op, const void* __restrict__ x, void* __restrict__ y, int64_t n,
                                                   float p0, float p1, int xdt, int ydt) {
    GRID_STRIDE(i, n) {
        float v = ld_dt(x, i, xdt);
        if (op == COMAT_UN_SILU) v = silu_f(v);
        else if (op == COMAT_UN_GELU) v = gelu_f(v);
        else if (op == COMAT_UN_AFFINE) v = p0 * v + p1;
        st_dt(y, i, v, ydt);
    }
}

__global__ __launch_bounds__(NT) void unary_bwd_kernel(int op, const void* __restrict__ dy, const void* __restrict__ x,
                                                       void* __restrict__ dx, int64_t n, int dt) {
    GRID_STRIDE(i, n) {
        const float v = ld_dt(x, i, dt);
        float g = ld_dt(dy, i, dt);
        if (op == COMAT_UN_SILU) g *= silu_grad_f(v);
        else if (op == COMAT_UN_GELU) g *= gelu_grad_f(v);
        st_dt(dx, i, g, dt);
    }
}

__global__ __launch_bounds__(NT) void axpby_kernel(float a, const void* __restrict__ x, float b,
                                                   const void* __restrict__ y, void* __restrict__ out, int64_t n,
                                                   int xdt, int ydt, int odt) {
    GRID_STRIDE(i, n) {
        float v = a * ld_dt(x, i, xdt);
        if (y) v += b * ld_dt(y, i, ydt);
        st_dt(out, i, v, odt);
    }
}

__global__ __launch_bounds__(NT) void geglu_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t M,
                                                       int D, int dt) {
    const int64_t n = M * D;
    GRID_STRIDE(i, n) {
        const int64_t m = i / D;
        const int d = (int)(i - m * D);
        const float a = ld_dt(x, m * 2 * D + d, dt), g = ld_dt(x, m * 2 * D + D + d, dt);
        st_dt(y, i, a * gelu_f(g), dt);
    }
}

__global__ __launch_bounds__(NT) void geglu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                       void* __restrict__ dx, int64_t M, int D, int dt) {
    const int64_t n = M * D;
    GRID_STRIDE(i, n) {
        const int64_t m = i / D;
        const int d = (int)(i - m * D);
        const float a = ld_dt(x, m * 2 * D + d, dt), g = ld_dt(x, m * 2 * D + D + d, dt);
        const float go = ld_dt(dy, i, dt);
        st_dt(dx, m * 2 * D + d, go * gelu_f(g), dt);
        st_dt(dx, m * 2 * D + D + d, go * a * gelu_grad_f(g), dt);
    }
}

// GEGLU on the INTERLEAVED layout of the fused projection (comat_gemm, epi2): every 32 columns of x [M, 2 D] hold 16 value
// channels followed by their 16 gate channels, so that one lane of the GEMM epilogue owns both halves of 8 channels.
// Work item = 8 channels of one row: 16-byte accesses (bf16), 2 x 16 (fp32).
template <typename T> struct V8;
template <> struct V8<bf16_t> {
    uint4 u;
    __device__ __forceinline__ void load(const bf16_t* p) { u = *(const uint4*)p; }
    __device__ __forceinline__ void store(bf16_t* p) const { *(uint4*)p = u; }
    __device__ __forceinline__ float get(int e) const { return bf16_to_f32(((const bf16_t*)&u)[e]); }
    __device__ __forceinline__ void set(int e, float v) { ((bf16_t*)&u)[e] = f32_to_bf16(v); }
};
template <> struct V8<float> {
    float v[8];  // (one array: indexing past a float4 member into its neighbour was undefined behaviour - ADVICE r4)
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __device__ __forceinline__ float get(int e) const { return v[e]; }
    __device__ __forceinline__ void set(int e, float x) { v[e] = x; }
};

template <typename T>
__global__ __launch_bounds__(NT) void geglu_il_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t M, int D) {
    const int D8 = D / 8;
    const int64_t n = M * D8;
    GRID_STRIDE(i, n) {
        const int64_t m = i / D8;
        const int c8 = (int)(i - m * D8);           // channels 8 c8 .. + 8
        const int col = (c8 >> 1) * 32 + (c8 & 1) * 8;  // their value columns; the gate columns are 16 further
        V8<T> a, g, o;
        a.load(x + m * 2 * D + col);
        g.load(x + m * 2 * D + col + 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) o.set(e, a.get(e) * gelu_f(g.get(e)));
        o.store(y + m * D + c8 * 8);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void geglu_il_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                                          int64_t M, int D) {
    const int D8 = D / 8;
    const int64_t n = M * D8;
    GRID_STRIDE(i, n) {
        const int64_t m = i / D8;
        const int c8 = (int)(i - m * D8);
        const int col = (c8 >> 1) * 32 + (c8 & 1) * 8;
        V8<T> a, g, go, da, dg;
        a.load(x + m * 2 * D + col);
        g.load(x + m * 2 * D + col + 16);
        go.load(dy + m * D + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            da.set(e, go.get(e) * gelu_f(g.get(e)));
            dg.set(e, go.get(e) * a.get(e) * gelu_grad_f(g.get(e)));
        }
        da.store(dx + m * 2 * D + col);
        dg.store(dx + m * 2 * D + col + 16);
    }
}

__global__ __launch_bounds__(NT) void copy2d_kernel(const void* __restrict__ src, int64_t lds_, void* __restrict__ dst,
                                                    int64_t ldd, int64_t rows, int64_t cols, int sdt, int ddt) {
    const int64_t n = rows * cols;
    GRID_STRIDE(i, n) {
        const int64_t r = i / cols, c = i - r * cols;
        st_dt(dst, r * ldd + c, ld_dt(src, r * lds_ + c, sdt), ddt);
    }
}

// two strided 2-D copies in ONE launch, 16 bytes per thread and step: the channel concat of a skip connection (out[:, :Ca] = a,
// out[:, Ca:] = b) and its backward split - twice per up-path ResnetBlock and direction (336 two-copy pairs per C2 step)
struct Copy2dPair {
    const char* src[2];
    char* dst[2];
    int64_t lds[2], ldd[2];  // bytes between rows
    int vpr[2];              // 16-byte vectors per row
};
__global__ __launch_bounds__(NT) void copy2d_pair_kernel(Copy2dPair p, int64_t rows) {
    const int vt = p.vpr[0] + p.vpr[1];
    const int64_t n = rows * vt;
    GRID_STRIDE(i, n) {
        const int64_t r = i / vt;
        int v = (int)(i - r * vt);
        const int w = v >= p.vpr[0] ? 1 : 0;
        v -= w * p.vpr[0];
        *(uint4*)(p.dst[w] + r * p.ldd[w] + (int64_t)v * 16) = *(const uint4*)(p.src[w] + r * p.lds[w] + (int64_t)v * 16);
    }
}

__global__ __launch_bounds__(NT) void add_rowvec_kernel(const void* __restrict__ x, const void* __restrict__ v,
                                                        void* __restrict__ out, int64_t rows, int64_t cols, int dt) {
    const int64_t n = rows * cols;
    GRID_STRIDE(i, n) { st_dt(out, i, ld_dt(x, i, dt) + ld_dt(v, i % cols, dt), dt); }
}

__global__ __launch_bounds__(NT) void sumpool2x2_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int H,
                                                        int W, int C, int dt) {
    const int64_t n = (int64_t)B * H * W * C;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int64_t b = t / H;
        const int64_t W2 = 2 * (int64_t)W;
        const int64_t base = ((b * 2 * H + 2 * h) * W2 + 2 * w) * C + c;
        const float v = ld_dt(x, base, dt) + ld_dt(x, base + C, dt) + ld_dt(x, base + W2 * C, dt) +
                        ld_dt(x, base + W2 * C + C, dt);
        st_dt(y, i, v, dt);
    }
}

__global__ __launch_bounds__(NT) void permute_kernel(const void* __restrict__ x, void* __restrict__ y, int B, int C,
                                                     int H, int W, int to_nhwc, int xdt, int ydt) {
    const int64_t n = (int64_t)B * C * H * W;
    GRID_STRIDE(i, n) {  // i indexes the OUTPUT
        if (to_nhwc) {
            const int c = (int)(i % C);
            int64_t t = i / C;
            const int w = (int)(t % W);
            t /= W;
            const int h = (int)(t % H);
            const int64_t b = t / H;
            st_dt(y, i, ld_dt(x, ((b * C + c) * H + h) * W + w, xdt), ydt);
        } else {
            const int w = (int)(i % W);
            int64_t t = i / W;
            const int h = (int)(t % H);
            t /= H;
            const int c = (int)(t % C);
            const int64_t b = t / C;
            st_dt(y, i, ld_dt(x, ((b * H + h) * W + w) * C + c, xdt), ydt);
        }
    }
}

__global__ __launch_bounds__(NT) void cfg_ddpm_fwd_kernel(const float* __restrict__ x, const void* __restrict__ eps2,
                                                          const float* __restrict__ z, float* __restrict__ xp,
                                                          int64_t n, float s, float cx, float ce, float sigma,
                                                          int edt) {
    GRID_STRIDE(i, n) {
        const float eu = ld_dt(eps2, i, edt), ec = ld_dt(eps2, n + i, edt);
        const float eps = eu + s * (ec - eu);
        float v = cx * x[i] + ce * eps;
        if (z) v += sigma * z[i];
        xp[i] = v;
    }
}

__global__ __launch_bounds__(NT) void cfg_ddpm_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx,
                                                          void* __restrict__ deps2, int64_t n, float s, float cx,
                                                          float ce, int edt) {
    GRID_STRIDE(i, n) {
        const float gi = g[i];
        if (dx) dx[i] = cx * gi;
        st_dt(deps2, i, ce * (1.0f - s) * gi, edt);
        st_dt(deps2, n + i, ce * s * gi, edt);
    }
}

// Batched transpose + cast of many small matrices in one launch: block b handles the 32x32 tile described by
// tiles[b] = (src_off, dst_off, rows, cols, r0, c0): dst[dst_off + c*rows + r] = src[src_off + r*cols + c].
__global__ __launch_bounds__(NT) void transpose_tiles_kernel(const float* __restrict__ src, void* __restrict__ dst,
                                                             const int64_t* __restrict__ tiles, int out_dt) {
    __shared__ float tile[32][33];
    const int64_t* d = tiles + (int64_t)blockIdx.x * 6;
    const int64_t so = d[0], dof = d[1], rows = d[2], cols = d[3], r0 = d[4], c0 = d[5];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 32; j += NT / 32) {
        const int64_t r = r0 + ty + j, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + j][tx] = src[so + r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += NT / 32) {
        const int64_t c = c0 + ty + j, r = r0 + tx;
        if (r < rows && c < cols) st_dt(dst, dof + c * rows + r, tile[tx][ty + j], out_dt);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int comat_unary(int32_t op, const void* x, void* y, int64_t n, float p0, float p1, int32_t x_dtype,
                           int32_t y_dtype, void* stream) {
    COMAT_REQUIRE(x && y && n > 0, "comat_unary: bad args");
    COMAT_REQUIRE(op >= 0 && op <= 3, "comat_unary: bad op %d", op);
    COMAT_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype), "comat_unary: bad dtype");
    hipLaunchKernelGGL(unary_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, ST, op, x, y, n, p0, p1, x_dtype, y_dtype);
    return comat_check_launch("comat_unary");
}

extern "C" int comat_unary_bwd(int32_t op, const void* dy, const void* x, void* dx, int64_t n, int32_t dtype,
                               void* stream) {
    COMAT_REQUIRE(dy && x && dx && n > 0, "comat_unary_bwd: bad args");
    COMAT_REQUIRE(op == COMAT_UN_SILU || op == COMAT_UN_GELU, "comat_unary_bwd: bad op %d", op);
    COMAT_REQUIRE(dtype_ok(dtype), "comat_unary_bwd: bad dtype");
    hipLaunchKernelGGL(unary_bwd_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, ST, op, dy, x, dx, n, dtype);
    return comat_check_launch("comat_unary_bwd");
}

extern "C" int comat_axpby(float a, const void* x, float b, const void* y, void* out, int64_t n, int32_t x_dtype,
                           int32_t y_dtype, int32_t out_dtype, void* stream) {
    COMAT_REQUIRE(x && out && n > 0, "comat_axpby: bad args");
    COMAT_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype) && (!y || dtype_ok(y_dtype)), "comat_axpby: bad dtype");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, ST, a, x, b, y, out, n, x_dtype, y_dtype,
                       out_dtype);
    return comat_check_launch("comat_axpby");
}

extern "C" int comat_geglu_fwd(const void* x, void* y, int64_t M, int32_t D, int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && y && M > 0 && D > 0 && dtype_ok(dtype), "comat_geglu_fwd: bad args");
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_1d(M * D, NT)), dim3(NT), 0, ST, x, y, M, D, dtype);
    return comat_check_launch("comat_geglu_fwd");
}

extern "C" int comat_geglu_il_fwd(const void* x, void* y, int64_t M, int32_t D, int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && y && M > 0 && D > 0 && D % 16 == 0 && dtype_ok(dtype), "comat_geglu_il_fwd: bad args (D must be a multiple of 16)");
    COMAT_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "comat_geglu_il_fwd: operands must be 16-byte aligned");
    const int g = grid_1d(M * (D / 8), NT, 1 << 20);
    if (dtype == COMAT_BF16) hipLaunchKernelGGL(geglu_il_fwd_kernel<bf16_t>, dim3(g), dim3(NT), 0, ST, (const bf16_t*)x, (bf16_t*)y, M, D);
    else hipLaunchKernelGGL(geglu_il_fwd_kernel<float>, dim3(g), dim3(NT), 0, ST, (const float*)x, (float*)y, M, D);
    return comat_check_launch("comat_geglu_il_fwd");
}

extern "C" int comat_geglu_il_bwd(const void* dy, const void* x, void* dx, int64_t M, int32_t D, int32_t dtype, void* stream) {
    COMAT_REQUIRE(dy && x && dx && M > 0 && D > 0 && D % 16 == 0 && dtype_ok(dtype), "comat_geglu_il_bwd: bad args (D must be a multiple of 16)");
    COMAT_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0, "comat_geglu_il_bwd: operands must be 16-byte aligned");
    const int g = grid_1d(M * (D / 8), NT, 1 << 20);
    if (dtype == COMAT_BF16)
        hipLaunchKernelGGL(geglu_il_bwd_kernel<bf16_t>, dim3(g), dim3(NT), 0, ST, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, M, D);
    else hipLaunchKernelGGL(geglu_il_bwd_kernel<float>, dim3(g), dim3(NT), 0, ST, (const float*)dy, (const float*)x, (float*)dx, M, D);
    return comat_check_launch("comat_geglu_il_bwd");
}

extern "C" int comat_geglu_bwd(const void* dy, const void* x, void* dx, int64_t M, int32_t D, int32_t dtype,
                               void* stream) {
    COMAT_REQUIRE(dy && x && dx && M > 0 && D > 0 && dtype_ok(dtype), "comat_geglu_bwd: bad args");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_1d(M * D, NT)), dim3(NT), 0, ST, dy, x, dx, M, D, dtype);
    return comat_check_launch("comat_geglu_bwd");
}

extern "C" int comat_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                            int32_t src_dtype, int32_t dst_dtype, void* stream) {
    COMAT_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "comat_copy2d: bad args");
    COMAT_REQUIRE(dtype_ok(src_dtype) && dtype_ok(dst_dtype), "comat_copy2d: bad dtype");
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_1d(rows * cols, NT)), dim3(NT), 0, ST, src, ld_src, dst, ld_dst, rows,
                       cols, src_dtype, dst_dtype);
    return comat_check_launch("comat_copy2d");
}

extern "C" int comat_copy2d_pair(const void* src0, int64_t ld_src0, void* dst0, int64_t ld_dst0, int64_t cols0, const void* src1,
                                 int64_t ld_src1, void* dst1, int64_t ld_dst1, int64_t cols1, int64_t rows, int32_t dtype,
                                 void* stream) {
    COMAT_REQUIRE(src0 && dst0 && src1 && dst1 && rows > 0 && cols0 > 0 && cols1 > 0, "comat_copy2d_pair: bad args");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_copy2d_pair: bad dtype");
    const int eb = dtype == COMAT_F32 ? 4 : 2, epv = 16 / eb;
    COMAT_REQUIRE(ld_src0 >= cols0 && ld_dst0 >= cols0 && ld_src1 >= cols1 && ld_dst1 >= cols1, "comat_copy2d_pair: leading dimension too small");
    COMAT_REQUIRE(cols0 % epv == 0 && cols1 % epv == 0 && ld_src0 % epv == 0 && ld_dst0 % epv == 0 && ld_src1 % epv == 0 &&
                      ld_dst1 % epv == 0 && (((uintptr_t)src0 | (uintptr_t)dst0 | (uintptr_t)src1 | (uintptr_t)dst1) & 15) == 0 &&
                      (cols0 + cols1) / epv < (1 << 30),
                  "comat_copy2d_pair: 16-byte rows only (columns and leading dimensions in whole 16-byte vectors, aligned pointers)");
    Copy2dPair p;
    p.src[0] = (const char*)src0; p.src[1] = (const char*)src1; p.dst[0] = (char*)dst0; p.dst[1] = (char*)dst1;
    p.lds[0] = ld_src0 * eb; p.lds[1] = ld_src1 * eb; p.ldd[0] = ld_dst0 * eb; p.ldd[1] = ld_dst1 * eb;
    p.vpr[0] = (int)(cols0 / epv); p.vpr[1] = (int)(cols1 / epv);
    hipLaunchKernelGGL(copy2d_pair_kernel, dim3(grid_1d(rows * (p.vpr[0] + p.vpr[1]), NT, 1 << 16)), dim3(NT), 0, ST, p, rows);
    return comat_check_launch("comat_copy2d_pair");
}

extern "C" int comat_add_rowvec(const void* x, const void* v, void* out, int64_t rows, int64_t cols, int32_t dtype,
                                void* stream) {
    COMAT_REQUIRE(x && v && out && rows > 0 && cols > 0 && dtype_ok(dtype), "comat_add_rowvec: bad args");
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_1d(rows * cols, NT)), dim3(NT), 0, ST, x, v, out, rows, cols,
                       dtype);
    return comat_check_launch("comat_add_rowvec");
}

extern "C" int comat_sumpool2x2(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                void* stream) {
    COMAT_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && dtype_ok(dtype), "comat_sumpool2x2: bad args");
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(grid_1d((int64_t)B * H * W * C, NT)), dim3(NT), 0, ST, x, y, B, H, W, C,
                       dtype);
    return comat_check_launch("comat_sumpool2x2");
}

extern "C" int comat_permute_nchw_nhwc(const void* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W,
                                       int32_t to_nhwc, int32_t x_dtype, int32_t y_dtype, void* stream) {
    COMAT_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "comat_permute_nchw_nhwc: bad args");
    COMAT_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype), "comat_permute_nchw_nhwc: bad dtype");
    hipLaunchKernelGGL(permute_kernel, dim3(grid_1d((int64_t)B * C * H * W, NT)), dim3(NT), 0, ST, x, y, B, C, H, W,
                       to_nhwc, x_dtype, y_dtype);
    return comat_check_launch("comat_permute_nchw_nhwc");
}

extern "C" int comat_cfg_ddpm_fwd(const float* x, const void* eps2, const float* z, float* x_prev, int64_t n, float s,
                                  float cx, float ce, float sigma, int32_t eps_dtype, void* stream) {
    COMAT_REQUIRE(x && eps2 && x_prev && n > 0 && dtype_ok(eps_dtype), "comat_cfg_ddpm_fwd: bad args");
    hipLaunchKernelGGL(cfg_ddpm_fwd_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, ST, x, eps2, z, x_prev, n, s, cx, ce,
                       sigma, eps_dtype);
    return comat_check_launch("comat_cfg_ddpm_fwd");
}

extern "C" int comat_cfg_ddpm_bwd(const float* g, float* dx, void* deps2, int64_t n, float s, float cx, float ce,
                                  int32_t eps_dtype, void* stream) {
    COMAT_REQUIRE(g && deps2 && n > 0 && dtype_ok(eps_dtype), "comat_cfg_ddpm_bwd: bad args");
    hipLaunchKernelGGL(cfg_ddpm_bwd_kernel, dim3(grid_1d(n, NT)), dim3(NT), 0, ST, g, dx, deps2, n, s, cx, ce,
                       eps_dtype);
    return comat_check_launch("comat_cfg_ddpm_bwd");
}

extern "C" int comat_transpose_cast_tiles(const float* src, void* dst, const int64_t* tiles, int64_t n_tiles,
                                          int32_t out_dtype, void* stream) {
    COMAT_REQUIRE(src && dst && tiles && n_tiles > 0 && n_tiles < (1ll << 31) && dtype_ok(out_dtype),
                  "comat_transpose_cast_tiles: bad args");
    hipLaunchKernelGGL(transpose_tiles_kernel, dim3((unsigned)n_tiles), dim3(NT), 0, ST, src, dst, tiles, out_dtype);
    return comat_check_launch("comat_transpose_cast_tiles");
}
