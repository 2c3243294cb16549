// image.hip — image-space data movement between the VAE decoder and the BLIP captioner, plus embedding lookup.
#include "common.h"

namespace {

constexpr int NT = 256;
#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < (n); i += (int64_t)gridDim.x * NT)

// Separable sparse resampling (crop + antialiased bicubic + per-channel affine in one pass).  With transposed tap
// tables the same kernel is the adjoint operator (gradient w.r.t. the source image).
__global__ __launch_bounds__(NT) void resample2d_kernel(const void* __restrict__ in, void* __restrict__ out, int B,
                                                        int Hin, int Win, int Hout, int Wout, int C,
                                                        const int32_t* __restrict__ ystart,
                                                        const float* __restrict__ ywt,
                                                        const int32_t* __restrict__ xstart,
                                                        const float* __restrict__ xwt, int KT,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int idt, int odt) {
    const int64_t n = (int64_t)B * Hout * Wout * C;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int ox = (int)(t % Wout);
        t /= Wout;
        const int oy = (int)(t % Hout);
        const int64_t b = t / Hout;
        const int ys = ystart[oy], xs = xstart[ox];
        float acc = 0.f;
        for (int ty = 0; ty < KT; ++ty) {
            const float wy = ywt[oy * KT + ty];
            const int sy = ys + ty;
            if (wy == 0.f || sy < 0 || sy >= Hin) continue;
            float racc = 0.f;
            for (int tx = 0; tx < KT; ++tx) {
                const float wx = xwt[ox * KT + tx];
                const int sx = xs + tx;
                if (wx == 0.f || sx < 0 || sx >= Win) continue;
                racc += wx * ld_dt(in, ((b * Hin + sy) * Win + sx) * C + c, idt);
            }
            acc += wy * racc;
        }
        const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
        st_dt(out, i, sc * acc + sh, odt);
    }
}

// img [B,H,W,C] <-> patches [B*(H/P)*(W/P), P*P*C] with inner order (ky, kx, c)
__global__ __launch_bounds__(NT) void patchify_kernel(const void* __restrict__ src, void* __restrict__ dst, int B, int H,
                                                      int W, int C, int P, int inverse, int dt) {
    const int64_t n = (int64_t)B * H * W * C;
    const int nW = W / P, nH = H / P;
    GRID_STRIDE(i, n) {  // i indexes the patch matrix
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int kx = (int)(t % P);
        t /= P;
        const int ky = (int)(t % P);
        t /= P;
        const int pw = (int)(t % nW);
        t /= nW;
        const int ph = (int)(t % nH);
        const int64_t b = t / nH;
        const int64_t j = ((b * H + ph * P + ky) * W + pw * P + kx) * C + c;
        if (!inverse) st_dt(dst, i, ld_dt(src, j, dt), dt);
        else st_dt(dst, j, ld_dt(src, i, dt), dt);
    }
}

__global__ __launch_bounds__(NT) void embedding_kernel(const int64_t* __restrict__ ids, const void* __restrict__ table,
                                                       void* __restrict__ out, int64_t n, int dim, int64_t vocab,
                                                       int dt) {
    const int64_t total = n * dim;
    GRID_STRIDE(i, total) {
        const int64_t r = i / dim;
        const int d = (int)(i - r * dim);
        int64_t id = ids[r];
        if (id < 0) id = 0;
        if (id >= vocab) id = vocab - 1;
        st_dt(out, i, ld_dt(table, id * dim + d, dt), dt);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int comat_resample2d(const void* in, void* out, int32_t B, int32_t Hin, int32_t Win, int32_t Hout,
                                int32_t Wout, int32_t C, const int32_t* ystart, const float* ywt, const int32_t* xstart,
                                const float* xwt, int32_t KT, const float* scale, const float* shift, int32_t in_dtype,
                                int32_t out_dtype, void* stream) {
    COMAT_REQUIRE(in && out && ystart && ywt && xstart && xwt, "comat_resample2d: null pointer");
    COMAT_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && KT > 0, "comat_resample2d: bad shape");
    COMAT_REQUIRE(dtype_ok(in_dtype) && dtype_ok(out_dtype), "comat_resample2d: bad dtype");
    hipLaunchKernelGGL(resample2d_kernel, dim3(grid_1d((int64_t)B * Hout * Wout * C, NT)), dim3(NT), 0, ST, in, out, B,
                       Hin, Win, Hout, Wout, C, ystart, ywt, xstart, xwt, KT, scale, shift, in_dtype, out_dtype);
    return comat_check_launch("comat_resample2d");
}

extern "C" int comat_patchify(const void* img, void* patches, int32_t B, int32_t H, int32_t W, int32_t C, int32_t P,
                              int32_t inverse, int32_t dtype, void* stream) {
    COMAT_REQUIRE(img && patches, "comat_patchify: null pointer");
    COMAT_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "comat_patchify: bad shape");
    COMAT_REQUIRE(dtype_ok(dtype), "comat_patchify: bad dtype");
    // forward: src = img, dst = patches; inverse: src = patches, dst = img
    const void* src = inverse ? patches : img;
    void* dst = inverse ? (void*)img : patches;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_1d((int64_t)B * H * W * C, NT)), dim3(NT), 0, ST, src, dst, B, H, W, C,
                       P, inverse, dtype);
    return comat_check_launch("comat_patchify");
}

extern "C" int comat_embedding(const int64_t* ids, const void* table, void* out, int64_t n, int32_t dim, int64_t vocab,
                               int32_t dtype, void* stream) {
    COMAT_REQUIRE(ids && table && out && n > 0 && dim > 0 && vocab > 0 && dtype_ok(dtype), "comat_embedding: bad args");
    hipLaunchKernelGGL(embedding_kernel, dim3(grid_1d(n * dim, NT)), dim3(NT), 0, ST, ids, table, out, n, dim, vocab,
                       dtype);
    return comat_check_launch("comat_embedding");
}
