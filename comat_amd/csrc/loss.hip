// loss.hip — loss heads of the CoMat step: BLIP token cross-entropy, discriminator BCE head, and the
// attribute-concentration gather over captured cross-attention maps.
#include "common.h"

namespace {

constexpr int NT = 256;

// ---- cross entropy: one block per token row -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                    float* __restrict__ logp, float* __restrict__ row_lse,
                                                    float* __restrict__ row_loss, int V, int64_t ld,
                                                    int ignore_index, float ls) {
    __shared__ float sbuf[4];
    const int64_t t = blockIdx.x;
    const T* z = logits + t * ld;
    float m = -INFINITY;
    for (int v = threadIdx.x; v < V; v += NT) m = fmaxf(m, ldf<T>(z + v));
    m = block_max_256(m, sbuf);
    float se = 0.f, sz = 0.f;
    for (int v = threadIdx.x; v < V; v += NT) {
        const float zv = ldf<T>(z + v);
        se += __expf(zv - m);
        sz += zv;
    }
    se = block_sum_256(se, sbuf);
    sz = block_sum_256(sz, sbuf);
    if (threadIdx.x == 0) {
        const float lse = m + __logf(se);
        row_lse[t] = lse;
        const int64_t y = labels[t];
        if (y == ignore_index || y < 0 || y >= V) {
            logp[t] = 0.f;
            row_loss[t] = -1.0f;  // marks an ignored row (a CE loss is never negative)
        } else {
            const float lp = ldf<T>(z + y) - lse;
            logp[t] = lp;
            row_loss[t] = (1.0f - ls) * (-lp) + ls * (lse - sz / V);
        }
    }
}

// fixed-order sum over the token rows (deterministic; T is a few dozen)
__global__ void ce_final_kernel(const float* __restrict__ row_loss, int64_t T, float* __restrict__ loss_sum_cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float sum = 0.f, cnt = 0.f;
        for (int64_t t = 0; t < T; ++t)
            if (row_loss[t] >= 0.f) { sum += row_loss[t]; cnt += 1.f; }
        loss_sum_cnt[0] = sum;
        loss_sum_cnt[1] = cnt;
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                    const float* __restrict__ row_lse, T* __restrict__ dlogits, int V,
                                                    int64_t ld, int ignore_index, float ls,
                                                    const float* __restrict__ g_up,
                                                    const float* __restrict__ loss_sum_cnt) {
    const float gscale = g_up[0] / fmaxf(loss_sum_cnt[1], 1.0f);  // upstream gradient / number of valid tokens
    const int64_t t = blockIdx.x;
    const T* z = logits + t * ld;
    T* d = dlogits + t * ld;
    const int64_t y = labels[t];
    const bool valid = !(y == ignore_index || y < 0 || y >= V);
    const float lse = row_lse[t];
    const float sm = ls / V;
    for (int v = threadIdx.x; v < V; v += NT) {
        float g = 0.f;
        if (valid) {
            g = __expf(ldf<T>(z + v) - lse) - sm;
            if (v == y) g -= (1.0f - ls);
            g *= gscale;
        }
        stf<T>(d + v, g);
    }
}

// ---- discriminator head: Linear(4,1) + BCE-with-logits, mean over pixels ---------------------------------------------
__device__ __forceinline__ float bce_logits(float z, float t) {
    return fmaxf(z, 0.f) - z * t + log1pf(__expf(-fabsf(z)));
}

__global__ __launch_bounds__(NT) void disc_head_fwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ target, float* __restrict__ ws,
                                                           int64_t P, int64_t pps, int dt) {
    __shared__ float sbuf[4];
    float acc = 0.f;
    const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], bb = b[0];
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < P; p += (int64_t)gridDim.x * NT) {
        const float z = ld_dt(x, 4 * p, dt) * w0 + ld_dt(x, 4 * p + 1, dt) * w1 + ld_dt(x, 4 * p + 2, dt) * w2 +
                        ld_dt(x, 4 * p + 3, dt) * w3 + bb;
        acc += bce_logits(z, target[p / pps]);
    }
    acc = block_sum_256(acc, sbuf);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;  // reduced in fixed order by reduce_cols_kernel
}

// out[c] (+)= scale * sum_b ws[b*ncol + c], b in fixed order: the deterministic second stage of the loss reductions
__global__ __launch_bounds__(NT) void reduce_cols_kernel(const float* __restrict__ ws, int nparts, int ncol,
                                                         float scale, float* __restrict__ out, int accumulate) {
    __shared__ float sbuf[4];
    for (int c = 0; c < ncol; ++c) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < nparts; i += NT) acc += ws[(int64_t)i * ncol + c];
        acc = block_sum_256(acc, sbuf);
        if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + scale * acc;
    }
}

__global__ __launch_bounds__(NT) void disc_head_bwd_kernel(const void* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ g_up,
                                                           void* __restrict__ dx, float* __restrict__ ws,
                                                           int want_wgrad, int64_t P, int64_t pps, int dt) {
    __shared__ float sbuf[4];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, ab = 0.f;
    const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], bb = b[0];
    const float gs = g_up[0] / (float)P;  // upstream gradient is a device scalar (no host sync)
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < P; p += (int64_t)gridDim.x * NT) {
        const float x0 = ld_dt(x, 4 * p, dt), x1 = ld_dt(x, 4 * p + 1, dt), x2 = ld_dt(x, 4 * p + 2, dt),
                    x3 = ld_dt(x, 4 * p + 3, dt);
        const float z = x0 * w0 + x1 * w1 + x2 * w2 + x3 * w3 + bb;
        const float dz = gs * (1.0f / (1.0f + __expf(-z)) - target[p / pps]);
        if (dx) {
            st_dt(dx, 4 * p, dz * w0, dt);
            st_dt(dx, 4 * p + 1, dz * w1, dt);
            st_dt(dx, 4 * p + 2, dz * w2, dt);
            st_dt(dx, 4 * p + 3, dz * w3, dt);
        }
        a0 += dz * x0; a1 += dz * x1; a2 += dz * x2; a3 += dz * x3; ab += dz;
    }
    if (want_wgrad) {
        a0 = block_sum_256(a0, sbuf);
        a1 = block_sum_256(a1, sbuf);
        a2 = block_sum_256(a2, sbuf);
        a3 = block_sum_256(a3, sbuf);
        ab = block_sum_256(ab, sbuf);
        if (threadIdx.x == 0) {
            float* o = ws + (int64_t)blockIdx.x * 5;
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = ab;
        }
    }
}

// ---- attribute-concentration gather over one captured map [heads, npix, L] ----------------------------------------
// Grid (256-pixel chunk, head).  No atomics: per-(block, head, token) spatial sums and per-head token maps go to the
// workspace and are combined in a fixed order by attnmap_final_kernel (bit-reproducible, still fully parallel).
//   ws  = [nblk, heads, n_tok, 2] partial (masked sum, sum)   followed by   [heads, n_tok, npix] per-head values
__global__ __launch_bounds__(NT) void attnmap_fwd_kernel(const void* __restrict__ amap, const float* __restrict__ mask,
                                                         const int32_t* __restrict__ tok_idx,
                                                         const int32_t* __restrict__ tok_obj, float* __restrict__ ws,
                                                         int heads, int npix, int L, int n_tok, int dt) {
    __shared__ float sbuf[4];
    const int h = blockIdx.y;
    const int px = blockIdx.x * NT + threadIdx.x;
    float* wsv = ws + (int64_t)gridDim.x * heads * n_tok * 2;
    for (int t = 0; t < n_tok; ++t) {
        float v = 0.f, vm = 0.f;
        if (px < npix) {
            v = ld_dt(amap, ((int64_t)h * npix + px) * L + tok_idx[t], dt);
            vm = v * mask[(int64_t)tok_obj[t] * npix + px];
            wsv[((int64_t)h * n_tok + t) * npix + px] = v;
        }
        const float sn = block_sum_256(vm, sbuf);
        const float sd = block_sum_256(v, sbuf);
        if (threadIdx.x == 0) {
            float* o = ws + (((int64_t)blockIdx.x * heads + h) * n_tok + t) * 2;
            o[0] = sn;
            o[1] = sd;
        }
    }
}
// The same pass with the block's slice of the map staged through LDS.  The map is [heads, npix, L] with L = 77 token
// columns of which the loss reads a handful: fetching them one 2-byte element per thread touches every 64-byte segment
// of every row (measured: 4.5 MB moved for 262 KB used, 16 us at ~0.3 TB/s).  Here a block copies its RPB consecutive
// rows (one contiguous RPB * L * esize chunk) with coalesced 16-byte loads and picks the token columns out of LDS: the
// same bytes, but as one full-rate stream.  Same partial-sum layout and summation order as attnmap_fwd_kernel.
template <typename T, int RPB>
__global__ __launch_bounds__(NT) void attnmap_fwd_lds_kernel(const T* __restrict__ amap, const float* __restrict__ mask,
                                                             const int32_t* __restrict__ tok_idx,
                                                             const int32_t* __restrict__ tok_obj, float* __restrict__ ws,
                                                             int heads, int npix, int L, int n_tok) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    float* sbuf = (float*)tile;                   // 4 floats for the block reductions, then the tile (16-byte aligned)
    T* rows = (T*)(tile + 16);
    const int h = blockIdx.y;
    const int px0 = blockIdx.x * RPB;
    const int nrows = min(RPB, npix - px0);
    const int64_t nbytes = (int64_t)nrows * L * (int)sizeof(T);
    const char* src = (const char*)(amap + ((int64_t)h * npix + px0) * L);
    for (int64_t o = (int64_t)threadIdx.x * 16; o < nbytes; o += (int64_t)NT * 16)  // host guarantees nbytes % 16 == 0
        *(uint4*)((char*)rows + o) = *(const uint4*)(src + o);
    __syncthreads();
    const int px = px0 + threadIdx.x;
    const bool act = (int)threadIdx.x < nrows;
    float* wsv = ws + (int64_t)gridDim.x * heads * n_tok * 2;
    for (int t = 0; t < n_tok; ++t) {
        float v = 0.f, vm = 0.f;
        if (act) {
            v = ldf<T>(rows + (int64_t)threadIdx.x * L + tok_idx[t]);
            vm = v * mask[(int64_t)tok_obj[t] * npix + px];
            wsv[((int64_t)h * n_tok + t) * npix + px] = v;
        }
        const float sn = block_sum_256(vm, sbuf);
        const float sd = block_sum_256(v, sbuf);
        if (threadIdx.x == 0) {
            float* o = ws + (((int64_t)blockIdx.x * heads + h) * n_tok + t) * 2;
            o[0] = sn;
            o[1] = sd;
        }
    }
}

// num[h,t] += sum_blk partial; den likewise; avg[t,px] += (1/heads) * sum_h value[h,t,px]   (fixed orders)
__global__ __launch_bounds__(NT) void attnmap_final_kernel(const float* __restrict__ ws, int nblk, int heads, int n_tok,
                                                           int npix, float* __restrict__ num, float* __restrict__ den,
                                                           float* __restrict__ avg) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int ht = heads * n_tok;
    if (i < ht) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < nblk; ++k) {
            a += ws[((int64_t)k * ht + i) * 2];
            b += ws[((int64_t)k * ht + i) * 2 + 1];
        }
        num[i] += a;
        den[i] += b;
    }
    const float* wsv = ws + (int64_t)nblk * ht * 2;
    const int64_t tp = (int64_t)n_tok * npix;
    if (i < tp) {
        float a = 0.f;
        for (int h = 0; h < heads; ++h) a += wsv[(int64_t)h * tp + i];
        avg[i] += a / heads;
    }
}

__global__ __launch_bounds__(NT) void attnmap_bwd_kernel(const float* __restrict__ g_num,
                                                         const float* __restrict__ g_den,
                                                         const float* __restrict__ g_avg,
                                                         const float* __restrict__ mask,
                                                         const int32_t* __restrict__ tok_idx,
                                                         const int32_t* __restrict__ tok_obj, void* __restrict__ damap,
                                                         int heads, int npix, int L, int n_tok, int dt) {
    const int h = blockIdx.y;
    const int px = blockIdx.x * NT + threadIdx.x;
    if (px >= npix) return;
    const float inv_h = 1.0f / heads;
    for (int t = 0; t < n_tok; ++t) {  // sequential per (h, px): repeated token columns accumulate safely
        const int64_t i = ((int64_t)h * npix + px) * L + tok_idx[t];
        float g = g_num[h * n_tok + t] * mask[(int64_t)tok_obj[t] * npix + px] + g_den[h * n_tok + t];
        if (g_avg) g += g_avg[(int64_t)t * npix + px] * inv_h;
        st_dt(damap, i, ld_dt(damap, i, dt) + g, dt);
    }
}

// Backward through LDS: the block builds its RPB dense rows of dA (zeros + the few token columns) in LDS and writes
// them out as one coalesced stream, so the caller does NOT zero-fill dA (one pass over the map instead of a memset plus
// scattered 2-byte read-modify-writes).
template <typename T, int RPB>
__global__ __launch_bounds__(NT) void attnmap_bwd_lds_kernel(const float* __restrict__ g_num,
                                                             const float* __restrict__ g_den,
                                                             const float* __restrict__ g_avg,
                                                             const float* __restrict__ mask,
                                                             const int32_t* __restrict__ tok_idx,
                                                             const int32_t* __restrict__ tok_obj, T* __restrict__ damap,
                                                             int heads, int npix, int L, int n_tok) {
    extern __shared__ __attribute__((aligned(16))) char tile[];
    float* acc = (float*)tile;  // [RPB][L] fp32 accumulation (repeated token columns add up), then packed in place
    const int h = blockIdx.y;
    const int px0 = blockIdx.x * RPB;
    const int nrows = min(RPB, npix - px0);
    for (int i = threadIdx.x; i < nrows * L; i += NT) acc[i] = 0.f;
    __syncthreads();
    const int px = px0 + threadIdx.x;
    if ((int)threadIdx.x < nrows) {
        const float inv_h = 1.0f / heads;
        for (int t = 0; t < n_tok; ++t) {  // sequential per pixel: deterministic, repeated columns accumulate
            float g = g_num[h * n_tok + t] * mask[(int64_t)tok_obj[t] * npix + px] + g_den[h * n_tok + t];
            if (g_avg) g += g_avg[(int64_t)t * npix + px] * inv_h;
            acc[threadIdx.x * L + tok_idx[t]] += g;
        }
    }
    __syncthreads();
    T* dst = damap + ((int64_t)h * npix + px0) * L;
    if (sizeof(T) == 4) {
        const int64_t nbytes = (int64_t)nrows * L * 4;
        for (int64_t o = (int64_t)threadIdx.x * 16; o < nbytes; o += (int64_t)NT * 16)
            *(uint4*)((char*)dst + o) = *(const uint4*)((const char*)acc + o);
    } else {
        const int n8 = nrows * L / 8;  // host guarantees (nrows * L) % 8 == 0
        for (int i = threadIdx.x; i < n8; i += NT) {
            union { uint4 u; bf16_t hh[8]; } pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk.hh[e] = f32_to_bf16(acc[i * 8 + e]);
            *(uint4*)((bf16_t*)dst + (int64_t)i * 8) = pk.u;
        }
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int comat_cross_entropy_fwd(const void* logits, const int64_t* labels, float* logp, float* row_lse,
                                       float* row_loss, float* loss_sum_cnt, int64_t T, int32_t V, int64_t ld,
                                       int32_t ignore_index, float label_smoothing, int32_t dtype, void* stream) {
    COMAT_REQUIRE(logits && labels && logp && row_lse && row_loss && loss_sum_cnt,
                  "comat_cross_entropy_fwd: null pointer");
    COMAT_REQUIRE(T > 0 && V > 0 && ld >= V && T < (1ll << 31) && dtype_ok(dtype), "comat_cross_entropy_fwd: bad args");
    if (dtype == COMAT_BF16)
        hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3((unsigned)T), dim3(NT), 0, ST, (const bf16_t*)logits, labels, logp,
                           row_lse, row_loss, V, ld, ignore_index, label_smoothing);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3((unsigned)T), dim3(NT), 0, ST, (const float*)logits, labels, logp,
                           row_lse, row_loss, V, ld, ignore_index, label_smoothing);
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(64), 0, ST, (const float*)row_loss, T, loss_sum_cnt);
    return comat_check_launch("comat_cross_entropy_fwd");
}

extern "C" int comat_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* row_lse, void* dlogits,
                                       int64_t T, int32_t V, int64_t ld, int32_t ignore_index, float label_smoothing,
                                       const float* g_up, const float* loss_sum_cnt, int32_t dtype, void* stream) {
    COMAT_REQUIRE(logits && labels && row_lse && dlogits && g_up && loss_sum_cnt,
                  "comat_cross_entropy_bwd: null pointer");
    COMAT_REQUIRE(T > 0 && V > 0 && ld >= V && T < (1ll << 31) && dtype_ok(dtype), "comat_cross_entropy_bwd: bad args");
    if (dtype == COMAT_BF16)
        hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3((unsigned)T), dim3(NT), 0, ST, (const bf16_t*)logits, labels,
                           row_lse, (bf16_t*)dlogits, V, ld, ignore_index, label_smoothing, g_up, loss_sum_cnt);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3((unsigned)T), dim3(NT), 0, ST, (const float*)logits, labels, row_lse,
                           (float*)dlogits, V, ld, ignore_index, label_smoothing, g_up, loss_sum_cnt);
    return comat_check_launch("comat_cross_entropy_bwd");
}

extern "C" int comat_disc_head_fwd(const void* x, const float* w, const float* b, const float* target, float* loss,
                                   float* ws, int64_t P, int64_t pix_per_sample, int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && w && b && target && loss && ws, "comat_disc_head_fwd: null pointer");
    COMAT_REQUIRE(P > 0 && pix_per_sample > 0 && dtype_ok(dtype), "comat_disc_head_fwd: bad args");
    const int parts = grid_1d(P, NT, 512);
    hipLaunchKernelGGL(disc_head_fwd_kernel, dim3(parts), dim3(NT), 0, ST, x, w, b, target, ws, P, pix_per_sample,
                       dtype);
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(NT), 0, ST, (const float*)ws, parts, 1, 1.0f / (float)P, loss,
                       0);
    return comat_check_launch("comat_disc_head_fwd");
}

extern "C" int comat_disc_head_bwd(const void* x, const float* w, const float* b, const float* target,
                                   const float* g_up, void* dx, float* dwb, float* ws, int64_t P,
                                   int64_t pix_per_sample, int32_t dtype, void* stream) {
    COMAT_REQUIRE(x && w && b && target && g_up, "comat_disc_head_bwd: null pointer");
    COMAT_REQUIRE(!dwb || ws, "comat_disc_head_bwd: weight gradients need the workspace");
    COMAT_REQUIRE(P > 0 && pix_per_sample > 0 && dtype_ok(dtype), "comat_disc_head_bwd: bad args");
    const int parts = grid_1d(P, NT, 512);
    hipLaunchKernelGGL(disc_head_bwd_kernel, dim3(parts), dim3(NT), 0, ST, x, w, b, target, g_up, dx, ws,
                       dwb ? 1 : 0, P, pix_per_sample, dtype);
    if (dwb) hipLaunchKernelGGL(reduce_cols_kernel, dim3(1), dim3(NT), 0, ST, (const float*)ws, parts, 5, 1.0f, dwb, 1);
    return comat_check_launch("comat_disc_head_bwd");
}

extern "C" int comat_attnmap_gather_fwd(const void* amap, const float* mask, const int32_t* tok_idx,
                                        const int32_t* tok_obj, float* num, float* den, float* avg, float* ws,
                                        int32_t heads, int32_t npix, int32_t L, int32_t n_tok, int32_t dtype,
                                        void* stream) {
    COMAT_REQUIRE(amap && mask && tok_idx && tok_obj && num && den && avg && ws,
                  "comat_attnmap_gather_fwd: null pointer");
    COMAT_REQUIRE(heads > 0 && heads <= 65535 && npix > 0 && L > 0 && n_tok > 0 && dtype_ok(dtype),
                  "comat_attnmap_gather_fwd: bad args");
    // LDS-staged gather when every block's row chunk is 16-byte aligned and a whole number of 16-byte vectors
    constexpr int RPB16 = 256, RPB32 = 128;
    const int es = dtype == COMAT_BF16 ? 2 : 4, rpb = dtype == COMAT_BF16 ? RPB16 : RPB32;
    const bool staged = ((uintptr_t)amap % 16) == 0 && ((int64_t)npix * L * es) % 16 == 0 && ((int64_t)rpb * L * es) % 16 == 0 &&
                        ((int64_t)(npix % rpb) * L * es) % 16 == 0 && (int64_t)rpb * L * es + 16 <= 64 * 1024;
    const int nblk = staged ? (npix + rpb - 1) / rpb : (npix + NT - 1) / NT;
    if (staged && dtype == COMAT_BF16)
        hipLaunchKernelGGL((attnmap_fwd_lds_kernel<bf16_t, RPB16>), dim3(nblk, heads), dim3(NT), 16 + RPB16 * L * 2, ST,
                           (const bf16_t*)amap, mask, tok_idx, tok_obj, ws, heads, npix, L, n_tok);
    else if (staged)
        hipLaunchKernelGGL((attnmap_fwd_lds_kernel<float, RPB32>), dim3(nblk, heads), dim3(NT), 16 + RPB32 * L * 4, ST,
                           (const float*)amap, mask, tok_idx, tok_obj, ws, heads, npix, L, n_tok);
    else
        hipLaunchKernelGGL(attnmap_fwd_kernel, dim3(nblk, heads), dim3(NT), 0, ST, amap, mask, tok_idx, tok_obj, ws, heads,
                           npix, L, n_tok, dtype);
    const int64_t work = (int64_t)n_tok * npix > (int64_t)heads * n_tok ? (int64_t)n_tok * npix : (int64_t)heads * n_tok;
    hipLaunchKernelGGL(attnmap_final_kernel, dim3((unsigned)cdiv64(work, NT)), dim3(NT), 0, ST, (const float*)ws, nblk,
                       heads, n_tok, npix, num, den, avg);
    return comat_check_launch("comat_attnmap_gather_fwd");
}

extern "C" int comat_attnmap_gather_bwd(const float* g_num, const float* g_den, const float* g_avg, const float* mask,
                                        const int32_t* tok_idx, const int32_t* tok_obj, void* damap, int32_t heads,
                                        int32_t npix, int32_t L, int32_t n_tok, int32_t dtype, void* stream) {
    COMAT_REQUIRE(g_num && g_den && mask && tok_idx && tok_obj && damap, "comat_attnmap_gather_bwd: null pointer");
    COMAT_REQUIRE(heads > 0 && heads <= 65535 && npix > 0 && L > 0 && n_tok > 0 && dtype_ok(dtype),
                  "comat_attnmap_gather_bwd: bad args");
    // dense-tile writer through LDS (it writes EVERY element of dA) when the row chunks vectorise; otherwise zero-fill
    // here and scatter
    constexpr int RPBB = 128;  // fp32 accumulation tile: 128 rows x L floats (39 KB at L = 77)
    const int es = dtype == COMAT_BF16 ? 2 : 4;
    const bool staged = ((uintptr_t)damap % 16) == 0 && ((int64_t)npix * L * es) % 16 == 0 && ((int64_t)RPBB * L * es) % 16 == 0 &&
                        ((int64_t)(npix % RPBB) * L * es) % 16 == 0 && (int64_t)RPBB * L * 4 <= 64 * 1024;
    if (staged && dtype == COMAT_BF16)
        hipLaunchKernelGGL((attnmap_bwd_lds_kernel<bf16_t, RPBB>), dim3((npix + RPBB - 1) / RPBB, heads), dim3(NT),
                           RPBB * L * 4, ST, g_num, g_den, g_avg, mask, tok_idx, tok_obj, (bf16_t*)damap, heads, npix, L, n_tok);
    else if (staged)
        hipLaunchKernelGGL((attnmap_bwd_lds_kernel<float, RPBB>), dim3((npix + RPBB - 1) / RPBB, heads), dim3(NT),
                           RPBB * L * 4, ST, g_num, g_den, g_avg, mask, tok_idx, tok_obj, (float*)damap, heads, npix, L, n_tok);
    else {
        if (hipMemsetAsync(damap, 0, (size_t)heads * npix * L * es, ST) != hipSuccess) {
            comat_set_error("comat_attnmap_gather_bwd: memset failed");
            return COMAT_ELAUNCH;
        }
        hipLaunchKernelGGL(attnmap_bwd_kernel, dim3((npix + NT - 1) / NT, heads), dim3(NT), 0, ST, g_num, g_den, g_avg, mask,
                           tok_idx, tok_obj, damap, heads, npix, L, n_tok, dtype);
    }
    return comat_check_launch("comat_attnmap_gather_bwd");
}
