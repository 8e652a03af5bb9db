// Batch normalisation (+ReLU) over the rows of a channels-last matrix x[M][C] (row stride ld).
//
// Serves BatchNorm2d(32) inside the corrector g (gridnet_models.py:134-146; a channels-last
// [B][H][W][C] grid is an [M=B*H*W][C] matrix) and BatchNorm1d(100|50) of the count MLP
// (Tutorial_visium_count.ipynb cell 12).  torch defaults: eps 1e-5, momentum 0.1, biased variance for
// normalisation, unbiased for running_var.
//
// Statistics are two-pass (mean, then centred second moment) with fixed-order partial slabs, so results
// are deterministic and free of E[x^2]-E[x]^2 cancellation.  Reductions: per-thread strided rows ->
// LDS across the 4 row-lanes of a 256-thread workgroup -> slab per workgroup -> fixed-order final sum.
// HBM-bound: 4 B read per element per pass.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 256;
constexpr int MAX_SLABS = 512;      // slabs per reduction: more rows per workgroup instead of more slabs

inline int slab_count(long M) {
    const long nb = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    return (int)(nb < MAX_SLABS ? nb : MAX_SLABS);
}

// Fixed-order (b = 0, 1, 2, ...) sum of n values p[b * stride] with the LOADS issued 16 at a time: the same arithmetic as a
// plain loop, without one memory round trip per term (a serial loop over 128 slabs is ~10 us; a DenseNet training step at
// batch 32 runs 121 BatchNorms forward and backward and mostly waited for such chains).
__device__ __forceinline__ float seq_sum(const float* __restrict__ p, int n, size_t stride) {
    float s = 0.f;
    int b = 0;
    for (; b + 16 <= n; b += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(b + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; b < n; ++b) s += p[(size_t)b * stride];
    return s;
}
__device__ __forceinline__ float4 seq_sum4(const float* __restrict__ p, int n, size_t stride) {       // 4 adjacent columns
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = 0;
    for (; b + 8 <= n; b += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(b + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; b < n; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    return s;
}

// colsum_kernel from 16-B loads, SAME summation order (per channel: four row-lanes, each adding its rows r0 + rl + 4k in
// order, combined as (0+1)+(2+3)), so the slabs are bit-identical to the scalar kernel's.  64 threads = 16 channel quads x 4
// row lanes; a lane's loads go out 16 rows at a time.
template <int MODE>
__global__ __launch_bounds__(64) void colsum_v4_kernel(const float* __restrict__ x, long ld, long M, int C,
                                                       const float* __restrict__ sum_partial, int nprev,
                                                       float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    const int c = blockIdx.y * 64 + 4 * cq;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float mu[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) {
            const float4 s = seq_sum4(sum_partial + c, nprev, (size_t)C);
            mu[0] = s.x / (float)M; mu[1] = s.y / (float)M; mu[2] = s.z / (float)M; mu[3] = s.w / (float)M;
        }
        for (long r0 = (long)blockIdx.x * ROWS_PER_BLOCK; r0 < M; r0 += (long)gridDim.x * ROWS_PER_BLOCK) {
            const long r1 = min(r0 + ROWS_PER_BLOCK, M);
            for (long r = r0 + rl; r < r1; r += 64) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const long rr = r + 4 * u;
                    if (rr < r1) v[u] = *reinterpret_cast<const float4*>(x + rr * ld + c);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (r + 4 * u < r1) {
                        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (MODE == 0) a[j] += e[j];
                            else { const float d = e[j] - mu[j]; a[j] = fmaf(d, d, a[j]); }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a[j];
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            partial[(size_t)blockIdx.x * C + c + j] =
                (red[0][4 * cq + j] + red[1][4 * cq + j]) + (red[2][4 * cq + j] + red[3][4 * cq + j]);
    }
}

// partial[blockIdx.x][c] = sum over the block's rows of f(x[r][c]); MODE 0: x ; 1: (x-mean[c])^2
template <int MODE>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, long M, int C,
                                                     const float* __restrict__ sum_partial, int nprev,
                                                     float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float mean = 0.f;
    if (MODE == 1 && c < C) {
        mean = seq_sum(sum_partial + c, nprev, (size_t)C) / (float)M;
    }
    float acc = 0.f;
    if (c < C) {
        for (long r0 = (long)blockIdx.x * ROWS_PER_BLOCK; r0 < M; r0 += (long)gridDim.x * ROWS_PER_BLOCK) {
            const long r1 = min(r0 + ROWS_PER_BLOCK, M);
            for (long r = r0 + rl; r < r1; r += 4) {
                const float v = x[r * ld + c];
                if (MODE == 0) acc += v;
                else { const float d = v - mean; acc = fmaf(d, d, acc); }
            }
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C)
        partial[(size_t)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// mean/var from the slabs; running-stat update; folded scale/shift; saved mean & invstd for backward
__global__ void bn_finalize_train_kernel(const float* __restrict__ sum_partial, const float* __restrict__ m2_partial,
                                         int nblk, long M, int C, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float momentum, float eps,
                                         float* __restrict__ scale, float* __restrict__ shift,
                                         float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    const float s = seq_sum(sum_partial + c, nblk, (size_t)C), m2 = seq_sum(m2_partial + c, nblk, (size_t)C);
    const float mean = s / (float)M;
    const float var = m2 / (float)M;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = bt - mean * g * invstd;
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void bn_fold_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float eps, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = bt - running_mean[c] * g * invstd;
    if (save_mean) save_mean[c] = running_mean[c];
    if (save_invstd) save_invstd[c] = invstd;
}

// y = [relu](x*scale + shift)
__global__ __launch_bounds__(256) void scale_shift_relu_kernel(const float* __restrict__ x, long ldx,
                                                               float* __restrict__ y, long ldy, long M, int C,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu) {
    const long total = M * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        float v = fmaf(x[r * ldx + c], scale[c], shift[c]);
        if (relu) v = fmaxf(v, 0.f);
        y[r * ldy + c] = v;
    }
}

// backward pass 1: partial[blk][0][c] = sum dz ; partial[blk][1][c] = sum dz * xhat
// dz = dy * (relu ? 1[x*scale+shift > 0] : 1), xhat = (x-mean)*invstd
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, long lddy,
                                                             const float* __restrict__ x, long ldx, long M, int C,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int relu,
                                                             float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    float a1 = 0.f, a2 = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        for (long r0 = (long)blockIdx.x * ROWS_PER_BLOCK; r0 < M; r0 += (long)gridDim.x * ROWS_PER_BLOCK) {
            const long r1 = min(r0 + ROWS_PER_BLOCK, M);
            for (long r = r0 + rl; r < r1; r += 4) {
                const float xv = x[r * ldx + c];
                float dz = dy[r * lddy + c];
                if (relu && fmaf(xv, sc, sh) <= 0.f) dz = 0.f;
                a1 += dz;
                a2 = fmaf(dz, (xv - mu) * is, a2);
            }
        }
    }
    red[0][rl][cl] = a1;
    red[1][rl][cl] = a2;
    __syncthreads();
    if (rl == 0 && c < C) {
        partial[((size_t)blockIdx.x * 2 + 0) * C + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        partial[((size_t)blockIdx.x * 2 + 1) * C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

// bn_bwd_partial_kernel from 16-B loads with the same summation order (see colsum_v4_kernel): bit-identical slabs
__global__ __launch_bounds__(64) void bn_bwd_partial_v4_kernel(const float* __restrict__ dy, long lddy,
                                                               const float* __restrict__ x, long ldx, long M, int C,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, int relu,
                                                               float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    const int c = blockIdx.y * 64 + 4 * cq;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float4 sc4 = *reinterpret_cast<const float4*>(scale + c), sh4 = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu4 = *reinterpret_cast<const float4*>(mean + c), is4 = *reinterpret_cast<const float4*>(invstd + c);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
        for (long r0 = (long)blockIdx.x * ROWS_PER_BLOCK; r0 < M; r0 += (long)gridDim.x * ROWS_PER_BLOCK) {
            const long r1 = min(r0 + ROWS_PER_BLOCK, M);
            for (long r = r0 + rl; r < r1; r += 32) {
                float4 xv[8], dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long rr = r + 4 * u;
                    if (rr < r1) {
                        xv[u] = *reinterpret_cast<const float4*>(x + rr * ldx + c);
                        dv[u] = *reinterpret_cast<const float4*>(dy + rr * lddy + c);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (r + 4 * u < r1) {
                        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                        const float de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float dz = de[j];
                            if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                            a1[j] += dz;
                            a2[j] = fmaf(dz, (xe[j] - mu[j]) * is[j], a2[j]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[0][rl][4 * cq + j] = a1[j]; red[1][rl][4 * cq + j] = a2[j]; }
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = 4 * cq + j;
            partial[((size_t)blockIdx.x * 2 + 0) * C + c + j] = (red[0][0][cc] + red[0][1][cc]) + (red[0][2][cc] + red[0][3][cc]);
            partial[((size_t)blockIdx.x * 2 + 1) * C + c + j] = (red[1][0][cc] + red[1][1][cc]) + (red[1][2][cc] + red[1][3][cc]);
        }
    }
}

// backward pass 2: sums -> dgamma/dbeta (optionally accumulated), sums[2][C] kept for the dx pass.
// 64 channels x 4 slab-lanes per workgroup; each lane sums every 4th slab in order, the four partial sums are then
// added in a fixed order (deterministic).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ partial, int nblk, int C,
                                                            float* __restrict__ sums, float* dgamma, float* dbeta,
                                                            int accumulate) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        for (int b = sl; b < nblk; b += 4) {
            s1 += partial[((size_t)b * 2 + 0) * C + c];
            s2 += partial[((size_t)b * 2 + 1) * C + c];
        }
    }
    red[0][sl][cl] = s1;
    red[1][sl][cl] = s2;
    __syncthreads();
    if (sl == 0 && c < C) {
        s1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        s2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        sums[c] = s1;
        sums[C + c] = s2;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
    }
}

// backward pass 3: dx.  training: scale*(dz - s1/M - xhat*s2/M) ; eval: scale*dz   (scale = gamma*invstd)
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dy, long lddy,
                                                        const float* __restrict__ x, long ldx,
                                                        float* __restrict__ dx, long lddx, long M, int C,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ sums, int relu, int training,
                                                        int dx_accumulate) {
    const long total = M * C;
    const float invM = 1.0f / (float)M;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        const float xv = x[r * ldx + c];
        float dz = dy[r * lddy + c];
        if (relu && fmaf(xv, scale[c], shift[c]) <= 0.f) dz = 0.f;
        float g = dz;
        if (training) {
            const float xhat = (xv - mean[c]) * invstd[c];
            g = dz - sums[c] * invM - xhat * sums[C + c] * invM;
        }
        const float o = scale[c] * g;
        dx[r * lddx + c] = dx_accumulate ? dx[r * lddx + c] + o : o;
    }
}


// 4 channels per thread, 16-B accesses (C % 4 == 0, aligned pointers and leading dimensions)
__global__ __launch_bounds__(256) void bn_bwd_dx_vec4_kernel(const float* __restrict__ dy, long lddy,
                                                             const float* __restrict__ x, long ldx,
                                                             float* __restrict__ dx, long lddx, long M, int C4,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ sums, int relu, int training,
                                                             int dx_accumulate) {
    const long total = M * C4;
    const float invM = 1.0f / (float)M;
    const int C = 4 * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C4;
        const int c = 4 * (int)(idx - r * C4);
        const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 dv = *reinterpret_cast<const float4*>(dy + r * lddy + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
        const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float dz = ds[e];
            if (relu && fmaf(xs[e], scs[e], shs[e]) <= 0.f) dz = 0.f;
            float g = dz;
            if (training) {
                const float xhat = (xs[e] - mean[c + e]) * invstd[c + e];
                g = dz - sums[c + e] * invM - xhat * sums[C + c + e] * invM;
            }
            o[e] = scs[e] * g;
        }
        float4* dst = reinterpret_cast<float4*>(dx + r * lddx + c);
        if (dx_accumulate) {
            const float4 old = *dst;
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Eval-mode (running-statistics) BN+ReLU backward in ONE pass over the data: dx (+)= scale*dz needs no batch sums, so
// the column sums for dgamma/dbeta are accumulated by the same threads that write dx.  Thread = 4 channels x a strided
// set of rows; slab layout as bn_bwd_partial_kernel ([blk][2][C]).
__global__ __launch_bounds__(256) void bn_bwd_eval_fused_kernel(const float* __restrict__ dy, long lddy,
                                                                const float* __restrict__ x, long ldx,
                                                                float* __restrict__ dx, long lddx, long M, int C,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int relu,
                                                                int dx_accumulate, float* __restrict__ partial,
                                                                int pool_S = 0) {
    // pool_S > 0 (a transition's norm -> relu in front of its 2x2 average pool, densenet.py:50-54): dy is the gradient of the
    // POOLED map [imgs * (pool_S/2)^2][C]; row r of x (position (y, x) of an S x S map) takes a quarter of its window's value -
    // the unpooled gradient (a full-size pass: write 4x, read 4x) is never materialised.
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;       // 16 channel-quads x 16 row lanes
    const int c = 4 * (blockIdx.y * 16 + cl);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
        const float scs[4] = {sc.x, sc.y, sc.z, sc.w}, shs[4] = {sh.x, sh.y, sh.z, sh.w};
        const float mus[4] = {mu.x, mu.y, mu.z, mu.w}, iss[4] = {is.x, is.y, is.z, is.w};
        for (long r = (long)blockIdx.x * 16 + rl; r < M; r += (long)gridDim.x * 16) {
            const float4 xv = *reinterpret_cast<const float4*>(x + r * ldx + c);
            long rd = r;
            float dscale = 1.f;
            if (pool_S > 0) {
                const int S = pool_S, So = S >> 1;
                const long img = r / ((long)S * S);
                const int rem = (int)(r - img * S * S);
                const int yy = rem / S, xx = rem - yy * S;
                rd = (img * So + (yy >> 1)) * So + (xx >> 1);
                dscale = ((yy >> 1) < So && (xx >> 1) < So) ? 0.25f : 0.f;         // (odd S: the last row / column is not pooled)
                if (dscale == 0.f) rd = 0;
            }
            float4 dv = *reinterpret_cast<const float4*>(dy + rd * lddy + c);
            if (pool_S > 0) dv = make_float4(dscale * dv.x, dscale * dv.y, dscale * dv.z, dscale * dv.w);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float ds[4] = {dv.x, dv.y, dv.z, dv.w}, o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xe = xs[e];
                if (relu == 2) {                            // x holds a = relu(scale x + shift): mask = a > 0, and where
                    if (xe <= 0.f) ds[e] = 0.f;             // the mask is set x = (a - shift) / scale (scale != 0)
                    xe = (xe - shs[e]) / scs[e];
                } else if (relu && fmaf(xe, scs[e], shs[e]) <= 0.f) {
                    ds[e] = 0.f;
                }
                s1[e] += ds[e];
                s2[e] = fmaf(ds[e], (xe - mus[e]) * iss[e], s2[e]);
                o[e] = scs[e] * ds[e];
            }
            if (dx) {
                float4* dst = reinterpret_cast<float4*>(dx + r * lddx + c);
                if (dx_accumulate) {
                    const float4 old = *dst;
                    o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
                }
                *dst = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][rl][4 * cl + e] = s1[e]; red[1][rl][4 * cl + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int q = threadIdx.x >> 6, cc = threadIdx.x & 63;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += red[q][j][cc];
        const int cg = blockIdx.y * 64 + cc;
        if (cg < C) partial[((size_t)blockIdx.x * 2 + q) * C + cg] = a;
    }
}

// out[c] = fixed-order sum of the slabs (used for bias gradients)
__global__ void slab_reduce_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out,
                                   int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = seq_sum(partial + c, nblk, (size_t)C);
    out[c] = accumulate ? out[c] + s : s;
}


// ------------------------------------------------------------------------------------------------ small-M single-launch forms
// A DenseNet training step at batch 32 (BASELINE config 2) runs 121 BatchNorms forward and backward on matrices of 512 -
// 32 768 rows; three launches each way (column sums, second moments, finalize | partial sums, reduce, dx) are latency, not
// work: 18 / 28 us per BatchNorm.  For M <= BN_SMALL_M a workgroup of 16 channels x 256 row lanes does the whole thing - it
// walks its column strip (<= 512 KB, L2-hot on the second walk) with 8 rows per lane in flight, reduces through LDS in a
// fixed order and finishes: one launch, C / 16 workgroups.  Same two-pass arithmetic as the slab kernels; only the (fixed)
// summation order differs.  (64 row lanes - 8 dependent load rounds per walk at M = 2048 - ran SLOWER than the three launches.)
constexpr long BN_SMALL_M = 4992;    // one Visium grid (g, the count MLP) and blocks 3-4 of a batch of 32 patches; at 8192 rows a strip is 1 MB per workgroup and C / 16 workgroups are too few CUs
constexpr int BN_SL = 256;                                     // row lanes

// fixed-order sum over the row lanes of red[lane][16 channels]: 64 threads (4 groups x 16 channels) add 64 lanes each
// (four interleaved chains, combined pairwise), then thread t < 16 adds the 4 groups pairwise.  Two barriers inside; every thread must call it.
__device__ __forceinline__ float small_colsum(float (*red)[16], float (*grp)[16], int t) {
    if (t < 64) {                                              // 64 lanes as 4 interleaved chains, combined pairwise
        const int ch = t & 15, gq = t >> 4;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
        for (int l = 0; l < BN_SL / 4; l += 4) {
            const int b = gq * (BN_SL / 4) + l;
            s0 += red[b][ch]; s1 += red[b + 1][ch]; s2 += red[b + 2][ch]; s3 += red[b + 3][ch];
        }
        grp[gq][ch] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    float tot = 0.f;
    if (t < 16) tot = (grp[0][t] + grp[1][t]) + (grp[2][t] + grp[3][t]);
    __syncthreads();
    return tot;
}

// KEEP (M <= 8 BN_SL = 2048 rows: blocks 3-4 of a batch of 32 patches): a lane's <= 8 rows stay in registers between the two
// walks - no second round of loads.
template <bool KEEP>
__global__ __launch_bounds__(1024) void bn_train_stats_small_kernel(
    const float* __restrict__ x, long ld, long M, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
    float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ y = nullptr, long ldy = 0, int relu = 0) {
    // y != NULL: also y = [relu](scale x + shift) - the apply pass of the BatchNorm rides on the statistics launch
    __shared__ float red[BN_SL][16];
    __shared__ float grp[4][16];
    __shared__ float bc[16];
    __shared__ float ss[2][16];
    const int t = threadIdx.x, cq = t & 3, rl = t >> 2;
    const int c = blockIdx.x * 16 + 4 * cq;
    const bool on = c < C;                                     // 4 | C: a quad is in or out as a whole
    const float* px = x + (on ? c : 0);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    float4 kv[8];                                              // KEEP: this lane's rows rl + BN_SL u
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = rl + BN_SL * u;
            kv[u] = *reinterpret_cast<const float4*>(px + (rr < M ? rr : 0) * ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rl + BN_SL * u < M) { a[0] += kv[u].x; a[1] += kv[u].y; a[2] += kv[u].z; a[3] += kv[u].w; }
    } else
    for (long r = rl; r < M; r += BN_SL * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = r + BN_SL * u;
            v[u] = *reinterpret_cast<const float4*>(px + (rr < M ? rr : r) * ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + BN_SL * u < M) { a[0] += v[u].x; a[1] += v[u].y; a[2] += v[u].z; a[3] += v[u].w; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a[j];
    __syncthreads();
    const float tot = small_colsum(red, grp, t);
    if (t < 16) bc[t] = tot / (float)M;
    __syncthreads();
    const float mu[4] = {bc[4 * cq], bc[4 * cq + 1], bc[4 * cq + 2], bc[4 * cq + 3]};
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rl + BN_SL * u < M) {
                const float d0 = kv[u].x - mu[0], d1 = kv[u].y - mu[1], d2 = kv[u].z - mu[2], d3 = kv[u].w - mu[3];
                q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]); q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
            }
    } else
    for (long r = rl; r < M; r += BN_SL * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = r + BN_SL * u;
            v[u] = *reinterpret_cast<const float4*>(px + (rr < M ? rr : r) * ld);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r + BN_SL * u < M) {
                const float d0 = v[u].x - mu[0], d1 = v[u].y - mu[1], d2 = v[u].z - mu[2], d3 = v[u].w - mu[3];
                q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]); q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = q[j];
    __syncthreads();
    const float m2 = small_colsum(red, grp, t);
    if (blockIdx.x == 0 && t == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (t < 16) {
        const int cc = blockIdx.x * 16 + t;
        if (cc < C) {
            const float mean = bc[t];
            const float var = m2 / (float)M;
            const float invstd = 1.0f / sqrtf(var + eps);
            const float g = gamma ? gamma[cc] : 1.f, bt = beta ? beta[cc] : 0.f;
            scale[cc] = g * invstd;
            shift[cc] = bt - mean * g * invstd;
            ss[0][t] = g * invstd;
            ss[1][t] = bt - mean * g * invstd;
            save_mean[cc] = mean;
            save_invstd[cc] = invstd;
            if (running_mean) running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * mean;
            if (running_var) {
                const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
                running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * unbiased;
            }
        }
    }
    if (!y) return;
    __syncthreads();
    if (!on) return;
    const float sc[4] = {ss[0][4 * cq], ss[0][4 * cq + 1], ss[0][4 * cq + 2], ss[0][4 * cq + 3]};
    const float sh[4] = {ss[1][4 * cq], ss[1][4 * cq + 1], ss[1][4 * cq + 2], ss[1][4 * cq + 3]};
    auto apply = [&](const float4& v, long r) {
        float o[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3])};
        if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
    };
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rl + BN_SL * u < M) apply(kv[u], rl + BN_SL * u);
    } else {
        for (long r = rl; r < M; r += BN_SL * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long rr = r + BN_SL * u;
                v[u] = *reinterpret_cast<const float4*>(px + (rr < M ? rr : r) * ld);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + BN_SL * u < M) apply(v[u], r + BN_SL * u);
        }
    }
}

// backward of y = [relu](bn(x)), batch or running statistics, in one launch: sums, then dx (+)=.  KEEP as above (x and dy rows)
template <bool KEEP>
__global__ __launch_bounds__(1024) void bn_bwd_small_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, float* __restrict__ dx, long lddx, long M,
    int C, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* dgamma, float* dbeta, int relu, int training, int accumulate,
    int dx_accumulate) {
    __shared__ float red[BN_SL][16];
    __shared__ float grp[4][16];
    __shared__ float bc[2][16];
    const int t = threadIdx.x, cq = t & 3, rl = t >> 2;
    const int c = blockIdx.x * 16 + 4 * cq;
    const bool on = c < C;
    const int cc4 = on ? c : 0;
    const float4 sc4 = *reinterpret_cast<const float4*>(scale + cc4), sh4 = *reinterpret_cast<const float4*>(shift + cc4);
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + cc4), is4 = *reinterpret_cast<const float4*>(invstd + cc4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
    const float* px = x + cc4;
    const float* pd = dy + cc4;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 kx[8], kd[8];                                       // KEEP: this lane's rows rl + BN_SL u of x and dy
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = rl + BN_SL * u < M ? rl + BN_SL * u : 0;
            kx[u] = *reinterpret_cast<const float4*>(px + rr * ldx);
            kd[u] = *reinterpret_cast<const float4*>(pd + rr * lddy);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (rl + BN_SL * u < M) {
                const float xe[4] = {kx[u].x, kx[u].y, kx[u].z, kx[u].w}, de[4] = {kd[u].x, kd[u].y, kd[u].z, kd[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dz = de[j];
                    if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                    a1[j] += dz;
                    a2[j] = fmaf(dz, (xe[j] - mu[j]) * is[j], a2[j]);
                }
            }
    } else
    for (long r = rl; r < M; r += BN_SL * 4) {
        float4 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = r + BN_SL * u < M ? r + BN_SL * u : r;
            xv[u] = *reinterpret_cast<const float4*>(px + rr * ldx);
            dv[u] = *reinterpret_cast<const float4*>(pd + rr * lddy);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r + BN_SL * u < M) {
                const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dz = de[j];
                    if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                    a1[j] += dz;
                    a2[j] = fmaf(dz, (xe[j] - mu[j]) * is[j], a2[j]);
                }
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a1[j];
    __syncthreads();
    const float t1 = small_colsum(red, grp, t);
    if (t < 16) bc[0][t] = t1;
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a2[j];
    __syncthreads();
    const float t2 = small_colsum(red, grp, t);
    if (t < 16) {
        bc[1][t] = t2;
        const int cc = blockIdx.x * 16 + t;
        if (cc < C) {
            if (dbeta) dbeta[cc] = accumulate ? dbeta[cc] + t1 : t1;
            if (dgamma) dgamma[cc] = accumulate ? dgamma[cc] + t2 : t2;
        }
    }
    __syncthreads();
    if (!dx || !on) return;
    const float invM = 1.0f / (float)M;
    float s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = bc[0][4 * cq + j]; s2[j] = bc[1][4 * cq + j]; }
    if (KEEP) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long r = rl + BN_SL * u;
            if (r >= M) continue;
            const float xe[4] = {kx[u].x, kx[u].y, kx[u].z, kx[u].w}, de[4] = {kd[u].x, kd[u].y, kd[u].z, kd[u].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dz = de[j];
                if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                float g = dz;
                if (training) {
                    const float xhat = (xe[j] - mu[j]) * is[j];
                    g = dz - s1[j] * invM - xhat * s2[j] * invM;
                }
                o[j] = sc[j] * g;
            }
            float4* dst = reinterpret_cast<float4*>(dx + r * lddx + c);
            if (dx_accumulate) { const float4 ov = *dst; o[0] += ov.x; o[1] += ov.y; o[2] += ov.z; o[3] += ov.w; }
            *dst = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    for (long r = rl; r < M; r += BN_SL * 4) {
        float4 xv[4], dv[4], ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long rr = r + BN_SL * u < M ? r + BN_SL * u : r;
            xv[u] = *reinterpret_cast<const float4*>(px + rr * ldx);
            dv[u] = *reinterpret_cast<const float4*>(pd + rr * lddy);
            if (dx_accumulate) ov[u] = *reinterpret_cast<const float4*>(dx + rr * lddx + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r + BN_SL * u < M) {
                const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, de[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dz = de[j];
                    if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                    float g = dz;
                    if (training) {
                        const float xhat = (xe[j] - mu[j]) * is[j];
                        g = dz - s1[j] * invM - xhat * s2[j] * invM;
                    }
                    o[j] = sc[j] * g;
                }
                if (dx_accumulate) { o[0] += ov[u].x; o[1] += ov[u].y; o[2] += ov[u].z; o[3] += ov[u].w; }
                *reinterpret_cast<float4*>(dx + (r + BN_SL * u) * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------ round 5: several workgroups per channel block
// At 2048 < M <= 8192 rows (one Visium grid: g's two BatchNorm2d(32) and the count MLP's BatchNorm1d; block 2 of a batch of
// 32 patches) the single-workgroup forms above walk their strip three times with three dependent rounds of loads each: 20 /
// 25 us per call, on C / 16 workgroups - TWO compute units for g's 32 channels.  Here BN_R workgroups share a channel block: a
// row lane holds its <= 8 rows in registers (ONE round of loads for all walks, as KEEP), every workgroup reduces its 256 row
// lanes with small_colsum, and the BN_R partial sums of a channel block are exchanged through `sync` memory behind a barrier of
// those BN_R workgroups (counters zeroed by a memset node in front of the launch; arrivals: release fence + agent-scope
// atomic, then ONE acquire before plain loads - MI355X_MICROARCH.md, inter-workgroup visibility).  The workgroups of a launch
// are always co-resident (C / 16 x BN_R of them, <= 256 on 256 CUs); the spin is bounded all the same.  Partials are added in
// workgroup order: deterministic; the summation ORDER differs from the single-workgroup forms (rows are dealt to 1024 lanes
// instead of 256), the arithmetic (two-pass variance) does not.
constexpr int BN_R = 4;
constexpr long BN_MULTI_M = 8L * BN_SL * BN_R;                  // 8192 rows
// Floats per workgroup partial = ONE 128-B line (16 used), 128-B aligned: a line holds ONE producer's bytes and is fetched
// by nobody before that producer has arrived.  Agent-scope loads are served by the reader's XCD L2 (MI355X_MICROARCH.md); with
// 64-B slots packed back to back a NEIGHBOURING group's consumer could pull a shared line into that L2 before its last producer
// wrote.  (Defensive: the failures that prompted it - dx off by 0.2 at C = 512 / 1024, only when the whole suite ran - turned
// out to be torch's single-threaded fp32 CPU BatchNorm backward, the tests' reference, which is that far from fp64 at those
// sizes; the tests now compare with a float64 reference.)
constexpr int BN_PS = 32;

// The exchanged partial sums are written and read with agent-scope (sc1: write-through) accesses: the BN_R workgroups of a
// channel block sit on different XCDs, whose L2s are not coherent - plain stores would leave a producer's bytes in ITS L2
// until the release writes them back, and two producers' halves of one line as two partially valid copies of it.
__device__ __forceinline__ void part_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float part_load(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void bn_group_barrier(unsigned* counter, unsigned target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the fence's own wait: MI355X_MICROARCH.md)
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// A PERSISTENT sync area (caller-owned, zeroed once: gnx_bn_sync_words) resets itself: a workgroup that has passed its last
// barrier counts itself out, and the last of the BN_R to leave - when nobody polls any more - zeroes the block's three words.
__device__ __forceinline__ void bn_group_leave(unsigned* words) {
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(words + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == BN_R - 1) {
            __hip_atomic_store(words, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(words + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(words + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// counters: [C/16 blocks][arrivals 1, arrivals 2, leavers] unsigned; part (128-B aligned): partial sums [2 phases][C/16][BN_R][BN_PS] floats
__global__ __launch_bounds__(1024) void bn_train_stats_multi_kernel(
    const float* __restrict__ x, long ld, long M, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
    float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ y, long ldy, int relu, unsigned* __restrict__ counters, float* __restrict__ part, int self_reset) {
    __shared__ float red[BN_SL][16];
    __shared__ float grp[4][16];
    __shared__ float bc[16];
    __shared__ float ss[2][16];
    const int t = threadIdx.x, cq = t & 3, rl = t >> 2;
    const int cb = blockIdx.x / BN_R, rg = blockIdx.x % BN_R, nb = gridDim.x / BN_R;
    const int c = cb * 16 + 4 * cq;
    const bool on = c < C;
    const float* px = x + (on ? c : 0);
    const long gl = (long)rg * BN_SL + rl;                     // this lane of the channel block's BN_R x 256 row lanes
    constexpr long STEP = (long)BN_SL * BN_R;
    float4 kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long rr = gl + STEP * u;
        kv[u] = *reinterpret_cast<const float4*>(px + (rr < M ? rr : 0) * ld);
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (gl + STEP * u < M) { a[0] += kv[u].x; a[1] += kv[u].y; a[2] += kv[u].z; a[3] += kv[u].w; }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a[j];
    __syncthreads();
    float tot = small_colsum(red, grp, t);
    float* const p1 = part + ((size_t)cb * BN_R) * BN_PS;
    float* const p2 = part + ((size_t)(nb + cb) * BN_R) * BN_PS;
    if (t < 16) part_store(p1 + rg * BN_PS + t, tot);
    bn_group_barrier(counters + 3 * cb, BN_R);
    if (t < 16) {
        float s_ = part_load(p1 + t);
#pragma unroll
        for (int k = 1; k < BN_R; ++k) s_ += part_load(p1 + k * BN_PS + t);
        bc[t] = s_ / (float)M;
    }
    __syncthreads();
    const float mu[4] = {bc[4 * cq], bc[4 * cq + 1], bc[4 * cq + 2], bc[4 * cq + 3]};
    float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (gl + STEP * u < M) {
            const float d0 = kv[u].x - mu[0], d1 = kv[u].y - mu[1], d2 = kv[u].z - mu[2], d3 = kv[u].w - mu[3];
            q[0] = fmaf(d0, d0, q[0]); q[1] = fmaf(d1, d1, q[1]); q[2] = fmaf(d2, d2, q[2]); q[3] = fmaf(d3, d3, q[3]);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = q[j];
    __syncthreads();
    tot = small_colsum(red, grp, t);
    if (t < 16) part_store(p2 + rg * BN_PS + t, tot);
    bn_group_barrier(counters + 3 * cb + 1, BN_R);
    if (self_reset) bn_group_leave(counters + 3 * cb);
    if (blockIdx.x == 0 && t == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (t < 16) {
        float m2 = part_load(p2 + t);
#pragma unroll
        for (int k = 1; k < BN_R; ++k) m2 += part_load(p2 + k * BN_PS + t);
        const int cc = cb * 16 + t;
        if (cc < C) {
            const float mean = bc[t];
            const float var = m2 / (float)M;
            const float invstd = 1.0f / sqrtf(var + eps);
            const float g = gamma ? gamma[cc] : 1.f, bt = beta ? beta[cc] : 0.f;
            ss[0][t] = g * invstd;
            ss[1][t] = bt - mean * g * invstd;
            if (rg == 0) {                                      // one of the BN_R writes the results
                scale[cc] = g * invstd;
                shift[cc] = bt - mean * g * invstd;
                save_mean[cc] = mean;
                save_invstd[cc] = invstd;
                if (running_mean) running_mean[cc] = (1.f - momentum) * running_mean[cc] + momentum * mean;
                if (running_var) {
                    const float unbiased = M > 1 ? m2 / (float)(M - 1) : var;
                    running_var[cc] = (1.f - momentum) * running_var[cc] + momentum * unbiased;
                }
            }
        }
    }
    if (!y) return;
    __syncthreads();
    if (!on) return;
    const float sc[4] = {ss[0][4 * cq], ss[0][4 * cq + 1], ss[0][4 * cq + 2], ss[0][4 * cq + 3]};
    const float sh[4] = {ss[1][4 * cq], ss[1][4 * cq + 1], ss[1][4 * cq + 2], ss[1][4 * cq + 3]};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long r = gl + STEP * u;
        if (r >= M) continue;
        float o[4] = {fmaf(kv[u].x, sc[0], sh[0]), fmaf(kv[u].y, sc[1], sh[1]), fmaf(kv[u].z, sc[2], sh[2]), fmaf(kv[u].w, sc[3], sh[3])};
        if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_multi_kernel(
    const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx, float* __restrict__ dx, long lddx, long M,
    int C, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, float* dgamma, float* dbeta, int relu, int training, int accumulate,
    int dx_accumulate, unsigned* __restrict__ counters, float* __restrict__ part, int self_reset) {
    __shared__ float red[BN_SL][16];
    __shared__ float grp[4][16];
    __shared__ float bc[2][16];
    const int t = threadIdx.x, cq = t & 3, rl = t >> 2;
    const int cb = blockIdx.x / BN_R, rg = blockIdx.x % BN_R, nb = gridDim.x / BN_R;
    const int c = cb * 16 + 4 * cq;
    const bool on = c < C;
    const int cc4 = on ? c : 0;
    const float4 sc4 = *reinterpret_cast<const float4*>(scale + cc4), sh4 = *reinterpret_cast<const float4*>(shift + cc4);
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + cc4), is4 = *reinterpret_cast<const float4*>(invstd + cc4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
    const float* px = x + cc4;
    const float* pd = dy + cc4;
    const long gl = (long)rg * BN_SL + rl;
    constexpr long STEP = (long)BN_SL * BN_R;
    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    float4 kx[8], kd[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long rr = gl + STEP * u < M ? gl + STEP * u : 0;
        kx[u] = *reinterpret_cast<const float4*>(px + rr * ldx);
        kd[u] = *reinterpret_cast<const float4*>(pd + rr * lddy);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (gl + STEP * u < M) {
            const float xe[4] = {kx[u].x, kx[u].y, kx[u].z, kx[u].w}, de[4] = {kd[u].x, kd[u].y, kd[u].z, kd[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dz = de[j];
                if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
                a1[j] += dz;
                a2[j] = fmaf(dz, (xe[j] - mu[j]) * is[j], a2[j]);
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a1[j];
    __syncthreads();
    const float t1 = small_colsum(red, grp, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[rl][4 * cq + j] = a2[j];
    __syncthreads();
    const float t2 = small_colsum(red, grp, t);
    float* const p1 = part + ((size_t)cb * BN_R) * BN_PS;
    float* const p2 = part + ((size_t)(nb + cb) * BN_R) * BN_PS;
    if (t < 16) {
        part_store(p1 + rg * BN_PS + t, t1);
        part_store(p2 + rg * BN_PS + t, t2);
    }
    bn_group_barrier(counters + 3 * cb, BN_R);
    if (self_reset) bn_group_leave(counters + 3 * cb);
    if (t < 16) {
        float s1_ = part_load(p1 + t), s2_ = part_load(p2 + t);
#pragma unroll
        for (int k = 1; k < BN_R; ++k) { s1_ += part_load(p1 + k * BN_PS + t); s2_ += part_load(p2 + k * BN_PS + t); }
        bc[0][t] = s1_;
        bc[1][t] = s2_;
        const int cc = cb * 16 + t;
        if (cc < C && rg == 0) {
            if (dbeta) dbeta[cc] = accumulate ? dbeta[cc] + s1_ : s1_;
            if (dgamma) dgamma[cc] = accumulate ? dgamma[cc] + s2_ : s2_;
        }
    }
    __syncthreads();
    if (!dx || !on) return;
    const float invM = 1.0f / (float)M;
    float s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[j] = bc[0][4 * cq + j]; s2[j] = bc[1][4 * cq + j]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long r = gl + STEP * u;
        if (r >= M) continue;
        const float xe[4] = {kx[u].x, kx[u].y, kx[u].z, kx[u].w}, de[4] = {kd[u].x, kd[u].y, kd[u].z, kd[u].w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dz = de[j];
            if (relu && fmaf(xe[j], sc[j], sh[j]) <= 0.f) dz = 0.f;
            float g = dz;
            if (training) {
                const float xhat = (xe[j] - mu[j]) * is[j];
                g = dz - s1[j] * invM - xhat * s2[j] * invM;
            }
            o[j] = sc[j] * g;
        }
        float4* dst = reinterpret_cast<float4*>(dx + r * lddx + c);
        if (dx_accumulate) { const float4 ov = *dst; o[0] += ov.x; o[1] += ov.y; o[2] += ov.z; o[3] += ov.w; }
        *dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// the sync area of the multi-workgroup forms inside a gnx_bn_workspace: counters (zeroed per launch), then partial sums
inline long bn_sync_floats(int C) {
    const long nb = (C + 15) / 16;
    return 3 * nb /* counters */ + 2 * nb * BN_R * BN_PS + 64 /* 128-B alignment of the partials */;
}

inline int elementwise_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

static bool bn_multi_ok(long M, int C) {
    return M > 8 * BN_SL && M <= BN_MULTI_M && C % 4 == 0 && ((C + 15) / 16) * BN_R <= 256;
}
// counters + partial sums of the multi-workgroup forms: behind the slab area of the workspace; the counters are zeroed here
static void bn_multi_sync(float* workspace, long M, int C, hipStream_t stream, unsigned* persistent, unsigned** counters,
                          float** part) {
    const long nb = (C + 15) / 16;
    float* base = workspace + 2L * slab_count(M) * C + 2L * C;
    *part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(base + 3 * nb) + 127) & ~(uintptr_t)127);
    if (persistent) {                                           // caller-owned, zeroed once, self-resetting: no memset node
        *counters = persistent;
        return;
    }
    *counters = reinterpret_cast<unsigned*>(base);
    (void)hipMemsetAsync(base, 0, 3 * nb * sizeof(unsigned), stream);
}

// floats of workspace needed by gnx_bn_train_stats / gnx_bn_relu_bwd for an [M][C] matrix
GNX_EXPORT long gnx_bn_workspace(long M, int C) {
    return 2L * slab_count(M) * C + 2L * C + bn_sync_floats(C);
}

// Training-mode statistics of x[M][C]: fills scale/shift (folded affine), save_mean/save_invstd and
// applies torch's running-stat update (momentum; unbiased running_var; num_batches_tracked += 1).
static int bn_train_stats_impl(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, long long* num_batches_tracked,
                               float momentum, float eps, float* scale, float* shift, float* save_mean,
                               float* save_invstd, float* workspace, unsigned* sync, hipStream_t stream) {
    if (!x || !scale || !shift || !save_mean || !save_invstd || !workspace || M <= 0 || C <= 0 || ld < C)
        return GNX_ERR_BAD_ARG;
    if (bn_multi_ok(M, C) && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        unsigned* counters;
        float* part;
        bn_multi_sync(workspace, M, C, stream, sync, &counters, &part);
        bn_train_stats_multi_kernel<<<gnx_cdiv(C, 16) * BN_R, 1024, 0, stream>>>(x, ld, M, C, gamma, beta, running_mean, running_var,
                                                                                num_batches_tracked, momentum, eps, scale, shift,
                                                                                save_mean, save_invstd, nullptr, 0, 0, counters, part,
                                                                                sync ? 1 : 0);
        return gnx_launch_status();
    }
    if (M <= BN_SMALL_M && C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        if (M <= 8 * BN_SL)
            bn_train_stats_small_kernel<true><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(x, ld, M, C, gamma, beta, running_mean,
                                                                                  running_var, num_batches_tracked, momentum,
                                                                                  eps, scale, shift, save_mean, save_invstd);
        else
            bn_train_stats_small_kernel<false><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(x, ld, M, C, gamma, beta, running_mean,
                                                                                   running_var, num_batches_tracked, momentum,
                                                                                   eps, scale, shift, save_mean, save_invstd);
        return gnx_launch_status();
    }
    const int nblk = slab_count(M);
    float* p_sum = workspace;
    float* p_m2 = workspace + (size_t)nblk * C;
    dim3 grid(nblk, gnx_cdiv(C, 64));
    if (C % 4 == 0 && ld % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0) {
        colsum_v4_kernel<0><<<grid, 64, 0, stream>>>(x, ld, M, C, nullptr, 0, p_sum);
        colsum_v4_kernel<1><<<grid, 64, 0, stream>>>(x, ld, M, C, p_sum, nblk, p_m2);
    } else {
        colsum_kernel<0><<<grid, 256, 0, stream>>>(x, ld, M, C, nullptr, 0, p_sum);
        colsum_kernel<1><<<grid, 256, 0, stream>>>(x, ld, M, C, p_sum, nblk, p_m2);
    }
    bn_finalize_train_kernel<<<gnx_cdiv(C, 64), 64, 0, stream>>>(p_sum, p_m2, nblk, M, C, gamma, beta,
                                                                 running_mean, running_var, num_batches_tracked,
                                                                 momentum, eps, scale, shift, save_mean,
                                                                 save_invstd);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_bn_train_stats(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches_tracked,
                                  float momentum, float eps, float* scale, float* shift, float* save_mean,
                                  float* save_invstd, float* workspace, hipStream_t stream) {
    return bn_train_stats_impl(x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift,
                               save_mean, save_invstd, workspace, nullptr, stream);
}
// The same with the layer's PERSISTENT sync words (see gnx_bn_train_stats_apply_sync): no memset node in front of the
// several-workgroups-per-channel-block form.  sync == NULL, or any other shape: exactly gnx_bn_train_stats.
GNX_EXPORT int gnx_bn_train_stats_sync(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, long long* num_batches_tracked,
                                       float momentum, float eps, float* scale, float* shift, float* save_mean,
                                       float* save_invstd, float* workspace, void* sync, hipStream_t stream) {
    return bn_train_stats_impl(x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift,
                               save_mean, save_invstd, workspace, reinterpret_cast<unsigned*>(sync), stream);
}

GNX_EXPORT int gnx_scale_shift_relu(const float* x, long ldx, float* y, long ldy, long M, int C, const float* scale,
                                    const float* shift, int relu, hipStream_t stream);

// gnx_bn_train_stats followed by gnx_scale_shift_relu (y = [relu](scale x + shift)) - in ONE launch where the matrix is small
// enough for the single-launch statistics form (M <= 4992 rows, 4 | C, 16-B aligned rows of x and y), otherwise the two calls.
static int bn_train_stats_apply_impl(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, long long* num_batches_tracked,
                                     float momentum, float eps, float* scale, float* shift, float* save_mean,
                                     float* save_invstd, float* y, long ldy, int relu, float* workspace, unsigned* sync,
                                     hipStream_t stream) {
    if (!x || !y || !scale || !shift || !save_mean || !save_invstd || !workspace || M <= 0 || C <= 0 || ld < C || ldy < C)
        return GNX_ERR_BAD_ARG;
    if (bn_multi_ok(M, C) && ld % 4 == 0 && ldy % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        unsigned* counters;
        float* part;
        bn_multi_sync(workspace, M, C, stream, sync, &counters, &part);
        bn_train_stats_multi_kernel<<<gnx_cdiv(C, 16) * BN_R, 1024, 0, stream>>>(x, ld, M, C, gamma, beta, running_mean, running_var,
                                                                                num_batches_tracked, momentum, eps, scale, shift,
                                                                                save_mean, save_invstd, y, ldy, relu, counters, part,
                                                                                sync ? 1 : 0);
        return gnx_launch_status();
    }
    if (M <= BN_SMALL_M && C % 4 == 0 && ld % 4 == 0 && ldy % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        if (M <= 8 * BN_SL)
            bn_train_stats_small_kernel<true><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(
                x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, save_mean,
                save_invstd, y, ldy, relu);
        else
            bn_train_stats_small_kernel<false><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(
                x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, save_mean,
                save_invstd, y, ldy, relu);
        return gnx_launch_status();
    }
    const int rc = gnx_bn_train_stats(x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                                      scale, shift, save_mean, save_invstd, workspace, stream);
    if (rc != GNX_OK) return rc;
    return gnx_scale_shift_relu(x, ld, y, ldy, M, C, scale, shift, relu, stream);
}
GNX_EXPORT int gnx_bn_train_stats_apply(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, long long* num_batches_tracked,
                                        float momentum, float eps, float* scale, float* shift, float* save_mean,
                                        float* save_invstd, float* y, long ldy, int relu, float* workspace,
                                        hipStream_t stream) {
    return bn_train_stats_apply_impl(x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                                     shift, save_mean, save_invstd, y, ldy, relu, workspace, nullptr, stream);
}
// 32-bit words of the persistent, self-resetting sync area of the `_sync` entry points (zero them once)
GNX_EXPORT long gnx_bn_sync_words(int C) { return 3L * ((C + 15) / 16); }
GNX_EXPORT int gnx_bn_train_stats_apply_sync(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var, long long* num_batches_tracked,
                                             float momentum, float eps, float* scale, float* shift, float* save_mean,
                                             float* save_invstd, float* y, long ldy, int relu, float* workspace, void* sync,
                                             hipStream_t stream) {
    return bn_train_stats_apply_impl(x, ld, M, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, scale,
                                     shift, save_mean, save_invstd, y, ldy, relu, workspace, reinterpret_cast<unsigned*>(sync), stream);
}

// Eval-mode fold: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale
GNX_EXPORT int gnx_bn_fold_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, float* scale, float* shift,
                                float* save_mean, float* save_invstd, hipStream_t stream) {
    if (C <= 0 || !running_mean || !running_var || !scale || !shift) return GNX_ERR_BAD_ARG;
    bn_fold_eval_kernel<<<gnx_cdiv(C, 256), 256, 0, stream>>>(C, gamma, beta, running_mean, running_var, eps,
                                                              scale, shift, save_mean, save_invstd);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_scale_shift_relu(const float* x, long ldx, float* y, long ldy, long M, int C,
                                    const float* scale, const float* shift, int relu, hipStream_t stream) {
    if (!x || !y || !scale || !shift || M < 0 || C <= 0 || ldx < C || ldy < C) return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    scale_shift_relu_kernel<<<elementwise_grid(M * C), 256, 0, stream>>>(x, ldx, y, ldy, M, C, scale, shift, relu);
    return gnx_launch_status();
}

// Backward of y = [relu](bn(x)).  dgamma/dbeta may be null; `training` selects batch-stat or running-stat form;
// `accumulate` adds into dgamma/dbeta, `dx_accumulate` adds into dx (DenseNet: gradients of a block buffer's columns
// arrive from every later layer of the block).
static int bn_relu_bwd_impl(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long M,
                            int C, const float* scale, const float* shift, const float* save_mean,
                            const float* save_invstd, float* dgamma, float* dbeta, int relu, int training,
                            int accumulate, int dx_accumulate, float* workspace, unsigned* sync, hipStream_t stream) {
    if (!dy || !x || !scale || !shift || !save_mean || !save_invstd || !workspace || M <= 0 || C <= 0)
        return GNX_ERR_BAD_ARG;
    const int nblk = slab_count(M);
    float* partial = workspace;
    float* sums = workspace + (size_t)2 * nblk * C;
    dim3 grid(nblk, gnx_cdiv(C, 64));
    {
        const bool v4all = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && (!dx || lddx % 4 == 0) &&
                           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) |
                             reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(scale) |
                             reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(save_mean) |
                             reinterpret_cast<uintptr_t>(save_invstd)) & 15) == 0;
        if (relu == 2 && (training || !v4all)) return GNX_ERR_UNSUPPORTED;     // activated-input form: eval statistics only
        if (v4all && relu != 2 && bn_multi_ok(M, C)) {
            unsigned* counters;
            float* part;
            bn_multi_sync(workspace, M, C, stream, sync, &counters, &part);
            bn_bwd_multi_kernel<<<gnx_cdiv(C, 16) * BN_R, 1024, 0, stream>>>(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift, save_mean,
                                                                           save_invstd, dgamma, dbeta, relu, training, accumulate,
                                                                           dx_accumulate, counters, part, sync ? 1 : 0);
            return gnx_launch_status();
        }
        if (v4all && relu != 2 && M <= BN_SMALL_M) {
            if (M <= 8 * BN_SL)
                bn_bwd_small_kernel<true><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift,
                                                                              save_mean, save_invstd, dgamma, dbeta, relu,
                                                                              training, accumulate, dx_accumulate);
            else
                bn_bwd_small_kernel<false><<<gnx_cdiv(C, 16), 1024, 0, stream>>>(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift,
                                                                               save_mean, save_invstd, dgamma, dbeta, relu,
                                                                               training, accumulate, dx_accumulate);
            return gnx_launch_status();
        }
        if (!training && v4all) {
            // one pass: dx and the dgamma/dbeta column sums together
            bn_bwd_eval_fused_kernel<<<grid, 256, 0, stream>>>(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift, save_mean,
                                                               save_invstd, relu, dx_accumulate, partial);
            bn_bwd_reduce_kernel<<<gnx_cdiv(C, 64), 256, 0, stream>>>(partial, nblk, C, sums, dgamma, dbeta, accumulate);
            return gnx_launch_status();
        }
    }
    if (C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(scale) |
          reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(save_mean) |
          reinterpret_cast<uintptr_t>(save_invstd)) & 15) == 0)
        bn_bwd_partial_v4_kernel<<<grid, 64, 0, stream>>>(dy, lddy, x, ldx, M, C, scale, shift, save_mean, save_invstd, relu,
                                                          partial);
    else
        bn_bwd_partial_kernel<<<grid, 256, 0, stream>>>(dy, lddy, x, ldx, M, C, scale, shift, save_mean, save_invstd,
                                                        relu, partial);
    bn_bwd_reduce_kernel<<<gnx_cdiv(C, 64), 256, 0, stream>>>(partial, nblk, C, sums, dgamma, dbeta, accumulate);
    if (dx) {
        const bool v4 = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) |
                          reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(scale) |
                          reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(save_mean) |
                          reinterpret_cast<uintptr_t>(save_invstd) | reinterpret_cast<uintptr_t>(sums)) & 15) == 0;
        if (v4)
            bn_bwd_dx_vec4_kernel<<<elementwise_grid(M * (C / 4)), 256, 0, stream>>>(
                dy, lddy, x, ldx, dx, lddx, M, C / 4, scale, shift, save_mean, save_invstd, sums, relu, training,
                dx_accumulate);
        else
            bn_bwd_dx_kernel<<<elementwise_grid(M * C), 256, 0, stream>>>(dy, lddy, x, ldx, dx, lddx, M, C, scale,
                                                                          shift, save_mean, save_invstd, sums, relu,
                                                                          training, dx_accumulate);
    }
    return gnx_launch_status();
}
GNX_EXPORT int gnx_bn_relu_bwd(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long M,
                               int C, const float* scale, const float* shift, const float* save_mean,
                               const float* save_invstd, float* dgamma, float* dbeta, int relu, int training,
                               int accumulate, int dx_accumulate, float* workspace, hipStream_t stream) {
    return bn_relu_bwd_impl(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift, save_mean, save_invstd, dgamma, dbeta, relu, training,
                            accumulate, dx_accumulate, workspace, nullptr, stream);
}
// The same with a PERSISTENT sync area for the several-workgroups-per-channel-block form (2049 ... 8192 rows): `sync` points at
// gnx_bn_sync_words(C) 32-bit words the caller zeroed ONCE and keeps for this BatchNorm layer (never shared by two launches
// that may run at the same time); the kernel leaves them zero again, so no memset node precedes the launch (4.7 us per call
// under graph replay).  sync == NULL, or any other shape: exactly gnx_bn_relu_bwd.
GNX_EXPORT int gnx_bn_relu_bwd_sync(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long M,
                                    int C, const float* scale, const float* shift, const float* save_mean,
                                    const float* save_invstd, float* dgamma, float* dbeta, int relu, int training,
                                    int accumulate, int dx_accumulate, float* workspace, void* sync, hipStream_t stream) {
    return bn_relu_bwd_impl(dy, lddy, x, ldx, dx, lddx, M, C, scale, shift, save_mean, save_invstd, dgamma, dbeta, relu, training,
                            accumulate, dx_accumulate, workspace, reinterpret_cast<unsigned*>(sync), stream);
}

// out[c] = sum_r x[r][c]  (bias gradient of a Linear layer); workspace: gnx_bn_workspace(M, C) floats
// BN (eval statistics) -> ReLU adjoint reading the gradient of the 2x2-average-POOLED map (a transition, densenet.py:50-54):
//   dX[m][c] = scale[c] * 0.25 dYp[pool(m)][c] * [scale x + shift > 0],  dbeta / dgamma from the same pass
// == gnx_avgpool2_bwd followed by gnx_bn_relu_bwd(relu = 1, training = 0) without the full-size intermediate.
// x / dX: [imgs*S*S][C]; dYp: [imgs*(S/2)^2][C].  4 | C, 16-B aligned operands; GNX_ERR_UNSUPPORTED otherwise.
GNX_EXPORT int gnx_bn_relu_bwd_pooled(const float* dYp, long lddy, const float* x, long ldx, float* dx, long lddx, long imgs,
                                      int S, int C, const float* scale, const float* shift, const float* save_mean,
                                      const float* save_invstd, float* dgamma, float* dbeta, int accumulate,
                                      float* workspace, hipStream_t stream) {
    if (!dYp || !x || !dx || !scale || !shift || !save_mean || !save_invstd || !workspace || imgs <= 0 || S < 2 || C <= 0)
        return GNX_ERR_BAD_ARG;
    const long M = imgs * S * S;
    const bool v4all = C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 &&
                       ((reinterpret_cast<uintptr_t>(dYp) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                         reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                         reinterpret_cast<uintptr_t>(save_mean) | reinterpret_cast<uintptr_t>(save_invstd)) & 15) == 0;
    if (!v4all) return GNX_ERR_UNSUPPORTED;
    const int nblk = slab_count(M);
    float* partial = workspace;
    float* sums = workspace + (size_t)2 * nblk * C;
    dim3 grid(nblk, gnx_cdiv(C, 64));
    bn_bwd_eval_fused_kernel<<<grid, 256, 0, stream>>>(dYp, lddy, x, ldx, dx, lddx, M, C, scale, shift, save_mean, save_invstd, 1,
                                                       0, partial, S);
    bn_bwd_reduce_kernel<<<gnx_cdiv(C, 64), 256, 0, stream>>>(partial, nblk, C, sums, dgamma, dbeta, accumulate);
    return gnx_launch_status();
}

// out[img, oy, ox][c] = mean over the 2x2 window of relu(scale[c] in[img, 2oy + dy, 2ox + dx][c] + shift[c]): the activated,
// pooled input of a transition's 1x1 conv (pool-first), 16-B accesses - the operand of the transition's weight gradient.
__global__ __launch_bounds__(256) void bnrelu_avgpool2_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                              long ldo, long Mout, int C4, int S,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift) {
    const long total = Mout * C4;
    const int So = S >> 1;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C4;
        const int c = 4 * (int)(idx - row * C4);
        const long img = row / ((long)So * So);
        const int rem = (int)(row - img * So * So);
        const int oy = rem / So, ox = rem - oy * So;
        const long src = ((img * S + 2 * oy) * S + 2 * ox) * ldi + c;
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(in + src + ((q >> 1) * (long)S + (q & 1)) * ldi);
            a.x += fmaxf(fmaf(v.x, sc.x, sh.x), 0.f); a.y += fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
            a.z += fmaxf(fmaf(v.z, sc.z, sh.z), 0.f); a.w += fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        }
        *reinterpret_cast<float4*>(out + row * ldo + c) = make_float4(0.25f * a.x, 0.25f * a.y, 0.25f * a.z, 0.25f * a.w);
    }
}
GNX_EXPORT int gnx_bnrelu_avgpool2(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S,
                                   const float* scale, const float* shift, hipStream_t stream) {
    if (!in || !out || !scale || !shift || imgs < 0 || C <= 0 || S < 2 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (C % 4 != 0 || S % 2 != 0 || ldi % 4 != 0 || ldo % 4 != 0 ||
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(scale) |
          reinterpret_cast<uintptr_t>(shift)) & 15) != 0)
        return GNX_ERR_UNSUPPORTED;
    const long Mout = imgs * (S / 2) * (S / 2);
    if (Mout == 0) return GNX_OK;
    bnrelu_avgpool2_kernel<<<elementwise_grid(Mout * (C / 4)), 256, 0, stream>>>(in, ldi, out, ldo, Mout, C / 4, S, scale, shift);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_colsum(const float* x, long ld, long M, int C, float* out, int accumulate, float* workspace,
                          hipStream_t stream) {
    if (!x || !out || !workspace || M <= 0 || C <= 0 || ld < C) return GNX_ERR_BAD_ARG;
    const int nblk = slab_count(M);
    dim3 grid(nblk, gnx_cdiv(C, 64));
    if (C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        colsum_v4_kernel<0><<<grid, 64, 0, stream>>>(x, ld, M, C, nullptr, 0, workspace);
    else
        colsum_kernel<0><<<grid, 256, 0, stream>>>(x, ld, M, C, nullptr, 0, workspace);
    slab_reduce_kernel<<<gnx_cdiv(C, 64), 64, 0, stream>>>(workspace, nblk, C, out, accumulate);
    return gnx_launch_status();
}
