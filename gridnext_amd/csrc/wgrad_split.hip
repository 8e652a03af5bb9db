// conv1's WEIGHT gradient on split bf16 operands (opt-in; the gradient-side companion of conv1x1_split.hip):
//     dW[n][k] (+)= sum_m dY[m][n] relu(scale[k] X[m][k] + shift[k])
// (torch.autograd through /root/reference/gridnext/densenet.py:35-37, driven by training.py:164-171 with f_opt) with fp32 dY and X
// in HBM, every operand written as hi + lo in bf16 and a product as three v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (a_lo b_hi + a_hi b_lo + a_hi b_hi).  The fp32-instruction kernel (wgrad1_t_kernel, 107-116 TFLOP/s) is bound by the matrix
// pipe; this one reads its operands once and is bound by HBM.
//
// The contraction runs over PIXELS: a workgroup owns a 128 x 128 block of (n, k) - a wave 64 x 64 - and a contiguous range of
// 64-pixel tiles (a slab).  A tile's dY and X pieces go global -> registers -> (activation) -> split -> four row-major
// [pixel][128] bf16 planes in the LDS, and the matrix operands - 8 consecutive pixels of one channel per lane - come out of them
// by TRANSPOSING reads (ds_read_b64_tr_b16), as in dense_bwd_f16.hip.  Slabs are summed in a fixed order by a second kernel:
// deterministic, no float atomics.
#include "fwd_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(8)));
typedef __fp16 fp16x8 __attribute__((__vector_size__(16)));

constexpr int WS_RS = 320;                 // bytes per row of a [pixel][128 ch] plane read by transposing reads (256 + 64: the 64-B
                                           // segments of 4 consecutive rows a 32-lane half touches fall on distinct banks)
constexpr int WS_PLANE = 64 * WS_RS;

// 8 contraction elements (pixels 8h .. 8h + 7 of a 16-pixel step) of one channel out of a row-major [pixel][channel] plane: two
// transposing reads of a 4-row x 16-column block each (16-bit elements: the bit patterns are bf16 here).  EXEC must be all ones.
__device__ __forceinline__ bf8 tr8(const char* lo, const char* hi) {
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)lo);
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)hi);
    const fp16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf8, v);
}

__global__ __launch_bounds__(256, 2) void wgrad1x1_split_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ X,
                                                                long ldx, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* __restrict__ ws, long M, int N,
                                                                int K, long tiles_per_slab, int n_kb, int n_nb, long n_slabs) {
    __shared__ __attribute__((aligned(16))) char smem[4 * WS_PLANE];     // dY hi | dY lo | act(X) hi | act(X) lo
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    // block order: the n_kb * n_nb workgroups of one slab (they read the same dY / X rows) get ids 8 apart - one XCD
    const int nblk = n_kb * n_nb;
    const int bx = blockIdx.x, rem8 = bx % (8 * nblk), kn = rem8 / 8;
    const int kb = kn % n_kb, nb = kn / n_kb;
    const long slab = (long)(bx / (8 * nblk)) * 8 + (rem8 & 7);
    if (slab >= n_slabs) return;                                         // (padding of the last group of 8; whole workgroup)
    const int chunk = t & 31, row0 = t >> 5;                             // loads: thread = 4 channels of one pixel, 8 pixels per pass
    const int ncol = nb * 128 + chunk * 4, kcol = kb * 128 + chunk * 4;
    const bool nok = ncol < N, kok = kcol < K;                           // 4 | N, 4 | K
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sc = (scale && kok) ? *reinterpret_cast<const f32x4*>(scale + kcol) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sh = (shift && kok) ? *reinterpret_cast<const f32x4*>(shift + kcol) : z;
    const bool act = scale != nullptr;
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    f32x4 p[8], q[8];
    // rows beyond M re-read row M - 1 and are zeroed at the split (both operands: 0 x anything finite)
    const float* const dyc = dY + (nok ? ncol : 0);
    const float* const xc = X + (kok ? kcol : 0);
    for (long tile = tile0; tile <= tile1; ++tile) {
        if (tile > tile0) {
            const long mprev = (tile - 1) * 64;
            lds_barrier();                                               // the previous tile's fragment reads are done
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = mprev + row0 + 8 * i < M;
                bf4 phi, plo, qhi, qlo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = (ok && nok) ? p[i][e] : 0.f;
                    float qv = q[i][e];
                    if (act) qv = fmaxf(fmaf(qv, sc[e], sh[e]), 0.f);
                    qv = (ok && kok) ? qv : 0.f;
                    phi[e] = (__bf16)pv;
                    plo[e] = (__bf16)(pv - (float)phi[e]);
                    qhi[e] = (__bf16)qv;
                    qlo[e] = (__bf16)(qv - (float)qhi[e]);
                }
                char* const d = smem + (row0 + 8 * i) * WS_RS + chunk * 8;
                *reinterpret_cast<bf4*>(d) = phi;
                *reinterpret_cast<bf4*>(d + WS_PLANE) = plo;
                *reinterpret_cast<bf4*>(d + 2 * WS_PLANE) = qhi;
                *reinterpret_cast<bf4*>(d + 3 * WS_PLANE) = qlo;
            }
            lds_barrier();
        }
        if (tile < tile1) {                                              // in flight while the previous tile multiplies
            const long m0 = tile * 64 + row0, rlast = M - 1 - m0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long row = m0 + (8 * i < rlast ? 8 * i : rlast);
                p[i] = *reinterpret_cast<const f32x4*>(dyc + row * lddy);
                q[i] = *reinterpret_cast<const f32x4*>(xc + row * ldx);
            }
        }
        if (tile > tile0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf8 a_hi[2], a_lo[2], b_hi[2], b_lo[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const char* pa = smem + (16 * ks + trow) * WS_RS + (64 * wm + 32 * j + tcol) * 2;
                    const char* pb = smem + 2 * WS_PLANE + (16 * ks + trow) * WS_RS + (64 * wn + 32 * j + tcol) * 2;
                    a_hi[j] = tr8(pa, pa + 4 * WS_RS);
                    a_lo[j] = tr8(pa + WS_PLANE, pa + WS_PLANE + 4 * WS_RS);
                    b_hi[j] = tr8(pb, pb + 4 * WS_RS);
                    b_lo[j] = tr8(pb + WS_PLANE, pb + WS_PLANE + 4 * WS_RS);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[i], b_hi[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_lo[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[i], b_hi[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    float* const out = ws + slab * (long)N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = kb * 128 + 64 * wn + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb * 128 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < N && k < K) out[(long)n * K + k] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void wgrad_split_reduce_kernel(const float* __restrict__ ws, long nslab, long n, float* __restrict__ out,
                                                                int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (long k = 0; k < nslab; ++k) s += ws[k * n + i];                 // fixed order
    out[i] = accumulate ? out[i] + s : s;
}

// ------------------------------------------------------------------------------------------------ conv2's weight gradient
// ws[slab][tap][n][k] = sum over the slab's pixels p of dY[p - (dy, dx)][n] A[p][k], tap = 3 (dy + 1) + (dx + 1), where the output
// pixel p - (dy, dx) lies in p's map (torch.autograd through densenet.py:41; A = the activated bottleneck, dY = the layer's 32
// gradient columns).  64-pixel tiles of A; the dY strip of the tile [P0 - S - 1, P0 + 64 + S + 1) is staged once and read at nine
// row offsets; rows a tap may not read are redirected - by the lane that supplies that row's address - to a row of zeros.  Wave w
// owns channels k = 32 w .. 32 w + 31 for all nine taps.  One workgroup per CU: nine accumulators and two operand planes each
// need the register file of a whole SIMD.
template <int S>
__global__ __launch_bounds__(256, 1) void wgrad3x3_split_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ A,
                                                                long lda, float* __restrict__ ws, long M, long tiles_per_slab) {
    constexpr int NROWS = 66 + 2 * S;
    constexpr int SPLANE = (NROWS + 1) * 64;                             // a strip plane: [row][32 ch] bf16 + the row of zeros
    constexpr int NI = (NROWS * 8 + 255) / 256;                          // 16-B pieces of the strip per thread
    __shared__ __attribute__((aligned(16))) char smem[2 * WS_PLANE + 2 * SPLANE];
    char* const strip = smem + 2 * WS_PLANE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long slab = blockIdx.x;
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    const int chunk = t & 31, row0 = t >> 5;
    const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    if (t < 32) reinterpret_cast<unsigned*>(strip + (t < 16 ? 0 : SPLANE) + NROWS * 64)[t & 15] = 0u;
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
    constexpr int S2 = S * S;
    f32x4 av[8], sv[NI];
    for (long tile = tile0; tile <= tile1; ++tile) {
        const long Pp = (tile - 1) * 64;                                 // the tile in the staging registers
        if (tile > tile0) {
            lds_barrier();                                               // the previous tile's reads are done
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = Pp + row0 + 8 * i < M;
                bf4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ok ? av[i][e] : 0.f;
                    hi[e] = (__bf16)v;
                    lo[e] = (__bf16)(v - (float)hi[e]);
                }
                char* const d = smem + (row0 + 8 * i) * WS_RS + chunk * 8;
                *reinterpret_cast<bf4*>(d) = hi;
                *reinterpret_cast<bf4*>(d + WS_PLANE) = lo;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int item = t + 256 * i;                            // (strip row, 16-B piece of its 32 channels)
                const long u = Pp - S - 1 + (item >> 3);
                if ((item >> 3) < NROWS) {
                    const bool ok = u >= 0 && u < M;
                    bf4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ok ? sv[i][e] : 0.f;
                        hi[e] = (__bf16)v;
                        lo[e] = (__bf16)(v - (float)hi[e]);
                    }
                    char* const d = strip + (item >> 3) * 64 + (item & 7) * 8;
                    *reinterpret_cast<bf4*>(d) = hi;
                    *reinterpret_cast<bf4*>(d + SPLANE) = lo;
                }
            }
            lds_barrier();
        }
        if (tile < tile1) {                                              // in flight while the previous tile multiplies
            const long P0 = tile * 64;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                long row = P0 + row0 + 8 * i;
                row = row < M ? row : M - 1;
                av[i] = *reinterpret_cast<const f32x4*>(A + row * lda + chunk * 4);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int item = t + 256 * i;
                long u = P0 - S - 1 + (item >> 3);
                u = u < 0 ? 0 : (u < M ? u : M - 1);
                sv[i] = *reinterpret_cast<const f32x4*>(dY + u * lddy + (item & 7) * 4);
            }
        }
        if (tile > tile0) {
            const int rem0 = (int)(Pp & (S2 - 1));                       // position of the tile's first pixel inside its map
#pragma unroll 1
            for (int ks = 0; ks < 4; ++ks) {
                const char* pb = smem + (16 * ks + trow) * WS_RS + (32 * wave + tcol) * 2;
                const bf8 b_hi = tr8(pb, pb + 4 * WS_RS);
                const bf8 b_lo = tr8(pb + WS_PLANE, pb + WS_PLANE + 4 * WS_RS);
                // which taps may read the rows this lane supplies (pixel p of A; its partner is the output pixel p - (dy, dx))
                unsigned vlo = 0, vhi = 0;
                {
                    const int rem_lo = (rem0 + 16 * ks + trow) & (S2 - 1), rem_hi = (rem0 + 16 * ks + trow + 4) & (S2 - 1);
                    const int ylo = rem_lo / S, xlo = rem_lo & (S - 1), yhi = rem_hi / S, xhi = rem_hi & (S - 1);
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                        if (ylo - dy >= 0 && ylo - dy < S && xlo - dx >= 0 && xlo - dx < S) vlo |= 1u << tap;
                        if (yhi - dy >= 0 && yhi - dy < S && xhi - dx >= 0 && xhi - dx < S) vhi |= 1u << tap;
                    }
                }
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {                         // three taps at a time: their six fragments, then nine products
                    bf8 a_hi[3], a_lo[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int tap = 3 * g3 + j;
                        const int off = (tap / 3 - 1) * S + (tap % 3 - 1);
                        const char* lo = strip + (((vlo >> tap) & 1) ? (16 * ks + trow - off + S + 1) : NROWS) * 64 + tcol * 2;
                        const char* hi = strip + (((vhi >> tap) & 1) ? (16 * ks + trow + 4 - off + S + 1) : NROWS) * 64 + tcol * 2;
                        a_hi[j] = tr8(lo, hi);
                        a_lo[j] = tr8(lo + SPLANE, hi + SPLANE);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int tap = 3 * g3 + j;
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[j], b_hi, acc[tap], 0, 0, 0);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[j], b_lo, acc[tap], 0, 0, 0);
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[j], b_hi, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
    }
    float* const out = ws + slab * (9L * 32 * 128);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(tap * 32 + n) * 128 + 32 * wave + (lane & 31)] = acc[tap][r];
        }
}

// dW [n][k][tap] (conv2.weight's layout) (+)= sum over slabs of ws[slab][tap][n][k], fixed order
__global__ __launch_bounds__(256) void wgrad3x3_split_reduce_kernel(const float* __restrict__ ws, long nslab, float* __restrict__ out,
                                                                   int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;                        // index into [n][k][tap]
    if (i >= 32 * 128 * 9) return;
    const int tap = i % 9, k = i / 9 % 128, n = i / (9 * 128);
    const long src = ((long)tap * 32 + n) * 128 + k;
    float s = 0.f;
    for (long j = 0; j < nslab; ++j) s += ws[j * (9L * 32 * 128) + src];
    out[i] = accumulate ? out[i] + s : s;
}

struct W3Plan {
    long slabs, per;
};
W3Plan w3_plan(long M) {
    W3Plan p;
    const long tiles = (M + 63) / 64;
    long want = 256 < tiles ? 256 : tiles;                               // one workgroup per CU
    p.per = (tiles + want - 1) / want;
    p.slabs = (tiles + p.per - 1) / p.per;
    return p;
}

struct WsPlan {
    long tiles, slabs, per;
    int n_kb, n_nb;
};
WsPlan ws_plan(long M, int N, int K) {
    WsPlan p;
    p.n_kb = (K + 127) / 128;
    p.n_nb = (N + 127) / 128;
    p.tiles = (M + 63) / 64;
    const long blocks = (long)p.n_kb * p.n_nb;
    long want = 512 / blocks / 8 * 8;                                    // two workgroups per CU, whole groups of 8 slabs
    if (want < 8) want = 8;
    if (want > p.tiles) want = p.tiles;
    p.per = (p.tiles + want - 1) / want;
    p.slabs = (p.tiles + p.per - 1) / p.per;
    return p;
}

}  // namespace

GNX_EXPORT long gnx_wgrad1x1_split_workspace(long M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    return ws_plan(M, N, K).slabs * (long)N * K;
}
// dW [N][K] (+)= dY^T act(X): gnx_wgrad_bnrelu (taps = 1, pool = 0) on split bf16 operands.  scale = shift = NULL: no activation.
// fp32 operands, 16-B aligned, 4 | lddy, ldx, N, K; else GNX_ERR_UNSUPPORTED.  workspace: gnx_wgrad1x1_split_workspace floats.
GNX_EXPORT int gnx_wgrad1x1_split(const float* dY, long lddy, const float* X, long ldx, const float* scale, const float* shift,
                                  float* dW, float* workspace, long M, int N, int K, int accumulate, hipStream_t stream) {
    if (!dY || !X || !dW || !workspace || M < 1 || N < 1 || K < 1 || lddy < N || ldx < K || (scale == nullptr) != (shift == nullptr))
        return GNX_ERR_BAD_ARG;
    if ((N & 3) || (K & 3) || (lddy & 3) || (ldx & 3) || (reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(X) & 15) ||
        (scale && ((reinterpret_cast<uintptr_t>(scale) & 15) || (reinterpret_cast<uintptr_t>(shift) & 15))))
        return GNX_ERR_UNSUPPORTED;
    const WsPlan p = ws_plan(M, N, K);
    const long groups = (p.slabs + 7) / 8;
    const long grid = groups * 8 * p.n_kb * p.n_nb;
    if (grid >= (1L << 31)) return GNX_ERR_UNSUPPORTED;
    wgrad1x1_split_kernel<<<(int)grid, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, N, K, p.per, p.n_kb, p.n_nb,
                                                        p.slabs);
    const long n = (long)N * K;
    wgrad_split_reduce_kernel<<<(int)((n + 255) / 256), 256, 0, stream>>>(workspace, p.slabs, n, dW, accumulate);
    return gnx_launch_status();
}

GNX_EXPORT long gnx_wgrad3x3_split_workspace(long M) { return M < 1 ? 0 : w3_plan(M).slabs * (9L * 32 * 128); }
// dW [32][128][3][3] (+)= the weight gradient of conv2 (gnx_wgrad_bnrelu with taps = 9, N = 32, K = 128, no prologue: A is the
// ACTIVATED bottleneck) on split bf16 operands.  Maps of S x S pixels, S in {4, 8, 16, 32, 64}, S * S | M; fp32 operands, 16-B
// aligned, 4 | lddy, lda; else GNX_ERR_UNSUPPORTED.  workspace: gnx_wgrad3x3_split_workspace(M) floats.
GNX_EXPORT int gnx_wgrad3x3_split(const float* dY, long lddy, const float* A, long lda, float* dW, float* workspace, long M, int S,
                                  int accumulate, hipStream_t stream) {
    if (!dY || !A || !dW || !workspace || M < 1 || lddy < 32 || lda < 128 || S < 1) return GNX_ERR_BAD_ARG;
    if ((lddy & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(A) & 15) || M % ((long)S * S))
        return GNX_ERR_UNSUPPORTED;
    const W3Plan p = w3_plan(M);
    const int grid = (int)p.slabs;
    switch (S) {
        case 4: wgrad3x3_split_kernel<4><<<grid, 256, 0, stream>>>(dY, lddy, A, lda, workspace, M, p.per); break;
        case 8: wgrad3x3_split_kernel<8><<<grid, 256, 0, stream>>>(dY, lddy, A, lda, workspace, M, p.per); break;
        case 16: wgrad3x3_split_kernel<16><<<grid, 256, 0, stream>>>(dY, lddy, A, lda, workspace, M, p.per); break;
        case 32: wgrad3x3_split_kernel<32><<<grid, 256, 0, stream>>>(dY, lddy, A, lda, workspace, M, p.per); break;
        case 64: wgrad3x3_split_kernel<64><<<grid, 256, 0, stream>>>(dY, lddy, A, lda, workspace, M, p.per); break;
        default: return GNX_ERR_UNSUPPORTED;
    }
    wgrad3x3_split_reduce_kernel<<<(32 * 128 * 9 + 255) / 256, 256, 0, stream>>>(workspace, p.slabs, dW, accumulate);
    return gnx_launch_status();
}
