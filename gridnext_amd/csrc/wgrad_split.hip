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
