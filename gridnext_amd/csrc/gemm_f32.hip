// General fp32 MFMA GEMM for the Linear layers of the count-MLP spot head (forward and both gradients).
//
//   C[M][N] (ldc) = opA(A)[M][K] * opB(B)[K][N] (+ bias[N]) (+ C when `accumulate`)
//   a_kmajor = 0: A[m*lda + k]   (row-major activations / dY)      1: A[k*lda + m]
//              (1 is how the reference's count grids arrive: (B, genes, H, W) with the spot index
//               contiguous - gridnet_models.py:167-169 permutes+copies them; here they are read in place)
//   b_kmajor = 0: B[n*ldb + k]   (nn.Linear weight [out][in])      1: B[k*ldb + n]
// Replaces F.linear forward/backward of the nn.Sequential defined in
// /root/reference/notebooks/Tutorial_visium_count.ipynb cell 12 / Tutorial_multimodal.ipynb cell 23.
//
// Exact fp32: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; bitwise an fmaf chain).  64x64 output tile
// per 256-thread workgroup (4 waves, one 32x32 MFMA tile each), K step 64 (r1: 16 - two workgroup barriers per 8 MFMAs
// of a wave, 43 TFLOP/s on the 2000 -> 500 layer; now per 32), LDS image [k][m] so every
// fragment read is a conflict-free ds_read_b32 whatever the source layout; next K-tile is prefetched
// into registers while the current one is multiplied.  Bound: fp32 matrix peak 157.3 TFLOP/s
// (2*M*N*K FLOP) for the 2000->500 layer, HBM for the count stream (K-major A is read once).
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 64, LD = 68, NSUB = BK / 16;      // a K tile = NSUB sub-tiles of 16

struct TileRegs { float v[NSUB][4]; };

// source K-contiguous ([row][k]): thread -> (row = t>>2, k quad = t&3)
// source row-contiguous ([k][row]): thread -> (k = t>>4, row quad = t&15)
template <bool KMAJOR>
__device__ __forceinline__ TileRegs load_tile(const float* __restrict__ src, long ld, long row0, long nrows,
                                              long k0, long K, bool vec_ok) {
    TileRegs r;
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        if (!KMAJOR) {
            const long row = row0 + (t >> 2), k = k0 + 16 * u + 4 * (t & 3);
            if (row < nrows && vec_ok && k + 3 < K) {
                const float4 q = *reinterpret_cast<const float4*>(src + row * ld + k);
                r.v[u][0] = q.x; r.v[u][1] = q.y; r.v[u][2] = q.z; r.v[u][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.v[u][j] = (row < nrows && k + j < K) ? src[row * ld + k + j] : 0.f;
            }
        } else {
            const long k = k0 + 16 * u + (t >> 4), row = row0 + 4 * (t & 15);
            if (k < K && vec_ok && row + 3 < nrows) {
                const float4 q = *reinterpret_cast<const float4*>(src + k * ld + row);
                r.v[u][0] = q.x; r.v[u][1] = q.y; r.v[u][2] = q.z; r.v[u][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.v[u][j] = (k < K && row + j < nrows) ? src[k * ld + row + j] : 0.f;
            }
        }
    }
    return r;
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const TileRegs& r) {
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        if (!KMAJOR) {
            const int row = t >> 2, kq = t & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) lds[(16 * u + 4 * kq + j) * LD + row] = r.v[u][j];
        } else {
            const int k = 16 * u + (t >> 4), rq = t & 15;
            *reinterpret_cast<float4*>(lds + k * LD + 4 * rq) = make_float4(r.v[u][0], r.v[u][1], r.v[u][2], r.v[u][3]);
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long lda,
                                                       const float* __restrict__ B, long ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       long ldc, long M, long N, long K, int a_vec, int b_vec,
                                                       int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
    const long m0 = (long)blockIdx.x * BM, n0 = (long)blockIdx.y * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long nkt = (K + BK - 1) / BK;
    TileRegs ra = load_tile<A_KMAJOR>(A, lda, m0, M, 0, K, a_vec);
    TileRegs rb = load_tile<B_KMAJOR>(B, ldb, n0, N, 0, K, b_vec);
    for (long kt = 0; kt < nkt; ++kt) {
        store_tile<A_KMAJOR>(As, ra);
        store_tile<B_KMAJOR>(Bs, rb);
        __syncthreads();
        if (kt + 1 < nkt) {
            ra = load_tile<A_KMAJOR>(A, lda, m0, M, (kt + 1) * BK, K, a_vec);
            rb = load_tile<B_KMAJOR>(B, ldb, n0, N, (kt + 1) * BK, K, b_vec);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[(2 * kk + h) * LD + 32 * wm + i];
            const float b = Bs[(2 * kk + h) * LD + 32 * wn + i];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const long col = n0 + 32 * wn + i;
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) {
                float v = acc[r] + bv;
                if (accumulate) v += C[row * ldc + col];
                C[row * ldc + col] = v;
            }
        }
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

GNX_EXPORT int gnx_gemm_f32(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
                            const float* bias, float* C, long ldc, long M, long N, long K, int accumulate,
                            hipStream_t stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0 || ldc < N) return GNX_ERR_BAD_ARG;
    if (M == 0 || N == 0) return GNX_OK;
    // float4 loads need 16-B aligned bases and leading dimensions that keep every row 16-B aligned
    const int a_vec = aligned16(A) && (lda % 4 == 0);
    const int b_vec = aligned16(B) && (ldb % 4 == 0);
    dim3 grid(gnx_cdiv(M, BM), gnx_cdiv(N, BN));
    if (grid.y > 65535) return GNX_ERR_UNSUPPORTED;
#define GNX_LAUNCH(AK, BKM) \
    gemm_f32_kernel<AK, BKM><<<grid, 256, 0, stream>>>(A, lda, B, ldb, bias, C, ldc, M, N, K, a_vec, b_vec, accumulate)
    if (!a_kmajor && !b_kmajor) GNX_LAUNCH(false, false);
    else if (a_kmajor && !b_kmajor) GNX_LAUNCH(true, false);
    else if (!a_kmajor && b_kmajor) GNX_LAUNCH(false, true);
    else GNX_LAUNCH(true, true);
#undef GNX_LAUNCH
    return gnx_launch_status();
}
