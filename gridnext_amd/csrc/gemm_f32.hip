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
#include <stdlib.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 64, LD = 68, NSUB = BK / 16;      // a K tile = NSUB sub-tiles of 16

struct TileRegs { float v[NSUB][4]; };

constexpr int GEMM_OOB = 0x7ffffff0;      // a byte offset past the end of any operand the entry point admits (< 2^31 bytes)
typedef unsigned gemm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gld4(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const gemm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
// the whole operand as one buffer: `major` rows of `ld` floats (K-major: major = K; row-major: major = rows)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float* p, long major, long ld, long minor) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(((major - 1) * ld + minor) * 4), 0x00020000);
}


// source K-contiguous ([row][k]): thread -> (row = t>>2, k quad = t&3)
// source row-contiguous ([k][row]): thread -> (k = t>>4, row quad = t&15)
// VEC: 16-B buffer loads, branch-free (rows past the operand's end read as 0; the overhang inside a source row gets an
// offset past the end) - the operand's rows are 16-B aligned and whole quads (K % 4 == 0 resp. nrows % 4 == 0).  Otherwise
// element by element behind bounds checks (any alignment, any shape; slow: every checked load is its own branch).
template <bool KMAJOR, bool VEC>
__device__ __forceinline__ TileRegs load_tile(const float* __restrict__ src, __amdgpu_buffer_rsrc_t rs, long ld, long row0,
                                              long nrows, long k0, long K) {
    TileRegs r;
    const int t = threadIdx.x;
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        if (!KMAJOR) {
            const long row = row0 + (t >> 2), k = k0 + 16 * u + 4 * (t & 3);
            if constexpr (VEC) {
                const float4 q = gld4(rs, k < K ? (int)((row * ld + k) * 4) : GEMM_OOB);
                r.v[u][0] = q.x; r.v[u][1] = q.y; r.v[u][2] = q.z; r.v[u][3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r.v[u][j] = (row < nrows && k + j < K) ? src[row * ld + k + j] : 0.f;
            }
        } else {
            const long k = k0 + 16 * u + (t >> 4), row = row0 + 4 * (t & 15);
            if constexpr (VEC) {
                const float4 q = gld4(rs, row < nrows ? (int)((k * ld + row) * 4) : GEMM_OOB);
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

template <bool A_KMAJOR, bool B_KMAJOR, bool A_VEC, bool B_VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long lda,
                                                       const float* __restrict__ B, long ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       long ldc, long M, long N, long K, int accumulate,
                                                       float* __restrict__ slabs) {
    // gridDim.z > 1 (tall, narrow products: few 64 x 64 tiles, long K - the composed 2000 -> 100 layer of the count MLP over a
    // 4 992-spot grid is 156 tiles for 256 CUs): workgroup z takes the z-th share of the K tiles and writes its partial tile
    // into slabs [z][M][N], summed in index order (+ bias) by gemm_split_reduce_kernel: deterministic.
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
    const long m0 = (long)blockIdx.x * BM, n0 = (long)blockIdx.y * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long nkt_all = (K + BK - 1) / BK;
    const long kt_first = nkt_all * blockIdx.z / gridDim.z, nkt = nkt_all * (blockIdx.z + 1) / gridDim.z;
    const __amdgpu_buffer_rsrc_t rA = A_VEC ? (A_KMAJOR ? operand_rsrc(A, K, lda, M) : operand_rsrc(A, M, lda, K)) : operand_rsrc(A, 1, 0, 4);
    const __amdgpu_buffer_rsrc_t rB = B_VEC ? (B_KMAJOR ? operand_rsrc(B, K, ldb, N) : operand_rsrc(B, N, ldb, K)) : operand_rsrc(B, 1, 0, 4);
    TileRegs ra = load_tile<A_KMAJOR, A_VEC>(A, rA, lda, m0, M, kt_first * BK, K);
    TileRegs rb = load_tile<B_KMAJOR, B_VEC>(B, rB, ldb, n0, N, kt_first * BK, K);
    for (long kt = kt_first; kt < nkt; ++kt) {
        store_tile<A_KMAJOR>(As, ra);
        store_tile<B_KMAJOR>(Bs, rb);
        __syncthreads();
        if ((A_VEC && B_VEC) || kt + 1 < nkt) {                 // vector loads past the end just return zeros
            ra = load_tile<A_KMAJOR, A_VEC>(A, rA, lda, m0, M, (kt + 1) * BK, K);
            rb = load_tile<B_KMAJOR, B_VEC>(B, rB, ldb, n0, N, (kt + 1) * BK, K);
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
        if (gridDim.z > 1) {
            float* const slab = slabs + (size_t)blockIdx.z * M * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) slab[row * N + col] = acc[r];
            }
            return;
        }
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


// ---- round 2: large-M form (the 2000 -> 500 layer over a whole grid: M = 4992 B spots).  The kernel above reads its
// fragments with two ds_read_b32 per MFMA and reached 56 TFLOP/s there.  Here the LDS images are [row][k] with k
// contiguous (pitch 36 = 4 * 9 floats), so a fragment read is one ds_read_b128 feeding four MFMAs (k order inside a group of
// 8 is (0,4),(1,5),(2,6),(3,7) for the lane halves - the same on both operands, so the contraction is unchanged; sums
// differ from the kernel above at rounding level only).  A workgroup owns 128 m x 64 n, a wave 64 m x 32 n (two
// accumulators: 3 fragment reads per 8 MFMAs).  A K-major operand ([k][row]: the count grids as they arrive) is transposed
// in registers (4 x 4 blocks) on its way into the LDS; a k-contiguous operand goes in as it is.  Next tile prefetched into
// registers.  Needs 16-B aligned rows on both operands (the entry point falls back to the kernel above otherwise).
constexpr int TM = 128, TN = 64, TK = 32, TP = TK + 4;

template <int ROWS, bool KMAJOR>
struct TStage {
    static_assert(ROWS == 64 || ROWS % 128 == 0, "");
    static constexpr int NV = ROWS * TK / 4 / 256;          // float4 per thread
    float4 v[NV];
    // KMAJOR ([k][row] source), ROWS = 128 NB: thread -> k quad kq = t & 7, row quads (t >> 3) + 32 u (u < NB): one 4 x 4
    //         block each, v[4 u + e] = src[k = 4 kq + e][4 rq ..]; ROWS = 64: 16 row quads - two threads share a block
    // else  : thread -> k quad kq = t & 7, rows (t >> 3) + 32 u; v[u] = src[row][4 kq ..]
    // Loads are BRANCH-FREE: a raw buffer load returns 0 past the end of the operand (K-major: k >= K; row-major: row >=
    // nrows); for the other overhang (inside a row of the source) the OFFSET is replaced by one past the end.  (`cond ? *p : 0` compiled to an
    // exec-masked branch per load with `s_waitcnt vmcnt(0)` behind each - no prefetch at all, 56 TFLOP/s whatever the tile.)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0, int K) {
        const int t = threadIdx.x;
        if constexpr (KMAJOR) {
            if constexpr (ROWS >= 128) {
                const int kq = t & 7;
#pragma unroll
                for (int u = 0; u < ROWS / 128; ++u) {
                    const int row = row0 + 4 * ((t >> 3) + 32 * u);
                    const bool ok = row < nrows;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[(4 * u + e) % NV] = gld4(rs, ok ? ((k0 + 4 * kq + e) * ld + row) * 4 : GEMM_OOB);
                }
            } else {                                        // 16 row quads x 8 k quads = 128 blocks: thread takes half a block
                const int blk = t >> 1, half = t & 1;       // k rows 4 kq + 2 half + {0, 1}
                const int kq2 = blk & 7;
                const int row = row0 + 4 * (blk >> 3);
                const bool ok = row < nrows;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    v[e % NV] = gld4(rs, ok ? ((k0 + 4 * kq2 + 2 * half + e) * ld + row) * 4 : GEMM_OOB);
            }
        } else {
            const int kq = t & 7;
            const int k = k0 + 4 * kq;
            const bool ok = k < K;
#pragma unroll
            for (int u = 0; u < NV; ++u)
                v[u] = gld4(rs, ok ? ((row0 + (t >> 3) + 32 * u) * ld + k) * 4 : GEMM_OOB);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ lds) const {
        const int t = threadIdx.x;
        if constexpr (KMAJOR) {
            if constexpr (ROWS >= 128) {
                const int kq = t & 7;
#pragma unroll
                for (int u = 0; u < ROWS / 128; ++u) {
                    const float4 &a = v[(4 * u) % NV], &b = v[(4 * u + 1) % NV], &c = v[(4 * u + 2) % NV], &d4 = v[(4 * u + 3) % NV];
                    float* d = lds + (4 * ((t >> 3) + 32 * u)) * TP + 4 * kq;
                    *reinterpret_cast<float4*>(d) = make_float4(a.x, b.x, c.x, d4.x);
                    *reinterpret_cast<float4*>(d + TP) = make_float4(a.y, b.y, c.y, d4.y);
                    *reinterpret_cast<float4*>(d + 2 * TP) = make_float4(a.z, b.z, c.z, d4.z);
                    *reinterpret_cast<float4*>(d + 3 * TP) = make_float4(a.w, b.w, c.w, d4.w);
                }
            } else {
                const int blk = t >> 1, half = t & 1, kq2 = blk & 7, rq = blk >> 3;
                float* d = lds + (4 * rq) * TP + 4 * kq2 + 2 * half;
                *reinterpret_cast<float2*>(d) = make_float2(v[0].x, v[1 % NV].x);
                *reinterpret_cast<float2*>(d + TP) = make_float2(v[0].y, v[1 % NV].y);
                *reinterpret_cast<float2*>(d + 2 * TP) = make_float2(v[0].z, v[1 % NV].z);
                *reinterpret_cast<float2*>(d + 3 * TP) = make_float2(v[0].w, v[1 % NV].w);
            }
        } else {
            const int kq = t & 7;
#pragma unroll
            for (int u = 0; u < NV; ++u)
                *reinterpret_cast<float4*>(lds + ((t >> 3) + 32 * u) * TP + 4 * kq) = v[u];
        }
    }
};

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f32_big_kernel(const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb,
                                                              const float* __restrict__ bias, float* __restrict__ C,
                                                              long ldc, long M, long N, long K, int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[TM * TP];
    __shared__ __attribute__((aligned(16))) float Bs[TN * TP];
    // block id -> (m tile, n tile): id = 8 slot + xcd, m tile = 8 (slot / NT) + xcd, n tile = slot % NT.  The NT workgroups
    // that share an A tile get ids 8 apart: same XCD (same L2), dispatched together - A comes from HBM once, not NT times.
    const int NT = (int)((N + TN - 1) / TN);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const long m0 = (long)(8 * (slot / NT) + xcd) * TM, n0 = (long)(slot % NT) * TN;
    if (m0 >= M) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    TStage<TM, A_KMAJOR> ra;
    TStage<TN, B_KMAJOR> rb;
    const long nkt = (K + TK - 1) / TK;
    const __amdgpu_buffer_rsrc_t rA = A_KMAJOR ? operand_rsrc(A, K, lda, M) : operand_rsrc(A, M, lda, K);
    const __amdgpu_buffer_rsrc_t rB = B_KMAJOR ? operand_rsrc(B, K, ldb, N) : operand_rsrc(B, N, ldb, K);
    ra.load(rA, (int)lda, (int)m0, (int)M, 0, (int)K);
    rb.load(rB, (int)ldb, (int)n0, (int)N, 0, (int)K);
    const float* pa = As + (64 * wm + i) * TP + 4 * h;          // + 32 TP: the second m sub-tile
    const float* pb = Bs + (32 * wn + i) * TP + 4 * h;
    for (long kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                        // the previous tile's fragment reads are done
        ra.store(As);
        rb.store(Bs);
        __syncthreads();
        {                                                       // branch-free: past the last tile the loads return zeros
            ra.load(rA, (int)lda, (int)m0, (int)M, (int)(kt + 1) * TK, (int)K);
            rb.load(rB, (int)ldb, (int)n0, (int)N, (int)(kt + 1) * TK, (int)K);
        }
        asm volatile("" ::: "memory");                          // keep the prefetch in front of the multiply
#pragma unroll
        for (int g = 0; g < TK / 8; ++g) {
            const float4 a0 = *reinterpret_cast<const float4*>(pa + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(pa + 32 * TP + 8 * g);
            const float4 b = *reinterpret_cast<const float4*>(pb + 8 * g);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
        }
    }
    const long col = n0 + 32 * wn + i;
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 64 * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) {
                    float v = (a == 0 ? acc0[r] : acc1[r]) + bv;
                    if (accumulate) v += C[row * ldc + col];
                    C[row * ldc + col] = v;
                }
            }
    }
}

// ---- the 2000 -> 500 layer over a whole grid (M = 4992 B spots, the largest item of a count-only step): the same LDS
// images, a workgroup of 256 m x 128 n (a wave 128 m x 64 n: 4 x 2 accumulators, 6 fragment reads per 32 MFMAs) - 12 B of
// operand per kFLOP through the load path instead of 23 (the 128 x 64 form above sat at 60 TFLOP/s on ~10 B per cycle and
// CU) - and the K range split over `S` workgroups so that ~2 per CU exist although M x N is only 20 x 4 such tiles.
// S > 1: partial tiles go to slabs [S][M][N], summed in index order (+ bias) by gemm_split_reduce_kernel: deterministic.
// Block id = 8 slot + xcd; unit (m tile, split) = 8 (slot / NT) + xcd, n tile = slot % NT: the NT workgroups that share an
// A strip run on one XCD, together.
constexpr int WM = 256, WN = 128;
template <bool A_KMAJOR, bool B_KMAJOR, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_f32_wide_kernel(const float* __restrict__ A, long lda,
                                                               const float* __restrict__ B, long ldb,
                                                               const float* __restrict__ bias, float* __restrict__ C,
                                                               long ldc, long M, long N, long K, int accumulate,
                                                               int S, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) float As[WM * TP];
    __shared__ __attribute__((aligned(16))) float Bs[WN * TP];
    const int NT = (int)((N + WN - 1) / WN), MT = (int)((M + WM - 1) / WM);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = 8 * (slot / NT) + xcd;
    if (unit >= MT * S) return;
    const int split = unit / MT;
    const long m0 = (long)(unit % MT) * WM, n0 = (long)(slot % NT) * WN;
    const long nkt = (K + TK - 1) / TK, per = (nkt + S - 1) / S;
    const long kt0 = split * per, kt1 = kt0 + per < nkt ? kt0 + per : nkt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    TStage<WM, A_KMAJOR> ra;
    TStage<WN, B_KMAJOR> rb;
    const __amdgpu_buffer_rsrc_t rA = A_KMAJOR ? operand_rsrc(A, K, lda, M) : operand_rsrc(A, M, lda, K);
    const __amdgpu_buffer_rsrc_t rB = B_KMAJOR ? operand_rsrc(B, K, ldb, N) : operand_rsrc(B, N, ldb, K);
    ra.load(rA, (int)lda, (int)m0, (int)M, (int)kt0 * TK, (int)K);
    rb.load(rB, (int)ldb, (int)n0, (int)N, (int)kt0 * TK, (int)K);
    const float* pa = As + (128 * wm + i) * TP + 4 * h;         // + 32 a TP: m sub-tile a
    const float* pb = Bs + (64 * wn + i) * TP + 4 * h;          // + 32 b TP: n sub-tile b
    for (long kt = kt0; kt < kt1; ++kt) {
        __syncthreads();                                        // the previous tile's fragment reads are done
        ra.store(As);
        rb.store(Bs);
        __syncthreads();
        {                                                       // branch-free (a split's last prefetch is simply not used)
            ra.load(rA, (int)lda, (int)m0, (int)M, (int)(kt + 1) * TK, (int)K);
            rb.load(rB, (int)ldb, (int)n0, (int)N, (int)(kt + 1) * TK, (int)K);
        }
        asm volatile("" ::: "memory");                          // keep the prefetch in front of the multiply
#pragma unroll
        for (int g = 0; g < TK / 8; ++g) {
            float4 fa[4], fb[2];
#pragma unroll
            for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const float4*>(pa + 32 * a * TP + 8 * g);
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const float4*>(pb + 32 * b * TP + 8 * g);
#define GNX_GW_STEP(c)                                                                                   \
    _Pragma("unroll") for (int a = 0; a < 4; ++a)                                                        \
        _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                    \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].c, fb[b].c, acc[a][b], 0, 0, 0);
            GNX_GW_STEP(x) GNX_GW_STEP(y) GNX_GW_STEP(z) GNX_GW_STEP(w)
#undef GNX_GW_STEP
        }
    }
    float* dst = S > 1 ? slabs + (size_t)split * M * N : C;
    const long ldd = S > 1 ? N : ldc;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const long col = n0 + 64 * wn + 32 * b + i;
        if (col >= N) continue;
        const float bv = (S == 1 && bias) ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 128 * wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) {
                    float v = acc[a][b][r] + bv;
                    if (S == 1 && accumulate) v += C[row * ldc + col];
                    dst[row * ldd + col] = v;
                }
            }
    }
}

// C[m][n] = bias[n] + sum_s slabs[s][m][n] (+ C[m][n]), slabs added in index order
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ slabs, int S, long M, long N,
                                                                const float* __restrict__ bias, float* __restrict__ C,
                                                                long ldc, int accumulate) {
    const long total = M * N;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / N, n = idx - m * N;
        float v = slabs[idx];
        for (int s2 = 1; s2 < S; ++s2) v += slabs[(size_t)s2 * total + idx];
        if (bias) v += bias[n];
        if (accumulate) v += C[m * ldc + n];
        C[m * ldc + n] = v;
    }
}

// K splits of the wide form for this shape (1 = no workspace needed; 0 = the shape is not the wide form's)
int gemm_wide_splits(long M, long N, long K) {
    // (round 5, measured: ONE column of tiles - the composed 2000 -> 100 count MLP over a 4 992-spot grid - split 8 ways took
    //  60.7 + 5.3 us against 53 us on the 64 x 64 kernel's 156 workgroups: N >= 256 stays the rule)
    if (M < 2048 || N < 256 || K < 512) return 0;
    const long tiles = gnx_cdiv(M, WM) * gnx_cdiv(N, WN), nkt = gnx_cdiv(K, TK);
    const int target = 512;
    long s = target / tiles;
    if (s > nkt / 4) s = nkt / 4;                              // at least 4 K tiles per workgroup
    if (s > 16) s = 16;
    return (int)(s < 1 ? 1 : s);
}

// K splits of the 64 x 64 kernel (1 = none) for narrow products with a long K, whose tiles do not fill the chip (the composed
// 2000 -> 100 layer of the count MLP over a 4 992-spot grid: 156 tiles for 256 CUs).  A function of (N, K) ONLY - never of M -
// so that a row's result does not depend on how many rows the call has (f evaluated on a subset of the spots equals the same
// rows of the full evaluation bit for bit: tests/test_gpu_models.py); the slab workspace bounds it for absurd M.
// Measured (tools/diag/gemm_tall_time.py, M = 4992, K = 2000, N = 100, reduce included): 50.6 us unsplit, 45.8 / 36.0 / 37.7 /
// 37.4 / 38.7 us with 2 / 3 / 4 / 6 / 8 splits (the MFMA floor of the padded product is 16 us).
int gemm_tall_splits(long M, long N, long K) {
    if (N > 128 || K < 1024) return 1;
    long s = gnx_cdiv(K, BK) / 8;                              // at least 8 K tiles per workgroup
    if (s > 3) s = 3;
    if (s < 1 || s * M * N > (1L << 26)) return 1;
    return (int)s;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// floats of workspace gnx_gemm_f32_ws wants for this shape (0: none)
GNX_EXPORT long gnx_gemm_f32_workspace(long M, long N, long K) {
    const int s = gemm_wide_splits(M, N, K), t = gemm_tall_splits(M, N, K);
    return s > 1 ? (long)s * M * N : (s == 0 && t > 1 ? (long)t * M * N : 0);
}

static int gemm_f32_impl(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
                         const float* bias, float* C, long ldc, long M, long N, long K, int accumulate,
                         float* workspace, hipStream_t stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0 || ldc < N) return GNX_ERR_BAD_ARG;
    if (M == 0 || N == 0) return GNX_OK;
    // float4 loads need 16-B aligned bases and leading dimensions that keep every row 16-B aligned
    const int a_vec = aligned16(A) && (lda % 4 == 0);
    const int b_vec = aligned16(B) && (ldb % 4 == 0);
    // whole-grid batches: the LDS-transposed forms (need float4-able rows on both operands: K-major rows are M / N long);
    const long a_bytes = 4 * ((a_kmajor ? K : M) * lda + 16), b_bytes = 4 * ((b_kmajor ? K : N) * ldb + 16);
    const bool t_ok = a_vec && b_vec && (a_kmajor ? M % 4 == 0 : K % 4 == 0) &&
                      (b_kmajor ? N % 4 == 0 : K % 4 == 0) && a_bytes < (1L << 31) && b_bytes < (1L << 31) &&
                      (K + 2 * TK) * (lda > ldb ? lda : ldb) < (1L << 29);       // 32-bit byte offsets, one tile past the end
    int splits = t_ok ? gemm_wide_splits(M, N, K) : 0;
    if (splits > 1 && !workspace) splits = 1;
    if (splits >= 1) {
        const long units8 = (gnx_cdiv(M, WM) * splits + 7) / 8 * 8;
        const long nwg = units8 * gnx_cdiv(N, WN);
        if (nwg > (1L << 30)) return GNX_ERR_UNSUPPORTED;
#define GNX_LAUNCHW(AK, BKM)                                                                                               \
    do {                                                                                                                   \
        gemm_f32_wide_kernel<AK, BKM, 2><<<(unsigned)nwg, 256, 0, stream>>>(A, lda, B, ldb, bias, C, ldc, M, N, K,         \
                                                                           accumulate, splits, workspace);                \
    } while (0)
        if (!a_kmajor && !b_kmajor) GNX_LAUNCHW(false, false);
        else if (a_kmajor && !b_kmajor) GNX_LAUNCHW(true, false);
        else if (!a_kmajor && b_kmajor) GNX_LAUNCHW(false, true);
        else GNX_LAUNCHW(true, true);
#undef GNX_LAUNCHW
        if (splits > 1) {
            long blocks = gnx_cdiv(M * N, 256);
            if (blocks > 2048) blocks = 2048;
            gemm_split_reduce_kernel<<<(unsigned)blocks, 256, 0, stream>>>(workspace, splits, M, N, bias, C, ldc, accumulate);
        }
        return gnx_launch_status();
    }
    const bool big_ok = t_ok && M >= 2048 && N >= 256;
    if (big_ok) {
        const long mt8 = (gnx_cdiv(M, TM) + 7) / 8 * 8;           // m tiles rounded up to whole XCD rounds (extras exit)
        if (mt8 * gnx_cdiv(N, TN) > (1L << 30)) return GNX_ERR_UNSUPPORTED;
        dim3 gridb((unsigned)(mt8 * gnx_cdiv(N, TN)));
#define GNX_LAUNCHB(AK, BKM) \
    gemm_f32_big_kernel<AK, BKM><<<gridb, 256, 0, stream>>>(A, lda, B, ldb, bias, C, ldc, M, N, K, accumulate)
        if (!a_kmajor && !b_kmajor) GNX_LAUNCHB(false, false);
        else if (a_kmajor && !b_kmajor) GNX_LAUNCHB(true, false);
        else if (!a_kmajor && b_kmajor) GNX_LAUNCHB(false, true);
        else GNX_LAUNCHB(true, true);
#undef GNX_LAUNCHB
        return gnx_launch_status();
    }
    const int tall = workspace ? gemm_tall_splits(M, N, K) : 1;
    dim3 grid(gnx_cdiv(M, BM), gnx_cdiv(N, BN), tall);
    if (grid.y > 65535) return GNX_ERR_UNSUPPORTED;
    // buffer (vector) loads per operand: 16-B aligned rows of whole quads, 32-bit byte offsets up to one tile past the end
    const bool av = a_vec && (a_kmajor ? M % 4 == 0 : K % 4 == 0) && a_bytes < (1L << 31) && (K + 2 * BK) * lda < (1L << 29) &&
                    (M + 2 * BM) * lda < (1L << 29);
    const bool bv = b_vec && (b_kmajor ? N % 4 == 0 : K % 4 == 0) && b_bytes < (1L << 31) && (K + 2 * BK) * ldb < (1L << 29) &&
                    (N + 2 * BN) * ldb < (1L << 29);
#define GNX_LAUNCH2(AK, BKM, AV, BV) \
    gemm_f32_kernel<AK, BKM, AV, BV><<<grid, 256, 0, stream>>>(A, lda, B, ldb, bias, C, ldc, M, N, K, accumulate, workspace)
#define GNX_LAUNCH(AK, BKM)                       \
    do {                                          \
        if (av && bv) GNX_LAUNCH2(AK, BKM, true, true);        \
        else if (av) GNX_LAUNCH2(AK, BKM, true, false);        \
        else if (bv) GNX_LAUNCH2(AK, BKM, false, true);        \
        else GNX_LAUNCH2(AK, BKM, false, false);               \
    } while (0)
    if (!a_kmajor && !b_kmajor) GNX_LAUNCH(false, false);
    else if (a_kmajor && !b_kmajor) GNX_LAUNCH(true, false);
    else if (!a_kmajor && b_kmajor) GNX_LAUNCH(false, true);
    else GNX_LAUNCH(true, true);
#undef GNX_LAUNCH
#undef GNX_LAUNCH2
    if (tall > 1) {
        long blocks = gnx_cdiv(M * N, 256);
        if (blocks > 2048) blocks = 2048;
        gemm_split_reduce_kernel<<<(unsigned)blocks, 256, 0, stream>>>(workspace, tall, M, N, bias, C, ldc, accumulate);
    }
    return gnx_launch_status();
}

GNX_EXPORT int gnx_gemm_f32(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
                            const float* bias, float* C, long ldc, long M, long N, long K, int accumulate,
                            hipStream_t stream) {
    return gemm_f32_impl(A, lda, a_kmajor, B, ldb, b_kmajor, bias, C, ldc, M, N, K, accumulate, nullptr, stream);
}

// The same product with `workspace` (gnx_gemm_f32_workspace floats, may be NULL when that is 0): large shapes may then
// split K over workgroups and sum the slabs in a fixed order.
GNX_EXPORT int gnx_gemm_f32_ws(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
                               const float* bias, float* C, long ldc, long M, long N, long K, int accumulate,
                               float* workspace, hipStream_t stream) {
    return gemm_f32_impl(A, lda, a_kmajor, B, ldb, b_kmajor, bias, C, ldc, M, N, K, accumulate, workspace, stream);
}
