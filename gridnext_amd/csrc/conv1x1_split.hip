// norm1 -> relu1 -> conv1 (-> norm2 -> relu2) of a dense layer (/root/reference/gridnext/densenet.py:35-40) on fp32 tensors with
// SPLIT OPERANDS on the 16-bit matrix cores (opt-in form of gnx_conv1x1_bnrelu_act; round 5, a first kernel of the next round's plan).
//
// v_mfma_f32_32x32x2_f32 - the instruction the fp32 path is bound by - runs at 157 TFLOP/s; v_mfma_f32_32x32x16_bf16 at 16x that.
// Every fp32 operand is written as the sum of two bf16 numbers, a = a_hi + a_lo (a_hi = a rounded to bf16, a_lo = a - a_hi rounded
// to bf16: together 16 significant bits), and a product as THREE matrix instructions with fp32 accumulation,
//     a b ~= a_lo b_hi + a_hi b_lo + a_hi b_hi,
// dropping a_lo b_lo (2^-16 of the product).  bf16 keeps fp32's exponent range, so there is nothing to scale and nothing that can
// overflow.  tools/diag/split_operand_feasibility.py puts a whole DenseNet-121 through this arithmetic on the CPU: logits 7e-6 of
// their range and the cross entropy 1e-6 from float64 (plain fp32: 5e-7 / 7e-8; fp16 operands: 9e-4 / 8e-5) - two orders of
// magnitude inside north_star's 1e-4.  Tensors in HBM stay fp32: with three 16-bit instructions per product the kernel is bound
// by HBM (A is read once, 4 bytes per element), not by the matrix pipe.
//
// A workgroup (8 waves) owns 256 rows x 128 output channels and walks K in chunks of 64: the chunk of A goes global -> registers
// (16-B loads, a row's 256 B contiguous) -> norm1 / relu1 -> split -> two bf16 planes in the LDS; the chunk of W comes pre-split
// and zero-padded from gnx_conv1x1_split_pack ([chunk][hi | lo][128 n][64 k]); the next chunk's loads are in flight while the
// current one multiplies (a wave: 32 rows x 128 channels, 12 matrix instructions per 16 k against 10 LDS reads of 16 B).
#include "fwd_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (a plain vector: HIP's uint4 struct kept the staging array in memory)

constexpr int SP_CK = 64;                 // k per chunk
constexpr int SP_RS = 144;                // bytes between two rows of an LDS plane (64 bf16 + 16 B: ds_read_b128 of 32 rows conflict-free)
constexpr int SP_PLANE = 128 * SP_RS;     // one [128][64] bf16 plane (W)
constexpr int SP_ROWS = 256;              // rows of A per workgroup (8 waves x 32)
constexpr int SP_A = 2 * SP_ROWS * SP_RS; // A hi + A lo

__global__ __launch_bounds__(256) void conv1x1_split_pack_kernel(const float* __restrict__ W, __bf16* __restrict__ Wp, int K, int nchunks) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;                 // one (chunk, n, kk)
    if (i >= (long)nchunks * 128 * SP_CK) return;
    const int kk = (int)(i % SP_CK), n = (int)(i / SP_CK % 128), chunk = (int)(i / (SP_CK * 128));
    const int k = chunk * SP_CK + kk;
    const float w = k < K ? W[(long)n * K + k] : 0.f;
    const __bf16 hi = (__bf16)w;
    const __bf16 lo = (__bf16)(w - (float)hi);
    Wp[((long)(chunk * 2 + 0) * 128 + n) * SP_CK + kk] = hi;
    Wp[((long)(chunk * 2 + 1) * 128 + n) * SP_CK + kk] = lo;
}

__global__ __launch_bounds__(512, 1) void conv1x1_split_kernel(const float* __restrict__ A, long lda, const __bf16* __restrict__ Wp,
                                                               float* __restrict__ out, long ldc, long M, int K,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ oscale, const float* __restrict__ oshift,
                                                               long tiles) {
    __shared__ __attribute__((aligned(16))) char smem[SP_A + 2 * SP_PLANE];   // A hi | A lo ([256][64]) | W hi | W lo ([128][64])
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int col = t & 15, row0 = t >> 4;                               // A loads: thread = 4 k of one row, 32 rows per pass
    const int nchunks = (K + SP_CK - 1) / SP_CK;
    if ((long)blockIdx.x >= tiles) return;
    // One flat sequence of (tile, chunk) items per workgroup: while item n is written to the LDS and multiplied, item n + 1 - the
    // next chunk, or the first chunk of the workgroup's NEXT tile - is already in flight (one fetch site; no wait between a load
    // and the point its value is needed).  Rows beyond M re-read the last row (never stored); columns beyond K re-read the row's
    // last four and are zeroed through scale = shift = 0 at the split (W is zero-padded there too).
    f32x16 acc[4];
    f32x4 av[8], sc, sh;
    u32x4 wv[4];
    bool ktail;
    float os[4], ob[4];                                                  // norm2 of this lane's four output channels
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        os[nb] = oscale ? oscale[32 * nb + r] : 1.f;
        ob[nb] = oscale ? oshift[32 * nb + r] : 0.f;
    }
    long ftile = blockIdx.x;                                             // the item being fetched
    int fc = 0;
#define SP_FETCH()                                                                                                          \
    {                                                                                                                       \
        const int k = fc * SP_CK + col * 4;                                                                                 \
        ktail = k >= K;                                                  /* 4 | K: a thread's four columns are all in or all out */ \
        const int kc = ktail ? K - 4 : k;                                                                                   \
        const long rbase = ftile * SP_ROWS + row0, rlast = M - 1 - rbase;                                                   \
        sc = *reinterpret_cast<const f32x4*>(scale + kc);                                                                   \
        sh = *reinterpret_cast<const f32x4*>(shift + kc);                                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                       \
            av[i] = *reinterpret_cast<const f32x4*>(A + (rbase + (32 * i < rlast ? 32 * i : rlast)) * lda + kc);            \
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(Wp + (long)fc * 2 * 128 * SP_CK);                                \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) wv[i] = wsrc[i * 512 + t];                                            \
    }
    SP_FETCH();
    for (;;) {
        const long ptile = ftile;                                        // the item in the staging registers
        const int pc = fc;
        lds_barrier();                                                   // the previous item's fragment reads are done
        {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 s4 = ktail ? z : sc, b4 = ktail ? z : sh;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(fmaf(av[i][e], s4[e], b4[e]), 0.f);
                    hi[e] = (__bf16)v;
                    lo[e] = (__bf16)(v - (float)hi[e]);
                }
                char* const p = smem + (row0 + 32 * i) * SP_RS + col * 8;
                *reinterpret_cast<bf4*>(p) = hi;
                *reinterpret_cast<bf4*>(p + SP_A / 2) = lo;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int L = i * 512 + t;                               // 16-B piece of [plane][n][64 k]
                *reinterpret_cast<u32x4*>(smem + SP_A + (L >> 10) * SP_PLANE + ((L & 1023) >> 3) * SP_RS + (L & 7) * 16) = wv[i];
            }
        }
        lds_barrier();
        if (++fc == nchunks) {
            fc = 0;
            ftile += gridDim.x;
        }
        const bool more = ftile < tiles;
        if (more) SP_FETCH();
        if (pc == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char* const pa = smem + (32 * wave + r) * SP_RS + (16 * s + 8 * h) * 2;
            const bf8 a_hi = *reinterpret_cast<const bf8*>(pa);
            const bf8 a_lo = *reinterpret_cast<const bf8*>(pa + SP_A / 2);
            bf8 b_hi[4], b_lo[4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const char* const pb = smem + SP_A + (32 * nb + r) * SP_RS + (16 * s + 8 * h) * 2;
                b_hi[nb] = *reinterpret_cast<const bf8*>(pb);
                b_lo[nb] = *reinterpret_cast<const bf8*>(pb + SP_PLANE);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {                             // (small terms first)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi[nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo[nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi[nb], acc[nb], 0, 0, 0);
            }
        }
        if (pc == nchunks - 1) {
            // D[i][j]: i = (q & 3) + 8 (q >> 2) + 4 h the row of A, j = lane & 31 the output channel of the 32-wide block
            const long m0 = ptile * SP_ROWS;
            const bool whole = m0 + SP_ROWS <= M;                        // (all tiles but the last: stores without a branch each)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float* const po = out + (m0 + 32 * wave + 4 * h) * ldc + 32 * nb + r;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = (q & 3) + 8 * (q >> 2);
                    float v = acc[nb][q];
                    if (oscale) v = fmaxf(fmaf(v, os[nb], ob[nb]), 0.f);
                    if (whole || m0 + 32 * wave + 4 * h + i < M) po[i * ldc] = v;
                }
            }
        }
        if (!more) break;
    }
#undef SP_FETCH
}

}  // namespace

// W: conv1.weight as [128][K] fp32 (row n = output channel) -> Wp: (K + 63) / 64 chunks of [hi | lo][128][64] bf16, zero beyond K.
GNX_EXPORT long gnx_conv1x1_split_pack_halves(int K) { return K < 1 ? 0 : (long)((K + SP_CK - 1) / SP_CK) * 2 * 128 * SP_CK; }
GNX_EXPORT int gnx_conv1x1_split_pack(const float* W, void* Wp, int K, hipStream_t stream) {
    if (!W || !Wp || K < 1) return GNX_ERR_BAD_ARG;
    const int nchunks = (K + SP_CK - 1) / SP_CK;
    const long n = (long)nchunks * 128 * SP_CK;
    conv1x1_split_pack_kernel<<<(int)((n + 255) / 256), 256, 0, stream>>>(W, static_cast<__bf16*>(Wp), K, nchunks);
    return gnx_launch_status();
}
// out[m][n] = act( sum_k relu(scale[k] A[m][k] + shift[k]) W[n][k] ),  act = relu(out_scale[n] . + out_shift[n]) or, with
// out_scale = NULL, the identity: gnx_conv1x1_bnrelu_act / gnx_conv1x1_bnrelu (pool = 0) for N = 128 on split bf16 operands.
// A, out: fp32, 16-B aligned, 4 | lda; 4 | K.  Anything else: GNX_ERR_UNSUPPORTED (the caller keeps the fp32 instruction).
GNX_EXPORT int gnx_conv1x1_bnrelu_act_split(const float* A, long lda, const void* Wp, float* out, long ldc, long M, int K,
                                            const float* scale, const float* shift, const float* out_scale,
                                            const float* out_shift, hipStream_t stream) {
    if (!A || !Wp || !out || !scale || !shift || M < 0 || K < 1 || lda < K || ldc < 128 || (out_scale && !out_shift))
        return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    if ((K & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(scale) & 15) ||
        (reinterpret_cast<uintptr_t>(shift) & 15) || (reinterpret_cast<uintptr_t>(Wp) & 15))
        return GNX_ERR_UNSUPPORTED;
    const long tiles = (M + SP_ROWS - 1) / SP_ROWS;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    conv1x1_split_kernel<<<grid, 512, 0, stream>>>(A, lda, static_cast<const __bf16*>(Wp), out, ldc, M, K, scale, shift, out_scale,
                                                   out_shift, tiles);
    return gnx_launch_status();
}
