// conv2 of a dense layer (3x3, 128 -> 32, padding 1; /root/reference/gridnext/densenet.py:41) on an fp32 bottleneck that conv1
// stored activated (norm2 -> relu2 on its store), with SPLIT bf16 OPERANDS - the 3x3 companion of conv1x1_split.hip (opt-in).
// Every fp32 operand = hi + lo in bf16, a product = a_lo b_hi + a_hi b_lo + a_hi b_hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation; tensors in HBM stay fp32.  No Winograd: with 16-bit matrix instructions the pass is bound by the bottleneck's
// bytes (512 per pixel), not by multiplications.
//
// The convolution is nine SHIFTED products over one staged piece of the [pixels][128] matrix: a workgroup (8 waves) owns 256 or 512
// consecutive pixels (whole maps or parts of maps, any alignment) and stages them with S + 1 pixels on either side - the reach
// of the taps - 32 channels at a time as two bf16 planes in the LDS; tap (dy, dx) of pixel p reads row p + dy S + dx of that
// piece, and a lane whose tap falls outside the map reads a ROW OF ZEROS instead (one address select per tap and tile, no data
// select).  The 32-channel chunk of the weights comes pre-split from gnx_conv3x3_split_pack ([chunk][hi | lo][tap][32 n][32 k]).
// A wave: 32 or 64 pixels x 32 output channels, three accumulators per 32 pixels (one per product kind) summed at the end in a
// fixed order.
// As in conv1x1_split.hip the (tile, chunk) items of a workgroup form one flat sequence and the next item's loads are in flight
// while the current one multiplies.
#include "fwd_common.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int C3_CK = 32;                  // channels per chunk
constexpr int C3_NCH = 128 / C3_CK;        // chunks
constexpr int C3_RS = 80;                  // bytes between two LDS rows (32 bf16 + 16 B: 16 consecutive rows cover all 64 banks once)
constexpr int C3_WPLANE = 9 * 32 * C3_RS;  // one weight plane of a chunk in the LDS: [tap][n] rows
constexpr int C3_WCHUNK = 2 * 9 * 32 * C3_CK;   // 16-bit elements per packed weight chunk

__global__ __launch_bounds__(256) void conv3x3_split_pack_kernel(const float* __restrict__ W, __bf16* __restrict__ Wp) {
    const int i = blockIdx.x * 256 + threadIdx.x;                        // one (chunk, tap, n, kk)
    if (i >= C3_NCH * 9 * 32 * C3_CK) return;
    const int kk = i % C3_CK, n = i / C3_CK % 32, tap = i / (C3_CK * 32) % 9, chunk = i / (C3_CK * 32 * 9);
    const float w = W[((long)n * 128 + chunk * C3_CK + kk) * 9 + tap];   // [n][k][ky][kx], tap = 3 ky + kx
    const __bf16 hi = (__bf16)w;
    const __bf16 lo = (__bf16)(w - (float)hi);
    Wp[((long)(chunk * 2 + 0) * 9 + tap) * 32 * C3_CK + n * C3_CK + kk] = hi;
    Wp[((long)(chunk * 2 + 1) * 9 + tap) * 32 * C3_CK + n * C3_CK + kk] = lo;
}

// PB: 32-pixel blocks per wave (a workgroup owns 8 x 32 PB pixels).  PB = 2 shares every weight fragment between a wave's two
// blocks - 6 LDS reads per 6 matrix instructions instead of 8 - and halves the halo and the weight traffic per pixel; PB = 1 fills
// the chip on small matrices.  A pixel's sums are formed in the same order either way: the choice never changes a result.
template <int S, int PB>
__global__ __launch_bounds__(512, 1) void conv3x3_split_kernel(const float* __restrict__ A, long lda, const __bf16* __restrict__ Wp,
                                                               float* __restrict__ out, long ldc, long M, long tiles) {
    constexpr int C3_ROWS = 256 * PB;
    constexpr int HALO = S + 1;
    constexpr int EXT = C3_ROWS + 2 * HALO;                              // staged pixel rows
    constexpr int APLANE = (EXT + 1) * C3_RS;                            // (+ the row of zeros)
    constexpr int NPA = (EXT * 8 + 511) / 512;                           // 16-B pieces of a chunk of A per thread
    constexpr int NPW = (C3_WCHUNK * 2 / 16 + 511) / 512;                // ... of a chunk of W
    __shared__ __attribute__((aligned(16))) char smem[2 * APLANE + 2 * C3_WPLANE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    if ((long)blockIdx.x >= tiles) return;
    if (t < 2 * C3_RS / 4) reinterpret_cast<unsigned*>(smem + (t < C3_RS / 4 ? 0 : APLANE) + EXT * C3_RS)[t % (C3_RS / 4)] = 0u;
    f32x16 acc[PB][3];
    f32x4 av[NPA];
    u32x4 wv[NPW];
    long ftile = blockIdx.x;
    int fc = 0;
#define C3_FETCH()                                                                                                           \
    {                                                                                                                        \
        const long g0 = ftile * C3_ROWS - HALO;                                                                              \
        _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                                    \
            const int L = i * 512 + t;                                                                                       \
            long row = g0 + (L >> 3);                            /* (pieces beyond EXT re-read a valid row; never written) */ \
            row = row < 0 ? 0 : (row < M ? row : M - 1);                                                                     \
            av[i] = *reinterpret_cast<const f32x4*>(A + row * lda + fc * C3_CK + (L & 7) * 4);                               \
        }                                                                                                                    \
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(Wp + (long)fc * C3_WCHUNK);                                       \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                                    \
            const int L = i * 512 + t;                                                                                       \
            wv[i] = wsrc[L < C3_WCHUNK * 2 / 16 ? L : 0];                                                                    \
        }                                                                                                                    \
    }
    C3_FETCH();
    int atap[PB][9];                                                     // LDS address of this lane's row per block and tap (hi plane, k = 8 h)
    for (;;) {
        const long ptile = ftile;
        const int pc = fc;
        if (pc == 0) {
            // which taps of this lane's pixels fall inside their map: the others read the row of zeros
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int e0 = 32 * PB * wave + 32 * j + r;
                const long p = ptile * C3_ROWS + e0;
                const int x = (int)(p & (S - 1)), y = (int)((p / S) & (S - 1));
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    const bool ok = (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S;
                    atap[j][tap] = (ok ? HALO + e0 + dy * S + dx : EXT) * C3_RS + 16 * h;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[j][k][q] = 0.f;
            }
        }
        lds_barrier();                                                   // the previous item's fragment reads are done
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int L = i * 512 + t;
            if (L < EXT * 8) {
                bf4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = (__bf16)av[i][e];
                    lo[e] = (__bf16)(av[i][e] - (float)hi[e]);
                }
                char* const pd = smem + (L >> 3) * C3_RS + (L & 7) * 8;
                *reinterpret_cast<bf4*>(pd) = hi;
                *reinterpret_cast<bf4*>(pd + APLANE) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int L = i * 512 + t;                                   // 16-B piece of [plane][tap][n][32 k]: 4 pieces per row
            if (L < C3_WCHUNK * 2 / 16) *reinterpret_cast<u32x4*>(smem + 2 * APLANE + (L >> 2) * C3_RS + (L & 3) * 16) = wv[i];
        }
        lds_barrier();
        if (++fc == C3_NCH) {
            fc = 0;
            ftile += gridDim.x;
        }
        const bool more = ftile < tiles;
        if (more) C3_FETCH();
        const char* const wb = smem + 2 * APLANE + r * C3_RS + 16 * h;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf8 b_hi = *reinterpret_cast<const bf8*>(wb + tap * 32 * C3_RS + 32 * s);
                const bf8 b_lo = *reinterpret_cast<const bf8*>(wb + C3_WPLANE + tap * 32 * C3_RS + 32 * s);
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const char* const pa = smem + atap[j][tap] + 32 * s;
                    const bf8 a_hi = *reinterpret_cast<const bf8*>(pa);
                    const bf8 a_lo = *reinterpret_cast<const bf8*>(pa + APLANE);
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[j][1], 0, 0, 0);
                    acc[j][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[j][2], 0, 0, 0);
                }
            }
        }
        if (pc == C3_NCH - 1) {
            // D[i][j]: i = (q & 3) + 8 (q >> 2) + 4 h the pixel of the block's 32, j = lane & 31 the output channel
            const bool whole = ptile * C3_ROWS + C3_ROWS <= M;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const long p0 = ptile * C3_ROWS + 32 * PB * wave + 32 * j + 4 * h;
                float* const po = out + p0 * ldc + r;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = (q & 3) + 8 * (q >> 2);
                    const float v = (acc[j][0][q] + acc[j][1][q]) + acc[j][2][q];
                    if (whole || p0 + i < M) po[i * ldc] = v;
                }
            }
        }
        if (!more) break;
    }
#undef C3_FETCH
}

}  // namespace

// W: conv2.weight [32][128][3][3] fp32 -> Wp: 4 chunks of [hi | lo][tap][32 n][32 k] bf16 (gnx_conv3x3_split_pack_halves() elements).
GNX_EXPORT long gnx_conv3x3_split_pack_halves(void) { return (long)C3_NCH * C3_WCHUNK; }
GNX_EXPORT int gnx_conv3x3_split_pack(const float* W, void* Wp, hipStream_t stream) {
    if (!W || !Wp) return GNX_ERR_BAD_ARG;
    conv3x3_split_pack_kernel<<<(C3_NCH * 9 * 32 * C3_CK + 255) / 256, 256, 0, stream>>>(W, static_cast<__bf16*>(Wp));
    return gnx_launch_status();
}
// out[p][n] = sum over taps and k of A[p + (dy, dx)][k] W[n][k][dy + 1][dx + 1] on maps of S x S pixels (M = images * S * S rows of
// A, zero padding at the map borders): gnx_conv3x3_bnrelu without prologue for N = 32, K = 128 on split bf16 operands.  A, out:
// fp32; 16-B aligned A, 4 | lda.  S in {4, 8, 16, 32, 64}, S * S | M.  Anything else: GNX_ERR_UNSUPPORTED.
GNX_EXPORT int gnx_conv3x3_split(const float* A, long lda, const void* Wp, float* out, long ldc, long M, int S, hipStream_t stream) {
    if (!A || !Wp || !out || M < 0 || lda < 128 || ldc < 32 || S < 1) return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    if ((lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(Wp) & 15) || M % ((long)S * S))
        return GNX_ERR_UNSUPPORTED;
    // 64 pixels per wave where the matrix still gives every CU four tiles or more, else 32
    const int pb = (M + 511) / 512 >= 1024 ? 2 : 1;
    const long tiles = (M + 256 * pb - 1) / (256 * pb);
    const int grid = (int)(tiles < 256 ? tiles : 256);
    const __bf16* const wp = static_cast<const __bf16*>(Wp);
#define C3_GO(SS)                                                                                              \
    if (pb == 2) conv3x3_split_kernel<SS, 2><<<grid, 512, 0, stream>>>(A, lda, wp, out, ldc, M, tiles);        \
    else conv3x3_split_kernel<SS, 1><<<grid, 512, 0, stream>>>(A, lda, wp, out, ldc, M, tiles);                \
    break;
    switch (S) {
        case 4: C3_GO(4)
        case 8: C3_GO(8)
        case 16: C3_GO(16)
        case 32: C3_GO(32)
        case 64: C3_GO(64)
        default: return GNX_ERR_UNSUPPORTED;
    }
#undef C3_GO
    return gnx_launch_status();
}
