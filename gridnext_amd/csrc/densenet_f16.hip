// fp16-MFMA variants of the two DenseNet conv kernels (BASELINE.json config 5: "fp16 MFMA conv path").
//
// Same contract as gnx_conv1x1_bnrelu / gnx_conv3x3_bnrelu (conv1x1.hip, conv3x3.hip): fp32 activations and weights in HBM,
// fp32 BN+ReLU prologue, fp32 accumulation, fp32 outputs.  Only the MFMA operands are narrowed: values are rounded to
// IEEE fp16 as they are written to LDS and multiplied with v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate), so the
// result differs from the fp32 path by operand rounding (~2^-11 relative per product); tests state the tolerance and
// report the CE difference - no 1e-4 claim is made for this path.
// LDS image: [row][32 halves + 8 pad] = 80-B rows (5 sixteen-byte slots, odd => ds_read_b128 conflict-free); a lane
// reads its 8 consecutive k (k = 16*s + 8*h + j) with one ds_read_b128 per operand per MFMA.
// This first version keeps the fp32 kernels' tiling (K chunks of 32); with the matrix phase 16x shorter it is bound
// by staging, not by the matrix cores - restructuring (whole-K tiles, fp16 activations in HBM) is future work.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 40;      // halves per LDS row

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float act1(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }
__device__ __forceinline__ half4 to_h4(float4 v) {
    half4 r = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    return r;
}
__device__ __forceinline__ half8 ldh8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }
// 4 consecutive activations at element offset `off` of a matrix stored as fp32 or (IN16) fp16
template <bool IN16>
__device__ __forceinline__ float4 lda4(const float* A, long off) {
    if constexpr (IN16) {
        const half4 h = *reinterpret_cast<const half4*>(reinterpret_cast<const _Float16*>(A) + off);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else {
        return ld4(A + off);
    }
}

// ---- conv1x1: 128x128 tile, 4 waves x (64x64), K chunks of 32; requires aligned pointers and K % 4 == 0
// OUT16: the store applies the consumer's BN + ReLU and rounds to fp16 (out is _Float16 [M][N], ldc in halves): the
// bottleneck as conv3x3_dma_kernel<.., H16> takes it.
// IN16: A holds fp16 activations (lda in halves): config 5 with fp16 block buffers.
template <bool POOL, bool OUT16 = false, bool IN16 = false>
__global__ __launch_bounds__(256) void conv1x1_f16_kernel(const float* __restrict__ A, long lda,
                                                          const float* __restrict__ W, float* __restrict__ out, long ldc,
                                                          long M, int N, int K, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int S_in,
                                                          const float* __restrict__ oscale = nullptr,
                                                          const float* __restrict__ oshift = nullptr) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 128 * LDH];
    _Float16* const As = smem;
    _Float16* const Bs = smem + 128 * LDH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    const int kq = t & 7, r0 = t >> 3;
    const long m0 = (long)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    const bool has_act = scale != nullptr;
    long src[4];
    bool rok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long row = m0 + r0 + 32 * p;
        rok[p] = row < M;
        const long rr = rok[p] ? row : 0;
        if (POOL) {
            const int So = S_in >> 1;
            const long img = rr / (So * So);
            const int rem = (int)(rr - img * So * So);
            const int oy = rem / So, ox = rem - oy * So;
            src[p] = ((img * S_in + 2 * oy) * S_in + 2 * ox) * lda;
        } else {
            src[p] = rr * lda;
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nkt = (K + 31) / 32;
    auto mfma_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const half8 a0 = ldh8(&As[(64 * wm + i) * LDH + 16 * s + 8 * h]);
            const half8 a1 = ldh8(&As[(64 * wm + 32 + i) * LDH + 16 * s + 8 * h]);
            const half8 b0 = ldh8(&Bs[(64 * wn + i) * LDH + 16 * s + 8 * h]);
            const half8 b1 = ldh8(&Bs[(64 * wn + 32 + i) * LDH + 16 * s + 8 * h]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    if constexpr (!POOL) {
        // register-prefetched: chunk kt+1's global loads are in flight while chunk kt is converted, staged and multiplied
        // (the kernel streams the fp32 block buffer: it is HBM-bound, the matrix phase is 1/16 of the fp32 kernel's)
        float4 va[4], vb[4];
        auto fetch = [&](int kt) {
            const int k = kt * 32 + 4 * kq;
            const int kc = k < K ? k : 0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                va[p] = lda4<IN16>(A, src[p] + kc);
                const int n = n0 + r0 + 32 * p;
                vb[p] = ld4(W + (long)(n < N ? n : N - 1) * K + kc);
            }
        };
        fetch(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int k = kt * 32 + 4 * kq;
            const bool kok = k < K;
            const int kc = kok ? k : 0;
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_act) { sc4 = ld4(scale + kc); sh4 = ld4(shift + kc); }
            __syncthreads();                               // the previous chunk's fragment reads are done
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = va[p];
                if (has_act) {
                    v.x = act1(v.x, sc4.x, sh4.x); v.y = act1(v.y, sc4.y, sh4.y);
                    v.z = act1(v.z, sc4.z, sh4.z); v.w = act1(v.w, sc4.w, sh4.w);
                }
                if (!(rok[p] && kok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 w = vb[p];
                if (!(n0 + r0 + 32 * p < N && kok)) w = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<half4*>(&As[(r0 + 32 * p) * LDH + 4 * kq]) = to_h4(v);
                *reinterpret_cast<half4*>(&Bs[(r0 + 32 * p) * LDH + 4 * kq]) = to_h4(w);
            }
            __syncthreads();
            fetch(kt + 1 < nkt ? kt + 1 : kt);             // branch-free; the last one is a harmless re-read
            asm volatile("" ::: "memory");                 // keep the prefetch in front of the multiply
            mfma_chunk();
        }
    } else
    for (int kt = 0; kt < nkt; ++kt) {
        const int k = kt * 32 + 4 * kq;
        const bool kok = k < K;
        const int kc = kok ? k : 0;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act) { sc4 = ld4(scale + kc); sh4 = ld4(shift + kc); }
        float4 va[4], vb[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (POOL) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = lda4<IN16>(A, src[p] + ((q >> 1) * (long)S_in + (q & 1)) * lda + kc);
                    if (has_act) {
                        v.x = act1(v.x, sc4.x, sh4.x); v.y = act1(v.y, sc4.y, sh4.y);
                        v.z = act1(v.z, sc4.z, sh4.z); v.w = act1(v.w, sc4.w, sh4.w);
                    }
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
                va[p] = make_float4(0.25f * s.x, 0.25f * s.y, 0.25f * s.z, 0.25f * s.w);
            } else {
                float4 v = lda4<IN16>(A, src[p] + kc);
                if (has_act) {
                    v.x = act1(v.x, sc4.x, sh4.x); v.y = act1(v.y, sc4.y, sh4.y);
                    v.z = act1(v.z, sc4.z, sh4.z); v.w = act1(v.w, sc4.w, sh4.w);
                }
                va[p] = v;
            }
            if (!(rok[p] && kok)) va[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + r0 + 32 * p;
            vb[p] = ld4(W + (long)(n < N ? n : N - 1) * K + kc);
            if (!(n < N && kok)) vb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<half4*>(&As[(r0 + 32 * p) * LDH + 4 * kq]) = to_h4(va[p]);
            *reinterpret_cast<half4*>(&Bs[(r0 + 32 * p) * LDH + 4 * kq]) = to_h4(vb[p]);
        }
        __syncthreads();
        mfma_chunk();
    }
    if constexpr (OUT16) {
        // activated, rounded, and turned through the LDS (free after the K loop) so that a lane stores 16 B (8 channels of a
        // row) instead of 2: each wave owns [32 rows][64 + 8 halves] for its 32 x 64 half tile, one pass per mt (8 | N)
        constexpr int LDO = 72;
        _Float16* const Os = smem + wave * 32 * LDO;
        _Float16* const o16 = reinterpret_cast<_Float16*>(out);
        __syncthreads();                                   // everyone is done with the last chunk's fragments
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = n0 + 64 * wn + 32 * nt + i;
                const bool oact = oscale != nullptr;      // no consumer activation (transitions): plain rounding
                const float osc = (oact && col < N) ? oscale[col] : 1.f, osh = (oact && col < N) ? oshift[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaf(acc[mt][nt][r], osc, osh);
                    Os[((r & 3) + 8 * (r >> 2) + 4 * h) * LDO + 32 * nt + i] = (_Float16)(oact ? fmaxf(v, 0.f) : v);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {                  // 32 rows x 8 pieces of 16 B
                const int piece = lane + 64 * p, rr = piece >> 3, c8 = piece & 7;
                const long row = m0 + 64 * wm + 32 * mt + rr;
                const int col = n0 + 64 * wn + 8 * c8;
                if (row < M && col < N)
                    *reinterpret_cast<half8*>(o16 + row * ldc + col) = *reinterpret_cast<const half8*>(&Os[rr * LDO + 8 * c8]);
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + 64 * wn + 32 * nt + i;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) out[row * ldc + col] = acc[mt][nt][r];
            }
        }
}

// ---- conv1 on fp16 block buffers (dense layers of config 5): A16 [M][K] halves, W16 [N][K] halves (the weight rounded once),
// out16 = fp16(relu(osc * y + osh)).  128 x 128 tile, K chunks of 64: a thread moves 16 B (8 halves) per row piece, two
// workgroup barriers per 64 channels instead of per 32, chunk k+1's loads in flight while chunk k is activated, staged and
// multiplied.  LDS rows of 72 halves (9 sixteen-byte slots: conflict-free ds_read_b128).
constexpr int LDH2 = 72;
__global__ __launch_bounds__(256) void conv1x1_h16_kernel(const _Float16* __restrict__ A, long lda,
                                                          const _Float16* __restrict__ W, _Float16* __restrict__ out,
                                                          long ldc, long M, int N, int K,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ oscale,
                                                          const float* __restrict__ oshift, long obs = 32) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * 128 * LDH2];
    _Float16* const As = smem;
    _Float16* const Bs = smem + 128 * LDH2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    const int kq = t & 7, r0 = t >> 3;                     // this thread's 8 k of a chunk, first of its 4 rows
    const long m0 = (long)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;
    long srcA[4], srcW[4];
    bool rok[4], nok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long row = m0 + r0 + 32 * p;
        rok[p] = row < M;
        srcA[p] = (rok[p] ? row : 0) * lda;
        const int n = n0 + r0 + 32 * p;
        nok[p] = n < N;
        srcW[p] = (long)(nok[p] ? n : 0) * K;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nkt = (K + 63) / 64;
    half8 va[4], vb[4];
    auto fetch = [&](int kt) {
        const int k = kt * 64 + 8 * kq;
        const int kc = k < K ? k : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            va[p] = ldh8(A + srcA[p] + kc);
            vb[p] = ldh8(W + srcW[p] + kc);
        }
    };
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k = kt * 64 + 8 * kq;
        const bool kok = k < K;                            // 8 | K: the whole piece is in or out
        const int kc = kok ? k : 0;
        float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sh0;
        if (scale) { sc0 = ld4(scale + kc); sc1 = ld4(scale + kc + 4); sh0 = ld4(shift + kc); sh1 = ld4(shift + kc + 4); }
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        __syncthreads();                                   // the previous chunk's fragment reads are done
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            half8 a = va[p], w = vb[p];
            if (scale) {                                   // (NULL: the operand is taken as it is - a pooled, activated input)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (_Float16)act1((float)va[p][j], sc[j], sh[j]);
            }
            if (!(rok[p] && kok)) a = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (!(nok[p] && kok)) w = half8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<half8*>(&As[(r0 + 32 * p) * LDH2 + 8 * kq]) = a;
            *reinterpret_cast<half8*>(&Bs[(r0 + 32 * p) * LDH2 + 8 * kq]) = w;
        }
        __syncthreads();
        fetch(kt + 1 < nkt ? kt + 1 : kt);                 // branch-free; the last one is a harmless re-read
        asm volatile("" ::: "memory");                     // keep the prefetch in front of the multiply
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 a0 = ldh8(&As[(64 * wm + i) * LDH2 + 16 * s + 8 * h]);
            const half8 a1 = ldh8(&As[(64 * wm + 32 + i) * LDH2 + 16 * s + 8 * h]);
            const half8 b0 = ldh8(&Bs[(64 * wn + i) * LDH2 + 16 * s + 8 * h]);
            const half8 b1 = ldh8(&Bs[(64 * wn + 32 + i) * LDH2 + 16 * s + 8 * h]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // activated, rounded, turned through the LDS: 16-B stores (see conv1x1_f16_kernel<.., OUT16>)
    _Float16* const Os = smem + wave * 32 * LDH2;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + 64 * wn + 32 * nt + i;
            const bool oact = oscale != nullptr;          // NULL (transitions): the product is stored as it is, rounded
            const float osc = (oact && col < N) ? oscale[col] : 1.f, osh = (oact && col < N) ? oshift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(acc[mt][nt][r], osc, osh);
                Os[((r & 3) + 8 * (r >> 2) + 4 * h) * LDH2 + 32 * nt + i] = (_Float16)(oact ? fmaxf(v, 0.f) : v);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = lane + 64 * p, rr = piece >> 3, c8 = piece & 7;
            const long row = m0 + 64 * wm + 32 * mt + rr;
            const int col = n0 + 64 * wn + 8 * c8;
            if (row < M && col < N)       // (obs = 32: row-major; ldc = 32, obs = rows * 32: channel-blocked [N / 32][rows][32])
                *reinterpret_cast<half8*>(out + row * ldc + (col >> 5) * obs + (col & 31)) =
                    *reinterpret_cast<const half8*>(&Os[rr * LDH2 + 8 * c8]);
        }
    }
}


// ---- the same with 256-row tiles (round 2).  The kernel above moves, per 128-row tile and 64-channel chunk, 16 KB of
// activations AND 16 KB of weights from L2 into the CU: twice the activation bytes, and L2 -> CU is what bounded it (3.3 TB/s
// of activation traffic = 6.6 TB/s ~ 11 B per cycle and CU, the rate the load path delivers).  With 256 rows per workgroup a
// weight chunk serves twice the rows (1.5x the activation bytes).  4 waves, each 128 rows x 64 columns (4 x 2 accumulators):
// 6 fragment reads per 8 MFMAs.  55 KB of LDS: two workgroups per CU.
__global__ __launch_bounds__(256, 2) void conv1x1_h16_m256_kernel(const _Float16* __restrict__ A, long lda,
                                                                  const _Float16* __restrict__ W, _Float16* __restrict__ out,
                                                                  long ldc, long M, int N, int K,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ oscale,
                                                                  const float* __restrict__ oshift, long obs = 32) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[(256 + 128) * LDH2];
    _Float16* const As = smem;
    _Float16* const Bs = smem + 256 * LDH2;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    const int kq = t & 7, r0 = t >> 3;                     // this thread's 8 k of a chunk, first of its 8 (A) / 4 (W) rows
    const long m0 = (long)blockIdx.x * 256;
    const int n0 = blockIdx.y * 128;
    long srcA[8], srcW[4];
    bool rok[8], nok[4];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const long row = m0 + r0 + 32 * p;
        rok[p] = row < M;
        srcA[p] = (rok[p] ? row : 0) * lda;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n0 + r0 + 32 * p;
        nok[p] = n < N;
        srcW[p] = (long)(nok[p] ? n : 0) * K;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nkt = (K + 63) / 64;
    half8 va[8], vb[4];
    auto fetch = [&](int kt) {
        const int k = kt * 64 + 8 * kq;
        const int kc = k < K ? k : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) va[p] = ldh8(A + srcA[p] + kc);
#pragma unroll
        for (int p = 0; p < 4; ++p) vb[p] = ldh8(W + srcW[p] + kc);
    };
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int k = kt * 64 + 8 * kq;
        const bool kok = k < K;                            // 8 | K: the whole piece is in or out
        const int kc = kok ? k : 0;
        float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sh0;
        if (scale) { sc0 = ld4(scale + kc); sc1 = ld4(scale + kc + 4); sh0 = ld4(shift + kc); sh1 = ld4(shift + kc + 4); }
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        __syncthreads();                                   // the previous chunk's fragment reads are done
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            half8 a = va[p];
            if (scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (_Float16)act1((float)va[p][j], sc[j], sh[j]);
            }
            if (!(rok[p] && kok)) a = half8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<half8*>(&As[(r0 + 32 * p) * LDH2 + 8 * kq]) = a;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            half8 w = vb[p];
            if (!(nok[p] && kok)) w = half8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<half8*>(&Bs[(r0 + 32 * p) * LDH2 + 8 * kq]) = w;
        }
        __syncthreads();
        fetch(kt + 1 < nkt ? kt + 1 : kt);                 // branch-free; the last one is a harmless re-read
        asm volatile("" ::: "memory");                     // keep the prefetch in front of the multiply
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 b0 = ldh8(&Bs[(64 * wn + i) * LDH2 + 16 * s + 8 * h]);
            const half8 b1 = ldh8(&Bs[(64 * wn + 32 + i) * LDH2 + 16 * s + 8 * h]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const half8 af = ldh8(&As[(128 * wm + 32 * a + i) * LDH2 + 16 * s + 8 * h]);
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, b1, acc[a][1], 0, 0, 0);
            }
        }
    }
    // activated, rounded, turned through the LDS: 16-B stores
    _Float16* const Os = smem + wave * 32 * LDH2;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + 64 * wn + 32 * nt + i;
            const bool oact = oscale != nullptr;          // NULL (transitions): the product is stored as it is, rounded
            const float osc = (oact && col < N) ? oscale[col] : 1.f, osh = (oact && col < N) ? oshift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(acc[mt][nt][r], osc, osh);
                Os[((r & 3) + 8 * (r >> 2) + 4 * h) * LDH2 + 32 * nt + i] = (_Float16)(oact ? fmaxf(v, 0.f) : v);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int piece = lane + 64 * p, rr = piece >> 3, c8 = piece & 7;
            const long row = m0 + 128 * wm + 32 * mt + rr;
            const int col = n0 + 64 * wn + 8 * c8;
            if (row < M && col < N)
                *reinterpret_cast<half8*>(out + row * ldc + (col >> 5) * obs + (col & 31)) =
                    *reinterpret_cast<const half8*>(&Os[rr * LDH2 + 8 * c8]);
        }
    }
}

// ---- transitions of config 5 on fp16 block buffers, pool-first in two steps (round 2).  The one-kernel form (the general
// conv1x1_f16_kernel<POOL>) gathers the four source rows of every pooled row with 8-B loads, without prefetch, once per
// 128-column tile of the output: 17.5 ms per 256-px array for the three transitions, ~1 TB/s.  Step 1 is this kernel:
// P[img, oy, ox][c] = mean over the 2 x 2 window of relu(scale[c] x + shift[c]) (norm -> relu -> pool, densenet.py:50-54 with
// the pool moved in front of the 1x1 conv: both are linear), 16-B loads and stores, fp32 arithmetic, one pass over the
// block buffer.  Step 2 is conv1x1_h16_kernel on P with no prologue and no consumer activation (a quarter of the rows).
__global__ __launch_bounds__(256) void bnrelu_avgpool2_h16_kernel(const _Float16* __restrict__ in, long ldi,
                                                                  _Float16* __restrict__ out, long ldo, long Mout, int C8,
                                                                  int S, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, long ibs = 32) {
    const long total = Mout * C8;
    const int So = S >> 1;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C8;
        const int c = 8 * (int)(idx - row * C8);
        const long img = row / ((long)So * So);
        const int rem = (int)(row - img * So * So);
        const int oy = rem / So, ox = rem - oy * So;
        const long src = ((img * S + 2 * oy) * S + 2 * ox) * ldi + (c >> 5) * ibs + (c & 31);   // ibs = 32: row-major input
        const float4 sc0 = ld4(scale + c), sc1 = ld4(scale + c + 4), sh0 = ld4(shift + c), sh1 = ld4(shift + c + 4);
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        half8 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ldh8(in + src + ((q >> 1) * (long)S + (q & 1)) * ldi);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = act1((float)v[0][j], sc[j], sh[j]);                    // the four in the order of the one-kernel form
            a += act1((float)v[1][j], sc[j], sh[j]);
            a += act1((float)v[2][j], sc[j], sh[j]);
            a += act1((float)v[3][j], sc[j], sh[j]);
            o[j] = (_Float16)(0.25f * a);
        }
        *reinterpret_cast<half8*>(out + row * ldo + c) = o;
    }
}

// ---- conv3x3 (pad 1): same contiguous-strip scheme as the fp32 kernel, fp16 LDS image
__global__ __launch_bounds__(256) void conv3x3_f16_kernel(const float* __restrict__ A, long lda,
                                                          const float* __restrict__ Wr, float* __restrict__ out, long ldc,
                                                          long M, int N, int K, int S, const float* __restrict__ scale,
                                                          const float* __restrict__ shift) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    const int strip = 128 + 2 * S + 2;
    _Float16* As = ldsh;                         // [strip + 1 zero row][LDH]
    _Float16* Bs = ldsh + (strip + 1) * LDH;     // [9][32][LDH]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const long P0 = (long)blockIdx.x * 128;
    const long base = P0 - S - 1;
    const int n0 = blockIdx.y * 32;
    const bool has_act = scale != nullptr;
    const int kq = t & 7, r0 = t >> 3;
    if (t < LDH) As[strip * LDH + t] = (_Float16)0.f;
    const long P = P0 + 32 * wave + i;
    int aoff[9];
    {
        unsigned mask = 0;
        if (P < M) {
            const int rem = (int)(P % ((long)S * S));
            const int y = rem / S, x = rem - y * S;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) mask |= 1u << tap;
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = (S + 1) + (tap / 3 - 1) * S + (tap % 3 - 1);
            aoff[tap] = ((mask >> tap) & 1u) ? (32 * wave + i + off) * LDH + 8 * h : strip * LDH;
        }
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int nload = n0 + r0 < N ? n0 + r0 : N - 1;

    for (int k0 = 0; k0 < K; k0 += 32) {
        const int k = k0 + 4 * kq;
        const bool kok = k < K;
        const int kc = kok ? k : 0;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act) { sc4 = ld4(scale + kc); sh4 = ld4(shift + kc); }
        __syncthreads();
        for (int row = r0; row < strip; row += 32) {
            const long Pr = base + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Pr >= 0 && Pr < M && kok) {
                v = ld4(A + Pr * lda + kc);
                if (has_act) {
                    v.x = act1(v.x, sc4.x, sh4.x); v.y = act1(v.y, sc4.y, sh4.y);
                    v.z = act1(v.z, sc4.z, sh4.z); v.w = act1(v.w, sc4.w, sh4.w);
                }
            }
            *reinterpret_cast<half4*>(&As[row * LDH + 4 * kq]) = to_h4(v);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float4 v = ld4(Wr + ((long)j * N + nload) * K + kc);
            if (!(n0 + r0 < N && kok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<half4*>(&Bs[(r0 + 32 * j) * LDH + 4 * kq]) = to_h4(v);
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const half8 a0 = ldh8(As + aoff[tap]);
            const half8 b0 = ldh8(&Bs[(tap * 32 + i) * LDH + 8 * h]);
            const half8 a1 = ldh8(As + aoff[tap] + 16);
            const half8 b1 = ldh8(&Bs[(tap * 32 + i) * LDH + 16 + 8 * h]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc1, 0, 0, 0);
        }
    }
    const int col = n0 + i;
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = P0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) out[row * ldc + col] = acc0[r] + acc1[r];
        }
    }
}

bool al16h(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

GNX_EXPORT int gnx_conv1x1_bnrelu_f16(const float* A, long lda, const float* W, float* out, long ldc, long M, int N,
                                      int K, const float* scale, const float* shift, int pool, int S_in,
                                      hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0 || lda < K || ldc < N || (!scale) != (!shift))
        return GNX_ERR_BAD_ARG;
    if (!(al16h(A) && al16h(W) && lda % 4 == 0 && K % 4 == 0 && (!scale || (al16h(scale) && al16h(shift)))))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    dim3 grid(gnx_cdiv(M, 128), gnx_cdiv(N, 128));
    if (pool) conv1x1_f16_kernel<true><<<grid, 256, 0, stream>>>(A, lda, W, out, ldc, M, N, K, scale, shift, S_in);
    else conv1x1_f16_kernel<false><<<grid, 256, 0, stream>>>(A, lda, W, out, ldc, M, N, K, scale, shift, S_in);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_conv3x3_bnrelu_f16(const float* A, long lda, const float* Wr, float* out, long ldc, long M, int N,
                                      int K, int S, const float* scale, const float* shift, hipStream_t stream) {
    if (!A || !Wr || !out || M < 0 || N <= 0 || K <= 0 || S <= 0 || lda < K || ldc < N || (!scale) != (!shift) ||
        (M % ((long)S * S)) != 0)
        return GNX_ERR_BAD_ARG;
    if (!(al16h(A) && al16h(Wr) && lda % 4 == 0 && K % 4 == 0 && (!scale || (al16h(scale) && al16h(shift)))))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    const size_t lds_bytes = ((size_t)(128 + 2 * S + 2 + 1) * LDH + 9 * 32 * LDH) * sizeof(_Float16);
    if (lds_bytes > 64 * 1024) return GNX_ERR_UNSUPPORTED;
    dim3 grid(gnx_cdiv(M, 128), gnx_cdiv(N, 32));
    conv3x3_f16_kernel<<<grid, 256, lds_bytes, stream>>>(A, lda, Wr, out, ldc, M, N, K, S, scale, shift);
    return gnx_launch_status();
}

// gnx_conv1x1_bnrelu_f16 (pool = 0) storing the bottleneck ACTIVATED and in fp16: out16[m][n] = fp16(relu(out_scale[n] *
// y[m][n] + out_shift[n])), ldc16 in halves - the operand of gnx_conv3x3_f16_dma.
GNX_EXPORT int gnx_conv1x1_bnrelu_f16_act16(const float* A, long lda, const float* W, void* out16, long ldc16, long M, int N,
                                            int K, const float* scale, const float* shift, const float* out_scale,
                                            const float* out_shift, hipStream_t stream) {
    if (!A || !W || !out16 || !out_scale || !out_shift || M < 0 || N <= 0 || K <= 0 || lda < K || ldc16 < N ||
        (!scale) != (!shift))
        return GNX_ERR_BAD_ARG;
    if (!(al16h(A) && al16h(W) && al16h(out16) && lda % 4 == 0 && K % 4 == 0 && N % 8 == 0 && ldc16 % 8 == 0 &&
          (!scale || (al16h(scale) && al16h(shift)))))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    dim3 grid(gnx_cdiv(M, 128), gnx_cdiv(N, 128));
    conv1x1_f16_kernel<false, true><<<grid, 256, 0, stream>>>(A, lda, W, reinterpret_cast<float*>(out16), ldc16, M, N, K,
                                                              scale, shift, 0, out_scale, out_shift);
    return gnx_launch_status();
}

// Config 5 with fp16 block buffers: A16 [M][K] halves (lda16), out16 [M][N] halves (ldc16).  out_scale / out_shift may both be
// NULL (transition: no consumer activation at the store); pool as gnx_conv1x1_bnrelu.
GNX_EXPORT int gnx_conv1x1_bnrelu_f16_h(const void* A16, long lda16, const float* W, void* out16, long ldc16, long M, int N,
                                        int K, const float* scale, const float* shift, const float* out_scale,
                                        const float* out_shift, int pool, int S_in, hipStream_t stream) {
    if (!A16 || !W || !out16 || M < 0 || N <= 0 || K <= 0 || lda16 < K || ldc16 < N || (!scale) != (!shift) ||
        (!out_scale) != (!out_shift) || (pool && S_in < 2))
        return GNX_ERR_BAD_ARG;
    if (!(al16h(W) && al16h(out16) && (reinterpret_cast<uintptr_t>(A16) & 7) == 0 && lda16 % 4 == 0 && K % 4 == 0 &&
          N % 8 == 0 && ldc16 % 8 == 0 && (!scale || (al16h(scale) && al16h(shift))) && (!pool || S_in % 2 == 0)))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    dim3 grid(gnx_cdiv(M, 128), gnx_cdiv(N, 128));
    const float* A = reinterpret_cast<const float*>(A16);
    float* out = reinterpret_cast<float*>(out16);
    if (pool)
        conv1x1_f16_kernel<true, true, true><<<grid, 256, 0, stream>>>(A, lda16, W, out, ldc16, M, N, K, scale, shift, S_in,
                                                                       out_scale, out_shift);
    else
        conv1x1_f16_kernel<false, true, true><<<grid, 256, 0, stream>>>(A, lda16, W, out, ldc16, M, N, K, scale, shift, 0,
                                                                        out_scale, out_shift);
    return gnx_launch_status();
}

// The 2 x 2 mean of the activated block buffer (step 1 of the two-step transition of config 5):
// out16[img, oy, ox][c] = fp16(mean_{2x2} relu(scale[c] in16[img, 2oy + dy, 2ox + dx][c] + shift[c])).  8 | C, even S,
// 16-B aligned rows.
GNX_EXPORT int gnx_bnrelu_avgpool2_h16(const void* in16, long ldi, void* out16, long ldo, long imgs, int C, int S,
                                       const float* scale, const float* shift, hipStream_t stream) {
    if (!in16 || !out16 || !scale || !shift || imgs < 0 || C <= 0 || S < 2 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (C % 8 != 0 || S % 2 != 0 || ldi % 8 != 0 || ldo % 8 != 0 || !al16h(in16) || !al16h(out16) || !al16h(scale) ||
        !al16h(shift))
        return GNX_ERR_UNSUPPORTED;
    const long Mout = imgs * (S / 2) * (S / 2);
    if (Mout == 0) return GNX_OK;
    long blocks = (Mout * (C / 8) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    bnrelu_avgpool2_h16_kernel<<<(int)blocks, 256, 0, stream>>>(reinterpret_cast<const _Float16*>(in16), ldi,
                                                                reinterpret_cast<_Float16*>(out16), ldo, Mout, C / 8, S,
                                                                scale, shift);
    return gnx_launch_status();
}
// the same reading the channel-blocked buffer [C / 32][rows_total][32] halves of the fused dense layers (32 | C)
GNX_EXPORT int gnx_bnrelu_avgpool2_h16_cb(const void* in16, long rows_total, void* out16, long ldo, long imgs, int C, int S,
                                          const float* scale, const float* shift, hipStream_t stream) {
    if (!in16 || !out16 || !scale || !shift || imgs < 0 || C <= 0 || S < 2 || rows_total < imgs * S * S || ldo < C)
        return GNX_ERR_BAD_ARG;
    if (C % 32 != 0 || S % 2 != 0 || ldo % 8 != 0 || !al16h(in16) || !al16h(out16) || !al16h(scale) || !al16h(shift))
        return GNX_ERR_UNSUPPORTED;
    const long Mout = imgs * (S / 2) * (S / 2);
    if (Mout == 0) return GNX_OK;
    long blocks = (Mout * (C / 8) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    bnrelu_avgpool2_h16_kernel<<<(int)blocks, 256, 0, stream>>>(reinterpret_cast<const _Float16*>(in16), 32,
                                                                reinterpret_cast<_Float16*>(out16), ldo, Mout, C / 8, S,
                                                                scale, shift, rows_total * 32);
    return gnx_launch_status();
}

// Dense-layer conv1 of config 5 on fp16 block buffers with fp16 weights (W16 = the weight rounded once, [N][K] halves):
// gnx_conv1x1_bnrelu_f16_h (pool = 0) in chunks of 64 channels and 16-B loads.  32 | K, 8 | N.  scale / shift NULL: no
// prologue (the operand is used as it is); out_scale / out_shift NULL: no consumer activation at the store - together the
// second step of a transition (gnx_bnrelu_avgpool2_h16 first).
static int conv1x1_h16_launch(const void* A16, long lda16, const void* W16, void* out16, long ldc16, long obs, long M, int N,
                              int K, const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                              hipStream_t stream) {
    if (!A16 || !W16 || !out16 || (!scale) != (!shift) || (!out_scale) != (!out_shift) || M < 0 || N <= 0 || K <= 0 ||
        lda16 < K || (obs == 32 ? ldc16 < N : (ldc16 != 32 || N % 32 != 0)))
        return GNX_ERR_BAD_ARG;
    if (!(al16h(A16) && al16h(W16) && al16h(out16) && al16h(scale) && al16h(shift) && lda16 % 8 == 0 && K % 8 == 0 &&
          N % 8 == 0 && ldc16 % 8 == 0))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    // 256-row tiles (a weight chunk staged per 256 rows instead of 128) once they still fill the chip twice over
    if (M >= 256L * 512) {       // (round 4, per-layer table of the trained config-5 step: +0 ... 9 % also at K = 64 ... 160)
        dim3 grid(gnx_cdiv(M, 256), gnx_cdiv(N, 128));
        conv1x1_h16_m256_kernel<<<grid, 256, 0, stream>>>(reinterpret_cast<const _Float16*>(A16), lda16,
                                                          reinterpret_cast<const _Float16*>(W16),
                                                          reinterpret_cast<_Float16*>(out16), ldc16, M, N, K, scale, shift,
                                                          out_scale, out_shift, obs);
        return gnx_launch_status();
    }
    dim3 grid(gnx_cdiv(M, 128), gnx_cdiv(N, 128));
    conv1x1_h16_kernel<<<grid, 256, 0, stream>>>(reinterpret_cast<const _Float16*>(A16), lda16,
                                                 reinterpret_cast<const _Float16*>(W16), reinterpret_cast<_Float16*>(out16),
                                                 ldc16, M, N, K, scale, shift, out_scale, out_shift, obs);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_conv1x1_bnrelu_h16(const void* A16, long lda16, const void* W16, void* out16, long ldc16, long M, int N,
                                      int K, const float* scale, const float* shift, const float* out_scale,
                                      const float* out_shift, hipStream_t stream) {
    return conv1x1_h16_launch(A16, lda16, W16, out16, ldc16, 32, M, N, K, scale, shift, out_scale, out_shift, stream);
}
// the same storing into the channel-blocked buffer [.. / 32][rows_total][32] halves of the fused dense layers: the N output
// channels become its first N / 32 blocks (a transition's output = the next dense block's first channels); 32 | N
GNX_EXPORT int gnx_conv1x1_bnrelu_h16_cb(const void* A16, long lda16, const void* W16, void* out16, long rows_total, long M,
                                         int N, int K, const float* scale, const float* shift, const float* out_scale,
                                         const float* out_shift, hipStream_t stream) {
    if (rows_total < M) return GNX_ERR_BAD_ARG;
    return conv1x1_h16_launch(A16, lda16, W16, out16, 32, rows_total * 32, M, N, K, scale, shift, out_scale, out_shift, stream);
}
