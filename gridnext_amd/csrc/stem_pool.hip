// DenseNet-BC forward: stem (conv0, optionally fused with norm0/relu0/pool0), BN+ReLU+maxpool, BN+ReLU+global average
// pool (densenet.py:98-112, :153-156).
#include <type_traits>
#include "fwd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ uint8 patches
// The reference turns 8-bit patch files into floats on the HOST (torchvision ToTensor: u8 / 255, image_datasets.py:102-105,
// optionally Normalize: (v - mean) / std) and ships 4 bytes per pixel over PCIe and through HBM.  Here patches stay uint8
// up to the stem kernel's operand load.  Bit-for-bit the same floats: u / 255 as one reciprocal multiply plus one
// Newton step on fused multiply-adds, which is the correctly rounded quotient for every u in 0..255 (checked exhaustively
// against torch's division, tests/test_gpu_kernels.py; the bare multiply is off by one ulp for 126 of the 256 values);
// Normalize's division the same way with 1 / std.  nrm = {mean[3], std[3], 1/std[3]} or NULL.
__device__ __forceinline__ float u8_unit(float u) {
    constexpr float R255 = 1.0f / 255.0f;
    const float q = u * R255;
    const float e = fmaf(-255.0f, q, u);
    return fmaf(e, R255, q);
}
__device__ __forceinline__ float u8_pixel(float u, bool norm, float mean, float sd, float rsd) {
    float v = u8_unit(u);
    if (norm) {
        const float t = v - mean;
        const float q = t * rsd;
        const float e = fmaf(-sd, q, t);
        v = fmaf(e, rsd, q);
    }
    return v;
}
// 4 consecutive pixels of one channel plane: a dword of bytes -> 4 floats
__device__ __forceinline__ float4 u8x4_pixels(uint32_t wv, bool norm, float mean, float sd, float rsd) {
    return make_float4(u8_pixel((float)(wv & 0xffu), norm, mean, sd, rsd),
                       u8_pixel((float)((wv >> 8) & 0xffu), norm, mean, sd, rsd),
                       u8_pixel((float)((wv >> 16) & 0xffu), norm, mean, sd, rsd),
                       u8_pixel((float)(wv >> 24), norm, mean, sd, rsd));
}

// out[img][c][y][x] (float) = ToTensor (+ Normalize) of x8[img][c][y][x]: 4 pixels per thread (HW % 4 == 0)
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ x8, float* __restrict__ out, long n4,
                                                        long hw4, int C, const float* __restrict__ nrm) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c = (int)((i / hw4) % C);
        const bool norm = nrm != nullptr && c < 3;
        const float mean = norm ? nrm[c] : 0.f, sd = norm ? nrm[3 + c] : 1.f, rsd = norm ? nrm[6 + c] : 1.f;
        const uint32_t wv = reinterpret_cast<const uint32_t*>(x8)[i];
        reinterpret_cast<float4*>(out)[i] = u8x4_pixels(wv, norm, mean, sd, rsd);
    }
}

// ------------------------------------------------------------------------------------------------ stem conv (NCHW in, NHWC out)
// out[(img,oy,ox)][o] = sum_{c,ky,kx} x[img][c][oy*st+ky-pad][ox*st+kx-pad] * w[o][c][ky][kx]
// im2col is built in LDS one input channel at a time: K-chunk = KH x 8 (kx padded to 8 with zero weights).
constexpr int ST_BM = 128, ST_BN = 64;

__global__ __launch_bounds__(256) void conv_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ out, long ldc, long M, int Cin, int H,
                                                        int Wd, int Ho, int Wo, int O, int KH, int KW, int stride,
                                                        int pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KC = KH * 8, LD = KC + 4;
    float* As = lds;                 // [128][LD]
    float* Bs = lds + ST_BM * LD;    // [64][LD]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)blockIdx.x * ST_BM;
    const int n0 = blockIdx.y * ST_BN;
    const int kx = t & 7, rr = t >> 3;

    long ibase[4];
    int iy0[4], ix0[4];
    bool rok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long row = m0 + rr + 32 * p;
        rok[p] = row < M;
        const long img = row / ((long)Ho * Wo);
        const int rem = (int)(row - img * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        ibase[p] = img * Cin * (long)H * Wd;
        iy0[p] = oy * stride - pad;
        ix0[p] = ox * stride - pad + kx;
    }
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (int c = 0; c < Cin; ++c) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool xok = rok[p] && kx < KW && ix0[p] >= 0 && ix0[p] < Wd;
            const float* xc = x + ibase[p] + (long)c * H * Wd;
            for (int ky = 0; ky < KH; ++ky) {
                const int iy = iy0[p] + ky;
                float v = 0.f;
                if (xok && iy >= 0 && iy < H) v = xc[(long)iy * Wd + ix0[p]];
                As[(rr + 32 * p) * LD + ky * 8 + kx] = v;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int o = n0 + rr + 32 * p;
            for (int ky = 0; ky < KH; ++ky) {
                float v = 0.f;
                if (o < O && kx < KW) v = w[(((long)o * Cin + c) * KH + ky) * KW + kx];
                Bs[(rr + 32 * p) * LD + ky * 8 + kx] = v;
            }
        }
        __syncthreads();
        for (int s = 0; s < KH; ++s) {
            const float4 a0 = ld4(&As[(64 * wm + i) * LD + 8 * s + 4 * h]);
            const float4 a1 = ld4(&As[(64 * wm + 32 + i) * LD + 8 * s + 4 * h]);
            const float4 b = ld4(&Bs[(32 * wn + i) * LD + 8 * s + 4 * h]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc[1], 0, 0, 0);
        }
    }
    const int col = n0 + 32 * wn + i;
    if (col < O) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) out[row * ldc + col] = acc[mt][r];
            }
    }
}

// Patch-resident form of the stem conv (used for the two stems DenseNet has: 7x7 s2 p3 and 3x3 s1 p1, 3 input channels).
// A workgroup owns an 8x16 tile of output positions: the input patch it needs (all channels, zero-padded borders) is
// staged into LDS once with coalesced loads, and the MFMA A-fragments are read STRAIGHT from that patch - the four
// consecutive kx of a fragment are four consecutive floats of a patch row - so no im2col image is ever built.  The whole
// weight tensor ([O<=64][Cin][KH][8], kx zero-padded to 8) lives in LDS for the lifetime of the (persistent) workgroup.
template <int STRIDE, int KH, int CIN>
__global__ __launch_bounds__(256) void conv_stem_patch_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ out, long ldc, int Cin, int H, int Wd,
                                                              int Ho, int Wo, int O, int KW, int pad, int tiles_x,
                                                              int tiles_y, long ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PH = 7 * STRIDE + KH, PW = (15 * STRIDE + 8 + 1) & ~1;
    const int KT = Cin * KH * 8, LDB = KT + 4;
    float* Bs = lds;                          // [64][LDB]
    float* Ps = lds + 64 * LDB;               // [Cin][PH][PW]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    for (int idx = t; idx < 64 * KT; idx += 256) {
        const int n = idx / KT, rem = idx - n * KT;
        const int kx = rem & 7, cky = rem >> 3;          // cky = c*KH + ky
        float v = 0.f;
        if (n < O && kx < KW) v = w[((long)n * Cin * KH + cky) * KW + kx];
        Bs[n * LDB + rem] = v;
    }
    const int trow = 32 * wave + i;                       // this lane's row of the 128-position tile
    const int oyl = trow >> 4, oxl = trow & 15;
    const float* pa = Ps + (STRIDE * oyl) * PW + STRIDE * oxl + 4 * h;
    const float* pb = Bs + i * LDB + 4 * h;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long img = tile / ((long)tiles_x * tiles_y);
        const int trem = (int)(tile - img * tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * 8, ox0 = tx * 16;
        const int iy0 = oy0 * STRIDE - pad, ix0 = ox0 * STRIDE - pad;
        __syncthreads();                                  // previous tile's fragment reads (and the Bs fill) are done
        for (int idx = t; idx < Cin * PH * PW; idx += 256) {
            const int px = idx % PW, py = (idx / PW) % PH, c = idx / (PW * PH);
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < Wd) v = x[((img * Cin + c) * H + iy) * (long)Wd + ix];
            Ps[idx] = v;
        }
        __syncthreads();
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        auto load_a = [&](int step) {
            const float* ap = pa + ((step / KH) * PH + (step % KH)) * PW;
            if (STRIDE % 2 == 0) {
                const float2 lo = *reinterpret_cast<const float2*>(ap);
                const float2 hi = *reinterpret_cast<const float2*>(ap + 2);
                return make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            return make_float4(ap[0], ap[1], ap[2], ap[3]);
        };
        constexpr int NSTEP = CIN * KH;
        float4 a = load_a(0), b0 = ld4(pb), b1 = ld4(pb + 32 * LDB);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            float4 na = a, nb0 = b0, nb1 = b1;
            if (step + 1 < NSTEP) {
                na = load_a(step + 1);
                nb0 = ld4(pb + (step + 1) * 8);
                nb1 = ld4(pb + 32 * LDB + (step + 1) * 8);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            a = na; b0 = nb0; b1 = nb1;
            __builtin_amdgcn_sched_group_barrier(0x100, STRIDE % 2 == 0 ? 4 : 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = oy0 + (rr >> 4), ox = ox0 + (rr & 15);
            if (oy < Ho && ox < Wo) {
                float* o = out + ((img * Ho + oy) * (long)Wo + ox) * ldc;
                if (i < O) o[i] = acc0[r];
                if (32 + i < O) o[32 + i] = acc1[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ stem + BN + ReLU + maxpool
// features.conv0 -> norm0 -> relu0 -> pool0 (densenet.py:105-110) in ONE kernel for the 128-px geometry (7x7 s2 p3 conv to a
// 64-wide map, 3x3 s2 p1 max pool): the 5.2 GB conv0 output of an array never goes to HBM (the two-kernel path writes
// it and reads it back: 4.9 + 1.4 ms).  A persistent workgroup sweeps an image top to bottom in tiles of 2 conv rows x 64
// columns (= the 128 positions of the MFMA M dimension, full width: no horizontal halo); tile t yields conv rows 2t and
// 2t+1, pooled row t = max over conv rows {2t-1, 2t, 2t+1} and columns {2px-1, 2px, 2px+1}.  Row 2t-1 is the previous
// tile's second row: every thread owns the same (px, 4 channels) items in every tile, so that carry lives in registers.
// relu >= 0, so "outside the map" contributes 0 exactly as in gnx_bnrelu_maxpool.
constexpr int SP_LDT = 40;         // floats per position of the activated 32-channel half tile in LDS: 4 * 40 = 32 (mod 64)
constexpr int SP_LDTH = 72;        // the fp16 stem's tile: halves per position (36 dwords: lane halves 4 positions apart land 16 banks apart)
                                   // banks, so the two lane halves of an accumulator store (positions p, p + 4) hit disjoint banks
// WO = width of the conv0 map: 64 (128-px patches: tiles of 2 rows, one pooled row each, one carried row) or 128
// (256-px patches: tiles of 1 row; a pooled row is emitted on every odd conv row from the two carried rows and the new one).
// H16: the pooled output is stored as fp16 ([rows][ldo halves]): config 5 with fp16 block buffers.
// U8: the patches are uint8 [imgs][3][H][W]; ToTensor (+ Normalize, nrm != NULL) happens between the load and the LDS
// stash (a quarter of the input bytes: 245 MB instead of 981 MB per 128-px array).
// IDX: also record, per pooled element, WHICH window element won (0..8 row-major in the 3 x 3 window, the first maximal one
// of the scan - torch's rule, as gnx_bnrelu_maxpool_argmax records it) in amax [pooled rows][O] bytes: the f-trained step
// then needs neither the conv0 map nor a second pass for its pool0 / norm0 adjoint.
template <int WO, bool H16 = false, bool U8 = false, bool IDX = false>
__global__ __launch_bounds__(256) void conv_stem_pool_kernel(const void* __restrict__ xv, const float* __restrict__ w,
                                                             float* __restrict__ out, long ldo, int H, int Wd, int O,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, long imgs,
                                                             const float* __restrict__ nrm,
                                                             unsigned char* __restrict__ amax = nullptr) {
    const float* __restrict__ x = reinterpret_cast<const float*>(xv);
    const uint8_t* __restrict__ x8 = reinterpret_cast<const uint8_t*>(xv);
    constexpr int CIN = 3, KH = 7, KW = 7, STRIDE = 2, PAD = 3;
    constexpr int RT = 128 / WO;                          // conv rows per tile
    constexpr int PH = STRIDE * (RT - 1) + KH, PW = ((WO - 1) * STRIDE + 8 + 1 + 1) & ~1;      // input patch per channel
    constexpr int F4R = WO * STRIDE / 4;                  // 16-B pieces per input row (row width 2 WO)
    constexpr int NPC = CIN * PH * F4R, NPRE = (NPC + 255) / 256;       // patch pieces, per thread
    constexpr int NIT = (WO / 2) * 8 / 256;               // pooling items per thread and pass: (pooled x, 4 of 32 channels)
    constexpr int KT = CIN * KH * 8, LDB = KT + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto store4 = [](float* o, long off, const float4& v) {     // 4 channels of a pooled position at element offset off
        if constexpr (H16) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            const half4 hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(o) + off) = hv;
        } else {
            *reinterpret_cast<float4*>(o + off) = v;
        }
    };
    float* Bs = lds;                          // [64][LDB] weights, kx zero-padded to 8
    float* Ps = lds + 64 * LDB;               // [CIN][PH][PW]
    float* Ts = Ps + CIN * PH * PW;           // [128 positions][SP_LDT] activated conv tile
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    for (int idx = t; idx < 64 * KT; idx += 256) {
        const int n = idx / KT, rem = idx - n * KT;
        const int kx = rem & 7, cky = rem >> 3;          // cky = c*KH + ky
        float v = 0.f;
        if (n < O && kx < KW) v = w[((long)n * CIN * KH + cky) * KW + kx];
        Bs[n * LDB + rem] = v;
    }
    for (int idx = t; idx < CIN * PH * PW; idx += 256) Ps[idx] = 0.f;          // the pad columns stay zero for good
    const int trow = 32 * wave + i;                       // this lane's position in the RT x WO tile
    const float* pa = Ps + (STRIDE * (trow / WO)) * PW + STRIDE * (trow % WO) + 4 * h;
    const float* pb = Bs + i * LDB + 4 * h;
    const float sc0 = i < O ? scale[i] : 0.f, sh0 = i < O ? shift[i] : 0.f;
    const float sc1 = 32 + i < O ? scale[32 + i] : 0.f, sh1 = 32 + i < O ? shift[32 + i] : 0.f;
    const int c4 = t & 7, pxa = t >> 3;                   // pooling items of this thread: (pxa + 32 q, c4) per pass
    const int Ho2 = (H + 2 * PAD - KH) / STRIDE + 1;      // conv rows (even), pooled rows Ho2 / 2
    const int ntt = Ho2 / RT;                             // tiles per image

    // Patch staging: CIN * PH input rows of 2 WO floats as 16-B pieces, fetched one tile AHEAD into registers while the
    // current tile multiplies.
    float4 pre[U8 ? 1 : NPRE];
    uint32_t pre8[U8 ? NPRE : 1];             // U8: a piece is the same 4 pixels, as one dword of bytes
    bool in8[U8 ? NPRE : 1];                  // ... and whether it lies inside the image (outside stays 0.0f, not "pixel 0":
                                              // the conv pads the NORMALISED input with zeros)
    auto fetch_patch = [&](long img, int tt) {
        const int iy0 = RT * tt * STRIDE - PAD;
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int j = t + 256 * q;
            const int f4 = j % F4R, py = (j / F4R) % PH, c = (j / F4R) / PH;
            const int iy = iy0 + py;
            const bool inside = j < NPC && iy >= 0 && iy < H;
            if constexpr (U8) {
                pre8[q] = 0u;
                in8[q] = inside;
                if (inside)
                    pre8[q] = *reinterpret_cast<const uint32_t*>(x8 + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
            } else {
                pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inside) pre[q] = ld4(x + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
            }
        }
    };
    auto stash_patch = [&]() {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int j = t + 256 * q;
            if (j < NPC) {
                float* d = Ps + (j / F4R) * PW + PAD + 4 * (j % F4R);        // patch column = ix + PAD
                if constexpr (U8) {
                    const int c = (j / F4R) / PH;
                    const bool norm = nrm != nullptr;
                    const float4 v = in8[q] ? u8x4_pixels(pre8[q], norm, norm ? nrm[c] : 0.f, norm ? nrm[3 + c] : 1.f,
                                                          norm ? nrm[6 + c] : 1.f)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                } else {
                    d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
                }
            }
        }
    };
    __syncthreads();                                      // Bs and the zeroed patch are in place
    fetch_patch(blockIdx.x, 0);

    for (long img = blockIdx.x; img < imgs; img += gridDim.x) {
        float4 carry[2][NIT], carry2[2][NIT];             // horizontal maxima of the last (and, WO = 128, second-last) row
        unsigned cidx[2][NIT], cidx2[2][NIT];             // IDX: their column (0..2) per channel, one byte each
        const float c0v = IDX ? -1.f : 0.f;               // IDX: "no such row" must lose against every real value (>= 0)
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            carry[0][q] = carry[1][q] = carry2[0][q] = carry2[1][q] = make_float4(c0v, c0v, c0v, c0v);
            cidx[0][q] = cidx[1][q] = cidx2[0][q] = cidx2[1][q] = 0u;
        }
        for (int tt = 0; tt < ntt; ++tt) {
            __syncthreads();                              // previous tile's fragment and Ts reads are done
            stash_patch();
            {
                long nimg = img;
                int nt2 = tt + 1;
                if (nt2 == ntt) { nt2 = 0; nimg += gridDim.x; }
                if (nimg >= imgs) nimg = blockIdx.x;       // past the end: a harmless re-read
                fetch_patch(nimg, nt2);
            }
            asm volatile("" ::: "memory");                // keep the prefetch in front of the multiply
            __syncthreads();
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            auto load_a = [&](int step) {
                const float* ap = pa + ((step / KH) * PH + (step % KH)) * PW;
                const float2 lo = *reinterpret_cast<const float2*>(ap);
                const float2 hi = *reinterpret_cast<const float2*>(ap + 2);
                return make_float4(lo.x, lo.y, hi.x, hi.y);
            };
            constexpr int NSTEP = CIN * KH;
            float4 a = load_a(0), b0 = ld4(pb), b1 = ld4(pb + 32 * LDB);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                float4 na = a, nb0 = b0, nb1 = b1;
                if (step + 1 < NSTEP) {
                    na = load_a(step + 1);
                    nb0 = ld4(pb + (step + 1) * 8);
                    nb1 = ld4(pb + 32 * LDB + (step + 1) * 8);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                a = na; b0 = nb0; b1 = nb1;
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            // norm0 + relu0 on the accumulators, tile to LDS (lane = channel, register = position) and pool0 - in two
            // passes of 32 channels (acc0, then acc1), so that the activated tile takes 20 KB instead of 37 and TWO
            // workgroups fit a CU: one's patch stash, pooling and barriers then hide behind the other's MFMAs
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) __syncthreads();           // pass 0's pooling reads are done
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Ts[rr * SP_LDT + i] = pass == 0 ? fmaxf(fmaf(acc0[r], sc0, sh0), 0.f) : fmaxf(fmaf(acc1[r], sc1, sh1), 0.f);
                }
                __syncthreads();
                const int ch = 32 * pass + 4 * c4;        // first of this item's 4 output channels
#pragma unroll
                for (int q = 0; q < NIT; ++q) {
                    const int px = pxa + 32 * q;
                    float4 m0 = make_float4(c0v, c0v, c0v, c0v), m1 = m0;        // horizontal maxima of the tile's rows
                    unsigned h0 = 0u, h1 = 0u;                                   // IDX: their columns, a byte per channel
                    auto hmax = [](float4& m, unsigned& hx, const float4& v, unsigned col) {
                        if (IDX) {                                               // strict >: the first maximum of the scan wins
                            if (v.x > m.x) { m.x = v.x; hx = (hx & 0xffffff00u) | col; }
                            if (v.y > m.y) { m.y = v.y; hx = (hx & 0xffff00ffu) | (col << 8); }
                            if (v.z > m.z) { m.z = v.z; hx = (hx & 0xff00ffffu) | (col << 16); }
                            if (v.w > m.w) { m.w = v.w; hx = (hx & 0x00ffffffu) | (col << 24); }
                        } else {
                            m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
                        }
                    };
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int ox = 2 * px + dx;
                        if (ox < 0) continue;                                    // ox <= WO - 1 always
                        hmax(m0, h0, ld4(&Ts[ox * SP_LDT + 4 * c4]), (unsigned)(dx + 1));
                        if (RT == 2) hmax(m1, h1, ld4(&Ts[(WO + ox) * SP_LDT + 4 * c4]), (unsigned)(dx + 1));
                    }
                    // vertical: rows in scan order (ra, rb, rc) with their columns; window index = 3 row + column
                    auto vmax = [](const float4& ra, unsigned ia, const float4& rb, unsigned ib, const float4& rc, unsigned ic,
                                   float4& o4, unsigned& oi) {
                        if (IDX) {
                            o4 = ra; oi = ia;                                                      // row 0: index = column
                            const unsigned jb = ib + 0x03030303u, jc = ic + 0x06060606u;
                            if (rb.x > o4.x) { o4.x = rb.x; oi = (oi & 0xffffff00u) | (jb & 0x000000ffu); }
                            if (rb.y > o4.y) { o4.y = rb.y; oi = (oi & 0xffff00ffu) | (jb & 0x0000ff00u); }
                            if (rb.z > o4.z) { o4.z = rb.z; oi = (oi & 0xff00ffffu) | (jb & 0x00ff0000u); }
                            if (rb.w > o4.w) { o4.w = rb.w; oi = (oi & 0x00ffffffu) | (jb & 0xff000000u); }
                            if (rc.x > o4.x) { o4.x = rc.x; oi = (oi & 0xffffff00u) | (jc & 0x000000ffu); }
                            if (rc.y > o4.y) { o4.y = rc.y; oi = (oi & 0xffff00ffu) | (jc & 0x0000ff00u); }
                            if (rc.z > o4.z) { o4.z = rc.z; oi = (oi & 0xff00ffffu) | (jc & 0x00ff0000u); }
                            if (rc.w > o4.w) { o4.w = rc.w; oi = (oi & 0x00ffffffu) | (jc & 0xff000000u); }
                        } else {
                            o4 = make_float4(fmaxf(fmaxf(ra.x, rb.x), rc.x), fmaxf(fmaxf(ra.y, rb.y), rc.y),
                                             fmaxf(fmaxf(ra.z, rb.z), rc.z), fmaxf(fmaxf(ra.w, rb.w), rc.w));
                        }
                    };
                    if (RT == 2) {                        // rows 2tt, 2tt+1 + the carried row 2tt-1 -> pooled row tt
                        float4 o4;
                        unsigned oi = 0u;
                        vmax(carry[pass][q], cidx[pass][q], m0, h0, m1, h1, o4, oi);
                        carry[pass][q] = m1;
                        cidx[pass][q] = h1;
                        if (ch < O) {
                            const long prow = (img * (Ho2 / 2) + tt) * (long)(WO / 2) + px;
                            store4(out, prow * ldo + ch, o4);
                            if (IDX) *reinterpret_cast<unsigned*>(amax + prow * O + ch) = oi;
                        }
                    } else {                              // one row per tile: emit pooled row (tt-1)/2 on odd rows
                        if (tt & 1) {
                            float4 o4;
                            unsigned oi = 0u;
                            vmax(carry2[pass][q], cidx2[pass][q], carry[pass][q], cidx[pass][q], m0, h0, o4, oi);
                            if (ch < O) {
                                const long prow = (img * (Ho2 / 2) + (tt >> 1)) * (long)(WO / 2) + px;
                                store4(out, prow * ldo + ch, o4);
                                if (IDX) *reinterpret_cast<unsigned*>(amax + prow * O + ch) = oi;
                            }
                        }
                        carry2[pass][q] = carry[pass][q];
                        cidx2[pass][q] = cidx[pass][q];
                        carry[pass][q] = m0;
                        cidx[pass][q] = h0;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the same stem, fp16 MFMA
// Config 5 (fp16 MFMA conv path): the kernel above multiplies in fp32 - 18 ms of a 112-ms array at 256 px, 15x the matrix
// time of the fp16 form.  Here the patch and the weights are staged as halves and multiplied with v_mfma_f32_32x32x16_f16
// (fp32 accumulate): k = 8 (c * 7 + ky) + kx, one instruction covers two (c, ky) groups - lane half h takes group 2 s + h,
// its 8 consecutive kx are 8 consecutive halves of a patch row.  Those start at column 2 ox (4-B aligned only), so the A
// fragment is four ds_read_b32 (conflict-free: consecutive lanes, consecutive dwords); the weights [64][176 + 8] halves are
// 16-B aligned (ds_read_b128).  The 22nd group does not exist: its weights are zero and its A fragment re-reads group 20
// (finite values).  Epilogue (norm0 + relu0 on the accumulators, pool0 through the LDS, fp16 store) as the fp32 kernel.
// U8: uint8 patches (ToTensor / Normalize at the stash, as above).
typedef _Float16 sp_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 sp_half4 __attribute__((ext_vector_type(4)));
template <int WO, bool U8>
__global__ __launch_bounds__(256) void conv_stem_pool_f16_kernel(const void* __restrict__ xv, const float* __restrict__ w,
                                                                 _Float16* __restrict__ out, long ldo, int H, int Wd, int O,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, long imgs,
                                                                 const float* __restrict__ nrm, long obs = 32) {
    const float* __restrict__ x = reinterpret_cast<const float*>(xv);
    const uint8_t* __restrict__ x8 = reinterpret_cast<const uint8_t*>(xv);
    constexpr int CIN = 3, KH = 7, KW = 7, STRIDE = 2, PAD = 3;
    constexpr int RT = 128 / WO;                          // conv rows per tile
    // The input rows live in a RING of PH slots per channel (row iy in slot (iy + PAD) mod PH): a tile's conv rows read the
    // 2 RT + 5 rows iy = 2 RT tt - 3 ... 2 RT tt + 2 RT + 1, of which only the last NEW = 2 RT are new - the tile stages those
    // (the full window per tile was 3.5x / 2.25x the loads, conversions and LDS writes: 2.4 of the 5.0 ms of a 256-px array).
    // An image starts with NV stage-only steps (tt < 0) that bring in rows -3 ... 1 (zeros outside the image).
    constexpr int PH = 2 * RT + 5, NEW = STRIDE * RT, NV = RT == 1 ? 3 : 2;    // (exactly the window: 52 KB of LDS, 3 per CU)
    constexpr int PW = (WO - 1) * STRIDE + 8;             // halves per patch row: columns 2 ox ... 2 ox + 7 (4-B aligned rows)
    constexpr int F4R = WO * STRIDE / 4;                  // 4-pixel pieces per input row (row width 2 WO)
    constexpr int NPC = CIN * NEW * F4R, NPRE = (NPC + 255) / 256;
    constexpr int NIT = (WO / 2) * 16 / 256;             // pooled (position, 4-channel group) items per thread: all 64 channels at once
    constexpr int NG = CIN * KH, NSTEP = (NG + 1) / 2;    // 21 (c, ky) groups of 8 kx, two per MFMA
    constexpr int LDBH = NSTEP * 16 + 8;                  // halves per weight row (184: 23 sixteen-byte slots)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16* Bs = reinterpret_cast<_Float16*>(lds);      // [64][LDBH]
    _Float16* Ps = Bs + 64 * LDBH;                        // [CIN][PH][PW]
    // The activated conv tile is kept as fp16: rounding is monotonic, so the maximum of the rounded values IS the rounded
    // maximum the fp32 tile gave - bit-identical output, half the tile (20 -> 9 KB: three workgroups per CU instead of two)
    _Float16* Ts = Ps + ((CIN * PH * PW + 7) & ~7);       // [128 positions][SP_LDTH], 16-B aligned
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    for (int idx = t; idx < 64 * LDBH; idx += 256) {
        const int n = idx / LDBH, rem = idx - n * LDBH;
        const int kx = rem & 7, cky = rem >> 3;          // cky = c*KH + ky
        float v = 0.f;
        if (n < O && kx < KW && cky < NG) v = w[((long)n * CIN * KH + cky) * KW + kx];
        Bs[idx] = (_Float16)v;
    }
    for (int idx = t; idx < CIN * PH * PW; idx += 256) Ps[idx] = (_Float16)0.f;     // the pad columns stay zero for good
    const int trow = 32 * wave + i;                       // this lane's position in the RT x WO tile
    const _Float16* pa = Ps + STRIDE * (trow % WO);
    const int rr2 = STRIDE * (trow / WO);                 // (wave-uniform: 32 | WO)
    const _Float16* pb = Bs + i * LDBH + 8 * h;
    const float sc0 = i < O ? scale[i] : 0.f, sh0 = i < O ? shift[i] : 0.f;
    const float sc1 = 32 + i < O ? scale[32 + i] : 0.f, sh1 = 32 + i < O ? shift[32 + i] : 0.f;
    const int c4 = t & 15, pxa = t >> 4;
    const int Ho2 = (H + 2 * PAD - KH) / STRIDE + 1;
    const int ntt = Ho2 / RT;

    float4 pre[U8 ? 1 : NPRE];
    uint32_t pre8[U8 ? NPRE : 1];
    bool in8[U8 ? NPRE : 1];
    auto fetch_patch = [&](long img, int tt) {
        const int iy0 = NEW * tt + 2;                     // the tile's new rows
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int j = t + 256 * q;
            const int f4 = j % F4R, py = (j / F4R) % NEW, c = (j / F4R) / NEW;
            const int iy = iy0 + py;
            const bool inside = j < NPC && iy >= 0 && iy < H;
            if constexpr (U8) {
                pre8[q] = 0u;
                in8[q] = inside;
                if (inside)
                    pre8[q] = *reinterpret_cast<const uint32_t*>(x8 + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
            } else {
                pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inside) pre[q] = ld4(x + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
            }
        }
    };
    auto stash_patch = [&](int tt) {
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int j = t + 256 * q;
            if (j < NPC) {
                const int py = (j / F4R) % NEW, cc = (j / F4R) / NEW;
                const int slot = (NEW * tt + 2 + py + PAD + 4 * PH) % PH;
                _Float16* d = Ps + (cc * PH + slot) * PW + PAD + 4 * (j % F4R);  // patch column = ix + PAD (odd: 3 + 4 f4)
                float4 v;
                if constexpr (U8) {
                    const int c = (j / F4R) / NEW;
                    const bool norm = nrm != nullptr;
                    v = in8[q] ? u8x4_pixels(pre8[q], norm, norm ? nrm[c] : 0.f, norm ? nrm[3 + c] : 1.f, norm ? nrm[6 + c] : 1.f)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    v = pre[q];
                }
                typedef _Float16 sp_half2 __attribute__((ext_vector_type(2)));
                d[0] = (_Float16)v.x;                             // columns 3 + 4 f4 .. 6 + 4 f4: the middle pair is 4-B aligned
                const sp_half2 mid = {(_Float16)v.y, (_Float16)v.z};
                *reinterpret_cast<sp_half2*>(d + 1) = mid;
                d[3] = (_Float16)v.w;
            }
        }
    };
    __syncthreads();                                      // Bs and the zeroed patch are in place
    fetch_patch(blockIdx.x, -NV);

    for (long img = blockIdx.x; img < imgs; img += gridDim.x) {
        sp_half4 carry[NIT], carry2[NIT];
        const sp_half4 hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
        for (int q = 0; q < NIT; ++q) carry[q] = carry2[q] = hz;
        for (int tt = -NV; tt < ntt; ++tt) {
            __syncthreads();
            stash_patch(tt);
            {
                long nimg = img;
                int nt2 = tt + 1;
                if (nt2 == ntt) { nt2 = -NV; nimg += gridDim.x; }
                if (nimg >= imgs) nimg = blockIdx.x;
                fetch_patch(nimg, nt2);
            }
            asm volatile("" ::: "memory");
            if (tt < 0) continue;                         // stage-only steps at the top of an image
            __syncthreads();
            // slot of this wave's conv row, tap ky: (base + ky) mod PH with base = (NEW tt + rr2) mod PH, wave-uniform.
            // One copy of the multiply loop per value: the patch-row offsets stay instruction immediates.
            const int base = __builtin_amdgcn_readfirstlane((NEW * tt + rr2) % PH);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            auto multiply = [&](auto bc) {
                constexpr int B = decltype(bc)::value;
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    // group 2 s + h: (c, ky) -> patch row; the missing 22nd group re-reads the 21st (its weights are zero)
                    const int g0 = 2 * s, g1 = 2 * s + 1 < NG ? 2 * s + 1 : NG - 1;
                    const int off0 = ((g0 / KH) * PH + (B + g0 % KH) % PH) * PW;
                    const int off1 = ((g1 / KH) * PH + (B + g1 % KH) % PH) * PW;
                    // the 8 kx of position ox are patch columns 2 ox .. 2 ox + 7: four dwords at dword index ox of the row
                    const uint32_t* ap = reinterpret_cast<const uint32_t*>(pa + (h ? off1 : off0));
                    typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
                    const u32x4a av = {ap[0], ap[1], ap[2], ap[3]};
                    const sp_half8 a = __builtin_bit_cast(sp_half8, av);
                    const sp_half8 b0 = *reinterpret_cast<const sp_half8*>(pb + 16 * s);
                    const sp_half8 b1 = *reinterpret_cast<const sp_half8*>(pb + 32 * LDBH + 16 * s);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc1, 0, 0, 0);
                }
            };
            switch (base) {
                case 0: multiply(std::integral_constant<int, 0>{}); break;
                case 1: multiply(std::integral_constant<int, 1>{}); break;
                case 2: multiply(std::integral_constant<int, 2>{}); break;
                case 3: multiply(std::integral_constant<int, 3>{}); break;
                case 4: multiply(std::integral_constant<int, 4>{}); break;
                case 5: multiply(std::integral_constant<int, 5>{}); break;
                case 6: multiply(std::integral_constant<int, 6>{}); break;
                case 7: multiply(std::integral_constant<int, 7 % PH>{}); break;       // (RT = 2 only)
                default: multiply(std::integral_constant<int, 8 % PH>{}); break;
            }
            // both 32-channel halves of the tile go to the LDS together (a position's row holds all 64 channels), ONE barrier,
            // one pooling sweep - three barriers per tile instead of five
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                    Ts[rr * SP_LDTH + i] = (_Float16)fmaxf(fmaf(acc0[r], sc0, sh0), 0.f);
                    Ts[rr * SP_LDTH + 32 + i] = (_Float16)fmaxf(fmaf(acc1[r], sc1, sh1), 0.f);
                }
                __syncthreads();
                const int ch = 4 * c4;
#pragma unroll
                for (int q = 0; q < NIT; ++q) {
                    const int px = pxa + 16 * q;
                    sp_half4 m0 = hz, m1 = hz;
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int ox = 2 * px + dx;
                        if (ox < 0) continue;
                        m0 = __builtin_elementwise_max(m0, *reinterpret_cast<const sp_half4*>(&Ts[ox * SP_LDTH + 4 * c4]));
                        if (RT == 2)
                            m1 = __builtin_elementwise_max(m1, *reinterpret_cast<const sp_half4*>(&Ts[(WO + ox) * SP_LDTH + 4 * c4]));
                    }
                    sp_half4 o4;
                    bool emit;
                    long orow;
                    if (RT == 2) {
                        o4 = __builtin_elementwise_max(__builtin_elementwise_max(carry[q], m0), m1);
                        carry[q] = m1;
                        emit = true;
                        orow = (img * (Ho2 / 2) + tt) * (long)(WO / 2) + px;
                    } else {
                        o4 = __builtin_elementwise_max(__builtin_elementwise_max(carry2[q], carry[q]), m0);
                        emit = (tt & 1) != 0;
                        orow = (img * (Ho2 / 2) + (tt >> 1)) * (long)(WO / 2) + px;
                        carry2[q] = carry[q];
                        carry[q] = m0;
                    }
                    if (emit && ch < O) {
                        // (element (row, c) lives at row * ldo + (c >> 5) * obs + (c & 31): obs = 32 is the row-major matrix,
                        // ldo = 32 with obs = rows * 32 the channel-blocked buffer [c / 32][rows][32] of the fused dense layers)
                        *reinterpret_cast<sp_half4*>(out + orow * ldo + (ch >> 5) * obs + (ch & 31)) = o4;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ BN+ReLU+maxpool 3x3 s2 p1
// IDX: also store, per pooled element, WHICH window element is the maximum (0..8, row-major in the 3 x 3 window; the first
// maximal one of the scan, as torch's max_pool2d picks it): the backward then routes gradients by index - no value
// comparison, no re-read of the conv0 map, ties resolved exactly as the reference does.
template <bool IDX>
__global__ __launch_bounds__(256) void bnrelu_maxpool_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                             long ldo, long Mout, int C, int Hi, int Wi, int Ho, int Wo,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             unsigned char* __restrict__ amax) {
    const long total = Mout * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C;
        const int c = (int)(idx - row * C);
        const long img = row / ((long)Ho * Wo);
        const int rem = (int)(row - img * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float sc = scale[c], sh = shift[c];
        float m = IDX ? -1.f : 0.f;     // relu output is >= 0 and the window always holds a valid tap
        int am = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= Wi) continue;
                const float v = fmaxf(fmaf(in[((img * Hi + iy) * Wi + ix) * ldi + c], sc, sh), 0.f);
                if (IDX) { if (v > m) { m = v; am = 3 * (dy + 1) + dx + 1; } }
                else m = fmaxf(m, v);
            }
        }
        out[row * ldo + c] = m;
        if (IDX) amax[row * C + c] = (unsigned char)am;
    }
}

// same, 4 channels per thread with 16-B accesses (C % 4 == 0, aligned pointers / leading dimensions)
template <bool IDX>
__global__ __launch_bounds__(256) void bnrelu_maxpool_vec4_kernel(const float* __restrict__ in, long ldi,
                                                                  float* __restrict__ out, long ldo, long Mout, int C4,
                                                                  int Hi, int Wi, int Ho, int Wo,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  unsigned char* __restrict__ amax) {
    const long total = Mout * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C4;
        const int c = 4 * (int)(idx - row * C4);
        const long img = row / ((long)Ho * Wo);
        const int rem = (int)(row - img * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float4 sc = ld4(scale + c), sh = ld4(shift + c);
        const float m0 = IDX ? -1.f : 0.f;
        float4 m = make_float4(m0, m0, m0, m0);
        unsigned am = 0;                                      // four window indices, one per byte
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= Wi) continue;
                const float4 v = ld4(in + ((img * Hi + iy) * Wi + ix) * ldi + c);
                const float a0 = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), a1 = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
                const float a2 = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), a3 = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
                if (IDX) {
                    const unsigned li = 3 * (dy + 1) + dx + 1;
                    if (a0 > m.x) { m.x = a0; am = (am & 0xffffff00u) | li; }
                    if (a1 > m.y) { m.y = a1; am = (am & 0xffff00ffu) | (li << 8); }
                    if (a2 > m.z) { m.z = a2; am = (am & 0xff00ffffu) | (li << 16); }
                    if (a3 > m.w) { m.w = a3; am = (am & 0x00ffffffu) | (li << 24); }
                } else {
                    m.x = fmaxf(m.x, a0); m.y = fmaxf(m.y, a1); m.z = fmaxf(m.z, a2); m.w = fmaxf(m.w, a3);
                }
            }
        }
        *reinterpret_cast<float4*>(out + row * ldo + c) = m;
        if (IDX) *reinterpret_cast<unsigned*>(amax + row * (4L * C4) + c) = am;
    }
}

// ------------------------------------------------------------------------------------------------ BN+ReLU+global average pool
// out[img][c] = mean over the S2 positions of relu(x*scale+shift)
template <bool IN16 = false>
__global__ __launch_bounds__(256) void bnrelu_avgpool_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                             long ldo, int C, int S2, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, long ibs = 32) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long img = blockIdx.x;
    float acc = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c];
        for (int r = rl; r < S2; r += 4) {
            const long off = (img * S2 + r) * ldi + (c >> 5) * ibs + (c & 31);     // ibs = 32: the row-major matrix
            const float v = IN16 ? (float)reinterpret_cast<const _Float16*>(in)[off] : in[off];
            acc += fmaxf(fmaf(v, sc, sh), 0.f);
        }
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C)
        out[img * ldo + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)S2;
}

}  // namespace

// x [imgs][Cin][H][W] (NCHW, as the datasets deliver patches) -> out [imgs*Ho*Wo][O] channels-last (ldc)
GNX_EXPORT int gnx_conv_stem(const float* x, const float* w, float* out, long ldc, long imgs, int Cin, int H, int W,
                             int O, int KH, int KW, int stride, int pad, hipStream_t stream) {
    if (!x || !w || !out || imgs < 0 || Cin <= 0 || O <= 0 || KH <= 0 || KH > 7 || KW <= 0 || KW > 8 || stride <= 0 ||
        ldc < O)
        return GNX_ERR_BAD_ARG;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return GNX_ERR_BAD_ARG;
    const long M = imgs * Ho * Wo;
    if (M == 0) return GNX_OK;
    if (O <= 64 && Cin == 3 && ((stride == 2 && KH == 7 && KW == 7) || (stride == 1 && KH == 3 && KW == 3))) {
        const int tiles_x = gnx_cdiv(Wo, 16), tiles_y = gnx_cdiv(Ho, 8);
        const long ntiles = imgs * tiles_x * tiles_y;
        const int PH = 7 * stride + KH, PW = (15 * stride + 8 + 1) & ~1;
        const size_t lds2 = ((size_t)64 * (Cin * KH * 8 + 4) + (size_t)Cin * PH * PW) * sizeof(float);
        const int grid2 = (int)(ntiles < 256 * 3 ? ntiles : 256 * 3);
        if (stride == 2)
            conv_stem_patch_kernel<2, 7, 3><<<grid2, 256, lds2, stream>>>(x, w, out, ldc, Cin, H, W, Ho, Wo, O, KW, pad,
                                                                      tiles_x, tiles_y, ntiles);
        else
            conv_stem_patch_kernel<1, 3, 3><<<grid2, 256, lds2, stream>>>(x, w, out, ldc, Cin, H, W, Ho, Wo, O, KW, pad,
                                                                      tiles_x, tiles_y, ntiles);
        return gnx_launch_status();
    }
    const size_t lds_bytes = (size_t)(ST_BM + ST_BN) * (KH * 8 + 4) * sizeof(float);
    dim3 grid(gnx_cdiv(M, ST_BM), gnx_cdiv(O, ST_BN));
    conv_stem_kernel<<<grid, 256, lds_bytes, stream>>>(x, w, out, ldc, M, Cin, H, W, Ho, Wo, O, KH, KW, stride, pad);
    return gnx_launch_status();
}

// x [imgs][3][H][W] (NCHW patches), w [O][3][7][7] -> out [imgs*(Ho/2)*(Wo/2)][O] (ldo):
// maxpool3x3s2p1(relu(scale * conv7x7s2p3(x) + shift)) without the intermediate map.  Supported geometry: Cin = 3, the
// conv output is 64 or 128 wide and even-high (128- / 256-px patches), O <= 64 and O % 4 == 0; anything else returns
// GNX_ERR_UNSUPPORTED and the caller runs gnx_conv_stem + gnx_bnrelu_maxpool.
template <bool H16, bool U8 = false>
static int stem_pool_launch(const void* x, const float* w, float* out, long ldo, long imgs, int Cin, int H, int W, int O,
                            int KH, int KW, int stride, int pad, const float* scale, const float* shift, hipStream_t stream,
                            const float* nrm = nullptr, unsigned char* amax = nullptr) {
    if (!x || !w || !out || !scale || !shift || imgs < 0 || Cin <= 0 || O <= 0 || H <= 0 || W <= 0 || ldo < O)
        return GNX_ERR_BAD_ARG;
    if (amax && (H16 || U8 || (reinterpret_cast<uintptr_t>(amax) & 3) != 0)) return GNX_ERR_UNSUPPORTED;
    if (Cin != 3 || KH != 7 || KW != 7 || stride != 2 || pad != 3 || O > 64 || O % 4 != 0 || ldo % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(out) & (H16 ? 7 : 15)) != 0)
        return GNX_ERR_UNSUPPORTED;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if ((Wo != 64 && Wo != 128) || W != 2 * Wo || Ho % 2 != 0 || Ho <= 0 ||
        (reinterpret_cast<uintptr_t>(x) & (U8 ? 3 : 15)) != 0)
        return GNX_ERR_UNSUPPORTED;                        // 16-B (u8: 4-B) row pieces, whole 128-position tiles
    if (imgs == 0) return GNX_OK;
    const int RT = 128 / Wo, PH = 2 * (RT - 1) + 7, PW = ((Wo - 1) * 2 + 8 + 2) & ~1;
    const size_t lds_bytes = ((size_t)64 * (3 * 7 * 8 + 4) + (size_t)3 * PH * PW + (size_t)128 * SP_LDT) * sizeof(float);
    const int per_cu = lds_bytes <= 80 * 1024 ? 2 : 1;     // 128-px geometry: 77 KB, two workgroups per CU
    const int grid = (int)(imgs < 256 * per_cu ? imgs : 256 * per_cu);
    if (Wo == 64) {
        static bool conf = false;
        if (!conf) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_pool_kernel<64, H16, U8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                return GNX_ERR_LAUNCH;
            conf = true;
        }
        if constexpr (!H16 && !U8) {
            if (amax) {
                static bool confi = false;
                if (!confi) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_pool_kernel<64, false, false, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                        return GNX_ERR_LAUNCH;
                    confi = true;
                }
                conv_stem_pool_kernel<64, false, false, true><<<grid, 256, lds_bytes, stream>>>(x, w, out, ldo, H, W, O, scale,
                                                                                                shift, imgs, nrm, amax);
                return gnx_launch_status();
            }
        }
        conv_stem_pool_kernel<64, H16, U8><<<grid, 256, lds_bytes, stream>>>(x, w, out, ldo, H, W, O, scale, shift, imgs, nrm);
    } else {
        static bool conf = false;
        if (!conf) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_pool_kernel<128, H16, U8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                return GNX_ERR_LAUNCH;
            conf = true;
        }
        if constexpr (!H16 && !U8) {
            if (amax) {
                static bool confi = false;
                if (!confi) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_pool_kernel<128, false, false, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                        return GNX_ERR_LAUNCH;
                    confi = true;
                }
                conv_stem_pool_kernel<128, false, false, true><<<grid, 256, lds_bytes, stream>>>(x, w, out, ldo, H, W, O, scale,
                                                                                                 shift, imgs, nrm, amax);
                return gnx_launch_status();
            }
        }
        conv_stem_pool_kernel<128, H16, U8><<<grid, 256, lds_bytes, stream>>>(x, w, out, ldo, H, W, O, scale, shift, imgs, nrm);
    }
    return gnx_launch_status();
}
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool(const float* x, const float* w, float* out, long ldo, long imgs, int Cin,
                                            int H, int W, int O, int KH, int KW, int stride, int pad, const float* scale,
                                            const float* shift, hipStream_t stream) {
    return stem_pool_launch<false>(x, w, out, ldo, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, stream);
}
// the same, also recording each pooled element's window index (0..8, torch's first-maximum rule) in argmax [pooled rows][O]
// bytes: the forward of the f-trained step under running statistics (training.py:126) - its backward routes by index
// (gnx_maxpool_bwd_argmax_bnrelu) and never needs the conv0 map
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool_argmax(const float* x, const float* w, float* out, long ldo,
                                                   unsigned char* argmax, long imgs, int Cin, int H, int W, int O, int KH,
                                                   int KW, int stride, int pad, const float* scale, const float* shift,
                                                   hipStream_t stream) {
    if (!argmax) return GNX_ERR_BAD_ARG;
    return stem_pool_launch<false>(x, w, out, ldo, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, stream, nullptr,
                                   argmax);
}
// the same with the pooled map stored as fp16 [rows][ldo halves] (config 5 with fp16 block buffers)
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool_h16(const float* x, const float* w, void* out16, long ldo, long imgs, int Cin,
                                                int H, int W, int O, int KH, int KW, int stride, int pad,
                                                const float* scale, const float* shift, hipStream_t stream) {
    return stem_pool_launch<true>(x, w, reinterpret_cast<float*>(out16), ldo, imgs, Cin, H, W, O, KH, KW, stride, pad, scale,
                                  shift, stream);
}

// The fused stem with fp16 MATRIX OPERANDS (config 5's fp16 MFMA conv path: patch and weights rounded to fp16 as they are
// staged, fp32 accumulate) storing the pooled map as fp16.  x: float patches, or uint8 patches when x_is_u8 (then norm as
// for gnx_conv_stem_bnrelu_maxpool_u8).  Same geometries as gnx_conv_stem_bnrelu_maxpool; GNX_ERR_UNSUPPORTED otherwise.
template <bool U8>
static int stem_pool_f16_launch(const void* x, const float* w, void* out16, long ldo, long imgs, int Cin, int H, int W, int O,
                                int KH, int KW, int stride, int pad, const float* scale, const float* shift,
                                const float* nrm, hipStream_t stream, long obs = 32) {
    if (!x || !w || !out16 || !scale || !shift || imgs < 0 || Cin <= 0 || O <= 0 || H <= 0 || W <= 0 ||
        (obs == 32 ? ldo < O : ldo != 32))
        return GNX_ERR_BAD_ARG;
    if (Cin != 3 || KH != 7 || KW != 7 || stride != 2 || pad != 3 || O > 64 || O % 4 != 0 || ldo % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(out16) & 7) != 0)
        return GNX_ERR_UNSUPPORTED;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if ((Wo != 64 && Wo != 128) || W != 2 * Wo || Ho % 2 != 0 || Ho <= 0 ||
        (reinterpret_cast<uintptr_t>(x) & (U8 ? 3 : 15)) != 0)
        return GNX_ERR_UNSUPPORTED;
    if (imgs == 0) return GNX_OK;
    const int RT = 128 / Wo, PH = 2 * RT + 5, PW = (Wo - 1) * 2 + 8;           // (the kernel's ring of input rows)
    const size_t lds_bytes = ((size_t)64 * (11 * 16 + 8) + (size_t)((3 * PH * PW + 7) & ~7) + (size_t)128 * SP_LDTH) * 2;
    const int per_cu = lds_bytes <= 53 * 1024 ? 3 : 2;     // 256-px geometry: 52 992 B, three workgroups per CU
    const int grid = (int)(imgs < 256 * per_cu ? imgs : 256 * per_cu);
    _Float16* o = reinterpret_cast<_Float16*>(out16);
    if (Wo == 64)
        conv_stem_pool_f16_kernel<64, U8><<<grid, 256, lds_bytes, stream>>>(x, w, o, ldo, H, W, O, scale, shift, imgs, nrm, obs);
    else
        conv_stem_pool_f16_kernel<128, U8><<<grid, 256, lds_bytes, stream>>>(x, w, o, ldo, H, W, O, scale, shift, imgs, nrm, obs);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool_f16mul(const void* x, int x_is_u8, const float* w, void* out16, long ldo,
                                                   long imgs, int Cin, int H, int W, int O, int KH, int KW, int stride,
                                                   int pad, const float* scale, const float* shift, const float* norm,
                                                   hipStream_t stream) {
    if (x_is_u8)
        return stem_pool_f16_launch<true>(x, w, out16, ldo, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, norm, stream);
    if (norm) return GNX_ERR_BAD_ARG;                      // float patches are taken as already transformed
    return stem_pool_f16_launch<false>(x, w, out16, ldo, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, nullptr, stream);
}
// The same storing into a CHANNEL-BLOCKED buffer [O / 32][rows_total][32] halves (the layout gnx_dense_layer_f16 streams:
// a layer's 32-channel slice of consecutive pixels is contiguous memory): element (row, c) at (c >> 5) * rows_total * 32 +
// row * 32 + (c & 31).  32 | O.
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool_f16mul_cb(const void* x, int x_is_u8, const float* w, void* out16, long rows_total,
                                                      long imgs, int Cin, int H, int W, int O, int KH, int KW, int stride,
                                                      int pad, const float* scale, const float* shift, const float* norm,
                                                      hipStream_t stream) {
    if (rows_total <= 0 || O % 32 != 0) return GNX_ERR_BAD_ARG;
    if (x_is_u8)
        return stem_pool_f16_launch<true>(x, w, out16, 32, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, norm, stream,
                                          rows_total * 32);
    if (norm) return GNX_ERR_BAD_ARG;
    return stem_pool_f16_launch<false>(x, w, out16, 32, imgs, Cin, H, W, O, KH, KW, stride, pad, scale, shift, nullptr, stream,
                                       rows_total * 32);
}

// The fused stem on uint8 patches x8 [imgs][3][H][W] (image_datasets.py:102-105 does ToTensor on the host):
// ToTensor (u8 / 255) and, with norm != NULL, Normalize ((v - mean[c]) / std[c]; norm = {mean[3], std[3], 1/std[3]} on the
// device) are applied between the load and the LDS stash - bit-for-bit the floats the float entry point would be given.
// out_f16 != 0: the pooled map is stored as fp16 (config 5's fp16 block buffers).  Geometry as gnx_conv_stem_bnrelu_maxpool.
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool_u8(const uint8_t* x8, const float* w, void* out, long ldo, long imgs, int Cin,
                                               int H, int W, int O, int KH, int KW, int stride, int pad, const float* scale,
                                               const float* shift, const float* norm, int out_f16, hipStream_t stream) {
    if (out_f16)
        return stem_pool_launch<true, true>(x8, w, reinterpret_cast<float*>(out), ldo, imgs, Cin, H, W, O, KH, KW, stride, pad,
                                            scale, shift, stream, norm);
    return stem_pool_launch<false, true>(x8, w, reinterpret_cast<float*>(out), ldo, imgs, Cin, H, W, O, KH, KW, stride, pad,
                                         scale, shift, stream, norm);
}

// ToTensor (+ Normalize) as its own pass: x8 [imgs][C][H][W] uint8 -> out float, the same floats as above.  For the paths
// that need float patches in HBM (training forward: conv0's weight gradient re-reads them; geometries the fused stem does
// not take).  H * W % 4 == 0, 4-B aligned input, 16-B aligned output; norm as above (C must be 3 with it) or NULL.
GNX_EXPORT int gnx_u8_to_f32(const uint8_t* x8, float* out, long imgs, int C, int H, int W, const float* norm,
                             hipStream_t stream) {
    if (!x8 || !out || imgs < 0 || C <= 0 || H <= 0 || W <= 0 || (norm && C != 3)) return GNX_ERR_BAD_ARG;
    const long hw = (long)H * W;
    if (hw % 4 != 0 || (reinterpret_cast<uintptr_t>(x8) & 3) != 0 || !al16(out)) return GNX_ERR_UNSUPPORTED;
    const long n4 = imgs * C * hw / 4;
    if (n4 == 0) return GNX_OK;
    const long blocks = (n4 + 255) / 256;
    u8_to_f32_kernel<<<(int)(blocks < 256 * 16 ? blocks : 256 * 16), 256, 0, stream>>>(x8, out, n4, hw / 4, C, norm);
    return gnx_launch_status();
}

// in [imgs*Hi*Wi][C] (ldi) -> out [imgs*Ho*Wo][C] (ldo): max over 3x3 s2 p1 windows of relu(in*scale+shift)
static int bnrelu_maxpool_launch(const float* in, long ldi, float* out, long ldo, unsigned char* amax, long imgs, int C,
                                 int Hi, int Wi, const float* scale, const float* shift, hipStream_t stream) {
    if (!in || !out || !scale || !shift || imgs < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || ldi < C || ldo < C)
        return GNX_ERR_BAD_ARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const long Mout = imgs * Ho * Wo;
    if (Mout == 0) return GNX_OK;
    if (C % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0 && al16(in) && al16(out) && al16(scale) && al16(shift) &&
        (reinterpret_cast<uintptr_t>(amax) & 3) == 0) {
        long blocks = (Mout * (C / 4) + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        if (amax)
            bnrelu_maxpool_vec4_kernel<true><<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C / 4, Hi, Wi, Ho, Wo,
                                                                               scale, shift, amax);
        else
            bnrelu_maxpool_vec4_kernel<false><<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C / 4, Hi, Wi, Ho, Wo,
                                                                                scale, shift, nullptr);
        return gnx_launch_status();
    }
    long blocks = (Mout * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (amax)
        bnrelu_maxpool_kernel<true><<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C, Hi, Wi, Ho, Wo, scale, shift, amax);
    else
        bnrelu_maxpool_kernel<false><<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C, Hi, Wi, Ho, Wo, scale, shift,
                                                                      nullptr);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_bnrelu_maxpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int Hi, int Wi,
                                  const float* scale, const float* shift, hipStream_t stream) {
    return bnrelu_maxpool_launch(in, ldi, out, ldo, nullptr, imgs, C, Hi, Wi, scale, shift, stream);
}
// The same, also recording which window element (0..8, row-major in the 3 x 3 window; the first maximal one, as torch's
// max_pool2d) gave each pooled value: argmax [imgs*Ho*Wo][C] bytes, consumed by gnx_maxpool_bwd_argmax.
GNX_EXPORT int gnx_bnrelu_maxpool_argmax(const float* in, long ldi, float* out, long ldo, unsigned char* argmax, long imgs,
                                         int C, int Hi, int Wi, const float* scale, const float* shift, hipStream_t stream) {
    if (!argmax) return GNX_ERR_BAD_ARG;
    return bnrelu_maxpool_launch(in, ldi, out, ldo, argmax, imgs, C, Hi, Wi, scale, shift, stream);
}

// in [imgs*S2][C] (ldi) -> out [imgs][C] (ldo): mean over positions of relu(in*scale+shift)
GNX_EXPORT int gnx_bnrelu_avgpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S2,
                                  const float* scale, const float* shift, hipStream_t stream) {
    if (!in || !out || !scale || !shift || imgs < 0 || C <= 0 || S2 <= 0 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (imgs == 0) return GNX_OK;
    if (imgs > 2147483647L) return GNX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)imgs, gnx_cdiv(C, 64));
    bnrelu_avgpool_kernel<false><<<grid, 256, 0, stream>>>(in, ldi, out, ldo, C, S2, scale, shift);
    return gnx_launch_status();
}
// the same reading fp16 activations [rows][ldi halves] (config 5 with fp16 block buffers)
GNX_EXPORT int gnx_bnrelu_avgpool_h16(const void* in16, long ldi, float* out, long ldo, long imgs, int C, int S2,
                                      const float* scale, const float* shift, hipStream_t stream) {
    if (!in16 || !out || !scale || !shift || imgs < 0 || C <= 0 || S2 <= 0 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (imgs == 0) return GNX_OK;
    if (imgs > 2147483647L) return GNX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)imgs, gnx_cdiv(C, 64));
    bnrelu_avgpool_kernel<true><<<grid, 256, 0, stream>>>(reinterpret_cast<const float*>(in16), ldi, out, ldo, C, S2, scale,
                                                          shift);
    return gnx_launch_status();
}
// the same reading the channel-blocked buffer [C / 32][rows_total][32] halves of the fused dense layers
GNX_EXPORT int gnx_bnrelu_avgpool_h16_cb(const void* in16, long rows_total, float* out, long ldo, long imgs, int C, int S2,
                                         const float* scale, const float* shift, hipStream_t stream) {
    if (!in16 || !out || !scale || !shift || imgs < 0 || C <= 0 || S2 <= 0 || rows_total < imgs * S2 || ldo < C || C % 32 != 0)
        return GNX_ERR_BAD_ARG;
    if (imgs == 0) return GNX_OK;
    if (imgs > 2147483647L) return GNX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)imgs, gnx_cdiv(C, 64));
    bnrelu_avgpool_kernel<true><<<grid, 256, 0, stream>>>(reinterpret_cast<const float*>(in16), 32, out, ldo, C, S2, scale, shift,
                                                          rows_total * 32);
    return gnx_launch_status();
}
