// DenseNet-BC backward kernels (gfx950, exact fp32 on the matrix cores).
//
// Autograd of /root/reference/gridnext/densenet.py is torch's; these kernels compute the same gradients:
//   data gradients of conv1x1 / conv3x3 reuse the FORWARD kernels of conv1x1.hip / conv3x3.hip with transformed weights
//     (gnx_transpose_weight: W[N][K] -> [K][N];  gnx_repack_conv3x3_bwd: W[N][K][3][3] -> [flipped tap][K][N]);
//   gnx_wgrad_bnrelu   : weight gradient of conv1x1 (taps=1, optionally through the transition's 2x2 average) and
//                        conv3x3 (taps=9): dW[tap][n][k] = sum_m dY[m][n] * act(X)[nbr(m,tap)][k], the BN+ReLU of the
//                        forward prologue recomputed while staging (the activated tensor is never stored);
//   gnx_conv0_wgrad    : weight gradient of the stem conv from NCHW patches (patch-resident, like the forward);
//   gnx_maxpool_bwd, gnx_avgpool2_bwd, gnx_rows_broadcast : the pooling adjoints.
// The reduction over positions (millions of rows) is split over workgroups; each writes a partial slab and
// gnx_wgrad_reduce sums the slabs in a fixed order (deterministic, no float atomics).
// MFMA operand maps for a weight gradient (reduction index = position m): A[i=n][kk=m] = dY[m][n],
// B[kk=m][j=k] = act(X)[m'][k]; both are read from [position][channel] LDS images with conflict-free ds_read_b32.
#include "common.h"
#include <type_traits>

namespace {

__device__ __forceinline__ float act1(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }

constexpr int WG_BM = 64;        // positions per staged tile
constexpr int WG_KR = 128;       // k-range per workgroup (4 waves x 32)
constexpr int LDX = WG_KR + 4;   // LDS row stride of the X strip (floats)

// slab[(split*T + tap)*N*K + n*K + k]
// NT = 32-wide n-tiles per workgroup (1x1: NT=4 so a staged X row meets 128 output channels; 3x3: NT=1, 9 taps)
template <int T, int NT, bool POOL>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dY, long lddy,
                                                    const float* __restrict__ X, long ldx,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    float* __restrict__ slabs, long M, int N, int K, int S,
                                                    long tiles_per_split, int vec, int vecY) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int halo = (T == 9) ? S + 1 : 0;
    const int strip = WG_BM + 2 * halo;
    constexpr int LDY = 32 * NT;
    float* Ys = lds;                               // [WG_BM][LDY]
    float* Xs = lds + WG_BM * LDY;                 // [strip][LDX]
    unsigned* Vm = reinterpret_cast<unsigned*>(Xs + strip * LDX);   // [WG_BM] tap-validity bits (T == 9)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const int n0 = blockIdx.y * 32 * NT;
    const int kbase = blockIdx.z * WG_KR;
    const int kc = kbase + 32 * wave;              // this wave's 32 k columns
    const bool has_act = scale != nullptr;
    const long ntiles = (M + WG_BM - 1) / WG_BM;
    const long tile0 = (long)blockIdx.x * tiles_per_split;
    const long tile1 = min(tile0 + tiles_per_split, ntiles);

    f32x16 acc[T * NT];
#pragma unroll
    for (int a = 0; a < T * NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    for (long tile = tile0; tile < tile1; ++tile) {
        const long m0 = tile * WG_BM;
        __syncthreads();
        // dY tile: [64][32*NT]
        if (vecY && n0 + LDY <= N) {
            for (int idx = t; idx < WG_BM * (LDY / 4); idx += 256) {
                const int r = idx / (LDY / 4), c4 = idx - r * (LDY / 4);
                const long m = m0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < M) v = *reinterpret_cast<const float4*>(dY + m * lddy + n0 + 4 * c4);
                *reinterpret_cast<float4*>(&Ys[r * LDY + 4 * c4]) = v;
            }
        } else {
            for (int idx = t; idx < WG_BM * LDY; idx += 256) {
                const int r = idx / LDY, c = idx - r * LDY;
                const long m = m0 + r;
                Ys[idx] = (m < M && n0 + c < N) ? dY[m * lddy + n0 + c] : 0.f;
            }
        }
        // X strip (activated), [strip][128]
        for (int idx = t; idx < strip * (WG_KR / 4); idx += 256) {
            const int r = idx / (WG_KR / 4), q = idx - r * (WG_KR / 4);
            const int k = kbase + 4 * q;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            const long p = m0 - halo + r;
            if (p >= 0 && p < M && k < K) {
                if (POOL) {
                    const int So = S >> 1;
                    const long img = p / (So * So);
                    const int rem = (int)(p - img * So * So);
                    const int oy = rem / So, ox = rem - oy * So;
                    const long src = ((img * S + 2 * oy) * S + 2 * ox) * ldx;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k + e < K) {
                            const float sc = has_act ? scale[k + e] : 1.f, sh = has_act ? shift[k + e] : 0.f;
                            float s4 = 0.f;
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) {
                                const float xv = X[src + ((qd >> 1) * (long)S + (qd & 1)) * ldx + k + e];
                                s4 += has_act ? act1(xv, sc, sh) : xv;
                            }
                            v[e] = 0.25f * s4;
                        }
                    }
                } else if (vec && k + 3 < K) {
                    const float4 xv = *reinterpret_cast<const float4*>(X + p * ldx + k);
                    v[0] = xv.x; v[1] = xv.y; v[2] = xv.z; v[3] = xv.w;
                    if (has_act) {
                        const float4 sc = *reinterpret_cast<const float4*>(scale + k);
                        const float4 sh = *reinterpret_cast<const float4*>(shift + k);
                        v[0] = act1(v[0], sc.x, sh.x); v[1] = act1(v[1], sc.y, sh.y);
                        v[2] = act1(v[2], sc.z, sh.z); v[3] = act1(v[3], sc.w, sh.w);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < K) {
                            const float xv = X[p * ldx + k + e];
                            v[e] = has_act ? act1(xv, scale[k + e], shift[k + e]) : xv;
                        }
                }
            }
            *reinterpret_cast<float4*>(&Xs[r * LDX + 4 * q]) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (T == 9) {
            for (int r = t; r < WG_BM; r += 256) {
                const long m = m0 + r;
                unsigned mask = 0;
                if (m < M) {
                    const int rem = (int)(m % ((long)S * S));
                    const int y = rem / S, x = rem - y * S;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                        if (yy >= 0 && yy < S && xx >= 0 && xx < S) mask |= 1u << tap;
                    }
                }
                Vm[r] = mask;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int mm = 0; mm < WG_BM; mm += 2) {
            const float a = Ys[(mm + h) * LDY + i];
            if (T == 9) {
                const unsigned vm = Vm[mm + h];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int off = (S + 1) + (tap / 3 - 1) * S + (tap % 3 - 1);
                    const float b = Xs[(mm + h + off) * LDX + 32 * wave + i];
                    const float am = ((vm >> tap) & 1u) ? a : 0.f;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(am, b, acc[tap], 0, 0, 0);
                }
            } else {
                const float b = Xs[(mm + h) * LDX + 32 * wave + i];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int nt = 1; nt < NT; ++nt) {
                    const float an = Ys[(mm + h) * LDY + 32 * nt + i];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(an, b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    const int k = kc + i;
    if (k < K) {
#pragma unroll
        for (int a = 0; a < T * NT; ++a) {
            const int tap = (T == 1) ? 0 : a, nt = (T == 1) ? a : 0;
            float* dst = slabs + ((long)blockIdx.x * T + tap) * N * K;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) dst[(long)n * K + k] = acc[a][r];
            }
        }
    }
}

// Register-prefetched form for the common case (no pooling, 16-B aligned operands, 4 | K, a whole 32*NT-wide column tile,
// strip <= 256 * NX / 32 rows): the next tile's dY rows and X strip are fetched into registers (branch-free: clamped
// addresses, zeroed at the stash) while the current tile multiplies, so the global latency the plain kernel exposes once per
// 64 positions hides behind the 128 (1x1) / 288 (3x3) MFMAs of a tile.  Same slab layout, same fixed summation order.
template <int T, int NT, int NX>
__global__ __launch_bounds__(256) void wgrad_pf_kernel(const float* __restrict__ dY, long lddy,
                                                       const float* __restrict__ X, long ldx,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ slabs, long M, int N, int K, int S,
                                                       long tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int halo = (T == 9) ? S + 1 : 0;
    const int strip = WG_BM + 2 * halo;
    constexpr int LDY = 32 * NT, NLY = WG_BM * (LDY / 4) / 256;
    float* Ys = lds;                               // [WG_BM][LDY]
    float* Xs = lds + WG_BM * LDY;                 // [strip][LDX]
    unsigned* Vm = reinterpret_cast<unsigned*>(Xs + strip * LDX);   // [WG_BM] tap-validity bits (T == 9)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const int n0 = blockIdx.y * 32 * NT;
    const int kbase = blockIdx.z * WG_KR;
    const int kc = kbase + 32 * wave;              // this wave's 32 k columns
    const bool has_act = scale != nullptr;
    const long ntiles = (M + WG_BM - 1) / WG_BM;
    const long tile0 = (long)blockIdx.x * tiles_per_split;
    const long tile1 = min(tile0 + tiles_per_split, ntiles);
    if (tile0 >= tile1) {                          // an empty split still owes its (zero) slab
        const int k = kc + i;
        if (k < K)
            for (int a = 0; a < T * NT; ++a) {
                const int tap = (T == 1) ? 0 : a, nt = (T == 1) ? a : 0;
                float* dst = slabs + ((long)blockIdx.x * T + tap) * N * K;
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (n < N) dst[(long)n * K + k] = 0.f;
                }
            }
        return;
    }

    f32x16 acc[T * NT];
#pragma unroll
    for (int a = 0; a < T * NT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // this thread's X column quad is the same in every tile (256 % 32 == 0)
    const int q = t & 31;
    const int kq = kbase + 4 * q;
    const bool kok = kq < K;                       // 4 | K: the whole quad is in or out
    const int kld = kok ? kq : 0;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_act) { sc4 = *reinterpret_cast<const float4*>(scale + kld); sh4 = *reinterpret_cast<const float4*>(shift + kld); }
    float4 ry[NLY], rx[NX];
    auto fetch = [&](long tile) {
        const long m0 = tile * WG_BM;
        if constexpr (T == 1) {
            // BUFFER loads (a global_load issued by a wave that also multiplies costs the matrix pipe ~40 cycles, a
            // buffer_load nothing: tools/ubench/mfma_2x2.hip).  One resource per tile and operand; rows past M read 0.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const long left = M - m0;
            const int rows = (int)(left < WG_BM ? left : WG_BM);
            const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(dY + m0 * lddy), 0, (int)(((long)(rows - 1) * lddy + N) * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(X + m0 * ldx), 0, (int)(((long)(rows - 1) * ldx + K) * 4), 0x00020000);
            constexpr int RPJ = 256 / (LDY / 4);           // dY rows per 256 pieces
            const int c4 = t % (LDY / 4), r0y = t / (LDY / 4);
            const int voy = (int)((r0y * lddy + n0 + 4 * c4) * 4), vox = (int)(((t >> 5) * ldx + kld) * 4);
#pragma unroll
            for (int j = 0; j < NLY; ++j) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rY, voy, (int)(j * RPJ * lddy * 4), 0);
                ry[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rX, vox, (int)(j * 8 * ldx * 4), 0);
                rx[j] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NLY; ++j) {
            const int idx = t + 256 * j;
            const int r = idx / (LDY / 4), c4 = idx - r * (LDY / 4);
            long m = m0 + r;
            m = m < M ? m : M - 1;
            ry[j] = *reinterpret_cast<const float4*>(dY + m * lddy + n0 + 4 * c4);
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int r = (t >> 5) + 8 * j;        // strip row of piece t + 256 j
            long pp = m0 - halo + r;
            pp = pp < 0 ? 0 : (pp < M ? pp : M - 1);
            rx[j] = *reinterpret_cast<const float4*>(X + pp * ldx + kld);       // rows past the strip: a harmless re-read
        }
    };
    auto stash = [&](long tile) {
        const long m0 = tile * WG_BM;
#pragma unroll
        for (int j = 0; j < NLY; ++j) {
            const int idx = t + 256 * j;
            const int r = idx / (LDY / 4), c4 = idx - r * (LDY / 4);
            float4 v = ry[j];
            if (m0 + r >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Ys[r * LDY + 4 * c4]) = v;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int r = (t >> 5) + 8 * j;
            if (r < strip) {
                const long pp = m0 - halo + r;
                float4 v = rx[j];
                if (has_act)
                    v = make_float4(act1(v.x, sc4.x, sh4.x), act1(v.y, sc4.y, sh4.y), act1(v.z, sc4.z, sh4.z),
                                    act1(v.w, sc4.w, sh4.w));
                if (!kok || pp < 0 || pp >= M) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&Xs[r * LDX + 4 * q]) = v;
            }
        }
        if (T == 9) {
            for (int r = t; r < WG_BM; r += 256) {
                const long m = m0 + r;
                unsigned mask = 0;
                if (m < M) {
                    const int rem = (int)(m % ((long)S * S));
                    const int y = rem / S, x = rem - y * S;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                        if (yy >= 0 && yy < S && xx >= 0 && xx < S) mask |= 1u << tap;
                    }
                }
                Vm[r] = mask;
            }
        }
    };

    fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        __syncthreads();                           // the previous tile's fragment reads are done
        stash(tile);
        __syncthreads();
        fetch(tile + 1 < tile1 ? tile + 1 : tile); // branch-free; the last one is a harmless re-read
        asm volatile("" ::: "memory");             // keep the prefetch in front of the multiply
#pragma unroll 4
        for (int mm = 0; mm < WG_BM; mm += 2) {
            const float a = Ys[(mm + h) * LDY + i];
            if (T == 9) {
                const unsigned vm = Vm[mm + h];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int off = (S + 1) + (tap / 3 - 1) * S + (tap % 3 - 1);
                    const float b = Xs[(mm + h + off) * LDX + 32 * wave + i];
                    const float am = ((vm >> tap) & 1u) ? a : 0.f;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(am, b, acc[tap], 0, 0, 0);
                }
            } else {
                const float b = Xs[(mm + h) * LDX + 32 * wave + i];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
#pragma unroll
                for (int nt = 1; nt < NT; ++nt) {
                    const float an = Ys[(mm + h) * LDY + 32 * nt + i];
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(an, b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    const int k = kc + i;
    if (k < K) {
#pragma unroll
        for (int a = 0; a < T * NT; ++a) {
            const int tap = (T == 1) ? 0 : a, nt = (T == 1) ? a : 0;
            float* dst = slabs + ((long)blockIdx.x * T + tap) * N * K;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (n < N) dst[(long)n * K + k] = acc[a][r];
            }
        }
    }
}


// ================================================================================================ round 2: transposed images
// The kernels above read their MFMA operands one float at a time (ds_read_b32) from [position][channel] LDS images and, for
// the 3x3, mask image borders with a v_cndmask per MFMA inside the matrix stream (r1: 0.59 / 0.52 of the fp32 matrix peak).
// An operand of a weight-gradient MFMA is "my channel, two positions" (the reduction runs over positions), so four
// consecutive MFMA steps of a lane want FOUR CONSECUTIVE POSITIONS of one channel: with the LDS images transposed to
// [channel][position] that is one ds_read_b128.  Which two positions meet in one MFMA is free - lane half h takes positions
// 8g + 4h + s in step s of group g - it only reorders the fp32 sum over positions.
//   * staging: a thread loads a 4 position x 4 channel block (four coalesced 16-B loads; lanes of an 8-lane group walk
//     8 position blocks of one channel quad, so every row segment a wave touches is a full 128-B line), applies the
//     forward's BN+ReLU, and writes the four channel rows with ds_write_b128 (8 lanes = 128 contiguous bytes: conflict-free);
//   * rows are padded to 4 * odd floats: the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads (0 conflicts);
//   * 1x1: a wave owns 64 n x 64 k (2 x 2 accumulators): 4 fragment reads per 16 MFMAs (r1: 20 ds_read_b32);
//   * 3x3: dW[tap][n][k] = sum_m' dY[m' - dy S - dx][n] X[m'][k]: the SHIFT goes to the narrow operand (32 channels).  Three
//     copies of the dY strip are staged, one per dx, with the horizontal border baked in (copy_dx[q] = dY[q - dx] where
//     column(q) - dx lies inside the row, else 0); the vertical shift dy S is a multiple of 4 positions, i.e. an aligned
//     offset into the same copy; a group of 8 positions lies in ONE image row (S >= 8), so "row y' has no neighbour above /
//     below" is wave-uniform and is a scalar branch that SKIPS the 12 MFMAs of those taps (exact: they would add zeros).
//     No select, no mask in the matrix stream; 10 fragment reads per 36 MFMAs (r1: 10 per 9 plus 9 v_cndmask).
// Tiles are prefetched into registers one ahead (as wgrad_pf_kernel); two workgroups share a CU (<= 70 KB of LDS each),
// so one's staging hides behind the other's MFMAs.  Slab layout and the fixed-order reduce are unchanged.
constexpr int WT_PAD = 4;

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// ---- 1x1: dW[n][k] = sum_m dY[m][n] act(X[m][k]).  A workgroup owns 128 n x 256 k (a wave 64 n x 128 k: 2 x 4 accumulators)
// over its range of 32-position tiles: per tile it takes in 32 x (128 + 256) floats for 128 MFMAs per wave - 6 B per cycle
// and CU at the full matrix rate, where a 128 x 128 workgroup needs 8 (the r1 kernel and a first 128 x 128 transposed form
// both stalled at ~105 TFLOP/s on exactly that: ~10 B per cycle and CU is what the load path delivers).  The workgroups
// of one position range (one per 256-wide k block) get block ids with equal residue mod 8 and consecutive quotients: same
// XCD, launched together, so the dY tile they all read comes from HBM once.  grid (splits * k blocks, N / 128), 256 threads.
constexpr int W1_TILE = 32, W1_P = W1_TILE + WT_PAD;          // 36 = 4 * 9
// (bx, by): the block's coordinates in the grid of ONE layer's launch - blockIdx of wgrad1_t_kernel, or derived from the flat
// block id of wgrad1_t_batch_kernel, which runs several layers' grids as one launch.
__device__ __forceinline__ void wgrad1_t_body(const float* __restrict__ dY, long lddy, const float* __restrict__ X, long ldx,
                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                              float* __restrict__ slabs, long M, int N, int K, int nsplit, int kblocks,
                                              long tiles_per_split, int bx, int by) {
    __shared__ __attribute__((aligned(16))) float Yt[128 * W1_P];       // [n][position]
    __shared__ __attribute__((aligned(16))) float Xt[256 * W1_P];       // [k][position]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    // block id -> (split, k block): id = 8 * slot + xcd, slot = group * kblocks + kb, split = 8 * group + xcd  (8 | nsplit)
    const int xcd = bx & 7, slot = bx >> 3;
    const int split = 8 * (slot / kblocks) + xcd, kb = slot % kblocks;
    const int n0 = by * 128, kbase = kb * 256;
    const int wn = wave >> 1, wk = wave & 1;
    const bool has_act = scale != nullptr;
    const long ntiles = (M + W1_TILE - 1) / W1_TILE;
    const long tile0 = (long)split * tiles_per_split;
    const long tile1 = min(tile0 + tiles_per_split, ntiles);

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging roles: position block lane & 7 everywhere; X: channel quads 8 (wave + 4 j) + (lane >> 3), j = 0, 1;
    // dY: channel quad 8 wave + (lane >> 3)
    const int pb = lane & 7, ql = lane >> 3;
    const int qy = 8 * wave + ql;
    int qx[2];
    bool kok[2];
    float4 sc4[2], sh4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        qx[j] = 8 * (wave + 4 * j) + ql;
        kok[j] = kbase + 4 * qx[j] < K;                        // 4 | K: a quad is in or out as a whole
        const int kld = kok[j] ? kbase + 4 * qx[j] : 0;
        sc4[j] = make_float4(1.f, 1.f, 1.f, 1.f);
        sh4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act) { sc4[j] = *reinterpret_cast<const float4*>(scale + kld); sh4[j] = *reinterpret_cast<const float4*>(shift + kld); }
    }
    float4 ry[4], rx[2][4];
    auto fetch = [&](long tile) {
        const long m0 = tile * W1_TILE;
        const long left = M - m0;
        const int rows = (int)(left < W1_TILE ? left : W1_TILE);
        // one buffer resource per tile and operand: rows past M read as 0 (BUFFER loads: free beside MFMA waves)
        const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(dY + m0 * lddy), 0, (int)(((long)(rows - 1) * lddy + N) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(X + m0 * ldx), 0, (int)(((long)(rows - 1) * ldx + K) * 4), 0x00020000);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long row = 4 * pb + e;
            ry[e] = bufld4(rY, (int)((row * lddy + n0 + 4 * qy) * 4), 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                rx[j][e] = bufld4(rX, (int)((row * ldx + (kok[j] ? kbase + 4 * qx[j] : 0)) * 4), 0);
        }
    };
    auto stash = [&](long tile) {
        const long m0 = tile * W1_TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = rx[j][e];
                if (has_act)
                    v[e] = make_float4(act1(v[e].x, sc4[j].x, sh4[j].x), act1(v[e].y, sc4[j].y, sh4[j].y),
                                       act1(v[e].z, sc4[j].z, sh4[j].z), act1(v[e].w, sc4[j].w, sh4[j].w));
                if (!kok[j] || m0 + 4 * pb + e >= M) v[e] = make_float4(0.f, 0.f, 0.f, 0.f);   // act(0) != 0
            }
            float* dx = Xt + (4 * qx[j]) * W1_P + 4 * pb;
            *reinterpret_cast<float4*>(dx) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            *reinterpret_cast<float4*>(dx + W1_P) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            *reinterpret_cast<float4*>(dx + 2 * W1_P) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
            *reinterpret_cast<float4*>(dx + 3 * W1_P) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
        float* dy = Yt + (4 * qy) * W1_P + 4 * pb;
        *reinterpret_cast<float4*>(dy) = make_float4(ry[0].x, ry[1].x, ry[2].x, ry[3].x);
        *reinterpret_cast<float4*>(dy + W1_P) = make_float4(ry[0].y, ry[1].y, ry[2].y, ry[3].y);
        *reinterpret_cast<float4*>(dy + 2 * W1_P) = make_float4(ry[0].z, ry[1].z, ry[2].z, ry[3].z);
        *reinterpret_cast<float4*>(dy + 3 * W1_P) = make_float4(ry[0].w, ry[1].w, ry[2].w, ry[3].w);
    };

    const float* pa = Yt + (64 * wn + i) * W1_P + 4 * h;          // + 32 W1_P: the second n sub-tile
    const float* pbk = Xt + (128 * wk + i) * W1_P + 4 * h;        // + 32 b W1_P: k sub-tile b
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        __syncthreads();                           // the previous tile's fragment reads are done
        stash(tile);
        __syncthreads();
        fetch(tile + 1 < tile1 ? tile + 1 : tile); // branch-free; the last one is a harmless re-read
        asm volatile("" ::: "memory");             // keep the prefetch in front of the multiply
#pragma unroll
        for (int g = 0; g < W1_TILE / 8; ++g) {
            const float4 a0 = lds4(pa + 8 * g), a1 = lds4(pa + 32 * W1_P + 8 * g);
            const float4 b0 = lds4(pbk + 8 * g), b1 = lds4(pbk + 32 * W1_P + 8 * g);
            const float4 b2 = lds4(pbk + 64 * W1_P + 8 * g), b3 = lds4(pbk + 96 * W1_P + 8 * g);
#define GNX_W1_STEP(c)                                                                        \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc[0][0], 0, 0, 0); \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc[0][1], 0, 0, 0); \
            acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b2.c, acc[0][2], 0, 0, 0); \
            acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b3.c, acc[0][3], 0, 0, 0); \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc[1][0], 0, 0, 0); \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc[1][1], 0, 0, 0); \
            acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b2.c, acc[1][2], 0, 0, 0); \
            acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b3.c, acc[1][3], 0, 0, 0);
            GNX_W1_STEP(x) GNX_W1_STEP(y) GNX_W1_STEP(z) GNX_W1_STEP(w)
#undef GNX_W1_STEP
        }
    }
    // slab[split][n][k] (an empty split writes its zeros)
    float* dst = slabs + (long)split * N * K;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int k = kbase + 128 * wk + 32 * b + i;
        if (k >= K) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wn + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                dst[(long)n * K + k] = acc[a][b][r];
            }
    }
}
__global__ __launch_bounds__(256, 2) void wgrad1_t_kernel(const float* __restrict__ dY, long lddy,
                                                          const float* __restrict__ X, long ldx,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ slabs, long M, int N, int K, int nsplit,
                                                          int kblocks, long tiles_per_split) {
    wgrad1_t_body(dY, lddy, X, ldx, scale, shift, slabs, M, N, K, nsplit, kblocks, tiles_per_split, blockIdx.x, blockIdx.y);
}

// ---- several layers' weight gradients as ONE launch (gnx_wgrad_bnrelu_batch).  At a batch of 32 patches a dense layer's
// weight-gradient kernel is 16-64 workgroups of a few tiles each and costs its launch (14-18 us) whatever it computes; the
// layers of a dense block are independent (each has its own dY / bottleneck, all read the same block buffer), so their grids
// are laid end to end: entry e owns flat block ids [first_block, first_block + blocks).  The kernel body, the slab layout and
// the fixed-order reduce are those of the single launches: the results are bit-identical.
struct WgBatchEntry {
    const float* dY; const float* X; const float* scale; const float* shift; float* slabs; float* dW;
    long lddy, ldx, M, tps;
    int N, K, ns, kblocks, first_block, accumulate;
};
constexpr int WG_BATCH = 24;                                   // entries per launch (2.5 KB of kernel arguments)
struct WgBatch { WgBatchEntry e[WG_BATCH]; int n; };
__device__ __forceinline__ int wg_batch_find(const WgBatch& b, int bid) {
    int k = 0;
    while (k + 1 < b.n && bid >= b.e[k + 1].first_block) ++k;  // (uniform: a scalar loop over at most 24 entries)
    return k;
}
__global__ __launch_bounds__(256, 2) void wgrad1_t_batch_kernel(const WgBatch b) {
    const int k = wg_batch_find(b, blockIdx.x);
    const WgBatchEntry& e = b.e[k];
    const int local = blockIdx.x - e.first_block, gx = e.ns * e.kblocks;
    wgrad1_t_body(e.dY, e.lddy, e.X, e.ldx, e.scale, e.shift, e.slabs, e.M, e.N, e.K, e.ns, e.kblocks, e.tps, local % gx, local / gx);
}

// ---- 3x3: dW[tap][n][k], N == 32, 128 | K, maps of S x S with S in {4, 8, 16, 32};  grid (split, 1, K / 128), 256 threads
constexpr int W9_TILE = 32, W9_PX = W9_TILE + WT_PAD;         // 36 = 4 * 9
template <int S>
__device__ __forceinline__ void wgrad9_t_body(const float* __restrict__ dY, long lddy, const float* __restrict__ X, long ldx,
                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                              float* __restrict__ slabs, long M, int K, long tiles_per_split, int bx, int bz) {
    constexpr int NPOS = W9_TILE + 2 * S;         // dY strip: positions m0 - S .. m0 + 32 + S - 1
    constexpr int P2 = NPOS + 8 + WT_PAD;         // + 8 zero columns; 108 / 76 / 60 = 4 * odd
    constexpr int NB = NPOS / 4;                  // position blocks of the strip: 24 / 16 / 12
    __shared__ __attribute__((aligned(16))) float Xt[128 * W9_PX];       // [k][position]
    __shared__ __attribute__((aligned(16))) float Yc[3 * 32 * P2];       // [dx + 1][n][strip position]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const int kbase = bz * 128;
    const bool has_act = scale != nullptr;
    const long ntiles = M / W9_TILE;
    const long tile0 = (long)bx * tiles_per_split;
    const long tile1 = min(tile0 + tiles_per_split, ntiles);

    f32x16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    // staging roles.  X: position block lane & 7, channel quad 8 wave + (lane >> 3).  dY: the first NB * 8 threads, block
    // t % NB of quad t / NB, positions 4 pbY - 1 .. 4 pbY + 4 of the strip (one to each side for the dx = +-1 copies)
    const int pbX = lane & 7, qX = 8 * wave + (lane >> 3);
    const int kX = kbase + 4 * qX;
    float4 scX = make_float4(1.f, 1.f, 1.f, 1.f), shX = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_act) { scX = *reinterpret_cast<const float4*>(scale + kX); shX = *reinterpret_cast<const float4*>(shift + kX); }
    const bool doY = t < NB * 8;
    const int pbY = t % NB, qY = doY ? t / NB : 0;
    float4 rx[4], ry[6];
    auto fetch = [&](long tile) {
        const long m0 = tile * W9_TILE;
#pragma unroll
        for (int e = 0; e < 4; ++e) rx[e] = *reinterpret_cast<const float4*>(X + (m0 + 4 * pbX + e) * ldx + kX);
        if (doY) {
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                long p = m0 - S + 4 * pbY + e - 1;
                p = p < 0 ? 0 : (p < M ? p : M - 1);          // clamped: rows outside [0, M) are zeroed at the stash
                ry[e] = *reinterpret_cast<const float4*>(dY + p * lddy + 4 * qY);
            }
        }
    };
    auto stash = [&](long tile) {
        const long m0 = tile * W9_TILE;
        {
            float4 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = rx[e];
                if (has_act)
                    v[e] = make_float4(act1(v[e].x, scX.x, shX.x), act1(v[e].y, scX.y, shX.y), act1(v[e].z, scX.z, shX.z),
                                       act1(v[e].w, scX.w, shX.w));
            }
            float* d = Xt + (4 * qX) * W9_PX + 4 * pbX;
            *reinterpret_cast<float4*>(d) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
            *reinterpret_cast<float4*>(d + W9_PX) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
            *reinterpret_cast<float4*>(d + 2 * W9_PX) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
            *reinterpret_cast<float4*>(d + 3 * W9_PX) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
        }
        if (doY) {
            float4 v[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const long p = m0 - S + 4 * pbY + e - 1;
                v[e] = (p >= 0 && p < M) ? ry[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // column of strip position 4 pbY + e (S | m0 - S + 4 pbY's row start: S is a power of two, 4 | S)
            const int x0 = (int)((m0 + 4 * pbY) & (S - 1));
#pragma unroll
            for (int c = 0; c < 3; ++c) {                     // copy c holds dY[q - dx], dx = c - 1
                float4 w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int xs = x0 + e - (c - 1);          // source column: must stay inside the row
                    w[e] = (xs >= 0 && xs < S) ? v[e + 1 - (c - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float* d = Yc + (c * 32 + 4 * qY) * P2 + 4 * pbY;
                *reinterpret_cast<float4*>(d) = make_float4(w[0].x, w[1].x, w[2].x, w[3].x);
                *reinterpret_cast<float4*>(d + P2) = make_float4(w[0].y, w[1].y, w[2].y, w[3].y);
                *reinterpret_cast<float4*>(d + 2 * P2) = make_float4(w[0].z, w[1].z, w[2].z, w[3].z);
                *reinterpret_cast<float4*>(d + 3 * P2) = make_float4(w[0].w, w[1].w, w[2].w, w[3].w);
            }
        }
    };

    for (int idx = t; idx < 3 * 32 * 8; idx += 256) Yc[(idx >> 3) * P2 + NPOS + (idx & 7)] = 0.f;       // the zero columns
    const float* pbx = Xt + (32 * wave + i) * W9_PX + 4 * h;
    const float* pay = Yc + i * P2 + 4 * h;                   // + strip position (tile position 0 with dy = 0: S)
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        __syncthreads();
        stash(tile);
        __syncthreads();
        fetch(tile + 1 < tile1 ? tile + 1 : tile);
        asm volatile("" ::: "memory");
        // A group of 8 positions lies in one image row (S >= 8).  In the top row the taps with dy = +1 have no source row, in
        // the bottom row those with dy = -1: their fragment reads are pointed at 8 zero columns kept at the end of every
        // strip row (a scalar select on the read offset - no branch, no select on the data, nothing in the matrix stream
        // depends on it; the 12 MFMAs then add exact zeros: 2 / (3 S) of the work).  A branch around those MFMAs instead
        // kept the compiler from hoisting the next group's reads over the current MFMAs: 97 vs 128 TFLOP/s at S = 16.
        const long m0 = tile * W9_TILE;
#pragma unroll
        for (int g = 0; g < W9_TILE / 8; ++g) {
            // image row of this lane half's 4 positions (S >= 8: the same for both halves, i.e. wave-uniform)
            const int yrow = (int)(((m0 + 8 * g + (S < 8 ? 4 * h : 0)) / S) & (S - 1));
            const float4 b = lds4(pbx + 8 * g);
#pragma unroll
            for (int dyi = 0; dyi < 3; ++dyi) {               // dy = dyi - 1: needs 0 <= yrow - dy < S
                const bool none = (dyi == 2 && yrow == 0) || (dyi == 0 && yrow == S - 1);
                const int j0 = none ? NPOS : S + 8 * g - (dyi - 1) * S;
                const float* base = pay + j0;
                const float4 am = lds4(base), a0 = lds4(base + 32 * P2), ap = lds4(base + 64 * P2);   // dx = -1, 0, +1
                f32x16& cm = acc[3 * dyi], &c0 = acc[3 * dyi + 1], &cp = acc[3 * dyi + 2];
#define GNX_W9_STEP(c)                                                           \
                cm = __builtin_amdgcn_mfma_f32_32x32x2f32(am.c, b.c, cm, 0, 0, 0); \
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b.c, c0, 0, 0, 0); \
                cp = __builtin_amdgcn_mfma_f32_32x32x2f32(ap.c, b.c, cp, 0, 0, 0);
                GNX_W9_STEP(x) GNX_W9_STEP(y) GNX_W9_STEP(z) GNX_W9_STEP(w)
#undef GNX_W9_STEP
            }
        }
    }
    const int k = kbase + 32 * wave + i;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float* dst = slabs + ((long)bx * 9 + tap) * 32 * K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[(long)n * K + k] = acc[tap][r];
        }
    }
}
template <int S>
__global__ __launch_bounds__(256, 2) void wgrad9_t_kernel(const float* __restrict__ dY, long lddy,
                                                          const float* __restrict__ X, long ldx,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ slabs, long M, int K, long tiles_per_split) {
    wgrad9_t_body<S>(dY, lddy, X, ldx, scale, shift, slabs, M, K, tiles_per_split, blockIdx.x, blockIdx.z);
}
// (entry e: ns splits x kblocks = K / 128 channel blocks, laid out split-fastest)
template <int S>
__global__ __launch_bounds__(256, 2) void wgrad9_t_batch_kernel(const WgBatch b) {
    const int k = wg_batch_find(b, blockIdx.x);
    const WgBatchEntry& e = b.e[k];
    const int local = blockIdx.x - e.first_block;
    wgrad9_t_body<S>(e.dY, e.lddy, e.X, e.ldx, e.scale, e.shift, e.slabs, e.M, e.K, e.tps, local % e.ns, local / e.ns);
}

// dW (torch layout [N][K][T]) = fixed-order sum over splits of slab[split][tap][n][k]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int nsplit, int T, int N, int K,
                                    float* __restrict__ dW, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)T * N * K;
    if (idx >= total) return;
    const int k = (int)(idx % K), n = (int)((idx / K) % N), tap = (int)(idx / ((long)K * N));
    // four independent partial sums keep several loads in flight; combined in a fixed order
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sp = 0;
    for (; sp + 15 < nsplit; sp += 16) {                   // the same four chains, 16 loads in flight
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(sp + u) * total + idx];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
    }
    for (; sp + 3 < nsplit; sp += 4) {
        s0 += slabs[(long)sp * total + idx];
        s1 += slabs[(long)(sp + 1) * total + idx];
        s2 += slabs[(long)(sp + 2) * total + idx];
        s3 += slabs[(long)(sp + 3) * total + idx];
    }
    for (; sp < nsplit; ++sp) s0 += slabs[(long)sp * total + idx];
    const float s = (s0 + s1) + (s2 + s3);
    float* dst = dW + ((long)n * K + k) * T + tap;
    *dst = accumulate ? *dst + s : s;
}

// the same reduction for the entries of a batch: blockIdx.y = entry, grid-stride over its T * N * K elements
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const WgBatch b, int T) {
    const WgBatchEntry& e = b.e[blockIdx.y];
    const long total = (long)T * e.N * e.K;
    const int nsplit = e.ns, K = e.K, N = e.N;
    const float* __restrict__ slabs = e.slabs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int k = (int)(idx % K), n = (int)((idx / K) % N), tap = (int)(idx / ((long)K * N));
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;         // (the chains and their order: wgrad_reduce_kernel's)
        int sp = 0;
        for (; sp + 15 < nsplit; sp += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(sp + u) * total + idx];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
        }
        for (; sp + 3 < nsplit; sp += 4) {
            s0 += slabs[(long)sp * total + idx];
            s1 += slabs[(long)(sp + 1) * total + idx];
            s2 += slabs[(long)(sp + 2) * total + idx];
            s3 += slabs[(long)(sp + 3) * total + idx];
        }
        for (; sp < nsplit; ++sp) s0 += slabs[(long)sp * total + idx];
        const float s = (s0 + s1) + (s2 + s3);
        float* dst = e.dW + ((long)n * K + k) * T + tap;
        *dst = e.accumulate ? *dst + s : s;
    }
}

__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * K) return;
    const int k = (int)(idx % K), n = (int)(idx / K);
    wt[(long)k * N + n] = w[idx];
}

// W[N][K][3][3] -> Wb[tap'][K][N] with tap' = 8 - tap (the adjoint of a pad-1 cross-correlation)
__global__ void repack3x3_bwd_kernel(const float* __restrict__ w, float* __restrict__ wb, int N, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)9 * N * K;
    if (idx >= total) return;
    const int tap = (int)(idx % 9), k = (int)((idx / 9) % K), n = (int)(idx / (9L * K));
    wb[((long)(8 - tap) * K + k) * N + n] = w[idx];
}

// ---- all of a network's weight re-layouts of one kind in ONE launch.  A training step re-lays every conv weight out after
// every optimizer step (conv2 -> [tap][N][K] for the forward, -> [8 - tap][K][N] for the data gradient, conv1 -> [K][N] for
// its data gradient): 3 x 58 launches of ~4 us for DenseNet-121 - at batch 32 a tenth of the step's launches.
// table[l] = {src, dst, N, K} lives in device memory (built once per model: the tensors persist); kind: 0 = conv2 forward,
// 1 = conv2 data gradient, 2 = transpose.  grid (blocks, layers).
struct RelayoutEntry { const float* src; float* dst; int N, K; };
__global__ __launch_bounds__(256) void relayout_batch_kernel(const RelayoutEntry* __restrict__ table, int kind) {
    const RelayoutEntry e = table[blockIdx.y];
    const long total = (kind == 2 ? 1L : 9L) * e.N * e.K;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        if (kind == 0) {                                   // dst[tap][n][k] = src[n][k][tap]
            const int k = (int)(idx % e.K), n = (int)((idx / e.K) % e.N), tap = (int)(idx / ((long)e.K * e.N));
            e.dst[idx] = e.src[((long)n * e.K + k) * 9 + tap];
        } else if (kind == 1) {                            // dst[8 - tap][k][n] = src[n][k][tap]
            const int tap = (int)(idx % 9), k = (int)((idx / 9) % e.K), n = (int)(idx / (9L * e.K));
            e.dst[((long)(8 - tap) * e.K + k) * e.N + n] = e.src[idx];
        } else {                                           // dst[k][n] = src[n][k]
            const int k = (int)(idx % e.K), n = (int)(idx / e.K);
            e.dst[(long)k * e.N + n] = e.src[idx];
        }
    }
}

// out[img*S2 + p][c] = in[img][c] * alpha
__global__ void rows_broadcast_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out, long ldo,
                                      long rows, int C, int S2, float alpha) {
    const long total = rows * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        out[r * ldo + c] = in[(r / S2) * ldi + c] * alpha;
    }
}

// dA[(img, 2oy+dy, 2ox+dx)][c] = 0.25 * dP[(img, oy, ox)][c]; positions not covered by a window (odd S) get 0
__global__ void avgpool2_bwd_kernel(const float* __restrict__ dP, long ldp, float* __restrict__ dA, long lda,
                                    long Min, int C, int S) {
    const long total = Min * C;
    const int So = S >> 1;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        const long img = r / ((long)S * S);
        const int rem = (int)(r - img * S * S);
        const int y = rem / S, x = rem - y * S;
        float v = 0.f;
        if ((y >> 1) < So && (x >> 1) < So) v = 0.25f * dP[((img * So + (y >> 1)) * So + (x >> 1)) * ldp + c];
        dA[r * lda + c] = v;
    }
}

// adjoint of out = maxpool3x3s2p1(relu(in*scale+shift)) with respect to the ACTIVATED input:
// dAct[pos][c] = sum over the (<=4) windows containing pos of dOut[window] * [act(pos) is that window's maximum and > 0]
// Ties: torch's max_pool2d sends a window's gradient to ONE element, the first maximal one in its row-major scan of the
// window (`val > maxval` while scanning); constant regions - white slide background, all-zero background spots whose
// conv0 map is the constant relu(shift) - tie everywhere, so "every element equal to the maximum" is not the same function.
// pool_first_max: no element of window (oy, ox) that the scan visits before (y, x) has the value a (= the window's maximum).
__device__ __forceinline__ bool pool_first_max(const float* __restrict__ in, long ldi, long img, int Hi, int Wi, int oy,
                                               int ox, int y, int x, int c, float sc, float sh, float a) {
    const int y0 = 2 * oy - 1 < 0 ? 0 : 2 * oy - 1;
    const int x0 = 2 * ox - 1 < 0 ? 0 : 2 * ox - 1, x1 = 2 * ox + 1 >= Wi ? Wi - 1 : 2 * ox + 1;
    for (int iy = y0; iy <= y; ++iy) {
        const int xe = iy == y ? x - 1 : x1;
        for (int ix = x0; ix <= xe; ++ix) {
            const long r = (img * Hi + iy) * (long)Wi + ix;
            if (fmaxf(fmaf(in[r * ldi + c], sc, sh), 0.f) == a) return false;
        }
    }
    return true;
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ in, long ldi, const float* __restrict__ pooled, long ldp,
                                   const float* __restrict__ dOut, long lddo, float* __restrict__ dAct, long lda,
                                   long Min, int C, int Hi, int Wi, int Ho, int Wo, const float* __restrict__ scale,
                                   const float* __restrict__ shift) {
    const long total = Min * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C;
        const int c = (int)(idx - r * C);
        const long img = r / ((long)Hi * Wi);
        const int rem = (int)(r - img * Hi * Wi);
        const int y = rem / Wi, x = rem - y * Wi;
        const float a = fmaxf(fmaf(in[r * ldi + c], scale[c], shift[c]), 0.f);
        float g = 0.f;
        if (a > 0.f) {
            // windows: oy with 2oy-1 <= y <= 2oy+1
            for (int oy = (y) / 2; oy <= (y + 1) / 2; ++oy) {
                if (oy < 0 || oy >= Ho) continue;
                for (int ox = (x) / 2; ox <= (x + 1) / 2; ++ox) {
                    if (ox < 0 || ox >= Wo) continue;
                    const long o = (img * Ho + oy) * Wo + ox;
                    if (pooled[o * ldp + c] == a && pool_first_max(in, ldi, img, Hi, Wi, oy, ox, y, x, c, scale[c], shift[c], a))
                        g += dOut[o * lddo + c];
                }
            }
        }
        dAct[r * lda + c] = g;
    }
}

// 4 channels per thread, 16-B accesses (4 | C, aligned pointers and leading dimensions): same arithmetic, a quarter of the
// index math and memory instructions
__global__ __launch_bounds__(256) void avgpool2_bwd_vec4_kernel(const float* __restrict__ dP, long ldp,
                                                                float* __restrict__ dA, long lda, long Min, int C4,
                                                                int S) {
    const long total = Min * C4;
    const int So = S >> 1;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C4;
        const int c = 4 * (int)(idx - r * C4);
        const long img = r / ((long)S * S);
        const int rem = (int)(r - img * S * S);
        const int y = rem / S, x = rem - y * S;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((y >> 1) < So && (x >> 1) < So) {
            const float4 d = *reinterpret_cast<const float4*>(dP + ((img * So + (y >> 1)) * So + (x >> 1)) * ldp + c);
            v = make_float4(0.25f * d.x, 0.25f * d.y, 0.25f * d.z, 0.25f * d.w);
        }
        *reinterpret_cast<float4*>(dA + r * lda + c) = v;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_vec4_kernel(const float* __restrict__ in, long ldi,
                                                               const float* __restrict__ pooled, long ldp,
                                                               const float* __restrict__ dOut, long lddo,
                                                               float* __restrict__ dAct, long lda, long Min, int C4,
                                                               int Hi, int Wi, int Ho, int Wo,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
    const long total = Min * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C4;
        const int c = 4 * (int)(idx - r * C4);
        const long img = r / ((long)Hi * Wi);
        const int rem = (int)(r - img * Hi * Wi);
        const int y = rem / Wi, x = rem - y * Wi;
        const float4 xi = *reinterpret_cast<const float4*>(in + r * ldi + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
        const float a0 = fmaxf(fmaf(xi.x, sc.x, sh.x), 0.f), a1 = fmaxf(fmaf(xi.y, sc.y, sh.y), 0.f);
        const float a2 = fmaxf(fmaf(xi.z, sc.z, sh.z), 0.f), a3 = fmaxf(fmaf(xi.w, sc.w, sh.w), 0.f);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows: oy with 2oy-1 <= y <= 2oy+1 (same visiting order as the scalar kernel: oy outer, ox inner)
        for (int oy = y / 2; oy <= (y + 1) / 2; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = x / 2; ox <= (x + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const long o = (img * Ho + oy) * Wo + ox;
                const float4 p = *reinterpret_cast<const float4*>(pooled + o * ldp + c);
                const float4 d = *reinterpret_cast<const float4*>(dOut + o * lddo + c);
                if (a0 > 0.f && p.x == a0 && pool_first_max(in, ldi, img, Hi, Wi, oy, ox, y, x, c, sc.x, sh.x, a0)) g.x += d.x;
                if (a1 > 0.f && p.y == a1 && pool_first_max(in, ldi, img, Hi, Wi, oy, ox, y, x, c + 1, sc.y, sh.y, a1)) g.y += d.y;
                if (a2 > 0.f && p.z == a2 && pool_first_max(in, ldi, img, Hi, Wi, oy, ox, y, x, c + 2, sc.z, sh.z, a2)) g.z += d.z;
                if (a3 > 0.f && p.w == a3 && pool_first_max(in, ldi, img, Hi, Wi, oy, ox, y, x, c + 3, sc.w, sh.w, a3)) g.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(dAct + r * lda + c) = g;
    }
}

// The adjoint by recorded window index (gnx_bnrelu_maxpool_argmax): element (y, x) receives window (oy, ox)'s gradient iff
// that window's recorded maximum is this element.  No activation is recomputed, the conv0 map is not read (5.2 GB per
// 128-px array), ties go where torch sends them.  4 channels per thread.
// BN (template): also the norm0 -> relu0 adjoint for running statistics - dAct becomes the gradient of the conv0 map itself,
//   dPre[e] = scale[c] * sum over the windows whose maximum is e of dOut[w] * [pooled[w] > 0]
// (pooled[w] = relu(bn(pre[e])) for exactly those windows, so the ReLU mask is read off the POOLED map: a quarter of the
// size, already in the block buffer - the 5.2 GB conv0 map of a 128-px array is neither kept nor re-read).
template <bool BN>
__global__ __launch_bounds__(256) void maxpool_bwd_argmax_kernel(const unsigned char* __restrict__ amax,
                                                                 const float* __restrict__ dOut, long lddo,
                                                                 float* __restrict__ dAct, long lda, long Min, int C4,
                                                                 int Hi, int Wi, int Ho, int Wo,
                                                                 const float* __restrict__ pooled, long ldp,
                                                                 const float* __restrict__ scale) {
    const long total = Min * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C4;
        const int c = 4 * (int)(idx - r * C4);
        const long img = r / ((long)Hi * Wi);
        const int rem = (int)(r - img * Hi * Wi);
        const int y = rem / Wi, x = rem - y * Wi;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = y / 2; oy <= (y + 1) / 2; ++oy) {
            if (oy >= Ho) continue;
            for (int ox = x / 2; ox <= (x + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const long o = (img * Ho + oy) * Wo + ox;
                const unsigned li = 3 * (y - 2 * oy + 1) + (x - 2 * ox + 1);
                const unsigned am = *reinterpret_cast<const unsigned*>(amax + o * (4L * C4) + c);
                float4 d = *reinterpret_cast<const float4*>(dOut + o * lddo + c);
                if (BN) {
                    const float4 a = *reinterpret_cast<const float4*>(pooled + o * ldp + c);
                    d = make_float4(a.x > 0.f ? d.x : 0.f, a.y > 0.f ? d.y : 0.f, a.z > 0.f ? d.z : 0.f, a.w > 0.f ? d.w : 0.f);
                }
                if ((am & 0xffu) == li) g.x += d.x;
                if (((am >> 8) & 0xffu) == li) g.y += d.y;
                if (((am >> 16) & 0xffu) == li) g.z += d.z;
                if ((am >> 24) == li) g.w += d.w;
            }
        }
        if (BN) {
            const float4 sc = *reinterpret_cast<const float4*>(scale + c);
            g = make_float4(sc.x * g.x, sc.y * g.y, sc.z * g.z, sc.w * g.w);
        }
        *reinterpret_cast<float4*>(dAct + r * lda + c) = g;
    }
}

// ---- stem conv weight gradient: dW[o][c][ky][kx] = sum_{img,oy,ox} dS[(img,oy,ox)][o] * x[img][c][oy*st+ky-pad][ox*st+kx-pad]
// Workgroup = 8x16 output tile; the input patch is staged like the forward; wave w reduces positions 32w..32w+31 of the
// tile into 2 (o-tiles) x CIN x 2 (k-tiles of 32 over ky*8+kx) accumulators and the four waves write separate slabs.
// FAST (7x7 s2 p3 stem, 4 | W, O == 64, whole 8x16 tiles, 16-B aligned operands): the patch rows and the dS tile are
// fetched as 16-B pieces one tile AHEAD into registers while the current tile multiplies (the plain form stages 42 scalar
// loads per thread synchronously: 47 us per tile against 5 us of MFMA work).  The patch then starts at the 16-B aligned
// column 32 tx - 4 (one left of ix0 = 32 tx - 3) and is 44 floats wide.
template <int STRIDE, int KH, int CIN, bool FAST = false>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dS,
                                                          long ldd, float* __restrict__ slabs, int H, int Wd, int Ho,
                                                          int Wo, int O, int KW, int pad, int tiles_x, int tiles_y,
                                                          long ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PH = 7 * STRIDE + KH, PW = FAST ? 44 : (15 * STRIDE + 8 + 1) & ~1;
    constexpr int KT = KH * 8;                      // k index inside a channel: ky*8 + kx
    constexpr int NKT = (KT + 31) / 32;
    float* Ps = lds;                                // [CIN][PH][PW] (+ slack row for the padded k lanes)
    float* Ds = lds + CIN * PH * PW + 64;           // [128][64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    f32x16 acc[2][CIN][NKT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int q = 0; q < NKT; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][q][r] = 0.f;
    // this lane's k = 32*q + i -> (ky, kx); lanes past KT read a harmless in-range address and are dropped at the store
    int koff[NKT];
#pragma unroll
    for (int q = 0; q < NKT; ++q) {
        const int k = 32 * q + i;
        const int ky = k < KT ? k >> 3 : 0, kx = k < KT ? k & 7 : 0;
        koff[q] = ky * PW + kx + (FAST ? 1 : 0);
    }
    float4 rp[3], rd[8];                            // FAST: next tile's patch pieces and dS pieces
    auto tile_geo = [&](long tile, long& img, int& oy0, int& ox0) {
        img = tile / ((long)tiles_x * tiles_y);
        const int trem = (int)(tile - img * tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        oy0 = ty * 8;
        ox0 = tx * 16;
    };
    auto fetch = [&](long tile) {
        long img; int oy0, ox0;
        tile_geo(tile, img, oy0, ox0);
        const int iy0 = oy0 * STRIDE - pad, ixb0 = ox0 * STRIDE - 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int item = t + 256 * j;
            const int rowc = item / 11, j4 = item - rowc * 11;
            const int c = rowc / PH, py = rowc - c * PH;
            const int iy = iy0 + py, ixb = ixb0 + 4 * j4;
            rp[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < CIN * PH * 11 && iy >= 0 && iy < H && ixb >= 0 && ixb + 3 < Wd)
                rp[j] = *reinterpret_cast<const float4*>(x + ((img * CIN + c) * H + iy) * (long)Wd + ixb);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int item = t + 256 * j;
            const int r = item >> 4, o4 = item & 15;
            rd[j] = *reinterpret_cast<const float4*>(dS + ((img * Ho + oy0 + (r >> 4)) * (long)Wo + ox0 + (r & 15)) * ldd + 4 * o4);
        }
    };
    if (FAST) fetch(blockIdx.x);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long img = tile / ((long)tiles_x * tiles_y);
        const int trem = (int)(tile - img * tiles_x * tiles_y);
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int oy0 = ty * 8, ox0 = tx * 16;
        const int iy0 = oy0 * STRIDE - pad, ix0 = ox0 * STRIDE - pad;
        __syncthreads();
        if (FAST) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int item = t + 256 * j;
                if (item < CIN * PH * 11) *reinterpret_cast<float4*>(&Ps[4 * item]) = rp[j];       // [row][11 pieces]
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(&Ds[4 * (t + 256 * j)]) = rd[j];
            __syncthreads();
            const long nxt = tile + gridDim.x;
            fetch(nxt < ntiles ? nxt : tile);          // branch-free; the last one is a harmless re-read
            asm volatile("" ::: "memory");             // keep the prefetch in front of the multiply
        } else {
        for (int idx = t; idx < CIN * PH * PW; idx += 256) {
            const int px = idx % PW, py = (idx / PW) % PH, c = idx / (PW * PH);
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < Wd) v = x[((img * CIN + c) * H + iy) * (long)Wd + ix];
            Ps[idx] = v;
        }
        for (int idx = t; idx < 128 * 64; idx += 256) {
            const int r = idx >> 6, o = idx & 63;
            const int oy = oy0 + (r >> 4), ox = ox0 + (r & 15);
            float v = 0.f;
            if (oy < Ho && ox < Wo && o < O) v = dS[((img * Ho + oy) * (long)Wo + ox) * ldd + o];
            Ds[idx] = v;
        }
        __syncthreads();
        }
        for (int mm = 0; mm < 32; mm += 2) {
            const int r = 32 * wave + mm + h;            // tile position handled by this lane half
            const int poff = (STRIDE * (r >> 4)) * PW + STRIDE * (r & 15);
            const float a0 = Ds[r * 64 + i], a1 = Ds[r * 64 + 32 + i];
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int q = 0; q < NKT; ++q) {
                    const float b = Ps[c * PH * PW + poff + koff[q]];
                    acc[0][c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0][c][q], 0, 0, 0);
                    acc[1][c][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1][c][q], 0, 0, 0);
                }
        }
    }
    // slab[(block*4 + wave)][o][c][ky][kx]  (torch weight layout, KW columns)
    float* dst = slabs + ((long)blockIdx.x * 4 + wave) * O * CIN * KH * KW;
#pragma unroll
    for (int q = 0; q < NKT; ++q) {
        const int k = 32 * q + i;
        const int ky = k >> 3, kx = k & 7;
        if (k >= KT || kx >= KW) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < CIN; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (o < O) dst[(((long)o * CIN + c) * KH + ky) * KW + kx] = acc[a][c][q][r];
                }
    }
}

// out[idx] (+)= sum over the slabs, fixed order: 64 columns x 16 slab lanes per workgroup, a lane sums its contiguous
// sixteenth in index order (16 loads in flight), the sixteenths are combined in index order.  (One lane per column walking
// 2048 slabs took 0.5 ms - a thirtieth of a batch-32 DenseNet training step.)
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ slabs, int nslab, long n,
                                                        float* __restrict__ out, int accumulate) {
    __shared__ float part[16][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + cl;
    float s = 0.f;
    if (idx < n) {
        const int per = (nslab + 15) / 16;
        int b = sl * per;
        const int be = b + per < nslab ? b + per : nslab;
        for (; b + 16 <= be; b += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slabs[(long)(b + u) * n + idx];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        for (; b < be; ++b) s += slabs[(long)b * n + idx];
    }
    part[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && idx < n) {
        float tot = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) tot += part[u][cl];
        out[idx] = accumulate ? out[idx] + tot : tot;
    }
}

bool al16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int ew_grid(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
int wgrad_splits(long M, int N, int K) {
    const long ntiles = (M + WG_BM - 1) / WG_BM;
    const long per = (long)gnx_cdiv(N, 128) * gnx_cdiv(K, WG_KR);
    long s = 1024 / (per > 0 ? per : 1);
    if (s < 1) s = 1;
    if (s > ntiles) s = ntiles;
    if (s > 512) s = 512;
    return (int)s;
}

// position splits of the round-2 1x1 kernel: a multiple of 8 (its block ids encode the XCD), about 512 workgroups in all
// (two per CU) over its ceil(K / 256) k blocks, never more than there are 32-position tiles; 0 = the shape is not its own
int wgrad1_t_splits(long M, int N, int K) {
    if (K % 4 != 0 || K <= 128 || N % 128 != 0) return 0;
    const long nt = (M + W1_TILE - 1) / W1_TILE;
    long s = 512 / ((long)gnx_cdiv(K, 256) * (N / 128));
    if (s > nt) s = nt;
    s &= ~7L;
    return (int)s;
}

}  // namespace

// floats of slab workspace for gnx_wgrad_bnrelu
GNX_EXPORT long gnx_wgrad_workspace(long M, int N, int K, int taps) {
    long s = wgrad_splits(M, N, K);
    if (taps == 1 && wgrad1_t_splits(M, N, K) > s) s = wgrad1_t_splits(M, N, K);
    return s * taps * N * K;
}

// dW[n][k][tap] (= torch [N][K][kh][kw], taps = 1 or 9) of out = conv(act(X)) given dY = d out.
//   taps=9: X is [M = imgs*S*S][K]; taps=1, pool=0: plain 1x1; taps=1, pool=1: X is on the S x S grid and M counts the
//   (S/2)^2 pooled positions (the transition).  scale/shift NULL = no activation.
GNX_EXPORT int gnx_wgrad_bnrelu(const float* dY, long lddy, const float* X, long ldx, const float* scale,
                                const float* shift, float* dW, float* workspace, long M, int N, int K, int S, int taps,
                                int pool, int accumulate, hipStream_t stream) {
    if (!dY || !X || !dW || !workspace || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K || (taps != 1 && taps != 9) ||
        (!scale) != (!shift) || (taps == 9 && (S <= 0 || M % ((long)S * S) != 0 || pool)) || (pool && S < 2))
        return GNX_ERR_BAD_ARG;
    const int nsplit = wgrad_splits(M, N, K);
    const long ntiles = (M + WG_BM - 1) / WG_BM;
    const long tps = (ntiles + nsplit - 1) / nsplit;
    // round 2: transposed-image kernels for the dense layers' own shapes
    const bool aligned = al16b(X) && al16b(dY) && ldx % 4 == 0 && lddy % 4 == 0 && (!scale || (al16b(scale) && al16b(shift)));
    if (aligned && !pool) {
        const long total = (long)taps * N * K;
        // 128 < K: the 128 x 256 workgroup (K <= 128 would leave half of it multiplying zeros: the r1 kernel keeps those)
        const int ns1 = taps == 1 ? wgrad1_t_splits(M, N, K) : 0;
        if (ns1 > 0 && 64L * (lddy > ldx ? lddy : ldx) < (1L << 28)) {
            const long nt1 = (M + W1_TILE - 1) / W1_TILE;
            const int kblocks = gnx_cdiv(K, 256);
            dim3 grid1(ns1 * kblocks, N / 128);
            wgrad1_t_kernel<<<grid1, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, N, K, ns1, kblocks,
                                                       (nt1 + ns1 - 1) / ns1);
            wgrad_reduce_kernel<<<gnx_cdiv(total, 256), 256, 0, stream>>>(workspace, ns1, 1, N, K, dW, accumulate);
            return gnx_launch_status();
        }
        if (taps == 9 && N == 32 && K % 128 == 0 && (S == 4 || S == 8 || S == 16 || S == 32) && M % W9_TILE == 0) {
            const long nt9 = M / W9_TILE;
            dim3 grid9(nsplit, 1, K / 128);
            const long tps9 = (nt9 + nsplit - 1) / nsplit;
            if (S == 32) wgrad9_t_kernel<32><<<grid9, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, K, tps9);
            else if (S == 16) wgrad9_t_kernel<16><<<grid9, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, K, tps9);
            else if (S == 8) wgrad9_t_kernel<8><<<grid9, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, K, tps9);
            else wgrad9_t_kernel<4><<<grid9, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, K, tps9);
            wgrad_reduce_kernel<<<gnx_cdiv(total, 256), 256, 0, stream>>>(workspace, nsplit, 9, N, K, dW, accumulate);
            return gnx_launch_status();
        }
    }
    const int halo = taps == 9 ? S + 1 : 0;
    const int nt = taps == 9 ? 1 : 4;
    const size_t lds_bytes = ((size_t)WG_BM * 32 * nt + (size_t)(WG_BM + 2 * halo) * LDX + WG_BM) * sizeof(float);
    if (lds_bytes > 160 * 1024) return GNX_ERR_UNSUPPORTED;
    const int vec = al16b(X) && ldx % 4 == 0 && (!scale || (al16b(scale) && al16b(shift)));
    const int vecY = al16b(dY) && lddy % 4 == 0;
    dim3 grid(nsplit, gnx_cdiv(N, 32 * nt), gnx_cdiv(K, WG_KR));
#define GNX_WG(T, NTT, P)                                                                                               \
    do {                                                                                                           \
        static size_t conf = 0;                                                                                    \
        if (lds_bytes > conf) {                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<T, NTT, P>),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)     \
                return GNX_ERR_LAUNCH;                                                                             \
            conf = lds_bytes;                                                                                      \
        }                                                                                                          \
        wgrad_kernel<T, NTT, P><<<grid, 256, lds_bytes, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, N, K, S, \
                                                             tps, vec, vecY);                                      \
    } while (0)
    // register-prefetched form where the shape allows (see wgrad_pf_kernel)
    const bool pf = !pool && vec && vecY && K % 4 == 0 && N % (32 * nt) == 0;
#define GNX_WGPF(T, NTT, NXX)                                                                                         \
    do {                                                                                                              \
        static size_t conf = 0;                                                                                       \
        if (lds_bytes > conf) {                                                                                       \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_pf_kernel<T, NTT, NXX>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)        \
                return GNX_ERR_LAUNCH;                                                                                \
            conf = lds_bytes;                                                                                         \
        }                                                                                                             \
        wgrad_pf_kernel<T, NTT, NXX><<<grid, 256, lds_bytes, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M,  \
                                                                       N, K, S, tps);                                 \
    } while (0)
    // 3x3: the 144 accumulator registers + a prefetched strip (68 more) leave one wave per SIMD - measured slower (56 vs
    // 51 ms per step) than the plain kernel at two workgroups per CU: not instantiated
    if (pf && taps == 1) GNX_WGPF(1, 4, 8);
    else if (taps == 9) GNX_WG(9, 1, false);
    else if (pool) GNX_WG(1, 4, true);
    else GNX_WG(1, 4, false);
#undef GNX_WGPF
#undef GNX_WG
    const long total = (long)taps * N * K;
    wgrad_reduce_kernel<<<gnx_cdiv(total, 256), 256, 0, stream>>>(workspace, nsplit, taps, N, K, dW, accumulate);
    return gnx_launch_status();
}

// n independent weight gradients of ONE kind (taps = 1: plain 1x1; taps = 9: 3x3 on S x S maps, the same S for all) as one
// launch per 24 of them plus one batched reduce - exactly gnx_wgrad_bnrelu(item, taps, pool = 0) for every item, bit for
// bit, for the shapes its transposed-image kernels take (the dense layers' own); anything else: GNX_ERR_UNSUPPORTED and
// nothing is launched (make the single calls).  `items`: host array of GnxWgradItem (include/gridnext_hip.h).
struct GnxWgradItem {
    const float* dY; long lddy; const float* X; long ldx; const float* scale; const float* shift; float* dW; float* workspace;
    long M; int N, K, S, accumulate;
};
GNX_EXPORT int gnx_wgrad_bnrelu_batch(const void* items_v, int n, int taps, hipStream_t stream) {
    const GnxWgradItem* it = static_cast<const GnxWgradItem*>(items_v);
    if (!it || n < 0 || (taps != 1 && taps != 9)) return GNX_ERR_BAD_ARG;
    if (n == 0) return GNX_OK;
    // every item must be one the transposed-image kernels take, exactly as gnx_wgrad_bnrelu decides
    for (int q = 0; q < n; ++q) {
        const GnxWgradItem& a = it[q];
        if (!a.dY || !a.X || !a.dW || !a.workspace || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.lddy < a.N || a.ldx < a.K ||
            (!a.scale) != (!a.shift))
            return GNX_ERR_BAD_ARG;
        const bool aligned = al16b(a.X) && al16b(a.dY) && a.ldx % 4 == 0 && a.lddy % 4 == 0 &&
                             (!a.scale || (al16b(a.scale) && al16b(a.shift)));
        if (!aligned) return GNX_ERR_UNSUPPORTED;
        if (taps == 1) {
            if (wgrad1_t_splits(a.M, a.N, a.K) <= 0 || 64L * (a.lddy > a.ldx ? a.lddy : a.ldx) >= (1L << 28)) return GNX_ERR_UNSUPPORTED;
        } else {
            if (a.S != it[0].S || a.N != 32 || a.K % 128 != 0 || !(a.S == 4 || a.S == 8 || a.S == 16 || a.S == 32) ||
                a.M % W9_TILE != 0 || a.M % ((long)a.S * a.S) != 0)
                return GNX_ERR_UNSUPPORTED;
        }
    }
    for (int q0 = 0; q0 < n; q0 += WG_BATCH) {
        WgBatch b;
        b.n = n - q0 < WG_BATCH ? n - q0 : WG_BATCH;
        int blocks = 0;
        long max_total = 0;
        for (int q = 0; q < b.n; ++q) {
            const GnxWgradItem& a = it[q0 + q];
            WgBatchEntry& e = b.e[q];
            e.dY = a.dY; e.X = a.X; e.scale = a.scale; e.shift = a.shift; e.slabs = a.workspace; e.dW = a.dW;
            e.lddy = a.lddy; e.ldx = a.ldx; e.M = a.M; e.N = a.N; e.K = a.K; e.accumulate = a.accumulate;
            e.first_block = blocks;
            if (taps == 1) {
                e.ns = wgrad1_t_splits(a.M, a.N, a.K);
                e.kblocks = gnx_cdiv(a.K, 256);
                const long nt1 = (a.M + W1_TILE - 1) / W1_TILE;
                e.tps = (nt1 + e.ns - 1) / e.ns;
                blocks += e.ns * e.kblocks * (a.N / 128);
            } else {
                e.ns = wgrad_splits(a.M, a.N, a.K);
                e.kblocks = a.K / 128;
                const long nt9 = a.M / W9_TILE;
                e.tps = (nt9 + e.ns - 1) / e.ns;
                blocks += e.ns * e.kblocks;
            }
            const long total = (long)taps * a.N * a.K;
            if (total > max_total) max_total = total;
        }
        if (taps == 1) {
            wgrad1_t_batch_kernel<<<blocks, 256, 0, stream>>>(b);
        } else {
            switch (it[0].S) {
                case 32: wgrad9_t_batch_kernel<32><<<blocks, 256, 0, stream>>>(b); break;
                case 16: wgrad9_t_batch_kernel<16><<<blocks, 256, 0, stream>>>(b); break;
                case 8: wgrad9_t_batch_kernel<8><<<blocks, 256, 0, stream>>>(b); break;
                default: wgrad9_t_batch_kernel<4><<<blocks, 256, 0, stream>>>(b); break;
            }
        }
        int gx = gnx_cdiv(max_total, 256);
        if (gx > 64) gx = 64;
        wgrad_reduce_batch_kernel<<<dim3(gx, b.n), 256, 0, stream>>>(b, taps);
    }
    return gnx_launch_status();
}

GNX_EXPORT int gnx_transpose_weight(const float* w, float* wt, int N, int K, hipStream_t stream) {
    if (!w || !wt || N <= 0 || K <= 0) return GNX_ERR_BAD_ARG;
    transpose_kernel<<<gnx_cdiv((long)N * K, 256), 256, 0, stream>>>(w, wt, N, K);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_repack_conv3x3_bwd(const float* w, float* wb, int N, int K, hipStream_t stream) {
    if (!w || !wb || N <= 0 || K <= 0) return GNX_ERR_BAD_ARG;
    repack3x3_bwd_kernel<<<gnx_cdiv(9L * N * K, 256), 256, 0, stream>>>(w, wb, N, K);
    return gnx_launch_status();
}

// `table`: n entries {const float* src; float* dst; int N; int K;} (24 bytes each) in DEVICE memory; kind 0 / 1 / 2 =
// gnx_repack_conv3x3 / gnx_repack_conv3x3_bwd / gnx_transpose_weight applied to every entry, one launch.
GNX_EXPORT int gnx_relayout_weights_batch(const void* table, int n, int kind, hipStream_t stream) {
    if (!table || n < 0 || kind < 0 || kind > 2 || (reinterpret_cast<uintptr_t>(table) & 7) != 0) return GNX_ERR_BAD_ARG;
    if (n == 0) return GNX_OK;
    if (n > 65535) return GNX_ERR_UNSUPPORTED;
    relayout_batch_kernel<<<dim3(16, n), 256, 0, stream>>>(static_cast<const RelayoutEntry*>(table), kind);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_rows_broadcast(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S2, float alpha,
                                  hipStream_t stream) {
    if (!in || !out || imgs <= 0 || C <= 0 || S2 <= 0 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    rows_broadcast_kernel<<<ew_grid(imgs * S2 * C), 256, 0, stream>>>(in, ldi, out, ldo, imgs * S2, C, S2, alpha);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_avgpool2_bwd(const float* dP, long ldp, float* dA, long lda, long imgs, int C, int S,
                                hipStream_t stream) {
    if (!dP || !dA || imgs <= 0 || C <= 0 || S < 2 || ldp < C || lda < C) return GNX_ERR_BAD_ARG;
    if (C % 4 == 0 && ldp % 4 == 0 && lda % 4 == 0 && al16b(dP) && al16b(dA)) {
        avgpool2_bwd_vec4_kernel<<<ew_grid(imgs * S * S * (C / 4)), 256, 0, stream>>>(dP, ldp, dA, lda, imgs * S * S,
                                                                                      C / 4, S);
        return gnx_launch_status();
    }
    avgpool2_bwd_kernel<<<ew_grid(imgs * S * S * C), 256, 0, stream>>>(dP, ldp, dA, lda, imgs * S * S, C, S);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_maxpool_bwd(const float* in, long ldi, const float* pooled, long ldp, const float* dOut, long lddo,
                               float* dAct, long lda, long imgs, int C, int Hi, int Wi, const float* scale,
                               const float* shift, hipStream_t stream) {
    if (!in || !pooled || !dOut || !dAct || !scale || !shift || imgs <= 0 || C <= 0 || Hi <= 0 || Wi <= 0)
        return GNX_ERR_BAD_ARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    if (C % 4 == 0 && ldi % 4 == 0 && ldp % 4 == 0 && lddo % 4 == 0 && lda % 4 == 0 && al16b(in) && al16b(pooled) &&
        al16b(dOut) && al16b(dAct) && al16b(scale) && al16b(shift)) {
        maxpool_bwd_vec4_kernel<<<ew_grid(imgs * Hi * Wi * (C / 4)), 256, 0, stream>>>(
            in, ldi, pooled, ldp, dOut, lddo, dAct, lda, imgs * Hi * Wi, C / 4, Hi, Wi, Ho, Wo, scale, shift);
        return gnx_launch_status();
    }
    maxpool_bwd_kernel<<<ew_grid(imgs * Hi * Wi * C), 256, 0, stream>>>(in, ldi, pooled, ldp, dOut, lddo, dAct, lda,
                                                                       imgs * Hi * Wi, C, Hi, Wi, Ho, Wo, scale, shift);
    return gnx_launch_status();
}

// dAct [imgs*Hi*Wi][C] (lda) from the window indices recorded by gnx_bnrelu_maxpool_argmax and dOut [imgs*Ho*Wo][C] (lddo).
// The gradient is with respect to the ACTIVATED map (the ReLU mask is applied by the BN adjoint that follows).  4 | C.
GNX_EXPORT int gnx_maxpool_bwd_argmax(const unsigned char* argmax, const float* dOut, long lddo, float* dAct, long lda,
                                      long imgs, int C, int Hi, int Wi, hipStream_t stream) {
    if (!argmax || !dOut || !dAct || imgs <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || lddo < C || lda < C) return GNX_ERR_BAD_ARG;
    if (C % 4 != 0 || lddo % 4 != 0 || lda % 4 != 0 || !al16b(dOut) || !al16b(dAct) ||
        (reinterpret_cast<uintptr_t>(argmax) & 3) != 0)
        return GNX_ERR_UNSUPPORTED;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    maxpool_bwd_argmax_kernel<false><<<ew_grid(imgs * Hi * Wi * (C / 4)), 256, 0, stream>>>(
        argmax, dOut, lddo, dAct, lda, imgs * Hi * Wi, C / 4, Hi, Wi, Ho, Wo, nullptr, 0, nullptr);
    return gnx_launch_status();
}

// The same adjoint carried through norm0 -> relu0 with RUNNING statistics (training.py:126 keeps f in eval mode): dPre
// [imgs*Hi*Wi][C] is the gradient of the conv0 map; `pooled` [imgs*Ho*Wo][C] (ldp) is the pooled ACTIVATED output of the
// forward (the first C columns of block 1's buffer), `scale` the folded gamma / sqrt(var + eps).  The affine gradients of
// norm0 come from gnx_bn_relu_bwd(relu = 2, dx = NULL) on (dOut, pooled).  4 | C, 16-B aligned operands.
GNX_EXPORT int gnx_maxpool_bwd_argmax_bnrelu(const unsigned char* argmax, const float* dOut, long lddo, const float* pooled,
                                             long ldp, const float* scale, float* dPre, long lda, long imgs, int C, int Hi,
                                             int Wi, hipStream_t stream) {
    if (!argmax || !dOut || !pooled || !scale || !dPre || imgs <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || lddo < C || lda < C ||
        ldp < C)
        return GNX_ERR_BAD_ARG;
    if (C % 4 != 0 || lddo % 4 != 0 || lda % 4 != 0 || ldp % 4 != 0 || !al16b(dOut) || !al16b(dPre) || !al16b(pooled) ||
        !al16b(scale) || (reinterpret_cast<uintptr_t>(argmax) & 3) != 0)
        return GNX_ERR_UNSUPPORTED;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    maxpool_bwd_argmax_kernel<true><<<ew_grid(imgs * Hi * Wi * (C / 4)), 256, 0, stream>>>(
        argmax, dOut, lddo, dPre, lda, imgs * Hi * Wi, C / 4, Hi, Wi, Ho, Wo, pooled, ldp, scale);
    return gnx_launch_status();
}

GNX_EXPORT long gnx_conv0_wgrad_workspace(long imgs, int H, int W, int O, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const long ntiles = imgs * gnx_cdiv(Wo, 16) * gnx_cdiv(Ho, 8);
    const long blocks = ntiles < 512 ? ntiles : 512;
    return blocks * 4 * (long)O * 3 * KH * KW;
}

// dW0 [O][3][KH][KW] from x [imgs][3][H][W] and dS [imgs*Ho*Wo][O] (ldd)
GNX_EXPORT int gnx_conv0_wgrad(const float* x, const float* dS, long ldd, float* dW, float* workspace, long imgs, int H,
                               int W, int O, int KH, int KW, int stride, int pad, int accumulate, hipStream_t stream) {
    if (!x || !dS || !dW || !workspace || imgs <= 0 || O <= 0 || O > 64 || ldd < O) return GNX_ERR_BAD_ARG;
    if (!((stride == 2 && KH == 7 && KW == 7) || (stride == 1 && KH == 3 && KW == 3))) return GNX_ERR_UNSUPPORTED;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int tiles_x = gnx_cdiv(Wo, 16), tiles_y = gnx_cdiv(Ho, 8);
    const long ntiles = imgs * tiles_x * tiles_y;
    const int blocks = (int)(ntiles < 512 ? ntiles : 512);
    const int PH = 7 * stride + KH, PW = (15 * stride + 8 + 1) & ~1;
    const size_t lds_bytes = ((size_t)3 * PH * PW + 64 + 128 * 64) * sizeof(float);
    const bool fastld = stride == 2 && pad == 3 && W % 4 == 0 && O == 64 && Ho % 8 == 0 && Wo % 16 == 0 && ldd % 4 == 0 &&
                        al16b(x) && al16b(dS);
    if (fastld) {
        const size_t lds_fast = ((size_t)3 * PH * 44 + 64 + 128 * 64) * sizeof(float);
        conv0_wgrad_kernel<2, 7, 3, true><<<blocks, 256, lds_fast, stream>>>(x, dS, ldd, workspace, H, W, Ho, Wo, O, KW,
                                                                             pad, tiles_x, tiles_y, ntiles);
    } else if (stride == 2)
        conv0_wgrad_kernel<2, 7, 3><<<blocks, 256, lds_bytes, stream>>>(x, dS, ldd, workspace, H, W, Ho, Wo, O, KW, pad,
                                                                        tiles_x, tiles_y, ntiles);
    else
        conv0_wgrad_kernel<1, 3, 3><<<blocks, 256, lds_bytes, stream>>>(x, dS, ldd, workspace, H, W, Ho, Wo, O, KW, pad,
                                                                        tiles_x, tiles_y, ntiles);
    const long n = (long)O * 3 * KH * KW;
    slab_sum_kernel<<<gnx_cdiv(n, 64), 1024, 0, stream>>>(workspace, blocks * 4, n, dW, accumulate);
    return gnx_launch_status();
}
