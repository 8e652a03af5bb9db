// fp16-MFMA backward of the DenseNet dense layers (BASELINE config 5 with f trained: /root/reference/gridnext/densenet.py:35-44
// differentiated by torch.autograd, driven by training.py:164-171 with f_opt; BatchNorm on running statistics, training.py:126).
//
// Everything a dense layer's backward touches lives in HBM as fp16, row-major: the block buffer X16 [M][c_total] (the tape of
// the taped fp16 forward), the block GRADIENT buffer G16 of the same shape, the activated bottleneck A16 [M][128] (the tape:
// conv1's output after norm2 + relu2, as gnx_conv1x1_bnrelu_h16 stores it) and its gradient dB16 [M][128].  Matrix work runs
// on v_mfma_f32_32x32x16_f16 with fp32 accumulation; parameter gradients and BatchNorm sums are fp32.  Gradients carry a
// power-of-two LOSS SCALE `ls` = {s, 1/s} (device floats, chosen per backward by the caller so that the largest gradient
// entering the network sits at 2^12): every fp16 gradient tensor holds s x the true value, every fp32 result is multiplied
// by 1/s when its slabs are reduced.  `flag` (device int) is OR-ed with 1 when a reduced result is not finite (overflow).
//
// Per layer, last to first, with cin = the layer's input channels and dY = G16[:, cin : cin + 32]:
//   gnx_wgrad3x3_f16                    dW2[n][k][ky][kx] = sum_q dY[q][n] A[q + (ky-1, kx-1)][k]
//   gnx_conv3x3_dgrad_bnrelu_bwd_f16    dB = scale2 * conv3x3^T(dY) * [A > 0]; dgamma2, dbeta2 (x_hat recovered from A)
//   gnx_wgrad1x1_f16                    dW1[n][k] = sum_m dB[m][n] relu(bn1(X))[m][k]
//   gnx_conv1x1_dgrad_bnrelu_bwd_f16    G[:, :cin] += scale1 * (dB . W1) * [bn1(X) > 0]; dgamma1, dbeta1
// All four are bound by HBM (8 to 10 bytes per block-buffer element and layer against 2 x 128 flops): a tile's operands go
// global -> registers -> LDS with 16-B accesses, the next tile's loads are in flight while the current one multiplies, two to
// three workgroups share a CU.  Contractions over PIXELS (the weight gradients) read both operands out of row-major
// [pixel][channel] LDS tiles with the transposing read ds_read_b64_tr_b16 (a 4-pixel x 16-channel block per 16 lanes, each
// lane receiving one channel's 4 pixels); contractions over channels read rows (ds_read_b128).  Weights stay in registers as
// MFMA fragments for a workgroup's lifetime (conv2's nine taps: 72 registers; a 128-channel slice of conv1: 32).
// Sums over pixels (weight gradients, BatchNorm adjoint sums) are per-workgroup slabs reduced in a fixed order: deterministic,
// no float atomics.
//
// LAYOUTS (round 5).  The block buffers, their gradients and the activated bottleneck are addressed through (ld, bs): element
// (row, c) lies at row * ld + (c >> 5) * bs + (c & 31).  bs = 32 is the row-major [rows][ld] matrix described above;
// ld = 32, bs = rows_total * 32 is the CHANNEL-BLOCKED form [C / 32][rows_total][32] the fused forward kernel
// (gnx_dense_layer_f16_tape) reads and writes - the gradient path of densenet_train_f16.py runs on that one, so forward and
// backward share buffers; a layer's 32 new channels (its dY) are then one contiguous [rows][32] matrix.
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(8)));
typedef __fp16 fp16x8 __attribute__((__vector_size__(16)));

constexpr int T_RS = 320;   // bytes per row of a [pixel][128 ch] LDS tile read by TRANSPOSING reads: the 64-B segments of 4
                            // consecutive rows that a 32-lane half touches fall on distinct banks (256 + 64)
constexpr int R_RS = 272;   // ... of a tile read by ROWS (ds_read_b128 of 16 consecutive rows: 16 distinct bank quads)
constexpr int Y_RS = 80;    // a 32-channel dY strip row read by rows (64 B + 16)

__device__ __forceinline__ h8 ldg8(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }
__device__ __forceinline__ h8 zero8() {
    h8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
    return z;
}
__device__ __forceinline__ void lds_barrier() {     // orders LDS traffic only: global loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// The 8 contraction elements (pixels 8h .. 8h + 7 of a 16-pixel step) of one channel for a 32x32x16 MFMA operand, out of a
// row-major [pixel][channel] tile: two transposing reads of a 4-row x 16-column block each.  `lo` / `hi`: this lane's address
// for rows 0-3 / 4-7 of its half's 8 pixels = &tile[row 8h + (lane & 15) / 4 (+ 4)][channel base + 16 ((lane >> 4) & 1) +
// 4 (lane & 3)]; the lane receives channel base + (lane & 31).  EXEC must be all ones.
__device__ __forceinline__ h8 tr8(const char* lo, const char* hi) {
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)lo);
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)hi);
    const fp16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h8, v);
}
__device__ __forceinline__ void zero_acc(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// ------------------------------------------------------------------------------------------------ 1x1 weight gradient
// ws[slab][n][k] = sum over the slab's pixels of dY[m][n] act(X[m][k]); 64-pixel tiles; a workgroup owns a 128 x 128 block of
// (n, k) - a wave 64 x 64 - and a contiguous range of tiles.
template <bool PRO>
__global__ __launch_bounds__(256, 2) void wgrad1x1_f16_kernel(const _Float16* __restrict__ dY, long lddy,
                                                              const _Float16* __restrict__ X, long ldx,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ ws, long M, int N, int K, long tiles_per_slab,
                                                              int n_kb, int n_nb, long n_slabs) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * T_RS];
    char* const Pt = smem;
    char* const Qt = smem + 64 * T_RS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    // block order: the n_kb * n_nb workgroups of one slab (they read the same dY / X rows) get ids 8 apart - one XCD, launched
    // together - so the rows they share come out of that XCD's L2 after the first read
    const int nblk = n_kb * n_nb;
    const int bx = blockIdx.x, rem8 = bx % (8 * nblk), kn = rem8 / 8;
    const int kb = kn % n_kb, nb = kn / n_kb;
    const long slab = (long)(bx / (8 * nblk)) * 8 + (rem8 & 7);
    if (slab >= n_slabs) return;                                    // (padding of the last group of 8; whole workgroup)
    const int chunk = t & 15, row0 = t >> 4;
    const int ncol = nb * 128 + chunk * 8, kcol = kb * 128 + chunk * 8;
    const bool nok = ncol < N, kok = kcol < K;
    float sc[8], sh[8];
    if (PRO) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = kok ? scale[kcol + j] : 0.f;
            sh[j] = kok ? shift[kcol + j] : 0.f;
        }
    }
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) zero_acc(acc[a][b]);
    const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    h8 p[4], q[4];
    auto fetch = [&](long tile) {
        const long m0 = tile * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = m0 + row0 + 16 * i;
            const bool ok = row < M;
            p[i] = (ok && nok) ? ldg8(dY + row * lddy + ncol) : zero8();
            q[i] = (ok && kok) ? ldg8(X + row * ldx + kcol) : zero8();
        }
    };
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        lds_barrier();                                              // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (PRO) {      // (rows past M hold zeros in P: whatever the activation makes of a zero there multiplies a zero)
#pragma unroll
                for (int j = 0; j < 8; ++j) q[i][j] = (_Float16)fmaxf(fmaf((float)q[i][j], sc[j], sh[j]), 0.f);
            }
            *reinterpret_cast<h8*>(Pt + (row0 + 16 * i) * T_RS + chunk * 16) = p[i];
            *reinterpret_cast<h8*>(Qt + (row0 + 16 * i) * T_RS + chunk * 16) = q[i];
        }
        lds_barrier();
        if (tile + 1 < tile1) fetch(tile + 1);                      // in flight while this tile multiplies
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h8 a[2], b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* pa = Pt + (16 * ks + trow) * T_RS + (64 * wm + 32 * j + tcol) * 2;
                const char* pb = Qt + (16 * ks + trow) * T_RS + (64 * wn + 32 * j + tcol) * 2;
                a[j] = tr8(pa, pa + 4 * T_RS);
                b[j] = tr8(pb, pb + 4 * T_RS);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------ 3x3 weight gradient
// ws[slab][tap][n][k] = sum over the slab's pixels p of dY[p - (dy, dx)][n] A[p][k], tap = 3 (dy + 1) + (dx + 1), where the
// output pixel p - (dy, dx) lies in p's image.  64-pixel tiles of A [64][128]; the dY strip of the tile [P0 - S - 1,
// P0 + 64 + S + 1) is staged once and read at nine row offsets; rows a tap may not read are redirected - by the lane that
// supplies that row's address - to a row of zeros.  Wave w owns channels k = 32 w .. 32 w + 31 for all nine taps.
__global__ __launch_bounds__(256, 2) void wgrad3x3_f16_kernel(const _Float16* __restrict__ dY, long lddy,
                                                              const _Float16* __restrict__ A, long lda, long bsa,
                                                              float* __restrict__ ws, long M, int S, long tiles_per_slab) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const int nrows = 66 + 2 * S;
    char* const At = dyn;
    char* const strip = dyn + 64 * T_RS;
    char* const zrow = strip + nrows * 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long slab = blockIdx.x;
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    const int chunk = t & 15, row0 = t >> 4;
    const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    if (t < 4) *reinterpret_cast<h8*>(zrow + 16 * t) = zero8();
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) zero_acc(acc[k]);
    const int S2 = S * S;
    const int lg = (S & (S - 1)) == 0 ? __builtin_ctz(S) : -1;
    h8 av[4], sv[4];
    auto fetch = [&](long tile) {
        const long P0 = tile * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = P0 + row0 + 16 * i;
            av[i] = row < M ? ldg8(A + row * lda + (chunk >> 2) * bsa + (chunk & 3) * 8) : zero8();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = t + 256 * i;                           // (strip row, 16-B piece)
            const long u = P0 - S - 1 + (item >> 2);
            sv[i] = ((item >> 2) < nrows && u >= 0 && u < M) ? ldg8(dY + u * lddy + (item & 3) * 8) : zero8();
        }
    };
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        const long P0 = tile * 64;
        const int rem0 = (int)(P0 % S2);                            // position of the tile's first pixel inside its image
        lds_barrier();                                              // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<h8*>(At + (row0 + 16 * i) * T_RS + chunk * 16) = av[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = t + 256 * i;
            if ((item >> 2) < nrows) *reinterpret_cast<h8*>(strip + (item >> 2) * 64 + (item & 3) * 16) = sv[i];
        }
        lds_barrier();
        if (tile + 1 < tile1) fetch(tile + 1);                      // in flight while this tile multiplies
#pragma unroll 1
        for (int ks = 0; ks < 4; ++ks) {
            const char* pb = At + (16 * ks + trow) * T_RS + (32 * wave + tcol) * 2;
            const h8 b = tr8(pb, pb + 4 * T_RS);
            // which taps may read the rows this lane supplies (pixel p of A; its partner is the output pixel p - (dy, dx))
            unsigned vlo = 0, vhi = 0;
            {
                int ylo, xlo, yhi, xhi;
                if (lg >= 0) {                                      // power-of-two maps: no integer division in the loop
                    const int rem_lo = (rem0 + 16 * ks + trow) & (S2 - 1), rem_hi = (rem0 + 16 * ks + trow + 4) & (S2 - 1);
                    ylo = rem_lo >> lg, xlo = rem_lo & (S - 1), yhi = rem_hi >> lg, xhi = rem_hi & (S - 1);
                } else {
                    const int rem_lo = (rem0 + 16 * ks + trow) % S2, rem_hi = (rem0 + 16 * ks + trow + 4) % S2;
                    ylo = rem_lo / S, xlo = rem_lo - ylo * S, yhi = rem_hi / S, xhi = rem_hi - yhi * S;
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    if (ylo - dy >= 0 && ylo - dy < S && xlo - dx >= 0 && xlo - dx < S) vlo |= 1u << tap;
                    if (yhi - dy >= 0 && yhi - dy < S && xhi - dx >= 0 && xhi - dx < S) vhi |= 1u << tap;
                }
            }
            // all nine taps' dY fragments are requested before the first of them is multiplied (one wait per k-step, not per MFMA)
            h8 a[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = (tap / 3 - 1) * S + (tap % 3 - 1);
                const char* lo = ((vlo >> tap) & 1) ? strip + (16 * ks + trow - off + S + 1) * 64 + tcol * 2 : zrow + tcol * 2;
                const char* hi = ((vhi >> tap) & 1) ? strip + (16 * ks + trow + 4 - off + S + 1) * 64 + tcol * 2 : zrow + tcol * 2;
                a[tap] = tr8(lo, hi);
            }
            __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise sinks each read pair to its MFMA)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap], b, acc[tap], 0, 0, 0);
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

// The same for power-of-two maps (every DenseNet-121 map at 128 / 256 px) without any per-tap address arithmetic: the dY rows a
// 64-pixel tile needs - its own image rows and the one above / below - are staged as a ZERO-PADDED image, (rows + 2) x (S + 2)
// positions per image, so that "the output pixel p - (dy, dx)" is the padded position of p minus a compile-time constant and
// is a zero wherever it leaves the image.  A lane's two row addresses are computed once per k-step; the nine taps are nine
// immediates.  (The general kernel spends ~150 vector instructions per k-step and wave on validity masks and selects
// against 9 MFMAs.)
template <int S>
__global__ __launch_bounds__(256, 2) void wgrad3x3_f16_p2_kernel(const _Float16* __restrict__ dY, long lddy,
                                                                 const _Float16* __restrict__ A, long lda, long bsa,
                                                                 float* __restrict__ ws, long M, long tiles_per_slab) {
    constexpr int NIMG = S >= 8 ? 1 : 64 / (S * S);                 // images per tile
    constexpr int NR = S >= 8 ? 64 / S : S;                         // image rows per tile and image
    constexpr int PW = S + 2, PIMG = (NR + 2) * PW;                 // padded row length, padded positions per image
    constexpr int NPOS = NIMG * PIMG;
    constexpr int ITEMS = NIMG * (NR + 2) * S * 4;                  // (padded row, column, 16-B piece) to stage per tile
    constexpr int LG = S == 4 ? 2 : S == 8 ? 3 : S == 16 ? 4 : S == 32 ? 5 : 6;
    __shared__ __attribute__((aligned(16))) char smem[64 * T_RS + NPOS * 64];
    char* const At = smem;
    char* const strip = smem + 64 * T_RS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long slab = blockIdx.x;
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    const int chunk = t & 15, row0 = t >> 4;
    const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    for (int i = t; i < NPOS * 4; i += 256) *reinterpret_cast<h8*>(strip + i * 16) = zero8();   // incl. the pad columns, for good
    f32x16 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) zero_acc(acc[k]);
    constexpr int NSV = (ITEMS + 255) / 256;
    h8 av[4], sv[NSV];
    auto fetch = [&](long tile) {
        const long P0 = tile * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = P0 + row0 + 16 * i;
            av[i] = row < M ? ldg8(A + row * lda + (chunk >> 2) * bsa + (chunk & 3) * 8) : zero8();
        }
        const int y0 = S >= 8 ? (int)((P0 & (S * S - 1)) >> LG) : 0;      // the tile's first row inside its image
        const long img0 = P0 - ((long)y0 << LG);                           // first pixel of the (first) image
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int item = t + 256 * i, piece = item & 3, x = (item >> 2) & (S - 1), pr = ((item >> 2) >> LG) % (NR + 2),
                      j = ((item >> 2) >> LG) / (NR + 2);
            const int y = y0 - 1 + pr;
            const long u = img0 + (long)j * S * S + ((long)y << LG) + x;
            sv[i] = (item < ITEMS && y >= 0 && y < S && u < M) ? ldg8(dY + u * lddy + piece * 8) : zero8();
        }
    };
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        lds_barrier();                                              // the previous tile's reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<h8*>(At + (row0 + 16 * i) * T_RS + chunk * 16) = av[i];
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int item = t + 256 * i, piece = item & 3, x = (item >> 2) & (S - 1), pr = ((item >> 2) >> LG) % (NR + 2),
                      j = ((item >> 2) >> LG) / (NR + 2);
            if (item < ITEMS) *reinterpret_cast<h8*>(strip + (j * PIMG + pr * PW + x + 1) * 64 + piece * 16) = sv[i];
        }
        lds_barrier();
        if (tile + 1 < tile1) fetch(tile + 1);                      // in flight while this tile multiplies
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* pb = At + (16 * ks + trow) * T_RS + (32 * wave + tcol) * 2;
            const h8 b = tr8(pb, pb + 4 * T_RS);
            // padded position of this lane's row pixels (local index i of the tile): image j, local row, column
            const int ilo = 16 * ks + trow, ihi = ilo + 4;
            const int plo = (S >= 8 ? 0 : (ilo >> (2 * LG)) * PIMG) + (((ilo >> LG) & (NR - 1)) + 1) * PW + (ilo & (S - 1)) + 1;
            const int phi = (S >= 8 ? 0 : (ihi >> (2 * LG)) * PIMG) + (((ihi >> LG) & (NR - 1)) + 1) * PW + (ihi & (S - 1)) + 1;
            const char* lo = strip + plo * 64 + tcol * 2;
            const char* hi = strip + phi * 64 + tcol * 2;
            h8 a[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = -((tap / 3 - 1) * PW + (tap % 3 - 1)) * 64;              // compile-time: an instruction immediate
                a[tap] = tr8(lo + off, hi + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap], b, acc[tap], 0, 0, 0);
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

// ------------------------------------------------------------------------------------------------ conv2 data gradient + norm2/relu2 adjoint
// dB[p][m] = scale2[m] [A[p][m] > 0] sum_{tap, n} dY[p - (dy, dx)][n] W2[n][m][tap]; S0[m] = sum_p d, S1[m] = sum_p d A[p][m]
// with d the masked gradient.  D = [m][pixel]: wave w owns bottleneck channels 32 w .. 32 w + 31 (its weights, nine taps x two
// 16-channel steps, are register fragments), all 128 pixels of the tile; lanes are pixels, so a pixel whose tap leaves the
// image reads the zero row instead (address select, never a data select).
// SP > 0: S = SP is a power of two and tiles are whole - the dY rows of a tile are staged as a zero-padded image (as in
// wgrad3x3_f16_p2_kernel): a lane's nine tap addresses are its pixel's padded position plus nine immediates, no validity test.
template <int SP>
__global__ __launch_bounds__(256, 2) void dgrad3x3_bn_f16_kernel(const _Float16* __restrict__ dY, long lddy,
                                                                 const _Float16* __restrict__ W2b,
                                                                 const _Float16* __restrict__ A, long lda, long bsa,
                                                                 _Float16* __restrict__ dB,
                                                                 const float* __restrict__ scale2, float* __restrict__ ws, long M,
                                                                 int S_rt, long tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const int S = SP > 0 ? SP : S_rt;
    constexpr int NIMG = SP >= 16 ? 1 : (SP > 0 ? 128 / (SP * SP) : 1);
    constexpr int NR = SP >= 16 ? 128 / SP : (SP > 0 ? SP : 1);
    constexpr int PW = SP + 2, PIMG = (NR + 2) * PW, NPOS = NIMG * PIMG;
    constexpr int LG = SP == 4 ? 2 : SP == 8 ? 3 : SP == 16 ? 4 : SP == 32 ? 5 : 6;
    constexpr int P_ITEMS = NIMG * (NR + 2) * (SP > 0 ? SP : 1) * 4;
    const int nrows = SP > 0 ? NPOS : 130 + 2 * S;
    char* const At = dyn;
    char* const strip = dyn + 128 * R_RS;
    char* const zrow = strip + nrows * Y_RS;
    float* const sc2 = reinterpret_cast<float*>(zrow + Y_RS);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const long ntiles = (M + 127) / 128;
    const long tile0 = blockIdx.x * tiles_per_wg;
    const long tile1 = tile0 + tiles_per_wg < ntiles ? tile0 + tiles_per_wg : ntiles;
    const int chunk = t & 15, row0 = t >> 4;
    if (t < 5) *reinterpret_cast<h8*>(zrow + 16 * t) = zero8();
    if (SP > 0)
        for (int i = t; i < NPOS * 5; i += 256) *reinterpret_cast<h8*>(strip + i * 16) = zero8();   // incl. the pad columns, for good
    if (t < 128) sc2[t] = scale2[t];
    h8 wf[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int s = 0; s < 2; ++s) wf[tap][s] = ldg8(W2b + ((tap * 128 + 32 * wave + r) * 32 + 16 * s + 8 * h));
    float S0[16], S1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) S0[i] = S1[i] = 0.f;
    const int S2 = S * S;
    const int lg = (S & (S - 1)) == 0 ? __builtin_ctz(S) : -1;
    const int strip_items = SP > 0 ? P_ITEMS : nrows * 4;           // <= 1032: five per thread (padded form: four)
    constexpr int NSV = SP > 0 ? (P_ITEMS + 255) / 256 : 5;
    h8 av[8], sv[NSV];
    auto fetch = [&](long tile) {
        const long P0 = tile * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = P0 + row0 + 16 * i;
            av[i] = row < M ? ldg8(A + row * lda + (chunk >> 2) * bsa + (chunk & 3) * 8) : zero8();
        }
        if constexpr (SP > 0) {
            const int y0 = SP >= 16 ? (int)((P0 & (SP * SP - 1)) >> LG) : 0;
            const long img0 = P0 - ((long)y0 << LG);
#pragma unroll
            for (int i = 0; i < NSV; ++i) {
                const int item = t + 256 * i, piece = item & 3, x = (item >> 2) & (SP - 1), pr = ((item >> 2) >> LG) % (NR + 2),
                          j = ((item >> 2) >> LG) / (NR + 2);
                const int y = y0 - 1 + pr;
                const long u = img0 + (long)j * SP * SP + ((long)y << LG) + x;
                sv[i] = (item < P_ITEMS && y >= 0 && y < SP && u < M) ? ldg8(dY + u * lddy + piece * 8) : zero8();
            }
        } else {
#pragma unroll
            for (int i = 0; i < NSV; ++i) {
                const int item = t + 256 * i;
                const long u = P0 - S - 1 + (item >> 2);
                sv[i] = (item < strip_items && u >= 0 && u < M) ? ldg8(dY + u * lddy + (item & 3) * 8) : zero8();
            }
        }
    };
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        const long P0 = tile * 128;
        const int rem0 = (int)(P0 % S2);
        lds_barrier();                                              // the previous tile's store-out reads are done
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<h8*>(At + (row0 + 16 * i) * R_RS + chunk * 16) = av[i];
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int item = t + 256 * i;
            if constexpr (SP > 0) {
                const int piece = item & 3, x = (item >> 2) & (SP - 1), pr = ((item >> 2) >> LG) % (NR + 2), j = ((item >> 2) >> LG) / (NR + 2);
                if (item < P_ITEMS) *reinterpret_cast<h8*>(strip + (j * PIMG + pr * PW + x + 1) * Y_RS + piece * 16) = sv[i];
            } else {
                if (item < strip_items) *reinterpret_cast<h8*>(strip + (item >> 2) * Y_RS + (item & 3) * 16) = sv[i];
            }
        }
        lds_barrier();
        if (tile + 1 < tile1) fetch(tile + 1);                      // in flight while this tile multiplies
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            f32x16 acc;
            zero_acc(acc);
            const int px = 32 * c + r;
            int y = 0, x = 0;
            const char* centre = strip;
            if constexpr (SP > 0) {
                centre = strip + ((SP >= 16 ? 0 : (px >> (2 * LG)) * PIMG) + (((px >> LG) & (NR - 1)) + 1) * PW + (px & (SP - 1)) + 1) * Y_RS;
            } else if (lg >= 0) {
                const int rem = (rem0 + px) & (S2 - 1);
                y = rem >> lg, x = rem & (S - 1);
            } else {
                const int rem = (rem0 + px) % S2;
                y = rem / S, x = rem - y * S;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // a kernel row's six fragments (three taps x two 16-channel steps) are requested together, then multiplied
                h8 b[6];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int dy = ky - 1, dx = kx - 1;
                    const char* src;
                    if constexpr (SP > 0) {
                        src = centre - (dy * PW + dx) * Y_RS;       // an immediate
                    } else {
                        const bool ok = y - dy >= 0 && y - dy < S && x - dx >= 0 && x - dx < S;
                        src = ok ? strip + (px - dy * S - dx + S + 1) * Y_RS : zrow;
                    }
                    b[2 * kx] = *reinterpret_cast<const h8*>(src + (8 * h) * 2);
                    b[2 * kx + 1] = *reinterpret_cast<const h8*>(src + (16 + 8 * h) * 2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[3 * ky + (j >> 1)][j & 1], b[j], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // adjoint of norm2 -> relu2 on the accumulators; the tile is rewritten in place (each element by the lane that
            // read it; the matrix operands above come from the strip, never from this tile)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m0 = 32 * wave + 8 * g + 4 * h;
                h4* const cell = reinterpret_cast<h4*>(At + px * R_RS + m0 * 2);
                const h4 av = *cell;
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc2 + m0);
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float af = (float)av[e];
                    const float d = af > 0.f ? acc[4 * g + e] : 0.f;
                    S0[4 * g + e] += d;
                    S1[4 * g + e] += d * af;
                    o[e] = (_Float16)(d * s4[e]);
                }
                *cell = o;
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = P0 + row0 + 16 * i;
            if (row < M) *reinterpret_cast<h8*>(dB + row * 128 + chunk * 8) = *reinterpret_cast<const h8*>(At + (row0 + 16 * i) * R_RS + chunk * 16);
        }
    }
    // sums over the workgroup's pixels: lanes of one half are 32 pixels of the same 16 channels
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            S0[i] += __shfl_xor(S0[i], o, 64);
            S1[i] += __shfl_xor(S1[i], o, 64);
        }
    }
    if (r == 0) {
        float* const out = ws + (long)blockIdx.x * 256;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = 32 * wave + 8 * (i >> 2) + 4 * h + (i & 3);
            out[m] = S0[i];
            out[128 + m] = S1[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv2 backward in ONE pass (round 5)
// The two kernels above read the same operands - the layer's dY strip and the activated bottleneck A - once each: 320 of the
// 896 bytes per pixel the pair moves.  Here ONE workgroup of eight waves stages a 128-pixel tile once and splits the work by
// ROLE (the waves of a role are the kernels above, unchanged in their arithmetic and summation order):
//   waves 0-3  data gradient + norm2 -> relu2 adjoint (dgrad3x3_bn_f16_kernel): D = [m][pixel], wave w = bottleneck channels
//              32 w .., W2's nine taps as 72 register fragments, dB written over the tile copy it reads A from;
//   waves 4-7  weight gradient (wgrad3x3_f16_p2_kernel): D = [n][k] per tap, wave w = channels k = 32 w .., nine accumulators
//              (144 registers) alive for the workgroup's lifetime, both operands by transposing reads.
// 72 MFMAs per wave and tile either way.  The two roles want different LDS row strides (rows read by ds_read_b128 need
// 4 x odd dwords, transposing reads 16 mod 64), so A and the zero-padded dY image are staged twice - LDS writes are cheap,
// HBM reads are what the pass is bound by: 64 + 256 bytes per pixel in, 256 out.  Power-of-two maps, whole 128-pixel tiles.
template <int SP>
__global__ __launch_bounds__(512, 1) void conv3x3_bwd_f16_kernel(const _Float16* __restrict__ dY, long lddy,
                                                                 const _Float16* __restrict__ W2b,
                                                                 const _Float16* __restrict__ A, long lda, long bsa,
                                                                 _Float16* __restrict__ dB, const float* __restrict__ scale2,
                                                                 float* __restrict__ ws_bn, float* __restrict__ ws_w, long M,
                                                                 long tiles_per_wg) {
    constexpr int NIMG = SP >= 16 ? 1 : 128 / (SP * SP);            // images per tile
    constexpr int NR = SP >= 16 ? 128 / SP : SP;                    // image rows per tile and image
    constexpr int PW = SP + 2, PIMG = (NR + 2) * PW, NPOS = NIMG * PIMG;
    constexpr int LG = SP == 4 ? 2 : SP == 8 ? 3 : SP == 16 ? 4 : SP == 32 ? 5 : 6;
    constexpr int P_ITEMS = NIMG * (NR + 2) * SP * 4;               // (padded row, column, 16-B piece) to stage per tile
    constexpr int NSV = (P_ITEMS + 511) / 512;
    __shared__ __attribute__((aligned(16))) char smem[128 * R_RS + 128 * T_RS + NPOS * (Y_RS + 64) + 512];
    char* const AtR = smem;                                         // rows (the data-gradient role; rewritten with dB)
    char* const AtT = smem + 128 * R_RS;                            // transposing reads (the weight-gradient role)
    char* const stripR = AtT + 128 * T_RS;
    char* const stripT = stripR + NPOS * Y_RS;
    float* const sc2 = reinterpret_cast<float*>(stripT + NPOS * 64);
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const long ntiles = M / 128;
    const long tile0 = blockIdx.x * tiles_per_wg;
    const long tile1 = tile0 + tiles_per_wg < ntiles ? tile0 + tiles_per_wg : ntiles;
    const int chunk = t & 15, row0 = t >> 4;                        // staging: 16-B column, rows row0 + 32 i
    for (int i = t; i < NPOS * 5; i += 512) *reinterpret_cast<h8*>(stripR + i * 16) = zero8();   // incl. the pad columns, for good
    for (int i = t; i < NPOS * 4; i += 512) *reinterpret_cast<h8*>(stripT + i * 16) = zero8();
    if (t < 128) sc2[t] = scale2[t];
    const long acol = (long)(chunk >> 2) * bsa + (chunk & 3) * 8;
    h8 av[4], sv[NSV];
    auto fetch = [&](long tile) {
        const long P0 = tile * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = ldg8(A + (P0 + row0 + 32 * i) * lda + acol);
        const int y0 = SP >= 16 ? (int)((P0 & (SP * SP - 1)) >> LG) : 0;
        const long img0 = P0 - ((long)y0 << LG);
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int item = t + 512 * i, piece = item & 3, x = (item >> 2) & (SP - 1), pr = ((item >> 2) >> LG) % (NR + 2),
                      j = ((item >> 2) >> LG) / (NR + 2);
            const int y = y0 - 1 + pr;
            const long u = img0 + (long)j * SP * SP + ((long)y << LG) + x;
            sv[i] = (item < P_ITEMS && y >= 0 && y < SP && u < M) ? ldg8(dY + u * lddy + piece * 8) : zero8();
        }
    };
    const int wq = wave & 3;
    const int trow = 8 * h + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto stage = [&]() {                                            // registers -> both copies of the tile and of the padded dY image
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<h8*>(AtR + (row0 + 32 * i) * R_RS + chunk * 16) = av[i];
            *reinterpret_cast<h8*>(AtT + (row0 + 32 * i) * T_RS + chunk * 16) = av[i];
        }
#pragma unroll
        for (int i = 0; i < NSV; ++i) {
            const int item = t + 512 * i, piece = item & 3, x = (item >> 2) & (SP - 1), pr = ((item >> 2) >> LG) % (NR + 2),
                      j = ((item >> 2) >> LG) / (NR + 2);
            if (item < P_ITEMS) {
                *reinterpret_cast<h8*>(stripR + (j * PIMG + pr * PW + x + 1) * Y_RS + piece * 16) = sv[i];
                *reinterpret_cast<h8*>(stripT + (j * PIMG + pr * PW + x + 1) * 64 + piece * 16) = sv[i];
            }
        }
    };
    auto store_out = [&](long P0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<h8*>(dB + (P0 + row0 + 32 * i) * 128 + chunk * 8) =
                *reinterpret_cast<const h8*>(AtR + (row0 + 32 * i) * R_RS + chunk * 16);
    };
    if (tile0 < tile1) fetch(tile0);
    // The two roles are two separate tile loops with the same three barriers per tile (a workgroup barrier counts waves, not
    // code addresses): written as one loop with a role branch inside, the register state of BOTH roles - 72 + 32 and 144 -
    // was live across it in every wave (560 spills).
    if (wave < 4) {
        // ------------------------------------------------ data gradient + adjoint (as dgrad3x3_bn_f16_kernel)
        h8 wf[9][2];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 2; ++q) wf[tap][q] = ldg8(W2b + ((tap * 128 + 32 * wq + r) * 32 + 16 * q + 8 * h));
        float S0[16], S1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) S0[i] = S1[i] = 0.f;
        for (long tile = tile0; tile < tile1; ++tile) {
            lds_barrier();                                          // the previous tile's reads (and its store-out) are done
            stage();
            lds_barrier();
            if (tile + 1 < tile1) fetch(tile + 1);                  // in flight while this tile multiplies
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                f32x16 acc;
                zero_acc(acc);
                const int px = 32 * c + r;
                const char* const centre =
                    stripR + ((SP >= 16 ? 0 : (px >> (2 * LG)) * PIMG) + (((px >> LG) & (NR - 1)) + 1) * PW + (px & (SP - 1)) + 1) * Y_RS;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    h8 b[6];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const char* src = centre - ((ky - 1) * PW + (kx - 1)) * Y_RS;       // an immediate
                        b[2 * kx] = *reinterpret_cast<const h8*>(src + (8 * h) * 2);
                        b[2 * kx + 1] = *reinterpret_cast<const h8*>(src + (16 + 8 * h) * 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[3 * ky + (j >> 1)][j & 1], b[j], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m0 = 32 * wq + 8 * g + 4 * h;
                    h4* const cell = reinterpret_cast<h4*>(AtR + px * R_RS + m0 * 2);
                    const h4 a4 = *cell;
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc2 + m0);
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float af = (float)a4[e];
                        const float d = af > 0.f ? acc[4 * g + e] : 0.f;
                        S0[4 * g + e] += d;
                        S1[4 * g + e] += d * af;
                        o[e] = (_Float16)(d * s4[e]);
                    }
                    *cell = o;
                }
            }
            lds_barrier();
            store_out(tile * 128);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                S0[i] += __shfl_xor(S0[i], o, 64);
                S1[i] += __shfl_xor(S1[i], o, 64);
            }
        }
        if (r == 0) {
            float* const out = ws_bn + (long)blockIdx.x * 256;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = 32 * wq + 8 * (i >> 2) + 4 * h + (i & 3);
                out[m] = S0[i];
                out[128 + m] = S1[i];
            }
        }
    } else {
        // ------------------------------------------------ weight gradient (as wgrad3x3_f16_p2_kernel, 8 k-steps of 16 pixels)
        f32x16 wacc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) zero_acc(wacc[k]);
        for (long tile = tile0; tile < tile1; ++tile) {
            lds_barrier();
            stage();
            lds_barrier();
            if (tile + 1 < tile1) fetch(tile + 1);
#pragma unroll 1
            for (int ks = 0; ks < 8; ++ks) {
                const char* pb = AtT + (16 * ks + trow) * T_RS + (32 * wq + tcol) * 2;
                const h8 b = tr8(pb, pb + 4 * T_RS);
                const int ilo = 16 * ks + trow, ihi = ilo + 4;
                const int plo = (SP >= 16 ? 0 : (ilo >> (2 * LG)) * PIMG) + (((ilo >> LG) & (NR - 1)) + 1) * PW + (ilo & (SP - 1)) + 1;
                const int phi = (SP >= 16 ? 0 : (ihi >> (2 * LG)) * PIMG) + (((ihi >> LG) & (NR - 1)) + 1) * PW + (ihi & (SP - 1)) + 1;
                const char* lo = stripT + plo * 64 + tcol * 2;
                const char* hi = stripT + phi * 64 + tcol * 2;
                h8 a[9];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int off = -((tap / 3 - 1) * PW + (tap % 3 - 1)) * 64;              // compile-time: an instruction immediate
                    a[tap] = tr8(lo + off, hi + off);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wacc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap], b, wacc[tap], 0, 0, 0);
            }
            lds_barrier();
            store_out(tile * 128);
        }
        float* const out = ws_w + (long)blockIdx.x * (9L * 32 * 128);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int n = (q & 3) + 8 * (q >> 2) + 4 * h;
                out[(tap * 32 + n) * 128 + 32 * wq + (lane & 31)] = wacc[tap][q];
            }
    }
}

// ------------------------------------------------------------------------------------------------ conv1 data gradient + norm1/relu1 adjoint
// G[p][c] += scale1[c] d,  d = [scale1[c] X[p][c] + shift1[c] > 0] sum_m dB[p][m] W1[m][c];  S0[c] = sum_p d,
// S1[c] = sum_p d (X[p][c] - mean[c]).  A workgroup owns 128 input channels (a wave 32, its weight slice in registers) and a
// contiguous range of 64-pixel tiles; D = [channel][pixel].
// WGRAD: the same pass also takes conv1's WEIGHT gradient from the tiles it has staged - wsw[slab][m][c] = sum over the slab's
// pixels of dB[p][m] relu(bn1(X[p][c])) - with both operands read out of the row-major tiles by transposing reads (the
// activation is applied to the fragment: a lane holds 8 pixels of ONE channel): gnx_wgrad1x1_f16's pass over dB and X (a
// quarter of this kernel's bytes) is not made.  64 more accumulator registers: two workgroups per CU instead of three.
template <bool WGRAD>
__global__ __launch_bounds__(256, WGRAD ? 2 : 3) void dgrad1x1_bn_f16_kernel(
    const _Float16* __restrict__ dB, const _Float16* __restrict__ W1t, const _Float16* __restrict__ X, long ldx, long bsx,
    _Float16* __restrict__ G, long ldg, long bsg, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, float* __restrict__ ws, long M, int cin, int n_cb, long tiles_per_slab, long n_slabs,
    float* __restrict__ wsw) {
    __shared__ __attribute__((aligned(16))) char smem[3 * 64 * R_RS + 3 * 128 * 4];
    char* const Bt = smem;
    char* const Xt = smem + 64 * R_RS;
    char* const Gt = smem + 2 * 64 * R_RS;
    float* const cst = reinterpret_cast<float*>(smem + 3 * 64 * R_RS);     // [scale | shift | mean][128]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    // block order as in the 1x1 weight gradient: the n_cb workgroups of one slab (same dB rows) share an XCD
    const int rem8 = blockIdx.x % (8 * n_cb), cb = rem8 / 8;
    const long slab = (long)(blockIdx.x / (8 * n_cb)) * 8 + (rem8 & 7);
    if (slab >= n_slabs) return;
    const int cbase = cb * 128;
    const bool active = cbase + 32 * wave < cin;                           // 32 | cin: a wave's channels are all in or all out
    const long ntiles = (M + 63) / 64;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    const int chunk = t & 15, row0 = t >> 4;
    const bool cok = cbase + chunk * 8 < cin;
    const long xcol = (long)((cbase >> 5) + (chunk >> 2)) * bsx + (chunk & 3) * 8;     // this thread's 8 channels: (c >> 5) bs + (c & 31)
    const long gcol = (long)((cbase >> 5) + (chunk >> 2)) * bsg + (chunk & 3) * 8;
    if (t < 128) {
        const bool ok = cbase + t < cin;
        cst[t] = ok ? scale[cbase + t] : 0.f;
        cst[128 + t] = ok ? shift[cbase + t] : 0.f;
        cst[256 + t] = ok ? mean[cbase + t] : 0.f;
    }
    h8 wf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) wf[s] = active ? ldg8(W1t + ((long)(cbase + 32 * wave + r) * 128 + 16 * s + 8 * h)) : zero8();
    float S0[16], S1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) S0[i] = S1[i] = 0.f;
    f32x16 wacc[WGRAD ? 4 : 1];
#pragma unroll
    for (int i = 0; i < (WGRAD ? 4 : 1); ++i) zero_acc(wacc[i]);
    const int trow = 8 * h + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    h8 bv[4], xv[4], gv[4];
    auto fetch = [&](long tile) {
        const long m0 = tile * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = m0 + row0 + 16 * i;
            const bool ok = row < M;
            bv[i] = ok ? ldg8(dB + row * 128 + chunk * 8) : zero8();
            xv[i] = (ok && cok) ? ldg8(X + row * ldx + xcol) : zero8();
            gv[i] = (ok && cok) ? ldg8(G + row * ldg + gcol) : zero8();
        }
    };
    if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        const long m0 = tile * 64;
        lds_barrier();                                              // the previous tile's store-out reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<h8*>(Bt + (row0 + 16 * i) * R_RS + chunk * 16) = bv[i];
            *reinterpret_cast<h8*>(Xt + (row0 + 16 * i) * R_RS + chunk * 16) = xv[i];
            *reinterpret_cast<h8*>(Gt + (row0 + 16 * i) * R_RS + chunk * 16) = gv[i];
        }
        lds_barrier();
        if (tile + 1 < tile1) fetch(tile + 1);                      // (G rows of the next tile: written by nobody else)
        if (WGRAD && active) {
            // dW1[m][c] += sum_p dB[p][m] act(X[p][c]): D = [m][c], k = pixels; this wave's 32 channels, all 128 m
            const float wsc = cst[32 * wave + r], wsh = cst[128 + 32 * wave + r];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const char* px_ = Xt + (16 * ks + trow) * R_RS + (32 * wave + tcol) * 2;
                h8 b = tr8(px_, px_ + 4 * R_RS);
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = (_Float16)fmaxf(fmaf((float)b[j], wsc, wsh), 0.f);
                h8 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const char* pa = Bt + (16 * ks + trow) * R_RS + (32 * mt + tcol) * 2;
                    a[mt] = tr8(pa, pa + 4 * R_RS);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) wacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b, wacc[mt], 0, 0, 0);
            }
        }
        if (active) {
#pragma unroll 1
            for (int c = 0; c < 2; ++c) {
                f32x16 acc;
                zero_acc(acc);
                h8 b[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const h8*>(Bt + (32 * c + r) * R_RS + (16 * s + 8 * h) * 2);
                __builtin_amdgcn_sched_barrier(0);                  // all eight fragments requested, then multiplied
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], b[s], acc, 0, 0, 0);
                const int px = 32 * c + r;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = 32 * wave + 8 * g + 4 * h;
                    const h4 x4 = *reinterpret_cast<const h4*>(Xt + px * R_RS + c0 * 2);
                    h4* const cell = reinterpret_cast<h4*>(Gt + px * R_RS + c0 * 2);
                    const h4 g4 = *cell;
                    const f32x4 sc4 = *reinterpret_cast<const f32x4*>(cst + c0);
                    const f32x4 sh4 = *reinterpret_cast<const f32x4*>(cst + 128 + c0);
                    const f32x4 mu4 = *reinterpret_cast<const f32x4*>(cst + 256 + c0);
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xf = (float)x4[e];
                        const float d = fmaf(xf, sc4[e], sh4[e]) > 0.f ? acc[4 * g + e] : 0.f;
                        S0[4 * g + e] += d;
                        S1[4 * g + e] += d * (xf - mu4[e]);
                        o[e] = (_Float16)fmaf(d, sc4[e], (float)g4[e]);
                    }
                    *cell = o;
                }
            }
        }
        lds_barrier();
        if (cok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = m0 + row0 + 16 * i;
                if (row < M)
                    *reinterpret_cast<h8*>(G + row * ldg + gcol) =
                        *reinterpret_cast<const h8*>(Gt + (row0 + 16 * i) * R_RS + chunk * 16);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            S0[i] += __shfl_xor(S0[i], o, 64);
            S1[i] += __shfl_xor(S1[i], o, 64);
        }
    }
    if (r == 0 && active) {
        const long cw = (long)n_cb * 128;
        float* const out = ws + slab * 2 * cw;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cbase + 32 * wave + 8 * (i >> 2) + 4 * h + (i & 3);
            out[c] = S0[i];
            out[cw + c] = S1[i];
        }
    }
    if (WGRAD && active) {
        const long cw = (long)n_cb * 128;
        float* const out = wsw + slab * 128 * cw + cbase + 32 * wave + r;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[(long)(32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h) * cw] = wacc[mt][i];
    }
}

// ------------------------------------------------------------------------------------------------ elementwise adjoints (HBM passes)
// norm_final -> relu -> global average (densenet.py:153-156): G[p][c] = s * scale[c] [bn(X[p][c]) > 0] dfeats[img][c] / S2 and the
// BatchNorm sums (true scale: computed from the fp32 gradient).  One thread = 8 channels of one image slot.
// Thread -> (slot, 8 channels): row-major buffers give consecutive threads consecutive 16-B columns of one row; channel-
// blocked buffers (`blocked`) give a wave ONE 32-channel block of 16 slots - 4 threads per 64-B block row - so that its
// accesses stay whole contiguous pieces of memory.
__device__ __forceinline__ bool slot_chunk(long gid, int C, bool blocked, long slots, long& slot, int& chunk) {
    if (blocked) {
        const int nb = C >> 5;
        const long rest = gid >> 6;
        chunk = (int)(rest % nb) * 4 + (int)(gid & 3);
        slot = (rest / nb) * 16 + ((gid >> 2) & 15);
    } else {
        const int nch = C / 8;
        chunk = (int)(gid % nch);
        slot = gid / nch;
    }
    return slot < slots;
}
__global__ __launch_bounds__(256) void tail_bwd_f16_kernel(const float* __restrict__ dfeats, long ldf,
                                                           const _Float16* __restrict__ X, long ldx, long bsx,
                                                           _Float16* __restrict__ G, long ldg, long bsg,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ ls, float* __restrict__ ws, long imgs, int C,
                                                           int S2, int slots, int blocked) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    int chunk;
    long slot;
    if (!slot_chunk(gid, C, blocked != 0, slots, slot, chunk)) return;
    const int c0 = chunk * 8;
    const long xcol = (long)(c0 >> 5) * bsx + (c0 & 31), gcol = (long)(c0 >> 5) * bsg + (c0 & 31);
    float sc[8], sh[8], mu[8], S0[8], S1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[c0 + j];
        sh[j] = shift[c0 + j];
        mu[j] = mean[c0 + j];
        S0[j] = S1[j] = 0.f;
    }
    const float s = ls[0], inv = 1.f / (float)S2;
    for (long img = slot; img < imgs; img += slots) {
        float d[8];
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(dfeats + img * ldf + c0);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(dfeats + img * ldf + c0 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[j] = d0[j] * inv;
            d[4 + j] = d1[j] * inv;
        }
        for (int p = 0; p < S2; ++p) {
            const long row = img * S2 + p;
            const h8 x = ldg8(X + row * ldx + xcol);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = (float)x[j];
                const float v = fmaf(xf, sc[j], sh[j]) > 0.f ? d[j] : 0.f;
                S0[j] += v;
                S1[j] += v * (xf - mu[j]);
                o[j] = (_Float16)(v * sc[j] * s);
            }
            *reinterpret_cast<h8*>(G + row * ldg + gcol) = o;
        }
    }
    float* const out = ws + slot * 2L * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        out[c0 + j] = S0[j];
        out[C + c0 + j] = S1[j];
    }
}

// Transition norm -> relu -> (conv) -> avgpool 2x2 (densenet.py:47-54, evaluated pool-first): given the gradient of the POOLED
// activated map dP [imgs (S/2)^2][C], G[p][c] = scale[c] [bn(X[p][c]) > 0] dP[pool(p)][c] / 4 for all four source pixels (this
// INITIALISES the block gradient) and the BatchNorm sums (scaled by s like dP).  One thread = 8 channels of one pooled pixel.
__global__ __launch_bounds__(256) void trans_bwd_f16_kernel(const _Float16* __restrict__ dP, long ldp, long bsp,
                                                            const _Float16* __restrict__ X, long ldx, long bsx,
                                                            _Float16* __restrict__ G, long ldg, long bsg,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ mean,
                                                            float* __restrict__ ws, long imgs, int C, int S, int slots,
                                                            int blocked) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    int chunk;
    long slot;
    if (!slot_chunk(gid, C, blocked != 0, slots, slot, chunk)) return;
    const int c0 = chunk * 8;
    const long pcol = (long)(c0 >> 5) * bsp + (c0 & 31), xcol = (long)(c0 >> 5) * bsx + (c0 & 31),
               gcol = (long)(c0 >> 5) * bsg + (c0 & 31);
    float sc[8], sh[8], mu[8], S0[8], S1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[c0 + j];
        sh[j] = shift[c0 + j];
        mu[j] = mean[c0 + j];
        S0[j] = S1[j] = 0.f;
    }
    const int So = S / 2;
    const long Mp = imgs * So * So;
    for (long q = slot; q < Mp; q += slots) {
        const long img = q / (So * So);
        const int rem = (int)(q - img * So * So);
        const int oy = rem / So, ox = rem - oy * So;
        const h8 dp = ldg8(dP + q * ldp + pcol);
        float d[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 0.25f * (float)dp[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long row = (img * S + 2 * oy + (k >> 1)) * S + 2 * ox + (k & 1);
            const h8 x = ldg8(X + row * ldx + xcol);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xf = (float)x[j];
                const float v = fmaf(xf, sc[j], sh[j]) > 0.f ? d[j] : 0.f;
                S0[j] += v;
                S1[j] += v * (xf - mu[j]);
                o[j] = (_Float16)(v * sc[j]);
            }
            *reinterpret_cast<h8*>(G + row * ldg + gcol) = o;
        }
    }
    float* const out = ws + slot * 2L * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        out[c0 + j] = S0[j];
        out[C + c0 + j] = S1[j];
    }
}

// out[m][c] = G16[m][c] / s for c < C (the gradient of block 1's first channels = of the pooled stem map, back in fp32)
__global__ __launch_bounds__(256) void h16_cols_to_f32_kernel(const _Float16* __restrict__ G, long ldg, float* __restrict__ out,
                                                              long ldo, long M, int C, const float* __restrict__ ls,
                                                              int* __restrict__ flag) {
    const int nch = C / 8;
    const long total = M * nch;
    const float inv = ls[1];
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / nch;
        const int c0 = (int)(i - m * nch) * 8;
        const h8 v = ldg8(G + m * ldg + c0);
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = (float)v[j] * inv;
            b[j] = (float)v[4 + j] * inv;
            bad = bad || !(fabsf(a[j]) <= 3.0e38f) || !(fabsf(b[j]) <= 3.0e38f);
        }
        *reinterpret_cast<f32x4*>(out + m * ldo + c0) = a;
        *reinterpret_cast<f32x4*>(out + m * ldo + c0 + 4) = b;
    }
    if (bad && flag) atomicOr(flag, 1);
}

// ------------------------------------------------------------------------------------------------ fixed-order slab reductions
// out[map(i)] (+)= inv * sum_s ws[s * n + i].  mode 0: map = identity; mode 1: i = (tap, n, k) of [9][32][128] -> torch's
// conv2.weight order [n][k][tap]; mode -K: i = (row, col) of [rows][128 ceil(K / 128)] -> [rows][K] (columns past K dropped).
// Sixteen lanes share an output element: lane j sums slabs j, j + 16, ... in order, the sixteen partial sums are added in a
// fixed tree (deterministic).  Consecutive 16-lane groups take consecutive elements: a slab row is read in 64-B pieces.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, long nslab, long n, float* __restrict__ out,
                                                           const float* __restrict__ ls, int accumulate, int mode,
                                                           int* __restrict__ flag) {
    const int sl = threadIdx.x >> 4;                               // slab lane
    const long i = (long)blockIdx.x * 16 + (threadIdx.x & 15);     // element: a block owns 16 consecutive elements
    float s = 0.f;
    if (i < n)
        for (long k = sl; k < nslab; k += 16) s += ws[k * n + i];
    __shared__ float part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 16 && i < n) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += part[threadIdx.x + 16 * j];
        tot *= ls ? ls[1] : 1.f;
        long o = i;
        if (mode == 1) {
            const int tap = (int)(i / (32 * 128)), rem = (int)(i % (32 * 128));
            o = (long)rem * 9 + tap;
        } else if (mode < 0) {                                      // slabs [rows][cw] (cw = 128 ceil(K / 128)) -> out [rows][K], K = -mode
            const long K = -mode, cw = (K + 127) / 128 * 128;
            const long row = i / cw, col = i - row * cw;
            if (col >= K) return;
            o = row * K + col;
        }
        if (accumulate) tot += out[o];
        out[o] = tot;
        if (flag && !(fabsf(tot) <= 3.0e38f)) atomicOr(flag, 1);
    }
}
// BatchNorm sums: slabs [s][2][cw] of (S0, S1) -> dbeta[c] = inv S0, dgamma[c] = inv f(S0, S1):
// mode 0: dgamma = S1 p0[c] (S1 = sum d (x - mean), p0 = invstd);  mode 1: dgamma = (S1 - p1[c] S0) / p0[c] (S1 = sum d a with
// a = relu(gamma x_hat + beta) the stored activation, p0 = gamma, p1 = beta: x_hat = (a - beta) / gamma where d != 0).
// A 256-thread block owns 16 channels x 16 slab lanes (same scheme as above).
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ ws, long nslab, int C, long cw,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        const float* __restrict__ p0, const float* __restrict__ p1, int mode,
                                                        const float* __restrict__ ls, int accumulate, int* __restrict__ flag) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const int sl = threadIdx.x >> 4;
    float s0 = 0.f, s1 = 0.f;
    if (c < C)
        for (long k = sl; k < nslab; k += 16) {
            s0 += ws[(2 * k) * cw + c];
            s1 += ws[(2 * k + 1) * cw + c];
        }
    __shared__ float part[2][256];
    part[0][threadIdx.x] = s0;
    part[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        s0 = s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s0 += part[0][threadIdx.x + 16 * j];
            s1 += part[1][threadIdx.x + 16 * j];
        }
        const float inv = ls ? ls[1] : 1.f;
        float dg = mode == 0 ? s1 * p0[c] : (s1 - p1[c] * s0) / p0[c];
        dg *= inv;
        float db = s0 * inv;
        if (dgamma) {
            if (accumulate) dg += dgamma[c];
            dgamma[c] = dg;
        }
        if (dbeta) {
            if (accumulate) db += dbeta[c];
            dbeta[c] = db;
        }
        if (flag && (!(fabsf(dg) <= 3.0e38f) || !(fabsf(db) <= 3.0e38f))) atomicOr(flag, 1);
    }
}

bool al16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

struct SlabPlan {
    long tiles, slabs, per;
};
SlabPlan plan_slabs(long M, int tile, long want) {
    SlabPlan p;
    p.tiles = (M + tile - 1) / tile;
    long slabs = want < 1 ? 1 : want;
    if (slabs > p.tiles) slabs = p.tiles;
    if (slabs < 1) slabs = 1;
    p.per = (p.tiles + slabs - 1) / slabs;
    p.slabs = p.per > 0 ? (p.tiles + p.per - 1) / p.per : 1;
    if (p.slabs < 1) p.slabs = 1;
    return p;
}
// Pixel slabs for kernels whose grid is (slabs rounded up to 8) x `blocks` channel blocks: the largest multiple of 8 that keeps
// the WHOLE grid co-resident (`cap` workgroups = 256 CUs x 2 or 3).  ceil(cap / blocks) did not: with 3, 5, 6 or 7 channel
// blocks (cin = 288-384, 544-896: 32 of DenseNet-121's 58 dense layers) it launched 520-560 workgroups for 512 places, and
// the handful left over ran as a second round after the first workgroups had finished - 1.3-1.4x the time of those layers.
long slabs_for(long cap, long blocks) {
    const long s = cap / blocks / 8 * 8;
    return s < 8 ? 8 : s;
}
SlabPlan plan_wgrad1(long M, int N, int K) {
    const long blocks = (long)((K + 127) / 128) * ((N + 127) / 128);
    return plan_slabs(M, 64, slabs_for(768, blocks));               // three workgroups per CU
}
SlabPlan plan_dgrad1(long M, int K, bool wgrad = false) {
    const long n_cb = (K + 127) / 128;
    return plan_slabs(M, 64, slabs_for(wgrad ? 512 : 768, n_cb));
}

}  // namespace

// dW[N][K] (fp32, (+)=) = 1/s sum_m dY16[m][n] act(X16[m][k]), act = relu(scale[k] x + shift[k]) or the identity when scale is
// NULL.  8 | N, 8 | K, 16-B aligned rows.
GNX_EXPORT long gnx_wgrad1x1_f16_workspace(long M, int N, int K) { return plan_wgrad1(M, N, K).slabs * (long)N * K; }
GNX_EXPORT int gnx_wgrad1x1_f16(const void* dY16, long lddy, const void* X16, long ldx, const float* scale, const float* shift,
                                float* dW, float* workspace, long M, int N, int K, const float* ls, int accumulate, int* flag,
                                hipStream_t stream) {
    if (!dY16 || !X16 || !dW || !workspace || !ls || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K || (!scale) != (!shift))
        return GNX_ERR_BAD_ARG;
    if (N % 8 || K % 8 || lddy % 8 || ldx % 8 || !al16b(dY16) || !al16b(X16)) return GNX_ERR_UNSUPPORTED;
    const SlabPlan p = plan_wgrad1(M, N, K);
    const int n_kb = (K + 127) / 128, n_nb = (N + 127) / 128;
    const long grid = (p.slabs + 7) / 8 * 8 * n_kb * n_nb;
    const _Float16* dY = reinterpret_cast<const _Float16*>(dY16);
    const _Float16* X = reinterpret_cast<const _Float16*>(X16);
    if (scale)
        wgrad1x1_f16_kernel<true><<<(int)grid, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, N, K, p.per, n_kb, n_nb, p.slabs);
    else
        wgrad1x1_f16_kernel<false><<<(int)grid, 256, 0, stream>>>(dY, lddy, X, ldx, scale, shift, workspace, M, N, K, p.per, n_kb, n_nb, p.slabs);
    const long n = (long)N * K;
    reduce_slabs_kernel<<<(int)((n + 15) / 16), 256, 0, stream>>>(workspace, p.slabs, n, dW, ls, accumulate, 0, flag);
    return gnx_launch_status();
}

// dW2[32][128][3][3] (fp32, torch layout, (+)=) = 1/s sum_p dY16[p - (dy, dx)][n] A16[p][k] over S x S images.
GNX_EXPORT long gnx_wgrad3x3_f16_workspace(long M) { return plan_slabs(M, 64, 512).slabs * 9L * 32 * 128; }
// `_lb`: A16 given as (lda, bsa) - element (row, k) at row * lda + (k >> 5) * bsa + (k & 31); (128, 32) is the row-major
// [M][128] matrix of gnx_wgrad3x3_f16, (32, rows_total * 32) the channel-blocked tape of gnx_dense_layer_f16_tape.
GNX_EXPORT int gnx_wgrad3x3_f16_lb(const void* dY16, long lddy, const void* A16, long lda, long bsa, float* dW, float* workspace,
                                   long M, int S, const float* ls, int accumulate, int* flag, hipStream_t stream) {
    if (!dY16 || !A16 || !dW || !workspace || !ls || M <= 0 || S <= 0 || lddy < 32 || M % ((long)S * S) != 0 || lda < 32 || bsa < 32)
        return GNX_ERR_BAD_ARG;
    if (lddy % 8 || lda % 8 || bsa % 8 || !al16b(dY16) || !al16b(A16) || S > 64) return GNX_ERR_UNSUPPORTED;
    const SlabPlan p = plan_slabs(M, 64, 512);
    const _Float16* dY = reinterpret_cast<const _Float16*>(dY16);
    const _Float16* A = reinterpret_cast<const _Float16*>(A16);
    const size_t lds = (size_t)64 * T_RS + (size_t)(66 + 2 * S) * 64 + 64;
    switch (M % 64 == 0 ? S : 0) {                                  // power-of-two maps, whole tiles: the padded-image form
        case 4: wgrad3x3_f16_p2_kernel<4><<<(int)p.slabs, 256, 0, stream>>>(dY, lddy, A, lda, bsa, workspace, M, p.per); break;
        case 8: wgrad3x3_f16_p2_kernel<8><<<(int)p.slabs, 256, 0, stream>>>(dY, lddy, A, lda, bsa, workspace, M, p.per); break;
        case 16: wgrad3x3_f16_p2_kernel<16><<<(int)p.slabs, 256, 0, stream>>>(dY, lddy, A, lda, bsa, workspace, M, p.per); break;
        case 32: wgrad3x3_f16_p2_kernel<32><<<(int)p.slabs, 256, 0, stream>>>(dY, lddy, A, lda, bsa, workspace, M, p.per); break;
        case 64: wgrad3x3_f16_p2_kernel<64><<<(int)p.slabs, 256, 0, stream>>>(dY, lddy, A, lda, bsa, workspace, M, p.per); break;
        default: wgrad3x3_f16_kernel<<<(int)p.slabs, 256, lds, stream>>>(dY, lddy, A, lda, bsa, workspace, M, S, p.per);
    }
    const long n = 9L * 32 * 128;
    reduce_slabs_kernel<<<(int)((n + 15) / 16), 256, 0, stream>>>(workspace, p.slabs, n, dW, ls, accumulate, 1, flag);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_wgrad3x3_f16(const void* dY16, long lddy, const void* A16, float* dW, float* workspace, long M, int S,
                                const float* ls, int accumulate, int* flag, hipStream_t stream) {
    return gnx_wgrad3x3_f16_lb(dY16, lddy, A16, 128, 32, dW, workspace, M, S, ls, accumulate, flag, stream);
}

// dB16[M][128] = scale2 * conv3x3^T(dY16, W2) * [A16 > 0]; dgamma2 / dbeta2 (fp32, (+)=) with x_hat recovered from the stored
// activation (gamma2 != 0 everywhere).  W2b16: conv2.weight as [tap][128][32] halves (W2b[tap][m][n] = W[n][m][tap]).
GNX_EXPORT long gnx_conv3x3_dgrad_bnrelu_bwd_f16_workspace(long M) { return plan_slabs(M, 128, 512).slabs * 256L; }
// `_lb`: A16 as (lda, bsa), see gnx_wgrad3x3_f16_lb; dB16 stays the row-major [M][128] scratch matrix.
GNX_EXPORT int gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb(const void* dY16, long lddy, const void* W2b16, const void* A16, long lda, long bsa,
                                                   void* dB16, long M, int S, const float* scale2, const float* gamma2,
                                                   const float* beta2, float* dgamma, float* dbeta, float* workspace,
                                                   const float* ls, int accumulate, int* flag, hipStream_t stream) {
    if (!dY16 || !W2b16 || !A16 || !dB16 || !scale2 || !gamma2 || !beta2 || !workspace || !ls || M <= 0 || S <= 0 || lddy < 32 ||
        M % ((long)S * S) != 0 || lda < 32 || bsa < 32)
        return GNX_ERR_BAD_ARG;
    if (lddy % 8 || lda % 8 || bsa % 8 || !al16b(dY16) || !al16b(A16) || !al16b(dB16) || !al16b(W2b16) || S > 64)
        return GNX_ERR_UNSUPPORTED;
    const SlabPlan p = plan_slabs(M, 128, 512);
    const _Float16* dY = reinterpret_cast<const _Float16*>(dY16);
    const _Float16* Wb = reinterpret_cast<const _Float16*>(W2b16);
    const _Float16* A = reinterpret_cast<const _Float16*>(A16);
    _Float16* dB = reinterpret_cast<_Float16*>(dB16);
    const int grid = (int)p.slabs;
#define GNX_DG3(SS, NPOS)                                                                                                    \
    dgrad3x3_bn_f16_kernel<SS><<<grid, 256, (size_t)128 * R_RS + (size_t)(NPOS) * Y_RS + Y_RS + 512, stream>>>(dY, lddy, Wb, A, lda, bsa, \
                                                                                                              dB, scale2, workspace, M, S, p.per)
    switch (M % 128 == 0 ? S : 0) {                                 // power-of-two maps, whole tiles: the padded-image form
        case 4: GNX_DG3(4, 8 * 36); break;
        case 8: GNX_DG3(8, 2 * 100); break;
        case 16: GNX_DG3(16, 10 * 18); break;
        case 32: GNX_DG3(32, 6 * 34); break;
        case 64: GNX_DG3(64, 4 * 66); break;
        default: GNX_DG3(0, 130 + 2 * S);
    }
#undef GNX_DG3
    if (dgamma || dbeta)
        bn_reduce_kernel<<<8, 256, 0, stream>>>(workspace, p.slabs, 128, 128, dgamma, dbeta, gamma2, beta2, 1, ls, accumulate, flag);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_conv3x3_dgrad_bnrelu_bwd_f16(const void* dY16, long lddy, const void* W2b16, const void* A16, void* dB16, long M,
                                                int S, const float* scale2, const float* gamma2, const float* beta2,
                                                float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate,
                                                int* flag, hipStream_t stream) {
    return gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb(dY16, lddy, W2b16, A16, 128, 32, dB16, M, S, scale2, gamma2, beta2, dgamma, dbeta,
                                               workspace, ls, accumulate, flag, stream);
}

// conv2's whole backward in ONE pass over dY16 and A16 (round 5): dB16 (as gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb), dgamma2 / dbeta2
// AND dW2 [32][128][3][3] (as gnx_wgrad3x3_f16_lb) - one workgroup of eight waves stages each 128-pixel tile once, four waves
// take the data gradient, four the weight gradient.  S in {4, 8, 16, 32, 64} and 128 | M; anything else: GNX_ERR_UNSUPPORTED
// (callers make the two calls).  workspace: gnx_conv3x3_bwd_f16_workspace(M) floats.
static long conv3_bwd_grid(long M) {
    const long tiles = M / 128;
    return tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256;            // one 512-thread workgroup per compute unit
}
GNX_EXPORT long gnx_conv3x3_bwd_f16_workspace(long M) { return conv3_bwd_grid(M) * (256L + 9L * 32 * 128); }
GNX_EXPORT int gnx_conv3x3_bwd_f16_lb(const void* dY16, long lddy, const void* W2b16, const void* A16, long lda, long bsa, void* dB16,
                                      float* dW, long M, int S, const float* scale2, const float* gamma2, const float* beta2,
                                      float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                                      hipStream_t stream) {
    if (!dY16 || !W2b16 || !A16 || !dB16 || !dW || !scale2 || !gamma2 || !beta2 || !workspace || !ls || M <= 0 || S <= 0 ||
        lddy < 32 || M % ((long)S * S) != 0 || lda < 32 || bsa < 32)
        return GNX_ERR_BAD_ARG;
    if (lddy % 8 || lda % 8 || bsa % 8 || !al16b(dY16) || !al16b(A16) || !al16b(dB16) || !al16b(W2b16) || M % 128 != 0 ||
        (S != 4 && S != 8 && S != 16 && S != 32 && S != 64))
        return GNX_ERR_UNSUPPORTED;
    const long grid = conv3_bwd_grid(M);
    const long tiles = M / 128, per = (tiles + grid - 1) / grid;
    const long used = (tiles + per - 1) / per;                      // workgroups that own at least one tile (slabs written)
    float* const ws_bn = workspace;
    float* const ws_w = workspace + grid * 256;
    const _Float16* dY = reinterpret_cast<const _Float16*>(dY16);
    const _Float16* Wb = reinterpret_cast<const _Float16*>(W2b16);
    const _Float16* A = reinterpret_cast<const _Float16*>(A16);
    _Float16* dB = reinterpret_cast<_Float16*>(dB16);
#define GNX_C3B(SS) conv3x3_bwd_f16_kernel<SS><<<(int)used, 512, 0, stream>>>(dY, lddy, Wb, A, lda, bsa, dB, scale2, ws_bn, ws_w, M, per)
    switch (S) {
        case 4: GNX_C3B(4); break;
        case 8: GNX_C3B(8); break;
        case 16: GNX_C3B(16); break;
        case 32: GNX_C3B(32); break;
        default: GNX_C3B(64);
    }
#undef GNX_C3B
    if (dgamma || dbeta)
        bn_reduce_kernel<<<8, 256, 0, stream>>>(ws_bn, used, 128, 128, dgamma, dbeta, gamma2, beta2, 1, ls, accumulate, flag);
    const long n = 9L * 32 * 128;
    reduce_slabs_kernel<<<(int)((n + 15) / 16), 256, 0, stream>>>(ws_w, used, n, dW, ls, accumulate, 1, flag);
    return gnx_launch_status();
}

// The fp16 operands of one dense layer's backward from its fp32 weights, in ONE launch (the torch expressions they replace -
// w1.reshape(128, K).t().half() and w2.permute(2, 3, 1, 0).reshape(9, 128, 32).half() - are two transposing copies of ~30 us
// each per layer and step): W1t16 [K][128] = conv1.weight [128][K] transposed (the operand of the conv1 backward kernels),
// W2b16 [tap 9][m 128][n 32] = conv2.weight [32][128][3][3] (the operand of the conv2 data-gradient kernels).  Rounded once.
namespace {
__global__ __launch_bounds__(256) void dense_bwd_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                                             _Float16* __restrict__ w1t, _Float16* __restrict__ w2b, int K) {
    const int n1 = 128 * K, total = n1 + 9 * 128 * 32;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        if (idx < n1) {
            const int c = idx >> 7, m = idx & 127;                 // consecutive threads write consecutive halves of a row of W1t
            w1t[idx] = (_Float16)w1[(long)m * K + c];
        } else {
            const int j = idx - n1, n = j & 31, m = (j >> 5) & 127, tap = j >> 12;
            w2b[j] = (_Float16)w2[((long)n * 128 + m) * 9 + tap];
        }
    }
}
}  // namespace
GNX_EXPORT int gnx_dense_bwd_f16_pack(const float* w1, const float* w2, void* W1t16, void* W2b16, int K, hipStream_t stream) {
    if (!w1 || !w2 || !W1t16 || !W2b16 || K <= 0) return GNX_ERR_BAD_ARG;
    const int total = 128 * K + 9 * 128 * 32;
    dense_bwd_pack_kernel<<<gnx_cdiv(total, 1024), 256, 0, stream>>>(w1, w2, reinterpret_cast<_Float16*>(W1t16),
                                                                      reinterpret_cast<_Float16*>(W2b16), K);
    return gnx_launch_status();
}

// G16[:, :K] += scale1 * (dB16 . W1) * [bn1(X16) > 0]; dgamma1 / dbeta1 (fp32, (+)=).  W1t16: conv1.weight transposed to
// [K][128] halves.  32 | K.
GNX_EXPORT long gnx_conv1x1_dgrad_bnrelu_bwd_f16_workspace(long M, int K) {
    return plan_dgrad1(M, K).slabs * 2L * ((K + 127) / 128) * 128;
}
// with dW != NULL the pass also produces conv1's weight gradient dW[128][K] (fp32, (+)=) - then `workspace` must hold
// gnx_conv1x1_dgrad_wgrad_f16_workspace(M, K) floats
GNX_EXPORT long gnx_conv1x1_dgrad_wgrad_f16_workspace(long M, int K) {
    return plan_dgrad1(M, K, true).slabs * (2L + 128) * ((K + 127) / 128) * 128;
}
static int dgrad1_launch(const void* dB16, const void* W1t16, const void* X16, long ldx, long bsx, void* G16, long ldg, long bsg,
                         long M, int K, const float* scale, const float* shift, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* dW, float* workspace, const float* ls, int accumulate, int* flag,
                         hipStream_t stream) {
    if (!dB16 || !W1t16 || !X16 || !G16 || !scale || !shift || !mean || !invstd || !workspace || !ls || M <= 0 || K <= 0 ||
        ldx < 32 || ldg < 32 || bsx < 32 || bsg < 32 || (bsx == 32 && ldx < K) || (bsg == 32 && ldg < K))
        return GNX_ERR_BAD_ARG;
    if (K % 32 || ldx % 8 || ldg % 8 || bsx % 8 || bsg % 8 || !al16b(dB16) || !al16b(W1t16) || !al16b(X16) || !al16b(G16))
        return GNX_ERR_UNSUPPORTED;
    const SlabPlan p = plan_dgrad1(M, K, dW != nullptr);
    const int n_cb = (K + 127) / 128;
    const long cw = (long)n_cb * 128;
    float* const wsw = workspace + p.slabs * 2 * cw;
    const int grid = (int)((p.slabs + 7) / 8 * 8 * n_cb);
    const _Float16* dB = reinterpret_cast<const _Float16*>(dB16);
    const _Float16* Wt = reinterpret_cast<const _Float16*>(W1t16);
    const _Float16* X = reinterpret_cast<const _Float16*>(X16);
    _Float16* G = reinterpret_cast<_Float16*>(G16);
    if (dW)
        dgrad1x1_bn_f16_kernel<true><<<grid, 256, 0, stream>>>(dB, Wt, X, ldx, bsx, G, ldg, bsg, scale, shift, mean, workspace, M, K, n_cb,
                                                              p.per, p.slabs, wsw);
    else
        dgrad1x1_bn_f16_kernel<false><<<grid, 256, 0, stream>>>(dB, Wt, X, ldx, bsx, G, ldg, bsg, scale, shift, mean, workspace, M, K, n_cb,
                                                               p.per, p.slabs, nullptr);
    if (dgamma || dbeta)
        bn_reduce_kernel<<<(K + 15) / 16, 256, 0, stream>>>(workspace, p.slabs, K, cw, dgamma, dbeta, invstd, nullptr, 0, ls, accumulate,
                                                           flag);
    if (dW) {
        const long n = 128 * cw;
        reduce_slabs_kernel<<<(int)((n + 15) / 16), 256, 0, stream>>>(wsw, p.slabs, n, dW, ls, accumulate, -K, flag);
    }
    return gnx_launch_status();
}
GNX_EXPORT int gnx_conv1x1_dgrad_bnrelu_bwd_f16(const void* dB16, const void* W1t16, const void* X16, long ldx, void* G16, long ldg,
                                                long M, int K, const float* scale, const float* shift, const float* mean,
                                                const float* invstd, float* dgamma, float* dbeta, float* workspace,
                                                const float* ls, int accumulate, int* flag, hipStream_t stream) {
    return dgrad1_launch(dB16, W1t16, X16, ldx, 32, G16, ldg, 32, M, K, scale, shift, mean, invstd, dgamma, dbeta, nullptr, workspace,
                         ls, accumulate, flag, stream);
}
GNX_EXPORT int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16(const void* dB16, const void* W1t16, const void* X16, long ldx, void* G16,
                                                      long ldg, long M, int K, const float* scale, const float* shift,
                                                      const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                                      float* dW, float* workspace, const float* ls, int accumulate, int* flag,
                                                      hipStream_t stream) {
    if (!dW) return GNX_ERR_BAD_ARG;
    return dgrad1_launch(dB16, W1t16, X16, ldx, 32, G16, ldg, 32, M, K, scale, shift, mean, invstd, dgamma, dbeta, dW, workspace, ls,
                         accumulate, flag, stream);
}
// `_lb`: X16 and G16 given as (ld, bs) - element (row, c) at row * ld + (c >> 5) * bs + (c & 31): (c_total, 32) = row-major,
// (32, rows_total * 32) = the channel-blocked block buffers of the fused forward.  dW == NULL: no weight gradient.
GNX_EXPORT int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb(const void* dB16, const void* W1t16, const void* X16, long ldx, long bsx,
                                                         void* G16, long ldg, long bsg, long M, int K, const float* scale,
                                                         const float* shift, const float* mean, const float* invstd,
                                                         float* dgamma, float* dbeta, float* dW, float* workspace, const float* ls,
                                                         int accumulate, int* flag, hipStream_t stream) {
    return dgrad1_launch(dB16, W1t16, X16, ldx, bsx, G16, ldg, bsg, M, K, scale, shift, mean, invstd, dgamma, dbeta, dW, workspace, ls,
                         accumulate, flag, stream);
}

// norm_final -> relu -> global average pool, backward: G16[M][C] = s * ..., dgamma / dbeta (fp32).  8 | C.
static long tail_slots(long imgs, int C) {
    long slots = 256L * 512 / (C / 8);
    if (slots > imgs) slots = imgs;
    return slots < 1 ? 1 : slots;
}
GNX_EXPORT long gnx_tail_bwd_f16_workspace(long imgs, int C) { return tail_slots(imgs, C) * 2L * C; }
// `_lb`: X16 / G16 as (ld, bs), see gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb (channel-blocked: 32 | C).
GNX_EXPORT int gnx_tail_bwd_f16_lb(const float* dfeats, long ldf, const void* X16, long ldx, long bsx, void* G16, long ldg, long bsg,
                                   long imgs, int C, int S2, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, float* dgamma, float* dbeta, float* workspace, const float* ls,
                                   int accumulate, int* flag, hipStream_t stream) {
    if (!dfeats || !X16 || !G16 || !scale || !shift || !mean || !invstd || !workspace || !ls || imgs <= 0 || C <= 0 || S2 <= 0 ||
        ldf < C || ldx < 32 || ldg < 32 || bsx < 32 || bsg < 32 || (bsx == 32 && ldx < C) || (bsg == 32 && ldg < C))
        return GNX_ERR_BAD_ARG;
    const int blocked = (bsx != 32 || bsg != 32) ? 1 : 0;
    if (C % 8 || ldf % 4 || ldx % 8 || ldg % 8 || bsx % 8 || bsg % 8 || !al16b(dfeats) || !al16b(X16) || !al16b(G16) ||
        (blocked && C % 32))
        return GNX_ERR_UNSUPPORTED;
    const long slots = tail_slots(imgs, C);
    const long threads = (blocked ? (slots + 15) / 16 * 16 : slots) * (C / 8);
    tail_bwd_f16_kernel<<<(int)((threads + 255) / 256), 256, 0, stream>>>(dfeats, ldf, reinterpret_cast<const _Float16*>(X16), ldx, bsx,
                                                                         reinterpret_cast<_Float16*>(G16), ldg, bsg, scale, shift,
                                                                         mean, ls, workspace, imgs, C, S2, (int)slots, blocked);
    if (dgamma || dbeta)
        bn_reduce_kernel<<<(C + 15) / 16, 256, 0, stream>>>(workspace, slots, C, C, dgamma, dbeta, invstd, nullptr, 0, nullptr,
                                                             accumulate, flag);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_tail_bwd_f16(const float* dfeats, long ldf, const void* X16, long ldx, void* G16, long ldg, long imgs, int C,
                                int S2, const float* scale, const float* shift, const float* mean, const float* invstd,
                                float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                                hipStream_t stream) {
    return gnx_tail_bwd_f16_lb(dfeats, ldf, X16, ldx, 32, G16, ldg, 32, imgs, C, S2, scale, shift, mean, invstd, dgamma, dbeta,
                               workspace, ls, accumulate, flag, stream);
}

// transition norm -> relu -> avgpool 2x2, backward from the pooled gradient dP16 [imgs (S/2)^2][C]: G16[imgs S^2][C] (written),
// dgamma / dbeta (fp32).  8 | C, even S.
static long trans_slots(long Mp, int C) {
    long slots = 256L * 2048 / (C / 8);
    if (slots > Mp) slots = Mp;
    return slots < 1 ? 1 : slots;
}
GNX_EXPORT long gnx_trans_bwd_f16_workspace(long imgs, int C, int S) { return trans_slots(imgs * (S / 2) * (S / 2), C) * 2L * C; }
// `_lb`: dP16, X16 and G16 each as (ld, bs), see gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb (channel-blocked: 32 | C).
GNX_EXPORT int gnx_trans_bwd_f16_lb(const void* dP16, long ldp, long bsp, const void* X16, long ldx, long bsx, void* G16, long ldg,
                                    long bsg, long imgs, int C, int S, const float* scale, const float* shift, const float* mean,
                                    const float* invstd, float* dgamma, float* dbeta, float* workspace, const float* ls,
                                    int accumulate, int* flag, hipStream_t stream) {
    if (!dP16 || !X16 || !G16 || !scale || !shift || !mean || !invstd || !workspace || !ls || imgs <= 0 || C <= 0 || S < 2 ||
        ldp < 32 || ldx < 32 || ldg < 32 || bsp < 32 || bsx < 32 || bsg < 32 || (bsp == 32 && ldp < C) || (bsx == 32 && ldx < C) ||
        (bsg == 32 && ldg < C))
        return GNX_ERR_BAD_ARG;
    const int blocked = (bsp != 32 || bsx != 32 || bsg != 32) ? 1 : 0;
    if (C % 8 || S % 2 || ldp % 8 || ldx % 8 || ldg % 8 || bsp % 8 || bsx % 8 || bsg % 8 || !al16b(dP16) || !al16b(X16) ||
        !al16b(G16) || (blocked && C % 32))
        return GNX_ERR_UNSUPPORTED;
    const long slots = trans_slots(imgs * (S / 2) * (S / 2), C);
    const long threads = (blocked ? (slots + 15) / 16 * 16 : slots) * (C / 8);
    trans_bwd_f16_kernel<<<(int)((threads + 255) / 256), 256, 0, stream>>>(reinterpret_cast<const _Float16*>(dP16), ldp, bsp,
                                                                          reinterpret_cast<const _Float16*>(X16), ldx, bsx,
                                                                          reinterpret_cast<_Float16*>(G16), ldg, bsg, scale, shift,
                                                                          mean, workspace, imgs, C, S, (int)slots, blocked);
    if (dgamma || dbeta)
        bn_reduce_kernel<<<(C + 15) / 16, 256, 0, stream>>>(workspace, slots, C, C, dgamma, dbeta, invstd, nullptr, 0, ls, accumulate,
                                                             flag);
    return gnx_launch_status();
}
GNX_EXPORT int gnx_trans_bwd_f16(const void* dP16, long ldp, const void* X16, long ldx, void* G16, long ldg, long imgs, int C, int S,
                                 const float* scale, const float* shift, const float* mean, const float* invstd, float* dgamma,
                                 float* dbeta, float* workspace, const float* ls, int accumulate, int* flag, hipStream_t stream) {
    return gnx_trans_bwd_f16_lb(dP16, ldp, 32, X16, ldx, 32, G16, ldg, 32, imgs, C, S, scale, shift, mean, invstd, dgamma, dbeta,
                                workspace, ls, accumulate, flag, stream);
}

// out[M][C] (fp32, ldo) = G16[:, :C] / s.  8 | C.
GNX_EXPORT int gnx_h16_cols_to_f32(const void* G16, long ldg, float* out, long ldo, long M, int C, const float* ls, int* flag,
                                   hipStream_t stream) {
    if (!G16 || !out || !ls || M <= 0 || C <= 0 || ldg < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (C % 8 || ldg % 8 || ldo % 4 || !al16b(G16) || !al16b(out)) return GNX_ERR_UNSUPPORTED;
    long blocks = (M * (C / 8) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    h16_cols_to_f32_kernel<<<(int)blocks, 256, 0, stream>>>(reinterpret_cast<const _Float16*>(G16), ldg, out, ldo, M, C, ls, flag);
    return gnx_launch_status();
}
