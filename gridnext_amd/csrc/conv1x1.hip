// DenseNet-BC forward kernels for the image spot classifier f (gfx950, exact fp32 on the matrix cores).
//
// Replaces, for /root/reference/gridnext/densenet.py:
//   _DenseLayer.forward :35-44   cat -> BN -> ReLU -> conv1x1   => gnx_conv1x1_bnrelu   (BN+ReLU fused into the A-operand load)
//                                BN -> ReLU -> conv3x3 p1       => gnx_conv3x3_bnrelu   (same prologue, zero padding applied AFTER the activation)
//   _DenseBlock.forward :70-75   torch.cat of features           => none: every layer writes its `growth` channels into
//                                                                   a column range of one channels-last block buffer
//   _Transition :47-54           BN -> ReLU -> conv1x1 -> avgpool2 => gnx_conv1x1_bnrelu(pool=1): the 2x2 average is taken on the
//                                                                   activated input first (both maps are linear: 4x fewer MACs)
//   stem :105-112                conv7x7 s2 p3 | conv3x3 s1 p1   => gnx_conv_stem ; BN -> ReLU -> maxpool3 s2 p1 => gnx_bnrelu_maxpool
//   tail  :152-156               BN -> ReLU -> adaptive_avg_pool(1,1) -> flatten => gnx_bnrelu_avgpool (classifier: gnx_gemm_f32)
//
// HBM layout: activations are channels-last matrices X[M = spots*S*S][C] with a leading dimension (a dense block is ONE
// buffer of leading dimension C_total; a layer reads columns [0, C_in) and writes [C_in, C_in+growth)).
// Weights: conv1x1 as torch stores them [N][K]; conv3x3 repacked once to [tap][N][K] (gnx_repack_conv3x3).
//
// MFMA: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate = an exact fmaf chain; 157.3 TFLOP/s dense peak).
// LDS images are [row][K-chunk of 32 + 4 pad] (144-B rows): 16-B aligned for ds_read_b128 and conflict-free
// (row stride = 9 sixteen-byte slots, odd).  One ds_read_b128 per operand feeds four MFMAs: lane half h supplies
// k = 8*s + 4*h + j for the j-th of them, identically for A and B, so the k-order inside the sum is a fixed permutation.
// This file: the 1x1 convolutions (dense-layer bottleneck, transition, data gradients).
#include "fwd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ conv1x1
// out[m][n] = sum_k act(A[m][k]) * W[n][k];  act(a) = relu(a*scale[k]+shift[k]) (identity when scale == nullptr)
// POOL: row m is a position of the (S_in/2)^2 grid; its A-row is the mean of the 4 activated source rows.
constexpr int C1_BM = 128, C1_BN = 128, C1_BK = 32;

template <bool POOL, bool FAST>
__global__ __launch_bounds__(256) void conv1x1_kernel(const float* __restrict__ A, long lda,
                                                      const float* __restrict__ W, float* __restrict__ out, long ldc,
                                                      long M, int N, int K, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int S_in, int vecA, int vecW,
                                                      const float* __restrict__ oscale,
                                                      const float* __restrict__ oshift, int ldw = 0, int ksplit = 0,
                                                      float* __restrict__ slabs = nullptr) {
    // ksplit > 0 (small M: few 128-row tiles, a long K loop that is latency, not work): blockIdx.z owns k in [z ksplit,
    // (z + 1) ksplit) and writes its partial tile to slabs[z][M][N]; conv1x1_split_reduce_kernel sums them in index order.
    if (ldw == 0) ldw = K;
    if (ksplit > 0) {
        const int kz0 = blockIdx.z * ksplit;
        A += kz0;
        W += kz0;
        if (scale) { scale += kz0; shift += kz0; }
        K = K - kz0 < ksplit ? K - kz0 : ksplit;
        out = slabs + (size_t)blockIdx.z * M * N;
        ldc = N;
        oscale = nullptr;
    }
    __shared__ __attribute__((aligned(16))) float As[C1_BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[C1_BN * LDK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, i = lane & 31;
    const int kq = t & 7, r0 = t >> 3;
    const long m0 = (long)blockIdx.x * C1_BM;
    const int n0 = blockIdx.y * C1_BN;

    long src[4];
    bool rok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const long row = m0 + r0 + 32 * p;
        rok[p] = row < M;
        if (POOL) {
            const int So = S_in >> 1;
            const long img = row / (So * So);
            const int rem = (int)(row - img * So * So);
            const int oy = rem / So, ox = rem - oy * So;
            src[p] = ((img * S_in + 2 * oy) * S_in + 2 * ox) * lda;
        } else {
            src[p] = row * lda;
        }
    }
    const bool has_act = scale != nullptr;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4], sc4, sh4;
    // FAST (aligned pointers, K % 4 == 0): branch-free loads from clamped addresses, zeroed at the LDS write, so the
    // compiler keeps the prefetch in flight across the MFMA loop.  Otherwise the bounds-checked scalar-safe loads.
    auto fetch = [&](int k0) {
        const int k = k0 + 4 * kq;
        const int valid = K - k;
        const int kc = valid > 0 ? k : 0;
        sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
        sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {
            if (has_act) { sc4 = ld4(scale + kc); sh4 = ld4(shift + kc); }
        } else if (has_act && valid > 0) {
            sc4 = ld4_safe(scale + k, valid, vecA);
            sh4 = ld4_safe(shift + k, valid, vecA);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (FAST) {
                const long sp = rok[p] ? src[p] : 0;
                if (POOL) {
                    const float* b0 = A + sp + kc;
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 v = ld4(b0 + ((q >> 1) * (long)S_in + (q & 1)) * lda);
                        if (has_act) v = act4(v, sc4, sh4);
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    ra[p] = make_float4(0.25f * sum.x, 0.25f * sum.y, 0.25f * sum.z, 0.25f * sum.w);
                } else {
                    ra[p] = ld4(A + sp + kc);
                }
                continue;
            }
            if (!rok[p] || valid <= 0) { ra[p] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
            if (POOL) {
                const float* b0 = A + src[p] + k;
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = ld4_safe(b0 + ((q >> 1) * (long)S_in + (q & 1)) * lda, valid, vecA);
                    if (has_act) v = act4(v, sc4, sh4);
                    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                }
                ra[p] = make_float4(0.25f * sum.x, 0.25f * sum.y, 0.25f * sum.z, 0.25f * sum.w);
            } else {
                ra[p] = ld4_safe(A + src[p] + k, valid, vecA);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int n = n0 + r0 + 32 * p;
            if (FAST) {
                rb[p] = ld4(W + (long)(n < N ? n : N - 1) * ldw + kc);
            } else {
                rb[p] = (n < N && valid > 0) ? ld4_safe(W + (long)n * ldw + k, valid, vecW)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    const bool interior = FAST && m0 + C1_BM <= M && (K & 31) == 0 && n0 + C1_BN <= N;
    auto stash = [&](int k0) {
        if (interior) {          // workgroup-uniform: full tile, whole K tiles -> activation + store only
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float4 v = ra[p];
                if (!POOL && has_act) v = act4(v, sc4, sh4);
                *reinterpret_cast<float4*>(&As[(r0 + 32 * p) * LDK + 4 * kq]) = v;
                *reinterpret_cast<float4*>(&Bs[(r0 + 32 * p) * LDK + 4 * kq]) = rb[p];
            }
            return;
        }
        const int valid = K - (k0 + 4 * kq);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = ra[p];
            if (!POOL && has_act && (FAST || rok[p])) v = act4(v, sc4, sh4);
            // lanes of the K tail must hold exact zeros (they meet zero weights; 0*garbage must stay 0)
            if (valid < 4) {
                if (valid < 1) v.x = 0.f;
                if (valid < 2) v.y = 0.f;
                if (valid < 3) v.z = 0.f;
                v.w = 0.f;
            }
            if (!rok[p]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 wv = rb[p];
            if (FAST && (valid < 4 || n0 + r0 + 32 * p >= N)) wv = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&As[(r0 + 32 * p) * LDK + 4 * kq]) = v;
            *reinterpret_cast<float4*>(&Bs[(r0 + 32 * p) * LDK + 4 * kq]) = wv;
        }
    };

    // Loop shape: loads of tile kt+1 are issued, tile kt is multiplied out of LDS, then (same iteration) the loaded
    // registers are activated and written to LDS.  Issue and consumption sit in ONE iteration on purpose: when the
    // prefetch registers were carried across the back edge the compiler shuffled them right after the loads and had
    // to wait for them before the MFMA loop, exposing the memory latency it was meant to hide.
    const int nkt = (K + C1_BK - 1) / C1_BK;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        __builtin_amdgcn_s_setprio(3);          // non-MFMA work at raised priority (see conv3x3_pipe_kernel)
        if (kt + 1 < nkt) fetch((kt + 1) * C1_BK);
        __builtin_amdgcn_s_setprio(0);
        // fragments of step s+1 are read while step s multiplies; the four accumulators are visited round-robin so
        // consecutive MFMAs never depend on each other
        const float* apA = &As[(64 * wm + i) * LDK + 4 * h];
        const float* bpB = &Bs[(64 * wn + i) * LDK + 4 * h];
        float4 a0 = ld4(apA), a1 = ld4(apA + 32 * LDK), b0 = ld4(bpB), b1 = ld4(bpB + 32 * LDK);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (s < 3) {
                na0 = ld4(apA + 8 * (s + 1));
                na1 = ld4(apA + 32 * LDK + 8 * (s + 1));
                nb0 = ld4(bpB + 8 * (s + 1));
                nb1 = ld4(bpB + 32 * LDK + 8 * (s + 1));
            }
#define GNX_MM(c) \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc[0][0], 0, 0, 0); \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc[0][1], 0, 0, 0); \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc[1][0], 0, 0, 0); \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc[1][1], 0, 0, 0);
            GNX_MM(x) GNX_MM(y) GNX_MM(z) GNX_MM(w)
#undef GNX_MM
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        __builtin_amdgcn_s_setprio(3);
        __syncthreads();
        if (kt + 1 < nkt) {
            stash((kt + 1) * C1_BK);
            __syncthreads();
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + 64 * wn + 32 * nt + i;
            if (oscale && col < N) {          // output activation: the consumer's BN+ReLU applied at the store
                const float osc = oscale[col], osh = oshift[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaxf(fmaf(acc[mt][nt][r], osc, osh), 0.f);
            }
            if (interior) {
                float* o = out + (m0 + 64 * wm + 32 * mt + 4 * h) * ldc + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * ldc] = acc[mt][nt][r];
                continue;
            }
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = m0 + 64 * wm + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) out[row * ldc + col] = acc[mt][nt][r];
            }
        }
}

// out[m][n] = sum_z slabs[z][m][n] in index order (+ the consumer's folded BN + ReLU); 4 | N, 16-B accesses
__global__ __launch_bounds__(256) void conv1x1_split_reduce_kernel(const float* __restrict__ slabs, int S, long M, int N,
                                                                   float* __restrict__ out, long ldc,
                                                                   const float* __restrict__ oscale,
                                                                   const float* __restrict__ oshift) {
    const long total = M * (N / 4);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / (N / 4);
        const int n = 4 * (int)(idx - m * (N / 4));
        float4 v = ld4(slabs + m * N + n);
        for (int z = 1; z < S; ++z) {
            const float4 u = ld4(slabs + ((size_t)z * M + m) * N + n);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (oscale) {
            const float4 sc = ld4(oscale + n), sh = ld4(oshift + n);
            v = make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f),
                            fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
        }
        *reinterpret_cast<float4*>(out + m * ldc + n) = v;
    }
}

// K splits of the small-M form: 0 = not its shape.  Few row tiles (M <= 8192 at N <= 128: <= 64 workgroups) and a K loop of
// >= 8 chunks; ~4 chunks per split, at most 16 splits.
int conv1x1_small_splits(long M, int N, int K) {
    if (M > 8192 || K < 256 || K % 32 != 0 || N % 4 != 0) return 0;
    const long tiles = gnx_cdiv(M, C1_BM) * gnx_cdiv(N, C1_BN);
    int s = K / 128;
    while (s > 1 && tiles * s > 256) --s;
    if (tiles * s < 32 && K / 64 <= 16) s = K / 64;            // block 4 at batch 32: 4 tiles - two chunks per split
    if (s > 16) s = 16;
    return s < 2 ? 0 : s;
}

// ------------------------------------------------------------------------------------------------ conv1x1, wave-specialised
// For whole 128 x 128 x 32 tiles (128 | M, 128 | N, 32 | K, aligned operands).  PERSISTENT workgroups of 8 waves, two per
// CU, each walking its tiles (round-robin over the (M/128) x (N/128) tile grid, N fastest):
//  * waves 4-7 are PRODUCERS: global -> registers (two chunks ahead) -> BN+ReLU (activations only) -> ds_write into the
//    group-of-16 layout of the conv3x3 DMA kernel.  The prologue depends on the consumer layer, so the producer layer
//    cannot apply it; the weights take the same road.  The producers' chunk stream runs straight across tile
//    boundaries, so a tile's first chunks are already staged while the consumers store the previous tile.
//  * waves 0-3 are CONSUMERS: per chunk 4 sub-steps of 4 inline-asm ds_read_b128 (base + immediate) and 16 MFMAs - no
//    VALU, no VMEM, which is what the matrix pipe needs from its wave (tools/ubench/mfma_loop_variants.hip).
//  * ONE s_barrier per chunk for all 8 waves, double-buffered LDS (64 KB + the BN vectors).
// Why persistent: stamped (tools/ubench/ws_stamps.py), a one-tile-per-workgroup version kept the consumers' chunk loop at
// ~95 % matrix-pipe occupancy (two workgroups covering each other's barrier waits) but lost ~14 % of the kernel OUTSIDE
// the loop: workgroup launch, BN-vector copy, first-load latency and the store epilogue of every tile.
// lane -> (row, 16-B chunk) for loads and LDS writes: row = lane & 15 (+16 per group), chunk = 4 * half + (lane >> 4):
// a quarter-wave writes 256 contiguous bytes (conflict-free) and reads 64-B row segments.
#ifndef GNX_WS_STAMP
#define GNX_WS_STAMP 0
#endif
template <int GP>                  // groups of 16 rows per producer wave
struct C1Stage {
    float4 a[2 * GP], w[2 * GP];
    int k0;                        // first K index of the staged chunk (for the BN vectors at the stash)
};
constexpr int C1_KMAX = 2048;      // scale/shift vectors are staged in LDS up to this K

// DGBN (training backward: conv1's data gradient fused with norm1 -> relu1's backward): A = dL/d(conv1 out) [M][K = 128],
// W = conv1 weight transposed [N = cin][K]; the store, instead of writing the product g0[m][c], reads the layer's input
// x[m][c] (bn.x, the block buffer) and the block gradient out[m][c], and writes
//   out[m][c] += sc[c] g[m][c],   g = g0 where sc[c] x + sh[c] > 0, else 0            (densenet.py:35-37 backwards)
// and the column sums sum_m g, sum_m g xhat (xhat = (x - mean) invstd) of each 64-row half tile go to bn.slab
// [2 M/128][2][N] for dbeta / dgamma.  Five passes over [M][cin] (write g0, read g0, read x, read + write the gradient) become
// three.
struct C1BnBwd {
    const float* x; int ldx;
    const float* sc; const float* sh; const float* mean; const float* inv;
    float* slab;
};
template <bool ACT, bool POOL, int NP, bool DGBN = false>
__global__ __launch_bounds__(64 * (4 + NP), NP == 4 ? 4 : 3) void conv1x1_ws_kernel(const float* __restrict__ A, int lda,
                                                            const float* __restrict__ W, float* __restrict__ out,
                                                            int ldc, int K, int N, int tilesN, int T, int S_in,
                                                            long rows_in, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ oscale,
                                                            const float* __restrict__ oshift, C1BnBwd bn = C1BnBwd(),
                                                            int tile_runs = 0) {
    constexpr int OPB = 128 * 32 * 4;                      // bytes of one operand chunk
    extern __shared__ __attribute__((aligned(16))) float lds_f[];  // [buffer][A | B] then scale[K], shift[K]
    char* const lds = reinterpret_cast<char*>(lds_f);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = K >> 5;
    const int G = gridDim.x;
    const int bx = blockIdx.x;
    // Workgroup -> tile order.  One column tile per row tile (conv1 of a dense layer: N = 128): round-robin, so that at any
    // moment the chip reads one contiguous window of A.  Several column tiles per row tile all read the same A rows:
    //  * default (forward transitions): the column tiles of one row tile go to workgroups of the same round on the SAME XCD
    //    (workgroup b runs on XCD b % 8, one L2 per XCD) - measured 6.6 GB per transition launch before, 1.9x algorithmic;
    //  * tile_runs (conv1's fused data gradient, cin > 128): a workgroup takes a CONTIGUOUS run of the column-fastest tile
    //    list, so consecutive tiles of a workgroup share their dY rows.  The fabric-side fetch counter shows dY requested once
    //    per column tile either way (1.3-1.5x the algorithmic bytes at cin = 288 ... 896: the Infinity Cache serves it, the
    //    counter does not tell), but the kernel runs 5 % faster this way.
    const bool runs = tilesN > 1 && tile_runs;
    const int run0 = (int)((long)bx * T / G), run1 = (int)((long)(bx + 1) * T / G);
    const int mine = runs ? run1 - run0 : (T - bx + G - 1) / G;      // tiles of this workgroup (gridDim.x <= T)
    const int total = mine * nk;                           // chunks = barriers, the same number for all 8 waves
    const int jmap = (tilesN > 1 && G % (8 * tilesN) == 0)
                         ? tilesN * ((bx & 7) + 8 * (bx / (8 * tilesN))) + (bx >> 3) % tilesN : bx;
    auto tile_of = [&](int round) {
        if (runs) return round < mine ? run0 + round : T;
        const int base = round * G;
        return base + (base + G <= T ? jmap : bx);
    };
    // Every global load a wave issues beside the MFMA waves costs matrix-pipe time in proportion to its bytes, LDS reads
    // do not (tools/ubench/mfma_2x2.hip): the per-chunk scale/shift vectors come from an LDS copy made once per
    // workgroup instead of four more 1-KB loads per producer wave and chunk.
    float* sS = reinterpret_cast<float*>(lds + 4 * OPB);
    if (ACT) {
        const float ps = POOL ? 0.25f : 1.f;               // pool-first transition: relu(s x + t) / 4 = relu((s/4) x + t/4)
        for (int k = t; k < K; k += 64 * (4 + NP)) { sS[k] = ps * scale[k]; sS[K + k] = ps * shift[k]; }
        __syncthreads();
    }

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------ producer
        constexpr int GP = 8 / NP;                           // 16-row groups per producer wave (NP = 4: 2, NP = 8: 1)
        const int pw = wave - 4, lr = lane & 15, lc = lane >> 4;
        const int voA0 = ((16 * GP * pw + lr) * lda + 4 * lc) * 4, voA1 = voA0 + 64 * lda;
        const int voW0 = ((16 * GP * pw + lr) * K + 4 * lc) * 4, voW1 = voW0 + 64 * K;
        char* st = lds + (GP * pw) * 2048 + lc * 256 + lr * 16;     // + rg * 2048 + half * 1024 (+ OPB for W)
        auto bld = [](const __amdgpu_buffer_rsrc_t& r, int vo, int so) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
            return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                               __uint_as_float(v[3]));
        };
        // the producers' own (tile, chunk) iterator; behind the last chunk it re-reads the first tile into a buffer
        // nobody reads again (branch-free)
        int pround = 0, ptile = tile_of(0), pkt = 0;
        if constexpr (POOL) {
            static_assert(!POOL || NP == 4, "pooling producers own 32 rows each");
            // _Transition (densenet.py:47-54), pool-first: the staged row of pooled position m is the mean of the 4
            // activated source rows (2oy + {0,1}, 2ox + {0,1}).  16 activation loads per lane and chunk: one register
            // stage (the path is HBM-bound: 4x the bytes of the plain 1x1 for the same MFMA work).
            const int So = S_in >> 1, So2 = So * So;
            float4 pa[16], pwv[4];
            int pk0 = 0;
            auto srcrow = [&](int mrow) {              // pooled row index -> first of its 4 source rows
                const int img = mrow / So2, rem = mrow - img * So2;
                const int oy = rem / So, ox = rem - oy * So;
                return ((long)img * S_in + 2 * oy) * S_in + 2 * ox;
            };
            // Per-TILE state of the loads (resources, the lane's two source-row offsets): the integer divisions of srcrow()
            // cost ~60 VALU instructions, which the producers - VALU-starved beside the MFMA waves - paid on every chunk.
            __amdgpu_buffer_rsrc_t rA, rW;
            int vo0 = 0, vo1 = 0;
            auto settile = [&]() {
                const int tm = ptile / tilesN, tn = ptile - tm * tilesN;
                const long base = srcrow(tm * 128);
                long left = (rows_in - base) * lda - (lda - K);                 // floats behind the resource base
                const long span = (long)(4 * 128 + 2 * S_in + 4) * lda;
                if (left > span) left = span;
                rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + base * lda), 0, (int)(left * 4), 0x00020000);
                rW = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(W + (long)tn * 128 * K), 0, (N - tn * 128 < 128 ? N - tn * 128 : 128) * K * 4,
                    0x00020000);          // rows past N read as 0 (buffer bounds check): ragged last column tile
                vo0 = (int)(srcrow(tm * 128 + 32 * pw + lr) - base) * lda * 4 + lc * 16;
                vo1 = (int)(srcrow(tm * 128 + 32 * pw + lr + 16) - base) * lda * 4 + lc * 16;
            };
            settile();
            auto loadp = [&]() {
                const int kb = pkt << 7;
                pk0 = pkt << 5;
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // (row group, half)
                    const int vo = (q >> 1) ? vo1 : vo0, so = kb + (q & 1) * 64;
#pragma unroll
                    for (int u = 0; u < 4; ++u)        // the 4 pooled sources
                        pa[4 * q + u] = bld(rA, vo, so + ((u >> 1) * S_in + (u & 1)) * lda * 4);
                }
                pwv[0] = bld(rW, voW0, kb);
                pwv[1] = bld(rW, voW0, kb + 64);
                pwv[2] = bld(rW, voW1, kb);
                pwv[3] = bld(rW, voW1, kb + 64);
                if (++pkt == nk) {
                    pkt = 0;
                    ptile = tile_of(++pround);
                    if (ptile >= T) ptile = bx;
                    settile();
                }
            };
            auto stashp = [&](int buf) {
                char* d = st + buf * 2 * OPB;
                float4 sc[2], sh[2];
                if (ACT) {
                    const int k0 = pk0 + 4 * lc;
                    sc[0] = ld4(sS + k0);
                    sc[1] = ld4(sS + k0 + 16);
                    sh[0] = ld4(sS + K + k0);
                    sh[1] = ld4(sS + K + k0 + 16);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // the LDS copy of scale / shift carries the pool's 1/4 (exact: a power of two), so the mean is the sum
                    float4 sum = pa[4 * q];
                    if (ACT) sum = act4(sum, sc[q & 1], sh[q & 1]);
#pragma unroll
                    for (int u = 1; u < 4; ++u) {
                        float4 v = pa[4 * q + u];
                        if (ACT) v = act4(v, sc[q & 1], sh[q & 1]);
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    if (!ACT) sum = make_float4(0.25f * sum.x, 0.25f * sum.y, 0.25f * sum.z, 0.25f * sum.w);
                    *reinterpret_cast<float4*>(d + (q >> 1) * 2048 + (q & 1) * 1024) = sum;
                    *reinterpret_cast<float4*>(d + OPB + (q >> 1) * 2048 + (q & 1) * 1024) = pwv[q];
                }
            };
            __builtin_amdgcn_s_setprio(3);
            loadp();
            stashp(0);
            for (int g = 0; g < total; ++g) {
                lds_barrier();                              // chunk g published; consumers done with the other buffer
                loadp();
                stashp((g + 1) & 1);
            }
            return;
        }
        auto load = [&](C1Stage<GP>& s) {
            const int tm = ptile / tilesN, tn = ptile - tm * tilesN;
            // buffer loads (128-bit resource + 32-bit lane offset + scalar chunk offset)
            const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(A + (long)tm * 128 * lda), 0, (127 * lda + K) * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(W + (long)tn * 128 * K), 0, (N - tn * 128 < 128 ? N - tn * 128 : 128) * K * 4,
                    0x00020000);          // rows past N read as 0 (buffer bounds check): ragged last column tile
            const int kb = pkt << 7;                       // byte offset of the chunk
            s.k0 = pkt << 5;
#pragma unroll
            for (int q = 0; q < 2 * GP; ++q) s.a[q] = bld(rA, (q >> 1) ? voA1 : voA0, kb + (q & 1) * 64);
#pragma unroll
            for (int q = 0; q < 2 * GP; ++q) s.w[q] = bld(rW, (q >> 1) ? voW1 : voW0, kb + (q & 1) * 64);
            if (++pkt == nk) { pkt = 0; ptile = tile_of(++pround); }
            if (ptile >= T) ptile = bx;
        };
#if GNX_WS_STAMP
        long seg[4] = {0, 0, 0, 0};
        long seg_t = 0;
#define GNX_SEG(i) do { const long now = __builtin_amdgcn_s_memtime(); seg[i] += now - seg_t; seg_t = now; } while (0)
#else
#define GNX_SEG(i)
#endif
        auto stash = [&](const C1Stage<GP>& s, int buf) {
            GNX_SEG(0);                                     // barrier release -> loads issued
            char* d = st + buf * 2 * OPB;
            float4 sc[2], sh[2];
            if (ACT) {
                const int k0 = s.k0 + 4 * lc;
                sc[0] = ld4(sS + k0);
                sc[1] = ld4(sS + k0 + 16);
                sh[0] = ld4(sS + K + k0);
                sh[1] = ld4(sS + K + k0 + 16);
            }
#if GNX_WS_STAMP
            if (GP == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            GNX_SEG(1);                                     // operands of the stash have arrived (loads + BN vectors)
#endif
#pragma unroll
            for (int q = 0; q < 2 * GP; ++q) {
                float4 v = s.a[q];
                if (ACT) v = act4(v, sc[q & 1], sh[q & 1]);
                *reinterpret_cast<float4*>(d + (q >> 1) * 2048 + (q & 1) * 1024) = v;
                *reinterpret_cast<float4*>(d + OPB + (q >> 1) * 2048 + (q & 1) * 1024) = s.w[q];
            }
            GNX_SEG(2);                                     // activation + LDS writes issued
#if GNX_WS_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            GNX_SEG(3);                                     // LDS writes done
#endif
        };
        __builtin_amdgcn_s_setprio(3);
        C1Stage<GP> s0, s1;
        load(s0);
        load(s1);
        stash(s0, 0);
#if GNX_WS_STAMP
        long p_wait = 0;
        const long p_begin = __builtin_amdgcn_s_memtime();
#define GNX_PBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long tb = __builtin_amdgcn_s_memtime(); \
                        asm volatile("s_barrier" ::: "memory"); seg_t = __builtin_amdgcn_s_memtime(); p_wait += seg_t - tb; } while (0)
#else
#define GNX_PBAR() lds_barrier()
#endif
        for (int g = 0; g < total; g += 2) {
            GNX_PBAR();                                     // chunk g published; consumers done with buffer 1
            load(s0);
            stash(s1, 1);
            if (g + 1 >= total) break;
            GNX_PBAR();                                     // chunk g+1 published; consumers done with buffer 0
            load(s1);
            stash(s0, 0);
        }
#undef GNX_PBAR
#if GNX_WS_STAMP
        if (lane == 0 && pw == 0) {      // debug build: producer wave 4 of each workgroup -> (barrier wait, total) cycles
            const long p_total = __builtin_amdgcn_s_memtime() - p_begin;
            float* dbg = out + (long)(blockIdx.x / tilesN) * 128 * ldc + (blockIdx.x % tilesN) * 128;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[4 * ldc + 0] = (float)p_wait;
            dbg[4 * ldc + 1] = (float)p_total;
            dbg[4 * ldc + 2] = (float)total;
            for (int q = 0; q < 4; ++q) dbg[4 * ldc + 3 + q] = (float)seg[q];
        }
#endif
        return;
    }
    // ---------------------------------------------------------------------------------------------- consumer
    const int h = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lb = lds_addr(lds);
    // rows 64 wm + 32 mt + i of A, 64 wn + 32 nt + i of B; mt / nt = +2 groups = +4096 B
    const unsigned fA = lb + (4 * wm + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    const unsigned fB = lb + OPB + (4 * wn + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    int g = 0;
#if GNX_WS_STAMP
    long c_wait = 0;
    const long c_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int round = 0, tile = tile_of(0); tile < T; tile = tile_of(++round)) {
        f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
        for (int kt = 0; kt < nk; ++kt, ++g) {
#if GNX_WS_STAMP
            const long tb0 = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_barrier" ::: "memory");
#if GNX_WS_STAMP
            c_wait += __builtin_amdgcn_s_memtime() - tb0;
#endif
            const unsigned a = fA + (g & 1) * 2 * OPB, b = fB + (g & 1) * 2 * OPB;
            f32x4 a0 = lds_read4<0>(a), a1 = lds_read4<4096>(a), b0 = lds_read4<0>(b), b1 = lds_read4<4096>(b);
            static_for<0, 4>([&](auto s_c) {
                constexpr int sstep = decltype(s_c)::value;
                f32x4 na0, na1, nb0, nb1;
                if constexpr (sstep < 3) {
                    na0 = lds_read4<(sstep + 1) * 512>(a);
                    na1 = lds_read4<(sstep + 1) * 512 + 4096>(a);
                    nb0 = lds_read4<(sstep + 1) * 512>(b);
                    nb1 = lds_read4<(sstep + 1) * 512 + 4096>(b);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc11, 0, 0, 0);
                }
                if constexpr (sstep < 3) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
            });
        }
        const int tm = tile / tilesN, tn = tile - tm * tilesN;
        // BUFFER stores: like the loads, a global_store issued beside MFMA waves costs the matrix pipe, a buffer_store does
        // not (resource = this tile's 128 output rows; lane offset constant, row offset scalar)
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (long)tm * 128 * ldc, 0, 128 * ldc * 4,
                                                                            0x00020000);
        auto store = [&](f32x16& acc, int mt, int nt) {
            const int col = tn * 128 + 64 * wn + 32 * nt + i;
            if (col - i >= N) return;                      // 32 | N: whole fragments in or out
            if (oscale) {                                  // (buffer loads: see the stores below)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(oscale), 0, N * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(oshift), 0, N * 4, 0x00020000);
                const float osc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, col * 4, 0, 0));
                const float osh = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rt, col * 4, 0, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(fmaf(acc[r], osc, osh), 0.f);
            }
            const int vo = ((64 * wm + 4 * h) * ldc + col) * 4;
            const int so = 32 * mt * ldc * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r]), ro, vo, so + ((r & 3) + 8 * (r >> 2)) * ldc * 4, 0);
        };
        if constexpr (DGBN) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(bn.x + (long)tm * 128 * bn.ldx), 0, 128 * bn.ldx * 4, 0x00020000);
            auto bwd = [&](f32x16& a0, f32x16& a1, int nt) {           // the two row halves (mt = 0, 1) of column fragment nt
                const int col = tn * 128 + 64 * wn + 32 * nt + i;
                if (col - i >= N) return;
                const float sc = bn.sc[col], sh = bn.sh[col], mu = bn.mean[col], is = bn.inv[col];
                const int vox = ((64 * wm + 4 * h) * bn.ldx + col) * 4, vod = ((64 * wm + 4 * h) * ldc + col) * 4;
                float sb = 0.f, sg = 0.f;
                auto half = [&](f32x16& acc, int mt) {
                    float xv[16], dv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * mt + (r & 3) + 8 * (r >> 2);
                        xv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vox, row * bn.ldx * 4, 0));
                        dv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, vod, row * ldc * 4, 0));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * mt + (r & 3) + 8 * (r >> 2);
                        const float g = fmaf(xv[r], sc, sh) > 0.f ? acc[r] : 0.f;
                        sb += g;
                        sg = fmaf(g, (xv[r] - mu) * is, sg);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(g, sc, dv[r])), ro, vod, row * ldc * 4, 0);
                    }
                };
                half(a0, 0);
                half(a1, 1);
                sb += __shfl_xor(sb, 32);
                sg += __shfl_xor(sg, 32);
                if (h == 0) {
                    float* sl = bn.slab + ((long)(2 * tm + wm) * 2) * N + col;
                    sl[0] = sb;
                    sl[N] = sg;
                }
            };
            bwd(acc00, acc10, 0);
            bwd(acc01, acc11, 1);
            continue;
        }
        store(acc00, 0, 0);
        store(acc01, 0, 1);
        store(acc10, 1, 0);
        store(acc11, 1, 1);
    }
#if GNX_WS_STAMP
    if (lane == 0) {                     // debug build: consumer waves -> (barrier wait, total) cycles over all their tiles
        const long c_total = __builtin_amdgcn_s_memtime() - c_begin;
        float* dbg = out + (long)(blockIdx.x / tilesN) * 128 * ldc + (blockIdx.x % tilesN) * 128;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[wave * ldc + 0] = (float)c_wait;
        dbg[wave * ldc + 1] = (float)c_total;
    }
#endif
}

// slab[nblk][2][C] -> part[R][2][C]: block y sums its contiguous share of the slabs in a fixed order (bn_bwd_reduce_kernel
// of bn.hip then finishes over the R rows)
__global__ __launch_bounds__(256) void slab_fold_kernel(const float* __restrict__ slab, long nblk, int C, int R,
                                                        float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long per = (nblk + R - 1) / R, b0 = blockIdx.y * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    float s1 = 0.f, s2 = 0.f;
    if (c < C)
        for (long b = b0 + sl; b < b1; b += 4) {
            s1 += slab[(b * 2 + 0) * C + c];
            s2 += slab[(b * 2 + 1) * C + c];
        }
    red[0][sl][cl] = s1;
    red[1][sl][cl] = s2;
    __syncthreads();
    if (sl == 0 && c < C) {
        part[((long)blockIdx.y * 2 + 0) * C + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        part[((long)blockIdx.y * 2 + 1) * C + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}
// out[c] (+)= sum over the R rows of part[.][q][c], fixed order: 64 columns x 8 row lanes per workgroup, a lane sums its
// contiguous eighth in index order, the eighths are combined in index order (one lane per column took 34 us per call - 256
// dependent loads - 58 times a step)
__global__ __launch_bounds__(512) void slab_finish_kernel(const float* __restrict__ part, int R, int C, float* dbeta,
                                                          float* dgamma, int accumulate) {
    __shared__ float red[2][8][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int per = (R + 7) / 8, b0 = rl * per, b1 = b0 + per < R ? b0 + per : R;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v1[u] = part[((long)(b + u) * 2 + 0) * C + c]; v2[u] = part[((long)(b + u) * 2 + 1) * C + c]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
        }
        for (; b < b1; ++b) { s1 += part[((long)b * 2 + 0) * C + c]; s2 += part[((long)b * 2 + 1) * C + c]; }
    }
    red[0][rl][cl] = s1;
    red[1][rl][cl] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { t1 += red[0][u][cl]; t2 += red[1][u][cl]; }
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + t1 : t1;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + t2 : t2;
    }
}

// ------------------------------------------------------------------------------------------------ conv1x1, LDS-clamp form
// The eval forward's _DenseLayer norm1 -> relu1 -> conv1 (-> norm2 -> relu2) with NO vector ALU work on the staging side.
//   relu(sc x + sh) = sc clamp(x) + sh,  clamp(x) = max(x, t) for sc > 0, min(x, t) for sc < 0,  t = -sh / sc,
// so with Wf[n][k] = W[n][k] sc[k] and b[n] = sum_k W[n][k] sh[k] (conv1x1_fold_kernel, once per weight/BN update)
//   y[m][n] = sum_k Wf[n][k] clamp_k(x[m][k]) + b[n].
// The producers of conv1x1_ws_kernel are bounded by their VALU issue rate beside the MFMA waves (one slot per ~120
// cycles whatever the instruction; tools/ubench/mfma_2x2.hip), and a VALU clamp changes nothing about that.  LDS
// instructions are issued every ~19 cycles in the same place, and the LDS has float atomics: here the raw activations go
// global -> LDS by DMA (no registers) and are clamped IN the LDS by ds_max_f32 / ds_min_f32 against per-lane bounds
// (16 + 16 per producer wave and chunk, ~780 cycles against the consumers' 4 096).  Per chunk and producer wave: 8 DMA
// instructions, 4 16-B loads of bounds, 32 LDS atomics, no VALU.
// LDS: three A stages (consumed | being clamped | DMA in flight) + two B stages (consumed | in flight: weights come from
// L2), 16 KB each = 80 KB, two workgroups per CU.  vmcnt retires in order, so per window the issue order is
// B(q+1), bounds(q+2), A(q+2): `vmcnt(12)` = A(q+1) and its bounds have landed, `vmcnt(8)` = B(q+1) has.
template <int OFF, int MIN>
__device__ __forceinline__ void lds_fclamp(unsigned addr, float v) {
    if (MIN) asm volatile("ds_min_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    else asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
struct C1Bounds { f32x4 lo0, lo1, hi0, hi1; };              // this lane's 8 k of a chunk: k = 4 c + (lane & 3), c = 0..7

__global__ __launch_bounds__(512, 4) void conv1x1_clamp_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ Wf,
                                                               const float* __restrict__ bounds, float* __restrict__ out,
                                                               int ldc, int K, int N, int tilesN, int T,
                                                               const float* __restrict__ oscale,
                                                               const float* __restrict__ oshift) {
    constexpr int OPB = 128 * 32 * 4;                      // bytes of one operand chunk
    constexpr int BOFF = 3 * OPB;                          // the two B stages lie behind the three A stages
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char* const lds = reinterpret_cast<char*>(lds_f);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = K >> 5;
    const int G = gridDim.x;
    const int bx = blockIdx.x;
    const int mine = (T - bx + G - 1) / G;                 // tiles of this workgroup (gridDim.x <= T)
    const int total = mine * nk;                           // chunks = barriers, the same number for all 8 waves
    const int jmap = (tilesN > 1 && G % (8 * tilesN) == 0)
                         ? tilesN * ((bx & 7) + 8 * (bx / (8 * tilesN))) + (bx >> 3) % tilesN : bx;
    auto tile_of = [&](int round) {
        const int base = round * G;
        return base + (base + G <= T ? jmap : bx);
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------ producer
        const int pw = wave - 4;
        const unsigned voA = ((unsigned)(lane & 15) * lda + 4 * (lane >> 4)) * 4u;      // per-lane constant byte offsets
        const unsigned voW = ((unsigned)(lane & 15) * K + 4 * (lane >> 4)) * 4u;
        const unsigned vob = (lane & 3) * 32u;             // this lane's 8 bounds within a chunk's 32 (bytes)
        const unsigned cl = lds_addr(lds) + pw * 4096 + lane * 4;                       // clamp address in stage 0
        // (tile, chunk) iterators; behind the last chunk they re-read this workgroup's first tile into stages nobody reads
        struct It { int round, tile, kt; };
        It ia = {0, tile_of(0), 0}, ib = ia;
        auto advance = [&](It& it) {
            if (++it.kt == nk) { it.kt = 0; it.tile = tile_of(++it.round); }
            if (it.tile >= T) it.tile = bx;
        };
        // DMA as BUFFER loads (128-bit resource in SGPRs + per-lane constant 32-bit offset + scalar offset): beside saturated
        // MFMA waves a global_load costs the matrix pipe ~40 cycles per wave-instruction, a buffer_load nothing
        // (tools/ubench/mfma_2x2.hip: 16 per chunk -> 117 / 130 (LDS form) against 152 TFLOP/s) - and no VALU address
        // arithmetic either.  M0 = LDS destination of lane 0; the scalar offset moves only the source.
        auto dma4 = [&](const __amdgpu_buffer_rsrc_t& r, int so, int rowskip, unsigned vo, unsigned m0) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         ::"s"(m0), "v"(vo), "s"(r), "s"(so) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         ::"s"(m0 + 1024), "v"(vo), "s"(r), "s"(so + 64) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         ::"s"(m0 + 2048), "v"(vo), "s"(r), "s"(so + rowskip) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         ::"s"(m0 + 3072), "v"(vo), "s"(r), "s"(so + rowskip + 64) : "memory");
        };
        const unsigned ldsb = lds_addr(lds) + pw * 4096;
        auto issue_a = [&](int stage) {
            const int tm = ia.tile / tilesN;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(A + ((long)tm * 128 + 32 * pw) * lda), 0, (31 * lda + K) * 4, 0x00020000);
            dma4(r, ia.kt << 7, lda * 64, voA, ldsb + stage * OPB);
            advance(ia);
        };
        auto issue_b = [&](int stage) {
            const int tn = ib.tile % tilesN;
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(Wf + ((long)tn * 128 + 32 * pw) * K), 0, 32 * K * 4, 0x00020000);
            dma4(r, ib.kt << 7, K * 64, voW, ldsb + BOFF + stage * OPB);
            advance(ib);
        };
        // bounds of the chunk `ia` points at: asm loads, so they keep their place in the vmcnt order
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bounds), 0, 8 * K, 0x00020000);
        auto load_bounds = [&](C1Bounds& b) {
            const int so = ia.kt << 7, soh = so + 4 * K;
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(b.lo0) : "v"(vob), "s"(rb), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(b.lo1) : "v"(vob), "s"(rb), "s"(so) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(b.hi0) : "v"(vob), "s"(rb), "s"(soh) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(b.hi1) : "v"(vob), "s"(rb), "s"(soh) : "memory");
        };
#if GNX_WS_STAMP
        long seg[4] = {0, 0, 0, 0};
        long seg_t = 0, p_wait = 0;
#endif
        auto clamp = [&](int stage, C1Bounds& b) {
            // A(stage) and b have landed once at most 12 younger VMEM operations are outstanding
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(b.lo0), "+v"(b.lo1), "+v"(b.hi0), "+v"(b.hi1)::"memory");
            const unsigned a = cl + stage * OPB;
            static_for<0, 16>([&](auto ic) {                // instruction i: rows 16 (i >> 3) + (lane >> 2), k = 4 (i & 7) + (lane & 3)
                constexpr int i = decltype(ic)::value, c = i & 7;
                lds_fclamp<256 * i, 0>(a, c < 4 ? b.lo0[c & 3] : b.lo1[c & 3]);
                lds_fclamp<256 * i, 1>(a, c < 4 ? b.hi0[c & 3] : b.hi1[c & 3]);
            });
        };
#if GNX_WS_STAMP
#define GNX_PBAR() do { asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); GNX_SEG(3); const long tb = __builtin_amdgcn_s_memtime(); \
                        asm volatile("s_barrier" ::: "memory"); seg_t = __builtin_amdgcn_s_memtime(); p_wait += seg_t - tb; } while (0)
#else
#define GNX_PBAR() lds_barrier()
#endif
        __builtin_amdgcn_s_setprio(3);
        C1Bounds b0, b1;
        issue_b(0);
        load_bounds(b0);
        issue_a(0);
        load_bounds(b1);
        issue_a(1);
        {
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(b0.lo0), "+v"(b0.lo1), "+v"(b0.hi0), "+v"(b0.hi1)::"memory");
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value, c = i & 7;
                lds_fclamp<256 * i, 0>(cl, c < 4 ? b0.lo0[c & 3] : b0.lo1[c & 3]);
                lds_fclamp<256 * i, 1>(cl, c < 4 ? b0.hi0[c & 3] : b0.hi1[c & 3]);
            });
        }
        int s3 = 0;                                         // q % 3
#if GNX_WS_STAMP
        const long p_begin = __builtin_amdgcn_s_memtime();
        seg_t = p_begin;
#endif
        for (int q = 0; q < total; q += 2) {
            GNX_PBAR();                                     // chunk q published; consumers are done with chunk q-1
            int s1 = s3 + 1 == 3 ? 0 : s3 + 1, s2 = s1 + 1 == 3 ? 0 : s1 + 1;
            issue_b((q + 1) & 1);
            GNX_SEG(0);
            load_bounds(b0);
            GNX_SEG(1);
            issue_a(s2);
            GNX_SEG(2);
            clamp(s1, b1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 // B(q+1) has landed
            s3 = s1;
            if (q + 1 >= total) break;
            GNX_PBAR();
            s1 = s3 + 1 == 3 ? 0 : s3 + 1; s2 = s1 + 1 == 3 ? 0 : s1 + 1;
            issue_b(q & 1);
            GNX_SEG(0);
            load_bounds(b1);
            GNX_SEG(1);
            issue_a(s2);
            GNX_SEG(2);
            clamp(s1, b0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            s3 = s1;
        }
#undef GNX_PBAR
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // no DMA may outlive the workgroup's LDS
#if GNX_WS_STAMP
        if (lane == 0 && pw == 0) {      // debug build: producer wave 4 of each workgroup -> (barrier wait, total) cycles
            const long p_total = __builtin_amdgcn_s_memtime() - p_begin;
            float* dbg = out + (long)(blockIdx.x / tilesN) * 128 * ldc + (blockIdx.x % tilesN) * 128;
            dbg[4 * ldc + 0] = (float)p_wait;
            dbg[4 * ldc + 1] = (float)p_total;
            dbg[4 * ldc + 2] = (float)total;
            for (int q = 0; q < 4; ++q) dbg[4 * ldc + 3 + q] = (float)seg[q];
        }
#endif
        return;
    }
    // ---------------------------------------------------------------------------------------------- consumer
    const int h = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lb = lds_addr(lds);
    const unsigned fA = lb + (4 * wm + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    const unsigned fB = lb + BOFF + (4 * wn + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    int g = 0, g3 = 0;
#if GNX_WS_STAMP
    long c_wait = 0;
    const long c_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int round = 0, tile = tile_of(0); tile < T; tile = tile_of(++round)) {
        f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
        for (int kt = 0; kt < nk; ++kt, ++g) {
#if GNX_WS_STAMP
            const long tb0 = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_barrier" ::: "memory");
#if GNX_WS_STAMP
            c_wait += __builtin_amdgcn_s_memtime() - tb0;
#endif
            const unsigned a = fA + g3 * OPB, b = fB + (g & 1) * OPB;
            g3 = g3 + 1 == 3 ? 0 : g3 + 1;
            f32x4 a0 = lds_read4<0>(a), a1 = lds_read4<4096>(a), b0 = lds_read4<0>(b), b1 = lds_read4<4096>(b);
            static_for<0, 4>([&](auto s_c) {
                constexpr int sstep = decltype(s_c)::value;
                f32x4 na0, na1, nb0, nb1;
                if constexpr (sstep < 3) {
                    na0 = lds_read4<(sstep + 1) * 512>(a);
                    na1 = lds_read4<(sstep + 1) * 512 + 4096>(a);
                    nb0 = lds_read4<(sstep + 1) * 512>(b);
                    nb1 = lds_read4<(sstep + 1) * 512 + 4096>(b);
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc11, 0, 0, 0);
                }
                if constexpr (sstep < 3) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
            });
        }
        const int tm = tile / tilesN, tn = tile - tm * tilesN;
        auto store = [&](f32x16& acc, int mt, int nt) {
            const int col = tn * 128 + 64 * wn + 32 * nt + i;
            const float osc = oscale[col], osh = oshift[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaxf(fmaf(acc[r], osc, osh), 0.f);
            float* o = out + (long)(tm * 128 + 64 * wm + 32 * mt + 4 * h) * ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(long)(((r & 3) + 8 * (r >> 2)) * ldc)] = acc[r];
        };
        store(acc00, 0, 0);
        store(acc01, 0, 1);
        store(acc10, 1, 0);
        store(acc11, 1, 1);
    }
#if GNX_WS_STAMP
    if (lane == 0) {                     // debug build: consumer waves -> (barrier wait, total) cycles over all their tiles
        const long c_total = __builtin_amdgcn_s_memtime() - c_begin;
        float* dbg = out + (long)(blockIdx.x / tilesN) * 128 * ldc + (blockIdx.x % tilesN) * 128;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[wave * ldc + 0] = (float)c_wait;
        dbg[wave * ldc + 1] = (float)c_total;
    }
#endif
}

// Wf[n][k] = W[n][k] sc[k];  bounds[0][.] / [1][.] = lower / upper clamp per k, stored per 32-k chunk as [k & 3][k >> 2]
// (a producer lane's 8 values contiguous);  out_shift_f[n] = out_scale[n] * sum_k W[n][k] sh[k] + out_shift[n].
// Channels whose scale is 0 (or so small that -sh/sc is not finite) are constants relu(sh): weight 0, value into the sum.
__global__ __launch_bounds__(256) void conv1x1_fold_kernel(const float* __restrict__ W, const float* __restrict__ sc,
                                                           const float* __restrict__ sh,
                                                           const float* __restrict__ osc, const float* __restrict__ osh,
                                                           float* __restrict__ Wf, float* __restrict__ bounds,
                                                           float* __restrict__ oshf, int N, int K) {
    __shared__ double red[256];
    const int n = blockIdx.x, t = threadIdx.x;
    double sum = 0.0;
    for (int k = t; k < K; k += 256) {
        const float s = sc[k], b = sh[k], tt = -b / s;
        const bool ok = s != 0.f && fabsf(tt) <= 3.0e38f;          // false for NaN too
        const float w = W[(long)n * K + k];
        Wf[(long)n * K + k] = ok ? w * s : 0.f;
        sum += (double)w * (double)(ok ? b : fmaxf(b, 0.f));
        if (n == 0) {
            const int pos = (k & ~31) + (k & 3) * 8 + ((k & 31) >> 2);
            bounds[pos] = ok && s > 0.f ? tt : -INFINITY;
            bounds[K + pos] = ok && s < 0.f ? tt : INFINITY;
        }
    }
    red[t] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) oshf[n] = (float)((double)osc[n] * red[0] + (double)osh[n]);
}

}  // namespace

// out[M][N] (ldc) = act(A[M][K] (lda)) . W[N][K]^T ; pool != 0: A is on an S_in x S_in grid per image and M counts the
// (S_in/2)^2 pooled positions.  scale/shift may both be NULL (no activation).
// conv1's fused data gradient: workgroups take contiguous runs of the column-fastest tile list (the forward transitions
// keep their XCD placement instead).  Same box, same run: 52.5 -> 50.0 ms per f-trained step; the forward transitions
// measured no better with runs (2.97 ms either way within noise), so they keep their order.
static int c1_tile_runs() { return 1; }

static int conv1x1_launch(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                          const float* scale, const float* shift, int pool, int S_in, const float* oscale,
                          const float* oshift, hipStream_t stream, float* workspace = nullptr) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0 || lda < K || ldc < N || (!scale) != (!shift) ||
        (!oscale) != (!oshift))
        return GNX_ERR_BAD_ARG;
    if (pool && (S_in < 2)) return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    const int vecA = al16(A) && lda % 4 == 0 && K % 4 == 0 && (!scale || (al16(scale) && al16(shift)));
    const int vecW = al16(W) && K % 4 == 0;
    dim3 grid(gnx_cdiv(M, C1_BM), gnx_cdiv(N, C1_BN));
    const bool fast = vecA && vecW;
    // small M with a workspace: K split over blockIdx.z (a batch-32 training step: 40 of its 58 conv1 launches ran 16 or 4
    // workgroups through 8-31 chunks each: 30-40 us of latency per launch)
    const int splits = (workspace && fast && !pool && al16(out) && ldc % 4 == 0 && al16(workspace) &&
                        (!oscale || (al16(oscale) && al16(oshift)))) ? conv1x1_small_splits(M, N, K) : 0;
    if (splits > 1) {
        const int ksplit = ((K / 32 + splits - 1) / splits) * 32;
        const int nz = (K + ksplit - 1) / ksplit;
        dim3 gs(grid.x, grid.y, nz);
        conv1x1_kernel<false, true><<<gs, 256, 0, stream>>>(A, lda, W, out, ldc, M, N, K, scale, shift, S_in, vecA, vecW, nullptr,
                                                            nullptr, K, ksplit, workspace);
        long blocks = gnx_cdiv(M * (N / 4), 256);
        if (blocks > 1024) blocks = 1024;
        conv1x1_split_reduce_kernel<<<(unsigned)blocks, 256, 0, stream>>>(workspace, nz, M, N, out, ldc, oscale, oshift);
        return gnx_launch_status();
    }
    if (fast && M % 128 == 0 && N % 32 == 0 && K % 32 == 0 && K <= C1_KMAX && (!pool || (S_in % 2 == 0 && scale)) &&
        4 * M < (1L << 31) && lda < (1 << 16) && ldc < (1 << 16)) {     // int row / lane offsets
        const size_t lds_ws = 4 * 128 * 32 * 4 + (scale ? 8 * (size_t)K : 0);
        static bool conf = false;
        if (!conf) {
            const int mx = 4 * 128 * 32 * 4 + 8 * C1_KMAX;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<true, false, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<false, false, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<true, false, 8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<false, false, 8>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<true, true, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mx) != hipSuccess)
                return GNX_ERR_LAUNCH;
            conf = true;
        }
        const int tilesN = (N + 127) / 128;
        const long T = (M / 128) * tilesN;
        // NP = producer waves: 4 (two 8-wave workgroups per CU) or 8 (one 12-wave workgroup per CU)
        // (4 producer waves, two 8-wave workgroups per CU: 8 in one 12-wave workgroup per CU measured slower)
        const int per_cu = 2;
        const int wgs = (int)(T < 256 * per_cu ? T : 256 * per_cu);
#define GNX_WS(ACTV, POOLV, NPV)                                                                                     \
    conv1x1_ws_kernel<ACTV, POOLV, NPV><<<wgs, 64 * (4 + NPV), lds_ws, stream>>>(                                     \
        A, (int)lda, W, out, (int)ldc, K, N, tilesN, (int)T, S_in, 4 * M, scale, shift, oscale, oshift)
        if (pool) GNX_WS(true, true, 4);
        else if (scale) GNX_WS(true, false, 4);
        else GNX_WS(false, false, 4);
#undef GNX_WS
        return gnx_launch_status();
    }
#define GNX_C1(P, F)                                                                                               \
    conv1x1_kernel<P, F><<<grid, 256, 0, stream>>>(A, lda, W, out, ldc, M, N, K, scale, shift, S_in, vecA, vecW, \
                                                   oscale, oshift)
    if (pool) { if (fast) GNX_C1(true, true); else GNX_C1(true, false); }
    else { if (fast) GNX_C1(false, true); else GNX_C1(false, false); }
#undef GNX_C1
    return gnx_launch_status();
}

GNX_EXPORT int gnx_conv1x1_bnrelu(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                                  const float* scale, const float* shift, int pool, int S_in, hipStream_t stream) {
    return conv1x1_launch(A, lda, W, out, ldc, M, N, K, scale, shift, pool, S_in, nullptr, nullptr, stream);
}

// floats of workspace gnx_conv1x1_bnrelu_ws can use for this shape (0: it would not split)
GNX_EXPORT long gnx_conv1x1_workspace(long M, int N, int K) {
    const int s = conv1x1_small_splits(M, N, K);
    return s > 1 ? (long)s * M * N : 0;
}
// gnx_conv1x1_bnrelu (pool = 0) given a workspace: for matrices of few row tiles (small training batches) the K range is
// split over workgroups and the partial tiles are summed in a fixed order.  workspace may be NULL (= the plain entry point).
GNX_EXPORT int gnx_conv1x1_bnrelu_ws(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                                     const float* scale, const float* shift, float* workspace, hipStream_t stream) {
    return conv1x1_launch(A, lda, W, out, ldc, M, N, K, scale, shift, 0, 0, nullptr, nullptr, stream, workspace);
}

// As gnx_conv1x1_bnrelu (pool = 0) with the CONSUMER's folded BN + ReLU applied at the store:
// out[m][n] = relu(out_scale[n] * (act(A) . W^T)[m][n] + out_shift[n]).  The eval forward stores the bottleneck this way,
// so the 3x3 convolution that follows needs no prologue and can stream its operand global -> LDS by DMA.
GNX_EXPORT int gnx_conv1x1_bnrelu_act(const float* A, long lda, const float* W, float* out, long ldc, long M, int N,
                                      int K, const float* scale, const float* shift, const float* out_scale,
                                      const float* out_shift, hipStream_t stream) {
    if (!out_scale || !out_shift) return GNX_ERR_BAD_ARG;
    return conv1x1_launch(A, lda, W, out, ldc, M, N, K, scale, shift, 0, 0, out_scale, out_shift, stream);
}

// Fold norm1 (scale, shift: the PRODUCER-side BN of this conv) and the consumer's norm2 (out_scale, out_shift) into the
// operands of gnx_conv1x1_clamped_act: Wf [N][K], bounds [2][K], out_shift_f [N].  32 | K.  Once per weight / BN update.
GNX_EXPORT int gnx_conv1x1_fold_clamp(const float* W, const float* scale, const float* shift, const float* out_scale,
                                      const float* out_shift, float* Wf, float* bounds, float* out_shift_f, int N, int K,
                                      hipStream_t stream) {
    if (!W || !scale || !shift || !out_scale || !out_shift || !Wf || !bounds || !out_shift_f || N <= 0 || K <= 0)
        return GNX_ERR_BAD_ARG;
    if (K % 32 != 0) return GNX_ERR_UNSUPPORTED;
    conv1x1_fold_kernel<<<N, 256, 0, stream>>>(W, scale, shift, out_scale, out_shift, Wf, bounds, out_shift_f, N, K);
    return gnx_launch_status();
}

// out[m][n] = relu(out_scale[n] * sum_k Wf[n][k] clamp_k(A[m][k]) + out_shift_f[n]) == gnx_conv1x1_bnrelu_act on the
// unfolded operands up to rounding (conv1x1_clamp_kernel).  Whole tiles only: 128 | M, 128 | N, 32 | K, 16-B aligned
// operands; anything else returns GNX_ERR_UNSUPPORTED (run gnx_conv1x1_bnrelu_act then).
GNX_EXPORT int gnx_conv1x1_clamped_act(const float* A, long lda, const float* Wf, const float* bounds, float* out, long ldc,
                                       long M, int N, int K, const float* out_scale, const float* out_shift_f,
                                       hipStream_t stream) {
    if (!A || !Wf || !bounds || !out || !out_scale || !out_shift_f || M < 0 || N <= 0 || K <= 0 || lda < K || ldc < N)
        return GNX_ERR_BAD_ARG;
    if (M % 128 != 0 || N % 128 != 0 || K % 32 != 0 || !al16(A) || !al16(Wf) || !al16(bounds) || lda % 4 != 0 ||
        M >= (1L << 29) || lda >= (1 << 16) || ldc >= (1 << 16))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    constexpr int lds_bytes = 5 * 128 * 32 * 4;
    static bool conf = false;
    if (!conf) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_clamp_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return GNX_ERR_LAUNCH;
        conf = true;
    }
    const int tilesN = N / 128;
    const long T = (M / 128) * tilesN;
    const int wgs = (int)(T < 512 ? T : 512);
    conv1x1_clamp_kernel<<<wgs, 512, lds_bytes, stream>>>(A, (int)lda, Wf, bounds, out, (int)ldc, K, N, tilesN, (int)T,
                                                          out_scale, out_shift_f);
    return gnx_launch_status();
}

// floats of workspace gnx_conv1x1_dgrad_bnrelu_bwd needs
constexpr int C1_DGBN_R = 128;
GNX_EXPORT long gnx_conv1x1_dgrad_bn_workspace(long M, int N) {
    return (2 * (M / 128) * 2 + (long)C1_DGBN_R * 2) * (long)N;
}

// conv1's data gradient fused with norm1 -> relu1's backward (eval statistics), accumulated into the block gradient:
//   dX[m][c] += scale[c] g[m][c],  g = (dY . Wt^T)[m][c] where scale[c] X[m][c] + shift[c] > 0, else 0     (c < N = cin)
//   dbeta[c] (+)= sum_m g,  dgamma[c] (+)= sum_m g (X - mean) invstd
// == gnx_conv1x1_bnrelu(dY, Wt) followed by gnx_bn_relu_bwd(relu = 1, training = 0, dx_accumulate = 1), in three passes over
// [M][N] instead of five.  dY [M][K] (lddy), Wt [N][K] (gnx_transpose_weight), X / dX [M][>= N] (ldx / lddx).
// Whole tiles only (128 | M, 32 | N, 32 | K, aligned): GNX_ERR_UNSUPPORTED otherwise - run the two calls then.
GNX_EXPORT int gnx_conv1x1_dgrad_bnrelu_bwd(const float* dY, long lddy, const float* Wt, const float* X, long ldx, float* dX,
                                            long lddx, long M, int N, int K, const float* scale, const float* shift,
                                            const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                            int accumulate, float* workspace, hipStream_t stream) {
    if (!dY || !Wt || !X || !dX || !scale || !shift || !mean || !invstd || !workspace || M < 0 || N <= 0 || K <= 0 ||
        lddy < K || ldx < N || lddx < N)
        return GNX_ERR_BAD_ARG;
    if (M % 128 != 0 || N % 32 != 0 || K % 32 != 0 || K > C1_KMAX || !al16(dY) || !al16(Wt) || lddy % 4 != 0 ||
        4 * M >= (1L << 31) || lddy >= (1 << 16) || ldx >= (1 << 16) || lddx >= (1 << 16))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    const size_t lds_ws = 4 * 128 * 32 * 4;
    static bool conf = false;
    if (!conf) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_ws_kernel<false, false, 4, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 32 * 4 + 8 * C1_KMAX) != hipSuccess)
            return GNX_ERR_LAUNCH;
        conf = true;
    }
    const int tilesN = (N + 127) / 128;
    const long T = (M / 128) * tilesN;
    const int wgs = (int)(T < 512 ? T : 512);
    float* slab = workspace;
    float* part = workspace + 2 * (M / 128) * 2 * (long)N;
    C1BnBwd bn = {X, (int)ldx, scale, shift, mean, invstd, slab};
    conv1x1_ws_kernel<false, false, 4, true><<<wgs, 512, lds_ws, stream>>>(dY, (int)lddy, Wt, dX, (int)lddx, K, N, tilesN,
                                                                           (int)T, 0, 4 * M, nullptr, nullptr, nullptr,
                                                                           nullptr, bn, c1_tile_runs());
    if (dgamma || dbeta) {
        slab_fold_kernel<<<dim3(gnx_cdiv(N, 64), C1_DGBN_R), 256, 0, stream>>>(slab, 2 * (M / 128), N, C1_DGBN_R, part);
        slab_finish_kernel<<<gnx_cdiv(N, 64), 512, 0, stream>>>(part, C1_DGBN_R, N, dbeta, dgamma, accumulate);
    }
    return gnx_launch_status();
}
