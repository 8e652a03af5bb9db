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
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int LDK = 36;     // floats per LDS row: 32 + 4 pad

__device__ __forceinline__ float act1(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed form: (x,y) and (z,w) stay in their even-aligned register pairs (v_pk_fma_f32 / v_pk_max_f32), so the
// compiler has no reason to shuffle freshly loaded registers (which would force a wait right behind the loads)
__device__ __forceinline__ float4 act4(float4 v, float4 sc, float4 sh) {
    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const f32x2 slo = {sc.x, sc.y}, shi = {sc.z, sc.w}, tlo = {sh.x, sh.y}, thi = {sh.z, sh.w}, zero = {0.f, 0.f};
    lo = __builtin_elementwise_max(__builtin_elementwise_fma(lo, slo, tlo), zero);
    hi = __builtin_elementwise_max(__builtin_elementwise_fma(hi, shi, thi), zero);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence whose wait is vmcnt(0):
// it would drain the global prefetch loads that are meant to stay in flight across the barrier (measured: the
// ping-pong memory phase took 12.5k cycles instead of ~2k with __syncthreads()).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// bounds-checked / unaligned-safe 4-float load of src[0..3], zero beyond `valid` elements
__device__ __forceinline__ float4 ld4_safe(const float* p, int valid, bool vec) {
    if (vec && valid >= 4) return ld4(p);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) r.x = p[0];
    if (valid > 1) r.y = p[1];
    if (valid > 2) r.z = p[2];
    if (valid > 3) r.w = p[3];
    return r;
}

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
                                                      const float* __restrict__ oshift) {
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
                rb[p] = ld4(W + (long)(n < N ? n : N - 1) * K + kc);
            } else {
                rb[p] = (n < N && valid > 0) ? ld4_safe(W + (long)n * K + k, valid, vecW)
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

// ------------------------------------------------------------------------------------------------ conv3x3 (pad 1)
// out[P][n] = sum_tap sum_k pad0(act(A))[nbr(P,tap)][k] * Wr[tap][n][k]
// A tile of 128 consecutive flattened positions (image, y, x) needs the flattened range [P0-S-1, P0+127+S+1]:
// it is staged ONCE per K-chunk as a contiguous strip, tap (dy,dx) of row i is strip row i + (S+1) + dy*S + dx,
// and the per-lane 9-bit validity mask (image border / row wrap) zeroes a fragment after the read.
constexpr int C3_BM = 128, C3_BN = 32;

__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ A, long lda,
                                                      const float* __restrict__ Wr, float* __restrict__ out, long ldc,
                                                      long M, int N, int K, int S, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int vecA, int vecW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int strip = C3_BM + 2 * S + 2;
    float* As = lds;                    // [strip][LDK]
    float* Bs = lds + strip * LDK;      // [9][32][LDK]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const long P0 = (long)blockIdx.x * C3_BM;
    const long base = P0 - S - 1;
    const int n0 = blockIdx.y * C3_BN;
    const bool has_act = scale != nullptr;

    // validity of the 9 taps for this lane's output position
    const long P = P0 + 32 * wave + i;
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
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int kq = t & 7;
    for (int k0 = 0; k0 < K; k0 += 32) {
        const int k = k0 + 4 * kq;
        const int valid = K - k;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act && valid > 0) {
            sc4 = ld4_safe(scale + k, valid, vecA);
            sh4 = ld4_safe(shift + k, valid, vecA);
        }
        __syncthreads();
        for (int row = t >> 3; row < strip; row += 32) {
            const long Pr = base + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Pr >= 0 && Pr < M && valid > 0) {
                v = ld4_safe(A + Pr * lda + k, valid, vecA);
                if (has_act) v = act4(v, sc4, sh4);
                if (valid < 4) {
                    if (valid < 2) v.y = 0.f;
                    if (valid < 3) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            *reinterpret_cast<float4*>(&As[row * LDK + 4 * kq]) = v;
        }
        for (int idx = t >> 3; idx < 9 * 32; idx += 32) {
            const int tap = idx >> 5, n = n0 + (idx & 31);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < N && valid > 0) v = ld4_safe(Wr + ((long)tap * N + n) * K + k, valid, vecW);
            *reinterpret_cast<float4*>(&Bs[idx * LDK + 4 * kq]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = (S + 1) + (tap / 3 - 1) * S + (tap % 3 - 1);
            const bool ok = (mask >> tap) & 1u;
            const float* ap = &As[(32 * wave + i + off) * LDK + 4 * h];
            const float* bp = &Bs[(tap * 32 + i) * LDK + 4 * h];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float4 a = ld4(ap + 8 * s);
                const float4 b = ld4(bp + 8 * s);
                if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    }
    const int col = n0 + i;
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = P0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) out[row * ldc + col] = acc[r];
        }
    }
}

// Software-pipelined form: the next K-chunk's strip rows and weight rows are fetched into registers while the
// current chunk is multiplied (issue-early / write-late), so HBM/L2 latency hides under the 144 MFMAs of a chunk.
// NJ = ceil(strip/32) strip rows per thread (compile-time so the prefetch array stays in registers).
template <int NJ>
__global__ __launch_bounds__(256) void conv3x3_pipe_kernel(const float* __restrict__ A, long lda,
                                                           const float* __restrict__ Wr, float* __restrict__ out,
                                                           long ldc, long M, int N, int K, int S,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int vecA, int vecW) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int strip = C3_BM + 2 * S + 2;
    float* As = lds;
    float* Bs = lds + strip * LDK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const long P0 = (long)blockIdx.x * C3_BM;
    const long base = P0 - S - 1;
    const int n0 = blockIdx.y * C3_BN;
    const bool has_act = scale != nullptr;
    const int kq = t & 7, r0 = t >> 3;

    const long P = P0 + 32 * wave + i;
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
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // a tap that falls outside the image reads a row of zeros kept behind the weight tile (no select on the data path)
    float* Zs = Bs + 9 * 32 * LDK;
    if (t < LDK) Zs[t] = 0.f;
    int aoff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (S + 1) + (tap / 3 - 1) * S + (tap % 3 - 1);
        aoff[tap] = ((mask >> tap) & 1u) ? (32 * wave + i + off) * LDK + 4 * h : (strip + 9 * 32) * LDK;
    }

    // Branch-free prefetch (this kernel is only launched when every pointer is 16-B aligned and K % 4 == 0, so a
    // k-quad is entirely valid or entirely past K): out-of-range rows/quads load from a clamped, always-valid address
    // and are zeroed when written to LDS.  No branch, no wait between the loads and the MFMA loop that hides them.
    float4 ra[NJ], rb[9], sc4, sh4;
    const int nload = n0 + r0 < N ? n0 + r0 : N - 1;
    auto fetch = [&](int k0) {
        const int k = k0 + 4 * kq;
        const int kc = k < K ? k : 0;
        sc4 = make_float4(1.f, 1.f, 1.f, 1.f);
        sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_act) {
            sc4 = ld4(scale + kc);
            sh4 = ld4(shift + kc);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            long Pr = base + r0 + 32 * j;
            Pr = Pr < 0 ? 0 : (Pr >= M ? M - 1 : Pr);
            ra[j] = ld4(A + Pr * lda + kc);
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) rb[j] = ld4(Wr + ((long)j * N + nload) * K + kc);
    };
    // workgroup-uniform: every strip row is a real position, K is a whole number of chunks, all 32 columns exist ->
    // nothing to zero, the LDS write is activation + store only (the stash is the non-MFMA work of the kernel:
    // PMC showed ~2.3 VALU instructions per MFMA before this fast path)
    const bool interior = base >= 0 && base + strip <= M && (K & 31) == 0 && n0 + C3_BN <= N;
    auto stash = [&](int k0) {
        if (interior) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int row = r0 + 32 * j;
                if (row >= strip) continue;
                float4 v = ra[j];
                if (has_act) v = act4(v, sc4, sh4);
                *reinterpret_cast<float4*>(&As[row * LDK + 4 * kq]) = v;
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * j) * LDK + 4 * kq]) = rb[j];
            return;
        }
        const bool kok = k0 + 4 * kq < K;
        const bool nok = n0 + r0 < N;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = r0 + 32 * j;
            if (row >= strip) continue;
            const long Pr = base + row;
            float4 v = ra[j];
            if (has_act) v = act4(v, sc4, sh4);
            if (!(kok && Pr >= 0 && Pr < M)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&As[row * LDK + 4 * kq]) = v;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float4 v = rb[j];
            if (!(kok && nok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Bs[(r0 + 32 * j) * LDK + 4 * kq]) = v;
        }
    };

    const float* bbase = &Bs[i * LDK + 4 * h];
    fetch(0);
    __syncthreads();            // zero row written
    stash(0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += 32) {
        // loads/activation/LDS writes run at raised priority: a co-resident workgroup's back-to-back MFMAs on the same
        // SIMD otherwise starve them of issue slots (measured ~90 cycles per instruction in the ping-pong variant)
        __builtin_amdgcn_s_setprio(3);
        if (k0 + 32 < K) fetch(k0 + 32);        // issued here, consumed at the end of THIS iteration (see conv1x1)
        __builtin_amdgcn_s_setprio(0);
        float4 a = ld4(lds + aoff[0]), b = ld4(bbase);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int tap = step >> 2, sidx = step & 3;
            float4 na = a, nb = b;
            if (step < 35) {
                const int ntap = (step + 1) >> 2, ns = (step + 1) & 3;
                na = ld4(lds + aoff[ntap] + 8 * ns);
                nb = ld4(bbase + ntap * 32 * LDK + 8 * ns);
            }
            (void)tap; (void)sidx;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc1, 0, 0, 0);
            a = na; b = nb;
            // pin the order: the two LDS reads of step+1 go out ahead of the four MFMAs of this step
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (k0 + 32 < K) {
            __builtin_amdgcn_s_setprio(3);
            __syncthreads();
            stash(k0 + 32);
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
        }
    }
    __builtin_amdgcn_s_setprio(3);
    const int col = n0 + i;
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = P0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) out[row * ldc + col] = acc0[r] + acc1[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv3x3, LDS-DMA form
// For inputs that need NO prologue (scale == nullptr: the eval forward stores the bottleneck already activated, see
// gnx_conv1x1_bnrelu's output activation).  The strip rows and the weight rows of a 32-wide K chunk go global -> LDS
// directly (global_load_lds_dwordx4: no VGPRs, no ds_write), double-buffered, one persistent workgroup per CU: the DMA
// of chunk g+1 is in flight while chunk g's 144 MFMAs run; one raw s_barrier per chunk orders everything
// (vmcnt for the wave's own DMAs -> barrier -> multiply, issuing the next chunk's DMAs between the MFMAs).
//
// What the measurements behind this shape said (tools/ubench/mfma_loop_variants.hip, one wave per SIMD):
//  * every VALU instruction a wave issues inside the MFMA loop costs matrix-pipe time (2 v_add per step: -7 %;
//    4 v_cndmask on the operands per step: -25 %), so the steady-state step is 2 ds_read + 1 s_waitcnt + 4 MFMA and
//    nothing else: fragment addresses are per-lane bases + instruction immediates, border masking is folded into the
//    bases once per tile (a masked tap's base points at a zero region), the DMA addresses are a scalar base (SALU) +
//    a per-lane constant offset;
//  * hipcc puts s_waitcnt vmcnt(0) in front of every LDS read that may alias a pending LDS-DMA write, which would
//    serialise the DMA of chunk g+1 with the multiply of chunk g: the fragment reads are inline asm with hand-counted
//    lgkmcnt (2 reads stay in flight).
//
// LDS image of a chunk: rows in groups of 16 ("double pieces", 2 KB), chunk-major inside a group:
//     byte(row r, 16-B chunk c) = (r >> 4) * 2048 + c * 256 + (r & 15) * 16
// A quarter-wave of a ds_read_b128 (16 consecutive rows, one c) then covers all 64 banks exactly once, and the four
// k-subchunks of a lane are base + {0, 512, 1024, 1536}.  One DMA instruction (64 lanes x 16 B, lane-linear in LDS)
// writes half a group: 16 rows x 4 chunks; lane L fetches row (L & 15), chunk 4 * half + (L >> 4).
#ifndef GNX_DMA_DBG
#define GNX_DMA_DBG 0
#endif

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Requires (checked by the dispatcher): M % (32 NW) == 0, M * max(lda, ldc) < 2^31, N == 32, K % (2 KC) == 0, 16-B
// aligned pointers and leading dimensions.  NW waves per workgroup, each owning 32 output rows of the 32*NW-row tile;
// KC = K elements per chunk (32: 128-B LDS rows, 1 workgroup per CU; 16: 64-B rows, half the LDS, 2 workgroups per CU
// whose barriers, prologues and stores then hide behind each other's MFMAs).  With KC = 16 a group of 16 rows is 1 KB
// (chunk c at c * 256, four chunks) and one DMA instruction writes a whole group.
// NK1 (data-gradient shape: K == KC = 32 input channels, N = 32 * NT output channels, NT even): a tile's chunks are its
// NT column tiles instead of K chunks - the strip is re-fetched (L2) with each column tile's weights, every chunk starts
// from zero accumulators and ends with its 16 stores.
template <int S, int NW, int KC, bool NK1 = false>
__global__ __launch_bounds__(64 * NW) void conv3x3_dma_kernel(const float* __restrict__ A, int lda,
                                                              const float* __restrict__ Wr, float* __restrict__ out,
                                                              int ldc, int M, int K, int N) {
    constexpr int BM = 32 * NW;
    constexpr int strip = BM + 2 * S + 2;
    constexpr int SR = (strip + 15) & ~15;                 // strip rows padded to whole groups of 16
    constexpr int ROWB = KC * 4;                           // bytes per LDS row
    constexpr int GB = 16 * ROWB, PPG = GB / 1024;         // group bytes; DMA pieces (1 KB) per group
    constexpr int BUFB = (SR + 9 * 32) * ROWB;             // bytes per buffer: strip groups, then 2 groups per tap
    constexpr int NPA = (SR >> 4) * PPG, NPW = 18 * PPG;   // DMA pieces per chunk
    constexpr int NSA = (NPA + NW - 1) / NW, NSW = (NPW + NW - 1) / NW;     // DMA slots per wave and chunk
    constexpr int SPT = KC / 8, STEPS = 9 * SPT;           // MFMA steps (8 k each) per tap and per chunk
    constexpr int ZB = 2 * BUFB;                           // zero region (2 KB) behind the two buffers
    static_assert(NSA + NSW <= STEPS, "one DMA slot per MFMA step");
    static_assert(ZB + 2048 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) char lds[ZB + 2048];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int z = t; z < 512; z += 64 * NW) reinterpret_cast<float*>(lds + ZB)[z] = 0.f;
    const int T = M / BM;
    const int nk2 = NK1 ? N / 64 : K / (2 * KC);           // pairs of chunks (NK1: of column tiles)
    int G = gridDim.x;                                     // pinned in an SGPR: no s_load may sit among the counted
    asm volatile("" : "+s"(G));                            // lgkmcnt waits of the fragment reads
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD has its own L2.  In every FULL
    // round of G tiles each XCD takes a contiguous eighth, so the halo rows neighbouring tiles share are fetched once per
    // L2 instead of once per XCD; the last, partial round keeps the plain order (it would otherwise leave whole XCDs
    // idle: measured -4.5 % at S = 16, 19.5 rounds).
#ifndef GNX_XCD_ORDER
#define GNX_XCD_ORDER 1
#endif
    const int bx = blockIdx.x;
#if GNX_XCD_ORDER == 2
    // groups of 4 consecutive tiles per XCD, groups interleaved over the XCDs
    const int bid = (G & 31) == 0 ? ((bx >> 5) << 5) + ((bx & 7) << 2) + ((bx >> 3) & 3) : bx;
#else
    const int bid = (GNX_XCD_ORDER && (G & 7) == 0) ? (bx & 7) * (G >> 3) + (bx >> 3) : bx;
#endif
    auto tile_of = [&](int round) {                        // >= T: this workgroup has no tile in that round
        const int base = round * G;
        return base + (base + G <= T ? bid : bx);
    };
    const unsigned lb = lds_addr(lds);

    // ---- DMA side.  Per-lane constant byte offsets; everything else is scalar.
    const unsigned voffA = ((unsigned)(lane & 15) * lda + 4 * (lane >> 4)) * 4u;
    const unsigned voffW = ((unsigned)(lane & 15) * K + 4 * (lane >> 4)) * 4u;
    // next chunk's DMA state.  Past this workgroup's last tile the DMA re-reads its first tile into the buffer nobody
    // will read: cheaper than a branch around every slot.  gridDim.x <= T: every workgroup owns at least one tile.
    int nround = 0, ntile = tile_of(0), nchunk = 0;
    auto issue_slot = [&](auto slot_c, char* dst) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (slot < NSA) {
            const int p = wave + NW * slot;                                    // piece: group p / PPG, part p % PPG
            if ((NPA % NW) && slot == NSA - 1 && p >= NPA) return;
            const int grp = p / PPG, part = p % PPG;
            const int row0 = ntile * BM - S - 1 + 16 * grp;                    // first strip row of the group
            char* d = dst + grp * GB + part * 1024;
            if (__builtin_expect(row0 >= 0 && row0 + 15 < M, 1)) {
                const char* sb = reinterpret_cast<const char*>(A + (long)row0 * lda + (NK1 ? 0 : KC * nchunk) + 16 * part);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(sb + voffA), (float*)d, 16, 0, 0);
            } else {
                // array ends: rows outside [0, M) are only ever "read" by masked taps, any in-range row will do
                int Pr = row0 + (lane & 15);
                Pr = Pr < 0 ? 0 : (Pr >= M ? M - 1 : Pr);
                __builtin_amdgcn_global_load_lds(A + (long)Pr * lda + (NK1 ? 0 : KC * nchunk) + 16 * part + 4 * (lane >> 4),
                                                 (float*)d, 16, 0, 0);
            }
        } else {
            const int p = wave + NW * (slot - NSA);                            // weight group p / PPG = 2 tap + (n >> 4)
            if ((NPW % NW) && slot == NSA + NSW - 1 && p >= NPW) return;
            const int grp = p / PPG, part = p % PPG;
            // weight rows [tap][N][K]: tap = grp >> 1, n = 32 * (column tile) + 16 * (grp & 1) + lane row
            const char* sb = reinterpret_cast<const char*>(
                Wr + (long)((grp >> 1) * N + (NK1 ? 32 * nchunk : 0) + (grp & 1) * 16) * K + (NK1 ? 0 : KC * nchunk) + 16 * part);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(sb + voffW),
                                             (float*)(dst + SR * ROWB + grp * GB + part * 1024), 16, 0, 0);
        }
    };
    auto advance_next = [&]() {
        if (++nchunk == 2 * nk2) { nchunk = 0; ntile = tile_of(++nround); }
        if (ntile >= T) ntile = bid;
    };
    static_for<0, NSA + NSW>([&](auto sc) { issue_slot(sc, lds); });
    advance_next();

    // ---- fragment side.  Per-lane constant bases relative to a buffer.
    const int R0 = 32 * wave + i + (S + 1);                // strip row of this lane's output pixel
    unsigned relA[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int R = R0 + (tap / 3 - 1) * S + (tap % 3 - 1);
        relA[tap] = (R >> 4) * GB + (R & 15) * 16 + h * 256;
    }
    const unsigned relB = SR * ROWB + (i >> 4) * GB + (i & 15) * 16 + h * 256;
    const unsigned bB0 = lb + relB, bB1 = lb + BUFB + relB;

    bool stored = false;
    for (int round = 0, tile = tile_of(0); tile < T; tile = tile_of(++round)) {
        const int P = tile * BM + 32 * wave + i;
        const int rem = P % (S * S);
        const int y = rem / S, x = rem - y * S;
        unsigned bA0[9], bA1[9];                           // masked taps point at the zero region
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < S && xx >= 0 && xx < S;
            bA0[tap] = ok ? lb + relA[tap] : lb + ZB;
            bA1[tap] = ok ? lb + BUFB + relA[tap] : lb + ZB;
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

        auto store16 = [&](int col0) {
            float* o = out + (long)(tile * BM + 32 * wave + 4 * h) * ldc + col0 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(long)(((r & 3) + 8 * (r >> 2)) * ldc)] = acc0[r] + acc1[r];     // exactly 16 stores
            stored = true;
        };
        auto do_chunk = [&](auto par_c, int ct) {
            constexpr int par = decltype(par_c)::value;
            if constexpr (NK1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            }
            // this wave's DMAs into the current buffer have landed (the 16 stores of a just-finished tile may stay in
            // flight: they are younger than those DMAs and vmcnt retires in order)
            if (stored) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stored = false;
#if GNX_DMA_DBG != 2
            asm volatile("s_barrier" ::: "memory");     // everyone's data visible; everyone done with the other buffer
#endif
            char* nxt = lds + (par ? 0 : BUFB);
            const unsigned bB = par ? bB1 : bB0;
            auto rdA = [&](auto e_c) {
                constexpr int e = decltype(e_c)::value;
                return lds_read4<(e % SPT) * 512>(par ? bA1[e / SPT] : bA0[e / SPT]);
            };
            auto rdB = [&](auto e_c) {
                constexpr int e = decltype(e_c)::value;
                return lds_read4<(e / SPT) * 2 * GB + (e % SPT) * 512>(bB);
            };
            f32x4 a = rdA(std::integral_constant<int, 0>{}), bq = rdB(std::integral_constant<int, 0>{});
            static_for<0, STEPS>([&](auto step_c) {
                constexpr int step = decltype(step_c)::value;
                f32x4 na, nb;
                if constexpr (step < STEPS - 1) {
                    na = rdA(std::integral_constant<int, step + 1>{});
                    nb = rdB(std::integral_constant<int, step + 1>{});
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(bq));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(bq));
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bq[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bq[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bq[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bq[3], acc1, 0, 0, 0);
#if GNX_DMA_DBG != 1
                if constexpr (step < NSA + NSW) issue_slot(step_c, nxt);
#endif
                if constexpr (step < STEPS - 1) { a = na; bq = nb; }
            });
            advance_next();
            if constexpr (NK1) store16(32 * ct);
        };
        for (int c2 = 0; c2 < nk2; ++c2) {
            do_chunk(std::integral_constant<int, 0>{}, 2 * c2);
            do_chunk(std::integral_constant<int, 1>{}, 2 * c2 + 1);
        }
        if constexpr (!NK1) store16(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <bool ACT, bool POOL, int NP>
__global__ __launch_bounds__(64 * (4 + NP), NP == 4 ? 4 : 3) void conv1x1_ws_kernel(const float* __restrict__ A, int lda,
                                                            const float* __restrict__ W, float* __restrict__ out,
                                                            int ldc, int K, int N, int tilesN, int T, int S_in,
                                                            long rows_in, const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ oscale,
                                                            const float* __restrict__ oshift) {
    constexpr int OPB = 128 * 32 * 4;                      // bytes of one operand chunk
    extern __shared__ __attribute__((aligned(16))) float lds_f[];  // [buffer][A | B] then scale[K], shift[K]
    char* const lds = reinterpret_cast<char*>(lds_f);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = K >> 5;
    const int G = gridDim.x;
    const int mine = (T - (int)blockIdx.x + G - 1) / G;    // tiles of this workgroup (gridDim.x <= T)
    const int total = mine * nk;                           // chunks = barriers, the same number for all 8 waves
    // Workgroup -> tile order.  With several column tiles per row tile (transitions: N = 256 / 512) the column tiles of one
    // row tile read the same (4x pooled) activation rows: put them on the SAME XCD (workgroup b runs on XCD b % 8, one L2
    // per XCD) so those rows come from HBM once - measured 6.6 GB per transition launch before, 1.9x the algorithmic bytes.
    // Full rounds of G tiles only; a partial last round keeps the plain order (every workgroup with blockIdx.x < rest works).
    const int bx = blockIdx.x;
    const int jmap = (tilesN > 1 && G % (8 * tilesN) == 0)
                         ? tilesN * ((bx & 7) + 8 * (bx / (8 * tilesN))) + (bx >> 3) % tilesN : bx;
    auto tile_of = [&](int round) {
        const int base = round * G;
        return base + (base + G <= T ? jmap : bx);
    };
    // Every global load a wave issues beside the MFMA waves costs matrix-pipe time in proportion to its bytes, LDS reads
    // do not (tools/ubench/mfma_2x2.hip): the per-chunk scale/shift vectors come from an LDS copy made once per
    // workgroup instead of four more 1-KB loads per producer wave and chunk.
    float* sS = reinterpret_cast<float*>(lds + 4 * OPB);
    if (ACT) {
        for (int k = t; k < K; k += 64 * (4 + NP)) { sS[k] = scale[k]; sS[K + k] = shift[k]; }
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
            auto loadp = [&]() {
                const int tm = ptile / tilesN, tn = ptile - tm * tilesN;
                const long base = srcrow(tm * 128);
                long left = (rows_in - base) * lda - (lda - K);                 // floats behind the resource base
                const long span = (long)(4 * 128 + 2 * S_in + 4) * lda;
                if (left > span) left = span;
                const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(A + base * lda), 0, (int)(left * 4), 0x00020000);
                const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(W + (long)tn * 128 * K), 0, (N - tn * 128 < 128 ? N - tn * 128 : 128) * K * 4,
                    0x00020000);          // rows past N read as 0 (buffer bounds check): ragged last column tile
                const int vo0 = (int)(srcrow(tm * 128 + 32 * pw + lr) - base) * lda * 4 + lc * 16;
                const int vo1 = (int)(srcrow(tm * 128 + 32 * pw + lr + 16) - base) * lda * 4 + lc * 16;
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
                if (++pkt == nk) { pkt = 0; ptile = tile_of(++pround); }
                if (ptile >= T) ptile = bx;
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
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float4 v = pa[4 * q + u];
                        if (ACT) v = act4(v, sc[q & 1], sh[q & 1]);
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    *reinterpret_cast<float4*>(d + (q >> 1) * 2048 + (q & 1) * 1024) =
                        make_float4(0.25f * sum.x, 0.25f * sum.y, 0.25f * sum.z, 0.25f * sum.w);
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
        auto store = [&](f32x16& acc, int mt, int nt) {
            const int col = tn * 128 + 64 * wn + 32 * nt + i;
            if (col - i >= N) return;                      // 32 | N: whole fragments in or out
            if (oscale) {
                const float osc = oscale[col], osh = oshift[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = fmaxf(fmaf(acc[r], osc, osh), 0.f);
            }
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

// ------------------------------------------------------------------------------------------------ conv3x3, Winograd F(2,3) along x
// The 3x3 convolution of a pre-activated operand with 1.5x fewer multiplies: along a row, two neighbouring outputs
//   y0 = d0 g0 + d1 g1 + d2 g2,  y1 = d1 g0 + d2 g1 + d3 g2      (d = 4 consecutive inputs, g = the 3 taps of one kernel row)
// are  y0 = m0 + m1 + m2,  y1 = m1 - m2 - m3  with
//   m0 = (d0 - d2) g0,  m1 = (d1 + d2) (g0 + g1 + g2)/2,  m2 = (d2 - d1) (g0 - g1 + g2)/2,  m3 = (d1 - d3) g2:
// 4 multiplies instead of 6, summed over the 3 kernel rows and all channels BEFORE the output transform, so a wave keeps 4
// accumulators M0..M3 for its 32 output PAIRS x 32 channels and runs 12 "taps" (3 rows x 4) instead of 18 per pair.
// Per K chunk (16 channels) the workgroup first turns the DMA'd strip into the four transformed arrays
// V0..V3[pair][k] in LDS (VALU phase: nothing else of this workgroup runs on the matrix pipe meanwhile, so the VALU
// stream is not starved), then multiplies out of V and the transformed weights U[row][xi][n][k] (gnx_winograd_conv3x3
// _weights; DMA'd).  S must be a power of two (pairs never straddle an image row; 256-pixel tiles start at x = 0 mod S);
// the zero padding left and right of a row is applied in the transform, above and below by pointing the (row, xi) fragment
// bases of a masked row at a zero region, as in the direct kernel.  Differences from the direct sum are rounding only
// (coefficients 1 and 1/2).
#ifndef GNX_WINO_DBG
#define GNX_WINO_DBG 0        // ablations for tools/kbench.py: 1 = no input transform, 2 = no DMA after the first chunk
#endif
template <int S>
__global__ __launch_bounds__(256) void conv3x3_wino_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ Wu, float* __restrict__ out,
                                                          int ldc, int M, int K) {
    constexpr int BM = 256, NPAIR = 128;                  // pixels / output pairs per tile (4 waves x 32 pairs)
    constexpr int RAWN = BM + 2 * S + 2, SRAW = (RAWN + 15) & ~15;        // raw strip pixels [P0 - S - 1, P0 + BM + S + 1)
    constexpr int NV = NPAIR + S, NVR = (NV + 15) & ~15;                   // V pairs [P0 - S, P0 + BM + S) / 2
    constexpr int RAWB = SRAW * 64, UB = 12 * 32 * 64, VB = NVR * 64;      // bytes: raw chunk, U chunk, one V_xi array
    constexpr int OFF_RAW = 0, OFF_U = 2 * RAWB, OFF_V = OFF_U + 2 * UB, OFF_Z = OFF_V + 4 * VB;
    static_assert(OFF_Z + 1024 <= 160 * 1024, "LDS");
    constexpr int NPR = SRAW / 16, NPU = 24;               // DMA pieces (16 rows x 64 B) per chunk
    constexpr int NSR = (NPR + 3) / 4, NSU = NPU / 4;      // slots per wave
    static_assert(NSR + NSU <= 24, "one DMA slot per MFMA step");
    __shared__ __attribute__((aligned(16))) char lds[OFF_Z + 1024];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int z = t; z < 256; z += 256) reinterpret_cast<float*>(lds + OFF_Z)[z] = 0.f;
    const int T = (M + BM - 1) / BM;                       // the last tile may be ragged (stores guarded, loads clamped)
    const int nk2 = K >> 5;                                // pairs of 16-channel chunks
    int G = gridDim.x;
    asm volatile("" : "+s"(G));
    const int bx = blockIdx.x;
    const int bid = (G & 7) == 0 ? (bx & 7) * (G >> 3) + (bx >> 3) : bx;       // XCD-aware order, as the direct kernel
    auto tile_of = [&](int round) {
        const int base = round * G;
        return base + (base + G <= T ? bid : bx);
    };
    const unsigned lb = lds_addr(lds);

    // ---- DMA side (16 rows x 4 chunks per piece: lane -> row lane & 15, 16-B chunk lane >> 4)
    // Raw-strip pieces with an odd index are stored rotated by one 16-B slot: the transform reads every OTHER pixel
    // (q = 2 jp + e), 16 lanes = 32 pixels = two pieces x the same 8 slots - a 2-way bank conflict on all its reads
    // unless the second piece's slots are shifted.  A wave's pieces all have the parity of its wave index.
    const int pix = ((lane & 15) - (wave & 1)) & 15;       // pixel of the piece this lane fetches into slot lane & 15
    const unsigned voffA = ((unsigned)pix * lda + 4 * (lane >> 4)) * 4u;
    const unsigned voffU = ((unsigned)(lane & 15) * K + 4 * (lane >> 4)) * 4u;
    int nround = 0, ntile = tile_of(0), nchunk = 0;
    auto issue_slot = [&](auto slot_c, int buf) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (slot < NSR) {
            const int p = wave + 4 * slot;
            if ((NPR % 4) && slot == NSR - 1 && p >= NPR) return;
            const int row0 = ntile * BM - S - 1 + 16 * p;
            char* d = lds + OFF_RAW + buf * RAWB + p * 1024;
            if (__builtin_expect(row0 >= 0 && row0 + 15 < M, 1)) {
                const char* sb = reinterpret_cast<const char*>(A + (long)row0 * lda + 16 * nchunk);
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(sb + voffA), (float*)d, 16, 0, 0);
            } else {                                       // array ends: any in-range row (only masked rows use it)
                int Pr = row0 + pix;
                Pr = Pr < 0 ? 0 : (Pr >= M ? M - 1 : Pr);
                __builtin_amdgcn_global_load_lds(A + (long)Pr * lda + 16 * nchunk + 4 * (lane >> 4), (float*)d, 16, 0, 0);
            }
        } else {
            const int p = wave + 4 * (slot - NSR);         // U group p: tap p >> 1, rows 16 (p & 1) ..
            const char* sb = reinterpret_cast<const char*>(Wu + (long)(16 * p) * K + 16 * nchunk);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(sb + voffU),
                                             (float*)(lds + OFF_U + buf * UB + p * 1024), 16, 0, 0);
        }
    };
    auto advance_next = [&]() {
        if (++nchunk == 2 * nk2) { nchunk = 0; ntile = tile_of(++nround); }
        if (ntile >= T) ntile = bid;
    };
    static_for<0, NSR + NSU>([&](auto sc) { issue_slot(sc, 0); });
    advance_next();

    // ---- transform side: item = (V pair jp, 16-B chunk c), jp fastest across lanes
    constexpr int NIT = (NVR * 4 + 255) / 256;
    // ---- fragment side
    const int r = 32 * wave + i;                           // output pair of this lane within the tile
    const unsigned fU = lb + OFF_U + (i >> 4) * 1024 + (i & 15) * 16 + h * 256;

    bool stored = false;
    for (int round = 0, tile = tile_of(0); tile < T; tile = tile_of(++round)) {
        const int P = tile * BM + 2 * r;
        const int y = (P % (S * S)) / S;
        unsigned bV[12];                                   // (row dy, xi) fragment bases; masked rows -> zero region
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int jp = r + S / 2 + (dy - 1) * (S / 2);
            const bool ok = (dy == 1) || (dy == 0 ? y > 0 : y < S - 1);
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
                bV[dy * 4 + xi] = ok ? lb + OFF_V + xi * VB + (jp >> 4) * 1024 + (jp & 15) * 16 + h * 256 : lb + OFF_Z;
        }
        f32x16 acc[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[xi][q][e] = 0.f;

        auto do_chunk = [&](auto par_c) {
            constexpr int par = decltype(par_c)::value;
            if (stored) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");      // the 32 stores of a finished tile may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stored = false;
            asm volatile("s_barrier" ::: "memory");       // chunk's raw strip + U visible; V and the other buffers free
            // input transform: V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3 (d0 / d3 zero beside the row ends)
#if GNX_WINO_DBG != 1
            {
                const char* raw = lds + OFF_RAW + par * RAWB;
#pragma unroll
                for (int j = 0; j < NIT; ++j) {
                    const int it = t + 256 * j;
                    const int c = it / NVR, jp = it - c * NVR;
                    if (it < NVR * 4 && jp < NV) {             // pairs past NV are never read
                        const int xin = (2 * jp) & (S - 1);
                        // the zero padding beside the row ends: read the zero region instead of the neighbour (an address
                        // choice the compiler hoists out of every loop - no select on the data)
                        auto at = [&](int q) {
                            return raw + (q >> 4) * 1024 + c * 256 + (((q & 15) + ((q >> 4) & 1)) & 15) * 16;
                        };
                        const char* zr = lds + OFF_Z;
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        struct F4 { f2 lo, hi; };
                        auto rd = [&](const char* p) { return *reinterpret_cast<const F4*>(p); };
                        const F4 d0 = rd(xin == 0 ? zr : at(2 * jp)), d1 = rd(at(2 * jp + 1)), d2 = rd(at(2 * jp + 2)),
                                 d3 = rd(xin + 2 == S ? zr : at(2 * jp + 3));
                        char* v = lds + OFF_V + (jp >> 4) * 1024 + c * 256 + (jp & 15) * 16;
                        *reinterpret_cast<F4*>(v) = F4{d0.lo - d2.lo, d0.hi - d2.hi};
                        *reinterpret_cast<F4*>(v + VB) = F4{d1.lo + d2.lo, d1.hi + d2.hi};
                        *reinterpret_cast<F4*>(v + 2 * VB) = F4{d2.lo - d1.lo, d2.hi - d1.hi};
                        *reinterpret_cast<F4*>(v + 3 * VB) = F4{d1.lo - d3.lo, d1.hi - d3.hi};
                    }
                }
            }
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");       // V visible
            const unsigned bU = fU + par * UB;
            auto rdA = [&](auto e_c) {                     // step e = 2 * tap + s2
                constexpr int e = decltype(e_c)::value;
                return lds_read4<(e & 1) * 512>(bV[e >> 1]);
            };
            auto rdB = [&](auto e_c) {
                constexpr int e = decltype(e_c)::value;
                return lds_read4<(e >> 1) * 2048 + (e & 1) * 512>(bU);
            };
            f32x4 a = rdA(std::integral_constant<int, 0>{}), bq = rdB(std::integral_constant<int, 0>{});
            static_for<0, 24>([&](auto step_c) {
                constexpr int step = decltype(step_c)::value;
                constexpr int xi = (step >> 1) & 3;
                f32x4 na, nb;
                if constexpr (step < 23) {
                    na = rdA(std::integral_constant<int, step + 1>{});
                    nb = rdB(std::integral_constant<int, step + 1>{});
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(bq));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(bq));
                }
                acc[xi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bq[0], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bq[1], acc[xi][1], 0, 0, 0);
                acc[xi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bq[2], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bq[3], acc[xi][1], 0, 0, 0);
#if GNX_WINO_DBG != 2
                if constexpr (step < NSR + NSU) issue_slot(step_c, par ^ 1);
#endif
                if constexpr (step < 23) { a = na; bq = nb; }
            });
            advance_next();
        };
        for (int c2 = 0; c2 < nk2; ++c2) {
            do_chunk(std::integral_constant<int, 0>{});
            do_chunk(std::integral_constant<int, 1>{});
        }
        // output transform: y(2p) = M0 + M1 + M2, y(2p+1) = M1 - M2 - M3
        float* o = out + (long)(tile * BM + 2 * (32 * wave + 4 * h)) * ldc + i;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float m0 = acc[0][0][e] + acc[0][1][e], m1 = acc[1][0][e] + acc[1][1][e];
            const float m2 = acc[2][0][e] + acc[2][1][e], m3 = acc[3][0][e] + acc[3][1][e];
            const long row = 2 * ((e & 3) + 8 * (e >> 2));
            if (tile * BM + 2 * (32 * wave + 4 * h) + row < M) {       // M is even: both pixels of a pair are in or out
                o[row * ldc] = (m0 + m1) + m2;
                o[(row + 1) * ldc] = (m1 - m2) - m3;
            }
        }
        stored = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// g [N][K][3][3] (torch) -> U [row dy][xi][N][K]: U0 = g0, U1 = (g0 + g1 + g2)/2, U2 = (g0 - g1 + g2)/2, U3 = g2 (along kx)
__global__ void winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ wu, int N, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)3 * N * K;
    if (idx >= total) return;
    const int k = (int)(idx % K), n = (int)((idx / K) % N), dy = (int)(idx / ((long)K * N));
    const float* g = w + ((long)n * K + k) * 9 + dy * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    const long nk = (long)N * K;
    float* u = wu + (long)dy * 4 * nk + (long)n * K + k;
    u[0] = g0;
    u[nk] = 0.5f * ((g0 + g2) + g1);
    u[2 * nk] = 0.5f * ((g0 + g2) - g1);
    u[3 * nk] = g2;
}

// [N][K][3][3] (torch) -> [tap][N][K]
__global__ void repack3x3_kernel(const float* __restrict__ w, float* __restrict__ wr, int N, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)9 * N * K;
    if (idx >= total) return;
    const int k = (int)(idx % K), n = (int)((idx / K) % N), tap = (int)(idx / ((long)K * N));
    wr[idx] = w[((long)n * K + k) * 9 + tap];
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
constexpr int SP_LDT = 72;         // floats per position of the activated tile in LDS: 4 * 72 = 32 (mod 64) banks, so the
                                   // two lane halves of an accumulator store (positions p and p + 4) hit disjoint banks
__global__ __launch_bounds__(256) void conv_stem_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ out, long ldo, int H, int Wd, int O,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, long imgs) {
    constexpr int CIN = 3, KH = 7, KW = 7, STRIDE = 2, PAD = 3, WO = 64;
    constexpr int PH = STRIDE + KH, PW = (63 * STRIDE + 8 + 1 + 1) & ~1;       // 9 x 136 input patch per channel
    constexpr int KT = CIN * KH * 8, LDB = KT + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
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
    const int trow = 32 * wave + i;                       // this lane's position in the 2 x 64 tile
    const float* pa = Ps + (STRIDE * (trow >> 6)) * PW + STRIDE * (trow & 63) + 4 * h;
    const float* pb = Bs + i * LDB + 4 * h;
    const float sc0 = i < O ? scale[i] : 0.f, sh0 = i < O ? shift[i] : 0.f;
    const float sc1 = 32 + i < O ? scale[32 + i] : 0.f, sh1 = 32 + i < O ? shift[32 + i] : 0.f;
    const int c4 = t & 15, pxa = t >> 4;                  // pooling items of this thread: (pxa, c4) and (pxa + 16, c4)
    const int Ho2 = (H + 2 * PAD - KH) / STRIDE + 1;      // conv rows (== 64 for the 128-px stem), pooled rows Ho2 / 2
    const int ntt = Ho2 / 2;

    // Patch staging: 27 rows (3 channels x 9 input rows) of Wd = 128 floats = 864 16-B pieces, 4 per thread (the last
    // partly idle), fetched one tile AHEAD into registers while the current tile multiplies.
    float4 pre[4];
    auto fetch_patch = [&](long img, int tt) {
        const int iy0 = 2 * tt * STRIDE - PAD;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = t + 256 * q;
            const int f4 = j & 31, py = (j >> 5) % PH, c = (j >> 5) / PH;
            const int iy = iy0 + py;
            pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < CIN * PH * 32 && iy >= 0 && iy < H) pre[q] = ld4(x + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
        }
    };
    auto stash_patch = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = t + 256 * q;
            if (j < CIN * PH * 32) {
                float* d = Ps + (j >> 5) * PW + PAD + 4 * (j & 31);          // patch column = ix + PAD
                d[0] = pre[q].x; d[1] = pre[q].y; d[2] = pre[q].z; d[3] = pre[q].w;
            }
        }
    };
    __syncthreads();                                      // Bs and the zeroed patch are in place
    fetch_patch(blockIdx.x, 0);

    for (long img = blockIdx.x; img < imgs; img += gridDim.x) {
        float4 carry[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
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
            // norm0 + relu0 on the accumulators, tile to LDS (lane = channel, register = position)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                Ts[rr * SP_LDT + i] = fmaxf(fmaf(acc0[r], sc0, sh0), 0.f);
                Ts[rr * SP_LDT + 32 + i] = fmaxf(fmaf(acc1[r], sc1, sh1), 0.f);
            }
            __syncthreads();
            // pool0: pooled row tt, two (px, 4-channel) items per thread
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int px = pxa + 16 * q;
                float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;            // horizontal max of conv rows 2tt, 2tt+1
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int ox = 2 * px + dx;
                    if (ox < 0) continue;                                        // ox <= 63 always
                    const float4 v0 = ld4(&Ts[ox * SP_LDT + 4 * c4]);
                    const float4 v1 = ld4(&Ts[(WO + ox) * SP_LDT + 4 * c4]);
                    m0 = make_float4(fmaxf(m0.x, v0.x), fmaxf(m0.y, v0.y), fmaxf(m0.z, v0.z), fmaxf(m0.w, v0.w));
                    m1 = make_float4(fmaxf(m1.x, v1.x), fmaxf(m1.y, v1.y), fmaxf(m1.z, v1.z), fmaxf(m1.w, v1.w));
                }
                const float4 cv = carry[q];
                const float4 o4 = make_float4(fmaxf(fmaxf(cv.x, m0.x), m1.x), fmaxf(fmaxf(cv.y, m0.y), m1.y),
                                              fmaxf(fmaxf(cv.z, m0.z), m1.z), fmaxf(fmaxf(cv.w, m0.w), m1.w));
                carry[q] = m1;
                if (4 * c4 < O)
                    *reinterpret_cast<float4*>(out + ((img * (Ho2 / 2) + tt) * (long)(WO / 2) + px) * ldo + 4 * c4) = o4;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ BN+ReLU+maxpool 3x3 s2 p1
__global__ __launch_bounds__(256) void bnrelu_maxpool_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                             long ldo, long Mout, int C, int Hi, int Wi, int Ho, int Wo,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift) {
    const long total = Mout * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C;
        const int c = (int)(idx - row * C);
        const long img = row / ((long)Ho * Wo);
        const int rem = (int)(row - img * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float sc = scale[c], sh = shift[c];
        float m = 0.f;     // relu output is >= 0 and the window always holds a valid tap
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= Wi) continue;
                m = fmaxf(m, fmaf(in[((img * Hi + iy) * Wi + ix) * ldi + c], sc, sh));
            }
        }
        out[row * ldo + c] = m;
    }
}

// same, 4 channels per thread with 16-B accesses (C % 4 == 0, aligned pointers / leading dimensions)
__global__ __launch_bounds__(256) void bnrelu_maxpool_vec4_kernel(const float* __restrict__ in, long ldi,
                                                                  float* __restrict__ out, long ldo, long Mout, int C4,
                                                                  int Hi, int Wi, int Ho, int Wo,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift) {
    const long total = Mout * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx / C4;
        const int c = 4 * (int)(idx - row * C4);
        const long img = row / ((long)Ho * Wo);
        const int rem = (int)(row - img * Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float4 sc = ld4(scale + c), sh = ld4(shift + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= Wi) continue;
                const float4 v = ld4(in + ((img * Hi + iy) * Wi + ix) * ldi + c);
                m.x = fmaxf(m.x, fmaf(v.x, sc.x, sh.x));
                m.y = fmaxf(m.y, fmaf(v.y, sc.y, sh.y));
                m.z = fmaxf(m.z, fmaf(v.z, sc.z, sh.z));
                m.w = fmaxf(m.w, fmaf(v.w, sc.w, sh.w));
            }
        }
        *reinterpret_cast<float4*>(out + row * ldo + c) = m;
    }
}

// ------------------------------------------------------------------------------------------------ BN+ReLU+global average pool
// out[img][c] = mean over the S2 positions of relu(x*scale+shift)
__global__ __launch_bounds__(256) void bnrelu_avgpool_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out,
                                                             long ldo, int C, int S2, const float* __restrict__ scale,
                                                             const float* __restrict__ shift) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const long img = blockIdx.x;
    float acc = 0.f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c];
        for (int r = rl; r < S2; r += 4) acc += fmaxf(fmaf(in[(img * S2 + r) * ldi + c], sc, sh), 0.f);
    }
    red[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C)
        out[img * ldo + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)S2;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// out[M][N] (ldc) = act(A[M][K] (lda)) . W[N][K]^T ; pool != 0: A is on an S_in x S_in grid per image and M counts the
// (S_in/2)^2 pooled positions.  scale/shift may both be NULL (no activation).
static int conv1x1_launch(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                          const float* scale, const float* shift, int pool, int S_in, const float* oscale,
                          const float* oshift, hipStream_t stream) {
    if (!A || !W || !out || M < 0 || N <= 0 || K <= 0 || lda < K || ldc < N || (!scale) != (!shift) ||
        (!oscale) != (!oshift))
        return GNX_ERR_BAD_ARG;
    if (pool && (S_in < 2)) return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    const int vecA = al16(A) && lda % 4 == 0 && K % 4 == 0 && (!scale || (al16(scale) && al16(shift)));
    const int vecW = al16(W) && K % 4 == 0;
    dim3 grid(gnx_cdiv(M, C1_BM), gnx_cdiv(N, C1_BN));
    const bool fast = vecA && vecW;
    if (fast && M % 128 == 0 && N % 32 == 0 && K % 32 == 0 && K <= C1_KMAX && (!pool || (S_in % 2 == 0 && scale)) &&
        4 * M < (1L << 31) && lda < (1 << 16) && ldc < (1 << 16) && !getenv("GNX_NO_WS1")) {     // int row / lane offsets
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
        static const int np8 = getenv("GNX_WS_NP8") ? 1 : 0;
        const int np = (!pool && np8) ? 8 : 4;
        const int per_cu = np == 4 ? 2 : 1;
        const int wgs = (int)(T < 256 * per_cu ? T : 256 * per_cu);
#define GNX_WS(ACTV, POOLV, NPV)                                                                                     \
    conv1x1_ws_kernel<ACTV, POOLV, NPV><<<wgs, 64 * (4 + NPV), lds_ws, stream>>>(                                     \
        A, (int)lda, W, out, (int)ldc, K, N, tilesN, (int)T, S_in, 4 * M, scale, shift, oscale, oshift)
        if (pool) GNX_WS(true, true, 4);
        else if (scale) { if (np == 8) GNX_WS(true, false, 8); else GNX_WS(true, false, 4); }
        else { if (np == 8) GNX_WS(false, false, 8); else GNX_WS(false, false, 4); }
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

// As gnx_conv1x1_bnrelu (pool = 0) with the CONSUMER's folded BN + ReLU applied at the store:
// out[m][n] = relu(out_scale[n] * (act(A) . W^T)[m][n] + out_shift[n]).  The eval forward stores the bottleneck this way,
// so the 3x3 convolution that follows needs no prologue and can stream its operand global -> LDS by DMA.
GNX_EXPORT int gnx_conv1x1_bnrelu_act(const float* A, long lda, const float* W, float* out, long ldc, long M, int N,
                                      int K, const float* scale, const float* shift, const float* out_scale,
                                      const float* out_shift, hipStream_t stream) {
    if (!out_scale || !out_shift) return GNX_ERR_BAD_ARG;
    return conv1x1_launch(A, lda, W, out, ldc, M, N, K, scale, shift, 0, 0, out_scale, out_shift, stream);
}

GNX_EXPORT int gnx_repack_conv3x3(const float* w, float* wr, int N, int K, hipStream_t stream) {
    if (!w || !wr || N <= 0 || K <= 0) return GNX_ERR_BAD_ARG;
    repack3x3_kernel<<<gnx_cdiv(9L * N * K, 256), 256, 0, stream>>>(w, wr, N, K);
    return gnx_launch_status();
}

// w [N][K][3][3] -> wu [3][4][N][K] (12 N K floats): the Winograd F(2,3)-along-x transform of the weights
GNX_EXPORT int gnx_winograd_conv3x3_weights(const float* w, float* wu, int N, int K, hipStream_t stream) {
    if (!w || !wu || N <= 0 || K <= 0) return GNX_ERR_BAD_ARG;
    winograd_weights_kernel<<<gnx_cdiv(3L * N * K, 256), 256, 0, stream>>>(w, wu, N, K);
    return gnx_launch_status();
}

// out[M][32] (ldc) = conv3x3_pad1(A[M = imgs*S*S][K] (lda)) for an operand that needs no prologue, with Winograd F(2,3)
// along x (1.5x fewer matrix operations than the direct form; results differ by rounding only).  Shapes: N == 32,
// 32 | K, S in {4, 8, 16, 32, 64}, 16-B aligned operands; anything else returns GNX_ERR_UNSUPPORTED (use
// gnx_conv3x3_bnrelu with scale = shift = NULL and the [tap][N][K] weights).
GNX_EXPORT int gnx_conv3x3_winograd(const float* A, long lda, const float* Wu, float* out, long ldc, long M, int N, int K,
                                    int S, hipStream_t stream) {
    if (!A || !Wu || !out || M < 0 || N <= 0 || K <= 0 || S <= 0 || lda < K || ldc < N || (M % ((long)S * S)) != 0)
        return GNX_ERR_BAD_ARG;
    if (N != 32 || (K & 31) != 0 || !al16(A) || !al16(Wu) || lda % 4 != 0 ||
        M * (lda > ldc ? lda : ldc) >= (1L << 31))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    const long wgs = (M + 255) / 256 > 256 ? 256 : (M + 255) / 256;
#define GNX_WINO(SS)                                                                                        \
    conv3x3_wino_kernel<SS><<<(int)wgs, 256, 0, stream>>>(A, (int)lda, Wu, out, (int)ldc, (int)M, K);        \
    return gnx_launch_status()
    switch (S) {
        case 4: GNX_WINO(4);
        case 8: GNX_WINO(8);
        case 16: GNX_WINO(16);
        case 32: GNX_WINO(32);
        case 64: GNX_WINO(64);
        default: return GNX_ERR_UNSUPPORTED;
    }
#undef GNX_WINO
}

// out[M][N] (ldc) = conv3x3_pad1(act(A[M = imgs*S*S][K] (lda))) with weights repacked to [tap][N][K]
GNX_EXPORT int gnx_conv3x3_bnrelu(const float* A, long lda, const float* Wr, float* out, long ldc, long M, int N, int K,
                                  int S, const float* scale, const float* shift, hipStream_t stream) {
    if (!A || !Wr || !out || M < 0 || N <= 0 || K <= 0 || S <= 0 || lda < K || ldc < N || (!scale) != (!shift) ||
        (M % ((long)S * S)) != 0)
        return GNX_ERR_BAD_ARG;
    if (M == 0) return GNX_OK;
    const size_t lds_bytes = ((size_t)(C3_BM + 2 * S + 2) * LDK + 9 * 32 * LDK + LDK) * sizeof(float);
    if (lds_bytes > 160 * 1024) return GNX_ERR_UNSUPPORTED;
    static size_t configured = 0;
    if (lds_bytes > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return GNX_ERR_LAUNCH;
        configured = lds_bytes;
    }
    const int vecA = al16(A) && lda % 4 == 0 && K % 4 == 0 && (!scale || (al16(scale) && al16(shift)));
    const int vecW = al16(Wr) && K % 4 == 0;
    dim3 grid(gnx_cdiv(M, C3_BM), gnx_cdiv(N, C3_BN));
    const int nj = gnx_cdiv(C3_BM + 2 * S + 2, 32);
#define GNX_PIPE(NJ)                                                                                              \
    do {                                                                                                          \
        static size_t conf = 0;                                                                                   \
        if (lds_bytes > conf) {                                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pipe_kernel<NJ>),                       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)    \
                return GNX_ERR_LAUNCH;                                                                            \
            conf = lds_bytes;                                                                                     \
        }                                                                                                         \
        conv3x3_pipe_kernel<NJ><<<grid, 256, lds_bytes, stream>>>(A, lda, Wr, out, ldc, M, N, K, S, scale, shift, \
                                                                  vecA, vecW);                                    \
    } while (0)
    const bool fast = vecA && vecW;      // aligned pointers/leading dimensions and K % 4 == 0
    // LDS-DMA persistent form for prologue-free inputs (the eval forward's pre-activated bottleneck)
    if (!scale && fast && N == C3_BN && (K & 63) == 0 && (M % C3_BM) == 0 && M * (lda > ldc ? lda : ldc) < (1L << 31) &&
        !getenv("GNX_NO_DMA")) {
        // variant 1 = 4 waves (128-row tiles), 2 = 8 waves (256-row tiles).  Measured sustained (tools/kbench.py --noact
        // --reps 300): 2 wins wherever its tiles fill the chip (139 vs 133 TFLOP/s), 1 where they quantise badly (S = 4 at
        // 4992 spots: 82 vs 107).  (A third variant - 64-B LDS rows, two 4-wave workgroups per CU - measured like 1.)
        static const int forced = getenv("GNX_DMA_VARIANT") ? atoi(getenv("GNX_DMA_VARIANT")) : -1;
        const int variant = forced >= 0 ? forced : (M / 256 >= 1024 ? 2 : 1);
#define GNX_DMA(SS)                                                                                              \
    do {                                                                                                         \
        if constexpr (SS <= 32) {                                                                                \
            if (variant == 2 && M % 256 == 0 && K % 64 == 0) {                                                   \
                const long wgs = M / 256 > 256 ? 256 : M / 256;                                                  \
                conv3x3_dma_kernel<SS, 8, 32><<<(int)wgs, 512, 0, stream>>>(A, (int)lda, Wr, out, (int)ldc,      \
                                                                            (int)M, K, N);                       \
                return gnx_launch_status();                                                                      \
            }                                                                                                    \
        }                                                                                                        \
        const long wgs = M / 128 > 256 ? 256 : M / 128;                                                          \
        conv3x3_dma_kernel<SS, 4, 32><<<(int)wgs, 256, 0, stream>>>(A, (int)lda, Wr, out, (int)ldc, (int)M, K,   \
                                                                    N);                                          \
        return gnx_launch_status();                                                                              \
    } while (0)
        switch (S) {
            case 4: GNX_DMA(4);
            case 7: GNX_DMA(7);
            case 8: GNX_DMA(8);
            case 14: GNX_DMA(14);
            case 16: GNX_DMA(16);
            case 28: GNX_DMA(28);
            case 32: GNX_DMA(32);
            case 56: GNX_DMA(56);
            case 64: GNX_DMA(64);
            default: break;
        }
#undef GNX_DMA
    }
    // the same kernel in its data-gradient shape (dX = conv3x3(dY, W^T): K = 32 channels in, N = 128 out): column tiles
    // of 32 take the place of K chunks
    if (!scale && fast && K == 32 && (N & 63) == 0 && (M % C3_BM) == 0 && M * (lda > ldc ? lda : ldc) < (1L << 31) &&
        !getenv("GNX_NO_DMA")) {
#define GNX_DMAG(SS)                                                                                             \
    do {                                                                                                         \
        if constexpr (SS <= 32) {                                                                                \
            if (M % 256 == 0 && M / 256 >= 1024) {                                                               \
                conv3x3_dma_kernel<SS, 8, 32, true><<<256, 512, 0, stream>>>(A, (int)lda, Wr, out, (int)ldc,     \
                                                                             (int)M, K, N);                      \
                return gnx_launch_status();                                                                      \
            }                                                                                                    \
        }                                                                                                        \
        const long wgs = M / 128 > 256 ? 256 : M / 128;                                                          \
        conv3x3_dma_kernel<SS, 4, 32, true><<<(int)wgs, 256, 0, stream>>>(A, (int)lda, Wr, out, (int)ldc,        \
                                                                          (int)M, K, N);                         \
        return gnx_launch_status();                                                                              \
    } while (0)
        switch (S) {
            case 4: GNX_DMAG(4);
            case 7: GNX_DMAG(7);
            case 8: GNX_DMAG(8);
            case 14: GNX_DMAG(14);
            case 16: GNX_DMAG(16);
            case 28: GNX_DMAG(28);
            case 32: GNX_DMAG(32);
            case 56: GNX_DMAG(56);
            case 64: GNX_DMAG(64);
            default: break;
        }
#undef GNX_DMAG
    }
    if (fast && nj <= 5) GNX_PIPE(5);
    else if (fast && nj == 6) GNX_PIPE(6);
    else if (fast && nj == 7) GNX_PIPE(7);
    else if (fast && nj <= 9) GNX_PIPE(9);
    else conv3x3_kernel<<<grid, 256, lds_bytes, stream>>>(A, lda, Wr, out, ldc, M, N, K, S, scale, shift, vecA, vecW);
#undef GNX_PIPE
    return gnx_launch_status();
}

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
// conv output is 64 wide and even-high (128-px patches), O <= 64 and O % 4 == 0; anything else returns
// GNX_ERR_UNSUPPORTED and the caller runs gnx_conv_stem + gnx_bnrelu_maxpool.
GNX_EXPORT int gnx_conv_stem_bnrelu_maxpool(const float* x, const float* w, float* out, long ldo, long imgs, int Cin,
                                            int H, int W, int O, int KH, int KW, int stride, int pad, const float* scale,
                                            const float* shift, hipStream_t stream) {
    if (!x || !w || !out || !scale || !shift || imgs < 0 || Cin <= 0 || O <= 0 || H <= 0 || W <= 0 || ldo < O)
        return GNX_ERR_BAD_ARG;
    if (Cin != 3 || KH != 7 || KW != 7 || stride != 2 || pad != 3 || O > 64 || O % 4 != 0 || ldo % 4 != 0 || !al16(out))
        return GNX_ERR_UNSUPPORTED;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if (Wo != 64 || W != 128 || Ho % 2 != 0 || Ho <= 0 || !al16(x)) return GNX_ERR_UNSUPPORTED;    // 16-B row pieces
    if (imgs == 0) return GNX_OK;
    const size_t lds_bytes = ((size_t)64 * (3 * 7 * 8 + 4) + (size_t)3 * 9 * 136 + (size_t)128 * SP_LDT) * sizeof(float);
    static bool conf = false;
    if (!conf) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stem_pool_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return GNX_ERR_LAUNCH;
        conf = true;
    }
    const int grid = (int)(imgs < 256 ? imgs : 256);
    conv_stem_pool_kernel<<<grid, 256, lds_bytes, stream>>>(x, w, out, ldo, H, W, O, scale, shift, imgs);
    return gnx_launch_status();
}

// in [imgs*Hi*Wi][C] (ldi) -> out [imgs*Ho*Wo][C] (ldo): max over 3x3 s2 p1 windows of relu(in*scale+shift)
GNX_EXPORT int gnx_bnrelu_maxpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int Hi, int Wi,
                                  const float* scale, const float* shift, hipStream_t stream) {
    if (!in || !out || !scale || !shift || imgs < 0 || C <= 0 || Hi <= 0 || Wi <= 0 || ldi < C || ldo < C)
        return GNX_ERR_BAD_ARG;
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const long Mout = imgs * Ho * Wo;
    if (Mout == 0) return GNX_OK;
    if (C % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0 && al16(in) && al16(out) && al16(scale) && al16(shift)) {
        long blocks = (Mout * (C / 4) + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        bnrelu_maxpool_vec4_kernel<<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C / 4, Hi, Wi, Ho, Wo,
                                                                     scale, shift);
        return gnx_launch_status();
    }
    long blocks = (Mout * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    bnrelu_maxpool_kernel<<<(int)blocks, 256, 0, stream>>>(in, ldi, out, ldo, Mout, C, Hi, Wi, Ho, Wo, scale, shift);
    return gnx_launch_status();
}

// in [imgs*S2][C] (ldi) -> out [imgs][C] (ldo): mean over positions of relu(in*scale+shift)
GNX_EXPORT int gnx_bnrelu_avgpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S2,
                                  const float* scale, const float* shift, hipStream_t stream) {
    if (!in || !out || !scale || !shift || imgs < 0 || C <= 0 || S2 <= 0 || ldi < C || ldo < C) return GNX_ERR_BAD_ARG;
    if (imgs == 0) return GNX_OK;
    if (imgs > 2147483647L) return GNX_ERR_UNSUPPORTED;
    dim3 grid((unsigned)imgs, gnx_cdiv(C, 64));
    bnrelu_avgpool_kernel<<<grid, 256, 0, stream>>>(in, ldi, out, ldo, C, S2, scale, shift);
    return gnx_launch_status();
}

