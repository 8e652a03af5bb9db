// Backward of the DenseNet stem on the fp16-MFMA gradient path (BASELINE config 5 with f trained):
// features.conv0 -> norm0 -> relu0 -> pool0 (/root/reference/gridnext/densenet.py:105-110, differentiated by torch.autograd under
// training.py:164-171; BatchNorm on running statistics, training.py:126) in ONE kernel that keeps nothing of the forward but
// the patches.
//
// The fp32 form of this adjoint needs, per array of 256-px patches, the window indices of pool0 (recorded by an fp32 stem
// forward six times slower than the fp16 one), a 21-GB fp32 gradient of the conv0 map written by the pool adjoint and read
// back by the weight gradient: 58 ms of a 317-ms step.  Here a persistent workgroup sweeps an image by POOLED rows and for
// pooled row py (conv rows R0 = 2py-1, R1 = 2py, R2 = 2py+1)
//   1. stages the four input rows x 3 channels that are new to its window (fp16, zero-padded, as the forward stages them; the
//      rows live in an LDS ring) and RECOMPUTES the conv rows R1, R2 of relu0(norm0(conv0)) with the forward's own instruction
//      sequence (gnx_conv_stem_bnrelu_maxpool_f16mul: v_mfma_f32_32x32x16_f16 over k = 8 (c 7 + ky) + kx, fp32 accumulate,
//      fp16 result) - bit-identical values, so the maxima are the ones the forward stored; R0 is the previous step's R2;
//   2. finds every window's winner by torch's rule (the first maximal element of the row-major 3 x 3 scan; positions outside
//      the map never win) and routes s * dP[py][px][c] to it (relu0's mask is the winner being > 0) - the gradient of the
//      ACTIVATED conv0 map restricted to pooled row py's windows (norm0's scale0[c] multiplies the fp32 sums at the end);
//      R0's share (plus what the previous step carried for it) and R1's are complete: a quarter of them is written over the
//      activations in LDS as fp16 (a position can win four windows: no overflow); R2's share is CARRIED in registers;
//      norm0's adjoint sums S0 = sum d, S1 = sum d a (a = the winning activation) come off the same registers;
//   3. contracts the rows R0, R1 with the im2col of the staged rows: dW0[o][k] += sum_pos dz[pos][o] col[pos][k], both operands
//      transposed on the way out of LDS (dz by ds_read_b64_tr_b16; the im2col column of lane k = (c, ky, kx) is eight
//      stride-2 halves of one patch row).  The accumulators live in registers for the workgroup's lifetime.
// Every conv row is computed once and contracted once.  Per-workgroup partial sums are reduced in a fixed order
// (deterministic), multiplied by 4 / s.
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(8)));
typedef __fp16 fp16x8 __attribute__((__vector_size__(16)));

constexpr int SB_LDT = 72;      // halves per position of the activation / gradient tile (64 channels + 16 B)
constexpr int SB_NK = 192;      // k columns of a workgroup's weight-gradient slab: 8 (c 7 + ky) + kx, padded to 6 x 32

// see dense_bwd_f16.hip: the 8 contraction elements of one column out of a row-major tile, two transposing reads
__device__ __forceinline__ h8 tr8(const char* lo, const char* hi) {
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)lo);
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)hi);
    const fp16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h8, v);
}

// WO = width of the conv0 map: 64 (128-px patches) or 128 (256-px patches); 64 output channels.  NTHR threads: 256 (two
// workgroups per CU at WO = 64) or 512 (WO = 128, whose 98 KB of LDS allow one workgroup per CU: eight waves, two per SIMD, so
// that one's LDS round trips hide behind the other's matrix / vector work; the two wave quartets take one conv row each of the
// weight-gradient contraction and write a slab each).
//
// Step py of an image handles pooled row py, i.e. conv rows R0 = 2py-1, R1 = 2py, R2 = 2py+1:
//   * input rows live in a RING of 12 slots per channel (row iy in slot (iy + 5) mod 12); a step stages only its four new rows
//     4py+2 .. 4py+5 (two stage-only steps py = -2, -1 bring in the top of an image);
//   * the activated conv rows live in three row slots (row cr in slot (cr + 1) mod 3): R1 and R2 are computed, R0 is the R2 of
//     the step before - still there, because that step kept ITS share of R2's gradient in registers instead of writing it over
//     the activations;
//   * the routed gradient of R0 (the carried share + this step's) and of R1 is complete: it is written over their activations and
//     contracted with the im2col of the staged rows; R2's share is carried.  One extra step py = HP flushes the last row.
// Every conv row is computed once and contracted once.  py mod 3 fixes both the ring phase (4 py mod 12) and the row-slot
// rotation: the step body exists three times with all LDS offsets as instruction immediates.
template <int WO, int NTHR>
__global__ __launch_bounds__(NTHR, WO == 64 ? 2 : 1) void stem_bwd_f16_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
    const _Float16* __restrict__ G, long ldg, float* __restrict__ ws_bn, float* __restrict__ ws_dw, long imgs) {
    constexpr int CIN = 3, KH = 7, KW = 7, STRIDE = 2, PAD = 3;
    constexpr int H = 2 * WO, Wd = 2 * WO;
    constexpr int PH = 12, NEW = 4;                          // ring slots per channel; new input rows per step
    constexpr int PW = ((WO - 1) * STRIDE + 8 + 1 + 1 + 7) & ~7;
    constexpr int F4R = WO * STRIDE / 4;                    // 4-pixel pieces per input row
    constexpr int NW = NTHR / 64, NKH = NW / 4;              // waves; wave quartets (each takes a part of the contraction)
    constexpr int NPC = CIN * NEW * F4R, NPRE = (NPC + NTHR - 1) / NTHR;
    constexpr int NG = CIN * KH, NSTEP = (NG + 1) / 2;      // 21 (c, ky) groups of 8 kx, two per MFMA
    constexpr int LDBH = NSTEP * 16 + 8;                    // halves per weight row (184)
    constexpr int TPR = WO / 32;                            // 32-position tiles per conv row
    static_assert(2 * TPR == NW, "one tile of the two new conv rows per wave");
    constexpr int HP = WO / 2;                              // pooled map side
    constexpr int NGRP = NTHR / 16;                         // pooled-column groups (16 threads = 64 channels each)
    constexpr int NPX = HP / NGRP;                          // pooled columns per thread
    constexpr int NCOL = 2 * NPX + 3;                       // conv columns a thread looks at: 2 px0 - 1 .. 2 (px0 + NPX) + 1
    constexpr int ROWB = WO * SB_LDT * 2;                   // bytes of one row slot of the activation / gradient tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* const Bs = reinterpret_cast<_Float16*>(smem);                 // [64][LDBH]
    _Float16* const Ps = Bs + 64 * LDBH;                                    // [CIN][PH][PW]
    _Float16* const Ts = Ps + CIN * PH * PW;                                // [3 row slots][WO][SB_LDT]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    for (int idx = t; idx < 64 * LDBH; idx += NTHR) {
        const int n = idx / LDBH, rem = idx - n * LDBH;
        const int kx = rem & 7, cky = rem >> 3;
        float v = 0.f;
        if (kx < KW && cky < NG) v = w[((long)n * CIN * KH + cky) * KW + kx];
        Bs[idx] = (_Float16)v;
    }
    for (int idx = t; idx < CIN * PH * PW; idx += NTHR) Ps[idx] = (_Float16)0.f;     // the pad columns stay zero for good
    const _Float16* const pb = Bs + i * LDBH + 8 * h;
    const float sc0 = scale[i], sh0 = shift[i], sc1 = scale[32 + i], sh1 = shift[32 + i];
    // pooling items: 4 channels x NPX pooled columns per thread
    const int c4 = t & 15, pxg = t >> 4, px0 = NPX * pxg;
    float S0[4] = {0.f, 0.f, 0.f, 0.f}, S1[4] = {0.f, 0.f, 0.f, 0.f};
    float cdz[2 * NPX][4];                                   // R2's share of the routed gradient, columns 2 px0 + k: next step's R0
#pragma unroll
    for (int k = 0; k < 2 * NPX; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) cdz[k][e] = 0.f;
    // weight gradient: 2 (o) x 6 (k) tiles of 32 x 32, three per wave of a quartet: tile T = 3 trip + u -> (k tile T >> 1, o tile
    // T & 1)
    const int trip = wave & 3, kh = wave >> 2;
    const int ntA = (3 * trip) >> 1, ntB = (3 * trip + 2) >> 1;
    f32x16 wacc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) wacc[u][r] = 0.f;
    // this lane's im2col column of k tiles A and B: k = 32 nt + i = 8 g + kx, g = c 7 + ky (groups past the 21st do not exist:
    // they re-read the last one and their results are dropped)
    auto col_of = [&](int nt, int& cc, int& ky, int& kx) {
        const int k = 32 * nt + i;
        int g = k >> 3;
        g = g < NG ? g : NG - 1;
        cc = g / KH;
        ky = g % KH;
        kx = k & 7;
    };
    int ccA, kyA, kxA, ccB, kyB, kxB;
    col_of(ntA, ccA, kyA, kxA);
    col_of(ntB, ccB, kyB, kxB);
    const int trow = 8 * h + ((lane & 15) >> 2);
    const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const char* const ptr0 = reinterpret_cast<const char*>(Ts) + trow * (SB_LDT * 2) + tcol * 2;

    float4 pre[NPRE];
    h4 gp[NPX + 1];
    auto fetch = [&](long img, int py) {
        const int iy0 = NEW * py + 2;                        // the step's new input rows
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int j = t + NTHR * q;
            const int f4 = j % F4R, prow = (j / F4R) % NEW, c = (j / F4R) / NEW;
            const int iy = iy0 + prow;
            pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NPC && iy >= 0 && iy < H)
                pre[q] = *reinterpret_cast<const float4*>(x + ((img * CIN + c) * H + iy) * (long)Wd + 4 * f4);
        }
#pragma unroll
        for (int wd = 0; wd <= NPX; ++wd) {
            const int px = px0 + wd;
            const h4 hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            gp[wd] = hz;
            if (px < HP && py >= 0 && py < HP) gp[wd] = *reinterpret_cast<const h4*>(G + ((img * HP + py) * HP + px) * ldg + 4 * c4);
        }
    };
    __syncthreads();
    if ((long)blockIdx.x < imgs) fetch(blockIdx.x, -2);

    for (long img = blockIdx.x; img < imgs; img += gridDim.x) {
        for (int py = -2; py <= HP; ++py) {
            __syncthreads();                                 // the previous step's fragment reads are done
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const int j = t + NTHR * q;
                if (j < NPC) {
                    const int prow = (j / F4R) % NEW, c = (j / F4R) / NEW;
                    const int slot = (NEW * py + 2 + prow + 5 + 2 * PH) % PH;
                    _Float16* d = Ps + (c * PH + slot) * PW + PAD + 4 * (j % F4R);
                    const float4 v = pre[q];
                    d[0] = (_Float16)v.x;
                    const h2 mid = {(_Float16)v.y, (_Float16)v.z};
                    *reinterpret_cast<h2*>(d + 1) = mid;
                    d[3] = (_Float16)v.w;
                }
            }
            h4 g4[NPX + 1];
#pragma unroll
            for (int wd = 0; wd <= NPX; ++wd) g4[wd] = gp[wd];
            {
                long nimg = img;
                int npy = py + 1;
                if (npy > HP) { npy = -2; nimg += gridDim.x; }
                if (nimg >= imgs) nimg = blockIdx.x;
                fetch(nimg, npy);
            }
            asm volatile("" ::: "memory");
            if (py < 0) continue;                            // stage-only steps at the top of an image
            __syncthreads();
            auto step = [&](auto phase) {
                constexpr int PHASE = decltype(phase)::value;
                constexpr int B = 4 * PHASE;                 // ring slot of input row 4py-5 (tap 0 of R0)
                constexpr int SL0 = (2 * PHASE) % 3, SL1 = (2 * PHASE + 1) % 3, SL2 = (2 * PHASE + 2) % 3;
                _Float16* const T0 = Ts + SL0 * (WO * SB_LDT);
                _Float16* const T1 = Ts + SL1 * (WO * SB_LDT);
                _Float16* const T2 = Ts + SL2 * (WO * SB_LDT);
                // ---- 1. the two new conv rows, activated, as the forward computes them (one 32-position tile per wave)
                if (py < HP) {
                    const int ox = 32 * (wave % TPR) + i;
                    const _Float16* const pa = Ps + STRIDE * ox;
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
                    auto multiply = [&](auto rc) {
                        constexpr int RREL = decltype(rc)::value;       // conv row R<RREL>
#pragma unroll
                        for (int s = 0; s < NSTEP; ++s) {
                            const int g0 = 2 * s, g1 = 2 * s + 1 < NG ? 2 * s + 1 : NG - 1;
                            const int off0 = ((g0 / KH) * PH + (B + 2 * RREL + g0 % KH) % PH) * PW;
                            const int off1 = ((g1 / KH) * PH + (B + 2 * RREL + g1 % KH) % PH) * PW;
                            const uint32_t* ap = reinterpret_cast<const uint32_t*>(pa + (h ? off1 : off0));
                            typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
                            const u32x4a av = {ap[0], ap[1], ap[2], ap[3]};
                            const h8 a = __builtin_bit_cast(h8, av);
                            const h8 b0 = *reinterpret_cast<const h8*>(pb + 16 * s);
                            const h8 b1 = *reinterpret_cast<const h8*>(pb + 32 * LDBH + 16 * s);
                            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, acc1, 0, 0, 0);
                        }
                    };
                    const bool second = wave >= TPR;                    // (wave-uniform) waves TPR.. take R2
                    if (second) multiply(std::integral_constant<int, 2>{});
                    else multiply(std::integral_constant<int, 1>{});
                    _Float16* const Td = second ? T2 : T1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pos = 32 * (wave % TPR) + (r & 3) + 8 * (r >> 2) + 4 * h;
                        Td[pos * SB_LDT + i] = (_Float16)fmaxf(fmaf(acc0[r], sc0, sh0), 0.f);
                        Td[pos * SB_LDT + 32 + i] = (_Float16)fmaxf(fmaf(acc1[r], sc1, sh1), 0.f);
                    }
                }
                __syncthreads();
                // ---- 2. pool0's adjoint: winners of windows px0 .. px0 + NPX (the last one for the column it shares with this
                //         thread's last window), the gradient of columns 2 px0 .. 2 (px0 + NPX) - 1
                h4 ycol[3][NCOL];                                       // column 2 px0 - 1 + k of R0, R1, R2
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int k = 0; k < NCOL; ++k) {
                        const int col = 2 * px0 - 1 + k;
                        const h4 none = {(_Float16)(-1.f), (_Float16)(-1.f), (_Float16)(-1.f), (_Float16)(-1.f)};
                        const _Float16* const Tr = rr == 0 ? T0 : rr == 1 ? T1 : T2;
                        ycol[rr][k] = none;
                        if (col >= 0 && col < WO && !(rr == 0 && py == 0))      // conv row -1 (py = 0): never a winner
                            ycol[rr][k] = *reinterpret_cast<const h4*>(Tr + col * SB_LDT + 4 * c4);
                    }
                float dzv[NPX + 1][4];
                int widx[NPX + 1][4];
#pragma unroll
                for (int wd = 0; wd <= NPX; ++wd) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 m = (_Float16)(-1.f);
                        int idx = 15;
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                            for (int dc = 0; dc < 3; ++dc) {
                                const _Float16 v = ycol[rr][2 * wd + dc][e];
                                if (v > m) { m = v; idx = 3 * rr + dc; }
                            }
                        const float mf = (float)m;
                        const float gy = mf > 0.f ? (float)g4[wd][e] : 0.f;
                        if (wd < NPX) {
                            S0[e] += gy;
                            S1[e] += gy * mf;
                        }
                        dzv[wd][e] = gy;
                        widx[wd][e] = idx;
                    }
                }
                auto share = [&](int rr, int k, int e) {                // this step's windows' share of (row rr, column 2 px0 + k)
                    const int wd = k >> 1;
                    if ((k & 1) == 0) return widx[wd][e] == 3 * rr + 1 ? dzv[wd][e] : 0.f;
                    return (widx[wd][e] == 3 * rr + 2 ? dzv[wd][e] : 0.f) + (widx[wd + 1][e] == 3 * rr ? dzv[wd + 1][e] : 0.f);
                };
                h4 dz[2][2 * NPX];
#pragma unroll
                for (int k = 0; k < 2 * NPX; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (a position can win up to four windows, each share an fp16 value: a quarter of the sum cannot
                        // overflow; the reduction multiplies by 4)
                        dz[0][k][e] = (_Float16)(0.25f * (cdz[k][e] + share(0, k, e)));
                        dz[1][k][e] = (_Float16)(0.25f * share(1, k, e));
                        cdz[k][e] = share(2, k, e);
                    }
                __syncthreads();                                        // every activation has been read
#pragma unroll
                for (int k = 0; k < 2 * NPX; ++k) {
                    *reinterpret_cast<h4*>(T0 + (2 * px0 + k) * SB_LDT + 4 * c4) = dz[0][k];
                    *reinterpret_cast<h4*>(T1 + (2 * px0 + k) * SB_LDT + 4 * c4) = dz[1][k];
                }
                __syncthreads();
                // ---- 3. dW0[o][k] += sum over the positions of R0 and R1 of dz[pos][o] col[pos][k]
                constexpr int NKS = 2 * WO / 16 / NKH;                   // 16-position steps per quartet
#pragma unroll
                for (int kk = 0; kk < NKS; ++kk) {
                    // NKH = 2: quartet kh takes conv row R<kh>;  NKH = 1: the first half of the steps is R0, the second R1
                    const int rrel = NKH == 2 ? kh : (kk >= NKS / 2 ? 1 : 0);
                    const int ox0 = NKH == 2 ? 16 * kk : 16 * (kk % (NKS / 2));
                    const _Float16* const pcA = Ps + (ccA * PH + (B + 2 * rrel + kyA) % PH) * PW + kxA + 16 * h + STRIDE * ox0;
                    const _Float16* const pcB = Ps + (ccB * PH + (B + 2 * rrel + kyB) % PH) * PW + kxB + 16 * h + STRIDE * ox0;
                    const char* const pz = ptr0 + (rrel ? SL1 : SL0) * ROWB + ox0 * (SB_LDT * 2);
                    const h8 a0 = tr8(pz, pz + 4 * SB_LDT * 2);
                    const h8 a1 = tr8(pz + 64, pz + 64 + 4 * SB_LDT * 2);
                    h8 bA, bB;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bA[j] = pcA[2 * j];
                        bB[j] = pcB[2 * j];
                    }
                    if (trip & 1) {
                        wacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bA, wacc[0], 0, 0, 0);
                        wacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bB, wacc[1], 0, 0, 0);
                        wacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bB, wacc[2], 0, 0, 0);
                    } else {
                        wacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bA, wacc[0], 0, 0, 0);
                        wacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bA, wacc[1], 0, 0, 0);
                        wacc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bB, wacc[2], 0, 0, 0);
                    }
                }
            };
            switch (py % 3) {
                case 0: step(std::integral_constant<int, 0>{}); break;
                case 1: step(std::integral_constant<int, 1>{}); break;
                default: step(std::integral_constant<int, 2>{}); break;
            }
        }
    }
    // ---- the workgroup's slabs
    {
        float* const out = ws_dw + ((long)blockIdx.x * NKH + kh) * 64 * SB_NK;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int T = 3 * trip + u, nt = T >> 1, mt = T & 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[o * SB_NK + 32 * nt + i] = wacc[u][r];
            }
        }
    }
    __syncthreads();
    float* const red = reinterpret_cast<float*>(Ts);         // [column groups][S0 | S1][64]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[pxg * 128 + 4 * c4 + e] = S0[e];
        red[pxg * 128 + 64 + 4 * c4 + e] = S1[e];
    }
    __syncthreads();
    if (t < 128) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NGRP; ++k) s += red[k * 128 + t];
        ws_bn[(long)blockIdx.x * 128 + t] = s;
    }
}

// dW0[o][c][ky][kx] (+)= 4 scale0[o] / s sum over workgroups of ws[b][o][8 (c 7 + ky) + kx]; a 256-thread block owns 16 elements x
// 16 slab lanes
__global__ __launch_bounds__(256) void stem_dw_reduce_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ dW,
                                                             const float* __restrict__ scale, const float* __restrict__ ls,
                                                             int accumulate, int* __restrict__ flag) {
    const int el = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    constexpr int NEL = 64 * 147;
    float s = 0.f;
    if (el < NEL) {
        const int o = el / 147, rem = el - o * 147, cky = rem / 7, kx = rem - cky * 7;
        const float* p = ws + (long)o * SB_NK + 8 * cky + kx;
        for (int b = sl; b < nblk; b += 16) s += p[(long)b * 64 * SB_NK];
    }
    __shared__ float part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 16 && el < NEL) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += part[threadIdx.x + 16 * j];
        s *= 4.f * ls[1] * scale[el / 147];                   // (the kernel contracts a quarter of the routed gradient)
        if (accumulate) s += dW[el];
        dW[el] = s;
        if (flag && !(fabsf(s) <= 3.0e38f)) atomicOr(flag, 1);
    }
}

// norm0's sums: slabs [b][S0 | S1][64] -> dbeta = S0 / s, dgamma = (S1 - beta S0) / gamma / s (a = gamma x_hat + beta where d != 0)
__global__ __launch_bounds__(256) void stem_bn_reduce_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ ls,
                                                             int accumulate, int* __restrict__ flag) {
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
    float s0 = 0.f, s1 = 0.f;
    for (int b = sl; b < nblk; b += 16) {
        s0 += ws[(long)b * 128 + c];
        s1 += ws[(long)b * 128 + 64 + c];
    }
    __shared__ float part[2][256];
    part[0][threadIdx.x] = s0;
    part[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < 16) {
        s0 = s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s0 += part[0][threadIdx.x + 16 * j];
            s1 += part[1][threadIdx.x + 16 * j];
        }
        const float inv = ls[1];
        float dg = (s1 - beta[c] * s0) / gamma[c] * inv, db = s0 * inv;
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

template <int WO>
constexpr size_t stem_bwd_lds() {
    return 2 * (size_t)(64 * (11 * 16 + 8) + 3 * 12 * (((WO - 1) * 2 + 8 + 1 + 1 + 7) & ~7) + 3 * WO * SB_LDT);
}
int stem_bwd_grid(long imgs, int P) {
    const long cap = P == 128 ? 512 : 256;
    return (int)(imgs < cap ? imgs : cap);
}
int stem_bwd_slabs(long imgs, int P) { return stem_bwd_grid(imgs, P) * (P == 128 ? 1 : 2); }

}  // namespace

// The stem's backward on the fp16 gradient path.  x [imgs][3][P][P] float patches (P = 128 or 256), w = conv0.weight
// [64][3][7][7], scale / shift = norm0 folded (running statistics), gamma / beta = norm0.weight / .bias, G16 = the block-1
// gradient buffer [imgs (P/4)^2][ldg] halves whose first 64 columns hold s x the gradient of the pooled stem map.
// dW [64][3][7][7], dgamma, dbeta [64] (fp32, (+)= when accumulate; any of them may be NULL).  ls = {s, 1/s} on the device.
GNX_EXPORT long gnx_stem_bwd_f16_workspace(long imgs, int P) {
    return (long)stem_bwd_grid(imgs, P) * 128 + (long)stem_bwd_slabs(imgs, P) * 64 * SB_NK;
}
GNX_EXPORT int gnx_stem_bwd_f16(const float* x, const float* w, const float* scale, const float* shift, const float* gamma,
                                const float* beta, const void* G16, long ldg, float* dW, float* dgamma, float* dbeta,
                                float* workspace, long imgs, int P, int O, const float* ls, int accumulate, int* flag,
                                hipStream_t stream) {
    if (!x || !w || !scale || !shift || !gamma || !beta || !G16 || !workspace || !ls || imgs <= 0 || ldg < 64) return GNX_ERR_BAD_ARG;
    if ((P != 128 && P != 256) || O != 64 || ldg % 4 || (reinterpret_cast<uintptr_t>(G16) & 7) ||
        (reinterpret_cast<uintptr_t>(x) & 15))
        return GNX_ERR_UNSUPPORTED;
    const int grid = stem_bwd_grid(imgs, P);
    float* const ws_bn = workspace;
    float* const ws_dw = workspace + (long)grid * 128;
    const _Float16* G = reinterpret_cast<const _Float16*>(G16);
    static bool conf = false;
    if (!conf) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_f16_kernel<128, 512>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_bwd_lds<128>()) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(stem_bwd_f16_kernel<64, 256>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)stem_bwd_lds<64>()) != hipSuccess)
            return GNX_ERR_LAUNCH;
        conf = true;
    }
    if (P == 256)
        stem_bwd_f16_kernel<128, 512><<<grid, 512, stem_bwd_lds<128>(), stream>>>(x, w, scale, shift, G, ldg, ws_bn, ws_dw, imgs);
    else
        stem_bwd_f16_kernel<64, 256><<<grid, 256, stem_bwd_lds<64>(), stream>>>(x, w, scale, shift, G, ldg, ws_bn, ws_dw, imgs);
    if (dW)
        stem_dw_reduce_kernel<<<(64 * 147 + 15) / 16, 256, 0, stream>>>(ws_dw, stem_bwd_slabs(imgs, P), dW, scale, ls, accumulate,
                                                                       flag);
    if (dgamma || dbeta) stem_bn_reduce_kernel<<<4, 256, 0, stream>>>(ws_bn, grid, dgamma, dbeta, gamma, beta, ls, accumulate, flag);
    return gnx_launch_status();
}
