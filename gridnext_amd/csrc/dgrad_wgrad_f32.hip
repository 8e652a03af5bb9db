// conv1's backward of the f-trained step in ONE fp32 pass (round 4): data gradient + norm1 -> relu1's adjoint accumulated into
// the block gradient, the BatchNorm sums AND conv1's weight gradient, from the same staged tiles.
// (torch.autograd through /root/reference/gridnext/densenet.py:35-37 under training.py:126, :164-171; fp32 = the reference's
// arithmetic.)  It replaces gnx_conv1x1_dgrad_bnrelu_bwd + gnx_wgrad_bnrelu(taps = 1), which stream the same [M][cin] strips
// and the same [M][128] bottleneck gradient twice (the second pass was 42 of the 250 ms of an f-trained 128-px array).
//
// Structure = csrc/dense_bwd_f16.hip's conv1 kernel with v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per instruction):
//   * a workgroup (4 waves) owns 128 input channels (a wave 32) and a contiguous range of 32-pixel tiles; its slice of conv1's
//     transposed weight lives in REGISTERS as MFMA fragments for its lifetime (64 registers per lane);
//   * per tile: dB [32][128] and X [32][128 of cin] go global -> registers -> LDS (16-B pieces, the next tile's loads in flight
//     while this one multiplies);
//   * data gradient D = [pixel][channel]: A = dB rows (one ds_read_b128 feeds four MFMAs: the k order of the weight fragments is
//     permuted to match), B = the weight fragments; a lane then holds ONE channel's 16 pixels, so the adjoint needs that
//     channel's three constants and two running sums in registers, and the block gradient is read-modified-written straight
//     from the accumulator layout (a wave instruction = two rows' 128-B segments), its old values requested before the products;
//   * weight gradient D = [m][channel], contraction over the tile's 32 pixels: for the fp32 MFMA both operands are plain
//     row reads of the row-major tiles (A[i][k] = dB[pixel 2 s + h][m i], B[k][j] = act(X[pixel 2 s + h][channel j])): five
//     ds_read_b32 per four MFMAs, the activation applied to the fragment (one value per lane and step);
//   * with both products the kernel is bound by the matrix pipe (2 x 2 x 128 flops per element against 16 bytes), not by HBM.
// Sums over pixels leave as per-workgroup slabs reduced in a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int F_RS = 132;          // floats per row of a [pixel][128] LDS tile: 16 consecutive rows' 16-B pieces on 16 bank quads

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}
__device__ __forceinline__ void lds_barrier32() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// WPB = waves (32-channel columns) per workgroup.  WPB = 4: the full 128-channel blocks, next tile prefetched into registers.
// WPB = 1 .. 3: the cin mod 128 channels that are left - a launch of narrower workgroups over FOUR TIMES as many pixel slabs per
// wave fewer, so that every SIMD still has its two waves (a 4-wave workgroup with one live wave would take a full block's time
// for a quarter of its work: cin = 160 ran at 67 TFLOP/s against 100 at cin = 128).
template <int WPB>
__global__ __launch_bounds__(64 * WPB, 2) void dgrad_wgrad1x1_f32_kernel(
    const float* __restrict__ dB, long lddb, const float* __restrict__ W1t, const float* __restrict__ X, long ldx,
    float* __restrict__ G, long ldg, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, float* __restrict__ ws, float* __restrict__ wsw, long M, int c_first, int n_cb, long cw,
    long tiles_per_slab, long n_slabs, int K) {
    constexpr int NT = 64 * WPB, XW = 32 * WPB, XP = XW / 4, X_RS = XW + 4;
    __shared__ __attribute__((aligned(16))) float smem[32 * F_RS + 32 * X_RS];
    float* const Bt = smem;
    float* const Xt = smem + 32 * F_RS;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int rem8 = blockIdx.x % (8 * n_cb), cb = rem8 / 8;               // the channel blocks of one slab share an XCD
    const long slab = (long)(blockIdx.x / (8 * n_cb)) * 8 + (rem8 & 7);
    if (slab >= n_slabs) return;
    const int cbase = c_first + cb * XW;
    const int ch = cbase + 32 * wave + r;                                  // THIS LANE'S input channel (both products' column)
    const bool active = cbase + 32 * wave < K;      // (a last block of three columns runs as a 4-wave workgroup with one idle wave)
    const long ntiles = M / 32;
    const long tile0 = slab * tiles_per_slab;
    const long tile1 = tile0 + tiles_per_slab < ntiles ? tile0 + tiles_per_slab : ntiles;
    const float sc = active ? scale[ch] : 0.f, sh = active ? shift[ch] : 0.f, mu = active ? mean[ch] : 0.f;
    // weight fragments: MFMA (jj, e) of the data gradient multiplies bottleneck channel m = 8 jj + 4 h + e (the order in which a
    // lane's ds_read_b128 of dB delivers them)
    f32x4 wf[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) wf[jj] = active ? ldg4(W1t + (long)ch * 128 + 8 * jj + 4 * h) : zero4();
    float S0 = 0.f, S1 = 0.f;
    f32x16 wacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) wacc[i][j] = 0.f;
    const int chunk = t & 31, row0 = t >> 5;                               // WPB = 4: 32 sixteen-byte pieces per row, 8 rows per pass
    const bool cok = cbase + chunk * 4 < K;
    f32x4 bv[WPB == 4 ? 4 : 1], xv[WPB == 4 ? 4 : 1];
    auto fetch = [&](long tile) {                                          // (WPB = 4 only)
        const long m0 = tile * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = m0 + row0 + 8 * i;
            bv[i] = ldg4(dB + row * lddb + chunk * 4);
            xv[i] = cok ? ldg4(X + row * ldx + cbase + chunk * 4) : zero4();
        }
    };
    if constexpr (WPB == 4)
        if (tile0 < tile1) fetch(tile0);
    for (long tile = tile0; tile < tile1; ++tile) {
        const long m0 = tile * 32;
        lds_barrier32();                                            // the previous tile's fragment reads are done
        if constexpr (WPB == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<f32x4*>(Bt + (row0 + 8 * i) * F_RS + chunk * 4) = bv[i];
                *reinterpret_cast<f32x4*>(Xt + (row0 + 8 * i) * X_RS + chunk * 4) = xv[i];
            }
        } else {
            // narrow workgroups: straight through (four of them share a CU; one's wait is the others' matrix time)
            for (int p = t; p < 32 * 32; p += NT)
                *reinterpret_cast<f32x4*>(Bt + (p >> 5) * F_RS + (p & 31) * 4) = ldg4(dB + (m0 + (p >> 5)) * lddb + (p & 31) * 4);
            for (int p = t; p < 32 * XP; p += NT)
                *reinterpret_cast<f32x4*>(Xt + (p / XP) * X_RS + (p % XP) * 4) = ldg4(X + (m0 + p / XP) * ldx + cbase + (p % XP) * 4);
        }
        lds_barrier32();
        if constexpr (WPB == 4)
            if (tile + 1 < tile1) fetch(tile + 1);                  // in flight while this tile multiplies
        if (active) {
            // the block gradient's old values in the accumulator layout (lane = channel: a wave instruction reads two rows'
            // 128-B segments), requested now, needed after both products.  A buffer resource over the tile's 32 rows: one
            // per-lane offset register and wave-uniform row offsets instead of 16 per-lane 64-bit addresses
            float gold[16];
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(G + m0 * ldg, 0, (int)(32 * ldg * 4), 0x00020000);
            const int vo = (int)((4 * h * ldg + ch) * 4);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                gold[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rg, vo, (int)(((i & 3) + 8 * (i >> 2)) * ldg * 4), 0));
            // ---- weight gradient: dW1[m][c] += sum_p dB[p][m] relu(scale x + shift)[p][c]   (D = [m][c], k = pixels)
            // (software-pipelined by hand: step s + 1's five operands are requested before step s multiplies - left to itself
            // the compiler waits for each pair of reads right in front of the two MFMAs that use it)
            float an[4], xn;
            xn = Xt[h * X_RS + 32 * wave + r];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) an[mt] = Bt[h * F_RS + 32 * mt + r];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt] = an[mt];
                const float b = fmaxf(fmaf(xn, sc, sh), 0.f);
                if (s + 1 < 16) {
                    xn = Xt[(2 * s + 2 + h) * X_RS + 32 * wave + r];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) an[mt] = Bt[(2 * s + 2 + h) * F_RS + 32 * mt + r];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) wacc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b, wacc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- data gradient: D[p][c] = sum_m dB[p][m] W1[m][c]   (A = dB rows, B = the weight fragments)
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(Bt + r * F_RS + 8 * jj + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], wf[jj][e], acc, 0, 0, 0);
            }
            // ---- norm1 -> relu1's adjoint on the accumulators: this lane's channel, 16 pixels
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int px = (i & 3) + 8 * (i >> 2) + 4 * h;
                const float x = Xt[px * X_RS + 32 * wave + r];
                const float d = fmaf(x, sc, sh) > 0.f ? acc[i] : 0.f;
                S0 += d;
                S1 += d * (x - mu);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(d, sc, gold[i])), rg, vo, (int)(((i & 3) + 8 * (i >> 2)) * ldg * 4), 0);
            }
        }
    }
    S0 += __shfl_xor(S0, 32, 64);
    S1 += __shfl_xor(S1, 32, 64);
    if (h == 0 && active) {
        float* const out = ws + slab * 2 * cw;
        out[ch] = S0;
        out[cw + ch] = S1;
    }
    if (active) {
        float* const out = wsw + slab * 128 * cw + ch;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) out[(long)(32 * mt + (i & 3) + 8 * (i >> 2) + 4 * h) * cw] = wacc[mt][i];
    }
}

// 16 slab lanes per output element, fixed tree (as dense_bwd_f16.hip's reductions)
__global__ __launch_bounds__(256) void dw_reduce_f32_kernel(const float* __restrict__ wsw, long nslab, int K, long cw, int c0,
                                                            int nc, float* __restrict__ dW, int accumulate) {
    const int sl = threadIdx.x >> 4;
    const long e = (long)blockIdx.x * 16 + (threadIdx.x & 15);       // element of [128][nc]
    const long row = e / nc;
    const int col = c0 + (int)(e - row * nc);
    const bool ok = row < 128;
    const long n = 128 * cw, i = row * cw + col;
    float s = 0.f;
    if (ok)
        for (long k = sl; k < nslab; k += 16) s += wsw[k * n + i];
    __shared__ float part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 16 && ok) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += part[threadIdx.x + 16 * j];
        float* const o = dW + row * K + col;
        *o = accumulate ? *o + tot : tot;
    }
}
__global__ __launch_bounds__(256) void bn_reduce_f32_kernel(const float* __restrict__ ws, long nslab, long cw, int c0, int nc,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            const float* __restrict__ invstd, int accumulate) {
    const int j0 = blockIdx.x * 16 + (threadIdx.x & 15);
    const int c = c0 + j0;
    const int sl = threadIdx.x >> 4;
    float s0 = 0.f, s1 = 0.f;
    if (j0 < nc)
        for (long k = sl; k < nslab; k += 16) {
            s0 += ws[(2 * k) * cw + c];
            s1 += ws[(2 * k + 1) * cw + c];
        }
    __shared__ float part[2][256];
    part[0][threadIdx.x] = s0;
    part[1][threadIdx.x] = s1;
    __syncthreads();
    if (threadIdx.x < 16 && j0 < nc) {
        s0 = s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s0 += part[0][threadIdx.x + 16 * j];
            s1 += part[1][threadIdx.x + 16 * j];
        }
        const float dg = s1 * invstd[c];
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + dg : dg;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s0 : s0;
    }
}

struct Plan32 {
    long slabs, per;
};
Plan32 plan32(long M, long want) {
    const long tiles = M / 32;
    long slabs = want < 1 ? 1 : want;
    if (slabs > tiles) slabs = tiles;
    if (slabs < 1) slabs = 1;
    Plan32 p;
    p.per = (tiles + slabs - 1) / slabs;
    p.slabs = (tiles + p.per - 1) / p.per;
    return p;
}
// full 128-channel blocks: two 4-wave workgroups per CU; the left-over columns: eight waves per CU in narrower workgroups
// (three left-over columns run as one more 4-wave block with an idle wave: measured faster than a 3-wave launch)
int full_blocks(int K) { return K / 128 + ((K % 128) / 32 == 3 ? 1 : 0); }
int rest_cols(int K) { return (K % 128) / 32 == 3 ? 0 : (K % 128) / 32; }
Plan32 plan_full(long M, int K) { return plan32(M, (512 + full_blocks(K) - 1) / (full_blocks(K) > 0 ? full_blocks(K) : 1)); }
Plan32 plan_rest(long M, int K) { return plan32(M, 2048 / (rest_cols(K) > 0 ? rest_cols(K) : 1)); }
bool al16f(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// G[:, :K] += scale * (dB . W1) * [scale X + shift > 0]; dgamma / dbeta; dW[128][K] = dB^T relu(scale X + shift): the fp32
// forms of gnx_conv1x1_dgrad_bnrelu_bwd and gnx_wgrad_bnrelu(taps = 1) in one pass.  dB [M][128] (lddb), W1t = conv1.weight
// transposed [K][128], 32 | K, 32 | M, 4 | ld*, 16-B aligned; anything else: GNX_ERR_UNSUPPORTED (callers run the two kernels).
GNX_EXPORT long gnx_conv1x1_dgrad_wgrad_workspace(long M, int K) {
    if (M < 32 || K < 32) return 0;
    const long cw = (long)((K + 127) / 128) * 128;
    const long a = full_blocks(K) ? plan_full(M, K).slabs : 0, b = rest_cols(K) ? plan_rest(M, K).slabs : 0;
    return (a + b) * (2L + 128) * cw;
}
GNX_EXPORT int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd(const float* dB, long lddb, const float* W1t, const float* X, long ldx, float* G,
                                                  long ldg, long M, int K, const float* scale, const float* shift,
                                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dW,
                                                  float* workspace, int accumulate, hipStream_t stream) {
    if (!dB || !W1t || !X || !G || !scale || !shift || !mean || !invstd || !dW || !workspace || M <= 0 || K <= 0 || lddb < 128 ||
        ldx < K || ldg < K)
        return GNX_ERR_BAD_ARG;
    if (K % 32 || M % 32 || lddb % 4 || ldx % 4 || ldg % 4 || ldg >= (1 << 20) || !al16f(dB) || !al16f(W1t) || !al16f(X) ||
        !al16f(G))
        return GNX_ERR_UNSUPPORTED;
    const long cw = (long)((K + 127) / 128) * 128;
    const int n_full = full_blocks(K), rest = rest_cols(K);
    const int c_full = n_full * 128 < K ? n_full * 128 : K;                  // channels the full-block launch covers
    float* wsp = workspace;
    if (n_full) {
        const Plan32 p = plan_full(M, K);
        float* const ws = wsp;
        float* const wsw = wsp + p.slabs * 2 * cw;
        wsp += p.slabs * (2 + 128) * cw;
        const int grid = (int)((p.slabs + 7) / 8 * 8 * n_full);
        dgrad_wgrad1x1_f32_kernel<4><<<grid, 256, 0, stream>>>(dB, lddb, W1t, X, ldx, G, ldg, scale, shift, mean, ws, wsw, M, 0, n_full,
                                                              cw, p.per, p.slabs, K);
        if (dgamma || dbeta)
            bn_reduce_f32_kernel<<<(c_full + 15) / 16, 256, 0, stream>>>(ws, p.slabs, cw, 0, c_full, dgamma, dbeta, invstd, accumulate);
        dw_reduce_f32_kernel<<<(int)((128L * c_full + 15) / 16), 256, 0, stream>>>(wsw, p.slabs, K, cw, 0, c_full, dW, accumulate);
    }
    if (rest) {
        const Plan32 p = plan_rest(M, K);
        float* const ws = wsp;
        float* const wsw = wsp + p.slabs * 2 * cw;
        const int grid = (int)((p.slabs + 7) / 8 * 8);
        const int c0 = c_full;
#define GNX_REST(W)                                                                                                          \
    dgrad_wgrad1x1_f32_kernel<W><<<grid, 64 * W, 0, stream>>>(dB, lddb, W1t, X, ldx, G, ldg, scale, shift, mean, ws, wsw, M, c0, 1, cw, \
                                                             p.per, p.slabs, K)
        if (rest == 1) GNX_REST(1);
        else GNX_REST(2);
#undef GNX_REST
        if (dgamma || dbeta)
            bn_reduce_f32_kernel<<<(32 * rest + 15) / 16, 256, 0, stream>>>(ws, p.slabs, cw, c0, 32 * rest, dgamma, dbeta, invstd,
                                                                           accumulate);
        dw_reduce_f32_kernel<<<(int)((128L * 32 * rest + 15) / 16), 256, 0, stream>>>(wsw, p.slabs, K, cw, c0, 32 * rest, dW, accumulate);
    }
    return gnx_launch_status();
}
