// DenseNet-BC forward: the 3x3 convolution of the dense layers (conv2, densenet.py:41) in its four forms - generic,
// register-pipelined (with BN+ReLU prologue), LDS-DMA direct (ready operand; also the data-gradient shape) and Winograd
// F(2,3) along x - see the file header of conv1x1.hip for the layout conventions.
#include "fwd_common.h"

namespace {

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

// Requires (checked by the dispatcher): M % (32 NW) == 0, M * max(lda, ldc) < 2^31, N == 32, K % (2 KC) == 0, 16-B
// aligned pointers and leading dimensions.  NW waves per workgroup, each owning 32 output rows of the 32*NW-row tile;
// KC = K elements per chunk (32: 128-B LDS rows, 1 workgroup per CU; 16: 64-B rows, half the LDS, 2 workgroups per CU
// whose barriers, prologues and stores then hide behind each other's MFMAs).  With KC = 16 a group of 16 rows is 1 KB
// (chunk c at c * 256, four chunks) and one DMA instruction writes a whole group.
// NK1 (data-gradient shape: K == KC = 32 input channels, N = 32 * NT output channels, NT even): a tile's chunks are its
// NT column tiles instead of K chunks - the strip is re-fetched (L2) with each column tile's weights, every chunk starts
// from zero accumulators and ends with its 16 stores.
// H16 (config 5's fp16 path): the operands are fp16 in HBM - the activated bottleneck gnx_conv1x1_bnrelu_f16_act16 stores and
// the tap-major weights rounded once.  Two halves are one "float" to everything but the multiply: A / Wr / lda / K are passed
// in float units (K/2, lda/2), a 128-B LDS row holds 64 channels, and a fragment read (16 B = 8 consecutive k of a lane's
// row; the two lane halves 16 k) feeds ONE v_mfma_f32_32x32x16_f16 where the fp32 form issues four 32x32x2.
// O16: the output is fp16 too ([M][N] halves, ldc in halves): config 5 with fp16 block buffers.
// BNADJ (round 2, with NK1 and N == 128: conv2's data gradient fused with norm2 -> relu2's adjoint in eval-statistics mode):
// the column-tile store, instead of writing g = dL/d(conv2 input), reads the ACTIVATED bottleneck a[m][c] (bn.a), and writes
//   out[m][c] = sc[c] g[m][c] where a > 0, else 0                                   (dL/d(conv1 output))
// while the column sums sum_m g [a > 0] and sum_m g [a > 0] xhat, xhat = ((a - sh) / sc - mean) invstd, of everything the
// workgroup processes stay in registers and leave as ONE slab per wave: bn.slab [gridDim.x * NW][2][N] (dbeta, dgamma).
// The separate pass (read g, read a, write the result: 25 ms of a 289-ms step) disappears.
struct C3BnAdj {
    const float* a; int lda;
    const float* sc; const float* sh; const float* mean; const float* inv;
    float* slab;
};
template <int S, int NW, int KC, bool NK1 = false, bool H16 = false, bool O16 = false, bool BNADJ = false>
__global__ __launch_bounds__(64 * NW) void conv3x3_dma_kernel(const float* __restrict__ A, int lda,
                                                              const float* __restrict__ Wr, float* __restrict__ out,
                                                              int ldc, int M, int K, int N, C3BnAdj bn = C3BnAdj()) {
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
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wr), 0, 9 * N * K * 4, 0x00020000);
    auto issue_slot = [&](auto slot_c, char* dst) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (slot < NSA) {
            const int p = wave + NW * slot;                                    // piece: group p / PPG, part p % PPG
            if ((NPA % NW) && slot == NSA - 1 && p >= NPA) return;
            const int grp = p / PPG, part = p % PPG;
            const int row0 = ntile * BM - S - 1 + 16 * grp;                    // first strip row of the group
            char* d = dst + grp * GB + part * 1024;
            if (__builtin_expect(row0 >= 0 && row0 + 15 < M, 1)) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(A + (long)row0 * lda), 0, (15 * lda + K) * 4, 0x00020000);
                dma16_buf(rs, voffA, ((NK1 ? 0 : KC * nchunk) + 16 * part) * 4, lds_addr(d));
            } else {
                // array ends: rows outside [0, M) are only ever "read" by masked taps, any in-range row will do
                int Pr = row0 + (lane & 15);
                Pr = Pr < 0 ? 0 : (Pr >= M ? M - 1 : Pr);
                dma16_global(A + (long)Pr * lda + (NK1 ? 0 : KC * nchunk) + 16 * part + 4 * (lane >> 4), lds_addr(d));
            }
        } else {
            const int p = wave + NW * (slot - NSA);                            // weight group p / PPG = 2 tap + (n >> 4)
            if ((NPW % NW) && slot == NSA + NSW - 1 && p >= NPW) return;
            // A tile of exactly two chunks (fp16 operands, 128 bottleneck channels: K = 64 float units) always puts chunk 0 in
            // buffer 0 and chunk 1 in buffer 1: the weight images are the same bytes for every tile, so they are fetched with
            // the workgroup's first two chunks only and stay RESIDENT.  (They were 74 of the 139 KB a 128-row tile moved
            // through the LDS-DMA path, and that path - ~6.4 TB/s chip-wide - is what bounded the fp16 kernel.)
            if (H16 && !NK1 && nk2 == 1 && nround > 0) return;
            const int grp = p / PPG, part = p % PPG;
            // weight rows [tap][N][K]: tap = grp >> 1, n = 32 * (column tile) + 16 * (grp & 1) + lane row
            dma16_buf(rW, voffW,
                      (((grp >> 1) * N + (NK1 ? 32 * nchunk : 0) + (grp & 1) * 16) * K + (NK1 ? 0 : KC * nchunk) + 16 * part) * 4,
                      lds_addr(dst + SR * ROWB + grp * GB + part * 1024));
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
    float adj_b[4] = {0.f, 0.f, 0.f, 0.f}, adj_g[4] = {0.f, 0.f, 0.f, 0.f};      // BNADJ: this lane's column of column tile ct
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

        auto store16 = [&](int col0, auto ct_c) {
            if constexpr (BNADJ) {
                constexpr int ctc = decltype(ct_c)::value;
                const int col = col0 + i;
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (long)tile * BM * ldc, 0, BM * ldc * 4,
                                                                                    0x00020000);
                const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(bn.a + (long)tile * BM * bn.lda), 0, BM * bn.lda * 4, 0x00020000);
                const int vo = ((32 * wave + 4 * h) * ldc + col) * 4, va = ((32 * wave + 4 * h) * bn.lda + col) * 4;
                float av[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    av[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, va, ((r & 3) + 8 * (r >> 2)) * bn.lda * 4, 0));
                const float sc = bn.sc[col], sh = bn.sh[col], mu = bn.mean[col], is = bn.inv[col];
                const float rsc = 1.f / sc;
                float sb = adj_b[ctc], sg = adj_g[ctc];
#pragma unroll
                for (int r = 0; r < 16; ++r) {             // exactly 16 (buffer) stores
                    const float d = av[r] > 0.f ? acc0[r] + acc1[r] : 0.f;
                    sb += d;
                    sg = fmaf(d, ((av[r] - sh) * rsc - mu) * is, sg);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc * d), ro, vo, ((r & 3) + 8 * (r >> 2)) * ldc * 4, 0);
                }
                adj_b[ctc] = sb;
                adj_g[ctc] = sg;
                stored = true;
                return;
            }
            if constexpr (O16) {
                _Float16* o16 = reinterpret_cast<_Float16*>(out);
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(o16 + (long)tile * BM * ldc, 0,
                                                                                    BM * ldc * 2, 0x00020000);
                const int vo = ((32 * wave + 4 * h) * ldc + col0 + i) * 2;
#pragma unroll
                for (int r = 0; r < 16; ++r) {             // exactly 16 (buffer) stores
                    const _Float16 hv = (_Float16)(acc0[r] + acc1[r]);
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, hv), ro, vo,
                                                          ((r & 3) + 8 * (r >> 2)) * ldc * 2, 0);
                }
                stored = true;
                return;
            }
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + (long)tile * BM * ldc, 0, BM * ldc * 4,
                                                                                0x00020000);
            const int vo = ((32 * wave + 4 * h) * ldc + col0 + i) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)                   // exactly 16 (buffer) stores
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[r] + acc1[r]), ro, vo,
                                                      ((r & 3) + 8 * (r >> 2)) * ldc * 4, 0);
            stored = true;
        };
        auto do_chunk = [&](auto par_c, int ct, auto ct_c) {
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
                if constexpr (H16) {
                    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
                    const half8 ah = __builtin_bit_cast(half8, a), bh = __builtin_bit_cast(half8, bq);
                    if constexpr (step & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bq[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bq[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], bq[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bq[3], acc1, 0, 0, 0);
                }
#if GNX_DMA_DBG != 1
                if constexpr (step < NSA + NSW) issue_slot(step_c, nxt);
#endif
                if constexpr (step < STEPS - 1) { a = na; bq = nb; }
            });
            advance_next();
            if constexpr (NK1) store16(32 * ct, ct_c);
        };
        if constexpr (BNADJ) {                             // N == 128: four column tiles, their sums in named registers
            do_chunk(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{});
            do_chunk(std::integral_constant<int, 1>{}, 1, std::integral_constant<int, 1>{});
            do_chunk(std::integral_constant<int, 0>{}, 2, std::integral_constant<int, 2>{});
            do_chunk(std::integral_constant<int, 1>{}, 3, std::integral_constant<int, 3>{});
        } else {
            for (int c2 = 0; c2 < nk2; ++c2) {
                do_chunk(std::integral_constant<int, 0>{}, 2 * c2, std::integral_constant<int, 0>{});
                do_chunk(std::integral_constant<int, 1>{}, 2 * c2 + 1, std::integral_constant<int, 0>{});
            }
        }
        if constexpr (!NK1) store16(0, std::integral_constant<int, 0>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BNADJ) {
        // one slab per wave: [2][N]; the two lane halves hold different rows of the same column
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float sb = adj_b[ct] + __shfl_xor(adj_b[ct], 32), sg = adj_g[ct] + __shfl_xor(adj_g[ct], 32);
            if (h == 0) {
                float* sl = bn.slab + (long)(blockIdx.x * NW + wave) * 2 * N + 32 * ct + i;
                sl[0] = sb;
                sl[N] = sg;
            }
        }
    }
}

// slab[nblk][2][C] -> dbeta[c] (+)= sum_b slab[b][0][c], dgamma[c] (+)= sum_b slab[b][1][c].  64 columns per workgroup, 16
// row lanes per column each summing a contiguous sixteenth of the slabs in index order (16 loads in flight), combined in a
// fixed order: deterministic.  (Four row lanes took 17 us per call - a chain of 64 dependent batches - 58 times a step.)
constexpr int C3_ADJ_LANES = 16;
__global__ __launch_bounds__(64 * C3_ADJ_LANES) void c3_adj_reduce_kernel(const float* __restrict__ slab, int nblk, int C,
                                                                          float* __restrict__ dbeta,
                                                                          float* __restrict__ dgamma, int accumulate) {
    __shared__ float part[C3_ADJ_LANES][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cl;                  // over 2 C values: q = idx / C, c = idx % C
    float s0 = 0.f;
    if (idx < 2 * C) {
        const int q = idx / C, c = idx - q * C;
        const int per = (nblk + C3_ADJ_LANES - 1) / C3_ADJ_LANES;
        int b = rl * per;
        const int be = b + per < nblk ? b + per : nblk;
        for (; b + 16 <= be; b += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slab[((long)(b + u) * 2 + q) * C + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) s0 += v[u];
        }
        for (; b < be; ++b) s0 += slab[((long)b * 2 + q) * C + c];
    }
    part[rl][cl] = s0;
    __syncthreads();
    if (rl == 0 && idx < 2 * C) {
        const int q = idx / C, c = idx - q * C;
        float tot = 0.f;
#pragma unroll
        for (int u = 0; u < C3_ADJ_LANES; ++u) tot += part[u][cl];
        float* dst = q == 0 ? dbeta : dgamma;
        if (dst) dst[c] = accumulate ? dst[c] + tot : tot;
    }
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
    const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wu), 0, 12 * 32 * K * 4, 0x00020000);
    auto issue_slot = [&](auto slot_c, int buf) {
        constexpr int slot = decltype(slot_c)::value;
        if constexpr (slot < NSR) {
            const int p = wave + 4 * slot;
            if ((NPR % 4) && slot == NSR - 1 && p >= NPR) return;
            const int row0 = ntile * BM - S - 1 + 16 * p;
            char* d = lds + OFF_RAW + buf * RAWB + p * 1024;
            if (__builtin_expect(row0 >= 0 && row0 + 15 < M, 1)) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(A + (long)row0 * lda), 0, (15 * lda + K) * 4, 0x00020000);
                dma16_buf(rs, voffA, 64 * nchunk, lds_addr(d));
            } else {                                       // array ends: any in-range row (only masked rows use it)
                int Pr = row0 + pix;
                Pr = Pr < 0 ? 0 : (Pr >= M ? M - 1 : Pr);
                dma16_global(A + (long)Pr * lda + 16 * nchunk + 4 * (lane >> 4), lds_addr(d));
            }
        } else {
            const int p = wave + 4 * (slot - NSR);         // U group p: tap p >> 1, rows 16 (p & 1) ..
            dma16_buf(rU, voffU, (16 * p * K + 16 * nchunk) * 4, lds_addr(lds + OFF_U + buf * UB + p * 1024));
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
        // buffer stores (free beside MFMA waves where a global_store is not); the resource ends with the array, so the rows
        // a ragged last tile does not have are dropped by the bounds check: always exactly 32 stores
        {
            const int left = M - tile * BM;
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                out + (long)tile * BM * ldc, 0, (left < BM ? left : BM) * ldc * 4, 0x00020000);
            const int vo = (2 * (32 * wave + 4 * h) * ldc + i) * 4;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc[0][0][e] + acc[0][1][e], m1 = acc[1][0][e] + acc[1][1][e];
                const float m2 = acc[2][0][e] + acc[2][1][e], m3 = acc[3][0][e] + acc[3][1][e];
                const int row = 2 * ((e & 3) + 8 * (e >> 2));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((m0 + m1) + m2), ro, vo, row * ldc * 4, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((m1 - m2) - m3), ro, vo, (row + 1) * ldc * 4, 0);
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

}  // namespace

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
// floats of workspace for gnx_conv3x3_dgrad_bnrelu_bwd: one [2][N] slab per wave of at most 256 workgroups of 8 waves
GNX_EXPORT long gnx_conv3x3_dgrad_bn_workspace(long M, int N) { return 256L * 8 * 2 * N; }

// conv2's data gradient fused with norm2 -> relu2's adjoint (eval statistics, activated bottleneck: gnx_bn_relu_bwd(relu = 2)):
//   dX[m][c] = scale[c] g[m][c] [A_act[m][c] > 0],   g = conv3x3(dY, Wb)[m][c]          (c < N = 128, dY has K = 32 channels)
//   dbeta[c] (+)= sum_m g [a > 0],   dgamma[c] (+)= sum_m g [a > 0] ((a - shift) / scale - mean) invstd
// == gnx_conv3x3_bnrelu(dY, Wb -> g) followed by gnx_bn_relu_bwd(g, A_act, relu = 2, training = 0), without the pass over
// [M][128] in between.  Wb: gnx_repack_conv3x3_bwd's layout.  Shapes the LDS-DMA kernel takes in its data-gradient form with
// N == 128 only; GNX_ERR_UNSUPPORTED otherwise (run the two calls).
GNX_EXPORT int gnx_conv3x3_dgrad_bnrelu_bwd(const float* dY, long lddy, const float* Wb, const float* A_act, long lda,
                                            float* dX, long lddx, long M, int N, int K, int S, const float* scale,
                                            const float* shift, const float* mean, const float* invstd, float* dgamma,
                                            float* dbeta, int accumulate, float* workspace, hipStream_t stream) {
    if (!dY || !Wb || !A_act || !dX || !scale || !shift || !mean || !invstd || !workspace || M < 0 || N <= 0 || K <= 0 ||
        S <= 0 || lddy < K || lda < N || lddx < N || (M % ((long)S * S)) != 0)
        return GNX_ERR_BAD_ARG;
    if (K != 32 || N != 128 || (M % C3_BM) != 0 || !al16(dY) || !al16(Wb) || lddy % 4 != 0 ||
        M * (lddy > lddx ? (lddy > lda ? lddy : lda) : (lddx > lda ? lddx : lda)) >= (1L << 31))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    C3BnAdj bn = {A_act, (int)lda, scale, shift, mean, invstd, workspace};
    int nblk = 0;
#define GNX_DMAA(SS)                                                                                              \
    do {                                                                                                          \
        if constexpr (SS <= 32) {                                                                                 \
            if (M % 256 == 0 && M / 256 >= 1024) {                                                                \
                nblk = 256 * 8;                                                                                   \
                conv3x3_dma_kernel<SS, 8, 32, true, false, false, true><<<256, 512, 0, stream>>>(                 \
                    dY, (int)lddy, Wb, dX, (int)lddx, (int)M, K, N, bn);                                          \
                break;                                                                                            \
            }                                                                                                     \
        }                                                                                                         \
        const long wgs = M / 128 > 256 ? 256 : M / 128;                                                           \
        nblk = (int)wgs * 4;                                                                                      \
        conv3x3_dma_kernel<SS, 4, 32, true, false, false, true><<<(int)wgs, 256, 0, stream>>>(                    \
            dY, (int)lddy, Wb, dX, (int)lddx, (int)M, K, N, bn);                                                  \
    } while (0)
    switch (S) {
        case 4: GNX_DMAA(4); break;
        case 8: GNX_DMAA(8); break;
        case 16: GNX_DMAA(16); break;
        case 32: GNX_DMAA(32); break;
        case 64: GNX_DMAA(64); break;
        default: return GNX_ERR_UNSUPPORTED;
    }
#undef GNX_DMAA
    if (dgamma || dbeta)
        c3_adj_reduce_kernel<<<gnx_cdiv(2 * N, 64), 64 * C3_ADJ_LANES, 0, stream>>>(workspace, nblk, N, dbeta, dgamma, accumulate);
    return gnx_launch_status();
}

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
    if (!scale && fast && N == C3_BN && (K & 63) == 0 && (M % C3_BM) == 0 && M * (lda > ldc ? lda : ldc) < (1L << 31)) {
        // variant 1 = 4 waves (128-row tiles), 2 = 8 waves (256-row tiles).  Measured sustained (tools/kbench.py --noact
        // --reps 300): 2 wins wherever its tiles fill the chip (139 vs 133 TFLOP/s), 1 where they quantise badly (S = 4 at
        // 4992 spots: 82 vs 107).  (A third variant - 64-B LDS rows, two 4-wave workgroups per CU - measured like 1.)
        const int variant = M / 256 >= 1024 ? 2 : 1;
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
    if (!scale && fast && K == 32 && (N & 63) == 0 && (M % C3_BM) == 0 && M * (lda > ldc ? lda : ldc) < (1L << 31)) {
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

// conv3x3 (pad 1) of an fp16, already activated operand (config 5): A16 [M][K] halves (lda16), Wr16 [9][N][K] halves
// (gnx_repack_conv3x3's layout rounded to fp16), fp32 accumulation, fp32 out [M][N] (ldc).  conv3x3_dma_kernel in its H16
// form: N == 32, 128 | K, 128 | M, 16-B aligned; S as the fp32 DMA path.  Anything else: GNX_ERR_UNSUPPORTED.
template <bool O16>
static int conv3x3_f16_dma_launch(const void* A16, long lda16, const void* Wr16, float* out, long ldc, long M, int N, int K,
                                  int S, hipStream_t stream) {
    if (!A16 || !Wr16 || !out || M < 0 || N <= 0 || K <= 0 || S <= 0 || lda16 < K || ldc < N || (M % ((long)S * S)) != 0)
        return GNX_ERR_BAD_ARG;
    if (N != C3_BN || (K & 127) != 0 || (M % C3_BM) != 0 || (lda16 & 7) != 0 || !al16(A16) || !al16(Wr16) ||
        M * (lda16 > ldc ? lda16 : ldc) >= (1L << 31))
        return GNX_ERR_UNSUPPORTED;
    if (M == 0) return GNX_OK;
    const float* A = reinterpret_cast<const float*>(A16);
    const float* Wr = reinterpret_cast<const float*>(Wr16);
    const int Kf = K / 2, ldaf = (int)(lda16 / 2);
    const int variant = M / 256 >= 1024 ? 2 : 1;
#define GNX_DMAH(SS)                                                                                             \
    do {                                                                                                         \
        if constexpr (SS <= 32) {                                                                                \
            if (variant == 2 && M % 256 == 0) {                                                                  \
                const long wgs = M / 256 > 256 ? 256 : M / 256;                                                  \
                conv3x3_dma_kernel<SS, 8, 32, false, true, O16><<<(int)wgs, 512, 0, stream>>>(A, ldaf, Wr, out,       \
                                                                                         (int)ldc, (int)M, Kf, N); \
                return gnx_launch_status();                                                                      \
            }                                                                                                    \
        }                                                                                                        \
        const long wgs = M / 128 > 256 ? 256 : M / 128;                                                          \
        conv3x3_dma_kernel<SS, 4, 32, false, true, O16><<<(int)wgs, 256, 0, stream>>>(A, ldaf, Wr, out, (int)ldc,     \
                                                                                 (int)M, Kf, N);                 \
        return gnx_launch_status();                                                                              \
    } while (0)
    switch (S) {
        case 4: GNX_DMAH(4);
        case 8: GNX_DMAH(8);
        case 16: GNX_DMAH(16);
        case 32: GNX_DMAH(32);
        case 64: GNX_DMAH(64);
        default: break;
    }
#undef GNX_DMAH
    return GNX_ERR_UNSUPPORTED;
}

GNX_EXPORT int gnx_conv3x3_f16_dma(const void* A16, long lda16, const void* Wr16, float* out, long ldc, long M, int N, int K,
                                   int S, hipStream_t stream) {
    return conv3x3_f16_dma_launch<false>(A16, lda16, Wr16, out, ldc, M, N, K, S, stream);
}
// the same with an fp16 output [M][N] (ldc16 in halves): config 5 with fp16 block buffers
GNX_EXPORT int gnx_conv3x3_f16_dma_h(const void* A16, long lda16, const void* Wr16, void* out16, long ldc16, long M, int N,
                                     int K, int S, hipStream_t stream) {
    return conv3x3_f16_dma_launch<true>(A16, lda16, Wr16, reinterpret_cast<float*>(out16), ldc16, M, N, K, S, stream);
}
