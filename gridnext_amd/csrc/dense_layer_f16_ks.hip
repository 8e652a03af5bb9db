// The fused fp16 dense layer (norm1 -> relu1 -> conv1 1x1 K -> 128 -> norm2 -> relu2 -> conv2 3x3 128 -> 32,
// /root/reference/gridnext/densenet.py:35-44) in its K-SPLIT form for 64 x 64 and 32 x 32 maps: the 128 bottleneck channels are
// split over the four waves of a workgroup and NEVER leave their registers between conv1 and conv2.
//
// Why (tools/ubench/conv2_regs.hip, conv2_operands.hip; profiles/r05c_conv2_regs.txt): conv2 as shipped in dense_layer_f16.hip
// reads both MFMA operands from LDS (1 141 TFLOP/s chip-wide with four waves per CU, and the clock falls to 1.5 GHz under the
// LDS traffic); with the operands in registers the same loop runs at 1 680-1 940.  What makes that possible:
//  * conv1 computes D[channel][pixel]; a wave owns bottleneck channels 32 w .. 32 w + 31 of all 128 pixels of a step.  Its
//    accumulators, activated and packed to fp16 IN PLACE, are already the B operand of conv2's MFMA for this wave's k-slice when
//    W2's k order follows the accumulator layout (k-step s, lane half h, element e <-> channel 16 s + 8 (e >> 2) + 4 h + (e & 3)).
//  * W2's slice of the wave (9 taps x 2 k-steps) is 72 VGPRs, loaded once per launch.
//  * dy taps are other registers (a 32-pixel MFMA tile is one image row at S = 32, one x-parity of a row at S = 64);
//    dx = -1 / +1 taps are DPP lane shifts of a fragment (wave_shr:1 / wave_shl:1 plus a row shift that repairs the seam of
//    the 32-lane halves: two VALU moves per register, zero shifted in = the image border).  At S = 64 a row's even and odd
//    pixels are separate tiles, so half of the shifted fragments are the other tile unshifted.
//  * the four k-slices' partial sums of an output tile are added through LDS (fp32, fixed order wave 0..3: deterministic).
// Two workgroups of four waves per CU.  Every wave stages its quarter of the input itself: global -> registers -> norm1 + relu1
// -> LDS slot, one pair of 32-channel stages per workgroup barrier; conv1 runs as two passes of 64 pixels per 128-pixel step
// (32 accumulator registers instead of 64: the register file is what bounds this form); all loads are inline asm with
// hand-counted vmcnt, per pair and wave in the order W1(pair p + 1) x 4, x(pair p + 3) x 2 (see pair_iter).
// Out rows complete one step late (they need the next image row): the fragments of the previous step's last two rows stay
// in registers; an image's last row is finished during the next image's first step (or a final flush).
// STATUS: correct (tests/test_gpu_kernels.py::test_dense_layer_f16_fused[form 1]) and NOT the default - equal or slower than
// dense_layer_f16.hip on config 5's shapes (DESIGN.md Appendix A has the stamps and the reasons); selected by
// gnx_dense_layer_f16_set_form(1).
#include "fwd_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t(), 0, 0, 0)) u32x4;

constexpr int KS_SLOT = 4096;                  // one activated stage of a pass: 64 px x 32 channels (slots 0..3 at LDS offset 0)
constexpr int KS_PART = 4 * KS_SLOT;           // partial sums: [tile of the half 2][wave 4] x 4 KB
constexpr int KS_OT = KS_PART + 8 * 4096;      // norm2: scale[128], shift[128]
constexpr int KS_CT = KS_OT + 1024;            // norm1: [stage][16-B column 4][scale 8 | shift 8] floats (K <= 512)
constexpr int KS_LDS = KS_CT + 16 * 256;
static_assert(2 * KS_LDS <= 160 * 1024, "two workgroups per CU");

// t = src moved one lane up (RIGHT: lane n takes lane n - 1) or down inside each 32-lane half, zero shifted in at the end of
// the half (= the image border).  Two DPP moves per register: wave_shr / wave_shl with a zero for the lane that has no source,
// then a row shift restricted (row_mask / bank_mask act on the destination only) to the four lanes at the seam of the halves,
// which rewrites the one lane the wave shift filled from the other half with the zero it needs.  (An EXEC mask that leaves
// that lane untouched does not work: a DPP read from a lane EXEC disables is invalid too, so its neighbour loses its source.)
// The moves are the compiler's to schedule between the MFMAs: up to ~5 VALU issues hide in the shadow of a 32x32x16 MFMA.
template <bool RIGHT>
__device__ __forceinline__ void ks_shift(u32x4 (&t)[2], const u32x4 (&s)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = (int)s[k][q];
            int r;
            if (RIGHT) {
                r = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);     // wave_shr:1, lane 0 <- 0
                r = __builtin_amdgcn_update_dpp(r, v, 0x111, 0x4, 0x1, true);     // row_shr:1 on lanes 32..35: lane 32 <- 0
            } else {
                r = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true);     // wave_shl:1, lane 63 <- 0
                r = __builtin_amdgcn_update_dpp(r, v, 0x101, 0x2, 0x8, true);     // row_shl:1 on lanes 28..31: lane 31 <- 0
            }
            t[k][q] = (unsigned)r;
        }
}

__device__ __forceinline__ f32x16 ks_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// Diagnostic build only (tools/ubench/ks_stamps.py compiles this file with -DGNX_KS_STAMP into its own library): waves 0 and 2
// of every workgroup sum the shader cycles they spend in each segment of a step.  The product build has no stamp.
#ifdef GNX_KS_STAMP
#define KS_STAMP_PARAM , unsigned long long* __restrict__ stamps
#define KS_T0() unsigned long long ks_t = __builtin_readcyclecounter(), ks_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define KS_LAP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); ks_acc[k] += n_ - ks_t; ks_t = n_; } while (0)
#define KS_OUT() do { if (stamps && lane == 0 && (wave & 1) == 0) for (int q_ = 0; q_ < 8; ++q_) stamps[((long)blockIdx.x * 2 + (wave >> 1)) * 8 + q_] = ks_acc[q_]; } while (0)
#else
#define KS_STAMP_PARAM
#define KS_T0() do {} while (0)
#define KS_LAP(k) do {} while (0)
#define KS_OUT() do {} while (0)
#endif

// ONEPAIR: K = 64 (two stages = one pair per pass): the pair's slot parity is the pass, no null pair pads the pass.
template <int S, bool ONEPAIR>
__global__ __launch_bounds__(256, 2) void dense_layer_f16_ks_kernel(_Float16* __restrict__ X, long bstride, int n_img, int K,
                                                                    const _Float16* __restrict__ w1p,
                                                                    const _Float16* __restrict__ w2p,
                                                                    const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                                    const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                                    _Float16* __restrict__ Aout, long abstride
                                                                    KS_STAMP_PARAM) {
    static_assert(S == 64 || S == 32, "map sizes of this form");
    constexpr int J = S * S / 128;                             // steps per image (2 rows at S = 64, 4 at S = 32)
    __shared__ __attribute__((aligned(16))) char lds[KS_LDS];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int nst = K >> 5, KS = K >> 4;
    // pairs per pass, rounded up to even (a null pair: staging and a barrier only) unless the pass is one pair
    const int npe = ONEPAIR ? 1 : (((nst + 1) >> 1) + 1) & ~1;
    const unsigned lb = lds_addr(lds);

    // ---- tables, W2 slice
    if (t < 128) {
        reinterpret_cast<float*>(lds + KS_OT)[t] = sc2[t];
        reinterpret_cast<float*>(lds + KS_OT + 512)[t] = sh2[t];
    }
    for (int k = t; k < K; k += 256) {
        float* d = reinterpret_cast<float*>(lds + KS_CT) + (k >> 3) * 16 + (k & 7);
        d[0] = sc1[k];
        d[8] = sh1[k];
    }
    // W2 fragments of this wave's k-slice, re-ordered from the packed image ((tap * 8 + ks) * 64 + lane) * 8 + q =
    // W2[lane & 31][16 ks + 8 (lane >> 5) + q][tap]: element e of (k-step s, lane (m, h)) is channel 32 w + 16 s + 8 (e >> 2) +
    // 4 h + (e & 3) = packed k-step 2 w + s, lane m + 32 (e >> 2), q = 4 h + (e & 3)
    u32x4 w2r[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        const char* p = reinterpret_cast<const char*>(w2p) + ((((f >> 1) * 8 + 2 * wave + (f & 1)) * 64 + i) * 16 + 8 * h);
        const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 32 * 16);
        w2r[f] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing of the compiler's own stays in flight behind here

    // ---- norm1 + relu1 on a raw piece (a lane holds 16 B = 8 channels of one pixel): fp32 fma on the fp16 value, one rounding
    struct ActRegs { f32x4 s0, s1, b0, b1; };
    auto act2 = [](unsigned x, float sa, float ba, float sb, float bb) {
        unsigned r;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_pk_max_f16 %0, %0, 0"
            : "=&v"(r) : "v"(x), "v"(sa), "v"(ba), "v"(sb), "v"(bb));
        return r;
    };
    auto activated = [&](const u32x4& v, const ActRegs& c) {
        u32x4 o;
        o[0] = act2(v[0], c.s0[0], c.b0[0], c.s0[1], c.b0[1]);
        o[1] = act2(v[1], c.s0[2], c.b0[2], c.s0[3], c.b0[3]);
        o[2] = act2(v[2], c.s1[0], c.b1[0], c.s1[1], c.b1[1]);
        o[3] = act2(v[3], c.s1[2], c.b1[2], c.s1[3], c.b1[3]);
        return o;
    };

    // ---- lane geometry.  A step is 128 pixels (2 image rows at S = 64, 4 at S = 32) whose conv1 runs as two PASSES of 64
    // pixels (32 accumulator registers per wave instead of 64: the register file is what bounds this kernel).  Staging: wave w
    // loads the pass's pixels 16 w .. 16 w + 15 of a stage (1 KB contiguous in the channel-blocked buffer): lane = (pixel
    // j = lane & 15, 16-B column c = lane >> 4).  MFMA tile beta (0, 1) of the pass, lane n: S = 32: pixel 32 beta + n (a tile =
    // an image row); S = 64: x = 2 n + beta (the pass is one image row).  LDS slot: byte(tile position p = 32 beta + n, column) =
    // (p >> 4) * 1024 + column * 256 + (p & 15) * 16, at S = 64 the last term XOR 128 in the odd tile (the even and the odd
    // pixels of a staged piece then fall into different banks).
    const int sj = lane & 15, scol = lane >> 4;
    const int xvoff = sj * 64 + scol * 16;
    unsigned wa;                                               // LDS write offset of the wave's piece inside a slot
    unsigned rdE, rdO;                                         // fragment read bases (even / odd tile)
    if constexpr (S == 32) {
        wa = lb + wave * 1024 + scol * 256 + sj * 16;
        rdE = rdO = lb + (i >> 4) * 1024 + h * 256 + (i & 15) * 16;
    } else {
        wa = lb + (2 * (sj & 1) + (wave >> 1)) * 1024 + scol * 256 + (((8 * (wave & 1) + (sj >> 1)) * 16) ^ ((sj & 1) << 7));
        rdE = lb + (i >> 4) * 1024 + h * 256 + (i & 15) * 16;
        rdO = lb + (i >> 4) * 1024 + h * 256 + (((i & 15) * 16) ^ 128);
    }
    const unsigned ctb = lb + KS_CT + 64 * scol;
    const __amdgpu_buffer_rsrc_t rW1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w1p), 0, (unsigned)(128 * K * 2), 0x00020000);

    // ---- the two load streams (inline asm, hand-counted).  x: the wave's 16 pixels of stage (2 cp + st) of pass cq of step
    // (cu, cj); per pair of stages the wave issues x(a) W(a) W(a) x(b) W(b) W(b).
    u32x4 xr[2][2], wr[2][2];                                  // x: [pair parity][stage]; W1: [stage][k-step]
    int cu = bid, cj = 0, cq = 0, cp = 0;                      // cursor of the x stream: image, step, pass, pair
    auto load_x = [&](u32x4& dst, int st) {
        const int s = 2 * cp + st;
        const bool in = cu < n_img && s < nst;
        const _Float16* base =
            X + (in ? s : 0) * bstride + ((((long)(in ? cu : bid)) * J + cj) * 128 + 64 * cq + 16 * wave) * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, 1024, 0x00020000);
        const int vo = xvoff + (in ? 0 : 0x7f000000);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(vo), "s"(rs) : "memory");
    };
    auto advance_x = [&]() {
        if (++cp == npe) {
            cp = 0;
            if (++cq == 2) {
                cq = 0;
                if (++cj == J) { cj = 0; cu += G; }
            }
        }
    };
    auto load_w = [&](u32x4(&dst)[2], int s) {                 // W1 fragments (k-steps 2 s, 2 s + 1) of this wave's 32 channels
        const int vo = s < nst ? (wave * KS + 2 * s) * 1024 + lane * 16 : 0x7f000000;
        asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\tbuffer_load_dwordx4 %1, %2, %3, 0 offen offset:1024"
                     : "=&v"(dst[0]), "=&v"(dst[1]) : "v"(vo), "s"(rW1) : "memory");
    };
    auto request_k = [&](ActRegs& ak, int ss) {
        const unsigned a = ctb + ss * 256;
        ak.s0 = lds_read4<0>(a);
        ak.s1 = lds_read4<16>(a);
        ak.b0 = lds_read4<32>(a);
        ak.b1 = lds_read4<48>(a);
    };
    // norm1 + relu1 of a landed piece -> slot `slot` (its LDS write is one entry of the lgkm queue)
    auto stage_out = [&](auto slot_c, const u32x4& v, const ActRegs& ak) {
        constexpr int slot = decltype(slot_c)::value;
        const u32x4 o = activated(v, ak);
        const unsigned a = wa;
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(o), "n"(slot * KS_SLOT) : "memory");
    };

    f32x16 c1[2];
    KS_T0();
    // One pair of stages: staging of the NEXT pair of the stream (into the slots pair p does not use), MFMAs of pair p (slots
    // 2 PAR, 2 PAR + 1), the W1 loads of the next pair, the x loads of the pair THREE ahead.  Both stages of a pair go through
    // each phase together: one exposed LDS latency for the norm1 constants, one for the eight fragments.  Loads return in
    // order, so every wait for a W1 fragment also waits for every x load issued before it: the x loads go LAST in the
    // iteration, behind the W1 requests - an x load issued at the end of iteration p is first waited on two iterations later,
    // where it is staged.  Queue of outstanding loads at entry, oldest first: x(p+1, a) x(p+1, b) | W(p, a) W(p, a) W(p, b)
    // W(p, b) x(p+2, a) x(p+2, b).
    auto pair_iter = [&](auto par_c, auto first_c, int p) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        const int pn = p + 1 == npe ? 0 : p + 1;               // the next pair of the stream (its stages: 2 pn, 2 pn + 1)
        {
            ActRegs ka, kb;
            request_k(ka, 2 * pn < nst ? 2 * pn : 0);
            request_k(kb, 2 * pn + 1 < nst ? 2 * pn + 1 : 0);
            // both stages of the next pair have landed (six younger loads), their constants too
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)"
                         : "+v"(xr[1 - PAR][0]), "+v"(xr[1 - PAR][1]), "+v"(ka.s0), "+v"(ka.s1), "+v"(ka.b0), "+v"(ka.b1),
                           "+v"(kb.s0), "+v"(kb.s1), "+v"(kb.b0), "+v"(kb.b1)::"memory");
            KS_LAP(0);
            stage_out(std::integral_constant<int, 2 * (1 - PAR)>{}, xr[1 - PAR][0], ka);
            stage_out(std::integral_constant<int, 2 * (1 - PAR) + 1>{}, xr[1 - PAR][1], kb);
            KS_LAP(1);
        }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (2 * p < nst) {
            // Pair p has a real stage.  (An odd step's last pair has one: its second stage multiplies W1 fragments requested
            // out of range - zeros - into a slot that holds norm1 + relu1 of zeros, finite values: the products are exact zeros.
            // No branch may separate a load's issue from its wait: hipcc copies registers at control-flow merges, and a copy of
            // a register whose load is still in flight - invisible to it - reads garbage.)
            u32x4 av[8];
            static_for<0, 8>([&](auto n_c) {                   // n = 4 st + 2 ks + beta
                constexpr int n = decltype(n_c)::value, st = n >> 2, beta = n & 1;
                av[n] = __builtin_bit_cast(u32x4, lds_read4<(2 * PAR + st) * KS_SLOT + beta * 2048 + ((n >> 1) & 1) * 512>(
                                                      (S == 64 && beta) ? rdO : rdE));
            });
            // W(p, a): W(p, b) and x(p+2) are younger; W(p, b): x(p+2) are
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(wr[0][0]), "+v"(wr[0][1])::"memory");
            static_for<0, 4>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[n]) : "n"(7 - n));
                c1[n & 1] = ks_mfma(wr[0][n >> 1], av[n], (FIRST && n < 2) ? zero16 : c1[n & 1]);
            });
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(wr[1][0]), "+v"(wr[1][1])::"memory");
            static_for<4, 8>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[n]) : "n"(7 - n));
                c1[n & 1] = ks_mfma(wr[1][(n >> 1) & 1], av[n], c1[n & 1]);
            });
        } else {
            // a null pair (the pass's pair count is rounded up to even): its W1 requests (out of range) have landed too
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        KS_LAP(2);
        load_w(wr[0], 2 * pn);
        load_w(wr[1], 2 * pn + 1);
        load_x(xr[1 - PAR][0], 0);                             // pair p + 3 of the stream, into the registers pair p + 1 left
        load_x(xr[1 - PAR][1], 1);
        advance_x();
        lds_barrier();
        KS_LAP(3);                                                 // pair barrier
    };

    // ---- prologue: pair 0 of the first pass into slots 0, 1; then the canonical queue x(1) | W(0) x(2)
    __syncthreads();                                           // tables
    {
        load_x(xr[0][0], 0);
        load_x(xr[0][1], 1);
        advance_x();
        ActRegs k0, k1;
        request_k(k0, 0);
        request_k(k1, nst > 1 ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(k0.s0), "+v"(k0.s1), "+v"(k0.b0), "+v"(k0.b1), "+v"(k1.s0), "+v"(k1.s1),
                       "+v"(k1.b0), "+v"(k1.b1)::"memory");
        stage_out(std::integral_constant<int, 0>{}, xr[0][0], k0);
        stage_out(std::integral_constant<int, 1>{}, xr[0][1], k1);
        load_x(xr[1][0], 0);                                   // pair 1 (staged in iteration 0, parity 1)
        load_x(xr[1][1], 1);
        advance_x();
        load_w(wr[0], 0);
        load_w(wr[1], 1);
        load_x(xr[0][0], 0);                                   // pair 2
        load_x(xr[0][1], 1);
        advance_x();
        lds_barrier();
    }

    // ---- conv2 state: fragments [k-step] of 32 px x this wave's 32 channels
    constexpr int NPREV = S == 64 ? 4 : 2;                     // S = 64: rows r - 2, r - 1 as (even, odd) tiles; S = 32: 2 rows
    u32x4 prev[NPREV][2], cur[4][2];
    u32x4 tl[2], tr[2];                                        // shifted fragments (dx = -1 / +1 taps)
#pragma unroll
    for (int r = 0; r < NPREV; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) prev[r][s] = u32x4{0u, 0u, 0u, 0u};
    // taps of one in-row into one output tile: dxs = which fragments serve dx = -1, 0, +1
    auto taps3 = [&](f32x16& acc, int dyi, const u32x4 (&L)[2], const u32x4 (&C)[2], const u32x4 (&R)[2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc = ks_mfma(w2r[(dyi * 3 + 0) * 2 + s], L[s], acc);
            acc = ks_mfma(w2r[(dyi * 3 + 1) * 2 + s], C[s], acc);
            acc = ks_mfma(w2r[(dyi * 3 + 2) * 2 + s], R[s], acc);
        }
    };
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
    };
    auto put_partial = [&](const f32x16& acc, int bl) {        // this wave's partial sum of tile bl of the half
        char* d = lds + KS_PART + (bl * 4 + wave) * 4096 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(d + q * 1024) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    };
    // the four partial sums of tile bl, added in wave order, rounded to fp16 -> the layer's new channel block, pixel row `row`
    // (a 32-bit offset into a buffer resource over the block: no per-lane 64-bit address lives anywhere in this kernel)
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(X + (K >> 5) * bstride, 0, (unsigned)(bstride * 2), 0x00020000);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto finish = [&](int bl, unsigned row) {
        const char* d = lds + KS_PART + bl * 4 * 4096 + lane * 16;
        const unsigned lo = row * 64u + 8u * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(d + c * 4096 + q * 1024);
            const f32x4 sum = ((v[0] + v[1]) + v[2]) + v[3];
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)sum[e];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rOut, (int)(lo + 16u * q), 0, 0);
        }
    };

    // conv1 of one pass (64 pixels) over the step's K channels, then norm2 + relu2 rounded once to fp16 and packed: the B
    // fragments of conv2, cur[2 pass + tile][k-step g >> 1]; with a tape also to HBM
    auto conv1_pass = [&](auto pass_c, int u, int j) {
        constexpr int pass = decltype(pass_c)::value;
        if constexpr (ONEPAIR) {
            pair_iter(std::integral_constant<int, pass>{}, std::true_type{}, 0);
        } else {
            pair_iter(std::integral_constant<int, 0>{}, std::true_type{}, 0);
            pair_iter(std::integral_constant<int, 1>{}, std::false_type{}, 1);
            for (int p = 2; p < npe; p += 2) {
                pair_iter(std::integral_constant<int, 0>{}, std::false_type{}, p);
                pair_iter(std::integral_constant<int, 1>{}, std::false_type{}, p + 1);
            }
        }
        f32x4 osc[4], osh[4];
        const unsigned ot = lb + KS_OT + (32 * wave + 4 * h) * 4;
        static_for<0, 4>([&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            osc[g] = lds_read4<g * 32>(ot);
            osh[g] = lds_read4<512 + g * 32>(ot);
        });
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(osc[0]), "+v"(osc[1]), "+v"(osc[2]), "+v"(osc[3]), "+v"(osh[0]), "+v"(osh[1]), "+v"(osh[2]), "+v"(osh[3]));
#pragma unroll
        for (int beta = 0; beta < 2; ++beta)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned o;
                    asm("v_fma_mixlo_f16 %0, %1, %2, %3\n\t"
                        "v_fma_mixhi_f16 %0, %4, %5, %6\n\t"
                        "v_pk_max_f16 %0, %0, 0"
                        : "=&v"(o)
                        : "v"(c1[beta][4 * g + 2 * q]), "v"(osc[g][2 * q]), "v"(osh[g][2 * q]), "v"(c1[beta][4 * g + 2 * q + 1]),
                          "v"(osc[g][2 * q + 1]), "v"(osh[g][2 * q + 1]));
                    cur[2 * pass + beta][g >> 1][(g & 1) * 2 + q] = o;
                }
        if (Aout) {
            // the TAPE of the gradient path: the activated bottleneck as [4 channel blocks][rows][32] halves, 8 B per lane and
            // channel group straight from the fragments
            const __amdgpu_buffer_rsrc_t rA =
                __builtin_amdgcn_make_buffer_rsrc(Aout + wave * abstride, 0, (unsigned)(abstride * 2), 0x00020000);
            const unsigned R0 = ((unsigned)u * J + j) * 128 + 64 * pass;
#pragma unroll
            for (int beta = 0; beta < 2; ++beta) {
                const unsigned px = S == 32 ? 32 * beta + i : 2 * i + beta;
                const unsigned lo = (R0 + px) * 64u + 8u * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    __builtin_amdgcn_raw_buffer_store_b64(
                        u32x2{cur[2 * pass + beta][g >> 1][(g & 1) * 2], cur[2 * pass + beta][g >> 1][(g & 1) * 2 + 1]}, rA,
                        (int)(lo + 16u * g), 0, 0);
            }
        }
        KS_LAP(4);                                                 // norm2 + relu2 (+ tape)
    };

    // Scheduling hint for a block of conv2: after the first in-row's shifts, one MFMA then up to VPM VALU moves, so that the lane
    // shifts of the next in-row run in the shadow of the MFMAs instead of in bursts in front of them.
#define KS_INTERLEAVE(NM, VPM)                                          \
    do {                                                                \
        __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);             \
        _Pragma("unroll") for (int k_ = 0; k_ < (NM); ++k_) {           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          \
            __builtin_amdgcn_sched_group_barrier(0x002, (VPM), 0);      \
        }                                                               \
    } while (0)
    int u = bid, j = 0, u_prev = -1;
    bool flush = false;
    while (true) {
        const bool first = j == 0;
        if (!flush) conv1_pass(std::integral_constant<int, 0>{}, u, j);
        // ---- conv2, first half: the out row(s) the previous step left open
        const bool border = first || flush;                    // they belong to the previous image; no row below them
        const bool half0 = !(border && u_prev < 0);
        f32x16 a0, a1;
        zero(a0);
        zero(a1);
        if constexpr (S == 64) {
            if (half0) {
                // out row r - 1 (tiles: even x, odd x): in rows r - 2 (prev 0, 1), r - 1 (prev 2, 3), r (cur 0, 1)
                ks_shift<true>(tl, prev[1]);
                ks_shift<false>(tr, prev[0]);
                taps3(a0, 0, tl, prev[0], prev[1]);
                taps3(a1, 0, prev[0], prev[1], tr);
                ks_shift<true>(tl, prev[3]);
                ks_shift<false>(tr, prev[2]);
                taps3(a0, 1, tl, prev[2], prev[3]);
                taps3(a1, 1, prev[2], prev[3], tr);
                KS_INTERLEAVE(24, 3);
                if (!border) {
                    ks_shift<true>(tl, cur[1]);
                    ks_shift<false>(tr, cur[0]);
                    taps3(a0, 2, tl, cur[0], cur[1]);
                    taps3(a1, 2, cur[0], cur[1], tr);
                    KS_INTERLEAVE(12, 3);
                }
            }
        } else {
            // a0: out row 4 j - 1 (the previous image's row 31 at a border) from rows prev 0, prev 1, cur 0;
            // a1: out row 4 j of THIS image from prev 1 (not at its first step), cur 0, cur 1
            if (half0) {
                ks_shift<true>(tl, prev[0]);
                ks_shift<false>(tr, prev[0]);
                taps3(a0, 0, tl, prev[0], tr);
            }
            ks_shift<true>(tl, prev[1]);
            ks_shift<false>(tr, prev[1]);
            if (half0) taps3(a0, 1, tl, prev[1], tr);
            if (!border) taps3(a1, 0, tl, prev[1], tr);
            if (!flush) {
                ks_shift<true>(tl, cur[0]);
                ks_shift<false>(tr, cur[0]);
                if (!border) taps3(a0, 2, tl, cur[0], tr);
                taps3(a1, 1, tl, cur[0], tr);
                ks_shift<true>(tl, cur[1]);
                ks_shift<false>(tr, cur[1]);
                taps3(a1, 2, tl, cur[1], tr);
                KS_INTERLEAVE(18, 4);
            }
        }
        // (the sums of the previous half were read before the pass's pair barriers; the flush has no pass in front of it)
        KS_LAP(5);                                             // conv2
        if (flush) lds_barrier();
        put_partial(a0, 0);
        put_partial(a1, 1);
        lds_barrier();                                         // R1
        KS_LAP(6);                                             // partial sums out + barrier
        if (wave < 2) {
            if constexpr (S == 64) {
                // tile `wave` of out row 2 j - 1 (of the previous image: its row 63)
                const unsigned rowbase = border ? ((unsigned)u_prev * 64 + 63) * 64 : ((unsigned)u * 64 + 2 * j - 1) * 64;
                if (half0) finish(wave, rowbase + 2 * i + wave);
            } else {
                if (wave == 0) {
                    const unsigned rowbase = border ? ((unsigned)u_prev * 32 + 31) * 32 : ((unsigned)u * 32 + 4 * j - 1) * 32;
                    if (half0) finish(0, rowbase + i);
                } else if (!flush) {
                    finish(1, ((unsigned)u * 32 + 4 * j) * 32 + i);
                }
            }
        }
        KS_LAP(7);                                             // finish
        if (flush) break;
        // ---- second pass, second half
        conv1_pass(std::integral_constant<int, 1>{}, u, j);
        zero(a0);
        zero(a1);
        if constexpr (S == 64) {
            // out row r: in rows r - 1 (prev 2, 3; not at an image's first step), r (cur 0, 1), r + 1 (cur 2, 3)
            if (!first) {
                ks_shift<true>(tl, prev[3]);
                ks_shift<false>(tr, prev[2]);
                taps3(a0, 0, tl, prev[2], prev[3]);
                taps3(a1, 0, prev[2], prev[3], tr);
            }
            ks_shift<true>(tl, cur[1]);
            ks_shift<false>(tr, cur[0]);
            taps3(a0, 1, tl, cur[0], cur[1]);
            taps3(a1, 1, cur[0], cur[1], tr);
            ks_shift<true>(tl, cur[3]);
            ks_shift<false>(tr, cur[2]);
            taps3(a0, 2, tl, cur[2], cur[3]);
            taps3(a1, 2, cur[2], cur[3], tr);
            KS_INTERLEAVE(24, 3);
        } else {
            // a0: out row 4 j + 1 from cur 0, 1, 2; a1: out row 4 j + 2 from cur 1, 2, 3
            ks_shift<true>(tl, cur[0]);
            ks_shift<false>(tr, cur[0]);
            taps3(a0, 0, tl, cur[0], tr);
            ks_shift<true>(tl, cur[1]);
            ks_shift<false>(tr, cur[1]);
            taps3(a0, 1, tl, cur[1], tr);
            taps3(a1, 0, tl, cur[1], tr);
            ks_shift<true>(tl, cur[2]);
            ks_shift<false>(tr, cur[2]);
            taps3(a0, 2, tl, cur[2], tr);
            taps3(a1, 1, tl, cur[2], tr);
            ks_shift<true>(tl, cur[3]);
            ks_shift<false>(tr, cur[3]);
            taps3(a1, 2, tl, cur[3], tr);
            KS_INTERLEAVE(36, 3);
        }
        // (the first half's sums were read before the second pass's pair barriers)
        KS_LAP(5);
        put_partial(a0, 0);
        put_partial(a1, 1);
        lds_barrier();                                         // R2
        KS_LAP(6);
        if (wave >= 2) {
            if constexpr (S == 64)
                finish(wave - 2, ((unsigned)u * 64 + 2 * j) * 64 + 2 * i + (wave - 2));
            else
                finish(wave - 2, ((unsigned)u * 32 + 4 * j + 1 + (wave - 2)) * 32 + i);
        }
        KS_LAP(7);
        // ---- the rows the next step still needs
        if constexpr (S == 64) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s) prev[r][s] = cur[r][s];
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s) prev[r][s] = cur[2 + r][s];
        }
        if (++j == J) {
            j = 0;
            u_prev = u;
            u += G;
            if (u >= n_img) flush = true;
        }
    }
    KS_OUT();
    // the loads requested past the end of the stream (out of range: zeros) land before the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef KS_INTERLEAVE

}  // namespace

#ifdef GNX_KS_STAMP
static unsigned long long* g_ks_stamps = nullptr;
#endif
// Launcher of the k-split form (called by dense_layer_launch in dense_layer_f16.hip; argument checks are done there).
int gnx_dense_layer_f16_ks_launch(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                                  const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* A16,
                                  long a_rows_total, int cus, hipStream_t stream) {
    if ((S != 64 && S != 32) || K > 512 || n_img <= 0 || rows_total * 64 >= (1L << 32) || (A16 && a_rows_total * 64 >= (1L << 32)))
        return GNX_ERR_UNSUPPORTED;
    const int grid = (int)(n_img < 2L * cus ? n_img : 2L * cus);
    _Float16* X = reinterpret_cast<_Float16*>(X16);
    const _Float16* w1 = reinterpret_cast<const _Float16*>(w1p);
    const _Float16* w2 = reinterpret_cast<const _Float16*>(w2p);
    _Float16* A = reinterpret_cast<_Float16*>(A16);
#ifdef GNX_KS_STAMP
#define KS_STAMP_ARG , g_ks_stamps
#else
#define KS_STAMP_ARG
#endif
#define KS_GO(SS, OP)                                                                                                      \
    dense_layer_f16_ks_kernel<SS, OP><<<grid, 256, 0, stream>>>(X, rows_total * 32, (int)n_img, K, w1, w2, scale1, shift1,      \
                                                                scale2, shift2, A, a_rows_total * 32 KS_STAMP_ARG)
    if (S == 64) {
        if (K == 64) KS_GO(64, true); else KS_GO(64, false);
    } else {
        if (K == 64) KS_GO(32, true); else KS_GO(32, false);
    }
#undef KS_GO
    return gnx_launch_status();
}
#ifdef GNX_KS_STAMP
GNX_EXPORT void gnx_ks_set_stamps(void* buf) { g_ks_stamps = reinterpret_cast<unsigned long long*>(buf); }
GNX_EXPORT int gnx_ks_launch(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                             const float* scale1, const float* shift1, const float* scale2, const float* shift2, int cus,
                             hipStream_t stream) {
    return gnx_dense_layer_f16_ks_launch(X16, rows_total, n_img, S, K, w1p, w2p, scale1, shift1, scale2, shift2, nullptr, 0, cus,
                                         stream);
}
#endif
