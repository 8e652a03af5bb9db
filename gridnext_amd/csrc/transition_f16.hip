// A DenseNet transition of the fp16-MFMA path (BASELINE config 5) as ONE kernel on channel-blocked fp16 block buffers:
//   norm -> relu -> conv (1x1, K -> N = K / 2) -> avgpool 2x2        (/root/reference/gridnext/densenet.py:47-53)
// evaluated pool-first (the mean of a 2 x 2 window commutes with the 1 x 1 convolution): the pooled, activated operand exists
// in the LDS only.  The two-kernel form (gnx_bnrelu_avgpool2_h16_cb + gnx_conv1x1_bnrelu_h16_cb) writes it to HBM and reads it
// back: per array at 256 px 18.3 + 4.6 + 4.6 + 2.3 GB where this kernel moves 18.3 + 2.3.
//
// Same skeleton as dense_layer_f16_kernel (dense_layer_f16.hip), without its second half: a step = 128 pooled pixels, one
// persistent workgroup per CU, 8 waves;
//   waves 4-7 (feeders): per stage of 32 channels, feeder f loads the four source pixels of its 32 pooled pixels (8 x 16 B per
//     lane: lane = pooled pixel (lane & 15) of a 16-pixel piece, 16-B column lane >> 4), applies norm + relu in fp32 on each
//     (the operations, and their order, of bnrelu_avgpool2_h16_kernel: the pooled operand is bit-identical), averages, rounds
//     to fp16 and writes the LDS slot; its stages of all the workgroup's steps are one stream through a 2-deep register ring (64 KB per CU in flight);
//   waves 0-3 (consumers): [32 NT output channels per wave] x [128 pooled pixels] accumulators, W fragments straight from a
//     pre-packed fragment image in global memory (L2), two stages per workgroup barrier; the raw sums go to the next block's
//     buffer as fp16 (no activation: the next block's norm1 is applied by its consumers).
// grid.y selects 128 NT of the N output channels (N = 512 at NT = 2: two passes over the input, 1.3 GB of 18.3 per array).
#include "fwd_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t(), 0, 0, 0)) u32x4;

constexpr int TR_SLOT = 8192;                  // 128 px x 32 channels: byte(px, chunk) = (px >> 4) * 1024 + chunk * 256 + (px & 15) * 16
constexpr int TR_AR = 0;                       // four slots: stage s of a step in slot s & 3
constexpr int TR_CT = TR_AR + 4 * TR_SLOT;     // norm constants: [stage][16-B column 4][scale 8 | shift 8] floats (K <= 1024)
constexpr int TR_LDS = TR_CT + 8192;

template <int NT>
__global__ __launch_bounds__(512) void transition_f16_kernel(const _Float16* __restrict__ X, long xbs, long rows_in,
                                                             _Float16* __restrict__ Y, long ybs, int n_steps, int S, int K,
                                                             const _Float16* __restrict__ wp, const float* __restrict__ sc,
                                                             const float* __restrict__ sh, _Float16* __restrict__ Pout, long ldp) {
    __shared__ __attribute__((aligned(16))) char lds[TR_LDS];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int nst = K >> 5, np = (nst + 1) >> 1;
    const unsigned lb = lds_addr(lds);
    const int N = bid < n_steps ? (n_steps - bid + G - 1) / G : 0;     // this workgroup's steps: bid, bid + G, ...

    if (wave >= 4) {
        // ================================================================= feeders
        const int fw = wave - 4;
        for (int k = t - 256; k < K; k += 256) {                // the norm table: [stage][column][scale 8 | shift 8]
            float* d = reinterpret_cast<float*>(lds + TR_CT) + (k >> 3) * 16 + (k & 7);
            d[0] = sc[k];
            d[8] = sh[k];
        }
        const int So = S >> 1, lgSo = 31 - __builtin_clz(So), col = lane >> 4;
        // the lane's two pooled pixels (pieces) of step st -> byte offsets of their top-left source pixels in a channel block
        auto src_off = [&](int st, unsigned (&vo)[2]) {
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                const long q = (long)st * 128 + 32 * fw + 16 * pc + (lane & 15);
                const long img = q >> (2 * lgSo);
                const int rem = (int)(q & ((1 << (2 * lgSo)) - 1)), oy = rem >> lgSo, ox = rem & (So - 1);
                vo[pc] = (unsigned)(((img * S + 2 * oy) * S + 2 * ox) * 64 + col * 16);
            }
        };
        constexpr int RD = 2;                                  // (two stages = 64 KB per CU in flight)
        u32x4 rq[RD][8];                                       // [ring][piece 2 x (dy, dx) 4]
        int ln = 0, ls = 0;                                    // the next load: step (of this workgroup), stage
        unsigned lvo[2];
        src_off(bid, lvo);
        const unsigned nrec = (unsigned)(rows_in * 64);
        // (inline asm loads, waited by hand: see dense_layer_f16.hip - every load is waited for before its registers die;
        // past the end of the stream the offsets leave the resource: zeros, no traffic)
        auto load_next = [&](u32x4(&dst)[8]) {
            const bool in = ln < N;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(X) + (in ? ls : 0) * xbs, 0, nrec, 0x00020000);
            const unsigned o0 = in ? lvo[0] : 0xff000000u, o1 = in ? lvo[1] : 0xff000000u;
            const unsigned dyo = (unsigned)S * 64;
            const unsigned a0 = o0 + dyo, a1 = o1 + dyo;
            asm volatile("buffer_load_dwordx4 %0, %4, %6, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen offset:64\n\t"
                         "buffer_load_dwordx4 %2, %5, %6, 0 offen\n\tbuffer_load_dwordx4 %3, %5, %6, 0 offen offset:64"
                         : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(o0), "v"(a0), "s"(rs) : "memory");
            asm volatile("buffer_load_dwordx4 %0, %4, %6, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen offset:64\n\t"
                         "buffer_load_dwordx4 %2, %5, %6, 0 offen\n\tbuffer_load_dwordx4 %3, %5, %6, 0 offen offset:64"
                         : "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]), "=&v"(dst[7]) : "v"(o1), "v"(a1), "s"(rs) : "memory");
            if (++ls == nst) {
                ls = 0;
                ++ln;
                src_off(bid + (ln < N ? ln : 0) * G, lvo);
            }
        };
        auto landed = [&](u32x4(&v)[8]) {                      // the oldest of the RD stages in flight
            asm volatile("s_waitcnt vmcnt(%8)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                         : "n"(8 * RD - 8) : "memory");
        };
        const int Gt = N * nst;
        static_for<0, RD>([&](auto k_c) { load_next(rq[decltype(k_c)::value]); });
        lds_barrier();                                         // B_init: the table is in the LDS
        const unsigned ctb = lb + TR_CT + 64 * col;
        f32x4 ks0, ks1, kb0, kb1;                              // norm constants of the stage applied next (8 channels)
        auto request_k = [&](int ss) {
            const unsigned a = ctb + ss * 256;
            ks0 = lds_read4<0>(a);
            ks1 = lds_read4<16>(a);
            kb0 = lds_read4<32>(a);
            kb1 = lds_read4<48>(a);
        };
        request_k(0);
        // relu(norm(x)) of the four source pixels, summed in the order (0,0) (0,1) (1,0) (1,1), times 1/4, rounded to fp16:
        // exactly bnrelu_avgpool2_h16_kernel's arithmetic
        auto pooled = [&](const u32x4& p00, const u32x4& p01, const u32x4& p10, const u32x4& p11) {
            const half8 v0 = __builtin_bit_cast(half8, p00), v1 = __builtin_bit_cast(half8, p01),
                        v2 = __builtin_bit_cast(half8, p10), v3 = __builtin_bit_cast(half8, p11);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float s_ = j < 4 ? ks0[j & 3] : ks1[j & 3], b_ = j < 4 ? kb0[j & 3] : kb1[j & 3];
                float a = act1((float)v0[j], s_, b_);
                a += act1((float)v1[j], s_, b_);
                a += act1((float)v2[j], s_, b_);
                a += act1((float)v3[j], s_, b_);
                o[j] = (_Float16)(0.25f * a);
            }
            return __builtin_bit_cast(u32x4, o);
        };
        int sa = 0, bdone = 0;                                 // in-step index of the stage applied next; pair barriers passed in its step
        // the TAPE of the gradient path (gnx_transition_f16_tape): the pooled activated operand also goes to HBM as a row-major
        // [pooled pixel][K] matrix - the operand of the 1x1 weight gradient - 16 B per lane and piece (the first output pass
        // writes it; the stores only make the hand-counted load waits conservative)
        const bool tape = Pout != nullptr && blockIdx.y == 0;
        long prow = (long)bid * 128 + 32 * fw + (lane & 15);   // the lane's first pooled pixel of the step being applied
        auto gapply = [&](auto ph_c, int g) {
            constexpr int P = decltype(ph_c)::value;
            if (sa == 0 && g > 0) {                            // a new step: what is left of the previous one's barriers, then E
                for (; bdone < np; ++bdone) lds_barrier();
                lds_barrier();                                 // E: the consumers are done with every slot
                bdone = 0;
            }
            for (; bdone < (sa >> 1); ++bdone) lds_barrier();  // ... B_(pair - 1): the pair's slots have been read
            landed(rq[P]);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ks0), "+v"(ks1), "+v"(kb0), "+v"(kb1));
            char* d = lds + TR_AR + (sa & 3) * TR_SLOT + 2 * fw * 1024 + lane * 16;
            const u32x4 p0 = pooled(rq[P][0], rq[P][1], rq[P][2], rq[P][3]);
            const u32x4 p1 = pooled(rq[P][4], rq[P][5], rq[P][6], rq[P][7]);
            *reinterpret_cast<u32x4*>(d) = p0;
            *reinterpret_cast<u32x4*>(d + 1024) = p1;
            if (tape) {
                _Float16* q = Pout + prow * ldp + sa * 32 + col * 8;
                *reinterpret_cast<u32x4*>(q) = p0;
                *reinterpret_cast<u32x4*>(q + 16 * ldp) = p1;
            }
            load_next(rq[P]);
            if (sa + 1 == nst) prow += (long)G * 128;
            sa = sa + 1 == nst ? 0 : sa + 1;
            request_k(sa);
        };
#define TR_GS(k) gapply(std::integral_constant<int, k>{}, g + k); if (g + k + 1 >= Gt) break;
        if (Gt > 0) {
            for (int g = 0;; g += RD) { TR_GS(0) TR_GS(1) }
            for (; bdone < np; ++bdone) lds_barrier();         // the last step's remaining pair barriers, and E
            lds_barrier();
        }
#undef TR_GS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stages requested past the end of the stream
        __builtin_amdgcn_sched_barrier(0);
        return;
    }

    // ===================================================================== consumers
    const int KS = K >> 4;
    const int cb0 = (blockIdx.y * 4 + wave) * NT;              // this wave's first 32-channel block of the output
    const __amdgpu_buffer_rsrc_t rW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(wp), 0, (unsigned)((long)gridDim.y * 128 * NT * K * 2), 0x00020000);
    u32x4 fr[NT == 1 ? 4 : 2][NT][2];                          // W fragment ring [stage & 3 (NT = 1) or & 1][channel block][k-step]
    const int lane16 = lane * 16;
    auto load_w = [&](u32x4(&dst)[NT][2], int ws) {            // (a stage past the last one: the last one's again - in range, unused)
        const int wsc = ws < nst ? ws : nst - 1;
#pragma unroll
        for (int tq = 0; tq < NT; ++tq) {
            const int so = ((cb0 + tq) * KS + 2 * wsc) * 1024;  // scalar offset: no per-stage address registers
            dst[tq][0] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16, so, 0);
            dst[tq][1] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16 + 1024, so, 0);
        }
    };
    load_w(fr[0], 0);
    if constexpr (NT == 1) load_w(fr[1], 1);
    lds_barrier();                                             // B_init
    f32x16 c1[NT][4];                                          // [channel block][pixel block]
    const unsigned laneA = lb + TR_AR + (i >> 4) * 1024 + h * 256 + (i & 15) * 16;
    auto pair = [&](auto ph_c, int s, auto first_c) {          // stages s, s + 1 (one barrier); first: srcC = 0
        constexpr int P = decltype(ph_c)::value;               // s & 3: 0 or 2
        constexpr bool FIRST = decltype(first_c)::value;
        lds_barrier();                                         // B_(s / 2)
        if constexpr (NT == 1) {
            load_w(fr[(P + 2) & 3], s + 2);
            load_w(fr[(P + 3) & 3], s + 3);
        }
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (NT == 1) {
            // the second stage's fragments are requested while the first stage multiplies
            f32x4 av[16];
            static_for<0, 8>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;        // n = 4 ks + rb
                av[n] = lds_read4<P * TR_SLOT + (n & 3) * 2048 + (n >> 2) * 512>(laneA);
            });
            static_for<0, 8>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                av[8 + n] = lds_read4<(P + 1) * TR_SLOT + (n & 3) * 2048 + (n >> 2) * 512>(laneA);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(av[n]));
                c1[0][n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fr[P][0][n >> 2]),
                                                                      __builtin_bit_cast(half8, av[n]),
                                                                      FIRST && n < 4 ? zero16 : c1[0][n & 3], 0, 0, 0);
            });
            if (s + 1 < nst)
                static_for<0, 8>([&](auto n_c) {
                    constexpr int n = decltype(n_c)::value;
                    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[8 + n]) : "n"(7 - n));
                    c1[0][n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fr[P + 1][0][n >> 2]),
                                                                          __builtin_bit_cast(half8, av[8 + n]), c1[0][n & 3], 0, 0, 0);
                });
            else                                               // an odd step's last pair: the reads still own their registers
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[8]), "+v"(av[9]), "+v"(av[10]), "+v"(av[11]), "+v"(av[12]),
                                                      "+v"(av[13]), "+v"(av[14]), "+v"(av[15]));
        } else {
            // 256 accumulator + ring registers leave room for one stage's fragments: stage by stage (16 NT MFMAs per stage
            // stand behind each first-fragment latency)
            // (and for a two-deep W ring, entry = stage & 1, the next stage's fragments requested at a stage's start: this kernel
            // waits for its 32 KB of input per stage, the consumers have the time)
            static_for<0, 2>([&](auto st_c) {
                constexpr int st = decltype(st_c)::value;
                if (st == 0 || s + 1 < nst) {
                    load_w(fr[(st + 1) & 1], s + st + 1);
                    f32x4 av[8];
                    static_for<0, 8>([&](auto n_c) {
                        constexpr int n = decltype(n_c)::value;
                        av[n] = lds_read4<(P + st) * TR_SLOT + (n & 3) * 2048 + (n >> 2) * 512>(laneA);
                    });
                    static_for<0, 8>([&](auto n_c) {
                        constexpr int n = decltype(n_c)::value;
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[n]) : "n"(7 - n));
#pragma unroll
                        for (int tq = 0; tq < NT; ++tq)
                            c1[tq][n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                __builtin_bit_cast(half8, fr[st][tq][n >> 2]), __builtin_bit_cast(half8, av[n]),
                                FIRST && st == 0 && n < 4 ? zero16 : c1[tq][n & 3], 0, 0, 0);
                    });
                }
            });
        }
    };
    // no loop: the (up to 16) pairs of a step one after the other with an exit after each - the compiler then counts the W
    // loads in flight exactly (around a loop it drains them at every trip)
    auto pairs_from = [&](auto k_c) {
        auto impl = [&](auto& self, auto kk_c) -> void {
            constexpr int k = decltype(kk_c)::value;
            pair(std::integral_constant<int, (2 * k) & 3>{}, 2 * k, std::integral_constant<bool, k == 0>{});
            if constexpr (k + 1 < 16)
                if (2 * k + 2 < nst) self(self, std::integral_constant<int, k + 1>{});
        };
        impl(impl, k_c);
    };
    for (int n = 0; n < N; ++n) {
        const long R0 = ((long)bid + (long)n * G) * 128;       // the step's first pooled row
        pairs_from(std::integral_constant<int, 0>{});
        load_w(fr[0], 0);                                      // the next step's first stages (the same weights)
        if constexpr (NT == 1) load_w(fr[1], 1);
        lds_barrier();                                         // E: every consumer is done with the step's slots
        // the raw sums as fp16 into the next block's buffer [c / 32][rows][32]: lane = pixel, 4 x 4 consecutive channels
#pragma unroll
        for (int tq = 0; tq < NT; ++tq)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                _Float16* p = Y + (long)(cb0 + tq) * ybs + (R0 + 32 * rb + i) * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (_Float16)c1[tq][rb][4 * g + q];
                    *reinterpret_cast<half4*>(p + 8 * g) = o;
                }
            }
    }
}

// W [N][K] fp32 (torch layout of the 1x1 conv) -> fragment order halves:
// ((cb * K/16 + ks) * 64 + lane) * 8 + q = W[32 cb + (lane & 31)][16 ks + 8 (lane >> 5) + q]
__global__ void tr_pack_w_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int N, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * K) return;
    const int q = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long f = idx >> 9;
    const int KS = K >> 4, cb = (int)(f / KS), ks = (int)(f - (long)cb * KS);
    out[idx] = (_Float16)w[(long)(32 * cb + (lane & 31)) * K + 16 * ks + 8 * (lane >> 5) + q];
}

}  // namespace

// conv.weight [N][K] (fp32, torch layout) -> wp (N * K halves): the fragment-ordered fp16 operand of gnx_transition_f16
// (rounded once; cache it per weight version).  32 | K, 128 | N.
GNX_EXPORT int gnx_transition_f16_pack(const float* w, void* wp, int N, int K, hipStream_t stream) {
    if (!w || !wp || N <= 0 || K <= 0) return GNX_ERR_BAD_ARG;
    if (K % 32 != 0 || N % 128 != 0) return GNX_ERR_UNSUPPORTED;
    tr_pack_w_kernel<<<gnx_cdiv((long)N * K, 256), 256, 0, stream>>>(w, reinterpret_cast<_Float16*>(wp), N, K);
    return gnx_launch_status();
}

// The transition on CHANNEL-BLOCKED fp16 buffers: X16 [K / 32][rows_in][32] (rows = n_img * S * S pixels, the first of rows_in)
// -> Y16 [.. / 32][rows_out][32], channel blocks [0, N / 32) of it, rows n_img * (S / 2)^2.  S in {8, 16, 32, 64}; 32 | K,
// 64 <= K <= 1024; 128 | N <= 512; 128 | n_img * (S / 2)^2; scale / shift: the folded running-statistics BatchNorm (K).
static int transition_launch(const void* X16, long rows_in, long n_img, int S, int K, int N, const void* wp, const float* scale,
                             const float* shift, void* Y16, long rows_out, void* P16, long ldp, hipStream_t stream) {
    if (!X16 || !Y16 || !wp || !scale || !shift || n_img < 0 || K <= 0 || N <= 0 || S <= 0 || rows_in < n_img * (long)S * S ||
        rows_out < n_img * (long)(S / 2) * (S / 2) || (P16 && ldp < K))
        return GNX_ERR_BAD_ARG;
    if (P16 && (!al16(P16) || ldp % 8 != 0)) return GNX_ERR_UNSUPPORTED;
    const long mout = n_img * (long)(S / 2) * (S / 2);
    if ((S != 8 && S != 16 && S != 32 && S != 64) || K % 32 != 0 || K < 64 || K > 1024 || N % 128 != 0 || N > 512 ||
        mout % 128 != 0 || !al16(X16) || !al16(Y16) || !al16(wp) || rows_in * 64 >= (1L << 32) - (1L << 25) ||
        mout / 128 >= (1L << 31))
        return GNX_ERR_UNSUPPORTED;
    if (mout == 0) return GNX_OK;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GNX_ERR_LAUNCH;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long steps = mout / 128;
    const _Float16* X = reinterpret_cast<const _Float16*>(X16);
    _Float16* Y = reinterpret_cast<_Float16*>(Y16);
    const _Float16* w = reinterpret_cast<const _Float16*>(wp);
    _Float16* Pt = reinterpret_cast<_Float16*>(P16);
    if (N % 256 == 0) {
        const int gx = (int)(steps < cus ? steps : cus);       // (N = 512: the two channel halves share a CU's time)
        transition_f16_kernel<2><<<dim3(gx, N / 256), 512, 0, stream>>>(X, rows_in * 32, rows_in, Y, rows_out * 32, (int)steps, S, K,
                                                                         w, scale, shift, Pt, ldp);
    } else {
        const int gx = (int)(steps < cus ? steps : cus);
        transition_f16_kernel<1><<<dim3(gx, N / 128), 512, 0, stream>>>(X, rows_in * 32, rows_in, Y, rows_out * 32, (int)steps, S, K, w,
                                                                  scale, shift, Pt, ldp);
    }
    return gnx_launch_status();
}
GNX_EXPORT int gnx_transition_f16(const void* X16, long rows_in, long n_img, int S, int K, int N, const void* wp,
                                  const float* scale, const float* shift, void* Y16, long rows_out, hipStream_t stream) {
    return transition_launch(X16, rows_in, n_img, S, K, N, wp, scale, shift, Y16, rows_out, nullptr, 0, stream);
}
// The same transition as the TAPED forward of the fp16 gradient path: additionally stores the pooled activated operand
// avgpool2(relu(norm(x))) - the tile the feeders put into the LDS anyway, bit for bit gnx_bnrelu_avgpool2_h16's output - as a
// row-major fp16 matrix P16 [n_img * (S / 2)^2][ldp] (columns [0, K)), the operand of the 1x1 weight gradient.  Everything else
// is gnx_transition_f16 bit for bit.
GNX_EXPORT int gnx_transition_f16_tape(const void* X16, long rows_in, long n_img, int S, int K, int N, const void* wp,
                                       const float* scale, const float* shift, void* Y16, long rows_out, void* P16, long ldp,
                                       hipStream_t stream) {
    if (!P16) return GNX_ERR_BAD_ARG;
    return transition_launch(X16, rows_in, n_img, S, K, N, wp, scale, shift, Y16, rows_out, P16, ldp, stream);
}
