// One DenseNet dense layer of the fp16-MFMA path (BASELINE config 5) as ONE kernel on fp16 block buffers:
//   cat -> norm1 -> relu1 -> conv1 (1x1, K -> 128) -> norm2 -> relu2 -> conv2 (3x3 pad 1, 128 -> 32) -> 32 new columns
// (/root/reference/gridnext/densenet.py:35-44 is one `forward`).  The unfused pair (gnx_conv1x1_bnrelu_h16 +
// gnx_conv3x3_f16_dma_h) writes the 128-channel bottleneck to HBM and reads it back: 35.0 + 14.1 MB per spot at 256 px where
// this kernel moves 26.6 MB - the K input columns in, 32 columns out, nothing else.
//
// Work unit: an image (S x S map of one spot) for S >= 16, swept top to bottom in steps of 128 pixels; a tile of 128
// pixels (whole images) for S <= 8.  One persistent workgroup per CU, 8 waves.  Two kernels:
// * dense_layer_f16_kernel<S> (S <= 32, K up to 992 - HBM-side):
//   waves 4-7 (feeders): the [128 px][K] input strip of every step, in stages of 32 channels, global -> registers ->
//     norm1 + relu1 (fp32 fma on the fp16 value, one rounding: v_fma_mix) -> LDS slot; the stages of ALL the workgroup's
//     steps are one stream through an 8-deep register ring (seven stages = 56 KB per CU in flight);
//   waves 0-3 (consumers): conv1 as [128 ch] x [128 px] per step, wave w owning output channels 32w..32w+31 for all 128
//     pixels (its W1 fragments come straight from global memory in a pre-packed fragment order: 1 KB coalesced per
//     fragment, no LDS), two stages per workgroup barrier; norm2 + relu2 on the accumulators; the activated bottleneck
//     tile [128 px][128 ch] goes to LDS as fp16 - and never to HBM; conv2 reads it back with per-tap shifted fragment
//     addresses (W2, 72 KB in fragment order, is LDS-resident for the workgroup's lifetime).
// * dense_layer_f16_s64_kernel (64 x 64 maps, K <= 224 - few conv1 stages in front of 288 conv2 MFMAs per step): conv1 of
//   step n (waves 4-7, operands straight from global memory) overlaps conv2 of step n - 1 (waves 0-3) through a
//   double-buffered bottleneck tile; see there.
// conv2 runs in SCATTER form so that the LDS holds only the step's own bottleneck rows (32 KB) instead of a ring with halo
// rows: a step's bottleneck rows are multiplied into every output row they touch - the previous step's last row (its
// dy = +1 taps), the step's own rows, the next step's first row (dy = -1) - and the accumulators of output rows that still
// wait for a later step stay in registers.  Which wave finishes and which wave carries rotates with the step, so that an
// accumulator never changes owner; every output sums its taps in the order dy = -1, 0, +1 (dx inside), whatever the
// chunking: deterministic.  Image borders and rows outside the step's tile are lanes whose fragment address points at a
// zero region: no select on data, nothing in the MFMA stream depends on them.
// MFMA orientation: D[channel][pixel] (A operand = weights, B operand = activations), so a lane owns a pixel and its 16
// accumulator registers are 4 x 4 consecutive channels: the bottleneck goes to LDS with ds_write_b64, the output to HBM
// with 8-B stores.
#include "fwd_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t(), 0, 0, 0)) u32x4;

constexpr int DL_W2 = 0;                       // [tap 9][kstep 8][lane 64][8 halves]
constexpr int DL_W2_BYTES = 9 * 8 * 1024;
constexpr int DL_BT = DL_W2 + DL_W2_BYTES;     // bottleneck tile: byte(px, chunk) = (px >> 4) * 4096 + chunk * 256 + (px & 15) * 16
constexpr int DL_BT_BYTES = 128 * 256;
constexpr int DL_NS = 4;                       // activated input stages in the LDS (stage s of a step: slot s & 3)
constexpr int DL_SLOT = 8192;                  // 128 px x 32 channels: byte(px, chunk) = (px >> 4) * 1024 + chunk * 256 + (px & 15) * 16
constexpr int DL_AR = DL_BT + DL_BT_BYTES;
constexpr int DL_Z = DL_AR + DL_NS * DL_SLOT;  // 4 KB of zeros (masked fragment lanes; immediates reach 7 * 512 + 256 + 16)
constexpr int DL_OT = DL_Z + 4096;             // norm2: scale[128], shift[128]
constexpr int DL_CT = DL_OT + 1024;            // norm1 of the whole layer: [stage][16-B column 4][scale 8 | shift 8] floats (K <= 1024)
constexpr int DL_LDS = DL_CT + 8192;
static_assert(DL_LDS <= 160 * 1024, "LDS");

// Diagnostic build only (tools/ubench/dl_stamps.py compiles this file with -DGNX_DL_STAMP into its own library): per
// workgroup, wave 0 and wave 4 sum the shader cycles they spend in each segment of a step and leave them in a buffer of
// their own.  In the product build the macros are empty: no stamp executes.
#ifdef GNX_DL_STAMP
#define GNX_DL_STAMP_PARAM , unsigned long long* __restrict__ stamps, int abl
#ifdef GNX_DL_NOABL                             // timers only: the ablation tests are run-time branches inside the MFMA streams
#define DL_ABL(bit) false
#else
#define DL_ABL(bit) (abl & (bit))
#endif
#define DL_T0() unsigned long long dl_t = __builtin_readcyclecounter(), dl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DL_LAP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); dl_acc[k] += n_ - dl_t; dl_t = n_; } while (0)
#define DL_OUT(base) do { if (stamps && lane == 0) for (int q_ = 0; q_ < 8; ++q_) stamps[(long)blockIdx.x * 24 + (base) + q_] = dl_acc[q_]; } while (0)
#else
#define GNX_DL_STAMP_PARAM
#define DL_ABL(bit) false
#define DL_T0() do {} while (0)
#define DL_LAP(k) do {} while (0)
#define DL_OUT(base) do {} while (0)
#endif

template <int S>
__global__ __launch_bounds__(512) void dense_layer_f16_kernel(_Float16* __restrict__ X, long bstride, int n_units, int K,
                                                              const _Float16* __restrict__ w1p,
                                                              const _Float16* __restrict__ w2p,
                                                              const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                              const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                              _Float16* __restrict__ Aout, long abstride
                                                              GNX_DL_STAMP_PARAM) {
    constexpr int J = S >= 16 ? S * S / 128 : 1;           // steps per unit
    constexpr int LOG2S = S == 32 ? 5 : S == 16 ? 4 : S == 8 ? 3 : 2;
    static_assert(S <= 32, "64 x 64 maps: dense_layer_f16_s64_kernel");
    __shared__ __attribute__((aligned(16))) char lds[DL_LDS];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int nst = K >> 5;                                // stages per step
    const unsigned lb = lds_addr(lds);
    DL_T0();

    // norm1 + relu1 on raw input pieces (16 px x 32 channels of fp16 each; a lane holds 16 B = 8 channels of one pixel): fp32
    // fma on the fp16 value, rounded once to fp16 (v_fma_mix), relu packed.
    struct ActRegs { f32x4 s0, s1, b0, b1; };
    auto act2 = [](unsigned x, float sa, float ba, float sb, float bb) {                  // two halves of one register
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

    // (Tried and dropped, measured: conv2 of step n as jobs of 24 MFMAs on the FEEDER waves, one per applied pair of step n + 1,
    // the consumers doing conv1 + epilogue only (extra barrier F before the epilogue): 8-22 % SLOWER - a wave's share is 3-4
    // dependent chains of ~1500 cycles whatever wave runs them, and behind the feeders' own chain they lengthen the pair
    // period; the step stays bound by one group's serial work.)
    // (Tried and dropped, measured: four slot pairs - two of them the bottleneck tile's own bytes, dead outside epilogue .. conv2 -
    // with the feeders up to three pairs ahead of the consumers: no faster (S = 32 K = 480 1.42 -> 1.49 ms).  The consumers do
    // not wait for data: a pair costs them ~900 cycles for 512 of MFMA - barrier, first-fragment latency, W1 issue.)
    // (Tried and dropped, measured: every tap's eight k-steps split between consumer w (0-3) and feeder w (4-7), the feeder's
    // half of a finished block handed over through the LDS between two more barriers: 3-15 % SLOWER - a 12-MFMA stream is
    // half pipeline fill, and the address set-up per tap does not shrink.)
    // (Tried and dropped, measured: conv2 cut into jobs of 24 MFMAs run one per stage behind the NEXT step's conv1 stages, so
    // that stage barriers - and with them the loaders' refills - keep passing during conv2: S = 32 / 16 layers got 12-37 %
    // SLOWER, S = 64 / 8 unchanged.  A stage's period is set by its slowest wave, and the consumer waves are no faster per stage
    // than the DMA stream: added to their stages the jobs add to the step instead of hiding in it.)
    // ---- conv2 of one step, scatter form: the step's bottleneck tile (LDS) into every output block it touches.  At S = 64 ALL
    // eight waves take part (after the step's E barrier the loaders and activators have nothing else to do, and two waves per
    // SIMD hide each other's LDS latency: stamped, four waves alone needed ~58 cycles per MFMA); smaller maps: waves 0-3.
    auto conv2_step = [&](int u, int j, f32x16& a0, f32x16& a1) {
            const long R0 = ((long)u * J + j) * 128;
            auto tap3 = [&](f32x16& acc, int O_rel, int dy) {  // the three dx taps of row offset dy into the block at O_rel
                const int o = O_rel + i;
                const int pin = S >= 16 ? 128 * j + o : (o & (S * S - 1));
                const int y = pin >> LOG2S, x = pin & (S - 1);
                unsigned aA[3];
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int r = o + dy * S + dx;
                    const bool ok = (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S && (unsigned)r < 128u;
                    aA[dx + 1] = ok ? lb + DL_BT + (r >> 4) * 4096 + (r & 15) * 16 + h * 256 : lb + DL_Z + h * 256;
                }
                const unsigned aW = lb + DL_W2 + (dy + 1) * 3 * 8192 + lane * 16;
                // 24 (dx, k-step) products; the fragment pairs of the next D are in flight while one multiplies (a ring of
                // D + 2 register pairs: a pair is overwritten two MFMAs after the MFMA that read it)
                constexpr int D = 7, NSL = D + 2, NE = 24;
                f32x4 ra[NSL] = {}, rw[NSL] = {};
                auto request = [&](auto e_c) {
                    constexpr int e = decltype(e_c)::value, dxi = e / 8, ks = e % 8;
                    if (!DL_ABL(256)) ra[e % NSL] = lds_read4<ks * 512>(aA[dxi]);
                    if (!DL_ABL(128)) rw[e % NSL] = lds_read4<dxi * 8192 + ks * 1024>(aW);
                };
                static_for<0, D>(request);
                static_for<0, NE>([&](auto e_c) {
                    constexpr int e = decltype(e_c)::value;
                    if constexpr (e + D < NE) request(std::integral_constant<int, e + D>{});
                    constexpr int younger = 2 * (e + D < NE ? D : NE - 1 - e);
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ra[e % NSL]), "+v"(rw[e % NSL]) : "n"(younger));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[e % NSL]),
                                                                 __builtin_bit_cast(half8, ra[e % NSL]), acc, 0, 0, 0);
                });
            };
            auto store = [&](const f32x16& acc, int O_rel) {
                _Float16* p = X + (K >> 5) * bstride + (R0 + O_rel + i) * 32 + 4 * h;   // the layer's new block [K / 32]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (_Float16)acc[4 * g + q];
                    *reinterpret_cast<half4*>(p + 8 * g) = o;
                }
            };
            auto zero = [&](f32x16& acc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            };
            // Which blocks this wave works on in this step: accumulator a0 on the block at O0 with row offsets dy = lo0..hi0,
            // a1 on the block at O1 with lo1..hi1 (an empty range: nothing); zN: the accumulator starts from zero in this
            // step (else it is carried in from the previous one); sN: it is complete after this step and goes to HBM.
            int O0, lo0, hi0, O1 = 0, lo1 = 0, hi1 = -1;
            bool z0, z1 = false, s0, s1 = false;
            if constexpr (S >= 16) {
                // S = 32: a block = one image row, a step = 4 rows; S = 16: a block = two rows, a step = 8 rows - there the
                // head's own block still lacks the dy = -1 taps of its second row and the tail's block the dy = +1 taps of
                // its first (their other lanes are outside the tile: zeros).
                if (wave >= 4) return;
                const int role = (wave + j) & 3;
                if (role == 0) {
                    O0 = -32; lo0 = 1; hi0 = j == 0 ? 0 : 1; z0 = false; s0 = j != 0;
                    O1 = 0; lo1 = S == 16 ? -1 : 0; hi1 = 1; z1 = j == 0; s1 = true;
                } else if (role == 3) {
                    O0 = 96; lo0 = -1; hi0 = S == 16 ? 1 : 0; z0 = true; s0 = j == J - 1;
                    O1 = 128; lo1 = -1; hi1 = j == J - 1 ? -2 : -1; z1 = true;
                } else {
                    O0 = 32 * role; lo0 = -1; hi0 = 1; z0 = true; s0 = true;
                }
            } else {
                if (wave >= 4) return;
                O0 = 32 * wave; lo0 = -1; hi0 = 1; z0 = true; s0 = true;
            }
            if (z0) zero(a0);
            if (z1) zero(a1);
            if (DL_ABL(8)) { hi0 = lo0 - 1; hi1 = lo1 - 1; }
#pragma unroll 1
            for (int dy = lo0; dy <= hi0; ++dy) tap3(a0, O0, dy);
            DL_LAP(4);
            if (s0 && !DL_ABL(64)) store(a0, O0);
            DL_LAP(5);
#pragma unroll 1
            for (int dy = lo1; dy <= hi1; ++dy) tap3(a1, O1, dy);
            DL_LAP(4);
            if (s1 && !DL_ABL(64)) store(a1, O1);
            DL_LAP(5);
    };

    if (wave >= 4) {
        // ================================================================= feeders (waves 4-7)
        // The input strip of a step goes global -> REGISTERS -> (norm1 + relu1) -> LDS.  (The first version moved it by LDS-DMA
        // into a ring of raw stages that was then activated in place: 8 KB per 0.52 us and CU whatever the ring depth or the
        // number of issuing waves - the LDS-DMA path itself delivers ~25 GB/s per CU.)  Feeder f takes pixels 32 f .. 32 f + 31
        // of every stage (two 1-KB pieces: lane = pixel (lane & 15) of the piece, 16-B column lane >> 4), consumers only
        // multiply.
        const int fw = wave - 4;
        {
            const int pt = t - 256;                            // 0..255
            reinterpret_cast<f32x4*>(lds + DL_Z)[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pt < 128) {
                reinterpret_cast<float*>(lds + DL_OT)[pt] = sc2[pt];
                reinterpret_cast<float*>(lds + DL_OT + 512)[pt] = sh2[pt];
            }
            for (int k = pt; k < K; k += 256) {                // norm1's table: [stage][column][scale 8 | shift 8]
                float* d = reinterpret_cast<float*>(lds + DL_CT) + (k >> 3) * 16 + (k & 7);
                d[0] = sc1[k];
                d[8] = sh1[k];
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rW2 =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w2p), 0, DL_W2_BYTES, 0x00020000);
#pragma unroll
            for (int p = 0; p < 18; ++p) {
                const int piece = fw + 4 * p;
                dma16_buf(rW2, lane * 16, piece * 1024, lb + DL_W2 + piece * 1024);
            }
        }
        // X is channel-blocked, [channel / 32][pixel row][32]: the 32 channels a stage needs of 64 consecutive pixel rows are
        // ONE contiguous 4 KB of memory (in a row-major [row][channels] buffer they were 64 pieces of 64 B, 0.5-2 KB apart)
        // Feeder (par, hw) = (fw >> 1, fw & 1) takes the stages of parity par of every step and of each of them the pixel rows
        // 64 hw .. 64 hw + 63: 4 KB per stage and wave (four 1-KB pieces: lane = pixel (lane & 15) of the piece, 16-B column
        // lane >> 4), ONE chain of [wait for the data -> norm1 + relu1 -> four LDS writes -> four loads] per pair of stages and
        // wave.  (With every feeder on a quarter of every stage the chain ran twice per pair: the pair period was the feeders'.)
        const int par = fw >> 1, hw = fw & 1;
        const int voffA = ((lane & 15) * 32 + 8 * (lane >> 4)) * 2;
        auto strip = [&](int u, int j) {                       // this feeder's 64 pixel rows of step (u, j), channel block 0
            return static_cast<const _Float16*>(X) + (((long)u * J + j) * 128 + 64 * hw) * 32;
        };
        const unsigned ctb = lb + DL_CT + 64 * (lane >> 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of W2
        // The wave's stages of ALL the workgroup's steps are one stream through a 4-deep register ring (its own stage k sits
        // in entry k & 3: three of its stages = six of the step's = 48 KB per CU in flight whatever the step length - an HBM
        // miss takes ~900 cycles on an idle chip and two to three times that under load).  Inline asm loads: the compiler's
        // own vmcnt bookkeeping drains a ring like this at every loop trip; these loads are invisible to it and every use
        // waits by hand - exactly RD stages of four loads are outstanding at each apply, the oldest is the one applied.  The
        // feeders issue no other vector-memory operation after the prologue.
        constexpr int RD = 4;
        u32x4 rq[RD][4];
        const int own = (nst + 1 - par) >> 1;              // this wave's stages per step (nst >= 2: at least one)
        int lu = bid, lj = 0, ls = par;                    // the next load: unit, step, stage
        const _Float16* lrows = strip(lu, lj);
        auto load_next = [&](u32x4(&dst)[4]) {
            const bool in = lu < n_units && !DL_ABL(4);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16*>(lrows) + (in ? ls : 0) * bstride, 0, 4096, 0x00020000);
            const int vo = voffA + (in ? 0 : 0x7f000000);
            asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024\n\t"
                         "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048\n\tbuffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072"
                         : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]) : "v"(vo), "s"(rs) : "memory");
            ls += 2;
            if (ls >= nst) {
                ls = par;
                if (++lj == J) { lj = 0; lu += G; }
                lrows = strip(lu < n_units ? lu : bid, lj);
            }
        };
        auto landed = [&](u32x4(&v)[4]) {                  // the oldest of the RD stages in flight
            asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "n"(4 * RD - 4) : "memory");
        };
        const int nsteps = bid < n_units ? ((n_units - bid + G - 1) / G) * J : 0;
        const int Gt = nsteps * own;                       // this wave's stages
        static_for<0, RD>([&](auto k_c) { load_next(rq[decltype(k_c)::value]); });
        lds_barrier();                                     // B_init: tables and W2 are in the LDS
        // The consumers take the step's stages in PAIRS (one barrier per 64 channels): pair p sits in slots (2p, 2p + 1) & 3 and
        // is read between the barriers B_p and B_p+1; its slots are pair p - 2's, free once B_p-1 has passed; a step's first
        // pair goes in after the previous step's E barrier.  Every feeder executes every barrier, whether or not the pair has
        // a stage of its parity (an odd step's last pair has none of parity 1).
        const int np = (nst + 1) >> 1;                     // pairs (= pair barriers) per step
        ActRegs ak;                                        // norm1 constants of the wave's next stage
        auto request_k = [&](int ss) {
            const unsigned a = ctb + ss * 256;
            ak.s0 = lds_read4<0>(a);
            ak.s1 = lds_read4<16>(a);
            ak.b0 = lds_read4<32>(a);
            ak.b1 = lds_read4<48>(a);
        };
        request_k(par);
        int sa = par, bdone = 0;                           // in-step index of the stage applied next; pair barriers passed in its step
        auto gapply = [&](auto ph_c, int g) {
            constexpr int P = decltype(ph_c)::value;
            if (sa == par && g > 0) {                      // a new step: what is left of the previous one's barriers, then E
                for (; bdone < np; ++bdone) lds_barrier();
                DL_LAP(0);
                lds_barrier();                             // E: its bottleneck tile is complete, every slot is free
                DL_LAP(3);
                bdone = 0;
            }
            for (; bdone < (sa >> 1); ++bdone) lds_barrier();  // ... B_(pair - 1): the pair's slots have been read
            DL_LAP(0);
            landed(rq[P]);
            DL_LAP(7);
            // (younger LDS operations of this wave than the constants: none but a barrier's drain - they were requested
            // after the previous stage's writes)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ak.s0), "+v"(ak.s1), "+v"(ak.b0), "+v"(ak.b1));
            char* d = lds + DL_AR + (sa & 3) * DL_SLOT + 4 * hw * 1024 + lane * 16;
            if (!DL_ABL(1)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(d + q * 1024) = activated(rq[P][q], ak);
            }
            DL_LAP(1);
            load_next(rq[P]);
            sa += 2;
            if (sa >= nst) sa = par;
            request_k(sa);
            DL_LAP(2);
        };
#define DL_GS(k) gapply(std::integral_constant<int, k>{}, g + k); if (g + k + 1 >= Gt) break;
        if (Gt > 0) {
            for (int g = 0;; g += RD) { DL_GS(0) DL_GS(1) DL_GS(2) DL_GS(3) }
            for (; bdone < np; ++bdone) lds_barrier();     // the last step's remaining pair barriers, and E
            lds_barrier();
        }
#undef DL_GS
        // the stages requested past the end of the stream: nothing may reuse their registers before they have landed
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[0][2]), "+v"(rq[0][3]), "+v"(rq[1][0]), "+v"(rq[1][1]),
                       "+v"(rq[1][2]), "+v"(rq[1][3]), "+v"(rq[2][0]), "+v"(rq[2][1]), "+v"(rq[2][2]), "+v"(rq[2][3]),
                       "+v"(rq[3][0]), "+v"(rq[3][1]), "+v"(rq[3][2]), "+v"(rq[3][3])::"memory");
        if (wave == 4) DL_OUT(8);
        return;
    }

    // ===================================================================== consumers (waves 0-3)
    const int nb = wave;                                       // conv1: output channels 32 nb .. 32 nb + 31
    __builtin_amdgcn_s_setprio(3);                             // (the consumers bound the step: their instructions first)
    const int KS = K >> 4;
    const __amdgpu_buffer_rsrc_t rW1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w1p), 0, (unsigned)(128 * K * 2), 0x00020000);
    // W1 fragment ring [stage & 3][k-step]: every step starts at ring phase 0 with its stages 0..2 already requested (before
    // the previous step's conv2 - their latency hides behind it), stage s + 3 is requested when stage s starts.
    u32x4 fr[4][2];
    auto load_w = [&](u32x4(&dst)[2], int ws) {
        const int vo = (ws < nst && !DL_ABL(32)) ? (nb * KS + 2 * ws) * 1024 + lane * 16 : 0x7ffff000;
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rW1, vo, 0, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rW1, vo + 1024, 0, 0);
    };
    auto preload_w = [&]() {                                   // (a pair requests the next pair's fragments itself)
        load_w(fr[0], 0);
        load_w(fr[1], 1);
    };
    preload_w();
    lds_barrier();                                             // B_init
    f32x16 c1[4], a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const unsigned laneA = lb + DL_AR + (i >> 4) * 1024 + h * 256 + (i & 15) * 16;
    // two stages per barrier (the feeders' pair protocol): the second stage's fragments are requested while the first
    // stage multiplies; its W1 fragments of the NEXT pair are requested at the pair's start (L2 hits: one pair ahead is enough)
    auto pair = [&](auto ph_c, int s, auto first_c) {          // first: the step's first pair starts from srcC = 0
        constexpr int P = decltype(ph_c)::value;               // s & 3: 0 or 2
        constexpr bool FIRST = decltype(first_c)::value;
        lds_barrier();                                         // B_(s / 2)
        DL_LAP(0);
        load_w(fr[(P + 2) & 3], s + 2);
        load_w(fr[(P + 3) & 3], s + 3);
        f32x4 av[16];
        if (!DL_ABL(2)) {
        static_for<0, 8>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;            // n = 4 ks + rb
            av[n] = lds_read4<P * DL_SLOT + (n & 3) * 2048 + (n >> 2) * 512>(laneA);
        });
        static_for<0, 8>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            av[8 + n] = lds_read4<(P + 1) * DL_SLOT + (n & 3) * 2048 + (n >> 2) * 512>(laneA);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(av[n]));
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            c1[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fr[P][n >> 2]),
                                                               __builtin_bit_cast(half8, av[n]),
                                                               FIRST && n < 4 ? zero16 : c1[n & 3], 0, 0, 0);
        });
        if (s + 1 < nst)
            static_for<0, 8>([&](auto n_c) {
                constexpr int n = decltype(n_c)::value;
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[8 + n]) : "n"(7 - n));
                c1[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fr[P + 1][n >> 2]),
                                                                   __builtin_bit_cast(half8, av[8 + n]), c1[n & 3], 0, 0, 0);
            });
        else                                                   // an odd step's last pair: the reads still own their registers
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[8]), "+v"(av[9]), "+v"(av[10]), "+v"(av[11]), "+v"(av[12]), "+v"(av[13]),
                                                  "+v"(av[14]), "+v"(av[15]));
        }
        DL_LAP(1);
    };

    auto pairs_from = [&](auto k_c) {
        auto impl = [&](auto& self, auto kk_c) -> void {
            constexpr int k = decltype(kk_c)::value;
            pair(std::integral_constant<int, (2 * k) & 3>{}, 2 * k, std::integral_constant<bool, k == 0>{});
            if constexpr (k + 1 < 16)
                if (2 * k + 2 < nst) self(self, std::integral_constant<int, k + 1>{});
        };
        impl(impl, k_c);
    };

    for (int u = bid; u < n_units; u += G)
        for (int j = 0; j < J; ++j) {
            // ---- conv1 over the step's K channels (the first pair starts the accumulators from the constant zero)
            // no loop: the (up to 16) pairs of a step are laid out one after the other with an exit after each, so that the
            // compiler counts the W1 loads in flight exactly - around a loop it drained them (vmcnt(0)) at every trip
            pairs_from(std::integral_constant<int, 0>{});
            // ---- norm2 + relu2, rounded to fp16, into the bottleneck tile
            if (!DL_ABL(16))
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 osc = *reinterpret_cast<const f32x4*>(lds + DL_OT + (32 * nb + 8 * g + 4 * h) * 4);
                const f32x4 osh = *reinterpret_cast<const f32x4*>(lds + DL_OT + 512 + (32 * nb + 8 * g + 4 * h) * 4);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    unsigned o[2];                             // fp32 fma rounded once to fp16 (v_fma_mix), relu packed
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        asm("v_fma_mixlo_f16 %0, %1, %2, %3\n\t"
                            "v_fma_mixhi_f16 %0, %4, %5, %6\n\t"
                            "v_pk_max_f16 %0, %0, 0"
                            : "=&v"(o[q])
                            : "v"(c1[rb][4 * g + 2 * q]), "v"(osc[2 * q]), "v"(osh[2 * q]), "v"(c1[rb][4 * g + 2 * q + 1]),
                              "v"(osc[2 * q + 1]), "v"(osh[2 * q + 1]));
                    const int px = 32 * rb + i;
                    *reinterpret_cast<uint2*>(lds + DL_BT + (px >> 4) * 4096 + (4 * nb + g) * 256 + (px & 15) * 16 + 8 * h) =
                        make_uint2(o[0], o[1]);
                }
            }
            DL_LAP(2);
            lds_barrier();                                     // E
            DL_LAP(3);
            preload_w();                                       // the next step's first stages (the same weights): behind E, not
                                                               // in front of it - every wave of the workgroup waits for E
            if (Aout) {
                // the TAPE of the gradient path (gnx_dense_layer_f16_tape): the step's activated bottleneck tile, LDS -> HBM, as
                // [4 channel blocks][rows][32] halves.  Consumer w copies channel block w (16-B columns 4w .. 4w + 3) of all 128
                // pixels: lane = (pixel & 15, column & 3), so a read is conflict-free (16 consecutive pixels of one column are 256
                // contiguous bytes) and a store is 1 KB of contiguous memory.  The tile is stable until the next step's first
                // pair barrier, which no wave passes before every consumer has left this step.
                const unsigned la = lb + DL_BT + (4 * nb + (lane >> 4)) * 256 + (lane & 15) * 16;
                _Float16* const ga = Aout + nb * abstride + ((((long)u * J + j) * 128) + (lane & 15)) * 32 + (lane >> 4) * 8;
                f32x4 tv[8];
                static_for<0, 8>([&](auto k_c) { tv[decltype(k_c)::value] = lds_read4<decltype(k_c)::value * 4096>(la); });
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]),
                                                      "+v"(tv[6]), "+v"(tv[7]));
#pragma unroll
                for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(ga + k * 16 * 32) = tv[k];
            }
            conv2_step(u, j, a0, a1);
        }
    if (wave == 0) DL_OUT(0);
}

// ------------------------------------------------------------------------------------------------ S = 64 (block 1: K <= 224)
// At 64 x 64 maps a step has only 2-7 stages of conv1 in front of 288 MFMAs of conv2: run one after the other (the kernel
// above did: conv1 stages, epilogue, conv2 on all eight waves) a 128-pixel step took 6600-10100 cycles against 2800-4100 of
// MFMA work.  Here the two halves of the layer OVERLAP: waves 4-7 ("front") compute conv1 + norm2/relu2 of step n into one
// half of a double-buffered bottleneck tile while waves 0-3 ("back") run conv2 of step n - 1 from the other half; ONE
// workgroup barrier per step.  The front waves take their operands straight from global memory in fragment order (no LDS
// staging, no barrier inside a step): wave (pixel half, channel half) owns 64 pixels x 64 bottleneck channels; a pixel's 32
// channels of a block are 64 contiguous bytes, so a lane's 8 channels are one 16-B load and a fragment one coalesced 2 KB.
// That form pulls every input byte through the texture path twice and W1 twice (32 KB per stage and CU, ~740 cycles at the
// ~43 B/clk the path sustains): affordable with few stages per step, not for S <= 32 (K up to 992), which stay above.
// Timers-only stamps (no ablation branches in the MFMA streams): back 3900 cycles of conv2 + 370 of stores per step.
__global__ __launch_bounds__(512) void dense_layer_f16_s64_kernel(_Float16* __restrict__ X, long bstride, int n_units, int K,
                                                              const _Float16* __restrict__ w1p,
                                                              const _Float16* __restrict__ w2p,
                                                              const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                              const float* __restrict__ sc2, const float* __restrict__ sh2,
                                                              _Float16* __restrict__ Aout, long abstride
                                                              GNX_DL_STAMP_PARAM) {
    constexpr int S = 64, J = S * S / 128, LOG2S = 6;      // 32 steps of two image rows per unit
    __shared__ __attribute__((aligned(16))) char lds[DL_LDS];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int nst = K >> 5;                                // stages per step
    const unsigned lb = lds_addr(lds);
    DL_T0();

    // norm1 + relu1 on raw input pieces (16 px x 32 channels of fp16 each; a lane holds 16 B = 8 channels of one pixel): fp32
    // fma on the fp16 value, rounded once to fp16 (v_fma_mix), relu packed.
    struct ActRegs { f32x4 s0, s1, b0, b1; };
    auto act2 = [](unsigned x, float sa, float ba, float sb, float bb) {                  // two halves of one register
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

    // this workgroup's steps: unit bid, bid + G, ... , each J steps; step n is (u, j) = (bid + (n / J) G, n % J)
    const int N = bid < n_units ? ((n_units - bid + G - 1) / G) * J : 0;

    // ---- conv2 of step n, scatter form: the step's bottleneck tile (LDS, parity n & 1) into every output block it touches.
    auto conv2_step = [&](int n, f32x16& a0, f32x16& a1) {
            const int u = bid + (n / J) * G, j = n & (J - 1);
            const unsigned btb = lb + DL_BT + (n & 1) * DL_BT_BYTES;
            const long R0 = ((long)u * J + j) * 128;
            auto tap3 = [&](f32x16& acc, int O_rel, int dy) {  // the three dx taps of row offset dy into the block at O_rel
                const int o = O_rel + i;
                const int pin = 128 * j + o;
                const int y = pin >> LOG2S, x = pin & (S - 1);
                unsigned aA[3];
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int r = o + dy * S + dx;
                    const bool ok = (unsigned)(y + dy) < (unsigned)S && (unsigned)(x + dx) < (unsigned)S && (unsigned)r < 128u;
                    aA[dx + 1] = ok ? btb + (r >> 4) * 4096 + (r & 15) * 16 + h * 256 : lb + DL_Z + h * 256;
                }
                const unsigned aW = lb + DL_W2 + (dy + 1) * 3 * 8192 + lane * 16;
                // 24 (dx, k-step) products; the fragment pairs of the next D are in flight while one multiplies (a ring of
                // D + 2 register pairs: a pair is overwritten two MFMAs after the MFMA that read it)
                constexpr int D = 7, NSL = D + 2, NE = 24;
                f32x4 ra[NSL] = {}, rw[NSL] = {};
                auto request = [&](auto e_c) {
                    constexpr int e = decltype(e_c)::value, dxi = e / 8, ks = e % 8;
                    if (!DL_ABL(256)) ra[e % NSL] = lds_read4<ks * 512>(aA[dxi]);
                    if (!DL_ABL(128)) rw[e % NSL] = lds_read4<dxi * 8192 + ks * 1024>(aW);
                };
                static_for<0, D>(request);
                static_for<0, NE>([&](auto e_c) {
                    constexpr int e = decltype(e_c)::value;
                    if constexpr (e + D < NE) request(std::integral_constant<int, e + D>{});
                    constexpr int younger = 2 * (e + D < NE ? D : NE - 1 - e);
                    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ra[e % NSL]), "+v"(rw[e % NSL]) : "n"(younger));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[e % NSL]),
                                                                 __builtin_bit_cast(half8, ra[e % NSL]), acc, 0, 0, 0);
                });
            };
            auto store = [&](const f32x16& acc, int O_rel) {
                _Float16* p = X + (K >> 5) * bstride + (R0 + O_rel + i) * 32 + 4 * h;   // the layer's new block [K / 32]
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = (_Float16)acc[4 * g + q];
                    *reinterpret_cast<half4*>(p + 8 * g) = o;
                }
            };
            auto zero = [&](f32x16& acc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            };
            // Which blocks this wave works on in this step: accumulator a0 on the block at O0 with row offsets dy = lo0..hi0,
            // a1 on the block at O1 with lo1..hi1 (an empty range: nothing); zN: the accumulator starts from zero in this
            // step (else it is carried in from the previous one); sN: it is complete after this step and goes to HBM.
            int O0, lo0, hi0, O1 = 0, lo1 = 0, hi1 = -1;
            bool z0, z1 = false, s0, s1 = false;
            // a step = image rows 2j, 2j + 1; a block = half a row; the step touches output rows 2j - 1 .. 2j + 2.  Wave (g, xh)
            // works on x half xh and alternates between FINISHING rows 2j - 1 (its dy = +1 taps) and 2j (dy = 0, +1) and
            // OPENING rows 2j + 1 (dy = -1, 0) and 2j + 2 (dy = -1): 72 MFMAs either way, and what a wave opens in one step
            // is what it finishes in the next.
            const int xh = wave & 1;
            if ((((wave >> 1) + j) & 1) == 0) {
                O0 = -64 + 32 * xh; lo0 = 1; hi0 = j == 0 ? 0 : 1; z0 = false; s0 = j != 0;
                O1 = 32 * xh; lo1 = 0; hi1 = 1; z1 = j == 0; s1 = true;
            } else {
                O0 = 64 + 32 * xh; lo0 = -1; hi0 = 0; z0 = true; s0 = j == J - 1;
                O1 = 128 + 32 * xh; lo1 = -1; hi1 = j == J - 1 ? -2 : -1; z1 = true;
            }
            if (z0) zero(a0);
            if (z1) zero(a1);
            if (DL_ABL(8)) { hi0 = lo0 - 1; hi1 = lo1 - 1; }
#pragma unroll 1
            for (int dy = lo0; dy <= hi0; ++dy) tap3(a0, O0, dy);
            DL_LAP(4);
            if (s0 && !DL_ABL(64)) store(a0, O0);
            DL_LAP(5);
#pragma unroll 1
            for (int dy = lo1; dy <= hi1; ++dy) tap3(a1, O1, dy);
            DL_LAP(4);
            if (s1 && !DL_ABL(64)) store(a1, O1);
            DL_LAP(5);
    };

    if (wave < 4) {
        // ================================================================= back (waves 0-3): tables, W2, then conv2
        {
            reinterpret_cast<f32x4*>(lds + DL_Z)[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < 128) {
                reinterpret_cast<float*>(lds + DL_OT)[t] = sc2[t];
                reinterpret_cast<float*>(lds + DL_OT + 512)[t] = sh2[t];
            }
            for (int k = t; k < K; k += 256) {                 // norm1's table: [stage][column][scale 8 | shift 8]
                float* d = reinterpret_cast<float*>(lds + DL_CT) + (k >> 3) * 16 + (k & 7);
                d[0] = sc1[k];
                d[8] = sh1[k];
            }
            const __amdgpu_buffer_rsrc_t rW2 =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w2p), 0, DL_W2_BYTES, 0x00020000);
#pragma unroll
            for (int p = 0; p < 18; ++p) {
                const int piece = wave + 4 * p;
                dma16_buf(rW2, lane * 16, piece * 1024, lb + DL_W2 + piece * 1024);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of W2
        lds_barrier();                                         // B_init: tables and W2 are in the LDS
        f32x16 a0, a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
        for (int n = 0; n <= N; ++n) {                         // barrier n: tile n is complete, tile n - 1 is free
            if (n >= 1 && Aout) {
                // the tape (see dense_layer_f16_kernel): back wave w copies channel block w of tile n - 1, LDS -> HBM.  The front
                // waves' loads are hand-counted (vmcnt); these stores belong to the back waves, whose memory operations are the
                // compiler's to count.
                const int m = n - 1, u = bid + (m / J) * G, j = m & (J - 1);
                const unsigned la = lb + DL_BT + (m & 1) * DL_BT_BYTES + (4 * wave + (lane >> 4)) * 256 + (lane & 15) * 16;
                _Float16* const ga = Aout + wave * abstride + ((((long)u * J + j) * 128) + (lane & 15)) * 32 + (lane >> 4) * 8;
                f32x4 tv[8];
                static_for<0, 8>([&](auto k_c) { tv[decltype(k_c)::value] = lds_read4<decltype(k_c)::value * 4096>(la); });
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]),
                                                      "+v"(tv[6]), "+v"(tv[7]));
#pragma unroll
                for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4*>(ga + k * 16 * 32) = tv[k];
            }
            if (n >= 1) conv2_step(n - 1, a0, a1);
            DL_LAP(6);
            if (n < N) lds_barrier();
            DL_LAP(0);
        }
        if (wave == 0) DL_OUT(0);
        return;
    }

    // ===================================================================== front (waves 4-7): conv1 of step n -> tile n & 1
    const int fw = wave - 4, ph = fw & 1, ch = fw >> 1;         // pixels 64 ph .. + 63, bottleneck channels 64 ch .. + 63
    const int KS = K >> 4;
    // from four stages on the front bounds the step: its instructions go first on the SIMD it shares with a conv2 wave
    // (K = 224: 3.75 -> 3.55 ms; below that the two halves are balanced and any priority, either way, costs 3-5 %)
    if (nst >= 4) __builtin_amdgcn_s_setprio(3);
    const __amdgpu_buffer_rsrc_t rW1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w1p), 0, (unsigned)(128 * K * 2), 0x00020000);
    // Both operand streams run CONTINUOUSLY over all the workgroup's steps through register rings of four stages (a stage = 32
    // input channels): the input fragments [pixel block][k-step] three stages ahead (raw: norm1 is applied when the stage is
    // used; at K = 64 that is a step and a half ahead - an HBM miss under load takes 2-3 thousand cycles), the W1 fragments
    // [channel block][k-step] two ahead (L2 hits, ~1000 cycles when every CU asks for the same lines).  Inline asm loads with
    // hand-counted waits (the compiler drains its own loads at every trip of a loop like this one); per stage the wave issues
    // x(g + 3) then W(g + 2), so behind x(g) there are 7 younger groups of four loads and behind W(g) four.  Every load is
    // waited for before its registers die.  Past the end of the stream the offsets leave the resource: zeros, no traffic.
    u32x4 rx[4][2][2], rw[4][2][2];
    const int vX = i * 64 + h * 16;                            // lane (pixel i, k half h) in a [64 px][64 B] window
    auto step_rows = [&](int n) {                              // this wave's 64 pixel rows of step n, channel block 0
        const int u = bid + (n / J) * G, j = n & (J - 1);
        return static_cast<const _Float16*>(X) + (((long)u * J + j) * 128 + 64 * ph) * 32;
    };
    int xn = 0, xs = 0;                                        // step / stage of the next input load
    const _Float16* xrows = step_rows(0);
    auto load_x = [&](u32x4(&dst)[2][2]) {
        const bool in = xn < N && !DL_ABL(4);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16*>(xrows) + (in ? xs : 0) * bstride, 0, 4096, 0x00020000);
        const int vo = vX + (in ? 0 : 0x7f000000);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[0][0]) : "v"(vo), "s"(r) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:32" : "=v"(dst[0][1]) : "v"(vo), "s"(r) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:2048" : "=v"(dst[1][0]) : "v"(vo), "s"(r) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:2080" : "=v"(dst[1][1]) : "v"(vo), "s"(r) : "memory");
        if (++xs == nst) {
            xs = 0;
            ++xn;
            xrows = step_rows(xn < N ? xn : 0);
        }
    };
    int wn = 0, ws = 0;                                        // step / stage of the next W1 load (the stage wraps with the step)
    auto load_w = [&](u32x4(&dst)[2][2]) {
        const bool in = wn < N && !DL_ABL(32);
        const int vo = lane * 16 + (in ? 0 : 0x7f000000);
        const int so0 = (2 * ch * KS + 2 * ws) * 1024, so1 = so0 + KS * 1024;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst[0][0]) : "v"(vo), "s"(rW1), "s"(so0) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(dst[0][1]) : "v"(vo), "s"(rW1), "s"(so0) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst[1][0]) : "v"(vo), "s"(rW1), "s"(so1) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:1024" : "=v"(dst[1][1]) : "v"(vo), "s"(rW1), "s"(so1) : "memory");
        if (++ws == nst) {
            ws = 0;
            ++wn;
        }
    };
    ActRegs ac[2];                                             // norm1 constants of the stage used next, per k-step
    const unsigned ctb = lb + DL_CT + 64 * h;
    auto request_consts = [&](int ss) {
        const unsigned a = ctb + ss * 256;
        ac[0].s0 = lds_read4<0>(a);
        ac[0].s1 = lds_read4<16>(a);
        ac[0].b0 = lds_read4<32>(a);
        ac[0].b1 = lds_read4<48>(a);
        ac[1].s0 = lds_read4<128>(a);
        ac[1].s1 = lds_read4<144>(a);
        ac[1].b0 = lds_read4<160>(a);
        ac[1].b1 = lds_read4<176>(a);
    };
    f32x16 c1[2][2];                                           // [channel block][pixel block]
    // ---- norm2 + relu2, rounded to fp16, into the bottleneck tile of the step's parity.  One base register per table and
    // immediate offsets for everything static (written plainly, the compiler hoists 30-odd per-store addresses out of the stage
    // loop and spills ring registers to make room for them)
    const unsigned ot_base = lb + DL_OT + (64 * ch + 4 * h) * 4;
    const unsigned bt_base = lb + DL_BT + ((64 * ph + i) >> 4) * 4096 + (i & 15) * 16 + 8 * h + 8 * ch * 256;
    auto epilogue = [&](int parity) {
        const unsigned bt = bt_base + parity * DL_BT_BYTES;
        if (!DL_ABL(16)) {
        // all sixteen constant reads first, ONE wait (a wait per channel group exposed the LDS latency eight times: 1980
        // cycles per step); the ring registers of the stages in flight leave room for the 64 registers at this point
        f32x4 osc[8], osh[8];
        static_for<0, 8>([&](auto q_c) {
            constexpr int q8 = decltype(q_c)::value, nbl = q8 >> 2, g = q8 & 3;
            osc[q8] = lds_read4<(32 * nbl + 8 * g) * 4>(ot_base);
            osh[q8] = lds_read4<512 + (32 * nbl + 8 * g) * 4>(ot_base);
        });
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(osc[0]), "+v"(osc[1]), "+v"(osc[2]), "+v"(osc[3]), "+v"(osc[4]), "+v"(osc[5]), "+v"(osc[6]), "+v"(osc[7]),
                       "+v"(osh[0]), "+v"(osh[1]), "+v"(osh[2]), "+v"(osh[3]), "+v"(osh[4]), "+v"(osh[5]), "+v"(osh[6]), "+v"(osh[7]));
        static_for<0, 8>([&](auto q_c) {
            constexpr int q8 = decltype(q_c)::value, nbl = q8 >> 2, g = q8 & 3;
            const f32x4 sc_ = osc[q8], sh_ = osh[q8];
            static_for<0, 2>([&](auto r_c) {
                constexpr int rbl = decltype(r_c)::value;
                unsigned o0, o1;                               // fp32 fma rounded once to fp16 (v_fma_mix), relu packed
                asm("v_fma_mixlo_f16 %0, %1, %2, %3\n\t"
                    "v_fma_mixhi_f16 %0, %4, %5, %6\n\t"
                    "v_pk_max_f16 %0, %0, 0"
                    : "=&v"(o0)
                    : "v"(c1[nbl][rbl][4 * g]), "v"(sc_[0]), "v"(sh_[0]), "v"(c1[nbl][rbl][4 * g + 1]), "v"(sc_[1]), "v"(sh_[1]));
                asm("v_fma_mixlo_f16 %0, %1, %2, %3\n\t"
                    "v_fma_mixhi_f16 %0, %4, %5, %6\n\t"
                    "v_pk_max_f16 %0, %0, 0"
                    : "=&v"(o1)
                    : "v"(c1[nbl][rbl][4 * g + 2]), "v"(sc_[2]), "v"(sh_[2]), "v"(c1[nbl][rbl][4 * g + 3]), "v"(sc_[3]), "v"(sh_[3]));
                const uint2 ov = make_uint2(o0, o1);
                const unsigned btl = bt;                       // (an asm operand cannot name a capture of an enclosing lambda)
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(btl), "v"(ov), "n"(rbl * 8192 + (4 * nbl + g) * 256) : "memory");
            });
        });
        }
    };
    auto stage = [&](auto ph_c, int s) {                       // (s: the stage's index in its step)
        constexpr int P = decltype(ph_c)::value;
        // (the stage's own norm1 constants, requested here and not a stage early: carried across the step's end they cost 60
        // registers more than the file has; the eight load issues below cover most of the LDS latency)
        request_consts(s);
        load_x(rx[(P + 3) & 3]);
        load_w(rw[(P + 2) & 3]);
        asm volatile("s_waitcnt vmcnt(28)" : "+v"(rx[P][0][0]), "+v"(rx[P][0][1]), "+v"(rx[P][1][0]), "+v"(rx[P][1][1])::"memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ac[0].s0), "+v"(ac[0].s1), "+v"(ac[0].b0), "+v"(ac[0].b1), "+v"(ac[1].s0), "+v"(ac[1].s1),
                       "+v"(ac[1].b0), "+v"(ac[1].b1));
        u32x4 av[2][2];
#pragma unroll
        for (int rbl = 0; rbl < 2; ++rbl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) av[rbl][ks] = DL_ABL(1) ? rx[P][rbl][ks] : activated(rx[P][rbl][ks], ac[ks]);
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(rw[P][0][0]), "+v"(rw[P][0][1]), "+v"(rw[P][1][0]), "+v"(rw[P][1][1])::"memory");
        if (!DL_ABL(2)) {
            // a step's first k-step starts its accumulators from the constant zero (srcC = 0): no 64 moves per step
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (s == 0) {
#pragma unroll
                for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
                    for (int rbl = 0; rbl < 2; ++rbl)
                        c1[nbl][rbl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[P][nbl][0]),
                                                                              __builtin_bit_cast(half8, av[rbl][0]), zero16, 0, 0, 0);
            } else {
#pragma unroll
                for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
                    for (int rbl = 0; rbl < 2; ++rbl)
                        c1[nbl][rbl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[P][nbl][0]),
                                                                              __builtin_bit_cast(half8, av[rbl][0]), c1[nbl][rbl],
                                                                              0, 0, 0);
            }
#pragma unroll
            for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
                for (int rbl = 0; rbl < 2; ++rbl)
                    c1[nbl][rbl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[P][nbl][1]),
                                                                          __builtin_bit_cast(half8, av[rbl][1]), c1[nbl][rbl],
                                                                          0, 0, 0);
        }
    };

    load_x(rx[0]);
    load_x(rx[1]);
    load_w(rw[0]);
    load_x(rx[2]);
    load_w(rw[1]);
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(rx[0][0][0]), "+v"(rx[0][0][1]), "+v"(rx[0][1][0]), "+v"(rx[0][1][1]), "+v"(rx[1][0][0]), "+v"(rx[1][0][1]),
                   "+v"(rx[1][1][0]), "+v"(rx[1][1][1]), "+v"(rx[2][0][0]), "+v"(rx[2][0][1]), "+v"(rx[2][1][0]), "+v"(rx[2][1][1]),
                   "+v"(rw[0][0][0]), "+v"(rw[0][0][1]), "+v"(rw[0][1][0]), "+v"(rw[0][1][1]), "+v"(rw[1][0][0]), "+v"(rw[1][0][1]),
                   "+v"(rw[1][1][0]), "+v"(rw[1][1][1])::"memory");
    lds_barrier();                                             // B_init
    const int Gt = N * nst;
    // (the step / stage counters are plain locals of this scope, never captured: captured by the lambdas they ended up in
    // scratch memory, i.e. per lane - divergent branches and scratch loads in the middle of the hand-counted stream)
    int n = 0, s = 0;                                          // the step / stage being multiplied
#define DL_ST(k)                                                                                     \
    stage(std::integral_constant<int, k>{}, s);                                                       \
    if (s + 1 == nst) {                                                                               \
        DL_LAP(0);                                                                                    \
        epilogue(n & 1);                                                                              \
        DL_LAP(1);                                                                                    \
        lds_barrier(); /* barrier n: tile n is complete, tile n - 1 is free */                        \
        DL_LAP(2);                                                                                    \
        s = 0;                                                                                        \
        ++n;                                                                                          \
    } else {                                                                                          \
        ++s;                                                                                          \
    }                                                                                                 \
    if (g + k + 1 >= Gt) break;
    if (Gt > 0)
        for (int g = 0;; g += 4) { DL_ST(0) DL_ST(1) DL_ST(2) DL_ST(3) }
#undef DL_ST
    // the stages requested past the end of the stream (out of range: zeros) must land before anything reuses their registers;
    // naming all 32 ring registers here would keep them live through the whole loop (253 spills): a bare wait that nothing
    // may be scheduled across does the same
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (wave == 4) DL_OUT(8);
}

// W1 [128][K] fp32 -> fragment order halves: ((nb * K/16 + ks) * 64 + lane) * 8 + q = W[32 nb + (lane & 31)][16 ks + 8 (lane >> 5) + q]
__global__ void dl_pack_w1_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 128 * K) return;
    const int q = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int KS = K >> 4, nb = f / KS, ks = f - nb * KS;
    out[idx] = (_Float16)w[(long)(32 * nb + (lane & 31)) * K + 16 * ks + 8 * (lane >> 5) + q];
}
// W2 torch [32][128][3][3] fp32 -> ((tap * 8 + ks) * 64 + lane) * 8 + q = W2[lane & 31][16 ks + 8 (lane >> 5) + q][tap]
__global__ void dl_pack_w2_kernel(const float* __restrict__ w, _Float16* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 9 * 8 * 512) return;
    const int q = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int tap = f >> 3, ks = f & 7;
    out[idx] = (_Float16)w[((long)(lane & 31) * 128 + 16 * ks + 8 * (lane >> 5) + q) * 9 + tap];
}

}  // namespace

// conv1.weight [128][K] (fp32, torch layout) -> w1p (128 * K halves), conv2.weight [32][128][3][3] -> w2p (36864 halves): the
// fragment-ordered fp16 operands of gnx_dense_layer_f16 (rounded once; cache them per weight version).
GNX_EXPORT int gnx_dense_layer_f16_pack(const float* w1, const float* w2, void* w1p, void* w2p, int K, hipStream_t stream) {
    if (!w1 || !w2 || !w1p || !w2p || K <= 0) return GNX_ERR_BAD_ARG;
    if (K % 32 != 0) return GNX_ERR_UNSUPPORTED;
    dl_pack_w1_kernel<<<gnx_cdiv(128L * K, 256), 256, 0, stream>>>(w1, reinterpret_cast<_Float16*>(w1p), K);
    dl_pack_w2_kernel<<<gnx_cdiv(9 * 8 * 512, 256), 256, 0, stream>>>(w2, reinterpret_cast<_Float16*>(w2p));
    return gnx_launch_status();
}

#ifdef GNX_DL_STAMP
// ablation bits (diagnostic): 1 no norm1 arithmetic / LDS writes, 2 no conv1 reads / MFMAs, 4 no input loads, 8 no conv2 taps,
// 16 no epilogue, 32 no W1 fragment loads, 64 no output stores, 128 / 256 no conv2 W / A fragment reads
static unsigned long long* g_dl_stamps = nullptr;
static int g_dl_abl = 0;
GNX_EXPORT void gnx_dense_layer_f16_set_stamps(void* buf, int abl) {
    g_dl_stamps = reinterpret_cast<unsigned long long*>(buf);
    g_dl_abl = abl;
}
#endif

// The dense layer on a CHANNEL-BLOCKED fp16 block buffer X16 [channels / 32][rows_total][32] (element (row, c) at
// (c >> 5) * rows_total * 32 + row * 32 + (c & 31); rows = n_img * S * S pixels, the first of rows_total): reads channel
// blocks [0, K / 32), writes block K / 32.  bn_size * growth = 128 and growth = 32 are fixed; S in {4, 8, 16, 32, 64};
// 32 | K, 64 <= K <= 1024; X16 16-B aligned; n_img * S * S a multiple of 128.  scale / shift: the folded running-statistics
// BatchNorms (norm1: K, norm2: 128).
// the k-split form for 64 x 64 and 32 x 32 maps (dense_layer_f16_ks.hip)
int gnx_dense_layer_f16_ks_launch(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                                  const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* A16,
                                  long a_rows_total, int cus, hipStream_t stream);
// Which kernel runs 64 x 64 and 32 x 32 maps (K <= 512): 0 = the form that keeps W2 and the bottleneck tile in LDS (default:
// faster or equal at every shape of config 5 but K = 64), 1 = the k-split form.  Process-wide.
static int g_dense_layer_form = 0;
GNX_EXPORT int gnx_dense_layer_f16_set_form(int form) {
    if (form != 0 && form != 1) return GNX_ERR_BAD_ARG;
    g_dense_layer_form = form;
    return GNX_OK;
}
static bool dense_layer_ksplit() { return g_dense_layer_form == 1; }

static int dense_layer_launch(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                              const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* A16,
                              long a_rows_total, hipStream_t stream) {
    if (!X16 || !w1p || !w2p || !scale1 || !shift1 || !scale2 || !shift2 || n_img < 0 || K <= 0 || S <= 0 ||
        rows_total < n_img * (long)S * S || (A16 && a_rows_total < n_img * (long)S * S))
        return GNX_ERR_BAD_ARG;
    if (K % 32 != 0 || K < 64 || K > 1024 || !al16(X16) || !al16(w1p) || !al16(w2p) || (n_img * S * S) % 128 != 0 ||
        n_img * (long)S * S / 128 >= (1L << 31) || (A16 && !al16(A16)))
        return GNX_ERR_UNSUPPORTED;
    _Float16* const At = reinterpret_cast<_Float16*>(A16);
    const long abs_ = a_rows_total * 32;
    if (n_img == 0) return GNX_OK;
    const long units = S >= 16 ? n_img : n_img * S * S / 128;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GNX_ERR_LAUNCH;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if ((S == 64 || S == 32) && K <= 512 && rows_total * 64 < (1L << 32) && (!A16 || a_rows_total * 64 < (1L << 32)) &&
        dense_layer_ksplit())
        return gnx_dense_layer_f16_ks_launch(X16, rows_total, n_img, S, K, w1p, w2p, scale1, shift1, scale2, shift2, A16,
                                             a_rows_total, cus, stream);
    const int grid = (int)(units < cus ? units : cus);
    _Float16* X = reinterpret_cast<_Float16*>(X16);
    const _Float16* w1 = reinterpret_cast<const _Float16*>(w1p);
    const _Float16* w2 = reinterpret_cast<const _Float16*>(w2p);
#ifdef GNX_DL_STAMP
#define GNX_DL_STAMP_ARG , g_dl_stamps, g_dl_abl
#else
#define GNX_DL_STAMP_ARG
#endif
#define GNX_DL(SS)                                                                                                   \
    dense_layer_f16_kernel<SS><<<grid, 512, 0, stream>>>(X, rows_total * 32, (int)units, K, w1, w2, scale1, shift1, scale2,  \
                                                          shift2, At, abs_                                            \
                                                          GNX_DL_STAMP_ARG);                                          \
    return gnx_launch_status()
    switch (S) {
        case 4: GNX_DL(4);
        case 8: GNX_DL(8);
        case 16: GNX_DL(16);
        case 32: GNX_DL(32);
        case 64:
            dense_layer_f16_s64_kernel<<<grid, 512, 0, stream>>>(X, rows_total * 32, (int)units, K, w1, w2, scale1, shift1,
                                                                 scale2, shift2, At, abs_ GNX_DL_STAMP_ARG);
            return gnx_launch_status();
        default: break;
    }
#undef GNX_DL
    return GNX_ERR_UNSUPPORTED;
}
GNX_EXPORT int gnx_dense_layer_f16(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                                   const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                   hipStream_t stream) {
    return dense_layer_launch(X16, rows_total, n_img, S, K, w1p, w2p, scale1, shift1, scale2, shift2, nullptr, 0, stream);
}
// The same layer as the TAPED forward of the fp16 gradient path (gridnext_amd/densenet_train_f16.py): additionally stores the
// activated bottleneck relu2(norm2(conv1(...))) - the tile the kernel holds in the LDS anyway - as A16 [4][a_rows_total][32]
// halves (channel-blocked like X16; rows = n_img * S * S), the operand of conv2's weight gradient and of norm2's adjoint.
// Everything else, bit for bit, is gnx_dense_layer_f16.
GNX_EXPORT int gnx_dense_layer_f16_tape(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                                        const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                        void* A16, long a_rows_total, hipStream_t stream) {
    if (!A16) return GNX_ERR_BAD_ARG;
    return dense_layer_launch(X16, rows_total, n_img, S, K, w1p, w2p, scale1, shift1, scale2, shift2, A16, a_rows_total, stream);
}
