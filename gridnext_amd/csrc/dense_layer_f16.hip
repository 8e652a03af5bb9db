// One DenseNet dense layer of the fp16-MFMA path (BASELINE config 5) as ONE kernel on fp16 block buffers:
//   cat -> norm1 -> relu1 -> conv1 (1x1, K -> 128) -> norm2 -> relu2 -> conv2 (3x3 pad 1, 128 -> 32) -> 32 new columns
// (/root/reference/gridnext/densenet.py:35-44 is one `forward`).  The unfused pair (gnx_conv1x1_bnrelu_h16 +
// gnx_conv3x3_f16_dma_h) writes the 128-channel bottleneck to HBM and reads it back: 35.0 + 14.1 MB per spot at 256 px where
// this kernel moves 26.6 MB - the K input columns in, 32 columns out, nothing else.
//
// Work unit: an image (S x S map of one spot) for S >= 16, swept top to bottom in steps of 128 pixels; a tile of 128
// pixels (whole images) for S <= 8.  One persistent workgroup per CU, 8 waves:
//   waves 4-7 (producers): stream the step's [128 px][K] input strip global -> LDS by buffer DMA in stages of 32 channels
//     (a ring of 6 x 8 KB, four stages in flight), apply norm1 + relu1 in place on the stage they fetched (fp32 fma on the
//     fp16 value, one rounding: v_fma_mix), one stage ahead of the consumers;
//   waves 0-3 (consumers): conv1 as [128 ch] x [128 px] per step, wave w owning output channels 32w..32w+31 for all 128
//     pixels (its W1 fragments come straight from global memory in a pre-packed fragment order: 1 KB coalesced per
//     fragment, three stages ahead, no LDS); norm2 + relu2 on the accumulators; the activated bottleneck tile
//     [128 px][128 ch] goes to LDS as fp16 - and never to HBM; conv2 reads it back with per-tap shifted fragment addresses
//     (W2, 72 KB in fragment order, is LDS-resident for the workgroup's lifetime).
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
constexpr int DL_NS = 6;                       // input stage ring
constexpr int DL_SLOT = 8192;                  // 128 px x 32 channels: byte(px, chunk) = (px >> 4) * 1024 + chunk * 256 + (px & 15) * 16
constexpr int DL_AR = DL_BT + DL_BT_BYTES;
constexpr int DL_Z = DL_AR + DL_NS * DL_SLOT;  // 4 KB of zeros (masked fragment lanes; immediates reach 7 * 512 + 256 + 16)
constexpr int DL_OT = DL_Z + 4096;             // norm2: scale[128], shift[128]
constexpr int DL_CT = DL_OT + 1024;             // norm1 constants of the stage in each ring slot: [slot][16-B column 4][scale 8 | shift 8] floats
constexpr int DL_LDS = DL_CT + DL_NS * 256;
static_assert(DL_LDS <= 160 * 1024, "LDS");
constexpr int DL_PF = 3;                       // W1 fragment stages in flight ahead of their use

// Diagnostic build only (tools/ubench/dl_stamps.py compiles this file with -DGNX_DL_STAMP into its own library): per
// workgroup, wave 0 and wave 4 sum the shader cycles they spend in each segment of a step and leave them in a buffer of
// their own.  In the product build the macros are empty: no stamp executes.
#ifdef GNX_DL_STAMP
#define GNX_DL_STAMP_PARAM , unsigned long long* __restrict__ stamps, int abl
#define DL_ABL(bit) (abl & (bit))
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
__global__ __launch_bounds__(512) void dense_layer_f16_kernel(_Float16* __restrict__ X, long ld, int n_units, int K,
                                                              const _Float16* __restrict__ w1p,
                                                              const _Float16* __restrict__ w2p,
                                                              const float* __restrict__ sc1, const float* __restrict__ sh1,
                                                              const float* __restrict__ sc2, const float* __restrict__ sh2
                                                              GNX_DL_STAMP_PARAM) {
    constexpr int J = S >= 16 ? S * S / 128 : 1;           // steps per unit
    constexpr int LOG2S = S == 64 ? 6 : S == 32 ? 5 : S == 16 ? 4 : S == 8 ? 3 : 2;
    __shared__ __attribute__((aligned(16))) char lds[DL_LDS];
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, i = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    const int nst = K >> 5;                                // stages per step
    const unsigned lb = lds_addr(lds);

    // norm1 + relu1 in place on 1-KB pieces (16 px x 32 channels each) of a landed stage, lane = (pixel, 16-B column) as the DMA
    // wrote them: a lane's 8 channels are the same in every piece, their scale / shift come from the slot's side area.  fp32
    // fma on the fp16 value, rounded once to fp16 (v_fma_mix), relu packed.  The pass is an LDS round trip whose latency
    // (read ~150 cycles, write ~100) every wave of the workgroup used to wait out before the stage's barrier (stamped: ~450
    // cycles per piece, whatever the piece count).  It is therefore SPLIT over two stages: a wave requests the raw pieces of
    // stage t + 2 (and their constants) right after barrier B_t - they arrive behind its other work - and applies the
    // arithmetic and the write first thing after B_{t+1}, long before B_{t+2} needs them.
    struct ActRegs { f32x4 s0, s1, b0, b1; };
    auto act_request = [&](auto np_c, unsigned base, unsigned ct, f32x4 (&v)[decltype(np_c)::value], ActRegs& c) {
        constexpr int NP = decltype(np_c)::value;
        static_for<0, NP>([&](auto p_c) { v[decltype(p_c)::value] = lds_read4<decltype(p_c)::value * 1024>(base); });
        c.s0 = lds_read4<0>(ct);
        c.s1 = lds_read4<16>(ct);
        c.b0 = lds_read4<32>(ct);
        c.b1 = lds_read4<48>(ct);
    };
    auto act_apply = [&](auto np_c, char* base, f32x4 (&v)[decltype(np_c)::value], const ActRegs& c) {
        constexpr int NP = decltype(np_c)::value;
        if (DL_ABL(1)) return;
        auto act2 = [](float x, float sa, float ba, float sb, float bb) {              // two halves of one register
            unsigned r;
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_pk_max_f16 %0, %0, 0"
                : "=&v"(r) : "v"(x), "v"(sa), "v"(ba), "v"(sb), "v"(bb));
            return r;
        };
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            u32x4 o;
            o[0] = act2(v[p][0], c.s0[0], c.b0[0], c.s0[1], c.b0[1]);
            o[1] = act2(v[p][1], c.s0[2], c.b0[2], c.s0[3], c.b0[3]);
            o[2] = act2(v[p][2], c.s1[0], c.b1[0], c.s1[1], c.b1[1]);
            o[3] = act2(v[p][3], c.s1[2], c.b1[2], c.s1[3], c.b1[3]);
            *reinterpret_cast<u32x4*>(base + p * 1024) = o;
        }
    };

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
                constexpr int D = 6, NSL = D + 2, NE = 24;
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
                _Float16* p = X + (R0 + O_rel + i) * ld + K + 4 * h;
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
            if constexpr (S == 64) {
                // a step = image rows 2j, 2j + 1; a block = half a row; the step touches output rows 2j - 1 .. 2j + 2, i.e. eight
                // blocks, one per wave: wave (ph, xh) owns the rows congruent to ph mod 4 (x half xh), so a row that waits for the
                // next step stays with its wave.  Rows 2j - 1 and 2j are finished here (their dy = -1 / 0 taps came last step),
                // rows 2j + 1 and 2j + 2 are opened.
                const int xh = wave & 1, m = ((wave >> 1) - 2 * j + 1) & 3;
                O0 = (m == 0 ? -64 : m == 1 ? 0 : m == 2 ? 64 : 128) + 32 * xh;
                lo0 = m == 0 ? 1 : m == 1 ? 0 : -1;
                hi0 = m == 0 ? (j == 0 ? 0 : 1) : m == 1 ? 1 : m == 2 ? 0 : (j == J - 1 ? -2 : -1);
                z0 = m >= 2 || (m == 1 && j == 0);
                s0 = m == 0 ? j != 0 : m == 1 ? true : m == 2 ? j == J - 1 : false;
            } else if constexpr (S >= 16) {
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
            if (s0) store(a0, O0);
#pragma unroll 1
            for (int dy = lo1; dy <= hi1; ++dy) tap3(a1, O1, dy);
            if (s1) store(a1, O1);
    };

    if (wave >= 4) {
        // ================================================================= producers
        const int pw = wave - 4;
        f32x16 pa0, pa1;                                       // conv2 accumulators (S = 64: these waves take part)
#pragma unroll
        for (int r = 0; r < 16; ++r) { pa0[r] = 0.f; pa1[r] = 0.f; }
        {
            const int pt = t - 256;                            // 0..255
            reinterpret_cast<f32x4*>(lds + DL_Z)[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (pt < 128) {
                reinterpret_cast<float*>(lds + DL_OT)[pt] = sc2[pt];
                reinterpret_cast<float*>(lds + DL_OT + 512)[pt] = sh2[pt];
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rW2 =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w2p), 0, DL_W2_BYTES, 0x00020000);
#pragma unroll
            for (int p = 0; p < 18; ++p) {
                const int piece = pw + 4 * p;
                dma16_buf(rW2, lane * 16, piece * 1024, lb + DL_W2 + piece * 1024);
            }
        }
        // Waves 4, 5 are LOADERS, waves 6, 7 ACTIVATORS: the DMA issue of a stage (~100 cycles per 1-KB piece) and its
        // norm1 + relu1 pass (an LDS round trip) then run side by side instead of one after the other in each wave - stamped,
        // the serial form spent 250 + 540 cycles per stage and was what every wave of the workgroup waited for.
        if (pw < 2) {
            // ---- loaders.  A stage = four 1-KB pieces per loader (16 px x 32 channels each; loader l: pixels 64 l .. 64 l + 63)
            // + the stage's norm1 constants (scale / shift of its 32 channels, 256 B into the slot's side area: 16 lanes, the two
            // loaders take turns): the constants travel with the data.  Past this workgroup's last stage the cursor re-reads the
            // workgroup's first unit into slots nobody will read: cheaper than a branch around every issue (and vmcnt stays
            // countable).
            int cu = bid, cj = 0, cs = 0, cslot = 0, cturn = 0;
            const unsigned voffA = (unsigned)(((lane & 15) * ld + 8 * (lane >> 4)) * 2);
            // constants piece, lane l < 16: 16-B column q = l >> 2 of the stage, {scale lo, scale hi, shift lo, shift hi}[l & 3]
            const float* const csrc = ((lane & 2) ? sh1 : sc1) + 8 * ((lane >> 2) & 3) + 4 * (lane & 1);
            const _Float16* pbase = X + ((long)bid * J * 128 + 64 * pw) * ld;             // this wave's rows of the cursor's step
            const long step_stride = 128 * ld;
            auto issue = [&]() {
                if (DL_ABL(4) && cslot != 99) { cslot = cslot == DL_NS - 1 ? 0 : cslot + 1; return; }
                if (cturn == pw && lane < 16) dma16_global(csrc + 32 * cs, lb + DL_CT + cslot * 256);
                cturn ^= 1;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<_Float16*>(pbase) + 16 * g4 * ld, 0, (unsigned)(16 * ld * 2), 0x00020000);
                    dma16_buf(rs, voffA, cs * 64, lb + DL_AR + cslot * DL_SLOT + (4 * pw + g4) * 1024);
                }
                cslot = cslot == DL_NS - 1 ? 0 : cslot + 1;
                if (++cs == nst) {
                    cs = 0;
                    pbase += step_stride;
                    if (++cj == J) {
                        cj = 0;
                        cu += G;
                        if (cu >= n_units) cu = bid;
                        pbase = X + ((long)cu * J * 128 + 64 * pw) * ld;
                    }
                }
            };
#pragma unroll
            for (int q = 0; q < DL_NS - 1; ++q) issue();       // stages 0..4
            // W2, stages 0, 1 and 2 and their constants have landed (left in flight: the 8 pieces of stages 3 and 4, and a
            // constants piece among them or none: then a piece more is waited for - never less than needed)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            lds_barrier();                                     // B_init
            DL_T0();
            for (int u = bid; u < n_units; u += G)
                for (int j = 0; j < J; ++j) {
                    for (int s = 0; s < nst; ++s) {
                        lds_barrier();                         // B_t: stage t - 1 is consumed
                        DL_LAP(0);
                        issue();                               // stage t + 5 into the slot of stage t - 1
                        DL_LAP(1);
                        // stage t + 3 and its constants have landed: its raw pieces are requested after the next barrier
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        DL_LAP(2);
                    }
                    lds_barrier();                             // E
                    if constexpr (S == 64) conv2_step(u, j, pa0, pa1);
                    DL_LAP(4);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave == 4) DL_OUT(8);
            return;
        }
        // ---- activators: norm1 + relu1 on pixels 64..127 of a landed stage (activator a: the two pieces 64 + 32 a ..); the
        // consumer waves, idle for most of a stage, take a piece each of pixels 0..63 behind their MFMAs: the pass is an LDS
        // round trip (~200 cycles per piece, stamped), and the stage's barrier waits for its slowest wave.
        const int aw = pw - 2;
        int aslot = 0;                                         // slot whose pieces are held in registers (requested last stage)
        f32x4 av2[2];
        ActRegs ac;
        const unsigned abase = lb + DL_AR + (4 + 2 * aw) * 1024 + lane * 16, actb = lb + DL_CT + 64 * (lane >> 4);
        auto request = [&](int sl) {
            act_request(std::integral_constant<int, 2>{}, abase + sl * DL_SLOT, actb + sl * 256, av2, ac);
        };
        auto apply = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av2[0]), "+v"(av2[1]), "+v"(ac.s0), "+v"(ac.s1), "+v"(ac.b0), "+v"(ac.b1));
            act_apply(std::integral_constant<int, 2>{}, lds + DL_AR + aslot * DL_SLOT + (4 + 2 * aw) * 1024 + lane * 16, av2, ac);
        };
        // (Tried and dropped, measured: an L2 touch-prefetch from these waves - one dword per pixel row of the stage 7 or 13
        // stages ahead of the loaders, never waited for - made every shape 5-15 % SLOWER; DMA pieces of 8 rows x 128 B instead
        // of 16 rows x 64 B moved the DMA stream alone from 4.0 to 4.3 TB/s and the whole kernel not at all.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of W2
        lds_barrier();                                         // B_init: stages 0, 1 and 2 are in the LDS
        request(0);
        apply();                                               // stage 0 (the consumers: their pieces of it)
        aslot = 1;
        request(1);
        DL_T0();
        for (int u = bid; u < n_units; u += G)
            for (int j = 0; j < J; ++j) {
                for (int s = 0; s < nst; ++s) {
                    lds_barrier();                             // B_t: stage t is visible to the consumers; stage t + 2 has landed
                    DL_LAP(0);
                    apply();                                   // stage t + 1, requested during the last stage
                    aslot = aslot == DL_NS - 1 ? 0 : aslot + 1;
                    request(aslot);                            // stage t + 2: arrives while the others work
                    DL_LAP(3);
                }
                lds_barrier();                                 // E: the step's bottleneck tile is complete
                if constexpr (S == 64) conv2_step(u, j, pa0, pa1);
                DL_LAP(4);
            }
        if (wave == 6) DL_OUT(16);
        return;
    }

    // ===================================================================== consumers
    const int nb = wave;                                       // conv1: output channels 32 nb .. 32 nb + 31
    const int KS = K >> 4;
    const __amdgpu_buffer_rsrc_t rW1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(w1p), 0, (unsigned)(128 * K * 2), 0x00020000);
    // W1 fragment ring [stage % 4][k-step]: every step starts at ring phase 0 with its stages 0..2 already requested (before
    // the previous step's conv2 - their latency hides behind it), stage s + 3 is requested when stage s starts.
    u32x4 fr[4][2];
    // (a stage index past the step's last one makes the per-lane offset exceed the resource's extent: the load returns zeros
    // without touching memory - no branch, so the compiler can count the loads in flight exactly)
    auto load_w = [&](u32x4(&dst)[2], int ws) {
        const int vo = ws < nst ? (nb * KS + 2 * ws) * 1024 + lane * 16 : 0x7ffff000;
        dst[0] = __builtin_amdgcn_raw_buffer_load_b128(rW1, vo, 0, 0);
        dst[1] = __builtin_amdgcn_raw_buffer_load_b128(rW1, vo + 1024, 0, 0);
    };
    auto preload_w = [&]() {
        load_w(fr[0], 0);
        load_w(fr[1], 1);
        load_w(fr[2], 2);
    };
    preload_w();
    lds_barrier();                                             // B_init
    // this wave's share of the norm1 + relu1 pass: piece `wave` (pixels 16 wave .. 16 wave + 15) of the stage after next
    f32x4 cv1[1];
    ActRegs cc;
    const unsigned cbase = lb + DL_AR + wave * 1024 + lane * 16, cctb = lb + DL_CT + 64 * (lane >> 4);
    int hslot = 0;                                             // slot whose piece is held in registers
    auto request_mine = [&](int sl) {
        act_request(std::integral_constant<int, 1>{}, cbase + sl * DL_SLOT, cctb + sl * 256, cv1, cc);
    };
    auto apply_mine = [&]() {
        act_apply(std::integral_constant<int, 1>{}, lds + DL_AR + hslot * DL_SLOT + wave * 1024 + lane * 16, cv1, cc);
    };
    request_mine(0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv1[0]), "+v"(cc.s0), "+v"(cc.s1), "+v"(cc.b0), "+v"(cc.b1));
    apply_mine();
    hslot = 1;
    request_mine(1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv1[0]), "+v"(cc.s0), "+v"(cc.s1), "+v"(cc.b0), "+v"(cc.b1));
    f32x16 c1[4], a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const unsigned laneA = lb + DL_AR + (i >> 4) * 1024 + h * 256 + (i & 15) * 16;
    int slot = 0;
    DL_T0();
    // one stage: the 8 operand fragments (4 pixel blocks x 2 k-steps) are requested together, each MFMA waits for its own
    auto stage = [&](auto ph_c, int s) {
        constexpr int P = decltype(ph_c)::value;
        lds_barrier();                                         // B_t
        DL_LAP(0);
        apply_mine();                                          // stage t + 1's piece, requested during the last stage
        hslot = hslot == DL_NS - 1 ? 0 : hslot + 1;
        load_w(fr[(P + DL_PF) & 3], s + DL_PF);
        request_mine(hslot);                                   // stage t + 2's: older than the fragment reads below, so the
        const unsigned ab = laneA + slot * DL_SLOT;            // counted waits there cover it
        f32x4 av[8];
        if (!DL_ABL(2)) {
        static_for<0, 8>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;            // n = 4 ks + rb
            av[n] = lds_read4<(n & 3) * 2048 + (n >> 2) * 512>(ab);
        });
        static_for<0, 8>([&](auto n_c) {
            constexpr int n = decltype(n_c)::value;
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(av[n]) : "n"(7 - n));
            c1[n & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fr[P][n >> 2]),
                                                               __builtin_bit_cast(half8, av[n]), c1[n & 3], 0, 0, 0);
        });
        }
        slot = slot == DL_NS - 1 ? 0 : slot + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cv1[0]), "+v"(cc.s0), "+v"(cc.s1), "+v"(cc.b0), "+v"(cc.b1));
        DL_LAP(1);
    };

    for (int u = bid; u < n_units; u += G)
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) c1[rb][r] = 0.f;
            // ---- conv1 over the step's K channels
            for (int s = 0;; s += 4) {
                stage(std::integral_constant<int, 0>{}, s);
                if (s + 1 >= nst) break;
                stage(std::integral_constant<int, 1>{}, s + 1);
                if (s + 2 >= nst) break;
                stage(std::integral_constant<int, 2>{}, s + 2);
                if (s + 3 >= nst) break;
                stage(std::integral_constant<int, 3>{}, s + 3);
                if (s + 4 >= nst) break;
            }
            preload_w();                                       // the next step's first stages (the same weights)
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
            conv2_step(u, j, a0, a1);
            DL_LAP(4);
        }
    if (wave == 0) DL_OUT(0);
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
// ablation bits (diagnostic): 1 no norm1 pass, 2 no conv1 reads / MFMAs, 4 no DMA issue, 8 no conv2 taps, 16 no epilogue
static unsigned long long* g_dl_stamps = nullptr;
static int g_dl_abl = 0;
GNX_EXPORT void gnx_dense_layer_f16_set_stamps(void* buf, int abl) {
    g_dl_stamps = reinterpret_cast<unsigned long long*>(buf);
    g_dl_abl = abl;
}
#endif

// The dense layer on an fp16 block buffer X16 [n_img * S * S][ld16]: reads columns [0, K), writes columns [K, K + 32).
// bn_size * growth = 128 and growth = 32 are fixed; S in {4, 8, 16, 32, 64}; 32 | K; 8 | ld16, K + 32 <= ld16; X16 16-B
// aligned; n_img * S * S a multiple of 128.  scale / shift: the folded running-statistics BatchNorms (norm1: K, norm2: 128).
GNX_EXPORT int gnx_dense_layer_f16(void* X16, long ld16, long n_img, int S, int K, const void* w1p, const void* w2p,
                                   const float* scale1, const float* shift1, const float* scale2, const float* shift2,
                                   hipStream_t stream) {
    if (!X16 || !w1p || !w2p || !scale1 || !shift1 || !scale2 || !shift2 || n_img < 0 || K <= 0 || ld16 < K + 32)
        return GNX_ERR_BAD_ARG;
    if (K % 32 != 0 || ld16 % 8 != 0 || !al16(X16) || !al16(w1p) || !al16(w2p) || (n_img * S * S) % 128 != 0 ||
        ld16 > 32768 || n_img * (long)S * S / 128 >= (1L << 31))
        return GNX_ERR_UNSUPPORTED;
    if (n_img == 0) return GNX_OK;
    const long units = S >= 16 ? n_img : n_img * S * S / 128;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GNX_ERR_LAUNCH;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
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
    dense_layer_f16_kernel<SS><<<grid, 512, 0, stream>>>(X, ld16, (int)units, K, w1, w2, scale1, shift1, scale2, shift2 \
                                                          GNX_DL_STAMP_ARG);                                          \
    return gnx_launch_status()
    switch (S) {
        case 4: GNX_DL(4);
        case 8: GNX_DL(8);
        case 16: GNX_DL(16);
        case 32: GNX_DL(32);
        case 64: GNX_DL(64);
        default: break;
    }
#undef GNX_DL
    return GNX_ERR_UNSUPPORTED;
}
