// Micro-benchmark for a k-split, REGISTER-FED conv2 (3x3, 128 -> 32 channels, fp16 MFMA) of the fused dense layer:
//   * a wave holds the activated bottleneck of ITS 32 channels (conv1's accumulator registers ARE the B operand of conv2 when
//     W2's k order follows the accumulator layout) and its slice of W2 (9 taps x 2 k-steps = 72 VGPRs);
//   * dy taps are other registers, dx = -1 / +1 taps are v_mov_b32_dpp wave_shr:1 / wave_shl:1 of the dx = 0 fragment under an
//     EXEC mask that leaves the two lanes at the 32-lane seam untouched (they stay zero: the image border) - ONE VALU per
//     register, no LDS read at all;
//   * the four k-slices' partial sums go through LDS once per step (3 x 4 KB written and read per wave), one barrier.
// Two workgroups of four waves per CU (two independent chains per SIMD).  VPR = DPP moves per out row (48: 32-px rows, every
// use shifted; 24: 64-px rows with the x parity split over two tiles - half the shifted fragments are the other tile's registers).
//   hipcc -w --offload-arch=gfx950 -O3 tools/ubench/conv2_regs.hip -o tools/ubench/build/conv2_regs && tools/ubench/build/conv2_regs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// t = src shifted by one lane towards higher (RIGHT) or lower lanes; the seam lanes of t are never written (they hold zero)
template <bool RIGHT>
__device__ __forceinline__ void shift8(i32x4& t0, i32x4& t1, const i32x4& s0, const i32x4& s1) {
    if (RIGHT)
        asm volatile(
            "s_mov_b64 exec, %[m]\n\ts_nop 4\n\t"
            "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %2, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %4, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %13 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %6, %14 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %15 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
            : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]),
              [m] "s"(0xfffffffefffffffeull));
    else
        asm volatile(
            "s_mov_b64 exec, %[m]\n\ts_nop 4\n\t"
            "v_mov_b32_dpp %0, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %9 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %2, %10 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %11 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %4, %12 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %13 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32_dpp %6, %14 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %15 wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
            : "v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(s1[0]), "v"(s1[1]), "v"(s1[2]), "v"(s1[3]),
              [m] "s"(0x7fffffff7fffffffull));
}

// HALF: only every second shifted fragment costs DPP moves (64-px rows, x parity split over two tiles)
template <bool HALF, bool EXCH>
__global__ __launch_bounds__(256, 2) void k(const _Float16* __restrict__ in, _Float16* __restrict__ out, int iters, int* check) {
    __shared__ __attribute__((aligned(16))) char lds[12 * 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    i32x4 wreg[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) wreg[f] = *reinterpret_cast<const i32x4*>(in + ((f * 4 + wave) * 64 + lane) * 8);
    i32x4 rows[6][2];                                      // six rows of 32 px x this wave's 32 channels: [row][k-step]
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int s = 0; s < 2; ++s) rows[r][s] = *reinterpret_cast<const i32x4*>(in + 65536 + ((r * 2 + s) * 64 + lane) * 8);
    i32x4 tl[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, tr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 fin;
#pragma unroll
    for (int r = 0; r < 16; ++r) fin[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {                       // four out rows per step, each from zero: 27 (dy, dx, k-step) products
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const i32x4* src = rows[o + dy];
                if (!HALF || ((o + dy) & 1) == 0) shift8<true>(tl[0], tl[1], src[0], src[1]);
                if (!HALF || ((o + dy) & 1) == 1) shift8<false>(tr[0], tr[1], src[0], src[1]);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wreg[(3 * dy) * 2 + s]),
                                                                 __builtin_bit_cast(h8, tl[s]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wreg[(3 * dy + 1) * 2 + s]),
                                                                 __builtin_bit_cast(h8, src[s]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, wreg[(3 * dy + 2) * 2 + s]),
                                                                 __builtin_bit_cast(h8, tr[s]), acc, 0, 0, 0);
                }
            }
            if (EXCH) {
                if (o == wave) {
                    fin = acc;
                } else {                                    // this wave's partial sum of out row o, for wave o to add
                    const int slot = o * 3 + ((wave - o - 1) & 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(lds + slot * 4096 + q * 1024 + lane * 16) =
                            f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) fin[r] += acc[r];
            }
        }
        if (EXCH) {
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + (wave * 3 + p) * 4096 + q * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fin[4 * q + e] += v[e];
                }
            _Float16* po = out + ((size_t)blockIdx.x * 4 + wave) * 2048 + (lane & 31) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 ov;
#pragma unroll
                for (int q = 0; q < 4; ++q) ov[q] = (_Float16)fin[4 * g + q];
                *reinterpret_cast<h4*>(po + 8 * g) = ov;
            }
            __syncthreads();                                // the partial-sum region is free again
        }
        // the next step: two rows stay (the previous step's last two), four are "new" (a register permutation stands in for conv1)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const i32x4 a = rows[0][s], b = rows[1][s];
            rows[0][s] = rows[4][s];
            rows[1][s] = rows[5][s];
            rows[4][s] = rows[2][s];
            rows[5][s] = rows[3][s];
            rows[2][s] = a;
            rows[3][s] = b;
        }
    }
    if (check) {                                           // the seam lanes of the shifted fragments must still be zero
        int bad = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((lane & 31) == 0 && tl[s][q] != 0) bad = 1;
                if ((lane & 31) == 31 && tr[s][q] != 0) bad = 1;
            }
        if (bad) atomicAdd(check, 1);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += fin[r];
    if (s == 12345.678f) out[0] = (_Float16)s;
}

template <bool HALF, bool EXCH>
static void run(const char* what) {
    const int grid = 512, iters = 4000;
    _Float16 *in, *out;
    int* chk;
    hipMalloc(&in, 400000);
    hipMemset(in, 0x11, 400000);
    hipMalloc(&out, (size_t)grid * 4 * 2048 * 2);
    hipMalloc(&chk, 4);
    hipMemset(chk, 0, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<HALF, EXCH><<<grid, 256>>>(in, out, 100, chk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<HALF, EXCH><<<grid, 256>>>(in, out, iters, nullptr);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    int bad = 0;
    hipMemcpy(&bad, chk, 4, hipMemcpyDeviceToHost);
    const double mfma = (double)grid * 4 * iters * 72.0;
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-84s %7.3f ms  %7.1f TFLOP/s  (%.1f %% of 2500)%s\n", what, ms, tf, tf / 25.0, bad ? "  SEAM LANES NOT ZERO" : "");
    hipFree(in);
    hipFree(out);
    hipFree(chk);
}

int main() {
    run<false, false>("registers only, 48 DPP moves per out row (32-px rows), no exchange");
    run<true, false>("registers only, 24 DPP moves per out row (64-px rows, parity tiles), no exchange");
    run<false, true>("registers only, 48 DPP moves per out row + partial sums through LDS + stores");
    run<true, true>("registers only, 24 DPP moves per out row + partial sums through LDS + stores");
    return 0;
}
