// Micro-benchmark for the NEXT form of the fp16 conv2 (3x3, 128 -> 32 channels) inside the fused dense layer: how fast can a CU
// issue v_mfma_f32_32x32x16_f16 depending on where the two operands come from?
//   mode 0  as shipped (dense_layer_f16.hip): W2 fragments AND activation fragments from LDS, a W2 fragment shared by a wave's two
//           pixel tiles: 1.5 ds_read_b128 per MFMA;
//   mode 1  W2 in REGISTERS (the 128 bottleneck channels split over the four waves: 2 k-steps x 9 taps = 18 fragments = 72
//           VGPRs, as the backward's conv2 kernels hold it), activation fragments from LDS: 1 read per MFMA (the partial sums of
//           the four k-slices then have to be added through LDS: not part of this loop);
//   mode 2  mode 1 + the dx = -1 / +1 fragments DERIVED from the dx = 0 fragment in registers: lane n of the B operand holds
//           pixel column n, so a shift by one pixel is v_mov_b32_dpp wave_shr:1 / wave_shl:1 on the fragment's 4 VGPRs plus a
//           select for the two lanes at the ends of each 32-lane half (their neighbour pixel comes from one extra 16-B read per
//           row): 2 reads + 8 DPP moves + 8 selects per THREE MFMAs.
// One workgroup per CU, NW waves (4 = one per SIMD; 8 = two - in modes 1 / 2 the k-slice of a wave is then one k-step: 36
// registers of weights, eight partial sums to add).  Prints the matrix rate of the whole chip.
//   hipcc -w --offload-arch=gfx950 -O3 tools/ubench/conv2_operands.hip -o tools/ubench/build/conv2_operands && tools/ubench/build/conv2_operands
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int RS = 272;                  // bytes per activation row (128 channels + 16)
constexpr int ROWS = 320;                // rows of the activation tile in LDS
constexpr int W_BYTES = 9 * 8 * 64 * 16; // W2 in fragment order: [tap][k-step][lane] x 16 B

__device__ __forceinline__ h8 shift_lanes(h8 v, h8 edge, bool right, int lane) {
    i32x4 x = __builtin_bit_cast(i32x4, v);
    const i32x4 e = __builtin_bit_cast(i32x4, edge);
    i32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = right ? __builtin_amdgcn_update_dpp(0, x[q], 0x138, 0xf, 0xf, false)      // wave_shr:1
                            : __builtin_amdgcn_update_dpp(0, x[q], 0x130, 0xf, 0xf, false);     // wave_shl:1
        const bool end = right ? (lane & 31) == 0 : (lane & 31) == 31;
        r[q] = end ? e[q] : s;
    }
    return __builtin_bit_cast(h8, r);
}

template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const act = lds;
    char* const wl = lds + ROWS * RS;
    for (int i = threadIdx.x; i < (ROWS * RS + W_BYTES) / 4; i += 64 * NW) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    constexpr int KSW = (MODE >= 1 && NW == 8) ? 1 : 2;   // k-steps of a wave's slice of the 128 bottleneck channels (modes 1, 2)
    h8 wreg[18];
    if (MODE >= 1) {
#pragma unroll
        for (int f = 0; f < 18; ++f) wreg[f] = *reinterpret_cast<const h8*>(wl + ((f * 4 + (wave & 3)) * 64 + lane) * 16);
    }
    for (int it = 0; it < iters; ++it) {
        const int rot = (it & 7) * 2;                       // the fragment addresses move with the step: nothing is loop-invariant
        if (MODE == 0) {
            // wave = 2 pixel tiles; all 8 k-steps, 9 taps: per (tap, k-step) 1 W read, 2 activation reads, 2 MFMAs.  Half of the
            // k-steps per call of this loop body, so that every mode issues 72 MFMAs per wave and step.
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const h8 w = *reinterpret_cast<const h8*>(wl + ((tap * 8 + ks + 4 * (it & 1)) * 64 + lane) * 16);
                    const int row = 34 + rot + (tap / 3) * 32 + (tap % 3) - 1 + n;      // a shifted copy of the pixel rows
                    const h8 b0 = *reinterpret_cast<const h8*>(act + (row + 64 * (wave & 1)) * RS + (16 * ks + 8 * h) * 2);
                    const h8 b1 = *reinterpret_cast<const h8*>(act + (row + 64 * (wave & 1) + 32) * RS + (16 * ks + 8 * h) * 2);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, b1, acc[1], 0, 0, 0);
                }
        } else if (MODE == 1) {
            // wave = a k-slice of 32 channels (2 k-steps), all 4 pixel tiles of the step, 9 taps: 1 activation read per MFMA
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks) {
                        const int row = 34 + rot + (tap / 3) * 32 + (tap % 3) - 1 + n + 32 * pt;
                        const h8 b = *reinterpret_cast<const h8*>(act + row * RS + (16 * KSW * (wave & 7) + 16 * ks + 8 * h) * 2);
                        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[tap * 2 + ks], b, acc[pt], 0, 0, 0);
                    }
        } else {
            // as mode 1, but per (pixel tile, dy, k-step) ONE fragment read + one 16-B read of the two neighbour pixels; the
            // dx = -1 / +1 fragments are lane shifts of the dx = 0 fragment
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int ks = 0; ks < KSW; ++ks) {
                        const int row = 34 + rot + dy * 32 + n + 32 * pt;
                        const int col = (16 * KSW * (wave & 7) + 16 * ks + 8 * h) * 2;
                        const h8 b = *reinterpret_cast<const h8*>(act + row * RS + col);
                        // lanes 0 / 32 fetch the pixel left of the tile, lanes 31 / 63 the one right of it, everyone else its own
                        const int erow = n == 0 ? row - 1 : (n == 31 ? row + 1 : row);
                        const h8 e = *reinterpret_cast<const h8*>(act + erow * RS + col);
                        const h8 bl = shift_lanes(b, e, true, lane);
                        const h8 br = shift_lanes(b, e, false, lane);
                        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(3 * dy) * 2 + ks], bl, acc[pt], 0, 0, 0);
                        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(3 * dy + 1) * 2 + ks], b, acc[pt], 0, 0, 0);
                        acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[(3 * dy + 2) * 2 + ks], br, acc[pt], 0, 0, 0);
                    }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 64 * NW + threadIdx.x] = s;
}

template <int MODE, int NW>
static void run(const char* what) {
    const int grid = 256, iters = 4000;
    float* out;
    hipMalloc(&out, (size_t)grid * 64 * NW * 4);
    const size_t ldsb = ROWS * RS + W_BYTES;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE, NW><<<grid, 64 * NW, ldsb>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, NW><<<grid, 64 * NW, ldsb>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double mfma = (double)grid * NW * iters * ((MODE >= 1 && NW == 8) ? 36.0 : 72.0);
    const double tf = mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("%-68s %d waves: %7.3f ms  %7.1f TFLOP/s  (%.1f %% of 2500)\n", what, NW, ms, tf, tf / 25.0);
    hipFree(out);
}

int main() {
    run<0, 4>("mode 0: W2 and activations from LDS (1.5 reads / MFMA)");
    run<0, 8>("mode 0: W2 and activations from LDS (1.5 reads / MFMA)");
    run<1, 4>("mode 1: W2 in registers, activations from LDS (1 read / MFMA)");
    run<1, 8>("mode 1: W2 in registers (k split over 8 waves), activations from LDS");
    run<2, 4>("mode 2: + dx shifts by DPP (2 reads + 16 VALU / 3 MFMAs)");
    run<2, 8>("mode 2: + dx shifts by DPP, k split over 8 waves");
    return 0;
}
