// Micro-benchmark: what v_mfma_f32_32x32x2_f32 sustains on gfx950 in the regimes our conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4608 * 2];
    for (int i = threadIdx.x; i < 4608 * 2; i += blockDim.x) lds[i] = seed + i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* ap = &lds[((lane & 31) + 32 * wave) * 36 + 4 * (lane >> 5)];
    const float* bp = &lds[4608 + (lane & 31) * 36 + 4 * (lane >> 5)];
    float4 a = *(const float4*)ap, b = *(const float4*)bp;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float4 na = a, nb = b;
            if (LDSREAD) {
                na = *(const float4*)(ap + 8 * ((s + 1) & 3));
                nb = *(const float4*)(bp + 8 * ((s + 1) & 3));
            }
            acc[0 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[0 % NACC], 0, 0, 0);
            acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[1 % NACC], 0, 0, 0);
            acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[2 % NACC], 0, 0, 0);
            acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[3 % NACC], 0, 0, 0);
            a = na; b = nb;
            if (LDSREAD) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
    }
    float s = 0.f;
    for (int a2 = 0; a2 < NACC; ++a2)
        for (int r = 0; r < 16; ++r) s += acc[a2][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDSREAD>
void run(const char* name, int blocks_per_cu, int iters = 2000) {
    const int grid = 256 * blocks_per_cu;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<NACC, LDSREAD><<<grid, 256>>>(out, 10, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    k<NACC, LDSREAD><<<grid, 256>>>(out, iters, 0.5f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)grid * 4 /*waves*/ * iters * 32 /*mfma per iter*/ * 4096.0;
    printf("%-40s blocks/CU %d : %7.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1, false>("1 acc (dependent chain), regs only", 1);
    run<2, false>("2 acc, regs only", 1);
    run<4, false>("4 acc, regs only", 1);
    run<1, false>("1 acc, regs only", 2);
    run<2, false>("2 acc, regs only", 2);
    run<4, false>("4 acc, regs only", 2);
    run<2, true>("2 acc + 2 ds_read_b128 per 4 mfma", 1);
    run<2, true>("2 acc + 2 ds_read_b128 per 4 mfma", 2);
    run<4, true>("4 acc + 2 ds_read_b128 per 4 mfma", 2);
    run<4, true>("4 acc + 2 ds_read_b128 per 4 mfma", 3);
    // sustained: does the clock hold over 20..200 ms of back-to-back matrix work?
    run<2, true>("sustained 2 acc + ds_read", 1, 20000);
    run<2, true>("sustained 2 acc + ds_read", 1, 200000);
    run<2, true>("sustained 2 acc + ds_read", 2, 100000);
    run<4, false>("sustained 4 acc regs only", 1, 200000);
    return 0;
}
