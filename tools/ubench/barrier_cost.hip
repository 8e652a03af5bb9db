// Micro-benchmark: what one workgroup barrier costs 8 waves (512 threads, one workgroup per CU) and what the small things
// dense_layer_f16.hip does between two barriers add to it.   hipcc -w --offload-arch=gfx950 -O3 -std=c++17
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t(), 0, 0, 0)) u32x4;
// MODE bits: 1 = waves 0-3 issue 4 out-of-range buffer loads per round (never waited), 2 = waves 4-7 read 4 x 16 B of LDS and
// wait for them twice per round, 4 = waves 0-3 issue 4 IN-range loads of one 4 KB line set (L2 hits) and wait for the previous
// round's, 8 = s_waitcnt lgkmcnt(0) before the barrier (lds_barrier), 16 = waves 4-7 write 2 x ds_write_b128 twice per round
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, const float* src, int rounds) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 512) ((float*)lds)[i] = i;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
    u32x4 a[4] = {}, b[4] = {};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned lb = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < rounds; ++it) {
        if (wave < 4) {
            if (MODE & 1)
                for (int q = 0; q < 4; ++q) a[q] = __builtin_amdgcn_raw_buffer_load_b128(r, 0x7f000000 + lane * 16, 0, 0);
            if (MODE & 4) {
                for (int q = 0; q < 4; ++q) acc[q] += __builtin_bit_cast(f32x4, b[q])[0];
                for (int q = 0; q < 4; ++q)
                    b[q] = __builtin_amdgcn_raw_buffer_load_b128(r, ((it & 63) * 4 + q) * 1024 + lane * 16, wave * 262144, 0);
            }
        } else {
            for (int rep = 0; rep < 2; ++rep) {
                if (MODE & 2) {
                    f32x4 v0, v1, v2, v3;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(lb));
                    acc += v0 + v1 + v2 + v3;
                }
                if (MODE & 16) {
                    asm volatile("ds_write_b128 %0, %1 offset:32768\n\tds_write_b128 %0, %1 offset:33792" ::"v"(lb), "v"(acc) : "memory");
                }
            }
        }
        if (MODE & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int q = 0; q < 4; ++q) acc[q] += __builtin_bit_cast(f32x4, a[q])[0] + __builtin_bit_cast(f32x4, b[q])[1];
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc, const float* src) {
    const int rounds = 20000;
    k<MODE><<<256, 512>>>(out, cyc, src, rounds);
    hipDeviceSynchronize();
    k<MODE><<<256, 512>>>(out, cyc, src, rounds);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 256; ++b) s += h[b];
    printf("%-72s %7.1f cycles/round\n", name, s / 256 / rounds);
}
int main() {
    float *out, *src;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&src, 4 << 20);
    hipMemset(src, 0, 4 << 20);
    hipMalloc(&cyc, 256 * 8);
    run<0>("barrier only", out, cyc, src);
    run<8>("lgkmcnt(0) + barrier", out, cyc, src);
    run<1>("+ 4 out-of-range loads (waves 0-3)", out, cyc, src);
    run<4>("+ 4 L2-hit loads, waited one round later (waves 0-3)", out, cyc, src);
    run<2 | 8>("+ 2 x (4 LDS reads + wait) (waves 4-7)", out, cyc, src);
    run<16 | 8>("+ 2 x 2 ds_write_b128 (waves 4-7)", out, cyc, src);
    run<1 | 2 | 8 | 16>("out-of-range loads + LDS reads + writes", out, cyc, src);
    run<4 | 2 | 8 | 16>("L2 loads + LDS reads + writes", out, cyc, src);
    return 0;
}
