// Micro-benchmark: the conv2 inner loop of dense_layer_f16.hip in isolation - a chain of v_mfma_f32_32x32x16_f16 on ONE
// accumulator, each fed by two ds_read_b128 requested D MFMAs ahead - one wave per SIMD, and what changes its cycles per
// MFMA.   hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_chain.hip -o tools/ubench/build/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
template <int OFF>
__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// MODE bits: 1 = no A (activation) reads, 2 = no W reads, 4 = two accumulator chains (even / odd products),
//            32 / 64 = the MFMA as inline asm with its accumulator in arch / acc VGPRs,
//            8 = waves 4-7 stay resident and spin on s_sleep (a second, idle wave per SIMD), 16 = random-ish lane addresses for A
template <int MODE, int D, int NTHR>
__global__ __launch_bounds__(NTHR) void k(float* out, unsigned long long* cyc, int calls, float seed) {
    __shared__ __attribute__((aligned(16))) char lds[150 * 1024];
    for (int i = threadIdx.x; i < 150 * 256; i += blockDim.x) {
        _Float16 h4[2] = {(_Float16)(seed * ((i * 37) % 19 - 9)), (_Float16)(seed * ((i * 11) % 23 - 11))};
        ((unsigned*)lds)[i] = *(unsigned*)h4;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    if (wave >= 4) {
        if (MODE & 8) for (int q = 0; q < calls * 8; ++q) __builtin_amdgcn_s_sleep(32);
        return;
    }
    const unsigned lb = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < calls; ++c) {
        unsigned aA[3];
        for (int dx = 0; dx < 3; ++dx) {
            const int r = (i + dx + c) & 127;
            aA[dx] = lb + 73728 + (r >> 4) * 4096 + (r & 15) * 16 + h * 256 + wave * 32768 % 65536;
            if (MODE & 16) aA[dx] = lb + 73728 + ((r * 5) & 15) * 4096 + (r & 15) * 16 + h * 256;
        }
        const unsigned aW = lb + (c % 3) * 3 * 8192 + lane * 16;
        constexpr int NSL = D + 2, NE = 24;
        f32x4 ra[NSL] = {}, rw[NSL] = {};
        auto request = [&](auto e_c) {
            constexpr int e = decltype(e_c)::value, dxi = e / 8, ks = e % 8;
            if (!(MODE & 1)) ra[e % NSL] = lds_read4<ks * 512>(aA[dxi]);
            if (!(MODE & 2)) rw[e % NSL] = lds_read4<dxi * 8192 + ks * 1024>(aW);
        };
        static_for<0, D>(request);
        static_for<0, NE>([&](auto e_c) {
            constexpr int e = decltype(e_c)::value;
            if constexpr (e + D < NE) request(std::integral_constant<int, e + D>{});
            constexpr int per = ((MODE & 1) ? 0 : 1) + ((MODE & 2) ? 0 : 1);
            constexpr int younger = per * (e + D < NE ? D : NE - 1 - e);
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ra[e % NSL]), "+v"(rw[e % NSL]) : "n"(younger));
            if (MODE & 32)                                      // accumulator pinned in ARCH VGPRs
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(rw[e % NSL]), "v"(ra[e % NSL]));
            else if (MODE & 64)                                 // accumulator pinned in ACC VGPRs
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(rw[e % NSL]), "v"(ra[e % NSL]));
            else if ((MODE & 4) && (e & 1))
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[e % NSL]),
                                                              __builtin_bit_cast(half8, ra[e % NSL]), acc2, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, rw[e % NSL]),
                                                             __builtin_bit_cast(half8, ra[e % NSL]), acc, 0, 0, 0);
        });
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * NTHR + threadIdx.x] = s;
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int D, int NTHR>
void run(const char* name, float* out, unsigned long long* cyc, int calls) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, D, NTHR><<<256, NTHR>>>(out, cyc, calls, 0.01f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE, D, NTHR><<<256, NTHR>>>(out, cyc, calls, 0.01f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int b = 0; b < 256; ++b) s += h[b];
    const double per = s / 256 / (calls * 24.0);
    printf("%-44s %7.1f cycles/MFMA   %.3f ms/launch  clock %.2f GHz\n", name, per, ms / 5, s / 256 / (ms / 5 * 1e6));
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int calls = 4000;
    run<0, 6, 256>("D=6, A+W reads, 4 waves", out, cyc, calls);
    run<0, 6, 512>("D=6, A+W reads, 8 waves (4 exit)", out, cyc, calls);
    run<8, 6, 512>("D=6, A+W reads, 4 + 4 sleeping", out, cyc, calls);
    run<3, 6, 256>("no reads", out, cyc, calls);
    run<1, 6, 256>("W reads only", out, cyc, calls);
    run<2, 6, 256>("A reads only", out, cyc, calls);
    run<4, 6, 256>("A+W, two chains", out, cyc, calls);
    run<7, 6, 256>("no reads, two chains", out, cyc, calls);
    run<0, 3, 256>("D=3, A+W", out, cyc, calls);
    run<0, 4, 256>("D=4, A+W", out, cyc, calls);
    run<16, 6, 256>("D=6, A rows spread over 16 pixel groups", out, cyc, calls);
    run<32, 6, 256>("D=6, A+W, accumulator in arch VGPRs (asm)", out, cyc, calls);
    run<64, 6, 256>("D=6, A+W, accumulator in acc VGPRs (asm)", out, cyc, calls);
    run<32 | 3, 6, 256>("no reads, accumulator in arch VGPRs (asm)", out, cyc, calls);
    run<64 | 3, 6, 256>("no reads, accumulator in acc VGPRs (asm)", out, cyc, calls);
    return 0;
}
