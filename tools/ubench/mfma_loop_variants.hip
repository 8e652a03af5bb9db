// Micro-benchmark: which construct of the LDS-DMA conv3x3 inner loop costs matrix-pipe time?
//   hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_loop_variants.hip -o tools/ubench/build/mfma_var
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

// MODE 0: compiler reads, padded rows.  1: asm reads + counted waits, padded rows.  2: asm reads, 128-B swizzled rows.
// 3: as 2 plus per-step address select against a zero row.  4: as 3 with 36-step unrolled chunks + barrier per chunk.
// 5: as 4 but masked lanes are zeroed AFTER the read (4 v_cndmask per step).  6: as 4 with a runtime buffer base.
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, unsigned maskseed) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = seed + i * 1e-6f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
    const unsigned base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)lds;
    unsigned aoff[4], boff[4];
    for (int s = 0; s < 4; ++s) {
        if (MODE <= 1) {
            aoff[s] = 4 * ((i + 32 * wave) * 36 + 4 * h + 8 * s);
            boff[s] = 4 * (4608 + i * 36 + 4 * h + 8 * s);
        } else {
            const int R = i + 32 * wave + 5, RB = 200 + i;
            aoff[s] = 4 * (R * 32 + 4 * ((2 * s + h) ^ ((R >> 1) & 7)));
            boff[s] = 4 * (RB * 32 + 4 * ((2 * s + h) ^ ((RB >> 1) & 7)));
        }
    }
    const unsigned mask = maskseed | (lane == 0 ? 0u : 0xffffffffu);       // lane 0 masked on some steps
    const unsigned zb = base + 4 * 16000;
    if (MODE == 0) {
        const float* ap = lds + aoff[0] / 4;
        const float* bp = lds + boff[0] / 4;
        float4 a = *(const float4*)ap, b = *(const float4*)bp;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float4 na = *(const float4*)(ap + 8 * ((s + 1) & 3));
                float4 nb = *(const float4*)(bp + 8 * ((s + 1) & 3));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc1, 0, 0, 0);
                a = na; b = nb;
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
    } else {
        constexpr int STEPS = MODE >= 4 ? 36 : 8;
        for (int it = 0; it < iters; ++it) {
            if (MODE >= 4) asm volatile("s_barrier" ::: "memory");
            const unsigned rb = MODE == 6 ? base + 64u * (it & 1) : base;
            auto aaddr = [&](int e) {
                const unsigned o = rb + aoff[e & 3] + (MODE >= 2 ? 128u * ((e >> 2) % 3) : 0u);
                if (MODE >= 3 && MODE != 5) return ((mask >> (e >> 2)) & 1u) ? o : zb;
                return o;
            };
            auto baddr = [&](int e) { return rb + boff[e & 3] + (MODE >= 4 ? 4096u * (e >> 2) % 16384u : 0u); };
            f32x4 a = lds_read4(aaddr(0)), b = lds_read4(baddr(0));
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                f32x4 na, nb;
                if (s < STEPS - 1) {
                    na = lds_read4(aaddr(s + 1));
                    nb = lds_read4(baddr(s + 1));
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
                }
                if (MODE == 5 && !((mask >> (s >> 2)) & 1u)) a = f32x4{0.f, 0.f, 0.f, 0.f};
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc1, 0, 0, 0);
                if (s < STEPS - 1) { a = na; b = nb; }
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int iters) {
    const int grid = 256 * blocks_per_cu;
    constexpr int STEPS = MODE >= 4 ? 36 : 8;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    k<MODE><<<grid, 256>>>(out, 10, 0.5f, 0x155u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<MODE><<<grid, 256>>>(out, iters, 0.5f, 0x155u);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    double flops = (double)grid * 4 * iters * (STEPS * 4) * 4096.0;
    printf("%-52s blocks/CU %d : %7.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<0>("0 compiler reads, padded rows", 1, 4000);
    run<1>("1 asm reads + counted lgkmcnt, padded rows", 1, 4000);
    run<2>("2 asm reads, swizzled 128-B rows", 1, 4000);
    run<3>("3 + address select vs zero row", 1, 4000);
    run<4>("4 + 36-step chunks, barrier per chunk", 1, 900);
    run<5>("5 zero after read (4 cndmask/step), 36-step chunks", 1, 900);
    run<6>("6 runtime buffer base, 36-step chunks", 1, 900);
    run<4>("4 again", 1, 900);
    run<1>("1 asm reads + counted lgkmcnt, padded rows", 2, 4000);
    run<3>("3 + address select vs zero row", 2, 4000);
    return 0;
}
