// Micro-benchmark: the conv1x1 consumer pattern (2 x 2 fragments of 32x32, 4 ds_read_b128 + 16 MFMA per sub-step) in
// isolation: no staging, no barrier unless asked.   hipcc -w --offload-arch=gfx950 -O3 tools/ubench/mfma_2x2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
static int g_scale = 1;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF, int MIN>
__device__ __forceinline__ void lds_fmax(unsigned addr, float v) {          // LDS[addr + OFF] = max/min(LDS[..], v)
    if (MIN) asm volatile("ds_min_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
    else asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int I, int N, int BOTH>
__device__ __forceinline__ void lds_fmax_seq(unsigned addr, const float* v) {
    if constexpr (I < N) {
        lds_fmax<I * 256, 0>(addr, v[I & 15]);
        if constexpr (BOTH) lds_fmax<I * 256, 1>(addr, v[(I + 7) & 15]);
        lds_fmax_seq<I + 1, N, BOTH>(addr, v);
    }
}
// MODE 0: as the kernel (a0 feeds two consecutive MFMAs).  1: order changed so consecutive MFMAs share no source
// register.  2: mode 0 + s_barrier per chunk.
// PW (producer work per chunk, waves 4-7): bit 0 = 64 v_fma_f32, bit 1 = 8 ds_write_b128, bit 2 = 16 global_load_dwordx4,
// bit 3 = run the producer at s_setprio 3, bits 4.. = load variants (see the code)
template <int MODE, int NTHR, int PW = 0>
__global__ __launch_bounds__(NTHR) void k(float* out, int iters, float seed, const float* src = nullptr) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = seed + i * 1e-6f;
    __syncthreads();
    if (threadIdx.x >= 256) {                       // "producer" waves: synthetic staging work + the barrier
        if (MODE != 2) return;
        if (PW & 8) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = seed * j;
        float4 g[16];
        for (int j = 0; j < 16; ++j) g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* sp = reinterpret_cast<const float4*>(src) + (size_t)blockIdx.x * 4096 + (threadIdx.x - 256);   // max index 511*4096 + 255 + 15*256 < 512*4096
        char* wp = lds + 32768 + (threadIdx.x - 256) * 16;
        for (int it = 0; it < iters; ++it) {
            if (PW & 4) {
#pragma unroll
                for (int j = 0; j < 16; ++j) g[j] = sp[((it * 16 + j) & 15) * 256];
            }
            if (PW & 16) {          // 16 LDS-DMA loads (1 KB each) into the unused half of the LDS
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(sp + ((it * 16 + j) & 15) * 256),
                                                     (float*)(lds + 32768 + ((threadIdx.x - 256) >> 6) * 4096 + (j & 3) * 1024),
                                                     16, 0, 0);
            }
            if (PW & 128) {         // 16 dwordx4 loads, scalar base + 32-bit lane offset (saddr form)
                const char* sb = reinterpret_cast<const char*>(src) + (size_t)blockIdx.x * 65536;
                const unsigned vo = (threadIdx.x - 256) * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    g[j] = *reinterpret_cast<const float4*>(sb + ((it * 16 + j) & 15) * 4096 + vo);
            }
            if (PW & 256) {         // 16 buffer_load_dwordx4
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(reinterpret_cast<const char*>(src)) + (size_t)blockIdx.x * 65536, 0, 65536, 0x00020000);
                const int vo = (threadIdx.x - 256) * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, ((it * 16 + j) & 15) * 4096, 0);
                    g[j] = make_float4(__int_as_float(r[0]), __int_as_float(r[1]), __int_as_float(r[2]), __int_as_float(r[3]));
                }
            }
            if (PW & 2048) {        // 16 buffer_load_dwordx4 ... lds (DMA through a buffer resource)
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(reinterpret_cast<const char*>(src)) + (size_t)blockIdx.x * 65536, 0, 65536, 0x00020000);
                const unsigned vo = (threadIdx.x - 256) * 16;
                const unsigned m0b = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + 32768 + ((threadIdx.x - 256) >> 6) * 4096);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 ::"s"(m0b + (j & 3) * 1024), "v"(vo), "s"(rsrc), "s"(((it * 16 + j) & 15) * 4096) : "memory", "m0");
            }
            if (PW & 512) {         // 16 dwordx2 loads
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float2 v2 = reinterpret_cast<const float2*>(sp)[((it * 16 + j) & 15) * 512];
                    g[j].x = v2.x; g[j].y = v2.y;
                }
            }
            if (PW & 1024) {        // 16 dwordx4 loads that hit the L1 (4 KB footprint per wave)
#pragma unroll
                for (int j = 0; j < 16; ++j) g[j] = sp[(j & 3) * 256];
            }
            if (PW & 32) {          // 16 dword (4 B / lane) loads
#pragma unroll
                for (int j = 0; j < 16; ++j) g[j].x = reinterpret_cast<const float*>(sp)[((it * 16 + j) & 15) * 1024];
            }
            if (PW & 64) {          // 16 ds_read_b128 (LDS -> VGPR) by the producers
#pragma unroll
                for (int j = 0; j < 16; ++j) g[j] = *reinterpret_cast<const float4*>(wp + (j & 3) * 4096);
            }
            if (PW & 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], 1.0001f, seed);
            }
            if (PW & 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<float4*>(wp + ((it & 1) ? 0 : 4096) + j * 4096 % 16384) =
                        make_float4(v[j] + g[j].x, v[j + 1] + g[j + 8].y, v[j + 2], v[j + 3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
        }
        float sacc = 0.f;
        for (int j = 0; j < 16; ++j) sacc += v[j] + g[j].x + g[j].w;
        if (sacc == 12345.678f) out[0] = sacc;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lb = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds;
    const unsigned fA = lb + (4 * wm + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    const unsigned fB = lb + 16384 + (4 * wn + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    f32x16 acc00, acc01, acc10, acc11;
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) asm volatile("s_barrier" ::: "memory");
        const unsigned a = fA + (it & 1) * 32768, b = fB + (it & 1) * 32768;
        f32x4 a0 = lds_read4<0>(a), a1 = lds_read4<4096>(a), b0 = lds_read4<0>(b), b1 = lds_read4<4096>(b);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 na0, na1, nb0, nb1;
            if (s < 3) {
                na0 = lds_read4<512>(a + 512 * s); na1 = lds_read4<512 + 4096>(a + 512 * s);
                nb0 = lds_read4<512>(b + 512 * s); nb1 = lds_read4<512 + 4096>(b + 512 * s);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 1) {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc00, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc11, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc10, 0, 0, 0);
                } else {
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc11, 0, 0, 0);
                }
            }
            if (s < 3) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc00[r] + acc01[r] + acc10[r] + acc11[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- How long does a producer wave's own instruction stream take next to two saturated MFMA waves per SIMD?
// OP: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_max_f32, 3 v_max_i32 (integer ReLU on float bits), 4 v_and_b32, 5 v_med3_f32,
// 6 ds_read_b128, 7 ds_max_f32 (LDS float atomic), 8 ds_max_f32 + ds_min_f32, 9 = 16 of each.  Each producer wave times 64 dependent-free instructions of that kind per chunk with s_memtime.
// MF: 0 = consumers use v_mfma_f32_32x32x2_f32 (64 cycles each), 1 = v_mfma_f32_16x16x4_f32 (32 cycles each, twice as many)
template <int OP, int MF = 0>
__global__ __launch_bounds__(512) void klat(float* out, long* cyc, int iters, float seed, const float* src = nullptr, long* cyc2 = nullptr) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)lds)[i] = seed + i * 1e-6f;
    __syncthreads();
    if (threadIdx.x >= 256) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = seed * (j + 1);
        long tot = 0, tot2 = 0;
        for (int it = 0; it < iters; ++it) {
            const long t0 = __builtin_amdgcn_s_memtime();
            if (OP >= 10 && OP <= 13) {
                // VMEM issue time: 16 loads of 1 KB per wave-instruction, streaming (each chunk a fresh 16 KB per wave)
                // 10 global_load_dwordx4, 11 global_load_lds_dwordx4 (DMA), 12 global_load_dword (256 B), 13 dwordx4 from L2
                const long slot = ((long)(blockIdx.x * 4 + ((threadIdx.x - 256) >> 6)) * iters + (OP == 13 ? 0 : it)) & 131071;
                const float* p = src + slot * 4096 + (threadIdx.x & 63) * 4;
                f32x4 g[16];
                char* d = lds + ((threadIdx.x - 256) >> 6) * 16384;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (OP == 10 || OP == 13) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g[j]) : "v"(p + j * 256) : "memory");
                    if (OP == 12) asm volatile("global_load_dword %0, %1, off" : "=v"(g[j][0]) : "v"(p + j * 256) : "memory");
                    if (OP == 11) __builtin_amdgcn_global_load_lds(p + j * 256, (float*)(d + j * 1024), 16, 0, 0);
                }
                asm volatile("s_nop 0" ::: "memory");
                const long t1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tot2 += __builtin_amdgcn_s_memtime() - t1;
                if (OP != 11) for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(g[j]));
                tot += t1 - t0;
                asm volatile("s_barrier" ::: "memory");
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0001f), "v"(seed));
                    if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(seed));
                    if (OP == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v[j]) : "v"(0));
                    if (OP == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[j]) : "v"(0x7fffffff));
                    if (OP == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(-1e30f), "v"(1e30f));
                }
            if (OP == 1) {
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f2 x = {v[2 * j], v[2 * j + 1]};
                        const f2 a = {1.0001f, 1.0001f}, b = {seed, seed};
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                        v[2 * j] = x[0]; v[2 * j + 1] = x[1];
                    }
            }
            if (OP == 6) {
                f32x4 g[16];
                const unsigned a = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + (threadIdx.x - 256) * 16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) g[j] = lds_read4<0>(a + (j & 3) * 4096);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                v[0] += g[0][0] + g[15][3];
            }
            if (OP == 7 || OP == 8 || OP == 9) {          // LDS float atomics: 64 x (256 B per wave-instruction, conflict-free)
                const unsigned a = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds +
                                   ((threadIdx.x - 256) >> 6) * 16384 + (threadIdx.x & 63) * 4;
                if (OP == 9) { lds_fmax_seq<0, 16, 1>(a, v); }     // what a chunk needs: 16 max + 16 min
                else lds_fmax_seq<0, 64, OP == 8>(a, v);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("s_nop 0" ::: "memory");
            tot += __builtin_amdgcn_s_memtime() - t0;
            asm volatile("s_barrier" ::: "memory");
        }
        float sacc = 0.f;
        for (int j = 0; j < 16; ++j) sacc += v[j];
        if (sacc == 12345.678f) out[0] = sacc;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + ((threadIdx.x - 256) >> 6)] = tot;
        if (cyc2 && (threadIdx.x & 63) == 0) cyc2[blockIdx.x * 4 + ((threadIdx.x - 256) >> 6)] = tot2;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, i = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lb = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds;
    const unsigned fA = lb + (4 * wm + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    const unsigned fB = lb + 16384 + (4 * wn + (i >> 4)) * 2048 + (i & 15) * 16 + h * 256;
    if (MF == 1) {
        // 64 x 64 wave tile out of 16 x 16 x 4 MFMAs: 4 A + 4 B fragments (16 rows x 16 k per ds_read_b128), 16 accumulators
        f32x4 acc[4][4];
        for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned gA = lb + (4 * wm) * 2048 + (lane & 15) * 16 + (lane >> 4) * 256;
        const unsigned gB = lb + 16384 + (4 * wn) * 2048 + (lane & 15) * 16 + (lane >> 4) * 256;
        for (int it = 0; it < iters; ++it) {
            asm volatile("s_barrier" ::: "memory");
            const unsigned a = gA + (it & 1) * 32768, b = gB + (it & 1) * 32768;
#pragma unroll
            for (int s = 0; s < 2; ++s) {                  // two 16-k halves of the 32-k chunk
                f32x4 fa[4], fb[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) { fa[x] = lds_read4<0>(a + x * 2048 + s * 1024); fb[x] = lds_read4<0>(b + x * 2048 + s * 1024); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                            acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[x][c], fb[y][c], acc[x][y], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        return;
    }
    f32x16 acc00, acc01, acc10, acc11;
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_barrier" ::: "memory");
        const unsigned a = fA + (it & 1) * 32768, b = fB + (it & 1) * 32768;
        f32x4 a0 = lds_read4<0>(a), a1 = lds_read4<4096>(a), b0 = lds_read4<0>(b), b1 = lds_read4<4096>(b);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 na0, na1, nb0, nb1;
            if (s < 3) {
                na0 = lds_read4<512>(a + 512 * s); na1 = lds_read4<512 + 4096>(a + 512 * s);
                nb0 = lds_read4<512>(b + 512 * s); nb1 = lds_read4<512 + 4096>(b + 512 * s);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b0[c], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c], b1[c], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b0[c], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c], b1[c], acc11, 0, 0, 0);
            }
            if (s < 3) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc00[r] + acc01[r] + acc10[r] + acc11[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void runvmem(const char* name) {
    const int grid = 512, iters = 2000;
    float* out; long* cyc; long* cyc2;
    static float* src = nullptr;
    if (!src) { (void)hipMalloc(&src, 131072ull * 16384 + 65536); (void)hipMemset(src, 0, 131072ull * 16384 + 65536); }
    (void)hipMalloc(&out, sizeof(float) * grid * 256);
    (void)hipMalloc(&cyc, sizeof(long) * grid * 4);
    (void)hipMalloc(&cyc2, sizeof(long) * grid * 4);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    klat<OP, 0><<<grid, 512>>>(out, cyc, 10, 0.5f, src, cyc2);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    klat<OP, 0><<<grid, 512>>>(out, cyc, iters, 0.5f, src, cyc2);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    static long h[2048], h2[2048];
    (void)hipMemcpy(h, cyc, sizeof(long) * grid * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2, cyc2, sizeof(long) * grid * 4, hipMemcpyDeviceToHost);
    double tot = 0, tot2 = 0; for (int q = 0; q < grid * 4; ++q) { tot += h[q]; tot2 += h2[q]; }
    double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-40s : MFMA waves %7.1f TFLOP/s; producer: %6.0f cycles to issue 16 (%5.1f each), then %6.0f until landed\n", name,
           flops / ms / 1e9, tot / (grid * 4) / iters, tot / (grid * 4) / iters / 16, tot2 / (grid * 4) / iters);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(cyc2);
}
template <int OP, int MF = 0>
void runlat(const char* name) {
    const int grid = 512, iters = 4000;
    float* out; long* cyc;
    (void)hipMalloc(&out, sizeof(float) * grid * 256);
    (void)hipMalloc(&cyc, sizeof(long) * grid * 4);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    klat<OP, MF><<<grid, 512>>>(out, cyc, 10, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    klat<OP, MF><<<grid, 512>>>(out, cyc, iters, 0.5f);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    static long h[2048];
    (void)hipMemcpy(h, cyc, sizeof(long) * grid * 4, hipMemcpyDeviceToHost);
    double tot = 0; for (int q = 0; q < grid * 4; ++q) tot += h[q];
    double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-40s : MFMA waves %7.1f TFLOP/s; producer: %6.0f cycles per 64 instructions (%5.1f each)\n", name,
           flops / ms / 1e9, tot / (grid * 4) / iters, tot / (grid * 4) / iters / 64);
    (void)hipFree(out); (void)hipFree(cyc);
}
template <int MODE, int NTHR, int PW = 0>
void run(const char* name, int bpc, int iters) {
    iters *= g_scale;
    const int grid = 256 * bpc;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    static float* src = nullptr;
    if (!src) { (void)hipMalloc(&src, 512ull * 4096 * 16); (void)hipMemset(src, 0, 512ull * 4096 * 16); }
    k<MODE, NTHR, PW><<<grid, NTHR>>>(out, 10, 0.5f, src);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    k<MODE, NTHR, PW><<<grid, NTHR>>>(out, iters, 0.5f, src);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    double flops = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-56s WG/CU %d : %7.3f ms  %7.1f TFLOP/s\n", name, bpc, ms, flops / ms / 1e9);
    fflush(stdout);
    (void)hipFree(out);
}
int main(int argc, char** argv) {
    if (argc > 1) g_scale = atoi(argv[1]);
    run<0, 256>("warm-up", 2, 20000);
    if (argc > 2) goto lat;
    run<0, 256>("0 kernel order (a0 feeds 2 consecutive MFMAs)", 1, 10000);
    run<0, 256>("0 kernel order", 2, 10000);
    run<1, 256>("1 no shared source between consecutive MFMAs", 1, 10000);
    run<1, 256>("1 no shared source", 2, 10000);
    run<2, 256>("2 + barrier per chunk, 4 waves", 2, 10000);
    run<2, 512>("2 + barrier per chunk, 4 + 4 idle producer waves", 2, 10000);
    run<2, 512, 1>("producers: 64 v_fma / chunk", 2, 10000);
    run<2, 512, 9>("producers: 64 v_fma / chunk, setprio 3", 2, 10000);
    run<2, 512, 2>("producers: 8 ds_write_b128 / chunk", 2, 10000);
    run<2, 512, 4>("producers: 16 global_load_dwordx4 / chunk", 2, 10000);
    run<2, 512, 16>("producers: 16 global_load_lds_dwordx4 (DMA) / chunk", 2, 10000);
    run<2, 512, 32>("producers: 16 global_load_dword / chunk", 2, 10000);
    run<2, 512, 64>("producers: 16 ds_read_b128 / chunk", 2, 10000);
    run<2, 512, 128>("producers: 16 dwordx4, saddr + voffset", 2, 10000);
    run<2, 512, 256>("producers: 16 buffer_load_dwordx4", 2, 10000);
    run<2, 512, 2048>("producers: 16 buffer_load_dwordx4 lds (DMA)", 2, 10000);
    run<2, 512, 512>("producers: 16 global_load_dwordx2", 2, 10000);
    run<2, 512, 1024>("producers: 16 dwordx4 hitting L1", 2, 10000);
    run<2, 512, 7>("producers: all three", 2, 10000);
    run<2, 512, 15>("producers: all three, setprio 3", 2, 10000);
lat:
    runlat<0>("64 v_fma_f32 / chunk");
    runlat<1>("64 v_pk_fma_f32 / chunk");
    runlat<2>("64 v_max_f32 / chunk");
    runlat<3>("64 v_max_i32 / chunk");
    runlat<4>("64 v_and_b32 / chunk");
    runlat<5>("64 v_med3_f32 / chunk");
    runlat<6>("64 ds_read_b128 / chunk");
    runlat<7>("64 ds_max_f32 / chunk");
    runlat<8>("64 ds_max_f32 + 64 ds_min_f32 / chunk");
    runlat<9>("16 ds_max_f32 + 16 ds_min_f32 / chunk");
    runvmem<10>("16 global_load_dwordx4 (HBM) / chunk");
    runvmem<11>("16 global_load_lds_dwordx4 (HBM) / chunk");
    runvmem<12>("16 global_load_dword (HBM) / chunk");
    runvmem<13>("16 global_load_dwordx4 (L2) / chunk");
    if (argc > 2) return 0;
    runlat<0, 1>("16x16x4 consumers: 64 v_fma_f32 / chunk");
    runlat<1, 1>("16x16x4 consumers: 64 v_pk_fma_f32 / chunk");
    runlat<2, 1>("16x16x4 consumers: 64 v_max_f32 / chunk");
    runlat<6, 1>("16x16x4 consumers: 64 ds_read_b128 / chunk");
    return 0;
}
