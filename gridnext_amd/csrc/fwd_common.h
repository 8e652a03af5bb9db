// Helpers shared by the forward DenseNet kernels (conv1x1.hip, conv3x3.hip, stem_pool.hip): activation, 16-B LDS/global
// access, the LDS-only barrier, inline-asm fragment reads with immediate offsets, compile-time loops.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {


constexpr int LDK = 36;     // floats per LDS row: 32 + 4 pad

__device__ __forceinline__ float act1(float v, float sc, float sh) { return fmaxf(fmaf(v, sc, sh), 0.f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed form: (x,y) and (z,w) stay in their even-aligned register pairs (v_pk_fma_f32 / v_pk_max_f32), so the
// compiler has no reason to shuffle freshly loaded registers (which would force a wait right behind the loads)
__device__ __forceinline__ float4 act4(float4 v, float4 sc, float4 sh) {
    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const f32x2 slo = {sc.x, sc.y}, shi = {sc.z, sc.w}, tlo = {sh.x, sh.y}, thi = {sh.z, sh.w}, zero = {0.f, 0.f};
    lo = __builtin_elementwise_max(__builtin_elementwise_fma(lo, slo, tlo), zero);
    hi = __builtin_elementwise_max(__builtin_elementwise_fma(hi, shi, thi), zero);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence whose wait is vmcnt(0):
// it would drain the global prefetch loads that are meant to stay in flight across the barrier (measured: the
// ping-pong memory phase took 12.5k cycles instead of ~2k with __syncthreads()).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// bounds-checked / unaligned-safe 4-float load of src[0..3], zero beyond `valid` elements
__device__ __forceinline__ float4 ld4_safe(const float* p, int valid, bool vec) {
    if (vec && valid >= 4) return ld4(p);
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid > 0) r.x = p[0];
    if (valid > 1) r.y = p[1];
    if (valid > 2) r.z = p[2];
    if (valid > 3) r.w = p[3];
    return r;
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read4(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// 1 KB of LDS-DMA (16 B per lane, LDS destination lds_dst + 16 lane) through a BUFFER resource: source = resource base +
// per-lane offset voff + scalar offset soff.  Beside saturated MFMA waves the global_load_lds form of the same transfer costs
// the matrix pipe ~40 cycles per wave-instruction, the buffer form nothing (tools/ubench/mfma_2x2.hip: 16 per chunk and
// wave -> 129 against 150 TFLOP/s); bytes past the resource's extent read as zero.
__device__ __forceinline__ void dma16_buf(const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_dst), "v"(voff), "s"(r), "s"(soff) : "memory");
}
// The same transfer with per-lane 64-bit addresses (array ends, where rows are clamped per lane).  M0 is written inside the
// asm statements only: no kernel mixes these helpers with the compiler's own LDS-DMA builtin, which keeps M0 itself.
__device__ __forceinline__ void dma16_global(const float* lane_src, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_dst), "v"(lane_src) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
