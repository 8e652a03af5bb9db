// Shared helpers for the gfx950 kernels of the GridNext f∘g hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GNX_OK 0
#define GNX_ERR_BAD_ARG (-1)
#define GNX_ERR_LAUNCH (-2)
#define GNX_ERR_UNSUPPORTED (-3)

#define GNX_EXPORT extern "C" __attribute__((visibility("default")))

static inline int gnx_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GNX_OK : GNX_ERR_LAUNCH;
}

static inline int gnx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront reductions (shuffles, no LDS)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024); result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* lds /* >= 16 floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) lds[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += lds[i];   // fixed order: deterministic
    return r;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
