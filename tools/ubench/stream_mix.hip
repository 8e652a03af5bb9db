// Micro-benchmark: what HBM sustains on gfx950 for the READ:WRITE mixes our memory-bound kernels have, with no arithmetic in
// the way - the ceiling `roofline.frac` of those kernels should be read against (the 8 TB/s peak is a read-only figure).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_mix.hip -o tools/ubench/build/stream_mix && tools/ubench/build/stream_mix
// Each workgroup streams whole "rows": RD 16-B pieces read and WR 16-B pieces written per row (row strides = exactly those,
// i.e. dense), grid-stride over rows, 256 threads; the written value depends on everything read.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int RD, int WR>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long rows) {
    // 256 threads take 256 / G rows at a time, G = lanes per row
    constexpr int G = RD >= WR ? (RD >= 16 ? 16 : RD) : (WR >= 16 ? 16 : WR);
    constexpr int RPB = 256 / G;
    const int g = threadIdx.x % G, rl = threadIdx.x / G;
    for (long r0 = (long)blockIdx.x * RPB; r0 < rows; r0 += (long)gridDim.x * RPB) {
        const long r = r0 + rl;
        if (r >= rows) continue;
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = g; j < RD; j += G) {
            const uint4 v = in[r * RD + j];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {                  // every lane's reads reach every written piece
            acc.x ^= __shfl_xor(acc.x, o, 64); acc.y ^= __shfl_xor(acc.y, o, 64);
            acc.z ^= __shfl_xor(acc.z, o, 64); acc.w ^= __shfl_xor(acc.w, o, 64);
        }
#pragma unroll
        for (int j = g; j < WR; j += G) out[r * WR + j] = acc;
        if (WR == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[0] = acc;      // keeps the reads alive
    }
}

__global__ void fill_kernel(uint4* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        p[i] = make_uint4(h, h * 0x9e3779b9u, h ^ 0x85ebca6bu, h * 0xc2b2ae35u + 1u);
    }
}

template <int RD, int WR>
void run(const char* what, long rows, int wgs) {
    uint4 *in = nullptr, *out = nullptr;
    hipMalloc(&in, (size_t)rows * (RD ? RD : 1) * 16);
    hipMalloc(&out, (size_t)rows * (WR ? WR : 1) * 16);
    fill_kernel<<<4096, 256>>>(in, rows * (RD ? RD : 1));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    mix_kernel<RD, WR><<<wgs, 256>>>(in, out, rows);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) mix_kernel<RD, WR><<<wgs, 256>>>(in, out, rows);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)rows * (RD + WR) * 16;
    printf("%-58s rows %9ld  wgs %5d  %7.3f ms  %5.2f TB/s (read %4.0f %%)\n", what, rows, wgs, ms, bytes / ms / 1e9,
           100.0 * RD / (RD + WR));
    hipFree(in); hipFree(out);
}

int main(int argc, char** argv) {
    const long rows = 5111808;                    // block 1 of a 128-px array
    for (int wgs : {1024, 4096}) {
        run<28, 0>("read only, 448 B rows", rows, wgs);
        run<0, 16>("write only, 256 B rows", rows, wgs);
        run<16, 16>("copy 256 B -> 256 B", rows, wgs);
        run<28, 16>("conv1 fp16 K=224: read 448 B, write 256 B", rows, wgs);
        run<8, 16>("conv1 fp16 K=64: read 128 B, write 256 B", rows, wgs);
        run<62, 16>("conv1 fp16 K=496: read 992 B, write 256 B", rows / 4, wgs);
        run<32, 4>("conv2 fp16: read 512 B, write 64 B", rows, wgs);
        run<32, 32>("fp32 pass: read 512 B, write 512 B", rows, wgs);
        run<64, 32>("read X + read dX, write dX (conv1 dgrad epilogue mix)", rows / 2, wgs);
    }
    return 0;
}
