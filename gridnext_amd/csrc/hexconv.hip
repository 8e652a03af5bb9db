// Hexagonal 7-neighbour convolution for the corrector g (forward, data gradient, weight gradient).
//
// Replaces hexagdly.Conv2d(kernel_size=1, stride=1) as called from
// /root/reference/gridnext/gridnet_models.py:130-147, including the
// rot90/flip sandwich of :178-185 (mode 1 applies the stencil directly on the
// Visium odd-right grid, so no data is moved).
//
// Data layout in HBM: activations are channels-last [B][H][W][C] fp32 (a spot's
// channel vector is contiguous: f writes it, g reads it, the masked CE reads it).
// Parameters keep hexagdly's shapes: kernel0 [O][I][3][1], kernel1 [O][I][2][2],
// bias [O].  Tap t (0..6): t<3 -> kernel0[..][a=t]; t>=3 -> kernel1[..][a][b], a=(t-3)>>1, b=(t-3)&1.
//
// Geometry (P = parity axis, Q = run axis):
//   mode 0 (hexagdly addressing): P = W (column index), Q = H (row index)
//   mode 1 (Visium odd-right)   : P = H (row index),    Q = W (column index)
//   tap a of kernel0      : (dp, dq) = (0, a-1)
//   tap (a,b) of kernel1  : (dp, dq) = (2b-1, (p even ? -1 : 0) + a)
// Roofline: HBM/latency bound (7*I*O MAC per position on a 4992-position grid); see DESIGN.md.
#include "common.h"

namespace {

struct HexGeom {
    int B, H, W, mode;
    __device__ __forceinline__ void pq(int y, int x, int& p, int& q) const {
        if (mode) { p = y; q = x; } else { p = x; q = y; }
    }
    // flat position index of the tap-t neighbour of (y,x), or -1 if outside the grid
    __device__ __forceinline__ int nbr(int b, int y, int x, int t) const {
        int p, q;
        pq(y, x, p, q);
        int dp, dq;
        if (t < 3) { dp = 0; dq = t - 1; }
        else { const int a = (t - 3) >> 1, bb = (t - 3) & 1; dp = 2 * bb - 1; dq = ((p & 1) ? 0 : -1) + a; }
        const int np = p + dp, nq = q + dq;
        const int ny = mode ? np : nq, nx = mode ? nq : np;
        if (ny < 0 || ny >= H || nx < 0 || nx >= W) return -1;
        return (b * H + ny) * W + nx;
    }
    // position s whose tap-t neighbour is (y,x), or -1 (transpose of nbr)
    __device__ __forceinline__ int src(int b, int y, int x, int t) const {
        int p, q;
        pq(y, x, p, q);
        int sp, sq;
        if (t < 3) { sp = p; sq = q - (t - 1); }
        else {
            const int a = (t - 3) >> 1, bb = (t - 3) & 1;
            sp = p - (2 * bb - 1);
            sq = q - (((sp & 1) ? 0 : -1) + a);
        }
        const int sy = mode ? sp : sq, sx = mode ? sq : sp;
        if (sy < 0 || sy >= H || sx < 0 || sx >= W) return -1;
        return (b * H + sy) * W + sx;
    }
};

__device__ __forceinline__ float load_w(const float* k0, const float* k1, int I, int t, int o, int i) {
    return t < 3 ? k0[(o * I + i) * 3 + t] : k1[(o * I + i) * 4 + (t - 3)];
}

constexpr int POS_PER_BLOCK = 16;      // 312 workgroups on a 78 x 64 grid (64 per block left 178 CUs idle: 38 us per launch)

// y[pos][o] = bias[o] + sum_t sum_i W_t[o][i] * x[nbr_t(pos)][i]
__global__ __launch_bounds__(256) void hexconv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ k0, const float* __restrict__ k1,
    const float* __restrict__ bias, float* __restrict__ y, HexGeom g, int I, int O, int opad) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [7][I][O]
    for (int idx = threadIdx.x; idx < 7 * I * O; idx += blockDim.x) {
        const int o = idx % O, i = (idx / O) % I, t = idx / (O * I);
        wl[idx] = load_w(k0, k1, I, t, o, i);
    }
    __syncthreads();
    const int o = threadIdx.x % opad, pl0 = threadIdx.x / opad, pstep = blockDim.x / opad;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * POS_PER_BLOCK;
    if (o >= O) return;
    const bool vec = (I & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const float bo = bias ? bias[o] : 0.f;
    for (int pl = pl0; pl < POS_PER_BLOCK; pl += pstep) {
        const int pos = base + pl;
        if (pos >= npos) break;
        const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
        float acc = bo;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int n = g.nbr(b, yy, xx, t);
            if (n < 0) continue;
            const float* xr = x + (size_t)n * I;
            const float* wr = wl + t * I * O + o;
            if (vec) {                                      // 16-B loads of the neighbour's row, same order of the adds
                for (int i = 0; i < I; i += 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(xr + i);
                    acc = fmaf(wr[i * O], xv.x, acc);
                    acc = fmaf(wr[(i + 1) * O], xv.y, acc);
                    acc = fmaf(wr[(i + 2) * O], xv.z, acc);
                    acc = fmaf(wr[(i + 3) * O], xv.w, acc);
                }
            } else {
                for (int i = 0; i < I; ++i) acc = fmaf(wr[i * O], xr[i], acc);
            }
        }
        y[(size_t)pos * O + o] = acc;
    }
}

// dx[pos][i] = sum_t sum_o W_t[o][i] * dy[src_t(pos)][o]
__global__ __launch_bounds__(256) void hexconv_bwd_data_kernel(
    const float* __restrict__ dy, const float* __restrict__ k0, const float* __restrict__ k1,
    float* __restrict__ dx, HexGeom g, int I, int O, int ipad) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [7][O][I]
    for (int idx = threadIdx.x; idx < 7 * I * O; idx += blockDim.x) {
        const int i = idx % I, o = (idx / I) % O, t = idx / (O * I);
        wl[idx] = load_w(k0, k1, I, t, o, i);
    }
    __syncthreads();
    const int i = threadIdx.x % ipad, pl0 = threadIdx.x / ipad, pstep = blockDim.x / ipad;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * POS_PER_BLOCK;
    if (i >= I) return;
    const bool vec = (O & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    for (int pl = pl0; pl < POS_PER_BLOCK; pl += pstep) {
        const int pos = base + pl;
        if (pos >= npos) break;
        const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int s = g.src(b, yy, xx, t);
            if (s < 0) continue;
            const float* dr = dy + (size_t)s * O;
            const float* wr = wl + t * I * O + i;
            if (vec) {
                for (int o = 0; o < O; o += 4) {
                    const float4 dv = *reinterpret_cast<const float4*>(dr + o);
                    acc = fmaf(wr[o * I], dv.x, acc);
                    acc = fmaf(wr[(o + 1) * I], dv.y, acc);
                    acc = fmaf(wr[(o + 2) * I], dv.z, acc);
                    acc = fmaf(wr[(o + 3) * I], dv.w, acc);
                }
            } else {
                for (int o = 0; o < O; ++o) acc = fmaf(wr[o * I], dr[o], acc);
            }
        }
        dx[(size_t)pos * I + i] = acc;
    }
}

// partial[blk][t][o][i] = sum over the block's positions of dy[pos][o] * x[nbr_t(pos)][i];
// partial[blk][7*O*I + o] = sum dy[pos][o]
constexpr int WG_CHUNK = 16;       // positions staged per LDS pass
__global__ __launch_bounds__(256) void hexconv_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    HexGeom g, int I, int O, int pos_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dys = lds;                          // [WG_CHUNK][O]
    float* xs = lds + WG_CHUNK * O;            // [WG_CHUNK][7][I]
    __shared__ int nbrs[WG_CHUNK * 7];
    const int nout = 7 * O * I + O;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * pos_per_block;
    constexpr int MAXACC = 32;                 // covers 7*64*64/1024... sized for I,O<=32 at 256 threads
    float acc[MAXACC];
#pragma unroll
    for (int j = 0; j < MAXACC; ++j) acc[j] = 0.f;
    for (int c0 = 0; c0 < pos_per_block; c0 += WG_CHUNK) {
        const int cnt = pos_per_block - c0 < WG_CHUNK ? pos_per_block - c0 : WG_CHUNK;      // positions of this pass
        __syncthreads();
        for (int idx = threadIdx.x; idx < cnt * O; idx += blockDim.x) {
            const int pos = base + c0 + idx / O;
            dys[idx] = pos < npos ? dy[(size_t)pos * O + idx % O] : 0.f;
        }
        // neighbour rows once per (position, tap) - not once per staged element - then row copies
        if (threadIdx.x < cnt * 7) {
            const int t = threadIdx.x % 7, pl = threadIdx.x / 7;
            const int pos = base + c0 + pl;
            int n = -1;
            if (pos < npos) {
                const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
                n = g.nbr(b, yy, xx, t);
            }
            nbrs[threadIdx.x] = n;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < cnt * 7 * I; idx += blockDim.x) {
            const int q = idx / I, i = idx - q * I;           // q = pl * 7 + t
            const int n = nbrs[q];
            xs[idx] = n >= 0 ? x[(size_t)n * I + i] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXACC; ++j) {
            const int out = threadIdx.x + j * 256;
            float a = acc[j];
            if (out < 7 * O * I) {
                const int i = out % I, o = (out / I) % O, t = out / (O * I);
                for (int pl = 0; pl < cnt; ++pl) a = fmaf(dys[pl * O + o], xs[(pl * 7 + t) * I + i], a);
            } else if (out < nout) {
                const int o = out - 7 * O * I;
                for (int pl = 0; pl < cnt; ++pl) a += dys[pl * O + o];
            }
            acc[j] = a;
        }
    }
#pragma unroll
    for (int j = 0; j < MAXACC; ++j) {
        const int out = threadIdx.x + j * 256;
        if (out < nout) partial[(size_t)blockIdx.x * nout + out] = acc[j];
    }
}

// fixed-order sum of the partial slabs, scattered into hexagdly's parameter layouts (accumulating or not)
__global__ void hexconv_reduce_weight_kernel(const float* __restrict__ partial, int nblk, int I, int O,
                                             float* __restrict__ dk0, float* __restrict__ dk1,
                                             float* __restrict__ dbias, int accumulate) {
    const int nout = 7 * O * I + O;
    const int out = blockIdx.x * blockDim.x + threadIdx.x;
    if (out >= nout) return;
    // slabs added in index order, their loads issued 16 at a time (a plain loop over ~1000 slabs is one memory round trip
    // per term: 74 us for 29 MB)
    float s = 0.f;
    int b = 0;
    for (; b + 16 <= nblk; b += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = partial[(size_t)(b + u) * nout + out];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; b < nblk; ++b) s += partial[(size_t)b * nout + out];
    float* dst;
    if (out < 7 * O * I) {
        const int i = out % I, o = (out / I) % O, t = out / (O * I);
        dst = t < 3 ? dk0 + (o * I + i) * 3 + t : dk1 + (o * I + i) * 4 + (t - 3);
    } else {
        if (!dbias) return;
        dst = dbias + (out - 7 * O * I);
    }
    *dst = accumulate ? *dst + s : s;
}

int pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

GNX_EXPORT int gnx_hexconv_fwd(const float* x, const float* kernel0, const float* kernel1, const float* bias,
                               float* y, int B, int H, int W, int I, int O, int mode, hipStream_t stream) {
    if (!x || !kernel0 || !kernel1 || !y || B < 0 || H <= 0 || W <= 0 || I <= 0 || O <= 0 || I > 64 || O > 64)
        return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    if (npos == 0) return GNX_OK;
    HexGeom g{B, H, W, mode};
    hexconv_fwd_kernel<<<gnx_cdiv(npos, POS_PER_BLOCK), 256, 7 * I * O * sizeof(float), stream>>>(
        x, kernel0, kernel1, bias, y, g, I, O, pow2_at_least(O));
    return gnx_launch_status();
}

GNX_EXPORT int gnx_hexconv_bwd_data(const float* dy, const float* kernel0, const float* kernel1, float* dx,
                                    int B, int H, int W, int I, int O, int mode, hipStream_t stream) {
    if (!dy || !kernel0 || !kernel1 || !dx || B < 0 || H <= 0 || W <= 0 || I <= 0 || O <= 0 || I > 64 || O > 64)
        return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    if (npos == 0) return GNX_OK;
    HexGeom g{B, H, W, mode};
    hexconv_bwd_data_kernel<<<gnx_cdiv(npos, POS_PER_BLOCK), 256, 7 * I * O * sizeof(float), stream>>>(
        dy, kernel0, kernel1, dx, g, I, O, pow2_at_least(I));
    return gnx_launch_status();
}

// positions per workgroup of the weight-gradient kernel: aim at ~1024 workgroups (one 78 x 64 array at 128 positions
// per workgroup was 39 workgroups on a 256-CU chip, 240 us per layer), whole LDS passes, at most 128
static int hex_ppb(long npos) {
    long ppb = (npos + 1023) / 1024;
    ppb = (ppb + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
    if (ppb < WG_CHUNK) ppb = WG_CHUNK;
    if (ppb > 128) ppb = 128;
    return (int)ppb;
}

// workspace floats needed by gnx_hexconv_bwd_weight
GNX_EXPORT long gnx_hexconv_bwd_weight_workspace(int B, int H, int W, int I, int O) {
    const long npos = (long)B * H * W;
    return (long)gnx_cdiv(npos, hex_ppb(npos)) * (7L * O * I + O);
}

GNX_EXPORT int gnx_hexconv_bwd_weight(const float* x, const float* dy, float* dkernel0, float* dkernel1,
                                      float* dbias, float* workspace, int B, int H, int W, int I, int O,
                                      int mode, int accumulate, hipStream_t stream) {
    if (!x || !dy || !dkernel0 || !dkernel1 || !workspace || I <= 0 || O <= 0 || I > 32 || O > 32 ||
        H <= 0 || W <= 0 || B < 0)
        return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    const int ppb = hex_ppb(npos);
    const int nblk = gnx_cdiv(npos, ppb);
    const int nout = 7 * O * I + O;
    HexGeom g{B, H, W, mode};
    if (nblk > 0) {
        const size_t lds = (size_t)(WG_CHUNK * O + WG_CHUNK * 7 * I) * sizeof(float);
        hexconv_bwd_weight_kernel<<<nblk, 256, lds, stream>>>(x, dy, workspace, g, I, O, ppb);
    }
    hexconv_reduce_weight_kernel<<<gnx_cdiv(nout, 256), 256, 0, stream>>>(workspace, nblk, I, O, dkernel0,
                                                                          dkernel1, dbias, accumulate);
    return gnx_launch_status();
}
