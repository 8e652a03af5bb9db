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
//
// Channel counts: any I, O >= 1.  One launch handles up to 64 x 64 (forward, data gradient) or 32 x 32 (weight gradient)
// channels - the corrector's own shapes (f_dim <= 64 -> 32 -> n_classes) are single launches; wider layers (many classes,
// classify=False feature inputs) are tiled over channel chunks by the entry points, the later input chunks accumulating
// onto the earlier ones' result.
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
// One (input chunk, output chunk) pair of channels: I, O are the chunk's sizes, i0 / o0 its first channels, IF / OF the
// layer's full channel counts (= the row strides of x, y and of the parameter tensors).
__global__ __launch_bounds__(256) void hexconv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ k0, const float* __restrict__ k1,
    const float* __restrict__ bias, float* __restrict__ y, HexGeom g, int I, int O, int opad,
    int IF, int OF, int i0, int o0) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [7][I][O]
    for (int idx = threadIdx.x; idx < 7 * I * O; idx += blockDim.x) {
        const int o = idx % O, i = (idx / O) % I, t = idx / (O * I);
        wl[idx] = load_w(k0, k1, IF, t, o0 + o, i0 + i);
    }
    __syncthreads();
    const int o = threadIdx.x % opad, pl0 = threadIdx.x / opad, pstep = blockDim.x / opad;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * POS_PER_BLOCK;
    if (o >= O) return;
    const bool vec = ((I | IF | i0) & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const float bo = bias ? bias[o0 + o] : 0.f;
    for (int pl = pl0; pl < POS_PER_BLOCK; pl += pstep) {
        const int pos = base + pl;
        if (pos >= npos) break;
        const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
        float acc = i0 == 0 ? bo : y[(size_t)pos * OF + o0 + o];      // later input chunks continue the sum
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int n = g.nbr(b, yy, xx, t);
            if (n < 0) continue;
            const float* xr = x + (size_t)n * IF + i0;
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
        y[(size_t)pos * OF + o0 + o] = acc;
    }
}

// dx[pos][i] = sum_t sum_o W_t[o][i] * dy[src_t(pos)][o]
__global__ __launch_bounds__(256) void hexconv_bwd_data_kernel(
    const float* __restrict__ dy, const float* __restrict__ k0, const float* __restrict__ k1,
    float* __restrict__ dx, HexGeom g, int I, int O, int ipad, int IF, int OF, int i0, int o0) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [7][O][I]
    for (int idx = threadIdx.x; idx < 7 * I * O; idx += blockDim.x) {
        const int i = idx % I, o = (idx / I) % O, t = idx / (O * I);
        wl[idx] = load_w(k0, k1, IF, t, o0 + o, i0 + i);
    }
    __syncthreads();
    const int i = threadIdx.x % ipad, pl0 = threadIdx.x / ipad, pstep = blockDim.x / ipad;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * POS_PER_BLOCK;
    if (i >= I) return;
    const bool vec = ((O | OF | o0) & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    for (int pl = pl0; pl < POS_PER_BLOCK; pl += pstep) {
        const int pos = base + pl;
        if (pos >= npos) break;
        const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
        float acc = o0 == 0 ? 0.f : dx[(size_t)pos * IF + i0 + i];    // later output chunks continue the sum
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int s = g.src(b, yy, xx, t);
            if (s < 0) continue;
            const float* dr = dy + (size_t)s * OF + o0;
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
        dx[(size_t)pos * IF + i0 + i] = acc;
    }
}

// partial[blk][t][o][i] = sum over the block's positions of dy[pos][o] * x[nbr_t(pos)][i];
// partial[blk][7*O*I + o] = sum dy[pos][o]
constexpr int WG_CHUNK = 16;       // positions staged per LDS pass
__global__ __launch_bounds__(256) void hexconv_bwd_weight_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    HexGeom g, int I, int O, int pos_per_block, int IF, int OF, int i0, int o0) {
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
            dys[idx] = pos < npos ? dy[(size_t)pos * OF + o0 + idx % O] : 0.f;
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
            xs[idx] = n >= 0 ? x[(size_t)n * IF + i0 + i] : 0.f;
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
                                             float* __restrict__ dbias, int accumulate, int IF, int i0, int o0) {
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
        const int oi = (o0 + o) * IF + i0 + i;
        dst = t < 3 ? dk0 + oi * 3 + t : dk1 + oi * 4 + (t - 3);
    } else {
        if (!dbias || i0 != 0) return;                  // the bias gradient comes from the first input chunk's launches
        dst = dbias + o0 + (out - 7 * O * I);
    }
    *dst = accumulate ? *dst + s : s;
}

int pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }

constexpr int HEX_CHUNK = 64;      // channels per launch, forward and data gradient (7 * 64 * 64 weights = 112 KB of LDS)
constexpr int HEX_WCHUNK = 32;     // channels per launch, weight gradient (MAXACC accumulators per thread)

}  // namespace

GNX_EXPORT int gnx_hexconv_fwd(const float* x, const float* kernel0, const float* kernel1, const float* bias,
                               float* y, int B, int H, int W, int I, int O, int mode, hipStream_t stream) {
    if (!x || !kernel0 || !kernel1 || !y || B < 0 || H <= 0 || W <= 0 || I <= 0 || O <= 0) return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    if (npos == 0) return GNX_OK;
    HexGeom g{B, H, W, mode};
    for (int o0 = 0; o0 < O; o0 += HEX_CHUNK)
        for (int i0 = 0; i0 < I; i0 += HEX_CHUNK) {         // input chunks in order: chunk k continues chunk k-1's sums
            const int oc = O - o0 < HEX_CHUNK ? O - o0 : HEX_CHUNK, ic = I - i0 < HEX_CHUNK ? I - i0 : HEX_CHUNK;
            hexconv_fwd_kernel<<<gnx_cdiv(npos, POS_PER_BLOCK), 256, 7 * ic * oc * sizeof(float), stream>>>(
                x, kernel0, kernel1, bias, y, g, ic, oc, pow2_at_least(oc), I, O, i0, o0);
        }
    return gnx_launch_status();
}

GNX_EXPORT int gnx_hexconv_bwd_data(const float* dy, const float* kernel0, const float* kernel1, float* dx,
                                    int B, int H, int W, int I, int O, int mode, hipStream_t stream) {
    if (!dy || !kernel0 || !kernel1 || !dx || B < 0 || H <= 0 || W <= 0 || I <= 0 || O <= 0) return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    if (npos == 0) return GNX_OK;
    HexGeom g{B, H, W, mode};
    for (int i0 = 0; i0 < I; i0 += HEX_CHUNK)
        for (int o0 = 0; o0 < O; o0 += HEX_CHUNK) {
            const int oc = O - o0 < HEX_CHUNK ? O - o0 : HEX_CHUNK, ic = I - i0 < HEX_CHUNK ? I - i0 : HEX_CHUNK;
            hexconv_bwd_data_kernel<<<gnx_cdiv(npos, POS_PER_BLOCK), 256, 7 * ic * oc * sizeof(float), stream>>>(
                dy, kernel0, kernel1, dx, g, ic, oc, pow2_at_least(ic), I, O, i0, o0);
        }
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
    const long ic = I < HEX_WCHUNK ? I : HEX_WCHUNK, oc = O < HEX_WCHUNK ? O : HEX_WCHUNK;     // one chunk pair at a time
    return (long)gnx_cdiv(npos, hex_ppb(npos)) * (7L * oc * ic + oc);
}

GNX_EXPORT int gnx_hexconv_bwd_weight(const float* x, const float* dy, float* dkernel0, float* dkernel1,
                                      float* dbias, float* workspace, int B, int H, int W, int I, int O,
                                      int mode, int accumulate, hipStream_t stream) {
    if (!x || !dy || !dkernel0 || !dkernel1 || !workspace || I <= 0 || O <= 0 || H <= 0 || W <= 0 || B < 0)
        return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    const int ppb = hex_ppb(npos);
    const int nblk = gnx_cdiv(npos, ppb);
    HexGeom g{B, H, W, mode};
    // one (input chunk, output chunk) pair after the other on the stream, each through the same workspace
    for (int o0 = 0; o0 < O; o0 += HEX_WCHUNK)
        for (int i0 = 0; i0 < I; i0 += HEX_WCHUNK) {
            const int oc = O - o0 < HEX_WCHUNK ? O - o0 : HEX_WCHUNK, ic = I - i0 < HEX_WCHUNK ? I - i0 : HEX_WCHUNK;
            const int nout = 7 * oc * ic + oc;
            if (nblk > 0) {
                const size_t lds = (size_t)(WG_CHUNK * oc + WG_CHUNK * 7 * ic) * sizeof(float);
                hexconv_bwd_weight_kernel<<<nblk, 256, lds, stream>>>(x, dy, workspace, g, ic, oc, ppb, I, O, i0, o0);
            }
            hexconv_reduce_weight_kernel<<<gnx_cdiv(nout, 256), 256, 0, stream>>>(workspace, nblk, ic, oc, dkernel0, dkernel1,
                                                                                  dbias, accumulate, I, i0, o0);
        }
    return gnx_launch_status();
}
