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

// Weight gradient: hexconv_bwd_weight2_kernel below (its first matrix-core form - 64 positions per workgroup, 7 neighbour rows
// of every position staged element by element, a slab per wave - was removed in round 3).
constexpr int HW_POS = 64;         // (workspace sizing keeps the first version's slab count: an upper bound)


// ---- round 2: forward and data gradient on the matrix cores.  The scalar kernels above run 7 * I FMAs per output out of LDS
// (17 us per launch on a 78 x 64 grid - five forward and four data-gradient launches were a sixth of a count-only step).
// Both are, per tap, a [positions x CK] x [CK x CN] product (forward: contraction CK = input channels, data gradient:
// CK = output channels and the neighbour map transposed): a workgroup takes 32 positions x 32 output columns, its four waves
// split the seven taps (0,4 | 1,5 | 2,6 | 3), each wave multiplying with v_mfma_f32_32x32x2_f32 straight from global memory
// - a lane loads 16 B of its position's neighbour row (channels 8 g + 4 h ..) and the four matching weights (L2) - and the
// four partial 32 x 32 tiles are summed through LDS in a fixed order.  No operand staging, one barrier.  NG = CK / 8.
template <int NG, bool BWD>
__global__ __launch_bounds__(256) void hexconv_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ k0, const float* __restrict__ k1,
    const float* __restrict__ bias, float* __restrict__ out, HexGeom g, int IF, int OF) {
    __shared__ float red[4][16][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, i = lane & 31;
    const int npos = g.B * g.H * g.W;
    const int base = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int CN = BWD ? IF : OF, LDI = BWD ? OF : IF, LDO = BWD ? IF : OF;      // output columns; row strides
    const int pos = base + i;
    const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
    const int col = n0 + i;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tap = wave + 4 * tt;
        if (tap >= 7) break;                                   // wave-uniform
        const int nb = pos < npos ? (BWD ? g.src(b, yy, xx, tap) : g.nbr(b, yy, xx, tap)) : -1;
        float4 a[NG];
        float w[NG][4];
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            a[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nb >= 0) a[q] = *reinterpret_cast<const float4*>(in + (size_t)nb * LDI + 8 * q + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 8 * q + 4 * h + c;               // contraction channel
                w[q][c] = col < CN ? (BWD ? load_w(k0, k1, IF, tap, k, col) : load_w(k0, k1, IF, tap, col, k)) : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, w[q][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, w[q][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, w[q][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, w[q][3], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const float bo = (!BWD && bias && col < CN) ? bias[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave + 4 * j;
        const float v = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
        const int p = base + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (p < npos && col < CN) out[(size_t)p * LDO + col] = v + bo;
    }
}

// ---- round 2, second form of the weight gradient: no staging at all.  A workgroup takes 32 positions; wave w owns taps
// w and w + 4 for ALL of them (so the waves' results are disjoint: one slab per workgroup, a quarter of the slab traffic
// and of the reduce) and feeds the MFMAs straight from global memory - lane (h, c): dy[position 2 pp + h][o = c] and
// x[neighbour_t(position)][ch = c], both 128-B coalesced rows.  (The first form staged 64 x 7 neighbour rows element by
// element, a division and a neighbour computation per float: 35 us per launch.)
constexpr int HW2_POS = 32;
// (bx: the block's index in ONE layer's grid - blockIdx.x, or derived from the flat block id of the batched launch)
__device__ __forceinline__ void hexconv_bwd_weight2_body(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    const HexGeom& g, int I, int O, int IF, int OF, int i0, int o0, int bx) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, c = lane & 31;
    const int npos = g.B * g.H * g.W;
    const int base = bx * HW2_POS;
    const int ta = wave, tb = wave + 4;                         // tb == 7: this wave has one tap only
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float bsum = 0.f;
#pragma unroll                                                  // (all 16 position pairs' loads in flight: one round trip, not four)
    for (int pp = 0; pp < HW2_POS / 2; ++pp) {
        const int pos = base + 2 * pp + h;
        float a = 0.f, b0 = 0.f, b1 = 0.f;
        if (pos < npos) {
            const int xx = pos % g.W, yy = (pos / g.W) % g.H, b = pos / (g.W * g.H);
            if (c < O) a = dy[(size_t)pos * OF + o0 + c];
            if (c < I) {
                const int na = g.nbr(b, yy, xx, ta);
                if (na >= 0) b0 = x[(size_t)na * IF + i0 + c];
                if (tb < 7) {
                    const int nb = g.nbr(b, yy, xx, tb);
                    if (nb >= 0) b1 = x[(size_t)nb * IF + i0 + c];
                }
            }
        }
        bsum += a;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
        if (tb < 7) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    const int nout = 7 * O * I + O;
    float* dst = partial + (size_t)bx * nout;
    if (c < I) {                                               // D[row = o][col = ch = c]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (o < O) {
                dst[(ta * O + o) * I + c] = acc0[r];
                if (tb < 7) dst[(tb * O + o) * I + c] = acc1[r];
            }
        }
    }
    if (wave == 0) {
        bsum += __shfl_xor(bsum, 32, 64);                      // the two position halves of output channel c
        if (h == 0 && c < O) dst[7 * O * I + c] = bsum;
    }
}
__global__ __launch_bounds__(256) void hexconv_bwd_weight2_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
    HexGeom g, int I, int O, int IF, int OF, int i0, int o0) {
    hexconv_bwd_weight2_body(x, dy, partial, g, I, O, IF, OF, i0, o0, blockIdx.x);
}

// ---- the weight gradients of SEVERAL hex layers as one launch (gnx_hexconv_bwd_weight_batch): the five layers of the
// corrector g are independent once their output gradients exist, 156 workgroups of 32 positions each, and cost a launch
// (12 us + 8 us of reduce) apiece - a quarter of a count-only training step.  Their grids are laid end to end (entry e owns
// flat block ids [first_block, first_block + nblk)); kernel body, slabs and the fixed-order reduce are the single launches'.
struct HexWEntry {
    const float* x; const float* dy; float* partial; float* dk0; float* dk1; float* dbias;
    HexGeom g;
    int I, O, nblk, first_block, accumulate;
};
constexpr int HEXW_BATCH = 8;
struct HexWBatch { HexWEntry e[HEXW_BATCH]; int n; };
__global__ __launch_bounds__(256) void hexconv_bwd_weight2_batch_kernel(const HexWBatch b) {
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.e[k + 1].first_block) ++k;      // (uniform: at most 8 entries)
    const HexWEntry& e = b.e[k];
    hexconv_bwd_weight2_body(e.x, e.dy, e.partial, e.g, e.I, e.O, e.I, e.O, 0, 0, blockIdx.x - e.first_block);
}

// fixed-order sum of the partial slabs, scattered into hexagdly's parameter layouts (accumulating or not)
__device__ __forceinline__ void hexconv_reduce_weight_body(const float* __restrict__ partial, int nblk, int I, int O,
                                                           float* __restrict__ dk0, float* __restrict__ dk1,
                                                           float* __restrict__ dbias, int accumulate, int IF, int i0, int o0,
                                                           int out) {
    const int nout = 7 * O * I + O;
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
__global__ void hexconv_reduce_weight_kernel(const float* __restrict__ partial, int nblk, int I, int O,
                                             float* __restrict__ dk0, float* __restrict__ dk1,
                                             float* __restrict__ dbias, int accumulate, int IF, int i0, int o0) {
    hexconv_reduce_weight_body(partial, nblk, I, O, dk0, dk1, dbias, accumulate, IF, i0, o0, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void hexconv_reduce_weight_batch_kernel(const HexWBatch b) {        // grid (blocks of 64 outputs, entries)
    const HexWEntry& e = b.e[blockIdx.y];
    hexconv_reduce_weight_body(e.partial, e.nblk, e.I, e.O, e.dk0, e.dk1, e.dbias, e.accumulate, e.I, 0, 0,
                               blockIdx.x * blockDim.x + threadIdx.x);
}

int pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }

constexpr int HEX_CHUNK = 64;      // channels per launch, forward and data gradient (7 * 64 * 64 weights = 112 KB of LDS)
constexpr int HEX_WCHUNK = 32;     // channels per launch, weight gradient (one 32 x 32 MFMA tile per tap)

// the MFMA forward / data-gradient form takes contraction widths 8, 16, 32, 64 on 16-B aligned rows (other widths:
// the scalar kernels)
bool hex_mfma_ok(int CK, const float* in, long npos) {
    return (CK == 8 || CK == 16 || CK == 32 || CK == 64) && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
           npos < (1L << 30);
}

}  // namespace

GNX_EXPORT int gnx_hexconv_fwd(const float* x, const float* kernel0, const float* kernel1, const float* bias,
                               float* y, int B, int H, int W, int I, int O, int mode, hipStream_t stream) {
    if (!x || !kernel0 || !kernel1 || !y || B < 0 || H <= 0 || W <= 0 || I <= 0 || O <= 0) return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    if (npos == 0) return GNX_OK;
    HexGeom g{B, H, W, mode};
    if (hex_mfma_ok(I, x, npos)) {
        dim3 grid(gnx_cdiv(npos, 32), gnx_cdiv(O, 32));
        if (I == 8) hexconv_mfma_kernel<1, false><<<grid, 256, 0, stream>>>(x, kernel0, kernel1, bias, y, g, I, O);
        else if (I == 16) hexconv_mfma_kernel<2, false><<<grid, 256, 0, stream>>>(x, kernel0, kernel1, bias, y, g, I, O);
        else if (I == 32) hexconv_mfma_kernel<4, false><<<grid, 256, 0, stream>>>(x, kernel0, kernel1, bias, y, g, I, O);
        else hexconv_mfma_kernel<8, false><<<grid, 256, 0, stream>>>(x, kernel0, kernel1, bias, y, g, I, O);
        return gnx_launch_status();
    }
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
    if (hex_mfma_ok(O, dy, npos)) {
        dim3 grid(gnx_cdiv(npos, 32), gnx_cdiv(I, 32));
        if (O == 8) hexconv_mfma_kernel<1, true><<<grid, 256, 0, stream>>>(dy, kernel0, kernel1, nullptr, dx, g, I, O);
        else if (O == 16) hexconv_mfma_kernel<2, true><<<grid, 256, 0, stream>>>(dy, kernel0, kernel1, nullptr, dx, g, I, O);
        else if (O == 32) hexconv_mfma_kernel<4, true><<<grid, 256, 0, stream>>>(dy, kernel0, kernel1, nullptr, dx, g, I, O);
        else hexconv_mfma_kernel<8, true><<<grid, 256, 0, stream>>>(dy, kernel0, kernel1, nullptr, dx, g, I, O);
        return gnx_launch_status();
    }
    for (int i0 = 0; i0 < I; i0 += HEX_CHUNK)
        for (int o0 = 0; o0 < O; o0 += HEX_CHUNK) {
            const int oc = O - o0 < HEX_CHUNK ? O - o0 : HEX_CHUNK, ic = I - i0 < HEX_CHUNK ? I - i0 : HEX_CHUNK;
            hexconv_bwd_data_kernel<<<gnx_cdiv(npos, POS_PER_BLOCK), 256, 7 * ic * oc * sizeof(float), stream>>>(
                dy, kernel0, kernel1, dx, g, ic, oc, pow2_at_least(ic), I, O, i0, o0);
        }
    return gnx_launch_status();
}

// workspace floats needed by gnx_hexconv_bwd_weight
GNX_EXPORT long gnx_hexconv_bwd_weight_workspace(int B, int H, int W, int I, int O) {
    const long npos = (long)B * H * W;
    const long ic = I < HEX_WCHUNK ? I : HEX_WCHUNK, oc = O < HEX_WCHUNK ? O : HEX_WCHUNK;     // one chunk pair at a time
    return 4L * gnx_cdiv(npos, HW_POS) * (7L * oc * ic + oc);                                // one slab per wave
}

GNX_EXPORT int gnx_hexconv_bwd_weight(const float* x, const float* dy, float* dkernel0, float* dkernel1,
                                      float* dbias, float* workspace, int B, int H, int W, int I, int O,
                                      int mode, int accumulate, hipStream_t stream) {
    if (!x || !dy || !dkernel0 || !dkernel1 || !workspace || I <= 0 || O <= 0 || H <= 0 || W <= 0 || B < 0)
        return GNX_ERR_BAD_ARG;
    const long npos = (long)B * H * W;
    const int nblk = gnx_cdiv(npos, HW2_POS);
    HexGeom g{B, H, W, mode};
    // one (input chunk, output chunk) pair after the other on the stream, each through the same workspace
    for (int o0 = 0; o0 < O; o0 += HEX_WCHUNK)
        for (int i0 = 0; i0 < I; i0 += HEX_WCHUNK) {
            const int oc = O - o0 < HEX_WCHUNK ? O - o0 : HEX_WCHUNK, ic = I - i0 < HEX_WCHUNK ? I - i0 : HEX_WCHUNK;
            const int nout = 7 * oc * ic + oc;
            if (nblk > 0) {
                hexconv_bwd_weight2_kernel<<<nblk, 256, 0, stream>>>(x, dy, workspace, g, ic, oc, I, O, i0, o0);
            }
            hexconv_reduce_weight_kernel<<<gnx_cdiv(nout, 64), 64, 0, stream>>>(workspace, nblk, ic, oc, dkernel0,
                                                                                dkernel1, dbias, accumulate, I, i0, o0);
        }
    return gnx_launch_status();
}

// n hex layers' weight gradients as ONE launch plus one batched reduce: exactly gnx_hexconv_bwd_weight for every item, bit for
// bit (same kernel body, slab layout and reduce order).  Layers wider than 32 channels on either side (which the single entry
// point cuts into chunk pairs) or more than 8 items: GNX_ERR_UNSUPPORTED, nothing launched.  `items`: HOST array of
// gnx_hexconv_wgrad_item (include/gridnext_hip.h); each item's workspace: gnx_hexconv_bwd_weight_workspace floats.
struct GnxHexWgradItem {
    const float* x; const float* dy; float* dkernel0; float* dkernel1; float* dbias; float* workspace;
    int B, H, W, I, O, mode, accumulate, pad;
};
GNX_EXPORT int gnx_hexconv_bwd_weight_batch(const void* items_v, int n, hipStream_t stream) {
    const GnxHexWgradItem* it = static_cast<const GnxHexWgradItem*>(items_v);
    if (!it || n < 0) return GNX_ERR_BAD_ARG;
    if (n == 0) return GNX_OK;
    if (n > HEXW_BATCH) return GNX_ERR_UNSUPPORTED;
    HexWBatch b;
    b.n = n;
    int blocks = 0, max_nout = 0;
    for (int q = 0; q < n; ++q) {
        const GnxHexWgradItem& a = it[q];
        if (!a.x || !a.dy || !a.dkernel0 || !a.dkernel1 || !a.workspace || a.I <= 0 || a.O <= 0 || a.H <= 0 || a.W <= 0 || a.B < 0)
            return GNX_ERR_BAD_ARG;
        if (a.I > HEX_WCHUNK || a.O > HEX_WCHUNK) return GNX_ERR_UNSUPPORTED;
        const long npos = (long)a.B * a.H * a.W;
        if (npos <= 0 || npos >= (1L << 30)) return GNX_ERR_UNSUPPORTED;
        HexWEntry& e = b.e[q];
        e.x = a.x; e.dy = a.dy; e.partial = a.workspace; e.dk0 = a.dkernel0; e.dk1 = a.dkernel1; e.dbias = a.dbias;
        e.g = HexGeom{a.B, a.H, a.W, a.mode};
        e.I = a.I; e.O = a.O; e.nblk = gnx_cdiv(npos, HW2_POS); e.first_block = blocks; e.accumulate = a.accumulate;
        blocks += e.nblk;
        const int nout = 7 * a.O * a.I + a.O;
        if (nout > max_nout) max_nout = nout;
    }
    hexconv_bwd_weight2_batch_kernel<<<blocks, 256, 0, stream>>>(b);
    hexconv_reduce_weight_batch_kernel<<<dim3(gnx_cdiv(max_nout, 64), n), 64, 0, stream>>>(b);
    return gnx_launch_status();
}
