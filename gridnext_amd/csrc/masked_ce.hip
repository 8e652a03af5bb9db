// Foreground-masked softmax cross-entropy over a grid of logits, fused with argmax/accuracy counting.
//
// Replaces the permute -> reshape -> boolean-mask gather -> labels-1 -> nn.CrossEntropyLoss() -> torch.max
// sequence of /root/reference/gridnext/training.py:152-160 and :176-177 (and the same masking in
// gridnext/utils.py:37-47).  Logits are channels-last rows z[M][C] (M = B*H*W, row stride ld);
// labels int64[M], 0 = background, 1..C = classes.
//   loss      = mean_{label>0}( logsumexp(z) - z[label-1] ) / accum_iters
//   dz        = (softmax(z) - onehot) * dloss / (n_fg * accum_iters) on foreground rows, 0 elsewhere
// label_base = 1 is the grid loop (0 = background); label_base = 0 is the plain CE of train_spotwise
// (training.py:61-62: every row counts, labels 0..C-1).
//   stats     = {n_fg, n_correct} (argmax = first maximal index, as torch.max)
// No dynamic shapes, no host sync: n_fg stays on the device.  Fixed-order two-stage reduction (deterministic).
// HBM-bound: (4*C + 8) B per spot forward.  Any C >= 1 (one thread walks a row's C logits; the path's C is 5..20).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void masked_ce_fwd_kernel(const float* __restrict__ z, long ld,
                                                            const long long* __restrict__ labels, long M, int C,
                                                            int label_base,
                                                            double* __restrict__ partial /* [nblk][3] */,
                                                            long long* __restrict__ preds) {
    __shared__ double red[3][4];
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double loss = 0.0, nfg = 0.0, ncorrect = 0.0;
    if (r < M) {
        const long long lab = labels[r];
        const float* zr = z + r * ld;
        float mx = zr[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = zr[c];
            if (v > mx) { mx = v; arg = c; }
        }
        if (preds) preds[r] = arg;
        if (lab >= label_base) {
            const long long cls = lab - label_base;
            float s = 0.f;
            for (int c = 0; c < C; ++c) s += expf(zr[c] - mx);
            const float lse = mx + logf(s);
            // a class index past C is a caller error (torch raises); poison the loss instead of reading out of bounds
            loss = cls < C ? (double)(lse - zr[cls]) : (double)__builtin_nanf("");
            nfg = 1.0;
            ncorrect = (arg == (int)cls) ? 1.0 : 0.0;
        }
    }
    loss = wave_sum_d(loss);
    nfg = wave_sum_d(nfg);
    ncorrect = wave_sum_d(ncorrect);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = loss; red[1][wid] = nfg; red[2][wid] = ncorrect; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        partial[(size_t)blockIdx.x * 3 + k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    }
}

__global__ void masked_ce_finalize_kernel(const double* __restrict__ partial, int nblk, float accum_iters,
                                          float* __restrict__ loss, long long* __restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0, nfg = 0.0, nc = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s += partial[(size_t)b * 3 + 0];
        nfg += partial[(size_t)b * 3 + 1];
        nc += partial[(size_t)b * 3 + 2];
    }
    // mean over an empty selection is NaN in torch (0/0); keep that behaviour
    *loss = (float)((s / nfg) / (double)accum_iters);
    stats[0] = (long long)nfg;
    stats[1] = (long long)nc;
}

__global__ __launch_bounds__(256) void masked_ce_bwd_kernel(const float* __restrict__ z, long ld,
                                                            const long long* __restrict__ labels, long M, int C,
                                                            int label_base,
                                                            const long long* __restrict__ stats,
                                                            const float* __restrict__ dloss, float accum_iters,
                                                            float* __restrict__ dz, long lddz) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const long long lab = labels[r];
    float* dr = dz + r * lddz;
    if (lab < label_base) {
        for (int c = 0; c < C; ++c) dr[c] = 0.f;
        return;
    }
    const float* zr = z + r * ld;
    float mx = zr[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, zr[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(zr[c] - mx);
    const float g = (dloss ? *dloss : 1.f) / ((float)stats[0] * accum_iters);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) {
        float p = expf(zr[c] - mx) * inv;
        if (c == (int)(lab - label_base)) p -= 1.f;
        dr[c] = p * g;
    }
}

// probs[r][:] = softmax(z[r][:]); preds[r] = first argmax   (evaluation: utils.py:43-47)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ z, long ld, long M, int C,
                                                           float* __restrict__ probs, long ldp,
                                                           long long* __restrict__ preds) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= M) return;
    const float* zr = z + r * ld;
    float mx = zr[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = zr[c];
        if (v > mx) { mx = v; arg = c; }
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(zr[c] - mx);
    const float inv = 1.f / s;
    for (int c = 0; c < C; ++c) probs[r * ldp + c] = expf(zr[c] - mx) * inv;
    if (preds) preds[r] = arg;
}

}  // namespace

// doubles of workspace for M rows
GNX_EXPORT long gnx_masked_ce_workspace(long M) { return 3L * gnx_cdiv(M, 256); }

GNX_EXPORT int gnx_masked_ce_fwd(const float* logits, long ld, const long long* labels, long M, int C,
                                 int label_base, float accum_iters, float* loss, long long* stats, long long* preds,
                                 double* workspace, hipStream_t stream) {
    if (!logits || !labels || !loss || !stats || !workspace || M <= 0 || C <= 0 || ld < C)
        return GNX_ERR_BAD_ARG;
    const int nblk = gnx_cdiv(M, 256);
    masked_ce_fwd_kernel<<<nblk, 256, 0, stream>>>(logits, ld, labels, M, C, label_base, workspace, preds);
    masked_ce_finalize_kernel<<<1, 64, 0, stream>>>(workspace, nblk, accum_iters, loss, stats);
    return gnx_launch_status();
}

GNX_EXPORT int gnx_masked_ce_bwd(const float* logits, long ld, const long long* labels, long M, int C,
                                 int label_base, const long long* stats, const float* dloss, float accum_iters, float* dlogits,
                                 long lddz, hipStream_t stream) {
    if (!logits || !labels || !stats || !dlogits || M <= 0 || C <= 0 || ld < C || lddz < C)
        return GNX_ERR_BAD_ARG;
    masked_ce_bwd_kernel<<<gnx_cdiv(M, 256), 256, 0, stream>>>(logits, ld, labels, M, C, label_base, stats, dloss,
                                                                 accum_iters, dlogits, lddz);
    return gnx_launch_status();
}

// The loops' per-batch bookkeeping (training.py:73-75, :176-178: running_loss += loss.item() * batch_size, running_corrects +=
// correct, ...) as ONE launch on device-resident sums: acc[0] += (double)*loss * weight, acc[1] += *correct,
// acc[2] += counted ? *counted : counted_const.  (Five torch elementwise launches before: ~30 us of host time per batch in loops
// whose whole step is 250-350 us.)  Sums of int64 counts are exact in double up to 2^53.
namespace {
__global__ void meter_add_kernel(double* __restrict__ acc, const float* __restrict__ loss, double weight,
                                 const long long* __restrict__ correct, const long long* __restrict__ counted,
                                 double counted_const) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    acc[0] += (double)*loss * weight;
    acc[1] += (double)*correct;
    acc[2] += counted ? (double)*counted : counted_const;
}
}  // namespace
GNX_EXPORT int gnx_meter_add(double* acc, const float* loss, double weight, const long long* correct, const long long* counted,
                             double counted_const, hipStream_t stream) {
    if (!acc || !loss || !correct) return GNX_ERR_BAD_ARG;
    meter_add_kernel<<<1, 64, 0, stream>>>(acc, loss, weight, correct, counted, counted_const);
    return gnx_launch_status();
}

// Row softmax + argmax of channels-last logits (all_fgd_predictions, gridnext/utils.py:36-47)
GNX_EXPORT int gnx_softmax_rows(const float* logits, long ld, long M, int C, float* probs, long ldp, long long* preds,
                                hipStream_t stream) {
    if (!logits || !probs || M <= 0 || C <= 0 || ld < C || ldp < C) return GNX_ERR_BAD_ARG;
    softmax_rows_kernel<<<gnx_cdiv(M, 256), 256, 0, stream>>>(logits, ld, M, C, probs, ldp, preds);
    return gnx_launch_status();
}
