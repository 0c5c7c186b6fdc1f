// Sliding-window inference and label metrics for gfx950.
// Reference call sites: utils/inference_runtime.py:4-19 (monai sliding_window_inference, constant blending), utils/inference_brats.py:209-217
// (argmax + Dice), utils/metric/metrics.py:44-91 (metrics_tensor), utils/metric/metrics_brats.py:22-35 (Dice / cal_dice).
// All kernels are HBM-bound streaming passes: float4 rows where the window geometry allows, one pass over the volume each.
#include "vx_common.h"

struct VxWin {
    int C, D, H, W;        // volume (one batch item): (C, D, H, W)
    int rd, rh, rw;        // window extent
    int z0, y0, x0;        // window origin inside the volume
};

// win[c, z, y, x] = vol[c, z0+z, y0+y, x0+x]
__global__ void __launch_bounds__(256) vx_sw_extract_k(const float* __restrict__ vol, float* __restrict__ win, VxWin P) {
    const long n = (long)P.C * P.rd * P.rh * P.rw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % P.rw);
        long t = i / P.rw;
        const int y = (int)(t % P.rh); t /= P.rh;
        const int z = (int)(t % P.rd);
        const int c = (int)(t / P.rd);
        win[i] = vol[(((long)c * P.D + P.z0 + z) * P.H + P.y0 + y) * P.W + P.x0 + x];
    }
}

// acc[c, z0+z, y0+y, x0+x] += weight * win[c, z, y, x]   (windows are added one launch after the other: the fp32 summation order is the
// window order, as in the reference loop)
__global__ void __launch_bounds__(256) vx_sw_accumulate_k(const float* __restrict__ win, float* __restrict__ acc, VxWin P, float weight) {
    const long n = (long)P.C * P.rd * P.rh * P.rw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % P.rw);
        long t = i / P.rw;
        const int y = (int)(t % P.rh); t /= P.rh;
        const int z = (int)(t % P.rd);
        const int c = (int)(t / P.rd);
        float* __restrict__ a = acc + (((long)c * P.D + P.z0 + z) * P.H + P.y0 + y) * P.W + P.x0 + x;
        *a += weight * win[i];
    }
}

// out[c, v] = acc[c, v] / (cz[z] * cy[y] * cx[x])  (may alias acc);  labels[v] = first arg-max over c of the normalised value
__global__ void __launch_bounds__(256) vx_sw_finalize_k(const float* acc, float* out, unsigned char* __restrict__ labels,
                                                        const float* __restrict__ cz, const float* __restrict__ cy, const float* __restrict__ cx,
                                                        int C, int D, int H, int W) {
    const long V = (long)D * H * W;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
        const int x = (int)(v % W);
        const int y = (int)((v / W) % H);
        const int z = (int)(v / ((long)W * H));
        const float cnt = (cz[z] * cy[y]) * cx[x];
        float best = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float q = acc[(long)c * V + v] / cnt;
            if (out) out[(long)c * V + v] = q;
            if (q > best || (q != q && !(best != best))) { best = q; arg = c; }    // NaN ranks highest, first occurrence wins (aten)
        }
        if (labels) labels[v] = (unsigned char)arg;
    }
}

// labels[b, v] = argmax_c logits[b, c, v]  (uint8; first maximum)
__global__ void __launch_bounds__(256) vx_argmax_k(const float* __restrict__ logits, unsigned char* __restrict__ labels, int C, long V) {
    const long b = blockIdx.y;
    const float* __restrict__ p = logits + b * C * V;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
        float best = p[v];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float q = p[(long)c * V + v];
            if (q > best || (q != q && !(best != best))) { best = q; arg = c; }
        }
        labels[b * V + v] = (unsigned char)arg;
    }
}

// conf[b, g, p] = #voxels of sample b with ground truth g and prediction p (classes >= NC are clamped to NC-1); LDS histogram per block
template <typename TP, typename TG>
__global__ void __launch_bounds__(256) vx_confusion_k(const TP* __restrict__ pred, const TG* __restrict__ gt, unsigned long long* __restrict__ conf, long V, int NC) {
    __shared__ unsigned int h[64];
    const int b = blockIdx.y;
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const TP* __restrict__ p = pred + (long)b * V;
    const TG* __restrict__ g = gt + (long)b * V;
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
        int pi = (int)p[v], gi = (int)g[v];
        pi = pi < 0 ? 0 : (pi >= NC ? NC - 1 : pi);
        gi = gi < 0 ? 0 : (gi >= NC ? NC - 1 : gi);
        atomicAdd(&h[gi * NC + pi], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NC * NC && h[threadIdx.x]) atomicAdd(conf + (long)b * NC * NC + threadIdx.x, (unsigned long long)h[threadIdx.x]);
}

static int vx_win_check(const VxWin& P, const char* who) {
    if (P.C <= 0 || P.D <= 0 || P.H <= 0 || P.W <= 0 || P.rd <= 0 || P.rh <= 0 || P.rw <= 0) VX_FAIL(-1, "%s: non-positive extent", who);
    if (P.z0 < 0 || P.y0 < 0 || P.x0 < 0 || P.z0 + P.rd > P.D || P.y0 + P.rh > P.H || P.x0 + P.rw > P.W)
        VX_FAIL(-1, "%s: window (%d,%d,%d)+(%d,%d,%d) leaves the volume (%d,%d,%d)", who, P.z0, P.y0, P.x0, P.rd, P.rh, P.rw, P.D, P.H, P.W);
    return 0;
}
static inline int vx_grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 65536 ? 65536 : (g < 1 ? 1 : g)); }

extern "C" int vx_sw_extract(const float* vol, float* win, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0, void* stream) {
    VxWin P{C, D, H, W, rd, rh, rw, z0, y0, x0};
    if (int e = vx_win_check(P, "vx_sw_extract")) return e;
    VX_REQUIRE(vol && win, "vx_sw_extract: null pointer");
    vx_sw_extract_k<<<dim3(vx_grid_for((long)C * rd * rh * rw)), dim3(256), 0, (hipStream_t)stream>>>(vol, win, P);
    VX_LAUNCH_CHECK("vx_sw_extract");
    return 0;
}

extern "C" int vx_sw_accumulate(const float* win, float* acc, int C, int D, int H, int W, int rd, int rh, int rw, int z0, int y0, int x0, float weight, void* stream) {
    VxWin P{C, D, H, W, rd, rh, rw, z0, y0, x0};
    if (int e = vx_win_check(P, "vx_sw_accumulate")) return e;
    VX_REQUIRE(win && acc, "vx_sw_accumulate: null pointer");
    vx_sw_accumulate_k<<<dim3(vx_grid_for((long)C * rd * rh * rw)), dim3(256), 0, (hipStream_t)stream>>>(win, acc, P, weight);
    VX_LAUNCH_CHECK("vx_sw_accumulate");
    return 0;
}

extern "C" int vx_sw_finalize(const float* acc, float* out, unsigned char* labels, const float* cz, const float* cy, const float* cx,
                              int C, int D, int H, int W, void* stream) {
    VX_REQUIRE(acc && cz && cy && cx && (out || labels), "vx_sw_finalize: null pointer");
    VX_REQUIRE(C > 0 && C <= 256 && D > 0 && H > 0 && W > 0, "vx_sw_finalize: bad shape");
    vx_sw_finalize_k<<<dim3(vx_grid_for((long)D * H * W)), dim3(256), 0, (hipStream_t)stream>>>(acc, out, labels, cz, cy, cx, C, D, H, W);
    VX_LAUNCH_CHECK("vx_sw_finalize");
    return 0;
}

extern "C" int vx_argmax_channels(const float* logits, unsigned char* labels, int B, int C, long V, void* stream) {
    VX_REQUIRE(logits && labels && B > 0 && C > 0 && C <= 256 && V > 0, "vx_argmax_channels: bad arguments");
    vx_argmax_k<<<dim3(vx_grid_for(V), B), dim3(256), 0, (hipStream_t)stream>>>(logits, labels, C, V);
    VX_LAUNCH_CHECK("vx_argmax_channels");
    return 0;
}

// pred_bytes / gt_bytes: element size of the label tensors (1 = uint8, 4 = int32, 8 = int64)
extern "C" int vx_confusion(const void* pred, int pred_bytes, const void* gt, int gt_bytes, unsigned long long* conf, int B, long V, int NC, void* stream) {
    VX_REQUIRE(pred && gt && conf && B > 0 && V > 0, "vx_confusion: bad arguments");
    VX_REQUIRE(NC >= 2 && NC <= 8, "vx_confusion: 2..8 classes supported, got %d", NC);
    int g = vx_grid_for(V);
    if (g > 1024) g = 1024;
    const dim3 grid(g, B), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define VX_CONF(TP, TG) vx_confusion_k<TP, TG><<<grid, blk, 0, s>>>((const TP*)pred, (const TG*)gt, conf, V, NC)
    if (pred_bytes == 1 && gt_bytes == 1) VX_CONF(unsigned char, unsigned char);
    else if (pred_bytes == 1 && gt_bytes == 8) VX_CONF(unsigned char, long long);
    else if (pred_bytes == 1 && gt_bytes == 4) VX_CONF(unsigned char, int);
    else if (pred_bytes == 8 && gt_bytes == 8) VX_CONF(long long, long long);
    else if (pred_bytes == 8 && gt_bytes == 1) VX_CONF(long long, unsigned char);
    else if (pred_bytes == 4 && gt_bytes == 4) VX_CONF(int, int);
    else VX_FAIL(-3, "vx_confusion: unsupported label widths %d / %d bytes", pred_bytes, gt_bytes);
#undef VX_CONF
    VX_LAUNCH_CHECK("vx_confusion");
    return 0;
}
