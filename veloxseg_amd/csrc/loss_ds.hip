// Deep-supervision loss with the trilinear up-sampling of the low-resolution heads FUSED IN (gfx950, fp32 data, fp64 accumulators).
// Reference: model/VeloxSeg.py:177-184,202 (scale_prediction: F.interpolate(mode="trilinear", align_corners=True) of every head to the input size)
// followed by utils/loss.py:30-48 (CE + Dice per head).  The three up-sampled (B, ncls, S^3) tensors -- 35 % of the compulsory activation traffic of a
// training step -- are never materialised:
//   forward  : vx_seg_loss_ds_fwd   one sweep over the full-resolution voxels; head 0 is read, heads 1.. are interpolated on the fly from their
//                                   (B, C, d, h, w) grids (same arithmetic as vx_upsample_trilinear_fwd, so the values are identical), labels read once
//   backward : vx_seg_loss_ds_bwd   d(logit) of head 0 is stored; for heads 1.. the gradient at the up-sampled logits is formed in registers and pushed
//                                   through the ADJOINT of the interpolation inside the kernel: along W in LDS per row (owner lanes sum the fine voxels
//                                   of their coarse column), along H into per-wave LDS accumulators of the block's Z slice; a small second kernel
//                                   (vx_seg_loss_ds_adj_z) finishes the adjoint along D for all heads.  2 launches instead of 10.
#include "vx_common.h"
#include <stdlib.h>
#include "../../include/veloxseg_hip.h"

__device__ __forceinline__ int vx_lab(const void* lab, int kind, long i) {
    if (kind == 0) return (int)((const long long*)lab)[i];
    if (kind == 1) return ((const int*)lab)[i];
    return (int)((const unsigned char*)lab)[i];
}
__device__ __forceinline__ void vx_lab4(const void* lab, int kind, long i4, int (&y)[4]) {      // labels of 4 consecutive voxels, i4 % 4 == 0
    if (kind == 0) {
        const longlong2 a = ((const longlong2*)lab)[i4 >> 1], c2 = ((const longlong2*)lab)[(i4 >> 1) + 1];
        y[0] = (int)a.x; y[1] = (int)a.y; y[2] = (int)c2.x; y[3] = (int)c2.y;
    } else if (kind == 1) {
        const int4 a = ((const int4*)lab)[i4 >> 2];
        y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
    } else {
        const uchar4 a = ((const uchar4*)lab)[i4 >> 2];
        y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
    }
}
// source index pair + weight of output index j (align_corners=True): identical to vx_up_coord of loss.hip
__device__ __forceinline__ void vx_ds_coord(int j, int nin, int nout, int& i0, int& i1, float& lam) {
    if (nout == nin) { i0 = j; i1 = j; lam = 0.0f; return; }
    const float ratio = nout > 1 ? (float)(nin - 1) / (float)(nout - 1) : 0.0f;
    const float s = ratio * (float)j;
    i0 = (int)s;
    lam = s - (float)i0;
    i1 = i0 + (i0 < nin - 1 ? 1 : 0);
}

// the same with the ratio (nin - 1) / (nout - 1) given (an IEEE fp32 quotient on the host equals the device's)
__device__ __forceinline__ void vx_ds_coord_r(int j, int nin, float ratio, int& i0, int& i1, float& lam) {
    const float s = ratio * (float)j;
    i0 = (int)s;
    lam = s - (float)i0;
    i1 = i0 + (i0 < nin - 1 ? 1 : 0);
}

struct VxDs {
    const float* l0;            // head 0, full resolution (B, C, D, H, W)
    const float* low[3];        // heads 1..nh-1 on their own grids
    int ld[3][3];               // (d, h, w) of those grids
    float* dl0;                 // backward: gradient of head 0
    float* t2[3];               // backward: (B, C, D, h, w) partially reduced gradients of heads 1..
    int nh, B, D, H, W;
    int stage;                  // 1: the Z-interpolated slices of a sample's grids are staged in LDS (they fit); 0: 8 taps gathered from global memory
    int nsplit;                 // forward: blocks per Z slice
    int dbg;                    // timing experiments (VX_DS_DBG): bit 0 no final atomics, bit 1 no low-resolution heads
    float rh[3], rw[3];         // column-owner kernels: (h - 1) / (H - 1), (w - 1) / (W - 1) of heads 1.. (the quotient vx_ds_coord forms, computed once on the host)
};

// logits of head hh (>= 1) at the 4 voxels (Z, Y, X0..X0+3): z[c][j]
// `lowb` = the (C, d, h, w) grid of THIS sample (staged in LDS by vx_ds_stage: the 8 taps of a voxel are gathers, and gathers from global memory were
// what the first version of these kernels spent their time on)
template <int C>
__device__ __forceinline__ void vx_ds_interp(const VxDs& P, int hh, const float* __restrict__ lowb, int Z, int Y, int X0, float (&z)[C][4]) {
    const int d = P.ld[hh][0], h = P.ld[hh][1], w = P.ld[hh][2];
    int a0, b0, a1, b1;
    float l0, l1;
    vx_ds_coord(Z, d, P.D, a0, b0, l0);
    vx_ds_coord(Y, h, P.H, a1, b1, l1);
    const float k0 = 1.0f - l0, k1 = 1.0f - l1;
    int a2[4], b2[4];
    float l2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vx_ds_coord(X0 + j, w, P.W, a2[j], b2[j], l2[j]);
    // (reached only without staging -- P.stage == 0, grids too large for LDS -- where `lowb` is the head's tensor in global memory: an explicit address space keeps the
    // 8 taps from becoming FLAT loads)
    typedef const __attribute__((address_space(1))) float* glb_cf;
    glb_cf lowg = (glb_cf)(unsigned long long)lowb;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        glb_cf xb = lowg + (long)c * d * h * w;
        glb_cf r00 = xb + ((long)a0 * h + a1) * w;
        glb_cf r01 = xb + ((long)a0 * h + b1) * w;
        glb_cf r10 = xb + ((long)b0 * h + a1) * w;
        glb_cf r11 = xb + ((long)b0 * h + b1) * w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float k2 = 1.0f - l2[j];
            z[c][j] = k0 * (k1 * (k2 * r00[a2[j]] + l2[j] * r00[b2[j]]) + l1 * (k2 * r01[a2[j]] + l2[j] * r01[b2[j]])) +
                      l0 * (k1 * (k2 * r10[a2[j]] + l2[j] * r10[b2[j]]) + l1 * (k2 * r11[a2[j]] + l2[j] * r11[b2[j]]));
        }
    }
}

// stage the grids of sample b (heads 1..nh-1) into LDS; lows[hh] receives the LDS base of head hh.  Returns the floats used.
template <int C>
__device__ __forceinline__ int vx_ds_stage(const VxDs& P, int b, float* __restrict__ lds, const float* (&lows)[3]) {
    int off = 0;
    for (int hh = 0; hh < 3; ++hh) {
        lows[hh] = lds + off;
        if (hh >= P.nh - 1) continue;
        const int n = C * P.ld[hh][0] * P.ld[hh][1] * P.ld[hh][2];
        const float* __restrict__ src = P.low[hh] + (long)b * n;
        if (!P.stage) { lows[hh] = src; continue; }
        for (int e0 = threadIdx.x; e0 < n; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; v[u] = src[e < n ? e : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * 256; if (e < n) lds[off + e] = v[u]; }
        }
        off += (n + 3) & ~3;
    }
    return off;
}

// A block works on ONE Z slice, so the Z step of the interpolation is the same for all of its voxels: the grids of sample b are interpolated along Z once,
// while they are staged ([c][h][w] per head, C * h * w floats instead of C * d * h * w), and a voxel then gathers 4 taps per head and class instead of 8.
template <int C>
__device__ __forceinline__ void vx_ds_stage_slice(const VxDs& P, int b, int Z, float* __restrict__ lds, const float* (&lows)[3]) {
    int off = 0;
    for (int hh = 0; hh < 3; ++hh) {
        lows[hh] = lds + off;
        if (hh >= P.nh - 1) continue;
        const int d = P.ld[hh][0], hw = P.ld[hh][1] * P.ld[hh][2];
        const int n = (C == 2 ? 1 : C) * hw;
        const float* __restrict__ src = P.low[hh] + (long)b * C * d * hw;
        int a0, b0; float l0;
        vx_ds_coord(Z, d, P.D, a0, b0, l0);
        const float k0 = 1.0f - l0;
        for (int e = threadIdx.x; e < n; e += 256) {
            if constexpr (C == 2) {
                // two classes: soft-max, cross-entropy and the Dice terms only see z1 - z0, and the interpolation is linear -- ONE grid (the difference) is staged
                // and interpolated, the logits of a voxel are (0, difference); the gradients of the two classes are opposite, one of them goes through the adjoint
                lds[off + e] = (k0 * src[((long)d + a0) * hw + e] + l0 * src[((long)d + b0) * hw + e]) - (k0 * src[(long)a0 * hw + e] + l0 * src[(long)b0 * hw + e]);
            } else {
                const int c = e / hw, r = e - c * hw;
                lds[off + e] = k0 * src[((long)c * d + a0) * hw + r] + l0 * src[((long)c * d + b0) * hw + r];
            }
        }
        off += (n + 3) & ~3;
    }
}
// logits of head hh (>= 1) at the 4 voxels (Y, X0..X0+3) of the block's slice: z[c][j]
template <int C>
__device__ __forceinline__ void vx_ds_interp2(const VxDs& P, int hh, const float* __restrict__ sl, int Y, int X0, float (&z)[C][4]) {
    const int h = P.ld[hh][1], w = P.ld[hh][2];
    int a1, b1;
    float l1;
    vx_ds_coord(Y, h, P.H, a1, b1, l1);
    const float k1 = 1.0f - l1;
    int a2[4], b2[4];
    float l2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vx_ds_coord(X0 + j, w, P.W, a2[j], b2[j], l2[j]);
    // `sl` is the staged slice in LDS (this function is only reached with P.stage): say so -- through the generic pointer of `lows[]` (LDS or global, chosen at run
    // time) these 16 gathers per class were FLAT loads, which are slower than ds_read and count on both wait counters
    typedef const __attribute__((address_space(3))) float* lds_cf;
    lds_cf sl3 = (lds_cf)sl;
    if constexpr (C == 2) {          // the staged slice is the difference grid (vx_ds_stage_slice)
        lds_cf r0 = sl3 + a1 * w;
        lds_cf r1 = sl3 + b1 * w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float k2 = 1.0f - l2[j];
            z[0][j] = 0.0f;
            z[1][j] = k1 * (k2 * r0[a2[j]] + l2[j] * r0[b2[j]]) + l1 * (k2 * r1[a2[j]] + l2[j] * r1[b2[j]]);
        }
    } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
        lds_cf r0 = sl3 + (c * h + a1) * w;
        lds_cf r1 = sl3 + (c * h + b1) * w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float k2 = 1.0f - l2[j];
            z[c][j] = k1 * (k2 * r0[a2[j]] + l2[j] * r0[b2[j]]) + l1 * (k2 * r1[a2[j]] + l2[j] * r1[b2[j]]);
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ forward
// acc layout (double) as vx_seg_loss_fwd: head h at h*(1 + B*C*3): [ce_sum, (I, P, T) x (b, c)].  grid (D, B): a block = one Z slice
template <int C>
__global__ void __launch_bounds__(256) vx_seg_loss_ds_fwd_k(VxDs P, const void* __restrict__ lab, int lab_kind, double* __restrict__ acc) {
    constexpr int NS = 1 + 2 * C;
    extern __shared__ __attribute__((aligned(16))) float vx_ds_lds[];
    const int b = blockIdx.y, Z = blockIdx.x / P.nsplit, part = blockIdx.x % P.nsplit;
    const int W4 = P.W >> 2;
    const long V = (long)P.D * P.H * P.W;
    const float* lows[3];
    if (P.stage) vx_ds_stage_slice<C>(P, b, Z, vx_ds_lds, lows);
    else vx_ds_stage<C>(P, b, vx_ds_lds, lows);
    __syncthreads();
    const int nq = P.H * W4, q_lo = (int)((long)nq * part / P.nsplit), q_hi = (int)((long)nq * (part + 1) / P.nsplit);
    float S[4][NS], T[C];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int k = 0; k < NS; ++k) S[h][k] = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) T[c] = 0.0f;
    for (int qs = q_lo + threadIdx.x; qs < q_hi; qs += 256) {
        const int X0 = (qs % W4) * 4, Y = qs / W4;
        const long q = ((long)Z * P.H * P.W >> 2) + qs;          // quad index inside the sample
        int y[4];
        vx_lab4(lab, lab_kind, (long)b * V + 4 * q, y);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) T[c] += (y[j] == c) ? 1.0f : 0.0f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (h < P.nh) {
                float z[C][4];
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float4 t = *reinterpret_cast<const float4*>(P.l0 + ((long)b * C + c) * V + 4 * q);
                        z[c][0] = t.x; z[c][1] = t.y; z[c][2] = t.z; z[c][3] = t.w;
                    }
                } else if (P.stage) vx_ds_interp2<C>(P, h - 1, lows[h - 1], Y, X0, z);
                else vx_ds_interp<C>(P, h - 1, lows[h - 1], Z, Y, X0, z);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mx = z[0][j];
#pragma unroll
                    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c][j]);
                    float e[C], se = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { e[c] = __expf(z[c][j] - mx); se += e[c]; }      // (v_exp_f32 / v_rcp_f32 / v_log_f32: 1 ulp; the library calls were half of the kernel's issue slots)
                    const float inv = __frcp_rn(se);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float pc = e[c] * inv;
                        S[h][1 + C + c] += pc;
                        if (c == y[j]) { S[h][1 + c] += pc; S[h][0] += (mx + __logf(se)) - z[c][j]; }
                    }
                }
            }
        }
    }
    __shared__ float red[4][4 * NS + C];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int k = 0; k < NS; ++k) { const float v = vx_wave_sum(S[h][k]); if (lane == 0) red[wid][h * NS + k] = v; }
#pragma unroll
    for (int c = 0; c < C; ++c) { const float v = vx_wave_sum(T[c]); if (lane == 0) red[wid][4 * NS + c] = v; }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < 4 * NS + C) {
        const double v = (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k];
        if (k < 4 * NS) {
            const int h = k / NS, r = k % NS;
            if (h < P.nh) {
                double* __restrict__ ah = acc + (long)h * (1 + (long)P.B * C * 3);
                if (r == 0) atomicAdd(ah, v);
                else if (r <= C) atomicAdd(ah + 1 + ((long)b * C + (r - 1)) * 3, v);
                else atomicAdd(ah + 1 + ((long)b * C + (r - 1 - C)) * 3 + 1, v);
            }
        } else {
            const int c = k - 4 * NS;
            for (int h = 0; h < P.nh; ++h) atomicAdd(acc + (long)h * (1 + (long)P.B * C * 3) + 1 + ((long)b * C + c) * 3 + 2, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ backward
// block = (b, Z slice); wave = RPW rows of the slice per step (RPW = 64 / (W/4) row segments of W/4 lanes, 4 voxels per lane).
// LDS: gbuf[wave][row][head][c][W] (gradients at the up-sampled logits of one step) and accw[wave][sum_h C*h_h*w_h] (this wave's H/W-reduced sums).
template <int C>
__global__ void __launch_bounds__(256) vx_seg_loss_ds_bwd_k(VxDs P, const void* __restrict__ lab, int lab_kind, const float* __restrict__ coef, int coef_stride,
                                                            const float* __restrict__ gout, int nacc) {
    extern __shared__ __attribute__((aligned(16))) float vx_ds_lds[];
    const int b = blockIdx.y, Z = blockIdx.x / P.nsplit, part = blockIdx.x % P.nsplit;
    const int y_lo = (int)((long)P.H * part / P.nsplit), y_hi = (int)((long)P.H * (part + 1) / P.nsplit);      // this block's rows of the slice
    const int W4 = P.W >> 2, RPW = 64 / W4, nlow = P.nh - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long V = (long)P.D * P.H * P.W;
    // two classes with staged slices: the gradients at the up-sampled logits are opposite (g0 = -g1) -- only class 1 goes through the adjoints, class 0 is its negative
    const int c_lo = (C == 2 && P.stage) ? 1 : 0, CA = C - c_lo;
    float* __restrict__ accw = vx_ds_lds + (long)wave * nacc;                                   // [head][c][y][x]
    float* __restrict__ gbuf = vx_ds_lds + 4L * nacc + (long)wave * (RPW * 3 * C * P.W);        // [row][head][c][X]
    // banded adjoint tables of the W axis: coarse column xl of head hh collects the fine columns lo .. lo + BW - 1 with weights wtab (zero past the band)
    float* __restrict__ wtab = vx_ds_lds + 4L * nacc + 4L * (RPW * 3 * C * P.W);                // [head][xl][BW]
    int bw[3], toff[3], ntab = 0;
    for (int hh = 0; hh < 3; ++hh) {
        const int wl = P.ld[hh][2];
        const float ratio = P.W > 1 ? (float)(wl - 1) / (float)(P.W - 1) : 0.0f;
        bw[hh] = (hh < nlow) ? ((wl == P.W || ratio <= 0.0f) ? P.W : min(P.W, (int)(2.0f / ratio) + 4)) : 0;
        toff[hh] = ntab;
        if (hh < nlow) ntab += wl * bw[hh];
    }
    int* __restrict__ lotab = reinterpret_cast<int*>(wtab + ((ntab + 3) & ~3));                  // [head][xl]: first fine column of the band
    const float* lows[3];
    if (P.stage) vx_ds_stage_slice<C>(P, b, Z, reinterpret_cast<float*>(lotab + 3 * P.W), lows);
    else vx_ds_stage<C>(P, b, reinterpret_cast<float*>(lotab + 3 * P.W), lows);
    for (int hh = 0; hh < nlow; ++hh) {
        const int wl = P.ld[hh][2];
        const float ratio = P.W > 1 ? (float)(wl - 1) / (float)(P.W - 1) : 0.0f;
        for (int e = threadIdx.x; e < wl * bw[hh]; e += 256) {
            const int xl = e / bw[hh], k = e - xl * bw[hh];
            int lo = 0;
            if (wl != P.W && ratio > 0.0f) { lo = (int)ceilf(((float)xl - 1.0f) / ratio) - 1; if (lo < 0) lo = 0; }
            const int X = lo + k;
            float wgt = 0.0f;
            if (X < P.W) {
                int i0, i1; float lam;
                vx_ds_coord(X, wl, P.W, i0, i1, lam);
                wgt = (i0 == xl ? 1.0f - lam : 0.0f) + (i1 == xl ? lam : 0.0f);
            }
            wtab[toff[hh] + e] = wgt;
            if (k == 0) lotab[hh * P.W + xl] = lo;
        }
    }
    for (int k = lane; k < nacc; k += 64) accw[k] = 0.0f;
    __syncthreads();
    const float go = gout ? gout[0] : 1.0f;
    int aoff[3];                                                                                // offsets of the heads inside accw
    { int o = 0; for (int hh = 0; hh < 3; ++hh) { aoff[hh] = o; if (hh < nlow) o += CA * P.ld[hh][1] * P.ld[hh][2]; } }
    const int rsl = lane / W4, X0 = (lane % W4) * 4;                                            // row slot of this lane, first voxel
    for (int Y0 = y_lo + wave * RPW; Y0 < y_hi; Y0 += 4 * RPW) {
        const int Y = Y0 + rsl;
        const bool rowok = rsl < RPW && Y < y_hi;
        if (rowok) {
            int y[4];
            vx_lab4(lab, lab_kind, (long)b * V + ((long)Z * P.H + Y) * P.W + X0, y);
            for (int h = 0; h < P.nh; ++h) {
                float z[C][4];
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float4 t = *reinterpret_cast<const float4*>(P.l0 + ((long)b * C + c) * V + ((long)Z * P.H + Y) * P.W + X0);
                        z[c][0] = t.x; z[c][1] = t.y; z[c][2] = t.z; z[c][3] = t.w;
                    }
                } else if (P.stage) vx_ds_interp2<C>(P, h - 1, lows[h - 1], Y, X0, z);
                else vx_ds_interp<C>(P, h - 1, lows[h - 1], Z, Y, X0, z);
                const float* __restrict__ coef_h = coef + (long)h * coef_stride;
                const float wce = coef_h[0];
                float al[C], be[C];
#pragma unroll
                for (int c = 0; c < C; ++c) { al[c] = coef_h[1 + ((long)b * C + c) * 2]; be[c] = coef_h[2 + ((long)b * C + c) * 2]; }
                float g[C][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mx = z[0][j];
#pragma unroll
                    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c][j]);
                    float se = 0.0f, dot = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { z[c][j] = __expf(z[c][j] - mx); se += z[c][j]; }
                    const float inv = __frcp_rn(se);
#pragma unroll
                    for (int c = 0; c < C; ++c) { z[c][j] *= inv; dot = fmaf(z[c][j], (c == y[j] ? al[c] : 0.0f) + be[c], dot); }
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float gg = (c == y[j] ? al[c] : 0.0f) + be[c];
                        g[c][j] = go * (wce * (z[c][j] - (c == y[j] ? 1.0f : 0.0f)) + z[c][j] * (gg - dot));
                    }
                }
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (h == 0) *reinterpret_cast<float4*>(P.dl0 + ((long)b * C + c) * V + ((long)Z * P.H + Y) * P.W + X0) = make_float4(g[c][0], g[c][1], g[c][2], g[c][3]);
                    else if (c >= c_lo) *reinterpret_cast<float4*>(gbuf + (((long)rsl * 3 + (h - 1)) * CA + (c - c_lo)) * P.W + X0) = make_float4(g[c][0], g[c][1], g[c][2], g[c][3]);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // adjoint along W (owner = (head, c, coarse x)), then along H into this wave's accumulators.  An owner's band is split over SP lanes (64 / owners,
        // a power of two: 2 / 4 / 8 for the 16 / 8 / 4-wide heads of a 2-class 128^3 patch) whose partial sums meet through shuffles: on the owner lanes
        // alone the three heads were ~300 dependent LDS iterations per step on at most half of the lanes.
        // Owners walk the step's rows one after the other, so two rows never add into one accumulator at the same time.
        for (int hh = 0; hh < nlow; ++hh) {
            const int hl = P.ld[hh][1], wl = P.ld[hh][2];
            const int nown = CA * wl;
            int sp = 1;
            while (sp < 64 && nown * (sp * 2) <= 64) sp *= 2;
            const int part = lane & (sp - 1);
            for (int o0 = 0; o0 < nown; o0 += 64 / sp) {
                const int o = o0 + lane / sp;
                const bool own = o < nown;
                const int oc = own ? o : 0;
                const int c = oc / wl, xl = oc - c * wl;
                const int lo = lotab[hh * P.W + xl], nb = bw[hh];
                const float* __restrict__ wt = wtab + toff[hh] + xl * nb;
                for (int r = 0; r < RPW; ++r) {
                    const int Yr = Y0 + r;
                    if (Yr >= y_hi) break;
                    const float* __restrict__ gr = gbuf + (((long)r * 3 + hh) * CA + c) * P.W;
                    float s = 0.0f;
#pragma unroll 4
                    for (int k = part; k < nb; k += sp) s = fmaf(wt[k], gr[min(lo + k, P.W - 1)], s);      // independent LDS reads: the band's weights are zero past its end
                    for (int m = sp >> 1; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
                    if (own && part == 0) {
                        int a1, b1; float l1;
                        vx_ds_coord(Yr, hl, P.H, a1, b1, l1);
                        float* __restrict__ dst = accw + aoff[hh] + (long)c * hl * wl + xl;
                        dst[(long)a1 * wl] += (1.0f - l1) * s;
                        if (b1 != a1) dst[(long)b1 * wl] += l1 * s;
                        else dst[(long)a1 * wl] += l1 * s;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // sum the 4 waves' accumulators and store this Z slice of the (B, C, D, h, w) partial gradients
    for (int hh = 0; hh < nlow; ++hh) {
        const int hl = P.ld[hh][1], wl = P.ld[hh][2], n = CA * hl * wl;
        for (int e = threadIdx.x; e < n; e += 256) {
            const int c = e / (hl * wl), r = e - c * hl * wl;
            const float s = (vx_ds_lds[aoff[hh] + e] + vx_ds_lds[nacc + aoff[hh] + e]) + (vx_ds_lds[2 * nacc + aoff[hh] + e] + vx_ds_lds[3 * nacc + aoff[hh] + e]);
            P.t2[hh][((((long)b * C + c + c_lo) * P.D + Z) * P.nsplit + part) * hl * wl + r] = s;
            if (c_lo) P.t2[hh][((((long)b * C) * P.D + Z) * P.nsplit + part) * hl * wl + r] = -s;
        }
    }
}

// adjoint along D for all heads: dlow_h[b, c, z, y, x] = sum_Z A[Z][z] t2_h[b, c, Z, y, x].  A block = 32 consecutive output elements x 8 parts of their Z band
// (one thread per element walked its whole band -- up to 2 * D / d + 2 planes -- alone: 23 us of dependent loads for 1.4 MB; the parts meet in LDS)
struct VxDsZ { const float* t2[3]; float* out[3]; int ld[3][3]; int nlow, BC, D, nsplit; long n[3]; };
__global__ void __launch_bounds__(256) vx_seg_loss_ds_adj_z_k(VxDsZ P) {
    __shared__ float red[8][32];
    const int l32 = threadIdx.x & 31, zp = threadIdx.x >> 5;
    long e = (long)blockIdx.x * 32 + l32;
    int hh = 0;
    while (hh < P.nlow && e >= P.n[hh]) { e -= P.n[hh]; ++hh; }
    const bool live = hh < P.nlow;
    float s = 0.0f;
    if (live) {
        const int d = P.ld[hh][0], hw = P.ld[hh][1] * P.ld[hh][2];
        const int r = (int)(e % hw), z = (int)((e / hw) % d);
        const long bc = e / ((long)hw * d);
        const float ratio = P.D > 1 ? (float)(d - 1) / (float)(P.D - 1) : 0.0f;
        int lo = 0, hi = P.D - 1;
        if (d != P.D && ratio > 0.0f) {
            lo = (int)ceilf(((float)z - 1.0f) / ratio) - 1; if (lo < 0) lo = 0;
            hi = (int)floorf(((float)z + 1.0f) / ratio) + 1; if (hi > P.D - 1) hi = P.D - 1;
        }
        const float* __restrict__ src = P.t2[hh] + bc * P.D * P.nsplit * hw + r;
        for (int Z = lo + zp; Z <= hi; Z += 8) {
            int i0, i1; float lam;
            vx_ds_coord(Z, d, P.D, i0, i1, lam);
            const float wgt = (i0 == z ? 1.0f - lam : 0.0f) + (i1 == z ? lam : 0.0f);
            float t = 0.0f;
            for (int q = 0; q < P.nsplit; ++q) t += src[((long)Z * P.nsplit + q) * hw];          // the slice's row parts (one block each)
            s = fmaf(wgt, t, s);
        }
    }
    red[zp][l32] = s;
    __syncthreads();
    if (zp == 0 && live) P.out[hh][e] = ((red[0][l32] + red[1][l32]) + (red[2][l32] + red[3][l32])) + ((red[4][l32] + red[5][l32]) + (red[6][l32] + red[7][l32]));
}

// ------------------------------------------------------------------------------------------------------------------------------ column-owner kernels
// The same two sweeps with a different thread -> voxel map (round 5).  A thread owns ONE quad column X0 of the block's Z slice and walks consecutive rows
// (256 / (W/4) row groups per block), so everything that depends on X only is computed ONCE per thread instead of once per quad:
//   * per low-resolution head the NK (3 or 4) coarse columns i0 .. i0 + NK - 1 that the quad's four voxels interpolate from, and a dense 4 x NK weight matrix
//     cw[j][k] (two non-zeros per row): a voxel's logit is  sum_k cw[j][k] * (k1 * row_a[i0 + k] + l1 * row_b[i0 + k])  -- 2 NK LDS reads per head and grid
//     instead of 16 gathers with their index arithmetic (the old kernels spent most of their issue slots on vx_ds_coord and address math);
//   * backward: the ADJOINT of that interpolation is the transposed matrix: the quad's gradient collapses to NK column sums (registers), which are accumulated over
//     the thread's consecutive rows in two register rows (coarse rows a1 and a1 + 1) and flushed to the wave's LDS accumulators with ds_add_f32 only when a1 advances
//     (every 1 / scale rows): ~40 LDS atomics per thread and slice replace the gradient staging buffer, the band tables and their ~300 dependent LDS round trips per step.
// Everything else (soft-max arithmetic, accumulator layout, the D adjoint behind it) is unchanged; vx_seg_loss_ds_set_columns(0) / VELOXSEG_DS_COLUMNS=0 selects the
// row-sweep kernels above.
template <int NK>
__device__ __forceinline__ void vx_ds_col_setup(int X0, int wl, float ratio, int& i0, float (&cw)[4][NK]) {
    int a[4], b[4];
    float l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vx_ds_coord_r(X0 + j, wl, ratio, a[j], b[j], l[j]);
    i0 = a[0];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < NK; ++k) cw[j][k] = ((a[j] - i0) == k ? 1.0f - l[j] : 0.0f) + ((b[j] - i0) == k ? l[j] : 0.0f);
}

// the Z-interpolated slices as vx_ds_stage_slice, each followed by 4 zeroed floats (a thread reads NK columns from i0 on: past the end of a row the weights are zero,
// the values must be finite)
template <int C>
__device__ __forceinline__ void vx_ds_stage_slice_pad(const VxDs& P, int b, int Z, float* __restrict__ lds, int (&loff)[3]) {
    int off = 0;
    for (int hh = 0; hh < 3; ++hh) {
        loff[hh] = off;
        if (hh >= P.nh - 1) continue;
        const int d = P.ld[hh][0], hw = P.ld[hh][1] * P.ld[hh][2];
        const int n = (C == 2 ? 1 : C) * hw;
        const float* __restrict__ src = P.low[hh] + (long)b * C * d * hw;
        int a0, b0; float l0;
        vx_ds_coord(Z, d, P.D, a0, b0, l0);
        const float k0 = 1.0f - l0;
        for (int e = threadIdx.x; e < n + 4; e += 256) {
            float v = 0.0f;
            if (e < n) {
                if constexpr (C == 2) v = (k0 * src[((long)d + a0) * hw + e] + l0 * src[((long)d + b0) * hw + e]) - (k0 * src[(long)a0 * hw + e] + l0 * src[(long)b0 * hw + e]);
                else { const int c = e / hw, r = e - c * hw; v = k0 * src[((long)c * d + a0) * hw + r] + l0 * src[((long)c * d + b0) * hw + r]; }
            }
            lds[off + e] = v;
        }
        off += (n + 4 + 3) & ~3;
    }
}
static size_t vx_ds_slice_pad_floats(const VxDs& P, int C) {
    size_t n = 0;
    for (int hh = 0; hh < P.nh - 1; ++hh) n += ((size_t)(C == 2 ? 1 : C) * P.ld[hh][1] * P.ld[hh][2] + 4 + 3) & ~(size_t)3;
    return n;
}

// logits of head hh at the thread's quad in row Y: z[c][j]; (a1, l1) of the row are returned for the adjoint
template <int C, int NK>
__device__ __forceinline__ void vx_ds_interp_col(const VxDs& P, int hh, const float* __restrict__ sl, int Y, int i0, const float (&cw)[4][NK], float (&z)[C][4], int& a1, float& l1) {
    const int h = P.ld[hh][1], w = P.ld[hh][2];
    int b1;
    vx_ds_coord_r(Y, h, P.rh[hh], a1, b1, l1);
    const float k1 = 1.0f - l1;
    typedef const __attribute__((address_space(3))) float* lds_cf;
    lds_cf s3 = (lds_cf)sl;
    constexpr int NG = C == 2 ? 1 : C;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        lds_cf r0 = s3 + (g * h + a1) * w + i0;
        lds_cf r1 = s3 + (g * h + b1) * w + i0;
        float v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) v[k] = k1 * r0[k] + l1 * r1[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < NK; ++k) t = fmaf(cw[j][k], v[k], t);
            if constexpr (C == 2) { z[0][j] = 0.0f; z[1][j] = t; }
            else z[g][j] = t;
        }
    }
}

// TL = element type of the full-resolution head l0 and (backward) of its gradient dl0: float, or vx_bf16 in the bf16 storage mode (P.l0 / P.dl0 are then bf16 arrays)
template <int C, int NK, typename TL = float>
__global__ void __launch_bounds__(256) vx_seg_loss_ds_fwd_col_k(VxDs P, const void* __restrict__ lab, int lab_kind, double* __restrict__ acc) {
    constexpr int NS = 1 + 2 * C;
    const TL* __restrict__ l0p = reinterpret_cast<const TL*>(P.l0);
    extern __shared__ __attribute__((aligned(16))) float vx_ds_lds[];
    const int b = blockIdx.y, Z = blockIdx.x / P.nsplit, part = blockIdx.x % P.nsplit;
    const int W4 = P.W >> 2, NRG = 256 / W4;
    const long V = (long)P.D * P.H * P.W;
    int loff[3];
    vx_ds_stage_slice_pad<C>(P, b, Z, vx_ds_lds, loff);
    const int y_lo = (int)((long)P.H * part / P.nsplit), y_hi = (int)((long)P.H * (part + 1) / P.nsplit);
    const int RG = (y_hi - y_lo + NRG - 1) / NRG;
    const int X0 = ((int)threadIdx.x % W4) * 4, yg = (int)threadIdx.x / W4;
    const int y0 = y_lo + yg * RG, y1 = min(y_hi, y0 + RG);
    int i0[3];
    float cw[3][4][NK];
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) {
        if (hh < P.nh - 1) vx_ds_col_setup<NK>(X0, P.ld[hh][2], P.rw[hh], i0[hh], cw[hh]);
        else { i0[hh] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k) cw[hh][j][k] = 0.0f; }
    }
    __syncthreads();
    float S[4][NS], T[C], nrows = 0.0f;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int k = 0; k < NS; ++k) S[h][k] = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) T[c] = 0.0f;
    // the row's global operands one row ahead of the arithmetic
    // the rows' global operands TWO rows ahead of the arithmetic (two register buffers, used alternately: a wave keeps two rows of loads in flight)
    int ynA[4] = {0, 0, 0, 0}, ynB[4] = {0, 0, 0, 0};
    float4 lnA[C], lnB[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { lnA[c] = make_float4(0.f, 0.f, 0.f, 0.f); lnB[c] = lnA[c]; }
    auto fetch = [&](int Y, int (&yn)[4], float4 (&ln)[C]) {
        const long o = ((long)Z * P.H + Y) * P.W + X0;
        vx_lab4(lab, lab_kind, (long)b * V + o, yn);
#pragma unroll
        for (int c = 0; c < C; ++c) ln[c] = vx_ld4(l0p, ((long)b * C + c) * V + o);
    };
    if (y0 < y1) fetch(y0, ynA, lnA);
    if (y0 + 1 < y1) fetch(y0 + 1, ynB, lnB);
    auto row = [&](int Y, int (&yn)[4], float4 (&ln)[C]) {
        int y[4] = {yn[0], yn[1], yn[2], yn[3]};
        float z0[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) { z0[c][0] = ln[c].x; z0[c][1] = ln[c].y; z0[c][2] = ln[c].z; z0[c][3] = ln[c].w; }
        if (Y + 2 < y1) fetch(Y + 2, yn, ln);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) T[c] += (y[j] == c) ? 1.0f : 0.0f;
        float m0[4], m1[4];          // (two classes) the label as two 0 / 1 factors, shared by the four heads
#pragma unroll
        for (int j = 0; j < 4; ++j) { m0[j] = y[j] == 0 ? 1.0f : 0.0f; m1[j] = y[j] == 1 ? 1.0f : 0.0f; }
        nrows += 1.0f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (h < P.nh && !((P.dbg & 2) && h > 0)) {
                float z[C][4];
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[c][j] = z0[c][j];
                } else {
                    int a1; float l1;
                    vx_ds_interp_col<C, NK>(P, h - 1, vx_ds_lds + loff[h - 1], Y, i0[h - 1], cw[h - 1], z, a1, l1);
                }
                if constexpr (C == 2) {
                    // two classes: soft-max = logistic of d = z1 - z0.  One exponential and one reciprocal per voxel, and ONE logarithm per quad: the cross-entropy
                    // terms log(1 + e_j) of the four voxels are the logarithm of the product (1 + e in (1, 2]: the product stays below 16).  Only the class-1 sums are
                    // accumulated here -- S[h][1 + C + 1] = sum p1, S[h][1 + 1] = sum_{y = 1} p1, S[h][1 + 0] = sum_{y = 0} p1 -- the class-0 ones follow from
                    // p0 = 1 - p1 after the loop (sum p0 = voxels - sum p1, sum_{y = 0} p0 = T0 - sum_{y = 0} p1)
                    float prod = 1.0f, extra = 0.0f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dd = z[1][j] - z[0][j];
                        const float e = __expf(-fabsf(dd)), inv = __builtin_amdgcn_rcpf(1.0f + e);
                        const float p1 = dd >= 0.0f ? inv : e * inv;
                        S[h][1 + C + 1] += p1;
                        S[h][1 + 1] = fmaf(p1, m1[j], S[h][1 + 1]);
                        S[h][1 + 0] = fmaf(p1, m0[j], S[h][1 + 0]);
                        prod *= fmaf(e, m0[j] + m1[j], 1.0f);
                        extra += fmaxf(dd * (m0[j] - m1[j]), 0.0f);          // lse - z_y = log(1 + e) + max(0, -(d * (2y - 1)))
                    }
                    S[h][0] += __logf(prod) + extra;
                } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mx = z[0][j];
#pragma unroll
                    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c][j]);
                    float e[C], se = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { e[c] = __expf(z[c][j] - mx); se += e[c]; }
                    const float inv = __builtin_amdgcn_rcpf(se);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float pc = e[c] * inv;
                        S[h][1 + C + c] += pc;
                        if (c == y[j]) { S[h][1 + c] += pc; S[h][0] += (mx + __logf(se)) - z[c][j]; }
                    }
                }
                }
            }
        }
    };
    for (int Y = y0; Y < y1; Y += 2) {
        row(Y, ynA, lnA);
        if (Y + 1 < y1) row(Y + 1, ynB, lnB);
    }
    if constexpr (C == 2) {          // class-0 sums from the class-1 ones (see the loop)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            S[h][1 + C + 0] = 4.0f * nrows - S[h][1 + C + 1];
            S[h][1 + 0] = T[0] - S[h][1 + 0];
        }
    }
    __shared__ float red[4][4 * NS + C];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int k = 0; k < NS; ++k) { const float v = vx_wave_sum(S[h][k]); if (lane == 0) red[wid][h * NS + k] = v; }
#pragma unroll
    for (int c = 0; c < C; ++c) { const float v = vx_wave_sum(T[c]); if (lane == 0) red[wid][4 * NS + c] = v; }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < 4 * NS + C && !(P.dbg & 1)) {
        const double v = (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k];
        if (k < 4 * NS) {
            const int h = k / NS, r = k % NS;
            if (h < P.nh) {
                double* __restrict__ ah = acc + (long)h * (1 + (long)P.B * C * 3);
                if (r == 0) atomicAdd(ah, v);
                else if (r <= C) atomicAdd(ah + 1 + ((long)b * C + (r - 1)) * 3, v);
                else atomicAdd(ah + 1 + ((long)b * C + (r - 1 - C)) * 3 + 1, v);
            }
        } else {
            const int c = k - 4 * NS;
            for (int h = 0; h < P.nh; ++h) atomicAdd(acc + (long)h * (1 + (long)P.B * C * 3) + 1 + ((long)b * C + c) * 3 + 2, v);
        }
    }
}

// backward, column owners.  LDS: accw[wave][nacc] (layout [head][ca][y][x], as the row-sweep kernel) | padded slices
template <int C, int NK, typename TL = float>
__global__ void __launch_bounds__(256) vx_seg_loss_ds_bwd_col_k(VxDs P, const void* __restrict__ lab, int lab_kind, const float* __restrict__ coef, int coef_stride,
                                                                const float* __restrict__ gout, int nacc) {
    extern __shared__ __attribute__((aligned(16))) float vx_ds_lds[];
    const TL* __restrict__ l0p = reinterpret_cast<const TL*>(P.l0);
    TL* __restrict__ dl0p = reinterpret_cast<TL*>(P.dl0);
    const int b = blockIdx.y, Z = blockIdx.x / P.nsplit, part = blockIdx.x % P.nsplit;
    const int W4 = P.W >> 2, NRG = 256 / W4, nlow = P.nh - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long V = (long)P.D * P.H * P.W;
    constexpr int c_lo = C == 2 ? 1 : 0, CA = C - c_lo;
    float* __restrict__ accw = vx_ds_lds + (long)wave * nacc;
    float* __restrict__ slices = vx_ds_lds + 4L * nacc;
    int loff[3];
    vx_ds_stage_slice_pad<C>(P, b, Z, slices, loff);
    for (int k = lane; k < nacc; k += 64) accw[k] = 0.0f;
    const int y_lo = (int)((long)P.H * part / P.nsplit), y_hi = (int)((long)P.H * (part + 1) / P.nsplit);
    const int RG = (y_hi - y_lo + NRG - 1) / NRG;
    const int X0 = ((int)threadIdx.x % W4) * 4, yg = (int)threadIdx.x / W4;
    const int y0 = y_lo + yg * RG, y1 = min(y_hi, y0 + RG);
    int i0[3], aoff[3];
    float cw[3][4][NK];
    { int o = 0; for (int hh = 0; hh < 3; ++hh) { aoff[hh] = o; if (hh < nlow) o += CA * P.ld[hh][1] * P.ld[hh][2]; } }
#pragma unroll
    for (int hh = 0; hh < 3; ++hh) {
        if (hh < nlow) vx_ds_col_setup<NK>(X0, P.ld[hh][2], P.rw[hh], i0[hh], cw[hh]);
        else { i0[hh] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < NK; ++k) cw[hh][j][k] = 0.0f; }
    }
    __syncthreads();
    const float go = gout ? gout[0] : 1.0f;
    // per head: the two coarse rows (cur, cur + 1) this thread is accumulating into
    float aA[3][CA][NK], aB[3][CA][NK];
    int cur[3] = {-1, -1, -1};
#pragma unroll
    for (int hh = 0; hh < 3; ++hh)
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int k = 0; k < NK; ++k) { aA[hh][c][k] = 0.0f; aB[hh][c][k] = 0.0f; }
    auto flush_row = [&](int hh, int row, const float (&a)[CA][NK]) {
        const int hl = P.ld[hh][1], wl = P.ld[hh][2];
        if ((unsigned)row >= (unsigned)hl) return;
#pragma unroll
        for (int c = 0; c < CA; ++c)
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (i0[hh] + k < wl && a[c][k] != 0.0f) atomicAdd(accw + aoff[hh] + ((long)c * hl + row) * wl + i0[hh] + k, a[c][k]);
    };
    int yn[4] = {0, 0, 0, 0};
    float4 ln[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ln[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int Y) {
        const long o = ((long)Z * P.H + Y) * P.W + X0;
        vx_lab4(lab, lab_kind, (long)b * V + o, yn);
#pragma unroll
        for (int c = 0; c < C; ++c) ln[c] = vx_ld4(l0p, ((long)b * C + c) * V + o);
    };
    if (y0 < y1) fetch(y0);
    for (int Y = y0; Y < y1; ++Y) {          // (the row's global operands one row ahead of the arithmetic; two rows ahead -- two register buffers -- costs 60 VGPRs here and gains nothing)
        int y[4] = {yn[0], yn[1], yn[2], yn[3]};
        float z0[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c) { z0[c][0] = ln[c].x; z0[c][1] = ln[c].y; z0[c][2] = ln[c].z; z0[c][3] = ln[c].w; }
        if (Y + 1 < y1) fetch(Y + 1);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (h < P.nh) {
                float z[C][4];
                int a1 = 0; float l1 = 0.0f;
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[c][j] = z0[c][j];
                } else vx_ds_interp_col<C, NK>(P, h - 1, slices + loff[h - 1], Y, i0[h - 1], cw[h - 1], z, a1, l1);
                const float* __restrict__ coef_h = coef + (long)h * coef_stride;
                const float wce = coef_h[0];
                float al[C], be[C];
#pragma unroll
                for (int c = 0; c < C; ++c) { al[c] = coef_h[1 + ((long)b * C + c) * 2]; be[c] = coef_h[2 + ((long)b * C + c) * 2]; }
                float g[C][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float dot = 0.0f;
                    if constexpr (C == 2) {          // logistic form: one exponential per voxel
                        const float dd = z[1][j] - z[0][j];
                        const float e = __expf(-fabsf(dd)), inv = __builtin_amdgcn_rcpf(1.0f + e);
                        const float pb = inv, ps = e * inv;
                        z[1][j] = dd >= 0.0f ? pb : ps;
                        z[0][j] = dd >= 0.0f ? ps : pb;
                    } else {
                        float mx = z[0][j];
#pragma unroll
                        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c][j]);
                        float se = 0.0f;
#pragma unroll
                        for (int c = 0; c < C; ++c) { z[c][j] = __expf(z[c][j] - mx); se += z[c][j]; }
                        const float inv = __builtin_amdgcn_rcpf(se);
#pragma unroll
                        for (int c = 0; c < C; ++c) z[c][j] *= inv;
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) dot = fmaf(z[c][j], (c == y[j] ? al[c] : 0.0f) + be[c], dot);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float gg = (c == y[j] ? al[c] : 0.0f) + be[c];
                        g[c][j] = go * (wce * (z[c][j] - (c == y[j] ? 1.0f : 0.0f)) + z[c][j] * (gg - dot));
                    }
                }
                if (h == 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        vx_st4(dl0p, ((long)b * C + c) * V + ((long)Z * P.H + Y) * P.W + X0, make_float4(g[c][0], g[c][1], g[c][2], g[c][3]));
                } else {
                    const int hh = h - 1;
                    if (a1 != cur[hh]) {                                   // the row pair moves on (by one row, every 1 / scale rows): the finished coarse row leaves the registers
                        if (cur[hh] >= 0) flush_row(hh, cur[hh], aA[hh]);
                        if (a1 == cur[hh] + 1) {
#pragma unroll
                            for (int c = 0; c < CA; ++c)
#pragma unroll
                                for (int k = 0; k < NK; ++k) { aA[hh][c][k] = aB[hh][c][k]; aB[hh][c][k] = 0.0f; }
                        } else {
                            if (cur[hh] >= 0) flush_row(hh, cur[hh] + 1, aB[hh]);
#pragma unroll
                            for (int c = 0; c < CA; ++c)
#pragma unroll
                                for (int k = 0; k < NK; ++k) { aA[hh][c][k] = 0.0f; aB[hh][c][k] = 0.0f; }
                        }
                        cur[hh] = a1;
                    }
                    const float k1 = 1.0f - l1;
#pragma unroll
                    for (int c = 0; c < CA; ++c)
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            float ck = 0.0f;
#pragma unroll
                            for (int j = 0; j < 4; ++j) ck = fmaf(cw[hh][j][k], g[c + c_lo][j], ck);
                            aA[hh][c][k] = fmaf(k1, ck, aA[hh][c][k]);
                            aB[hh][c][k] = fmaf(l1, ck, aB[hh][c][k]);
                        }
                }
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < 3; ++hh)
        if (hh < nlow && cur[hh] >= 0) { flush_row(hh, cur[hh], aA[hh]); flush_row(hh, cur[hh] + 1, aB[hh]); }
    __syncthreads();
    for (int hh = 0; hh < nlow; ++hh) {
        const int hl = P.ld[hh][1], wl = P.ld[hh][2], n = CA * hl * wl;
        for (int e = threadIdx.x; e < n; e += 256) {
            const int c = e / (hl * wl), r = e - c * hl * wl;
            const float s = (vx_ds_lds[aoff[hh] + e] + vx_ds_lds[nacc + aoff[hh] + e]) + (vx_ds_lds[2 * nacc + aoff[hh] + e] + vx_ds_lds[3 * nacc + aoff[hh] + e]);
            P.t2[hh][((((long)b * C + c + c_lo) * P.D + Z) * P.nsplit + part) * hl * wl + r] = s;
            if (c_lo) P.t2[hh][((((long)b * C) * P.D + Z) * P.nsplit + part) * hl * wl + r] = -s;
        }
    }
}

// does the column-owner map cover this geometry?  W/4 <= 64, every head narrower than the volume, and a quad never touching more than NK coarse columns
// (returns NK = 3 or 4, or 0)
static int g_ds_columns = -1;
extern "C" int vx_seg_loss_ds_set_columns(int on) { g_ds_columns = on ? 1 : 0; return 0; }
static int vx_ds_columns_nk(const VxDs& P) {
    if (g_ds_columns < 0) { const char* e = getenv("VELOXSEG_DS_COLUMNS"); g_ds_columns = (e && e[0] == '0') ? 0 : 1; }
    if (!g_ds_columns || P.nh < 2) return 0;
    const int W4 = P.W >> 2;
    if (W4 <= 0 || W4 > 64) return 0;          // (W/4 = 24 / 12 -- the 96^3 patches: 256 / (W/4) row groups, the block's last 256 % (W/4) threads get no rows)
    int nk = 3;
    for (int hh = 0; hh < P.nh - 1; ++hh) {
        const int wl = P.ld[hh][2];
        if (wl >= P.W || P.ld[hh][1] > P.H) return 0;
        const float ratio = P.W > 1 ? (float)(wl - 1) / (float)(P.W - 1) : 0.0f;
        for (int X0 = 0; X0 < P.W; X0 += 4) {
            const int a0 = (int)(ratio * (float)X0);
            const int a3 = (int)(ratio * (float)(X0 + 3));
            const int b3 = a3 + (a3 < wl - 1 ? 1 : 0);
            const int span = b3 - a0 + 1;
            if (span > 4) return 0;
            if (span > nk) nk = span;
        }
    }
    return nk;
}

// ------------------------------------------------------------------------------------------------------------------------------ host
static int vx_ds_fill(VxDs& P, const float* l0, const float* l1, const float* l2, const float* l3, const int* dims, int nh, int B, int C, int D, int H, int W, const char* who) {
    VX_REQUIRE(nh >= 1 && nh <= 4 && l0 && B > 0 && (C == 2 || C == 3 || C == 4) && D > 0 && H > 0 && W > 0, "%s: bad args (C must be 2..4)", who);
    VX_REQUIRE((W & 3) == 0 && (W >> 2) <= 64, "%s: W must be a multiple of 4, at most 256 (got %d)", who, W);
    P.l0 = l0; P.low[0] = l1; P.low[1] = l2; P.low[2] = l3; P.nh = nh; P.B = B; P.D = D; P.H = H; P.W = W;
    for (int hh = 0; hh < 3; ++hh) {
        for (int k = 0; k < 3; ++k) P.ld[hh][k] = hh < nh - 1 ? dims[3 * hh + k] : 1;
        P.rh[hh] = H > 1 ? (float)(P.ld[hh][1] - 1) / (float)(H - 1) : 0.0f;
        P.rw[hh] = W > 1 ? (float)(P.ld[hh][2] - 1) / (float)(W - 1) : 0.0f;
        if (hh < nh - 1) VX_REQUIRE(P.low[hh] && P.ld[hh][0] > 0 && P.ld[hh][0] <= D && P.ld[hh][1] > 0 && P.ld[hh][1] <= H && P.ld[hh][2] > 0 && P.ld[hh][2] <= W,
                                    "%s: head %d: bad grid", who, hh + 1);
    }
    return 0;
}

// LDS floats of the Z-interpolated slices of heads 1.. (vx_ds_stage_slice)
static size_t vx_ds_slice_floats(const VxDs& P, int C) {
    size_t n = 0;
    for (int hh = 0; hh < P.nh - 1; ++hh) n += ((size_t)C * P.ld[hh][1] * P.ld[hh][2] + 3) & ~(size_t)3;
    return n;
}

// bf16 storage mode (the *_h entries): head 0 and its gradient are vx_bf16 arrays -- column-owner kernels only; call-scoped flag
static thread_local int t_ds_h16 = 0;
namespace { struct DsH16Scope { int prev; explicit DsH16Scope(int h) : prev(t_ds_h16) { t_ds_h16 = h; } ~DsH16Scope() { t_ds_h16 = prev; } }; }

// 1 when the fused kernels cover this shape (else: up-sample + vx_seg_loss_fwd / _bwd4)
extern "C" int vx_seg_loss_ds_ok(int C, int D, int H, int W) {
    (void)D; (void)H;
    return (C == 2 || C == 3 || C == 4) && (W & 3) == 0 && (W >> 2) <= 64;      // (a wave takes floor(64 / (W/4)) rows per step: 24-quad rows -- 96^3 patches -- leave 16 lanes idle)
}

extern "C" int vx_seg_loss_ds_fwd(const float* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                                  double* acc, int B, int C, int D, int H, int W, void* stream) {
    VxDs P = {};
    if (int e = vx_ds_fill(P, l0, l1, l2, l3, low_dims, nh, B, C, D, H, W, "vx_seg_loss_ds_fwd")) return e;
    VX_REQUIRE(labels && acc && lab_kind >= 0 && lab_kind <= 2, "vx_seg_loss_ds_fwd: bad labels / accumulator");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double) * (size_t)nh * (1 + (size_t)B * C * 3), st) != hipSuccess) VX_FAIL(-2, "vx_seg_loss_ds_fwd: memset failed");
    // blocks per Z slice: every block stages the slice's interpolated grids, so a split repeats that work -- worth it only while a part keeps >= 4 sweeps of 256 quads
    // (measured alone: 96^3 x 4: 1 / 2 / 4 / 8 parts = 75 / 65 / 93 / 137 us; 128^3 x 4: 135 / 136 / 142 / 175 us; in the step 128^3 is 0.25 % faster with 4 parts than with 1)
    P.nsplit = 1;
    while ((long)D * B * P.nsplit < 2048 && (long)H * (W >> 2) / (P.nsplit * 2) >= 1024) P.nsplit *= 2;
    const dim3 grid(D * P.nsplit, B), blk(256);
    size_t shm = vx_ds_slice_floats(P, C) * sizeof(float);
    P.stage = shm <= 120 * 1024;          // else: 8-tap gathers from global memory
    if (!P.stage) shm = 0;
#define VX_DS_FWD(CC)                                                                                                                 \
    {                                                                                                                                 \
        static size_t cap = 64 * 1024;          /* the kernel also has static LDS: raise the dynamic limit only as far as needed */            \
        if (shm > cap) { if (hipFuncSetAttribute((const void*)vx_seg_loss_ds_fwd_k<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) { (void)hipGetLastError(); VX_FAIL(-2, "vx_seg_loss_ds_fwd: cannot reserve %zu bytes of LDS", shm); } cap = shm; } \
        vx_seg_loss_ds_fwd_k<CC><<<grid, blk, shm, st>>>(P, labels, lab_kind, acc);                                                   \
    }
    const int nk = P.stage ? vx_ds_columns_nk(P) : 0;
    const size_t shm_col = vx_ds_slice_pad_floats(P, C) * sizeof(float);
    if (nk && shm_col <= 48 * 1024) {          // column-owner map (see above)
        // two blocks per slice (VELOXSEG_DS_FWD_SPLIT; measured at 128^3 x 4: 1 / 2 / 4 parts = 64 / 61 / 91 us): a thread's set-up (column weights) is amortised over
        // its rows, and every block ends in ~20 double atomics on the same few accumulators
        static int fsplit = -1;
        if (fsplit < 0) { const char* e = getenv("VELOXSEG_DS_FWD_SPLIT"); fsplit = e ? atoi(e) : 2; if (fsplit < 1 || fsplit > 8) fsplit = 2; }
        P.nsplit = fsplit;
        { const char* e = getenv("VX_DS_DBG"); P.dbg = e ? atoi(e) : 0; }
        const dim3 grid(D * P.nsplit, B);
#define VX_DS_FWD_COL(CC)                                                                                                              \
        { if (t_ds_h16) { if (nk == 3) vx_seg_loss_ds_fwd_col_k<CC, 3, vx_bf16><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, acc);            \
                          else vx_seg_loss_ds_fwd_col_k<CC, 4, vx_bf16><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, acc); }                  \
          else if (nk == 3) vx_seg_loss_ds_fwd_col_k<CC, 3><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, acc);                       \
          else vx_seg_loss_ds_fwd_col_k<CC, 4><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, acc); }
        if (C == 2) VX_DS_FWD_COL(2) else if (C == 3) VX_DS_FWD_COL(3) else VX_DS_FWD_COL(4)
#undef VX_DS_FWD_COL
        VX_LAUNCH_CHECK("vx_seg_loss_ds_fwd (columns)");
        return 0;
    }
    VX_REQUIRE(!t_ds_h16, "vx_seg_loss_ds_fwd_h: a 16-bit head 0 needs the column-owner kernels (vx_seg_loss_ds_h16_ok)");
    if (C == 2) VX_DS_FWD(2) else if (C == 3) VX_DS_FWD(3) else VX_DS_FWD(4)
#undef VX_DS_FWD
    VX_LAUNCH_CHECK("vx_seg_loss_ds_fwd");
    return 0;
}

// workspace (floats) of vx_seg_loss_ds_bwd: the (B, C, D x row parts, h, w) partial gradients of heads 1.., sized for the ACTIVE split (resolved once per process:
// 1 by default; VELOXSEG_DS_BWD_SPLIT is an A/B knob that measured slower overall -- sizing for its maximum held 8 x the memory inside the capture pools)
#define VX_DS_MAX_SPLIT 8
static int ds_bwd_split() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VELOXSEG_DS_BWD_SPLIT"); const int u = e ? atoi(e) : 1; v = (u >= 1 && u <= VX_DS_MAX_SPLIT) ? u : 1; }
    return v;
}
extern "C" int vx_seg_loss_ds_ws_floats(const int* low_dims, int nh, int B, int C, int D) {
    long n = 0;
    for (int hh = 0; hh < nh - 1; ++hh) n += (long)B * C * D * ds_bwd_split() * low_dims[3 * hh + 1] * low_dims[3 * hh + 2];
    VX_REQUIRE(n < 0x7fffffffL, "vx_seg_loss_ds_ws_floats: workspace too large");
    return (int)n;
}

extern "C" int vx_seg_loss_ds_bwd(const float* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                                  const float* coef, int coef_stride, const float* gout, float* dl0, float* dl1, float* dl2, float* dl3, float* ws,
                                  int B, int C, int D, int H, int W, void* stream) {
    VxDs P = {};
    if (int e = vx_ds_fill(P, l0, l1, l2, l3, low_dims, nh, B, C, D, H, W, "vx_seg_loss_ds_bwd")) return e;
    VX_REQUIRE(labels && coef && dl0 && (nh < 2 || (dl1 && ws)) && (nh < 3 || dl2) && (nh < 4 || dl3), "vx_seg_loss_ds_bwd: null pointer");
    P.dl0 = dl0;
    float* outs[3] = {dl1, dl2, dl3};
    VxDsZ Zp = {};
    int nacc = 0;
    long off = 0, total = 0;
    for (int hh = 0; hh < nh - 1; ++hh) {
        P.t2[hh] = ws + off;
        Zp.t2[hh] = ws + off;
        Zp.out[hh] = outs[hh];
        for (int k = 0; k < 3; ++k) Zp.ld[hh][k] = P.ld[hh][k];
        Zp.n[hh] = (long)B * C * P.ld[hh][0] * P.ld[hh][1] * P.ld[hh][2];
        total += Zp.n[hh];
        off += (long)B * C * D * ds_bwd_split() * P.ld[hh][1] * P.ld[hh][2];
        nacc += C * P.ld[hh][1] * P.ld[hh][2];          // (two classes + staged slices: the kernel uses half of it)
    }
    const int RPW = 64 / (W >> 2);
    // blocks per Z slice (row parts): a block is one long serial program per wave (soft-max side, then the W / H adjoints through LDS) -- with one block per slice the
    // 128^3 x 4 patch is 512 blocks = 2 waves per SIMD and the kernel waits on its LDS round trips (198 us); every part keeps >= 2 steps of 4 waves x RPW rows
    P.nsplit = 1;
    // (measured, 128^3 x 4: 1 / 4 parts = 181 / 158 us, but the D adjoint behind it then reads 4 x the partial sums: 20 -> 57 us -- the default stays at one part)
    if (H / ds_bwd_split() >= 1) P.nsplit = ds_bwd_split();          // (the workspace above is laid out for exactly this many parts)
    else VX_FAIL(-1, "vx_seg_loss_ds_bwd: VELOXSEG_DS_BWD_SPLIT=%d exceeds the %d rows of a slice", ds_bwd_split(), H);
    Zp.nlow = nh - 1; Zp.BC = B * C; Zp.D = D; Zp.nsplit = P.nsplit;
    size_t ntab = 0;
    for (int hh = 0; hh < nh - 1; ++hh) {
        const int wl = P.ld[hh][2];
        const float ratio = W > 1 ? (float)(wl - 1) / (float)(W - 1) : 0.0f;
        const int bwd_ = (wl == W || ratio <= 0.0f) ? W : ((int)(2.0f / ratio) + 4 < W ? (int)(2.0f / ratio) + 4 : W);
        ntab += (size_t)wl * bwd_;
    }
    const size_t shm_base = ((size_t)4 * nacc + (size_t)4 * RPW * 3 * C * W + ((ntab + 3) & ~(size_t)3) + (size_t)3 * W) * sizeof(float);
    P.stage = shm_base + vx_ds_slice_floats(P, C) * sizeof(float) <= 150 * 1024;
    const size_t shm = shm_base + (P.stage ? vx_ds_slice_floats(P, C) * sizeof(float) : 0);
    VX_REQUIRE(shm <= 150 * 1024, "vx_seg_loss_ds_bwd: the low-resolution grids do not fit LDS (%zu bytes)", shm);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(D * P.nsplit, B), blk(256);
#define VX_DS_BWD(CC)                                                                                                                 \
    {                                                                                                                                 \
        static size_t cap = 64 * 1024;                                                                                                \
        if (shm > cap) { if (hipFuncSetAttribute((const void*)vx_seg_loss_ds_bwd_k<CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) { (void)hipGetLastError(); VX_FAIL(-2, "vx_seg_loss_ds_bwd: cannot reserve %zu bytes of LDS", shm); } cap = shm; } \
        vx_seg_loss_ds_bwd_k<CC><<<grid, blk, shm, st>>>(P, labels, lab_kind, coef, coef_stride, gout, nacc);                         \
    }
    const int nk = P.stage ? vx_ds_columns_nk(P) : 0;
    const size_t shm_col = ((size_t)4 * nacc + vx_ds_slice_pad_floats(P, C)) * sizeof(float);
    if (nk && shm_col <= 48 * 1024) {
#define VX_DS_BWD_COL(CC)                                                                                                              \
        { if (t_ds_h16) { if (nk == 3) vx_seg_loss_ds_bwd_col_k<CC, 3, vx_bf16><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, coef, coef_stride, gout, nacc);   \
                          else vx_seg_loss_ds_bwd_col_k<CC, 4, vx_bf16><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, coef, coef_stride, gout, nacc); }         \
          else if (nk == 3) vx_seg_loss_ds_bwd_col_k<CC, 3><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, coef, coef_stride, gout, nacc);   \
          else vx_seg_loss_ds_bwd_col_k<CC, 4><<<grid, blk, shm_col, st>>>(P, labels, lab_kind, coef, coef_stride, gout, nacc); }
        if (C == 2) VX_DS_BWD_COL(2) else if (C == 3) VX_DS_BWD_COL(3) else VX_DS_BWD_COL(4)
#undef VX_DS_BWD_COL
    } else if (t_ds_h16) { VX_FAIL(-1, "vx_seg_loss_ds_bwd_h: a 16-bit head 0 needs the column-owner kernels (vx_seg_loss_ds_h16_ok)"); }
    else
    if (C == 2) VX_DS_BWD(2) else if (C == 3) VX_DS_BWD(3) else VX_DS_BWD(4)
#undef VX_DS_BWD
    if (nh > 1) vx_seg_loss_ds_adj_z_k<<<dim3(vx_cdiv(total, 32)), blk, 0, st>>>(Zp);
    VX_LAUNCH_CHECK("vx_seg_loss_ds_bwd");
    return 0;
}


// ---- bf16 storage mode: head 0 (the full-resolution logits) and its gradient as 16-bit arrays; the low-resolution heads, the accumulators and the workspace stay fp32
// 1 when both directions would run the column-owner kernels at this geometry (the only ones with 16-bit instances)
extern "C" int vx_seg_loss_ds_h16_ok(const int* low_dims, int nh, int B, int C, int D, int H, int W) {
    if (!vx_seg_loss_ds_ok(C, D, H, W) || nh < 2 || nh > 4 || !low_dims) return 0;
    VxDs P = {};
    static const float dummy = 0.0f;
    if (vx_ds_fill(P, &dummy, &dummy, &dummy, &dummy, low_dims, nh, B, C, D, H, W, "vx_seg_loss_ds_h16_ok") != 0) return 0;
    if (vx_ds_slice_floats(P, C) * sizeof(float) > 120 * 1024) return 0;                     // (forward: P.stage)
    const int nk = vx_ds_columns_nk(P);
    if (!nk) return 0;
    int nacc = 0;
    for (int hh = 0; hh < nh - 1; ++hh) nacc += C * P.ld[hh][1] * P.ld[hh][2];
    if (vx_ds_slice_pad_floats(P, C) * sizeof(float) > 48 * 1024) return 0;
    if (((size_t)4 * nacc + vx_ds_slice_pad_floats(P, C)) * sizeof(float) > 48 * 1024) return 0;
    // backward: P.stage of vx_seg_loss_ds_bwd
    const int RPW = 64 / (W >> 2);
    size_t ntab = 0;
    for (int hh = 0; hh < nh - 1; ++hh) {
        const int wl = P.ld[hh][2];
        const float ratio = W > 1 ? (float)(wl - 1) / (float)(W - 1) : 0.0f;
        const int bwd_ = (wl == W || ratio <= 0.0f) ? W : ((int)(2.0f / ratio) + 4 < W ? (int)(2.0f / ratio) + 4 : W);
        ntab += (size_t)wl * bwd_;
    }
    const size_t shm_base = ((size_t)4 * nacc + (size_t)4 * RPW * 3 * C * W + ((ntab + 3) & ~(size_t)3) + (size_t)3 * W) * sizeof(float);
    return shm_base + vx_ds_slice_floats(P, C) * sizeof(float) <= 150 * 1024 ? 1 : 0;
}
extern "C" int vx_seg_loss_ds_fwd_h(const void* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                                    double* acc, int B, int C, int D, int H, int W, int l0_h16, void* stream) {
    DsH16Scope sc(l0_h16 ? 1 : 0);
    return vx_seg_loss_ds_fwd((const float*)l0, l1, l2, l3, low_dims, nh, labels, lab_kind, acc, B, C, D, H, W, stream);
}
extern "C" int vx_seg_loss_ds_bwd_h(const void* l0, const float* l1, const float* l2, const float* l3, const int* low_dims, int nh, const void* labels, int lab_kind,
                                    const float* coef, int coef_stride, const float* gout, void* dl0, float* dl1, float* dl2, float* dl3, float* ws,
                                    int B, int C, int D, int H, int W, int l0_h16, void* stream) {
    DsH16Scope sc(l0_h16 ? 1 : 0);
    return vx_seg_loss_ds_bwd((const float*)l0, l1, l2, l3, low_dims, nh, labels, lab_kind, coef, coef_stride, gout, (float*)dl0, dl1, dl2, dl3, ws, B, C, D, H, W, stream);
}
