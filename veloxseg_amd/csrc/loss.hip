// Loss-side kernels for gfx950 (fp32 data, fp64 accumulators).
// Reference: utils/loss.py:30-66 (CE + MONAI DiceLoss deep supervision, MSE reconstruction, SDKT Gram MSE),
//            model/components/common_function.py:8-14 (Gram), model/VeloxSeg.py:177-184 (trilinear up-sampling of
//            deep-supervision heads, align_corners=True).
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

#define VX_MAXC 8   // max classes handled in registers

__device__ __forceinline__ int vx_label(const void* lab, int kind, long i) {
    if (kind == 0) return (int)((const long long*)lab)[i];
    if (kind == 1) return ((const int*)lab)[i];
    return (int)((const unsigned char*)lab)[i];
}

// ---------------------------------------------------------------------------------------------
// seg loss forward: per head h: ce_sum, and per (b,c): I = sum p*t, P = sum p, T = sum t
// acc layout (double): head h at h*(1 + B*C*3): [ce_sum, (I,P,T) x (b,c)]
// grid (chunks, B); each thread strides over voxels of batch b
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vx_seg_loss_fwd_k(const float* __restrict__ l0, const float* __restrict__ l1, const float* __restrict__ l2,
                                                         const float* __restrict__ l3, int nh, const void* __restrict__ lab, int lab_kind,
                                                         double* __restrict__ acc, int B, int C, long V) {
    const int b = blockIdx.y;
    __shared__ float red[4];
    const float* heads[4] = {l0, l1, l2, l3};
    for (int h = 0; h < nh; ++h) {
        const float* __restrict__ lg = heads[h] + (long)b * C * V;
        float ce = 0.0f, I[VX_MAXC], P[VX_MAXC], T[VX_MAXC];
#pragma unroll
        for (int c = 0; c < VX_MAXC; ++c) { I[c] = 0.0f; P[c] = 0.0f; T[c] = 0.0f; }
        for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < V; v += (long)gridDim.x * 256) {
            const int y = vx_label(lab, lab_kind, (long)b * V + v);
            float z[VX_MAXC], mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < VX_MAXC; ++c) if (c < C) { z[c] = lg[(long)c * V + v]; mx = fmaxf(mx, z[c]); }
            float se = 0.0f, zy = 0.0f;
#pragma unroll
            for (int c = 0; c < VX_MAXC; ++c) if (c < C) { if (c == y) zy = z[c]; z[c] = expf(z[c] - mx); se += z[c]; }
            const float inv = 1.0f / se;
            // cross entropy through log-sum-exp, as nn.CrossEntropyLoss computes it: grows linearly with the logit gap instead of saturating at
            // -log(FLT_MIN) like -log(softmax) does (labels outside [0, C) contribute nothing; the python surface validates the range)
            if ((unsigned)y < (unsigned)C) ce += (mx + logf(se)) - zy;
#pragma unroll
            for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
                const float p = z[c] * inv;
                P[c] += p;
                if (c == y) { I[c] += p; T[c] += 1.0f; }
            }
        }
        double* __restrict__ ah = acc + (long)h * (1 + (long)B * C * 3);
        float s = vx_block_sum_256(ce, red);
        if (threadIdx.x == 0) atomicAdd(ah, (double)s);
        for (int c = 0; c < C; ++c) {
            float vI = 0.f, vP = 0.f, vT = 0.f;
#pragma unroll
            for (int k = 0; k < VX_MAXC; ++k) if (k == c) { vI = I[k]; vP = P[k]; vT = T[k]; }
            vI = vx_block_sum_256(vI, red);
            vP = vx_block_sum_256(vP, red);
            vT = vx_block_sum_256(vT, red);
            if (threadIdx.x == 0) {
                double* d = ah + 1 + ((long)b * C + c) * 3;
                atomicAdd(d, (double)vI); atomicAdd(d + 1, (double)vP); atomicAdd(d + 2, (double)vT);
            }
        }
    }
}

// Vectorised variant (V % 4 == 0, C compile-time): 4 voxels per thread and iteration as float4 rows, all heads in ONE sweep so the labels
// are read once, wave-shuffle + LDS block reduction of the nh*(1+2C)+C sums, one fp64 atomic per sum and block.
template <int C>
__global__ void __launch_bounds__(256) vx_seg_loss_fwd4_k(const float* __restrict__ l0, const float* __restrict__ l1, const float* __restrict__ l2,
                                                          const float* __restrict__ l3, int nh, const void* __restrict__ lab, int lab_kind,
                                                          double* __restrict__ acc, int B, long V) {
    constexpr int NS = 1 + 2 * C;                       // per head: ce, I[C], P[C]
    const int b = blockIdx.y;
    const float* heads[4] = {l0, l1, l2, l3};
    float S[4][NS], T[C];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int k = 0; k < NS; ++k) S[h][k] = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) T[c] = 0.0f;
    const long V4 = V >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < V4; q += (long)gridDim.x * 256) {
        int y[4];
        if (lab_kind == 0) {
            const longlong2 a = ((const longlong2*)lab)[((long)b * V >> 1) + 2 * q], c2 = ((const longlong2*)lab)[((long)b * V >> 1) + 2 * q + 1];
            y[0] = (int)a.x; y[1] = (int)a.y; y[2] = (int)c2.x; y[3] = (int)c2.y;
        } else if (lab_kind == 1) {
            const int4 a = ((const int4*)lab)[((long)b * V >> 2) + q];
            y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
        } else {
            const uchar4 a = ((const uchar4*)lab)[((long)b * V >> 2) + q];
            y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) T[c] += (y[j] == c) ? 1.0f : 0.0f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            if (h < nh) {
                const float4* __restrict__ lg = (const float4*)(heads[h] + (long)b * C * V);
                float z[C][4];
#pragma unroll
                for (int c = 0; c < C; ++c) { const float4 t = lg[(long)c * V4 + q]; z[c][0] = t.x; z[c][1] = t.y; z[c][2] = t.z; z[c][3] = t.w; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mx = z[0][j];
#pragma unroll
                    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c][j]);
                    float e[C], se = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) { e[c] = expf(z[c][j] - mx); se += e[c]; }
                    const float inv = 1.0f / se;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float pc = e[c] * inv;
                        S[h][1 + C + c] += pc;
                        if (c == y[j]) { S[h][1 + c] += pc; S[h][0] += (mx + logf(se)) - z[c][j]; }     // CE = logsumexp - z_y (nn.CrossEntropyLoss)
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
            if (h < nh) {
                double* __restrict__ ah = acc + (long)h * (1 + (long)B * C * 3);
                if (r == 0) atomicAdd(ah, v);
                else if (r <= C) atomicAdd(ah + 1 + ((long)b * C + (r - 1)) * 3, v);            // I
                else atomicAdd(ah + 1 + ((long)b * C + (r - 1 - C)) * 3 + 1, v);                // P
            }
        } else {
            const int c = k - 4 * NS;
            for (int h = 0; h < nh; ++h) atomicAdd(acc + (long)h * (1 + (long)B * C * 3) + 1 + ((long)b * C + c) * 3 + 2, v);   // T (same for every head)
        }
    }
}

// squared-difference sum -> acc[0] (double)
__global__ void __launch_bounds__(256) vx_sqdiff_sum_k(const float* __restrict__ a, const float* __restrict__ b, long n, double* __restrict__ acc) {
    float s = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float d = a[i] - b[i]; s = fmaf(d, d, s); }
    __shared__ float red[4];
    s = vx_block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}

// finalize: scalar loss + per-(head,b,c) dice coefficients for the backward pass.
//   coef layout (float): head h at h*(1 + B*C*2): [w_ce = w_h/(B*V), (alpha, beta) x (b,c)]  with dDice_h/dp_c[v] = alpha*t + beta
//   misc: [rc_coef = 2*w_rc/N_rc, gram_coef = 2*w_f/(M*B*Cg*Cg)]
__global__ void vx_loss_finalize_k(const double* __restrict__ acc, int nh, int B, int C, long V, const float* __restrict__ wds,
                                   const double* __restrict__ rc_acc, long n_rc, float w_rc,
                                   const float* __restrict__ gs, const float* __restrict__ g0, const float* __restrict__ g1,
                                   const float* __restrict__ g2, const float* __restrict__ g3, int M, int Cg, float w_f,
                                   float* __restrict__ loss_out, float* __restrict__ coef) {
    // Gram (SDKT) term: the only O(B*Cg*Cg) part -- all 64 lanes share it, lane 0 then does the scalar bookkeeping
    double feat = 0.0;
    const long ng = (long)B * Cg * Cg;
    if (gs && M > 0) {
        const float* gm_[4] = {g0, g1, g2, g3};
        for (int m = 0; m < M; ++m) {
            double s_ = 0.0;
            for (long i = threadIdx.x; i < ng; i += 64) { const double d = (double)gs[i] - (double)gm_[m][i]; s_ += d * d; }
            feat += vx_wave_sum(s_) / (double)ng;
        }
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double eps = 1e-5;
    double total = 0.0;
    for (int h = 0; h < nh; ++h) {
        const double* ah = acc + (long)h * (1 + (long)B * C * 3);
        float* ch = coef + (long)h * (1 + (long)B * C * 2);
        const double w = (double)wds[h];
        const double ce = ah[0] / ((double)B * (double)V);
        double dice = 0.0;
        const double wd = w / ((double)B * (double)(C - 1));
        ch[0] = (float)(w / ((double)B * (double)V));
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c) {
                const double I = ah[1 + ((long)b * C + c) * 3], P = ah[2 + ((long)b * C + c) * 3], T = ah[3 + ((long)b * C + c) * 3];
                const double U = P + T + eps;
                float al = 0.0f, be = 0.0f;
                if (c >= 1) {
                    dice += 1.0 - (2.0 * I + eps) / U;
                    al = (float)(-2.0 * wd / U);
                    be = (float)(wd * (2.0 * I + eps) / (U * U));
                }
                ch[1 + ((long)b * C + c) * 2] = al;
                ch[2 + ((long)b * C + c) * 2] = be;
            }
        dice /= (double)B * (double)(C - 1);
        total += w * (ce + dice);
    }
    float* misc = coef + (long)nh * (1 + (long)B * C * 2);
    if (rc_acc) {
        total += (double)w_rc * rc_acc[0] / (double)n_rc;
        misc[0] = (float)(2.0 * (double)w_rc / (double)n_rc);
    } else misc[0] = 0.0f;
    if (gs && M > 0) {
        total += (double)w_f * feat / (double)M;
        misc[1] = (float)(2.0 * (double)w_f / ((double)M * (double)ng));
    } else misc[1] = 0.0f;
    loss_out[0] = (float)total;
}

// seg loss backward for one head: dlogit_c = gout * [ w_ce (p_c - t_c) + p_c (g_c - sum_k p_k g_k) ],  g_c = (c>=1) ? alpha*t_c + beta : 0
__global__ void __launch_bounds__(256) vx_seg_loss_bwd_k(const float* __restrict__ lg_all, const void* __restrict__ lab, int lab_kind,
                                                         const float* __restrict__ coef_h, const float* __restrict__ gout,
                                                         float* __restrict__ dl_all, int B, int C, long V) {
    const int b = blockIdx.y;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float go = gout ? gout[0] : 1.0f;
    const float* __restrict__ lg = lg_all + (long)b * C * V;
    float* __restrict__ dl = dl_all + (long)b * C * V;
    const int y = vx_label(lab, lab_kind, (long)b * V + v);
    const float wce = coef_h[0];
    float z[VX_MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < VX_MAXC; ++c) if (c < C) { z[c] = lg[(long)c * V + v]; mx = fmaxf(mx, z[c]); }
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < VX_MAXC; ++c) if (c < C) { z[c] = expf(z[c] - mx); se += z[c]; }
    const float inv = 1.0f / se;
    float g[VX_MAXC], dot = 0.0f;
#pragma unroll
    for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
        z[c] *= inv;
        const float al = coef_h[1 + ((long)b * C + c) * 2], be = coef_h[2 + ((long)b * C + c) * 2];
        g[c] = (c == y ? al : 0.0f) + be;
        dot = fmaf(z[c], g[c], dot);
    }
#pragma unroll
    for (int c = 0; c < VX_MAXC; ++c) if (c < C) dl[(long)c * V + v] = go * (wce * (z[c] - (c == y ? 1.0f : 0.0f)) + z[c] * (g[c] - dot));
}

// all deep-supervision heads in one launch: the label (8 bytes per voxel as int64) is read once instead of once per head
struct VxLoss4 { const float* lg[4]; float* dl[4]; };
template <int VEC>      // VEC = 4: four consecutive voxels per thread, 16-byte loads and stores (V % 4 == 0)
__global__ void __launch_bounds__(256) vx_seg_loss_bwd4_k(VxLoss4 P, int nh, const void* __restrict__ lab, int lab_kind, const float* __restrict__ coef, int coef_stride,
                                                          const float* __restrict__ gout, int B, int C, long V) {
    const int b = blockIdx.y;
    const long v = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (v >= V) return;
    const float go = gout ? gout[0] : 1.0f;
    int y[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) y[u] = vx_label(lab, lab_kind, (long)b * V + v + u);
    for (int h = 0; h < nh; ++h) {
        const float* __restrict__ lg = P.lg[h] + (long)b * C * V;
        float* __restrict__ dl = P.dl[h] + (long)b * C * V;
        const float* __restrict__ coef_h = coef + (long)h * coef_stride;
        const float wce = coef_h[0];
        float z[VX_MAXC][VEC], mx[VEC], se[VEC], dot[VEC];
#pragma unroll
        for (int u = 0; u < VEC; ++u) { mx[u] = -INFINITY; se[u] = 0.0f; dot[u] = 0.0f; }
#pragma unroll
        for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
            if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(lg + (long)c * V + v); z[c][0] = t.x; z[c][1 % VEC] = t.y; z[c][2 % VEC] = t.z; z[c][3 % VEC] = t.w; }
            else z[c][0] = lg[(long)c * V + v];
#pragma unroll
            for (int u = 0; u < VEC; ++u) mx[u] = fmaxf(mx[u], z[c][u]);
        }
#pragma unroll
        for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
#pragma unroll
            for (int u = 0; u < VEC; ++u) { z[c][u] = expf(z[c][u] - mx[u]); se[u] += z[c][u]; }
        }
        float al[VX_MAXC], be[VX_MAXC];
#pragma unroll
        for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
            al[c] = coef_h[1 + ((long)b * C + c) * 2]; be[c] = coef_h[2 + ((long)b * C + c) * 2];
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                z[c][u] *= 1.0f / se[u];
                dot[u] = fmaf(z[c][u], (c == y[u] ? al[c] : 0.0f) + be[c], dot[u]);
            }
        }
#pragma unroll
        for (int c = 0; c < VX_MAXC; ++c) if (c < C) {
            float o[VEC];
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                const float g = (c == y[u] ? al[c] : 0.0f) + be[c];
                o[u] = go * (wce * (z[c][u] - (c == y[u] ? 1.0f : 0.0f)) + z[c][u] * (g - dot[u]));
            }
            if (VEC == 4) *reinterpret_cast<float4*>(dl + (long)c * V + v) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
            else dl[(long)c * V + v] = o[0];
        }
    }
}

// da = gout * coef * (a - b)
__global__ void __launch_bounds__(256) vx_mse_bwd_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ coef,
                                                    const float* __restrict__ gout, float* __restrict__ da, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) da[i] = (gout ? gout[0] : 1.0f) * coef[0] * (a[i] - b[i]);
}

// dGs = gout*coef*sum_m (Gs - Gm) ; dGm = -gout*coef*(Gs - Gm)
__global__ void __launch_bounds__(256) vx_gram_mse_bwd_k(const float* __restrict__ gs, const float* __restrict__ g0, const float* __restrict__ g1,
                                                         const float* __restrict__ g2, const float* __restrict__ g3, int M, const float* __restrict__ coef,
                                                         const float* __restrict__ gout, float* __restrict__ dgs, float* __restrict__ d0, float* __restrict__ d1,
                                                         float* __restrict__ d2, float* __restrict__ d3, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float k = (gout ? gout[0] : 1.0f) * coef[0];
    const float* gm[4] = {g0, g1, g2, g3};
    float* dm[4] = {d0, d1, d2, d3};
    float s = 0.0f;
    for (int m = 0; m < M; ++m) { const float d = gs[i] - gm[m][i]; s += d; dm[m][i] = -k * d; }
    dgs[i] = k * s;
}

// ---------------------------------------------------------------------------------------------
// Gram matrix G[b,m,n] = sum_v x[b,m,v] x[b,n,v] / (C*V)    (C <= 32)
// ---------------------------------------------------------------------------------------------
#define VX_GT 64
__global__ void __launch_bounds__(256) vx_gram_fwd_k(const float* __restrict__ x, float* __restrict__ G, int C, long V, int tiles_per_block) {
    __shared__ float tile[32][VX_GT + 1];
    const int b = blockIdx.y;
    const float* __restrict__ xb = x + (long)b * C * V;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const long v_begin = (long)blockIdx.x * tiles_per_block * VX_GT;
    for (int tl = 0; tl < tiles_per_block; ++tl) {
        const long v0 = v_begin + (long)tl * VX_GT;
        if (v0 >= V) break;
        __syncthreads();
        for (int e = threadIdx.x; e < C * VX_GT; e += 256) {
            const int c = e / VX_GT, k = e % VX_GT;
            tile[c][k] = (v0 + k < V) ? xb[(long)c * V + v0 + k] : 0.0f;
        }
        __syncthreads();
        int slot = 0;
        for (int pr = threadIdx.x; pr < C * C; pr += 256, ++slot) {
            const int m = pr / C, n = pr % C;
            float s = 0.0f;
#pragma unroll 8
            for (int k = 0; k < VX_GT; ++k) s = fmaf(tile[m][k], tile[n][k], s);
            acc[slot] += s;
        }
    }
    const float norm = 1.0f / ((float)C * (float)V);
    int slot = 0;
    for (int pr = threadIdx.x; pr < C * C; pr += 256, ++slot) atomicAdd(G + (long)b * C * C + pr, acc[slot] * norm);
}

// dx[b,m,v] = sum_n (dG[b,m,n] + dG[b,n,m]) x[b,n,v] / (C*V)
__global__ void __launch_bounds__(256) vx_gram_bwd_k(const float* __restrict__ x, const float* __restrict__ dG, float* __restrict__ dx, int C, long V) {
    const int b = blockIdx.y;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* __restrict__ xb = x + (long)b * C * V + v;
    const float* __restrict__ gb = dG + (long)b * C * C;
    const float norm = 1.0f / ((float)C * (float)V);
    float xv[32];
#pragma unroll
    for (int n = 0; n < 32; ++n) xv[n] = (n < C) ? xb[(long)n * V] : 0.0f;
    for (int m = 0; m < C; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < 32; ++n) if (n < C) s = fmaf(gb[m * C + n] + gb[n * C + m], xv[n], s);
        dx[(long)b * C * V + (long)m * V + v] = s * norm;
    }
}

// ---------------------------------------------------------------------------------------------
// trilinear up-sampling, align_corners=True
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void vx_up_coord(int j, int nin, int nout, int& i0, int& i1, float& lam) {
    if (nout == nin) { i0 = j; i1 = j; lam = 0.0f; return; }
    const float ratio = nout > 1 ? (float)(nin - 1) / (float)(nout - 1) : 0.0f;
    const float s = ratio * (float)j;
    i0 = (int)s;
    lam = s - (float)i0;
    i1 = i0 + (i0 < nin - 1 ? 1 : 0);
}

__global__ void __launch_bounds__(256) vx_upsample_fwd_k(const float* __restrict__ x, float* __restrict__ out, int d, int h, int w, int D, int H, int W) {
    const long Vo = (long)D * H * W, Vi = (long)d * h * w;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= Vo) return;
    const long bc = blockIdx.y;
    const int X = (int)(v % W), Y = (int)((v / W) % H), Z = (int)(v / ((long)W * H));
    int a0, b0, a1, b1, a2, b2;
    float l0, l1, l2;
    vx_up_coord(Z, d, D, a0, b0, l0);
    vx_up_coord(Y, h, H, a1, b1, l1);
    vx_up_coord(X, w, W, a2, b2, l2);
    const float* __restrict__ xb = x + bc * Vi;
    auto T = [&](int z, int y, int xx) { return xb[((long)z * h + y) * w + xx]; };
    const float k0 = 1.0f - l0, k1 = 1.0f - l1, k2 = 1.0f - l2;
    out[bc * Vo + v] = k0 * (k1 * (k2 * T(a0, a1, a2) + l2 * T(a0, a1, b2)) + l1 * (k2 * T(a0, b1, a2) + l2 * T(a0, b1, b2))) +
                       l0 * (k1 * (k2 * T(b0, a1, a2) + l2 * T(b0, a1, b2)) + l1 * (k2 * T(b0, b1, a2) + l2 * T(b0, b1, b2)));
}

// same arithmetic, 4 consecutive outputs of a row per thread (W % 4 == 0): the z/y coordinates and the 4 source rows are shared, 16-byte stores
__global__ void __launch_bounds__(256) vx_upsample_fwd4_k(const float* __restrict__ x, float* __restrict__ out, int d, int h, int w, int D, int H, int W) {
    const int W4 = W >> 2;
    const long Vo4 = (long)D * H * W4, Vi = (long)d * h * w;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= Vo4) return;
    const long bc = blockIdx.y;
    const int X0 = (int)(q % W4) * 4, Y = (int)((q / W4) % H), Z = (int)(q / ((long)W4 * H));
    int a0, b0, a1, b1;
    float l0, l1;
    vx_up_coord(Z, d, D, a0, b0, l0);
    vx_up_coord(Y, h, H, a1, b1, l1);
    const float* __restrict__ xb = x + bc * Vi;
    const float* __restrict__ r00 = xb + ((long)a0 * h + a1) * w;
    const float* __restrict__ r01 = xb + ((long)a0 * h + b1) * w;
    const float* __restrict__ r10 = xb + ((long)b0 * h + a1) * w;
    const float* __restrict__ r11 = xb + ((long)b0 * h + b1) * w;
    const float k0 = 1.0f - l0, k1 = 1.0f - l1;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int a2, b2;
        float l2;
        vx_up_coord(X0 + j, w, W, a2, b2, l2);
        const float k2 = 1.0f - l2;
        o[j] = k0 * (k1 * (k2 * r00[a2] + l2 * r00[b2]) + l1 * (k2 * r01[a2] + l2 * r01[b2])) +
               l0 * (k1 * (k2 * r10[a2] + l2 * r10[b2]) + l1 * (k2 * r11[a2] + l2 * r11[b2]));
    }
    *(float4*)(out + (bc * (long)D * H + ((long)Z * H + Y)) * W + X0) = make_float4(o[0], o[1], o[2], o[3]);
}

// adjoint as separable 1-D passes.  pass over axis `ax` of a (n0,n1,n2) volume: out[.., j_in, ..] = sum_J A[J][j_in] * in[.., J, ..]
// one thread = one output element; threads run along the innermost axis, so reads are coalesced for ax = 0,1 (the big passes).
__global__ void __launch_bounds__(256) vx_upsample_adj_axis_k(const float* __restrict__ in, float* __restrict__ out,
                                                              int N0, int N1, int N2, int ax, int nin, int nout) {
    // input dims (N0,N1,N2) with N[ax] == nout (fine); output dims equal except axis ax -> nin (coarse)
    const int O0 = ax == 0 ? nin : N0, O1 = ax == 1 ? nin : N1, O2 = ax == 2 ? nin : N2;
    const long Vo = (long)O0 * O1 * O2, Vin = (long)N0 * N1 * N2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= Vo) return;
    const long bc = blockIdx.y;
    const int o2 = (int)(e % O2), o1 = (int)((e / O2) % O1), o0 = (int)(e / ((long)O2 * O1));
    const int t = ax == 0 ? o0 : (ax == 1 ? o1 : o2);
    const long stride = ax == 0 ? (long)N1 * N2 : (ax == 1 ? N2 : 1);
    const long base = bc * Vin + ((long)(ax == 0 ? 0 : o0) * N1 + (ax == 1 ? 0 : o1)) * N2 + (ax == 2 ? 0 : o2);
    // fine positions that touch coarse index t: src = ratio*J in (t-1, t+1)
    int lo = 0, hi = nout - 1;
    if (nout != nin && nout > 1 && nin > 1) {
        const float inv = (float)(nout - 1) / (float)(nin - 1);
        lo = (int)floorf((float)(t - 1) * inv);
        hi = (int)ceilf((float)(t + 1) * inv);
        if (lo < 0) lo = 0;
        if (hi > nout - 1) hi = nout - 1;
    } else if (nout == nin) { lo = t; hi = t; }
    float acc = 0.0f;
    for (int J = lo; J <= hi; ++J) {
        int i0, i1; float lam;
        vx_up_coord(J, nin, nout, i0, i1, lam);
        const float wgt = (i0 == t ? 1.0f - lam : 0.0f) + (i1 == t ? lam : 0.0f);
        if (wgt != 0.0f) acc = fmaf(wgt, in[base + (long)J * stride], acc);
    }
    out[bc * Vo + e] = acc;
}

// ---------------------------------------------------------------------------------------------
// fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics, amsgrad=False, maximize=False)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vx_adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

// the same update on four consecutive elements per thread (16-byte accesses; the arithmetic of an element is unchanged: IEEE division and square root as torch.optim.AdamW)
__global__ void __launch_bounds__(256) vx_adamw4_k(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v,
                                                   long n4, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g4 = g[i], p4 = p[i], m4 = m[i], v4 = v[i];
    const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    float po[4], mo[4], vo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gi = gv[k] * gscale;
        const float pi = pv[k] * (1.0f - lr * wd);
        const float mi = b1 * mv[k] + (1.0f - b1) * gi;
        const float vi = b2 * vv[k] + (1.0f - b2) * gi * gi;
        mo[k] = mi; vo[k] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        po[k] = pi - (lr / bc1) * (mi / denom);
    }
    m[i] = make_float4(mo[0], mo[1], mo[2], mo[3]);
    v[i] = make_float4(vo[0], vo[1], vo[2], vo[3]);
    p[i] = make_float4(po[0], po[1], po[2], po[3]);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
extern "C" int vx_seg_loss_fwd(const float* l0, const float* l1, const float* l2, const float* l3, int nh, const void* labels, int lab_kind,
                               double* acc, int B, int C, long V, void* stream) {
    VX_REQUIRE(nh >= 1 && nh <= 4 && l0 && labels && acc && C >= 2 && C <= VX_MAXC && B > 0 && V > 0, "vx_seg_loss_fwd: bad args (C must be 2..%d)", VX_MAXC);
    VX_REQUIRE(lab_kind >= 0 && lab_kind <= 2, "vx_seg_loss_fwd: label kind must be 0(int64) 1(int32) 2(uint8)");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double) * (size_t)nh * (1 + (size_t)B * C * 3), st) != hipSuccess) VX_FAIL(-2, "vx_seg_loss_fwd: memset failed");
    if ((V & 3) == 0 && C <= 4) {
        int chunks = vx_cdiv(V >> 2, 256 * 4);
        const int cap = B >= 4 ? 256 : 512;
        if (chunks > cap) chunks = cap;
        const dim3 grid(chunks, B), blk(256);
        if (C == 2) vx_seg_loss_fwd4_k<2><<<grid, blk, 0, st>>>(l0, l1, l2, l3, nh, labels, lab_kind, acc, B, V);
        else if (C == 3) vx_seg_loss_fwd4_k<3><<<grid, blk, 0, st>>>(l0, l1, l2, l3, nh, labels, lab_kind, acc, B, V);
        else vx_seg_loss_fwd4_k<4><<<grid, blk, 0, st>>>(l0, l1, l2, l3, nh, labels, lab_kind, acc, B, V);
        VX_LAUNCH_CHECK("vx_seg_loss_fwd");
        return 0;
    }
    int chunks = vx_cdiv(V, 256 * 8);
    if (chunks > 1024) chunks = 1024;
    hipLaunchKernelGGL(vx_seg_loss_fwd_k, dim3(chunks, B), dim3(256), 0, st, l0, l1, l2, l3, nh, labels, lab_kind, acc, B, C, V);
    VX_LAUNCH_CHECK("vx_seg_loss_fwd");
    return 0;
}

extern "C" int vx_sqdiff_sum(const float* a, const float* b, long n, double* acc, void* stream) {
    VX_REQUIRE(a && b && acc && n > 0, "vx_sqdiff_sum: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double), st) != hipSuccess) VX_FAIL(-2, "vx_sqdiff_sum: memset failed");
    int blocks = vx_cdiv(n, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(vx_sqdiff_sum_k, dim3(blocks), dim3(256), 0, st, a, b, n, acc);
    VX_LAUNCH_CHECK("vx_sqdiff_sum");
    return 0;
}

// batch-strided forms for the staged loss (engine.TrainEngine, tape mode): every reconstruction decoder adds ITS sum of squares against its own
// channels of the network input (a channel slice: b_batch_stride floats between samples) into the shared accumulator, inside its own branch
__global__ void __launch_bounds__(256) vx_sqdiff_sum_bs_k(const float* __restrict__ a, const float* __restrict__ b, long n, long bstride, double* __restrict__ acc) {
    const float* __restrict__ as = a + (long)blockIdx.y * n;
    const float* __restrict__ bs = b + (long)blockIdx.y * bstride;
    float s = 0.0f;
    if (((n | bstride) & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b) & 15) == 0)) {      // 16-byte loads, two independent quads in flight per thread (scalar loads: 1.2 TB/s)
        const long n4 = n >> 2, step = (long)gridDim.x * 256;
        const float4* __restrict__ a4 = reinterpret_cast<const float4*>(as);
        const float4* __restrict__ b4 = reinterpret_cast<const float4*>(bs);
        float s0 = 0.0f, s1 = 0.0f;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        for (; i + step < n4; i += 2 * step) {
            const float4 x0 = a4[i], y0 = b4[i], x1 = a4[i + step], y1 = b4[i + step];
            float d;
            d = x0.x - y0.x; s0 = fmaf(d, d, s0); d = x0.y - y0.y; s0 = fmaf(d, d, s0); d = x0.z - y0.z; s0 = fmaf(d, d, s0); d = x0.w - y0.w; s0 = fmaf(d, d, s0);
            d = x1.x - y1.x; s1 = fmaf(d, d, s1); d = x1.y - y1.y; s1 = fmaf(d, d, s1); d = x1.z - y1.z; s1 = fmaf(d, d, s1); d = x1.w - y1.w; s1 = fmaf(d, d, s1);
        }
        if (i < n4) {
            const float4 x0 = a4[i], y0 = b4[i];
            float d;
            d = x0.x - y0.x; s0 = fmaf(d, d, s0); d = x0.y - y0.y; s0 = fmaf(d, d, s0); d = x0.z - y0.z; s0 = fmaf(d, d, s0); d = x0.w - y0.w; s0 = fmaf(d, d, s0);
        }
        s = s0 + s1;
    } else
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float d = as[i] - bs[i]; s = fmaf(d, d, s); }
    __shared__ float red[4];
    s = vx_block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}
extern "C" int vx_sqdiff_sum_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, void* stream) {
    VX_REQUIRE(a && b && acc && n_per_sample > 0 && B > 0 && b_batch_stride >= n_per_sample, "vx_sqdiff_sum_bs: bad args");
    int blocks = vx_cdiv(n_per_sample, 256 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(vx_sqdiff_sum_bs_k, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, a, b, n_per_sample, b_batch_stride, acc);
    VX_LAUNCH_CHECK("vx_sqdiff_sum_bs");
    return 0;
}
// The same sum AND the MSE gradient in one pass (staged loss, round 5): the gradient's coefficient 2 w_rc / N_rc is known before the loss value is -- N_rc is the size of the
// network input -- so the reconstruction branch writes da = scale (a - b) while it forms its sum of squares, and its backward fan starts from that tensor: the second read of
// a and b (vx_mse_bwd_bs: 67 MB per reconstruction decoder at 128^3 x 4) and one launch per decoder go away.
// TA = element type of the reconstruction a and of its gradient da (float, or vx_bf16 in the bf16 storage mode); b = the network input (fp32)
template <typename TA>
__global__ void __launch_bounds__(256) vx_sqdiff_sum_grad_bs_k(const TA* __restrict__ a, const float* __restrict__ b, long n, long bstride, double* __restrict__ acc,
                                                               float scale, TA* __restrict__ da) {
    const TA* __restrict__ as = a + (long)blockIdx.y * n;
    const float* __restrict__ bs = b + (long)blockIdx.y * bstride;
    TA* __restrict__ ds = da + (long)blockIdx.y * n;
    float s = 0.0f;
    if (((n | bstride) & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)da) & 15) == 0)) {
        const long n4 = n >> 2, step = (long)gridDim.x * 256;
        const float4* __restrict__ b4 = reinterpret_cast<const float4*>(bs);
        float s0 = 0.0f, s1 = 0.0f;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        for (; i + step < n4; i += 2 * step) {
            const float4 x0 = vx_ld4(as, 4 * i), y0 = b4[i], x1 = vx_ld4(as, 4 * (i + step)), y1 = b4[i + step];
            const float4 e0 = make_float4(x0.x - y0.x, x0.y - y0.y, x0.z - y0.z, x0.w - y0.w), e1 = make_float4(x1.x - y1.x, x1.y - y1.y, x1.z - y1.z, x1.w - y1.w);
            s0 = fmaf(e0.x, e0.x, s0); s0 = fmaf(e0.y, e0.y, s0); s0 = fmaf(e0.z, e0.z, s0); s0 = fmaf(e0.w, e0.w, s0);
            s1 = fmaf(e1.x, e1.x, s1); s1 = fmaf(e1.y, e1.y, s1); s1 = fmaf(e1.z, e1.z, s1); s1 = fmaf(e1.w, e1.w, s1);
            vx_st4(ds, 4 * i, make_float4(scale * e0.x, scale * e0.y, scale * e0.z, scale * e0.w));
            vx_st4(ds, 4 * (i + step), make_float4(scale * e1.x, scale * e1.y, scale * e1.z, scale * e1.w));
        }
        if (i < n4) {
            const float4 x0 = vx_ld4(as, 4 * i), y0 = b4[i];
            const float4 e0 = make_float4(x0.x - y0.x, x0.y - y0.y, x0.z - y0.z, x0.w - y0.w);
            s0 = fmaf(e0.x, e0.x, s0); s0 = fmaf(e0.y, e0.y, s0); s0 = fmaf(e0.z, e0.z, s0); s0 = fmaf(e0.w, e0.w, s0);
            vx_st4(ds, 4 * i, make_float4(scale * e0.x, scale * e0.y, scale * e0.z, scale * e0.w));
        }
        s = s0 + s1;
    } else
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float d = vx_ld1(as, i) - bs[i]; s = fmaf(d, d, s); vx_st1(ds, i, scale * d); }
    __shared__ float red[4];
    s = vx_block_sum_256(s, red);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}
// a_h16 != 0: a and da are vx_bf16 arrays (bf16 storage mode: the full-resolution reconstruction and its gradient)
extern "C" int vx_sqdiff_sum_grad_bs_h(const void* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, float scale, void* da, int a_h16, void* stream) {
    VX_REQUIRE(a && b && acc && da && n_per_sample > 0 && B > 0 && b_batch_stride >= n_per_sample, "vx_sqdiff_sum_grad_bs: bad args");
    int blocks = vx_cdiv(n_per_sample, 256 * 16);
    if (blocks > 1024) blocks = 1024;
    if (a_h16) hipLaunchKernelGGL(vx_sqdiff_sum_grad_bs_k<vx_bf16>, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, (const vx_bf16*)a, b, n_per_sample, b_batch_stride, acc, scale, (vx_bf16*)da);
    else hipLaunchKernelGGL(vx_sqdiff_sum_grad_bs_k<float>, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, (const float*)a, b, n_per_sample, b_batch_stride, acc, scale, (float*)da);
    VX_LAUNCH_CHECK("vx_sqdiff_sum_grad_bs");
    return 0;
}
extern "C" int vx_sqdiff_sum_grad_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, double* acc, float scale, float* da, void* stream) {
    return vx_sqdiff_sum_grad_bs_h(a, b, n_per_sample, b_batch_stride, B, acc, scale, da, 0, stream);
}
__global__ void __launch_bounds__(256) vx_mse_bwd_bs_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ coef,
                                                       const float* __restrict__ gout, float* __restrict__ da, long n, long bstride) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const long o = (long)blockIdx.y * n + i;
        da[o] = (gout ? gout[0] : 1.0f) * coef[0] * (a[o] - b[(long)blockIdx.y * bstride + i]);
    }
}
extern "C" int vx_mse_bwd_bs(const float* a, const float* b, long n_per_sample, long b_batch_stride, int B, const float* coef, const float* gout, float* da, void* stream) {
    VX_REQUIRE(a && b && coef && da && n_per_sample > 0 && B > 0 && b_batch_stride >= n_per_sample, "vx_mse_bwd_bs: bad args");
    hipLaunchKernelGGL(vx_mse_bwd_bs_k, dim3(vx_cdiv(n_per_sample, 256), B), dim3(256), 0, (hipStream_t)stream, a, b, coef, gout, da, n_per_sample, b_batch_stride);
    VX_LAUNCH_CHECK("vx_mse_bwd_bs");
    return 0;
}

extern "C" int vx_loss_finalize(const double* seg_acc, int nh, int B, int C, long V, const float* head_weights,
                                const double* rc_acc, long n_rc, float w_rc,
                                const float* gram_seg, const float* g0, const float* g1, const float* g2, const float* g3, int M, int Cg, float w_f,
                                float* loss_out, float* coef, void* stream) {
    VX_REQUIRE(seg_acc && head_weights && loss_out && coef && nh >= 1 && nh <= 4 && M >= 0 && M <= 4, "vx_loss_finalize: bad args");
    hipLaunchKernelGGL(vx_loss_finalize_k, dim3(1), dim3(64), 0, (hipStream_t)stream, seg_acc, nh, B, C, V, head_weights, rc_acc, n_rc, w_rc,
                       gram_seg, g0, g1, g2, g3, M, Cg, w_f, loss_out, coef);
    VX_LAUNCH_CHECK("vx_loss_finalize");
    return 0;
}

extern "C" int vx_seg_loss_bwd(const float* logits, const void* labels, int lab_kind, const float* coef_head, const float* gout,
                               float* dlogits, int B, int C, long V, void* stream) {
    VX_REQUIRE(logits && labels && coef_head && dlogits && C >= 2 && C <= VX_MAXC, "vx_seg_loss_bwd: bad args");
    hipLaunchKernelGGL(vx_seg_loss_bwd_k, dim3(vx_cdiv(V, 256), B), dim3(256), 0, (hipStream_t)stream, logits, labels, lab_kind, coef_head, gout, dlogits, B, C, V);
    VX_LAUNCH_CHECK("vx_seg_loss_bwd");
    return 0;
}

// the nh <= 4 heads of vx_seg_loss_fwd in one launch; coef = the forward's coefficient block, head h at coef + h * coef_stride floats
extern "C" int vx_seg_loss_bwd4(const float* lg0, const float* lg1, const float* lg2, const float* lg3, int nh, const void* labels, int lab_kind, const float* coef,
                                int coef_stride, const float* gout, float* dl0, float* dl1, float* dl2, float* dl3, int B, int C, long V, void* stream) {
    VX_REQUIRE(nh >= 1 && nh <= 4 && lg0 && dl0 && labels && coef && C >= 2 && C <= VX_MAXC && B > 0 && V > 0, "vx_seg_loss_bwd4: bad args");
    VxLoss4 P; P.lg[0] = lg0; P.lg[1] = lg1; P.lg[2] = lg2; P.lg[3] = lg3; P.dl[0] = dl0; P.dl[1] = dl1; P.dl[2] = dl2; P.dl[3] = dl3;
    for (int h = 0; h < nh; ++h) VX_REQUIRE(P.lg[h] && P.dl[h], "vx_seg_loss_bwd4: head %d is missing", h);
    if ((V & 3) == 0) hipLaunchKernelGGL(vx_seg_loss_bwd4_k<4>, dim3(vx_cdiv(V / 4, 256), B), dim3(256), 0, (hipStream_t)stream, P, nh, labels, lab_kind, coef, coef_stride, gout, B, C, V);
    else hipLaunchKernelGGL(vx_seg_loss_bwd4_k<1>, dim3(vx_cdiv(V, 256), B), dim3(256), 0, (hipStream_t)stream, P, nh, labels, lab_kind, coef, coef_stride, gout, B, C, V);
    VX_LAUNCH_CHECK("vx_seg_loss_bwd4");
    return 0;
}

extern "C" int vx_mse_bwd(const float* a, const float* b, const float* coef, const float* gout, float* da, long n, void* stream) {
    VX_REQUIRE(a && b && coef && da && n > 0, "vx_mse_bwd: bad args");
    hipLaunchKernelGGL(vx_mse_bwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, coef, gout, da, n);
    VX_LAUNCH_CHECK("vx_mse_bwd");
    return 0;
}

extern "C" int vx_gram_mse_bwd(const float* gs, const float* g0, const float* g1, const float* g2, const float* g3, int M, const float* coef,
                               const float* gout, float* dgs, float* d0, float* d1, float* d2, float* d3, long n, void* stream) {
    VX_REQUIRE(gs && coef && dgs && M >= 1 && M <= 4 && n > 0, "vx_gram_mse_bwd: bad args");
    hipLaunchKernelGGL(vx_gram_mse_bwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gs, g0, g1, g2, g3, M, coef, gout, dgs, d0, d1, d2, d3, n);
    VX_LAUNCH_CHECK("vx_gram_mse_bwd");
    return 0;
}

extern "C" int vx_gram_fwd(const float* x, float* G, int B, int C, long V, void* stream) {
    VX_REQUIRE(x && G && B > 0 && C > 0 && C <= 32 && V > 0, "vx_gram_fwd: bad args (C <= 32)");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(G, 0, sizeof(float) * (size_t)B * C * C, st) != hipSuccess) VX_FAIL(-2, "vx_gram_fwd: memset failed");
    const long tiles = (V + VX_GT - 1) / VX_GT;
    int tpb = (int)((tiles + 255) / 256);
    if (tpb < 1) tpb = 1;
    hipLaunchKernelGGL(vx_gram_fwd_k, dim3(vx_cdiv(tiles, tpb), B), dim3(256), 0, st, x, G, C, V, tpb);
    VX_LAUNCH_CHECK("vx_gram_fwd");
    return 0;
}

extern "C" int vx_gram_bwd(const float* x, const float* dG, float* dx, int B, int C, long V, void* stream) {
    VX_REQUIRE(x && dG && dx && B > 0 && C > 0 && C <= 32 && V > 0, "vx_gram_bwd: bad args (C <= 32)");
    hipLaunchKernelGGL(vx_gram_bwd_k, dim3(vx_cdiv(V, 256), B), dim3(256), 0, (hipStream_t)stream, x, dG, dx, C, V);
    VX_LAUNCH_CHECK("vx_gram_bwd");
    return 0;
}

extern "C" int vx_upsample_trilinear_fwd(const float* x, float* out, long BC, int d, int h, int w, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && out && BC > 0 && d > 0 && h > 0 && w > 0 && D >= d && H >= h && W >= w, "vx_upsample_trilinear_fwd: bad args");
    if ((W & 3) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(vx_upsample_fwd4_k, dim3(vx_cdiv((long)D * H * (W >> 2), 256), (unsigned)BC), dim3(256), 0, (hipStream_t)stream, x, out, d, h, w, D, H, W);
    else
        hipLaunchKernelGGL(vx_upsample_fwd_k, dim3(vx_cdiv((long)D * H * W, 256), (unsigned)BC), dim3(256), 0, (hipStream_t)stream, x, out, d, h, w, D, H, W);
    VX_LAUNCH_CHECK("vx_upsample_trilinear_fwd");
    return 0;
}

extern "C" int vx_upsample_trilinear_bwd(const float* dout, float* dx, float* ws, long BC, int d, int h, int w, int D, int H, int W, void* stream) {
    VX_REQUIRE(dout && dx && ws && BC > 0 && d > 0 && h > 0 && w > 0 && D >= d && H >= h && W >= w, "vx_upsample_trilinear_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    float* t1 = ws;                                   // (BC, d, H, W)
    float* t2 = ws + BC * (long)d * H * W;            // (BC, d, h, W)
    hipLaunchKernelGGL(vx_upsample_adj_axis_k, dim3(vx_cdiv((long)d * H * W, 256), (unsigned)BC), dim3(256), 0, st, dout, t1, D, H, W, 0, d, D);
    hipLaunchKernelGGL(vx_upsample_adj_axis_k, dim3(vx_cdiv((long)d * h * W, 256), (unsigned)BC), dim3(256), 0, st, (const float*)t1, t2, d, H, W, 1, h, H);
    hipLaunchKernelGGL(vx_upsample_adj_axis_k, dim3(vx_cdiv((long)d * h * w, 256), (unsigned)BC), dim3(256), 0, st, (const float*)t2, dx, d, h, W, 2, w, W);
    VX_LAUNCH_CHECK("vx_upsample_trilinear_bwd");
    return 0;
}

extern "C" int vx_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, long step, float grad_scale, void* stream) {
    VX_REQUIRE(p && g && m && v && n > 0 && step >= 1, "vx_adamw_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const bool al = ((((unsigned long long)p) | ((unsigned long long)g) | ((unsigned long long)m) | ((unsigned long long)v)) & 15ull) == 0;
    const long n4 = al ? n >> 2 : 0;
    if (n4 > 0)
        hipLaunchKernelGGL(vx_adamw4_k, dim3(vx_cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g),
                           reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n4, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    if (n - 4 * n4 > 0)          // unaligned buffers, or the last n % 4 elements
        hipLaunchKernelGGL(vx_adamw_k, dim3(vx_cdiv(n - 4 * n4, 256)), dim3(256), 0, (hipStream_t)stream, p + 4 * n4, g + 4 * n4, m + 4 * n4, v + 4 * n4, n - 4 * n4, lr, beta1, beta2, eps,
                           weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    VX_LAUNCH_CHECK("vx_adamw_step");
    return 0;
}

thread_local char vx_err_buf[512] = {0};
extern "C" const char* vx_last_error(void) { return vx_err_buf; }
extern "C" int vx_abi_version(void) { return VX_ABI_VERSION; }
