// Fused spatial stage of the JLC block for gfx950 (reference model/components/conv_blocks.py:41-75):
//     o = x + sum_{k in 1,3,5} GELU(InstanceNorm(GroupConv3d_k(x)))                    (conv_blocks.py:51-58, 72-73)
// forward  : vx_jlc_conv_fwd  -- the three grouped convolutions from ONE LDS halo tile (the 3^3 and 1^3 taps are the inner taps of the 5^3
//                                neighbourhood), per-(b,c) partial sums of every output in the epilogue (no statistics pass)
//            vx_jlc_mid_fwd   -- folds the partial sums, o = x + sum_k GELU(IN(y_k)), and emits the partial sums of o for the channel stage
//                                (the channel stage itself is vx_mlp_fwd, mlp.hip)
// backward : vx_jlc_mid_bwd   -- d_o = dout + IN-backward(dn) (dn, partial sums: vx_mlp_bwd), t_k = d_o * GELU'(IN(y_k)) partial sums
//            vx_jlc_gk        -- g_k = InstanceNorm backward of t_k (recomputed), the gradients at the three conv outputs
//            vx_jlc_conv_bwd  -- dx = d_o + sum_k conv_k^T(g_k): three adjoint convolutions into one register tile
// (the weight gradients keep using conv_wgrad.hip).  Partial sums instead of atomics: no zero-fill launches, and the result does not
// depend on the order in which blocks finish.
//
// MI355X mapping of the convolutions: block = (b, COT output channels of one group, TD x TH x (4*TWq) voxel tile); thread = 4 voxels along W
// x COT channels x (3 kernel sizes forward) fp32 accumulators; per (ci, kd, kh) two ds_read_b128 of the input row + broadcast ds_read_b128 of
// the weights feed up to 4*(5+3+1)*COT v_fmac.  Element-wise kernels: 16-byte accesses, one (b,c) row chunk per block.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

struct VxJlc {
    const float *x, *w1, *w3, *w5, *b1, *b3, *b5;     // forward: input + the three conv weights (C, C/G, k,k,k) / biases
    const float *g1, *g3, *g5, *res;                  // backward: gradients at the conv outputs, residual gradient d_o
    float *y1, *y3, *y5, *dx;
    double* part;                                     // forward: [3][B*C][ntiles][2] (sum, sumsq)
    int B, C, G, D, H, W;
    int TD, TH, TWq, nTd, nTh, nTw, cic, nthr;
    int parts, nsp;      // input-channel split: the block's nthr threads = nsp spatial slots x parts channel parts (2 on small volumes: twice the waves per SIMD)
};

// stage the K-halo of `ncc` channels starting at channel c_first of sample b into xs[cil][HD][HH][HWp] (zero outside the volume)
template <int K>
__device__ __forceinline__ void vx_jlc_stage(const float* __restrict__ src, float* __restrict__ xs, const VxJlc& p, int b, int c_first, int ncc,
                                             int d0, int h0, int w0, int HD, int HH, int HWp) {
    constexpr int P = K / 2;
    const int tid = threadIdx.x, nthr = p.nthr;
    int hw = tid % HWp, t1 = tid / HWp;
    int hh = t1 % HH, t2 = t1 / HH;
    int hd = t2 % HD, cil = t2 / HD;
    int r = nthr;                                    // nthr decomposed in the staging index space (hw fastest): incremental addressing, no division in the loop
    const int st_hw = r % HWp; r /= HWp;
    const int st_hh = r % HH; r /= HH;
    const int st_hd = r % HD; r /= HD;
    const int st_c = r;
    const long chan = (long)p.D * p.H * p.W;
    const float* __restrict__ xb = src + ((long)b * p.C + c_first) * chan;
    const int total = ncc * HD * HH * HWp;
    constexpr int SU = 8;                            // loads in flight per thread (a rolled load -> store loop pays one L2 round trip per element)
    for (int e = tid; e < total; e += nthr * SU) {
        float v[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int id = d0 - P + hd, ih = h0 - P + hh, iw = w0 - P + hw;
            const bool ok = (e + u * nthr < total) && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float t_ = xb[ok ? cil * chan + ((long)id * p.H + ih) * p.W + iw : 0];
            v[u] = ok ? t_ : 0.0f;
            hw += st_hw; if (hw >= HWp) { hw -= HWp; ++hh; }
            hh += st_hh; if (hh >= HH) { hh -= HH; ++hd; }
            hd += st_hd; if (hd >= HD) { hd -= HD; ++cil; }
            cil += st_c;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
            if (e + u * nthr < total) xs[e + u * nthr] = v[u];
    }
}

// weight slice staging with 8 independent loads in flight per thread; `src(e)` maps the LDS element index to the global weight element
template <class F>
__device__ __forceinline__ void vx_jlc_stage_w(float* __restrict__ ws, int total, int nthr, F src) {
    for (int e0 = threadIdx.x; e0 < total; e0 += nthr * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthr; v[u] = src(e < total ? e : 0); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthr; if (e < total) ws[e] = v[u]; }
    }
}


// COT consecutive weights from LDS (one read)
template <int COT>
__device__ __forceinline__ void vx_jlc_ldw(const float* __restrict__ w, float (&wv)[COT]) {
    if constexpr (COT == 4) { const float4 t = *reinterpret_cast<const float4*>(w); wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w; }
    else if constexpr (COT == 2) { const float2 t = *reinterpret_cast<const float2*>(w); wv[0] = t.x; wv[1] = t.y; }
    else wv[0] = w[0];
}

__device__ __forceinline__ void vx_jlc_tile(const VxJlc& p, int& d0, int& h0, int& w0, int& td, int& th, int& tq, bool& active) {
    const int tile = blockIdx.x;
    const int tw_i = tile % p.nTw, th_i = (tile / p.nTw) % p.nTh, td_i = tile / (p.nTw * p.nTh);
    d0 = td_i * p.TD; h0 = th_i * p.TH; w0 = tw_i * p.TWq * 4;
    const int sp = threadIdx.x % p.nsp;
    tq = sp % p.TWq; th = (sp / p.TWq) % p.TH; td = sp / (p.TWq * p.TH);
    active = td < p.TD;
}

// --------------------------------------------------------------------------------------------------------------------- forward convolutions
template <int COT>
__global__ void __launch_bounds__(512) vx_jlc_conv_fwd_k(VxJlc p) {
    extern __shared__ __attribute__((aligned(16))) float vx_jlc_lds[];
    constexpr int NT = 125 + 27 + 1;                 // taps of the three kernels, in this order, per (ci, co)
    const int Cg = p.C / p.G;
    const int co0 = blockIdx.y * COT, g = co0 / Cg, b = blockIdx.z;
    int d0, h0, w0, td, th, tq;
    bool active;
    vx_jlc_tile(p, d0, h0, w0, td, th, tq, active);
    const int HD = p.TD + 4, HH = p.TH + 4, HWp = p.TWq * 4 + 4;
    const int plane = HD * HH * HWp;
    float* __restrict__ xs = vx_jlc_lds;
    float* __restrict__ ws = vx_jlc_lds + p.cic * plane;
    const int tid = threadIdx.x;
    const int part = tid / p.nsp;
    float a5[4][COT], a3[4][COT], a1[4][COT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < COT; ++j) { a5[u][j] = 0.0f; a3[u][j] = 0.0f; a1[u][j] = 0.0f; }
    for (int cc = 0; cc < Cg; cc += p.cic) {
        const int ncc = min(p.cic, Cg - cc);
        __syncthreads();
        vx_jlc_stage<5>(p.x, xs, p, b, g * Cg + cc, ncc, d0, h0, w0, HD, HH, HWp);
        vx_jlc_stage_w(ws, ncc * NT * COT, p.nthr, [&](int e) {
            const int j = e % COT, t = (e / COT) % NT, cil = e / (COT * NT);
            const long wr = (long)(co0 + j) * Cg + (cc + cil);
            return t < 125 ? p.w5[wr * 125 + t] : t < 152 ? p.w3[wr * 27 + (t - 125)] : p.w1[wr];
        });
        __syncthreads();
        if (active) {
            for (int cil = part; cil < ncc; cil += p.parts) {
#pragma unroll
                for (int kd = 0; kd < 5; ++kd) {
#pragma unroll
                    for (int kh = 0; kh < 5; ++kh) {
                        const float* __restrict__ xrow = xs + ((cil * HD + td + kd) * HH + th + kh) * HWp + 4 * tq;
                        const float4 xa = *reinterpret_cast<const float4*>(xrow);
                        const float4 xb = *reinterpret_cast<const float4*>(xrow + 4);
                        const float xr[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                        const float* __restrict__ wp5 = ws + (cil * NT + (kd * 5 + kh) * 5) * COT;
#pragma unroll
                        for (int kw = 0; kw < 5; ++kw) {
                            {
                                float wv[COT];
                                vx_jlc_ldw<COT>(wp5 + kw * COT, wv);
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int j = 0; j < COT; ++j) a5[u][j] = fmaf(wv[j], xr[u + kw], a5[u][j]);
                            }
                        }
                        if (kd >= 1 && kd <= 3 && kh >= 1 && kh <= 3) {
                            const float* __restrict__ wp3 = ws + (cil * NT + 125 + ((kd - 1) * 3 + (kh - 1)) * 3) * COT;
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) {
                                {
                                    float wv[COT];
                                    vx_jlc_ldw<COT>(wp3 + kw * COT, wv);
#pragma unroll
                                    for (int u = 0; u < 4; ++u)
#pragma unroll
                                        for (int j = 0; j < COT; ++j) a3[u][j] = fmaf(wv[j], xr[u + kw + 1], a3[u][j]);
                                }
                            }
                        }
                        if (kd == 2 && kh == 2) {
                            const float* __restrict__ wp1 = ws + (cil * NT + 152) * COT;
                            {
                                float wv[COT];
                                vx_jlc_ldw<COT>(wp1, wv);
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int j = 0; j < COT; ++j) a1[u][j] = fmaf(wv[j], xr[u + 2], a1[u][j]);
                            }
                        }
                    }
                }
            }
        }
    }
    for (int pk = 1; pk < p.parts; ++pk) {           // sum the channel parts through LDS, one after the other (the halo / weight tiles are dead)
        __syncthreads();
        float* __restrict__ rb = vx_jlc_lds + (long)(tid % p.nsp) * (12 * COT);
        if (part == pk) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < COT; ++j) { rb[u * COT + j] = a5[u][j]; rb[(4 + u) * COT + j] = a3[u][j]; rb[(8 + u) * COT + j] = a1[u][j]; }
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < COT; ++j) { a5[u][j] += rb[u * COT + j]; a3[u][j] += rb[(4 + u) * COT + j]; a1[u][j] += rb[(8 + u) * COT + j]; }
        }
    }
    // epilogue: bias, store, per-channel partial sums of this tile
    const int od = d0 + td, oh = h0 + th, ow = w0 + 4 * tq;
    const bool inb = active && part == 0 && od < p.D && oh < p.H;
    __syncthreads();
    float* __restrict__ red = vx_jlc_lds;            // [wave][3*COT*2]
    const int lane = tid & 63, wave = tid >> 6, nwave = (p.nsp + 63) >> 6;     // waves of part 0 hold the sums
    const long ntiles = gridDim.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* bias = k == 0 ? p.b1 : k == 1 ? p.b3 : p.b5;
        float* yk = k == 0 ? p.y1 : k == 1 ? p.y3 : p.y5;
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            const float bv = bias ? bias[co0 + j] : 0.0f;
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = (k == 0 ? a1[u][j] : k == 1 ? a3[u][j] : a5[u][j]) + bv;
            float s = 0.0f, q = 0.0f;
            if (inb) {
                float* dst = yk + ((((long)b * p.C + co0 + j) * p.D + od) * p.H + oh) * (long)p.W + ow;
                if ((p.W & 3) == 0 && ow + 3 < p.W) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { s += o[u]; q = fmaf(o[u], o[u], q); }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (ow + u < p.W) { dst[u] = o[u]; s += o[u]; q = fmaf(o[u], o[u], q); }
                }
            }
            s = vx_wave_sum(s);
            q = vx_wave_sum(q);
            if (lane == 0) { red[(wave * 3 * COT + k * COT + j) * 2] = s; red[(wave * 3 * COT + k * COT + j) * 2 + 1] = q; }
        }
    }
    __syncthreads();
    if (tid < 3 * COT) {
        double s = 0.0, q = 0.0;
        for (int wv = 0; wv < nwave; ++wv) { s += (double)red[(wv * 3 * COT + tid) * 2]; q += (double)red[(wv * 3 * COT + tid) * 2 + 1]; }
        const int k = tid / COT, j = tid - k * COT;
        double* dst = p.part + ((((long)k * p.B + b) * p.C + co0 + j) * ntiles + blockIdx.x) * 2;
        dst[0] = s;
        dst[1] = q;
    }
}

// --------------------------------------------------------------------------------------------------------------------- input gradient
// acc += conv_K^T(g): the adjoint of a "same" conv is a "same" conv with the weights transposed inside the group and flipped in space
template <int K, int COT>
__device__ __forceinline__ void vx_jlc_adj(const float* __restrict__ gsrc, const float* __restrict__ w, const VxJlc& p, float* __restrict__ lds, int b, int g, int ci0,
                                           int d0, int h0, int w0, int td, int th, int tq, bool active, float (&acc)[4][COT]) {
    constexpr int K3 = K * K * K;
    const int Cg = p.C / p.G;
    const int HD = p.TD + K - 1, HH = p.TH + K - 1, HWp = p.TWq * 4 + 4;
    const int plane = HD * HH * HWp;
    float* __restrict__ xs = lds;
    float* __restrict__ ws = lds + p.cic * plane;
    const int tid = threadIdx.x;
    for (int cc = 0; cc < Cg; cc += p.cic) {          // cc: channel of g (an OUTPUT channel of the forward conv) inside the group
        const int ncc = min(p.cic, Cg - cc);
        __syncthreads();
        vx_jlc_stage<K>(gsrc, xs, p, b, g * Cg + cc, ncc, d0, h0, w0, HD, HH, HWp);
        vx_jlc_stage_w(ws, ncc * K3 * COT, p.nthr, [&](int e) {
            const int j = e % COT, t = (e / COT) % K3, cil = e / (COT * K3);
            // forward weight w[co = g*Cg + cc + cil][ci_in_group = ci0 + j - g*Cg][K3 - 1 - t]
            return w[((long)(g * Cg + cc + cil) * Cg + (ci0 + j - g * Cg)) * K3 + (K3 - 1 - t)];
        });
        __syncthreads();
        if (active) {
            for (int cil = threadIdx.x / p.nsp; cil < ncc; cil += p.parts) {
#pragma unroll
                for (int kd = 0; kd < K; ++kd) {
#pragma unroll
                    for (int kh = 0; kh < K; ++kh) {
                        const float* __restrict__ xrow = xs + ((cil * HD + td + kd) * HH + th + kh) * HWp + 4 * tq;
                        const float4 xa = *reinterpret_cast<const float4*>(xrow);
                        const float4 xb = *reinterpret_cast<const float4*>(xrow + 4);
                        const float xr[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                        const float* __restrict__ wp = ws + (cil * K3 + (kd * K + kh) * K) * COT;
#pragma unroll
                        for (int kw = 0; kw < K; ++kw) {
                            {
                                float wv[COT];
                                vx_jlc_ldw<COT>(wp + kw * COT, wv);
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int j = 0; j < COT; ++j) acc[u][j] = fmaf(wv[j], xr[u + kw], acc[u][j]);
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int COT>
__global__ void __launch_bounds__(512) vx_jlc_conv_bwd_k(VxJlc p) {
    extern __shared__ __attribute__((aligned(16))) float vx_jlc_lds[];
    const int Cg = p.C / p.G;
    const int ci0 = blockIdx.y * COT, g = ci0 / Cg, b = blockIdx.z;        // ci0: first INPUT channel (of the forward conv) this block produces dx for
    int d0, h0, w0, td, th, tq;
    bool active;
    vx_jlc_tile(p, d0, h0, w0, td, th, tq, active);
    float acc[4][COT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[u][j] = 0.0f;
    vx_jlc_adj<5, COT>(p.g5, p.w5, p, vx_jlc_lds, b, g, ci0, d0, h0, w0, td, th, tq, active, acc);
    vx_jlc_adj<3, COT>(p.g3, p.w3, p, vx_jlc_lds, b, g, ci0, d0, h0, w0, td, th, tq, active, acc);
    if (p.parts > 1) {
        const int part = threadIdx.x / p.nsp;
        float* __restrict__ rb = vx_jlc_lds + (long)(threadIdx.x % p.nsp) * (4 * COT);
        for (int pk = 1; pk < p.parts; ++pk) {
            __syncthreads();
            if (part == pk) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < COT; ++j) rb[u * COT + j] = acc[u][j];
            }
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < COT; ++j) acc[u][j] += rb[u * COT + j];
            }
        }
        if (part != 0) return;
    }
    const int od = d0 + td, oh = h0 + th, ow = w0 + 4 * tq;
    if (!active || od >= p.D || oh >= p.H) return;
    // 1x1x1 branch + residual straight from global memory
    const long chan = (long)p.D * p.H * p.W;
    const long sp_off = ((long)od * p.H + oh) * p.W + ow;
    const bool vec = (p.W & 3) == 0 && ow + 3 < p.W;
    for (int cc = 0; cc < Cg; ++cc) {
        const float* src = p.g1 + ((long)b * p.C + g * Cg + cc) * chan + sp_off;
        float gv[4];
        if (vec) { const float4 t = *reinterpret_cast<const float4*>(src); gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w; }
        else {
#pragma unroll
            for (int u = 0; u < 4; ++u) gv[u] = ow + u < p.W ? src[u] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            const float wv = p.w1[(long)(g * Cg + cc) * Cg + (ci0 + j - g * Cg)];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u][j] = fmaf(wv, gv[u], acc[u][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const long off = ((long)b * p.C + ci0 + j) * chan + sp_off;
        if (vec) {
            const float4 r = *reinterpret_cast<const float4*>(p.res + off);
            *reinterpret_cast<float4*>(p.dx + off) = make_float4(acc[0][j] + r.x, acc[1][j] + r.y, acc[2][j] + r.z, acc[3][j] + r.w);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ow + u < p.W) p.dx[off + u] = acc[u][j] + p.res[off + u];
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------- element-wise stages
// TI = element type of the block-INTERNAL tensors (y_k, o, dn, d_o, g_k): float, or vx_bf16 in the bf16 storage mode (they are `const void*` here, typed by the
// kernel's template argument); the block's input x and the incoming gradient dout are fp32 in both modes
struct VxJlcMid {
    const float *x, *dout;
    const void *y1, *y3, *y5, *o, *dn, *d_o;
    const double *part_y;        // [3][BC][nty][2]
    const float *part_dn;        // [BC][npd][2]   (sum dn, sum dn*nhat)   from vx_mlp_bwd
    const float *part_t_in;      // [3][BC][nch][2] (sum t, sum t*yhat)
    const float *stats_y, *stats_o;    // [3][BC][2], [BC][2]  (mean, rstd)
    void *out_o, *out_do, *g1, *g3, *g5;
    float *stats_y_out, *part_t;
    double* part_o;              // [BC][nch][2]
    long BC, V, chunk;
    int nty, npd, nch;
    float eps;
};

// fold n (sum, sumsq) pairs of one row with one wave; result in every lane
__device__ __forceinline__ void vx_fold_d(const double* __restrict__ pp, int n, double& s, double& q) {
    const int lane = threadIdx.x & 63;
    s = 0.0; q = 0.0;
    for (int i = lane; i < n; i += 64) { s += pp[2 * i]; q += pp[2 * i + 1]; }
    s = vx_wave_sum(s);
    q = vx_wave_sum(q);
}
__device__ __forceinline__ void vx_fold_f(const float* __restrict__ pp, int n, float& s, float& q) {
    const int lane = threadIdx.x & 63;
    s = 0.0f; q = 0.0f;
    for (int i = lane; i < n; i += 64) { s += pp[2 * i]; q += pp[2 * i + 1]; }
    s = vx_wave_sum(s);
    q = vx_wave_sum(q);
}

// o = x + sum_k GELU((y_k - mean_k) * rstd_k); partial (sum, sumsq) of o per chunk.  grid (nch, BC)
template <typename TI>
__global__ void __launch_bounds__(256) vx_jlc_mid_fwd_k(VxJlcMid p) {
    const TI* __restrict__ y1 = (const TI*)p.y1; const TI* __restrict__ y3 = (const TI*)p.y3; const TI* __restrict__ y5 = (const TI*)p.y5;
    TI* __restrict__ out_o = (TI*)p.out_o;
    __shared__ float st[6];
    __shared__ double redd[8];
    const long bc = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 192) {                                  // waves 0..2 fold the statistics of y_1, y_3, y_5
        const int k = tid >> 6;
        double s, q;
        vx_fold_d(p.part_y + ((long)k * p.BC + bc) * p.nty * 2, p.nty, s, q);
        if ((tid & 63) == 0) {
            const double m = s / (double)p.V;
            double var = q / (double)p.V - m * m;
            var = var < 0.0 ? 0.0 : var;
            const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            st[2 * k] = mean; st[2 * k + 1] = rstd;
            if (blockIdx.x == 0) { p.stats_y_out[((long)k * p.BC + bc) * 2] = mean; p.stats_y_out[((long)k * p.BC + bc) * 2 + 1] = rstd; }
        }
    }
    __syncthreads();
    const float m1 = st[0], r1 = st[1], m3 = st[2], r3 = st[3], m5 = st[4], r5 = st[5];
    const long base = bc * p.V, c0 = (long)blockIdx.x * p.chunk;
    const long c1 = c0 + p.chunk < p.V ? c0 + p.chunk : p.V;
    float s = 0.0f, q = 0.0f;
    for (long i = c0 + 4 * tid; i < c1; i += 1024) {
        const float4 xv = vx_ld4(p.x, base + i), a = vx_ld4(y1, base + i), b3 = vx_ld4(y3, base + i), c5 = vx_ld4(y5, base + i);
        float4 o;
        o.x = xv.x + vx_gelu_fast((a.x - m1) * r1) + vx_gelu_fast((b3.x - m3) * r3) + vx_gelu_fast((c5.x - m5) * r5);
        o.y = xv.y + vx_gelu_fast((a.y - m1) * r1) + vx_gelu_fast((b3.y - m3) * r3) + vx_gelu_fast((c5.y - m5) * r5);
        o.z = xv.z + vx_gelu_fast((a.z - m1) * r1) + vx_gelu_fast((b3.z - m3) * r3) + vx_gelu_fast((c5.z - m5) * r5);
        o.w = xv.w + vx_gelu_fast((a.w - m1) * r1) + vx_gelu_fast((b3.w - m3) * r3) + vx_gelu_fast((c5.w - m5) * r5);
        // (16-bit o: the statistics of the following norm are those of the values its readers load)
        o.x = vx_round_as<TI>(o.x); o.y = vx_round_as<TI>(o.y); o.z = vx_round_as<TI>(o.z); o.w = vx_round_as<TI>(o.w);
        vx_st4(out_o, base + i, o);
        s += (o.x + o.y) + (o.z + o.w);
        q = fmaf(o.x, o.x, q); q = fmaf(o.y, o.y, q); q = fmaf(o.z, o.z, q); q = fmaf(o.w, o.w, q);
    }
    s = vx_wave_sum(s);
    q = vx_wave_sum(q);
    if ((tid & 63) == 0) { redd[2 * (tid >> 6)] = (double)s; redd[2 * (tid >> 6) + 1] = (double)q; }
    __syncthreads();
    if (tid == 0) {
        double* dst = p.part_o + (bc * p.nch + blockIdx.x) * 2;
        dst[0] = (redd[0] + redd[2]) + (redd[4] + redd[6]);
        dst[1] = (redd[1] + redd[3]) + (redd[5] + redd[7]);
    }
}

// d_o = dout + rstd_o * (dn - mean(dn) - nhat * mean(dn * nhat)); partial sums of t_k = d_o * GELU'(yhat_k) and t_k * yhat_k.  grid (nch, BC)
template <typename TI>
__global__ void __launch_bounds__(256) vx_jlc_mid_bwd_k(VxJlcMid p) {
    const TI* __restrict__ y1 = (const TI*)p.y1; const TI* __restrict__ y3 = (const TI*)p.y3; const TI* __restrict__ y5 = (const TI*)p.y5;
    const TI* __restrict__ o_ = (const TI*)p.o; const TI* __restrict__ dn_ = (const TI*)p.dn;
    TI* __restrict__ out_do = (TI*)p.out_do;
    __shared__ float mm[2];
    __shared__ float redf[4][6];
    const long bc = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 64) {
        float s, q;
        vx_fold_f(p.part_dn + bc * p.npd * 2, p.npd, s, q);
        if (tid == 0) { mm[0] = s / (float)p.V; mm[1] = q / (float)p.V; }
    }
    __syncthreads();
    const float dm1 = mm[0], dm2 = mm[1];
    const float mo = p.stats_o[2 * bc], ro = p.stats_o[2 * bc + 1];
    float mk[3], rk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { mk[k] = p.stats_y[((long)k * p.BC + bc) * 2]; rk[k] = p.stats_y[((long)k * p.BC + bc) * 2 + 1]; }
    const long base = bc * p.V, c0 = (long)blockIdx.x * p.chunk;
    const long c1 = c0 + p.chunk < p.V ? c0 + p.chunk : p.V;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = c0 + 4 * tid; i < c1; i += 1024) {
        const float4 go = vx_ld4(p.dout, base + i), dn = vx_ld4(dn_, base + i), ov = vx_ld4(o_, base + i);
        const float4 a = vx_ld4(y1, base + i), b3 = vx_ld4(y3, base + i), c5 = vx_ld4(y5, base + i);
        const float gov[4] = {go.x, go.y, go.z, go.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w}, oo[4] = {ov.x, ov.y, ov.z, ov.w};
        const float yv[3][4] = {{a.x, a.y, a.z, a.w}, {b3.x, b3.y, b3.z, b3.w}, {c5.x, c5.y, c5.z, c5.w}};
        float dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float nh = (oo[u] - mo) * ro;
            dv[u] = vx_round_as<TI>(gov[u] + ro * (dnv[u] - dm1 - nh * dm2));      // (16-bit d_o: vx_jlc_gk recomputes t_k from the stored value)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float yh = (yv[k][u] - mk[k]) * rk[k];
                const float t = dv[u] * vx_gelu_grad_fast(yh);
                acc[2 * k] += t;
                acc[2 * k + 1] = fmaf(t, yh, acc[2 * k + 1]);
            }
        }
        vx_st4(out_do, base + i, make_float4(dv[0], dv[1], dv[2], dv[3]));
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const float t = vx_wave_sum(acc[e]);
        if ((tid & 63) == 0) redf[tid >> 6][e] = t;
    }
    __syncthreads();
    if (tid < 6) {
        const int k = tid >> 1;
        p.part_t[(((long)k * p.BC + bc) * p.nch + blockIdx.x) * 2 + (tid & 1)] = (redf[0][tid] + redf[1][tid]) + (redf[2][tid] + redf[3][tid]);
    }
}

// g_k = rstd_k * (t_k - mean(t_k) - yhat_k * mean(t_k * yhat_k)), t_k recomputed from d_o and y_k.  grid (nch, BC)
template <typename TI>
__global__ void __launch_bounds__(256) vx_jlc_gk_k(VxJlcMid p) {
    const TI* __restrict__ y1 = (const TI*)p.y1; const TI* __restrict__ y3 = (const TI*)p.y3; const TI* __restrict__ y5 = (const TI*)p.y5;
    const TI* __restrict__ d_o = (const TI*)p.d_o;
    TI* __restrict__ g1 = (TI*)p.g1; TI* __restrict__ g3 = (TI*)p.g3; TI* __restrict__ g5 = (TI*)p.g5;
    __shared__ float mm[6];
    const long bc = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 192) {
        const int k = tid >> 6;
        float s, q;
        vx_fold_f(p.part_t_in + ((long)k * p.BC + bc) * p.nch * 2, p.nch, s, q);
        if ((tid & 63) == 0) { mm[2 * k] = s / (float)p.V; mm[2 * k + 1] = q / (float)p.V; }
    }
    __syncthreads();
    float mk[3], rk[3], t1[3], t2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        mk[k] = p.stats_y[((long)k * p.BC + bc) * 2]; rk[k] = p.stats_y[((long)k * p.BC + bc) * 2 + 1];
        t1[k] = mm[2 * k]; t2[k] = mm[2 * k + 1];
    }
    const long base = bc * p.V, c0 = (long)blockIdx.x * p.chunk;
    const long c1 = c0 + p.chunk < p.V ? c0 + p.chunk : p.V;
    for (long i = c0 + 4 * tid; i < c1; i += 1024) {
        const float4 dv = vx_ld4(d_o, base + i);
        const float4 a = vx_ld4(y1, base + i), b3 = vx_ld4(y3, base + i), c5 = vx_ld4(y5, base + i);
        const float d4[4] = {dv.x, dv.y, dv.z, dv.w};
        const float yv[3][4] = {{a.x, a.y, a.z, a.w}, {b3.x, b3.y, b3.z, b3.w}, {c5.x, c5.y, c5.z, c5.w}};
        float gk[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float yh = (yv[k][u] - mk[k]) * rk[k];
                const float t = d4[u] * vx_gelu_grad_fast(yh);
                gk[k][u] = rk[k] * (t - t1[k] - yh * t2[k]);
            }
        vx_st4(g1, base + i, make_float4(gk[0][0], gk[0][1], gk[0][2], gk[0][3]));
        vx_st4(g3, base + i, make_float4(gk[1][0], gk[1][1], gk[1][2], gk[1][3]));
        vx_st4(g5, base + i, make_float4(gk[2][0], gk[2][1], gk[2][2], gk[2][3]));
    }
}

// --------------------------------------------------------------------------------------------------------------------- host
static int vx_jlc_geom(VxJlc& p, int B, int C, int G, int D, int H, int W, int K, int& COT, size_t& shm) {
    VX_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0 && D > 0 && H > 0 && W > 0, "vx_jlc: bad shape");
    const int Cg = C / G;
    VX_REQUIRE(Cg % 4 == 0, "vx_jlc: group width must be a multiple of 4 (got %d)", Cg);
    p.B = B; p.C = C; p.G = G; p.D = D; p.H = H; p.W = W;
    p.TWq = vx_cdiv(W, 4) < 8 ? vx_cdiv(W, 4) : 8;
    p.TH = H < 8 ? H : 8;
    int td = 256 / (p.TWq * p.TH);
    if (td > D) td = D;
    if (td > 8) td = 8;
    if (td < 1) td = 1;
    p.TD = td;
    // the tiling must not depend on the batch size: the order in which a sample's partial sums are folded would change with B, and a sample's
    // result must be bit-identical whatever it is batched with (tests/test_hip_model_gpu.py); 4 = the nominal batch of the training step
    auto nblk = [&]() { return (long)vx_cdiv(D, p.TD) * vx_cdiv(H, p.TH) * vx_cdiv(W, p.TWq * 4) * (C / COT) * 4; };
    // output channels per thread: 4 (8 would need 256 VGPRs forward: one wave per SIMD).  2 or 1 on the coarse grids are FASTER ALONE (8^3: 25 vs 30 us,
    // 16^3: 48 vs 55 us -- more blocks) and SLOWER IN THE STEP (autopet128 704 vs 729 patches/s, autopet96 988 vs 1037): every input value is then re-read
    // from LDS for fewer FMAs, and in the step these kernels share the CUs with the other lanes' kernels -- issue slots, not latency, are what they cost.
    COT = 4;
    while (p.TD > 1 && nblk() < 128 && (p.TD / 2) * p.TWq * p.TH >= 64) p.TD /= 2;      // very small volumes only: thin tiles re-stage most of their halo (16^3 measured: 121 us thin vs full tiles)
    p.nTd = vx_cdiv(D, p.TD); p.nTh = vx_cdiv(H, p.TH); p.nTw = vx_cdiv(W, p.TWq * 4);
    p.nsp = (p.TD * p.TH * p.TWq + 63) / 64 * 64;
    if (p.nsp > 256) p.nsp = 256;
    const int plane = (p.TD + K - 1) * (p.TH + K - 1) * (p.TWq * 4 + 4);
    const int taps = K == 5 ? 153 : 125;             // forward stages all three kernels' taps; the adjoint the largest kernel's
    int cic = Cg < 4 ? Cg : 4;
    auto lds = [&](int c) { return (size_t)c * (plane + taps * COT) * sizeof(float); };
    while (cic > 1 && lds(cic) > 64 * 1024) cic >>= 1;
    // few blocks or few spatial slots: the staged input channels are split over 2 or 4 thread groups per spatial slot (more waves per SIMD, fewer FMAs per
    // thread), their partial sums folded through LDS
    p.parts = 1;
    if (nblk() < 512) {
        if (cic % 4 == 0 && p.nsp * 4 <= 512 && nblk() < 384) p.parts = 4;
        else if (cic % 2 == 0 && p.nsp * 2 <= 512) p.parts = 2;
    }
    p.nthr = p.nsp * p.parts;
    p.cic = cic;
    shm = lds(cic);
    if (p.parts > 1 && shm < (size_t)p.nsp * 12 * COT * sizeof(float)) shm = (size_t)p.nsp * 12 * COT * sizeof(float);
    if (shm < 4 * 3 * 8 * 2 * sizeof(float)) shm = 4 * 3 * 8 * 2 * sizeof(float);
    return 0;
}

extern "C" int vx_jlc_ntiles(int B, int C, int G, int D, int H, int W) {
    VxJlc p = {};
    int COT; size_t shm;
    if (vx_jlc_geom(p, B, C, G, D, H, W, 5, COT, shm) != 0) return -1;
    return p.nTd * p.nTh * p.nTw;
}

extern "C" int vx_jlc_conv_fwd(const float* x, const float* w1, const float* w3, const float* w5, const float* b1, const float* b3, const float* b5,
                               float* y1, float* y3, float* y5, double* part, int B, int C, int G, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && w1 && w3 && w5 && y1 && y3 && y5 && part, "vx_jlc_conv_fwd: null pointer");
    VxJlc p = {};
    int COT; size_t shm;
    if (int rc = vx_jlc_geom(p, B, C, G, D, H, W, 5, COT, shm)) return rc;
    p.x = x; p.w1 = w1; p.w3 = w3; p.w5 = w5; p.b1 = b1; p.b3 = b3; p.b5 = b5; p.y1 = y1; p.y3 = y3; p.y5 = y5; p.part = part;
    dim3 grid(p.nTd * p.nTh * p.nTw, C / COT, B);
    hipStream_t st = (hipStream_t)stream;
    if (COT == 4) vx_jlc_conv_fwd_k<4><<<grid, dim3(p.nthr), shm, st>>>(p);
    else if (COT == 2) vx_jlc_conv_fwd_k<2><<<grid, dim3(p.nthr), shm, st>>>(p);
    else vx_jlc_conv_fwd_k<1><<<grid, dim3(p.nthr), shm, st>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_conv_fwd");
    return 0;
}

extern "C" int vx_jlc_conv_bwd(const float* g1, const float* g3, const float* g5, const float* w1, const float* w3, const float* w5, const float* d_o,
                               float* dx, int B, int C, int G, int D, int H, int W, void* stream) {
    VX_REQUIRE(g1 && g3 && g5 && w1 && w3 && w5 && d_o && dx, "vx_jlc_conv_bwd: null pointer");
    VxJlc p = {};
    int COT; size_t shm;
    if (int rc = vx_jlc_geom(p, B, C, G, D, H, W, 5, COT, shm)) return rc;
    p.g1 = g1; p.g3 = g3; p.g5 = g5; p.w1 = w1; p.w3 = w3; p.w5 = w5; p.res = d_o; p.dx = dx;
    dim3 grid(p.nTd * p.nTh * p.nTw, C / COT, B);
    hipStream_t st = (hipStream_t)stream;
    if (COT == 4) vx_jlc_conv_bwd_k<4><<<grid, dim3(p.nthr), shm, st>>>(p);
    else if (COT == 2) vx_jlc_conv_bwd_k<2><<<grid, dim3(p.nthr), shm, st>>>(p);
    else vx_jlc_conv_bwd_k<1><<<grid, dim3(p.nthr), shm, st>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_conv_bwd");
    return 0;
}

static long vx_jlc_chunk(long BC, long V) {
    (void)BC;                                          // a function of the row length only (batch-size independent folding order)
    return V >= 16384 ? 4096 : 1024;                   // elements of one (b,c) row per block
}
extern "C" int vx_jlc_nchunks(long BC, long V) { return vx_cdiv(V, vx_jlc_chunk(BC, V)); }

// `h16` != 0: the block-internal tensors (y_k, o, dn, d_o, g_k) are vx_bf16 arrays (bf16 storage mode); x and dout are fp32 either way
extern "C" int vx_jlc_mid_fwd_h(const float* x, const void* y1, const void* y3, const void* y5, const double* part_y, int nty, float* stats_y, void* o,
                                double* part_o, long BC, long V, float eps, int h16, void* stream) {
    VX_REQUIRE(x && y1 && y3 && y5 && part_y && stats_y && o && part_o && BC > 0 && V > 0 && V % 4 == 0, "vx_jlc_mid_fwd: bad args");
    VxJlcMid p = {};
    p.x = x; p.y1 = y1; p.y3 = y3; p.y5 = y5; p.part_y = part_y; p.nty = nty; p.stats_y_out = stats_y; p.out_o = o; p.part_o = part_o;
    p.BC = BC; p.V = V; p.eps = eps; p.chunk = vx_jlc_chunk(BC, V); p.nch = vx_cdiv(V, p.chunk);
    if (h16) vx_jlc_mid_fwd_k<vx_bf16><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    else vx_jlc_mid_fwd_k<float><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_mid_fwd");
    return 0;
}
extern "C" int vx_jlc_mid_fwd(const float* x, const float* y1, const float* y3, const float* y5, const double* part_y, int nty, float* stats_y, float* o,
                              double* part_o, long BC, long V, float eps, void* stream) {
    return vx_jlc_mid_fwd_h(x, y1, y3, y5, part_y, nty, stats_y, o, part_o, BC, V, eps, 0, stream);
}

extern "C" int vx_jlc_mid_bwd_h(const float* dout, const void* dn, const float* part_dn, int npd, const void* o, const float* stats_o, const void* y1,
                                const void* y3, const void* y5, const float* stats_y, void* d_o, float* part_t, long BC, long V, int h16, void* stream) {
    VX_REQUIRE(dout && dn && part_dn && o && stats_o && y1 && y3 && y5 && stats_y && d_o && part_t && BC > 0 && V > 0 && V % 4 == 0, "vx_jlc_mid_bwd: bad args");
    VxJlcMid p = {};
    p.dout = dout; p.dn = dn; p.part_dn = part_dn; p.npd = npd; p.o = o; p.stats_o = stats_o; p.y1 = y1; p.y3 = y3; p.y5 = y5; p.stats_y = stats_y;
    p.out_do = d_o; p.part_t = part_t; p.BC = BC; p.V = V; p.chunk = vx_jlc_chunk(BC, V); p.nch = vx_cdiv(V, p.chunk);
    if (h16) vx_jlc_mid_bwd_k<vx_bf16><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    else vx_jlc_mid_bwd_k<float><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_mid_bwd");
    return 0;
}
extern "C" int vx_jlc_mid_bwd(const float* dout, const float* dn, const float* part_dn, int npd, const float* o, const float* stats_o, const float* y1,
                              const float* y3, const float* y5, const float* stats_y, float* d_o, float* part_t, long BC, long V, void* stream) {
    return vx_jlc_mid_bwd_h(dout, dn, part_dn, npd, o, stats_o, y1, y3, y5, stats_y, d_o, part_t, BC, V, 0, stream);
}

extern "C" int vx_jlc_gk_h(const void* d_o, const void* y1, const void* y3, const void* y5, const float* stats_y, const float* part_t, void* g1, void* g3,
                           void* g5, long BC, long V, int h16, void* stream) {
    VX_REQUIRE(d_o && y1 && y3 && y5 && stats_y && part_t && g1 && g3 && g5 && BC > 0 && V > 0 && V % 4 == 0, "vx_jlc_gk: bad args");
    VxJlcMid p = {};
    p.d_o = d_o; p.y1 = y1; p.y3 = y3; p.y5 = y5; p.stats_y = stats_y; p.part_t_in = part_t; p.g1 = g1; p.g3 = g3; p.g5 = g5;
    p.BC = BC; p.V = V; p.chunk = vx_jlc_chunk(BC, V); p.nch = vx_cdiv(V, p.chunk);
    if (h16) vx_jlc_gk_k<vx_bf16><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    else vx_jlc_gk_k<float><<<dim3(p.nch, (unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_gk");
    return 0;
}
extern "C" int vx_jlc_gk(const float* d_o, const float* y1, const float* y3, const float* y5, const float* stats_y, const float* part_t, float* g1, float* g3,
                         float* g5, long BC, long V, void* stream) {
    return vx_jlc_gk_h(d_o, y1, y3, y5, stats_y, part_t, g1, g3, g5, BC, V, 0, stream);
}
