// Normalisation + element-wise kernels for gfx950 (fp32, NCDHW).
//   InstanceNorm3d (affine=False, eps 1e-5, biased var; reference common_function.py:63-66) is split into
//   a per-(b,c) statistics kernel and "apply" kernels that fuse the activation, the JLC 3-way sum and
//   the residual (conv_blocks.py:72-75, 18-21, 36-39; Encoder.py:351-360; Decoder.py:85-88,160-164).
//   channels-first LayerNorm (attention_utils.py:29-43, eps 1e-6) is thread-local: one thread = one voxel.
#include "vx_common.h"
#include <type_traits>
#include "../../include/veloxseg_hip.h"

// ---------------------------------------------------------------------------------------------
// InstanceNorm statistics: one 256-thread block per (b,c) row of V contiguous floats.
// stats[2*bc] = mean, stats[2*bc+1] = rstd.  fp64 accumulation -> deterministic and cancellation-safe.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vx_in_stats_part_k(const float* __restrict__ x, double* __restrict__ part, long V, int S,
                                                          float* __restrict__ stats, float eps) {
    const long bc = blockIdx.x;
    const int sp = blockIdx.y;
    const float* __restrict__ row = x + bc * V;
    double s = 0.0, ss = 0.0;
    for (long v = (long)sp * 256 + threadIdx.x; v < V; v += (long)S * 256) {
        const double t = (double)row[v];
        s += t;
        ss += t * t;
    }
    s = vx_wave_sum(s);
    ss = vx_wave_sum(ss);
    __shared__ double sm[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[wid] = s; sm[4 + wid] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S1 = sm[0] + sm[1] + sm[2] + sm[3], S2 = sm[4] + sm[5] + sm[6] + sm[7];
        if (S == 1) {          // whole row in this block: finalize here, no second launch
            const double m = S1 / (double)V;
            double var = S2 / (double)V - m * m;
            if (var < 0.0) var = 0.0;
            stats[2 * bc] = (float)m;
            stats[2 * bc + 1] = (float)(1.0 / sqrt(var + (double)eps));
        } else {
            part[(bc * S + sp) * 2] = S1;
            part[(bc * S + sp) * 2 + 1] = S2;
        }
    }
}
__global__ void __launch_bounds__(256) vx_in_stats_fin_k(const double* __restrict__ part, float* __restrict__ stats, long BC, long V, int S, float eps) {
    const long bc = (long)blockIdx.x * 256 + threadIdx.x;
    if (bc >= BC) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < S; ++k) { s += part[(bc * S + k) * 2]; ss += part[(bc * S + k) * 2 + 1]; }
    const double m = s / (double)V;
    double var = ss / (double)V - m * m;
    if (var < 0.0) var = 0.0;
    stats[2 * bc] = (float)m;
    stats[2 * bc + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// out = (res ? res : 0) + sum_{k<nk} act((y_k - mean_k) * rstd_k)
__global__ void __launch_bounds__(256) vx_in_apply_fwd_k(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2,
                                                         const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2,
                                                         int nk, int act, const float* __restrict__ res, float* __restrict__ out, long V) {
    const long bc = blockIdx.y;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const long i = bc * V + v;
    float acc = res ? res[i] : 0.0f;
    {
        float z = (y0[i] - s0[2 * bc]) * s0[2 * bc + 1];
        acc += act ? vx_gelu(z) : z;
    }
    if (nk > 1) {
        float z = (y1[i] - s1[2 * bc]) * s1[2 * bc + 1];
        acc += act ? vx_gelu(z) : z;
    }
    if (nk > 2) {
        float z = (y2[i] - s2[2 * bc]) * s2[2 * bc + 1];
        acc += act ? vx_gelu(z) : z;
    }
    out[i] = acc;
}

// backward statistics for one branch: m[2bc] = mean(dz), m[2bc+1] = mean(dz*z), dz = dout*act'(z)
__global__ void __launch_bounds__(256) vx_in_bwd_stats_part_k(const float* __restrict__ dout, const float* __restrict__ y,
                                                              const float* __restrict__ st, int act, double* __restrict__ part, long V, int S,
                                                              float* __restrict__ m_out) {
    const long bc = blockIdx.x;
    const int sp = blockIdx.y;
    const float mean = st[2 * bc], rstd = st[2 * bc + 1];
    double a = 0.0, c = 0.0;
    for (long v = (long)sp * 256 + threadIdx.x; v < V; v += (long)S * 256) {
        const long i = bc * V + v;
        const float z = (y[i] - mean) * rstd;
        const float dz = act ? dout[i] * vx_gelu_grad(z) : dout[i];
        a += (double)dz;
        c += (double)dz * (double)z;
    }
    a = vx_wave_sum(a);
    c = vx_wave_sum(c);
    __shared__ double sm[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[wid] = a; sm[4 + wid] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double S1 = sm[0] + sm[1] + sm[2] + sm[3], S2 = sm[4] + sm[5] + sm[6] + sm[7];
        if (S == 1) {
            m_out[2 * bc] = (float)(S1 / (double)V);
            m_out[2 * bc + 1] = (float)(S2 / (double)V);
        } else {
            part[(bc * S + sp) * 2] = S1;
            part[(bc * S + sp) * 2 + 1] = S2;
        }
    }
}
__global__ void __launch_bounds__(256) vx_in_bwd_stats_fin_k(const double* __restrict__ part, float* __restrict__ m, long BC, long V, int S) {
    const long bc = (long)blockIdx.x * 256 + threadIdx.x;
    if (bc >= BC) return;
    double a = 0.0, c = 0.0;
    for (int k = 0; k < S; ++k) { a += part[(bc * S + k) * 2]; c += part[(bc * S + k) * 2 + 1]; }
    m[2 * bc] = (float)(a / (double)V);
    m[2 * bc + 1] = (float)(c / (double)V);
}

// dy = rstd * (dz - m1 - z*m2)
// db (optional): bias gradient of the conv that produced y, db[c] += sum_{b,v} dy -- mathematically zero behind an InstanceNorm, numerically the
// same round-off residue a separate reduction of dy yields; fused here it costs one block reduction instead of a launch per conv
__global__ void __launch_bounds__(256) vx_in_bwd_apply_k(const float* __restrict__ dout, const float* __restrict__ y,
                                                         const float* __restrict__ st, const float* __restrict__ m, int act,
                                                         float* __restrict__ dy, long V, float* __restrict__ db, int C) {
    __shared__ float sm[4];
    const long bc = blockIdx.y;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    float o = 0.0f;
    if (v < V) {
        const long i = bc * V + v;
        const float mean = st[2 * bc], rstd = st[2 * bc + 1];
        const float z = (y[i] - mean) * rstd;
        const float dz = act ? dout[i] * vx_gelu_grad(z) : dout[i];
        o = rstd * (dz - m[2 * bc] - z * m[2 * bc + 1]);
        dy[i] = o;
    }
    if (db != nullptr) {
        const float s_ = vx_block_sum_256(o, sm);
        if (threadIdx.x == 0) atomicAdd(db + (int)(bc % C), s_);
    }
}

// ---------------------------------------------------------------------------------------------
// channels-first LayerNorm, one thread per voxel (coalesced along V for every channel).
// ---------------------------------------------------------------------------------------------
// per-voxel mean / rstd over C with a pivot-shifted single sweep (numerically equivalent to the two-pass form at fp32)
__device__ __forceinline__ void vx_ln_stats(const float* __restrict__ xb, int C, long V, float eps, float& u, float& r) {
    const float pivot = xb[0];
    float s = 0.0f, q = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) { const float d = xb[(long)c * V] - pivot; s += d; q = fmaf(d, d, q); }
    const float md = s / (float)C;
    float var = q / (float)C - md * md;
    var = var < 0.0f ? 0.0f : var;
    u = pivot + md;
    r = 1.0f / sqrtf(var + eps);
}

__global__ void __launch_bounds__(256) vx_ln_cf_fwd_k(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ out, int C, long V, float eps) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* __restrict__ xb = x + (long)blockIdx.y * C * V + v;
    float* __restrict__ ob = out + (long)blockIdx.y * C * V + v;
    float u, r;
    vx_ln_stats(xb, C, V, eps, u, r);
#pragma unroll 8
    for (int c = 0; c < C; ++c) ob[(long)c * V] = fmaf(gamma[c], (xb[(long)c * V] - u) * r, beta[c]);
}

// dx = r*(g - mean_c(g) - xhat*mean_c(g*xhat)), g = dout*gamma; per-voxel (mean, rstd) are saved to ws for the parameter pass
__global__ void __launch_bounds__(256) vx_ln_cf_bwd_k(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dout,
                                                      float* __restrict__ dx, float* __restrict__ ws, int C, long V, float eps, const float* __restrict__ add) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const long base = (long)blockIdx.y * C * V + v;
    const float* __restrict__ xb = x + base;
    const float* __restrict__ db = dout + base;
    float u, r;
    vx_ln_stats(xb, C, V, eps, u, r);
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float xh = (xb[(long)c * V] - u) * r;
        const float g = db[(long)c * V] * gamma[c];
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    s1 /= (float)C;
    s2 /= (float)C;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
        const float xh = (xb[(long)c * V] - u) * r;
        const float o = r * (db[(long)c * V] * gamma[c] - s1 - xh * s2);
        dx[base + (long)c * V] = add ? add[base + (long)c * V] + o : o;
    }
    ws[2 * ((long)blockIdx.y * V + v)] = u;
    ws[2 * ((long)blockIdx.y * V + v) + 1] = r;
}

// dgamma[c] += sum_{b,v} dout*xhat ; dbeta[c] += sum dout.   grid (C, chunks)
__global__ void __launch_bounds__(256) vx_ln_cf_bwd_param_k(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ ws,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, long V) {
    const int c = blockIdx.x;
    float dg = 0.0f, db = 0.0f;
    const long n = (long)B * V;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n; i += (long)gridDim.y * 256) {
        const long b = i / V, v = i % V;
        const long idx = (b * C + c) * V + v;
        const float d = dout[idx];
        dg = fmaf(d, (x[idx] - ws[2 * i]) * ws[2 * i + 1], dg);
        db += d;
    }
    __shared__ float sm[4];
    dg = vx_block_sum_256(dg, sm);
    db = vx_block_sum_256(db, sm);
    if (threadIdx.x == 0) { atomicAdd(dgamma + c, dg); atomicAdd(dbeta + c, db); }
}

// ---------------------------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------------------------
// h = drop(gelu(a))
__global__ void __launch_bounds__(256) vx_gelu_drop_fwd_k(const float* __restrict__ a, float* __restrict__ h, long n, VxDrop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) h[i] = vx_gelu(a[i]) * vx_drop(d, (uint64_t)i);
}
// da = dh * mask * gelu'(a)
__global__ void __launch_bounds__(256) vx_gelu_drop_bwd_k(const float* __restrict__ dh, const float* __restrict__ a, float* __restrict__ da, long n, VxDrop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) da[i] = dh[i] * vx_drop(d, (uint64_t)i) * vx_gelu_grad(a[i]);
}
// out = alpha*x + drop(z)   (x may be null -> out = drop(z))
__global__ void __launch_bounds__(256) vx_axpy_drop_fwd_k(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ out,
                                                          float alpha, long n, VxDrop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (x ? alpha * x[i] : 0.0f) + z[i] * vx_drop(d, (uint64_t)i);
}
// dz = mask*dout ; dx = alpha*dout (if dx != null)
__global__ void __launch_bounds__(256) vx_axpy_drop_bwd_k(const float* __restrict__ dout, float* __restrict__ dx, float* __restrict__ dz,
                                                          float alpha, long n, VxDrop d) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = dout[i];
    if (dz) dz[i] = g * vx_drop(d, (uint64_t)i);
    if (dx) dx[i] = alpha * g;
}
// out = a + b (+ c)
__global__ void __launch_bounds__(256) vx_add_k(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i] + (c ? c[i] : 0.0f);
}
// db[c] += sum_{b,v} dy[b,c,v] : one block per channel
__global__ void __launch_bounds__(256) vx_channel_sum_k(const float* __restrict__ dy, float* __restrict__ db, int B, int C, long V) {
    const int c = blockIdx.x;
    float s = 0.0f;
    const long n = (long)B * V;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n; i += (long)gridDim.y * 256) {
        const long b = i / V, v = i % V;
        s += dy[(b * C + c) * V + v];
    }
    __shared__ float sm[4];
    s = vx_block_sum_256(s, sm);
    if (threadIdx.x == 0) atomicAdd(db + c, s);
}

// space-to-depth(2): PatchMerging.faeture_sample (attention_utils.py:144-159).  out[b, sub*C + c, d,h,w] = x[b,c,2d+i,2h+j,2w+k], sub = i*4+j*2+k
__global__ void __launch_bounds__(256) vx_s2d2_k(const float* __restrict__ x, float* __restrict__ out, int C, int D, int H, int W, int inverse) {
    // thread per (coarse) output element; D,H,W are the COARSE dims
    const long Vc = (long)D * H * W;
    const long n = (long)8 * C * Vc;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = blockIdx.y;
    const long v = i % Vc;
    const int ch = (int)(i / Vc);
    const int sub = ch / C, c = ch % C;
    const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
    const int fd = 2 * d + (sub >> 2), fh = 2 * h + ((sub >> 1) & 1), fw = 2 * w + (sub & 1);
    const long fi = (((long)b * C + c) * (2 * D) + fd) * (long)(2 * H) * (2 * W) + (long)fh * (2 * W) + fw;
    const long ci = (long)b * 8 * C * Vc + i;
    if (inverse) out[fi] = x[ci]; else out[ci] = x[fi];
}

// ---------------------------------------------------------------------------------------------
// Long rows in TWO launches per direction (was 2 nk + 1 forward, 3 nk backward): the partial sums of all nk inputs come from one launch
// (grid.z = input) and the consumer kernels fold the S partials of their row themselves (S <= 16 doubles per quantity) instead of waiting
// for a finalisation launch; the forward's first block of every row also stores (mean, rstd) for the backward.
// ---------------------------------------------------------------------------------------------
struct VxIn3 { const float* y[3]; float* st[3]; float* dy[3]; };

__global__ void __launch_bounds__(256) vx_in_stats_part3_k(VxIn3 P, double* __restrict__ part, long BC, long V, int S) {
    const long bc = blockIdx.x;
    const int sp = blockIdx.y, k = blockIdx.z;
    const float* __restrict__ row = P.y[k] + bc * V;
    double s = 0.0, ss = 0.0;
    for (long v = (long)sp * 256 + threadIdx.x; v < V; v += (long)S * 256) {
        const double t = (double)row[v];
        s += t;
        ss += t * t;
    }
    s = vx_wave_sum(s);
    ss = vx_wave_sum(ss);
    __shared__ double sm[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[wid] = s; sm[4 + wid] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* __restrict__ pp = part + (((long)k * BC + bc) * S + sp) * 2;
        pp[0] = sm[0] + sm[1] + sm[2] + sm[3];
        pp[1] = sm[4] + sm[5] + sm[6] + sm[7];
    }
}

__global__ void __launch_bounds__(256) vx_in_apply_fwd_fin_k(VxIn3 P, const double* __restrict__ part, int S, float eps, int nk, int act,
                                                             const float* __restrict__ res, float* __restrict__ out, long BC, long V) {
    __shared__ float ms[6];
    const long bc = blockIdx.y;
    if ((int)threadIdx.x < nk) {
        const int k = threadIdx.x;
        const double* __restrict__ pp = part + (((long)k * BC + bc) * S) * 2;
        double s = 0.0, ss = 0.0;
        for (int j = 0; j < S; ++j) { s += pp[2 * j]; ss += pp[2 * j + 1]; }
        const double m = s / (double)V;
        double var = ss / (double)V - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
        ms[2 * k] = mean; ms[2 * k + 1] = rstd;
        if (blockIdx.x == 0) { P.st[k][2 * bc] = mean; P.st[k][2 * bc + 1] = rstd; }
    }
    __syncthreads();
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const long i = bc * V + v;
    float acc = res ? res[i] : 0.0f;
    for (int k = 0; k < nk; ++k) {
        const float z = (P.y[k][i] - ms[2 * k]) * ms[2 * k + 1];
        acc += act ? vx_gelu(z) : z;
    }
    out[i] = acc;
}

__global__ void __launch_bounds__(256) vx_in_bwd_stats_part3_k(const float* __restrict__ dout, VxIn3 P, int act, double* __restrict__ part, long BC, long V, int S) {
    const long bc = blockIdx.x;
    const int sp = blockIdx.y, k = blockIdx.z;
    if (P.dy[k] == nullptr) return;
    const float mean = P.st[k][2 * bc], rstd = P.st[k][2 * bc + 1];
    const float* __restrict__ y = P.y[k];
    double a = 0.0, c = 0.0;
    for (long v = (long)sp * 256 + threadIdx.x; v < V; v += (long)S * 256) {
        const long i = bc * V + v;
        const float z = (y[i] - mean) * rstd;
        const float dz = act ? dout[i] * vx_gelu_grad(z) : dout[i];
        a += (double)dz;
        c += (double)dz * (double)z;
    }
    a = vx_wave_sum(a);
    c = vx_wave_sum(c);
    __shared__ double sm[8];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[wid] = a; sm[4 + wid] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* __restrict__ pp = part + (((long)k * BC + bc) * S + sp) * 2;
        pp[0] = sm[0] + sm[1] + sm[2] + sm[3];
        pp[1] = sm[4] + sm[5] + sm[6] + sm[7];
    }
}

__global__ void __launch_bounds__(256) vx_in_bwd_apply_fin_k(const float* __restrict__ dout, VxIn3 P, const double* __restrict__ part, int S, int act, long BC, long V,
                                                             const float* __restrict__ add0) {
    __shared__ float mm[2];
    const long bc = blockIdx.y;
    const int k = blockIdx.z;
    if (P.dy[k] == nullptr) return;
    if (threadIdx.x == 0) {
        const double* __restrict__ pp = part + (((long)k * BC + bc) * S) * 2;
        double a = 0.0, c = 0.0;
        for (int j = 0; j < S; ++j) { a += pp[2 * j]; c += pp[2 * j + 1]; }
        mm[0] = (float)(a / (double)V); mm[1] = (float)(c / (double)V);
    }
    __syncthreads();
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const long i = bc * V + v;
    const float mean = P.st[k][2 * bc], rstd = P.st[k][2 * bc + 1];
    const float z = (P.y[k][i] - mean) * rstd;
    const float dz = act ? dout[i] * vx_gelu_grad(z) : dout[i];
    const float o = rstd * (dz - mm[0] - z * mm[1]);
    P.dy[k][i] = (k == 0 && add0) ? add0[i] + o : o;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

static int vx_in_split(long BC, long V) {      // row split so that ~1024 blocks are in flight, each with >= 2048 elements
    long S = 1024 / (BC > 0 ? BC : 1);
    if (S > V / 4096) S = V / 4096;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    return (int)S;
}

extern "C" int vx_in_stats(const float* x, float* stats, double* part_ws, long BC, long V, float eps, void* stream) {
    VX_REQUIRE(x && stats && part_ws && BC > 0 && V > 1, "vx_in_stats: bad args (InstanceNorm needs more than 1 spatial element; BC=%ld V=%ld)", BC, V);
    const int S = vx_in_split(BC, V);
    hipLaunchKernelGGL(vx_in_stats_part_k, dim3((unsigned)BC, S), dim3(256), 0, (hipStream_t)stream, x, part_ws, V, S, stats, eps);
    if (S > 1) hipLaunchKernelGGL(vx_in_stats_fin_k, dim3(vx_cdiv(BC, 256)), dim3(256), 0, (hipStream_t)stream, part_ws, stats, BC, V, S, eps);
    VX_LAUNCH_CHECK("vx_in_stats");
    return 0;
}

extern "C" int vx_in_apply_fwd(const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                               int nk, int act, const float* res, float* out, long BC, long V, void* stream) {
    VX_REQUIRE(nk >= 1 && nk <= 3 && y0 && s0 && out && (nk < 2 || (y1 && s1)) && (nk < 3 || (y2 && s2)), "vx_in_apply_fwd: bad args");
    hipLaunchKernelGGL(vx_in_apply_fwd_k, dim3(vx_cdiv(V, 256), (unsigned)BC), dim3(256), 0, (hipStream_t)stream, y0, y1, y2, s0, s1, s2, nk, act, res, out, V);
    VX_LAUNCH_CHECK("vx_in_apply_fwd");
    return 0;
}

static int vx_in_bwd_run(const float* dout, const float* y, const float* st, int act, float* m_ws, double* part_ws, float* dy, long BC, long V, float* db, int C,
                         void* stream) {
    VX_REQUIRE(dout && y && st && m_ws && part_ws && dy, "vx_in_bwd: null pointer");
    VX_REQUIRE(db == nullptr || (C > 0 && BC % C == 0), "vx_in_bwd_db: BC must be a multiple of C");
    const int S = vx_in_split(BC, V);
    hipLaunchKernelGGL(vx_in_bwd_stats_part_k, dim3((unsigned)BC, S), dim3(256), 0, (hipStream_t)stream, dout, y, st, act, part_ws, V, S, m_ws);
    if (S > 1) hipLaunchKernelGGL(vx_in_bwd_stats_fin_k, dim3(vx_cdiv(BC, 256)), dim3(256), 0, (hipStream_t)stream, part_ws, m_ws, BC, V, S);
    hipLaunchKernelGGL(vx_in_bwd_apply_k, dim3(vx_cdiv(V, 256), (unsigned)BC), dim3(256), 0, (hipStream_t)stream, dout, y, st, m_ws, act, dy, V, db, C);
    VX_LAUNCH_CHECK("vx_in_bwd");
    return 0;
}
extern "C" int vx_in_bwd(const float* dout, const float* y, const float* st, int act, float* m_ws, double* part_ws, float* dy, long BC, long V, void* stream) {
    return vx_in_bwd_run(dout, y, st, act, m_ws, part_ws, dy, BC, V, nullptr, 1, stream);
}
extern "C" int vx_in_bwd_db(const float* dout, const float* y, const float* st, int act, float* m_ws, double* part_ws, float* dy, long BC, long V, float* db, int C,
                            void* stream) {
    return vx_in_bwd_run(dout, y, st, act, m_ws, part_ws, dy, BC, V, db, C, stream);
}

// Long-row InstanceNorm-sum in two launches.  part_ws: nk * BC * 16 * 2 doubles.  st_k receive (mean, rstd) per row for the backward.
extern "C" int vx_in_fwd_split(const float* y0, const float* y1, const float* y2, float* s0, float* s1, float* s2, double* part_ws,
                               int nk, int act, const float* res, float* out, long BC, long V, float eps, void* stream) {
    VX_REQUIRE(nk >= 1 && nk <= 3 && y0 && s0 && out && part_ws && (nk < 2 || (y1 && s1)) && (nk < 3 || (y2 && s2)) && BC > 0 && V > 1, "vx_in_fwd_split: bad args");
    const int S = vx_in_split(BC, V);
    VxIn3 P; P.y[0] = y0; P.y[1] = y1; P.y[2] = y2; P.st[0] = s0; P.st[1] = s1; P.st[2] = s2; P.dy[0] = P.dy[1] = P.dy[2] = nullptr;
    hipLaunchKernelGGL(vx_in_stats_part3_k, dim3((unsigned)BC, S, nk), dim3(256), 0, (hipStream_t)stream, P, part_ws, BC, V, S);
    hipLaunchKernelGGL(vx_in_apply_fwd_fin_k, dim3(vx_cdiv(V, 256), (unsigned)BC), dim3(256), 0, (hipStream_t)stream, P, (const double*)part_ws, S, eps, nk, act, res, out, BC, V);
    VX_LAUNCH_CHECK("vx_in_fwd_split");
    return 0;
}
// backward of the same: dy_k may be NULL (no gradient needed for that input)
extern "C" int vx_in_bwd_split(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                               double* part_ws, int nk, int act, float* dy0, float* dy1, float* dy2, const float* add0, long BC, long V, void* stream) {
    VX_REQUIRE(add0 == nullptr || (dy0 && add0 != dy0), "vx_in_bwd_split: add0 needs a distinct dy0");
    VX_REQUIRE(nk >= 1 && nk <= 3 && dout && y0 && s0 && part_ws && (nk < 2 || (y1 && s1)) && (nk < 3 || (y2 && s2)) && BC > 0 && V > 1, "vx_in_bwd_split: bad args");
    if (!dy0 && !dy1 && !dy2) return 0;
    const int S = vx_in_split(BC, V);
    VxIn3 P; P.y[0] = y0; P.y[1] = y1; P.y[2] = y2; P.st[0] = const_cast<float*>(s0); P.st[1] = const_cast<float*>(s1); P.st[2] = const_cast<float*>(s2);
    P.dy[0] = dy0; P.dy[1] = nk > 1 ? dy1 : nullptr; P.dy[2] = nk > 2 ? dy2 : nullptr;
    hipLaunchKernelGGL(vx_in_bwd_stats_part3_k, dim3((unsigned)BC, S, nk), dim3(256), 0, (hipStream_t)stream, dout, P, act, part_ws, BC, V, S);
    hipLaunchKernelGGL(vx_in_bwd_apply_fin_k, dim3(vx_cdiv(V, 256), (unsigned)BC, nk), dim3(256), 0, (hipStream_t)stream, dout, P, (const double*)part_ws, S, act, BC, V, add0);
    VX_LAUNCH_CHECK("vx_in_bwd_split");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm on FEW voxels with MANY channels (PatchMerging LN over 8C at the 8^3 / 4^3 levels: C = 256..1024, B*V = 256..2048): one thread per
// voxel walks the channel axis serially with ~B*V threads on the whole chip (81 us for 131 k elements).  Here 16 lanes share one voxel
// (4 voxels per wave, 16 per block), each lane takes every 16th channel, sums meet in a 16-lane shuffle tree.  Same pivot-shifted
// single-sweep statistics as vx_ln_stats.
// ---------------------------------------------------------------------------------------------------------------------------
template <int LPV>
__device__ __forceinline__ float vx_sum_lpv(float v) {
#pragma unroll
    for (int o = LPV / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#define vx_sum16 vx_sum_lpv<LPV>

// LPV lanes per voxel: 16 (16 voxels per block), or 64 where even that leaves the chip empty (B*V <= 1024 voxels of >= 256 channels: the PatchMerging
// LayerNorm in front of level 4 ran 16 blocks for 38 us)
template <bool BWD, int LPV = 16>
__global__ void __launch_bounds__(256) vx_ln_cf_lanes_k(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ dout, float* __restrict__ out, float* __restrict__ ws, int C, long V,
                                                        long BV, float eps) {
    constexpr int VPB = 256 / LPV;                                       // voxels per block
    const long vox = (long)blockIdx.x * VPB + (threadIdx.x / LPV);       // over (b, v)
    const int sub = threadIdx.x & (LPV - 1);
    const bool live = vox < BV;
    const long vv = live ? vox : 0;
    const long b = vv / V, v = vv % V;
    const float* __restrict__ xb = x + b * C * V + v;
    const float pivot = xb[0];
    float s_ = 0.0f, q_ = 0.0f;
    for (int c = sub; c < C; c += LPV) { const float d = xb[(long)c * V] - pivot; s_ += d; q_ = fmaf(d, d, q_); }
    s_ = vx_sum16(s_);
    q_ = vx_sum16(q_);
    const float md = s_ / (float)C;
    float var = q_ / (float)C - md * md;
    var = var < 0.0f ? 0.0f : var;
    const float u = pivot + md, r = 1.0f / sqrtf(var + eps);
    if (!BWD) {
        float* __restrict__ ob = out + b * C * V + v;
        if (live)
            for (int c = sub; c < C; c += LPV) ob[(long)c * V] = fmaf(gamma[c], (xb[(long)c * V] - u) * r, beta[c]);
    } else {
        const float* __restrict__ db = dout + b * C * V + v;
        float s1 = 0.0f, s2 = 0.0f;
        for (int c = sub; c < C; c += LPV) {
            const float xh = (xb[(long)c * V] - u) * r;
            const float g = db[(long)c * V] * gamma[c];
            s1 += g;
            s2 = fmaf(g, xh, s2);
        }
        s1 = vx_sum16(s1) / (float)C;
        s2 = vx_sum16(s2) / (float)C;
        if (live) {
            float* __restrict__ ob = out + b * C * V + v;
            const float* __restrict__ ab = beta ? beta + b * C * V + v : nullptr;      // backward: `beta` carries the optional gradient to add (same layout as out)
            for (int c = sub; c < C; c += LPV) {
                const float xh = (xb[(long)c * V] - u) * r;
                const float o = r * (db[(long)c * V] * gamma[c] - s1 - xh * s2);
                ob[(long)c * V] = ab ? ab[(long)c * V] + o : o;
            }
            if (sub == 0) { ws[2 * vv] = u; ws[2 * vv + 1] = r; }
        }
    }
}
#undef vx_sum16
static inline bool vx_ln_use_lanes(int B, int C, long V) { return (long)B * V <= 16384 && C >= 64; }
static inline bool vx_ln_wide_lanes(int B, int C, long V) { return (long)B * V <= 1024 && C >= 256; }

extern "C" int vx_ln_cf_fwd(const float* x, const float* gamma, const float* beta, float* out, int B, int C, long V, float eps, void* stream) {
    VX_REQUIRE(x && gamma && beta && out && B > 0 && C > 0 && V > 0, "vx_ln_cf_fwd: bad args");
    if (vx_ln_use_lanes(B, C, V)) {
        if (vx_ln_wide_lanes(B, C, V)) vx_ln_cf_lanes_k<false, 64><<<dim3(vx_cdiv((long)B * V, 4)), dim3(256), 0, (hipStream_t)stream>>>(x, gamma, beta, nullptr, out, nullptr, C, V, (long)B * V, eps);
        else
        vx_ln_cf_lanes_k<false><<<dim3(vx_cdiv((long)B * V, 16)), dim3(256), 0, (hipStream_t)stream>>>(x, gamma, beta, nullptr, out, nullptr, C, V, (long)B * V, eps);
        VX_LAUNCH_CHECK("vx_ln_cf_fwd");
        return 0;
    }
    hipLaunchKernelGGL(vx_ln_cf_fwd_k, dim3(vx_cdiv(V, 256), B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, out, C, V, eps);
    VX_LAUNCH_CHECK("vx_ln_cf_fwd");
    return 0;
}

// parts: 1 = input gradient (+ the per-voxel statistics in ws), 2 = parameter gradients from ws, 3 = both
static int vx_ln_cf_bwd_run(const float* x, const float* gamma, const float* dout, const float* add, float* dx, float* dgamma, float* dbeta, float* ws,
                            int B, int C, long V, float eps, void* stream, int parts = 3) {
    VX_REQUIRE(x && dout && ws && B > 0 && C > 0 && V > 0, "vx_ln_cf_bwd: bad pointers");
    if (parts & 1) {
        VX_REQUIRE(gamma && dx && add != dx, "vx_ln_cf_bwd: bad pointers");
        if (vx_ln_use_lanes(B, C, V) && vx_ln_wide_lanes(B, C, V))
            vx_ln_cf_lanes_k<true, 64><<<dim3(vx_cdiv((long)B * V, 4)), dim3(256), 0, (hipStream_t)stream>>>(x, gamma, add, dout, dx, ws, C, V, (long)B * V, eps);
        else if (vx_ln_use_lanes(B, C, V))
            vx_ln_cf_lanes_k<true><<<dim3(vx_cdiv((long)B * V, 16)), dim3(256), 0, (hipStream_t)stream>>>(x, gamma, add, dout, dx, ws, C, V, (long)B * V, eps);
        else
            hipLaunchKernelGGL(vx_ln_cf_bwd_k, dim3(vx_cdiv(V, 256), B), dim3(256), 0, (hipStream_t)stream, x, gamma, dout, dx, ws, C, V, eps, add);
    }
    if (parts & 2) {
        VX_REQUIRE(dgamma && dbeta, "vx_ln_cf_bwd: bad pointers");
        int chunks = vx_cdiv((long)B * V, 256 * 8);
        if (chunks > 64) chunks = 64;
        hipLaunchKernelGGL(vx_ln_cf_bwd_param_k, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, x, dout, ws, dgamma, dbeta, B, C, V);
    }
    VX_LAUNCH_CHECK("vx_ln_cf_bwd");
    return 0;
}
// The two halves of vx_ln_cf_bwd / vx_ln_cf_bwd_add as entries of their own: only the input gradient is on the backward's dependent chain, the
// parameter gradients (from x, dout and the per-voxel statistics the first half left in ws) are a sink that the caller may launch later.
extern "C" int vx_ln_cf_bwd_data(const float* x, const float* gamma, const float* dout, const float* add, float* dx, float* ws, int B, int C, long V, float eps,
                                 void* stream) {
    return vx_ln_cf_bwd_run(x, gamma, dout, add, dx, nullptr, nullptr, ws, B, C, V, eps, stream, 1);
}
extern "C" int vx_ln_cf_bwd_param(const float* x, const float* dout, const float* ws, float* dgamma, float* dbeta, int B, int C, long V, void* stream) {
    return vx_ln_cf_bwd_run(x, nullptr, dout, nullptr, nullptr, dgamma, dbeta, const_cast<float*>(ws), B, C, V, 0.0f, stream, 2);
}
extern "C" int vx_ln_cf_bwd(const float* x, const float* gamma, const float* dout, float* dx, float* dgamma, float* dbeta, float* ws,
                            int B, int C, long V, float eps, void* stream) {
    return vx_ln_cf_bwd_run(x, gamma, dout, nullptr, dx, dgamma, dbeta, ws, B, C, V, eps, stream);
}
// dx = add + LayerNorm backward: the sum with the residual-branch gradient of the same tensor in the kernel's store
extern "C" int vx_ln_cf_bwd_add(const float* x, const float* gamma, const float* dout, const float* add, float* dx, float* dgamma, float* dbeta, float* ws,
                                int B, int C, long V, float eps, void* stream) {
    VX_REQUIRE(add, "vx_ln_cf_bwd_add: add is null");
    return vx_ln_cf_bwd_run(x, gamma, dout, add, dx, dgamma, dbeta, ws, B, C, V, eps, stream);
}

extern "C" int vx_gelu_drop_fwd(const float* a, float* h, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(a && h && n > 0, "vx_gelu_drop_fwd: bad args");
    hipLaunchKernelGGL(vx_gelu_drop_fwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, h, n, vx_mk_drop(seed_ptr, dstream, p));
    VX_LAUNCH_CHECK("vx_gelu_drop_fwd");
    return 0;
}
extern "C" int vx_gelu_drop_bwd(const float* dh, const float* a, float* da, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(dh && a && da && n > 0, "vx_gelu_drop_bwd: bad args");
    hipLaunchKernelGGL(vx_gelu_drop_bwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dh, a, da, n, vx_mk_drop(seed_ptr, dstream, p));
    VX_LAUNCH_CHECK("vx_gelu_drop_bwd");
    return 0;
}
extern "C" int vx_axpy_drop_fwd(const float* x, const float* z, float* out, float alpha, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(z && out && n > 0, "vx_axpy_drop_fwd: bad args");
    hipLaunchKernelGGL(vx_axpy_drop_fwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, z, out, alpha, n, vx_mk_drop(seed_ptr, dstream, p));
    VX_LAUNCH_CHECK("vx_axpy_drop_fwd");
    return 0;
}
extern "C" int vx_axpy_drop_bwd(const float* dout, float* dx, float* dz, float alpha, long n, const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(dout && n > 0, "vx_axpy_drop_bwd: bad args");
    hipLaunchKernelGGL(vx_axpy_drop_bwd_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, dz, alpha, n, vx_mk_drop(seed_ptr, dstream, p));
    VX_LAUNCH_CHECK("vx_axpy_drop_bwd");
    return 0;
}
extern "C" int vx_add(const float* a, const float* b, const float* c, float* out, long n, void* stream) {
    VX_REQUIRE(a && b && out && n > 0, "vx_add: bad args");
    hipLaunchKernelGGL(vx_add_k, dim3(vx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, n);
    VX_LAUNCH_CHECK("vx_add");
    return 0;
}
// out[k] = a[k] + b[k] (+ c[k]) for up to 16 tensors in ONE launch: the per-branch gradients of the tensors the encoder hands to the M + 1 decoders
// (12 of them, two adds each as separate aten launches before) are summed at the start of the encoder backward.  blockIdx.y = tensor.
struct VxAddMany { const float* a[16]; const float* b[16]; const float* c[16]; float* out[16]; long n[16]; };
__global__ void __launch_bounds__(256) vx_add_many_k(VxAddMany P) {
    const int k = blockIdx.y;
    const long n = P.n[k];
    const float* __restrict__ a = P.a[k]; const float* __restrict__ b = P.b[k]; const float* __restrict__ c = P.c[k];
    float* __restrict__ o = P.out[k];
    const bool v4 = (n & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)o) & 15) == 0);
    if (v4) {
        const long n4 = n >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 x = reinterpret_cast<const float4*>(a)[i];
            const float4 y = reinterpret_cast<const float4*>(b)[i];
            x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
            if (c) { const float4 z = reinterpret_cast<const float4*>(c)[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
            reinterpret_cast<float4*>(o)[i] = x;
        }
    } else
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) o[i] = a[i] + b[i] + (c ? c[i] : 0.0f);
}
extern "C" int vx_add_many(const float* const* a, const float* const* b, const float* const* c, float* const* out, const long* n, int count, void* stream) {
    VX_REQUIRE(a && b && out && n && count >= 1 && count <= 16, "vx_add_many: 1..16 tensors");
    VxAddMany P;
    long nmax = 0;
    for (int k = 0; k < 16; ++k) {
        const bool in = k < count;
        P.a[k] = in ? a[k] : nullptr; P.b[k] = in ? b[k] : nullptr; P.c[k] = (in && c) ? c[k] : nullptr; P.out[k] = in ? out[k] : nullptr; P.n[k] = in ? n[k] : 0;
        if (in) { VX_REQUIRE(a[k] && b[k] && out[k] && n[k] > 0, "vx_add_many: null tensor %d", k); if (n[k] > nmax) nmax = n[k]; }
    }
    int gx = vx_cdiv(nmax, 1024);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(vx_add_many_k, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, P);
    VX_LAUNCH_CHECK("vx_add_many");
    return 0;
}
extern "C" int vx_channel_sum(const float* dy, float* db, int B, int C, long V, void* stream) {
    VX_REQUIRE(dy && db && B > 0 && C > 0 && V > 0, "vx_channel_sum: bad args");
    int chunks = vx_cdiv((long)B * V, 256 * 16);
    if (chunks > 32) chunks = 32;
    hipLaunchKernelGGL(vx_channel_sum_k, dim3(C, chunks), dim3(256), 0, (hipStream_t)stream, dy, db, B, C, V);
    VX_LAUNCH_CHECK("vx_channel_sum");
    return 0;
}
extern "C" int vx_space_to_depth2(const float* x, float* out, int B, int C, int Dc, int Hc, int Wc, int inverse, void* stream) {
    VX_REQUIRE(x && out && B > 0 && C > 0 && Dc > 0 && Hc > 0 && Wc > 0, "vx_space_to_depth2: bad args");
    const long n = (long)8 * C * Dc * Hc * Wc;
    hipLaunchKernelGGL(vx_s2d2_k, dim3(vx_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, x, out, C, Dc, Hc, Wc, inverse);
    VX_LAUNCH_CHECK("vx_space_to_depth2");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// patchify: out[b, c*K^3 + (kd*K + kh)*K + kw, z, y, x] = in[b, c, z*K + kd, y*K + kh, x*K + kw]
// A conv with kernel == stride and no padding (PatchEmbed, Encoder.py:150-156) is then a 1x1 conv over C*K^3 channels whose weight
// is the conv weight viewed as (Cout, C*K^3): forward and weight gradient run on the pointwise MFMA kernels instead of the generic
// direct-conv ones (weight gradient of the 128^3 PatchEmbed: 220 us -> s2d once in the forward + 15 us).
// thread = one (input row, x): reads K contiguous floats, writes one float to each of K channel planes (coalesced across lanes).
// ---------------------------------------------------------------------------------------------------------------------------
// (TO: element type of the patchified copy -- float, or vx_bf16 in the bf16 storage mode, round 6: a thread then takes TWO consecutive x, so that a plane gets 4-byte stores)
template <int K, typename TO = float>
__global__ void __launch_bounds__(256) vx_patchify_k(const float* __restrict__ in, TO* __restrict__ out, int C, int d, int h, int w, long bstride) {
    constexpr int XP = std::is_same<TO, float>::value ? 1 : 2;      // x positions per thread (the host checked w % XP == 0)
    const long rows = (long)C * d * K * h * K;                  // input rows of one sample, each w*K floats long
    const int wq = w / XP;
    const long n = rows * wq;
    const long b = blockIdx.y;
    const long Vo = (long)d * h * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % wq) * XP;
        long r = i / wq;
        const int kh = (int)(r % K); r /= K;
        const int y = (int)(r % h); r /= h;
        const int kd = (int)(r % K); r /= K;
        const int z = (int)(r % d);
        const int c = (int)(r / d);
        const float* __restrict__ src = in + b * bstride + (((long)c * (d * K) + (z * K + kd)) * (long)(h * K) + (y * K + kh)) * (long)(w * K) + (long)x * K;
        float v[XP][K];
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            if (K == 4) { const float4 t = *reinterpret_cast<const float4*>(src + p * K); v[p][0] = t.x; v[p][1] = t.y; v[p][2] = t.z; v[p][K - 1] = t.w; }
            else if (K == 2) { const float2 t = *reinterpret_cast<const float2*>(src + p * K); v[p][0] = t.x; v[p][K - 1] = t.y; }
            else {
#pragma unroll
                for (int kw = 0; kw < K; ++kw) v[p][kw] = src[p * K + kw];
            }
        }
        TO* __restrict__ dst = out + ((b * C + c) * (long)(K * K * K) + (long)(kd * K + kh) * K) * Vo + ((long)z * h + y) * w + x;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            if constexpr (XP == 1) vx_st1(dst, (long)kw * Vo, v[0][kw]);
            else *reinterpret_cast<uint32_t*>(dst + (long)kw * Vo) = vx_pack_bf16x2(v[0][kw], v[1][kw]);
        }
    }
}

// batch_stride (floats) = distance between the samples of x: C * (d K)(h K)(w K) for a contiguous tensor, larger for a channel slice of a wider one
// (the per-modality chunks of the network input, Encoder.py:192 of the reference) -- no .contiguous() copy in front of the patch embedding
extern "C" int vx_patchify_bs_h(const float* x, long batch_stride, void* out_, int B, int C, int d, int h, int w, int K, int out_h16, void* stream);
extern "C" int vx_patchify_bs(const float* x, long batch_stride, float* out, int B, int C, int d, int h, int w, int K, void* stream) {
    return vx_patchify_bs_h(x, batch_stride, out, B, C, d, h, w, K, 0, stream);
}
// out_h16 != 0: the patchified copy is a vx_bf16 array (bf16 storage mode: read by vx_pw_conv_fwd_h / vx_pw_conv_bwd_weight_h)
extern "C" int vx_patchify_bs_h(const float* x, long batch_stride, void* out_, int B, int C, int d, int h, int w, int K, int out_h16, void* stream) {
    float* out = (float*)out_;
    VX_REQUIRE(x && out && B > 0 && C > 0 && d > 0 && h > 0 && w > 0, "vx_patchify: bad args");
    VX_REQUIRE(K == 2 || K == 3 || K == 4, "vx_patchify: patch size 2, 3 or 4 (got %d)", K);
    VX_REQUIRE(batch_stride >= (long)C * d * K * h * K * w * K && (K == 3 || batch_stride % K == 0), "vx_patchify: bad batch stride %ld", batch_stride);
    const long n = (long)C * d * K * h * K * w;
    int g = vx_cdiv(n, 256 * 4);
    if (g > 16384) g = 16384;
    const dim3 grid(g, B), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_h16) {
        VX_REQUIRE((K == 4 || K == 2) && w % 2 == 0, "vx_patchify: 16-bit output for patch size 2 / 4 and an even grid width");
        if (K == 4) vx_patchify_k<4, vx_bf16><<<grid, blk, 0, st>>>(x, (vx_bf16*)out_, C, d, h, w, batch_stride);
        else vx_patchify_k<2, vx_bf16><<<grid, blk, 0, st>>>(x, (vx_bf16*)out_, C, d, h, w, batch_stride);
        VX_LAUNCH_CHECK("vx_patchify");
        return 0;
    }
    if (K == 4) vx_patchify_k<4><<<grid, blk, 0, st>>>(x, out, C, d, h, w, batch_stride);
    else if (K == 3) vx_patchify_k<3><<<grid, blk, 0, st>>>(x, out, C, d, h, w, batch_stride);
    else vx_patchify_k<2><<<grid, blk, 0, st>>>(x, out, C, d, h, w, batch_stride);
    VX_LAUNCH_CHECK("vx_patchify");
    return 0;
}
extern "C" int vx_patchify(const float* x, float* out, int B, int C, int d, int h, int w, int K, void* stream) {
    return vx_patchify_bs(x, (long)C * d * K * h * K * w * K, out, B, C, d, h, w, K, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------
// InstanceNorm on SHORT rows (V <= VX_IN_ROW_MAX): statistics and application in ONE launch.  One 256-thread block owns one (b,c) row of
// every input, keeps it in registers (<= 16 floats per thread and input), reduces in fp64 exactly like vx_in_stats_part_k, then applies.
// Levels 2-4 of the 128^3 pyramid (16^3, 8^3, 4^3 voxels) take this path: 1 launch instead of n+1 forward and of 2n backward.
// ---------------------------------------------------------------------------------------------------------------------------
#define VX_IN_ROW_MAX 4096
__device__ __forceinline__ void vx_block_sum2_f64(double& a, double& c, double* sm) {
    a = vx_wave_sum(a);
    c = vx_wave_sum(c);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { sm[wid] = a; sm[4 + wid] = c; }
    __syncthreads();
    a = sm[0] + sm[1] + sm[2] + sm[3];
    c = sm[4] + sm[5] + sm[6] + sm[7];
}

// NK inputs, R = ceil(V / 256) elements per thread: both compile-time (round 5) -- the input / statistics pointers were picked from arrays with a run-time index
// (generic-address FLAT loads, which also count on the LDS wait counter) and every row walked all 16 element slots of the longest row (4^3 rows: 15 of 16 predicated off)
template <int NK, int R>
__global__ void __launch_bounds__(256) vx_in_row_fwd_k(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ y2,
                                                       float* __restrict__ s0, float* __restrict__ s1, float* __restrict__ s2,
                                                       int act, const float* __restrict__ res, float* __restrict__ out, int V, float eps) {
    __shared__ double sm[8];
    const long bc = blockIdx.x;
    const float* const ys[3] = {y0, y1, y2};
    float* const ss[3] = {s0, s1, s2};
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int v = r * 256 + threadIdx.x; acc[r] = (res && v < V) ? res[bc * V + v] : 0.0f; }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const float* __restrict__ row = ys[k] + bc * V;
        float x[R];
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = r * 256 + threadIdx.x;
            x[r] = v < V ? row[v] : 0.0f;
            const double t = (double)x[r];
            a += t;
            c += t * t;
        }
        vx_block_sum2_f64(a, c, sm);
        const double m = a / (double)V;
        double var = c / (double)V - m * m;
        if (var < 0.0) var = 0.0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (threadIdx.x == 0) { ss[k][2 * bc] = mean; ss[k][2 * bc + 1] = rstd; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float z = (x[r] - mean) * rstd;
            acc[r] += act ? vx_gelu(z) : z;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { const int v = r * 256 + threadIdx.x; if (v < V) out[bc * V + v] = acc[r]; }
}

template <int NK, int R>
__global__ void __launch_bounds__(256) vx_in_row_bwd_k(const float* __restrict__ dout, const float* __restrict__ y0, const float* __restrict__ y1,
                                                       const float* __restrict__ y2, const float* __restrict__ s0, const float* __restrict__ s1,
                                                       const float* __restrict__ s2, int act, float* __restrict__ d0, float* __restrict__ d1,
                                                       float* __restrict__ d2, int V, float* __restrict__ b0, float* __restrict__ b1, float* __restrict__ b2, int C,
                                                       const float* __restrict__ add0) {
    __shared__ double sm[8];
    __shared__ float smf[4];
    const long bc = blockIdx.x;
    const float* const ys[3] = {y0, y1, y2};
    const float* const ss[3] = {s0, s1, s2};
    float* const ds[3] = {d0, d1, d2};
    float* const dbs[3] = {b0, b1, b2};
    float g[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int v = r * 256 + threadIdx.x; g[r] = v < V ? dout[bc * V + v] : 0.0f; }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        if (ds[k] == nullptr) continue;                 // this input needs no gradient (block-uniform)
        const float* __restrict__ row = ys[k] + bc * V;
        const float mean = ss[k][2 * bc], rstd = ss[k][2 * bc + 1];
        float z[R], dz[R];
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = r * 256 + threadIdx.x;
            z[r] = v < V ? (row[v] - mean) * rstd : 0.0f;
            dz[r] = v < V ? (act ? g[r] * vx_gelu_grad(z[r]) : g[r]) : 0.0f;
            a += (double)dz[r];
            c += (double)dz[r] * (double)z[r];
        }
        vx_block_sum2_f64(a, c, sm);
        const float m1 = (float)(a / (double)V), m2 = (float)(c / (double)V);
        float osum = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int v = r * 256 + threadIdx.x;
            if (v < V) {
                const float o = rstd * (dz[r] - m1 - z[r] * m2);
                ds[k][bc * V + v] = (k == 0 && add0) ? add0[bc * V + v] + o : o;        // add0: another gradient of the same tensor (residual branch), summed here
                osum += o;
            }
        }
        if (dbs[k] != nullptr) {              // bias gradient of the producing conv (see vx_in_bwd_apply_k)
            const float s_ = vx_block_sum_256(osum, smf);
            if (threadIdx.x == 0) atomicAdd(dbs[k] + (int)(bc % C), s_);
        }
    }
}

// <NK, R> instance for (nk, V): R = 1, 2, 4, 8 or 16 elements per thread
#define VX_IN_ROW_DISPATCH(KERNEL, nk_, V_, ...)                                                                              \
    do {                                                                                                                      \
        const int r_ = (V_) <= 256 ? 1 : (V_) <= 512 ? 2 : (V_) <= 1024 ? 4 : (V_) <= 2048 ? 8 : 16;                          \
        switch ((nk_) * 100 + r_) {                                                                                           \
            case 101: KERNEL<1, 1> __VA_ARGS__; break; case 102: KERNEL<1, 2> __VA_ARGS__; break; case 104: KERNEL<1, 4> __VA_ARGS__; break;   \
            case 108: KERNEL<1, 8> __VA_ARGS__; break; case 116: KERNEL<1, 16> __VA_ARGS__; break;                             \
            case 201: KERNEL<2, 1> __VA_ARGS__; break; case 202: KERNEL<2, 2> __VA_ARGS__; break; case 204: KERNEL<2, 4> __VA_ARGS__; break;   \
            case 208: KERNEL<2, 8> __VA_ARGS__; break; case 216: KERNEL<2, 16> __VA_ARGS__; break;                             \
            case 301: KERNEL<3, 1> __VA_ARGS__; break; case 302: KERNEL<3, 2> __VA_ARGS__; break; case 304: KERNEL<3, 4> __VA_ARGS__; break;   \
            case 308: KERNEL<3, 8> __VA_ARGS__; break; default: KERNEL<3, 16> __VA_ARGS__; break;                              \
        }                                                                                                                     \
    } while (0)

// d0 = add0 + InstanceNorm backward (single input): folds the add of a residual-branch gradient into the row kernel
extern "C" int vx_in_row_bwd_add(const float* dout, const float* y0, const float* s0, int act, const float* add0, float* d0, long BC, long V, void* stream) {
    VX_REQUIRE(dout && y0 && s0 && add0 && d0 && add0 != d0 && BC > 0 && V > 1 && V <= VX_IN_ROW_MAX, "vx_in_row_bwd_add: bad args");
    VX_IN_ROW_DISPATCH(vx_in_row_bwd_k, 1, V, <<<dim3((unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(dout, y0, nullptr, nullptr, s0, nullptr, nullptr, act, d0, nullptr, nullptr, (int)V,
                                                                                                            nullptr, nullptr, nullptr, 1, add0));
    VX_LAUNCH_CHECK("vx_in_row_bwd_add");
    return 0;
}

extern "C" int vx_in_row_max(void) { return VX_IN_ROW_MAX; }

extern "C" int vx_in_row_fwd(const float* y0, const float* y1, const float* y2, float* s0, float* s1, float* s2, int nk, int act, const float* res,
                             float* out, long BC, long V, float eps, void* stream) {
    VX_REQUIRE(y0 && s0 && out && nk >= 1 && nk <= 3 && BC > 0, "vx_in_row_fwd: bad args");
    VX_REQUIRE(V > 1, "vx_in_row_fwd: InstanceNorm needs more than 1 spatial element per channel (got %ld), as nn.InstanceNorm3d does", V);
    VX_REQUIRE(V <= VX_IN_ROW_MAX, "vx_in_row_fwd: rows of at most %d elements (got %ld): use vx_in_stats + vx_in_apply_fwd", VX_IN_ROW_MAX, V);
    VX_REQUIRE((nk < 2 || (y1 && s1)) && (nk < 3 || (y2 && s2)), "vx_in_row_fwd: missing input");
    VX_IN_ROW_DISPATCH(vx_in_row_fwd_k, nk, V, <<<dim3((unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(y0, y1, y2, s0, s1, s2, act, res, out, (int)V, eps));
    VX_LAUNCH_CHECK("vx_in_row_fwd");
    return 0;
}

extern "C" int vx_in_row_bwd(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                             int nk, int act, float* d0, float* d1, float* d2, long BC, long V, void* stream) {
    VX_REQUIRE(dout && y0 && s0 && nk >= 1 && nk <= 3 && BC > 0 && V > 1 && V <= VX_IN_ROW_MAX, "vx_in_row_bwd: bad args");
    VX_IN_ROW_DISPATCH(vx_in_row_bwd_k, nk, V, <<<dim3((unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(dout, y0, y1, y2, s0, s1, s2, act, d0, d1, d2, (int)V, nullptr, nullptr, nullptr, 1, nullptr));
    VX_LAUNCH_CHECK("vx_in_row_bwd");
    return 0;
}
extern "C" int vx_in_row_bwd_db(const float* dout, const float* y0, const float* y1, const float* y2, const float* s0, const float* s1, const float* s2,
                                int nk, int act, float* d0, float* d1, float* d2, float* db0, float* db1, float* db2, int C, long BC, long V, void* stream) {
    VX_REQUIRE(dout && y0 && s0 && nk >= 1 && nk <= 3 && BC > 0 && V > 1 && V <= VX_IN_ROW_MAX && C > 0 && BC % C == 0, "vx_in_row_bwd_db: bad args");
    VX_IN_ROW_DISPATCH(vx_in_row_bwd_k, nk, V, <<<dim3((unsigned)BC), dim3(256), 0, (hipStream_t)stream>>>(dout, y0, y1, y2, s0, s1, s2, act, d0, d1, d2, (int)V, db0, db1, db2, C, nullptr));
    VX_LAUNCH_CHECK("vx_in_row_bwd_db");
    return 0;
}
