// Generic direct 3-D convolution family for gfx950 (fp32, NCDHW, cubic kernel, uniform stride/pad, groups).
//
// One triple (fwd / bwd_data / bwd_weight) covers every convolution on the VeloxSeg path
// (reference: model/components/conv_blocks.py:4-75, model/Decoder.py:73-76,150-158,
// model/Encoder.py:334-337, model/components/PWA.py:291-298, attention_utils.py:56-57,141):
//   * 1x1 channel mixers (K=1), JLC grouped 1/3/5 convs, patch-expand 3^3 (+PixelShuffle store),
//     DownConv (k=2p-1, stride p), PatchEmbed (k=s=p);
//   * ConvTranspose3d k2 s2 (UpConv) runs as the ADJOINT: its forward is conv3d_bwd_data, its
//     input-gradient is conv3d_fwd, its weight-gradient is conv3d_bwd_weight with roles swapped.
// Weights are indexed only by block/loop-uniform values, so hipcc keeps them on the scalar path
// (s_load + v_fmac with an SGPR operand); activations are read coalesced along W.
// The input may be the channel-concatenation of two tensors (x: first C1 channels, x2: the rest),
// which removes every torch.cat on the path; the output may be stored pixel-shuffled
// (model/components/superpixel.py:16).
#include "vx_common.h"
#include <type_traits>
#include "../../include/veloxseg_hip.h"

struct VxConv {
    int B, Cin, Di, Hi, Wi;
    int Cout, Do, Ho, Wo;
    int K, S, P, G;
    int C1;   // channels coming from x; channels [C1, Cin) come from x2
    int ps;   // pixel-shuffle factor of the y storage (1 = plain)
};

__device__ __forceinline__ long vx_y_index(const VxConv& p, int b, int co, int d, int h, int w) {
    if (p.ps == 1) return ((((long)b * p.Cout + co) * p.Do + d) * p.Ho + h) * (long)p.Wo + w;
    const int s = p.ps;
    const int s3 = co % s;
    int t = co / s;
    const int s2 = t % s;
    t /= s;
    const int s1 = t % s;
    const int c = t / s;
    const int Cc = p.Cout / (s * s * s);
    return ((((long)b * Cc + c) * (p.Do * s) + d * s + s1) * (long)(p.Ho * s) + h * s + s2) * (long)(p.Wo * s) + w * s + s3;
}

__device__ __forceinline__ const float* vx_in_chan(const VxConv& p, const float* __restrict__ x, const float* __restrict__ x2, int b, int c) {
    const long V = (long)p.Di * p.Hi * p.Wi;
    return (c < p.C1) ? x + ((long)b * p.C1 + c) * V : x2 + ((long)b * (p.Cin - p.C1) + (c - p.C1)) * V;
}

// ------------------------------------------------------------------------------------------
// forward: one thread = one output voxel x COT output channels (same group).  The weight slice of the block's
// COT channels is staged in LDS as [ci][tap][COT] (CIC input channels per pass) and read back as broadcast
// ds_read_b128, so strided / odd-sized kernels (DownConv k7s4, k3s2, PatchEmbed k=s) do not depend on scalar-load merging.
// ------------------------------------------------------------------------------------------
template <int KT, int COT>
__global__ void __launch_bounds__(256) vx_conv3d_fwd_k(const float* __restrict__ x, const float* __restrict__ x2,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, VxConv p, int cic) {
    extern __shared__ __attribute__((aligned(16))) float vx_wlds[];
    const int K = KT > 0 ? KT : p.K;
    const int K3 = K * K * K;
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * COT;
    const int Cin_g = p.Cin / p.G, Cout_g = p.Cout / p.G;
    const int g = co0 / Cout_g;
    const bool valid = v < Vo;
    const long vv = valid ? v : 0;
    const int ow = (int)(vv % p.Wo);
    const int oh = (int)((vv / p.Wo) % p.Ho);
    const int od = (int)(vv / ((long)p.Wo * p.Ho));
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = bias ? bias[co0 + j] : 0.0f;
    const int id0 = od * p.S - p.P, ih0 = oh * p.S - p.P, iw0 = ow * p.S - p.P;
    for (int cc = 0; cc < Cin_g; cc += cic) {
        const int ncc = min(cic, Cin_g - cc);
        __syncthreads();
        for (int e = threadIdx.x; e < ncc * K3 * COT; e += 256) {
            const int j = e % COT, t = (e / COT) % K3, cil = e / (COT * K3);
            vx_wlds[e] = w[((long)(co0 + j) * Cin_g + cc + cil) * K3 + t];
        }
        __syncthreads();
        if (valid) {
            for (int cil = 0; cil < ncc; ++cil) {
                const float* __restrict__ xc = vx_in_chan(p, x, x2, b, g * Cin_g + cc + cil);
                const float* __restrict__ wc = vx_wlds + (long)cil * K3 * COT;
                for (int kd = 0; kd < K; ++kd) {
                    const int id = id0 + kd;
                    const bool okd = (unsigned)id < (unsigned)p.Di;
                    for (int kh = 0; kh < K; ++kh) {
                        const int ih = ih0 + kh;
                        const bool okh = okd && (unsigned)ih < (unsigned)p.Hi;
                        const long rowoff = ((long)id * p.Hi + ih) * p.Wi;
#pragma unroll
                        for (int kw = 0; kw < K; ++kw) {
                            const int iw = iw0 + kw;
                            const float xv = (okh && (unsigned)iw < (unsigned)p.Wi) ? xc[rowoff + iw] : 0.0f;
                            const float* wp = wc + ((kd * K + kh) * K + kw) * COT;
#pragma unroll
                            for (int j = 0; j < COT; ++j) acc[j] = fmaf(wp[j], xv, acc[j]);
                        }
                    }
                }
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < COT; ++j) y[vx_y_index(p, b, co0 + j, od, oh, ow)] = acc[j];
}

// ------------------------------------------------------------------------------------------
// backward-data: one thread = one input voxel x CIT input channels (same group); weights staged in LDS as [co][tap][CIT]
//   dx[b,ci,pos] (+)= sum_{co in group} sum_t dy[b,co,q] * w[co,ci,t],   q*S - P + t = pos
// ------------------------------------------------------------------------------------------
template <int KT, int CIT>
__global__ void __launch_bounds__(256) vx_conv3d_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w,
                                                            const float* __restrict__ bias_like,  // optional bias added (ConvTranspose fwd)
                                                            float* __restrict__ dx, float* __restrict__ dx2, VxConv p, int accumulate, int coc) {
    extern __shared__ __attribute__((aligned(16))) float vx_wlds[];
    const int K = KT > 0 ? KT : p.K;
    const int K3 = K * K * K;
    const long Vi = (long)p.Di * p.Hi * p.Wi;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int ci0 = blockIdx.y * CIT;             // absolute input channel
    const int Cin_g = p.Cin / p.G, Cout_g = p.Cout / p.G;
    const int g = ci0 / Cin_g;
    const int cil = ci0 - g * Cin_g;              // channel inside group
    const bool valid = v < Vi;
    const long vv = valid ? v : 0;
    const int iw = (int)(vv % p.Wi);
    const int ih = (int)((vv / p.Wi) % p.Hi);
    const int id = (int)(vv / ((long)p.Wi * p.Hi));
    float acc[CIT];
#pragma unroll
    for (int j = 0; j < CIT; ++j) acc[j] = bias_like ? bias_like[ci0 + j] : 0.0f;
    for (int c0 = 0; c0 < Cout_g; c0 += coc) {
        const int ncoc = min(coc, Cout_g - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < ncoc * K3 * CIT; e += 256) {
            const int j = e % CIT, t = (e / CIT) % K3, col = e / (CIT * K3);
            vx_wlds[e] = w[((long)(g * Cout_g + c0 + col) * Cin_g + cil + j) * K3 + t];
        }
        __syncthreads();
        if (valid && p.ps == 1) {
            // taps of the right residue per axis, resolved ONCE per thread (the integer divisions by the run-time stride used to sit inside the
            // output-channel loop: 0.36 ms for the 64->128 stride-2 DownConv on an 8^3 volume); the channel loop is then loads + FMAs only
            constexpr int KM = KT > 0 ? KT : 7;
            int nd = 0, nh = 0, nw = 0, od[KM], oh[KM], ow[KM], td[KM], th[KM], tw[KM];
            for (int kd = (id + p.P) % p.S; kd < K; kd += p.S) {
                const int qd = (id + p.P - kd) / p.S;
                if (id + p.P - kd >= 0 && qd < p.Do && nd < KM) { od[nd] = qd * p.Ho * p.Wo; td[nd] = kd * K * K; ++nd; }
            }
            for (int kh = (ih + p.P) % p.S; kh < K; kh += p.S) {
                const int qh = (ih + p.P - kh) / p.S;
                if (ih + p.P - kh >= 0 && qh < p.Ho && nh < KM) { oh[nh] = qh * p.Wo; th[nh] = kh * K; ++nh; }
            }
            for (int kw = (iw + p.P) % p.S; kw < K; kw += p.S) {
                const int qw = (iw + p.P - kw) / p.S;
                if (iw + p.P - kw >= 0 && qw < p.Wo && nw < KM) { ow[nw] = qw; tw[nw] = kw; ++nw; }
            }
            const long Vo = (long)p.Do * p.Ho * p.Wo;
            const float* __restrict__ dyb = dy + ((long)b * p.Cout + g * Cout_g + c0) * Vo;
            for (int col = 0; col < ncoc; ++col) {
                const float* __restrict__ wc = vx_wlds + (long)col * K3 * CIT;
                const float* __restrict__ dyc = dyb + (long)col * Vo;
#pragma unroll
                for (int a = 0; a < KM; ++a) {
                    if (a < nd) {
#pragma unroll
                        for (int e = 0; e < KM; ++e) {
                            if (e < nh) {
#pragma unroll
                                for (int f = 0; f < KM; ++f) {
                                    if (f < nw) {
                                        const float dv = dyc[od[a] + oh[e] + ow[f]];
                                        const float* wp = wc + (td[a] + th[e] + tw[f]) * CIT;
#pragma unroll
                                        for (int j = 0; j < CIT; ++j) acc[j] = fmaf(wp[j], dv, acc[j]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
        } else if (valid) {
            for (int col = 0; col < ncoc; ++col) {
                const int co = g * Cout_g + c0 + col;
                const float* __restrict__ wc = vx_wlds + (long)col * K3 * CIT;
                // only taps with (pos + P - k) divisible by the stride contribute: start at the right residue and step by S
                for (int kd = (id + p.P) % p.S; kd < K; kd += p.S) {
                    const int qd = (id + p.P - kd) / p.S;
                    if (id + p.P - kd < 0 || qd >= p.Do) continue;
                    for (int kh = (ih + p.P) % p.S; kh < K; kh += p.S) {
                        const int qh = (ih + p.P - kh) / p.S;
                        if (ih + p.P - kh < 0 || qh >= p.Ho) continue;
                        for (int kw = (iw + p.P) % p.S; kw < K; kw += p.S) {
                            const int qw = (iw + p.P - kw) / p.S;
                            if (iw + p.P - kw < 0 || qw >= p.Wo) continue;
                            const float dv = dy[vx_y_index(p, b, co, qd, qh, qw)];
                            const float* wp = wc + ((kd * K + kh) * K + kw) * CIT;
#pragma unroll
                            for (int j = 0; j < CIT; ++j) acc[j] = fmaf(wp[j], dv, acc[j]);
                        }
                    }
                }
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < CIT; ++j) {
        const int c = ci0 + j;
        float* dst = (c < p.C1) ? dx + ((long)b * p.C1 + c) * Vi + v : dx2 + ((long)b * (p.Cin - p.C1) + (c - p.C1)) * Vi + v;
        if (accumulate) *dst += acc[j]; else *dst = acc[j];
    }
}

// ------------------------------------------------------------------------------------------
// backward-weight: one thread = one weight (co, ci, tap); blockIdx.y = voxel chunk, blockIdx.z = b.
//   dw[co,ci,t] += sum_q dy[b,co,q] * x[b, g*Cin_g+ci, q*S-P+t];  db[co] += sum_q dy[b,co,q]
// Partial sums are added with one float atomic per (weight, chunk).
// ------------------------------------------------------------------------------------------
template <int KT>
__global__ void __launch_bounds__(256) vx_conv3d_bwd_weight_k(const float* __restrict__ x, const float* __restrict__ x2,
                                                              const float* __restrict__ dy, float* __restrict__ dw,
                                                              float* __restrict__ db, VxConv p, int vox_per_chunk) {
    const int K = KT > 0 ? KT : p.K;
    const int K3 = K * K * K;
    const int Cin_g = p.Cin / p.G, Cout_g = p.Cout / p.G;
    const long nW = (long)p.Cout * Cin_g * K3;
    const long widx = (long)blockIdx.x * 256 + threadIdx.x;
    if (widx >= nW) return;
    const int t = (int)(widx % K3);
    const int ci = (int)((widx / K3) % Cin_g);
    const int co = (int)(widx / ((long)K3 * Cin_g));
    const int g = co / Cout_g;
    const int kw = t % K, kh = (t / K) % K, kd = t / (K * K);
    const int b = blockIdx.z;
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    const long q0 = (long)blockIdx.y * vox_per_chunk;
    const long q1 = (q0 + vox_per_chunk < Vo) ? q0 + vox_per_chunk : Vo;
    const float* __restrict__ xc = vx_in_chan(p, x, x2, b, g * Cin_g + ci);
    const bool bias_lane = (db != nullptr) && ci == 0 && t == 0;
    float acc = 0.0f, bacc = 0.0f;
    int qw = (int)(q0 % p.Wo), qh = (int)((q0 / p.Wo) % p.Ho), qd = (int)(q0 / ((long)p.Wo * p.Ho));
    for (long q = q0; q < q1; ++q) {
        const float dv = dy[vx_y_index(p, b, co, qd, qh, qw)];
        const int id = qd * p.S - p.P + kd, ih = qh * p.S - p.P + kh, iw = qw * p.S - p.P + kw;
        const bool ok = (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
        const float xv = ok ? xc[((long)id * p.Hi + ih) * p.Wi + iw] : 0.0f;
        acc = fmaf(dv, xv, acc);
        if (bias_lane) bacc += dv;
        if (++qw == p.Wo) { qw = 0; if (++qh == p.Ho) { qh = 0; ++qd; } }
    }
    atomicAdd(dw + widx, acc);
    if (bias_lane) atomicAdd(db + co, bacc);
}

// ------------------------------------------------------------------------------------------
// host launchers (C ABI)
// ------------------------------------------------------------------------------------------
static int vx_conv_fill(VxConv& p, int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int C1, int ps, const char* who) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || K <= 0 || S <= 0 || P < 0 || G <= 0 || ps <= 0) VX_FAIL(-1, "%s: bad sizes", who);
    if (Cin % G || Cout % G) VX_FAIL(-1, "%s: channels (%d,%d) not divisible by groups %d", who, Cin, Cout, G);
    if (Cout % (ps * ps * ps)) VX_FAIL(-1, "%s: Cout %d not divisible by ps^3", who, Cout);
    p.B = B; p.Cin = Cin; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cout = Cout;
    p.Do = (Di + 2 * P - K) / S + 1; p.Ho = (Hi + 2 * P - K) / S + 1; p.Wo = (Wi + 2 * P - K) / S + 1;
    if (p.Do <= 0 || p.Ho <= 0 || p.Wo <= 0) VX_FAIL(-1, "%s: empty output", who);
    p.K = K; p.S = S; p.P = P; p.G = G; p.C1 = (C1 <= 0 || C1 > Cin) ? Cin : C1; p.ps = ps;
    return 0;
}

static int vx_pick_tile(int per_group) {
    if (per_group % 16 == 0) return 16;
    if (per_group % 8 == 0) return 8;
    if (per_group % 4 == 0) return 4;
    if (per_group % 2 == 0) return 2;
    return 1;
}

template <int N> using vx_ic = std::integral_constant<int, N>;
template <class F> static void vx_dispatch_kt(int K, F&& f) {
    switch (K) {
        case 1: f(vx_ic<1>{}); break;
        case 3: f(vx_ic<3>{}); break;
        case 5: f(vx_ic<5>{}); break;
        default: f(vx_ic<0>{}); break;
    }
}
template <class F> static void vx_dispatch_tile(int T, F&& f) {
    switch (T) {
        case 16: f(vx_ic<16>{}); break;
        case 8: f(vx_ic<8>{}); break;
        case 4: f(vx_ic<4>{}); break;
        case 2: f(vx_ic<2>{}); break;
        default: f(vx_ic<1>{}); break;
    }
}

extern "C" int vx_conv3d_fwd(const float* x, const float* x2, int C1, const float* w, const float* bias, float* y,
                             int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps,
                             void* stream) {
    VxConv p;
    if (int e = vx_conv_fill(p, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, C1, ps, "vx_conv3d_fwd")) return e;
    if (!x || !w || !y || (p.C1 < Cin && !x2)) VX_FAIL(-1, "vx_conv3d_fwd: null pointer");
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    int T = vx_pick_tile(Cout / G);
    while (T > 1 && (long)vx_cdiv(Vo, 256) * (Cout / T) * B < 512) T >>= 1;        // small outputs: more blocks
    const int K3 = K * K * K, Cin_g = Cin / G;
    int cic = Cin_g;
    while (cic > 1 && (size_t)cic * K3 * T * sizeof(float) > 32 * 1024) cic = (cic + 1) / 2;
    const size_t shm = (size_t)cic * K3 * T * sizeof(float);
    VX_REQUIRE(shm <= 150 * 1024, "vx_conv3d_fwd: weight slice does not fit LDS (K=%d)", K);
    dim3 grid(vx_cdiv(Vo, 256), Cout / T, B);
    hipStream_t st = (hipStream_t)stream;
    vx_dispatch_kt(K, [&](auto kt) {
        vx_dispatch_tile(T, [&](auto tt) {
            vx_conv3d_fwd_k<decltype(kt)::value, decltype(tt)::value><<<grid, dim3(256), shm, st>>>(x, x2, w, bias, y, p, cic);
        });
    });
    VX_LAUNCH_CHECK("vx_conv3d_fwd");
    return 0;
}

extern "C" int vx_conv3d_bwd_data(const float* dy, const float* w, const float* bias_like, float* dx, float* dx2, int C1,
                                  int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps,
                                  int accumulate, void* stream) {
    VxConv p;
    if (int e = vx_conv_fill(p, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, C1, ps, "vx_conv3d_bwd_data")) return e;
    if (!dy || !w || !dx || (p.C1 < Cin && !dx2)) VX_FAIL(-1, "vx_conv3d_bwd_data: null pointer");
    const long Vi = (long)Di * Hi * Wi;
    int T = vx_pick_tile(Cin / G);
    if (p.C1 < Cin) { while (T > 1 && (p.C1 % T)) T >>= 1; }   // a tile must not straddle the concat boundary
    while (T > 1 && (long)vx_cdiv(Vi, 256) * (Cin / T) * B < 512) T >>= 1;
    const int K3 = K * K * K, Cout_g = Cout / G;
    int coc = Cout_g;
    while (coc > 1 && (size_t)coc * K3 * T * sizeof(float) > 32 * 1024) coc = (coc + 1) / 2;
    const size_t shm = (size_t)coc * K3 * T * sizeof(float);
    VX_REQUIRE(shm <= 150 * 1024, "vx_conv3d_bwd_data: weight slice does not fit LDS (K=%d)", K);
    dim3 grid(vx_cdiv(Vi, 256), Cin / T, B);
    hipStream_t st = (hipStream_t)stream;
    vx_dispatch_kt(K, [&](auto kt) {
        vx_dispatch_tile(T, [&](auto tt) {
            vx_conv3d_bwd_data_k<decltype(kt)::value, decltype(tt)::value><<<grid, dim3(256), shm, st>>>(dy, w, bias_like, dx, dx2, p, accumulate, coc);
        });
    });
    VX_LAUNCH_CHECK("vx_conv3d_bwd_data");
    return 0;
}

extern "C" int vx_conv3d_bwd_weight(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                                    int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps,
                                    void* stream) {
    VxConv p;
    if (int e = vx_conv_fill(p, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, C1, ps, "vx_conv3d_bwd_weight")) return e;
    if (!x || !dy || !dw || (p.C1 < Cin && !x2)) VX_FAIL(-1, "vx_conv3d_bwd_weight: null pointer");
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    const long nW = (long)Cout * (Cin / G) * K * K * K;
    const int gx = vx_cdiv(nW, 256);
    long chunks = 4096 / ((long)gx * B);
    if (chunks < 1) chunks = 1;
    if (chunks > (Vo + 63) / 64) chunks = (Vo + 63) / 64;
    const int vpc = (int)((Vo + chunks - 1) / chunks);
    dim3 grid(gx, vx_cdiv(Vo, vpc), B);
    hipStream_t st = (hipStream_t)stream;
    vx_dispatch_kt(K, [&](auto kt) {
        vx_conv3d_bwd_weight_k<decltype(kt)::value><<<grid, dim3(256), 0, st>>>(x, x2, dy, dw, db, p, vpc);
    });
    VX_LAUNCH_CHECK("vx_conv3d_bwd_weight");
    return 0;
}
