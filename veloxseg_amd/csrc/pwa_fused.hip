// Fused per-voxel chains of a Paired-Window-Attention transformer block for gfx950, ALL modalities of the block in one launch.
//
//   "pre"  (reference PWA.py:291-298,329-344 with attention_utils.py:29-43):   xn = LN(x);  q = Wq xn + bq;  k = Wk xn + bk;  v = Wv xn + bv
//          also PatchMerging (attention_utils.py:127-168): the 8-way strided gather is the tile load, then LN(8C) and the 8C -> 2C reduction
//   "post" (PWA.py:377,433-439 with attention_utils.py:45-71):   y = a x + Drop(Wm s + bm);  out = y + Drop(W2 Drop(GELU(W1 LN(y) + b1)) + b2)
// Before: LN, q, k, v, mix (+ LN, linear1, linear2 at the 8^3 / 4^3 levels) were 4-8 launches per modality and block forward and 10-16 backward, each at
// the 5-15 us launch floor, on a dependent chain.  Now: one launch per chain and direction for all modalities, plus ONE grouped
// weight-gradient launch (pointwise.hip vx_pw_wgrad_group) that is a sink of the backward pass.
//
// MI355X mapping: a block owns one tile of NT = 16 T voxels of one (modality, sample) and keeps the tile's activations in LDS as act[channel][voxel]
// (row stride 64 floats for T = 4; 20 for T = 1 so that the four k-groups of an MFMA B operand fall on different banks).  A chain stage is
//   out[M x NT] = W[M x K] . act[K x NT]
// on v_mfma_f32_16x16x4_f32: the waves of the block split the 16-row tiles of M; the A operand (weights, <= 128 KB, L2-resident and shared by every
// block) comes straight from global memory as 16-byte loads -- the K axis of an MFMA step is permuted (lane group q supplies k = 16 j + 4 q + i at
// step i) so that one float4 feeds four steps; the B operand is one ds_read_b128 (T = 4: four interleaved n-tiles, voxel 4 r + j) or ds_read_b32.
// The accumulators go through the stage's epilogue (bias, GELU, dropout, residual) back into LDS as the next stage's B operand, or to global memory.
// LayerNorm over the channel axis runs on the LDS tile (two-pass mean / variance as the reference's u = mean, s = mean((x - u)^2)).
// Dropout masks are those of the per-operator kernels (element index of the (B, C, V) tensor of that site; vx_common.h), GELU is vx_cdf_pdf.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

typedef float vx_f32x4 __attribute__((ext_vector_type(4)));
#define VX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {
template <int T> struct Geo { static constexpr int NT = 16 * T; static constexpr int S = (T == 1) ? 20 : 64; };

// out rows [16 mt, 16 mt + 16) for the row tiles of this wave.  W(m, k) = w[m * wsm + k * wsk]; TR = (wsk != 1).  Up to 3 K-segments with their own
// weight pointers (concatenated reduction axis: the three projections of the pre-backward).  `in` = LDS act[K total][S].  epi(mt, acc):
// acc[j][reg] = out[16 mt + 4 q + reg][voxel (T == 4 ? 4 r + j : r)]
template <int T, int NW, bool TR, class Epi>
__device__ __forceinline__ void tile_gemm(int nseg, const float* const (&w)[3], const int (&Kseg)[3], int wsm, int wsk, int Mrows, const float* __restrict__ in, Epi&& epi) {
    constexpr int S = Geo<T>::S;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    for (int mt = wave; mt * 16 < Mrows; mt += NW) {
        vx_f32x4 acc[T];
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
        // chunks of 64 k (4 x 16-byte A loads per lane), software-pipelined: the loads of chunk n + 1 are issued before the MFMAs of chunk n
        int kin = 0;
        float a[4][4], an[4][4];
        // (the weight pointer is picked from w[] with a run-time index: say that it is GLOBAL memory, or the loads become FLAT -- slower, and counted on the LDS wait
        // counter as well, so that every wait for a B operand also waited for the weight prefetch)
        typedef const __attribute__((address_space(1))) float* gcf;
        typedef const __attribute__((address_space(1))) vx_f32x4* gcf4;
        auto load_a = [&](float (&dst)[4][4], const float* __restrict__ wrow_, int K, int k0) {
            gcf wrow = (gcf)(unsigned long long)wrow_;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kb = min(k0 + 16 * c, K - 16) + 4 * q;          // clamped: chunks beyond K reload the last one (never used)
                if (!TR) {
                    const vx_f32x4 t = *(gcf4)(wrow + kb);
                    dst[c][0] = t[0]; dst[c][1] = t[1]; dst[c][2] = t[2]; dst[c][3] = t[3];
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[c][i] = wrow[(long)(kb + i) * wsk];
                }
            }
        };
        for (int sg = 0; sg < nseg; ++sg) {
            const int K = Kseg[sg];
            const float* __restrict__ wrow = w[sg] + (long)(16 * mt + r) * wsm;
            load_a(a, wrow, K, 0);
            for (int k0 = 0; k0 < K; k0 += 64) {
                if (k0 + 64 < K) load_a(an, wrow, K, k0 + 64);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (k0 + 16 * c < K) {                    // wave-uniform
                        const float* __restrict__ brow = in + (long)(kin + k0 + 16 * c + 4 * q) * S;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (T == 4) {
                                const float4 bv = *reinterpret_cast<const float4*>(brow + i * S + 4 * r);
                                acc[0] = VX_MFMA(a[c][i], bv.x, acc[0]);
                                acc[1] = VX_MFMA(a[c][i], bv.y, acc[1]);
                                acc[2] = VX_MFMA(a[c][i], bv.z, acc[2]);
                                acc[3] = VX_MFMA(a[c][i], bv.w, acc[3]);
                            } else {
                                acc[0] = VX_MFMA(a[c][i], brow[i * S + r], acc[0]);
                            }
                        }
                    }
                }
                if (k0 + 64 < K) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int i = 0; i < 4; ++i) a[c][i] = an[c][i];
                }
            }
            kin += K;
        }
        epi(mt, acc);
    }
}
template <int T, int NW, bool TR, class Epi>
__device__ __forceinline__ void tile_gemm1(const float* w, int K, int wsm, int wsk, int Mrows, const float* __restrict__ in, Epi&& epi) {
    const float* const ws[3] = {w, nullptr, nullptr};
    const int ks[3] = {K, 0, 0};
    tile_gemm<T, NW, TR>(1, ws, ks, wsm, wsk, Mrows, in, epi);
}

// rows [0, R) of a (R, V) slab of one sample into LDS act[row][S]; columns beyond V read as 0.  16 threads per row (float4 each for T = 4).
// Eight rows per thread are in flight at once: the loads are unconditional (clamped address, value selected afterwards) -- a loop of the form
// `if (ok) v = src[i]; lds[e] = v;` pays one memory latency per trip (DESIGN.md lesson -1).
template <int T, int NW>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src /* row 0 of the sample */, int R, long V, long v0) {
    constexpr int S = Geo<T>::S, RP = 4 * NW, U = 8;
    const int cg = threadIdx.x & 15;
    for (int row0 = threadIdx.x >> 4; row0 < R; row0 += U * RP) {
        if constexpr (T == 4) {
            const long v = v0 + 4 * cg;
            const bool ok = v < V;
            const long vc = ok ? v : 0;
            float4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int row = min(row0 + u * RP, R - 1); t[u] = *reinterpret_cast<const float4*>(src + (long)row * V + vc); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + u * RP;
                if (row < R) *reinterpret_cast<float4*>(dst + row * S + 4 * cg) = ok ? t[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const long v = v0 + cg;
            const bool ok = v < V;
            const long vc = ok ? v : 0;
            float t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int row = min(row0 + u * RP, R - 1); t[u] = src[(long)row * V + vc]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = row0 + u * RP;
                if (row < R) dst[row * S + cg] = ok ? t[u] : 0.0f;
            }
        }
    }
}

// per-voxel statistics of the LDS tile X[C][S] over the channel axis (two passes); every thread returns those of ITS column col = tid % NT.
// red: NP * NT floats of scratch.  Ends with a barrier-free state: callers must __syncthreads() before overwriting red again.
template <int T, int NW>
__device__ __forceinline__ void tile_ln_stats(const float* __restrict__ X, int C, float eps, float* __restrict__ red, float& mean, float& rstd) {
    constexpr int NT = Geo<T>::NT, S = Geo<T>::S, NP = 64 * NW / NT;
    const int col = threadIdx.x % NT, part = threadIdx.x / NT;
    float s = 0.0f;
    for (int c = part; c < C; c += NP) s += X[c * S + col];
    red[part * NT + col] = s;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll 4
    for (int p_ = 0; p_ < NP; ++p_) tot += red[p_ * NT + col];
    mean = tot / (float)C;
    __syncthreads();
    float qq = 0.0f;
    for (int c = part; c < C; c += NP) { const float d = X[c * S + col] - mean; qq = fmaf(d, d, qq); }
    red[part * NT + col] = qq;
    __syncthreads();
    float var = 0.0f;
#pragma unroll 4
    for (int p_ = 0; p_ < NP; ++p_) var += red[p_ * NT + col];
    rstd = 1.0f / sqrtf(var / (float)C + eps);
}

__device__ __forceinline__ VxDropCtx drop_ctx_dev(const void* seed_ptr, unsigned long long site, float p) {
    VxDrop d;
    d.seed_ptr = p > 0.0f ? (const uint64_t*)seed_ptr : nullptr;
    d.stream = site;
    d.p = p;
    return vx_drop_ctx(d);
}
// dropout masks of the accumulator elements (row, voxel(s)) of one register: T = 4 -> four consecutive voxels, one Philox call when aligned
template <int T>
__device__ __forceinline__ void acc_masks(const VxDropCtx& dc, uint64_t row, long V, long v, float (&m)[4]) {
    if (!dc.on) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
    if constexpr (T == 4) vx_masks_vox4(dc, row, V, v, m);      // V % 4 == 0 and v % 4 == 0 on the T = 4 path
    else { m[0] = vx_drop1(dc, row * (uint64_t)V + (uint64_t)v); m[1] = m[2] = m[3] = 1.0f; }
}
}  // namespace

// =====================================================================================================================
// pre: LN + up to three 1x1 projections
// =====================================================================================================================
struct VxLnPwMod {
    const float *x, *gamma, *beta;
    const float* w[3];
    const float* b[3];
    float* xn;
    float* out[3];
};
struct VxLnPw {
    VxLnPwMod m[4];
    int C, NS;
    int J[3];
    long V;
    int tiles_per_b;
    float eps;
    int s2d, C0, gh, gw;      // s2d: x is (B, C0, 2gd, 2gh, 2gw) and channel k = sub * C0 + c of voxel (d,h,w) is x[c][2d + (sub>>2)][2h + ((sub>>1)&1)][2w + (sub&1)]
};

template <int T, int NW>
__device__ __forceinline__ void load_tile_s2d(float* __restrict__ dst, const float* __restrict__ xb /* sample base */, int C, int C0, int gh, int gw, long V, long v0) {
    constexpr int S = Geo<T>::S, RP = 4 * NW, U = 8;
    static_assert(T == 1, "s2d gather: T = 1 tiles only");
    const int cg = threadIdx.x & 15;
    const long v = v0 + cg;
    const bool ok = v < V;
    const long vc = ok ? v : 0;
    const int w_ = (int)(vc % gw), h_ = (int)((vc / gw) % gh), d_ = (int)(vc / ((long)gw * gh));
    const int HW2 = (2 * gh) * (2 * gw);
    const long gd2 = 2 * (V / ((long)gh * gw));
    const long cstride = gd2 * HW2;                                   // one input channel (fine volume)
    const float* __restrict__ xv = xb + (long)(2 * d_) * HW2 + (long)(2 * h_) * (2 * gw) + 2 * w_;
    // row = sub * C0 + c walked with a running (sub, c) pair: the division by C0 and the 64-bit products per element were most of this loop's instructions
    int row = threadIdx.x >> 4;
    int sub = row / C0, c = row - sub * C0;                           // (once per thread)
    for (int row0 = row; row0 < C; row0 += U * RP) {
        float t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = row0 + u * RP < C;                        // rows beyond C: reload the thread's last valid row (never stored)
            const int so = (sub >> 2) * HW2 + ((sub >> 1) & 1) * (2 * gw) + (sub & 1);
            t[u] = xv[(long)c * cstride + so];
            if (in && row0 + (u + 1) * RP < C) {                      // advance (sub, c) by RP rows (for u = U - 1: to the next batch's first row)
                c += RP;
                while (c >= C0) { c -= C0; ++sub; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rw = row0 + u * RP;
            if (rw < C) dst[rw * S + cg] = ok ? t[u] : 0.0f;
        }
    }
}

// parameters into LDS: dst[i] = src[i] (src == NULL -> fill), all loads of the block issued before the barrier that follows
template <int NW>
__device__ __forceinline__ void stage_vec(float* __restrict__ dst, const float* __restrict__ src, int n, float fill) {
    for (int i = threadIdx.x; i < n; i += 64 * NW) dst[i] = src ? src[i] : fill;
}

template <int T, int NW>
__global__ void __launch_bounds__(64 * NW) vx_ln_pw_fwd_k(VxLnPw p) {
    constexpr int NT = Geo<T>::NT, S = Geo<T>::S, NP = 64 * NW / NT;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const VxLnPwMod& M = p.m[blockIdx.y];
    const int C = p.C;
    const int Jt = p.J[0] + p.J[1] + p.J[2];
    float* __restrict__ X = vx_pf_lds;
    float* __restrict__ red = X + C * S;             // NP * NT
    float* __restrict__ gam = red + NP * NT;         // C
    float* __restrict__ bet = gam + C;               // C
    float* __restrict__ bia = bet + C;               // Jt
    const int b = blockIdx.x / p.tiles_per_b;
    const long V = p.V, v0 = (long)(blockIdx.x % p.tiles_per_b) * NT;
    stage_vec<NW>(gam, M.gamma, C, 1.0f);
    stage_vec<NW>(bet, M.beta, C, 0.0f);
    {
        int j0 = 0;
        for (int sg = 0; sg < p.NS; ++sg) { stage_vec<NW>(bia + j0, M.b[sg], p.J[sg], 0.0f); j0 += p.J[sg]; }
    }
    if constexpr (T == 1) {
        if (p.s2d) load_tile_s2d<T, NW>(X, M.x + (long)b * C * V, C, p.C0, p.gh, p.gw, V, v0);      // (C0 * 8 V fine elements per sample = C * V)
        else load_tile<T, NW>(X, M.x + (long)b * C * V, C, V, v0);
    } else load_tile<T, NW>(X, M.x + (long)b * C * V, C, V, v0);
    __syncthreads();
    float mean, rstd;
    tile_ln_stats<T, NW>(X, C, p.eps, red, mean, rstd);
    {
        const int col = threadIdx.x % NT, part = threadIdx.x / NT;
        const bool live = v0 + col < V;
        float* __restrict__ xn = M.xn ? M.xn + (long)b * C * V + v0 + col : nullptr;
        for (int c = part; c < C; c += NP) {
            const float n = fmaf(gam[c], (X[c * S + col] - mean) * rstd, bet[c]);
            X[c * S + col] = n;
            if (xn && live) xn[(long)c * V] = n;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    int j0 = 0;
    for (int sg = 0; sg < p.NS; ++sg) {
        const int J = p.J[sg];
        const float* __restrict__ bias = bia + j0;
        float* __restrict__ out = M.out[sg] + (long)b * J * V;
        tile_gemm1<T, NW, false>(M.w[sg], C, C, 1, J, X, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = 16 * mt + 4 * q + reg;
                const float bb = bias[m];
                if constexpr (T == 4) {
                    const long v = v0 + 4 * r;
                    if (v < V) *reinterpret_cast<float4*>(out + (long)m * V + v) = make_float4(acc[0][reg] + bb, acc[1][reg] + bb, acc[2][reg] + bb, acc[3][reg] + bb);
                } else {
                    const long v = v0 + r;
                    if (v < V) out[(long)m * V + v] = acc[0][reg] + bb;
                }
            }
        });
        j0 += J;
    }
}

// backward: dxn = sum_s W_s^T dout_s ;  dx = dres + LN'(dxn) ;  per-block partial sums of dgamma / dbeta -> part[(mod-local block) * 2C + {c, C + c}]
struct VxLnPwBwdMod {
    const float *x, *gamma;
    const float* w[3];
    const float* dout[3];
    const float* dres;
    float* dx;
    float* part;
};
struct VxLnPwBwd {
    VxLnPwBwdMod m[4];
    int C, NS;
    int J[3];
    long V;
    int tiles_per_b;
    float eps;
    int s2d, C0, gh, gw;
};

template <int T, int NW>
__global__ void __launch_bounds__(64 * NW) vx_ln_pw_bwd_k(VxLnPwBwd p) {
    constexpr int NT = Geo<T>::NT, S = Geo<T>::S, NP = 64 * NW / NT;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const VxLnPwBwdMod& M = p.m[blockIdx.y];
    const int C = p.C;
    int Jt = 0;
    for (int sg = 0; sg < p.NS; ++sg) Jt += p.J[sg];
    float* __restrict__ D = vx_pf_lds;               // [Jt][S]  the output gradients; its first C rows are re-used for dres once the GEMM is done
    float* __restrict__ X = D + Jt * S;              // [C][S]   x, then xhat
    float* __restrict__ G = X + C * S;               // [C][S]   dxn
    float* __restrict__ red = G + C * S;             // 2 NP NT
    float* __restrict__ gam = red + 2 * NP * NT;     // C
    const int b = blockIdx.x / p.tiles_per_b;
    const long V = p.V, v0 = (long)(blockIdx.x % p.tiles_per_b) * NT;
    stage_vec<NW>(gam, M.gamma, C, 1.0f);
    {
        int j0 = 0;
        for (int sg = 0; sg < p.NS; ++sg) {
            load_tile<T, NW>(D + j0 * S, M.dout[sg] + (long)b * p.J[sg] * V, p.J[sg], V, v0);
            j0 += p.J[sg];
        }
    }
    if constexpr (T == 1) {
        if (p.s2d) load_tile_s2d<T, NW>(X, M.x + (long)b * C * V, C, p.C0, p.gh, p.gw, V, v0);
        else load_tile<T, NW>(X, M.x + (long)b * C * V, C, V, v0);
    } else load_tile<T, NW>(X, M.x + (long)b * C * V, C, V, v0);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    {
        const float* const ws[3] = {M.w[0], M.w[1], M.w[2]};
        const int ks[3] = {p.J[0], p.J[1], p.J[2]};
        tile_gemm<T, NW, true>(p.NS, ws, ks, 1, C, C, D, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int m = 16 * mt + 4 * q + reg;
                if constexpr (T == 4) *reinterpret_cast<float4*>(G + m * S + 4 * r) = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
                else G[m * S + r] = acc[0][reg];
            }
        });
    }
    float mean, rstd;
    tile_ln_stats<T, NW>(X, C, p.eps, red, mean, rstd);        // (its first barrier also publishes G and retires every read of D)
    const bool has_res = M.dres != nullptr && Jt >= C;         // Jt >= C always holds for the PWA projections (2 ch_qk + ch_v >= C)
    if (has_res) load_tile<T, NW>(D, M.dres + (long)b * C * V, C, V, v0);
    const int col = threadIdx.x % NT, part = threadIdx.x / NT;
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = part; c < C; c += NP) {
        const float xh = (X[c * S + col] - mean) * rstd;
        const float g = G[c * S + col] * gam[c];
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    __syncthreads();                                           // red is re-used; the dres tile is published
    red[part * NT + col] = s1;
    red[(NP + part) * NT + col] = s2;
    __syncthreads();
    float t1 = 0.0f, t2 = 0.0f;
#pragma unroll 4
    for (int p_ = 0; p_ < NP; ++p_) { t1 += red[p_ * NT + col]; t2 += red[(NP + p_) * NT + col]; }
    t1 /= (float)C; t2 /= (float)C;
    const bool live = v0 + col < V;
    float* __restrict__ pp = M.part ? M.part + (long)blockIdx.x * 2 * C : nullptr;
    // PatchMerging: the gradient goes back through the strided gather (every fine voxel has exactly one (sub, coarse voxel))
    long s2d_base = 0, HW2 = 0, gd2 = 0;
    if (p.s2d) {
        const long v = live ? v0 + col : 0;
        const int w_ = (int)(v % p.gw), h_ = (int)((v / p.gw) % p.gh), d_ = (int)(v / ((long)p.gw * p.gh));
        HW2 = (long)(2 * p.gh) * (2 * p.gw);
        gd2 = 2 * (V / ((long)p.gh * p.gw));
        s2d_base = (long)b * C * V + (long)(2 * d_) * HW2 + (long)(2 * h_) * (2 * p.gw) + 2 * w_;
    }
    int ssub = p.s2d ? part / p.C0 : 0, sc0 = p.s2d ? part - ssub * p.C0 : 0;      // s2d: channel c = ssub * C0 + sc0, walked without a division per channel
    const long cstr2 = gd2 * HW2;
    for (int c = part; c < C; c += NP) {
        const float xh = (X[c * S + col] - mean) * rstd;
        const float dn = G[c * S + col];
        const float g = dn * gam[c];
        float dg = dn * xh, db = dn;                           // columns beyond V hold dn = 0
#pragma unroll
        for (int o = 1; o < NT; o <<= 1) { dg += __shfl_xor(dg, o, 64); db += __shfl_xor(db, o, 64); }
        if (pp && col == 0) { pp[c] = dg; pp[C + c] = db; }
        if (live) {
            float dxv = rstd * (g - t1 - xh * t2);
            if (p.s2d) {
                M.dx[s2d_base + (long)sc0 * cstr2 + (long)(ssub >> 2) * HW2 + (long)((ssub >> 1) & 1) * (2 * p.gw) + (ssub & 1)] = dxv;
            } else {
                if (has_res) dxv += D[c * S + col];
                else if (M.dres) dxv += M.dres[((long)b * C + c) * V + v0 + col];
                M.dx[((long)b * C + c) * V + v0 + col] = dxv;
            }
        }
        if (p.s2d) { sc0 += NP; while (sc0 >= p.C0) { sc0 -= p.C0; ++ssub; } }
    }
}

// =====================================================================================================================
// post: mix conv + residual -> LN -> FFN -> residual          (the 16^3 / 8^3 / 4^3 levels, where the expanded FFN is not covered by mlp.hip or the tile chain is faster)
// =====================================================================================================================
struct VxPostMod {
    const float *s, *x, *wm, *bm, *gamma, *beta, *w1, *b1, *w2, *b2;
    float *y, *out;
    // backward
    const float* dout;
    float *ds, *dxres, *part;
    float *sc_n, *sc_h, *sc_da, *sc_dz, *sc_dmix;      // operands of the grouped weight-gradient launch: n (C,V), h (R,V), da (R,V), dz (C,V), dmix (C,V) per sample
    unsigned long long site_mix, site1, site2;
};
struct VxPost {
    VxPostMod m[4];
    int C, Cv, R;
    long V;
    int tiles_per_b;
    float eps, alpha, p_mix, p_ffn;
    const void* seed_ptr;
};

// T consecutive floats of an LDS / global row (16-byte aligned for T = 4)
template <int T>
__device__ __forceinline__ void ldv(const float* __restrict__ q, float (&o)[T]) {
    if constexpr (T == 4) { const float4 t = *reinterpret_cast<const float4*>(q); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
    else o[0] = q[0];
}
template <int T>
__device__ __forceinline__ void stv(float* __restrict__ q, const float (&o)[T]) {
    if constexpr (T == 4) *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
    else q[0] = o[0];
}

// T = 4: 64-voxel tiles, 16-byte accesses and one Philox call per four voxels (the 16^3 level); T = 1: 16-voxel tiles (8^3 / 4^3)
template <int T, int NW>
__global__ void __launch_bounds__(64 * NW) vx_pwa_post_fwd_k(VxPost p) {
    constexpr int NT = Geo<T>::NT, S = Geo<T>::S, NP = 64 * NW / NT;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const VxPostMod& M = p.m[blockIdx.y];
    const int C = p.C, Cv = p.Cv, R = p.R;
    float* __restrict__ Sx = vx_pf_lds;              // [Cv][S]
    float* __restrict__ Y = Sx + Cv * S;             // [C][S]   x, then y
    float* __restrict__ N = Y + C * S;               // [C][S]
    float* __restrict__ Hh = N + C * S;              // [R][S]
    float* __restrict__ red = Hh + R * S;            // NP NT
    float* __restrict__ prm = red + NP * NT;         // bm (C) | gamma (C) | beta (C) | b2 (C) | b1 (R)
    float* __restrict__ bm = prm, *gam = prm + C, *bet = prm + 2 * C, *b2 = prm + 3 * C, *b1 = prm + 4 * C;
    const int b = blockIdx.x / p.tiles_per_b;
    const long V = p.V, v0 = (long)(blockIdx.x % p.tiles_per_b) * NT;
    const VxDropCtx dm = drop_ctx_dev(p.seed_ptr, M.site_mix, p.p_mix);
    const VxDropCtx d1 = drop_ctx_dev(p.seed_ptr, M.site1, p.p_ffn);
    const VxDropCtx d2 = drop_ctx_dev(p.seed_ptr, M.site2, p.p_ffn);
    stage_vec<NW>(bm, M.bm, C, 0.0f);
    stage_vec<NW>(gam, M.gamma, C, 1.0f);
    stage_vec<NW>(bet, M.beta, C, 0.0f);
    stage_vec<NW>(b2, M.b2, C, 0.0f);
    stage_vec<NW>(b1, M.b1, R, 0.0f);
    load_tile<T, NW>(Sx, M.s + (long)b * Cv * V, Cv, V, v0);
    load_tile<T, NW>(Y, M.x + (long)b * C * V, C, V, v0);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int cr = T * r;                            // first tile column of this lane's accumulator elements
    const long v = v0 + cr;
    const bool vlive = v < V;                        // (T = 4: V % 4 == 0, the four voxels live or die together)
    // y = alpha x + drop(Wm s + bm)          (element (m, r) of Y is read and written by its owner only)
    tile_gemm1<T, NW, false>(M.wm, Cv, Cv, 1, C, Sx, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * q + reg;
            const long row = (long)b * C + m;
            float yv[T], mk[4];
            ldv<T>(Y + m * S + cr, yv);
            if (vlive) {
                acc_masks<T>(dm, (uint64_t)row, V, v, mk);
#pragma unroll
                for (int j = 0; j < T; ++j) yv[j] = fmaf(p.alpha, yv[j], (acc[j][reg] + bm[m]) * mk[j]);
                stv<T>(M.y + row * V + v, yv);
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) yv[j] = 0.0f;
            }
            stv<T>(Y + m * S + cr, yv);
        }
    });
    __syncthreads();
    float mean, rstd;
    tile_ln_stats<T, NW>(Y, C, p.eps, red, mean, rstd);
    {
        const int col = threadIdx.x % NT, part = threadIdx.x / NT;
        for (int c = part; c < C; c += NP) N[c * S + col] = fmaf(gam[c], (Y[c * S + col] - mean) * rstd, bet[c]);
    }
    __syncthreads();
    // h = drop1(gelu(W1 n + b1))
    tile_gemm1<T, NW, false>(M.w1, C, C, 1, R, N, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            float hv[T], mk[4];
            if (vlive) {
                acc_masks<T>(d1, (uint64_t)((long)b * R + j), V, v, mk);
#pragma unroll
                for (int i = 0; i < T; ++i) hv[i] = vx_gelu_fast(acc[i][reg] + b1[j]) * mk[i];
            } else {
#pragma unroll
                for (int i = 0; i < T; ++i) hv[i] = 0.0f;
            }
            stv<T>(Hh + j * S + cr, hv);
        }
    });
    __syncthreads();
    // out = y + drop2(W2 h + b2)
    tile_gemm1<T, NW, false>(M.w2, R, R, 1, C, Hh, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * q + reg;
            if (vlive) {
                const long row = (long)b * C + m;
                float ov[T], mk[4];
                ldv<T>(Y + m * S + cr, ov);
                acc_masks<T>(d2, (uint64_t)row, V, v, mk);
#pragma unroll
                for (int i = 0; i < T; ++i) ov[i] = fmaf(acc[i][reg] + b2[m], mk[i], ov[i]);
                stv<T>(M.out + row * V + v, ov);
            }
        }
    });
}

template <int T, int NW>
__global__ void __launch_bounds__(64 * NW) vx_pwa_post_bwd_k(VxPost p) {
    constexpr int NT = Geo<T>::NT, S = Geo<T>::S, NP = 64 * NW / NT, NRP = 4 * NW;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const VxPostMod& M = p.m[blockIdx.y];
    const int C = p.C, Cv = p.Cv, R = p.R;
    float* __restrict__ DO = vx_pf_lds;              // [C][S]  dout
    float* __restrict__ DZ = DO + C * S;             // [C][S]  dout * mask2, later dn
    float* __restrict__ Y = DZ + C * S;              // [C][S]  y, later xhat
    float* __restrict__ N = Y + C * S;               // [C][S]  LN(y), later dmix
    float* __restrict__ A = N + C * S;               // [R][S]  pre-activation a, then da
    float* __restrict__ red = A + R * S;             // 2 NP NT
    float* __restrict__ cs = red + 2 * NP * NT;      // rstd | t1 | t2 per tile column (3 NT)
    float* __restrict__ gam = cs + 3 * NT, *bet = gam + C, *b1 = bet + C;      // C | C | R
    const int b = blockIdx.x / p.tiles_per_b;
    const long V = p.V, v0 = (long)(blockIdx.x % p.tiles_per_b) * NT;
    const VxDropCtx dm = drop_ctx_dev(p.seed_ptr, M.site_mix, p.p_mix);
    const VxDropCtx d1 = drop_ctx_dev(p.seed_ptr, M.site1, p.p_ffn);
    const VxDropCtx d2 = drop_ctx_dev(p.seed_ptr, M.site2, p.p_ffn);
    const int col = threadIdx.x % NT, part = threadIdx.x / NT;          // column layout: LN statistics, one voxel per thread
    const bool clive = v0 + col < V;
    const int qd = threadIdx.x & 15, rp = threadIdx.x >> 4, cq = T * qd;   // group layout: T consecutive voxels per thread (the loops that draw dropout masks)
    const long vq = v0 + cq;
    const bool qlive = vq < V;
    stage_vec<NW>(gam, M.gamma, C, 1.0f);
    stage_vec<NW>(bet, M.beta, C, 0.0f);
    stage_vec<NW>(b1, M.b1, R, 0.0f);
    load_tile<T, NW>(Y, M.y + (long)b * C * V, C, V, v0);
    load_tile<T, NW>(DO, M.dout + (long)b * C * V, C, V, v0);
    __syncthreads();
    // dz = dout * mask2 (site 2 over (B, C, V)); also kept in global for dW2 = dz h^T
    for (int c = rp; c < C; c += NRP) {
        const long row = (long)b * C + c;
        float g[T], mk[4];
        ldv<T>(DO + c * S + cq, g);
        if (qlive) {
            acc_masks<T>(d2, (uint64_t)row, V, vq, mk);
#pragma unroll
            for (int j = 0; j < T; ++j) g[j] *= mk[j];
            stv<T>(M.sc_dz + row * V + vq, g);
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) g[j] = 0.0f;
        }
        stv<T>(DZ + c * S + cq, g);
    }
    float mean, rstd;
    tile_ln_stats<T, NW>(Y, C, p.eps, red, mean, rstd);
    for (int c = part; c < C; c += NP) {
        const float xh = (Y[c * S + col] - mean) * rstd;
        const float n = fmaf(gam[c], xh, bet[c]);
        N[c * S + col] = n;
        Y[c * S + col] = xh;
        if (clive) M.sc_n[((long)b * C + c) * V + v0 + col] = n;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int cr = T * r;
    const long v = v0 + cr;
    const bool vlive = v < V;
    // a = W1 n + b1 ; h = drop1(gelu(a)) -> global (dW2's operand)
    tile_gemm1<T, NW, false>(M.w1, C, C, 1, R, N, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            float a[T];
#pragma unroll
            for (int i = 0; i < T; ++i) a[i] = acc[i][reg] + b1[j];
            stv<T>(A + j * S + cr, a);
            if (vlive) {
                const long row = (long)b * R + j;
                float mk[4];
                acc_masks<T>(d1, (uint64_t)row, V, v, mk);
#pragma unroll
                for (int i = 0; i < T; ++i) a[i] = vx_gelu_fast(a[i]) * mk[i];
                stv<T>(M.sc_h + row * V + v, a);
            }
        }
    });
    __syncthreads();
    // da = (W2^T dz) * mask1 * gelu'(a)      (each accumulator element meets its own a: same (row, column) ownership as the stage above)
    tile_gemm1<T, NW, true>(M.w2, C, 1, R, R, DZ, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            const long row = (long)b * R + j;
            float da[T];
            if (vlive) {
                float mk[4];
                ldv<T>(A + j * S + cr, da);
                acc_masks<T>(d1, (uint64_t)row, V, v, mk);
#pragma unroll
                for (int i = 0; i < T; ++i) da[i] = acc[i][reg] * mk[i] * vx_gelu_grad_fast(da[i]);
                stv<T>(M.sc_da + row * V + v, da);
            } else {
#pragma unroll
                for (int i = 0; i < T; ++i) da[i] = 0.0f;
            }
            stv<T>(A + j * S + cr, da);
        }
    });
    __syncthreads();
    // dn = W1^T da  -> DZ (dz is no longer needed)
    tile_gemm1<T, NW, true>(M.w1, R, 1, C, C, A, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            float dn[T];
#pragma unroll
            for (int i = 0; i < T; ++i) dn[i] = acc[i][reg];
            stv<T>(DZ + (16 * mt + 4 * q + reg) * S + cr, dn);
        }
    });
    __syncthreads();
    // LN backward: dy = dout + rstd (g - mean(g) - xhat mean(g xhat)),  g = dn gamma
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = part; c < C; c += NP) {
        const float g = DZ[c * S + col] * gam[c];
        s1 += g;
        s2 = fmaf(g, Y[c * S + col], s2);
    }
    red[part * NT + col] = s1;
    red[(NP + part) * NT + col] = s2;
    __syncthreads();
    float t1 = 0.0f, t2 = 0.0f;
#pragma unroll 4
    for (int p_ = 0; p_ < NP; ++p_) { t1 += red[p_ * NT + col]; t2 += red[(NP + p_) * NT + col]; }
    t1 /= (float)C; t2 /= (float)C;
    float rs_[T], t1_[T], t2_[T];
    if constexpr (T == 4) {                                          // the group layout below needs the statistics of four columns
        if (part == 0) { cs[col] = rstd; cs[NT + col] = t1; cs[2 * NT + col] = t2; }
        __syncthreads();
        ldv<T>(cs + cq, rs_); ldv<T>(cs + NT + cq, t1_); ldv<T>(cs + 2 * NT + cq, t2_);
    } else { rs_[0] = rstd; t1_[0] = t1; t2_[0] = t2; }              // (T = 1: col == qd)
    float* __restrict__ pp = M.part + (long)blockIdx.x * 2 * C;
    for (int c = rp; c < C; c += NRP) {
        const long row = (long)b * C + c;
        float xh[T], dn[T], dy[T];
        ldv<T>(Y + c * S + cq, xh);
        ldv<T>(DZ + c * S + cq, dn);
        float dg = 0.0f, db = 0.0f;
#pragma unroll
        for (int j = 0; j < T; ++j) { dg = fmaf(dn[j], xh[j], dg); db += dn[j]; }     // columns beyond V hold dn = 0
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { dg += __shfl_xor(dg, o, 64); db += __shfl_xor(db, o, 64); }
        if (qd == 0) { pp[c] = dg; pp[C + c] = db; }
        if (qlive) {
            float mk[4], dx[T];
            ldv<T>(DO + c * S + cq, dy);
            acc_masks<T>(dm, (uint64_t)row, V, vq, mk);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                dy[j] += rs_[j] * (dn[j] * gam[c] - t1_[j] - xh[j] * t2_[j]);
                dx[j] = p.alpha * dy[j];
                dy[j] *= mk[j];
            }
            stv<T>(M.dxres + row * V + vq, dx);
            stv<T>(M.sc_dmix + row * V + vq, dy);
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) dy[j] = 0.0f;
        }
        stv<T>(N + c * S + cq, dy);                                  // N (LN output) is no longer needed
    }
    __syncthreads();
    // ds = Wm^T dmix
    tile_gemm1<T, NW, true>(M.wm, C, 1, Cv, Cv, N, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * q + reg;
            if (vlive) {
                float o[T];
#pragma unroll
                for (int i = 0; i < T; ++i) o[i] = acc[i][reg];
                stv<T>(M.ds + ((long)b * Cv + m) * V + v, o);
            }
        }
    });
}

// --------------------------------------------------------------------------------------------------------------------- host
static inline int pf_tile_t(long V) { return ((V & 3) == 0 && V >= 2048) ? 4 : 1; }
static inline int pf_nw(long blocks) { return blocks < 256 ? 8 : 4; }

extern "C" int vx_ln_pw_tiles(int B, long V) {          // answer, not a status: blocks per modality of the pre kernels (= rows of the dgamma / dbeta partials)
    const int T = pf_tile_t(V);
    return B * vx_cdiv(V, 16 * T);
}

// dynamic LDS a block of these kernels may ask for on THIS device (queried once; 160 KB on gfx950, 64 KB on gfx942): the *_ok gates are bounded by it, so a shape
// that does not fit falls back to the per-operator kernels instead of failing at launch.  Without a device (the CPU-side symbol checks) the gfx950 figure.
static size_t pf_lds_limit() {
    static size_t lim = 0;
    if (!lim) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && v > 0) lim = (size_t)v;
        else if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) lim = (size_t)v;
        else lim = 160 * 1024;
    }
    return lim;
}
static inline size_t pf_lds_budget() { const size_t l = pf_lds_limit(); return l >= 160 * 1024 ? 150 * 1024 : l - (l >> 4); }

extern "C" int vx_ln_pw_ok(int C, int NS, const int* J, long V, int s2d) {
    if (C % 16 || C < 16 || C > 1024 || NS < 1 || NS > 3 || V < 1) return 0;
    int Jt = 0;
    for (int s = 0; s < NS; ++s) { if (J[s] % 16 || J[s] < 16) return 0; Jt += J[s]; }
    const int T = s2d ? 1 : pf_tile_t(V);
    const int S = T == 1 ? 20 : 64;
    const size_t lds = ((size_t)(Jt + 2 * C) * S + 2 * 32 * 16 + 2 * C + Jt + 64) * sizeof(float);      // the backward kernel is the larger one
    return lds <= pf_lds_budget() ? 1 : 0;
}

template <int T, int NW, class P>
static void pf_launch(void (*k)(P), const P& p, dim3 grid, size_t shm, hipStream_t st) {
    if (shm > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pf_lds_limit());
        if (e != hipSuccess) return;       // not launched: the sticky error surfaces in the caller's VX_LAUNCH_CHECK (the *_ok gates keep such shapes off this path)
    }
    k<<<grid, dim3(64 * NW), shm, st>>>(p);
}

/* ptrs (host array), per modality 13 entries: x, gamma, beta, w0, b0, w1, b1, w2, b2, xn, out0, out1, out2 (unused segments / biases NULL) */
extern "C" int vx_ln_pw_fwd(const void* const* ptrs, int M, int NS, const int* J, int B, int C, long V, float eps, int s2d, int gd, int gh, int gw, void* stream) {
    VX_REQUIRE(ptrs && J && M >= 1 && M <= 4 && B > 0, "vx_ln_pw_fwd: bad args");
    VX_REQUIRE(vx_ln_pw_ok(C, NS, J, V, s2d), "vx_ln_pw_fwd: unsupported shape C=%d NS=%d V=%ld", C, NS, V);
    VX_REQUIRE(!s2d || ((long)gd * gh * gw == V && C % 8 == 0), "vx_ln_pw_fwd: s2d geometry");
    VxLnPw p = {};
    for (int m = 0; m < M; ++m) {
        const void* const* q = ptrs + 13 * m;
        VxLnPwMod& d = p.m[m];
        d.x = (const float*)q[0]; d.gamma = (const float*)q[1]; d.beta = (const float*)q[2];
        for (int s = 0; s < 3; ++s) { d.w[s] = (const float*)q[3 + 2 * s]; d.b[s] = (const float*)q[4 + 2 * s]; d.out[s] = (float*)q[10 + s]; }
        d.xn = (float*)q[9];
        VX_REQUIRE(d.x && d.gamma && d.beta, "vx_ln_pw_fwd: null input");
        for (int s = 0; s < NS; ++s) VX_REQUIRE(d.w[s] && d.out[s], "vx_ln_pw_fwd: null segment");
    }
    p.C = C; p.NS = NS; p.V = V; p.eps = eps; p.s2d = s2d; p.C0 = C / 8; p.gh = gh; p.gw = gw;
    for (int s = 0; s < 3; ++s) p.J[s] = s < NS ? J[s] : 0;
    const int T = s2d ? 1 : pf_tile_t(V);
    p.tiles_per_b = vx_cdiv(V, 16 * T);
    const long blocks = (long)B * p.tiles_per_b;
    dim3 grid((unsigned)blocks, M);
    hipStream_t st = (hipStream_t)stream;
    const int S = T == 1 ? 20 : 64;
    int Jt = 0;
    for (int s = 0; s < NS; ++s) Jt += J[s];
    const size_t extra = 2 * C + Jt;                      // gamma, beta, biases
    if (T == 4) pf_launch<4, 4>(vx_ln_pw_fwd_k<4, 4>, p, grid, ((size_t)C * S + 4 * 64 + extra) * sizeof(float), st);
    else if (pf_nw(blocks * M) == 8) pf_launch<1, 8>(vx_ln_pw_fwd_k<1, 8>, p, grid, ((size_t)C * S + 32 * 16 + extra) * sizeof(float), st);
    else pf_launch<1, 4>(vx_ln_pw_fwd_k<1, 4>, p, grid, ((size_t)C * S + 16 * 16 + extra) * sizeof(float), st);
    VX_LAUNCH_CHECK("vx_ln_pw_fwd");
    return 0;
}

/* ptrs, per modality 11 entries: x, gamma, w0, w1, w2, dout0, dout1, dout2, dres (NULL ok), dx, part (vx_ln_pw_tiles(B, V) x 2C floats) */
extern "C" int vx_ln_pw_bwd(const void* const* ptrs, int M, int NS, const int* J, int B, int C, long V, float eps, int s2d, int gd, int gh, int gw, void* stream) {
    VX_REQUIRE(ptrs && J && M >= 1 && M <= 4 && B > 0, "vx_ln_pw_bwd: bad args");
    VX_REQUIRE(vx_ln_pw_ok(C, NS, J, V, s2d), "vx_ln_pw_bwd: unsupported shape C=%d NS=%d V=%ld", C, NS, V);
    VX_REQUIRE(!s2d || ((long)gd * gh * gw == V && C % 8 == 0), "vx_ln_pw_bwd: s2d geometry");
    VxLnPwBwd p = {};
    int Jt = 0;
    for (int m = 0; m < M; ++m) {
        const void* const* q = ptrs + 11 * m;
        VxLnPwBwdMod& d = p.m[m];
        d.x = (const float*)q[0]; d.gamma = (const float*)q[1];
        for (int s = 0; s < 3; ++s) { d.w[s] = (const float*)q[2 + s]; d.dout[s] = (const float*)q[5 + s]; }
        d.dres = (const float*)q[8]; d.dx = (float*)q[9]; d.part = (float*)q[10];
        VX_REQUIRE(d.x && d.gamma && d.dx, "vx_ln_pw_bwd: null input");
        for (int s = 0; s < NS; ++s) VX_REQUIRE(d.w[s] && d.dout[s], "vx_ln_pw_bwd: null segment");
        VX_REQUIRE(!(s2d && d.dres), "vx_ln_pw_bwd: dres is not supported with the s2d gather");
    }
    p.C = C; p.NS = NS; p.V = V; p.eps = eps; p.s2d = s2d; p.C0 = C / 8; p.gh = gh; p.gw = gw;
    for (int s = 0; s < 3; ++s) { p.J[s] = s < NS ? J[s] : 0; Jt += p.J[s]; }
    const int T = s2d ? 1 : pf_tile_t(V);
    p.tiles_per_b = vx_cdiv(V, 16 * T);
    const long blocks = (long)B * p.tiles_per_b;
    dim3 grid((unsigned)blocks, M);
    hipStream_t st = (hipStream_t)stream;
    const int S = T == 1 ? 20 : 64;
    const size_t base = (size_t)(Jt + 2 * C) * S + C;
    if (T == 4) pf_launch<4, 4>(vx_ln_pw_bwd_k<4, 4>, p, grid, (base + 2 * 4 * 64) * sizeof(float), st);
    else if (pf_nw(blocks * M) == 8) pf_launch<1, 8>(vx_ln_pw_bwd_k<1, 8>, p, grid, (base + 2 * 32 * 16) * sizeof(float), st);
    else pf_launch<1, 4>(vx_ln_pw_bwd_k<1, 4>, p, grid, (base + 2 * 16 * 16) * sizeof(float), st);
    VX_LAUNCH_CHECK("vx_ln_pw_bwd");
    return 0;
}

// tile width of the post kernels: 64 voxels (T = 4) where the grid is large enough to fill the chip with them (VELOXSEG_POST_T4=0: 16-voxel tiles everywhere)
static int post_tile_t(long V) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("VELOXSEG_POST_T4"); on = (e && atoi(e) == 0) ? 0 : 1; }
    return on ? pf_tile_t(V) : 1;
}
static inline size_t post_lds_floats(int T, int NW, int C, int Cv, int R, bool bwd) {
    const size_t S = T == 4 ? 64 : 20, NT = 16 * T, NP = 64 * NW / NT;
    return bwd ? (size_t)(4 * C + R) * S + 2 * NP * NT + 3 * NT + 2 * C + R : (size_t)(Cv + 2 * C + R) * S + NP * NT + 4 * C + R;
}
extern "C" int vx_pwa_post_ok(int C, int Cv, int R, long V) {
    if (C % 16 || Cv % 16 || R % 16 || C < 16 || Cv < 16 || R < 16 || V < 1) return 0;
    const int T = post_tile_t(V), NW = T == 4 ? 4 : 8;
    const size_t lds = (post_lds_floats(T, NW, C, Cv, R, true) > post_lds_floats(T, NW, C, Cv, R, false) ? post_lds_floats(T, NW, C, Cv, R, true) : post_lds_floats(T, NW, C, Cv, R, false)) * sizeof(float);
    return lds <= pf_lds_budget() ? 1 : 0;
}
extern "C" int vx_pwa_post_tiles(int B, long V) { return B * vx_cdiv(V, 16 * post_tile_t(V)); }

/* ptrs, per modality 24 entries: s, x, wm, bm, gamma, beta, w1, b1, w2, b2, y, out, dout, ds, dxres, part, sc_n, sc_h, sc_da, sc_dz, sc_dmix, site_mix, site1, site2
 * (the three sites as integers cast to pointers); forward uses entries 0..11 and the sites, backward all but `out` */
static int post_fill(VxPost& p, const void* const* ptrs, int M, int B, int C, int Cv, int R, long V, float eps, float alpha, const void* seed_ptr, float p_mix, float p_ffn, bool bwd) {
    for (int m = 0; m < M; ++m) {
        const void* const* q = ptrs + 24 * m;
        VxPostMod& d = p.m[m];
        d.s = (const float*)q[0]; d.x = (const float*)q[1]; d.wm = (const float*)q[2]; d.bm = (const float*)q[3]; d.gamma = (const float*)q[4]; d.beta = (const float*)q[5];
        d.w1 = (const float*)q[6]; d.b1 = (const float*)q[7]; d.w2 = (const float*)q[8]; d.b2 = (const float*)q[9]; d.y = (float*)q[10]; d.out = (float*)q[11];
        d.dout = (const float*)q[12]; d.ds = (float*)q[13]; d.dxres = (float*)q[14]; d.part = (float*)q[15];
        d.sc_n = (float*)q[16]; d.sc_h = (float*)q[17]; d.sc_da = (float*)q[18]; d.sc_dz = (float*)q[19]; d.sc_dmix = (float*)q[20];
        d.site_mix = (unsigned long long)(uintptr_t)q[21]; d.site1 = (unsigned long long)(uintptr_t)q[22]; d.site2 = (unsigned long long)(uintptr_t)q[23];
        VX_REQUIRE(d.wm && d.gamma && d.beta && d.w1 && d.b1 && d.w2 && d.b2 && d.y, "vx_pwa_post: null parameter");
        if (!bwd) VX_REQUIRE(d.s && d.x && d.out, "vx_pwa_post_fwd: null tensor");
        else VX_REQUIRE(d.dout && d.ds && d.dxres && d.part && d.sc_n && d.sc_h && d.sc_da && d.sc_dz && d.sc_dmix, "vx_pwa_post_bwd: null tensor");
    }
    p.C = C; p.Cv = Cv; p.R = R; p.V = V; p.eps = eps; p.alpha = alpha; p.p_mix = p_mix; p.p_ffn = p_ffn; p.seed_ptr = seed_ptr;
    p.tiles_per_b = vx_cdiv(V, 16 * post_tile_t(V));
    (void)B;
    return 0;
}

extern "C" int vx_pwa_post_fwd(const void* const* ptrs, int M, int B, int C, int Cv, int R, long V, float eps, float alpha, const void* seed_ptr, float p_mix, float p_ffn,
                               void* stream) {
    VX_REQUIRE(ptrs && M >= 1 && M <= 4 && B > 0, "vx_pwa_post_fwd: bad args");
    VX_REQUIRE(vx_pwa_post_ok(C, Cv, R, V), "vx_pwa_post_fwd: unsupported shape C=%d Cv=%d R=%d V=%ld", C, Cv, R, V);
    VxPost p = {};
    if (int e = post_fill(p, ptrs, M, B, C, Cv, R, V, eps, alpha, seed_ptr, p_mix, p_ffn, false)) return e;
    const long blocks = (long)B * p.tiles_per_b;
    dim3 grid((unsigned)blocks, M);
    hipStream_t st = (hipStream_t)stream;
    if (post_tile_t(V) == 4) pf_launch<4, 4>(vx_pwa_post_fwd_k<4, 4>, p, grid, post_lds_floats(4, 4, C, Cv, R, false) * sizeof(float), st);
    else if (pf_nw(blocks * M) == 8) pf_launch<1, 8>(vx_pwa_post_fwd_k<1, 8>, p, grid, post_lds_floats(1, 8, C, Cv, R, false) * sizeof(float), st);
    else pf_launch<1, 4>(vx_pwa_post_fwd_k<1, 4>, p, grid, post_lds_floats(1, 4, C, Cv, R, false) * sizeof(float), st);
    VX_LAUNCH_CHECK("vx_pwa_post_fwd");
    return 0;
}

extern "C" int vx_pwa_post_bwd(const void* const* ptrs, int M, int B, int C, int Cv, int R, long V, float eps, float alpha, const void* seed_ptr, float p_mix, float p_ffn,
                               void* stream) {
    VX_REQUIRE(ptrs && M >= 1 && M <= 4 && B > 0, "vx_pwa_post_bwd: bad args");
    VX_REQUIRE(vx_pwa_post_ok(C, Cv, R, V), "vx_pwa_post_bwd: unsupported shape C=%d Cv=%d R=%d V=%ld", C, Cv, R, V);
    VxPost p = {};
    if (int e = post_fill(p, ptrs, M, B, C, Cv, R, V, eps, alpha, seed_ptr, p_mix, p_ffn, true)) return e;
    const long blocks = (long)B * p.tiles_per_b;
    dim3 grid((unsigned)blocks, M);
    hipStream_t st = (hipStream_t)stream;
    if (post_tile_t(V) == 4) pf_launch<4, 4>(vx_pwa_post_bwd_k<4, 4>, p, grid, post_lds_floats(4, 4, C, Cv, R, true) * sizeof(float), st);
    else if (pf_nw(blocks * M) == 8) pf_launch<1, 8>(vx_pwa_post_bwd_k<1, 8>, p, grid, post_lds_floats(1, 8, C, Cv, R, true) * sizeof(float), st);
    else pf_launch<1, 4>(vx_pwa_post_bwd_k<1, 4>, p, grid, post_lds_floats(1, 4, C, Cv, R, true) * sizeof(float), st);
    VX_LAUNCH_CHECK("vx_pwa_post_bwd");
    return 0;
}

// =====================================================================================================================
// channel stage of the JLC block at the levels mlp.hip has no instance for (C = 64 / 128: the 8^3 / 4^3 grids of the headline network)
//     out = o + Drop(W2 GELU(W1 IN(o) + b1) + b2)                                  (reference conv_blocks.py:60-66,74)
// IN = InstanceNorm over the volume: its statistics come from the producer's per-(b, c) partial sums (jlc.hip vx_jlc_mid_fwd), folded by every block for
// its sample as vx_mlp_in_stats does.  Same tile-GEMM chain as the PWA "post" kernels (T = 1 tiles of 16 voxels), same contract as vx_mlp_fwd / vx_mlp_bwd
// with NORM = 0: the backward returns dn (gradient at the normalised input) and per-tile partial sums (sum dn, sum dn nhat) per (b, c), which
// vx_jlc_mid_bwd folds into the InstanceNorm backward.  The weight gradients are two jobs of the grouped launch (pointwise.hip vx_pw_wgrad_group) over
// the operands this kernel leaves in scratch: dW2 = dz h^T, dW1 = da nhat^T.
// =====================================================================================================================
struct VxInMlp {
    const float *o, *w1, *b1, *w2, *b2, *dout;
    const double* part;          // forward: [B*C][nparts][2] (sum, sumsq) or null (stats given)
    float* stats;                // (B*C, 2) mean, rstd: written by the forward when part != null, read otherwise
    float *out, *dn, *part_dn;   // part_dn: [B*C][tiles_per_b][2]
    float *sc_n, *sc_h, *sc_da, *sc_dz;
    int C, R, nparts, tiles_per_b;
    long V;
    float eps, p;
    unsigned long long site;
    const void* seed_ptr;
};
namespace {
template <int NW>
__device__ __forceinline__ void inmlp_stats(const VxInMlp& p, int b, float* __restrict__ mu, float* __restrict__ rs, bool writer) {
    for (int c = threadIdx.x; c < p.C; c += 64 * NW) {
        const long bc = (long)b * p.C + c;
        float mean, rstd;
        if (p.part != nullptr) {
            double s = 0.0, q = 0.0;
            const double* pp = p.part + bc * p.nparts * 2;
            for (int i = 0; i < p.nparts; ++i) { s += pp[2 * i]; q += pp[2 * i + 1]; }
            const double m = s / (double)p.V;
            double var = q / (double)p.V - m * m;
            var = var < 0.0 ? 0.0 : var;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            if (writer) { p.stats[2 * bc] = mean; p.stats[2 * bc + 1] = rstd; }
        } else {
            mean = p.stats[2 * bc];
            rstd = p.stats[2 * bc + 1];
        }
        mu[c] = mean;
        rs[c] = rstd;
    }
}
}  // namespace

template <int NW>
__global__ void __launch_bounds__(64 * NW) vx_inmlp_fwd_k(VxInMlp p) {
    constexpr int T = 1, NT = 16, S = Geo<1>::S, NP = 64 * NW / NT;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const int C = p.C, R = p.R;
    float* __restrict__ O = vx_pf_lds;               // [C][S]
    float* __restrict__ N = O + C * S;               // [C][S]
    float* __restrict__ Hh = N + C * S;              // [R][S]
    float* __restrict__ mu = Hh + R * S, *rs = mu + C, *b2 = rs + C, *b1 = b2 + C;      // C | C | C | R
    const int b = blockIdx.x / p.tiles_per_b, tile = blockIdx.x % p.tiles_per_b;
    const long V = p.V, v0 = (long)tile * NT;
    const VxDropCtx d2 = drop_ctx_dev(p.seed_ptr, p.site, p.p);
    inmlp_stats<NW>(p, b, mu, rs, tile == 0);
    stage_vec<NW>(b2, p.b2, C, 0.0f);
    stage_vec<NW>(b1, p.b1, R, 0.0f);
    load_tile<T, NW>(O, p.o + (long)b * C * V, C, V, v0);
    __syncthreads();
    {
        const int col = threadIdx.x % NT, part = threadIdx.x / NT;
        for (int c = part; c < C; c += NP) N[c * S + col] = (O[c * S + col] - mu[c]) * rs[c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long v = v0 + r;
    const bool vlive = v < V;
    tile_gemm1<T, NW, false>(p.w1, C, C, 1, R, N, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            Hh[j * S + r] = vlive ? vx_gelu_fast(acc[0][reg] + b1[j]) : 0.0f;
        }
    });
    __syncthreads();
    tile_gemm1<T, NW, false>(p.w2, R, R, 1, C, Hh, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * q + reg;
            if (vlive) {
                const long idx = ((long)b * C + m) * V + v;
                p.out[idx] = fmaf(acc[0][reg] + b2[m], vx_drop1(d2, (uint64_t)idx), O[m * S + r]);
            }
        }
    });
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) vx_inmlp_bwd_k(VxInMlp p) {
    constexpr int T = 1, NT = 16, S = Geo<1>::S, NP = 64 * NW / NT;
    extern __shared__ __attribute__((aligned(16))) float vx_pf_lds[];
    const int C = p.C, R = p.R;
    float* __restrict__ DZ = vx_pf_lds;              // [C][S]  dout * mask, later dn
    float* __restrict__ N = DZ + C * S;              // [C][S]  nhat
    float* __restrict__ A = N + C * S;               // [R][S]  pre-activation a, then da
    float* __restrict__ mu = A + R * S, *rs = mu + C, *b1 = rs + C;      // C | C | R
    const int b = blockIdx.x / p.tiles_per_b, tile = blockIdx.x % p.tiles_per_b;
    const long V = p.V, v0 = (long)tile * NT;
    const VxDropCtx d2 = drop_ctx_dev(p.seed_ptr, p.site, p.p);
    const int col = threadIdx.x % NT, part = threadIdx.x / NT;
    const bool clive = v0 + col < V;
    inmlp_stats<NW>(p, b, mu, rs, false);
    stage_vec<NW>(b1, p.b1, R, 0.0f);
    load_tile<T, NW>(N, p.o + (long)b * C * V, C, V, v0);
    load_tile<T, NW>(DZ, p.dout + (long)b * C * V, C, V, v0);
    __syncthreads();
    for (int c = part; c < C; c += NP) {
        const long idx = ((long)b * C + c) * V + v0 + col;
        const float g = clive ? DZ[c * S + col] * vx_drop1(d2, (uint64_t)idx) : 0.0f;
        const float nh = clive ? (N[c * S + col] - mu[c]) * rs[c] : 0.0f;
        DZ[c * S + col] = g;
        N[c * S + col] = nh;
        if (clive) { p.sc_dz[idx] = g; p.sc_n[idx] = nh; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long v = v0 + r;
    const bool vlive = v < V;
    // a = W1 nhat + b1 ; h = gelu(a) -> global (dW2's operand)
    tile_gemm1<T, NW, false>(p.w1, C, C, 1, R, N, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            const float a = acc[0][reg] + b1[j];
            A[j * S + r] = a;
            if (vlive) p.sc_h[((long)b * R + j) * V + v] = vx_gelu_fast(a);
        }
    });
    __syncthreads();
    // da = (W2^T dz) * gelu'(a)
    tile_gemm1<T, NW, true>(p.w2, C, 1, R, R, DZ, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * mt + 4 * q + reg;
            const float da = vlive ? acc[0][reg] * vx_gelu_grad_fast(A[j * S + r]) : 0.0f;
            A[j * S + r] = da;
            if (vlive) p.sc_da[((long)b * R + j) * V + v] = da;
        }
    });
    __syncthreads();
    // dn = W1^T da -> global, and this tile's (sum dn, sum dn nhat) per channel: the 16 voxels of an accumulator register are the 16 lanes of a q group
    tile_gemm1<T, NW, true>(p.w1, R, 1, C, C, A, [&](int mt, vx_f32x4 (&acc)[T]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = 16 * mt + 4 * q + reg;
            const float dn = vlive ? acc[0][reg] : 0.0f;
            if (vlive) p.dn[((long)b * C + m) * V + v] = dn;
            float s1 = dn, s2 = dn * N[m * S + r];
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) { s1 += __shfl_xor(s1, o_, 64); s2 += __shfl_xor(s2, o_, 64); }
            if (r == 0) {
                float* __restrict__ pp = p.part_dn + (((long)b * C + m) * p.tiles_per_b + tile) * 2;
                pp[0] = s1; pp[1] = s2;
            }
        }
    });
}

extern "C" int vx_inmlp_ok(int C, int R, long V) {
    if (C % 16 || R % 16 || C < 16 || R < 16 || C > 256 || V < 1) return 0;
    const size_t lds = ((size_t)(2 * C + R) * 20 + 3 * C + R) * sizeof(float);
    return lds <= pf_lds_budget() ? 1 : 0;
}
extern "C" int vx_inmlp_tiles(long V) { return vx_cdiv(V, 16); }      // partial-sum rows per (b, c) of vx_inmlp_bwd (batch-size independent)

extern "C" int vx_inmlp_fwd(const float* o, const double* part, int nparts, float* stats, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                            int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site, float p_drop, void* stream) {
    VX_REQUIRE(o && stats && w1 && b1 && w2 && b2 && out && B > 0 && (part == nullptr || nparts > 0), "vx_inmlp_fwd: bad args");
    VX_REQUIRE(vx_inmlp_ok(C, R, V), "vx_inmlp_fwd: unsupported shape C=%d R=%d V=%ld", C, R, V);
    VX_REQUIRE(o != out, "vx_inmlp_fwd: in-place is not supported");
    VxInMlp p = {};
    p.o = o; p.part = part; p.nparts = nparts; p.stats = stats; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.out = out;
    p.C = C; p.R = R; p.V = V; p.eps = eps; p.p = p_drop; p.site = site; p.seed_ptr = seed_ptr; p.tiles_per_b = vx_cdiv(V, 16);
    const long blocks = (long)B * p.tiles_per_b;
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = ((size_t)(2 * C + R) * 20 + 3 * C + R) * sizeof(float);
    if (pf_nw(blocks) == 8) pf_launch<1, 8>(vx_inmlp_fwd_k<8>, p, dim3((unsigned)blocks), shm, st);
    else pf_launch<1, 4>(vx_inmlp_fwd_k<4>, p, dim3((unsigned)blocks), shm, st);
    VX_LAUNCH_CHECK("vx_inmlp_fwd");
    return 0;
}
/* dn (B, C, V), part_dn (B*C, vx_inmlp_tiles(V), 2), scratch sc_n / sc_dz (B, C, V), sc_h / sc_da (B, R, V) */
extern "C" int vx_inmlp_bwd(const float* o, const float* stats, const float* w1, const float* b1, const float* w2, const float* dout, float* dn, float* part_dn,
                            float* sc_n, float* sc_h, float* sc_da, float* sc_dz, int B, int C, int R, long V, const void* seed_ptr, unsigned long long site, float p_drop,
                            void* stream) {
    VX_REQUIRE(o && stats && w1 && b1 && w2 && dout && dn && part_dn && sc_n && sc_h && sc_da && sc_dz && B > 0, "vx_inmlp_bwd: bad args");
    VX_REQUIRE(vx_inmlp_ok(C, R, V), "vx_inmlp_bwd: unsupported shape C=%d R=%d V=%ld", C, R, V);
    VxInMlp p = {};
    p.o = o; p.stats = const_cast<float*>(stats); p.w1 = w1; p.b1 = b1; p.w2 = w2; p.dout = dout; p.dn = dn; p.part_dn = part_dn;
    p.sc_n = sc_n; p.sc_h = sc_h; p.sc_da = sc_da; p.sc_dz = sc_dz;
    p.C = C; p.R = R; p.V = V; p.p = p_drop; p.site = site; p.seed_ptr = seed_ptr; p.tiles_per_b = vx_cdiv(V, 16);
    const long blocks = (long)B * p.tiles_per_b;
    hipStream_t st = (hipStream_t)stream;
    const size_t shm = ((size_t)(2 * C + R) * 20 + 2 * C + R) * sizeof(float);
    if (pf_nw(blocks) == 8) pf_launch<1, 8>(vx_inmlp_bwd_k<8>, p, dim3((unsigned)blocks), shm, st);
    else pf_launch<1, 4>(vx_inmlp_bwd_k<4>, p, dim3((unsigned)blocks), shm, st);
    VX_LAUNCH_CHECK("vx_inmlp_bwd");
    return 0;
}
