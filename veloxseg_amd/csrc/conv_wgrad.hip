// Tiled weight-gradient kernel for spatial convolutions (any cubic K, stride, groups, optional pixel-shuffled dy).
//
//   dw[co, ci, tap] += sum_{b, q} dy[b, co, q] * x[b, g*Cin_g + ci, q*S - P + tap]
//
// Mapping (MI355X): one 256-thread block = (batch b, group g, COT output channels, a run of output tiles).
//   * the x halo tile of the block's output tile lives in LDS ([Cin_g][HD][HH][HW]);
//   * a thread owns NP (ci, tap) pairs x COT output channels in registers (NP*COT accumulators);
//   * dy[b, co, q] does not depend on the thread: it is read through the SCALAR path (s_load) and enters the
//     v_fmac as an SGPR operand -> 1 LDS read feeds COT FMAs, no vector load of dy at all;
//   * partial sums are flushed with one float atomic per (weight, block) after the block's last tile.
// Replaces the weight-gradient half of aten::convolution_backward for conv_blocks.py:10-17,51-58 and
// Decoder.py:73-76,150-153 (reference), and the ConvTranspose3d k2s2 weight gradient via the adjoint view.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"
#include <type_traits>

struct VxWg {
    int B, Cin, Di, Hi, Wi, Cout, Do, Ho, Wo, K, S, P, G, C1, ps;
    int TD, TH, TW;            // output tile
    int HD, HH, HW;            // halo tile = (T-1)*S + K
    int nTd, nTh, nTw;         // tiles per axis
    int tiles_per_block;
    int st_hw, st_hh, st_hd, st_c;   // 256 = ((st_c*HD + st_hd)*HH + st_hh)*HW + st_hw: per-iteration advance of the halo staging loop
    float* part;               // optional partial-sum workspace [slices][Cout * Cin_g * K^3]; nullptr -> float atomics into dw
};

__device__ __forceinline__ long vx_wg_dy_index(const VxWg& p, int b, int co, int d, int h, int w) {
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

template <int KT, int COT, int NP, int DYMODE>   // DYMODE 0: scalar dy loads, 1: 16-B loads along W (ps=1, Wo%4==0), 2: 16-B loads along co (ps=4)
__global__ void __launch_bounds__(256) vx_wgrad_tiled_k(const float* __restrict__ x, const float* __restrict__ x2,
                                                        const float* __restrict__ dy, float* __restrict__ dw, VxWg p) {
    extern __shared__ __attribute__((aligned(16))) float vx_halo[];
    const int K = KT > 0 ? KT : p.K;
    const int K3 = K * K * K;
    const int Cin_g = p.Cin / p.G, Cout_g = p.Cout / p.G;
    const int chunks_per_group = Cout_g / COT;
    const int g = blockIdx.y / chunks_per_group;
    const int co0 = g * Cout_g + (blockIdx.y % chunks_per_group) * COT;
    const int b = blockIdx.z;
    const int npairs = Cin_g * K3;
    const int plane = p.HD * p.HH * p.HW;
    const long Vi = (long)p.Di * p.Hi * p.Wi;
    const int ntiles = p.nTd * p.nTh * p.nTw;
    const int t_begin = blockIdx.x * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, ntiles);

    const int st_hw0 = threadIdx.x % p.HW, st_t1 = threadIdx.x / p.HW;
    const int st_hh0 = st_t1 % p.HH, st_t2 = st_t1 / p.HH;
    const int st_hd0 = st_t2 % p.HD, st_c0 = st_t2 / p.HD;
    for (int pass0 = 0; pass0 < npairs; pass0 += 256 * NP) {
        float acc[NP][COT];
        int loff[NP];
        bool pok[NP];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const int pi = pass0 + n * 256 + threadIdx.x;
            pok[n] = pi < npairs;
            const int pc = pok[n] ? pi : 0;
            const int tap = pc % K3, ci = pc / K3;
            const int kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
            loff[n] = ci * plane + (kd * p.HH + kh) * p.HW + kw;
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[n][j] = 0.0f;
        }
        for (int t = t_begin; t < t_end; ++t) {
            const int tw = t % p.nTw, th = (t / p.nTw) % p.nTh, td = t / (p.nTw * p.nTh);
            const int od0 = td * p.TD, oh0 = th * p.TH, ow0 = tw * p.TW;
            const int id0 = od0 * p.S - p.P, ih0 = oh0 * p.S - p.P, iw0 = ow0 * p.S - p.P;
            __syncthreads();
            {   // element e = ((ci*HD + hd)*HH + hh)*HW + hw for e = tid, tid+256, ...: (hw, hh, hd, ci) advance by the host-decomposed
                // stride of 256, so the loop body has no integer division (4 per element used to cost as much as the FMA phase)
                // 4 elements per iteration with UNCONDITIONAL loads (clamped address, value selected afterwards): with a branch around each load
                // the loop ran one memory latency per element
                int hw = st_hw0, hh = st_hh0, hd = st_hd0, ci = st_c0;
                const int total = Cin_g * plane;
                for (int e = threadIdx.x; e < total; e += 256 * 4) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
                        const bool ok = (e + u * 256 < total) && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                        const int c = ok ? g * Cin_g + ci : g * Cin_g;
                        const float* src = (c < p.C1) ? x + ((long)b * p.C1 + c) * Vi : x2 + ((long)b * (p.Cin - p.C1) + (c - p.C1)) * Vi;
                        const float t_ = src[ok ? ((long)id * p.Hi + ih) * p.Wi + iw : 0];
                        v[u] = ok ? t_ : 0.0f;
                        hw += p.st_hw; if (hw >= p.HW) { hw -= p.HW; ++hh; }
                        hh += p.st_hh; if (hh >= p.HH) { hh -= p.HH; ++hd; }
                        hd += p.st_hd; if (hd >= p.HD) { hd -= p.HD; ++ci; }
                        ci += p.st_c;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e + u * 256 < total) vx_halo[e + u * 256] = v[u];
                }
            }
            __syncthreads();
            const int nd = min(p.TD, p.Do - od0), nh = min(p.TH, p.Ho - oh0), nw = min(p.TW, p.Wo - ow0);
            for (int qd = 0; qd < nd; ++qd)
                for (int qh = 0; qh < nh; ++qh) {
                    const int rowoff = (qd * p.S * p.HH + qh * p.S) * p.HW;
                    int qw = 0;
                    if (DYMODE == 1) {
                        // ps == 1: one row pointer per output channel, hoisted out of the voxel loop (the generic index function re-derived it -- with
                        // a branch on ps -- for every channel and 4-voxel group), and 8 voxels per iteration so that the 2*COT scalar loads and the
                        // 8*NP LDS reads of an iteration are all in flight before its 8*NP*COT FMAs
                        const float* dyr[COT];
#pragma unroll
                        for (int j = 0; j < COT; ++j)
                            dyr[j] = dy + ((((long)b * p.Cout + co0 + j) * p.Do + od0 + qd) * p.Ho + oh0 + qh) * (long)p.Wo + ow0;
                        for (; qw + 8 <= nw; qw += 8) {
                            float xv[NP][8];
#pragma unroll
                            for (int n = 0; n < NP; ++n)
#pragma unroll
                                for (int u = 0; u < 8; ++u) xv[n][u] = vx_halo[loff[n] + rowoff + (qw + u) * p.S];
                            float4 da[COT], dbv[COT];
#pragma unroll
                            for (int j = 0; j < COT; ++j) {
                                da[j] = *reinterpret_cast<const float4*>(dyr[j] + qw);
                                dbv[j] = *reinterpret_cast<const float4*>(dyr[j] + qw + 4);
                            }
#pragma unroll
                            for (int j = 0; j < COT; ++j) {
                                const float dv[8] = {da[j].x, da[j].y, da[j].z, da[j].w, dbv[j].x, dbv[j].y, dbv[j].z, dbv[j].w};
#pragma unroll
                                for (int u = 0; u < 8; ++u)
#pragma unroll
                                    for (int n = 0; n < NP; ++n) acc[n][j] = fmaf(dv[u], xv[n][u], acc[n][j]);
                            }
                        }
                    }
                    for (; qw + 4 <= nw; qw += 4) {
                        float xv[NP][4];
#pragma unroll
                        for (int n = 0; n < NP; ++n)
#pragma unroll
                            for (int u = 0; u < 4; ++u) xv[n][u] = vx_halo[loff[n] + rowoff + (qw + u) * p.S];
                        if (DYMODE == 1) {
#pragma unroll
                            for (int j = 0; j < COT; ++j) {
                                const float4 d4 = *reinterpret_cast<const float4*>(dy + vx_wg_dy_index(p, b, co0 + j, od0 + qd, oh0 + qh, ow0 + qw));   // wave-uniform -> s_load_dwordx4
                                const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int n = 0; n < NP; ++n) acc[n][j] = fmaf(dv[u], xv[n][u], acc[n][j]);
                            }
                        } else if (DYMODE == 2 && COT >= 4) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
#pragma unroll
                                for (int j4 = 0; j4 < COT; j4 += 4) {
                                    const float4 d4 = *reinterpret_cast<const float4*>(dy + vx_wg_dy_index(p, b, co0 + j4, od0 + qd, oh0 + qh, ow0 + qw + u));   // 4 consecutive co
                                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                                        for (int n = 0; n < NP; ++n) acc[n][(j4 + jj) < COT ? (j4 + jj) : 0] = fmaf(dv[jj], xv[n][u], acc[n][(j4 + jj) < COT ? (j4 + jj) : 0]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < COT; ++j) {
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const float dv = dy[vx_wg_dy_index(p, b, co0 + j, od0 + qd, oh0 + qh, ow0 + qw + u)];   // wave-uniform -> s_load
#pragma unroll
                                    for (int n = 0; n < NP; ++n) acc[n][j] = fmaf(dv, xv[n][u], acc[n][j]);
                                }
                            }
                        }
                    }
                    for (; qw < nw; ++qw) {
                        float xv[NP];
#pragma unroll
                        for (int n = 0; n < NP; ++n) xv[n] = vx_halo[loff[n] + rowoff + qw * p.S];
#pragma unroll
                        for (int j = 0; j < COT; ++j) {
                            const float dv = dy[vx_wg_dy_index(p, b, co0 + j, od0 + qd, oh0 + qh, ow0 + qw)];
#pragma unroll
                            for (int n = 0; n < NP; ++n) acc[n][j] = fmaf(dv, xv[n], acc[n][j]);
                        }
                    }
                }
        }
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const int pi = pass0 + n * 256 + threadIdx.x;
            if (pi < npairs) {
                if (p.part) {                  // every (slice, weight) is written by exactly one block: plain coalesced stores
                    float* __restrict__ pd = p.part + ((long)blockIdx.z * gridDim.x + blockIdx.x) * ((long)p.Cout * npairs);
#pragma unroll
                    for (int j = 0; j < COT; ++j) pd[(long)(co0 + j) * npairs + pi] = acc[n][j];
                } else {
#pragma unroll
                    for (int j = 0; j < COT; ++j) atomicAdd(dw + (long)(co0 + j) * npairs + pi, acc[n][j]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row-sliding weight gradient for the JLC grouped convolutions (conv_blocks.py:51-58: k = 3 / 5, stride 1, padding k/2, Cin/G == Cout/G == CG).
// vx_wgrad_tiled_k gives a thread (ci, tap) pairs: every LDS read feeds Cout/G = 4 or 8 FMAs, and the LDS pipe -- not the VALU -- sets its
// 12-15 % of the fp32 peak.  Here a thread owns (ci, kd, kh) and ALL K taps of the kw row for all CG output channels (K * CG accumulators): for U
// consecutive output voxels of a row it reads the U + K - 1 input values once (ds_read_b128) and slides them across the kw taps in registers --
// U * K * CG FMAs per U + K - 1 LDS dwords (160 per 12 at k = 5, CG = 4) -- with dy entering as scalar (SGPR) operands as before.
// A block = (b, group, run of output tiles); its CG * K * K owner threads fill WPS waves, and the block's 4 / WPS sub-blocks take alternate output
// rows of the same LDS halo tile; their sums meet in LDS before one float atomic per (weight, block) (or one store into the partial-sum slice).
// ---------------------------------------------------------------------------------------------------------------------------
typedef float vx_wf2 __attribute__((ext_vector_type(2)));
template <int K, int CG, int U>
__global__ void __launch_bounds__(256) vx_wgrad_rows_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, VxWg p) {
    constexpr int T = CG * K * K;
    constexpr int WPS = (T + 63) / 64;                 // waves per sub-block
    constexpr int NS = 4 / WPS;                        // sub-blocks per block (alternate output rows)
    constexpr int NX = (U + K - 1 + 3) & ~3;           // input values per chunk, whole float4s
    static_assert(WPS == 1 || WPS == 2 || WPS == 4, "owner threads must tile the block's waves");
    extern __shared__ __attribute__((aligned(16))) float vx_halo[];      // [CG][HD][HH][HW], HW a multiple of 4
    constexpr int K3 = K * K * K;
    const int g = blockIdx.y, b = blockIdx.z;
    const int co0 = g * CG;
    const int plane = p.HD * p.HH * p.HW;
    const long Vi = (long)p.Di * p.Hi * p.Wi;
    const int ntiles = p.nTd * p.nTh * p.nTw;
    const int t_begin = blockIdx.x * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, ntiles);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sub = wave / WPS;
    const int r = threadIdx.x - sub * (WPS * 64);
    const bool owner = r < T;
    const int rc = owner ? r : 0;
    const int ci = rc / (K * K), kd = (rc / K) % K, kh = rc % K;
    const int toff = ci * plane + (kd * p.HH + kh) * p.HW;
    // accumulators as float2 (x: even voxels of a chunk, y: odd voxels): the FMAs are v_pk_fma_f32 with an SGPR pair (two consecutive dy values) as one
    // operand and an aligned register pair of the input row as the other -- no per-FMA register shuffling
    vx_wf2 acc2[K][CG];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < CG; ++j) acc2[k][j] = (vx_wf2){0.0f, 0.0f};
    const int st_hw0 = threadIdx.x % p.HW, st_t1 = threadIdx.x / p.HW;
    const int st_hh0 = st_t1 % p.HH, st_t2 = st_t1 / p.HH;
    const int st_hd0 = st_t2 % p.HD, st_c0 = st_t2 / p.HD;
    for (int t = t_begin; t < t_end; ++t) {
        const int tw = t % p.nTw, th = (t / p.nTw) % p.nTh, td = t / (p.nTw * p.nTh);
        const int od0 = td * p.TD, oh0 = th * p.TH, ow0 = tw * p.TW;
        const int id0 = od0 - p.P, ih0 = oh0 - p.P, iw0 = ow0 - p.P;
        __syncthreads();
        {   // halo staging as in vx_wgrad_tiled_k: (hw, hh, hd, ci) advance by the host-decomposed stride of 256, unconditional (clamped) loads, 4 in flight
            int hw = st_hw0, hh = st_hh0, hd = st_hd0, cc = st_c0;
            const int total = CG * plane;
            for (int e = threadIdx.x; e < total; e += 256 * 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
                    const bool ok = (e + u * 256 < total) && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                    const float* src = x + ((long)b * p.Cin + co0 + (ok ? cc : 0)) * Vi;
                    const float t_ = src[ok ? ((long)id * p.Hi + ih) * p.Wi + iw : 0];
                    v[u] = ok ? t_ : 0.0f;
                    hw += p.st_hw; if (hw >= p.HW) { hw -= p.HW; ++hh; }
                    hh += p.st_hh; if (hh >= p.HH) { hh -= p.HH; ++hd; }
                    hd += p.st_hd; if (hd >= p.HD) { hd -= p.HD; ++cc; }
                    cc += p.st_c;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (e + u * 256 < total) vx_halo[e + u * 256] = v[u];
            }
        }
        __syncthreads();
        const int nd = min(p.TD, p.Do - od0), nh = min(p.TH, p.Ho - oh0), nw = min(p.TW, p.Wo - ow0);      // nw % U == 0 (launcher)
        for (int rr = sub; rr < nd * nh; rr += NS) {
            const int qd = rr / nh, qh = rr - qd * nh;
            const float* __restrict__ xrow = vx_halo + toff + (qd * p.HH + qh) * p.HW;
            const float* __restrict__ dyr = dy + ((((long)b * p.Cout + co0) * p.Do + od0 + qd) * p.Ho + oh0 + qh) * (long)p.Wo + ow0;      // wave-uniform
            const long cstride = (long)p.Do * p.Ho * p.Wo;
            for (int qw = 0; qw < nw; qw += U) {
                float xr[NX];
#pragma unroll
                for (int i = 0; i < NX; i += 4) {
                    const float4 v4 = *reinterpret_cast<const float4*>(xrow + qw + i);
                    xr[i] = v4.x; xr[i + 1] = v4.y; xr[i + 2] = v4.z; xr[i + 3] = v4.w;
                }
                vx_wf2 xe[NX / 2], xo[NX / 2];         // (x[2i], x[2i+1]) and (x[2i+1], x[2i+2])
#pragma unroll
                for (int i = 0; i < NX / 2; ++i) {
                    xe[i] = (vx_wf2){xr[2 * i], xr[2 * i + 1]};
                    xo[i] = (vx_wf2){xr[2 * i + 1], xr[2 * i + 2 < NX ? 2 * i + 2 : 2 * i + 1]};
                }
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    vx_wf2 d2[U / 2];
#pragma unroll
                    for (int u = 0; u < U; u += 4) {
                        const float4 d4 = *reinterpret_cast<const float4*>(dyr + j * cstride + qw + u);      // s_load_dwordx4
                        d2[u / 2] = (vx_wf2){d4.x, d4.y};
                        d2[u / 2 + 1] = (vx_wf2){d4.z, d4.w};
                    }
#pragma unroll
                    for (int u2 = 0; u2 < U / 2; ++u2)
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const int idx = 2 * u2 + k;
                            acc2[k][j] = __builtin_elementwise_fma(d2[u2], (idx & 1) ? xo[idx / 2] : xe[idx / 2], acc2[k][j]);
                        }
                }
            }
        }
    }
    // Flush: the sub-blocks' sums meet in an LDS image of the group's weights ([CG co][CG * K^3]), then ALL threads add consecutive addresses to dw
    // (a lane's own K * CG sums are K floats apart per lane: flushed from registers, a wave's atomic touched 10-20 cache lines instead of 2, and
    // the flush was 2/3 of the kernel)
    constexpr int NPAIRS = CG * K3;
    const int pi0 = ci * K3 + (kd * K + kh) * K;
    float* __restrict__ img = vx_halo;                 // [CG][NPAIRS]
    for (int s2 = 0; s2 < NS; ++s2) {
        __syncthreads();
        if (sub == s2 && owner) {
#pragma unroll
            for (int j = 0; j < CG; ++j)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float v = acc2[k][j].x + acc2[k][j].y;
                    img[j * NPAIRS + pi0 + k] = s2 == 0 ? v : img[j * NPAIRS + pi0 + k] + v;
                }
        }
    }
    __syncthreads();
    if (p.part) {
        float* __restrict__ pd = p.part + ((long)blockIdx.z * gridDim.x + blockIdx.x) * ((long)p.Cout * NPAIRS) + (long)co0 * NPAIRS;
        for (int i = threadIdx.x; i < CG * NPAIRS; i += 256) pd[i] = img[i];
    } else {
        float* __restrict__ dst = dw + (long)co0 * NPAIRS;
        for (int i = threadIdx.x; i < CG * NPAIRS; i += 256) atomicAdd(dst + i, img[i]);
    }
}

// db[co] += sum_{b,q} dy[b,co,q]  (dy possibly pixel-shuffled); grid (Cout, chunks)
__global__ void __launch_bounds__(256) vx_bias_grad_k(const float* __restrict__ dy, float* __restrict__ db, VxWg p) {
    const int co = blockIdx.x;
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    const long n = (long)p.B * Vo;
    float s = 0.0f;
    for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n; i += (long)gridDim.y * 256) {
        const int b = (int)(i / Vo);
        const long q = i % Vo;
        const int w = (int)(q % p.Wo), h = (int)((q / p.Wo) % p.Ho), d = (int)(q / ((long)p.Wo * p.Ho));
        s += dy[vx_wg_dy_index(p, b, co, d, h, w)];
    }
    __shared__ float sm[4];
    s = vx_block_sum_256(s, sm);
    if (threadIdx.x == 0) atomicAdd(db + co, s);
}

// dw[i] += sum over slices of part[s][i]; grid (cdiv(n,256), slice chunks): coalesced reads, one float atomic per (weight, chunk)
__global__ void __launch_bounds__(256) vx_wg_reduce_k(const float* __restrict__ part, float* __restrict__ dw, long n, int slices) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int per = (slices + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = min(s0 + per, slices);
    float a0 = 0.0f, a1 = 0.0f;
    int s_ = s0;
    for (; s_ + 1 < s1; s_ += 2) { a0 += part[(long)s_ * n + i]; a1 += part[(long)(s_ + 1) * n + i]; }
    if (s_ < s1) a0 += part[(long)s_ * n + i];
    if (s0 < s1) atomicAdd(dw + i, a0 + a1);
}

template <int N> using vx_ic2 = std::integral_constant<int, N>;
static int vx_wg_rows_enabled = 1;
extern "C" int vx_wgrad_set_rows(int on) { vx_wg_rows_enabled = on ? 1 : 0; return 0; }      // A/B switch for tests and tools (vx_wgrad_rows_k vs vx_wgrad_tiled_k)
#ifndef VX_WG_MIN_BLOCKS_SMALL
#define VX_WG_MIN_BLOCKS_SMALL 384
#endif
#ifndef VX_WG_MIN_BLOCKS
#define VX_WG_MIN_BLOCKS 2048   // the inner loop is latency-bound (scalar dy loads + LDS reads feed few FMAs per thread): 8 waves per SIMD beat big tiles
#endif

// ws == nullptr: float atomics straight into dw.  ws != nullptr (>= *need floats): blocks store partial sums, vx_wg_reduce_k folds them.
// query: only compute *need (0 when the atomic path is the better one: few blocks per weight).
static int vx_wg_run(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db, float* ws, long ws_floats, long* need, bool query,
                     int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream) {
    VX_REQUIRE(x && dy && dw && B > 0 && Cin > 0 && Cout > 0 && K > 0 && S > 0 && G > 0 && ps > 0, "vx_conv3d_bwd_weight_tiled: bad args");
    VX_REQUIRE(Cin % G == 0 && Cout % G == 0, "vx_conv3d_bwd_weight_tiled: channels not divisible by groups");
    VxWg p;
    p.B = B; p.Cin = Cin; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cout = Cout; p.K = K; p.S = S; p.P = P; p.G = G; p.ps = ps;
    p.C1 = (C1 <= 0 || C1 > Cin) ? Cin : C1;
    VX_REQUIRE(p.C1 == Cin || x2, "vx_conv3d_bwd_weight_tiled: x2 missing");
    p.Do = (Di + 2 * P - K) / S + 1; p.Ho = (Hi + 2 * P - K) / S + 1; p.Wo = (Wi + 2 * P - K) / S + 1;
    VX_REQUIRE(p.Do > 0 && p.Ho > 0 && p.Wo > 0, "vx_conv3d_bwd_weight_tiled: empty output");
    const int Cin_g = Cin / G, Cout_g = Cout / G;
    // output tile: a W-run of up to 32 voxels, grown in H then D until the halo of all Cin_g channels fills ~64 KB of LDS
    p.TW = p.Wo < 32 ? p.Wo : 32; p.TH = 1; p.TD = 1;
    auto halo_bytes = [&](int td, int th, int tw) {
        return (size_t)Cin_g * ((td - 1) * S + K) * ((th - 1) * S + K) * ((tw - 1) * S + K) * sizeof(float);
    };
    const size_t budget = 64 * 1024;
    while (halo_bytes(p.TD, p.TH, p.TW) > budget && p.TW > 4) p.TW = (p.TW + 1) / 2;
    VX_REQUIRE(halo_bytes(p.TD, p.TH, p.TW) <= 150 * 1024, "vx_conv3d_bwd_weight_tiled: halo tile does not fit LDS (Cin/G=%d K=%d)", Cin_g, K);
    for (;;) {
        bool grown = false;
        if (p.TH < p.Ho && p.TH < 8 && halo_bytes(p.TD, p.TH * 2, p.TW) <= budget) { p.TH *= 2; grown = true; }
        if (p.TD < p.Do && p.TD < 4 && halo_bytes(p.TD * 2, p.TH, p.TW) <= budget) { p.TD *= 2; grown = true; }
        if (!grown) break;
    }
    if (p.TH > p.Ho) p.TH = p.Ho;
    if (p.TD > p.Do) p.TD = p.Do;
    // JLC grouped convs (k 3 / 5, stride 1, Cin/G == Cout/G in {4, 8}) on rows of >= 16 voxels: the row-sliding kernel (vx_wgrad_rows_k)
    const bool rows_ok = vx_wg_rows_enabled && S == 1 && ps == 1 && (K == 3 || K == 5) && P == K / 2 && Cin_g == Cout_g && (Cin_g == 4 || Cin_g == 8) && p.C1 == Cin &&
                         G > 1 && p.Wo >= 16 && p.Wo % 4 == 0;
    if (rows_ok) {
        // its halo staging and its FMA phase cost about the same, so the tiles stay compact (halo / tile volume): halve the taller of H and 2*D until
        // ~1024 blocks exist (measured at 32^3: 4 x 4 x 32 tiles 44 us, 1 x 8 x 32 63 us; at 16^3: 2 x 4 x 16 34 us)
        auto nblk = [&]() { return (long)vx_cdiv(p.Do, p.TD) * vx_cdiv(p.Ho, p.TH) * vx_cdiv(p.Wo, p.TW) * G * B; };
        const long want = ((long)p.Do * p.Ho * p.Wo <= 4096) ? VX_WG_MIN_BLOCKS_SMALL : 1024;
        while (nblk() < want && (p.TD > 1 || p.TH > 1)) {
            if (p.TH >= 2 * p.TD && p.TH > 1) p.TH = (p.TH + 1) / 2;
            else if (p.TD > 1) p.TD = (p.TD + 1) / 2;
            else p.TH = (p.TH + 1) / 2;
        }
    } else
    {   // small volumes: prefer more, smaller tiles (>= ~512 blocks) over LDS-filling ones; the atomic flush stays cheap
        const int COT0 = (Cout_g % 8 == 0) ? 8 : (Cout_g % 4 == 0) ? 4 : (Cout_g % 2 == 0) ? 2 : 1;
        auto nblk = [&]() { return (long)vx_cdiv(p.Do, p.TD) * vx_cdiv(p.Ho, p.TH) * vx_cdiv(p.Wo, p.TW) * G * (Cout_g / COT0) * B; };
        // small volumes (<= 16^3): every extra tile re-stages a halo that is mostly border, so aim lower
        const long want = ((long)p.Do * p.Ho * p.Wo <= 4096) ? VX_WG_MIN_BLOCKS_SMALL : VX_WG_MIN_BLOCKS;
        while (nblk() < want && p.TD > 1) p.TD = (p.TD + 1) / 2;
        while (nblk() < want && p.TH > 1) p.TH = (p.TH + 1) / 2;
        while (nblk() < want && p.TW > 8) p.TW = (p.TW + 1) / 2;
    }
    p.HD = (p.TD - 1) * S + K; p.HH = (p.TH - 1) * S + K; p.HW = (p.TW - 1) * S + K;
    // voxels per chunk of the row-sliding kernel (0 = the pair kernel); its halo rows are whole float4s
    const int rows_u = rows_ok ? ((p.Wo % 8 == 0 && p.TW % 8 == 0) ? 8 : (p.TW % 4 == 0) ? 4 : 0) : 0;
    if (rows_u) p.HW = (p.HW + 3) & ~3;
    {
        int r_ = 256;
        p.st_hw = r_ % p.HW; r_ /= p.HW;
        p.st_hh = r_ % p.HH; r_ /= p.HH;
        p.st_hd = r_ % p.HD; r_ /= p.HD;
        p.st_c = r_;
    }
    p.nTd = vx_cdiv(p.Do, p.TD); p.nTh = vx_cdiv(p.Ho, p.TH); p.nTw = vx_cdiv(p.Wo, p.TW);
    const int ntiles = p.nTd * p.nTh * p.nTw;
    const int COT = rows_u ? Cout_g : (Cout_g % 8 == 0) ? 8 : (Cout_g % 4 == 0) ? 4 : (Cout_g % 2 == 0) ? 2 : 1;
    const int gy = G * (Cout_g / COT);
    // enough blocks to fill 256 CUs a few times over, but several tiles per block to amortise the atomic flush
    int tpb = (int)(((long)ntiles * gy * B + 2047) / 2048);
    if (tpb < 1) tpb = 1;
    if (tpb > 8) tpb = 8;
    p.tiles_per_block = tpb;
    const int npairs = Cin_g * K * K * K;
    const int NP = npairs > 512 ? 4 : (npairs > 256 ? 2 : 1);
    size_t shm = halo_bytes(p.TD, p.TH, p.TW);
    if (rows_u) {
        shm = (size_t)Cin_g * p.HD * p.HH * p.HW * sizeof(float);
        const size_t red = (size_t)Cin_g * Cin_g * K * K * K * sizeof(float);      // the flush image of the group's weights reuses the halo region
        if (shm < red) shm = red;
    }
    dim3 grid(vx_cdiv(ntiles, tpb), gy, B);
    hipStream_t st = (hipStream_t)stream;
    const long nw = (long)Cout * npairs;
    const int slices = (int)grid.x * B;
    // partial-sum path pays off when many blocks hit every weight (float atomics on one address serialise in L2: the 7^3 stride-4
    // DownConv issued 11 M atomics on 11 k addresses = 0.5 ms) and the workspace stays small
    const long want = (slices >= 16 && (long)slices * nw <= (64l << 20)) ? (long)slices * nw : 0;
    if (need) *need = want;
    if (query) return 0;
    p.part = (ws && want > 0 && ws_floats >= want) ? ws : nullptr;
    const int dymode = (ps == 1 && p.Wo % 4 == 0 && p.TW % 4 == 0) ? 1 : ((ps == 4 && COT >= 4) ? 2 : 0);
    auto launch = [&](auto kt, auto cot, auto np) {
        constexpr int KT_ = decltype(kt)::value, COT_ = decltype(cot)::value, NP_ = decltype(np)::value;
        if (dymode == 1) vx_wgrad_tiled_k<KT_, COT_, NP_, 1><<<grid, dim3(256), shm, st>>>(x, x2, dy, dw, p);
        else if (dymode == 2) vx_wgrad_tiled_k<KT_, COT_, NP_, 2><<<grid, dim3(256), shm, st>>>(x, x2, dy, dw, p);
        else vx_wgrad_tiled_k<KT_, COT_, NP_, 0><<<grid, dim3(256), shm, st>>>(x, x2, dy, dw, p);
    };
    auto with_np = [&](auto kt, auto cot) {
        if (NP == 4) launch(kt, cot, vx_ic2<4>{});
        else if (NP == 2) launch(kt, cot, vx_ic2<2>{});
        else launch(kt, cot, vx_ic2<1>{});
    };
    auto with_cot = [&](auto kt) {
        switch (COT) {
            case 8: with_np(kt, vx_ic2<8>{}); break;
            case 4: with_np(kt, vx_ic2<4>{}); break;
            case 2: with_np(kt, vx_ic2<2>{}); break;
            default: with_np(kt, vx_ic2<1>{}); break;
        }
    };
    if (rows_u) {
#define VX_ROWS(K_, CG_, U_) vx_wgrad_rows_k<K_, CG_, U_><<<grid, dim3(256), shm, st>>>(x, dy, dw, p)
        if (K == 5 && Cin_g == 4) { if (rows_u == 8) VX_ROWS(5, 4, 8); else VX_ROWS(5, 4, 4); }
        else if (K == 5) { if (rows_u == 8) VX_ROWS(5, 8, 8); else VX_ROWS(5, 8, 4); }
        else if (Cin_g == 4) { if (rows_u == 8) VX_ROWS(3, 4, 8); else VX_ROWS(3, 4, 4); }
        else { if (rows_u == 8) VX_ROWS(3, 8, 8); else VX_ROWS(3, 8, 4); }
#undef VX_ROWS
    } else
    switch (K) {
        case 3: with_cot(vx_ic2<3>{}); break;
        case 5: with_cot(vx_ic2<5>{}); break;
        default: with_cot(vx_ic2<0>{}); break;
    }
    if (p.part) {
        int sc = slices / 32;
        if (sc < 1) sc = 1;
        if (sc > 32) sc = 32;
        vx_wg_reduce_k<<<dim3(vx_cdiv(nw, 256), sc), dim3(256), 0, st>>>(p.part, dw, nw, slices);
    }
    if (db) {
        int chunks = vx_cdiv((long)B * p.Do * p.Ho * p.Wo, 256 * 16);
        if (chunks > 32) chunks = 32;
        vx_bias_grad_k<<<dim3(Cout, chunks), dim3(256), 0, st>>>(dy, db, p);
    }
    VX_LAUNCH_CHECK("vx_conv3d_bwd_weight_tiled");
    return 0;
}

extern "C" int vx_conv3d_bwd_weight_tiled(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                                          int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream) {
    return vx_wg_run(x, x2, C1, dy, dw, db, nullptr, 0, nullptr, false, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, ps, stream);
}

extern "C" int vx_conv3d_bwd_weight_ws_floats(int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps) {
    long need = 0;
    static const float dummy = 0.0f;
    const int rc = vx_wg_run(&dummy, &dummy, 0, &dummy, const_cast<float*>(&dummy), nullptr, nullptr, 0, &need, true, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, ps, nullptr);
    if (rc != 0) return rc;
    return need > 0x7fffffffl ? 0 : (int)need;
}

extern "C" int vx_conv3d_bwd_weight_tiled_ws(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                             int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, int G, int ps, void* stream) {
    return vx_wg_run(x, x2, C1, dy, dw, db, ws, ws_floats, nullptr, false, B, Cin, Di, Hi, Wi, Cout, K, S, P, G, ps, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Stem DownConv (Conv3d k7 s4 p3, 1..4 modalities -> 16 channels; reference Encoder.py conv-chain stem): weight gradient as an fp32-MFMA GEMM.
//   dW[co][n] = sum_{b,q} dy[b,co,q] * patch[b,q][n],   n = (ci, kd, kh, kw),  M = 16 co, N = CB*343 pairs (NT tiles of 16), K = B*Vout voxels.
// The tiled VALU kernel above needs 0.44 ms for these 2.9 GFLOP (4 % of the fp32 peak: 686 pairs on 1024 thread slots, one LDS read per 8 FMAs
// with stride-4 addresses).  Here a block stages the x halo of a 1 x 4 x 16 output tile ([CB][7][19][67] floats) in LDS; wave = one output
// row = 4 k-steps of 4 voxels; per k-step ONE A operand (dy, lane (co r, voxel q)) feeds NT MFMAs whose B operands are LDS gathers at
// per-lane pair offsets; the NT accumulator tiles stay in registers across the block's tiles; the four waves are then added in LDS (plain
// phased read-add-write) and stored as one partial slice; vx_wg_reduce_k folds the slices into dw.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float vx_wf4 __attribute__((ext_vector_type(4)));
#define VX_DW_HD 7
#define VX_DW_HH 19
#define VX_DW_HW 67
#define VX_DW_PLANE (VX_DW_HD * VX_DW_HH * VX_DW_HW)
template <int NT>
__global__ void __launch_bounds__(256, 2) vx_down_wgrad_mfma_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                            float* __restrict__ db, int B, int Cin, int Di, int Hi, int Wi, int Cout, int Do, int Ho,
                                                            int Wo, int CB, int tiles_per_block, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float vx_halo[];          // [CB][7][19][67]; reused as [16][NT*16] for the final reduction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * 16, cib = blockIdx.z * CB;
    const int npairs = CB * 343;
    // per-pair LDS offsets in an LDS table behind the halo (43 more live registers would push the kernel past 256 = one wave per SIMD)
    int* __restrict__ offtab = reinterpret_cast<int*>(vx_halo + ((CB * VX_DW_PLANE + 3) & ~3));
    for (int n = threadIdx.x; n < NT * 16; n += 256) {
        const int tap = n % 343, ci = n / 343;
        const int kw = tap % 7, kh = (tap / 7) % 7, kd = tap / 49;
        offtab[n] = n < npairs ? ci * VX_DW_PLANE + (kd * VX_DW_HH + kh) * VX_DW_HW + kw : -1;
    }
    vx_wf4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (vx_wf4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const int nTw = (Wo + 15) / 16, nTh = Ho / 4;
    const long Vi = (long)Di * Hi * Wi, Vo = (long)Do * Ho * Wo;
    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(t_begin + tiles_per_block, ntiles);
    for (int t = t_begin; t < t_end; ++t) {
        int tt = t;
        const int tw = tt % nTw; tt /= nTw;
        const int th = tt % nTh; tt /= nTh;
        const int d = tt % Do;
        const int b = tt / Do;
        const int h0 = th * 4, w0 = tw * 16;
        const int id0 = 4 * d - 3, ih0 = 4 * h0 - 3, iw0 = 4 * w0 - 3;
        __syncthreads();
        // halo staging, 8 elements per thread and iteration: the loads are unconditional (clamped address, value selected afterwards) so that all
        // eight are in flight together -- with a branch around each load the loop ran one memory latency per element (250 of the kernel's 400 us)
        {
            const float* __restrict__ xb = x + ((long)b * Cin + cib) * Vi;
            const int total = CB * VX_DW_PLANE;
            for (int e0 = threadIdx.x; e0 < total; e0 += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = min(e0 + u * 256, total - 1);
                    const int hw = e % VX_DW_HW, row = e / VX_DW_HW;
                    const int hh = row % VX_DW_HH, r2 = row / VX_DW_HH;
                    const int hd = r2 % VX_DW_HD, ci = r2 / VX_DW_HD;
                    const int id = id0 + hd, ih = ih0 + hh, iw = iw0 + hw;
                    const bool ok = (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
                    const long idx = ok ? (long)ci * Vi + ((long)id * Hi + ih) * Wi + iw : 0;
                    const float t_ = xb[idx];
                    v[u] = ok ? t_ : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 256 < total) vx_halo[e0 + u * 256] = v[u];
            }
        }
        __syncthreads();
        const float* __restrict__ dyr = dy + ((long)b * Cout + co0 + r) * Vo + ((long)d * Ho + h0 + wave) * Wo;
        float av[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {                       // (columns beyond Wo -- a partly idle last tile -- contribute zeros)
            const int wq = w0 + q + 4 * s4;
            const float t_ = dyr[min(wq, Wo - 1)];
            av[s4] = wq < Wo ? t_ : 0.0f;
        }
#pragma unroll 1
        for (int s4 = 0; s4 < 4; ++s4) {          // not unrolled: 4 x 43 operand prefetches would not fit 256 registers
            bsum += av[s4];
            const int base = (4 * wave) * VX_DW_HW + 4 * (4 * s4 + q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int o = offtab[nt * 16 + r];
                const float bv = o >= 0 ? vx_halo[o + base] : 0.0f;
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv, acc[nt], 0, 0, 0);
            }
        }
    }
    // block reduction: D row 4q+reg = co, col r = pair n of tile nt
    float* __restrict__ red = vx_halo;                                      // [16][NT*16]
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float* __restrict__ e = &red[(4 * q + reg) * (NT * 16) + nt * 16 + r];
                    *e = (wv == 0) ? acc[nt][reg] : *e + acc[nt][reg];
                }
        }
    }
    __syncthreads();
    const long nw = (long)Cout * Cin * 343;
    float* __restrict__ pd = part + (long)blockIdx.x * nw;
    for (int e = threadIdx.x; e < 16 * NT * 16; e += 256) {
        const int co = e / (NT * 16), n = e % (NT * 16);
        if (n < npairs) pd[((long)(co0 + co) * Cin + cib) * 343 + n] = red[e];
    }
    if (db != nullptr && blockIdx.z == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0) atomicAdd(db + co0 + r, bsum);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same weight gradient on the f16 matrix pipe (round 5; VERDICT r4 item 6: 247 us in the profiled step = 8 % of either roof).
//   dW[co][ci][kd][kh][kw] = sum_{b,od,oh,ow} dy[b,co,od,oh,ow] * x[b,ci,4od+kd-3,4oh+kh-3,4ow+kw-3]
// One v_mfma_f32_16x16x32_f16 reduces over the 32 outputs of one output ROW (Wo = 32): rows = the 16 output channels (A = dy, a lane's 8 reduction values are 8
// consecutive ow: two float4 loads), columns = 16 taps = 2 (kd, kh) rows x 8 kw slots (7 taps + one idle), B = x: a lane needs x[4 ow + kw - 3] for 8 consecutive ow, a
// stride-4 walk of the input row -- so the halo rows are staged DE-INTERLEAVED, Q[row][kw][j] = x[row][4 j + kw - 3] (7 x 32 halfs per input row: one aligned 16-byte LDS
// read per operand piece; an input element lands in one or two of the seven arrays).  fp32 accuracy from two fp16 pieces per operand scaled by per-tensor powers of two
// (three piece products, as in the forward vx_stem_fwd_k; the scales come from a max kernel over x and dy, so that one accumulator set serves every tile).
// A block = 4 waves = 4 consecutive output rows of one (b, od) for ONE input channel (grid.y = Cin): 7 x 19 staged input rows = 119 KB of LDS (one block per CU), 25
// accumulator tiles (49 (kd, kh) rows in pairs), tiles of the block's share one after the other; block reduction through LDS, one partial slice per block, vx_wg_reduce_k.
// ---------------------------------------------------------------------------------------------------------------------------
typedef _Float16 vx_sw_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vx_sw_h2 __attribute__((ext_vector_type(2)));
typedef float vx_sw_f2 __attribute__((ext_vector_type(2)));
#define VX_SW_NROW (7 * 19)
#define VX_SW_ROWH (7 * 32)
#define VX_SW_NT 25
__device__ __forceinline__ void vx_sw_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const vx_sw_f2 v = {a, b};
    const vx_sw_h2 h = __builtin_convertvector(v, vx_sw_h2);
    const vx_sw_h2 l = __builtin_convertvector(v - __builtin_convertvector(h, vx_sw_f2), vx_sw_h2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ int vx_sw_exp16(float m) {          // |m| * 2^e in [2^13, 2^14); 0 for m = 0 / non-finite
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
    int e = 13 - ilogbf(m);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
// mx[0] = max |x|, mx[1] = max |dy| as float bits (non-negative floats order like unsigned integers); mx zeroed by the caller
// (round 6) xin != null: max |x| is already known (left by the forward stem kernel, vx_conv_mfma_fwd_mx) -- it is copied, and only dy is read (grid (n, 1), x = null)
__global__ void __launch_bounds__(256) vx_stem_absmax_k(const float4* __restrict__ x, long nx4, const float4* __restrict__ dy, long ny4, unsigned* __restrict__ mx,
                                                        const unsigned* __restrict__ xin) {
    if (xin != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(mx, *xin);
    const bool second = blockIdx.y == 1 || x == nullptr;
    const float4* __restrict__ p = second ? dy : x;
    const long n = second ? ny4 : nx4;
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float4 v = p[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = vx_wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(mx + (second ? 1 : 0), __float_as_uint(m));
}
__global__ void __launch_bounds__(256) vx_stem_wgrad_f16_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, float* __restrict__ db,
                                                           const unsigned* __restrict__ mx, int B, int Cin, int Di, int Hi, int Wi, int Do, int Ho, int tiles_per_block, int ntiles) {
    constexpr int U = 9, NIT = (VX_SW_NROW * 32 + 256 * U - 1) / (256 * U);
    const int Wo = Wi >> 2, nq = Wi >> 2;              // 128-wide rows: 32 outputs = the 32 reduction slots; 96-wide rows (the shipped 96^3 patches): 24, the last lane group idles on zeros
    extern __shared__ __attribute__((aligned(16))) unsigned char vx_sw_lds[];
    unsigned short* __restrict__ qh = reinterpret_cast<unsigned short*>(vx_sw_lds);            // [133][7][32] hi pieces
    unsigned short* __restrict__ ql = qh + VX_SW_NROW * VX_SW_ROWH;                              // lo pieces
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n = lane & 15, G = lane >> 4;
    const int ci = blockIdx.y;
    const int ex = vx_sw_exp16(__uint_as_float(mx[0])), ey = vx_sw_exp16(__uint_as_float(mx[1]));
    const float sx = ldexpf(1.0f, ex), sy = ldexpf(1.0f, ey);
    typedef float sw_f4 __attribute__((ext_vector_type(4)));
    sw_f4 acc[VX_SW_NT];
#pragma unroll
    for (int nt = 0; nt < VX_SW_NT; ++nt) acc[nt] = (sw_f4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const int kw = (n & 7) < 7 ? (n & 7) : 0;                  // (slot 7 of a row is idle: its column is never stored)
    const long Vi = (long)Di * Hi * Wi, Vo = (long)Do * Ho * Wo;
    const int nHg = Ho / 4;
    // (round 6) consecutive block ids go to different XCDs; tiles of neighbouring output planes share 3 of their 7 staged input planes: every XCD takes a contiguous run of
    // blocks, so the overlap is served by its L2 instead of being fetched by two XCDs (counted 134 MB per launch for 75 MB of operands)
    const int bx = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int t_begin = bx * tiles_per_block, t_end = min(t_begin + tiles_per_block, ntiles);
    for (int t = t_begin; t < t_end; ++t) {
        int tt = t;
        const int hg = tt % nHg; tt /= nHg;
        const int od = tt % Do;
        const int b = tt / Do;
        __syncthreads();                                        // the previous tile's operand reads are done
        // this wave's dy row (A operand): issued before the staging so that it travels with it
        const bool gok = 8 * G < Wo;
        const float* __restrict__ dyr = dy + ((long)b * 16 + n) * Vo + ((long)od * Ho + 4 * hg + wave) * Wo + (gok ? 8 * G : 0);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 d0r = *reinterpret_cast<const float4*>(dyr), d1r = *reinterpret_cast<const float4*>(dyr + 4);
        const float4 d0 = gok ? d0r : z4, d1 = gok ? d1r : z4;
        const float* __restrict__ xb = x + ((long)b * Cin + ci) * Vi;
#pragma unroll 1
        for (int pass = 0; pass < NIT; ++pass) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = threadIdx.x + (pass * U + u) * 256;
                const int row = min(it >> 5, VX_SW_NROW - 1), m = it & 31;
                const int kd = row / 19, hh = row - kd * 19;
                const int id = 4 * od - 3 + kd, ih = 16 * hg - 3 + hh;
                const bool ok = (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi && m < nq;          // (m >= Wi / 4: beyond the row -- zeros, so that no slot holds stale LDS)
                const float4 t_ = *reinterpret_cast<const float4*>(xb + ((long)(ok ? id : 0) * Hi + (ok ? ih : 0)) * Wi + (ok ? 4 * m : 0));
                v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int it = threadIdx.x + (pass * U + u) * 256;
                if (it < VX_SW_NROW * 32) {
                    const int row = it >> 5, m = it & 31;
                    uint32_t h01, l01, h23, l23;
                    vx_sw_split2(v[u].x * sx, v[u].y * sx, h01, l01);
                    vx_sw_split2(v[u].z * sx, v[u].w * sx, h23, l23);
                    const int rb = row * VX_SW_ROWH + m;
                    // input w = 4 m + e  ->  u = w + 3 = 4 j + p:  e = 0 -> (p 3, j m);  e = 1, 2, 3 -> (p 0, 1, 2, j m + 1) and, as tap p + 4, (j m)
                    qh[rb + 3 * 32] = (unsigned short)h01;            ql[rb + 3 * 32] = (unsigned short)l01;
                    qh[rb + 4 * 32] = (unsigned short)(h01 >> 16);    ql[rb + 4 * 32] = (unsigned short)(l01 >> 16);
                    qh[rb + 5 * 32] = (unsigned short)h23;            ql[rb + 5 * 32] = (unsigned short)l23;
                    qh[rb + 6 * 32] = (unsigned short)(h23 >> 16);    ql[rb + 6 * 32] = (unsigned short)(l23 >> 16);
                    if (m < 31) {
                        qh[rb + 1] = (unsigned short)(h01 >> 16);          ql[rb + 1] = (unsigned short)(l01 >> 16);
                        qh[rb + 32 + 1] = (unsigned short)h23;             ql[rb + 32 + 1] = (unsigned short)l23;
                        qh[rb + 64 + 1] = (unsigned short)(h23 >> 16);     ql[rb + 64 + 1] = (unsigned short)(l23 >> 16);
                    }
                    if (m == 0) {                                     // left padding: u = 0, 1, 2 are the inputs w = -3, -2, -1
                        qh[rb] = 0; ql[rb] = 0; qh[rb + 32] = 0; ql[rb + 32] = 0; qh[rb + 64] = 0; ql[rb + 64] = 0;
                    }
                }
            }
        }
        __syncthreads();
        bsum += ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
        uint4 ah, al;
        vx_sw_split2(d0.x * sy, d0.y * sy, ah.x, al.x);
        vx_sw_split2(d0.z * sy, d0.w * sy, ah.y, al.y);
        vx_sw_split2(d1.x * sy, d1.y * sy, ah.z, al.z);
        vx_sw_split2(d1.z * sy, d1.w * sy, ah.w, al.w);
        const vx_sw_h8 a_hi = __builtin_bit_cast(vx_sw_h8, ah), a_lo = __builtin_bit_cast(vx_sw_h8, al);
        int kd = 0, kh = n >> 3;                                // (kd, kh) row c0 = 2 nt + (n >> 3) of this lane's column
#pragma unroll
        for (int nt = 0; nt < VX_SW_NT; ++nt) {
            const int kdc = kd < 7 ? kd : 6;                    // (the 50th row -- tile 24, upper half -- is idle)
            const int off = ((kdc * 19 + 4 * wave + kh) * 7 + kw) * 32 + 8 * G;
            const uint4 bh = *reinterpret_cast<const uint4*>(qh + off), bl = *reinterpret_cast<const uint4*>(ql + off);
            const vx_sw_h8 b_hi = __builtin_bit_cast(vx_sw_h8, bh), b_lo = __builtin_bit_cast(vx_sw_h8, bl);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_hi, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b_lo, acc[nt], 0, 0, 0);
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b_hi, acc[nt], 0, 0, 0);
            kh += 2;
            if (kh >= 7) { kh -= 7; ++kd; }
        }
    }
    // block reduction: D row 4 G + reg = co, column n = (row c0 = 2 nt + (n >> 3), tap n & 7)
    float* __restrict__ red = reinterpret_cast<float*>(vx_sw_lds);                          // [16][25 * 16]
    for (int wv = 0; wv < 4; ++wv) {
        __syncthreads();
        if (wave == wv) {
#pragma unroll
            for (int nt = 0; nt < VX_SW_NT; ++nt)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    float* __restrict__ e = &red[(4 * G + reg) * (VX_SW_NT * 16) + nt * 16 + n];
                    *e = (wv == 0) ? acc[nt][reg] : *e + acc[nt][reg];
                }
        }
    }
    __syncthreads();
    const float inv = ldexpf(1.0f, -(ex + ey));
    const long nw = (long)16 * Cin * 343;
    float* __restrict__ pd = part + (long)blockIdx.x * nw;
    for (int e = threadIdx.x; e < 16 * VX_SW_NT * 16; e += 256) {
        const int co = e / (VX_SW_NT * 16), col = e % (VX_SW_NT * 16);
        const int c0 = 2 * (col >> 4) + ((col >> 3) & 1), tap = col & 7;
        if (c0 < 49 && tap < 7) pd[((long)co * Cin + ci) * 343 + c0 * 7 + tap] = red[e] * inv;
    }
    if (db != nullptr && ci == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (G == 0) atomicAdd(db + n, bsum);
    }
}
static int vx_stem_wg_f16 = -1;
static int stem_wg_f16_on() {
    if (vx_stem_wg_f16 < 0) { const char* e = getenv("VELOXSEG_STEM_WG_F16"); vx_stem_wg_f16 = e ? atoi(e) : 1; if (vx_stem_wg_f16 < 0) vx_stem_wg_f16 = 0; }
    return vx_stem_wg_f16;
}
extern "C" int vx_down_wgrad_set_f16(int on) { vx_stem_wg_f16 = on ? 1 : 0; return 0; }      // A/B knob (tests): the stem weight gradient on the f16 pipe (default) or the fp32 MFMA kernel
static bool stem_wg_f16_ok(int Cin, int Di, int Hi, int Wi, int Cout) { return Cout == 16 && (Wi == 128 || Wi == 96 || Wi == 64) && Hi % 16 == 0 && Di % 4 == 0 && Cin >= 1 && Cin <= 8; }

static int vx_down_cfg(int B, int Cin, int Di, int Hi, int Wi, int Cout, int& CB, int& Do, int& Ho, int& Wo, int& ntiles, int& tpb, int& nblk) {
    Do = (Di + 6 - 7) / 4 + 1; Ho = (Hi + 6 - 7) / 4 + 1; Wo = (Wi + 6 - 7) / 4 + 1;
    if (Cout % 16 != 0 || Wo < 1 || Ho % 4 != 0 || Do < 1) return 1;      // rows that are not a multiple of 16 (24: the 96^3 patches) leave part of their last tile idle
    CB = (Cin % 2 == 0) ? 2 : 1;
    if (Cin % CB != 0) return 1;
    ntiles = B * Do * (Ho / 4) * ((Wo + 15) / 16);
    const int per_slice = (Cout / 16) * (Cin / CB);
    int want = 512 / per_slice;                       // 2 blocks per CU = 2 waves per SIMD (<= 256 registers, 2 x 72 KB of LDS): one stages while the other runs MFMAs
    if (want < 1) want = 1;
    tpb = (ntiles + want - 1) / want;
    if (tpb < 1) tpb = 1;
    nblk = (ntiles + tpb - 1) / tpb;
    return 0;
}
// workspace floats of vx_down_wgrad_mfma (0: shape not covered -> use vx_conv3d_bwd_weight_tiled)
extern "C" int vx_down_wgrad_ws_floats(int B, int Cin, int Di, int Hi, int Wi, int Cout) {
    int CB, Do, Ho, Wo, ntiles, tpb, nblk;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || vx_down_cfg(B, Cin, Di, Hi, Wi, Cout, CB, Do, Ho, Wo, ntiles, tpb, nblk)) return 0;
    const long n = (long)nblk * Cout * Cin * 343 + 64;          // (+ the two maxima of the f16 kernel)
    return n > 0x7fffffffL ? 0 : (int)n;
}
// Conv3d(k7, s4, p3) weight + bias gradient (dw +=, db += ; db may be NULL).  Returns 1 (nothing launched) when the shape is not covered.
extern "C" int vx_down_wgrad_mfma_mx(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, const unsigned* x_absmax,
                                     int B, int Cin, int Di, int Hi, int Wi, int Cout, void* stream);
extern "C" int vx_down_wgrad_mfma(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats,
                                  int B, int Cin, int Di, int Hi, int Wi, int Cout, void* stream) {
    return vx_down_wgrad_mfma_mx(x, dy, dw, db, ws, ws_floats, nullptr, B, Cin, Di, Hi, Wi, Cout, stream);
}
// x_absmax (may be null): the bits of max |x| left by vx_conv_mfma_fwd_mx for the same x -- the f16-pipe kernel then reads only dy to find its scales
extern "C" int vx_down_wgrad_mfma_mx(const float* x, const float* dy, float* dw, float* db, float* ws, long ws_floats, const unsigned* x_absmax,
                                     int B, int Cin, int Di, int Hi, int Wi, int Cout, void* stream) {
    VX_REQUIRE(x && dy && dw && ws && B > 0 && Cin > 0 && Cout > 0, "vx_down_wgrad_mfma: bad args");
    int CB, Do, Ho, Wo, ntiles, tpb, nblk;
    if (vx_down_cfg(B, Cin, Di, Hi, Wi, Cout, CB, Do, Ho, Wo, ntiles, tpb, nblk)) return 1;
    const long nw = (long)Cout * Cin * 343;
    VX_REQUIRE(ws_floats >= (long)nblk * nw, "vx_down_wgrad_mfma: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (stem_wg_f16_on() && stem_wg_f16_ok(Cin, Di, Hi, Wi, Cout) && ws_floats >= (long)nblk * nw + 64) {
        static bool attr = false;
        if (!attr) {
            VX_REQUIRE(hipFuncSetAttribute((const void*)vx_stem_wgrad_f16_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "vx_down_wgrad_mfma: LDS attribute");
            attr = true;
        }
        unsigned* mx = reinterpret_cast<unsigned*>(ws + (long)nblk * nw);
        VX_REQUIRE(hipMemsetAsync(mx, 0, 8, st) == hipSuccess, "vx_down_wgrad_mfma: memset");
        const long nx4 = (long)B * Cin * Di * Hi * Wi / 4, ny4 = (long)B * Cout * Do * Ho * Wo / 4;
        if (x_absmax != nullptr) vx_stem_absmax_k<<<dim3(256, 1), dim3(256), 0, st>>>(nullptr, 0, reinterpret_cast<const float4*>(dy), ny4, mx, x_absmax);
        else vx_stem_absmax_k<<<dim3(256, 2), dim3(256), 0, st>>>(reinterpret_cast<const float4*>(x), nx4, reinterpret_cast<const float4*>(dy), ny4, mx, nullptr);
        const int nt16 = B * Do * (Ho / 4);                        // tiles: (b, od, group of 4 output rows)
        int nb16 = 256 / Cin;                                       // one block per CU (119 KB of LDS)
        if (nb16 < 1) nb16 = 1;
        if (nb16 > nt16) nb16 = nt16;
        if (nb16 > nblk) nb16 = nblk;                               // (the slices of the workspace)
        const int tp16 = (nt16 + nb16 - 1) / nb16;
        nb16 = (nt16 + tp16 - 1) / tp16;
        const size_t shm16 = (size_t)2 * VX_SW_NROW * VX_SW_ROWH * 2;
        vx_stem_wgrad_f16_k<<<dim3(nb16, Cin), dim3(256), shm16, st>>>(x, dy, ws, db, mx, B, Cin, Di, Hi, Wi, Do, Ho, tp16, nt16);
        int sc16 = nb16 / 32;
        if (sc16 < 1) sc16 = 1;
        vx_wg_reduce_k<<<dim3(vx_cdiv(nw, 256), sc16), dim3(256), 0, st>>>(ws, dw, nw, nb16);
        VX_LAUNCH_CHECK("vx_down_wgrad_mfma (f16 pipe)");
        return 0;
    }
    const size_t shm = sizeof(float) * ((((size_t)CB * VX_DW_PLANE + 3) & ~(size_t)3) + (size_t)(CB == 2 ? 43 : 22) * 16);
    dim3 grid(nblk, Cout / 16, Cin / CB);
    if (CB == 2) vx_down_wgrad_mfma_k<43><<<grid, 256, shm, st>>>(x, dy, ws, db, B, Cin, Di, Hi, Wi, Cout, Do, Ho, Wo, CB, tpb, ntiles);
    else vx_down_wgrad_mfma_k<22><<<grid, 256, shm, st>>>(x, dy, ws, db, B, Cin, Di, Hi, Wi, Cout, Do, Ho, Wo, CB, tpb, ntiles);
    int sc = nblk / 32;
    if (sc < 1) sc = 1;
    if (sc > 32) sc = 32;
    vx_wg_reduce_k<<<dim3(vx_cdiv(nw, 256), sc), dim3(256), 0, st>>>(ws, dw, nw, nblk);
    VX_LAUNCH_CHECK("vx_down_wgrad_mfma");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a dense strided Conv3d as a gather-GEMM on the fp32 matrix pipe (round 5): the DownConvs of encoder levels 2 - 4 (Conv3d k3 s2 p1, 16 -> 32 -> 64 ->
// 128 channels; reference conv_blocks.py:4-21) ran on the tiled VALU kernel above at 78 - 82 us for 0.45 GFLOP each (3.7 % of the fp32 roof), on the critical path of the
// encoder backward.   dW[co][ci][tap] = sum_{b, v} dy[b, co, v] * x[b, ci, S v + tap - P]      M = Cout, N = (ci, tap) pairs, K = B x Vout output voxels.
// v_mfma_f32_16x16x4_f32, four k-steps per iteration: step j of lane group q is voxel v0 + 4 q + j, so that a lane's four A values are ONE 16-byte load of its dy row and
// its four B values are four gathers of its pair's input walk (x is 0.5 - 8 MB at these levels: L2 / MALL resident; out-of-range taps read as zero).  A wave owns one
// 16-pair tile and every output-channel tile (MT accumulators) over a chunk of a sample's voxels; the four waves of a block take four chunks of the same pair tile and are
// added through LDS: one float atomic per (weight, block).  Exact fp32 products, fp32 accumulation.
// ---------------------------------------------------------------------------------------------------------------------------
struct VxGw {
    const float *x, *dy;
    float *dw, *db;          // db (may be NULL): += sum of dy, formed by the waves of pair tile 0 from the A operands they load anyway
    int B, Cin, Di, Hi, Wi, Cout, Do, Ho, Wo, K, S, P;
    int KV, npairs, chunk, nchunk, al4;      // taps per channel, Cin * KV, voxels per wave chunk, chunks per sample, output rows 16-byte loadable
};
template <int MT>
__global__ void __launch_bounds__(256) vx_wgrad_gather_mfma_k(VxGw p) {
    __shared__ float red[4][MT][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), r = lane & 15, q = lane >> 4;
    const int nt = blockIdx.x;                                  // pair tile
    const int cg = blockIdx.y * 4 + wave;                       // (sample, chunk)
    const int b = cg / p.nchunk, ch = cg - b * p.nchunk;
    const bool wok = b < p.B;
    const int n = nt * 16 + r;
    const bool nok = n < p.npairs;
    const int nc = nok ? n : 0;
    const int ci = nc / p.KV, tap = nc - ci * p.KV;
    const int kw = tap % p.K, kh = (tap / p.K) % p.K, kd = tap / (p.K * p.K);
    const long Vi = (long)p.Di * p.Hi * p.Wi, Vo = (long)p.Do * p.Ho * p.Wo;
    const float* __restrict__ xb = p.x + ((long)(wok ? b : 0) * p.Cin + ci) * Vi;
    const float* __restrict__ dyb = p.dy + (long)(wok ? b : 0) * p.Cout * Vo;
    vx_wf4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (vx_wf4){0.f, 0.f, 0.f, 0.f};
    const int HoWo = p.Ho * p.Wo;
    const long v_lo = (long)ch * p.chunk, v_hi = min((long)(ch + 1) * p.chunk, Vo);
    const bool bias = p.db != nullptr && nt == 0;
    float bs[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bs[mt] = 0.0f;
    if (wok)
        for (long v0 = v_lo; v0 < v_hi; v0 += 16) {
            const long vq = v0 + 4 * q;                          // this lane group's four voxels vq .. vq + 3 (Vo % 4 == 0: never straddles the end)
            const bool vok = vq < v_hi;
            float4 av[MT];
            if (p.al4) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(dyb + (long)(16 * mt + r) * Vo + (vok ? vq : 0));
            } else {                                              // rows whose length is not a multiple of 4 (3^3 outputs: the 96^3 patches): four scalar loads
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float* __restrict__ rp = dyb + (long)(16 * mt + r) * Vo;
                    const float t0 = rp[vq < v_hi ? vq : 0], t1 = rp[vq + 1 < v_hi ? vq + 1 : 0], t2 = rp[vq + 2 < v_hi ? vq + 2 : 0], t3 = rp[vq + 3 < v_hi ? vq + 3 : 0];
                    av[mt] = make_float4(vq < v_hi ? t0 : 0.f, vq + 1 < v_hi ? t1 : 0.f, vq + 2 < v_hi ? t2 : 0.f, vq + 3 < v_hi ? t3 : 0.f);
                }
            }
            float bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool vj = vok && vq + j < v_hi;
                const int v = (int)(vj ? vq + j : 0);
                const int od = v / HoWo, rem = v - od * HoWo, oh = rem / p.Wo, ow = rem - oh * p.Wo;
                const int id = od * p.S - p.P + kd, ih = oh * p.S - p.P + kh, iw = ow * p.S - p.P + kw;
                const bool ok = vj && nok && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi;
                const float t_ = xb[ok ? ((long)id * p.Hi + ih) * p.Wi + iw : 0];
                bv[j] = ok ? t_ : 0.0f;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float a4[4] = {vok ? av[mt].x : 0.0f, vok ? av[mt].y : 0.0f, vok ? av[mt].z : 0.0f, vok ? av[mt].w : 0.0f};
                bs[mt] += (a4[0] + a4[1]) + (a4[2] + a4[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bv[j], acc[mt], 0, 0, 0);
            }
        }
    if (bias && wok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float t_ = bs[mt];
            t_ += __shfl_xor(t_, 16, 64);
            t_ += __shfl_xor(t_, 32, 64);
            if (q == 0) atomicAdd(p.db + 16 * mt + r, t_);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave][mt][reg][lane] = acc[mt][reg];
    __syncthreads();
    // D row 4 q + reg = co of tile mt, column r = pair n
    for (int e = threadIdx.x; e < MT * 4 * 64; e += 256) {
        const int l = e & 63, reg = (e >> 6) & 3, mt = e >> 8;
        const int co = 16 * mt + 4 * (l >> 4) + reg, nn = nt * 16 + (l & 15);
        if (nn < p.npairs) {
            const float v = (red[0][mt][reg][l] + red[1][mt][reg][l]) + (red[2][mt][reg][l] + red[3][mt][reg][l]);
            atomicAdd(p.dw + (long)co * p.npairs + nn, v);
        }
    }
}
static int vx_wg_gather = -1;
extern "C" int vx_conv_wgrad_gather_set(int on) { vx_wg_gather = on ? 1 : 0; return 0; }       // A/B knob (tests)
extern "C" int vx_conv_wgrad_gather_ok(int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P) {
    if (vx_wg_gather < 0) { const char* e = getenv("VELOXSEG_WGRAD_GATHER"); vx_wg_gather = e ? atoi(e) : 1; }
    if (!vx_wg_gather || B <= 0 || Cin <= 0 || K < 1 || K > 5 || S < 1 || P < 0) return 0;
    if (Cout != 32 && Cout != 64 && Cout != 128) return 0;
    const int Do = (Di + 2 * P - K) / S + 1, Ho = (Hi + 2 * P - K) / S + 1, Wo = (Wi + 2 * P - K) / S + 1;
    if (Do < 1 || Ho < 1 || Wo < 1) return 0;
    const long Vo = (long)Do * Ho * Wo;
    return (Vo <= (1L << 20) && (long)Cin * K * K * K <= (1L << 20)) ? 1 : 0;
}
/* dw += the weight gradient of Conv3d(Cin -> Cout, kernel K, stride S, padding P, no groups), db += the bias gradient (db may be NULL); exact fp32 products on
   v_mfma_f32_16x16x4_f32.  Shapes: vx_conv_wgrad_gather_ok. */
extern "C" int vx_conv_wgrad_gather_mfma(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Di, int Hi, int Wi, int Cout, int K, int S, int P, void* stream) {
    VX_REQUIRE(x && dy && dw, "vx_conv_wgrad_gather_mfma: null pointer");
    VX_REQUIRE(vx_conv_wgrad_gather_ok(B, Cin, Di, Hi, Wi, Cout, K, S, P) == 1, "vx_conv_wgrad_gather_mfma: shape not covered (Cin=%d Cout=%d k%d s%d p%d %dx%dx%d)", Cin, Cout, K, S, P, Di, Hi, Wi);
    VxGw p = {};
    p.x = x; p.dy = dy; p.dw = dw; p.db = db; p.B = B; p.Cin = Cin; p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Cout = Cout; p.K = K; p.S = S; p.P = P;
    p.Do = (Di + 2 * P - K) / S + 1; p.Ho = (Hi + 2 * P - K) / S + 1; p.Wo = (Wi + 2 * P - K) / S + 1;
    p.KV = K * K * K; p.npairs = Cin * p.KV;
    const long Vo = (long)p.Do * p.Ho * p.Wo;
    p.al4 = (Vo % 4 == 0) ? 1 : 0;
    const int ntile = vx_cdiv(p.npairs, 16);
    // waves = pair tiles x (B x chunks): aim at ~2 k waves (two per SIMD), chunks of >= 64 voxels (multiples of 16)
    long nch = 2048 / ((long)ntile * B);
    if (nch < 1) nch = 1;
    long chunk = (Vo + nch - 1) / nch;
    chunk = (chunk + 15) / 16 * 16;
    if (chunk < 64) chunk = 64;
    p.chunk = (int)chunk;
    p.nchunk = (int)((Vo + chunk - 1) / chunk);
    const dim3 grid((unsigned)ntile, (unsigned)vx_cdiv(B * p.nchunk, 4));
    hipStream_t st = (hipStream_t)stream;
    if (Cout == 32) vx_wgrad_gather_mfma_k<2><<<grid, dim3(256), 0, st>>>(p);
    else if (Cout == 64) vx_wgrad_gather_mfma_k<4><<<grid, dim3(256), 0, st>>>(p);
    else vx_wgrad_gather_mfma_k<8><<<grid, dim3(256), 0, st>>>(p);
    VX_LAUNCH_CHECK("vx_conv_wgrad_gather_mfma");
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// The three weight gradients of a JLC block (grouped convolutions k = 5 / 3 / 1, stride 1, "same" padding: conv_blocks.py:51-58) at SMALL volumes as gather-GEMMs in one
// launch -- the 8^3 / 4^3 levels of the 128^3 configurations, where the Toeplitz kernel of jlc_mfma.hip spends 41 / 29 us on 0.3 / 0.16 GFLOP (a march over planes with two
// barriers per plane for a handful of blocks), and the 6^3 / 3^3 levels of the 96^3 configurations, which it does not cover at all (W % 4).  Same scheme as
// vx_wgrad_gather_mfma_k: rows = the output channels of ONE group (16 rows; a group of 8 fills half of them), columns = 16 (input channel, tap) pairs of that group,
// reduction = the voxels of a sample chunk, four per iteration; blockIdx.x walks the pair tiles of k = 5, then k = 3, then k = 1; blockIdx.z = group.
// ---------------------------------------------------------------------------------------------------------------------------
struct VxJg {
    const float* x;
    const float* g[3];           // g5, g3, g1
    float* dw[3];
    int B, C, G, CG, D, H, W;
    int t3, t1;                  // first pair tile of k = 3 / k = 1
    int chunk, nchunk;
};
__global__ void __launch_bounds__(256) vx_jlc_wgrad_gather_k(VxJg p) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), r = lane & 15, q = lane >> 4;
    const int tile = blockIdx.x;
    const int sel = tile >= p.t1 ? 2 : (tile >= p.t3 ? 1 : 0);
    const int K = sel == 0 ? 5 : (sel == 1 ? 3 : 1), KV = K * K * K, P = K >> 1;
    const int nt = tile - (sel == 2 ? p.t1 : (sel == 1 ? p.t3 : 0));
    const int npairs = p.CG * KV;
    const int grp = blockIdx.z;
    const int cg = blockIdx.y * 4 + wave;
    const int b = cg / p.nchunk, ch = cg - b * p.nchunk;
    const bool wok = b < p.B;
    const int n = nt * 16 + r;
    const bool nok = n < npairs;
    const int nc = nok ? n : 0;
    const int ci = nc / KV, tap = nc - ci * KV;
    const int kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
    const long V = (long)p.D * p.H * p.W;
    const float* __restrict__ xb = p.x + ((long)(wok ? b : 0) * p.C + grp * p.CG + ci) * V;
    const bool rok = r < p.CG;                                   // (a group of 8: rows 8 .. 15 idle)
    const float* __restrict__ dyr = p.g[sel] + ((long)(wok ? b : 0) * p.C + grp * p.CG + (rok ? r : 0)) * V;
    vx_wf4 acc = {0.f, 0.f, 0.f, 0.f};
    const int HW = p.H * p.W;
    const long v_lo = (long)ch * p.chunk, v_hi = min((long)(ch + 1) * p.chunk, V);
    if (wok)
        for (long v0 = v_lo; v0 < v_hi; v0 += 16) {
            const long vq = v0 + 4 * q;
            float a4[4], bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long v = vq + j;
                const bool vok = v < v_hi;                          // (V need not be a multiple of 4 here: 3^3 = 27)
                const int vi = (int)(vok ? v : 0);
                const float ta = dyr[vi];
                a4[j] = (vok && rok) ? ta : 0.0f;
                const int od = vi / HW, rem = vi - od * HW, oh = rem / p.W, ow = rem - oh * p.W;
                const int id = od - P + kd, ih = oh - P + kh, iw = ow - P + kw;
                const bool ok = vok && nok && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const float tb = xb[ok ? (id * p.H + ih) * p.W + iw : 0];
                bv[j] = ok ? tb : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], bv[j], acc, 0, 0, 0);
        }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave][reg][lane] = acc[reg];
    __syncthreads();
    {
        const int l = threadIdx.x & 63, reg = threadIdx.x >> 6;
        const int co = 4 * (l >> 4) + reg, nn = nt * 16 + (l & 15);
        if (co < p.CG && nn < npairs) {
            const float v = (red[0][reg][l] + red[1][reg][l]) + (red[2][reg][l] + red[3][reg][l]);
            atomicAdd(p.dw[sel] + (long)(grp * p.CG + co) * npairs + nn, v);          // dw_K layout (C, CG, K, K, K)
        }
    }
}
static int vx_jg_on = -1;
static long vx_jg_max_v = 512;
extern "C" int vx_jlc_wgrad_gather_set(int on) { vx_jg_on = on ? 1 : 0; return 0; }      // A/B knob (tests)
extern "C" int vx_jlc_wgrad_gather_ok(int C, int G, int D, int H, int W) {
    if (vx_jg_on < 0) { const char* e = getenv("VELOXSEG_JLC_WG_GATHER"); vx_jg_on = e ? atoi(e) : 1; const char* m = getenv("VELOXSEG_JLC_WG_GATHER_MAX_V"); if (m) vx_jg_max_v = atol(m); }
    if (!vx_jg_on || C <= 0 || G <= 0 || C % G) return 0;
    const int CG = C / G;
    return ((CG == 8 || CG == 16 || CG == 4) && (long)D * H * W <= vx_jg_max_v && D > 0 && H > 0 && W > 0) ? 1 : 0;
}
/* dw1 / dw3 / dw5 += the weight gradients of the three grouped convolutions of a JLC block (any of the three may be NULL); x, g1, g3, g5: (B, C, D, H, W) */
extern "C" int vx_jlc_wgrad_gather(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                                   void* stream) {
    VX_REQUIRE(x && g1 && g3 && g5 && B > 0, "vx_jlc_wgrad_gather: null pointer");
    VX_REQUIRE(vx_jlc_wgrad_gather_ok(C, G, D, H, W) == 1, "vx_jlc_wgrad_gather: shape not covered (C=%d G=%d %dx%dx%d)", C, G, D, H, W);
    VX_REQUIRE(dw1 && dw3 && dw5, "vx_jlc_wgrad_gather: the three weight gradients are formed together");
    VxJg p = {};
    p.x = x; p.g[0] = g5; p.g[1] = g3; p.g[2] = g1; p.dw[0] = dw5; p.dw[1] = dw3; p.dw[2] = dw1;
    p.B = B; p.C = C; p.G = G; p.CG = C / G; p.D = D; p.H = H; p.W = W;
    const int n5 = vx_cdiv(p.CG * 125, 16), n3 = vx_cdiv(p.CG * 27, 16), n1 = vx_cdiv(p.CG, 16);
    p.t3 = n5; p.t1 = n5 + n3;
    const int ntile = n5 + n3 + n1;
    const long V = (long)D * H * W;
    long nch = 4096 / ((long)ntile * G * B);
    if (nch < 1) nch = 1;
    long chunk = (V + nch - 1) / nch;
    chunk = (chunk + 15) / 16 * 16;
    if (chunk < 64) chunk = 64;
    p.chunk = (int)chunk;
    p.nchunk = (int)((V + chunk - 1) / chunk);
    const dim3 grid((unsigned)ntile, (unsigned)vx_cdiv(B * p.nchunk, 4), (unsigned)G);
    vx_jlc_wgrad_gather_k<<<grid, dim3(256), 0, (hipStream_t)stream>>>(p);
    VX_LAUNCH_CHECK("vx_jlc_wgrad_gather");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1x1x1 GROUPED conv (the k = 1 member of the JLC spatial convs, conv_blocks.py:51-58): weight + bias gradient in one pass.
//   dw[g*CG + co, ci] += sum_{b,v} dy[b, g*CG + co, v] * x[b, g*CG + ci, v];  db[co] += sum dy
// The tiled kernel above maps (ci, tap) pairs to threads, i.e. 4..16 of 256 threads work when K = 1 (141 us at 32^3 x 4 for 17 MFLOP).
// Here a thread owns voxels: 16-byte rows of the CIG inputs and COT outputs of its group, COT*CIG register accumulators, one
// wave-shuffle + LDS reduction at the end and one contiguous atomic flush per block.   grid (chunks, G * CG/COT, B)
// ---------------------------------------------------------------------------------------------------------------------------
template <int CIG, int COT>
__global__ void __launch_bounds__(256) vx_gconv1_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                         float* __restrict__ db, int C, long V) {
    const int sub = CIG / COT;                         // output-channel chunks per group
    const int g = blockIdx.y / sub, cc = blockIdx.y % sub;
    const int b = blockIdx.z;
    const int co0 = g * CIG + cc * COT, ci0 = g * CIG;
    const long V4 = V >> 2;
    const float4* __restrict__ xp = (const float4*)(x + ((long)b * C + ci0) * V);
    const float4* __restrict__ dp = (const float4*)(dy + ((long)b * C + co0) * V);
    float acc[COT][CIG], bs[COT];
#pragma unroll
    for (int o = 0; o < COT; ++o) {
        bs[o] = 0.0f;
#pragma unroll
        for (int i = 0; i < CIG; ++i) acc[o][i] = 0.0f;
    }
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < V4; q += (long)gridDim.x * 256) {
        float4 xv[CIG], dv[COT];
#pragma unroll
        for (int i = 0; i < CIG; ++i) xv[i] = xp[(long)i * V4 + q];
#pragma unroll
        for (int o = 0; o < COT; ++o) dv[o] = dp[(long)o * V4 + q];
#pragma unroll
        for (int o = 0; o < COT; ++o) {
            bs[o] += (dv[o].x + dv[o].y) + (dv[o].z + dv[o].w);
#pragma unroll
            for (int i = 0; i < CIG; ++i)
                acc[o][i] = fmaf(dv[o].w, xv[i].w, fmaf(dv[o].z, xv[i].z, fmaf(dv[o].y, xv[i].y, fmaf(dv[o].x, xv[i].x, acc[o][i]))));
        }
    }
    __shared__ float red[4][COT * CIG + COT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < COT; ++o) {
#pragma unroll
        for (int i = 0; i < CIG; ++i) { const float v = vx_wave_sum(acc[o][i]); if (lane == 0) red[wid][o * CIG + i] = v; }
        const float v = vx_wave_sum(bs[o]);
        if (lane == 0) red[wid][COT * CIG + o] = v;
    }
    __syncthreads();
    const int k = threadIdx.x;
    if (k < COT * CIG + COT) {
        const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (k < COT * CIG) atomicAdd(dw + (long)co0 * CIG + k, v);                 // dw is (C, CIG): rows co0 .. co0+COT-1 are contiguous
        else if (db) atomicAdd(db + co0 + (k - COT * CIG), v);
    }
}

extern "C" int vx_gconv1_bwd_weight(const float* x, const float* dy, float* dw, float* db, int B, int C, int G, long V, void* stream) {
    VX_REQUIRE(x && dy && dw && B > 0 && C > 0 && G > 0 && V > 0 && C % G == 0, "vx_gconv1_bwd_weight: bad args");
    const int CG = C / G;
    VX_REQUIRE((V & 3) == 0 && (CG == 4 || CG == 8 || CG == 16), "vx_gconv1_bwd_weight: needs V %% 4 == 0 and group width 4, 8 or 16 (got V=%ld, width %d)", V, CG);
    int chunks = vx_cdiv(V >> 2, 256 * 4);
    if (chunks > 64) chunks = 64;
    hipStream_t st = (hipStream_t)stream;
    if (CG == 4) vx_gconv1_wgrad_k<4, 4><<<dim3(chunks, G, B), dim3(256), 0, st>>>(x, dy, dw, db, C, V);
    else if (CG == 8) vx_gconv1_wgrad_k<8, 4><<<dim3(chunks, G * 2, B), dim3(256), 0, st>>>(x, dy, dw, db, C, V);
    else vx_gconv1_wgrad_k<16, 4><<<dim3(chunks, G * 4, B), dim3(256), 0, st>>>(x, dy, dw, db, C, V);
    VX_LAUNCH_CHECK("vx_gconv1_bwd_weight");
    return 0;
}
