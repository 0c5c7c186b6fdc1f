// Register-blocked, LDS-staged direct convolution for stride-1 "same" 3-D kernels (K = 3 or 5, pad K/2, groups) on gfx950.
// Serves the FORWARD of the JLC grouped convs (conv_blocks.py:51-58) and of the patch-expand convs (Decoder.py:73-76,150-153,
// with the PixelShuffle of superpixel.py:16 fused into the store) and, through `wmode = 1` (weights transposed inside the
// group and flipped in space while they are staged), their INPUT GRADIENTS (the adjoint of a same-conv is a same-conv).
//
// MI355X mapping: one 256-thread block = (b, COT output channels of one group, one TD x TH x (4*TWq) output tile);
//   a thread owns 4 consecutive voxels along W x COT channels = 4*COT fp32 accumulators;
//   the input halo tile of CIC input channels and the matching weight slice [CIC][K^3][COT] live in LDS;
//   per (ci, kd, kh) a thread issues 2 ds_read_b128 for its 8-voxel input row segment and K*COT/4 broadcast ds_read_b128 for
//   the weights, then 4*K*COT v_fmac -> the loop is VALU-bound (>= 11 FMAs per LDS instruction).
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

struct VxS1 {
    int B, Cin, Cout, D, H, W, G;
    int TD, TH, TWq;          // tile in threads (TWq threads along W, 4 voxels each)
    int HD, HH, HWp;          // halo dims (HWp multiple of 4)
    int nTd, nTh, nTw;
    int wmode;                // 0: w[co][ci][t]   1: adjoint: w[ci'][co'][K^3-1-t] with in-group transposition
    int in_ps, out_ps;        // pixel-shuffle factor of the input / output STORAGE (1 = plain NCDHW)
    int accumulate;           // y += result
    int cic;                  // input channels staged per pass
    int parts;                // input-channel split: the block's 256 threads = SP spatial slots x `parts` channel parts (small tiles), LDS-reduced
    int st_hw, st_hh, st_hd, st_c;   // 256 decomposed in the staging index space (hw fastest): per-thread incremental addressing
};

__device__ __forceinline__ long vx_ps_index(int C, int D, int H, int W, int ps, int b, int c, int d, int h, int w) {
    if (ps == 1) return ((((long)b * C + c) * D + d) * H + h) * (long)W + w;
    const int s = ps;
    const int s3 = c % s;
    int t = c / s;
    const int s2 = t % s;
    t /= s;
    const int s1 = t % s;
    const int cc = t / s;
    const int Cc = C / (s * s * s);
    return ((((long)b * Cc + cc) * (D * s) + d * s + s1) * (long)(H * s) + h * s + s2) * (long)(W * s) + w * s + s3;
}

template <int K, int COT>
__global__ void __launch_bounds__(256) vx_conv_s1_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, VxS1 p) {
    extern __shared__ __attribute__((aligned(16))) float vx_s1_lds[];
    constexpr int K3 = K * K * K;
    constexpr int P = K / 2;
    const int Cin_g = p.Cin / p.G, Cout_g = p.Cout / p.G;
    const int co0 = blockIdx.y * COT;
    const int g = co0 / Cout_g;
    const int b = blockIdx.z;
    const int tile = blockIdx.x;
    const int tw_i = tile % p.nTw, th_i = (tile / p.nTw) % p.nTh, td_i = tile / (p.nTw * p.nTh);
    const int d0 = td_i * p.TD, h0 = th_i * p.TH, w0 = tw_i * p.TWq * 4;
    const int tid = threadIdx.x;
    // small tiles (8^3, 4^3 volumes) leave most of the 256 threads without a spatial slot; they take a share of the input channels instead
    const int SP = p.TWq * p.TH * p.TD;
    const int part = p.parts > 1 ? tid / SP : 0;
    const int sp = p.parts > 1 ? tid - part * SP : tid;
    const int tq = sp % p.TWq, th = (sp / p.TWq) % p.TH, td = sp / (p.TWq * p.TH);
    const bool active = td < p.TD && part < p.parts;
    const int plane = p.HD * p.HH * p.HWp;
    float* __restrict__ xs = vx_s1_lds;
    float* __restrict__ ws = vx_s1_lds + p.cic * plane;

    float acc[4][COT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[u][j] = 0.0f;

    for (int cc = 0; cc < Cin_g; cc += p.cic) {
        const int ncc = min(p.cic, Cin_g - cc);
        __syncthreads();
        // stage the halo of `ncc` input channels: element index e = ((cil*HD + hd)*HH + hh)*HWp + hw, e = tid, tid+256, ...
        // (hd,hh,hw,cil) advance incrementally by the host-decomposed stride: no integer division in the loop.
        if (p.in_ps == 1) {
            int hw = tid % p.HWp, t1 = tid / p.HWp;
            int hh = t1 % p.HH, t2 = t1 / p.HH;
            int hd = t2 % p.HD, cil = t2 / p.HD;
            const long chan_stride = (long)p.D * p.H * p.W;
            const float* __restrict__ xb = x + ((long)b * p.Cin + g * Cin_g + cc) * chan_stride;
            // 4 elements per iteration, loads unconditional (clamped address, value selected afterwards) so that they are all in flight together:
            // with a branch around each load the loop paid one memory latency per element
            const int total = ncc * plane;
            for (int e = tid; e < total; e += 256 * 4) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int id = d0 - P + hd, ih = h0 - P + hh, iw = w0 - P + hw;
                    const bool ok = (e + u * 256 < total) && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    const float t_ = xb[ok ? cil * chan_stride + ((long)id * p.H + ih) * p.W + iw : 0];
                    v[u] = ok ? t_ : 0.0f;
                    hw += p.st_hw; if (hw >= p.HWp) { hw -= p.HWp; ++hh; }
                    hh += p.st_hh; if (hh >= p.HH) { hh -= p.HH; ++hd; }
                    hd += p.st_hd; if (hd >= p.HD) { hd -= p.HD; ++cil; }
                    cil += p.st_c;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (e + u * 256 < total) xs[e + u * 256] = v[u];
            }
        } else {
            // pixel-shuffled source: channel fastest (4 consecutive channels = 16 contiguous bytes)
            for (int e = tid; e < ncc * plane; e += 256) {
                const int cil = e % ncc, r = e / ncc;
                const int hw = r % p.HWp, hh = (r / p.HWp) % p.HH, hd = r / (p.HWp * p.HH);
                const int id = d0 - P + hd, ih = h0 - P + hh, iw = w0 - P + hw;
                float v = 0.0f;
                if ((unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    v = x[vx_ps_index(p.Cin, p.D, p.H, p.W, p.in_ps, b, g * Cin_g + cc + cil, id, ih, iw)];
                xs[cil * plane + r] = v;
            }
        }
        // stage the weight slice [ncc][K3][COT]
        for (int e = tid; e < ncc * K3 * COT; e += 256) {
            const int j = e % COT, t = (e / COT) % K3, cil = e / (COT * K3);
            float v;
            if (p.wmode == 0) v = w[((long)(co0 + j) * Cin_g + (cc + cil)) * K3 + t];
            else v = w[((long)(g * Cin_g + cc + cil) * Cout_g + (co0 + j - g * Cout_g)) * K3 + (K3 - 1 - t)];
            ws[e] = v;
        }
        __syncthreads();
        if (active) {
            for (int cil = part; cil < ncc; cil += p.parts) {
#pragma unroll
                for (int kd = 0; kd < K; ++kd) {
#pragma unroll
                    for (int kh = 0; kh < K; ++kh) {
                        const float* __restrict__ xrow = xs + ((cil * p.HD + td + kd) * p.HH + th + kh) * p.HWp + 4 * tq;
                        const float4 xa = *reinterpret_cast<const float4*>(xrow);
                        const float4 xb = *reinterpret_cast<const float4*>(xrow + 4);
                        const float xr[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                        const float* __restrict__ wp = ws + (cil * K3 + (kd * K + kh) * K) * COT;
#pragma unroll
                        for (int kw = 0; kw < K; ++kw) {
#pragma unroll
                            for (int j4 = 0; j4 < COT; j4 += 4) {
                                const float4 wv = *reinterpret_cast<const float4*>(wp + kw * COT + j4);
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    acc[u][j4 + 0] = fmaf(wv.x, xr[u + kw], acc[u][j4 + 0]);
                                    acc[u][j4 + 1] = fmaf(wv.y, xr[u + kw], acc[u][j4 + 1]);
                                    acc[u][j4 + 2] = fmaf(wv.z, xr[u + kw], acc[u][j4 + 2]);
                                    acc[u][j4 + 3] = fmaf(wv.w, xr[u + kw], acc[u][j4 + 3]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (p.parts > 1) {             // sum the channel parts: [part][slot][4*COT] through LDS (the halo / weight tiles are dead by now)
        __syncthreads();
        if (active) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < COT; ++j) vx_s1_lds[(part * SP + sp) * (4 * COT) + u * COT + j] = acc[u][j];
        }
        __syncthreads();
        if (!active || part != 0) return;
        for (int q = 1; q < p.parts; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < COT; ++j) acc[u][j] += vx_s1_lds[(q * SP + sp) * (4 * COT) + u * COT + j];
    }
    if (!active) return;
    const int od = d0 + td, oh = h0 + th, ow = w0 + 4 * tq;
    if (od >= p.D || oh >= p.H) return;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const float bv = bias ? bias[co0 + j] : 0.0f;
        if (p.out_ps == 1 && (p.W & 3) == 0 && ow + 3 < p.W) {
            float4* dst = reinterpret_cast<float4*>(y + ((((long)b * p.Cout + co0 + j) * p.D + od) * p.H + oh) * (long)p.W + ow);
            float4 o = make_float4(acc[0][j] + bv, acc[1][j] + bv, acc[2][j] + bv, acc[3][j] + bv);
            if (p.accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            *dst = o;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ow + u < p.W) {
                    float* dst = y + vx_ps_index(p.Cout, p.D, p.H, p.W, p.out_ps, b, co0 + j, od, oh, ow + u);
                    const float o = acc[u][j] + bv;
                    *dst = p.accumulate ? *dst + o : o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// x: (B, Cin, D,H,W) [stored pixel-shuffled when in_ps > 1], y: (B, Cout, D,H,W) [pixel-shuffled when out_ps > 1].
// w is ALWAYS the forward convolution's PyTorch weight (Cf_out, Cf_in/G, K,K,K):
//   wmode 0: Cin = Cf_in,  Cout = Cf_out  (forward)
//   wmode 1: Cin = Cf_out, Cout = Cf_in   (input gradient: x := dy, y := dx)
// ------------------------------------------------------------------------------------------------------------------
extern "C" int vx_conv_s1(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int D, int H, int W,
                          int K, int G, int wmode, int in_ps, int out_ps, int accumulate, void* stream) {
    VX_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0 && G > 0, "vx_conv_s1: bad args");
    VX_REQUIRE(K == 3 || K == 5, "vx_conv_s1: K must be 3 or 5 (got %d)", K);
    VX_REQUIRE(Cin % G == 0 && Cout % G == 0, "vx_conv_s1: channels not divisible by groups");
    VX_REQUIRE(in_ps >= 1 && out_ps >= 1 && Cin % (in_ps * in_ps * in_ps) == 0 && Cout % (out_ps * out_ps * out_ps) == 0, "vx_conv_s1: bad pixel-shuffle factor");
    const int Cout_g = Cout / G, Cin_g = Cin / G;
    VX_REQUIRE(Cout_g % 4 == 0, "vx_conv_s1: output channels per group must be a multiple of 4 (got %d)", Cout_g);
    VxS1 p;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.D = D; p.H = H; p.W = W; p.G = G;
    p.wmode = wmode; p.in_ps = in_ps; p.out_ps = out_ps; p.accumulate = accumulate;
    p.TWq = vx_cdiv(W, 4) < 8 ? vx_cdiv(W, 4) : 8;
    p.TH = H < 8 ? H : 8;
    int td = 256 / (p.TWq * p.TH);
    if (td > D) td = D;
    if (td > 8) td = 8;
    if (td < 1) td = 1;
    p.TD = td;
    p.HD = p.TD + K - 1; p.HH = p.TH + K - 1; p.HWp = (p.TWq * 4 + K - 1 + 3) / 4 * 4;
    if (p.HWp < p.TWq * 4 + 4) p.HWp = p.TWq * 4 + 4;     // the 8-float row read of the last thread must stay inside the row
    p.nTd = vx_cdiv(D, p.TD); p.nTh = vx_cdiv(H, p.TH); p.nTw = vx_cdiv(W, p.TWq * 4);
    int COT = (Cout_g % 16 == 0) ? 16 : (Cout_g % 8 == 0) ? 8 : 4;
    // few output channels (e.g. the 16-channel input gradient of a patch-expand conv): narrower register blocks, more blocks
    while (COT > 4 && (long)p.nTd * p.nTh * p.nTw * (Cout / COT) * B < 512) COT >>= 1;
    // small volumes (16^3, 8^3): still too few blocks for 256 CUs -> thinner tiles in D (more halo planes staged per output plane, but the
    // kernel is latency-bound at one wave per SIMD); keep at least one full wave of active threads
    while (p.TD > 1 && (long)p.nTd * p.nTh * p.nTw * (Cout / COT) * B < 768 && (p.TD / 2) * p.TWq * p.TH >= 64) {
        p.TD /= 2;
        p.HD = p.TD + K - 1;
        p.nTd = vx_cdiv(D, p.TD);
    }
    const int plane = p.HD * p.HH * p.HWp;
    // input channels per pass: keep LDS (halo + weights) under ~48 KB so that 3 blocks share a CU
    // input-channel parts for tiles with fewer than 256 spatial slots
    int parts = 1;
    {
        const int SP = p.TWq * p.TH * p.TD;
        while (parts * 2 * SP <= 256 && parts * 2 <= Cin_g && Cin_g % (parts * 2) == 0) parts *= 2;
    }
    int cic = Cin_g < 4 ? Cin_g : 4;
    if (cic < parts) cic = parts;
    auto lds_bytes = [&](int c) { return (size_t)c * (plane + K * K * K * COT) * sizeof(float); };
    while (cic > 1 && cic > parts && lds_bytes(cic) > 48 * 1024) cic >>= 1;
    while (parts > 1 && lds_bytes(cic) > 64 * 1024) { parts >>= 1; if (cic > 4 && cic > parts) cic = parts > 4 ? parts : 4; }
    VX_REQUIRE(lds_bytes(cic) <= 150 * 1024, "vx_conv_s1: tile does not fit LDS");
    p.cic = cic;
    p.parts = parts;
    {   // 256 = ((st_c*HD + st_hd)*HH + st_hh)*HWp + st_hw
        int r = 256;
        p.st_hw = r % p.HWp; r /= p.HWp;
        p.st_hh = r % p.HH; r /= p.HH;
        p.st_hd = r % p.HD; r /= p.HD;
        p.st_c = r;
    }
    dim3 grid(p.nTd * p.nTh * p.nTw, Cout / COT, B);
    hipStream_t st = (hipStream_t)stream;
    size_t shm = lds_bytes(cic);
    if (parts > 1) {
        const size_t red = (size_t)parts * (p.TWq * p.TH * p.TD) * 4 * COT * sizeof(float);
        if (red > shm) shm = red;
    }
#define VX_S1_LAUNCH(KK, CC) vx_conv_s1_k<KK, CC><<<grid, dim3(256), shm, st>>>(x, w, bias, y, p)
    if (K == 3) {
        if (COT == 16) VX_S1_LAUNCH(3, 16); else if (COT == 8) VX_S1_LAUNCH(3, 8); else VX_S1_LAUNCH(3, 4);
    } else {
        if (COT == 16) VX_S1_LAUNCH(5, 16); else if (COT == 8) VX_S1_LAUNCH(5, 8); else VX_S1_LAUNCH(5, 4);
    }
#undef VX_S1_LAUNCH
    VX_LAUNCH_CHECK("vx_conv_s1");
    return 0;
}
