// Dense strided 3-D convolution (groups = 1) as an implicit GEMM on v_mfma_f32_16x16x4_f32: the DownConv layers between the encoder levels
// (k = 3, s = 2, p = 1: Encoder.py:29-58 / conv_blocks.py:8-27 of the reference) and the stem (k = 7, s = 4, p = 3), forward and input gradient.
//
// The generic direct kernels these replace spent 70-180 us per launch on 0.5-3 GFLOP (a few blocks, one output per lane, scalar FMAs).  Here
//   forward : M = output voxels (16 per wave tile), N = output channels, K = (tap, ci) flattened, ci fastest
//   backward: M = input voxels of ONE stride-parity class (so the set of taps that reach them is the same for the whole tile: for s = 2 a voxel
//             is reached by 1, 2, 4 or 8 of the 27 taps depending on the parity of its coordinates), N = input channels, K = (tap, co)
// The A operand (activations) is gathered straight from global memory -- every element is re-read from L2 by the few tiles that share it; the
// whole input of these layers is a few MB -- with the per-k offsets precomputed in LDS; the B operand (weights) is read from an operand-order
// image that a prep kernel writes once per launch (lane l of k-step s, column tile t reads word ((s * NT + t) * 64 + l): one coalesced 256-byte
// load per MFMA).  fp32 in, fp32 accumulate: the same products as the direct kernel, summed in a different order (1e-6 relative).
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

typedef float v4f __attribute__((ext_vector_type(4)));

struct VxCm {
    const float* x;      // forward: input (B, Cin, Di, Hi, Wi); backward: dy (B, Cout, Do, Ho, Wo)
    const float* wop;    // operand-order weights
    const float* bias;
    float* y;            // forward: output (B, Cout, Do, Ho, Wo); backward: dx (B, Cin, Di, Hi, Wi)
    int B, Cin, Di, Hi, Wi, Cout, Do, Ho, Wo;
    int K, S, P;
    int ksteps, NTtot, accumulate;
};

// wop[(ks * NTtot + nt) * 64 + lane]: forward  kk = 4 ks + lane / 16 = tap * Cin + ci,  column = nt * 16 + lane % 16 = co
//                                      backward kk = tap * Cout + co,                     column = ci
__global__ void __launch_bounds__(256) vx_conv_mfma_wprep_k(const float* __restrict__ w, float* __restrict__ wop, int Cout, int Cin, int K3, int ksteps, int NTtot, int backward) {
    const long n = (long)ksteps * NTtot * 64;
    const int Kdim = backward ? Cout : Cin, Ndim = backward ? Cin : Cout;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int lane = (int)(e & 63);
        const long t = e >> 6;
        const int nt = (int)(t % NTtot), ks = (int)(t / NTtot);
        const int kk = 4 * ks + (lane >> 4), col = nt * 16 + (lane & 15);
        const int tap = kk / Kdim, kc = kk - tap * Kdim;
        float v = 0.0f;
        if (tap < K3 && col < Ndim) {
            const int co = backward ? kc : col, ci = backward ? col : kc;
            v = w[((long)co * Cin + ci) * K3 + tap];
        }
        wop[e] = v;
    }
}

// KSPLIT = false: a block is 4 row tiles (one per wave), every wave walks all k-steps.  KSPLIT = true (few row tiles: the deep levels): a block is ONE
// row tile, wave w walks the k-steps w, w + 4, ... and the four partial tiles are summed through LDS -- a 4x shorter dependent chain per wave.
// The k loop is software-pipelined by hand: U gathers and U * NT weight loads are issued before the U * NT MFMAs that consume them.
#define VX_CM_U 8
template <int NT, bool KSPLIT>
__global__ void __launch_bounds__(256) vx_conv_mfma_fwd_k(VxCm P) {
    extern __shared__ int vx_cm_tab[];            // [2][4 * ksteps]: element offset relative to the window origin, packed (dz, dy, dx); then the K-split staging
    const int Ktot = P.Cin * P.K * P.K * P.K;
    const int nk = 4 * P.ksteps;
    int* __restrict__ koff = vx_cm_tab;
    int* __restrict__ kd = vx_cm_tab + nk;
    const long Vi = (long)P.Di * P.Hi * P.Wi, Vo = (long)P.Do * P.Ho * P.Wo;
    for (int kk = threadIdx.x; kk < nk; kk += 256) {
        const int tap = kk / P.Cin, ci = kk - tap * P.Cin;
        const int dz = tap / (P.K * P.K), dy = (tap / P.K) % P.K, dx = tap % P.K;
        koff[kk] = kk < Ktot ? (int)(ci * Vi + ((long)dz * P.Hi + dy) * P.Wi + dx) : 0;
        kd[kk] = kk < Ktot ? (dz | (dy << 8) | (dx << 16)) : -1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const long Mtot = (long)P.B * Vo;
    const long tile0 = KSPLIT ? (long)blockIdx.x * 16 : ((long)blockIdx.x * 4 + wave) * 16;
    const bool tile_ok = tile0 < Mtot;            // (no early return: the K-split epilogue has a block barrier)
    const long m = tile0 + row;
    const bool mok = tile_ok && m < Mtot;
    const long mm = mok ? m : Mtot - 1;
    const int b = (int)(mm / Vo);
    const int o = (int)(mm - (long)b * Vo);
    const int ox = o % P.Wo, oy = (o / P.Wo) % P.Ho, oz = o / (P.Wo * P.Ho);
    const int iz0 = oz * P.S - P.P, iy0 = oy * P.S - P.P, ix0 = ox * P.S - P.P;
    const float* __restrict__ xb = P.x + (long)b * P.Cin * Vi + ((long)iz0 * P.Hi + iy0) * P.Wi + ix0;
    const int nt0 = blockIdx.y * NT;
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ wp = P.wop + (long)nt0 * 64 + lane;
    const long wstride = (long)P.NTtot * 64;
    const int kfirst = KSPLIT ? wave : 0, kstride = KSPLIT ? 4 : 1;
    if (tile_ok)
        for (int ks0 = kfirst; ks0 < P.ksteps; ks0 += VX_CM_U * kstride) {
            float a[VX_CM_U], bq[VX_CM_U][NT];
#pragma unroll
            for (int u = 0; u < VX_CM_U; ++u) {
                const int ks = ks0 + u * kstride;
                const bool kv = ks < P.ksteps;
                const int kk = kv ? 4 * ks + kq : 0;
                const int d = kv ? kd[kk] : -1;
                const int dz = d & 255, dy = (d >> 8) & 255, dx = (d >> 16) & 255;
                const bool ok = mok && d >= 0 && (unsigned)(iz0 + dz) < (unsigned)P.Di && (unsigned)(iy0 + dy) < (unsigned)P.Hi && (unsigned)(ix0 + dx) < (unsigned)P.Wi;
                a[u] = ok ? xb[koff[kk]] : 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) bq[u][t] = (kv && nt0 + t < P.NTtot) ? wp[(long)ks * wstride + t * 64] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < VX_CM_U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bq[u][t], acc[t], 0, 0, 0);
        }
    // D register i of lane l = (row 4 * (l / 16) + i, column l % 16)
    if (KSPLIT) {
        float* __restrict__ red = reinterpret_cast<float*>(vx_cm_tab + 2 * nk);      // [4 waves][NT][4][64]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave * NT + t) * 4 + i) * 64 + lane] = acc[t][i];
        __syncthreads();
        if (!tile_ok) return;
        const int i = wave;                          // wave w finishes register i = w of every lane: rows 4 * kq + w
        const long mr = tile0 + 4 * kq + i;
        if (mr >= Mtot) return;
        const int br = (int)(mr / Vo);
        const long orr = mr - (long)br * Vo;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = (nt0 + t) * 16 + row;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * NT + t) * 4 + i) * 64 + lane];
            if (co < P.Cout) P.y[((long)br * P.Cout + co) * Vo + orr] = v + (P.bias ? P.bias[co] : 0.0f);
        }
        return;
    }
    if (!tile_ok) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long mr = tile0 + 4 * kq + i;
        if (mr >= Mtot) continue;
        const int br = (int)(mr / Vo);
        const long orr = mr - (long)br * Vo;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int co = (nt0 + t) * 16 + row;
            if (co < P.Cout) P.y[((long)br * P.Cout + co) * Vo + orr] = acc[t][i] + (P.bias ? P.bias[co] : 0.0f);
        }
    }
}

// input gradient: blockIdx.y = stride-parity class (pz, py, px); the class's voxels are i = S * j + p, j over (Di/S, Hi/S, Wi/S)
template <int NT, bool KSPLIT>
__global__ void __launch_bounds__(256) vx_conv_mfma_bwd_data_k(VxCm P) {
    extern __shared__ int vx_cm_tab[];            // K-split staging only
    const int S = P.S, K = P.K;
    const int cls = blockIdx.y;
    const int px = cls % S, py = (cls / S) % S, pz = cls / (S * S);
    const int Dj = P.Di / S, Hj = P.Hi / S, Wj = P.Wi / S;
    const long Vj = (long)Dj * Hj * Wj, Vi = (long)P.Di * P.Hi * P.Wi, Vo = (long)P.Do * P.Ho * P.Wo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kq = lane >> 4;
    const long Mtot = (long)P.B * Vj;
    const long tile0 = KSPLIT ? (long)blockIdx.x * 16 : ((long)blockIdx.x * 4 + wave) * 16;
    const bool tile_ok = tile0 < Mtot;
    const long m = tile0 + row;
    const bool mok = tile_ok && m < Mtot;
    const long mm = mok ? m : Mtot - 1;
    const int b = (int)(mm / Vj);
    const int j = (int)(mm - (long)b * Vj);
    const int jx = j % Wj, jy = (j / Wj) % Hj, jz = j / (Wj * Hj);
    const int iz = jz * S + pz, iy = jy * S + py, ix = jx * S + px;
    const float* __restrict__ dyb = P.x + (long)b * P.Cout * Vo;
    const int nt0 = blockIdx.z * NT;
    v4f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    const long wstride = (long)P.NTtot * 64;
    const int cs = P.Cout / 4;                       // k-steps per tap (Cout % 4 == 0)
    const int cfirst = KSPLIT ? wave : 0, cstride = KSPLIT ? 4 : 1;
    if (tile_ok)
        for (int dz = 0; dz < K; ++dz) {
            if ((pz + P.P - dz) % S != 0) continue;      // only the taps that reach this parity class
            const int oz = (iz + P.P - dz) / S;
            for (int dy = 0; dy < K; ++dy) {
                if ((py + P.P - dy) % S != 0) continue;
                const int oy = (iy + P.P - dy) / S;
                for (int dx = 0; dx < K; ++dx) {
                    if ((px + P.P - dx) % S != 0) continue;
                    const int ox = (ix + P.P - dx) / S;
                    const bool ok = mok && iz + P.P - dz >= 0 && iy + P.P - dy >= 0 && ix + P.P - dx >= 0 && oz < P.Do && oy < P.Ho && ox < P.Wo;
                    const float* __restrict__ src = dyb + ((long)oz * P.Ho + oy) * P.Wo + ox;
                    const int tap = (dz * K + dy) * K + dx;
                    const float* __restrict__ wp = P.wop + ((long)tap * cs * P.NTtot + nt0) * 64 + lane;
                    for (int c0 = cfirst; c0 < cs; c0 += VX_CM_U * cstride) {
                        float a[VX_CM_U], bq[VX_CM_U][NT];
#pragma unroll
                        for (int u = 0; u < VX_CM_U; ++u) {
                            const int c4 = c0 + u * cstride;
                            const bool kv = c4 < cs;
                            a[u] = (ok && kv) ? src[(long)(4 * c4 + kq) * Vo] : 0.0f;
#pragma unroll
                            for (int t = 0; t < NT; ++t) bq[u][t] = (kv && nt0 + t < P.NTtot) ? wp[(long)c4 * wstride + t * 64] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < VX_CM_U; ++u)
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bq[u][t], acc[t], 0, 0, 0);
                    }
                }
            }
        }
    auto store_row = [&](long mr, int t, float v) {
        const int br = (int)(mr / Vj);
        const int jr = (int)(mr - (long)br * Vj);
        const int rx = jr % Wj, ry = (jr / Wj) % Hj, rz = jr / (Wj * Hj);
        const long vi = ((long)(rz * S + pz) * P.Hi + (ry * S + py)) * P.Wi + (rx * S + px);
        const int ci = (nt0 + t) * 16 + row;
        if (ci < P.Cin) {
            float* dst = P.y + ((long)br * P.Cin + ci) * Vi + vi;
            *dst = P.accumulate ? *dst + v : v;
        }
    };
    if (KSPLIT) {
        float* __restrict__ red = reinterpret_cast<float*>(vx_cm_tab);      // [4 waves][NT][4][64]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave * NT + t) * 4 + i) * 64 + lane] = acc[t][i];
        __syncthreads();
        if (!tile_ok) return;
        const int i = wave;
        const long mr = tile0 + 4 * kq + i;
        if (mr >= Mtot) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[((w * NT + t) * 4 + i) * 64 + lane];
            store_row(mr, t, v);
        }
        return;
    }
    if (!tile_ok) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long mr = tile0 + 4 * kq + i;
        if (mr >= Mtot) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) store_row(mr, t, acc[t][i]);
    }
}


// ------------------------------------------------------------------------------------------------------------------------------- host
extern "C" int vx_conv_mfma_ok(int Cin, int Cout, int D, int H, int W, int K, int S, int P, int G, int ps) {
    if (G != 1 || ps != 1 || K > 7 || S < 2 || S > 4 || P != K / 2 || (K & 1) == 0) return 0;
    if (D % S || H % S || W % S) return 0;
    if (Cout % 4 != 0 || Cin < 1 || (long)Cin * K * K * K > 8192) return 0;
    return 1;
}

static int cm_ksteps_fwd(int Cin, int K) { return (Cin * K * K * K + 3) / 4; }

extern "C" int vx_conv_mfma_ws_floats(int Cin, int Cout, int K, int backward) {
    const int K3 = K * K * K;
    if (backward) return K3 * (Cout / 4) * ((Cin + 15) / 16) * 64;
    return cm_ksteps_fwd(Cin, K) * ((Cout + 15) / 16) * 64;
}

extern "C" int vx_conv_mfma_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int B, int Cin, int D, int H, int W, int Cout, int K, int S, int P,
                                void* stream) {
    VX_REQUIRE(x && w && y && ws && B > 0, "vx_conv_mfma_fwd: null argument");
    VX_REQUIRE(vx_conv_mfma_ok(Cin, Cout, D, H, W, K, S, P, 1, 1), "vx_conv_mfma_fwd: unsupported shape Cin=%d Cout=%d %dx%dx%d K=%d S=%d P=%d", Cin, Cout, D, H, W, K, S, P);
    hipStream_t st = (hipStream_t)stream;
    VxCm A;
    A.x = x; A.bias = bias; A.y = y; A.wop = ws;
    A.B = B; A.Cin = Cin; A.Di = D; A.Hi = H; A.Wi = W; A.Cout = Cout;
    A.Do = (D + 2 * P - K) / S + 1; A.Ho = (H + 2 * P - K) / S + 1; A.Wo = (W + 2 * P - K) / S + 1;
    A.K = K; A.S = S; A.P = P; A.accumulate = 0;
    A.ksteps = cm_ksteps_fwd(Cin, K); A.NTtot = (Cout + 15) / 16;
    const long nw = (long)A.ksteps * A.NTtot * 64;
    vx_conv_mfma_wprep_k<<<dim3(vx_cdiv(nw, 256) > 1024 ? 1024 : vx_cdiv(nw, 256)), dim3(256), 0, st>>>(w, ws, Cout, Cin, K * K * K, A.ksteps, A.NTtot, 0);
    const long Mtot = (long)B * A.Do * A.Ho * A.Wo;
    // few row tiles (deep levels): one column tile per block, and below ~256 blocks the four waves of a block split K instead of the rows
    // K is split on the geometry of a NOMINAL batch of 4, so that a sample's sums are folded in the same order whatever it is batched with
    const long Mnom = 4L * A.Do * A.Ho * A.Wo;
    int NT = A.NTtot >= 4 ? 4 : (A.NTtot >= 2 ? 2 : 1);
    while (NT > 1 && (long)vx_cdiv(Mnom, 64) * vx_cdiv(A.NTtot, NT) < 256) NT >>= 1;
    const bool ksplit = (long)vx_cdiv(Mnom, 64) * vx_cdiv(A.NTtot, NT) < 256;
    if (ksplit) NT = 1;
    const int gx = ksplit ? vx_cdiv(Mtot, 16) : vx_cdiv(Mtot, 64);
    const size_t shm = sizeof(int) * 8 * (size_t)A.ksteps + (ksplit ? sizeof(float) * 4 * NT * 4 * 64 : 0);
    VX_REQUIRE(shm <= 64 * 1024, "vx_conv_mfma_fwd: k table does not fit LDS");
    const dim3 grid(gx, vx_cdiv(A.NTtot, NT));
    if (ksplit) vx_conv_mfma_fwd_k<1, true><<<grid, dim3(256), shm, st>>>(A);       // (NT is 1 whenever K is split)
    else if (NT == 4) vx_conv_mfma_fwd_k<4, false><<<grid, dim3(256), shm, st>>>(A);
    else if (NT == 2) vx_conv_mfma_fwd_k<2, false><<<grid, dim3(256), shm, st>>>(A);
    else vx_conv_mfma_fwd_k<1, false><<<grid, dim3(256), shm, st>>>(A);
    VX_LAUNCH_CHECK("vx_conv_mfma_fwd");
    return 0;
}

extern "C" int vx_conv_mfma_bwd_data(const float* dy, const float* w, float* dx, float* ws, int B, int Cin, int D, int H, int W, int Cout, int K, int S, int P, int accumulate,
                                     void* stream) {
    VX_REQUIRE(dy && w && dx && ws && B > 0, "vx_conv_mfma_bwd_data: null argument");
    VX_REQUIRE(vx_conv_mfma_ok(Cin, Cout, D, H, W, K, S, P, 1, 1), "vx_conv_mfma_bwd_data: unsupported shape Cin=%d Cout=%d %dx%dx%d K=%d S=%d P=%d", Cin, Cout, D, H, W, K, S, P);
    hipStream_t st = (hipStream_t)stream;
    VxCm A;
    A.x = dy; A.bias = nullptr; A.y = dx; A.wop = ws;
    A.B = B; A.Cin = Cin; A.Di = D; A.Hi = H; A.Wi = W; A.Cout = Cout;
    A.Do = (D + 2 * P - K) / S + 1; A.Ho = (H + 2 * P - K) / S + 1; A.Wo = (W + 2 * P - K) / S + 1;
    A.K = K; A.S = S; A.P = P; A.accumulate = accumulate;
    A.ksteps = K * K * K * (Cout / 4); A.NTtot = (Cin + 15) / 16;
    const long nw = (long)A.ksteps * A.NTtot * 64;
    vx_conv_mfma_wprep_k<<<dim3(vx_cdiv(nw, 256) > 1024 ? 1024 : vx_cdiv(nw, 256)), dim3(256), 0, st>>>(w, ws, Cout, Cin, K * K * K, A.ksteps, A.NTtot, 1);
    const long Mtot = (long)B * (D / S) * (H / S) * (W / S);
    const int ncls = S * S * S;
    const long Mnom = 4L * (D / S) * (H / S) * (W / S);       // (nominal batch of 4: see vx_conv_mfma_fwd)
    int NT = A.NTtot >= 4 ? 4 : (A.NTtot >= 2 ? 2 : 1);
    while (NT > 1 && (long)vx_cdiv(Mnom, 64) * ncls * vx_cdiv(A.NTtot, NT) < 256) NT >>= 1;
    const bool ksplit = (long)vx_cdiv(Mnom, 64) * ncls * vx_cdiv(A.NTtot, NT) < 256;
    if (ksplit) NT = 1;
    const int gx = ksplit ? vx_cdiv(Mtot, 16) : vx_cdiv(Mtot, 64);
    const size_t shm = ksplit ? sizeof(float) * 4 * NT * 4 * 64 : 0;
    const dim3 grid(gx, ncls, vx_cdiv(A.NTtot, NT));
    if (ksplit) vx_conv_mfma_bwd_data_k<1, true><<<grid, dim3(256), shm, st>>>(A);
    else if (NT == 4) vx_conv_mfma_bwd_data_k<4, false><<<grid, dim3(256), shm, st>>>(A);
    else if (NT == 2) vx_conv_mfma_bwd_data_k<2, false><<<grid, dim3(256), shm, st>>>(A);
    else vx_conv_mfma_bwd_data_k<1, false><<<grid, dim3(256), shm, st>>>(A);
    VX_LAUNCH_CHECK("vx_conv_mfma_bwd_data");
    return 0;
}
