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
#include <stdlib.h>
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


// ---------------------------------------------------------------------------------------------------------------------------- the stem on the f16 pipe
// k = 7, s = 4, p = 3, Cout = 16 (DownConv 1 of the conv encoder: Encoder.py:29-58): the gather kernel above re-reads every input voxel ~5 times from L2 with 4-byte
// loads (360 MB of gather traffic for a 67 MB input: 109 us).  Here a block stages the input rows of (1 x 4 x 16) outputs in LDS ONCE, as two fp16 pieces of x * 2^ex
// (ex from the tile's maximum), and the product is a Toeplitz GEMM along W:  rows = the 16 output channels, columns = 16 outputs along W, reduction = (ci, kd, kh) rows
// x 8 kw slots (7 taps + one zero weight): a lane's 8 reduction values are the 8 CONSECUTIVE inputs 4 ow - 3 .. 4 ow + 4 of one input row -- two ds_read_b64 (the address
// is 8-byte aligned) -- and four (ci, kd, kh) rows fill a v_mfma_f32_16x16x32_f16.  Weight images (two pieces of w * 2^ew) in operand order, prepared per launch.
typedef _Float16 cm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cm_h2 __attribute__((ext_vector_type(2)));
typedef float cm_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t cm_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cm_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const cm_f2 v = {a, b};
    const cm_h2 h = __builtin_convertvector(v, cm_h2);
    const cm_h2 l = __builtin_convertvector(v - __builtin_convertvector(h, cm_f2), cm_h2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ int cm_exp16(float m) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
    int e = 13 - ilogbf(m);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
#define VX_STEM_RL 72          // halfs per staged input row (67 used)
// img[(step * 2 + piece) * 64 + lane]: lane (co = lane & 15, G = lane >> 4): row r = 4 step + G = (ci * 7 + kd) * 7 + kh, the 8 halfs = w[co][ci][kd][kh][0..6], 0
__global__ void __launch_bounds__(1024) vx_stem_wprep_k(const float* __restrict__ w, cm_u4* __restrict__ img, float* __restrict__ esc, int Cin, int nsteps, unsigned* __restrict__ xmax) {
    __shared__ float sm[16];
    if (xmax != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *xmax = 0u;          // (the stem kernel that follows on this stream folds max |x| into it)
    const int n = 16 * Cin * 343;
    float mx = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, fabsf(w[i]));
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    float m2 = sm[threadIdx.x & 15];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
    const int ew = cm_exp16(m2);
    if (threadIdx.x == 0 && blockIdx.x == 0) esc[0] = (float)ew;
    const float sc = ldexpf(1.0f, ew);
    // one block per reduction step (every block finds the tensor's maximum itself: 22 K floats from L2)
    for (int t = blockIdx.x * 64 + threadIdx.x; t < (blockIdx.x + 1) * 64 && threadIdx.x < 64; t += 1024) {
        const int lane = t & 63, step = t >> 6;
        const int co = lane & 15, r = 4 * step + (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r < Cin * 49 && j < 7) ? w[((long)co * Cin * 49 + r) * 7 + j] * sc : 0.0f;
        uint32_t h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cm_split2(v[2 * q], v[2 * q + 1], h[q], l[q]);
        img[(step * 2) * 64 + lane] = (cm_u4){h[0], h[1], h[2], h[3]};
        img[(step * 2 + 1) * 64 + lane] = (cm_u4){l[0], l[1], l[2], l[3]};
    }
}
// block = (b, od, 4 output rows oh0.., 16 outputs ow0..); wave = one output row
// NP = operand pieces: 2 = every fp32 product from two scaled fp16 pieces (three MFMAs per step: the fp32 mode); 1 (round 6, the bf16 mode: reference speed_test.py:122,127
// autocast) = plain fp16 operands scaled per block / per tensor -- 11 significant bits against bf16's 8 --, ONE MFMA per step and half the LDS (4 input channels: 77 instead
// of 153 KB, two blocks per CU instead of one)
template <int CIN, int NP = 2>
__global__ void __launch_bounds__(256) vx_stem_fwd_k(const float* __restrict__ x, const cm_u4* __restrict__ img, const float* __restrict__ esc, const float* __restrict__ bias,
                                                     float* __restrict__ y, int B, int Di, int Hi, int Wi, int Do, int Ho, int Wo, unsigned* __restrict__ xmax) {
    constexpr int NROW = CIN * 7 * 19, NSTEP = (CIN * 49 + 3) / 4, NIT = (NROW * (VX_STEM_RL / 4) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_lds[];
    _Float16* __restrict__ xh = reinterpret_cast<_Float16*>(cm_lds);                   // [NROW][72] hi pieces, then (NP = 2) the lo pieces
    _Float16* __restrict__ xl = xh + (NP == 2 ? NROW * VX_STEM_RL : 0);
    int* __restrict__ rowoff = reinterpret_cast<int*>(xh + NP * NROW * VX_STEM_RL);    // [4 NSTEP]: staged row of reduction row r (for output row 0), in halfs
    float* __restrict__ red = reinterpret_cast<float*>(rowoff + 4 * NSTEP);            // [4]
    const int nwb = (Wo + 15) / 16, nhb = Ho / 4;              // (a ragged last tile along W -- 24 outputs per row at 96^3 -- stores only its valid columns)
    int t = blockIdx.x;
    if ((gridDim.x & 7) == 0) t = (t & 7) * (int)(gridDim.x >> 3) + (t >> 3);          // (round 6) consecutive block ids go to different XCDs: every XCD walks a contiguous run of tiles, whose 7^3 halos overlap in ITS L2
    const int wb = t % nwb; t /= nwb;
    const int hb = t % nhb; t /= nhb;
    const int od = t % Do;
    const int b = t / Do;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, G = lane >> 4;
    for (int r = threadIdx.x; r < 4 * NSTEP; r += 256) {
        const int ci = r / 49, kd = (r / 7) % 7, kh = r % 7;
        rowoff[r] = r < CIN * 49 ? ((ci * 7 + kd) * 19 + kh) * VX_STEM_RL : 0;
    }
    // ---- stage: item = (staged row, 4 consecutive inputs); the rows start at input w = 64 wb - 3
    const long Vi = (long)Di * Hi * Wi;
    const float* __restrict__ xb = x + (long)b * CIN * Vi;
    float v[NIT][4];
    float mx = 0.0f;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int it = threadIdx.x + u * 256;
        const int row = it / (VX_STEM_RL / 4), seg = it - row * (VX_STEM_RL / 4);
        const int ci = row / (7 * 19), hd = (row / 19) % 7, hh = row % 19;
        const int id = 4 * od - 3 + hd, ih = 16 * hb - 3 + hh, iw0 = 64 * wb - 3 + 4 * seg;
        const bool rok = it < NROW * (VX_STEM_RL / 4) && (unsigned)id < (unsigned)Di && (unsigned)ih < (unsigned)Hi;
        const float* __restrict__ src = xb + (long)ci * Vi + ((long)(rok ? id : 0) * Hi + (rok ? ih : 0)) * Wi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = iw0 + j;
            const float t_ = src[(unsigned)iw < (unsigned)Wi ? iw : 0];
            v[u][j] = (rok && (unsigned)iw < (unsigned)Wi) ? t_ : 0.0f;
            mx = fmaxf(mx, fabsf(v[u][j]));
        }
    }
    mx = vx_wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float bmx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // (round 6) max |x| of the whole input as a by-product (every input is staged by some block; non-negative floats order like unsigned integers): the weight
    // gradient of this layer scales its operands by it and no longer reads the 67 MB input a second time to find it
    if (xmax != nullptr && threadIdx.x == 0 && bmx > 0.0f) atomicMax(xmax, __float_as_uint(bmx));
    const int ex = cm_exp16(bmx);
    const float sc = ldexpf(1.0f, ex);
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int it = threadIdx.x + u * 256;
        if (it < NROW * (VX_STEM_RL / 4)) {
            uint32_t h0, l0, h1, l1;
            cm_split2(v[u][0] * sc, v[u][1] * sc, h0, l0);
            cm_split2(v[u][2] * sc, v[u][3] * sc, h1, l1);
            *reinterpret_cast<uint2*>(xh + (long)it * 4) = make_uint2(h0, h1);
            if constexpr (NP == 2) *reinterpret_cast<uint2*>(xl + (long)it * 4) = make_uint2(l0, l1);
        }
    }
    __syncthreads();
    // ---- this wave's output row oh = 4 hb + wave: lane (output ow = 16 wb + n, rows of step s: 4 s + G)
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const int base = (4 * wave) * VX_STEM_RL + 4 * n;                                 // (+ rowoff[r]): first of the lane's 8 inputs
    const cm_u4* __restrict__ ig = img + lane;
    cm_u4 ah = ig[0], al = ig[64];
#pragma unroll 2
    for (int s = 0; s < NSTEP; ++s) {
        cm_u4 nh = ah, nl = al;
        if (s + 1 < NSTEP) { nh = ig[(s + 1) * 128]; if constexpr (NP == 2) nl = ig[(s + 1) * 128 + 64]; }
        const int o = rowoff[4 * s + G] + base;
        const uint2 b0 = *reinterpret_cast<const uint2*>(xh + o), b1 = *reinterpret_cast<const uint2*>(xh + o + 4);
        const cm_u4 bh = {b0.x, b0.y, b1.x, b1.y};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cm_h8, ah), __builtin_bit_cast(cm_h8, bh), acc, 0, 0, 0);
        if constexpr (NP == 2) {
            const uint2 c0 = *reinterpret_cast<const uint2*>(xl + o), c1 = *reinterpret_cast<const uint2*>(xl + o + 4);
            const cm_u4 bl = {c0.x, c0.y, c1.x, c1.y};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cm_h8, ah), __builtin_bit_cast(cm_h8, bl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cm_h8, al), __builtin_bit_cast(cm_h8, bh), acc, 0, 0, 0);
        }
        ah = nh; al = nl;
    }
    const float f = ldexpf(1.0f, -(ex + (int)esc[0]));
    const long Vo = (long)Do * Ho * Wo;
    const long o0 = ((long)od * Ho + 4 * hb + wave) * Wo + 16 * wb + n;
    if (16 * wb + n < Wo) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = 4 * G + i;
            y[((long)b * 16 + co) * Vo + o0] = fmaf(acc[i], f, bias ? bias[co] : 0.0f);
        }
    }
}
// ON by default (VELOXSEG_STEM_F16=0 / vx_conv_mfma_set_stem_f16(0): the fp32 gather kernel; 2 = also 4-channel inputs).  Round 4 shipped it switched off: with it the
// taped step showed one deviating replay in ~150.  Round 5 found why: not this kernel and not the tape -- its v_mfma_f32_16x16x32_f16 issues
// trigger a gfx950 register-read hazard in packed-fp32 instructions of WHATEVER kernel shares the SIMD (here the bias add of vx_ln_pw_fwd_k on the other lane);
// veloxseg_amd/_isa_fix.py removes the hazardous instruction form from every kernel of the library at build time.
static int vx_stem_f16 = -1;
static int vx_stem_mode() {
    if (vx_stem_f16 < 0) { const char* e = getenv("VELOXSEG_STEM_F16"); vx_stem_f16 = e ? atoi(e) : 1; if (vx_stem_f16 < 0) vx_stem_f16 = 0; }
    return vx_stem_f16;
}
extern "C" int vx_conv_mfma_set_stem_f16(int on) { vx_stem_f16 = on < 0 ? 0 : on; return 0; }      // A/B knob: the stem (k7 s4, 16 channels) on the f16 pipe or the fp32 gather kernel (default)
// plain fp16 operands (one piece) for the stem forward: the bf16 mode (functional.set_precision("bf16") -> vx_conv_mfma_set_stem_pieces(1)); 2 = the fp32 mode
static int g_stem_pieces = 2;
extern "C" int vx_conv_mfma_set_stem_pieces(int np) { if (np != 1 && np != 2) return -1; g_stem_pieces = np; return 0; }
extern "C" int vx_conv_mfma_stem_pieces(void) { return g_stem_pieces; }
static bool vx_stem_ok(int Cin, int Cout, int D, int H, int W, int K, int S, int P) {
    // (4 input channels: with one piece the block needs 77 KB of LDS and the kernel is selected; with two, 153 KB = one block per CU, it is not faster than the gather kernel)
    return vx_stem_mode() && K == 7 && S == 4 && P == 3 && Cout == 16 && (Cin == 1 || Cin == 2 || (Cin == 4 && (vx_stem_mode() > 1 || g_stem_pieces == 1))) && W % 64 == 0 && H % 16 == 0 && D % 4 == 0;      // (4 input channels = 153 KB of LDS, one block per CU: brats128 B = 4 851 vs 864 patches/s, off unless the knob is 2)      // (rows of 24 outputs -- the 96^3 patches -- measured no gain: 1207 vs 1216, 724 vs 731 patches/s)
}
extern "C" int vx_conv_mfma_ws_floats(int Cin, int Cout, int K, int backward) {
    const int K3 = K * K * K;
    if (backward) return K3 * (Cout / 4) * ((Cin + 15) / 16) * 64;
    const int n = cm_ksteps_fwd(Cin, K) * ((Cout + 15) / 16) * 64;
    const int stem = (K == 7 && Cout == 16) ? ((Cin * 49 + 3) / 4) * 2 * 64 * 4 + 4 : 0;          // (the f16-pipe stem: two-piece images + the scale exponent)
    return n > stem ? n : stem;
}

extern "C" int vx_conv_mfma_fwd_mx(const float* x, const float* w, const float* bias, float* y, float* ws, unsigned* x_absmax, int B, int Cin, int D, int H, int W, int Cout,
                                   int K, int S, int P, void* stream);
extern "C" int vx_conv_mfma_fwd(const float* x, const float* w, const float* bias, float* y, float* ws, int B, int Cin, int D, int H, int W, int Cout, int K, int S, int P,
                                void* stream) {
    return vx_conv_mfma_fwd_mx(x, w, bias, y, ws, nullptr, B, Cin, D, H, W, Cout, K, S, P, stream);
}
// 1 when vx_conv_mfma_fwd_mx with these arguments writes max |x| (the f16-pipe stem kernel takes the layer)
extern "C" int vx_conv_mfma_fwd_writes_absmax(int Cin, int Cout, int D, int H, int W, int K, int S, int P) {
    return (vx_conv_mfma_ok(Cin, Cout, D, H, W, K, S, P, 1, 1) && vx_stem_ok(Cin, Cout, D, H, W, K, S, P)) ? 1 : 0;
}
// x_absmax (may be null; one unsigned, device): where vx_conv_mfma_fwd_writes_absmax says so, the bits of max |x| are left there for vx_down_wgrad_mfma_mx
extern "C" int vx_conv_mfma_fwd_mx(const float* x, const float* w, const float* bias, float* y, float* ws, unsigned* x_absmax, int B, int Cin, int D, int H, int W, int Cout,
                                   int K, int S, int P, void* stream) {
    VX_REQUIRE(x && w && y && ws && B > 0, "vx_conv_mfma_fwd: null argument");
    VX_REQUIRE(vx_conv_mfma_ok(Cin, Cout, D, H, W, K, S, P, 1, 1), "vx_conv_mfma_fwd: unsupported shape Cin=%d Cout=%d %dx%dx%d K=%d S=%d P=%d", Cin, Cout, D, H, W, K, S, P);
    hipStream_t st = (hipStream_t)stream;
    if (vx_stem_ok(Cin, Cout, D, H, W, K, S, P)) {
        const int nsteps = (Cin * 49 + 3) / 4, Do = D / 4, Ho = H / 4, Wo = W / 4;
        cm_u4* img = reinterpret_cast<cm_u4*>(ws);
        float* esc = ws + (long)nsteps * 2 * 64 * 4;
        vx_stem_wprep_k<<<dim3((unsigned)nsteps), dim3(1024), 0, st>>>(w, img, esc, Cin, nsteps, x_absmax);
        // (one piece only where it brings the f16-pipe kernel in at all -- 4 input channels: brats128 bf16 816 -> 830 patches/s; with 2 channels the two-piece kernel stays:
        //  autopet128 bf16 1156 vs 1151 with one piece -- its shorter blocks only crowd the other lanes, the stem is not on that step's critical path)
        const int np = (g_stem_pieces == 1 && Cin == 4) ? 1 : 2;
        const size_t shm = (size_t)Cin * 7 * 19 * VX_STEM_RL * 2 * np + (size_t)4 * nsteps * 4 + 64;
        const dim3 grid((unsigned)((long)B * Do * (Ho / 4) * ((Wo + 15) / 16)));
#define VX_STEM(CI, NP_) { static bool once = false; if (!once) { if (hipFuncSetAttribute((const void*)vx_stem_fwd_k<CI, NP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); once = true; } \
        vx_stem_fwd_k<CI, NP_><<<grid, dim3(256), shm, st>>>(x, img, esc, bias, y, B, D, H, W, Do, Ho, Wo, x_absmax); }
        if (np == 1) { if (Cin == 1) VX_STEM(1, 1) else if (Cin == 2) VX_STEM(2, 1) else VX_STEM(4, 1) }
        else { if (Cin == 1) VX_STEM(1, 2) else if (Cin == 2) VX_STEM(2, 2) else VX_STEM(4, 2) }
#undef VX_STEM
        VX_LAUNCH_CHECK("vx_conv_mfma_fwd (stem, f16 pipe)");
        return 0;
    }
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
