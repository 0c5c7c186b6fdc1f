// JLC grouped convolutions on the bf16 matrix pipe with fp32-exact products (reference model/components/conv_blocks.py:51-58: three grouped Conv3d, k = 1, 3, 5,
// "same" padding, groups = C / min_dim_group, i.e. 4 / 8 / 8 / 16 channels per group at the four levels).
//
// Why a Toeplitz GEMM.  With 4..16 channels per group the implicit GEMM of one group is (4..16) x (4..16)*K^3: a 16 x 16 MFMA tile over (co, ci) is 6-25 % full.
// The W axis supplies the missing rows: for one (kd, kh) the convolution along W of 4 consecutive outputs is a 4 x 8 banded (Toeplitz) matrix
//     T[wo][j] = w[kd][kh][j - 2 - wo + K/2]   (0 <= . < K, else 0),     out[w0 + wo] += sum_j T[wo][j] * in[w0 - 2 + j],
// so   A[(co, wo)][(ci, j)]  (16 x 32, rows = 4 output channels x 4 outputs along W, k = 4 input channels x 8 inputs along W)
// and  B[(ci, j)][position]  (32 x 16, column = one of 16 (d, h, w-block) positions; a lane's 8 k-values are 8 CONSECUTIVE inputs along W of one channel:
//                             one 16-byte read of the LDS halo row)
// make v_mfma_f32_16x16x32_bf16 do 4 co x 4 ci x K taps x 4 outputs x 16 positions per issue: 62.5 % of its multipliers carry a product at K = 5 (37.5 % at K = 3).
// fp32 accuracy on the bf16 pipe as in expand_mfma.hip: every operand is x = x0 + x1 + x2 (bf16 pieces, 24 mantissa bits), a product is the six piece products of
// weight >= 2^-24, fp32 accumulate (NS = 3; NS = 1 is the bf16 opt-in mode).  Price: 6 x 16 clocks per 5120 useful MACs = 53 MAC/clk/SIMD against 32 for v_pk_fma_f32
// at its peak (the fp32 VALU kernels of jlc.hip reach 9).
//
// Operands.  The banded weight matrices are expanded ONCE per step by vx_jlc_tz_prep into operand-order images (one coalesced 16-byte load per lane, piece and
// entry; forward and input-gradient images: the adjoint of a "same" conv is a "same" conv with the weights transposed inside the group and flipped in space).
// The input halo is split into its pieces when it is staged (once per element, it is then read by up to 25 (kd, kh) entries).
//
// forward : one halo (K = 5) of x feeds all three convolutions; per-(b, c) partial (sum, sumsq) of every output in the epilogue, exactly as vx_jlc_conv_fwd.
// backward: dx = d_o + conv5^T(g5) + conv3^T(g3) + conv1^T(g1): g5 and g3 are staged one after the other (halo 2 / 1) into the same accumulators, the
//           1x1x1 term and the residual are added from global memory in the epilogue.
#include "vx_common.h"
#include <type_traits>
#include "../../include/veloxseg_hip.h"

typedef __bf16 tz_bf8 __attribute__((ext_vector_type(8)));
typedef float tz_f4 __attribute__((ext_vector_type(4)));

struct TzGeo {          // LDS halo geometry of one source (elements = bf16)
    int HD, HH;         // halo rows along D, H (= TD + 2 hw, TH + 2 hw)
    int Sd, Sc, Sp;     // strides: d-plane, channel plane, piece
    int nrows;          // CG * HD * HH
    unsigned mHH, mHD;  // magic reciprocals (exact for the index ranges used here)
    int qd0, nq;        // first quad / quads per row that this source stages (quad qd covers w0 - 4 + 4 qd .. + 3)
    unsigned mNQ;
};

struct VxTz {
    const void* src[3];             // forward: x (fp32); backward: g5, g3, g1 (fp32, or vx_bf16 when the kernel's H16 argument is set)
    const uint4* img;               // operand images of this direction
    const float* esc;               // fp16 two-piece mode: scale exponents of the weight images [G][3] (K = 5, 3, 1)
    const float* bias[3];           // forward (order k = 5, 3, 1); may be null
    const void* res;                // backward: d_o (fp32 / vx_bf16 as the g_k)
    const float* w1;                // backward: the 1x1x1 weights (C, C/G)
    void* out[3];                   // forward: y5, y3, y1 (fp32 / vx_bf16); backward: dx (fp32)
    double* part;                   // forward: [3][B*C][ntiles][2], k = 0 -> y1, 1 -> y3, 2 -> y5 (layout of vx_jlc_conv_fwd)
    int B, C, G, D, H, W;
    int TD, TH, TWB;                // tile: TD x TH x 4*TWB outputs
    int nTd, nTh, nTw;
    int RW;                         // halo row length (elements) = 4 TWB + 4
    int P, nNt;                     // positions per tile (TD*TH*TWB), N-tiles (16 positions each)
    unsigned mTWB, mTD;
    TzGeo geo[2];                   // [0]: halo 2 (forward / g5), [1]: halo 1 (g3)
    int red_off;                    // byte offset of the statistics scratch behind the halo
    int dbg;                        // timing experiments only (tools/jlc_tz_probe.py): bit 0 = skip the staging, bit 1 = skip the MFMA loops
};

__device__ __forceinline__ uint32_t tz_pack(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
// i / d through the precomputed reciprocal m = ceil(2^32 / d) (exact for i * d < 2^32; m == 0 encodes d == 1)
__device__ __forceinline__ unsigned tz_divm(unsigned i, unsigned m) { return m ? __umulhi(i, m) : i; }
__device__ __forceinline__ float tz_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float tz_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// ---- two fp16 pieces (the "22-bit" mode, pieces knob = 22): x * 2^e = h0 + h1 with e chosen per tile (activations) / per conv group (weights) so that the largest
// magnitude sits in [2^14, 2^15).  11 + 11 significant bits against the 24 of fp32; a product = THREE MFMAs (h0 k0 + h0 k1 + h1 k0, the neglected h1 k1 <= 2^-22 of the
// product), half the matrix-pipe work and two thirds of the LDS traffic of the three-bf16-piece scheme.  Small elements fall into fp16's subnormals: absolute error
// <= 2^-25 in scaled units = 2^-40 of the tile's maximum.  Power-of-two scales: exact, undone on the fp32 accumulators in the epilogue.
typedef _Float16 tz_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t tz_pack16(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float tz_lo16(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[0]; }
__device__ __forceinline__ float tz_hi16(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[1]; }
// exponent e with |m| * 2^e in [2^14, 2^15) (0 for m = 0 or a non-finite m: the values then pass through unscaled)
__device__ __forceinline__ int tz_exp16(float m) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
    int e = 14 - ilogbf(m);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}

// ------------------------------------------------------------------------------------------------------------------ operand images of the weights
// entry e: K = 5 -> kd*5 + kh (0..24); K = 3 -> 25 + kd*3 + kh; K = 1 -> 34.   img[((((g*35 + e)*MT + mt)*KS + ks)*NS + s)*64 + lane] (uint4 = 8 bf16)
template <int CG, int NS>
__global__ void __launch_bounds__(256) vx_tz_prep_k(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ w5, uint4* __restrict__ img_f,
                                                    uint4* __restrict__ img_b, int G) {
    constexpr int MT = CG / 4, KS = CG / 4;
    const long total = (long)G * 35 * MT * KS * 64;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long r = t >> 6;
    const int ks = (int)(r % KS); r /= KS;
    const int mt = (int)(r % MT); r /= MT;
    const int e = (int)(r % 35);
    const int g = (int)(r / 35);
    int K, kd, kh;
    const float* w;
    if (e < 25) { K = 5; kd = e / 5; kh = e % 5; w = w5; }
    else if (e < 34) { K = 3; kd = (e - 25) / 3; kh = (e - 25) % 3; w = w3; }
    else { K = 1; kd = 0; kh = 0; w = w1; }
    const int hw = K / 2, K3 = K * K * K;
    const int m = lane & 15, q = lane >> 4;
    const int a_l = mt * 4 + (m >> 2), wo = m & 3;      // row channel (local), output along W
    const int b_l = ks * 4 + q;                          // k channel (local)
    float vf[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kw = j - 2 - wo + hw;
        float f = 0.0f, bwd = 0.0f;
        if (kw >= 0 && kw < K) {
            // forward: row = output channel, k = input channel
            f = w[((long)(g * CG + a_l) * CG + b_l) * K3 + (kd * K + kh) * K + kw];
            // input gradient: row = INPUT channel of the forward conv, k = its output channel, taps flipped
            bwd = w[((long)(g * CG + b_l) * CG + a_l) * K3 + ((K - 1 - kd) * K + (K - 1 - kh)) * K + (K - 1 - kw)];
        }
        vf[j] = f; vb[j] = bwd;
    }
    const long o = (t >> 6) * NS * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t pf[4], pb[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            pf[j2] = tz_pack(vf[2 * j2], vf[2 * j2 + 1]);
            pb[j2] = tz_pack(vb[2 * j2], vb[2 * j2 + 1]);
            if (s + 1 < NS) {
                vf[2 * j2] -= tz_lo(pf[j2]); vf[2 * j2 + 1] -= tz_hi(pf[j2]);
                vb[2 * j2] -= tz_lo(pb[j2]); vb[2 * j2 + 1] -= tz_hi(pb[j2]);
            }
        }
        img_f[o + (long)s * 64] = make_uint4(pf[0], pf[1], pf[2], pf[3]);
        img_b[o + (long)s * 64] = make_uint4(pb[0], pb[1], pb[2], pb[3]);
    }
}

// scale exponents of the fp16 two-piece weight images: esc[g][c] (c = 0, 1, 2 for K = 5, 3, 1) from the largest magnitude of the (group, K) weight block (contiguous
// CG * CG * K^3 floats); grid (G, 3)
__global__ void __launch_bounds__(256) vx_tz_wmax_k(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ w5, float* __restrict__ esc, int CG) {
    __shared__ float sm[4];
    const int g = blockIdx.x, c = blockIdx.y;
    const int K3 = c == 0 ? 125 : c == 1 ? 27 : 1;
    const float* __restrict__ wg = (c == 0 ? w5 : c == 1 ? w3 : w1) + (long)g * CG * CG * K3;
    float mx = 0.0f;
    for (int i = threadIdx.x; i < CG * CG * K3; i += 256) mx = fmaxf(mx, fabsf(wg[i]));
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) esc[g * 3 + c] = (float)tz_exp16(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])));
}
// the same images as two scaled fp16 pieces (pieces knob = 22): one WAVE per image entry
template <int CG>
__global__ void __launch_bounds__(256) vx_tz_prep16_k(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ w5, uint4* __restrict__ img_f,
                                                      uint4* __restrict__ img_b, const float* __restrict__ esc, int G) {
    constexpr int MT = CG / 4, KS = CG / 4;
    const long total = (long)G * 35 * MT * KS;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= total) return;
    const int lane = threadIdx.x & 63;
    long r = t;
    const int ks = (int)(r % KS); r /= KS;
    const int mt = (int)(r % MT); r /= MT;
    const int e = (int)(r % 35);
    const int g = (int)(r / 35);
    int K, kd, kh, c;
    const float* w;
    if (e < 25) { K = 5; kd = e / 5; kh = e % 5; w = w5; c = 0; }
    else if (e < 34) { K = 3; kd = (e - 25) / 3; kh = (e - 25) % 3; w = w3; c = 1; }
    else { K = 1; kd = 0; kh = 0; w = w1; c = 2; }
    const int hw = K / 2, K3 = K * K * K;
    const int ew = (int)esc[g * 3 + c];
    const float sc = ldexpf(1.0f, ew);
    const int m = lane & 15, q = lane >> 4;
    const int a_l = mt * 4 + (m >> 2), wo = m & 3, b_l = ks * 4 + q;
    float vf[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kw = j - 2 - wo + hw;
        float f = 0.0f, bwd = 0.0f;
        if (kw >= 0 && kw < K) {
            f = w[((long)(g * CG + a_l) * CG + b_l) * K3 + (kd * K + kh) * K + kw] * sc;
            bwd = w[((long)(g * CG + b_l) * CG + a_l) * K3 + ((K - 1 - kd) * K + (K - 1 - kh)) * K + (K - 1 - kw)] * sc;
        }
        vf[j] = f; vb[j] = bwd;
    }
    const long o = t * 2 * 64 + lane;
    uint32_t f0[4], f1[4], b0[4], b1[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        f0[j2] = tz_pack16(vf[2 * j2], vf[2 * j2 + 1]);
        f1[j2] = tz_pack16(vf[2 * j2] - tz_lo16(f0[j2]), vf[2 * j2 + 1] - tz_hi16(f0[j2]));
        b0[j2] = tz_pack16(vb[2 * j2], vb[2 * j2 + 1]);
        b1[j2] = tz_pack16(vb[2 * j2] - tz_lo16(b0[j2]), vb[2 * j2 + 1] - tz_hi16(b0[j2]));
    }
    img_f[o] = make_uint4(f0[0], f0[1], f0[2], f0[3]);
    img_f[o + 64] = make_uint4(f1[0], f1[1], f1[2], f1[3]);
    img_b[o] = make_uint4(b0[0], b0[1], b0[2], b0[3]);
    img_b[o + 64] = make_uint4(b1[0], b1[1], b1[2], b1[3]);
}

// ------------------------------------------------------------------------------------------------------------------ halo staging
// Halo of one source in LDS as NS bf16 planes: elem[s][ci][hd][hh][e], e = w - (w0 - 2) in [0, RW).  A thread stages quads (4 consecutive w, one 16-byte load);
// the pieces of a quad leave as two packed pairs per piece (4-byte LDS stores: e = 4 qd - 2 is even, not a multiple of 4).
template <int CG, int NS, typename TS = float>
__device__ __forceinline__ void tz_stage(const TS* __restrict__ src, unsigned char* __restrict__ lds, const VxTz& p, const TzGeo& ge, int hw, int b, int g, int d0, int h0,
                                         int w0, int nthr) {
    const int total = ge.nrows * ge.nq;
    const long chan = (long)p.D * p.H * p.W;
    const TS* __restrict__ sb = src + ((long)b * p.C + (long)g * CG) * chan;
    constexpr int SU = 4;
    for (int it0 = threadIdx.x; it0 < total; it0 += nthr * SU) {
        float4 v[SU];
        int eo[SU];      // element offset of the quad's first pair (e = 4 qd - 2) inside piece 0, or < 0: nothing to store
        int qd_[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int it = it0 + u * nthr;
            const bool live = it < total;
            const unsigned iu = live ? (unsigned)it : 0u;
            const unsigned r1 = tz_divm(iu, ge.mNQ);
            const int qd = (int)(iu - r1 * ge.nq) + ge.qd0;
            const unsigned r2 = tz_divm(r1, ge.mHH);
            const int hh = (int)(r1 - r2 * ge.HH);
            const unsigned ci = tz_divm(r2, ge.mHD);
            const int hd = (int)(r2 - ci * ge.HD);
            const int id = d0 - hw + hd, ih = h0 - hw + hh, iw = w0 - 4 + 4 * qd;
            const bool ok = live && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const float4 t_ = vx_ld4(sb, ok ? (long)ci * chan + ((long)id * p.H + ih) * p.W + iw : 0L);
            v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
            eo[u] = live ? (int)ci * ge.Sc + hd * ge.Sd + hh * p.RW + 4 * qd - 2 : -(1 << 30);
            qd_[u] = qd;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (eo[u] <= -(1 << 29)) continue;
            float a0 = v[u].x, a1 = v[u].y, a2 = v[u].z, a3 = v[u].w;
            const bool first = qd_[u] > 0, second = qd_[u] * 4 < p.RW;       // quad 0 contributes its last two values only, the last quad its first two
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uint32_t lo = tz_pack(a0, a1), hi = tz_pack(a2, a3);
                uint32_t* dst = reinterpret_cast<uint32_t*>(lds + 2 * ((long)s * ge.Sp + eo[u]));
                if (first) dst[0] = lo;
                if (second) dst[1] = hi;
                if (s + 1 < NS) { a0 -= tz_lo(lo); a1 -= tz_hi(lo); a2 -= tz_lo(hi); a3 -= tz_hi(hi); }
            }
        }
    }
}

// fp16 two-piece staging: every quad of the thread is loaded FIRST (<= TZ_NQ16 per thread: host-checked), the tile's largest magnitude is reduced over the block, then
// the scaled values are split and stored.  Returns the tile's exponent (block-uniform).  `scratch`: >= 9 floats of LDS outside the halo.
#define TZ_NQ16 16
template <int CG>
__device__ __forceinline__ int tz_stage16(const float* __restrict__ src, unsigned char* __restrict__ lds, const VxTz& p, const TzGeo& ge, int hw, int b, int g, int d0, int h0,
                                          int w0, int nthr, float* __restrict__ scratch) {
    const int total = ge.nrows * ge.nq;
    const long chan = (long)p.D * p.H * p.W;
    const float* __restrict__ sb = src + ((long)b * p.C + (long)g * CG) * chan;
    float4 v[TZ_NQ16];
    int eo[TZ_NQ16];
    float mx = 0.0f;
#pragma unroll
    for (int u = 0; u < TZ_NQ16; ++u) {
        const int it = (int)threadIdx.x + u * nthr;
        const bool live = it < total;
        const unsigned iu = live ? (unsigned)it : 0u;
        const unsigned r1 = tz_divm(iu, ge.mNQ);
        const int qd = (int)(iu - r1 * ge.nq) + ge.qd0;
        const unsigned r2 = tz_divm(r1, ge.mHH);
        const int hh = (int)(r1 - r2 * ge.HH);
        const unsigned ci = tz_divm(r2, ge.mHD);
        const int hd = (int)(r2 - ci * ge.HD);
        const int id = d0 - hw + hd, ih = h0 - hw + hh, iw = w0 - 4 + 4 * qd;
        const bool ok = live && (unsigned)id < (unsigned)p.D && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const float4 t_ = *reinterpret_cast<const float4*>(sb + (ok ? (long)ci * chan + ((long)id * p.H + ih) * p.W + iw : 0));
        v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
        // bit 0 / 1 of the low bits: store the first / second pair (quad 0 contributes its last two values only, the last quad its first two)
        const int e_ = (int)ci * ge.Sc + hd * ge.Sd + hh * p.RW + 4 * qd - 2;
        eo[u] = live ? ((e_ + 2) << 2) | (qd > 0 ? 1 : 0) | (qd * 4 < p.RW ? 2 : 0) : -1;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
    }
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    {
        const int nw = nthr >> 6;
        float m2 = 0.0f;
        for (int i = 0; i < nw; ++i) m2 = fmaxf(m2, scratch[i]);
        mx = m2;
    }
    const int e = tz_exp16(mx);
    const float sc = ldexpf(1.0f, e);
#pragma unroll
    for (int u = 0; u < TZ_NQ16; ++u) {
        if (eo[u] < 0) continue;
        const int e0 = (eo[u] >> 2) - 2;
        const float a0 = v[u].x * sc, a1 = v[u].y * sc, a2 = v[u].z * sc, a3 = v[u].w * sc;
        const uint32_t lo0 = tz_pack16(a0, a1), hi0 = tz_pack16(a2, a3);
        const uint32_t lo1 = tz_pack16(a0 - tz_lo16(lo0), a1 - tz_hi16(lo0)), hi1 = tz_pack16(a2 - tz_lo16(hi0), a3 - tz_hi16(hi0));
        uint32_t* d0_ = reinterpret_cast<uint32_t*>(lds + 2 * (long)e0);
        uint32_t* d1_ = reinterpret_cast<uint32_t*>(lds + 2 * ((long)ge.Sp + e0));
        if (eo[u] & 1) { d0_[0] = lo0; d1_[0] = lo1; }
        if (eo[u] & 2) { d0_[1] = hi0; d1_[1] = hi1; }
    }
    return e;
}

// ------------------------------------------------------------------------------------------------------------------ accumulate one convolution
// acc[mt][nt] += sum over the K*K (kd, kh) entries and the KS k-steps.  `boff[nt]`: byte offset of the lane's position (and of its q-th channel) in piece 0;
// `ebase`: byte offset of entry (0, 0) (a K = 3 / 1 convolution reading a K = 5 halo starts one / two rows and planes in).
// Software pipeline: a step (entry, k-step) is cut into units of <= 2 N-tiles; the LDS reads of unit u + 1 are issued before the MFMAs of unit u (at most 12
// reads in flight: lgkmcnt counts to 15), the weight operands of step t + 1 (global, L2-resident) before the MFMAs of step t.
template <int NTB, int NS>
__device__ __forceinline__ void tz_read_b(uint4 (&bv)[NTB][NS], const unsigned char* __restrict__ lds, const int* boff, int off, int Sp) {
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // two 8-byte reads (the address is 8-, not 16-byte aligned: a misaligned ds_read_b128 is replayed at 64 cycles); volatile keeps the compiler from
            // fusing them into ds_read2_b64, which runs at half the rate of two ds_read_b64.  Explicit LDS address space: address-space inference skips
            // volatile accesses, a generic pointer would make these FLAT loads.
            typedef const volatile __attribute__((address_space(3))) unsigned long long* tz_lds_u64;
            tz_lds_u64 bp = (tz_lds_u64)(lds + boff[nt] + off + 2 * s * Sp);
            const unsigned long long lo = bp[0], hi = bp[1];
            bv[nt][s] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
        }
}
template <int NTB, int MTW, int NS, bool F16>
__device__ __forceinline__ void tz_mfma(tz_f4 (*acc)[NTB], const uint4 (&a)[MTW][NS], const uint4 (&bv)[NTB][NS]) {
    constexpr int NP = NS == 3 ? 6 : NS == 2 ? 3 : 1;
    // piece pairs (weight piece, activation piece), smallest terms first
    constexpr int PW[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int PW2[3] = {1, 0, 0}, PA2[3] = {0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        const int sw = NS == 3 ? PW[pr] : NS == 2 ? PW2[pr] : 0, sa = NS == 3 ? PA[pr] : NS == 2 ? PA2[pr] : 0;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTB; ++nt)
                if constexpr (F16)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tz_h8, a[mt][sw]), __builtin_bit_cast(tz_h8, bv[nt][sa]), acc[mt][nt], 0, 0, 0);
                else
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tz_bf8, a[mt][sw]), __builtin_bit_cast(tz_bf8, bv[nt][sa]), acc[mt][nt], 0, 0, 0);
    }
}
template <int MTW, int NS, int KS>
__device__ __forceinline__ void tz_read_a(uint4 (&a)[MTW][NS], const uint4* __restrict__ ap) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int s = 0; s < NS; ++s) a[mt][s] = ap[((long)mt * KS) * NS * 64 + s * 64];
}
template <int K, int CG, int NT, int MTW, int NS, bool F16>
__device__ __forceinline__ void tz_accumulate(tz_f4 (&acc)[MTW][NT], const uint4* __restrict__ aimg, const unsigned char* __restrict__ lds, const int (&boff)[NT], int ebase,
                                              int Sd, int RW, int Sc, int Sp, int mt0) {
    constexpr int MT = CG / 4, KS = CG / 4;
    constexpr int NTB = NT >= 2 ? 2 : 1, NU = NT / NTB;
    static_assert(NU == 1 || NU == 2, "units per step");
    constexpr int SPR = K * KS;                         // steps per kd row: (kh, ks); fully unrolled, the kd loop is rolled
    constexpr long STEP = (long)NS * 64, ENTRY = (long)MT * KS * NS * 64;      // uint4 per step / per entry of the image
    const int lane = threadIdx.x & 63;
    const uint4* __restrict__ ap = aimg + ((long)mt0 * KS) * NS * 64 + lane;   // (kd, kh = 0, ks = 0) of this wave's first M-tile
    // weight operands: a FIFO two steps deep (global loads from the L2-resident image; one step of MFMAs does not cover their latency)
    uint4 a0[MTW][NS], a1[MTW][NS], a2[MTW][NS];
    auto a_ptr = [&](int st) { return ap + (long)(st / KS) * ENTRY + (long)(st % KS) * STEP; };      // step index relative to the current kd row (may run into the next)
    tz_read_a<MTW, NS, KS>(a0, a_ptr(0));
    tz_read_a<MTW, NS, KS>(a1, a_ptr(1));
    // accumulators regrouped per unit: acc[mt][u * NTB + i]
    tz_f4 au[NU][MTW][NTB];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int i = 0; i < NTB; ++i) au[u][mt][i] = acc[mt][u * NTB + i];
    uint4 b0[NTB][NS], b1[NTB][NS];
    int rowoff = ebase;                  // byte offset of (kd, kh = 0, ks = 0)
    tz_read_b<NTB, NS>(b0, lds, boff, rowoff, Sp);
#pragma unroll 1
    for (int kd = 0; kd < K; ++kd) {
#pragma unroll
        for (int st = 0; st < SPR; ++st) {
            const int kh = st / KS, ks = st % KS;
            const int off = rowoff + 2 * (kh * RW + ks * 4 * Sc);
            // next step (possibly the first of the next kd row)
            const int stn = st + 1;
            const int off_n = stn < SPR ? rowoff + 2 * ((stn / KS) * RW + (stn % KS) * 4 * Sc) : rowoff + 2 * Sd;
            const bool more = stn < SPR || kd + 1 < K;
            // (the image has spare entries behind it, so the last prefetches stay in bounds)
            tz_read_a<MTW, NS, KS>(a2, a_ptr(st + 2));
            if constexpr (NU == 2) {
                tz_read_b<NTB, NS>(b1, lds, boff + NTB, off, Sp);
                tz_mfma<NTB, MTW, NS, F16>(au[0], a0, b0);
                if (more) tz_read_b<NTB, NS>(b0, lds, boff, off_n, Sp);
                tz_mfma<NTB, MTW, NS, F16>(au[1], a0, b1);
            } else {
                if (more) tz_read_b<NTB, NS>(b1, lds, boff, off_n, Sp);
                tz_mfma<NTB, MTW, NS, F16>(au[0], a0, b0);
#pragma unroll
                for (int i = 0; i < NTB; ++i)
#pragma unroll
                    for (int s = 0; s < NS; ++s) b0[i][s] = b1[i][s];
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int s = 0; s < NS; ++s) { a0[mt][s] = a1[mt][s]; a1[mt][s] = a2[mt][s]; }
        }
        rowoff += 2 * Sd;
        ap += (long)K * ENTRY;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int i = 0; i < NTB; ++i) acc[mt][u * NTB + i] = au[u][mt][i];
}

// ------------------------------------------------------------------------------------------------------------------ the kernel
// H16 (round 6, bf16 storage mode): the block-internal tensors this kernel touches -- forward: the outputs y_k; backward: g_k and d_o -- are vx_bf16 arrays
template <int CG, int NT, int MTW, int NS, bool BWD, bool F16, bool H16 = false>
__global__ void __launch_bounds__(512) vx_tz_k(VxTz p) {
    static_assert(!H16 || !F16, "16-bit storage is a mode of the bf16-operand kernels");
    typedef typename std::conditional<H16, vx_bf16, float>::type TI;
    extern __shared__ __attribute__((aligned(16))) unsigned char tz_lds[];
    constexpr int MT = CG / 4, KS = CG / 4, MG = MT / MTW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = lane & 15, q = lane >> 4;
    const int nthr = blockDim.x;
    // blocks of one (b, g) volume on one XCD (they share halo lines in that XCD's L2): consecutive ids go to different XCDs, so de-interleave
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    }
    const int ntile = p.nTd * p.nTh * p.nTw;
    const int tile = bid % ntile;
    const int g = (bid / ntile) % p.G, b = bid / (ntile * p.G);
    const int tw_i = tile % p.nTw, th_i = (tile / p.nTw) % p.nTh, td_i = tile / (p.nTw * p.nTh);
    const int d0 = td_i * p.TD, h0 = th_i * p.TH, w0 = tw_i * p.TWB * 4;
    const int mg = wave % MG, ng = wave / MG;
    const int mt0 = mg * MTW;

    // the lane's positions
    int boff[NT], pd[NT], ph[NT], pw[NT];
    bool live[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int pos = (ng * NT + nt) * 16 + n;
        live[nt] = pos < p.P;
        pos = live[nt] ? pos : p.P - 1;
        // position order: w-block fastest, then D, then H -- the 16 / TWB rows of an N-tile are consecutive d-planes, whose (padded) stride makes the
        // operand reads of a 32-lane group bank-conflict free (tz_geo)
        const unsigned r1 = tz_divm((unsigned)pos, p.mTWB);
        const int wb = pos - (int)r1 * p.TWB;
        const unsigned r2 = tz_divm(r1, p.mTD);
        pd[nt] = (int)(r1 - r2 * p.TD);
        ph[nt] = (int)r2;
        pw[nt] = 4 * wb;
        live[nt] = live[nt] && d0 + pd[nt] < p.D && h0 + ph[nt] < p.H && w0 + pw[nt] < p.W;
    }
    tz_f4 acc[MTW][NT];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (tz_f4){0.f, 0.f, 0.f, 0.f};
    };
    auto set_boff = [&](const TzGeo& ge) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) boff[nt] = 2 * (q * ge.Sc + pd[nt] * ge.Sd + ph[nt] * p.RW + pw[nt]);
    };
    const long chan = (long)p.D * p.H * p.W;
    const uint4* __restrict__ img_g = p.img + (long)g * 35 * MT * KS * NS * 64;

    if constexpr (!BWD) {
        const TzGeo& ge = p.geo[0];
        int ex = 0;                                                          // fp16 mode: the tile's scale exponent
        if constexpr (F16) ex = tz_stage16<CG>((const float*)p.src[0], tz_lds, p, ge, 2, b, g, d0, h0, w0, nthr, reinterpret_cast<float*>(tz_lds + p.red_off));
        else if (!(p.dbg & 1)) tz_stage<CG, NS>((const float*)p.src[0], tz_lds, p, ge, 2, b, g, d0, h0, w0, nthr);
        set_boff(ge);
        __syncthreads();
        float* red = reinterpret_cast<float*>(tz_lds + p.red_off);            // [3][waves][MTW][4][2]
        const int nwave = nthr >> 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {              // c = 0: K = 5, 1: K = 3, 2: K = 1
            zero();
            if (p.dbg & 2) {}
            else if (c == 0) tz_accumulate<5, CG, NT, MTW, NS, F16>(acc, img_g, tz_lds, boff, 0, ge.Sd, p.RW, ge.Sc, ge.Sp, mt0);
            else if (c == 1) tz_accumulate<3, CG, NT, MTW, NS, F16>(acc, img_g + (long)25 * MT * KS * NS * 64, tz_lds, boff, 2 * (ge.Sd + p.RW), ge.Sd, p.RW, ge.Sc, ge.Sp, mt0);
            else tz_accumulate<1, CG, NT, MTW, NS, F16>(acc, img_g + (long)34 * MT * KS * NS * 64, tz_lds, boff, 4 * (ge.Sd + p.RW), ge.Sd, p.RW, ge.Sc, ge.Sp, mt0);
            if constexpr (F16) {                                             // undo the two power-of-two scales on the fp32 accumulators (exact)
                const float fs = ldexpf(1.0f, -(ex + (int)p.esc[g * 3 + c]));
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] *= fs;
            }
            TI* __restrict__ y = (TI*)p.out[c];
            const float* __restrict__ bias = p.bias[c];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const int co = g * CG + (mt0 + mt) * 4 + q;
                const float bv = bias ? bias[co] : 0.0f;
                float s = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (live[nt]) {
                        // (16-bit y: the InstanceNorm statistics are those of the values the readers load)
                        const float o0 = vx_round_as<TI>(acc[mt][nt][0] + bv), o1 = vx_round_as<TI>(acc[mt][nt][1] + bv), o2 = vx_round_as<TI>(acc[mt][nt][2] + bv), o3 = vx_round_as<TI>(acc[mt][nt][3] + bv);
                        vx_st4(y, ((long)b * p.C + co) * chan + ((long)(d0 + pd[nt]) * p.H + h0 + ph[nt]) * p.W + w0 + pw[nt], make_float4(o0, o1, o2, o3));
                        s += (o0 + o1) + (o2 + o3);
                        s2 = fmaf(o0, o0, s2); s2 = fmaf(o1, o1, s2); s2 = fmaf(o2, o2, s2); s2 = fmaf(o3, o3, s2);
                    }
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if (n == 0) {
                    float* r = red + ((((long)c * nwave + wave) * MTW + mt) * 4 + q) * 2;
                    r[0] = s; r[1] = s2;
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 3 * CG) {
            const int c = threadIdx.x / CG, cl = threadIdx.x % CG;
            const int mtg = cl >> 2, qq = cl & 3, mgi = mtg / MTW, mti = mtg % MTW;
            double s = 0.0, s2 = 0.0;
            const int nng = nwave / MG;
            for (int w_ = 0; w_ < nng; ++w_) {
                const float* r = red + ((((long)c * nwave + (w_ * MG + mgi)) * MTW + mti) * 4 + qq) * 2;
                s += (double)r[0]; s2 += (double)r[1];
            }
            const int k = 2 - c;                                   // part order: y1, y3, y5
            double* dst = p.part + ((((long)k * p.B + b) * p.C + g * CG + cl) * ntile + tile) * 2;
            dst[0] = s; dst[1] = s2;
        }
    } else {
        zero();
        tz_f4 acc5[MTW][NT];                                                 // fp16 mode: the K = 5 term in true units while the K = 3 term accumulates (their tiles have different scales)
        float* scr = reinterpret_cast<float*>(tz_lds + p.red_off);
        {
            const TzGeo& ge = p.geo[0];
            int ex = 0;
            if constexpr (F16) ex = tz_stage16<CG>((const float*)p.src[0], tz_lds, p, ge, 2, b, g, d0, h0, w0, nthr, scr);
            else if (!(p.dbg & 1)) tz_stage<CG, NS, TI>((const TI*)p.src[0], tz_lds, p, ge, 2, b, g, d0, h0, w0, nthr);
            set_boff(ge);
            __syncthreads();
            if (!(p.dbg & 2)) tz_accumulate<5, CG, NT, MTW, NS, F16>(acc, img_g, tz_lds, boff, 0, ge.Sd, p.RW, ge.Sc, ge.Sp, mt0);
            if constexpr (F16) {
                const float fs = ldexpf(1.0f, -(ex + (int)p.esc[g * 3 + 0]));
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { acc5[mt][nt] = acc[mt][nt] * fs; acc[mt][nt] = (tz_f4){0.f, 0.f, 0.f, 0.f}; }
            }
        }
        __syncthreads();
        {
            const TzGeo& ge = p.geo[1];
            int ex = 0;
            if constexpr (F16) ex = tz_stage16<CG>((const float*)p.src[1], tz_lds, p, ge, 1, b, g, d0, h0, w0, nthr, scr);
            else if (!(p.dbg & 1)) tz_stage<CG, NS, TI>((const TI*)p.src[1], tz_lds, p, ge, 1, b, g, d0, h0, w0, nthr);
            set_boff(ge);
            __syncthreads();
            if (!(p.dbg & 2)) tz_accumulate<3, CG, NT, MTW, NS, F16>(acc, img_g + (long)25 * MT * KS * NS * 64, tz_lds, boff, 0, ge.Sd, p.RW, ge.Sc, ge.Sp, mt0);
            if constexpr (F16) {
                const float fs = ldexpf(1.0f, -(ex + (int)p.esc[g * 3 + 1]));
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc5[mt][nt] + acc[mt][nt] * fs;
            }
        }
        // epilogue: + conv1^T(g1) + d_o straight from global memory
        const TI* __restrict__ g1 = (const TI*)p.src[2];
        const TI* __restrict__ res = (const TI*)p.res;
        float* __restrict__ dxo = (float*)p.out[0];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!live[nt]) continue;
            const long sp_off = ((long)(d0 + pd[nt]) * p.H + h0 + ph[nt]) * p.W + w0 + pw[nt];
            for (int cc = 0; cc < CG; ++cc) {
                const float4 gv = vx_ld4(g1, ((long)b * p.C + g * CG + cc) * chan + sp_off);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    const float wv = p.w1[(long)(g * CG + cc) * CG + (mt0 + mt) * 4 + q];
                    acc[mt][nt][0] = fmaf(wv, gv.x, acc[mt][nt][0]); acc[mt][nt][1] = fmaf(wv, gv.y, acc[mt][nt][1]);
                    acc[mt][nt][2] = fmaf(wv, gv.z, acc[mt][nt][2]); acc[mt][nt][3] = fmaf(wv, gv.w, acc[mt][nt][3]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const long off = ((long)b * p.C + g * CG + (mt0 + mt) * 4 + q) * chan + sp_off;
                const float4 r = vx_ld4(res, off);
                *reinterpret_cast<float4*>(dxo + off) = make_float4(acc[mt][nt][0] + r.x, acc[mt][nt][1] + r.y, acc[mt][nt][2] + r.z, acc[mt][nt][3] + r.w);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
static unsigned tz_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / (unsigned long long)d); }      // see tz_divm

static int g_tz_pieces = 22;                  // (the mode functional.set_precision("fp32") installs: the default does not depend on whether that was ever called)
// The operand image of a block (vx_jlc_tz_prep) is laid out for the mode in force when it was built; its consumers -- the input gradient, possibly much later the
// deferred weight gradients -- must use THAT mode, not whatever the process-wide switch says by then: the *_ns entries take it as an argument, the operator
// code records it in its state at forward time.  (thread-local override: the bodies below read tz_pieces().)
static thread_local int t_tz_pieces = 0;
static inline int tz_pieces() { return t_tz_pieces ? t_tz_pieces : g_tz_pieces; }
namespace { struct TzPiecesScope { int prev; explicit TzPiecesScope(int ns) : prev(t_tz_pieces) { t_tz_pieces = ns; } ~TzPiecesScope() { t_tz_pieces = prev; } }; }
static inline bool tz_pieces_valid(int ns) { return (ns >= 1 && ns <= 3) || ns == 22; }
// bf16 storage of the block-internal tensors (the *_h entries, round 6): a call-scoped flag like the pieces mode; only with plain bf16 operands (pieces = 1)
static thread_local int t_tz_h16 = 0;
namespace { struct TzH16Scope { int prev; explicit TzH16Scope(int h) : prev(t_tz_h16) { t_tz_h16 = h; } ~TzH16Scope() { t_tz_h16 = prev; } }; }
static int g_tz_dbg = 0;
extern "C" int vx_jlc_tz_set_debug(int mask) { g_tz_dbg = mask; return 0; }
// 3 / 2 / 1 bf16 pieces per fp32 operand (6 / 3 / 1 piece products), or 22 = two scaled fp16 pieces (22 significant bits, 3 piece products)
extern "C" int vx_jlc_tz_set_pieces(int ns) { if (!((ns >= 1 && ns <= 3) || ns == 22)) return -1; g_tz_pieces = ns; return 0; }
static inline int tz_ns(int pieces) { return pieces == 22 ? 2 : pieces; }
extern "C" int vx_jlc_tz_pieces(void) { return g_tz_pieces; }

struct TzPlan { int CG, NT, MTW, nwaves; size_t shm; };

static void tz_geo(TzGeo& ge, const VxTz& p, int CG, int hw) {
    ge.HD = p.TD + 2 * hw; ge.HH = p.TH + 2 * hw;
    // Strides padded for the operand reads (ds_read_b64: two groups of 32 lanes = 16 positions x 2 channels, bank = (byte / 4) mod 64).  The 16 positions of an
    // N-tile are TWB w-blocks (8 bytes each, contiguous) x 16 / TWB consecutive d-planes; with the d-plane stride an ODD multiple of the TWB * 8-byte interval
    // (mod 256 bytes) the planes tile 32 banks, and with the channel stride = 128 bytes (mod 256) the second channel of the group takes the other 32.
    ge.Sd = ge.HH * p.RW;
    if (p.TWB >= 2) {
        const int iv = 4 * p.TWB;                      // interval in elements
        while (!((ge.Sd % iv) == 0 && ((ge.Sd / iv) & 1) && true)) ge.Sd += 4;
        // (mod 128 elements the same test: iv divides 128)
    }
    ge.Sc = ge.HD * ge.Sd;
    while ((ge.Sc & 127) != 64) ge.Sc += 4;
    ge.Sp = CG * ge.Sc;
    ge.nrows = CG * ge.HD * ge.HH;
    ge.mHH = tz_magic(ge.HH); ge.mHD = tz_magic(ge.HD);
    if (hw == 0) { ge.qd0 = 1; ge.nq = p.TWB; } else { ge.qd0 = 0; ge.nq = p.TWB + 2; }
    ge.mNQ = tz_magic(ge.nq);
}

// tile / wave plan: a function of (C/G, D, H, W) only -- never of the batch (a sample's partial sums must fold in the same order whatever it is batched with)
static int tz_plan(VxTz& p, TzPlan& pl, int B, int C, int G, int D, int H, int W, int pieces) {
    const int NS = tz_ns(pieces);
    if (B <= 0 || C <= 0 || G <= 0 || C % G || D <= 0 || H <= 0 || W <= 0 || (W & 3)) return -1;
    const int CG = C / G;
    if (CG != 4 && CG != 8 && CG != 16) return -1;
    p.B = B; p.C = C; p.G = G; p.D = D; p.H = H; p.W = W;
    static int td_env = -1, th_env = -1;
    if (td_env < 0) { const char* e = getenv("VELOXSEG_TZ_TD"); td_env = e ? atoi(e) : 0; const char* e2 = getenv("VELOXSEG_TZ_TH"); th_env = e2 ? atoi(e2) : 0; }
    const int tdm = td_env > 0 ? td_env : 8, thm = th_env > 0 ? th_env : 8;
    p.TWB = W / 4 < 8 ? W / 4 : 8;
    p.TH = H < thm ? H : thm;
    p.TD = D < tdm ? D : tdm;
    auto lds_bytes = [&](int td, int th, int twb) {
        VxTz q = p; TzGeo ge;
        q.TD = td; q.TH = th; q.TWB = twb; q.RW = 4 * twb + 4;
        tz_geo(ge, q, CG, 2);
        return (size_t)NS * ge.Sp * 2;
    };
    while (lds_bytes(p.TD, p.TH, p.TWB) > 150 * 1024) {
        if (p.TD > 1 && p.TD >= p.TH) p.TD = (p.TD + 1) / 2;
        else if (p.TH > 1) p.TH = (p.TH + 1) / 2;
        else return -1;
    }
    p.nTd = vx_cdiv(D, p.TD); p.nTh = vx_cdiv(H, p.TH); p.nTw = vx_cdiv(W / 4, p.TWB);
    p.RW = 4 * p.TWB + 4;
    p.P = p.TD * p.TH * p.TWB;
    p.nNt = vx_cdiv(p.P, 16);
    p.mTWB = tz_magic(p.TWB); p.mTD = tz_magic(p.TD);
    tz_geo(p.geo[0], p, CG, 2);
    tz_geo(p.geo[1], p, CG, 1);
    const int MT = CG / 4;
    pl.CG = CG;
    // waves = (M-tile groups) x (N-tile groups), at most 8 (two per SIMD), at most 4 accumulator tiles per wave.  Most waves first (latency hiding), then more
    // M-tiles per wave (an activation operand read from LDS then feeds more MFMAs: LDS, not the L2-resident weight images, is the scarcer path), then more N-tiles
    int best = -1;
    pl.NT = 1; pl.MTW = 1;
    for (int mtw = 1; mtw <= MT; mtw <<= 1)
        for (int nt = 1; nt <= 4; nt <<= 1) {
            if (mtw * nt > 4) continue;
            const int waves = (MT / mtw) * vx_cdiv(p.nNt, nt);
            if (waves > 8) continue;
            const int score = waves * 100 + mtw * 10 + nt;
            if (score > best) { best = score; pl.NT = nt; pl.MTW = mtw; }
        }
    if (best < 0) return -1;
    pl.nwaves = (MT / pl.MTW) * vx_cdiv(p.nNt, pl.NT);
    if (pl.nwaves > 8) return -1;
    if (pieces == 22) {             // fp16 two-piece staging holds every quad of a thread in registers while the tile's maximum is reduced
        const long nthr = 64L * pl.nwaves;
        if ((long)p.geo[0].nrows * p.geo[0].nq > TZ_NQ16 * nthr || (long)p.geo[1].nrows * p.geo[1].nq > TZ_NQ16 * nthr) return -1;
    }
    size_t halo = lds_bytes(p.TD, p.TH, p.TWB);
    halo = (halo + 15) & ~(size_t)15;
    p.red_off = (int)halo;
    pl.shm = halo + (size_t)3 * pl.nwaves * pl.MTW * 4 * 2 * sizeof(float);
    return 0;
}

// Smallest volume (voxels per channel) the matrix-pipe kernels are SELECTED for (vx_jlc_tz_ok; the entry points themselves take any supported shape).  Measured
// stand-alone at B = 4 (tools/jlc_tz_probe.py): 32^3 58 -> 32 us, 16^3 57 -> 42 us, 12^3 55 -> 42 us, 8^3 31 -> 34 us, 4^3 41 -> 52 us: at 8^3 and below a launch has 32 blocks and the chain
// stage -> 35 dependent entries -> epilogue is latency-bound either way.
static long g_tz_min_v = 1024;
extern "C" int vx_jlc_tz_set_min_voxels(long v) { g_tz_min_v = v < 0 ? 0 : v; return 0; }
extern "C" int vx_jlc_tz_ok(int C, int G, int D, int H, int W) {
    VxTz p = {};
    TzPlan pl;
    if ((long)D * H * W < g_tz_min_v) return 0;
    return tz_plan(p, pl, 1, C, G, D, H, W, tz_pieces()) == 0 ? 1 : 0;
}
extern "C" int vx_jlc_tz_ntiles(int C, int G, int D, int H, int W) {
    VxTz p = {};
    TzPlan pl;
    if (tz_plan(p, pl, 1, C, G, D, H, W, tz_pieces()) != 0) return -1;
    return p.nTd * p.nTh * p.nTw;
}
// uint4 elements of ONE direction's image (+ one spare entry for the operand prefetch); the workspace of vx_jlc_tz_prep holds two (forward, input gradient)
static long tz_img_elems(int C, int G, int pieces) {
    const int NS = tz_ns(pieces);
    const int CG = C / G, MT = CG / 4;
    return (long)G * 35 * MT * MT * NS * 64 + (long)2 * MT * MT * NS * 64;        // + two spare entries: the operand FIFO reads two steps ahead
}
extern "C" int vx_jlc_tz_img_floats(int C, int G) { return (int)(2 * tz_img_elems(C, G, tz_pieces()) * 4 + ((3 * G + 3) & ~3)); }       // + the scale exponents of the fp16 mode

extern "C" int vx_jlc_tz_prep(const float* w1, const float* w3, const float* w5, float* img, int C, int G, void* stream) {
    VX_REQUIRE(w1 && w3 && w5 && img && G > 0 && C % G == 0, "vx_jlc_tz_prep: bad args");
    const int CG = C / G, NS = tz_ns(tz_pieces());
    VX_REQUIRE(CG == 4 || CG == 8 || CG == 16, "vx_jlc_tz_prep: group width %d", CG);
    const long ne = tz_img_elems(C, G, tz_pieces());
    uint4* f = reinterpret_cast<uint4*>(img);
    uint4* bw = f + ne;
    const int MT = CG / 4;
    const long total = (long)G * 35 * MT * MT * 64;
    const dim3 grid((unsigned)vx_cdiv(total, 256));
    hipStream_t st = (hipStream_t)stream;
    if (tz_pieces() == 22) {
        float* esc = img + 2 * ne * 4;
        const dim3 g16((unsigned)vx_cdiv((long)G * 35 * MT * MT, 4));
        vx_tz_wmax_k<<<dim3((unsigned)G, 3), dim3(256), 0, st>>>(w1, w3, w5, esc, CG);
        if (CG == 4) vx_tz_prep16_k<4><<<g16, dim3(256), 0, st>>>(w1, w3, w5, f, bw, esc, G);
        else if (CG == 8) vx_tz_prep16_k<8><<<g16, dim3(256), 0, st>>>(w1, w3, w5, f, bw, esc, G);
        else vx_tz_prep16_k<16><<<g16, dim3(256), 0, st>>>(w1, w3, w5, f, bw, esc, G);
        VX_LAUNCH_CHECK("vx_jlc_tz_prep");
        return 0;
    }
#define TZ_PREP(cg, ns) vx_tz_prep_k<cg, ns><<<grid, dim3(256), 0, st>>>(w1, w3, w5, f, bw, G)
    if (NS == 3) { if (CG == 4) TZ_PREP(4, 3); else if (CG == 8) TZ_PREP(8, 3); else TZ_PREP(16, 3); }
    else if (NS == 2) { if (CG == 4) TZ_PREP(4, 2); else if (CG == 8) TZ_PREP(8, 2); else TZ_PREP(16, 2); }
    else { if (CG == 4) TZ_PREP(4, 1); else if (CG == 8) TZ_PREP(8, 1); else TZ_PREP(16, 1); }
#undef TZ_PREP
    VX_LAUNCH_CHECK("vx_jlc_tz_prep");
    return 0;
}

template <int CG, int NT, int MTW, int NS, bool BWD, bool F16, bool H16 = false>
static int tz_launch_t(const VxTz& p, const TzPlan& pl, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)vx_tz_k<CG, NT, MTW, NS, BWD, F16, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
        attr = true;
    }
    const dim3 grid((unsigned)(p.nTd * p.nTh * p.nTw * p.G * p.B));
    vx_tz_k<CG, NT, MTW, NS, BWD, F16, H16><<<grid, dim3(64 * pl.nwaves), pl.shm, st>>>(p);
    return 0;
}
template <int NS, bool BWD, bool F16 = false, bool H16 = false>
static int tz_launch(const VxTz& p, const TzPlan& pl, hipStream_t st) {
#define TZ_CASE(cg, nt, mtw) if (pl.CG == cg && pl.NT == nt && pl.MTW == mtw) return tz_launch_t<cg, nt, mtw, NS, BWD, F16, H16>(p, pl, st)
    TZ_CASE(4, 1, 1); TZ_CASE(4, 2, 1); TZ_CASE(4, 4, 1);
    TZ_CASE(8, 1, 1); TZ_CASE(8, 2, 1); TZ_CASE(8, 4, 1); TZ_CASE(8, 1, 2); TZ_CASE(8, 2, 2);
    TZ_CASE(16, 1, 1); TZ_CASE(16, 2, 1); TZ_CASE(16, 4, 1); TZ_CASE(16, 1, 2); TZ_CASE(16, 2, 2); TZ_CASE(16, 1, 4);
#undef TZ_CASE
    return -3;
}

extern "C" int vx_jlc_tz_fwd(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                             int B, int C, int G, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && img && y1 && y3 && y5 && part, "vx_jlc_tz_fwd: null pointer");
    VxTz p = {};
    TzPlan pl;
    const int NS = tz_ns(tz_pieces());
    VX_REQUIRE(tz_plan(p, pl, B, C, G, D, H, W, tz_pieces()) == 0, "vx_jlc_tz_fwd: unsupported shape C=%d G=%d %dx%dx%d", C, G, D, H, W);
    p.src[0] = x; p.img = reinterpret_cast<const uint4*>(img);
    p.esc = img + 2 * tz_img_elems(C, G, tz_pieces()) * 4;
    p.bias[0] = b5; p.bias[1] = b3; p.bias[2] = b1;
    p.out[0] = y5; p.out[1] = y3; p.out[2] = y1; p.part = part; p.dbg = g_tz_dbg;
    VX_REQUIRE(!t_tz_h16 || tz_pieces() == 1, "vx_jlc_tz_fwd: 16-bit outputs need plain bf16 operands (pieces = 1, the bf16 mode)");
    const int rc = tz_pieces() == 22 ? tz_launch<2, false, true>(p, pl, (hipStream_t)stream)
                   : NS == 3 ? tz_launch<3, false>(p, pl, (hipStream_t)stream) : NS == 2 ? tz_launch<2, false>(p, pl, (hipStream_t)stream)
                   : t_tz_h16 ? tz_launch<1, false, false, true>(p, pl, (hipStream_t)stream) : tz_launch<1, false>(p, pl, (hipStream_t)stream);
    VX_REQUIRE(rc == 0, "vx_jlc_tz_fwd: no kernel instance (rc %d) for group width %d, NT %d, MTW %d", rc, pl.CG, pl.NT, pl.MTW);
    VX_LAUNCH_CHECK("vx_jlc_tz_fwd");
    return 0;
}

extern "C" int vx_jlc_tz_bwd(const float* g1, const float* g3, const float* g5, const float* img, const float* w1, const float* d_o, float* dx,
                             int B, int C, int G, int D, int H, int W, void* stream) {
    VX_REQUIRE(g1 && g3 && g5 && img && w1 && d_o && dx, "vx_jlc_tz_bwd: null pointer");
    VxTz p = {};
    TzPlan pl;
    const int NS = tz_ns(tz_pieces());
    VX_REQUIRE(tz_plan(p, pl, B, C, G, D, H, W, tz_pieces()) == 0, "vx_jlc_tz_bwd: unsupported shape C=%d G=%d %dx%dx%d", C, G, D, H, W);
    p.src[0] = g5; p.src[1] = g3; p.src[2] = g1;
    p.img = reinterpret_cast<const uint4*>(img) + tz_img_elems(C, G, tz_pieces());
    p.esc = img + 2 * tz_img_elems(C, G, tz_pieces()) * 4;
    p.w1 = w1; p.res = d_o; p.out[0] = dx; p.dbg = g_tz_dbg;
    VX_REQUIRE(!t_tz_h16 || tz_pieces() == 1, "vx_jlc_tz_bwd: 16-bit inputs need plain bf16 operands (pieces = 1, the bf16 mode)");
    const int rc = tz_pieces() == 22 ? tz_launch<2, true, true>(p, pl, (hipStream_t)stream)
                   : NS == 3 ? tz_launch<3, true>(p, pl, (hipStream_t)stream) : NS == 2 ? tz_launch<2, true>(p, pl, (hipStream_t)stream)
                   : t_tz_h16 ? tz_launch<1, true, false, true>(p, pl, (hipStream_t)stream) : tz_launch<1, true>(p, pl, (hipStream_t)stream);
    VX_REQUIRE(rc == 0, "vx_jlc_tz_bwd: no kernel instance (rc %d) for group width %d, NT %d, MTW %d", rc, pl.CG, pl.NT, pl.MTW);
    VX_LAUNCH_CHECK("vx_jlc_tz_bwd");
    return 0;
}

// ======================================================================================================================================================================
// Weight gradients of the three grouped convolutions on the same pipe:  dW_K[co][ci][kd][kh][kw] = sum_{b, d, h, w} g_K[b][co][d][h][w] * x[b][ci][d+kd-K/2][h+kh-K/2][w+kw-K/2].
// GEMM of one MFMA: the reduction (k) axis is a whole W row -- 32 slots = nb samples x WS columns (WS = W rounded up to 8, nb = 32 / WS: the sum over the batch is part of
// the weight gradient, so small levels fill the slots with more samples) -- rows = (4 output channels x 4 consecutive H rows of g), columns = (4 input channels x 4
// consecutive H rows of x).  The 16 x 16 result holds every (g row j, x row i) pair: the pair belongs to tap kh = i - j + const, so one MFMA covers up to 5 taps along H
// ("Toeplitz in H": 10 of 16 entries used); kd is the plane offset between the two operands and kw a shift of the x operand ALONG the reduction axis, which is formed in
// registers from a 16-element aligned window (even shifts = register selection, odd shifts = 4 v_alignbit per piece).  fp32 products from 3 bf16 pieces per operand as above.
// A block = 4 waves owns (sample group, conv group, 4 x 4 channel pair, H tile of TH rows, D chunk) and marches over the x planes of its chunk: the current x plane and rings
// of 5 / 3 / 1 planes of g5 / g3 / g1 live in LDS as bf16 pieces; waves 0..2 hold the accumulators of (K = 5, kd = wave), (K = 3, kd = wave) and (wave 0) K = 1, wave 3 those
// of K = 5, kd = 3 and 4 (69 tiles of 16 x 16 per 4 x 4 channel pair in all).  After the march the tiles are folded along their diagonals into an LDS image of the three
// weight-gradient slices and added to global memory with float atomics (as every weight-gradient kernel of the library).
struct VxWgT {
    const float *x, *g5, *g3, *g1;
    float *dw5, *dw3, *dw1;
    int B, C, G, D, H, W, CG;
    int WS, nb, nsg;                 // slots per sample, samples per reduction chunk, sample groups
    int TH, nHt, DC, nDc, nXB;       // H tile (multiple of 4), D chunk, x row blocks per tile (TH / 4 + 1)
    int XR;                          // x row length in LDS (WS + 8: e = w + 4)
    int XP, GP, SP;                  // elements: x plane / one g plane / piece stride (XB XP + (R5 + R3 + R1) GP)
    int XB, R5, R3, R1, GB;          // x buffers (1; 2 with producer waves), ring slots of g5 / g3 / g1 (5 / 3 / 1; one more each with producer waves), element offset of slot 0 (XB XP)
    int dbg;                         // timing experiments: bit 0 no staging, bit 1 no MFMA phase, bit 2 no fold / atomics
};

template <int NS>
__device__ __forceinline__ void wg_split_store(float4 v, unsigned char* lds, long e0, long SP) {          // 4 consecutive elements at element offset e0 (a multiple of 4) of piece 0
    float a0 = v.x, a1 = v.y, a2 = v.z, a3 = v.w;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t lo = tz_pack(a0, a1), hi = tz_pack(a2, a3);
        *reinterpret_cast<uint2*>(lds + 2 * (s * SP + e0)) = make_uint2(lo, hi);
        if (s + 1 < NS) { a0 -= tz_lo(lo); a1 -= tz_hi(lo); a2 -= tz_lo(hi); a3 -= tz_hi(hi); }
    }
}

// Staging of one step = the x plane dx (halo rows / columns) and the newly needed g planes (g5: dx + 2, g3: dx + 1, g1: dx; owned planes only), as one list of
// quads (4 consecutive w): item -> (tensor, sample, channel, row, quad).  The loads of step dx + 1 are issued BEFORE the MFMAs of step dx and committed (split into
// pieces, stored to LDS) after them: a step's global latency hides behind the previous step's arithmetic.
#define WG_NPF 6
// one descriptor per (thread, item), decoded ONCE (the divisions cost more than a step's arithmetic when repeated every step): `g` = element offset inside plane 0 of the
// source tensor (the plane term pl * H * W is added per step), `l` = LDS element offset inside slot 0 (x: the x buffer), flags: bits 0..1 kind (0 x, 1 g5, 2 g3, 3 g1),
// bit 2 live (the item exists), bit 3 ok (inside the volume; else zeros are stored)
// (H16: the g_k tensors are vx_bf16 arrays -- their items' addresses advance 2 bytes per element and are fetched with 8-byte loads; x is fp32 either way)
struct WgDesc { const char* b; int l, f; };      // b: the item's element in plane 0 of ITS tensor (pointer chosen once: a per-step pointer select goes through a scratch table)
template <bool H16 = false>
__device__ __forceinline__ WgDesc wg_desc(const VxWgT& p, int it, int sg, int cob, int cib, int h0) {
    WgDesc r;
    r.b = (const char*)p.x; r.l = 0; r.f = 0;
    const int nqx = p.XR / 4, rows = p.TH + 4;
    const int nx = p.nb * 4 * rows * nqx;
    const int chan = p.D * p.H * p.W;
    if (it < nx) {
        const int qd = it % nqx; int t = it / nqx;
        const int row = t % rows; t /= rows;
        const int ci = t & 3, s = t >> 2;
        const int b = sg * p.nb + s, ih = h0 - 2 + row, iw = 4 * qd - 4;
        const bool ok = b < p.B && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        r.f = 4 | (ok ? 8 : 0);
        r.b = (const char*)(p.x + (ok ? (long)(b * p.C + cib + ci) * chan + ih * p.W + iw : 0));
        r.l = ((s * 4 + ci) * rows + row) * p.XR + 4 * qd;
        return r;
    }
    it -= nx;
    const int nqg = p.WS / 4;
    const int ng = p.nb * 4 * p.TH * nqg;
    const int kind = it / ng;                  // 0: g5, 1: g3, 2: g1
    if (kind > 2) return r;
    it -= kind * ng;
    const int qd = it % nqg; int t = it / nqg;
    const int row = t % p.TH; t /= p.TH;
    const int co = t & 3, s = t >> 2;
    const int b = sg * p.nb + s, ih = h0 + row, iw = 4 * qd;
    const bool ok = b < p.B && ih < p.H && iw < p.W;
    r.f = (kind + 1) | 4 | (ok ? 8 : 0);
    r.b = (const char*)(kind == 0 ? p.g5 : kind == 1 ? p.g3 : p.g1) + (H16 ? 2L : 4L) * (ok ? (long)(b * p.C + cob + co) * chan + ih * p.W + iw : 0);
    r.l = p.GB + ((s * 4 + co) * (p.TH + 1) + row) * p.WS + 4 * qd;
    return r;
}
// plane of item kind k at step dx: x: dx; g5: dx + 2; g3: dx + 1; g1: dx  (no table: (0x18 >> 2k) & 3 = 0, 2, 1, 0)
__device__ __forceinline__ int wg_dk(int k) { return (0x18 >> (2 * k)) & 3; }
// ring slot of g plane pl / element offset of the x buffer of step dx (planes from -2 on: + 60 keeps the dividend positive and is a multiple of every ring size used)
// (SP = producer waves: 6 / 4 / 2 slots and two x buffers, else 5 / 3 / 1 and one -- compile-time, or the remainders cost a division each and the kernel spills)
template <bool SP> __device__ __forceinline__ int wg_s5(const VxWgT&, int pl) { return (pl + 60) % (SP ? 6 : 5); }
template <bool SP> __device__ __forceinline__ int wg_s3(const VxWgT&, int pl) { return (SP ? 6 : 5) + (pl + 60) % (SP ? 4 : 3); }
template <bool SP> __device__ __forceinline__ int wg_s1(const VxWgT&, int pl) { return SP ? 10 + ((pl + 60) & 1) : 8; }
template <bool SP> __device__ __forceinline__ int wg_xo(const VxWgT& p, int dx) { return SP ? ((dx + 60) & 1) * p.XP : 0; }
template <bool H16 = false>
__device__ __forceinline__ void wg_prefetch(float4 (&pf)[WG_NPF], const WgDesc (&ds)[WG_NPF], const VxWgT& p, int dx, int dg0, int dg1) {
    const long HW = (long)p.H * p.W;
#pragma unroll
    for (int u = 0; u < WG_NPF; ++u) {
        const int k = ds[u].f & 3;
        const int pl = dx + wg_dk(k);
        const bool on = (ds[u].f & 8) && pl >= (k == 0 ? 0 : dg0) && pl < (k == 0 ? p.D : dg1);
        // explicit GLOBAL address space (a generic pointer makes FLAT loads, which count on lgkmcnt as well: every LDS wait of the MFMA phase would then also wait for
        // this step's prefetch)
        typedef const __attribute__((address_space(1))) tz_f4* wg_gptr;
        if constexpr (H16) {
            typedef unsigned int wg_u2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(1))) wg_u2* wg_gptr2;
            const bool isx = k == 0;
            const unsigned long long ga = (unsigned long long)ds[u].b + (isx ? 4ull : 2ull) * (unsigned long long)(on ? (long)pl * HW : 0);
            float4 r;
            if (isx) { const tz_f4 v = *(wg_gptr)ga; r = make_float4(v[0], v[1], v[2], v[3]); }
            else { const wg_u2 v = *(wg_gptr2)ga; r = make_float4(vx_bf16_lo(v[0]), vx_bf16_hi(v[0]), vx_bf16_lo(v[1]), vx_bf16_hi(v[1])); }
            pf[u] = on ? r : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
        const unsigned long long ga = (unsigned long long)ds[u].b + 4ull * (unsigned long long)(on ? (long)pl * HW : 0);      // (dead items read a valid address)
        const tz_f4 v = *(wg_gptr)ga;
        pf[u] = on ? make_float4(v[0], v[1], v[2], v[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <int NS, bool SP = false>
__device__ __forceinline__ void wg_commit(const float4 (&pf)[WG_NPF], const WgDesc (&ds)[WG_NPF], unsigned char* lds, const VxWgT& p, int dx, int dg0, int dg1) {
    // ring slots of the planes that arrive at this step (wave-uniform, once per step)
    const int l5 = wg_s5<SP>(p, dx + 2) * p.GP, l3 = wg_s3<SP>(p, dx + 1) * p.GP, l1 = wg_s1<SP>(p, dx) * p.GP, lx = wg_xo<SP>(p, dx);
#pragma unroll
    for (int u = 0; u < WG_NPF; ++u) {
        const int k = ds[u].f & 3;
        const int pl = dx + wg_dk(k);
        const bool on = (ds[u].f & 4) && pl >= (k == 0 ? 0 : dg0) && pl < (k == 0 ? p.D : dg1);
        const int lo = k == 0 ? lx : k == 1 ? l5 : k == 2 ? l3 : l1;
        if (on) wg_split_store<NS>(pf[u], lds, (long)ds[u].l + lo, p.SP);
    }
}

// fp16 two-piece staging of one step (pieces knob = 22).  Every plane gets its OWN power-of-two scale (largest magnitude of the staged tile -> [2^14, 2^15)); the
// exponents live in LDS: etab[0] = this step's x plane, etab[1 + slot] = the g plane in ring slot `slot` (valid for the five / three / one steps the plane stays).
// A (g plane, x plane) pair therefore arrives at the accumulators with exponent etab[0] + etab[1 + slot]; the MFMA phase keeps, per accumulator unit, the exponent
// its tiles currently carry and multiplies them by the (exact) power of two when a new pair differs.  Needs one more barrier per step than the bf16 staging.
__device__ __forceinline__ void wg_commit16(const float4 (&pf)[WG_NPF], const WgDesc (&ds)[WG_NPF], unsigned char* lds, const VxWgT& p, int dx, int dg0, int dg1,
                                            float* __restrict__ mtab, int* __restrict__ etab) {
    const int s5 = wg_s5<false>(p, dx + 2), s3 = wg_s3<false>(p, dx + 1), s1_ = wg_s1<false>(p, dx);
    const int l5 = s5 * p.GP, l3 = s3 * p.GP, l1 = s1_ * p.GP;
    float mk[4] = {0.f, 0.f, 0.f, 0.f};
    bool on_[WG_NPF];
#pragma unroll
    for (int u = 0; u < WG_NPF; ++u) {
        const int k = ds[u].f & 3;
        const int pl = dx + wg_dk(k);
        on_[u] = (ds[u].f & 4) && pl >= (k == 0 ? 0 : dg0) && pl < (k == 0 ? p.D : dg1);
        const float m_ = on_[u] ? fmaxf(fmaxf(fabsf(pf[u].x), fabsf(pf[u].y)), fmaxf(fabsf(pf[u].z), fabsf(pf[u].w))) : 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) mk[kk] = fmaxf(mk[kk], k == kk ? m_ : 0.0f);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) mk[kk] = vx_wave_max(mk[kk]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) mtab[(threadIdx.x >> 6) * 4 + kk] = mk[kk];
    }
    __syncthreads();
    int ek[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ek[kk] = tz_exp16(fmaxf(fmaxf(mtab[kk], mtab[4 + kk]), fmaxf(mtab[8 + kk], mtab[12 + kk])));
    if (threadIdx.x == 0) {
        etab[0] = ek[0];
        etab[1 + s5] = ek[1]; etab[1 + s3] = ek[2]; etab[1 + s1_] = ek[3];       // (planes that are not staged this step have no items: their slots are not read before a later staging)
    }
#pragma unroll
    for (int u = 0; u < WG_NPF; ++u) {
        if (!on_[u]) continue;
        const int k = ds[u].f & 3;
        const int lo = k == 0 ? 0 : k == 1 ? l5 : k == 2 ? l3 : l1;
        const float sc = ldexpf(1.0f, k == 0 ? ek[0] : k == 1 ? ek[1] : k == 2 ? ek[2] : ek[3]);
        const float a0 = pf[u].x * sc, a1 = pf[u].y * sc, a2 = pf[u].z * sc, a3 = pf[u].w * sc;
        const uint32_t lo0 = tz_pack16(a0, a1), hi0 = tz_pack16(a2, a3);
        const uint32_t lo1 = tz_pack16(a0 - tz_lo16(lo0), a1 - tz_hi16(lo0)), hi1 = tz_pack16(a2 - tz_lo16(hi0), a3 - tz_hi16(hi0));
        const long e0 = (long)ds[u].l + lo;
        *reinterpret_cast<uint2*>(lds + 2 * e0) = make_uint2(lo0, hi0);
        *reinterpret_cast<uint2*>(lds + 2 * ((long)p.SP + e0)) = make_uint2(lo1, hi1);
    }
}

template <int NS>
__device__ __forceinline__ void wg_read_a(uint4 (&a)[NS], const unsigned char* lds, const VxWgT& p, int slot, int arow_base, int local, int th_lane) {
    // lane (m = (co, j), q): row local + j of the g block, redirected to the zero row when it is outside the tile
    const int j = threadIdx.x & 3;
    const int r = local + j;
    const int rr = (unsigned)r < (unsigned)th_lane ? r : p.TH;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        a[s] = *reinterpret_cast<const uint4*>(lds + 2 * ((long)s * p.SP + p.GB + (long)slot * p.GP + (long)arow_base + (long)rr * p.WS));
}

template <int NS, bool F16 = false>
__device__ __forceinline__ void wg_mfma6(tz_f4& acc, const uint4 (&a)[NS], const uint4 (&b)[NS]) {
    constexpr int NP = NS == 3 ? 6 : NS == 2 ? 3 : 1;
    constexpr int PW[6] = {1, 2, 0, 1, 0, 0}, PA[6] = {1, 0, 2, 0, 1, 0};
    constexpr int PW2[3] = {1, 0, 0}, PA2[3] = {0, 1, 0};
#pragma unroll
    for (int pr = 0; pr < NP; ++pr) {
        const int sa = NS == 3 ? PW[pr] : NS == 2 ? PW2[pr] : 0, sb = NS == 3 ? PA[pr] : NS == 2 ? PA2[pr] : 0;
        if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(tz_h8, a[sa]), __builtin_bit_cast(tz_h8, b[sb]), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tz_bf8, a[sa]), __builtin_bit_cast(tz_bf8, b[sb]), acc, 0, 0, 0);
    }
}

// the x operand shifted by o elements (2..6) out of the 16-element window w[0..7] (dwords)
__device__ __forceinline__ uint4 wg_shift(const uint32_t (&w)[8], int o) {
    const int d = o >> 1;
    if ((o & 1) == 0) return make_uint4(w[d], w[d + 1], w[d + 2], w[d + 3]);
    return make_uint4(__builtin_amdgcn_alignbit(w[d + 1], w[d], 16), __builtin_amdgcn_alignbit(w[d + 2], w[d + 1], 16), __builtin_amdgcn_alignbit(w[d + 3], w[d + 2], 16),
                      __builtin_amdgcn_alignbit(w[d + 4], w[d + 3], 16));
}

// DBG = true: the timing-experiment build (vx_jlc_tz_set_debug); the production instance carries none of its branches
// SPEC = true (round 5): 512 threads.  Waves 4..7 are PRODUCERS -- they load, split and store the operands of step dx + 1 (second x buffer, one more slot per g ring) while
// waves 0..3 run the MFMA phase of step dx; one barrier per step.  The one-role kernel spent two thirds of a step outside its MFMAs: the barrier pair around the commit,
// the piece splitting and the exposed latency of the next plane's loads (tools/jlc_wg_probe.py), with one wave per SIMD to hide none of it.
template <int NS, bool DBG, bool F16 = false, bool SPEC = false, bool H16 = false>
__global__ void __launch_bounds__(SPEC ? 512 : 256, SPEC ? 1 : 2) vx_jlc_wg_k(VxWgT p_) {
    const VxWgT& p = p_;
    const int dbg = DBG ? p_.dbg : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r16 = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int dc = bid % p.nDc; bid /= p.nDc;
    const int ht = bid % p.nHt; bid /= p.nHt;
    const int MT = p.CG >> 2;
    const int ih = bid % MT; bid /= MT;
    const int ch = bid % MT; bid /= MT;
    const int g = bid % p.G;
    const int sg = bid / p.G;
    const int h0 = ht * p.TH;
    const int dg0 = dc * p.DC, dg1 = min(p.D, dg0 + p.DC);          // owned g planes
    const int cob = g * p.CG + ch * 4, cib = g * p.CG + ih * 4;

    // zero rows of every g slot (never overwritten)
    constexpr int nslot = SPEC ? 12 : 9;
    for (int it = threadIdx.x; it < nslot * p.nb * 4 * (p.WS / 2) * NS; it += blockDim.x) {
        int r = it;
        const int e2 = r % (p.WS / 2); r /= (p.WS / 2);
        const int sc = r % (p.nb * 4); r /= (p.nb * 4);
        const int slot = r % nslot, s = r / nslot;
        *reinterpret_cast<uint32_t*>(wg_lds + 2 * ((long)s * p.SP + p.GB + (long)slot * p.GP + ((long)sc * (p.TH + 1) + p.TH) * p.WS + 2 * e2)) = 0u;
    }
    // lane constants: sample / column of the lane's 8 reduction slots
    // (WS = 24: slots 24..31 belong to no sample -- those lanes read the zero row as their g operand, whatever x operand they pair it with)
    const int smp_ = (8 * q) / p.WS;
    const bool nosmp = smp_ >= p.nb;
    const int smp = nosmp ? 0 : smp_, wq = nosmp ? 0 : 8 * q - smp_ * p.WS;
    const int arow_base = ((smp * 4 + (r16 >> 2)) * (p.TH + 1)) * p.WS + wq;              // + row * WS
    const int brow_base = ((smp * 4 + (r16 >> 2)) * (p.TH + 4) + (r16 & 3)) * p.XR + wq;  // + 4 xb * XR
    const int th_lane = nosmp ? 0 : p.TH;                                                // rows >= th_lane are redirected to the zero row

    // 20 accumulator tiles per wave, one register set for both roles.  waves 0..2: [t*5 + kw] = K5 (kd = wave), [10 + t*3 + kw] = K3 (kd = wave), [16] = K1 (wave 0);
    // wave 3: [kk*10 + t*5 + kw] = K5 (kd = 3 + kk)
    tz_f4 acc[20];
    // fp16 mode: maxima / exponent tables behind the piece planes; the exponent each accumulator unit currently carries (wave-uniform; 0 with zero tiles at the start)
    float* mtab = reinterpret_cast<float*>(wg_lds + 2 * (long)NS * p.SP);
    int* etab = reinterpret_cast<int*>(mtab + 16);
    int ecur[5] = {0, 0, 0, 0, 0};
    const bool producer = SPEC && wave >= 4;
    // The two roles are two separate loops (the same number of barriers each): in ONE loop the staging registers of the producers and the 80 accumulator registers of the
    // consumers were live together and the kernel spilled 164 VGPRs.  The accumulators exist on the consumers' path only.
    if (SPEC && producer) {
        float4 pf[WG_NPF];
        WgDesc ds[WG_NPF];
#pragma unroll
        for (int u = 0; u < WG_NPF; ++u) ds[u] = wg_desc<H16>(p, (int)(threadIdx.x & 255) + u * 256, sg, cob, cib, h0);
        wg_prefetch<H16>(pf, ds, p, dg0 - 2, dg0, dg1);
        __syncthreads();                 // (the zero rows)
        wg_commit<NS, true>(pf, ds, wg_lds, p, dg0 - 2, dg0, dg1);        // the first step's operands; the second step's loads in flight
        wg_prefetch<H16>(pf, ds, p, dg0 - 1, dg0, dg1);
        __syncthreads();
        for (int dx = dg0 - 2; dx <= dg1 + 1; ++dx) {
            // step dx + 1: its x buffer and ring slots are not read by the consumers' step dx (one more buffer / slot than the planes in use)
            if (dx + 1 <= dg1 + 1) {
                wg_commit<NS, true>(pf, ds, wg_lds, p, dx + 1, dg0, dg1);
                if (dx + 2 <= dg1 + 1) wg_prefetch<H16>(pf, ds, p, dx + 2, dg0, dg1);
            }
            __syncthreads();
        }
    } else {
#pragma unroll
    for (int t = 0; t < 20; ++t) acc[t] = (tz_f4){0.f, 0.f, 0.f, 0.f};
    // the march: x planes dg0 - 2 .. dg1 + 1 (every owned g plane arrives exactly when its first partner does: g5[dx + 2], g3[dx + 1], g1[dx]); a step whose
    // x plane lies outside the volume only stages
    float4 pf[WG_NPF];
    WgDesc ds[WG_NPF];
    if constexpr (!SPEC) {
#pragma unroll
        for (int u = 0; u < WG_NPF; ++u) ds[u] = wg_desc<H16>(p, (int)threadIdx.x + u * 256, sg, cob, cib, h0);
        wg_prefetch<H16>(pf, ds, p, dg0 - 2, dg0, dg1);
    } else {
        __syncthreads();
        __syncthreads();
    }
    for (int dx = dg0 - 2; dx <= dg1 + 1; ++dx) {
        if constexpr (SPEC) {
            if ((unsigned)dx >= (unsigned)p.D) { __syncthreads(); continue; }
        } else {
        __syncthreads();                 // the previous step's reads of the x buffer and of the ring slots overwritten below are done
        if constexpr (F16) wg_commit16(pf, ds, wg_lds, p, dx, dg0, dg1, mtab, etab);
        else if (!(dbg & 1)) wg_commit<NS>(pf, ds, wg_lds, p, dx, dg0, dg1);
        __syncthreads();
        if (dx + 1 <= dg1 + 1 && !(dbg & 1)) wg_prefetch<H16>(pf, ds, p, dx + 1, dg0, dg1);
        if (dbg & 2) continue;
        if ((unsigned)dx >= (unsigned)p.D) continue;
        }
        // wave-uniform data of the step (hoisted: ring-slot arithmetic per x block was a visible share of the kernel).  A wave has two "slots" of accumulators:
        //   waves 0..2: slot 0 = K5 at kd = wave, slot 1 = K3 at kd = wave;   wave 3: slot 0 = K5 at kd = 3, slot 1 = K5 at kd = 4;   wave 0 also K1 -> acc[13]
        // slot-1 tiles hold tap kw = (k + 1) % 5 in acc[.. + k] (K3: kw3 = k), so that BOTH kinds of slot-1 unit read the x operand shifted by k + 1 and no
        // per-lane select between shifted operands is needed
        const bool k3w = wave < 3;
        const int kd0 = k3w ? wave : 3, kd1 = k3w ? wave : 4;
        const int pl0 = dx - kd0 + 2, pl1 = k3w ? dx - kd1 + 1 : dx - kd1 + 2;
        const bool ok0 = pl0 >= dg0 && pl0 < dg1, ok1 = pl1 >= dg0 && pl1 < dg1, okc = wave == 0 && dx >= dg0 && dx < dg1;
        const int so0 = wg_s5<SPEC>(p, pl0), so1 = k3w ? wg_s3<SPEC>(p, pl1) : wg_s5<SPEC>(p, pl1);
        const int xo8 = wg_xo<SPEC>(p, dx) >> 3;                               // this step's x buffer (element offset / 8)
        if constexpr (F16) {
            // bring each accumulator slot to the exponent of this step's (g plane, x plane) pair: an exact power-of-two multiply of its tiles, once per step and only
            // when the exponent differs from the one the tiles carry (planes of one tensor mostly share it).  ecur[0] / [1] / [4]: slot 0 / slot 1 / the K1 tile
            const int e_x = __builtin_amdgcn_readfirstlane(etab[0]);
            auto bring = [&](int which, int slot, int lo, int n_) {
                const int ep = e_x + __builtin_amdgcn_readfirstlane(etab[1 + slot]);
                if (ep == ecur[which]) return;
                const int dd = ep - ecur[which];
                const float f = ldexpf(1.0f, dd > 120 ? 120 : (dd < -120 ? -120 : dd));
#pragma unroll
                for (int t = 0; t < 20; ++t)
                    if (t >= lo && t < lo + n_ && !(which == 1 && k3w && (t == 13 || t == 14 || t == 18 || t == 19))) acc[t] *= f;
                ecur[which] = ep;
            };
            if (ok0) bring(0, so0, 0, 10);
            if (ok1) bring(1, so1, 10, 10);
            if (okc) {
                const int ep = e_x + __builtin_amdgcn_readfirstlane(etab[1 + wg_s1<SPEC>(p, dx)]);
                if (ep != ecur[4]) {
                    const int dd = ep - ecur[4];
                    acc[13] *= ldexpf(1.0f, dd > 120 ? 120 : (dd < -120 ? -120 : dd));
                    ecur[4] = ep;
                }
            }
        }
        for (int xb = 0; xb < p.nXB; ++xb) {
            // the lane's 16-element window of x row 4 xb + i, every piece
            uint32_t w[NS][8];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                // (element offsets are multiples of 8: XR, SP, brow_base -- say so, or the reads become 4-byte ds_read2_b32)
                const int eo = ((s * (p.SP >> 3) + xo8 + (brow_base >> 3) + 4 * xb * (p.XR >> 3)) << 3);
                const uint4* bp = reinterpret_cast<const uint4*>(wg_lds + 2 * (long)eo);
                const uint4 lo = bp[0], hi = bp[1];
                w[s][0] = lo.x; w[s][1] = lo.y; w[s][2] = lo.z; w[s][3] = lo.w; w[s][4] = hi.x; w[s][5] = hi.y; w[s][6] = hi.z; w[s][7] = hi.w;
            }
            uint4 sh[5][NS];              // kw = 0..4  <->  element offset 2..6
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int s = 0; s < NS; ++s) sh[k][s] = wg_shift(w[s], k + 2);
            uint4 a[NS];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = u >> 1, typ = u & 1;
                const bool k3 = sl == 1 && k3w;                              // this unit is a K = 3 unit
                if (!(sl ? ok1 : ok0)) continue;
                // type 0: g rows 4 xb + j (K3: 4 xb - 1 + j) -> kh = i - j; type 1: g rows 4 xb - 4 + j (K3: 4 xb - 5 + j) -> kh = i - j + 4
                const int local = 4 * xb - (typ ? 4 : 0) - (k3 ? 1 : 0);
                if (local + 3 < 0 || local >= p.TH) continue;                // no row of the block inside the tile
                if (!(dbg & 16)) wg_read_a<NS>(a, wg_lds, p, sl ? so1 : so0, arow_base, local, th_lane);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if (k >= 3 && k3) continue;
                    if (!(dbg & 8)) wg_mfma6<NS, F16>(acc[5 * u + k], a, sh[sl ? (k + 1) % 5 : k]);
                }
            }
            if (okc) {
                wg_read_a<NS>(a, wg_lds, p, wg_s1<SPEC>(p, dx), arow_base, 4 * xb - 2, th_lane);
                wg_mfma6<NS, F16>(acc[13], a, sh[2]);
            }
        }
        if constexpr (SPEC) __syncthreads();          // step dx is read, step dx + 1 is staged
    }
    }
    if (dbg & 4) return;
    // Epilogue.  A tile holds every (g row j, x row i) pair of its two 4-row blocks; tap kh = i - j (+ 4 for the second block type) collects a diagonal, and the two
    // types of one (conv, kd, kw) meet in kh = 1..3.  The tiles go to LDS with plain 16-byte stores (LDS float atomics retire about one lane per clock: 80 of them
    // per lane were 13 us of this kernel), then one thread per weight sums its <= 8 entries and issues ONE global atomic.  Two rounds (wave slot 0, then slot 1):
    // [wave 4][tile 10][column n = (ci, i) 16][row m = (co, j) 16] floats = 40 KB per round, inside the dead staging area.
    float* tl = reinterpret_cast<float*>(wg_lds);
    if constexpr (F16) {                 // the tiles back in true units
        const float f0 = ldexpf(1.0f, -ecur[0]), f1 = ldexpf(1.0f, -ecur[1]), f13 = ldexpf(1.0f, -ecur[4]);
#pragma unroll
        for (int t = 0; t < 20; ++t) {
            if (t < 10) acc[t] *= f0;
            else if (wave < 3 && (t == 13 || t == 14 || t == 18 || t == 19)) { if (t == 13) acc[t] *= f13; }      // (K3 slots own 3 tiles per type; 13 is wave 0's K1 tile, zero elsewhere)
            else acc[t] *= f1;
        }
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        __syncthreads();
        if (!SPEC || wave < 4) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
            *reinterpret_cast<float4*>(tl + (((wave * 10 + t) * 16 + r16) * 16 + 4 * q)) = make_float4(acc[sl * 10 + t][0], acc[sl * 10 + t][1], acc[sl * 10 + t][2], acc[sl * 10 + t][3]);
        }
        __syncthreads();
        // outputs of this round in ADDRESS order (a wave-instruction of atomics then covers whole runs of one (co, ci) row: scattered 4-byte atomics are several
        // times slower per element): round 0 = K5 taps kd 0..3 (100 contiguous floats per pair), round 1 = K5 kd 4 (25 per pair) and all of K3 (27 per pair)
        const int n5 = sl == 0 ? 1600 : 400, n3 = sl == 0 ? 0 : 432;
        for (int it = threadIdx.x; it < n5 + n3; it += blockDim.x) {
            const bool k3 = it >= n5;
            const int o = k3 ? it - n5 : it;
            const int K = k3 ? 3 : 5, per = k3 ? 27 : (sl == 0 ? 100 : 25);
            const int pair = o / per, r = o - pair * per;
            const int kdl = r / (K * K), r2 = r - kdl * K * K;
            const int kh = r2 / K, kw = r2 - kh * K;
            const int kd = k3 ? kdl : (sl == 0 ? kdl : 4);
            const int wv = k3 ? kd : (kd < 3 ? kd : 3);
            const int co = pair >> 2, ci = pair & 3;
            // (slot-1 tiles of the K5 wave hold tap kw in tile (kw + 4) % 5: see the MFMA phase)
            const int kt = (!k3 && sl == 1) ? (kw + 4) % 5 : kw;
            float v = 0.0f;
#pragma unroll
            for (int typ = 0; typ < 2; ++typ) {
                const float* T = tl + ((wv * 10 + typ * 5 + kt) * 16) * 16;
                const int d = kh - (typ ? 4 : 0);                    // i - j
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int i = jj + d;
                    if (i >= 0 && i < 4) v += T[(ci * 4 + i) * 16 + co * 4 + jj];
                }
            }
            float* dst = k3 ? p.dw3 : p.dw5;
            const int K3 = K * K * K;
            if (dst) atomicAdd(dst + ((long)(cob + co) * p.CG + ih * 4 + ci) * K3 + (kd * K + kh) * K + kw, v);
        }
        if (sl == 1 && p.dw1) {                                     // K = 1: wave 0, tile 13 (= slot-1 tile 3), the diagonal i == j
            for (int pair = threadIdx.x; pair < 16; pair += blockDim.x) {
                const int co = pair >> 2, ci = pair & 3;
                const float* T = tl + ((0 * 10 + 3) * 16) * 16;
                float v = 0.0f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v += T[(ci * 4 + jj) * 16 + co * 4 + jj];
                atomicAdd(p.dw1 + (long)(cob + co) * p.CG + ih * 4 + ci, v);
            }
        }
    }
}

static int g_wg_blocks = 0;
extern "C" int vx_jlc_wgrad_tz_set_blocks(int n) { g_wg_blocks = n > 0 ? n : 0; return 0; }      // blocks per launch (0: the default rule in wg_plan)
static int g_wg_spec = -1;
extern "C" int vx_jlc_wgrad_tz_set_spec(int on) { g_wg_spec = on ? 1 : 0; return 0; }      // A/B: producer / consumer waves (default on; VELOXSEG_WG_SPEC)
static bool wg_spec_on() {
    if (g_wg_spec < 0) { const char* e = getenv("VELOXSEG_WG_SPEC"); g_wg_spec = (e && e[0] == '0') ? 0 : 1; }
    return g_wg_spec == 1;
}
static int wg_plan(VxWgT& p, size_t& shm, int B, int C, int G, int D, int H, int W, int NS, bool spec = false) {
    if (B <= 0 || C <= 0 || G <= 0 || C % G || D <= 0 || H <= 0 || W <= 0 || (W & 3) || (H & 3) || W > 32) return -1;
    const int CG = C / G;
    if (CG != 4 && CG != 8 && CG != 16) return -1;
    p.B = B; p.C = C; p.G = G; p.D = D; p.H = H; p.W = W; p.CG = CG;
    p.WS = (W + 7) / 8 * 8;
    p.nb = 32 / p.WS;
    p.nsg = vx_cdiv(B, p.nb);
    static int th_env = -1;
    if (th_env < 0) { const char* e = getenv("VELOXSEG_WG_TZ_TH"); th_env = e ? atoi(e) : 0; }
    p.TH = (H % 8 == 0 && th_env != 4) ? 8 : 4;       // (12-row tiles would need more than WG_NPF staging items per thread)
    p.nHt = H / p.TH;
    p.nXB = p.TH / 4 + 1;
    p.XR = p.WS + 8;
    p.XP = p.nb * 4 * (p.TH + 4) * p.XR;
    p.GP = p.nb * 4 * (p.TH + 1) * p.WS;
    p.XB = spec ? 2 : 1; p.R5 = spec ? 6 : 5; p.R3 = spec ? 4 : 3; p.R1 = spec ? 2 : 1;
    p.GB = p.XB * p.XP;
    p.SP = p.GB + (p.R5 + p.R3 + p.R1) * p.GP;
    const int MT = CG / 4;
    const long base_blocks = (long)p.nsg * G * MT * MT * p.nHt;
    static int target_env = -1;
    if (target_env < 0) { const char* e = getenv("VELOXSEG_WG_TZ_BLOCKS"); target_env = (e && atoi(e) > 0) ? atoi(e) : 0; }
    // Default: 128 blocks (HALF the chip) with producer / consumer waves, 256 for the one-role kernel.  With the producer waves a launch on 128 CUs takes what the one-role
    // kernel took on 256 (53.7 vs 52.8 us average), and what the sink leaves free is what the other lanes' chains use: 1040 patches/s against 1021 with 256 blocks (where
    // the kernel alone is 42.7 us: bench.py reports that figure as roofline.full_chip) and 1027 for the one-role kernel.  vx_jlc_wgrad_tz_set_blocks / VELOXSEG_WG_TZ_BLOCKS.
    const int target = g_wg_blocks > 0 ? g_wg_blocks : (target_env > 0 ? target_env : (spec ? 128 : 256));
    // Blocks per launch: 256 = one per CU.  Stand-alone, 512 (two resident blocks per CU, one's staging behind the other's MFMAs) is faster (67 vs 84 us at 32^3), but
    // this kernel is a SINK that runs beside the backward chains of the other lanes: at 74 KB of LDS and 248 VGPRs per block, two blocks per CU leave those chains
    // nothing to run on -- the step is 774 patches/s with 512 blocks, 798 with 256, 782 with the VALU weight-gradient kernels (VELOXSEG_WG_TZ_BLOCKS for the A/B)
    int nchunks = (int)((target + base_blocks - 1) / base_blocks);
    // at least two owned planes per chunk (a chunk walks its planes + 4: one-plane chunks do 5 steps for one plane's work).  On the 8-plane level one plane per chunk
    // (VELOXSEG_WG_TZ_MINDC8=1: 256 blocks instead of 128) is FASTER ALONE -- 31.9 vs 39.2 us, tools/jlc_wg_probe.py -- and SLOWER IN THE STEP (1020 vs 1028 patches/s,
    // same box, twice): this kernel is a sink beside the other lanes' chains, and what it costs them is CUs, not microseconds (as with 512 blocks at 32^3 in round 4)
    static int mindc8 = 0;
    if (!mindc8) { const char* e = getenv("VELOXSEG_WG_TZ_MINDC8"); mindc8 = (e && atoi(e) == 1) ? 1 : 2; }
    const int min_dc = D <= 8 ? mindc8 : 2;
    if (nchunks > D / min_dc) nchunks = D / min_dc;
    if (nchunks < 1) nchunks = 1;
    p.DC = vx_cdiv(D, nchunks);
    { static int dc_env = -1; if (dc_env < 0) { const char* e = getenv("VELOXSEG_WG_TZ_DC"); dc_env = e ? atoi(e) : 0; } if (dc_env > 0 && dc_env <= D) p.DC = dc_env; }      // (experiments)
    p.nDc = vx_cdiv(D, p.DC);
    if ((long)p.nb * 4 * (p.TH + 4) * (p.XR / 4) + 3L * p.nb * 4 * p.TH * (p.WS / 4) > (long)WG_NPF * 256) return -1;      // one step's staging list must fit the per-thread items
    shm = (size_t)NS * p.SP * 2 + 128;              // (+ the maxima / exponent tables of the fp16 mode)
    if (shm < 4 * 10 * 256 * sizeof(float)) shm = 4 * 10 * 256 * sizeof(float);      // the epilogue's tile area
    if (shm > 150 * 1024) return -1;
    return 0;
}

static long g_wg_min_v = -1;
static int g_wg_f16 = 0;
extern "C" int vx_jlc_wgrad_tz_set_f16(int on) { g_wg_f16 = on ? 1 : 0; return 0; }
extern "C" int vx_jlc_wgrad_tz_set_min_voxels(long v) { g_wg_min_v = v < 0 ? 0 : v; return 0; }
extern "C" int vx_jlc_wgrad_tz_ok(int C, int G, int D, int H, int W) {
    VxWgT p = {};
    size_t shm;
    if (g_wg_min_v < 0) { const char* e = getenv("VELOXSEG_WG_TZ_MIN_V"); g_wg_min_v = e ? atol(e) : 0; }
    if ((long)D * H * W < g_wg_min_v) return 0;
    return wg_plan(p, shm, 1, C, G, D, H, W, tz_pieces() == 22 ? 3 : tz_pieces()) == 0 ? 1 : 0;
}

extern "C" int vx_jlc_wgrad_tz(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                               void* stream) {
    VX_REQUIRE(x && g1 && g3 && g5, "vx_jlc_wgrad_tz: null pointer");
    VxWgT p = {};
    size_t shm;
    // pieces = 22 (two scaled fp16 pieces): the weight-gradient kernel has that mode too (per-plane exponents, exact accumulator rescaling), but it is 4 us SLOWER at
    // 32^3 than three bf16 pieces (68 vs 64 us): the MFMAs are 16 of the kernel's 64 us, and the per-step maxima + the extra barrier cost more than halving them
    // saves.  VELOXSEG_WG_TZ_F16=1 selects it (A/B, tests); default: three bf16 pieces.
    static int wg16 = -1;
    if (wg16 < 0) { const char* e = getenv("VELOXSEG_WG_TZ_F16"); wg16 = (e && e[0] == '1') ? 1 : 0; }
    const bool f16 = tz_pieces() == 22 && (wg16 == 1 || g_wg_f16 == 1);
    const int NS = f16 ? 2 : (tz_pieces() == 22 ? 3 : tz_pieces());
    VX_REQUIRE(!t_tz_h16 || NS == 1, "vx_jlc_wgrad_tz: 16-bit g_k need plain bf16 operands (pieces = 1, the bf16 mode)");
    // producer / consumer waves: the default (three bf16 pieces) instance, when its deeper staging (two x buffers, one more slot per ring) fits
    bool spec = !f16 && NS == 3 && (g_tz_dbg >> 4) == 0 && wg_spec_on();
    if (spec) { VxWgT q = {}; size_t sh2; spec = wg_plan(q, sh2, B, C, G, D, H, W, NS, true) == 0; }
    VX_REQUIRE(wg_plan(p, shm, B, C, G, D, H, W, NS, spec) == 0, "vx_jlc_wgrad_tz: unsupported shape C=%d G=%d %dx%dx%d", C, G, D, H, W);
    p.x = x; p.g1 = g1; p.g3 = g3; p.g5 = g5; p.dw1 = dw1; p.dw3 = dw3; p.dw5 = dw5; p.dbg = g_tz_dbg >> 4;
    const int MT = p.CG / 4;
    const dim3 grid((unsigned)((long)p.nsg * G * MT * MT * p.nHt * p.nDc));
    hipStream_t st = (hipStream_t)stream;
#define WG_LAUNCH(ns)                                                                                                                                      \
    do {                                                                                                                                                   \
        static bool attr = false;                                                                                                                          \
        if (!attr) {                                                                                                                                       \
            VX_REQUIRE(hipFuncSetAttribute((const void*)vx_jlc_wg_k<ns, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&           \
                       hipFuncSetAttribute((const void*)vx_jlc_wg_k<ns, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "vx_jlc_wgrad_tz: LDS attribute"); \
            attr = true;                                                                                                                                   \
        }                                                                                                                                                  \
        if (p.dbg) vx_jlc_wg_k<ns, true><<<grid, dim3(256), shm, st>>>(p);                                                                                 \
        else vx_jlc_wg_k<ns, false><<<grid, dim3(256), shm, st>>>(p);                                                                                      \
    } while (0)
    if (spec) {
        static bool attrs = false;
        if (!attrs) { VX_REQUIRE(hipFuncSetAttribute((const void*)vx_jlc_wg_k<3, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "vx_jlc_wgrad_tz: LDS attribute"); attrs = true; }
        vx_jlc_wg_k<3, false, false, true><<<grid, dim3(512), shm, st>>>(p);
    } else if (f16) {
        static bool attr16 = false;
        if (!attr16) { VX_REQUIRE(hipFuncSetAttribute((const void*)vx_jlc_wg_k<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "vx_jlc_wgrad_tz: LDS attribute"); attr16 = true; }
        vx_jlc_wg_k<2, false, true><<<grid, dim3(256), shm, st>>>(p);
    } else if (NS == 3) WG_LAUNCH(3); else if (NS == 2) WG_LAUNCH(2);
    else if (t_tz_h16) {                 // bf16 storage mode: g_k are vx_bf16 arrays (plain bf16 operands)
        static bool attrh = false;
        if (!attrh) { VX_REQUIRE(hipFuncSetAttribute((const void*)vx_jlc_wg_k<1, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "vx_jlc_wgrad_tz: LDS attribute"); attrh = true; }
        vx_jlc_wg_k<1, false, false, false, true><<<grid, dim3(256), shm, st>>>(p);
    } else WG_LAUNCH(1);
#undef WG_LAUNCH
    VX_LAUNCH_CHECK("vx_jlc_wgrad_tz");
    return 0;
}

// ---- the same entries with the pieces mode of the operand image given explicitly (see t_tz_pieces)
extern "C" int vx_jlc_tz_img_floats_ns(int C, int G, int pieces) {
    if (!tz_pieces_valid(pieces)) return -1;
    TzPiecesScope sc(pieces);
    return vx_jlc_tz_img_floats(C, G);
}
extern "C" int vx_jlc_tz_prep_ns(const float* w1, const float* w3, const float* w5, float* img, int C, int G, int pieces, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_tz_prep_ns: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    return vx_jlc_tz_prep(w1, w3, w5, img, C, G, stream);
}
extern "C" int vx_jlc_tz_fwd_ns(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                                int B, int C, int G, int D, int H, int W, int pieces, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_tz_fwd_ns: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    return vx_jlc_tz_fwd(x, img, b1, b3, b5, y1, y3, y5, part, B, C, G, D, H, W, stream);
}
extern "C" int vx_jlc_tz_bwd_ns(const float* g1, const float* g3, const float* g5, const float* img, const float* w1, const float* d_o, float* dx,
                                int B, int C, int G, int D, int H, int W, int pieces, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_tz_bwd_ns: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    return vx_jlc_tz_bwd(g1, g3, g5, img, w1, d_o, dx, B, C, G, D, H, W, stream);
}
// bf16 storage mode (h16 != 0; pieces must be 1): y_k / g_k / d_o are vx_bf16 arrays, x and dx stay fp32
extern "C" int vx_jlc_tz_fwd_h(const float* x, const float* img, const float* b1, const float* b3, const float* b5, void* y1, void* y3, void* y5, double* part,
                               int B, int C, int G, int D, int H, int W, int pieces, int h16, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_tz_fwd_h: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    TzH16Scope hs(h16 ? 1 : 0);
    return vx_jlc_tz_fwd(x, img, b1, b3, b5, (float*)y1, (float*)y3, (float*)y5, part, B, C, G, D, H, W, stream);
}
extern "C" int vx_jlc_tz_bwd_h(const void* g1, const void* g3, const void* g5, const float* img, const float* w1, const void* d_o, float* dx,
                               int B, int C, int G, int D, int H, int W, int pieces, int h16, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_tz_bwd_h: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    TzH16Scope hs(h16 ? 1 : 0);
    return vx_jlc_tz_bwd((const float*)g1, (const float*)g3, (const float*)g5, img, w1, (const float*)d_o, dx, B, C, G, D, H, W, stream);
}
extern "C" int vx_jlc_wgrad_tz_h(const float* x, const void* g1, const void* g3, const void* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                                 int pieces, int h16, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_wgrad_tz_h: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    TzH16Scope hs(h16 ? 1 : 0);
    return vx_jlc_wgrad_tz(x, (const float*)g1, (const float*)g3, (const float*)g5, dw1, dw3, dw5, B, C, G, D, H, W, stream);
}
extern "C" int vx_jlc_wgrad_tz_ns(const float* x, const float* g1, const float* g3, const float* g5, float* dw1, float* dw3, float* dw5, int B, int C, int G, int D, int H, int W,
                                  int pieces, void* stream) {
    VX_REQUIRE(tz_pieces_valid(pieces), "vx_jlc_wgrad_tz_ns: pieces %d", pieces);
    TzPiecesScope sc(pieces);
    return vx_jlc_wgrad_tz(x, g1, g3, g5, dw1, dw3, dw5, B, C, G, D, H, W, stream);
}
