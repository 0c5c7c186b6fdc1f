// JLC grouped convolutions (k = 1, 3, 5 from one input; reference conv_blocks.py:51-58,72-75) at the COARSE encoder levels -- 8^3 voxels with 8 channels per group,
// 4^3 with 16 (and the 6^3 / 3^3 volumes of the 96^3 configurations) -- as channels-last implicit GEMMs on the 16x16x32 f16 matrix pipe.  The Toeplitz form of
// jlc_mfma.hip needs rows of >= 16 voxels along W and its weight images grow to 420 KB per group here; the fp32 VALU kernels of jlc.hip run these levels at
// 42-45 us per launch for < 0.4 GFLOP (one wave per SIMD, latency bound).
//   GEMM of one (sample, group):  rows = output channels of the group (8 / 16), columns = 16 voxels, reduction = (input channel, tap): an MFMA step covers 4 taps x 8
//   channels (CG = 8) or 2 taps x 16 channels (CG = 16).  Taps are ordered centre-first: index 0..26 = the 3^3 centre in K = 3 order, 27 = padding, 28..125 = the 98
//   outer taps of the 5^3 cube, so the K = 3 product is the first 7 / 14 steps of the same operand stream and K = 1 is the step holding tap 13.
//   B operand: the group's input volume with its halo, channels-last in LDS as two fp16 pieces of x * 2^ex (ex from the block's own maximum): one ds_read_b128 per
//   piece, voxel and step.  A operand: per-group weight images in operand order (two fp16 pieces of w * 2^ew, ew per (group, kernel size)), streamed from L2 with a
//   two-deep register FIFO.  Products hi*hi + hi*lo + lo*hi (22 mantissa bits), fp32 accumulation, exact rescale by 2^-(ex+ew).
//   forward  : y1, y3, y5 + bias and the per-(b, c) partial sums (sum, sum of squares) the InstanceNorm statistics need (one slot per wave)
//   backward : dx = conv5^T(g5) + conv3^T(g3) + conv1^T(g1) + d_o: the same stream with the transposed images, three halos (pad 2 / 1 / 0), offsets negated
// A block = 4 waves = one (sample, group) and a chunk of 4 x TPW voxel tiles; every block stages the whole group volume (<= 16 KB of input).
#include "vx_common.h"
#include <stdlib.h>
#include "../../include/veloxseg_hip.h"

typedef _Float16 cl_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cl_h2 __attribute__((ext_vector_type(2)));
typedef float cl_f2 __attribute__((ext_vector_type(2)));
typedef float cl_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t cl_u4 __attribute__((ext_vector_type(4)));
#define CL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cl_h8, (a)), __builtin_bit_cast(cl_h8, (b)), (c), 0, 0, 0)

__device__ __forceinline__ void cl_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const cl_f2 v = {a, b};
    const cl_h2 h = __builtin_convertvector(v, cl_h2);
    const cl_h2 l = __builtin_convertvector(v - __builtin_convertvector(h, cl_f2), cl_h2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}
// exponent e with |m| * 2^e in [2^13, 2^14) (0 for m = 0 or a non-finite m)
__device__ __forceinline__ int cl_exp16(float m) {
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
    int e = 13 - ilogbf(m);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}

// ---- tap order: od / oh / ow = offset in [-2, 2]; k5 / k3 = index into the 5^3 / 3^3 weight block (255 = not a tap of that kernel)
struct ClTaps { signed char od[128], oh[128], ow[128]; unsigned char k5[128], k3[128]; };
constexpr ClTaps cl_make_taps() {
    ClTaps t{};
    for (int i = 0; i < 128; ++i) { t.od[i] = 0; t.oh[i] = 0; t.ow[i] = 0; t.k5[i] = 255; t.k3[i] = 255; }
    for (int i = 0; i < 27; ++i) {
        const int a = i / 9, b = (i / 3) % 3, c = i % 3;
        t.od[i] = (signed char)(a - 1); t.oh[i] = (signed char)(b - 1); t.ow[i] = (signed char)(c - 1);
        t.k3[i] = (unsigned char)i; t.k5[i] = (unsigned char)(((a + 1) * 5 + (b + 1)) * 5 + (c + 1));
    }
    int n = 28;
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            for (int c = 0; c < 5; ++c) {
                if (a >= 1 && a <= 3 && b >= 1 && b <= 3 && c >= 1 && c <= 3) continue;
                t.od[n] = (signed char)(a - 2); t.oh[n] = (signed char)(b - 2); t.ow[n] = (signed char)(c - 2);
                t.k5[n] = (unsigned char)((a * 5 + b) * 5 + c);
                ++n;
            }
    return t;
}
__constant__ ClTaps cl_taps = cl_make_taps();

template <int CG> struct ClK {
    static constexpr int TPS = 32 / CG;                 // taps per MFMA step
    static constexpr int NS5 = CG == 8 ? 32 : 63;       // steps of the K = 5 stream (126 taps)
    static constexpr int NS3 = 28 / TPS;                // the first NS3 steps hold the 3^3 centre
    static constexpr int S1 = 13 / TPS;                 // the step holding the centre tap
    static constexpr int NE = NS5 + NS3 + 1;            // image entries per group and direction
    static constexpr int NH = CG / 8;                   // 8-channel halves of a voxel's channel vector (one LDS array each)
};

// ------------------------------------------------------------------------------------------------------------------ weight images
// esc[g][c] (c = 0, 1, 2 for K = 5, 3, 1): scale exponent of the (group, kernel size) block; grid (G, 3)
__global__ void __launch_bounds__(1024) vx_cl_wmax_k(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ w5, float* __restrict__ esc, int CG) {
    __shared__ float sm[16];
    const int g = blockIdx.x, c = blockIdx.y;
    const int K3 = c == 0 ? 125 : c == 1 ? 27 : 1;
    const float4* __restrict__ wg = reinterpret_cast<const float4*>((c == 0 ? w5 : c == 1 ? w3 : w1) + (long)g * CG * CG * K3);      // (CG^2 K^3 floats: a multiple of 64)
    const int n4 = CG * CG * K3 / 4;
    float mx = 0.0f;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024 * 4) {          // four independent loads per thread in flight
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024; v[u] = wg[i < n4 ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
    }
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x < 16) {
        float m2 = sm[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
        if (threadIdx.x == 0) esc[g * 3 + c] = (float)cl_exp16(m2);
    }
}
// img[dir][g][entry][piece][lane] (uint4 = 8 halfs).  Entry e: K = 5 steps, then K = 3 steps, then the K = 1 step.  A operand lane (row r = lane & 15, group Gl = lane >> 4):
// forward  r = output channel, the 8 halfs = 8 input channels of the lane group's tap;  backward  r = input channel, the 8 halfs = 8 output channels (taps unchanged: the
// kernel negates the offsets).  CG = 16: a lane group = tap (Gl >> 1), channel half (Gl & 1).
template <int CG>
__global__ void __launch_bounds__(256) vx_cl_prep_k(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ w5, cl_u4* __restrict__ img,
                                                    const float* __restrict__ esc, int G) {
    using KK = ClK<CG>;
    const long total = (long)2 * G * KK::NE * 64;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long q = t >> 6;
    const int e = (int)(q % KK::NE); q /= KK::NE;
    const int g = (int)(q % G);
    const int dir = (int)(q / G);
    const int r = lane & 15, Gl = lane >> 4;
    int c, s;
    if (e < KK::NS5) { c = 0; s = e; } else if (e < KK::NS5 + KK::NS3) { c = 1; s = e - KK::NS5; } else { c = 2; s = KK::S1; }
    const int tap = CG == 8 ? 4 * s + Gl : 2 * s + (Gl >> 1);
    const int half = CG == 8 ? 0 : (Gl & 1);
    const int K3 = c == 0 ? 125 : c == 1 ? 27 : 1;
    const float* __restrict__ w = c == 0 ? w5 : c == 1 ? w3 : w1;
    int kidx = c == 0 ? cl_taps.k5[tap] : c == 1 ? cl_taps.k3[tap] : (tap == 13 ? 0 : 255);
    const float sc = ldexpf(1.0f, (int)esc[g * 3 + c]);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int other = 8 * half + j;
        const int co = dir == 0 ? r : other, ci = dir == 0 ? other : r;
        v[j] = (kidx != 255 && r < CG) ? w[((long)(g * CG + co) * CG + ci) * K3 + kidx] * sc : 0.0f;
    }
    uint32_t h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) cl_split2(v[2 * p], v[2 * p + 1], h[p], l[p]);
    const long o = (((long)dir * G + g) * KK::NE + e) * 128 + lane;
    img[o] = (cl_u4){h[0], h[1], h[2], h[3]};
    img[o + 64] = (cl_u4){l[0], l[1], l[2], l[3]};
}

// ------------------------------------------------------------------------------------------------------------------ kernels
struct ClHalo { int P, HH, HW, npos; };        // pad, padded H / W extents, voxels of the padded volume
struct VxCl {
    int B, C, G, D, H, W, V, NT, nchunk;
    ClHalo h5, h3, h1;
};
// one group volume (CG channels x V voxels, channel stride V) -> LDS, channels-last, two fp16 pieces of v * 2^e with e from the block's maximum.  dst: [piece][half][npos] x 16 B,
// zero borders.  Returns e.  (Two passes over the input: the second one hits L1 / L2.)
template <int CG>
__device__ __forceinline__ int cl_stage(unsigned char* __restrict__ dst, const float* __restrict__ src, const VxCl& p, const ClHalo& hl, float* __restrict__ red) {
    constexpr int NH = CG / 8;
    float mx = 0.0f;
    for (int i = threadIdx.x; i < CG * p.V / 4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int i = CG * p.V / 4 * 4 + threadIdx.x; i < CG * p.V; i += 256) mx = fmaxf(mx, fabsf(src[i]));
    mx = vx_wave_max(mx);
    __syncthreads();                          // (red may still be read from the previous call)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    const int nz = 2 * NH * hl.npos;          // zero everything (borders), 16 bytes per item
    for (int i = threadIdx.x; i < nz; i += 256) reinterpret_cast<cl_u4*>(dst)[i] = (cl_u4){0u, 0u, 0u, 0u};
    __syncthreads();
    const int e = cl_exp16(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    const float sc = ldexpf(1.0f, e);
    for (int i = threadIdx.x; i < NH * p.V; i += 256) {
        const int half = i / p.V, v = i - half * p.V;
        const int w = v % p.W, hh = (v / p.W) % p.H, d = v / (p.W * p.H);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = src[(long)(8 * half + j) * p.V + v] * sc;
        uint32_t h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cl_split2(x[2 * q], x[2 * q + 1], h[q], l[q]);
        const int hp = ((d + hl.P) * hl.HH + (hh + hl.P)) * hl.HW + (w + hl.P);
        reinterpret_cast<cl_u4*>(dst)[half * hl.npos + hp] = (cl_u4){h[0], h[1], h[2], h[3]};
        reinterpret_cast<cl_u4*>(dst)[(NH + half) * hl.npos + hp] = (cl_u4){l[0], l[1], l[2], l[3]};
    }
    return e;
}
// acc[t] += A(entry e0 + s) x B(voxel tile t, taps of step s0 + s), s = 0 .. ns-1.  xs: this lane's base in the staged volume (its channel half); hp[t]: byte offset of
// the lane's voxel; toff: byte offsets of the taps in the volume's geometry; sign = +1 forward / -1 backward.
template <int CG, int TPW>
__device__ __forceinline__ void cl_accum(cl_f4 (&acc)[TPW], const cl_u4* __restrict__ img, int ns, int s0, const unsigned char* __restrict__ xs, int lo_off,
                                         const int (&hp)[TPW], const int* __restrict__ toff, int sign, int Gl) {
    cl_u4 ah = img[0], al = img[64];
#pragma unroll 2
    for (int s = 0; s < ns; ++s) {
        cl_u4 nh = ah, nl = al;
        if (s + 1 < ns) { nh = img[(long)(s + 1) * 128]; nl = img[(long)(s + 1) * 128 + 64]; }
        const int tap = CG == 8 ? 4 * (s0 + s) + Gl : 2 * (s0 + s) + (Gl >> 1);
        const int off = sign * toff[tap];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const cl_u4 bh = *reinterpret_cast<const cl_u4*>(xs + hp[t] + off);
            const cl_u4 bl = *reinterpret_cast<const cl_u4*>(xs + lo_off + hp[t] + off);
            acc[t] = CL_MFMA(ah, bh, acc[t]);
            acc[t] = CL_MFMA(ah, bl, acc[t]);
            acc[t] = CL_MFMA(al, bh, acc[t]);
        }
        ah = nh; al = nl;
    }
}
__device__ __forceinline__ void cl_fill_toff(int* __restrict__ toff, const ClHalo& hl) {
    for (int i = threadIdx.x; i < 128; i += 256) toff[i] = ((cl_taps.od[i] * hl.HH + cl_taps.oh[i]) * hl.HW + cl_taps.ow[i]) * 16;
}

template <int CG, int TPW>
__global__ void __launch_bounds__(256) vx_jlc_cl_fwd_k(const float* __restrict__ x, const cl_u4* __restrict__ img, const float* __restrict__ esc,
                                                       const float* __restrict__ b1, const float* __restrict__ b3, const float* __restrict__ b5,
                                                       float* __restrict__ y1, float* __restrict__ y3, float* __restrict__ y5, double* __restrict__ part, VxCl p) {
    using KK = ClK<CG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_lds[];
    int* __restrict__ toff = reinterpret_cast<int*>(cl_lds);                  // [128]
    float* __restrict__ red = reinterpret_cast<float*>(cl_lds + 512);         // [4]
    unsigned char* __restrict__ xs = cl_lds + 1024;
    const int chunk = blockIdx.x, bg = blockIdx.y, b = bg / p.G, g = bg - b * p.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, Gl = lane >> 4;
    cl_fill_toff(toff, p.h5);
    const int ex = cl_stage<CG>(xs, x + ((long)b * p.C + (long)g * CG) * p.V, p, p.h5, red);
    __syncthreads();
    const int half_off = (CG == 16 ? (Gl & 1) : 0) * p.h5.npos * 16, lo_off = KK::NH * p.h5.npos * 16;
    int hp[TPW], pv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = (chunk * 4 + wave) * TPW + t;
        const int v = 16 * tile + n;
        pv[t] = (tile < p.NT && v < p.V) ? v : -1;
        const int vc = v < p.V ? v : p.V - 1;
        const int w = vc % p.W, hh = (vc / p.W) % p.H, d = vc / (p.W * p.H);
        hp[t] = (((d + 2) * p.h5.HH + (hh + 2)) * p.h5.HW + (w + 2)) * 16;
    }
    cl_f4 a5[TPW], a3[TPW], a1[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) { a5[t] = (cl_f4){0.f, 0.f, 0.f, 0.f}; a3[t] = a5[t]; a1[t] = a5[t]; }
    const cl_u4* __restrict__ ig = img + ((long)g * KK::NE) * 128 + lane;
    if ((chunk * 4 + wave) * TPW < p.NT) {
        cl_accum<CG, TPW>(a5, ig, KK::NS5, 0, xs + half_off, lo_off, hp, toff, 1, Gl);
        cl_accum<CG, TPW>(a3, ig + (long)KK::NS5 * 128, KK::NS3, 0, xs + half_off, lo_off, hp, toff, 1, Gl);
        cl_accum<CG, TPW>(a1, ig + (long)(KK::NS5 + KK::NS3) * 128, 1, KK::S1, xs + half_off, lo_off, hp, toff, 1, Gl);
    }
    // ---- results: lane (voxel n of the tile, output channels 4 Gl + i); per-(b, c) partial sums of this wave
    const float f5 = ldexpf(1.0f, -(ex + (int)esc[g * 3 + 0])), f3 = ldexpf(1.0f, -(ex + (int)esc[g * 3 + 1])), f1 = ldexpf(1.0f, -(ex + (int)esc[g * 3 + 2]));
    const long BC = (long)p.B * p.C;
    const int nty = p.nchunk * 4, slot = chunk * 4 + wave;
    const bool rows = 4 * Gl < CG;
    float s[3][4], ss[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = g * CG + (rows ? 4 * Gl + i : 0);
        const float c1 = b1 ? b1[co] : 0.0f, c3 = b3 ? b3[co] : 0.0f, c5 = b5 ? b5[co] : 0.0f;
        s[0][i] = s[1][i] = s[2][i] = 0.0f; ss[0][i] = ss[1][i] = ss[2][i] = 0.0f;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (pv[t] >= 0 && rows) {
                const long o = ((long)b * p.C + co) * p.V + pv[t];
                const float v1 = fmaf(a1[t][i], f1, c1), v3 = fmaf(a3[t][i], f3, c3), v5 = fmaf(a5[t][i], f5, c5);
                y1[o] = v1; y3[o] = v3; y5[o] = v5;
                s[0][i] += v1; ss[0][i] = fmaf(v1, v1, ss[0][i]);
                s[1][i] += v3; ss[1][i] = fmaf(v3, v3, ss[1][i]);
                s[2][i] += v5; ss[2][i] = fmaf(v5, v5, ss[2][i]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { s[k][i] += __shfl_xor(s[k][i], o, 64); ss[k][i] += __shfl_xor(ss[k][i], o, 64); }
            if (n == 0 && rows) {
                double* dst = part + (((long)k * BC + (long)b * p.C + g * CG + 4 * Gl + i) * nty + slot) * 2;
                dst[0] = (double)s[k][i]; dst[1] = (double)ss[k][i];
            }
        }
}

template <int CG, int TPW>
__global__ void __launch_bounds__(256) vx_jlc_cl_bwd_k(const float* __restrict__ g1, const float* __restrict__ g3, const float* __restrict__ g5, const cl_u4* __restrict__ img,
                                                       const float* __restrict__ esc, const float* __restrict__ d_o, float* __restrict__ dx, VxCl p) {
    using KK = ClK<CG>;
    extern __shared__ __attribute__((aligned(16))) unsigned char cl_lds[];
    int* __restrict__ toff5 = reinterpret_cast<int*>(cl_lds);                 // [128]
    int* __restrict__ toff3 = toff5 + 128;                                    // [128] (the 3^3 centre in the pad-1 geometry)
    float* __restrict__ red = reinterpret_cast<float*>(cl_lds + 1024);        // [4]
    unsigned char* __restrict__ x5 = cl_lds + 2048;
    unsigned char* __restrict__ x3 = x5 + (size_t)2 * KK::NH * p.h5.npos * 16;
    unsigned char* __restrict__ x1 = x3 + (size_t)2 * KK::NH * p.h3.npos * 16;
    const int chunk = blockIdx.x, bg = blockIdx.y, b = bg / p.G, g = bg - b * p.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, Gl = lane >> 4;
    cl_fill_toff(toff5, p.h5);
    cl_fill_toff(toff3, p.h3);
    const long gbase = ((long)b * p.C + (long)g * CG) * p.V;
    const int e5 = cl_stage<CG>(x5, g5 + gbase, p, p.h5, red);
    const int e3 = cl_stage<CG>(x3, g3 + gbase, p, p.h3, red);
    const int e1 = cl_stage<CG>(x1, g1 + gbase, p, p.h1, red);
    __syncthreads();
    const int hsel = CG == 16 ? (Gl & 1) : 0;
    int hp5[TPW], hp3[TPW], hp1[TPW], pv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = (chunk * 4 + wave) * TPW + t;
        const int v = 16 * tile + n;
        pv[t] = (tile < p.NT && v < p.V) ? v : -1;
        const int vc = v < p.V ? v : p.V - 1;
        const int w = vc % p.W, hh = (vc / p.W) % p.H, d = vc / (p.W * p.H);
        hp5[t] = (((d + 2) * p.h5.HH + (hh + 2)) * p.h5.HW + (w + 2)) * 16;
        hp3[t] = (((d + 1) * p.h3.HH + (hh + 1)) * p.h3.HW + (w + 1)) * 16;
        hp1[t] = ((d * p.h1.HH + hh) * p.h1.HW + w) * 16;
    }
    cl_f4 acc[TPW], tot[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) tot[t] = (cl_f4){0.f, 0.f, 0.f, 0.f};
    const cl_u4* __restrict__ ig = img + ((long)(p.G + g) * KK::NE) * 128 + lane;          // (direction 1)
    if ((chunk * 4 + wave) * TPW < p.NT) {
        {
            const float f = ldexpf(1.0f, -(e5 + (int)esc[g * 3 + 0]));
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = (cl_f4){0.f, 0.f, 0.f, 0.f};
            cl_accum<CG, TPW>(acc, ig, KK::NS5, 0, x5 + hsel * p.h5.npos * 16, KK::NH * p.h5.npos * 16, hp5, toff5, -1, Gl);
#pragma unroll
            for (int t = 0; t < TPW; ++t) tot[t] += acc[t] * f;
        }
        {
            const float f = ldexpf(1.0f, -(e3 + (int)esc[g * 3 + 1]));
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = (cl_f4){0.f, 0.f, 0.f, 0.f};
            cl_accum<CG, TPW>(acc, ig + (long)KK::NS5 * 128, KK::NS3, 0, x3 + hsel * p.h3.npos * 16, KK::NH * p.h3.npos * 16, hp3, toff3, -1, Gl);
#pragma unroll
            for (int t = 0; t < TPW; ++t) tot[t] += acc[t] * f;
        }
        {
            const float f = ldexpf(1.0f, -(e1 + (int)esc[g * 3 + 2]));
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = (cl_f4){0.f, 0.f, 0.f, 0.f};
            // (pad 0: the centre tap's offset is 0 in every geometry -- toff5[13] = 0 -- and the other taps of the step have zero weights; their addresses stay inside x1
            //  only if they are not applied: the step reads with the offsets of toff1 = all zero)
            cl_accum<CG, TPW>(acc, ig + (long)(KK::NS5 + KK::NS3) * 128, 1, KK::S1, x1 + hsel * p.h1.npos * 16, KK::NH * p.h1.npos * 16, hp1, toff3 + 27, 0, Gl);
#pragma unroll
            for (int t = 0; t < TPW; ++t) tot[t] += acc[t] * f;
        }
    }
    if (4 * Gl < CG) {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (pv[t] >= 0) {
                    const long o = ((long)b * p.C + g * CG + 4 * Gl + i) * p.V + pv[t];
                    dx[o] = tot[t][i] + (d_o ? d_o[o] : 0.0f);
                }
    }
}

// ------------------------------------------------------------------------------------------------------------------ host
static int g_cl_enabled = 1;
static ClHalo cl_halo(int D, int H, int W, int P) {
    ClHalo h;
    h.P = P; h.HH = H + 2 * P;
    h.HW = (W == 4) ? (P == 0 ? 4 : 12) : W + 2 * P;          // W = 4: rows of 4 voxels (64 B) at a stride of 192 B are conflict-free for ds_read_b128
    h.npos = (D + 2 * P) * h.HH * h.HW;
    return h;
}
static bool cl_plan(VxCl& p, int B, int C, int G, int D, int H, int W, int& tpw, size_t& shm_f, size_t& shm_b) {
    if (B <= 0 || G <= 0 || C % G != 0) return false;      // (the enable switch gates vx_jlc_cl_ok only: a block whose forward ran here finishes its backward here)
    const int CG = C / G;
    if ((CG != 8 && CG != 16) || D < 1 || H < 1 || W < 1 || D > 8 || H > 8 || W > 8) return false;
    p.B = B; p.C = C; p.G = G; p.D = D; p.H = H; p.W = W; p.V = D * H * W;
    p.NT = (p.V + 15) / 16;
    tpw = p.NT >= 32 ? 2 : 1;
    p.nchunk = (p.NT + 4 * tpw - 1) / (4 * tpw);
    p.h5 = cl_halo(D, H, W, 2); p.h3 = cl_halo(D, H, W, 1); p.h1 = cl_halo(D, H, W, 0);
    const size_t nh = CG / 8;
    shm_f = 1024 + 2 * nh * p.h5.npos * 16;
    shm_b = 2048 + 2 * nh * ((size_t)p.h5.npos + p.h3.npos + p.h1.npos) * 16;
    return shm_b <= 150 * 1024;
}
extern "C" int vx_jlc_cl_set_enabled(int on) { g_cl_enabled = on ? 1 : 0; return 0; }
extern "C" int vx_jlc_cl_ok(int C, int G, int D, int H, int W) {
    VxCl p; int tpw; size_t sf, sb;
    return (g_cl_enabled && cl_plan(p, 1, C, G, D, H, W, tpw, sf, sb)) ? 1 : 0;
}
extern "C" int vx_jlc_cl_ntiles(int C, int G, int D, int H, int W) {
    VxCl p; int tpw; size_t sf, sb;
    if (!cl_plan(p, 1, C, G, D, H, W, tpw, sf, sb)) return -1;
    return p.nchunk * 4;
}
extern "C" int vx_jlc_cl_img_floats(int C, int G) {
    if (G <= 0 || C % G != 0 || !(C / G == 8 || C / G == 16)) return -1;
    const long ne = C / G == 8 ? ClK<8>::NE : ClK<16>::NE;
    return (int)((long)2 * G * ne * 128 * 4 + 3L * G + 4);          // both directions + the scale exponents
}
static float* cl_esc(float* img, int C, int G) { return img + (long)2 * G * (C / G == 8 ? ClK<8>::NE : ClK<16>::NE) * 128 * 4; }
extern "C" int vx_jlc_cl_prep(const float* w1, const float* w3, const float* w5, float* img, int C, int G, void* stream) {
    VX_REQUIRE(w1 && w3 && w5 && img && G > 0 && C % G == 0 && (C / G == 8 || C / G == 16), "vx_jlc_cl_prep: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int CG = C / G;
    float* esc = cl_esc(img, C, G);
    vx_cl_wmax_k<<<dim3((unsigned)G, 3), dim3(1024), 0, st>>>(w1, w3, w5, esc, CG);
    if (CG == 8) vx_cl_prep_k<8><<<dim3((unsigned)vx_cdiv((long)2 * G * ClK<8>::NE * 64, 256)), dim3(256), 0, st>>>(w1, w3, w5, reinterpret_cast<cl_u4*>(img), esc, G);
    else vx_cl_prep_k<16><<<dim3((unsigned)vx_cdiv((long)2 * G * ClK<16>::NE * 64, 256)), dim3(256), 0, st>>>(w1, w3, w5, reinterpret_cast<cl_u4*>(img), esc, G);
    VX_LAUNCH_CHECK("vx_jlc_cl_prep");
    return 0;
}
template <class K> static void cl_attr(K kernel) { if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError(); }
extern "C" int vx_jlc_cl_fwd(const float* x, const float* img, const float* b1, const float* b3, const float* b5, float* y1, float* y3, float* y5, double* part,
                             int B, int C, int G, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && img && y1 && y3 && y5 && part, "vx_jlc_cl_fwd: null pointer");
    VxCl p; int tpw; size_t sf, sb;
    if (!cl_plan(p, B, C, G, D, H, W, tpw, sf, sb)) VX_FAIL(-3, "vx_jlc_cl_fwd: shape not covered (C=%d G=%d %dx%dx%d)", C, G, D, H, W);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.nchunk, (unsigned)(B * G));
    const float* esc = cl_esc(const_cast<float*>(img), C, G);
    const cl_u4* im = reinterpret_cast<const cl_u4*>(img);
#define CL_F(CG_, T_) { static bool once = false; if (!once) { cl_attr(vx_jlc_cl_fwd_k<CG_, T_>); once = true; } \
        vx_jlc_cl_fwd_k<CG_, T_><<<grid, dim3(256), sf, st>>>(x, im, esc, b1, b3, b5, y1, y3, y5, part, p); }
    if (C / G == 8) { if (tpw == 2) CL_F(8, 2) else CL_F(8, 1) }
    else { if (tpw == 2) CL_F(16, 2) else CL_F(16, 1) }
#undef CL_F
    VX_LAUNCH_CHECK("vx_jlc_cl_fwd");
    return 0;
}
extern "C" int vx_jlc_cl_bwd(const float* g1, const float* g3, const float* g5, const float* img, const float* d_o, float* dx, int B, int C, int G, int D, int H, int W,
                             void* stream) {
    VX_REQUIRE(g1 && g3 && g5 && img && dx, "vx_jlc_cl_bwd: null pointer");
    VxCl p; int tpw; size_t sf, sb;
    if (!cl_plan(p, B, C, G, D, H, W, tpw, sf, sb)) VX_FAIL(-3, "vx_jlc_cl_bwd: shape not covered (C=%d G=%d %dx%dx%d)", C, G, D, H, W);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)p.nchunk, (unsigned)(B * G));
    const float* esc = cl_esc(const_cast<float*>(img), C, G);
    const cl_u4* im = reinterpret_cast<const cl_u4*>(img);
#define CL_B(CG_, T_) { static bool once = false; if (!once) { cl_attr(vx_jlc_cl_bwd_k<CG_, T_>); once = true; } \
        vx_jlc_cl_bwd_k<CG_, T_><<<grid, dim3(256), sb, st>>>(g1, g3, g5, im, esc, d_o, dx, p); }
    if (C / G == 8) { if (tpw == 2) CL_B(8, 2) else CL_B(8, 1) }
    else { if (tpw == 2) CL_B(16, 2) else CL_B(16, 1) }
#undef CL_B
    VX_LAUNCH_CHECK("vx_jlc_cl_bwd");
    return 0;
}
