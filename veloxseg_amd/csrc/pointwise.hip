// 1x1x1 convolutions (channel mixers) for gfx950, fp32 NCDHW.
//   reference call sites: PWA.py:291-298 (q/k/v, mix), attention_utils.py:56-57 (FFN), :141 (PatchMerging reduction),
//   conv_blocks.py:64-68 (JLC channel stage), Encoder.py:334-337 (modal mixer, with the torch.cat of :344-347 folded in),
//   Decoder.py:54-57 (enc2rc on cat(attn_m, enc)), :155-158 (deep-supervision heads).
//
// forward / input-gradient: one thread = one voxel, the weight row it needs is wave-uniform and is fetched with wide
//   scalar loads (s_load_dwordx4) and consumed as SGPR operands of v_fmac -> zero LDS, zero vector weight traffic;
//   activations are read coalesced along V (256 B per wave per channel).
// weight-gradient: dW = dY . X^T is a GEMM whose reduction axis is the voxel axis -> fp32 MFMA (v_mfma_f32_16x16x4_f32,
//   exact fp32 FMA chain), operands straight from global memory as 16-byte loads (the K axis may be permuted freely as
//   long as A and B use the same permutation), one wave per (16 co x 16 ci) tile and voxel chunk, float atomics to flush.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"
#include <type_traits>

typedef float vx_f32x4 __attribute__((ext_vector_type(4)));

// (TX: element type of the INPUT activations -- float, or vx_bf16 for the patchified network input of the bf16 storage mode, round 6)
template <typename TX>
__device__ __forceinline__ const TX* vx_pw_row(const TX* __restrict__ x, const TX* __restrict__ x2, int C1, int Cin, int b, int c, long V) {
    return (c < C1) ? x + ((long)b * C1 + c) * V : x2 + ((long)b * (Cin - C1) + (c - C1)) * V;
}

// Optional epilogue of the forward / input-gradient kernels, for the "1x1 conv -> GELU (+ dropout) -> 1x1 conv" stage of the JLC and FFN blocks
// (conv_blocks.py:64-68, attention_utils.py:56-66): mode 1 stores the pre-activation in aux and writes drop(gelu(.)); mode 2 multiplies the input
// gradient by mask * gelu'(aux).  idx = flat NCDHW index of the destination element = the element index of vx_gelu_drop_fwd / _bwd (same masks).
struct VxPwEpi {
    int mode;
    float* aux;
    VxDrop drop;
    float alpha;
};
__device__ __forceinline__ float vx_pw_epi(const VxPwEpi& e, float val, long idx) {
    if (e.mode == 1) { e.aux[idx] = val; return vx_gelu(val) * vx_drop(e.drop, (uint64_t)idx); }
    if (e.mode == 2) return val * vx_drop(e.drop, (uint64_t)idx) * vx_gelu_grad(e.aux[idx]);
    if (e.mode == 3) return fmaf(e.alpha, e.aux[idx], val * vx_drop(e.drop, (uint64_t)idx));       // residual: alpha * res + drop(conv)
    return val;
}
static inline VxPwEpi vx_no_epi() { VxPwEpi e; e.mode = 0; e.aux = nullptr; e.drop.seed_ptr = nullptr; e.drop.stream = 0; e.drop.p = 0.0f; e.alpha = 1.0f; return e; }

// y[b,co,v] = bias[co] + sum_ci w[co,ci] * x[b,ci,v]          (Cin % 4 == 0)
template <int COT, typename TX = float>
__global__ void __launch_bounds__(256) vx_pw_fwd_k(const TX* __restrict__ x, const TX* __restrict__ x2, int C1, int Cin,
                                                   const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                   int Cout, long V, VxPwEpi epi) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, b = blockIdx.z;
    if (v >= V) return;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = bias ? bias[co0 + j] : 0.0f;
    constexpr int CIB = 16;                       // input channels fetched per batch: 16 independent loads in flight per thread
    for (int ci = 0; ci < Cin; ci += CIB) {
        float xv[CIB];
#pragma unroll
        for (int u = 0; u < CIB; ++u) xv[u] = (ci + u < Cin) ? vx_ld1(vx_pw_row(x, x2, C1, Cin, b, ci + u, V), v) : 0.0f;
#pragma unroll
        for (int u4 = 0; u4 < CIB; u4 += 4) {
            if (ci + u4 < Cin) {                  // wave-uniform
#pragma unroll
                for (int j = 0; j < COT; ++j) {
                    const float4 wv = *reinterpret_cast<const float4*>(w + (long)(co0 + j) * Cin + ci + u4);   // uniform -> s_load_dwordx4
                    acc[j] = fmaf(wv.x, xv[u4 + 0], acc[j]);
                    acc[j] = fmaf(wv.y, xv[u4 + 1], acc[j]);
                    acc[j] = fmaf(wv.z, xv[u4 + 2], acc[j]);
                    acc[j] = fmaf(wv.w, xv[u4 + 3], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const long idx = ((long)b * Cout + co0 + j) * V + v;
        y[idx] = vx_pw_epi(epi, acc[j], idx);
    }
}

// The same product for the large volumes (V >= 16 K voxels per sample: the 32^3 level), where the kernel above spends its time on scalar weight loads between the FMAs
// (64 s_load_dwordx4 + waits per 16 input channels) with 4-byte activation loads: one WAVE per block, 4 consecutive voxels per lane (16-byte loads / stores, every input
// channel's load in flight before the first FMA), the 16 x Cin weight tile transposed in LDS and read as broadcast ds_read_b128.  Cout % 16 == 0, Cin <= 64, V % 4 == 0.
template <int CINB, typename TX = float>
__global__ void __launch_bounds__(64) vx_pw_fwd_v4_k(const TX* __restrict__ x, const TX* __restrict__ x2, int C1, int Cin,
                                                     const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int Cout, long V) {
    __shared__ __attribute__((aligned(16))) float wt[64 * 16];          // [ci][16 output channels]
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f4* gf4;
    const int lane = threadIdx.x;
    const long v = ((long)blockIdx.x * 64 + lane) * 4;
    const int co0 = blockIdx.y * 16, b = blockIdx.z;
    for (int e = lane; e < 16 * Cin; e += 64) { const int co = e / Cin, ci = e - co * Cin; wt[ci * 16 + co] = w[(long)(co0 + co) * Cin + ci]; }
    f4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float bj = bias ? bias[co0 + j] : 0.0f; acc[j] = (f4){bj, bj, bj, bj}; }
    __syncthreads();
    if (v >= V) return;
    for (int c0 = 0; c0 < Cin; c0 += CINB) {
        f4 xv[CINB];
#pragma unroll
        for (int u = 0; u < CINB; ++u) {
            const int c = c0 + u;
            // (address arithmetic on integers + an explicit global address space: a pointer select between x and x2 would make these FLAT loads)
            const unsigned long long base = (c < C1) ? (unsigned long long)(x + ((long)b * C1 + c) * V) : (unsigned long long)(x2 + ((long)b * (Cin - C1) + (c - C1)) * V);
            if constexpr (std::is_same<TX, float>::value) xv[u] = (c < Cin) ? *(gf4)(base + (unsigned long long)v * 4ull) : (f4){0.f, 0.f, 0.f, 0.f};
            else {
                typedef unsigned int u2_ __attribute__((ext_vector_type(2)));
                typedef const __attribute__((address_space(1))) u2_* gu2;
                const u2_ t = (c < Cin) ? *(gu2)(base + (unsigned long long)v * 2ull) : (u2_){0u, 0u};
                xv[u] = (f4){vx_bf16_lo(t[0]), vx_bf16_hi(t[0]), vx_bf16_lo(t[1]), vx_bf16_hi(t[1])};
            }
        }
#pragma unroll
        for (int u = 0; u < CINB; ++u) {
            if (c0 + u < Cin) {
                const f4* wr = reinterpret_cast<const f4*>(wt + (c0 + u) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 w4 = wr[q];
                    acc[4 * q + 0] += w4[0] * xv[u];
                    acc[4 * q + 1] += w4[1] * xv[u];
                    acc[4 * q + 2] += w4[2] * xv[u];
                    acc[4 * q + 3] += w4[3] * xv[u];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) *reinterpret_cast<f4*>(y + ((long)b * Cout + co0 + j) * V + v) = acc[j];
}

// dx[b,ci,v] (=|+=) sum_co w[co,ci] * dy[b,co,v]              (CIT input channels per thread, CIT % 4 == 0)
template <int CIT>
__device__ __forceinline__ void vx_pw_bwd_data_body(const int vbx, const int vby, const int vbz, const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                        float* __restrict__ dx2, int C1, int Cin, int Cout, long V, int accumulate, VxPwEpi epi) {
    const long v = (long)vbx * 256 + threadIdx.x;
    const int ci0 = vby * CIT, b = vbz;
    if (v >= V) return;
    float acc[CIT];
#pragma unroll
    for (int i = 0; i < CIT; ++i) acc[i] = 0.0f;
    const float* __restrict__ dyb = dy + (long)b * Cout * V + v;
    constexpr int COB = 8;                        // output channels fetched per batch
    for (int co0 = 0; co0 < Cout; co0 += COB) {
        float dv[COB];
#pragma unroll
        for (int u = 0; u < COB; ++u) dv[u] = (co0 + u < Cout) ? dyb[(long)(co0 + u) * V] : 0.0f;
#pragma unroll
        for (int u = 0; u < COB; ++u) {
            if (co0 + u < Cout) {                 // wave-uniform
                const float* __restrict__ wr = w + (long)(co0 + u) * Cin + ci0;
#pragma unroll
                for (int i = 0; i < CIT; i += 4) {
                    const float4 wv = *reinterpret_cast<const float4*>(wr + i);
                    acc[i] = fmaf(wv.x, dv[u], acc[i]);
                    acc[i + 1] = fmaf(wv.y, dv[u], acc[i + 1]);
                    acc[i + 2] = fmaf(wv.z, dv[u], acc[i + 2]);
                    acc[i + 3] = fmaf(wv.w, dv[u], acc[i + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CIT; ++i) {
        const int c = ci0 + i;
        float* dst = (c < C1) ? dx + ((long)b * C1 + c) * V + v : dx2 + ((long)b * (Cin - C1) + (c - C1)) * V + v;
        const float o = vx_pw_epi(epi, acc[i], ((long)b * Cin + c) * V + v);          // epilogue only without a concat destination (C1 == Cin)
        if (accumulate) *dst += o; else *dst = o;
    }
}
template <int CIT>
__global__ void __launch_bounds__(256) vx_pw_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                        float* __restrict__ dx2, int C1, int Cin, int Cout, long V, int accumulate, VxPwEpi epi) {
    vx_pw_bwd_data_body<CIT>(blockIdx.x, blockIdx.y, blockIdx.z, dy, w, dx, dx2, C1, Cin, Cout, V, accumulate, epi);
}

// dW[co,ci] += sum_{b,v} dy[b,co,v] x[b,ci,v] ; db[co] += sum dy.   One wave = one (16 co x 16 ci) tile x one voxel chunk; 64 voxels per
// iteration with all eight 16-byte operand loads issued before the 16 MFMAs; the 4 waves of a block (same tile, adjacent chunks) are
// summed through LDS so that a block issues 256 float atomics, not 1024 (with ~2 k waves on one 16x16 tile the atomics on its 256
// addresses, not the loads, were the cost: 56 us for 17 MB).
template <typename TX = float>
__device__ __forceinline__ void vx_pw_wgrad_body(const int vbx, const int vby, const TX* __restrict__ x, const TX* __restrict__ x2, int C1, int Cin,
                                                     const float* __restrict__ dy, int Cout, long V, int B, float* __restrict__ dw,
                                                     float* __restrict__ db, int vox_per_wave, int chunks_per_b, int n_ci_tiles) {
    __shared__ float red[4][4 * 64 + 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long cw = (long)vbx * 4 + wave;
    const bool live = cw < (long)B * chunks_per_b;
    const int b = live ? (int)(cw / chunks_per_b) : 0;
    const long v0 = live ? (cw % chunks_per_b) * (long)vox_per_wave : 0;
    const long v1 = live ? ((v0 + vox_per_wave < V) ? v0 + vox_per_wave : V) : 0;
    const int mt = vby / n_ci_tiles, nt = vby % n_ci_tiles;
    const int r = lane & 15, q = lane >> 4;
    const int co = mt * 16 + r, ci = nt * 16 + r;
    const bool co_ok = co < Cout, ci_ok = ci < Cin;
    const float* __restrict__ arow = dy + ((long)b * Cout + (co_ok ? co : 0)) * V;
    const TX* __restrict__ brow = vx_pw_row(x, x2, C1, Cin, b, ci_ok ? ci : 0, V);
    vx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    const bool vec = (V & 3) == 0;
    long vb = v0;
    if (vec) {
        for (; vb + 64 <= v1; vb += 64) {              // wave-uniform trip count; lane covers vb + 16 s + 4 q .. + 3, s = 0..3
            float4 av[4], bv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const long v = vb + 16 * s4 + 4 * q;
                av[s4] = co_ok ? *reinterpret_cast<const float4*>(arow + v) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[s4] = ci_ok ? vx_ld4(brow, v) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                bsum += (av[s4].x + av[s4].y) + (av[s4].z + av[s4].w);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].x, bv[s4].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].y, bv[s4].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].z, bv[s4].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4].w, bv[s4].w, acc, 0, 0, 0);
            }
        }
    }
    for (; vb < v1; vb += 16) {                        // tail / unaligned volumes
        const long v = vb + 4 * q;
        float a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = v + u < v1;
            a4[u] = (in && co_ok) ? arow[v + u] : 0.0f;
            b4[u] = (in && ci_ok) ? vx_ld1(brow, v + u) : 0.0f;
        }
        bsum += (a4[0] + a4[1]) + (a4[2] + a4[3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u], b4[u], acc, 0, 0, 0);
    }
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[wave][reg * 64 + lane] = acc[reg];
    if (q == 0) red[wave][256 + r] = bsum;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const float s_ = (red[0][reg * 64 + lane] + red[1][reg * 64 + lane]) + (red[2][reg * 64 + lane] + red[3][reg * 64 + lane]);
        const int m = mt * 16 + 4 * q + reg, n = nt * 16 + r;
        if (m < Cout && n < Cin) atomicAdd(dw + (long)m * Cin + n, s_);
    }
    if (db != nullptr && nt == 0 && q == 0 && co_ok)
        atomicAdd(db + co, (red[0][256 + r] + red[1][256 + r]) + (red[2][256 + r] + red[3][256 + r]));
}
template <typename TX = float>
__global__ void __launch_bounds__(256) vx_pw_wgrad_k(const TX* __restrict__ x, const TX* __restrict__ x2, int C1, int Cin,
                                                     const float* __restrict__ dy, int Cout, long V, int B, float* __restrict__ dw,
                                                     float* __restrict__ db, int vox_per_wave, int chunks_per_b, int n_ci_tiles) {
    vx_pw_wgrad_body<TX>(blockIdx.x, blockIdx.y, x, x2, C1, Cin, dy, Cout, V, B, dw, db, vox_per_wave, chunks_per_b, n_ci_tiles);
}

template <int N> using vx_ic3 = std::integral_constant<int, N>;

static int vx_pw_v4 = 1;
extern "C" int vx_pw_conv_set_v4(int on) { vx_pw_v4 = on ? 1 : 0; return 0; }      // A/B knob: the one-wave 16-byte forward kernel for large volumes (vx_pw_fwd_v4_k)
static int vx_pw_conv_fwd_impl(const float* x, const float* x2, int C1, const float* w, const float* bias, float* y,
                               int B, int Cin, int Cout, long V, void* stream, const VxPwEpi& epi, int x_h16 = 0) {
    VX_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0 && V > 0, "vx_pw_conv_fwd: bad args");
    VX_REQUIRE(!x_h16 || (!x2 && epi.mode == 0), "vx_pw_conv_fwd: a 16-bit input takes no concat source and no epilogue");
    VX_REQUIRE(Cin % 4 == 0, "vx_pw_conv_fwd: Cin must be a multiple of 4 (got %d)", Cin);
    if (C1 <= 0 || C1 > Cin) C1 = Cin;
    VX_REQUIRE(C1 == Cin || x2, "vx_pw_conv_fwd: x2 missing");
    VX_REQUIRE(C1 % 4 == 0, "vx_pw_conv_fwd: concat split must be a multiple of 4");
    int T = (Cout % 16 == 0) ? 16 : (Cout % 8 == 0) ? 8 : (Cout % 4 == 0) ? 4 : (Cout % 2 == 0) ? 2 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (vx_pw_v4 && epi.mode == 0 && Cout % 16 == 0 && Cin <= 64 && V % 4 == 0 && V >= 16384) {
        const dim3 g4(vx_cdiv(V, 256), Cout / 16, B);
        if (x_h16) {
            const vx_bf16* xh = reinterpret_cast<const vx_bf16*>(x);
            if (Cin <= 32) vx_pw_fwd_v4_k<32, vx_bf16><<<g4, 64, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V);
            else vx_pw_fwd_v4_k<16, vx_bf16><<<g4, 64, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V);
        } else
        if (Cin <= 32) vx_pw_fwd_v4_k<32><<<g4, 64, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V);
        else vx_pw_fwd_v4_k<16><<<g4, 64, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V);
        VX_LAUNCH_CHECK("vx_pw_conv_fwd (v4)");
        return 0;
    }
    while (T > 1 && (long)vx_cdiv(V, 256) * (Cout / T) * B < 512) T >>= 1;     // small volumes: trade register blocking for more blocks
    dim3 grid(vx_cdiv(V, 256), Cout / T, B);
    if (x_h16) {
        const vx_bf16* xh = reinterpret_cast<const vx_bf16*>(x);
        switch (T) {
            case 16: vx_pw_fwd_k<16, vx_bf16><<<grid, 256, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V, epi); break;
            case 8: vx_pw_fwd_k<8, vx_bf16><<<grid, 256, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V, epi); break;
            case 4: vx_pw_fwd_k<4, vx_bf16><<<grid, 256, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V, epi); break;
            case 2: vx_pw_fwd_k<2, vx_bf16><<<grid, 256, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V, epi); break;
            default: vx_pw_fwd_k<1, vx_bf16><<<grid, 256, 0, st>>>(xh, nullptr, C1, Cin, w, bias, y, Cout, V, epi); break;
        }
        VX_LAUNCH_CHECK("vx_pw_conv_fwd");
        return 0;
    }
    switch (T) {
        case 16: vx_pw_fwd_k<16><<<grid, 256, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V, epi); break;
        case 8: vx_pw_fwd_k<8><<<grid, 256, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V, epi); break;
        case 4: vx_pw_fwd_k<4><<<grid, 256, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V, epi); break;
        case 2: vx_pw_fwd_k<2><<<grid, 256, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V, epi); break;
        default: vx_pw_fwd_k<1><<<grid, 256, 0, st>>>(x, x2, C1, Cin, w, bias, y, Cout, V, epi); break;
    }
    VX_LAUNCH_CHECK("vx_pw_conv_fwd");
    return 0;
}

extern "C" int vx_pw_conv_fwd(const float* x, const float* x2, int C1, const float* w, const float* bias, float* y,
                              int B, int Cin, int Cout, long V, void* stream) {
    return vx_pw_conv_fwd_impl(x, x2, C1, w, bias, y, B, Cin, Cout, V, stream, vx_no_epi());
}
// x_h16 != 0: x is a vx_bf16 array (bf16 storage mode: the patchified network input); no concat source
extern "C" int vx_pw_conv_fwd_h(const void* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, long V, int x_h16, void* stream) {
    return vx_pw_conv_fwd_impl((const float*)x, nullptr, Cin, w, bias, y, B, Cin, Cout, V, stream, vx_no_epi(), x_h16 ? 1 : 0);
}

static int vx_pw_conv_bwd_data_impl(const float* dy, const float* w, float* dx, float* dx2, int C1,
                                    int B, int Cin, int Cout, long V, int accumulate, void* stream, const VxPwEpi& epi) {
    VX_REQUIRE(dy && w && dx && B > 0 && Cin > 0 && Cout > 0 && V > 0, "vx_pw_conv_bwd_data: bad args");
    VX_REQUIRE(Cin % 4 == 0, "vx_pw_conv_bwd_data: Cin must be a multiple of 4 (got %d)", Cin);
    if (C1 <= 0 || C1 > Cin) C1 = Cin;
    VX_REQUIRE(C1 == Cin || dx2, "vx_pw_conv_bwd_data: dx2 missing");
    int T = (Cin % 16 == 0) ? 16 : (Cin % 8 == 0) ? 8 : 4;
    while (T > 4 && (C1 % T)) T >>= 1;
    while (T > 4 && (long)vx_cdiv(V, 256) * (Cin / T) * B < 512) T >>= 1;
    VX_REQUIRE(C1 % T == 0, "vx_pw_conv_bwd_data: concat split must be a multiple of 4");
    dim3 grid(vx_cdiv(V, 256), Cin / T, B);
    hipStream_t st = (hipStream_t)stream;
    switch (T) {
        case 16: vx_pw_bwd_data_k<16><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, accumulate, epi); break;
        case 8: vx_pw_bwd_data_k<8><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, accumulate, epi); break;
        default: vx_pw_bwd_data_k<4><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, accumulate, epi); break;
    }
    VX_LAUNCH_CHECK("vx_pw_conv_bwd_data");
    return 0;
}

extern "C" int vx_pw_conv_bwd_data(const float* dy, const float* w, float* dx, float* dx2, int C1,
                                   int B, int Cin, int Cout, long V, int accumulate, void* stream) {
    return vx_pw_conv_bwd_data_impl(dy, w, dx, dx2, C1, B, Cin, Cout, V, accumulate, stream, vx_no_epi());
}

static thread_local int t_pw_x_h16 = 0;
extern "C" int vx_pw_conv_bwd_weight(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                                     int B, int Cin, int Cout, long V, void* stream);
// x_h16 != 0: x is a vx_bf16 array (the patchified network input of the bf16 storage mode)
extern "C" int vx_pw_conv_bwd_weight_h(const void* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, long V, int x_h16, void* stream) {
    const int prev = t_pw_x_h16;
    t_pw_x_h16 = x_h16 ? 1 : 0;
    const int rc = vx_pw_conv_bwd_weight((const float*)x, nullptr, Cin, dy, dw, db, B, Cin, Cout, V, stream);
    t_pw_x_h16 = prev;
    return rc;
}
extern "C" int vx_pw_conv_bwd_weight(const float* x, const float* x2, int C1, const float* dy, float* dw, float* db,
                                     int B, int Cin, int Cout, long V, void* stream) {
    VX_REQUIRE(x && dy && dw && B > 0 && Cin > 0 && Cout > 0 && V > 0, "vx_pw_conv_bwd_weight: bad args");
    if (C1 <= 0 || C1 > Cin) C1 = Cin;
    VX_REQUIRE(C1 == Cin || x2, "vx_pw_conv_bwd_weight: x2 missing");
    const int mt = vx_cdiv(Cout, 16), nt = vx_cdiv(Cin, 16);
    // voxel chunk per wave: multiple of 16; aim at >= ~2048 waves in flight over all tiles
    long waves_per_tile = 1024 / ((long)mt * nt);
    if (waves_per_tile < 4) waves_per_tile = 4;
    long vpw = ((long)B * V + waves_per_tile - 1) / waves_per_tile;
    vpw = (vpw + 63) / 64 * 64;
    if (vpw < 256) vpw = 256;
    const int chunks_per_b = vx_cdiv(V, vpw);
    dim3 grid(vx_cdiv((long)B * chunks_per_b, 4), mt * nt);
    if (t_pw_x_h16) vx_pw_wgrad_k<vx_bf16><<<grid, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const vx_bf16*>(x), nullptr, C1, Cin, dy, Cout, V, B, dw, db, (int)vpw, chunks_per_b, nt);
    else vx_pw_wgrad_k<float><<<grid, 256, 0, (hipStream_t)stream>>>(x, x2, C1, Cin, dy, Cout, V, B, dw, db, (int)vpw, chunks_per_b, nt);
    VX_LAUNCH_CHECK("vx_pw_conv_bwd_weight");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Patch embedding (a convolution with kernel == stride == 4, no padding; PatchEmbed, Encoder.py:150-156 of the reference) WITHOUT the patchified copy
// (round 6): the product and its weight gradient read the network input in place.  Until round 5 vx_patchify wrote the (Cin 64, Vo)
// channel image (one read + one write of the whole input) and the 1x1 kernels read it twice more; here the 64 "channels" of an output voxel are gathered
// by address: a lane owns 4 consecutive output voxels of one row, whose 16 taps (kw = 0..3 of each) are 64 contiguous bytes of the input row (kd, kh)
// -- four 16-byte loads per row, and the (voxel, kw) transposition is register naming.
//   forward     : one wave per block, 16 output channels, the 16 x Ck weight tile in LDS, channels summed in the order of the patchified product
//                 (bias first, then c = ((cin 4 + kd) 4 + kh) 4 + kw ascending: the results are bit-identical to vx_patchify + vx_pw_fwd_v4_k)
//   weight grad : block = one (cin, kd) and a chunk of voxels, wave = kh: 16 co x 4 kw sums per lane over its voxels (256 FMAs per 20 loads), summed
//                 over the lanes through LDS, one atomic per (co, kw) and wave.  The rows of a chunk sit on one XCD (vx_xcd_rows): its dy is read once from HBM.
// ------------------------------------------------------------------------------------------------------------------
template <int RB>
__global__ void __launch_bounds__(64) vx_patch4_fwd_k(const float* __restrict__ x, long bstride, int Cin, int Do, int Ho, int Wo, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float vx_p4_wt[];      // [Ck][16 output channels]
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f4* gf4;
    const int lane = threadIdx.x;
    const int Ck = Cin * 64;
    const long Vo = (long)Do * Ho * Wo;
    const long v = ((long)blockIdx.x * 64 + lane) * 4;
    const int co0 = blockIdx.y * 16, b = blockIdx.z;
    for (int e = lane; e < 16 * Ck; e += 64) { const int co = e & 15, c = e >> 4; vx_p4_wt[e] = w[(long)(co0 + co) * Ck + c]; }
    f4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float bj = bias ? bias[co0 + j] : 0.0f; acc[j] = (f4){bj, bj, bj, bj}; }
    __syncthreads();
    if (v >= Vo) return;
    const int wo = (int)(v % Wo), ho = (int)((v / Wo) % Ho), dz = (int)(v / ((long)Wo * Ho));
    const long W = 4L * Wo, HW = 4L * Ho * W, Vin = 4L * Do * HW;
    const unsigned long long base = (unsigned long long)(x + (long)b * bstride + 4L * dz * HW + 4L * ho * W + 4L * wo);
    const int nrows = Cin * 16;
    for (int r0 = 0; r0 < nrows; r0 += RB) {
        f4 xr[RB][4];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + u, cin = r >> 4, kd = (r >> 2) & 3, kh = r & 3;
            const unsigned long long p = base + (unsigned long long)((long)cin * Vin + kd * HW + kh * W) * 4ull;
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[u][j] = *(gf4)(p + 16ull * j);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const f4 xv = (f4){xr[u][0][kw], xr[u][1][kw], xr[u][2][kw], xr[u][3][kw]};
                const f4* wr = reinterpret_cast<const f4*>(vx_p4_wt + ((r0 + u) * 4 + kw) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 w4 = wr[q];
                    acc[4 * q + 0] += w4[0] * xv;
                    acc[4 * q + 1] += w4[1] * xv;
                    acc[4 * q + 2] += w4[2] * xv;
                    acc[4 * q + 3] += w4[3] * xv;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) *reinterpret_cast<f4*>(y + ((long)b * Cout + co0 + j) * Vo + v) = acc[j];
}

// grid (Cin * 4 * (Cout / 16), chunks); block 256 = 4 waves (kh); quads of a chunk: [chunk * qpc, (chunk + 1) * qpc) of the B * Vo / 4 voxel quads
__global__ void __launch_bounds__(256) vx_patch4_wgrad_k(const float* __restrict__ x, long bstride, int Cin, int Do, int Ho, int Wo, const float* __restrict__ dy,
                                                         int Cout, int B, float* __restrict__ dw, float* __restrict__ db, long qpc) {
    __shared__ float red[4][64 * 32];
    typedef float f4 __attribute__((ext_vector_type(4)));
    int bx, chunk;
    vx_xcd_rows(bx, chunk);
    const int lane = threadIdx.x & 63;
    const int kh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nrk = Cin * 4;
    const int rk = bx % nrk, cot = bx / nrk;
    const int cin = rk >> 2, kd = rk & 3, co0 = cot * 16;
    const int Ck = Cin * 64;
    const long Vo = (long)Do * Ho * Wo, Q = Vo >> 2, NQ = (long)B * Q;
    const long W = 4L * Wo, HW = 4L * Ho * W, Vin = 4L * Do * HW;
    const bool do_bias = db != nullptr && rk == 0 && kh == 0;
    f4 acc[16];                         // acc[co][kw]
    float bs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[j] = (f4){0.f, 0.f, 0.f, 0.f}; bs[j] = 0.0f; }
    const long q1 = ((long)chunk + 1) * qpc < NQ ? ((long)chunk + 1) * qpc : NQ;
    for (long qd = (long)chunk * qpc + lane; qd < q1; qd += 64) {
        const int b = (int)(qd / Q);
        const long v = (qd - (long)b * Q) * 4;
        const int wo = (int)(v % Wo), ho = (int)((v / Wo) % Ho), dz = (int)(v / ((long)Wo * Ho));
        const float* __restrict__ px = x + (long)b * bstride + (long)cin * Vin + (4L * dz + kd) * HW + (4L * ho + kh) * W + 4L * wo;
        const float* __restrict__ pd = dy + ((long)b * Cout + co0) * Vo + v;
        f4 xr[4], g[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const f4*>(px + 4 * j);
#pragma unroll
        for (int j = 0; j < 16; ++j) g[j] = *reinterpret_cast<const f4*>(pd + (long)j * Vo);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[j] += g[j][u] * xr[u];
        }
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < 16; ++j) bs[j] += (g[j][0] + g[j][1]) + (g[j][2] + g[j][3]);
        }
    }
    // sum over the lanes: halves added by one cross-lane exchange, then [64 sums][32 lanes] through LDS (column index swizzled: conflict-free both ways)
    float* __restrict__ rw = red[kh];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int a = 4 * j + kw;
            const float t = acc[j][kw] + __shfl_xor(acc[j][kw], 32, 64);
            if (lane < 32) rw[a * 32 + (lane ^ (a & 31))] = t;
        }
    __syncthreads();
    {
        float s_ = 0.0f;
        const float* row = rw + lane * 32;
#pragma unroll 8
        for (int l = 0; l < 32; ++l) s_ += row[l ^ (lane & 31)];
        const int j = lane >> 2, kw = lane & 3;
        atomicAdd(dw + (long)(co0 + j) * Ck + ((cin * 4 + kd) * 4 + kh) * 4 + kw, s_);
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float t = vx_wave_sum(bs[j]);
            if (lane == 0) atomicAdd(db + co0 + j, t);
        }
    }
}

// 1 where the fused patch embedding takes the layer (callers fall back to vx_patchify + the 1x1 kernels elsewhere: other patch sizes, ragged rows, tiny grids)
extern "C" int vx_patch_embed_ok(int Cin, int Cout, int Do, int Ho, int Wo, int K) {
    static const int off = getenv("VELOXSEG_PATCH_FUSED") && getenv("VELOXSEG_PATCH_FUSED")[0] == '0';      // (A/B)
    static const long minv = getenv("VELOXSEG_PATCH_FUSED_MINV") ? atol(getenv("VELOXSEG_PATCH_FUSED_MINV")) : 8192;
    return (!off && K == 4 && Cin >= 1 && Cin <= 8 && Cout % 16 == 0 && Wo % 4 == 0 && Do > 0 && Ho > 0 && (long)Do * Ho * Wo >= minv) ? 1 : 0;
}
// y[b, co, z, y, x] = bias[co] + sum w[co, cin, kd, kh, kw] * x[b, cin, 4z + kd, 4y + kh, 4x + kw];  batch_stride: floats between the samples of x (a channel slice
// of a wider tensor is read in place)
extern "C" int vx_patch_embed_fwd(const float* x, long batch_stride, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Do, int Ho, int Wo,
                                  void* stream) {
    VX_REQUIRE(x && w && y && B > 0, "vx_patch_embed_fwd: bad args");
    VX_REQUIRE(vx_patch_embed_ok(Cin, Cout, Do, Ho, Wo, 4) == 1 || (Cin >= 1 && Cin <= 8 && Cout % 16 == 0 && Wo % 4 == 0), "vx_patch_embed_fwd: needs 16 | Cout, 4 | Wo, Cin <= 8");
    VX_REQUIRE(batch_stride >= (long)Cin * 64 * Do * Ho * Wo && batch_stride % 4 == 0 && ((uintptr_t)x & 15) == 0, "vx_patch_embed_fwd: bad batch stride / alignment");
    const long Vo = (long)Do * Ho * Wo;
    const dim3 grid(vx_cdiv(Vo, 256), Cout / 16, B);
    vx_patch4_fwd_k<4><<<grid, 64, (size_t)Cin * 64 * 16 * sizeof(float), (hipStream_t)stream>>>(x, batch_stride, Cin, Do, Ho, Wo, w, bias, y, Cout);
    VX_LAUNCH_CHECK("vx_patch_embed_fwd");
    return 0;
}
// dw[co, cin, kd, kh, kw] += sum_{b, voxels} dy[b, co, v] * x[b, cin, patch of v];  db[co] += sum dy  (db may be null)
extern "C" int vx_patch_embed_bwd_weight(const float* x, long batch_stride, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int Do, int Ho, int Wo,
                                         void* stream) {
    VX_REQUIRE(x && dy && dw && B > 0, "vx_patch_embed_bwd_weight: bad args");
    VX_REQUIRE(Cin >= 1 && Cin <= 8 && Cout % 16 == 0 && Wo % 4 == 0, "vx_patch_embed_bwd_weight: needs 16 | Cout, 4 | Wo, Cin <= 8");
    VX_REQUIRE(batch_stride >= (long)Cin * 64 * Do * Ho * Wo && batch_stride % 4 == 0 && ((uintptr_t)x & 15) == 0, "vx_patch_embed_bwd_weight: bad batch stride / alignment");
    const long NQ = (long)B * Do * Ho * Wo / 4;
    const int nbx = Cin * 4 * (Cout / 16);
    // chunks: about 1024 blocks in all, at least 2 quads per lane, a multiple of 8 chunks (vx_xcd_rows)
    static const long blocks_env = getenv("VELOXSEG_PATCH_WG_BLOCKS") ? atol(getenv("VELOXSEG_PATCH_WG_BLOCKS")) : 256;      // (A/B)
    long chunks = blocks_env / nbx;
    if (chunks > NQ / 128) chunks = NQ / 128;
    chunks = chunks / 8 * 8;
    if (chunks < 8) chunks = 8;
    const long qpc = ((NQ + chunks - 1) / chunks + 63) / 64 * 64;
    chunks = (NQ + qpc - 1) / qpc;
    vx_patch4_wgrad_k<<<dim3((unsigned)nbx, (unsigned)chunks), 256, 0, (hipStream_t)stream>>>(x, batch_stride, Cin, Do, Ho, Wo, dy, Cout, B, dw, db, qpc);
    VX_LAUNCH_CHECK("vx_patch_embed_bwd_weight");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// ConvTranspose3d(k=2, s=2): every output voxel has exactly one (input voxel, tap) -> a pointwise conv to 8*Co channels
// with a depth-to-space store.  One thread = one INPUT voxel x COT output channels x 8 taps.
// ------------------------------------------------------------------------------------------------------------------
template <int COT>
__global__ void __launch_bounds__(256) vx_upconv_fwd_k(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int Ci, int Co, int d, int h, int wd) {
    const long Vi = (long)d * h * wd;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, b = blockIdx.z;
    if (v >= Vi) return;
    float acc[COT][8];
#pragma unroll
    for (int j = 0; j < COT; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[j][t] = bias ? bias[co0 + j] : 0.0f;
    const float* __restrict__ xb = x + (long)b * Ci * Vi + v;
    constexpr int CIB = 8;
    for (int ci = 0; ci < Ci; ci += CIB) {
        float xv[CIB];
#pragma unroll
        for (int u = 0; u < CIB; ++u) xv[u] = (ci + u < Ci) ? xb[(long)(ci + u) * Vi] : 0.0f;
#pragma unroll
        for (int u = 0; u < CIB; ++u) {
            if (ci + u < Ci) {
#pragma unroll
                for (int j = 0; j < COT; ++j) {
                    const float* wp = w + ((long)(ci + u) * Co + co0 + j) * 8;      // uniform -> 2 x s_load_dwordx4
                    const float4 w0 = *reinterpret_cast<const float4*>(wp);
                    const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
                    acc[j][0] = fmaf(w0.x, xv[u], acc[j][0]); acc[j][1] = fmaf(w0.y, xv[u], acc[j][1]);
                    acc[j][2] = fmaf(w0.z, xv[u], acc[j][2]); acc[j][3] = fmaf(w0.w, xv[u], acc[j][3]);
                    acc[j][4] = fmaf(w1.x, xv[u], acc[j][4]); acc[j][5] = fmaf(w1.y, xv[u], acc[j][5]);
                    acc[j][6] = fmaf(w1.z, xv[u], acc[j][6]); acc[j][7] = fmaf(w1.w, xv[u], acc[j][7]);
                }
            }
        }
    }
    const int xw = (int)(v % wd), xh = (int)((v / wd) % h), xd = (int)(v / ((long)wd * h));
    const int H2 = 2 * h, W2 = 2 * wd;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        float* __restrict__ yb = y + (((long)b * Co + co0 + j) * (2 * d)) * (long)H2 * W2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float2* dst = reinterpret_cast<float2*>(yb + ((long)(2 * xd + i) * H2 + (2 * xh + k)) * W2 + 2 * xw);
                *dst = make_float2(acc[j][(i * 2 + k) * 2], acc[j][(i * 2 + k) * 2 + 1]);
            }
    }
}

template <int CIT>
__global__ void __launch_bounds__(256) vx_upconv_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                            int Ci, int Co, int d, int h, int wd) {
    const long Vi = (long)d * h * wd;
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    const int ci0 = blockIdx.y * CIT, b = blockIdx.z;
    if (v >= Vi) return;
    const int xw = (int)(v % wd), xh = (int)((v / wd) % h), xd = (int)(v / ((long)wd * h));
    const int H2 = 2 * h, W2 = 2 * wd;
    float acc[CIT];
#pragma unroll
    for (int i = 0; i < CIT; ++i) acc[i] = 0.0f;
    for (int co = 0; co < Co; ++co) {
        const float* __restrict__ yb = dy + (((long)b * Co + co) * (2 * d)) * (long)H2 * W2;
        float dv[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float2 t = *reinterpret_cast<const float2*>(yb + ((long)(2 * xd + i) * H2 + (2 * xh + k)) * W2 + 2 * xw);
                dv[(i * 2 + k) * 2] = t.x;
                dv[(i * 2 + k) * 2 + 1] = t.y;
            }
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
            const float* wp = w + ((long)(ci0 + i) * Co + co) * 8;
            const float4 w0 = *reinterpret_cast<const float4*>(wp);
            const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
            float s = acc[i];
            s = fmaf(w0.x, dv[0], s); s = fmaf(w0.y, dv[1], s); s = fmaf(w0.z, dv[2], s); s = fmaf(w0.w, dv[3], s);
            s = fmaf(w1.x, dv[4], s); s = fmaf(w1.y, dv[5], s); s = fmaf(w1.z, dv[6], s); s = fmaf(w1.w, dv[7], s);
            acc[i] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < CIT; ++i) dx[((long)b * Ci + ci0 + i) * Vi + v] = acc[i];
}

// ------------------------------------------------------------------------------------------------------------------
// Small-volume 1x1 convolution as an fp32-MFMA GEMM (levels 2-4: V <= 4096 voxels, 32..384 channels): the thread-per-voxel
// kernels above are latency-bound there (one wave = 64 voxels walks the whole channel axis serially).  Here one wave owns a
// 16 (out channels) x 16 (voxels) tile and walks the reduction axis with v_mfma_f32_16x16x4_f32, operands straight from global.
//   dst[b, m, v] = bias[m] + sum_k Wt(m,k) * src[b, k, v],   Wt(m,k) = w[m*wsm + k*wsk]
//   forward: m = co, k = ci, wsm = Cin, wsk = 1;   input gradient: m = ci, k = co, wsm = 1, wsk = Cin.
// src (forward) or dst (input gradient) may be the channel concat of two tensors.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vx_pw_mfma_k(const float* __restrict__ src, const float* __restrict__ src2, int S1,
                                                    const float* __restrict__ w, int wsm, int wsk, const float* __restrict__ bias,
                                                    float* __restrict__ dst, float* __restrict__ dst2, int D1,
                                                    int Mch, int Kch, long V, int B, int n_vt, int accumulate,
                                                    int mode, int cd, int ch, int cw, int ksplit, VxPwEpi epi) {
    // mode 0: plain.  mode 1 (ConvTranspose k2s2 forward): row m = co*8 + tap is stored depth-to-space into y[b][co][2d+i][2h+j][2w+k].
    // mode 2 (ConvTranspose k2s2 input gradient): reduction row k = co*8 + tap is gathered space-to-depth from dy.  (cd,ch,cw) = coarse dims.
    // ksplit = 1: the 4 waves of a block own 4 voxel tiles.  ksplit = 4 (small problems: too few tiles to fill the chip and a long,
    // latency-bound reduction chain per wave): the 4 waves share ONE tile, each walks a quarter of the reduction axis, LDS sums them.
    __shared__ float vx_ksum[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ks = ksplit > 1 ? wave : 0;
    const long tile_raw = ksplit > 1 ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;           // over (b, voxel tile)
    const bool live = tile_raw < (long)B * n_vt;
    if (!live && ksplit == 1) return;
    const long tile = live ? tile_raw : 0;
    const int b = (int)(tile / n_vt);
    const long v0 = (tile % n_vt) * 16;
    const int mt = blockIdx.y;
    const int kper = ksplit > 1 ? ((Kch + 4 * 16 - 1) / (4 * 16)) * 16 : Kch;
    const int kbeg = ks * kper, kend = min(Kch, kbeg + kper);
    const int r = lane & 15, q = lane >> 4;
    const int m_a = mt * 16 + r;                              // A row owned by this lane
    const bool m_ok = m_a < Mch;
    const long v_b = v0 + r;                                  // B column owned by this lane
    const bool v_ok = v_b < V;
    vx_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ wrow = w + (long)(m_ok ? m_a : 0) * wsm;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * s + q;
            const bool k_ok = k < kend;
            av[s] = (m_ok && k_ok) ? wrow[(long)k * wsk] : 0.0f;
            if (mode == 2) {
                float t = 0.0f;
                if (v_ok && k_ok) {
                    const int co = k >> 3, tp = k & 7;
                    const int xw = (int)(v_b % cw), xh = (int)((v_b / cw) % ch), xd = (int)(v_b / ((long)cw * ch));
                    t = src[((((long)b * (Kch >> 3) + co) * (2 * cd) + 2 * xd + (tp >> 2)) * (2 * ch) + 2 * xh + ((tp >> 1) & 1)) * (long)(2 * cw) + 2 * xw + (tp & 1)];
                }
                bv[s] = t;
            } else {
                const float* srow = (k < S1) ? src + ((long)b * S1 + k) * V : src2 + ((long)b * (Kch - S1) + (k - S1)) * V;
                bv[s] = (v_ok && k_ok) ? srow[v_b] : 0.0f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
    }
    if (ksplit > 1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) vx_ksum[wave][reg * 64 + lane] = acc[reg];
        __syncthreads();
        if (wave != 0 || !live) return;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            acc[reg] = (vx_ksum[0][reg * 64 + lane] + vx_ksum[1][reg * 64 + lane]) + (vx_ksum[2][reg * 64 + lane] + vx_ksum[3][reg * 64 + lane]);
    }
    if (!v_ok) return;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int m = mt * 16 + 4 * q + reg;
        if (m < Mch) {
            if (mode == 1) {
                const int co = m >> 3, tp = m & 7;
                const int xw = (int)(v_b % cw), xh = (int)((v_b / cw) % ch), xd = (int)(v_b / ((long)cw * ch));
                dst[((((long)b * (Mch >> 3) + co) * (2 * cd) + 2 * xd + (tp >> 2)) * (2 * ch) + 2 * xh + ((tp >> 1) & 1)) * (long)(2 * cw) + 2 * xw + (tp & 1)] =
                    acc[reg] + (bias ? bias[co] : 0.0f);
            } else {
                float* drow = (m < D1) ? dst + ((long)b * D1 + m) * V : dst2 + ((long)b * (Mch - D1) + (m - D1)) * V;
                const float o = vx_pw_epi(epi, acc[reg] + (bias ? bias[m] : 0.0f), ((long)b * Mch + m) * V + v_b);      // epilogue: D1 == Mch only
                drow[v_b] = accumulate ? drow[v_b] + o : o;
            }
        }
    }
}

// Plain mode, V % 4 == 0: one wave = 16 rows x 64 voxels as FOUR interleaved n-tiles (column r of n-tile j = voxel v0 + 4r + j), so the B
// operands of the four MFMAs of a k-step are ONE 16-byte load per lane (256 contiguous bytes per reduction row and wave instead of 64) and
// the results leave as 16-byte stores; the A operand (weights) is shared by the four MFMAs.  Same ksplit scheme as vx_pw_mfma_k.
__device__ __forceinline__ void vx_pw_mfma4_body(const int vbx, const int vby, const float* __restrict__ src, const float* __restrict__ src2, int S1,
                                                     const float* __restrict__ w, int wsm, int wsk, const float* __restrict__ bias,
                                                     float* __restrict__ dst, float* __restrict__ dst2, int D1,
                                                     int Mch, int Kch, long V, int B, int n_vt, int accumulate, int ksplit, VxPwEpi epi) {
    __shared__ float vx_ksum4[4][16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ks = ksplit > 1 ? wave : 0;
    const long tile_raw = ksplit > 1 ? (long)vbx : (long)vbx * 4 + wave;           // over (b, 64-voxel tile)
    const bool live = tile_raw < (long)B * n_vt;
    if (!live && ksplit == 1) return;
    const long tile = live ? tile_raw : 0;
    const int b = (int)(tile / n_vt);
    const long v0 = (tile % n_vt) * 64;
    const int mt = vby;
    const int r = lane & 15, q = lane >> 4;
    const int m_a = mt * 16 + r;
    const bool m_ok = m_a < Mch;
    const long v_b = v0 + 4 * r;                               // this lane's 4 voxels: v_b .. v_b + 3
    const bool v_ok = v_b < V;                                 // V % 4 == 0: all four or none
    const int kper = ksplit > 1 ? ((Kch + 4 * 16 - 1) / (4 * 16)) * 16 : Kch;
    const int kbeg = ks * kper, kend = min(Kch, kbeg + kper);
    vx_f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ wrow = w + (long)(m_ok ? m_a : 0) * wsm;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float av[4];
        float4 bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * s + q;
            const bool k_ok = k < kend;
            av[s] = (m_ok && k_ok) ? wrow[(long)k * wsk] : 0.0f;
            const float* srow = (k < S1) ? src + ((long)b * S1 + k) * V : src2 + ((long)b * (Kch - S1) + (k - S1)) * V;
            bv[s] = (v_ok && k_ok) ? *reinterpret_cast<const float4*>(srow + v_b) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].w, acc[3], 0, 0, 0);
        }
    }
    if (ksplit > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) vx_ksum4[wave][(j * 4 + reg) * 64 + lane] = acc[j][reg];
        __syncthreads();
        if (wave != 0 || !live) return;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int e = (j * 4 + reg) * 64 + lane;
                acc[j][reg] = (vx_ksum4[0][e] + vx_ksum4[1][e]) + (vx_ksum4[2][e] + vx_ksum4[3][e]);
            }
    }
    if (!v_ok) return;
    // D_j: row = 4q + reg, col = r  ->  out[m = mt*16 + 4q + reg][voxels v_b .. v_b+3] = (acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg])
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int m = mt * 16 + 4 * q + reg;
        if (m < Mch) {
            float* drow = (m < D1) ? dst + ((long)b * D1 + m) * V : dst2 + ((long)b * (Mch - D1) + (m - D1)) * V;
            const float bb = bias ? bias[m] : 0.0f;
            float4 o = make_float4(acc[0][reg] + bb, acc[1][reg] + bb, acc[2][reg] + bb, acc[3][reg] + bb);
            if (epi.mode) {                                   // D1 == Mch
                const long idx = ((long)b * Mch + m) * V + v_b;
                if (epi.mode == 1) *reinterpret_cast<float4*>(epi.aux + idx) = o;
                const float4 ax = epi.mode >= 2 ? *reinterpret_cast<const float4*>(epi.aux + idx) : o;
                if (epi.mode == 3) {
                    o.x = fmaf(epi.alpha, ax.x, o.x * vx_drop(epi.drop, (uint64_t)idx));
                    o.y = fmaf(epi.alpha, ax.y, o.y * vx_drop(epi.drop, (uint64_t)idx + 1));
                    o.z = fmaf(epi.alpha, ax.z, o.z * vx_drop(epi.drop, (uint64_t)idx + 2));
                    o.w = fmaf(epi.alpha, ax.w, o.w * vx_drop(epi.drop, (uint64_t)idx + 3));
                } else if (epi.mode == 1) {
                    o.x = vx_gelu(ax.x) * vx_drop(epi.drop, (uint64_t)idx);
                    o.y = vx_gelu(ax.y) * vx_drop(epi.drop, (uint64_t)idx + 1);
                    o.z = vx_gelu(ax.z) * vx_drop(epi.drop, (uint64_t)idx + 2);
                    o.w = vx_gelu(ax.w) * vx_drop(epi.drop, (uint64_t)idx + 3);
                } else {
                    o.x = o.x * vx_drop(epi.drop, (uint64_t)idx) * vx_gelu_grad(ax.x);
                    o.y = o.y * vx_drop(epi.drop, (uint64_t)idx + 1) * vx_gelu_grad(ax.y);
                    o.z = o.z * vx_drop(epi.drop, (uint64_t)idx + 2) * vx_gelu_grad(ax.z);
                    o.w = o.w * vx_drop(epi.drop, (uint64_t)idx + 3) * vx_gelu_grad(ax.w);
                }
            }
            float4* dp = reinterpret_cast<float4*>(drow + v_b);
            if (accumulate) { const float4 old = *dp; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            *dp = o;
        }
    }
}
__global__ void __launch_bounds__(256) vx_pw_mfma4_k(const float* __restrict__ src, const float* __restrict__ src2, int S1,
                                                     const float* __restrict__ w, int wsm, int wsk, const float* __restrict__ bias,
                                                     float* __restrict__ dst, float* __restrict__ dst2, int D1,
                                                     int Mch, int Kch, long V, int B, int n_vt, int accumulate, int ksplit, VxPwEpi epi) {
    vx_pw_mfma4_body(blockIdx.x, blockIdx.y, src, src2, S1, w, wsm, wsk, bias, dst, dst2, D1, Mch, Kch, V, B, n_vt, accumulate, ksplit, epi);
}

static int vx_pw_mfma4_enabled = 1;
extern "C" int vx_pw_mfma_set_wide(int on) { vx_pw_mfma4_enabled = on ? 1 : 0; return 0; }

// split the reduction axis over the 4 waves of a block when there are too few (tile, wave) pairs to fill 256 CUs x 4 SIMDs and the chain is long
static inline int vx_pw_ksplit(long wave_tiles, int Kch) { return (wave_tiles < 2048 && Kch >= 64) ? 4 : 1; }

static int vx_pw_conv_mfma_impl(const float* src, const float* src2, int S1, const float* w, int transpose_w, const float* bias,
                                float* dst, float* dst2, int D1, int B, int Mch, int Kch, int Cin_of_w, long V, int accumulate, void* stream, const VxPwEpi& epi) {
    VX_REQUIRE(src && w && dst && B > 0 && Mch > 0 && Kch > 0 && V > 0, "vx_pw_conv_mfma: bad args");
    if (S1 <= 0 || S1 > Kch) S1 = Kch;
    if (D1 <= 0 || D1 > Mch) D1 = Mch;
    VX_REQUIRE((S1 == Kch || src2) && (D1 == Mch || dst2), "vx_pw_conv_mfma: second tensor of a concat is missing");
    const int wsm = transpose_w ? 1 : Cin_of_w, wsk = transpose_w ? Cin_of_w : 1;
    if ((V & 3) == 0 && vx_pw_mfma4_enabled) {
        const int n_vt4 = vx_cdiv(V, 64);
        const int ks4 = vx_pw_ksplit((long)B * n_vt4 * vx_cdiv(Mch, 16), Kch);
        dim3 g4(ks4 > 1 ? (unsigned)((long)B * n_vt4) : vx_cdiv((long)B * n_vt4, 4), vx_cdiv(Mch, 16));
        vx_pw_mfma4_k<<<g4, 256, 0, (hipStream_t)stream>>>(src, src2, S1, w, wsm, wsk, bias, dst, dst2, D1, Mch, Kch, V, B, n_vt4, accumulate, ks4, epi);
        VX_LAUNCH_CHECK("vx_pw_conv_mfma");
        return 0;
    }
    const int n_vt = vx_cdiv(V, 16);
    const int ksplit = vx_pw_ksplit((long)B * n_vt * vx_cdiv(Mch, 16), Kch);
    dim3 grid(ksplit > 1 ? (unsigned)((long)B * n_vt) : vx_cdiv((long)B * n_vt, 4), vx_cdiv(Mch, 16));
    vx_pw_mfma_k<<<grid, 256, 0, (hipStream_t)stream>>>(src, src2, S1, w, wsm, wsk, bias, dst, dst2, D1, Mch, Kch, V, B, n_vt, accumulate, 0, 0, 0, 0, ksplit, epi);
    VX_LAUNCH_CHECK("vx_pw_conv_mfma");
    return 0;
}

extern "C" int vx_pw_conv_mfma(const float* src, const float* src2, int S1, const float* w, int transpose_w, const float* bias,
                               float* dst, float* dst2, int D1, int B, int Mch, int Kch, int Cin_of_w, long V, int accumulate, void* stream) {
    return vx_pw_conv_mfma_impl(src, src2, S1, w, transpose_w, bias, dst, dst2, D1, B, Mch, Kch, Cin_of_w, V, accumulate, stream, vx_no_epi());
}

// Backward of a 1x1 conv in ONE launch: blocks [0, n1) compute the input gradient (the 16 x 64 MFMA tiles of vx_pw_mfma4_body on w^T), blocks
// [n1, n1 + n2) the weight / bias gradient (vx_pw_wgrad_body).  The two halves are independent; at the 16^3 .. 4^3 levels each is a 7-11 us
// launch, and ~70 such pairs per step ran back to back on one stream.
__global__ void __launch_bounds__(256) vx_pw_bwd_fused_k(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ dx2,
                                                         int C1, int Cin, int Cout, long V, int B, int n_vt, int accumulate, int ksplit, int n1, int gx1,
                                                         const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ dw, float* __restrict__ db,
                                                         int vox_per_wave, int chunks_per_b, int n_ci_tiles, int gx2) {
    const int id = blockIdx.x;
    if (id < n1) {
        VxPwEpi e; e.mode = 0; e.aux = nullptr; e.drop.seed_ptr = nullptr; e.drop.stream = 0; e.drop.p = 0.0f; e.alpha = 1.0f;
        vx_pw_mfma4_body(id % gx1, id / gx1, dy, nullptr, Cout, w, 1, Cin, nullptr, dx, dx2, C1, Cin, Cout, V, B, n_vt, accumulate, ksplit, e);
    } else {
        const int j = id - n1;
        vx_pw_wgrad_body(j % gx2, j / gx2, x, x2, C1, Cin, dy, Cout, V, B, dw, db, vox_per_wave, chunks_per_b, n_ci_tiles);
    }
}

extern "C" int vx_pw_conv_bwd_fused(const float* dy, const float* w, const float* x, const float* x2, int C1, float* dx, float* dx2, float* dw, float* db,
                                    int B, int Cin, int Cout, long V, int accumulate, void* stream) {
    VX_REQUIRE(dy && w && x && dx && dw && B > 0 && Cin > 0 && Cout > 0 && V > 0, "vx_pw_conv_bwd_fused: bad args");
    if (C1 <= 0 || C1 > Cin) C1 = Cin;
    VX_REQUIRE(C1 == Cin || (x2 && dx2), "vx_pw_conv_bwd_fused: second tensor of a concat is missing");
    if ((V & 3) != 0 || !vx_pw_mfma4_enabled) {          // no wide tiles: the two separate launches
        if (int e = vx_pw_conv_mfma_impl(dy, nullptr, 0, w, 1, nullptr, dx, dx2, C1, B, Cin, Cout, Cin, V, accumulate, stream, vx_no_epi())) return e;
        return vx_pw_conv_bwd_weight(x, x2, C1, dy, dw, db, B, Cin, Cout, V, stream);
    }
    const int n_vt4 = vx_cdiv(V, 64);
    const int ks4 = vx_pw_ksplit((long)B * n_vt4 * vx_cdiv(Cin, 16), Cout);
    const int gx1 = ks4 > 1 ? (int)((long)B * n_vt4) : vx_cdiv((long)B * n_vt4, 4), gy1 = vx_cdiv(Cin, 16);
    const int mt = vx_cdiv(Cout, 16), nt = vx_cdiv(Cin, 16);
    long waves_per_tile = 1024 / ((long)mt * nt);
    if (waves_per_tile < 4) waves_per_tile = 4;
    long vpw = ((long)B * V + waves_per_tile - 1) / waves_per_tile;
    vpw = (vpw + 63) / 64 * 64;
    if (vpw < 256) vpw = 256;
    const int chunks_per_b = vx_cdiv(V, vpw);
    const int gx2 = vx_cdiv((long)B * chunks_per_b, 4), gy2 = mt * nt;
    const long n1 = (long)gx1 * gy1, n2 = (long)gx2 * gy2;
    VX_REQUIRE(n1 + n2 < 0x7fffffffL, "vx_pw_conv_bwd_fused: grid too large");
    vx_pw_bwd_fused_k<<<dim3((unsigned)(n1 + n2)), 256, 0, (hipStream_t)stream>>>(dy, w, dx, dx2, C1, Cin, Cout, V, B, n_vt4, accumulate, ks4, (int)n1, gx1,
                                                                               x, x2, dw, db, (int)vpw, chunks_per_b, nt, gx2);
    VX_LAUNCH_CHECK("vx_pw_conv_bwd_fused");
    return 0;
}

// Several weight gradients (and the folds of LayerNorm-parameter partial sums) in ONE launch: the sink of a fused PWA chain's backward
// (csrc/pwa_fused.hip).  Job j owns blocks [blk0, blk0 + nblk); a weight-gradient job is vx_pw_wgrad_body over its own (x, dy), a fold job adds
// `rows` partial rows of width 2C (dgamma | dbeta, one row per block of the chain kernel) into the two parameter gradients, 64 rows per block.
struct VxWgJob { const float* x; const float* dy; float* dw; float* db; int Cin, Cout; long V; int B, vpw, chunks_per_b, nt, gx, blk0, nblk; };
struct VxFoldJob { const float* part; float* dg; float* dbeta; int C, rows, blk0, nblk; };
#define VX_WGG_JOBS 24
#define VX_WGG_FOLDS 16
struct VxWgGroup { VxWgJob j[VX_WGG_JOBS]; VxFoldJob f[VX_WGG_FOLDS]; int nj, nf; };
__global__ void __launch_bounds__(256) vx_pw_wgrad_group_k(VxWgGroup g) {
    const int id = blockIdx.x;
    for (int k = 0; k < g.nj; ++k) {
        const VxWgJob& J = g.j[k];
        if (id >= J.blk0 && id < J.blk0 + J.nblk) {
            const int l = id - J.blk0;
            vx_pw_wgrad_body<float>(l % J.gx, l / J.gx, J.x, (const float*)nullptr, J.Cin, J.Cin, J.dy, J.Cout, J.V, J.B, J.dw, J.db, J.vpw, J.chunks_per_b, J.nt);
            return;
        }
    }
    for (int k = 0; k < g.nf; ++k) {
        const VxFoldJob& F = g.f[k];
        if (id >= F.blk0 && id < F.blk0 + F.nblk) {
            // 64 rows of width n = 2C per block.  Narrow rows (n < 256: the 32-wide rows of the 32^3 level) are shared by 256 / n thread groups that take every G-th row, and
            // eight loads are in flight per thread: one column per thread walking its 64 rows two at a time was a chain of 32 memory round trips -- the longest block of the
            // launch that closes the encoder backward
            const int r0 = (id - F.blk0) * 64, r1 = min(F.rows, r0 + 64), n = 2 * F.C;
            const int P = n < 256 ? n : 256, G = 256 / P;
            const int ci = threadIdx.x % P, gi = threadIdx.x / P;
            if (gi >= G) return;
            for (int i = ci; i < n; i += P) {
                float acc = 0.0f;
                int rr = r0 + gi;
                for (; rr + 7 * G < r1; rr += 8 * G) {
                    float t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = F.part[(long)(rr + u * G) * n + i];
                    acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
                }
                for (; rr < r1; rr += G) acc += F.part[(long)rr * n + i];
                atomicAdd(i < F.C ? F.dg + i : F.dbeta + (i - F.C), acc);
            }
            return;
        }
    }
}

/* jobs: ptrs 4 per job (x (B,Cin,V), dy (B,Cout,V), dw (Cout,Cin) +=, db (Cout) += or NULL), dims 4 per job (Cin, Cout, V, B);
 * folds: fptrs 3 per fold (part (rows, 2C), dgamma +=, dbeta +=), fdims 2 per fold (C, rows) */
extern "C" int vx_pw_wgrad_group(const void* const* ptrs, const long* dims, int nj, const void* const* fptrs, const int* fdims, int nf, void* stream) {
    VX_REQUIRE(nj >= 0 && nj <= VX_WGG_JOBS && nf >= 0 && nf <= VX_WGG_FOLDS && (nj + nf) > 0 && (nj == 0 || (ptrs && dims)) && (nf == 0 || (fptrs && fdims)), "vx_pw_wgrad_group: bad args");
    VxWgGroup g = {};
    g.nj = nj; g.nf = nf;
    long blk = 0;
    for (int k = 0; k < nj; ++k) {
        VxWgJob& J = g.j[k];
        J.x = (const float*)ptrs[4 * k]; J.dy = (const float*)ptrs[4 * k + 1]; J.dw = (float*)ptrs[4 * k + 2]; J.db = (float*)ptrs[4 * k + 3];
        J.Cin = (int)dims[4 * k]; J.Cout = (int)dims[4 * k + 1]; J.V = dims[4 * k + 2]; J.B = (int)dims[4 * k + 3];
        VX_REQUIRE(J.x && J.dy && J.dw && J.Cin > 0 && J.Cout > 0 && J.V > 0 && J.B > 0, "vx_pw_wgrad_group: bad job %d", k);
        const int mt = vx_cdiv(J.Cout, 16), nt = vx_cdiv(J.Cin, 16);
        static int wgg_waves = -1, wgg_minv = -1;
        if (wgg_waves < 0) { const char* e = getenv("VELOXSEG_WGG_WAVES"); wgg_waves = (e && atoi(e) > 0) ? atoi(e) : 1024; }
        if (wgg_minv < 0) { const char* e = getenv("VELOXSEG_WGG_MINV"); wgg_minv = (e && atoi(e) > 0) ? atoi(e) : 256; }
        long waves_per_tile = wgg_waves / ((long)mt * nt);
        if (waves_per_tile < 4) waves_per_tile = 4;
        long vpw = ((long)J.B * J.V + waves_per_tile - 1) / waves_per_tile;
        vpw = (vpw + 63) / 64 * 64;
        if (vpw < wgg_minv) vpw = wgg_minv;
        J.vpw = (int)vpw; J.chunks_per_b = vx_cdiv(J.V, vpw); J.nt = nt;
        J.gx = vx_cdiv((long)J.B * J.chunks_per_b, 4);
        J.blk0 = (int)blk; J.nblk = J.gx * mt * nt;
        blk += J.nblk;
    }
    for (int k = 0; k < nf; ++k) {
        VxFoldJob& F = g.f[k];
        F.part = (const float*)fptrs[3 * k]; F.dg = (float*)fptrs[3 * k + 1]; F.dbeta = (float*)fptrs[3 * k + 2];
        F.C = fdims[2 * k]; F.rows = fdims[2 * k + 1];
        VX_REQUIRE(F.part && F.dg && F.dbeta && F.C > 0 && F.rows > 0, "vx_pw_wgrad_group: bad fold %d", k);
        F.blk0 = (int)blk; F.nblk = vx_cdiv(F.rows, 64);
        blk += F.nblk;
    }
    VX_REQUIRE(blk < 0x7fffffffL, "vx_pw_wgrad_group: grid too large");
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("VX_WGG_DBG"); dbg = (e && e[0] == '1') ? 1 : 0; }
    if (dbg) {
        fprintf(stderr, "[wgg] %d jobs %d folds %ld blocks:", nj, nf, blk);
        for (int k = 0; k < nj; ++k) fprintf(stderr, " (%d->%d V=%ld B=%d vpw=%d blk=%d)", g.j[k].Cin, g.j[k].Cout, g.j[k].V, g.j[k].B, g.j[k].vpw, g.j[k].nblk);
        for (int k = 0; k < nf; ++k) fprintf(stderr, " [fold C=%d rows=%d]", g.f[k].C, g.f[k].rows);
        fprintf(stderr, "\n");
    }
    vx_pw_wgrad_group_k<<<dim3((unsigned)blk), 256, 0, (hipStream_t)stream>>>(g);
    VX_LAUNCH_CHECK("vx_pw_wgrad_group");
    return 0;
}

// The same one-launch backward for LARGE volumes (one voxel per thread for the input gradient): blocks [0, n1) run vx_pw_bwd_data_body, the rest the
// MFMA weight-gradient body.
template <int CIT>
__global__ void __launch_bounds__(256) vx_pw_bwd_fused_big_k(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, float* __restrict__ dx2,
                                                             int C1, int Cin, int Cout, long V, int B, int accumulate, int n1, int gx1, int gy1,
                                                             const float* __restrict__ x, const float* __restrict__ x2, float* __restrict__ dw, float* __restrict__ db,
                                                             int vox_per_wave, int chunks_per_b, int n_ci_tiles, int gx2) {
    const int id = blockIdx.x;
    if (id < n1) {
        VxPwEpi e; e.mode = 0; e.aux = nullptr; e.drop.seed_ptr = nullptr; e.drop.stream = 0; e.drop.p = 0.0f; e.alpha = 1.0f;
        const int bx = id % gx1, t = id / gx1;
        vx_pw_bwd_data_body<CIT>(bx, t % gy1, t / gy1, dy, w, dx, dx2, C1, Cin, Cout, V, accumulate, e);
    } else {
        const int j = id - n1;
        vx_pw_wgrad_body(j % gx2, j / gx2, x, x2, C1, Cin, dy, Cout, V, B, dw, db, vox_per_wave, chunks_per_b, n_ci_tiles);
    }
}

extern "C" int vx_pw_conv_bwd_fused_big(const float* dy, const float* w, const float* x, const float* x2, int C1, float* dx, float* dx2, float* dw, float* db,
                                        int B, int Cin, int Cout, long V, int accumulate, void* stream) {
    VX_REQUIRE(dy && w && x && dx && dw && B > 0 && Cin > 0 && Cout > 0 && V > 0, "vx_pw_conv_bwd_fused_big: bad args");
    VX_REQUIRE(Cin % 4 == 0, "vx_pw_conv_bwd_fused_big: Cin must be a multiple of 4 (got %d)", Cin);
    if (C1 <= 0 || C1 > Cin) C1 = Cin;
    VX_REQUIRE(C1 == Cin || (x2 && dx2), "vx_pw_conv_bwd_fused_big: second tensor of a concat is missing");
    int T = (Cin % 16 == 0) ? 16 : (Cin % 8 == 0) ? 8 : 4;
    while (T > 4 && (C1 % T)) T >>= 1;
    while (T > 4 && (long)vx_cdiv(V, 256) * (Cin / T) * B < 512) T >>= 1;
    VX_REQUIRE(C1 % T == 0, "vx_pw_conv_bwd_fused_big: concat split must be a multiple of 4");
    const int gx1 = vx_cdiv(V, 256), gy1 = Cin / T;
    const long n1 = (long)gx1 * gy1 * B;
    const int mt = vx_cdiv(Cout, 16), nt = vx_cdiv(Cin, 16);
    long waves_per_tile = 1024 / ((long)mt * nt);
    if (waves_per_tile < 4) waves_per_tile = 4;
    long vpw = ((long)B * V + waves_per_tile - 1) / waves_per_tile;
    vpw = (vpw + 63) / 64 * 64;
    if (vpw < 256) vpw = 256;
    const int chunks_per_b = vx_cdiv(V, vpw);
    const int gx2 = vx_cdiv((long)B * chunks_per_b, 4);
    const long n2 = (long)gx2 * mt * nt;
    VX_REQUIRE(n1 + n2 < 0x7fffffffL, "vx_pw_conv_bwd_fused_big: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(n1 + n2));
    switch (T) {
        case 16: vx_pw_bwd_fused_big_k<16><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, B, accumulate, (int)n1, gx1, gy1, x, x2, dw, db, (int)vpw, chunks_per_b, nt, gx2); break;
        case 8: vx_pw_bwd_fused_big_k<8><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, B, accumulate, (int)n1, gx1, gy1, x, x2, dw, db, (int)vpw, chunks_per_b, nt, gx2); break;
        default: vx_pw_bwd_fused_big_k<4><<<grid, 256, 0, st>>>(dy, w, dx, dx2, C1, Cin, Cout, V, B, accumulate, (int)n1, gx1, gy1, x, x2, dw, db, (int)vpw, chunks_per_b, nt, gx2); break;
    }
    VX_LAUNCH_CHECK("vx_pw_conv_bwd_fused_big");
    return 0;
}

// a = W x + bias (kept for the backward), h = drop(gelu(a)): the first half of "1x1 conv -> GELU -> dropout -> 1x1 conv" in one launch.
// mfma != 0 routes to the MFMA tile kernels (small volumes), else to the one-voxel-per-thread kernel (Cin % 4 == 0).
extern "C" int vx_pw_conv_gelu_fwd(const float* x, const float* w, const float* bias, float* a, float* h, int B, int Cin, int Cout, long V, int mfma,
                                   const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(x && w && a && h && a != h, "vx_pw_conv_gelu_fwd: bad args");
    VxPwEpi e; e.mode = 1; e.aux = a; e.drop = vx_mk_drop(seed_ptr, dstream, p); e.alpha = 1.0f;
    if (mfma) return vx_pw_conv_mfma_impl(x, nullptr, Cin, w, 0, bias, h, nullptr, 0, B, Cout, Cin, Cin, V, 0, stream, e);
    return vx_pw_conv_fwd_impl(x, nullptr, Cin, w, bias, h, B, Cin, Cout, V, stream, e);
}
// da = (W^T dy) * mask * gelu'(a): the input gradient of the SECOND 1x1 conv (w: Cout x Cin, dy: B x Cout x V) with the GELU/dropout backward of
// the stage in its epilogue; a, da: B x Cin x V.
extern "C" int vx_pw_conv_gelu_bwd_data(const float* dy, const float* w, const float* a, float* da, int B, int Cin, int Cout, long V, int mfma,
                                        const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(dy && w && a && da && a != da, "vx_pw_conv_gelu_bwd_data: bad args");
    VxPwEpi e; e.mode = 2; e.aux = const_cast<float*>(a); e.drop = vx_mk_drop(seed_ptr, dstream, p); e.alpha = 1.0f;
    if (mfma) return vx_pw_conv_mfma_impl(dy, nullptr, 0, w, 1, nullptr, da, nullptr, Cin, B, Cin, Cout, Cin, V, 0, stream, e);
    return vx_pw_conv_bwd_data_impl(dy, w, da, nullptr, Cin, B, Cin, Cout, V, 0, stream, e);
}

// out = alpha * res + drop(W x + bias): the residual + dropout that follows the second 1x1 conv of the JLC / FFN stage in the conv epilogue.
// The mask is the one vx_axpy_drop_fwd / _bwd generate for the same (seed_ptr, dstream, p) over the same element indices.
extern "C" int vx_pw_conv_res_fwd(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int Cin, int Cout, long V, int mfma,
                                  float alpha, const void* seed_ptr, unsigned long long dstream, float p, void* stream) {
    VX_REQUIRE(x && w && res && out, "vx_pw_conv_res_fwd: bad args");
    VxPwEpi e; e.mode = 3; e.aux = const_cast<float*>(res); e.drop = vx_mk_drop(seed_ptr, dstream, p); e.alpha = alpha;
    if (mfma) return vx_pw_conv_mfma_impl(x, nullptr, Cin, w, 0, bias, out, nullptr, 0, B, Cout, Cin, Cin, V, 0, stream, e);
    return vx_pw_conv_fwd_impl(x, nullptr, Cin, w, bias, out, B, Cin, Cout, V, stream, e);
}

// ConvTranspose3d(k = 2, s = 2) on the coarse decoder levels (conv_blocks.py:29-35) as the 16 x 64 tiles of vx_pw_mfma4_body (round 5): a lane owns FOUR consecutive
// coarse voxels of one row (wd % 4 == 0), so
//   FWD  (rows m = co * 8 + tap, reduction over ci): the B operand is one 16-byte load of x per reduction row, and the lane's accumulators of the tap pair
//        (.., kk = 0), (.., kk = 1) -- registers reg, reg + 1 -- are EIGHT consecutive fine voxels of y: two 16-byte stores instead of eight scattered 4-byte ones;
//   BWD  (rows m = ci, reduction over k = co * 8 + tap): the B operand of (co, tap) is the stride-2 gather dy[2 w + kk + 2 j]: two 16-byte loads of the 8 fine voxels,
//        even or odd elements picked in registers; the result rows leave as 16-byte stores.
// The 16-voxel tiles of vx_pw_mfma_k did all of this with 4-byte accesses and a division chain per element: 18 launches of 11 - 29 us per pass, 0.32 ms.
template <int BWD>
__global__ void __launch_bounds__(256) vx_upconv_mfma4_k(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ dst,
                                                         int Ci, int Co, int cd, int ch, int cw, int B, int n_vt, int ksplit) {
    __shared__ float vx_ksum4[4][16 * 64];
    const int Mch = BWD ? Ci : Co * 8, Kch = BWD ? Co * 8 : Ci;
    const long V = (long)cd * ch * cw;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ks = ksplit > 1 ? wave : 0;
    const long tile_raw = ksplit > 1 ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave;           // over (b, 64-voxel tile)
    const bool live = tile_raw < (long)B * n_vt;
    if (!live && ksplit == 1) return;
    const long tile = live ? tile_raw : 0;
    const int b = (int)(tile / n_vt);
    const long v0 = (tile % n_vt) * 64;
    const int mt = blockIdx.y;
    const int r = lane & 15, q = lane >> 4;
    const int m_a = mt * 16 + r;
    const bool m_ok = m_a < Mch;
    const long v_b = v0 + 4 * r;                               // this lane's 4 coarse voxels: one row (cw % 4 == 0)
    const bool v_ok = v_b < V;
    const long vc = v_ok ? v_b : 0;
    const int xw = (int)(vc % cw), xh = (int)((vc / cw) % ch), xd = (int)(vc / ((long)cw * ch));
    const long fH = 2L * ch, fW = 2L * cw;
    const long fbase = ((long)(2 * xd) * fH + 2 * xh) * fW + 2 * xw;      // fine voxel (2 xd, 2 xh, 2 xw)
    const int kper = ksplit > 1 ? ((Kch + 4 * 16 - 1) / (4 * 16)) * 16 : Kch;
    const int kbeg = ks * kper, kend = min(Kch, kbeg + kper);
    vx_f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    // A(m, k): forward w[ci = k][co * 8 + tap = m] (row stride 1, k stride Co * 8); backward w[ci = m][k]
    const float* __restrict__ wrow = w + (BWD ? (long)(m_ok ? m_a : 0) * Kch : (long)(m_ok ? m_a : 0));
    const long wsk = BWD ? 1 : (long)Co * 8;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        float av[4];
        float4 bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * s + q;
            const bool k_ok = k < kend;
            av[s] = (m_ok && k_ok) ? wrow[(long)k * wsk] : 0.0f;
            if (BWD) {
                const int kc = k_ok ? k : 0;
                const int co = kc >> 3, tp = kc & 7;
                const float* __restrict__ fp_ = src + ((long)b * Co + co) * (8 * V) + fbase + (long)(tp >> 2) * fH * fW + (long)((tp >> 1) & 1) * fW;
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
                if (v_ok && k_ok) { lo = *reinterpret_cast<const float4*>(fp_); hi = *reinterpret_cast<const float4*>(fp_ + 4); }
                bv[s] = (tp & 1) ? make_float4(lo.y, lo.w, hi.y, hi.w) : make_float4(lo.x, lo.z, hi.x, hi.z);
            } else {
                bv[s] = (v_ok && k_ok) ? *reinterpret_cast<const float4*>(src + ((long)b * Ci + k) * V + v_b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s].w, acc[3], 0, 0, 0);
        }
    }
    if (ksplit > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) vx_ksum4[wave][(j * 4 + reg) * 64 + lane] = acc[j][reg];
        __syncthreads();
        if (wave != 0 || !live) return;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int e = (j * 4 + reg) * 64 + lane;
                acc[j][reg] = (vx_ksum4[0][e] + vx_ksum4[1][e]) + (vx_ksum4[2][e] + vx_ksum4[3][e]);
            }
    }
    if (!v_ok) return;
    if (BWD) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = mt * 16 + 4 * q + reg;
            if (m < Mch) *reinterpret_cast<float4*>(dst + ((long)b * Ci + m) * V + v_b) = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
        }
    } else {
        // registers (reg, reg + 1), reg even = taps (i, jj, kk = 0 / 1) of one output channel: 8 consecutive fine voxels of the row (2 xd + i, 2 xh + jj)
#pragma unroll
        for (int rp = 0; rp < 4; rp += 2) {
            const int m = mt * 16 + 4 * q + rp;
            if (m < Mch) {
                const int co = m >> 3, tp = m & 7;
                const float bb = bias ? bias[co] : 0.0f;
                float* __restrict__ op = dst + ((long)b * Co + co) * (8 * V) + fbase + (long)(tp >> 2) * fH * fW + (long)((tp >> 1) & 1) * fW;
                *reinterpret_cast<float4*>(op) = make_float4(acc[0][rp] + bb, acc[0][rp + 1] + bb, acc[1][rp] + bb, acc[1][rp + 1] + bb);
                *reinterpret_cast<float4*>(op + 4) = make_float4(acc[2][rp] + bb, acc[2][rp + 1] + bb, acc[3][rp] + bb, acc[3][rp + 1] + bb);
            }
        }
    }
}
static int vx_upconv_mfma4_on = -1;
extern "C" int vx_upconv_set_mfma4(int on) { vx_upconv_mfma4_on = on ? 1 : 0; return 0; }
static bool vx_upconv_mfma4_ok(int wd) {
    if (vx_upconv_mfma4_on < 0) { const char* e = getenv("VELOXSEG_UPCONV_MFMA4"); vx_upconv_mfma4_on = (e && e[0] == '0') ? 0 : 1; }
    return vx_upconv_mfma4_on == 1 && (wd & 3) == 0;
}

extern "C" int vx_upconv_k2s2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int d, int h, int wd, void* stream) {
    VX_REQUIRE(x && w && y && B > 0 && Ci > 0 && Co > 0 && d > 0 && h > 0 && wd > 0, "vx_upconv_k2s2_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long Vc = (long)d * h * wd;
    if (Vc <= 4096 && vx_upconv_mfma4_ok(wd)) {   // coarse levels, rows of 4 k voxels: 16 x 64 MFMA tiles with 16-byte accesses
        const int n_vt4 = vx_cdiv(Vc, 64);
        const int ks4 = vx_pw_ksplit((long)B * n_vt4 * vx_cdiv(Co * 8, 16), Ci);
        dim3 g4(ks4 > 1 ? (unsigned)((long)B * n_vt4) : vx_cdiv((long)B * n_vt4, 4), vx_cdiv(Co * 8, 16));
        vx_upconv_mfma4_k<0><<<g4, 256, 0, st>>>(x, w, bias, y, Ci, Co, d, h, wd, B, n_vt4, ks4);
        VX_LAUNCH_CHECK("vx_upconv_k2s2_fwd");
        return 0;
    }
    if (Vc <= 4096) {   // coarse levels: MFMA tiles, rows m = co*8 + tap stored depth-to-space
        const int n_vt = vx_cdiv(Vc, 16);
        const int ksplit = vx_pw_ksplit((long)B * n_vt * vx_cdiv(Co * 8, 16), Ci);
        dim3 g2(ksplit > 1 ? (unsigned)((long)B * n_vt) : vx_cdiv((long)B * n_vt, 4), vx_cdiv(Co * 8, 16));
        vx_pw_mfma_k<<<g2, 256, 0, st>>>(x, nullptr, Ci, w, 1, Co * 8, bias, y, nullptr, Co * 8, Co * 8, Ci, Vc, B, n_vt, 0, 1, d, h, wd, ksplit, vx_no_epi());
        VX_LAUNCH_CHECK("vx_upconv_k2s2_fwd");
        return 0;
    }
    const int T = (Co % 8 == 0) ? 8 : (Co % 4 == 0) ? 4 : (Co % 2 == 0) ? 2 : 1;
    dim3 grid(vx_cdiv((long)d * h * wd, 256), Co / T, B);
    switch (T) {
        case 8: vx_upconv_fwd_k<8><<<grid, 256, 0, st>>>(x, w, bias, y, Ci, Co, d, h, wd); break;
        case 4: vx_upconv_fwd_k<4><<<grid, 256, 0, st>>>(x, w, bias, y, Ci, Co, d, h, wd); break;
        case 2: vx_upconv_fwd_k<2><<<grid, 256, 0, st>>>(x, w, bias, y, Ci, Co, d, h, wd); break;
        default: vx_upconv_fwd_k<1><<<grid, 256, 0, st>>>(x, w, bias, y, Ci, Co, d, h, wd); break;
    }
    VX_LAUNCH_CHECK("vx_upconv_k2s2_fwd");
    return 0;
}

extern "C" int vx_upconv_k2s2_bwd_data(const float* dy, const float* w, float* dx, int B, int Ci, int Co, int d, int h, int wd, void* stream) {
    VX_REQUIRE(dy && w && dx && B > 0 && Ci > 0 && Co > 0 && d > 0 && h > 0 && wd > 0, "vx_upconv_k2s2_bwd_data: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long Vc = (long)d * h * wd;
    if (Vc <= 4096 && vx_upconv_mfma4_ok(wd)) {
        const int n_vt4 = vx_cdiv(Vc, 64);
        const int ks4 = vx_pw_ksplit((long)B * n_vt4 * vx_cdiv(Ci, 16), Co * 8);
        dim3 g4(ks4 > 1 ? (unsigned)((long)B * n_vt4) : vx_cdiv((long)B * n_vt4, 4), vx_cdiv(Ci, 16));
        vx_upconv_mfma4_k<1><<<g4, 256, 0, st>>>(dy, w, nullptr, dx, Ci, Co, d, h, wd, B, n_vt4, ks4);
        VX_LAUNCH_CHECK("vx_upconv_k2s2_bwd_data");
        return 0;
    }
    if (Vc <= 4096) {   // coarse levels: MFMA tiles, reduction rows k = co*8 + tap gathered space-to-depth from dy
        const int n_vt = vx_cdiv(Vc, 16);
        const int ksplit = vx_pw_ksplit((long)B * n_vt * vx_cdiv(Ci, 16), Co * 8);
        dim3 g2(ksplit > 1 ? (unsigned)((long)B * n_vt) : vx_cdiv((long)B * n_vt, 4), vx_cdiv(Ci, 16));
        vx_pw_mfma_k<<<g2, 256, 0, st>>>(dy, nullptr, Co * 8, w, Co * 8, 1, nullptr, dx, nullptr, Ci, Ci, Co * 8, Vc, B, n_vt, 0, 2, d, h, wd, ksplit, vx_no_epi());
        VX_LAUNCH_CHECK("vx_upconv_k2s2_bwd_data");
        return 0;
    }
    const int T = (Ci % 8 == 0) ? 8 : (Ci % 4 == 0) ? 4 : (Ci % 2 == 0) ? 2 : 1;
    dim3 grid(vx_cdiv((long)d * h * wd, 256), Ci / T, B);
    switch (T) {
        case 8: vx_upconv_bwd_data_k<8><<<grid, 256, 0, st>>>(dy, w, dx, Ci, Co, d, h, wd); break;
        case 4: vx_upconv_bwd_data_k<4><<<grid, 256, 0, st>>>(dy, w, dx, Ci, Co, d, h, wd); break;
        case 2: vx_upconv_bwd_data_k<2><<<grid, 256, 0, st>>>(dy, w, dx, Ci, Co, d, h, wd); break;
        default: vx_upconv_bwd_data_k<1><<<grid, 256, 0, st>>>(dy, w, dx, Ci, Co, d, h, wd); break;
    }
    VX_LAUNCH_CHECK("vx_upconv_k2s2_bwd_data");
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// ConvTranspose3d(k = 2, s = 2) weight gradient (reference conv_blocks.py:29-35):  dW[ci][co][tap] += sum_{b, v} x[b, ci, v] * dy[b, co, 2v + tap]
// = a GEMM with M = Ci, N = Co * 8 (channel, tap), K = B * V coarse voxels: 0.03-0.13 GFLOP per layer.  The generic strided-conv weight-gradient kernel
// it used to borrow (x and dy swapped) took 25-70 us per layer; here a block stages a chunk of KC coarse voxels -- x[Ci][KC] and the 16 co x 8 taps fine
// values dy[128][KC] -- in LDS and its 4 waves hold all Ci x 128 outputs of that chunk on v_mfma_f32_16x16x4_f32 (wave w: column tiles 2w, 2w+1, every
// row tile), then add them to dW with float atomics in 64-byte runs.  grid = (K chunks, Co / 16).
// ------------------------------------------------------------------------------------------------------------------
template <int MT>        // row tiles = Ci / 16
__global__ void __launch_bounds__(256) vx_upconv_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int Ci, int Co, int d, int h, int wd,
                                                         int KC, int chunks_per_b) {
    typedef float vx_f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float vx_uw_lds[];
    const int P = KC + 4;                             // row pitch: lanes (r, q) of an operand read fall on 64 different banks
    float* __restrict__ xs = vx_uw_lds;               // [Ci][P]
    float* __restrict__ ds = xs + Ci * P;             // [128 = (co_local, tap)][P]
    const int b = blockIdx.x / chunks_per_b, v0 = (blockIdx.x % chunks_per_b) * KC;
    const int co0 = blockIdx.y * 16;
    const long V = (long)d * h * wd;
    const int H2 = 2 * h, W2 = 2 * wd;
    for (int e = threadIdx.x; e < Ci * KC; e += 256) {
        const int ci = e / KC, kv = e - ci * KC;
        xs[ci * P + kv] = v0 + kv < V ? x[((long)b * Ci + ci) * V + v0 + kv] : 0.0f;
    }
    for (int e = threadIdx.x; e < 128 * KC; e += 256) {
        const int col = e / KC, kv = e - col * KC;
        const int v = v0 + kv;
        float val = 0.0f;
        if (v < V) {
            const int w_ = v % wd, t_ = v / wd, h_ = t_ % h, d_ = t_ / h;
            const int co = co0 + (col >> 3), i = (col >> 2) & 1, j = (col >> 1) & 1, k = col & 1;
            val = dy[(((long)b * Co + co) * (2 * d) + 2 * d_ + i) * (long)H2 * W2 + (long)(2 * h_ + j) * W2 + 2 * w_ + k];
        }
        ds[col * P + kv] = val;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
    vx_f4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m][0] = (vx_f4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (vx_f4){0.f, 0.f, 0.f, 0.f}; }
    const float* __restrict__ bp0 = ds + ((2 * wave) * 16 + r) * P + q;
    const float* __restrict__ bp1 = bp0 + 16 * P;
    const float* __restrict__ ap = xs + r * P + q;
    for (int k0 = 0; k0 < KC; k0 += 4) {
        const float b0 = bp0[k0], b1 = bp1[k0];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float a = ap[m * 16 * P + k0];
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[m][1], 0, 0, 0);
        }
    }
    // acc[m][t][reg] = dW[ci = 16 m + 4 q + reg][col = 16 (2 wave + t) + r]; dw layout (Ci, Co, 8): 16 consecutive columns are 64 contiguous bytes
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = 16 * m + 4 * q + reg, col = 16 * (2 * wave + t) + r;
                atomicAdd(dw + ((long)ci * Co + co0) * 8 + col, acc[m][t][reg]);
            }
}
extern "C" int vx_upconv_k2s2_wgrad_ok(int Ci, int Co) { return (Ci % 16 == 0 && Co % 16 == 0 && Ci >= 16 && Ci <= 128) ? 1 : 0; }
/* dw (Ci, Co, 2, 2, 2) += ; x (B, Ci, d, h, wd), dy (B, Co, 2d, 2h, 2wd) */
extern "C" int vx_upconv_k2s2_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int Co, int d, int h, int wd, void* stream) {
    VX_REQUIRE(x && dy && dw && B > 0 && d > 0 && h > 0 && wd > 0, "vx_upconv_k2s2_wgrad: bad args");
    VX_REQUIRE(vx_upconv_k2s2_wgrad_ok(Ci, Co), "vx_upconv_k2s2_wgrad: unsupported channels Ci=%d Co=%d", Ci, Co);
    const long V = (long)d * h * wd;
    int KC = Ci <= 64 ? 64 : 32;
    while (KC > 16 && (long)B * vx_cdiv(V, (long)KC) * (Co / 16) < 256) KC >>= 1;      // enough blocks to fill the chip on the coarse grids
    const int chunks_per_b = (int)vx_cdiv(V, (long)KC);
    dim3 grid((unsigned)(B * chunks_per_b), (unsigned)(Co / 16));
    const size_t shm = (size_t)(Ci + 128) * (KC + 4) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    switch (Ci / 16) {
#define VX_UW(MT_) case MT_: vx_upconv_wgrad_k<MT_><<<grid, 256, shm, st>>>(x, dy, dw, Ci, Co, d, h, wd, KC, chunks_per_b); break;
        VX_UW(1) VX_UW(2) VX_UW(3) VX_UW(4) VX_UW(5) VX_UW(6) VX_UW(7) VX_UW(8)
#undef VX_UW
    }
    VX_LAUNCH_CHECK("vx_upconv_k2s2_wgrad");
    return 0;
}
