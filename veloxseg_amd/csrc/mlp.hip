// Fused channel MLP for gfx950 on the matrix cores:  out = x + Drop2( W2 . Drop1( GELU( W1 . norm(x) + b1 ) ) + b2 )
//   norm = InstanceNorm with given per-(b,c) statistics  -> the channel stage of a JLC block (reference conv_blocks.py:64-69,74)
//   norm = channels-first LayerNorm (gamma, beta)        -> the FFN tail of a PWA transformer block (PWA.py:437, attention_utils.py:45-71)
// One launch per direction instead of {norm, 1x1 conv + GELU, 1x1 conv + residual} forward and ~6 launches backward, and nothing of size
// R x V (the expanded activation) ever reaches memory: the backward pass recomputes it from x.
//
// MI355X mapping (v_mfma_f32_16x16x4_f32: A lane = (row l%16, k l/16), B lane = (k l/16, col l%16), D reg i = (row 4*(l/16)+i, col l%16)):
//   * a wave owns TPW tiles of 16 voxels (TPW = 4: voxel = v0 + 4*(l%16) + t, every global access is a 16-byte load/store of the four tiles;
//     TPW = 1 for small volumes, so that >= 1 k waves exist);
//   * "N chain": a = W1 . n with the voxels on the COLUMNS.  The accumulator registers of a (rows 4*(l/16)+i) are, register by register, valid
//     B operands of the next GEMM whose reduction runs over those rows: z = W2 . h and, backward, du = W1^T . da chain without any data movement;
//     the weights are the A operands, staged once per block in LDS in operand order (one conflict-free ds_read_b32 per MFMA);
//   * "T pass" (backward): the same products with the voxels on the ROWS (the x / dout registers of the N chain serve as A operands, the same LDS
//     weight images as B operands).  Its accumulators h^T, da^T have the voxel index on l/16 -- exactly the A operand of the weight-gradient
//     GEMMs dW2^T = h . dz^T, dW1 = da . n^T whose reduction runs over voxels; dz^T / n^T come straight from global memory (16-byte loads).
//     So both weight gradients, the bias gradients and the norm statistics of the backward pass are formed in registers, with no LDS transposes.
// Dropout masks are those of vx_gelu_drop_* (site 1, element index in the (B,R,V) tensor) and vx_axpy_drop_* (site 2, (B,C,V)): one Philox call
// per 4 consecutive voxels of a channel (TPW = 4: they sit in one lane; TPW = 1: shared inside a lane quad by a DPP transpose).
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

typedef float vx_f32x4 __attribute__((ext_vector_type(4)));
#define VX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// TI = element type of x and (backward) of the returned input gradient dn: float, or vx_bf16 when the block-internal tensors of a JLC block are kept in bf16
// (round 6, bf16 storage mode: x = o, dn are internal to the block; out / dout are the block's fp32 boundary tensors).  `x` / `dn` are void* typed by the kernel.
struct VxMlp {
    const void* x;
    const float *gamma, *beta, *w1, *b1, *w2, *b2, *dout;
    const double* part;      // NORM 0 forward: partial (sum, sumsq) pairs of x per (b,c): [B*C][nparts][2]; null -> stats holds (mean, rstd)
    float* stats;            // NORM 0: (B*C, 2) mean, rstd -- written by the forward when part != null, read otherwise
    float* out;
    void* dn;
    float *part_out, *dgamma, *dbeta, *dw1, *db1, *dw2, *db2;
    long V;
    int nparts, iters;
    float eps;
    VxDrop d1, d2;
};

// InstanceNorm statistics of this block's sample into LDS (threads 0..C-1): folded from the producer's partial sums, or read as given
template <int C>
__device__ __forceinline__ void vx_mlp_in_stats(const VxMlp& p, bool fold, int b, float* mu_s, float* rs_s, bool writer) {
    const int tid = threadIdx.x;
    if (tid < C) {
        const long bc = (long)b * C + tid;
        float mean, rstd;
        if (fold) {
            double s = 0.0, q = 0.0;
            const double* pp = p.part + bc * p.nparts * 2;
            for (int i = 0; i < p.nparts; ++i) { s += pp[2 * i]; q += pp[2 * i + 1]; }
            const double m = s / (double)p.V;
            double var = q / (double)p.V - m * m;
            var = var < 0.0 ? 0.0 : var;
            mean = (float)m;
            rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            if (writer && p.stats != nullptr) { p.stats[2 * bc] = mean; p.stats[2 * bc + 1] = rstd; }
        } else {
            mean = p.stats[2 * bc];
            rstd = p.stats[2 * bc + 1];
        }
        mu_s[tid] = mean;
        rs_s[tid] = rstd;
    }
}

// LDS weight images in MFMA operand order (index = ((tile, step) * 64 + lane)):
//   A1[(jb*(C/4) + ks)*64 + l] = W1[16jb + l%16][4ks + l/16]          rows j of W1 as A operand (a = W1 n)   == W1^T as B operand of the T pass
//   A2[((cb*(R/16) + rb)*4 + i)*64 + l] = W2[16cb + l%16][16rb + 4(l/16) + i]    A operand of z = W2 h over the accumulator rows of block rb
//   A3[(jb*(C/4) + ks)*64 + l] = W2[4ks + l/16][16jb + l%16]          rows j of W2^T as A operand (dh = W2^T dz) == W2 as B operand of the T pass
//   A4[((cb*(R/16) + rb)*4 + i)*64 + l] = W1[16rb + 4(l/16) + i][16cb + l%16]    A operand of du = W1^T da
template <int C, int R>
__device__ __forceinline__ void vx_mlp_stage_rows(float* __restrict__ dst, const float* __restrict__ w, int ld_row, int ld_col, int nthr) {
    // dst[(jb*(C/4)+ks)*64 + l] = w[(16jb + l%16) * ld_row + (4ks + l/16) * ld_col]; fully unrolled (R*C % 256 == 0) so that every load is in flight
    // before the first LDS store: a rolled loop paid one L2 round trip per element
    (void)nthr;
    float v[R * C / 256];
#pragma unroll
    for (int k = 0; k < R * C / 256; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int l = e & 63, s = e >> 6, ks = s % (C / 4), jb = s / (C / 4);
        v[k] = w[(long)(16 * jb + (l & 15)) * ld_row + (long)(4 * ks + (l >> 4)) * ld_col];
    }
#pragma unroll
    for (int k = 0; k < R * C / 256; ++k) dst[threadIdx.x + 256 * k] = v[k];
}
template <int C, int R>
__device__ __forceinline__ void vx_mlp_stage_acc(float* __restrict__ dst, const float* __restrict__ w, int ld_c, int ld_j, int nthr) {
    // dst[((cb*(R/16)+rb)*4+i)*64 + l] = w[(16cb + l%16) * ld_c + (16rb + 4(l/16) + i) * ld_j]
    (void)nthr;
    float v[R * C / 256];
#pragma unroll
    for (int k = 0; k < R * C / 256; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int l = e & 63, s = e >> 6, i = s & 3, rb = (s >> 2) % (R / 16), cb = (s >> 2) / (R / 16);
        v[k] = w[(long)(16 * cb + (l & 15)) * ld_c + (long)(16 * rb + 4 * (l >> 4) + i) * ld_j];
    }
#pragma unroll
    for (int k = 0; k < R * C / 256; ++k) dst[threadIdx.x + 256 * k] = v[k];
}

template <int TPW, typename T> __device__ __forceinline__ void vx_ldt(const T* __restrict__ p, float (&o)[TPW]) {
    if constexpr (TPW == 4) { const float4 t = vx_ld4(p, 0L); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; }
    else o[0] = vx_ld1(p, 0L);
}
template <int TPW, typename T> __device__ __forceinline__ void vx_stt(T* __restrict__ p, const float (&o)[TPW]) {
    if constexpr (TPW == 4) vx_st4(p, 0L, make_float4(o[0], o[1], o[2], o[3]));
    else vx_st1(p, 0L, o[0]);
}

// N-layout input of a wave's voxel group: n[ks][t] = norm(x)[channel 4ks + q][voxel vc + t]; LN also returns the per-voxel (mean, rstd)
template <int C, int NORM, int TPW, typename TI = float>
__device__ __forceinline__ void vx_mlp_load_n(const TI* __restrict__ xs, long V, long vc, int q, const float* __restrict__ c0, const float* __restrict__ c1, float eps,
                                              float (&n)[C / 4][TPW], float (&uu)[TPW], float (&rr)[TPW]) {
    constexpr int KS = C / 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) vx_ldt<TPW>(xs + (long)(4 * ks + q) * V + vc, n[ks]);
    if (NORM == 1) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            float s = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s += n[ks][t];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            const float u = s / (float)C;
            float qq = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { const float d = n[ks][t] - u; qq = fmaf(d, d, qq); }
            qq += __shfl_xor(qq, 16, 64); qq += __shfl_xor(qq, 32, 64);
            const float r = 1.0f / sqrtf(qq / (float)C + eps);
            uu[t] = u; rr[t] = r;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) n[ks][t] = fmaf(c0[4 * ks + q], (n[ks][t] - u) * r, c1[4 * ks + q]);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float mu = c0[4 * ks + q], rs = c1[4 * ks + q];
#pragma unroll
            for (int t = 0; t < TPW; ++t) n[ks][t] = (n[ks][t] - mu) * rs;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) { uu[t] = 0.0f; rr[t] = 1.0f; }
    }
}

// --------------------------------------------------------------------------------------------------------------------- forward
template <int C, int R, int NORM, int TPW, typename TI = float>
__global__ void __launch_bounds__(256) vx_mlp_fwd_k(VxMlp p) {
    constexpr int KS = C / 4, RB = R / 16, CB = C / 16, VS = TPW, GV = 16 * TPW;
    __shared__ float A1[R * C], A2[R * C], b1s[R], b2s[C], mu_s[C], rs_s[C];
    const int tid = threadIdx.x, b = blockIdx.y, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    vx_mlp_stage_rows<C, R>(A1, p.w1, C, 1, 256);
    vx_mlp_stage_acc<C, R>(A2, p.w2, R, 1, 256);
    for (int e = tid; e < R; e += 256) b1s[e] = p.b1[e];
    if (tid < C) b2s[tid] = p.b2[tid];
    if (NORM == 0) vx_mlp_in_stats<C>(p, p.part != nullptr, b, mu_s, rs_s, blockIdx.x == 0);
    else if (tid < C) { mu_s[tid] = p.gamma[tid]; rs_s[tid] = p.beta[tid]; }
    __syncthreads();
    const long V = p.V;
    const VxDropCtx d1 = vx_drop_ctx(p.d1), d2 = vx_drop_ctx(p.d2);
    const TI* __restrict__ xs = (const TI*)p.x + (long)b * C * V;
    float* __restrict__ os = p.out + (long)b * C * V;
    for (int it = 0; it < p.iters; ++it) {
        const long v0 = (((long)blockIdx.x * p.iters + it) * 4 + wave) * GV;
        if (v0 >= V) break;                                    // wave-uniform
        const long vl = v0 + VS * m;                           // first voxel of this lane (its TPW voxels are consecutive)
        const bool live = vl < V;
        const long vc = live ? vl : V - VS;                    // idle lanes load a valid address and store nothing
        float n[KS][TPW], uu[TPW], rr[TPW];
        vx_mlp_load_n<C, NORM, TPW, TI>(xs, V, vc, q, mu_s, rs_s, p.eps, n, uu, rr);
        vx_f32x4 z[CB][TPW];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) z[cb][t][i] = b2s[16 * cb + 4 * q + i];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            vx_f32x4 a[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[t][i] = b1s[16 * rb + 4 * q + i];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float wa = A1[(rb * KS + ks) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) a[t] = VX_MFMA(wa, n[ks][t], a[t]);
            }
            // h = drop1(gelu(a)) in place
            if constexpr (TPW == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float m1[4];
                    vx_masks_vox4(d1, (uint64_t)b * R + 16 * rb + 4 * q + i, V, vc, m1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[t][i] = vx_gelu_fast(a[t][i]) * m1[t];
                }
            } else {
                float m1[4];
                vx_masks_rows4(d1, (uint64_t)b * R + 16 * rb + 4 * q, 1, V, vc, m1);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[0][i] = vx_gelu_fast(a[0][i]) * m1[i];
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float wa = A2[((cb * RB + rb) * 4 + i) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < TPW; ++t) z[cb][t] = VX_MFMA(wa, a[t][i], z[cb][t]);
                }
        }
        // out = x + drop2(z), in the accumulator layout: channel 16cb + 4q + i, voxels vl .. vl + TPW - 1
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            float m2r[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (TPW == 1) vx_masks_rows4(d2, (uint64_t)b * C + 16 * cb + 4 * q, 1, V, vc, m2r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = 16 * cb + 4 * q + i;
                float res[TPW], o[TPW];
                vx_ldt<TPW>(xs + (long)ch * V + vc, res);
                if constexpr (TPW == 4) {
                    float m2[4];
                    vx_masks_vox4(d2, (uint64_t)b * C + ch, V, vc, m2);
#pragma unroll
                    for (int t = 0; t < 4; ++t) o[t] = res[t] + z[cb][t][i] * m2[t];
                } else o[0] = res[0] + z[cb][0][i] * m2r[i];
                if (live) vx_stt<TPW>(os + (long)ch * V + vl, o);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------- backward
template <int C, int R, int NORM, int TPW, typename TI = float>
__global__ void __launch_bounds__(256) vx_mlp_bwd_k(VxMlp p) {
    constexpr int KS = C / 4, RB = R / 16, CB = C / 16, VS = TPW, GV = 16 * TPW;
    extern __shared__ __attribute__((aligned(16))) float vx_mlp_lds[];
    float* __restrict__ A1 = vx_mlp_lds;
    float* __restrict__ A3 = A1 + R * C;
    float* __restrict__ A4 = A3 + R * C;
    float* __restrict__ b1s = A4 + R * C;
    float* __restrict__ mu_s = b1s + R;       // IN: mean   | LN: gamma
    float* __restrict__ rs_s = mu_s + C;      // IN: rstd   | LN: beta
    float* __restrict__ red = rs_s + C;       // flush area: [R*C] dW2^T tiles, [R*C] dW1 tiles, [4][R + 3C] row sums
    const int tid = threadIdx.x, b = blockIdx.y, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    {
    vx_mlp_stage_rows<C, R>(A1, p.w1, C, 1, 256);           // W1[j][c]
    vx_mlp_stage_rows<C, R>(A3, p.w2, 1, R, 256);           // W2[c][j] read as [j][c]
    vx_mlp_stage_acc<C, R>(A4, p.w1, 1, C, 256);            // W1[j][c] read as [c][j]
    }
    for (int e = tid; e < R; e += 256) b1s[e] = p.b1[e];
    if (NORM == 0) vx_mlp_in_stats<C>(p, false, b, mu_s, rs_s, false);
    else if (tid < C) { mu_s[tid] = p.gamma[tid]; rs_s[tid] = p.beta[tid]; }
    __syncthreads();
    const long V = p.V;
    const VxDropCtx d1 = vx_drop_ctx(p.d1), d2 = vx_drop_ctx(p.d2);
    const TI* __restrict__ xs = (const TI*)p.x + (long)b * C * V;
    const float* __restrict__ gs = p.dout + (long)b * C * V;
    TI* __restrict__ os = (TI*)p.dn + (long)b * C * V;

    vx_f32x4 aW2[RB][CB], aW1[RB][CB];                      // D tiles: reg i = (j = 16jb + 4q + i, c = 16cb + m)
    float sb1[RB], sb2[CB], S1[CB][4], S2[CB][4];           // lane partials: db1 (j = 16jb + m), db2 (c = 16cb + m), norm sums (c = 16cb + 4q + i)
#pragma unroll
    for (int jb = 0; jb < RB; ++jb) {
        sb1[jb] = 0.0f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) { aW2[jb][cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f}; aW1[jb][cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        sb2[cb] = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { S1[cb][i] = 0.0f; S2[cb][i] = 0.0f; }
    }

    // The dropout masks are needed twice per element: in the N layout (voxels on the MFMA columns: the input-gradient chain) and in the T layout (voxels on the rows:
    // the weight-gradient products).  Drawing them twice was 32 Philox calls per lane and trip -- over a third of this kernel's issue slots.  They are drawn once, in the
    // N layout; every lane leaves its keep bits (bit 16 rb + 4 i + t: row 16 rb + 4 q + i of site 1, bit 16 RB + 4 ks + t: channel 4 ks + q of site 2; voxel vl + t) in
    // a per-wave LDS record, and the T pass picks the nibbles of its own (row, 4 voxels) out of the records of the four lanes that drew them.  (`red` is idle until
    // the flush; LDS operations of one wave are executed in order.)
    constexpr int MB2 = 16 * RB, MDW = (16 * RB + 4 * KS + 31) / 32;
    uint32_t* __restrict__ mrec = reinterpret_cast<uint32_t*>(red) + wave * 64 * MDW;      // [MDW][64 lanes]
    for (int it = 0; it < p.iters; ++it) {
        const long v0 = (((long)blockIdx.x * p.iters + it) * 4 + wave) * GV;
        if (v0 >= V) break;                                    // wave-uniform
        const long vl = v0 + VS * m;
        const bool live = vl < V;
        const long vc = live ? vl : V - VS;
        const float lz = live ? 1.0f : 0.0f;
        uint32_t rec[MDW];
#pragma unroll
        for (int k = 0; k < MDW; ++k) rec[k] = 0u;
        // ---- N-layout operands: channel 4ks + q, voxels vl .. vl + TPW - 1
        float n[KS][TPW], dz[KS][TPW], uu[TPW], rr[TPW];
        vx_mlp_load_n<C, NORM, TPW, TI>(xs, V, vc, q, mu_s, rs_s, p.eps, n, uu, rr);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) vx_ldt<TPW>(gs + (long)(4 * ks + q) * V + vc, dz[ks]);
        if constexpr (TPW == 4) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float m2[4];
                vx_masks_vox4(d2, (uint64_t)b * C + 4 * ks + q, V, vc, m2);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    dz[ks][t] *= m2[t] * lz;
                    rec[(MB2 + 4 * ks + t) >> 5] |= (m2[t] != 0.0f ? 1u : 0u) << ((MB2 + 4 * ks + t) & 31);
                }
            }
        } else {
#pragma unroll
            for (int k0 = 0; k0 < KS; k0 += 4) {
                float m2[4];
                vx_masks_rows4(d2, (uint64_t)b * C + 4 * k0 + q, 4, V, vc, m2);
#pragma unroll
                for (int s = 0; s < 4; ++s) dz[k0 + s][0] *= m2[s] * lz;
            }
        }
        // ---- N chain: du = W1^T ( (W2^T dz) * mask1 * gelu'(a) ),  a = W1 n + b1
        vx_f32x4 du[CB][TPW];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int t = 0; t < TPW; ++t) du[cb][t] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            vx_f32x4 a[TPW], dh[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                dh[t] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) a[t][i] = b1s[16 * rb + 4 * q + i];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float w1a = A1[(rb * KS + ks) * 64 + lane], w2a = A3[(rb * KS + ks) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) { a[t] = VX_MFMA(w1a, n[ks][t], a[t]); dh[t] = VX_MFMA(w2a, dz[ks][t], dh[t]); }
            }
            if constexpr (TPW == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float m1[4];
                    vx_masks_vox4(d1, (uint64_t)b * R + 16 * rb + 4 * q + i, V, vc, m1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        dh[t][i] *= m1[t] * vx_gelu_grad_fast(a[t][i]);
                        rec[(16 * rb + 4 * i + t) >> 5] |= (m1[t] != 0.0f ? 1u : 0u) << ((16 * rb + 4 * i + t) & 31);
                    }
                }
            } else {
                float m1[4];
                vx_masks_rows4(d1, (uint64_t)b * R + 16 * rb + 4 * q, 1, V, vc, m1);
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[0][i] *= m1[i] * vx_gelu_grad_fast(a[0][i]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float wa = A4[((cb * RB + rb) * 4 + i) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < TPW; ++t) du[cb][t] = VX_MFMA(wa, dh[t][i], du[cb][t]);
                }
        }
        // ---- norm backward (+ residual) in the accumulator layout: channel 16cb + 4q + i, voxels vl ..
        if (NORM == 1) {
            float xh[CB][4][TPW], s1[TPW], s2[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) { s1[t] = 0.0f; s2[t] = 0.0f; }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * cb + 4 * q + i;
                    const float g = mu_s[ch];
                    vx_ldt<TPW>(xs + (long)ch * V + vc, xh[cb][i]);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        xh[cb][i][t] = (xh[cb][i][t] - uu[t]) * rr[t];
                        const float d = du[cb][t][i];
                        S1[cb][i] += d;
                        S2[cb][i] = fmaf(d, xh[cb][i][t], S2[cb][i]);
                        s1[t] = fmaf(d, g, s1[t]);
                        s2[t] = fmaf(d * g, xh[cb][i][t], s2[t]);
                    }
                }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                s1[t] += __shfl_xor(s1[t], 16, 64); s1[t] += __shfl_xor(s1[t], 32, 64);
                s2[t] += __shfl_xor(s2[t], 16, 64); s2[t] += __shfl_xor(s2[t], 32, 64);
                s1[t] /= (float)C; s2[t] /= (float)C;
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * cb + 4 * q + i;
                    const float g = mu_s[ch];
                    float go[TPW], o[TPW];
                    vx_ldt<TPW>(gs + (long)ch * V + vc, go);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) o[t] = go[t] + rr[t] * (du[cb][t][i] * g - s1[t] - xh[cb][i][t] * s2[t]);
                    if (live) vx_stt<TPW>(os + (long)ch * V + vl, o);
                }
        } else {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * cb + 4 * q + i;
                    const float mu = mu_s[ch], rs = rs_s[ch];
                    float xv[TPW], o[TPW];
                    vx_ldt<TPW>(xs + (long)ch * V + vc, xv);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const float d = du[cb][t][i];
                        o[t] = d;
                        S1[cb][i] += d;
                        S2[cb][i] = fmaf(d, (xv[t] - mu) * rs, S2[cb][i]);
                    }
                    if (live) vx_stt<TPW>(os + (long)ch * V + vl, o);
                }
        }
        // ---- T pass: voxels on the rows.  Tile t = voxels {v0 + VS*m' + t}; accumulator reg i <-> m' = 4q + i <-> voxel v0 + VS*(4q+i) + t.
        //      T-layout global operands: channel 16cb + m, the lane's 4*TPW consecutive voxels v0 + 4*VS*q ..
        const long vt = v0 + 4 * VS * q;                       // multiple of 4
        if constexpr (TPW == 4) {
            if (d1.on || d2.on) {
#pragma unroll
                for (int k = 0; k < MDW; ++k) mrec[k * 64 + lane] = rec[k];
                __builtin_amdgcn_wave_barrier();
            }
        }
        float dzT[CB][TPW][4], nT[CB][TPW][4];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int ch = 16 * cb + m;
            const float c0 = mu_s[ch], c1 = rs_s[ch];
#pragma unroll
            for (int f = 0; f < TPW; ++f) {                    // f-th float4 of the lane's run: TPW = 4 -> (i = f, t = component); TPW = 1 -> (t = 0, i = component)
                const long vf = vt + 4 * f;
                const bool lf = vf < V;
                const long va = lf ? vf : V - 4;
                const float4 g4 = *reinterpret_cast<const float4*>(gs + (long)ch * V + va);
                const float4 x4 = vx_ld4(xs, (long)ch * V + va);
                float m2[4];
                if constexpr (TPW == 4) {
                    // voxels v0 + 4 (4 q + f) + e of channel 16 cb + m = 4 ks + q': drawn by lane (m' = 4 q + f, q' = m & 3), ks = 4 cb + (m >> 2)
                    if (d2.on) {
                        const uint32_t nib = mrec[((MB2 + 16 * cb) >> 5) * 64 + 16 * (m & 3) + 4 * q + f] >> (((MB2 + 16 * cb) & 31) + 4 * (m >> 2));
#pragma unroll
                        for (int e = 0; e < 4; ++e) m2[e] = ((nib >> e) & 1u) ? d2.inv_keep : 0.0f;
                    } else { m2[0] = m2[1] = m2[2] = m2[3] = 1.0f; }
                } else vx_masks_vox4(d2, (uint64_t)b * C + ch, V, va, m2);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = TPW == 4 ? e : 0, i = TPW == 4 ? f : e;
                    const float dzv = lf ? gv[e] * m2[e] : 0.0f;
                    dzT[cb][t][i] = dzv;
                    sb2[cb] += dzv;
                    float nv;
                    if (NORM == 1) {
                        // the per-voxel statistics live in the N layout at the lanes with l%16 == 4q + i: fetch them from lane 4q + i
                        const float u = __shfl(uu[t], 4 * q + i, 64), r = __shfl(rr[t], 4 * q + i, 64);
                        nv = fmaf(c0, (xv[e] - u) * r, c1);
                    } else nv = (xv[e] - c0) * c1;
                    nT[cb][t][i] = nv;
                }
            }
        }
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            vx_f32x4 aT[TPW], hT[TPW];
            const float bj = b1s[16 * jb + m];
#pragma unroll
            for (int t = 0; t < TPW; ++t) { aT[t] = (vx_f32x4){bj, bj, bj, bj}; hT[t] = (vx_f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float w1b = A1[(jb * KS + ks) * 64 + lane], w2b = A3[(jb * KS + ks) * 64 + lane];
#pragma unroll
                for (int t = 0; t < TPW; ++t) { aT[t] = VX_MFMA(n[ks][t], w1b, aT[t]); hT[t] = VX_MFMA(dz[ks][t], w2b, hT[t]); }     // hT holds dh^T for now
            }
            // element (j = 16jb + m, voxel v0 + VS*(4q+i) + t): h^T = gelu(a) m1 -> aT ; da^T = dh m1 gelu'(a) -> hT
#pragma unroll
            for (int f = 0; f < TPW; ++f) {
                const long vf = vt + 4 * f;
                const long va = vf < V ? vf : V - 4;
                float m1[4];
                if constexpr (TPW == 4) {
                    // row 16 jb + m = 16 rb + 4 q' + i: drawn by lane (m' = 4 q + f, q' = m >> 2) with i = m & 3
                    if (d1.on) {
                        const uint32_t nib = mrec[((16 * jb) >> 5) * 64 + 16 * (m >> 2) + 4 * q + f] >> (((16 * jb) & 31) + 4 * (m & 3));
#pragma unroll
                        for (int e = 0; e < 4; ++e) m1[e] = ((nib >> e) & 1u) ? d1.inv_keep : 0.0f;
                    } else { m1[0] = m1[1] = m1[2] = m1[3] = 1.0f; }
                    (void)va;
                } else vx_masks_vox4(d1, (uint64_t)b * R + 16 * jb + m, V, va, m1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = TPW == 4 ? e : 0, i = TPW == 4 ? f : e;
                    const float av = aT[t][i];
                    float cdf, pdf;
                    vx_cdf_pdf(av, cdf, pdf);
                    const float da = hT[t][i] * m1[e] * (cdf + av * pdf);
                    aT[t][i] = av * cdf * m1[e];
                    hT[t][i] = da;
                    sb1[jb] += da;
                }
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int t = 0; t < TPW; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        aW2[jb][cb] = VX_MFMA(aT[t][i], dzT[cb][t][i], aW2[jb][cb]);
                        aW1[jb][cb] = VX_MFMA(hT[t][i], nT[cb][t][i], aW1[jb][cb]);
                    }
        }
    }
    // ---- flush: the 4 waves add their tiles into one LDS image in turn, then one float atomic per element and block
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int jb = 0; jb < RB; ++jb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = 16 * jb + 4 * q + i, c = 16 * cb + m;
                        float* d2p = red + j * C + c;
                        float* d1p = red + R * C + j * C + c;
                        *d2p = (wv == 0) ? aW2[jb][cb][i] : *d2p + aW2[jb][cb][i];
                        *d1p = (wv == 0) ? aW1[jb][cb][i] : *d1p + aW1[jb][cb][i];
                    }
        }
        __syncthreads();
    }
    // one float atomic per element and block, 256 contiguous bytes per wave instruction; the blocks start at different offsets so that the
    // blocks that finish together do not walk the same addresses in lock step (adds to one address serialise at ~25 ns each)
    {
        const int rot = (int)((blockIdx.x + gridDim.x * blockIdx.y) * 1031u % (unsigned)(R * C / 64)) * 64;
        for (int k = tid; k < R * C; k += 256) {
            int e = k + rot;
            e = e >= R * C ? e - R * C : e;
            const int c = e / R, j = e - c * R;
            atomicAdd(p.dw2 + e, red[j * C + c]);              // dW2 (C, R), destination order
            atomicAdd(p.dw1 + e, red[R * C + e]);              // dW1 (R, C)
        }
    }
    // row sums: bias gradients (reduce over q: lanes m, m+16, m+32, m+48) and norm sums (reduce over m: the 16 lanes of a q group), then over the
    // 4 waves through LDS -> ONE atomic (or partial-sum store) per element and block
    float* __restrict__ rsum = red + 2 * R * C;               // [wave][R + 3C]: db1 (R) | db2 (C) | S1,S2 interleaved (2C)
    constexpr int NS = R + 3 * C;
#pragma unroll
    for (int jb = 0; jb < RB; ++jb) {
        float s = sb1[jb];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (q == 0) rsum[wave * NS + 16 * jb + m] = s;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        float s = sb2[cb];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (q == 0) rsum[wave * NS + R + 16 * cb + m] = s;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t1 = S1[cb][i], t2 = S2[cb][i];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
            if (m == 0) { rsum[wave * NS + R + C + 2 * (16 * cb + 4 * q + i)] = t1; rsum[wave * NS + R + C + 2 * (16 * cb + 4 * q + i) + 1] = t2; }
        }
    __syncthreads();
    if (tid < NS) {
        const float s = (rsum[tid] + rsum[NS + tid]) + (rsum[2 * NS + tid] + rsum[3 * NS + tid]);
        if (tid < R) atomicAdd(p.db1 + tid, s);
        else if (tid < R + C) atomicAdd(p.db2 + (tid - R), s);
        else {
            const int e = tid - R - C, c = e >> 1;
            if (NORM == 1) atomicAdd((e & 1) ? p.dgamma + c : p.dbeta + c, s);
            else p.part_out[(((long)b * C + c) * gridDim.x + blockIdx.x) * 2 + (e & 1)] = s;
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------- host
static size_t vx_mlp_bwd_lds(int C, int R) { return ((size_t)5 * R * C + R + 2 * C + 4 * (R + 3 * C)) * sizeof(float); }
// batch-size independent launch geometry (a sample's partial sums are folded in the same order whatever it is batched with)
static int vx_mlp_tpw(long V) { return V >= 16384 ? 4 : 1; }
static int vx_mlp_iters(long V) { return V >= 32768 ? 2 : 1; }
static int vx_mlp_blocks(long V) { return vx_cdiv(V, (long)64 * vx_mlp_tpw(V) * vx_mlp_iters(V)); }

extern "C" int vx_mlp_supported(int C, int R, long V) {
    // C = 16 only.  The C = 32 instances (R = 96 / 64) were compiled until round 5 and never launched by default (C >= 32 runs on the tile-GEMM chains of pwa_fused.hip,
    // functional.TILE_MIN_C): their backward needed 512 VGPRs with 27 .. 176 spilled registers -- removed instead of kept as a slower A/B
    return ((C == 16 && R == 48) || (C == 16 && R == 32)) && V % 4 == 0 && V >= 64;
}

extern "C" int vx_mlp_bwd_nparts(int B, int C, long V) {
    (void)B; (void)C;
    return vx_mlp_blocks(V);
}

#define VX_MLP_SHAPES(X) X(16, 48) X(16, 32)

// h16 != 0 (norm = 0 only): x is a vx_bf16 array -- the o tensor of a JLC block in the bf16 storage mode; out stays fp32
extern "C" int vx_mlp_fwd_h(const void* x, int norm, const double* part, int nparts, float* stats, const float* gamma, const float* beta,
                            const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int C, int R, long V, float eps,
                            const void* seed_ptr, unsigned long long site1, float p1, unsigned long long site2, float p2, int h16, void* stream) {
    VX_REQUIRE(x && w1 && b1 && w2 && b2 && out && B > 0, "vx_mlp_fwd: bad args");
    VX_REQUIRE(!h16 || !norm, "vx_mlp_fwd: 16-bit input exists for the InstanceNorm (JLC) form only");
    VX_REQUIRE(vx_mlp_supported(C, R, V), "vx_mlp_fwd: unsupported shape C=%d R=%d V=%ld", C, R, V);
    VX_REQUIRE(norm ? (gamma && beta) : (stats != nullptr || part != nullptr), "vx_mlp_fwd: norm parameters missing");
    VX_REQUIRE(x != (const void*)out, "vx_mlp_fwd: in-place is not supported");
    VxMlp p = {};
    p.x = x; p.gamma = gamma; p.beta = beta; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
    p.part = part; p.nparts = nparts; p.stats = stats; p.out = out; p.V = V; p.eps = eps; p.iters = 1;      // forward: no cross-block sums, more waves hide the global round trips
    p.d1 = vx_mk_drop(seed_ptr, site1, p1);
    p.d2 = vx_mk_drop(seed_ptr, site2, p2);
    dim3 grid(vx_cdiv(V, (long)64 * vx_mlp_tpw(V)), B);
    hipStream_t st = (hipStream_t)stream;
    const int tpw = vx_mlp_tpw(V);
    bool done = false;
#define VX_MLP_FWD(CC, RR)                                                                                     \
    if (!done && C == CC && R == RR) {                                                                         \
        done = true;                                                                                           \
        if (norm) { if (tpw == 4) vx_mlp_fwd_k<CC, RR, 1, 4><<<grid, dim3(256), 0, st>>>(p); else vx_mlp_fwd_k<CC, RR, 1, 1><<<grid, dim3(256), 0, st>>>(p); } \
        else if (h16) { if (tpw == 4) vx_mlp_fwd_k<CC, RR, 0, 4, vx_bf16><<<grid, dim3(256), 0, st>>>(p); else vx_mlp_fwd_k<CC, RR, 0, 1, vx_bf16><<<grid, dim3(256), 0, st>>>(p); } \
        else { if (tpw == 4) vx_mlp_fwd_k<CC, RR, 0, 4><<<grid, dim3(256), 0, st>>>(p); else vx_mlp_fwd_k<CC, RR, 0, 1><<<grid, dim3(256), 0, st>>>(p); }      \
    }
    VX_MLP_SHAPES(VX_MLP_FWD)
#undef VX_MLP_FWD
    VX_LAUNCH_CHECK("vx_mlp_fwd");
    return 0;
}
extern "C" int vx_mlp_fwd(const float* x, int norm, const double* part, int nparts, float* stats, const float* gamma, const float* beta,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B, int C, int R, long V, float eps,
                          const void* seed_ptr, unsigned long long site1, float p1, unsigned long long site2, float p2, void* stream) {
    return vx_mlp_fwd_h(x, norm, part, nparts, stats, gamma, beta, w1, b1, w2, b2, out, B, C, R, V, eps, seed_ptr, site1, p1, site2, p2, 0, stream);
}

template <int C, int R, int NORM, int TPW, typename TI = float>
static void vx_mlp_bwd_launch(const VxMlp& p, dim3 grid, size_t shm, hipStream_t st) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)vx_mlp_bwd_k<C, R, NORM, TPW, TI>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    vx_mlp_bwd_k<C, R, NORM, TPW, TI><<<grid, dim3(256), shm, st>>>(p);
}

// h16 != 0 (norm = 0 only): x (= o of the JLC block) and the returned dx (= dn) are vx_bf16 arrays; dout stays fp32
extern "C" int vx_mlp_bwd_h(const void* x, int norm, const float* stats, const float* gamma, const float* beta, const float* w1, const float* b1,
                            const float* w2, const float* dout, void* dx, float* part_out, float* dgamma, float* dbeta, float* dw1, float* db1,
                            float* dw2, float* db2, int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site1, float p1,
                            unsigned long long site2, float p2, int h16, void* stream) {
    VX_REQUIRE(x && w1 && b1 && w2 && dout && dx && dw1 && db1 && dw2 && db2 && B > 0, "vx_mlp_bwd: bad args");
    VX_REQUIRE(!h16 || !norm, "vx_mlp_bwd: 16-bit tensors exist for the InstanceNorm (JLC) form only");
    VX_REQUIRE(vx_mlp_supported(C, R, V), "vx_mlp_bwd: unsupported shape C=%d R=%d V=%ld", C, R, V);
    VX_REQUIRE(norm ? (gamma && beta && dgamma && dbeta) : (stats && part_out), "vx_mlp_bwd: norm parameters missing");
    VxMlp p = {};
    p.x = x; p.gamma = gamma; p.beta = beta; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.dout = dout;
    p.stats = const_cast<float*>(stats); p.dn = dx; p.part_out = part_out; p.dgamma = dgamma; p.dbeta = dbeta;
    p.dw1 = dw1; p.db1 = db1; p.dw2 = dw2; p.db2 = db2; p.V = V; p.eps = eps;
    p.iters = vx_mlp_iters(V);
    p.d1 = vx_mk_drop(seed_ptr, site1, p1);
    p.d2 = vx_mk_drop(seed_ptr, site2, p2);
    dim3 grid(vx_mlp_blocks(V), B);
    const size_t shm = vx_mlp_bwd_lds(C, R);
    hipStream_t st = (hipStream_t)stream;
    const int tpw = vx_mlp_tpw(V);
    bool done = false;
#define VX_MLP_BWD(CC, RR)                                                                                     \
    if (!done && C == CC && R == RR) {                                                                         \
        done = true;                                                                                           \
        if (norm) { if (tpw == 4) vx_mlp_bwd_launch<CC, RR, 1, 4>(p, grid, shm, st); else vx_mlp_bwd_launch<CC, RR, 1, 1>(p, grid, shm, st); } \
        else if (h16) { if (tpw == 4) vx_mlp_bwd_launch<CC, RR, 0, 4, vx_bf16>(p, grid, shm, st); else vx_mlp_bwd_launch<CC, RR, 0, 1, vx_bf16>(p, grid, shm, st); } \
        else { if (tpw == 4) vx_mlp_bwd_launch<CC, RR, 0, 4>(p, grid, shm, st); else vx_mlp_bwd_launch<CC, RR, 0, 1>(p, grid, shm, st); }      \
    }
    VX_MLP_SHAPES(VX_MLP_BWD)
#undef VX_MLP_BWD
    VX_LAUNCH_CHECK("vx_mlp_bwd");
    return 0;
}
extern "C" int vx_mlp_bwd(const float* x, int norm, const float* stats, const float* gamma, const float* beta, const float* w1, const float* b1,
                          const float* w2, const float* dout, float* dx, float* part_out, float* dgamma, float* dbeta, float* dw1, float* db1,
                          float* dw2, float* db2, int B, int C, int R, long V, float eps, const void* seed_ptr, unsigned long long site1, float p1,
                          unsigned long long site2, float p2, void* stream) {
    return vx_mlp_bwd_h(x, norm, stats, gamma, beta, w1, b1, w2, dout, dx, part_out, dgamma, dbeta, dw1, db1, dw2, db2, B, C, R, V, eps, seed_ptr, site1, p1, site2, p2, 0, stream);
}
