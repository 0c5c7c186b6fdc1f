// Paired-Window Attention on the matrix cores (gfx950, v_mfma_f32_16x16x4_f32): reference PWA.py:308-327 (attention_operation) with the relative
// position bias of attention_utils.py:120-125.  Used by vx_pwa_attn_fwd / vx_pwa_attn_bwd (pwa.hip) whenever a window's tokens tile into 16-token
// blocks (l % 64 == 0: every 128^3 configuration); other geometries keep the fp32-VALU kernels of pwa.hip.
//
// MFMA operand geometry (A lane = (row l%16, k l/16), B lane = (k l/16, col l%16), D reg i = (row 4*(l/16)+i, col l%16)); m = l%16, q' = l/16:
//   forward / dQ pass  -- "S^T" orientation, a wave owns 16 QUERIES (columns) and walks the key tiles of its window:
//        S^T[key, q] = K[key, :] . Qs[q, :]        A = K (LDS, staged once per block in operand order), B = scaled Q (registers); K = c_qk exactly:
//                                                  c_qk / 4 MFMAs per 16 x 16 tile, no padding
//        soft-max / bias / Philox mask on the 4 accumulator registers of a lane = 4 CONSECUTIVE keys of one query (one Philox call, the
//        element -> word mapping of the VALU kernels, so the masks are identical); online maximum per query (2 cross-lane steps per tile)
//        O^T[c, q] += V^T[c, key] . P^T[key, q]    the accumulator registers of S^T ARE the B operand (reduction over the tile's 16 keys = its rows):
//                                                  4 MFMAs per tile and 16 output channels, A = V^T from LDS (rows >= c_v are zero padding)
//        dP^T = V . dO^T,  dQ^T[c, q] += K^T[c, key] . dS^T[key, q]  likewise (backward pass 1, which also sums d(bias) and delta = rowsum(dO * O))
//   dK / dV pass       -- "S" orientation, a wave owns 16 KEYS (columns) and walks the query tiles:
//        S[q, key] = Qs . K^T, dP = dO . V^T,  dV^T[c, key] += dO^T[c, q] . (P*M)[q, key],  dK^T[c, key] += Qs^T[c, q] . dS[q, key]
// A block = 8 waves = 128 queries (or keys) of ONE window: K and V (resp. Q, dO, LSE, delta) of the whole window are staged in LDS once per block
// and shared by the 8 waves, instead of being re-fetched by every 64-row unit as in the VALU kernels.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

typedef float vx_f32x4 __attribute__((ext_vector_type(4)));
#define VX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define VX_DTABLE_REPLICAS 16          // must match pwa.hip

struct VxAttnM {
    int BH, heads, Nt, ML, l, M;
    int n[3];
    float scale;
    int lin_cst, Tsz;
};

// Staging loops: 8 independent loads per thread are issued before the first LDS store (a rolled load -> store loop pays one L2 round trip per
// element: that was half of the forward kernel's time).
// lin[t] of every token of a window + the bias table of head a (bias[k] = table[k * heads + a])
__device__ __forceinline__ void vx_am_tables(const VxAttnM& A, const float* __restrict__ table, int a, int* lin, float* bias, int nthr) {
    for (int t = threadIdx.x; t < A.l; t += nthr) {
        const int t2 = t % A.n[2], t1 = (t / A.n[2]) % A.n[1], t0 = t / (A.n[2] * A.n[1]);
        lin[t] = (t0 * (2 * A.n[1] - 1) + t1) * (2 * A.n[2] - 1) + t2;
    }
    for (int k0 = threadIdx.x; k0 < A.Tsz; k0 += nthr * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * nthr; v[u] = table[(long)(k < A.Tsz ? k : 0) * A.heads + a]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * nthr; if (k < A.Tsz) bias[k] = v[u]; }
    }
}
// rows of a (ML, C) matrix in MFMA operand order: dst[((tile * C/4 + ks) * 64) + lane] = src[(16 tile + lane%16) * C + 4 ks + lane/16] * mul.
// Thread -> (row, float4 column group): one 16-byte global load, four scattered LDS stores (ML * C / 4 float4s, 8 per thread in flight).
template <int C>
__device__ __forceinline__ void vx_am_stage_op(float* __restrict__ dst, const float* __restrict__ src, int ML, float mul, int nthr) {
    const int n4 = ML * (C / 4);
    for (int e0 = threadIdx.x; e0 < n4; e0 += nthr * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * nthr; v[u] = reinterpret_cast<const float4*>(src)[e < n4 ? e : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * nthr;
            if (e < n4) {
                const int row = e / (C / 4), ks = e - row * (C / 4);
                float* d = dst + ((row >> 4) * (C / 4) + ks) * 64 + (row & 15);      // lane = 16 * (c % 4) + row % 16
                d[0] = v[u].x * mul; d[16] = v[u].y * mul; d[32] = v[u].z * mul; d[48] = v[u].w * mul;
            }
        }
    }
}
// element (row 16 tile + r, column c) of a matrix staged by vx_am_stage_op
template <int C> __device__ __forceinline__ int vx_am_op_index(int tile, int r, int c) { return ((tile * (C / 4) + (c >> 2)) * 64) + (c & 3) * 16 + r; }

// ---------------------------------------------------------------------------------------------------------------------------- forward
template <int CQ, int CV>
__global__ void __launch_bounds__(512) vx_pwa_attn_mfma_fwd_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                              const float* __restrict__ table, float* __restrict__ O, float* __restrict__ LSE,
                                                              VxAttnM A, VxDrop drop) {
    constexpr int KSQ = CQ / 4, CVB = (CV + 15) / 16, SV = CV + 2;
    extern __shared__ __attribute__((aligned(16))) float vx_am_lds[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_am_lds);
    float* __restrict__ bias = vx_am_lds + ((A.l + 3) & ~3);
    float* __restrict__ Kop = bias + ((A.Tsz + 3) & ~3);
    float* __restrict__ Vs = Kop + A.ML * CQ;
    const long win = blockIdx.y;
    const int a = (int)((win / A.Nt) % A.heads);
    const float* __restrict__ kp = K + win * A.ML * CQ;
    const float* __restrict__ vp = Vt + win * A.ML * CV;
    vx_am_tables(A, table, a, lin, bias, 512);
    vx_am_stage_op<CQ>(Kop, kp, A.ML, 1.0f, 512);
    for (int e0 = threadIdx.x; e0 < A.ML * (CV / 4); e0 += 512 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int e = e0 + u * 512; v[u] = reinterpret_cast<const float4*>(vp)[e < A.ML * (CV / 4) ? e : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 512;
            if (e < A.ML * (CV / 4)) { const int key = e / (CV / 4), c4 = e - key * (CV / 4); float* d = Vs + key * SV + 4 * c4; d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, qg = lane >> 4;
    const int q0 = (blockIdx.x * 8 + wave) * 16;
    if (q0 >= A.ML) return;
    const VxDropCtx dc = vx_drop_ctx(drop);
    const long row = win * A.ML + q0 + m;
    float qb[KSQ];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) qb[ks] = Q[row * CQ + 4 * ks + qg] * A.scale;
    const int lin_q = lin[(q0 + m) % A.l] + A.lin_cst;
    float mrun = -INFINITY, lsum = 0.0f;
    vx_f32x4 oacc[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) oacc[cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    // KU key tiles per iteration: their S tiles, bias gathers, exponentials and Philox draws are independent, so the LDS / MFMA / transcendental
    // latencies of one tile hide behind the others', and the per-query maximum needs ONE cross-lane reduction per KU tiles
    constexpr int KU = 4;
    const int ntile = A.ML / 16;                     // l % 16 == 0 and ML >= 64: a multiple of KU
    vx_f32x4 oacc2[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) oacc2[cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt0 = 0; kt0 < ntile; kt0 += KU) {
        vx_f32x4 s[KU];
        int4 lk[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            s[u] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
            lk[u] = *reinterpret_cast<const int4*>(&lin[(16 * (kt0 + u)) % A.l + 4 * qg]);
        }
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
            for (int u = 0; u < KU; ++u) s[u] = VX_MFMA(Kop[((kt0 + u) * KSQ + ks) * 64 + lane], qb[ks], s[u]);
        float mt = -INFINITY;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            s[u][0] += bias[lin_q - lk[u].x]; s[u][1] += bias[lin_q - lk[u].y]; s[u][2] += bias[lin_q - lk[u].z]; s[u][3] += bias[lin_q - lk[u].w];
            mt = fmaxf(mt, fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3])));
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(mrun, mt);
        const float alpha = __expf(mrun - mn);
        mrun = mn;
        float psum = 0.0f;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float m4[4];
            vx_masks_vox4(dc, (uint64_t)row, A.ML, 16 * (kt0 + u) + 4 * qg, m4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float p = __expf(s[u][i] - mn); psum += p; s[u][i] = p * m4[i]; }
        }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int cb = 0; cb < CVB; ++cb) { oacc[cb] *= alpha; oacc2[cb] *= alpha; }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb) {
                    const int ch = 16 * cb + m;
                    const float va = ch < CV ? Vs[(16 * (kt0 + u) + 4 * qg + i) * SV + ch] : 0.0f;
                    if (u & 1) oacc2[cb] = VX_MFMA(va, s[u][i], oacc2[cb]);      // two accumulators: half the dependent-MFMA chain
                    else oacc[cb] = VX_MFMA(va, s[u][i], oacc[cb]);
                }
    }
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) oacc[cb] += oacc2[cb];
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) {
        const int ch = 16 * cb + 4 * qg;                 // this lane's 4 channels of query m
        if (ch < CV) *reinterpret_cast<float4*>(O + row * CV + ch) = make_float4(oacc[cb][0] * inv, oacc[cb][1] * inv, oacc[cb][2] * inv, oacc[cb][3] * inv);
    }
    if (qg == 0) LSE[row] = mrun + __logf(lsum);
}

// ---------------------------------------------------------------------------------------------------------------------------- backward 1: dQ, delta, d(bias)
template <int CQ, int CV>
__global__ void __launch_bounds__(512) vx_pwa_attn_mfma_bwd_q_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                                const float* __restrict__ table, const float* __restrict__ O, const float* __restrict__ LSE,
                                                                const float* __restrict__ dO, float* __restrict__ dQ, float* __restrict__ Delta,
                                                                float* __restrict__ dtable_rep, VxAttnM A, VxDrop drop) {
    constexpr int KSQ = CQ / 4, KSV = CV / 4;
    extern __shared__ __attribute__((aligned(16))) float vx_am_lds[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_am_lds);
    float* __restrict__ bias = vx_am_lds + ((A.l + 3) & ~3);
    const int tpad = (A.Tsz + 3) & ~3;
    float* __restrict__ gtab = bias + tpad;                   // 4 replicas of the bias-gradient table, one per lane group
    float* __restrict__ Kop = gtab + 4 * tpad;
    float* __restrict__ Vop = Kop + A.ML * CQ;
    const long win = blockIdx.y;
    const int a = (int)((win / A.Nt) % A.heads);
    vx_am_tables(A, table, a, lin, bias, 512);
    for (int k = threadIdx.x; k < 4 * tpad; k += 512) gtab[k] = 0.0f;
    vx_am_stage_op<CQ>(Kop, K + win * A.ML * CQ, A.ML, 1.0f, 512);
    vx_am_stage_op<CV>(Vop, Vt + win * A.ML * CV, A.ML, 1.0f, 512);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, qg = lane >> 4;
    const int q0 = (blockIdx.x * 8 + wave) * 16;
    if (q0 < A.ML) {
        const VxDropCtx dc = vx_drop_ctx(drop);
        const long row = win * A.ML + q0 + m;
        float qb[KSQ], dob[KSV];
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) qb[ks] = Q[row * CQ + 4 * ks + qg] * A.scale;
        float delta = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KSV; ++ks) { dob[ks] = dO[row * CV + 4 * ks + qg]; delta = fmaf(dob[ks], O[row * CV + 4 * ks + qg], delta); }
        delta += __shfl_xor(delta, 16, 64);
        delta += __shfl_xor(delta, 32, 64);
        const float lse = LSE[row];
        const int lin_q = lin[(q0 + m) % A.l] + A.lin_cst;
        vx_f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dq2 = {0.f, 0.f, 0.f, 0.f};
        constexpr int KU = 4;
        const int ntile = A.ML / 16;
        float* __restrict__ gmine = gtab + qg * tpad;          // this lane group's replica: the 16 lanes of an instruction hold 16 DIFFERENT queries of one key => distinct bins
        for (int kt0 = 0; kt0 < ntile; kt0 += KU) {
            vx_f32x4 s[KU], dp[KU];
            int4 lk[KU];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                s[u] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
                dp[u] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
                lk[u] = *reinterpret_cast<const int4*>(&lin[(16 * (kt0 + u)) % A.l + 4 * qg]);
            }
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
                for (int u = 0; u < KU; ++u) s[u] = VX_MFMA(Kop[((kt0 + u) * KSQ + ks) * 64 + lane], qb[ks], s[u]);
#pragma unroll
            for (int ks = 0; ks < KSV; ++ks)
#pragma unroll
                for (int u = 0; u < KU; ++u) dp[u] = VX_MFMA(Vop[((kt0 + u) * KSV + ks) * 64 + lane], dob[ks], dp[u]);
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int bi[4] = {lin_q - lk[u].x, lin_q - lk[u].y, lin_q - lk[u].z, lin_q - lk[u].w};
                float m4[4];
                vx_masks_vox4(dc, (uint64_t)row, A.ML, 16 * (kt0 + u) + 4 * qg, m4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf(s[u][i] + bias[bi[i]] - lse);
                    const float ds = p * (dp[u][i] * m4[i] - delta);
                    atomicAdd(&gmine[bi[i]], ds);
                    s[u][i] = ds;
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ka = m < CQ ? Kop[vx_am_op_index<CQ>(kt0 + u, 4 * qg + i, m < CQ ? m : 0)] : 0.0f;     // K[16kt + 4q' + i][c = m]
                    if (u & 1) dq2 = VX_MFMA(ka, s[u][i], dq2);
                    else dq = VX_MFMA(ka, s[u][i], dq);
                }
        }
        dq += dq2;
        if (4 * qg < CQ) *reinterpret_cast<float4*>(dQ + row * CQ + 4 * qg) = make_float4(dq[0] * A.scale, dq[1] * A.scale, dq[2] * A.scale, dq[3] * A.scale);
        if (qg == 0) Delta[row] = delta;
    }
    __syncthreads();
    // one of VX_DTABLE_REPLICAS copies per block: thousands of blocks adding into the same few hundred addresses serialise (folded by the dK/dV kernel)
    float* __restrict__ dst = dtable_rep + (long)((blockIdx.x + gridDim.x * blockIdx.y) % VX_DTABLE_REPLICAS) * A.Tsz * A.heads;
    for (int k = threadIdx.x; k < A.Tsz; k += 512) {
        const float g = (gtab[k] + gtab[tpad + k]) + (gtab[2 * tpad + k] + gtab[3 * tpad + k]);
        if (g != 0.0f) atomicAdd(dst + (long)k * A.heads + a, g);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------- backward 2: dK, dV
template <int CQ, int CV>
__global__ void __launch_bounds__(512) vx_pwa_attn_mfma_bwd_kv_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                                 const float* __restrict__ table, const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                                 const float* __restrict__ dO, float* __restrict__ dK, float* __restrict__ dV,
                                                                 const float* __restrict__ dtable_rep, float* __restrict__ dtable, VxAttnM A, VxDrop drop) {
    constexpr int KSQ = CQ / 4, KSV = CV / 4, CVB = (CV + 15) / 16;
    {   // fold the dQ kernel's replicated bias-gradient tables into dtable (it ran before this kernel on the same stream): one owner thread per entry
        const long nb = (long)gridDim.x * gridDim.y, bid = blockIdx.x + (long)gridDim.x * blockIdx.y;
        for (long k = bid * 512 + threadIdx.x; k < (long)A.Tsz * A.heads; k += nb * 512) {
            float g = 0.0f;
#pragma unroll
            for (int r = 0; r < VX_DTABLE_REPLICAS; ++r) g += dtable_rep[(long)r * A.Tsz * A.heads + k];
            dtable[k] += g;
        }
    }
    extern __shared__ __attribute__((aligned(16))) float vx_am_lds[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_am_lds);
    float* __restrict__ bias = vx_am_lds + ((A.l + 3) & ~3);
    float* __restrict__ Qop = bias + ((A.Tsz + 3) & ~3);
    float* __restrict__ dOop = Qop + A.ML * CQ;
    float* __restrict__ lse_s = dOop + A.ML * CV;
    float* __restrict__ del_s = lse_s + A.ML;
    const long win = blockIdx.y;
    const int a = (int)((win / A.Nt) % A.heads);
    vx_am_tables(A, table, a, lin, bias, 512);
    vx_am_stage_op<CQ>(Qop, Q + win * A.ML * CQ, A.ML, A.scale, 512);
    vx_am_stage_op<CV>(dOop, dO + win * A.ML * CV, A.ML, 1.0f, 512);
    for (int e = threadIdx.x; e < A.ML; e += 512) { lse_s[e] = LSE[win * A.ML + e]; del_s[e] = Delta[win * A.ML + e]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, qg = lane >> 4;
    const int k0 = (blockIdx.x * 8 + wave) * 16;
    if (k0 >= A.ML) return;
    const VxDropCtx dc = vx_drop_ctx(drop);
    const long krow = win * A.ML + k0 + m;
    float kb[KSQ], vb[KSV];
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) kb[ks] = K[krow * CQ + 4 * ks + qg];
#pragma unroll
    for (int ks = 0; ks < KSV; ++ks) vb[ks] = Vt[krow * CV + 4 * ks + qg];
    const int lin_k = lin[(k0 + m) % A.l] - A.lin_cst;
    vx_f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) dv[cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int KU = 4;
    const int ntile = A.ML / 16;
    vx_f32x4 dk2 = {0.f, 0.f, 0.f, 0.f}, dv2[CVB];
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) dv2[cb] = (vx_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int qt0 = 0; qt0 < ntile; qt0 += KU) {
        vx_f32x4 s[KU], dp[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { s[u] = (vx_f32x4){0.f, 0.f, 0.f, 0.f}; dp[u] = (vx_f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
            for (int u = 0; u < KU; ++u) s[u] = VX_MFMA(Qop[((qt0 + u) * KSQ + ks) * 64 + lane], kb[ks], s[u]);
#pragma unroll
        for (int ks = 0; ks < KSV; ++ks)
#pragma unroll
            for (int u = 0; u < KU; ++u) dp[u] = VX_MFMA(dOop[((qt0 + u) * KSV + ks) * 64 + lane], vb[ks], dp[u]);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int qt = qt0 + u;
            const int4 lq = *reinterpret_cast<const int4*>(&lin[(16 * qt) % A.l + 4 * qg]);
            const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[16 * qt + 4 * qg]);
            const float4 d4 = *reinterpret_cast<const float4*>(&del_s[16 * qt + 4 * qg]);
            const int bi[4] = {lq.x - lin_k, lq.y - lin_k, lq.z - lin_k, lq.w - lin_k};
            const float lse[4] = {l4.x, l4.y, l4.z, l4.w}, del[4] = {d4.x, d4.y, d4.z, d4.w};
            float m4[4];
            vx_masks_rows4(dc, (uint64_t)(win * A.ML + 16 * qt + 4 * qg), 1, A.ML, k0 + m, m4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __expf(s[u][i] + bias[bi[i]] - lse[i]);
                s[u][i] = p * (dp[u][i] * m4[i] - del[i]);      // dS
                dp[u][i] = p * m4[i];                            // P * M
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int cb = 0; cb < CVB; ++cb) {
                    const int ch = 16 * cb + m;
                    const float da = ch < CV ? dOop[vx_am_op_index<CV>(qt0 + u, 4 * qg + i, ch < CV ? ch : 0)] : 0.0f;          // dO[16qt + 4q' + i][ch]
                    if (u & 1) dv2[cb] = VX_MFMA(da, dp[u][i], dv2[cb]);
                    else dv[cb] = VX_MFMA(da, dp[u][i], dv[cb]);
                }
                const float qa = m < CQ ? Qop[vx_am_op_index<CQ>(qt0 + u, 4 * qg + i, m < CQ ? m : 0)] : 0.0f;                 // scale * Q[16qt + 4q' + i][c = m]
                if (u & 1) dk2 = VX_MFMA(qa, s[u][i], dk2);
                else dk = VX_MFMA(qa, s[u][i], dk);
            }
    }
    dk += dk2;
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) dv[cb] += dv2[cb];
    if (4 * qg < CQ) *reinterpret_cast<float4*>(dK + krow * CQ + 4 * qg) = make_float4(dk[0], dk[1], dk[2], dk[3]);
#pragma unroll
    for (int cb = 0; cb < CVB; ++cb) {
        const int ch = 16 * cb + 4 * qg;
        if (ch < CV) *reinterpret_cast<float4*>(dV + krow * CV + ch) = make_float4(dv[cb][0], dv[cb][1], dv[cb][2], dv[cb][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------- host
static bool vx_am_fill(VxAttnM& A, const VxPwaPlan* P, int B, int M, int cq) {
    A.BH = B * P->heads; A.heads = P->heads; A.Nt = P->Ntot; A.l = P->l; A.M = M; A.ML = M * P->l;
    A.n[0] = P->n[0]; A.n[1] = P->n[1]; A.n[2] = P->n[2];
    A.scale = 1.0f / sqrtf((float)cq);
    A.lin_cst = ((P->n[0] - 1) * (2 * P->n[1] - 1) + (P->n[1] - 1)) * (2 * P->n[2] - 1) + (P->n[2] - 1);
    A.Tsz = (2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
    return true;
}
static size_t vx_am_tab_floats(const VxAttnM& A) { return (size_t)((A.l + 3) & ~3) + (size_t)((A.Tsz + 3) & ~3); }
static size_t vx_am_lds_fwd(const VxAttnM& A, int cq, int cv) { return (vx_am_tab_floats(A) + (size_t)A.ML * cq + (size_t)A.ML * (cv + 2)) * 4; }
static size_t vx_am_lds_q(const VxAttnM& A, int cq, int cv) { return (vx_am_tab_floats(A) + (size_t)4 * ((A.Tsz + 3) & ~3) + (size_t)A.ML * (cq + cv)) * 4; }
static size_t vx_am_lds_kv(const VxAttnM& A, int cq, int cv) { return (vx_am_tab_floats(A) + (size_t)A.ML * (cq + cv + 2)) * 4; }

// Which passes take the MFMA kernels: bit 0 forward, bit 1 backward.  Default = forward only.  Measured on MI355X at the bench shape (8^3-window level,
// 1024 keys, c_qk = c_v = 8, p_drop 0.1; profiles/r02*): forward 153 us MFMA vs 157 us VALU; dK/dV pass 232 vs 181 us; dQ pass 640 vs 273 us.  At head
// widths 4..16 the two GEMMs are 6 of ~180 instructions per 16 x 16 tile: the tile's time is the per-pair soft-max side (Philox draw = half of the VALU
// issue, exp, bias gather), and d(bias) on LDS float atomics costs ~200 clocks per wave instruction in this orientation (the VALU dQ kernel's lanes
// are 64 different queries of one key, which is what lets it use plain read-add-write windows).  So the backward stays on the VALU kernels.
static int vx_am_enabled = 1;
extern "C" int vx_pwa_attn_set_mfma(int mask) { vx_am_enabled = mask & 3; return 0; }

// bit mask of the passes this geometry can run on the MFMA kernels (0 = none), after the vx_pwa_attn_set_mfma selection
extern "C" int vx_pwa_attn_mfma_ok(const VxPwaPlan* P, int B, int M, int cq, int cv) {
    if (!vx_am_enabled || P == nullptr || B <= 0 || M <= 0) return 0;
    if (P->l % 64 != 0 || cq % 4 != 0 || cq > 16 || cv % 4 != 0 || cv > 32) return 0;
    if (!((cq == 4 && cv == 4) || (cq == 8 && cv == 8) || (cq == 8 && cv == 16) || (cq == 16 && cv == 32) || (cq == 16 && cv == 16) || (cq == 4 && cv == 8))) return 0;
    VxAttnM A;
    vx_am_fill(A, P, B, M, cq);
    const size_t cap = 150 * 1024;
    return (vx_am_lds_fwd(A, cq, cv) <= cap && vx_am_lds_q(A, cq, cv) <= cap && vx_am_lds_kv(A, cq, cv) <= cap) ? vx_am_enabled : 0;
}

template <int A_, int B_> struct vx_am_pair { static constexpr int a = A_, b = B_; };
template <class F> static bool vx_am_dispatch(int cq, int cv, F&& f) {
#define VX_CASE(X, Y) if (cq == X && cv == Y) { f(vx_am_pair<X, Y>{}); return true; }
    VX_CASE(4, 4) VX_CASE(8, 8) VX_CASE(8, 16) VX_CASE(16, 32) VX_CASE(16, 16) VX_CASE(4, 8)
#undef VX_CASE
    return false;
}
template <class K> static void vx_am_attr(K kernel) { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }

// same contracts as vx_pwa_attn_fwd / vx_pwa_attn_bwd (include/veloxseg_hip.h); the caller checked vx_pwa_attn_mfma_ok and zeroed the replicas
int vx_pwa_attn_mfma_fwd(const float* Q, const float* K, const float* V, const float* table, float* O, float* LSE, const VxPwaPlan* plan, int B, int M,
                         int cq, int cv, VxDrop d, void* stream) {
    VxAttnM A;
    vx_am_fill(A, plan, B, M, cq);
    const dim3 grid(vx_cdiv(A.ML / 16, 8), (unsigned)((long)A.BH * A.Nt));
    const size_t shm = vx_am_lds_fwd(A, cq, cv);
    vx_am_dispatch(cq, cv, [&](auto pr) {
        constexpr int CQ = decltype(pr)::a, CV = decltype(pr)::b;
        static bool once = false;
        if (!once) { vx_am_attr(vx_pwa_attn_mfma_fwd_k<CQ, CV>); once = true; }
        vx_pwa_attn_mfma_fwd_k<CQ, CV><<<grid, dim3(512), shm, (hipStream_t)stream>>>(Q, K, V, table, O, LSE, A, d);
    });
    return 0;
}
int vx_pwa_attn_mfma_bwd(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE, const float* dO, float* dQ,
                         float* dK, float* dV, float* dtable, float* delta, float* rep, const VxPwaPlan* plan, int B, int M, int cq, int cv, VxDrop d,
                         void* stream) {
    VxAttnM A;
    vx_am_fill(A, plan, B, M, cq);
    const dim3 grid(vx_cdiv(A.ML / 16, 8), (unsigned)((long)A.BH * A.Nt));
    const size_t shm_q = vx_am_lds_q(A, cq, cv), shm_kv = vx_am_lds_kv(A, cq, cv);
    vx_am_dispatch(cq, cv, [&](auto pr) {
        constexpr int CQ = decltype(pr)::a, CV = decltype(pr)::b;
        static bool once = false;
        if (!once) { vx_am_attr(vx_pwa_attn_mfma_bwd_q_k<CQ, CV>); vx_am_attr(vx_pwa_attn_mfma_bwd_kv_k<CQ, CV>); once = true; }
        vx_pwa_attn_mfma_bwd_q_k<CQ, CV><<<grid, dim3(512), shm_q, (hipStream_t)stream>>>(Q, K, V, table, O, LSE, dO, dQ, delta, rep, A, d);
        vx_pwa_attn_mfma_bwd_kv_k<CQ, CV><<<grid, dim3(512), shm_kv, (hipStream_t)stream>>>(Q, K, V, table, LSE, delta, dO, dK, dV, rep, dtable, A, d);
    });
    return 0;
}
