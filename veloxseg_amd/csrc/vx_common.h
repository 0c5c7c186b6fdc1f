// Shared device/host helpers for the VeloxSeg gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define VX_WAVE 64

extern thread_local char vx_err_buf[512];

#define VX_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(vx_err_buf, sizeof(vx_err_buf), __VA_ARGS__); \
        return (code);                                       \
    } while (0)

#define VX_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) VX_FAIL(-1, __VA_ARGS__); \
    } while (0)

// Launch check: hipGetLastError only (no sync; graph-capturable).
#define VX_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) VX_FAIL(-2, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int vx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions -------------------------------------------------------------
__device__ __forceinline__ float vx_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double vx_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float vx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a 256-thread block (4 waves); result valid in thread 0.  `sm` needs >= 4 floats.
__device__ __forceinline__ float vx_block_sum_256(float v, float* sm) {
    v = vx_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wid] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// ---- 16-bit storage of activations (round 6: the bf16 STORAGE mode of BASELINE configs[1]; reference speed_test.py:122,127 = torch.amp.autocast) ---------------
// A tensor kept in HBM as bf16 is an array of `vx_bf16` (storage only: every kernel converts to fp32 on load and rounds to nearest-even on store; sums, statistics,
// soft-max and the loss stay fp32).  Kernels are templated on the element type of the tensors that may be 16-bit and reach memory through the overloads below, so the
// fp32 instances are the unchanged code.  4 consecutive elements = one 16-byte (fp32) or 8-byte (bf16) access.
struct vx_bf16 { unsigned short v; };
__device__ __forceinline__ uint32_t vx_pack_bf16x2(float a, float b) {      // (a -> low half, b -> high half), round to nearest even: v_cvt_pk_bf16_f32 on gfx950
    typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
    const bf2_ v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float vx_bf16_lo(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float vx_bf16_hi(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ float4 vx_ld4(const float* p, long i) { return *reinterpret_cast<const float4*>(p + i); }
__device__ __forceinline__ float4 vx_ld4(const vx_bf16* p, long i) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + i);
    return make_float4(vx_bf16_lo(v.x), vx_bf16_hi(v.x), vx_bf16_lo(v.y), vx_bf16_hi(v.y));
}
__device__ __forceinline__ void vx_st4(float* p, long i, float4 v) { *reinterpret_cast<float4*>(p + i) = v; }
__device__ __forceinline__ void vx_st4(vx_bf16* p, long i, float4 v) { *reinterpret_cast<uint2*>(p + i) = make_uint2(vx_pack_bf16x2(v.x, v.y), vx_pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ float vx_ld1(const float* p, long i) { return p[i]; }
__device__ __forceinline__ float vx_ld1(const vx_bf16* p, long i) { return __builtin_bit_cast(float, (uint32_t)p[i].v << 16); }
__device__ __forceinline__ void vx_st1(float* p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void vx_st1(vx_bf16* p, long i, float v) { p[i].v = (unsigned short)(vx_pack_bf16x2(v, 0.0f) & 0xffffu); }
// what a value becomes when it is stored as T and loaded again (statistics of a tensor that is kept in bf16 are taken over the ROUNDED values its readers will see)
template <typename T> __device__ __forceinline__ float vx_round_as(float v) { return v; }
template <> __device__ __forceinline__ float vx_round_as<vx_bf16>(float v) { return vx_bf16_lo(vx_pack_bf16x2(v, 0.0f)); }

// ---- XCD-aware block coordinates (round 6) --------------------------------------------------------------------------------------------------------
// The blocks of a 2-D grid are dispatched in linear order (x fastest) and consecutive linear ids go to different XCDs (8 L2 slices).  When the blocks of one ROW (same
// blockIdx.y: one attention window, one sample ...) share their operands, the row is spread over all eight L2s and every one of them fetches the operands from HBM.
// vx_xcd_rows gives the block other coordinates: eight rows are interleaved, so that all blocks of a row carry the same (linear id & 7) = sit on one XCD.  Rows beyond
// the last multiple of eight keep their own coordinates.
__device__ __forceinline__ void vx_xcd_rows(int& bx, int& by, int on = 1) {
    const int nbx = (int)gridDim.x;
    const long L = (long)blockIdx.x + (long)nbx * blockIdx.y;
    const long full = (long)nbx * ((gridDim.y >> 3) << 3);
    if (on && L < full) {
        const long g = L / (8L * nbx);
        const int r = (int)(L - g * 8L * nbx);
        by = (int)(g * 8 + (r & 7));
        bx = r >> 3;
    } else { bx = (int)blockIdx.x; by = (int)blockIdx.y; }
}

// ---- exact-erf GELU (nn.GELU() default) ---------------------------------------------------
__device__ __forceinline__ float vx_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float vx_gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi(x) = exp(-x^2/2) / sqrt(2 pi) from ONE v_exp: erf through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, i.e. <= 7.5e-8 on Phi: at the rounding level of fp32 erff), whose exponential exp(-(x/sqrt2)^2) is the one phi needs.
// Used by the fused block kernels, where GELU / GELU' are evaluated for every element of the expanded activation on the fly.
__device__ __forceinline__ void vx_cdf_pdf(float x, float& cdf, float& pdf) {
    const float e = __expf(-0.5f * x * x);
    const float y = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, y, 1.0f));          // v_rcp_f32 (1 ulp; __frcp_rn is a full IEEE division: 11 instructions per GELU)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;          // 0.5 * erfc(|x| / sqrt 2)
    cdf = x >= 0.0f ? 1.0f - half_erfc : half_erfc;
    pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float vx_gelu_fast(float x) { float c, p; vx_cdf_pdf(x, c, p); return x * c; }
__device__ __forceinline__ float vx_gelu_grad_fast(float x) { float c, p; vx_cdf_pdf(x, c, p); return fmaf(x, p, c); }

// ---- counter-based RNG for dropout: Philox4x32-10 ------------------------------------------
struct VxPhilox {
    uint32_t c[4];
    uint32_t k[2];
};
__device__ __forceinline__ void vx_philox_round(uint32_t (&c)[4], const uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // one 32 x 32 -> 64 multiply per word pair (v_mad_u64_u32) instead of v_mul_hi_u32 + v_mul_lo_u32: all three are quarter-rate, so this halves the
    // multiplier time of a round; same bits
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0], p1 = (uint64_t)M1 * (uint64_t)c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// 4 x 32 random bits for (seed, stream id, counter).  7 rounds: the smallest round count of Philox4x32 that passes BigCrush (Salmon et al., SC'11,
// table 2; 10 is the library default with a safety margin).  Dropout masks need far less, and the 32-bit multiplies of a round are quarter-rate
// VALU instructions: the generator was a visible share of every kernel that draws masks.
#define VX_PHILOX_ROUNDS 7
__device__ __forceinline__ void vx_philox4(uint64_t seed, uint64_t stream, uint64_t ctr, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < VX_PHILOX_ROUNDS; ++r) {
        vx_philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// keep-mask scale for element `idx` of dropout stream `stream`: returns 0 or 1/(1-p)
__device__ __forceinline__ float vx_dropout_scale(uint64_t seed, uint64_t stream, uint64_t idx, float p, float inv_keep) {
    uint32_t r[4];
    vx_philox4(seed, stream, idx >> 2, r);
    const uint32_t bits = r[idx & 3];
    const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);   // [0,1)
    return u >= p ? inv_keep : 0.0f;
}

// Dropout descriptor passed by value to kernels.  `seed_ptr` points at a device uint64 pair
// {seed, step}; the step is bumped by the host/optimizer so a captured graph replays new masks.
struct VxDrop {
    const uint64_t* seed_ptr;   // nullptr => dropout off
    uint64_t stream;            // unique per dropout site
    float p;
};
__device__ __forceinline__ float vx_drop(const VxDrop& d, uint64_t idx) {
    if (d.seed_ptr == nullptr || d.p <= 0.0f) return 1.0f;
    const uint64_t seed = d.seed_ptr[0] + 0x9E3779B97F4A7C15ull * d.seed_ptr[1];
    return vx_dropout_scale(seed, d.stream, idx, d.p, 1.0f / (1.0f - d.p));
}

// Dropout context hoisted out of inner loops: the {seed, step} pair is read once (scalar loads), masks are derived from raw Philox words.
// Element idx always maps to word (idx & 3) of philox(seed, stream, idx >> 2), so any kernel may amortise one Philox call over the four
// elements that share a counter without changing the mask.
static inline VxDrop vx_mk_drop(const void* seed_ptr, unsigned long long stream, float p) {
    VxDrop d;
    d.seed_ptr = (p > 0.0f) ? (const uint64_t*)seed_ptr : nullptr;
    d.stream = stream;
    d.p = p;
    return d;
}
struct VxDropCtx {
    uint64_t seed, stream;
    float p, inv_keep;
    bool on;
    uint32_t thr16;      // attention sites (vx_attn_ctx): keep iff the element's 16 random bits >= thr16
};
__device__ __forceinline__ VxDropCtx vx_drop_ctx(const VxDrop& d) {
    VxDropCtx c;
    c.on = d.seed_ptr != nullptr && d.p > 0.0f;
    c.stream = d.stream;
    c.p = d.p;
    c.inv_keep = c.on ? 1.0f / (1.0f - d.p) : 1.0f;
    c.seed = c.on ? d.seed_ptr[0] + 0x9E3779B97F4A7C15ull * d.seed_ptr[1] : 0ull;
    c.thr16 = 0;
    return c;
}
__device__ __forceinline__ float vx_mask_of_bits(const VxDropCtx& c, uint32_t bits) {
    const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);
    return u >= c.p ? c.inv_keep : 0.0f;
}
__device__ __forceinline__ float vx_drop1(const VxDropCtx& c, uint64_t idx) {
    if (!c.on) return 1.0f;
    uint32_t r[4];
    vx_philox4(c.seed, c.stream, idx >> 2, r);
    return vx_mask_of_bits(c, r[idx & 3]);
}
// masks of elements idx0 .. idx0+3; one Philox call when idx0 is a multiple of 4 (`aligned`, wave-uniform), four otherwise
__device__ __forceinline__ void vx_drop4(const VxDropCtx& c, uint64_t idx0, bool aligned, float (&m)[4]) {
    if (!c.on) {
        m[0] = m[1] = m[2] = m[3] = 1.0f;
    } else if (aligned) {
        uint32_t r[4];
        vx_philox4(c.seed, c.stream, idx0 >> 2, r);
#pragma unroll
        for (int t = 0; t < 4; ++t) m[t] = vx_mask_of_bits(c, r[t]);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) m[t] = vx_drop1(c, idx0 + t);
    }
}
// Quad transpose of Philox words: the 4 lanes of an aligned quad hold words r[0..3] of four different counters (lane t: counter t).
// Returns in out[t] the word (lane & 3) of the quad's lane t, i.e. lane s ends with element s of each of the four counters.
#define VX_QUAD_BCAST(v, T) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(v), (T) * 0x55, 0xF, 0xF, true))
__device__ __forceinline__ void vx_quad_transpose4(const uint32_t (&r)[4], uint32_t (&out)[4]) {
    const int s = threadIdx.x & 3;
#define VX_QT(T)                                                                                                   \
    {                                                                                                              \
        const uint32_t w0 = VX_QUAD_BCAST(r[0], T), w1 = VX_QUAD_BCAST(r[1], T), w2 = VX_QUAD_BCAST(r[2], T), w3 = VX_QUAD_BCAST(r[3], T); \
        out[T] = s == 0 ? w0 : s == 1 ? w1 : s == 2 ? w2 : w3;                                                      \
    }
    VX_QT(0) VX_QT(1) VX_QT(2) VX_QT(3)
#undef VX_QT
}

// masks of the 4 consecutive elements (row, vox4 .. vox4+3), vox4 % 4 == 0: one Philox call
__device__ __forceinline__ void vx_masks_vox4(const VxDropCtx& dc, uint64_t row, long V, long vox4, float (&m)[4]) {
    if (!dc.on) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
    uint32_t r[4];
    vx_philox4(dc.seed, dc.stream, (row * (uint64_t)V + (uint64_t)vox4) >> 2, r);
#pragma unroll
    for (int t = 0; t < 4; ++t) m[t] = vx_mask_of_bits(dc, r[t]);
}
// masks of the elements (row0 + T * rstride, vox), T = 0..3, when the 4 lanes of a quad hold 4 consecutive voxels (vox & 3 == lane & 3): lane s draws
// the counter of row T = s and the words are handed round by a DPP quad transpose
__device__ __forceinline__ void vx_masks_rows4(const VxDropCtx& dc, uint64_t row0, uint64_t rstride, long V, long vox, float (&m)[4]) {
    if (!dc.on) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
    uint32_t r[4], w[4];
    const uint64_t row = row0 + (uint64_t)(threadIdx.x & 3) * rstride;
    vx_philox4(dc.seed, dc.stream, (row * (uint64_t)V + (uint64_t)vox) >> 2, r);
    vx_quad_transpose4(r, w);
#pragma unroll
    for (int t = 0; t < 4; ++t) m[t] = vx_mask_of_bits(dc, w[t]);
}

// ---- attention-site dropout: 16 random bits per decision, EIGHT decisions per Philox call (round 5) ---------------------------------
// The 24-bit scheme above spends one Philox4x32 call (7 rounds of quarter-rate 32-bit multiplies) on four elements; the MFMA attention forward draws one mask per score
// and the generator was ~40 % of its issue slots.  For the attention site only -- every attention kernel of pwa.hip / pwa_mfma.hip, forward and backward, draws through
// the helpers below, so the masks agree -- element idx maps to the 16-bit chunk (idx & 7) of philox(seed, stream, idx >> 3): word (idx & 7) >> 1, half idx & 1.
// p is represented as thr16 / 65536 (p = 0.1 -> 6554 / 65536 = 0.100006) and the keep scale is 1 / (1 - thr16 / 65536), so the mask stays unbiased.
static inline unsigned vx_attn_thr16(float p) { const float t = p * 65536.0f + 0.5f; return p <= 0.0f ? 0u : (t >= 65535.0f ? 65535u : (unsigned)t); }
static inline float vx_attn_keep_scale(float p) { return p > 0.0f ? 65536.0f / (65536.0f - (float)vx_attn_thr16(p)) : 1.0f; }      // host side: what a kernel that reads stored keep bits multiplies by
__device__ __forceinline__ VxDropCtx vx_attn_ctx(const VxDrop& d) {
    VxDropCtx c = vx_drop_ctx(d);
    const float t = d.p * 65536.0f + 0.5f;
    c.thr16 = c.on ? (t >= 65535.0f ? 65535u : (uint32_t)t) : 0u;
    c.inv_keep = c.on ? 65536.0f / (65536.0f - (float)c.thr16) : 1.0f;
    return c;
}
__device__ __forceinline__ float vx_attn_keep(const VxDropCtx& c, uint32_t v16) { return v16 >= c.thr16 ? c.inv_keep : 0.0f; }
__device__ __forceinline__ float vx_attn_drop1(const VxDropCtx& c, uint64_t idx) {
    if (!c.on) return 1.0f;
    uint32_t r[4];
    vx_philox4(c.seed, c.stream, idx >> 3, r);
    const uint32_t cc = (uint32_t)idx & 7u;
    return vx_attn_keep(c, (r[cc >> 1] >> (16u * (cc & 1u))) & 0xffffu);
}
// masks of the elements idx0 .. idx0 + 3; one Philox call when idx0 is a multiple of 4 (`aligned`, wave-uniform), four otherwise
__device__ __forceinline__ void vx_attn_drop4(const VxDropCtx& c, uint64_t idx0, bool aligned, float (&m)[4]) {
    if (!c.on) {
        m[0] = m[1] = m[2] = m[3] = 1.0f;
    } else if (aligned) {
        uint32_t r[4];
        vx_philox4(c.seed, c.stream, idx0 >> 3, r);
        const bool hi = ((uint32_t)idx0 & 4u) != 0u;
        const uint32_t w0 = hi ? r[2] : r[0], w1 = hi ? r[3] : r[1];
        m[0] = vx_attn_keep(c, w0 & 0xffffu); m[1] = vx_attn_keep(c, w0 >> 16); m[2] = vx_attn_keep(c, w1 & 0xffffu); m[3] = vx_attn_keep(c, w1 >> 16);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) m[t] = vx_attn_drop1(c, idx0 + t);
    }
}
// masks of the 16 consecutive elements idx0 .. idx0 + 15 (idx0 % 8 == 0): two Philox calls
__device__ __forceinline__ void vx_attn_drop16(const VxDropCtx& c, uint64_t idx0, float (&m)[16]) {
    if (!c.on) {
#pragma unroll
        for (int t = 0; t < 16; ++t) m[t] = 1.0f;
        return;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t r[4];
        vx_philox4(c.seed, c.stream, (idx0 >> 3) + h, r);
#pragma unroll
        for (int w = 0; w < 4; ++w) { m[8 * h + 2 * w] = vx_attn_keep(c, r[w] & 0xffffu); m[8 * h + 2 * w + 1] = vx_attn_keep(c, r[w] >> 16); }
    }
}
__device__ __forceinline__ void vx_attn_masks_vox4(const VxDropCtx& dc, uint64_t row, long V, long vox4, float (&m)[4]) {      // (row, vox4 .. vox4 + 3), V % 4 == 0, vox4 % 4 == 0
    vx_attn_drop4(dc, row * (uint64_t)V + (uint64_t)vox4, true, m);
}
// masks of the elements (row0 + T * rstride, vox), T = 0..3, when the 4 lanes of a quad hold 4 consecutive voxels (vox & 3 == lane & 3) and V % 4 == 0: lane s draws
// the counter of row T = s, picks the word pair of its 4-element half and the pairs are handed round with DPP quad broadcasts
__device__ __forceinline__ void vx_attn_masks_rows4(const VxDropCtx& dc, uint64_t row0, uint64_t rstride, long V, long vox, float (&m)[4]) {
    if (!dc.on) { m[0] = m[1] = m[2] = m[3] = 1.0f; return; }
    uint32_t r[4];
    const int s = threadIdx.x & 3;
    const uint64_t idx = (row0 + (uint64_t)s * rstride) * (uint64_t)V + (uint64_t)vox;
    vx_philox4(dc.seed, dc.stream, idx >> 3, r);
    const bool hi = ((uint32_t)idx & 4u) != 0u;
    const uint32_t w0 = hi ? r[2] : r[0], w1 = hi ? r[3] : r[1];
#define VX_AQ(T)                                                                                  \
    {                                                                                             \
        const uint32_t a0 = VX_QUAD_BCAST(w0, T), a1 = VX_QUAD_BCAST(w1, T);                      \
        const uint32_t w = (s & 2) ? a1 : a0;                                                     \
        m[T] = vx_attn_keep(dc, (s & 1) ? (w >> 16) : (w & 0xffffu));                             \
    }
    VX_AQ(0) VX_AQ(1) VX_AQ(2) VX_AQ(3)
#undef VX_AQ
}
