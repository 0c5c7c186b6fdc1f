// Paired-Window Attention kernels for gfx950 (fp32).
// Reference: model/components/PWA.py:106-140 (gather), :308-327 (attention), :177-200 (scatter),
//            model/components/attention_utils.py:83-125 (relative position bias).  Index maps: SURVEY.md A1.
//
// Token layout: tok[b, head a, window N (all scales concatenated), modality-major token T = m*l + t, lane c].
// Attention: one WAVE (64 lanes) = 64 query rows of one (b, head, window); K/V rows are wave-uniform so
// they travel on the scalar path (s_load) and feed v_fma as SGPR operands; the soft-max is an online
// (flash-style) per-lane recurrence, scores never leave registers.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"

// Index arithmetic of the gather / scatter kernels.  Their divisors are run-time plan constants (grid, cell and window extents), and the compiler's 32-bit integer
// division is ~30 instructions with four quarter-rate multiplies: with 10-15 of them per thread these memory movers were bound by their index math (920 instructions
// to move four floats in vx_pwa_gather_all_bwd_v_k).  For 0 <= a < 2^22: with rb = the reciprocal (v_rcp_f32: 1 ulp) scaled DOWN by 2^-22, float(a) * rb is never
// above a / b (the reciprocal's ulp and the two roundings are 2^-23 + 2^-24 + 2^-24 = 2^-22 at most) and below it by less than (a / b) 2^-21 + 1 <= 3, so the truncated
// product is the quotient or up to two below it: two conditional steps finish it (~12 full-rate instructions).  tests/test_index_math_cpu.py checks the claim in float32
// arithmetic for every reciprocal within 1 ulp.
struct VxFd { int b; float rb; };
__device__ __forceinline__ VxFd vx_fd(int b) { VxFd d; d.b = b; d.rb = __builtin_amdgcn_rcpf((float)b) * 0.99999976f; return d; }
__device__ __forceinline__ int vx_fdivmod(int a, const VxFd& d, int& r) {          // 0 <= a < 2^22
    int q = (int)((float)a * d.rb);
    r = a - __mul24(q, d.b);
    if (r >= d.b) { ++q; r -= d.b; }
    if (r >= d.b) { ++q; r -= d.b; }
    return q;
}
__device__ __forceinline__ int vx_fdiv(int a, const VxFd& d) { int r; return vx_fdivmod(a, d, r); }
// voxel / cell index of a volume -> (x0, x1, x2) for extents (., e1, e2); `small` = the volume has fewer than 2^22 elements (block-uniform)
__device__ __forceinline__ void vx_unflatten(long v, int e1, int e2, bool small, int& x0, int& x1, int& x2) {
    if (small) x0 = vx_fdivmod(vx_fdivmod((int)v, vx_fd(e2), x2), vx_fd(e1), x1);
    else { x2 = (int)(v % e2); x1 = (int)((v / e2) % e1); x0 = (int)(v / ((long)e2 * e1)); }
}

__device__ __forceinline__ int vx_scale_of_window(const VxPwaPlan& P, int N) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < P.nb && N >= P.woff[k]) i = k;
    return i;
}

// ---------------------------------------------------------------------------------------------
// gather: max-pool(s_i) of the scale-i channel slice, then window partition
// grid: (ceil(V/256), nb*h*c channels, B); thread = one pooled cell of one channel
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vx_pwa_gather_fwd_k(const float* __restrict__ src, float* __restrict__ tok, VxPwaPlan P, int c, int m, int M) {
    const int ch = blockIdx.y, b = blockIdx.z;
    const int i = ch / (P.heads * c), a = (ch / c) % P.heads, cc = ch % c;
    const int s0 = P.small[i][0], s1 = P.small[i][1], s2 = P.small[i][2];
    const int p0n = P.grid[0] / s0, p1n = P.grid[1] / s1, p2n = P.grid[2] / s2;
    const int pv = blockIdx.x * 256 + threadIdx.x;
    if (pv >= p0n * p1n * p2n) return;
    const int p2 = pv % p2n, p1 = (pv / p2n) % p1n, p0 = pv / (p2n * p1n);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const float* __restrict__ sc = src + ((long)b * (P.nb * P.heads * c) + ch) * V;
    float best = -INFINITY;
    for (int d = 0; d < s0; ++d)
        for (int h = 0; h < s1; ++h)
            for (int w = 0; w < s2; ++w) {
                const float val = sc[((long)(p0 * s0 + d) * P.grid[1] + (p1 * s1 + h)) * P.grid[2] + (p2 * s2 + w)];
                if (val > best || val != val) best = val;
            }
    const int W0 = p0 / P.n[0], t0 = p0 % P.n[0], W1 = p1 / P.n[1], t1 = p1 % P.n[1], W2 = p2 / P.n[2], t2 = p2 % P.n[2];
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
    tok[((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + cc] = best;
}

// gradient goes to the first arg-max of each pooled cell (aten max_pool3d tie rule), zero elsewhere
__global__ void __launch_bounds__(256) vx_pwa_gather_bwd_k(const float* __restrict__ src, const float* __restrict__ dtok, float* __restrict__ dsrc,
                                                           VxPwaPlan P, int c, int m, int M) {
    const int ch = blockIdx.y, b = blockIdx.z;
    const int i = ch / (P.heads * c), a = (ch / c) % P.heads, cc = ch % c;
    const int s0 = P.small[i][0], s1 = P.small[i][1], s2 = P.small[i][2];
    const int p0n = P.grid[0] / s0, p1n = P.grid[1] / s1, p2n = P.grid[2] / s2;
    const int pv = blockIdx.x * 256 + threadIdx.x;
    if (pv >= p0n * p1n * p2n) return;
    const int p2 = pv % p2n, p1 = (pv / p2n) % p1n, p0 = pv / (p2n * p1n);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const long cbase = ((long)b * (P.nb * P.heads * c) + ch) * V;
    float best = -INFINITY;
    long bidx = -1;
    for (int d = 0; d < s0; ++d)
        for (int h = 0; h < s1; ++h)
            for (int w = 0; w < s2; ++w) {
                const long idx = ((long)(p0 * s0 + d) * P.grid[1] + (p1 * s1 + h)) * P.grid[2] + (p2 * s2 + w);
                const float val = src[cbase + idx];
                if (bidx < 0 || val > best || val != val) { best = val; bidx = idx; }
            }
    const int W0 = p0 / P.n[0], t0 = p0 % P.n[0], W1 = p1 / P.n[1], t1 = p1 % P.n[1], W2 = p2 / P.n[2], t2 = p2 % P.n[2];
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
    const float g = dtok[((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + cc];
    for (int d = 0; d < s0; ++d)
        for (int h = 0; h < s1; ++h)
            for (int w = 0; w < s2; ++w) {
                const long idx = ((long)(p0 * s0 + d) * P.grid[1] + (p1 * s1 + h)) * P.grid[2] + (p2 * s2 + w);
                dsrc[cbase + idx] = (idx == bidx) ? g : 0.0f;
            }
}

// ---------------------------------------------------------------------------------------------
// fused gather of ALL q/k/v tensors of ALL modalities in one launch, arg-max saved for the backward pass.
//   A pooled cell (s0*s1*s2 voxels) is reduced cooperatively by T = min(64, cell size) adjacent lanes (shuffle reduction with the
//   aten tie rule: first index wins), so the coarse scales (8^3 = 512 voxels per cell) no longer serialise in one thread.
// ---------------------------------------------------------------------------------------------
struct VxGatherPtrs {
    const float* src[12];   // q0,k0,v0,q1,k1,v1,...
    float* dsrc[12];
};

__global__ void __launch_bounds__(256) vx_pwa_gather_all_fwd_k(VxGatherPtrs ptrs, float* __restrict__ tq, float* __restrict__ tk, float* __restrict__ tv,
                                                               int* __restrict__ iq, int* __restrict__ ik, int* __restrict__ iv,
                                                               VxPwaPlan P, int cq, int cv, int M) {
    // blockIdx.y enumerates (tensor, channel): tensors of kind q,k have nb*h*cq channels, kind v has nb*h*cv
    const int nq = P.nb * P.heads * cq, nv_ = P.nb * P.heads * cv;
    const int per_m = 2 * nq + nv_;
    const int m = blockIdx.y / per_m;
    int rch = blockIdx.y % per_m;
    int kind, c;
    if (rch < nq) { kind = 0; c = cq; } else if (rch < 2 * nq) { kind = 1; c = cq; rch -= nq; } else { kind = 2; c = cv; rch -= 2 * nq; }
    const int ch = rch, b = blockIdx.z;
    const float* __restrict__ src = ptrs.src[3 * m + kind];
    float* __restrict__ tok = kind == 0 ? tq : (kind == 1 ? tk : tv);
    int* __restrict__ tix = kind == 0 ? iq : (kind == 1 ? ik : iv);
    const int i = ch / (P.heads * c), a = (ch / c) % P.heads, cc = ch % c;
    const int s0 = P.small[i][0], s1 = P.small[i][1], s2 = P.small[i][2];
    const int csz = s0 * s1 * s2;
    int T = 1;
    while (T < 64 && T < csz) T <<= 1;
    const int p0n = P.grid[0] / s0, p1n = P.grid[1] / s1, p2n = P.grid[2] / s2;
    const int ncell = p0n * p1n * p2n;
    const int cells_per_block = 256 / T;
    const int sub = threadIdx.x % T;
    const int cell = blockIdx.x * cells_per_block + threadIdx.x / T;
    const bool cok = cell < ncell;
    const int cl = cok ? cell : 0;
    const int p2 = cl % p2n, p1 = (cl / p2n) % p1n, p0 = cl / (p2n * p1n);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const float* __restrict__ sc = src + ((long)b * (P.nb * P.heads * c) + ch) * V;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int e = sub; e < csz; e += T) {
        const int w = e % s2, h = (e / s2) % s1, d = e / (s2 * s1);
        const int idx = ((p0 * s0 + d) * P.grid[1] + (p1 * s1 + h)) * P.grid[2] + (p2 * s2 + w);
        const float val = sc[idx];
        if (val > best || bidx == 0x7fffffff) { best = val; bidx = idx; }     // increasing idx per lane: strict > keeps the first
    }
    for (int o = T >> 1; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
    }
    if (cok && sub == 0) {
        const int W0 = p0 / P.n[0], t0 = p0 % P.n[0], W1 = p1 / P.n[1], t1 = p1 % P.n[1], W2 = p2 / P.n[2], t2 = p2 % P.n[2];
        const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
        const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
        const long ti = ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + cc;
        tok[ti] = best;
        tix[ti] = bidx;
    }
}

// backward: one thread per INPUT voxel: dsrc = (saved arg-max of my cell == me) ? dtok : 0   (no memset, no atomics, coalesced stores)
__global__ void __launch_bounds__(256) vx_pwa_gather_all_bwd_k(VxGatherPtrs ptrs, const float* __restrict__ dtq, const float* __restrict__ dtk, const float* __restrict__ dtv,
                                                               const int* __restrict__ iq, const int* __restrict__ ik, const int* __restrict__ iv,
                                                               VxPwaPlan P, int cq, int cv, int M) {
    const int nq = P.nb * P.heads * cq, nv_ = P.nb * P.heads * cv;
    const int per_m = 2 * nq + nv_;
    const int m = blockIdx.y / per_m;
    int rch = blockIdx.y % per_m;
    int kind, c;
    if (rch < nq) { kind = 0; c = cq; } else if (rch < 2 * nq) { kind = 1; c = cq; rch -= nq; } else { kind = 2; c = cv; rch -= 2 * nq; }
    const int ch = rch, b = blockIdx.z;
    float* __restrict__ dsrc = ptrs.dsrc[3 * m + kind];
    const float* __restrict__ dtok = kind == 0 ? dtq : (kind == 1 ? dtk : dtv);
    const int* __restrict__ tix = kind == 0 ? iq : (kind == 1 ? ik : iv);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const int i = ch / (P.heads * c), a = (ch / c) % P.heads, cc = ch % c;
    const int x2 = (int)(v % P.grid[2]), x1 = (int)((v / P.grid[2]) % P.grid[1]), x0 = (int)(v / ((long)P.grid[2] * P.grid[1]));
    const int p0 = x0 / P.small[i][0], p1 = x1 / P.small[i][1], p2 = x2 / P.small[i][2];
    const int W0 = p0 / P.n[0], t0 = p0 % P.n[0], W1 = p1 / P.n[1], t1 = p1 % P.n[1], W2 = p2 / P.n[2], t2 = p2 % P.n[2];
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
    const long ti = ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + cc;
    const bool ident = P.small[i][0] * P.small[i][1] * P.small[i][2] == 1;       // 1 x 1 x 1 cells: the arg-max is the voxel itself (the vectorised forward stores no index there)
    dsrc[((long)b * (P.nb * P.heads * c) + ch) * V + v] = (ident || tix[ti] == (int)v) ? dtok[ti] : 0.0f;
}


// Channel-vectorised forms (c % 4 == 0, every shipped config): one lane owns CH = 4 or 8 consecutive channels of its cell / voxel, so the
// token side is read and written as 16-32 contiguous bytes per lane (neighbouring tokens follow each other in memory) instead of one float
// every c floats, while the volume side stays one coalesced access per channel.  blockIdx.y = ((m * 3 + kind) * nb + i) * heads * (c / CH) ...
template <int CH>
__global__ void __launch_bounds__(256) vx_pwa_gather_all_fwd_v_k(VxGatherPtrs ptrs, float* __restrict__ tq, float* __restrict__ tk, float* __restrict__ tv,
                                                                 int* __restrict__ iq, int* __restrict__ ik, int* __restrict__ iv,
                                                                 VxPwaPlan P, int cq, int cv, int M) {
    // blockIdx.y enumerates (m, kind, branch i, head a, channel chunk) with the chunk count of the widest kind; narrower kinds leave early
    const int nchq = cq / CH, nchv = cv / CH, nchm = nchq > nchv ? nchq : nchv;
    int y = blockIdx.y, chunk, a, i, kind;
    y = vx_fdivmod(y, vx_fd(nchm), chunk);
    y = vx_fdivmod(y, vx_fd(P.heads), a);
    y = vx_fdivmod(y, vx_fd(P.nb), i);
    const int m = vx_fdivmod(y, vx_fd(3), kind);
    const int c = kind == 2 ? cv : cq;
    if (chunk * CH >= c) return;
    const int b = blockIdx.z;
    const float* __restrict__ src = ptrs.src[3 * m + kind];
    float* __restrict__ tok = kind == 0 ? tq : (kind == 1 ? tk : tv);
    int* __restrict__ tix = kind == 0 ? iq : (kind == 1 ? ik : iv);
    const int s0 = P.small[i][0], s1 = P.small[i][1], s2 = P.small[i][2];
    const int csz = s0 * s1 * s2;
    int T = 1, lt = 0;
    while (T < 64 && T < csz) { T <<= 1; ++lt; }
    const int p0n = vx_fdiv(P.grid[0], vx_fd(s0)), p1n = vx_fdiv(P.grid[1], vx_fd(s1)), p2n = vx_fdiv(P.grid[2], vx_fd(s2));
    const int ncell = p0n * p1n * p2n;
    const int cells_per_block = 256 >> lt;
    if ((long)blockIdx.x * cells_per_block >= ncell) return;
    const int sub = threadIdx.x & (T - 1);
    const int cell = blockIdx.x * cells_per_block + (threadIdx.x >> lt);
    const bool cok = cell < ncell;
    const int cl = cok ? cell : 0;
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    int p2, p1, p0;
    vx_unflatten(cl, p1n, p2n, V < (1 << 22), p0, p1, p2);
    const int ch0 = (i * P.heads + a) * c + chunk * CH;
    const float* __restrict__ sc = src + ((long)b * (P.nb * P.heads * c) + ch0) * V;
    float best[CH];
    int bidx[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) { best[k] = -INFINITY; bidx[k] = 0x7fffffff; }
    const VxFd f2 = vx_fd(s2), f1 = vx_fd(s1);
    const int base = (p0 * s0 * P.grid[1] + p1 * s1) * P.grid[2] + p2 * s2;
    for (int e = sub; e < csz; e += T) {
        int w, h;
        const int d = vx_fdivmod(vx_fdivmod(e, f2, w), f1, h);
        const int idx = base + (d * P.grid[1] + h) * P.grid[2] + w;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float val = sc[(long)k * V + idx];
            if (val > best[k] || bidx[k] == 0x7fffffff) { best[k] = val; bidx[k] = idx; }     // increasing idx per lane: strict > keeps the first
        }
    }
    for (int o = T >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const float ov = __shfl_xor(best[k], o, 64);
            const int oi = __shfl_xor(bidx[k], o, 64);
            if (ov > best[k] || (ov == best[k] && oi < bidx[k])) { best[k] = ov; bidx[k] = oi; }
        }
    }
    if (cok && sub == 0) {
        int t0, t1, t2;
        const int W0 = vx_fdivmod(p0, vx_fd(P.n[0]), t0), W1 = vx_fdivmod(p1, vx_fd(P.n[1]), t1), W2 = vx_fdivmod(p2, vx_fd(P.n[2]), t2);
        const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
        const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
        const long ti = ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + chunk * CH;
#pragma unroll
        for (int k = 0; k < CH; k += 4) {
            *reinterpret_cast<float4*>(tok + ti + k) = make_float4(best[k], best[k + 1], best[k + 2], best[k + 3]);
            // 1 x 1 x 1 cells (the first scale of every shipped config): the arg-max is the voxel itself -- vx_pwa_gather_all_bwd_v_k knows that too, and the index
            // tensor (as large as the tokens) is neither written nor read for this scale
            if (csz > 1) *reinterpret_cast<int4*>(tix + ti + k) = make_int4(bidx[k], bidx[k + 1], bidx[k + 2], bidx[k + 3]);
        }
    }
}

template <int CH>
__global__ void __launch_bounds__(256) vx_pwa_gather_all_bwd_v_k(VxGatherPtrs ptrs, const float* __restrict__ dtq, const float* __restrict__ dtk, const float* __restrict__ dtv,
                                                                 const int* __restrict__ iq, const int* __restrict__ ik, const int* __restrict__ iv,
                                                                 VxPwaPlan P, int cq, int cv, int M) {
    const int nchq = cq / CH, nchv = cv / CH, nchm = nchq > nchv ? nchq : nchv;
    int y = blockIdx.y, chunk, a, i, kind;
    y = vx_fdivmod(y, vx_fd(nchm), chunk);
    y = vx_fdivmod(y, vx_fd(P.heads), a);
    y = vx_fdivmod(y, vx_fd(P.nb), i);
    const int m = vx_fdivmod(y, vx_fd(3), kind);
    const int c = kind == 2 ? cv : cq;
    if (chunk * CH >= c) return;
    const int b = blockIdx.z;
    float* __restrict__ dsrc = ptrs.dsrc[3 * m + kind];
    const float* __restrict__ dtok = kind == 0 ? dtq : (kind == 1 ? dtk : dtv);
    const int* __restrict__ tix = kind == 0 ? iq : (kind == 1 ? ik : iv);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    int x2, x1, x0;
    vx_unflatten(v, P.grid[1], P.grid[2], V < (1 << 22), x0, x1, x2);
    // cell -> (window, token) per axis: x / small = p,  p / n = W,  p % n = t
    int t0, t1, t2;
    const int W0 = vx_fdivmod(vx_fdiv(x0, vx_fd(P.small[i][0])), vx_fd(P.n[0]), t0);
    const int W1 = vx_fdivmod(vx_fdiv(x1, vx_fd(P.small[i][1])), vx_fd(P.n[1]), t1);
    const int W2 = vx_fdivmod(vx_fdiv(x2, vx_fd(P.small[i][2])), vx_fd(P.n[2]), t2);
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const int t = (t0 * P.n[1] + t1) * P.n[2] + t2;
    const long ti = ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l + t) * c + chunk * CH;
    const int ch0 = (i * P.heads + a) * c + chunk * CH;
    float* __restrict__ dc = dsrc + ((long)b * (P.nb * P.heads * c) + ch0) * V + v;
    const bool ident = P.small[i][0] * P.small[i][1] * P.small[i][2] == 1;       // (see the forward: no indices for 1 x 1 x 1 cells)
#pragma unroll
    for (int k = 0; k < CH; k += 4) {
        const float4 g = *reinterpret_cast<const float4*>(dtok + ti + k);
        const int4 ix = ident ? make_int4((int)v, (int)v, (int)v, (int)v) : *reinterpret_cast<const int4*>(tix + ti + k);
        dc[(long)(k + 0) * V] = ix.x == (int)v ? g.x : 0.0f;
        dc[(long)(k + 1) * V] = ix.y == (int)v ? g.y : 0.0f;
        dc[(long)(k + 2) * V] = ix.z == (int)v ? g.z : 0.0f;
        dc[(long)(k + 3) * V] = ix.w == (int)v ? g.w : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// scatter: per-window trilinear up-sampling (align_corners=True) of the n^3 window outputs
// ---------------------------------------------------------------------------------------------
// all M modalities of one direction in ONE launch: grid.x = (blocks of one modality) * M; block (mm, bx) works on modality m0 + mm with its own tensor
struct VxScPtrs { const float* in[4]; float* out[4]; };
// (round 6) the window ranges of the scales whose adjoint kernels ADD with atomics: the identity-scale kernel, which runs first on the stream, zeroes them (no separate fill launch)
struct VxScZero { int n; int w0[4], wn[4]; int assign; float* extra; long extra_n4; };      // extra: one more buffer the identity-scale kernel zeroes (float4 units; the attention backward's bias-gradient replicas)

__device__ __forceinline__ void vx_src_coord(int j, int n, int bw, int& i0, int& i1, float& lam) {
    if (bw == n) { i0 = j; i1 = j; lam = 0.0f; return; }
    const float ratio = (float)(n - 1) / (float)(bw - 1);
    const float s = ratio * (float)j;
    i0 = (int)s;
    lam = s - (float)i0;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
}

__global__ void __launch_bounds__(256) vx_pwa_scatter_fwd_k(const float* __restrict__ tok, VxScPtrs ptrs, VxPwaPlan P, int c, int m0, int M, int nx) {
    const int mm = blockIdx.x / nx, bx = blockIdx.x - mm * nx;
    const int m = m0 + mm;
    float* __restrict__ out = ptrs.out[mm];
    const int ch = blockIdx.y, b = blockIdx.z;
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const long v = (long)bx * 256 + threadIdx.x;
    if (v >= V) return;
    const int i = ch / (P.heads * c), a = (ch / c) % P.heads, cc = ch % c;
    const int x2 = (int)(v % P.grid[2]), x1 = (int)((v / P.grid[2]) % P.grid[1]), x0 = (int)(v / ((long)P.grid[2] * P.grid[1]));
    const int bw0 = P.n[0] * P.small[i][0], bw1 = P.n[1] * P.small[i][1], bw2 = P.n[2] * P.small[i][2];
    const int W0 = x0 / bw0, W1 = x1 / bw1, W2 = x2 / bw2;
    int a0, b0, a1, b1, a2, b2;
    float l0, l1, l2;
    vx_src_coord(x0 % bw0, P.n[0], bw0, a0, b0, l0);
    vx_src_coord(x1 % bw1, P.n[1], bw1, a1, b1, l1);
    vx_src_coord(x2 % bw2, P.n[2], bw2, a2, b2, l2);
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const float* __restrict__ tw = tok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l) * c + cc;
    auto T = [&](int t0, int t1, int t2) { return tw[(long)((t0 * P.n[1] + t1) * P.n[2] + t2) * c]; };
    const float k0 = 1.0f - l0, k1 = 1.0f - l1, k2 = 1.0f - l2;
    const float val = k0 * (k1 * (k2 * T(a0, a1, a2) + l2 * T(a0, a1, b2)) + l1 * (k2 * T(a0, b1, a2) + l2 * T(a0, b1, b2))) +
                      l0 * (k1 * (k2 * T(b0, a1, a2) + l2 * T(b0, a1, b2)) + l1 * (k2 * T(b0, b1, a2) + l2 * T(b0, b1, b2)));
    out[((long)b * (P.nb * P.heads * c) + ch) * V + v] = val;
}

// the same with 4 channels per lane (c % 4 == 0): the 8 taps of a voxel are 8 float4 reads of whole token rows instead of 32 scalar reads at stride c
__global__ void __launch_bounds__(256) vx_pwa_scatter_fwd_v_k(const float* __restrict__ tok, VxScPtrs ptrs, VxPwaPlan P, int c, int m0, int M, int nx) {
    int bx;
    const int mm = vx_fdivmod(blockIdx.x, vx_fd(nx), bx);
    const int m = m0 + mm;
    float* __restrict__ out = ptrs.out[mm];
    const int c4 = c >> 2;
    const int chq = blockIdx.y, b = blockIdx.z;                 // chq enumerates (scale i, head a, channel quad)
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const long v = (long)bx * 256 + threadIdx.x;
    if (v >= V) return;
    int q4, a;
    const int i = vx_fdivmod(vx_fdivmod(chq, vx_fd(c4), q4), vx_fd(P.heads), a);
    int x2, x1, x0;
    vx_unflatten(v, P.grid[1], P.grid[2], V < (1 << 22), x0, x1, x2);
    const int bw0 = P.n[0] * P.small[i][0], bw1 = P.n[1] * P.small[i][1], bw2 = P.n[2] * P.small[i][2];
    int j0, j1, j2;
    const int W0 = vx_fdivmod(x0, vx_fd(bw0), j0), W1 = vx_fdivmod(x1, vx_fd(bw1), j1), W2 = vx_fdivmod(x2, vx_fd(bw2), j2);
    int a0, b0, a1, b1, a2, b2;
    float l0, l1, l2;
    vx_src_coord(j0, P.n[0], bw0, a0, b0, l0);
    vx_src_coord(j1, P.n[1], bw1, a1, b1, l1);
    vx_src_coord(j2, P.n[2], bw2, a2, b2, l2);
    const int N = P.woff[i] + (W0 * P.nwin[i][1] + W1) * P.nwin[i][2] + W2;
    const float* __restrict__ tw = tok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l) * c + 4 * q4;
    auto T = [&](int t0, int t1, int t2) { return *reinterpret_cast<const float4*>(tw + (long)((t0 * P.n[1] + t1) * P.n[2] + t2) * c); };
    const float k0 = 1.0f - l0, k1 = 1.0f - l1, k2 = 1.0f - l2;
    const float4 t000 = T(a0, a1, a2), t001 = T(a0, a1, b2), t010 = T(a0, b1, a2), t011 = T(a0, b1, b2);
    const float4 t100 = T(b0, a1, a2), t101 = T(b0, a1, b2), t110 = T(b0, b1, a2), t111 = T(b0, b1, b2);
    // (the scalar kernel's nesting of the lerps, so that the values are bit-identical)
#define VX_SCF(f) (k0 * (k1 * (k2 * t000.f + l2 * t001.f) + l1 * (k2 * t010.f + l2 * t011.f)) + l0 * (k1 * (k2 * t100.f + l2 * t101.f) + l1 * (k2 * t110.f + l2 * t111.f)))
    float* __restrict__ ob = out + ((long)b * (P.nb * P.heads * c) + (long)(i * P.heads + a) * c + 4 * q4) * V + v;
    ob[0] = VX_SCF(x);
    ob[V] = VX_SCF(y);
    ob[2 * V] = VX_SCF(z);
    ob[3 * V] = VX_SCF(w);
#undef VX_SCF
}

// adjoint of the scatter: one block = (b, head, window, voxel chunk); the window's l x c token gradients are accumulated in LDS
// (ds_add_f32) from the block's output voxels (8 corners each), then flushed with one float atomic per token element.
__global__ void __launch_bounds__(256) vx_pwa_scatter_bwd_k(VxScPtrs ptrs, float* __restrict__ dtok, VxPwaPlan P, int c, int m0, int M, int scale, int nx) {
    const int mm = blockIdx.x / nx, bx = blockIdx.x - mm * nx;
    const int m = m0 + mm;
    const float* __restrict__ dout = ptrs.in[mm];
    extern __shared__ __attribute__((aligned(16))) float vx_sacc[];      // [l][c]
    const int b = blockIdx.z / P.heads, a = blockIdx.z % P.heads;
    const int i = scale;
    const int N = P.woff[i] + blockIdx.y;
    const int bw0 = P.n[0] * P.small[i][0], bw1 = P.n[1] * P.small[i][1], bw2 = P.n[2] * P.small[i][2];
    const int nv = bw0 * bw1 * bw2;
    const int items = nv * c;
    const int chunks = nx;                            // (per modality)
    for (int k = threadIdx.x; k < P.l * c; k += 256) vx_sacc[k] = 0.0f;
    __syncthreads();
    const int Nl = N - P.woff[i];
    const int W2 = Nl % P.nwin[i][2], W1 = (Nl / P.nwin[i][2]) % P.nwin[i][1], W0 = Nl / (P.nwin[i][2] * P.nwin[i][1]);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const int chbase = (i * P.heads + a) * c;
    const float* __restrict__ db = dout + ((long)b * (P.nb * P.heads * c) + chbase) * V;
    for (int e = bx * 256 + threadIdx.x; e < items; e += chunks * 256) {
        const int vox = e % nv, cc = e / nv;
        const int j2 = vox % bw2, j1 = (vox / bw2) % bw1, j0 = vox / (bw2 * bw1);
        int a0, b0, a1, b1, a2, b2;
        float l0, l1, l2;
        vx_src_coord(j0, P.n[0], bw0, a0, b0, l0);
        vx_src_coord(j1, P.n[1], bw1, a1, b1, l1);
        vx_src_coord(j2, P.n[2], bw2, a2, b2, l2);
        const float g = db[(long)cc * V + ((long)(W0 * bw0 + j0) * P.grid[1] + (W1 * bw1 + j1)) * P.grid[2] + (W2 * bw2 + j2)];
        const float k0 = 1.0f - l0, k1 = 1.0f - l1, k2 = 1.0f - l2;
        auto A = [&](int t0, int t1, int t2, float wt) { if (wt != 0.0f) atomicAdd(&vx_sacc[((t0 * P.n[1] + t1) * P.n[2] + t2) * c + cc], wt * g); };
        A(a0, a1, a2, k0 * k1 * k2); A(a0, a1, b2, k0 * k1 * l2); A(a0, b1, a2, k0 * l1 * k2); A(a0, b1, b2, k0 * l1 * l2);
        A(b0, a1, a2, l0 * k1 * k2); A(b0, a1, b2, l0 * k1 * l2); A(b0, b1, a2, l0 * l1 * k2); A(b0, b1, b2, l0 * l1 * l2);
    }
    __syncthreads();
    float* __restrict__ dt = dtok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l) * c;
    for (int k = threadIdx.x; k < P.l * c; k += 256) {
        const float sacc = vx_sacc[k];
        if (sacc != 0.0f) atomicAdd(dt + k, sacc);
    }
}

// small window 1x1x1 (every shipped config: min_small_window_sizes = [[1,1,1]] * 4): the per-window resampling is the identity, so the adjoint
// is a transpose of the window's (c, voxels) slab into its (tokens, c) rows -- staged through LDS (pitch c + 1), coalesced on both sides, no atomics.
// One block = one (b, head, window); every element of the destination rows is written exactly once, which equals "+=" on the zeroed buffer.
__global__ void __launch_bounds__(256) vx_pwa_scatter_bwd_ident_k(VxScPtrs ptrs, float* __restrict__ dtok, VxPwaPlan P, int c, int m0, int M, int scale, int nx, VxScZero z) {
    int bx;
    const int mm = vx_fdivmod(blockIdx.x, vx_fd(nx), bx);
    const int m = m0 + mm;
    const float* __restrict__ dout = ptrs.in[mm];
    extern __shared__ __attribute__((aligned(16))) float vx_sacc[];      // [l][c + 1]
    int a;
    const int b = vx_fdivmod(blockIdx.y, vx_fd(P.heads), a);
    const int i = scale;
    const int Nl = bx, N = P.woff[i] + Nl;
    const int n0 = P.n[0], n1 = P.n[1], n2 = P.n[2];
    int W2, W1;
    const int W0 = vx_fdivmod(vx_fdivmod(Nl, vx_fd(P.nwin[i][2]), W2), vx_fd(P.nwin[i][1]), W1);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const float* __restrict__ db = dout + ((long)b * (P.nb * P.heads * c) + (long)(i * P.heads + a) * c) * V;
    const int l = P.l, pitch = c + 1;
    const VxFd fl = vx_fd(l), fn2 = vx_fd(n2), fn1 = vx_fd(n1), fc = vx_fd(c);
    const long wbase = ((long)(W0 * n0) * P.grid[1] + W1 * n1) * P.grid[2] + W2 * n2;
    for (int e = threadIdx.x; e < l * c; e += 256) {
        int vox, j2, j1;
        const int cc = vx_fdivmod(e, fl, vox);
        const int j0 = vx_fdivmod(vx_fdivmod(vox, fn2, j2), fn1, j1);
        vx_sacc[vox * pitch + cc] = db[(long)cc * V + wbase + ((long)j0 * P.grid[1] + j1) * P.grid[2] + j2];
    }
    __syncthreads();
    float* __restrict__ dt = dtok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * l) + (long)m * l) * c;
    if (z.assign) {                                      // (round 6) the destination is NOT zeroed by the caller: this kernel writes its rows and zeroes the rows of the atomic scales
        for (int k = threadIdx.x; k < l * c; k += 256) { int kc; const int kt = vx_fdivmod(k, fc, kc); dt[k] = vx_sacc[kt * pitch + kc]; }
        for (int r = 0; r < z.n; ++r) {
            float* __restrict__ zb = dtok + (((long)b * P.heads + a) * P.Ntot + z.w0[r]) * ((long)M * l) * c;
            const long len = (long)z.wn[r] * M * l * c;
            for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < len; k += (long)gridDim.x * 256) zb[k] = 0.0f;
        }
        if (z.extra != nullptr) {
            float4* __restrict__ e4 = reinterpret_cast<float4*>(z.extra);
            const long nb = (long)gridDim.x * gridDim.y, bid = blockIdx.x + (long)gridDim.x * blockIdx.y;
            for (long k = bid * 256 + threadIdx.x; k < z.extra_n4; k += nb * 256) e4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    for (int k = threadIdx.x; k < l * c; k += 256) { int kc; const int kt = vx_fdivmod(k, fc, kc); dt[k] += vx_sacc[kt * pitch + kc]; }       // sole owner of these rows: "+=" without atomics
}

// General small window (2x2x2 .. 4x4x4): the adjoint as a GATHER, one block = one (window, b*head, channel).  Every token collects, per axis, the
// <= VX_SC_TAPS output voxels whose trilinear stencil touches it (per-axis tap tables built once per block in LDS) from the channel's voxel slab
// staged in LDS; PT = 256 / l threads share a token (they split the outermost tap axis) and are summed with shuffles.  Replaces 8 scattered
// ds_add_f32 per voxel of the atomic kernel below (45 us per launch at the second scale) by 8 LDS reads per voxel.
#define VX_SC_TAPS 12
__global__ void __launch_bounds__(256) vx_pwa_scatter_bwd_gather_k(VxScPtrs ptrs, float* __restrict__ dtok, VxPwaPlan P, int c, int m0, int M, int scale, int PT, int nx, int assign) {
    int bx;
    const int mm = vx_fdivmod(blockIdx.x, vx_fd(nx), bx);
    const int m = m0 + mm;
    const float* __restrict__ dout = ptrs.in[mm];
    extern __shared__ __attribute__((aligned(16))) float vx_sacc[];      // [nv] slab | taps
    int a;
    const int b = vx_fdivmod(blockIdx.y, vx_fd(P.heads), a);
    const int cc = blockIdx.z;
    const int i = scale;
    const int Nl = bx, N = P.woff[i] + Nl;
    const int n0 = P.n[0], n1 = P.n[1], n2 = P.n[2];
    const int bw0 = n0 * P.small[i][0], bw1 = n1 * P.small[i][1], bw2 = n2 * P.small[i][2];
    const int nv = bw0 * bw1 * bw2, l = P.l;
    float* __restrict__ slab = vx_sacc;
    const int nmax = max(n0, max(n1, n2));
    float* __restrict__ tapw = slab + ((nv + 3) & ~3);                                    // [3][nmax][TAPS]
    int* __restrict__ tapj = reinterpret_cast<int*>(tapw + 3 * nmax * VX_SC_TAPS);        // [3][nmax][TAPS]
    int* __restrict__ tapn = tapj + 3 * nmax * VX_SC_TAPS;                                // [3][nmax]
    if (threadIdx.x < 3) {
        const int ax = threadIdx.x;
        const int n = ax == 0 ? n0 : ax == 1 ? n1 : n2, bw = ax == 0 ? bw0 : ax == 1 ? bw1 : bw2;
        for (int t = 0; t < n; ++t) tapn[ax * nmax + t] = 0;
        for (int j = 0; j < bw; ++j) {
            int t_a, t_b; float lam;
            vx_src_coord(j, n, bw, t_a, t_b, lam);
            int& ca = tapn[ax * nmax + t_a];
            tapj[(ax * nmax + t_a) * VX_SC_TAPS + ca] = j; tapw[(ax * nmax + t_a) * VX_SC_TAPS + ca] = 1.0f - lam; ++ca;
            if (lam != 0.0f) {
                int& cb = tapn[ax * nmax + t_b];
                tapj[(ax * nmax + t_b) * VX_SC_TAPS + cb] = j; tapw[(ax * nmax + t_b) * VX_SC_TAPS + cb] = lam; ++cb;
            }
        }
    }
    int W2, W1;
    const int W0 = vx_fdivmod(vx_fdivmod(Nl, vx_fd(P.nwin[i][2]), W2), vx_fd(P.nwin[i][1]), W1);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const float* __restrict__ db = dout + ((long)b * (P.nb * P.heads * c) + (long)(i * P.heads + a) * c + cc) * V;
    const VxFd fb2 = vx_fd(bw2), fb1 = vx_fd(bw1), fn2 = vx_fd(n2), fn1 = vx_fd(n1);
    for (int e = threadIdx.x; e < nv; e += 256) {
        int j2, j1;
        const int j0 = vx_fdivmod(vx_fdivmod(e, fb2, j2), fb1, j1);
        slab[e] = db[((long)(W0 * bw0 + j0) * P.grid[1] + (W1 * bw1 + j1)) * P.grid[2] + (W2 * bw2 + j2)];
    }
    __syncthreads();
    float* __restrict__ dt = dtok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * l) + (long)m * l) * c + cc;
    int sub;                                                // PT threads (adjacent lanes) per token: they split the taps of axis 0
    const int tfirst = vx_fdivmod(threadIdx.x, vx_fd(PT), sub), tstep = vx_fdiv(256, vx_fd(PT));
    for (int t = tfirst; t < l; t += tstep) {
        int t2, t1;
        const int t0 = vx_fdivmod(vx_fdivmod(t, fn2, t2), fn1, t1);
        const int c0 = tapn[t0], c1 = tapn[nmax + t1], c2 = tapn[2 * nmax + t2];
        const int* __restrict__ J0 = tapj + t0 * VX_SC_TAPS; const float* __restrict__ Wt0 = tapw + t0 * VX_SC_TAPS;
        const int* __restrict__ J1 = tapj + (nmax + t1) * VX_SC_TAPS; const float* __restrict__ Wt1 = tapw + (nmax + t1) * VX_SC_TAPS;
        const int* __restrict__ J2 = tapj + (2 * nmax + t2) * VX_SC_TAPS; const float* __restrict__ Wt2 = tapw + (2 * nmax + t2) * VX_SC_TAPS;
        float acc = 0.0f;
        for (int u0 = sub; u0 < c0; u0 += PT)
            for (int u1 = 0; u1 < c1; ++u1) {
                const float w01 = Wt0[u0] * Wt1[u1];
                const float* __restrict__ row = slab + (J0[u0] * bw1 + J1[u1]) * bw2;
                float r = 0.0f;
                for (int u2 = 0; u2 < c2; ++u2) r = fmaf(Wt2[u2], row[J2[u2]], r);
                acc = fmaf(w01, r, acc);
            }
        for (int o = 1; o < PT; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0) { if (assign) dt[(long)t * c] = acc; else dt[(long)t * c] += acc; }               // sole owner of this element: "+=" without atomics (assign: the destination was not zeroed)
    }
}


// Large small-windows (cells wider than 4: the coarse scale of the 32^3 level has 8^3-voxel cells, one window = the whole volume): the adjoint as three
// separable 1-D reductions.  One block = (window, b * head, channel, chunk of ZC planes of the window's voxel box).  Per plane: stage it in LDS
// (coalesced), reduce along W with the axis' tap weights (thread = (row, coarse column)), then along H (thread = (coarse row, coarse column)) and
// add the plane's n1 x n2 result into the block's n0 x n1 x n2 accumulator with the two D weights of that plane; one float atomic per token
// element and block at the end.  Replaces 8 scattered ds_add_f32 per voxel of vx_pwa_scatter_bwd_k (58 us per launch there) by plain LDS reads.
__global__ void __launch_bounds__(256) vx_pwa_scatter_bwd_sep_k(VxScPtrs ptrs, float* __restrict__ dtok, VxPwaPlan P, int c, int m0, int M, int scale, int ZC, int nx) {
    int bx;
    const int mm = vx_fdivmod(blockIdx.x, vx_fd(nx), bx);
    const int m = m0 + mm;
    const float* __restrict__ dout = ptrs.in[mm];
    extern __shared__ __attribute__((aligned(16))) float vx_sacc[];
    int a;
    const int b = vx_fdivmod(blockIdx.y, vx_fd(P.heads), a);
    const int i = scale;
    int cc;
    const int zc = vx_fdivmod(blockIdx.z, vx_fd(c), cc);
    const int Nl = bx, N = P.woff[i] + Nl;
    const int n0 = P.n[0], n1 = P.n[1], n2 = P.n[2];
    const int bw0 = n0 * P.small[i][0], bw1 = n1 * P.small[i][1], bw2 = n2 * P.small[i][2];
    float* __restrict__ plane = vx_sacc;                        // [bw1][bw2]
    float* __restrict__ tmp = plane + bw1 * bw2;                // [bw1][n2]
    float* __restrict__ acc = tmp + bw1 * n2;                   // [n0][n1][n2]
    float* __restrict__ A2 = acc + n0 * n1 * n2;                // [n2][bw2]  weight of fine column j2 for coarse column t2
    float* __restrict__ A1 = A2 + n2 * bw2;                     // [n1][bw1]
    const VxFd fb2 = vx_fd(bw2), fb1 = vx_fd(bw1), fn2 = vx_fd(n2);
    for (int e = threadIdx.x; e < n2 * bw2; e += 256) {
        int j;
        const int t = vx_fdivmod(e, fb2, j);
        int i0, i1; float lam;
        vx_src_coord(j, n2, bw2, i0, i1, lam);
        A2[e] = (i0 == t ? 1.0f - lam : 0.0f) + (i1 == t ? lam : 0.0f);
    }
    for (int e = threadIdx.x; e < n1 * bw1; e += 256) {
        int j;
        const int t = vx_fdivmod(e, fb1, j);
        int i0, i1; float lam;
        vx_src_coord(j, n1, bw1, i0, i1, lam);
        A1[e] = (i0 == t ? 1.0f - lam : 0.0f) + (i1 == t ? lam : 0.0f);
    }
    for (int e = threadIdx.x; e < n0 * n1 * n2; e += 256) acc[e] = 0.0f;
    int W2, W1;
    const int W0 = vx_fdivmod(vx_fdivmod(Nl, vx_fd(P.nwin[i][2]), W2), vx_fd(P.nwin[i][1]), W1);
    const long V = (long)P.grid[0] * P.grid[1] * P.grid[2];
    const float* __restrict__ db = dout + ((long)b * (P.nb * P.heads * c) + (long)(i * P.heads + a) * c + cc) * V;
    const int jz0 = zc * ZC, jz1 = min(bw0, jz0 + ZC);
    for (int j0 = jz0; j0 < jz1; ++j0) {
        __syncthreads();
        const float* __restrict__ src = db + ((long)(W0 * bw0 + j0) * P.grid[1] + W1 * bw1) * P.grid[2] + W2 * bw2;
        for (int e = threadIdx.x; e < bw1 * bw2; e += 256) {
            int j2;
            const int j1 = vx_fdivmod(e, fb2, j2);
            plane[e] = src[(long)j1 * P.grid[2] + j2];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < bw1 * n2; e += 256) {      // along W
            int t2;
            const int j1 = vx_fdivmod(e, fn2, t2);
            const float* __restrict__ row = plane + j1 * bw2;
            const float* __restrict__ w2 = A2 + t2 * bw2;
            float s_ = 0.0f;
            for (int j2 = 0; j2 < bw2; ++j2) s_ = fmaf(w2[j2], row[j2], s_);
            tmp[e] = s_;
        }
        __syncthreads();
        int t0a, t0b; float l0;
        vx_src_coord(j0, n0, bw0, t0a, t0b, l0);
        for (int e = threadIdx.x; e < n1 * n2; e += 256) {       // along H, then into the two coarse planes this fine plane touches
            int t2;
            const int t1 = vx_fdivmod(e, fn2, t2);
            const float* __restrict__ w1 = A1 + t1 * bw1;
            float s_ = 0.0f;
            for (int j1 = 0; j1 < bw1; ++j1) s_ = fmaf(w1[j1], tmp[j1 * n2 + t2], s_);
            acc[(t0a * n1 + t1) * n2 + t2] += (1.0f - l0) * s_;      // each (t1, t2) is owned by one thread; planes are walked one after the other
            if (t0b != t0a) acc[(t0b * n1 + t1) * n2 + t2] += l0 * s_;
            else acc[(t0a * n1 + t1) * n2 + t2] += l0 * s_;
        }
    }
    __syncthreads();
    float* __restrict__ dt = dtok + ((((long)b * P.heads + a) * P.Ntot + N) * ((long)M * P.l) + (long)m * P.l) * c + cc;
    for (int t = threadIdx.x; t < n0 * n1 * n2; t += 256) {
        const float v = acc[t];
        if (v != 0.0f) atomicAdd(dt + (long)t * c, v);
    }
}

// ---------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------
struct VxAttn {
    int BH;        // B*heads
    int heads;
    int Nt;        // windows
    int ML, l, M;  // tokens per window (all modalities), per modality, modalities
    int n[3];
    int cq, cv;
    float scale;
    int lin_cst;   // (n0-1)(2n1-1)(2n2-1) + (n1-1)(2n2-1) + (n2-1)
};

__device__ __forceinline__ int vx_lin_of_token(const VxAttn& A, int T) {
    const int t = T % A.l;
    const int t2 = t % A.n[2], t1 = (t / A.n[2]) % A.n[1], t0 = t / (A.n[2] * A.n[1]);
    return (t0 * (2 * A.n[1] - 1) + t1) * (2 * A.n[2] - 1) + t2;
}

// Per-block LDS tables shared by the attention kernels: lin[t] (t < l) and the bias table of every head (Tsz x heads floats).
// LDS layout (dynamic): [lin: l ints][bias: heads*Tsz floats][per-wave slabs ...]
// head >= 0: only that head's column is staged (bias[k]); a block whose units all belong to one head needs no more, and the 8^3-window level's
// 15^3-entry table then takes 13.5 KB of LDS instead of 54 KB (the difference between one and three resident blocks per CU in the backward)
__device__ __forceinline__ void vx_attn_tables(const VxAttn& A, const float* __restrict__ table, int Tsz, int* lin, float* bias, int head = -1) {
    for (int t = threadIdx.x; t < A.l; t += 256) {
        const int t2 = t % A.n[2], t1 = (t / A.n[2]) % A.n[1], t0 = t / (A.n[2] * A.n[1]);
        lin[t] = (t0 * (2 * A.n[1] - 1) + t1) * (2 * A.n[2] - 1) + t2;
    }
    if (head >= 0) {
        for (int k = threadIdx.x; k < Tsz; k += 256) bias[k] = table[(long)k * A.heads + head];
        return;
    }
    for (int e = threadIdx.x; e < Tsz * A.heads; e += 256) {       // transpose (Tsz, heads) -> [head][Tsz]
        const int a = e / Tsz, k = e % Tsz;
        bias[e] = table[(long)k * A.heads + a];
    }
}

// unit = (bh, window, 64-query chunk).  A block is 4 waves = (4/S) units x S key splits: wave (u, s) walks the 64-row K/V slabs
// s, s+S, s+2S, ... of its window (staged in a per-wave LDS slab: coalesced global read, one row per lane, consumed as broadcast
// ds_read_b128) and the S partial soft-max states of a unit are merged through LDS.  S > 1 keeps the 1024-key windows of the 8^3
// level from running one wave per SIMD.  Attention dropout: one Philox call per 4 consecutive keys (vx_drop4).
#define VX_KV_ROWS 64
template <int CQ, int CV>
__global__ void __launch_bounds__(256) vx_pwa_attn_fwd_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                         const float* __restrict__ table, float* __restrict__ O, float* __restrict__ LSE,
                                                         int Tsz, VxAttn A, VxDrop drop, int S, int one_head, unsigned short* __restrict__ mbits) {
    constexpr int RS = CQ + CV;
    static_assert(RS >= CV + 2, "the merge reuses the slab rows");
    extern __shared__ __attribute__((aligned(16))) float vx_sm[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_sm);
    const int lin_pad = (A.l + 3) & ~3;
    float* __restrict__ bias_all = vx_sm + lin_pad;
    float* __restrict__ slabs = bias_all + (((long)Tsz * (one_head ? 1 : A.heads) + 3) & ~3);
    const int chunks = (A.ML + 63) / 64;
    const long units = (long)A.BH * A.Nt * chunks;
    vx_attn_tables(A, table, Tsz, lin, bias_all, one_head ? (int)((((long)blockIdx.x * (4 / S) / chunks) / A.Nt) % A.heads) : -1);
    __syncthreads();
    const VxDropCtx dc = vx_attn_ctx(drop);
    const bool al4 = (A.ML & 3) == 0;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = wave % S;
    const long u_raw = (long)blockIdx.x * (4 / S) + wave / S;
    const bool active = u_raw < units;
    const long u = active ? u_raw : units - 1;
    float* __restrict__ slab = slabs + (long)wave * VX_KV_ROWS * RS;
    const int lane = threadIdx.x & 63;
    const int chunk = (int)(u % chunks);
    const long win = u / chunks;                       // (bh*Nt + N)
    const int a = (int)((win / A.Nt) % A.heads);
    const float* __restrict__ bias = bias_all + (one_head ? 0L : (long)a * Tsz);
    const int i = chunk * 64 + lane;
    const bool ok = active && i < A.ML;
    const int iq = (i < A.ML) ? i : A.ML - 1;
    const float* __restrict__ qp = Q + (win * A.ML + iq) * CQ;
    float q[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) q[c] = qp[c] * A.scale;
    const int lin_i = lin[iq % A.l] + A.lin_cst;
    float acc[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) acc[c] = 0.0f;
    float mrun = -INFINITY, lsum = 0.0f;
    const float* __restrict__ kp = K + win * A.ML * CQ;
    const float* __restrict__ vp = Vt + win * A.ML * CV;
    const uint64_t drow = ((uint64_t)win * A.ML + iq) * (uint64_t)A.ML;
    for (int j0 = split * VX_KV_ROWS; j0 < A.ML; j0 += S * VX_KV_ROWS) {
        const int nk = min(VX_KV_ROWS, A.ML - j0);
        int tj = j0 % A.l;                             // key token index modulo l (keys are modality-major)
        __builtin_amdgcn_wave_barrier();
        if (lane < nk) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) slab[lane * RS + c] = kp[(long)(j0 + lane) * CQ + c];
#pragma unroll
            for (int c = 0; c < CV; ++c) slab[lane * RS + CQ + c] = vp[(long)(j0 + lane) * CV + c];
        }
        __builtin_amdgcn_wave_barrier();
        unsigned wb = 0;                               // keep bits of the 16 keys of the current mask word (vx_pwa_attn_mbits_words)
        for (int jj = 0; jj < nk; jj += 4) {
            float m4[4];
            vx_attn_drop4(dc, drow + j0 + jj, al4, m4);
            if (mbits != nullptr && dc.on) {
#pragma unroll
                for (int t = 0; t < 4; ++t) wb |= (m4[t] != 0.0f ? 1u : 0u) << ((jj + t) & 15);
                if (((jj + 4) & 15) == 0 || jj + 4 >= nk) {
                    if (ok) mbits[(win * ((A.ML + 15) >> 4) + ((j0 + jj) >> 4)) * A.ML + i] = (unsigned short)wb;
                    wb = 0;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (jj + t < nk) {
                    const float* __restrict__ row = slab + (jj + t) * RS;
                    float s = 0.0f;
#pragma unroll
                    for (int c = 0; c < CQ; ++c) s = fmaf(q[c], row[c], s);
                    s += bias[lin_i - lin[tj]];
                    if (++tj == A.l) tj = 0;
                    const float mn = fmaxf(mrun, s);
                    const float alpha = __expf(mrun - mn);
                    const float p = __expf(s - mn);
                    lsum = lsum * alpha + p;
                    const float pd = p * m4[t];
#pragma unroll
                    for (int c = 0; c < CV; ++c) acc[c] = fmaf(pd, row[CQ + c], acc[c] * alpha);
                    mrun = mn;
                }
            }
        }
    }
    if (S > 1) {                                       // merge the S partial states of a unit (every split saw >= 1 slab: launcher)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CV; ++c) slab[lane * RS + c] = acc[c];
        slab[lane * RS + CV] = mrun;
        slab[lane * RS + CV + 1] = lsum;
        __syncthreads();
        if (split != 0) return;
        for (int s2 = 1; s2 < S; ++s2) {
            const float* __restrict__ o = slab + (long)s2 * VX_KV_ROWS * RS + lane * RS;
            const float m2 = o[CV], l2 = o[CV + 1];
            const float mn = fmaxf(mrun, m2);
            const float a1 = __expf(mrun - mn), a2 = __expf(m2 - mn);
            lsum = lsum * a1 + l2 * a2;
#pragma unroll
            for (int c = 0; c < CV; ++c) acc[c] = acc[c] * a1 + o[c] * a2;
            mrun = mn;
        }
    }
    if (ok) {
        const float inv = 1.0f / lsum;
        float* __restrict__ op = O + (win * A.ML + i) * CV;
#pragma unroll
        for (int c = 0; c < CV; ++c) op[c] = acc[c] * inv;
        LSE[win * A.ML + i] = mrun + __logf(lsum);
    }
}

// backward A: lane = query row.  dQ, delta = rowsum(dO*O), d(bias table).  Same (unit, key split) decomposition as the forward;
// the d(bias) of a unit is accumulated in ONE LDS table shared by its S waves (ds_add_f32), then flushed with float atomics.
#define VX_PRIV_BINS 256
#define VX_DTABLE_REPLICAS 16
template <int CQ, int CV>
__device__ __forceinline__ void vx_attn_bwd_q_body(const int bid, const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                   const float* __restrict__ table, const float* __restrict__ O, const float* __restrict__ LSE,
                                                   const float* __restrict__ dO, float* __restrict__ dQ, float* __restrict__ Delta,
                                                   float* __restrict__ dtable_rep, int Tsz, const VxAttn& A, const VxDrop& drop, int S, int one_head, const unsigned short* __restrict__ mbits = nullptr) {
    constexpr int RS = CQ + CV;
    // lin | bias tables | bias-gradient tables | [4] K/V slabs.  one_head (every unit of a block belongs to one head: launcher): ONE bias column and ONE
    // bias-gradient table shared by the block's waves (they reach it through ds_add_f32 only); else all heads and a table per unit.
    extern __shared__ __attribute__((aligned(16))) float vx_sm[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_sm);
    const int lin_pad = (A.l + 3) & ~3;
    float* __restrict__ bias_all = vx_sm + lin_pad;
    float* __restrict__ gtabs = bias_all + (((long)Tsz * (one_head ? 1 : A.heads) + 3) & ~3);
    const int upb = 4 / S;
    const int ntab = one_head ? 1 : upb;
    float* __restrict__ slabs = gtabs + (((long)ntab * Tsz + 3) & ~3);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int split = wave % S;
    float* __restrict__ stab = gtabs + (one_head ? 0L : (long)(wave / S) * Tsz);
    float* __restrict__ slab = slabs + (long)wave * VX_KV_ROWS * RS;
    const int chunks = (A.ML + 63) / 64;
    const long units = (long)A.BH * A.Nt * chunks;
    const int a_blk = (int)((((long)bid * upb / chunks) / A.Nt) % A.heads);      // head of the block's first unit
    vx_attn_tables(A, table, Tsz, lin, bias_all, one_head ? a_blk : -1);
    for (int k = threadIdx.x; k < ntab * Tsz; k += 256) gtabs[k] = 0.0f;
    __syncthreads();
    const VxDropCtx dc = vx_attn_ctx(drop);
    const bool al4 = (A.ML & 3) == 0;
    const long u_raw = (long)bid * upb + wave / S;
    const bool active = u_raw < units;
    const long u = active ? u_raw : units - 1;
    const int chunk = (int)(u % chunks);
    const long win = u / chunks;
    const int a = (int)((win / A.Nt) % A.heads);
    const float* __restrict__ bias = bias_all + (one_head ? 0L : (long)a * Tsz);
    const int i = chunk * 64 + lane;
    const bool ok = active && i < A.ML;
    const int iq = (i < A.ML) ? i : A.ML - 1;
    const long row = win * A.ML + iq;
    float q[CQ], dq[CQ], dov[CV];
#pragma unroll
    for (int c = 0; c < CQ; ++c) { q[c] = Q[row * CQ + c] * A.scale; dq[c] = 0.0f; }
    float delta = 0.0f;
#pragma unroll
    for (int c = 0; c < CV; ++c) { dov[c] = dO[row * CV + c]; delta = fmaf(dov[c], O[row * CV + c], delta); }
    const float lse = LSE[row];
    const int lin_i = lin[iq % A.l] + A.lin_cst;
    const float* __restrict__ kp = K + win * A.ML * CQ;
    const float* __restrict__ vp = Vt + win * A.ML * CV;
    const uint64_t drow = (uint64_t)row * (uint64_t)A.ML;
    // keep bits written by the forward (vx_pwa_attn_fwd_mb) instead of re-drawing the Philox words: one 2-byte load per 4 pairs against a quarter of a
    // Philox4x32-7 call (~25 issue slots per pair).  Aligned windows only (l % 4 == 0): a quad of keys then sits inside one 16-key word
    const bool use_bits = mbits != nullptr && dc.on && al4 && ((A.l & 3) == 0);
    const unsigned short* __restrict__ mbq = use_bits ? mbits + win * ((A.ML + 15) >> 4) * A.ML + iq : nullptr;
    // Keys are walked TOKEN-major: a slab holds TS tokens x M modalities (row m*TS + tt <-> key m*l + t0 + tt).  The M keys of a token share
    // the relative-position bin, so their ds are summed in registers and ONE ds_add_f32 per token reaches the bias-gradient table instead
    // of one per key (LDS float atomics retire ~1 lane per clock: they were 70 % of this kernel).
    const int TS = A.M == 1 ? 64 : A.M == 2 ? 32 : A.M == 4 ? 16 : (64 / A.M) & ~3;
    // Bias-gradient bins of one slab: the wave's 64 queries x TS tokens touch the bin range [li_min - lin(t_last), li_max - lin(t0)].  When it fits
    // VX_PRIV_BINS entries and the wave's rows are distinct positions of ONE modality (=> distinct bins per instruction), the sums go through a
    // wave-private LDS window with plain read-add-write (LDS executes a wave's instructions in order) and only the window is flushed with
    // ds_add_f32 -- float LDS atomics with 64 scattered addresses were 40-65 % of this kernel.
    volatile float* __restrict__ priv = slabs + (long)4 * VX_KV_ROWS * RS + wave * VX_PRIV_BINS;
    int li_min = ok ? lin_i : 0x7fffffff, li_max = ok ? lin_i : -0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { li_min = min(li_min, __shfl_xor(li_min, o, 64)); li_max = max(li_max, __shfl_xor(li_max, o, 64)); }
    const int row_first = chunk * 64, row_last = min(chunk * 64 + 63, A.ML - 1);
    const bool one_modality = active && (row_first / A.l == row_last / A.l);
    for (int k = lane; k < VX_PRIV_BINS; k += 64) priv[k] = 0.0f;
    for (int t0 = split * TS; t0 < A.l; t0 += S * TS) {
        const int nt = min(TS, A.l - t0);
        const int base = li_min - lin[t0 + nt - 1];
        const int range = li_max - li_min + lin[t0 + nt - 1] - lin[t0] + 1;
        const bool use_priv = one_modality && range <= VX_PRIV_BINS;
        __builtin_amdgcn_wave_barrier();
        {
            const int mrow = lane / TS, tt = lane - mrow * TS;
            if (mrow < A.M && tt < nt) {
                const long j = (long)mrow * A.l + t0 + tt;
#pragma unroll
                for (int c = 0; c < CQ; ++c) slab[lane * RS + c] = kp[j * CQ + c];
#pragma unroll
                for (int c = 0; c < CV; ++c) slab[lane * RS + CQ + c] = vp[j * CV + c];
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int tt = 0; tt < nt; tt += 4) {
            float dsum[4] = {0.f, 0.f, 0.f, 0.f};
            int bi[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) bi[t] = lin_i - lin[min(t0 + tt + t, A.l - 1)];
            for (int mk = 0; mk < A.M; ++mk) {
                float m4[4];
                const uint64_t kidx = drow + (uint64_t)mk * A.l + t0 + tt;
                if (use_bits) {        // the forward's keep bits: word (key tile, query), bit = key & 15; the 4 keys of an aligned quad share a word
                    const int jkey = mk * A.l + t0 + tt;
                    const unsigned w = (unsigned)mbq[(long)(jkey >> 4) * A.ML] >> (jkey & 15);
#pragma unroll
                    for (int t = 0; t < 4; ++t) m4[t] = ((w >> t) & 1u) ? dc.inv_keep : 0.0f;
                } else
                vx_attn_drop4(dc, kidx, al4 && ((A.l & 3) == 0), m4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (tt + t < nt) {
                        const float* __restrict__ kr = slab + (mk * TS + tt + t) * RS;
                        float s = 0.0f;
#pragma unroll
                        for (int c = 0; c < CQ; ++c) s = fmaf(q[c], kr[c], s);
                        s += bias[bi[t]];
                        const float p = __expf(s - lse);
                        float dp = 0.0f;
#pragma unroll
                        for (int c = 0; c < CV; ++c) dp = fmaf(dov[c], kr[CQ + c], dp);
                        dp *= m4[t];
                        const float ds = ok ? p * (dp - delta) : 0.0f;
#pragma unroll
                        for (int c = 0; c < CQ; ++c) dq[c] = fmaf(ds, kr[c], dq[c]);
                        dsum[t] += ds;
                    }
                }
            }
            if (use_priv) {
                if (ok) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (tt + t < nt) { const int k = bi[t] - base; priv[k] = priv[k] + dsum[t]; }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (tt + t < nt) atomicAdd(stab + bi[t], dsum[t]);
            }
        }
        if (use_priv) {
            for (int k = lane; k < range; k += 64) {
                const float g = priv[k];
                if (g != 0.0f) { atomicAdd(stab + base + k, g); priv[k] = 0.0f; }
            }
        }
    }
    if (S > 1) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CQ; ++c) slab[lane * RS + c] = dq[c];
    }
    __syncthreads();
    if (split == 0) {
        for (int s2 = 1; s2 < S; ++s2) {
            const float* __restrict__ o = slab + (long)s2 * VX_KV_ROWS * RS + lane * RS;
#pragma unroll
            for (int c = 0; c < CQ; ++c) dq[c] += o[c];
        }
        if (ok) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) dQ[row * CQ + c] = dq[c] * A.scale;
            Delta[row] = delta;
        }
    }
    // one of VX_DTABLE_REPLICAS copies per block: thousands of blocks adding into the same few hundred addresses serialise in L2
    float* __restrict__ dst = dtable_rep + (long)(bid % VX_DTABLE_REPLICAS) * Tsz * A.heads;
    if (one_head) {    // the block's shared table (complete after the barrier above), all threads
        for (int k = threadIdx.x; k < Tsz; k += 256) {
            const float g = gtabs[k];
            if (g != 0.0f) atomicAdd(dst + (long)k * A.heads + a_blk, g);
        }
    } else if (active) {
        for (int k = split * 64 + lane; k < Tsz; k += 64 * S) {
            const float g = stab[k];
            if (g != 0.0f) atomicAdd(dst + (long)k * A.heads + a, g);
        }
    }
}

template <int CQ, int CV>
__global__ void __launch_bounds__(256) vx_pwa_attn_bwd_q_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                           const float* __restrict__ table, const float* __restrict__ O, const float* __restrict__ LSE,
                                                           const float* __restrict__ dO, float* __restrict__ dQ, float* __restrict__ Delta,
                                                           float* __restrict__ dtable_rep, int Tsz, VxAttn A, VxDrop drop, int S, int one_head, const unsigned short* __restrict__ mbits) {
    vx_attn_bwd_q_body<CQ, CV>((int)blockIdx.x, Q, K, Vt, table, O, LSE, dO, dQ, Delta, dtable_rep, Tsz, A, drop, S, one_head, mbits);
}

// backward B: lane = key row.  dK, dV.  Query-side rows (q, dO, lse, delta) are staged per wave in LDS; the S splits of a unit walk
// interleaved query slabs and are summed through LDS.  Dropout: lane (quad position t) draws the Philox counter of query row i+t
// for its quad's 4 keys; a quad transpose (DPP) hands every lane its own key's word for the 4 rows -> one Philox call per 4 pairs.
// Delta == nullptr: delta = rowsum(dO * O) is recomputed from O while the query slab is staged, and the replicas are NOT folded here (the kernel then
// has no input from the dQ pass and runs in the same launch: vx_pwa_attn_bwd_both_k)
template <int CQ, int CV>
__device__ __forceinline__ void vx_attn_bwd_kv_body(const int bid, const int nbid, const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                    const float* __restrict__ table, const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                    const float* __restrict__ O, const float* __restrict__ dO, float* __restrict__ dK, float* __restrict__ dV,
                                                    const float* __restrict__ dtable_rep, float* __restrict__ dtable,
                                                    int Tsz, const VxAttn& A, const VxDrop& drop, int S, int one_head, const unsigned short* __restrict__ mbits = nullptr) {
    constexpr int RS = CQ + CV + 4;          // q[CQ], dO[CV], lse, delta, pad
    if (Delta) {   // fold the dQ kernel's replicated bias-gradient tables into dtable (it ran before this kernel on the same stream): one owner thread per entry
        for (long k = (long)bid * 256 + threadIdx.x; k < (long)Tsz * A.heads; k += (long)nbid * 256) {
            float g = 0.0f;
#pragma unroll
            for (int r = 0; r < VX_DTABLE_REPLICAS; ++r) g += dtable_rep[(long)r * Tsz * A.heads + k];
            dtable[k] += g;
        }
    }
    extern __shared__ __attribute__((aligned(16))) float vx_sm[];
    int* __restrict__ lin = reinterpret_cast<int*>(vx_sm);
    const int lin_pad = (A.l + 3) & ~3;
    float* __restrict__ bias_all = vx_sm + lin_pad;
    float* __restrict__ slabs = bias_all + (((long)Tsz * (one_head ? 1 : A.heads) + 3) & ~3);
    const int chunks = (A.ML + 63) / 64;
    const long units = (long)A.BH * A.Nt * chunks;
    vx_attn_tables(A, table, Tsz, lin, bias_all, one_head ? (int)((((long)bid * (4 / S) / chunks) / A.Nt) % A.heads) : -1);
    __syncthreads();
    const VxDropCtx dc = vx_attn_ctx(drop);
    const bool al4 = (A.ML & 3) == 0;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int split = wave % S;
    const long u_raw = (long)bid * (4 / S) + wave / S;
    const bool active = u_raw < units;
    const long u = active ? u_raw : units - 1;
    float* __restrict__ slab = slabs + (long)wave * VX_KV_ROWS * RS;
    const int lane = threadIdx.x & 63;
    const int chunk = (int)(u % chunks);
    const long win = u / chunks;
    const int a = (int)((win / A.Nt) % A.heads);
    const float* __restrict__ bias = bias_all + (one_head ? 0L : (long)a * Tsz);
    const int j = chunk * 64 + lane;
    const bool ok = active && j < A.ML;
    const int jk = (j < A.ML) ? j : A.ML - 1;
    const long krow = win * A.ML + jk;
    float k[CQ], dk[CQ], v[CV], dv[CV];
#pragma unroll
    for (int c = 0; c < CQ; ++c) { k[c] = K[krow * CQ + c]; dk[c] = 0.0f; }
#pragma unroll
    for (int c = 0; c < CV; ++c) { v[c] = Vt[krow * CV + c]; dv[c] = 0.0f; }
    const int lin_j = lin[jk % A.l] - A.lin_cst;
    const float* __restrict__ qp = Q + win * A.ML * CQ;
    const float* __restrict__ dop = dO + win * A.ML * CV;
    const int qt = lane & 3;
    const bool use_bits = mbits != nullptr && dc.on && al4 && ((A.l & 3) == 0);
    const unsigned short* __restrict__ mbk = use_bits ? mbits + (win * ((A.ML + 15) >> 4) + (jk >> 4)) * A.ML : nullptr;
    for (int i0 = split * VX_KV_ROWS; i0 < A.ML; i0 += S * VX_KV_ROWS) {
        const int nq = min(VX_KV_ROWS, A.ML - i0);
        int ti = i0 % A.l;
        __builtin_amdgcn_wave_barrier();
        if (lane < nq) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) slab[lane * RS + c] = qp[(long)(i0 + lane) * CQ + c];
#pragma unroll
            for (int c = 0; c < CV; ++c) slab[lane * RS + CQ + c] = dop[(long)(i0 + lane) * CV + c];
            slab[lane * RS + CQ + CV] = LSE[win * A.ML + i0 + lane];
            float dl;
            if (Delta) dl = Delta[win * A.ML + i0 + lane];
            else {
                const float* __restrict__ orow = O + (win * A.ML + i0 + lane) * CV;
                dl = 0.0f;
#pragma unroll
                for (int c = 0; c < CV; ++c) dl = fmaf(dop[(long)(i0 + lane) * CV + c], orow[c], dl);      // same order as the dQ pass: identical bits
            }
            slab[lane * RS + CQ + CV + 1] = dl;
        }
        __builtin_amdgcn_wave_barrier();
        for (int ii = 0; ii < nq; ii += 4) {
            float m4[4];
            if (!dc.on) {
                m4[0] = m4[1] = m4[2] = m4[3] = 1.0f;
            } else if (use_bits) {                     // the forward's keep bits of rows i0+ii .. +3 for this key's tile: four 16-bit words = one 8-byte load
                const uint2 w2 = *reinterpret_cast<const uint2*>(mbk + i0 + ii);
                const unsigned sh = (unsigned)(jk & 15);
                m4[0] = ((w2.x >> sh) & 1u) ? dc.inv_keep : 0.0f;
                m4[1] = ((w2.x >> (16 + sh)) & 1u) ? dc.inv_keep : 0.0f;
                m4[2] = ((w2.y >> sh) & 1u) ? dc.inv_keep : 0.0f;
                m4[3] = ((w2.y >> (16 + sh)) & 1u) ? dc.inv_keep : 0.0f;
            } else if (al4) {                          // ML % 4 == 0: quads are counter-aligned for every row, nq % 4 == 0
                vx_attn_masks_rows4(dc, (uint64_t)(win * A.ML + i0 + ii), 1, A.ML, jk, m4);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) m4[t] = vx_attn_drop1(dc, ((uint64_t)(win * A.ML + min(i0 + ii + t, A.ML - 1))) * (uint64_t)A.ML + jk);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (ii + t < nq) {
                    const float* __restrict__ qr = slab + (ii + t) * RS;
                    float s = 0.0f;
#pragma unroll
                    for (int c = 0; c < CQ; ++c) s = fmaf(qr[c], k[c], s);
                    s = s * A.scale + bias[lin[ti] - lin_j];
                    if (++ti == A.l) ti = 0;
                    const float p = __expf(s - qr[CQ + CV]);
                    const float msk = m4[t];
                    float dp = 0.0f;
#pragma unroll
                    for (int c = 0; c < CV; ++c) dp = fmaf(qr[CQ + c], v[c], dp);
                    const float pd = p * msk;
#pragma unroll
                    for (int c = 0; c < CV; ++c) dv[c] = fmaf(pd, qr[CQ + c], dv[c]);
                    const float ds = p * (dp * msk - qr[CQ + CV + 1]) * A.scale;
#pragma unroll
                    for (int c = 0; c < CQ; ++c) dk[c] = fmaf(ds, qr[c], dk[c]);
                }
            }
        }
    }
    if (S > 1) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < CQ; ++c) slab[lane * RS + c] = dk[c];
#pragma unroll
        for (int c = 0; c < CV; ++c) slab[lane * RS + CQ + c] = dv[c];
        __syncthreads();
        if (split != 0) return;
        for (int s2 = 1; s2 < S; ++s2) {
            const float* __restrict__ o = slab + (long)s2 * VX_KV_ROWS * RS + lane * RS;
#pragma unroll
            for (int c = 0; c < CQ; ++c) dk[c] += o[c];
#pragma unroll
            for (int c = 0; c < CV; ++c) dv[c] += o[c + CQ];
        }
    }
    if (ok) {
#pragma unroll
        for (int c = 0; c < CQ; ++c) dK[krow * CQ + c] = dk[c];
#pragma unroll
        for (int c = 0; c < CV; ++c) dV[krow * CV + c] = dv[c];
    }
}

template <int CQ, int CV>
__global__ void __launch_bounds__(256) vx_pwa_attn_bwd_kv_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                            const float* __restrict__ table, const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                            const float* __restrict__ dO, float* __restrict__ dK, float* __restrict__ dV,
                                                            const float* __restrict__ dtable_rep, float* __restrict__ dtable,
                                                            int Tsz, VxAttn A, VxDrop drop, int S, int one_head, const unsigned short* __restrict__ mbits) {
    vx_attn_bwd_kv_body<CQ, CV>((int)blockIdx.x, (int)gridDim.x, Q, K, Vt, table, LSE, Delta, nullptr, dO, dK, dV, dtable_rep, dtable, Tsz, A, drop, S, one_head, mbits);
}
// Both passes in ONE launch: even blocks run the dQ pass, odd blocks the dK / dV pass of the same block index.  Each pass alone leaves the SIMDs waiting
// (dQ: 53 % of its wave cycles in s_waitcnt, dK/dV: 31 % + 20 % stalled; profiles/r02_sq_wave_breakdown.txt); interleaved on the same CUs they fill each
// other's gaps, and the stream loses one launch boundary.  The dK/dV blocks recompute delta, and the replica fold moves to vx_attn_fold_k.
template <int CQ, int CV>
__global__ void __launch_bounds__(256) vx_pwa_attn_bwd_both_k(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                                                              const float* __restrict__ table, const float* __restrict__ O, const float* __restrict__ LSE,
                                                              const float* __restrict__ dO, float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
                                                              float* __restrict__ Delta, float* __restrict__ dtable_rep, int Tsz, VxAttn A, VxDrop drop, int S, int one_head,
                                                              const unsigned short* __restrict__ mbits) {
    const int bid = (int)(blockIdx.x >> 1);
    if (blockIdx.x & 1) vx_attn_bwd_kv_body<CQ, CV>(bid, (int)(gridDim.x >> 1), Q, K, Vt, table, LSE, nullptr, O, dO, dK, dV, nullptr, nullptr, Tsz, A, drop, S, one_head, mbits);
    else vx_attn_bwd_q_body<CQ, CV>(bid, Q, K, Vt, table, O, LSE, dO, dQ, Delta, dtable_rep, Tsz, A, drop, S, one_head, mbits);
}
// dtable[k] += sum over the replicas (vx_pwa_attn_bwd_both_k's dQ blocks filled them)
__global__ void __launch_bounds__(256) vx_attn_fold_k(const float* __restrict__ rep, float* __restrict__ dtable, long n, int nrep) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float g = 0.0f;
    for (int r = 0; r < nrep; ++r) g += rep[(long)r * n + k];
    dtable[k] += g;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline int plane_l(const VxPwaPlan* P) { return P->l; }
static int vx_plan_check(const VxPwaPlan* P, const char* who) {
    if (!P) VX_FAIL(-1, "%s: null plan", who);
    if (P->nb < 1 || P->nb > 4 || P->heads < 1 || P->l != P->n[0] * P->n[1] * P->n[2]) VX_FAIL(-1, "%s: bad plan (nb=%d heads=%d l=%d)", who, P->nb, P->heads, P->l);
    for (int i = 0; i < P->nb; ++i)
        for (int k = 0; k < 3; ++k)
            if (P->small[i][k] < 1 || P->nwin[i][k] < 1 || P->nwin[i][k] * P->n[k] * P->small[i][k] != P->grid[k])
                VX_FAIL(-1, "%s: scale %d does not tile the grid on axis %d", who, i, k);
    return 0;
}

static int vx_gather_vec_enabled = 1;
extern "C" int vx_pwa_gather_set_vec(int on) { vx_gather_vec_enabled = on ? 1 : 0; return 0; }      // A/B knob (tests): 0 = one lane per (cell / voxel, channel) in the gather and the scatter forward
extern "C" int vx_pwa_gather_fwd(const float* src, float* tok, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_gather_fwd")) return e;
    VX_REQUIRE(src && tok && c > 0 && m >= 0 && m < M && B > 0, "vx_pwa_gather_fwd: bad args");
    const long V = (long)plan->grid[0] * plan->grid[1] * plan->grid[2];
    hipLaunchKernelGGL(vx_pwa_gather_fwd_k, dim3(vx_cdiv(V, 256), plan->nb * plan->heads * c, B), dim3(256), 0, (hipStream_t)stream, src, tok, *plan, c, m, M);
    VX_LAUNCH_CHECK("vx_pwa_gather_fwd");
    return 0;
}

extern "C" int vx_pwa_gather_bwd(const float* src, const float* dtok, float* dsrc, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_gather_bwd")) return e;
    VX_REQUIRE(src && dtok && dsrc && c > 0 && m >= 0 && m < M && B > 0, "vx_pwa_gather_bwd: bad args");
    const long V = (long)plan->grid[0] * plan->grid[1] * plan->grid[2];
    hipLaunchKernelGGL(vx_pwa_gather_bwd_k, dim3(vx_cdiv(V, 256), plan->nb * plan->heads * c, B), dim3(256), 0, (hipStream_t)stream, src, dtok, dsrc, *plan, c, m, M);
    VX_LAUNCH_CHECK("vx_pwa_gather_bwd");
    return 0;
}

extern "C" int vx_pwa_gather_all_fwd(const float* const* srcs, float* tq, float* tk, float* tv, int* iq, int* ik, int* iv,
                                     const VxPwaPlan* plan, int cq, int cv, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_gather_all_fwd")) return e;
    VX_REQUIRE(srcs && tq && tk && tv && iq && ik && iv && cq > 0 && cv > 0 && M >= 1 && M <= 4 && B > 0, "vx_pwa_gather_all_fwd: bad args");
    VxGatherPtrs ptrs;
    for (int k = 0; k < 3 * M; ++k) { VX_REQUIRE(srcs[k], "vx_pwa_gather_all_fwd: null source %d", k); ptrs.src[k] = srcs[k]; ptrs.dsrc[k] = nullptr; }
    const long V = (long)plan->grid[0] * plan->grid[1] * plan->grid[2];
    const int per_m = plan->nb * plan->heads * (2 * cq + cv);
    if (cq % 4 == 0 && cv % 4 == 0 && vx_gather_vec_enabled) {
        const int CH = (cq % 8 == 0 && cv % 8 == 0) ? 8 : 4;
        const int nchm = (cq > cv ? cq : cv) / CH;
        const dim3 g(vx_cdiv(V, 256), M * 3 * plan->nb * plan->heads * nchm, B);
        if (CH == 8) hipLaunchKernelGGL(vx_pwa_gather_all_fwd_v_k<8>, g, dim3(256), 0, (hipStream_t)stream, ptrs, tq, tk, tv, iq, ik, iv, *plan, cq, cv, M);
        else hipLaunchKernelGGL(vx_pwa_gather_all_fwd_v_k<4>, g, dim3(256), 0, (hipStream_t)stream, ptrs, tq, tk, tv, iq, ik, iv, *plan, cq, cv, M);
        VX_LAUNCH_CHECK("vx_pwa_gather_all_fwd");
        return 0;
    }
    // grid.x covers the finest scale (one lane per cell, 256 cells per block); coarser scales use fewer cells but T lanes each
    hipLaunchKernelGGL(vx_pwa_gather_all_fwd_k, dim3(vx_cdiv(V, 256), per_m * M, B), dim3(256), 0, (hipStream_t)stream, ptrs, tq, tk, tv, iq, ik, iv, *plan, cq, cv, M);
    VX_LAUNCH_CHECK("vx_pwa_gather_all_fwd");
    return 0;
}

extern "C" int vx_pwa_gather_all_bwd(const float* dtq, const float* dtk, const float* dtv, const int* iq, const int* ik, const int* iv, float* const* dsrcs,
                                     const VxPwaPlan* plan, int cq, int cv, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_gather_all_bwd")) return e;
    VX_REQUIRE(dsrcs && dtq && dtk && dtv && iq && ik && iv && cq > 0 && cv > 0 && M >= 1 && M <= 4 && B > 0, "vx_pwa_gather_all_bwd: bad args");
    VxGatherPtrs ptrs;
    for (int k = 0; k < 3 * M; ++k) { VX_REQUIRE(dsrcs[k], "vx_pwa_gather_all_bwd: null destination %d", k); ptrs.dsrc[k] = dsrcs[k]; ptrs.src[k] = nullptr; }
    const long V = (long)plan->grid[0] * plan->grid[1] * plan->grid[2];
    const int per_m = plan->nb * plan->heads * (2 * cq + cv);
    if (cq % 4 == 0 && cv % 4 == 0 && vx_gather_vec_enabled) {
        const int CH = (cq % 8 == 0 && cv % 8 == 0) ? 8 : 4;
        const int nchm = (cq > cv ? cq : cv) / CH;
        const dim3 g(vx_cdiv(V, 256), M * 3 * plan->nb * plan->heads * nchm, B);
        if (CH == 8) hipLaunchKernelGGL(vx_pwa_gather_all_bwd_v_k<8>, g, dim3(256), 0, (hipStream_t)stream, ptrs, dtq, dtk, dtv, iq, ik, iv, *plan, cq, cv, M);
        else hipLaunchKernelGGL(vx_pwa_gather_all_bwd_v_k<4>, g, dim3(256), 0, (hipStream_t)stream, ptrs, dtq, dtk, dtv, iq, ik, iv, *plan, cq, cv, M);
        VX_LAUNCH_CHECK("vx_pwa_gather_all_bwd");
        return 0;
    }
    hipLaunchKernelGGL(vx_pwa_gather_all_bwd_k, dim3(vx_cdiv(V, 256), per_m * M, B), dim3(256), 0, (hipStream_t)stream, ptrs, dtq, dtk, dtv, iq, ik, iv, *plan, cq, cv, M);
    VX_LAUNCH_CHECK("vx_pwa_gather_all_bwd");
    return 0;
}

static int vx_scatter_fwd_launch(const float* tok, const VxScPtrs& ptrs, const VxPwaPlan* plan, int c, int m0, int mcount, int M, int B, void* stream) {
    const long V = (long)plan->grid[0] * plan->grid[1] * plan->grid[2];
    const int nx = vx_cdiv(V, 256);
    if ((c & 3) == 0 && vx_gather_vec_enabled)
        hipLaunchKernelGGL(vx_pwa_scatter_fwd_v_k, dim3(nx * mcount, plan->nb * plan->heads * (c >> 2), B), dim3(256), 0, (hipStream_t)stream, tok, ptrs, *plan, c, m0, M, nx);
    else
        hipLaunchKernelGGL(vx_pwa_scatter_fwd_k, dim3(nx * mcount, plan->nb * plan->heads * c, B), dim3(256), 0, (hipStream_t)stream, tok, ptrs, *plan, c, m0, M, nx);
    VX_LAUNCH_CHECK("vx_pwa_scatter_fwd");
    return 0;
}
extern "C" int vx_pwa_scatter_fwd(const float* tok, float* out, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_scatter_fwd")) return e;
    VX_REQUIRE(tok && out && c > 0 && m >= 0 && m < M && B > 0, "vx_pwa_scatter_fwd: bad args");
    VxScPtrs ptrs = {};
    ptrs.out[0] = out;
    return vx_scatter_fwd_launch(tok, ptrs, plan, c, m, 1, M, B, stream);
}
// every modality in one launch per kernel kind (outs[m], m < M <= 4)
extern "C" int vx_pwa_scatter_fwd_all(const float* tok, float* const* outs, const VxPwaPlan* plan, int c, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_scatter_fwd_all")) return e;
    VX_REQUIRE(tok && outs && c > 0 && M >= 1 && M <= 4 && B > 0, "vx_pwa_scatter_fwd_all: bad args");
    VxScPtrs ptrs = {};
    for (int m = 0; m < M; ++m) { VX_REQUIRE(outs[m], "vx_pwa_scatter_fwd_all: null output %d", m); ptrs.out[m] = outs[m]; }
    return vx_scatter_fwd_launch(tok, ptrs, plan, c, 0, M, M, B, stream);
}

static int vx_scatter_ident_enabled = 1;
static const int vx_scatter_gather_max = 2;      // cells up to 2^3: the per-token gather; wider cells: the separable kernel (4^3 cells measured 36 -> 17 us per launch)
extern "C" int vx_pwa_scatter_set_ident(int on) { vx_scatter_ident_enabled = on ? 1 : 0; return 0; }      // A/B knob: 0 = always the general (LDS-atomic) adjoint
// which adjoint kernel takes scale i: 0 identity (sole owner), 1 per-token gather (sole owner), 2 separable (atomics), 3 general (atomics)
static int vx_scatter_bwd_kind(const VxPwaPlan* plan, int c, int B, int i) {
    if (plan->small[i][0] == 1 && plan->small[i][1] == 1 && plan->small[i][2] == 1 && vx_scatter_ident_enabled) return 0;
    const long nv = (long)plan->n[0] * plan->small[i][0] * plan->n[1] * plan->small[i][1] * plan->n[2] * plan->small[i][2];
    const int smax = plan->small[i][0] > plan->small[i][1] ? (plan->small[i][0] > plan->small[i][2] ? plan->small[i][0] : plan->small[i][2])
                                                            : (plan->small[i][1] > plan->small[i][2] ? plan->small[i][1] : plan->small[i][2]);
    const int nmax = plan->n[0] > plan->n[1] ? (plan->n[0] > plan->n[2] ? plan->n[0] : plan->n[2]) : (plan->n[1] > plan->n[2] ? plan->n[1] : plan->n[2]);
    const size_t shm2 = sizeof(float) * (((size_t)nv + 3) / 4 * 4 + (size_t)3 * nmax * VX_SC_TAPS * 2 + 3 * nmax);
    if (smax <= vx_scatter_gather_max && shm2 <= 64 * 1024 && c <= 65535 && vx_scatter_ident_enabled) return 1;
    const int bw0 = plan->n[0] * plan->small[i][0], bw1 = plan->n[1] * plan->small[i][1], bw2 = plan->n[2] * plan->small[i][2];
    (void)bw0;
    const size_t shm3 = sizeof(float) * ((size_t)bw1 * bw2 + (size_t)bw1 * plan->n[2] + (size_t)plane_l(plan) + (size_t)plan->n[2] * bw2 + (size_t)plan->n[1] * bw1);
    if (vx_scatter_ident_enabled && shm3 <= 60 * 1024 && (long)B * plan->heads * c <= 65535) return 2;
    return 3;
}
// writes_all: the caller did NOT zero dtok (needs every modality in this call and scale 0 on the identity kernel -- checked by the entry)
static int vx_scatter_bwd_launch(const VxScPtrs& ptrs, float* dtok, const VxPwaPlan* plan, int c, int m0, int mcount, int M, int B, void* stream, bool writes_all = false,
                                 float* extra_zero = nullptr, long extra_floats = 0) {
    VxScZero z = {};
    if (writes_all) {
        z.assign = 1;
        z.extra = extra_zero; z.extra_n4 = extra_zero ? extra_floats / 4 : 0;
        for (int i = 1; i < plan->nb; ++i)
            if (vx_scatter_bwd_kind(plan, c, B, i) >= 2) { z.w0[z.n] = plan->woff[i]; z.wn[z.n] = plan->nwin[i][0] * plan->nwin[i][1] * plan->nwin[i][2]; ++z.n; }
    }
    const size_t shm = sizeof(float) * (size_t)plane_l(plan) * c;
    VX_REQUIRE(shm <= 128 * 1024, "vx_pwa_scatter_bwd: window (%d tokens x %d) does not fit LDS", plan->l, c);
    for (int i = 0; i < plan->nb; ++i) {
        const long nv = (long)plan->n[0] * plan->small[i][0] * plan->n[1] * plan->small[i][1] * plan->n[2] * plan->small[i][2];
        int chunks = (int)((nv * c + 4095) / 4096);
        if (chunks > 64) chunks = 64;
        const int nwin = plan->nwin[i][0] * plan->nwin[i][1] * plan->nwin[i][2];
        if (plan->small[i][0] == 1 && plan->small[i][1] == 1 && plan->small[i][2] == 1 && vx_scatter_ident_enabled) {
            const size_t shm1 = sizeof(float) * (size_t)plane_l(plan) * (c + 1);
            VX_REQUIRE(shm1 <= 128 * 1024, "vx_pwa_scatter_bwd: window (%d tokens x %d) does not fit LDS", plan->l, c);
            hipLaunchKernelGGL(vx_pwa_scatter_bwd_ident_k, dim3(nwin * mcount, B * plan->heads), dim3(256), shm1, (hipStream_t)stream, ptrs, dtok, *plan, c, m0, M, i, nwin, z);
            continue;
        }
        {
            const int smax = plan->small[i][0] > plan->small[i][1] ? (plan->small[i][0] > plan->small[i][2] ? plan->small[i][0] : plan->small[i][2])
                                                                    : (plan->small[i][1] > plan->small[i][2] ? plan->small[i][1] : plan->small[i][2]);
            const int nmax = plan->n[0] > plan->n[1] ? (plan->n[0] > plan->n[2] ? plan->n[0] : plan->n[2]) : (plan->n[1] > plan->n[2] ? plan->n[1] : plan->n[2]);
            const size_t shm2 = sizeof(float) * (((size_t)nv + 3) / 4 * 4 + (size_t)3 * nmax * VX_SC_TAPS * 2 + 3 * nmax);
            if (smax <= vx_scatter_gather_max && shm2 <= 64 * 1024 && c <= 65535 && vx_scatter_ident_enabled) {      // <= 4x4x4: at most 5 + 4 taps per token and axis (VX_SC_TAPS = 12)
                int PT = 256 / plan->l;                       // threads per token: power of two in 1..4
                PT = PT >= 4 ? 4 : (PT >= 2 ? 2 : 1);
                hipLaunchKernelGGL(vx_pwa_scatter_bwd_gather_k, dim3(nwin * mcount, B * plan->heads, c), dim3(256), shm2, (hipStream_t)stream, ptrs, dtok, *plan, c, m0, M, i, PT, nwin, z.assign);
                continue;
            }
        }
        {
            // cells wider than 4: separable adjoint (plane in LDS, W then H then D), ZC planes per block so that ~256+ blocks exist
            const int bw0 = plan->n[0] * plan->small[i][0], bw1 = plan->n[1] * plan->small[i][1], bw2 = plan->n[2] * plan->small[i][2];
            const size_t shm3 = sizeof(float) * ((size_t)bw1 * bw2 + (size_t)bw1 * plan->n[2] + (size_t)plane_l(plan) + (size_t)plan->n[2] * bw2 + (size_t)plan->n[1] * bw1);
            if (vx_scatter_ident_enabled && shm3 <= 60 * 1024 && (long)B * plan->heads * c <= 65535) {
                int ZC = bw0;
                while (ZC > 1 && (long)nwin * B * plan->heads * c * vx_cdiv(bw0, ZC) < 256) ZC = (ZC + 1) / 2;
                hipLaunchKernelGGL(vx_pwa_scatter_bwd_sep_k, dim3(nwin * mcount, B * plan->heads, c * vx_cdiv(bw0, ZC)), dim3(256), shm3, (hipStream_t)stream, ptrs, dtok, *plan, c, m0, M, i, ZC, nwin);
                continue;
            }
        }
        hipLaunchKernelGGL(vx_pwa_scatter_bwd_k, dim3(chunks * mcount, nwin, B * plan->heads), dim3(256), shm, (hipStream_t)stream, ptrs, dtok, *plan, c, m0, M, i, chunks);
    }
    VX_LAUNCH_CHECK("vx_pwa_scatter_bwd");
    return 0;
}

extern "C" int vx_pwa_scatter_bwd(const float* dout, float* dtok, const VxPwaPlan* plan, int c, int m, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_scatter_bwd")) return e;
    VX_REQUIRE(dout && dtok && c > 0 && m >= 0 && m < M && B > 0, "vx_pwa_scatter_bwd: bad args");
    VxScPtrs ptrs = {};
    ptrs.in[0] = dout;
    return vx_scatter_bwd_launch(ptrs, dtok, plan, c, m, 1, M, B, stream);
}
// the same WITHOUT a zeroed destination (round 6: the fill launch in front of every PWA backward was 6 us on the encoder backward's chain): the sole-owner kernels assign,
// the identity-scale kernel -- first on the stream -- zeroes the window ranges of the scales that add with atomics.  Returns 1 (nothing launched) where that does not
// apply (scale 0 not on the identity kernel, more than 4 atomic scales): the caller zeroes dtok and calls vx_pwa_scatter_bwd_all.
extern "C" int vx_pwa_scatter_bwd_all_wz(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, float* extra_zero, long extra_floats, void* stream);
extern "C" int vx_pwa_scatter_bwd_all_w(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, void* stream) {
    return vx_pwa_scatter_bwd_all_wz(douts, dtok, plan, c, M, B, nullptr, 0, stream);
}
// ... and one more buffer zeroed by the same launch (extra_floats % 4 == 0, 16-byte aligned): the bias-gradient replicas of the attention backward that follows
// (vx_pwa_attn_bwd_rep_offset; vx_pwa_attn_bwd_mark_rep_zeroed tells that call to skip its own zeroing launch)
extern "C" int vx_pwa_scatter_bwd_all_wz(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, float* extra_zero, long extra_floats, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_scatter_bwd_all_w")) return e;
    VX_REQUIRE(douts && dtok && c > 0 && M >= 1 && M <= 4 && B > 0, "vx_pwa_scatter_bwd_all_w: bad args");
    VX_REQUIRE(extra_zero == nullptr || (extra_floats > 0 && extra_floats % 4 == 0 && ((uintptr_t)extra_zero & 15) == 0), "vx_pwa_scatter_bwd_all_wz: the extra buffer must be 16-byte aligned, a multiple of 4 floats");
    static const int off = getenv("VELOXSEG_SCATTER_BWD_W") && getenv("VELOXSEG_SCATTER_BWD_W")[0] == '0';      // (A/B)
    if (off || plan->nb > 4 || vx_scatter_bwd_kind(plan, c, B, 0) != 0) return 1;
    for (int i = 1; i < plan->nb; ++i) if (vx_scatter_bwd_kind(plan, c, B, i) == 0) return 1;      // (a second identity scale would zero nothing but is not expected: keep the old path)
    VxScPtrs ptrs = {};
    for (int m = 0; m < M; ++m) { VX_REQUIRE(douts[m], "vx_pwa_scatter_bwd_all_w: null gradient %d", m); ptrs.in[m] = douts[m]; }
    return vx_scatter_bwd_launch(ptrs, dtok, plan, c, 0, M, M, B, stream, true, extra_zero, extra_floats);
}
// every modality in one launch per scale (douts[m] may not be NULL; dtok zeroed by the caller)
extern "C" int vx_pwa_scatter_bwd_all(const float* const* douts, float* dtok, const VxPwaPlan* plan, int c, int M, int B, void* stream) {
    if (int e = vx_plan_check(plan, "vx_pwa_scatter_bwd_all")) return e;
    VX_REQUIRE(douts && dtok && c > 0 && M >= 1 && M <= 4 && B > 0, "vx_pwa_scatter_bwd_all: bad args");
    VxScPtrs ptrs = {};
    for (int m = 0; m < M; ++m) { VX_REQUIRE(douts[m], "vx_pwa_scatter_bwd_all: null gradient %d", m); ptrs.in[m] = douts[m]; }
    return vx_scatter_bwd_launch(ptrs, dtok, plan, c, 0, M, M, B, stream);
}

static int vx_attn_fill(VxAttn& A, const VxPwaPlan* P, int B, int M, int cq, int cv, const char* who) {
    if (int e = vx_plan_check(P, who)) return e;
    if (B <= 0 || M <= 0 || M > 16) VX_FAIL(-1, "%s: bad B/M (1..16 modalities)", who);
    A.BH = B * P->heads; A.heads = P->heads; A.Nt = P->Ntot; A.l = P->l; A.M = M; A.ML = M * P->l;
    A.n[0] = P->n[0]; A.n[1] = P->n[1]; A.n[2] = P->n[2];
    A.cq = cq; A.cv = cv; A.scale = 1.0f / sqrtf((float)cq);
    A.lin_cst = ((P->n[0] - 1) * (2 * P->n[1] - 1) + (P->n[1] - 1)) * (2 * P->n[2] - 1) + (P->n[2] - 1);
    return 0;
}

// key/query split S of the attention kernels: enough waves to fill 256 CUs x 4 SIMDs several times over, never more splits than slabs
static int vx_attn_bwd_fused = 1;
extern "C" int vx_pwa_attn_set_fused_bwd(int on) { vx_attn_bwd_fused = on ? 1 : 0; return 0; }      // A/B: dQ and dK/dV passes in one launch (default) or two
static int vx_attn_split_override = 0;
extern "C" int vx_pwa_attn_set_split(int S) {
    if (S != 0 && S != 1 && S != 2 && S != 4) VX_FAIL(-1, "vx_pwa_attn_set_split: S must be 0 (auto), 1, 2 or 4");
    vx_attn_split_override = S;
    return 0;
}
static int vx_attn_split(long units, int ML, int M) {
    const int TS = M == 1 ? 64 : M == 2 ? 32 : M == 4 ? 16 : (64 / M) & ~3;
    const int l = ML / M;
    const int nslabs_q = (l + TS - 1) / TS;                     // token-major slabs of the dQ kernel
    const int nslabs = ((ML + 63) / 64) < nslabs_q ? (ML + 63) / 64 : nslabs_q;
    int S = 1;
    if (vx_attn_split_override) S = vx_attn_split_override;
    else if (units * 2 < 4096) S = 4;
    else if (units < 4096) S = 2;
    while (S > nslabs) S >>= 1;
    return S;
}

template <int A_, int B_> struct vx_pair { static constexpr int a = A_, b = B_; };
// returns false when the (c_qk, c_v) pair has no instantiation
template <class F> static bool vx_attn_dispatch(int cq, int cv, F&& f) {
#define VX_CASE(X, Y) if (cq == X && cv == Y) { f(vx_pair<X, Y>{}); return true; }
    VX_CASE(4, 4) VX_CASE(8, 8) VX_CASE(8, 16) VX_CASE(16, 16) VX_CASE(16, 32) VX_CASE(4, 8) VX_CASE(2, 2) VX_CASE(2, 4)
    VX_CASE(4, 16) VX_CASE(8, 32) VX_CASE(2, 8) VX_CASE(16, 64) VX_CASE(8, 64) VX_CASE(32, 32)
#undef VX_CASE
    return false;
}

// MFMA kernels (pwa_mfma.hip) for windows whose tokens tile into 16-token blocks
int vx_pwa_attn_mfma_fwd(const float* Q, const float* K, const float* V, const float* table, float* O, float* LSE, unsigned short* mbits, const VxPwaPlan* plan, int B, int M,
                         int cq, int cv, VxDrop d, void* stream);
extern "C" int vx_pwa_attn_bwd1_ok(const VxPwaPlan* plan, int B, int M, int cq, int cv);
int vx_pwa_attn_bwd1(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE, const float* dO, float* dQ,
                     float* dK, float* dV, float* rep, const unsigned short* mbits, const VxPwaPlan* plan, int B, int M, int cq, int cv, VxDrop d, void* stream);
int vx_pwa_attn_mfma_bwd(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE, const float* dO, float* dQ,
                         float* dK, float* dV, float* dtable, float* delta, float* rep, const VxPwaPlan* plan, int B, int M, int cq, int cv, VxDrop d,
                         void* stream);
extern "C" int vx_pwa_attn_bwd1h_ok(const VxPwaPlan* plan, int B, int M, int cq, int cv);
int vx_pwa_attn_bwd1h(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE, const float* dO, float* dQ,
                      float* dK, float* dV, float* rep, const unsigned short* mbits, const VxPwaPlan* plan, int B, int M, int cq, int cv, VxDrop d, void* stream);

// OFF by default: the VALU backward is bound by the latency of its LDS slab reads (53 % / 31 % of the wave cycles in s_waitcnt, profiles/r02_sq_wave_breakdown.txt),
// not by its VALU work -- with the bits read back the step is unchanged (746-749 vs 748-751 patches/s): the path stays as a tested A/B variant
static int vx_attn_valu_bits = 0;
extern "C" int vx_pwa_attn_set_valu_bits(int on) { vx_attn_valu_bits = on ? 1 : 0; return 0; }      // A/B (tests, probes): the VALU backward reads the forward's keep bits
extern "C" int vx_pwa_attn_mbits_words(const VxPwaPlan* plan, int B, int M) {
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, 4, 4, "vx_pwa_attn_mbits_words")) return e;
    const long n = (long)A.BH * A.Nt * ((A.ML + 15) >> 4) * A.ML;
    VX_REQUIRE(n < 0x7fffffffL, "vx_pwa_attn_mbits_words: too large");
    return (int)n;
}
// 1 when keeping the forward's mask bits pays for this geometry: the one-pass backward is selected AND would otherwise draw one Philox word set per
// ELEMENT (l % 4 != 0: the 4 queries of a lane do not share a counter); with aligned windows re-drawing costs the same as reading the bits back
extern "C" int vx_pwa_attn_mbits_useful(const VxPwaPlan* plan, int B, int M, int cq, int cv) {
    if (plan == nullptr) return 0;
    if (vx_pwa_attn_bwd1h_ok(plan, B, M, cq, cv) == 1) return 1;        // the f16-pipe backward has no Philox path: it reads the forward's bits
    if (vx_pwa_attn_bwd1_ok(plan, B, M, cq, cv) == 1) return (plan->l & 3) != 0 ? 1 : 0;
    // the fp32-VALU backward evaluates the soft-max side of a pair TWICE (dQ pass, dK/dV pass): with aligned windows both passes read the bits back
    if (vx_attn_valu_bits && (plan->l & 3) == 0 && !(vx_pwa_attn_mfma_ok(plan, B, M, cq, cv) & 2)) return 1;
    return 0;
}
extern "C" int vx_pwa_attn_fwd(const float* Q, const float* K, const float* V, const float* table, float* O, float* LSE,
                               const VxPwaPlan* plan, int B, int M, int cq, int cv,
                               const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream) {
    return vx_pwa_attn_fwd_mb(Q, K, V, table, O, LSE, plan, B, M, cq, cv, seed_ptr, dstream, p_drop, nullptr, stream);
}
extern "C" int vx_pwa_attn_fwd_mb(const float* Q, const float* K, const float* V, const float* table, float* O, float* LSE,
                                  const VxPwaPlan* plan, int B, int M, int cq, int cv,
                                  const void* seed_ptr, unsigned long long dstream, float p_drop, void* mbits_, void* stream) {
    unsigned short* mbits = p_drop > 0 ? (unsigned short*)mbits_ : nullptr;
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, cq, cv, "vx_pwa_attn_fwd")) return e;
    VX_REQUIRE(Q && K && V && table && O && LSE, "vx_pwa_attn_fwd: null pointer");
    const long units = (long)A.BH * A.Nt * ((A.ML + 63) / 64);
    VxDrop d; d.seed_ptr = p_drop > 0 ? (const uint64_t*)seed_ptr : nullptr; d.stream = dstream; d.p = p_drop;
    if (vx_pwa_attn_mfma_ok(plan, B, M, cq, cv) & 1) {
        const int rc = vx_pwa_attn_mfma_fwd(Q, K, V, table, O, LSE, mbits, plan, B, M, cq, cv, d, stream);
        if (rc) return rc;
        VX_LAUNCH_CHECK("vx_pwa_attn_fwd (mfma)");
        return 0;
    }
    const int Tsz = (2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
    const size_t tab_f = (size_t)((A.l + 3) & ~3) + (((size_t)Tsz * A.heads + 3) & ~(size_t)3);
    const int S = vx_attn_split(units, A.ML, A.M);
    const int one_head = (((long)A.Nt * ((A.ML + 63) / 64)) % (4 / S) == 0) ? 1 : 0;      // every unit of a block in one head: only that bias column is staged
    const size_t shm = ((one_head ? (size_t)((A.l + 3) & ~3) + (((size_t)Tsz + 3) & ~(size_t)3) : tab_f) + (size_t)4 * 64 * (cq + cv)) * sizeof(float);
    VX_REQUIRE(shm <= 160 * 1024, "vx_pwa_attn_fwd: tables do not fit LDS (%d entries x %d heads)", Tsz, A.heads);
    const bool found = vx_attn_dispatch(cq, cv, [&](auto pr) {
        vx_pwa_attn_fwd_k<decltype(pr)::a, decltype(pr)::b><<<dim3(vx_cdiv(units, 4 / S)), dim3(256), shm, (hipStream_t)stream>>>(Q, K, V, table, O, LSE, Tsz, A, d, S, one_head, mbits);
    });
    if (!found) VX_FAIL(-3, "PWA attention: unsupported head widths c_qk=%d c_v=%d", cq, cv);
    VX_LAUNCH_CHECK("vx_pwa_attn_fwd");
    return 0;
}

__global__ void __launch_bounds__(256) vx_zero4_k(float4* __restrict__ p, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int vx_pwa_attn_bwd_ws_floats(const VxPwaPlan* plan, int B, int M) {
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, 4, 4, "vx_pwa_attn_bwd_ws_floats")) return e;
    const long Tsz = (long)(2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
    const long rows = (long)A.BH * A.Nt * A.ML;
    const long n = ((rows + 3) & ~3L) + (long)VX_DTABLE_REPLICAS * Tsz * A.heads;
    VX_REQUIRE(n < 0x7fffffffL, "vx_pwa_attn_bwd_ws_floats: workspace too large");
    return (int)n;
}

// (round 6) where the bias-gradient replicas start inside delta_ws (floats; they run to the end of the workspace), and a one-shot note from the caller that it has
// zeroed them already (the scatter adjoint in front did: vx_pwa_scatter_bwd_all_wz): the NEXT attention backward of this host thread on that workspace launches no vx_zero4_k
extern "C" int vx_pwa_attn_bwd_rep_offset(const VxPwaPlan* plan, int B, int M) {
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, 4, 4, "vx_pwa_attn_bwd_rep_offset")) return e;
    const long rows = (long)A.BH * A.Nt * A.ML;
    VX_REQUIRE(((rows + 3) & ~3L) < 0x7fffffffL, "vx_pwa_attn_bwd_rep_offset: workspace too large");
    return (int)((rows + 3) & ~3L);
}
static thread_local const float* t_rep_zeroed = nullptr;
extern "C" int vx_pwa_attn_bwd_mark_rep_zeroed(const float* delta_ws) { t_rep_zeroed = delta_ws; return 0; }
static int vx_pwa_attn_bwd_run(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                               const float* dO, float* dQ, float* dK, float* dV, float* dtable, float* delta_ws,
                               const VxPwaPlan* plan, int B, int M, int cq, int cv,
                               const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream, bool fold, const unsigned short* mbits = nullptr);
extern "C" int vx_pwa_attn_bwd(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                               const float* dO, float* dQ, float* dK, float* dV, float* dtable, float* delta_ws,
                               const VxPwaPlan* plan, int B, int M, int cq, int cv,
                               const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream) {
    return vx_pwa_attn_bwd_run(Q, K, V, table, O, LSE, dO, dQ, dK, dV, dtable, delta_ws, plan, B, M, cq, cv, seed_ptr, dstream, p_drop, stream, true);
}
// vx_pwa_attn_bwd without its last step: the bias-table gradient stays in the replicas of delta_ws until vx_pwa_attn_bwd_fold adds them to dtable (a
// parameter gradient: nothing on the backward's dependent chain waits for it, the caller may launch the fold later).  Returns 1 and launches nothing
// when this geometry / mode folds inside its kernels (MFMA backward, two-launch backward): the caller then uses vx_pwa_attn_bwd.
extern "C" int vx_pwa_attn_bwd_nofold(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                                      const float* dO, float* dQ, float* dK, float* dV, float* delta_ws,
                                      const VxPwaPlan* plan, int B, int M, int cq, int cv,
                                      const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream) {
    if (!vx_pwa_attn_bwd1_ok(plan, B, M, cq, cv) && (!vx_attn_bwd_fused || (vx_pwa_attn_mfma_ok(plan, B, M, cq, cv) & 2))) return 1;
    return vx_pwa_attn_bwd_run(Q, K, V, table, O, LSE, dO, dQ, dK, dV, delta_ws /* unused, non-null */, delta_ws, plan, B, M, cq, cv, seed_ptr, dstream, p_drop, stream, false);
}
/* the same two entries with the forward's dropout mask words (vx_pwa_attn_fwd_mb): the one-pass MFMA backward then reads one bit per pair instead of
 * re-drawing the Philox words (mbits NULL or p_drop == 0: as the entries above) */
extern "C" int vx_pwa_attn_bwd_mb(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                                  const float* dO, float* dQ, float* dK, float* dV, float* dtable, float* delta_ws,
                                  const VxPwaPlan* plan, int B, int M, int cq, int cv,
                                  const void* seed_ptr, unsigned long long dstream, float p_drop, const void* mbits, void* stream) {
    return vx_pwa_attn_bwd_run(Q, K, V, table, O, LSE, dO, dQ, dK, dV, dtable, delta_ws, plan, B, M, cq, cv, seed_ptr, dstream, p_drop, stream, true,
                               p_drop > 0 ? (const unsigned short*)mbits : nullptr);
}
extern "C" int vx_pwa_attn_bwd_nofold_mb(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                                         const float* dO, float* dQ, float* dK, float* dV, float* delta_ws,
                                         const VxPwaPlan* plan, int B, int M, int cq, int cv,
                                         const void* seed_ptr, unsigned long long dstream, float p_drop, const void* mbits, void* stream) {
    if (!vx_pwa_attn_bwd1_ok(plan, B, M, cq, cv) && (!vx_attn_bwd_fused || (vx_pwa_attn_mfma_ok(plan, B, M, cq, cv) & 2))) return 1;
    return vx_pwa_attn_bwd_run(Q, K, V, table, O, LSE, dO, dQ, dK, dV, delta_ws /* unused, non-null */, delta_ws, plan, B, M, cq, cv, seed_ptr, dstream, p_drop, stream, false,
                               p_drop > 0 ? (const unsigned short*)mbits : nullptr);
}
extern "C" int vx_pwa_attn_bwd_fold(const float* delta_ws, float* dtable, const VxPwaPlan* plan, int B, int M, void* stream) {
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, 4, 4, "vx_pwa_attn_bwd_fold")) return e;
    VX_REQUIRE(delta_ws && dtable, "vx_pwa_attn_bwd_fold: null pointer");
    const long Tsz = (long)(2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
    const long rows = (long)A.BH * A.Nt * A.ML;
    const long nt = Tsz * A.heads;
    vx_attn_fold_k<<<dim3((unsigned)vx_cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream>>>(delta_ws + ((rows + 3) & ~3L), dtable, nt, VX_DTABLE_REPLICAS);
    VX_LAUNCH_CHECK("vx_pwa_attn_bwd_fold");
    return 0;
}
// the folds of up to 8 attention backward passes (e.g. the four transformer levels of one step) in ONE launch: each was a 2-27-block launch at the launch
// floor, one after the other at the end of the encoder backward
struct VxFoldMany { const float* rep[8]; float* dt[8]; long n[8]; };
__global__ void __launch_bounds__(256) vx_attn_fold_many_k(VxFoldMany P, int nrep) {
    const int j = blockIdx.y;
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= P.n[j]) return;
    float g = 0.0f;
    for (int r = 0; r < nrep; ++r) g += P.rep[j][(long)r * P.n[j] + k];
    P.dt[j][k] += g;
}
extern "C" int vx_pwa_attn_bwd_fold_many(const float* const* delta_ws, float* const* dtables, const VxPwaPlan* const* plans, int count, int B, int M, void* stream) {
    VX_REQUIRE(delta_ws && dtables && plans && count >= 1 && count <= 8, "vx_pwa_attn_bwd_fold_many: bad args");
    VxFoldMany P = {};
    long nmax = 0;
    for (int j = 0; j < count; ++j) {
        VxAttn A;
        if (int e = vx_attn_fill(A, plans[j], B, M, 4, 4, "vx_pwa_attn_bwd_fold_many")) return e;
        VX_REQUIRE(delta_ws[j] && dtables[j], "vx_pwa_attn_bwd_fold_many: null pointer");
        const long Tsz = (long)(2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
        const long rows = (long)A.BH * A.Nt * A.ML;
        P.rep[j] = delta_ws[j] + ((rows + 3) & ~3L);
        P.dt[j] = dtables[j];
        P.n[j] = Tsz * A.heads;
        nmax = P.n[j] > nmax ? P.n[j] : nmax;
    }
    vx_attn_fold_many_k<<<dim3((unsigned)vx_cdiv(nmax, 256), (unsigned)count), dim3(256), 0, (hipStream_t)stream>>>(P, VX_DTABLE_REPLICAS);
    VX_LAUNCH_CHECK("vx_pwa_attn_bwd_fold_many");
    return 0;
}
static int vx_pwa_attn_bwd_run(const float* Q, const float* K, const float* V, const float* table, const float* O, const float* LSE,
                               const float* dO, float* dQ, float* dK, float* dV, float* dtable, float* delta_ws,
                               const VxPwaPlan* plan, int B, int M, int cq, int cv,
                               const void* seed_ptr, unsigned long long dstream, float p_drop, void* stream, bool fold, const unsigned short* mbits) {
    VxAttn A;
    if (int e = vx_attn_fill(A, plan, B, M, cq, cv, "vx_pwa_attn_bwd")) return e;
    VX_REQUIRE(Q && K && V && table && O && LSE && dO && dQ && dK && dV && dtable && delta_ws, "vx_pwa_attn_bwd: null pointer");
    const long units = (long)A.BH * A.Nt * ((A.ML + 63) / 64);
    const int Tsz = (2 * A.n[0] - 1) * (2 * A.n[1] - 1) * (2 * A.n[2] - 1);
    const size_t tab_f = (size_t)((A.l + 3) & ~3) + (((size_t)Tsz * A.heads + 3) & ~(size_t)3);
    const int S = vx_attn_split(units, A.ML, A.M);
    // every unit of a block in one head (the head changes every Nt * chunks units): one bias column, one shared bias-gradient table per block
    const int one_head = (((long)A.Nt * ((A.ML + 63) / 64)) % (4 / S) == 0) ? 1 : 0;
    const size_t tab_b = one_head ? (size_t)((A.l + 3) & ~3) + (((size_t)Tsz + 3) & ~(size_t)3) : tab_f;
    const size_t shm = (tab_b + (((size_t)(one_head ? 1 : 4 / S) * Tsz + 3) & ~(size_t)3) + (size_t)4 * 64 * (cq + cv) + (size_t)4 * VX_PRIV_BINS) * sizeof(float);
    const size_t shm_kv = (tab_b + (size_t)4 * 64 * (cq + cv + 4)) * sizeof(float);
    VX_REQUIRE(shm <= 160 * 1024, "vx_pwa_attn_bwd: bias table too large for LDS (%d entries)", Tsz);
    VxDrop d; d.seed_ptr = p_drop > 0 ? (const uint64_t*)seed_ptr : nullptr; d.stream = dstream; d.p = p_drop;
    // workspace: [rows] delta | [VX_DTABLE_REPLICAS][Tsz*heads] bias-gradient replicas (zeroed here, folded into dtable by the dK/dV kernel)
    const long rows = (long)A.BH * A.Nt * A.ML;
    float* rep = delta_ws + ((rows + 3) & ~3L);
    const long rep_floats = (long)VX_DTABLE_REPLICAS * Tsz * A.heads;
    const unsigned nblk = (unsigned)vx_cdiv(units, 4 / S);
    const bool prezeroed = t_rep_zeroed == delta_ws;
    t_rep_zeroed = nullptr;
    if (!prezeroed) vx_zero4_k<<<dim3((unsigned)vx_cdiv(rep_floats / 4, 256)), dim3(256), 0, (hipStream_t)stream>>>(reinterpret_cast<float4*>(rep), rep_floats / 4);   // VX_DTABLE_REPLICAS % 4 == 0
    if (vx_pwa_attn_bwd1h_ok(plan, B, M, cq, cv) && (!(p_drop > 0 && seed_ptr) || mbits != nullptr)) {      // one pass, two fp16 pieces per operand on the 16x16x32 pipe (csrc/pwa_mfma.hip)
        const int rc = vx_pwa_attn_bwd1h(Q, K, V, table, O, LSE, dO, dQ, dK, dV, rep, mbits, plan, B, M, cq, cv, d, stream);
        if (rc) VX_FAIL(rc, "vx_pwa_attn_bwd: f16-pipe one-pass kernel could not be launched");
        const long nt = (long)Tsz * A.heads;
        if (fold) vx_attn_fold_k<<<dim3((unsigned)vx_cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream>>>(rep, dtable, nt, VX_DTABLE_REPLICAS);
        VX_LAUNCH_CHECK("vx_pwa_attn_bwd (one pass, f16 pipe)");
        return 0;
    }
    if (vx_pwa_attn_bwd1_ok(plan, B, M, cq, cv)) {             // one evaluation of the soft-max side per pair, every GEMM on MFMA (csrc/pwa_mfma.hip)
        const int rc = vx_pwa_attn_bwd1(Q, K, V, table, O, LSE, dO, dQ, dK, dV, rep, mbits, plan, B, M, cq, cv, d, stream);
        if (rc) VX_FAIL(rc, "vx_pwa_attn_bwd: one-pass kernel could not be launched");
        const long nt = (long)Tsz * A.heads;
        if (fold) vx_attn_fold_k<<<dim3((unsigned)vx_cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream>>>(rep, dtable, nt, VX_DTABLE_REPLICAS);
        VX_LAUNCH_CHECK("vx_pwa_attn_bwd (one pass)");
        return 0;
    }
    if (vx_pwa_attn_mfma_ok(plan, B, M, cq, cv) & 2) {
        const int rc = vx_pwa_attn_mfma_bwd(Q, K, V, table, O, LSE, dO, dQ, dK, dV, dtable, delta_ws, rep, plan, B, M, cq, cv, d, stream);
        if (rc) return rc;
        VX_LAUNCH_CHECK("vx_pwa_attn_bwd (mfma)");
        return 0;
    }
    const bool found = vx_attn_dispatch(cq, cv, [&](auto pr) {
        constexpr int CQ = decltype(pr)::a, CV = decltype(pr)::b;
        if (vx_attn_bwd_fused) {
            const size_t shm2 = shm > shm_kv ? shm : shm_kv;
            vx_pwa_attn_bwd_both_k<CQ, CV><<<dim3(2 * nblk), dim3(256), shm2, (hipStream_t)stream>>>(Q, K, V, table, O, LSE, dO, dQ, dK, dV, delta_ws, rep, Tsz, A, d, S, one_head, mbits);
            const long nt = (long)Tsz * A.heads;
            if (fold) vx_attn_fold_k<<<dim3((unsigned)vx_cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream>>>(rep, dtable, nt, VX_DTABLE_REPLICAS);
            return;
        }
        vx_pwa_attn_bwd_q_k<CQ, CV><<<dim3(nblk), dim3(256), shm, (hipStream_t)stream>>>(Q, K, V, table, O, LSE, dO, dQ, delta_ws, rep, Tsz, A, d, S, one_head, mbits);
        vx_pwa_attn_bwd_kv_k<CQ, CV><<<dim3(nblk), dim3(256), shm_kv, (hipStream_t)stream>>>(Q, K, V, table, LSE, delta_ws, dO, dK, dV, rep, dtable, Tsz, A, d, S, one_head, mbits);
    });
    if (!found) VX_FAIL(-3, "PWA attention: unsupported head widths c_qk=%d c_v=%d", cq, cv);
    VX_LAUNCH_CHECK("vx_pwa_attn_bwd");
    return 0;
}
