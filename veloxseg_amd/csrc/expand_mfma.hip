// Patch-expand convolution (Conv3d 16 -> 64*Cc, k3 p1, + PixelShuffle(4); reference Decoder.py:73-76,150-153, superpixel.py:16)
// as fp32-MFMA implicit GEMMs whose operands are read STRAIGHT FROM GLOBAL MEMORY in 256-byte runs.
//
// The pixel-shuffled tensor makes this possible: output channel co = ((c*4+s1)*4+s2)*4+s3 of coarse voxel (d,h,w) lives at
// fine[c][4d+s1][4h+s2][4w+s3], so for 16 consecutive coarse voxels along W the 4 values s3 = 0..3 form 64 CONSECUTIVE floats.
// With v_mfma_f32_16x16x4_f32 (A: lane(r,q) = A[row r][k q], B: lane(r,q) = B[k q][col r], D: lane(r,q), reg = D[row 4q+reg][col r]):
//   input gradient  dx[p, ci] = sum_{t,co} dyf[co, p-t+1] W[co,ci,t]:   rows = 16 coarse voxels, cols = 16 ci, k = s3
//        -> A is one coalesced 256-B load of the fine gradient, B one coalesced 256-B load of the tap-major weights WT[t][co][ci].
// Exact fp32 (the MFMA is a k-ordered fmaf chain), so it is interchangeable with the VALU kernels within round-off.
#include "vx_common.h"
#include <type_traits>
#include "../../include/veloxseg_hip.h"

// consecutive block ids go to different XCDs (8 L2 slices): tiles that share halo planes should sit on ONE XCD, whose L2 then serves the overlap -- de-interleave the
// block id so that every XCD walks a contiguous run of tiles (as vx_tz_k does)
__device__ __forceinline__ int vx_xcd_tile(int bid, int nb) { return (nb & 7) == 0 ? (bid & 7) * (nb >> 3) + (bid >> 3) : bid; }

typedef float vx_f4 __attribute__((ext_vector_type(4)));

// WT[t][co][ci] = W[co][ci][t]
__global__ void __launch_bounds__(256) vx_weight_tap_major_k(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int K3) {
    const long n = (long)Cout * Cin * K3;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int ci = (int)(e % Cin);
    const int co = (int)((e / Cin) % Cout);
    const int t = (int)(e / ((long)Cin * Cout));
    wt[e] = w[((long)co * Cin + ci) * K3 + t];
}

// x_cl[b][voxel][c] = x[b][c][voxel]  (C = 16): lets the weight-gradient kernel read 4 voxels x 16 channels as ONE 256-byte run
__global__ void __launch_bounds__(256) vx_to_channels_last16_k(const float* __restrict__ x, float* __restrict__ xcl, long V) {
    __shared__ float tile[16][65];
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * 64;
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int c = e / 64, k = e % 64;
        tile[c][k] = (v0 + k < V) ? x[((long)b * 16 + c) * V + v0 + k] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const int k = e / 16, c = e % 16;
        if (v0 + k < V) xcl[((long)b * V + v0 + k) * 16 + c] = tile[c][k];
    }
}

// one wave = MT consecutive 16-voxel tiles of the coarse volume (flattened d,h,w), all 16 input channels
template <int MT>
__global__ void __launch_bounds__(256) vx_expand_bwd_data_mfma_k(const float* __restrict__ dyf, const float* __restrict__ wt, float* __restrict__ dx,
                                                                 int B, int Cc, int D, int H, int W, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long V = (long)D * H * W;
    const int tiles = (int)((V + 15) / 16);
    const int groups = (tiles + MT - 1) / MT;
    const long gw = (long)blockIdx.x * 4 + wave;
    if (gw >= (long)B * groups) return;
    const int b = (int)(gw / groups);
    const int tile0 = (int)(gw % groups) * MT;
    const int r = lane & 15, q = lane >> 4;
    const int Cout = Cc * 64;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;                       // one fine channel
    int pd[MT], ph[MT], pw[MT];
    bool pok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const long p = (long)(tile0 + m) * 16 + r;
        pok[m] = (tile0 + m) < tiles && p < V;
        const long pp = pok[m] ? p : 0;
        pw[m] = (int)(pp % W);
        ph[m] = (int)((pp / W) % H);
        pd[m] = (int)(pp / ((long)W * H));
    }
    vx_f4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ dyb = dyf + (long)b * Cc * fplane;
    for (int t = 0; t < 27; ++t) {
        const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
        long abase[MT];
        bool aok[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int qd = pd[m] - td + 1, qh = ph[m] - th + 1, qw = pw[m] - tw + 1;
            aok[m] = pok[m] && (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
            abase[m] = ((long)(4 * qd) * FH + 4 * qh) * FW + 4 * qw + q;
        }
        const float* __restrict__ wtt = wt + ((long)t * Cout) * 16 + q * 16 + r;      // + co_base*16
        for (int c = 0; c < Cc; ++c) {
            for (int s1 = 0; s1 < 4; ++s1) {
                const long coff = (long)c * fplane + (long)s1 * FH * FW;
                const int co_base = ((c * 4 + s1) * 4) * 4;
                float bv[4], av[MT][4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    bv[s2] = wtt[(long)(co_base + s2 * 4) * 16];
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[m][s2] = aok[m] ? dyb[coff + abase[m] + (long)s2 * FW] : 0.0f;
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][s2], bv[s2], acc[m], 0, 0, 0);
            }
        }
    }
    // D: row = voxel 4q+reg of the tile, col = ci r
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const long p = (long)(tile0 + m) * 16 + 4 * q + reg;
            if ((tile0 + m) < tiles && p < V) {
                float* dst = dx + ((long)b * 16 + r) * V + p;
                *dst = accumulate ? *dst + acc[m][reg] : acc[m][reg];
            }
        }
    }
}


static int vx_expand_lds_enabled = 1;
extern "C" int vx_expand_set_lds(int on) { vx_expand_lds_enabled = on; return 0; }      // 1 (default): halo AND weights in LDS; 2: halo only; 0: every operand from global

// LDS-tiled variant (D % 4 == 0, H % 4 == 0, W % 4 == 0; rows that are not a multiple of 16 -- the 24-wide rows of the shipped 96^3 patches -- leave part of
// their last tile idle).  The kernel above reads every A operand (64 contiguous floats of the fine gradient)
// from L2/HBM once per tap: 693 MB of memory-side traffic per launch for ~50 MB of data (profiles/r01q_pmc_traffic.json).  Here a block owns a
// 4 x 4 x 16 coarse tile (wave = one d-slice = 4 M-tiles of 16 voxels along W); for each (c, s1) group the 6 x 6 x 18 halo of that group's
// fine rows ([hd][hh][s2][72 floats] = 41.5 KB) is staged once with 16-byte loads and all 27 taps x 4 s2 read their A operands from LDS
// (ds_read_b32, 64 consecutive floats per wave: conflict-free).
__global__ void __launch_bounds__(256) vx_expand_bwd_data_lds_k(const float* __restrict__ dyf, const float* __restrict__ wt, float* __restrict__ dx,
                                                                int B, int Cc, int D, int H, int W, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float vx_halo_t[];          // [6][6][4][72]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = blockIdx.x;
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const int Cout = Cc * 64;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const float* __restrict__ dyb = dyf + (long)b * Cc * fplane;
    vx_f4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cc; ++c) {
        for (int s1 = 0; s1 < 4; ++s1) {
            __syncthreads();
            // stage: 144 rows (hd, hh, s2) x 18 float4
            // 144 * 18 = 2592 float4 = 10.1 per thread: up to 11 unconditional loads in flight per thread (clamped address, value selected afterwards)
            {
                float4 v[11];
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int e = min((int)threadIdx.x + u * 256, 144 * 18 - 1);
                    const int f4 = e % 18, row = e / 18;
                    const int s2 = row & 3, hh = (row >> 2) % 6, hd = row / 24;
                    const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + f4;
                    const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
                    const float4 t_ = *reinterpret_cast<const float4*>(dyb + (long)c * fplane + (ok ? ((long)(4 * qd + s1) * FH + 4 * qh + s2) * FW + 4 * qw : 0));
                    v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int e = (int)threadIdx.x + u * 256;
                    if (e < 144 * 18) *reinterpret_cast<float4*>(vx_halo_t + (e / 18) * 72 + (e % 18) * 4) = v[u];
                }
            }
            __syncthreads();
            const int co_base = ((c * 4 + s1) * 4) * 4;
            for (int t = 0; t < 27; ++t) {
                const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
                const float* __restrict__ wtt = wt + ((long)t * Cout) * 16 + q * 16 + r;
                const int hd = wave - td + 2;
                const int lbase = (r - tw + 2) * 4 + q;
                float bv[4], av[4][4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    bv[s2] = wtt[(long)(co_base + s2 * 4) * 16];
#pragma unroll
                    for (int m = 0; m < 4; ++m) av[m][s2] = vx_halo_t[((hd * 6 + (m - th + 2)) * 4 + s2) * 72 + lbase];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][s2], bv[s2], acc[m], 0, 0, 0);
            }
        }
    }
    // D: row = voxel 4q+reg of the M-tile (w = w0 + 4q + reg), col = ci r
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long pbase = ((long)(d0 + wave) * H + (h0 + m)) * W + w0 + 4 * q;
        float* dst = dx + ((long)b * 16 + r) * V + pbase;
        if (w0 + 4 * q >= W) continue;                          // (W % 4 == 0: a lane's four voxels are inside or outside together)
        float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        if (accumulate) { const float4 old = *reinterpret_cast<float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
    }
}

// the same with the B operand (the weights of the group) staged in LDS beside the halo: 41.5 + 27 KB, two blocks per CU
__global__ void __launch_bounds__(256) vx_expand_bwd_data_lds_w_k(const float* __restrict__ dyf, const float* __restrict__ wt, float* __restrict__ dx,
                                                                int B, int Cc, int D, int H, int W, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float vx_halo_t[];          // [6][6][4][72] | [27][256] weights of the current (c, s1) group
    float* __restrict__ wl = vx_halo_t + 6 * 6 * 4 * 72;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = blockIdx.x;
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const int Cout = Cc * 64;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const float* __restrict__ dyb = dyf + (long)b * Cc * fplane;
    vx_f4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cc; ++c) {
        for (int s1 = 0; s1 < 4; ++s1) {
            __syncthreads();
            // stage: 144 rows (hd, hh, s2) x 18 float4
            // 144 * 18 = 2592 float4 = 10.1 per thread: up to 11 unconditional loads in flight per thread (clamped address, value selected afterwards)
            {
                // the group's weights: tap t, 16 outputs x 16 inputs = 256 contiguous floats of the tap-major copy, already in operand order
                // ((s2 * 4 + q) * 16 + r): a straight copy, 27 independent loads in flight, instead of 4 dependent global loads per tap and wave
                float wv[27];
                const float* __restrict__ wsrc = wt + (long)(((c * 4 + s1) * 4) * 4) * 16 + threadIdx.x;
#pragma unroll
                for (int t = 0; t < 27; ++t) wv[t] = wsrc[(long)t * Cout * 16];
#pragma unroll
                for (int t = 0; t < 27; ++t) wl[t * 256 + threadIdx.x] = wv[t];
            }
            {
                float4 v[11];
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int e = min((int)threadIdx.x + u * 256, 144 * 18 - 1);
                    const int f4 = e % 18, row = e / 18;
                    const int s2 = row & 3, hh = (row >> 2) % 6, hd = row / 24;
                    const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + f4;
                    const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
                    const float4 t_ = *reinterpret_cast<const float4*>(dyb + (long)c * fplane + (ok ? ((long)(4 * qd + s1) * FH + 4 * qh + s2) * FW + 4 * qw : 0));
                    v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int e = (int)threadIdx.x + u * 256;
                    if (e < 144 * 18) *reinterpret_cast<float4*>(vx_halo_t + (e / 18) * 72 + (e % 18) * 4) = v[u];
                }
            }
            __syncthreads();
#pragma unroll 3
            for (int t = 0; t < 27; ++t) {
                const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
                const float* __restrict__ wtt = wl + t * 256 + lane;
                const int hd = wave - td + 2;
                const int lbase = (r - tw + 2) * 4 + q;
                float bv[4], av[4][4];
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    bv[s2] = wtt[s2 * 64];
#pragma unroll
                    for (int m = 0; m < 4; ++m) av[m][s2] = vx_halo_t[((hd * 6 + (m - th + 2)) * 4 + s2) * 72 + lbase];
                }
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][s2], bv[s2], acc[m], 0, 0, 0);
            }
        }
    }
    // D: row = voxel 4q+reg of the M-tile (w = w0 + 4q + reg), col = ci r
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long pbase = ((long)(d0 + wave) * H + (h0 + m)) * W + w0 + 4 * q;
        float* dst = dx + ((long)b * 16 + r) * V + pbase;
        if (w0 + 4 * q >= W) continue;                          // (W % 4 == 0: a lane's four voxels are inside or outside together)
        float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        if (accumulate) { const float4 old = *reinterpret_cast<float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
    }
}

extern "C" int vx_expand_bwd_data_mfma(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W,
                                       int accumulate, void* stream) {
    VX_REQUIRE(dy_fine && w && wt_ws && dx && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_bwd_data_mfma: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int Cout = Cc * 64;
    const long nW = (long)Cout * 16 * 27;
    vx_weight_tap_major_k<<<vx_cdiv(nW, 256), 256, 0, st>>>(w, wt_ws, Cout, 16, 27);
    const long V = (long)D * H * W;
    if (vx_expand_lds_enabled && D % 4 == 0 && H % 4 == 0 && W % 4 == 0) {
        const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
        if (vx_expand_lds_enabled == 1) {
            const size_t shm = (6 * 6 * 4 * 72 + 27 * 256) * sizeof(float);
            static bool attr_set = false;
            if (!attr_set) { (void)hipFuncSetAttribute((const void*)vx_expand_bwd_data_lds_w_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_set = true; }
            vx_expand_bwd_data_lds_w_k<<<dim3((unsigned)nblk), 256, shm, st>>>(dy_fine, wt_ws, dx, B, Cc, D, H, W, accumulate);
        } else
        vx_expand_bwd_data_lds_k<<<dim3((unsigned)nblk), 256, 6 * 6 * 4 * 72 * sizeof(float), st>>>(dy_fine, wt_ws, dx, B, Cc, D, H, W, accumulate);
        VX_LAUNCH_CHECK("vx_expand_bwd_data_mfma");
        return 0;
    }
    const int tiles = (int)((V + 15) / 16);
    constexpr int MT = 4;
    const int groups = (tiles + MT - 1) / MT;
    vx_expand_bwd_data_mfma_k<MT><<<vx_cdiv((long)B * groups, 4), 256, 0, st>>>(dy_fine, wt_ws, dx, B, Cc, D, H, W, accumulate);
    VX_LAUNCH_CHECK("vx_expand_bwd_data_mfma");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// forward: y_fine[c][4d+s1][4h+s2][4w+s3] = bias[co] + sum_{ci,t} W[co][ci][t] x[ci][(d,h,w) + t - 1],  co = ((c*4+s1)*4+s2)*4+s3
//   D tile: rows = the 16 output channels (s2, s3) of one (c, s1) group, cols = 16 coarse voxels along W  ->  lane (r, q) ends with
//   (s2 = q, s3 = 0..3) of voxel w0 + r = ONE float4 of the fine row 4h+q, and the 16 lanes r write 256 contiguous bytes.
//   k = (tap t, channel quad j): A[row][k q] = WT[t][co_base+row][4j+q] (tap-major weights, 1 KB contiguous per load),
//   B[k q][col r] = x[4j+q][voxel r + tap] from the block's LDS halo ([ci][6*6*18 voxels], plane pitch 656 = 16 mod 32 banks: 2 lanes per bank,
//   the minimum for 64 lanes).  Block = 4 x 4 x 16 coarse tile, wave = d-slice, 4 N-tiles (h rows) per wave share every A operand.
//   The VALU kernel this replaces (vx_conv_s1_k<3,16>) ran the same 7.2 GFLOP at 29 TFLOP/s.
// ------------------------------------------------------------------------------------------------------------------
#define VX_EF_PITCH 656
__global__ void __launch_bounds__(256) vx_expand_fwd_mfma_k(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                            float* __restrict__ y, int B, int Cc, int D, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float vx_xh[];          // [16][VX_EF_PITCH]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = blockIdx.x;
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const int Cout = Cc * 64;
    const float* __restrict__ xb = x + (long)b * 16 * V;
    // halo: 6 x 6 x 18 voxels x 16 channels, zero outside the volume; 8 unconditional loads in flight per thread (clamped address, value selected afterwards)
    for (int e0 = threadIdx.x; e0 < 16 * 648; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u * 256, 16 * 648 - 1);
            const int hv = e % 648, ci = e / 648;
            const int hw = hv % 18, hh = (hv / 18) % 6, hd = hv / 108;
            const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + hw;
            const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
            const float t_ = xb[ok ? (long)ci * V + ((long)qd * H + qh) * W + qw : 0];
            v[u] = ok ? t_ : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            if (e < 16 * 648) vx_xh[(e / 648) * VX_EF_PITCH + (e % 648)] = v[u];
        }
    }
    __syncthreads();
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    for (int g = 0; g < Cc * 4; ++g) {                                     // (c, s1) groups
        const int c = g >> 2, s1 = g & 3;
        const int co_base = g * 16;
        vx_f4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < 27; ++t) {
            const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
            const float* __restrict__ wtt = wt + ((long)t * Cout + co_base + r) * 16 + q;
            const float* __restrict__ xt = vx_xh + q * VX_EF_PITCH + ((wave + td) * 6 + th) * 18 + r + tw;
            float av[4], bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = wtt[4 * j];
#pragma unroll
                for (int m = 0; m < 4; ++m) bv[j][m] = xt[4 * j * VX_EF_PITCH + m * 18];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j][m], acc[m], 0, 0, 0);
        }
        // D: row 4q+reg = (s2 = q, s3 = reg), col r = voxel w0 + r
        const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + co_base + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* __restrict__ yb = y + ((long)b * Cc + c) * fplane + ((long)(4 * (d0 + wave) + s1) * FH + q) * FW + 4 * (w0 + r);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (w0 + r < W) *reinterpret_cast<float4*>(yb + (long)(4 * (h0 + m)) * FW) = make_float4(acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w);
    }
}

// The same forward with the A operand (the tap-major weights of one (c, s1) group: 27 taps x 16 outputs x 16 inputs = 27 KB) staged in LDS once per
// group and block, in operand order (element (t, j, lane) at (t*4 + j)*64 + lane: conflict-free ds_read_b32), instead of four dependent global loads
// per tap in each of the four waves: the SQ counters showed the kernel above waiting 65 % of its wave cycles (profiles/r02_sq_wave_breakdown.txt),
// 7.6 % active.  LDS 42 + 27 KB: still two blocks per CU.
__global__ void __launch_bounds__(256) vx_expand_fwd_mfma_w_k(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                                                              float* __restrict__ y, int B, int Cc, int D, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float vx_xh[];          // [16][VX_EF_PITCH] halo | [27][4][64] weights of the current group
    float* __restrict__ wl = vx_xh + 16 * VX_EF_PITCH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = blockIdx.x;
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const int Cout = Cc * 64;
    const float* __restrict__ xb = x + (long)b * 16 * V;
    for (int e0 = threadIdx.x; e0 < 16 * 648; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = min(e0 + u * 256, 16 * 648 - 1);
            const int hv = e % 648, ci = e / 648;
            const int hw = hv % 18, hh = (hv / 18) % 6, hd = hv / 108;
            const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + hw;
            const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
            const float t_ = xb[ok ? (long)ci * V + ((long)qd * H + qh) * W + qw : 0];
            v[u] = ok ? t_ : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * 256;
            if (e < 16 * 648) vx_xh[(e / 648) * VX_EF_PITCH + (e % 648)] = v[u];
        }
    }
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    // staging slot of this thread: source element tid of a tap's 16 x 16 block = (output row tid >> 4, input channel tid & 15) -> operand slot
    const int st_dst = ((threadIdx.x & 15) >> 2) * 64 + (threadIdx.x & 3) * 16 + (threadIdx.x >> 4);
    for (int g = 0; g < Cc * 4; ++g) {                                     // (c, s1) groups
        const int c = g >> 2, s1 = g & 3;
        const int co_base = g * 16;
        {
            float wv[27];
            const float* __restrict__ src = wt + (long)co_base * 16 + threadIdx.x;
#pragma unroll
            for (int t = 0; t < 27; ++t) wv[t] = src[(long)t * Cout * 16];       // 27 independent 1 KB-coalesced loads in flight
            __syncthreads();                                               // (the previous group's MFMAs have read wl; first pass: the halo stores are done)
#pragma unroll
            for (int t = 0; t < 27; ++t) wl[t * 256 + st_dst] = wv[t];
            __syncthreads();
        }
        vx_f4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
        for (int t = 0; t < 27; ++t) {
            const int tw = t % 3, th = (t / 3) % 3, td = t / 9;
            const float* __restrict__ wa = wl + t * 256 + lane;
            const float* __restrict__ xt = vx_xh + q * VX_EF_PITCH + ((wave + td) * 6 + th) * 18 + r + tw;
            float av[4], bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                av[j] = wa[j * 64];
#pragma unroll
                for (int m = 0; m < 4; ++m) bv[j][m] = xt[4 * j * VX_EF_PITCH + m * 18];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j][m], acc[m], 0, 0, 0);
        }
        const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + co_base + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* __restrict__ yb = y + ((long)b * Cc + c) * fplane + ((long)(4 * (d0 + wave) + s1) * FH + q) * FW + 4 * (w0 + r);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (w0 + r < W) *reinterpret_cast<float4*>(yb + (long)(4 * (h0 + m)) * FW) = make_float4(acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w);
    }
}
static int vx_expand_fwd_wlds = 1;
extern "C" int vx_expand_set_fwd_wlds(int on) { vx_expand_fwd_wlds = on ? 1 : 0; return 0; }      // A/B: weights of a group staged in LDS (default) or loaded per tap

// returns 1 when the shape is not covered (caller uses the direct convolution), 0 on success
extern "C" int vx_expand_fwd_mfma(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && w && wt_ws && y && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_fwd_mfma: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int Cout = Cc * 64;
    const long nW = (long)Cout * 16 * 27;
    vx_weight_tap_major_k<<<vx_cdiv(nW, 256), 256, 0, st>>>(w, wt_ws, Cout, 16, 27);
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    if (vx_expand_fwd_wlds) {
        const size_t shm = (16 * VX_EF_PITCH + 27 * 256) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)vx_expand_fwd_mfma_w_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_set = true; }
        vx_expand_fwd_mfma_w_k<<<dim3((unsigned)nblk), 256, shm, st>>>(x, wt_ws, bias, y, B, Cc, D, H, W);
    } else
    vx_expand_fwd_mfma_k<<<dim3((unsigned)nblk), 256, 16 * VX_EF_PITCH * sizeof(float), st>>>(x, wt_ws, bias, y, B, Cc, D, H, W);
    VX_LAUNCH_CHECK("vx_expand_fwd_mfma");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co, ci, t] += sum_{b,p} dyf[co, p] * x[ci, p + t - 1]   (+ db[co] += sum dyf[co, p])
//   rows = 16 output channels of one (c, s1) group (row r <-> s2 = r>>2, s3 = r&3), cols = 16 input channels, k = 4 coarse voxels along W.
//   A (fine gradient): lane (r,q) reads fine[c][4d+s1][4h+s2][4(w4+q)+s3]  -> 4 fine rows x 64 contiguous bytes per load;
//   B (coarse input, CHANNELS-LAST copy): lane (r,q) reads xcl[p + t - 1 (voxel w4+q+tw-1)][ci=r] -> one 256-byte run per tap.
//   One wave = one (c,s1) group x a run of k-steps; 27 accumulator tiles (108 VGPRs); one float atomic per weight per wave at the end.
// ------------------------------------------------------------------------------------------------------------------
template <int GS>
__global__ void __launch_bounds__(256) vx_expand_wgrad_mfma_k(const float* __restrict__ x, const float* __restrict__ dyf, float* __restrict__ dw,
                                                              float* __restrict__ db, int B, int Cc, int D, int H, int W, int steps_per_wave, int chunks_per_b) {
    // One wave = TWO (c, s1) groups (s1 = 2 sp, 2 sp + 1) sharing the 27 coarse-input operands of every k-step: 29 operand loads feed 54 MFMAs
    // (was 28 for 27; the kernel ran at the L2 read rate, not at the MFMA rate: every group re-reads all of x once per tap).  2 x 108 accumulator
    // registers + the double-buffered operands fit the 512-register file of a wave that owns its SIMD (launch bounds 256 = 1 wave per SIMD).
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ float red[16 * 16 * 27];
    const long gw_raw = (long)blockIdx.x * 4 + wave;
    const bool active = gw_raw < (long)B * chunks_per_b;
    const long gw = active ? gw_raw : 0;
    const int b = (int)(gw / chunks_per_b);
    const int chunk = (int)(gw % chunks_per_b);
    const int mt = blockIdx.y;                          // (c, s1 pair)
    const int c = mt / (4 / GS), sp = mt % (4 / GS);
    const int r = lane & 15, q = lane >> 4;
    const int W4 = (W + 3) / 4;                         // k-steps per coarse row
    const long nsteps = (long)D * H * W4;
    const long V = (long)D * H * W;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const long gstride = FH * FW;                       // s1 -> s1 + 1
    const float* __restrict__ dyb = dyf + ((long)b * Cc + c) * fplane + (long)(GS * sp) * gstride + (long)(r >> 2) * FW + (r & 3);
    const float* __restrict__ xb = x + (long)b * V * 16 + r;       // channels-last: [voxel][16]
    vx_f4 acc[GS][27];
#pragma unroll
    for (int g = 0; g < GS; ++g)
#pragma unroll
        for (int t = 0; t < 27; ++t) acc[g][t] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    float bsum[GS];
#pragma unroll
    for (int g = 0; g < GS; ++g) bsum[g] = 0.0f;
    const long s_begin = (long)chunk * steps_per_wave;
    const long s_end = !active ? s_begin : ((s_begin + steps_per_wave < nsteps) ? s_begin + steps_per_wave : nsteps);
    // software pipeline: the operand loads of step s+1 are issued before the MFMAs of step s; the position of the NEXT step to load is advanced
    // incrementally (three run-time integer divisions per step used to cost about as many issue cycles as the MFMAs they fed)
    int nw4 = (int)(s_begin % W4), nh = (int)((s_begin / W4) % H), nd = (int)(s_begin / ((long)W4 * H));
    long ns = s_begin;
    const long HW16 = (long)H * W * 16, W16 = (long)W * 16;
    auto load_step = [&](float (&av)[GS], float (&bv)[27]) {
        const int pw = 4 * nw4 + q;
        const bool vok = pw < W && ns < s_end;
        const long fo = ((long)(4 * nd) * FH + 4 * nh) * FW + 4 * pw;
#pragma unroll
        for (int g = 0; g < GS; ++g) av[g] = vok ? dyb[fo + g * gstride] : 0.0f;
        const float* __restrict__ xp = xb + (((long)nd * H + nh) * W + pw) * 16;
        const bool okd[3] = {vok && nd > 0, vok, vok && nd < D - 1};
        const bool okh[3] = {nh > 0, true, nh < H - 1};
        const bool okw[3] = {pw > 0, true, pw < W - 1};
#pragma unroll
        for (int td = 0; td < 3; ++td)
#pragma unroll
            for (int th = 0; th < 3; ++th)
#pragma unroll
                for (int tw = 0; tw < 3; ++tw)
                    bv[(td * 3 + th) * 3 + tw] = (okd[td] && okh[th] && okw[tw]) ? xp[(td - 1) * HW16 + (th - 1) * W16 + (tw - 1) * 16] : 0.0f;
        ++ns;
        if (++nw4 == W4) { nw4 = 0; if (++nh == H) { nh = 0; ++nd; } }
    };
    float av0[GS], bv0[27], av1[GS], bv1[27];
    load_step(av0, bv0);
    for (long s = s_begin; s < s_end; s += 2) {
        load_step(av1, bv1);
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            bsum[g] += av0[g];
#pragma unroll
            for (int t = 0; t < 27; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[g], bv0[t], acc[g][t], 0, 0, 0);
        }
        load_step(av0, bv0);
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            bsum[g] += av1[g];
#pragma unroll
            for (int t = 0; t < 27; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[g], bv1[t], acc[g][t], 0, 0, 0);
        }
    }
    // D: row = 4q+reg -> co = group*16 + row, col = ci r.  The 4 waves of the block (same groups, different voxel runs) are summed in LDS in the
    // final [co][ci][t] order, one group after the other, then flushed as CONTIGUOUS float atomics (64 consecutive floats per wave instruction).
#pragma unroll
    for (int g = 0; g < GS; ++g) {
        // the four waves hold the same (lane -> element) map: they add their tiles into `red` one after the other with plain LDS read-add-write
        // (scattered ds_add_f32 retire a few lanes per clock: 108 of them per wave cost as much as the 64 k-steps of MFMAs they concluded)
        for (int wv = 0; wv < 4; ++wv) {
            __syncthreads();
            if (wave == wv) {
#pragma unroll
                for (int t = 0; t < 27; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        float* __restrict__ e = &red[((4 * q + reg) * 16 + r) * 27 + t];
                        const float v = active ? acc[g][t][reg] : 0.0f;
                        *e = (wv == 0) ? v : *e + v;
                    }
            }
        }
        __syncthreads();
        const int grp = c * 4 + GS * sp + g;
        float* __restrict__ dwg = dw + (long)grp * 16 * 16 * 27;
        for (int e = threadIdx.x; e < 16 * 16 * 27; e += 256) atomicAdd(dwg + e, red[e]);
        if (db != nullptr && active) {
            float bs = bsum[g];
            bs += __shfl_xor(bs, 16, 64);
            bs += __shfl_xor(bs, 32, 64);
            if (q == 0) atomicAdd(db + grp * 16 + r, bs);
        }
    }
}

extern "C" int vx_expand_wgrad_mfma(const float* x, float* xcl_ws, const float* dy_fine, float* dw, float* db, int B, int Cc, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && xcl_ws && dy_fine && dw && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_wgrad_mfma: bad args");
    {
        const long Vx = (long)D * H * W;
        vx_to_channels_last16_k<<<dim3(vx_cdiv(Vx, 64), B), 256, 0, (hipStream_t)stream>>>(x, xcl_ws, Vx);
    }
    const long nsteps = (long)D * H * ((W + 3) / 4);
    constexpr int gs = 2;                            // one wave carries two of the four s1 groups of a channel block
    const int groups = Cc * 4 / gs;
    long waves_per_group = 1024 / groups;            // one resident wave per SIMD (508 registers): one round of 1024 waves, the flush amortised over >= 128 k-steps
    if (waves_per_group < 1) waves_per_group = 1;
    long spw = ((long)B * nsteps + waves_per_group - 1) / waves_per_group;
    if (spw < 16) spw = 16;
    const int chunks_per_b = vx_cdiv(nsteps, spw);
    dim3 grid(vx_cdiv((long)B * chunks_per_b, 4), groups);
    vx_expand_wgrad_mfma_k<gs><<<grid, 256, 0, (hipStream_t)stream>>>(xcl_ws, dy_fine, dw, db, B, Cc, D, H, W, (int)spw, chunks_per_b);
    VX_LAUNCH_CHECK("vx_expand_wgrad_mfma");
    return 0;
}

// ======================================================================================================================================
// bf16 opt-in mode (BASELINE configs[1]; veloxseg_amd.set_precision("bf16")): the patch-expand forward and input gradient with bf16 MFMA
// OPERANDS (v_mfma_f32_16x16x32_bf16: 16 SIMD-clk for 8192 MACs, fp32 accumulate), storage fp32 as everywhere else.  K of one MFMA = TWO taps x 16
// channels (forward) / two taps x 16 (s2, s3) sub-positions (input gradient): lane (r, q) holds k = 8 q + j, i.e. tap 2 p + (q >> 1) and the
// channels / sub-positions 8 (q & 1) + j.  27 taps = 14 pairs (the last one half empty): 56 MFMAs per (c, s1) group instead of 432 fp32 ones.
// Weights are converted once per launch into an operand-order bf16 image (one 16-byte load per lane and pair); activations are rounded to
// bf16 (nearest-even) when they enter LDS (forward) or when they leave it (input gradient).  Error: 2^-9 relative per product, fp32 sums.
// ======================================================================================================================================
typedef __bf16 vx_bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t vx_pack_bf16(float a, float b) {            // (a -> low half, b -> high half), round to nearest even
    // plain casts: hipcc emits v_cvt_pk_bf16_f32, which keeps a NaN a NaN (the integer rounding trick turns some NaNs into 0 or infinity)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ vx_bf8 vx_as_bf8(uint4 v) { return __builtin_bit_cast(vx_bf8, v); }

// img[((g * 14 + p) * 64 + lane) * 8 + j], g = (c, s1) group of 16 output channels
//   forward : row r = lane & 15 -> co = 16 g + r;  tap = 2 p + (q >> 1), ci = 8 (q & 1) + j
//   backward: col r -> ci = r;                     tap = 2 p + (q >> 1), co = 16 g + 4 s2 + s3 with s2 = 2 (q & 1) + (j >> 2), s3 = j & 3
__global__ void __launch_bounds__(256) vx_expand_wimg_bf16_k(const float* __restrict__ w, uint32_t* __restrict__ img, int groups, int backward) {
    const long n = (long)groups * 14 * 64 * 4;               // packed pairs
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int jp = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long t = e >> 8;
    const int p = (int)(t % 14), g = (int)(t / 14);
    const int r = lane & 15, q = lane >> 4;
    const int tap = 2 * p + (q >> 1);
    float v[2] = {0.f, 0.f};
    if (tap < 27)
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * jp + u;
            int co, ci;
            if (backward) { co = 16 * g + 4 * (2 * (q & 1) + (j >> 2)) + (j & 3); ci = r; }
            else { co = 16 * g + r; ci = 8 * (q & 1) + j; }
            v[u] = w[((long)co * 16 + ci) * 27 + tap];
        }
    img[e] = vx_pack_bf16(v[0], v[1]);
}

// TY = element type of the pixel-shuffled output (float, or vx_bf16 in the bf16 storage mode: the full-resolution logits / reconstructions)
template <typename TY>
__global__ void __launch_bounds__(256) vx_expand_fwd_bf16_k(const float* __restrict__ x, const uint4* __restrict__ wimg, const float* __restrict__ bias,
                                                            TY* __restrict__ y, int B, int Cc, int D, int H, int W) {
    __shared__ uint4 xh[2 * 648];                       // [channel half][6 x 6 x 18 halo voxel] x 8 bf16
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = vx_xcd_tile(blockIdx.x, gridDim.x);
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const float* __restrict__ xb = x + (long)b * 16 * V;
    for (int e = threadIdx.x; e < 2 * 648; e += 256) {
        const int hv = e % 648, half = e / 648;
        const int hw = hv % 18, hh = (hv / 18) % 6, hd = hv / 108;
        const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + hw;
        const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
        const float* __restrict__ src = xb + (long)(8 * half) * V + (ok ? ((long)qd * H + qh) * W + qw : 0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(long)j * V];
        uint4 pk = make_uint4(0u, 0u, 0u, 0u);
        if (ok) pk = make_uint4(vx_pack_bf16(v[0], v[1]), vx_pack_bf16(v[2], v[3]), vx_pack_bf16(v[4], v[5]), vx_pack_bf16(v[6], v[7]));
        xh[e] = pk;
    }
    __syncthreads();
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    // this lane's B-operand base per tap pair: tap = 2 p + (q >> 1), channel half q & 1, voxel column r
    for (int g = 0; g < Cc * 4; ++g) {                                     // (c, s1) groups
        const int c = g >> 2, s1 = g & 3;
        const int co_base = g * 16;
        vx_f4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
        const uint4* __restrict__ wg = wimg + (long)g * 14 * 64 + lane;
#pragma unroll 2
        for (int p = 0; p < 14; ++p) {
            const int t = 2 * p + (q >> 1);
            const int tt = t < 27 ? t : 26;                                // (the empty half pair has zero weights)
            const int tw = tt % 3, th = (tt / 3) % 3, td = tt / 9;
            const uint4 av = wg[p * 64];
            const uint4* __restrict__ xt = xh + (q & 1) * 648 + ((wave + td) * 6 + th) * 18 + r + tw;
            uint4 bv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) bv[m] = xt[m * 18];
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vx_as_bf8(av), vx_as_bf8(bv[m]), acc[m], 0, 0, 0);
        }
        // D: row 4q+reg = (s2 = q, s3 = reg), col r = voxel w0 + r
        const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + co_base + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        TY* __restrict__ yb = y + ((long)b * Cc + c) * fplane + ((long)(4 * (d0 + wave) + s1) * FH + q) * FW + 4 * (w0 + r);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (w0 + r < W) vx_st4(yb, (long)(4 * (h0 + m)) * FW, make_float4(acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w));
    }
}

// y_h16 != 0: y is a vx_bf16 array (bf16 storage mode)
extern "C" int vx_expand_fwd_mfma_bf16_h(const float* x, const float* w, const float* bias, float* wt_ws, void* y, int B, int Cc, int D, int H, int W, int y_h16, void* stream) {
    VX_REQUIRE(x && w && wt_ws && y && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_fwd_mfma_bf16: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int groups = Cc * 4;
    vx_expand_wimg_bf16_k<<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 0);
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    if (y_h16) vx_expand_fwd_bf16_k<vx_bf16><<<dim3((unsigned)nblk), 256, 0, st>>>(x, reinterpret_cast<const uint4*>(wt_ws), bias, (vx_bf16*)y, B, Cc, D, H, W);
    else vx_expand_fwd_bf16_k<float><<<dim3((unsigned)nblk), 256, 0, st>>>(x, reinterpret_cast<const uint4*>(wt_ws), bias, (float*)y, B, Cc, D, H, W);
    VX_LAUNCH_CHECK("vx_expand_fwd_mfma_bf16");
    return 0;
}
extern "C" int vx_expand_fwd_mfma_bf16(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, void* stream) {
    return vx_expand_fwd_mfma_bf16_h(x, w, bias, wt_ws, y, B, Cc, D, H, W, 0, stream);
}

// input gradient, bf16 operands: the block / wave / halo layout of vx_expand_bwd_data_lds_k (fp32 halo of one (c, s1) group in LDS), rows = 16 coarse
// voxels, cols = 16 ci; A = 2 x 4 contiguous fine-gradient floats of this lane's tap (rounded to bf16 here), B = the operand-order weight image
// (round 6) the halo of a (c, s1) group is staged in LDS AS bf16 -- rounded once when it is staged, or copied when the gradient already is a bf16 array (TD = vx_bf16) --
// instead of as fp32 with four v_cvt_pk per MFMA operand in the inner loop: the loop was VALU-bound (32 conversions + 8 16-byte LDS reads per 4 MFMAs and lane); now it
// reads two 8-byte pieces per operand and converts nothing.  Half the LDS (21 KB per block).
template <typename TD>
__global__ void __launch_bounds__(256) vx_expand_bwd_data_bf16_k(const TD* __restrict__ dyf, const uint4* __restrict__ wimg, float* __restrict__ dx,
                                                                 int B, int Cc, int D, int H, int W, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned short vx_halo_h[];          // [6][6][4][72] bf16
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;          // W % 16 != 0 (W % 4 == 0): the last tile along W is partly outside the volume
    int tile = vx_xcd_tile(blockIdx.x, gridDim.x);
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const TD* __restrict__ dyb = dyf + (long)b * Cc * fplane;
    vx_f4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    // the thread's staging items do not depend on (c, s1): offsets and inside flags once
    long goff[11];
    int loff[11];
#pragma unroll
    for (int u = 0; u < 11; ++u) {
        const int e = min((int)threadIdx.x + u * 256, 144 * 18 - 1);
        const int f4 = e % 18, row = e / 18;
        const int s2 = row & 3, hh = (row >> 2) % 6, hd = row / 24;
        const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + f4;
        const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
        goff[u] = ok ? ((long)(4 * qd) * FH + 4 * qh + s2) * FW + 4 * qw : -1;
        loff[u] = ((int)threadIdx.x + u * 256 < 144 * 18) ? (e / 18) * 72 + (e % 18) * 4 : -1;
    }
    for (int c = 0; c < Cc; ++c) {
        for (int s1 = 0; s1 < 4; ++s1) {
            __syncthreads();
            {
                uint2 v[11];
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const long o = (long)c * fplane + (goff[u] >= 0 ? goff[u] + (long)s1 * FH * FW : 0);
                    if constexpr (std::is_same<TD, vx_bf16>::value) {
                        const uint2 t_ = *reinterpret_cast<const uint2*>(dyb + o);
                        v[u] = goff[u] >= 0 ? t_ : make_uint2(0u, 0u);
                    } else {
                        const float4 t_ = *reinterpret_cast<const float4*>(dyb + o);
                        v[u] = goff[u] >= 0 ? make_uint2(vx_pack_bf16(t_.x, t_.y), vx_pack_bf16(t_.z, t_.w)) : make_uint2(0u, 0u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 11; ++u)
                    if (loff[u] >= 0) *reinterpret_cast<uint2*>(vx_halo_h + loff[u]) = v[u];
            }
            __syncthreads();
            const uint4* __restrict__ wg = wimg + (long)(c * 4 + s1) * 14 * 64 + lane;
#pragma unroll 2
            for (int p = 0; p < 14; ++p) {
                const int t = 2 * p + (q >> 1);
                const int tt = t < 27 ? t : 26;
                const int tw = tt % 3, th = (tt / 3) % 3, td = tt / 9;
                const uint4 bv = wg[p * 64];
                const int hd = wave - td + 2;
                const int s2 = 2 * (q & 1);
                const unsigned short* __restrict__ ht = vx_halo_h + ((hd * 6 + (2 - th)) * 4 + s2) * 72 + (r - tw + 2) * 4;
                uint4 av[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint2 lo = *reinterpret_cast<const uint2*>(ht + m * 4 * 72);
                    const uint2 hi = *reinterpret_cast<const uint2*>(ht + m * 4 * 72 + 72);
                    av[m] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vx_as_bf8(av[m]), vx_as_bf8(bv), acc[m], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long pbase = ((long)(d0 + wave) * H + (h0 + m)) * W + w0 + 4 * q;
        float* dst = dx + ((long)b * 16 + r) * V + pbase;
        if (w0 + 4 * q >= W) continue;                          // (W % 4 == 0: a lane's four voxels are inside or outside together)
        float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        if (accumulate) { const float4 old = *reinterpret_cast<float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
    }
}

// dy_h16 != 0: the fine gradient is a vx_bf16 array (bf16 storage mode)
extern "C" int vx_expand_bwd_data_mfma_bf16_h(const void* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W,
                                              int accumulate, int dy_h16, void* stream) {
    VX_REQUIRE(dy_fine && w && wt_ws && dx && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_bwd_data_mfma_bf16: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int groups = Cc * 4;
    vx_expand_wimg_bf16_k<<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 1);
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    if (dy_h16) vx_expand_bwd_data_bf16_k<vx_bf16><<<dim3((unsigned)nblk), 256, 6 * 6 * 4 * 72 * sizeof(unsigned short), st>>>((const vx_bf16*)dy_fine, reinterpret_cast<const uint4*>(wt_ws), dx, B, Cc, D, H, W, accumulate);
    else vx_expand_bwd_data_bf16_k<float><<<dim3((unsigned)nblk), 256, 6 * 6 * 4 * 72 * sizeof(unsigned short), st>>>((const float*)dy_fine, reinterpret_cast<const uint4*>(wt_ws), dx, B, Cc, D, H, W, accumulate);
    VX_LAUNCH_CHECK("vx_expand_bwd_data_mfma_bf16");
    return 0;
}
extern "C" int vx_expand_bwd_data_mfma_bf16(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W,
                                            int accumulate, void* stream) {
    return vx_expand_bwd_data_mfma_bf16_h(dy_fine, w, wt_ws, dx, B, Cc, D, H, W, accumulate, 0, stream);
}


// ======================================================================================================================================
// fp32-ACCURATE patch-expand on the bf16 matrix pipe: every fp32 operand is split into NS bf16 pieces (x = x0 + x1 (+ x2), each piece the bf16 rounding
// of what the previous ones left) and a product is the sum of the piece products whose weight exceeds the target accuracy:
//   NS = 2: x0 w0 + x0 w1 + x1 w0                        (3 MFMAs, neglects 2^-18-relative terms: ~1e-5 relative per product)
//   NS = 3: ... + x0 w2 + x2 w0 + x1 w1                   (6 MFMAs, neglects 2^-27-relative terms: the fp32 product to its last bits; fp32 accumulate)
// v_mfma_f32_16x16x32_bf16 does 8192 MACs in 16 SIMD clocks against 1024 in 32 for v_mfma_f32_16x16x4_f32: 6 bf16 MFMAs per 32 k-steps = 96 clocks
// against 256 (2.7 x), 3 = 48 clocks (5.3 x).  Layout, tiling and the operand-order weight image are those of the bf16 opt-in kernels above (K of one
// MFMA = two taps x 16 channels); the halo is split once when it is staged (forward) or when it is packed (input gradient).
// These kernels compute in fp32 ACCURACY on fp32 storage: they are the default of the fp32 mode when the unchanged fp32 parity tests pass with them
// (functional.EXPAND_SPLIT; tests/test_hip_model_gpu.py runs the whole-network oracle comparisons through them).
// ======================================================================================================================================
template <int NS>
__device__ __forceinline__ void vx_split8(const float (&v)[8], uint4 (&out)[NS]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        uint32_t pk[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
            const bf2 h = {(__bf16)r[2 * j2], (__bf16)r[2 * j2 + 1]};
            pk[j2] = __builtin_bit_cast(uint32_t, h);
            if (s + 1 < NS) { r[2 * j2] -= (float)h[0]; r[2 * j2 + 1] -= (float)h[1]; }
        }
        out[s] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}
// img[s][((g * 14 + p) * 64 + lane) * 8 + j]: piece s of the weights in the operand order of vx_expand_wimg_bf16_k
template <int NS>
__global__ void __launch_bounds__(256) vx_expand_wimg_split_k(const float* __restrict__ w, uint32_t* __restrict__ img, int groups, int backward) {
    const long n = (long)groups * 14 * 64 * 4;               // packed pairs per piece
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int jp = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long t = e >> 8;
    const int p = (int)(t % 14), g = (int)(t / 14);
    const int r = lane & 15, q = lane >> 4;
    const int tap = 2 * p + (q >> 1);
    float v[2] = {0.f, 0.f};
    if (tap < 27)
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * jp + u;
            int co, ci;
            if (backward) { co = 16 * g + 4 * (2 * (q & 1) + (j >> 2)) + (j & 3); ci = r; }
            else { co = 16 * g + r; ci = 8 * (q & 1) + j; }
            v[u] = w[((long)co * 16 + ci) * 27 + tap];
        }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 h = {(__bf16)v[0], (__bf16)v[1]};
        img[(long)s * n + e] = __builtin_bit_cast(uint32_t, h);
        v[0] -= (float)h[0]; v[1] -= (float)h[1];
    }
}
// ---- two scaled fp16 pieces per operand (ns = 22; the scheme of csrc/jlc_mfma.hip): x * 2^e = h0 + h1, 11 + 11 significant bits, a product = three MFMAs (h0 k0 + h0 k1 +
// h1 k0); e per staged tile (activations / gradients) and per weight tensor, undone on the fp32 accumulators (exact powers of two)
typedef _Float16 vx_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ vx_h8 vx_as_h8(uint4 v) { return __builtin_bit_cast(vx_h8, v); }
__device__ __forceinline__ uint32_t vx_pack_h2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float vx_h2_lo(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[0]; }
__device__ __forceinline__ float vx_h2_hi(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[1]; }
__device__ __forceinline__ int vx_exp16(float m) {              // |m| * 2^e in [2^14, 2^15); 0 for m = 0 / non-finite
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 0;
    const int e = 14 - ilogbf(m);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
// largest magnitude of a whole weight tensor: the blocks' maxima meet in out[1] through atomicMax on the bit pattern (non-negative floats order like unsigned integers;
// the host zeroes the word first).  Consumers take the scale exponent from it: vx_expand_wexp(ew).  (One block of 1024 threads walked the whole tensor before: 12 us for the
// 110 K weights of the 128^3 two-modality decoders, 30 us for BraTS' -- on every decoder's forward and backward chain.)
// (round 6) the blocks' maxima are left in part[blockIdx.x] (out + 4 ...) and met by the consumer -- vx_expand_wimg_f16_k, the next launch on the stream -- which also stores
// the tensor's maximum into out[1] for the matrix kernel: no atomics, hence no memset node in front (one 6 us launch less on every decoder's forward chain)
__global__ void __launch_bounds__(256) vx_expand_wmax_k(const float* __restrict__ w, long n, float* __restrict__ out) {
    __shared__ float sm[4];
    float mx = 0.0f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 t = reinterpret_cast<const float4*>(w)[i];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))));
    }
    if (blockIdx.x == 0)
        for (long i = 4 * n4 + threadIdx.x; i < n; i += 256) mx = fmaxf(mx, fabsf(w[i]));
    mx = vx_wave_max(mx);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) out[4 + blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__device__ __forceinline__ int vx_expand_wexp(const float* __restrict__ ew) { return vx_exp16(__uint_as_float(reinterpret_cast<const unsigned*>(ew)[1])); }
// returns the number of partial maxima left at ew + 4 (<= 64): the argument `nparts` of vx_expand_wimg_f16_k
static inline int vx_expand_wmax_launch(const float* w, long n, float* ew, hipStream_t st) {
    static const bool ms = getenv("VELOXSEG_EXPAND_WMAX_MEMSET") && getenv("VELOXSEG_EXPAND_WMAX_MEMSET")[0] == '1';      // (A/B: the memset node of rounds 4 - 5 back in front; results unchanged)
    if (ms && hipMemsetAsync(ew, 0, 2 * sizeof(float), st) != hipSuccess) return 1;
    int nb = (int)((n / 4 + 255) / 256);
    if (nb > 64) nb = 64;
    if (nb < 1) nb = 1;
    vx_expand_wmax_k<<<nb, 256, 0, st>>>(w, n, ew);
    return nb;
}
// the operand-order weight image as two scaled fp16 pieces: img[s][...] as vx_expand_wimg_split_k<2>
// nparts > 0: the tensor's maximum is met here from vx_expand_wmax_k's partial maxima (ew_out + 4 ...) and stored to ew_out[1] by block 0; nparts == 0: ew holds it already
__global__ void __launch_bounds__(256) vx_expand_wimg_f16_k(const float* __restrict__ w, uint32_t* __restrict__ img, const float* __restrict__ ew, int groups, int backward,
                                                            int nparts, float* __restrict__ ew_out) {
    const long n = (long)groups * 14 * 64 * 4;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    int wexp;
    if (nparts > 0) {
        float mx = 0.0f;
        for (int i = 0; i < nparts; ++i) mx = fmaxf(mx, ew_out[4 + i]);
        wexp = vx_exp16(mx);
        if (e == 0) { reinterpret_cast<unsigned*>(ew_out)[0] = 0u; reinterpret_cast<unsigned*>(ew_out)[1] = __float_as_uint(mx); }
    } else wexp = vx_expand_wexp(ew);
    if (e >= n) return;
    const float sc = ldexpf(1.0f, wexp);
    const int jp = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long t = e >> 8;
    const int p = (int)(t % 14), g = (int)(t / 14);
    const int r = lane & 15, q = lane >> 4;
    const int tap = 2 * p + (q >> 1);
    float v[2] = {0.f, 0.f};
    if (tap < 27)
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * jp + u;
            int co, ci;
            if (backward) { co = 16 * g + 4 * (2 * (q & 1) + (j >> 2)) + (j & 3); ci = r; }
            else { co = 16 * g + r; ci = 8 * (q & 1) + j; }
            v[u] = w[((long)co * 16 + ci) * 27 + tap] * sc;
        }
    const uint32_t h0 = vx_pack_h2(v[0], v[1]);
    img[e] = h0;
    img[n + e] = vx_pack_h2(v[0] - vx_h2_lo(h0), v[1] - vx_h2_hi(h0));
}
// 8 scaled floats -> two fp16 pieces
__device__ __forceinline__ void vx_split8_h(const float (&v)[8], float sc, uint4 (&out)[2]) {
    uint32_t p0[4], p1[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        const float a = v[2 * j2] * sc, b = v[2 * j2 + 1] * sc;
        p0[j2] = vx_pack_h2(a, b);
        p1[j2] = vx_pack_h2(a - vx_h2_lo(p0[j2]), b - vx_h2_hi(p0[j2]));
    }
    out[0] = make_uint4(p0[0], p0[1], p0[2], p0[3]);
    out[1] = make_uint4(p1[0], p1[1], p1[2], p1[3]);
}
// block-wide maximum through `scratch` (>= 4 floats of LDS; 256 threads); contains a barrier
__device__ __forceinline__ float vx_block_max_256(float m, float* __restrict__ scratch) {
    m = vx_wave_max(m);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// the piece products that are kept, smallest first: (activation piece, weight piece)
template <int NS> struct VxSplitTerms;
template <> struct VxSplitTerms<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int W[1] = {0}; };      // plain bf16 operands (the bf16 opt-in mode's weight gradient)
template <> struct VxSplitTerms<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int W[3] = {0, 1, 0}; };
template <> struct VxSplitTerms<3> { static constexpr int N = 6; static constexpr int A[6] = {1, 2, 0, 1, 0, 0}; static constexpr int W[6] = {1, 0, 2, 0, 1, 0}; };

template <int NS, bool F16 = false>
__global__ void __launch_bounds__(256) vx_expand_fwd_split_k(const float* __restrict__ x, const uint4* __restrict__ wimg, const float* __restrict__ bias,
                                                             float* __restrict__ y, int B, int Cc, int D, int H, int W, const float* __restrict__ ew = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint4 vx_xs[];          // [NS][channel half][6 x 6 x 18 halo voxel] x 8 bf16
    using TT = VxSplitTerms<NS>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;
    int tile = vx_xcd_tile(blockIdx.x, gridDim.x);          // (round 6: every XCD walks a contiguous run of tiles -- halo overlap from its own L2)
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const float* __restrict__ xb = x + (long)b * 16 * V;
    float fscale = 1.0f;                                                   // fp16 mode: 2^-(tile exponent + weight exponent), applied to the accumulators
    if constexpr (F16) {
        // every entry of the thread first (6 x 8 floats), the tile's largest magnitude over the block, then the scaled split
        float v[6][8];
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = (int)threadIdx.x + i * 256;
            const int ec = e < 2 * 648 ? e : 0;
            const int hv = ec % 648, half = ec / 648;
            const int hw = hv % 18, hh = (hv / 18) % 6, hd = hv / 108;
            const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + hw;
            const bool ok = e < 2 * 648 && (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
            const float* __restrict__ src = xb + (long)(8 * half) * V + (ok ? ((long)qd * H + qh) * W + qw : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t_ = src[(long)j * V]; v[i][j] = ok ? t_ : 0.0f; mx = fmaxf(mx, fabsf(v[i][j])); }
        }
        float* scratch = reinterpret_cast<float*>(vx_xs + NS * (2 * 648));
        const int ex = vx_exp16(vx_block_max_256(mx, scratch));
        const float sc = ldexpf(1.0f, ex);
        fscale = ldexpf(1.0f, -(ex + vx_expand_wexp(ew)));
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = (int)threadIdx.x + i * 256;
            if (e < 2 * 648) {
                uint4 pk[2];
                vx_split8_h(v[i], sc, pk);
                vx_xs[e] = pk[0];
                vx_xs[2 * 648 + e] = pk[1];
            }
        }
    } else {
    for (int e = threadIdx.x; e < 2 * 648; e += 256) {
        const int hv = e % 648, half = e / 648;
        const int hw = hv % 18, hh = (hv / 18) % 6, hd = hv / 108;
        const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + hw;
        const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
        const float* __restrict__ src = xb + (long)(8 * half) * V + (ok ? ((long)qd * H + qh) * W + qw : 0);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t_ = src[(long)j * V]; v[j] = ok ? t_ : 0.0f; }
        uint4 pk[NS];
        vx_split8<NS>(v, pk);
#pragma unroll
        for (int s = 0; s < NS; ++s) vx_xs[s * (2 * 648) + e] = pk[s];
    }
    }
    __syncthreads();
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const long wn = (long)Cc * 4 * 14 * 64;                                // uint4 per weight piece
    for (int g = 0; g < Cc * 4; ++g) {                                     // (c, s1) groups
        const int c = g >> 2, s1 = g & 3;
        const int co_base = g * 16;
        vx_f4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f};
        const uint4* __restrict__ wg = wimg + (long)g * 14 * 64 + lane;
#pragma unroll 2
        for (int p = 0; p < 14; ++p) {
            const int t = 2 * p + (q >> 1);
            const int tt = t < 27 ? t : 26;                                // (the empty half pair has zero weights)
            const int tw = tt % 3, th = (tt / 3) % 3, td = tt / 9;
            uint4 av[NS], bv[NS][4];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                av[s] = wg[s * wn + p * 64];
                const uint4* __restrict__ xt = vx_xs + s * (2 * 648) + (q & 1) * 648 + ((wave + td) * 6 + th) * 18 + r + tw;
#pragma unroll
                for (int m = 0; m < 4; ++m) bv[s][m] = xt[m * 18];
            }
#pragma unroll
            for (int k = 0; k < TT::N; ++k)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr (F16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vx_as_h8(av[TT::W[k]]), vx_as_h8(bv[TT::A[k]][m]), acc[m], 0, 0, 0);
                    else acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vx_as_bf8(av[TT::W[k]]), vx_as_bf8(bv[TT::A[k]][m]), acc[m], 0, 0, 0);
                }
        }
        if constexpr (F16) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] *= fscale;
        }
        const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + co_base + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* __restrict__ yb = y + ((long)b * Cc + c) * fplane + ((long)(4 * (d0 + wave) + s1) * FH + q) * FW + 4 * (w0 + r);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (w0 + r < W) *reinterpret_cast<float4*>(yb + (long)(4 * (h0 + m)) * FW) = make_float4(acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w);
    }
}

// 4 floats -> NS pieces of 4 bf16 (8 bytes each)
template <int NS>
__device__ __forceinline__ void vx_split4(const float4 v, uint2 (&out)[NS]) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 h0 = {(__bf16)r[0], (__bf16)r[1]}, h1 = {(__bf16)r[2], (__bf16)r[3]};
        out[s] = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
        if (s + 1 < NS) { r[0] -= (float)h0[0]; r[1] -= (float)h0[1]; r[2] -= (float)h1[0]; r[3] -= (float)h1[1]; }
    }
}
// input gradient: the halo of the fine gradient is split into its NS bf16 pieces ONCE, while it is staged ([piece][(hd, hh, s2) row][18 coarse voxels] x 4 bf16);
// the A operand of a tap pair is then two 8-byte LDS reads per piece (splitting it again for every tap cost more VALU time than the MFMAs saved)
template <int NS, bool F16 = false>
__global__ void __launch_bounds__(256) vx_expand_bwd_data_split_k(const float* __restrict__ dyf, const uint4* __restrict__ wimg, float* __restrict__ dx,
                                                                  int B, int Cc, int D, int H, int W, int accumulate, const float* __restrict__ ew = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint2 vx_hs[];          // [NS][144 rows][18]
    using TT = VxSplitTerms<NS>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    const int nTw = (W + 15) / 16, nTh = H / 4, nTd = D / 4;
    int tile = vx_xcd_tile(blockIdx.x, gridDim.x);
    const int tw_i = tile % nTw; tile /= nTw;
    const int th_i = tile % nTh; tile /= nTh;
    const int td_i = tile % nTd;
    const int b = tile / nTd;
    const int d0 = td_i * 4, h0 = th_i * 4, w0 = tw_i * 16;
    const long V = (long)D * H * W;
    const long FH = 4L * H, FW = 4L * W;
    const long fplane = (4L * D) * FH * FW;
    const float* __restrict__ dyb = dyf + (long)b * Cc * fplane;
    const long wn = (long)Cc * 4 * 14 * 64;
    vx_f4 acc[4];
    vx_f4 tot[4];                      // fp16 mode: the sum over the (c, s1) groups in true units (every group's staged tile has its own scale)
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f}; tot[m] = (vx_f4){0.f, 0.f, 0.f, 0.f}; }
    // the staging list of a thread (11 quads of the 144 x 18 halo tile) does not depend on (c, s1): element offsets inside a channel's fine volume at s1 = 0 and the
    // inside-the-volume flags, once (round 5: the index arithmetic -- two divisions by constants and a 64-bit multiply-add per quad, 65 quarter-rate multiplies per
    // (c, s1) iteration -- was a third of the kernel's issue slots)
    int soff[11];
    unsigned okm = 0;
#pragma unroll
    for (int u = 0; u < 11; ++u) {
        const int e = min((int)threadIdx.x + u * 256, 144 * 18 - 1);
        const int f4 = e % 18, row = e / 18;
        const int s2 = row & 3, hh = (row >> 2) % 6, hd = row / 24;
        const int qd = d0 - 1 + hd, qh = h0 - 1 + hh, qw = w0 - 1 + f4;
        const bool ok = (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W;
        soff[u] = ok ? (int)(((long)(4 * qd) * FH + 4 * qh + s2) * FW + 4 * qw) : 0;          // (< 2^31: a channel's fine volume is at most 2^31 elements where this kernel is used)
        okm |= (ok ? 1u : 0u) << u;
    }
    const long s1stride = FH * FW;
    for (int c = 0; c < Cc; ++c) {
        for (int s1 = 0; s1 < 4; ++s1) {
            __syncthreads();
            float gscale = 1.0f;
            {
                float4 v[11];
                const float* __restrict__ cb_ = dyb + (long)c * fplane + (long)s1 * s1stride;
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const bool ok = (okm >> u) & 1u;
                    const float4 t_ = *reinterpret_cast<const float4*>(ok ? cb_ + soff[u] : dyb);
                    v[u] = ok ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if constexpr (F16) {
                    float mx = 0.0f;
#pragma unroll
                    for (int u = 0; u < 11; ++u) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));     // (clamped duplicates repeat a valid entry)
                    const int ex = vx_exp16(vx_block_max_256(mx, reinterpret_cast<float*>(vx_hs + NS * (144 * 18))));
                    const float sc = ldexpf(1.0f, ex);
                    gscale = ldexpf(1.0f, -(ex + vx_expand_wexp(ew)));
#pragma unroll
                    for (int u = 0; u < 11; ++u) {
                        const int e = (int)threadIdx.x + u * 256;
                        if (e < 144 * 18) {
                            const float a0 = v[u].x * sc, a1 = v[u].y * sc, a2 = v[u].z * sc, a3 = v[u].w * sc;
                            const uint32_t l0 = vx_pack_h2(a0, a1), h0 = vx_pack_h2(a2, a3);
                            vx_hs[e] = make_uint2(l0, h0);
                            vx_hs[144 * 18 + e] = make_uint2(vx_pack_h2(a0 - vx_h2_lo(l0), a1 - vx_h2_hi(l0)), vx_pack_h2(a2 - vx_h2_lo(h0), a3 - vx_h2_hi(h0)));
                        }
                    }
                } else {
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int e = (int)threadIdx.x + u * 256;
                    if (e < 144 * 18) {
                        uint2 pc[NS];
                        vx_split4<NS>(v[u], pc);
#pragma unroll
                        for (int s = 0; s < NS; ++s) vx_hs[s * (144 * 18) + e] = pc[s];
                    }
                }
                }
            }
            __syncthreads();
            const uint4* __restrict__ wg = wimg + (long)(c * 4 + s1) * 14 * 64 + lane;
#pragma unroll 2
            for (int p = 0; p < 14; ++p) {
                const int t = 2 * p + (q >> 1);
                const int tt = t < 27 ? t : 26;
                const int tw = tt % 3, th = (tt / 3) % 3, td = tt / 9;
                uint4 bv[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) bv[s] = wg[s * wn + p * 64];
                const int hd = wave - td + 2;
                const int s2 = 2 * (q & 1);
                const uint2* __restrict__ ht = vx_hs + ((hd * 6 + (2 - th)) * 4 + s2) * 18 + (r - tw + 2);
                uint4 av[4][NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const uint2 lo = ht[s * (144 * 18) + m * 4 * 18], hi = ht[s * (144 * 18) + m * 4 * 18 + 18];
                        av[m][s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
#pragma unroll
                for (int k = 0; k < TT::N; ++k)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if constexpr (F16) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vx_as_h8(av[m][TT::A[k]]), vx_as_h8(bv[TT::W[k]]), acc[m], 0, 0, 0);
                        else acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vx_as_bf8(av[m][TT::A[k]]), vx_as_bf8(bv[TT::W[k]]), acc[m], 0, 0, 0);
                    }
            }
            if constexpr (F16) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { tot[m] += acc[m] * gscale; acc[m] = (vx_f4){0.f, 0.f, 0.f, 0.f}; }
            }
        }
    }
    if constexpr (F16) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = tot[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const long pbase = ((long)(d0 + wave) * H + (h0 + m)) * W + w0 + 4 * q;
        float* dst = dx + ((long)b * 16 + r) * V + pbase;
        if (w0 + 4 * q >= W) continue;
        float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        if (accumulate) { const float4 old = *reinterpret_cast<float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
    }
}

// floats of wt_ws for NS pieces: NS operand-order bf16 images of (Cc * 4) groups x 14 tap pairs x 64 lanes x 16 bytes
// (ns = 22: two scaled fp16 pieces + one float for the weight tensor's scale exponent)
extern "C" int vx_expand_split_ws_floats(int Cc, int ns) { return (Cc <= 0 || !(ns == 2 || ns == 3 || ns == 22)) ? -1 : Cc * 4 * 14 * 64 * 4 * (ns == 22 ? 2 : ns) + (ns == 22 ? 4 + 64 : 0); }      // (+ the scale word and the 64 partial maxima)

// returns 1 when the shape is not covered (the caller uses the fp32 MFMA kernels), 0 on success.  ns = 2 (3 products) or 3 (6 products: fp32-exact products)
extern "C" int vx_expand_fwd_mfma_split(const float* x, const float* w, const float* bias, float* wt_ws, float* y, int B, int Cc, int D, int H, int W, int ns, void* stream) {
    VX_REQUIRE(x && w && wt_ws && y && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0 && (ns == 2 || ns == 3 || ns == 22), "vx_expand_fwd_mfma_split: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int groups = Cc * 4;
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    const size_t shm = (size_t)(ns == 22 ? 2 : ns) * 2 * 648 * sizeof(uint4) + (ns == 22 ? 32 : 0);
    if (ns == 22) {
        float* ew = wt_ws + (long)groups * 14 * 64 * 4 * 2;
        const int nparts = vx_expand_wmax_launch(w, (long)groups * 16 * 16 * 27, ew, st);
        vx_expand_wimg_f16_k<<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), ew, groups, 0, nparts, ew);
        vx_expand_fwd_split_k<2, true><<<dim3((unsigned)nblk), 256, shm, st>>>(x, reinterpret_cast<const uint4*>(wt_ws), bias, y, B, Cc, D, H, W, ew);
    } else if (ns == 2) {
        vx_expand_wimg_split_k<2><<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 0);
        vx_expand_fwd_split_k<2><<<dim3((unsigned)nblk), 256, shm, st>>>(x, reinterpret_cast<const uint4*>(wt_ws), bias, y, B, Cc, D, H, W);
    } else {
        vx_expand_wimg_split_k<3><<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 0);
        vx_expand_fwd_split_k<3><<<dim3((unsigned)nblk), 256, shm, st>>>(x, reinterpret_cast<const uint4*>(wt_ws), bias, y, B, Cc, D, H, W);
    }
    VX_LAUNCH_CHECK("vx_expand_fwd_mfma_split");
    return 0;
}
// ---- (round 6) the two weight images of a layer built AHEAD of its forward (fp16-piece mode, ns = 22; engine.TrainEngine does it on a side lane at the head of the step):
// vx_expand_prep_split22 fills wt_fwd (image + scale word) and wt_bwd (the input gradient's image; same scale word, read from wt_fwd); the *_prepared entries launch the
// matrix kernels only.  Workspaces: vx_expand_split_ws_floats(Cc, 22) floats each.
extern "C" int vx_expand_prep_split22(const float* w, float* wt_fwd, float* wt_bwd, int Cc, void* stream) {
    VX_REQUIRE(w && wt_fwd && wt_bwd && Cc > 0, "vx_expand_prep_split22: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int groups = Cc * 4;
    float* ew = wt_fwd + (long)groups * 14 * 64 * 4 * 2;
    const int nparts = vx_expand_wmax_launch(w, (long)groups * 16 * 16 * 27, ew, st);
    const unsigned nb = (unsigned)vx_cdiv((long)groups * 14 * 64 * 4, 256);
    vx_expand_wimg_f16_k<<<nb, 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_fwd), ew, groups, 0, nparts, ew);
    vx_expand_wimg_f16_k<<<nb, 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_bwd), ew, groups, 1, 0, nullptr);
    VX_LAUNCH_CHECK("vx_expand_prep_split22");
    return 0;
}
extern "C" int vx_expand_fwd_mfma_split_prepared(const float* x, const float* bias, const float* wt_fwd, float* y, int B, int Cc, int D, int H, int W, void* stream) {
    VX_REQUIRE(x && wt_fwd && y && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_fwd_mfma_split_prepared: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    const int groups = Cc * 4;
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    const size_t shm = (size_t)2 * 2 * 648 * sizeof(uint4) + 32;
    vx_expand_fwd_split_k<2, true><<<dim3((unsigned)nblk), 256, shm, (hipStream_t)stream>>>(x, reinterpret_cast<const uint4*>(wt_fwd), bias, y, B, Cc, D, H, W, wt_fwd + (long)groups * 14 * 64 * 4 * 2);
    VX_LAUNCH_CHECK("vx_expand_fwd_mfma_split_prepared");
    return 0;
}
extern "C" int vx_expand_bwd_data_mfma_split_prepared(const float* dy_fine, const float* wt_bwd, const float* ew_fwd, float* dx, int B, int Cc, int D, int H, int W, int accumulate,
                                                      void* stream) {
    VX_REQUIRE(dy_fine && wt_bwd && ew_fwd && dx && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0, "vx_expand_bwd_data_mfma_split_prepared: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    const size_t shm = (size_t)2 * 144 * 18 * sizeof(uint2) + 32;
    vx_expand_bwd_data_split_k<2, true><<<dim3((unsigned)nblk), 256, shm, (hipStream_t)stream>>>(dy_fine, reinterpret_cast<const uint4*>(wt_bwd), dx, B, Cc, D, H, W, accumulate, ew_fwd);
    VX_LAUNCH_CHECK("vx_expand_bwd_data_mfma_split_prepared");
    return 0;
}
extern "C" int vx_expand_bwd_data_mfma_split_ew(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int ns,
                                                const float* ew_fwd, void* stream);
extern "C" int vx_expand_bwd_data_mfma_split(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int ns,
                                             void* stream) {
    return vx_expand_bwd_data_mfma_split_ew(dy_fine, w, wt_ws, dx, B, Cc, D, H, W, accumulate, ns, nullptr, stream);
}
// the scale word of the weight tensor as the forward of the same layer left it (vx_expand_split_ew_offset floats into ITS workspace; the weights have not changed since):
// ns = 22 then launches neither the memset nor vx_expand_wmax_k -- two 6 us launches per decoder off the backward chain (round 6).  ew_fwd null: found here.
extern "C" int vx_expand_split_ew_offset(int Cc) { return Cc * 4 * 14 * 64 * 4 * 2; }
extern "C" int vx_expand_bwd_data_mfma_split_ew(const float* dy_fine, const float* w, float* wt_ws, float* dx, int B, int Cc, int D, int H, int W, int accumulate, int ns,
                                                const float* ew_fwd, void* stream) {
    VX_REQUIRE(dy_fine && w && wt_ws && dx && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0 && (ns == 2 || ns == 3 || ns == 22), "vx_expand_bwd_data_mfma_split: bad args");
    if (D % 4 != 0 || H % 4 != 0 || W % 4 != 0) return 1;
    hipStream_t st = (hipStream_t)stream;
    const int groups = Cc * 4;
    const long nblk = (long)B * (D / 4) * (H / 4) * ((W + 15) / 16);
    const size_t shm = (size_t)(ns == 22 ? 2 : ns) * 144 * 18 * sizeof(uint2) + (ns == 22 ? 32 : 0);
    if (ns == 22) {
        float* ew_own = wt_ws + (long)groups * 14 * 64 * 4 * 2;
        const int nparts = ew_fwd == nullptr ? vx_expand_wmax_launch(w, (long)groups * 16 * 16 * 27, ew_own, st) : 0;
        const float* ew = ew_fwd ? ew_fwd : ew_own;
        vx_expand_wimg_f16_k<<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), ew, groups, 1, nparts, ew_own);
        vx_expand_bwd_data_split_k<2, true><<<dim3((unsigned)nblk), 256, shm, st>>>(dy_fine, reinterpret_cast<const uint4*>(wt_ws), dx, B, Cc, D, H, W, accumulate, ew);
    } else if (ns == 2) {
        vx_expand_wimg_split_k<2><<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 1);
        vx_expand_bwd_data_split_k<2><<<dim3((unsigned)nblk), 256, shm, st>>>(dy_fine, reinterpret_cast<const uint4*>(wt_ws), dx, B, Cc, D, H, W, accumulate);
    } else {
        vx_expand_wimg_split_k<3><<<vx_cdiv((long)groups * 14 * 64 * 4, 256), 256, 0, st>>>(w, reinterpret_cast<uint32_t*>(wt_ws), groups, 1);
        vx_expand_bwd_data_split_k<3><<<dim3((unsigned)nblk), 256, shm, st>>>(dy_fine, reinterpret_cast<const uint4*>(wt_ws), dx, B, Cc, D, H, W, accumulate);
    }
    VX_LAUNCH_CHECK("vx_expand_bwd_data_mfma_split");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix pipe with fp32-exact products (the split of the kernels above):
//   dW[co, ci, t] += sum_{b,p} dyf[co, p] * x[ci, p + t - 1]      (+ db[co] += sum dyf[co, p])
//   one MFMA = a whole coarse row: rows = the 16 output channels (s2, s3) of one (c, s1) group, cols = 16 input channels, K = 32 coarse voxels along W.
//   block = (channel block c, tap plane td) x a strip of coarse rows (b, d, h0 .. h0+HL-1, 32-wide W chunk); wave = s1: 9 accumulator tiles per wave,
//   three blocks per CU (the staging of one hides behind the MFMAs of the others).
//   B operand (x): the rows h-1 .. h+1 of plane d+td-1 live in LDS as bf16 pieces, [piece][row slot (4, ring)][ci][32 voxels], split ONCE when a row is
//                  staged and read UNSHIFTED (one aligned 16-byte read per piece and row).
//   A operand (fine gradient): lane (r = (s2, s3), q) owns fine[c][4d+s1][4h+s2][4(8q+j)+s3], j = -1..8 (10 dword loads per row, prefetched one row ahead);
//                  the +-1 shift along W of a tap is applied HERE, once per row: dW[tw] = sum_w' dy[w' - tw + 1] x[w'], the three shifted operand vectors of a
//                  piece are 5 funnel shifts (v_alignbit) of its packed pairs.  (Shifting x instead cost 72 funnel shifts per row and wave -- every
//                  (row, piece) read three ways -- and left the kernel VALU-bound at twice the MFMA time.)
//   Flush: a block's tiles go through LDS into ITS row of a partial-sum workspace (plain stores); vx_expand_wgrad_fold_k adds the rows into dW in a fixed
//   order, so the weight gradient is reproducible run to run.  (History: a wave that owned all 27 taps flushed 55 K sums per CU -- 14 M float atomics,
//   38 of 107 us; atomics in runs of 9 floats into the (co, ci, 27) layout were 8 x slower per element than full-line ones: hence the workspace rows.)
// ------------------------------------------------------------------------------------------------------------------
#define VX_WS_PITCH 20                 // dwords per (piece, slot, ci) row image: 32 bf16 + 8 bytes (bank spread)
template <int NS, typename TD = float>
__global__ void __launch_bounds__(256, 3) vx_expand_wgrad_split_k(const float* __restrict__ x, const TD* __restrict__ dyf, float* __restrict__ part, float* __restrict__ db,
                                                                   int B, int Cc, int D, int H, int W, int HL, int nHs, int nWc, int nUnits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t vx_wx[];       // [NS][4][16][VX_WS_PITCH], then the flush image [4][16][16][9]
    using TT = VxSplitTerms<NS>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 15, q = lane >> 4;
    // block id -> (unit = (strip, c), plane): the three planes of a unit are neighbours in dispatch order on the same XCD (ids 8 apart), so the fine
    // gradient rows they share are in that XCD's L2 for the second and third reader
    const int xcd = blockIdx.x & 7, m_ = blockIdx.x >> 3;
    const int pd = m_ % 3, unit = (m_ / 3) * 8 + xcd;
    if (unit >= nUnits) return;
    const int c = unit % Cc;
    int strip = unit / Cc;
    const int strip_id = strip;
    const int wc = strip % nWc; strip /= nWc;
    const int hs = strip % nHs; strip /= nHs;
    const int d = strip % D;
    const int b = strip / D;
    const int dq = d - 1 + pd;
    if ((unsigned)dq >= (unsigned)D) return;                              // this plane is the zero padding: nothing to add (the fold kernel skips the row)
    const int h_lo = hs * HL, h_hi = (h_lo + HL < H) ? h_lo + HL : H, wc0 = wc * 32;
    const long V = (long)D * H * W;
    const float* __restrict__ xb = x + (long)b * 16 * V + (long)dq * H * W;
    // ---- staging of one coarse row (16 channels x 32 voxels): threads 0..127, one float4 each ----
    float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
    const int st_ci = (threadIdx.x >> 3) & 15, st_seg = threadIdx.x & 7;
    auto fetch = [&](int hq) {
        if (threadIdx.x < 128) {
            const int w = wc0 + 4 * st_seg;
            pf = ((unsigned)hq < (unsigned)H && w < W) ? *reinterpret_cast<const float4*>(xb + (long)st_ci * V + (long)hq * W + w) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int hq) {
        if (threadIdx.x < 128) {
            const int slot = hq & 3;
            uint2 pk[NS];
            vx_split4<NS>(pf, pk);
#pragma unroll
            for (int s = 0; s < NS; ++s) *reinterpret_cast<uint2*>(vx_wx + (((s * 4 + slot) * 16 + st_ci) * VX_WS_PITCH + 2 * st_seg)) = pk[s];
        }
    };
    // ---- the fine gradient of this wave's group: lane (r, q), voxels 8q-1 .. 8q+8 ----
    const int s1 = wave, s2 = r >> 2, s3 = r & 3;
    const long FH = 4L * H, FW = 4L * W;
    const TD* __restrict__ dyb = dyf + (((long)b * Cc + c) * (4L * D) + 4 * d + s1) * FH * FW + (long)s2 * FW + 4L * (wc0 + 8 * q) + s3;
    float dv[10];
    auto fetch_dy = [&](int h) {
        const TD* __restrict__ p = dyb + 4L * h * FW;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int w = wc0 + 8 * q + j - 1;
            dv[j] = (h < h_hi && (unsigned)w < (unsigned)W) ? vx_ld1(p, 4L * (j - 1)) : 0.0f;
        }
    };
    vx_f4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (vx_f4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.0f;
    for (int hq = h_lo - 1; hq <= h_lo + 1; ++hq) { fetch(hq); stash(hq); }      // prologue: rows h_lo-1, h_lo, h_lo+1
    fetch_dy(h_lo);
    __syncthreads();
    for (int h = h_lo; h < h_hi; ++h) {
        // pieces of the 10 values, packed in pairs (0,1) .. (6,7) of the 8 own voxels; the neighbours go to the half a funnel shift takes them from
        uint4 a[NS][3];
        {
            float rr[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) rr[j] = dv[j];
#pragma unroll
            for (int j = 1; j < 9; ++j) bsum += dv[j];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                uint32_t p[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {
                    const bf2 hh = {(__bf16)rr[1 + 2 * j2], (__bf16)rr[2 + 2 * j2]};
                    p[j2] = __builtin_bit_cast(uint32_t, hh);
                    if (s + 1 < NS) { rr[1 + 2 * j2] -= (float)hh[0]; rr[2 + 2 * j2] -= (float)hh[1]; }
                }
                const bf2 he = {(__bf16)rr[9], (__bf16)rr[0]};                // low half: voxel 8q+8, high half: voxel 8q-1
                if (s + 1 < NS) { rr[9] -= (float)he[0]; rr[0] -= (float)he[1]; }
                const uint32_t ends = __builtin_bit_cast(uint32_t, he);
                const uint32_t s01 = __builtin_amdgcn_alignbit(p[1], p[0], 16), s12 = __builtin_amdgcn_alignbit(p[2], p[1], 16), s23 = __builtin_amdgcn_alignbit(p[3], p[2], 16);
                a[s][1] = make_uint4(p[0], p[1], p[2], p[3]);
                a[s][2] = make_uint4(__builtin_amdgcn_alignbit(p[0], ends, 16), s01, s12, s23);          // element j = dy[j - 1]
                a[s][0] = make_uint4(s01, s12, s23, __builtin_amdgcn_alignbit(ends, p[3], 16));          // element j = dy[j + 1]
            }
        }
        const bool more = h + 1 < h_hi;
        if (more) { fetch(h + 2); fetch_dy(h + 1); }
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            const int slot = (h - 1 + ph) & 3;
            uint4 bv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) bv[s] = *reinterpret_cast<const uint4*>(vx_wx + (((s * 4 + slot) * 16 + r) * VX_WS_PITCH + 4 * q));
#pragma unroll
            for (int tw = 0; tw < 3; ++tw)
#pragma unroll
                for (int k = 0; k < TT::N; ++k)
                    acc[ph * 3 + tw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vx_as_bf8(a[TT::W[k]][tw]), vx_as_bf8(bv[TT::A[k]]), acc[ph * 3 + tw], 0, 0, 0);
        }
        if (more) stash(h + 2);
        __syncthreads();
    }
    // ---- flush: red[wave][co][ci][9] through LDS, then one contiguous 36 KB row of the workspace ----
    float* __restrict__ red = reinterpret_cast<float*>(vx_wx);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wave * 2304 + ((4 * q + reg) * 16 + r) * 9 + t] = acc[t][reg];
    __syncthreads();
    float4* __restrict__ prow = reinterpret_cast<float4*>(part + (((long)strip_id * Cc + c) * 3 + pd) * 9216);
    for (int e = threadIdx.x; e < 2304; e += 256) prow[e] = reinterpret_cast<const float4*>(red)[e];
    if (db != nullptr && pd == 1) {                                       // (the centre plane exists for every d: one block per strip adds the bias sums)
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0) atomicAdd(db + (c * 4 + s1) * 16 + r, bsum);
    }
}
// dw[(c, wave) group][co][ci][9 pd + t] += sum over the strips whose plane d + pd - 1 exists of part[strip][c][pd][wave][co][ci][t]
// block = 16 float4 outputs x 16 parts of the strip list (summed through LDS in a fixed order)
__global__ void __launch_bounds__(256) vx_expand_wgrad_fold_k(const float* __restrict__ part, float* __restrict__ dw, int Cc, int D, int nStrips, int perD) {
    __shared__ float4 sm[16][16];
    const int i = threadIdx.x & 15, k = threadIdx.x >> 4;
    const int o4 = blockIdx.x * 16 + i;                                   // float4 index into (c, pd, 9216 / 4)
    const int total4 = Cc * 3 * 2304;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = 0, pd = 0, e4 = 0;
    if (o4 < total4) {
        e4 = o4 % 2304;
        pd = (o4 / 2304) % 3;
        c = o4 / (3 * 2304);
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(part) + ((long)c * 3 + pd) * 2304 + e4;
        const long stride4 = (long)Cc * 3 * 2304;
#pragma unroll 4
        for (int st = k; st < nStrips; st += 16) {
            const int d = (st / perD) % D;
            if ((unsigned)(d - 1 + pd) < (unsigned)D) {
                const float4 v = p4[(long)st * stride4];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
    }
    sm[k][i] = s;
    __syncthreads();
    if (k == 0 && o4 < total4) {
        float4 v = sm[0][i];
#pragma unroll
        for (int kk = 1; kk < 16; ++kk) { const float4 u = sm[kk][i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = 4 * e4 + u;
            const int wv = e / 2304, rem = e - wv * 2304, cc = rem / 9, t = rem - cc * 9;
            dw[((long)c * 4 + wv) * 6912 + cc * 27 + 9 * pd + t] += vv[u];
        }
    }
}

static void vx_wgs_plan(int B, int Cc, int D, int H, int W, int& HL, int& nHs, int& nWc) {
    nWc = (W + 31) / 32;
    const long base = (long)B * D * nWc * Cc * 3;
    static const long target = getenv("VELOXSEG_EXPAND_WG_BLOCKS") ? atol(getenv("VELOXSEG_EXPAND_WG_BLOCKS")) : 384;      // (round 6 A/B, same box: 768 -> 384 blocks for the one-channel decoders -- half the partial-sum rows -- autopet128 1051.6 -> 1056.4, autopet96 +0.4 %, hecktor -0.2 %; this kernel is a sink beside the encoder backward)
    int want = (int)((target + base - 1) / base);            // strips per (b, d, chunk, c, plane)
    if (want < 1) want = 1;
    HL = (H + want - 1) / want;
    if (HL < 8) HL = H < 8 ? H : 8;
    nHs = (H + HL - 1) / HL;
}
// floats of the partial-sum workspace of vx_expand_wgrad_mfma_split
extern "C" int vx_expand_wgrad_split_ws_floats(int B, int Cc, int D, int H, int W) {
    if (B <= 0 || Cc <= 0 || D <= 0 || H <= 0 || W <= 0) return -1;
    int HL, nHs, nWc;
    vx_wgs_plan(B, Cc, D, H, W, HL, nHs, nWc);
    const long n = (long)B * D * nHs * nWc * Cc * 3 * 9216;
    return n > 0x7fffffffL ? -1 : (int)n;
}
// returns 1 when the shape is not covered (the caller uses the fp32 MFMA kernel), 0 on success.  part_ws: vx_expand_wgrad_split_ws_floats floats
// dy_h16 != 0 (ns = 1 only): the fine gradient is a vx_bf16 array (bf16 storage mode)
extern "C" int vx_expand_wgrad_mfma_split_h(const float* x, const void* dy_fine_, float* dw, float* db, float* part_ws, long ws_floats, int B, int Cc, int D, int H, int W,
                                            int ns, int dy_h16, void* stream) {
    const float* dy_fine = (const float*)dy_fine_;
    VX_REQUIRE(x && dy_fine && dw && part_ws && B > 0 && Cc > 0 && D > 0 && H > 0 && W > 0 && ns >= 1 && ns <= 3, "vx_expand_wgrad_mfma_split: bad args");
    VX_REQUIRE(!dy_h16 || ns == 1, "vx_expand_wgrad_mfma_split: a 16-bit gradient needs plain bf16 operands (ns = 1)");
    if (W % 4 != 0) return 1;
    int HL, nHs, nWc;
    vx_wgs_plan(B, Cc, D, H, W, HL, nHs, nWc);
    const long nStrips = (long)B * D * nHs * nWc;
    VX_REQUIRE(ws_floats >= nStrips * Cc * 3 * 9216, "vx_expand_wgrad_mfma_split: workspace too small");
    const size_t shm = (size_t)ns * 4 * 16 * VX_WS_PITCH * sizeof(uint32_t);
    const size_t need = 4 * 2304 * sizeof(float);
    const size_t lds = shm > need ? shm : need;
    const int nUnits = (int)(nStrips * Cc);
    dim3 grid((unsigned)(((nUnits + 7) / 8) * 8 * 3));
    if (ns == 1 && dy_h16) vx_expand_wgrad_split_k<1, vx_bf16><<<grid, 256, lds, (hipStream_t)stream>>>(x, (const vx_bf16*)dy_fine_, part_ws, db, B, Cc, D, H, W, HL, nHs, nWc, nUnits);
    else if (ns == 1) vx_expand_wgrad_split_k<1><<<grid, 256, lds, (hipStream_t)stream>>>(x, dy_fine, part_ws, db, B, Cc, D, H, W, HL, nHs, nWc, nUnits);
    else if (ns == 2) vx_expand_wgrad_split_k<2><<<grid, 256, lds, (hipStream_t)stream>>>(x, dy_fine, part_ws, db, B, Cc, D, H, W, HL, nHs, nWc, nUnits);
    else vx_expand_wgrad_split_k<3><<<grid, 256, lds, (hipStream_t)stream>>>(x, dy_fine, part_ws, db, B, Cc, D, H, W, HL, nHs, nWc, nUnits);
    vx_expand_wgrad_fold_k<<<vx_cdiv((long)Cc * 3 * 2304, 16), 256, 0, (hipStream_t)stream>>>(part_ws, dw, Cc, D, (int)nStrips, nHs * nWc);
    VX_LAUNCH_CHECK("vx_expand_wgrad_mfma_split");
    return 0;
}
extern "C" int vx_expand_wgrad_mfma_split(const float* x, const float* dy_fine, float* dw, float* db, float* part_ws, long ws_floats, int B, int Cc, int D, int H, int W,
                                          int ns, void* stream) {
    return vx_expand_wgrad_mfma_split_h(x, dy_fine, dw, db, part_ws, ws_floats, B, Cc, D, H, W, ns, 0, stream);
}
