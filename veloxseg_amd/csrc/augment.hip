// Training-input stand-in on the GPU: foreground bounding box, label-driven crop centres, z-rotation.
// Reference call sites: utils/train_autopet.py:132-152 (MONAI CropForegroundd, RandCropByPosNegLabeld(pos=1, neg=1, num_samples=2),
// RandRotated(range_z = 15 deg, prob 0.5, bilinear image / nearest label)); utils/runtime.py:115-122.  MONAI is not in the image: semantics restated
// in oracle/augment_oracle.py (unpinned against MONAI itself).  All kernels are streaming passes (HBM-bound).
#include "vx_common.h"

// bbox of {x > thr}: out[6] = (min_d, min_h, min_w, max_d, max_h, max_w) as int32, initialised by the caller to (INT_MAX x3, -1 x3)
__global__ void __launch_bounds__(256) vx_bbox_gt_k(const float* __restrict__ x, float thr, int C, int D, int H, int W, int* __restrict__ out) {
    const long V = (long)D * H * W, n = (long)C * V;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (x[i] > thr) {
            const long v = i % V;
            const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
            lo[0] = min(lo[0], d); lo[1] = min(lo[1], h); lo[2] = min(lo[2], w);
            hi[0] = max(hi[0], d); hi[1] = max(hi[1], h); hi[2] = max(hi[2], w);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], o, 64)); hi[k] = max(hi[k], __shfl_xor(hi[k], o, 64)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { if (lo[k] != 0x7fffffff) atomicMin(out + k, lo[k]); if (hi[k] >= 0) atomicMax(out + 3 + k, hi[k]); }
    }
}

// global minimum of x (float) via int-ordered atomics: out[0] holds the float bits, initialised to +inf by the caller
__global__ void __launch_bounds__(256) vx_min_k(const float* __restrict__ x, long n, float* __restrict__ out) {
    float m = INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fminf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) {
        // float atomic min through compare-and-swap on the bit pattern
        unsigned int* a = reinterpret_cast<unsigned int*>(out);
        unsigned int old = *a;
        while (__uint_as_float(old) > m) {
            const unsigned int prev = atomicCAS(a, old, __float_as_uint(m));
            if (prev == old) break;
            old = prev;
        }
    }
}

// counts[chunk] = number of voxels in [chunk*CH, (chunk+1)*CH) whose label is (fg ? > 0 : == 0); labels as uint8 / int32 / int64 / fp32 (lab_bytes 1 / 4 / 8 / -4)
template <typename T>
__global__ void __launch_bounds__(256) vx_label_chunk_count_k(const T* __restrict__ lab, long n, int CH, int fg, int* __restrict__ counts) {
    const long c0 = (long)blockIdx.x * CH;
    int c = 0;
    for (long i = c0 + threadIdx.x; i < c0 + CH && i < n; i += 256) { const bool p = lab[i] > 0; c += (fg ? p : !p) ? 1 : 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __shared__ int sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// index of the k-th (0-based, in flat order) matching voxel inside one chunk; one block, serial over 256-wide strips with a block scan
template <typename T>
__global__ void __launch_bounds__(256) vx_label_kth_in_chunk_k(const T* __restrict__ lab, long n, long c0, int CH, int fg, int k, long* __restrict__ out) {
    __shared__ int sm[256];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (long s = c0; s < c0 + CH && s < n; s += 256) {
        const long i = s + threadIdx.x;
        int p = 0;
        if (i < n && i < c0 + CH) { const bool q = lab[i] > 0; p = (fg ? q : !q) ? 1 : 0; }
        sm[threadIdx.x] = p;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {              // inclusive Hillis-Steele scan
            const int v = threadIdx.x >= o ? sm[threadIdx.x - o] : 0;
            __syncthreads();
            sm[threadIdx.x] += v;
            __syncthreads();
        }
        const int incl = sm[threadIdx.x], b = base;
        if (p && b + incl - 1 == k) out[0] = i;
        __syncthreads();
        if (threadIdx.x == 255) base = b + incl;
        __syncthreads();
        if (base > k) return;
    }
}

// rotation about the LAST spatial axis (the (D, H) plane turns; every W column keeps its index): out[c, d, h, w] = sample(x[c], R(d - cd, h - ch) + (cd, ch), w),
// centre (n-1)/2, border padding (coordinates clamped), bilinear (mode 0) or nearest (mode 1).  (cs, sn) = cos / sin of the angle.
__global__ void __launch_bounds__(256) vx_rotate_z_k(const float* __restrict__ x, float* __restrict__ out, int C, int D, int H, int W, float cs, float sn, int mode) {
    const long V = (long)D * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)C * V) return;
    const long v = i % V;
    const int c = (int)(i / V);
    const int w = (int)(v % W), h = (int)((v / W) % H), d = (int)(v / ((long)W * H));
    const float cd = 0.5f * (float)(D - 1), ch = 0.5f * (float)(H - 1);
    const float pd = (float)d - cd, ph = (float)h - ch;
    float sd = cs * pd - sn * ph + cd, sh = sn * pd + cs * ph + ch;
    sd = fminf(fmaxf(sd, 0.0f), (float)(D - 1));
    sh = fminf(fmaxf(sh, 0.0f), (float)(H - 1));
    const float* __restrict__ xc = x + (long)c * V;
    if (mode == 1) {
        const int nd = (int)nearbyintf(sd), nh = (int)nearbyintf(sh);
        out[i] = xc[((long)nd * H + nh) * W + w];
        return;
    }
    const int d0 = (int)floorf(sd), h0 = (int)floorf(sh);
    const int d1 = min(d0 + 1, D - 1), h1 = min(h0 + 1, H - 1);
    const float ld = sd - (float)d0, lh = sh - (float)h0;
    const float v00 = xc[((long)d0 * H + h0) * W + w], v01 = xc[((long)d0 * H + h1) * W + w];
    const float v10 = xc[((long)d1 * H + h0) * W + w], v11 = xc[((long)d1 * H + h1) * W + w];
    out[i] = (1.0f - ld) * ((1.0f - lh) * v00 + lh * v01) + ld * ((1.0f - lh) * v10 + lh * v11);
}

static inline int vx_grid(long n) { long g = (n + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

extern "C" int vx_min_value(const float* x, long n, float* out_init_inf, void* stream) {
    VX_REQUIRE(x && out_init_inf && n > 0, "vx_min_value: bad args");
    int g = vx_grid(n); if (g > 2048) g = 2048;
    vx_min_k<<<dim3(g), dim3(256), 0, (hipStream_t)stream>>>(x, n, out_init_inf);
    VX_LAUNCH_CHECK("vx_min_value");
    return 0;
}

extern "C" int vx_bbox_gt(const float* x, float thr, int C, int D, int H, int W, int* out6, void* stream) {
    VX_REQUIRE(x && out6 && C > 0 && D > 0 && H > 0 && W > 0, "vx_bbox_gt: bad args");
    int g = vx_grid((long)C * D * H * W); if (g > 2048) g = 2048;
    vx_bbox_gt_k<<<dim3(g), dim3(256), 0, (hipStream_t)stream>>>(x, thr, C, D, H, W, out6);
    VX_LAUNCH_CHECK("vx_bbox_gt");
    return 0;
}

extern "C" int vx_label_chunk_count(const void* labels, int lab_bytes, long n, int chunk, int fg, int* counts, void* stream) {
    VX_REQUIRE(labels && counts && n > 0 && chunk >= 256, "vx_label_chunk_count: bad args");
    const int nb = (int)((n + chunk - 1) / chunk);
    hipStream_t st = (hipStream_t)stream;
    if (lab_bytes == 1) vx_label_chunk_count_k<unsigned char><<<dim3(nb), dim3(256), 0, st>>>((const unsigned char*)labels, n, chunk, fg, counts);
    else if (lab_bytes == 4) vx_label_chunk_count_k<int><<<dim3(nb), dim3(256), 0, st>>>((const int*)labels, n, chunk, fg, counts);
    else if (lab_bytes == 8) vx_label_chunk_count_k<long long><<<dim3(nb), dim3(256), 0, st>>>((const long long*)labels, n, chunk, fg, counts);
    else if (lab_bytes == -4) vx_label_chunk_count_k<float><<<dim3(nb), dim3(256), 0, st>>>((const float*)labels, n, chunk, fg, counts);
    else VX_FAIL(-3, "vx_label_chunk_count: label width %d bytes not supported", lab_bytes);
    VX_LAUNCH_CHECK("vx_label_chunk_count");
    return 0;
}

extern "C" int vx_label_kth_in_chunk(const void* labels, int lab_bytes, long n, long chunk_start, int chunk, int fg, int k, long* out_index, void* stream) {
    VX_REQUIRE(labels && out_index && n > 0 && chunk >= 256 && k >= 0 && chunk_start >= 0 && chunk_start < n, "vx_label_kth_in_chunk: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (lab_bytes == 1) vx_label_kth_in_chunk_k<unsigned char><<<dim3(1), dim3(256), 0, st>>>((const unsigned char*)labels, n, chunk_start, chunk, fg, k, out_index);
    else if (lab_bytes == 4) vx_label_kth_in_chunk_k<int><<<dim3(1), dim3(256), 0, st>>>((const int*)labels, n, chunk_start, chunk, fg, k, out_index);
    else if (lab_bytes == 8) vx_label_kth_in_chunk_k<long long><<<dim3(1), dim3(256), 0, st>>>((const long long*)labels, n, chunk_start, chunk, fg, k, out_index);
    else if (lab_bytes == -4) vx_label_kth_in_chunk_k<float><<<dim3(1), dim3(256), 0, st>>>((const float*)labels, n, chunk_start, chunk, fg, k, out_index);
    else VX_FAIL(-3, "vx_label_kth_in_chunk: label width %d bytes not supported", lab_bytes);
    VX_LAUNCH_CHECK("vx_label_kth_in_chunk");
    return 0;
}

extern "C" int vx_rotate_z(const float* x, float* out, int C, int D, int H, int W, float cos_a, float sin_a, int mode, void* stream) {
    VX_REQUIRE(x && out && x != out && C > 0 && D > 0 && H > 0 && W > 0 && (mode == 0 || mode == 1), "vx_rotate_z: bad args");
    const long n = (long)C * D * H * W;
    vx_rotate_z_k<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, out, C, D, H, W, cos_a, sin_a, mode);
    VX_LAUNCH_CHECK("vx_rotate_z");
    return 0;
}
