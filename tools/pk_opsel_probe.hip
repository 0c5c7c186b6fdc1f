// Stand-alone probe for the cause of the taped step's timing-dependent deviation: does a packed-fp32 add whose LOW half selects the HIGH register
// of its second source pair (v_pk_add_f32 ... op_sel:[0,1] -- what hipcc makes of `float4(acc0 + b, acc1 + b, acc2 + b, acc3 + b)` with b in an odd register of a
// ds_read2_b32 result) compute the right sum while waves of ANOTHER kernel that runs f16 MFMAs out of a large LDS allocation share the SIMD?
//   hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_probe.hip -o /tmp/pkp && /tmp/pkp
// victim_k:    per iteration ds_read2_b32 {b0, b1} from LDS, then  r = v_pk_add_f32 {a0, a1}, {b0, b1} op_sel:[0,1]  (expected {a0 + b1, a1 + b1}) and the control
//              form op_sel_hi:[1,0] (expected {a0 + b0, a1 + b0}); mismatches are counted per (form, half, 16-lane quarter).
// aggressor_k: the shape of csrc/conv_mfma.hip vx_stem_fwd_k -- 256 threads, ~77 KB of dynamic LDS, per step two ds_read_b64 pairs and three v_mfma_f32_16x16x32_f16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// F: 0 ds_read2_b32 + v_pk_add_f32 op_sel:[0,1] | 1 the same, b read from LDS once before the loop | 2 s_nop 15 between the LDS wait and the add | 3 v_pk_mul_f32
//    4 op_sel:[1,0] (the low half takes the HIGH register of the FIRST source) | 5 op_sel:[0,1] op_sel_hi:[1,0] (low half <- b1, high half <- b0) | 6 two ds_read_b32
//    7 ds_read_b64 | 8 b1 copied to an even register first (v_mov) and the add uses op_sel_hi:[1,0] on that -- the shape of a work-around
template <int F>
__global__ void __launch_bounds__(256) victim_k(const float* __restrict__ bias_g, unsigned* __restrict__ err, float* __restrict__ first, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bia = lds + 1312;                                  // where csrc/pwa_fused.hip vx_ln_pw_fwd_k<4, 4> keeps its biases (C = 16)
    for (int i = threadIdx.x; i < 1312; i += 256) lds[i] = (float)i * 0.001f;
    if (threadIdx.x < 48) bia[threadIdx.x] = bias_g[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4;
    unsigned bad[2] = {0, 0};
    f2 bpre = {bia[4 * q], bia[4 * q + 1]};
    for (int it = 0; it < iters; ++it) {
        const int m = F == 1 ? 4 * q : 4 * q + 16 * (it % 3);
        const unsigned addr = (unsigned)(uintptr_t)(bia + m);          // LDS byte address (low 32 bits of the generic pointer)
        const float a0 = 1.0f + (float)(it & 7), a1 = 0.5f + (float)lane;
        f2 a = {a0, a1}, b, r1;
        if (F == 1) { b = bpre; asm volatile("" : "+v"(b)); }
        else if (F == 6) { float x0, x1; asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(addr) : "memory"); b = (f2){x0, x1}; }
        else if (F == 7) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
        else asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
        if (F == 2) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        float e0, e1;
        const float b0 = F == 1 ? bpre[0] : bia[m], b1 = F == 1 ? bpre[1] : bia[m + 1];
        if (F == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r1) : "v"(a), "v"(b)); e0 = a0 * b1; e1 = a1 * b1; }
        else if (F == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r1) : "v"(a), "v"(b)); e0 = a1 + b0; e1 = a1 + b1; }
        else if (F == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r1) : "v"(a), "v"(b)); e0 = a0 + b1; e1 = a1 + b0; }
        else if (F == 8) { float c0 = b[1]; asm volatile("" : "+v"(c0)); f2 c = {c0, c0}; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r1) : "v"(a), "v"(c)); e0 = a0 + b1; e1 = a1 + b1; }
        else if (F == 9) { f2 c = {0.25f, 0.75f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r1) : "v"(a), "v"(b), "v"(c)); e0 = fmaf(a0, b1, 0.25f); e1 = fmaf(a1, b1, 0.75f); }
        else if (F == 10) { f2 c = {0.25f, 0.75f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r1) : "v"(b), "v"(a), "v"(c)); e0 = fmaf(b1, a0, 0.25f); e1 = fmaf(b1, a1, 0.75f); }
        else if (F == 11) { f2 c = {2.0f, 3.0f}; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r1) : "v"(a), "v"(c), "v"(b)); e0 = fmaf(a0, 2.0f, b1); e1 = fmaf(a1, 3.0f, b1); }
        else if (F == 12) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r1) : "v"(b), "v"(a)); e0 = b1; e1 = a1; }
        else if (F == 13) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r1) : "v"(a), "v"(b)); e0 = a0; e1 = b1; }
        else if (F == 14) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r1) : "v"(b), "v"(a)); e0 = b1 + a0; e1 = b1 + a1; }
        else { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r1) : "v"(a), "v"(b)); e0 = a0 + b1; e1 = a1 + b1; }
        if (r1[0] != e0) { if (!bad[0] && atomicAdd(err + 15, 1u) == 0u) { first[0] = r1[0]; first[1] = e0; first[2] = a0; first[3] = b0; first[4] = b1; first[5] = a1; first[6] = (float)lane; } bad[0]++; }
        bad[1] += r1[1] != e1;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (bad[h]) atomicAdd(err + h * 4 + q, bad[h]);
}

// G: 0 ds_read_b64 pairs from ~77 KB of LDS + f16 MFMAs (the stem kernel's inner loop) | 1 the MFMAs only (operands in registers; the LDS is allocated, not read)
//    2 the LDS reads only | 3 as 0 with 2 KB of LDS | 4 fp32 MFMAs (16x16x4) instead | 5 packed-fp32 VALU work only
template <int G>
__global__ void __launch_bounds__(256) aggressor_k(const float* __restrict__ x, float* __restrict__ y, int steps, int nh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char big[];
    _Float16* xh = reinterpret_cast<_Float16*>(big);
    for (int i = threadIdx.x; i < nh; i += 256) xh[i] = (_Float16)(x[(blockIdx.x * 256 + i) & 0xfffff] * 0.01f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    h8 a = {1, 2, 3, 4, 5, 6, 7, 8};
    uint4 bb = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    f2 pk = {1.0f, 2.0f};
    for (int s = 0; s < steps; ++s) {
        if (G == 0 || G == 2 || G == 3) {
            const int o = ((s * 72 * 19 + wave * 288 + 4 * (lane & 15) + 72 * (lane >> 4)) & ~3) % (nh - 16);
            const uint2 b0 = *reinterpret_cast<const uint2*>(xh + o), b1 = *reinterpret_cast<const uint2*>(xh + o + 4);
            if (G == 2) { bb.x ^= b0.x; bb.y ^= b0.y; bb.z ^= b1.x; bb.w ^= b1.y; }
            else bb = (uint4){b0.x, b0.y, b1.x, b1.y};
        }
        const h8 bh = __builtin_bit_cast(h8, bb);
        if (G == 4) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[0], acc[k & 3], acc, 0, 0, 0);
        } else if (G == 5) {
#pragma unroll
            for (int k = 0; k < 24; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk) : "v"(pk));
        } else if (G == 6) {
            typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
            const bf8 ab = __builtin_bit_cast(bf8, bb);
#pragma unroll
            for (int k = 0; k < 3; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
        } else if (G == 7) {
            typedef float f16v __attribute__((ext_vector_type(16)));
            f16v big_acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) big_acc[k] = acc[k & 3];
#pragma unroll
            for (int k = 0; k < 2; ++k) big_acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, big_acc, 0, 0, 0);
            acc[0] = big_acc[0] + big_acc[7]; acc[1] = big_acc[9];
        } else if (G == 8) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            const h4 a4 = {1, 2, 3, 4};
            const h4 b4 = __builtin_bit_cast(h4, (uint2){bb.x, bb.y});
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
        } else if (G == 9) {
            const long a8 = 0x3838383838383838L, b8 = (long)bb.x | ((long)bb.y << 32);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc, 0, 0, 0);
        } else if (G != 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, a, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh, acc, 0, 0, 0);
        }
    }
    y[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + pk[0] + __builtin_bit_cast(float, bb.x);
}

template <int F> static void run_victim(hipStream_t A, const float* bias, unsigned* err, float* first) {
    hipLaunchKernelGGL(victim_k<F>, dim3(8192), dim3(256), (1312 + 48) * 4, A, bias, err, first, 3);
}
template <int G> static void run_aggr(hipStream_t B, const float* x, float* y, size_t lds, int nh) {
    hipLaunchKernelGGL(aggressor_k<G>, dim3(2048), dim3(256), lds, B, x, y, 25, nh);
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 100;
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    float *bias, *x, *y, *first;
    unsigned* err;
    CK(hipMalloc(&bias, 64 * 4)); CK(hipMalloc(&x, (1 << 20) * 4)); CK(hipMalloc(&y, 4096 * 256 * 4)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&first, 64));
    float hb[64];
    for (int i = 0; i < 64; ++i) hb[i] = (i & 1) ? 2.5e-4f : -2.5e-4f * (1 + i % 3);
    CK(hipMemcpy(bias, hb, 256, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (1 << 20) * 4));
    const int nh = 2 * 133 * 72 * 2;                        // halfs: hi + lo pieces of the stem kernel's staged rows (Cin = 2)
    const size_t big = (size_t)nh * 2 + 464;
#define ATTR(G) CK(hipFuncSetAttribute((const void*)aggressor_k<G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ATTR(0) ATTR(1) ATTR(2) ATTR(3) ATTR(4) ATTR(5) ATTR(6) ATTR(7) ATTR(8) ATTR(9)
    const char* fname[15] = {"ds_read2_b32 -> v_pk_add_f32 op_sel:[0,1]", "operand read from LDS before the loop", "s_nop 32 between LDS wait and add", "v_pk_mul_f32 op_sel:[0,1]",
                            "v_pk_add_f32 op_sel:[1,0]", "op_sel:[0,1] op_sel_hi:[1,0]", "two ds_read_b32 -> op_sel:[0,1]", "ds_read_b64 -> op_sel:[0,1]", "v_mov to an even register + op_sel_hi:[1,0]",
                            "v_pk_fma_f32 op_sel:[0,1,0] (src1)", "v_pk_fma_f32 op_sel:[1,0,0] (src0)", "v_pk_fma_f32 op_sel:[0,0,1] (src2)", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]",
                            "v_pk_add_f32 op_sel:[1,0] (operand in src0)"};
    const char* gname[11] = {"(alone)", "LDS reads + f16 MFMAs, 77 KB LDS", "f16 MFMAs only (77 KB LDS allocated)", "LDS reads only (77 KB)", "LDS reads + f16 MFMAs, 2 KB LDS", "fp32 MFMAs 16x16x4",
                             "packed-fp32 VALU only", "bf16 MFMAs 16x16x32", "f16 MFMAs 32x32x16", "f16 MFMAs 16x16x16", "fp8 MFMAs 16x16x32"};
    for (int F = 0; F < 15; ++F)
        for (int G = -1; G < 10; ++G) {
            if (F > 0 && G > 0) continue;                   // the aggressor forms are swept with the base victim only
            CK(hipMemset(err, 0, 64)); CK(hipMemset(first, 0, 64));
            CK(hipDeviceSynchronize());
            for (int rep = 0; rep < reps; ++rep) {
                switch (G) { case 0: run_aggr<0>(B, x, y, big, nh); break; case 1: run_aggr<1>(B, x, y, big, nh); break; case 2: run_aggr<2>(B, x, y, big, nh); break;
                             case 3: run_aggr<3>(B, x, y, 2048, 1024); break; case 4: run_aggr<4>(B, x, y, big, nh); break; case 5: run_aggr<5>(B, x, y, big, nh); break;
                             case 6: run_aggr<6>(B, x, y, big, nh); break; case 7: run_aggr<7>(B, x, y, big, nh); break; case 8: run_aggr<8>(B, x, y, big, nh); break; case 9: run_aggr<9>(B, x, y, big, nh); break; default: break; }
                switch (F) { case 0: run_victim<0>(A, bias, err, first); break; case 1: run_victim<1>(A, bias, err, first); break; case 2: run_victim<2>(A, bias, err, first); break;
                             case 3: run_victim<3>(A, bias, err, first); break; case 4: run_victim<4>(A, bias, err, first); break; case 5: run_victim<5>(A, bias, err, first); break;
                             case 6: run_victim<6>(A, bias, err, first); break; case 7: run_victim<7>(A, bias, err, first); break; case 8: run_victim<8>(A, bias, err, first); break;
                             case 9: run_victim<9>(A, bias, err, first); break; case 10: run_victim<10>(A, bias, err, first); break; case 11: run_victim<11>(A, bias, err, first); break;
                             case 12: run_victim<12>(A, bias, err, first); break; case 13: run_victim<13>(A, bias, err, first); break; default: run_victim<14>(A, bias, err, first); break; }
                if ((rep & 15) == 15) CK(hipDeviceSynchronize());
            }
            CK(hipDeviceSynchronize());
            unsigned h[16]; float f[16];
            CK(hipMemcpy(h, err, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 64, hipMemcpyDeviceToHost));
            printf("%-44s | %-38s | low half wrong, lanes 0-15/16-31/32-47/48-63: %u/%u/%u/%u  high half: %u/%u/%u/%u", fname[F], gname[G + 1], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            if (h[15]) printf("  first: got %.9g want %.9g (a0 %.9g a1 %.9g b0 %.9g b1 %.9g lane %d)", f[0], f[1], f[2], f[5], f[3], f[4], (int)f[6]);
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
