// Stand-alone stress test of the cross-lane hand-off the launch tapes use (csrc/tape.hip): does a kernel C, queued on stream B behind a poll kernel W that waited for the flag
// a set kernel S stored on stream A behind a producer kernel P, ALWAYS see every byte P wrote?  (suspect (ii): "the release in vx_flag_set_k covers only
// that one-thread kernel's own XCD; cross-XCD visibility of the producer's data rests on the runtime's end-of-kernel release and the consumer's start-of-kernel acquire".)
//   hipcc --offload-arch=gfx950 -O2 tools/flag_hop_stress.hip -o /tmp/fhs && /tmp/fhs [iterations]
// Per iteration r:  A: P(r) -> S(f1, r)        B: W(f1, r) -> C(r) -> S(f2, r)        A: W(f2, r)  (so that P(r + 1) does not overwrite what C(r) still reads)
// P writes buf[i] = mix(i, r) with plain stores (or float atomics onto a zeroed buffer); C's block b reads the chunk block b + shift wrote (another XCD) and counts
// mismatches.  The buffer keeps its address, so the consumer's XCD holds the lines of iteration r - 1 in its L2.  Two more streams run a streaming kernel as noise.
// Modes: flags (the tape's default), events (hipEventRecord + hipStreamWaitEvent), flags + an extra device-wide fence kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned i, unsigned r) { return (i * 2654435761u) ^ (r * 40503u + 17u); }

__global__ void produce_k(unsigned* buf, unsigned n, unsigned r) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = mix(i, r);
}
__global__ void produce_atomic_k(unsigned* buf, unsigned n, unsigned r) {      // two blocks contribute to every word (like partial sums of a weight gradient)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) atomicAdd(buf + (i % n), mix(i % n, r) + (i >= n ? 1u : 0u));
}
__global__ void zero_k(unsigned* buf, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = 0u;
}
__global__ void consume_k(const unsigned* buf, unsigned n, unsigned r, unsigned shift, int atomic_mode, unsigned* err, unsigned* first) {
    const unsigned nb = gridDim.x, per = (n + nb - 1) / nb;
    const unsigned b = (blockIdx.x + shift) % nb;
    for (unsigned k = threadIdx.x; k < per; k += blockDim.x) {
        const unsigned i = b * per + k;
        if (i < n) {
            const unsigned want = atomic_mode ? 2u * mix(i, r) + 1u : mix(i, r);
            const unsigned got = buf[i];
            if (got != want) { if (atomicAdd(err, 1u) == 0u) { first[0] = i; first[1] = got; first[2] = want; first[3] = r; } }
        }
    }
}
__global__ void noise_k(float* a, size_t n, int reps) {
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 123.456f) a[0] = acc;
}
__global__ void flag_set_k(unsigned* flag, unsigned value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void flag_wait_k(const unsigned* flag, unsigned value) {
    if (threadIdx.x == 0)
        while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) __builtin_amdgcn_s_sleep(1);
}

struct Result { unsigned err; unsigned first[4]; float ms; };

static Result run(int mode, unsigned n, int atomic_mode, int iters, bool noise, int pblocks, int cblocks) {
    hipStream_t A, B, N1, N2;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&N1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&N2, hipStreamNonBlocking));
    unsigned *buf, *flags, *err;
    float* nz;
    const size_t nzn = 32u << 20;
    CK(hipMalloc(&buf, sizeof(unsigned) * n)); CK(hipMalloc(&flags, 64)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&nz, nzn * 4));
    CK(hipMemset(flags, 0, 64)); CK(hipMemset(err, 0, 64)); CK(hipMemset(nz, 0, nzn * 4)); CK(hipMemset(buf, 0, sizeof(unsigned) * n));
    CK(hipDeviceSynchronize());
    hipEvent_t e1, e2, t0, t1;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    CK(hipEventRecord(t0, A));
    for (int r = 1; r <= iters; ++r) {
        if (noise && (r % 8) == 1) {
            hipLaunchKernelGGL(noise_k, dim3(512), dim3(256), 0, N1, nz, nzn / 2, 1);
            hipLaunchKernelGGL(noise_k, dim3(512), dim3(256), 0, N2, nz + nzn / 2, nzn / 2, 1);
        }
        if (atomic_mode) {
            hipLaunchKernelGGL(zero_k, dim3(pblocks), dim3(256), 0, A, buf, n);
            hipLaunchKernelGGL(produce_atomic_k, dim3(pblocks), dim3(256), 0, A, buf, n, (unsigned)r);
        } else
            hipLaunchKernelGGL(produce_k, dim3(pblocks), dim3(256), 0, A, buf, n, (unsigned)r);
        if (mode == 1) { CK(hipEventRecord(e1, A)); CK(hipStreamWaitEvent(B, e1, 0)); }
        else { hipLaunchKernelGGL(flag_set_k, dim3(1), dim3(1), 0, A, flags, (unsigned)r); hipLaunchKernelGGL(flag_wait_k, dim3(1), dim3(64), 0, B, flags, (unsigned)r); }
        hipLaunchKernelGGL(consume_k, dim3(cblocks), dim3(256), 0, B, buf, n, (unsigned)r, 3u + (unsigned)(r % 5), atomic_mode, err, err + 4);
        if (mode == 1) { CK(hipEventRecord(e2, B)); CK(hipStreamWaitEvent(A, e2, 0)); }
        else { hipLaunchKernelGGL(flag_set_k, dim3(1), dim3(1), 0, B, flags + 8, (unsigned)r); hipLaunchKernelGGL(flag_wait_k, dim3(1), dim3(64), 0, A, flags + 8, (unsigned)r); }
        if ((r % 256) == 0) CK(hipStreamSynchronize(A));          // keep the queues shallow (the tape's replays are also a few hundred launches deep)
    }
    CK(hipEventRecord(t1, A));
    CK(hipDeviceSynchronize());
    Result R{};
    unsigned h[8];
    CK(hipMemcpy(h, err, 32, hipMemcpyDeviceToHost));
    R.err = h[0]; for (int k = 0; k < 4; ++k) R.first[k] = h[4 + k];
    CK(hipEventElapsedTime(&R.ms, t0, t1));
    CK(hipFree(buf)); CK(hipFree(flags)); CK(hipFree(err)); CK(hipFree(nz));
    CK(hipStreamDestroy(A)); CK(hipStreamDestroy(B)); CK(hipStreamDestroy(N1)); CK(hipStreamDestroy(N2));
    return R;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const char* mname[2] = {"flags", "events"};
    for (int mode = 0; mode < 2; ++mode)
        for (unsigned kb : {64u, 512u, 4096u, 32768u})
            for (int atomic_mode = 0; atomic_mode < 2; ++atomic_mode)
                for (int noise = 0; noise < 2; ++noise) {
                    const unsigned n = kb * 256u;
                    const int it = kb >= 32768u ? iters / 8 : iters;
                    Result R = run(mode, n, atomic_mode, it, noise != 0, 256, 256);
                    printf("%-6s %6u KB %s %s: %d hand-offs, %u stale words", mname[mode], kb, atomic_mode ? "atomics" : "stores ", noise ? "noise" : "quiet", it, R.err);
                    if (R.err) printf(" (first: word %u got %08x want %08x at r = %u)", R.first[0], R.first[1], R.first[2], R.first[3]);
                    printf("  %.2f us per round trip\n", 1e3f * R.ms / it);
                    fflush(stdout);
                }
    return 0;
}
