// Launch tape: a captured training stage replayed as plain kernel launches on several HIP streams.
//
// Why: the training step is ~640 short kernels (median ~15 us).  Launched eagerly through autograd the host needs ~10 ms to enqueue them, which
// is also what the GPU needs to run them one after the other -- the forked streams never overlap because the host is not ahead.  hipGraphLaunch
// on ROCm 7.2 costs ~17 us of host time per kernel node and does not overlap branches either.  The capture itself is
// fine, though: it holds every launch with its final arguments, addresses from a private memory pool, and the exact dependency DAG.  This file
// walks a captured hipGraph_t once (nodes, edges, launch parameters), lays the nodes out on a few stream "lanes" (a node continues the lane
// of a predecessor whenever it can; an event is needed only for edges that cross lanes), and replays the list with hipLaunchKernel /
// hipMemsetAsync: ~3 us of host time per node, no interpreter, no allocator, no autograd, and the lanes run concurrently.
//
// Any schedule that respects the DAG is as valid as hipGraphLaunch's (the capture-mode allocator hands a block to a second tensor only
// along graph edges), so memory safety is the captured graph's.  The caller keeps the captured graph (and its pool) alive: the kernel
// argument arrays returned by hipGraphKernelNodeGetParams point into the graph's nodes.
#include "vx_common.h"
#include "../../include/veloxseg_hip.h"
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#include <atomic>
#include <vector>
#include <algorithm>
#include <unordered_map>

namespace {
enum { T_KERNEL = 0, T_MEMSET = 1, T_COPY = 2 };
struct TapeNode {
    int type = T_KERNEL;
    int lane = 0;
    int marker = -1;                    // >= 0: a vx_tape_mark(id) node -- not launched, an event is recorded in its place
    int mode = 0;                       // kernels: 0 = undecided, 1 = hipLaunchKernel (host stub), 2 = hipModuleLaunchKernel (hipFunction_t)
    hipKernelNodeParams k{};
    hipMemsetParams ms{};
    hipGraph_t copy_graph = nullptr;    // T_COPY: a one-node graph (the captured memcpy) and its executable
    hipGraphExec_t copy_exec = nullptr;
    std::vector<int> waits;             // events (indices of earlier nodes on other lanes) to wait for before the launch
    bool record = false;                // some later node on another lane waits for this one
    hipEvent_t ev = nullptr;
    int flag = -1;                      // record nodes: slot in the tape's device flag words (cross-lane dependencies through flag kernels)
};
}  // namespace

struct VxTape {
    int lane_rot = 0;          // lanes l of this tape run on pool stream (l + lane_rot) % kPool (vx_tape_set_lane_rotation)
    std::vector<TapeNode> nodes;        // in launch order (a topological order of the captured DAG)
    std::vector<hipStream_t> lanes;     // set at replay: the caller's stream for a one-lane tape, else the process-wide lane streams
    std::vector<int> lane_last;         // last node of each lane (-1: unused)
    hipEvent_t start = nullptr;
    std::vector<hipEvent_t> lane_end;
    int n_kernels = 0, n_events = 0, n_cross = 0;
    unsigned* flags = nullptr;          // one device word per recording node; replay r stores r, waiters poll for >= r (no reset between replays)
    unsigned seq = 0;
    int flag_start = 0, n_flags = 0;
    hipStream_t last_s0 = nullptr;
};

#define HIPQ(call, what)                                                                      \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) VX_FAIL(-2, "vx_tape: %s: %s", what, hipGetErrorString(e__)); \
    } while (0)


// ---------------------------------------------------------------------------------------------------------------- lane streams
// ROCm multiplexes every hipStream of the process onto GPU_MAX_HW_QUEUES (4) hardware queues, and two streams that share a hardware queue run
// one after the other.  Which queue a stream lands on cannot be queried, but it can be measured: a 1-block kernel that spins for ~60 us is put
// on two streams at once; 60 us means two queues, 120 us means one.  The first use picks, out of a few freshly created streams, four that
// overlap pairwise; every multi-lane tape of the process runs its lanes on those and only gates / joins them with the caller's stream.  (The
// caller's stream is not used as a lane: work on the NULL stream -- PyTorch's default -- was measured not to overlap with any other stream.)
__global__ void vx_spin_k(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
// a one-block kernel that keeps one wave busy for `us` microseconds on `stream`: the stand-in for a collective in tools/comm_standin_probe.py
extern "C" int vx_spin_us(float us, void* stream) {
    VX_REQUIRE(us >= 0.0f && us <= 1e6f, "vx_spin_us: 0 .. 1 s");
    hipLaunchKernelGGL(vx_spin_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0f));      // wall_clock64 counts at 100 MHz
    VX_LAUNCH_CHECK("vx_spin_us");
    return 0;
}
// Cross-lane dependencies without events: the producing lane runs a one-thread kernel that stores a sequence number to a device flag; the waiting
// lane runs a one-wave kernel that polls the flag until it reaches that number.  On this runtime an event record + stream wait costs ~14 us of
// queue time per hop (tools/event_hop_probe.py: every flag combination of hipEventCreateWithFlags); two back-to-back tiny kernels cost ~1.6 us each.
// Progress: the set kernel of a dependency is always SUBMITTED before its poll kernel (tape order is topological, a hop submits set then wait), and a
// hardware queue runs its packets in submission order, so the earliest unfinished packet of the process can always run -- whatever the stream ->
// hardware-queue mapping.  The code nevertheless uses the flags ONLY when the lane calibration found the kPool lanes on pairwise different hardware
// queues (flags_ok(): g_pool.distinct == kPool); with aliased lanes, under a kernel-serialising profiler, or with VELOXSEG_TAPE_FLAGS=0 the same
// dependencies are events.
// A poll gives up after g_flag_timeout_ticks (default 5 s, VELOXSEG_TAPE_FLAG_TIMEOUT_MS / vx_tape_set_flag_timeout_ms; 0 = never): it then bumps a
// host-visible error word and RETURNS (the kernels behind it run with an unmet dependency -- wrong numbers, but no trap, no dead GPU context); the
// next vx_tape_replay / vx_tape_hop / vx_tape_flag_timeouts call reports it through its status and vx_last_error().
__global__ void vx_flag_set_k(unsigned* flag, unsigned value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void vx_flag_fill_k(unsigned* flags, int n, unsigned value) {       // error path of a replay: release every waiter of the tape
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(flags + i, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void vx_flag_wait_k(const unsigned* flag, unsigned value, long long timeout_ticks, unsigned* err) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (timeout_ticks > 0 && wall_clock64() - t0 > timeout_ticks) {      // wall_clock64 counts at 100 MHz
                if (err) __hip_atomic_fetch_add(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
}
static long long g_flag_timeout_ticks = -1;          // -1: from the environment
static unsigned* g_flag_err_host = nullptr;          // pinned, device-visible word: number of polls that gave up
static unsigned* g_flag_err_dev = nullptr;
static long long flag_timeout_ticks() {
    if (g_flag_timeout_ticks < 0) {
        const char* e = getenv("VELOXSEG_TAPE_FLAG_TIMEOUT_MS");
        const long long ms = (e && e[0]) ? atoll(e) : 5000;
        g_flag_timeout_ticks = ms < 0 ? 0 : ms * 100000LL;
    }
    return g_flag_timeout_ticks;
}
static unsigned* flag_err_word() {
    if (!g_flag_err_dev) {
        if (hipHostMalloc((void**)&g_flag_err_host, sizeof(unsigned) * 4, hipHostMallocMapped) != hipSuccess) { g_flag_err_host = nullptr; return nullptr; }
        g_flag_err_host[0] = 0;
        if (hipHostGetDevicePointer((void**)&g_flag_err_dev, g_flag_err_host, 0) != hipSuccess) g_flag_err_dev = nullptr;
    }
    return g_flag_err_dev;
}
extern "C" int vx_tape_set_flag_timeout_ms(int ms) { g_flag_timeout_ticks = ms <= 0 ? 0 : (long long)ms * 100000LL; return 0; }
// number of polls that gave up since the last call (answer, not a status); clears the count
extern "C" int vx_tape_flag_timeouts(void) {
    if (!g_flag_err_host) return 0;
    const unsigned n = __atomic_exchange_n(g_flag_err_host, 0u, __ATOMIC_RELAXED);
    return (int)n;
}
static inline void flag_set(hipStream_t s, unsigned* f, unsigned v) { hipLaunchKernelGGL(vx_flag_set_k, dim3(1), dim3(1), 0, s, f, v); }
static inline void flag_wait(hipStream_t s, const unsigned* f, unsigned v) {
    hipLaunchKernelGGL(vx_flag_wait_k, dim3(1), dim3(64), 0, s, f, v, flag_timeout_ticks(), flag_err_word());
}
static int g_use_flags = -1;            // -1: from the environment (VELOXSEG_TAPE_FLAGS, default on)
static bool use_flags() {
    if (g_use_flags < 0) {
        const char* e = getenv("VELOXSEG_TAPE_FLAGS");
        // rocprofv3 --pmc runs ONE kernel at a time, across all queues: a polling kernel would never see its flag set
        const char* pmc = getenv("ROCPROF_COUNTER_COLLECTION");
        const char* ctr = getenv("ROCPROF_COUNTERS");
        const char* ser = getenv("AMD_SERIALIZE_KERNEL");
        const bool serialised = (pmc && pmc[0] == '1') || (ctr && ctr[0]) || (ser && ser[0] && ser[0] != '0');
        g_use_flags = ((e && e[0] == '0') || serialised) ? 0 : 1;
    }
    return g_use_flags == 1;
}
extern "C" int vx_tape_set_flags(int on) { g_use_flags = on ? 1 : 0; return 0; }
static bool flags_ok();                 // use_flags() AND the calibrated lanes sit on pairwise different hardware queues (defined below the lane pool)
// `dst` waits for everything enqueued on `src` so far -- hipEventRecord + hipStreamWaitEvent, or (flags on) a set kernel on src and a poll kernel on dst.
// `slot` (0..255) names the call site: a site must always use the same src stream (its sequence numbers rely on that stream's order).
static unsigned* g_hop_flags = nullptr;
static unsigned g_hop_seq[256] = {};
static hipEvent_t g_hop_ev[256] = {};
static void* g_hop_src[256] = {};
extern "C" int vx_tape_hop(int slot, void* src, void* dst) {
    if (slot < 0 || slot >= 256) return -1;
    if (src == dst) return 0;
    if (int nto = vx_tape_flag_timeouts()) VX_FAIL(-3, "vx_tape_hop: %d cross-lane poll(s) of an earlier replay gave up after the flag timeout (results of that step are not trustworthy)", nto);
    if (flags_ok()) {
        if (!g_hop_flags) {
            // hipMemset of device memory may return before the fill has run (it is queued on the NULL stream): a poll on another stream would then read
            // whatever the allocation held and pass at once.  That was the first replay of a process going wrong when the GPU was shared with
            // another process (tools/two_procs.sh) -- the words must be zero before anything is launched
            if (hipMalloc((void**)&g_hop_flags, sizeof(unsigned) * 256) != hipSuccess || hipMemset(g_hop_flags, 0, sizeof(unsigned) * 256) != hipSuccess ||
                hipDeviceSynchronize() != hipSuccess) return -2;
        }
        if (g_hop_src[slot] && g_hop_src[slot] != src) (void)hipStreamSynchronize((hipStream_t)g_hop_src[slot]);      // the site changed its source stream (rare): keep the order
        g_hop_src[slot] = src;
        const unsigned seq = ++g_hop_seq[slot];
        flag_set((hipStream_t)src, g_hop_flags + slot, seq);
        flag_wait((hipStream_t)dst, (const unsigned*)(g_hop_flags + slot), seq);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (!g_hop_ev[slot] && hipEventCreateWithFlags(&g_hop_ev[slot], hipEventDisableTiming) != hipSuccess) return -2;
    if (hipEventRecord(g_hop_ev[slot], (hipStream_t)src) != hipSuccess || hipStreamWaitEvent((hipStream_t)dst, g_hop_ev[slot], 0) != hipSuccess) return -2;
    return 0;
}      // A/B: cross-lane dependencies through flag kernels (1) or events (0)
extern "C" int vx_tape_flag_set(void* flag, int value, void* stream) {
    hipLaunchKernelGGL(vx_flag_set_k, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)flag, (unsigned)value);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int vx_tape_flag_wait(const void* flag, int value, void* stream) {
    flag_wait((hipStream_t)stream, (const unsigned*)flag, (unsigned)value);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
namespace {
constexpr int kPool = 4;
struct LanePool {
    bool ready = false;
    int distinct = 0;                   // how many of the lane streams were measured to overlap pairwise
    int shared_with_caller = -1;        // the lane whose stream sits on the hardware queue of the caller's stream (-1: none was found)
    hipStream_t lane[kPool] = {nullptr, nullptr, nullptr, nullptr};
} g_pool;

// elapsed microseconds of one spin on each of the two streams (b == nullptr: on `a` alone), launched together behind `gate`
static long long g_spin_ticks = 6000;             // wall_clock64 counts at 100 MHz: 60 us (raised by spin_calibrate when launches / events are slow, e.g. under a profiler)
static float g_pair_limit_us = 95.f;              // a pair that takes less than this overlapped (one spin = 60 us; two in a row = 120 us)
int pair_us(hipStream_t gate, hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t ea, hipEvent_t eb, hipEvent_t e1, float* out) {
    const long long ticks = g_spin_ticks;
    HIPQ(hipEventRecord(e0, gate), "hipEventRecord");
    HIPQ(hipStreamWaitEvent(a, e0, 0), "hipStreamWaitEvent");
    if (b) HIPQ(hipStreamWaitEvent(b, e0, 0), "hipStreamWaitEvent");
    hipLaunchKernelGGL(vx_spin_k, dim3(1), dim3(64), 0, a, ticks);
    if (b) hipLaunchKernelGGL(vx_spin_k, dim3(1), dim3(64), 0, b, ticks);
    HIPQ(hipEventRecord(ea, a), "hipEventRecord");
    if (b) HIPQ(hipEventRecord(eb, b), "hipEventRecord");
    HIPQ(hipStreamWaitEvent(gate, ea, 0), "hipStreamWaitEvent");
    if (b) HIPQ(hipStreamWaitEvent(gate, eb, 0), "hipStreamWaitEvent");
    HIPQ(hipEventRecord(e1, gate), "hipEventRecord");
    HIPQ(hipEventSynchronize(e1), "hipEventSynchronize");
    HIPQ(hipEventElapsedTime(out, e0, e1), "hipEventElapsedTime");
    *out *= 1e3f;
    return 0;
}
// The verdict "two spins took less than 95 us, so they overlapped" assumes that the fork / join around them costs a few microseconds.  Under rocprofv3 --kernel-trace every
// dispatch and every event hop is intercepted: ONE 60 us spin then measures 100+ us, every pair looks serialised, the tapes fell back to event waits and the profiled
// schedule was not the benched one (`lanes_on_distinct_hw_queues: 2`, 653.8 instead of 1041 patches/s).  So the yardstick is measured: one spin alone
// on one stream = spin + overhead; when the overhead is more than a third of the spin, the spin is lengthened to 8 x the overhead (at most 4 ms), and a pair counts as
// overlapped when it takes less than (one spin alone) + half a spin.
int spin_calibrate(hipStream_t gate, hipStream_t a, hipEvent_t (&ev)[4]) {
    float us = 0.f;
    for (int round = 0; round < 3; ++round) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { if (int rc = pair_us(gate, a, nullptr, ev[0], ev[1], ev[2], ev[3], &us)) return rc; best = us < best ? us : best; }
        const float spin = (float)g_spin_ticks * 0.01f, over = best > spin ? best - spin : 0.f;
        g_pair_limit_us = best + 0.5f * spin;
        if (over * 3.f <= spin || g_spin_ticks >= 400000) break;
        long long t = (long long)(over * 8.f * 100.f);
        g_spin_ticks = t > 400000 ? 400000 : (t < 6000 ? 6000 : t);
    }
    return 0;
}

// one attempt: pick kPool pairwise-overlapping streams out of NC fresh ones; *nd = how many were found (the rest are filled with sharing streams)
int pool_attempt(hipStream_t (&chosen)[kPool], int* nd) {
    hipEvent_t ev[4];
    for (auto& e : ev) HIPQ(hipEventCreate(&e), "hipEventCreate");
    constexpr int NC = 16;
    hipStream_t cand[NC], gate;
    // (default priority: lanes created with hipStreamCreateWithPriority -- highest or lowest, all or some -- made the step 1.7x slower)
    for (auto& c : cand) HIPQ(hipStreamCreateWithFlags(&c, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    HIPQ(hipStreamCreateWithFlags(&gate, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    bool used[NC] = {};
    int n = 0;
    float us = 0.f;
    int rc = pair_us(gate, cand[0], cand[1], ev[0], ev[1], ev[2], ev[3], &us);        // warm-up (code object load)
    if (rc == 0) rc = spin_calibrate(gate, cand[0], ev);                              // the yardstick: one spin alone, and a spin long enough for this runtime's hop costs
    auto overlap = [&](hipStream_t a, hipStream_t b, bool* ok) {
        float best = 1e9f;
        for (int rep = 0; rep < 3 && rc == 0; ++rep) { rc = pair_us(gate, a, b, ev[0], ev[1], ev[2], ev[3], &us); best = us < best ? us : best; }
        *ok = best < g_pair_limit_us;             // two spins in a row = one alone + a whole spin
    };
    for (int c = 0; c < NC && n < kPool && rc == 0; ++c) {
        bool ok = true;
        for (int k = 0; k < n && ok && rc == 0; ++k) overlap(chosen[k], cand[c], &ok);
        if (ok && rc == 0) { chosen[n++] = cand[c]; used[c] = true; }
    }
    // second look at every pair of the selection (a disturbed measurement must not freeze a bad choice for the life of the process)
    bool all_ok = (n == kPool);
    for (int i = 0; i < n && all_ok && rc == 0; ++i)
        for (int j = i + 1; j < n && all_ok && rc == 0; ++j) overlap(chosen[i], chosen[j], &all_ok);
    *nd = all_ok ? n : (n < kPool ? n : kPool - 1);
    for (int c = 0; c < NC; ++c) if (!used[c] && n < kPool && rc == 0) { chosen[n++] = cand[c]; used[c] = true; }      // fewer hardware queues than lanes: share
    for (int c = 0; c < NC; ++c) if (!used[c]) (void)hipStreamDestroy(cand[c]);
    (void)hipStreamDestroy(gate);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int pool_init(hipStream_t main) {
    if (g_pool.ready) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) VX_FAIL(-1, "vx_tape: the lane streams cannot be chosen during a stream capture");
    HIPQ(hipDeviceSynchronize(), "hipDeviceSynchronize");        // the spin measurements need a quiet device
    hipStream_t best[kPool] = {};
    int best_n = -1;
    for (int attempt = 0; attempt < 3 && best_n < kPool; ++attempt) {
        hipStream_t chosen[kPool] = {};
        int nd = 0;
        int rc = pool_attempt(chosen, &nd);
        if (rc != 0) return rc;
        if (nd > best_n) {
            for (int k = 0; k < kPool; ++k) { if (best[k]) (void)hipStreamDestroy(best[k]); best[k] = chosen[k]; }
            best_n = nd;
        } else
            for (int k = 0; k < kPool; ++k) if (chosen[k]) (void)hipStreamDestroy(chosen[k]);
    }
    g_pool.distinct = best_n;
    // Five streams -- the caller's and the four lanes -- share four hardware queues: one lane sits on the caller's queue, behind the gates, joins and polls
    // the caller's stream runs.  Which one it is used to be an accident of stream creation order, and it decides 10-14 % of the step
    // (tools/lane_placement_probe.py: every assignment with that stream on lane 2 -- the lightly loaded lane of the encoder tapes -- gives 5.4 ms, every other
    // one 5.95-6.2 ms; one process in ten came up with a different order).  Find the lane stream that does not overlap with the caller's stream and give it lane 2.
    {
        hipEvent_t ev[4];
        for (auto& e : ev) HIPQ(hipEventCreate(&e), "hipEventCreate");
        hipStream_t gate;
        HIPQ(hipStreamCreateWithFlags(&gate, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        int shared = -1;
        float worst = 0.f;
        for (int k = 0; k < kPool && best[k]; ++k) {
            float bestus = 1e9f, us = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                int rc = pair_us(gate, main, best[k], ev[0], ev[1], ev[2], ev[3], &us);
                if (rc != 0) return rc;
                bestus = us < bestus ? us : bestus;
            }
            if (bestus >= g_pair_limit_us && bestus > worst) { worst = bestus; shared = k; }      // two spins in a row
        }
        g_pool.shared_with_caller = shared;
        if (shared >= 0 && shared != 2 && kPool > 2) { hipStream_t t = best[2]; best[2] = best[shared]; best[shared] = t; g_pool.shared_with_caller = 2; }
        (void)hipStreamDestroy(gate);
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    for (int k = 0; k < kPool; ++k) g_pool.lane[k] = best[k];
    g_pool.ready = true;
    return 0;
}
}  // namespace

// A marker is a captured no-op kernel that carries an id.  The tape does not launch it: at its place in the schedule (after every node it depends on)
// it records an event on its lane, which another stream (the communication stream of a data-parallel step) can wait for while the rest of
// the tape is still running: "bucket `id` of the flat gradient is complete from here on".
__global__ void vx_tape_marker_k(int id) { (void)id; }
extern "C" int vx_tape_mark(int id, void* stream) {
    VX_REQUIRE(id >= 0, "vx_tape_mark: id must be >= 0");
    hipLaunchKernelGGL(vx_tape_marker_k, dim3(1), dim3(1), 0, (hipStream_t)stream, id);
    VX_LAUNCH_CHECK("vx_tape_mark");
    return 0;
}

// which calibrated stream serves which lane (diagnostics / the engine's placement check): lane k <- stream perm[k] of the current assignment
extern "C" int vx_tape_permute_lanes(const int* perm) {
    VX_REQUIRE(perm && g_pool.ready, "vx_tape_permute_lanes: the lane streams have not been chosen yet");
    hipStream_t cur[kPool];
    bool seen[kPool] = {};
    for (int k = 0; k < kPool; ++k) {
        VX_REQUIRE(perm[k] >= 0 && perm[k] < kPool && !seen[perm[k]], "vx_tape_permute_lanes: not a permutation");
        seen[perm[k]] = true;
        cur[k] = g_pool.lane[perm[k]];
    }
    HIPQ(hipDeviceSynchronize(), "hipDeviceSynchronize");
    for (int k = 0; k < kPool; ++k) g_pool.lane[k] = cur[k];
    return 0;
}
static bool flags_ok() { return use_flags() && (!g_pool.ready || g_pool.distinct >= kPool); }
extern "C" int vx_tape_lane_on_caller_queue(void) { return !g_pool.ready ? 5 : g_pool.shared_with_caller < 0 ? 4 : g_pool.shared_with_caller; }      // answer: lane 0..3, 4 = none found, 5 = lanes not chosen yet
extern "C" int vx_tape_spin_us(void) { return (int)(g_spin_ticks / 100); }      // answer: the spin length (us) the lane calibration settled on (60 normally; longer when launches / event hops are slow, e.g. under rocprofv3)
extern "C" int vx_tape_lanes_distinct(void) { return g_pool.ready ? g_pool.distinct : -1; }      // how many lane streams were measured to overlap pairwise (-1: not chosen yet)

extern "C" int vx_tape_lane_stream(void* any_stream, int lane, void** out) {
    VX_REQUIRE(out && lane >= 0, "vx_tape_lane_stream: bad arguments");
    int rc = pool_init((hipStream_t)any_stream);
    if (rc) return rc;
    *out = (void*)g_pool.lane[lane % kPool];
    return 0;
}

// dur != nullptr: PROFILE-GUIDED layout.  dur[i] = measured duration (us) of node i of the tape that vx_tape_build lays out for the same graph
// (vx_tape_profile): the nodes are then list-scheduled -- longest remaining path first, each on the lane where it can start earliest (a hop between
// lanes costs VX_TAPE_HOP_US, every node VX_TAPE_GAP_US of dispatch gap) -- and the tape replays them in that order on those lanes.  The greedy layout of
// vx_tape_build knows no durations: it continues a predecessor's lane with whichever successor was captured first, which parks sinks (deferred weight
// gradients) on the longest lane while another is nearly idle.
#define VX_TAPE_HOP_US 3.0f
#define VX_TAPE_GAP_US 2.0f
static int tape_build_impl(void* graph_, int max_lanes, const float* dur, int ndur, VxTape** out) {
    VX_REQUIRE(graph_ && out && max_lanes >= 1 && max_lanes <= 16, "vx_tape_build: bad arguments");
    hipGraph_t graph = (hipGraph_t)graph_;
    size_t nn = 0, ne = 0;
    HIPQ(hipGraphGetNodes(graph, nullptr, &nn), "hipGraphGetNodes");
    std::vector<hipGraphNode_t> gn(nn);
    if (nn) HIPQ(hipGraphGetNodes(graph, gn.data(), &nn), "hipGraphGetNodes");
    HIPQ(hipGraphGetEdges(graph, nullptr, nullptr, &ne), "hipGraphGetEdges");
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) HIPQ(hipGraphGetEdges(graph, ef.data(), et.data(), &ne), "hipGraphGetEdges");
    std::unordered_map<hipGraphNode_t, int> idx;
    for (size_t i = 0; i < nn; ++i) idx[gn[i]] = (int)i;
    const int N = (int)nn;
    std::vector<std::vector<int>> pred(N), succ(N);
    for (size_t e = 0; e < ne; ++e) {
        auto a = idx.find(ef[e]), b = idx.find(et[e]);
        VX_REQUIRE(a != idx.end() && b != idx.end(), "vx_tape_build: edge to an unknown node");
        pred[b->second].push_back(a->second);
        succ[a->second].push_back(b->second);
    }
    // node kinds; event-record / event-wait / empty nodes carry dependencies only
    std::vector<int> kind(N, -1);
    std::vector<TapeNode> raw(N);
    for (int i = 0; i < N; ++i) {
        hipGraphNodeType t;
        HIPQ(hipGraphNodeGetType(gn[i], &t), "hipGraphNodeGetType");
        if (t == hipGraphNodeTypeKernel) {
            kind[i] = T_KERNEL;
            HIPQ(hipGraphKernelNodeGetParams(gn[i], &raw[i].k), "hipGraphKernelNodeGetParams");
            VX_REQUIRE(raw[i].k.func != nullptr, "vx_tape_build: kernel node without a function");
            if (raw[i].k.func == reinterpret_cast<void*>(&vx_tape_marker_k) && raw[i].k.kernelParams && raw[i].k.kernelParams[0])
                raw[i].marker = *reinterpret_cast<const int*>(raw[i].k.kernelParams[0]);
        } else if (t == hipGraphNodeTypeMemset) {
            kind[i] = T_MEMSET;
            HIPQ(hipGraphMemsetNodeGetParams(gn[i], &raw[i].ms), "hipGraphMemsetNodeGetParams");
            VX_REQUIRE(raw[i].ms.height <= 1, "vx_tape_build: 2-D memset nodes are not supported");
            VX_REQUIRE(raw[i].ms.elementSize == 1 || raw[i].ms.elementSize == 2 || raw[i].ms.elementSize == 4, "vx_tape_build: memset element size %u", raw[i].ms.elementSize);
        } else if (t == hipGraphNodeTypeMemcpy) {
            // hipGraphMemcpyNodeGetParams returns uninitialised memory for the 1-D nodes hipMemcpyAsync is captured as (ROCm 7.2), so the copy cannot be
            // read back.  It is replayed as what it is instead: a clone of the captured graph with every other node removed, instantiated once and
            // launched on the node's lane (~20 us of host time; aten copies contiguous tensors this way, e.g. torch.cat over dim 1 at batch 1).
            kind[i] = T_COPY;
            hipGraph_t one = nullptr;
            HIPQ(hipGraphClone(&one, graph), "hipGraphClone");
            hipGraphNode_t keep = nullptr;
            HIPQ(hipGraphNodeFindInClone(&keep, gn[i], one), "hipGraphNodeFindInClone");
            size_t cn = 0;
            HIPQ(hipGraphGetNodes(one, nullptr, &cn), "hipGraphGetNodes");
            std::vector<hipGraphNode_t> cns(cn);
            HIPQ(hipGraphGetNodes(one, cns.data(), &cn), "hipGraphGetNodes");
            for (hipGraphNode_t c : cns) if (c != keep) HIPQ(hipGraphDestroyNode(c), "hipGraphDestroyNode");
            HIPQ(hipGraphInstantiate(&raw[i].copy_exec, one, nullptr, nullptr, 0), "hipGraphInstantiate");
            raw[i].copy_graph = one;
        } else if (t == hipGraphNodeTypeEmpty || t == hipGraphNodeTypeEventRecord || t == hipGraphNodeTypeWaitEvent) {
            kind[i] = -1;
        } else
            VX_FAIL(-1, "vx_tape_build: graph node type %d is not supported", (int)t);
        raw[i].type = kind[i];
    }
    // topological order, capture order as the tie-break (Kahn with a min-heap on the node index)
    std::vector<int> indeg(N), order;
    std::vector<int> heap;
    for (int i = 0; i < N; ++i) { indeg[i] = (int)pred[i].size(); if (!indeg[i]) heap.push_back(i); }
    auto cmp = [](int a, int b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        int u = heap.back(); heap.pop_back();
        order.push_back(u);
        for (int v : succ[u]) if (--indeg[v] == 0) { heap.push_back(v); std::push_heap(heap.begin(), heap.end(), cmp); }
    }
    VX_REQUIRE((int)order.size() == N, "vx_tape_build: the graph has a cycle");
    // contract the dependency-only nodes: real predecessors of every node, in topological order
    std::vector<std::vector<int>> rp(N);
    for (int u : order) {
        std::vector<int>& r = rp[u];
        for (int p : pred[u]) {
            if (kind[p] >= 0) r.push_back(p);
            else r.insert(r.end(), rp[p].begin(), rp[p].end());
        }
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
    }
    std::vector<int> pos(N, -1);
    std::vector<int> real;
    for (int u : order) if (kind[u] >= 0) { pos[u] = (int)real.size(); real.push_back(u); }
    const int R = (int)real.size();
    std::vector<int> pre_lane;                          // profile-guided: lane of node i of the (re-ordered) list
    if (dur != nullptr && R > 1 && max_lanes > 1) {
        VX_REQUIRE(ndur == R, "vx_tape_build_pgo: %d durations for %d nodes", ndur, R);
        std::vector<std::vector<int>> pr(R), sc(R);
        for (int i = 0; i < R; ++i)
            for (int p : rp[real[i]]) { pr[i].push_back(pos[p]); sc[pos[p]].push_back(i); }
        std::vector<float> d(R), bl(R, 0.0f);
        for (int i = 0; i < R; ++i) d[i] = (dur[i] > 0.5f && dur[i] < 1e6f ? dur[i] : 0.5f) + VX_TAPE_GAP_US;
        for (int i = R - 1; i >= 0; --i) {                 // `real` is a topological order
            float m = 0.0f;
            for (int s2 : sc[i]) m = std::max(m, bl[s2]);
            bl[i] = d[i] + m;
        }
        std::vector<int> left(R), lane_of(R, -1), sched;
        std::vector<float> fin(R, 0.0f), lane_free(max_lanes, 0.0f);
        std::vector<char> done(R, 0);
        for (int i = 0; i < R; ++i) left[i] = (int)pr[i].size();
        for (int step = 0; step < R; ++step) {
            int best = -1;
            for (int i = 0; i < R; ++i)                    // the ready node with the longest remaining path (capture order breaks ties)
                if (!done[i] && left[i] == 0 && (best < 0 || bl[i] > bl[best])) best = i;
            VX_REQUIRE(best >= 0, "vx_tape_build_pgo: no ready node (cycle?)");
            int bl_lane = -1; float best_est = 0.0f;
            int pl = -1; float pf = -1.0f;                 // lane of the predecessor that finishes last
            for (int p : pr[best]) if (fin[p] > pf) { pf = fin[p]; pl = lane_of[p]; }
            for (int l = 0; l < max_lanes; ++l) {
                float est = lane_free[l];
                for (int p : pr[best]) est = std::max(est, fin[p] + (lane_of[p] == l ? 0.0f : VX_TAPE_HOP_US));
                if (bl_lane < 0 || est < best_est - 0.01f || (est < best_est + 0.01f && l == pl)) { bl_lane = l; best_est = est; }
            }
            lane_of[best] = bl_lane;
            fin[best] = best_est + d[best];
            lane_free[bl_lane] = fin[best];
            done[best] = 1;
            sched.push_back(best);
            for (int s2 : sc[best]) left[s2]--;
        }
        std::vector<int> real2(R);
        pre_lane.resize(R);
        for (int k = 0; k < R; ++k) { real2[k] = real[sched[k]]; pre_lane[k] = lane_of[sched[k]]; }
        real.swap(real2);
        for (int k = 0; k < R; ++k) pos[real[k]] = k;
    }
    std::vector<int> nsucc(R, 0);                       // successors not yet placed
    for (int u : real) for (int p : rp[u]) nsucc[pos[p]]++;

    VxTape* T = new VxTape();
    T->nodes.resize(R);
    std::vector<int> tail;                              // tail[lane] = last node placed on the lane
    // ancestors reached through the lane order make some edges redundant; track for each node, per lane, the latest node known to precede it
    std::vector<std::vector<int>> known(R);
    for (int i = 0; i < R; ++i) {
        const int u = real[i];
        TapeNode& nd = T->nodes[i];
        nd = raw[u];
        std::vector<int> ps;
        for (int p : rp[u]) ps.push_back(pos[p]);
        int lane = -1;
        if (!pre_lane.empty()) {
            lane = pre_lane[i];
            while ((int)tail.size() <= lane) tail.push_back(-1);
        }
        if (lane < 0)
        for (int p : ps) if (tail[T->nodes[p].lane] == p) { lane = T->nodes[p].lane; break; }      // continue a predecessor's lane
        if (lane < 0) {
            for (size_t l = 0; l < tail.size() && lane < 0; ++l)                                    // a lane whose tail has nothing left to feed
                if (tail[l] >= 0 && nsucc[tail[l]] == 0) lane = (int)l;
            if (lane < 0 && (int)tail.size() < max_lanes) { tail.push_back(-1); lane = (int)tail.size() - 1; }
            if (lane < 0) lane = ps.empty() ? 0 : T->nodes[ps[0]].lane;
        }
        nd.lane = lane;
        std::vector<int>& kn = known[i];
        kn.assign(max_lanes, -1);
        if (tail[lane] >= 0) { kn = known[tail[lane]]; kn[lane] = tail[lane]; }
        // waits: latest predecessor per foreign lane that the lane order does not already cover
        std::vector<int> need(max_lanes, -1);
        for (int p : ps) { const int lp = T->nodes[p].lane; if (lp != lane && p > kn[lp]) need[lp] = std::max(need[lp], p); }
        for (int l = 0; l < max_lanes; ++l)
            if (need[l] >= 0) {
                nd.waits.push_back(need[l]);
                T->nodes[need[l]].record = true;
                T->n_cross++;
                for (int l2 = 0; l2 < max_lanes; ++l2) kn[l2] = std::max(kn[l2], known[need[l]][l2]);
                kn[l] = std::max(kn[l], need[l]);
            }
        for (int p : ps) nsucc[p]--;
        tail[lane] = i;
        if (nd.type == T_KERNEL) T->n_kernels++;
    }
    T->lane_last = tail;
    T->lanes.assign(tail.size(), nullptr);
    T->lane_end.assign(tail.size(), nullptr);
    for (size_t l = 0; l < tail.size(); ++l) HIPQ(hipEventCreateWithFlags(&T->lane_end[l], hipEventDisableTiming), "hipEventCreateWithFlags");
    HIPQ(hipEventCreateWithFlags(&T->start, hipEventDisableTiming), "hipEventCreateWithFlags");
    int nflags = 0;
    for (auto& nd : T->nodes)
        if (nd.record || nd.marker >= 0) {
            HIPQ(hipEventCreateWithFlags(&nd.ev, hipEventDisableTiming), "hipEventCreateWithFlags"); T->n_events++;
            if (nd.marker < 0) nd.flag = nflags++;
        }
    T->flag_start = nflags;                       // + 1 word for the start gate, + 1 per lane for the joins
    nflags += 1 + (int)tail.size();
    T->n_flags = nflags;
    HIPQ(hipMalloc((void**)&T->flags, sizeof(unsigned) * (size_t)nflags), "hipMalloc");
    HIPQ(hipMemset(T->flags, 0, sizeof(unsigned) * (size_t)nflags), "hipMemset");
    HIPQ(hipDeviceSynchronize(), "hipDeviceSynchronize");      // (the fill is queued on the NULL stream: it must have run before the first poll)
    *out = T;
    return 0;
}
extern "C" int vx_tape_build(void* graph_, int max_lanes, VxTape** out) { return tape_build_impl(graph_, max_lanes, nullptr, 0, out); }
extern "C" int vx_tape_build_pgo(void* graph_, int max_lanes, const float* dur_us, int n, VxTape** out) {
    VX_REQUIRE(dur_us && n > 0, "vx_tape_build_pgo: no durations");
    return tape_build_impl(graph_, max_lanes, dur_us, n, out);
}

extern "C" int vx_tape_info(const VxTape* T, int* n_nodes, int* n_kernels, int* n_lanes, int* n_events) {
    VX_REQUIRE(T, "vx_tape_info: null tape");
    if (n_nodes) *n_nodes = (int)T->nodes.size();
    if (n_kernels) *n_kernels = T->n_kernels;
    if (n_lanes) *n_lanes = (int)T->lanes.size();
    if (n_events) *n_events = T->n_events;
    return 0;
}

// Schedule fuzzing (tests / tools/tape_soak.py): with probability `prob` a replay puts a 0 .. max_us spin kernel in front of a node on its lane, so that the lanes
// drift against each other in a different way in every replay -- a dependency the tape does not carry then shows in tens of replays instead of thousands,
// whatever the durations of the kernels around it happen to be.  Seeded (the sequence of delays is reproducible), off by default.
// (replays may come from several host threads -- data-parallel ranks in one process, tests: the generator state is an atomic, and with fuzzing off -- the default --
// fuzz_delay returns before touching it)
static std::atomic<unsigned> g_fuzz_state{0};
static float g_fuzz_max_us = 0.0f, g_fuzz_prob = 0.0f;
static inline float fuzz_u01() {
    unsigned o = g_fuzz_state.load(std::memory_order_relaxed), n;
    do { n = o * 1664525u + 1013904223u; } while (!g_fuzz_state.compare_exchange_weak(o, n, std::memory_order_relaxed));
    return (float)(n >> 8) * (1.0f / 16777216.0f);
}
extern "C" int vx_tape_set_fuzz(int seed, float max_us, float prob) {
    VX_REQUIRE(max_us >= 0.0f && max_us <= 1e4f && prob >= 0.0f && prob <= 1.0f, "vx_tape_set_fuzz: max_us 0 .. 1e4, prob 0 .. 1");
    g_fuzz_state.store((unsigned)seed * 2654435761u + 12345u, std::memory_order_relaxed);
    g_fuzz_max_us = max_us;
    g_fuzz_prob = prob;
    return 0;
}
static inline void fuzz_delay(hipStream_t s) {
    if (g_fuzz_max_us > 0.0f && fuzz_u01() < g_fuzz_prob)
        hipLaunchKernelGGL(vx_spin_k, dim3(1), dim3(64), 0, s, (long long)(fuzz_u01() * g_fuzz_max_us * 100.0f));
}

static int tape_launch(TapeNode& nd, hipStream_t s) {
    if (nd.type == T_KERNEL) {
        const hipKernelNodeParams& k = nd.k;
        if (nd.mode == 0) {          // captured from a host stub (<<<>>>, hipLaunchKernel) or from a module function (hipModuleLaunchKernel)?
            hipError_t e = k.kernelParams ? hipLaunchKernel(k.func, k.gridDim, k.blockDim, k.kernelParams, k.sharedMemBytes, s) : hipErrorInvalidDeviceFunction;
            if (e == hipSuccess) nd.mode = 1;
            else {
                (void)hipGetLastError();
                HIPQ(hipModuleLaunchKernel((hipFunction_t)k.func, k.gridDim.x, k.gridDim.y, k.gridDim.z, k.blockDim.x, k.blockDim.y, k.blockDim.z, k.sharedMemBytes, s,
                                           k.kernelParams, k.extra), "hipModuleLaunchKernel");
                nd.mode = 2;
            }
        } else if (nd.mode == 1) HIPQ(hipLaunchKernel(k.func, k.gridDim, k.blockDim, k.kernelParams, k.sharedMemBytes, s), "hipLaunchKernel");
        else HIPQ(hipModuleLaunchKernel((hipFunction_t)k.func, k.gridDim.x, k.gridDim.y, k.gridDim.z, k.blockDim.x, k.blockDim.y, k.blockDim.z, k.sharedMemBytes, s,
                                        k.kernelParams, k.extra), "hipModuleLaunchKernel");
    } else if (nd.type == T_MEMSET) {
        const hipMemsetParams& m = nd.ms;
        if (m.elementSize == 1) HIPQ(hipMemsetAsync(m.dst, (int)m.value, m.width, s), "hipMemsetAsync");
        else if (m.elementSize == 2) HIPQ(hipMemsetD16Async((hipDeviceptr_t)m.dst, (unsigned short)m.value, m.width, s), "hipMemsetD16Async");
        else HIPQ(hipMemsetD32Async((hipDeviceptr_t)m.dst, (int)m.value, m.width, s), "hipMemsetD32Async");
    }
    else if (nd.type == T_COPY) HIPQ(hipGraphLaunch(nd.copy_exec, s), "hipGraphLaunch");
    return 0;
}

static int tape_replay_body(VxTape* T, hipStream_t s0, bool fl, unsigned seq, int limit = -1, bool tail = true) {
    const size_t L = T->lanes.size();
    int ni = 0;
    if (L > 1) {
        if (fl) {
            if (T->last_s0 && T->last_s0 != s0) (void)hipStreamSynchronize(T->last_s0);      // another caller stream than last time (rare): its set kernel must not be overtaken
            T->last_s0 = s0;
            flag_set(s0, T->flags + T->flag_start, seq);
            for (size_t l = 0; l < L && l < (size_t)kPool; ++l)
                if (T->lanes[l] != s0) flag_wait(T->lanes[l], (const unsigned*)(T->flags + T->flag_start), seq);
        } else {
            HIPQ(hipEventRecord(T->start, s0), "hipEventRecord");
            for (size_t l = 0; l < L && l < (size_t)kPool; ++l) HIPQ(hipStreamWaitEvent(T->lanes[l], T->start, 0), "hipStreamWaitEvent");
        }
    }
    // cross-lane dependencies: flag kernels (~2 us of queue time per hop) instead of event record + stream wait (~14 us: tools/event_hop_probe.py);
    // see the comment at vx_flag_set_k for why this cannot deadlock and when it is used
    for (TapeNode& nd : T->nodes) {
        if (limit >= 0 && ni++ >= limit) break;            // (vx_tape_replay_prefix: the rest follows on the caller's stream after the join)
        hipStream_t s = T->lanes[nd.lane];
        for (int w : nd.waits) {
            const TapeNode& src = T->nodes[w];
            if (fl && src.flag >= 0) flag_wait(s, (const unsigned*)(T->flags + src.flag), seq);
            else HIPQ(hipStreamWaitEvent(s, src.ev, 0), "hipStreamWaitEvent");
        }
        if (L > 1) fuzz_delay(s);
        if (nd.marker < 0) { int rc = tape_launch(nd, s); if (rc) return rc; }
        if (nd.marker >= 0 || (nd.record && !(fl && nd.flag >= 0))) HIPQ(hipEventRecord(nd.ev, s), "hipEventRecord");
        else if (nd.record) flag_set(s, T->flags + nd.flag, seq);
    }
    if (fl && hipGetLastError() != hipSuccess) VX_FAIL(-2, "vx_tape_replay: a flag kernel could not be launched");
    if (L > 1)
        for (size_t l = 0; l < L; ++l)
            if (T->lane_last[l] >= 0 && T->lanes[l] != s0) {
                if (fl && l < (size_t)kPool) {
                    flag_set(T->lanes[l], T->flags + T->flag_start + 1 + l, seq);
                    flag_wait(s0, (const unsigned*)(T->flags + T->flag_start + 1 + l), seq);
                } else {
                    HIPQ(hipEventRecord(T->lane_end[l], T->lanes[l]), "hipEventRecord");
                    HIPQ(hipStreamWaitEvent(s0, T->lane_end[l], 0), "hipStreamWaitEvent");
                }
            }
    if (limit >= 0 && tail)
        for (size_t i = (size_t)limit; i < T->nodes.size(); ++i)
            if (T->nodes[i].marker < 0) { int rc = tape_launch(T->nodes[i], s0); if (rc) return rc; }
    return 0;
}

static int tape_replay_impl(VxTape* T, void* stream, int limit, bool tail);
// Two tapes that are replayed from different caller streams at the same time (two window batches of a sliding-window inference in flight: utils/inference_runtime.py) queue behind each
// other lane by lane when they use the same pool streams in the same order.  A rotation k sends this tape's lane l to pool stream (l + k) % 4: with k = 2 its two main chains
// (lanes 0, 1) sit on the hardware queues the other tape uses for its side lanes.
extern "C" int vx_tape_set_lane_rotation(VxTape* T, int k) {
    VX_REQUIRE(T && k >= 0 && k < kPool, "vx_tape_set_lane_rotation: 0 .. 3");
    T->lane_rot = k;
    return 0;
}
extern "C" int vx_tape_replay(VxTape* T, void* stream) { return tape_replay_impl(T, stream, -1, true); }      // (-1 = the whole tape)
// diagnostics (tools/tape_memdiff.py): the first `k` nodes the way vx_tape_replay runs them (lanes, cross-lane waits), the lanes joined, then the remaining nodes one after
// the other on the caller's stream -- a binary search over k finds the launch from which on a timing-dependent deviation exists
extern "C" int vx_tape_replay_prefix(VxTape* T, void* stream, int k) {
    VX_REQUIRE(T, "vx_tape_replay_prefix: bad arguments");
    return tape_replay_impl(T, stream, k < 0 ? -k : k, k >= 0);            // (k < 0: the first -k nodes only, nothing after the join)
}
static int tape_replay_impl(VxTape* T, void* stream, int limit, bool tail) {
    VX_REQUIRE(T, "vx_tape_replay: null tape");
    if (T->nodes.empty()) return 0;
    if (int nto = vx_tape_flag_timeouts()) VX_FAIL(-3, "vx_tape_replay: %d cross-lane poll(s) of an earlier replay gave up after the flag timeout (results of that step are not trustworthy)", nto);
    hipStream_t s0 = (hipStream_t)stream;
    const size_t L = T->lanes.size();
    if (L == 1) T->lanes[0] = s0;                 // a single chain runs on the caller's stream
    else {
        int rc = pool_init(s0);
        if (rc) return rc;
        for (size_t l = 0; l < L; ++l) T->lanes[l] = g_pool.lane[(l + (size_t)T->lane_rot) % kPool];      // (lane_rot: vx_tape_set_lane_rotation -- a second tape replayed BESIDE this one takes the other queues for its main chains)
    }
    const bool fl = flags_ok() && T->flags != nullptr;
    const unsigned seq = ++T->seq;
    const int rc = tape_replay_body(T, s0, fl, seq, limit, tail);
    if (rc != 0 && fl) {
        // a launch failed half-way: polls may already be queued whose set kernels never will be.  Release them (every word of this tape >= seq) from a
        // stream of its own, so that the queues drain instead of spinning into the timeout.
        hipStream_t rs = nullptr;
        if (hipStreamCreateWithFlags(&rs, hipStreamNonBlocking) == hipSuccess) {
            hipLaunchKernelGGL(vx_flag_fill_k, dim3(1), dim3(256), 0, rs, T->flags, T->n_flags, seq);
            (void)hipStreamSynchronize(rs);
            (void)hipStreamDestroy(rs);
        }
        (void)hipGetLastError();
    }
    return rc;
}

extern "C" int vx_tape_wait_marker(VxTape* T, int id, void* stream) {
    VX_REQUIRE(T && id >= 0, "vx_tape_wait_marker: bad arguments");
    for (TapeNode& nd : T->nodes)
        if (nd.marker == id) { HIPQ(hipStreamWaitEvent((hipStream_t)stream, nd.ev, 0), "hipStreamWaitEvent"); return 0; }
    VX_FAIL(-1, "vx_tape_wait_marker: the tape holds no marker %d", id);
}

extern "C" int vx_tape_has_marker(const VxTape* T, int id) {      // answer, not a status
    if (!T) return 0;
    for (const TapeNode& nd : T->nodes) if (nd.marker == id) return 1;
    return 0;
}

extern "C" int vx_tape_free(VxTape* T) {
    if (!T) return 0;
    for (auto& nd : T->nodes) {
        if (nd.ev) (void)hipEventDestroy(nd.ev);
        if (nd.copy_exec) (void)hipGraphExecDestroy(nd.copy_exec);
        if (nd.copy_graph) (void)hipGraphDestroy(nd.copy_graph);
    }
    for (size_t l = 0; l < T->lane_end.size(); ++l) if (T->lane_end[l]) (void)hipEventDestroy(T->lane_end[l]);
    if (T->start) (void)hipEventDestroy(T->start);
    if (T->flags) (void)hipFree(T->flags);
    delete T;
    return 0;
}

// ---- introspection (tools/tape_critical_path.py): where does a stage spend its time, and which kernels are on its critical path?
extern "C" int vx_tape_profile(VxTape* T, void* stream, int reps, float* us) {
    VX_REQUIRE(T && us && reps >= 1, "vx_tape_profile: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPQ(hipEventCreate(&e0), "hipEventCreate");
    HIPQ(hipEventCreate(&e1), "hipEventCreate");
    for (size_t i = 0; i < T->nodes.size(); ++i) us[i] = 1e30f;
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i < T->nodes.size(); ++i) {          // launch order, one node at a time: the values computed are those of a replay
            HIPQ(hipEventRecord(e0, s), "hipEventRecord");
            int rc = tape_launch(T->nodes[i], s);
            if (rc) return rc;
            HIPQ(hipEventRecord(e1, s), "hipEventRecord");
            HIPQ(hipEventSynchronize(e1), "hipEventSynchronize");
            float ms = 0.f;
            HIPQ(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
            us[i] = ms * 1e3f < us[i] ? ms * 1e3f : us[i];
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

// One node of the tape on `stream` (markers launch nothing): the building block of the serial re-ordering audit (veloxseg_amd/tape_audit.py), which launches the
// nodes of a step one after the other on ONE stream in a random order that respects lane order + cross-lane waits -- every such order must give the same result.
extern "C" int vx_tape_launch_node(VxTape* T, int i, void* stream) {
    VX_REQUIRE(T && i >= 0 && i < (int)T->nodes.size(), "vx_tape_launch_node: bad arguments");
    TapeNode& nd = T->nodes[i];
    if (nd.marker >= 0) return 0;
    return tape_launch(nd, (hipStream_t)stream);
}
// Raw bytes of launch parameter k of kernel node i (memset nodes: k = 0 -> {dst, width * elementSize} as two 8-byte words).  The graph node owns one host allocation per
// parameter and HIP does not expose the sizes: at most malloc_usable_size() bytes are copied (an upper bound of the parameter's size), *got = how many.  The caller knows
// the parameter list from the kernel's (demangled) name: veloxseg_amd/tape_audit.py node_pointers().
extern "C" int vx_tape_node_param(const VxTape* T, int i, int k, int nparams, int size, void* out, int cap, int* got) {
    VX_REQUIRE(T && out && got && cap > 0 && k >= 0 && i >= 0 && i < (int)T->nodes.size(), "vx_tape_node_param: bad arguments");
    const TapeNode& nd = T->nodes[i];
    *got = 0;
    if (nd.type == T_MEMSET) {
        VX_REQUIRE(k == 0 && cap >= 16, "vx_tape_node_param: a memset node has one pseudo parameter of 16 bytes");
        unsigned long long v[2] = {(unsigned long long)nd.ms.dst, (unsigned long long)nd.ms.width * nd.ms.elementSize};
        memcpy(out, v, 16);
        *got = 16;
        return 0;
    }
    /* the parameter array has as many entries as the kernel's signature (the caller parsed it: nparams); `size` = the parameter's byte size when the caller knows it
       (8 for a pointer).  size 0 = a by-value struct whose size the demangled name does not give: the runtime keeps every captured parameter in a malloc block of its
       own (ROCclr GraphKernelNode::copyParams), so the block size bounds the copy -- diagnostics only (tape_audit.py), never on a launch path */
    VX_REQUIRE(nparams > 0 && k < nparams, "vx_tape_node_param: parameter %d of a kernel with %d parameters", k, nparams);
    if (nd.type != T_KERNEL || !nd.k.kernelParams || !nd.k.kernelParams[k]) return 0;
    size_t n = size > 0 ? (size_t)size : malloc_usable_size(nd.k.kernelParams[k]);
    if (n > (size_t)cap) n = (size_t)cap;
    memcpy(out, nd.k.kernelParams[k], n);
    *got = (int)n;
    return 0;
}
/* 0 kernel, 1 memset, 2 copy (one-node graph), 3 marker */
extern "C" int vx_tape_node_kind(const VxTape* T, int i) {
    if (!T || i < 0 || i >= (int)T->nodes.size()) return -1;
    return T->nodes[i].marker >= 0 ? 3 : T->nodes[i].type;
}

/* waits[i * stride + w]: the nodes (on other lanes) node i waits for, -1 padded */
extern "C" int vx_tape_waits(const VxTape* T, int* waits, int stride) {
    VX_REQUIRE(T && waits && stride >= 1, "vx_tape_waits: bad arguments");
    for (size_t i = 0; i < T->nodes.size(); ++i) {
        const TapeNode& nd = T->nodes[i];
        VX_REQUIRE((int)nd.waits.size() <= stride, "vx_tape_waits: node %d waits for %d nodes, stride %d", (int)i, (int)nd.waits.size(), stride);
        for (int w = 0; w < stride; ++w) waits[i * stride + w] = w < (int)nd.waits.size() ? nd.waits[w] : -1;
    }
    return 0;
}

/* lane[i], grid[i] (workgroups, 0 for non-kernels), waits: up to 4 per node (-1 padded), names: name_stride bytes per node */
extern "C" int vx_tape_describe(const VxTape* T, int* lane, int* grid, int* waits4, char* names, int name_stride) {
    VX_REQUIRE(T && lane && grid && waits4 && names && name_stride >= 8, "vx_tape_describe: bad arguments");
    for (size_t i = 0; i < T->nodes.size(); ++i) {
        const TapeNode& nd = T->nodes[i];
        lane[i] = nd.lane;
        grid[i] = nd.type == T_KERNEL ? (int)(nd.k.gridDim.x * nd.k.gridDim.y * nd.k.gridDim.z) : 0;
        for (int w = 0; w < 4; ++w) waits4[4 * i + w] = w < (int)nd.waits.size() ? nd.waits[w] : -1;
        const char* nm = nd.type == T_MEMSET ? "memset" : nd.type == T_COPY ? "memcpy" : nullptr;
        if (!nm) { nm = nd.mode == 2 ? hipKernelNameRef((hipFunction_t)nd.k.func) : hipKernelNameRefByPtr(nd.k.func, nullptr); if (!nm) nm = "?"; }
        snprintf(names + (size_t)i * name_stride, name_stride, "%s", nm);
    }
    return 0;
}
